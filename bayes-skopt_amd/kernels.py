"""Kernel objects on the host and their mapping to the device's canonical hyper-parameter vector.

The reference builds its GP kernels from skopt's kernel classes, which are scikit-learn's kernels
plus ``gradient_x`` (``bask/utils.py:6``, ``bask/bayesgpr.py:12``); scikit-learn ships in the image,
skopt does not, so the host-side kernel *objects* (theta get/set in log space, bounds,
``clone_with_theta``; ``sklearn/kernels.py:285-338, 733-760``) are scikit-learn's.  No kernel
*arithmetic* runs on the host for the canonical forms: a kernel expression tree is analysed once into a
``KernelPlan`` that maps its ``theta`` to the canonical device vector ``h = [log c, log l_1..log l_d, log s2]``
(``include/bgp.h``), and every K(X, X) / K(X*, X) is evaluated by the HIP kernels.

Canonical expression trees (device Gram build):
    [ConstantKernel *] S(length_scale) [+ WhiteKernel]        form "product"
    [ConstantKernel +] S(length_scale) [+ WhiteKernel]        form "sum"
with S in {RBF, Matern(nu = 0.5 | 1.5 | 2.5)}, isotropic or anisotropic, in any operand order, any
hyper-parameter optionally "fixed".

Every OTHER tree scikit-learn can evaluate (two stationary terms, products of stationaries, general Matern nu,
RationalQuadratic, ExpSineSquared, DotProduct, ... -- the reference accepts any kernel, ``bask/bayesgpr.py:148-159``)
gets a ``GramPlan``: the host evaluates ``kernel_(X)`` with the kernel object, exactly as the reference does
(``sklearn/_gpr.py:582``), and the device does the factorisation, the solves, the log-likelihood, the inverse and
the predictive products (``bgp_lml_batch_gram`` / ``bgp_posterior_batch_gram`` / ``bgp_predict_batch_gram``).
There is still no CPU path for the factorisation.
"""
import numpy as np
from sklearn.gaussian_process.kernels import (  # noqa: F401  (re-exported: the bask kernel vocabulary)
    RBF,
    ConstantKernel,
    Kernel,
    Matern,
    Product,
    Sum,
    WhiteKernel,
)

__all__ = ["RBF", "ConstantKernel", "Matern", "WhiteKernel", "Sum", "Product", "KernelPlan", "GramPlan", "analyse_kernel",
           "param_for_white_kernel_in_sum", "gradient_x"]


def gradient_x(kernel, x, X_train):
    """``d k(x, X_train_i) / d x`` for every training point: (n, d).  The method skopt's kernel classes add to scikit-learn's
    (``kernel_.gradient_x(X[0], self.X_train_)`` in skopt's ``predict(return_mean_grad=True)``, the routine
    ``bask/bayesgpr.py:633`` forwards to, used by ``bask/optimizer.py:494`` through ``expected_minimum``), restated for
    scikit-learn kernel objects by recursion over the expression tree: Sum -> sum of the children's gradients, Product ->
    product rule with the children's cross-covariance rows, Exponentiation -> chain rule; leaves RBF, Matern (0.5 / 1.5 / 2.5 in
    closed form as skopt has them, any other nu through ``d/dz [z^nu K_nu(z)] = -z^nu K_{nu-1}(z)``), RationalQuadratic,
    ExpSineSquared, DotProduct; ConstantKernel and WhiteKernel contribute zeros."""
    from scipy.special import gamma, kv
    from sklearn.gaussian_process.kernels import DotProduct, Exponentiation, ExpSineSquared, RationalQuadratic

    x = np.asarray(x, dtype=np.float64).ravel()
    X_train = np.atleast_2d(np.asarray(X_train, dtype=np.float64))
    n, d = X_train.shape
    if isinstance(kernel, Sum):
        return gradient_x(kernel.k1, x, X_train) + gradient_x(kernel.k2, x, X_train)
    if isinstance(kernel, Product):
        f = kernel.k1(x[None, :], X_train)[0]
        g = kernel.k2(x[None, :], X_train)[0]
        return f[:, None] * gradient_x(kernel.k2, x, X_train) + g[:, None] * gradient_x(kernel.k1, x, X_train)
    if isinstance(kernel, Exponentiation):
        base = kernel.kernel(x[None, :], X_train)[0]
        return (kernel.exponent * base ** (kernel.exponent - 1.0))[:, None] * gradient_x(kernel.kernel, x, X_train)
    if isinstance(kernel, (ConstantKernel, WhiteKernel)):
        return np.zeros((n, d))
    if isinstance(kernel, DotProduct):
        return X_train.copy()
    if isinstance(kernel, ExpSineSquared):
        diff = x[None, :] - X_train
        dist = np.sqrt(np.sum(diff * diff, axis=1))
        arg = np.pi * dist / kernel.periodicity
        k = np.exp(-2.0 * (np.sin(arg) / kernel.length_scale) ** 2)
        with np.errstate(divide="ignore", invalid="ignore"):
            # d/dx exp(-2 sin^2(pi r / p) / l^2) = k * (-2 sin(2 pi r / p) / l^2) * (pi / p) * diff / r
            fac = np.where(dist > 0, k * (-2.0 * np.sin(2.0 * arg) / kernel.length_scale**2) * (np.pi / kernel.periodicity) / dist, 0.0)
        return fac[:, None] * diff
    if isinstance(kernel, (Matern, RBF, RationalQuadratic)):
        ell = np.broadcast_to(np.asarray(kernel.length_scale, dtype=np.float64), (d,))
        diff = (x[None, :] - X_train) / ell           # (x - X_i) / l
        r2 = np.sum(diff * diff, axis=1)
        if isinstance(kernel, RationalQuadratic):
            fac = -((1.0 + r2 / (2.0 * kernel.alpha)) ** (-kernel.alpha - 1.0))
        elif isinstance(kernel, Matern) and not np.isinf(kernel.nu):
            r = np.sqrt(r2)
            nu = float(kernel.nu)
            with np.errstate(divide="ignore", invalid="ignore"):
                if nu == 0.5:
                    fac = np.where(r > 0, -np.exp(-r) / r, 0.0)
                elif nu == 1.5:
                    fac = -3.0 * np.exp(-np.sqrt(3.0) * r)
                elif nu == 2.5:
                    fac = -(5.0 / 3.0) * (1.0 + np.sqrt(5.0) * r) * np.exp(-np.sqrt(5.0) * r)
                else:
                    # k = c z^nu K_nu(z), z = sqrt(2 nu) r, c = 2^(1 - nu) / Gamma(nu);  dk/dr = -c sqrt(2 nu) z^nu K_{nu-1}(z)
                    z = np.sqrt(2.0 * nu) * r
                    dk_dr = -(2.0 ** (1.0 - nu) / gamma(nu)) * np.sqrt(2.0 * nu) * z**nu * kv(nu - 1.0, z)
                    fac = np.where(r > 0, dk_dr / r, 0.0)
                    if nu > 1.0:  # (the limit r -> 0 of (dk/dr) / r is finite for nu > 1: -nu / (nu - 1))
                        fac = np.where(r > 0, fac, -nu / (nu - 1.0))
        else:  # RBF (and Matern with nu = inf)
            fac = -np.exp(-0.5 * r2)
        return fac[:, None] * diff / ell
    raise NotImplementedError("gradient_x is not implemented for %s" % type(kernel).__name__)


def param_for_white_kernel_in_sum(kernel, kernel_str=""):
    """Locate a WhiteKernel inside (nested) Sum kernels; returns (present, param_name).

    Same contract as skopt's ``_param_for_white_kernel_in_Sum`` imported at
    ``bask/bayesgpr.py:9-11`` and used at ``:328-333``.
    """
    if kernel_str != "":
        kernel_str = kernel_str + "__"
    if isinstance(kernel, Sum):
        for param, child in kernel.get_params(deep=False).items():
            if isinstance(child, WhiteKernel):
                return True, kernel_str + param
            present, child_str = param_for_white_kernel_in_sum(child, kernel_str + param)
            if present:
                return True, child_str
    return False, "_"


def _stationary_name(k):
    if isinstance(k, Matern):  # NB: sklearn's Matern subclasses RBF -- test it first
        nu = float(k.nu)
        for val, name in ((0.5, "matern12"), (1.5, "matern32"), (2.5, "matern52")):
            if nu == val:
                return name
        if np.isinf(nu):
            return "rbf"
        raise NotImplementedError(f"Matern nu={nu} has no device kernel (supported: 0.5, 1.5, 2.5, inf)")
    if isinstance(k, RBF):
        return "rbf"
    return None


class GramPlan:
    """A kernel expression tree without a canonical device form: its matrices are evaluated on the host with the
    scikit-learn kernel object (what the reference does for every kernel, ``sklearn/_gpr.py:582``) and handed to
    the device, which does everything behind them.  ``why`` keeps the reason the canonical analysis gave."""

    generic = True
    form, stationary = "product", "matern52"  # (what the device context is created with; its Gram build is never used)

    def __init__(self, n_theta, why):
        self.n_theta = n_theta
        self.why = why

    def canonical(self, theta, d):
        raise NotImplementedError("this kernel has no canonical device form (%s): its matrices come from the host" % self.why)


class KernelPlan:
    """theta (p,) of one kernel expression tree  <->  canonical h (d+2,)."""

    generic = False

    def __init__(self, form, stationary, const, ell, white, n_theta, ard):
        self.form = form  # "product" | "sum"
        self.stationary = stationary
        # each of const / white: ("free", theta_index) | ("fixed", log_value) | ("absent", log_value)
        # ell: ("free", [theta indices]) | ("fixed", log_values array)
        self.const, self.ell, self.white = const, ell, white
        self.n_theta = n_theta
        self.ard = ard
        # theta IS the canonical vector already (every component free, anisotropic length scales, theta order = h order: the
        # reference's default kernel, bask/utils.py:144-150 + WhiteKernel): the sampler's per-half-step mapping is a no-op
        self._identity_d = None
        if const[0] == "free" and white[0] == "free" and ell[0] == "free" and len(ell[1]) > 1:
            idx = [const[1]] + list(ell[1]) + [white[1]]
            if idx == list(range(n_theta)):
                self._identity_d = len(ell[1])

    def canonical(self, theta, d):
        """(B, p) or (p,) theta -> (B, d+2) canonical vectors."""
        T = np.atleast_2d(np.asarray(theta, dtype=np.float64))
        if T.shape[1] != self.n_theta:
            raise ValueError(f"theta has {T.shape[1]} entries, kernel has {self.n_theta} free hyper-parameters")
        if self._identity_d == d:
            return np.ascontiguousarray(T)
        B = T.shape[0]
        H = np.empty((B, d + 2))
        kind, v = self.const
        H[:, 0] = T[:, v] if kind == "free" else v
        kind, v = self.ell
        if kind == "free":
            if len(v) == 1:
                H[:, 1 : d + 1] = T[:, v[0]][:, None]
            else:
                if len(v) != d:
                    raise ValueError(f"anisotropic length scale has {len(v)} entries but X has {d} columns")
                H[:, 1 : d + 1] = T[:, v]
        else:
            v = np.asarray(v, dtype=np.float64)
            if v.size not in (1, d):
                raise ValueError(f"anisotropic length scale has {v.size} entries but X has {d} columns")
            H[:, 1 : d + 1] = v
        kind, v = self.white
        H[:, d + 1] = T[:, v] if kind == "free" else v
        return H

    def grad_to_theta(self, G, d):
        """Chain rule: dLML/dh (B, d+2) -> dLML/dtheta (B, p) (fixed entries dropped, isotropic
        length scale = sum over the replicated columns)."""
        G = np.atleast_2d(G)
        out = np.zeros((G.shape[0], self.n_theta))
        if self.const[0] == "free":
            out[:, self.const[1]] += G[:, 0]
        if self.ell[0] == "free":
            idx = self.ell[1]
            if len(idx) == 1:
                out[:, idx[0]] += G[:, 1 : d + 1].sum(axis=1)
            else:
                out[:, idx] += G[:, 1 : d + 1]
        if self.white[0] == "free":
            out[:, self.white[1]] += G[:, d + 1]
        return out


def _flatten(kernel, cls):
    if isinstance(kernel, cls):
        return _flatten(kernel.k1, cls) + _flatten(kernel.k2, cls)
    return [kernel]


def _leaves_in_theta_order(kernel, start=0):
    """Leaves of the tree in ``KernelOperator.theta`` order (k1 then k2, kernels.py:747) with the
    index of their first free hyper-parameter in theta."""
    if isinstance(kernel, (Sum, Product)):
        left, nxt = _leaves_in_theta_order(kernel.k1, start)
        right, nxt = _leaves_in_theta_order(kernel.k2, nxt)
        return left + right, nxt
    n_free = int(kernel.n_dims)
    return [(kernel, start, n_free)], start + n_free


def analyse_kernel(kernel, strict=False):
    """Analyse a kernel expression tree: a ``KernelPlan`` for the canonical forms (device Gram build), a ``GramPlan``
    (host-evaluated kernel matrices, device arithmetic) for every other tree scikit-learn can evaluate
    (``bask/bayesgpr.py:148-159`` accepts any kernel).  ``strict=True`` raises ``NotImplementedError`` instead of
    returning a ``GramPlan``."""
    if not isinstance(kernel, Kernel):
        raise TypeError(f"expected a scikit-learn kernel object, got {type(kernel)}")
    try:
        return _analyse_canonical(kernel)
    except NotImplementedError as exc:
        if strict:
            raise
        with np.errstate(divide="ignore"):
            return GramPlan(len(kernel.theta), str(exc))


def _analyse_canonical(kernel):
    with np.errstate(divide="ignore"):  # a zeroed WhiteKernel has theta = log(0)
        leaves, n_theta = _leaves_in_theta_order(kernel)
    index_of = {id(k): (i0, nf) for k, i0, nf in leaves}

    def slot(k):
        i0, nf = index_of[id(k)]
        return i0, nf

    terms = _flatten(kernel, Sum)
    white = ("absent", -np.inf)
    const_add = None
    stat_term = None
    for t in terms:
        if isinstance(t, WhiteKernel):
            if white[0] != "absent":
                raise NotImplementedError("more than one WhiteKernel in the kernel")
            i0, nf = slot(t)
            white = ("free", i0) if nf == 1 else ("fixed", float(np.log(t.noise_level)) if t.noise_level > 0 else -np.inf)
        elif isinstance(t, ConstantKernel):
            if const_add is not None:
                raise NotImplementedError("more than one additive ConstantKernel")
            const_add = t
        else:
            if stat_term is not None:
                raise NotImplementedError(f"more than one non-constant term in the sum: {kernel}")
            stat_term = t
    if stat_term is None:
        raise NotImplementedError(f"kernel has no stationary (RBF/Matern) component: {kernel}")

    factors = _flatten(stat_term, Product)
    const_mul = None
    stat = None
    for f in factors:
        if isinstance(f, ConstantKernel):
            if const_mul is not None:
                raise NotImplementedError("more than one multiplicative ConstantKernel")
            const_mul = f
        elif _stationary_name(f) is not None:
            if stat is not None:
                raise NotImplementedError("product of two stationary kernels is not supported")
            stat = f
        else:
            raise NotImplementedError(f"unsupported kernel component {type(f).__name__} in {kernel}")
    if stat is None:
        raise NotImplementedError(f"kernel has no stationary (RBF/Matern) component: {kernel}")
    if const_mul is not None and const_add is not None:
        raise NotImplementedError("kernel with both a multiplicative and an additive constant is not supported")

    def const_slot(ck):
        i0, nf = slot(ck)
        return ("free", i0) if nf == 1 else ("fixed", float(np.log(ck.constant_value)))

    if const_add is not None:
        form, const = "sum", const_slot(const_add)
    elif const_mul is not None:
        form, const = "product", const_slot(const_mul)
    else:
        form, const = "product", ("absent", 0.0)  # c = 1

    i0, nf = slot(stat)
    ard = bool(stat.anisotropic)
    if nf > 0:
        ell = ("free", list(range(i0, i0 + nf)))
    else:
        ell = ("fixed", np.log(np.atleast_1d(np.asarray(stat.length_scale, dtype=np.float64))))
    return KernelPlan(form, _stationary_name(stat), const, ell, white, n_theta, ard)
