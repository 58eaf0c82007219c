// Posterior build, predict, LML gradient, PVRS and sample_y.
//
// Posterior build (the BayesGPR.theta setter, bask/bayesgpr.py:200-217): the batched Cholesky of
// bgp_chol.hip is run on the augmented matrix [[K, .], [I, 0]] for nblk steps, which leaves
//   top-left      L                      -> BayesGPR.L_
//   bottom-right  -K^-1 (Schur complement)-> BayesGPR.K_inv_   (the reference forms the explicit
//                                            inverse too: L_inv.dot(L_inv.T), :207-208)
//   rhs, lower    -alpha = -(K^-1 y)      -> BayesGPR.alpha_
// K^-1 and alpha of every posterior stay resident in HBM for the predict / pvrs / gradient calls.
//
// Predict (bask/bayesgpr.py:622-635 -> skopt predict), per resident posterior:
//   K_* = k(Xq, X)  (tiled cross-kernel build);  mean = K_* alpha;
//   var = k_** - rowsum((K_* K^-1) o K_*)   -- one NT tile GEMM on the fp64 MFMA with the row-dot
//   fused into its epilogue (K^-1 symmetric, so K_* K^-1 = K_* (K^-1)^T is an NT product);
//   cov = K_** - (K_* K^-1) K_*^T.
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__global__ void aug_init_kernel(double* __restrict__ Kbuf, double* __restrict__ yw, int npad, int B) {
  // identity into the bottom-left block of every augmented matrix, zero the lower half of the rhs
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= npad || b >= B) return;
  const size_t ld = 2 * (size_t)npad;
  Kbuf[(size_t)b * ld * ld + (size_t)(npad + i) * ld + i] = 1.0;
  yw[(size_t)b * ld + npad + i] = 0.0;
}

// K^-1 (full symmetric npad x npad) and alpha out of the augmented workspace.
__global__ void __launch_bounds__(256) extract_kinv_kernel(const double* __restrict__ Kbuf,
                                                            const double* __restrict__ yw,
                                                            double* __restrict__ Kinv, double* __restrict__ alpha,
                                                            int npad, int boff) {
  const int b = blockIdx.y;
  const size_t ld = 2 * (size_t)npad;
  const double* M = Kbuf + (size_t)b * ld * ld;
  double* out = Kinv + (size_t)(boff + b) * npad * npad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)npad * npad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / npad), j = (int)(idx - (size_t)i * npad);
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    out[idx] = -M[(size_t)(npad + hi) * ld + npad + lo];
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < npad; i += blockDim.x)
      alpha[(size_t)(boff + b) * npad + i] = -yw[(size_t)b * ld + npad + i];
}

// compact n x n lower factor (zeros above the diagonal) of matrix b into scratch
__global__ void extract_L_kernel(const double* __restrict__ Kbuf, double* __restrict__ out, int n, int ld,
                                 size_t mstride, int b) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)n * n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    out[idx] = (j <= i) ? Kbuf[(size_t)b * mstride + (size_t)i * ld + j] : 0.0;
  }
}

// Batched launches below run item b = blockIdx.y of a chunk; operands that live in the resident posterior arrays
// (K^-1, alpha) are addressed through `pidx[b]` (NULL: item b is posterior b0 + b ... callers pass explicit maps).
// mean_i = sum_j Ks[i][j] alpha[j]   (one wave per row)
__global__ void __launch_bounds__(256) matvec_rows_kernel(const double* __restrict__ Ks, int lds, size_t sK,
                                                           const double* __restrict__ v, size_t sv,
                                                           const int* __restrict__ pidx, int n, int m,
                                                           double* __restrict__ out, size_t so) {
  const int b = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* K = Ks + (size_t)b * sK;
  const double* vb = v + (size_t)(pidx ? pidx[b] : b) * sv;
  double s = 0.0;
  for (int j = lane; j < n; j += 64) s += K[(size_t)row * lds + j] * vb[j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[(size_t)b * so + row] = s;
}

// kernel_.diag(X) incl. the white level (sklearn/kernels.py:868-884, 968-984) from a canonical vector on the device
static __device__ __forceinline__ double dev_kernel_diag(const double* hk, int d, int form) {
  const double cst = exp(hk[0]), s2 = exp(hk[d + 1]);
  return ((form == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0) + s2;
}

// var_i = max(0, diag_b - q_i)
__global__ void finish_var_kernel(const double* __restrict__ q, size_t sq, const double* __restrict__ H, int d, int form,
                                  int m, double* __restrict__ var, size_t so) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= m) return;
  const double v = dev_kernel_diag(H + (size_t)b * (d + 2), d, form) - q[(size_t)b * sq + i];
  var[(size_t)b * so + i] = v < 0.0 ? 0.0 : v;
}

// Ordered sum of the column-tile partials of a batched row reduction (row dots of rowquad4_kernel, means of
// kbuild_cross_kernel): out[b][i] = sum_t part[(b tn + t) M + i], t ascending -- no floating-point atomics.
__global__ void rowdot_reduce_kernel(const double* __restrict__ part, int tn, int M, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= M) return;
  double s = 0.0;
  for (int t = 0; t < tn; t++) s += part[((size_t)b * tn + t) * M + i];
  out[(size_t)b * M + i] = s;  // (batched row dots are packed with stride M)
}

// column-tile partials of the EPI-1 row dots (own buffer: the callers' scratch layouts stay as they are)
static int ensure_rowpart(bgp_ctx* c, size_t doubles) {
  if (doubles > c->cap_rowpart) {
    if (c->drowpart) (void)hipFree(c->drowpart);
    c->drowpart = nullptr;
    c->cap_rowpart = 0;
    BGP_HIP(hipMalloc(&c->drowpart, doubles * sizeof(double)));
    c->cap_rowpart = doubles;
  }
  return BGP_OK;
}

__global__ void add_diag_kernel(double* __restrict__ C, int ld, int m, double v);

void bgp_launch_rowquad(hipStream_t st, const double* A, int lda, size_t sA, const double* S, int lds_, size_t sS,
                        const int* pidx, int M, int n, int nb, double* part);
// NT products on the LDS-DMA ring (bgp_syrk4.hip): mode 0: C = A B^T; mode 1: C -= A B^T on the lower tiles of a square C
void bgp_launch_gemm4(hipStream_t st, int mode, const double* A, const double* Bm, int ldx, int M, int N, int K, double* C,
                      int ldc, int nb, size_t sA, size_t sB, size_t sC, const int* pidxB);
int bgp_rowquad_tile();

// out[b][i] = a_i^T S_b a_i for the rows of A_b (M x n, zero padded; S symmetric): half-product on the LDS-DMA ring
// (rowquad4_kernel, bgp_syrk4.hip) + ordered reduction of the column-tile partials.  out is packed with stride M.
static int launch_rowquad(bgp_ctx* c, const double* A, int lda, size_t sA, const double* S, int lds_, size_t sS,
                          const int* pidx, int M, int n, int nb, double* out) {
  const int tn = n / bgp_rowquad_tile();
  int rc = ensure_rowpart(c, (size_t)nb * tn * M);
  if (rc) return rc;
  bgp_launch_rowquad(c->stream, A, lda, sA, S, lds_, sS, pidx, M, n, nb, c->drowpart);
  BGP_HIP(hipGetLastError());
  hipLaunchKernelGGL(rowdot_reduce_kernel, dim3((M + 255) / 256, nb), dim3(256), 0, c->stream, c->drowpart, tn, M, out);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

static inline int pad128(int v) { return ((v + 127) / 128) * 128; }

// ------------------------------------------------------------------------------------------
// posterior build
// ------------------------------------------------------------------------------------------
static int ensure_resident(bgp_ctx* c, int B) {
  // K^-1 needs B npad^2 doubles, alpha B npad: the two capacities are tracked separately (a context reused through
  // bgp_ctx_update_data may see n shrink and B grow: B=1 at npad=1024 and B=64 at npad=128 need the same K^-1
  // bytes but 8x the alpha bytes)
  const size_t need = (size_t)B * c->npad * c->npad, need_a = (size_t)B * c->npad;
  if (need > c->cap_kinv) {
    if (c->dKinv) (void)hipFree(c->dKinv);
    c->dKinv = nullptr;
    c->cap_kinv = 0;
    BGP_HIP(hipMalloc(&c->dKinv, need * sizeof(double)));
    c->cap_kinv = need;
  }
  if (need_a > c->cap_alpha) {
    if (c->dalpha_sol) (void)hipFree(c->dalpha_sol);
    c->dalpha_sol = nullptr;
    c->cap_alpha = 0;
    BGP_HIP(hipMalloc(&c->dalpha_sol, need_a * sizeof(double)));
    c->cap_alpha = need_a;
  }
  return BGP_OK;
}

int bgp_posterior_build(bgp_ctx* c, int B, const double* h, int use_alpha, double* L, double* alpha, double* K_inv,
                        double* lml, int* status, const double* Kgram) {
  const int npad = c->npad, n = c->n;
  const size_t p = c->d + 2;
  const size_t ld = 2 * (size_t)npad;
  int rc = ensure_resident(c, B);
  if (rc) return rc;
  // augmented matrices are 4x the LML workspace per item
  int chunk = (int)(c->cap_mat / (ld * ld));
  if (chunk < 1) {
    rc = bgp_grow_workspace(c, ld * ld);
    if (rc) return rc;
    chunk = 1;
  }
  chunk = std::min(chunk, c->max_batch);
  for (int off = 0; off < B; off += chunk) {
    const int nb = std::min(chunk, B - off);
    if (!Kgram)
      BGP_HIP(bgp_memcpy_async(c->dh, h + (size_t)off * p, nb * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BGP_HIP(hipMemsetAsync(c->dstatus, 0, nb * sizeof(int), c->stream));
    // zero the bottom halves ([I | 0] rows), then K into the top-left, identity, rhs
    for (int b = 0; b < nb; b++)
      BGP_HIP(hipMemsetAsync(c->dK + (size_t)b * ld * ld + (size_t)npad * ld, 0, (size_t)npad * ld * sizeof(double),
                             c->stream));
    rc = Kgram ? bgp_gram_load(c, nb, Kgram + (size_t)off * n * n, 1, use_alpha) : bgp_launch_kbuild(c, nb, 0, 1, use_alpha);
    if (rc) return rc;
    hipLaunchKernelGGL(aug_init_kernel, dim3((npad + 255) / 256, nb), dim3(256), 0, c->stream, c->dK, c->dyw, npad, nb);
    rc = bgp_launch_cholesky(c, nb, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(extract_kinv_kernel, dim3(256, nb), dim3(256), 0, c->stream, c->dK, c->dyw, c->dKinv,
                       c->dalpha_sol, npad, off);
    BGP_HIP(hipGetLastError());
    if (lml) BGP_HIP(bgp_memcpy_async(lml + off, c->dlml, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) BGP_HIP(bgp_memcpy_async(status + off, c->dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (alpha)
      BGP_HIP(bgp_memcpy2d_async(alpha + (size_t)off * n, (size_t)n * sizeof(double),
                               c->dalpha_sol + (size_t)off * npad, (size_t)npad * sizeof(double),
                               (size_t)n * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    if (K_inv)
      for (int b = 0; b < nb; b++)
        BGP_HIP(bgp_memcpy2d_async(K_inv + (size_t)(off + b) * n * n, (size_t)n * sizeof(double),
                                 c->dKinv + (size_t)(off + b) * npad * npad, (size_t)npad * sizeof(double),
                                 (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
    if (L) {
      rc = bgp_ensure_scratch(c, (size_t)n * n);
      if (rc) return rc;
      for (int b = 0; b < nb; b++) {
        hipLaunchKernelGGL(extract_L_kernel, dim3(512), dim3(256), 0, c->stream, c->dK, c->dscratch, n, (int)ld,
                           ld * ld, b);
        BGP_HIP(bgp_memcpy_async(L + (size_t)(off + b) * n * n, c->dscratch, (size_t)n * n * sizeof(double),
                               hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(bgp_stream_sync(c->stream));
      }
    }
    BGP_HIP(bgp_stream_sync(c->stream));
  }
  c->post_B = B;
  return BGP_OK;
}

extern "C" int bgp_posterior_batch(bgp_ctx* c, int B, const double* h, double* L, double* alpha, double* K_inv,
                                   double* lml, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_posterior_batch");
  if (!c || !h || B <= 0) {
    bgp_set_error("bgp_posterior_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  return bgp_posterior_build(c, B, h, 1, L, alpha, K_inv, lml, status);
}

// The same posterior build from HOST-evaluated kernel matrices (generic kernel expression trees, bgp_gram.hip): K is B
// matrices of n x n (kernel_(X_train) without the alpha term), everything behind it as bgp_posterior_batch.
extern "C" int bgp_posterior_batch_gram(bgp_ctx* c, int B, const double* K, int use_alpha, double* L, double* alpha,
                                        double* K_inv, double* lml, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_posterior_batch_gram");
  if (!c || !K || B <= 0) {
    bgp_set_error("bgp_posterior_batch_gram: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  return bgp_posterior_build(c, B, nullptr, use_alpha, L, alpha, K_inv, lml, status, K);
}

// ------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------
// scratch layout helper
struct Scratch {
  double* base;
  size_t used;
  double* take(size_t n) {
    double* p = base + used;
    used += (n + 1) & ~(size_t)1;  // keep 16-byte alignment
    return p;
  }
};

static double kernel_diag_value(const bgp_ctx* c, const double* hk) {
  // kernel_.diag(X) incl. the white level: sklearn/kernels.py:868-884, 968-984
  const double cst = std::exp(hk[0]), s2 = std::exp(hk[c->d + 1]);
  const double base = (c->ks.form == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
  return base + s2;
}

__global__ void add_diag_batch_kernel(double* __restrict__ C, int ld, size_t sC, int m, const double* __restrict__ H,
                                      int d) {
  // + white level of item b on the diagonal of its m x m block (kernel_(X) of a Sum with WhiteKernel)
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i < m) C[(size_t)b * sC + (size_t)i * ld + i] += exp(H[(size_t)b * (d + 2) + d + 1]);
}

// Posteriors are processed in chunks of `nb` items whose scratch slices sit side by side: every launch covers the
// whole chunk (grid.y = item), nothing synchronises inside a chunk, and the host waits once at the end of the call.
// Results per item are the same bits as an item-by-item loop (each item's tiles are reduced in the same order).
struct AcqPlan {  // closed-form acquisitions evaluated on the device-resident mean / variance (bgp_acq_batch)
  int n_acq = 0;
  const int* kinds = nullptr;
  const double* params = nullptr;
  double y_mean = 0.0, y_std = 1.0;
  int n_samples = 1;
  double* out = nullptr;  // host, n_acq * m
};
static int acq_run(bgp_ctx* c, hipStream_t st, int B, int m, int mpad, const double* dmean, const double* dvar,
                   const AcqPlan& ap, double* dT, int* dbad, double* dmumin, double* dacc);

static int predict_run(bgp_ctx* c, int B, const double* h_kernel, int m, const double* Xq, double* mean, double* var,
                       double* cov, const AcqPlan* ap) {
  if (B > c->post_B) {
    bgp_set_error("bgp_predict_batch: %d posteriors requested but %d resident (call bgp_posterior_batch first)", B,
                  c->post_B);
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m);
  const size_t p = d + 2;
  const size_t per_item = (size_t)mpad * npad + 2 * (size_t)mpad + (cov ? (size_t)mpad * npad + (size_t)mpad * mpad : 0);
  const size_t budget = (size_t)1 << 30;  // doubles of scratch per chunk (8 GiB of the 288 GB)
  int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, budget / per_item));
  if (chunk >= 8 && chunk < B) chunk &= ~7;  // whole rounds of the item -> XCD pinning (rowquad4_kernel)
  const int n_acq = ap ? ap->n_acq : 0;
  size_t need = (size_t)m * d + 2 + (size_t)B * p + 2 + (size_t)chunk * per_item + 2 * (size_t)B * mpad + 64 + (size_t)chunk * (npad / 128) * mpad +
                (size_t)n_acq * ((size_t)B * mpad + (size_t)B + mpad) + 2 * (size_t)B + 64;
  int rc = bgp_ensure_scratch(c, need);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dXq = s.take((size_t)m * d);
  double* dH = s.take((size_t)B * p);
  double* dKs = s.take((size_t)chunk * mpad * npad);
  double* dmpart = s.take((size_t)chunk * (npad / 128) * mpad);  // column-tile partials of the means
  double* dqB = s.take((size_t)B * mpad);    // variance of every item (stays on the device for the acquisitions)
  double* doutB = s.take((size_t)B * mpad);  // mean of every item
  double *dP = nullptr, *dCov = nullptr;
  if (cov) {
    dP = s.take((size_t)chunk * mpad * npad);
    dCov = s.take((size_t)chunk * mpad * mpad);
  }
  const size_t sKs = (size_t)mpad * npad, sCv = (size_t)mpad * mpad;
  BGP_HIP(bgp_memcpy_async(dXq, Xq, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dH, h_kernel, (size_t)B * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->has_warp) {  // BayesGPR.predict warps the query points with the current warpers (bask/bayesgpr.py:630-632)
    rc = bgp_launch_warp(c, c->stream, dXq, c->dwarp, dXq, m, 1, 0);
    if (rc) return rc;
  }
  for (int off = 0; off < B; off += chunk) {
    const int nb = std::min(chunk, B - off);
    const double* dHc = dH + (size_t)off * p;
    const double* Kinv = c->dKinv + (size_t)off * npad * npad;  // items off .. off+nb-1 are posteriors off .. off+nb-1
    const double* al = c->dalpha_sol + (size_t)off * npad;
    double *dq = dqB + (size_t)off * mpad, *dout = doutB + (size_t)off * mpad;
    // K_* and, from the same registers, the column-tile partials of the mean K_* alpha (added in tile order below)
    rc = bgp_launch_kcross_matvec(c, nb, dHc, m, dXq, n, c->dXeff, dKs, npad, sKs, al, (size_t)npad, dmpart);
    if (rc) return rc;
    hipLaunchKernelGGL(rowdot_reduce_kernel, dim3((mpad + 255) / 256, nb), dim3(256), 0, c->stream, dmpart, npad / 128, mpad,
                       dout);
    if (mean)
      BGP_HIP(bgp_memcpy2d_async(mean + (size_t)off * m, (size_t)m * sizeof(double), dout, (size_t)mpad * sizeof(double),
                               (size_t)m * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    // q_i = k_i^T K^-1 k_i  (= rowsum((K_* K^-1) o K_*), evaluated on the lower block triangle of K^-1)
    rc = launch_rowquad(c, dKs, npad, sKs, Kinv, npad, (size_t)npad * npad, nullptr, mpad, npad, nb, dq);
    if (rc) return rc;
    if (cov) {
      bgp_launch_gemm4(c->stream, 0, dKs, Kinv, npad, mpad, npad, npad, dP, npad, nb, sKs, (size_t)npad * npad, sKs, nullptr);
      // K_** (no white noise off the diagonal; the diagonal gets c(+1) + s2 like kernel_(X)); then cov = K_** - P K_*^T
      // in place (every element is read and written by the same lane)
      rc = bgp_launch_kcross_batch(c, nb, dHc, m, dXq, m, dXq, dCov, mpad, sCv);
      if (rc) return rc;
      hipLaunchKernelGGL(add_diag_batch_kernel, dim3((m + 255) / 256, nb), dim3(256), 0, c->stream, dCov, mpad, sCv, m,
                         dHc, d);
      bgp_launch_gemm4(c->stream, 2, dP, dKs, npad, mpad, mpad, npad, dCov, mpad, nb, sKs, sKs, sCv, nullptr);
      for (int b = 0; b < nb; b++)
        BGP_HIP(bgp_memcpy2d_async(cov + (size_t)(off + b) * m * m, (size_t)m * sizeof(double), dCov + (size_t)b * sCv,
                                 (size_t)mpad * sizeof(double), (size_t)m * sizeof(double), m, hipMemcpyDeviceToHost,
                                 c->stream));
    }
    hipLaunchKernelGGL(finish_var_kernel, dim3((m + 255) / 256, nb), dim3(256), 0, c->stream, dq, (size_t)mpad, dHc, d,
                       c->ks.form, m, dq, (size_t)mpad);  // in place
    BGP_HIP(hipGetLastError());
    if (var)
      BGP_HIP(bgp_memcpy2d_async(var + (size_t)off * m, (size_t)m * sizeof(double), dq, (size_t)mpad * sizeof(double),
                               (size_t)m * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    // (the next chunk reuses the scratch slices: stream order keeps its launches behind these copies)
  }
  if (n_acq) {
    double* dT = s.take((size_t)n_acq * B * mpad);
    double* dacc = s.take((size_t)n_acq * mpad);
    double* dmumin = s.take((size_t)B + BGP_ACQ_MAX);
    int* dbad = reinterpret_cast<int*>(s.take((size_t)n_acq * B / 2 + BGP_ACQ_MAX));
    rc = acq_run(c, c->stream, B, m, mpad, doutB, dqB, *ap, dT, dbad, dmumin, dacc);
    if (rc) return rc;
  }
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

extern "C" int bgp_predict_batch(bgp_ctx* c, int B, const double* h_kernel, int m, const double* Xq, double* mean,
                                 double* var, double* cov) {
  BGP_REQUIRE_IDLE(c, "bgp_predict_batch");
  if (!c || !h_kernel || !Xq || !mean || !var || m <= 0 || B <= 0) {
    bgp_set_error("bgp_predict_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  return predict_run(c, B, h_kernel, m, Xq, mean, var, cov, nullptr);
}

// var_i = max(0, kss_i - q_i) with the prior variances kernel_.diag(Xq) supplied by the host (generic kernels)
__global__ void finish_var_gram_kernel(const double* __restrict__ q, const double* __restrict__ kss, size_t sq, int m,
                                       double* __restrict__ var) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (i >= m) return;
  const double v = kss[(size_t)b * sq + i] - q[(size_t)b * sq + i];
  var[(size_t)b * sq + i] = v < 0.0 ? 0.0 : v;
}

// Predict with HOST-evaluated cross covariances for the B resident posteriors (generic kernel expression trees): Ks is B x m x n
// (kernel_(Xq, X_train) per item), kss B x m (kernel_.diag(Xq)), Kss B x m x m (kernel_(Xq); only with cov).  The products --
// mean = K_* alpha, var = kss - rowsum((K_* K^-1) o K_*), cov = K_** - (K_* K^-1) K_*^T -- are the kernels of bgp_predict_batch.
extern "C" int bgp_predict_batch_gram(bgp_ctx* c, int B, int m, const double* Ks, const double* kss, const double* Kss,
                                      double* mean, double* var, double* cov) {
  BGP_REQUIRE_IDLE(c, "bgp_predict_batch_gram");
  if (!c || !Ks || !kss || !mean || !var || m <= 0 || B <= 0 || (cov && !Kss)) {
    bgp_set_error("bgp_predict_batch_gram: bad argument");
    return BGP_ERR_INVALID;
  }
  if (B > c->post_B) {
    bgp_set_error("bgp_predict_batch_gram: %d posteriors requested but %d resident (call bgp_posterior_batch_gram first)", B,
                  c->post_B);
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, mpad = pad128(m);
  const size_t sKs = (size_t)mpad * npad, sCv = (size_t)mpad * mpad;
  const size_t per_item = sKs + 3 * (size_t)mpad + (cov ? sKs + sCv : 0);
  const size_t budget = (size_t)1 << 29;
  int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, budget / per_item));
  if (chunk >= 8 && chunk < B) chunk &= ~7;
  int rc = bgp_ensure_scratch(c, (size_t)chunk * per_item + 64);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dKs = s.take((size_t)chunk * sKs);
  double* dq = s.take((size_t)chunk * mpad);
  double* dout = s.take((size_t)chunk * mpad);
  double* dkss = s.take((size_t)chunk * mpad);
  double *dP = nullptr, *dCov = nullptr;
  if (cov) {
    dP = s.take((size_t)chunk * sKs);
    dCov = s.take((size_t)chunk * sCv);
  }
  for (int off = 0; off < B; off += chunk) {
    const int nb = std::min(chunk, B - off);
    const double* Kinv = c->dKinv + (size_t)off * npad * npad;
    const double* al = c->dalpha_sol + (size_t)off * npad;
    // K_* with its zero padding (rows m .. mpad, columns n .. npad)
    BGP_HIP(hipMemsetAsync(dKs, 0, (size_t)nb * sKs * sizeof(double), c->stream));
    for (int b = 0; b < nb; b++)
      BGP_HIP(bgp_memcpy2d_async(dKs + (size_t)b * sKs, (size_t)npad * sizeof(double), Ks + (size_t)(off + b) * m * n,
                                 (size_t)n * sizeof(double), (size_t)n * sizeof(double), m, hipMemcpyHostToDevice, c->stream));
    BGP_HIP(bgp_memcpy2d_async(dkss, (size_t)mpad * sizeof(double), kss + (size_t)off * m, (size_t)m * sizeof(double),
                               (size_t)m * sizeof(double), nb, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((m + 3) / 4, nb), dim3(256), 0, c->stream, dKs, npad, sKs, al, (size_t)npad,
                       (const int*)nullptr, npad, m, dout, (size_t)mpad);
    BGP_HIP(bgp_memcpy2d_async(mean + (size_t)off * m, (size_t)m * sizeof(double), dout, (size_t)mpad * sizeof(double),
                               (size_t)m * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    rc = launch_rowquad(c, dKs, npad, sKs, Kinv, npad, (size_t)npad * npad, nullptr, mpad, npad, nb, dq);
    if (rc) return rc;
    if (cov) {
      bgp_launch_gemm4(c->stream, 0, dKs, Kinv, npad, mpad, npad, npad, dP, npad, nb, sKs, (size_t)npad * npad, sKs, nullptr);
      BGP_HIP(hipMemsetAsync(dCov, 0, (size_t)nb * sCv * sizeof(double), c->stream));
      for (int b = 0; b < nb; b++)
        BGP_HIP(bgp_memcpy2d_async(dCov + (size_t)b * sCv, (size_t)mpad * sizeof(double), Kss + (size_t)(off + b) * m * m,
                                   (size_t)m * sizeof(double), (size_t)m * sizeof(double), m, hipMemcpyHostToDevice, c->stream));
      bgp_launch_gemm4(c->stream, 2, dP, dKs, npad, mpad, mpad, npad, dCov, mpad, nb, sKs, sKs, sCv, nullptr);
      for (int b = 0; b < nb; b++)
        BGP_HIP(bgp_memcpy2d_async(cov + (size_t)(off + b) * m * m, (size_t)m * sizeof(double), dCov + (size_t)b * sCv,
                                   (size_t)mpad * sizeof(double), (size_t)m * sizeof(double), m, hipMemcpyDeviceToHost,
                                   c->stream));
    }
    hipLaunchKernelGGL(finish_var_gram_kernel, dim3((m + 255) / 256, nb), dim3(256), 0, c->stream, dq, dkss, (size_t)mpad, m, dq);
    BGP_HIP(hipGetLastError());
    BGP_HIP(bgp_memcpy2d_async(var + (size_t)off * m, (size_t)m * sizeof(double), dq, (size_t)mpad * sizeof(double),
                               (size_t)m * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(bgp_stream_sync(c->stream));  // (the next chunk reuses the staged host slices' arena and the scratch)
  }
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// Closed-form acquisition functions on the device (bask/acquisition.py:154-172 ExpectedImprovement, :197-201
// Expectation, :204-216 LCB), averaged over the hyper-posterior draws exactly like evaluate_acquisitions
// (:112-139): per draw b  tmp = acq(mu_b, std_b);  a draw whose values are not all finite contributes nothing;
// out = sum_b tmp_b / n_samples in draw order (one thread per candidate walks the draws: no atomics on doubles).
// mu = y_std * mean + y_mean, std = sqrt(var * y_std^2) as BayesGPR.predict returns them (skopt predict).
// ------------------------------------------------------------------------------------------
static __device__ __forceinline__ double acq_ndtr(double a) {
  // Phi(a) the way scipy.special.ndtr (cephes) evaluates it: erf near 0, erfc in the tails
  const double x = a * 0.70710678118654752440, z = fabs(x);
  if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
  double y = 0.5 * erfc(z);
  if (x > 0) y = 1.0 - y;
  return y;
}

static __device__ __forceinline__ double acq_value(int kind, double param, double mu, double sd, double mumin) {
#pragma clang fp contract(off)
  if (kind == BGP_ACQ_MEAN) return -mu;
  if (kind == BGP_ACQ_STD) return sd;
  if (kind == BGP_ACQ_LCB) return param * sd - mu;
  // expected improvement over y_opt (default: this draw's lowest mean); zero where std is not positive
  if (!(sd > 0.0)) return 0.0;
  const double y_opt = isnan(param) ? mumin : param;
  const double x = (y_opt - mu) / sd;
  const double f = x * acq_ndtr(x) + exp(-(x * x) / 2.0) / 2.5066282746310002;  // sqrt(2 pi)
  return f * sd;
}

// np.min over one draw's means (NaN if any is NaN)
__global__ void __launch_bounds__(256) acq_mumin_kernel(const double* __restrict__ mean, size_t smean, int m,
                                                         double y_mean, double y_std, double* __restrict__ mumin) {
#pragma clang fp contract(off)
  const int b = blockIdx.x, tid = threadIdx.x;
  double best = INFINITY;
  int nan = 0;
  for (int i = tid; i < m; i += 256) {
    const double mu = y_std * mean[(size_t)b * smean + i] + y_mean;
    if (isnan(mu)) nan = 1;
    best = fmin(best, mu);
  }
  __shared__ double sb[256];
  __shared__ int sn[256];
  sb[tid] = best, sn[tid] = nan;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) sb[tid] = fmin(sb[tid], sb[tid + o]), sn[tid] |= sn[tid + o];
    __syncthreads();
  }
  if (tid == 0) mumin[b] = sn[0] ? NAN : sb[0];
}

__global__ void __launch_bounds__(256) acq_values_kernel(const double* __restrict__ mean, const double* __restrict__ var,
                                                          size_t sm, int m, int B, double y_mean, double y_std, int n_acq,
                                                          const int* __restrict__ kinds,
                                                          const double* __restrict__ params,
                                                          const double* __restrict__ mumin, double* __restrict__ T,
                                                          int* __restrict__ bad) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= m) return;
  const double mu = y_std * mean[(size_t)b * sm + i] + y_mean;
  const double sd = sqrt(var[(size_t)b * sm + i] * (y_std * y_std));
  for (int k = 0; k < n_acq; k++) {
    const double v = acq_value(kinds[k], params[k], mu, sd, mumin[b]);
    T[((size_t)k * B + b) * sm + i] = v;
    if (!isfinite(v)) atomicOr(&bad[k * B + b], 1);
  }
}

__global__ void __launch_bounds__(256) acq_sum_kernel(const double* __restrict__ T, const int* __restrict__ bad, size_t sm,
                                                       int m, int B, int n_samples, double* __restrict__ acc) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
  if (i >= m) return;
  double s = 0.0;
  const double ns = (double)n_samples;
  for (int b = 0; b < B; b++)
    if (!bad[k * B + b]) s += T[((size_t)k * B + b) * sm + i] / ns;
  acc[(size_t)k * sm + i] = s;
}

__global__ void acq_square_kernel(double* __restrict__ v, size_t sv, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i < m) v[(size_t)b * sv + i] *= v[(size_t)b * sv + i];
}

static int acq_run(bgp_ctx* c, hipStream_t st, int B, int m, int mpad, const double* dmean, const double* dvar,
                   const AcqPlan& ap, double* dT, int* dbad, double* dmumin, double* dacc) {
  // kinds / params ride in the tail of the draws' scratch (a few words)
  int* dkinds = dbad + (size_t)ap.n_acq * B;
  (void)c;
  double* dparams = dmumin + B;  // (mumin slice was taken with B + slack below)
  BGP_HIP(bgp_memcpy_async(dkinds, ap.kinds, (size_t)ap.n_acq * sizeof(int), hipMemcpyHostToDevice, st));
  BGP_HIP(bgp_memcpy_async(dparams, ap.params, (size_t)ap.n_acq * sizeof(double), hipMemcpyHostToDevice, st));
  BGP_HIP(hipMemsetAsync(dbad, 0, (size_t)ap.n_acq * B * sizeof(int), st));
  hipLaunchKernelGGL(acq_mumin_kernel, dim3(B), dim3(256), 0, st, dmean, (size_t)mpad, m, ap.y_mean, ap.y_std, dmumin);
  hipLaunchKernelGGL(acq_values_kernel, dim3((m + 255) / 256, B), dim3(256), 0, st, dmean, dvar, (size_t)mpad, m, B,
                     ap.y_mean, ap.y_std, ap.n_acq, dkinds, dparams, dmumin, dT, dbad);
  hipLaunchKernelGGL(acq_sum_kernel, dim3((m + 255) / 256, ap.n_acq), dim3(256), 0, st, dT, dbad, (size_t)mpad, m, B,
                     ap.n_samples, dacc);
  BGP_HIP(hipGetLastError());
  BGP_HIP(bgp_memcpy2d_async(ap.out, (size_t)m * sizeof(double), dacc, (size_t)mpad * sizeof(double),
                           (size_t)m * sizeof(double), ap.n_acq, hipMemcpyDeviceToHost, st));
  return BGP_OK;
}

static int acq_check(const char* who, int n_acq, const int* kinds, const double* params, int n_samples, const double* out) {
  if (n_acq <= 0 || n_acq > BGP_ACQ_MAX || !kinds || !params || !out || n_samples <= 0) {
    bgp_set_error("%s: bad argument (1 <= n_acq <= %d)", who, BGP_ACQ_MAX);
    return BGP_ERR_INVALID;
  }
  for (int k = 0; k < n_acq; k++)
    if (kinds[k] < BGP_ACQ_EI || kinds[k] > BGP_ACQ_STD) {
      bgp_set_error("%s: unknown acquisition kind %d", who, kinds[k]);
      return BGP_ERR_INVALID;
    }
  return BGP_OK;
}

extern "C" int bgp_acq_batch(bgp_ctx* c, int B, const double* h_kernel, int m, const double* Xq, double y_mean,
                             double y_std, int n_acq, const int* kinds, const double* params, int n_samples,
                             double* out) {
  BGP_REQUIRE_IDLE(c, "bgp_acq_batch");
  if (!c || !h_kernel || !Xq || m <= 0 || B <= 0) {
    bgp_set_error("bgp_acq_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  int rc = acq_check("bgp_acq_batch", n_acq, kinds, params, n_samples, out);
  if (rc) return rc;
  AcqPlan ap;
  ap.n_acq = n_acq, ap.kinds = kinds, ap.params = params, ap.y_mean = y_mean, ap.y_std = y_std, ap.n_samples = n_samples;
  ap.out = out;
  return predict_run(c, B, h_kernel, m, Xq, nullptr, nullptr, nullptr, &ap);
}

// The same closed forms on caller-supplied (mu, std) rows: B x m each, already in y units (y_mean = 0, y_std = 1).
extern "C" int bgp_acq_values(bgp_ctx* c, int B, int m, const double* mu, const double* std_, int n_acq, const int* kinds,
                              const double* params, int n_samples, double* out) {
  BGP_REQUIRE_IDLE(c, "bgp_acq_values");
  if (!c || !mu || !std_ || m <= 0 || B <= 0) {
    bgp_set_error("bgp_acq_values: bad argument");
    return BGP_ERR_INVALID;
  }
  int rc = acq_check("bgp_acq_values", n_acq, kinds, params, n_samples, out);
  if (rc) return rc;
  BGP_HIP(hipSetDevice(c->device));
  const int mpad = pad128(m);
  rc = bgp_ensure_scratch(c, 2 * (size_t)B * mpad + (size_t)n_acq * ((size_t)B * mpad + B + mpad) + 2 * (size_t)B + 64);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dmu = s.take((size_t)B * mpad);
  double* dvar = s.take((size_t)B * mpad);
  double* dT = s.take((size_t)n_acq * B * mpad);
  double* dacc = s.take((size_t)n_acq * mpad);
  double* dmumin = s.take((size_t)B + BGP_ACQ_MAX);
  int* dbad = reinterpret_cast<int*>(s.take((size_t)n_acq * B / 2 + BGP_ACQ_MAX));
  BGP_HIP(bgp_memcpy2d_async(dmu, (size_t)mpad * sizeof(double), mu, (size_t)m * sizeof(double), (size_t)m * sizeof(double),
                           B, hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy2d_async(dvar, (size_t)mpad * sizeof(double), std_, (size_t)m * sizeof(double),
                           (size_t)m * sizeof(double), B, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(acq_square_kernel, dim3((m + 255) / 256, B), dim3(256), 0, c->stream, dvar, (size_t)mpad, m);
  AcqPlan ap;
  ap.n_acq = n_acq, ap.kinds = kinds, ap.params = params, ap.n_samples = n_samples, ap.out = out;
  rc = acq_run(c, c->stream, B, m, mpad, dmu, dvar, ap, dT, dbad, dmumin, dacc);
  if (rc) return rc;
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// LML gradient:  g_k = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dh_k   (sklearn/_gpr.py:615-647)
// One workgroup per lower-triangular 128x128 tile (off-diagonal tiles count twice); per-dimension
// sums are reduced in the workgroup and stored to gpart[b][tile][k]; grad_reduce_kernel adds the tiles in a
// fixed order (bitwise reproducible gradients: no floating-point atomics).
// ------------------------------------------------------------------------------------------
#define GR_DK 16
__global__ void __launch_bounds__(256) lml_grad_kernel(const double* __restrict__ X,
                                                        const double* __restrict__ H,
                                                        const double* __restrict__ Kinv,
                                                        const double* __restrict__ alpha_sol,
                                                        double* __restrict__ gpart, int n, int d, int npad, int nblk,
                                                        int form, int stat, int B) {
  const int ntiles = nblk * (nblk + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntiles, B, b, t);
  if (b >= B) return;
  double* gp = gpart + ((size_t)b * ntiles + t) * (d + 2);
  int ti, tj;
  bgp_tri_decode(t, ti, tj);
  __shared__ double xi[GR_DK][BGP_TILE_LD];
  __shared__ double xj[GR_DK][BGP_TILE_LD];
  __shared__ double ell[GR_DK];
  __shared__ double red[4][GR_DK + 2];
  const double* h = H + (size_t)b * (d + 2);
  const double* Ki = Kinv + (size_t)b * npad * npad;
  const double* al = alpha_sol + (size_t)b * npad;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, lane = tid & 63, w = tid >> 6;
  const int i0 = ti * 128, j0 = tj * 128;
  const double wtile = (ti == tj) ? 1.0 : 2.0;
  double F[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) F[r][c] = 0.0;
  // pass 1: squared scaled distances
  for (int k0 = 0; k0 < d; k0 += GR_DK) {
    const int kc = min(GR_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += 256) {
      int row = idx / kc, k = idx - row * kc;
      int gi = i0 + row, gj = j0 + row;
      xi[k][row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / ell[k] : 0.0;
      xj[k][row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / ell[k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[8], bb[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) bb[c] = xj[k][tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          double df = a[r] - bb[c];
          F[r][c] += df * df;
        }
    }
  }
  // F_ij = W_ij * f(r_ij);   constant / noise terms on the fly
  const double cst = exp(h[0]), s2 = exp(h[d + 1]);
  const double cf = (form == BGP_FORM_PRODUCT) ? cst : 1.0;
  double g_const = 0.0, g_noise = 0.0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int gi = i0 + ty + 16 * r;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gj = j0 + tx + 16 * c;
      double f = 0.0;
      if (gi < n && gj < n) {
        const double Wij = al[gi] * al[gj] - Ki[(size_t)gi * npad + gj];
        if (gi == gj) {
          g_const += Wij * ((form == BGP_FORM_PRODUCT) ? cst * 1.0 : cst);
          g_noise += Wij * s2;
        } else {
          const double r2 = F[r][c];
          double S, fr;
          if (stat == BGP_RBF) {
            S = exp(-0.5 * r2);
            fr = S;
          } else if (stat == BGP_MATERN12) {
            const double rr = sqrt(r2);
            S = exp(-rr);
            fr = (rr > 0.0) ? S / rr : 0.0;
          } else if (stat == BGP_MATERN32) {
            const double tt = sqrt(r2) * 1.7320508075688772;
            const double e = exp(-tt);
            S = (1.0 + tt) * e;
            fr = 3.0 * e;
          } else {
            const double tt = sqrt(r2) * 2.23606797749979;
            const double e = exp(-tt);
            S = (1.0 + tt + tt * tt / 3.0) * e;
            fr = (5.0 / 3.0) * (tt + 1.0) * e;
          }
          g_const += wtile * Wij * ((form == BGP_FORM_PRODUCT) ? cst * S : cst);
          f = wtile * Wij * cf * fr;
        }
      }
      F[r][c] = f;
    }
  }
  // pass 2: per-dimension sums  sum_ij F_ij (x_ik - x_jk)^2 / l_k^2
  for (int k0 = 0; k0 < d; k0 += GR_DK) {
    const int kc = min(GR_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    if (d > GR_DK) {  // tiles still staged from pass 1 when d fits in one chunk
      for (int idx = tid; idx < kc * 128; idx += 256) {
        int row = idx / kc, k = idx - row * kc;
        int gi = i0 + row, gj = j0 + row;
        xi[k][row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / ell[k] : 0.0;
        xj[k][row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / ell[k] : 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < kc; k++) {
      double a[8], bb[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) bb[c] = xj[k][tx + 16 * c];
      double sk = 0.0;
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          double df = a[r] - bb[c];
          sk += F[r][c] * df * df;
        }
      for (int o = 32; o > 0; o >>= 1) sk += __shfl_xor(sk, o);
      if (lane == 0) red[w][k] = sk;
    }
    __syncthreads();
    if (tid < kc) gp[1 + k0 + tid] = 0.5 * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    g_const += __shfl_xor(g_const, o);
    g_noise += __shfl_xor(g_noise, o);
  }
  __syncthreads();
  if (lane == 0) {
    red[w][0] = g_const;
    red[w][1] = g_noise;
  }
  __syncthreads();
  if (tid == 0) {
    gp[0] = 0.5 * (red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    gp[d + 1] = 0.5 * (red[0][1] + red[1][1] + red[2][1] + red[3][1]);
  }
}

__global__ void grad_reduce_kernel(const double* __restrict__ gpart, double* __restrict__ grad, int ntiles, int p) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= p) return;
  double s = 0.0;
  for (int t = 0; t < ntiles; t++) s += gpart[((size_t)b * ntiles + t) * p + k];
  grad[(size_t)b * p + k] = s;
}

extern "C" int bgp_lml_grad_batch(bgp_ctx* c, int B, const double* h, double* lml, double* grad, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_lml_grad_batch");
  if (!c || !h || !lml || !grad || B <= 0) {
    bgp_set_error("bgp_lml_grad_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  const size_t p = c->d + 2;
  std::vector<int> st(B, 0);
  int rc = bgp_posterior_build(c, B, h, 1, nullptr, nullptr, nullptr, lml, st.data());
  if (rc) return rc;
  c->post_B = 0;  // the resident K^-1 belong to a gradient evaluation, not to a posterior
  const int ntiles = c->nblk * (c->nblk + 1) / 2;
  rc = bgp_ensure_scratch(c, (size_t)B * p * (2 + ntiles));
  if (rc) return rc;
  double* dgrad = c->dscratch;
  double* dH = c->dscratch + (size_t)B * p;
  double* dgpart = c->dscratch + (size_t)B * p * 2;
  BGP_HIP(hipMemsetAsync(dgpart, 0, (size_t)B * p * ntiles * sizeof(double), c->stream));
  BGP_HIP(bgp_memcpy_async(dH, h, (size_t)B * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(lml_grad_kernel, dim3(8 * ((B + 7) / 8) * ntiles), dim3(256), 0, c->stream, c->dXeff, dH, c->dKinv,
                     c->dalpha_sol, dgpart, c->n, c->d, c->npad, c->nblk, c->ks.form, c->ks.stationary, B);
  BGP_HIP(hipGetLastError());
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(B), dim3(((int)p + 63) / 64 * 64), 0, c->stream, dgpart, dgrad, ntiles,
                     (int)p);
  BGP_HIP(hipGetLastError());
  BGP_HIP(bgp_memcpy_async(grad, dgrad, (size_t)B * p * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  for (int b = 0; b < B; b++) {
    if (st[b] != 0)
      for (size_t k = 0; k < p; k++) grad[(size_t)b * p + k] = 0.0;  // sklearn/_gpr.py:589: (-inf, zeros)
    if (status) status[b] = st[b];
  }
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// PVRS (bask/acquisition.py:328-338) through the bordered-inverse identity:
//   covs[i] = sum_t [ k_t^T Kinv k_t + (k(x_t, x_i) - k_i^T Kinv k_t)^2 / (kappa - k_i^T Kinv k_i) ]
// ------------------------------------------------------------------------------------------
__global__ void pvrs_combine_kernel(const double* __restrict__ G, int ldg, const double* __restrict__ Kti, int ldk,
                                    const double* __restrict__ u, const double* __restrict__ st, double kappa, int m,
                                    int T, double* __restrict__ covs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double lam2 = kappa - u[i];
  double s = 0.0;
  for (int t = 0; t < T; t++) {
    const double e = Kti[(size_t)i * ldk + t] - G[(size_t)i * ldg + t];
    s += st[t] + e * e / lam2;
  }
  covs[i] = s;
}

extern "C" int bgp_pvrs(bgp_ctx* c, const double* h_kernel, int m, const double* Xcand, int T,
                        const double* Xthompson, double* covs) {
  BGP_REQUIRE_IDLE(c, "bgp_pvrs");
  if (!c || !h_kernel || !Xcand || !Xthompson || !covs || m <= 0 || T <= 0) {
    bgp_set_error("bgp_pvrs: bad argument");
    return BGP_ERR_INVALID;
  }
  if (c->post_B < 1) {
    bgp_set_error("bgp_pvrs: no resident posterior (call bgp_posterior_batch / bgp_pvrs_prepare first)");
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m), Tpad = pad128(T);
  const size_t p = d + 2;
  size_t need = (size_t)m * d + (size_t)T * d + p + 128 + (size_t)mpad * npad + 2 * (size_t)Tpad * npad +
                2 * (size_t)mpad * Tpad + 2 * (size_t)mpad + 2 * (size_t)Tpad;
  int rc = bgp_ensure_scratch(c, need);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dXc = s.take((size_t)m * d);
  double* dXt = s.take((size_t)T * d);
  double* dhk = s.take(p);
  double* dKc = s.take((size_t)mpad * npad);   // k(cand, train)
  double* dKT = s.take((size_t)Tpad * npad);   // k(thompson, train)
  double* dPT = s.take((size_t)Tpad * npad);   // K_T Kinv
  double* dG = s.take((size_t)mpad * Tpad);    // K_c Kinv K_T^T
  double* dKti = s.take((size_t)mpad * Tpad);  // k(cand, thompson)
  double* du = s.take(mpad);
  double* dcov = s.take(mpad);
  double* dst = s.take(Tpad);
  const double* Kinv = c->dKinv;
  BGP_HIP(bgp_memcpy_async(dXc, Xcand, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dXt, Xthompson, (size_t)T * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dhk, h_kernel, p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->has_warp) {  // candidates and Thompson points are compared in the warped space (bask/acquisition.py:324-327)
    rc = bgp_launch_warp(c, c->stream, dXc, c->dwarp, dXc, m, 1, 0);
    if (rc) return rc;
    rc = bgp_launch_warp(c, c->stream, dXt, c->dwarp, dXt, T, 1, 0);
    if (rc) return rc;
  }
  BGP_HIP(hipMemsetAsync(dKc, 0, ((size_t)mpad * npad + 2 * (size_t)Tpad * npad + 2 * (size_t)mpad * Tpad +
                                   2 * (size_t)mpad + 2 * (size_t)Tpad + 16) * sizeof(double), c->stream));
  rc = bgp_launch_kcross(c, dhk, m, dXc, n, c->dXeff, dKc, npad, 0);
  if (rc) return rc;
  rc = bgp_launch_kcross(c, dhk, T, dXt, n, c->dXeff, dKT, npad, 0);
  if (rc) return rc;
  rc = bgp_launch_kcross(c, dhk, m, dXc, T, dXt, dKti, Tpad, 0);
  if (rc) return rc;
  // P_T = K_T Kinv ; s_t = rowsum(P_T o K_T) ; u_i = rowsum((K_c Kinv) o K_c) ; G = K_c P_T^T
  bgp_launch_gemm4(c->stream, 0, dKT, Kinv, npad, Tpad, npad, npad, dPT, npad, 1, 0, 0, 0, nullptr);
  rc = launch_rowquad(c, dKT, npad, 0, Kinv, npad, 0, nullptr, Tpad, npad, 1, dst);
  if (rc) return rc;
  rc = launch_rowquad(c, dKc, npad, 0, Kinv, npad, 0, nullptr, mpad, npad, 1, du);
  if (rc) return rc;
  bgp_launch_gemm4(c->stream, 0, dKc, dPT, npad, mpad, Tpad, npad, dG, Tpad, 1, 0, 0, 0, nullptr);
  hipLaunchKernelGGL(pvrs_combine_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dG, Tpad, dKti, Tpad, du,
                     dst, kernel_diag_value(c, h_kernel), m, T, dcov);
  BGP_HIP(hipGetLastError());
  BGP_HIP(bgp_memcpy_async(covs, dcov, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

extern "C" int bgp_pvrs_prepare(bgp_ctx* c, const double* h_kernel, int has_alpha_vec, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_pvrs_prepare");
  // K_aug's leading block: kernel_(X_train) + alpha only when alpha is a vector
  // (bask/acquisition.py:332-333)
  if (!c || !h_kernel) {
    bgp_set_error("bgp_pvrs_prepare: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  return bgp_posterior_build(c, 1, h_kernel, has_alpha_vec ? 1 : 0, nullptr, nullptr, nullptr, nullptr, status);
}

// ------------------------------------------------------------------------------------------
// sample_y:  f = mean + L_cov z  with  L_cov = chol(cov + jitter I)  (same batched Cholesky, B = 1)
// ------------------------------------------------------------------------------------------
__global__ void cov_prepare_kernel(double* __restrict__ C, int m, int mpad, double jitter) {
  // jitter on the diagonal, identity padding, zero rhs handled by the caller; blockIdx.y = matrix of a batch
  C += (size_t)blockIdx.y * mpad * mpad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)mpad * mpad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / mpad), j = (int)(idx - (size_t)i * mpad);
    if (i >= m || j >= m)
      C[idx] = (i == j) ? 1.0 : 0.0;
    else if (i == j)
      C[idx] += jitter;
  }
}

__global__ void zero_upper_kernel(double* __restrict__ C, int mpad) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)mpad * mpad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / mpad), j = (int)(idx - (size_t)i * mpad);
    if (j > i) C[idx] = 0.0;
  }
}

__global__ void add_diag_kernel(double* __restrict__ C, int ld, int m, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) C[(size_t)i * ld + i] += v;
}

__global__ void add_mean_rows_kernel(double* __restrict__ out, int ldo, const double* __restrict__ mean, int m,
                                     int rows) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j < m && i < rows) out[(size_t)i * ldo + j] += mean[j];
}

static void free_child(bgp_ctx* w) {
  if (!w) return;
  if (w->dK) (void)hipFree(w->dK);
  if (w->dW) (void)hipFree(w->dW);
  if (w->dyw) (void)hipFree(w->dyw);
  if (w->dacc) (void)hipFree(w->dacc);
  if (w->dlml) (void)hipFree(w->dlml);
  if (w->dstatus) (void)hipFree(w->dstatus);
  // (launch-free factorisation of the covariance: its own flag block)
  if (w->ps_flags) (void)hipFree(w->ps_flags);
  if (w->ps_trace) (void)hipFree(w->ps_trace);
  if (w->ps_herr) (void)hipHostFree(w->ps_herr);
  delete w;
}

static int ensure_child(bgp_ctx* c, int mpad, int nb, bgp_ctx** out);

// out[r] = mean + L z[r] for ALL draws r of one posterior: one wave per row of the lower factor, which is read once
// (the draws' normal vectors stay in L2); rows of z / out have stride ldz.  Fixed order: bitwise reproducible.
#define TRI_MAXD 16
__global__ void __launch_bounds__(256) tri_matmul_draws_kernel(const double* __restrict__ L, int mpad,
                                                                const double* __restrict__ z, int ldz, int n_draws,
                                                                const double* __restrict__ mean, int m,
                                                                double* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* Lr = L + (size_t)row * mpad;
  for (int r0 = 0; r0 < n_draws; r0 += TRI_MAXD) {
    const int nd = min(TRI_MAXD, n_draws - r0);
    double acc[TRI_MAXD];
#pragma unroll
    for (int r = 0; r < TRI_MAXD; r++) acc[r] = 0.0;
    for (int j = lane; j <= row; j += 64) {
      const double l = Lr[j];
#pragma unroll
      for (int r = 0; r < TRI_MAXD; r++)
        if (r < nd) acc[r] += l * z[(size_t)(r0 + r) * ldz + j];
    }
#pragma unroll
    for (int r = 0; r < TRI_MAXD; r++) {
      double s = acc[r];
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (r < nd && lane == 0) out[(size_t)(r0 + r) * ldz + row] = s + mean[row];
    }
  }
}

extern "C" int bgp_sample_y(bgp_ctx* c, int b, const double* h_kernel, int m, const double* Xq, int n_draws,
                            const double* z, double jitter, double* out) {
  BGP_REQUIRE_IDLE(c, "bgp_sample_y");
  if (!c || !h_kernel || !Xq || !z || !out || m <= 0 || n_draws <= 0 || b < 0) {
    bgp_set_error("bgp_sample_y: bad argument");
    return BGP_ERR_INVALID;
  }
  if (b >= c->post_B) {
    bgp_set_error("bgp_sample_y: posterior %d not resident (%d resident)", b, c->post_B);
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m), rpad = pad128(n_draws);
  const size_t p = d + 2;
  // child workspace for the m x m Cholesky (shares the stream); cached on the context and grown on demand
  // (an m = 10 000 candidate grid needs 0.8 GB: re-allocating it per call costs more than the factorisation)
  bgp_ctx* w = nullptr;
  int rc = ensure_child(c, mpad, 1, &w);
  if (rc) return rc;
  w->n = m;
  w->npad = mpad;
  w->nblk = mpad / 128;
  do {
    size_t need = (size_t)m * d + p + 2 * (size_t)mpad * npad + 2 * (size_t)mpad + 2 * (size_t)rpad * mpad + 64;
    rc = bgp_ensure_scratch(c, need);
    if (rc) break;
    Scratch s{c->dscratch, 0};
    double* dXq = s.take((size_t)m * d);
    double* dhk = s.take(p);
    double* dKs = s.take((size_t)mpad * npad);
    double* dP = s.take((size_t)mpad * npad);
    double* dmean = s.take(mpad);
    double* dZ = s.take((size_t)rpad * mpad);
    double* dO = s.take((size_t)rpad * mpad);
    const double* Kinv = c->dKinv + (size_t)b * npad * npad;
    const double* al = c->dalpha_sol + (size_t)b * npad;
    hipError_t e = hipSuccess;
#define SY(call)                       \
  if ((e = (call)) != hipSuccess) {    \
    bgp_set_error("bgp_sample_y: %s failed: %s", #call, hipGetErrorString(e)); \
    rc = BGP_ERR_HIP;                  \
    break;                             \
  }
    SY(bgp_memcpy_async(dXq, Xq, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SY(bgp_memcpy_async(dhk, h_kernel, p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (c->has_warp && (rc = bgp_launch_warp(c, c->stream, dXq, c->dwarp, dXq, m, 1, 0))) break;
    SY(hipMemsetAsync(dKs, 0, (size_t)mpad * npad * sizeof(double), c->stream));
    SY(hipMemsetAsync(dZ, 0, (size_t)rpad * mpad * sizeof(double), c->stream));
    SY(bgp_memcpy2d_async(dZ, (size_t)mpad * sizeof(double), z, (size_t)m * sizeof(double), (size_t)m * sizeof(double),
                        n_draws, hipMemcpyHostToDevice, c->stream));
    if ((rc = bgp_launch_kcross(c, dhk, m, dXq, n, c->dXeff, dKs, npad, 0))) break;
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((m + 3) / 4, 1), dim3(256), 0, c->stream, dKs, npad, (size_t)0, al,
                       (size_t)0, (const int*)nullptr, n, m, dmean, (size_t)0);
    // P = K_* K^-1, then cov = K_** - P K_*^T in the child's matrix, lower tiles only (all the factorisation reads):
    // both products on the LDS-DMA ring (gemm4_kernel)
    bgp_launch_gemm4(c->stream, 0, dKs, Kinv, npad, mpad, npad, npad, dP, npad, 1, 0, 0, 0, nullptr);
    int st = 0;
    for (int attempt = 0; attempt < 2 && !rc; attempt++) {
      if ((rc = bgp_launch_kcross(c, dhk, m, dXq, m, dXq, w->dK, mpad, 0))) break;
      hipLaunchKernelGGL(add_diag_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, w->dK, mpad, m,
                         std::exp(h_kernel[d + 1]));
      bgp_launch_gemm4(c->stream, 1, dP, dKs, npad, mpad, mpad, npad, w->dK, mpad, 1, 0, 0, 0, nullptr);
      hipLaunchKernelGGL(cov_prepare_kernel, dim3(1024), dim3(256), 0, c->stream, w->dK, m, mpad, jitter);
      SY(hipMemsetAsync(w->dyw, 0, (size_t)mpad * sizeof(double), c->stream));
      SY(hipMemsetAsync(w->dstatus, 0, sizeof(int), c->stream));
      // the covariance's factorisation -- ONE matrix of 79 block columns at 10 000 candidates, the longest launch chain of
      // a tell -- on the launch-free path (12.6 -> 10.9 ms per call, same draws); should a wait time out, the covariance
      // is rebuilt and factorised by launches, and the context stays on them
      const bool ps = bgp_persist_fits(w, 1) && (c->persist == 1 || (c->persist == -1 && bgp_persist_auto_rule(w->nblk, 1))) &&
                      bgp_ps_allowed(c);
      if (ps) {
        c->ps_calls++;
        w->persist = 1;
        if ((rc = bgp_launch_cholesky_persist(w, 1, 0))) break;
      } else if ((rc = bgp_launch_cholesky(w, 1, 0))) {
        break;
      }
      SY(bgp_memcpy_async(&st, w->dstatus, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      SY(bgp_stream_sync(c->stream));
      if (ps && w->ps_herr && *w->ps_herr != 0) {
        *w->ps_herr = 0;
        bgp_ps_note_timeout(c, "the covariance is rebuilt and factorised");
        continue;
      }
      break;
    }
    if (rc) break;
    if (st != 0) {
      bgp_set_error("bgp_sample_y: predictive covariance not positive definite at pivot %d (jitter %.3g)", st, jitter);
      rc = BGP_ERR_NOTPD;
      break;
    }
    // out = mean + L z for every draw (the factor's strict upper triangle is never read)
    hipLaunchKernelGGL(tri_matmul_draws_kernel, dim3((m + 3) / 4), dim3(256), 0, c->stream, w->dK, mpad, dZ, mpad, n_draws,
                       dmean, m, dO);
    SY(hipGetLastError());
    SY(bgp_memcpy2d_async(out, (size_t)m * sizeof(double), dO, (size_t)mpad * sizeof(double),
                        (size_t)m * sizeof(double), n_draws, hipMemcpyDeviceToHost, c->stream));
    SY(bgp_stream_sync(c->stream));
#undef SY
  } while (0);
  (void)hipStreamSynchronize(c->stream);
  if (rc) bgp_xfer_drop_pending();  // (a failed call unpacks nothing into the caller's buffers later)
  bgp_xfer_release(c->stream);
  return rc;
}

// out_b = mean_b + L_b z_b for the lower factors left by the batched Cholesky (one wave per row, fixed order)
__global__ void __launch_bounds__(256) tri_matvec_kernel(const double* __restrict__ Lb, int mpad, const double* __restrict__ z,
                                                          const double* __restrict__ mean, int m,
                                                          double* __restrict__ out) {
  const int b = blockIdx.y, row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* L = Lb + (size_t)b * mpad * mpad + (size_t)row * mpad;
  const double* zb = z + (size_t)b * mpad;
  double s = 0.0;
  for (int j = lane; j <= row; j += 64) s += L[j] * zb[j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[(size_t)b * mpad + row] = s + mean[(size_t)b * mpad + row];
}

// workspace of the m x m covariance factorisations: a child context sharing the parent's stream, cached on the
// parent and grown on demand (rows mpad, nb matrices side by side)
static int ensure_child(bgp_ctx* c, int mpad, int nb, bgp_ctx** out) {
  bgp_ctx* w = c->child;
  if (w && (w->cap_n < (size_t)mpad || w->max_batch < nb)) {
    free_child(w);
    w = c->child = nullptr;
  }
  if (!w) {
    w = new bgp_ctx();
    w->device = c->device;
    w->stream = c->stream;
    w->ncu = c->ncu;
    w->d = c->d;
    w->max_batch = nb;
    const size_t B8 = 8 * ((size_t)(nb + 7) / 8);
    if (hipMalloc(&w->dK, (size_t)nb * mpad * mpad * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dW, (size_t)nb * (mpad / 128) * 128 * 128 * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dyw, (size_t)nb * mpad * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dacc, B8 * 4 * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dlml, B8 * sizeof(double)) != hipSuccess || hipMalloc(&w->dstatus, B8 * sizeof(int)) != hipSuccess) {
      bgp_set_error("hipMalloc of the %d x (%d x %d) covariance workspace failed", nb, mpad, mpad);
      (void)hipGetLastError();
      free_child(w);
      return BGP_ERR_HIP;
    }
    w->cap_n = mpad;
    c->child = w;
  }
  w->panels = c->panels;
  w->panels_auto = c->panels_auto;
  *out = w;
  return BGP_OK;
}

// One function draw per resident posterior (the hyper-posterior branch of BayesGPR.sample_y, bask/bayesgpr.py:679-718,
// and the Thompson-sampling acquisition): item i uses posterior pidx[i] with the kernel parameters h_kernel[i] (noise
// already switched off by the caller where the reference does), its own standard-normal vector z[i] and returns
// out[i] = mean_i + chol(cov_i + jitter I) z[i].  All items of a chunk share every launch (grid.y = item), including
// ONE batched Cholesky of their covariance matrices.  status[i] != 0: covariance i not positive definite at that
// jitter (out[i] undefined) -- the caller retries those items with a larger jitter.
extern "C" int bgp_sample_y_batch(bgp_ctx* c, int B, const int* pidx, const double* h_kernel, int m, const double* Xq,
                                  const double* z, double jitter, double* out, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_sample_y_batch");
  if (!c || !pidx || !h_kernel || !Xq || !z || !out || !status || m <= 0 || B <= 0) {
    bgp_set_error("bgp_sample_y_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  for (int i = 0; i < B; i++)
    if (pidx[i] < 0 || pidx[i] >= c->post_B) {
      bgp_set_error("bgp_sample_y_batch: posterior %d not resident (%d resident)", pidx[i], c->post_B);
      return BGP_ERR_STATE;
    }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m);
  const size_t p = d + 2;
  const size_t sKs = (size_t)mpad * npad, sCv = (size_t)mpad * mpad;
  const size_t per_item = 2 * sKs + sCv + 3 * (size_t)mpad;  // K_*, P, cov (child), mean / z / out
  const size_t budget = (size_t)1 << 30;                     // doubles per chunk (8 GiB)
  const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)B, budget / per_item));
  bgp_ctx* w = nullptr;
  int rc = ensure_child(c, mpad, chunk, &w);
  if (rc) return rc;
  w->n = m;
  w->npad = mpad;
  w->nblk = mpad / 128;
  size_t need = (size_t)m * d + 2 + (size_t)B * p + 2 + (size_t)chunk * (2 * sKs + 3 * (size_t)mpad) + (size_t)B + 64;
  rc = bgp_ensure_scratch(c, need);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dXq = s.take((size_t)m * d);
  double* dH = s.take((size_t)B * p);
  double* dKs = s.take((size_t)chunk * sKs);
  double* dP = s.take((size_t)chunk * sKs);
  double* dmean = s.take((size_t)chunk * mpad);
  double* dZ = s.take((size_t)chunk * mpad);
  double* dO = s.take((size_t)chunk * mpad);
  int* dpidx = reinterpret_cast<int*>(s.take(((size_t)B + 1) / 2 + 1));
  BGP_HIP(bgp_memcpy_async(dXq, Xq, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dH, h_kernel, (size_t)B * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dpidx, pidx, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
  if (c->has_warp) {
    rc = bgp_launch_warp(c, c->stream, dXq, c->dwarp, dXq, m, 1, 0);
    if (rc) return rc;
  }
  for (int off = 0; off < B; off += chunk) {
    const int nb = std::min(chunk, B - off);
    const double* dHc = dH + (size_t)off * p;
    const int* dpc = dpidx + off;
    BGP_HIP(hipMemsetAsync(dZ, 0, (size_t)nb * mpad * sizeof(double), c->stream));
    BGP_HIP(bgp_memcpy2d_async(dZ, (size_t)mpad * sizeof(double), z + (size_t)off * m, (size_t)m * sizeof(double),
                             (size_t)m * sizeof(double), nb, hipMemcpyHostToDevice, c->stream));
    rc = bgp_launch_kcross_batch(c, nb, dHc, m, dXq, n, c->dXeff, dKs, npad, sKs);
    if (rc) return rc;
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((m + 3) / 4, nb), dim3(256), 0, c->stream, dKs, npad, sKs, c->dalpha_sol,
                       (size_t)npad, dpc, n, m, dmean, (size_t)mpad);
    // P = K_* K^-1
    bgp_launch_gemm4(c->stream, 0, dKs, c->dKinv, npad, mpad, npad, npad, dP, npad, nb, sKs, (size_t)npad * npad, sKs, dpc);
    // cov = K_** - P K_*^T in place in the child's matrices, + jitter, identity padding
    rc = bgp_launch_kcross_batch(c, nb, dHc, m, dXq, m, dXq, w->dK, mpad, sCv);
    if (rc) return rc;
    hipLaunchKernelGGL(add_diag_batch_kernel, dim3((m + 255) / 256, nb), dim3(256), 0, c->stream, w->dK, mpad, sCv, m, dHc,
                       d);
    bgp_launch_gemm4(c->stream, 1, dP, dKs, npad, mpad, mpad, npad, w->dK, mpad, nb, sKs, sKs, sCv, nullptr);
    hipLaunchKernelGGL(cov_prepare_kernel, dim3(256, nb), dim3(256), 0, c->stream, w->dK, m, mpad, jitter);
    BGP_HIP(hipMemsetAsync(w->dyw, 0, (size_t)nb * mpad * sizeof(double), c->stream));
    BGP_HIP(hipMemsetAsync(w->dstatus, 0, (size_t)nb * sizeof(int), c->stream));
    rc = bgp_launch_cholesky(w, nb, 0);
    if (rc) return rc;
    hipLaunchKernelGGL(tri_matvec_kernel, dim3((m + 3) / 4, nb), dim3(256), 0, c->stream, w->dK, mpad, dZ, dmean, m, dO);
    BGP_HIP(hipGetLastError());
    BGP_HIP(bgp_memcpy_async(status + off, w->dstatus, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(bgp_memcpy2d_async(out + (size_t)off * m, (size_t)m * sizeof(double), dO, (size_t)mpad * sizeof(double),
                             (size_t)m * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
  }
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

void bgp_free_child(bgp_ctx* c) {
  if (c && c->child) {
    free_child(c->child);
    c->child = nullptr;
  }
}
