// Device-resident ensemble sampler: the red-blue stretch move of emcee 3.1.6 (the sampler bask/bayesgpr.py:510-530 runs) with
// the walkers, their log-probabilities, the proposals, the log-priors, the accept test and the chain itself in HBM.  The host
// draws every random number of the run up front -- in emcee's stream order, none of them depends on a log-probability -- and
// enqueues, per half-step, ONE small kernel (accept the previous half-step, propose the next: mcmc_step_kernel) in front of the
// LML batch of the proposals (bgp_lml_enqueue_dev: Gram build + factorisation, the launch-free kernel where it applies).  No
// transfer and no synchronisation between the first and the last half-step: the host-driven loop (sampler.py) pays an upload, a
// download, a stream synchronisation and its own bookkeeping per half-step (40-85 us of a 0.53 ms half-step at n = 1024,
// most of a 0.09 ms half-step at n = 128).
//
// Same arithmetic as the host-driven loop, operation by operation (q = c - (c - s) z; the priors summed in theta order; lp =
// prior + LML, non-finite -> -inf; accept iff (p - 1) log z + lp_new - lp_old > log u), except that exp / pow of the two
// default prior families come from the device's libm instead of numpy's: log-probabilities agree to ~1e-15 relative, the
// walkers' positions are identical unless an accept test is decided by that last bit.  gfx950 only.
#include "bgp_common.h"

struct McmcArgs {
  int W, p, Ns, hp;       // walkers, entries of a walker, proposals per half-step, d + 2
  int nhalf;              // half-steps of the run (2 x steps)
  double* coords;         // W x p
  double* logp;           // W
  long long* nacc;        // W
  double* q;              // Ns x p: proposals of the half-step in flight
  double* prior;          // Ns
  double* pterm;          // Ns x p: the log-prior terms of the proposals, summed in theta order by one thread per proposal
  int* status;            // Ns: statuses of the LML batch (reset here, in front of it)
  unsigned* ps_flags;     // flag block of the launch-free factorisation (reset here) or nullptr
  int ps_words;
  double* dh;             // Ns x hp: canonical hyper-parameters of the proposals (the LML batch reads them)
  const double* lml;      // Ns: the LML batch's results
  const int* h_src;       // hp: index into a walker, or -1: h_fixed
  const double* h_fixed;  // hp
  const int* prior_kind;  // p: 1 half-Normal on sqrt(exp(t)), 2 round-flat on exp(t) (both with the log-space Jacobian)
  const double* prior_par;  // p x 5
  const int* movers;      // nhalf x Ns   (the plan of the whole run)
  const int* partners;    // nhalf x Ns
  const double* zz;       // nhalf x Ns
  const double* factors;  // nhalf x Ns
  const double* logu;     // nhalf x Ns
  double* chain;          // steps x W x p
  double* lps;            // steps x W
  unsigned* info;         // [0] a proposal had a non-finite coordinate, [1] a launch-free factorisation abandoned its waits
  const unsigned* ps_err; // error word of the launch-free kernel of the half-step just finished, or nullptr
};

static __device__ __forceinline__ double mcmc_prior(int kind, const double* par, double t) {
#pragma clang fp contract(off)
  if (kind == 1) {
    // priors.halfnorm_logpdf_logspace:  c - 0.5 * exp(t) / (scale * scale) + 0.5 * t
    return par[0] - 0.5 * exp(t) / par[1] + 0.5 * t;
  }
  // utils._collect_priors.ls_prior:  (-2.0 * ((x / lo) ** p_lo + (x / hi) ** p_hi) - log_norm) + t,  x = exp(t)
  const double x = exp(t);
  const double a = pow(x / par[0], par[2]), b = pow(x / par[1], par[3]);
  return (-2.0 * (a + b) - par[4]) + t;
}

// ONE workgroup.  h = index of the half-step to PROPOSE (0 .. nhalf); the accept phase closes half-step h - 1.
__global__ void __launch_bounds__(1024) mcmc_step_kernel(McmcArgs a, int h) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x, nt = blockDim.x, p = a.p, Ns = a.Ns;
  if (h > 0) {
    const int g = h - 1;
    const int* mv = a.movers + (size_t)g * Ns;
    if (tid == 0 && a.ps_err && *a.ps_err != 0) a.info[1] = 1u;
    for (int i = tid; i < Ns; i += nt) {
      double lp = a.prior[i] + a.lml[i];
      if (!(lp > -INFINITY && lp < INFINITY)) lp = -INFINITY;  // (NaN included, as _log_prob_finish)
      const int m = mv[i];
      const bool acc = a.factors[(size_t)g * Ns + i] + lp - a.logp[m] > a.logu[(size_t)g * Ns + i];
      a.status[i] = acc ? -1 : 0;  // (the batch is over: its status words carry the accept flags to the copy below)
      if (acc) {
        a.logp[m] = lp;
        a.nacc[m] += 1;
      }
    }
    __syncthreads();
    for (int e = tid; e < Ns * p; e += nt) {
      const int i = e / p;
      if (a.status[i]) a.coords[(size_t)mv[i] * p + (e - i * p)] = a.q[e];
    }
    __syncthreads();
    if (g & 1) {  // second half of step g / 2: the ensemble goes into the chain
      const int step = g >> 1;
      for (int e = tid; e < a.W * p; e += nt) a.chain[(size_t)step * a.W * p + e] = a.coords[e];
      for (int e = tid; e < a.W; e += nt) a.lps[(size_t)step * a.W + e] = a.logp[e];
    }
  }
  if (h >= a.nhalf) return;
  // the resets the LML batch would otherwise enqueue as dispatches of their own (the previous batch, and with it every reader
  // of these words, is over: this kernel runs behind it on the stream)
  for (int e = tid; e < Ns; e += nt) a.status[e] = 0;
  if (a.ps_flags)
    for (int e = tid; e < a.ps_words; e += nt) a.ps_flags[e] = 0u;
  const int* mv = a.movers + (size_t)h * Ns;
  const int* pr = a.partners + (size_t)h * Ns;
  for (int e = tid; e < Ns * p; e += nt) {
    const int i = e / p, k = e - i * p;
    const double z = a.zz[(size_t)h * Ns + i];
    const double s = a.coords[(size_t)mv[i] * p + k], c = a.coords[(size_t)pr[i] * p + k];
    const double v = c - (c - s) * z;
    a.q[e] = v;
    if (!(v > -INFINITY && v < INFINITY)) a.info[0] = 1u;
    a.pterm[e] = mcmc_prior(a.prior_kind[k], a.prior_par + 5 * k, v);
  }
  __syncthreads();
  for (int i = tid; i < Ns; i += nt) {
    double lp = 0.0;
    for (int k = 0; k < p; k++) lp += a.pterm[(size_t)i * p + k];
    a.prior[i] = lp;
  }
  for (int e = tid; e < Ns * a.hp; e += nt) {
    const int i = e / a.hp, j = e - i * a.hp;
    a.dh[e] = a.h_src[j] >= 0 ? a.q[(size_t)i * p + a.h_src[j]] : a.h_fixed[j];
  }
}

namespace {
struct DevBlock {  // one allocation for the whole run, released on every exit path
  char* base = nullptr;
  size_t used = 0, cap = 0;
  ~DevBlock() {
    if (base) (void)hipFree(base);
  }
  template <class T>
  T* take(size_t count) {
    const size_t off = (used + 255) & ~(size_t)255;
    used = off + count * sizeof(T);
    return base ? reinterpret_cast<T*>(base + off) : nullptr;
  }
};
}  // namespace

// See include/bgp.h.  nsteps steps of W walkers with p entries each; Ns = W / 2 proposals per half-step (W even).
extern "C" int bgp_mcmc_run(bgp_ctx* c, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                            const double* prior_par, const double* coords0, const double* logp0, const int* movers,
                            const int* partners, const double* zz, const double* factors, const double* logu, double* chain,
                            double* logp, double* coords_out, double* logp_out, long long* naccepted, int* info) {
  if (!c || !h_src || !h_fixed || !prior_kind || !prior_par || !coords0 || !logp0 || !movers || !partners || !zz || !factors ||
      !logu || !chain || !logp || !coords_out || !logp_out || !naccepted || !info || W < 2 || (W & 1) || p < 1 || nsteps < 1) {
    bgp_set_error("bgp_mcmc_run: bad argument (W must be even and >= 2)");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_mcmc_run");
  const int Ns = W / 2, hp = c->d + 2, nhalf = 2 * nsteps;
  if (Ns > c->max_batch) {
    bgp_set_error("bgp_mcmc_run: %d proposals per half-step exceed max_batch = %d", Ns, c->max_batch);
    return BGP_ERR_INVALID;
  }
  if (c->timing) {
    bgp_set_error("bgp_mcmc_run: per-launch timing is on (bgp_set_timing): use the host-driven sampler");
    return BGP_ERR_STATE;
  }
  for (int k = 0; k < p; k++)
    if (prior_kind[k] != 1 && prior_kind[k] != 2) {
      bgp_set_error("bgp_mcmc_run: prior kind %d of entry %d is not one of the device's (1 half-Normal, 2 round-flat)", prior_kind[k], k);
      return BGP_ERR_INVALID;
    }
  for (int j = 0; j < hp; j++)
    if (h_src[j] >= p) {
      bgp_set_error("bgp_mcmc_run: canonical entry %d reads walker entry %d of %d", j, h_src[j], p);
      return BGP_ERR_INVALID;
    }
  BGP_HIP(hipSetDevice(c->device));
  const size_t plan = (size_t)nhalf * Ns;
  DevBlock blk;
  McmcArgs a;
  for (int pass = 0; pass < 2; pass++) {  // pass 0 sizes the block, pass 1 hands out the pointers
    blk.used = 0;
    a.coords = blk.take<double>((size_t)W * p);
    a.logp = blk.take<double>(W);
    a.nacc = blk.take<long long>(W);
    a.q = blk.take<double>((size_t)Ns * p);
    a.prior = blk.take<double>(Ns);
    a.pterm = blk.take<double>((size_t)Ns * p);
    a.h_src = blk.take<int>(hp);
    a.h_fixed = blk.take<double>(hp);
    a.prior_kind = blk.take<int>(p);
    a.prior_par = blk.take<double>((size_t)5 * p);
    a.movers = blk.take<int>(plan);
    a.partners = blk.take<int>(plan);
    a.zz = blk.take<double>(plan);
    a.factors = blk.take<double>(plan);
    a.logu = blk.take<double>(plan);
    a.chain = blk.take<double>((size_t)nsteps * W * p);
    a.lps = blk.take<double>((size_t)nsteps * W);
    a.info = blk.take<unsigned>(2);
    if (pass == 0) {
      blk.cap = blk.used;
      BGP_HIP(hipMalloc((void**)&blk.base, blk.cap));
    }
  }
  a.W = W;
  a.p = p;
  a.Ns = Ns;
  a.hp = hp;
  a.nhalf = nhalf;
  a.dh = c->dh;
  a.lml = c->dlml;
  a.status = c->dstatus;
  a.ps_flags = nullptr;
  a.ps_words = 0;
  if (c->nblk > 1 && bgp_persist_fits(c, Ns)) {  // a launch-free call may follow: its flag block is reset by the step kernel
    const int rcf = bgp_ps_ensure_flags(c, Ns);
    if (rcf) return rcf;
    a.ps_flags = c->ps_flags;
    a.ps_words = (int)ps_flag_words(Ns, c->nblk);
  }
  const int threads = (Ns * p > 512) ? 1024 : 256;
  hipStream_t st = c->stream;
  auto up = [&](const void* dst, const void* src, size_t bytes) {
    return bgp_memcpy_async(const_cast<void*>(dst), src, bytes, hipMemcpyHostToDevice, st);
  };
  for (int attempt = 0; attempt < 2; attempt++) {
    BGP_HIP(up(a.coords, coords0, (size_t)W * p * sizeof(double)));
    BGP_HIP(up(a.logp, logp0, (size_t)W * sizeof(double)));
    BGP_HIP(hipMemsetAsync(a.nacc, 0, (size_t)W * sizeof(long long), st));
    BGP_HIP(hipMemsetAsync(a.info, 0, 2 * sizeof(unsigned), st));
    if (attempt == 0) {
      BGP_HIP(up(a.h_src, h_src, hp * sizeof(int)));
      BGP_HIP(up(a.h_fixed, h_fixed, hp * sizeof(double)));
      BGP_HIP(up(a.prior_kind, prior_kind, p * sizeof(int)));
      BGP_HIP(up(a.prior_par, prior_par, (size_t)5 * p * sizeof(double)));
      BGP_HIP(up(a.movers, movers, plan * sizeof(int)));
      BGP_HIP(up(a.partners, partners, plan * sizeof(int)));
      BGP_HIP(up(a.zz, zz, plan * sizeof(double)));
      BGP_HIP(up(a.factors, factors, plan * sizeof(double)));
      BGP_HIP(up(a.logu, logu, plan * sizeof(double)));
    }
    int rc = BGP_OK;
    c->ps_resident = 1;
    a.ps_err = nullptr;
    for (int h = 0; h <= nhalf && rc == BGP_OK; h++) {
      hipLaunchKernelGGL(mcmc_step_kernel, dim3(1), dim3(threads), 0, st, a, h);
      if (h == nhalf) break;
      c->ps_inflight = 0;
      rc = bgp_lml_enqueue_dev(c, Ns, 0);
      // (the launch-free kernel's error word is reset in front of every call: the next step kernel folds it into info[1])
      a.ps_err = (rc == BGP_OK && c->ps_inflight && c->ps_flags) ? c->ps_flags + PS_ERROR : nullptr;
    }
    c->ps_resident = 0;
    c->ps_inflight = 0;
    unsigned hinfo[2] = {0u, 0u};
    if (rc == BGP_OK) {
      const hipError_t e = hipGetLastError();
      if (e != hipSuccess) {
        bgp_set_error("bgp_mcmc_run: %s", hipGetErrorString(e));
        rc = BGP_ERR_HIP;
      }
    }
    if (rc == BGP_OK && bgp_memcpy_async(hinfo, a.info, sizeof(hinfo), hipMemcpyDeviceToHost, st) != hipSuccess) rc = BGP_ERR_HIP;
    if (rc != BGP_OK || bgp_stream_sync(st) != hipSuccess) {  // drain whatever was enqueued; the context stays usable
      (void)hipStreamSynchronize(st);
      for (int g = 0; g < BGP_MAX_STREAMS; g++)
        if (c->gstream[g]) (void)hipStreamSynchronize(c->gstream[g]);
      bgp_xfer_drop_pending();
      (void)hipGetLastError();
      if (rc == BGP_OK) {
        bgp_set_error("bgp_mcmc_run: the run failed on the device");
        rc = BGP_ERR_HIP;
      }
      return rc;
    }
    if (hinfo[1] != 0 && attempt == 0) {
      // a launch-free factorisation gave its waits up somewhere in the run: everything behind it is void.  The whole run is
      // redone on the launch schedule (bit-identical results), loudly, and the context's time-out policy takes note.
      bgp_ps_note_timeout(c, "the sampler's run is redone");
      continue;
    }
    info[0] = (int)hinfo[0];
    info[1] = attempt;
    break;
  }
  BGP_HIP(bgp_memcpy_async(chain, a.chain, (size_t)nsteps * W * p * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(logp, a.lps, (size_t)nsteps * W * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(coords_out, a.coords, (size_t)W * p * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(logp_out, a.logp, (size_t)W * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(naccepted, a.nacc, (size_t)W * sizeof(long long), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_stream_sync(st));
  return BGP_OK;
}
