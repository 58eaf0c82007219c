// Device-resident ensemble sampler: the red-blue stretch move of emcee 3.1.6 (the sampler bask/bayesgpr.py:510-530 runs) with
// the walkers, their log-probabilities, the proposals, the log-priors, the accept test and the chain itself in HBM.  The host
// draws the run's random numbers AHEAD of the device -- in emcee's stream order, none of them depends on a log-probability -- and
// hands them over as the plan of the run in segments (bgp_mcmc_begin / bgp_mcmc_steps / bgp_mcmc_end); per half-step the device
// runs ONE small kernel (accept the previous half-step, propose the next: mcmc_step_kernel) in front of the LML batch of the
// proposals (bgp_lml_enqueue_dev: Gram build + factorisation, the launch-free kernel where it applies) -- or, n <= 128, one fused
// kernel that does all of it per walker (mcmc_small_kernel, bgp_chol.hip).  No transfer and no synchronisation between the first
// and the last half-step: the host-driven loop (sampler.py) pays an upload, a download, a stream synchronisation and its own
// bookkeeping per half-step (40-85 us of a 0.53 ms half-step at n = 1024, most of a 0.09 ms half-step at n = 128).
//
// Same arithmetic as the host-driven loop, operation by operation (q = c - (c - s) z; the priors summed in theta order; lp =
// prior + LML, non-finite -> -inf; accept iff (p - 1) log z + lp_new - lp_old > log u), except inside the two default prior
// families, where exp comes from the device's libm instead of numpy's and the round-flat powers are taken in log space
// (mcmc_prior, bgp_mcmc.h): log-probabilities agree to ~1e-14 relative (measured: 8.3e-15 over 319 176 half-steps), the walkers'
// positions are identical unless an accept test is decided by those last bits (never, in those runs).  gfx950 only.
#include <memory>
#include <vector>
#include "bgp_common.h"
#include "bgp_mcmc.h"

// ONE workgroup.  h = index of the half-step to PROPOSE (0 .. nhalf); the accept phase closes half-step h - 1.  Every microsecond of
// this kernel stands between two LML batches, so the ensemble, the accept flags, the new proposals and their prior terms live in
// LDS (W p + W + 3 Ns + 2 Ns p doubles <= MCMC_LDS_DOUBLES: bgp_mcmc_begin checks) and the kernel is three rounds of memory latency
// -- everything in, the old proposals of the accepted walkers, everything out -- instead of one per phase.
#define MCMC_LDS_DOUBLES 16384  // 128 KB
__global__ void __launch_bounds__(1024) mcmc_step_kernel(McmcArgs a, int h) {
#pragma clang fp contract(off)
  __shared__ double lds[MCMC_LDS_DOUBLES];
  const int tid = threadIdx.x, nt = blockDim.x, p = a.p, Ns = a.Ns, W = a.W;
  double* const co = lds;                  // the ensemble
  double* const lg = co + (size_t)W * p;   // its log-probabilities
  double* const af = lg + W;               // accept flags of half-step h - 1
  double* const qn = af + Ns;              // proposals of half-step h
  double* const pt = qn + (size_t)Ns * p;  // their log-prior terms
  double* const tf = pt + (size_t)Ns * p;  // accept test of half-step h - 1: (p - 1) log z ...
  double* const tu = tf + Ns;              // ... and log u
  for (int e = tid; e < W * p; e += nt) co[e] = a.coords[e];
  for (int e = tid; e < W; e += nt) lg[e] = a.logp[e];
  // the resets the LML batch would otherwise enqueue as dispatches of their own (the previous batch, and with it every reader
  // of these words, is over: this kernel runs behind it on the stream); the previous launch-free call's error word first
  if (tid == 0 && h > 0 && a.ps_err && *a.ps_err != 0) a.info[1] = 1u;
  const int g = h - 1;
  const int* const mvg = a.movers + (size_t)(g > 0 ? g : 0) * Ns;
  if (h > 0)
    for (int i = tid; i < Ns; i += nt) {  // (the accept test's operands are requested with the ensemble)
      double lp = a.prior[i] + a.lml[i];
      if (!(lp > -INFINITY && lp < INFINITY)) lp = -INFINITY;  // (NaN included, as _log_prob_finish)
      af[i] = lp;  // (replaced by the accept flag below, by the same thread)
      tf[i] = a.factors[(size_t)g * Ns + i];
      tu[i] = a.logu[(size_t)g * Ns + i];
    }
  __syncthreads();
  if (h < a.nhalf) {
    if (a.ps_flags)
      for (int e = tid; e < a.ps_words; e += nt) a.ps_flags[e] = 0u;
    for (int e = tid; e < Ns; e += nt) a.status[e] = 0;
  }
  if (h > 0) {
    for (int i = tid; i < Ns; i += nt) {
      const int m = mvg[i];
      const double lp = af[i];
      const bool acc = tf[i] + lp - lg[m] > tu[i];
      af[i] = acc ? 1.0 : 0.0;
      if (acc) {
        lg[m] = lp;
        a.logp[m] = lp;
        a.nacc[m] += 1;
      }
    }
    __syncthreads();
    for (int e = tid; e < Ns * p; e += nt) {
      const int i = e / p;
      if (af[i] != 0.0) {
        const double v = a.q[e];
        co[(size_t)mvg[i] * p + (e - i * p)] = v;
        a.coords[(size_t)mvg[i] * p + (e - i * p)] = v;
      }
    }
    __syncthreads();
    if (g & 1) {  // second half of step g / 2: the ensemble goes into the chain
      const int step = g >> 1;
      for (int e = tid; e < W * p; e += nt) a.chain[(size_t)step * W * p + e] = co[e];
      for (int e = tid; e < W; e += nt) a.lps[(size_t)step * W + e] = lg[e];
    }
  }
  if (h >= a.nhalf) return;
  const int* mv = a.movers + (size_t)h * Ns;
  const int* pr = a.partners + (size_t)h * Ns;
  for (int e = tid; e < Ns * p; e += nt) {
    const int i = e / p, k = e - i * p;
    const double z = a.zz[(size_t)h * Ns + i];
    const double s = co[(size_t)mv[i] * p + k], c = co[(size_t)pr[i] * p + k];
    const double v = c - (c - s) * z;
    qn[e] = v;
    a.q[e] = v;
    if (!(v > -INFINITY && v < INFINITY)) a.info[0] = 1u;
    pt[e] = mcmc_prior(a.prior_kind[k], a.prior_par + 5 * k, v);
  }
  __syncthreads();
  for (int i = tid; i < Ns; i += nt) {
    double lp = 0.0;
    for (int k = 0; k < p; k++) lp += pt[(size_t)i * p + k];
    a.prior[i] = lp;
  }
  for (int e = tid; e < Ns * a.hp; e += nt) {
    const int i = e / a.hp, j = e - i * a.hp;
    a.dh[e] = a.h_src[j] >= 0 ? qn[(size_t)i * p + a.h_src[j]] : a.h_fixed[j];
  }
}

namespace {
struct DevBlock {  // one allocation for the whole run, released with it
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

// An open run (bgp_mcmc_begin .. bgp_mcmc_end): device block, kernel arguments, how far the plan has been enqueued, and host
// copies of the start ensemble and of the plan handed over so far (a run whose launch-free factorisation timed out is redone
// from them on the launch schedule).
struct bgp_mcmc_state {
  DevBlock blk;
  McmcArgs a;
  int nsteps = 0, threads = 256;
  int enq_half = 0;  // half-steps enqueued so far
  int failed = BGP_OK;
  std::vector<double> coords0, logp0, zz, factors, logu;
  std::vector<int> movers, partners;
};

static void mcmc_drain(bgp_ctx* c) {
  (void)hipStreamSynchronize(c->stream);
  for (int g = 0; g < BGP_MAX_STREAMS; g++)
    if (c->gstream[g]) (void)hipStreamSynchronize(c->gstream[g]);
  bgp_xfer_drop_pending();
  (void)hipGetLastError();
}

// bgp_ctx_destroy / a failed call: the run is dropped, whatever it had enqueued is drained
void bgp_mcmc_abandon(bgp_ctx* c) {
  if (!c || !c->mcmc) return;
  mcmc_drain(c);
  delete c->mcmc;
  c->mcmc = nullptr;
  c->ps_resident = 0;
  c->ps_inflight = 0;
  if (c->pending_B < 0) c->pending_B = 0;
}

// half-steps [h0, h1) of the plan that is on the device: the step kernel that opens each (and closes its predecessor) + its LML
// batch; n <= 128: the fused kernel alone
static int mcmc_enqueue(bgp_ctx* c, bgp_mcmc_state* r, int h0, int h1) {
  McmcArgs& a = r->a;
  hipStream_t st = c->stream;
  int rc = BGP_OK;
  c->ps_resident = 1;
  if (c->nblk == 1) {  // proposal, Gram build, factorisation and accept test in ONE launch per half-step
    for (int h = h0; h < h1 && rc == BGP_OK; h++) rc = bgp_launch_mcmc_small(c, st, a, h);
  } else {
    for (int h = h0; h < h1 && rc == BGP_OK; h++) {
      hipLaunchKernelGGL(mcmc_step_kernel, dim3(1), dim3(r->threads), 0, st, a, h);
      c->ps_inflight = 0;
      rc = bgp_lml_enqueue_dev(c, a.Ns, 0);
      // (the launch-free kernel's error word is reset in front of every call: the next step kernel folds it into info[1])
      a.ps_err = (rc == BGP_OK && c->ps_inflight && c->ps_flags) ? c->ps_flags + PS_ERROR : nullptr;
    }
  }
  c->ps_resident = 0;
  c->ps_inflight = 0;
  if (rc == BGP_OK) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      bgp_set_error("bgp_mcmc: %s", hipGetErrorString(e));
      rc = BGP_ERR_HIP;
    }
  }
  return rc;
}

static int mcmc_upload_start(bgp_ctx* c, bgp_mcmc_state* r) {
  McmcArgs& a = r->a;
  BGP_HIP(bgp_memcpy_async(a.coords, r->coords0.data(), r->coords0.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(a.logp, r->logp0.data(), r->logp0.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemsetAsync(a.nacc, 0, (size_t)a.W * sizeof(long long), c->stream));
  BGP_HIP(hipMemsetAsync(a.info, 0, 2 * sizeof(unsigned), c->stream));
  a.ps_err = nullptr;
  return BGP_OK;
}

// See include/bgp.h.
extern "C" int bgp_mcmc_begin(bgp_ctx* c, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                              const double* prior_par, const double* coords0, const double* logp0) {
  if (!c || !h_src || !h_fixed || !prior_kind || !prior_par || !coords0 || !logp0 || W < 2 || (W & 1) || p < 1 || nsteps < 1) {
    bgp_set_error("bgp_mcmc_begin: bad argument (W must be even and >= 2)");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_mcmc_begin");
  const int Ns = W / 2, hp = c->d + 2, nhalf = 2 * nsteps;
  if (Ns > c->max_batch) {
    bgp_set_error("bgp_mcmc_begin: %d proposals per half-step exceed max_batch = %d", Ns, c->max_batch);
    return BGP_ERR_INVALID;
  }
  if (c->nblk > 1 && (size_t)W * p + W + (size_t)3 * Ns + (size_t)2 * Ns * p > MCMC_LDS_DOUBLES) {
    bgp_set_error("bgp_mcmc_begin: an ensemble of %d walkers x %d entries does not fit the step kernel's LDS", W, p);
    return BGP_ERR_INVALID;
  }
  if (c->timing) {
    bgp_set_error("bgp_mcmc_begin: per-launch timing is on (bgp_set_timing): use the host-driven sampler");
    return BGP_ERR_STATE;
  }
  for (int k = 0; k < p; k++)
    if (prior_kind[k] != 1 && prior_kind[k] != 2) {
      bgp_set_error("bgp_mcmc_begin: prior kind %d of entry %d is not one of the device's (1 half-Normal, 2 round-flat)", prior_kind[k], k);
      return BGP_ERR_INVALID;
    }
  for (int j = 0; j < hp; j++)
    if (h_src[j] >= p) {
      bgp_set_error("bgp_mcmc_begin: canonical entry %d reads walker entry %d of %d", j, h_src[j], p);
      return BGP_ERR_INVALID;
    }
  BGP_HIP(hipSetDevice(c->device));
  const size_t plan = (size_t)nhalf * Ns;
  std::unique_ptr<bgp_mcmc_state> r(new bgp_mcmc_state);
  McmcArgs& a = r->a;
  for (int pass = 0; pass < 2; pass++) {  // pass 0 sizes the block, pass 1 hands out the pointers
    DevBlock& blk = r->blk;
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
  a.ps_err = nullptr;
  if (c->nblk > 1 && bgp_persist_fits(c, Ns)) {  // a launch-free call may follow: its flag block is reset by the step kernel
    const int rcf = bgp_ps_ensure_flags(c, Ns);
    if (rcf) return rcf;
    a.ps_flags = c->ps_flags;
    a.ps_words = (int)ps_flag_words(Ns, c->nblk);
  }
  r->nsteps = nsteps;
  r->threads = 1024;  // (one pass over the (proposal, entry) pairs of every ensemble the reference's defaults produce)
  r->coords0.assign(coords0, coords0 + (size_t)W * p);
  r->logp0.assign(logp0, logp0 + W);
  hipStream_t st = c->stream;
  BGP_HIP(bgp_memcpy_async(const_cast<int*>(a.h_src), h_src, hp * sizeof(int), hipMemcpyHostToDevice, st));
  BGP_HIP(bgp_memcpy_async(const_cast<double*>(a.h_fixed), h_fixed, hp * sizeof(double), hipMemcpyHostToDevice, st));
  BGP_HIP(bgp_memcpy_async(const_cast<int*>(a.prior_kind), prior_kind, p * sizeof(int), hipMemcpyHostToDevice, st));
  BGP_HIP(bgp_memcpy_async(const_cast<double*>(a.prior_par), prior_par, (size_t)5 * p * sizeof(double), hipMemcpyHostToDevice, st));
  {
    const int rcu = mcmc_upload_start(c, r.get());
    if (rcu) return rcu;
  }
  c->mcmc = r.release();
  c->pending_B = -1;  // (every other entry point answers "busy" until bgp_mcmc_end)
  return BGP_OK;
}

extern "C" int bgp_mcmc_steps(bgp_ctx* c, int nseg, const int* movers, const int* partners, const double* zz, const double* factors,
                              const double* logu) {
  if (!c || !c->mcmc) {
    bgp_set_error("bgp_mcmc_steps: no run is open (bgp_mcmc_begin)");
    return BGP_ERR_STATE;
  }
  bgp_mcmc_state* r = c->mcmc;
  McmcArgs& a = r->a;
  if (nseg < 1 || !movers || !partners || !zz || !factors || !logu || r->enq_half + 2 * nseg > a.nhalf) {
    bgp_set_error("bgp_mcmc_steps: bad argument (%d steps behind %d of %d)", nseg, r->enq_half / 2, r->nsteps);
    return BGP_ERR_INVALID;
  }
  if (r->failed) return r->failed;
  BGP_HIP(hipSetDevice(c->device));
  const size_t cnt = (size_t)2 * nseg * a.Ns, off = (size_t)r->enq_half * a.Ns;
  r->movers.insert(r->movers.end(), movers, movers + cnt);
  r->partners.insert(r->partners.end(), partners, partners + cnt);
  r->zz.insert(r->zz.end(), zz, zz + cnt);
  r->factors.insert(r->factors.end(), factors, factors + cnt);
  r->logu.insert(r->logu.end(), logu, logu + cnt);
  hipStream_t st = c->stream;
  int rc = BGP_OK;
  auto up = [&](const void* dst, const void* src, size_t bytes) {
    if (rc == BGP_OK && bgp_memcpy_async(const_cast<void*>(dst), src, bytes, hipMemcpyHostToDevice, st) != hipSuccess) rc = BGP_ERR_HIP;
  };
  up(a.movers + off, movers, cnt * sizeof(int));
  up(a.partners + off, partners, cnt * sizeof(int));
  up(a.zz + off, zz, cnt * sizeof(double));
  up(a.factors + off, factors, cnt * sizeof(double));
  up(a.logu + off, logu, cnt * sizeof(double));
  if (rc == BGP_OK) rc = mcmc_enqueue(c, r, r->enq_half, r->enq_half + 2 * nseg);
  if (rc != BGP_OK) {
    r->failed = rc;
    mcmc_drain(c);
    return rc;
  }
  r->enq_half += 2 * nseg;
  return BGP_OK;
}

extern "C" int bgp_mcmc_end(bgp_ctx* c, double* chain, double* logp, double* coords_out, double* logp_out, long long* naccepted,
                            int* info) {
  if (!c || !c->mcmc) {
    bgp_set_error("bgp_mcmc_end: no run is open (bgp_mcmc_begin)");
    return BGP_ERR_STATE;
  }
  bgp_mcmc_state* r = c->mcmc;
  McmcArgs& a = r->a;
  struct Closer {  // the run ends here whatever happens
    bgp_ctx* c;
    ~Closer() { bgp_mcmc_abandon(c); }
  } closer{c};
  if (r->failed) return r->failed;
  if (!chain || !logp || !coords_out || !logp_out || !naccepted || !info || r->enq_half != a.nhalf) {
    bgp_set_error("bgp_mcmc_end: bad argument, or only %d of %d steps were handed over", r->enq_half / 2, r->nsteps);
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  hipStream_t st = c->stream;
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = BGP_OK;
    if (attempt == 1) {  // the whole run again, on the launch schedule (the plan is on the device already)
      rc = mcmc_upload_start(c, r);
      if (rc == BGP_OK) rc = mcmc_enqueue(c, r, 0, a.nhalf);
    }
    if (rc == BGP_OK && c->nblk > 1) {  // the step kernel that closes the last half-step
      hipLaunchKernelGGL(mcmc_step_kernel, dim3(1), dim3(r->threads), 0, st, a, a.nhalf);
      if (hipGetLastError() != hipSuccess) rc = BGP_ERR_HIP;
    }
    unsigned hinfo[2] = {0u, 0u};
    if (rc == BGP_OK && bgp_memcpy_async(hinfo, a.info, sizeof(hinfo), hipMemcpyDeviceToHost, st) != hipSuccess) rc = BGP_ERR_HIP;
    if (rc != BGP_OK || bgp_stream_sync(st) != hipSuccess) {
      if (rc == BGP_OK) {
        bgp_set_error("bgp_mcmc_end: the run failed on the device");
        rc = BGP_ERR_HIP;
      }
      return rc;  // (Closer drains)
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
  BGP_HIP(bgp_memcpy_async(chain, a.chain, (size_t)r->nsteps * a.W * a.p * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(logp, a.lps, (size_t)r->nsteps * a.W * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(coords_out, a.coords, (size_t)a.W * a.p * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(logp_out, a.logp, (size_t)a.W * sizeof(double), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_memcpy_async(naccepted, a.nacc, (size_t)a.W * sizeof(long long), hipMemcpyDeviceToHost, st));
  BGP_HIP(bgp_stream_sync(st));
  return BGP_OK;
}

// The three calls in one: the whole plan handed over at once.
extern "C" int bgp_mcmc_run(bgp_ctx* c, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                            const double* prior_par, const double* coords0, const double* logp0, const int* movers,
                            const int* partners, const double* zz, const double* factors, const double* logu, double* chain,
                            double* logp, double* coords_out, double* logp_out, long long* naccepted, int* info) {
  if (!movers || !partners || !zz || !factors || !logu || !chain || !logp || !coords_out || !logp_out || !naccepted || !info) {
    bgp_set_error("bgp_mcmc_run: NULL argument");
    return BGP_ERR_INVALID;
  }
  int rc = bgp_mcmc_begin(c, W, p, nsteps, h_src, h_fixed, prior_kind, prior_par, coords0, logp0);
  if (rc) return rc;
  rc = bgp_mcmc_steps(c, nsteps, movers, partners, zz, factors, logu);
  if (rc) {
    bgp_mcmc_abandon(c);
    return rc;
  }
  return bgp_mcmc_end(c, chain, logp, coords_out, logp_out, naccepted, info);
}
