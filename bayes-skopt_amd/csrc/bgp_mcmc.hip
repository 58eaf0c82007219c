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
// LDS (W p + W + 3 Ns + 2 Ns p doubles <= MCMC_LDS_DOUBLES) and the kernel is three rounds of memory latency -- everything in, the
// old proposals of the accepted walkers, everything out -- instead of one per phase.  An ensemble beyond that (HBM = true) works on
// the same arrays where they live in HBM (same operations, same order; one workgroup, so a barrier orders its own stores).
// Sharded runs (a.glml): every rank runs this kernel on the same inputs -- the plan, and the log-likelihoods of ALL proposals out
// of the all-gather that sits between the LML batch and this kernel on the stream -- and writes the canonical hyper-parameters of
// ITS rows only.
#define MCMC_LDS_DOUBLES 20480  // 160 KB: the whole LDS of a compute unit
template <bool HBM>
__global__ void __launch_bounds__(1024) mcmc_step_kernel(McmcArgs a, int h) {
#pragma clang fp contract(off)
  __shared__ double lds[HBM ? 8 : MCMC_LDS_DOUBLES];
  const int tid = threadIdx.x, nt = blockDim.x, p = a.p, Ns = a.Ns, W = a.W;
  double* const co = HBM ? a.coords : lds;                        // the ensemble
  double* const lg = HBM ? a.logp : co + (size_t)W * p;           // its log-probabilities
  double* const af = HBM ? a.scr : lg + W;                        // accept flags of half-step h - 1
  double* const qn = HBM ? a.q : af + Ns;                         // proposals of half-step h
  double* const pt = HBM ? a.pterm : qn + (size_t)Ns * p;         // their log-prior terms
  double* const tf = HBM ? a.scr + Ns : pt + (size_t)Ns * p;      // accept test of half-step h - 1: (p - 1) log z ...
  double* const tu = tf + Ns;                                     // ... and log u
  if (!HBM) {
    for (int e = tid; e < W * p; e += nt) co[e] = a.coords[e];
    for (int e = tid; e < W; e += nt) lg[e] = a.logp[e];
  }
  // the resets the LML batch would otherwise enqueue as dispatches of their own (the previous batch, and with it every reader
  // of these words, is over: this kernel runs behind it on the stream); the previous launch-free call's error word first, and
  // the status words the ranks of a sharded run sent with their values
  if (tid == 0 && h > 0) {
    if (a.ps_err && *a.ps_err != 0) a.info[1] = a.info[5] = 1u;  // ([5]: THIS context's own call timed out)
    if (a.glml)
      for (int r = 0; r < a.world; r++) {
        const double sw = a.glml[(size_t)r * a.slot + a.slot - 1];
        if (sw == (double)BGP_RANK_REDO) a.info[1] = 1u;
        else if (!(sw == 0.0)) a.info[4] = (sw > 0.0 && sw < 2e9) ? (unsigned)sw : (unsigned)BGP_ERR_COMM;
      }
  }
  const int g = h - 1;
  const int* const mvg = a.movers + (size_t)(g > 0 ? g : 0) * Ns;
  if (h > 0)
    for (int i = tid; i < Ns; i += nt) {  // (the accept test's operands are requested with the ensemble)
      double lp = a.prior[i] + (a.glml ? a.glml[a.lml_idx[i]] : a.lml[i]);
      if (!(lp > -INFINITY && lp < INFINITY)) lp = -INFINITY;  // (NaN included, as _log_prob_finish)
      af[i] = lp;  // (replaced by the accept flag below, by the same thread)
      tf[i] = a.factors[(size_t)g * Ns + i];
      tu[i] = a.logu[(size_t)g * Ns + i];
    }
  __syncthreads();
  if (h < a.nhalf) {
    if (a.ps_flags)
      for (int e = tid; e < a.ps_words; e += nt) a.ps_flags[e] = 0u;
    for (int e = tid; e < a.row_n; e += nt) a.status[e] = 0;
  }
  if (h > 0) {
    for (int i = tid; i < Ns; i += nt) {
      const int m = mvg[i];
      const double lp = af[i];
      const bool acc = tf[i] + lp - lg[m] > tu[i];
      af[i] = acc ? 1.0 : 0.0;
      if (acc) {
        lg[m] = lp;
        if (!HBM) a.logp[m] = lp;
        a.nacc[m] += 1;
      }
    }
    __syncthreads();
    for (int e = tid; e < Ns * p; e += nt) {
      const int i = e / p;
      if (af[i] != 0.0) {
        const double v = a.q[e];
        co[(size_t)mvg[i] * p + (e - i * p)] = v;
        if (!HBM) a.coords[(size_t)mvg[i] * p + (e - i * p)] = v;
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
    if (!HBM) a.q[e] = v;
    if (!(v > -INFINITY && v < INFINITY)) {  // (every writer of a launch writes the same values)
      a.info[0] = 1u;
      if (v != v) {
        if (a.info[3] == 0u) a.info[3] = (unsigned)h + 1u;
      } else if (a.info[2] == 0u) {
        a.info[2] = (unsigned)h + 1u;
      }
    }
    pt[e] = mcmc_prior(a.prior_kind[k], a.prior_par + 5 * k, v);
  }
  __syncthreads();
  const int ng = p - a.nwarp, dw = a.nwarp >> 1;
  for (int i = tid; i < Ns; i += nt) {
    double lp = 0.0;
    for (int k = 0; k < ng; k++) lp += pt[(size_t)i * p + k];
    if (dw) {  // _eval_warp_priors: the pairs (alpha_k, beta_k) summed on their own, column by column, then added
      double w = 0.0;
      for (int k = 0; k < dw; k++) w += pt[(size_t)i * p + ng + k] + pt[(size_t)i * p + ng + dw + k];
      lp = lp + w;
    }
    a.prior[i] = lp;
  }
  for (int e = tid; e < a.row_n * a.hp; e += nt) {
    const int i = e / a.hp, j = e - i * a.hp;
    a.dh[e] = a.h_src[j] >= 0 ? qn[(size_t)(a.row_lo + i) * p + a.h_src[j]] : a.h_fixed[j];
  }
  for (int e = tid; e < a.row_n * a.nwarp; e += nt) {
    const int i = e / a.nwarp, k = e - i * a.nwarp;
    a.dwarp[e] = qn[(size_t)(a.row_lo + i) * p + ng + k];
  }
}

static void mcmc_launch_step(const McmcArgs& a, int h, int threads, bool hbm, hipStream_t st) {
  if (hbm)
    hipLaunchKernelGGL(mcmc_step_kernel<true>, dim3(1), dim3(threads), 0, st, a, h);
  else
    hipLaunchKernelGGL(mcmc_step_kernel<false>, dim3(1), dim3(threads), 0, st, a, h);
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

// An open run (bgp_mcmc_begin .. bgp_mcmc_end): device block, kernel arguments, how far the plan has been enqueued, and a host
// copy of the start ensemble (a run whose launch-free factorisation timed out is redone on the launch schedule from it and from
// the plan, which stays on the device).
struct bgp_mcmc_state {
  DevBlock blk;
  McmcArgs a;
  int nsteps = 0, threads = 256;
  int enq_half = 0;  // half-steps enqueued so far
  int failed = BGP_OK;
  bool hbm = false;     // the ensemble does not fit the step kernel's LDS: the HBM form
  bool small = false;   // n <= 128: the fused one-launch half-step (mcmc_small_kernel)
  bool warped = false;  // walkers carry their own input warp
  bgp_comm* comm = nullptr;  // sharded ensemble: the communicator whose ranks share the half-steps' proposal blocks
  int per = 0;               // ... rows per rank (ceil(Ns / world)): the all-gather moves per + 1 doubles per rank
  std::vector<double> coords0, logp0;
  std::vector<hipEvent_t> seg_ev;  // one event behind every segment handed over (bgp_mcmc_progress)
  std::vector<int> seg_steps;      // steps complete when that event has passed
  int seg_done = 0;                // events known to have passed
  ~bgp_mcmc_state() {
    for (hipEvent_t e : seg_ev) (void)hipEventDestroy(e);
  }
};

static void mcmc_drain(bgp_ctx* c) {
  (void)hipStreamSynchronize(c->stream);
  for (int g = 0; g < BGP_MAX_STREAMS; g++)
    if (c->gstream[g]) (void)hipStreamSynchronize(c->gstream[g]);
  bgp_xfer_drop_pending();
  (void)hipGetLastError();
}

// bgp_ctx_destroy / a failed call: the run is dropped, whatever it had enqueued is drained.  A sharded run that has not handed
// all of its half-steps to the device takes the communicator down first: its peers have (or will have) collectives on their
// streams that this rank will never join, and ncclCommAbort makes them fail instead of wait.
void bgp_mcmc_abandon(bgp_ctx* c) {
  if (!c || !c->mcmc) return;
  bgp_mcmc_state* r = c->mcmc;
  if (r->comm && (r->failed || r->enq_half < r->a.nhalf)) (void)bgp_comm_abort(r->comm);
  mcmc_drain(c);
  delete r;
  c->mcmc = nullptr;
  c->ps_resident = 0;
  c->ps_inflight = 0;
  c->ps_forbid = 0;
  if (c->pending_B < 0) c->pending_B = 0;
}

// half-steps [h0, h1) of the plan that is on the device: the step kernel that opens each (and closes its predecessor) + the LML
// batch of this context's rows (+ sharded: the all-gather of every rank's log-likelihoods, on the same stream); n <= 128: the
// fused kernel alone
static int mcmc_enqueue(bgp_ctx* c, bgp_mcmc_state* r, int h0, int h1) {
  McmcArgs& a = r->a;
  hipStream_t st = c->stream;
  int rc = BGP_OK;
  c->ps_resident = 1;
  if (r->small) {  // proposal, Gram build, factorisation and accept test in ONE launch per half-step
    for (int h = h0; h < h1 && rc == BGP_OK; h++) rc = bgp_launch_mcmc_small(c, st, a, h);
  } else {
    for (int h = h0; h < h1 && rc == BGP_OK; h++) {
      mcmc_launch_step(a, h, r->threads, r->hbm, st);
      c->ps_inflight = 0;
      if (a.row_n > 0) rc = bgp_lml_enqueue_dev(c, a.row_n, r->warped ? 1 : 0);
      // (the launch-free kernel's error word is reset in front of every call: the next step kernel folds it into info[1])
      a.ps_err = (rc == BGP_OK && c->ps_inflight && c->ps_flags) ? c->ps_flags + PS_ERROR : nullptr;
      if (rc == BGP_OK && r->comm) rc = bgp_comm_enqueue_lml_gather(r->comm, c, st, a.row_n, r->per, a.ps_err);
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
  BGP_HIP(hipMemsetAsync(a.info, 0, MCMC_INFO_WORDS * sizeof(unsigned), c->stream));
  a.ps_err = nullptr;
  return BGP_OK;
}

// See include/bgp.h.
extern "C" int bgp_mcmc_begin_ex(bgp_ctx* c, bgp_comm* comm, int nwarp, int W, int p, int nsteps, const int* h_src,
                                 const double* h_fixed, const int* prior_kind, const double* prior_par, const double* coords0,
                                 const double* logp0) {
  if (!c || !h_src || !h_fixed || !prior_kind || !prior_par || !coords0 || !logp0 || W < 2 || (W & 1) || p < 1 || nsteps < 1) {
    bgp_set_error("bgp_mcmc_begin: bad argument (W must be even and >= 2)");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_mcmc_begin");
  const int Ns = W / 2, hp = c->d + 2, nhalf = 2 * nsteps;
  if (nwarp != 0 && (nwarp != 2 * c->d || p <= nwarp)) {
    bgp_set_error("bgp_mcmc_begin: nwarp = %d must be 0 or 2 d = %d, behind at least one kernel entry (p = %d)", nwarp, 2 * c->d, p);
    return BGP_ERR_INVALID;
  }
  int rank = 0, world = 1;
  if (comm) {
    const int rcc = bgp_comm_rank(comm, &rank, &world);
    if (rcc) return rcc;
  }
  const int per = (Ns + world - 1) / world;
  const int row_lo = (int)(((long long)Ns * rank) / world), row_hi = (int)(((long long)Ns * (rank + 1)) / world);
  if (row_hi - row_lo > c->max_batch) {
    bgp_set_error("bgp_mcmc_begin: %d proposals per half-step and rank exceed max_batch = %d", row_hi - row_lo, c->max_batch);
    return BGP_ERR_INVALID;
  }
  if (c->timing) {
    bgp_set_error("bgp_mcmc_begin: per-launch timing is on (bgp_set_timing): use the host-driven sampler");
    return BGP_ERR_STATE;
  }
  for (int k = 0; k < p; k++)
    if (prior_kind[k] < 1 || prior_kind[k] > 3) {
      bgp_set_error("bgp_mcmc_begin: prior kind %d of entry %d is not one of the device's (1 half-Normal, 2 round-flat, 3 Normal)",
                    prior_kind[k], k);
      return BGP_ERR_INVALID;
    }
  for (int j = 0; j < hp; j++)
    if (h_src[j] >= p - nwarp) {
      bgp_set_error("bgp_mcmc_begin: canonical entry %d reads walker entry %d of %d", j, h_src[j], p - nwarp);
      return BGP_ERR_INVALID;
    }
  BGP_HIP(hipSetDevice(c->device));
  if (nwarp) {
    const int rcw = bgp_ensure_warp_buffers(c);
    if (rcw) return rcw;
  }
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
    a.scr = blk.take<double>((size_t)3 * Ns);
    a.h_src = blk.take<int>(hp);
    a.h_fixed = blk.take<double>(hp);
    a.prior_kind = blk.take<int>(p);
    a.prior_par = blk.take<double>((size_t)5 * p);
    a.lml_idx = blk.take<int>(Ns);
    a.movers = blk.take<int>(plan);
    a.partners = blk.take<int>(plan);
    a.zz = blk.take<double>(plan);
    a.factors = blk.take<double>(plan);
    a.logu = blk.take<double>(plan);
    a.chain = blk.take<double>((size_t)nsteps * W * p);
    a.lps = blk.take<double>((size_t)nsteps * W);
    a.info = blk.take<unsigned>(MCMC_INFO_WORDS);
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
  a.row_lo = row_lo;
  a.row_n = row_hi - row_lo;
  a.world = world;
  a.slot = per + 1;
  a.glml = nullptr;
  a.nwarp = nwarp;
  a.dwarp = nwarp ? c->dwarpB : nullptr;
  a.dh = c->dh;
  a.lml = c->dlml;
  a.status = c->dstatus;
  a.ps_flags = nullptr;
  a.ps_words = 0;
  a.ps_err = nullptr;
  r->comm = comm;
  r->per = per;
  r->warped = nwarp != 0;
  r->small = c->nblk == 1 && !comm && !nwarp && p <= 64 && hp <= 64;
  r->hbm = !r->small && (size_t)W * p + W + (size_t)3 * Ns + (size_t)2 * Ns * p > MCMC_LDS_DOUBLES;
  hipStream_t st = c->stream;
  if (comm) {
    a.glml = bgp_comm_recv(comm, (size_t)a.slot * world);
    if (!a.glml) {
      bgp_set_error("bgp_mcmc_begin: the communicator has no receive buffer of %d x %d doubles", world, a.slot);
      return BGP_ERR_HIP;
    }
    std::vector<int> idx(Ns);
    for (int rr = 0; rr < world; rr++) {
      const int lo = (int)(((long long)Ns * rr) / world), hi = (int)(((long long)Ns * (rr + 1)) / world);
      for (int i = lo; i < hi; i++) idx[i] = rr * a.slot + (i - lo);
    }
    BGP_HIP(bgp_memcpy_async(const_cast<int*>(a.lml_idx), idx.data(), Ns * sizeof(int), hipMemcpyHostToDevice, st));
  }
  if (c->nblk > 1 && a.row_n > 0 && bgp_persist_fits(c, a.row_n)) {  // a launch-free call may follow: its flag block is reset by the step kernel
    const int rcf = bgp_ps_ensure_flags(c, a.row_n);
    if (rcf) return rcf;
    a.ps_flags = c->ps_flags;
    a.ps_words = (int)ps_flag_words(a.row_n, c->nblk);
  }
  r->nsteps = nsteps;
  r->threads = 1024;  // (one pass over the (proposal, entry) pairs of every ensemble the reference's defaults produce)
  r->coords0.assign(coords0, coords0 + (size_t)W * p);
  r->logp0.assign(logp0, logp0 + W);
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

extern "C" int bgp_mcmc_begin(bgp_ctx* c, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                              const double* prior_par, const double* coords0, const double* logp0) {
  return bgp_mcmc_begin_ex(c, nullptr, 0, W, p, nsteps, h_src, h_fixed, prior_kind, prior_par, coords0, logp0);
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
    if (r->comm) (void)bgp_comm_abort(r->comm);  // (the peers' collectives behind this point will never see this rank)
    mcmc_drain(c);
    return rc;
  }
  r->enq_half += 2 * nseg;
  {  // a mark behind the segment for bgp_mcmc_progress (the steps of the segment are complete up to its last accept, which
     // the NEXT step kernel performs: a progress bar does not mind)
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev, st) == hipSuccess) {
      r->seg_ev.push_back(ev);
      r->seg_steps.push_back(r->enq_half / 2);
    } else {
      if (ev) (void)hipEventDestroy(ev);
      (void)hipGetLastError();
    }
  }
  return BGP_OK;
}

// How far the device has got: steps of the open run whose segment has been worked through.  Never blocks.
extern "C" int bgp_mcmc_progress(bgp_ctx* c, int* steps_done) {
  if (!c || !c->mcmc || !steps_done) {
    bgp_set_error("bgp_mcmc_progress: no run is open (bgp_mcmc_begin), or NULL argument");
    return BGP_ERR_STATE;
  }
  bgp_mcmc_state* r = c->mcmc;
  while (r->seg_done < (int)r->seg_ev.size()) {
    const hipError_t e = hipEventQuery(r->seg_ev[r->seg_done]);
    if (e != hipSuccess) {
      (void)hipGetLastError();  // (hipErrorNotReady, or a failure bgp_mcmc_end will report)
      break;
    }
    r->seg_done++;
  }
  *steps_done = r->seg_done ? r->seg_steps[r->seg_done - 1] : 0;
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
  // a sharded run waits with the communicator's bound: a peer that died leaves collectives on this stream that never complete
  auto wait = [&]() -> int {
    if (r->comm) {
      const int rw = bgp_comm_wait_stream(r->comm, st, "bgp_mcmc_end");
      if (rw) return rw;
      return bgp_stream_sync(st) == hipSuccess ? BGP_OK : BGP_ERR_HIP;  // (passed already: unpacks the staged downloads)
    }
    return bgp_stream_sync(st) == hipSuccess ? BGP_OK : BGP_ERR_HIP;
  };
  for (int attempt = 0; attempt < 2; attempt++) {
    int rc = BGP_OK;
    if (attempt == 1) {  // the whole run again, on the launch schedule (the plan is on the device already)
      c->ps_forbid = 1;  // (every rank of a sharded run: the ranks that did not time out redo theirs alike)
      rc = mcmc_upload_start(c, r);
      if (rc == BGP_OK) rc = mcmc_enqueue(c, r, 0, a.nhalf);
      c->ps_forbid = 0;
    }
    if (rc == BGP_OK && !r->small) {  // the step kernel that closes the last half-step
      mcmc_launch_step(a, a.nhalf, r->threads, r->hbm, st);
      if (hipGetLastError() != hipSuccess) rc = BGP_ERR_HIP;
    }
    unsigned hinfo[MCMC_INFO_WORDS] = {0u};
    if (rc == BGP_OK && bgp_memcpy_async(hinfo, a.info, sizeof(hinfo), hipMemcpyDeviceToHost, st) != hipSuccess) rc = BGP_ERR_HIP;
    if (rc == BGP_OK) rc = wait();
    if (rc != BGP_OK) {
      if (rc == BGP_ERR_HIP) bgp_set_error("bgp_mcmc_end: the run failed on the device");
      r->failed = rc;  // (Closer drains; a sharded run takes its communicator down)
      return rc;
    }
    if (hinfo[4] != 0) {  // a rank reported a failure of its own through its status word: every rank sees the same word
      bgp_set_error("bgp_mcmc_end: a rank of the sharded run reported error %u in its share", hinfo[4]);
      r->failed = BGP_ERR_COMM;
      return BGP_ERR_COMM;
    }
    if (hinfo[1] != 0 && attempt == 0) {
      // a launch-free factorisation gave its waits up somewhere in the run: everything behind it is void.  The whole run is
      // redone on the launch schedule (bit-identical results), loudly, and the context's time-out policy takes note.
      if (r->comm && !hinfo[5])
        fprintf(stderr, "libbgp: a peer's launch-free factorisation timed out: this rank redoes the sampler's run with it\n");
      else
        bgp_ps_note_timeout(c, "the sampler's run is redone");
      continue;
    }
    info[0] = (int)hinfo[0];
    info[1] = attempt;
    info[2] = hinfo[2] && (!hinfo[3] || hinfo[2] <= hinfo[3]) ? (int)hinfo[2] - 1 : (hinfo[3] ? (int)hinfo[3] - 1 : -1);
    info[3] = hinfo[3] && (!hinfo[2] || hinfo[3] < hinfo[2]) ? 1 : 0;
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
