// Batched right-looking blocked Cholesky + fused forward substitution + LML
// (SURVEY.md 8a rows a2, a3).
//
// Replaces, per walker:  L = cholesky(K, lower=True)          sklearn/_gpr.py:587 (LAPACK dpotrf)
//                        alpha = cho_solve((L, True), y)      sklearn/_gpr.py:597 (dpotrs)
//                        -1/2 y.alpha - sum log diag L - n/2 log 2pi       sklearn/_gpr.py:609-613
// using  y^T K^-1 y = z^T z  with  z = L^-1 y  (one forward substitution, fused into the panel
// kernels; the back substitution is only needed for posterior builds, bgp_post.hip).
//
// Block size NB = 128; per block column k, batched over the B walkers:
//   potrf_kernel  one workgroup per walker: diagonal block in LDS -> L_kk, W_kk = L_kk^-1,
//                 z_k = W_kk y_k, running log-det and z^T z                    (latency: 128 sequential pivots)
//   panel solve   X_i = A_ik W_kk^T  (fp64 MFMA),  y_i -= X_i z_k
//   trailing upd. A_ij -= X_i X_j^T  (fp64 MFMA; the n^3/3 bulk -- the kernel the roofline fraction is quoted on)
// Panel solve and trailing update are trsm4_kernel / syrk4_kernel on the LDS-DMA ring (bgp_syrk4.hip):
//   * LML path (bgp_lml_batch; the MCMC hot loop): scheduled in groups of P block columns by bgp_launch_cholesky_slice;
//   * posterior builds on the augmented matrix (bgp_post.hip; once per sample(), per hyper-posterior draw of an
//     acquisition and per gradient evaluation): the same kernels in single-panel mode with the active-row remap of
//     bgp_rowblk (bgp_device.h);
//   * small batches of the LML path and sample_y's covariance: ONE persistent kernel instead of the launches
//     (bgp_launch_cholesky_persist below, ps_kernel in bgp_syrk4.hip, the diagonal-block code of both in bgp_pf.h).
// (Round 1's VGPR-staged trsm_kernel / syrk_kernel / syrk2_kernel / trsm8_kernel and the left-looking variant are A/B
// references of the benches under tools/legacy/ now; they are not part of libbgp.so.)
#include "bgp_common.h"
#include "bgp_device.h"

#include "bgp_gemm.h"

#include <cstdlib>
#include <vector>

// ------------------------------------------------------------------------------------------
// potrf: diagonal block k of every walker, one workgroup (8 waves) per walker, block in LDS (pf_block, bgp_pf.h).
//
// The 128x128 block is processed as 8x8 sub-blocks of 16 by a two-stage software pipeline inside the
// workgroup.  Wave 0 is the PANEL wave: per step sb it completes row block sb (panel product with the
// previous 16x16 inverse, last rank-16 term of the diagonal block), factorises the diagonal block in
// registers (one matrix row per lane, 64-bit DPP row broadcasts) and publishes its inverse M_sb.  Waves 1-3 and
// 5-7 are UPDATE waves running one phase behind: phase p (after M_p is published) forms the panel blocks X_{I,p}
// of the rows I >= p+2 and applies
//     column p+1: terms t = p-1, p        column p+2: terms t <= p-1  (and t = p on its diagonal block)
// so every block column c is complete (terms t <= c-3 in phase c-2, t = c-2, c-1 in phase c-1) when the
// panel wave needs it; they also form row sb-1 of W = L^-1 while the panel wave factorises block sb.  The
// only thing on the critical path is the 16-pivot chain (8 x ~2 us); one barrier per step.
// Panel products are formed transposed (X^T = M T^T): their C-layout registers are directly the A and
// the B operand of the following rank-16 updates (for v_mfma_f64_16x16x4_f64 the C/D row map
// (lane>>4)+4*reg coincides with the A and the B operand's k map for k-slice reg), so nothing
// round-trips through LDS.  W is kept TRANSPOSED in the otherwise unused upper triangle of the LDS
// tile, where later rows read it with the ordinary row-major operand pattern; z_k = W y_k, log-det and
// z^T z finish the launch.
// ------------------------------------------------------------------------------------------
#ifdef PF_TRACE  // phase timestamps of workgroup 0 (tools/potrf_bench.hip); compiled out of the product library
__device__ unsigned long long pf_trace[32];
__device__ unsigned long long pf_trace_w[8 * 32];  // per wave: [2 sb] start of the wave's step, [2 sb + 1] its end (before the barrier)
#define PF_T(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) pf_trace[i] = wall_clock64(); } while (0)
#define PF_TW(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) pf_trace_w[(threadIdx.x >> 6) * 32 + (i)] = wall_clock64(); } while (0)
extern "C" int bgp_debug_potrf_trace(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_trace), sizeof(pf_trace)) == hipSuccess ? 0 : 1;
}
extern "C" int bgp_debug_potrf_trace_w(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pf_trace_w), sizeof(pf_trace_w)) == hipSuccess ? 0 : 1;
}
#else
#define PF_T(i)
#define PF_TW(i)
#endif
#include "bgp_pf.h"
#include "bgp_mcmc.h"

template <int GEN, int STAT, int FORM>
__global__ void __launch_bounds__(PF_THREADS) potrf_kernel(double* __restrict__ Kbuf, double* __restrict__ Wbuf,
                                                     double* __restrict__ yw, double* __restrict__ accb,
                                                     double* __restrict__ lml, int* __restrict__ status, int n,
                                                     int ld, size_t mstride, int ystride, int nblk, int k, PfGen gen) {
  const int b = blockIdx.x;
  if (!GEN && status[b] != 0) return;
  PfPre nopre;  // (launch schedule: nothing to request ahead)
  nopre.state = 0;
  (void)pf_block<GEN, STAT, FORM, 0>(b, Kbuf, Wbuf, yw, accb, lml, status, n, ld, mstride, ystride, nblk, k, gen, false, nullptr,
                                     PsArgs(), nopre);
}

// Device-resident ensemble sampler, n <= 128 (bgp_mcmc.hip): half-step h in ONE launch.  Workgroup i = proposal i of the half-step:
// q = c - (c - s) z from the ensemble in HBM, its log-prior (terms summed in theta order) and canonical hyper-parameters; the Gram
// block, its factorisation and the log-likelihood exactly as potrf_kernel<1, ...> (pf_block); then the accept test of ITS walker --
// the only writer of that walker's row, and nobody reads a mover's row in the half-step in which it moves (partners come from the
// other half) -- and the walker's row of the chain (a walker moves once per step: its row of step h / 2 is final here).
template <int STAT, int FORM>
__global__ void __launch_bounds__(PF_THREADS) mcmc_small_kernel(McmcArgs a, int h, double* __restrict__ lml, int* __restrict__ status,
                                                               int n, PfGen gen) {
#pragma clang fp contract(off)
  __shared__ double sh_qv[64], sh_pt[64];
  __shared__ double sh_prior;
  __shared__ int sh_acc;
  const int i = blockIdx.x, tid = threadIdx.x, p = a.p;
  const int m = a.movers[(size_t)h * a.Ns + i], pr = a.partners[(size_t)h * a.Ns + i];
  if (tid < p) {
    const double z = a.zz[(size_t)h * a.Ns + i];
    const double s = a.coords[(size_t)m * p + tid], c = a.coords[(size_t)pr * p + tid];
    const double v = c - (c - s) * z;
    sh_qv[tid] = v;
    if (!(v > -INFINITY && v < INFINITY)) {
      a.info[0] = 1u;
      if (v != v) {
        if (a.info[3] == 0u) a.info[3] = (unsigned)h + 1u;
      } else if (a.info[2] == 0u) {
        a.info[2] = (unsigned)h + 1u;
      }
    }
    sh_pt[tid] = mcmc_prior(a.prior_kind[tid], a.prior_par + 5 * tid, v);
  }
  __syncthreads();
  if (tid == 0) {
    double lp = 0.0;
    for (int k = 0; k < p; k++) lp += sh_pt[k];
    sh_prior = lp;
  }
  if (tid < a.hp) a.dh[(size_t)i * a.hp + tid] = a.h_src[tid] >= 0 ? sh_qv[a.h_src[tid]] : a.h_fixed[tid];
  __syncthreads();  // (the hyper-parameters are read back from memory by this workgroup only)
  PfPre nopre;
  nopre.state = 0;
  (void)pf_block<1, STAT, FORM, 0>(i, nullptr, nullptr, nullptr, nullptr, lml, status, n, 128, (size_t)0, 0, 1, 0, gen, false, nullptr,
                                   PsArgs(), nopre);
  __syncthreads();
  if (tid == 0) {
    double lp = sh_prior + lml[i];
    if (!(lp > -INFINITY && lp < INFINITY)) lp = -INFINITY;
    const bool acc = a.factors[(size_t)h * a.Ns + i] + lp - a.logp[m] > a.logu[(size_t)h * a.Ns + i];
    if (acc) {
      a.logp[m] = lp;
      a.nacc[m] += 1;
    }
    sh_acc = acc ? 1 : 0;
    a.lps[(size_t)(h >> 1) * a.W + m] = acc ? lp : a.logp[m];
  }
  __syncthreads();
  if (tid < p) {
    const double v = sh_acc ? sh_qv[tid] : a.coords[(size_t)m * p + tid];
    if (sh_acc) a.coords[(size_t)m * p + tid] = v;
    a.chain[((size_t)(h >> 1) * a.W + m) * p + tid] = v;
  }
}

int bgp_launch_mcmc_small(bgp_ctx* ctx, hipStream_t st, const McmcArgs& a, int h) {
  if (a.p > 64 || a.hp > 64) {
    bgp_set_error("bgp_launch_mcmc_small: more than 64 entries per walker");
    return BGP_ERR_INVALID;
  }
  PfGen g;
  g.X = ctx->dXeff;
  g.alpha = ctx->dalpha;
  g.H = ctx->dh;
  g.y = ctx->dy;
  g.d = ctx->d;
  KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
              hipLaunchKernelGGL((mcmc_small_kernel<S, F>), dim3(a.Ns), dim3(PF_THREADS), 0, st, a, h, ctx->dlml, ctx->dstatus, ctx->n, g));
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

// Debugging aid / accuracy test: the pivot root of the diagonal-block factorisation (pf_pivot_root) on n arguments.
__global__ void pivot_root_kernel(const double* __restrict__ x, double* __restrict__ s, double* __restrict__ iv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a, b;
  pf_pivot_root(x[i], a, b);
  s[i] = a;
  iv[i] = b;
}
extern "C" int bgp_debug_pivot_root(int device, int n, const double* x, double* sqrt_out, double* rsqrt_out) {
  if (n <= 0 || !x || !sqrt_out || !rsqrt_out) {
    bgp_set_error("bgp_debug_pivot_root: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(device));
  double* d = nullptr;
  BGP_HIP(hipMalloc(&d, (size_t)3 * n * sizeof(double)));
  hipError_t e = hipMemcpy(d, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(pivot_root_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d, d + n, d + 2 * (size_t)n, n);
    e = hipMemcpy(sqrt_out, d + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess) e = hipMemcpy(rsqrt_out, d + 2 * (size_t)n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  BGP_HIP(e);
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// Launch-free factorisation of SMALL batches (B <= 64 matrices, or one large matrix): ONE persistent kernel per batch
// instead of ~3 dependent launches per block column (ps_kernel, bgp_syrk4.hip; the diagonal-block code is bgp_pf.h).
//
// Why: with few matrices the chain of dependent launches is the critical path -- potrf(k) keeps B of the 256 CUs busy
// for 30 us while the rest idle, and the trailing update of step k cannot overlap the next panel (stream / event
// look-ahead was measured slower than the chain it shortens: events cost more than they hide).  Here the diagonal-block
// chain and the tile work run SIDE BY SIDE inside one launch and talk through device-scope flags:
//   * chain role (workgroups 0 .. B-1, one per matrix; ps_chain_role, bgp_pf.h): walks the block columns J = 0 .. nblk-1:
//     factorises block (J, J) (pf_block: the same code as potrf_kernel), publishes L_JJ / W_JJ / z_J (wready[J]), then
//     solves block (J+1, J) and applies the last panel to block (J+1, J+1) ITSELF -- the next tile never leaves its LDS;
//   * tile role (the other workgroups, one per remaining CU; ps_tile_role, bgp_syrk4.hip): draw left-looking block tasks
//     from ticket counters in an order that is topological for the dependency graph -- panel solves S(I, J), I >= J+2, and
//     the pre-updates that hand blocks (I, I-1) and (I, I) to the chain -- waiting on / raising xready, subrdy, diagrdy.
// Both roles need a whole CU (157 / 128 KB of LDS, one array), the grid has at most one workgroup per CU and the
// dispatcher places workgroups in index order: the chain's B workgroups are resident before any tile workgroup, whatever
// else the GPU is doing.  (Round 3's first version ran two kernels on a pair of CU-masked streams: event hops, cold
// hardware queues -- ~45 us per call -- masks that only place properly for 1-4 or 8 CUs per XCD, and time-sliced queues
// once a few dozen masked streams were alive.)
// Every wait is bounded (PsArgs::spin_limit): a timeout raises the error word, both roles drain, and the host redoes
// the batch on the multi-launch path.  Arithmetic, operand order and summation order are those of the multi-launch path:
// the log-likelihoods are bit-identical (tests/test_gpu_persist.py).
// ------------------------------------------------------------------------------------------
int bgp_ps_total_tasks(int B, int nblk, int np, int gen);
void bgp_launch_ps(hipStream_t st, const PsArgs& a, int nwg);

// BGP_PS_TRACE=1: the time stamps of the last launch-free call (100 MHz wall clock): dims = {B, nblk, total tasks};
// chain (B x nblk x 8) then tile (total x 8): see tools/persist_trace.py.
extern "C" int bgp_debug_ps_trace(bgp_ctx* c, int* dims, unsigned long long* out, size_t cap) {
  if (!c || !dims) return BGP_ERR_INVALID;
  dims[0] = c->ps_trace_B;
  dims[1] = c->ps_trace_nblk;
  dims[2] = c->ps_trace_total;
  const size_t need = (size_t)c->ps_trace_B * c->ps_trace_nblk * 8 + (size_t)c->ps_trace_total * 8;
  if (!out || !c->ps_trace || need == 0) return BGP_OK;
  if (cap < need) return BGP_ERR_INVALID;
  BGP_HIP(hipSetDevice(c->device));
  BGP_HIP(hipMemcpy(out, c->ps_trace, need * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return BGP_OK;
}

// Can this batch take the launch-free path?  (at least two block columns; a CU per matrix and at least as many, and at
// least 32, left for the tile workers -- a partitioned device with 32 CUs never takes it)
int bgp_persist_fits(bgp_ctx* c, int B) {
  return B >= 1 && B <= 64 && c->nblk >= 2 && c->nblk <= 255 && c->ncu - B >= std::max(32, B);
}

// Host side of the launch-free factorisation: the B Gram matrices of the batch are already on c->stream (K-build);
// this enqueues ONE kernel behind them -- B chain workgroups + one tile workgroup for every other CU -- and the copy of
// the error word to pinned memory (ctx->ps_herr): != 0 after the synchronisation means a wait timed out and the caller
// redoes the batch on the multi-launch path.
// the flag block of a launch-free call with B matrices exists (grown with 50 % head room; contents undefined)
int bgp_ps_ensure_flags(bgp_ctx* c, int B) {
  const size_t words = ps_flag_words(B, c->nblk);
  if (words > c->cap_psflags) {
    if (c->ps_flags) (void)hipFree(c->ps_flags);
    c->ps_flags = nullptr;
    c->cap_psflags = 0;
    BGP_HIP(hipMalloc(&c->ps_flags, (words + words / 2) * sizeof(unsigned)));
    c->cap_psflags = words + words / 2;
  }
  return BGP_OK;
}

int bgp_launch_cholesky_persist(bgp_ctx* c, int B, int build_gram) {
  const int nblk = c->nblk, ld = c->npad;
  if (!bgp_persist_fits(c, B)) {
    bgp_set_error("bgp_launch_cholesky_persist: B = %d, nblk = %d outside the persistent path's range", B, nblk);
    return BGP_ERR_INVALID;
  }
  const int ncu = c->ncu;
  if (!c->ps_herr) {
    BGP_HIP(hipHostMalloc((void**)&c->ps_herr, sizeof(unsigned), hipHostMallocDefault));
    *c->ps_herr = 0;
  }
  const size_t words = ps_flag_words(B, nblk);
  {
    const int rcf = bgp_ps_ensure_flags(c, B);
    if (rcf) return rcf;
  }
  // (the device-resident sampler's step kernel has zeroed the block in front of this call: one dispatch less per half-step)
  if (!c->ps_resident) BGP_HIP(hipMemsetAsync(c->ps_flags, 0, words * sizeof(unsigned), c->stream));
  static unsigned long long limit = 0;
  if (!limit) {
    const char* e = getenv("BGP_PS_TIMEOUT_MS");
    // 100 MHz wall clock; default 500 ms: a healthy call lasts at most ~20 ms, and two PROCESSES that share a GPU and meet in
    // this path can block each other's resident workgroups (measured: one time-out in 1 800 calls each, results correct)
    limit = 100000ull * (unsigned long long)((e && atoi(e) > 0) ? atoi(e) : 500);
    const char* et = getenv("BGP_PS_TIMEOUT_TICKS");  // (tests: a bound no wait can meet)
    if (et && atoll(et) > 0) limit = (unsigned long long)atoll(et);
  }
  PsArgs a;
  a.K = c->dK;
  a.W = c->dW;
  a.yw = c->dyw;
  a.acc = c->dacc;
  a.lml = c->dlml;
  a.status = c->dstatus;
  a.flags = c->ps_flags;
  a.n = c->n;
  a.ld = ld;
  a.nblk = nblk;
  a.B = B;
  a.ystride = ld;
  a.mstride = (size_t)ld * ld;
  {
    // chain pairs (bgp_pf.h): two workgroups per matrix alternate over the block columns, the idle one preparing the next
    // diagonal block UNDER the other's factorisation; needs 2 * Bpad CUs and at least as many (and 32) left for the tile role.
    // BGP_PS_PAIR = 0 / 1 fixes it.
    static int want = -2;
    if (want == -2) {
      const char* e = getenv("BGP_PS_PAIR");
      want = e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }
    a.Bpad = 8 * ((B + 7) / 8);
    // automatic: where the chain is the bound and the pairs measured faster (bgp_pair_auto_rule, bgp_common.h)
    const bool fits = nblk >= 3 && ncu - 2 * a.Bpad >= std::max(32, B);
    a.pair = (fits && (want == 1 || (want == -1 && bgp_pair_auto_rule(nblk, B)))) ? 1 : 0;
    a.nchain = a.pair ? 2 * a.Bpad : B;
    // P(I) -- the pre-update of block (I, I-1) that the chain waits for -- as ONE task behind a single chain workgroup, in four
    // 64 x 64 quadrants (and Dg(I) in three) behind chain pairs, whose cycle runs THROUGH this task (the
    // two-slice and the mixed variants measured slower and left the library in round 5)
    a.psplit = a.pair ? 4 : 1;
    a.dsplit = a.psplit == 4 ? 3 : 1;
    a.total = bgp_ps_total_tasks(B, nblk, a.psplit, 0);
    // pair mode: the panel solves S(J+2, J) and S(J+3, J) follow pf_block(J) row block by row block (S(J+3, J) feeds the
    // quadrants ahead of the next column's critical solve; streaming fewer measured slower)
    a.ncrit_stream = 3;
  }
  int tile_wgs = std::min(a.total, ncu - a.nchain);
  {
    // critical pool of the tile role: the three tasks at the head of a block column -- S(J+2, J), P(J+2), Dg(J+2) -- get
    // workgroups of their own, one per task of a column.  Measured in round 3: with up to ~10 block columns the chain waits less
    // (n = 1024 x 32: 0.63 -> 0.59 ms, 975 x 50: 0.93 -> 0.84); with more, these left-looking tasks are long and want the
    // look-ahead the single list gives them (n = 2048 x 9: 1.20 -> 1.37 ms with the pool).
    // Chain pairs with more block columns: their 12 critical tasks per column and matrix are SHORT (quadrants that follow their
    // inputs) and numerous -- behind the long bulk solves of one list they start late: a pool of 48 (one matrix) / 96 workgroups
    // (n = 4096 x 1: 1.361 -> 1.330 ms, x 2: 2.022 -> 1.861; 1536 x 4: 0.588 -> 0.545, x 8: 0.666 -> 0.578, x 9: 0.847 -> 0.651)
    // One chain workgroup per matrix, re-measured at the end of round 4: the pool only pays from ~40 matrices on (975 x 50: 0.743 ->
    // 0.725 ms, 896 x 48: 0.543 -> 0.527); below, the single list is 2-5 % faster now (1024 x 32: 0.528 -> 0.518, x 24: 0.518 ->
    // 0.493, 1152 x 24: 0.580 -> 0.555, 975 x 25: 0.525 -> 0.508)
    const int auto_crit = nblk <= 10 ? ((a.pair || B > 32) ? std::min((a.psplit + 1 + a.dsplit + (a.psplit == 4 ? 4 : 0)) * B, tile_wgs / 2) : 0)
                                     : (a.pair && a.psplit == 4 ? std::min(B == 1 ? 48 : 96, tile_wgs / 2) : 0);
    a.ncrit = auto_crit;
    // (at least one workgroup is left for the bulk list whenever it has tasks: critical tasks spin-wait on bulk solves of the
    // previous column)
    if (a.ncrit > tile_wgs - 1) a.ncrit = std::max(0, tile_wgs - 1);
  }
  // build_gram: the caller has NOT built the Gram matrices (plain LML batch: unwarped inputs, c->dh, diagonal additions).  With one
  // chain workgroup per matrix, one ticket list and the default kernel form the tile workers generate them block by block at the
  // head of that list -- the chain starts on block (0, 0) ~10 us into the launch instead of behind a whole Gram kernel and a kernel
  // boundary (n = 1024 x 32: 63 + ~15 us) -- else the Gram kernel goes in front as before.  BGP_PS_GEN = 0 / 1 fixes it.
  a.gen = 0;
  if (build_gram) {
    static int want = -2;
    if (want == -2) {
      const char* e = getenv("BGP_PS_GEN");
      want = e ? (atoi(e) != 0 ? 1 : 0) : -1;
    }
    const bool can = !a.pair && a.ncrit == 0 && a.total > 0 && c->ks.stationary == BGP_MATERN52 && c->ks.form == BGP_FORM_PRODUCT;
    if (can && (want == 1 || (want == -1 && bgp_ps_gen_auto_rule(nblk, B)))) {
      a.gen = 1;
      a.total = bgp_ps_total_tasks(B, nblk, a.psplit, 1);
    } else {
      const int rck = bgp_launch_kbuild(c, B, 0, 0, 1);
      if (rck) return rck;
    }
  }
  a.d = c->d;
  a.X = c->dXeff;
  a.alpha = c->dalpha;
  a.H = c->dh;
  a.y = c->dy;
  tile_wgs = std::min(a.total, ncu - a.nchain);  // (gen adds tasks)
  a.spin_limit = limit;
  a.trace = nullptr;
  {
    static int want = -1;
    if (want < 0) {
      const char* e = getenv("BGP_PS_TRACE");
      want = (e && atoi(e) != 0) ? 1 : 0;
    }
    if (want) {
      const size_t need = (size_t)B * nblk * 8 + (size_t)a.total * 8;
      if (need > c->cap_pstrace) {
        if (c->ps_trace) (void)hipFree(c->ps_trace);
        c->ps_trace = nullptr;
        c->cap_pstrace = 0;
        BGP_HIP(hipMalloc(&c->ps_trace, need * sizeof(unsigned long long)));
        c->cap_pstrace = need;
      }
      BGP_HIP(hipMemsetAsync(c->ps_trace, 0, need * sizeof(unsigned long long), c->stream));
      a.trace = c->ps_trace;
      c->ps_trace_B = B;
      c->ps_trace_nblk = nblk;
      c->ps_trace_total = a.total;
    }
  }
  bgp_launch_ps(c->stream, a, a.nchain + tile_wgs);
  if (!c->ps_resident)
    BGP_HIP(hipMemcpyAsync(c->ps_herr, c->ps_flags + PS_ERROR, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

// LDS-DMA pipelined trailing update and panel solve (bgp_syrk4.hip)
void bgp_launch_syrk4(hipStream_t st, int B8, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                      int K, int jstart, int colmode, int B);
void bgp_launch_trsm4(hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus, int ld, size_t mstride,
                      int ystride, int nblk, int k, int augmented);

void bgp_launch_potrf(bgp_ctx* ctx, hipStream_t st, int B, double* dK, double* dW, double* dyw, double* dacc,
                      double* dlml, int* dstatus, int ld, size_t mstride, int ystride, int k) {
  hipLaunchKernelGGL((potrf_kernel<0, 0, 0>), dim3(B), dim3(PF_THREADS), 0, st, dK, dW, dyw, dacc, dlml, dstatus, ctx->n, ld,
                     mstride, ystride, ctx->nblk, k, PfGen());
}

// Fused LML of a batch at n <= 128: ONE launch (Gram generation + factorisation + forward substitution + LML in the
// walker's workgroup).  dXb / xstride: per-walker warped inputs or the shared training set (xstride == 0 only).
int bgp_launch_lml_small(bgp_ctx* ctx, int off, int B, hipStream_t st) {
  PfGen g;
  g.X = ctx->dXeff;
  g.alpha = ctx->dalpha;
  g.H = ctx->dh + (size_t)off * (ctx->d + 2);
  g.y = ctx->dy;
  g.d = ctx->d;
  bgp_tbegin(ctx, 1, st);
  KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
              hipLaunchKernelGGL((potrf_kernel<1, S, F>), dim3(B), dim3(PF_THREADS), 0, st, (double*)nullptr,
                                 (double*)nullptr, (double*)nullptr, (double*)nullptr, ctx->dlml + off, ctx->dstatus + off,
                                 ctx->n, 128, (size_t)0, 0, 1, 0, g));
  bgp_tend(ctx, st);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

void bgp_launch_syrk4_gen(hipStream_t st, int B8, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                          int K, int jstart, int colmode, int B, const S4Gen& gen, int stationary, int form);

int bgp_launch_cholesky(bgp_ctx* ctx, int B, int augmented) {
  return bgp_launch_cholesky_slice(ctx, 0, B, ctx->stream, augmented);
}

// gen != 0 (LML path): only block column 0 of the kernel matrices has been built (bgp_launch_kbuild_x, full_square = 2); the
// updates of the first panel group generate every other block as they touch it first.
int bgp_launch_cholesky_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int augmented, int gen) {
  // augmented == 0: LML only (matrices npad x npad).  augmented != 0: posterior build on the
  // (2 npad) x (2 npad) augmented matrices [[K, .], [I, 0]] (see bgp_rowblk).
  const int nblk = ctx->nblk, npad = ctx->npad;
  const int ld = augmented ? 2 * npad : npad;
  const size_t mstride = (size_t)ld * ld;
  const int ystride = ld;
  const int B8 = 8 * ((B + 7) / 8);
  double* dK = ctx->dK + (size_t)off * mstride;
  double* dW = ctx->dW + (size_t)off * nblk * (128 * 128);
  double* dyw = ctx->dyw + (size_t)off * ystride;
  double* dacc = ctx->dacc + (size_t)off * 4;
  double* dlml = ctx->dlml + off;
  int* dstatus = ctx->dstatus + off;
  if (!augmented) {
    // LML path: multi-panel trailing updates.  A group
    // of P block columns is factorised with look-ahead column updates only, then everything to its right is
    // updated ONCE with the whole K = 128 P panel (P times fewer passes over the trailing matrix):
    //   potrf(k) trsm(k) | col k+1 (K=128) | potrf(k+1) trsm(k+1) | col k+2 (K=256) | ... | rest (K = 128 P)
    // P block columns per trailing update: 4 from n = 1536 upwards (config C: 16.5 vs 17.1 ms per step against P = 2
    // with the LDS-DMA kernels, whose look-ahead column launches are cheap enough), 2 below (P = 2, 3, 4 are equal
    // within noise at n = 1024); BGP_PANELS fixes it.
    const int P = ctx->panels_auto ? (nblk >= 12 ? 4 : 2) : ctx->panels;
    S4Gen ga{};
    if (gen) {
      const int rcg = bgp_lml_gen_args(ctx, off, &ga);
      if (rcg) return rcg;
      ctx->gen_batches++;
    }
    int k = 0;
    while (k < nblk) {
      const int np = std::min(P, nblk - k);
      for (int j = 0; j < np; j++) {
        bgp_tbegin(ctx, 1, st);
        hipLaunchKernelGGL((potrf_kernel<0, 0, 0>), dim3(B), dim3(PF_THREADS), 0, st, dK, dW, dyw, dacc, dlml, dstatus,
                           ctx->n, ld, mstride, ystride, nblk, k + j, PfGen());
        bgp_tend(ctx, st);
        if (k + j + 1 >= nblk) break;
        bgp_tbegin(ctx, 2, st);
        bgp_launch_trsm4(st, B, dK, dW, dyw, dstatus, ld, mstride, ystride, nblk, k + j, 0);
        bgp_tend(ctx, st);
        if (j + 1 < np) {  // look-ahead: block column k+j+1 with the panels k .. k+j
          bgp_tbegin(ctx, 5, st);
          if (gen && k == 0) {
            bgp_launch_syrk4_gen(st, B8, dK, dstatus, ld, mstride, nblk, k, 128 * (j + 1), k + j + 1, 1, B, ga, ctx->ks.stationary,
                                 ctx->ks.form);
            ctx->gen_launches++;
          } else
            bgp_launch_syrk4(st, B8, dK, dstatus, ld, mstride, nblk, k, 128 * (j + 1), k + j + 1, 1, B);
          bgp_tend(ctx, st);
        }
      }
      const int nt = nblk - (k + np);
      if (nt > 0) {
        bgp_tbegin(ctx, 3, st);
        if (gen && k == 0) {
          bgp_launch_syrk4_gen(st, B8, dK, dstatus, ld, mstride, nblk, k, 128 * np, k + np, 0, B, ga, ctx->ks.stationary, ctx->ks.form);
          ctx->gen_launches++;
        } else
          bgp_launch_syrk4(st, B8, dK, dstatus, ld, mstride, nblk, k, 128 * np, k + np, 0, B);
        bgp_tend(ctx, st);
      }
      k += np;
    }
    BGP_HIP(hipGetLastError());
    return BGP_OK;
  }
  // posterior build: nblk steps on the augmented matrix, the ring kernels in single-panel mode with the active-row
  // remap of bgp_rowblk (nblk active row blocks at every step)
  for (int k = 0; k < nblk; k++) {
    bgp_tbegin(ctx, 1, st);
    hipLaunchKernelGGL((potrf_kernel<0, 0, 0>), dim3(B), dim3(PF_THREADS), 0, st, dK, dW, dyw, dacc, dlml, dstatus, ctx->n,
                       ld, mstride, ystride, nblk, k, PfGen());
    bgp_tend(ctx, st);
    bgp_tbegin(ctx, 2, st);
    bgp_launch_trsm4(st, B, dK, dW, dyw, dstatus, ld, mstride, ystride, nblk, k, 1);
    bgp_tend(ctx, st);
    bgp_tbegin(ctx, 3, st);
    bgp_launch_syrk4(st, B8, dK, dstatus, ld, mstride, nblk, k, 128, 0, 2, B);
    bgp_tend(ctx, st);
  }
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}
