// Internal definitions shared by the libbgp translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bgp.h"

#define BGP_NB 128          // block size of the right-looking Cholesky == tile edge
#define BGP_TILE_LD 129     // LDS leading dimension of a 128x128 tile (odd -> conflict-free columns)
#define BGP_MAX_D 256       // maximum input dimension
#define BGP_MAX_STREAMS 8

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

void bgp_set_error(const char* fmt, ...);
// A call of THIS thread has failed: none of the downloads it enqueued may be unpacked into its caller's buffers later.  Only the
// calling thread's entries go (one context per host thread: another thread's context keeps its staged downloads), and their
// streams stay marked busy -- their copies may still be in flight into the arena -- until they are synchronised or destroyed.
void bgp_xfer_drop_pending();
// The stream is about to be destroyed (and has been synchronised): drop what is left of it without unpacking.
void bgp_xfer_forget(hipStream_t st);

#define BGP_HIP(call)                                                                        \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      bgp_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* clear the sticky error so that later calls are not blamed */  \
      bgp_xfer_drop_pending(); /* no download of this call may be unpacked into the caller's buffers later */ \
      return BGP_ERR_HIP;                                                                    \
    }                                                                                        \
  } while (0)

// ---- host <-> device transfers through pinned staging, and the active wait ----
// Every copy between a CALLER's buffer (pageable: numpy arrays) and the device goes through the library's own pinned
// arena.  An asynchronous copy straight from / to pageable memory makes the runtime lock and unlock the pages around it,
// and the unlocking was measured to trail the call: after a config-E PVRS tell (640 KB of candidates up, Thompson draws
// down) the device stayed "busy" for another 27 ms in five of six processes, which the NEXT tell's first synchronisation
// then paid (41 -> 70 ms per tell; tools/archive/tell_phase_probe.py).  bgp_memcpy_async / bgp_memcpy2d_async take the
// arguments of their HIP namesakes: host -> device packs the rows into the arena and copies from there; device -> host
// lands in the arena and is unpacked into the caller's buffer by bgp_stream_sync of that stream (every entry point
// synchronises before it returns).  Device -> device passes through.
struct BgpXfer {
  struct Block {
    char* p;
    size_t cap, off;
  };
  struct Pending {
    hipStream_t st;
    char* host;
    size_t hpitch;
    const char* stage;
    size_t width, height;
    std::thread::id owner;  // the thread that enqueued the download (a failing call drops ITS OWN downloads only)
  };
  std::vector<Block> blocks;
  std::vector<Pending> pending;
  std::vector<hipStream_t> busy;  // streams with staged host -> device data not yet known to have been consumed
  char* take(size_t bytes);
  void release(hipStream_t st);   // the stream has been synchronised: unpack its downloads, forget its uploads
  void maybe_reset();             // nothing staged is in flight any more: the arena starts over
};
// ONE arena per process behind a mutex (bgp_api.hip): the bookkeeping is keyed by STREAM, so whichever thread synchronises a
// stream unpacks that stream's downloads -- a thread-local arena left them behind when another thread than the enqueuing one
// waited.  Transfers above BGP_XFER_DIRECT bytes (predictive covariances, whole kernel matrices, debugging downloads) do not
// go through it: they are copied synchronously, straight between the caller's buffer and the device (nothing asynchronous
// is left behind either way), and an idle arena above BGP_XFER_KEEP bytes is given back.
#define BGP_XFER_DIRECT ((size_t)8 << 20)
#define BGP_XFER_KEEP ((size_t)64 << 20)
void bgp_xfer_release(hipStream_t st);
hipError_t bgp_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                              hipMemcpyKind kind, hipStream_t st);
static inline hipError_t bgp_memcpy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
  return bgp_memcpy2d_async(dst, bytes, src, bytes, bytes, 1, kind, st);
}

// Wait for a stream the way the sampler's inner loop needs it: ACTIVELY.  hipStreamSynchronize's default wait parks
// the thread, and for the 0.5-1 ms device calls of the small-batch regime (config E: 26 calls of 50 proposals at
// n ~ 1000 per tell) the wake-up was measured BISTABLE on MI355X -- the same tell took 25 ms or 52 ms of MCMC, +1 ms per
// call, run to run and tell to tell.  Polling hipStreamQuery from the calling thread (which has nothing else to do)
// removes that.  The first 2 ms are a tight poll (the calls this exists for last 0.1-2 ms); a longer call (n = 4096 batches,
// the 10 112^2 covariance of sample_y) is polled every ~20 us with the core handed back in between, and after 200 ms the
// runtime's blocking wait takes over.  BGP_WAIT=block selects the runtime's wait from the start (A/B measurements,
// oversubscribed hosts).
int bgp_wait_spins();
static inline void bgp_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
static inline hipError_t bgp_stream_sync(hipStream_t st) {
  hipError_t e = hipErrorNotReady;
  if (bgp_wait_spins()) {
    const auto t0 = std::chrono::steady_clock::now();
    bool slow = false;
    for (unsigned it = 0;; it++) {
      e = hipStreamQuery(st);
      if (e != hipErrorNotReady) break;
      (void)hipGetLastError();  // (hipErrorNotReady is recorded as the thread's last error)
      if (slow) {
        std::this_thread::sleep_for(std::chrono::microseconds(20));
        if ((it & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
      } else {
        if ((it & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) slow = true;
        bgp_cpu_relax();
      }
    }
  }
  if (e == hipErrorNotReady) e = hipStreamSynchronize(st);
  if (e == hipSuccess) bgp_xfer_release(st);
  return e;
}

struct bgp_ctx {
  int device = 0;
  int n = 0, d = 0, npad = 0, nblk = 0;
  int max_batch = 0;
  bgp_kernel_spec ks{};
  hipStream_t stream = nullptr;
  // walker groups: the batch of an LML call is split over nstreams HIP streams so that the
  // latency-bound potrf / small trsm launches of one group overlap the MFMA-bound syrk of another
  int panels = 2;        // right-looking LML path: block columns per trailing update (K = 128 * panels; env BGP_PANELS)
  int panels_auto = 1;   // no BGP_PANELS in the environment: chosen per problem size (bgp_chol.hip)
  int nstreams = 1;
  int streams_auto = 1;  // choose the group count per call from the batch size (see bgp_ctx_create)
  hipStream_t gstream[BGP_MAX_STREAMS] = {nullptr};
  hipEvent_t ev_ready = nullptr;
  hipEvent_t ev_done[BGP_MAX_STREAMS] = {nullptr};
  // resident training set
  double* dX = nullptr;      // n*d  (original inputs)
  double* dXeff = nullptr;   // what the kernels read: dX, or dXw1 when a context-level warp is set
  double* dXw1 = nullptr;    // training inputs through the context-level Beta-CDF warp
  double* dXwB = nullptr;    // per-walker warped inputs of a warped LML batch (max_batch * n * d)
  double* dwarp = nullptr;   // context-level warp parameters (2d, log space)
  double* dwarpB = nullptr;  // per-walker warp parameters (max_batch * 2d)
  size_t cap_xwb = 0;
  int has_warp = 0;
  double* dXs = nullptr;     // scaled inputs of the current batch, k-major: max_batch * dpad * npad (bgp_kbuild.hip)
  size_t cap_xs = 0;
  double* dy = nullptr;      // npad (zero padded)
  double* dalpha = nullptr;  // npad
  size_t cap_n = 0;          // capacity (rows) of the three buffers above
  // per-batch workspace
  double* dK = nullptr;      // max_batch * npad*npad   working matrices (become L in place)
  double* dW = nullptr;      // max_batch * nblk * 128*128  inverses of the diagonal blocks
  double* dyw = nullptr;     // max_batch * npad        working rhs (becomes z = L^-1 y)
  double* dacc = nullptr;    // max_batch * 4           {logdet, z^T z, -, -}
  double* dh = nullptr;      // max_batch * (d+2)       canonical hyper-parameters
  double* dlml = nullptr;    // max_batch
  int* dstatus = nullptr;    // max_batch
  size_t cap_mat = 0;        // capacity in doubles of dK
  size_t cap_w = 0;
  size_t cap_yw = 0;
  // resident posteriors (K^-1 full symmetric npad x npad each, alpha = K^-1 y)
  double* dKinv = nullptr;
  double* dalpha_sol = nullptr;
  size_t cap_kinv = 0;
  size_t cap_alpha = 0;
  // resident posterior state
  int post_B = 0;            // number of resident posteriors (0 = none)
  std::vector<double> post_h;
  // scratch for predict / pvrs (grown on demand)
  double* dscratch = nullptr;
  double* drowpart = nullptr;  // column-tile partials of the predictive-variance row dots
  size_t cap_rowpart = 0;
  size_t cap_scratch = 0;
  // asynchronous LML batch (bgp_lml_batch_submit / _wait): pinned result buffers and the pending batch size
  double* hstage = nullptr;  // pinned staging of the training set (bgp_ctx_update_data)
  size_t cap_stage = 0;
  double* hh = nullptr;      // pinned copy of the submitted hyper-parameter block
  double* hwarp = nullptr;   // pinned copy of the submitted per-walker warp parameters (max_batch * 2d)
  double* hlml = nullptr;
  int* hstatus = nullptr;
  size_t cap_pinned = 0;
  int pending_B = 0;
  bgp_ctx* child = nullptr;  // cached workspace of bgp_sample_y (covariance Cholesky)
  // launch-free factorisation of small batches (ps_kernel): flag block, pinned error word, trace buffer
  int ncu = 0;               // CUs of the device (the launch-free kernel takes one workgroup per CU)
  int persist = -1;          // env BGP_PERSIST: 0 never, 1 whenever possible, -1 (unset) automatic by batch size
  unsigned* ps_flags = nullptr;
  size_t cap_psflags = 0;
  unsigned* ps_herr = nullptr;  // pinned: error word of the last persistent call
  int ps_inflight = 0;          // a persistent call is on the stream (its error word is checked behind the sync)
  int ps_disabled = 0;          // a persistent call timed out: multi-launch path (see bgp_ps_note_timeout)
  int ps_cooldown = 0;          // eligible calls left on the multi-launch path before the launch-free one is tried again
  long long gen_batches = 0, gen_launches = 0;  // bgp_lml_gen_stats
  long long ps_calls = 0;       // launch-free calls enqueued by this context (bgp_persist_stats)
  long long ps_timeouts = 0;    // ... of which timed out and were redone by launches
  int pending_warped = 0;       // the pending batch carries per-walker warps (redo path of bgp_lml_batch_wait)
  struct bgp_mcmc_state* mcmc = nullptr;  // an open device-resident sampler run (bgp_mcmc_begin .. bgp_mcmc_end; bgp_mcmc.hip)
  int ps_forbid = 0;            // the device-resident sampler redoes a run after a time-out: launches only, on every rank
  int ps_resident = 0;          // the device-resident sampler is enqueuing: no per-call copy of the error word (its kernels read it)
  unsigned long long* ps_trace = nullptr;  // BGP_PS_TRACE=1: device buffer of in-kernel time stamps (bgp_debug_ps_trace)
  size_t cap_pstrace = 0;
  int ps_trace_B = 0, ps_trace_nblk = 0, ps_trace_total = 0;
  // timing
  int timing = 0;
  double t_ms[6] = {0, 0, 0, 0, 0, 0};  // K-build, potrf, trsm, syrk (all), whole call, look-ahead column launches of syrk
  int t_cnt[6] = {0, 0, 0, 0, 0, 0};
  std::vector<hipEvent_t> ev;  // (start, stop) pairs of the launches of the current call
  std::vector<int> evcat;
};

// The workspace belongs to a batch submitted with bgp_lml_batch_submit until bgp_lml_batch_wait has collected it.
#define BGP_REQUIRE_IDLE(c, who)                                                                     \
  do {                                                                                              \
    if ((c) && (c)->pending_B != 0) {                                                               \
      bgp_set_error(who ": a submitted batch is still pending (call bgp_lml_batch_wait)");          \
      return BGP_ERR_STATE;                                                                         \
    }                                                                                               \
  } while (0)

// Per-launch HIP-event timing on the context's stream (only when ctx->timing != 0).
// Categories: 0 K-build, 1 potrf, 2 trsm, 3 syrk (bulk update of a panel group), 5 syrk look-ahead column launch (counted
// under 3 as well; bgp_last_timing_columns reports the split).
static inline void bgp_tbegin(bgp_ctx* c, int cat, hipStream_t st = nullptr) {
  if (!c->timing) return;
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  (void)hipEventRecord(a, st ? st : c->stream);
  c->ev.push_back(a);
  c->ev.push_back(b);
  c->evcat.push_back(cat);
}
static inline void bgp_tend(bgp_ctx* c, hipStream_t st = nullptr) {
  if (!c->timing) return;
  (void)hipEventRecord(c->ev.back(), st ? st : c->stream);
}
static inline void bgp_tcollect(bgp_ctx* c) {
  if (!c->timing) return;
  (void)hipStreamSynchronize(c->stream);
  for (size_t i = 0; i < c->evcat.size(); i++) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev[2 * i], c->ev[2 * i + 1]);
    c->t_ms[c->evcat[i]] += ms;
    c->t_cnt[c->evcat[i]] += 1;
    if (c->evcat[i] == 5) {  // a look-ahead column launch is a trailing-update launch too
      c->t_ms[3] += ms;
      c->t_cnt[3] += 1;
    }
    (void)hipEventDestroy(c->ev[2 * i]);
    (void)hipEventDestroy(c->ev[2 * i + 1]);
  }
  c->ev.clear();
  c->evcat.clear();
}

#define BGP_MAX_DEVICES 64
// ---- launch-free factorisation of small batches (bgp_chol.hip: ps_chain_kernel, bgp_syrk4.hip: ps_tile_kernel) ----
// Flag block of one persistent factorisation (32-bit words, zeroed by a memset node in front of every call):
//   [PS_TICKET + 2 + 8 pool + x]  next task of the tile kernel's list (pool, XCD x)      [PS_ERROR]  != 0: a wait timed out: everybody leaves
//   wready[b * nblk + J]          1 when potrf(J) of matrix b has published L_JJ, W_JJ, z_J (or the matrix has failed)
//   diagrdy[b * nblk + I]         1 when the diagonal block (I, I) carries the panels 0 .. I-2 (the chain applies panel I-1)
//   xready[(b * nblk + I) * nblk + J]   1 when the panel block X_IJ is final, I > J
//   subrdy[b * nblk + I]          1 when block (I, I-1) carries the panels 0 .. I-2 (the chain solves it)
//   wrow[b * nblk + J]            pair mode: number of 16-row blocks of W_JJ = L_JJ^-1 that are complete in memory (0 .. 7; the
//                                 eighth goes out with wready[J])
//   xcol[(b * nblk + I) * 3 + d]  pair mode: number of 16-column blocks of the panel block X_{I, I-1-d} that are complete in memory
//                                 (0 .. 8), d = 0: the chain helper's block, d = 1, 2: the streamed panel solves S(I, I-2), S(I, I-3);
//                                 the quadrant pre-updates that they feed consume them column block by column block
//   s2rdy[b * nblk + I]           counts the quadrants of block (I, I-2) that carry the panels 0 .. I-3 (the solve S(I, I-2) waits for 4)
#define PS_TICKET 0
#define PS_ERROR 1
#define PS_HDR 32
#define PS_XCOL(B, nblk) (PS_HDR + (size_t)(B) * (nblk) * (4 + (nblk)))  // (behind wrow; three words per (b, I))
#define PS_S2RDY(B, nblk) (PS_HDR + (size_t)(B) * (nblk) * (7 + (nblk)))
struct PsArgs {
  double* K;          // B working matrices (ld x ld, row-major), become L in place
  double* W;          // B x nblk inverses of the diagonal blocks
  double* yw;         // B working right-hand sides (become z)
  double* acc;        // B x 4 running {log det, z^T z}
  double* lml;
  int* status;
  unsigned* flags;    // the block above
  int n, ld, nblk, B, ystride;
  size_t mstride;
  int total;          // tasks of the tile kernel (0: two block columns, the chain does everything)
  int ncrit;          // workgroups of the tile kernel's critical pool (0: one list)
  int pair;           // 1: TWO chain workgroups per matrix that alternate over the block columns (bgp_pf.h, ps_chain_role)
  int Bpad;           // pair mode: chain workgroup p of matrix b is block p * Bpad + b (Bpad = B rounded up to 8: same XCD)
  int nchain;         // chain workgroups at the head of the grid (B, or 2 * Bpad)
  int psplit;         // parts a pre-update task P(I) is dealt out in (1, or 4 quadrants behind chain pairs): subrdy[I] counts to psplit
  int dsplit;         // parts of a diagonal block's task Dg(I) (3 quadrants with psplit == 4, else 1): diagrdy[I] counts to dsplit
  int ncrit_stream;   // pair mode: panel solves S(I, J) with I <= J + ncrit_stream follow pf_block(J) row block by row block
  unsigned long long spin_limit;  // wall_clock64 ticks (100 MHz) a single wait may last before the call is abandoned
  unsigned long long* trace;      // debugging (BGP_PS_TRACE=1): wall-clock stamps, chain: 8 per (b, J), tile: 8 per task
  // gen = 1: the Gram matrices are generated INSIDE this launch -- the first B nblk (nblk + 1) / 2 tickets of the tile list are one
  // 128 x 128 block each (kb_gram_tile512; block (I, 0) also sets up row block I of the right-hand side), genrdy[b][I][J] tells
  // its consumer -- instead of by a Gram kernel in front of it (Matern-5/2 product form, one chain workgroup per matrix, one list)
  int gen, d;
  const double* X;      // n x d training inputs
  const double* alpha;  // n diagonal additions
  const double* H;      // B x (d + 2) canonical hyper-parameters
  const double* y;      // n
};
//   genrdy[(b * nblk + I) * nblk + J]   gen = 1: block (I, J) of matrix b has been generated (behind s2rdy)
#define PS_GEN(B, nblk) (PS_HDR + (size_t)(B) * (nblk) * (8 + (nblk)))
// Shapes at which the tile workers generate the Gram blocks inside the launch-free kernel (PsArgs::gen) instead of a Gram kernel in
// front of it: where the chain is the bound and the tile side has the slack to take the extra work (measured: bgp_chol.hip)
// (tools/gen_probe.py, ms per LML call without -> with: 1024 x 24 0.488 -> 0.464, x 16 0.476 -> 0.456, x 8 0.460 -> 0.449; 768 x 32
// 0.383 -> 0.362; 1536 x 16 0.729 -> 0.719; 2048 x 8 0.927 -> 0.884; 512 x 32 0.260 -> 0.250; but 1024 x 32 0.522 -> 0.527, 2048 x 16
// 1.308 -> 1.357 and 3072 x 8 1.921 -> 1.936: there the tile side is the bound already and the blocks are extra work for it)
static inline bool bgp_ps_gen_auto_rule(int nblk, int B) { return nblk >= 4 && nblk <= 16 && B <= 32 && B * nblk <= 192; }
static inline size_t ps_flag_words(int B, int nblk) { return PS_HDR + (size_t)B * nblk * (8 + 2 * (size_t)nblk); }
// Batch sizes at which the launch-free factorisation wins over the multi-launch schedule (tools/persist_probe.py on MI355X,
// DESIGN.md section 4; wall time per LML call, launch schedule / launch-free, by n and number of matrices):
//   n =  768: 8: 0.91, 32: 1.07;   896: 16: 1.02, 48: 1.11
//   n = 1024: 1: 0.93, 4: 0.95, 8: 0.98, 16: 1.06, 24: 1.12, 32: 1.19, 48: 1.09, 64: 0.97;  975 x 50 (config E): 1.10
//   n = 1280: 8: 1.04, 32: 1.15, 48: 0.96;   1536: 1: 1.01, 4: 1.06, 16: 1.29, 32: 1.06, 48: 0.92
//   n = 2048: 1: 1.02, 4: 1.12, 9: 1.40, 16: 1.22, 24: 1.07, 32: 0.99, 48: 0.88;   3072: 1: 1.08, 8: 1.25, 16: 1.06, 24: 0.95
//   n = 4096: 1: 1.15, 2: 1.12, 4: 1.08, 8: 1.13, 16: 0.98;  one 10 112 x 10 112 covariance (sample_y): 1.15;  n <= 640: 0.92-0.97
// So: at least 6 block columns, matrices x block columns <= 400 (beyond that the tile side is the bound and the launch
// schedule's kernels are the better tile workers), and >= 100 unless there are at least 12 block columns.
// Chain PAIRS (two chain workgroups per matrix that alternate over the block columns; the critical pre-updates in quadrants that
// follow the chain's and the streamed solves' blocks column block by column block, on a critical pool of their own: bgp_pf.h,
// bgp_chol.hip) win where the chain is the bound -- few matrices.  Wall ms per LML call, best of {launches, launch-free} ->
// pairs (tools/persist_probe.py, MI355X, round 4):
//   n =  384 x 1: 0.173 -> 0.159;   512 x 1 / 4 / 8: 0.220 / 0.225 / 0.223 -> 0.197 / 0.203 / 0.218;   768 x 1 / 16: 0.318 / 0.356 -> 0.271 / 0.317
//   n = 1024 x 1 / 8 / 16 (32: 0.528 -> 0.630): 0.413 / 0.447 / 0.475 -> 0.349 / 0.414 / 0.430;   1280 x 12: 0.601 -> 0.554;   1152 x 14: 0.543 -> 0.500
//   n = 1408 x 9 / 12 (16: 0.663 -> 0.671): 0.639 / 0.652 -> 0.611 / 0.624;   1536 x 1 / 4 / 8 / 9 / 12 (16: 0.763 -> 0.789): 0.650 / 0.664 / 0.675 / 0.712 / 0.729 -> 0.506 / 0.546 / 0.579 / 0.659 / 0.722
//   n = 1792 x 4 / 8 (12: 0.919 -> 0.953): 0.765 / 0.789 -> 0.623 / 0.662;   2048 x 1 / 2 / 4 / 8 (12: 1.207 -> 1.284, 16: 1.305 -> 1.503): 0.861 / 0.865 / 0.889 / 0.925 -> 0.653 / 0.673 / 0.746 / 0.839
//   n = 2560 x 4 / 6 (8: 1.275 -> 1.381): 1.154 / 1.234 -> 1.081 / 1.230;   3072 x 1 / 2 / 3 / 4 (6: 1.798 -> 1.908): 1.326 / 1.341 / 1.473 / 1.639 -> 1.023 / 1.195 / 1.399 / 1.609
//   n = 3584 x 2 / 3 (4: 2.191 -> 2.179): 1.636 / 1.904 -> 1.504 / 1.817;   4096 x 1 / 2 / 3 (4: 2.831 -> 2.969): 1.730 / 2.112 / 2.472 -> 1.327 / 1.862 / 2.398
//   n = 4992 x 1 / 5120 x 1 (x 2: 2.906 -> 2.893) / 6144 x 1: 2.113 / 2.190 / 2.629 -> 1.601 / 1.641 / 2.326;   not 7168 x 1 (3.195 -> 3.189), 8192 x 1 (4.201 -> 4.406), 10 112 x 1
static inline bool bgp_pair_auto_rule(int nblk, int nb) {
  if (nblk < 3) return false;
  if (nblk <= 10) return nb <= 16 && nb * nblk * nblk <= 1200;
  if (nblk <= 12) return nb <= 12;
  return nb <= 8 && nb * nblk * nblk <= (nb <= 3 ? 3100 : 2400);  // (the tile side's work goes with matrices x block columns^2)
}
static inline bool bgp_persist_auto_rule(int nblk, int nb) {
  return bgp_pair_auto_rule(nblk, nb) || (nblk >= 6 && nb * nblk <= 400 && (nblk >= 12 || nb * nblk >= 100));
}
// A launch-free call timed out (a wait outlasted BGP_PS_TIMEOUT_MS: its workgroups were not co-resident -- another context,
// process or RCCL kernel held CUs -- or the device was oversubscribed).  One transient event must not cost the context
// the path for life: the next BGP_PS_COOLDOWN eligible calls go by launches, then the launch-free path is tried again;
// after the third time-out it stays off (bgp_set_persist(ctx, 1) re-arms it).  bgp_persist_stats reports the counts.
int bgp_ps_cooldown_calls();  // BGP_PS_COOLDOWN, default 256 (bgp_api.hip)
static inline void bgp_ps_note_timeout(bgp_ctx* c, const char* what) {
  c->ps_timeouts++;
  c->ps_disabled = 1;
  c->ps_cooldown = c->ps_timeouts >= 3 ? 0 : bgp_ps_cooldown_calls() + 1;  // (+ 1: the redo of this very batch counts one)
  if (c->ps_timeouts <= 3)
    fprintf(stderr, "libbgp: warning: the launch-free factorisation timed out (a wait outlasted BGP_PS_TIMEOUT_MS); %s on the "
                    "multi-launch path, which this context keeps %s (time-out %lld of this context; bgp_persist_stats)\n", what,
            c->ps_cooldown ? "for its next eligible calls (BGP_PS_COOLDOWN, 256)" : "from now on", c->ps_timeouts);
}
// (call only when the batch is otherwise eligible: the cool-down counts eligible calls)
static inline bool bgp_ps_allowed(bgp_ctx* c) {
  if (!c->ps_disabled) return true;
  if (c->ps_cooldown > 0 && --c->ps_cooldown == 0) c->ps_disabled = 0;
  return false;
}
int bgp_launch_cholesky_persist(bgp_ctx* ctx, int B, int build_gram);
int bgp_lml_redo_if_abandoned(bgp_ctx* ctx, int B);
int bgp_lml_enqueue_dev(bgp_ctx* ctx, int nb, int warped);  // bgp_api.hip: Gram build + factorisation + LML of c->dh[0 .. nb), on the device only
int bgp_ensure_warp_buffers(bgp_ctx* ctx);                  // bgp_api.hip: the per-walker warp buffers of a warped LML batch exist
void bgp_mcmc_abandon(bgp_ctx* ctx);  // bgp_mcmc.hip: drop an open sampler run (context teardown, failed calls)
// the launch-free call of a batch whose results are discarded anyway: forget it (no time-out is counted, nothing is redone)
static inline void bgp_ps_clear_inflight(bgp_ctx* c) {
  c->ps_inflight = 0;
  if (c->ps_herr) *c->ps_herr = 0;
}
int bgp_persist_fits(bgp_ctx* ctx, int B);
int bgp_ps_ensure_flags(bgp_ctx* ctx, int B);

int bgp_ensure_scratch(bgp_ctx* ctx, size_t doubles);
void bgp_free_child(bgp_ctx* ctx);
// make the matrix workspace at least `doubles` large (and the per-item side buffers consistent)
int bgp_grow_workspace(bgp_ctx* ctx, size_t doubles);
// posterior build on the augmented matrices; use_alpha == 0 drops alpha_diag (PVRS quirk).  Kgram != nullptr: the B kernel
// matrices come from the host (n x n each, bgp_gram.hip) instead of the device Gram build, h is not read
int bgp_posterior_build(bgp_ctx* c, int B, const double* h, int use_alpha, double* L, double* alpha, double* K_inv,
                        double* lml, int* status, const double* Kgram = nullptr);
// host kernel matrices -> working matrices (bgp_gram.hip)
int bgp_gram_load(bgp_ctx* c, int nb, const double* K, int augmented, int use_alpha);

// ---- kernels launched across translation units ----
// K-build: lower-triangular tiles of the jittered Gram matrix of walker b into dK[b] (npad x npad).
int bgp_launch_kbuild(bgp_ctx* ctx, int B, int full_square, int augmented, int use_alpha);
// same for the slice [off, off+B) of the current batch on an explicit stream
int bgp_launch_kbuild_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented,
                            int use_alpha);
// per-walker inputs: dXb + b * xstride (xstride == 0: shared)
int bgp_launch_kbuild_x(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented, int use_alpha,
                        const double* dXb, size_t xstride);
// Cross kernel matrix k(Xq, X_train) for hyper-vector index b: out is m x ldo row-major (device).
int bgp_launch_kcross(bgp_ctx* ctx, const double* dh_b, int m, const double* dXq, int nx, const double* dXt,
                      double* dout, int ldo, int symmetric_diag_fix);
// the same for nb hyper-vectors dH (nb x (d+2)) into dout + b * ostride
int bgp_launch_kcross_batch(bgp_ctx* ctx, int nb, const double* dH, int m, const double* dXq, int nx, const double* dXt,
                            double* dout, int ldo, size_t ostride);
int bgp_launch_kcross_matvec(bgp_ctx* ctx, int nb, const double* dH, int m, const double* dXq, int nx, const double* dXt,
                             double* dout, int ldo, size_t ostride, const double* vec, size_t svec, double* dpart);
// Blocked Cholesky of the B matrices in dK (in place) + forward substitution + LML.
// Beta-CDF warp of n x d inputs for B parameter sets (bgp_warp.hip)
int bgp_launch_warp(bgp_ctx* c, hipStream_t st, const double* dX, const double* dW, double* dout, int n, int B,
                    size_t ostride);
int bgp_launch_cholesky(bgp_ctx* ctx, int B, int augmented);
// n <= 128: K-build + factorisation + LML of the slice [off, off+B) in ONE launch (bgp_chol.hip, potrf_kernel<1,..>)
int bgp_launch_lml_small(bgp_ctx* ctx, int off, int B, hipStream_t st);
int bgp_launch_cholesky_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int augmented, int gen = 0);
struct S4Gen {           // what the trailing update's Gram generator reads (S4GenF, bgp_s4.h)
  const double* Xs;     // scaled inputs, k-major: dpad x npad doubles per matrix slot
  const double* H;      // canonical hyper-parameters, d + 2 per matrix
  const double* alpha;  // per-point jitter added to the diagonal (nullable)
  int n, d, dpad, npad;
};
int bgp_lml_gen_eligible(const bgp_ctx* ctx, int B);
int bgp_lml_gen_args(const bgp_ctx* ctx, int off, S4Gen* out);
