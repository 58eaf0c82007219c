// RCCL behind the C-ABI: the one real exchange of the multi-GPU path -- the final all-gather of the
// posterior samples (and, in the exact single-ensemble mode, the B log-probabilities per half-step) -- without any
// PyTorch in the product path.  One communicator per process (one process per GPU), collectives on the
// communicator's own HIP stream over device-resident staging buffers that are grown on demand and reused.
// librccl is loaded lazily with dlopen: libbgp.so has no link-time dependency on it and single-GPU use never
// touches it.  The rendezvous (rank 0's ncclUniqueId to every rank) is the caller's: bayes-skopt_amd/distributed.py
// hands the 128 bytes over through a per-job file in a per-user directory on a single node (every rank then reports
// "have it" / "failed" before anybody enters ncclCommInitRank), or over a TCP socket on MASTER_ADDR:(MASTER_PORT + 1)
// across nodes.
// Device-resident exchange: bgp_lml_batch_wait_allgather gathers the log-likelihoods of a submitted batch straight
// out of every rank's context (no host staging on the send side): the per-half-step exchange of the exact
// single-ensemble sharding.
#include "bgp_common.h"
#include "bgp_mcmc.h"

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdlib>
#include <map>
#include <mutex>
#include <rccl/rccl.h>
#include <unistd.h>

namespace {
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                         // optional (bounded waits)
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;  // optional
};
RcclApi g_rccl;

int load_rccl() {
  if (g_rccl.handle) return BGP_OK;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    bgp_set_error("bgp_comm: cannot load librccl.so (%s)", dlerror());
    return BGP_ERR_STATE;
  }
#define BGP_SYM(field, name)                                          \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name)); \
  if (!g_rccl.field) {                                                \
    bgp_set_error("bgp_comm: librccl.so lacks %s", name);             \
    dlclose(h);                                                       \
    return BGP_ERR_STATE;                                             \
  }
  BGP_SYM(GetUniqueId, "ncclGetUniqueId")
  BGP_SYM(CommInitRank, "ncclCommInitRank")
  BGP_SYM(CommDestroy, "ncclCommDestroy")
  BGP_SYM(CommCount, "ncclCommCount")
  BGP_SYM(AllGather, "ncclAllGather")
  BGP_SYM(AllReduce, "ncclAllReduce")
  BGP_SYM(Broadcast, "ncclBroadcast")
  BGP_SYM(GetErrorString, "ncclGetErrorString")
#undef BGP_SYM
  g_rccl.CommAbort = reinterpret_cast<decltype(g_rccl.CommAbort)>(dlsym(h, "ncclCommAbort"));
  g_rccl.CommGetAsyncError = reinterpret_cast<decltype(g_rccl.CommGetAsyncError)>(dlsym(h, "ncclCommGetAsyncError"));
  g_rccl.handle = h;
  return BGP_OK;
}
}  // namespace

// Loop-back group: `world` communicators of ONE process on ONE device (one per host thread, each beside its own context) that
// exchange through device memory they all see.  It exists for the tests of the sharded sampler's row logic on a single GPU
// (two "ranks" = two contexts driven by two threads); only the in-stream gather of bgp_comm_enqueue_lml_gather is served.
// Per half-step: every rank copies its slot into the shared buffer of the round's parity on ITS stream and records its event;
// the threads meet (host barrier: an event can only be waited for once it has been recorded); every rank makes its stream
// wait for all events and copies the shared buffer into its own receive buffer.  Two parities suffice: a rank's write of round
// g + 2 sits behind its wait for the peers' events of round g + 1, which sit behind the peers' reads of round g.
struct LoopGroup {
  std::mutex mu;
  std::condition_variable cv;
  int world = 0, members = 0, arrived = 0;
  int aborted = 0;  // bgp_comm_abort by a member: enqueued work may never complete
  int left = 0;     // a member has been destroyed: nobody can MEET any more (what is enqueued still completes)
  unsigned long long generation = 0;
  double* dshared[2] = {nullptr, nullptr};  // 2 x world x LOOP_SLOT doubles
  std::vector<hipEvent_t> ev[2];            // per parity: one event per rank
  std::vector<unsigned long long> round;    // per rank: rounds enqueued
};
#define LOOP_SLOT (4096 + 1)
static std::mutex g_loop_mu;
static std::map<long long, LoopGroup*> g_loops;

struct bgp_comm {
  LoopGroup* loop = nullptr;  // loop-back communicator (no RCCL behind it)
  long long loop_key = 0;
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* dsend = nullptr;  // device-resident staging, grown on demand
  double* drecv = nullptr;
  size_t cap_send = 0, cap_recv = 0;
  double* hrecv = nullptr;  // pinned landing buffer of the device-resident gathers
  size_t cap_hrecv = 0;
  hipEvent_t ev_ctx = nullptr;  // the communicator's stream waits for a context's stream through this event
  int aborted = 0;              // the communicator was aborted (a collective outlasted its bound / an asynchronous error)
};

// Every collective is waited for with a BOUND.  RCCL itself has no time-out: a rank that died (or returned before the
// collective) leaves its peers in the kernel for ever.  The host polls the communicator's stream; an asynchronous error
// reported by RCCL, or BGP_COMM_TIMEOUT_S seconds (default 300) without completion, aborts the communicator
// (ncclCommAbort) and the call returns BGP_ERR_COMM: the process fails instead of hanging, and so do its peers.
static double comm_timeout_s() {
  static double v = 0.0;
  if (v == 0.0) {
    const char* e = getenv("BGP_COMM_TIMEOUT_S");
    v = (e && atof(e) > 0.0) ? atof(e) : 300.0;
  }
  return v;
}

static int comm_sync(bgp_comm* c, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned it = 0;; it++) {
    const hipError_t e = hipStreamQuery(c->stream);
    if (e == hipSuccess) {
      bgp_xfer_release(c->stream);
      return BGP_OK;
    }
    (void)hipGetLastError();
    if (e != hipErrorNotReady) {
      bgp_set_error("%s: the communicator's stream failed: %s", what, hipGetErrorString(e));
      bgp_xfer_drop_pending();
      return BGP_ERR_HIP;
    }
    if ((it & 255) != 255) {
      bgp_cpu_relax();
      continue;
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ncclResult_t ar = ncclSuccess;
    const bool have = g_rccl.CommGetAsyncError && c->comm && g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess;
    const bool bad = have && ar != ncclSuccess && ar != ncclInProgress;
    if (bad || el > comm_timeout_s()) {
      if (bad)
        bgp_set_error("%s: RCCL reports an asynchronous error (%s); communicator aborted", what, g_rccl.GetErrorString(ar));
      else
        bgp_set_error("%s: the collective did not complete within %.0f s (BGP_COMM_TIMEOUT_S): a peer has died or left; "
                      "communicator aborted", what, comm_timeout_s());
      if (g_rccl.CommAbort && c->comm) {
        (void)g_rccl.CommAbort(c->comm);
        c->comm = nullptr;
      }
      c->aborted = 1;
      bgp_xfer_drop_pending();
      return BGP_ERR_COMM;
    }
    if (el > 0.2) usleep(200);  // a long wait: stop burning the core
  }
}

// The result of a collective on its way to the caller's buffer.  Transfers above BGP_XFER_DIRECT bytes take bgp_memcpy2d_async's
// synchronous route, which waits for the stream with the runtime's UNBOUNDED wait -- behind a collective that is a wait for the
// peers: the bounded wait (time-out, asynchronous-error check, abort) comes first, the copy then finds an idle stream.
static int comm_sync(bgp_comm* c, const char* what);
static int comm_download(bgp_comm* c, double* host, size_t count, const char* what) {
  if (count * sizeof(double) > BGP_XFER_DIRECT) {
    const int rc = comm_sync(c, what);
    if (rc) return rc;
  }
  BGP_HIP(bgp_memcpy_async(host, c->drecv, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  return comm_sync(c, what);
}

#define BGP_COMM_LIVE(c, who)                                                                   \
  do {                                                                                          \
    if ((c)->aborted || !(c)->comm) {                                                           \
      bgp_set_error(who ": the communicator was aborted by an earlier failed collective");      \
      return BGP_ERR_COMM;                                                                      \
    }                                                                                           \
  } while (0)

#define BGP_COMM_LIVE_OR_LOOP(c, who)                                                           \
  do {                                                                                          \
    if ((c)->aborted || (!(c)->comm && !(c)->loop)) {                                           \
      bgp_set_error(who ": the communicator was aborted by an earlier failed collective");      \
      return BGP_ERR_COMM;                                                                      \
    }                                                                                           \
  } while (0)

#define BGP_NCCL(call)                                                                          \
  do {                                                                                          \
    ncclResult_t r__ = (call);                                                                  \
    if (r__ != ncclSuccess) {                                                                   \
      bgp_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r__), __FILE__, __LINE__); \
      return BGP_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)

static int comm_reserve(bgp_comm* c, size_t nsend, size_t nrecv) {
  if (nsend > c->cap_send) {
    if (c->dsend) (void)hipFree(c->dsend);
    c->dsend = nullptr;
    c->cap_send = 0;
    BGP_HIP(hipMalloc(&c->dsend, nsend * sizeof(double)));
    c->cap_send = nsend;
  }
  if (nrecv > c->cap_recv) {
    if (c->drecv) (void)hipFree(c->drecv);
    c->drecv = nullptr;
    c->cap_recv = 0;
    BGP_HIP(hipMalloc(&c->drecv, nrecv * sizeof(double)));
    c->cap_recv = nrecv;
  }
  return BGP_OK;
}

extern "C" int bgp_comm_available(void) { return load_rccl() == BGP_OK ? 1 : 0; }

// RCCL announces itself ("RCCL version : ...", five lines) on STDOUT when its first communicator is set up.  The job's
// standard output belongs to the caller (bench.py prints exactly one JSON line there): while RCCL initialises, file
// descriptor 1 points at standard error, C-level buffers flushed on both sides of the switch.
struct StdoutToStderr {
  int saved = -1;
  StdoutToStderr() {
    fflush(stdout);
    saved = dup(1);
    if (saved >= 0 && dup2(2, 1) < 0) {
      close(saved);
      saved = -1;
    }
  }
  ~StdoutToStderr() {
    if (saved < 0) return;
    fflush(stdout);
    (void)dup2(saved, 1);
    close(saved);
  }
};

extern "C" int bgp_comm_unique_id(void* id128) {
  if (!id128) {
    bgp_set_error("bgp_comm_unique_id: NULL argument");
    return BGP_ERR_INVALID;
  }
  int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  {
    StdoutToStderr quiet;
    ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) {
      bgp_set_error("ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
      return BGP_ERR_HIP;
    }
  }
  static_assert(sizeof(id) == BGP_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id128, &id, sizeof(id));
  return BGP_OK;
}

extern "C" void bgp_comm_destroy(bgp_comm* c);

extern "C" int bgp_comm_init(int device, int rank, int world, const void* id128, bgp_comm** out) {
  if (!out || !id128 || world < 1 || rank < 0 || rank >= world) {
    bgp_set_error("bgp_comm_init: bad argument (rank %d of %d)", rank, world);
    return BGP_ERR_INVALID;
  }
  *out = nullptr;
  int rc = load_rccl();
  if (rc) return rc;
  int ndev = bgp_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) {
    bgp_set_error("bgp_comm_init: no usable HIP device (count=%d, requested=%d)", ndev, device);
    return BGP_ERR_NODEVICE;
  }
  BGP_HIP(hipSetDevice(device));
  bgp_comm* c = new bgp_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r;
  {
    StdoutToStderr quiet;
    r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  }
  if (r != ncclSuccess) {
    bgp_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
    delete c;
    return BGP_ERR_HIP;
  }
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    bgp_set_error("bgp_comm_init: hipStreamCreate failed");
    (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return BGP_ERR_HIP;
  }
  // the buffers of the per-half-step exchange exist before the first half-step: no allocation (that could fail on one rank
  // only) between a submitted batch and its collective
  if (hipEventCreateWithFlags(&c->ev_ctx, hipEventDisableTiming) != hipSuccess ||
      comm_reserve(c, 4096 + 1, (size_t)(4096 + 1) * world) != BGP_OK ||
      hipHostMalloc((void**)&c->hrecv, (size_t)(4096 + 1) * world * sizeof(double), hipHostMallocDefault) != hipSuccess) {
    bgp_set_error("bgp_comm_init: allocating the exchange buffers failed");
    bgp_comm_destroy(c);
    return BGP_ERR_HIP;
  }
  c->cap_hrecv = (size_t)(4096 + 1) * world;
  *out = c;
  return BGP_OK;
}

extern "C" void bgp_comm_destroy(bgp_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream && !c->aborted) (void)hipStreamSynchronize(c->stream);
  if (c->stream && c->aborted) {
    // an aborted communicator may still have a staged download in flight into the pinned arena: wait for the stream with a
    // bound before its arena entries are forgotten (bgp_xfer_forget's contract: "has been synchronised")
    const auto t0 = std::chrono::steady_clock::now();
    while (hipStreamQuery(c->stream) == hipErrorNotReady && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) usleep(200);
    (void)hipGetLastError();
  }
  if (c->loop) {
    std::lock_guard<std::mutex> glock(g_loop_mu);
    LoopGroup* g = c->loop;
    bool last;
    {
      std::lock_guard<std::mutex> lock(g->mu);
      g->left = 1;  // (a member leaving ends the group's meetings; a peer still waiting for its complete run is not disturbed)
      g->cv.notify_all();
      last = --g->members == 0;
    }
    if (last) {
      for (int par = 0; par < 2; par++) {
        if (g->dshared[par]) (void)hipFree(g->dshared[par]);
        for (hipEvent_t e : g->ev[par]) (void)hipEventDestroy(e);
      }
      g_loops.erase(c->loop_key);
      delete g;
    }
  }
  if (c->ev_ctx) (void)hipEventDestroy(c->ev_ctx);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->dsend) (void)hipFree(c->dsend);
  if (c->drecv) (void)hipFree(c->drecv);
  if (c->hrecv) (void)hipHostFree(c->hrecv);
  if (c->stream) {
    bgp_xfer_forget(c->stream);
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
}

extern "C" int bgp_comm_abort(bgp_comm* c) {
  if (!c) {
    bgp_set_error("bgp_comm_abort: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (c->loop) {
    std::lock_guard<std::mutex> lock(c->loop->mu);
    c->loop->aborted = 1;
    c->loop->cv.notify_all();
  }
  if (!c->aborted && c->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);
  c->comm = nullptr;
  c->aborted = 1;
  return BGP_OK;
}

// Ranks RCCL itself counts in the communicator (ncclCommCount): what a caller reports as "the group that really formed".
extern "C" int bgp_comm_nranks(bgp_comm* c, int* nranks) {
  if (!c || !nranks) {
    bgp_set_error("bgp_comm_nranks: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (c->loop && !c->aborted) {
    *nranks = c->world;
    return BGP_OK;
  }
  BGP_COMM_LIVE(c, "bgp_comm_nranks");
  BGP_NCCL(g_rccl.CommCount(c->comm, nranks));
  return BGP_OK;
}

// Exact single-ensemble sharding (SURVEY.md 8e option 1): every rank has submitted ITS rows of the half-step's proposal
// block with bgp_lml_batch_submit (at most per_rank of them); this call replaces bgp_lml_batch_wait.
//
// ONE host synchronisation per half-step: the communicator's stream waits for the context's stream through an event (no
// host round trip), a pack kernel puts the rank's per_rank log-likelihoods and ONE status word side by side, RCCL
// all-gathers the world * (per_rank + 1) doubles, one copy brings them to pinned host memory, and the host waits for that
// (bounded: comm_sync).
//
// Every rank ENTERS the collective whatever happened locally -- RCCL has no time-out, a rank that returned early would
// leave its peers in the kernel.  The status word carries the news instead:
//   0               sound values;
//   local_error     the caller's own failure between submit and wait (an exception in the prior evaluation, a failed
//                   submit: any code > 0), or a failure of this function's local steps: the values are NaN;
//   BGP_RANK_REDO   the rank's launch-free factorisation timed out (read on the device from the kernel's error word): it
//                   redoes its batch by launches and EVERY rank, having seen the same word, goes round the gather again.
// errors_out[r] = status word of rank r (0 = sound).  Returns BGP_OK when the collective itself completed -- the caller
// looks at errors_out, the same on every rank, and every rank raises alike --, BGP_ERR_COMM when it did not (aborted).
__global__ void lml_pack_kernel(const double* __restrict__ dlml, int Bp, int per, const unsigned* __restrict__ ps_err,
                                int local_error, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned pe = ps_err ? __hip_atomic_load(ps_err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
  const int st = local_error ? local_error : (pe ? BGP_RANK_REDO : 0);
  if (i < per) out[i] = (i < Bp && st == 0) ? dlml[i] : __builtin_nan("");
  if (i == per) out[per] = (double)st;
}

extern "C" int bgp_lml_batch_wait_allgather(bgp_ctx* ctx, bgp_comm* c, int per_rank, int local_error, double* lml_all,
                                            int* errors_out) {
  if (!ctx || !c || !lml_all || !errors_out || per_rank <= 0 || local_error < 0) {
    // (an argument error of this kind is the same on every rank: nobody enters the collective)
    bgp_set_error("bgp_lml_batch_wait_allgather: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_COMM_LIVE(c, "bgp_lml_batch_wait_allgather");
  if (ctx->pending_B < 0) {
    bgp_set_error("bgp_lml_batch_wait_allgather: a device-resident sampler run is open on this context");
    return BGP_ERR_STATE;
  }
  const int Bp = ctx->pending_B;
  ctx->pending_B = 0;  // the pending batch is consumed here whatever happens (a rank without rows has nothing pending)
  if (per_rank > ctx->max_batch || Bp > per_rank || ctx->device != c->device) {
    bgp_set_error("bgp_lml_batch_wait_allgather: per_rank = %d must cover the pending batch (%d) and fit max_batch = %d, on "
                  "the communicator's device", per_rank, Bp, ctx->max_batch);
    local_error = local_error ? local_error : BGP_ERR_INVALID;  // ... and is reported THROUGH the collective
  }
  const size_t slot = (size_t)per_rank + 1, total = slot * c->world;
  int rc = BGP_OK;
  if (hipSetDevice(c->device) != hipSuccess) local_error = local_error ? local_error : BGP_ERR_HIP;
  if (slot > c->cap_send || total > c->cap_recv || total > c->cap_hrecv) {
    // larger than the buffers of bgp_comm_init (per_rank > 4096): grown here, the same on every rank
    rc = comm_reserve(c, slot, total);
    if (!rc && total > c->cap_hrecv) {
      if (c->hrecv) (void)hipHostFree(c->hrecv);
      c->hrecv = nullptr;
      c->cap_hrecv = 0;
      if (hipHostMalloc((void**)&c->hrecv, total * sizeof(double), hipHostMallocDefault) == hipSuccess)
        c->cap_hrecv = total;
      else
        rc = BGP_ERR_HIP;
    }
    if (rc) {  // without buffers this rank cannot take part: abort the communicator so that the peers fail too
      if (g_rccl.CommAbort && c->comm) (void)g_rccl.CommAbort(c->comm);
      c->comm = nullptr;
      c->aborted = 1;
      bgp_set_error("bgp_lml_batch_wait_allgather: growing the exchange buffers failed; communicator aborted");
      return BGP_ERR_COMM;
    }
  }
  // at most two rounds: the first may carry "redo" words (launch-free time-outs, read on the device), the second is final
  for (int round = 0; round < 2; round++) {
    // the context's work of this half-step -> the communicator's stream (device-side dependency, no host wait)
    if (hipEventRecord(c->ev_ctx, ctx->stream) != hipSuccess || hipStreamWaitEvent(c->stream, c->ev_ctx, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipStreamSynchronize(ctx->stream);  // (fall back to the host wait; the values are then certainly there)
    }
    const unsigned* pe = (round == 0 && ctx->ps_inflight && ctx->ps_flags) ? ctx->ps_flags + PS_ERROR : nullptr;
    hipLaunchKernelGGL(lml_pack_kernel, dim3((unsigned)((slot + 255) / 256)), dim3(256), 0, c->stream, ctx->dlml, Bp,
                       per_rank, pe, local_error, c->dsend);
    // (a failure to ENQUEUE the collective leaves the peers in it: abort the communicator so that they fail at once too)
    const ncclResult_t nr = g_rccl.AllGather(c->dsend, c->drecv, slot, ncclFloat64, c->comm, c->stream);
    hipError_t he = hipSuccess;
    if (nr == ncclSuccess) he = hipMemcpyAsync(c->hrecv, c->drecv, total * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (nr != ncclSuccess || he != hipSuccess) {
      bgp_set_error("bgp_lml_batch_wait_allgather: enqueueing the exchange failed (%s); communicator aborted",
                    nr != ncclSuccess ? g_rccl.GetErrorString(nr) : hipGetErrorString(he));
      (void)hipGetLastError();
      if (g_rccl.CommAbort && c->comm) (void)g_rccl.CommAbort(c->comm);
      c->comm = nullptr;
      c->aborted = 1;
      return BGP_ERR_COMM;
    }
    rc = comm_sync(c, "bgp_lml_batch_wait_allgather");
    if (rc) return rc;
    // (the communicator's stream waited for the context's: that one is complete too -- release its staged transfers)
    (void)bgp_stream_sync(ctx->stream);
    bool redo = false;
    for (int r = 0; r < c->world; r++) {
      const double sw = c->hrecv[r * slot + per_rank];
      errors_out[r] = (sw == sw && sw >= 0.0 && sw <= 2e9) ? (int)sw : BGP_ERR_COMM;
      redo = redo || errors_out[r] == BGP_RANK_REDO;
      memcpy(lml_all + (size_t)r * per_rank, c->hrecv + r * slot, (size_t)per_rank * sizeof(double));
    }
    // this rank's own launch-free call: note a time-out (and redo the batch by launches) or clear the in-flight mark -- also
    // when the rank reports a local failure (a mark left behind would make the NEXT call on the context count a spurious
    // time-out and redo a healthy batch)
    if (round == 0) {
      if (Bp > 0 && !local_error) {
        const int rr = bgp_lml_redo_if_abandoned(ctx, Bp);
        if (rr) local_error = rr;  // ... reported in the second round
      } else {
        bgp_ps_clear_inflight(ctx);
      }
    }
    if (!redo || round == 1) return BGP_OK;
  }
  return BGP_OK;
}

// ---- the sharded resident sampler's side of the communicator (bgp_mcmc.hip) ----
int bgp_comm_rank(const bgp_comm* c, int* rank, int* world) {
  if (!c || c->aborted || (!c->comm && !c->loop)) {
    bgp_set_error("bgp_mcmc_begin: the communicator is gone (aborted by an earlier failed collective)");
    return BGP_ERR_COMM;
  }
  *rank = c->rank;
  *world = c->world;
  return BGP_OK;
}

const double* bgp_comm_recv(bgp_comm* c, size_t doubles) {
  if (!c) return nullptr;
  if (doubles > c->cap_recv || doubles / c->world > c->cap_send) {
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    if (hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;  // (nothing of an earlier exchange reads the old buffers)
    if (comm_reserve(c, std::max(c->cap_send, doubles / c->world), std::max(c->cap_recv, doubles)) != BGP_OK) return nullptr;
  }
  return c->drecv;
}

static int loop_meet(LoopGroup* g, const char* what) {
  std::unique_lock<std::mutex> lock(g->mu);
  if (g->aborted || g->left) {
    bgp_set_error("%s: the loop-back group was aborted, or a member has left it", what);
    return BGP_ERR_COMM;
  }
  const unsigned long long gen = g->generation;
  if (++g->arrived == g->world) {
    g->arrived = 0;
    g->generation++;
    g->cv.notify_all();
    return BGP_OK;
  }
  const bool ok = g->cv.wait_for(lock, std::chrono::duration<double>(comm_timeout_s()), [&] { return g->generation != gen || g->aborted || g->left; });
  if (!ok || g->generation == gen) {
    g->aborted = 1;
    g->cv.notify_all();
    bgp_set_error("%s: a rank of the loop-back group did not arrive (or the group was aborted)", what);
    return BGP_ERR_COMM;
  }
  return BGP_OK;
}

int bgp_comm_enqueue_lml_gather(bgp_comm* c, bgp_ctx* ctx, hipStream_t st, int Bp, int per, const unsigned* ps_err) {
  BGP_COMM_LIVE_OR_LOOP(c, "bgp_mcmc (sharded run)");
  const size_t slot = (size_t)per + 1;
  if (slot > c->cap_send || slot * c->world > c->cap_recv || (c->loop && slot > LOOP_SLOT)) {
    bgp_set_error("bgp_mcmc (sharded run): %d rows per rank exceed the communicator's exchange buffers", per);
    return BGP_ERR_INVALID;
  }
  hipLaunchKernelGGL(lml_pack_kernel, dim3((unsigned)((slot + 255) / 256)), dim3(256), 0, st, ctx->dlml, Bp, per, ps_err, 0, c->dsend);
  if (!c->loop) {
    const ncclResult_t nr = g_rccl.AllGather(c->dsend, c->drecv, slot, ncclFloat64, c->comm, st);
    if (nr != ncclSuccess) {  // (a failure to ENQUEUE the collective leaves the peers in it: abort so that they fail at once too)
      bgp_set_error("bgp_mcmc (sharded run): enqueueing the all-gather failed (%s); communicator aborted", g_rccl.GetErrorString(nr));
      (void)bgp_comm_abort(c);
      return BGP_ERR_COMM;
    }
    return BGP_OK;
  }
  LoopGroup* g = c->loop;
  const int par = (int)(g->round[c->rank]++ & 1ull);
  BGP_HIP(hipMemcpyAsync(g->dshared[par] + (size_t)c->rank * LOOP_SLOT, c->dsend, slot * sizeof(double), hipMemcpyDeviceToDevice, st));
  BGP_HIP(hipEventRecord(g->ev[par][c->rank], st));
  const int rm = loop_meet(g, "bgp_mcmc (sharded run)");
  if (rm) return rm;
  for (int r = 0; r < c->world; r++) {
    if (r != c->rank) BGP_HIP(hipStreamWaitEvent(st, g->ev[par][r], 0));
    BGP_HIP(hipMemcpyAsync(c->drecv + (size_t)r * slot, g->dshared[par] + (size_t)r * LOOP_SLOT, slot * sizeof(double),
                           hipMemcpyDeviceToDevice, st));
  }
  // (no rank may re-record an event of this parity before every rank has enqueued its waits on it: two rounds on, behind the
  // next round's meeting -- which every rank only reaches after these waits)
  return BGP_OK;
}

int bgp_comm_wait_stream(bgp_comm* c, hipStream_t st, const char* what) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned it = 0;; it++) {
    const hipError_t e = hipStreamQuery(st);
    if (e == hipSuccess) return BGP_OK;
    (void)hipGetLastError();
    if (e != hipErrorNotReady) {
      bgp_set_error("%s: the stream failed: %s", what, hipGetErrorString(e));
      bgp_xfer_drop_pending();
      return BGP_ERR_HIP;
    }
    if ((it & 255) != 255) {
      bgp_cpu_relax();
      continue;
    }
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    ncclResult_t ar = ncclSuccess;
    const bool have = g_rccl.CommGetAsyncError && c->comm && g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess;
    const bool bad = have && ar != ncclSuccess && ar != ncclInProgress;
    bool loop_gone = false;
    if (c->loop) {
      std::lock_guard<std::mutex> lock(c->loop->mu);
      loop_gone = c->loop->aborted != 0;
    }
    if (bad || loop_gone || el > comm_timeout_s()) {
      if (bad)
        bgp_set_error("%s: RCCL reports an asynchronous error (%s); communicator aborted", what, g_rccl.GetErrorString(ar));
      else
        bgp_set_error("%s: the run's collectives did not complete within %.0f s (BGP_COMM_TIMEOUT_S): a peer has died or left; "
                      "communicator aborted", what, comm_timeout_s());
      (void)bgp_comm_abort(c);
      bgp_xfer_drop_pending();
      return BGP_ERR_COMM;
    }
    if (el > 0.2) usleep(200);  // a long wait: stop burning the core
  }
}

// Measurement hook (bench.py): `reps` rounds of the sharded resident sampler's per-half-step exchange -- pack kernel + all-gather of
// per + 1 doubles per rank -- enqueued back to back on the CONTEXT's stream between two HIP events: the device-side price of the
// exchange where the sampler pays it (in-stream, no host synchronisation), averaged per round.  Every rank calls it together.
extern "C" int bgp_comm_bench_lml_gather(bgp_ctx* ctx, bgp_comm* c, int per, int reps, double* ms_per_round) {
  if (!ctx || !c || !ms_per_round || per < 1 || reps < 1) {
    bgp_set_error("bgp_comm_bench_lml_gather: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(ctx, "bgp_comm_bench_lml_gather");
  BGP_HIP(hipSetDevice(ctx->device));
  if (!bgp_comm_recv(c, ((size_t)per + 1) * c->world)) {
    bgp_set_error("bgp_comm_bench_lml_gather: no exchange buffers for %d rows per rank", per);
    return BGP_ERR_HIP;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  BGP_HIP(hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) {
    (void)hipEventDestroy(e0);
    bgp_set_error("bgp_comm_bench_lml_gather: hipEventCreate failed");
    return BGP_ERR_HIP;
  }
  int rc = BGP_OK;
  for (int warm = 0; warm < 8 && rc == BGP_OK; warm++) rc = bgp_comm_enqueue_lml_gather(c, ctx, ctx->stream, 0, per, nullptr);
  if (rc == BGP_OK && hipEventRecord(e0, ctx->stream) != hipSuccess) rc = BGP_ERR_HIP;
  for (int i = 0; i < reps && rc == BGP_OK; i++) rc = bgp_comm_enqueue_lml_gather(c, ctx, ctx->stream, 0, per, nullptr);
  if (rc == BGP_OK && hipEventRecord(e1, ctx->stream) != hipSuccess) rc = BGP_ERR_HIP;
  if (rc == BGP_OK) rc = bgp_comm_wait_stream(c, ctx->stream, "bgp_comm_bench_lml_gather");
  float ms = 0.f;
  if (rc == BGP_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = BGP_ERR_HIP;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipGetLastError();
  if (rc == BGP_OK) *ms_per_round = (double)ms / reps;
  return rc;
}

// Loop-back communicator `rank` of `world` on `device`: the communicators of one process that name the same key form a group
// (see LoopGroup).  Test infrastructure of the sharded sampler on one GPU; no RCCL behind it.
extern "C" int bgp_comm_init_loopback(int device, int rank, int world, long long key, bgp_comm** out) {
  if (!out || world < 1 || world > 64 || rank < 0 || rank >= world) {
    bgp_set_error("bgp_comm_init_loopback: bad argument (rank %d of %d)", rank, world);
    return BGP_ERR_INVALID;
  }
  *out = nullptr;
  const int ndev = bgp_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) {
    bgp_set_error("bgp_comm_init_loopback: no usable HIP device (count=%d, requested=%d)", ndev, device);
    return BGP_ERR_NODEVICE;
  }
  BGP_HIP(hipSetDevice(device));
  bgp_comm* c = new bgp_comm();
  c->device = device;
  c->rank = rank;
  c->world = world;
  c->loop_key = key;
  {
    std::lock_guard<std::mutex> glock(g_loop_mu);
    LoopGroup*& g = g_loops[key];
    if (!g) {
      g = new LoopGroup();
      g->world = world;
      g->round.assign(world, 0ull);
      bool ok = true;
      for (int par = 0; par < 2 && ok; par++) {
        ok = hipMalloc((void**)&g->dshared[par], (size_t)world * LOOP_SLOT * sizeof(double)) == hipSuccess;
        g->ev[par].assign(world, nullptr);
        for (int r = 0; r < world && ok; r++) ok = hipEventCreateWithFlags(&g->ev[par][r], hipEventDisableTiming) == hipSuccess;
      }
      if (!ok) {
        bgp_set_error("bgp_comm_init_loopback: allocating the group's buffers failed");
        for (int par = 0; par < 2; par++) {
          if (g->dshared[par]) (void)hipFree(g->dshared[par]);
          for (hipEvent_t e : g->ev[par])
            if (e) (void)hipEventDestroy(e);
        }
        delete g;
        g_loops.erase(key);
        delete c;
        return BGP_ERR_HIP;
      }
    }
    if (g->world != world || g->members >= world) {
      bgp_set_error("bgp_comm_init_loopback: key %lld names a group of %d ranks with %d members", key, g->world, g->members);
      delete c;
      return BGP_ERR_INVALID;
    }
    g->members++;
    c->loop = g;
  }
  if (hipStreamCreate(&c->stream) != hipSuccess || comm_reserve(c, LOOP_SLOT, (size_t)LOOP_SLOT * world) != BGP_OK) {
    bgp_set_error("bgp_comm_init_loopback: allocating the exchange buffers failed");
    bgp_comm_destroy(c);
    return BGP_ERR_HIP;
  }
  *out = c;
  return BGP_OK;
}

extern "C" int bgp_comm_allgather(bgp_comm* c, const double* send, size_t count, double* recv) {
  if (!c || !send || !recv) {
    bgp_set_error("bgp_comm_allgather: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_COMM_LIVE(c, "bgp_comm_allgather");
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count * c->world);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, send, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.AllGather(c->dsend, c->drecv, count, ncclFloat64, c->comm, c->stream));
  return comm_download(c, recv, count * c->world, "bgp_comm_allgather");
}

extern "C" int bgp_comm_allreduce_max(bgp_comm* c, double* inout, size_t count) {
  if (!c || !inout) {
    bgp_set_error("bgp_comm_allreduce_max: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_COMM_LIVE(c, "bgp_comm_allreduce_max");
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, inout, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.AllReduce(c->dsend, c->drecv, count, ncclFloat64, ncclMax, c->comm, c->stream));
  return comm_download(c, inout, count, "bgp_comm_allreduce_max");
}

extern "C" int bgp_comm_broadcast(bgp_comm* c, double* buf, size_t count, int root) {
  if (!c || !buf || root < 0 || root >= c->world) {
    bgp_set_error("bgp_comm_broadcast: bad argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_COMM_LIVE(c, "bgp_comm_broadcast");
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, buf, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.Broadcast(c->dsend, c->drecv, count, ncclFloat64, root, c->comm, c->stream));
  return comm_download(c, buf, count, "bgp_comm_broadcast");
}

extern "C" int bgp_comm_barrier(bgp_comm* c) {
  double one = 1.0;
  return bgp_comm_allreduce_max(c, &one, 1);
}
