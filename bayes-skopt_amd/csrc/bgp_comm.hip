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

#include <dlfcn.h>
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
  g_rccl.handle = h;
  return BGP_OK;
}
}  // namespace

struct bgp_comm {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* dsend = nullptr;  // device-resident staging, grown on demand
  double* drecv = nullptr;
  size_t cap_send = 0, cap_recv = 0;
  double* hrecv = nullptr;  // pinned landing buffer of the device-resident gathers
  size_t cap_hrecv = 0;
};

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
  *out = c;
  return BGP_OK;
}

extern "C" void bgp_comm_destroy(bgp_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  if (c->dsend) (void)hipFree(c->dsend);
  if (c->drecv) (void)hipFree(c->drecv);
  if (c->hrecv) (void)hipHostFree(c->hrecv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// Ranks RCCL itself counts in the communicator (ncclCommCount): what a caller reports as "the group that really formed".
extern "C" int bgp_comm_nranks(bgp_comm* c, int* nranks) {
  if (!c || !nranks) {
    bgp_set_error("bgp_comm_nranks: NULL argument");
    return BGP_ERR_INVALID;
  }
  BGP_NCCL(g_rccl.CommCount(c->comm, nranks));
  return BGP_OK;
}

// Exact single-ensemble sharding (SURVEY.md 8e option 1): every rank has submitted ITS rows of the half-step's proposal
// block with bgp_lml_batch_submit (at most per_rank of them); this call replaces bgp_lml_batch_wait.  The communicator's
// stream waits for the context's stream, RCCL all-gathers per_rank doubles straight out of every context's device-resident
// log-likelihood vector, and ONE copy brings the world * per_rank values to the host (rank-major; the entries behind a
// rank's own row count are padding).  No host staging on the send side, no second synchronisation.
extern "C" int bgp_lml_batch_wait_allgather(bgp_ctx* ctx, bgp_comm* c, int per_rank, double* lml_all) {
  if (!ctx || !c || !lml_all || per_rank <= 0) {
    bgp_set_error("bgp_lml_batch_wait_allgather: bad argument");
    return BGP_ERR_INVALID;
  }
  if (per_rank > ctx->max_batch || ctx->pending_B > per_rank || ctx->device != c->device) {
    bgp_set_error("bgp_lml_batch_wait_allgather: per_rank = %d must cover the pending batch (%d) and fit max_batch = %d, on "
                  "the communicator's device", per_rank, ctx->pending_B, ctx->max_batch);
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  const size_t total = (size_t)per_rank * c->world;
  int rc = comm_reserve(c, 0, total);
  if (rc) return rc;
  if (total > c->cap_hrecv) {
    if (c->hrecv) (void)hipHostFree(c->hrecv);
    c->hrecv = nullptr;
    c->cap_hrecv = 0;
    BGP_HIP(hipHostMalloc((void**)&c->hrecv, total * sizeof(double), hipHostMallocDefault));
    c->cap_hrecv = total;
  }
  const int Bp = ctx->pending_B;
  ctx->pending_B = 0;  // (a rank without rows of its own has nothing pending: it contributes padding)
  // the local batch must be complete and sound before its values leave the device: a launch-free factorisation that
  // timed out is redone here (this wait is the host's only synchronisation with the context's stream per half-step)
  BGP_HIP(bgp_stream_sync(ctx->stream));
  if (Bp > 0) {
    rc = bgp_lml_redo_if_abandoned(ctx, Bp);
    if (rc) return rc;
  }
  BGP_NCCL(g_rccl.AllGather(ctx->dlml, c->drecv, (size_t)per_rank, ncclFloat64, c->comm, c->stream));
  BGP_HIP(hipMemcpyAsync(c->hrecv, c->drecv, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  memcpy(lml_all, c->hrecv, total * sizeof(double));
  return BGP_OK;
}

extern "C" int bgp_comm_allgather(bgp_comm* c, const double* send, size_t count, double* recv) {
  if (!c || !send || !recv) {
    bgp_set_error("bgp_comm_allgather: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count * c->world);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, send, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.AllGather(c->dsend, c->drecv, count, ncclFloat64, c->comm, c->stream));
  BGP_HIP(bgp_memcpy_async(recv, c->drecv, count * c->world * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

extern "C" int bgp_comm_allreduce_max(bgp_comm* c, double* inout, size_t count) {
  if (!c || !inout) {
    bgp_set_error("bgp_comm_allreduce_max: NULL argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, inout, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.AllReduce(c->dsend, c->drecv, count, ncclFloat64, ncclMax, c->comm, c->stream));
  BGP_HIP(bgp_memcpy_async(inout, c->drecv, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

extern "C" int bgp_comm_broadcast(bgp_comm* c, double* buf, size_t count, int root) {
  if (!c || !buf || root < 0 || root >= c->world) {
    bgp_set_error("bgp_comm_broadcast: bad argument");
    return BGP_ERR_INVALID;
  }
  if (count == 0) return BGP_OK;
  BGP_HIP(hipSetDevice(c->device));
  int rc = comm_reserve(c, count, count);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(c->dsend, buf, count * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_NCCL(g_rccl.Broadcast(c->dsend, c->drecv, count, ncclFloat64, root, c->comm, c->stream));
  BGP_HIP(bgp_memcpy_async(buf, c->drecv, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

extern "C" int bgp_comm_barrier(bgp_comm* c) {
  double one = 1.0;
  return bgp_comm_allreduce_max(c, &one, 1);
}
