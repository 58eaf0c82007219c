// C-ABI entry points of libbgp.so (see include/bgp.h for the contract and the reference seams).
#include "bgp_common.h"

#include <algorithm>
#include <cstdlib>

static thread_local std::string g_err;

void bgp_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

extern "C" const char* bgp_last_error(void) { return g_err.c_str(); }

#include <mutex>
static BgpXfer g_xfer;
static std::mutex g_xfer_mutex;

void bgp_xfer_drop_pending() {
  std::lock_guard<std::mutex> lock(g_xfer_mutex);
  BgpXfer& x = g_xfer;
  const std::thread::id me = std::this_thread::get_id();
  size_t keep = 0;
  for (size_t i = 0; i < x.pending.size(); i++) {
    const BgpXfer::Pending q = x.pending[i];
    if (q.owner != me) {
      x.pending[keep++] = q;
      continue;
    }
    // the copy may still be in flight into its arena block: the block must outlive it (release / forget of the stream clear this)
    if (std::find(x.busy.begin(), x.busy.end(), q.st) == x.busy.end()) x.busy.push_back(q.st);
  }
  x.pending.resize(keep);
}

void bgp_xfer_forget(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_xfer_mutex);
  BgpXfer& x = g_xfer;
  size_t keep = 0;
  for (size_t i = 0; i < x.pending.size(); i++)
    if (x.pending[i].st != st) x.pending[keep++] = x.pending[i];
  x.pending.resize(keep);
  x.busy.erase(std::remove(x.busy.begin(), x.busy.end(), st), x.busy.end());
  x.maybe_reset();
}

char* BgpXfer::take(size_t bytes) {
  bytes = (bytes + 255) & ~(size_t)255;
  for (Block& b : blocks)
    if (b.cap - b.off >= bytes) {
      char* p = b.p + b.off;
      b.off += bytes;
      return p;
    }
  Block nb;
  nb.cap = std::max(bytes, (size_t)4 << 20);
  nb.off = bytes;
  nb.p = nullptr;
  if (hipHostMalloc((void**)&nb.p, nb.cap, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  blocks.push_back(nb);
  return nb.p;
}

void BgpXfer::release(hipStream_t st) {
  size_t keep = 0;
  for (size_t i = 0; i < pending.size(); i++) {
    const Pending& q = pending[i];
    if (q.st != st) {
      pending[keep++] = q;
      continue;
    }
    for (size_t r = 0; r < q.height; r++) memcpy(q.host + r * q.hpitch, q.stage + r * q.width, q.width);
  }
  pending.resize(keep);
  busy.erase(std::remove(busy.begin(), busy.end(), st), busy.end());
  maybe_reset();
}

void BgpXfer::maybe_reset() {
  if (pending.empty() && busy.empty()) {  // nothing staged is in flight any more: the arena starts over
    size_t total = 0;
    for (Block& b : blocks) total += b.cap;
    if (blocks.size() > 1 || total > BGP_XFER_KEEP) {  // several blocks: one of the total size next time, capped
      for (Block& b : blocks) (void)hipHostFree(b.p);
      blocks.clear();
      if (total <= BGP_XFER_KEEP) {
        Block nb;
        nb.cap = total;
        nb.off = 0;
        nb.p = nullptr;
        if (hipHostMalloc((void**)&nb.p, nb.cap, hipHostMallocDefault) == hipSuccess) blocks.push_back(nb);
        else (void)hipGetLastError();
      }
    }
    for (Block& b : blocks) b.off = 0;
  }
}

void bgp_xfer_release(hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_xfer_mutex);
  g_xfer.release(st);
}

// the last context of the process is gone: give the pinned arena back (bgp_ctx_destroy)
static int g_live_contexts = 0;
static void xfer_free_all() {
  std::lock_guard<std::mutex> lock(g_xfer_mutex);
  if (!g_xfer.pending.empty() || !g_xfer.busy.empty()) return;
  for (BgpXfer::Block& b : g_xfer.blocks) (void)hipHostFree(b.p);
  g_xfer.blocks.clear();
}

hipError_t bgp_memcpy2d_async(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height,
                              hipMemcpyKind kind, hipStream_t st) {
  if (width == 0 || height == 0) return hipSuccess;
  if ((kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToHost) && width * height > BGP_XFER_DIRECT) {
    // large: synchronously, straight between the caller's buffer and the device, behind everything the stream holds
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    bgp_xfer_release(st);
    if (height == 1 || (dpitch == width && spitch == width)) return hipMemcpy(dst, src, width * height, kind);
    return hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind);
  }
  std::lock_guard<std::mutex> lock(g_xfer_mutex);
  BgpXfer& x = g_xfer;
  if (kind == hipMemcpyHostToDevice) {
    char* stage = x.take(width * height);
    if (!stage) return hipErrorOutOfMemory;
    for (size_t r = 0; r < height; r++) memcpy(stage + r * width, (const char*)src + r * spitch, width);
    if (std::find(x.busy.begin(), x.busy.end(), st) == x.busy.end()) x.busy.push_back(st);
    if (height == 1 || dpitch == width) return hipMemcpyAsync(dst, stage, width * height, hipMemcpyHostToDevice, st);
    return hipMemcpy2DAsync(dst, dpitch, stage, width, width, height, hipMemcpyHostToDevice, st);
  }
  if (kind == hipMemcpyDeviceToHost) {
    char* stage = x.take(width * height);
    if (!stage) return hipErrorOutOfMemory;
    x.pending.push_back({st, (char*)dst, dpitch, stage, width, height, std::this_thread::get_id()});
    if (height == 1 || spitch == width) return hipMemcpyAsync(stage, src, width * height, hipMemcpyDeviceToHost, st);
    return hipMemcpy2DAsync(stage, width, src, spitch, width, height, hipMemcpyDeviceToHost, st);
  }
  return hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, kind, st);
}

int bgp_wait_spins() {
  static const int spins = [] {
    const char* e = getenv("BGP_WAIT");
    return (e && strcmp(e, "block") == 0) ? 0 : 1;
  }();
  return spins;
}
extern "C" const char* bgp_version(void) { return "bgp 0.1 (gfx950, fp64 MFMA blocked Cholesky)"; }

extern "C" int bgp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

extern "C" int bgp_device_pci_bus_id(int device, char* buf, int len) {
  if (!buf || len < 16) {
    bgp_set_error("bgp_device_pci_bus_id: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipDeviceGetPCIBusId(buf, len, device));
  return BGP_OK;
}

static void free_dev(void* p) {
  if (p) (void)hipFree(p);
}

static int alloc_data(bgp_ctx* c, int n) {
  // training-set buffers sized for npad rows; grow-only
  const size_t npad = (size_t)((n + BGP_NB - 1) / BGP_NB) * BGP_NB;
  if (npad > c->cap_n) {
    free_dev(c->dX);
    free_dev(c->dy);
    free_dev(c->dalpha);
    free_dev(c->dXw1);
    free_dev(c->dXwB);
    c->dX = c->dy = c->dalpha = c->dXw1 = c->dXwB = nullptr;
    c->cap_xwb = 0;
    BGP_HIP(hipMalloc(&c->dX, npad * c->d * sizeof(double)));
    BGP_HIP(hipMalloc(&c->dy, npad * sizeof(double)));
    BGP_HIP(hipMalloc(&c->dalpha, npad * sizeof(double)));
    c->cap_n = npad;
  }
  return BGP_OK;
}

static int alloc_work(bgp_ctx* c) {
  const size_t npad = c->npad, nblk = c->nblk, mb = c->max_batch;
  const size_t need_mat = mb * npad * npad;
  const size_t need_w = mb * nblk * 128 * 128;
  const size_t need_yw = mb * 2 * npad;  // augmented right-hand sides are 2 npad long
  if (need_mat > c->cap_mat) {
    free_dev(c->dK);
    c->dK = nullptr;
    c->cap_mat = 0;
    BGP_HIP(hipMalloc(&c->dK, need_mat * sizeof(double)));
    c->cap_mat = need_mat;
  }
  if (need_w > c->cap_w) {
    free_dev(c->dW);
    c->dW = nullptr;
    c->cap_w = 0;
    BGP_HIP(hipMalloc(&c->dW, need_w * sizeof(double)));
    c->cap_w = need_w;
  }
  if (need_yw > c->cap_yw) {
    free_dev(c->dyw);
    c->dyw = nullptr;
    c->cap_yw = 0;
    BGP_HIP(hipMalloc(&c->dyw, need_yw * sizeof(double)));
    c->cap_yw = need_yw;
  }
  return BGP_OK;
}

int bgp_grow_workspace(bgp_ctx* c, size_t doubles) {
  if (doubles > c->cap_mat) {
    free_dev(c->dK);
    c->dK = nullptr;
    c->cap_mat = 0;
    BGP_HIP(hipMalloc(&c->dK, doubles * sizeof(double)));
    c->cap_mat = doubles;
  }
  return BGP_OK;
}

static int upload_data(bgp_ctx* c, int n, const double* X, const double* y, const double* alpha_diag) {
  if (n <= 0 || !X || !y || !alpha_diag) {
    bgp_set_error("bgp: n must be > 0 and X, y, alpha_diag non-NULL");
    return BGP_ERR_INVALID;
  }
  static const bool dbg = getenv("BGP_DEBUG_TIMES") != nullptr;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
  };
  const auto t0 = now();
  int rc = alloc_data(c, n);
  if (rc) return rc;
  c->n = n;
  c->npad = ((n + BGP_NB - 1) / BGP_NB) * BGP_NB;
  c->nblk = c->npad / BGP_NB;
  rc = alloc_work(c);
  if (rc) return rc;
  const auto t1 = now();
  // Staged through the context's own pinned buffer: copies from pageable memory (numpy arrays, fresh vectors) were
  // measured at 10-20 ms per tell on MI355X for these 80 KB (the runtime locks / unlocks the pages around each copy).
  const size_t nx = (size_t)n * c->d, np_ = c->npad, need = nx + 2 * np_;
  if (need > c->cap_stage) {
    if (c->hstage) (void)hipHostFree(c->hstage);
    c->hstage = nullptr;
    c->cap_stage = 0;
    BGP_HIP(hipHostMalloc((void**)&c->hstage, (need + need / 4) * sizeof(double), hipHostMallocDefault));
    c->cap_stage = need + need / 4;
  }
  double *hx = c->hstage, *hy = hx + nx, *ha = hy + np_;
  memcpy(hx, X, nx * sizeof(double));
  memset(hy, 0, 2 * np_ * sizeof(double));
  memcpy(hy, y, n * sizeof(double));
  memcpy(ha, alpha_diag, n * sizeof(double));
  BGP_HIP(hipMemcpyAsync(c->dX, hx, nx * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemcpyAsync(c->dy, hy, np_ * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemcpyAsync(c->dalpha, ha, np_ * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const auto t2 = now();
  BGP_HIP(bgp_stream_sync(c->stream));
  if (dbg) fprintf(stderr, "upload_data: alloc %.3f ms, stage+enqueue %.3f ms, sync %.3f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
  c->post_B = 0;
  c->has_warp = 0;  // new data: the caller re-installs the warp (bgp_ctx_set_warp)
  c->dXeff = c->dX;
  return BGP_OK;
}

// A/B measurements depend on the BGP_* switches: a misspelt one would silently measure the default twice.
extern char** environ;
static void warn_unknown_env_once() {
  static bool done = false;
  if (done) return;
  done = true;
  static const char* known[] = {"BGP_COMM_DIR", "BGP_COMM_PORT", "BGP_COMM_TCP", "BGP_DIST_BACKEND", "BGP_DIST_FORCE",
                                "BGP_COMM_JOB", "BGP_BENCH_TIMEOUT", "BGP_DEBUG_TIMES", "BGP_PANELS", "BGP_NO_ENV_DEFAULTS",
                                "BGP_PERSIST", "BGP_PS_COOLDOWN", "BGP_PS_GEN", "BGP_SYRK_GEN", "BGP_PS_PAIR", "BGP_PS_TIMEOUT_MS", "BGP_PS_TIMEOUT_TICKS", "BGP_PS_TRACE", "BGP_STREAMS", "BGP_WAIT", "BGP_COMM_TIMEOUT_S"};
  for (char** e = environ; e && *e; e++) {
    if (strncmp(*e, "BGP_", 4) != 0) continue;
    const char* eq = strchr(*e, '=');
    const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
    bool ok = false;
    for (const char* k : known) ok = ok || (strlen(k) == len && strncmp(k, *e, len) == 0);
    if (!ok) fprintf(stderr, "libbgp: warning: environment variable %.*s is not one this library reads\n", (int)len, *e);
  }
}

extern "C" int bgp_ctx_create(int device, int n, int d, const double* X, const double* y, const double* alpha_diag,
                              const bgp_kernel_spec* ks, int max_batch, bgp_ctx** out) {
  warn_unknown_env_once();
  if (!out || !ks) {
    bgp_set_error("bgp_ctx_create: NULL argument");
    return BGP_ERR_INVALID;
  }
  *out = nullptr;
  if (d <= 0 || d > BGP_MAX_D || ks->d != d) {
    bgp_set_error("bgp_ctx_create: d=%d out of range (1..%d) or != spec.d=%d", d, BGP_MAX_D, ks->d);
    return BGP_ERR_INVALID;
  }
  if (ks->form < 0 || ks->form > 1 || ks->stationary < 0 || ks->stationary > 3) {
    bgp_set_error("bgp_ctx_create: bad kernel spec (form=%d stationary=%d)", ks->form, ks->stationary);
    return BGP_ERR_INVALID;
  }
  if (max_batch <= 0) {
    bgp_set_error("bgp_ctx_create: max_batch must be > 0");
    return BGP_ERR_INVALID;
  }
  int ndev = bgp_device_count();
  if (ndev <= 0 || device < 0 || device >= ndev) {
    bgp_set_error("bgp_ctx_create: no usable HIP device (count=%d, requested=%d)", ndev, device);
    return BGP_ERR_NODEVICE;
  }
  BGP_HIP(hipSetDevice(device));
  bgp_ctx* c = new bgp_ctx();
  {
    std::lock_guard<std::mutex> lock(g_xfer_mutex);
    g_live_contexts++;
  }
  c->device = device;
  c->d = d;
  c->ks = *ks;
  c->max_batch = max_batch;
  if (hipDeviceGetAttribute(&c->ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) c->ncu = 0;
  if (hipStreamCreate(&c->stream) != hipSuccess) {
    bgp_set_error("hipStreamCreate failed");
    delete c;
    std::lock_guard<std::mutex> lock(g_xfer_mutex);
    g_live_contexts--;
    return BGP_ERR_HIP;
  }
  {
    // walker-group streams: BGP_STREAMS=k forces k groups; unset = automatic (two groups for batches of
    // >= 64 matrices, where the second group's kernels fill the tail of the first group's launches: +3.7 %
    // at BASELINE config C, results bit-identical; one group below that)
    const char* env = getenv("BGP_STREAMS");
    int ns = env ? atoi(env) : 2;
    c->streams_auto = env ? 0 : 1;
    if (ns < 1) ns = 1;
    if (ns > BGP_MAX_STREAMS) ns = BGP_MAX_STREAMS;
    c->nstreams = ns;
    const char* envps = getenv("BGP_PERSIST");  // 0: never, 1: whenever the batch fits (<= 64 matrices, n > 128); unset: automatic
    c->persist = envps ? (atoi(envps) != 0 ? 1 : 0) : -1;
    c->panels = 2;
    c->panels_auto = 1;
    const char* envp = getenv("BGP_PANELS");
    if (envp && atoi(envp) >= 1 && atoi(envp) <= 64) {
      c->panels = atoi(envp);
      c->panels_auto = 0;
    }
    (void)hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
    for (int g = 0; g < ns; g++) {
      (void)hipStreamCreate(&c->gstream[g]);
      (void)hipEventCreateWithFlags(&c->ev_done[g], hipEventDisableTiming);
    }
  }
  const size_t mb = max_batch;
  int rc = BGP_OK;
  do {
    if (hipMalloc(&c->dacc, mb * 4 * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->dh, mb * (d + 2) * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->dlml, mb * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->dstatus, mb * sizeof(int)) != hipSuccess) {
      bgp_set_error("hipMalloc of per-batch buffers failed");
      rc = BGP_ERR_HIP;
      break;
    }
    rc = upload_data(c, n, X, y, alpha_diag);
  } while (0);
  if (rc) {
    bgp_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return BGP_OK;
}

extern "C" int bgp_ctx_update_data(bgp_ctx* c, int n, const double* X, const double* y, const double* alpha_diag) {
  BGP_REQUIRE_IDLE(c, "bgp_ctx_update_data");
  if (!c) {
    bgp_set_error("bgp_ctx_update_data: NULL ctx");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  return upload_data(c, n, X, y, alpha_diag);
}

extern "C" void bgp_ctx_destroy(bgp_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  bgp_mcmc_abandon(c);  // (a sampler run left open: its work is drained, its block freed)
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  bgp_free_child(c);
  free_dev(c->dX);
  free_dev(c->dy);
  free_dev(c->dalpha);
  free_dev(c->dK);
  free_dev(c->dW);
  free_dev(c->dyw);
  free_dev(c->dacc);
  free_dev(c->dh);
  free_dev(c->dlml);
  free_dev(c->dstatus);
  free_dev(c->dalpha_sol);
  free_dev(c->dKinv);
  free_dev(c->dXw1);
  free_dev(c->dXwB);
  free_dev(c->dXs);
  free_dev(c->dwarp);
  free_dev(c->dwarpB);
  free_dev(c->dscratch);
  free_dev(c->drowpart);
  if (c->hstage) (void)hipHostFree(c->hstage);
  if (c->hlml) (void)hipHostFree(c->hlml);
  if (c->hstatus) (void)hipHostFree(c->hstatus);
  if (c->hh) (void)hipHostFree(c->hh);
  if (c->hwarp) (void)hipHostFree(c->hwarp);
  free_dev(c->ps_flags);
  free_dev(c->ps_trace);
  if (c->ps_herr) (void)hipHostFree(c->ps_herr);
  for (int g = 0; g < BGP_MAX_STREAMS; g++) {
    if (c->gstream[g]) {
      (void)hipStreamSynchronize(c->gstream[g]);
      bgp_xfer_forget(c->gstream[g]);
      (void)hipStreamDestroy(c->gstream[g]);
    }
    if (c->ev_done[g]) (void)hipEventDestroy(c->ev_done[g]);
  }
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->stream) {
    bgp_xfer_forget(c->stream);  // (synchronised above: nothing of this context is in flight into the arena any more)
    (void)hipStreamDestroy(c->stream);
  }
  delete c;
  bool last;
  {
    std::lock_guard<std::mutex> lock(g_xfer_mutex);
    last = --g_live_contexts == 0;
  }
  if (last) xfer_free_all();
}

int bgp_ensure_scratch(bgp_ctx* c, size_t doubles) {
  if (doubles > c->cap_scratch) {
    free_dev(c->dscratch);
    c->dscratch = nullptr;
    c->cap_scratch = 0;
    BGP_HIP(hipMalloc(&c->dscratch, doubles * sizeof(double)));
    c->cap_scratch = doubles;
  }
  return BGP_OK;
}

// Factorise one chunk (<= max_batch) of hyper-parameter vectors already validated by the caller.
static int factor_chunk(bgp_ctx* c, int B, const double* h, int full_square) {
  const size_t p = c->d + 2;
  BGP_HIP(bgp_memcpy_async(c->dh, h, B * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemsetAsync(c->dstatus, 0, B * sizeof(int), c->stream));
  int rc = bgp_launch_kbuild(c, B, full_square, 0, 1);
  if (rc) return rc;
  return BGP_OK;
}

static int lml_batch_impl(bgp_ctx* c, int B, const double* h, const double* warp, double* lml, int* status);

static bool bgp_persist_auto(const bgp_ctx* c, int nb) { return bgp_persist_auto_rule(c->nblk, nb); }

int bgp_ps_cooldown_calls() {
  static int v = 0;
  if (!v) {
    const char* e = getenv("BGP_PS_COOLDOWN");
    v = (e && atoi(e) > 0) ? atoi(e) : 256;
  }
  return v;
}

// See include/bgp.h.
extern "C" int bgp_lml_gen_stats(bgp_ctx* c, long long* out) {
  if (!c || !out) return BGP_ERR_INVALID;
  out[0] = c->gen_batches;
  out[1] = c->gen_launches;
  return BGP_OK;
}

// Launch-free path bookkeeping of a context: out[0] = launch-free calls enqueued, out[1] = of which timed out (redone by
// launches), out[2] = 1 while the path is switched off by a time-out, out[3] = eligible calls left before it is tried again
// (0 with out[2] == 1: off for good after three time-outs, until bgp_set_persist(ctx, 1)).
extern "C" int bgp_persist_stats(bgp_ctx* c, long long* out) {
  if (!c || !out) return BGP_ERR_INVALID;
  out[0] = c->ps_calls;
  out[1] = c->ps_timeouts;
  out[2] = c->ps_disabled;
  out[3] = c->ps_cooldown;
  return BGP_OK;
}

// A persistent call whose waits timed out (error word != 0 behind the synchronisation) is redone on the multi-launch
// path, loudly; bgp_ps_note_timeout decides how long the context stays there.
static int ps_check(bgp_ctx* c) {
  if (!c->ps_inflight) return 0;
  c->ps_inflight = 0;
  if (!c->ps_herr || *c->ps_herr == 0) return 0;
  *c->ps_herr = 0;
  bgp_ps_note_timeout(c, "the batch is redone");
  return 1;
}

extern "C" int bgp_lml_batch_submit(bgp_ctx* c, int B, const double* h);
extern "C" int bgp_lml_batch_wait(bgp_ctx* c, double* lml, int* status);

extern "C" int bgp_lml_batch(bgp_ctx* c, int B, const double* h, double* lml, int* status) {
  if (!c || !h || !lml || B < 0) {
    bgp_set_error("bgp_lml_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  // one chunk, no per-launch timing: the pinned submit / wait path (uploads and downloads that do not lock pageable
  // memory around every call: 0.079 -> 0.06 ms for 50 proposals at n = 128)
  if (c->pending_B != 0) {  // the pending batch owns the workspace until bgp_lml_batch_wait collects it
    bgp_set_error("bgp_lml_batch: a submitted batch is still pending (call bgp_lml_batch_wait)");
    return BGP_ERR_STATE;
  }
  if (B > 0 && B <= c->max_batch && !c->timing) {
    const int rc = bgp_lml_batch_submit(c, B, h);
    if (rc) return rc;
    return bgp_lml_batch_wait(c, lml, status);
  }
  return lml_batch_impl(c, B, h, nullptr, lml, status);
}

extern "C" int bgp_lml_batch_warped(bgp_ctx* c, int B, const double* h, const double* warp, double* lml,
                                    int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_lml_batch_warped");
  if (!c || !h || !warp || !lml || B < 0) {
    bgp_set_error("bgp_lml_batch_warped: bad argument");
    return BGP_ERR_INVALID;
  }
  return lml_batch_impl(c, B, h, warp, lml, status);
}

// Everything of an LML batch that runs on the device, enqueued on the context's stream(s): the nb canonical hyper-parameter
// vectors are in c->dh (and, warped, the per-walker warp parameters in c->dwarpB) already; Gram build, factorisation and the
// LML reduction leave c->dlml / c->dstatus.  No upload, no download, no synchronisation: bgp_lml_batch* wrap it with their
// transfers, the device-resident ensemble sampler (bgp_mcmc.hip) calls it between its own kernels.
int bgp_lml_enqueue_dev(bgp_ctx* c, int nb, int warped) {
  const size_t nd = (size_t)c->n * c->d;
  // walker groups on separate streams (sizes are multiples of 8: one matrix slot per XCD)
  int ng = (c->streams_auto && nb < 64) ? 1 : c->nstreams;
  int gsz = ((nb + ng - 1) / ng + 7) / 8 * 8;
  if (ng == 1 || nb < 16 || warped) {
  ng = 1;
  gsz = nb;
  }
  const bool warp = warped != 0;
  int rc = BGP_OK;
  const bool fused_small = c->nblk == 1 && !warp;
  // launch-free path (bgp_chol.hip, ps_chain_kernel): automatic below 64 matrices per call when the matrices have at
  // least two block columns; never under per-launch timing (there are no launches to time) or after a timeout
  const bool use_ps = !fused_small && !c->timing && !c->ps_forbid && bgp_persist_fits(c, nb) &&
                      (c->persist == 1 || (c->persist == -1 && bgp_persist_auto(c, nb))) && bgp_ps_allowed(c);
  if (use_ps) c->ps_calls++;
  // (ps_resident: the sampler's step kernel has reset the statuses in front of this batch)
  if (!fused_small && !c->ps_resident) BGP_HIP(hipMemsetAsync(c->dstatus, 0, nb * sizeof(int), c->stream));
  if (fused_small) {
    // n <= 128: Gram generation, factorisation and LML fused into one launch (status is reset in the kernel)
    rc = bgp_launch_lml_small(c, 0, nb, c->stream);
    if (rc) return rc;
  } else if (warp) {
    // per-walker Beta-CDF warp of the design matrix, then the right-looking path on per-walker inputs
    rc = bgp_launch_warp(c, c->stream, c->dX, c->dwarpB, c->dXwB, c->n, nb, nd);
    if (rc) return rc;
    const int gen = !use_ps && bgp_lml_gen_eligible(c, nb);
    rc = bgp_launch_kbuild_x(c, 0, nb, c->stream, gen ? 2 : 0, 0, 1, c->dXwB, nd);
    if (rc) return rc;
    if (use_ps) {
      rc = bgp_launch_cholesky_persist(c, nb, 0);
      if (!rc) c->ps_inflight = 1;
    } else {
      rc = bgp_launch_cholesky_slice(c, 0, nb, c->stream, 0, gen);
    }
    if (rc) return rc;
  } else if (use_ps) {
    // small batch: ONE launch-free kernel instead of ~3 launches per block column, the Gram build inside it or in front of it
    rc = bgp_launch_cholesky_persist(c, nb, 1);
    if (rc) return rc;
    c->ps_inflight = 1;
  } else if (ng == 1) {
    // (gen: the Gram kernel builds block column 0, the first panel group's updates generate the rest in their accumulators)
    const int gen = bgp_lml_gen_eligible(c, nb);
    rc = bgp_launch_kbuild(c, nb, gen ? 2 : 0, 0, 1);
    if (rc) return rc;
    rc = bgp_launch_cholesky_slice(c, 0, nb, c->stream, 0, gen);
    if (rc) return rc;
  } else {
    BGP_HIP(hipEventRecord(c->ev_ready, c->stream));
    for (int g = 0, o = 0; o < nb; g++, o += gsz) {
      const int gb = std::min(gsz, nb - o);
      hipStream_t st = c->gstream[g];
      BGP_HIP(hipStreamWaitEvent(st, c->ev_ready, 0));
      const int gen = bgp_lml_gen_eligible(c, gb);
      rc = bgp_launch_kbuild_slice(c, o, gb, st, gen ? 2 : 0, 0, 1);
      if (rc) return rc;
      rc = bgp_launch_cholesky_slice(c, o, gb, st, 0, gen);
      if (rc) return rc;
      BGP_HIP(hipEventRecord(c->ev_done[g], st));
      BGP_HIP(hipStreamWaitEvent(c->stream, c->ev_done[g], 0));
    }
  }
  return BGP_OK;
}

// per-walker warped inputs (max_batch x n x d) and warp parameters (max_batch x 2 d) of a warped LML batch
int bgp_ensure_warp_buffers(bgp_ctx* c) {
  const size_t need = (size_t)c->max_batch * c->n * c->d;
  if (need > c->cap_xwb) {
    free_dev(c->dXwB);
    c->dXwB = nullptr;
    c->cap_xwb = 0;
    BGP_HIP(hipMalloc(&c->dXwB, need * sizeof(double)));
    c->cap_xwb = need;
  }
  if (!c->dwarpB) BGP_HIP(hipMalloc(&c->dwarpB, (size_t)c->max_batch * 2 * c->d * sizeof(double)));
  return BGP_OK;
}

static int lml_batch_run(bgp_ctx* c, int B, const double* h, const double* warp, double* lml, int* status,
                         hipEvent_t& e0, hipEvent_t& e1, int defer_sync = 0) {
  if (B == 0) return BGP_OK;
  BGP_HIP(hipSetDevice(c->device));
  if (warp) {
    const int rcw = bgp_ensure_warp_buffers(c);
    if (rcw) return rcw;
  }
  const size_t p = c->d + 2;
  for (int k = 0; k < 6; k++) c->t_ms[k] = 0.0;
  for (int k = 0; k < 6; k++) c->t_cnt[k] = 0;
  for (int off = 0; off < B; off += c->max_batch) {
    const int nb = std::min(c->max_batch, B - off);
    if (c->timing) {
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, c->stream);
    }
    // (the submit path hands over the context's own pinned blocks; a caller's pageable block goes through the arena)
    const bool own = (h == c->hh);
    if (own)
      BGP_HIP(hipMemcpyAsync(c->dh, h + (size_t)off * p, nb * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    else
      BGP_HIP(bgp_memcpy_async(c->dh, h + (size_t)off * p, nb * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (warp)
      BGP_HIP(bgp_memcpy_async(c->dwarpB, warp + (size_t)off * 2 * c->d, (size_t)nb * 2 * c->d * sizeof(double),
                               hipMemcpyHostToDevice, c->stream));
    {
      const int rc = bgp_lml_enqueue_dev(c, nb, warp ? 1 : 0);
      if (rc) return rc;
    }
    if (own) {
      BGP_HIP(hipMemcpyAsync(lml + off, c->dlml, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      if (status) BGP_HIP(hipMemcpyAsync(status + off, c->dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    } else {
      BGP_HIP(bgp_memcpy_async(lml + off, c->dlml, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      if (status) BGP_HIP(bgp_memcpy_async(status + off, c->dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    if (c->timing) (void)hipEventRecord(e1, c->stream);
    if (defer_sync) return BGP_OK;  // (single chunk, timing off: bgp_lml_batch_wait synchronises)
    BGP_HIP(bgp_stream_sync(c->stream));
    if (ps_check(c)) {
      off -= c->max_batch;  // redo this chunk (ps_disabled is set: the multi-launch path takes it)
      continue;
    }
    bgp_tcollect(c);
    if (c->timing) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      c->t_ms[4] += ms;
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
      e0 = e1 = nullptr;
    }
  }
  return BGP_OK;
}

// One exit path: whatever step failed, the context's streams are drained (launches of the other walker group may
// still be in flight), the per-launch timing events of the call are released and the sticky HIP error is cleared, so
// that the context stays usable and nothing leaks.
static int lml_batch_impl(bgp_ctx* c, int B, const double* h, const double* warp, double* lml, int* status) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const int rc = lml_batch_run(c, B, h, warp, lml, status, e0, e1);
  if (rc != BGP_OK) {
    (void)hipStreamSynchronize(c->stream);
    for (int g = 0; g < BGP_MAX_STREAMS; g++)
      if (c->gstream[g]) (void)hipStreamSynchronize(c->gstream[g]);
    for (hipEvent_t ev : c->ev) (void)hipEventDestroy(ev);
    c->ev.clear();
    c->evcat.clear();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipGetLastError();
  }
  return rc;
}

// Asynchronous form of bgp_lml_batch for the sampler's inner loop: submit enqueues the whole half-step (upload of the
// proposals, K-build, factorisation, download of the B log-likelihoods into pinned host memory) and returns; the
// host evaluates the log-priors of the same proposals meanwhile and collects the device results with wait.
static int lml_submit_impl(bgp_ctx* c, int B, const double* h, const double* warp, const char* who) {
  if (!c || !h || B <= 0) {
    bgp_set_error("%s: bad argument", who);
    return BGP_ERR_INVALID;
  }
  if (c->pending_B != 0) {
    bgp_set_error("%s: a submitted batch is still pending (call bgp_lml_batch_wait)", who);
    return BGP_ERR_STATE;
  }
  if (c->timing) {  // per-launch timing synchronises inside the call: not an argument error, a mode (BGP_ERR_STATE)
    bgp_set_error("%s: per-launch timing is on (bgp_set_timing): use bgp_lml_batch", who);
    return BGP_ERR_STATE;
  }
  if (B > c->max_batch) {
    bgp_set_error("%s: B = %d exceeds max_batch = %d: use bgp_lml_batch", who, B, c->max_batch);
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  if ((size_t)B > c->cap_pinned) {
    if (c->hlml) (void)hipHostFree(c->hlml);
    if (c->hstatus) (void)hipHostFree(c->hstatus);
    if (c->hh) (void)hipHostFree(c->hh);
    c->hlml = nullptr;
    c->hstatus = nullptr;
    c->hh = nullptr;
    c->cap_pinned = 0;
    BGP_HIP(hipHostMalloc(&c->hlml, (size_t)c->max_batch * sizeof(double)));
    BGP_HIP(hipHostMalloc(&c->hstatus, (size_t)c->max_batch * sizeof(int)));
    BGP_HIP(hipHostMalloc(&c->hh, (size_t)c->max_batch * (c->d + 2) * sizeof(double)));
    c->cap_pinned = c->max_batch;
  }
  if (warp && !c->hwarp) BGP_HIP(hipHostMalloc(&c->hwarp, (size_t)c->max_batch * 2 * c->d * sizeof(double)));
  // the proposals go up from pinned memory too: an asynchronous copy out of a pageable numpy array costs the runtime
  // a page lock / unlock per call (tens of microseconds in front of every half-step)
  memcpy(c->hh, h, (size_t)B * (c->d + 2) * sizeof(double));
  if (warp) memcpy(c->hwarp, warp, (size_t)B * 2 * c->d * sizeof(double));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const int rc = lml_batch_run(c, B, c->hh, warp ? c->hwarp : nullptr, c->hlml, c->hstatus, e0, e1, 1);
  if (rc != BGP_OK) {
    (void)hipStreamSynchronize(c->stream);
    for (int g = 0; g < BGP_MAX_STREAMS; g++)
      if (c->gstream[g]) (void)hipStreamSynchronize(c->gstream[g]);
    (void)hipGetLastError();
    return rc;
  }
  c->pending_B = B;
  c->pending_warped = warp ? 1 : 0;
  return BGP_OK;
}

extern "C" int bgp_lml_batch_submit(bgp_ctx* c, int B, const double* h) {
  return lml_submit_impl(c, B, h, nullptr, "bgp_lml_batch_submit");
}

// The same for walkers that carry their own input warp (bask/bayesgpr.py:353-365): the (B, 2d) Beta parameters go up
// through pinned memory with the hyper-parameters; bgp_lml_batch_wait collects.
extern "C" int bgp_lml_batch_warped_submit(bgp_ctx* c, int B, const double* h, const double* warp) {
  if (!warp) {
    bgp_set_error("bgp_lml_batch_warped_submit: NULL warp");
    return BGP_ERR_INVALID;
  }
  return lml_submit_impl(c, B, h, warp, "bgp_lml_batch_warped_submit");
}

extern "C" int bgp_lml_batch_wait(bgp_ctx* c, double* lml, int* status) {
  if (!c || !lml) {
    bgp_set_error("bgp_lml_batch_wait: bad argument");
    return BGP_ERR_INVALID;
  }
  if (c->pending_B <= 0) {  // (< 0: a device-resident sampler run is open, not a submitted batch)
    bgp_set_error("bgp_lml_batch_wait: nothing submitted");
    return BGP_ERR_STATE;
  }
  const int B = c->pending_B;
  c->pending_B = 0;
  BGP_HIP(hipSetDevice(c->device));
  BGP_HIP(bgp_stream_sync(c->stream));
  if (ps_check(c)) {  // redo on the multi-launch path (the submitted block is still in the pinned buffers)
    const double* wp = c->pending_warped ? c->hwarp : nullptr;
    return lml_batch_impl(c, B, c->hh, wp, lml, status);
  }
  memcpy(lml, c->hlml, (size_t)B * sizeof(double));
  if (status) memcpy(status, c->hstatus, (size_t)B * sizeof(int));
  return BGP_OK;
}

// For collectors outside this file (bgp_lml_batch_wait_allgather): the context's stream has been synchronised; if the
// pending batch ran on the launch-free path and timed out, redo it (device-resident results included) on the
// multi-launch path.
int bgp_lml_redo_if_abandoned(bgp_ctx* c, int B) {
  if (!ps_check(c)) return BGP_OK;
  return lml_batch_impl(c, B, c->hh, c->pending_warped ? c->hwarp : nullptr, c->hlml, c->hstatus);
}

extern "C" int bgp_kernel_matrix(bgp_ctx* c, const double* h, double* K) {
  BGP_REQUIRE_IDLE(c, "bgp_kernel_matrix");
  if (!c || !h || !K) {
    bgp_set_error("bgp_kernel_matrix: NULL argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  int rc = factor_chunk(c, 1, h, 1);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy2d_async(K, (size_t)c->n * sizeof(double), c->dK, (size_t)c->npad * sizeof(double),
                           (size_t)c->n * sizeof(double), c->n, hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}

// Debugging aid: the working matrix (npad x npad, row-major: the factor L in its lower triangle after an LML call) and
// the working right-hand side (z = L^-1 y) of batch slot b, as the last bgp_lml_batch call left them.
extern "C" int bgp_debug_workspace(bgp_ctx* c, int b, double* Lout, double* zout) {
  BGP_REQUIRE_IDLE(c, "bgp_debug_workspace");
  if (!c || b < 0 || b >= c->max_batch) {
    bgp_set_error("bgp_debug_workspace: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  const size_t np = c->npad;
  if (Lout) BGP_HIP(hipMemcpy(Lout, c->dK + (size_t)b * np * np, np * np * sizeof(double), hipMemcpyDeviceToHost));
  if (zout) BGP_HIP(hipMemcpy(zout, c->dyw + (size_t)b * np, np * sizeof(double), hipMemcpyDeviceToHost));
  return BGP_OK;
}

// The factor sample_y's last call left in the child workspace (mpad x mpad doubles, L in the lower triangle; the strict upper
// triangle is whatever the covariance build left there).
extern "C" int bgp_debug_cov_factor(bgp_ctx* c, int* mpad, double* Lout) {
  BGP_REQUIRE_IDLE(c, "bgp_debug_cov_factor");
  if (!c || !mpad) {
    bgp_set_error("bgp_debug_cov_factor: bad argument");
    return BGP_ERR_INVALID;
  }
  *mpad = c->child ? c->child->npad : 0;
  if (!Lout || !c->child) return BGP_OK;
  BGP_HIP(hipSetDevice(c->device));
  const size_t np = c->child->npad;
  BGP_HIP(hipMemcpy(Lout, c->child->dK, np * np * sizeof(double), hipMemcpyDeviceToHost));
  return BGP_OK;
}

extern "C" int bgp_device_synchronize(int device) {
  BGP_HIP(hipSetDevice(device));
  BGP_HIP(hipDeviceSynchronize());
  return BGP_OK;
}

extern "C" int bgp_set_streams(bgp_ctx* c, int nstreams) {
  if (!c || nstreams < 1 || nstreams > BGP_MAX_STREAMS) {
    bgp_set_error("bgp_set_streams: nstreams must be in 1..%d", BGP_MAX_STREAMS);
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_set_streams");
  BGP_HIP(hipSetDevice(c->device));
  for (int g = 0; g < nstreams; g++) {
    if (!c->gstream[g]) {
      BGP_HIP(hipStreamCreate(&c->gstream[g]));
      BGP_HIP(hipEventCreateWithFlags(&c->ev_done[g], hipEventDisableTiming));
    }
  }
  c->nstreams = nstreams;
  c->streams_auto = 0;
  return BGP_OK;
}

// Launch-free factorisation of small batches: -1 = as the environment says (BGP_PERSIST; unset:
// off), 0 = never, 1 = whenever the batch fits (at most 64 matrices, at least two block columns).
// The look-ahead column launches of the trailing update (K = 128 .. 128 (P-1) on one 128-wide block column) inside the
// "syrk" figure of bgp_last_timing: their time and count, so that a caller can rate bulk and column launches apart.
extern "C" int bgp_last_timing_columns(bgp_ctx* c, double* ms, int* launches) {
  if (!c || !ms) return BGP_ERR_INVALID;
  *ms = c->t_ms[5];
  if (launches) *launches = c->t_cnt[5];
  return BGP_OK;
}

extern "C" int bgp_set_persist(bgp_ctx* c, int mode) {
  if (!c || mode < -1 || mode > 1) {
    bgp_set_error("bgp_set_persist: mode must be -1, 0 or 1");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_set_persist");
  if (mode == -1) {
    const char* envps = getenv("BGP_PERSIST");
    c->persist = envps ? (atoi(envps) != 0 ? 1 : 0) : -1;
  } else {
    c->persist = mode;
  }
  if (mode == 1) c->ps_disabled = c->ps_cooldown = 0;
  return BGP_OK;
}

extern "C" int bgp_set_timing(bgp_ctx* c, int enable) {
  if (!c) return BGP_ERR_INVALID;
  BGP_REQUIRE_IDLE(c, "bgp_set_timing");
  c->timing = enable ? 1 : 0;
  return BGP_OK;
}

extern "C" int bgp_last_timing(bgp_ctx* c, double* out_ms, int* counts) {
  if (!c || !out_ms) return BGP_ERR_INVALID;
  for (int k = 0; k < 5; k++) out_ms[k] = c->t_ms[k];
  if (counts)
    for (int k = 0; k < 4; k++) counts[k] = c->t_cnt[k];
  return BGP_OK;
}
