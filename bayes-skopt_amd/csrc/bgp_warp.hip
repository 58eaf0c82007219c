// Input warping: every input column is passed through a Beta CDF whose two
// parameters are inferred together with the kernel hyper-parameters (Snoek et al. 2014;
// bask/bayesgpr.py:249-316,353-365).  In the MCMC every walker therefore sees its OWN warped design
// matrix: Xw[b][i][k] = I_{x_ik}(exp(wa_bk), exp(wb_bk)) (regularised incomplete beta function =
// scipy.stats.beta(a, b).cdf, bask/bayesgpr.py:312-316), computed here on the device and consumed by
// the K-build through a per-walker X stride.
#include "bgp_common.h"

// Continued fraction of the incomplete beta function (modified Lentz), relative accuracy ~1e-15.
static __device__ double bgp_betacf(double a, double b, double x) {
  const double FPMIN = 1e-300, EPS = 1e-16;
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < FPMIN) d = FPMIN;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 1000; m++) {
    const double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c;
    if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d;
    if (fabs(d) < FPMIN) d = FPMIN;
    c = 1.0 + aa / c;
    if (fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    const double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < EPS) break;
  }
  return h;
}

static __device__ double bgp_betainc(double a, double b, double x) {
  if (!(x > 0.0)) return 0.0;  // cdf(x <= 0) = 0
  if (x >= 1.0) return 1.0;
  const double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log1p(-x));
  if (x < (a + 1.0) / (a + b + 2.0)) return bt * bgp_betacf(a, b, x) / a;
  return 1.0 - bt * bgp_betacf(b, a, 1.0 - x) / b;
}

// out[b][i][k] = I_{X[i][k]}(exp(W[b][k]), exp(W[b][d + k]));  W is (B, 2d) log-space parameters.
__global__ void __launch_bounds__(256) warp_kernel(const double* __restrict__ X, const double* __restrict__ W,
                                                    double* __restrict__ out, int n, int d, size_t ostride) {
  const int b = blockIdx.y;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)n * d) return;
  const int k = (int)(idx % d);
  const double a = exp(W[(size_t)b * 2 * d + k]), bb = exp(W[(size_t)b * 2 * d + d + k]);
  out[(size_t)b * ostride + idx] = bgp_betainc(a, bb, X[idx]);
}

int bgp_launch_warp(bgp_ctx* c, hipStream_t st, const double* dX, const double* dW, double* dout, int n, int B,
                    size_t ostride) {
  const size_t tot = (size_t)n * c->d;
  hipLaunchKernelGGL(warp_kernel, dim3((unsigned)((tot + 255) / 256), B), dim3(256), 0, st, dX, dW, dout, n, c->d,
                     ostride);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

// Context-level warp: subsequent posterior / predict / pvrs / gradient / sample_y / un-warped LML calls
// see the training inputs (and their query points) through this warp.  warp == NULL clears it.
extern "C" int bgp_ctx_set_warp(bgp_ctx* c, const double* warp) {
  BGP_REQUIRE_IDLE(c, "bgp_ctx_set_warp");
  if (!c) {
    bgp_set_error("bgp_ctx_set_warp: NULL ctx");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  if (!warp) {
    c->has_warp = 0;
    c->dXeff = c->dX;
    return BGP_OK;
  }
  const size_t nd = (size_t)c->cap_n * c->d;
  if (!c->dXw1) BGP_HIP(hipMalloc(&c->dXw1, nd * sizeof(double)));
  if (!c->dwarp) BGP_HIP(hipMalloc(&c->dwarp, 2 * (size_t)c->d * sizeof(double)));
  BGP_HIP(bgp_memcpy_async(c->dwarp, warp, 2 * (size_t)c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = bgp_launch_warp(c, c->stream, c->dX, c->dwarp, c->dXw1, c->n, 1, 0);
  if (rc) return rc;
  BGP_HIP(bgp_stream_sync(c->stream));
  c->has_warp = 1;
  c->dXeff = c->dXw1;
  return BGP_OK;
}

// Utility: Beta-CDF warp of m points on the device (host-side BayesGPR.warp()).
extern "C" int bgp_beta_cdf(bgp_ctx* c, int m, const double* X, const double* warp, double* out) {
  if (!c || !X || !warp || !out || m <= 0) {
    bgp_set_error("bgp_beta_cdf: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_REQUIRE_IDLE(c, "bgp_beta_cdf");
  BGP_HIP(hipSetDevice(c->device));
  const size_t md = (size_t)m * c->d;
  int rc = bgp_ensure_scratch(c, 2 * md + 2 * (size_t)c->d + 8);
  if (rc) return rc;
  double* dXi = c->dscratch;
  double* dXo = c->dscratch + md;
  double* dW = c->dscratch + 2 * md;
  BGP_HIP(bgp_memcpy_async(dXi, X, md * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(bgp_memcpy_async(dW, warp, 2 * (size_t)c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  rc = bgp_launch_warp(c, c->stream, dXi, dW, dXo, m, 1, 0);
  if (rc) return rc;
  BGP_HIP(bgp_memcpy_async(out, dXo, md * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(bgp_stream_sync(c->stream));
  return BGP_OK;
}
