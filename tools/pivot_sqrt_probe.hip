// Accuracy of the pivot root of the diagonal-block factorisation (bgp_pf.h, micro_chol_inv): sqrt(x) and 1/sqrt(x) from the
// hardware seed v_rsq_f64.  Variants:
//   0  seed only (what the hardware gives)
//   1  round 1-3: one coupled Goldschmidt step (dj = g + g r, inv = 2 (h + h r))
//   2  the same with inv = y0 + y0 r (no doubling on the chain)
//   3  variant 2 + residual corrections: dj += (x - dj^2) * inv/2,  inv += inv * (1 - dj * inv)
// Reports the maximal error in ulps of the correctly rounded results (host long double) over N log-uniform arguments in
// [1e-12, 1e6], and the time of a dependent chain of each variant (ns per root at one wave).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/pivot_sqrt_probe.hip -o /tmp/psp && /tmp/psp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int V>
static __device__ __forceinline__ void root(double x, double& dj, double& inv) {
#pragma clang fp contract(off)
  const double y0 = __builtin_amdgcn_rsq(x);
  if (V == 0) {
    dj = x * y0;
    inv = y0;
    return;
  }
  const double g = x * y0, h = 0.5 * y0;
  const double r = __builtin_fma(-g, h, 0.5);
  dj = __builtin_fma(g, r, g);
  if (V == 1) {
    const double hh = __builtin_fma(h, r, h);
    inv = hh + hh;
    return;
  }
  inv = __builtin_fma(y0, r, y0);
  if (V == 2) return;
  const double hy = 0.5 * inv;
  const double d = __builtin_fma(-dj, dj, x);
  dj = __builtin_fma(d, hy, dj);
  const double e = __builtin_fma(-dj, inv, 1.0);
  inv = __builtin_fma(inv, e, inv);
}

template <int V>
__global__ void eval_kernel(const double* x, double* s, double* iv, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a, b;
  root<V>(x[i], a, b);
  s[i] = a;
  iv[i] = b;
}

template <int V>
__global__ void chain_kernel(double* out, int iters, double x0) {
  double x = x0 + threadIdx.x * 1e-6, acc = 0.0;
  for (int i = 0; i < iters; i++) {
    double a, b;
    root<V>(x, a, b);
    acc += a;
    x = x * b * 1.0000001 + 1.0;  // the next argument depends on this root (as the next pivot does)
  }
  if (acc == 123.456) out[0] = acc;
}

static double ulps(double got, long double ref) {
  const double r = (double)ref;
  const double u = std::nextafter(std::fabs(r), INFINITY) - std::fabs(r);
  return (double)(std::fabs((long double)got - ref) / (long double)u);
}

template <int V>
static void run(const std::vector<double>& hx, double* dx, double* ds, double* di) {
  const int n = (int)hx.size();
  hipLaunchKernelGGL(eval_kernel<V>, dim3((n + 255) / 256), dim3(256), 0, 0, dx, ds, di, n);
  std::vector<double> s(n), iv(n);
  hipMemcpy(s.data(), ds, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(iv.data(), di, n * 8, hipMemcpyDeviceToHost);
  double ms = 0, mi = 0;
  long exact_s = 0, exact_i = 0;
  for (int i = 0; i < n; i++) {
    const long double rs = sqrtl((long double)hx[i]), ri = 1.0L / rs;
    const double us = ulps(s[i], rs), ui = ulps(iv[i], ri);
    ms = us > ms ? us : ms;
    mi = ui > mi ? ui : mi;
    exact_s += (s[i] == (double)rs);
    exact_i += (iv[i] == (double)ri);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double* dout;
  hipMalloc(&dout, 8);
  const int iters = 200000;
  hipLaunchKernelGGL(chain_kernel<V>, dim3(1), dim3(64), 0, 0, dout, 1000, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(chain_kernel<V>, dim3(1), dim3(64), 0, 0, dout, iters, 2.0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float t;
  hipEventElapsedTime(&t, e0, e1);
  printf("variant %d: sqrt max %.3f ulp (%.2f %% correctly rounded)   1/sqrt max %.3f ulp (%.2f %% correctly rounded)   chain %.1f ns per root\n",
         V, ms, 100.0 * exact_s / n, mi, 100.0 * exact_i / n, 1e6 * t / iters);
  hipFree(dout);
}

int main() {
  const int n = 10'000'000;
  std::vector<double> hx(n);
  srand48(1);
  for (int i = 0; i < n; i++) hx[i] = std::pow(10.0, -12.0 + 18.0 * drand48());
  double *dx, *ds, *di;
  hipMalloc(&dx, (size_t)n * 8);
  hipMalloc(&ds, (size_t)n * 8);
  hipMalloc(&di, (size_t)n * 8);
  hipMemcpy(dx, hx.data(), (size_t)n * 8, hipMemcpyHostToDevice);
  run<0>(hx, dx, ds, di);
  run<1>(hx, dx, ds, di);
  run<2>(hx, dx, ds, di);
  run<3>(hx, dx, ds, di);
  return 0;
}
