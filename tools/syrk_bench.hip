// syrk2 (trailing update) micro-benchmark: one rest update of BASELINE config C (n = 2048, 128 matrices,
// panels 0..1, K = 256) on synthetic data.  Build: tools/build_syrk_bench.sh [-DSYRK_EXP=n].
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" void bgp_debug_launch_syrk2(hipStream_t st, int B8, int ntile, double* dK, const int* dstatus, int ld,
                                       size_t mstride, int nblk, int kp, int K, int jstart, int colmode, int B);
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 2048, reps = 10;
  const int K = argc > 3 ? atoi(argv[3]) : 256;
  const int nblk = n / 128, np = K / 128;
  const size_t ms = (size_t)n * n;
  double* dK; int* dst;
  hipMalloc(&dK, ms * B * 8); hipMalloc(&dst, B * 4);
  hipMemset(dst, 0, B * 4);
  std::vector<double> h(ms);
  for (size_t i = 0; i < ms; i++) h[i] = 1e-3 * ((double)rand() / RAND_MAX - 0.5);
  for (int b = 0; b < B; b++) hipMemcpy(dK + b * ms, h.data(), ms * 8, hipMemcpyHostToDevice);
  const int nt = nblk - np, ntile = nt * (nt + 1) / 2, B8 = 8 * ((B + 7) / 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; r++) bgp_debug_launch_syrk2(0, B8, ntile, dK, dst, n, ms, nblk, 0, K, np, 0, B);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) bgp_debug_launch_syrk2(0, B8, ntile, dK, dst, n, ms, nblk, 0, K, np, 0, B);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float msec; hipEventElapsedTime(&msec, e0, e1);
  const double m = (double)(n - 128 * np), flops = (double)K * m * (m + 1) * B;
  printf("B=%d n=%d K=%d tiles=%d: %.3f ms/launch, %.1f TF (algorithmic)\n", B, n, K, ntile, msec / reps,
         flops / (msec / reps * 1e-3) / 1e12);
  return 0;
}
