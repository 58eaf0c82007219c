// trsm8 (panel solve) micro-benchmark at BASELINE config C: step k of n = 2048, 128 matrices.
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
void bgp_launch_trsm8(hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus, int ld, size_t mstride,
                      int ystride, int nblk, int k);
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 2048, k = argc > 3 ? atoi(argv[3]) : 0;
  const int nblk = n / 128, reps = 10;
  const size_t ms = (size_t)n * n;
  double *dK, *dW, *dy; int* dst;
  hipMalloc(&dK, ms * B * 8); hipMalloc(&dW, (size_t)B * nblk * 128 * 128 * 8); hipMalloc(&dy, (size_t)B * n * 8);
  hipMalloc(&dst, B * 4); hipMemset(dst, 0, B * 4);
  hipMemset(dK, 0, ms * B * 8); hipMemset(dW, 0, (size_t)B * nblk * 128 * 128 * 8); hipMemset(dy, 0, (size_t)B * n * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int r = 0; r < 2; r++) bgp_launch_trsm8(0, B, dK, dW, dy, dst, n, ms, n, nblk, k);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; r++) bgp_launch_trsm8(0, B, dK, dW, dy, dst, n, ms, n, nblk, k);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float msec; hipEventElapsedTime(&msec, e0, e1);
  const double tiles = (double)(nblk - k - 1) * B, bytes = tiles * 2 * 128 * 128 * 8;
  printf("B=%d n=%d k=%d tiles=%.0f: %.1f us/launch, %.2f TB/s of tile traffic (A read + X write)\n", B, n, k, tiles,
         msec / reps * 1e3, bytes / (msec / reps * 1e-3) / 1e12);
  return 0;
}
