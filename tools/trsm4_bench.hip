// Panel-solve micro-benchmark: trsm4_kernel on one launch shape of the LML path with ablations (which resource binds
// it?).  Build: tools/build_syrk4_bench.sh builds this too.   trsm4_bench [B n k]   defaults 128 2048 0
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
extern "C" int bgp_debug_launch_trsm4(int var, hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus,
                                      int ld, size_t mstride, int ystride, int nblk, int k);
int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 128, n = argc > 2 ? atoi(argv[2]) : 2048, k = argc > 3 ? atoi(argv[3]) : 0;
  const int nblk = n / 128;
  const size_t ms = (size_t)n * n;
  double *dK, *dW, *dy;
  int* dst;
  hipMalloc(&dK, ms * B * 8);
  hipMalloc(&dW, (size_t)B * nblk * 128 * 128 * 8);
  hipMalloc(&dy, (size_t)B * n * 8);
  hipMalloc(&dst, B * 4);
  hipMemset(dst, 0, B * 4);
  std::vector<double> h(ms), w((size_t)nblk * 128 * 128, 0.0), y(n, 0.5);
  srand(1);
  for (auto& v : h) v = (double)rand() / RAND_MAX - 0.5;
  for (int b = 0; b < nblk; b++)
    for (int i = 0; i < 128; i++)
      for (int j = 0; j <= i; j++) w[(size_t)b * 16384 + i * 128 + j] = ((double)rand() / RAND_MAX - 0.5) * 0.1;
  for (int b = 0; b < B; b++) {
    hipMemcpy(dK + b * ms, h.data(), ms * 8, hipMemcpyHostToDevice);
    hipMemcpy(dW + (size_t)b * w.size(), w.data(), w.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dy + (size_t)b * n, y.data(), n * 8, hipMemcpyHostToDevice);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int vars[8] = {0, 1, 2, 3, 8, 16, 10, 11};
  const char* names[8] = {"full", "no LDS-DMA", "no MFMA", "no LDS-DMA, no MFMA", "no stores", "W staged for chunk 0 only",
                          "no MFMA, no stores (loads only)", "barriers + y update only"};
  const double rows = (double)(nblk - k - 1) * 128, bytes = rows * 128 * 8 * 2 * B;
  for (int round = 0; round < 3; round++)
    for (int v = 0; v < 8; v++) {
      float best = 1e9f;
      for (int r = 0; r < 5; r++) {
        hipEventRecord(e0, 0);
        bgp_debug_launch_trsm4(vars[v], 0, B, dK, dW, dy, dst, n, ms, n, nblk, k);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (t < best) best = t;
      }
      if (round == 2)
        printf("%-34s %8.1f us   (panel r+w %.0f MB -> %.2f TB/s if this were the full kernel)\n", names[v], best * 1e3,
               bytes / 1e6, bytes / (best * 1e-3) / 1e12);
    }
  printf("launch status: %s\n", hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
