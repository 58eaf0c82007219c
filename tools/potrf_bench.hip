// potrf_kernel micro-benchmark + phase trace (build: see tools/build_potrf_bench.sh; links the library objects
// with bgp_chol.hip recompiled under -DPF_TRACE).  Usage: potrf_bench [B=32] [reps=20]
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bgp_common.h"

extern "C" int bgp_debug_potrf_trace(unsigned long long* out);
extern "C" int bgp_debug_potrf_trace_w(unsigned long long* out);
void bgp_launch_potrf(bgp_ctx* ctx, hipStream_t st, int B, double* dK, double* dW, double* dyw, double* dacc,
                      double* dlml, int* dstatus, int ld, size_t mstride, int ystride, int k);

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, reps = argc > 2 ? atoi(argv[2]) : 20;
  const int n = 128;
  std::vector<double> K((size_t)B * n * n), y((size_t)B * n);
  srand(1);
  for (int b = 0; b < B; b++) {
    std::vector<double> G(n * n);
    for (auto& g : G) g = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = (i == j) ? 1.0 : 0.0;
        for (int t = 0; t < n; t++) s += G[i * n + t] * G[j * n + t] / n;
        K[(size_t)b * n * n + i * n + j] = s;
      }
    for (int i = 0; i < n; i++) y[(size_t)b * n + i] = (double)rand() / RAND_MAX;
  }
  double *dK, *dK0, *dW, *dy, *dy0, *dacc, *dlml;
  int* dst;
  hipMalloc(&dK, K.size() * 8); hipMalloc(&dK0, K.size() * 8); hipMalloc(&dW, K.size() * 8);
  hipMalloc(&dy, y.size() * 8); hipMalloc(&dy0, y.size() * 8); hipMalloc(&dacc, B * 4 * 8); hipMalloc(&dlml, B * 8);
  hipMalloc(&dst, B * 4);
  hipMemcpy(dK0, K.data(), K.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(dy0, y.data(), y.size() * 8, hipMemcpyHostToDevice);
  hipMemset(dst, 0, B * 4);
  bgp_ctx ctx;
  ctx.n = n; ctx.nblk = 1; ctx.npad = n;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0.f;
  for (int r = 0; r < reps + 3; r++) {
    hipMemcpy(dK, dK0, K.size() * 8, hipMemcpyDeviceToDevice);
    hipMemcpy(dy, dy0, y.size() * 8, hipMemcpyDeviceToDevice);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    bgp_launch_potrf(&ctx, 0, B, dK, dW, dy, dacc, dlml, dst, n, (size_t)n * n, n, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (r >= 3) tot += ms;
  }
  std::vector<double> lml(B);
  hipMemcpy(lml.data(), dlml, B * 8, hipMemcpyDeviceToHost);
  // host check of matrix 0: -0.5 z'z - sum log L_ii - n/2 log 2pi via plain Cholesky
  {
    std::vector<double> L(K.begin(), K.begin() + n * n), z(y.begin(), y.begin() + n);
    double ld = 0.0;
    for (int j = 0; j < n; j++) {
      for (int t = 0; t < j; t++) L[j * n + j] -= L[j * n + t] * L[j * n + t];
      L[j * n + j] = std::sqrt(L[j * n + j]); ld += std::log(L[j * n + j]);
      for (int i = j + 1; i < n; i++) {
        for (int t = 0; t < j; t++) L[i * n + j] -= L[i * n + t] * L[j * n + t];
        L[i * n + j] /= L[j * n + j];
      }
    }
    double zz = 0.0;
    for (int i = 0; i < n; i++) { for (int t = 0; t < i; t++) z[i] -= L[i * n + t] * z[t]; z[i] /= L[i * n + i]; zz += z[i] * z[i]; }
    double ref = -0.5 * zz - ld - 0.5 * n * 1.8378770664093453;
    printf("lml[0] device %.15g host %.15g rel %.2e\n", lml[0], ref, std::fabs(lml[0] - ref) / std::fabs(ref));
  }
  printf("B=%d potrf event time %.2f us/launch\n", B, tot / reps * 1e3);
  unsigned long long tr[32];
  if (bgp_debug_potrf_trace(tr) == 0) {
    auto us = [&](int a, int b) { return (double)(tr[b] - tr[a]) * 0.01; };
    printf("trace (us): load %.2f | loop %.2f | Lstore+logdet %.2f | W %.2f | tail %.2f | total %.2f\n", us(0, 1),
           us(1, 23), us(23, 26), us(26, 27), us(27, 28), us(0, 28));
    printf("per-step:");
    for (int sb = 0; sb < 8; sb++) printf(" %.2f", us(sb == 0 ? 1 : 2 + (sb - 1) * 3, 2 + sb * 3));
    printf("\n  pre-factor:");
    for (int sb = 0; sb < 8; sb++) printf(" %.2f", us(sb == 0 ? 1 : 2 + (sb - 1) * 3, 3 + sb * 3));
    printf("\n  factor:");
    for (int sb = 0; sb < 8; sb++) printf(" %.2f", us(3 + sb * 3, 4 + sb * 3));
    printf("\n  post-factor+barrier:");
    for (int sb = 0; sb < 8; sb++) printf(" %.2f", us(4 + sb * 3, 2 + sb * 3));
    printf("\n");
    unsigned long long tw[8 * 32];
    if (bgp_debug_potrf_trace_w(tw) == 0) {
      // per wave and step: when the wave's step work ended, relative to the step's start on wave 0 (us)
      for (int w = 0; w < 8; w++) {
        printf("  wave %d work (us):", w);
        for (int sb = 0; sb < 8; sb++) printf(" %.2f", (double)(tw[w * 32 + 2 * sb + 1] - tw[2 * sb]) * 0.01);
        printf("\n");
      }
    }
  }
  return 0;
}
