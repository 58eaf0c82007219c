// MFMA + LDS operand-feed efficiency of the tile-GEMM inner loops, operands resident in LDS (no global
// loads, no barriers inside the loop).  hipcc --offload-arch=gfx950 -O3 -I bayes-skopt_amd/csrc tools/g8_loop_probe.hip
#include "bgp_gemm8.h"
#include <cstdio>
void bgp_set_error(const char*, ...) {}

template <int NR, int NC, int THREADS, int F444>
__global__ void __launch_bounds__(THREADS) loopk(double* out, int iters) {
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 128 * GK_LD; i += THREADS) { sm.A[i] = 1e-3 * (i % 97); sm.B[i] = 1e-3 * (i % 89); }
  __syncthreads();
  constexpr int WR = 128 / (NR * 16);          // waves along rows
  const int r0 = (w % WR) * NR * 16, c0 = (w / WR) * NC * 16;
  d4 acc[NR][NC];
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it++) {
    if (F444) g8_mma_block<NR, NC, 1, 0, -64>(sm.A, sm.B, acc, r0, c0, lane, 0);
    else gk_mma_block<NR, NC, 1, 0, -64>(sm.A, sm.B, acc, r0, c0, lane, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1.2345) out[0] = s;
}

template <int NR, int NC, int THREADS, int F444>
void run(const char* name, int wgs_per_cu, int iters) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * wgs_per_cu;
  hipLaunchKernelGGL((loopk<NR, NC, THREADS, F444>), dim3(blocks), dim3(THREADS), 0, 0, d, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((loopk<NR, NC, THREADS, F444>), dim3(blocks), dim3(THREADS), 0, 0, d, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * iters * 2.0 * 128 * 128 * 32;
  printf("%-44s wgs/CU=%d : %7.2f TF\n", name, wgs_per_cu, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  run<2, 4, 512, 1>("4x4x4  8 waves 32x64/wave", 1, 2000);
  run<2, 4, 512, 1>("4x4x4  8 waves 32x64/wave", 2, 2000);
  run<4, 4, 256, 1>("4x4x4  4 waves 64x64/wave", 1, 2000);
  run<4, 4, 256, 1>("4x4x4  4 waves 64x64/wave", 2, 2000);
  run<1, 8, 512, 1>("4x4x4  8 waves 16x128/wave", 1, 2000);
  run<4, 2, 512, 1>("4x4x4  8 waves 64x32/wave", 1, 2000);
  run<4, 4, 256, 0>("16x16x4 4 waves 64x64/wave", 1, 2000);
  run<4, 4, 256, 0>("16x16x4 4 waves 64x64/wave", 2, 2000);
  return 0;
}
