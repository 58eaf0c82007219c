// Inner-loop variants of the 128x128 tile update on v_mfma_f64_16x16x4_f64, operands resident in LDS
// (no global loads, no barriers): which fragment-read pattern keeps the MFMA pipe fed?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I bayes-skopt_amd/csrc tools/syrk_loop_probe.hip
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include "bgp_gemm.h"
#include <cstdio>
void bgp_set_error(const char*, ...) {}

// V0: gk_mma_block as shipped (NEG=1).  V1: NEG=0.  V2: fragments of k-step kk+1 read before the MFMAs of kk.
// V3: k re-mapped so that a lane's 8 k-steps are 8 CONSECUTIVE doubles (k = 8*lk + kk): 4 x ds_read_b128
//     per fragment row and chunk instead of 8 x ds_read_b64; all fragments of the chunk read up front.
// V4: V3 but read in two halves (k-steps 0..3 first, 4..7 while the first half multiplies).
template <int V>
static __device__ __forceinline__ void mma_variant(const double* __restrict__ As, const double* __restrict__ Bs,
                                                   d4 (&acc)[4][4], int r0, int c0, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  if (V == 0) { gk_mma_block<4, 4, 1, 0, -64>(As, Bs, acc, r0, c0, lane, 0); return; }
  if (V == 1) { gk_mma_block<4, 4, 0, 0, -64>(As, Bs, acc, r0, c0, lane, 0); return; }
  if (V == 2) {
    double a[4], b[4], an[4], bn[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = As[(r0 + i * 16 + lr) * GK_LD + lk]; b[i] = Bs[(c0 + i * 16 + lr) * GK_LD + lk]; }
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      if (kk < 7) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          an[i] = As[(r0 + i * 16 + lr) * GK_LD + (kk + 1) * 4 + lk];
          bn[i] = Bs[(c0 + i * 16 + lr) * GK_LD + (kk + 1) * 4 + lk];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++) { a[i] = an[i]; b[i] = bn[i]; }
    }
    return;
  }
  if (V == 3 || V == 4) {
    d2 a[4][4], b[4][4];  // [block][quarter]: doubles 2q, 2q+1 of the lane's 8 consecutive k
    const int h0 = (V == 4) ? 2 : 4;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int q = 0; q < h0; q++) {
        a[i][q] = *reinterpret_cast<const d2*>(&As[(r0 + i * 16 + lr) * GK_LD + lk * 8 + 2 * q]);
        b[i][q] = *reinterpret_cast<const d2*>(&Bs[(c0 + i * 16 + lr) * GK_LD + lk * 8 + 2 * q]);
      }
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      if (V == 4 && kk == 1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int q = 2; q < 4; q++) {
            a[i][q] = *reinterpret_cast<const d2*>(&As[(r0 + i * 16 + lr) * GK_LD + lk * 8 + 2 * q]);
            b[i][q] = *reinterpret_cast<const d2*>(&Bs[(c0 + i * 16 + lr) * GK_LD + lk * 8 + 2 * q]);
          }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 4; i++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][kk >> 1][kk & 1], b[j][kk >> 1][kk & 1], acc[i][j], 0, 0, 0);
    }
    return;
  }
}

template <int V>
__global__ void __launch_bounds__(256, 2) loopk(double* out, int iters) {
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  for (int i = tid; i < 128 * GK_LD; i += 256) { sm.A[i] = 1e-3 * (i % 97); sm.B[i] = 1e-3 * (i % 89); }
  __syncthreads();
  const int r0 = (w >> 1) * 64, c0 = (w & 1) * 64;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it++) mma_variant<V>(sm.A, sm.B, acc, r0, c0, lane);
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 1.2345) out[0] = s;
}

template <int V>
void run(const char* name, int wgs_per_cu, int iters) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wgs_per_cu;
  hipLaunchKernelGGL(loopk<V>, dim3(blocks), dim3(256), 0, 0, d, 4);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(loopk<V>, dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)blocks * iters * 2.0 * 128 * 128 * 32;
  printf("%-52s wgs/CU=%d : %7.2f TF\n", name, wgs_per_cu, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int wg = 1; wg <= 2; wg++) {
    run<0>("V0 shipped gk_mma_block (NEG)", wg, 2000);
    run<1>("V1 no negation", wg, 2000);
    run<2>("V2 next k-step's fragments read ahead", wg, 2000);
    run<3>("V3 k re-mapped: ds_read_b128, whole chunk up front", wg, 2000);
    run<4>("V4 k re-mapped: ds_read_b128, two halves", wg, 2000);
  }
  return 0;
}
