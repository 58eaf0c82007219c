// What limits v_mfma_f64_16x16x4_f64 issue on MI355X?  Varies accumulators per wave, waves per SIMD,
// operand reuse, and tries the 4x4x4_4b form.  hipcc --offload-arch=gfx950 -O3 tools/mfma_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int DISTINCT_AB>
__global__ void __launch_bounds__(256) k16(double* out, int iters, double a0, long long* cyc) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[NACC], b[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = 0.5 + i; }
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++)
      acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(DISTINCT_AB ? a[i] : a[0], DISTINCT_AB ? b[i] : b[0], acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// NA x NB register block: NA A fragments x NB B fragments, MFMAs in row-major order of the block (A changes every NB)
template <int NA, int NB>
__global__ void __launch_bounds__(256) kblk(double* out, int iters, double a0) {
  d4 acc[NA * NB];
#pragma unroll
  for (int i = 0; i < NA * NB; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[NA], b[NB];
#pragma unroll
  for (int i = 0; i < NA; i++) a[i] = a0 + threadIdx.x * 1e-9 + i;
#pragma unroll
  for (int i = 0; i < NB; i++) b[i] = 0.5 + i;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ia = 0; ia < NA; ia++)
#pragma unroll
      for (int ib = 0; ib < NB; ib++)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[ia * NB + ib]) : "v"(a[ia]), "v"(b[ib]));
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NA * NB; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}
template <int NA, int NB>
void runblk(int wps, int iters) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * wps;
  hipLaunchKernelGGL((kblk<NA, NB>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((kblk<NA, NB>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)blocks * 4 * iters * NA * NB;
  printf("16x16x4 block %d A x %d B, waves/SIMD=%d : %7.2f TF\n", NA, NB, wps, nm * 2048 / ms / 1e9);
  hipFree(d);
}

template <int NACC>
__global__ void __launch_bounds__(256) k4(double* out, int iters, double a0) {
  double acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = 0.0;
  double a = a0 + threadIdx.x * 1e-9, b = 0.5;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i];
  if (s == 123.456) out[0] = s;
}

template <int NACC, int D>
void run16(int wps, int iters) {
  double* d; long long* c;
  hipMalloc(&d, 8); hipMalloc(&c, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * wps;
  hipLaunchKernelGGL((k16<NACC, D>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.0, c);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k16<NACC, D>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, c);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc; hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);
  double nm = (double)blocks * 4 * iters * NACC;
  printf("16x16x4 nacc=%2d distinctAB=%d waves/SIMD=%d : %7.2f TF   cycles/MFMA/SIMD (counter)=%.1f\n", NACC, D, wps,
         nm * 2048 / ms / 1e9, (double)hc / ((double)iters * NACC * wps));
  hipFree(d); hipFree(c);
}
template <int NACC>
void run4(int wps, int iters) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * wps;
  hipLaunchKernelGGL((k4<NACC>), dim3(blocks), dim3(256), 0, 0, d, 16, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k4<NACC>), dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nm = (double)blocks * 4 * iters * NACC;
  printf("4x4x4_4b nacc=%2d waves/SIMD=%d : %7.2f TF\n", NACC, wps, nm * 512 / ms / 1e9);
  hipFree(d);
}
int main() {
  run16<1, 0>(1, 20000); run16<2, 0>(1, 20000); run16<4, 0>(1, 20000); run16<8, 0>(1, 10000); run16<16, 0>(1, 5000);
  run16<4, 0>(2, 10000); run16<8, 0>(2, 10000); run16<16, 1>(1, 5000); run16<16, 1>(2, 5000); run16<8, 1>(4, 5000);
  runblk<1, 8>(1, 10000); runblk<1, 8>(2, 10000); runblk<2, 4>(1, 10000); runblk<2, 4>(2, 10000); runblk<4, 4>(1, 5000); runblk<4, 4>(2, 5000);
  runblk<8, 1>(2, 10000); runblk<2, 2>(2, 10000); runblk<1, 4>(2, 10000); runblk<5, 1>(2, 10000);
  run4<8>(1, 20000); run4<8>(2, 20000); run4<16>(4, 10000);
  return 0;
}
