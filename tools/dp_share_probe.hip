// Do fp64 MFMA and fp64 VALU instructions of DIFFERENT waves on one SIMD run side by side on gfx950?  (round 6, docs/EXPERIMENTS.md
// G17: the trailing update's bulk launch hid only ~18 % of the Gram generation's VALU work.)  Four waves per SIMD (1024 workgroups
// of 1024 threads: 4 per ... no: 256 CUs x 4 SIMDs x 4 waves): mode 0: all four run a chain of v_mfma_f64_16x16x4_f64; mode 1: all
// four run chains of v_fma_f64; mode 2: two and two.  Prints time and the rate of each kind; side by side would make mode 2 as
// fast as the slower half alone.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N_IT 4096
__global__ void __launch_bounds__(256) probe(double* out, int mode) {
  const int w = threadIdx.x >> 6;  // wave w of the workgroup sits on SIMD w (four workgroups per CU: four waves per SIMD)
  const bool mfma = mode == 0 || (mode == 2 && (blockIdx.x & 1));
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  if (mfma) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < N_IT; i++) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    double x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
    for (int i = 0; i < N_IT * 4; i++) {  // 32 independent-enough fma per iteration
#pragma unroll
      for (int u = 0; u < 4; u++) {
        x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
        x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
      }
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + w;
  }
}
int main() {
  double* d;
  const int wgs = 256 * 4;  // four 256-thread workgroups per CU
  hipMalloc(&d, (size_t)wgs * 256 * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++)
    for (int mode = 0; mode < 3; mode++) {
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, d, mode);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, d, mode);
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double nm = (mode == 0 ? wgs : mode == 2 ? wgs / 2 : 0) * 4.0, nv = (mode == 1 ? wgs : mode == 2 ? wgs / 2 : 0) * 4.0;  // waves
      const double tf_m = nm * N_IT * 4.0 * (16 * 16 * 4 * 2) / (ms * 1e-3) / 1e12, tf_v = nv * N_IT * 4.0 * 32 * 64 * 2 / (ms * 1e-3) / 1e12;
      if (rep) printf("mode %d (%s): %.3f ms  MFMA %.1f TF  VALU %.1f TF\n", mode, mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : "half and half", ms, tf_m, tf_v);
    }
  return 0;
}
