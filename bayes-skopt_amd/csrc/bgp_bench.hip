// Measurement hooks: fp64 MFMA peak, HBM streaming copy, empirical MFMA fragment layout.
// Used by bench.py (roofline denominators measured on the box, next to the spec numbers) and by
// tests/ (layout self-check).  Not part of the reference surface.
#include "bgp_common.h"

// 16 independent accumulators per wave as a 4 x 4 register block (4 A and 4 B fragments reused like the trailing
// update's inner loop), back to back, two waves per SIMD: the form in which v_mfma_f64_16x16x4_f64 reaches its
// issue rate (a chain of 8 accumulators on ONE operand pair, the round-1 probe, stops at ~49 TF).
#define BGP_PEAK_MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__global__ void __launch_bounds__(256, 2) mfma_f64_peak_kernel(double* out, int iters, double a0, double b0) {
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    a[i] = a0 + threadIdx.x * 1e-9 + i;
    b[i] = b0 * 0.5 + i;
  }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) BGP_PEAK_MFMA(acc[i], a[i & 3], b[i >> 2]);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;  // keep the chains live
}

extern "C" int bgp_bench_mfma_f64(int device, int iters, double* tflops) {
  if (!tflops || iters <= 0) return BGP_ERR_INVALID;
  BGP_HIP(hipSetDevice(device));
  double* d = nullptr;
  BGP_HIP(hipMalloc(&d, 8));
  hipEvent_t e0, e1;
  BGP_HIP(hipEventCreate(&e0));
  BGP_HIP(hipEventCreate(&e1));
  const int blocks = 256 * 2;  // 2 workgroups x 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0, 1.0);  // warm-up
  BGP_HIP(hipDeviceSynchronize());
  BGP_HIP(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
  BGP_HIP(hipEventRecord(e1, 0));
  BGP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  BGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)blocks * 4.0 * (double)iters * 16.0 * (2.0 * 16 * 16 * 4);
  *tflops = flops / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return BGP_OK;
}

__global__ void __launch_bounds__(256) hbm_copy_kernel(const d2* __restrict__ src, d2* __restrict__ dst, size_t n2) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) dst[i] = src[i];
}

extern "C" int bgp_bench_hbm_copy(int device, long long bytes, int iters, double* gbps) {
  if (!gbps || bytes < 4096 || iters <= 0) return BGP_ERR_INVALID;
  BGP_HIP(hipSetDevice(device));
  const size_t n2 = (size_t)bytes / 16;
  d2 *a = nullptr, *b = nullptr;
  BGP_HIP(hipMalloc(&a, n2 * 16));
  BGP_HIP(hipMalloc(&b, n2 * 16));
  BGP_HIP(hipMemset(a, 1, n2 * 16));
  hipEvent_t e0, e1;
  BGP_HIP(hipEventCreate(&e0));
  BGP_HIP(hipEventCreate(&e1));
  hipLaunchKernelGGL(hbm_copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n2);
  BGP_HIP(hipDeviceSynchronize());
  BGP_HIP(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL(hbm_copy_kernel, dim3(2048), dim3(256), 0, 0, a, b, n2);
  BGP_HIP(hipEventRecord(e1, 0));
  BGP_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  BGP_HIP(hipEventElapsedTime(&ms, e0, e1));
  *gbps = 2.0 * (double)n2 * 16.0 * iters / (ms * 1e-3) / 1e9;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  return BGP_OK;
}

// D1[i][j] = i and D2[i][j] = j through the documented A/B operand layouts
// (A operand lane l = A[l&15][l>>4], B operand lane l = B[l>>4][l&15]); each lane then reports
// which (row, col) its 4 result registers hold.
__global__ void mfma_layout_kernel(int* rows, int* cols) {
  const int l = threadIdx.x;
  d4 z = (d4){0.0, 0.0, 0.0, 0.0};
  double a1 = (l < 16) ? (double)(l & 15) : 0.0;  // A[i][k=0] = i
  double b1 = (l < 16) ? 1.0 : 0.0;               // B[k=0][j] = 1
  d4 r1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, z, 0, 0, 0);
  double a2 = (l < 16) ? 1.0 : 0.0;
  double b2 = (l < 16) ? (double)(l & 15) : 0.0;
  d4 r2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, z, 0, 0, 0);
  for (int r = 0; r < 4; r++) {
    rows[l * 4 + r] = (int)r1[r];
    cols[l * 4 + r] = (int)r2[r];
  }
}

extern "C" int bgp_mfma_f64_layout(int device, int* rows, int* cols) {
  if (!rows || !cols) return BGP_ERR_INVALID;
  BGP_HIP(hipSetDevice(device));
  int *dr = nullptr, *dc = nullptr;
  BGP_HIP(hipMalloc(&dr, 256 * sizeof(int)));
  BGP_HIP(hipMalloc(&dc, 256 * sizeof(int)));
  hipLaunchKernelGGL(mfma_layout_kernel, dim3(1), dim3(64), 0, 0, dr, dc);
  BGP_HIP(hipDeviceSynchronize());
  BGP_HIP(hipMemcpy(rows, dr, 256 * sizeof(int), hipMemcpyDeviceToHost));
  BGP_HIP(hipMemcpy(cols, dc, 256 * sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(dr);
  (void)hipFree(dc);
  return BGP_OK;
}
