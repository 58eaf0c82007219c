// Does v_mfma_f64_16x16x4_f64 issue faster when other instructions sit between consecutive MFMAs?
// 16 accumulators per wave (the syrk2 register block), inline asm so the order is exactly as written.
// hipcc --offload-arch=gfx950 -O3 tools/archive/mfma_interleave_probe.hip -o tools/bin/mfma_interleave_probe
#pragma clang diagnostic ignored "-Wunused-value"
#pragma clang diagnostic ignored "-Wunused-result"
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

#define MFMA(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

// MODE 0: back to back; 1..8: that many v_add_u32 between MFMAs; 100: one ds_read_b64 between MFMAs;
// 101: one ds_read_b64 every second MFMA (the syrk2 ratio: 8 fragment reads per 16 MFMAs); 200: s_nop 7
template <int MODE>
__global__ void __launch_bounds__(256) kern(double* out, int iters, double a0) {
  __shared__ double lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = 1.0 + i * 1e-6;
  __syncthreads();
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = 0.5 + i; }
  unsigned dummy = threadIdx.x;
  const double* lp = lds + (threadIdx.x & 63) * 8;
  double ld0 = 0.0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      MFMA(acc[i], a[i & 3], b[i >> 2]);
      if (MODE >= 1 && MODE <= 8) {
#pragma unroll
        for (int q = 0; q < MODE; q++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(dummy));
      }
      if (MODE == 100 || (MODE == 101 && (i & 1))) {
        double t;
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"((unsigned)(size_t)lp), "n"(0));
        ld0 += 0.0 * 0 + 0.0;  // keep simple; the value is consumed after the loop
        asm volatile("" ::"v"(t));
      }
      if (MODE == 200) asm volatile("s_nop 7");
    }
  }
  double s = ld0 + (double)dummy;
#pragma unroll
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) out[0] = s;
}

template <int MODE>
void run(int wps, int iters) {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * wps;  // 4 waves per block: wps blocks per CU -> wps waves per SIMD
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(kern<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double nm = (double)blocks * 4 * iters * 16;
  printf("mode %3d waves/SIMD=%d : %6.2f TF\n", MODE, wps, nm * 2048 / ms / 1e9);
  hipFree(d);
}
int main() {
  for (int wps = 1; wps <= 2; wps++) {
    run<0>(wps, 4000); run<1>(wps, 4000); run<2>(wps, 4000); run<4>(wps, 4000); run<8>(wps, 4000);
    run<100>(wps, 4000); run<101>(wps, 4000); run<200>(wps, 4000);
  }
  return 0;
}
