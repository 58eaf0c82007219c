// fp64 throughput probe for MI355X: MFMA vs VALU FMA vs both pipes, at several occupancies.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/archive/fp64_probe.hip -o /tmp/fp64_probe && /tmp/fp64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 mfma only, 1 valu only, 2 both in the same wave
__global__ void __launch_bounds__(256) k(double* out, int iters, double a0, double b0) {
  d4 acc[8];
  double v[16];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int i = 0; i < 16; i++) v[i] = threadIdx.x * 1e-3 + i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; it++) {
    if (MODE != 1) {
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    if (MODE != 0) {
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = __builtin_fma(v[i], a, b);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  if (s == 123.456) out[0] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, double a0) {
  double* d;
  hipMalloc(&d, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 64, a0, 0.5);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, a0, 0.5);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double waves = blocks * 4.0;
  double mf = (MODE != 1) ? waves * iters * 8.0 * 2048.0 : 0.0;
  double vf = (MODE != 0) ? waves * iters * 128.0 * 64 * 2.0 : 0.0;
  printf("%-28s blocks=%5d  %8.3f ms  mfma %6.2f TF  valu %6.2f TF  total %6.2f TF\n", name, blocks, ms,
         mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
  hipFree(d);
}

int main() {
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * wps;
    int iters = 40000 / wps;
    run<0>("mfma only", blocks, iters, 1.0);
    run<1>("valu fma only", blocks, iters, 1.0);
    run<2>("mfma+valu same wave", blocks, iters, 1.0);
  }
  run<0>("mfma only, a=0 (zeros)", 256 * 4, 10000, 0.0);
  run<0>("mfma only, a=0.999", 256 * 4, 10000, 0.999);
  return 0;
}
