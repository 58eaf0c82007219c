// How fast does the chip START workgroups?  (round 6, docs/EXPERIMENTS.md G10.)  syrk4_kernel's time per 64 x 64 tile fits
// t(K) = 14 ns + 12.4 ns x K / 128 chip-wide at config C (profiles/r06_v1_bench_by_grid.txt): a fixed 14 ns per TILE = per
// workgroup, 26 % of the trailing update's time.  Is that the dispatcher?  Empty kernels of the trailing update's shape -- 256
// threads, 32 KB of LDS, ~70 VGPRs, so five workgroups per CU -- that do nothing, or busy-wait a given number of cycles:
// ns per workgroup chip-wide = launch time / workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/wg_launch_probe tools/wg_launch_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

template <int LDSB>
__global__ void __launch_bounds__(256, 4) wg_kernel(double* sink, int spin) {
  __shared__ __attribute__((aligned(1024))) char smem[LDSB];
  double acc[32];  // (keeps ~70 VGPRs live so that the occupancy is the trailing update's)
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = threadIdx.x + i;
  if (spin > 0) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) {
#pragma unroll
      for (int i = 0; i < 32; i++) acc[i] = acc[i] * 1.0000001 + 1e-9;
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 32; i++) s += acc[i];
  if (s == 1.2345e300) {
    smem[threadIdx.x] = 1;
    sink[0] = s + smem[(threadIdx.x + 1) & 255];
  }
}

template <int LDSB>
static void run(const char* name, double* sink, int nwg, int spin) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(wg_kernel<LDSB>, dim3(nwg), dim3(256), 0, 0, sink, spin);
  hipEventRecord(e0, 0);
  const int reps = 10;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(wg_kernel<LDSB>, dim3(nwg), dim3(256), 0, 0, sink, spin);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-28s %7d workgroups, spin %6d cycles: %8.1f us per launch = %6.2f ns per workgroup chip-wide\n", name, nwg, spin,
         ms / reps * 1e3, ms / reps * 1e6 / nwg);
}

int main() {
  double* sink = nullptr;
  hipMalloc(&sink, 64);
  for (int nwg : {7552, 38400}) {
    for (int spin : {0, 2000, 10000, 40000}) {
      run<32768>("256 thr, 32 KB LDS", sink, nwg, spin);
      run<1024>("256 thr,  1 KB LDS", sink, nwg, spin);
    }
  }
  return 0;
}
