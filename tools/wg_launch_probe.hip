// How fast does the chip START workgroups?  (round 6, docs/EXPERIMENTS.md G10.)  syrk4_kernel's time per 64 x 64 tile fits
// t(K) = 14 ns + 12.4 ns x K / 128 chip-wide at config C (profiles/r06_v1_bench_by_grid.txt): a fixed 14 ns per TILE = per
// workgroup, 26 % of the trailing update's time.  Is that the dispatcher?  Empty kernels of the trailing update's shape -- 256
// threads, 32 KB of LDS, ~70 VGPRs, so five workgroups per CU -- that do nothing, or busy-wait a given number of cycles:
// ns per workgroup chip-wide = launch time / workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/wg_launch_probe tools/wg_launch_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

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

// census: how many workgroups of this shape does ONE compute unit hold at a time?  Every workgroup stamps (XCC id, HW_ID, start,
// end); the host takes, per (XCC, shader engine, CU), the largest number of intervals that overlap.
template <int LDSB>
__global__ void __launch_bounds__(256, 4) census_kernel(unsigned long long* rec, int spin, double* sink) {
  __shared__ __attribute__((aligned(1024))) char smem[LDSB];
  double acc[32];
#pragma unroll
  for (int i = 0; i < 32; i++) acc[i] = threadIdx.x + i;
  const unsigned long long t0 = wall_clock64();
  const unsigned long long c0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - c0 < (unsigned long long)spin) {
#pragma unroll
    for (int i = 0; i < 32; i++) acc[i] = acc[i] * 1.0000001 + 1e-9;
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 32; i++) s += acc[i];
  if (s == 1.2345e300) {
    smem[threadIdx.x] = 1;
    sink[0] = s + smem[(threadIdx.x + 1) & 255];
  }
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    rec[(size_t)blockIdx.x * 4 + 0] = ((unsigned long long)(xcc & 15) << 32) | hw;
    rec[(size_t)blockIdx.x * 4 + 1] = t0;
    rec[(size_t)blockIdx.x * 4 + 2] = wall_clock64();
  }
}

template <int LDSB>
static void census(unsigned long long* drec, double* sink, int nwg, int spin) {
  hipMemset(drec, 0, (size_t)nwg * 4 * 8);
  hipLaunchKernelGGL(census_kernel<LDSB>, dim3(nwg), dim3(256), 0, 0, drec, spin, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)nwg * 4);
  hipMemcpy(h.data(), drec, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;  // per CU: (time, +1 / -1)
  for (int g = 0; g < nwg; g++) {
    const unsigned hw = (unsigned)(h[(size_t)g * 4] & 0xffffffffu);
    const unsigned xcc = (unsigned)(h[(size_t)g * 4] >> 32);
    // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
    const unsigned long long cu = ((unsigned long long)xcc << 16) | ((hw >> 8) & 0xff);
    ev[cu].push_back({h[(size_t)g * 4 + 1], +1});
    ev[cu].push_back({h[(size_t)g * 4 + 2], -1});
  }
  int worst = 0, best = 1 << 30;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0, mx = 0;
    for (auto& e : kv.second) {
      cur += e.second;
      mx = std::max(mx, cur);
    }
    worst = std::max(worst, mx);
    best = std::min(best, mx);
  }
  printf("census: 256 threads, %6d B of LDS: %zu compute units seen, workgroups resident at a time per CU: %d .. %d\n", LDSB, ev.size(),
         best, worst);
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
  unsigned long long* drec = nullptr;
  hipMalloc(&drec, (size_t)8192 * 4 * 8);
  census<32768>(drec, sink, 8192, 40000);
  census<32256>(drec, sink, 8192, 40000);
  census<31744>(drec, sink, 8192, 40000);
  census<30720>(drec, sink, 8192, 40000);
  census<28672>(drec, sink, 8192, 40000);
  census<24576>(drec, sink, 8192, 40000);
  census<16384>(drec, sink, 8192, 40000);
  return 0;
}
