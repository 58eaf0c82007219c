// Does a captured hipGraph shorten the fixed cost of one small LML call on MI355X?  The shape of the launch-free path's
// enqueue -- H2D (pinned), memset, two small kernels, memset, one longer kernel, three D2H copies (pinned), host polls for
// completion -- replayed 2000 times as plain stream calls and as ONE hipGraphLaunch.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/graph_probe tools/graph_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void small_kernel(double* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0000001 + 1.0; }
__global__ void spin_kernel(double* p, unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x == 0) p[blockIdx.x] += 1.0;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t st;
  CK(hipStreamCreate(&st));
  double *d, *h;
  unsigned* dflags;
  CK(hipMalloc(&d, 1 << 20));
  CK(hipMalloc(&dflags, 16384));
  CK(hipHostMalloc((void**)&h, 1 << 16, hipHostMallocDefault));
  auto enqueue = [&](hipStream_t s) {
    hipMemcpyAsync(d, h, 50 * 10 * 8, hipMemcpyHostToDevice, s);
    hipMemsetAsync(dflags, 0, 256, s);
    hipLaunchKernelGGL(small_kernel, dim3(64), dim3(256), 0, s, d, 16384);
    hipLaunchKernelGGL(small_kernel, dim3(256), dim3(256), 0, s, d, 65536);
    hipMemsetAsync(dflags, 0, 12000, s);
    hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), 0, s, d, 10000ull);  // 100 us of "factorisation"
    hipMemcpyAsync(h + 1024, dflags, 4, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h + 2048, d, 50 * 8, hipMemcpyDeviceToHost, s);
    hipMemcpyAsync(h + 3072, dflags + 64, 50 * 4, hipMemcpyDeviceToHost, s);
  };
  auto wait = [&]() { while (hipStreamQuery(st) == hipErrorNotReady) {} };
  const int reps = 2000;
  for (int i = 0; i < 50; i++) { enqueue(st); wait(); }
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) { enqueue(st); wait(); }
  auto t1 = std::chrono::steady_clock::now();
  const double plain = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  enqueue(st);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 50; i++) { CK(hipGraphLaunch(ge, st)); wait(); }
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < reps; i++) { hipGraphLaunch(ge, st); wait(); }
  t1 = std::chrono::steady_clock::now();
  const double graph = std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
  printf("per call: plain stream calls %.1f us, one hipGraphLaunch %.1f us (the kernels alone: ~100 us of spin + ~10 us)\n", plain, graph);
  return 0;
}
