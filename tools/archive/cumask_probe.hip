// hipExtStreamCreateWithCUMask on MI355X: which (XCC, SE, CU) does mask bit i select?  And: do two kernels on
// complementary masks really run side by side (a resident spinner on mask A, work on mask B)?
// build: hipcc --offload-arch=gfx950 -O2 -o tools/bin/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void where_kernel(unsigned* out) {
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);       // XCC_ID
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID: wave/simd/pipe/cu/sh/se...
    out[blockIdx.x * 2] = xcc;
    out[blockIdx.x * 2 + 1] = hwid;
  }
  // stay a little so that every block gets its own slot
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
}

__global__ void spinner(volatile unsigned* flag, unsigned* resident) {
  if (threadIdx.x == 0) {
    atomicAdd(resident, 1u);
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load((unsigned*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < 200000000ull) __builtin_amdgcn_s_sleep(8);
  }
}
// co-residency census: every block announces itself, then waits (bounded) until `want` blocks have announced
__global__ void __launch_bounds__(512) census(unsigned* cnt, unsigned want, unsigned* ok, unsigned* xccs) {
  extern __shared__ char big[];
  if (threadIdx.x == 0) {
    big[0] = 1;
    xccs[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u;
    atomicAdd(cnt, 1u);
    const unsigned long long t0 = wall_clock64();
    bool all = false;
    while (wall_clock64() - t0 < 20000000ull) {  // 200 ms
      if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) { all = true; break; }
      __builtin_amdgcn_s_sleep(16);
    }
    if (all) atomicAdd(ok, 1u);
  }
}
__global__ void worker(unsigned* cnt) { if (threadIdx.x == 0) atomicAdd(cnt, 1u); }
__global__ void release(unsigned* flag) { __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("CUs %d\n", p.multiProcessorCount);
  unsigned* dout;
  CK(hipMalloc(&dout, 4096 * 8));
  for (int bit : {0, 1, 2, 7, 8, 9, 31, 32, 33, 63, 64, 127, 128, 255}) {
    std::vector<uint32_t> mask(8, 0);
    mask[bit / 32] = 1u << (bit % 32);
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("bit %d: create failed\n", bit); continue; }
    CK(hipMemset(dout, 0xff, 64));
    hipLaunchKernelGGL(where_kernel, dim3(4), dim3(64), 0, st, dout);
    CK(hipStreamSynchronize(st));
    unsigned h[8];
    CK(hipMemcpy(h, dout, 32, hipMemcpyDeviceToHost));
    printf("bit %3d -> xcc %u hwid 0x%08x (cu %u sh %u se %u) | blk1 xcc %u hwid 0x%08x\n", bit, h[0], h[1], (h[1] >> 8) & 15, (h[1] >> 12) & 1,
           (h[1] >> 13) & 7, h[2], h[3]);
    CK(hipStreamDestroy(st));
  }
  // complementary masks: 32 spinners (one per CU: 64 KB LDS each would be better; here 1024 threads) on the low 32 bits, workers on the rest
  std::vector<uint32_t> ma(8, 0), mb(8, 0xffffffffu);
  ma[0] = 0xffffffffu;
  mb[0] = 0;
  hipStream_t sa, sb;
  CK(hipExtStreamCreateWithCUMask(&sa, 8, ma.data()));
  CK(hipExtStreamCreateWithCUMask(&sb, 8, mb.data()));
  unsigned *flag, *res, *cnt;
  CK(hipMalloc(&flag, 4)); CK(hipMalloc(&res, 4)); CK(hipMalloc(&cnt, 4));
  CK(hipMemset(flag, 0, 4)); CK(hipMemset(res, 0, 4)); CK(hipMemset(cnt, 0, 4));
  hipLaunchKernelGGL(spinner, dim3(32), dim3(1024), 0, sa, flag, res);
  hipLaunchKernelGGL(worker, dim3(100000), dim3(256), 0, sb, cnt);
  hipLaunchKernelGGL(release, dim3(1), dim3(1), 0, sb, flag);
  CK(hipStreamSynchronize(sb));
  CK(hipStreamSynchronize(sa));
  unsigned hr, hc;
  CK(hipMemcpy(&hr, res, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hc, cnt, 4, hipMemcpyDeviceToHost));
  printf("complementary masks: %u spinners were resident while %u worker blocks ran and released them: OK\n", hr, hc);
  // can 8k (and 8k - 3) workgroups of 512 threads + 157 KB LDS (the chain kernel's shape) be resident together on the low 8k bits?
  CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 161000));
  for (int k = 1; k <= 8; k++) {
    for (int nb : {8 * k, 8 * k - 3}) {
      std::vector<uint32_t> m(8, 0u);
      for (int i = 0; i < 8 * k; i++) m[i / 32] |= 1u << (i % 32);
      hipStream_t st;
      CK(hipExtStreamCreateWithCUMask(&st, 8, m.data()));
      unsigned *c2, *ok2, *xc;
      CK(hipMalloc(&c2, 4)); CK(hipMalloc(&ok2, 4)); CK(hipMalloc(&xc, 4 * 64));
      CK(hipMemset(c2, 0, 4)); CK(hipMemset(ok2, 0, 4));
      hipLaunchKernelGGL(census, dim3(nb), dim3(512), 161000, st, c2, (unsigned)nb, ok2, xc);
      CK(hipStreamSynchronize(st));
      unsigned hok, hx[64];
      CK(hipMemcpy(&hok, ok2, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx, xc, 4 * nb, hipMemcpyDeviceToHost));
      int per[8] = {0};
      for (int i = 0; i < nb; i++) per[hx[i] & 7]++;
      printf("k=%d: %2d chain-shaped workgroups: %2u saw all of them resident; per XCC:", k, nb, hok);
      for (int x = 0; x < 8; x++) printf(" %d", per[x]);
      printf("\n");
      CK(hipStreamDestroy(st));
    }
  }
  // the COMPLEMENT side (the tile kernel's shape: 512 threads + 128 KB LDS, one workgroup per CU): are 8 (32 - k) such
  // workgroups resident together on the complement of the low 8k bits -- also with 8 fewer, and with the chain side busy?
  for (int k : {1, 2, 3, 4, 8}) {
    for (int less : {0, 8, 16}) {
      const int nb = 8 * (32 - k) - less;
      std::vector<uint32_t> m(8, 0xffffffffu), ml(8, 0u);
      for (int i = 0; i < 8 * k; i++) m[i / 32] &= ~(1u << (i % 32)), ml[i / 32] |= 1u << (i % 32);
      hipStream_t st, sl;
      CK(hipExtStreamCreateWithCUMask(&st, 8, m.data()));
      CK(hipExtStreamCreateWithCUMask(&sl, 8, ml.data()));
      unsigned *c2, *ok2, *xc, *c3, *ok3, *xc3;
      CK(hipMalloc(&c2, 4)); CK(hipMalloc(&ok2, 4)); CK(hipMalloc(&xc, 4 * 256));
      CK(hipMalloc(&c3, 4)); CK(hipMalloc(&ok3, 4)); CK(hipMalloc(&xc3, 4 * 64));
      CK(hipMemset(c2, 0, 4)); CK(hipMemset(ok2, 0, 4)); CK(hipMemset(c3, 0, 4)); CK(hipMemset(ok3, 0, 4));
      hipLaunchKernelGGL(census, dim3(8 * k), dim3(512), 161000, sl, c3, (unsigned)(8 * k), ok3, xc3);
      hipLaunchKernelGGL(census, dim3(nb), dim3(512), 131072, st, c2, (unsigned)nb, ok2, xc);
      CK(hipStreamSynchronize(st));
      CK(hipStreamSynchronize(sl));
      unsigned hok, hok3, hx[256];
      CK(hipMemcpy(&hok, ok2, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&hok3, ok3, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hx, xc, 4 * nb, hipMemcpyDeviceToHost));
      int per[8] = {0};
      for (int i = 0; i < nb; i++) per[hx[i] & 7]++;
      printf("k=%d complement: %3d tile-shaped workgroups: %3u saw all of them resident (chain side: %u of %d); per XCC:", k, nb, hok, hok3, 8 * k);
      for (int x = 0; x < 8; x++) printf(" %d", per[x]);
      printf("\n");
      CK(hipStreamDestroy(st));
      CK(hipStreamDestroy(sl));
    }
  }
  // low 8k bits = k CUs in every XCC?  (bit i -> XCC i % 8, the driver's symmetric map); complement = the other 32 - k
  for (int k : {1, 2, 4, 8}) {
    for (int comp = 0; comp < 2; comp++) {
      std::vector<uint32_t> m(8, comp ? 0xffffffffu : 0u);
      for (int i = 0; i < 8 * k; i++) {
        if (comp) m[i / 32] &= ~(1u << (i % 32)); else m[i / 32] |= 1u << (i % 32);
      }
      hipStream_t st;
      CK(hipExtStreamCreateWithCUMask(&st, 8, m.data()));
      CK(hipMemset(dout, 0xff, 4096 * 8));
      hipLaunchKernelGGL(where_kernel, dim3(4096), dim3(64), 0, st, dout);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned> h(8192);
      CK(hipMemcpy(h.data(), dout, 8192 * 4, hipMemcpyDeviceToHost));
      int used[8][8][16] = {};
      for (int i = 0; i < 4096; i++) used[h[2 * i] & 7][(h[2 * i + 1] >> 13) & 7][(h[2 * i + 1] >> 8) & 15]++;
      printf("k=%d %s: distinct CUs per XCC:", k, comp ? "complement" : "low bits ");
      for (int x = 0; x < 8; x++) {
        int c = 0;
        for (int se = 0; se < 8; se++) for (int cu = 0; cu < 16; cu++) c += used[x][se][cu] > 0;
        printf(" %d", c);
      }
      printf("\n");
      CK(hipStreamDestroy(st));
    }
  }
  return 0;
}
