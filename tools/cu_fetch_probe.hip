// What ONE compute unit can pull of a block that another workgroup has just written (round 6; docs/EXPERIMENTS.md G1).
// The launch-free factorisation's chain workgroup fetches block (J+1, J) -- 128 x 128 doubles, rows of 1 KB at the matrix's
// leading dimension -- behind the tile task that produced it; in-kernel stamps read ~15 GB/s for that fetch (16 x 16-byte loads
// per lane, all 512 threads, everything in flight at once).  The round-5 review read that as "issued too narrowly" against the
// guide's 25 GB/s (one LDS-DMA loader wave) / 90 GB/s (three consumers) per CU.  This probe measures the fetch alone:
//   producer kernel: workgroup b writes block b (plain stores; kernel boundary = release)
//   consumer kernel: workgroup b reads block (b + shift) % B in one of the forms below and stamps wall_clock64 around it
//     form 0  registers, the chain's own form: 16 x global_load_dwordx4 per lane, 8 waves
//     form 1  LDS-DMA (global_load_lds_dwordx4, 1 KB per wave instruction) from NW = 1 / 2 / 4 / 8 waves, whole block in flight
// for B = 1 (alone on the chip) and B = 32 (config B's 32 chains at once), shift = 0 (producer on the same XCD by the b % 8
// placement) and shift = 1 (another XCD), and with a 600 MB sweep in between (block no longer in any cache).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/cu_fetch_probe tools/cu_fetch_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                        \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

constexpr int LD = 1024;                         // leading dimension of the matrices (doubles): config B's n
constexpr size_t MSTRIDE = (size_t)LD * LD;      // one matrix
constexpr size_t BLOCK_OFF = (size_t)128 * LD;   // block (1, 0) of the matrix

__global__ void __launch_bounds__(512) produce(double* M, int B, double seed) {
  double* blk = M + (size_t)blockIdx.x * MSTRIDE + BLOCK_OFF;
  for (int e = threadIdx.x; e < 128 * 128; e += 512) blk[(size_t)(e >> 7) * LD + (e & 127)] = seed + e;
}

__global__ void sweep(const double* big, size_t n, double* sink) {
  double s = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += big[i];
  if (s == 1.2345e300) *sink = s;
}

static __device__ __forceinline__ void glds(const double* gbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(gbase), "s"(lds_addr)
               : "memory");
}

template <int FORM, int NW>
__global__ void __launch_bounds__(512) consume(const double* M, int B, int shift, unsigned long long* stamps, double* sink) {
  __shared__ __attribute__((aligned(1024))) double lds[128 * 128];  // 128 KB
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* blk = M + (size_t)((b + shift) % B) * MSTRIDE + BLOCK_OFF;
  __syncthreads();
  const unsigned long long t0 = wall_clock64();
  double acc = 0.0;
  if (FORM == 0) {
    // the chain's A-fragment request: row 16 w + (lane & 15), 32 contiguous bytes per lane and k-group, 16 loads in flight
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v* q = reinterpret_cast<const d2v*>(blk + (size_t)(16 * w + (lane & 15)) * LD + 4 * (lane >> 4));
    d2v v[16];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      v[2 * g] = q[8 * g];
      v[2 * g + 1] = q[8 * g + 1];
    }
#pragma unroll
    for (int g = 0; g < 16; g++) acc += v[g][0] + v[g][1];
  } else if (FORM == 2 || FORM == 3) {
    // registers again, but every wave instruction covers WHOLE 128-byte lines (8 rows x 128 B, the LDS-DMA's pattern): 16 loads
    // per lane; FORM 3: the same with non-temporal loads
    typedef double d2v __attribute__((ext_vector_type(2)));
    d2v v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const int p = w * 16 + i, r8 = p >> 3, c = p & 7;
      const d2v* q = reinterpret_cast<const d2v*>(blk + (size_t)(8 * r8 + (lane >> 3)) * LD + 16 * c + 2 * (lane & 7));
      v[i] = FORM == 3 ? __builtin_nontemporal_load(q) : *q;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) acc += v[i][0] + v[i][1];
  } else if (FORM == 5) {
    // registers, 16 rows x 64 CONTIGUOUS bytes per instruction (four lanes of a row side by side; the other half of the row's
    // 128-byte line in the second load of the pair): reachable from the chain's pattern with one more lane-swap level
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v* q = reinterpret_cast<const d2v*>(blk + (size_t)(16 * w + (lane & 15)) * LD + 2 * (lane >> 4));
    d2v v[16];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      v[2 * g] = q[8 * g];
      v[2 * g + 1] = q[8 * g + 4];
    }
#pragma unroll
    for (int g = 0; g < 16; g++) acc += v[g][0] + v[g][1];
  } else if (FORM == 4) {
    // the chain's pattern with non-temporal loads
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v* q = reinterpret_cast<const d2v*>(blk + (size_t)(16 * w + (lane & 15)) * LD + 4 * (lane >> 4));
    d2v v[16];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      v[2 * g] = __builtin_nontemporal_load(q + 8 * g);
      v[2 * g + 1] = __builtin_nontemporal_load(q + 8 * g + 1);
    }
#pragma unroll
    for (int g = 0; g < 16; g++) acc += v[g][0] + v[g][1];
  } else {
    // 128 pieces of 1 KB (8 rows x 128 B each) dealt over NW waves, all issued before the first wait
    if (w < NW) {
      const unsigned lbase = (unsigned)(size_t)lds;
      for (int p = w; p < 128; p += NW) {
        const int r8 = p >> 3, c = p & 7;  // rows 8 r8 .. 8 r8 + 7, columns 16 c .. 16 c + 15
        const unsigned voff = (unsigned)(((8 * r8 + (lane >> 3)) * LD + 16 * c + 2 * (lane & 7)) * 8);
        glds(blk, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)p * 1024u)));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const unsigned long long t1 = wall_clock64();
  if (FORM == 1) acc = lds[tid * 7 % (128 * 128)];
  if (acc == 1.2345e300) *sink = acc;
  if (tid == 0) stamps[b] = t1 - t0;
}

// background traffic: `gridDim.x` workgroups (128 KB of LDS each: one per CU) stream a large array through LDS-DMA rings, eight
// waves x 16 KB in flight per workgroup, for `iters` rounds -- what the tile workers of the launch-free kernel do beside the chain
__global__ void __launch_bounds__(512) stream_load(const double* big, size_t nbig, int iters, double* sink) {
  __shared__ __attribute__((aligned(1024))) double lds[128 * 128];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lbase = (unsigned)(size_t)lds;
  const size_t span = nbig / gridDim.x;  // doubles per workgroup
  const double* base = big + (size_t)blockIdx.x * span;
  size_t off = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int p = 0; p < 16; p++) {
      const unsigned voff = (unsigned)(((size_t)(w * 16 + p) * 128 + 2 * lane) * 8);
      glds(base + off, voff, (unsigned)__builtin_amdgcn_readfirstlane((int)(lbase + (unsigned)(w * 16 + p) * 1024u)));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    off += 16384;  // 128 KB per round and workgroup
    if (off + 16384 > span) off = 0;
  }
  __syncthreads();
  if (lds[tid] == 1.2345e300) *sink = lds[tid];
}

template <int FORM, int NW>
static int run_loaded(const char* name, double* M, double* big, size_t nbig, unsigned long long* dst, double* sink, int B, int shift,
                      hipStream_t s1, hipStream_t s2) {
  std::vector<double> us;
  for (int rep = 0; rep < 8; rep++) {
    hipLaunchKernelGGL(stream_load, dim3(224), dim3(512), 0, s2, big, nbig, 400, sink);  // ~ a few hundred us of traffic
    hipLaunchKernelGGL(produce, dim3(B), dim3(512), 0, s1, M, B, (double)rep);
    hipLaunchKernelGGL((consume<FORM, NW>), dim3(B), dim3(512), 0, s1, M, B, shift, dst, sink);
    CHECK(hipStreamSynchronize(s1));
    const bool busy = hipStreamQuery(s2) == hipErrorNotReady;  // (the load was still running when the consumer finished)
    (void)hipGetLastError();
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(B);
    CHECK(hipMemcpy(h.data(), dst, B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (rep < 2 || !busy) continue;
    for (int b = 0; b < B; b++) us.push_back(h[b] / 100.0);
  }
  if (us.empty()) {
    printf("%-34s B=%2d shift=%d LOADED: the background kernel never overlapped the consumer\n", name, B, shift);
    return 0;
  }
  std::sort(us.begin(), us.end());
  const double med = us[us.size() / 2], best = us.front();
  printf("%-34s B=%2d shift=%d LOADED (224 CUs streaming)  median %6.2f us = %5.1f GB/s per CU   best %6.2f us = %5.1f GB/s  (%zu samples)\n",
         name, B, shift, med, 131072.0 / med / 1e3, best, 131072.0 / best / 1e3, us.size());
  return 0;
}

template <int FORM, int NW>
static int run(const char* name, double* M, double* big, size_t nbig, unsigned long long* dst, double* sink, int B, int shift, int cold) {
  std::vector<double> us;
  for (int rep = 0; rep < 12; rep++) {
    hipLaunchKernelGGL(produce, dim3(B), dim3(512), 0, 0, M, B, (double)rep);
    if (cold) hipLaunchKernelGGL(sweep, dim3(2048), dim3(256), 0, 0, big, nbig, sink);
    hipLaunchKernelGGL((consume<FORM, NW>), dim3(B), dim3(512), 0, 0, M, B, shift, dst, sink);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(B);
    CHECK(hipMemcpy(h.data(), dst, B * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (rep < 2) continue;
    for (int b = 0; b < B; b++) us.push_back(h[b] / 100.0);  // wall_clock64: 100 MHz
  }
  std::sort(us.begin(), us.end());
  const double med = us[us.size() / 2], best = us.front();
  printf("%-34s B=%2d shift=%d %s  median %6.2f us = %5.1f GB/s per CU   best %6.2f us = %5.1f GB/s\n", name, B, shift,
         cold ? "cold" : "warm", med, 131072.0 / med / 1e3, best, 131072.0 / best / 1e3);
  return 0;
}

int main() {
  const int BMAX = 32;
  double *M = nullptr, *big = nullptr, *sink = nullptr;
  unsigned long long* dst = nullptr;
  const size_t nbig = (size_t)600 * 1024 * 1024 / 8;
  CHECK(hipMalloc(&M, BMAX * MSTRIDE * sizeof(double)));
  CHECK(hipMalloc(&big, nbig * sizeof(double)));
  CHECK(hipMemset(big, 0, nbig * sizeof(double)));
  CHECK(hipMalloc(&sink, 8));
  CHECK(hipMalloc(&dst, BMAX * sizeof(unsigned long long)));
  for (int cold = 0; cold < 2; cold++)
    for (int B : {1, 32})
      for (int shift = 0; shift < (B > 1 ? 2 : 1); shift++) {
        if (run<0, 8>("registers 16 x dwordx4 / lane, 8 waves", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<2, 8>("registers, whole lines per instr", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<5, 8>("registers, 64 B contiguous per row", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<1, 1>("LDS-DMA, 1 wave", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<1, 2>("LDS-DMA, 2 waves", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<1, 4>("LDS-DMA, 4 waves", M, big, nbig, dst, sink, B, shift, cold)) return 1;
        if (run<1, 8>("LDS-DMA, 8 waves", M, big, nbig, dst, sink, B, shift, cold)) return 1;
      }
  hipStream_t s1, s2;
  CHECK(hipStreamCreate(&s1));
  CHECK(hipStreamCreate(&s2));
  for (int shift = 0; shift < 2; shift++) {
    if (run_loaded<0, 8>("registers 16 x dwordx4 / lane, 8 waves", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<2, 8>("registers, whole lines per instr", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<5, 8>("registers, 64 B contiguous per row", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<3, 8>("registers, whole lines, nt", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<4, 8>("registers, chain pattern, nt", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<1, 1>("LDS-DMA, 1 wave", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<1, 4>("LDS-DMA, 4 waves", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
    if (run_loaded<1, 8>("LDS-DMA, 8 waves", M, big, nbig, dst, sink, 32, shift, s1, s2)) return 1;
  }
  return 0;
}
