// Round-1 NT tile GEMM building blocks (operands staged through VGPRs into 128x32 LDS chunks, leading dimension 34).
// Retired from libbgp.so in round 3; kept for the A/B benches under tools/ (syrk2_kernel is the bit-identical reference
// of tools/syrk4_bench.hip).
#pragma once
#include "bgp_common.h"
#include "bgp_gemm.h"

#define GK_KC 32
#define GK_LD 34

struct __attribute__((aligned(16))) GemmSmem {
  double A[128 * GK_LD];
  double B[128 * GK_LD];
  double ypart[128];
};

// 128 x 32 chunk of a row-major matrix (leading dimension ld) -> LDS tile [128][GK_LD].
static __device__ __forceinline__ void gk_load_chunk(double* __restrict__ dst, const double* __restrict__ src,
                                                     size_t ld, int tid) {
  d2 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = tid + 256 * i;
    const int row = c >> 4, seg = c & 15;
    v[i] = *reinterpret_cast<const d2*>(src + (size_t)row * ld + seg * 2);
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int c = tid + 256 * i;
    const int row = c >> 4, seg = c & 15;
    *reinterpret_cast<d2*>(dst + row * GK_LD + seg * 2) = v[i];
  }
}

// Generic per-wave MFMA block: acc[i][j] (+)= sum_k A[r0+16i+..][k] * B[c0+16j+..][k] over one 32-wide
// chunk, for an NR x NC grid of 16x16 MFMA tiles whose top-left corner is (r0, c0) inside the
// 128x128 workgroup tile.
// MFMA operand layout (cdna_hip_programming.md section 3): A operand lane l = A[l&15][l>>4],
// B operand lane l = B[k=l>>4][j=l&15] = Bmat[l&15][l>>4]: both read [row = l&15][k = l>>4].
//   NEG   : use -A (trailing update subtracts)
//   KSKIP : the B matrix is lower triangular (W_kk): column block j only needs k <= its last column
//   CREL  : tile (i, j) is computed only when j + CREL <= i (lower-triangular part of a diagonal
//           workgroup tile); CREL = -64 disables the test.
template <int NR, int NC, int NEG, int KSKIP, int CREL>
static __device__ __forceinline__ void gk_mma_block(const double* __restrict__ As, const double* __restrict__ Bs,
                                                    d4 (&acc)[NR][NC], int r0, int c0, int lane, int k0) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < GK_KC / 4; kk++) {
    double a[NR], b[NC];
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const double av = As[(r0 + i * 16 + lr) * GK_LD + kk * 4 + lk];
      a[i] = NEG ? -av : av;
    }
#pragma unroll
    for (int j = 0; j < NC; j++) b[j] = Bs[(c0 + j * 16 + lr) * GK_LD + kk * 4 + lk];
#pragma unroll
    for (int j = 0; j < NC; j++) {
      if (KSKIP && (k0 + kk * 4 > c0 + j * 16 + 15)) continue;  // wave-uniform
#pragma unroll
      for (int i = 0; i < NR; i++) {
        if (j + CREL > i) continue;  // compile-time
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
      }
    }
  }
}

// Back-compat wrapper: the 2x2-wave layout (each wave a 64x64 sub-tile).
template <int NEG, int TRI>
static __device__ __forceinline__ void gk_mma_chunk(const double* __restrict__ As, const double* __restrict__ Bs,
                                                    d4 (&acc)[4][4], int wr, int wc, int lane, int k0) {
  gk_mma_block<4, 4, NEG, TRI, -64>(As, Bs, acc, wr * 64, wc * 64, lane, k0);
}

