// NT tile GEMM building blocks on v_mfma_f64_16x16x4_f64 (shared by bgp_chol.hip / bgp_post.hip).
#pragma once
#include "bgp_common.h"

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

// acc[i][j] (+)= sum_k A[64wr+16i+.. ][k] * B[64wc+16j+..][k] over one 32-wide chunk.
// MFMA operand layout (cdna_hip_programming.md section 3): A operand lane l = A[l&15][l>>4],
// B operand lane l = B[k=l>>4][j=l&15] = Bmat[l&15][l>>4]: both read [row = l&15][k = l>>4].
// TRI != 0: the B matrix is lower triangular (W_kk): column block j only needs k <= its last column.
template <int NEG, int TRI>
static __device__ __forceinline__ void gk_mma_chunk(const double* __restrict__ As, const double* __restrict__ Bs,
                                                    d4 (&acc)[4][4], int wr, int wc, int lane, int k0) {
  const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
  for (int kk = 0; kk < GK_KC / 4; kk++) {
    double a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double av = As[(wr * 64 + i * 16 + lr) * GK_LD + kk * 4 + lk];
      a[i] = NEG ? -av : av;
      b[i] = Bs[(wc * 64 + i * 16 + lr) * GK_LD + kk * 4 + lk];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (TRI && (k0 + kk * 4 > wc * 64 + j * 16 + 15)) continue;  // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; i++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

// C/D fragment layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
// (cdna_hip_programming.md:247-251; verified at run time by bgp_mfma_f64_layout + tests).
#define GK_ROW(wr, i, lane, r) ((wr) * 64 + (i) * 16 + ((lane) >> 4) + 4 * (r))
#define GK_COL(wc, j, lane) ((wc) * 64 + (j) * 16 + ((lane) & 15))

