// 8-wave (512-thread) NT tile GEMM building blocks on v_mfma_f64_4x4x4_4b_f64 -- the fast fp64 MFMA form
// on MI355X (73-75 TFLOP/s vs ~49 for 16x16x4, tools/mfma_issue_probe.hip).  Used by the left-looking
// update and the panel solve of the LML path (bgp_llchol.hip); the 4-wave 16x16x4 helpers of bgp_gemm.h
// remain in use for posterior builds and predict.
#pragma once
#include "bgp_gemm.h"

#define G8_THREADS 512  // 8 waves: 4 (rows) x 2 (cols), each wave a 32x64 block (64 accumulator VGPRs)

// 128 x 32 chunk of a row-major matrix (leading dimension ld, in doubles): global -> registers
// (g8_fetch_chunk, 4 double2 per thread) and registers -> LDS tile [128][GK_LD] (g8_stash_chunk).
// Split in two so that the global loads of chunk c+1 are in flight while chunk c is multiplied.
struct G8Regs {
  d2 v[4];
};
static __device__ __forceinline__ void g8_fetch_chunk(G8Regs& r, const double* __restrict__ src, size_t ld, int tid) {
  const int row0 = tid >> 4, seg = tid & 15;
  const double* p = src + (size_t)row0 * ld + seg * 2;
#pragma unroll
  for (int i = 0; i < 4; i++) r.v[i] = *reinterpret_cast<const d2*>(p + (size_t)(32 * i) * ld);
}
static __device__ __forceinline__ void g8_stash_chunk(double* __restrict__ dst, const G8Regs& r, int tid) {
  const int row0 = tid >> 4, seg = tid & 15;
#pragma unroll
  for (int i = 0; i < 4; i++) *reinterpret_cast<d2*>(dst + (row0 + 32 * i) * GK_LD + seg * 2) = r.v[i];
}
static __device__ __forceinline__ void g8_load_chunk(double* __restrict__ dst, const double* __restrict__ src,
                                                     size_t ld, int tid) {
  G8Regs r;
  g8_fetch_chunk(r, src, ld, tid);
  g8_stash_chunk(dst, r, tid);
}

template <int NR, int NC, int NEG, int KSKIP, int CREL>
static __device__ __forceinline__ void g8_mma_block(const double* __restrict__ As, const double* __restrict__ Bs,
                                                    d4 (&acc)[NR][NC], int r0, int c0, int lane, int k0);

// Main loop of C(128x128 tile) (+)= A(128 x K) * B(128 x K)^T with register prefetch of the next
// chunk: per chunk two barriers (LDS reuse), global latency hidden behind the MFMAs.
// SAMEB: B is the same matrix as A (diagonal syrk tile) -- staged once.
template <int NR, int NC, int NEG, int KSKIP, int CREL, int SAMEB, int PREFETCH = 1>
static __device__ __forceinline__ void g8_mainloop(GemmSmem& sm, const double* __restrict__ A, size_t lda,
                                                   const double* __restrict__ Bm, size_t ldb, int K,
                                                   d4 (&acc)[NR][NC], int r0, int c0, int tid, int lane) {
  if (!PREFETCH) {  // two workgroups per CU hide each other's staging instead (fewer live registers)
    for (int k0 = 0; k0 < K; k0 += GK_KC) {
      __syncthreads();
      g8_load_chunk(sm.A, A + k0, lda, tid);
      if (!SAMEB) g8_load_chunk(sm.B, Bm + k0, ldb, tid);
      __syncthreads();
      g8_mma_block<NR, NC, NEG, KSKIP, CREL>(sm.A, SAMEB ? sm.A : sm.B, acc, r0, c0, lane, k0);
    }
    return;
  }
  G8Regs ra, rb;
  g8_fetch_chunk(ra, A, lda, tid);
  if (!SAMEB) g8_fetch_chunk(rb, Bm, ldb, tid);
  for (int k0 = 0; k0 < K; k0 += GK_KC) {
    __syncthreads();  // everybody finished reading the previous chunk out of LDS
    g8_stash_chunk(sm.A, ra, tid);
    if (!SAMEB) g8_stash_chunk(sm.B, rb, tid);
    __syncthreads();
    if (k0 + GK_KC < K) {
      g8_fetch_chunk(ra, A + k0 + GK_KC, lda, tid);
      if (!SAMEB) g8_fetch_chunk(rb, Bm + k0 + GK_KC, ldb, tid);
    }
    g8_mma_block<NR, NC, NEG, KSKIP, CREL>(sm.A, SAMEB ? sm.A : sm.B, acc, r0, c0, lane, k0);
  }
}

// Generic per-wave MFMA block: acc[i][j] (+)= sum_k A[r0+16i+..][k] * B[c0+16j+..][k] over one 32-wide
// chunk, for an NR x NC grid of 16x16 output tiles whose top-left corner is (r0, c0) inside the
// 128x128 workgroup tile.
//
// Instruction choice (measured on MI355X, tools/mfma_issue_probe.hip): v_mfma_f64_16x16x4_f64 saturates
// at ~48-50 TFLOP/s (issue-limited, ~104 clk per instruction although the pipe is busy 64), whereas the
// four-block form v_mfma_f64_4x4x4_4b_f64 sustains 73-75 TFLOP/s (95 % of the 78.6 TF datasheet peak).
// Lane layout of the four-block form (tools/archive/mfma444_layout.hip):
//     A: lane = 16*k + 4*blk + i   B: lane = 16*k + 4*blk + j   C/D: lane = 16*i + 4*blk + j
// The four blocks are given the SAME 4x4 A sub-block (rows 4r..4r+3) and four adjacent 4-column groups
// of B, so one instruction produces a 4x16 strip of C:
//     A operand: lane l reads A[4r + (l&3)][k = l>>4]   (replicated over blk = (l>>2)&3)
//     B operand: lane l reads B[n = l&15][k = l>>4]     (the usual 16-row pattern)
//     C/D:       lane l holds C[4r + (l>>4)][l&15]
// i.e. element r of the 16x16x4 accumulator fragment (row (l>>4)+4r, col l&15): the C fragment
// layout, the tile loads/stores and the fused epilogues are unchanged.
//   NEG   : use -A (trailing update subtracts)
//   KSKIP : the B matrix is lower triangular (W_kk): column block j only needs k <= its last column
//   CREL  : tile (i, j) is computed only when j + CREL <= i (lower-triangular part of a diagonal
//           workgroup tile); CREL = -64 disables the test.
template <int NR, int NC, int NEG, int KSKIP, int CREL>
static __device__ __forceinline__ void g8_mma_block(const double* __restrict__ As, const double* __restrict__ Bs,
                                                    d4 (&acc)[NR][NC], int r0, int c0, int lane, int k0) {
  const int lr = lane & 15, lk = lane >> 4, l4 = lane & 3;
#pragma unroll
  for (int kk = 0; kk < GK_KC / 4; kk++) {
    double b[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) b[j] = Bs[(c0 + j * 16 + lr) * GK_LD + kk * 4 + lk];
#pragma unroll
    for (int i = 0; i < NR; i++) {
      double a[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const double av = As[(r0 + i * 16 + 4 * r + l4) * GK_LD + kk * 4 + lk];
        a[r] = NEG ? -av : av;
      }
#pragma unroll
      for (int j = 0; j < NC; j++) {
        if (KSKIP && (k0 + kk * 4 > c0 + j * 16 + 15)) continue;  // wave-uniform
        if (j + CREL > i) continue;                               // compile-time
#pragma unroll
        for (int r = 0; r < 4; r++)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[r], b[j], acc[i][j][r], 0, 0, 0);
      }
    }
  }
}

