// Accumulator-fragment helpers of v_mfma_f64_16x16x4_f64 shared by the ring kernels (bgp_syrk4.hip, bgp_post.hip).
#pragma once
#include "bgp_common.h"

// C/D fragment layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
// (cdna_hip_programming.md:247-251; verified at run time by bgp_mfma_f64_layout + tests).
#define GK_ROW(wr, i, lane, r) ((wr) * 64 + (i) * 16 + ((lane) >> 4) + 4 * (r))
#define GK_COL(wc, j, lane) ((wc) * 64 + (j) * 16 + ((lane) & 15))
#define GK_ROWB(r0, i, lane, r) ((r0) + (i) * 16 + ((lane) >> 4) + 4 * (r))
#define GK_COLB(c0, j, lane) ((c0) + (j) * 16 + ((lane) & 15))

// C tile <-> accumulators for an NR x NC block (same validity rule as gk_mma_block).
template <int NR, int NC, int CREL>
static __device__ __forceinline__ void gk_load_c(const double* __restrict__ C, size_t ld, d4 (&acc)[NR][NC], int r0,
                                                 int c0, int lane) {
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) {
      if (j + CREL > i) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = C[(size_t)GK_ROWB(r0, i, lane, r) * ld + GK_COLB(c0, j, lane)];
    }
}
template <int NR, int NC, int CREL>
static __device__ __forceinline__ void gk_store_c(double* __restrict__ C, size_t ld, const d4 (&acc)[NR][NC], int r0,
                                                  int c0, int lane) {
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) {
      if (j + CREL > i) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) C[(size_t)GK_ROWB(r0, i, lane, r) * ld + GK_COLB(c0, j, lane)] = acc[i][j][r];
    }
}
