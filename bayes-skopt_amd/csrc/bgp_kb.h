// Pieces of the kernel-matrix build shared by bgp_kbuild.hip (the Gram / cross kernels) and bgp_ps.hip (the launch-free
// factorisation's tile workers generating the Gram blocks of their own call).  gfx950 only.
#pragma once
#include "bgp_common.h"
#include "bgp_device.h"

#define KB_DK 16  // input dimensions staged per pass

// Epilogue of a 128 x 128 tile whose squared scaled distances sit in acc[r][c] (rows ty + 16 r, columns tx + 16 c):
// stationary kernel, constant, exact diagonal / identity padding (GRAM) or zero padding (cross matrices).
// R = rows per thread (8: 256 threads, rows ty + 16 r; 4: 512 threads, rows ty + 32 r).
template <int GRAM, int STAT, int FORM, int R = 8>
static __device__ __forceinline__ void kb_epilogue(double (&acc)[R][8], int na, int nb, int d,
                                                   const double* __restrict__ h, const double* __restrict__ alpha,
                                                   int i0, int j0, double* __restrict__ out, size_t ldo, int out_rows,
                                                   int out_cols, int tx, int ty) {
#pragma clang fp contract(off)
  const double cst = exp(h[0]);
  const bool interior = (i0 + 128 <= na) && (j0 + 128 <= nb) && (i0 + 128 <= out_rows) && (j0 + 128 <= out_cols) &&
                        !(GRAM && i0 == j0);
  if (interior) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      double* orow = out + (size_t)(i0 + ty + (128 / R) * r) * ldo + j0 + tx;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const double s = kb_stationary<STAT>(acc[r][c]);
        const double v = (FORM == BGP_FORM_PRODUCT) ? cst * s : cst + s;
        orow[16 * c] = v;
        if (!GRAM) acc[r][c] = v;  // (cross builds: the caller may go on with the values, see kbuild_cross_kernel)
      }
    }
    return;
  }
  const double s2 = exp(h[d + 1]);
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int gi = i0 + ty + (128 / R) * r;
    if (!GRAM) {
      // cross matrices are consumed by 128-tiled GEMMs: the tile's padding (rows >= out_rows, columns >= out_cols,
      // inside the 128-padded buffer) is written as zeros here, so no memset pass over the buffer is needed
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const int gj = j0 + tx + 16 * c;
        double v = 0.0;
        if (gi < out_rows && gj < out_cols) {
          const double sv = kb_stationary<STAT>(acc[r][c]);
          v = (FORM == BGP_FORM_PRODUCT) ? cst * sv : cst + sv;
        }
        out[(size_t)gi * ldo + gj] = v;
        acc[r][c] = v;
      }
      continue;
    }
    if (gi >= out_rows) continue;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gj = j0 + tx + 16 * c;
      if (gj >= out_cols) continue;
      double v;
      if (GRAM && (gi >= na || gj >= nb)) {
        v = (gi == gj) ? 1.0 : 0.0;  // identity padding: log det and z unaffected
      } else if (GRAM && gi == gj) {
        // fill_diagonal(1) (kernels.py:1738) -> c*1 (+1) -> + s2 (White) -> += alpha (_gpr.py:585)
        const double base = (FORM == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
        v = (base + s2);
        if (alpha) v += alpha[gi];
      } else {
        const double s = kb_stationary<STAT>(acc[r][c]);
        v = (FORM == BGP_FORM_PRODUCT) ? cst * s : cst + s;
      }
      out[(size_t)gi * ldo + gj] = v;
    }
  }
}

// One 128 x 128 Gram tile by a 512-thread workgroup (thread (tx, ty) of a 16 x 32 grid owns rows ty + 32 r, r < 4, columns
// tx + 16 c, c < 8), operands staged in caller-provided LDS (xi, xj: KB_DK x BGP_TILE_LD doubles each, ell: KB_DK).  Per element
// the arithmetic of kbuild_tile (x / l staged, (a - b)^2 accumulated with fma in ascending dimension, kb_epilogue): same bits.
template <int STAT, int FORM>
static __device__ __forceinline__ void kb_gram_tile512(const double* __restrict__ X, int n, int d, const double* __restrict__ h,
                                                       const double* __restrict__ alpha, int i0, int j0, double* __restrict__ out,
                                                       size_t ldo, int npad, double* xi, double* xj, double* ell) {
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) acc[r][c] = 0.0;
  for (int k0 = 0; k0 < d; k0 += KB_DK) {
    const int kc = min(KB_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += 512) {
      const int row = idx / kc, k = idx - row * kc;
      const int gi = i0 + row, gj = j0 + row;
      const double l = ell[k];
      xi[k * BGP_TILE_LD + row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / l : 0.0;
      xj[k * BGP_TILE_LD + row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / l : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[4], b[8];
#pragma unroll
      for (int r = 0; r < 4; r++) a[r] = xi[k * BGP_TILE_LD + ty + 32 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) b[c] = xj[k * BGP_TILE_LD + tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        double df[8];
#pragma unroll
        for (int c = 0; c < 8; c++) df[c] = a[r] - b[c];
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
        for (int c = 0; c < 8; c++) acc[r][c] = fma(df[c], df[c], acc[r][c]);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    }
  }
  kb_epilogue<1, STAT, FORM, 4>(acc, n, n, d, h, alpha, i0, j0, out, ldo, npad, npad, tx, ty);
}
