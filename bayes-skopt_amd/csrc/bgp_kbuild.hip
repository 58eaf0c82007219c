// Kernel-matrix build: tiled pairwise-distance kernel staging X tiles in LDS.
//
// Replaces kernel_(X_train_) + diagonal add -- sklearn/kernels.py:1708-1738 (Matern.__call__),
// :1553-1560 (RBF), :966 (Product), :866 (Sum), :1273 (Constant), :1402 (White),
// sklearn/_gpr.py:585 / bask/bayesgpr.py:204 (K[diag] += alpha).
//
// One 256-thread workgroup produces one 128x128 tile; thread (tx,ty) of the 16x16 thread grid owns
// the 8x8 strided micro-tile rows ty+16r, cols tx+16c, so that for a fixed (r,c) the 16 lanes of a
// row write 128 contiguous bytes (full cache lines) and LDS reads of the column operand are
// conflict-free.  X tiles are pre-divided by the walker's length scales while being staged.
// HBM-bound by design: 8 B written per pair, X re-read from L2.
#include "bgp_common.h"
#include "bgp_device.h"

#define KB_DK 16  // input dimensions staged per pass

// Generic tile body: out[(i0+..)][(j0+..)] = k(A_i, B_j); A is (na x d), Bm is (nb x d), row-major.
// GRAM != 0: A == Bm is the training set, diagonal gets c(+1) + s2 + alpha_i, padding gets identity.
// Tiles that are fully inside the data and off the diagonal take a check-free epilogue.
template <int GRAM, int STAT, int FORM>
static __device__ __forceinline__ void kbuild_tile(const double* __restrict__ A, int na,
                                                   const double* __restrict__ Bm, int nb, int d,
                                                   const double* __restrict__ h, const double* __restrict__ alpha,
                                                   int i0, int j0, double* __restrict__ out, size_t ldo, int out_rows,
                                                   int out_cols) {
  __shared__ double xi[KB_DK][BGP_TILE_LD];
  __shared__ double xj[KB_DK][BGP_TILE_LD];
  __shared__ double ell[KB_DK];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  double acc[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) acc[r][c] = 0.0;

  for (int k0 = 0; k0 < d; k0 += KB_DK) {
    const int kc = min(KB_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += 256) {
      int row = idx / kc, k = idx - row * kc;
      int gi = i0 + row, gj = j0 + row;
      double l = ell[k];
      xi[k][row] = (gi < na) ? A[(size_t)gi * d + k0 + k] / l : 0.0;
      xj[k][row] = (gj < nb) ? Bm[(size_t)gj * d + k0 + k] / l : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[8], b[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) b[c] = xj[k][tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          double df = a[r] - b[c];
          acc[r][c] += df * df;
        }
    }
  }
  const double cst = exp(h[0]);
  const bool interior = (i0 + 128 <= na) && (j0 + 128 <= nb) && (i0 + 128 <= out_rows) && (j0 + 128 <= out_cols) &&
                        !(GRAM && i0 == j0);
  if (interior) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      double* orow = out + (size_t)(i0 + ty + 16 * r) * ldo + j0 + tx;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        const double s = kb_stationary<STAT>(acc[r][c]);
        orow[16 * c] = (FORM == BGP_FORM_PRODUCT) ? cst * s : cst + s;
      }
    }
    return;
  }
  const double s2 = exp(h[d + 1]);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int gi = i0 + ty + 16 * r;
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

template <int STAT, int FORM>
__global__ void __launch_bounds__(256) kbuild_gram_kernel(const double* __restrict__ X,
                                                           const double* __restrict__ alpha,
                                                           const double* __restrict__ H, double* __restrict__ Kbuf,
                                                           const double* __restrict__ y, double* __restrict__ yw,
                                                           int n, int d, int npad, int nblk, int B, int full, int ld,
                                                           int use_alpha, size_t xstride) {
  const int ntiles = full ? nblk * nblk : nblk * (nblk + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntiles, B, b, t);
  if (b >= B) return;
  int ti, tj;
  if (full) {
    ti = t / nblk;
    tj = t - ti * nblk;
  } else {
    bgp_tri_decode(t, ti, tj);
  }
  const double* h = H + (size_t)b * (d + 2);
  // ld == npad for LML batches, 2*npad for the augmented matrices of posterior builds
  double* out = Kbuf + (size_t)b * ld * ld;
  // working right-hand side of walker b (becomes z = L^-1 y during the factorisation)
  if (ti == tj && threadIdx.x < 128) yw[(size_t)b * ld + ti * 128 + threadIdx.x] = y[ti * 128 + threadIdx.x];
  const double* Xb = X + (size_t)b * xstride;  // per-walker warped inputs (xstride == 0: shared)
  kbuild_tile<1, STAT, FORM>(Xb, n, Xb, n, d, h, use_alpha ? alpha : nullptr, ti * 128, tj * 128, out, (size_t)ld, npad,
                             npad);
}

// blockIdx.y = item of a batch: hyper-parameters h + b (d+2), output out + b ostride (the inputs are shared)
template <int STAT, int FORM>
__global__ void __launch_bounds__(256) kbuild_cross_kernel(const double* __restrict__ Xq, int m,
                                                            const double* __restrict__ Xt, int n, int d,
                                                            const double* __restrict__ h, double* __restrict__ out,
                                                            int ldo, int tiles_j, size_t ostride) {
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j, b = blockIdx.y;
  kbuild_tile<0, STAT, FORM>(Xq, m, Xt, n, d, h + (size_t)b * (d + 2), nullptr, ti * 128, tj * 128,
                             out + (size_t)b * ostride, (size_t)ldo, m, n);
}

int bgp_launch_kbuild(bgp_ctx* ctx, int B, int full_square, int augmented, int use_alpha) {
  return bgp_launch_kbuild_slice(ctx, 0, B, ctx->stream, full_square, augmented, use_alpha);
}

int bgp_launch_kbuild_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented,
                            int use_alpha) {
  return bgp_launch_kbuild_x(ctx, off, B, st, full_square, augmented, use_alpha, ctx->dXeff, 0);
}

int bgp_launch_kbuild_x(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented, int use_alpha,
                        const double* dXb, size_t xstride) {
  const int nblk = ctx->nblk;
  const size_t ldm = augmented ? 2 * (size_t)ctx->npad : (size_t)ctx->npad;
  const int ntiles = full_square ? nblk * nblk : nblk * (nblk + 1) / 2;
  const int grid = 8 * ((B + 7) / 8) * ntiles;
  bgp_tbegin(ctx, 0, st);
  KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
              hipLaunchKernelGGL((kbuild_gram_kernel<S, F>), dim3(grid), dim3(256), 0, st, dXb, ctx->dalpha,
                                 ctx->dh + (size_t)off * (ctx->d + 2), ctx->dK + (size_t)off * ldm * ldm, ctx->dy,
                                 ctx->dyw + (size_t)off * ldm, ctx->n, ctx->d, ctx->npad, nblk, B, full_square,
                                 (int)ldm, use_alpha, xstride));
  bgp_tend(ctx, st);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

int bgp_launch_kcross(bgp_ctx* ctx, const double* dh_b, int m, const double* dXq, int nx, const double* dXt,
                      double* dout, int ldo, int /*unused*/) {
  return bgp_launch_kcross_batch(ctx, 1, dh_b, m, dXq, nx, dXt, dout, ldo, 0);
}

int bgp_launch_kcross_batch(bgp_ctx* ctx, int nb, const double* dH, int m, const double* dXq, int nx, const double* dXt,
                            double* dout, int ldo, size_t ostride) {
  const int tiles_i = (m + 127) / 128, tiles_j = (nx + 127) / 128;
  KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
              hipLaunchKernelGGL((kbuild_cross_kernel<S, F>), dim3(tiles_i * tiles_j, nb), dim3(256), 0, ctx->stream, dXq,
                                 m, dXt, nx, ctx->d, dH, dout, ldo, tiles_j, ostride));
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}
