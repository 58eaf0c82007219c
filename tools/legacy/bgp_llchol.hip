// EXPERIMENTAL (enabled with BGP_LEFT_LOOKING=1; the default LML path is the right-looking two-panel
// schedule of bgp_chol.hip, which is faster on every measured configuration -- docs/EXPERIMENTS.md).
// Left-looking blocked Cholesky of the LML path on the fast fp64 MFMA form (v_mfma_f64_4x4x4_4b_f64).
// trsm8_kernel below IS on the default path.
//
// Step J (block column J, nb = 128), batched over the B walkers, three launches:
//   lupdate_kernel  one 512-thread workgroup per tile (I, J), I >= J:
//                     T(I,J) = K(I,J) - sum_{k<J} L(I,k) L(J,k)^T
//                   The K tile is GENERATED in the accumulator registers (the K-build of
//                   bgp_kbuild.hip fused: no separate pass, no HBM round trip of the Gram matrix), the
//                   sum runs over the whole row panels (K = 128 J) with register prefetch, and the
//                   tile is written ONCE.  Compared with the right-looking update (C tile read +
//                   written at every step: 256 KB of HBM traffic per 4.2 MF, i.e. on the MFMA/HBM
//                   ridge) this needs 128 KB of reads per 4.2 MF (the A panel; the B panel of a column
//                   is shared by its tiles and hits the XCD's L2) and no writes until the end.
//   potrf_kernel    diagonal block (bgp_chol.hip, unchanged)
//   trsm8_kernel    X(I,J) = T(I,J) W_JJ^T for I > J, fused right-hand-side update y_I -= X z_J
// The flops are the same n^3/3; what changes is where the bytes go and which MFMA form executes them.
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm_legacy.h"
#include "bgp_gemm8.h"

#define LU_DK 16

// K(I,J) tile straight into the MFMA C-fragment layout of this wave's 32x64 block:
// acc[i][j][r] <-> row r0 + 16 i + (lane>>4) + 4 r, col c0 + 16 j + (lane&15).
template <int STAT, int FORM, int NR, int NC, int THREADS>
static __device__ __forceinline__ void lu_generate_tile(GemmSmem& sm, const double* __restrict__ X, int n, int d,
                                                        const double* __restrict__ h,
                                                        const double* __restrict__ alpha, int I, int J,
                                                        d4 (&acc)[NR][NC], int r0, int c0, int tid, int lane) {
  double(*xi)[BGP_TILE_LD] = reinterpret_cast<double(*)[BGP_TILE_LD]>(sm.A);  // [LU_DK][129]
  double(*xj)[BGP_TILE_LD] = reinterpret_cast<double(*)[BGP_TILE_LD]>(sm.B);
  double* ell = sm.ypart;
  const int lr = lane & 15, lk = lane >> 4;
  const int i0 = I * 128, j0 = J * 128;
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < d; k0 += LU_DK) {
    const int kc = min(LU_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += THREADS) {
      const int row = idx / kc, k = idx - row * kc;
      const int gi = i0 + row, gj = j0 + row;
      const double l = ell[k];
      xi[k][row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / l : 0.0;
      xj[k][row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / l : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[NR][4], b[NC];
#pragma unroll
      for (int i = 0; i < NR; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) a[i][r] = xi[k][r0 + 16 * i + lk + 4 * r];
#pragma unroll
      for (int j = 0; j < NC; j++) b[j] = xj[k][c0 + 16 * j + lr];
#pragma unroll
      for (int i = 0; i < NR; i++)
#pragma unroll
        for (int j = 0; j < NC; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double df = a[i][r] - b[j];
            acc[i][j][r] += df * df;
          }
    }
  }
  __syncthreads();  // staging buffers are reused by the main loop
  const double cst = exp(h[0]);
  const bool interior = (i0 + 128 <= n) && (j0 + 128 <= n) && (I != J);
  if (interior) {
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double s = kb_stationary<STAT>(acc[i][j][r]);
          acc[i][j][r] = (FORM == BGP_FORM_PRODUCT) ? cst * s : cst + s;
        }
    return;
  }
  const double s2 = exp(h[d + 1]);
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = i0 + GK_ROWB(r0, i, lane, r), gj = j0 + GK_COLB(c0, j, lane);
        double v;
        if (gi >= n || gj >= n) {
          v = (gi == gj) ? 1.0 : 0.0;  // identity padding
        } else if (gi == gj) {
          // fill_diagonal(1) (kernels.py:1738) -> c*1 (+1) -> + s2 (White) -> += alpha (_gpr.py:585)
          const double base = (FORM == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
          v = (base + s2) + alpha[gi];
        } else {
          const double s = kb_stationary<STAT>(acc[i][j][r]);
          v = (FORM == BGP_FORM_PRODUCT) ? cst * s : cst + s;
        }
        acc[i][j][r] = v;
      }
}

template <int STAT, int FORM>
__global__ void __launch_bounds__(G8_THREADS) lupdate_kernel(const double* __restrict__ X,
                                                              const double* __restrict__ alpha,
                                                              const double* __restrict__ H,
                                                              double* __restrict__ Kbuf,
                                                              const double* __restrict__ y, double* __restrict__ yw,
                                                              const int* __restrict__ status, int n, int d, int ld,
                                                              size_t mstride, int nblk, int J, int B) {
  const int ntile = nblk - J;
  int b, t;
  bgp_map_block(blockIdx.x, ntile, B, b, t);
  if (b >= B || status[b] != 0) return;
  const int I = J + t;
  __shared__ GemmSmem sm;
  // 8 waves as 4 (rows) x 2 (cols): wave (wr, wc) owns the 32 x 64 block at (32 wr, 64 wc)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r0 = (w >> 1) * 32, c0 = (w & 1) * 64;
  double* M = Kbuf + (size_t)b * mstride;
  // working right-hand side: every block is initialised at step 0 (the panel solves of the earlier
  // steps already subtract from the blocks below them)
  if (J == 0 && tid < 128) yw[(size_t)b * ld + I * 128 + tid] = y[I * 128 + tid];

  d4 acc[2][4];
  lu_generate_tile<STAT, FORM, 2, 4, G8_THREADS>(sm, X, n, d, H + (size_t)b * (d + 2), alpha, I, J, acc, r0, c0, tid,
                                                 lane);
  const int K = J * 128;
  if (K > 0) {
    const double* LI = M + (size_t)(I * 128) * ld;  // row panel L(I, 0 : K)
    const double* LJ = M + (size_t)(J * 128) * ld;  // row panel L(J, 0 : K)
    if (I != J)
      g8_mainloop<2, 4, 1, 0, -64, 0>(sm, LI, (size_t)ld, LJ, (size_t)ld, K, acc, r0, c0, tid, lane);
    else
      g8_mainloop<2, 4, 1, 0, -64, 1>(sm, LI, (size_t)ld, LI, (size_t)ld, K, acc, r0, c0, tid, lane);
  }
  gk_store_c<2, 4, -64>(M + (size_t)(I * 128) * ld + J * 128, (size_t)ld, acc, r0, c0, lane);
}

// Panel solve on the 8-wave 4x4x4 core: X = T W^T (W lower triangular: k-skip), y_I -= X z_J.
__global__ void __launch_bounds__(G8_THREADS, 4) trsm8_kernel(double* __restrict__ Kbuf, const double* __restrict__ Wbuf,
                                                            double* __restrict__ yw, const int* __restrict__ status,
                                                            int ld, size_t mstride, int ystride, int nblk, int k,
                                                            int B) {
  const int nrb = nblk - k - 1;
  int b, t;
  bgp_map_block(blockIdx.x, nrb, B, b, t);
  if (b >= B || status[b] != 0) return;
  __shared__ GemmSmem sm;
  // 8 waves stacked along the rows (16 rows x 128 columns each): every wave sees the same triangular
  // structure of W_kk, so the k-skip leaves them equally loaded
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r0 = w * 16;
  const int ib = k + 1 + t;
  double* Atile = Kbuf + (size_t)b * mstride + (size_t)(ib * 128) * ld + k * 128;
  const double* W = Wbuf + ((size_t)b * nblk + k) * (128 * 128);
  d4 acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[0][j] = (d4){0.0, 0.0, 0.0, 0.0};
  g8_mainloop<1, 8, 0, 1, -64, 0>(sm, Atile, (size_t)ld, W, (size_t)128, 128, acc, r0, 0, tid, lane);
  // in-place overwrite is safe: the whole A tile was staged through LDS before the last chunk's MFMAs
  const double* zk = yw + (size_t)b * ystride + k * 128;
  double zc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) zc[j] = zk[GK_COLB(0, j, lane)];
  double* yi = yw + (size_t)b * ystride + ib * 128;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = GK_ROWB(r0, 0, lane, r);
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const double x = acc[0][j][r];
      Atile[(size_t)row * ld + GK_COLB(0, j, lane)] = x;
      part += x * zc[j];
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    part += __shfl_xor(part, 8);
    if ((lane & 15) == 0) yi[row] -= part;  // the row belongs to this wave alone
  }
}

void bgp_launch_trsm8(hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus, int ld, size_t mstride,
                      int ystride, int nblk, int k) {
  const int B8 = 8 * ((B + 7) / 8);
  hipLaunchKernelGGL(trsm8_kernel, dim3(B8 * (nblk - k - 1)), dim3(G8_THREADS), 0, st, dK, dW, dyw, dstatus, ld, mstride,
                     ystride, nblk, k, B);
}

// potrf_kernel lives in bgp_chol.hip
void bgp_launch_potrf(bgp_ctx* ctx, hipStream_t st, int B, double* dK, double* dW, double* dyw, double* dacc,
                      double* dlml, int* dstatus, int ld, size_t mstride, int ystride, int k);

int bgp_launch_cholesky_ll_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int use_alpha) {
  const int nblk = ctx->nblk, npad = ctx->npad, ld = npad;
  const size_t mstride = (size_t)ld * ld;
  const int B8 = 8 * ((B + 7) / 8);
  double* dK = ctx->dK + (size_t)off * mstride;
  double* dW = ctx->dW + (size_t)off * nblk * (128 * 128);
  double* dyw = ctx->dyw + (size_t)off * ld;
  double* dacc = ctx->dacc + (size_t)off * 4;
  double* dlml = ctx->dlml + off;
  int* dstatus = ctx->dstatus + off;
  const double* dH = ctx->dh + (size_t)off * (ctx->d + 2);
  const double* dalpha = ctx->dalpha;  // (use_alpha is always 1 on the LML path)
  (void)use_alpha;
  for (int J = 0; J < nblk; J++) {
    bgp_tbegin(ctx, J == 0 ? 0 : 3, st);  // step 0 is pure kernel-matrix generation
    KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
                hipLaunchKernelGGL((lupdate_kernel<S, F>), dim3(B8 * (nblk - J)), dim3(G8_THREADS), 0, st, ctx->dXeff,
                                   dalpha, dH, dK, ctx->dy, dyw, dstatus, ctx->n, ctx->d, ld, mstride, nblk, J, B));
    bgp_tend(ctx, st);
    bgp_tbegin(ctx, 1, st);
    bgp_launch_potrf(ctx, st, B, dK, dW, dyw, dacc, dlml, dstatus, ld, mstride, ld, J);
    bgp_tend(ctx, st);
    if (J < nblk - 1) {
      bgp_tbegin(ctx, 2, st);
      hipLaunchKernelGGL(trsm8_kernel, dim3(B8 * (nblk - J - 1)), dim3(G8_THREADS), 0, st, dK, dW, dyw, dstatus, ld,
                         mstride, ld, nblk, J, B);
      bgp_tend(ctx, st);
    }
  }
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}
