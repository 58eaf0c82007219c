// Batched right-looking blocked Cholesky + fused forward substitution + LML
// (SURVEY.md 8a rows a2, a3).
//
// Replaces, per walker:  L = cholesky(K, lower=True)          sklearn/_gpr.py:587 (LAPACK dpotrf)
//                        alpha = cho_solve((L, True), y)      sklearn/_gpr.py:597 (dpotrs)
//                        -1/2 y.alpha - sum log diag L - n/2 log 2pi       sklearn/_gpr.py:609-613
// using  y^T K^-1 y = z^T z  with  z = L^-1 y  (one forward substitution, fused into the panel
// kernels; the back substitution is only needed for posterior builds, bgp_post.hip).
//
// Per outer step k (block size NB = 128), three launches batched over the B walkers:
//   potrf_kernel  one workgroup per walker: diagonal block in LDS -> L_kk, W_kk = L_kk^-1,
//                 z_k = W_kk y_k, running log-det and z^T z                         (LDS-bound)
//   trsm_kernel   one workgroup per 128-row panel block:  X_i = A_ik W_kk^T  (fp64 MFMA),
//                 y_i -= X_i z_k                                                   (MFMA)
//   syrk_kernel   one workgroup per trailing 128x128 tile: A_ij -= X_i X_j^T   (fp64 MFMA;
//                 the n^3/3 bulk -- the kernel the roofline fraction is quoted on)
// All three share one NT tile GEMM on v_mfma_f64_16x16x4_f64: 4 waves as 2x2, each wave a 64x64
// sub-tile = 4x4 MFMA tiles (128 accumulator VGPRs), operands staged through LDS in 128x32
// chunks with leading dimension 34 (conflict-free ds_read_b64 for the 16-row x 2-k lane pattern).
#include "bgp_common.h"
#include "bgp_device.h"

#include "bgp_gemm.h"

// ------------------------------------------------------------------------------------------
// potrf: diagonal block k of every walker.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) potrf_kernel(double* __restrict__ Kbuf, double* __restrict__ Wbuf,
                                                     double* __restrict__ yw, double* __restrict__ accb,
                                                     double* __restrict__ lml, int* __restrict__ status, int n,
                                                     int ld, size_t mstride, int ystride, int nblk, int k) {
  const int b = blockIdx.x;
  if (status[b] != 0) return;
  __shared__ double s[128 * BGP_TILE_LD];
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double* T = Kbuf + (size_t)b * mstride + (size_t)(k * 128) * ld + k * 128;

  for (int idx = tid; idx < 128 * 64; idx += 256) {
    const int row = idx >> 6, seg = idx & 63;
    d2 v = *reinterpret_cast<const d2*>(T + (size_t)row * ld + seg * 2);
    s[row * BGP_TILE_LD + seg * 2] = v.x;
    s[row * BGP_TILE_LD + seg * 2 + 1] = v.y;
  }
  __syncthreads();

  const int i = tid & 127, hh = tid >> 7;
  int failed = 0;
  // unblocked right-looking factorisation in LDS (dpotf2 order: sqrt, scale by reciprocal, rank-1)
  for (int j = 0; j < 128; j++) {
    const double dj2 = s[j * BGP_TILE_LD + j];
    if (!(dj2 > 0.0)) {  // also catches NaN; identical for every thread -> uniform exit
      failed = j + 1;
      break;
    }
    const double dj = sqrt(dj2);
    const double inv = 1.0 / dj;
    __syncthreads();
    if (hh == 0) {
      if (i == j)
        s[j * BGP_TILE_LD + j] = dj;
      else if (i > j)
        s[i * BGP_TILE_LD + j] *= inv;
    }
    __syncthreads();
    if (i > j) {
      const double li = s[i * BGP_TILE_LD + j];
      for (int c = j + 1 + hh; c <= i; c += 2) s[i * BGP_TILE_LD + c] -= li * s[c * BGP_TILE_LD + j];
    }
    __syncthreads();
  }
  if (failed) {
    if (tid == 0) {
      status[b] = k * 128 + failed;  // 1-based index of the failing pivot
      lml[b] = -INFINITY;            // sklearn/_gpr.py:588-589
    }
    return;
  }

  // log-det contribution and write-back of L_kk (zeros above the diagonal)
  red[tid] = (hh == 0) ? log(s[i * BGP_TILE_LD + i]) : 0.0;
  for (int idx = tid; idx < 128 * 128; idx += 256) {
    const int row = idx >> 7, col = idx & 127;
    T[(size_t)row * ld + col] = (col <= row) ? s[row * BGP_TILE_LD + col] : 0.0;
  }
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  const double logdet_blk = red[0];
  __syncthreads();

  // in-place inverse of the lower-triangular block (dtrti2 order, columns right to left):
  //   M_jj = 1/L_jj ;  M_ij = -M_jj * sum_{c=j+1..i} M_ic L_cj
  for (int j = 127; j >= 0; j--) {
    const double ajj = 1.0 / s[j * BGP_TILE_LD + j];
    double v = 0.0;
    if (i > j)
      for (int c = j + 1 + hh; c <= i; c += 2) v += s[i * BGP_TILE_LD + c] * s[c * BGP_TILE_LD + j];
    if (hh == 1) red[i] = v;
    __syncthreads();
    if (hh == 0) {
      if (i > j)
        s[i * BGP_TILE_LD + j] = -ajj * (v + red[i]);
      else if (i == j)
        s[j * BGP_TILE_LD + j] = ajj;
    }
    __syncthreads();
  }

  // W_kk out (dense 128x128, zeros above the diagonal)
  double* W = Wbuf + ((size_t)b * nblk + k) * (128 * 128);
  for (int idx = tid; idx < 128 * 128; idx += 256) {
    const int row = idx >> 7, col = idx & 127;
    W[idx] = (col <= row) ? s[row * BGP_TILE_LD + col] : 0.0;
  }
  // z_k = W_kk y_k
  double* yk = yw + (size_t)b * ystride + k * 128;
  if (hh == 0) red[i] = yk[i];
  __syncthreads();
  double z = 0.0;
  if (hh == 0) {
    for (int c = 0; c <= i; c++) z += s[i * BGP_TILE_LD + c] * red[c];
  }
  __syncthreads();
  if (hh == 0) yk[i] = z;
  red[tid] = (hh == 0) ? z * z : 0.0;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) {
    double ld = logdet_blk, zz = red[0];
    if (k > 0) {
      ld += accb[b * 4 + 0];
      zz += accb[b * 4 + 1];
    }
    accb[b * 4 + 0] = ld;
    accb[b * 4 + 1] = zz;
    if (k == nblk - 1) lml[b] = -0.5 * zz - ld - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
  }
}

// ------------------------------------------------------------------------------------------
// trsm: X_i = A_ik W_kk^T for every row block i > k, then y_i -= X_i z_k.
// ------------------------------------------------------------------------------------------
// Active row blocks below the diagonal at step k: the nlow = nblk-k-1 remaining blocks of K, then
// (posterior builds only) the first k+1 block rows of the identity part of the augmented matrix
// [[K, .], [I, 0]], which starts at block row `aug`.  Running the same three kernels on the
// augmented matrix for nblk steps leaves L (top-left), L^-T (bottom-left), the Schur complement
// -K^-1 (bottom-right) and -alpha = -(K^-1 y) in the lower half of the working right-hand side.
static __device__ __forceinline__ int bgp_rowblk(int t, int k, int nlow, int aug) {
  return (t < nlow) ? (k + 1 + t) : (aug + (t - nlow));
}

__global__ void __launch_bounds__(256) trsm_kernel(double* __restrict__ Kbuf, const double* __restrict__ Wbuf,
                                                    double* __restrict__ yw, const int* __restrict__ status,
                                                    int ld, size_t mstride, int ystride, int nblk, int k, int nact,
                                                    int aug, int B) {
  int b, t;
  bgp_map_block(blockIdx.x, nact, b, t);
  if (b >= B || status[b] != 0) return;
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const int ib = bgp_rowblk(t, k, nblk - k - 1, aug);
  double* Atile = Kbuf + (size_t)b * mstride + (size_t)(ib * 128) * ld + k * 128;
  const double* W = Wbuf + ((size_t)b * nblk + k) * (128 * 128);

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  if (tid < 128) sm.ypart[tid] = 0.0;

  for (int k0 = 0; k0 < 128; k0 += GK_KC) {
    __syncthreads();
    gk_load_chunk(sm.A, Atile + k0, (size_t)ld, tid);
    gk_load_chunk(sm.B, W + k0, (size_t)128, tid);
    __syncthreads();
    gk_mma_chunk<0, 1>(sm.A, sm.B, acc, wr, wc, lane, k0);
  }
  // In-place overwrite is safe: every global read of this A tile was staged into LDS before the
  // last chunk's barrier, and no other workgroup touches the tile in this launch.
  const double* zk = yw + (size_t)b * ystride + k * 128;
  double zc[4];
#pragma unroll
  for (int j = 0; j < 4; j++) zc[j] = zk[GK_COL(wc, j, lane)];
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = GK_ROW(wr, i, lane, r);
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const double x = acc[i][j][r];
        Atile[(size_t)row * ld + GK_COL(wc, j, lane)] = x;
        part += x * zc[j];
      }
      // reduce over the 16 lanes that share this row (lane & 15 varies)
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      if ((lane & 15) == 0) atomicAdd(&sm.ypart[row], part);
    }
  }
  __syncthreads();
  if (tid < 128) yw[(size_t)b * ystride + ib * 128 + tid] -= sm.ypart[tid];
}

// ------------------------------------------------------------------------------------------
// syrk: trailing update A_ij -= X_i X_j^T for k < j <= i.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) syrk_kernel(double* __restrict__ Kbuf, const int* __restrict__ status,
                                                    int ld, size_t mstride, int nblk, int k, int nact, int aug,
                                                    int B) {
  const int ntile = nact * (nact + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntile, b, t);
  if (b >= B || status[b] != 0) return;
  int ti, tj;
  bgp_tri_decode(t, ti, tj);
  const int I = bgp_rowblk(ti, k, nblk - k - 1, aug), J = bgp_rowblk(tj, k, nblk - k - 1, aug);
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  double* M = Kbuf + (size_t)b * mstride;
  const double* XI = M + (size_t)(I * 128) * ld + k * 128;
  const double* XJ = M + (size_t)(J * 128) * ld + k * 128;
  double* C = M + (size_t)(I * 128) * ld + J * 128;

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = C[(size_t)GK_ROW(wr, i, lane, r) * ld + GK_COL(wc, j, lane)];

  const bool diag = (I == J);
  for (int k0 = 0; k0 < 128; k0 += GK_KC) {
    __syncthreads();
    gk_load_chunk(sm.A, XI + k0, (size_t)ld, tid);
    if (!diag) gk_load_chunk(sm.B, XJ + k0, (size_t)ld, tid);
    __syncthreads();
    gk_mma_chunk<1, 0>(sm.A, diag ? sm.A : sm.B, acc, wr, wc, lane, k0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) C[(size_t)GK_ROW(wr, i, lane, r) * ld + GK_COL(wc, j, lane)] = acc[i][j][r];
}

// ------------------------------------------------------------------------------------------
int bgp_launch_cholesky(bgp_ctx* ctx, int B, int augmented) {
  // augmented == 0: LML only (matrices npad x npad).  augmented != 0: posterior build on the
  // (2 npad) x (2 npad) augmented matrices [[K, .], [I, 0]] (see bgp_rowblk).
  const int nblk = ctx->nblk, npad = ctx->npad;
  const int ld = augmented ? 2 * npad : npad;
  const size_t mstride = (size_t)ld * ld;
  const int ystride = ld;
  const int B8 = 8 * ((B + 7) / 8);
  for (int k = 0; k < nblk; k++) {
    bgp_tbegin(ctx, 1);
    hipLaunchKernelGGL(potrf_kernel, dim3(B), dim3(256), 0, ctx->stream, ctx->dK, ctx->dW, ctx->dyw, ctx->dacc,
                       ctx->dlml, ctx->dstatus, ctx->n, ld, mstride, ystride, nblk, k);
    bgp_tend(ctx);
    const int nlow = nblk - k - 1;
    const int nact = augmented ? nblk : nlow;
    if (nact > 0) {
      bgp_tbegin(ctx, 2);
      hipLaunchKernelGGL(trsm_kernel, dim3(B8 * nact), dim3(256), 0, ctx->stream, ctx->dK, ctx->dW, ctx->dyw,
                         ctx->dstatus, ld, mstride, ystride, nblk, k, nact, nblk, B);
      bgp_tend(ctx);
      bgp_tbegin(ctx, 3);
      hipLaunchKernelGGL(syrk_kernel, dim3(B8 * (nact * (nact + 1) / 2)), dim3(256), 0, ctx->stream, ctx->dK,
                         ctx->dstatus, ld, mstride, nblk, k, nact, nblk, B);
      bgp_tend(ctx);
    }
  }
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}
