// Round-1 kernels of the blocked Cholesky, retired from libbgp.so in round 3 and kept as A/B references for the
// benches in tools/ (tools/build_syrk4_bench.sh, build_syrk_bench.sh, build_trsm_bench.sh): trsm_kernel / syrk_kernel
// (single panel, also the augmented matrix), syrk2_kernel (two-panel trailing update of the LML path; syrk4_kernel's
// results are bit-identical to it) and its launcher.  Compiled against the product headers (-I bayes-skopt_amd/csrc).
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm_legacy.h"

// ------------------------------------------------------------------------------------------
// trsm: X_i = A_ik W_kk^T for every row block i > k, then y_i -= X_i z_k.
// ------------------------------------------------------------------------------------------
// Active row blocks below the diagonal at step k: the nlow = nblk-k-1 remaining blocks of K, then
// (posterior builds only) the first k+1 block rows of the identity part of the augmented matrix
// [[K, .], [I, 0]], which starts at block row `aug`.  Running the same three kernels on the
// augmented matrix for nblk steps leaves L (top-left), L^-T (bottom-left), the Schur complement
// -K^-1 (bottom-right) and -alpha = -(K^-1 y) in the lower half of the working right-hand side.
// (bgp_rowblk: bgp_device.h)

__global__ void __launch_bounds__(256) trsm_kernel(double* __restrict__ Kbuf, const double* __restrict__ Wbuf,
                                                    double* __restrict__ yw, const int* __restrict__ status,
                                                    int ld, size_t mstride, int ystride, int nblk, int k, int nact,
                                                    int aug, int B) {
  int b, t;
  bgp_map_block(blockIdx.x, nact, B, b, t);
  if (b >= B || status[b] != 0) return;
  __shared__ GemmSmem sm;
  // 4 waves stacked along the rows (32 rows x 128 columns each): every wave sees the same
  // triangular structure of W_kk, so the k-skip leaves them equally loaded
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r0 = w * 32;
  const int ib = bgp_rowblk(t, k, nblk - k - 1, aug);
  double* Atile = Kbuf + (size_t)b * mstride + (size_t)(ib * 128) * ld + k * 128;
  const double* W = Wbuf + ((size_t)b * nblk + k) * (128 * 128);

  d4 acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  for (int k0 = 0; k0 < 128; k0 += GK_KC) {
    __syncthreads();
    gk_load_chunk(sm.A, Atile + k0, (size_t)ld, tid);
    gk_load_chunk(sm.B, W + k0, (size_t)128, tid);
    __syncthreads();
    gk_mma_block<2, 8, 0, 1, -64>(sm.A, sm.B, acc, r0, 0, lane, k0);
  }
  // In-place overwrite is safe: every global read of this A tile was staged into LDS before the
  // last chunk's barrier, and no other workgroup touches the tile in this launch.
  const double* zk = yw + (size_t)b * ystride + k * 128;
  double zc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) zc[j] = zk[GK_COLB(0, j, lane)];
  double* yi = yw + (size_t)b * ystride + ib * 128;
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = GK_ROWB(r0, i, lane, r);
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double x = acc[i][j][r];
        Atile[(size_t)row * ld + GK_COLB(0, j, lane)] = x;
        part += x * zc[j];
      }
      // reduce over the 16 lanes that share this row (lane & 15 varies); the row belongs to this
      // wave alone, so the right-hand side is updated directly
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      if ((lane & 15) == 0) yi[row] -= part;
    }
  }
}

static __device__ __forceinline__ void syrk_diag_tile(GemmSmem& sm, const double* __restrict__ XI,
                                                      double* __restrict__ C, int ld, int tid, int lane, int w,
                                                      int K = 128) {
  // wave 0 / 1: the two 64x64 triangles on the diagonal (10 MFMA tiles each);
  // wave 2 / 3: the 64x64 square below the diagonal cut into two 32x64 halves (8 MFMA tiles each).
  // Barriers and staging are common code; only the register block differs per wave.
  d4 acc[4][4];
  d4(&acc2)[2][4] = reinterpret_cast<d4(&)[2][4]>(acc);
  const int r0 = (w == 0) ? 0 : (w == 1) ? 64 : (w == 2) ? 64 : 96;
  const int c0 = (w == 1) ? 64 : 0;
  if (w < 2)
    gk_load_c<4, 4, 0>(C, (size_t)ld, acc, r0, c0, lane);
  else
    gk_load_c<2, 4, -64>(C, (size_t)ld, acc2, r0, c0, lane);
  for (int k0 = 0; k0 < K; k0 += GK_KC) {
    __syncthreads();
    gk_load_chunk(sm.A, XI + k0, (size_t)ld, tid);
    __syncthreads();
    if (w < 2)
      gk_mma_block<4, 4, 1, 0, 0>(sm.A, sm.A, acc, r0, c0, lane, k0);
    else
      gk_mma_block<2, 4, 1, 0, -64>(sm.A, sm.A, acc2, r0, c0, lane, k0);
  }
  if (w < 2)
    gk_store_c<4, 4, 0>(C, (size_t)ld, acc, r0, c0, lane);
  else
    gk_store_c<2, 4, -64>(C, (size_t)ld, acc2, r0, c0, lane);
}

// ------------------------------------------------------------------------------------------
// syrk: trailing update A_ij -= X_i X_j^T for k < j <= i.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) syrk_kernel(double* __restrict__ Kbuf, const int* __restrict__ status,
                                                    int ld, size_t mstride, int nblk, int k, int nact, int aug,
                                                    int B) {
  const int ntile = nact * (nact + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntile, B, b, t);
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

  if (I != J) {
    d4 acc[4][4];
    gk_load_c<4, 4, -64>(C, (size_t)ld, acc, wr * 64, wc * 64, lane);
    for (int k0 = 0; k0 < 128; k0 += GK_KC) {
      __syncthreads();
      gk_load_chunk(sm.A, XI + k0, (size_t)ld, tid);
      gk_load_chunk(sm.B, XJ + k0, (size_t)ld, tid);
      __syncthreads();
      gk_mma_block<4, 4, 1, 0, -64>(sm.A, sm.B, acc, wr * 64, wc * 64, lane, k0);
    }
    gk_store_c<4, 4, -64>(C, (size_t)ld, acc, wr * 64, wc * 64, lane);
  } else {
    // Diagonal tile: only its lower triangle (36 of the 64 16x16 MFMA tiles) is ever read again,
    // split 10 / 10 / 8 / 8 over the waves: two 4x4 triangles and the 4x4 square below the
    // diagonal cut in two.  X_I is staged once and serves as both operands.
    syrk_diag_tile(sm, XI, C, ld, tid, lane, w);
  }
}

// ------------------------------------------------------------------------------------------
// syrk2: trailing update of the LML path with a panel of width K = 128 or 256 (two factorised block
// columns at once: halves the C-tile traffic and the per-tile prologue/epilogue per flop).
//   A_IJ -= X_I X_J^T,  X_I = rows of block I, columns [kp*128, kp*128 + K)
//   colmode 1: only block column jstart (the look-ahead column the next potrf/trsm need)
//   colmode 0: every tile with I >= J >= jstart
// ------------------------------------------------------------------------------------------
template <int SPLIT>
__global__ void __launch_bounds__(256, 2) syrk2_kernel(double* __restrict__ Kbuf, const int* __restrict__ status,
                                                       int ld, size_t mstride, int nblk, int kp, int K, int jstart,
                                                       int colmode, int B) {
  const int nt = nblk - jstart;
  const int ntile = colmode ? nt : nt * (nt + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, SPLIT * ntile, B, b, t);
  if (b >= B || status[b] != 0) return;
  // SPLIT == 2 (small launches that would leave the GPU under-filled): two workgroups per tile, 64 rows each
  const int half = (SPLIT == 2) ? (t & 1) : 0;
  if (SPLIT == 2) t >>= 1;
  int ti, tj;
  if (colmode) {
    ti = t;
    tj = 0;
  } else {
    bgp_tri_decode(t, ti, tj);
  }
  const int I = jstart + ti, J = jstart + tj;
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  double* M = Kbuf + (size_t)b * mstride;
  const double* XI = M + (size_t)(I * 128) * ld + kp * 128;
  const double* XJ = M + (size_t)(J * 128) * ld + kp * 128;
  double* C = M + (size_t)(I * 128) * ld + J * 128;

  if (I != J) {
    if (SPLIT == 2) {
      // small launches are latency-bound: the next chunk's global loads are in flight (registers) while the
      // current one is multiplied -- the half-size accumulator block leaves room for the 64 staging VGPRs
      d4 acc[2][4];
      const int r0 = half * 64 + wr * 32;
      d2 va[8], vb[8];
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int c = tid + 256 * i;
        va[i] = *reinterpret_cast<const d2*>(XI + (size_t)(c >> 4) * ld + (c & 15) * 2);
        vb[i] = *reinterpret_cast<const d2*>(XJ + (size_t)(c >> 4) * ld + (c & 15) * 2);
      }
      gk_load_c<2, 4, -64>(C, (size_t)ld, acc, r0, wc * 64, lane);
      for (int k0 = 0; k0 < K; k0 += GK_KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const int c = tid + 256 * i;
          *reinterpret_cast<d2*>(sm.A + (c >> 4) * GK_LD + (c & 15) * 2) = va[i];
          *reinterpret_cast<d2*>(sm.B + (c >> 4) * GK_LD + (c & 15) * 2) = vb[i];
        }
        __syncthreads();
        if (k0 + GK_KC < K) {
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const int c = tid + 256 * i;
            va[i] = *reinterpret_cast<const d2*>(XI + k0 + GK_KC + (size_t)(c >> 4) * ld + (c & 15) * 2);
            vb[i] = *reinterpret_cast<const d2*>(XJ + k0 + GK_KC + (size_t)(c >> 4) * ld + (c & 15) * 2);
          }
        }
        gk_mma_block<2, 4, 1, 0, -64>(sm.A, sm.B, acc, r0, wc * 64, lane, k0);
      }
      gk_store_c<2, 4, -64>(C, (size_t)ld, acc, r0, wc * 64, lane);
    } else {
      d4 acc[4][4];
      gk_load_c<4, 4, -64>(C, (size_t)ld, acc, wr * 64, wc * 64, lane);
      for (int k0 = 0; k0 < K; k0 += GK_KC) {
        __syncthreads();
        gk_load_chunk(sm.A, XI + k0, (size_t)ld, tid);
        gk_load_chunk(sm.B, XJ + k0, (size_t)ld, tid);
        __syncthreads();
        gk_mma_block<4, 4, 1, 0, -64>(sm.A, sm.B, acc, wr * 64, wc * 64, lane, k0);
      }
      gk_store_c<4, 4, -64>(C, (size_t)ld, acc, wr * 64, wc * 64, lane);
    }
  } else {
    if (half) return;  // diagonal tiles (half the work of a full tile already) stay on one workgroup
    syrk_diag_tile(sm, XI, C, ld, tid, lane, w, K);
  }
}

// tiles x walkers below this many workgroups: split the off-diagonal tiles over two workgroups each
#define SYRK_SPLIT_BELOW 1024
static void launch_syrk2(hipStream_t st, int B8, int ntile, double* dK, const int* dstatus, int ld, size_t mstride,
                         int nblk, int kp, int K, int jstart, int colmode, int B) {
  if (B8 * ntile < SYRK_SPLIT_BELOW)
    hipLaunchKernelGGL(syrk2_kernel<2>, dim3(B8 * ntile * 2), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K,
                       jstart, colmode, B);
  else
    hipLaunchKernelGGL(syrk2_kernel<1>, dim3(B8 * ntile), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K,
                       jstart, colmode, B);
}

extern "C" void bgp_debug_launch_syrk2(hipStream_t st, int B8, int ntile, double* dK, const int* dstatus, int ld,
                                       size_t mstride, int nblk, int kp, int K, int jstart, int colmode, int B) {
  launch_syrk2(st, B8, ntile, dK, dstatus, ld, mstride, nblk, kp, K, jstart, colmode, B);
}

