// Trailing update of the LML path, software-pipelined:  A_IJ -= X_I X_J^T  over a panel of width K
// (the n^3/3 bulk of cholesky(K, lower=True), sklearn/_gpr.py:587).
//
// Same arithmetic and the same per-element summation order as syrk2_kernel (bgp_chol.hip) -- results are
// bit-identical to it -- but the operands reach the MFMAs differently.  syrk2 stages a 128x32 chunk through
// VGPRs between two barriers and waits for the global loads in the open (A, then B: two exposed L2 round
// trips per chunk).  Here:
//   * `global_load_lds_dwordx4` (LDS-DMA) writes 16-wide k-chunks straight into a two-stage LDS ring; the
//     loads of chunk c+1 are in flight while chunk c is multiplied, ONE barrier per chunk, no staging VGPRs,
//     no ds_write instructions, scalar-only address arithmetic in the loop;
//   * the LDS image is the DMA's lane-linear one (row-major [T][16] doubles, one wave instruction = 8 rows
//     x 128 B); bank conflicts are removed by an XOR swizzle of the 16-byte granules applied to the per-lane
//     SOURCE address and again on every fragment read (cdna_hip_programming.md 5.4 rule 21):
//         element (row, k) lives at byte  row*128 + (((k>>1) ^ ((row>>1)&7)) << 4) + ((k&1) << 3);
//   * the subtraction rides on the MFMA's A-negate modifier (blgp = 1 on the f64 forms): no VALU in the loop;
//   * fragments of k-step kk+1 are read while k-step kk multiplies;
//   * tile edge T = 64 (each wave a 32x32 block, 70 VGPRs, 32 KB of LDS: FOUR workgroups = 16 waves per CU -- the LDS is handed out
//     in granules of 1 280 B, a 32 768-byte workgroup takes 26 of a CU's 128: tools/wg_launch_probe.hip, round 6)
//     is what the library launches: measured on MI355X it ties the 128x128 tile (two workgroups per CU) on the
//     largest launch of BASELINE config C (62.6 vs 63.0 TF) although it moves twice the bytes per flop from
//     L2 and LDS, and wins everywhere else (B=16: 62.5 vs 55.4 TF; 320 tiles: 43 vs 28 TF) because small
//     launches fill the chip 4x better and the deeper occupancy hides the per-tile C round trip.  T = 128 is
//     kept as a template instantiation for tools/syrk4_bench.hip.  A persistent variant (workgroups walking
//     the tile list, next tile's first chunk in flight during the epilogue) was measured 8-12 % SLOWER: the
//     resident workgroups start in lockstep and stay there, so their C-tile traffic comes in bursts.
// Results do not depend on T or the launch geometry: every C element is owned by one lane and accumulated in
// the same k order (k ascending in steps of 4) -- bitwise reproducible and batch-split invariant.
// (The ring primitives live in bgp_s4.h; the tile workers of the launch-free factorisation, which use them too, in bgp_ps.hip.)
#include "bgp_s4.h"

// GENF = S4GenF<STAT, FORM>: the launch touches its blocks FIRST (panel group 0): C is generated, not loaded (bgp_s4.h).
template <int T, int VAR, class GENF = S4NoGen>
__global__ void __launch_bounds__(256, (T == 128) ? 2 : 4)
    syrk4_kernel(double* __restrict__ Kbuf, const int* __restrict__ status, int ld, size_t mstride, int nblk, int kp,
                 int K, int jstart, int colmode, int B, int total, unsigned long long* __restrict__ trace, int pw, GENF genf) {
  constexpr unsigned STAGEB = 2 * T * S4_ROWB;
  constexpr int NRF = T / 32;  // MFMA tiles per wave and direction (each wave a T/2 x T/2 block)
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGEB];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const int nt128 = (colmode == 2) ? nblk : nblk - jstart;  // (colmode 2: all nblk active rows of the augmented step)
  const S4Tile cur = s4_decode<T>(blockIdx.x, total, s4_ntile<T>(nt128, colmode == 2 ? 0 : colmode), Kbuf, status, ld,
                                  mstride, kp, jstart, colmode, nt128, B, pw);
  if (cur.q >= total) return;
  unsigned voff[T / 32];
  s4_src<T>(voff, ld, w, lane);
  if ((VAR & 4) && trace && threadIdx.x == 0) {
    trace[(size_t)cur.q * 8 + 4] = wall_clock64();                               // 100 MHz constant clock
    trace[(size_t)cur.q * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    trace[(size_t)cur.q * 8 + 6] = (unsigned long long)cur.label;
  }
  S4_STAMP(0);
  if (!cur.diag) {
    s4_tile<T, NRF, NRF, -64, VAR, 1, 0, GENF>(trace, lds0, cur, voff, ld, K, wr * (T / 2), wc * (T / 2), w, lane, 0, genf);
  } else if (w < 2) {
    // Diagonal tile: only its lower triangle is ever read again.  Waves 0 / 1: the two (T/2)^2 triangles on the
    // diagonal; waves 2 / 3: the square below the diagonal cut into two row halves (3/3/2/2 MFMA tiles at T = 64).
    s4_tile<T, NRF, NRF, 0, VAR, 1, 0, GENF>(trace, lds0, cur, voff, ld, K, w * (T / 2), w * (T / 2), w, lane, 0, genf);
  } else {
    s4_tile<T, NRF / 2, NRF, -64, VAR, 1, 0, GENF>(trace, lds0, cur, voff, ld, K, T / 2 + (w - 2) * (T / 4), 0, w, lane, 0, genf);
  }
  S4_STAMP(3);
  if ((VAR & 4) && trace && threadIdx.x == 0) trace[(size_t)cur.q * 8 + 7] = wall_clock64();
}

void bgp_launch_syrk4(hipStream_t st, int B8, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                      int K, int jstart, int colmode, int B) {
  const int total = B8 * (colmode == 2 ? s4_ntile<64>(nblk, 0) : s4_ntile<64>(nblk - jstart, colmode));
  const int pw = S4_PW;  // tile columns per L2-resident column panel (s4_panel_decode; 4 .. 16 measured within 2 %)
  hipLaunchKernelGGL((syrk4_kernel<64, 0>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K, jstart,
                     colmode, B, total, nullptr, pw, S4NoGen());
}

// The same launch for blocks no kernel has written yet: their Gram values are generated in the accumulators (S4GenF).
void bgp_launch_syrk4_gen(hipStream_t st, int B8, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                          int K, int jstart, int colmode, int B, const S4Gen& gen, int stationary, int form) {
  const int total = B8 * s4_ntile<64>(nblk - jstart, colmode);
  const int pw = S4_PW;
  KB_DISPATCH(stationary, form,
              hipLaunchKernelGGL((syrk4_kernel<64, 0, S4GenF<S, F>>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride,
                                 nblk, kp, K, jstart, colmode, B, total, nullptr, pw, S4GenF<S, F>{gen}));
}

// ------------------------------------------------------------------------------------------
// Panel solve on the same LDS-DMA ring:  X_i = A_ik W_kk^T  for every row block i > k (W_kk = L_kk^-1 from potrf,
// lower triangular), fused with the right-hand-side update  y_i -= X_i z_k  (the forward substitution of
// cho_solve, sklearn/_gpr.py:597).  One workgroup per 64 rows x all 128 columns of a row block (so the in-place
// overwrite stays inside the rows a workgroup has staged completely), four waves stacked along the rows (16 rows x
// 128 columns each: every wave sees the same triangular structure of W_kk, so the k-skip leaves them equally
// loaded), 48 KB of LDS: three workgroups per CU.  Chunk c (k in [16c, 16c+16)) only reaches the column blocks
// j >= c of W_kk^T.
// Replaces trsm8_kernel (VGPR staging between two barriers per chunk, 10 registers spilled at its 128-VGPR cap).
// ------------------------------------------------------------------------------------------
template <int VAR>
__global__ void __launch_bounds__(256, 3)
    trsm4_kernel(double* __restrict__ Kbuf, const double* __restrict__ Wbuf, double* __restrict__ yw,
                 const int* __restrict__ status, int ld, size_t mstride, int ystride, int nblk, int k, int B,
                 int augmented = 0) {
  constexpr unsigned AOPB = 64 * S4_ROWB, STAGEB = (64 + 128) * S4_ROWB;
  const int nrb = augmented ? nblk : nblk - k - 1;  // augmented matrix: nblk active row blocks at every step (bgp_rowblk)
  int b, t;
  bgp_map_block(blockIdx.x, 2 * nrb, B, b, t);
  if (b >= B || status[b] != 0) return;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGEB];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ib = augmented ? bgp_rowblk(t >> 1, k, nblk - k - 1, nblk) : k + 1 + (t >> 1), half = t & 1;
  double* A = Kbuf + (size_t)b * mstride + (size_t)(ib * 128 + half * 64) * ld + k * 128;
  const double* W = Wbuf + ((size_t)b * nblk + k) * (128 * 128);
  unsigned voffA[2], voffW[4];
  s4_src<64>(voffA, ld, w, lane);
  s4_src<128>(voffW, 128, w, lane);
  const int r0 = w * 16;
  unsigned pa[4], pb[4];
  s4_frag_addr(pa, lds0, r0, lane);
  s4_frag_addr(pb, lds0 + AOPB, 0, lane);
  d4 acc[1][8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[0][j] = (d4){0.0, 0.0, 0.0, 0.0};
  if (!(VAR & 1)) {
    s4_issue<64>(A, voffA, 0, lds0, w);
    s4_issue<128>(W, voffW, 0, lds0 + AOPB, w);
  }
  for (int c = 0; c < 8; c += 2) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      S4_WAIT_VM0();
      __builtin_amdgcn_s_barrier();
      if (!(VAR & 1) && c + s + 1 < 8) {
        const unsigned nb = lds0 + (unsigned)((s ^ 1) * STAGEB);
        s4_issue<64>(A, voffA, (c + s + 1) * S4_KC, nb, w);
        if (!(VAR & 16)) s4_issue_from<128>(W, voffW, (c + s + 1) * S4_KC, nb + AOPB, w, 16 * (c + s + 1));
      }
      s4_mma<1, 8, -64, VAR, 0>(pa, pb, s * STAGEB, acc, c + s);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Every read of this workgroup's A rows was staged through LDS before the last barrier (all waves passed it
  // after landing their own share): overwrite in place.  A row belongs to one wave: 16 lanes share it, fixed
  // shuffle order, and the right-hand side is updated directly -- bitwise reproducible.
  const double* zk = yw + (size_t)b * ystride + k * 128;
  double zc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) zc[j] = zk[GK_COLB(0, j, lane)];
  double* yi = yw + (size_t)b * ystride + ib * 128 + half * 64;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = GK_ROWB(r0, 0, lane, r);
    double part = 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const double x = acc[0][j][r];
      if (!(VAR & 8)) A[(size_t)row * ld + GK_COLB(0, j, lane)] = x;
      part += x * zc[j];
    }
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    part += __shfl_xor(part, 8);
    if ((lane & 15) == 0) yi[row] -= part;
  }
}

void bgp_launch_trsm4(hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus, int ld, size_t mstride,
                      int ystride, int nblk, int k, int augmented) {
  const int B8 = 8 * ((B + 7) / 8);
  const int nrb = augmented ? nblk : nblk - k - 1;
  hipLaunchKernelGGL(trsm4_kernel<0>, dim3(B8 * 2 * nrb), dim3(256), 0, st, dK, dW, dyw, dstatus, ld, mstride, ystride, nblk,
                     k, B, augmented);
}

// ------------------------------------------------------------------------------------------
// General NT product on the same ring for the posterior consumers (sample_y, predictive covariances):
//     MODE 0:  C  = A B^T          (C not read)             P = K_* K^-1
//     MODE 1:  C -= A B^T  on the tiles with ti >= tj only   cov = K_** - P K_*^T when only a Cholesky reads it
//     MODE 2:  C -= A B^T  on every tile                     the full predictive covariance of predict(return_cov)
// A (M x K) and B (N x K) share the leading dimension ldx, C (M x N) has ldc; M, N multiples of 64, K of 16.
// 64 x 64 tiles, four waves of 32 x 32; blockIdx.y = item of a batch (strides sA, sB, sC; pidxB maps item -> B slot).
// Tiles go by column panels of 8 (the B rows of the running tiles stay in L2, A streams once per panel).
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ void __launch_bounds__(256, 4)
    gemm4_kernel(const double* __restrict__ A, const double* __restrict__ Bm, int ldx, int K, double* __restrict__ C,
                 int ldc, int tm, int tn, size_t sA, size_t sB, size_t sC, const int* __restrict__ pidxB) {
  constexpr int T = 64;
  constexpr unsigned STAGEB = 2 * T * S4_ROWB;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGEB];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const int b = blockIdx.y;
  int ti, tj;
  if (MODE == 1) {
    s4_panel_decode((int)blockIdx.x, tm, ti, tj);  // lower triangle of a square tile grid (tm == tn)
  } else {
    const int t = blockIdx.x, per = S4_PW * tm, p = t / per, r = t - p * per;
    const int wdt = min(S4_PW, tn - p * S4_PW);  // (the last panel may be narrower)
    ti = r / wdt;
    tj = p * S4_PW + (r - ti * wdt);
    if (ti >= tm) return;
  }
  S4Tile cur;
  cur.XA = A + (size_t)b * sA + (size_t)(ti * T) * ldx;
  cur.XB = Bm + (size_t)(pidxB ? pidxB[b] : b) * sB + (size_t)(tj * T) * ldx;
  cur.C = C + (size_t)b * sC + (size_t)(ti * T) * ldc + tj * T;
  cur.diag = 0;
  cur.b = b, cur.gi0 = ti * T, cur.gj0 = tj * T;
  cur.q = 0, cur.label = 0;
  unsigned voff[T / 32];
  s4_src<T>(voff, ldx, w, lane);
  s4_tile<T, 2, 2, -64, 0, (MODE != 0) ? 1 : 0, (MODE == 0) ? 1 : 0>(nullptr, lds0, cur, voff, ldx, K, wr * (T / 2), wc * (T / 2), w,
                                                                        lane, ldc);
}

// mode 0: C = A B^T (all tiles); mode 1: C -= A B^T on the lower tiles of a square C; mode 2: C -= A B^T on all tiles.
// nb items (blockIdx.y).
void bgp_launch_gemm4(hipStream_t st, int mode, const double* A, const double* Bm, int ldx, int M, int N, int K, double* C,
                      int ldc, int nb, size_t sA, size_t sB, size_t sC, const int* pidxB) {
  const int tm = M / 64, tn = N / 64;
  if (mode == 1) {
    hipLaunchKernelGGL(gemm4_kernel<1>, dim3(tm * (tm + 1) / 2, nb), dim3(256), 0, st, A, Bm, ldx, K, C, ldc, tm, tn, sA, sB,
                       sC, pidxB);
  } else {
    const int npanel = (tn + S4_PW - 1) / S4_PW;
    if (mode == 2)
      hipLaunchKernelGGL(gemm4_kernel<2>, dim3(npanel * S4_PW * tm, nb), dim3(256), 0, st, A, Bm, ldx, K, C, ldc, tm, tn,
                         sA, sB, sC, pidxB);
    else
      hipLaunchKernelGGL(gemm4_kernel<0>, dim3(npanel * S4_PW * tm, nb), dim3(256), 0, st, A, Bm, ldx, K, C, ldc, tm, tn,
                         sA, sB, sC, pidxB);
  }
}

// ------------------------------------------------------------------------------------------
// Row quadratic forms on the ring:  q_i = a_i^T S a_i  for the rows a_i of A (M x n) and a SYMMETRIC S (n x n) --
// the predictive variance  k_*^T K^-1 k_*  of BayesGPR.predict (bask/bayesgpr.py:622-635 -> skopt's
// einsum("ki,kj,ij->k", K_trans, K_trans, K_inv)) and the same terms of PVRS (bask/acquisition.py:335-338).
// One workgroup per 64 x 64 tile (ti, tj) of  P = A S  restricted by symmetry to the block columns tk <= tj:
//     q_i = sum_tj [ 2 sum_{tk < tj} a_i[tk]^T S[tk,tj] a_i[tj]  +  a_i[tj]^T S[tj,tj] a_i[tj] ]
// i.e. HALF the flops of the full product; the tile's accumulators are doubled once before the diagonal block's
// chunks.  Each tile writes its 64 partial row sums to part[(b tn + tj) M + row]; rowdot_reduce_kernel adds the tn
// partials in tile order (no floating-point atomics: bitwise reproducible).  blockIdx.y = item of a batch
// (A + b sA, S = Sbase + pidx[b] sS).
// ------------------------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(256, (T == 128) ? 2 : 4)
    rowquad4_kernel(const double* __restrict__ A, int lda, size_t sA, const double* __restrict__ Sbase, int lds_,
                    size_t sS, const int* __restrict__ pidx, int tn, int M, int nitems, double* __restrict__ part) {
  constexpr unsigned OPB = T * S4_ROWB, STAGEB = 2 * OPB;
  constexpr int NRF = T / 32;
  __shared__ __attribute__((aligned(1024))) char smem[2 * STAGEB];
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  // XCD-aware order (placement only, results unchanged).  The dispatcher puts block id on XCD id % 8: with >= 8 items
  // item b is pinned to XCD b % 8 and that XCD walks all of its row tiles; with fewer items XCD x takes the row tiles
  // ti = x (mod 8) of every item (equal work per XCD).  Within that set the tiles go by PANELS of 256 columns of P
  // (tj), heavy (long k range) panels first; inside a panel row tile by row tile: the S panel (<= 2 MB) stays in the
  // XCD's L2 for all row tiles and the 256/T workgroups sharing a row tile of A run together (A streams from HBM once
  // per panel instead of once per column tile and XCD).
  constexpr int PWQ = 256 / T;
  const int tm = M / T;
  int b, ti, tj;
  {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tmx = (nitems >= 8) ? tm : (tm + 7) / 8;  // row tiles in this XCD's share of an item
    const int tiles = tmx * tn;
    const int mrow = q / tiles, t = q - mrow * tiles;
    b = (nitems >= 8) ? 8 * mrow + x : mrow;
    const int np = (tn + PWQ - 1) / PWQ, wl = tn - PWQ * (np - 1);  // the last panel may be narrower
    if (t < tmx * wl) {
      ti = t / wl;
      tj = tn - 1 - (t - ti * wl);
    } else {
      const int u = t - tmx * wl, pp = u / (PWQ * tmx), rem = u - pp * (PWQ * tmx);
      ti = rem / PWQ;
      tj = PWQ * (np - 2 - pp) + (PWQ - 1) - (rem - ti * PWQ);
    }
    if (nitems < 8) ti = 8 * ti + x;
    if (b >= nitems || ti >= tm) return;
  }
  const double* Ab = A + (size_t)b * sA + (size_t)(ti * T) * lda;
  const double* Sb = Sbase + (size_t)(pidx ? pidx[b] : b) * sS + (size_t)(tj * T) * lds_;
  unsigned voffA[T / 32], voffS[T / 32];
  s4_src<T>(voffA, lda, w, lane);
  s4_src<T>(voffS, lds_, w, lane);
  const int r0 = wr * (T / 2), c0 = wc * (T / 2);
  unsigned pa[4], pb[4];
  s4_frag_addr(pa, lds0, r0, lane);
  s4_frag_addr(pb, lds0 + OPB, c0, lane);
  d4 acc[NRF][NRF];
#pragma unroll
  for (int i = 0; i < NRF; i++)
#pragma unroll
    for (int j = 0; j < NRF; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const int nch = (tj + 1) * (T / S4_KC), cdiag = tj * (T / S4_KC);
  s4_issue<T>(Ab, voffA, 0, lds0, w);
  s4_issue<T>(Sb, voffS, 0, lds0 + OPB, w);
  for (int c = 0; c < nch; c += 2) {
    if (c == cdiag) {  // everything so far came from block columns tk < tj: it counts twice (S symmetric)
#pragma unroll
      for (int i = 0; i < NRF; i++)
#pragma unroll
        for (int j = 0; j < NRF; j++) acc[i][j] = acc[i][j] * 2.0;
    }
#pragma unroll
    for (int s = 0; s < 2; s++) {
      S4_WAIT_VM0();
      __builtin_amdgcn_s_barrier();
      if (c + s + 1 < nch) {
        const unsigned nb = lds0 + (unsigned)((s ^ 1) * STAGEB);
        s4_issue<T>(Ab, voffA, (c + s + 1) * S4_KC, nb, w);
        s4_issue<T>(Sb, voffS, (c + s + 1) * S4_KC, nb + OPB, w);
      }
      s4_mma<NRF, NRF, -64, 0, 0>(pa, pb, s * STAGEB, acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // row sums of P o A over this tile's T columns: 16 lanes share a row (fixed shuffle order), the two column
  // halves (wc = 0, 1) meet in LDS and are added in that order
  __syncthreads();  // every wave is past its last fragment read: the ring is free
  double* red = reinterpret_cast<double*>(smem);
  const double* E = Ab + tj * T;
#pragma unroll
  for (int i = 0; i < NRF; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = GK_ROWB(r0, i, lane, r);
      double ps = 0.0;
#pragma unroll
      for (int j = 0; j < NRF; j++) ps += acc[i][j][r] * E[(size_t)row * lda + GK_COLB(c0, j, lane)];
      ps += __shfl_xor(ps, 1);
      ps += __shfl_xor(ps, 2);
      ps += __shfl_xor(ps, 4);
      ps += __shfl_xor(ps, 8);
      if ((lane & 15) == 0) red[wc * T + row] = ps;
    }
  }
  __syncthreads();
  if (tid < T) part[((size_t)b * tn + tj) * M + ti * T + tid] = red[tid] + red[T + tid];
}

// q (nb x M, packed) needs `part` = nb * (n / T) * M doubles of scratch; M and n are multiples of 128.
// T = 64 (the 128-wide tile measured equal within noise at 128 posteriors x 10 000 points x n = 1024, 46 vs 43 ms per batched
// predict: the kernel is bound by streaming K_*, not by the tile shape).
int bgp_rowquad_tile() { return 64; }
void bgp_launch_rowquad(hipStream_t st, const double* A, int lda, size_t sA, const double* S, int lds_, size_t sS,
                        const int* pidx, int M, int n, int nb, double* part) {
  const int T = bgp_rowquad_tile(), tn = n / T, tm = M / T, tiles = tm * tn;
  const int grid = (nb >= 8) ? 8 * ((nb + 7) / 8) * tiles : 8 * ((tm + 7) / 8) * tn * nb;
  hipLaunchKernelGGL(rowquad4_kernel<64>, dim3(grid), dim3(256), 0, st, A, lda, sA, S, lds_, sS, pidx, tn, M, nb, part);
}

#ifdef S4_BENCH  // ablation / trace instantiations for tools/syrk4_bench.hip (not in the product library)
// trsm4 ablations (tools/trsm4_bench.hip): var bit 0 = no LDS-DMA, 1 = no MFMA, 3 = no stores, 4 = W staged once only
extern "C" int bgp_debug_launch_trsm4(int var, hipStream_t st, int B, double* dK, double* dW, double* dyw, int* dstatus,
                                      int ld, size_t mstride, int ystride, int nblk, int k) {
  const int B8 = 8 * ((B + 7) / 8);
  const dim3 grid(B8 * 2 * (nblk - k - 1));
#define T4_CASE(V)                                                                                                  \
  if (var == V) {                                                                                                   \
    hipLaunchKernelGGL(trsm4_kernel<V>, grid, dim3(256), 0, st, dK, dW, dyw, dstatus, ld, mstride, ystride, nblk, k, B, 0); \
    return (int)grid.x;                                                                                             \
  }
  T4_CASE(0) T4_CASE(1) T4_CASE(2) T4_CASE(3) T4_CASE(8) T4_CASE(16) T4_CASE(10) T4_CASE(11)
  return 0;
}

extern "C" int bgp_debug_launch_syrk4(int T, int var, hipStream_t st, int B8, double* dK, const int* dstatus, int ld,
                                      size_t mstride, int nblk, int kp, int K, int jstart, int colmode, int B,
                                      unsigned long long* trace) {
  const int nt = nblk - jstart;
  const int total = B8 * (T == 128 ? s4_ntile<128>(nt, colmode) : s4_ntile<64>(nt, colmode));
#define S4_CASE(TT, V)                                                                                               \
  if (T == TT && var == V) {                                                                                         \
    hipLaunchKernelGGL((syrk4_kernel<TT, V, S4NoGen>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K,    \
                       jstart, colmode, B, total, trace, S4_PW, S4NoGen());                                 \
    return total;                                                                                                    \
  }
  S4_CASE(128, 0) S4_CASE(128, 4) S4_CASE(64, 0) S4_CASE(64, 1) S4_CASE(64, 2) S4_CASE(64, 3) S4_CASE(64, 4)
  fprintf(stderr, "bgp_debug_launch_syrk4: no instantiation T=%d var=%d\n", T, var);
  return 0;
}
#endif

