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
//   * tile edge T = 64 (each wave a 32x32 block, 66 VGPRs, 32 KB of LDS: five workgroups = 20 waves per CU)
//     is what the library launches: measured on MI355X it ties the 128x128 tile (two workgroups per CU) on the
//     largest launch of BASELINE config C (62.6 vs 63.0 TF) although it moves twice the bytes per flop from
//     L2 and LDS, and wins everywhere else (B=16: 62.5 vs 55.4 TF; 320 tiles: 43 vs 28 TF) because small
//     launches fill the chip 4x better and the deeper occupancy hides the per-tile C round trip.  T = 128 is
//     kept as a template instantiation for tools/syrk4_bench.hip.  A persistent variant (workgroups walking
//     the tile list, next tile's first chunk in flight during the epilogue) was measured 8-12 % SLOWER: the
//     resident workgroups start in lockstep and stay there, so their C-tile traffic comes in bursts.
// Results do not depend on T or the launch geometry: every C element is owned by one lane and accumulated in
// the same k order (k ascending in steps of 4) -- bitwise reproducible and batch-split invariant.
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"
#include "bgp_ring.h"
#include "bgp_pf.h"

#include <algorithm>
#include <cstdlib>

#define S4_KC 16
#define S4_ROWB (S4_KC * 8)  // bytes per LDS row

// Wave w stages rows [w T/4, (w+1) T/4) of one T x 16 operand chunk: T/32 instructions x 8 rows.  voff[i] is this
// lane's (swizzled) byte offset for instruction i, the same for every operand panel and chunk: the panel origin
// and the chunk's k0 go into the wave-uniform base (scalar adds only).
template <int T>
static __device__ __forceinline__ void s4_issue(const double* X, const unsigned (&voff)[T / 32], int k0,
                                                unsigned lds_op_base, int w) {
#pragma unroll
  for (int i = 0; i < T / 32; i++)
    s4_glds(X + k0, voff[i], lds_op_base + (unsigned)(((T / 4) * w + 8 * i) * S4_ROWB));
}

// The same, rows >= minrow only (wave-uniform test): the panel solve's triangular operand -- chunk c never reads the
// rows below 16 c of W_kk, so they are not staged (their LDS slots keep stale data that no fragment read touches).
template <int T>
static __device__ __forceinline__ void s4_issue_from(const double* X, const unsigned (&voff)[T / 32], int k0,
                                                     unsigned lds_op_base, int w, int minrow) {
#pragma unroll
  for (int i = 0; i < T / 32; i++)
    if ((T / 4) * w + 8 * i + 8 > minrow) s4_glds(X + k0, voff[i], lds_op_base + (unsigned)(((T / 4) * w + 8 * i) * S4_ROWB));
}

template <int T>
static __device__ __forceinline__ void s4_src(unsigned (&voff)[T / 32], int ld, int w, int lane) {
#pragma unroll
  for (int i = 0; i < T / 32; i++) {
    const int row = (T / 4) * w + 8 * i + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = (unsigned)(row * ld + g * 2) * 8u;
  }
}

// Per-lane LDS byte addresses of the fragment rows: (row row0 + (lane & 15), k-step kk), swizzle applied.
static __device__ __forceinline__ void s4_frag_addr(unsigned (&p)[4], unsigned base, int row0, int lane) {
  const int lr = lane & 15, lk = lane >> 4, f = (lr >> 1) & 7;  // (row0 is a multiple of 16: f depends on lr only)
#pragma unroll
  for (int kk = 0; kk < 4; kk++)
    p[kk] = base + (unsigned)((row0 + lr) * S4_ROWB + ((((kk * 2) + (lk >> 1)) ^ f) << 4) + ((lk & 1) << 3));
}

// acc[i][j] -= A_i B_j^T over one 16-wide chunk.  pa / pb: see s4_frag_addr; `soff` = compile-time stage offset.
// (hipcc pairs the fragment reads into ds_read2st64_b64; hand-placed single ds_read_b64 with counted lgkmcnt
// waits -- conflict-free and twice the LDS rate on paper -- measured no faster: the LDS is not the limiter.)
// VAR (bench builds only, tools/syrk4_bench.hip): bit 0 = no LDS-DMA issue, bit 1 = no MFMAs, bit 2 = timestamps.
// jmin (wave-uniform): column blocks j < jmin are skipped (panel solve: B = W_kk is lower triangular and its upper
// blocks are never written, so this chunk's k range does not reach them); pass 0 for "all".
template <int NR, int NC, int CREL, int VAR, int NEGA = 1>
static __device__ __forceinline__ void s4_mma(const unsigned (&pa)[4], const unsigned (&pb)[4], int soff,
                                              d4 (&acc)[NR][NC], int jmin = 0) {
  typedef __attribute__((address_space(3))) const double* lds_cdp;
  double a[2][NR], b[2][NC];
#pragma unroll
  for (int i = 0; i < NR; i++) a[0][i] = *(lds_cdp)(uintptr_t)(pa[0] + soff + i * 16 * S4_ROWB);
#pragma unroll
  for (int j = 0; j < NC; j++) b[0][j] = *(lds_cdp)(uintptr_t)(pb[0] + soff + j * 16 * S4_ROWB);
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int cur = kk & 1, nxt = cur ^ 1;
    if (kk < 3) {
#pragma unroll
      for (int i = 0; i < NR; i++) a[nxt][i] = *(lds_cdp)(uintptr_t)(pa[kk + 1] + soff + i * 16 * S4_ROWB);
#pragma unroll
      for (int j = 0; j < NC; j++) b[nxt][j] = *(lds_cdp)(uintptr_t)(pb[kk + 1] + soff + j * 16 * S4_ROWB);
    }
#pragma unroll
    for (int j = 0; j < NC; j++) {
      if (j < jmin) continue;  // wave-uniform
#pragma unroll
      for (int i = 0; i < NR; i++) {
        if (j + CREL > i) continue;  // compile-time (lower part of a diagonal block)
        if (VAR & 2)
          asm volatile("" ::"v"(a[cur][i]), "v"(b[cur][j]));
        else
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][i], b[cur][j], acc[i][j], 0, 0, NEGA);
      }
    }
  }
}

struct S4Tile {
  const double* XA;  // rows of block I, panel columns
  const double* XB;  // rows of block J
  double* C;
  int b, gi0, gj0;  // batch slot, first matrix row / column of the tile (Gram generation)
  int q;      // position in the launch's tile list (>= total: none)
  int diag;   // I == J: X_I is both operands; only the lower triangle is updated
  int label;  // I * 1000 + J (bench timeline)
};

#define S4_STAMP(i)                                                      \
  do {                                                                   \
    if ((VAR & 4) && trace) {                                            \
      __builtin_amdgcn_sched_barrier(0);                                 \
      const unsigned long long t__ = __builtin_readcyclecounter();       \
      if (threadIdx.x == 0) trace[(size_t)cur.q * 8 + (i)] = t__;        \
      __builtin_amdgcn_sched_barrier(0);                                 \
    }                                                                    \
  } while (0)

// Gram entries of a wave's NR x NC block, produced in the accumulator layout instead of being loaded: the FIRST trailing
// update that touches a tile of K builds it (scaled inputs Xs are k-major and L2-resident: 256 KB per matrix at
// config C), so the Gram matrix is never written and read back except for block column 0.  Same arithmetic as the
// Gram kernels (bgp_kbuild.hip: differences squared and summed in dimension order with one fma each, then
// kb_epilogue's expressions without implicit contraction): bit-identical K.  OPT-IN (BGP_FUSED_GRAM=1): measured on
// MI355X the generation is NOT hidden under the other workgroups' MFMAs -- a VALU instruction costs the fp64 MFMA its
// issue slots (tools/mfma_interleave_probe.hip) -- so only the saved HBM round trip of K shows: 15.6 vs 15.9 ms per step
// at config C, while the trailing update's own launches get 11 % longer; small batches lose 2-5 %.
template <int NR, int NC, int CREL, int STAT, int FORM>
static __device__ __forceinline__ void s4_gen_c(const S4Gen& g, const S4Tile& cur, d4 (&acc)[NR][NC], int r0, int c0,
                                                int lane) {
  const double* Xs_b = g.Xs + (size_t)cur.b * g.dpad * g.npad;
  const double* h = g.H + (size_t)cur.b * (g.d + 2);
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  const double* pa = Xs_b + cur.gi0 + r0 + (lane >> 4);
  const double* pb = Xs_b + cur.gj0 + c0 + (lane & 15);
#pragma unroll 2
  for (int k = 0; k < g.d; k++) {
    double a[NR][4], bb[NC];
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) a[i][r] = pa[16 * i + 4 * r];
#pragma unroll
    for (int j = 0; j < NC; j++) bb[j] = pb[16 * j];
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++) {
        if (j + CREL > i) continue;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const double df = a[i][r] - bb[j];
          acc[i][j][r] = fma(df, df, acc[i][j][r]);
        }
      }
    pa += g.npad;
    pb += g.npad;
  }
  {
#pragma clang fp contract(off)
    const double cst = exp(h[0]), s2 = exp(h[g.d + 1]);
    const bool interior = !cur.diag && cur.gi0 + 64 <= g.n && cur.gj0 + 64 <= g.n;  // (T <= 64 rows / columns per wave block)
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++) {
        if (j + CREL > i) continue;
        const int gj = cur.gj0 + GK_COLB(c0, j, lane);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gi = cur.gi0 + GK_ROWB(r0, i, lane, r);
          double v;
          if (!interior && (gi >= g.n || gj >= g.n)) {
            v = (gi == gj) ? 1.0 : 0.0;  // identity padding
          } else if (!interior && gi == gj) {
            const double base = (FORM == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
            v = base + s2;
            if (g.alpha) v += g.alpha[gi];
          } else {
            const double sv = kb_stationary<STAT>(acc[i][j][r]);
            v = (FORM == BGP_FORM_PRODUCT) ? cst * sv : cst + sv;
          }
          acc[i][j][r] = v;
        }
      }
  }
}

// One tile for a wave's NR x NC block at (r0, c0) of the T x T workgroup tile.
// NEGA = 1: C -= A B^T (the factorisation's update), 0: C += A B^T; ZEROC: C is not loaded (starts from zero); ldc = leading
// dimension of C when it is not the operands' (gemm4_kernel).
template <int T, int NR, int NC, int CREL, int VAR, int GEN = 0, int STAT = 0, int FORM = 0, int NEGA = 1, int ZEROC = 0>
static __device__ __forceinline__ void s4_tile(unsigned long long* trace, unsigned lds0, const S4Tile& cur,
                                               const unsigned (&voff)[T / 32], int ld, int K, int r0, int c0, int w,
                                               int lane, const S4Gen& gen, int ldc = 0) {
  if (ldc == 0) ldc = ld;
  constexpr unsigned OPB = T * S4_ROWB, STAGEB = 2 * OPB;
  unsigned pa[4], pb[4];
  s4_frag_addr(pa, lds0, r0, lane);
  s4_frag_addr(pb, cur.diag ? lds0 : lds0 + OPB, c0, lane);
  const int nch = K / S4_KC;
  d4 acc[NR][NC];
  if (!(VAR & 1)) {  // chunk 0 -> stage 0
    s4_issue<T>(cur.XA, voff, 0, lds0, w);
    if (!cur.diag) s4_issue<T>(cur.XB, voff, 0, lds0 + OPB, w);
  }
  // The empty asm makes hipcc wait for its C loads HERE (its in-order vmcnt wait also covers chunk 0, needed
  // next anyway) instead of at their first use inside the loop, where such a wait would drain the LDS-DMA queue.
  if (GEN) {
    s4_gen_c<NR, NC, CREL, STAT, FORM>(gen, cur, acc, r0, c0, lane);
  } else if (ZEROC) {
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  } else {
    gk_load_c<NR, NC, CREL>(cur.C, (size_t)ldc, acc, r0, c0, lane);
  }
#pragma unroll
  for (int i = 0; i < NR; i++)
#pragma unroll
    for (int j = 0; j < NC; j++)
      if (j + CREL <= i) asm volatile("" : "+v"(acc[i][j]));
  S4_STAMP(1);
  for (int c = 0; c < nch; c += 2) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      S4_WAIT_VM0();                 // this wave's share of chunk c+s has landed
      __builtin_amdgcn_s_barrier();  // ... everybody's has; everybody finished reading chunk c+s-1
      if (!(VAR & 1) && c + s + 1 < nch) {
        const unsigned nb = lds0 + (unsigned)((s ^ 1) * STAGEB);
        s4_issue<T>(cur.XA, voff, (c + s + 1) * S4_KC, nb, w);
        if (!cur.diag) s4_issue<T>(cur.XB, voff, (c + s + 1) * S4_KC, nb + OPB, w);
      }
#ifdef BGP_FAULT_INJECT  // tests/fault/ only (never in libbgp.so): a trailing update that drops its last 16-wide k-chunk on
      // the tiles from matrix row 1536 on -- what tests/test_gpu_dense.py must turn red on
      if ((BGP_FAULT_INJECT & 1) && NEGA == 1 && !ZEROC && cur.gi0 >= 1536 && c + s == nch - 1) continue;
#endif
      s4_mma<NR, NC, CREL, VAR, NEGA>(pa, pb, s * STAGEB, acc);
      __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs of this chunk above the next wait + barrier
    }
  }
  S4_STAMP(2);
  gk_store_c<NR, NC, CREL>(cur.C, (size_t)ldc, acc, r0, c0, lane);
}

// Tile list of one launch, in blocks of T rows:  nt128 = trailing 128-row blocks, colmode 0: every tile with
// I >= J, colmode 1: only the tiles inside the first 128 columns (the look-ahead block column).
template <int T>
static __host__ __device__ __forceinline__ int s4_ntile(int nt128, int colmode) {
  const int nt = nt128 * (128 / T);
  if (!colmode) return nt * (nt + 1) / 2;
  return (T == 128) ? nt : 2 * nt - 1;
}

// Tile order of a matrix's lower triangle (nt x nt tiles): column panels of S4_PW tile columns, each swept top to
// bottom.  The tiles an XCD runs at any time then share ONE panel's B rows (S4_PW x 64 x K x 8 B = 2 MB at K = 512:
// resident in the XCD's 4 MB L2) and stream the A rows once per panel -- row-major order re-fetched every B row
// block for every tile row once the K = 512 panel (7 MB per matrix) had outgrown the L2 (rocprofv3: 1.25 GB per
// launch against 0.45 GB of C traffic).  Placement only: results do not depend on the order.
#define S4_PW 8
static __device__ __forceinline__ void s4_panel_decode(int t, int nt, int& ti, int& tj, int pw = S4_PW) {
  int p0 = 0;
  for (;;) {  // (at most nt / S4_PW iterations)
    const int rows = nt - p0;                       // tile rows of this panel
    const int w = rows < pw ? rows : pw;            // its width
    const int cnt = w * (w + 1) / 2 + (rows - w) * w;
    if (t < cnt) {
      const int head = w * (w + 1) / 2;             // triangular head (the panel's diagonal tiles), then full rows
      int r, c;
      if (t < head) {
        bgp_tri_decode(t, r, c);
      } else {
        r = w + (t - head) / w;
        c = (t - head) - (r - w) * w;
      }
      ti = p0 + r;
      tj = p0 + c;
      return;
    }
    t -= cnt;
    p0 += pw;
  }
}

template <int T>
static __device__ __forceinline__ S4Tile s4_decode(int q, int total, int ntile, double* Kbuf, const int* status, int ld,
                                                   size_t mstride, int kp, int jstart, int colmode, int nt128, int B,
                                                   int pw) {
  S4Tile d;
  d.XA = d.XB = nullptr;
  d.C = nullptr;
  d.diag = 0;
  d.b = d.gi0 = d.gj0 = 0;
  d.label = 0;
  d.q = total;
  do {
    int b, t;
    bgp_map_block(q, ntile, B, b, t);
    if (b >= B || status[b] != 0) break;  // padding slot / failed factorisation: nothing to update
    int ti, tj;
    if (colmode == 2) {
      // posterior build on the augmented matrix: the trailing set is the nblk ACTIVE block rows of bgp_rowblk (what is
      // left of K, then the first kp+1 block rows of the identity part); single panel kp, K = 128
      s4_panel_decode(t, nt128 * (128 / T), ti, tj, pw);
      const int nlow = nt128 - kp - 1;
      const size_t rI = (size_t)bgp_rowblk((ti * T) >> 7, kp, nlow, nt128) * 128 + ((ti * T) & 127);
      const size_t rJ = (size_t)bgp_rowblk((tj * T) >> 7, kp, nlow, nt128) * 128 + ((tj * T) & 127);
      double* M2 = Kbuf + (size_t)b * mstride;
      d.XA = M2 + rI * ld + kp * 128;
      d.XB = M2 + rJ * ld + kp * 128;
      d.C = M2 + rI * ld + rJ;
      d.diag = (ti == tj);
      d.b = b;
      d.gi0 = (int)rI;
      d.gj0 = (int)rJ;
      d.label = ti * 1000 + tj;
      d.q = q;
      break;
    } else if (!colmode) {
      s4_panel_decode(t, nt128 * (128 / T), ti, tj, pw);
    } else if (T == 128 || t < nt128 * 2) {
      ti = t;
      tj = 0;
    } else {
      ti = t - nt128 * 2 + 1;
      tj = 1;
    }
    double* M = Kbuf + (size_t)b * mstride;
    const size_t rowI = (size_t)jstart * 128 + (size_t)ti * T, rowJ = (size_t)jstart * 128 + (size_t)tj * T;
    d.XA = M + rowI * ld + kp * 128;
    d.XB = M + rowJ * ld + kp * 128;
    d.C = M + rowI * ld + rowJ;
    d.diag = (ti == tj);
    d.b = b;
    d.gi0 = (int)rowI;
    d.gj0 = (int)rowJ;
    d.label = ti * 1000 + tj;
    d.q = q;
  } while (0);
  return d;
}

template <int T, int VAR, int GEN = 0, int STAT = 0, int FORM = 0>
__global__ void __launch_bounds__(256, (T == 128) ? 2 : 4)
    syrk4_kernel(double* __restrict__ Kbuf, const int* __restrict__ status, int ld, size_t mstride, int nblk, int kp,
                 int K, int jstart, int colmode, int B, int total, unsigned long long* __restrict__ trace, int pw,
                 S4Gen gen) {
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
    s4_tile<T, NRF, NRF, -64, VAR, GEN, STAT, FORM>(trace, lds0, cur, voff, ld, K, wr * (T / 2), wc * (T / 2), w, lane, gen);
  } else if (w < 2) {
    // Diagonal tile: only its lower triangle is ever read again.  Waves 0 / 1: the two (T/2)^2 triangles on the
    // diagonal; waves 2 / 3: the square below the diagonal cut into two row halves (3/3/2/2 MFMA tiles at T = 64).
    s4_tile<T, NRF, NRF, 0, VAR, GEN, STAT, FORM>(trace, lds0, cur, voff, ld, K, w * (T / 2), w * (T / 2), w, lane, gen);
  } else {
    s4_tile<T, NRF / 2, NRF, -64, VAR, GEN, STAT, FORM>(trace, lds0, cur, voff, ld, K, T / 2 + (w - 2) * (T / 4), 0, w, lane, gen);
  }
  S4_STAMP(3);
  if ((VAR & 4) && trace && threadIdx.x == 0) trace[(size_t)cur.q * 8 + 7] = wall_clock64();
}

void bgp_launch_syrk4(hipStream_t st, int B8, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                      int K, int jstart, int colmode, int B, const S4Gen* gen) {
  const int total = B8 * (colmode == 2 ? s4_ntile<64>(nblk, 0) : s4_ntile<64>(nblk - jstart, colmode));
  static int pw = 0;
  if (!pw) {
    const char* e = getenv("BGP_PANEL_WIDTH");  // tile columns per L2-resident column panel (s4_panel_decode)
    pw = (e && atoi(e) >= 1 && atoi(e) <= 64) ? atoi(e) : S4_PW;
  }
  if (gen) {  // first touch of these tiles: they generate their Gram entries instead of loading them
    KB_DISPATCH(gen->stat, gen->form,
                hipLaunchKernelGGL((syrk4_kernel<64, 0, 1, S, F>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride,
                                   nblk, kp, K, jstart, colmode, B, total, nullptr, pw, *gen));
    return;
  }
  hipLaunchKernelGGL((syrk4_kernel<64, 0>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K, jstart,
                     colmode, B, total, nullptr, pw, S4Gen());
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
// Tile worker of the launch-free factorisation (see ps_chain_kernel, bgp_chol.hip, for the scheme).  Left-looking by
// blocks: a task owns one 128 x 128 block (I, Jc) of matrix b, one 512-thread workgroup (8 waves), and
//   1. loads it once and applies the finished panels to its left,  C -= X_I,p X_Jc,p^T, on a FOUR-stage LDS-DMA ring
//      (three 16-wide chunks in flight: a two-stage ring spent an L2 round trip of 2-5 us on every 0.4 us chunk --
//      tools/persist_trace.py), as far as the panels are final: it waits on xready only when it has caught up with the
//      factorisation;
//   2. S(I, J), I >= J+2: stores the block, waits for W_JJ (wready) and runs the panel solve X = C W_JJ^T in place with the
//      fused right-hand-side update y_I -= X z_J (the arithmetic of trsm4_kernel), then raises xready[I][J];
//      P(I) = block (I, I-1) and Dg(I) = block (I, I), I >= 2: the PRE-updates with the panels 0 .. I-2: they store the block
//      and raise subrdy[I] / diagrdy[I] -- the chain workgroup applies the last panel, solves and factorises them itself.
// Tasks are drawn from ticket counters.  Order, per block column J = 0 .. nblk-3 and across the matrices of the batch: the
// panel solve S(J+2, J) -- the block both pre-updates of the column wait for --, then P(J+2), Dg(J+2), then S(J+3 .., J): a
// topological order of the dependency graph (every task only waits for tasks with smaller tickets and for the chain), so
// the earliest unfinished task always belongs to a running workgroup: no deadlock whatever the number of resident
// workgroups.  With PsArgs::ncrit > 0 the three tasks at the head of every column have ticket lists and workgroups of
// their own (the first ncrit of the launch): a task the chain is going to wait for never queues behind a long update
// (each list is in topological order and together they hold every task: still no deadlock; a workgroup whose pool is
// exhausted helps the other one).
// Chain pairs (PsArgs::psplit == 4; bgp_pf.h: pf_pair_helper): the per-column cycle of the chain runs THROUGH the critical tasks, and
// one CU applies a panel to a 128 x 128 block in 14 us (0.307 TF of fp64 MFMA per CU).  They go out in 64 x 64 QUADRANTS on
// workgroups of their own (ps_ll_update_quad) -- P(J+2) in four, Dg(J+2) in three, and four quadrants Q of block (J+2, J) AHEAD
// of the critical solve S(J+2, J), which then only waits for them (s2rdy) and solves -- and every quadrant consumes its LAST
// panel chunk by chunk behind the blocks that feed it: the chain helper's X_{J+1,J} and the streamed solves (pf_stream_S)
// publish a count of 16-column blocks that are complete in memory (xcol; write-through stores), chunk c of the panel is column
// block c.  Order inside a column's critical group: Q, S, P, Dg (still topological: Q waits for solves of column J-1 only).
// Per C element the operations and their order are those of syrk4_kernel / trsm4_kernel (accumulator = C, MFMA k-steps
// ascending, A-negate): bit-identical factors.
// ------------------------------------------------------------------------------------------
// critical tasks per block column and matrix: S(J+2, J), the np parts of P(J+2) (np = PsArgs::psplit: 2 column slices or 4
// quadrants), the PS_ND(np) parts of Dg(J+2) and the PS_NQ(np) quadrants Q ahead of S(J+2, J)
#define PS_ND(np) ((np) == 4 ? 3 : 1)
#define PS_NQ(np) ((np) == 4 ? 4 : 0)
static __host__ __device__ __forceinline__ int ps_crit_per_matrix(int nblk, int np) {
  return nblk > 2 ? (np + 1 + PS_ND(np) + PS_NQ(np)) * (nblk - 2) : 0;
}
static __host__ __device__ __forceinline__ int ps_bulk_per_matrix(int nblk) { return nblk > 3 ? (nblk - 3) * (nblk - 2) / 2 : 0; }
static __host__ __device__ __forceinline__ int ps_tasks_per_matrix(int nblk, int np) { return ps_crit_per_matrix(nblk, np) + ps_bulk_per_matrix(nblk); }

// vmcnt(N) with a compile-time N
template <int N>
static __device__ __forceinline__ void s4_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// eight-wave staging of a 128-row operand chunk: wave w stages rows [16 w, 16 w + 16) = two instructions of 8 rows
static __device__ __forceinline__ void s8_src(unsigned (&voff)[2], int ld, int w, int lane) {
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = 16 * w + 8 * i + (lane >> 3);
    const int g = (lane & 7) ^ ((row >> 1) & 7);
    voff[i] = (unsigned)(row * ld + g * 2) * 8u;
  }
}
static __device__ __forceinline__ void s8_issue(const double* X, const unsigned (&voff)[2], int k0, unsigned lds_op_base, int w) {
#pragma unroll
  for (int i = 0; i < 2; i++) s4_glds(X + k0, voff[i], lds_op_base + (unsigned)((16 * w + 8 * i) * S4_ROWB));
}

// vmcnt(2 n) / vmcnt(4 n) for a run-time n in 0 .. 2 (the counted waits of the four-stage ring)
static __device__ __forceinline__ void s4_wait_vm_n2(int n) {
  if (n >= 2) s4_wait_vm<4>(); else if (n == 1) s4_wait_vm<2>(); else s4_wait_vm<0>();
}
static __device__ __forceinline__ void s4_wait_vm_n4(int n) {
  if (n >= 2) s4_wait_vm<8>(); else if (n == 1) s4_wait_vm<4>(); else s4_wait_vm<0>();
}

// acc (-)= A B^T over `nch` 16-wide chunks on a four-stage LDS-DMA ring, eight waves: 128 rows of A at XA, 128 rows of B at
// XB.  Waits are counted: the DMA returns in order, so "at most r younger chunks outstanding" = vmcnt(4 r) (four
// instructions per wave and chunk).  `tri`: B is lower triangular (panel solve: chunk c only reaches the column blocks >= c).
template <int NST, int NR, int NC, int NEGA>
static __device__ __forceinline__ void s8_ring_run(const double* XA, const unsigned (&voffA)[2], const double* XB,
                                                   const unsigned (&voffB)[2], int nch, unsigned lds0,
                                                   const unsigned (&pa)[4], const unsigned (&pb)[4], d4 (&acc)[NR][NC],
                                                   int w, int tri) {
  static_assert(NST == 4, "the counted waits are written for four stages");
  constexpr unsigned AOPB = 128 * S4_ROWB, STAGEB = 256 * S4_ROWB;
#pragma unroll
  for (int s = 0; s < NST - 1; s++) {
    if (s < nch) {
      s8_issue(XA, voffA, s * S4_KC, lds0 + s * STAGEB, w);
      s8_issue(XB, voffB, s * S4_KC, lds0 + s * STAGEB + AOPB, w);
    }
  }
  for (int c = 0; c < nch; c += NST) {
#pragma unroll
    for (int s = 0; s < NST; s++) {
      if (c + s >= nch) break;            // (wave- and workgroup-uniform)
      const int rem = nch - (c + s) - 1;  // chunks behind this one
      s4_wait_vm_n4(rem < NST - 2 ? rem : NST - 2);
      __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
      if (c + s + NST - 1 < nch) {
        const unsigned nb = lds0 + (unsigned)(((s + NST - 1) % NST) * STAGEB);
        s8_issue(XA, voffA, (c + s + NST - 1) * S4_KC, nb, w);
        s8_issue(XB, voffB, (c + s + NST - 1) * S4_KC, nb + AOPB, w);
      }
      s4_mma<NR, NC, -64, 0, NEGA>(pa, pb, s * STAGEB, acc, tri ? c + s : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

// The same ring for a DIAGONAL block's update, dt[u] -= X_ti X_tj^T on this wave's lower 16 x 16 tiles (t = w, w + 8, ... < 36
// in row-major order of the triangle: waves 0-3 five, 4-7 four): only what the factorisation reads is computed -- 36 tiles
// instead of the 48 that a 4 x 2 arrangement of 32 x 64 wave blocks covers with six busy waves of eight tiles each -- and all
// eight waves share it: 5 instead of 8 tiles on the busiest wave.  One image of the 128 rows per chunk (X_I is both operands).
// oa / ob: byte offsets of the tile's row blocks inside the image; per element the k order of every other update path.
template <int NST, int NT>
static __device__ __forceinline__ void s8_ring_run_diag(const double* XA, const unsigned (&voffA)[2], int nch, unsigned lds0,
                                                        const unsigned (&p0)[4], const unsigned (&oa)[5], const unsigned (&ob)[5],
                                                        d4 (&dt)[5], int w) {
  static_assert(NST == 4, "the counted waits are written for four stages");
  typedef __attribute__((address_space(3))) const double* lds_cdp;
  constexpr unsigned STAGEB = 256 * S4_ROWB;
#pragma unroll
  for (int s = 0; s < NST - 1; s++)
    if (s < nch) s8_issue(XA, voffA, s * S4_KC, lds0 + s * STAGEB, w);
  for (int c = 0; c < nch; c += NST) {
#pragma unroll
    for (int s = 0; s < NST; s++) {
      if (c + s >= nch) break;            // (wave- and workgroup-uniform)
      const int rem = nch - (c + s) - 1;  // chunks behind this one
      s4_wait_vm_n2(rem < NST - 2 ? rem : NST - 2);
      __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
      if (c + s + NST - 1 < nch) s8_issue(XA, voffA, (c + s + NST - 1) * S4_KC, lds0 + (unsigned)(((s + NST - 1) % NST) * STAGEB), w);
      double a[2][NT], b[2][NT];
#pragma unroll
      for (int u = 0; u < NT; u++) {
        a[0][u] = *(lds_cdp)(uintptr_t)(p0[0] + s * STAGEB + oa[u]);
        b[0][u] = *(lds_cdp)(uintptr_t)(p0[0] + s * STAGEB + ob[u]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int cur = kk & 1, nxt = cur ^ 1;
        if (kk < 3) {
#pragma unroll
          for (int u = 0; u < NT; u++) {
            a[nxt][u] = *(lds_cdp)(uintptr_t)(p0[kk + 1] + s * STAGEB + oa[u]);
            b[nxt][u] = *(lds_cdp)(uintptr_t)(p0[kk + 1] + s * STAGEB + ob[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) dt[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[cur][u], b[cur][u], dt[u], 0, 0, 1);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

#define PS_NST 4
// Left-looking update of a 128 x (16 NC x 2) slice of a block with the panels 0 .. npan-1 as far as they are final (it waits on
// xready only when it has caught up with the factorisation): C -= X_I,p X_J,p^T, waves as 4 x 2, each 32 rows x 16 NC columns.
// XB = the rows of block row Jc that belong to the slice's columns (the ring stages 128 rows from there: the rows behind a
// narrower slice are staged and not read).  Returns 0, or -1 when a wait was abandoned.
template <int NC>
static __device__ __forceinline__ int ps_ll_update(const PsArgs& a, const double* XA, const double* XB, double* C, int npan,
                                                   unsigned* xrI, unsigned* xrJ, unsigned* err, int* sh_q, unsigned lds0,
                                                   const unsigned (&voffX)[2], int ld, int w, int lane, int tid, int I,
                                                   unsigned long long* tr) {
  constexpr unsigned AOPB = 128 * S4_ROWB;
  const int wr = w >> 1, wc = w & 1;
  unsigned pa[4], pb[4];
  d4 acc[2][NC];
  s4_frag_addr(pa, lds0, wr * 32, lane);
  s4_frag_addr(pb, lds0 + AOPB, wc * 16 * NC, lane);
  gk_load_c<2, NC, -64>(C, (size_t)ld, acc, wr * 32, wc * 16 * NC, lane);
  int q = 0;
  while (q < npan) {
    if (tid == 0) {
      int qq = q;
      bool ok = true;
#define PS_READY(p) (ps_ld(xrI + (p)) >= 1u && ps_ld(xrJ + (p)) >= 1u)
      while (qq < npan && PS_READY(qq)) qq++;
      if (qq == q) {  // caught up with the factorisation: wait for the next panel
        ok = ps_wait_ge2(xrI + q, 1u, xrJ + q, 1u, err, a.spin_limit);
        qq = q + 1;
        while (ok && qq < npan && PS_READY(qq)) qq++;
      }
#undef PS_READY
      ps_acquire();
      *sh_q = ok ? qq : -1;
      if (tr && q == 0) tr[1] = wall_clock64();
      if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
    }
    __syncthreads();
    const int qq = *sh_q;
    if (qq < 0) return -1;  // abandoned
    int nch_run = (qq - q) * 8;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
    if ((BGP_FAULT_INJECT & 2) && I >= 12 && qq == npan) nch_run -= 1;
#endif
    s8_ring_run<PS_NST, 2, NC, 1>(XA + (size_t)q * 128, voffX, XB + (size_t)q * 128, voffX, nch_run, lds0, pa, pb, acc, w, 0);
    __syncthreads();  // (the ring and sh_q are free again)
    q = qq;
  }
  gk_store_c<2, NC, -64>(C, (size_t)ld, acc, wr * 32, wc * 16 * NC, lane);
  return 0;
}

// One QUADRANT (64 x 64) of a P block, the four-way split of the chain pairs' critical pre-update: 64 rows of X_I and 64 rows of
// X_Jc per chunk -- half the bytes of a 128 x 32 column slice, which stages 128 + 128 rows and reads 160 of them -- on an
// EIGHT-stage ring of 128-row images (the same 128 KB of LDS): seven chunks in flight instead of three.  The last panel's term is
// pure latency (its 8 chunks, written a microsecond ago by two other workgroups, arrive at the hand-off rate of
// MI355X_MICROARCH.md "handoff-payload"): 6.3 us -> see DESIGN.md section 10 with three 32 KB chunks in flight.  Waves as 4 x 2, each
// 16 rows x 32 columns; wave w stages rows 16 w .. 16 w + 15 of the image (waves 0-3: X_I, 4-7: X_Jc): two instructions per
// wave and chunk, "at most r younger chunks outstanding" = vmcnt(2 r).  Per element the k order of every other update path.
static __device__ __forceinline__ void q8_wait_vm(int r) {
  switch (r) {
    case 0: s4_wait_vm<0>(); break;
    case 1: s4_wait_vm<2>(); break;
    case 2: s4_wait_vm<4>(); break;
    case 3: s4_wait_vm<6>(); break;
    case 4: s4_wait_vm<8>(); break;
    case 5: s4_wait_vm<10>(); break;
    case 6: s4_wait_vm<12>(); break;
    default: s4_wait_vm<14>(); break;
  }
}
static __device__ __forceinline__ int ps_ll_update_quad(const PsArgs& a, const double* XA, const double* XB, double* C, int npan,
                                                        unsigned* xrI, unsigned* xrJ, unsigned* err, int* sh_q, unsigned lds0,
                                                        const unsigned (&voffX)[2], int ld, int w, int lane, int tid, int I,
                                                        unsigned long long* tr, const unsigned* xcA, const unsigned* xcB, int* sh_p) {
  constexpr unsigned QST = 128 * S4_ROWB;  // one stage: 64 rows of each operand
  constexpr int NST = 8;
  static_assert(NST * QST <= PF_LDS_BYTES, "the quadrant ring lives in the chain role's LDS array");
  const int wr = w >> 1, wc = w & 1;
  unsigned pa[4], pb[4];
  d4 acc[1][2];
  s4_frag_addr(pa, lds0, wr * 16, lane);
  s4_frag_addr(pb, lds0 + 64 * S4_ROWB, wc * 32, lane);
  gk_load_c<1, 2, -64>(C, (size_t)ld, acc, wr * 16, wc * 32, lane);
  // this wave's source: image row R = 16 w + ... is row R of X_I's 64 (waves 0-3) or row R - 64 of X_Jc's 64 (waves 4-7)
  const double* const Xsrc = w < 4 ? XA : XB - (size_t)64 * ld;
  int q = 0;
  while (q < npan) {
    if (tid == 0) {
      int qq = q;
      bool ok = true;
#define PS_READY(p) (ps_ld(xrI + (p)) >= 1u && ps_ld(xrJ + (p)) >= 1u)
      while (qq < npan && PS_READY(qq)) qq++;
      if (qq == q && q == npan - 1) {
        qq = -2;  // caught up at the LAST panel: follow its two blocks column block by column block (below)
      } else {
        if (qq == q) {  // caught up with the factorisation: wait for the next panel
          ok = ps_wait_ge2(xrI + q, 1u, xrJ + q, 1u, err, a.spin_limit);
          qq = q + 1;
          while (ok && qq < npan && PS_READY(qq)) qq++;
        }
        ps_acquire();
        if (!ok) qq = -1;
      }
#undef PS_READY
      *sh_q = qq;
      if (tr && q == 0) tr[1] = wall_clock64();
      if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
    }
    __syncthreads();
    const int qq = *sh_q;
    if (qq == -1) return -1;  // abandoned
    const double* const X0 = Xsrc + (size_t)q * 128;
    if (qq == -2) {
      // ---- the last panel, streamed: chunk c = column block c of X_{I,q} (the streamed solve S(I, q)) and of X_{Jc,q} (the chain
      // helper), each handed over through xcol as its stores complete (or whole, through xready: solves that are not
      // streamed, failed matrices).  Stage c of the ring holds chunk c: nothing is reused inside the panel.
      __syncthreads();  // (sh_q is free again)
      if (tr) tr[1] = wall_clock64();  // (the streamed panel begins)
      int have = 0, issued = 0;
#pragma unroll
      for (int c = 0; c < 8; c++) {
        if (have <= c) {
          __syncthreads();  // (everybody has read the previous round's count)
          if (tid == 0) {
            bool ok = true;
            int h = 0;
            const unsigned long long t0 = wall_clock64();
            for (unsigned it = 0;; it++) {
              const unsigned fa = ps_ld(xrI + q), fb = ps_ld(xrJ + q), ca = ps_ld(xcA), cb = ps_ld(xcB);  // (four loads in flight together)
              const int ha = fa >= 1u ? 8 : (int)ca, hb = fb >= 1u ? 8 : (int)cb;
              h = ha < hb ? ha : hb;
              if (h > c) break;
              __builtin_amdgcn_s_sleep(1);
              if ((it & 15) == 15) {
                if (ps_ld(err) != 0) {
                  ok = false;
                  break;
                }
                if (wall_clock64() - t0 > a.spin_limit) {
                  ps_st(err, 1u);
                  ok = false;
                  break;
                }
              }
            }
            ps_acquire();
            *sh_q = ok ? h : -1;
            if (tr && h == 8) tr[2] = wall_clock64();  // (both blocks are complete: what follows is pure work)
          }
          __syncthreads();
          have = *sh_q;
          __syncthreads();
          if (have < 0) return -1;  // abandoned
        }
#pragma unroll
        for (int s = 0; s < 8; s++)
          if (s >= issued && s < have) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
        issued = have;
        // a LOOK for further column blocks rides along with the wait for chunk c (its flag loads return behind this wave's chunk
        // loads, which it waits for anyway): their loads go out behind this chunk's barrier instead of after a poll of their own
        // (the count goes through one of two LDS words by the parity of c: a wave that reads late still reads ITS round's value)
        int* const slot = (c & 1) ? sh_p : sh_q;
        if (tid == 0) {
          int h = have;
          if (have < 8) {
            const unsigned fa = ps_ld(xrI + q), fb = ps_ld(xrJ + q), ca = ps_ld(xcA), cb = ps_ld(xcB);
            const int ha = fa >= 1u ? 8 : (int)ca, hb = fb >= 1u ? 8 : (int)cb;
            h = ha < hb ? ha : hb;
            if (h > have) ps_acquire();
            else h = have;
            if (tr && h == 8) tr[2] = wall_clock64();
          }
          *slot = h;
        }
        q8_wait_vm(issued - c - 1);  // (the chunks behind this one)
        pf_lds_barrier();            // (LDS only: the younger chunks stay in flight)
        have = *slot;
#pragma unroll
        for (int s = 0; s < 8; s++)
          if (s >= issued && s < have) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
        issued = have;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
        if ((BGP_FAULT_INJECT & 2) && I >= 12 && c == 7) continue;
#endif
        s4_mma<1, 2, -64, 0, 1>(pa, pb, c * QST, acc, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      break;
    }
    int nch = (qq - q) * 8;
#ifdef BGP_FAULT_INJECT  // (see s4_tile: the same fault in the launch-free tile tasks)
    if ((BGP_FAULT_INJECT & 2) && I >= 12 && qq == npan) nch -= 1;
#endif
#pragma unroll
    for (int s = 0; s < NST - 1; s++)
      if (s < nch) s8_issue(X0, voffX, s * S4_KC, lds0 + s * QST, w);
    for (int c = 0; c < nch; c += NST) {
#pragma unroll
      for (int s = 0; s < NST; s++) {
        if (c + s >= nch) break;            // (wave- and workgroup-uniform)
        const int rem = nch - (c + s) - 1;  // chunks behind this one
        q8_wait_vm(rem < NST - 2 ? rem : NST - 2);
        __builtin_amdgcn_s_barrier();  // everybody's share of this chunk has landed; everybody is done with the previous one
        if (c + s + NST - 1 < nch) s8_issue(X0, voffX, (c + s + NST - 1) * S4_KC, lds0 + (unsigned)(((s + NST - 1) % NST) * QST), w);
        s4_mma<1, 2, -64, 0, 1>(pa, pb, s * QST, acc, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();  // (the ring and sh_q are free again)
    q = qq;
  }
  gk_store_c<1, 2, -64>(C, (size_t)ld, acc, wr * 16, wc * 32, lane);
  return 0;
}

// wg = this workgroup's index among the tile workgroups of the launch
template <int PAIR>
static __device__ __forceinline__ void ps_tile_role(const PsArgs& a, int wg) {
  constexpr unsigned AOPB = 128 * S4_ROWB, STAGEB = 256 * S4_ROWB;
  static_assert(PS_NST * STAGEB <= PF_LDS_BYTES, "the operand ring lives in the chain role's LDS array");
  __shared__ int sh_t, sh_q;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)pf_lds_raw();
  const int tid = threadIdx.x, lane0 = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = a.nblk, B = a.B, ld = a.ld;
  unsigned* const flags = a.flags;
  unsigned* const err = flags + PS_ERROR;
  unsigned voffX[2], voffW[2];
  s8_src(voffX, ld, w, lane0);
  s8_src(voffW, 128, w, lane0);
  // XCD affinity (placement only): matrix b belongs to the ticket lists of XCD b % 8 -- where its chain workgroup runs
  // (block b of the chain kernel is dispatched to XCD b % 8) -- so a matrix's panels, W blocks and flags stay in ONE
  // XCD's L2 and the hand-offs are same-XCD; a workgroup whose own list is exhausted helps the next lists.
  const int xcc = (int)(__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u);
  const bool pools = a.ncrit > 0;
  int pool = (pools && wg < a.ncrit) ? 0 : 1, pools_done = 0;
  int list = 0;  // lists tried so far (own first)
  for (;;) {
    const int x = (xcc + list) & 7;
    const int Bx = (B - x + 7) / 8;  // matrices b = x, x + 8, ... < B
    // parts of a P task, of a Dg task, quadrants Q ahead of the critical solve; critical tasks per column and matrix
    const int NP = a.psplit, ND = PS_ND(NP), NQ = PS_NQ(NP), NK = NQ + 1 + NP + ND;
    const int per_matrix = !pools ? ps_tasks_per_matrix(nblk, NP) : (pool == 0 ? ps_crit_per_matrix(nblk, NP) : ps_bulk_per_matrix(nblk));
    if (tid == 0) {
      int tt = -1;
      if (Bx > 0 && per_matrix > 0) {
        tt = (int)__hip_atomic_fetch_add(flags + PS_TICKET + 2 + 8 * pool + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tt >= Bx * per_matrix) tt = -1;
      }
      sh_t = tt;
    }
    __syncthreads();
    int t = sh_t;
    __syncthreads();
    if (t < 0) {  // this list is finished: next one, then (pools) the other pool, or done
      if (++list == 8) {
        if (!pools || ++pools_done == 2) return;
        pool ^= 1;
        list = 0;
      }
      continue;
    }
    // (trace slot: unique per (pool, list, ticket) while B % 8 == 0)
    const int tglobal = (pools && pool == 1 ? B * ps_crit_per_matrix(nblk, NP) : 0) + (int)(((long long)t * 8 + x) % ((long long)B * per_matrix));
    // ---- ticket -> (column J, matrix b, kind, block row I).  kind 0: S(I, J); 1: P(I), part `part` of NP; 2: Dg(I), part of ND;
    // 3: Q(I), quadrant `part` of block (I, J) with I = J + 2.  Order inside a column's critical group: Q, S, P, Dg -- a task only
    // ever waits for tasks with EARLIER tickets (or for the chain): Q for the bulk solves of column J - 1, S(J+2, J) for its Q.
    int J = 0, kq = NQ, I;
    if (pools && pool == 0) {  // NK critical tasks per column and matrix
      J = t / (NK * Bx);
      t -= J * NK * Bx;
      kq = t / Bx;
      I = J + 2;
    } else {
      const int head = pools ? 0 : NK;  // (one list: the critical tasks lead their column)
      for (;;) {
        const int c = (head + nblk - J - 3) * Bx;
        if (t < c) break;
        t -= c;
        J++;
      }
      const int q0 = t / Bx;
      kq = q0 < head ? q0 : NQ;  // (a bulk task is a solve)
      I = q0 < head ? J + 2 : J + 3 + (q0 - head);
    }
    const int kind = kq < NQ ? 3 : (kq == NQ ? 0 : (kq <= NQ + NP ? 1 : 2));
    const int part = kind == 3 ? kq : (kind == 1 ? kq - NQ - 1 : kq - NQ - NP - 1);
    const int b = x + 8 * (t % Bx);
    const bool presub = kind == 1, diag = kind == 2, qpre = kind == 3;
    const int Jc = qpre ? J : J + kind;                  // block column of the task's block
    const int npan = (kind == 0 || qpre) ? J : J + 1;    // panels 0 .. npan-1 are applied here
    const bool qsolve = kind == 0 && NQ > 0 && I == J + 2;  // the critical solve: its block's pre-update came in quadrants
    int lane = lane0;
    asm volatile("" : "+v"(lane));  // (per-lane addresses of a task are formed in the task: hoisted out of this loop they spilled)
    unsigned* const wready = flags + PS_HDR + (size_t)b * nblk;
    unsigned* const diagrdy = flags + PS_HDR + (size_t)B * nblk + (size_t)b * nblk;
    unsigned* const xrI = flags + PS_HDR + (size_t)2 * B * nblk + ((size_t)b * nblk + I) * nblk;
    unsigned* const xrJ = flags + PS_HDR + (size_t)2 * B * nblk + ((size_t)b * nblk + Jc) * nblk;
    unsigned* const subrdy = flags + PS_HDR + (size_t)B * nblk * (2 + nblk) + (size_t)b * nblk;
    unsigned* const s2rdy = flags + PS_S2RDY(B, nblk) + (size_t)b * nblk;
    const int* const stat = a.status + b;
    double* const M = a.K + (size_t)b * a.mstride;
    double* const C = M + (size_t)I * 128 * ld + Jc * 128;
    unsigned long long* const tr = (a.trace && tid == 0) ? a.trace + (size_t)B * nblk * 8 + (size_t)tglobal * 8 : nullptr;
    if (tr) {
      tr[0] = wall_clock64();
      tr[7] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | ((unsigned long long)kind << 28) |
              ((unsigned long long)Jc << 20) | ((unsigned long long)I << 12) | (unsigned long long)b;
    }
    if (tid == 0) sh_q = (__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 1 : 0;
    __syncthreads();
    bool dead = sh_q != 0;  // the matrix has failed: nothing to compute, the task only passes its flag on
    __syncthreads();
    if (qpre) {
      // ---- 1p. a quadrant of block (J+2, J) with the panels 0 .. J-1, the last one (X_{J+2,J-1}: a solve of the previous column,
      // streamed with BGP_PS_STREAM >= 3; X_{J,J-1}: the chain helper's block) chunk by chunk: the 14 us of MFMA that the last
      // panel's term costs one CU sat between the helper's block and the start of the column's critical solve
      if (npan > 0 && !dead) {
        const double* const XI = M + (size_t)I * 128 * ld;
        if (ps_ll_update_quad(a, XI + (size_t)(64 * (part >> 1)) * ld, M + ((size_t)Jc * 128 + 64 * (part & 1)) * ld,
                              C + (size_t)(64 * (part >> 1)) * ld + 64 * (part & 1), npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane, tid, I,
                              tr, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 2, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + Jc) * 3,
                              &sh_t) < 0)
          return;  // abandoned
      }
      if (tr) tr[3] = wall_clock64();
      ps_publish_barrier();
      if (tid == 0) ps_signal_add(s2rdy + I);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    if (qsolve) {
      // ---- 1s. the critical solve's block arrives pre-updated: wait for its four quadrants
      if (tid == 0) {
        const bool ok = ps_wait_ge(s2rdy + I, 4u, err, a.spin_limit);
        ps_acquire();
        sh_q = !ok ? -1 : (__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : 1);
        if (tr) tr[1] = tr[2] = wall_clock64();
      }
      __syncthreads();
      const int r = sh_q;
      __syncthreads();
      if (r < 0) return;
      dead = dead || r == 0;
    } else if (npan > 0 && !dead && diag && ND == 3) {
      // ---- 1q. a diagonal block in quadrants (0,0), (1,0), (1,1), each the quadrant update of a P block with X_I on both sides (the
      // diagonal quadrants compute their upper tiles too -- nobody reads those): a third of the 8 us of MFMA a whole diagonal
      // block's last panel costs one CU, and the last panel streamed behind the solve S(I, I-2) like the P quadrants
      const double* const XI = M + (size_t)I * 128 * ld;
      const int qr = part == 0 ? 0 : 1, qc = part == 2 ? 1 : 0;
      const unsigned* const xc = flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 1;
      if (ps_ll_update_quad(a, XI + (size_t)(64 * qr) * ld, XI + (size_t)(64 * qc) * ld, C + (size_t)(64 * qr) * ld + 64 * qc, npan, xrI, xrI, err,
                            &sh_q, lds0, voffX, ld, w, lane, tid, I, tr, xc, xc, &sh_t) < 0)
        return;  // abandoned
    } else if (npan > 0 && !dead && diag) {
      // ---- 1d. a diagonal block: its 36 lower 16 x 16 tiles, dealt to the eight waves
      const double* const XA = M + (size_t)I * 128 * ld;
      const int lr = lane & 15, lk = lane >> 4;
      unsigned p0[4], oa[5], ob[5];
      s4_frag_addr(p0, lds0, 0, lane);
      d4 dt[5];
      int offc[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        const int tt = w + 8 * u;
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tt) ti++;
        const int tj = tt - ti * (ti + 1) / 2;
        oa[u] = (unsigned)(ti * 16 * S4_ROWB);
        ob[u] = (unsigned)(tj * 16 * S4_ROWB);
        offc[u] = (ti * 16 + lk) * ld + tj * 16 + lr;
        if (tt < 36) {
#pragma unroll
          for (int r = 0; r < 4; r++) dt[u][r] = C[(size_t)offc[u] + (size_t)(4 * r) * ld];
        } else {
          dt[u] = (d4){0.0, 0.0, 0.0, 0.0};
        }
      }
      int q = 0;
      while (q < npan) {
        if (tid == 0) {
          int qq = q;
          bool ok = true;
          while (qq < npan && ps_ld(xrI + qq) >= 1u) qq++;
          if (qq == q) {  // caught up with the factorisation: wait for the next panel
            ok = ps_wait_ge(xrI + q, 1u, err, a.spin_limit);
            qq = q + 1;
            while (ok && qq < npan && ps_ld(xrI + qq) >= 1u) qq++;
          }
          ps_acquire();
          sh_q = ok ? qq : -1;
          if (tr && q == 0) tr[1] = wall_clock64();
          if (tr && qq == npan) tr[2] = wall_clock64();  // (the last panels are ready: what follows is pure work)
        }
        __syncthreads();
        const int qq = sh_q;
        if (qq < 0) return;  // abandoned
        if (w < 4)
          s8_ring_run_diag<PS_NST, 5>(XA + (size_t)q * 128, voffX, (qq - q) * 8, lds0, p0, oa, ob, dt, w);
        else
          s8_ring_run_diag<PS_NST, 4>(XA + (size_t)q * 128, voffX, (qq - q) * 8, lds0, p0, oa, ob, dt, w);
        __syncthreads();  // (the ring and sh_q are free again)
        q = qq;
      }
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (w + 8 * u < 36) {
#pragma unroll
          for (int r = 0; r < 4; r++) C[(size_t)offc[u] + (size_t)(4 * r) * ld] = dt[u][r];
        }
      }
    } else if (npan > 0 && !dead) {
      // ---- 1. left-looking update with the panels 0 .. npan-1: waves as 4 x 2, each 32 rows x 64 columns (2 x 4 MFMA tiles);
      // a P task split NP ways owns 128 / NP columns of its block (each wave 32 rows x 32 or 16 columns)
      const double* const XA = M + (size_t)I * 128 * ld;
      int rc;
      if (presub && NP == 2)
        rc = ps_ll_update<2>(a, XA, M + ((size_t)Jc * 128 + 64 * part) * ld, C + 64 * part, npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w,
                             lane, tid, I, tr);
      else if (presub && NP == 4)  // quadrants: rows 64 (part >> 1) .., columns 64 (part & 1) ..
        rc = ps_ll_update_quad(a, XA + (size_t)(64 * (part >> 1)) * ld, M + ((size_t)Jc * 128 + 64 * (part & 1)) * ld,
                               C + (size_t)(64 * (part >> 1)) * ld + 64 * (part & 1), npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane,
                               tid, I, tr, flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + I) * 3 + 1,
                               flags + PS_XCOL(B, nblk) + ((size_t)b * nblk + Jc) * 3, &sh_t);
      else
        rc = ps_ll_update<4>(a, XA, M + (size_t)Jc * 128 * ld, C, npan, xrI, xrJ, err, &sh_q, lds0, voffX, ld, w, lane, tid, I, tr);
      if (rc < 0) return;  // abandoned
    }
    if (tr) tr[3] = wall_clock64();
    if (diag || presub) {
      ps_publish_barrier();
      if (tid == 0) ps_signal_add((diag ? diagrdy : subrdy) + I);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    // ---- 2. panel solve against W_JJ: waves stacked along the rows, 16 rows x 128 columns each
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's updated values have left the wave
    if (PAIR && I <= J + a.ncrit_stream) {  // chain pairs: the critical solve of the column follows pf_block(J) row block by row block
      __syncthreads();  // every wave's part of the block is in memory (the solve re-reads it as A fragments)
      const int r = dead ? -2 : pf_stream_S(a, b, J, I, &sh_q, &sh_t, tr);
      if (r == -1) return;
      if (tr) tr[5] = wall_clock64();
      ps_publish_barrier();
      if (tid == 0) ps_signal_add(xrI + J);
      if (tr) tr[6] = wall_clock64();
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      const bool ok = ps_wait_ge(wready + J, 1u, err, a.spin_limit);
      ps_acquire();
      sh_q = !ok ? -1 : (__hip_atomic_load(stat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 0 : 1);
      if (tr) tr[4] = wall_clock64();
    }
    __syncthreads();
    if (sh_q < 0) return;
    dead = dead || sh_q == 0;
    if (!dead) {
      const double* const Wm = a.W + ((size_t)b * nblk + J) * (128 * 128);
      const int r0 = w * 16;
      unsigned pa[4], pb[4];
      s4_frag_addr(pa, lds0, r0, lane);
      s4_frag_addr(pb, lds0 + AOPB, 0, lane);
      d4 acc[1][8];
#pragma unroll
      for (int j = 0; j < 8; j++) acc[0][j] = (d4){0.0, 0.0, 0.0, 0.0};
      // (W_JJ is lower triangular: chunk c only reaches the column blocks j >= c -- `tri`; its never-written upper
      // blocks are staged all the same, which keeps the instruction count per chunk fixed for the counted waits)
      // the right-hand-side operands are fetched under the solve (z_J: this wave's 8 columns; y_I: this lane's rows)
      const double* const zk = a.yw + (size_t)b * a.ystride + J * 128;
      double zc[8], yv[4];
#pragma unroll
      for (int j = 0; j < 8; j++) zc[j] = zk[GK_COLB(0, j, lane)];
      double* const yi = a.yw + (size_t)b * a.ystride + I * 128;
#pragma unroll
      for (int r = 0; r < 4; r++) yv[r] = yi[GK_ROWB(r0, 0, lane, r)];
      s8_ring_run<PS_NST, 1, 8, 0>(C, voffX, Wm, voffW, 8, lds0, pa, pb, acc, w, 1);
      // in place (every read of these rows was staged before the last barrier), right-hand side in the same pass:
      // one wave per row, fixed shuffle order (as trsm4_kernel)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = GK_ROWB(r0, 0, lane, r);
        double part = 0.0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const double x = acc[0][j][r];
          C[(size_t)row * ld + GK_COLB(0, j, lane)] = x;
          part += x * zc[j];
        }
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        if ((lane & 15) == 0) yi[row] = yv[r] - part;
      }
    }
    if (tr) tr[5] = wall_clock64();
    ps_publish_barrier();
    if (tid == 0) ps_signal_add(xrI + J);
    if (tr) tr[6] = wall_clock64();
    __syncthreads();
  }
}

// The launch-free factorisation: workgroups 0 .. B-1 are the chain (one per matrix, bgp_pf.h), the others the tile workers.
// One LDS array serves both roles (157 KB: one workgroup per CU); workgroups are placed in index order, so the chain is
// resident before any tile worker starts to spin.
template <int PAIR>
__global__ void __launch_bounds__(512, 1) ps_kernel(PsArgs a) {
  const int id = (int)blockIdx.x;
  if (id >= a.nchain) {
    ps_tile_role<PAIR>(a, id - a.nchain);
  } else if (!PAIR) {
    ps_chain_role<0>(a, id, 0);
  } else {
    const int p = id >= a.Bpad ? 1 : 0, b = id - p * a.Bpad;
    if (b < a.B) ps_chain_role<1>(a, b, p);  // (the padding slots of a pair group exit at once: their CUs go to tile workers)
  }
}

void bgp_launch_ps(hipStream_t st, const PsArgs& a, int nwg) {
  if (a.pair)
    hipLaunchKernelGGL(ps_kernel<1>, dim3(nwg), dim3(512), 0, st, a);
  else
    hipLaunchKernelGGL(ps_kernel<0>, dim3(nwg), dim3(512), 0, st, a);
}
int bgp_ps_total_tasks(int B, int nblk, int np) { return B * ps_tasks_per_matrix(nblk, np); }

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
  s4_tile<T, 2, 2, -64, 0, 0, 0, 0, (MODE != 0) ? 1 : 0, (MODE == 0) ? 1 : 0>(nullptr, lds0, cur, voff, ldx, K, wr * (T / 2),
                                                                        wc * (T / 2), w, lane, S4Gen(), ldc);
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
// T = 64 by default; BGP_ROWQUAD_T=128 selects the 128-wide tile (measured equal within noise at 128 posteriors x
// 10 000 points x n = 1024: 46 vs 43 ms per batched predict -- the kernel is bound by streaming K_* (83 MB per
// posterior, read by every XCD), not by the tile shape).
int bgp_rowquad_tile() {
  static int t = 0;
  if (!t) {
    const char* e = getenv("BGP_ROWQUAD_T");
    t = (e && atoi(e) == 128) ? 128 : 64;
  }
  return t;
}
void bgp_launch_rowquad(hipStream_t st, const double* A, int lda, size_t sA, const double* S, int lds_, size_t sS,
                        const int* pidx, int M, int n, int nb, double* part) {
  const int T = bgp_rowquad_tile(), tn = n / T, tm = M / T, tiles = tm * tn;
  const int grid = (nb >= 8) ? 8 * ((nb + 7) / 8) * tiles : 8 * ((tm + 7) / 8) * tn * nb;
  if (T == 64)
    hipLaunchKernelGGL(rowquad4_kernel<64>, dim3(grid), dim3(256), 0, st, A, lda, sA, S, lds_, sS, pidx, tn, M, nb, part);
  else
    hipLaunchKernelGGL(rowquad4_kernel<128>, dim3(grid), dim3(256), 0, st, A, lda, sA, S, lds_, sS, pidx, tn, M, nb, part);
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
    hipLaunchKernelGGL((syrk4_kernel<TT, V>), dim3(total), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K,    \
                       jstart, colmode, B, total, trace, S4_PW, S4Gen());                                            \
    return total;                                                                                                    \
  }
  S4_CASE(128, 0) S4_CASE(128, 4) S4_CASE(64, 0) S4_CASE(64, 1) S4_CASE(64, 2) S4_CASE(64, 3) S4_CASE(64, 4)
  fprintf(stderr, "bgp_debug_launch_syrk4: no instantiation T=%d var=%d\n", T, var);
  return 0;
}
#endif
