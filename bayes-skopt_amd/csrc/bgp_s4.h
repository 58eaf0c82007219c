// The LDS-DMA ring shared by the trailing update, the panel solve, the NT products (bgp_syrk4.hip) and the tile workers of the
// launch-free factorisation (bgp_ps.hip): operand issue (global -> LDS with global_load_lds_dwordx4, XOR-swizzled granules),
// fragment addresses, the MFMA step, one tile's accumulate-and-store (s4_tile) and the tile lists of a launch.  gfx950 only.
#pragma once
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"
#include "bgp_ring.h"

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

// ---- Gram generation at first touch (the trailing update of the FIRST panel group, bgp_chol.hip) ----
// The update that touches a block of the kernel matrix first does not load it: every lane computes the Gram values of its own
// accumulator entries -- from the k-major pre-scaled inputs xscale_kernel wrote for kbuild2_kernel, with kbuild2's arithmetic per
// element (differences and fma in ascending dimension, kb_stationary, constant, exact diagonal, identity padding: same bits) --
// while the first operand chunk is in flight; the Gram kernel in front of the factorisation then builds block column 0 only.
// Why: the build is fp64-VALU bound (VALUBusy 84 %) and the update MFMA bound, and the two pipes of a SIMD work side by side for
// different waves; and the block is neither written by one kernel nor read back by the next (2 x 2.3 GB per half-step at
// n = 2048 x 128 matrices).
// (struct S4Gen: bgp_common.h)
struct S4NoGen {
  static constexpr int active = 0;
};

// The 64-row slices of the tile's row block and column block for 16 dimensions are staged in the ring's SECOND stage (free until
// the main loop's first barrier has been passed): xi then xj, 16 x 64 doubles each, rows permuted so that the four rows of a lane's
// accumulator register quad (row = lq + 4 r) lie side by side.
template <int STAT, int FORM>
struct S4GenF {
  static constexpr int active = 1;
  S4Gen g;
  template <int NR, int NC, int CREL>
  __device__ __forceinline__ void init(const S4Tile& cur, unsigned lds_stage, d4 (&acc)[NR][NC], int r0, int c0, int lane) const {
#pragma clang fp contract(off)
    typedef __attribute__((address_space(3))) double* lds_dp;
    typedef __attribute__((address_space(3))) const double* lds_cdp;
    typedef __attribute__((address_space(3))) const d2* lds_cd2p;
    const int tid = threadIdx.x, lr = lane & 15, lq = lane >> 4;
    const double* Xb = g.Xs + (size_t)cur.b * g.dpad * g.npad;
    const double* h = g.H + (size_t)cur.b * (g.d + 2);
#pragma unroll
    for (int i = 0; i < NR; i++)
#pragma unroll
      for (int j = 0; j < NC; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
    const int sk = tid >> 4, sseg = tid & 15;  // staging: dimension sk, rows 4 sseg .. 4 sseg + 3 of both slices
    {  // (dpad == 16: one pass over the input dimensions -- bgp_lml_gen_eligible)
      {
        const double* src = Xb + (size_t)sk * g.npad;
        const d2 a0 = *reinterpret_cast<const d2*>(src + cur.gi0 + 4 * sseg), a1 = *reinterpret_cast<const d2*>(src + cur.gi0 + 4 * sseg + 2);
        const d2 b0 = *reinterpret_cast<const d2*>(src + cur.gj0 + 4 * sseg), b1 = *reinterpret_cast<const d2*>(src + cur.gj0 + 4 * sseg + 2);
        // row rho = 4 sseg + e of the slice -> slot (rho & 48) + (rho & 3) * 4 + ((rho >> 2) & 3)
        const unsigned slot = (unsigned)(((4 * sseg) & 48) + ((sseg & 3)));
        const unsigned wi = lds_stage + (unsigned)(sk * 64 + slot) * 8u, wj = wi + 8192u;
        *(lds_dp)(uintptr_t)(wi) = a0[0];
        *(lds_dp)(uintptr_t)(wi + 32u) = a0[1];
        *(lds_dp)(uintptr_t)(wi + 64u) = a1[0];
        *(lds_dp)(uintptr_t)(wi + 96u) = a1[1];
        *(lds_dp)(uintptr_t)(wj) = b0[0];
        *(lds_dp)(uintptr_t)(wj + 32u) = b0[1];
        *(lds_dp)(uintptr_t)(wj + 64u) = b1[0];
        *(lds_dp)(uintptr_t)(wj + 96u) = b1[1];
      }
      __syncthreads();
#pragma unroll 4
      for (int k = 0; k < S4_KC; k++) {
        double a[NR][4], b[NC];
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const unsigned ra = lds_stage + (unsigned)(k * 64 + r0 + 16 * i + 4 * lq) * 8u;
          const d2 lo = *(lds_cd2p)(uintptr_t)(ra), hi = *(lds_cd2p)(uintptr_t)(ra + 16u);
          a[i][0] = lo[0];
          a[i][1] = lo[1];
          a[i][2] = hi[0];
          a[i][3] = hi[1];
        }
#pragma unroll
        for (int j = 0; j < NC; j++) {
          // (column c of the slice sits at its permuted slot too)
          const int cc = c0 + 16 * j + lr;
          b[j] = *(lds_cdp)(uintptr_t)(lds_stage + 8192u + (unsigned)(k * 64 + (cc & 48) + (cc & 3) * 4 + ((cc >> 2) & 3)) * 8u);
        }
#pragma unroll
        for (int i = 0; i < NR; i++)
#pragma unroll
          for (int j = 0; j < NC; j++) {
            if (j + CREL > i) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const double df = a[i][r] - b[j];
              acc[i][j][r] = __builtin_fma(df, df, acc[i][j][r]);
            }
          }
      }
    }
    const double cst = exp(h[0]);
    const bool interior = !cur.diag && cur.gi0 + 64 <= g.n && cur.gj0 + 64 <= g.n;  // (workgroup-uniform)
    if (interior) {
#pragma unroll
      for (int i = 0; i < NR; i++)
#pragma unroll
        for (int j = 0; j < NC; j++) {
          if (j + CREL > i) continue;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double sv = kb_stationary<STAT>(acc[i][j][r]);
            acc[i][j][r] = (FORM == BGP_FORM_PRODUCT) ? cst * sv : cst + sv;
          }
        }
      return;
    }
    const double s2 = exp(h[g.d + 1]);
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
          if (gi >= g.n || gj >= g.n) {
            v = (gi == gj) ? 1.0 : 0.0;  // identity padding: log det and z unaffected
          } else if (gi == gj) {
            // fill_diagonal(1) (kernels.py:1738) -> c*1 (+1) -> + s2 (White) -> += alpha (_gpr.py:585)
            const double base = (FORM == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
            v = (base + s2);
            if (g.alpha) v += g.alpha[gi];
          } else {
            const double sv = kb_stationary<STAT>(acc[i][j][r]);
            v = (FORM == BGP_FORM_PRODUCT) ? cst * sv : cst + sv;
          }
          acc[i][j][r] = v;
        }
      }
  }
};

// One tile for a wave's NR x NC block at (r0, c0) of the T x T workgroup tile.
// NEGA = 1: C -= A B^T (the factorisation's update), 0: C += A B^T; ZEROC: C is not loaded (starts from zero); ldc = leading
// dimension of C when it is not the operands' (gemm4_kernel).
// GENF: S4NoGen, or S4GenF<STAT, FORM> (T = 64): C is generated instead of loaded (every wave of the workgroup must come through
// here exactly once: the generator has barriers of its own).
template <int T, int NR, int NC, int CREL, int VAR, int NEGA = 1, int ZEROC = 0, class GENF = S4NoGen>
static __device__ __forceinline__ void s4_tile(unsigned long long* trace, unsigned lds0, const S4Tile& cur,
                                               const unsigned (&voff)[T / 32], int ld, int K, int r0, int c0, int w,
                                               int lane, int ldc = 0, const GENF& genf = GENF()) {
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
  if constexpr (GENF::active) {
    static_assert(T == 64 && !ZEROC, "Gram generation is written for the 64 x 64 tile");
    genf.template init<NR, NC, CREL>(cur, lds0 + STAGEB, acc, r0, c0, lane);
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

