// Diagonal-block factorisation shared by potrf_kernel (bgp_chol.hip) and the launch-free factorisation's chain role
// (ps_kernel, bgp_syrk4.hip): the 128 x 128 block in LDS, its Cholesky factor, the inverse of the factor, z = W y, log-det
// (pf_block); and the chain's own share of a block column (pf_chain_next).  gfx950 only.
#pragma once
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"

#ifndef PF_T
#define PF_T(i)
#endif
#ifndef PF_TW
#define PF_TW(i)
#endif
#define PF_LD 130   // LDS leading dimension of the 128x128 block (== 2 mod 32: conflict-free MFMA operand reads)
#define PF_MLD 18   // leading dimension of the 16x16 inverse blocks

// 16x16 micro-Cholesky fused with the inverse of its factor, one matrix row per lane (lane & 15), all
// 16 pivots unrolled at compile time.  Pivot J broadcasts L[c][J] (lane c of register a[J]) to the whole
// 16-lane row with ONE 64-bit DPP move (row_newbcast) and uses it twice:
//   a[c]    -= L[lane][J] * L[c][J]          right-looking update of the block (lane = row)
//   macc[c] += L[c][J] * M[J][lane]          forward substitution for M = L^-1 (lane = column of M)
// so no SGPR round trip (v_readlane pairs) and the two dependency chains interleave.
template <int C>
static __device__ __forceinline__ double bc16(double v) {
  return __builtin_amdgcn_update_dpp(v, v, 0x150 + C, 0xF, 0xF, false);  // row_newbcast:C (all lanes written)
}

// One fused instruction per update: v_fmac_f64 with a DPP row broadcast on its first source (64-bit DPP
// supports exactly this control on gfx90a+):  D += bcast_C(a[J]) * S1.
#define PF_FMAC_BCAST(D, SRC, S1, C)                                                                  \
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(D) : "v"(SRC), "v"(S1), "n"(C))

template <int J, int C>
static __device__ __forceinline__ void micro_cols(double (&a)[16], double (&macc)[16], double naj, double mj) {
  if constexpr (C < 16) {
    PF_FMAC_BCAST(a[C], a[J], naj, C);     // a[C]    -= L[lane][J] * L[C][J]
    PF_FMAC_BCAST(macc[C], a[J], mj, C);   // macc[C] += L[C][J] * M[J][lane]
    micro_cols<J, C + 1>(a, macc, naj, mj);
  }
}

// The pivot's root and reciprocal root together, from the hardware seed v_rsq_f64 (relative error up to 1e-7):
// one coupled Goldschmidt step (rounds 1-3 stopped here: sqrt within 36 ulp, 1/sqrt within 20 ulp of the correctly rounded
// values over 1e7 arguments in [1e-12, 1e6] -- tools/pivot_sqrt_probe.hip) and one residual correction of each:
//     dj  += (x - dj^2) * inv / 2          ->  sqrt:   <= 0.5 ulp (99.98 % correctly rounded, like LAPACK's dpotrf pivot)
//     inv += inv * (1 - dj * inv)          ->  1/sqrt: <= 1.5 ulp
// (tests/test_gpu_lml.py::test_pivot_root_accuracy asserts the bounds through bgp_debug_pivot_root).  Four more dependent
// operations per pivot: 33 -> 40 ns per root on a dependent chain, < 1 us per 128 x 128 diagonal block.
static __device__ __forceinline__ void pf_pivot_root(double x, double& dj, double& inv) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double g = x * y0, h = 0.5 * y0;
  const double r = fma(-g, h, 0.5);
  const double g1 = fma(g, r, g);
  const double y1 = fma(y0, r, y0);
  const double hy = 0.5 * y1;
  const double d = fma(-g1, g1, x);
  dj = fma(d, hy, g1);
  const double e = fma(-dj, y1, 1.0);
  inv = fma(y1, e, y1);
}

// (No validity test inside the chain: a non-positive, NaN or overflowed pivot makes its own root and everything behind it NaN /
// Inf, so the FIRST diagonal entry of the finished block that is not a positive finite number is the failing pivot -- looked for
// by the idle wave, one step behind and off the wave that runs the 16 dependent pivots: pf_check_diag.  The test cost that wave
// five instructions per pivot of ~35.)
template <int J>
static __device__ __forceinline__ void micro_chol_inv(double (&a)[16], double (&macc)[16], double (&mrow)[16], int lr) {
  if constexpr (J < 16) {
    asm volatile("s_nop 1");  // a[J] was last written by the inline-asm updates above: DPP read hazard (2 wait states)
    const double djj = bc16<J>(a[J]);
    double dj, inv;  // sqrt(djj), 1 / sqrt(djj)
    pf_pivot_root(djj, dj, inv);
    a[J] = (lr == J) ? dj : a[J] * inv;
    const double naj = -a[J];
    const double mj = (((lr == J) ? 1.0 : 0.0) - macc[J]) * inv;  // M[J][lane]
    mrow[J] = mj;
    if constexpr (J < 15) asm volatile("s_nop 1" ::"v"(a[J]), "v"(naj), "v"(mj));  // VALU write -> DPP read
    micro_cols<J, J + 1>(a, macc, naj, mj);
    micro_chol_inv<J + 1>(a, macc, mrow, lr);
  }
}

// The validity test of the 16 pivots of diagonal sub-block `blk`, off the pivot chain: lanes 0 .. 15 of the calling wave (the idle
// wave 4) look at the block's diagonal in the LDS tile; the first entry that is not a positive finite number is the failing pivot
// (everything behind a failed pivot is NaN).  *fail = its 1-based index inside the 128 x 128 block, if none was recorded before.
static __device__ __forceinline__ void pf_check_diag(const double* __restrict__ s, int blk, int lane, int* fail, int ldt) {
  const int i = blk * 16 + (lane & 15);
  const double v = s[i * ldt + i];
  const unsigned long long m = __ballot(!(v > 0.0 && v < INFINITY)) & 0xffffull;
  if (m != 0 && lane == 0 && *fail == 0) *fail = blk * 16 + __ffsll((long long)m);
}

// acc (+/-)= sum_{t < nt} A_t B_t^T for 16x16 blocks whose operands sit in LDS rows `pa` / `pb` (pointers
// already offset to the lane's row and k-group) and advance by one 16-wide block column per term.  The
// next term's eight operand reads are issued before the current term's four MFMAs (one wave per SIMD
// here: nothing else hides the LDS latency).
template <int NEG>
static __device__ __forceinline__ d4 mma_run(d4 acc, const double* __restrict__ pa, const double* __restrict__ pb,
                                             int nt) {
  if (nt <= 0) return acc;
  double a0[4], b0[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    a0[kk] = pa[kk * 4];
    b0[kk] = pb[kk * 4];
  }
  for (int t = 0; t < nt; t++) {
    double a1[4], b1[4];
    const int tn = (t + 1 < nt) ? t + 1 : t;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      a1[kk] = pa[tn * 16 + kk * 4];
      b1[kk] = pb[tn * 16 + kk * 4];
    }
#pragma unroll
    for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a0[kk] : a0[kk], b0[kk], acc, 0, 0, 0);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      a0[kk] = a1[kk];
      b0[kk] = b1[kk];
    }
  }
  return acc;
}

// Row I of W = L^-1, block J <= I:  W[I][I] = M_I,  W[I][J] = -M_I * S,  S = sum_{K=J}^{I-1} L[I][K] W[K][J].
// pf_wsum forms the terms K = J .. Kend-1 of S (L[I][I-1] comes from `xrow` while the panel wave's in-place
// write of that block is still pending); pf_wfinish multiplies by -M_I (the C-layout sum is directly the B
// operand), writes the block to global memory (row-major W, lower blocks only: the panel solves skip k-steps
// beyond a column block) and transposed into LDS block (J, I).
static __device__ __forceinline__ d4 pf_wsum(d4 acc, const double* __restrict__ s, const double* __restrict__ Minv,
                                             const double* __restrict__ xrow, int I, int J, int Kbeg, int Kend,
                                             int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  if (Kbeg == J && Kbeg < Kend) {  // W[J][J] = M_J
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const double av = (J == I - 1 && xrow) ? xrow[lr * PF_MLD + kk * 4 + lk] : s[(I * 16 + lr) * PF_LD + J * 16 + kk * 4 + lk];
      const double bv = Minv[J * 16 * PF_MLD + (kk * 4 + lk) * PF_MLD + lr];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
    Kbeg++;
  }
  int Kmid = Kend;
  if (xrow && Kend == I) Kmid = I - 1;  // the last term's A operand lives in xrow
  if (Kmid > Kbeg)
    acc = mma_run<0>(acc, &s[(I * 16 + lr) * PF_LD + Kbeg * 16 + lk], &s[(J * 16 + lr) * PF_LD + Kbeg * 16 + lk],
                     Kmid - Kbeg);
  if (Kmid < Kend && Kmid >= Kbeg) {
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const double av = xrow[lr * PF_MLD + kk * 4 + lk];
      const double bv = s[(J * 16 + lr) * PF_LD + Kmid * 16 + kk * 4 + lk];
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
    }
  }
  return acc;
}

static __device__ __forceinline__ void pf_wfinish(d4 acc, double* __restrict__ s, const double* __restrict__ Minv,
                                                  double* __restrict__ Wg, int I, int J, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  d4 wn;
  if (J == I) {
#pragma unroll
    for (int r = 0; r < 4; r++) wn[r] = Minv[I * 16 * PF_MLD + (lk + 4 * r) * PF_MLD + lr];
  } else {
    wn = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const double av = -Minv[I * 16 * PF_MLD + lr * PF_MLD + kk * 4 + lk];
      wn = __builtin_amdgcn_mfma_f64_16x16x4f64(av, acc[kk], wn, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) s[(J * 16 + lr) * PF_LD + I * 16 + lk + 4 * r] = wn[r];
  }
  if (Wg) {
#pragma unroll
    for (int r = 0; r < 4; r++) Wg[(I * 16 + lk + 4 * r) * 128 + J * 16 + lr] = wn[r];
  }
}

#define PF_THREADS 512
// GEN (fused small-n LML, n <= 128: SURVEY.md section 7 step 5): the workgroup GENERATES the jittered Gram matrix of
// its walker straight into the LDS tile (same arithmetic, in the same order, as kbuild_tile: scaled inputs, squared
// differences in dimension order, stationary kernel, exact diagonal, identity padding) instead of loading a tile
// another launch wrote, takes y from the context and stores nothing but lml / status: ONE launch per LML batch, no
// Gram matrix in HBM.  `gen` carries the extra inputs.
struct PfGen {
  const double* X;      // n x d training inputs (original or warped)
  const double* alpha;  // n diagonal terms
  const double* H;      // B x (d + 2) canonical hyper-parameters
  const double* y;      // npad right-hand side (zero padded)
  int d;
};

template <int STAT, int FORM>
static __device__ __forceinline__ void pf_generate_tile(double* __restrict__ s, double* __restrict__ xs,
                                                        const PfGen& g, const double* __restrict__ h, int n, int tid) {
  // thread (tx, ty) of a 16 x 32 grid owns rows ty + 32 r (r < 4) and columns tx + 16 c (c < 8)
  const int tx = tid & 15, ty = tid >> 4;
  double acc[4][8];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) acc[r][c] = 0.0;
  const int d = g.d;
  for (int k0 = 0; k0 < d; k0 += 16) {
    const int kc = min(16, d - k0);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += PF_THREADS) {
      const int row = idx / kc, kk = idx - row * kc;
      xs[kk * 128 + row] = (row < n) ? g.X[(size_t)row * d + k0 + kk] / exp(h[1 + k0 + kk]) : 0.0;
    }
    __syncthreads();
    for (int kk = 0; kk < kc; kk++) {
      double a[4], bb[8];
#pragma unroll
      for (int r = 0; r < 4; r++) a[r] = xs[kk * 128 + ty + 32 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) bb[c] = xs[kk * 128 + tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const double df = a[r] - bb[c];
          acc[r][c] = fma(df, df, acc[r][c]);
        }
    }
  }
  {
#pragma clang fp contract(off)
  const double cst = exp(h[0]), s2 = exp(h[d + 1]);
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = ty + 32 * r;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gj = tx + 16 * c;
      double v;
      if (gi >= n || gj >= n) {
        v = (gi == gj) ? 1.0 : 0.0;
      } else if (gi == gj) {
        const double base = (FORM == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
        v = (base + s2) + g.alpha[gi];
      } else {
        const double sv = kb_stationary<STAT>(acc[r][c]);
        v = (FORM == BGP_FORM_PRODUCT) ? cst * sv : cst + sv;
      }
      s[gi * PF_LD + gj] = v;
    }
  }
}
}

// The workgroup's LDS: ONE set of function-scope arrays that pf_block and the chain kernel's own steps (pf_chain_next)
// both reach through this accessor (157.6 of the 160 KB of a CU).
struct PfLds {
  double* s;      // 128 x PF_LD: the diagonal block; lower triangle -> L, upper triangle <- W^T block by block
  double* Minv;   // 8 inverses of the 16 x 16 diagonal blocks (= the diagonal blocks of W)
  double* xrow;   // 2 x (16 x PF_MLD): X_{sb,sb-1} in operand layout, double-buffered by the parity of sb
  double* ylds;   // right-hand side of the block
  double* zpart;  // 4 x 128 partial sums of z = W y; [0, 128) holds z itself when pf_block returns
  double* red;
  int* fail;
};
// ONE raw array (the launch-free kernel's tile role stages its operand ring in the same bytes: PF_LDS_BYTES >= 128 KB)
#define PF_LDS_BYTES ((128 * PF_LD + 8 * 16 * PF_MLD + 2 * 16 * PF_MLD + 128 + 4 * 128 + 16) * 8 + 16)
static __device__ __forceinline__ char* pf_lds_raw() {
  __shared__ __attribute__((aligned(1024))) char raw[PF_LDS_BYTES];
  return raw;
}
static __device__ __forceinline__ PfLds pf_lds() {
  double* const base = reinterpret_cast<double*>(pf_lds_raw());
  PfLds l;
  l.s = base;
  l.Minv = l.s + 128 * PF_LD;
  l.xrow = l.Minv + 8 * 16 * PF_MLD;
  l.ylds = l.xrow + 2 * 16 * PF_MLD;
  l.zpart = l.ylds + 128;
  l.red = l.zpart + 4 * 128;
  l.fail = reinterpret_cast<int*>(l.red + 16);
  return l;
}

// Operands of the chain's OWN share of block column J (pf_chain_next: the solve of block (J+1, J)), requested EARLY.  Measured
// (tools/persist_trace.py, n = 1024 x 32): of the 29 us the chain spends behind pf_block(J), 8-9 are the 128 KB of block (J+1, J)
// reaching ONE workgroup at ~15 GB/s with the solve's MFMAs waiting behind their A fragments (an L2 warm-up by the idle wave
// changed nothing: it is this CU's own miss queue, not where the lines sit).  The block is usually final well before pf_block(J)
// ends (its pre-update only needs the panels 0 .. J-1): wave 4 -- idle in pf_block -- looks at the two flags once per 16-pivot
// step, requested in one step and examined in the next, never waiting, and acquires when both are up; behind pf_block(J) -- its
// stores drained, its registers free -- every wave issues its A-fragment loads (16 x 16 B per lane) at once, under the release and
// the flag store that publish W_JJ, instead of behind pf_chain_next's own poll + acquire + barrier (2-4 us per column).  (Issued
// inside pf_block's tail they would be waited for by the drain of its stores.)  Same values in the same registers, only earlier.
struct PfPre {
  double af[32];  // rows 16 w + lr of block (J+1, J): 32 contiguous bytes per lane and k-group (pf_afrag_issue's image: NOT yet transposed)
  int state;      // 0: not requested, 2: loads issued (pf_chain_next skips its wait and its loads)
};
static __device__ __forceinline__ void pf_pre_issue(const PsArgs& a, int b, int J, int w, int lr, int lk, PfPre& pre) {
  const int I = J + 1, ld = a.ld;
  const double* const Ab = a.K + (size_t)b * a.mstride + (size_t)I * 128 * ld + (size_t)J * 128;
  pf_afrag_issue(Ab + (size_t)(16 * w + lr) * ld, lk, pre.af);
  pre.state = 2;
}

// One diagonal block of one walker (the whole workgroup).  Returns 0, or the 1-based pivot index inside the block at
// which the factorisation failed (status / lml of the walker are set here either way).  Called once per launch by
// potrf_kernel and once per block column by the persistent chain kernel (ps_chain_kernel): the LDS tile is free again
// when the function returns through its trailing barrier.
template <int GEN, int STAT, int FORM, int PRE>
static __device__ __forceinline__ int pf_block(int b, double* __restrict__ Kbuf, double* __restrict__ Wbuf,
                                               double* __restrict__ yw, double* __restrict__ accb,
                                               double* __restrict__ lml, int* __restrict__ status, int n, int ld,
                                               size_t mstride, int ystride, int nblk, int k, const PfGen& gen,
                                               bool inlds, unsigned* wrow, const PsArgs& pa, PfPre& pre) {
  // PRE (the single chain workgroup of the launch-free factorisation): request the operands of pf_chain_next(k) behind the
  // last step when the tile workers have handed them over by then (struct PfPre)
  // inlds (chain kernel, k > 0): the block and its right-hand side are in LDS already (pf_chain_next left them there)
  // wrow (chain pairs): the 16-row blocks of W go out to the partner workgroup AS THEY ARE FORMED -- row block p is complete
  // in memory at the end of step p + 1; the otherwise idle wave 4 releases it and raises *wrow to the number of complete
  // row blocks (1 .. 7; the eighth and z go out with the caller's wready flag) while the other waves run the next step
  const PfLds lds = pf_lds();
  double* const s = lds.s;
  double* const Minv = lds.Minv;
  double* const xrow0 = lds.xrow;
  double* const ylds = lds.ylds;
  double* const zpart = lds.zpart;
  double* const red = lds.red;
  int& fail_lds = *lds.fail;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));  // (inside the chain kernel's column loop: nothing derived from the lane id is hoisted out of it)
  const int tid = tid_, lane = tid & 63, w = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double* T = Kbuf + (size_t)b * mstride + (size_t)(k * 128) * ld + k * 128;
  double* yk = yw + (size_t)b * ystride + k * 128;
  PF_T(0);

  if (GEN) {
    if (tid == 0) status[b] = 0;
    pf_generate_tile<STAT, FORM>(s, Minv, gen, gen.H + (size_t)b * (gen.d + 2), n, tid);  // (Minv: scratch until step 0)
  } else if (!inlds) {
    // lower triangle of the tile -> LDS, all 16 16-byte loads of a thread in flight at once (the block is
    // latency-bound: one workgroup streams 64 KB).  Thread t owns column pair seg = t & 63 of rows
    // (t >> 6) + 8 i; pairs entirely above the diagonal are never read.
    const int seg = tid & 63, rbase = tid >> 6;
    d2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int row = rbase + 8 * u;
      v[u] = (2 * seg <= row + 15) ? *reinterpret_cast<const d2*>(T + (size_t)row * ld + seg * 2) : (d2){0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int row = rbase + 8 * u;
      *reinterpret_cast<d2*>(&s[row * PF_LD + seg * 2]) = v[u];
    }
  }
  if (tid < 128 && !inlds) ylds[tid] = GEN ? gen.y[tid] : yk[tid];
  double ld_prev = 0.0, zz_prev = 0.0;  // running log-det and z^T z of the earlier diagonal blocks
  if (tid == 0) {
    fail_lds = 0;
    if (PRE) lds.fail[1] = 0;  // 1: wave 4 has seen the pre-updated block (k+1, k) final (and acquired)
    if (k > 0) {
      ld_prev = accb[b * 4 + 0];
      zz_prev = accb[b * 4 + 1];
    }
  }
  __syncthreads();
  PF_T(1);

  double* Wg = GEN ? nullptr : Wbuf + ((size_t)b * nblk + k) * (128 * 128);
  int failed = 0;
  d4 xpend = (d4){0.0, 0.0, 0.0, 0.0};  // wave 0: X_{sb,sb-1}^T, written in place one step later (the update
                                        // waves still read the unscaled block T_{sb,sb-1} during this step)
  d4 w7 = (d4){0.0, 0.0, 0.0, 0.0};     // update waves: partial sum of one block of row 7 of W, finished after the loop
  // update waves: 1-3 and 5-7 (two per SIMD so that one's LDS latency hides behind the other's MFMAs); wave 4
  // shares the panel wave's SIMD and stays idle (every VALU / MFMA issue there would delay the pivot chain)
  const int u6 = (w < 4) ? w - 1 : w - 2;  // 0..5 for the update waves
  unsigned pre_fs = 0, pre_fd = 0;
  int pre_seen = (PRE && k + 1 < nblk) ? 0 : 1;
#pragma unroll 1
  for (int sb = 0; sb < 8; sb++) {
    PF_TW(2 * sb);
    if (PRE && w == 4 && lane == 0 && !pre_seen) {
      // (column 0 has no pre-update tasks: its two blocks are final as generated -- by the Gram kernel in front of this launch,
      // or, pa.gen, by two tile workers of this launch: genrdy)
      const bool gen0 = k == 0 && pa.gen;
      const unsigned want_s = gen0 ? 1u : (unsigned)pa.psplit, want_d = gen0 ? 1u : (unsigned)pa.dsplit;
      if ((k == 0 && !pa.gen) || (sb > 0 && pre_fs >= want_s && pre_fd >= want_d)) {
        if (k > 0 || pa.gen) ps_acquire();
        lds.fail[1] = 1;
        pre_seen = 1;
      } else if (gen0) {
        const unsigned* const g1 = pa.flags + PS_GEN(pa.B, nblk) + ((size_t)b * nblk + 1) * nblk;
        pre_fs = ps_ld(g1 + 0);  // genrdy[1][0]
        pre_fd = ps_ld(g1 + 1);  // genrdy[1][1]
      } else {
        pre_fd = ps_ld(pa.flags + PS_HDR + (size_t)pa.B * nblk + (size_t)b * nblk + (k + 1));                  // diagrdy[k + 1]
        pre_fs = ps_ld(pa.flags + PS_HDR + (size_t)pa.B * nblk * (2 + nblk) + (size_t)b * nblk + (k + 1));   // subrdy[k + 1]
      }
    }
    if (w == 4 && sb > 0) pf_check_diag(s, sb - 1, lane, lds.fail, PF_LD);  // (seen by everybody behind this step's barrier)
    if (w == 0) {
      if (sb > 1) {  // pending panel block of the previous row
#pragma unroll
        for (int r = 0; r < 4; r++) s[((sb - 1) * 16 + lr) * PF_LD + (sb - 2) * 16 + lk + 4 * r] = xpend[r];
      }
      d4 dg;
#pragma unroll
      for (int r = 0; r < 4; r++) dg[r] = s[(sb * 16 + lk + 4 * r) * PF_LD + sb * 16 + lr];
      if (sb > 0) {
        d4 xt = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double av = Minv[(sb - 1) * 16 * PF_MLD + lr * PF_MLD + kk * 4 + lk];
          const double bv = s[(sb * 16 + lr) * PF_LD + (sb - 1) * 16 + kk * 4 + lk];
          xt = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xt, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(-xt[kk], xt[kk], dg, 0, 0, 0);
        xpend = xt;
#pragma unroll
        for (int r = 0; r < 4; r++) xrow0[(sb & 1) * 16 * PF_MLD + lr * PF_MLD + lk + 4 * r] = xt[r];
      }
      // C layout -> one matrix row per lane through the block's own LDS slot (nobody else touches it)
#pragma unroll
      for (int r = 0; r < 4; r++) s[(sb * 16 + lk + 4 * r) * PF_LD + sb * 16 + lr] = dg[r];
      double a[16], mrow[16], macc[16];
#pragma unroll
      for (int c = 0; c < 16; c++) {
        a[c] = s[(sb * 16 + lr) * PF_LD + sb * 16 + c];
        macc[c] = 0.0;
      }
#ifdef PF_TRACE
      asm volatile("s_nop 0" ::"v"(a[0]), "v"(a[15]));
#endif
      PF_T(3 + sb * 3);
      micro_chol_inv<0>(a, macc, mrow, lr);
#ifdef PF_TRACE
      asm volatile("s_nop 0" ::"v"(a[15]), "v"(mrow[15]), "v"(mrow[14]));
#endif
      PF_T(4 + sb * 3);
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) Minv[sb * 16 * PF_MLD + j * PF_MLD + lr] = mrow[j];
#pragma unroll
        for (int c = 0; c < 16; c++) s[(sb * 16 + lr) * PF_LD + sb * 16 + c] = (c <= lr) ? a[c] : 0.0;
      }
    } else if (sb > 0 && w != 4) {
      const int p = sb - 1;  // phase: M_p was published at the previous barrier
      d4 xt = (d4){0.0, 0.0, 0.0, 0.0};  // X_{p+1,p}^T (every update wave forms its own copy)
      if (p + 2 < 8) {
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double av = Minv[p * 16 * PF_MLD + lr * PF_MLD + kk * 4 + lk];
          const double bv = s[((p + 1) * 16 + lr) * PF_LD + p * 16 + kk * 4 + lk];
          xt = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xt, 0, 0, 0);
        }
      }
      for (int I = p + 2 + u6; I < 8; I += 6) {  // at most one row per update wave
        d4 xi = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const double av = Minv[p * 16 * PF_MLD + lr * PF_MLD + kk * 4 + lk];
          const double bv = s[(I * 16 + lr) * PF_LD + p * 16 + kk * 4 + lk];
          xi = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, xi, 0, 0, 0);
        }
        // column p+1: terms p-1 (operands from LDS) and p (registers)
        d4 acc;
#pragma unroll
        for (int r = 0; r < 4; r++) acc[r] = s[(I * 16 + lk + 4 * r) * PF_LD + (p + 1) * 16 + lr];
        if (p > 0)
          acc = mma_run<1>(acc, &s[(I * 16 + lr) * PF_LD + (p - 1) * 16 + lk],
                           &s[((p + 1) * 16 + lr) * PF_LD + (p - 1) * 16 + lk], 1);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xi[kk], xt[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; r++) s[(I * 16 + lk + 4 * r) * PF_LD + (p + 1) * 16 + lr] = acc[r];
        // the row's own panel block (its unscaled values were last read just above)
#pragma unroll
        for (int r = 0; r < 4; r++) s[(I * 16 + lr) * PF_LD + p * 16 + lk + 4 * r] = xi[r];
        // column p+2: terms t <= p-1, and on the row's own diagonal block (I == p+2) also term p, so the
        // panel wave is left with a single rank-16 term
        if (p > 0 || I == p + 2) {
#pragma unroll
          for (int r = 0; r < 4; r++) acc[r] = s[(I * 16 + lk + 4 * r) * PF_LD + (p + 2) * 16 + lr];
          acc = mma_run<1>(acc, &s[(I * 16 + lr) * PF_LD + lk], &s[((p + 2) * 16 + lr) * PF_LD + lk], p);
          if (I == p + 2) {
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xi[kk], xi[kk], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; r++) s[(I * 16 + lk + 4 * r) * PF_LD + (p + 2) * 16 + lr] = acc[r];
        }
      }
      // row p of W = L^-1: M_p, its panel blocks (the last one via xrow) and rows < p of W are visible
      // (heavy blocks = small J go to the waves without a phase row: rows occupy update waves 0 .. 5-p)
      for (int J = 5 - u6; J <= p; J += 6) {
        d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
        if (J < p) acc = pf_wsum(acc, s, Minv, xrow0 + (p & 1) * 16 * PF_MLD, p, J, J, p, lane);
        pf_wfinish(acc, s, Minv, Wg, p, J, lane);
      }
      // row block sb-2 of L is final and visible: stream it out now (lower triangle, 16-byte pairs)
      if (!GEN && sb >= 2 && sb <= 6) {  // (the last step is the update waves' busiest: blocks 5..7 go out after the loop)
        const int R = sb - 2, ut = u6 * 64 + lane;  // 384 update threads: 24 per row
        const int rr = ut / 24, c0 = ut - 24 * rr, row = R * 16 + rr;
        for (int seg = c0; seg < 8 * R + 8; seg += 24) {
          if (2 * seg <= row)
            *reinterpret_cast<d2*>(T + (size_t)row * ld + seg * 2) = *reinterpret_cast<const d2*>(&s[row * PF_LD + seg * 2]);
        }
      }
      // last step: rows <= 5 of W are complete -> the terms K <= 5 of row 7 (one block per update wave)
      if (sb == 7) w7 = pf_wsum(w7, s, Minv, nullptr, 7, u6, u6, 6, lane);  // block J = u6 (W row 6 went 5 - u6)
      if (wrow) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this step's W blocks have left the wave
    }
    PF_TW(2 * sb + 1);
    __syncthreads();
    PF_T(2 + sb * 3);
    failed = fail_lds;
    if (failed) break;  // uniform across the workgroup
    if (wrow && w == 4 && lane == 0 && sb >= 1) {  // row blocks 0 .. sb-1 of W are complete
      ps_release();
      ps_st(wrow, (unsigned)sb);
    }
  }
  if (!failed && w == 4) pf_check_diag(s, 7, lane, lds.fail, PF_LD);  // the last sub-block: read behind the next barrier
  if (failed) {
    if (tid == 0) {
      status[b] = k * 128 + failed;  // 1-based index of the failing pivot
      lml[b] = -INFINITY;            // sklearn/_gpr.py:588-589
    }
    return failed;
  }
  PF_T(26);
  // ---- L_kk out.  Only the lower triangle is written (16-byte stores; the element right of the diagonal in
  // a straddling pair is junk nobody reads: every consumer of this tile masks j <= i).  Block (7, 6) is still
  // in the panel wave's registers and goes out from there.  The stores drain while row 7 of W is formed.
  if (!GEN) {
    const int seg = tid & 63, rbase = tid >> 6;
#pragma unroll
    for (int i = 10; i < 16; i++) {  // row blocks 5, 6 and 7 (0..4 went out inside the loop)
      const int row = rbase + 8 * i;
      if (2 * seg <= row && !(row >= 112 && seg >= 48 && seg < 56))
        *reinterpret_cast<d2*>(T + (size_t)row * ld + seg * 2) = *reinterpret_cast<const d2*>(&s[row * PF_LD + seg * 2]);
    }
  }
  // ---- row 7 of W: last term (K = 6) and the multiplication by -M_7
  if (w == 0) {
    if (!GEN) {
#pragma unroll
      for (int r = 0; r < 4; r++) T[(size_t)(7 * 16 + lr) * ld + 6 * 16 + lk + 4 * r] = xpend[r];
    }
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
    acc = pf_wsum(acc, s, Minv, xrow0 + 16 * PF_MLD, 7, 6, 6, 7, lane);
    pf_wfinish(acc, s, Minv, Wg, 7, 6, lane);
    pf_wfinish(acc, s, Minv, Wg, 7, 7, lane);
  } else if (w != 4) {
    w7 = pf_wsum(w7, s, Minv, xrow0 + 16 * PF_MLD, 7, u6, 6, 7, lane);
    pf_wfinish(w7, s, Minv, Wg, 7, u6, lane);
  }
  double ldv = (tid < 128) ? log(s[tid * PF_LD + tid]) : 0.0;
  for (int o = 32; o > 0; o >>= 1) ldv += __shfl_xor(ldv, o);
  if (lane == 0) red[w] = ldv;
  __syncthreads();
  failed = fail_lds;  // (a failure inside the last 16 pivots)
  if (failed) {
    if (tid == 0) {
      status[b] = k * 128 + failed;
      lml[b] = -INFINITY;
    }
    return failed;
  }
  PF_T(27);
  // ---- z = W y from the LDS copy of W (transposed in the upper triangle, diagonal blocks in Minv); four
  // threads per row, fixed summation order (bitwise reproducible)
  {
    const int row = tid & 127, h = tid >> 7, Ib = row >> 4, ri = row & 15;  // h = 0..3
    double z0 = 0.0, z1 = 0.0;  // two chains (16 Ib is a multiple of 8), fixed order
    for (int j = h; j < 16 * Ib; j += 8) {
      z0 = fma(s[j * PF_LD + row], ylds[j], z0);
      z1 = fma(s[(j + 4) * PF_LD + row], ylds[j + 4], z1);
    }
    double zs = z0 + z1;
    for (int jj = h; jj <= ri; jj += 4) zs = fma(Minv[Ib * 16 * PF_MLD + ri * PF_MLD + jj], ylds[16 * Ib + jj], zs);
    zpart[h * 128 + row] = zs;
  }
  __syncthreads();
  double zv = 0.0;
  if (tid < 128) {
    zv = (zpart[tid] + zpart[128 + tid]) + (zpart[256 + tid] + zpart[384 + tid]);
    if (!GEN) yk[tid] = zv;
    zpart[tid] = zv;  // (a thread reads and writes its own column only: z stays in LDS for pf_chain_next)
  }
  double zz = zv * zv;
  for (int o = 32; o > 0; o >>= 1) zz += __shfl_xor(zz, o);
  if (lane == 0) red[8 + w] = zz;
  __syncthreads();
  if (tid == 0) {
    double ldt = red[0] + red[1];  // threads 0..127 (waves 0 and 1) hold the diagonal and z
    double zzt = red[8] + red[9];
    ldt += ld_prev;
    zzt += zz_prev;
    if (!GEN) {
      accb[b * 4 + 0] = ldt;
      accb[b * 4 + 1] = zzt;
    }
    if (k == nblk - 1) {
      // (no implicit fused multiply-add: this function is inlined into two kernels and the backend decides contraction
      // per call site -- the launch-free and the multi-launch path rounded this line differently, 0.5 ulp of n log 2 pi)
#pragma clang fp contract(off)
      double v = -0.5 * zzt - ldt - 0.5 * (double)n * 1.8378770664093453;  // log(2 pi)
      if (!(v > -INFINITY && v < INFINITY)) {  // overflow somewhere on the way: report like a failed factorisation
        v = -INFINITY;
        status[b] = n + 1;
      }
      lml[b] = v;
    }
  }
  PF_T(28);
  __syncthreads();  // (every wave is past its last LDS read: a caller may run the next block in this workgroup)
  return 0;
}

// dt[u] -= X_ti X_tj^T for this wave's NT tiles of the next diagonal block, X row-major in the LDS tile: the operands of
// k-step kk+1 are read while k-step kk multiplies (two waves per SIMD cover the rest of the LDS latency).
template <int NT>
static __device__ __forceinline__ void pf_diag_update(d4 (&dt)[5], const double* __restrict__ s, const int (&offa)[5],
                                                      const int (&offb)[5]) {
  double a0[NT], b0[NT];
#pragma unroll
  for (int u = 0; u < NT; u++) {
    a0[u] = s[offa[u]];
    b0[u] = s[offb[u]];
  }
#pragma unroll 2
  for (int t = 0; t < 32; t++) {
    double a1[NT], b1[NT];
    const int tn = (t + 1 < 32) ? t + 1 : t;
#pragma unroll
    for (int u = 0; u < NT; u++) {
      a1[u] = s[offa[u] + 4 * tn];
      b1[u] = s[offb[u] + 4 * tn];
    }
#pragma unroll
    for (int u = 0; u < NT; u++) dt[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], dt[u], 0, 0, 1);
#pragma unroll
    for (int u = 0; u < NT; u++) {
      a0[u] = a1[u];
      b0[u] = b1[u];
    }
  }
}

// The chain's own share of block column J (PsArgs::fat): the two blocks the next factorisation waits for never leave the
// chain's workgroup.  W_JJ is still in LDS (transposed in the upper triangle of the tile, diagonal blocks in Minv), z_J in
// zpart; the tile workers have applied the panels 0 .. J-1 to blocks (J+1, J) and (J+1, J+1) beforehand (subrdy / diagrdy).
//   1. X = A_{J+1,J} W_JJ^T: wave w owns rows 16 w .. 16 w + 15, the A operand comes from global memory straight into
//      fragment registers (32 doubles per lane), chunk c of k only reaches the column blocks j >= c; right-hand side
//      y_{J+1} -= X z_J in the same pass (one row per 16 lanes, fixed shuffle order): the arithmetic of trsm4_kernel;
//      X goes to global memory for the tile workers (xready[J+1][J]) and, row-major, into the LDS tile;
//   2. D_{J+1,J+1} -= X X^T on the 36 lower 16 x 16 tiles (accumulator = the block as the tile worker left it, k ascending
//      in steps of 4, A-negate: the arithmetic of syrk4_kernel), and the result IS the next LDS tile of pf_block: no
//      flag, no L2 round trip and no other workgroup between two factorisations.
// Returns 0, or -1 when a wait was abandoned.
static __device__ __forceinline__ int pf_chain_next(const PsArgs& a, int b, int J, int* ok_lds, unsigned long long* tr, PfPre& pre) {
  const PfLds lds = pf_lds();
  double* const s = lds.s;
  const double* const Minv = lds.Minv;
  double* const ylds = lds.ylds;
  const double* const zl = lds.zpart;
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int I = J + 1, ld = a.ld, nblk = a.nblk;
  unsigned* const flags = a.flags;
  unsigned* const err = flags + PS_ERROR;
  double* const Mb = a.K + (size_t)b * a.mstride;
  double* const Ab = Mb + (size_t)I * 128 * ld + (size_t)J * 128;
  const double* const Db = Mb + (size_t)I * 128 * ld + (size_t)I * 128;
  const bool early = pre.state == 2;  // (uniform) requested behind pf_block(J)'s last step: flags seen, acquired, loads issued
  if ((J > 0 || a.gen) && !early) {
    if (tid == 0) {
      const unsigned* const diagrdy = flags + PS_HDR + (size_t)a.B * nblk + (size_t)b * nblk;
      const unsigned* const subrdy = flags + PS_HDR + (size_t)a.B * nblk * (2 + nblk) + (size_t)b * nblk;
      const unsigned* const g1 = flags + PS_GEN(a.B, nblk) + ((size_t)b * nblk + 1) * nblk;  // genrdy[1][.]
      const bool ok = J > 0 ? ps_wait_ge2(subrdy + I, (unsigned)a.psplit, diagrdy + I, (unsigned)a.dsplit, err, a.spin_limit)
                            : ps_wait_ge2(g1 + 0, 1u, g1 + 1, 1u, err, a.spin_limit);
      ps_acquire();
      *ok_lds = ok ? 1 : 0;
    }
    __syncthreads();
    if (!*ok_lds) return -1;
  }
  if (tr && tid == 0) tr[J * 8 + 2] = wall_clock64();
  // ---- operands: A fragments (k = 4 t + lk of row 16 w + lr), this lane's rows of y, the D tiles of this wave
  if (!early) pf_pre_issue(a, b, J, w, lr, lk, pre);
  pre.state = 0;
  double (&af)[32] = pre.af;
  double yv[4];
  {
    const double* const yi = a.yw + (size_t)b * a.ystride + I * 128;
#pragma unroll
    for (int r = 0; r < 4; r++) yv[r] = yi[16 * w + lk + 4 * r];
  }
  pf_afrag_transpose(af);
  // lower 16 x 16 tiles t = w, w + 8, ... < 36 in row-major order of the triangle (waves 0-3: five, 4-7: four; nine per SIMD)
  d4 dt[5];
  int offa[5], offb[5], offc[5], offd[5];
#pragma unroll
  for (int u = 0; u < 5; u++) {
    const int t = w + 8 * u;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    offa[u] = (ti * 16 + lr) * PF_LD + lk;
    offb[u] = (tj * 16 + lr) * PF_LD + lk;
    offc[u] = (ti * 16 + lk) * PF_LD + tj * 16 + lr;
    offd[u] = (ti * 16 + lk) * ld + tj * 16 + lr;
  }
  // ---- 1. panel solve
  d4 x[8];
#pragma unroll
  for (int j = 0; j < 8; j++) x[j] = (d4){0.0, 0.0, 0.0, 0.0};
  // (the W operands of k-step t+1 are read while k-step t multiplies: with the reads placed just in front of their MFMAs the
  // two waves of a SIMD stalled on LDS together -- the second one finished its 144 MFMAs 5.6 us after the first)
  {
    double bq[2][8];
#pragma unroll
    for (int j = 1; j < 8; j++) bq[0][j] = s[lk * PF_LD + j * 16 + lr];
    bq[0][0] = Minv[lr * PF_MLD + lk];
#pragma unroll
    for (int t = 0; t < 32; t++) {
      const int c = t >> 2, cur = t & 1, nxt = cur ^ 1;
      if (t + 1 < 32) {
        const int cn = (t + 1) >> 2, kn = (t + 1) & 3;
        // column block cn: the diagonal block of W (its upper part holds zeros); j > cn: W[j][cn]^T in the upper triangle
        bq[nxt][cn] = Minv[cn * 16 * PF_MLD + lr * PF_MLD + 4 * kn + lk];
#pragma unroll
        for (int j = cn + 1; j < 8; j++) bq[nxt][j] = s[(cn * 16 + 4 * kn + lk) * PF_LD + j * 16 + lr];
      }
      const double av = af[t];
#pragma unroll
      for (int j = c; j < 8; j++) x[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bq[cur][j], x[j], 0, 0, 0);
    }
  }
  // (the D tiles are fetched under the epilogue: with them in flight during the solve the kernel spilled)
#pragma unroll
  for (int u = 0; u < 5; u++) {
    if (w + 8 * u < 36) {
#pragma unroll
      for (int r = 0; r < 4; r++) dt[u][r] = Db[(size_t)offd[u] + (size_t)(4 * r) * ld];
    } else {
      dt[u] = (d4){0.0, 0.0, 0.0, 0.0};
    }
  }
  if (tr && tid == 0) tr[J * 8 + 4] = wall_clock64();
  {
    double zc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) zc[j] = zl[16 * j + lr];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * w + lk + 4 * r;
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const double xv = x[j][r];
        Ab[(size_t)row * ld + 16 * j + lr] = xv;
        part = __builtin_fma(xv, zc[j], part);  // (trsm4_kernel's `part += x * zc[j]` is contracted the same way)
      }
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      if (lr == 0) ylds[row] = yv[r] - part;
    }
  }
  __syncthreads();  // nobody reads W in the tile any more (X's stores drain while it goes into the tile)
  if (tr && tid == 0) tr[J * 8 + 5] = wall_clock64();
#pragma unroll
  for (int j = 0; j < 8; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) s[(16 * w + lk + 4 * r) * PF_LD + 16 * j + lr] = x[j][r];
  ps_publish_barrier();  // X is in the tile, and every wave's part of it has reached memory
  if (tid == 0) ps_signal_add(flags + PS_HDR + (size_t)2 * a.B * nblk + ((size_t)b * nblk + I) * nblk + J);  // xready[I][J]
  if (tr && tid == 0) tr[J * 8 + 6] = wall_clock64();
  // ---- 2. the next diagonal block
  if (w < 4)
    pf_diag_update<5>(dt, s, offa, offb);
  else
    pf_diag_update<4>(dt, s, offa, offb);
  __syncthreads();  // everybody is done with X in the tile
  if (tr && tid == 0) tr[J * 8 + 7] = wall_clock64();
  if (tr && tid == 0) tr[J * 8 + 7] = wall_clock64();
#pragma unroll
  for (int u = 0; u < 5; u++) {
    if (w + 8 * u < 36) {
#pragma unroll
      for (int r = 0; r < 4; r++) s[offc[u] + 4 * r * PF_LD] = dt[u][r];
    }
  }
  __syncthreads();
  if (tr && tid == 0) tr[J * 8 + 3] = wall_clock64();
  return 0;
}

// ---- chain PAIRS (PsArgs::pair) ----------------------------------------------------------------------------------------
// With ONE chain workgroup per matrix a block column costs 55 us: 25 us of pf_block and then, one after the other on the
// same CU, the solve of block (J+1, J) and the last update of block (J+1, J+1) (pf_chain_next: 29 us, 17 of them MFMA issue).
// Nothing in those two steps needs the WHOLE of W_JJ at once: column block sb of X = A W^T only reads row block sb of W, which
// pf_block completes at the end of its step sb + 1, and the rank-16 term of the diagonal update with that column block of X
// can follow at once.  So a SECOND workgroup -- the one that will factorise column J + 1 -- does both under the first one's
// pf_block(J), row block by row block as pf_block publishes them (wrow), accumulates D_{J+1,J+1} in registers and drops it
// into ITS OWN LDS tile: when pf_block(J) ends only the last row block's share is left (a few us), and the tile is where the
// next factorisation needs it.  The two workgroups swap roles every column.  Per element the operations and their order are
// those of pf_chain_next (accumulators start from the block as the tile workers left it, k ascending): the same bits.
// Staging in the (free) tile region: W's lower row blocks PACKED -- row block q is 16 rows of 16 (q + 1) + 2 doubles (== 2 mod
// 16: conflict-free B-operand reads) at PH_WOFF(q) -- and two 128 x 16 column blocks of X.
#define PH_WLDQ(q) (16 * ((q) + 1) + 2)
#define PH_WOFF(q) (128 * (q) * ((q) + 1) + 32 * (q))
#define PH_XLD 18   // leading dimension of a staged 128 x 16 column block of X
static_assert(PH_WOFF(8) + 2 * 128 * PH_XLD <= 128 * PF_LD, "the helper's staging buffers live in the (free) tile region");

// barrier for LDS traffic only: global loads / stores of the wave stay in flight across it
static __device__ __forceinline__ void pf_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// `have` = row blocks of W_JJ known to be in memory (and acquired), `staged` = row blocks already in LDS.  Per step: wave 7 looks
// (without waiting) for row blocks published since; solve (row block SB from LDS) -> X out and into LDS -> barrier -> loads of row
// block SB + 1 issued if it is out -> rank-16 update -> row block SB + 1 into LDS -> barrier.  A row block that is not there when
// its step begins is waited for, acquired and staged in the open.
template <int SB, int NT>
static __device__ __forceinline__ int pf_pair_steps(const PsArgs& a, int b, int J, int* ok_lds, int have, int staged,
                                                    const double (&af)[32], double* __restrict__ Xg, int ld, d4 (&dt)[5],
                                                    const int (&tio)[5], const int (&tjo)[5], double* __restrict__ Wl,
                                                    double* __restrict__ Xl, int tid, int w, int lane, unsigned long long* tr,
                                                    int* peek_lds, unsigned* xcol) {
  if constexpr (SB < 8) {
    const int lr = lane & 15, lk = lane >> 4;
    unsigned* const flags = a.flags;
    if (NT > 0 && SB == 7 && tr && tid == 0) tr[J * 8 + 5] = wall_clock64();  // everything but the last row block is done
    if (have <= SB) {  // row block SB not known to be out yet: wait for it (one lane), then ONE acquire for everything published
      if (tid == 0) {
        unsigned* const wready = flags + PS_HDR + (size_t)b * a.nblk;
        unsigned* const wrow = flags + PS_HDR + (size_t)a.B * a.nblk * (3 + a.nblk) + (size_t)b * a.nblk;
        unsigned* const err = flags + PS_ERROR;
        const bool ok = (SB < 7) ? ps_wait_ge(wrow + J, (unsigned)(SB + 1), err, a.spin_limit)
                                 : ps_wait_ge(wready + J, 1u, err, a.spin_limit);
        int cnt = (SB < 7) ? (int)ps_ld(wrow + J) : 8;
        if (cnt > 7) cnt = (ps_ld(wready + J) >= 1u) ? 8 : 7;
        ps_acquire();
        if (__hip_atomic_load(a.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) cnt = -2;  // the matrix has failed
        *ok_lds = ok ? cnt : -1;
      }
      __syncthreads();
      have = *ok_lds;
      __syncthreads();
      if (have < 0) return have;
    }
    if (NT > 0 && SB == 7 && tr && tid == 0) tr[J * 8 + 7] = wall_clock64();  // the last row block and z are out
    // Wave 7 (four tiles instead of five in the update) LOOKS for row blocks that have come out since, without waiting, and
    // acquires them: the other waves learn the count behind this step's barrier and issue the loads of the block after next
    // under the update -- a helper that has caught up with pf_block otherwise pays a blocking poll, an acquire and an exposed
    // load per step (6-7 us against the 3.1 us of a pf_block step: it fell 14 us behind over the last five row blocks).
    if (SB < 7 && w == 7 && lane == 0) {
      int cnt = have;
      if (have < 8) {
        const unsigned* const wrow = flags + PS_HDR + (size_t)a.B * a.nblk * (3 + a.nblk) + (size_t)b * a.nblk;
        cnt = (int)ps_ld(wrow + J);
        if (cnt >= 7) cnt = (ps_ld(flags + PS_HDR + (size_t)b * a.nblk + J) >= 1u) ? 8 : 7;  // (the eighth goes out with wready)
        if (cnt > have) ps_acquire();
        else cnt = have;
      }
      *peek_lds = cnt;
    }
    // (the step's global addresses are formed HERE, from an opaque copy of the step number: formed at the top of the helper and
    // kept in scalar registers for all eight steps they made the kernel spill scalars to scratch)
    int sbo = SB;
    asm volatile("" : "+s"(sbo));
    const double* const Wg = a.W + ((size_t)b * a.nblk + J) * (128 * 128) + (size_t)(16 * 128) * (sbo - SB);
    const int sr = tid >> 5, sc0 = tid & 31;  // staging: thread -> (row, first column) of a 16-row block
    if (staged <= SB) {  // row block SB of W (16 rows x 16 (SB + 1) columns, lower blocks only) -> LDS, latency in the open
      const double* const src = Wg + (size_t)(16 * SB + sr) * 128;
      double* const dst = Wl + PH_WOFF(SB) + sr * PH_WLDQ(SB);
      double v[(16 * (SB + 1) + 31) / 32];
#pragma unroll
      for (int i = 0; i < (16 * (SB + 1) + 31) / 32; i++) v[i] = (sc0 + 32 * i < 16 * (SB + 1)) ? src[sc0 + 32 * i] : 0.0;
#pragma unroll
      for (int i = 0; i < (16 * (SB + 1) + 31) / 32; i++)
        if (sc0 + 32 * i < 16 * (SB + 1)) dst[sc0 + 32 * i] = v[i];
      staged = SB + 1;
      __syncthreads();
    }
    // column block SB of X = A W^T for this wave's 16 rows: k ascending, as pf_chain_next
    // (operand reads two 16-wide chunks deep, fenced: left to itself the compiler hoists every LDS read of the step above the
    // first MFMA -- 64 more live registers, and the kernel spilled)
    d4 xs = (d4){0.0, 0.0, 0.0, 0.0};
    {
      const double* const wq = Wl + PH_WOFF(SB) + lr * PH_WLDQ(SB) + lk;
      double bq[2][4];
#pragma unroll
      for (int kk = 0; kk < 4; kk++) bq[0][kk] = wq[4 * kk];
#pragma unroll
      for (int g = 0; g <= SB; g++) {
        if (g < SB) {
#pragma unroll
          for (int kk = 0; kk < 4; kk++) bq[(g + 1) & 1][kk] = wq[16 * (g + 1) + 4 * kk];
        }
#pragma unroll
        for (int kk = 0; kk < 4; kk++) xs = __builtin_amdgcn_mfma_f64_16x16x4f64(af[4 * g + kk], bq[g & 1][kk], xs, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ... out at once, in place (this wave's rows of A are in its registers), and into LDS for the update below.  (The 64
    // registers a resident X would take made the kernel spill: the right-hand side re-reads these stores at the end.)
    double* const Xs = Xl + (SB & 1) * (128 * PH_XLD);
    // (write-through stores, a whole 128-B line per row: the column block is handed on -- xcol -- as soon as every wave's stores
    // have completed, one step later, with no L2 write-back: the pre-update that consumes this block follows column block by
    // column block instead of starting when the whole block and its right-hand side are out)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      ps_st_wt(Xg + (unsigned)((16 * w + lk + 4 * r) * ld + 16 * SB + lr), xs[r]);
      if constexpr (NT > 0) Xs[(16 * w + lk + 4 * r) * PH_XLD + lr] = xs[r];
    }
    // The chain helper's steps are 3 us apart: the previous step's stores have long completed (all but this step's four: vmcnt(4)).
    // The streamed solve of a tile task (NT == 0) steps every microsecond, a write-through store takes two or three: it hands
    // on what completed THREE steps ago (vmcnt(12): nothing else of this wave is in flight here -- a prefetched row block of W was
    // consumed at the end of its step) and the last three blocks behind the caller's drain.
    constexpr int LAG = NT > 0 ? 1 : 3;
    if (SB >= LAG && xcol) {
      if (LAG == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    pf_lds_barrier();  // (LDS only: the A fragments' and D tiles' loads and X's stores stay in flight)
    if (SB >= LAG && xcol && w == 7 && lane == 0) ps_st(xcol, (unsigned)(SB - LAG + 1));
    if (SB < 7) have = *peek_lds;  // (>= the old value; a failed matrix is caught behind the steps)
    // the NEXT row block, if it is out: its loads fly under the update
    constexpr int NNX = (SB < 7) ? (16 * (SB + 2) + 31) / 32 : 1;
    double vnx[NNX];
    const bool pre = SB < 7 && have > SB + 1 && staged == SB + 1;
    if (pre) {
      const double* const src = Wg + (size_t)(16 * (SB + 1) + sr) * 128;
#pragma unroll
      for (int i = 0; i < NNX; i++) vnx[i] = (sc0 + 32 * i < 16 * (SB + 2)) ? src[sc0 + 32 * i] : 0.0;
    }
    // (a tile task's streamed solve has no update to fly them under and catches up with pf_block from behind: TWO row blocks at a
    // time while two are out, or the remote round trip is paid once per row block)
    constexpr int NNY = (NT == 0 && SB < 6) ? (16 * (SB + 3) + 31) / 32 : 1;
    double vny[NNY];
    const bool pre2 = NT == 0 && SB < 6 && pre && have > SB + 2;
    if (pre2) {
      const double* const src = Wg + (size_t)(16 * (SB + 2) + sr) * 128;
#pragma unroll
      for (int i = 0; i < NNY; i++) vny[i] = (sc0 + 32 * i < 16 * (SB + 3)) ? src[sc0 + 32 * i] : 0.0;
    }
    // rank-16 term of the next diagonal block: dt[u] -= X_ti X_tj^T over k = 16 SB .. 16 SB + 15   (NT == 0: the streamed
    // panel solve of a tile task -- no diagonal block)
    if constexpr (NT > 0) {
      double ua[2][NT], ub[2][NT];
      const double* const xl = Xs + lr * PH_XLD + lk;  // (tio / tjo: wave-uniform row offsets of the tile's two row blocks)
#pragma unroll
      for (int u = 0; u < NT; u++) {
        ua[0][u] = xl[tio[u]];
        ub[0][u] = xl[tjo[u]];
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        if (kk < 3) {
#pragma unroll
          for (int u = 0; u < NT; u++) {
            ua[(kk + 1) & 1][u] = xl[tio[u] + 4 * (kk + 1)];
            ub[(kk + 1) & 1][u] = xl[tjo[u] + 4 * (kk + 1)];
          }
        }
#pragma unroll
        for (int u = 0; u < NT; u++) dt[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[kk & 1][u], ub[kk & 1][u], dt[u], 0, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (pre) {
      double* const dst = Wl + PH_WOFF(SB + 1) + sr * PH_WLDQ(SB + 1);
#pragma unroll
      for (int i = 0; i < NNX; i++)
        if (sc0 + 32 * i < 16 * (SB + 2)) dst[sc0 + 32 * i] = vnx[i];
      staged = SB + 2;
      if (pre2) {
        double* const dst2 = Wl + PH_WOFF(SB + 2) + sr * PH_WLDQ(SB + 2);
#pragma unroll
        for (int i = 0; i < NNY; i++)
          if (sc0 + 32 * i < 16 * (SB + 3)) dst2[sc0 + 32 * i] = vny[i];
        staged = SB + 3;
      }
      pf_lds_barrier();
    }
    return pf_pair_steps<SB + 1, NT>(a, b, J, ok_lds, have, staged, af, Xg, ld, dt, tio, tjo, Wl, Xl, tid, w, lane, tr, peek_lds, xcol);
  } else {
    return have;
  }
}

// Returns 0 (block (J+1, J+1) and its right-hand side are in this workgroup's LDS), -1 when a wait was abandoned, -2 when the
// matrix has failed.
static __device__ __forceinline__ int pf_pair_helper(const PsArgs& a, int b, int J, int* ok_lds, int* peek_lds, unsigned long long* tr) {
  const PfLds lds = pf_lds();
  double* const s = lds.s;
  double* const ylds = lds.ylds;
  double* const Wl = s;
  double* const Xl = s + PH_WOFF(8);
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int I = J + 1, ld = a.ld, nblk = a.nblk;
  unsigned* const flags = a.flags;
  unsigned* const err = flags + PS_ERROR;
  double* const Mb = a.K + (size_t)b * a.mstride;
  double* const Ab = Mb + (size_t)I * 128 * ld + (size_t)J * 128;
  const double* const Db = Mb + (size_t)I * 128 * ld + (size_t)I * 128;
  // ONE wait and ONE acquire for everything the helper starts from: the tile workers' pre-updates of its two blocks (panels
  // 0 .. J-1) and the row blocks of W_JJ that are out already (it used to poll and acquire again inside step 0, behind the A
  // fragments' loads: 11 us from "ready" to the end of step 0 by the in-kernel stamps)
  int have0 = 0;
  {
    if (tid == 0) {
      // (every word of a poll is requested before the first is looked at: five dependent round trips to the flags -- two waits, the
      // row count, wready, the status -- were 4 us between the last pre-update's signal and the helper's start)
      const unsigned* const diagrdy = flags + PS_HDR + (size_t)a.B * nblk + (size_t)b * nblk + I;
      const unsigned* const subrdy = flags + PS_HDR + (size_t)a.B * nblk * (2 + nblk) + (size_t)b * nblk + I;
      const unsigned* const wrow = flags + PS_HDR + (size_t)a.B * nblk * (3 + nblk) + (size_t)b * nblk + J;
      const unsigned* const wrdy = flags + PS_HDR + (size_t)b * nblk + J;
      const unsigned need_s = J > 0 ? (unsigned)a.psplit : 0u, need_d = J > 0 ? (unsigned)a.dsplit : 0u;
      bool ok = true;
      int cnt = 0;
      const unsigned long long t0 = wall_clock64();
      for (unsigned it = 0;; it++) {
        const unsigned fs = ps_ld(subrdy), fd = ps_ld(diagrdy), fr = ps_ld(wrow), fw = ps_ld(wrdy);
        const int st = __hip_atomic_load(a.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        cnt = st != 0 ? -2 : (fr >= 7u ? (fw >= 1u ? 8 : 7) : (int)fr);  // (-2: the matrix has failed)
        if (fs >= need_s && fd >= need_d) break;
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
      *ok_lds = ok ? cnt : -1;
    }
    __syncthreads();
    have0 = *ok_lds;
    __syncthreads();
    if (have0 < 0) return have0;
  }
  if (tr && tid == 0) tr[J * 8 + 2] = wall_clock64();
  // (the D tiles first: the first rank-16 update needs all of them, the solves need the A fragments four at a time)
  d4 dt[5];
  int tio[5], tjo[5], tco[5];  // wave-uniform: row offsets of the tile's row blocks in a staged X column block; tile origin in the LDS tile
#pragma unroll
  for (int u = 0; u < 5; u++) {
    const int t = w + 8 * u;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= t) ti++;
    const int tj = t - ti * (ti + 1) / 2;
    tio[u] = ti * 16 * PH_XLD;
    tjo[u] = tj * 16 * PH_XLD;
    tco[u] = ti * 16 * PF_LD + tj * 16;
    if (t < 36) {
      const double* const dp = Db + (unsigned)((ti * 16 + lk) * ld + tj * 16 + lr);
#pragma unroll
      for (int r = 0; r < 4; r++) dt[u][r] = dp[(unsigned)(4 * r * ld)];
    } else {
      dt[u] = (d4){0.0, 0.0, 0.0, 0.0};
    }
  }
  // row blocks 0 and 1 of W (if out) first, then the A fragments: all in flight together; the staging stores only wait for their
  // own loads (loads return in order) and the barrier behind them leaves the A fragments in flight
  double af[32];
  int staged0 = 0;
  {
    const double* const Wg = a.W + ((size_t)b * nblk + J) * (128 * 128);
    const int sr = tid >> 5, sc0 = tid & 31;
    double w0 = 0.0, w1 = 0.0;
    if (have0 >= 1 && sc0 < 16) w0 = Wg[(size_t)sr * 128 + sc0];
    if (have0 >= 2) w1 = Wg[(size_t)(16 + sr) * 128 + sc0];
    pf_load_afrag(Ab + (size_t)(16 * w + lr) * ld, lk, af);
    if (have0 >= 1 && sc0 < 16) Wl[PH_WOFF(0) + sr * PH_WLDQ(0) + sc0] = w0;
    if (have0 >= 2) Wl[PH_WOFF(1) + sr * PH_WLDQ(1) + sc0] = w1;
    staged0 = have0 >= 2 ? 2 : (have0 >= 1 ? 1 : 0);
    pf_lds_barrier();
  }
  unsigned* const xcol = flags + PS_XCOL(a.B, nblk) + ((size_t)b * nblk + I) * 3;  // column blocks of X_{J+1,J} in memory
  const int have = (w < 4) ? pf_pair_steps<0, 5>(a, b, J, ok_lds, have0, staged0, af, Ab, ld, dt, tio, tjo, Wl, Xl, tid, w, lane, tr, peek_lds, xcol)
                           : pf_pair_steps<0, 4>(a, b, J, ok_lds, have0, staged0, af, Ab, ld, dt, tio, tjo, Wl, Xl, tid, w, lane, tr, peek_lds, xcol);
  if (have < 0) return have;
  if (tr && tid == 0) tr[J * 8 + 4] = wall_clock64();
  // (a factorisation that failed while the helper followed it releases every flag: look before the next block is taken over)
  if (tid == 0) *ok_lds = __hip_atomic_load(a.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? -2 : 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // X's last column block has left this wave (write-through: it is in memory)
  __syncthreads();
  const int gone = *ok_lds;
  if (tid == 0 && gone == 0) ps_st(xcol, 8u);  // ... and everybody's: the pre-update of block (J+2, J+1) takes its last chunk
  __syncthreads();
  if (gone < 0) return gone;
  // ---- z_J is out (the last wait was for wready[J]): the right-hand side, as pf_chain_next -- X read back from this wave's
  // own stores (drained above)
  {
    const double* const zg = a.yw + (size_t)b * a.ystride + J * 128;
    double zc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) zc[j] = zg[16 * j + lr];
    double yv[4];
    {
      const double* const yi = a.yw + (size_t)b * a.ystride + I * 128;
#pragma unroll
      for (int r = 0; r < 4; r++) yv[r] = yi[16 * w + lk + 4 * r];
    }
    double xv[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < 8; j++) xv[r][j] = Ab[(unsigned)((16 * w + lk + 4 * r) * ld + 16 * j + lr)];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * w + lk + 4 * r;
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) part = __builtin_fma(xv[r][j], zc[j], part);
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      if (lr == 0) ylds[row] = yv[r] - part;
    }
  }
  ps_publish_barrier();  // X has reached memory; nobody reads the staging buffers any more
  if (tid == 0) ps_signal_add(flags + PS_HDR + (size_t)2 * a.B * nblk + ((size_t)b * nblk + I) * nblk + J);  // xready[I][J]
  if (tr && tid == 0) tr[J * 8 + 6] = wall_clock64();
#pragma unroll
  for (int u = 0; u < 5; u++) {
    if (w + 8 * u < 36) {
#pragma unroll
      for (int r = 0; r < 4; r++) s[tco[u] + (lk + 4 * r) * PF_LD + lr] = dt[u][r];
    }
  }
  __syncthreads();
  if (tr && tid == 0) tr[J * 8 + 3] = wall_clock64();
  return 0;
}

// The same streamed solve for a TILE task (chain pairs only): S(J+2, J), the block both pre-updates of column J+2 wait for, does
// not wait for the whole of W_JJ either -- its workgroup has applied the panels 0 .. J-1 and stored the block, re-reads its own
// rows as A fragments and follows pf_block(J) row block by row block; X_{J+2,J} is out a few us after W_JJ's last row block
// instead of a whole 12 us solve later (the head of the tile side's critical hand-over).  Right-hand side and flag as the
// ring solve of ps_tile_role.  Returns 0, -1 (abandoned) or -2 (the matrix has failed: the caller only passes its flag on).
static __device__ __forceinline__ int pf_stream_S(const PsArgs& a, int b, int J, int I, int* ok_lds, int* peek_lds,
                                                  unsigned long long* trs = nullptr) {
  double* const Wl = reinterpret_cast<double*>(pf_lds_raw());
  int tid_ = threadIdx.x;
  asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int ld = a.ld;
  double* const Ab = a.K + (size_t)b * a.mstride + (size_t)I * 128 * ld + (size_t)J * 128;
  // Every row block of W_JJ that is out already goes to LDS in ONE burst, all its loads in flight together and ahead of the A
  // fragments': this task has no update to hide a row block's latency under, and a block fetched when its step begins -- or
  // one step ahead -- cost the remote round trip every step (2.2 us a step, 17.8 us for the eight: in-kernel stamps).
  int have0 = 0;
  {
    if (tid == 0) {
      const unsigned* const wrow = a.flags + PS_HDR + (size_t)a.B * a.nblk * (3 + a.nblk) + (size_t)b * a.nblk + J;
      const unsigned* const wrdy = a.flags + PS_HDR + (size_t)b * a.nblk + J;
      const unsigned fr = ps_ld(wrow), fw = ps_ld(wrdy);
      const int st = __hip_atomic_load(a.status + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ps_acquire();
      *ok_lds = st != 0 ? -2 : (fr >= 7u ? (fw >= 1u ? 8 : 7) : (int)fr);
    }
    __syncthreads();
    have0 = *ok_lds;
    __syncthreads();
    if (have0 < 0) return have0;  // the matrix has failed
  }
  double af[32];
  {
    const double* const Wg = a.W + ((size_t)b * a.nblk + J) * (128 * 128);
    const int sr = tid >> 5, sc0 = tid & 31;  // thread -> (row, first column) of a 16-row block, as the steps' staging
    double v[8][4];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (q < have0) {
#pragma unroll
        for (int i = 0; i < (16 * (q + 1) + 31) / 32; i++)
          v[q][i] = (sc0 + 32 * i < 16 * (q + 1)) ? Wg[(size_t)(16 * q + sr) * 128 + sc0 + 32 * i] : 0.0;
      }
    }
    pf_afrag_issue(Ab + (unsigned)((16 * w + lr) * ld), lk, af);
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (q < have0) {
        double* const dst = Wl + PH_WOFF(q) + sr * PH_WLDQ(q);
#pragma unroll
        for (int i = 0; i < (16 * (q + 1) + 31) / 32; i++)
          if (sc0 + 32 * i < 16 * (q + 1)) dst[sc0 + 32 * i] = v[q][i];
      }
    }
    pf_lds_barrier();
    pf_afrag_transpose(af);
  }
  if (trs) {  // (tracing only: the A fragments have landed)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trs[4] = wall_clock64();
  }
  d4 dt[5];
  const int tz[5] = {0, 0, 0, 0, 0};
  // (S(J+2, J) feeds the same pre-updates as the chain helper's block, S(J+3, J) the next column's: handed on column block by
  // column block as well)
  unsigned* const xcol = I <= J + 3 ? a.flags + PS_XCOL(a.B, a.nblk) + ((size_t)b * a.nblk + I) * 3 + (I - J - 1) : nullptr;
  const int have = pf_pair_steps<0, 0>(a, b, J, ok_lds, have0, have0, af, Ab, ld, dt, tz, tz, Wl, Wl, tid, w, lane, nullptr, peek_lds, xcol);
  if (have < 0) return have;
  if (trs) trs[3] = wall_clock64();  // (the eight steps are done)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (xcol) {
    __syncthreads();
    if (tid == 0) ps_st(xcol, 8u);
  }
  {
    const double* const zg = a.yw + (size_t)b * a.ystride + J * 128;
    double* const yi = a.yw + (size_t)b * a.ystride + I * 128;
    double zc[8], yv[4];
#pragma unroll
    for (int j = 0; j < 8; j++) zc[j] = zg[16 * j + lr];
#pragma unroll
    for (int r = 0; r < 4; r++) yv[r] = yi[16 * w + lk + 4 * r];
    double xv[4][8];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int j = 0; j < 8; j++) xv[r][j] = Ab[(unsigned)((16 * w + lk + 4 * r) * ld + 16 * j + lr)];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * w + lk + 4 * r;
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++) part = __builtin_fma(xv[r][j], zc[j], part);
      part += __shfl_xor(part, 1);
      part += __shfl_xor(part, 2);
      part += __shfl_xor(part, 4);
      part += __shfl_xor(part, 8);
      if (lr == 0) yi[row] = yv[r] - part;
    }
  }
  return 0;
}

// The chain role of the launch-free factorisation: workgroup b of ps_kernel walks the block columns of matrix b (p = 0), or
// -- chain pairs -- workgroups (b, 0) and (b, 1) take the even and the odd block columns and prepare each other's next block.
template <int PAIR>
static __device__ __forceinline__ void ps_chain_role(const PsArgs& a, int b, int p) {
  const int tid = threadIdx.x;
  __shared__ int ps_ok, ps_peek;
  unsigned* const flags = a.flags;
  unsigned* const wready = flags + PS_HDR + (size_t)b * a.nblk;
  unsigned* const wrow = flags + PS_HDR + (size_t)a.B * a.nblk * (3 + a.nblk) + (size_t)b * a.nblk;
  unsigned long long* const tr = a.trace ? a.trace + (size_t)b * a.nblk * 8 : nullptr;
  PfPre pre;
  pre.state = 0;
  if (!PAIR && a.gen) {  // block (0, 0) and the first 128 entries of the right-hand side come from a tile worker of this launch
    if (tid == 0) {
      const bool ok = ps_wait_ge(flags + PS_GEN(a.B, a.nblk) + (size_t)b * a.nblk * a.nblk, 1u, flags + PS_ERROR, a.spin_limit);
      ps_acquire();
      ps_ok = ok ? 1 : 0;
    }
    __syncthreads();
    if (!ps_ok) return;
    __syncthreads();
  }
  for (int J = 0; J < a.nblk; J++) {
    if (PAIR && (J & 1) != p) {  // the partner factorises column J: prepare block (J+1, J+1) under it
      if (J + 1 < a.nblk && pf_pair_helper(a, b, J, &ps_ok, &ps_peek, tr) < 0) return;
      continue;
    }
    if (tr && tid == 0) tr[J * 8 + 0] = wall_clock64();
    // (J > 0: the block and its right-hand side are in LDS, pf_chain_next / pf_pair_helper left them there)
    const int failed = pf_block<0, 0, 0, PAIR ? 0 : 1>(b, a.K, a.W, a.yw, a.acc, a.lml, a.status, a.n, a.ld, a.mstride, a.ystride,
                                                       a.nblk, J, PfGen(), J > 0, PAIR ? wrow + J : nullptr, a, pre);
    if (tr && tid == 0) tr[J * 8 + 1] = wall_clock64();
    ps_publish_barrier();
    // (single chain) wave 4 saw block (J+1, J) handed over while pf_block(J) ran: its A fragments are requested NOW, behind the
    // drain of this block's stores and under the release + flag store below -- by wave 0, whose release waits for its own
    // memory operations, behind them -- instead of after pf_chain_next's poll and acquire
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool early = !PAIR && !failed && J + 1 < a.nblk && pf_lds().fail[1] != 0;
    if (early && wv != 0) pf_pre_issue(a, b, J, wv, tid & 15, (tid & 63) >> 4, pre);
    if (tid == 0) {
      ps_release();
      // a failed matrix (status set above) releases every later column at once -- and the panel blocks this workgroup owes
      // the tile tasks: they see the status and only pass their own flags on
      for (int j = J; j < (failed ? a.nblk : J + 1); j++) ps_st(wready + j, 1u);
      if (failed) {
        for (int j = J; j + 1 < a.nblk; j++)
          ps_st(flags + PS_HDR + (size_t)2 * a.B * a.nblk + ((size_t)b * a.nblk + j + 1) * a.nblk + j, 1u);
        if (PAIR)
          for (int j = J; j < a.nblk; j++) ps_st(wrow + j, 8u);
      }
    }
    if (failed) return;
    if (early && wv == 0) pf_pre_issue(a, b, J, wv, tid & 15, (tid & 63) >> 4, pre);
    if (!PAIR && J + 1 < a.nblk && pf_chain_next(a, b, J, &ps_ok, tr, pre) < 0) return;  // (abandoned: the host redoes the batch)
  }
}

