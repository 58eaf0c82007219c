// RETIRED in round 5 (not compiled into libbgp.so; kept for the record, docs/EXPERIMENTS.md B5):
// Gram tiles generated inside the first trailing update that touches them, instead of being written by the Gram kernel and read
// back.  Bit-identical K and LML; measured 15.6 vs 15.9 ms per step at config C (the Gram build 0.84 -> 0.13 ms, the trailing
// update 5.6 -> 6.2 ms: on this chip a VALU instruction costs the fp64 MFMA its issue slots), 2-5 % slower on small batches.
// It was s4_tile<.., GEN = 1, STAT, FORM>'s accumulator initialisation (bgp_s4.h) behind syrk4_kernel<64, 0, 1, S, F>, fed by
// bgp_launch_kbuild_col0 (block column 0 only) through struct S4Gen {Xs, alpha, H, n, d, npad, dpad, stat, form}.
#if 0
// Gram entries of a wave's NR x NC block, produced in the accumulator layout instead of being loaded: the FIRST trailing
// update that touches a tile of K builds it (scaled inputs Xs are k-major and L2-resident: 256 KB per matrix at
// config C), so the Gram matrix is never written and read back except for block column 0.  Same arithmetic as the
// Gram kernels (bgp_kbuild.hip: differences squared and summed in dimension order with one fma each, then
// kb_epilogue's expressions without implicit contraction): bit-identical K.  OPT-IN (BGP_FUSED_GRAM=1): measured on
// MI355X the generation is NOT hidden under the other workgroups' MFMAs -- a VALU instruction costs the fp64 MFMA its
// issue slots (tools/archive/mfma_interleave_probe.hip) -- so only the saved HBM round trip of K shows: 15.6 vs 15.9 ms per step
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

#endif
