// Device helpers shared by the libbgp kernels (gfx950 only).
#pragma once
#include "bgp_common.h"

static __device__ __forceinline__ double bgp_stationary(double r2, int stat) {
  // same operation order as sklearn/kernels.py:1553-1560 and :1713-1733
  if (stat == BGP_RBF) return exp(-0.5 * r2);
  double dist = sqrt(r2);
  if (stat == BGP_MATERN12) return exp(-dist);
  if (stat == BGP_MATERN32) {
    double t = dist * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + t) * exp(-t);
  }
  double t = dist * 2.23606797749979;  // math.sqrt(5)
  return (1.0 + t + t * t / 3.0) * exp(-t);
}

// ---- in-launch hand-off between workgroups (persistent factorisation; cdna_hip_programming.md section 6, Guideline 16) ----
// Every shared word is accessed with relaxed AGENT-scope atomics (sc1: they bypass the per-CU L1); payload visibility
// comes from ONE agent-scope release on the producer (after every storing wave has drained its stores and the workgroup
// has met at a barrier) and ONE agent-scope acquire on the consumer (after the poll has succeeded, before a barrier
// that releases the other waves to their plain loads).  Never from placement, never from workgroup scope.
static __device__ __forceinline__ unsigned ps_ld(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void ps_st(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// ONE lane: spin (with s_sleep) until *p >= want.  False when somebody raised the error word or this wait outlasted
// `limit` ticks of the 100 MHz wall clock (it raises the error word itself then): every loop in both kernels leaves.
static __device__ __forceinline__ bool ps_wait_ge(const unsigned* p, unsigned want, unsigned* err, unsigned long long limit) {
  if (ps_ld(p) >= want) return true;
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; it++) {
    __builtin_amdgcn_s_sleep(2);
    if (ps_ld(p) >= want) return true;
    if ((it & 15) == 15) {
      if (ps_ld(err) != 0) return false;
      if (wall_clock64() - t0 > limit) {
        ps_st(err, 1u);
        return false;
      }
    }
  }
}
// ... for TWO words at once (both requested before either is looked at: one round trip to the flags instead of two)
static __device__ __forceinline__ bool ps_wait_ge2(const unsigned* p, unsigned want, const unsigned* q, unsigned wantq, unsigned* err,
                                                   unsigned long long limit) {
  const unsigned long long t0 = wall_clock64();
  for (unsigned it = 0;; it++) {
    const unsigned a = ps_ld(p), b = ps_ld(q);
    if (a >= want && b >= wantq) return true;
    __builtin_amdgcn_s_sleep(2);
    if ((it & 15) == 15) {
      if (ps_ld(err) != 0) return false;
      if (wall_clock64() - t0 > limit) {
        ps_st(err, 1u);
        return false;
      }
    }
  }
}
// EVERY thread, after its last payload store: drain, meet.  Then ONE lane: ps_signal_*.
static __device__ __forceinline__ void ps_publish_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
static __device__ __forceinline__ void ps_release() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the compiler may drop the wait behind buffer_wbl2: restated)
}
static __device__ __forceinline__ void ps_signal_add(unsigned* p) {
  ps_release();
  __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// write-through payload store (global_store_dwordx2 ... sc1): complete -- s_waitcnt vmcnt -- means in memory, no L2 write-back needed
// before the flag that hands it over (MI355X_MICROARCH.md "publish-large")
static __device__ __forceinline__ void ps_st_wt(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void ps_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// XCD-aware block -> (matrix b, tile t) map.  The dispatcher places block id on XCD id % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"); matrix b is pinned to XCD b % 8 and each XCD walks
// through its matrices one after another so that the panels a matrix's tiles share stay in that
// XCD's private 4 MiB L2.  Placement only affects speed, never results.
// With fewer than 8 matrices in the batch that pinning would leave XCDs idle, so the tiles of a matrix
// are spread round-robin over all XCDs instead (grids are sized 8*ceil(B/8)*tiles either way).
static __device__ __forceinline__ void bgp_map_block(int id, int tiles, int B, int& b, int& t) {
  if (B < 8) {
    b = id / tiles;
    t = id - b * tiles;
    return;
  }
  int x = id & 7;
  int q = id >> 3;
  int m = q / tiles;
  t = q - m * tiles;
  b = 8 * m + x;
}

// Active row block t at step k of a factorisation: the nlow = nblk-k-1 remaining blocks of K, then (posterior builds on
// the augmented matrix [[K, .], [I, 0]]) the first k+1 block rows of the identity part, which starts at block row aug.
static __device__ __forceinline__ int bgp_rowblk(int t, int k, int nlow, int aug) {
  return (t < nlow) ? (k + 1 + t) : (aug + (t - nlow));
}

static __device__ __forceinline__ void bgp_tri_decode(int t, int& ti, int& tj) {
  int r = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while (r * (r + 1) / 2 > t) r--;
  while ((r + 1) * (r + 2) / 2 <= t) r++;
  ti = r;
  tj = t - r * (r + 1) / 2;
}

// Domain-specific exp and sqrt for the Gram epilogues.  The library (ocml) versions carry range and class handling the
// domain does not need -- here the argument of exp is <= 0 and that of sqrt is a sum of squares of O(1) numbers, never
// denormal -- and measured in kbuild2_kernel's ISA that handling was most of the epilogue (v_cndmask / v_ldexp_f64 /
// v_cmp_class_f64 scaling steps around every call).  Every operation is an explicit fma / mul (no implicit contraction:
// the K-build kernels, the fused n <= 128 kernel and the tile generation inside the trailing update must produce the
// same bits).
//
// exp(-t), t >= 0: n = rint(-t log2 e), r = -t - n ln2 (two-part ln2, |r| <= 0.3466), Taylor polynomial to r^13
// (remainder 4e-18 relative), scaled by 2^n with ONE v_ldexp_f64 (which also produces the gradual underflow below
// t = 708 and 0 beyond 745).  Measured against libm over 2e7 arguments in [0, 745]: <= 2.3e-16 relative (<= 1 ulp).
// t > 800 is clamped (the result is 0 either way; keeps -inf out of the reduction), NaN propagates.
static __device__ __forceinline__ double kb_exp_neg(double t) {
#pragma clang fp contract(off)
  const double x = -((t > 800.0) ? 800.0 : t);
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, -6.93147180369123816490e-01, x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;               // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);    // 1/12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);   // 1/11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);   // 1/10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);  // 1/9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);    // 1/8!
  p = __builtin_fma(p, r, 0.0001984126984126984);   // 1/7!
  p = __builtin_fma(p, r, 0.001388888888888889);    // 1/6!
  p = __builtin_fma(p, r, 0.008333333333333333);    // 1/5!
  p = __builtin_fma(p, r, 0.041666666666666664);    // 1/4!
  p = __builtin_fma(p, r, 0.16666666666666666);     // 1/3!
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_ldexp(p, (int)n);
}

// sqrt(x), x >= 0 and not denormal: hardware seed v_rsq_f64, one coupled Goldschmidt step for (sqrt, 1/(2 sqrt)), one
// residual correction -- correctly rounded on every one of 2e7 test arguments in [1e-34, 1e4] with a 2^-23 seed; no
// scaling, no class test.  x == 0 (coinciding points): 0.
static __device__ __forceinline__ double kb_sqrt_pos(double x) {
#pragma clang fp contract(off)
  const double y0 = __builtin_amdgcn_rsq(x);
  double g = x * y0, h = 0.5 * y0;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return (x == 0.0) ? 0.0 : g;
}

// Compile-time stationary kernel (same operation order as sklearn/kernels.py:1553-1560, 1713-1733;
// the Matern-5/2 term K**2/3.0 is evaluated as t*t*(1/3): <= 1 ulp from the reference's division; exp / sqrt are the
// domain-specific versions above: <= 1 ulp from libm's).
// No implicit fused multiply-add here (hipcc's default contraction is decided per call site by the backend: two
// kernels inlining this function could otherwise round the Matern polynomial differently).
template <int STAT>
static __device__ __forceinline__ double kb_stationary(double r2) {
#pragma clang fp contract(off)
  if (STAT == BGP_RBF) return kb_exp_neg(0.5 * r2);
  const double dist = kb_sqrt_pos(r2);
  if (STAT == BGP_MATERN12) return kb_exp_neg(dist);
  if (STAT == BGP_MATERN32) {
    const double t = dist * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + t) * kb_exp_neg(t);
  }
  const double t = dist * 2.23606797749979;  // math.sqrt(5)
  return (1.0 + t + t * t * 0.3333333333333333) * kb_exp_neg(t);
}


// (stationary, form) -> instantiation
#define KB_DISPATCH(STATV, FORMV, CALL)                                      \
  do {                                                                       \
    const int key__ = (STATV)*2 + (FORMV);                                   \
    switch (key__) {                                                         \
      case 0: { constexpr int S = 0, F = 0; CALL; } break;                   \
      case 1: { constexpr int S = 0, F = 1; CALL; } break;                   \
      case 2: { constexpr int S = 1, F = 0; CALL; } break;                   \
      case 3: { constexpr int S = 1, F = 1; CALL; } break;                   \
      case 4: { constexpr int S = 2, F = 0; CALL; } break;                   \
      case 5: { constexpr int S = 2, F = 1; CALL; } break;                   \
      case 6: { constexpr int S = 3, F = 0; CALL; } break;                   \
      default: { constexpr int S = 3, F = 1; CALL; } break;                  \
    }                                                                        \
  } while (0)

// ---- A-operand fragments straight from global memory (the chain role of the launch-free factorisation, bgp_pf.h; the panel
// solve trsm5_kernel, bgp_syrk4.hip) ----
// A-operand fragments of 16 rows x 128 columns for the f64 16x16x4 MFMA: lane (lr, lk) wants A[row lr][4 t + lk], t = 0 .. 31 -- as
// 8-byte loads that is 32 B per row per instruction.  Loaded instead as 32 contiguous bytes per lane (two 16-byte loads, a whole
// 128-B line per row per pair of instructions) and transposed 4 x 4 across the four 16-lane rows with the gfx950 lane-swap
// instructions (v_permlane32_swap: rows 2, 3 of one register <-> rows 0, 1 of another; v_permlane16_swap: odd rows <-> even rows):
// the same values in the same registers, half the memory instructions and full lines (handed-off blocks arrive at 60-120 GB/s per
// workgroup, MI355X_MICROARCH.md "handoff-payload": the 128 KB block is most of a chain step's entry).
static __device__ __forceinline__ void pf_swap_rows32(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}
static __device__ __forceinline__ void pf_swap_rows16(double& x, double& y) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}
// (the two halves, for callers that put other work between the loads and the first use)
static __device__ __forceinline__ void pf_afrag_issue(const double* __restrict__ rowp, int lk, double (&af)[32]) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  const d2v* const q = reinterpret_cast<const d2v*>(rowp + 4 * lk);
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const d2v u = q[8 * g], v = q[8 * g + 1];
    af[4 * g] = u[0];
    af[4 * g + 1] = u[1];
    af[4 * g + 2] = v[0];
    af[4 * g + 3] = v[1];
  }
}
static __device__ __forceinline__ void pf_afrag_transpose(double (&af)[32]) {
#pragma unroll
  for (int g = 0; g < 8; g++) {
    pf_swap_rows32(af[4 * g], af[4 * g + 2]);
    pf_swap_rows32(af[4 * g + 1], af[4 * g + 3]);
    pf_swap_rows16(af[4 * g], af[4 * g + 1]);
    pf_swap_rows16(af[4 * g + 2], af[4 * g + 3]);
  }
}
// rowp = &A[this lane's row][0] (32-byte aligned), lk = lane >> 4
static __device__ __forceinline__ void pf_load_afrag(const double* __restrict__ rowp, int lk, double (&af)[32]) {
  typedef double d2v __attribute__((ext_vector_type(2)));
  const d2v* const q = reinterpret_cast<const d2v*>(rowp + 4 * lk);
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const d2v u = q[8 * g], v = q[8 * g + 1];
    af[4 * g] = u[0];
    af[4 * g + 1] = u[1];
    af[4 * g + 2] = v[0];
    af[4 * g + 3] = v[1];
  }
  pf_afrag_transpose(af);
}

