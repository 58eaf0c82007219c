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

// Compile-time stationary kernel (same operation order as sklearn/kernels.py:1553-1560, 1713-1733;
// the Matern-5/2 term K**2/3.0 is evaluated as t*t*(1/3): <= 1 ulp from the reference's division).
// No implicit fused multiply-add here (hipcc's default contraction is decided per call site by the backend: two
// kernels inlining this function could otherwise round the Matern polynomial differently).
template <int STAT>
static __device__ __forceinline__ double kb_stationary(double r2) {
#pragma clang fp contract(off)
  if (STAT == BGP_RBF) return exp(-0.5 * r2);
  const double dist = sqrt(r2);
  if (STAT == BGP_MATERN12) return exp(-dist);
  if (STAT == BGP_MATERN32) {
    const double t = dist * 1.7320508075688772;  // math.sqrt(3)
    return (1.0 + t) * exp(-t);
  }
  const double t = dist * 2.23606797749979;  // math.sqrt(5)
  return (1.0 + t + t * t * 0.3333333333333333) * exp(-t);
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

