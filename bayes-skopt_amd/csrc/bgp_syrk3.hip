// Trailing update on the fast fp64 MFMA form with DIRECT global->LDS staging.
//
// syrk3_kernel computes the same update as syrk2_kernel (bgp_chol.hip):  A_IJ -= X_I X_J^T over a panel
// of width K (128 or 256), one 128x128 tile per 256-thread workgroup, each wave a 64x64 block.  What
// differs is how the operands reach the MFMAs:
//   * `global_load_lds_dwordx4` writes each 128x32 chunk straight into LDS (no VGPR round trip, so the
//     128 accumulator registers of v_mfma_f64_4x4x4_4b_f64 still leave room for TWO workgroups per CU,
//     which is what hides one workgroup's C-tile traffic behind the other's MFMAs);
//   * the LDS image is the hardware's lane-linear one (row-major [128][32], a wave instruction writes
//     4 rows x 256 B); bank conflicts are avoided by an XOR swizzle of the 16-byte column chunks applied to
//     the per-lane SOURCE address and again on every read (cdna_hip_programming.md section 5.4 rule 21):
//         element (row, k) lives at  row*32 + (((k>>1) ^ (row&15)) << 1) + (k&1).
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"

#define S3_KC 32
struct __attribute__((aligned(16))) Smem3 {
  double A[128 * S3_KC];
  double B[128 * S3_KC];
};

// wave w stages rows [32w, 32w+32) of a 128 x 32 chunk: 8 instructions x (4 rows x 256 B)
static __device__ __forceinline__ void s3_issue(double* __restrict__ tile, const double* __restrict__ src, size_t ld,
                                                int w, int lane) {
  const int rsub = lane >> 4, slot = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int row = 32 * w + 4 * i + rsub;
    const int c = slot ^ (row & 15);
    const double* g = src + (size_t)row * ld + c * 2;
    double* l = tile + (32 * w + 4 * i) * S3_KC;  // wave-uniform; the hardware adds lane * 16 B
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
  }
}

// acc[i][j] (16x16 tile (i,j) of this wave's 64x64 block) -= A B^T over one chunk, 4x4x4 four-block MFMA
// (operand patterns as in bgp_gemm8.h: A lane = A[4r + (l&3)][l>>4], B lane = B[l&15][l>>4]).
// The swizzled addresses are decomposed into a few lane-dependent bases plus compile-time offsets so
// that the reads use immediate offsets instead of one address register each:
//   A: row = r0+16i+4r+(l&3):  (2kk+kh) ^ (4r + l4) = (((kk>>1) ^ r) << 2) | ((((kk&1)<<1)|kh) ^ l4)
//   B: row = c0+16j+(l&15):    (2kk+kh) ^ lr  is lane dependent, one xor per k-step
template <int CREL>
static __device__ __forceinline__ void s3_mma(const double* __restrict__ As, const double* __restrict__ Bs,
                                              d4 (&acc)[4][4], int r0, int c0, int lane) {
  const int lr = lane & 15, lk = lane >> 4, l4 = lane & 3, kb = lk & 1, kh = lk >> 1;
  const double* pA0 = As + (r0 + l4) * S3_KC + ((kh ^ l4) << 1) + kb;        // even k-steps
  const double* pA1 = As + (r0 + l4) * S3_KC + (((2 | kh) ^ l4) << 1) + kb;  // odd k-steps
  const double* pB = Bs + (c0 + lr) * S3_KC + kb;
#pragma unroll 2  // keeps the live fragment set small enough for two workgroups per CU
  for (int kk = 0; kk < S3_KC / 4; kk++) {
    const int sB = ((2 * kk + kh) ^ lr) << 1;
    double b[4];
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = pB[16 * j * S3_KC + sB];
    const double* pA = (kk & 1) ? pA1 : pA0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      double a[4];
#pragma unroll
      for (int r = 0; r < 4; r++) a[r] = -pA[(16 * i + 4 * r) * S3_KC + ((((kk >> 1) ^ r) << 2) << 1)];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j + CREL > i) continue;  // compile-time (lower part of a diagonal block)
#pragma unroll
        for (int r = 0; r < 4; r++)
          acc[i][j][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[r], b[j], acc[i][j][r], 0, 0, 0);
      }
    }
  }
}

__global__ void __launch_bounds__(256, 2) syrk3_kernel(double* __restrict__ Kbuf, const int* __restrict__ status,
                                                       int ld, size_t mstride, int nblk, int kp, int K, int jstart,
                                                       int colmode, int B) {
  const int nt = nblk - jstart;
  const int ntile = colmode ? nt : nt * (nt + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntile, B, b, t);
  if (b >= B || status[b] != 0) return;
  int ti, tj;
  if (colmode) {
    ti = t;
    tj = 0;
  } else {
    bgp_tri_decode(t, ti, tj);
  }
  const int I = jstart + ti, J = jstart + tj;
  __shared__ Smem3 sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r0 = (w >> 1) * 64, c0 = (w & 1) * 64;
  double* M = Kbuf + (size_t)b * mstride;
  const double* XI = M + (size_t)(I * 128) * ld + kp * 128;
  const double* XJ = M + (size_t)(J * 128) * ld + kp * 128;
  double* C = M + (size_t)(I * 128) * ld + J * 128;
  const bool diag = (I == J);
  if (diag && w == 1) {  // block (rows 0..63, cols 64..127) of a diagonal tile is never read again
    for (int k0 = 0; k0 < K; k0 += S3_KC) {
      __syncthreads();
      s3_issue(sm.A, XI + k0, (size_t)ld, w, lane);
      __syncthreads();
    }
    return;
  }
  d4 acc[4][4];
  gk_load_c<4, 4, -64>(C, (size_t)ld, acc, r0, c0, lane);
  for (int k0 = 0; k0 < K; k0 += S3_KC) {
    __syncthreads();  // everybody finished reading the previous chunk
    s3_issue(sm.A, XI + k0, (size_t)ld, w, lane);
    if (!diag) s3_issue(sm.B, XJ + k0, (size_t)ld, w, lane);
    __syncthreads();  // (hipcc drains vmcnt before the barrier: the DMA writes have landed)
    s3_mma<-64>(sm.A, diag ? sm.A : sm.B, acc, r0, c0, lane);
  }
  gk_store_c<4, 4, -64>(C, (size_t)ld, acc, r0, c0, lane);
}

void bgp_launch_syrk3(hipStream_t st, int grid, double* dK, const int* dstatus, int ld, size_t mstride, int nblk, int kp,
                      int K, int jstart, int colmode, int B) {
  hipLaunchKernelGGL(syrk3_kernel, dim3(grid), dim3(256), 0, st, dK, dstatus, ld, mstride, nblk, kp, K, jstart, colmode,
                     B);
}
