// Kernel-matrix build: tiled pairwise-distance kernel staging X tiles in LDS.
//
// Replaces kernel_(X_train_) + diagonal add -- sklearn/kernels.py:1708-1738 (Matern.__call__),
// :1553-1560 (RBF), :966 (Product), :866 (Sum), :1273 (Constant), :1402 (White),
// sklearn/_gpr.py:585 / bask/bayesgpr.py:204 (K[diag] += alpha).
//
// One 256-thread workgroup produces one 128x128 tile; thread (tx,ty) of the 16x16 thread grid owns
// the 8x8 strided micro-tile rows ty+16r, cols tx+16c, so that for a fixed (r,c) the 16 lanes of a
// row write 128 contiguous bytes (full cache lines) and LDS reads of the column operand are
// conflict-free.  X tiles are pre-divided by the walker's length scales while being staged.
// HBM-bound by design: 8 B written per pair, X re-read from L2.
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_ring.h"
#include "bgp_kb.h"

// Generic tile body: out[(i0+..)][(j0+..)] = k(A_i, B_j); A is (na x d), Bm is (nb x d), row-major.
// GRAM != 0: A == Bm is the training set, diagonal gets c(+1) + s2 + alpha_i, padding gets identity.
// Tiles that are fully inside the data and off the diagonal take a check-free epilogue.
template <int GRAM, int STAT, int FORM>
static __device__ __forceinline__ void kbuild_tile(const double* __restrict__ A, int na,
                                                   const double* __restrict__ Bm, int nb, int d,
                                                   const double* __restrict__ h, const double* __restrict__ alpha,
                                                   int i0, int j0, double* __restrict__ out, size_t ldo, int out_rows,
                                                   int out_cols, double (&acc)[8][8]) {
  __shared__ double xi[KB_DK][BGP_TILE_LD];
  __shared__ double xj[KB_DK][BGP_TILE_LD];
  __shared__ double ell[KB_DK];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) acc[r][c] = 0.0;

  for (int k0 = 0; k0 < d; k0 += KB_DK) {
    const int kc = min(KB_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += 256) {
      int row = idx / kc, k = idx - row * kc;
      int gi = i0 + row, gj = j0 + row;
      double l = ell[k];
      xi[k][row] = (gi < na) ? A[(size_t)gi * d + k0 + k] / l : 0.0;
      xj[k][row] = (gj < nb) ? Bm[(size_t)gj * d + k0 + k] / l : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[8], b[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) b[c] = xj[k][tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 8; r++) {  // (subtracts batched ahead of their squares: see kbuild2_kernel)
        double df[8];
#pragma unroll
        for (int c = 0; c < 8; c++) df[c] = a[r] - b[c];
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
#pragma unroll
        for (int c = 0; c < 8; c++) acc[r][c] = fma(df[c], df[c], acc[r][c]);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
    }
  }
  kb_epilogue<GRAM, STAT, FORM>(acc, na, nb, d, h, alpha, i0, j0, out, ldo, out_rows, out_cols, tx, ty);
}

template <int STAT, int FORM>
__global__ void __launch_bounds__(256) kbuild_gram_kernel(const double* __restrict__ X,
                                                           const double* __restrict__ alpha,
                                                           const double* __restrict__ H, double* __restrict__ Kbuf,
                                                           const double* __restrict__ y, double* __restrict__ yw,
                                                           int n, int d, int npad, int nblk, int B, int full, int ld,
                                                           int use_alpha, size_t xstride) {
  const int ntiles = full ? nblk * nblk : nblk * (nblk + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntiles, B, b, t);
  if (b >= B) return;
  int ti, tj;
  if (full) {
    ti = t / nblk;
    tj = t - ti * nblk;
  } else {
    bgp_tri_decode(t, ti, tj);
  }
  const double* h = H + (size_t)b * (d + 2);
  // ld == npad for LML batches, 2*npad for the augmented matrices of posterior builds
  double* out = Kbuf + (size_t)b * ld * ld;
  // working right-hand side of walker b (becomes z = L^-1 y during the factorisation)
  if (ti == tj && threadIdx.x < 128) yw[(size_t)b * ld + ti * 128 + threadIdx.x] = y[ti * 128 + threadIdx.x];
  const double* Xb = X + (size_t)b * xstride;  // per-walker warped inputs (xstride == 0: shared)
  double acc[8][8];
  kbuild_tile<1, STAT, FORM>(Xb, n, Xb, n, d, h, use_alpha ? alpha : nullptr, ti * 128, tj * 128, out, (size_t)ld, npad,
                             npad, acc);
}

// blockIdx.y = item of a batch: hyper-parameters h + b (d+2), output out + b ostride (the inputs are shared).
// vec != nullptr: the tile also contributes to the matrix-vector product  out_b vec_b  (the posterior mean K_* alpha
// of BayesGPR.predict) while its values are still in registers: dpart[(b tiles_j + tj) mpad + row] = the tile's 128-column
// share of the row's dot product (16 lanes share a row: fixed shuffle order); the caller adds the column tiles in
// order (rowdot_reduce_kernel) -- no second pass over K_*, no floating-point atomics.
template <int STAT, int FORM>
__global__ void __launch_bounds__(256) kbuild_cross_kernel(const double* __restrict__ Xq, int m,
                                                            const double* __restrict__ Xt, int n, int d,
                                                            const double* __restrict__ h, double* __restrict__ out,
                                                            int ldo, int tiles_j, size_t ostride,
                                                            const double* __restrict__ vec, size_t svec,
                                                            double* __restrict__ dpart, int mpad) {
  const int ti = blockIdx.x / tiles_j, tj = blockIdx.x - ti * tiles_j, b = blockIdx.y;
  double acc[8][8];
  kbuild_tile<0, STAT, FORM>(Xq, m, Xt, n, d, h + (size_t)b * (d + 2), nullptr, ti * 128, tj * 128,
                             out + (size_t)b * ostride, (size_t)ldo, m, n, acc);
  if (vec) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const double* vb = vec + (size_t)b * svec;
    double vv[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gj = tj * 128 + tx + 16 * c;
      vv[c] = (gj < n) ? vb[gj] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) {
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 8; c++) s += acc[r][c] * vv[c];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      s += __shfl_xor(s, 8);
      if (tx == 0) dpart[((size_t)b * tiles_j + tj) * mpad + ti * 128 + ty + 16 * r] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Pipelined Gram build (the LML and posterior paths).  kbuild_gram_kernel above loads, divides and transposes its
// X tiles at the head of every tile with nothing to overlap (two workgroups per CU at 128 accumulator registers): it
// ran at ~40 % of the fp64 VALU rate.  Here
//   * xscale_kernel writes the walker's scaled inputs ONCE, k-major:  Xs[b][k][i] = X_b[i][k] / l_k  (rows >= n and
//     dimensions >= d zero), so a tile operand for 16 dimensions is 16 contiguous 1 KB rows -- the LDS-DMA's
//     lane-linear image, no transpose, no division in the tile loop (same quotient as before, bit for bit);
//   * kbuild2_kernel gives every workgroup KB2_TPW consecutive tiles of one matrix and streams their (tile, 16
//     dimensions) operand chunks through a two-stage LDS ring with `global_load_lds_dwordx4`: the chunk after the
//     one being accumulated is always in flight.
// Same accumulation order (dimension ascending) and the same epilogue as kbuild_tile: identical K.
// ------------------------------------------------------------------------------------------
#define KB2_TPW 4
#define KB2_WAVES 4    // 256 threads, 8 x 8 pairs per thread.  (512 threads x 4 x 8 pairs -- twice the waves per SIMD --
#define KB2_R 8        // measured no faster: 0.875 vs 0.86 ms at config C; the kernel is fp64-VALU bound, not latency bound)
__global__ void __launch_bounds__(256) xscale_kernel(const double* __restrict__ X, size_t xstride,
                                                      const double* __restrict__ H, double* __restrict__ Xs, int n, int d,
                                                      int npad, int dpad) {
  const int b = blockIdx.y;
  const double* Xb = X + (size_t)b * xstride;
  const double* h = H + (size_t)b * (d + 2);
  double* out = Xs + (size_t)b * dpad * npad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)dpad * npad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(idx / npad), i = (int)(idx - (size_t)k * npad);
    out[idx] = (k < d && i < n) ? Xb[(size_t)i * d + k] / exp(h[1 + k]) : 0.0;
  }
}

// one chunk = operands of tile (ti, tj) for dimensions [16 kb, 16 kb + 16): xi then xj, 16 KB each; wave w issues
// its share of the 16 rows (dimensions) of both (a wave instruction = 64 lanes x 16 B = one 1 KB row)
static __device__ __forceinline__ void kb2_issue(const double* Xs_b, int npad, int ti, int tj, int kb, unsigned lds_buf,
                                                 int w, int lane) {
#pragma unroll
  for (int q = 0; q < KB_DK / KB2_WAVES; q++) {
    const int k = (KB_DK / KB2_WAVES) * w + q;
    const double* rowp = Xs_b + (size_t)(kb * KB_DK + k) * npad;
    s4_glds(rowp + ti * 128, (unsigned)lane * 16u, lds_buf + (unsigned)(k * 1024));
    s4_glds(rowp + tj * 128, (unsigned)lane * 16u, lds_buf + (unsigned)(KB_DK * 1024 + k * 1024));
  }
}

template <int STAT, int FORM>
__global__ void __launch_bounds__(64 * KB2_WAVES, 2) kbuild2_kernel(const double* __restrict__ Xs, const double* __restrict__ alpha,
                                                       const double* __restrict__ H, double* __restrict__ Kbuf,
                                                       const double* __restrict__ y, double* __restrict__ yw, int n,
                                                       int d, int npad, int dpad, int nblk, int B, int full, int ld,
                                                       int use_alpha) {
  const int ntiles = (full == 2) ? nblk : full ? nblk * nblk : nblk * (nblk + 1) / 2;  // full == 2: block column 0 only
  const int ngroups = (ntiles + KB2_TPW - 1) / KB2_TPW;
  int b, g;
  bgp_map_block(blockIdx.x, ngroups, B, b, g);
  if (b >= B) return;
  __shared__ __attribute__((aligned(1024))) double smem[2][2 * KB_DK * 128];  // [stage][xi | xj][k][128]
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)&smem[0][0];
  const int tid = threadIdx.x, lane = tid & 63, tx = tid & 15, ty = tid >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const double* h = H + (size_t)b * (d + 2);
  const double* Xs_b = Xs + (size_t)b * dpad * npad;
  double* out = Kbuf + (size_t)b * ld * ld;
  const int t0 = g * KB2_TPW, t1 = min(t0 + KB2_TPW, ntiles), nkb = dpad / KB_DK;
  const int nchunks = (t1 - t0) * nkb;
  auto decode = [&](int t, int& ti, int& tj) {
    if (full == 2) {
      ti = t;
      tj = 0;
    } else if (full) {
      ti = t / nblk;
      tj = t - ti * nblk;
    } else {
      bgp_tri_decode(t, ti, tj);
    }
  };
  int ti, tj;
  decode(t0, ti, tj);
  kb2_issue(Xs_b, npad, ti, tj, 0, lds0, w, lane);
  double acc[KB2_R][8];
  for (int c = 0; c < nchunks; c++) {
    const int t = t0 + c / nkb, kb = c - (c / nkb) * nkb;
    decode(t, ti, tj);
    if (kb == 0) {
#pragma unroll
      for (int r = 0; r < KB2_R; r++)
#pragma unroll
        for (int cc = 0; cc < 8; cc++) acc[r][cc] = 0.0;
      if ((ti == tj || full == 2) && tid < 128)
        yw[(size_t)b * ld + ti * 128 + tid] = y[ti * 128 + tid];  // working right-hand side
    }
    S4_WAIT_VM0();                 // this wave's share of chunk c has landed
    __builtin_amdgcn_s_barrier();  // ... everybody's has; everybody finished reading chunk c-1
    if (c + 1 < nchunks) {
      const int tn_ = t0 + (c + 1) / nkb, kbn = (c + 1) - ((c + 1) / nkb) * nkb;
      int tin, tjn;
      decode(tn_, tin, tjn);
      kb2_issue(Xs_b, npad, tin, tjn, kbn, lds0 + (unsigned)(((c + 1) & 1) * 2 * KB_DK * 1024), w, lane);
    }
    const double* xi = &smem[c & 1][0];
    const double* xj = &smem[c & 1][KB_DK * 128];
#pragma unroll 4
    for (int k = 0; k < KB_DK; k++) {
      double a[KB2_R], bb[8];
#pragma unroll
      for (int r = 0; r < KB2_R; r++) a[r] = xi[k * 128 + ty + (128 / KB2_R) * r];
#pragma unroll
      for (int cc = 0; cc < 8; cc++) bb[cc] = xj[k * 128 + tx + 16 * cc];
      // four differences, then their four squares: a subtract never feeds the very next instruction
#pragma unroll
      for (int r = 0; r < KB2_R; r++) {
#pragma unroll
        for (int c4 = 0; c4 < 8; c4 += 4) {
          double df[4];
#pragma unroll
          for (int cc = 0; cc < 4; cc++) df[cc] = a[r] - bb[c4 + cc];
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
          for (int cc = 0; cc < 4; cc++) acc[r][c4 + cc] = fma(df[cc], df[cc], acc[r][c4 + cc]);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
      }
    }
    if (kb == nkb - 1)
      kb_epilogue<1, STAT, FORM, KB2_R>(acc, n, n, d, h, use_alpha ? alpha : nullptr, ti * 128, tj * 128, out, (size_t)ld, npad,
                                 npad, tx, ty);
  }
}

static int ensure_xs(bgp_ctx* ctx, int dpad) {
  const size_t need = (size_t)ctx->max_batch * dpad * ctx->npad;
  if (need > ctx->cap_xs) {
    if (ctx->dXs) {
      (void)hipDeviceSynchronize();  // (another walker group's launches may still read the old buffer)
      (void)hipFree(ctx->dXs);
    }
    ctx->dXs = nullptr;
    ctx->cap_xs = 0;
    BGP_HIP(hipMalloc(&ctx->dXs, need * sizeof(double)));
    ctx->cap_xs = need;
  }
  return BGP_OK;
}

int bgp_launch_kbuild(bgp_ctx* ctx, int B, int full_square, int augmented, int use_alpha) {
  return bgp_launch_kbuild_slice(ctx, 0, B, ctx->stream, full_square, augmented, use_alpha);
}

int bgp_launch_kbuild_slice(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented,
                            int use_alpha) {
  return bgp_launch_kbuild_x(ctx, off, B, st, full_square, augmented, use_alpha, ctx->dXeff, 0);
}

int bgp_launch_kbuild_x(bgp_ctx* ctx, int off, int B, hipStream_t st, int full_square, int augmented, int use_alpha,
                        const double* dXb, size_t xstride) {
  const int nblk = ctx->nblk, npad = ctx->npad, d = ctx->d;
  const size_t ldm = augmented ? 2 * (size_t)npad : (size_t)npad;
  // full_square == 2: block column 0 only (the trailing update generates the other blocks at first touch, S4GenF in bgp_s4.h,
  // from the scaled inputs the pipelined build leaves behind: always that build)
  const int ntiles = (full_square == 2) ? nblk : full_square ? nblk * nblk : nblk * (nblk + 1) / 2;
  const int B8 = 8 * ((B + 7) / 8);
  double* dKo = ctx->dK + (size_t)off * ldm * ldm;
  const double* dH = ctx->dh + (size_t)off * (d + 2);
  double* dywo = ctx->dyw + (size_t)off * ldm;
  bgp_tbegin(ctx, 0, st);
  // the pipelined build pays a second (tiny) launch and groups KB2_TPW tiles per workgroup: below ~2000 tiles the
  // plain kernel is faster (n = 1024 x 32 walkers: 0.069 vs 0.10 ms)
  if (full_square != 2 && B8 * ntiles < 2048) {
    KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
                hipLaunchKernelGGL((kbuild_gram_kernel<S, F>), dim3(B8 * ntiles), dim3(256), 0, st, dXb, ctx->dalpha, dH,
                                   dKo, ctx->dy, dywo, ctx->n, d, npad, nblk, B, full_square, (int)ldm, use_alpha,
                                   xstride));
  } else {
    // scaled inputs of the walkers of this batch slice, k-major (grown on demand; one slot per walker of max_batch)
    const int dpad = ((d + KB_DK - 1) / KB_DK) * KB_DK;
    {
      const int rcx = ensure_xs(ctx, dpad);
      if (rcx) return rcx;
    }
    double* dXs = ctx->dXs + (size_t)off * dpad * npad;
    hipLaunchKernelGGL(xscale_kernel, dim3(64, B), dim3(256), 0, st, dXb, xstride, dH, dXs, ctx->n, d, npad, dpad);
    const int ngroups = (ntiles + KB2_TPW - 1) / KB2_TPW;
    KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
                hipLaunchKernelGGL((kbuild2_kernel<S, F>), dim3(B8 * ngroups), dim3(64 * KB2_WAVES), 0, st, dXs, ctx->dalpha, dH, dKo,
                                   ctx->dy, dywo, ctx->n, d, npad, dpad, nblk, B, full_square, (int)ldm, use_alpha));
  }
  bgp_tend(ctx, st);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

// Gram generation inside the trailing update (bgp_launch_cholesky_slice's `gen`): where the pipelined build would run anyway
// (the same threshold), on matrices of at least two block columns with at most 16 input dimensions (one staging pass; at d = 32
// the generator measured +1.2 % / -0.1 % at n = 4096 x 8 / 16 matrices, against -2 .. -4.5 % at d <= 16 from 32 matrices on:
// tools/gen_shapes_probe.py); BGP_SYRK_GEN=0 switches it off (A/B measurements).
int bgp_lml_gen_eligible(const bgp_ctx* ctx, int B) {
  const char* e = getenv("BGP_SYRK_GEN");  // (read per call: the tests flip it inside one process)
  const int on = (e && e[0] == '0') ? 0 : 1;
  const int B8 = 8 * ((B + 7) / 8);
  return on && ctx->d <= KB_DK && ctx->nblk >= 2 && B8 * (ctx->nblk * (ctx->nblk + 1) / 2) >= 2048;
}

// ... and what the generator reads: the scaled inputs of the batch slice at `off` (written by the build of block column 0)
int bgp_lml_gen_args(const bgp_ctx* ctx, int off, S4Gen* out) {
  const int dpad = ((ctx->d + KB_DK - 1) / KB_DK) * KB_DK;
  if (!ctx->dXs || (size_t)ctx->max_batch * dpad * ctx->npad > ctx->cap_xs) {
    bgp_set_error("bgp_lml_gen_args: the scaled inputs have not been built");
    return BGP_ERR_STATE;
  }
  out->Xs = ctx->dXs + (size_t)off * dpad * ctx->npad;
  out->H = ctx->dh + (size_t)off * (ctx->d + 2);
  out->alpha = ctx->dalpha;
  out->n = ctx->n;
  out->d = ctx->d;
  out->dpad = dpad;
  out->npad = ctx->npad;
  return BGP_OK;
}

int bgp_launch_kcross(bgp_ctx* ctx, const double* dh_b, int m, const double* dXq, int nx, const double* dXt,
                      double* dout, int ldo, int /*unused*/) {
  return bgp_launch_kcross_batch(ctx, 1, dh_b, m, dXq, nx, dXt, dout, ldo, 0);
}

int bgp_launch_kcross_batch(bgp_ctx* ctx, int nb, const double* dH, int m, const double* dXq, int nx, const double* dXt,
                            double* dout, int ldo, size_t ostride) {
  return bgp_launch_kcross_matvec(ctx, nb, dH, m, dXq, nx, dXt, dout, ldo, ostride, nullptr, 0, nullptr);
}

// ... and, with vec, the column-tile partials of  out_b vec_b  into dpart (nb x tiles_j x pad128(m))
int bgp_launch_kcross_matvec(bgp_ctx* ctx, int nb, const double* dH, int m, const double* dXq, int nx, const double* dXt,
                             double* dout, int ldo, size_t ostride, const double* vec, size_t svec, double* dpart) {
  const int tiles_i = (m + 127) / 128, tiles_j = (nx + 127) / 128;
  KB_DISPATCH(ctx->ks.stationary, ctx->ks.form,
              hipLaunchKernelGGL((kbuild_cross_kernel<S, F>), dim3(tiles_i * tiles_j, nb), dim3(256), 0, ctx->stream, dXq,
                                 m, dXt, nx, ctx->d, dH, dout, ldo, tiles_j, ostride, vec, svec, dpart, tiles_i * 128));
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}
