// Posterior build, predict, LML gradient, PVRS and sample_y.
//
// Posterior build (the BayesGPR.theta setter, bask/bayesgpr.py:200-217): the batched Cholesky of
// bgp_chol.hip is run on the augmented matrix [[K, .], [I, 0]] for nblk steps, which leaves
//   top-left      L                      -> BayesGPR.L_
//   bottom-right  -K^-1 (Schur complement)-> BayesGPR.K_inv_   (the reference forms the explicit
//                                            inverse too: L_inv.dot(L_inv.T), :207-208)
//   rhs, lower    -alpha = -(K^-1 y)      -> BayesGPR.alpha_
// K^-1 and alpha of every posterior stay resident in HBM for the predict / pvrs / gradient calls.
//
// Predict (bask/bayesgpr.py:622-635 -> skopt predict), per resident posterior:
//   K_* = k(Xq, X)  (tiled cross-kernel build);  mean = K_* alpha;
//   var = k_** - rowsum((K_* K^-1) o K_*)   -- one NT tile GEMM on the fp64 MFMA with the row-dot
//   fused into its epilogue (K^-1 symmetric, so K_* K^-1 = K_* (K^-1)^T is an NT product);
//   cov = K_** - (K_* K^-1) K_*^T.
#include "bgp_common.h"
#include "bgp_device.h"
#include "bgp_gemm.h"

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__global__ void aug_init_kernel(double* __restrict__ Kbuf, double* __restrict__ yw, int npad, int B) {
  // identity into the bottom-left block of every augmented matrix, zero the lower half of the rhs
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (i >= npad || b >= B) return;
  const size_t ld = 2 * (size_t)npad;
  Kbuf[(size_t)b * ld * ld + (size_t)(npad + i) * ld + i] = 1.0;
  yw[(size_t)b * ld + npad + i] = 0.0;
}

// K^-1 (full symmetric npad x npad) and alpha out of the augmented workspace.
__global__ void __launch_bounds__(256) extract_kinv_kernel(const double* __restrict__ Kbuf,
                                                            const double* __restrict__ yw,
                                                            double* __restrict__ Kinv, double* __restrict__ alpha,
                                                            int npad, int boff) {
  const int b = blockIdx.y;
  const size_t ld = 2 * (size_t)npad;
  const double* M = Kbuf + (size_t)b * ld * ld;
  double* out = Kinv + (size_t)(boff + b) * npad * npad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)npad * npad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / npad), j = (int)(idx - (size_t)i * npad);
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    out[idx] = -M[(size_t)(npad + hi) * ld + npad + lo];
  }
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < npad; i += blockDim.x)
      alpha[(size_t)(boff + b) * npad + i] = -yw[(size_t)b * ld + npad + i];
}

// compact n x n lower factor (zeros above the diagonal) of matrix b into scratch
__global__ void extract_L_kernel(const double* __restrict__ Kbuf, double* __restrict__ out, int n, int ld,
                                 size_t mstride, int b) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)n * n;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx - (size_t)i * n);
    out[idx] = (j <= i) ? Kbuf[(size_t)b * mstride + (size_t)i * ld + j] : 0.0;
  }
}

// mean_i = sum_j Ks[i][j] alpha[j]   (one wave per row)
__global__ void __launch_bounds__(256) matvec_rows_kernel(const double* __restrict__ Ks, int lds,
                                                           const double* __restrict__ v, int n, int m,
                                                           double* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= m) return;
  double s = 0.0;
  for (int j = lane; j < n; j += 64) s += Ks[(size_t)row * lds + j] * v[j];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[row] = s;
}

// var_i = max(0, diag - q_i)
__global__ void finish_var_kernel(const double* __restrict__ q, double diag, int m, double* __restrict__ var) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  double v = diag - q[i];
  var[i] = v < 0.0 ? 0.0 : v;
}

// ------------------------------------------------------------------------------------------
// General NT tile GEMM: C (M x N) = A (M x K) * B (N x K)^T, all row-major, M, N multiples of 128,
// K a multiple of 32 (buffers are zero padded).  Epilogues:
//   EPI 0: C = acc                         (ldc)
//   EPI 1: rowdot[tj][i] = sum_{j in column tile tj} acc[i][j] * E[i][j]  (E: M x N, lde) -- predictive
//          variance; the launcher then sums the column tiles in a fixed order (no fp atomics to global)
//   EPI 2: C = E - acc                                    -- predictive covariance
// ------------------------------------------------------------------------------------------
template <int EPI>
__global__ void __launch_bounds__(256) gemm_nt_kernel(const double* __restrict__ A, int lda,
                                                       const double* __restrict__ Bm, int ldb, int K,
                                                       double* __restrict__ C, int ldc,
                                                       const double* __restrict__ E, int lde,
                                                       double* __restrict__ rowdot, int tiles_n) {
  const int ti = blockIdx.x / tiles_n, tj = blockIdx.x - ti * tiles_n;
  __shared__ GemmSmem sm;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
  const double* At = A + (size_t)(ti * 128) * lda;
  const double* Bt = Bm + (size_t)(tj * 128) * ldb;
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};
  if (EPI == 1 && tid < 128) sm.ypart[tid] = 0.0;
  for (int k0 = 0; k0 < K; k0 += GK_KC) {
    __syncthreads();
    gk_load_chunk(sm.A, At + k0, (size_t)lda, tid);
    gk_load_chunk(sm.B, Bt + k0, (size_t)ldb, tid);
    __syncthreads();
    gk_mma_chunk<0, 0>(sm.A, sm.B, acc, wr, wc, lane, k0);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = GK_ROW(wr, i, lane, r);
      const size_t grow = (size_t)(ti * 128 + row);
      double part = 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const size_t gcol = (size_t)(tj * 128 + GK_COL(wc, j, lane));
        const double x = acc[i][j][r];
        if (EPI == 0) C[grow * ldc + gcol] = x;
        if (EPI == 1) part += x * E[grow * lde + gcol];
        if (EPI == 2) C[grow * ldc + gcol] = E[grow * lde + gcol] - x;
      }
      if (EPI == 1) {
        part += __shfl_xor(part, 1);
        part += __shfl_xor(part, 2);
        part += __shfl_xor(part, 4);
        part += __shfl_xor(part, 8);
        if ((lane & 15) == 0) atomicAdd(&sm.ypart[row], part);
      }
    }
  }
  if (EPI == 1) {
    __syncthreads();
    if (tid < 128) rowdot[(size_t)tj * ((size_t)gridDim.x / tiles_n * 128) + ti * 128 + tid] = sm.ypart[tid];
  }
}

__global__ void rowdot_reduce_kernel(const double* __restrict__ part, int tn, int M, double* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  double s = 0.0;
  for (int t = 0; t < tn; t++) s += part[(size_t)t * M + i];
  out[i] = s;
}

// column-tile partials of the EPI-1 row dots (own buffer: the callers' scratch layouts stay as they are)
static int ensure_rowpart(bgp_ctx* c, size_t doubles) {
  if (doubles > c->cap_rowpart) {
    if (c->drowpart) (void)hipFree(c->drowpart);
    c->drowpart = nullptr;
    c->cap_rowpart = 0;
    BGP_HIP(hipMalloc(&c->drowpart, doubles * sizeof(double)));
    c->cap_rowpart = doubles;
  }
  return BGP_OK;
}

template <int EPI>
static int launch_gemm_nt(bgp_ctx* c, const double* A, int lda, const double* Bm, int ldb, int M, int N, int K,
                          double* C, int ldc, const double* E, int lde, double* rowdot) {
  const int tm = M / 128, tn = N / 128;
  double* rd = rowdot;
  if (EPI == 1) {
    int rc = ensure_rowpart(c, (size_t)tn * M);
    if (rc) return rc;
    rd = c->drowpart;
  }
  hipLaunchKernelGGL(gemm_nt_kernel<EPI>, dim3(tm * tn), dim3(256), 0, c->stream, A, lda, Bm, ldb, K, C, ldc, E, lde,
                     rd, tn);
  BGP_HIP(hipGetLastError());
  if (EPI == 1) {
    hipLaunchKernelGGL(rowdot_reduce_kernel, dim3((M + 255) / 256), dim3(256), 0, c->stream, rd, tn, M, rowdot);
    BGP_HIP(hipGetLastError());
  }
  return BGP_OK;
}

__global__ void add_diag_kernel(double* __restrict__ C, int ld, int m, double v);

static inline int pad128(int v) { return ((v + 127) / 128) * 128; }

// ------------------------------------------------------------------------------------------
// posterior build
// ------------------------------------------------------------------------------------------
static int ensure_resident(bgp_ctx* c, int B) {
  const size_t need = (size_t)B * c->npad * c->npad;
  if (need > c->cap_kinv) {
    if (c->dKinv) (void)hipFree(c->dKinv);
    if (c->dalpha_sol) (void)hipFree(c->dalpha_sol);
    c->dKinv = c->dalpha_sol = nullptr;
    c->cap_kinv = 0;
    BGP_HIP(hipMalloc(&c->dKinv, need * sizeof(double)));
    BGP_HIP(hipMalloc(&c->dalpha_sol, (size_t)B * c->npad * sizeof(double)));
    c->cap_kinv = need;
  }
  return BGP_OK;
}

int bgp_posterior_build(bgp_ctx* c, int B, const double* h, int use_alpha, double* L, double* alpha, double* K_inv,
                        double* lml, int* status) {
  const int npad = c->npad, n = c->n;
  const size_t p = c->d + 2;
  const size_t ld = 2 * (size_t)npad;
  int rc = ensure_resident(c, B);
  if (rc) return rc;
  // augmented matrices are 4x the LML workspace per item
  int chunk = (int)(c->cap_mat / (ld * ld));
  if (chunk < 1) {
    rc = bgp_grow_workspace(c, ld * ld);
    if (rc) return rc;
    chunk = 1;
  }
  chunk = std::min(chunk, c->max_batch);
  for (int off = 0; off < B; off += chunk) {
    const int nb = std::min(chunk, B - off);
    BGP_HIP(hipMemcpyAsync(c->dh, h + (size_t)off * p, nb * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BGP_HIP(hipMemsetAsync(c->dstatus, 0, nb * sizeof(int), c->stream));
    // zero the bottom halves ([I | 0] rows), then K into the top-left, identity, rhs
    for (int b = 0; b < nb; b++)
      BGP_HIP(hipMemsetAsync(c->dK + (size_t)b * ld * ld + (size_t)npad * ld, 0, (size_t)npad * ld * sizeof(double),
                             c->stream));
    rc = bgp_launch_kbuild(c, nb, 0, 1, use_alpha);
    if (rc) return rc;
    hipLaunchKernelGGL(aug_init_kernel, dim3((npad + 255) / 256, nb), dim3(256), 0, c->stream, c->dK, c->dyw, npad, nb);
    rc = bgp_launch_cholesky(c, nb, 1);
    if (rc) return rc;
    hipLaunchKernelGGL(extract_kinv_kernel, dim3(256, nb), dim3(256), 0, c->stream, c->dK, c->dyw, c->dKinv,
                       c->dalpha_sol, npad, off);
    BGP_HIP(hipGetLastError());
    if (lml) BGP_HIP(hipMemcpyAsync(lml + off, c->dlml, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) BGP_HIP(hipMemcpyAsync(status + off, c->dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (alpha)
      BGP_HIP(hipMemcpy2DAsync(alpha + (size_t)off * n, (size_t)n * sizeof(double),
                               c->dalpha_sol + (size_t)off * npad, (size_t)npad * sizeof(double),
                               (size_t)n * sizeof(double), nb, hipMemcpyDeviceToHost, c->stream));
    if (K_inv)
      for (int b = 0; b < nb; b++)
        BGP_HIP(hipMemcpy2DAsync(K_inv + (size_t)(off + b) * n * n, (size_t)n * sizeof(double),
                                 c->dKinv + (size_t)(off + b) * npad * npad, (size_t)npad * sizeof(double),
                                 (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost, c->stream));
    if (L) {
      rc = bgp_ensure_scratch(c, (size_t)n * n);
      if (rc) return rc;
      for (int b = 0; b < nb; b++) {
        hipLaunchKernelGGL(extract_L_kernel, dim3(512), dim3(256), 0, c->stream, c->dK, c->dscratch, n, (int)ld,
                           ld * ld, b);
        BGP_HIP(hipMemcpyAsync(L + (size_t)(off + b) * n * n, c->dscratch, (size_t)n * n * sizeof(double),
                               hipMemcpyDeviceToHost, c->stream));
        BGP_HIP(hipStreamSynchronize(c->stream));
      }
    }
    BGP_HIP(hipStreamSynchronize(c->stream));
  }
  c->post_B = B;
  return BGP_OK;
}

extern "C" int bgp_posterior_batch(bgp_ctx* c, int B, const double* h, double* L, double* alpha, double* K_inv,
                                   double* lml, int* status) {
  if (!c || !h || B <= 0) {
    bgp_set_error("bgp_posterior_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  return bgp_posterior_build(c, B, h, 1, L, alpha, K_inv, lml, status);
}

// ------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------
// scratch layout helper
struct Scratch {
  double* base;
  size_t used;
  double* take(size_t n) {
    double* p = base + used;
    used += (n + 1) & ~(size_t)1;  // keep 16-byte alignment
    return p;
  }
};

static double kernel_diag_value(const bgp_ctx* c, const double* hk) {
  // kernel_.diag(X) incl. the white level: sklearn/kernels.py:868-884, 968-984
  const double cst = std::exp(hk[0]), s2 = std::exp(hk[c->d + 1]);
  const double base = (c->ks.form == BGP_FORM_PRODUCT) ? cst * 1.0 : cst + 1.0;
  return base + s2;
}

extern "C" int bgp_predict_batch(bgp_ctx* c, int B, const double* h_kernel, int m, const double* Xq, double* mean,
                                 double* var, double* cov) {
  if (!c || !h_kernel || !Xq || !mean || !var || m <= 0 || B <= 0) {
    bgp_set_error("bgp_predict_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  if (B > c->post_B) {
    bgp_set_error("bgp_predict_batch: %d posteriors requested but %d resident (call bgp_posterior_batch first)", B,
                  c->post_B);
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m);
  const size_t p = d + 2;
  size_t need = (size_t)m * d + 2 + p + (size_t)mpad * npad + 2 * (size_t)mpad + 64;
  if (cov) need += (size_t)mpad * npad + 2 * (size_t)mpad * mpad;
  int rc = bgp_ensure_scratch(c, need);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dXq = s.take((size_t)m * d);
  double* dhk = s.take(p);
  double* dKs = s.take((size_t)mpad * npad);
  double* dq = s.take(mpad);
  double* dout = s.take(mpad);
  double *dP = nullptr, *dKss = nullptr, *dCov = nullptr;
  if (cov) {
    dP = s.take((size_t)mpad * npad);
    dKss = s.take((size_t)mpad * mpad);
    dCov = s.take((size_t)mpad * mpad);
  }
  BGP_HIP(hipMemcpyAsync(dXq, Xq, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->has_warp) {  // BayesGPR.predict warps the query points with the current warpers (bask/bayesgpr.py:630-632)
    rc = bgp_launch_warp(c, c->stream, dXq, c->dwarp, dXq, m, 1, 0);
    if (rc) return rc;
  }
  for (int b = 0; b < B; b++) {
    const double* hk = h_kernel + (size_t)b * p;
    BGP_HIP(hipMemcpyAsync(dhk, hk, p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BGP_HIP(hipMemsetAsync(dKs, 0, (size_t)mpad * npad * sizeof(double), c->stream));
    BGP_HIP(hipMemsetAsync(dq, 0, (size_t)mpad * sizeof(double), c->stream));
    rc = bgp_launch_kcross(c, dhk, m, dXq, n, c->dXeff, dKs, npad, 0);
    if (rc) return rc;
    const double* Kinv = c->dKinv + (size_t)b * npad * npad;
    const double* al = c->dalpha_sol + (size_t)b * npad;
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((m + 3) / 4), dim3(256), 0, c->stream, dKs, npad, al, n, m, dout);
    BGP_HIP(hipMemcpyAsync(mean + (size_t)b * m, dout, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (!cov) {
      rc = launch_gemm_nt<1>(c, dKs, npad, Kinv, npad, mpad, npad, npad, nullptr, 0, dKs, npad, dq);
      if (rc) return rc;
    } else {
      rc = launch_gemm_nt<0>(c, dKs, npad, Kinv, npad, mpad, npad, npad, dP, npad, nullptr, 0, nullptr);
      if (rc) return rc;
      // K_** (no white noise off the diagonal; diag gets c(+1)+s2 like kernel_(X))
      BGP_HIP(hipMemsetAsync(dKss, 0, (size_t)mpad * mpad * sizeof(double), c->stream));
      rc = bgp_launch_kcross(c, dhk, m, dXq, m, dXq, dKss, mpad, 0);
      if (rc) return rc;
      hipLaunchKernelGGL(add_diag_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dKss, mpad, m,
                         std::exp(hk[d + 1]));
      rc = launch_gemm_nt<2>(c, dP, npad, dKs, npad, mpad, mpad, npad, dCov, mpad, dKss, mpad, nullptr);
      if (rc) return rc;
      BGP_HIP(hipMemcpy2DAsync(cov + (size_t)b * m * m, (size_t)m * sizeof(double), dCov,
                               (size_t)mpad * sizeof(double), (size_t)m * sizeof(double), m, hipMemcpyDeviceToHost,
                               c->stream));
      // variance from the covariance diagonal is not needed by the callers of return_cov
      rc = launch_gemm_nt<1>(c, dKs, npad, Kinv, npad, mpad, npad, npad, nullptr, 0, dKs, npad, dq);
      if (rc) return rc;
    }
    hipLaunchKernelGGL(finish_var_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dq, kernel_diag_value(c, hk),
                       m, dout);
    BGP_HIP(hipMemcpyAsync(var + (size_t)b * m, dout, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(hipStreamSynchronize(c->stream));
  }
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// LML gradient:  g_k = 1/2 sum_ij (alpha_i alpha_j - Kinv_ij) dK_ij/dh_k   (sklearn/_gpr.py:615-647)
// One workgroup per lower-triangular 128x128 tile (off-diagonal tiles count twice); per-dimension
// sums are reduced in the workgroup and stored to gpart[b][tile][k]; grad_reduce_kernel adds the tiles in a
// fixed order (bitwise reproducible gradients: no floating-point atomics).
// ------------------------------------------------------------------------------------------
#define GR_DK 16
__global__ void __launch_bounds__(256) lml_grad_kernel(const double* __restrict__ X,
                                                        const double* __restrict__ H,
                                                        const double* __restrict__ Kinv,
                                                        const double* __restrict__ alpha_sol,
                                                        double* __restrict__ gpart, int n, int d, int npad, int nblk,
                                                        int form, int stat, int B) {
  const int ntiles = nblk * (nblk + 1) / 2;
  int b, t;
  bgp_map_block(blockIdx.x, ntiles, B, b, t);
  if (b >= B) return;
  double* gp = gpart + ((size_t)b * ntiles + t) * (d + 2);
  int ti, tj;
  bgp_tri_decode(t, ti, tj);
  __shared__ double xi[GR_DK][BGP_TILE_LD];
  __shared__ double xj[GR_DK][BGP_TILE_LD];
  __shared__ double ell[GR_DK];
  __shared__ double red[4][GR_DK + 2];
  const double* h = H + (size_t)b * (d + 2);
  const double* Ki = Kinv + (size_t)b * npad * npad;
  const double* al = alpha_sol + (size_t)b * npad;
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4, lane = tid & 63, w = tid >> 6;
  const int i0 = ti * 128, j0 = tj * 128;
  const double wtile = (ti == tj) ? 1.0 : 2.0;
  double F[8][8];
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 8; c++) F[r][c] = 0.0;
  // pass 1: squared scaled distances
  for (int k0 = 0; k0 < d; k0 += GR_DK) {
    const int kc = min(GR_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    for (int idx = tid; idx < kc * 128; idx += 256) {
      int row = idx / kc, k = idx - row * kc;
      int gi = i0 + row, gj = j0 + row;
      xi[k][row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / ell[k] : 0.0;
      xj[k][row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / ell[k] : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < kc; k++) {
      double a[8], bb[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) bb[c] = xj[k][tx + 16 * c];
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          double df = a[r] - bb[c];
          F[r][c] += df * df;
        }
    }
  }
  // F_ij = W_ij * f(r_ij);   constant / noise terms on the fly
  const double cst = exp(h[0]), s2 = exp(h[d + 1]);
  const double cf = (form == BGP_FORM_PRODUCT) ? cst : 1.0;
  double g_const = 0.0, g_noise = 0.0;
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int gi = i0 + ty + 16 * r;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      const int gj = j0 + tx + 16 * c;
      double f = 0.0;
      if (gi < n && gj < n) {
        const double Wij = al[gi] * al[gj] - Ki[(size_t)gi * npad + gj];
        if (gi == gj) {
          g_const += Wij * ((form == BGP_FORM_PRODUCT) ? cst * 1.0 : cst);
          g_noise += Wij * s2;
        } else {
          const double r2 = F[r][c];
          double S, fr;
          if (stat == BGP_RBF) {
            S = exp(-0.5 * r2);
            fr = S;
          } else if (stat == BGP_MATERN12) {
            const double rr = sqrt(r2);
            S = exp(-rr);
            fr = (rr > 0.0) ? S / rr : 0.0;
          } else if (stat == BGP_MATERN32) {
            const double tt = sqrt(r2) * 1.7320508075688772;
            const double e = exp(-tt);
            S = (1.0 + tt) * e;
            fr = 3.0 * e;
          } else {
            const double tt = sqrt(r2) * 2.23606797749979;
            const double e = exp(-tt);
            S = (1.0 + tt + tt * tt / 3.0) * e;
            fr = (5.0 / 3.0) * (tt + 1.0) * e;
          }
          g_const += wtile * Wij * ((form == BGP_FORM_PRODUCT) ? cst * S : cst);
          f = wtile * Wij * cf * fr;
        }
      }
      F[r][c] = f;
    }
  }
  // pass 2: per-dimension sums  sum_ij F_ij (x_ik - x_jk)^2 / l_k^2
  for (int k0 = 0; k0 < d; k0 += GR_DK) {
    const int kc = min(GR_DK, d - k0);
    __syncthreads();
    if (tid < kc) ell[tid] = exp(h[1 + k0 + tid]);
    __syncthreads();
    if (d > GR_DK) {  // tiles still staged from pass 1 when d fits in one chunk
      for (int idx = tid; idx < kc * 128; idx += 256) {
        int row = idx / kc, k = idx - row * kc;
        int gi = i0 + row, gj = j0 + row;
        xi[k][row] = (gi < n) ? X[(size_t)gi * d + k0 + k] / ell[k] : 0.0;
        xj[k][row] = (gj < n) ? X[(size_t)gj * d + k0 + k] / ell[k] : 0.0;
      }
      __syncthreads();
    }
    for (int k = 0; k < kc; k++) {
      double a[8], bb[8];
#pragma unroll
      for (int r = 0; r < 8; r++) a[r] = xi[k][ty + 16 * r];
#pragma unroll
      for (int c = 0; c < 8; c++) bb[c] = xj[k][tx + 16 * c];
      double sk = 0.0;
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int c = 0; c < 8; c++) {
          double df = a[r] - bb[c];
          sk += F[r][c] * df * df;
        }
      for (int o = 32; o > 0; o >>= 1) sk += __shfl_xor(sk, o);
      if (lane == 0) red[w][k] = sk;
    }
    __syncthreads();
    if (tid < kc) gp[1 + k0 + tid] = 0.5 * (red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    g_const += __shfl_xor(g_const, o);
    g_noise += __shfl_xor(g_noise, o);
  }
  __syncthreads();
  if (lane == 0) {
    red[w][0] = g_const;
    red[w][1] = g_noise;
  }
  __syncthreads();
  if (tid == 0) {
    gp[0] = 0.5 * (red[0][0] + red[1][0] + red[2][0] + red[3][0]);
    gp[d + 1] = 0.5 * (red[0][1] + red[1][1] + red[2][1] + red[3][1]);
  }
}

__global__ void grad_reduce_kernel(const double* __restrict__ gpart, double* __restrict__ grad, int ntiles, int p) {
  const int b = blockIdx.x, k = threadIdx.x;
  if (k >= p) return;
  double s = 0.0;
  for (int t = 0; t < ntiles; t++) s += gpart[((size_t)b * ntiles + t) * p + k];
  grad[(size_t)b * p + k] = s;
}

extern "C" int bgp_lml_grad_batch(bgp_ctx* c, int B, const double* h, double* lml, double* grad, int* status) {
  if (!c || !h || !lml || !grad || B <= 0) {
    bgp_set_error("bgp_lml_grad_batch: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  const size_t p = c->d + 2;
  std::vector<int> st(B, 0);
  int rc = bgp_posterior_build(c, B, h, 1, nullptr, nullptr, nullptr, lml, st.data());
  if (rc) return rc;
  c->post_B = 0;  // the resident K^-1 belong to a gradient evaluation, not to a posterior
  const int ntiles = c->nblk * (c->nblk + 1) / 2;
  rc = bgp_ensure_scratch(c, (size_t)B * p * (2 + ntiles));
  if (rc) return rc;
  double* dgrad = c->dscratch;
  double* dH = c->dscratch + (size_t)B * p;
  double* dgpart = c->dscratch + (size_t)B * p * 2;
  BGP_HIP(hipMemsetAsync(dgpart, 0, (size_t)B * p * ntiles * sizeof(double), c->stream));
  BGP_HIP(hipMemcpyAsync(dH, h, (size_t)B * p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(lml_grad_kernel, dim3(8 * ((B + 7) / 8) * ntiles), dim3(256), 0, c->stream, c->dXeff, dH, c->dKinv,
                     c->dalpha_sol, dgpart, c->n, c->d, c->npad, c->nblk, c->ks.form, c->ks.stationary, B);
  BGP_HIP(hipGetLastError());
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(B), dim3(((int)p + 63) / 64 * 64), 0, c->stream, dgpart, dgrad, ntiles,
                     (int)p);
  BGP_HIP(hipGetLastError());
  BGP_HIP(hipMemcpyAsync(grad, dgrad, (size_t)B * p * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(hipStreamSynchronize(c->stream));
  for (int b = 0; b < B; b++) {
    if (st[b] != 0)
      for (size_t k = 0; k < p; k++) grad[(size_t)b * p + k] = 0.0;  // sklearn/_gpr.py:589: (-inf, zeros)
    if (status) status[b] = st[b];
  }
  return BGP_OK;
}

// ------------------------------------------------------------------------------------------
// PVRS (bask/acquisition.py:328-338) through the bordered-inverse identity:
//   covs[i] = sum_t [ k_t^T Kinv k_t + (k(x_t, x_i) - k_i^T Kinv k_t)^2 / (kappa - k_i^T Kinv k_i) ]
// ------------------------------------------------------------------------------------------
__global__ void pvrs_combine_kernel(const double* __restrict__ G, int ldg, const double* __restrict__ Kti, int ldk,
                                    const double* __restrict__ u, const double* __restrict__ st, double kappa, int m,
                                    int T, double* __restrict__ covs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const double lam2 = kappa - u[i];
  double s = 0.0;
  for (int t = 0; t < T; t++) {
    const double e = Kti[(size_t)i * ldk + t] - G[(size_t)i * ldg + t];
    s += st[t] + e * e / lam2;
  }
  covs[i] = s;
}

extern "C" int bgp_pvrs(bgp_ctx* c, const double* h_kernel, int m, const double* Xcand, int T,
                        const double* Xthompson, double* covs) {
  if (!c || !h_kernel || !Xcand || !Xthompson || !covs || m <= 0 || T <= 0) {
    bgp_set_error("bgp_pvrs: bad argument");
    return BGP_ERR_INVALID;
  }
  if (c->post_B < 1) {
    bgp_set_error("bgp_pvrs: no resident posterior (call bgp_posterior_batch / bgp_pvrs_prepare first)");
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m), Tpad = pad128(T);
  const size_t p = d + 2;
  size_t need = (size_t)m * d + (size_t)T * d + p + 128 + (size_t)mpad * npad + 2 * (size_t)Tpad * npad +
                2 * (size_t)mpad * Tpad + 2 * (size_t)mpad + 2 * (size_t)Tpad;
  int rc = bgp_ensure_scratch(c, need);
  if (rc) return rc;
  Scratch s{c->dscratch, 0};
  double* dXc = s.take((size_t)m * d);
  double* dXt = s.take((size_t)T * d);
  double* dhk = s.take(p);
  double* dKc = s.take((size_t)mpad * npad);   // k(cand, train)
  double* dKT = s.take((size_t)Tpad * npad);   // k(thompson, train)
  double* dPT = s.take((size_t)Tpad * npad);   // K_T Kinv
  double* dG = s.take((size_t)mpad * Tpad);    // K_c Kinv K_T^T
  double* dKti = s.take((size_t)mpad * Tpad);  // k(cand, thompson)
  double* du = s.take(mpad);
  double* dcov = s.take(mpad);
  double* dst = s.take(Tpad);
  const double* Kinv = c->dKinv;
  BGP_HIP(hipMemcpyAsync(dXc, Xcand, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemcpyAsync(dXt, Xthompson, (size_t)T * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
  BGP_HIP(hipMemcpyAsync(dhk, h_kernel, p * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if (c->has_warp) {  // candidates and Thompson points are compared in the warped space (bask/acquisition.py:324-327)
    rc = bgp_launch_warp(c, c->stream, dXc, c->dwarp, dXc, m, 1, 0);
    if (rc) return rc;
    rc = bgp_launch_warp(c, c->stream, dXt, c->dwarp, dXt, T, 1, 0);
    if (rc) return rc;
  }
  BGP_HIP(hipMemsetAsync(dKc, 0, ((size_t)mpad * npad + 2 * (size_t)Tpad * npad + 2 * (size_t)mpad * Tpad +
                                   2 * (size_t)mpad + 2 * (size_t)Tpad + 16) * sizeof(double), c->stream));
  rc = bgp_launch_kcross(c, dhk, m, dXc, n, c->dXeff, dKc, npad, 0);
  if (rc) return rc;
  rc = bgp_launch_kcross(c, dhk, T, dXt, n, c->dXeff, dKT, npad, 0);
  if (rc) return rc;
  rc = bgp_launch_kcross(c, dhk, m, dXc, T, dXt, dKti, Tpad, 0);
  if (rc) return rc;
  // P_T = K_T Kinv ; s_t = rowsum(P_T o K_T) ; u_i = rowsum((K_c Kinv) o K_c) ; G = K_c P_T^T
  rc = launch_gemm_nt<0>(c, dKT, npad, Kinv, npad, Tpad, npad, npad, dPT, npad, nullptr, 0, nullptr);
  if (rc) return rc;
  rc = launch_gemm_nt<1>(c, dKT, npad, Kinv, npad, Tpad, npad, npad, nullptr, 0, dKT, npad, dst);
  if (rc) return rc;
  rc = launch_gemm_nt<1>(c, dKc, npad, Kinv, npad, mpad, npad, npad, nullptr, 0, dKc, npad, du);
  if (rc) return rc;
  rc = launch_gemm_nt<0>(c, dKc, npad, dPT, npad, mpad, Tpad, npad, dG, Tpad, nullptr, 0, nullptr);
  if (rc) return rc;
  hipLaunchKernelGGL(pvrs_combine_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dG, Tpad, dKti, Tpad, du,
                     dst, kernel_diag_value(c, h_kernel), m, T, dcov);
  BGP_HIP(hipGetLastError());
  BGP_HIP(hipMemcpyAsync(covs, dcov, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  BGP_HIP(hipStreamSynchronize(c->stream));
  return BGP_OK;
}

extern "C" int bgp_pvrs_prepare(bgp_ctx* c, const double* h_kernel, int has_alpha_vec, int* status) {
  // K_aug's leading block: kernel_(X_train) + alpha only when alpha is a vector
  // (bask/acquisition.py:332-333)
  if (!c || !h_kernel) {
    bgp_set_error("bgp_pvrs_prepare: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  c->post_B = 0;
  return bgp_posterior_build(c, 1, h_kernel, has_alpha_vec ? 1 : 0, nullptr, nullptr, nullptr, nullptr, status);
}

// ------------------------------------------------------------------------------------------
// sample_y:  f = mean + L_cov z  with  L_cov = chol(cov + jitter I)  (same batched Cholesky, B = 1)
// ------------------------------------------------------------------------------------------
__global__ void cov_prepare_kernel(double* __restrict__ C, int m, int mpad, double jitter) {
  // jitter on the diagonal, identity padding, zero rhs handled by the caller
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)mpad * mpad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / mpad), j = (int)(idx - (size_t)i * mpad);
    if (i >= m || j >= m)
      C[idx] = (i == j) ? 1.0 : 0.0;
    else if (i == j)
      C[idx] += jitter;
  }
}

__global__ void zero_upper_kernel(double* __restrict__ C, int mpad) {
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (size_t)mpad * mpad;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / mpad), j = (int)(idx - (size_t)i * mpad);
    if (j > i) C[idx] = 0.0;
  }
}

__global__ void add_diag_kernel(double* __restrict__ C, int ld, int m, double v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < m) C[(size_t)i * ld + i] += v;
}

__global__ void add_mean_rows_kernel(double* __restrict__ out, int ldo, const double* __restrict__ mean, int m,
                                     int rows) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (j < m && i < rows) out[(size_t)i * ldo + j] += mean[j];
}

static void free_child(bgp_ctx* w) {
  if (!w) return;
  if (w->dK) (void)hipFree(w->dK);
  if (w->dW) (void)hipFree(w->dW);
  if (w->dyw) (void)hipFree(w->dyw);
  if (w->dacc) (void)hipFree(w->dacc);
  if (w->dlml) (void)hipFree(w->dlml);
  if (w->dstatus) (void)hipFree(w->dstatus);
  delete w;
}

extern "C" int bgp_sample_y(bgp_ctx* c, int b, const double* h_kernel, int m, const double* Xq, int n_draws,
                            const double* z, double jitter, double* out) {
  if (!c || !h_kernel || !Xq || !z || !out || m <= 0 || n_draws <= 0 || b < 0) {
    bgp_set_error("bgp_sample_y: bad argument");
    return BGP_ERR_INVALID;
  }
  if (b >= c->post_B) {
    bgp_set_error("bgp_sample_y: posterior %d not resident (%d resident)", b, c->post_B);
    return BGP_ERR_STATE;
  }
  BGP_HIP(hipSetDevice(c->device));
  const int npad = c->npad, n = c->n, d = c->d, mpad = pad128(m), rpad = pad128(n_draws);
  const size_t p = d + 2;
  // child workspace for the m x m Cholesky (shares the stream); cached on the context and grown on demand
  // (an m = 10 000 candidate grid needs 0.8 GB: re-allocating it per call costs more than the factorisation)
  bgp_ctx* w = c->child;
  if (w && w->cap_n < (size_t)mpad) {
    free_child(w);
    w = c->child = nullptr;
  }
  int rc = BGP_OK;
  if (!w) {
    w = new bgp_ctx();
    w->device = c->device;
    w->stream = c->stream;
    w->d = d;
    w->max_batch = 1;
    w->two_panel = c->two_panel;
    w->panels = c->panels;
    if (hipMalloc(&w->dK, (size_t)mpad * mpad * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dW, (size_t)(mpad / 128) * 128 * 128 * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dyw, (size_t)mpad * sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dacc, 4 * sizeof(double)) != hipSuccess || hipMalloc(&w->dlml, sizeof(double)) != hipSuccess ||
        hipMalloc(&w->dstatus, sizeof(int)) != hipSuccess) {
      bgp_set_error("bgp_sample_y: hipMalloc of the %d x %d covariance workspace failed", mpad, mpad);
      free_child(w);
      return BGP_ERR_HIP;
    }
    w->cap_n = mpad;
    c->child = w;
  }
  w->n = m;
  w->npad = mpad;
  w->nblk = mpad / 128;
  do {
    size_t need = (size_t)m * d + p + 2 * (size_t)mpad * npad + (size_t)mpad * mpad + 2 * (size_t)mpad +
                  2 * (size_t)rpad * mpad + 64;
    rc = bgp_ensure_scratch(c, need);
    if (rc) break;
    Scratch s{c->dscratch, 0};
    double* dXq = s.take((size_t)m * d);
    double* dhk = s.take(p);
    double* dKs = s.take((size_t)mpad * npad);
    double* dP = s.take((size_t)mpad * npad);
    double* dKss = s.take((size_t)mpad * mpad);
    double* dmean = s.take(mpad);
    double* dZ = s.take((size_t)rpad * mpad);
    double* dO = s.take((size_t)rpad * mpad);
    const double* Kinv = c->dKinv + (size_t)b * npad * npad;
    const double* al = c->dalpha_sol + (size_t)b * npad;
    hipError_t e = hipSuccess;
#define SY(call)                       \
  if ((e = (call)) != hipSuccess) {    \
    bgp_set_error("bgp_sample_y: %s failed: %s", #call, hipGetErrorString(e)); \
    rc = BGP_ERR_HIP;                  \
    break;                             \
  }
    SY(hipMemcpyAsync(dXq, Xq, (size_t)m * d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    SY(hipMemcpyAsync(dhk, h_kernel, p * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (c->has_warp && (rc = bgp_launch_warp(c, c->stream, dXq, c->dwarp, dXq, m, 1, 0))) break;
    SY(hipMemsetAsync(dKs, 0, (size_t)mpad * npad * sizeof(double), c->stream));
    SY(hipMemsetAsync(dKss, 0, (size_t)mpad * mpad * sizeof(double), c->stream));
    SY(hipMemsetAsync(dZ, 0, (size_t)rpad * mpad * sizeof(double), c->stream));
    SY(hipMemcpy2DAsync(dZ, (size_t)mpad * sizeof(double), z, (size_t)m * sizeof(double), (size_t)m * sizeof(double),
                        n_draws, hipMemcpyHostToDevice, c->stream));
    if ((rc = bgp_launch_kcross(c, dhk, m, dXq, n, c->dXeff, dKs, npad, 0))) break;
    hipLaunchKernelGGL(matvec_rows_kernel, dim3((m + 3) / 4), dim3(256), 0, c->stream, dKs, npad, al, n, m, dmean);
    if ((rc = launch_gemm_nt<0>(c, dKs, npad, Kinv, npad, mpad, npad, npad, dP, npad, nullptr, 0, nullptr))) break;
    if ((rc = bgp_launch_kcross(c, dhk, m, dXq, m, dXq, dKss, mpad, 0))) break;
    hipLaunchKernelGGL(add_diag_kernel, dim3((m + 255) / 256), dim3(256), 0, c->stream, dKss, mpad, m,
                       std::exp(h_kernel[d + 1]));
    // cov = K_** - P K_*^T straight into the child's matrix
    if ((rc = launch_gemm_nt<2>(c, dP, npad, dKs, npad, mpad, mpad, npad, w->dK, mpad, dKss, mpad, nullptr))) break;
    hipLaunchKernelGGL(cov_prepare_kernel, dim3(1024), dim3(256), 0, c->stream, w->dK, m, mpad, jitter);
    SY(hipMemsetAsync(w->dyw, 0, (size_t)mpad * sizeof(double), c->stream));
    SY(hipMemsetAsync(w->dstatus, 0, sizeof(int), c->stream));
    if ((rc = bgp_launch_cholesky(w, 1, 0))) break;
    int st = 0;
    SY(hipMemcpyAsync(&st, w->dstatus, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    SY(hipStreamSynchronize(c->stream));
    if (st != 0) {
      bgp_set_error("bgp_sample_y: predictive covariance not positive definite at pivot %d (jitter %.3g)", st, jitter);
      rc = BGP_ERR_NOTPD;
      break;
    }
    hipLaunchKernelGGL(zero_upper_kernel, dim3(1024), dim3(256), 0, c->stream, w->dK, mpad);
    // out = Z L^T (+ mean)
    if ((rc = launch_gemm_nt<0>(c, dZ, mpad, w->dK, mpad, rpad, mpad, mpad, dO, mpad, nullptr, 0, nullptr))) break;
    hipLaunchKernelGGL(add_mean_rows_kernel, dim3((m + 255) / 256, n_draws), dim3(256), 0, c->stream, dO, mpad, dmean,
                       m, n_draws);
    SY(hipGetLastError());
    SY(hipMemcpy2DAsync(out, (size_t)m * sizeof(double), dO, (size_t)mpad * sizeof(double),
                        (size_t)m * sizeof(double), n_draws, hipMemcpyDeviceToHost, c->stream));
    SY(hipStreamSynchronize(c->stream));
#undef SY
  } while (0);
  (void)hipStreamSynchronize(c->stream);
  return rc;
}

void bgp_free_child(bgp_ctx* c) {
  if (c && c->child) {
    free_child(c->child);
    c->child = nullptr;
  }
}
