// Generic kernel expression trees: the HOST evaluates kernel_(X) with the scikit-learn kernel object -- what the reference
// itself does for every kernel (sklearn/_gpr.py:582; bask/bayesgpr.py:148-159 accepts any skopt kernel, bask/utils.py:154-179
// builds priors by recursion over arbitrary Sum / Product trees) -- and the device does everything that follows: diagonal
// add, factorisation, solves, log-likelihood, inverse, predictive products.  Used by bayes-skopt_amd/bayesgpr.py for the trees
// kernels.py cannot map to the canonical device form (two stationary terms, products of stationaries, general Matern nu,
// RationalQuadratic, ExpSineSquared, DotProduct, ...).  No kernel arithmetic of such a tree exists on the device, and none of
// the factorisation exists on the host: there is still no CPU fallback.
#include "bgp_common.h"

// Rows of the working matrix that the upload did not write, the diagonal term and the working right-hand side:
//   i <  n:  K[i][i] += alpha_diag[i] (sklearn/_gpr.py:585)        i >= n:  identity padding (row i: zeros left of a unit diagonal)
//   yw[i] = y[i]  (y is zero padded)
__global__ void gram_fixup_kernel(double* __restrict__ Kbuf, const double* __restrict__ alpha, const double* __restrict__ y,
                                  double* __restrict__ yw, int n, int npad, int ld, size_t mstride, int use_alpha) {
  const int b = blockIdx.y;
  double* M = Kbuf + (size_t)b * mstride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npad; i += gridDim.x * blockDim.x) {
    if (i < n) {
      if (use_alpha) M[(size_t)i * ld + i] += alpha[i];
    } else {
      M[(size_t)i * ld + i] = 1.0;
    }
    yw[(size_t)b * ld + i] = y[i];
  }
  // padding rows: zeros left of the diagonal (one workgroup row-slice at a time; npad - n < 128 rows)
  for (int r = n + blockIdx.x; r < npad; r += gridDim.x)
    for (int j = threadIdx.x; j < r; j += blockDim.x) M[(size_t)r * ld + j] = 0.0;
}

// nb host matrices (n x n row-major, kernel_(X) WITHOUT the alpha term) -> the context's working matrices (ld = npad, or
// 2 npad for the augmented posterior build), padded and with the right-hand side set, as the device Gram build leaves them.
int bgp_gram_load(bgp_ctx* c, int nb, const double* K, int augmented, int use_alpha) {
  const int n = c->n, npad = c->npad;
  const size_t ld = augmented ? 2 * (size_t)npad : (size_t)npad;
  for (int b = 0; b < nb; b++)
    BGP_HIP(bgp_memcpy2d_async(c->dK + (size_t)b * ld * ld, ld * sizeof(double), K + (size_t)b * n * n, (size_t)n * sizeof(double),
                               (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(gram_fixup_kernel, dim3(std::max(1, std::min(64, npad / 64)), nb), dim3(256), 0, c->stream, c->dK, c->dalpha,
                     c->dy, c->dyw, n, npad, (int)ld, ld * ld, use_alpha);
  BGP_HIP(hipGetLastError());
  return BGP_OK;
}

extern "C" int bgp_lml_batch_gram(bgp_ctx* c, int B, const double* K, int use_alpha, double* lml, int* status) {
  BGP_REQUIRE_IDLE(c, "bgp_lml_batch_gram");
  if (!c || !K || !lml || B <= 0) {
    bgp_set_error("bgp_lml_batch_gram: bad argument");
    return BGP_ERR_INVALID;
  }
  BGP_HIP(hipSetDevice(c->device));
  const size_t nn = (size_t)c->n * c->n;
  for (int off = 0; off < B; off += c->max_batch) {
    const int nb = std::min(c->max_batch, B - off);
    BGP_HIP(hipMemsetAsync(c->dstatus, 0, nb * sizeof(int), c->stream));
    int rc = bgp_gram_load(c, nb, K + (size_t)off * nn, 0, use_alpha);
    if (!rc) rc = bgp_launch_cholesky(c, nb, 0);
    if (rc) {
      (void)hipStreamSynchronize(c->stream);
      (void)hipGetLastError();
      bgp_xfer_drop_pending();
      return rc;
    }
    BGP_HIP(bgp_memcpy_async(lml + off, c->dlml, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (status) BGP_HIP(bgp_memcpy_async(status + off, c->dstatus, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    BGP_HIP(bgp_stream_sync(c->stream));
  }
  return BGP_OK;
}
