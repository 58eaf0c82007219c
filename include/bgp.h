/*
 * bgp.h -- C-ABI of the MI355X-native BayesGPR hot path (libbgp.so).
 *
 * The reference (kiudee/bayes-skopt 0.11.0) has no FFI boundary: the seams this library sits
 * behind are Python call sites.  Each entry point below names the
 * reference interface it replaces.  The host side that binds these symbols with ctypes lives in
 * bayes-skopt_amd/_lib.py; INTEGRATION.md shows the stub a bask maintainer would add.
 *
 * Conventions
 *  - plain C types only; all host buffers are caller-owned, C-contiguous, fp64 (double) / int32;
 *  - return value 0 == BGP_OK, anything else is an error whose text bgp_last_error() returns
 *    (thread-local); no C++ exception crosses the ABI;
 *  - NUMERICAL failures (non-positive pivot == "not positive definite") are per-item:
 *    status[b] != 0 and lml[b] = -inf  (sklearn/_gpr.py:586-589 returns -inf there), never a
 *    call failure;
 *  - hyper-parameters are passed in the CANONICAL log-space layout
 *        h = [ log c, log l_1 .. log l_d, log s2 ]            (d + 2 doubles per item)
 *    with   K = c * S(r) + s2 * I   (form BGP_FORM_PRODUCT; ConstantKernel * Matern + WhiteKernel,
 *                                    bask/utils.py:144-150 + skopt's added WhiteKernel)
 *           K = c + S(r) + s2 * I   (form BGP_FORM_SUM;     examples/Fit-GP.ipynb cell 6)
 *    r_ij^2 = sum_k ((x_ik - x_jk) / l_k)^2 ; -inf encodes an absent / zeroed component
 *    (bask/bayesgpr.py:328-333 swaps in WhiteKernel(0.0)).  The mapping from a kernel object's
 *    theta (sklearn/kernels.py:733-760 concatenation order, fixed hyper-parameters dropped,
 *    isotropic length scale replicated) to h is host logic (bayes-skopt_amd/kernels.py);
 *  - one context per (host thread, device); calls on a context are serialised on its HIP stream.
 */
#ifndef BGP_H
#define BGP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGP_OK 0
#define BGP_ERR_INVALID 1  /* bad argument                      */
#define BGP_ERR_HIP 2      /* a HIP runtime call failed          */
#define BGP_ERR_NODEVICE 3 /* no usable gfx950 device            */
#define BGP_ERR_STATE 4    /* call order (e.g. predict before posterior) */
#define BGP_ERR_NOTPD 5    /* bgp_sample_y: covariance (+jitter) not positive definite */
#define BGP_ERR_COMM 6     /* a collective did not complete (peer lost, asynchronous RCCL error): communicator aborted */

enum { BGP_FORM_PRODUCT = 0, BGP_FORM_SUM = 1 };
enum { BGP_RBF = 0, BGP_MATERN12 = 1, BGP_MATERN32 = 2, BGP_MATERN52 = 3 };

typedef struct bgp_kernel_spec {
  int form;       /* BGP_FORM_*                                   */
  int stationary; /* BGP_RBF / BGP_MATERN*  (sklearn/kernels.py:1553-1560, 1713-1733) */
  int d;          /* input dimension                              */
} bgp_kernel_spec;

typedef struct bgp_ctx bgp_ctx;

/* Number of visible HIP devices (0 when none / runtime unusable). */
int bgp_device_count(void);
/* PCI bus id ("0000:05:00.0") of visible device `device` into buf (len >= 16): the identity that tells apart ranks which a
 * launcher has pinned to one GPU each through HIP_VISIBLE_DEVICES -- they all see "device 0" (bayes-skopt_amd/distributed.py). */
int bgp_device_pci_bus_id(int device, char* buf, int len);
/* Text of the last error on this thread. */
const char* bgp_last_error(void);
/* Library version string. */
const char* bgp_version(void);

/*
 * Create a context on `device` holding the training set (copied to HBM and kept resident):
 * X (n*d row-major), y (n, already normalised by the caller as bask/bayesgpr.py:470-478 does),
 * alpha_diag (n; the diagonal term sklearn/_gpr.py:585 adds -- scalar alpha already broadcast,
 * noise vector already added, bask/bayesgpr.py:338-349).
 * max_batch bounds the number of hyper-parameter vectors factorised concurrently (workspace is
 * max_batch * n_pad^2 doubles); larger B are processed in chunks.
 * Replaces: the data the reference keeps on self.X_train_/y_train_/alpha (bask/bayesgpr.py:485-488).
 */
int bgp_ctx_create(int device, int n, int d, const double* X, const double* y, const double* alpha_diag,
                   const bgp_kernel_spec* ks, int max_batch, bgp_ctx** out);

/* tell() grows the data set (bask/optimizer.py:288-320 -> bask/bayesgpr.py:469-488). d is fixed. */
int bgp_ctx_update_data(bgp_ctx* ctx, int n, const double* X, const double* y, const double* alpha_diag);

void bgp_ctx_destroy(bgp_ctx* ctx);

/*
 * Batched log-marginal-likelihood: for each of B canonical vectors h_b
 *     K_b = kernel(X) ; K_b[diag] += alpha_diag ; L_b = chol(K_b) ;
 *     lml_b = -1/2 y^T K_b^-1 y - sum log diag(L_b) - n/2 log(2 pi)
 * Replaces: GaussianProcessRegressor.log_marginal_likelihood(theta) (sklearn/_gpr.py:537-613)
 * as called per walker by BayesGPR._log_prob_fn (bask/bayesgpr.py:374) from
 * emcee.EnsembleSampler.compute_log_prob (bask/bayesgpr.py:510-524).
 * status[b] = 0 ok, >0 = 1-based index of the failing pivot (lml[b] = -inf). status may be NULL.
 */
int bgp_lml_batch(bgp_ctx* ctx, int B, const double* h, double* lml, int* status);

/*
 * Input warping (bask/bayesgpr.py:249-316, warp_inputs=True): every input column k is passed through the
 * CDF of Beta(exp(wa_k), exp(wb_k)) before the kernel is evaluated.  warp vectors are 2*d doubles in log
 * space, [wa_1..wa_d, wb_1..wb_d] -- the tail the reference appends to theta (bask/bayesgpr.py:355-357).
 *
 * bgp_lml_batch_warped: as bgp_lml_batch, but walker b sees the design matrix through its own warp
 *   warp[b*2d .. (b+1)*2d).  Replaces create_warpers() + rewarp() + log_marginal_likelihood per walker
 *   (bask/bayesgpr.py:353-374).
 * bgp_ctx_set_warp: context-level warp (the geometric-median warpers, bask/bayesgpr.py:535-541) seen by
 *   every later posterior / predict / pvrs / gradient / sample_y / bgp_lml_batch call, including their
 *   query points (bask/bayesgpr.py:630-632); NULL clears it; bgp_ctx_update_data clears it.
 * bgp_beta_cdf: warp m points (m*d row-major) on the device -- BayesGPR.warp() (bask/bayesgpr.py:249-264).
 */
int bgp_lml_batch_warped(bgp_ctx* ctx, int B, const double* h, const double* warp, double* lml, int* status);
int bgp_ctx_set_warp(bgp_ctx* ctx, const double* warp);
int bgp_beta_cdf(bgp_ctx* ctx, int m, const double* X, const double* warp, double* out);

/*
 * Asynchronous bgp_lml_batch (B <= max_batch): submit enqueues the whole batch on the device and returns, wait blocks
 * until it is done and hands back lml / status (status may be NULL); with per-launch timing on (bgp_set_timing) submit
 * returns BGP_ERR_STATE (the timed path synchronises inside bgp_lml_batch).  Between the two calls the host is free -- the
 * sampler evaluates the log-priors of the same proposals there (bask/bayesgpr.py:366-372 runs them back to back).
 * One batch may be pending per context, and it owns the workspace: until wait has collected it every other entry point
 * that computes on the context returns BGP_ERR_STATE.  Proposals and results travel through pinned host memory; same
 * results as bgp_lml_batch (which takes this path itself for a single chunk).
 */
int bgp_lml_batch_submit(bgp_ctx* ctx, int B, const double* h);
/* the same for per-walker warps (the asynchronous form of bgp_lml_batch_warped; bask/bayesgpr.py:353-374) */
int bgp_lml_batch_warped_submit(bgp_ctx* ctx, int B, const double* h, const double* warp);
int bgp_lml_batch_wait(bgp_ctx* ctx, double* lml, int* status);

/*
 * LML and its gradient w.r.t. the canonical vector (grad is B*(d+2)):
 *     g_k = 1/2 tr((alpha alpha^T - K^-1) dK/dh_k)
 * Replaces: log_marginal_likelihood(theta, eval_gradient=True) (sklearn/_gpr.py:615-647,
 * kernels.py:1746-1766) driven by L-BFGS-B inside fit (bask/bayesgpr.py:607).
 */
int bgp_lml_grad_batch(bgp_ctx* ctx, int B, const double* h, double* lml, double* grad, int* status);

/*
 * Parity helper: the jittered Gram matrix K(h) itself, n*n row-major (both triangles filled).
 * Replaces: kernel_(X_train_) + diagonal add (bask/bayesgpr.py:203-204).
 */
int bgp_kernel_matrix(bgp_ctx* ctx, const double* h, double* K);

/*
 * Posterior build for B hyper-posterior samples; factors stay RESIDENT on the device for the
 * predict / pvrs calls that follow.  Any of the output pointers may be NULL.
 *   L      B*n*n row-major lower Cholesky factors (upper triangle zero)   -> BayesGPR.L_
 *   alpha  B*n            K^-1 y                                          -> BayesGPR.alpha_
 *   K_inv  B*n*n          explicit inverse                                -> BayesGPR.K_inv_
 * Replaces: the BayesGPR.theta setter (bask/bayesgpr.py:200-217), called once per sample()
 * (:544) and once per hyper-posterior sample by evaluate_acquisitions (bask/acquisition.py:112-121).
 */
int bgp_posterior_batch(bgp_ctx* ctx, int B, const double* h, double* L, double* alpha, double* K_inv,
                        double* lml, int* status);

/*
 * Predict at m query points with the B resident posteriors.  h_kernel (B*(d+2)) are the
 * hyper-parameters currently in kernel_ -- they differ from the posterior's only inside
 * noise_set_to_zero() (log s2 = -inf), which does NOT rebuild the factors
 * (bask/bayesgpr.py:318-336).  Outputs (normalised-y units; the caller undoes y
 * normalisation): mean B*m, var B*m (clipped at 0), optional cov B*m*m (may be NULL).
 *   mean = K_* alpha ; var = k_** - diag(K_* K^-1 K_*^T) ; cov = K_** - K_* K^-1 K_*^T
 * Replaces: BayesGPR.predict (bask/bayesgpr.py:622-635 -> skopt predict).
 */
int bgp_predict_batch(bgp_ctx* ctx, int B, const double* h_kernel, int m, const double* Xq, double* mean,
                      double* var, double* cov);

/*
 * Closed-form acquisition functions averaged over the B resident posteriors (hyper-posterior draws), evaluated
 * on the device right behind the batched predict so that the B*m means and variances never cross PCIe.
 * For acquisition k (kinds[k], params[k]) and draw b, with mu = y_std * mean + y_mean, std = sqrt(var * y_std^2):
 *   BGP_ACQ_EI    std * (x Phi(x) + phi(x)), x = (y_opt - mu) / std, 0 where std <= 0; y_opt = params[k], or the
 *                 draw's lowest mu when params[k] is NaN          (bask/acquisition.py:154-172)
 *   BGP_ACQ_MEAN  -mu                                              (Expectation, :197-201)
 *   BGP_ACQ_LCB   params[k] * std - mu                             (LCB, :204-216)
 *   BGP_ACQ_STD   std                                              (LCB with alpha = "inf")
 * out[k*m + i] = sum over the draws, in order, of value / n_samples; a draw whose values are not all finite
 * contributes nothing (evaluate_acquisitions, bask/acquisition.py:112-139).  h_kernel as in bgp_predict_batch.
 * bgp_acq_values applies the same closed forms to caller-supplied mu / std (B*m each, y units).
 */
enum { BGP_ACQ_EI = 0, BGP_ACQ_MEAN = 1, BGP_ACQ_LCB = 2, BGP_ACQ_STD = 3 };
#define BGP_ACQ_MAX 8
int bgp_acq_batch(bgp_ctx* ctx, int B, const double* h_kernel, int m, const double* Xq, double y_mean, double y_std,
                  int n_acq, const int* kinds, const double* params, int n_samples, double* out);
int bgp_acq_values(bgp_ctx* ctx, int B, int m, const double* mu, const double* std_, int n_acq, const int* kinds,
                   const double* params, int n_samples, double* out);

/*
 * PVRS inner loop with the resident posterior built by bgp_pvrs_prepare (K without the candidate
 * row):  for every candidate i
 *   covs[i] = sum_t k_t,aug^T K_aug,i^-1 k_t,aug ,   K_aug,i = kernel([X_train; x_i])
 * computed with the bordered-Cholesky identity instead of m factorizations.
 * Replaces: PVRS.__call__ loop (bask/acquisition.py:328-338); with thompson == candidates it is
 * VarianceReduction (bask/acquisition.py:287-300).
 */
int bgp_pvrs(bgp_ctx* ctx, const double* h_kernel, int m, const double* Xcand, int T, const double* Xthompson,
             double* covs);
/* Build the resident posterior bgp_pvrs needs: K = kernel_(X_train) (+ alpha_diag only when
 * has_alpha_vec != 0 -- the reference adds alpha to the augmented diagonal only when it is
 * iterable, bask/acquisition.py:332-333).  status (1 int, may be NULL) as in bgp_lml_batch. */
int bgp_pvrs_prepare(bgp_ctx* ctx, const double* h_kernel, int has_alpha_vec, int* status);

/*
 * Draw f ~ N(mean, cov) at m points for resident posterior b using standard normals supplied by
 * the host (z: n_draws*m), via a Cholesky factor of cov (+jitter) instead of numpy's SVD.
 * Replaces: sklearn sample_y (sklearn/_gpr.py:522-526) reached from BayesGPR.sample_y
 * (bask/bayesgpr.py:637-718).  out: n_draws*m.  Returns BGP_ERR_NOTPD when cov + jitter*I is not
 * numerically positive definite (the caller retries with a larger jitter).
 */
int bgp_sample_y(bgp_ctx* ctx, int b, const double* h_kernel, int m, const double* Xq, int n_draws,
                 const double* z, double jitter, double* out);

/*
 * One draw per hyper-posterior sample, batched: item i draws f_i ~ N(mean_i, cov_i) at the m points from the
 * resident posterior pidx[i] with kernel parameters h_kernel[i] (B*(d+2)) and the standard normals z[i] (B*m);
 * out is B*m.  The B covariance matrices are factorised by ONE batched Cholesky.  status[i] != 0: cov_i + jitter*I
 * not numerically positive definite (out[i] undefined; the caller retries those items with a larger jitter).
 * Replaces: the per-sample loop of BayesGPR.sample_y(sample_mean=False) -- theta setter + sklearn sample_y per
 * chain row (bask/bayesgpr.py:679-718) -- after one bgp_posterior_batch over the drawn chain rows.
 */
int bgp_sample_y_batch(bgp_ctx* ctx, int B, const int* pidx, const double* h_kernel, int m, const double* Xq,
                       const double* z, double jitter, double* out, int* status);

/*
 * Generic kernel expression trees.  The reference accepts ANY skopt / scikit-learn kernel (bask/bayesgpr.py:148-159; priors by
 * recursion over arbitrary Sum / Product trees, bask/utils.py:154-179) and evaluates it on the host: K = kernel_(X_train)
 * (sklearn/_gpr.py:582).  For trees that have no canonical device form (two stationary terms, products of stationaries, general
 * Matern nu, RationalQuadratic, ...) the host side does exactly that with the scikit-learn kernel object and hands the matrices
 * over; the device does everything behind them -- diagonal add, factorisation, solves, log-likelihood, inverse, predictive
 * products: the same kernels as the canonical path, no factorisation on the host.
 *   K      B matrices of n*n doubles, row-major: kernel_(X_train) WITHOUT the alpha term; use_alpha != 0 adds the context's
 *          alpha_diag on the device (sklearn/_gpr.py:585)
 *   bgp_lml_batch_gram        as bgp_lml_batch        (sklearn/_gpr.py:537-613 via bask/bayesgpr.py:374)
 *   bgp_posterior_batch_gram  as bgp_posterior_batch  (bask/bayesgpr.py:200-217); the posteriors stay resident
 *   bgp_predict_batch_gram    as bgp_predict_batch for the B resident posteriors with host-evaluated cross covariances:
 *          Ks B*m*n = kernel_(Xq, X_train), kss B*m = kernel_.diag(Xq), Kss B*m*m = kernel_(Xq) (read only when cov != NULL)
 *          (bask/bayesgpr.py:622-635 -> skopt predict)
 */
int bgp_lml_batch_gram(bgp_ctx* ctx, int B, const double* K, int use_alpha, double* lml, int* status);
int bgp_posterior_batch_gram(bgp_ctx* ctx, int B, const double* K, int use_alpha, double* L, double* alpha, double* K_inv,
                             double* lml, int* status);
int bgp_predict_batch_gram(bgp_ctx* ctx, int B, int m, const double* Ks, const double* kss, const double* Kss, double* mean,
                           double* var, double* cov);

/* Number of walker groups (HIP streams) an LML batch is split over: one group's kernels fill the tail
 * of the other group's launches.  Default: automatic -- two groups for batches of >= 64 matrices (+3.7 % at
 * n = 2048 x 128 matrices on MI355X, results bit-identical), one group below that (every group's dependent
 * chain is as long as the whole batch's, nothing to gain).  This call, or the environment variable
 * BGP_STREAMS read at context creation, forces a fixed group count.
 *
 * Environment switches the library reads (a BGP_* variable it does not read is reported once on stderr).  Schedule switches --
 * each selects between code paths whose results are bit-identical (tests/test_gpu_edge.py, tests/test_gpu_persist.py):
 *   BGP_STREAMS (this call), BGP_PANELS (block columns per trailing update, 1..64; default 4 from n = 1536, else 2), BGP_PERSIST
 *   (bgp_set_persist), BGP_PS_PAIR (0 / 1: one or two chain workgroups per matrix on the launch-free path; default by shape),
 *   BGP_PS_GEN (0 / 1: the Gram blocks of a launch-free LML batch are built by a kernel in front of it / by its own tile workers
 *   at the head of their ticket list; default by shape), BGP_SYRK_GEN (0 / 1: on the launch schedule, every Gram block by the
 *   kernel in front / all but block column 0 in the accumulators of the first panel group's updates; default 1 where the
 *   pipelined Gram kernel would run, see bgp_lml_gen_stats).
 * Runtime switch set by the PYTHON package, not read by this library: bayes-skopt_amd/_lib.py exports HIP_FORCE_DEV_KERNARG=1 (kernel
 *   arguments in device memory: -6 % per call on the launch schedule at n = 1024 x 32) at import unless the user has set it --
 *   a process-wide setting that every other HIP user of the process inherits, effective only if the GPU has not been initialised
 *   before the import; BGP_NO_ENV_DEFAULTS=1 leaves the environment alone.  A C caller of libbgp.so sets it itself.
 * Waits and diagnostics (no effect on results): BGP_WAIT=block, BGP_PS_TIMEOUT_MS, BGP_PS_COOLDOWN, BGP_PS_TRACE,
 *   BGP_COMM_TIMEOUT_S (DESIGN.md sections 2, 7 and 4).  The A/B switches of earlier rounds (BGP_FUSED_GRAM, BGP_KBUILD1,
 *   BGP_SMALL_SPLIT, BGP_PS_NCRIT / _PSPLIT / _STREAM, BGP_PANEL_WIDTH, BGP_ROWQUAD_T) left the library in round 5 with the
 *   measured-slower variants they selected. */
int bgp_set_streams(bgp_ctx* ctx, int nstreams);

/* hipDeviceSynchronize on `device` (timing brackets in bench.py). */
int bgp_device_synchronize(int device);

/* ---- measurement hooks (bench.py / profiling; not part of the reference surface) ---- */

/* Average device time (ms, HIP events on the context's stream) of the kernels of the last
 * bgp_lml_batch call: out[0]=K-build, out[1]=potrf (diagonal blocks), out[2]=trsm (panels),
 * out[3]=syrk (trailing update), out[4]=whole call on device; counts[0..3] = launches. */
int bgp_last_timing(bgp_ctx* ctx, double* out_ms, int* counts);
/* Time (ms) and count of the look-ahead column launches of the trailing update (K = 128 .. 128 (P-1), one block column)
 * inside out[3] of bgp_last_timing, for the last timed bgp_lml_batch call. */
int bgp_last_timing_columns(bgp_ctx* ctx, double* ms, int* launches);
/* Launch-free factorisation of small batches (one persistent kernel per bgp_lml_batch call -- and per covariance of
 * bgp_sample_y -- instead of ~3 launches per block column; same bits): 1 = whenever the batch fits (<= 64 matrices,
 * n > 128), 0 = never, -1 = as BGP_PERSIST says (unset: where it measured faster on MI355X: at least 6 block columns of
 * 128, matrices x block columns <= 400; and -- with two chain workgroups per matrix -- few matrices from 3 block columns on:
 * bgp_persist_auto_rule / bgp_pair_auto_rule).  Replaces nothing in the reference; a scheduling choice behind cholesky() of
 * sklearn/_gpr.py:587. */
int bgp_set_persist(bgp_ctx* ctx, int mode);
/* Bookkeeping of that path: out[0] = launch-free calls enqueued by this context, out[1] = of which timed out (a wait
 * outlasted BGP_PS_TIMEOUT_MS, 500 ms: the workgroups were not co-resident; the batch was redone by launches, same
 * results), out[2] = 1 while a time-out keeps the path switched off, out[3] = eligible calls left before it is tried again
 * (BGP_PS_COOLDOWN, 256; after the third time-out the path stays off until bgp_set_persist(ctx, 1)).  A context that shares
 * its device with other contexts, processes or collectives should call bgp_set_persist(ctx, 0). */
int bgp_persist_stats(bgp_ctx* ctx, long long* out4);
/* Gram generation inside the trailing update (launch schedule of the LML path; csrc/bgp_s4.h S4GenF): the kernel matrix K(X, X) of
 * sklearn/_gpr.py:582-585 is built by the Gram kernel for block column 0 only; every other block is computed in the accumulators
 * of the update of the first panel group that touches it first (same arithmetic per element: same bits; BGP_SYRK_GEN=0 builds every
 * block with the Gram kernel in front).  out[0] = LML batches (per walker-group stream) factorised that way by this context,
 * out[1] = generating launches among their trailing updates. */
int bgp_lml_gen_stats(bgp_ctx* ctx, long long* out2);
/* Enable (1) / disable (0) per-kernel event timing (adds synchronisation; off by default). */
int bgp_set_timing(bgp_ctx* ctx, int enable);
/* Debugging aid (BGP_PS_TRACE=1): in-kernel wall-clock stamps (100 MHz) of the last launch-free call.  dims = {matrices,
 * block columns, tile tasks}; out (may be NULL to query dims) receives 8 stamps per (matrix, block column) of the chain
 * role, then 8 per tile task; cap = capacity of out in 64-bit words.  Read by tools/persist_trace.py. */
int bgp_debug_ps_trace(bgp_ctx* ctx, int* dims, unsigned long long* out, size_t cap);
/* Debugging aid: working matrix (npad x npad doubles; L in the lower triangle after an LML call) and working right-hand
 * side (npad doubles, z = L^-1 y) of batch slot b as the last bgp_lml_batch left them; either pointer may be NULL. */
int bgp_debug_workspace(bgp_ctx* ctx, int b, double* L, double* z);
/* Debugging aid: the Cholesky factor of the predictive covariance as the last bgp_sample_y left it (*mpad = its padded
 * edge, a multiple of 128, 0 when there is none; L: mpad x mpad doubles, lower triangle; may be NULL to query mpad). */
int bgp_debug_cov_factor(bgp_ctx* ctx, int* mpad, double* L);
/* Debugging aid / accuracy test: sqrt(x[i]) and 1 / sqrt(x[i]) as the diagonal-block factorisation forms its pivots
 * (hardware seed + coupled Goldschmidt step + one residual correction each; LAPACK dpotrf's sqrt via sklearn/_gpr.py:587). */
int bgp_debug_pivot_root(int device, int n, const double* x, double* sqrt_out, double* rsqrt_out);
/* fp64 MFMA micro-benchmark: TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 on the whole chip. */
int bgp_bench_mfma_f64(int device, int iters, double* tflops);
/* HBM copy micro-benchmark: GB/s (read+write) of a streaming double2 copy of `bytes` bytes. */
int bgp_bench_hbm_copy(int device, long long bytes, int iters, double* gbps);
/* Empirical C/D fragment layout of v_mfma_f64_16x16x4_f64: rows[64*4], cols[64*4] (lane*4+reg). */
int bgp_mfma_f64_layout(int device, int* rows, int* cols);

/*
 * Multi-GPU exchange over RCCL (one process per GPU).  The hot path shards by chains: no collective in
 * the sampling loop, ONE all-gather of the posterior samples at the end (bgp_comm_allgather; in the exact
 * single-ensemble option also the B log-probabilities of each half-step).  Host buffers in, host buffers out; the
 * staging buffers stay resident on the device.  bgp_comm_unique_id is called on rank 0 and its BGP_COMM_ID_BYTES bytes
 * are handed to every rank by the caller (bayes-skopt_amd/distributed.py: a per-job file in a per-user directory on a
 * single node, a TCP socket on MASTER_ADDR:(MASTER_PORT + 1) across nodes).
 * Replaces: nothing in the reference (it is single-process); mirrors what torch.distributed's all_gather /
 * all_reduce(MAX) / broadcast would do, without PyTorch in the product path.
 */
#define BGP_COMM_ID_BYTES 128
typedef struct bgp_comm bgp_comm;
int bgp_comm_available(void); /* 1 when librccl.so can be loaded in this process */
int bgp_comm_unique_id(void* id128);
int bgp_comm_init(int device, int rank, int world, const void* id128, bgp_comm** out);
void bgp_comm_destroy(bgp_comm* comm);
/* recv (world * count doubles) = concatenation of every rank's send (count doubles), rank-major */
int bgp_comm_allgather(bgp_comm* comm, const double* send, size_t count, double* recv);
int bgp_comm_allreduce_max(bgp_comm* comm, double* inout, size_t count);
int bgp_comm_broadcast(bgp_comm* comm, double* buf, size_t count, int root);
int bgp_comm_barrier(bgp_comm* comm);
/* This rank is going down outside a collective (interrupt, fatal error): ncclCommAbort, so that the peers' collectives fail at
 * once (BGP_ERR_COMM on their side) instead of waiting BGP_COMM_TIMEOUT_S; every later call on the communicator answers BGP_ERR_COMM. */
int bgp_comm_abort(bgp_comm* comm);
/* ranks RCCL counts in the communicator (ncclCommCount) */
int bgp_comm_nranks(bgp_comm* comm, int* nranks);
/* Loop-back communicator: rank `rank` of a group of `world` communicators of THIS process on ONE device (one per host thread, each
 * beside its own context; the communicators that name the same key form the group).  It serves the in-stream exchange of a
 * sharded bgp_mcmc_begin_ex run with device copies and events instead of RCCL -- the row-sharding logic of the multi-GPU sampler
 * exercised with world > 1 semantics on a single GPU (tests); every other collective answers BGP_ERR_COMM. */
int bgp_comm_init_loopback(int device, int rank, int world, long long key, bgp_comm** out);
/* Measurement hook (bench.py): `reps` rounds of the sharded resident sampler's per-half-step exchange (pack kernel + all-gather of
 * per + 1 doubles per rank) back to back on the CONTEXT's stream between two HIP events: the device-side price of the exchange
 * where the sampler pays it, ms per round.  Every rank of the communicator calls it together. */
int bgp_comm_bench_lml_gather(bgp_ctx* ctx, bgp_comm* comm, int per, int reps, double* ms_per_round);
/* Exact single-ensemble sharding of bask/bayesgpr.py:490-530 (ONE n_walkers ensemble, one RNG): every rank has
 * submitted its own rows of a half-step's proposal block with bgp_lml_batch_submit (possibly none); this replaces
 * bgp_lml_batch_wait and returns the log-likelihoods of ALL ranks (world * per_rank doubles, rank-major, gathered device to
 * device out of every context's resident result vector: the communicator's stream waits for the context's stream through
 * an event, ONE host synchronisation per half-step).
 * Every rank must ENTER this call once it has been agreed on (RCCL has no time-out): a rank whose own work failed between
 * submit and wait passes local_error > 0 instead of returning early; its values travel as NaN and errors_out[r] (world
 * ints, identical on every rank) carries the code, so that all ranks raise alike.  A rank whose launch-free factorisation
 * timed out redoes its batch by launches and all ranks repeat the gather (decided from the same gathered status word).
 * The pending batch is consumed whatever the outcome.  Returns BGP_OK when the collective completed (inspect errors_out),
 * BGP_ERR_COMM when it did not within BGP_COMM_TIMEOUT_S seconds (default 300) or RCCL reported an asynchronous error:
 * the communicator is aborted (ncclCommAbort) and every later collective on it answers BGP_ERR_COMM. */
int bgp_lml_batch_wait_allgather(bgp_ctx* ctx, bgp_comm* comm, int per_rank, int local_error, double* lml_all,
                                 int* errors_out);

/* ---- the ensemble sampler with its state resident in HBM ------------------------------------------------------------
 * Replaces the per-half-step traffic of emcee's loop (bask/bayesgpr.py:510-530 -> emcee 3.1.6 EnsembleSampler.sample with
 * one StretchMove: propose, compute_log_prob, accept, for each half of the ensemble): nsteps steps of W walkers (W even) with p
 * entries each run on the device without a transfer or a synchronisation in between.  The caller draws every random number of
 * the run in emcee's stream order (none depends on a log-probability) and passes them as the plan, 2 * nsteps half-steps of
 * Ns = W / 2 rows: movers / partners (walker indices), zz (the stretch factors z), factors ((p - 1) log z) and logu (log of
 * the accept draws).  A proposal is q = c - (c - s) z (s = coords[mover], c = coords[partner]); its canonical hyper-parameters
 * are h[j] = h_src[j] >= 0 ? q[h_src[j]] : h_fixed[j] (j < d + 2); its log-prior is the sum over the p entries, in order, of
 *   prior_kind 1:  par[0] - 0.5 exp(t) / par[1] + 0.5 t                      (half-Normal on sqrt(exp t), bask/utils.py:95-99)
 *   prior_kind 2:  (-2 (exp(par[2] (t - par[0])) + exp(par[3] (t - par[1]))) - par[4]) + t   (round-flat on exp t, bask/priors.py:7-57:
 *                  -2 ((x / lo)^p_lo + (x / hi)^p_hi) - log_norm with par = (ln lo, ln hi, p_lo, p_hi, log_norm))
 * (prior_par: p x 5); log-probability = log-prior + LML, non-finite -> -inf (bask/bayesgpr.py:351-379); accept iff
 * factors + lp_new - lp_old > logu.  Outputs: chain (nsteps x W x p) and logp (nsteps x W) after every step, the final
 * ensemble (coords_out, logp_out), accept counts, and info (4 ints): info[0] != 0 when a proposal had a non-finite coordinate
 * (emcee raises ValueError there: the caller should) -- info[2] is then the index of the first such half-step and info[3] says
 * whether its first offender was a NaN (1) or an infinity (0), as emcee's two messages distinguish --, info[1] = 1 when a
 * launch-free factorisation gave up its waits and the WHOLE run was redone on the launch schedule (same bits).  Needs the rows of a
 * half-step that this context factorises (W / 2, or its share of them) <= max_batch, no pending batch, per-launch timing off.  An
 * ensemble too large for the step kernel's 160 KB of LDS (W p + W + 3 Ns + 2 Ns p > 20 480 doubles) runs the same kernel on the
 * arrays in HBM; n <= 128 with at most 64 entries per walker takes the fused one-launch half-step.
 * bgp_mcmc_begin / bgp_mcmc_steps / bgp_mcmc_end are the same run with the plan handed over in segments (nseg steps = 2 nseg rows
 * of every plan array per call, in order): bgp_mcmc_steps uploads its rows, enqueues their half-steps and returns at once, so the
 * caller draws the next segment's random numbers while the device works through this one; bgp_mcmc_end (every step handed over)
 * waits and collects.  Between begin and end the context belongs to the run: every other entry point answers BGP_ERR_STATE;
 * bgp_ctx_destroy drops an open run.  bgp_mcmc_run = begin + one bgp_mcmc_steps with the whole plan + end.
 * bgp_mcmc_progress: steps of the open run whose segment the device has worked through (never blocks): what a progress bar shows
 * while the run is resident (emcee's tqdm bar, bask/bayesgpr.py:522-524 with progress=True, the default of fit, :550-564).
 *
 * bgp_mcmc_begin_ex adds the two things that change the SHAPE of a half-step:
 *   nwarp = 2 d   walkers that carry their own input warp (warp_inputs=True, bask/bayesgpr.py:353-365): the last 2 d entries of a
 *                 walker are [wa_1 .. wa_d, wb_1 .. wb_d] (log space) and every proposal's Gram matrix is built on the design
 *                 matrix seen through the Beta CDFs of ITS parameters (the kernels of bgp_lml_batch_warped); h_src only reads the
 *                 first p - 2 d entries.  Log-prior = sum over those entries, in order, + sum over k of (prior(wa_k) + prior(wb_k))
 *                 (the reference's two loops, :360-372).  prior_kind 3:  ((-(y y)) / 2 - par[2]) - par[3], y = (t - par[0]) / par[1]
 *                 (Normal(loc, scale) on t with par = (loc, scale, log sqrt(2 pi), log scale): scipy's norm.logpdf operation by
 *                 operation, the default warp priors of bask/bayesgpr.py:463-466).  0: no warp.
 *   comm          ONE ensemble sharded over the ranks of a communicator (SURVEY.md 8e option 1): every rank calls with the same
 *                 arguments and the same plan; the step kernel runs replicated on every rank, rank r factorises rows
 *                 [r Ns / G, (r + 1) Ns / G) of every half-step's proposal block, and an all-gather of ceil(Ns / G) + 1 doubles per
 *                 rank (log-likelihoods + a status word) sits between the LML batch and the next step kernel ON THE CONTEXT'S
 *                 STREAM -- no host synchronisation per half-step, the chain equals the one-GPU chain bit for bit.  A rank whose
 *                 launch-free factorisation timed out says so in its status word and EVERY rank redoes the whole run on the launch
 *                 schedule; bgp_mcmc_end waits with the communicator's bound (BGP_COMM_TIMEOUT_S) and returns BGP_ERR_COMM instead
 *                 of hanging; a rank that drops its run early aborts the communicator.  NULL: one rank. */
int bgp_mcmc_begin_ex(bgp_ctx* ctx, bgp_comm* comm, int nwarp, int W, int p, int nsteps, const int* h_src,
                      const double* h_fixed, const int* prior_kind, const double* prior_par, const double* coords0,
                      const double* logp0);
int bgp_mcmc_progress(bgp_ctx* ctx, int* steps_done);
int bgp_mcmc_begin(bgp_ctx* ctx, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                   const double* prior_par, const double* coords0, const double* logp0);
int bgp_mcmc_steps(bgp_ctx* ctx, int nseg, const int* movers, const int* partners, const double* zz, const double* factors,
                   const double* logu);
int bgp_mcmc_end(bgp_ctx* ctx, double* chain, double* logp, double* coords_out, double* logp_out, long long* naccepted, int* info);
int bgp_mcmc_run(bgp_ctx* ctx, int W, int p, int nsteps, const int* h_src, const double* h_fixed, const int* prior_kind,
                 const double* prior_par, const double* coords0, const double* logp0, const int* movers, const int* partners,
                 const double* zz, const double* factors, const double* logu, double* chain, double* logp, double* coords_out,
                 double* logp_out, long long* naccepted, int* info);

#ifdef __cplusplus
}
#endif
#endif /* BGP_H */
