// State of the device-resident ensemble sampler (bgp_mcmc.hip) shared with the fused n <= 128 kernel (bgp_chol.hip).
#pragma once
#include "bgp_common.h"

struct McmcArgs {
  int W, p, Ns, hp;       // walkers, entries of a walker, proposals per half-step, d + 2
  int nhalf;              // half-steps of the run (2 x steps)
  // Sharded ensemble (one ensemble over the ranks of a communicator, bask/bayesgpr.py:490-530 on G GPUs): the step kernel runs
  // replicated on every rank, THIS context factorises rows [row_lo, row_lo + row_n) of every half-step's proposal block and the
  // log-likelihoods of all rows come back through an all-gather enqueued on the same stream (glml: world slots of `slot` doubles,
  // the last one of a slot its rank's status word).  One rank: row_lo = 0, row_n = Ns, glml = nullptr.
  int row_lo, row_n;
  int world, slot;
  const double* glml;     // gathered log-likelihoods + status words, or nullptr (lml is read)
  const int* lml_idx;     // Ns: where proposal i's log-likelihood sits in glml
  // Walkers that carry their own input warp (bask/bayesgpr.py:353-365): the last nwarp = 2 d entries of a walker are the Beta-CDF
  // parameters [wa_1 .. wa_d, wb_1 .. wb_d]; the step kernel hands them to the warped Gram build (dwarp: row_n x nwarp)
  int nwarp;
  double* dwarp;
  double* scr;            // 3 Ns doubles (the HBM form of the step kernel: accept flags / test operands)
  double* coords;         // W x p
  double* logp;           // W
  long long* nacc;        // W
  double* q;              // Ns x p: proposals of the half-step in flight
  double* prior;          // Ns
  double* pterm;          // Ns x p: the log-prior terms of the proposals, summed in theta order by one thread per proposal
  int* status;            // row_n: statuses of the LML batch (reset here, in front of it)
  unsigned* ps_flags;     // flag block of the launch-free factorisation (reset here) or nullptr
  int ps_words;
  double* dh;             // row_n x hp: canonical hyper-parameters of this context's proposals (the LML batch reads them)
  const double* lml;      // row_n: the LML batch's results
  const int* h_src;       // hp: index into a walker, or -1: h_fixed
  const double* h_fixed;  // hp
  const int* prior_kind;  // p: 1 half-Normal on sqrt(exp(t)), 2 round-flat on exp(t) (both with the log-space Jacobian), 3 Normal on t
  const double* prior_par;  // p x 5
  const int* movers;      // nhalf x Ns   (the plan of the whole run)
  const int* partners;    // nhalf x Ns
  const double* zz;       // nhalf x Ns
  const double* factors;  // nhalf x Ns
  const double* logu;     // nhalf x Ns
  double* chain;          // steps x W x p
  double* lps;            // steps x W
  // [0] a proposal had a non-finite coordinate, [1] a launch-free factorisation (of any rank) abandoned its waits, [2] / [3] 1 + the
  // first half-step with an infinite / a NaN coordinate (0: none), [4] a rank's status word other than "redo" (its error code)
  unsigned* info;
  const unsigned* ps_err; // error word of the launch-free kernel of the half-step just finished, or nullptr
};
#define MCMC_INFO_WORDS 8
#define BGP_RANK_REDO 1000000  // status word of a rank whose launch-free factorisation timed out (bgp_comm.hip)

static __device__ __forceinline__ double mcmc_prior(int kind, const double* par, double t) {
#pragma clang fp contract(off)
  if (kind == 1) {
    // priors.halfnorm_logpdf_logspace:  c - 0.5 * exp(t) / (scale * scale) + 0.5 * t
    return par[0] - 0.5 * exp(t) / par[1] + 0.5 * t;
  }
  if (kind == 3) {
    // scipy.stats.norm(loc, scale).logpdf(t) -- the default warp priors, bask/bayesgpr.py:463-466 -- operation by operation:
    // y = (t - loc) / scale;  (-y**2 / 2.0 - log(sqrt(2 pi))) - log(scale)   (par = loc, scale, log sqrt(2 pi), log scale)
    const double y = (t - par[0]) / par[1];
    return ((-(y * y)) / 2.0 - par[2]) - par[3];
  }
  // utils._collect_priors.ls_prior:  (-2.0 * ((x / lo) ** p_lo + (x / hi) ** p_hi) - log_norm) + t,  x = exp(t) -- with the powers
  // taken in log space, (x / lo) ** p = exp(p (t - ln lo)) (par[0] = ln lo, par[1] = ln hi): three times cheaper than two pow() on
  // the one workgroup that stands between two LML batches, and as close to the exact value as numpy's route through exp(t)
  // (both carry ~|p| ulp: there the rounding of x is raised to the p-th power, here the rounding of the exponent)
  const double a = exp(par[2] * (t - par[0])), b = exp(par[3] * (t - par[1]));
  return (-2.0 * (a + b) - par[4]) + t;
}

// The fused n <= 128 form of a half-step (bgp_chol.hip): ONE launch, workgroup i proposes walker movers[h][i], builds and factorises
// its Gram matrix and accepts or rejects it -- no step kernel, no kernel boundary inside the half-step.
int bgp_launch_mcmc_small(bgp_ctx* ctx, hipStream_t st, const McmcArgs& a, int h);

// bgp_comm.hip: the per-half-step exchange of a sharded resident run, enqueued on the CONTEXT's stream `st` (no host
// synchronisation): a pack kernel puts the rank's Bp log-likelihoods (ctx->dlml) and its status word (0, or BGP_RANK_REDO read from
// the launch-free kernel's error word on the device) into the communicator's send slot, the all-gather of per + 1 doubles per rank
// lands in its receive buffer (bgp_comm_recv).  Loop-back communicators (ranks = threads of one process on one device, tests) do
// the same with device copies and events.
int bgp_comm_enqueue_lml_gather(bgp_comm* comm, bgp_ctx* ctx, hipStream_t st, int Bp, int per, const unsigned* ps_err);
const double* bgp_comm_recv(bgp_comm* comm, size_t doubles);  // the receive buffer, at least `doubles` long (nullptr: allocation failed)
int bgp_comm_rank(const bgp_comm* comm, int* rank, int* world);
// wait for `st` with the communicator's bound (BGP_COMM_TIMEOUT_S, asynchronous RCCL errors): BGP_ERR_COMM + abort instead of a hang
int bgp_comm_wait_stream(bgp_comm* comm, hipStream_t st, const char* what);
