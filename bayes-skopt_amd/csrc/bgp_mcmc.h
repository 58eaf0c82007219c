// State of the device-resident ensemble sampler (bgp_mcmc.hip) shared with the fused n <= 128 kernel (bgp_chol.hip).
#pragma once
#include "bgp_common.h"

struct McmcArgs {
  int W, p, Ns, hp;       // walkers, entries of a walker, proposals per half-step, d + 2
  int nhalf;              // half-steps of the run (2 x steps)
  double* coords;         // W x p
  double* logp;           // W
  long long* nacc;        // W
  double* q;              // Ns x p: proposals of the half-step in flight
  double* prior;          // Ns
  double* pterm;          // Ns x p: the log-prior terms of the proposals, summed in theta order by one thread per proposal
  int* status;            // Ns: statuses of the LML batch (reset here, in front of it)
  unsigned* ps_flags;     // flag block of the launch-free factorisation (reset here) or nullptr
  int ps_words;
  double* dh;             // Ns x hp: canonical hyper-parameters of the proposals (the LML batch reads them)
  const double* lml;      // Ns: the LML batch's results
  const int* h_src;       // hp: index into a walker, or -1: h_fixed
  const double* h_fixed;  // hp
  const int* prior_kind;  // p: 1 half-Normal on sqrt(exp(t)), 2 round-flat on exp(t) (both with the log-space Jacobian)
  const double* prior_par;  // p x 5
  const int* movers;      // nhalf x Ns   (the plan of the whole run)
  const int* partners;    // nhalf x Ns
  const double* zz;       // nhalf x Ns
  const double* factors;  // nhalf x Ns
  const double* logu;     // nhalf x Ns
  double* chain;          // steps x W x p
  double* lps;            // steps x W
  unsigned* info;         // [0] a proposal had a non-finite coordinate, [1] a launch-free factorisation abandoned its waits
  const unsigned* ps_err; // error word of the launch-free kernel of the half-step just finished, or nullptr
};

static __device__ __forceinline__ double mcmc_prior(int kind, const double* par, double t) {
#pragma clang fp contract(off)
  if (kind == 1) {
    // priors.halfnorm_logpdf_logspace:  c - 0.5 * exp(t) / (scale * scale) + 0.5 * t
    return par[0] - 0.5 * exp(t) / par[1] + 0.5 * t;
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
