// Posterior build, predict, PVRS, sample_y, LML gradient.
#include "bgp_common.h"

#define BGP_NOT_YET(name)                               \
  do {                                                  \
    bgp_set_error(name ": not implemented in this build"); \
    return BGP_ERR_STATE;                               \
  } while (0)

extern "C" int bgp_lml_grad_batch(bgp_ctx*, int, const double*, double*, double*, int*) { BGP_NOT_YET("bgp_lml_grad_batch"); }
extern "C" int bgp_posterior_batch(bgp_ctx*, int, const double*, double*, double*, double*, double*, int*) { BGP_NOT_YET("bgp_posterior_batch"); }
extern "C" int bgp_predict_batch(bgp_ctx*, int, const double*, int, const double*, double*, double*, double*) { BGP_NOT_YET("bgp_predict_batch"); }
extern "C" int bgp_pvrs(bgp_ctx*, const double*, int, const double*, int, const double*, double*) { BGP_NOT_YET("bgp_pvrs"); }
extern "C" int bgp_sample_y(bgp_ctx*, int, const double*, int, const double*, int, const double*, double, double*) { BGP_NOT_YET("bgp_sample_y"); }
