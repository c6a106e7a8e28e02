/* lbfgs_ref_trace.c -- drives the REFERENCE's vendored libLBFGS (oracle/_ref/liblbfgs_ref.so, compiled from
 * /root/reference/CRF/src/utils/lbfgs.c where it lies) with its default parameters (NULL, as
 * trainers/CRF_LBFGSTrainer.cpp:80 passes) over the problems of tests/host/lbfgs_problems.h and prints every accepted
 * iterate.  Run only in the build container by gen_lbfgs_golden.py; its output is the fixture lbfgs_ref.npz. */
#include <lbfgs.h>   /* the reference's own header: -I/root/reference/CRF/src/utils */
#include "lbfgs_problems.h"

static int g_prob, g_evals;
static lbfgsfloatval_t evaluate(void* inst, const lbfgsfloatval_t* x, lbfgsfloatval_t* g, const int n, const lbfgsfloatval_t step) {
  (void)inst; (void)n; (void)step;
  g_evals++;
  return lp_eval(g_prob, x, g);
}
static int progress(void* inst, const lbfgsfloatval_t* x, const lbfgsfloatval_t* g, const lbfgsfloatval_t fx,
                    const lbfgsfloatval_t xnorm, const lbfgsfloatval_t gnorm, const lbfgsfloatval_t step, int n, int k, int ls) {
  (void)inst; (void)g; (void)n;
  lp_print_iter(g_prob, k, ls, step, fx, xnorm, gnorm, x);
  return 0;
}
int main(void) {
  printf("codes %d %d %d %d %d %d %d %d %d %d\n", LBFGSERR_OUTOFINTERVAL, LBFGSERR_INCORRECT_TMINMAX, LBFGSERR_ROUNDING_ERROR,
         LBFGSERR_MINIMUMSTEP, LBFGSERR_MAXIMUMSTEP, LBFGSERR_MAXIMUMLINESEARCH, LBFGSERR_MAXIMUMITERATION,
         LBFGSERR_WIDTHTOOSMALL, LBFGSERR_INVALIDPARAMETERS, LBFGSERR_INCREASEGRADIENT);
  for (g_prob = 0; g_prob < LP_NPROB; g_prob++) {
    lbfgsfloatval_t* x = lbfgs_malloc(lp_dim[g_prob]);
    lbfgsfloatval_t fx = 0;
    int ret;
    lp_start(g_prob, x);
    g_evals = 0;
    ret = lbfgs(lp_dim[g_prob], x, &fx, evaluate, progress, NULL, NULL);
    printf("end %s %d %d %a\n", lp_name[g_prob], ret, g_evals, fx);
    lbfgs_free(x);
  }
  return 0;
}
