// lbfgs_trace.cpp -- drives asr-craft_amd/host/lbfgs.h over the problems of lbfgs_problems.h and prints every accepted
// iterate in the format of tests/golden/lbfgs_ref_trace.c (which does the same with the reference's libLBFGS).
#include "lbfgs.h"
extern "C" {
#include "lbfgs_problems.h"
}

int main() {
  printf("codes %d %d %d %d %d %d %d %d %d %d\n", crf_amd::LBFGS_FAIL_TRIAL_OUTSIDE, crf_amd::LBFGS_FAIL_BOUNDS_CROSSED,
         crf_amd::LBFGS_FAIL_NO_PROGRESS, crf_amd::LBFGS_FAIL_STEP_AT_MIN, crf_amd::LBFGS_FAIL_STEP_AT_MAX,
         crf_amd::LBFGS_FAIL_SEARCH_BUDGET, crf_amd::LBFGS_FAIL_ITERATION_CAP, crf_amd::LBFGS_FAIL_INTERVAL_TOO_NARROW,
         crf_amd::LBFGS_FAIL_BAD_ARGUMENT, crf_amd::LBFGS_FAIL_UPHILL_DIRECTION);
  for (int prob = 0; prob < LP_NPROB; prob++) {
    std::vector<double> x(lp_dim[prob]);
    lp_start(prob, x.data());
    double fx = 0;
    int evals = 0;
    const int ret = crf_amd::lbfgs_minimize(
        lp_dim[prob], x.data(), &fx,
        [&](const double* p, double* g, int, double) { evals++; return lp_eval(prob, p, g); },
        [&](const double* p, const double*, double f, double xn, double gn, double step, int, int k, int ls) {
          lp_print_iter(prob, k, ls, step, f, xn, gn, p);
          return 0;
        });
    printf("end %s %d %d %a\n", lp_name[prob], ret, evals, fx);
  }
  return 0;
}
