// Test translation unit for asr-craft_amd/host/lbfgs.h (the optimiser behind CRF_LBFGSTrainer; the reference links
// libLBFGS with default parameters, trainers/CRF_LBFGSTrainer.cpp:55-62).  Prints one line per problem:
//   <name> ret=<code> fx=<value> iters=<k> evals=<n> x=<x0> <x1> ...
// tests/test_host_lbfgs.py checks the minima against closed forms (the iterates themselves are pinned to the reference's
// library by lbfgs_trace.cpp).
#include <math.h>
#include <stdio.h>

#include "lbfgs.h"

static void report(const char* name, int ret, double fx, int iters, int evals, const std::vector<double>& x) {
  printf("%s ret=%d fx=%.17g iters=%d evals=%d x=", name, ret, fx, iters, evals);
  for (double v : x) printf(" %.17g", v);
  printf("\n");
}

int main() {
  {  // extended Rosenbrock, n = 20, the classic start
    const int n = 20;
    std::vector<double> x(n);
    for (int i = 0; i < n; i += 2) { x[i] = -1.2; x[i + 1] = 1.0; }
    int evals = 0, iters = 0;
    double fx = 0, prev = INFINITY;
    bool monotone = true;
    int ret = crf_amd::lbfgs_minimize(
        n, x.data(), &fx,
        [&](const double* p, double* g, int len, double) {
          double f = 0;
          evals++;
          for (int i = 0; i < len; i += 2) {
            double t1 = 1.0 - p[i], t2 = 10.0 * (p[i + 1] - p[i] * p[i]);
            g[i + 1] = 20.0 * t2;
            g[i] = -2.0 * (p[i] * g[i + 1] + t1);
            f += t1 * t1 + t2 * t2;
          }
          return f;
        },
        [&](const double*, const double*, double f, double, double, double, int, int k, int) {
          iters = k;
          if (f > prev) monotone = false;   // every accepted point satisfies the sufficient-decrease condition
          prev = f;
          return 0;
        });
    report("rosenbrock", ret, fx, iters, evals, x);
    printf("rosenbrock_monotone %d\n", monotone ? 1 : 0);
  }
  {  // ill-conditioned convex quadratic 0.5 sum c_i (x_i - m_i)^2, c_i = 10^(i/3)
    const int n = 12;
    std::vector<double> x(n, 0.0);
    int evals = 0, iters = 0;
    double fx = 0;
    int ret = crf_amd::lbfgs_minimize(
        n, x.data(), &fx,
        [&](const double* p, double* g, int len, double) {
          double f = 0;
          evals++;
          for (int i = 0; i < len; i++) {
            double c = pow(10.0, i / 3.0), d = p[i] - (i - 5.5);
            g[i] = c * d;
            f += 0.5 * c * d * d;
          }
          return f;
        },
        [&](const double*, const double*, double, double, double, double, int, int k, int) { iters = k; return 0; });
    report("quadratic", ret, fx, iters, evals, x);
  }
  {  // a start that already is the minimum
    std::vector<double> x(3, 0.0);
    double fx = 1;
    int evals = 0;
    int ret = crf_amd::lbfgs_minimize(3, x.data(), &fx, [&](const double* p, double* g, int len, double) {
      evals++;
      double f = 0;
      for (int i = 0; i < len; i++) { g[i] = 2 * p[i]; f += p[i] * p[i]; }
      return f;
    }, nullptr);
    report("at_minimum", ret, fx, 0, evals, x);
  }
  {  // the progress callback stops the run after 3 iterations; x is the last accepted point
    const int n = 4;
    std::vector<double> x(n, 3.0);
    double fx = 0;
    int evals = 0, iters = 0;
    int ret = crf_amd::lbfgs_minimize(
        n, x.data(), &fx,
        [&](const double* p, double* g, int len, double) {
          evals++;
          double f = 0;
          for (int i = 0; i < len; i++) { double e = exp(p[i] * (i + 1) * 0.3); g[i] = (i + 1) * 0.3 * e - 1.0; f += e - p[i]; }
          return f;
        },
        [&](const double*, const double*, double, double, double, double, int, int k, int) { iters = k; return k >= 3; });
    report("stopped", ret, fx, iters, evals, x);
  }
  return 0;
}
