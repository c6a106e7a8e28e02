/* lbfgs_problems.h -- the three objective functions of the L-BFGS iterate pin, in plain C so that the program that
 * drives the reference's vendored libLBFGS (tests/golden/lbfgs_ref_trace.c, run in the build container only) and the
 * program that drives asr-craft_amd/host/lbfgs.h (tests/host/lbfgs_trace.cpp) evaluate bit-identical values.
 *   0 quadratic : 0.5 sum c_i (x_i - m_i)^2, c_i = 10^(i/3), n = 12, start 0
 *   1 rosenbrock: extended Rosenbrock, n = 20, start (-1.2, 1, ...)
 *   2 lse50     : log sum_j exp(a_j . x + b_j) + 0.5e-2 |x|^2, 50 variables, 80 terms, integer-generated a, b, start 0
 *   3 steep10   : sum_i exp(k_i x_i) - k_i x_i + 0.1 sum_i cos(x_i x_{i+1}), k_i = 0.5 (i + 1), n = 10, start 2: steep
 *                 walls, so the searches overshoot, bracket and interpolate (all four step-selection cases occur)
 */
#ifndef LBFGS_PROBLEMS_H_
#define LBFGS_PROBLEMS_H_
#include <math.h>

#define LP_NPROB 4
static const char* const lp_name[LP_NPROB] = {"quadratic", "rosenbrock", "lse50", "steep10"};
static const int lp_dim[LP_NPROB] = {12, 20, 50, 10};

static void lp_start(int prob, double* x) {
  int i;
  for (i = 0; i < lp_dim[prob]; i++) x[i] = 0.0;
  if (prob == 1)
    for (i = 0; i < lp_dim[prob]; i += 2) { x[i] = -1.2; x[i + 1] = 1.0; }
  if (prob == 3)
    for (i = 0; i < lp_dim[prob]; i++) x[i] = 2.0;
}

#define LP_TERMS 80
static double lp_coef(unsigned j, unsigned i) {   /* exact small dyadic rationals in [-1, 1) */
  unsigned h = (j * 2654435761u) ^ (i * 40503u + 977u);
  h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
  return ((int)(h & 1023u) - 512) / 512.0;
}

static double lp_eval(int prob, const double* x, double* g) {
  const int n = lp_dim[prob];
  double f = 0.0;
  int i, j;
  if (prob == 0) {
    for (i = 0; i < n; i++) {
      const double c = pow(10.0, i / 3.0), d = x[i] - (i - 5.5);
      g[i] = c * d;
      f += 0.5 * c * d * d;
    }
  } else if (prob == 1) {
    for (i = 0; i < n; i += 2) {
      const double t1 = 1.0 - x[i], t2 = 10.0 * (x[i + 1] - x[i] * x[i]);
      g[i + 1] = 20.0 * t2;
      g[i] = -2.0 * (x[i] * g[i + 1] + t1);
      f += t1 * t1 + t2 * t2;
    }
  } else if (prob == 3) {
    for (i = 0; i < n; i++) {
      const double k = 0.5 * (i + 1), e = exp(k * x[i]);
      g[i] = k * e - k;
      f += e - k * x[i];
    }
    for (i = 0; i + 1 < n; i++) {
      const double p = x[i] * x[i + 1], sn = sin(p);
      f += 0.1 * cos(p);
      g[i] -= 0.1 * sn * x[i + 1];
      g[i + 1] -= 0.1 * sn * x[i];
    }
  } else {
    double z[LP_TERMS], zmax = -1e300, sum = 0.0;
    for (j = 0; j < LP_TERMS; j++) {
      double s = lp_coef(j, 1000u);
      for (i = 0; i < n; i++) s += lp_coef(j, i) * x[i];
      z[j] = s;
      if (s > zmax) zmax = s;
    }
    for (j = 0; j < LP_TERMS; j++) { z[j] = exp(z[j] - zmax); sum += z[j]; }
    f = zmax + log(sum);
    for (i = 0; i < n; i++) {
      double s = 0.0;
      for (j = 0; j < LP_TERMS; j++) s += z[j] * lp_coef(j, i);
      g[i] = s / sum + 1e-2 * x[i];
      f += 0.5e-2 * x[i] * x[i];
    }
  }
  return f;
}

/* one line per accepted iterate, hex floats: "it <prob> <k> <ls> <step> <fx> <xnorm> <gnorm> <x...>" */
#include <stdio.h>
static void lp_print_iter(int prob, int k, int ls, double step, double fx, double xnorm, double gnorm, const double* x) {
  int i;
  printf("it %s %d %d %a %a %a %a", lp_name[prob], k, ls, step, fx, xnorm, gnorm);
  for (i = 0; i < lp_dim[prob]; i++) printf(" %a", x[i]);
  printf("\n");
}
#endif
