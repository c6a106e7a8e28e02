// lbfgs.h -- limited-memory BFGS with a More'-Thuente line search, for CRF_LBFGSTrainer.
//
// The reference trains `crf_train_method=lbfgs` through libLBFGS (Naoaki Okazaki's C port of Nocedal's
// L-BFGS, vendored as CRF/src/utils/lbfgs.{c,h}; the trainer passes NULL parameters, so the library's
// defaults apply: m = 6, epsilon = 1e-5, More'-Thuente line search with ftol = 1e-4, gtol = 0.9,
// xtol = 1e-16, at most 40 evaluations per search, steps in [1e-20, 1e20], no orthant-wise term,
// no delta-based stop, no iteration cap -- CRF_LBFGSTrainer.cpp:55-62).  This file is an independent
// implementation of the two published algorithms with those defaults and the same callback protocol:
//   * Nocedal, "Updating quasi-Newton matrices with limited storage" (1980): two-loop recursion over the
//     last m (s, y) pairs, initial scaling ys / yy, first step 1 / ||d||, then 1;
//   * More' & Thuente, "Line search algorithms with guaranteed sufficient decrease" (1994): the
//     bracketing / interpolation search (MINPACK-2's dcsrch / dcstep) for the strong Wolfe conditions.
// Same iterates as the library are NOT claimed (libLBFGS cannot be run here against it: parity unpinned);
// the tests check descent, the Wolfe conditions of every accepted step and the optimum reached.
// Vector algebra is plain host code on lambda_len doubles: O(m n) per iteration next to a full pass of the
// GPU over the training set.
#ifndef CRF_AMD_LBFGS_H_
#define CRF_AMD_LBFGS_H_

#include <math.h>

#include <algorithm>
#include <functional>
#include <vector>

namespace crf_amd {

struct LbfgsParams {
  int m = 6;
  double epsilon = 1e-5;
  int max_iterations = 0;      // 0: until convergence or the progress callback stops it
  int max_linesearch = 40;
  double min_step = 1e-20, max_step = 1e20;
  double ftol = 1e-4, gtol = 0.9, xtol = 1e-16;
};

enum LbfgsStatus {
  LBFGS_OK = 0,
  LBFGS_STOP = 1,                 // the progress callback asked to stop
  LBFGS_ALREADY_MINIMIZED = 2,
  LBFGSERR_INCREASEGRADIENT = -994,
  LBFGSERR_MAXIMUMLINESEARCH = -998,
  LBFGSERR_MINIMUMSTEP = -1000,
  LBFGSERR_MAXIMUMSTEP = -999,
  LBFGSERR_ROUNDING_ERROR = -1001,
  LBFGSERR_WIDTHTOOSMALL = -996,
  LBFGSERR_MAXIMUMITERATION = -997,
};

// evaluate(x, g, n, step) -> f ;  progress(x, g, fx, xnorm, gnorm, step, n, k, ls) -> nonzero to stop
typedef std::function<double(const double*, double*, int, double)> LbfgsEvaluate;
typedef std::function<int(const double*, const double*, double, double, double, double, int, int, int)> LbfgsProgress;

namespace lbfgs_detail {

inline double dot(const std::vector<double>& a, const std::vector<double>& b) {
  double s = 0.0;
  for (size_t i = 0; i < a.size(); i++) s += a[i] * b[i];
  return s;
}
inline double dotp(const double* a, const double* b, int n) {
  double s = 0.0;
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

// One update of the interval of uncertainty and of the trial step (More' & Thuente section 4; MINPACK-2
// dcstep): x = best step so far, y = the other end point, t = the trial step.
inline int trial_interval(double* x, double* fx, double* dx, double* y, double* fy, double* dy, double* t, double* ft,
                          double* dt, double tmin, double tmax, bool* brackt) {
  if (*brackt) {
    if (*t <= std::min(*x, *y) || std::max(*x, *y) <= *t) return -1;      // trial value out of the interval
    if (0.0 <= *dx * (*t - *x)) return -2;                                  // the function does not decrease from x
    if (tmax < tmin) return -3;
  }
  const bool dsign = (*dt) * (*dx / fabs(*dx)) < 0.0;
  bool bound;
  double newt;
  const double stx = *x, fxv = *fx, dxv = *dx, stp = *t, fp = *ft, dp = *dt;
  if (fxv < fp) {
    // case 1: higher function value -- the minimum is bracketed; cubic through both points against the
    // quadratic through f(x), f'(x), f(t)
    *brackt = true;
    bound = true;
    const double theta = 3.0 * (fxv - fp) / (stp - stx) + dxv + dp;
    const double sc = std::max(fabs(theta), std::max(fabs(dxv), fabs(dp)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dxv / sc) * (dp / sc));
    if (stp < stx) gamma = -gamma;
    const double pn = (gamma - dxv) + theta, qd = ((gamma - dxv) + gamma) + dp;
    const double mc = stx + (pn / qd) * (stp - stx);
    const double mq = stx + ((dxv / ((fxv - fp) / (stp - stx) + dxv)) / 2.0) * (stp - stx);
    newt = fabs(mc - stx) < fabs(mq - stx) ? mc : mc + (mq - mc) / 2.0;
  } else if (dsign) {
    // case 2: lower value, derivatives of opposite sign -- bracketed; cubic against the secant step
    *brackt = true;
    bound = false;
    const double theta = 3.0 * (fxv - fp) / (stp - stx) + dxv + dp;
    const double sc = std::max(fabs(theta), std::max(fabs(dxv), fabs(dp)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dxv / sc) * (dp / sc));
    if (stp > stx) gamma = -gamma;
    const double pn = (gamma - dp) + theta, qd = ((gamma - dp) + gamma) + dxv;
    const double mc = stp + (pn / qd) * (stx - stp);
    const double mq = stp + (dp / (dp - dxv)) * (stx - stp);
    newt = fabs(mc - stp) > fabs(mq - stp) ? mc : mq;
  } else if (fabs(dp) < fabs(dxv)) {
    // case 3: lower value, same sign, the derivative shrinks -- the cubic may have no minimizer beyond t
    bound = true;
    const double theta = 3.0 * (fxv - fp) / (stp - stx) + dxv + dp;
    const double sc = std::max(fabs(theta), std::max(fabs(dxv), fabs(dp)));
    double gamma = sc * sqrt(std::max(0.0, (theta / sc) * (theta / sc) - (dxv / sc) * (dp / sc)));
    if (stp > stx) gamma = -gamma;
    const double pn = (gamma - dp) + theta, qd = (gamma + (dxv - dp)) + gamma;
    const double r = pn / qd;
    double mc;
    if (r < 0.0 && gamma != 0.0) mc = stp + r * (stx - stp);
    else mc = stp > stx ? tmax : tmin;
    const double mq = stp + (dp / (dp - dxv)) * (stx - stp);
    if (*brackt) newt = fabs(stp - mc) < fabs(stp - mq) ? mc : mq;
    else newt = fabs(stp - mc) > fabs(stp - mq) ? mc : mq;
  } else {
    // case 4: lower value, same sign, the derivative does not shrink
    bound = false;
    if (*brackt) {
      const double sty = *y, fyv = *fy, dyv = *dy;
      const double theta = 3.0 * (fp - fyv) / (sty - stp) + dyv + dp;
      const double sc = std::max(fabs(theta), std::max(fabs(dyv), fabs(dp)));
      double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dyv / sc) * (dp / sc));
      if (stp > sty) gamma = -gamma;
      const double pn = (gamma - dp) + theta, qd = ((gamma - dp) + gamma) + dyv;
      newt = stp + (pn / qd) * (sty - stp);
    } else {
      newt = stp > stx ? tmax : tmin;
    }
  }
  // the new interval
  if (*fx < *ft) {
    *y = *t; *fy = *ft; *dy = *dt;
  } else {
    if (dsign) { *y = *x; *fy = *fx; *dy = *dx; }
    *x = *t; *fx = *ft; *dx = *dt;
  }
  if (tmax < newt) newt = tmax;
  if (newt < tmin) newt = tmin;
  if (*brackt && bound) {   // keep the step within 2/3 of the interval from x
    const double mq = *x + 0.66 * (*y - *x);
    if (*x < *y) { if (mq < newt) newt = mq; }
    else { if (newt < mq) newt = mq; }
  }
  *t = newt;
  return 0;
}

// line search along s from xp: on success returns the number of evaluations and leaves x, f, g at the accepted point
inline int line_search(int n, std::vector<double>& x, double* f, std::vector<double>& g, const std::vector<double>& s,
                       double* stp, const std::vector<double>& xp, const LbfgsEvaluate& evaluate, const LbfgsParams& pr) {
  int count = 0, uinfo = 0;
  if (*stp <= 0.0) return -1;
  const double dginit = dot(g, s);
  if (0.0 < dginit) return LBFGSERR_INCREASEGRADIENT;
  bool brackt = false, stage1 = true;
  const double finit = *f, dgtest = pr.ftol * dginit;
  double width = pr.max_step - pr.min_step, prev_width = 2.0 * width;
  double stx = 0.0, sty = 0.0, fx = finit, fy = finit, dgx = dginit, dgy = dginit;
  for (;;) {
    double stmin, stmax;
    if (brackt) { stmin = std::min(stx, sty); stmax = std::max(stx, sty); }
    else { stmin = stx; stmax = *stp + 4.0 * (*stp - stx); }
    if (*stp < pr.min_step) *stp = pr.min_step;
    if (pr.max_step < *stp) *stp = pr.max_step;
    // unusual termination ahead: take the best point found so far
    if ((brackt && ((*stp <= stmin || stmax <= *stp) || pr.max_linesearch <= count + 1 || uinfo != 0)) ||
        (brackt && (stmax - stmin <= pr.xtol * stmax)))
      *stp = stx;
    for (int i = 0; i < n; i++) x[i] = xp[i] + *stp * s[i];
    *f = evaluate(x.data(), g.data(), n, *stp);
    const double dg = dot(g, s);
    const double ftest1 = finit + *stp * dgtest;
    ++count;
    if (brackt && ((*stp <= stmin || stmax <= *stp) || uinfo != 0)) return LBFGSERR_ROUNDING_ERROR;
    if (*stp == pr.max_step && *f <= ftest1 && dg <= dgtest) return LBFGSERR_MAXIMUMSTEP;
    if (*stp == pr.min_step && (ftest1 < *f || dgtest <= dg)) return LBFGSERR_MINIMUMSTEP;
    if (brackt && (stmax - stmin) <= pr.xtol * stmax) return LBFGSERR_WIDTHTOOSMALL;
    if (pr.max_linesearch <= count) return LBFGSERR_MAXIMUMLINESEARCH;
    if (*f <= ftest1 && fabs(dg) <= pr.gtol * (-dginit)) return count;   // sufficient decrease and curvature
    // the first stage looks for a point with a lower value of the modified function and a non-negative derivative
    if (stage1 && *f <= ftest1 && std::min(pr.ftol, pr.gtol) * dginit <= dg) stage1 = false;
    if (stage1 && ftest1 < *f && *f <= fx) {
      double fm = *f - *stp * dgtest, fxm = fx - stx * dgtest, fym = fy - sty * dgtest;
      double dgm = dg - dgtest, dgxm = dgx - dgtest, dgym = dgy - dgtest;
      uinfo = trial_interval(&stx, &fxm, &dgxm, &sty, &fym, &dgym, stp, &fm, &dgm, stmin, stmax, &brackt);
      fx = fxm + stx * dgtest; fy = fym + sty * dgtest;
      dgx = dgxm + dgtest; dgy = dgym + dgtest;
    } else {
      double ft = *f, dt = dg;
      uinfo = trial_interval(&stx, &fx, &dgx, &sty, &fy, &dgy, stp, &ft, &dt, stmin, stmax, &brackt);
    }
    if (brackt) {   // force a sufficient decrease of the interval
      if (0.66 * prev_width <= fabs(sty - stx)) *stp = stx + 0.5 * (sty - stx);
      prev_width = width;
      width = fabs(sty - stx);
    }
  }
}

}  // namespace lbfgs_detail

// minimises f from x (length n, updated in place); *fx_out = the last accepted function value
inline int lbfgs_minimize(int n, double* x_io, double* fx_out, const LbfgsEvaluate& evaluate, const LbfgsProgress& progress,
                          const LbfgsParams& pr = LbfgsParams()) {
  using namespace lbfgs_detail;
  const int m = pr.m;
  std::vector<double> x(x_io, x_io + n), xp(n), g(n), gp(n), d(n);
  struct Pair { std::vector<double> s, y; double ys = 0.0, alpha = 0.0; };
  std::vector<Pair> lm(m);
  for (auto& p : lm) { p.s.assign(n, 0.0); p.y.assign(n, 0.0); }
  double fx = evaluate(x.data(), g.data(), n, 0.0);
  for (int i = 0; i < n; i++) d[i] = -g[i];
  double xnorm = sqrt(dot(x, x)), gnorm = sqrt(dot(g, g));
  if (xnorm < 1.0) xnorm = 1.0;
  int ret = LBFGS_OK;
  if (gnorm / xnorm <= pr.epsilon) {
    ret = LBFGS_ALREADY_MINIMIZED;
  } else {
    double step = 1.0 / sqrt(dot(d, d));
    int k = 1, end = 0;
    for (;;) {
      xp = x;
      gp = g;
      const int ls = line_search(n, x, &fx, g, d, &step, xp, evaluate, pr);
      if (ls < 0) {   // back to the previous point
        x = xp;
        g = gp;
        ret = ls;
        break;
      }
      xnorm = sqrt(dot(x, x));
      gnorm = sqrt(dot(g, g));
      if (progress && progress(x.data(), g.data(), fx, xnorm, gnorm, step, n, k, ls)) { ret = LBFGS_STOP; break; }
      if (xnorm < 1.0) xnorm = 1.0;
      if (gnorm / xnorm <= pr.epsilon) { ret = LBFGS_OK; break; }
      if (pr.max_iterations != 0 && pr.max_iterations < k + 1) { ret = LBFGSERR_MAXIMUMITERATION; break; }
      // s_{k+1} = x_{k+1} - x_k, y_{k+1} = g_{k+1} - g_k
      Pair& it = lm[end];
      for (int i = 0; i < n; i++) { it.s[i] = x[i] - xp[i]; it.y[i] = g[i] - gp[i]; }
      const double ys = dot(it.y, it.s), yy = dot(it.y, it.y);
      it.ys = ys;
      // two-loop recursion: d = -H g
      const int bound = m <= k ? m : k;
      ++k;
      end = (end + 1) % m;
      for (int i = 0; i < n; i++) d[i] = -g[i];
      int j = end;
      for (int i = 0; i < bound; i++) {
        j = (j + m - 1) % m;
        Pair& q = lm[j];
        q.alpha = dot(q.s, d) / q.ys;
        for (int c = 0; c < n; c++) d[c] -= q.alpha * q.y[c];
      }
      const double scale = ys / yy;
      for (int c = 0; c < n; c++) d[c] *= scale;
      for (int i = 0; i < bound; i++) {
        Pair& q = lm[j];
        const double beta = dot(q.y, d) / q.ys;
        for (int c = 0; c < n; c++) d[c] += (q.alpha - beta) * q.s[c];
        j = (j + 1) % m;
      }
      step = 1.0;
    }
  }
  if (fx_out) *fx_out = fx;
  std::copy(x.begin(), x.end(), x_io);
  return ret;
}

}  // namespace crf_amd
#endif  // CRF_AMD_LBFGS_H_
