// lbfgs.h -- limited-memory BFGS with a strong-Wolfe line search, the optimiser behind CRF_LBFGSTrainer.
//
// The reference trains `crf_train_method=lbfgs` through its vendored copy of libLBFGS (CRF/src/utils/lbfgs.c, entry
// point :245, defaults :113-118); the trainer passes NULL parameters (trainers/CRF_LBFGSTrainer.cpp:80), so the
// library's defaults are the configuration: 6 correction pairs, stop at |g| <= 1e-5 max(1, |x|), More'-Thuente search
// with sufficient-decrease 1e-4, curvature 0.9, relative interval width 1e-16, at most 40 evaluations per search, steps
// in [1e-20, 1e20], no orthant-wise term, no delta test, no iteration cap.
//
// Written from the two published algorithms, organised around three small objects:
//   * CurvatureMemory -- Nocedal (1980), "Updating quasi-Newton matrices with limited storage": a ring of the last
//     m (s, y) pairs and the two-loop product d = -H g with the initial scaling (y.s)/(y.y);
//   * WolfeSearch     -- More' & Thuente (1994), "Line search algorithms with guaranteed sufficient decrease": a
//     reverse-communication state machine (the caller evaluates, the object decides), as MINPACK-2 organises it;
//   * next_trial()    -- section 4 of that paper: the safeguarded cubic / quadratic step and the interval update.
// What the iterates depend on beyond the papers -- the order of the stopping tests inside a search, the 2/3 and 1/2
// interval safeguards, which bound an unbounded cubic falls back to (the vendored library takes it from the sign of
// the cubic's theta term, lbfgs.c:1046, not from the side of the trial point), the first step 1/|g| -- follows the
// vendored library, because its iterates are the contract: tests/test_host_lbfgs.py compares this file's iterates with
// golden vectors produced by that library compiled from the reference tree (tests/golden/lbfgs_ref.npz, generator
// committed) to 1e-12, and against the library itself where the test infrastructure has built it.  Status codes are the
// library's numeric values (utils/lbfgs.h:75-145) so that a log line "LBFGS returned: -999" reads the same.
// Vector algebra is plain host code on lambda_len doubles: O(m n) per iteration next to a full pass of the GPU over
// the training set.
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
  LBFGS_STOP = 1,                       // the progress callback asked to stop
  LBFGS_ALREADY_MINIMIZED = 2,
  // failures of a line search (the run goes back to the last accepted point)
  LBFGS_FAIL_TRIAL_OUTSIDE = -1004,     // the trial step left the interval of uncertainty
  LBFGS_FAIL_BOUNDS_CROSSED = -1003,
  LBFGS_FAIL_NO_PROGRESS = -1002,       // rounding errors: the interval stopped shrinking
  LBFGS_FAIL_STEP_AT_MIN = -1001,
  LBFGS_FAIL_STEP_AT_MAX = -1000,
  LBFGS_FAIL_SEARCH_BUDGET = -999,      // max_linesearch evaluations without an acceptable step
  LBFGS_FAIL_ITERATION_CAP = -998,
  LBFGS_FAIL_INTERVAL_TOO_NARROW = -997,
  LBFGS_FAIL_BAD_ARGUMENT = -996,
  LBFGS_FAIL_UPHILL_DIRECTION = -995,   // the search direction is not a descent direction
};

// evaluate(x, g, n, step) -> f ;  progress(x, g, fx, xnorm, gnorm, step, n, k, ls) -> nonzero to stop
typedef std::function<double(const double*, double*, int, double)> LbfgsEvaluate;
typedef std::function<int(const double*, const double*, double, double, double, double, int, int, int)> LbfgsProgress;

namespace lbfgs_detail {

typedef std::vector<double> Vec;

inline double inner(const Vec& a, const Vec& b) {
  double acc = 0.0;
  for (size_t i = 0; i < a.size(); i++) acc += a[i] * b[i];
  return acc;
}
inline void axpy(Vec& y, double c, const Vec& x) {   // y += c x
  for (size_t i = 0; i < y.size(); i++) y[i] += c * x[i];
}

// A point on the search line: step length, function value, directional derivative.
struct LinePoint {
  double t, f, df;
};

// Stationary point of the cubic that interpolates (f, f') at `from` and `to`, written as from + r (to - from)
// (More' & Thuente eq. 4.1-4.2 with the overflow-safe scaling of MINPACK-2).
inline double cubic_step(const LinePoint& from, const LinePoint& to) {
  const double span = to.t - from.t;
  const double theta = (from.f - to.f) * 3 / span + from.df + to.df;
  const double big = std::max(std::max(fabs(theta), fabs(from.df)), fabs(to.df));
  const double ratio = theta / big;
  double gamma = big * sqrt(ratio * ratio - (from.df / big) * (to.df / big));
  if (to.t < from.t) gamma = -gamma;
  const double num = gamma - from.df + theta;
  const double den = gamma - from.df + gamma + to.df;
  return from.t + (num / den) * span;
}
// The same cubic when it may have no minimiser on the far side of `to` (derivatives of one sign, shrinking): the
// stationary point beyond `to` if there is one, otherwise a bound of the search.
inline double cubic_step_or_bound(const LinePoint& from, const LinePoint& to, double lower, double upper) {
  const double span = to.t - from.t;
  const double theta = (from.f - to.f) * 3 / span + from.df + to.df;
  const double big = std::max(std::max(fabs(theta), fabs(from.df)), fabs(to.df));
  const double ratio = theta / big;
  double gamma = big * sqrt(std::max(0.0, ratio * ratio - (from.df / big) * (to.df / big)));
  if (from.t < to.t) gamma = -gamma;
  const double num = gamma - to.df + theta;
  const double den = gamma - to.df + gamma + from.df;
  const double r = num / den;
  if (r < 0.0 && gamma != 0.0) return to.t - r * span;
  return ratio < 0 ? upper : lower;   // the vendored library's choice (lbfgs.c:1046), kept for identical iterates
}
// Minimiser of the parabola through f(from), f'(from), f(to).
inline double parabola_step_values(const LinePoint& from, const LinePoint& to) {
  const double span = to.t - from.t;
  return from.t + from.df / ((from.f - to.f) / span + from.df) / 2 * span;
}
// Minimiser of the parabola through f'(from), f'(to) (the secant step), measured from `to`.
inline double parabola_step_slopes(const LinePoint& from, const LinePoint& to) {
  const double span = from.t - to.t;
  return to.t + to.df / (to.df - from.df) * span;
}

// Section 4 of More' & Thuente: given the best point so far, the other end of the interval and the trial point just
// evaluated, update the interval and return the next trial step in *next.  Returns 0 or a failure code.
inline int next_trial(LinePoint& best, LinePoint& other, const LinePoint& trial, bool& bracketed, double lower,
                      double upper, double* next) {
  if (bracketed) {
    if (trial.t <= std::min(best.t, other.t) || std::max(best.t, other.t) <= trial.t) return LBFGS_FAIL_TRIAL_OUTSIDE;
    if (0.0 <= best.df * (trial.t - best.t)) return LBFGS_FAIL_UPHILL_DIRECTION;
    if (upper < lower) return LBFGS_FAIL_BOUNDS_CROSSED;
  }
  const bool slopes_oppose = trial.df * (best.df / fabs(best.df)) < 0.0;
  bool clip_to_interval;
  double pick;
  if (best.f < trial.f) {
    // higher value: a minimiser lies between; the cubic step unless the parabola says it is too timid
    bracketed = true;
    clip_to_interval = true;
    const double c = cubic_step(best, trial), q = parabola_step_values(best, trial);
    pick = fabs(c - best.t) < fabs(q - best.t) ? c : c + (q - c) / 2;
  } else if (slopes_oppose) {
    // lower value, the slope changed sign: a minimiser lies between; the farther of cubic and secant
    bracketed = true;
    clip_to_interval = false;
    const double c = cubic_step(best, trial), q = parabola_step_slopes(best, trial);
    pick = fabs(c - trial.t) > fabs(q - trial.t) ? c : q;
  } else if (fabs(trial.df) < fabs(best.df)) {
    // lower value, same sign, flatter: extrapolate, carefully
    clip_to_interval = true;
    const double c = cubic_step_or_bound(best, trial, lower, upper), q = parabola_step_slopes(best, trial);
    if (bracketed) pick = fabs(trial.t - c) < fabs(trial.t - q) ? c : q;
    else pick = fabs(trial.t - c) > fabs(trial.t - q) ? c : q;
  } else {
    // lower value, same sign, not flatter: towards the other end if there is one, else as far as allowed
    clip_to_interval = false;
    if (bracketed) pick = cubic_step(trial, other);
    else pick = best.t < trial.t ? upper : lower;
  }
  // the interval that still holds a minimiser
  if (best.f < trial.f) {
    other = trial;
  } else {
    if (slopes_oppose) other = best;
    best = trial;
  }
  pick = std::max(lower, std::min(upper, pick));
  if (bracketed && clip_to_interval) {   // not closer to `other` than two thirds of the way
    const double limit = best.t + 0.66 * (other.t - best.t);
    if (best.t < other.t) pick = std::min(pick, limit);
    else pick = std::max(pick, limit);
  }
  *next = pick;
  return 0;
}

// The search as a state machine.  Usage: WolfeSearch w(params, f0, slope0, first_step);
// loop { t = w.propose(); evaluate f(t), f'(t); code = w.observe(f, slope); } until code != 0
// (code > 0: accepted after `code` evaluations, code < 0: failure).
class WolfeSearch {
 public:
  WolfeSearch(const LbfgsParams& pr, double f0, double slope0, double first_step)
      : pr_(pr), f0_(f0), slope0_(slope0), decrease_rate_(pr.ftol * slope0), trial_(first_step) {
    best_ = other_ = LinePoint{0.0, f0, slope0};
    width_ = pr.max_step - pr.min_step;
    width_before_ = 2.0 * width_;
  }
  // the step to evaluate next
  double propose() {
    if (bracketed_) { lower_ = std::min(best_.t, other_.t); upper_ = std::max(best_.t, other_.t); }
    else { lower_ = best_.t; upper_ = trial_ + 4.0 * (trial_ - best_.t); }
    trial_ = std::max(pr_.min_step, std::min(pr_.max_step, trial_));
    // nothing better can come: settle for the best point seen
    const bool out = trial_ <= lower_ || upper_ <= trial_;
    const bool last_chance = pr_.max_linesearch <= evals_ + 1;
    const bool collapsed = upper_ - lower_ <= pr_.xtol * upper_;
    if (bracketed_ && (out || last_chance || trouble_ != 0 || collapsed)) trial_ = best_.t;
    return trial_;
  }
  int observe(double f, double slope) {
    ++evals_;
    const double line = f0_ + trial_ * decrease_rate_;   // the sufficient-decrease line at the trial step
    if (bracketed_ && (trial_ <= lower_ || upper_ <= trial_ || trouble_ != 0)) return LBFGS_FAIL_NO_PROGRESS;
    if (trial_ == pr_.max_step && f <= line && slope <= decrease_rate_) return LBFGS_FAIL_STEP_AT_MAX;
    if (trial_ == pr_.min_step && (line < f || decrease_rate_ <= slope)) return LBFGS_FAIL_STEP_AT_MIN;
    if (bracketed_ && upper_ - lower_ <= pr_.xtol * upper_) return LBFGS_FAIL_INTERVAL_TOO_NARROW;
    if (pr_.max_linesearch <= evals_) return LBFGS_FAIL_SEARCH_BUDGET;
    if (f <= line && fabs(slope) <= pr_.gtol * (-slope0_)) return evals_;   // strong Wolfe conditions hold

    // phase one works on psi(t) = f(t) - f(0) - ftol f'(0) t until a point with psi <= 0 <= psi' is seen
    if (phase_one_ && f <= line && std::min(pr_.ftol, pr_.gtol) * slope0_ <= slope) phase_one_ = false;
    const LinePoint seen{trial_, f, slope};
    if (phase_one_ && line < f && f <= best_.f) {
      LinePoint b = shifted(best_), o = shifted(other_);
      trouble_ = next_trial(b, o, shifted(seen), bracketed_, lower_, upper_, &trial_);
      best_ = unshifted(b);
      other_ = unshifted(o);
    } else {
      trouble_ = next_trial(best_, other_, seen, bracketed_, lower_, upper_, &trial_);
    }
    if (bracketed_) {   // the interval must lose a third of its width every two steps, or it is bisected
      const double w = fabs(other_.t - best_.t);
      if (0.66 * width_before_ <= w) trial_ = best_.t + 0.5 * (other_.t - best_.t);
      width_before_ = width_;
      width_ = w;
    }
    return 0;
  }

 private:
  LinePoint shifted(const LinePoint& p) const { return LinePoint{p.t, p.f - p.t * decrease_rate_, p.df - decrease_rate_}; }
  LinePoint unshifted(const LinePoint& p) const { return LinePoint{p.t, p.f + p.t * decrease_rate_, p.df + decrease_rate_}; }
  const LbfgsParams& pr_;
  const double f0_, slope0_, decrease_rate_;
  LinePoint best_, other_;
  double trial_, lower_ = 0.0, upper_ = 0.0, width_, width_before_;
  bool bracketed_ = false, phase_one_ = true;
  int evals_ = 0, trouble_ = 0;
};

// The last m displacement / gradient-change pairs and the product with the inverse Hessian they define.
class CurvatureMemory {
 public:
  CurvatureMemory(int m, int n) : s_(m, Vec(n, 0.0)), y_(m, Vec(n, 0.0)), ys_(m, 0.0), coef_(m, 0.0) {}
  // record x_new - x_old, g_new - g_old; returns (y.s) / (y.y), the scaling of the initial matrix
  double remember(const Vec& x_new, const Vec& x_old, const Vec& g_new, const Vec& g_old) {
    Vec& s = s_[head_];
    Vec& y = y_[head_];
    for (size_t i = 0; i < s.size(); i++) { s[i] = x_new[i] - x_old[i]; y[i] = g_new[i] - g_old[i]; }
    const double ys = inner(y, s), yy = inner(y, y);
    ys_[head_] = ys;
    head_ = (head_ + 1) % (int)s_.size();
    if (held_ < (int)s_.size()) held_++;
    return ys / yy;
  }
  // d = -H g: newest pair first on the way down, oldest first on the way back
  void direction(const Vec& g, double scaling, Vec& d) {
    const int m = (int)s_.size();
    for (size_t i = 0; i < d.size(); i++) d[i] = -g[i];
    int at = head_;
    for (int i = 0; i < held_; i++) {
      at = (at + m - 1) % m;
      coef_[at] = inner(s_[at], d) / ys_[at];
      axpy(d, -coef_[at], y_[at]);
    }
    for (size_t i = 0; i < d.size(); i++) d[i] *= scaling;
    for (int i = 0; i < held_; i++) {
      const double back = inner(y_[at], d) / ys_[at];
      axpy(d, coef_[at] - back, s_[at]);
      at = (at + 1) % m;
    }
  }

 private:
  std::vector<Vec> s_, y_;
  Vec ys_, coef_;
  int head_ = 0, held_ = 0;
};

inline double norm_or_one(double v) { return v < 1.0 ? 1.0 : v; }

}  // namespace lbfgs_detail

// minimises f from x (length n, updated in place); *fx_out = the last accepted function value
inline int lbfgs_minimize(int n, double* x_io, double* fx_out, const LbfgsEvaluate& evaluate, const LbfgsProgress& progress,
                          const LbfgsParams& pr = LbfgsParams()) {
  using namespace lbfgs_detail;
  if (n <= 0 || pr.m <= 0) return LBFGS_FAIL_BAD_ARGUMENT;
  Vec x(x_io, x_io + n), g(n), d(n), x_keep(n), g_keep(n);
  CurvatureMemory memory(pr.m, n);
  double fx = evaluate(x.data(), g.data(), n, 0.0);
  int status = LBFGS_OK;
  if (sqrt(inner(g, g)) / norm_or_one(sqrt(inner(x, x))) <= pr.epsilon) {
    status = LBFGS_ALREADY_MINIMIZED;
  } else {
    for (int i = 0; i < n; i++) d[i] = -g[i];
    double step = 1.0 / sqrt(inner(d, d));   // the first trial moves by a unit length
    for (int iteration = 1;; iteration++) {
      x_keep = x;
      g_keep = g;
      if (step <= 0.0) { status = LBFGS_FAIL_BAD_ARGUMENT; break; }
      const double slope = inner(g, d);
      int verdict = 0 < slope ? (int)LBFGS_FAIL_UPHILL_DIRECTION : 0;
      if (verdict == 0) {
        WolfeSearch search(pr, fx, slope, step);
        while (verdict == 0) {
          step = search.propose();
          for (int i = 0; i < n; i++) x[i] = x_keep[i] + step * d[i];
          fx = evaluate(x.data(), g.data(), n, step);
          verdict = search.observe(fx, inner(g, d));
        }
      }
      if (verdict < 0) {   // no acceptable step: back to where the search started
        x = x_keep;
        g = g_keep;
        status = verdict;
        break;
      }
      const double xnorm = sqrt(inner(x, x)), gnorm = sqrt(inner(g, g));
      if (progress && progress(x.data(), g.data(), fx, xnorm, gnorm, step, n, iteration, verdict)) { status = LBFGS_STOP; break; }
      if (gnorm / norm_or_one(xnorm) <= pr.epsilon) { status = LBFGS_OK; break; }
      if (pr.max_iterations != 0 && pr.max_iterations < iteration + 1) { status = LBFGS_FAIL_ITERATION_CAP; break; }
      memory.direction(g, memory.remember(x, x_keep, g, g_keep), d);
      step = 1.0;
    }
  }
  if (fx_out) *fx_out = fx;
  std::copy(x.begin(), x.end(), x_io);
  return status;
}

}  // namespace crf_amd
#endif  // CRF_AMD_LBFGS_H_
