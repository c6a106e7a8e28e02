// scrf_lse.h -- workgroup-wide column log-sum-exp of the log-domain workgroup kernels (k_fb, k_fb_segtrans):
// max-shifted like the reference's logAdd(double*, double max, int) (utils/CRF_LogMath.cpp:102-125).
#ifndef SCRF_LSE_H_
#define SCRF_LSE_H_

#include "scrf_common.h"

struct FbLse {
  int L, G, g, j;
  bool active;
  double* red_m;
  double* red_s;
};

template <class VAL>
__device__ __forceinline__ double col_lse(const FbLse& c, int n_i, VAL val, int* err) {
  // returns LSE_i val(i, j) to the threads of group 0 (others get garbage); 3 barriers
  double m = -INFINITY;
  if (c.active)
    for (int i = c.g; i < n_i; i += c.G) m = fmax(m, val(i, c.j, true));
  if (c.active) c.red_m[c.g * c.L + c.j] = m;
  __syncthreads();
  double mm = -INFINITY, s = 0.0;
  if (c.active) {
    for (int gg = 0; gg < c.G; gg++) mm = fmax(mm, c.red_m[gg * c.L + c.j]);
    for (int i = c.g; i < n_i; i += c.G) s += exp(val(i, c.j, false) - mm);
    c.red_s[c.g * c.L + c.j] = s;
  }
  __syncthreads();
  double r = 0.0;
  if (c.active && c.g == 0) {
    double tot = 0.0;
    for (int gg = 0; gg < c.G; gg++) tot += c.red_s[gg * c.L + c.j];
    if (!(tot > 0.0) || isinf(tot) || isnan(tot)) *err = SCRF_ERR_NUMERIC;  // logE(0) / NaN / Inf
    r = mm + log(tot);
  }
  return r;
}

#endif  // SCRF_LSE_H_
