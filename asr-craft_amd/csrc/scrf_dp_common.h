// scrf_dp_common.h -- device helpers shared by the wavefront-per-utterance DP kernels
// (scrf_dp.hip: log-domain recursion; scrf_dplin.hip: scaled linear-domain recursion).
#ifndef SCRF_DP_COMMON_H_
#define SCRF_DP_COMMON_H_

#include "scrf_kernels.h"

#include <float.h>
#include <math.h>

#define DP_WPB 12  // wavefronts (utterances) per workgroup: 3 per SIMD, one workgroup per CU (6-wave groups do not pair up on a CU)
// wavefronts per workgroup the LDS allows: 12 when they fit, else 8 / 4 / 2 / 1 (L = 64 with D >= 17 needs that)
static inline uint32_t dp_waves_per_block(size_t shared_bytes, size_t per_wave_bytes) {
  const uint32_t opts[5] = {DP_WPB, 8, 4, 2, 1};
  for (int i = 0; i < 5; i++)
    if (shared_bytes + opts[i] * per_wave_bytes <= 156 * 1024) return opts[i];
  return 1;
}

__device__ __forceinline__ double rdlane(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// wave-wide sum of doubles on the DPP path (quad permutes, row shifts, the two row broadcasts; lanes
// outside a shift read 0), total read back from lane 63 as a wave-uniform value: 18 VALU
// instructions and no LDS crossbar round trips
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
  return v + __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v) {
  v = dpp_add_f64<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v = dpp_add_f64<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
  v = dpp_add_f64<0x114, 0xf>(v);   // row_shr:4
  v = dpp_add_f64<0x118, 0xf>(v);   // row_shr:8
  v = dpp_add_f64<0x142, 0xa>(v);   // row_bcast:15
  v = dpp_add_f64<0x143, 0xc>(v);   // row_bcast:31
  return rdlane(v, 63);
}

// wave-wide maximum on the same DPP path; lanes outside a shift keep their own value (bound_ctrl off, old = v)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max_f64_dpp(double v) {
  v = dpp_max_f64<0xb1, 0xf>(v);    // quad_perm:[1,0,3,2]
  v = dpp_max_f64<0x4e, 0xf>(v);    // quad_perm:[2,3,0,1]
  v = dpp_max_f64<0x114, 0xf>(v);   // row_shr:4
  v = dpp_max_f64<0x118, 0xf>(v);   // row_shr:8
  v = dpp_max_f64<0x142, 0xa>(v);   // row_bcast:15
  v = dpp_max_f64<0x143, 0xc>(v);   // row_bcast:31
  return rdlane(v, 63);
}

// exp(x) for x <= 0 (including -inf -> 0): the log-sum-exp terms are always max-shifted, so the
// overflow/NaN handling of the library exp is dead weight in the recursion's inner loop.
// Cody-Waite reduction by ln2 (hi/lo) + degree-13 Taylor/Horner on |r| <= ln2/2 (truncation
// 4e-18), scaled with v_ldexp_f64 (which flushes to 0 below the subnormal range): ~1 ulp.
__device__ __forceinline__ double exp_nonpos(double x) {
  x = fmax(x, -1000.0);  // also maps -inf; exp(-1000) underflows to 0 through ldexp
  const double k = rint(x * 1.4426950408889634);
  double r = fma(k, -6.93147180369123816490e-01, x);
  r = fma(k, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;                 // 1/13!
  p = fma(p, r, 2.0876756987868100e-09);             // 1/12!
  p = fma(p, r, 2.5052108385441720e-08);             // 1/11!
  p = fma(p, r, 2.7557319223985888e-07);             // 1/10!
  p = fma(p, r, 2.7557319223985893e-06);             // 1/9!
  p = fma(p, r, 2.4801587301587302e-05);             // 1/8!
  p = fma(p, r, 1.9841269841269841e-04);             // 1/7!
  p = fma(p, r, 1.3888888888888889e-03);             // 1/6!
  p = fma(p, r, 8.3333333333333332e-03);             // 1/5!
  p = fma(p, r, 4.1666666666666664e-02);             // 1/4!
  p = fma(p, r, 1.6666666666666666e-01);             // 1/3!
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)k);
}

// sum_c bcast(a, c) * Em[c*L + lc]; four partial sums break the FMA dependency chain.
// AS = address space of Em is known at compile time (LDS or global), never a flat pointer.
template <class PTR>
__device__ __forceinline__ double matvec_bcast(const double a, PTR Em, const int L, const int lc) {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int c = 0;
  for (; c + 4 <= L; c += 4) {
    const double e0 = Em[(c + 0) * L + lc], e1 = Em[(c + 1) * L + lc];
    const double e2 = Em[(c + 2) * L + lc], e3 = Em[(c + 3) * L + lc];
    s0 = fma(rdlane(a, c + 0), e0, s0);
    s1 = fma(rdlane(a, c + 1), e1, s1);
    s2 = fma(rdlane(a, c + 2), e2, s2);
    s3 = fma(rdlane(a, c + 3), e3, s3);
  }
  for (; c < L; c++) s0 = fma(rdlane(a, c), Em[c * L + lc], s0);
  return (s0 + s1) + (s2 + s3);
}

#endif  // SCRF_DP_COMMON_H_
