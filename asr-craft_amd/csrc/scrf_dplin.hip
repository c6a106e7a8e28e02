// scrf_dplin.hip -- scaled linear-domain forward/backward for L <= 64 labels (training path).
//
// The log-domain recursion (scrf_dp.hip, and the reference: nodes/CRF_StdSegStateNode_WithoutDurLab_
// WithoutSegTransFtr.cpp computeAlpha :123-245, computeBeta :395-466, computeExpF :616-949) spends
// D exp() per label per frame inside a serial chain.  Here every quantity is carried as a
// mantissa vector times exp(scalar log-scale), the scalar being per frame:
//
//   es[row][l] = exp(S[row][l] - smax[row]),  smax[row] = max_l S[row][l]            (k_exp_rows)
//   exp(aPT[t'][l]) = p[t'][l] * exp(gp[t'])          exp(alpha[t][l]) = a[t][l] * exp(ga[t])
//   a[t][l] = sum_d c_d * p[t-d][l] * es[(t,d)][l],   c_d = exp(gp[t-d] + smax[(t,d)] - ga[t]),
//   ga[t]   = max_d (gp[t-d] + smax[(t,d)])           (one exp per (t, d), not per (t, d, l))
//   p[t][n] = 2^-k * sum_c a[t][c] * E[c][n],         gp[t] = ga[t] + shift + k ln2   (k: exponent of the max)
//
// and symmetrically backwards.  These are the reference's sums with the exponentials factored;
// results differ from the log-domain recursion by rounding only (~1e-15 relative).  Terms more
// than ~700 nats below their row's / frame's maximum flush to zero instead of to ~1e-300: they are
// below the resolution of every output.  A frame whose whole vector flushes (transition scores
// spanning > ~700 nats) raises SCRF_ERR_NUMERIC like the reference's log(0) check.
//
// The alpha-with-duration array (N_seg x L) is never written: gamma is rebuilt from p, es and
// beta in k_post_lin, R = Y - gamma overwrites es in place.
#include "scrf_dp_common.h"

#include <stdlib.h>

#define LN2_HI 6.93147180369123816490e-01
#define LN2_LO 1.90821492927058770002e-10

__device__ __forceinline__ double shfl_f64(double v, int src) {
  int lo = __shfl(__double2loint(v), src), hi = __shfl(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
// wave-wide max on the DPP path (no LDS crossbar round trips): quad permutes, row shifts and the
// two row broadcasts leave the result in lane 63, which is then read back as a scalar
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, dpp_i32<0xb1, 0xf>(v, v));    // quad_perm:[1,0,3,2]
  v = max(v, dpp_i32<0x4e, 0xf>(v, v));    // quad_perm:[2,3,0,1]
  v = max(v, dpp_i32<0x114, 0xf>(v, v));   // row_shr:4
  v = max(v, dpp_i32<0x118, 0xf>(v, v));   // row_shr:8
  v = max(v, dpp_i32<0x142, 0xa>(v, v));   // row_bcast:15
  v = max(v, dpp_i32<0x143, 0xc>(v, v));   // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}
// float max through the integer path: for x >= 0 the bit pattern orders like the value; negative
// values are mapped to the mirrored order first (standard total-order trick)
__device__ __forceinline__ float wave_max_f32_dpp(float x) {
  int b = __float_as_int(x);
  b = b >= 0 ? b : (int)(0x80000000u - (unsigned)b);   // monotone map of floats onto signed ints
  int m = wave_max_i32(b);
  m = m >= 0 ? m : (int)(0x80000000u - (unsigned)m);
  return __int_as_float(m);
}
// largest high word over the wave of a non-negative double vector: orders like the values do (to
// 32 bits), costs what a float max costs, and keeps the full double exponent range
__device__ __forceinline__ int wave_max_hi(double v) { return wave_max_i32(__double2hiint(v)); }

// sum_c a[c] * Em[c*L + lc] with the a-vector broadcast through a private LDS line: one uniform
// 16-byte read serves two labels, against two v_readlane + hazard padding per label
template <class PTR>
__device__ __forceinline__ double matvec_lds(const double a, double* abuf, PTR Em, const int L, const int lc,
                                             const int lane) {
  abuf[lane] = a;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int c = 0;
  for (; c + 8 <= L; c += 8) {
    const double2 a01 = *(const double2*)(abuf + c), a23 = *(const double2*)(abuf + c + 2);
    const double2 a45 = *(const double2*)(abuf + c + 4), a67 = *(const double2*)(abuf + c + 6);
    const double e0 = Em[(c + 0) * L + lc], e1 = Em[(c + 1) * L + lc], e2 = Em[(c + 2) * L + lc];
    const double e3 = Em[(c + 3) * L + lc], e4 = Em[(c + 4) * L + lc], e5 = Em[(c + 5) * L + lc];
    const double e6 = Em[(c + 6) * L + lc], e7 = Em[(c + 7) * L + lc];
    s0 = fma(a01.x, e0, s0); s1 = fma(a01.y, e1, s1); s2 = fma(a23.x, e2, s2); s3 = fma(a23.y, e3, s3);
    s0 = fma(a45.x, e4, s0); s1 = fma(a45.y, e5, s1); s2 = fma(a67.x, e6, s2); s3 = fma(a67.y, e7, s3);
  }
  for (; c < L; c++) s0 = fma(abuf[c], Em[c * L + lc], s0);
  return (s0 + s1) + (s2 + s3);
}
// unbiased binary exponent from a high word; *bad = zero / subnormal / inf / nan
__device__ __forceinline__ int hi_exp(int h, int* bad) {
  if (h < 0x00100000 || h >= 0x7ff00000) *bad = 1;
  return ((h >> 20) & 0x7ff) - 1023;
}

// ------------------------------------------------------------------------------------------
// k_true_scores: s_true[f] = S[(t, ld)][al] for labelled frames (the numerator's state term,
// gradbuilder :388-469), read before the scores are exponentiated in place.
// ------------------------------------------------------------------------------------------
__global__ void k_true_scores(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                              uint64_t n_frames, const double* __restrict__ S, double* __restrict__ s_true) {
  const uint64_t fi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= n_frames) return;
  const uint32_t L = lay.L, D = lay.D;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
  double v = 0.0;
  if (lab != SCRF_LAB_BAD && lab < L * D) {
    const uint32_t al = lab % L, ld = lab / L + 1;
    if (ld <= scrf_node_max_dur(t, D))
      v = S[((bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D) + ld - 1) * L + al];
  }
  s_true[fi] = v;
}

// ------------------------------------------------------------------------------------------
// k_exp_rows: in place S[row][:] -> exp(S[row][:] - smax[row]); 16 lanes per row, L <= 256.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_exp_rows(double* __restrict__ S, uint64_t n_rows, uint32_t L,
                                                  double* __restrict__ smax) {
  const uint64_t row = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const uint32_t sub = threadIdx.x & 15;
  if (row >= n_rows) return;
  double* p = S + row * L;
  double m = -INFINITY;
  for (uint32_t l = sub; l < L; l += 16) m = fmax(m, p[l]);
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) m = fmax(m, shfl_f64(m, (int)((threadIdx.x & 63) ^ o)));
  for (uint32_t l = sub; l < L; l += 16) p[l] = exp_nonpos(p[l] - m);
  if (sub == 0) smax[row] = m;
}

// ------------------------------------------------------------------------------------------
// k_dp_lin: one wavefront per (utterance, direction); lane = label.
// Outputs per frame: a (alpha mantissa), ga; p (alpha-plus-trans mantissa), gp; b (beta mantissa),
// gb; sd (sum over durations, mantissa), gsd.
// ------------------------------------------------------------------------------------------
template <int DMAX, int MPF, int LC>
__global__ __launch_bounds__(DP_WPB * 64, 3) void k_dp_lin(
    ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ ES,
    const double* __restrict__ smax, const double* __restrict__ E, const double* __restrict__ ET,
    const double* __restrict__ mshift, double* __restrict__ a_g, double* __restrict__ ga_g,
    double* __restrict__ p_g, double* __restrict__ gp_g, double* __restrict__ b_g, double* __restrict__ gb_g,
    double* __restrict__ sd_g, double* __restrict__ gsd_g, double* __restrict__ zx_out, int* __restrict__ status) {
  extern __shared__ double dsm[];
  // LC: the label count as a compile-time constant (0 = read it from the layout): every LDS address of the
  // transition step and of the ring becomes base + immediate.  The wave index is made explicitly wave-uniform:
  // utterance, length, bases and ring slots then live in SGPRs and the loop is scalar-controlled.
  const int L = LC ? LC : (int)lay.L, D = lay.D;
  const int LL = L * L;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int dir = blockIdx.x & 1;  // 0 forward, 1 backward
  const uint32_t ul = (blockIdx.x >> 1) * (blockDim.x >> 6) + wave;   // blockDim.x / 64 utterances per workgroup
  double* Es = dsm;                                         // [L*L] (time-invariant transitions only)
  double* ring = dsm + (MPF ? 0 : LL) + (size_t)wave * (D * L + 128);  // [D][L] mantissas, private to this wavefront
  double* abuf = ring + D * L;   // [64] broadcast line of the matvec operand
  double* cbuf = abuf + 64;      // [64] broadcast line of the per-duration scales
  if (!MPF) {
    const double* src = dir ? ET : E;
    for (int i = threadIdx.x; i < LL; i += blockDim.x) Es[i] = src[i];
    __syncthreads();
  }
  if (ul >= n_utts) return;
  const uint32_t u = u0 + ul;
  const int T = (int)bv.T[u];
  if (T == 0) return;
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* ESu = ES + s_base * L;
  const double* smu = smax + s_base;
  const bool act = lane < L;
  const int lc = act ? lane : L - 1;  // clamped label: idle lanes shadow the last label
  const double sh0 = MPF ? 0.0 : mshift[0];
  int err = 0;
  double gslot = 0.0;  // lane j: log-scale of the vector in ring slot j

  if (dir == 0) {
    // ---------------------------------------------------------------- forward
    double* au = a_g + f_base * L;
    double* pu = p_g + f_base * L;
    double a = ESu[lc];        // node 0: the only window is the initial segment of length 1
    double ga = smu[0];
    if (act) __builtin_nontemporal_store(a, &au[lane]);
    // the per-frame log-scales are wave-uniform: lane (frame & 63) keeps them and 64 frames go out in one store
    // (a store instruction per frame and scalar costs the wavefront as much as a 384-byte one)
    double ga_keep = ga, gp_keep = 0.0;
    int rpos = D - 1;
    // the nd windows ending at t: independent of the recursion
    auto load_windows = [&](int t, double (&es)[DMAX], double& smx) {
      const int nd = (int)scrf_node_max_dur(t, D);
      const uint64_t base = scrf_seg_base(t, D);
      if (nd == DMAX) {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) es[d0] = ESu[(base + d0) * L + lc];
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          const double x = ESu[(base + (d0 < nd ? d0 : 0)) * L + lc];
          es[d0] = (d0 < nd) ? x : 0.0;
        }
      }
      smx = smu[base + (lane < nd ? lane : 0)];
    };
    for (int t = 1; t < T; t++) {
      rpos = (rpos + 1 == D) ? 0 : rpos + 1;  // ring slot of node t-1
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      const bool full = (nd == DMAX) && (np == DMAX);  // steady state: every duration has a predecessor
      double es[DMAX], smx;
      load_windows(t, es, smx);   // issued first, covered by the transition step
      // transition out of node t-1: p = 2^-k * (a . E)
      double usum, sh = sh0;
      if (MPF) {
        usum = matvec_lds(a, abuf, E + (f_base + t) * (size_t)LL, L, lc, lane);
        sh = mshift[f_base + t];
      } else {
        usum = matvec_lds(a, abuf, Es, L, lc, lane);
      }
      const int k = hi_exp(wave_max_hi(act ? usum : 0.0), &err);
      const double p = ldexp(usum, -k);
      const double gp = ga + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      ring[rpos * L + lc] = p;  // idle lanes rewrite lane L-1's value with the same number
      if (lane == rpos) gslot = gp;
      if (act) __builtin_nontemporal_store(p, &pu[(size_t)(t - 1) * L + lane]);
      if (lane == ((t - 1) & 63)) gp_keep = gp;
      if (((t - 1) & 63) == 63) gp_g[f_base + (t - 1 - 63) + lane] = gp_keep;   // 64 frames' log-scales in one store
      // per-duration scales: lane d0 looks at predecessor node t-1-d0 (ring slot rpos-d0)
      int myslot = rpos - lane;
      if (myslot < 0) myslot += D;
      const double gprev = shfl_f64(gslot, (lane < np) ? myslot : 0);
      double x = (lane < np) ? gprev + smx : smx;   // lane == np (< nd): the initial segment
      x = (lane < nd) ? x : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      const double c = exp_nonpos(x - G);
      cbuf[lane] = c;
      double acc0 = 0.0, acc1 = 0.0;
      if (full) {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = rpos - d0;
          slot += (slot < 0) ? D : 0;
          const double pv = (d0 == 0) ? p : ring[slot * L + lc];
          const double w = es[d0] * cbuf[d0];
          if (d0 & 1) acc1 = fma(pv, w, acc1); else acc0 = fma(pv, w, acc0);
        }
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = rpos - d0;
          if (slot < 0) slot += D;
          const double r = (d0 == 0) ? p : ring[(d0 < np ? slot : rpos) * L + lc];
          const double pv = (d0 < np) ? r : 1.0;
          const double w = es[d0] * cbuf[d0];
          if (d0 & 1) acc1 = fma(pv, w, acc1); else acc0 = fma(pv, w, acc0);
        }
      }
      a = acc0 + acc1;
      ga = G;
      if (act) __builtin_nontemporal_store(a, &au[(size_t)t * L + lane]);
      if (lane == (t & 63)) ga_keep = ga;
      if ((t & 63) == 63) ga_g[f_base + (t - 63) + lane] = ga_keep;
    }
    if (((T - 1) & 63) != 63 && lane <= ((T - 1) & 63)) ga_g[f_base + ((T - 1) & ~63) + lane] = ga_keep;
    if (T >= 2 && ((T - 2) & 63) != 63 && lane <= ((T - 2) & 63)) gp_g[f_base + ((T - 2) & ~63) + lane] = gp_keep;
    // Zx = log sum_l exp(alpha[T-1][l])  (computeAlphaSum)
    const double tot = wave_sum_f64(act ? a : 0.0);
    const double Zx = ga + log(tot);
    if (!(Zx == Zx) || isinf(Zx)) err = 1;
    if (lane == 0) zx_out[u] = Zx;
  } else {
    // ---------------------------------------------------------------- backward
    double* bu = b_g + f_base * L;
    double* sdu = sd_g + f_base * L;
    int tpos = (T - 1) % D;
    ring[tpos * L + lc] = 1.0;  // setTailBeta: beta[T-1] = 0
    if (lane == tpos) gslot = 0.0;
    if (act) { bu[(size_t)(T - 1) * L + lane] = 1.0; sdu[(size_t)(T - 1) * L + lane] = 0.0; }
    if (lane == 0) { gb_g[f_base + T - 1] = 0.0; gsd_g[f_base + T - 1] = 0.0; }
    double gb_keep = 0.0, gsd_keep = 0.0;   // lane (frame & 63); frame T-1's zeros included
    // window (t+1+d0, d0+1): starts at t+1, ends at node t+1+d0
    auto load_windows = [&](int t, double (&es)[DMAX], double& smx) {
      const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
      const uint64_t sb = scrf_seg_base(t + 1, D);
      uint64_t myrow;
      if ((nn == DMAX) && (t + 1 >= D)) {
        // every node t+1.. carries D windows, so the window sits at row seg_base(t+1) + d0*(D+1)
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) es[d0] = ESu[(sb + (uint64_t)d0 * (DMAX + 1)) * L + lc];
        myrow = sb + (uint64_t)(lane < nn ? lane : 0) * (DMAX + 1);
      } else {
        uint64_t r = sb;
        myrow = sb;
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          const bool ok = d0 < nn;
          const double x = ESu[(ok ? r + d0 : 0) * L + lc];
          es[d0] = ok ? x : 0.0;
          if (lane == d0 && ok) myrow = r + d0;
          r += scrf_node_max_dur(t + 1 + d0, D);
        }
      }
      smx = smu[myrow];
    };
    for (int t = T - 2; t >= 0; t--) {
      const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
      tpos = (tpos == 0) ? D - 1 : tpos - 1;  // ring slot of node t
      double es[DMAX], smx;
      load_windows(t, es, smx);
      int myslot = tpos + lane + 1;  // node t + d0 + 1
      if (myslot >= D) myslot -= D;
      const double gnext = shfl_f64(gslot, (lane < nn) ? myslot : 0);
      const double x = (lane < nn) ? gnext + smx : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      const double c = exp_nonpos(x - G);
      cbuf[lane] = c;
      double acc0 = 0.0, acc1 = 0.0;
      if (nn == DMAX) {   // steady state: every duration has a successor node
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = tpos + d0 + 1;
          slot -= (slot >= D) ? D : 0;
          const double bv_ = ring[slot * L + lc];
          const double w = es[d0] * cbuf[d0];
          if (d0 & 1) acc1 = fma(bv_, w, acc1); else acc0 = fma(bv_, w, acc0);
        }
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = tpos + d0 + 1;
          if (slot >= D) slot -= D;
          const double bv_ = ring[((d0 < nn) ? slot : tpos) * L + lc];
          const double w = es[d0] * cbuf[d0];   // es = 0 past nn
          if (d0 & 1) acc1 = fma((d0 < nn) ? bv_ : 0.0, w, acc1); else acc0 = fma((d0 < nn) ? bv_ : 0.0, w, acc0);
        }
      }
      const double sd = acc0 + acc1;
      double w, sh = sh0;
      if (MPF) {
        w = matvec_lds(sd, abuf, ET + (f_base + t + 1) * (size_t)LL, L, lc, lane);
        sh = mshift[f_base + t + 1];
      } else {
        w = matvec_lds(sd, abuf, Es, L, lc, lane);
      }
      const int k = hi_exp(wave_max_hi(act ? w : 0.0), &err);
      const double b = ldexp(w, -k);
      const double gb = G + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      ring[tpos * L + lc] = b;
      if (lane == tpos) gslot = gb;
      if (act) { __builtin_nontemporal_store(sd, &sdu[(size_t)t * L + lane]); __builtin_nontemporal_store(b, &bu[(size_t)t * L + lane]); }
      if (lane == (t & 63)) { gsd_keep = G; gb_keep = gb; }
      if ((t & 63) == 0 && t + lane < T) { gsd_g[f_base + t + lane] = gsd_keep; gb_g[f_base + t + lane] = gb_keep; }   // frames t .. t+63 (descending walk)
    }
  }
  if (__any(err != 0) && lane == 0) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// ------------------------------------------------------------------------------------------
// k_dp_lin_mv: the same recursion with SEVERAL wavefronts per (utterance, direction) -- time-invariant transitions,
// L <= 64.  One wavefront per sweep (k_dp_lin) runs a step as one dependent chain -- 48 x 48 transition product, wave
// maximum, exp, D ring reads with their multiply-adds: 3.4 us -- and 11 resident sweeps per CU are all its LDS rings
// allow.  Here a workgroup of NW wavefronts shares ONE ring: wavefront w takes the durations d0 = w, w + NW, ... (a
// quarter of the rows to load, of the ring reads and of the multiply-adds) and the rows c = w CQ .. of the transition
// product, whose matrix entries it keeps in registers; the partial sums meet in LDS at two barriers per step.  The
// chain per step is a third as long, a step's loads are requested a step ahead with 14 registers, and the per-frame
// vectors are stored by wavefront 0 only.  Arithmetic: the same products, summed per wavefront first (results differ
// from k_dp_lin's by reassociation, ~1e-16 relative).
// MEASURED (config 2, 4096 utterances, same box): 6.08-6.32 ms against 6.04-6.14 ms for k_dp_lin -- four sweeps resident
// per CU (128 VGPRs) at 2.5 us per step move the same bytes per microsecond as eleven at 6.8 us: a step's rows are
// requested one step ahead and __syncthreads() drains the vector-memory counter, so every step still waits out an HBM
// round trip.  Variants that keep two steps of rows in flight were built and are NOT in the tree: an s_barrier behind
// s_waitcnt lgkmcnt(0) only, with the register sets rotated by copies (9.3 ms: a copy of a register whose load is in
// flight waits for the load) or in fixed roles with the loop unrolled three times (hipcc still places
// s_waitcnt vmcnt(<= 9) ahead of each step, and the code grows to 100 KB); capping the kernel at 128 VGPRs with the
// third set spills into the loop (16 ms).  Hence opt-in: SCRF_DPLIN_MV=1 (the parity tests cover it that way).
// ------------------------------------------------------------------------------------------
// Round 4: MPF = 1 takes per-frame transition matrices (each wavefront's quarter of the frame's matrix is requested a step
// ahead into a second register set: 148-182 registers), and the launcher picks that form BY ITSELF when the launch has too
// few sweeps to fill the chip (launch_dp_lin).
#define DPV_NW 4
template <int DMAX, int MPF>
__global__ __launch_bounds__(64 * DPV_NW, MPF ? 2 : 4) void k_dp_lin_mv(
    ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ ES,
    const double* __restrict__ smax, const double* __restrict__ E, const double* __restrict__ ET,
    const double* __restrict__ mshift, double* __restrict__ a_g, double* __restrict__ ga_g,
    double* __restrict__ p_g, double* __restrict__ gp_g, double* __restrict__ b_g, double* __restrict__ gb_g,
    double* __restrict__ sd_g, double* __restrict__ gsd_g, double* __restrict__ zx_out, int* __restrict__ status) {
  constexpr int NW = DPV_NW;
  constexpr int NDW = (DMAX + NW - 1) / NW;   // durations per wavefront
  constexpr int CQ = 64 / NW;                 // transition rows per wavefront (L <= 64)
  extern __shared__ double dsm[];
  const int L = (int)lay.L, D = (int)lay.D;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int dir = blockIdx.x & 1;
  const uint32_t ul = blockIdx.x >> 1;
  double* ring = dsm;                                   // [D][L], shared by the workgroup
  double* part = ring + (((size_t)D * L + 1) & ~(size_t)1);   // [NW][64] partial duration sums
  double* upart = part + NW * 64;                       // [NW][64] partial transition products
  double* abuf = upart + NW * 64 + wave * 64;           // this wavefront's broadcast line of the matvec operand
  double* cbuf = upart + NW * 64 + NW * 64 + wave * 64; // this wavefront's per-duration scales
  if (ul >= n_utts) return;
  const uint32_t u = u0 + ul;
  const int T = (int)bv.T[u];
  if (T == 0) return;
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* ESu = ES + s_base * L;
  const double* smu = smax + s_base;
  const bool act = lane < L;
  const int lc = act ? lane : L - 1;
  const bool w0 = wave == 0;
  const double sh0 = MPF ? 0.0 : mshift[0];
  const size_t LL = (size_t)L * L;
  int err = 0;
  double gslot = 0.0;   // lane j: log-scale of the vector in ring slot j (every wavefront keeps the same copy)
  // this wavefront's rows c0 .. c0 + CQ - 1 of the transition matrix, column lc
  const int c0 = wave * CQ;
  double er[CQ], er_n[CQ];
  auto load_rows = [&](const double* Em, double (&e)[CQ]) {
#pragma unroll
    for (int i = 0; i < CQ; i++) e[i] = (c0 + i < L) ? Em[(size_t)(c0 + i) * L + lc] : 0.0;
  };
  if (!MPF) load_rows(dir ? ET : E, er);
  // sum over all rows of v[c] * Em[c][lc]: this wavefront's share, then the four shares in wavefront order
  auto matvec = [&](const double v) {
    abuf[lane] = v;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < CQ; i += 2) {
      const double2 a01 = *(const double2*)(abuf + c0 + i);
      s0 = fma(a01.x, er[i], s0);
      s1 = fma(a01.y, er[i + 1], s1);
    }
    upart[wave * 64 + lane] = s0 + s1;
    __syncthreads();
    double tot = upart[lane];
#pragma unroll
    for (int w = 1; w < NW; w++) tot += upart[w * 64 + lane];
    return tot;
  };
  auto dursum = [&](const double acc) {
    part[wave * 64 + lane] = acc;
    __syncthreads();
    double tot = part[lane];
#pragma unroll
    for (int w = 1; w < NW; w++) tot += part[w * 64 + lane];
    return tot;
  };

  if (dir == 0) {
    // ---------------------------------------------------------------- forward
    double* au = a_g + f_base * L;
    double* pu = p_g + f_base * L;
    double a = ESu[lc];
    double ga = smu[0];
    if (w0 && act) __builtin_nontemporal_store(a, &au[lane]);
    double ga_keep = ga, gp_keep = 0.0;
    int rpos = D - 1;
    // this wavefront's windows ending at t (durations d0 = wave + NW i) and the row maxima of all of them (lane = d0)
    auto load_windows = [&](int t, double (&es)[NDW], double& smx) {
      const int tt = t < T ? t : T - 1;
      const int nd = (int)scrf_node_max_dur(tt, D);
      const uint64_t base = scrf_seg_base(tt, D);
#pragma unroll
      for (int i = 0; i < NDW; i++) {
        const int d0 = wave + NW * i;
        const double x = ESu[(base + (d0 < nd ? d0 : 0)) * L + lc];
        es[i] = (d0 < nd) ? x : 0.0;
      }
      smx = smu[base + (lane < nd ? lane : 0)];
    };
    double es_n[NDW], smx_n;
    load_windows(1, es_n, smx_n);
    double sh_n = sh0;
    if (MPF && T > 1) { load_rows(E + (f_base + 1) * LL, er_n); sh_n = mshift[f_base + 1]; }
    for (int t = 1; t < T; t++) {
      rpos = (rpos + 1 == D) ? 0 : rpos + 1;  // ring slot of node t-1
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      double es[NDW], smx = smx_n;
#pragma unroll
      for (int i = 0; i < NDW; i++) es[i] = es_n[i];
      load_windows(t + 1, es_n, smx_n);   // the next step's rows: a step of latency cover
      double sh = sh0;
      if (MPF) {   // node t's matrix (requested a step ago), then node t+1's
#pragma unroll
        for (int i = 0; i < CQ; i++) er[i] = er_n[i];
        sh = sh_n;
        const int tn = t + 1 < T ? t + 1 : T - 1;
        load_rows(E + (f_base + tn) * LL, er_n);
        sh_n = mshift[f_base + tn];
      }
      // transition out of node t-1: p = 2^-k * (a . E)
      const double usum = matvec(a);
      const int k = hi_exp(wave_max_hi(act ? usum : 0.0), &err);
      const double p = ldexp(usum, -k);
      const double gp = ga + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      if (w0) {
        ring[rpos * L + lc] = p;   // read by the others from the next step on (two barriers away)
        if (act) __builtin_nontemporal_store(p, &pu[(size_t)(t - 1) * L + lane]);
        if (lane == ((t - 1) & 63)) gp_keep = gp;
        if (((t - 1) & 63) == 63) gp_g[f_base + (t - 1 - 63) + lane] = gp_keep;
      }
      if (lane == rpos) gslot = gp;
      // per-duration scales: lane d0 looks at predecessor node t-1-d0 (ring slot rpos-d0)
      int myslot = rpos - lane;
      if (myslot < 0) myslot += D;
      const double gprev = shfl_f64(gslot, (lane < np) ? myslot : 0);
      double x = (lane < np) ? gprev + smx : smx;   // lane == np (< nd): the initial segment
      x = (lane < nd) ? x : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      cbuf[lane] = exp_nonpos(x - G);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int i = 0; i < NDW; i++) {
        const int d0 = wave + NW * i;
        if (d0 < DMAX) {
          int slot = rpos - d0;
          if (slot < 0) slot += D;
          const double r = (d0 == 0) ? p : ring[((d0 < np) ? slot : rpos) * L + lc];
          const double pv = (d0 < np) ? r : 1.0;
          const double w = es[i] * cbuf[d0 < D ? d0 : 0];   // es = 0 past nd
          if (i & 1) acc1 = fma(pv, w, acc1); else acc0 = fma(pv, w, acc0);
        }
      }
      a = dursum(acc0 + acc1);
      ga = G;
      if (w0) {
        if (act) __builtin_nontemporal_store(a, &au[(size_t)t * L + lane]);
        if (lane == (t & 63)) ga_keep = ga;
        if ((t & 63) == 63) ga_g[f_base + (t - 63) + lane] = ga_keep;
      }
    }
    if (w0) {
      if (((T - 1) & 63) != 63 && lane <= ((T - 1) & 63)) ga_g[f_base + ((T - 1) & ~63) + lane] = ga_keep;
      if (T >= 2 && ((T - 2) & 63) != 63 && lane <= ((T - 2) & 63)) gp_g[f_base + ((T - 2) & ~63) + lane] = gp_keep;
      const double tot = wave_sum_f64(act ? a : 0.0);
      const double Zx = ga + log(tot);
      if (!(Zx == Zx) || isinf(Zx)) err = 1;
      if (lane == 0) zx_out[u] = Zx;
    }
  } else {
    // ---------------------------------------------------------------- backward
    double* bu = b_g + f_base * L;
    double* sdu = sd_g + f_base * L;
    int tpos = (T - 1) % D;
    if (w0) ring[tpos * L + lc] = 1.0;  // setTailBeta: beta[T-1] = 0
    if (lane == tpos) gslot = 0.0;
    if (w0) {
      if (act) { bu[(size_t)(T - 1) * L + lane] = 1.0; sdu[(size_t)(T - 1) * L + lane] = 0.0; }
      if (lane == 0) { gb_g[f_base + T - 1] = 0.0; gsd_g[f_base + T - 1] = 0.0; }
    }
    // (no barrier here or at the end of a step: node t's vector, written by wavefront 0 after the step's two barriers, is
    // read at d0 = 0 by wavefront 0 itself in the next step and at d0 >= 1 by the others one or more steps -- two or
    // more barriers -- later; the slot it overwrites, node t + D, was last read before this step's first barrier)
    double gb_keep = 0.0, gsd_keep = 0.0;
    // window (t+1+d0, d0+1): starts at t+1, ends at node t+1+d0
    auto load_windows = [&](int t, double (&es)[NDW], double& smx) {
      const int tt = t >= 0 ? t : 0;
      const int nn = (T - 1 - tt <= D) ? T - 1 - tt : D;
      const uint64_t sb = scrf_seg_base(tt + 1, D);
      uint64_t myrow = sb;
      if ((nn == D) && (tt + 1 >= D)) {
        // every node tt+1.. carries D windows, so the window sits at row seg_base(tt+1) + d0*(D+1)
#pragma unroll
        for (int i = 0; i < NDW; i++) {
          const int d0 = wave + NW * i;
          const double x = ESu[(sb + (uint64_t)(d0 < nn ? d0 : 0) * (D + 1)) * L + lc];
          es[i] = (d0 < nn) ? x : 0.0;
        }
        myrow = sb + (uint64_t)(lane < nn ? lane : 0) * (D + 1);
      } else {
        uint64_t r = sb;
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          const bool ok = d0 < nn;
          if ((d0 % NW) == wave) {   // wave-uniform
            const double x = ESu[(ok ? r + d0 : 0) * L + lc];
            es[d0 / NW] = ok ? x : 0.0;
          }
          if (lane == d0 && ok) myrow = r + d0;
          r += scrf_node_max_dur(tt + 1 + d0 < T ? tt + 1 + d0 : T - 1, D);
        }
        if (DMAX % NW) {
#pragma unroll
          for (int i = 0; i < NDW; i++) if (wave + NW * i >= DMAX) es[i] = 0.0;
        }
      }
      smx = smu[myrow];
    };
    double es_n[NDW], smx_n;
    load_windows(T - 2, es_n, smx_n);
    double sh_n = sh0;
    if (MPF) { load_rows(ET + (f_base + T - 1) * LL, er_n); sh_n = mshift[f_base + T - 1]; }
    for (int t = T - 2; t >= 0; t--) {
      const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
      tpos = (tpos == 0) ? D - 1 : tpos - 1;  // ring slot of node t
      double es[NDW], smx = smx_n;
#pragma unroll
      for (int i = 0; i < NDW; i++) es[i] = es_n[i];
      load_windows(t - 1, es_n, smx_n);
      double sh = sh0;
      if (MPF) {   // node t+1's matrix (requested a step ago), then node t's
#pragma unroll
        for (int i = 0; i < CQ; i++) er[i] = er_n[i];
        sh = sh_n;
        load_rows(ET + (f_base + t) * LL, er_n);
        sh_n = mshift[f_base + t];
      }
      int myslot = tpos + lane + 1;  // node t + d0 + 1
      if (myslot >= D) myslot -= D;
      const double gnext = shfl_f64(gslot, (lane < nn) ? myslot : 0);
      const double x = (lane < nn) ? gnext + smx : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      cbuf[lane] = exp_nonpos(x - G);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int i = 0; i < NDW; i++) {
        const int d0 = wave + NW * i;
        if (d0 < DMAX) {
          int slot = tpos + d0 + 1;
          if (slot >= D) slot -= D;
          const double bv_ = ring[((d0 < nn) ? slot : tpos) * L + lc];
          const double w = es[i] * cbuf[d0 < D ? d0 : 0];   // es = 0 past nn
          if (i & 1) acc1 = fma((d0 < nn) ? bv_ : 0.0, w, acc1); else acc0 = fma((d0 < nn) ? bv_ : 0.0, w, acc0);
        }
      }
      const double sd = dursum(acc0 + acc1);
      const double wsum = matvec(sd);
      const int k = hi_exp(wave_max_hi(act ? wsum : 0.0), &err);
      const double b = ldexp(wsum, -k);
      const double gb = G + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      if (lane == tpos) gslot = gb;
      if (w0) {
        ring[tpos * L + lc] = b;
        if (act) { __builtin_nontemporal_store(sd, &sdu[(size_t)t * L + lane]); __builtin_nontemporal_store(b, &bu[(size_t)t * L + lane]); }
        if (lane == (t & 63)) { gsd_keep = G; gb_keep = gb; }
        if ((t & 63) == 0 && t + lane < T) { gsd_g[f_base + t + lane] = gsd_keep; gb_g[f_base + t + lane] = gb_keep; }
      }
    }
  }
  if (__any(err != 0) && lane == 0) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// ------------------------------------------------------------------------------------------
// k_dp_lin_mw: the same recursion for 64 < L <= 256: one workgroup of NW = ceil(L/64) wavefronts per
// (utterance, direction), lane = label.  The vector that feeds the L x L transition step is
// exchanged through LDS (two barriers per frame, one more for the wave maxima); everything a lane
// needs for the duration step is its own column of the ring, so that part needs no barrier.  The
// transition matrix stays in memory (L2-resident): rows are read coalesced.
// ------------------------------------------------------------------------------------------
// Round 4, LR > 0 (time-invariant transitions, L <= LR): the lane's column of the transition matrix lives in REGISTERS for
// the whole sweep.  One workgroup per CU is one wavefront per SIMD, i.e. all 512 vector + accumulation registers: 208 rows
// are 416 of them.  The transition step was 200 loads per lane and frame from L2, a few in flight at a time: 13 us per
// frame at L = 200, of which the loads' round trips were ~10.
template <int DMAX, int MPF, int LR, int TAIL>
__global__ __launch_bounds__(256) void k_dp_lin_mw(
    ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ ES,
    const double* __restrict__ smax, const double* __restrict__ E, const double* __restrict__ ET,
    const double* __restrict__ mshift, double* __restrict__ a_g, double* __restrict__ ga_g,
    double* __restrict__ p_g, double* __restrict__ gp_g, double* __restrict__ b_g, double* __restrict__ gb_g,
    double* __restrict__ sd_g, double* __restrict__ gsd_g, double* __restrict__ zx_out, int* __restrict__ status) {
  extern __shared__ double dsm[];
  const int L = lay.L, D = lay.D;
  const size_t LL = (size_t)L * L;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, NW = blockDim.x >> 6;
  const int dir = blockIdx.x & 1;  // 0 forward, 1 backward
  const uint32_t u = u0 + (blockIdx.x >> 1);
  double* ring = dsm;                 // [D][L] mantissas of the last D alpha-plus-trans / beta vectors
  double* abuf = ring + (((size_t)D * L + 1) & ~(size_t)1);  // [64*NW] operand of the transition step (16-byte aligned)
  double* gring = abuf + 64 * NW;     // [D] log-scales of the ring slots
  double* cbuf = gring + 64;          // [NW][64] per-duration scales, one line per wavefront
  int* red = (int*)(cbuf + 64 * NW);  // [NW] wave maxima (high words)
  double* redd = (double*)(red + 8);  // [NW] partial sums (Zx)
  const int T = (int)bv.T[u];
  if (T == 0) return;
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* ESu = ES + s_base * L;
  const double* smu = smax + s_base;
  const int l = tid;
  const bool act = l < L;
  const int lc = act ? l : L - 1;
  const double sh0 = MPF ? 0.0 : mshift[0];
  double* cw = cbuf + wave * 64;
  int err = 0;
  double er[LR > 0 ? LR : 1];
  if (LR > 0) {
    const double* Em = dir ? ET : E;
#pragma unroll
    for (int c = 0; c < LR; c++) er[c] = (c < L) ? Em[(size_t)(c < L ? c : 0) * L + lc] : 0.0;
  }

  // sum_c v[c] * Em[c*L + lc], v exchanged through abuf (caller places the barriers)
  auto matvec = [&](const double* Em) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (LR > 0) {
      // abuf holds zeros from L up to the workgroup's 64 * NW entries >= LR
#pragma unroll
      for (int c = 0; c < LR; c += 4) {
        const double2 a01 = *(const double2*)(abuf + c), a23 = *(const double2*)(abuf + c + 2);
        s0 = fma(a01.x, er[c], s0);
        s1 = fma(a01.y, er[c + 1], s1);
        s2 = fma(a23.x, er[c + 2], s2);
        s3 = fma(a23.y, er[c + 3], s3);
      }
      // rows past the register part (D = 40: the duration step's registers leave room for 128 rows) from memory, eight
      // loads in flight at a time
      if (TAIL) {
        // (the row addresses do not depend on the frame: left alone, the compiler computes all of them ahead of the frame
        // loop and keeps them in registers -- 350 spilled ones; the empty asm makes the base pointer a per-frame value)
        const double* Et = Em + lc;
        asm volatile("" : "+v"(Et));
#pragma unroll 1
        for (int c = LR; c < L; c += 8) {
          double e[8];
#pragma unroll
          for (int i = 0; i < 8; i++) e[i] = (c + i < L) ? Et[(size_t)(c + i < L ? c + i : 0) * L] : 0.0;
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            s0 = fma(abuf[c + i], e[i], s0);
            s1 = fma(abuf[c + i + 1], e[i + 1], s1);
          }
        }
      }
      return (s0 + s1) + (s2 + s3);
    }
    int c = 0;
    for (; c + 4 <= L; c += 4) {
      const double2 a01 = *(const double2*)(abuf + c), a23 = *(const double2*)(abuf + c + 2);
      s0 = fma(a01.x, Em[(size_t)(c + 0) * L + lc], s0);
      s1 = fma(a01.y, Em[(size_t)(c + 1) * L + lc], s1);
      s2 = fma(a23.x, Em[(size_t)(c + 2) * L + lc], s2);
      s3 = fma(a23.y, Em[(size_t)(c + 3) * L + lc], s3);
    }
    for (; c < L; c++) s0 = fma(abuf[c], Em[(size_t)c * L + lc], s0);
    return (s0 + s1) + (s2 + s3);
  };
  // exponent of the workgroup-wide maximum of a non-negative vector
  auto block_exp = [&](double v) {
    const int h = wave_max_hi(act ? v : 0.0);
    if (lane == 0) red[wave] = h;
    __syncthreads();
    int m = red[0];
    for (int w = 1; w < NW; w++) m = max(m, red[w]);
    return hi_exp(m, &err);
  };

  if (dir == 0) {
    // ---------------------------------------------------------------- forward
    double* au = a_g + f_base * L;
    double* pu = p_g + f_base * L;
    double a = ESu[lc];
    double ga = smu[0];
    if (act) au[l] = a;
    if (tid == 0) ga_g[f_base] = ga;
    int rpos = D - 1;
    for (int t = 1; t < T; t++) {
      rpos = (rpos + 1 == D) ? 0 : rpos + 1;  // ring slot of node t-1
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      const uint64_t base = scrf_seg_base(t, D);
      double es[DMAX];
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) {
        const double x = ESu[(base + (d0 < nd ? d0 : 0)) * L + lc];
        es[d0] = (d0 < nd) ? x : 0.0;
      }
      const double smx = smu[base + (lane < nd ? lane : 0)];
      abuf[tid] = act ? a : 0.0;
      __syncthreads();
      const double usum = matvec(MPF ? E + (f_base + t) * LL : E);
      const double sh = MPF ? mshift[f_base + t] : sh0;
      const int k = block_exp(usum);           // (barrier inside: abuf is free again after it)
      const double p = ldexp(usum, -k);
      const double gp = ga + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      ring[(size_t)rpos * L + lc] = p;
      if (tid == 0) { gring[rpos] = gp; gp_g[f_base + t - 1] = gp; }
      if (act) pu[(size_t)(t - 1) * L + l] = p;
      __syncthreads();                          // gring[rpos] visible to every wavefront
      int myslot = rpos - lane;
      if (myslot < 0) myslot += D;
      double x = (lane < np) ? gring[(lane < np) ? myslot : 0] + smx : smx;
      x = (lane < nd) ? x : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      cw[lane] = exp_nonpos(x - G);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) {
        int slot = rpos - d0;
        if (slot < 0) slot += D;
        const double r = (d0 == 0) ? p : ring[(size_t)(d0 < np ? slot : rpos) * L + lc];
        const double pv = (d0 < np) ? r : 1.0;
        const double w = es[d0] * cw[d0];
        if (d0 & 1) acc1 = fma(pv, w, acc1); else acc0 = fma(pv, w, acc0);
      }
      a = acc0 + acc1;
      ga = G;
      if (act) au[(size_t)t * L + l] = a;
      if (tid == 0) ga_g[f_base + t] = ga;
    }
    // Zx = log sum_l exp(alpha[T-1][l])
    const double part = wave_sum_f64(act ? a : 0.0);
    if (lane == 0) redd[wave] = part;
    __syncthreads();
    if (tid == 0) {
      double tot = 0.0;
      for (int w = 0; w < NW; w++) tot += redd[w];
      const double Zx = ga + log(tot);
      if (!(Zx == Zx) || isinf(Zx)) err = 1;
      zx_out[u] = Zx;
    }
  } else {
    // ---------------------------------------------------------------- backward
    double* bu = b_g + f_base * L;
    double* sdu = sd_g + f_base * L;
    int tpos = (T - 1) % D;
    ring[(size_t)tpos * L + lc] = 1.0;  // setTailBeta: beta[T-1] = 0
    if (tid == 0) { gring[tpos] = 0.0; gb_g[f_base + T - 1] = 0.0; gsd_g[f_base + T - 1] = 0.0; }
    if (act) { bu[(size_t)(T - 1) * L + l] = 1.0; sdu[(size_t)(T - 1) * L + l] = 0.0; }
    __syncthreads();
    for (int t = T - 2; t >= 0; t--) {
      const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
      tpos = (tpos == 0) ? D - 1 : tpos - 1;  // ring slot of node t
      double es[DMAX];
      uint64_t r = scrf_seg_base(t + 1, D), myrow = r;
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) {
        const bool ok = d0 < nn;
        const double x = ESu[(ok ? r + d0 : 0) * L + lc];
        es[d0] = ok ? x : 0.0;
        if (lane == d0 && ok) myrow = r + d0;
        r += scrf_node_max_dur(t + 1 + d0, D);
      }
      const double smx = smu[myrow];
      int myslot = tpos + lane + 1;
      if (myslot >= D) myslot -= D;
      const double x = (lane < nn) ? gring[(lane < nn) ? myslot : 0] + smx : -INFINITY;
      const double G = (double)wave_max_f32_dpp((float)x);
      cw[lane] = exp_nonpos(x - G);
      double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) {
        int slot = tpos + d0 + 1;
        if (slot >= D) slot -= D;
        const double bv_ = ring[(size_t)((d0 < nn) ? slot : tpos) * L + lc];
        const double w = es[d0] * cw[d0];
        if (d0 & 1) acc1 = fma((d0 < nn) ? bv_ : 0.0, w, acc1); else acc0 = fma((d0 < nn) ? bv_ : 0.0, w, acc0);
      }
      const double sd = acc0 + acc1;
      abuf[tid] = act ? sd : 0.0;
      __syncthreads();
      const double w = matvec(MPF ? ET + (f_base + t + 1) * LL : ET);
      const double sh = MPF ? mshift[f_base + t + 1] : sh0;
      const int k = block_exp(w);
      const double b = ldexp(w, -k);
      const double gb = G + sh + fma((double)k, LN2_HI, (double)k * LN2_LO);
      ring[(size_t)tpos * L + lc] = b;
      if (tid == 0) { gring[tpos] = gb; gsd_g[f_base + t] = G; gb_g[f_base + t] = gb; }
      if (act) { sdu[(size_t)t * L + l] = sd; bu[(size_t)t * L + l] = b; }
      __syncthreads();
    }
  }
  if (__any(err != 0) && lane == 0) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

int dplin_mw_supported(const ScrfLayout& lay) { return lay.L > 64 && lay.L <= 256 && lay.D <= 40; }
// the one-wavefront-per-utterance form (k_dp_lin): L <= 64, D <= 40 (the log-domain k_dp_wave stops at 32)
int dplin_supported(const ScrfLayout& lay) { return lay.L <= 64 && lay.D <= 40; }

template <int DMAX>
static void launch_dp_lin_mw_t(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                               const double* ES, const double* smax, const double* E, const double* ET,
                               const double* mshift, int m_per_frame, const ScrfDpLin& o, double* zx, int* status) {
  const uint32_t nw = (lay.L + 63) / 64;
  const size_t sm = sizeof(double) * ((((size_t)lay.D * lay.L + 1) & ~(size_t)1) + 64 * nw + 64 + 64 * nw + 16);
#define MW_GO(MPF, LR, TAIL)                                                                                                          \
  do {                                                                                                                                \
    hipFuncSetAttribute((const void*)k_dp_lin_mw<DMAX, MPF, LR, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);          \
    hipLaunchKernelGGL((k_dp_lin_mw<DMAX, MPF, LR, TAIL>), dim3(2 * n_utts), dim3(64 * nw), sm, st, lay, bv, u0, n_utts, ES, smax, E, \
                       ET, mshift, o.a, o.ga, o.p, o.gp, o.b, o.gb, o.sd, o.gsd, zx, status);                                         \
  } while (0)
  // the matrix column in registers (SCRF_DPLIN_EREG=0: from L2 every frame, as before round 4)
  static const bool ereg = !(getenv("SCRF_DPLIN_EREG") && atoi(getenv("SCRF_DPLIN_EREG")) == 0);
  // Measured (ms per launch, from L2 -> in registers): L = 200, D = 25, 512 x 300 frames 17.6 -> 9.4 (LR = 208: 49 spilled
  // registers, still a gain); L = 96, D = 25, 1024 utterances 10.2 -> 6.5; L = 128, D = 10: 5.4 -> 4.3.  At DMAX = 40 the
  // duration step's 80 registers leave room for 128 rows only (L = 200 in full spills 141 registers and LOSES: 13.7 ->
  // 17.8): there 128 rows live in registers and the rest comes from L2 (TAIL).
  static const bool tail_on = !(getenv("SCRF_DPLIN_TAIL") && atoi(getenv("SCRF_DPLIN_TAIL")) == 0);
  if (m_per_frame || !ereg) { if (m_per_frame) MW_GO(1, 0, 0); else MW_GO(0, 0, 0); }
  else if (lay.L <= 128) MW_GO(0, 128, 0);
  else if (DMAX <= 25 && lay.L <= 192) MW_GO(0, 192, 0);
  else if (DMAX <= 25 && lay.L <= 208) MW_GO(0, 208, 0);
  else if (DMAX > 25 && tail_on) MW_GO(0, 128, 1);
  else MW_GO(0, 0, 0);
#undef MW_GO
}

void launch_true_scores(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                        uint64_t n_frames, const double* S, double* s_true) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_true_scores, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, lay, bv, frame_u, u0,
                     n_frames, S, s_true);
}
void launch_exp_rows(hipStream_t st, double* S, uint64_t n_rows, uint32_t L, double* smax) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k_exp_rows, dim3((uint32_t)((n_rows + 15) / 16)), dim3(256), 0, st, S, n_rows, L, smax);
}

static int dp_cu_count() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t pr;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
  }
  return n_cu;
}

void launch_dp_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                   const double* ES, const double* smax, const double* E, const double* ET, const double* mshift,
                   int m_per_frame, const ScrfDpLin& o, double* zx, int* status) {
  if (n_utts == 0) return;
  if (lay.L > 64) {
    if (lay.D <= 10) launch_dp_lin_mw_t<10>(st, lay, bv, u0, n_utts, ES, smax, E, ET, mshift, m_per_frame, o, zx, status);
    else if (lay.D <= 25) launch_dp_lin_mw_t<25>(st, lay, bv, u0, n_utts, ES, smax, E, ET, mshift, m_per_frame, o, zx, status);
    else launch_dp_lin_mw_t<40>(st, lay, bv, u0, n_utts, ES, smax, E, ET, mshift, m_per_frame, o, zx, status);
    return;
  }
  // opt-in (SCRF_DPLIN_MV=1): parity-green, but measured no faster than one wavefront per sweep -- see the kernel's header
  // k_dp_lin_mv (four wavefronts per sweep): SCRF_DPLIN_MV=1 always, =0 never; unset: with per-frame transition matrices
  // when the launch has at most SCRF_DPLIN_MV_SWEEPS (default 3: the kernel's occupancy) sweeps per CU -- there the
  // single-wavefront kernel's step is the latency of one wavefront fetching an L x L matrix, and four fetch it in
  // parallel (config 3, 256 utterances: 1.86 -> 1.03 ms).  With time-invariant transitions it measured no faster at any
  // batch size (64 .. 4096 utterances), so it stays opt-in there.
  static const int mv_mode = getenv("SCRF_DPLIN_MV") ? (atoi(getenv("SCRF_DPLIN_MV")) != 0 ? 1 : 0) : -1;
  static const int mv_sweeps = getenv("SCRF_DPLIN_MV_SWEEPS") ? atoi(getenv("SCRF_DPLIN_MV_SWEEPS")) : 3;
  const bool use_mv = mv_mode == 1 || (mv_mode < 0 && m_per_frame && 2 * (uint64_t)n_utts <= (uint64_t)mv_sweeps * dp_cu_count());
  if (use_mv && lay.D >= 2) {
    // several wavefronts per sweep (k_dp_lin_mv): one workgroup of 4 per (utterance, direction)
    const size_t smv = sizeof(double) * ((((size_t)lay.D * lay.L + 1) & ~(size_t)1) + 4 * DPV_NW * 64);
#define DV_LAUNCH2(DM, MPF)                                                                                            \
  do {                                                                                                                 \
    hipFuncSetAttribute((const void*)k_dp_lin_mv<DM, MPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smv);      \
    hipLaunchKernelGGL((k_dp_lin_mv<DM, MPF>), dim3(2 * n_utts), dim3(64 * DPV_NW), smv, st, lay, bv, u0, n_utts, ES,  \
                       smax, E, ET, mshift, o.a, o.ga, o.p, o.gp, o.b, o.gb, o.sd, o.gsd, zx, status);                 \
  } while (0)
#define DV_LAUNCH(DM) do { if (m_per_frame) DV_LAUNCH2(DM, 1); else DV_LAUNCH2(DM, 0); } while (0)
    if (lay.D <= 10) DV_LAUNCH(10);
    else if (lay.D <= 16) DV_LAUNCH(16);
    else if (lay.D <= 25) DV_LAUNCH(25);
    else if (lay.D <= 32) DV_LAUNCH(32);
    else DV_LAUNCH(40);
#undef DV_LAUNCH
#undef DV_LAUNCH2
    return;
  }
  uint32_t wpb = dp_waves_per_block(sizeof(double) * (m_per_frame ? 0 : (size_t)lay.L * lay.L), sizeof(double) * ((size_t)lay.D * lay.L + 128));
  if (wpb > 8) {
    // Tail-aware workgroup size: one workgroup per CU, so the launch runs in ceil(workgroups / CUs) rounds of T steps,
    // and a step costs a fixed latency plus a share per resident wavefront (measured on MI355X at config 2:
    // 3.4 us + 0.31 us per wavefront).  8192 sweeps on 256 CUs: 12 wavefronts per workgroup are
    // 2.67 -> 3 rounds of 7.1 us steps, 11 are 2.91 -> 3 rounds of 6.8 us steps.
    const int n_cu = dp_cu_count();
    uint32_t best = wpb;
    double best_cost = 1e300;
    for (uint32_t w = wpb; w >= 8; w--) {
      const uint64_t blocks = 2 * (((uint64_t)n_utts + w - 1) / w);
      const double cost = (double)((blocks + n_cu - 1) / n_cu) * (3.4 + 0.31 * w);
      if (cost < best_cost - 1e-9) { best_cost = cost; best = w; }
    }
    wpb = best;
  }
  const uint32_t nblk = 2 * ((n_utts + wpb - 1) / wpb);
  const size_t sm = sizeof(double) * ((m_per_frame ? 0 : (size_t)lay.L * lay.L) + (size_t)wpb * (lay.D * lay.L + 128));
#define DL_LAUNCH2(DM, MPF, LC)                                                                                    \
  do {                                                                                                             \
    hipFuncSetAttribute((const void*)k_dp_lin<DM, MPF, LC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);  \
    hipLaunchKernelGGL((k_dp_lin<DM, MPF, LC>), dim3(nblk), dim3(wpb * 64), sm, st, lay, bv, u0, n_utts, ES, smax, \
                       E, ET, mshift, o.a, o.ga, o.p, o.gp, o.b, o.gb, o.sd, o.gsd, zx, status);                   \
  } while (0)
#define DL_LAUNCH(DM)                                        \
  do {                                                       \
    if (m_per_frame) DL_LAUNCH2(DM, 1, 0);                   \
    else if (DM == 25 && lay.L == 48) DL_LAUNCH2(DM, 0, 48); \
    else DL_LAUNCH2(DM, 0, 0);                               \
  } while (0)
  if (lay.D <= 1) DL_LAUNCH(1);
  else if (lay.D <= 4) DL_LAUNCH(4);
  else if (lay.D <= 10) DL_LAUNCH(10);
  else if (lay.D <= 16) DL_LAUNCH(16);
  else if (lay.D <= 25) DL_LAUNCH(25);
  else if (lay.D <= 32) DL_LAUNCH(32);
  else DL_LAUNCH(40);
#undef DL_LAUNCH
#undef DL_LAUNCH2
}

// ------------------------------------------------------------------------------------------
// k_post_lin: R = Y - gamma over es (in place),
//   gamma[(t,d)][l] = p[t-d][l] * es[(t,d)][l] * b[t][l] * exp(gp[t-d] + smax[(t,d)] + gb[t] - Zx)
// (computeExpF :673-702), plus the per-frame numerator term.  One workgroup per frame.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_post_lin(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u,
                                                  uint32_t u0, const uint32_t* __restrict__ next_lab,
                                                  const double* __restrict__ s_true, const double* __restrict__ M,
                                                  int m_per_frame, double* __restrict__ ES,
                                                  const double* __restrict__ smax, ScrfDpLin o,
                                                  const double* __restrict__ zx, double* __restrict__ numer_f,
                                                  int* __restrict__ status, double* __restrict__ mass_s) {
  __shared__ double fs[64];
  __shared__ double msum[4];
  const uint32_t L = lay.L, D = lay.D;
  const uint64_t fi = blockIdx.x;  // frame index inside the chunk
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t T = bv.T[u];
  const uint64_t row0 = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const uint32_t nd = scrf_node_max_dur(t, D), np = scrf_num_prev(t, D);
  const double Zx = zx[u];
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
  uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD;
  int err = 0;
  if (lab != SCRF_LAB_BAD) {
    if (lab >= L * D) err = SCRF_ERR_BAD_LABEL;
    al = lab % L;
    ld = lab / L + 1;
  }
  const double LN_MAX = 709.782712893384;
  if (threadIdx.x < nd) {
    const uint32_t d0 = threadIdx.x;
    const double x = ((d0 < np) ? o.gp[fi - 1 - d0] : 0.0) + smax[row0 + d0] + o.gb[fi] - Zx;
    if (x >= LN_MAX) err = SCRF_ERR_NUMERIC;
    fs[d0] = exp(x);
  }
  __syncthreads();
  const double* bt = o.b + fi * L;
  double gs = 0.0;
  for (uint32_t idx = threadIdx.x; idx < nd * L; idx += blockDim.x) {
    const uint32_t d0 = idx / L, l = idx - d0 * L;
    const double pv = (d0 < np) ? o.p[(fi - 1 - d0) * L + l] : 1.0;
    const double g = (pv * ES[(row0 + d0) * L + l]) * (bt[l] * fs[d0]);
    const double y = (l == al && d0 + 1 == ld) ? 1.0 : 0.0;
    ES[(row0 + d0) * L + l] = y - g;
    gs += g;
  }
  // state posterior mass of the node (checked against the transition mass by k_mass_check)
  gs = wave_sum_f64_dpp(gs);
  if ((threadIdx.x & 63) == 0) msum[threadIdx.x >> 6] = gs;
  __syncthreads();
  if (threadIdx.x == 0) {
    mass_s[fi] = (msum[0] + msum[1]) + (msum[2] + msum[3]);
    double nodeLi = 0.0;
    if (lab != SCRF_LAB_BAD && err == 0) {
      if (ld <= nd) nodeLi += s_true[fi];
      const uint32_t nl = next_lab[gf];
      if (t + 1 < T && nl != SCRF_LAB_BAD) {
        if (nl >= L * D) err = SCRF_ERR_BAD_LABEL;
        else {
          const double* Mn = M + (m_per_frame ? (fi + 1) * (size_t)L * L : 0);
          nodeLi += Mn[(size_t)al * L + nl % L];
        }
      }
    }
    numer_f[fi] = nodeLi;
  }
  if (err) atomicMax(&status[u], err);
}

void launch_post_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                     uint64_t n_frames, const uint32_t* next_lab, const double* s_true, const double* M,
                     int m_per_frame, double* ES, const double* smax, const ScrfDpLin& o, const double* zx,
                     double* numer_f, int* status, double* mass_s) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_post_lin, dim3((uint32_t)n_frames), dim3(256), 0, st, lay, bv, frame_u, u0, next_lab, s_true, M,
                     m_per_frame, ES, smax, o, zx, numer_f, status, mass_s);
}

// ------------------------------------------------------------------------------------------
// k_mass_check: the reference's posterior-mass self-checks (scrf_mass_ok, scrf_dp_common.h) per frame:
// state mass (summed by the posterior kernel while it formed gamma) against the transition mass
// sum_c exp(alpha[t][c] + beta[t][c] - Zx).  16 lanes per frame.  LIN: alpha/beta are mantissa vectors
// with per-frame log-scales (ScrfDpLin), else plain log-domain arrays.
// ------------------------------------------------------------------------------------------
template <int LIN>
__global__ __launch_bounds__(256) void k_mass_check(ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                                                    uint64_t n_frames, uint32_t L, int frame_model,
                                                    const double* __restrict__ av, const double* __restrict__ ga,
                                                    const double* __restrict__ bvec, const double* __restrict__ gb,
                                                    const double* __restrict__ zx, const double* __restrict__ mass_s,
                                                    int* __restrict__ status) {
  const uint64_t fi = (uint64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const uint32_t sub = threadIdx.x & 15;
  const bool live = fi < n_frames;
  const uint64_t fc = live ? fi : 0;
  const uint64_t gf = bv.frame_off[u0] + fc;
  const uint32_t u = frame_u[gf];
  const double Zx = zx[u];
  double s = 0.0;
  if (LIN) {
    for (uint32_t l = sub; l < L; l += 16) s = fma(av[fc * L + l], bvec[fc * L + l], s);
  } else {
    for (uint32_t l = sub; l < L; l += 16) s += exp(fmin(av[fc * L + l] + bvec[fc * L + l] - Zx, 700.0));
  }
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) s += shfl_f64(s, (int)((threadIdx.x & 63) ^ o));
  if (LIN) s *= exp(fmin(ga[fc] + gb[fc] - Zx, 700.0));
  if (live && sub == 0) {
    const bool last = (gf + 1 == bv.frame_off[u + 1]);
    if (!scrf_mass_ok(mass_s[fc], s, last, frame_model != 0)) atomicMax(&status[u], SCRF_ERR_NUMERIC);
  }
}
void launch_mass_check(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                       uint32_t L, int frame_model, int lin, const double* a, const double* ga, const double* b,
                       const double* gb, const double* zx, const double* mass_s, int* status) {
  if (n_frames == 0) return;
  const dim3 grid((uint32_t)((n_frames + 15) / 16));
  if (lin) hipLaunchKernelGGL(k_mass_check<1>, grid, dim3(256), 0, st, bv, frame_u, u0, n_frames, L, frame_model, a, ga, b, gb, zx, mass_s, status);
  else hipLaunchKernelGGL(k_mass_check<0>, grid, dim3(256), 0, st, bv, frame_u, u0, n_frames, L, frame_model, a, ga, b, gb, zx, mass_s, status);
}

// ------------------------------------------------------------------------------------------
// k_xi_lin: transition posteriors xi[t][c][n] = a[t][c] * exp(M[t+1][c][n]) * B[t][n] with
//   B[t][n] = sd[t][n] * exp(ga[t] + gsd[t] - Zx)   (computeExpF :773-776); B overwrites sd.
// ------------------------------------------------------------------------------------------
__global__ void k_xi_lin(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                         uint64_t n_frames, ScrfDpLin o, const double* __restrict__ zx) {
  const uint32_t L = lay.L;
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * L) return;
  const uint64_t fi = i / L;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const bool last = (gf + 1 == bv.frame_off[u + 1]);
  o.sd[i] = last ? 0.0 : o.sd[i] * exp(o.ga[fi] + o.gsd[fi] - zx[u]);
}
void launch_xi_lin(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                   uint64_t n_frames, const ScrfDpLin& o, const double* zx) {
  if (n_frames == 0) return;
  const uint64_t n = n_frames * lay.L;
  hipLaunchKernelGGL(k_xi_lin, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, bv, frame_u, u0, n_frames, o, zx);
}

// bias-only transitions: the per-frame factor of B alone, gsd[t] <- exp(ga[t] + gsd[t] - Zx) (0 at an
// utterance's last frame); k_atb multiplies it into the sd rows as it loads them (same product, same
// rounding as k_xi_lin's in-place pass, without the read-modify-write of the [T][L] array)
__global__ void k_xi_scale(ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0, uint64_t n_frames,
                           ScrfDpLin o, const double* __restrict__ zx) {
  const uint64_t fi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const bool last = (gf + 1 == bv.frame_off[u + 1]);
  o.gsd[fi] = last ? 0.0 : exp(o.ga[fi] + o.gsd[fi] - zx[u]);
}
void launch_xi_scale(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                     const ScrfDpLin& o, const double* zx) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_xi_scale, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, bv, frame_u, u0, n_frames, o, zx);
}

// parity hook: alpha = log(a) + ga, beta = log(b) + gb
__global__ void k_lin_to_log(uint64_t n_frames, uint32_t L, const double* __restrict__ m, const double* __restrict__ g,
                             double* __restrict__ out) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames * L) return;
  out[i] = log(m[i]) + g[i / L];
}
void launch_lin_to_log(hipStream_t st, uint64_t n_frames, uint32_t L, const double* m, const double* g, double* out) {
  if (n_frames == 0) return;
  const uint64_t n = n_frames * L;
  hipLaunchKernelGGL(k_lin_to_log, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, n_frames, L, m, g, out);
}
