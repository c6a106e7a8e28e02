// scrf_segtrans.hip -- forward / backward / posteriors of the STDSEG_NO_DUR model: the
// transition score of a segment comes from the segment's OWN window, so every window (t, d) that has a
// predecessor carries its own L x L matrix (nodes/CRF_StdSegStateNode_WithoutDurLab.cpp):
//
//   M2[(t,d)][p][l]   = computeTransMatrixValue(window d of node t, p, l)             (:69-110)
//   ad[(t,d)][l]      = LSE_p(alpha[t-d][p] + M2[(t,d)][p][l]) + S[(t,d)][l]  (d <= numPrev) | S   (:132-190)
//   alpha[t][l]       = LSE_d ad[(t,d)][l] ;  Zx = LSE_l alpha[T-1][l]
//   beta[t][c]        = LSE_{d,l}(M2[(t+d,d)][c][l] + beta[t+d][l] + S[(t+d,d)][l]) ; beta[T-1] = 0     (:248-310)
//   gamma[(t,d)][l]   = exp(ad + beta[t][l] - Zx)
//   xi[(t,d)][p][l]   = exp(alpha[t-d][p] + M2 + S + beta[t][l] - Zx)                  (:455-500)
//
// with the gradbuilder of trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp (the PREVIOUS label goes into
// computeExpF).  One workgroup per utterance, log domain, column-wise max-shifted log-sum-exp (scrf_lse.h)
// like the reference's LogMath.  Outputs: R = Y - gamma over ad (in place) and XI2 = Y - xi per window
// ([N_seg][L*L], rows of utterance-initial segments zero) -- the operands of the two expected-count
// contractions -- plus the numerator, Zx and the node's posterior-mass self-checks (:530-545: state and
// transition mass each within [-1e-6, 1 + 1e-6]; no state == transition check in this node).
// The recursion costs L*L*D per frame against L*(L+D) of the TIMIT-demo model: this is the reference's
// "secondary" model type, served for completeness, not tuned.
#include "scrf_dp_common.h"
#include "scrf_kernels.h"
#include "scrf_lse.h"

#include <float.h>
#include <stdlib.h>
#include <math.h>

__global__ void k_zero_initial_rows(ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0, uint64_t n_frames,
                                    uint32_t D, uint32_t LL, double* __restrict__ M2) {
  // the window of length t+1 ending at frame t < D starts the utterance: no predecessor, no transition matrix
  const uint64_t fi = blockIdx.x;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  if (t >= D) return;
  double* row = M2 + ((bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D) + t) * (uint64_t)LL;
  for (uint32_t i = threadIdx.x; i < LL; i += blockDim.x) row[i] = 0.0;
}
void launch_zero_initial_rows(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                              uint32_t D, uint32_t L, double* M2) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_zero_initial_rows, dim3((uint32_t)n_frames), dim3(256), 0, st, bv, frame_u, u0, n_frames, D, L * L, M2);
}

__global__ void k_fb_segtrans(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const uint32_t* __restrict__ prev_lab,
                              const double* __restrict__ S, const double* __restrict__ M2, double* __restrict__ AD,
                              double* __restrict__ alpha_g, double* __restrict__ beta_g, double* __restrict__ XI2,
                              double* __restrict__ numer_out, double* __restrict__ zx_out, int* __restrict__ status,
                              int write_post) {
  extern __shared__ double smem[];
  __shared__ double zx_s;
  __shared__ double mass2[2];
  const int L = lay.L, D = lay.D;
  const int NT = blockDim.x, tid = threadIdx.x;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const size_t LL = (size_t)L * L;
  const double* Su = S + s_base * L;
  const double* Mu = M2 + s_base * LL;
  double* ADu = AD + s_base * L;
  double* alu = alpha_g + f_base * L;
  double* beu = beta_g + f_base * L;
  const uint32_t* labs = bv.labels ? bv.labels + bv.frame_off[u] : nullptr;
  const uint32_t* plabs = prev_lab ? prev_lab + bv.frame_off[u] : nullptr;

  FbLse c;
  c.L = L;
  c.G = NT / L;
  if (c.G < 1) c.G = 1;
  c.g = tid / L;
  c.j = tid - c.g * L;
  c.active = c.g < c.G && tid < c.G * L;
  double* aring = smem;                        // [D][L] alpha of the last D nodes
  double* bring = aring + (size_t)D * L;       // [D][L] beta of the next D nodes
  double* tb = bring + (size_t)D * L;          // [D][L] beta[t+d] + S[(t+d,d)]
  c.red_m = tb + (size_t)D * L;                // [G][L]
  c.red_s = c.red_m + (size_t)c.G * L;         // [G][L]
  int err = 0;
  if (T == 0) {
    if (tid == 0) { status[u] = SCRF_ERR_EMPTY; numer_out[u] = 0.0; zx_out[u] = 0.0; }
    return;
  }

  // ---- forward -----------------------------------------------------------------------------
  for (int l = tid; l < L; l += NT) {   // computeFirstAlpha :200-208
    const double a = Su[l];
    ADu[l] = a;
    alu[l] = a;
    aring[l] = a;
  }
  __syncthreads();
  for (int t = 1; t < T; t++) {
    const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
    const uint64_t base = scrf_seg_base(t, D);
    double run_m = -INFINITY, run_s = 0.0, a_new = 0.0;   // group 0, thread j = l: log-sum over the durations
    for (int d = 1; d <= nd; d++) {
      double v;
      if (d <= np) {
        const double* pa = aring + (size_t)((t - d) % D) * L;
        const double* Mrow = Mu + (base + d - 1) * LL;
        const double r = col_lse(c, L, [&](int p, int l, bool) { return pa[p] + Mrow[(size_t)p * L + l]; }, &err);
        v = r + ((c.active && c.g == 0) ? Su[(base + d - 1) * L + c.j] : 0.0);
      } else {
        v = (c.active && c.g == 0) ? Su[(base + d - 1) * L + c.j] : 0.0;
      }
      if (c.active && c.g == 0) {
        ADu[(base + d - 1) * L + c.j] = v;
        if (v > run_m) { run_s = run_s * exp(run_m - v) + 1.0; run_m = v; }   // (exp(-inf) = 0 on the first value)
        else run_s += exp(v - run_m);
      }
    }
    if (c.active && c.g == 0) {
      if (!(run_s > 0.0) || isinf(run_s) || isnan(run_s)) err = SCRF_ERR_NUMERIC;
      a_new = run_m + log(run_s);
      alu[(size_t)t * L + c.j] = a_new;
    }
    __syncthreads();                       // every read of the slot that node t overwrites (node t - D) is done
    if (c.active && c.g == 0) aring[(size_t)(t % D) * L + c.j] = a_new;
    __syncthreads();
  }
  if (tid == 0) {  // computeAlphaSum: logAdd(alphaArray, L) in index order
    const double* al = aring + (size_t)((T - 1) % D) * L;
    double mx = al[0];
    for (int l = 1; l < L; l++) if (al[l] > mx) mx = al[l];
    double sum = 0.0;
    for (int l = 0; l < L; l++) sum += exp(al[l] - mx);
    zx_s = mx + log(sum);
  }
  __threadfence();   // alpha of every node is read back from memory by the posterior pass below
  __syncthreads();
  const double Zx = zx_s;
  if (isnan(Zx) || isinf(Zx)) err = SCRF_ERR_NUMERIC;

  // ---- backward + posteriors -----------------------------------------------------------------
  const double LN_MAX = 709.782712893384;  // log(DBL_MAX): expE overflow guard (CRF_LogMath.cpp:213)
  double numer = 0.0;
  for (int t = T - 1; t >= 0; t--) {
    const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
    double* bt = bring + (size_t)(t % D) * L;
    if (nn == 0) {
      for (int l = tid; l < L; l += NT) bt[l] = 0.0;   // setTailBeta
      __syncthreads();
    } else {
      for (int idx = tid; idx < nn * L; idx += NT) {   // tempBeta :262-271
        const int di = idx / L, l = idx - di * L;
        tb[idx] = bring[(size_t)((t + di + 1) % D) * L + l] + Su[(scrf_seg_base(t + di + 1, D) + di) * L + l];
      }
      __syncthreads();
      const double r = col_lse(c, nn * L,
                               [&](int i, int cl, bool) {
                                 const int di = i / L, l = i - di * L;
                                 return Mu[(scrf_seg_base(t + di + 1, D) + di) * LL + (size_t)cl * L + l] + tb[i];
                               },
                               &err);
      // bt is the ring slot of node t + D, whose last reader was the tempBeta fill above
      if (c.active && c.g == 0) bt[c.j] = r;
      __syncthreads();
    }
    for (int l = tid; l < L; l += NT) beu[(size_t)t * L + l] = bt[l];
    if (write_post) {
      // true labels of this node and of the nearest earlier labelled node (computeExpF :430-450)
      const uint32_t lab = labs ? labs[t] : SCRF_LAB_BAD;
      const uint32_t pl = plabs ? plabs[t] : SCRF_LAB_BAD;
      uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD, apl = SCRF_LAB_BAD;
      if (lab != SCRF_LAB_BAD) {
        if (lab >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
        al = lab % L;
        ld = lab / L + 1;
      }
      if (pl != SCRF_LAB_BAD) {
        if (pl >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
        apl = pl % L;
      }
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      const uint64_t base = scrf_seg_base(t, D);
      double gs = 0.0, xs = 0.0;
      // transition posteriors first: they read ad-free quantities; then gamma overwrites ad in place
      for (int d = 1; d <= nd; d++) {
        double* Xrow = XI2 + (s_base + base + d - 1) * LL;
        if (d <= np) {
          const double* pa = alu + (size_t)(t - d) * L;
          const double* Mrow = Mu + (base + d - 1) * LL;
          for (int idx = tid; idx < L * L; idx += NT) {
            const int p = idx / L, l = idx - p * L;
            const double a = pa[p] + Mrow[idx] + Su[(base + d - 1) * L + l] + bt[l] - Zx;
            if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
            const double x = exp(a);
            const double y = ((uint32_t)l == al && (uint32_t)d == ld && (uint32_t)p == apl) ? 1.0 : 0.0;
            Xrow[idx] = y - x;
            xs += x;
          }
        } else {
          for (int idx = tid; idx < L * L; idx += NT) Xrow[idx] = 0.0;
        }
      }
      for (int idx = tid; idx < nd * L; idx += NT) {
        const int di = idx / L, l = idx - di * L;
        const double a = ADu[(base + di) * L + l] + bt[l] - Zx;
        if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
        const double g = exp(a);
        const double y = ((uint32_t)l == al && (uint32_t)(di + 1) == ld) ? 1.0 : 0.0;
        ADu[(base + di) * L + l] = y - g;
        gs += g;
      }
      if (tid == 0) { mass2[0] = 0.0; mass2[1] = 0.0; }
      __syncthreads();
      for (int o = 32; o >= 1; o >>= 1) { gs += __shfl_xor(gs, o); xs += __shfl_xor(xs, o); }
      if ((tid & 63) == 0) { atomicAdd(&mass2[0], gs); atomicAdd(&mass2[1], xs); }
      __syncthreads();
      if (tid == 0) {
        const double sm = mass2[0], tm = np == 0 ? 1.0 : mass2[1];
        if (!(sm <= 1.000001) || !(sm >= -0.000001) || !(tm <= 1.000001) || !(tm >= -0.000001)) err = SCRF_ERR_NUMERIC;
        if (lab != SCRF_LAB_BAD && err == 0 && ld <= (uint32_t)nd) {
          double nodeLi = Su[(base + ld - 1) * L + al];
          if ((int)ld <= np && apl != SCRF_LAB_BAD) nodeLi += Mu[(base + ld - 1) * LL + (size_t)apl * L + al];
          numer += nodeLi;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    numer_out[u] = numer;
    zx_out[u] = Zx;
  }
  if (err) atomicMax(&status[u], err);
}

// ------------------------------------------------------------------------------------------
// k_fb_segtrans_w: the same recursion with the D windows of a node spread over the workgroup's wavefronts.  Forward:
// wavefront w takes the windows d = w+1, w+17, ... of node t, lanes over the label l, and walks the previous node's
// labels p once with a running (max, sum) per lane (M2 rows are read as whole 8*L-byte lines, 16 in flight); the
// D window values of a label are then folded in duration order.  Backward: wavefront w takes the previous labels
// c = w, w+16, ...; per label every lane folds its label's terms of all next windows (maximum first, then the shifted
// sum), and the lanes are combined once with DPP max / sum.  Posteriors: previous label over the wavefronts, label over
// the lanes.  In the forward walk the log-sum-exp shift is updated on the way (not found in a first pass as in col_lse
// / the reference's logAdd): same value up to the rounding of the shift.  Two workgroup barriers per node and
// direction instead of two per window.  LDS: 2*D*L doubles.
// ------------------------------------------------------------------------------------------
#define FBW_WAVES 16
__global__ __launch_bounds__(64 * FBW_WAVES) void k_fb_segtrans_w(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const uint32_t* __restrict__ prev_lab,
                              const double* __restrict__ S, const double* __restrict__ M2, double* __restrict__ AD,
                              double* __restrict__ alpha_g, double* __restrict__ beta_g, double* __restrict__ XI2,
                              double* __restrict__ numer_out, double* __restrict__ zx_out, int* __restrict__ status,
                              int write_post) {
  extern __shared__ double smem[];
  __shared__ double zx_s;
  __shared__ double mass2[2];
  const int L = lay.L, D = lay.D;
  const int NT = blockDim.x, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, n_waves = NT >> 6;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const size_t LL = (size_t)L * L;
  const double* Su = S + s_base * L;
  const double* Mu = M2 + s_base * LL;
  double* ADu = AD + s_base * L;
  double* alu = alpha_g + f_base * L;
  double* beu = beta_g + f_base * L;
  const uint32_t* labs = bv.labels ? bv.labels + bv.frame_off[u] : nullptr;
  const uint32_t* plabs = prev_lab ? prev_lab + bv.frame_off[u] : nullptr;
  // forward: aring [D][L] alpha of the last D nodes, vd [D][L] window values of the node in hand
  // backward (same memory): bring [D][L] beta of the next D nodes, tb [D][L] beta + S of the next windows
  double* aring = smem;
  double* vd = smem + (size_t)D * L;
  double* bring = smem;
  double* tb = smem + (size_t)D * L;
  int err = 0;
  if (T == 0) {
    if (tid == 0) { status[u] = SCRF_ERR_EMPTY; numer_out[u] = 0.0; zx_out[u] = 0.0; }
    return;
  }

  // ---- forward -----------------------------------------------------------------------------
  for (int l = tid; l < L; l += NT) {   // computeFirstAlpha :200-208
    const double a = Su[l];
    ADu[l] = a;
    alu[l] = a;
    aring[l] = a;
  }
  __syncthreads();
  for (int t = 1; t < T; t++) {
    const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
    const uint64_t base = scrf_seg_base(t, D);
    for (int d = 1 + wave; d <= nd; d += n_waves) {
      for (int l = lane; l < L; l += 64) {
        double v = Su[(base + d - 1) * L + l];
        if (d <= np) {
          const double* pa = aring + (size_t)((t - d) % D) * L;
          const double* Mrow = Mu + (base + d - 1) * LL + l;
          double m = -INFINITY, sum = 0.0;
          int p = 0;
          for (; p + 16 <= L; p += 16) {
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] = Mrow[(size_t)(p + i) * L];
#pragma unroll
            for (int i = 0; i < 16; i++) x[i] += pa[p + i];
            double bm = x[0];
#pragma unroll
            for (int i = 1; i < 16; i++) bm = fmax(bm, x[i]);
            const double nm = fmax(m, bm);
            double acc = sum * exp_nonpos(m - nm);
#pragma unroll
            for (int i = 0; i < 16; i++) acc += exp_nonpos(x[i] - nm);
            sum = acc; m = nm;
          }
          for (; p < L; p++) {
            const double x = pa[p] + Mrow[(size_t)p * L];
            const double nm = fmax(m, x);
            sum = sum * exp_nonpos(m - nm) + exp_nonpos(x - nm);
            m = nm;
          }
          if (!(sum > 0.0) || isinf(sum) || isnan(sum)) err = SCRF_ERR_NUMERIC;
          v = (m + log(sum)) + v;
        }
        ADu[(base + d - 1) * L + l] = v;
        vd[(size_t)(d - 1) * L + l] = v;
      }
    }
    __syncthreads();
    for (int l = tid; l < L; l += NT) {   // log-sum over the durations, in order (:176-190)
      double run_m = -INFINITY, run_s = 0.0;
      for (int d = 0; d < nd; d++) {
        const double v = vd[(size_t)d * L + l];
        if (v > run_m) { run_s = run_s * exp(run_m - v) + 1.0; run_m = v; }
        else run_s += exp(v - run_m);
      }
      if (!(run_s > 0.0) || isinf(run_s) || isnan(run_s)) err = SCRF_ERR_NUMERIC;
      const double a_new = run_m + log(run_s);
      alu[(size_t)t * L + l] = a_new;
      aring[(size_t)(t % D) * L + l] = a_new;   // the slot of node t - D, whose last readers were before the barrier
    }
    __syncthreads();
  }
  if (tid == 0) {  // computeAlphaSum: logAdd(alphaArray, L) in index order
    const double* al = aring + (size_t)((T - 1) % D) * L;
    double mx = al[0];
    for (int l = 1; l < L; l++) if (al[l] > mx) mx = al[l];
    double sum = 0.0;
    for (int l = 0; l < L; l++) sum += exp(al[l] - mx);
    zx_s = mx + log(sum);
  }
  __threadfence();   // alpha of every node is read back from memory by the posterior pass below
  __syncthreads();
  const double Zx = zx_s;
  if (isnan(Zx) || isinf(Zx)) err = SCRF_ERR_NUMERIC;

  // ---- backward + posteriors -----------------------------------------------------------------
  const double LN_MAX = 709.782712893384;  // log(DBL_MAX): expE overflow guard (CRF_LogMath.cpp:213)
  double numer = 0.0;
  for (int t = T - 1; t >= 0; t--) {
    const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
    double* bt = bring + (size_t)(t % D) * L;
    if (nn == 0) {
      for (int l = tid; l < L; l += NT) bt[l] = 0.0;   // setTailBeta
      __syncthreads();
    } else {
      for (int idx = tid; idx < nn * L; idx += NT) {   // tempBeta :262-271
        const int di = idx / L, l = idx - di * L;
        tb[idx] = bring[(size_t)((t + di + 1) % D) * L + l] + Su[(scrf_seg_base(t + di + 1, D) + di) * L + l];
      }
      __syncthreads();
      for (int c = wave; c < L; c += n_waves) {   // previous label c: every lane folds its label's terms of all next windows
        double wm = -INFINITY;
        for (int l = lane; l < L; l += 64)
          for (int di = 0; di < nn; di++)
            wm = fmax(wm, Mu[(scrf_seg_base(t + di + 1, D) + di) * LL + (size_t)c * L + l] + tb[(size_t)di * L + l]);
        wm = wave_max_f64_dpp(wm);
        double part = 0.0;
        for (int l = lane; l < L; l += 64)
          for (int di = 0; di < nn; di++)
            part += exp_nonpos((Mu[(scrf_seg_base(t + di + 1, D) + di) * LL + (size_t)c * L + l] + tb[(size_t)di * L + l]) - wm);
        part = wave_sum_f64_dpp(part);
        if (!(part > 0.0) || isinf(part) || isnan(part)) err = SCRF_ERR_NUMERIC;
        if (lane == 0) bt[c] = wm + log(part);   // the ring slot of node t + D, whose last reader was the tempBeta fill above
      }
      __syncthreads();
    }
    for (int l = tid; l < L; l += NT) beu[(size_t)t * L + l] = bt[l];
    if (write_post) {
      // true labels of this node and of the nearest earlier labelled node (computeExpF :430-450)
      const uint32_t lab = labs ? labs[t] : SCRF_LAB_BAD;
      const uint32_t pl = plabs ? plabs[t] : SCRF_LAB_BAD;
      uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD, apl = SCRF_LAB_BAD;
      if (lab != SCRF_LAB_BAD) {
        if (lab >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
        al = lab % L;
        ld = lab / L + 1;
      }
      if (pl != SCRF_LAB_BAD) {
        if (pl >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
        apl = pl % L;
      }
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      const uint64_t base = scrf_seg_base(t, D);
      double gs = 0.0, xs = 0.0;
      // transition posteriors first: they read ad-free quantities; then gamma overwrites ad in place
      for (int d = 1; d <= nd; d++) {
        double* Xrow = XI2 + (s_base + base + d - 1) * LL;
        if (d <= np) {
          const double* pa = alu + (size_t)(t - d) * L;
          const double* Mrow = Mu + (base + d - 1) * LL;
          for (int p = wave; p < L; p += n_waves) {
            const double pap = pa[p];
            for (int l = lane; l < L; l += 64) {
              const int idx = p * L + l;
              const double a = pap + Mrow[idx] + Su[(base + d - 1) * L + l] + bt[l] - Zx;
              if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
              const double x = a <= 0.0 ? exp_nonpos(a) : exp(a);
              const double y = ((uint32_t)l == al && (uint32_t)d == ld && (uint32_t)p == apl) ? 1.0 : 0.0;
              __builtin_nontemporal_store(y - x, &Xrow[idx]);   // 18 KB per window, read once by the contraction
              xs += x;
            }
          }
        } else {
          for (int idx = tid; idx < L * L; idx += NT) __builtin_nontemporal_store(0.0, &Xrow[idx]);
        }
      }
      for (int idx = tid; idx < nd * L; idx += NT) {
        const int di = idx / L, l = idx - di * L;
        const double a = ADu[(base + di) * L + l] + bt[l] - Zx;
        if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
        const double g = exp(a);
        const double y = ((uint32_t)l == al && (uint32_t)(di + 1) == ld) ? 1.0 : 0.0;
        ADu[(base + di) * L + l] = y - g;
        gs += g;
      }
      if (tid == 0) { mass2[0] = 0.0; mass2[1] = 0.0; }
      __syncthreads();
      for (int o = 32; o >= 1; o >>= 1) { gs += __shfl_xor(gs, o); xs += __shfl_xor(xs, o); }
      if ((tid & 63) == 0) { atomicAdd(&mass2[0], gs); atomicAdd(&mass2[1], xs); }
      __syncthreads();
      if (tid == 0) {
        const double sm = mass2[0], tm = np == 0 ? 1.0 : mass2[1];
        if (!(sm <= 1.000001) || !(sm >= -0.000001) || !(tm <= 1.000001) || !(tm >= -0.000001)) err = SCRF_ERR_NUMERIC;
        if (lab != SCRF_LAB_BAD && err == 0 && ld <= (uint32_t)nd) {
          double nodeLi = Su[(base + ld - 1) * L + al];
          if ((int)ld <= np && apl != SCRF_LAB_BAD) nodeLi += Mu[(base + ld - 1) * LL + (size_t)apl * L + al];
          numer += nodeLi;
        }
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    numer_out[u] = numer;
    zx_out[u] = Zx;
  }
  if (err) atomicMax(&status[u], err);
}

size_t fb_segtrans_smem_bytes(const ScrfLayout& lay, int NT) {
  int G = NT / (int)lay.L;
  if (G < 1) G = 1;
  return sizeof(double) * ((size_t)3 * lay.D * lay.L + (size_t)2 * G * lay.L);
}

void launch_fb_segtrans(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                        const uint32_t* prev_lab, const double* S, const double* M2, double* AD, double* alpha_g,
                        double* beta_g, double* XI2, double* numer, double* zx, int* status, int write_post) {
  if (n_utts == 0) return;
  const size_t smw = sizeof(double) * 2 * (size_t)lay.D * lay.L;
  if (smw <= 150 * 1024) {   // the wavefront-per-window form; larger D * L keeps the column-wise kernel
    hipFuncSetAttribute((const void*)k_fb_segtrans_w, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw);
    // wavefronts per workgroup: 16 when every CU has at most one utterance, 8 (two workgroups per CU) beyond that
    int nw = n_utts > 256 ? FBW_WAVES / 2 : FBW_WAVES;
    if (const char* e = getenv("SCRF_FBW_WAVES")) { const int v = atoi(e); if (v >= 1 && v <= FBW_WAVES) nw = v; }
    hipLaunchKernelGGL(k_fb_segtrans_w, dim3(n_utts), dim3(64 * nw), smw, st, lay, bv, u0, prev_lab, S, M2, AD, alpha_g, beta_g,
                       XI2, numer, zx, status, write_post);
    return;
  }
  const int NT = fb_block_threads(lay);
  const size_t sm = fb_segtrans_smem_bytes(lay, NT);
  hipFuncSetAttribute((const void*)k_fb_segtrans, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(k_fb_segtrans, dim3(n_utts), dim3(NT), sm, st, lay, bv, u0, prev_lab, S, M2, AD, alpha_g, beta_g, XI2,
                     numer, zx, status, write_post);
}

// ------------------------------------------------------------------------------------------
// Lattice of the STDSEG_NO_DUR model (decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h): state 0 =
// start, node t owns the L states 1 + t*L + l; arcs in AddArc order -- node by node, per label, per
// duration with a predecessor every previous label p (state(t-dur, p) -> state(t, l), labels
// l + L*(dur-1) + 1, weight float(-1 * (M2 + S)) = float(-getFullTransValue)), then the utterance-initial
// duration from the start state (float(-S)); last the L epsilon arcs of weight final_w into the final state.
// One workgroup per node; the arc index inside the node is closed form.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t segtrans_arc_base(uint32_t t, uint32_t L, uint32_t D) {
  // arcs emitted before node t: node tau < D has tau predecessors and one initial duration
  if (t <= D) return (uint64_t)L * ((uint64_t)L * t * (t - (t ? 1 : 0)) / 2 + t);
  return (uint64_t)L * ((uint64_t)L * D * (D - 1) / 2 + D) + (uint64_t)(t - D) * L * D * L;
}
uint64_t segtrans_num_arcs(uint32_t T, uint32_t L, uint32_t D) { return T ? segtrans_arc_base(T, L, D) + L : 0; }

__global__ void k_arcs_segtrans(ScrfLayout lay, uint32_t T, const double* __restrict__ S, const double* __restrict__ M2,
                                float final_w, scrf_arc* __restrict__ arcs) {
  const uint32_t L = lay.L, D = lay.D, t = blockIdx.x;
  const size_t LL = (size_t)L * L;
  if (t == T) {   // final arcs
    for (uint32_t l = threadIdx.x; l < L; l += blockDim.x)
      arcs[segtrans_arc_base(T, L, D) + l] = scrf_arc{(int32_t)(1 + (T - 1) * L + l), 0, 0, final_w, (int32_t)(1 + T * L)};
    return;
  }
  const uint32_t np = scrf_num_prev(t, D), nd = scrf_node_max_dur(t, D);
  const uint64_t base = scrf_seg_base(t, D), a0 = segtrans_arc_base(t, L, D);
  const uint32_t per_lab = np * L + (nd - np);
  for (uint32_t i = threadIdx.x; i < L * per_lab; i += blockDim.x) {
    const uint32_t l = i / per_lab, k = i - l * per_lab;
    scrf_arc a;
    a.dst = (int32_t)(1 + t * L + l);
    if (k < np * L) {
      const uint32_t dur = k / L + 1, p = k - (dur - 1) * L;
      a.src = (int32_t)(1 + (t - dur) * L + p);
      a.ilabel = a.olabel = (int32_t)(l + L * (dur - 1) + 1);
      a.w = (float)(-1 * (M2[(base + dur - 1) * LL + (size_t)p * L + l] + S[(base + dur - 1) * L + l]));
    } else {
      const uint32_t dur = np + 1 + (k - np * L);
      a.src = 0;
      a.ilabel = a.olabel = (int32_t)(l + L * (dur - 1) + 1);
      a.w = (float)(-1 * S[(base + dur - 1) * L + l]);
    }
    arcs[a0 + i] = a;
  }
}
void launch_arcs_segtrans(hipStream_t st, const ScrfLayout& lay, uint32_t T, const double* S, const double* M2, float final_w,
                          scrf_arc* arcs) {
  if (T == 0) return;
  hipLaunchKernelGGL(k_arcs_segtrans, dim3(T + 1), dim3(256), 0, st, lay, T, S, M2, final_w, arcs);
}

// ------------------------------------------------------------------------------------------
// Best path over that lattice without materialising it: ShortestPath's relaxation in state order with
// strict improvement (first relaxed wins): state(t, l) is reached from the start state first (the
// utterance-initial duration), then from state(t-dur, p) with dur DEscending (lower state ids first) and
// p ascending; the final state from state(T-1, l), l ascending.  Path cost = left-to-right float sum.
// One workgroup per utterance, thread = label; costs of the last D nodes in an LDS ring.
// ------------------------------------------------------------------------------------------
__global__ void k_viterbi_segtrans(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ S,
                                   const double* __restrict__ M2, uint16_t* __restrict__ bp_p, uint16_t* __restrict__ bp_d,
                                   uint32_t* __restrict__ out_labels, uint32_t* __restrict__ out_n,
                                   float* __restrict__ out_cost) {
  extern __shared__ float vring[];   // [D][L]
  const int L = lay.L, D = lay.D;
  const int tid = threadIdx.x, NT = blockDim.x;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const size_t LL = (size_t)L * L;
  const double* Su = S + s_base * L;
  const double* Mu = M2 + s_base * LL;
  uint16_t* bpp = bp_p + f_base * L;
  uint16_t* bpd = bp_d + f_base * L;
  uint32_t* outl = out_labels + bv.frame_off[u];
  for (int t = 0; t < T; t++) {
    const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
    const uint64_t base = scrf_seg_base(t, D);
    for (int l = tid; l < L; l += NT) {
      float best = INFINITY;
      int bd = 0, bpv = 0xffff;
      if (nd > np) {   // from the start state: distance 0 + float(-S)
        best = 0.0f + (float)(-1 * Su[(base + nd - 1) * L + l]);
        bd = nd;
      }
      for (int dur = np; dur >= 1; dur--) {
        const float* cp = vring + (size_t)((t - dur) % D) * L;
        const double* Mrow = Mu + (base + dur - 1) * LL;
        const double sv = Su[(base + dur - 1) * L + l];
        for (int p = 0; p < L; p++) {
          const float c = cp[p] + (float)(-1 * (Mrow[(size_t)p * L + l] + sv));
          if (c < best) { best = c; bd = dur; bpv = p; }
        }
      }
      bpd[(size_t)t * L + l] = (uint16_t)bd;
      bpp[(size_t)t * L + l] = (uint16_t)bpv;
      // the ring slot of node t is node t - D's, which this step still reads (dur = D): the new costs wait
      // in a staging row until every thread has finished reading
      vring[(size_t)D * L + l] = best;
    }
    __syncthreads();
    for (int l = tid; l < L; l += NT) vring[(size_t)(t % D) * L + l] = vring[(size_t)D * L + l];
    __syncthreads();
  }
  if (tid == 0) {
    float best = INFINITY;
    int bl = -1;
    const float* cl = vring + (size_t)((T - 1) % D) * L;
    for (int l = 0; l < L; l++) {
      const float cst = cl[l] + -0.0f;
      if (cst < best) { best = cst; bl = l; }
    }
    uint32_t n = 0;
    if (bl >= 0) {
      int t = T - 1, l = bl;
      while (true) {
        const int d = bpd[(size_t)t * L + l];
        outl[n++] = (uint32_t)(l + L * (d - 1));
        const int p = bpp[(size_t)t * L + l];
        if (p == 0xffff) break;
        t -= d;
        l = p;
      }
      for (uint32_t i = 0; i < n / 2; i++) {
        const uint32_t tmp = outl[i];
        outl[i] = outl[n - 1 - i];
        outl[n - 1 - i] = tmp;
      }
      best = best + 0.0f;  // Times(distance, Final = One)
    }
    out_n[u] = n;
    out_cost[u] = best;
  }
}
void launch_viterbi_segtrans(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                             const double* S, const double* M2, uint16_t* bp_p, uint16_t* bp_d, uint32_t* out_labels,
                             uint32_t* out_n, float* out_cost) {
  if (n_utts == 0) return;
  int NT = ((int)lay.L + 63) / 64 * 64;
  if (NT > 1024) NT = 1024;
  const size_t sm = sizeof(float) * (size_t)(lay.D + 1) * lay.L;
  hipFuncSetAttribute((const void*)k_viterbi_segtrans, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(k_viterbi_segtrans, dim3(n_utts), dim3(NT), sm, st, lay, bv, u0, S, M2, bp_p, bp_d, out_labels, out_n,
                     out_cost);
}
