// scrf_nstate.hip -- the n-state frame model (crf_states = K > 1; nodes/CRF_StdNStateNode.cpp): label c is state c % K
// of phone c / K (P = nLabs / K phones).  Allowed transitions: self (c -> c), next state inside a phone (c-1 -> c), end
// state of any phone -> start state of any phone; the weight layout holds exactly those
// (ftrmaps/CRF_StdFeatureMap.cpp:280-407, ScrfLayout::state_idx_k / trans_idx_k).
//
// Per frame: S[c] state values, TD[c] self transitions, TO[c] = transition c -> c+1 (unused for end states),
// TE[p*P + q] = end state of phone p -> start state of phone q (diagTransMatrix / offDiagTransMatrix /
// denseTransMatrix of the node).  Like scrf_stdseg.hip this model type is outside the benchmarked path: plain
// log-domain kernels in the reference's operation order, one workgroup per utterance.
#include "scrf_kernels.h"

#include <math.h>

#include <vector>

__device__ __forceinline__ double ns_exp(double x, int* err) {
  if (x >= 709.782712893384) *err = SCRF_ERR_NUMERIC;
  return exp(x);
}
__device__ __forceinline__ double ns_log(double x, int* err) {
  if (!(x > 0.0) || isinf(x)) *err = SCRF_ERR_NUMERIC;
  return log(x);
}
// CRF_LogMath::logAdd(double, double) (utils/CRF_LogMath.cpp:41-64)
__device__ __forceinline__ double ns_logadd2(double a, double b, int* err) {
  double x = a, y = b;
  if (y > x) { y = a; x = b; }
  return x + ns_log(1.0 + ns_exp(y - x, err), err);
}
__device__ __forceinline__ double ns_dot(const ScrfLayout& lay, const float* x, const double* lambda, uint32_t lc, bool state) {
  double v = 0.0;
  if (state) {
    if (lay.use_sf) for (uint32_t f = lay.sfs; f <= lay.sfe; f++) v = __dadd_rn(v, __dmul_rn((double)x[f], lambda[lc++]));
    if (lay.use_sb) v = __dadd_rn(v, __dmul_rn(lambda[lc], lay.sbv));
  } else {
    if (lay.use_tf) for (uint32_t f = lay.tfs; f <= lay.tfe; f++) v = __dadd_rn(v, __dmul_rn((double)x[f], lambda[lc++]));
    if (lay.use_tb) v = __dadd_rn(v, __dmul_rn(lambda[lc], lay.tbv));
  }
  return v;
}

// computeTransMatrix :68-90; one thread per (frame, label)
__global__ __launch_bounds__(256) void k_ns_scores(ScrfLayout lay, const float* __restrict__ X, uint64_t n_frames,
                                                   const double* __restrict__ lambda, double* __restrict__ S,
                                                   double* __restrict__ TD, double* __restrict__ TO, double* __restrict__ TE) {
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_frames * L) return;
  const uint64_t fr = e / L;
  const uint32_t c = (uint32_t)(e % L);
  const float* x = X + fr * lay.F;
  S[fr * L + c] = ns_dot(lay, x, lambda, lay.state_idx_k(c), true);
  TD[fr * L + c] = ns_dot(lay, x, lambda, lay.trans_idx_k(c, c), false);
  if (c % K == 0) {
    for (uint32_t p = 0; p < P; p++) TE[fr * P * P + p * P + c / K] = ns_dot(lay, x, lambda, lay.trans_idx_k(p * K + K - 1, c), false);
  } else {
    TO[fr * L + c - 1] = ns_dot(lay, x, lambda, lay.trans_idx_k(c - 1, c), false);
  }
  if ((c + 1) % K == 0) TO[fr * L + c] = 0.0;
}

// computeFirstAlpha / computeAlpha :100-157, computeAlphaSum, setTailBeta, computeBeta :170-207
#define NSFB_G 16   // lanes that share one end -> start log-sum-exp
__device__ __forceinline__ double ns_group_max(double v) {
#pragma unroll
  for (int o = NSFB_G / 2; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double ns_group_sum(double v) {
#pragma unroll
  for (int o = NSFB_G / 2; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// One workgroup per utterance.  Work items of a frame: one per label for the self / next-state terms, and a group of
// NSFB_G lanes per phone for the max-shifted log-sum-exp over the P end states (forward, into the phone's start
// state) or the P start states (backward, out of its end state); the group's lanes split the P terms, the max and the
// sum are combined with xor shuffles, so the sum runs in a tree order instead of the node's index order.
__global__ __launch_bounds__(1024) void k_ns_fb(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ S,
                                                const double* __restrict__ TD, const double* __restrict__ TO,
                                                const double* __restrict__ TE, double* __restrict__ alpha,
                                                double* __restrict__ beta, double* __restrict__ zx_out, int* __restrict__ status) {
  extern __shared__ double ns_dense[];   // [P]: the dense term of every phone for the frame in hand
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint32_t u = u0 + blockIdx.x;
  const uint32_t T = bv.T[u];
  if (T == 0) { if (threadIdx.x == 0) atomicMax(&status[u], SCRF_ERR_EMPTY); return; }
  const uint64_t fb = bv.frame_off[u] - bv.frame_off[u0];
  const double* Su = S + fb * L; const double* Du = TD + fb * L; const double* Ou = TO + fb * L; const double* Eu = TE + fb * P * P;
  double* au = alpha + fb * L; double* bu = beta + fb * L;
  const uint32_t grp = threadIdx.x / NSFB_G, gl = threadIdx.x % NSFB_G, n_grp = blockDim.x / NSFB_G;
  int err = 0;
  for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) au[c] = Su[c];
  __syncthreads();
  for (uint32_t t = 1; t < T; t++) {
    const double* pa = au + (uint64_t)(t - 1) * L;
    for (uint32_t q = grp; q < P; q += n_grp) {   // into the start state of phone q
      const double* Et = Eu + (uint64_t)t * P * P + q;
      double m = -INFINITY;
      for (uint32_t p = gl; p < P; p += NSFB_G) m = fmax(m, pa[p * K + K - 1] + Et[(uint64_t)p * P]);
      m = ns_group_max(m);
      double sum = 0.0;
      for (uint32_t p = gl; p < P; p += NSFB_G) sum += ns_exp((pa[p * K + K - 1] + Et[(uint64_t)p * P]) - m, &err);
      sum = ns_group_sum(sum);
      if (gl == 0) ns_dense[q] = m + ns_log(sum, &err);
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) {
      double v = pa[c] + Du[(uint64_t)t * L + c];
      if (c % K == 0) v = ns_logadd2(v, ns_dense[c / K], &err);
      else v = ns_logadd2(v, pa[c - 1] + Ou[(uint64_t)t * L + c - 1], &err);
      au[(uint64_t)t * L + c] = v + Su[(uint64_t)t * L + c];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double* a = au + (uint64_t)(T - 1) * L;
    double maxv = a[0];
    for (uint32_t i = 1; i < L; i++) maxv = fmax(maxv, a[i]);
    double sum = 0.0;
    for (uint32_t i = 0; i < L; i++) sum += ns_exp(a[i] - maxv, &err);
    zx_out[u] = maxv + ns_log(sum, &err);
  }
  for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) bu[(uint64_t)(T - 1) * L + c] = 0.0;
  __syncthreads();
  for (uint32_t t = T - 1; t-- > 0;) {
    const uint64_t n = t + 1;
    const double* bn = bu + n * L; const double* Sn = Su + n * L;
    for (uint32_t pe = grp; pe < P; pe += n_grp) {   // out of the end state of phone pe
      const double* En = Eu + n * P * P + (uint64_t)pe * P;
      double m = -INFINITY;
      for (uint32_t q = gl; q < P; q += NSFB_G) m = fmax(m, En[q] + (bn[q * K] + Sn[q * K]));
      m = ns_group_max(m);
      double sum = 0.0;
      for (uint32_t q = gl; q < P; q += NSFB_G) sum += ns_exp((En[q] + (bn[q * K] + Sn[q * K])) - m, &err);
      sum = ns_group_sum(sum);
      if (gl == 0) ns_dense[pe] = m + ns_log(sum, &err);
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < L; p += blockDim.x) {
      double v = (bn[p] + Sn[p]) + Du[n * L + p];
      if ((p + 1) % K == 0) v = ns_logadd2(v, ns_dense[(p + 1) / K - 1], &err);
      else v = ns_logadd2(v, Ou[n * L + p] + (bn[p + 1] + Sn[p + 1]), &err);
      bu[(uint64_t)t * L + p] = v;
    }
    __syncthreads();
  }
  if (err) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// computeExpF :273-331: G[t][c]; XD[t][c] (self), XO[t][c] (from c-1), XE[t][p*P + q]; masses per frame
__global__ __launch_bounds__(256) void k_ns_post(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                                                 uint64_t n_frames, const double* __restrict__ S, const double* __restrict__ TD,
                                                 const double* __restrict__ TO, const double* __restrict__ TE,
                                                 const double* __restrict__ alpha, const double* __restrict__ beta,
                                                 const double* __restrict__ zx, double* __restrict__ G, double* __restrict__ XD,
                                                 double* __restrict__ XO, double* __restrict__ XE, double* __restrict__ mass_s,
                                                 double* __restrict__ mass_t, int* __restrict__ status) {
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_frames * L) return;
  const uint64_t fr = e / L;
  const uint32_t c = (uint32_t)(e % L);
  const uint64_t gf = bv.frame_off[u0] + fr;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const double Zx = zx[u];
  int err = 0;
  const double sb = S[fr * L + c] + beta[fr * L + c];
  const double g = ns_exp(alpha[fr * L + c] + beta[fr * L + c] - Zx, &err);
  G[fr * L + c] = g;
  atomicAdd(&mass_s[fr], g);
  double mt = 0.0, xd = 0.0, xo = 0.0;
  if (t > 0) {
    const double* pa = alpha + (fr - 1) * L;
    xd = ns_exp(pa[c] + TD[fr * L + c] + S[fr * L + c] + beta[fr * L + c] - Zx, &err);
    mt += xd;
    if (c % K == 0) {
      for (uint32_t p = 0; p < P; p++) {
        const double x = ns_exp(pa[p * K + K - 1] + TE[fr * P * P + p * P + c / K] + S[fr * L + c] + beta[fr * L + c] - Zx, &err);
        XE[fr * P * P + p * P + c / K] = x;
        mt += x;
      }
    } else {
      xo = ns_exp(pa[c - 1] + TO[fr * L + c - 1] + S[fr * L + c] + beta[fr * L + c] - Zx, &err);
      mt += xo;
    }
    atomicAdd(&mass_t[fr], mt);
  } else if (c % K == 0) {
    for (uint32_t p = 0; p < P; p++) XE[fr * P * P + p * P + c / K] = 0.0;
  }
  (void)sb;
  XD[fr * L + c] = xd;
  XO[fr * L + c] = xo;
  if (err) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// per utterance: label range, numerator (score of the labelled sequence over allowed transitions), mass self-checks
// (:333-349: state and transition mass of every node within [0.9, 1.1]; first node: transition mass 1)
__global__ void k_ns_numer(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ S,
                           const double* __restrict__ TD, const double* __restrict__ TO, const double* __restrict__ TE,
                           const double* __restrict__ mass_s, const double* __restrict__ mass_t, double* __restrict__ numer,
                           int* __restrict__ status) {
  const uint32_t ul = blockIdx.x * blockDim.x + threadIdx.x;
  if (ul >= n_utts) return;
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint32_t u = u0 + ul, T = bv.T[u];
  const uint64_t gf0 = bv.frame_off[u], fb = gf0 - bv.frame_off[u0];
  double tot = 0.0;
  int err = 0;
  for (uint32_t t = 0; t < T; t++) {
    const double ms = mass_s[fb + t], mt = t == 0 ? 1.0 : mass_t[fb + t];
    if (!(ms <= 1.1) || !(ms >= 0.9) || !(mt <= 1.1) || !(mt >= 0.9)) err = err ? err : SCRF_ERR_NUMERIC;
    const uint32_t c = bv.labels ? bv.labels[gf0 + t] : SCRF_LAB_BAD;
    if (c == SCRF_LAB_BAD) continue;
    if (c >= L) { err = SCRF_ERR_BAD_LABEL; continue; }
    tot += S[(fb + t) * L + c];
    if (t == 0) continue;
    const uint32_t p = bv.labels[gf0 + t - 1];
    if (p >= L) continue;
    if (p == c) tot += TD[(fb + t) * L + c];
    else if (c % K == 0) { if ((p + 1) % K == 0) tot += TE[(fb + t) * P * P + (p / K) * P + c / K]; }
    else if (p == c - 1) tot += TO[(fb + t) * L + c - 1];
  }
  numer[u] = tot;
  if (err) atomicMax(&status[u], err);
}

// gradient: one thread per weight, frames ascending
__global__ __launch_bounds__(256) void k_ns_expf(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                                                 uint64_t n_frames, const float* __restrict__ X, const double* __restrict__ G,
                                                 const double* __restrict__ XD, const double* __restrict__ XO,
                                                 const double* __restrict__ XE, uint64_t frames_per_slice, double* __restrict__ slab) {
  // blockIdx.y: a slice of frames_per_slice frames whose partial sums go to slab[slice][weight]; k_ns_expf_reduce adds
  // the slices in order (fixed summation order: runs are reproducible)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lay.lambda_len) return;
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  // the label whose block holds weight i: state_idx is increasing
  uint32_t lo = 0, hi = L - 1;
  while (lo < hi) {
    const uint32_t mid = (lo + hi + 1) / 2;
    if (lay.state_idx_k(mid) <= i) lo = mid; else hi = mid - 1;
  }
  const uint32_t c = lo, r = i - lay.state_idx_k(c);
  const bool is_state = r < lay.nsf;
  uint32_t k = r, j = 0, plab = c;
  if (!is_state) { j = (r - lay.nsf) / lay.ntf; k = (r - lay.nsf) % lay.ntf; if (j > 0) plab = (c % K == 0) ? (j - 1) * K + K - 1 : c - 1; }
  const bool bias = is_state ? (lay.use_sb && k == lay.nsfe) : (lay.use_tb && k == lay.ntfe);
  const uint32_t col = is_state ? lay.sfs + k : lay.tfs + k;
  const double bval = is_state ? lay.sbv : lay.tbv;
  double expected = 0.0, observed = 0.0;
  const uint64_t f0 = (uint64_t)blockIdx.y * frames_per_slice;
  const uint64_t f1 = f0 + frames_per_slice < n_frames ? f0 + frames_per_slice : n_frames;
  for (uint64_t fr = f0; fr < f1; fr++) {
    const double x = bias ? bval : (double)X[fr * lay.F + col];
    const uint64_t gf = bv.frame_off[u0] + fr;
    const uint32_t u = frame_u[gf];
    const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
    const uint32_t tl = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
    if (is_state) {
      expected += G[fr * L + c] * x;
      if (tl == c) observed += x;
    } else if (t > 0 && bv.labels[gf - 1] <= L) {   // computeExpF :289-291: no transition terms behind an unlabelled frame
      const double xi = j == 0 ? XD[fr * L + c] : (c % K == 0 ? XE[fr * P * P + (plab / K) * P + c / K] : XO[fr * L + c]);
      expected += xi * x;
      if (tl == c && bv.labels[gf - 1] == plab) observed += x;
    }
  }
  slab[(uint64_t)blockIdx.y * lay.lambda_len + i] = observed - expected;
}
__global__ __launch_bounds__(256) void k_ns_expf_reduce(const double* __restrict__ slab, uint32_t n_slices, uint32_t n,
                                                        double* __restrict__ grad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (uint32_t s = 0; s < n_slices; s++) v += slab[(uint64_t)s * n + i];
  grad[i] += v;
}

// decoders/CRF_LatticeBuilder.h nStateBuildLattice: one thread per (frame, label) writes that state's incoming arcs
__global__ void k_ns_arcs(ScrfLayout lay, uint32_t T, const double* __restrict__ S, const double* __restrict__ TD,
                          const double* __restrict__ TO, const double* __restrict__ TE, float final_w, scrf_arc* __restrict__ arcs) {
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)(T + 1) * L) return;
  const uint32_t t = (uint32_t)(e / L), c = (uint32_t)(e % L);
  const uint64_t per_node = (uint64_t)P * (P + 1) + (uint64_t)(L - P) * 2;
  if (t == T) {   // final arcs, from every label of the last node
    scrf_arc a;
    a.src = 1 + (int)((uint64_t)(T - 1) * L + c); a.ilabel = 0; a.olabel = 0; a.w = final_w; a.dst = 1 + (int)((uint64_t)T * L);
    arcs[(uint64_t)L + (uint64_t)(T - 1) * per_node + c] = a;
    return;
  }
  const int cur_state = 1 + (int)((uint64_t)t * L + c);
  const double sv = S[(uint64_t)t * L + c];
  if (t == 0) {
    scrf_arc a;
    a.src = 0; a.ilabel = (int)c + 1; a.olabel = (int)c + 1; a.w = (float)(-1 * sv); a.dst = cur_state;
    arcs[c] = a;
    return;
  }
  const uint32_t nstart = (c + K - 1) / K;   // start states among the labels before c
  scrf_arc* out = arcs + (uint64_t)L + (uint64_t)(t - 1) * per_node + (uint64_t)nstart * (P + 1) + (uint64_t)(c - nstart) * 2;
  const int pbase = 1 + (int)((uint64_t)(t - 1) * L);
  scrf_arc a;
  a.ilabel = (int)c + 1; a.olabel = (int)c + 1; a.dst = cur_state;
  if (c % K == 0) {
    for (uint32_t p = 0; p < P; p++) {
      a.src = pbase + (int)(p * K + K - 1);
      a.w = (float)(-1 * (TE[(uint64_t)t * P * P + p * P + c / K] + sv));
      out[p] = a;
    }
    a.src = pbase + (int)c;
    a.w = (float)(-1 * (TD[(uint64_t)t * L + c] + sv));
    out[P] = a;
  } else {
    a.src = pbase + (int)c - 1;
    a.w = (float)(-1 * (TO[(uint64_t)t * L + c - 1] + sv));
    out[0] = a;
    a.src = pbase + (int)c;
    a.w = (float)(-1 * (TD[(uint64_t)t * L + c] + sv));
    out[1] = a;
  }
}

// ShortestPath on that lattice: states relaxed in id order, strict improvement; a state's candidates in ascending source
// state order (the previous node's labels ascending)
__global__ __launch_bounds__(256) void k_ns_viterbi(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ S,
                                                    const double* __restrict__ TD, const double* __restrict__ TO,
                                                    const double* __restrict__ TE, float* __restrict__ vc, uint16_t* __restrict__ bp,
                                                    uint32_t* __restrict__ out_labels, uint32_t* __restrict__ out_n,
                                                    float* __restrict__ out_cost) {
  const uint32_t L = lay.L, K = lay.K, P = L / K;
  const uint32_t u = u0 + blockIdx.x;
  const uint32_t T = bv.T[u];
  const uint64_t fb = bv.frame_off[u] - bv.frame_off[u0];
  const double* Su = S + fb * L; const double* Du = TD + fb * L; const double* Ou = TO + fb * L; const double* Eu = TE + fb * P * P;
  float* vu = vc + fb * L;
  uint16_t* bu = bp + fb * L;
  uint32_t* outl = out_labels + bv.frame_off[u];
  for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) { vu[c] = 0.0f + (float)(-1 * Su[c]); bu[c] = 0xffff; }
  __syncthreads();
  for (uint32_t t = 1; t < T; t++) {
    const float* pc = vu + (uint64_t)(t - 1) * L;
    for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) {
      const double sv = Su[(uint64_t)t * L + c];
      float best = INFINITY;
      uint32_t bpv = 0xffff;
      const float wself = (float)(-1 * (Du[(uint64_t)t * L + c] + sv));
      if (c % K == 0) {
        bool self_done = false;
        for (uint32_t p = 0; p < P; p++) {
          const uint32_t src = p * K + K - 1;
          if (!self_done && c < src) {   // the self arc's source comes before this end state
            const float cs = pc[c] + wself;
            if (cs < best) { best = cs; bpv = c; }
            self_done = true;
          }
          const float cc = pc[src] + (float)(-1 * (Eu[(uint64_t)t * P * P + p * P + c / K] + sv));
          if (cc < best) { best = cc; bpv = src; }
        }
        if (!self_done) {
          const float cs = pc[c] + wself;
          if (cs < best) { best = cs; bpv = c; }
        }
      } else {
        const float c1 = pc[c - 1] + (float)(-1 * (Ou[(uint64_t)t * L + c - 1] + sv));
        if (c1 < best) { best = c1; bpv = c - 1; }
        const float cs = pc[c] + wself;
        if (cs < best) { best = cs; bpv = c; }
      }
      vu[(uint64_t)t * L + c] = best;
      bu[(uint64_t)t * L + c] = (uint16_t)bpv;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float best = INFINITY;
    int bl = -1;
    for (uint32_t c = 0; c < L; c++) {
      const float cst = vu[(uint64_t)(T - 1) * L + c] + 0.0f;
      if (cst < best) { best = cst; bl = (int)c; }
    }
    uint32_t n = 0;
    if (bl >= 0) {
      uint32_t c = (uint32_t)bl;
      for (int t = (int)T - 1; t >= 0; t--) {
        outl[t] = c;
        n++;
        c = bu[(uint64_t)t * L + c];
      }
      best = best + 0.0f;
    }
    out_n[u] = n;
    out_cost[u] = best;
  }
}

// ------------------------------------------------------------------------------------------
void launch_ns_scores(hipStream_t st, const ScrfLayout& lay, const float* X, uint64_t n_frames, const double* lambda, double* S,
                      double* TD, double* TO, double* TE) {
  const uint64_t n = n_frames * lay.L;
  if (n == 0) return;
  hipLaunchKernelGGL(k_ns_scores, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, X, n_frames, lambda, S, TD, TO, TE);
}
void launch_ns_fb(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* S,
                  const double* TD, const double* TO, const double* TE, double* alpha, double* beta, double* zx, int* status) {
  if (n_utts == 0) return;
  const uint32_t P = lay.L / lay.K;
  uint32_t nt = 256;
  while (nt < 1024 && nt < P * NSFB_G) nt *= 2;   // a lane group per phone when they fit
  hipLaunchKernelGGL(k_ns_fb, dim3(n_utts), dim3(nt), sizeof(double) * P, st, lay, bv, u0, S, TD, TO, TE, alpha, beta, zx, status);
}
void launch_ns_post(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint32_t n_utts,
                    uint64_t n_frames, const double* S, const double* TD, const double* TO, const double* TE, const double* alpha,
                    const double* beta, const double* zx, double* G, double* XD, double* XO, double* XE, double* mass_s,
                    double* mass_t, double* numer, int* status) {
  const uint64_t n = n_frames * lay.L;
  if (n == 0) return;
  hipLaunchKernelGGL(k_ns_post, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, bv, frame_u, u0, n_frames, S, TD, TO, TE,
                     alpha, beta, zx, G, XD, XO, XE, mass_s, mass_t, status);
  hipLaunchKernelGGL(k_ns_numer, dim3((n_utts + 63) / 64), dim3(64), 0, st, lay, bv, u0, n_utts, S, TD, TO, TE, mass_s, mass_t, numer,
                     status);
}
uint32_t ns_expf_slices(uint64_t n_frames) {
  const uint64_t n = (n_frames + 127) / 128;   // at least 128 frames per slice
  return (uint32_t)(n < 1 ? 1 : (n > 256 ? 256 : n));
}
void launch_ns_expf(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                    const float* X, const double* G, const double* XD, const double* XO, const double* XE, double* slab, double* grad) {
  if (n_frames == 0) return;
  const uint32_t ns = ns_expf_slices(n_frames);
  const uint64_t fps = (n_frames + ns - 1) / ns;
  hipLaunchKernelGGL(k_ns_expf, dim3((lay.lambda_len + 255) / 256, ns), dim3(256), 0, st, lay, bv, frame_u, u0, n_frames, X, G, XD, XO,
                     XE, fps, slab);
  hipLaunchKernelGGL(k_ns_expf_reduce, dim3((lay.lambda_len + 255) / 256), dim3(256), 0, st, slab, ns, lay.lambda_len, grad);
}
uint64_t ns_num_arcs(uint32_t T, uint32_t L, uint32_t K) {
  if (T == 0) return 0;
  const uint64_t P = L / K;
  return (uint64_t)L + (uint64_t)(T - 1) * (P * (P + 1) + (uint64_t)(L - P) * 2) + L;
}
void launch_ns_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t T, const double* S, const double* TD, const double* TO,
                    const double* TE, float final_w, scrf_arc* arcs) {
  if (T == 0) return;
  const uint64_t n = (uint64_t)(T + 1) * lay.L;
  hipLaunchKernelGGL(k_ns_arcs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, T, S, TD, TO, TE, final_w, arcs);
}
void launch_ns_viterbi(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* S,
                       const double* TD, const double* TO, const double* TE, float* vc, uint16_t* bp, uint32_t* out_labels,
                       uint32_t* out_n, float* out_cost) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_ns_viterbi, dim3(n_utts), dim3(256), 0, st, lay, bv, u0, S, TD, TO, TE, vc, bp, out_labels, out_n, out_cost);
}
