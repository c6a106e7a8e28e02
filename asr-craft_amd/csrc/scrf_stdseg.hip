// scrf_stdseg.hip -- the STDSEG model (nodes/CRF_StdSegStateNode.cpp): labels carry the duration,
// clab = (dur-1)*La + phone (La = nActualLabs = nLabs / labMaxDur), every full label has its own state weights and the
// transition matrix runs over FULL labels (previous segment's phone AND duration), taken from the segment's own window.
//
// Per node t the reference holds alpha / beta / stateArray over the node's numAvailLabs = La * min(t+1, D) full labels
// and transMatrix[plab * nLabs + clab].  Here every node value lives per WINDOW ROW: row (t, dur) of an [N_seg][La]
// array is the node's entry clab = (dur-1)*La + phone -- a node's full-label vector is the node's rows back to back --
// and MX[row (t,dur)][plab][phone] is transMatrix[plab*nLabs + clab] (plab: a full label of node t-dur).
//
// This model type is outside the benchmarked path (its label space is La*D wide: (La*D)^2 transition terms per
// node); the kernels are the plain log-domain recursion in the reference's operation order -- one workgroup per
// utterance, sums in index order -- written for parity, not for the roofline.  The lay argument is the FULL-label
// layout (lay.L = nLabs).
#include "scrf_dp_common.h"
#include "scrf_kernels.h"

#include <math.h>

#include <vector>

// ------------------------------------------------------------------------------------------
// row -> (utterance, frame, duration)
// ------------------------------------------------------------------------------------------
__global__ void k_stdseg_rowinfo(ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0, uint64_t n_frames,
                                 uint32_t D, uint32_t* __restrict__ row_t, uint32_t* __restrict__ row_d,
                                 uint32_t* __restrict__ row_u) {
  const uint64_t fi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint64_t r0 = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const uint32_t nd = scrf_node_max_dur(t, D);
  for (uint32_t d = 1; d <= nd; d++) {
    row_t[r0 + d - 1] = t;
    row_d[r0 + d - 1] = d;
    row_u[r0 + d - 1] = u;
  }
}

// ------------------------------------------------------------------------------------------
// scores (computeTransMatrix :83-127): products and sums unfused, features ascending, bias last
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stdseg_scores(ScrfLayout lay, uint32_t La, const float* __restrict__ X,
                                                       uint64_t n_rows, const uint32_t* __restrict__ row_t,
                                                       const uint32_t* __restrict__ row_d,
                                                       const double* __restrict__ lambda, double* __restrict__ S,
                                                       double* __restrict__ MX) {
  const uint32_t NL = lay.L, D = lay.D;
  const uint64_t per_row = (uint64_t)(NL + 1) * La;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n_rows * per_row) return;
  const uint64_t row = e / per_row;
  const uint32_t r = (uint32_t)(e % per_row), q = r / La, lab = r % La;
  const uint32_t t = row_t[row], dur = row_d[row];
  const uint32_t clab = (dur - 1) * La + lab;
  const float* x = X + row * lay.F;
  if (q == NL) {
    uint32_t lc = lay.state_idx(clab);
    double v = 0.0;
    if (lay.use_sf)
      for (uint32_t f = lay.sfs; f <= lay.sfe; f++) v = __dadd_rn(v, __dmul_rn((double)x[f], lambda[lc++]));
    if (lay.use_sb) v = __dadd_rn(v, __dmul_rn(lambda[lc], lay.sbv));
    S[row * La + lab] = v;
    return;
  }
  const uint32_t plab = q;
  const uint32_t np = scrf_num_prev(t, D);
  double v = 0.0;
  if (dur <= np && plab < La * scrf_node_max_dur(t - dur, D)) {
    uint32_t lc = lay.trans_idx(plab, clab);
    if (lay.use_tf)
      for (uint32_t f = lay.tfs; f <= lay.tfe; f++) v = __dadd_rn(v, __dmul_rn((double)x[f], lambda[lc++]));
    if (lay.use_tb) v = __dadd_rn(v, __dmul_rn(lambda[lc], lay.tbv));
  }
  MX[(row * NL + plab) * La + lab] = v;
}

// ------------------------------------------------------------------------------------------
// forward / backward (computeAlpha :136-180, computeFirstAlpha :189-198, computeBeta :211-260, computeAlphaSum):
// one workgroup per utterance; thread = an entry (dur, phone) of the node; logAdd(acc, max, n) = max + log(sum of
// exp(acc[i] - max) in index order).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double stdseg_exp(double x, int* err) {
  if (x >= 709.782712893384) *err = SCRF_ERR_NUMERIC;   // expE: argument >= log(DBL_MAX)
  return exp(x);
}
__device__ __forceinline__ double stdseg_log(double x, int* err) {
  if (!(x > 0.0) || isinf(x)) *err = SCRF_ERR_NUMERIC;  // logE: log of zero / NaN / Inf
  return log(x);
}

#define SSFB_WAVES 16
__device__ __forceinline__ double ss_wave_max(double v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
// Workgroup of 16 wavefronts per utterance, lanes over the phone axis (rows of MX are read as whole 8*La-byte lines).
// Forward: a wavefront owns the window rows (t, dur) of the node, walks the previous node's full labels once and keeps
// a running (max, sum) per lane -- the node's max-shifted log-sum-exp (computeAlpha :150-215) with the shift updated
// on the way instead of found in a first pass, so MX streams from HBM once.  Backward: a wavefront owns full labels
// of node t; per label every lane folds the next rows' terms of its phone into a running (max, sum), the lanes are
// combined at the end (computeBeta :240-330).  Differences to the two-pass form are roundings of the shift only.
__global__ __launch_bounds__(64 * SSFB_WAVES) void k_stdseg_fb(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0,
                                                               const double* __restrict__ S, const double* __restrict__ MX,
                                                               double* __restrict__ alpha, double* __restrict__ beta,
                                                               double* __restrict__ zx_out, int* __restrict__ status) {
  extern __shared__ double ss_bs[];   // [D][La]: beta + state value of the rows (t + dur, dur)
  const uint32_t NL = lay.L, D = lay.D;
  const uint32_t u = u0 + blockIdx.x;
  const uint32_t T = bv.T[u];
  if (T == 0) {
    if (threadIdx.x == 0) atomicMax(&status[u], SCRF_ERR_EMPTY);
    return;
  }
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* Su = S + s_base * La;
  const double* Mu = MX + s_base * (uint64_t)NL * La;
  double* au = alpha + s_base * La;
  double* bu = beta + s_base * La;
  int err = 0;
  // ---- forward
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = scrf_seg_base(t, D);
    const uint32_t nd = scrf_node_max_dur(t, D), np = scrf_num_prev(t, D);
    for (uint32_t dur = 1 + wave; dur <= nd; dur += SSFB_WAVES) {
      for (uint32_t lab = lane; lab < La; lab += 64) {
        const uint64_t at = (base + dur - 1) * La + lab;
        double v = Su[at];
        if (dur <= np) {
          const double* pa = au + scrf_seg_base(t - dur, D) * La;
          const uint32_t pavail = La * scrf_node_max_dur(t - dur, D);
          const double* Mrow = Mu + (base + dur - 1) * (uint64_t)NL * La + lab;
          double m = -INFINITY, sum = 0.0;
          uint32_t plab = 0;
          for (; plab + 32 <= pavail; plab += 32) {   // 32 line-sized loads in flight per wavefront: the walk is HBM-latency bound
            double x[32];
#pragma unroll
            for (int i = 0; i < 32; i++) x[i] = Mrow[(uint64_t)(plab + i) * La];
#pragma unroll
            for (int i = 0; i < 32; i++) x[i] += pa[plab + i];
            double bm = x[0];
#pragma unroll
            for (int i = 1; i < 32; i++) bm = fmax(bm, x[i]);
            const double nm = fmax(m, bm);
            double acc = sum * exp_nonpos(m - nm);
#pragma unroll
            for (int i = 0; i < 32; i++) acc += exp_nonpos(x[i] - nm);
            sum = acc; m = nm;
          }
          for (; plab < pavail; plab++) {
            const double x = pa[plab] + Mrow[(uint64_t)plab * La];
            const double nm = fmax(m, x);
            sum = sum * exp_nonpos(m - nm) + exp_nonpos(x - nm);
            m = nm;
          }
          v = (m + stdseg_log(sum, &err)) + v;
        }
        au[at] = v;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double* a = au + scrf_seg_base(T - 1, D) * La;
    const uint32_t n = La * scrf_node_max_dur(T - 1, D);
    double maxv = a[0];
    for (uint32_t i = 1; i < n; i++) maxv = fmax(maxv, a[i]);
    double sum = 0.0;
    for (uint32_t i = 0; i < n; i++) sum += stdseg_exp(a[i] - maxv, &err);
    zx_out[u] = maxv + stdseg_log(sum, &err);
  }
  // ---- backward
  {
    const uint64_t base = scrf_seg_base(T - 1, D);
    for (uint32_t e = threadIdx.x; e < La * scrf_node_max_dur(T - 1, D); e += blockDim.x) bu[base * La + e] = 0.0;
  }
  __syncthreads();
  for (uint32_t t = T - 1; t-- > 0;) {
    const uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D;
    const uint64_t base = scrf_seg_base(t, D);
    const uint32_t avail = La * scrf_node_max_dur(t, D);
    for (uint32_t e = threadIdx.x; e < nn * La; e += blockDim.x) {
      const uint32_t dur = e / La + 1, lab = e - (dur - 1) * La;
      const uint64_t row = scrf_seg_base(t + dur, D) + dur - 1;
      ss_bs[e] = bu[row * La + lab] + Su[row * La + lab];
    }
    __syncthreads();
    for (uint32_t clab = wave; clab < avail; clab += SSFB_WAVES) {
      double m = -INFINITY, sum = 0.0;
      for (uint32_t lab = lane; lab < La; lab += 64) {
        for (uint32_t d0 = 1; d0 <= nn; d0 += 8) {   // the rows' loads first, then the running (max, sum)
          double x[8];
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const uint32_t dur = d0 + i;
            const uint64_t row = scrf_seg_base(t + (dur <= nn ? dur : nn), D) + (dur <= nn ? dur : nn) - 1;
            x[i] = Mu[(row * NL + clab) * La + lab];
          }
          double bm = -INFINITY;
#pragma unroll
          for (int i = 0; i < 8; i++) {
            const uint32_t dur = d0 + i;
            x[i] = dur <= nn ? x[i] + ss_bs[(dur - 1) * La + lab] : -INFINITY;
            bm = fmax(bm, x[i]);
          }
          const double nm = fmax(m, bm);
          double acc = sum * exp_nonpos(m - nm);
#pragma unroll
          for (int i = 0; i < 8; i++) acc += exp_nonpos(x[i] - nm);
          sum = acc; m = nm;
        }
      }
      const double wm = wave_max_f64_dpp(m);
      double part = (m == -INFINITY) ? 0.0 : sum * exp_nonpos(m - wm);   // lanes past the phone count hold nothing
      part = wave_sum_f64_dpp(part);
      if (lane == 0) bu[base * La + clab] = wm + stdseg_log(part, &err);
    }
    __syncthreads();
  }
  if (err) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// ------------------------------------------------------------------------------------------
// posteriors (computeExpF :345-424): G[row][phone] = exp(alpha + beta - Zx),
// XI[row][plab][phone] = exp(alpha_prev[plab] + MX + S + beta - Zx); per-node masses for the self-checks
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stdseg_post(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0,
                                                     uint64_t n_rows, const uint32_t* __restrict__ row_t,
                                                     const uint32_t* __restrict__ row_d, const uint32_t* __restrict__ row_u,
                                                     const double* __restrict__ S, const double* __restrict__ MX,
                                                     const double* __restrict__ alpha, const double* __restrict__ beta,
                                                     const double* __restrict__ zx, double* __restrict__ G,
                                                     double* __restrict__ XI, double* __restrict__ mass_s,
                                                     double* __restrict__ mass_t, int* __restrict__ status) {
  // one workgroup per window row: its (NL + 1) * La entries in strides, the two posterior masses of the row summed in
  // the workgroup and added to the frame's totals once (a frame has at most D rows)
  __shared__ double red[2][4];
  const uint32_t NL = lay.L, D = lay.D;
  const uint64_t row = blockIdx.x;
  const uint32_t t = row_t[row], dur = row_d[row], u = row_u[row];
  const double Zx = zx[u];
  const uint64_t fidx = (bv.frame_off[u] - bv.frame_off[u0]) + t;
  const uint32_t np = scrf_num_prev(t, D);
  const bool has_prev = dur <= np;
  const uint32_t pavail = has_prev ? La * scrf_node_max_dur(t - dur, D) : 0;
  const double* pa = has_prev ? alpha + ((bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t - dur, D)) * La : nullptr;
  int err = 0;
  double ms = 0.0, mt = 0.0;
  for (uint32_t lab = threadIdx.x; lab < La; lab += blockDim.x) {
    const uint64_t at = row * La + lab;
    const double g = stdseg_exp(alpha[at] + beta[at] - Zx, &err);
    G[at] = g;
    ms += g;
  }
  for (uint32_t r = threadIdx.x; r < NL * La; r += blockDim.x) {
    const uint32_t plab = r / La, lab = r - plab * La;
    const uint64_t at = row * La + lab;
    double x = 0.0;
    if (plab < pavail) {
      x = stdseg_exp(pa[plab] + MX[(row * NL + plab) * La + lab] + S[at] + beta[at] - Zx, &err);
      mt += x;
    }
    XI[(row * NL + plab) * La + lab] = x;
  }
  ms = wave_sum_f64(ms);
  mt = wave_sum_f64(mt);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ms; red[1][threadIdx.x >> 6] = mt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&mass_s[fidx], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    if (has_prev) atomicAdd(&mass_t[fidx], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
  if (err) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

// per utterance: label checks, the numerator (score of the labelled path: what computeStateExpF / computeTransExpF
// return for matching labels, summed over the nodes) and the posterior-mass self-checks (:402-421)
__global__ void k_stdseg_numer(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                               const uint32_t* __restrict__ prev_lab, const double* __restrict__ S,
                               const double* __restrict__ MX, const double* __restrict__ mass_s,
                               const double* __restrict__ mass_t, double* __restrict__ numer, int* __restrict__ status) {
  const uint32_t ul = blockIdx.x * blockDim.x + threadIdx.x;
  if (ul >= n_utts) return;
  const uint32_t NL = lay.L, D = lay.D;
  const uint32_t u = u0 + ul, T = bv.T[u];
  const uint64_t gf0 = bv.frame_off[u], f_base = gf0 - bv.frame_off[u0], s_base = bv.seg_off[u] - bv.seg_off[u0];
  double tot = 0.0;
  int err = 0;
  for (uint32_t t = 0; t < T; t++) {
    const double ms = mass_s[f_base + t], mt = t == 0 ? 1.0 : mass_t[f_base + t];
    // :402-421: each sum within [-0.000001, 1.000001]; this node type does not compare the two
    if (!(ms <= 1.000001) || !(ms >= -0.000001) || !(mt <= 1.000001) || !(mt >= -0.000001)) err = err ? err : SCRF_ERR_NUMERIC;
    const uint32_t lab = bv.labels ? bv.labels[gf0 + t] : SCRF_LAB_BAD;
    if (lab == SCRF_LAB_BAD) continue;
    if (lab >= NL) { err = SCRF_ERR_BAD_LABEL; continue; }
    const uint32_t dur = lab / La + 1, ph = lab % La;
    if (dur > scrf_node_max_dur(t, D)) continue;   // a label the node cannot carry matches nothing
    const uint64_t row = s_base + scrf_seg_base(t, D) + dur - 1;
    tot += S[row * La + ph];
    const uint32_t pl = prev_lab[gf0 + t];
    if (pl != SCRF_LAB_BAD) {
      if (pl >= NL) { err = SCRF_ERR_BAD_LABEL; continue; }
      if (dur <= scrf_num_prev(t, D) && pl < La * scrf_node_max_dur(t - dur, D)) tot += MX[(row * NL + pl) * La + ph];
    }
  }
  numer[u] = tot;
  if (err) atomicMax(&status[u], err);
}

// ------------------------------------------------------------------------------------------
// gradient: one thread per weight: grad[i] += (observed count) - (expected count) over the chunk's rows, rows ascending
// (computeStateExpF / computeTransExpF of ftrmaps/CRF_StdFeatureMap.cpp:130-223 summed over nodes and labels)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stdseg_expf(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                                                     uint64_t n_rows, const uint32_t* __restrict__ row_t,
                                                     const uint32_t* __restrict__ row_d, const uint32_t* __restrict__ row_u,
                                                     const uint32_t* __restrict__ prev_lab, const float* __restrict__ X,
                                                     const double* __restrict__ G, const double* __restrict__ XI,
                                                     double* __restrict__ grad) {
  // thread -> weight with the phone fastest: the lanes of a wavefront read consecutive phones of one (row, previous
  // label) line of XI / one row of G (a weight-major assignment reads one 128-byte line per lane and row)
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= lay.lambda_len) return;
  const uint32_t NL = lay.L;
  const uint32_t lab = tid % La, q = tid / La;
  const uint32_t dur = q / lay.stride + 1, r = q % lay.stride;
  const uint32_t clab = (dur - 1) * La + lab;
  const uint32_t i = clab * lay.stride + r;
  const bool is_state = r < lay.nsf;
  uint32_t plab = 0, k = r;
  if (!is_state) { plab = (r - lay.nsf) / lay.ntf; k = (r - lay.nsf) % lay.ntf; }
  // feature column of this weight, or the bias
  const bool bias = is_state ? (lay.use_sb && k == lay.nsfe) : (lay.use_tb && k == lay.ntfe);
  const uint32_t col = is_state ? lay.sfs + k : lay.tfs + k;
  const double bval = is_state ? lay.sbv : lay.tbv;
  double expected = 0.0, observed = 0.0;
  // the rows of this weight's duration, in row order: window (t, dur) of every node t >= dur - 1 of every utterance
  for (uint32_t u = u0; u < u0 + n_utts; u++) {
   const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
   const uint32_t T = bv.T[u];
   for (uint32_t t = dur - 1; t < T; t++) {
    const uint64_t row = s_base + scrf_seg_base(t, lay.D) + dur - 1;
    const double x = bias ? bval : (double)X[row * lay.F + col];
    const uint32_t tl = bv.labels ? bv.labels[bv.frame_off[u] + t] : SCRF_LAB_BAD;
    if (is_state) {
      expected += G[row * La + lab] * x;
      if (tl == clab) observed += x;
    } else {
      expected += XI[(row * NL + plab) * La + lab] * x;
      if (tl == clab && prev_lab[bv.frame_off[u] + t] == plab &&
          dur <= scrf_num_prev(t, lay.D) && plab < La * scrf_node_max_dur(t - dur, lay.D))
        observed += x;
    }
   }
  }
  grad[i] += observed - expected;
}

__global__ void k_stdseg_sums(const double* __restrict__ numer, const double* __restrict__ zx, uint32_t u0, uint32_t n,
                              double* __restrict__ sums) {
  if (blockIdx.x || threadIdx.x) return;
  double a = 0.0, b = 0.0;
  for (uint32_t i = 0; i < n; i++) { a += numer[u0 + i]; b += zx[u0 + i]; }
  sums[0] += a; sums[1] += b; sums[2] += (double)n;
}

// ------------------------------------------------------------------------------------------
// lattice (decoders/CRF_LatticeBuilder_StdSeg.h:40-590): state 0 = start; state of (t, clab) = 1 + row(t,dur)*La + phone;
// arcs per node dur ascending, phone ascending, previous full label ascending, weight float(-1*(transMatrix + stateArray)),
// utterance-initial durations one arc from the start with float(-1*stateArray); labels clab + 1; then the final state's
// epsilon arcs.  row_arc[row] = index of the row's first arc (host prefix sums); one thread per (row, phone).
// ------------------------------------------------------------------------------------------
__global__ void k_stdseg_arcs(ScrfLayout lay, uint32_t La, uint32_t T, uint64_t n_rows, const uint64_t* __restrict__ row_arc,
                              const double* __restrict__ S, const double* __restrict__ MX, float final_w,
                              scrf_arc* __restrict__ arcs) {
  const uint32_t NL = lay.L, D = lay.D;
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n_last = La * scrf_node_max_dur(T - 1, D);
  if (e >= n_rows * La + n_last) return;
  const int fin = 1 + (int)(n_rows * La);
  if (e >= n_rows * La) {   // final arcs
    const uint32_t pl = (uint32_t)(e - n_rows * La);
    scrf_arc a;
    a.src = 1 + (int)(scrf_seg_base(T - 1, D) * La + pl); a.ilabel = 0; a.olabel = 0; a.w = final_w; a.dst = fin;
    arcs[row_arc[n_rows] + pl] = a;
    return;
  }
  const uint64_t row = e / La;
  const uint32_t lab = (uint32_t)(e % La);
  // (t, dur) of the row
  uint32_t t, dur;
  {
    const uint64_t tri = (uint64_t)D * (D + 1) / 2;
    if (row < tri) {
      t = 0;
      while (scrf_seg_base(t + 1, D) <= row) t++;
    } else {
      t = D + (uint32_t)((row - tri) / D);
    }
    dur = (uint32_t)(row - scrf_seg_base(t, D)) + 1;
  }
  const uint32_t clab = (dur - 1) * La + lab;
  const int cur_state = 1 + (int)(row * La + lab);
  const double sv = S[row * La + lab];
  if (dur <= scrf_num_prev(t, D)) {
    const uint32_t pavail = La * scrf_node_max_dur(t - dur, D);
    const int pstart = 1 + (int)(scrf_seg_base(t - dur, D) * La);
    scrf_arc* out = arcs + row_arc[row] + (uint64_t)lab * pavail;
    for (uint32_t pl = 0; pl < pavail; pl++) {
      scrf_arc a;
      a.src = pstart + (int)pl; a.ilabel = (int)clab + 1; a.olabel = (int)clab + 1;
      a.w = (float)(-1 * (MX[(row * NL + pl) * La + lab] + sv));
      a.dst = cur_state;
      out[pl] = a;
    }
  } else {
    scrf_arc a;
    a.src = 0; a.ilabel = (int)clab + 1; a.olabel = (int)clab + 1; a.w = (float)(-1 * sv); a.dst = cur_state;
    arcs[row_arc[row] + lab] = a;
  }
}

// ------------------------------------------------------------------------------------------
// best path = ShortestPath on that lattice (tropical semiring on float, states relaxed in id order, strict
// improvement: a state keeps the first of equal candidates): a state (t, clab) is entered from node t-dur only, previous
// full label ascending, or from the start; the final state from the last node's labels ascending.
// One workgroup per utterance; vc: [N_seg][La] float path costs, bp: previous full label (0xffff = start).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stdseg_viterbi(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0,
                                                        const double* __restrict__ S, const double* __restrict__ MX,
                                                        float* __restrict__ vc, uint16_t* __restrict__ bp,
                                                        uint32_t* __restrict__ out_labels, uint32_t* __restrict__ out_n,
                                                        float* __restrict__ out_cost) {
  const uint32_t NL = lay.L, D = lay.D;
  const uint32_t u = u0 + blockIdx.x;
  const uint32_t T = bv.T[u];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* Su = S + s_base * La;
  const double* Mu = MX + s_base * (uint64_t)NL * La;
  float* vu = vc + s_base * La;
  uint16_t* bu = bp + s_base * La;
  uint32_t* outl = out_labels + bv.frame_off[u];
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = scrf_seg_base(t, D);
    const uint32_t nd = scrf_node_max_dur(t, D), np = scrf_num_prev(t, D);
    for (uint32_t e = threadIdx.x; e < nd * La; e += blockDim.x) {
      const uint32_t dur = e / La + 1, lab = e % La;
      const uint64_t at = (base + dur - 1) * La + lab;
      const double sv = Su[at];
      float best = INFINITY;
      uint32_t bpv = 0xffff;
      if (dur <= np) {
        const float* pc = vu + scrf_seg_base(t - dur, D) * La;
        const uint32_t pavail = La * scrf_node_max_dur(t - dur, D);
        const double* Mrow = Mu + (base + dur - 1) * (uint64_t)NL * La + lab;
        for (uint32_t pl = 0; pl < pavail; pl++) {
          const float c = pc[pl] + (float)(-1 * (Mrow[(uint64_t)pl * La] + sv));
          if (c < best) { best = c; bpv = pl; }
        }
      } else {
        best = 0.0f + (float)(-1 * sv);
      }
      vu[at] = best;
      bu[at] = (uint16_t)bpv;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float best = INFINITY;
    int bl = -1;
    const uint64_t lbase = scrf_seg_base(T - 1, D);
    const uint32_t n_last = La * scrf_node_max_dur(T - 1, D);
    for (uint32_t cl = 0; cl < n_last; cl++) {
      const float c = vu[lbase * La + cl] + -0.0f;
      if (c < best) { best = c; bl = (int)cl; }
    }
    uint32_t n = 0;
    if (bl >= 0) {
      int t = (int)T - 1;
      uint32_t cl = (uint32_t)bl;
      while (true) {
        outl[n++] = cl;
        const uint32_t dur = cl / La + 1;
        const uint32_t p = bu[scrf_seg_base((uint32_t)t, D) * La + cl];
        if (p == 0xffff) break;
        t -= (int)dur;
        cl = p;
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

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_stdseg_rowinfo(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                           uint32_t D, uint32_t* row_t, uint32_t* row_d, uint32_t* row_u) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_stdseg_rowinfo, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, bv, frame_u, u0, n_frames, D,
                     row_t, row_d, row_u);
}
void launch_stdseg_scores(hipStream_t st, const ScrfLayout& lay, uint32_t La, const float* X, uint64_t n_rows,
                          const uint32_t* row_t, const uint32_t* row_d, const double* lambda, double* S, double* MX) {
  const uint64_t n = n_rows * (uint64_t)(lay.L + 1) * La;
  if (n == 0) return;
  hipLaunchKernelGGL(k_stdseg_scores, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, La, X, n_rows, row_t, row_d,
                     lambda, S, MX);
}
void launch_stdseg_fb(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                      const double* S, const double* MX, double* alpha, double* beta, double* zx, int* status) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_stdseg_fb, dim3(n_utts), dim3(64 * SSFB_WAVES), sizeof(double) * lay.D * La, st, lay, La, bv, u0, S, MX, alpha, beta, zx, status);
}
void launch_stdseg_post(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                        uint64_t n_rows, const uint32_t* row_t, const uint32_t* row_d, const uint32_t* row_u,
                        const uint32_t* prev_lab, const double* S, const double* MX, const double* alpha, const double* beta,
                        const double* zx, double* G, double* XI, double* mass_s, double* mass_t, double* numer, int* status) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k_stdseg_post, dim3((uint32_t)n_rows), dim3(256), 0, st, lay, La, bv, u0, n_rows, row_t, row_d,
                     row_u, S, MX, alpha, beta, zx, G, XI, mass_s, mass_t, status);
  hipLaunchKernelGGL(k_stdseg_numer, dim3((n_utts + 63) / 64), dim3(64), 0, st, lay, La, bv, u0, n_utts, prev_lab, S, MX, mass_s,
                     mass_t, numer, status);
}
void launch_stdseg_expf(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint64_t n_rows,
                        const uint32_t* row_t, const uint32_t* row_d, const uint32_t* row_u, const uint32_t* prev_lab,
                        const float* X, const double* G, const double* XI, double* grad) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k_stdseg_expf, dim3((lay.lambda_len + 255) / 256), dim3(256), 0, st, lay, La, bv, u0, n_utts, n_rows, row_t, row_d,
                     row_u, prev_lab, X, G, XI, grad);
}
void launch_stdseg_sums(hipStream_t st, const double* numer, const double* zx, uint32_t u0, uint32_t n, double* sums) {
  hipLaunchKernelGGL(k_stdseg_sums, dim3(1), dim3(1), 0, st, numer, zx, u0, n, sums);
}
uint64_t stdseg_num_arcs(uint32_t T, uint32_t La, uint32_t D) {
  uint64_t na = 0;
  for (uint32_t t = 0; t < T; t++) {
    const uint32_t np = scrf_num_prev(t, D), nd = scrf_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= np; dur++) na += (uint64_t)La * La * scrf_node_max_dur(t - dur, D);
    na += (uint64_t)(nd - np) * La;
  }
  if (T > 0) na += (uint64_t)La * scrf_node_max_dur(T - 1, D);
  return na;
}
// first arc of every window row of one utterance (n_rows + 1 entries; the last = first final arc)
void stdseg_row_arc_offsets(uint32_t T, uint32_t La, uint32_t D, std::vector<uint64_t>* off) {
  off->clear();
  uint64_t na = 0;
  for (uint32_t t = 0; t < T; t++) {
    const uint32_t np = scrf_num_prev(t, D), nd = scrf_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++) {
      off->push_back(na);
      na += dur <= np ? (uint64_t)La * La * scrf_node_max_dur(t - dur, D) : La;
    }
  }
  off->push_back(na);
}
void launch_stdseg_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t La, uint32_t T, uint64_t n_rows, const uint64_t* row_arc,
                        const double* S, const double* MX, float final_w, scrf_arc* arcs) {
  if (T == 0) return;
  const uint64_t n = n_rows * La + (uint64_t)La * scrf_node_max_dur(T - 1, lay.D);
  hipLaunchKernelGGL(k_stdseg_arcs, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, lay, La, T, n_rows, row_arc, S, MX, final_w, arcs);
}
void launch_stdseg_viterbi(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                           const double* S, const double* MX, float* vc, uint16_t* bp, uint32_t* out_labels, uint32_t* out_n,
                           float* out_cost) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_stdseg_viterbi, dim3(n_utts), dim3(256), 0, st, lay, La, bv, u0, S, MX, vc, bp, out_labels, out_n, out_cost);
}
