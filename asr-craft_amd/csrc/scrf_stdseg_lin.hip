// scrf_stdseg_lin.hip -- the STDSEG model (nodes/CRF_StdSegStateNode.cpp, labels carry the duration) on the TRAINING
// path when its transitions carry only the bias (`stdstate` map: no transition features).
//
// The reference recursion runs over full labels clab = (dur-1)*La + phone:
//   alpha[t][clab] = S[t][clab] + logAdd_plab(alpha[t-dur][plab] + M[plab][clab])          (computeAlpha :136-180)
//   beta[t][plab]  = logAdd_{clab' = (dur',ph')}(M[plab][clab'] + S[t+dur'][clab'] + beta[t+dur'][clab'])   (:211-260)
// i.e. (La*D)^2 log-add terms per node (230 400 at 48 phones x 10 durations).  scrf_stdseg.hip keeps the reference's
// per-row transition matrices MX[N_seg][nLabs][La] and xi (545 MB each per TIMIT-shape utterance) and the log-domain
// sums: 0.25 k utterances/s.  With bias-only transitions M is ONE nLabs x nLabs table, so:
//   * E = exp(M - max M) is built once per call; the recursion carries mantissas with per-node log-scales (as k_dp_lin
//     does) and becomes a matrix-vector product per node with E: no exp / log inside the sum, 2 * nLabs of them per node;
//   * nothing per-row is materialised: node values live in [D][frames][La] arrays (duration-major, so that the rows of
//     one duration -- which share a state-weight block -- are a dense matrix for the MFMA contractions of scrf_mfma.hip);
//   * the expected transition counts are E o (A^T B) over the nodes, one fp64-MFMA product per duration (k_sl_atb),
//     xi never exists.
// Sums are re-associated (matrix-vector order instead of logAdd's index order): results agree with the log-domain
// kernels to ~1e-12 (FAST contract).  EXACT precision, the parity hooks, lattices and decoding keep scrf_stdseg.hip.
#include "scrf_dp_common.h"
#include "scrf_kernels.h"

#include <math.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define SL_NT 512          // threads of the recursion workgroup: one per full label (nLabs <= 512)
#define SL_LOG0 (-1e300)

// ------------------------------------------------------------------------------------------
// tables: M[p][c] = lambda[trans_idx(p, c)] * tbv (the bias is the only transition function), mmax = max M,
// E[p][c] = exp(M - mmax), ET = E^T.  One workgroup.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sl_tables(ScrfLayout lay, const double* __restrict__ lambda, double* __restrict__ E,
                                                    double* __restrict__ ET, double* __restrict__ mmax_out) {
  __shared__ double red[16];
  const uint32_t NL = lay.L, n = NL * NL;
  double m = -INFINITY;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) m = fmax(m, lambda[lay.trans_idx(i / NL, i % NL)] * lay.tbv);
  m = wave_max_f64_dpp(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = red[0];
  for (uint32_t w = 1; w < blockDim.x / 64; w++) m = fmax(m, red[w]);
  if (threadIdx.x == 0) *mmax_out = m;
  for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
    const uint32_t p = i / NL, c = i % NL;
    const double e = exp(lambda[lay.trans_idx(p, c)] * lay.tbv - m);
    E[i] = e;
    ET[(size_t)c * NL + p] = e;
  }
}

// xrow[d0][f] = row of X holding window (t, d0 + 1) of frame f of the chunk (0 when the node has no such window: the
// contraction's output for it is never read)
__global__ void k_sl_rows(ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0, uint64_t n_frames, uint32_t D,
                          uint64_t* __restrict__ xrow) {
  const uint64_t fi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint64_t r0 = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const uint32_t nd = scrf_node_max_dur(t, D);
  for (uint32_t d0 = 0; d0 < D; d0++) xrow[(uint64_t)d0 * n_frames + fi] = d0 < nd ? r0 + d0 : 0;
}

// workgroup maximum / sum of one value per thread (SL_NT threads = 8 wavefronts); two barriers each
__device__ __forceinline__ double sl_block_max(double v, double* red) {
  v = wave_max_f64_dpp(v);
  __syncthreads();   // red is free again
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double m = red[0];
#pragma unroll
  for (int w = 1; w < SL_NT / 64; w++) m = fmax(m, red[w]);
  return m;
}
__device__ __forceinline__ double sl_block_sum(double v, double* red) {
  v = wave_sum_f64_dpp(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int w = 1; w < SL_NT / 64; w++) s += red[w];   // fixed order
  return s;
}

// ------------------------------------------------------------------------------------------
// k_sl_fb: one workgroup per (utterance, direction); thread = full label.  Node arrays are [D][n_frames][La]
// (duration-major).  LDS: the mantissa vectors of the last / next D nodes (ring) and their log-scales.
//   forward : Ad (alpha, log), Am[f][NL] (mantissas a = exp(alpha - ga[f])), ga[f], zx[u]
//   backward: Bd (beta, log)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SL_NT) void k_sl_fb(ScrfLayout lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint64_t n_frames,
                                                 const double* __restrict__ Sd, const double* __restrict__ E,
                                                 const double* __restrict__ ET, const double* __restrict__ mmax_p,
                                                 double* __restrict__ Ad, double* __restrict__ Bd, double* __restrict__ Am,
                                                 double* __restrict__ ga, double* __restrict__ zx, int* __restrict__ status) {
  extern __shared__ double sl_sm[];
  const uint32_t NL = lay.L, D = lay.D;
  double* ring = sl_sm;               // [D][NL]
  double* gring = ring + D * NL;      // [D]
  double* vec = gring + D;            // [NL] backward: the weighted operand vector of the step
  double* wts = vec + NL;             // [D]
  double* red = wts + D;              // [8]
  const uint32_t u = u0 + (blockIdx.x >> 1);
  const bool bwd = blockIdx.x & 1;
  const uint32_t T = bv.T[u];
  const uint32_t c = threadIdx.x;
  const bool lab_ok = c < NL;
  const uint32_t d0c = lab_ok ? c / La : 0, ph = lab_ok ? c - d0c * La : 0;   // this thread's (duration - 1, phone)
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const double mmax = *mmax_p;
  if (T == 0) {
    if (threadIdx.x == 0 && !bwd) atomicMax(&status[u], SCRF_ERR_EMPTY);
    return;
  }
  int err = 0;
  const size_t nodeoff = ((size_t)d0c * n_frames + f_base) * La + ph;   // + t * La: this label's entry of node t
  if (!bwd) {
    for (uint32_t t = 0; t < T; t++) {
      const uint32_t nd = scrf_node_max_dur(t, D), np = scrf_num_prev(t, D);
      const bool valid = lab_ok && d0c < nd;
      double lg = SL_LOG0;
      if (valid) {
        const double s = Sd[nodeoff + (size_t)t * La];
        if (d0c < np) {
          const uint32_t tp = t - d0c - 1, slot = tp % D;
          const uint32_t pavail = La * scrf_node_max_dur(tp, D);
          const double* rp = ring + slot * NL;
          const double* Ec = E + c;
          double acc0 = 0.0, acc1 = 0.0;
          uint32_t p = 0;
          for (; p + 8 <= pavail; p += 8) {
            double e[8];
#pragma unroll
            for (int i = 0; i < 8; i++) e[i] = Ec[(size_t)(p + i) * NL];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { acc0 = fma(rp[p + i], e[i], acc0); acc1 = fma(rp[p + i + 1], e[i + 1], acc1); }
          }
          for (; p < pavail; p++) acc0 = fma(rp[p], Ec[(size_t)p * NL], acc0);
          const double acc = acc0 + acc1;
          lg = acc > 0.0 ? (s + mmax) + (gring[slot] + log(acc)) : SL_LOG0;
        } else {
          lg = s;   // the utterance-initial segment (dur == t + 1): computeFirstAlpha :189-198
        }
      }
      const double g = sl_block_max(lg, red);   // (its barriers also end every read of the slot node t overwrites)
      if (!(g > -1e299)) err = SCRF_ERR_NUMERIC;   // every label of the node flushed: the log-domain kernels decide
      const double a = valid ? exp(lg - g) : 0.0;
      if (lab_ok) {
        ring[(t % D) * NL + c] = a;
        Am[(f_base + t) * (size_t)NL + c] = a;
        if (valid) Ad[nodeoff + (size_t)t * La] = lg;
      }
      if (threadIdx.x == 0) { gring[t % D] = g; ga[f_base + t] = g; }
      __syncthreads();
    }
    // Zx = logAdd over the last node's labels (computeAlphaSum)
    const double tot = sl_block_sum(lab_ok ? ring[((T - 1) % D) * NL + c] : 0.0, red);
    if (threadIdx.x == 0) {
      if (!(tot > 0.0)) err = SCRF_ERR_NUMERIC;
      zx[u] = gring[(T - 1) % D] + log(tot);
    }
  } else {
    // node T-1: beta = 0 on its labels (setTailBeta); ring entry = exp(S + beta - gb)
    for (uint32_t t = T; t-- > 0;) {
      const uint32_t nd = scrf_node_max_dur(t, D);
      const uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D;
      const bool valid = lab_ok && d0c < nd;
      double be = 0.0;
      if (nn > 0) {
        // operand vector over clab' = (dur', ph'): the entry of node t + dur', brought to the common scale ref
        if (threadIdx.x < D) {
          double ref = -INFINITY;
          for (uint32_t j = 0; j < nn; j++) ref = fmax(ref, gring[(t + 1 + j) % D]);
          wts[threadIdx.x] = threadIdx.x < nn ? exp(gring[(t + 1 + threadIdx.x) % D] - ref) : 0.0;
          if (threadIdx.x == 0) red[SL_NT / 64] = ref;
        }
        __syncthreads();
        if (lab_ok) vec[c] = d0c < nn ? ring[((t + 1 + d0c) % D) * NL + c] * wts[d0c] : 0.0;
        __syncthreads();
        if (valid) {
          const uint32_t nv = nn * La;
          const double* Ep = ET + c;   // ET[clab'][plab = c]
          double acc0 = 0.0, acc1 = 0.0;
          uint32_t q = 0;
          for (; q + 8 <= nv; q += 8) {
            double e[8];
#pragma unroll
            for (int i = 0; i < 8; i++) e[i] = Ep[(size_t)(q + i) * NL];
#pragma unroll
            for (int i = 0; i < 8; i += 2) { acc0 = fma(vec[q + i], e[i], acc0); acc1 = fma(vec[q + i + 1], e[i + 1], acc1); }
          }
          for (; q < nv; q++) acc0 = fma(vec[q], Ep[(size_t)q * NL], acc0);
          const double acc = acc0 + acc1;
          be = acc > 0.0 ? mmax + (red[SL_NT / 64] + log(acc)) : SL_LOG0;
        }
      }
      double lv = SL_LOG0;
      if (valid) {
        Bd[nodeoff + (size_t)t * La] = be;
        lv = be > -1e299 ? Sd[nodeoff + (size_t)t * La] + be : SL_LOG0;
      }
      const double g = sl_block_max(lv, red);
      if (!(g > -1e299)) err = SCRF_ERR_NUMERIC;
      if (lab_ok) ring[(t % D) * NL + c] = valid ? exp(lv - g) : 0.0;
      if (threadIdx.x == 0) gring[t % D] = g;
      __syncthreads();
    }
  }
  if (err) atomicMax(&status[u], err);
}

// ------------------------------------------------------------------------------------------
// k_sl_post: per node (workgroup = frame, thread = full label): gamma = exp(alpha + beta - Zx), R = Y - gamma,
// B' = exp(S + beta + ga[t - dur] - Zx) for labels with a predecessor (the right-hand operand of the transition
// counts), the node's posterior masses (state: all labels; transition: labels with a predecessor, since the xi of a
// label sum to its gamma), the numerator terms and the label checks (computeExpF :345-424).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SL_NT) void k_sl_post(ScrfLayout lay, uint32_t La, ScrfBatchView bv, const uint32_t* __restrict__ frame_u,
                                                   uint32_t u0, uint64_t n_frames, const uint32_t* __restrict__ prev_lab,
                                                   const double* __restrict__ lambda, const double* __restrict__ Sd,
                                                   const double* __restrict__ Ad, const double* __restrict__ Bd,
                                                   const double* __restrict__ ga, const double* __restrict__ zx,
                                                   double* __restrict__ Rd, double* __restrict__ Bp, double* __restrict__ numer_f,
                                                   int* __restrict__ status) {
  __shared__ double red[SL_NT / 64 + 1];
  const uint32_t NL = lay.L, D = lay.D;
  const uint64_t fi = blockIdx.x;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t c = threadIdx.x;
  const bool lab_ok = c < NL;
  const uint32_t d0c = lab_ok ? c / La : 0, ph = lab_ok ? c - d0c * La : 0;
  const uint32_t nd = scrf_node_max_dur(t, D), np = scrf_num_prev(t, D);
  const bool valid = lab_ok && d0c < nd;
  const size_t at = ((size_t)d0c * n_frames + fi) * La + ph;
  const double Zx = zx[u];
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
  int err = 0;
  double g = 0.0, gt = 0.0;
  if (lab_ok) {
    double r = 0.0, bp = 0.0;
    if (valid) {
      const double al = Ad[at], be = Bd[at];
      const double x = al + be - Zx;
      if (x >= 709.782712893384) err = SCRF_ERR_NUMERIC;
      g = (al > -1e299 && be > -1e299) ? exp(x) : 0.0;
      r = ((lab == c) ? 1.0 : 0.0) - g;
      if (d0c < np) {
        gt = g;
        const double y = Sd[at] + be + ga[fi - d0c - 1] - Zx;
        if (y >= 709.782712893384) err = SCRF_ERR_NUMERIC;
        bp = be > -1e299 ? exp(y) : 0.0;
      }
    }
    Rd[at] = r;
    Bp[at] = bp;
  }
  const double ms = sl_block_sum(g, red);
  const double mt = sl_block_sum(gt, red);
  if (threadIdx.x == 0) {
    // :402-421: each sum within [-0.000001, 1.000001]; this node type does not compare the two
    const double mtt = np == 0 ? 1.0 : mt;
    if (!(ms <= 1.000001) || !(ms >= -0.000001) || !(mtt <= 1.000001) || !(mtt >= -0.000001)) err = err ? err : SCRF_ERR_NUMERIC;
    double li = 0.0;
    if (lab != SCRF_LAB_BAD) {
      if (lab >= NL) err = SCRF_ERR_BAD_LABEL;
      else {
        const uint32_t ld0 = lab / La, lph = lab % La;
        if (ld0 < nd) {   // a label the node cannot carry matches nothing
          li += Sd[((size_t)ld0 * n_frames + fi) * La + lph];
          const uint32_t pl = prev_lab[gf];
          if (pl != SCRF_LAB_BAD) {
            if (pl >= NL) err = SCRF_ERR_BAD_LABEL;
            else if (ld0 < np && pl < La * scrf_node_max_dur(t - ld0 - 1, D)) li += lambda[lay.trans_idx(pl, lab)] * lay.tbv;
          }
        }
      }
    }
    numer_f[fi] = li;
  }
  if (err) atomicMax(&status[u], err);
}

// numer[u] = sum of the utterance's node terms, frames ascending
__global__ void k_sl_numer(ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ numer_f,
                           double* __restrict__ numer) {
  const uint32_t ul = blockIdx.x * blockDim.x + threadIdx.x;
  if (ul >= n_utts) return;
  const uint32_t u = u0 + ul, T = bv.T[u];
  const double* nf = numer_f + (bv.frame_off[u] - bv.frame_off[u0]);
  double tot = 0.0;
  for (uint32_t t = 0; t < T; t++) tot += nf[t];
  numer[u] = tot;
}

// ------------------------------------------------------------------------------------------
// k_sl_atb: slab[z][p][c = (d0, ph)] = sum over the frames f of K-chunk z of Am[f - d0 - 1][p] * Bp[d0][f][ph]
// (Bp is 0 where the label has no predecessor, so a row of Am from before the utterance is multiplied by 0).
// One wavefront per (K-chunk, duration, group of SA_MG 16-row tiles of p): SA_MG x NT accumulator tiles
// (NT = ceil(La / 16) <= 4), operands straight from memory, one k-step = 4 frames.
// ------------------------------------------------------------------------------------------
#define SA_MG 3
template <int NT>
__global__ __launch_bounds__(64) void k_sl_atb(uint32_t NL, uint32_t La, uint32_t D, uint64_t n_frames, uint64_t rows_per_chunk,
                                               const double* __restrict__ Am, const double* __restrict__ Bp,
                                               double* __restrict__ slab) {
  const uint32_t lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const uint32_t n_mt = (NL + 15) / 16, n_mg = (n_mt + SA_MG - 1) / SA_MG;
  const uint32_t mg = blockIdx.x % n_mg, d0 = (blockIdx.x / n_mg) % D;
  const uint64_t z = blockIdx.x / ((uint64_t)n_mg * D);
  const uint64_t f_begin = z * rows_per_chunk, f_end = min(n_frames, f_begin + rows_per_chunk);
  v4f64 acc[SA_MG][NT];
#pragma unroll
  for (int i = 0; i < SA_MG; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const double* Bd0 = Bp + (size_t)d0 * n_frames * La;
  for (uint64_t f0 = f_begin; f0 < f_end; f0 += 4) {
    const uint64_t f = f0 + lk;
    const bool ok = f < f_end;
    const uint64_t fa = (ok && f >= d0 + 1) ? f - d0 - 1 : 0;   // (a clamped row meets Bp == 0: the chunk's first frames start an utterance)
    double a[SA_MG], b[NT];
#pragma unroll
    for (int i = 0; i < SA_MG; i++) {
      const uint32_t p = (mg * SA_MG + i) * 16 + li;
      a[i] = (ok && p < NL) ? Am[fa * NL + p] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const uint32_t phj = j * 16 + li;
      b[j] = (ok && f >= d0 + 1 && phj < La) ? Bd0[f * La + phj] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < SA_MG; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double* out = slab + z * (size_t)NL * NL;
#pragma unroll
  for (int i = 0; i < SA_MG; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t p = (mg * SA_MG + i) * 16 + lk + 4 * r, phj = j * 16 + li;
        if (p < NL && phj < La) out[(size_t)p * NL + d0 * La + phj] = acc[i][j][r];
      }
}

// grad[trans_idx(p, c)] += tbv * (observed(p -> c) - exp(M[p][c]) * sum_z slab[z][p][c]); z ascending (fixed order)
__global__ void k_sl_trans_grad(ScrfLayout lay, uint32_t n_chunks, const double* __restrict__ slab, const double* __restrict__ E,
                                const double* __restrict__ mmax_p, const double* __restrict__ obs, double* __restrict__ grad) {
  const uint32_t NL = lay.L;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NL * NL) return;
  double s = 0.0;
  for (uint32_t z = 0; z < n_chunks; z++) s += slab[(size_t)z * NL * NL + i];
  const double ex = E[i] * exp(*mmax_p) * s;
  grad[lay.trans_idx(i / NL, i % NL)] += lay.tbv * (obs[i] - ex);
}

// observed transitions of the chunk: obs[p][c] = number of labelled nodes (label c, previous label p) whose transition
// the node carries (counts are small integers: the order of the atomic additions does not change the sums)
__global__ void k_sl_obs(ScrfLayout lay, uint32_t La, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                         uint64_t n_frames, const uint32_t* __restrict__ prev_lab, double* __restrict__ obs) {
  const uint64_t fi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (fi >= n_frames) return;
  const uint32_t NL = lay.L, D = lay.D;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf], t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD, pl = prev_lab[gf];
  if (lab == SCRF_LAB_BAD || pl == SCRF_LAB_BAD || lab >= NL || pl >= NL) return;
  const uint32_t ld0 = lab / La;
  if (ld0 < scrf_node_max_dur(t, D) && ld0 < scrf_num_prev(t, D) && pl < La * scrf_node_max_dur(t - ld0 - 1, D))
    atomicAdd(&obs[(size_t)pl * NL + lab], 1.0);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
int stdseg_lin_supported(const ScrfLayout& lay, uint32_t La) {
  const size_t sm = sizeof(double) * ((size_t)lay.D * lay.L + 2 * lay.D + lay.L + SL_NT / 64 + 2);
  return !lay.use_tf && lay.use_tb && lay.L <= SL_NT && La <= 64 && lay.D >= 1 && sm <= 150 * 1024;
}
void launch_sl_tables(hipStream_t st, const ScrfLayout& lay, const double* lambda, double* E, double* ET, double* mmax) {
  hipLaunchKernelGGL(k_sl_tables, dim3(1), dim3(1024), 0, st, lay, lambda, E, ET, mmax);
}
void launch_sl_rows(hipStream_t st, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames, uint32_t D, uint64_t* xrow) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_sl_rows, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, bv, frame_u, u0, n_frames, D, xrow);
}
void launch_sl_fb(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint64_t n_frames,
                  const double* Sd, const double* E, const double* ET, const double* mmax, double* Ad, double* Bd, double* Am,
                  double* ga, double* zx, int* status) {
  if (n_utts == 0) return;
  const size_t sm = sizeof(double) * ((size_t)lay.D * lay.L + 2 * lay.D + lay.L + SL_NT / 64 + 2);
  hipFuncSetAttribute((const void*)k_sl_fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(k_sl_fb, dim3(2 * n_utts), dim3(SL_NT), sm, st, lay, La, bv, u0, n_frames, Sd, E, ET, mmax, Ad, Bd, Am, ga, zx, status);
}
void launch_sl_post(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                    uint32_t n_utts, uint64_t n_frames, const uint32_t* prev_lab, const double* lambda, const double* Sd,
                    const double* Ad, const double* Bd, const double* ga, const double* zx, double* Rd, double* Bp,
                    double* numer_f, double* numer, int* status) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_sl_post, dim3((uint32_t)n_frames), dim3(SL_NT), 0, st, lay, La, bv, frame_u, u0, n_frames, prev_lab, lambda, Sd,
                     Ad, Bd, ga, zx, Rd, Bp, numer_f, status);
  hipLaunchKernelGGL(k_sl_numer, dim3((n_utts + 63) / 64), dim3(64), 0, st, bv, u0, n_utts, numer_f, numer);
}
// slab: [n_chunks][NL][NL]; obs: [NL][NL] zeroed by the caller
void launch_sl_trans_counts(hipStream_t st, const ScrfLayout& lay, uint32_t La, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0,
                            uint64_t n_frames, const uint32_t* prev_lab, uint64_t rows_per_chunk, uint32_t n_chunks,
                            const double* Am, const double* Bp, const double* E, const double* mmax, double* slab, double* obs,
                            double* grad) {
  if (n_frames == 0 || n_chunks == 0) return;
  const uint32_t NL = lay.L;
  const uint32_t n_mt = (NL + 15) / 16, n_mg = (n_mt + SA_MG - 1) / SA_MG;
  const dim3 grid((uint32_t)((uint64_t)n_chunks * lay.D * n_mg));
  const uint32_t nt = (La + 15) / 16;
#define SA_GO(N) hipLaunchKernelGGL(k_sl_atb<N>, grid, dim3(64), 0, st, NL, La, lay.D, n_frames, rows_per_chunk, Am, Bp, slab)
  if (nt <= 1) SA_GO(1);
  else if (nt == 2) SA_GO(2);
  else if (nt == 3) SA_GO(3);
  else SA_GO(4);
#undef SA_GO
  hipLaunchKernelGGL(k_sl_obs, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, lay, La, bv, frame_u, u0, n_frames, prev_lab, obs);
  hipLaunchKernelGGL(k_sl_trans_grad, dim3((NL * NL + 255) / 256), dim3(256), 0, st, lay, n_chunks, slab, E, mmax, obs, grad);
}
