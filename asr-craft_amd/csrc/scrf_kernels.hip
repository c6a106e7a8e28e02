// scrf_kernels.hip -- HIP kernels of the segmental-CRF engine for gfx950 (MI355X, wave64).
//
// Kernel inventory (DESIGN.md has the roofline of each):
//   k_windows        segment-window synthesis from raw frames   (io/CRF_InFtrStream_SeqMultiWindow.cpp:413-455)
//   k_scores_exact   feature x weight dot products, reference order, unfused fp64
//                    (ftrmaps/CRF_StdFeatureMap.cpp:65-110 via nodes/...WithoutSegTransFtr.cpp:39-114)
//   k_fb             per-utterance forward / backward / posteriors (nodes/...WithoutSegTransFtr.cpp:123-949)
//   k_expf_gemm      expected-minus-observed feature counts        (ftrmaps/CRF_StdFeatureMap.cpp:130-223)
//   k_reduce_*       deterministic split-K reduction into the gradient
//   k_viterbi        float tropical best path == ShortestPath on the reference lattice
//   k_arcs_*         lattice arc emission in AddArc order (decoders/...WithoutSegTransFtr.h:30-407)
//   k_sgd_step       trainers/CRF_SGTrainer.cpp:299-325
//
// This translation unit is compiled with -ffp-contract=off: the EXACT kernels must not fuse
// multiply and add (the reference is built without FMA contraction).
#include "scrf_kernels.h"
#include "scrf_lse.h"

#include <float.h>
#include <math.h>

// ------------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t find_utt(const uint64_t* off, uint32_t u0, uint32_t u1, uint64_t x) {
  // largest u in [u0,u1) with off[u] <= x
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}

// ------------------------------------------------------------------------------------------
// k_windows: one block per frame of the chunk, threads over the raw feature index.
// Arithmetic identical to the reference: float running sum from the LAST frame backwards
// divided by the length, running max/min, 5 sampled frames at ceil(0.1*k*len)-1.
// ------------------------------------------------------------------------------------------
__global__ void k_windows(const float* __restrict__ frames, const uint64_t* __restrict__ sframe_off,
                          ScrfBatchView bv, uint32_t u0, uint32_t u1, uint32_t W, uint32_t D,
                          uint32_t lctx, uint32_t rctx, int extract, float* __restrict__ X,
                          uint32_t F, uint32_t out_col, int first_only) {
  const uint64_t gf = bv.frame_off[u0] + blockIdx.x;
  const uint32_t u = find_utt(bv.frame_off, u0, u1, gf);
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  // first_only bit 0: the caller reads this stream's columns from the node's FIRST window row alone (per-frame transition
  // features); bit 1: the five sampled blocks are not written (hybrid path: they go through the per-frame projections)
  const bool samples = !(first_only & 2);
  const uint32_t sh = samples ? 0u : 5u * W;   // without the sampled blocks the row is compact: [avg | max | min | onehot], stride F
  const uint32_t avail = (first_only & 1) ? 1u : scrf_node_max_dur(t, D);
  const uint64_t rowbase = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const float* last = frames + (sframe_off[u] + lctx + t) * (uint64_t)W;
  const bool segftr = (D != 1) && extract;
  const uint32_t body = segftr ? 8 * W + D : W;

  for (uint32_t j = threadIdx.x; j < W; j += blockDim.x) {
    float acc_sum = 0.0f, acc_max = last[j], acc_min = last[j];
    for (uint32_t w = 1; w <= avail; w++) {
      const float* first = last - (uint64_t)(w - 1) * W;
      float* o = X + (rowbase + w - 1) * (uint64_t)F + out_col;
      const float* lb = first - (uint64_t)lctx * W;
      for (uint32_t c = 0; c < lctx; c++) o[c * W + j] = lb[c * W + j];
      o += lctx * W;
      if (!segftr) {
        o[j] = first[j];
      } else {
        float ot = (float)((double)w * 0.1);
        if (samples) {
#pragma unroll
          for (int k = 0; k < 5; k++) {
            float prod = ot * (float)(2 * k + 1);
            uint32_t step = (uint32_t)ceilf(prod) - 1u;
            o[k * W + j] = first[(uint64_t)step * W + j];
          }
        }
        float v = first[j];
        acc_sum = __fadd_rn(acc_sum, v);
        o[5 * W + j - sh] = __fdiv_rn(acc_sum, (float)w);
        if (v > acc_max) acc_max = v;
        o[6 * W + j - sh] = acc_max;
        if (v < acc_min) acc_min = v;
        o[7 * W + j - sh] = acc_min;
      }
      o += body;
      const float* rb = extract ? last : first;
      for (uint32_t c = 0; c < rctx; c++) o[c * W + j] = rb[(uint64_t)(c + 1) * W + j];
    }
  }
  if (segftr) {
    // compact rows (no sampled blocks): the one-hot block runs to the end of the row, so that the pad floats are zeros
    const uint32_t Dp = samples ? D : F - 3 * W;
    for (uint32_t idx = threadIdx.x; idx < avail * Dp; idx += blockDim.x) {
      uint32_t w = idx / Dp + 1, k = idx % Dp;
      X[(rowbase + w - 1) * (uint64_t)F + out_col + lctx * W + 8 * W - sh + k] = (k + 1 == w) ? 1.0f : 0.0f;
    }
  }
}

void launch_windows(hipStream_t st, const float* frames, const uint64_t* sframe_off, ScrfBatchView bv,
                    uint32_t u0, uint32_t u1, uint64_t n_frames, uint32_t W, uint32_t D, uint32_t lctx,
                    uint32_t rctx, int extract, float* X, uint32_t F, uint32_t out_col, int first_only) {
  if (n_frames == 0) return;
  uint32_t bs = W <= 64 ? 64 : (W <= 128 ? 128 : 256);
  hipLaunchKernelGGL(k_windows, dim3((uint32_t)n_frames), dim3(bs), 0, st, frames, sframe_off, bv, u0, u1,
                     W, D, lctx, rctx, extract, X, F, out_col, first_only);
}

// first window row (d=1) of every frame of the chunk: the transition features of node t
__global__ void k_frame_rows(ScrfBatchView bv, uint32_t u0, uint32_t u1, uint32_t D, uint64_t n_frames,
                             uint64_t* __restrict__ xrow, int next) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + i;
  const uint32_t u = find_utt(bv.frame_off, u0, u1, gf);
  uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  if (next) t = (t + 1 < bv.T[u]) ? t + 1 : t;  // row of node t+1 (ExpF uses the NEXT node's features)
  xrow[i] = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
}

void launch_frame_rows(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t u1, uint32_t D,
                       uint64_t n_frames, uint64_t* xrow, int next) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_frame_rows, dim3((uint32_t)((n_frames + 255) / 256)), dim3(256), 0, st, bv, u0, u1, D,
                     n_frames, xrow, next);
}

// ------------------------------------------------------------------------------------------
// k_scores_exact: out[row][o] = sum_f (double)x[row][fs+f] * lambda[woff(o)+f]  (+ bias), the
// reference's sequential unfused chain.  block = (64 rows) x (4 output groups of NL outputs).
// ------------------------------------------------------------------------------------------
// The NL weights a wavefront needs per feature are wave-uniform: they are fetched through the scalar
// cache straight into SGPR operands of the multiplies (no LDS traffic for lambda); only the window
// values go through LDS.
#define SC_FC 32
template <int NL>
__global__ __launch_bounds__(256) void k_scores_exact(const float* __restrict__ X, uint32_t F,
                                                      const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                      const double* __restrict__ lambda, ScrfLayout lay,
                                                      int is_trans, uint32_t n_out, double* __restrict__ out) {
  __shared__ float Xs[64][SC_FC + 1];
  const uint32_t tx = threadIdx.x, ty = __builtin_amdgcn_readfirstlane(threadIdx.y), tid = ty * 64 + tx;
  const uint64_t row0 = (uint64_t)blockIdx.x * 64;
  const uint32_t o0 = blockIdx.y * 4 * NL;
  const uint32_t L = lay.L;
  const uint32_t fs = is_trans ? lay.tfs : lay.sfs;
  const uint32_t nfe = is_trans ? lay.ntfe : lay.nsfe;
  const int use_b = is_trans ? lay.use_tb : lay.use_sb;
  const double bv = is_trans ? lay.tbv : lay.sbv;

  double acc[NL];
  const double* wp[NL];   // wave-uniform: weight block of output o0 + ty*NL + k (clamped to a valid one)
#pragma unroll
  for (int k = 0; k < NL; k++) {
    acc[k] = 0.0;
    const uint32_t o = min(o0 + ty * NL + k, n_out - 1);
    wp[k] = lambda + (is_trans ? lay.trans_idx(o / L, o % L) : lay.state_idx(o));
  }

  for (uint32_t f0 = 0; f0 < nfe; f0 += SC_FC) {
    const uint32_t fc = min((uint32_t)SC_FC, nfe - f0);
    for (uint32_t idx = tid; idx < 64 * SC_FC; idx += 256) {
      uint32_t r = idx / SC_FC, c = idx % SC_FC;
      float v = 0.0f;
      if (row0 + r < n_rows && c < fc) {
        uint64_t xr = xrow ? xrow[row0 + r] : row0 + r;
        v = X[xr * F + fs + f0 + c];
      }
      Xs[r][c] = v;
    }
    __syncthreads();
    if (fc == SC_FC) {
#pragma unroll 8
      for (uint32_t c = 0; c < SC_FC; c++) {
        const double xv = (double)Xs[tx][c];
#pragma unroll
        for (int k = 0; k < NL; k++) acc[k] = __dadd_rn(acc[k], __dmul_rn(xv, wp[k][f0 + c]));
      }
    } else {
      for (uint32_t c = 0; c < fc; c++) {
        const double xv = (double)Xs[tx][c];
#pragma unroll
        for (int k = 0; k < NL; k++) acc[k] = __dadd_rn(acc[k], __dmul_rn(xv, wp[k][f0 + c]));
      }
    }
    __syncthreads();
  }
  if (row0 + tx < n_rows) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      uint32_t o = o0 + ty * NL + k;
      if (o < n_out) {
        double v = acc[k];
        if (use_b) v = __dadd_rn(v, __dmul_rn(wp[k][nfe], bv));
        out[(row0 + tx) * n_out + o] = v;
      }
    }
  }
}

void launch_scores_exact(hipStream_t st, const float* X, uint32_t F, const uint64_t* xrow, uint64_t n_rows,
                         const double* lambda, const ScrfLayout& lay, int is_trans, uint32_t n_out, double* out) {
  if (n_rows == 0 || n_out == 0) return;
  dim3 bs(64, 4);
  uint32_t gx = (uint32_t)((n_rows + 63) / 64);
  if (n_out <= 16 || (n_out % 48 != 0 && n_out <= 64)) {
    const int NL = 4;
    hipLaunchKernelGGL(k_scores_exact<NL>, dim3(gx, (n_out + 4 * NL - 1) / (4 * NL)), bs, 0, st, X, F, xrow,
                       n_rows, lambda, lay, is_trans, n_out, out);
  } else {
    const int NL = 12;
    hipLaunchKernelGGL(k_scores_exact<NL>, dim3(gx, (n_out + 4 * NL - 1) / (4 * NL)), bs, 0, st, X, F, xrow,
                       n_rows, lambda, lay, is_trans, n_out, out);
  }
}

// ------------------------------------------------------------------------------------------
// k_fb: forward, backward and posteriors of one utterance per workgroup (log domain, fp64).
//   aPT[t-1][n] = LSE_c(alpha[t-1][c] + M[t][c][n])                         (:1077-1108, :156)
//   ad[t][d][l] = aPT[t-d][l] + S[t][d][l]  (d <= numPrev) | S[t][d][l]      (:168-200)
//   alpha[t][l] = LSE_d ad[t][d][l] ;  Zx = LSE_l alpha[T-1][l]              (:204, StdSeg :447-462)
//   sd[t][n]    = LSE_d(S[t+d][d][n] + beta[t+d][n])                         (:416-429)
//   beta[t][c]  = LSE_n(M[t+1][c][n] + sd[t][n]) ; beta[T-1] = 0             (:445-455)
//   gamma = exp(ad + beta - Zx), xi = exp(alpha[t][c] + M[t+1][c][n] + sd[t][n] - Zx) (:685,:773)
// Outputs R = Y - gamma (over ad, in place) and Y - xi (per frame, or summed per utterance
// when transitions carry only a bias): the operands of the expected-count contraction.
// ------------------------------------------------------------------------------------------
__global__ void k_fb(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ S,
                     const double* __restrict__ M, int m_per_frame, double* __restrict__ AD,
                     double* __restrict__ alpha_g, double* __restrict__ beta_g, double* __restrict__ XI,
                     double* __restrict__ xi_acc, double* __restrict__ numer_out, double* __restrict__ zx_out,
                     int* __restrict__ status, int write_post, int frame_model) {
  extern __shared__ double smem[];
  __shared__ double mass2[2];
  const int L = lay.L, D = lay.D;
  const int NT = blockDim.x, tid = threadIdx.x;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* Su = S + s_base * L;
  double* ADu = AD + s_base * L;
  double* alu = alpha_g + f_base * L;
  const uint32_t* labs = bv.labels ? bv.labels + bv.frame_off[u] : nullptr;
  const size_t LL = (size_t)L * L;

  FbLse c;
  c.L = L;
  c.G = NT / L;
  if (c.G < 1) c.G = 1;
  c.g = tid / L;
  c.j = tid - c.g * L;
  c.active = c.g < c.G && tid < c.G * L;
  double* apt = smem;                 // [D][L]
  double* bring = apt + (size_t)D * L;  // [D][L]
  double* acur = bring + (size_t)D * L; // [L]
  double* sdv = acur + L;             // [L]
  c.red_m = sdv + L;                  // [G][L]
  c.red_s = c.red_m + (size_t)c.G * L; // [G][L]
  __shared__ double zx_s;
  int err = 0;

  if (T == 0) {
    if (tid == 0) { status[u] = SCRF_ERR_EMPTY; numer_out[u] = 0.0; zx_out[u] = 0.0; }
    return;
  }

  // ---- forward -------------------------------------------------------------------------
  for (int l = tid; l < L; l += NT) {
    double a = Su[l];
    ADu[l] = a;
    acur[l] = a;
    alu[l] = a;
  }
  __syncthreads();
  for (int t = 1; t < T; t++) {
    const double* Mt = M + (m_per_frame ? (f_base + t) * LL : 0);
    double r = col_lse(c, L, [&](int ci, int n, bool) { return acur[ci] + Mt[(size_t)ci * L + n]; }, &err);
    if (c.active && c.g == 0) apt[((t - 1) % D) * L + c.j] = r;
    __syncthreads();
    const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
    const uint64_t base = scrf_seg_base(t, D);
    r = col_lse(c, nd,
                [&](int di, int l, bool first) {
                  const int d = di + 1;
                  double s = Su[(base + di) * L + l];
                  double v = (d <= np) ? apt[((t - d) % D) * L + l] + s : s;
                  if (first) ADu[(base + di) * L + l] = v;
                  return v;
                },
                &err);
    if (c.active && c.g == 0) {
      acur[c.j] = r;
      alu[(size_t)t * L + c.j] = r;
    }
    __syncthreads();
  }
  if (tid == 0) {  // computeAlphaSum: logAdd(alphaArray, L) in index order
    double mx = acur[0];
    for (int l = 1; l < L; l++) if (acur[l] > mx) mx = acur[l];
    double sum = 0.0;
    for (int l = 0; l < L; l++) sum += exp(acur[l] - mx);
    zx_s = mx + log(sum);
  }
  __syncthreads();
  const double Zx = zx_s;
  if (isnan(Zx) || isinf(Zx)) err = SCRF_ERR_NUMERIC;

  // ---- backward + posteriors -------------------------------------------------------------
  const double LN_MAX = 709.782712893384;  // log(DBL_MAX): expE overflow guard (CRF_LogMath.cpp:213)
  double numer = 0.0;
  uint32_t cur_next = SCRF_LAB_BAD;
  for (int t = T - 1; t >= 0; t--) {
    const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
    const double* Mn = M + (m_per_frame ? (f_base + t + 1) * LL : 0);
    double* bt = bring + (size_t)(t % D) * L;
    if (nn == 0) {
      for (int l = tid; l < L; l += NT) bt[l] = 0.0;
      __syncthreads();
    } else {
      double r = col_lse(c, nn,
                         [&](int di, int n, bool) {
                           const int d = di + 1;
                           return Su[(scrf_seg_base(t + d, D) + di) * L + n] + bring[((t + d) % D) * L + n];
                         },
                         &err);
      if (c.active && c.g == 0) sdv[c.j] = r;
      __syncthreads();
      r = col_lse(c, L, [&](int n, int ci, bool) { return Mn[(size_t)ci * L + n] + sdv[n]; }, &err);
      if (c.active && c.g == 0) bt[c.j] = r;
      __syncthreads();
    }
    if (beta_g)
      for (int l = tid; l < L; l += NT) beta_g[(f_base + t) * L + l] = bt[l];

    // true labels of this node (computeExpF :620-644)
    uint32_t lab = labs ? labs[t] : SCRF_LAB_BAD;
    uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD, anl = SCRF_LAB_BAD;
    if (lab != SCRF_LAB_BAD) {
      if (lab >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
      al = lab % L;
      ld = lab / L + 1;
    }
    if (cur_next != SCRF_LAB_BAD) {
      if (cur_next >= (uint32_t)L * D) err = SCRF_ERR_BAD_LABEL;
      anl = cur_next % L;
    }
    const int nd = (int)scrf_node_max_dur(t, D);
    const uint64_t base = scrf_seg_base(t, D);
    if (write_post) {
      double gs = 0.0, xs = 0.0;   // posterior mass of the node: state / transition (computeExpF :917-947)
      for (int idx = tid; idx < nd * L; idx += NT) {
        const int di = idx / L, l = idx - di * L;
        double a = ADu[(base + di) * L + l] + bt[l] - Zx;
        if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
        double g = exp(a);
        double y = ((uint32_t)l == al && (uint32_t)(di + 1) == ld) ? 1.0 : 0.0;
        ADu[(base + di) * L + l] = y - g;
        gs += g;
      }
      if (nn > 0) {
        for (int idx = tid; idx < L * L; idx += NT) {
          const int ci = idx / L, n = idx - ci * L;
          double a = alu[(size_t)t * L + ci] + Mn[idx] + sdv[n] - Zx;
          if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
          double x = exp(a);
          double y = ((uint32_t)ci == al && (uint32_t)n == anl) ? 1.0 : 0.0;
          if (XI) XI[(f_base + t) * LL + idx] = y - x;
          else xi_acc[(size_t)blockIdx.x * LL + idx] += y - x;
          xs += x;
        }
      } else if (XI) {
        for (int idx = tid; idx < L * L; idx += NT) XI[(f_base + t) * LL + idx] = 0.0;
      }
      if (tid == 0) { mass2[0] = 0.0; mass2[1] = 0.0; }
      __syncthreads();
      for (int o = 32; o >= 1; o >>= 1) { gs += __shfl_xor(gs, o); xs += __shfl_xor(xs, o); }
      if ((tid & 63) == 0) { atomicAdd(&mass2[0], gs); atomicAdd(&mass2[1], xs); }
      __syncthreads();
      if (tid == 0 && !scrf_mass_ok(mass2[0], mass2[1], nn == 0, frame_model != 0)) err = SCRF_ERR_NUMERIC;
    }
    if (tid == 0 && lab != SCRF_LAB_BAD) {
      double nodeLi = 0.0;
      if (ld <= (uint32_t)nd) nodeLi += Su[(base + ld - 1) * L + al];
      if (nn > 0 && anl != SCRF_LAB_BAD) nodeLi += Mn[(size_t)al * L + anl];
      numer += nodeLi;
    }
    if (lab != SCRF_LAB_BAD) cur_next = lab;
    __syncthreads();
  }
  if (tid == 0) {
    numer_out[u] = numer;
    zx_out[u] = Zx;
  }
  if (err) atomicMax(&status[u], err);
}

size_t fb_smem_bytes(const ScrfLayout& lay, int NT) {
  int G = NT / (int)lay.L;
  if (G < 1) G = 1;
  return sizeof(double) * ((size_t)2 * lay.D * lay.L + 2 * lay.L + (size_t)2 * G * lay.L);
}

int fb_block_threads(const ScrfLayout& lay) { return lay.L <= 256 ? 256 : 1024; }

void launch_fb(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
               const double* S, const double* M, int m_per_frame, double* AD, double* alpha_g, double* beta_g,
               double* XI, double* xi_acc, double* numer, double* zx, int* status, int write_post, int frame_model) {
  if (n_utts == 0) return;
  int NT = fb_block_threads(lay);
  size_t sm = fb_smem_bytes(lay, NT);
  hipFuncSetAttribute((const void*)k_fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(k_fb, dim3(n_utts), dim3(NT), sm, st, lay, bv, u0, S, M, m_per_frame, AD, alpha_g, beta_g,
                     XI, xi_acc, numer, zx, status, write_post, frame_model);
}

// ------------------------------------------------------------------------------------------
// k_expf_gemm: slab[z][o][f] = sum_{rows of K-chunk z} A[row][o] * xs[row][f], xs = window
// features of the state/transition range with the bias value as the last column.  A = Y - gamma
// (state) or Y - xi (transition), so the result is directly (observed - expected) counts.
// ------------------------------------------------------------------------------------------
#define EG_KT 32
template <int NL>
__global__ __launch_bounds__(256) void k_expf_gemm(const double* __restrict__ A, uint32_t n_out,
                                                   const float* __restrict__ X, uint32_t F,
                                                   const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                   ScrfLayout lay, int is_trans, uint64_t rows_per_chunk,
                                                   double* __restrict__ slab) {
  __shared__ double As[EG_KT][4 * NL];
  const uint32_t tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
  const uint32_t fs = is_trans ? lay.tfs : lay.sfs;
  const uint32_t nfe = is_trans ? lay.ntfe : lay.nsfe;
  const uint32_t nfun = is_trans ? lay.ntf : lay.nsf;
  const double bias = is_trans ? lay.tbv : lay.sbv;
  const uint32_t col = blockIdx.x * 64 + tx;
  const uint32_t o0 = blockIdx.y * 4 * NL;
  const uint64_t r_begin = (uint64_t)blockIdx.z * rows_per_chunk;
  const uint64_t r_end = min(n_rows, r_begin + rows_per_chunk);
  double acc[NL];
#pragma unroll
  for (int k = 0; k < NL; k++) acc[k] = 0.0;

  for (uint64_t r0 = r_begin; r0 < r_end; r0 += EG_KT) {
    const uint32_t kt = (uint32_t)min((uint64_t)EG_KT, r_end - r0);
    for (uint32_t idx = tid; idx < EG_KT * 4 * NL; idx += 256) {
      uint32_t r = idx / (4 * NL), ol = idx % (4 * NL);
      double v = 0.0;
      if (r < kt && o0 + ol < n_out) v = A[(r0 + r) * n_out + o0 + ol];
      As[r][ol] = v;
    }
    __syncthreads();
    for (uint32_t r = 0; r < kt; r++) {
      double xv = 0.0;
      if (col < nfe) {
        uint64_t xr = xrow ? xrow[r0 + r] : r0 + r;
        xv = (double)X[xr * F + fs + col];
      } else if (col == nfe) {
        xv = bias;
      }
#pragma unroll
      for (int k = 0; k < NL; k++) acc[k] = fma(As[r][ty * NL + k], xv, acc[k]);
    }
    __syncthreads();
  }
  if (col < nfun) {
#pragma unroll
    for (int k = 0; k < NL; k++) {
      uint32_t o = o0 + ty * NL + k;
      if (o < n_out) slab[((uint64_t)blockIdx.z * n_out + o) * nfun + col] = acc[k];
    }
  }
}

void launch_expf_gemm(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                      const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, int is_trans,
                      uint64_t rows_per_chunk, uint32_t n_chunks, double* slab) {
  if (n_rows == 0 || n_chunks == 0) return;
  const uint32_t nfun = is_trans ? lay.ntf : lay.nsf;
  dim3 bs(64, 4);
  if (n_out <= 16) {
    const int NL = 4;
    hipLaunchKernelGGL(k_expf_gemm<NL>, dim3((nfun + 63) / 64, (n_out + 4 * NL - 1) / (4 * NL), n_chunks), bs, 0,
                       st, A, n_out, X, F, xrow, n_rows, lay, is_trans, rows_per_chunk, slab);
  } else {
    const int NL = 12;
    hipLaunchKernelGGL(k_expf_gemm<NL>, dim3((nfun + 63) / 64, (n_out + 4 * NL - 1) / (4 * NL), n_chunks), bs, 0,
                       st, A, n_out, X, F, xrow, n_rows, lay, is_trans, rows_per_chunk, slab);
  }
}

// grad[woff(o)+f] += sum_z slab[z][o][f].  Fixed association (bit-reproducible): the z range is cut into
// RS_G contiguous groups, each summed in ascending z by one thread, the group sums added in group order.
#define RS_G 8
__global__ __launch_bounds__(64 * RS_G) void k_reduce_slabs(const double* __restrict__ slab, uint32_t n_chunks, uint32_t n_out,
                                                            ScrfLayout lay, ScrfGemmSpec sp, double* __restrict__ grad) {
  __shared__ double part[RS_G][64];
  const uint32_t nfun = sp.nfun();
  const uint64_t n = (uint64_t)n_out * nfun;
  const uint32_t tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const uint64_t i = (uint64_t)blockIdx.x * 64 + tx;
  const uint32_t per = (n_chunks + RS_G - 1) / RS_G, z0 = g * per, z1 = min(n_chunks, z0 + per);
  double s = 0.0;
  if (i < n)
    for (uint32_t z = z0; z < z1; z++) s += slab[(uint64_t)z * n + i];
  part[g][tx] = s;
  __syncthreads();
  if (g == 0 && i < n) {
    double t = part[0][tx];
#pragma unroll
    for (int k = 1; k < RS_G; k++) t += part[k][tx];
    const uint32_t o = (uint32_t)(i / nfun), f = (uint32_t)(i % nfun);
    grad[sp.woff(lay, o) + f] += t;
  }
}

void launch_reduce_slabs(hipStream_t st, const double* slab, uint32_t n_chunks, uint32_t n_out,
                         const ScrfLayout& lay, const ScrfGemmSpec& sp, double* grad) {
  uint64_t n = (uint64_t)n_out * sp.nfun();
  if (n == 0 || n_chunks == 0) return;
  hipLaunchKernelGGL(k_reduce_slabs, dim3((uint32_t)((n + 63) / 64)), dim3(64 * RS_G), 0, st, slab, n_chunks, n_out,
                     lay, sp, grad);
}

// bias-only transitions: grad[trans_idx(c,n)] += tbv * sum_u xi_acc[u][c*L+n]
__global__ void k_reduce_xiacc(const double* __restrict__ xi_acc, uint32_t n_utts, ScrfLayout lay,
                               double* __restrict__ grad) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t LL = lay.L * lay.L;
  if (i >= LL) return;
  double s = 0.0;
  for (uint32_t u = 0; u < n_utts; u++) s += xi_acc[(uint64_t)u * LL + i];
  grad[lay.trans_idx(i / lay.L, i % lay.L)] += s * lay.tbv;
}

void launch_reduce_xiacc(hipStream_t st, const double* xi_acc, uint32_t n_utts, const ScrfLayout& lay,
                         double* grad) {
  if (n_utts == 0 || !lay.use_tb) return;
  uint32_t LL = lay.L * lay.L;
  hipLaunchKernelGGL(k_reduce_xiacc, dim3((LL + 255) / 256), dim3(256), 0, st, xi_acc, n_utts, lay, grad);
}

// sums of numer/zx over a range, accumulated into sums3 = {numer, zx, n_utts}; 64 lanes take
// contiguous slices, lane partials are combined in lane order (fixed, reproducible)
__global__ void k_batch_sums(const double* __restrict__ numer, const double* __restrict__ zx, uint32_t n,
                             double* __restrict__ sums3) {
  __shared__ double pa[64], pb[64];
  const uint32_t per = (n + 63) / 64, i0 = threadIdx.x * per, i1 = min(n, i0 + per);
  double a = 0.0, b = 0.0;
  for (uint32_t i = i0; i < i1; i++) { a += numer[i]; b += zx[i]; }
  pa[threadIdx.x] = a;
  pb[threadIdx.x] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    a = 0.0; b = 0.0;
    for (int j = 0; j < 64; j++) { a += pa[j]; b += pb[j]; }
    sums3[0] += a;
    sums3[1] += b;
    sums3[2] += (double)n;
  }
}
void launch_batch_sums(hipStream_t st, const double* numer, const double* zx, uint32_t n, double* sums3) {
  hipLaunchKernelGGL(k_batch_sums, dim3(1), dim3(64), 0, st, numer, zx, n, sums3);
}

// ------------------------------------------------------------------------------------------
// k_viterbi: best path over the reference lattice without materialising it.  Float tropical
// semiring, path cost = left-to-right float sum of float(-score) arc weights, relaxation in
// state order with strict improvement (first relaxed wins):
//   boundary(t,l) <- end(t-1,p), p ascending ; end(t,l) <- start (d=t+1) then boundary(t',l),
//   t' ascending (d descending) ; final <- end(T-1,l), l ascending, weight -0.0f.
// Frame model (CRF_LatticeBuilder.h:97-204): state(t,c) <- state(t-1,p), w = float(-(M+S)).
// ------------------------------------------------------------------------------------------
__global__ void k_viterbi(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ S,
                          const double* __restrict__ M, int m_per_frame, int frame_model,
                          const float* __restrict__ Wn, uint16_t* __restrict__ bp_b, uint16_t* __restrict__ bp_e,
                          uint32_t* __restrict__ out_labels, uint32_t* __restrict__ out_n,
                          float* __restrict__ out_cost) {
  extern __shared__ float vsm[];
  const int L = lay.L, D = lay.D;
  const int tid = threadIdx.x, NT = blockDim.x;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const double* Su = S + s_base * L;
  const float* Wu = Wn ? Wn + s_base * L : nullptr;   // float(-1 * score) already formed (segment model)
  const size_t LL = (size_t)L * L;
  float* de_prev = vsm;               // [L]
  float* de_cur = de_prev + L;        // [L]
  float* db = de_cur + L;             // [D][L]
  uint16_t* bpb = bp_b + f_base * L;
  uint16_t* bpe = bp_e + f_base * L;
  uint32_t* outl = out_labels + bv.frame_off[u];
  if (T == 0) {
    if (tid == 0) { out_n[u] = 0; out_cost[u] = INFINITY; }
    return;
  }
  for (int t = 0; t < T; t++) {
    const double* Mt = M + (m_per_frame ? (f_base + t) * LL : 0);
    const uint64_t base = scrf_seg_base(t, D);
    if (frame_model) {
      for (int l = tid; l < L; l += NT) {
        float best = INFINITY;
        int arg = 0;
        if (t == 0) {
          best = 0.0f + (float)(-1.0 * Su[l]);
        } else {
          for (int p = 0; p < L; p++) {
            float w = (float)(-1.0 * (Mt[(size_t)p * L + l] + Su[(size_t)t * L + l]));
            float cst = de_prev[p] + w;
            if (cst < best) { best = cst; arg = p; }
          }
        }
        de_cur[l] = best;
        bpb[(size_t)t * L + l] = (uint16_t)arg;
      }
      __syncthreads();
    } else {
      if (t >= 1) {
        for (int l = tid; l < L; l += NT) {
          float best = INFINITY;
          int arg = 0;
          for (int p = 0; p < L; p++) {
            float cst = de_prev[p] + (float)(-1.0 * Mt[(size_t)p * L + l]);
            if (cst < best) { best = cst; arg = p; }
          }
          db[(t % D) * L + l] = best;
          bpb[(size_t)t * L + l] = (uint16_t)arg;
        }
      }
      __syncthreads();
      for (int l = tid; l < L; l += NT) {
        float best = INFINITY;
        int arg = 0;
        if (t < D) {  // from the start state: duration t+1
          float cst = 0.0f + (Wu ? Wu[(base + t) * L + l] : (float)(-1.0 * Su[(base + t) * L + l]));
          if (cst < best) { best = cst; arg = t + 1; }
        }
        const int tp0 = (t - D + 1 > 1) ? t - D + 1 : 1;
        for (int tp = tp0; tp <= t; tp++) {
          const int d = t - tp + 1;
          float cst = db[(tp % D) * L + l] + (Wu ? Wu[(base + d - 1) * L + l] : (float)(-1.0 * Su[(base + d - 1) * L + l]));
          if (cst < best) { best = cst; arg = d; }
        }
        de_cur[l] = best;
        bpe[(size_t)t * L + l] = (uint16_t)arg;
      }
      __syncthreads();
    }
    for (int l = tid; l < L; l += NT) de_prev[l] = de_cur[l];
    __syncthreads();
  }
  if (tid == 0) {
    float best = INFINITY;
    int bl = -1;
    const float wf = frame_model ? 0.0f : -0.0f;
    for (int l = 0; l < L; l++) {
      float cst = de_prev[l] + wf;
      if (cst < best) { best = cst; bl = l; }
    }
    uint32_t n = 0;
    if (bl >= 0) {
      int t = T - 1, l = bl;
      if (frame_model) {
        while (true) {
          outl[n++] = (uint32_t)l;
          if (t == 0) break;
          l = bpb[(size_t)t * L + l];
          t--;
        }
      } else {
        while (true) {
          int d = bpe[(size_t)t * L + l];
          outl[n++] = (uint32_t)(l + L * (d - 1));
          int ts = t - d + 1;  // first frame of the segment
          if (ts == 0) break;
          l = bpb[(size_t)ts * L + l];
          t = ts - 1;
        }
      }
      for (uint32_t i = 0; i < n / 2; i++) {
        uint32_t tmp = outl[i];
        outl[i] = outl[n - 1 - i];
        outl[n - 1 - i] = tmp;
      }
      best = best + 0.0f;  // Times(distance, Final = One)
    }
    out_n[u] = n;
    out_cost[u] = best;
  }
}

void launch_viterbi(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                    const double* S, const double* M, int m_per_frame, int frame_model, uint16_t* bp_b,
                    uint16_t* bp_e, uint32_t* out_labels, uint32_t* out_n, float* out_cost, const float* Wn) {
  if (n_utts == 0) return;
  int NT = ((int)lay.L + 63) / 64 * 64;
  if (NT > 1024) NT = 1024;
  size_t sm = sizeof(float) * ((size_t)2 * lay.L + (size_t)lay.D * lay.L);
  hipFuncSetAttribute((const void*)k_viterbi, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL(k_viterbi, dim3(n_utts), dim3(NT), sm, st, lay, bv, u0, S, M, m_per_frame, frame_model,
                     frame_model ? nullptr : Wn, bp_b, bp_e, out_labels, out_n, out_cost);
}

// ------------------------------------------------------------------------------------------
// k_viterbi_fast: k_viterbi for the fast decode path -- float arc weights already formed, one
// time-invariant transition matrix, L <= 64: one WAVEFRONT per utterance (lane = label), the
// float(-M) matrix in LDS for the whole workgroup, the frame's D weights fetched into registers
// before the boundary step so that their latency hides under it.  Same additions in the same order
// and the same strict-improvement scans as k_viterbi (bit-identical labels and costs).
// ------------------------------------------------------------------------------------------
// LV > 0 (round 4; L <= LV, L a multiple of 4): the lane's column of float(-M) sits in registers and the previous node's
// costs are read four at a time (one 16-byte broadcast read instead of eight 4-byte ones per four predecessors); two
// comparison chains (even / odd predecessors) merged at the end -- the minimum, and among equal costs the smallest
// predecessor, exactly what the one strict-improvement scan selects.
#define VF_WAVES 4
template <int DMAX, int LV>
__global__ __launch_bounds__(64 * VF_WAVES) void k_viterbi_fast(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                                                                const float* __restrict__ Wn, const double* __restrict__ M,
                                                                uint16_t* __restrict__ bp_b, uint16_t* __restrict__ bp_e,
                                                                uint32_t* __restrict__ out_labels, uint32_t* __restrict__ out_n,
                                                                float* __restrict__ out_cost) {
  extern __shared__ float vsm[];
  const int L = lay.L, D = lay.D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* Mf = vsm;                                         // [L][L] float(-1 * M[p][l])
  float* de_prev = Mf + L * L + wave * (L + D * L);        // [L]
  float* db = de_prev + L;                                 // [D][L]
  for (int i = threadIdx.x; i < L * L; i += 64 * VF_WAVES) Mf[i] = (float)(-1.0 * M[i]);
  __syncthreads();
  const uint32_t ui = blockIdx.x * VF_WAVES + wave;
  if (ui >= n_utts) return;
  const uint32_t u = u0 + ui;
  const int T = (int)bv.T[u];
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t s_base = bv.seg_off[u] - bv.seg_off[u0];
  const float* Wu = Wn + s_base * L;
  uint16_t* bpb = bp_b + f_base * L;
  uint16_t* bpe = bp_e + f_base * L;
  uint32_t* outl = out_labels + bv.frame_off[u];
  if (T == 0) {
    if (lane == 0) { out_n[u] = 0; out_cost[u] = INFINITY; }
    return;
  }
  const int l = lane < L ? lane : L - 1;   // idle lanes shadow the last label, their stores are masked
  const bool act = lane < L;
  float mcol[LV > 0 ? LV : 1];
  if (LV > 0) {
#pragma unroll
    for (int p = 0; p < LV; p++) mcol[p] = p < L ? Mf[p * L + l] : INFINITY;   // past L: never an improvement
  }
  // the frame's D weights are requested a whole frame ahead (round 4: they were requested at the top of their own frame,
  // with only the boundary step to cover an HBM round trip)
  float wv_n[DMAX];
  auto fetch_w = [&](int t, float (&w)[DMAX]) {
    const int tt = t < T ? t : T - 1;
    const uint64_t base = scrf_seg_base(tt, D);
    const int nd = tt + 1 < D ? tt + 1 : D;
#pragma unroll
    for (int i = 0; i < DMAX; i++) w[i] = i < nd ? Wu[(base + i) * L + l] : 0.0f;
  };
  fetch_w(0, wv_n);
  for (int t = 0; t < T; t++) {
    const int nd = t + 1 < D ? t + 1 : D;
    float wv[DMAX];
#pragma unroll
    for (int i = 0; i < DMAX; i++) wv[i] = wv_n[i];
    fetch_w(t + 1, wv_n);
    if (t >= 1) {
      float best = INFINITY;
      int arg = 0;
      if (LV > 0) {
        float b1 = INFINITY;
        int a1 = 0;
#pragma unroll
        for (int p = 0; p < LV; p += 4) {
          if (p < L) {   // wave-uniform; L is a multiple of 4, so the four entries exist
            const float4 dp = *(const float4*)(de_prev + p);
            const float c0 = dp.x + mcol[p], c1 = dp.y + mcol[p + 1], c2 = dp.z + mcol[p + 2], c3 = dp.w + mcol[p + 3];
            if (c0 < best) { best = c0; arg = p; }
            if (c1 < b1) { b1 = c1; a1 = p + 1; }
            if (c2 < best) { best = c2; arg = p + 2; }
            if (c3 < b1) { b1 = c3; a1 = p + 3; }
          }
        }
        if (b1 < best || (b1 == best && a1 < arg)) { best = b1; arg = a1; }
      } else {
        for (int p = 0; p < L; p++) {
          const float cst = de_prev[p] + Mf[p * L + l];
          if (cst < best) { best = cst; arg = p; }
        }
      }
      if (act) { db[(t % D) * L + l] = best; bpb[(size_t)t * L + l] = (uint16_t)arg; }
    }
    __builtin_amdgcn_wave_barrier();
    float best = INFINITY;
    int arg = 0;
    if (t < D) {  // from the start state: duration t+1
#pragma unroll
      for (int i = 0; i < DMAX; i++)
        if (i == t) { const float cst = 0.0f + wv[i]; if (cst < best) { best = cst; arg = t + 1; } }
    }
    // tp ascending = d descending
#pragma unroll
    for (int i = DMAX - 1; i >= 0; i--) {
      const int d = i + 1, tp = t - i;
      if (d <= nd && tp >= 1) {
        const float cst = db[(tp % D) * L + l] + wv[i];
        if (cst < best) { best = cst; arg = d; }
      }
    }
    if (act) { de_prev[l] = best; bpe[(size_t)t * L + l] = (uint16_t)arg; }
    __builtin_amdgcn_wave_barrier();
  }
  __threadfence_block();
  if (lane == 0) {
    float best = INFINITY;
    int bl = -1;
    for (int k = 0; k < L; k++) {
      const float cst = de_prev[k] + -0.0f;
      if (cst < best) { best = cst; bl = k; }
    }
    uint32_t n = 0;
    if (bl >= 0) {
      int t = T - 1, k = bl;
      while (true) {
        const int d = bpe[(size_t)t * L + k];
        outl[n++] = (uint32_t)(k + L * (d - 1));
        const int ts = t - d + 1;
        if (ts == 0) break;
        k = bpb[(size_t)ts * L + k];
        t = ts - 1;
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

int viterbi_fast_supported(const ScrfLayout& lay) {
  return lay.L <= 64 && lay.D >= 2 && lay.D <= 40 && !lay.use_tf &&
         sizeof(float) * ((size_t)lay.L * lay.L + VF_WAVES * ((size_t)lay.L + (size_t)lay.D * lay.L)) <= 64 * 1024;
}
void launch_viterbi_fast(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const float* Wn,
                         const double* M, uint16_t* bp_b, uint16_t* bp_e, uint32_t* out_labels, uint32_t* out_n, float* out_cost) {
  if (n_utts == 0) return;
  const size_t sm = sizeof(float) * ((size_t)lay.L * lay.L + VF_WAVES * ((size_t)lay.L + (size_t)lay.D * lay.L));
  const dim3 grid((n_utts + VF_WAVES - 1) / VF_WAVES), block(64 * VF_WAVES);
#define VF_GO2(N, LV)                                                                                                          \
  do {                                                                                                                         \
    hipFuncSetAttribute((const void*)k_viterbi_fast<N, LV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);              \
    hipLaunchKernelGGL((k_viterbi_fast<N, LV>), grid, block, sm, st, lay, bv, u0, n_utts, Wn, M, bp_b, bp_e, out_labels, out_n, \
                       out_cost);                                                                                              \
  } while (0)
  // register column + 16-byte reads of the previous node's costs: L a multiple of 4 (the LDS vectors stay 16-byte aligned)
  static const bool vec_ok = !(getenv("SCRF_VITERBI_VEC") && atoi(getenv("SCRF_VITERBI_VEC")) == 0);
#define VF_GO(N)                                                     \
  do {                                                               \
    if (vec_ok && lay.L % 4 == 0 && lay.L <= 48) VF_GO2(N, 48);      \
    else if (vec_ok && lay.L % 4 == 0) VF_GO2(N, 64);                \
    else VF_GO2(N, 0);                                               \
  } while (0)
  if (lay.D <= 12) VF_GO(12);
  else if (lay.D <= 25) VF_GO(25);
  else VF_GO(40);
#undef VF_GO
#undef VF_GO2
}

// ------------------------------------------------------------------------------------------
// fast decode support (ScrfDecodeOut, scrf_common.h)
// ------------------------------------------------------------------------------------------
// w1[o] = sum of |lambda| over label o's state block; one wavefront per label
__global__ void k_state_l1(const double* __restrict__ lambda, ScrfLayout lay, double* __restrict__ w1) {
  const uint32_t o = blockIdx.x;
  const double* wp = lambda + lay.state_idx(o);
  const uint32_t n = lay.nsfe + (lay.use_sb ? 1 : 0);
  double a = 0.0;
  for (uint32_t f = threadIdx.x; f < n; f += 64) a += fabs(wp[f]);
  for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
  if (threadIdx.x == 0) w1[o] = a * (1.0 + 1e-12);   // the summation error of the norm itself
}
void launch_state_l1(hipStream_t st, const double* lambda, const ScrfLayout& lay, double* w1) {
  hipLaunchKernelGGL(k_state_l1, dim3(lay.L), dim3(64), 0, st, lambda, lay, w1);
}

// One wavefront per listed (row, output).  The window vector of the row is rebuilt from the raw
// frames with k_windows' arithmetic -- lanes over the raw column: float running sum from the LAST
// frame backwards / length, running max / min, sampled frames; one-hot duration -- into LDS in
// feature order.  The contraction keeps the reference's order with unfused multiply and add
// (k_scores_exact): the products are formed in parallel (each is rounded on its own, so that is the
// same number) and parked in LDS, then added one after the other in feature order -- every lane
// runs the same chain over broadcast reads, which the compiler requests ahead of the adds; bias
// last.  The arc weight is float(-1 * score).
// Round 4: the utterance of a row is found by a 64-way search (each lane probes one offset, two
// rounds for 4096 utterances) instead of twelve dependent loads, and the chain reads its terms
// from LDS instead of two v_readlane per term: 1.26 -> see DESIGN 4.5 (212 k entries per 4096
// utterances of config 2).
#define FX_WAVES 4
#define FX_MAXF 1024
__device__ __forceinline__ uint32_t find_utt_wave(const uint64_t* __restrict__ off, uint32_t u0, uint32_t u1, uint64_t x,
                                                  uint32_t lane) {
  // largest u in [u0, u1) with off[u] <= x (off[u0] <= x < off[u1]); every lane gets the answer
  uint32_t lo = u0, n = u1 - u0;
  while (n > 1) {
    const uint32_t step = (n + 63) / 64;            // probes lo + i * step, i = 0 .. 63
    const uint32_t pu = lo + lane * step;
    const bool le = pu < lo + n && off[pu] <= x;
    const unsigned long long m = __ballot(le);      // a prefix of ones (off is non-decreasing), bit 0 set
    const uint32_t i = (uint32_t)__popcll(m) - 1;
    const uint32_t nlo = lo + i * step;
    const uint32_t nn = min(step, lo + n - nlo);
    lo = nlo;
    n = nn;
  }
  return lo;
}
__global__ __launch_bounds__(64 * FX_WAVES) void k_decode_fixup(const float* __restrict__ frames, uint32_t W, ScrfBatchView bv,
                                                                uint32_t u0, uint32_t u1, const double* __restrict__ lambda,
                                                                ScrfLayout lay, const uint32_t* __restrict__ cnt,
                                                                const uint64_t* __restrict__ list, uint32_t cap,
                                                                float* __restrict__ wneg) {
  extern __shared__ double pr_all[];   // [FX_WAVES][F rounded up to 2]: sized by the launch, so that narrow streams keep 8 workgroups per CU
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n = min(*cnt, cap), D = lay.D, L = lay.L, F = 8 * W + D;
  double* pr = pr_all + (size_t)wave * ((F + 1) & ~1u);
  const uint64_t tri = (uint64_t)D * (D + 1) / 2;
  for (uint32_t i = blockIdx.x * FX_WAVES + wave; i < n; i += gridDim.x * FX_WAVES) {
    const uint64_t row = list[i] >> 16;
    const uint32_t o = (uint32_t)(list[i] & 0xffff);
    const uint64_t arow = row + bv.seg_off[u0];
    const uint32_t u = find_utt_wave(bv.seg_off, u0, u1, arow, lane);
    const uint64_t r = arow - bv.seg_off[u];
    uint32_t t, d;
    if (r < tri) {
      t = 0;
      while ((uint64_t)(t + 1) * (t + 2) / 2 <= r) t++;
      d = (uint32_t)(r - (uint64_t)t * (t + 1) / 2) + 1;
    } else {
      t = D + (uint32_t)((r - tri) / D);
      d = (uint32_t)((r - tri) % D) + 1;
    }
    const float* last = frames + (bv.frame_off[u] + t) * (uint64_t)W;
    const float* first = last - (uint64_t)(d - 1) * W;
    const float ot = (float)((double)d * 0.1);
    const double* wp = lambda + lay.state_idx(o);
    for (uint32_t c = lane; c < W; c += 64) {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const float prod = ot * (float)(2 * k + 1);
        const uint32_t step = (uint32_t)ceilf(prod) - 1u;
        pr[k * W + c] = __dmul_rn((double)first[(uint64_t)step * W + c], wp[k * W + c]);
      }
      float acc = 0.0f, mx = last[c], mn = last[c];
      for (uint32_t w = 0; w < d; w++) {
        const float v = (last - (uint64_t)w * W)[c];
        acc = __fadd_rn(acc, v);
        if (v > mx) mx = v;
        if (v < mn) mn = v;
      }
      pr[5 * W + c] = __dmul_rn((double)__fdiv_rn(acc, (float)d), wp[5 * W + c]);
      pr[6 * W + c] = __dmul_rn((double)mx, wp[6 * W + c]);
      pr[7 * W + c] = __dmul_rn((double)mn, wp[7 * W + c]);
    }
    for (uint32_t k = lane; k < D; k += 64) pr[8 * W + k] = __dmul_rn((k + 1 == d) ? 1.0 : 0.0, wp[8 * W + k]);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wavefront's own LDS stores have landed
    double s = 0.0;
    uint32_t f = 0;
    for (; f + 8 <= F; f += 8) {
      const double2 a = *(const double2*)(pr + f), b = *(const double2*)(pr + f + 2);
      const double2 c2 = *(const double2*)(pr + f + 4), e = *(const double2*)(pr + f + 6);
      s = __dadd_rn(s, a.x); s = __dadd_rn(s, a.y); s = __dadd_rn(s, b.x); s = __dadd_rn(s, b.y);
      s = __dadd_rn(s, c2.x); s = __dadd_rn(s, c2.y); s = __dadd_rn(s, e.x); s = __dadd_rn(s, e.y);
    }
    for (; f < F; f++) s = __dadd_rn(s, pr[f]);
    if (lay.use_sb) s = __dadd_rn(s, __dmul_rn(wp[lay.nsfe], lay.sbv));
    if (lane == 0) wneg[row * L + o] = (float)(-1 * s);
    __builtin_amdgcn_wave_barrier();
  }
}
void launch_decode_fixup(hipStream_t st, const float* frames, uint32_t W, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                         const double* lambda, const ScrfLayout& lay, const uint32_t* cnt, const uint64_t* list,
                         uint32_t cap, float* wneg) {
  if (cap == 0) return;
  const uint32_t blocks = std::min<uint32_t>((cap + FX_WAVES - 1) / FX_WAVES, 4096);
  const uint32_t F = 8 * W + lay.D;   // <= FX_MAXF (fused_supported)
  hipLaunchKernelGGL(k_decode_fixup, dim3(blocks), dim3(64 * FX_WAVES), sizeof(double) * FX_WAVES * ((F + 1) & ~1u), st, frames, W, bv, u0, u1,
                     lambda, lay, cnt, list, cap, wneg);
}

// ------------------------------------------------------------------------------------------
// k_arcs_seg / k_arcs_frame: one block per node (+1 for the final arcs), arcs written at the
// index the reference's AddArc call order gives them.
// ------------------------------------------------------------------------------------------
__global__ void k_arcs_seg(ScrfLayout lay, uint32_t T, const double* __restrict__ S,
                           const double* __restrict__ M, int m_per_frame, float final_w,
                           scrf_arc* __restrict__ arcs) {
  const uint32_t L = lay.L, D = lay.D;
  const uint32_t t = blockIdx.x;
  const size_t LL = (size_t)L * L;
  if (t == T) {  // final arcs (:366-399)
    const uint64_t ab = scrf_arc_base(T, L, D);
    const int32_t fin = scrf_node_start_state(T, L);
    for (uint32_t p = threadIdx.x; p < L; p += blockDim.x) {
      int32_t src = (T == 1) ? scrf_node_start_state(0, L) + (int32_t)p
                             : scrf_node_start_state(T - 1, L) + (int32_t)L + (int32_t)p;
      scrf_arc a = {src, 0, 0, final_w, fin};
      arcs[ab + p] = a;
    }
    return;
  }
  const uint64_t ab = scrf_arc_base(t, L, D);
  const uint32_t np = scrf_num_prev(t, D), nd = scrf_node_max_dur(t, D);
  const uint64_t base = scrf_seg_base(t, D);
  const int32_t nss = scrf_node_start_state(t, L);
  uint64_t eb = ab;
  if (np > 0) {  // boundary arcs (:260-296): for lab, for prev_lab
    const double* Mt = M + (m_per_frame ? (size_t)t * LL : 0);
    const int32_t pbase = (t == 1) ? scrf_node_start_state(0, L) : scrf_node_start_state(t - 1, L) + (int32_t)L;
    for (uint32_t idx = threadIdx.x; idx < L * L; idx += blockDim.x) {
      uint32_t lab = idx / L, pl = idx % L;
      scrf_arc a = {pbase + (int32_t)pl, 0, 0, (float)(-1.0 * Mt[(size_t)pl * L + lab]), nss + (int32_t)lab};
      arcs[ab + idx] = a;
    }
    eb += LL;
  }
  const int32_t ebase = (t == 0) ? nss : nss + (int32_t)L;  // end states of node t
  for (uint32_t idx = threadIdx.x; idx < L * nd; idx += blockDim.x) {  // (:300-325): for lab, for dur
    uint32_t lab = idx / nd, d = idx % nd + 1;
    int32_t src = (d <= np) ? scrf_node_start_state(t - d + 1, L) + (int32_t)lab : 0;
    int32_t lb = (int32_t)(lab + L * (d - 1) + 1);
    scrf_arc a = {src, lb, lb, (float)(-1.0 * S[(base + d - 1) * L + lab]), ebase + (int32_t)lab};
    arcs[eb + idx] = a;
  }
}

__global__ void k_arcs_frame(ScrfLayout lay, uint32_t T, const double* __restrict__ S,
                             const double* __restrict__ M, int m_per_frame, float final_w,
                             scrf_arc* __restrict__ arcs) {
  const uint32_t L = lay.L;
  const uint32_t t = blockIdx.x;
  const size_t LL = (size_t)L * L;
  if (t == T) {
    const uint64_t ab = (uint64_t)L + (uint64_t)(T - 1) * LL;
    const int32_t fin = (int32_t)(L * T + 1);
    for (uint32_t p = threadIdx.x; p < L; p += blockDim.x) {
      scrf_arc a = {(int32_t)(L * (T - 1) + p + 1), 0, 0, final_w, fin};
      arcs[ab + p] = a;
    }
    return;
  }
  if (t == 0) {
    for (uint32_t c = threadIdx.x; c < L; c += blockDim.x) {
      scrf_arc a = {0, (int32_t)c + 1, (int32_t)c + 1, (float)(-1.0 * S[c]), (int32_t)c + 1};
      arcs[c] = a;
    }
    return;
  }
  const uint64_t ab = (uint64_t)L + (uint64_t)(t - 1) * LL;
  const double* Mt = M + (m_per_frame ? (size_t)t * LL : 0);
  for (uint32_t idx = threadIdx.x; idx < L * L; idx += blockDim.x) {
    uint32_t c = idx / L, p = idx % L;
    float w = (float)(-1.0 * (Mt[(size_t)p * L + c] + S[(size_t)t * L + c]));
    scrf_arc a = {(int32_t)(L * (t - 1) + p + 1), (int32_t)c + 1, (int32_t)c + 1, w, (int32_t)(L * t + c + 1)};
    arcs[ab + idx] = a;
  }
}

void launch_arcs(hipStream_t st, const ScrfLayout& lay, uint32_t T, int frame_model, const double* S,
                 const double* M, int m_per_frame, float final_w, scrf_arc* arcs) {
  if (T == 0) return;
  if (frame_model)
    hipLaunchKernelGGL(k_arcs_frame, dim3(T + 1), dim3(256), 0, st, lay, T, S, M, m_per_frame, final_w, arcs);
  else
    hipLaunchKernelGGL(k_arcs_seg, dim3(T + 1), dim3(256), 0, st, lay, T, S, M, m_per_frame, final_w, arcs);
}

// ------------------------------------------------------------------------------------------
// optimizer and vector utilities
// ------------------------------------------------------------------------------------------
__global__ void k_sgd_step(double* __restrict__ lambda, double* __restrict__ lambda_acc,
                           double* __restrict__ gsa, double* __restrict__ grad, uint32_t n, double lr,
                           int adagrad, double eps) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double g = grad[i], lam = lambda[i];
  if (adagrad) {  // :314-315
    double a = __dadd_rn(gsa[i], __dmul_rn(g, g));
    gsa[i] = a;
    lam = __dadd_rn(lam, __dmul_rn(__ddiv_rn(lr, __dadd_rn(__dsqrt_rn(a), eps)), g));
  } else {  // :317
    lam = __dadd_rn(lam, __dmul_rn(lr, g));
  }
  lambda[i] = lam;
  lambda_acc[i] = __dadd_rn(lambda_acc[i], lam);  // :319
  grad[i] = 0.0;                                  // :321
}
void launch_sgd_step(hipStream_t st, double* lambda, double* lambda_acc, double* gsa, double* grad, uint32_t n,
                     double lr, int adagrad, double eps) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_sgd_step, dim3((n + 255) / 256), dim3(256), 0, st, lambda, lambda_acc, gsa, grad, n, lr,
                     adagrad, eps);
}

__global__ void k_scale(double* __restrict__ v, uint32_t n, double s, int divide) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = divide ? __ddiv_rn(v[i], s) : __dmul_rn(v[i], s);
}
void launch_scale(hipStream_t st, double* v, uint32_t n, double s, int divide) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_scale, dim3((n + 255) / 256), dim3(256), 0, st, v, n, s, divide);
}

__global__ void k_axpy(double* __restrict__ y, const double* __restrict__ x, uint32_t n) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}
void launch_add(hipStream_t st, double* y, const double* x, uint32_t n) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_axpy, dim3((n + 255) / 256), dim3(256), 0, st, y, x, n);
}
