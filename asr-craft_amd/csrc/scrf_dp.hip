// scrf_dp.hip -- wavefront-per-utterance forward/backward for L <= 64 labels (gfx950, wave64).
//
// Design (DESIGN.md "DP kernels"):
//  * one 64-lane wavefront owns one utterance, lane = label; no workgroup barriers in the
//    time loop, all cross-label traffic is v_readlane broadcasts and wave reductions;
//  * the L x L transition step runs in the linear domain:
//        aPT[n] = amax + shift + log( sum_c exp(alpha[c]-amax) * E[c][n] ),  E = exp(M - shift)
//    i.e. L exps + L*L FMAs per frame instead of L*L exps (mathematically the reference's
//    logAdd over c; differs in rounding only, ~1e-16 relative);
//  * the duration step stays in the log domain: alpha[l] = LSE_d(aPT[t-d][l] + S[t][d][l]);
//  * forward and backward are independent given the scores, so one launch runs them as two
//    halves of the grid (blockIdx parity), doubling the wavefronts in flight;
//  * posteriors (gamma, xi) need no recursion: they are separate fully parallel kernels.
// Reference arithmetic being reproduced: nodes/CRF_StdSegStateNode_WithoutDurLab_WithoutSegTransFtr.cpp
//   computeAlpha :123-245, computeAlphaPlusTrans :1077-1108, computeBeta :395-466, computeExpF :616-949.
#include "scrf_dp_common.h"

// E = exp(M - max(M)), its transpose, and the shift; one workgroup per L x L matrix
__global__ void k_exp_m(const double* __restrict__ M, uint32_t L, double* __restrict__ E,
                        double* __restrict__ ET, double* __restrict__ mshift) {
  __shared__ double red[256];
  const uint32_t LL = L * L;
  const double* Mb = M + (size_t)blockIdx.x * LL;
  double mx = -INFINITY;
  for (uint32_t i = threadIdx.x; i < LL; i += blockDim.x) mx = fmax(mx, Mb[i]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  mx = red[0];
  for (uint32_t i = threadIdx.x; i < LL; i += blockDim.x) {
    double e = exp(Mb[i] - mx);
    E[(size_t)blockIdx.x * LL + i] = e;
    ET[(size_t)blockIdx.x * LL + (size_t)(i % L) * L + i / L] = e;
  }
  if (threadIdx.x == 0) mshift[blockIdx.x] = mx;
}

// the same for L <= 64 (one matrix per frame with per-frame transition features: the TIMIT demo has 77 824 of them per
// 256-utterance step): the matrix is read ONCE into registers (the maximum needs every element before the first exp), the
// transpose goes through an LDS tile with an odd row stride so that both E and ET leave in coalesced rows (the scattered
// 8-byte stores of the general kernel: 1.76 ms at that size), the maximum by DPP shuffles + one LDS exchange between the wavefronts.
// Same values as k_exp_m: fmax is exact in any order, every element is exp(M - max) by the same routine.
template <int NE>   // elements per thread: ceil(L * L / 256)
__global__ __launch_bounds__(256) void k_exp_m_tile(const double* __restrict__ M, uint32_t L, double* __restrict__ E,
                                                    double* __restrict__ ET, double* __restrict__ mshift) {
  __shared__ double tile[64 * 65];
  __shared__ double wmax[4];
  const uint32_t LL = L * L, tid = threadIdx.x, ts = L + 1 - (L & 1);   // odd stride >= L
  const double* Mb = M + (size_t)blockIdx.x * LL;
  double v[NE];
  double mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    const uint32_t i = tid + k * 256;
    v[k] = i < LL ? __builtin_nontemporal_load(Mb + i) : -INFINITY;
    mx = fmax(mx, v[k]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
  if ((tid & 63) == 0) wmax[tid >> 6] = mx;
  __syncthreads();
  mx = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
  double* Eb = E + (size_t)blockIdx.x * LL;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    const uint32_t i = tid + k * 256;
    if (i < LL) {
      const double e = exp(v[k] - mx);
      Eb[i] = e;
      tile[(i / L) * ts + i % L] = e;
    }
  }
  __syncthreads();
  double* Tb = ET + (size_t)blockIdx.x * LL;
#pragma unroll
  for (int k = 0; k < NE; k++) {
    const uint32_t i = tid + k * 256;
    if (i < LL) Tb[i] = tile[(i % L) * ts + i / L];   // ET[r][c] = E[c][r]
  }
  if (tid == 0) mshift[blockIdx.x] = mx;
}

void launch_exp_m(hipStream_t st, const double* M, uint32_t L, uint64_t n_mat, double* E, double* ET,
                  double* mshift) {
  if (n_mat == 0) return;
  static const bool tile_off = getenv("SCRF_EXPM_TILE") && atoi(getenv("SCRF_EXPM_TILE")) == 0;   // A/B knob
  const uint32_t ne = (L * L + 255) / 256;
  if (L <= 64 && !tile_off) {
    if (ne <= 4) hipLaunchKernelGGL(k_exp_m_tile<4>, dim3((uint32_t)n_mat), dim3(256), 0, st, M, L, E, ET, mshift);
    else if (ne <= 9) hipLaunchKernelGGL(k_exp_m_tile<9>, dim3((uint32_t)n_mat), dim3(256), 0, st, M, L, E, ET, mshift);
    else hipLaunchKernelGGL(k_exp_m_tile<16>, dim3((uint32_t)n_mat), dim3(256), 0, st, M, L, E, ET, mshift);
    return;
  }
  hipLaunchKernelGGL(k_exp_m, dim3((uint32_t)n_mat), dim3(256), 0, st, M, L, E, ET, mshift);
}

template <int DMAX, int MPF>
__global__ __launch_bounds__(DP_WPB * 64, 3) void k_dp_wave(
    ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ S,
    const double* __restrict__ E, const double* __restrict__ ET, const double* __restrict__ mshift,
    double* __restrict__ AD, double* __restrict__ alpha_g, double* __restrict__ beta_g,
    double* __restrict__ sd_g, double* __restrict__ zx_out, int* __restrict__ status) {
  extern __shared__ double dsm[];
  const int L = lay.L, D = lay.D;
  const int LL = L * L;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int dir = blockIdx.x & 1;  // 0 forward, 1 backward
  const uint32_t ul = (blockIdx.x >> 1) * (blockDim.x >> 6) + wave;   // blockDim.x / 64 utterances per workgroup
  double* Es = dsm;                                         // [L*L] (time-invariant transitions only)
  double* ring = dsm + (MPF ? 0 : LL) + (size_t)wave * D * L;  // [D][L], private to this wavefront
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
  const double* Su = S + s_base * L;
  const bool act = lane < L;
  const int lc = act ? lane : L - 1;  // clamped label: idle lanes shadow the last label
  const double sh0 = MPF ? 0.0 : mshift[0];
  int err = 0;

  if (dir == 0) {
    // ---------------------------------------------------------------- forward
    double* ADu = AD + s_base * L;
    double* alu = alpha_g + f_base * L;
    double alpha = Su[lc];
    if (act) { ADu[lane] = alpha; alu[lane] = alpha; }
    int rpos = D - 1;
    for (int t = 1; t < T; t++) {
      rpos = (rpos + 1 == D) ? 0 : rpos + 1;  // ring slot of node t-1
      const int np = (int)scrf_num_prev(t, D), nd = (int)scrf_node_max_dur(t, D);
      const uint64_t base = scrf_seg_base(t, D);
      const bool full = (nd == DMAX) && (np == DMAX);  // steady state: every duration has a predecessor
      // scores of the nd windows ending at t: independent of the recursion, issued first
      double sv[DMAX];
      if (full) {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) sv[d0] = Su[(base + d0) * L + lc];
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) sv[d0] = Su[(base + (d0 < nd ? d0 : 0)) * L + lc];
      }
      const double amax = (double)wave_max_f32((float)alpha);
      const double a = exp_nonpos(alpha - amax);
      double usum;
      double sh = sh0;
      if (MPF) {
        usum = matvec_bcast(a, E + (f_base + t) * (size_t)LL, L, lc);
        sh = mshift[f_base + t];
      } else {
        usum = matvec_bcast(a, Es, L, lc);
      }
      if (!(usum > 0.0 && usum < INFINITY)) err = 1;
      const double apt = amax + sh + log(usum);
      ring[rpos * L + lc] = apt;  // idle lanes rewrite lane L-1's value with the same number
      double v[DMAX];
      double m = -INFINITY;
      if (full) {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = rpos - d0;
          slot += (slot < 0) ? D : 0;
          const double x = ((d0 == 0) ? apt : ring[slot * L + lc]) + sv[d0];
          v[d0] = x;
          m = fmax(m, x);
        }
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = rpos - d0;
          if (slot < 0) slot += D;
          const double r = (d0 == 0) ? apt : ring[(d0 < np ? slot : rpos) * L + lc];
          double x = (d0 < np) ? r + sv[d0] : sv[d0];
          x = (d0 < nd) ? x : -INFINITY;
          v[d0] = x;
          m = fmax(m, x);
        }
      }
      double ssum = 0.0;
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) ssum += exp_nonpos(v[d0] - m);
      alpha = m + log(ssum);
      if (act) {
        if (full) {
#pragma unroll
          for (int d0 = 0; d0 < DMAX; d0++) ADu[(base + d0) * L + lane] = v[d0];
        } else {
#pragma unroll
          for (int d0 = 0; d0 < DMAX; d0++)
            if (d0 < nd) ADu[(base + d0) * L + lane] = v[d0];
        }
        alu[(size_t)t * L + lane] = alpha;
      }
    }
    // Zx = LSE_l alpha[T-1][l]  (computeAlphaSum)
    const double mx = (double)wave_max_f32((float)alpha);
    const double tot = wave_sum_f64(act ? exp_nonpos(alpha - mx) : 0.0);
    const double Zx = mx + log(tot);
    if (!(Zx == Zx) || isinf(Zx)) err = 1;
    if (lane == 0) zx_out[u] = Zx;
  } else {
    // ---------------------------------------------------------------- backward
    double* beu = beta_g + f_base * L;
    double* sdu = sd_g + f_base * L;
    int tpos = (T - 1) % D;
    ring[tpos * L + lc] = 0.0;  // setTailBeta
    if (act) { beu[(size_t)(T - 1) * L + lane] = 0.0; sdu[(size_t)(T - 1) * L + lane] = 0.0; }
    for (int t = T - 2; t >= 0; t--) {
      const int nn = (T - 1 - t <= D) ? T - 1 - t : D;
      tpos = (tpos == 0) ? D - 1 : tpos - 1;  // ring slot of node t
      double v[DMAX];
      double m = -INFINITY;
      uint64_t sb = scrf_seg_base(t + 1, D);
      if (nn == DMAX && t + 1 >= D) {
        // steady state: all D successors exist and every node t+1.. carries D windows, so the
        // window (t+d0+1, d0+1) sits at row seg_base(t+1) + d0*(D+1)
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          int slot = tpos + d0 + 1;
          slot -= (slot >= D) ? D : 0;
          const double x = Su[(sb + (uint64_t)d0 * (DMAX + 1)) * L + lc] + ring[slot * L + lc];
          v[d0] = x;
          m = fmax(m, x);
        }
      } else {
#pragma unroll
        for (int d0 = 0; d0 < DMAX; d0++) {
          const bool ok = d0 < nn;
          int slot = tpos + d0 + 1;  // node t + d0 + 1
          if (slot >= D) slot -= D;
          const double sc = Su[(ok ? sb + d0 : 0) * L + lc];
          const double bt = ring[(ok ? slot : tpos) * L + lc];
          const double x = ok ? sc + bt : -INFINITY;
          v[d0] = x;
          m = fmax(m, x);
          sb += scrf_node_max_dur(t + 1 + d0, D);
        }
      }
      double ssum = 0.0;
#pragma unroll
      for (int d0 = 0; d0 < DMAX; d0++) ssum += exp_nonpos(v[d0] - m);
      const double sd = m + log(ssum);
      const double smax = (double)wave_max_f32((float)sd);
      const double b = exp_nonpos(sd - smax);
      double w;
      double sh = sh0;
      if (MPF) {
        w = matvec_bcast(b, ET + (f_base + t + 1) * (size_t)LL, L, lc);
        sh = mshift[f_base + t + 1];
      } else {
        w = matvec_bcast(b, Es, L, lc);
      }
      if (!(w > 0.0 && w < INFINITY)) err = 1;
      const double beta = smax + sh + log(w);
      ring[tpos * L + lc] = beta;
      if (act) { sdu[(size_t)t * L + lane] = sd; beu[(size_t)t * L + lane] = beta; }
    }
  }
  if (__any(err != 0) && lane == 0) atomicMax(&status[u], SCRF_ERR_NUMERIC);
}

int dp_wave_supported(const ScrfLayout& lay) { return lay.L <= 64 && lay.D <= 32; }

void launch_dp_wave(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                    const double* S, const double* E, const double* ET, const double* mshift, int m_per_frame,
                    double* AD, double* alpha_g, double* beta_g, double* sd_g, double* zx, int* status) {
  if (n_utts == 0) return;
  const uint32_t wpb = dp_waves_per_block(sizeof(double) * (m_per_frame ? 0 : (size_t)lay.L * lay.L), sizeof(double) * (size_t)lay.D * lay.L);
  const uint32_t nblk = 2 * ((n_utts + wpb - 1) / wpb);
  const size_t sm = sizeof(double) * ((m_per_frame ? 0 : (size_t)lay.L * lay.L) + (size_t)wpb * lay.D * lay.L);
#define DP_LAUNCH2(DM, MPF)                                                                                   \
  do {                                                                                                          \
    hipFuncSetAttribute((const void*)k_dp_wave<DM, MPF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);  \
    hipLaunchKernelGGL((k_dp_wave<DM, MPF>), dim3(nblk), dim3(wpb * 64), sm, st, lay, bv, u0, n_utts, S, E, ET, \
                       mshift, AD, alpha_g, beta_g, sd_g, zx, status);                                          \
  } while (0)
#define DP_LAUNCH(DM)                     \
  do {                                    \
    if (m_per_frame) DP_LAUNCH2(DM, 1);   \
    else DP_LAUNCH2(DM, 0);               \
  } while (0)
  if (lay.D <= 1) DP_LAUNCH(1);
  else if (lay.D <= 4) DP_LAUNCH(4);
  else if (lay.D <= 10) DP_LAUNCH(10);
  else if (lay.D <= 16) DP_LAUNCH(16);
  else if (lay.D <= 25) DP_LAUNCH(25);
  else DP_LAUNCH(32);
#undef DP_LAUNCH
#undef DP_LAUNCH2
}

// ------------------------------------------------------------------------------------------
// k_post_state: R = Y - gamma over ad (in place), gamma = exp(ad + beta - Zx)    (:673-702)
// plus the per-frame numerator term (true state score + true transition score).
// One workgroup per frame of the chunk.
// ------------------------------------------------------------------------------------------
__global__ void k_post_state(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                             const uint32_t* __restrict__ next_lab, const double* __restrict__ S,
                             const double* __restrict__ M, int m_per_frame, double* __restrict__ AD,
                             const double* __restrict__ beta_g, const double* __restrict__ zx,
                             double* __restrict__ numer_f, int* __restrict__ status, double* __restrict__ mass_s) {
  __shared__ double msum[4];
  const uint32_t L = lay.L, D = lay.D;
  const uint64_t fi = blockIdx.x;  // frame index inside the chunk
  const uint64_t gf = bv.frame_off[u0] + fi;
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (bv.frame_off[mid] <= gf) lo = mid; else hi = mid; }
  const uint32_t u = lo;
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t T = bv.T[u];
  const uint64_t row0 = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const uint32_t nd = scrf_node_max_dur(t, D);
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
  const double* bt = beta_g + fi * L;
  double gs = 0.0;
  for (uint32_t idx = threadIdx.x; idx < nd * L; idx += blockDim.x) {
    const uint32_t d0 = idx / L, l = idx - d0 * L;
    double a = AD[(row0 + d0) * L + l] + bt[l] - Zx;
    if (a >= LN_MAX) err = SCRF_ERR_NUMERIC;
    double y = (l == al && d0 + 1 == ld) ? 1.0 : 0.0;
    const double g = exp(a);
    AD[(row0 + d0) * L + l] = y - g;
    gs += g;
  }
  gs = wave_sum_f64(gs);
  if ((threadIdx.x & 63) == 0) msum[threadIdx.x >> 6] = gs;
  __syncthreads();
  if (threadIdx.x == 0) {
    mass_s[fi] = (msum[0] + msum[1]) + (msum[2] + msum[3]);
    double nodeLi = 0.0;
    if (lab != SCRF_LAB_BAD && err == 0) {
      if (ld <= nd) nodeLi += S[(row0 + ld - 1) * L + al];
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

void launch_post_state(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                       uint64_t n_frames, const uint32_t* next_lab, const double* S, const double* M,
                       int m_per_frame, double* AD, const double* beta_g, const double* zx, double* numer_f,
                       int* status, double* mass_s) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_post_state, dim3((uint32_t)n_frames), dim3(256), 0, st, lay, bv, u0, u1, next_lab, S, M,
                     m_per_frame, AD, beta_g, zx, numer_f, status, mass_s);
}

// numer[u] = sum of the per-frame terms in the reference's order (t = T-1 .. 0, :388-469)
__global__ void k_numer_reduce(ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* __restrict__ numer_f,
                               double* __restrict__ numer) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_utts) return;
  const uint32_t u = u0 + i;
  const uint64_t fb = bv.frame_off[u] - bv.frame_off[u0];
  double s = 0.0;
  for (uint32_t t = bv.T[u]; t-- > 0;) s += numer_f[fb + t];
  numer[u] = s;
}
void launch_numer_reduce(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, const double* numer_f,
                         double* numer) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_numer_reduce, dim3((n_utts + 63) / 64), dim3(64), 0, st, bv, u0, n_utts, numer_f, numer);
}

// ------------------------------------------------------------------------------------------
// transition posteriors xi[t][c][n] = exp(alpha[t][c] + M[t+1][c][n] + sd[t][n] - Zx) (:773-776)
// factorised as A[t][c] * exp(M[t+1][c][n]) * B[t][n],  A = exp(alpha - m_t), B = exp(sd + m_t - Zx).
// k_xi_factors: one wavefront per frame.
// ------------------------------------------------------------------------------------------
__global__ void k_xi_factors(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t u1, uint64_t n_frames,
                             const double* __restrict__ alpha_g, const double* __restrict__ sd_g,
                             const double* __restrict__ zx, double* __restrict__ A, double* __restrict__ B) {
  const uint32_t L = lay.L;
  const uint64_t fi = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const uint32_t lane = threadIdx.x & 63;
  if (fi >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + fi;
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (bv.frame_off[mid] <= gf) lo = mid; else hi = mid; }
  const uint32_t u = lo;
  const bool last = (gf + 1 == bv.frame_off[u + 1]);
  float mx = -INFINITY;
  for (uint32_t l = lane; l < L; l += 64) mx = fmaxf(mx, (float)alpha_g[fi * L + l]);
  const double m = (double)wave_max_f32(mx);
  const double Zx = zx[u];
  for (uint32_t l = lane; l < L; l += 64) {
    A[fi * L + l] = exp(alpha_g[fi * L + l] - m);
    B[fi * L + l] = last ? 0.0 : exp(sd_g[fi * L + l] + m - Zx);
  }
}
void launch_xi_factors(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                       uint64_t n_frames, const double* alpha_g, const double* sd_g, const double* zx, double* A,
                       double* B) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_xi_factors, dim3((uint32_t)((n_frames + 3) / 4)), dim3(256), 0, st, lay, bv, u0, u1,
                     n_frames, alpha_g, sd_g, zx, A, B);
}

// bias-only transitions: C[z][c][n] = sum_{frames of K-chunk z} A[f][c] * B[f][n]  (A^T B, K = frames)
// on the fp64 MFMA: one wavefront per K-chunk holds the LT x LT output tiles in registers and
// streams its frames four at a time straight from memory (each operand fragment is four 128-byte
// row pieces); the next group's fragments are in flight under the MFMAs.
typedef double atb_v4f64 __attribute__((ext_vector_type(4)));
template <int LT>
__global__ __launch_bounds__(256) void k_atb(const double* __restrict__ A, const double* __restrict__ B, uint32_t L,
                                             uint64_t n_frames, uint64_t rows_per_chunk, uint32_t n_chunks,
                                             double* __restrict__ slab, const double* __restrict__ bscale) {
  const uint32_t lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const uint32_t z = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (z >= n_chunks) return;
  // L > 64: the output is cut into 64 x 64 blocks, blockIdx.y = (row block, column block)
  const uint32_t nb = (L + 63) / 64, c0 = (blockIdx.y / nb) * 64, n0 = (blockIdx.y % nb) * 64;
  const uint64_t r_begin = (uint64_t)z * rows_per_chunk;
  const uint64_t r_end = min(n_frames, r_begin + rows_per_chunk);
  atb_v4f64 acc[LT][LT];
#pragma unroll
  for (int i = 0; i < LT; i++)
#pragma unroll
    for (int j = 0; j < LT; j++) acc[i][j] = (atb_v4f64){0.0, 0.0, 0.0, 0.0};
  double a_n[LT], b_n[LT];
  auto load = [&](uint64_t f0) {
    const uint64_t f = f0 + lk;
    const double sc = bscale ? (f < r_end ? bscale[f] : 0.0) : 1.0;   // per-frame factor of the B rows (k_xi_scale)
#pragma unroll
    for (int i = 0; i < LT; i++) {
      const uint32_t c = c0 + i * 16 + li, n = n0 + i * 16 + li;
      a_n[i] = (f < r_end && c < L) ? A[f * L + c] : 0.0;
      const double bv_ = (f < r_end && n < L) ? B[f * L + n] : 0.0;
      b_n[i] = bscale ? bv_ * sc : bv_;
    }
  };
  load(r_begin);
  for (uint64_t f0 = r_begin; f0 < r_end; f0 += 4) {
    double a[LT], b[LT];
#pragma unroll
    for (int i = 0; i < LT; i++) { a[i] = a_n[i]; b[i] = b_n[i]; }
    if (f0 + 4 < r_end) load(f0 + 4);
#pragma unroll
    for (int i = 0; i < LT; i++)
#pragma unroll
      for (int j = 0; j < LT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double* out = slab + (size_t)z * L * L;
#pragma unroll
  for (int i = 0; i < LT; i++)
#pragma unroll
    for (int j = 0; j < LT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t c = c0 + i * 16 + lk + 4 * r, n = n0 + j * 16 + li;
        if (c < L && n < L) out[(size_t)c * L + n] = acc[i][j][r];
      }
}
// grad[trans_idx(c,n)] -= tbv * exp(M0[c][n]) * sum_z C[z][c][n]   (expected transition-bias counts)
// fixed association: 16 contiguous z groups, ascending inside, group sums added in group order
#define RA_G 16
__global__ __launch_bounds__(64 * RA_G) void k_reduce_atb(const double* __restrict__ slab, uint32_t n_chunks,
                                                          const double* __restrict__ M0, ScrfLayout lay,
                                                          double* __restrict__ grad) {
  __shared__ double part[RA_G][64];
  const uint32_t LL = lay.L * lay.L;
  const uint32_t tx = threadIdx.x & 63, g = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 64 + tx;
  const uint32_t per = (n_chunks + RA_G - 1) / RA_G, z0 = g * per, z1 = min(n_chunks, z0 + per);
  double s = 0.0;
  if (i < LL)
    for (uint32_t z = z0; z < z1; z++) s += slab[(size_t)z * LL + i];
  part[g][tx] = s;
  __syncthreads();
  if (g == 0 && i < LL) {
    double t = part[0][tx];
#pragma unroll
    for (int k = 1; k < RA_G; k++) t += part[k][tx];
    grad[lay.trans_idx(i / lay.L, i % lay.L) + lay.ntfe] -= lay.tbv * exp(M0[i]) * t;
  }
}
int atb_supported(const ScrfLayout& lay) { return lay.L <= 256; }
void launch_atb(hipStream_t st, const ScrfLayout& lay, const double* A, const double* B, uint64_t n_frames,
                uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, const double* M0, double* grad,
                const double* bscale) {
  if (n_frames == 0 || n_chunks == 0 || !lay.use_tb) return;
  const uint32_t nb = (lay.L + 63) / 64;
  const dim3 grid((n_chunks + 3) / 4, nb * nb);
  const uint32_t lt = lay.L > 64 ? 4 : (lay.L + 15) / 16;
#define ATB_GO(N) hipLaunchKernelGGL(k_atb<N>, grid, dim3(256), 0, st, A, B, lay.L, n_frames, rows_per_chunk, n_chunks, slab, bscale)
  if (lt <= 1) ATB_GO(1);
  else if (lt == 2) ATB_GO(2);
  else if (lt == 3) ATB_GO(3);
  else ATB_GO(4);
#undef ATB_GO
  const uint32_t LL = lay.L * lay.L;
  hipLaunchKernelGGL(k_reduce_atb, dim3((LL + 63) / 64), dim3(64 * RA_G), 0, st, slab, n_chunks, M0, lay, grad);
}
// observed transition-bias counts of the whole batch (integers, precomputed at batch creation)
__global__ void k_add_trans_counts(const uint32_t* __restrict__ counts, ScrfLayout lay, double* __restrict__ grad) {
  const uint32_t LL = lay.L * lay.L;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= LL) return;
  grad[lay.trans_idx(i / lay.L, i % lay.L) + lay.ntfe] += lay.tbv * (double)counts[i];
}
void launch_add_trans_counts(hipStream_t st, const uint32_t* counts, const ScrfLayout& lay, double* grad) {
  if (!lay.use_tb) return;
  const uint32_t LL = lay.L * lay.L;
  hipLaunchKernelGGL(k_add_trans_counts, dim3((LL + 255) / 256), dim3(256), 0, st, counts, lay, grad);
}

// transition features: XI[f][c][n] = y - A[f][c] * E[f+1][c][n] * exp(shift[f+1]) * B[f][n]
__global__ void k_xi_full(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                          const uint32_t* __restrict__ next_lab, const double* __restrict__ A,
                          const double* __restrict__ B, const double* __restrict__ E,
                          const double* __restrict__ mshift, double* __restrict__ XI) {
  const uint32_t L = lay.L, D = lay.D, LL = L * L;
  const uint64_t fi = blockIdx.x;
  const uint64_t gf = bv.frame_off[u0] + fi;
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (bv.frame_off[mid] <= gf) lo = mid; else hi = mid; }
  const uint32_t u = lo;
  const bool last = (gf + 1 == bv.frame_off[u + 1]);
  double* out = XI + fi * (size_t)LL;
  if (last) {
    for (uint32_t i = threadIdx.x; i < LL; i += blockDim.x) out[i] = 0.0;
    return;
  }
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
  const uint32_t nl = next_lab[gf];
  const uint32_t al = (lab != SCRF_LAB_BAD && lab < L * D) ? lab % L : SCRF_LAB_BAD;
  const uint32_t anl = (nl != SCRF_LAB_BAD && nl < L * D) ? nl % L : SCRF_LAB_BAD;
  const double es = exp(mshift[fi + 1]);
  const double* En = E + (fi + 1) * (size_t)LL;
  for (uint32_t i = threadIdx.x; i < LL; i += blockDim.x) {
    const uint32_t c = i / L, n = i - c * L;
    const double y = (c == al && n == anl) ? 1.0 : 0.0;
    out[i] = y - A[fi * L + c] * En[i] * es * B[fi * L + n];
  }
}
void launch_xi_full(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                    uint64_t n_frames, const uint32_t* next_lab, const double* A, const double* B, const double* E,
                    const double* mshift, double* XI) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_xi_full, dim3((uint32_t)n_frames), dim3(256), 0, st, lay, bv, u0, u1, next_lab, A, B, E,
                     mshift, XI);
}
