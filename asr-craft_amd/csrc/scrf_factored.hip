// scrf_factored.hip -- recipe-factorised contractions (train_precision = SCRF_PREC_FACTORED).
//
// The standard segment recipe (io/CRF_InFtrStream_SeqMultiWindow.cpp:430-439) builds each window
// vector as [5 sampled frames | avg | max | min | one-hot dur].  The first 6W columns are LINEAR
// in the raw frames (a sample is a copy of frame pos_k(t,d); the average is a mean over the
// window), so for them
//     S_lin[t][d][l] = sum_k P[pos_k(t,d)][k][l] + (1/d) sum_{t' in window} P[t'][5][l],
//     P[f][k][l]     = sum_j frame[f][j] * lambda[state(l) + k*W + j]          (one small GEMM per frame)
// and the expected-count gradient is the adjoint: Z = C^T R gathered per frame, then
//     grad[state(l) + k*W + j] += sum_f Z[f][k][l] * frame[f][j].
// Only the columns [6W, 8W+D] (max, min, one-hot dur) + bias -- 31 % of the vector at config 2 --
// go through the dense MFMA kernels on a narrow window image X_mm.  Same mathematics as the
// reference; the average is formed from fp64 per-frame projections instead of the reference's
// float running sum, so values differ by ~1e-7 relative (training contract: 1e-4).
#include "scrf_kernels.h"

#include <math.h>

__device__ __forceinline__ uint32_t sample_step(uint32_t d, int k) {
  // (QNUInt32)ceil(one_tenth_win_len * i) - 1, i = 2k+1   (io/CRF_InFtrStream_SeqMultiWindow.cpp:566-569)
  const float ot = (float)((double)d * 0.1);
  return (uint32_t)ceilf(ot * (float)(2 * k + 1)) - 1u;
}

// XCD-aware work mapping (cdna_hip_programming.md T1): workgroups are dealt round-robin over
// the 8 XCDs, so group b handles work item (b % 8) * ceil(n/8) + b / 8: every XCD walks ONE
// contiguous range of frames, and the sliding window of rows its concurrent workgroups gather
// from (a few utterances) stays inside that XCD's 4 MiB L2.  Speed only, never correctness.
__device__ __forceinline__ uint64_t xcd_remap(uint32_t b, uint64_t n_items) {
  const uint64_t per = (n_items + 7) / 8;
  return (uint64_t)(b & 7) * per + (b >> 3);
}

__device__ __forceinline__ uint32_t find_utt_f(const uint64_t* off, uint32_t u0, uint32_t u1, uint64_t x) {
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if (off[mid] <= x) lo = mid; else hi = mid;
  }
  return lo;
}
// slot index (T+1 slots per utterance) -> utterance
__device__ __forceinline__ uint32_t find_utt_slot(const uint64_t* off, uint32_t u0, uint32_t u1, uint64_t si) {
  uint32_t lo = u0, hi = u1;
  while (hi - lo > 1) {
    uint32_t mid = (lo + hi) >> 1;
    if ((off[mid] - off[u0]) + (mid - u0) <= si) lo = mid; else hi = mid;
  }
  return lo;
}

// X_mm[row] = [running max (W) | running min (W) | one-hot dur (D)]: columns 6W.. of the window vector
__global__ void k_windows_mm(const float* __restrict__ frames, const uint64_t* __restrict__ sframe_off,
                             ScrfBatchView bv, uint32_t u0, uint32_t u1, uint32_t W, uint32_t D,
                             float* __restrict__ X, uint32_t F) {
  const uint64_t gf = bv.frame_off[u0] + blockIdx.x;
  const uint32_t u = find_utt_f(bv.frame_off, u0, u1, gf);
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint32_t avail = scrf_node_max_dur(t, D);
  const uint64_t rowbase = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const float* last = frames + (sframe_off[u] + t) * (uint64_t)W;
  for (uint32_t j = threadIdx.x; j < W; j += blockDim.x) {
    float mx = last[j], mn = last[j];
    for (uint32_t w = 1; w <= avail; w++) {
      const float x = *(last - (uint64_t)(w - 1) * W + j);
      if (x > mx) mx = x;
      if (x < mn) mn = x;
      float* o = X + (rowbase + w - 1) * (uint64_t)F;
      o[j] = mx;
      o[W + j] = mn;
    }
  }
  for (uint32_t idx = threadIdx.x; idx < avail * D; idx += blockDim.x) {
    const uint32_t w = idx / D + 1, k = idx % D;
    X[(rowbase + w - 1) * (uint64_t)F + 2 * W + k] = (k + 1 == w) ? 1.0f : 0.0f;
  }
}
void launch_windows_mm(hipStream_t st, const float* frames, const uint64_t* sframe_off, ScrfBatchView bv,
                       uint32_t u0, uint32_t u1, uint64_t n_frames, uint32_t W, uint32_t D, float* X, uint32_t F) {
  if (n_frames == 0) return;
  const uint32_t bs = W <= 64 ? 64 : (W <= 128 ? 128 : 256);
  hipLaunchKernelGGL(k_windows_mm, dim3((uint32_t)n_frames), dim3(bs), 0, st, frames, sframe_off, bv, u0, u1, W, D,
                     X, F);
}

// CA[f][l] = sum_{f' < f} P[f'][5][l] over the chunk's frames (running prefix; a window sum is the
// difference of two entries inside one utterance), three small passes: per-block sums, scan of
// the block sums, per-block prefix.  Also fills the [D][5] table of sample offsets.
#define LP_FB 256  // frames per block
__global__ __launch_bounds__(256) void k_lin_prefix_a(ScrfLayout lay, uint64_t n_frames, const double* __restrict__ P,
                                                      double* __restrict__ blocksum) {
  const uint32_t L = lay.L;
  const size_t PL = (size_t)6 * L;
  const uint64_t f0 = (uint64_t)blockIdx.x * LP_FB;
  for (uint32_t l = threadIdx.x; l < L; l += blockDim.x) {
    double s = 0.0;
    for (uint32_t i = 0; i < LP_FB && f0 + i < n_frames; i++) s += P[(f0 + i) * PL + 5 * L + l];
    blocksum[(size_t)blockIdx.x * L + l] = s;
  }
}
__global__ __launch_bounds__(256) void k_lin_prefix_b(ScrfLayout lay, uint32_t n_blocks, double* __restrict__ blocksum,
                                                      uint8_t* __restrict__ steps) {
  const uint32_t L = lay.L;
  for (uint32_t l = threadIdx.x; l < L; l += blockDim.x) {
    double run = 0.0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      const double v = blocksum[(size_t)b * L + l];
      blocksum[(size_t)b * L + l] = run;
      run += v;
    }
  }
  for (uint32_t i = threadIdx.x; i < lay.D * 5; i += blockDim.x) steps[i] = (uint8_t)sample_step(i / 5 + 1, i % 5);
}
__global__ __launch_bounds__(256) void k_lin_prefix_c(ScrfLayout lay, uint64_t n_frames, const double* __restrict__ P,
                                                      const double* __restrict__ blocksum, double* __restrict__ CA) {
  const uint32_t L = lay.L;
  const size_t PL = (size_t)6 * L;
  const uint64_t f0 = (uint64_t)blockIdx.x * LP_FB;
  for (uint32_t l = threadIdx.x; l < L; l += blockDim.x) {
    double run = blocksum[(size_t)blockIdx.x * L + l];
    for (uint32_t i = 0; i < LP_FB && f0 + i < n_frames; i++) {
      CA[(f0 + i) * L + l] = run;
      run += P[(f0 + i) * PL + 5 * L + l];
    }
    if (f0 + LP_FB >= n_frames) CA[n_frames * L + l] = run;
  }
}
void launch_lin_prefix(hipStream_t st, const ScrfLayout& lay, uint64_t n_frames, const double* P, double* CA,
                       double* blocksum, uint8_t* steps) {
  if (n_frames == 0) return;
  const uint32_t nb = (uint32_t)((n_frames + LP_FB - 1) / LP_FB);
  hipLaunchKernelGGL(k_lin_prefix_a, dim3(nb), dim3(64), 0, st, lay, n_frames, P, blocksum);
  hipLaunchKernelGGL(k_lin_prefix_b, dim3(1), dim3(256), 0, st, lay, nb, blocksum, steps);
  hipLaunchKernelGGL(k_lin_prefix_c, dim3(nb), dim3(64), 0, st, lay, n_frames, P, blocksum, CA);
}

// Z[slot][k][l] (k<5) = sum of R over the windows whose k-th sample is this frame;
// Z'[slot][5][l] = sum_d R[slot-1][d]/d (windows ending just before) - sum_d R[slot+d-1][d]/d (windows starting here)
__global__ __launch_bounds__(256) void k_lin_expf_z(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                                                    uint64_t n_slots, const double* __restrict__ R,
                                                    double* __restrict__ Z, const uint8_t* __restrict__ steps) {
  __shared__ uint8_t stp_s[5 * 64];
  const uint32_t L = lay.L, D = lay.D;
  for (uint32_t i = threadIdx.x; i < D * 5 && i < 5 * 64; i += blockDim.x) stp_s[i] = steps[i];
  __syncthreads();
  const uint64_t n_groups = (n_slots + 3) / 4;
  const uint64_t grp = xcd_remap(blockIdx.x, n_groups);
  if (grp >= n_groups) return;
  const uint64_t si = grp * 4 + (threadIdx.x >> 6);
  if (si >= n_slots) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t u = find_utt_slot(bv.frame_off, u0, u1, si);
  const uint32_t s = (uint32_t)(si - ((bv.frame_off[u] - bv.frame_off[u0]) + (u - u0)));
  const uint32_t T = bv.T[u];
  const double* Ru = R + (bv.seg_off[u] - bv.seg_off[u0]) * L;
  const size_t PL = (size_t)6 * L;
  for (uint32_t l = lane; l < L; l += 64) {
    double z[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double za = 0.0;
    if (s < T) {
      for (uint32_t d = 1; d <= D; d++) {
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const uint32_t stp = stp_s[(d - 1) * 5 + k];
          if (s >= stp) {
            const uint32_t t = s - stp + d - 1;
            if (t < T) z[k] += Ru[(scrf_seg_base(t, D) + d - 1) * L + l];
          }
        }
        const uint32_t te = s + d - 1;  // window [s, te] of length d starts here
        if (te < T) za -= Ru[(scrf_seg_base(te, D) + d - 1) * L + l] / (double)d;
      }
    }
    if (s >= 1) {
      const uint32_t t = s - 1;
      const uint32_t nd = scrf_node_max_dur(t, D);
      const uint64_t b = scrf_seg_base(t, D);
      for (uint32_t d = 1; d <= nd; d++) za += Ru[(b + d - 1) * L + l] / (double)d;
    }
#pragma unroll
    for (int k = 0; k < 5; k++) Z[si * PL + (size_t)k * L + l] = z[k];
    Z[si * PL + 5 * L + l] = za;
  }
}
void launch_lin_expf_z(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t u1,
                       uint64_t n_slots, const double* R, double* Z, const uint8_t* steps) {
  if (n_slots == 0) return;
  const uint64_t n_groups = (n_slots + 3) / 4;
  hipLaunchKernelGGL(k_lin_expf_z, dim3((uint32_t)(((n_groups + 7) / 8) * 8)), dim3(256), 0, st, lay, bv, u0, u1,
                     n_slots, R, Z, steps);
}

// avg block: Z[s][5][l] <- sum_{f > s} Z'[f][5][l]   (coefficient of frame s in sum_f CF[f] Z'[f]);
// one wavefront per utterance, also fills the slot -> frame-row map of the final contraction
__global__ __launch_bounds__(256) void k_suffix_avg(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                                                    double* __restrict__ Z, uint64_t* __restrict__ slot_row) {
  const uint32_t L = lay.L;
  const uint32_t ul = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ul >= n_utts) return;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t u = u0 + ul;
  const uint32_t T = bv.T[u];
  const uint64_t fb = bv.frame_off[u] - bv.frame_off[u0];
  const uint64_t sb = fb + ul;
  const size_t PL = (size_t)6 * L;
  for (uint32_t l = lane; l < L; l += 64) {
    double run = 0.0;
    for (uint32_t s = T + 1; s-- > 0;) {
      const double v = Z[(sb + s) * PL + 5 * L + l];
      Z[(sb + s) * PL + 5 * L + l] = run;
      run += v;
    }
  }
  for (uint32_t s = lane; s <= T; s += 64) slot_row[sb + s] = fb + (s < T ? s : T - 1);
}
void launch_suffix_avg(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                       double* Z, uint64_t* slot_row) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_suffix_avg, dim3((n_utts + 3) / 4), dim3(256), 0, st, lay, bv, u0, n_utts, Z, slot_row);
}
