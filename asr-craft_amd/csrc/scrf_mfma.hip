// scrf_mfma.hip -- MFMA (v_mfma_f64_16x16x4_f64) contractions of the FAST training path.
//
//   k_scores_mfma : out[row][o]   = sum_f x[row][f] * lambda[woff(o)+f] + bias   (M = rows, N = outputs, K = features)
//   k_expf_mfma   : slab[z][o][f] = sum_rows A[row][o] * xs[row][f]              (M = outputs, N = features, K = rows)
//
// Same mathematics as the EXACT kernels; the MFMA fuses and reorders the fp64 sums, so results
// differ from the reference chain in the last bits (training contract: 1e-4 relative).
//
// v_mfma_f64_16x16x4_f64 operand maps (cdna_hip_programming.md section 3): lane l holds
//   A[i = l&15][k = l>>4], B[k = l>>4][j = l&15]; D reg r: row = (l>>4) + 4r, col = l&15.
// LDS images are laid out so that every fragment read is conflict-free:
//   - k-major [k][n] doubles with an even row stride of 96 words: the two 32-lane halves of a
//     ds_read_b64 (k, k+1) fall on disjoint bank halves;
//   - row-major floats with stride == 2 (mod 32) words for the A operand of the score kernel.
#include "scrf_kernels.h"

#include <stdlib.h>
#include <string.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
// 16-byte loads from 4-byte aligned addresses: hipcc emits global_load_dwordx4 for these
struct __attribute__((packed, aligned(4))) f4u { float x, y, z, w; };
struct __attribute__((packed, aligned(8))) d2u { double x, y; };

#ifndef EM_ABL
#define EM_ABL 0     // timing ablations of k_expf_mfma (wrong results): 1 = no MFMA loop, 2 = no staging after the first chunks
#endif
#ifndef MM_PRIO
#define MM_PRIO 1    // s_setprio level of the staging phases (0 = no priority play; A/B knob)
#endif
#define SM_ROWS 256  // rows per workgroup (4 waves x 4 M-tiles)
#define SM_KC 32     // features per staged chunk (8 MFMA k-steps)
#define SM_XS 34     // float row stride of the X image (== 2 mod 32: conflict-free A fragments)
#define SM_NO 48     // outputs per workgroup (3 N-tiles)
#define SM_WS 49     // double row stride of the lambda image [k][SM_WS]: transposed stores stay <= 2-way

// Software pipeline: while the MFMAs of chunk i run from LDS, the global loads of chunk i+1
// are in flight into registers (8 x 16 B of X and 6 x 8 B of lambda per thread).
typedef float v4f32 __attribute__((ext_vector_type(4)));

// F32 = 1 (train_precision FAST32): v_mfma_f32_16x16x4_f32 -- an exact f32 FMA chain at ~3x the
// sustained rate of the f64 MFMA on this chip; lambda is rounded to f32 at staging, the K-sum
// runs in f32 (error ~1e-7 * sum|x*w|), bias and the linear epilogue are added in f64.
// D layout of the f32 form: row = 4*(lane>>4) + reg (the f64 form: (lane>>4) + 4*reg).
// launch bound 2 waves per SIMD: left alone the compiler takes 210 VGPRs + 96 AGPRs for the f64 form
// (one wavefront per SIMD, matrix pipe 47 % busy at the TIMIT transition scores); capped at 256 it
// needs 218 with no spills and the second workgroup per CU hides staging and barriers (1.35x)
// NT: N-tiles (16 outputs each) of this launch's workgroups: 3, or the 1 / 2 of the remainder launch when the output
// count is not a multiple of 48 (200 outputs = 4 x 48 + 8: empty tiles would cost 2/15 of the matrix time)
template <int F32, int NT>
__global__ __launch_bounds__(256, 2) void k_scores_mfma(const float* __restrict__ X, uint32_t F,
                                                     const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                     const double* __restrict__ lambda, ScrfLayout lay,
                                                     ScrfGemmSpec sp, uint32_t n_out, uint32_t o_base, uint32_t gy, double* __restrict__ out) {
  __shared__ float Xs[SM_ROWS * SM_XS];
  __shared__ double Ws[SM_KC * SM_WS];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t li = lane & 15, lk = lane >> 4;
  // 1-D launch: the gy output tiles of a row block are neighbours in the swizzled order (one XCD reads the block's X
  // rows once for all of them; the lambda tiles come from L2 / the Infinity Cache)
  const uint32_t swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint64_t row0 = (uint64_t)(swz / gy) * SM_ROWS;
  const uint32_t o0 = o_base + (swz % gy) * SM_NO;
  const uint32_t fs = sp.fs;
  const uint32_t nfe = sp.nfe;
  const int use_b = sp.use_bias;
  const double bv = sp.bias;

  v4f64 acc[F32 ? 1 : 4][F32 ? 1 : NT];
  v4f32 acc32[F32 ? 4 : 1][F32 ? NT : 1];
#pragma unroll
  for (int m = 0; m < (F32 ? 1 : 4); m++)
#pragma unroll
    for (int n = 0; n < (F32 ? 1 : NT); n++) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < (F32 ? 4 : 1); m++)
#pragma unroll
    for (int n = 0; n < (F32 ? NT : 1); n++) acc32[m][n] = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};
  float* Wsf = (float*)Ws;  // F32: the lambda image holds floats, row stride 2*SM_WS floats

  // staging coordinates: 8 threads cover one row's 32-float chunk, 32 rows per pass, 8 passes
  const uint32_t sq = tid & 7, sr = tid >> 3;
  const float* xbase[8];
#pragma unroll
  for (int it = 0; it < 8; it++) {
    const uint64_t r = row0 + sr + it * 32;
    const uint64_t rr = r < n_rows ? r : (n_rows - 1);
    const uint64_t xr = xrow ? xrow[rr] : rr;
    xbase[it] = X + xr * F + fs + sq * 4;
  }
  // lambda chunk: thread loads W[o][c] for idx = tid + k*256 -> o = idx / 32, c = idx % 32
  const double* wbase[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const uint32_t idx = tid + k * 256;
    uint32_t o = o0 + idx / SM_KC;
    if (o >= n_out) o = n_out - 1;
    wbase[k] = lambda + sp.woff(lay, o) + (idx % SM_KC);
  }
  f4u xr_[8];
  double wr_[6];
  // offset that keeps (sq*4 + offset) inside the feature range for the masked tail quads
  const uint32_t fclamp = (nfe > sq * 4 + 4) ? ((nfe - sq * 4 - 1) & ~31u) : 0;
  auto load_chunk = [&](uint32_t f0) {
    // branch-free: a quad that starts inside the feature range may read <= 12 B past it, quads
    // fully outside re-read the last valid quad; both are masked in store_chunk (buffers carry
    // 256 B of tail padding)
    const uint32_t fo = (f0 + sq * 4 < nfe) ? f0 : fclamp;
#pragma unroll
    for (int it = 0; it < 8; it++) xr_[it] = *(const f4u*)(xbase[it] + fo);
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint32_t c = (tid + k * 256) % SM_KC;
      const double w = wbase[k][(f0 + c < nfe) ? f0 : 0];
      wr_[k] = (f0 + c < nfe) ? w : 0.0;
    }
  };
  auto store_chunk = [&](uint32_t f0) {
    const uint32_t c0 = f0 + sq * 4;
#pragma unroll
    for (int it = 0; it < 8; it++) {
      f4u v = xr_[it];
      if (c0 + 0 >= nfe) v.x = 0.0f;
      if (c0 + 1 >= nfe) v.y = 0.0f;
      if (c0 + 2 >= nfe) v.z = 0.0f;
      if (c0 + 3 >= nfe) v.w = 0.0f;
      float* d = &Xs[(sr + it * 32) * SM_XS + sq * 4];
      *(float2*)(d) = make_float2(v.x, v.y);
      *(float2*)(d + 2) = make_float2(v.z, v.w);
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const uint32_t idx = tid + k * 256;
      if (F32) Wsf[(idx % SM_KC) * (2 * SM_WS) + idx / SM_KC] = (float)wr_[k];
      else Ws[(idx % SM_KC) * SM_WS + idx / SM_KC] = wr_[k];
    }
  };

  // waves that stage (vector work) outrank waves inside the MFMA loop of the co-resident workgroup (as in scrf_fused.hip)
  __builtin_amdgcn_s_setprio(MM_PRIO);
  if (nfe > 0) load_chunk(0);
  for (uint32_t f0 = 0; f0 < nfe; f0 += SM_KC) {
    store_chunk(f0);
    __syncthreads();
    if (f0 + SM_KC < nfe) load_chunk(f0 + SM_KC);
    __builtin_amdgcn_s_setprio(0);
#pragma unroll
    for (int ks = 0; ks < SM_KC / 4; ks++) {
      if (F32) {
        float b[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) b[n] = Wsf[(ks * 4 + lk) * (2 * SM_WS) + n * 16 + li];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const float a = Xs[(wave * 64 + m * 16 + li) * SM_XS + ks * 4 + lk];
#pragma unroll
          for (int n = 0; n < NT; n++)
            acc32[F32 ? m : 0][F32 ? n : 0] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[n], acc32[F32 ? m : 0][F32 ? n : 0], 0, 0, 0);
        }
      } else {
        double b[NT];
#pragma unroll
        for (int n = 0; n < NT; n++) b[n] = Ws[(ks * 4 + lk) * SM_WS + n * 16 + li];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const double a = (double)Xs[(wave * 64 + m * 16 + li) * SM_XS + ks * 4 + lk];
#pragma unroll
          for (int n = 0; n < NT; n++)
            acc[F32 ? 0 : m][F32 ? 0 : n] =
                __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[n], acc[F32 ? 0 : m][F32 ? 0 : n], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(MM_PRIO);
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NT; n++) {
    const uint32_t o = o0 + n * 16 + li;
    if (o >= n_out) continue;
    double bias = 0.0;
    if (use_b) bias = lambda[sp.woff(lay, o) + nfe] * bv;
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t rl = wave * 64 + m * 16 + (F32 ? 4 * lk + r : lk + 4 * r);
        const uint64_t row = row0 + rl;
        if (row >= n_rows) continue;
        double v = (F32 ? (double)acc32[F32 ? m : 0][F32 ? n : 0][r] : acc[F32 ? 0 : m][F32 ? 0 : n][r]) + bias;
        out[row * n_out + o] = v;
      }
  }
}

// ------------------------------------------------------------------------------------------
// k_scores_mfma_ws: the f64 score contraction with the roles split between wavefronts, like k_expf_mfma_ws below.  A
// workgroup tile is 256 rows x 96 outputs: consumer wavefront w (0-7) owns rows 64 (w & 3) and outputs 48 (w >> 2)
// (12 MFMAs per k-step from 4 + 3 operand reads), wavefronts 8-11 stage the X chunk (8 quads per thread) and the
// lambda chunk (12 weights per thread) of the NEXT 32 features into the other LDS image pair and request the chunk after
// it; one barrier per chunk.  The X rows are read from L2 once per 96 outputs instead of once per 48.
// lambda image [k][SW_WS] doubles, SW_WS = 113 (== 17 mod 32): the consumers' fragment reads (k = lane >> 4 rows, 16
// consecutive outputs) and the producers' transposed stores (32 lanes = 32 values of k for one output) both spread over
// all banks.
// Two shapes of the 8 consumer wavefronts: <4, 3> -- rows 64 (w & 3), outputs 48 (w >> 2): 96 outputs per workgroup; <2, 7> --
// rows 32 w, all 7 N-tiles: 112 outputs per workgroup (14 MFMAs per 2 + 7 reads).  An output count of 16 T is cut into
// a tiles of 6 and b tiles of 7 N-tiles (T = 6a + 7b: 200 outputs = 96 + 112 with 8 masked), so that no thin remainder
// launch re-reads all of X for a handful of outputs (config 5: 9 ms for 8 of 200).
#define SW_ROWS 256
#define SW_NO 96
#define SW_WS 113
#define SW_IMG (sizeof(float) * SW_ROWS * SM_XS + sizeof(double) * SM_KC * SW_WS)
template <int MW, int NTW>
__global__ __launch_bounds__(768) void k_scores_mfma_ws(const float* __restrict__ X, uint32_t F,
                                                       const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                       const double* __restrict__ lambda, ScrfLayout lay,
                                                       ScrfGemmSpec sp, uint32_t n_out, uint32_t o_base, uint32_t gy, double* __restrict__ out) {
  constexpr int RG = 256 / (16 * MW);            // row groups: 4 (64 rows each) or 8 (32 rows each)
  constexpr int OG = 8 / RG;                     // output groups
  constexpr int NOW = 16 * NTW * OG;             // outputs per workgroup
  constexpr int WPT = NOW * SM_KC / 256;         // lambda weights per producer thread and chunk
  static_assert(NOW <= SW_WS - 1 && NOW * SM_KC % 256 == 0, "tile does not fit the lambda image");
  extern __shared__ __attribute__((aligned(16))) unsigned char sw_smem[];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint64_t row0 = (uint64_t)(swz / gy) * SW_ROWS;
  const uint32_t o0 = o_base + (swz % gy) * NOW;
  const uint32_t fs = sp.fs, nfe = sp.nfe;
  const uint32_t n_it = (nfe + SM_KC - 1) / SM_KC;

  if (wave >= 8) {
    // ---------------- producers ----------------
    __builtin_amdgcn_s_setprio(2);
    const uint32_t pt = tid - 512;
    const uint32_t sq = pt & 7, sr = pt >> 3;   // 8 threads cover one row's 32-float chunk, 32 rows per pass, 8 passes
    const float* xbase[8];
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const uint64_t r = row0 + sr + it * 32;
      const uint64_t rr = r < n_rows ? r : (n_rows - 1);
      const uint64_t xr = xrow ? xrow[rr] : rr;
      xbase[it] = X + xr * F + fs + sq * 4;
    }
    // lambda chunk: W[o][c] for idx = pt + k * 256 -> c = idx % 32 (the same for every k), o = idx / 32
    const uint32_t wc = pt % SM_KC;
    const double* wbase[WPT];
#pragma unroll
    for (int k = 0; k < WPT; k++) {
      uint32_t o = o0 + (pt + k * 256) / SM_KC;
      if (o >= n_out) o = n_out - 1;
      wbase[k] = lambda + sp.woff(lay, o) + wc;
    }
    f4u xr_[8];
    double wr_[WPT];
    const uint32_t fclamp = (nfe > sq * 4 + 4) ? ((nfe - sq * 4 - 1) & ~31u) : 0;
    auto load = [&](uint32_t f0) {
      // as in k_scores_mfma: a quad that starts inside the feature range may read <= 12 B past it, quads outside re-read
      // a valid quad; both are masked at the store (buffers carry tail padding)
      const uint32_t fo = (f0 + sq * 4 < nfe) ? f0 : fclamp;
#pragma unroll
      for (int it = 0; it < 8; it++) xr_[it] = *(const f4u*)(xbase[it] + fo);
      const bool wok = f0 + wc < nfe;
#pragma unroll
      for (int k = 0; k < WPT; k++) {
        const double w = wbase[k][wok ? f0 : 0];
        wr_[k] = wok ? w : 0.0;
      }
    };
    auto store = [&](uint32_t f0, uint32_t buf) {
      float* Xs = (float*)(sw_smem + buf * SW_IMG);
      double* Ws = (double*)(sw_smem + buf * SW_IMG + sizeof(float) * SW_ROWS * SM_XS);
      const uint32_t c0 = f0 + sq * 4;
#pragma unroll
      for (int it = 0; it < 8; it++) {
        f4u v = xr_[it];
        if (c0 + 0 >= nfe) v.x = 0.0f;
        if (c0 + 1 >= nfe) v.y = 0.0f;
        if (c0 + 2 >= nfe) v.z = 0.0f;
        if (c0 + 3 >= nfe) v.w = 0.0f;
        float* d = &Xs[(sr + it * 32) * SM_XS + sq * 4];
        *(float2*)(d) = make_float2(v.x, v.y);
        *(float2*)(d + 2) = make_float2(v.z, v.w);
      }
#pragma unroll
      for (int k = 0; k < WPT; k++) Ws[wc * SW_WS + (pt + k * 256) / SM_KC] = wr_[k];
    };
    if (n_it) { load(0); store(0, 0); }
    if (n_it > 1) load(SM_KC);
    __syncthreads();
    for (uint32_t i = 0; i < n_it; i++) {
      if (i + 1 < n_it) store((i + 1) * SM_KC, (i + 1) & 1u);
      if (i + 2 < n_it) load((i + 2) * SM_KC);
      __syncthreads();
    }
    return;
  }
  // ---------------- consumers ----------------
  const uint32_t rg = wave % RG, og = wave / RG;
  v4f64 acc[MW][NTW];
#pragma unroll
  for (int m = 0; m < MW; m++)
#pragma unroll
    for (int n = 0; n < NTW; n++) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  for (uint32_t i = 0; i < n_it; i++) {
    const float* Xs = (const float*)(sw_smem + (i & 1u) * SW_IMG);
    const double* Ws = (const double*)(sw_smem + (i & 1u) * SW_IMG + sizeof(float) * SW_ROWS * SM_XS);
#pragma unroll
    for (int ks = 0; ks < SM_KC / 4; ks++) {
      double b[NTW];
#pragma unroll
      for (int n = 0; n < NTW; n++) b[n] = Ws[(ks * 4 + lk) * SW_WS + og * (16 * NTW) + n * 16 + li];
#pragma unroll
      for (int m = 0; m < MW; m++) {
        const double a = (double)Xs[(rg * (16 * MW) + m * 16 + li) * SM_XS + ks * 4 + lk];
#pragma unroll
        for (int n = 0; n < NTW; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[n], acc[m][n], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const int use_b = sp.use_bias;
  const double bv = sp.bias;
#pragma unroll
  for (int n = 0; n < NTW; n++) {
    const uint32_t o = o0 + og * (16 * NTW) + n * 16 + li;
    if (o >= n_out) continue;
    double bias = 0.0;
    if (use_b) bias = lambda[sp.woff(lay, o) + nfe] * bv;
#pragma unroll
    for (int m = 0; m < MW; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint64_t row = row0 + rg * (16 * MW) + m * 16 + lk + 4 * r;
        if (row >= n_rows) continue;
        out[row * n_out + o] = acc[m][n][r] + bias;
      }
  }
}

template <int F32>
static void launch_scores_mfma_f(hipStream_t st, const float* X, uint32_t F, const uint64_t* xrow, uint64_t n_rows,
                                 const double* lambda, const ScrfLayout& lay, const ScrfGemmSpec& sp, uint32_t n_out, double* out) {
  const uint32_t gx = (uint32_t)((n_rows + SM_ROWS - 1) / SM_ROWS);
  uint32_t o_base = 0;
  static const bool ws_off = getenv("SCRF_SCORES_MFMA_WS") && atoi(getenv("SCRF_SCORES_MFMA_WS")) == 0;   // A/B knob
  // (from 192 features on: with only a few 32-feature chunks the role split is all prologue -- the 40-feature per-window
  // transition scores of STDSEG_NO_DUR: 5.8 ms single-role, 6.1 split)
  if (!F32 && !ws_off && n_out >= SW_NO && sp.nfe >= 192) {
    // the wave-specialised form: 16 T outputs as a tiles of 96 and b tiles of 112 (T = 6a + 7b, fewest masked outputs); when
    // the count does not split that way, the whole 96-output tiles go here and the rest to the single-role kernels below
    const uint32_t T = (n_out + 15) / 16;
    uint32_t a = n_out / SW_NO, bt = 0;
    for (uint32_t b7 = 0; b7 < 6 && 7 * b7 <= T; b7++)
      if ((T - 7 * b7) % 6 == 0) { a = (T - 7 * b7) / 6; bt = b7; break; }
    if (a) {
      hipFuncSetAttribute((const void*)k_scores_mfma_ws<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * SW_IMG));
      hipLaunchKernelGGL((k_scores_mfma_ws<4, 3>), dim3(gx * a), dim3(768), 2 * SW_IMG, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, 0u, a, out);
    }
    if (bt) {
      hipFuncSetAttribute((const void*)k_scores_mfma_ws<2, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * SW_IMG));
      hipLaunchKernelGGL((k_scores_mfma_ws<2, 7>), dim3(gx * bt), dim3(768), 2 * SW_IMG, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, a * SW_NO, bt, out);
    }
    o_base = a * SW_NO + bt * 112;
    if (o_base >= n_out) return;
  }
  const uint32_t left = n_out - o_base;
  const uint32_t n_full = left / SM_NO, rem = left % SM_NO;
  if (n_full)
    hipLaunchKernelGGL((k_scores_mfma<F32, 3>), dim3(gx * n_full), dim3(256), 0, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, o_base, n_full, out);
  // the outputs past the last full 48: a launch whose workgroups carry only the N-tiles that hold outputs
  if (rem > 32)
    hipLaunchKernelGGL((k_scores_mfma<F32, 3>), dim3(gx), dim3(256), 0, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, o_base + n_full * SM_NO, 1u, out);
  else if (rem > 16)
    hipLaunchKernelGGL((k_scores_mfma<F32, 2>), dim3(gx), dim3(256), 0, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, o_base + n_full * SM_NO, 1u, out);
  else if (rem > 0)
    hipLaunchKernelGGL((k_scores_mfma<F32, 1>), dim3(gx), dim3(256), 0, st, X, F, xrow, n_rows, lambda, lay, sp, n_out, o_base + n_full * SM_NO, 1u, out);
}
void launch_scores_mfma(hipStream_t st, const float* X, uint32_t F, const uint64_t* xrow, uint64_t n_rows,
                        const double* lambda, const ScrfLayout& lay, const ScrfGemmSpec& sp, uint32_t n_out,
                        double* out, int f32) {
  if (n_rows == 0 || n_out == 0) return;
  if (f32) launch_scores_mfma_f<1>(st, X, F, xrow, n_rows, lambda, lay, sp, n_out, out);
  else launch_scores_mfma_f<0>(st, X, F, xrow, n_rows, lambda, lay, sp, n_out, out);
}

// ------------------------------------------------------------------------------------------
// M-tiles (16 outputs each) of a workgroup tile of the unsplit form.  f64: 4 -- a wavefront carries 64 outputs x 48 feature
// columns, 12 MFMAs per 7 operand loads and 96 per barrier (3: 9 per 6, 72 per barrier; measured at the TIMIT transition
// counts 14.8 -> 13.5 ms, config 5's state counts 96.6 -> 91.1 ms; 231 VGPRs, no spills).  The f32 form spills at 4 and stays at 3.
#ifndef EM_MTF
#define EM_MTF 4
#endif
// (the narrow forms keep 3: with 1-4 wavefronts per workgroup a 64-wide posterior image means a third more staging
// registers per thread -- the frame model's 40-column count contraction went from 0.58 to 1.54 ms with it)
#define EM_MTW(F32, NW) (((F32) || (NW) < 8) ? 3 : EM_MTF)
#define EM_SPLIT_NO 48        // outputs per wavefront of the split form (3 M-tiles)
// row stride (doubles) of the posterior image [k][outputs]: == 16 (mod 32), so that the two k rows a 32-lane half of a
// ds_read_b64 fragment read covers fall on disjoint bank halves (a stride of 64 or 192 doubles puts them on the same banks)
// (only the 64-wide image of the wide form is padded: the split form's 192-wide image measured FASTER unpadded, 4.8 against
// 6.6 ms at the per-window transition counts of STDSEG_NO_DUR)
constexpr int em_rss(int no) { return no == 64 ? 80 : no; }

// NW wavefronts per workgroup, each owning 48 feature columns (3 N-tiles): NW = 8 covers 384
// columns (the full 338-wide state block of config 2 in one workgroup, so R is read once);
// narrow contractions (factorised max/min/dur block, per-frame projections) use fewer waves.
// KC = rows per staged chunk (KC/4 MFMA k-steps): 32 for the wide tile, 64 for narrow tiles so
// that the staging/barrier cost per MFMA stays low when few wavefronts share one R tile.
// F32 = 1: operands rounded to f32 (R is in [-1,1]), v_mfma_f32_16x16x4_f32 within a staged chunk of
// EM_KC rows, chunk results flushed into f64 accumulators (so the K = millions-of-rows sum is f64).
// MT: M-tiles (16 outputs each) a wavefront carries: 3, or the 1 / 2 of the remainder launch (o_base = its first output)
// DB = 1 (the 8-wave form, which owns its CU): two LDS image pairs; a chunk's MFMAs run from one while the next chunk is
// stored into the other, one barrier per chunk instead of two
template <int HAS_XROW, int NW, int EM_KC, int F32, int SPLIT_OUT = 0, int MT = 3, int DB = 0, int MTW = 3>
__global__ __launch_bounds__(64 * NW) void k_expf_mfma(const double* __restrict__ A, uint32_t n_out,
                                                      const float* __restrict__ X, uint32_t F,
                                                      const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                      ScrfLayout lay, ScrfGemmSpec sp, uint64_t rows_per_chunk,
                                                      double* __restrict__ slab, uint32_t o_base, uint32_t gx, uint32_t gy, uint32_t o_step) {
  constexpr int NT = 64 * NW;            // threads
  // SPLIT_OUT: the wavefronts share one 48-column feature tile and own 48 outputs each (few feature functions, many
  // outputs: the per-window transition posteriors, n_out = L * L); otherwise 48 outputs and 48 feature columns per wavefront
  constexpr int NF = SPLIT_OUT ? 48 : 48 * NW;   // feature columns per workgroup
  constexpr int NO = SPLIT_OUT ? EM_SPLIT_NO * NW : 16 * MTW;   // outputs per workgroup (MTW M-tiles wide; MT <= MTW of them hold outputs)
  constexpr int XS = NF + 16;            // float row stride of the X image (== 16 mod 32)
  constexpr int QR = NF / 4;             // 16-byte quads per row
  constexpr int XIT = (EM_KC * QR + NT - 1) / NT;   // X quads per thread per chunk (= 6; the split form has threads without one)
  constexpr int AIT = (EM_KC * NO + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char em_smem[];
  constexpr int RSS = em_rss(NO);        // double row stride of the posterior image
  constexpr size_t IMG = sizeof(double) * EM_KC * RSS + sizeof(float) * EM_KC * XS;   // bytes of one image pair
  double* Rs = (double*)em_smem;                                  // [EM_KC][RSS]   (+ IMG bytes: the second pair when DB)
  float* Xs = (float*)(em_smem + sizeof(double) * EM_KC * RSS);  // [EM_KC][XS]
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t fs = sp.fs;
  const uint32_t nfe = sp.nfe;
  const uint32_t nfun = sp.nfun();
  const float bias = sp.use_bias ? (float)sp.bias : 0.0f;
  const bool bias_exact = (double)bias == sp.bias;
  // 1-D launch: the gx * gy tiles of a row chunk are neighbours in the swizzled order (they walk the same rows of A and X
  // at the same time: one XCD's L2 serves them)
  const uint32_t swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint32_t bx = swz % gx, by = (swz / gx) % gy, bz = swz / (gx * gy);
  const uint32_t fb = bx * NF;
  const uint32_t o0 = o_base + by * o_step;   // o_step: outputs between this launch's tiles (16 MT; the image stays NO wide)
  const uint32_t wo = SPLIT_OUT ? wave * 48 : 0, wf = SPLIT_OUT ? 0 : wave * 48;   // this wavefront's output / feature offset in the tile
  const uint64_t r_begin = (uint64_t)bz * rows_per_chunk;
  const uint64_t r_end = min(n_rows, r_begin + rows_per_chunk);

  v4f64 acc[MT][3];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < 3; n++) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};

  // Staging coordinates and tail masks are fixed per thread: computed once, outside the loop.
  f4u xr_[XIT];
  double ar_[AIT];
  uint32_t xrw[XIT], xcl[XIT], xlds[XIT], arw[AIT], acol[AIT], alds[AIT];
  bool aok[AIT];
  float xfill[XIT][4];   // value for masked components: bias column or zero
  bool xkeep[XIT][4];
#pragma unroll
  for (int k = 0; k < XIT; k++) {
    const uint32_t idx = tid + k * NT;
    xrw[k] = idx / QR;
    const uint32_t q = idx % QR;
    const uint32_t col = fb + q * 4;
    xlds[k] = xrw[k] * XS + q * 4;
    // quads outside the feature range re-read a valid quad (masked at the LDS store)
    xcl[k] = col < nfe ? col : (nfe >= 4 ? nfe - 4 : 0);
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const uint32_t cc = col + c;
      xkeep[k][c] = cc < nfe;
      xfill[k][c] = (cc == nfe && sp.use_bias) ? bias : 0.0f;
    }
  }
#pragma unroll
  for (int k = 0; k < AIT; k++) {
    const uint32_t idx = tid + k * NT;
    arw[k] = idx / NO;
    alds[k] = arw[k] * RSS + idx % NO;   // where the element goes in the posterior image
    const uint32_t o = o0 + idx % NO;
    aok[k] = idx < EM_KC * NO && o < n_out;
    acol[k] = o < n_out ? o : n_out - 1;
  }
  // branch-free loads: rows past the chunk end re-read its last row (may read <= 12 B past the
  // feature range; buffers carry tail padding)
  auto load_chunk = [&](uint64_t r0) {
    uint64_t xr[XIT];
#pragma unroll
    for (int k = 0; k < XIT; k++) {
      uint64_t row = r0 + xrw[k];
      row = row < r_end ? row : r_end - 1;
      xr[k] = HAS_XROW ? xrow[row] : row;
    }
#pragma unroll
    for (int k = 0; k < XIT; k++) xr_[k] = *(const f4u*)(X + xr[k] * F + fs + xcl[k]);
#pragma unroll
    for (int k = 0; k < AIT; k++) {
      uint64_t row = r0 + arw[k];
      const bool ok = row < r_end && aok[k];
      row = row < r_end ? row : r_end - 1;
      const double v = A[row * n_out + acol[k]];
      ar_[k] = ok ? v : 0.0;
    }
  };
  auto store_chunk = [&](uint64_t r0, uint32_t buf) {
    double* Rs = (double*)(em_smem + buf * IMG);
    float* Xs = (float*)(em_smem + buf * IMG + sizeof(double) * EM_KC * RSS);
#pragma unroll
    for (int k = 0; k < XIT; k++) {
      const bool rok = r0 + xrw[k] < r_end;
      const f4u v = xr_[k];
      const float e[4] = {v.x, v.y, v.z, v.w};
      float o[4];
#pragma unroll
      for (int c = 0; c < 4; c++) o[c] = !rok ? 0.0f : (xkeep[k][c] ? e[c] : xfill[k][c]);
      if (EM_KC * QR % NT == 0 || tid + k * NT < EM_KC * QR) *(float4*)(&Xs[xlds[k]]) = make_float4(o[0], o[1], o[2], o[3]);
    }
#pragma unroll
    for (int k = 0; k < AIT; k++)
      if (tid + k * NT < EM_KC * NO) Rs[alds[k]] = ar_[k];
  };

  __builtin_amdgcn_s_setprio(MM_PRIO);
  if (r_begin < r_end) load_chunk(r_begin);
  if (DB && r_begin < r_end) {
    store_chunk(r_begin, 0);
    if (r_begin + EM_KC < r_end) load_chunk(r_begin + EM_KC);
  }
  uint32_t cur = 0;
  for (uint64_t r0 = r_begin; r0 < r_end; r0 += EM_KC, cur ^= (DB ? 1u : 0u)) {
    if (!DB) store_chunk(r0, 0);
    __syncthreads();
    if (DB) {
      // the next chunk into the other image pair (its loads were requested a chunk ago), then the loads after that
#if EM_ABL != 2
      if (r0 + EM_KC < r_end) store_chunk(r0 + EM_KC, cur ^ 1u);
      if (r0 + 2 * (uint64_t)EM_KC < r_end) load_chunk(r0 + 2 * (uint64_t)EM_KC);
#endif
    } else if (r0 + EM_KC < r_end) {
      load_chunk(r0 + EM_KC);
    }
    const double* Rs = (const double*)(em_smem + cur * IMG);
    const float* Xs = (const float*)(em_smem + cur * IMG + sizeof(double) * EM_KC * RSS);
    __builtin_amdgcn_s_setprio(0);
    if (F32) {
      v4f32 c32[MT][3];
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < 3; n++) c32[m][n] = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < EM_KC / 4; ks++) {
        float a[MT], b[3];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = (float)Rs[(ks * 4 + lk) * RSS + wo + m * 16 + li];
#pragma unroll
        for (int n = 0; n < 3; n++) b[n] = Xs[(ks * 4 + lk) * XS + wf + n * 16 + li];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int n = 0; n < 3; n++) c32[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], c32[m][n], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < 3; n++)
#pragma unroll
          for (int r = 0; r < 4; r++) acc[m][n][r] += (double)c32[m][n][r];
    } else if (EM_ABL != 1) {
#pragma unroll
      for (int ks = 0; ks < EM_KC / 4; ks++) {
        double a[MT], b[3];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = Rs[(ks * 4 + lk) * RSS + wo + m * 16 + li];
#pragma unroll
        for (int n = 0; n < 3; n++) b[n] = (double)Xs[(ks * 4 + lk) * XS + wf + n * 16 + li];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
          for (int n = 0; n < 3; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(MM_PRIO);
    if (!DB) __syncthreads();
  }
  // the bias column was staged as float: rescale if the bias value is not exactly representable
  const double bfix = (bias_exact || !sp.use_bias) ? 1.0 : sp.bias / (double)bias;
#pragma unroll
  for (int n = 0; n < 3; n++) {
    const uint32_t col = fb + wf + n * 16 + li;
    if (col >= nfun) continue;
    const double sc = (col == nfe) ? bfix : 1.0;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t o = o0 + wo + m * 16 + (F32 ? 4 * lk + r : lk + 4 * r);
        if (o < n_out) slab[((uint64_t)bz * n_out + o) * nfun + col] = acc[m][n][r] * sc;
      }
  }
}

// ------------------------------------------------------------------------------------------
// k_expf_mfma_ws: the wide f64 form (8 x 48 feature columns, 64 outputs per workgroup) with the roles split between
// wavefronts -- ablations of the single-role kernel at the TIMIT transition counts: 13.1 ms complete, 10.0 ms with the
// staging of all but the first chunks removed, 4.0 ms with the MFMA loop removed: staging and matrix work ran back to
// back.  Here wavefronts 0-7 only run the MFMA k-loop over the current LDS image pair while wavefronts 8-11 (one per
// SIMD: a workgroup is 12 wavefronts, three per SIMD, 168 VGPRs each) fetch the chunk after the next from memory and
// store the next one into the other image pair: 192 of their threads own one 16-byte column of the X tile and 16 of its
// rows each (column masks are per-thread constants), 64 own two columns of the posterior tile.  One barrier per chunk.
// MT: M-tiles that hold outputs (the remainder launch carries fewer).
#define EW_KC 32
#define EW_NO 64
#define EW_NF 384
#define EW_XS (EW_NF + 16)
#define EW_RS 80   // em_rss(EW_NO)
#define EW_IMG (sizeof(double) * EW_KC * EW_RS + sizeof(float) * EW_KC * EW_XS)
template <int HAS_XROW, int MT>
__global__ __launch_bounds__(768) void k_expf_mfma_ws(const double* __restrict__ A, uint32_t n_out,
                                                     const float* __restrict__ X, uint32_t F,
                                                     const uint64_t* __restrict__ xrow, uint64_t n_rows,
                                                     ScrfLayout lay, ScrfGemmSpec sp, uint64_t rows_per_chunk,
                                                     double* __restrict__ slab, uint32_t o_base, uint32_t gx, uint32_t gy, uint32_t o_step) {
  extern __shared__ __attribute__((aligned(16))) unsigned char em_smem[];
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t fs = sp.fs, nfe = sp.nfe, nfun = sp.nfun();
  const float bias = sp.use_bias ? (float)sp.bias : 0.0f;
  const bool bias_exact = (double)bias == sp.bias;
  const uint32_t swz = xcd_swizzle(blockIdx.x, gridDim.x);
  const uint32_t bx = swz % gx, by = (swz / gx) % gy, bz = swz / (gx * gy);
  const uint32_t fb = bx * EW_NF;
  const uint32_t o0 = o_base + by * o_step;
  const uint64_t r_begin = (uint64_t)bz * rows_per_chunk;
  const uint64_t r_end = min(n_rows, r_begin + rows_per_chunk);
  const uint32_t n_it = r_begin < r_end ? (uint32_t)((r_end - r_begin + EW_KC - 1) / EW_KC) : 0u;

  if (wave >= 8) {
    // ---------------- producers ----------------
    __builtin_amdgcn_s_setprio(2);
    const uint32_t pt = tid - 512;
    if (pt < 192) {
      const uint32_t q = pt % 96, rg = pt / 96;   // 16-byte column of the X tile, rows rg, rg + 2, ...
      const uint32_t col = fb + q * 4;
      const uint32_t xcl = col < nfe ? col : (nfe >= 4 ? nfe - 4 : 0);   // columns outside the range re-read a valid quad (masked below)
      bool keep[4];
      float fill[4];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        keep[c] = col + c < nfe;
        fill[c] = (col + c == nfe && sp.use_bias) ? bias : 0.0f;
      }
      const float* xb = X + fs + xcl;
      f4u v[16];
      auto load = [&](uint64_t r0) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          uint64_t row = r0 + rg + 2 * k;
          row = row < r_end ? row : r_end - 1;
          const uint64_t xr = HAS_XROW ? xrow[row] : row;
          v[k] = *(const f4u*)(xb + xr * F);
        }
      };
      auto store = [&](uint64_t r0, uint32_t buf) {
        float* Xs = (float*)(em_smem + buf * EW_IMG + sizeof(double) * EW_KC * EW_RS);
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const bool rok = r0 + rg + 2 * k < r_end;
          const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
          float o[4];
#pragma unroll
          for (int c = 0; c < 4; c++) o[c] = !rok ? 0.0f : (keep[c] ? e[c] : fill[c]);
          *(float4*)(&Xs[(rg + 2 * k) * EW_XS + q * 4]) = make_float4(o[0], o[1], o[2], o[3]);
        }
      };
      if (n_it) { load(r_begin); store(r_begin, 0); }
      if (n_it > 1) load(r_begin + EW_KC);
      __syncthreads();
      for (uint32_t i = 0; i < n_it; i++) {
        if (i + 1 < n_it) store(r_begin + (uint64_t)(i + 1) * EW_KC, (i + 1) & 1u);
        if (i + 2 < n_it) load(r_begin + (uint64_t)(i + 2) * EW_KC);
        __syncthreads();
      }
    } else {
      const uint32_t at = pt - 192;               // 0..63: output columns 2c, 2c + 1 of the posterior tile, rows rg, rg + 2, ...
      const uint32_t c2 = at % 32, rg = at / 32;
      const uint32_t oa = o0 + 2 * c2, ob = oa + 1;
      const bool oka = oa < n_out, okb = ob < n_out;
      const double* pa = A + (oka ? oa : n_out - 1);
      const double* pb = A + (okb ? ob : n_out - 1);
      double va[16], vb[16];
      auto load = [&](uint64_t r0) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
          uint64_t row = r0 + rg + 2 * k;
          row = row < r_end ? row : r_end - 1;
          va[k] = pa[row * n_out];
          vb[k] = pb[row * n_out];
        }
      };
      auto store = [&](uint64_t r0, uint32_t buf) {
        double* Rs = (double*)(em_smem + buf * EW_IMG);
#pragma unroll
        for (int k = 0; k < 16; k++) {
          const bool rok = r0 + rg + 2 * k < r_end;
          *(double2*)(&Rs[(rg + 2 * k) * EW_RS + 2 * c2]) = make_double2(rok && oka ? va[k] : 0.0, rok && okb ? vb[k] : 0.0);
        }
      };
      if (n_it) { load(r_begin); store(r_begin, 0); }
      if (n_it > 1) load(r_begin + EW_KC);
      __syncthreads();
      for (uint32_t i = 0; i < n_it; i++) {
        if (i + 1 < n_it) store(r_begin + (uint64_t)(i + 1) * EW_KC, (i + 1) & 1u);
        if (i + 2 < n_it) load(r_begin + (uint64_t)(i + 2) * EW_KC);
        __syncthreads();
      }
    }
    return;
  }
  // ---------------- consumers ----------------
  const uint32_t wf = wave * 48;
  v4f64 acc[MT][3];
#pragma unroll
  for (int m = 0; m < MT; m++)
#pragma unroll
    for (int n = 0; n < 3; n++) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  for (uint32_t i = 0; i < n_it; i++) {
    const double* Rs = (const double*)(em_smem + (i & 1u) * EW_IMG);
    const float* Xs = (const float*)(em_smem + (i & 1u) * EW_IMG + sizeof(double) * EW_KC * EW_RS);
#pragma unroll
    for (int ks = 0; ks < EW_KC / 4; ks++) {
      double a[MT], b[3];
#pragma unroll
      for (int m = 0; m < MT; m++) a[m] = Rs[(ks * 4 + lk) * EW_RS + m * 16 + li];
#pragma unroll
      for (int n = 0; n < 3; n++) b[n] = (double)Xs[(ks * 4 + lk) * EW_XS + wf + n * 16 + li];
#pragma unroll
      for (int m = 0; m < MT; m++)
#pragma unroll
        for (int n = 0; n < 3; n++) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[m][n], 0, 0, 0);
    }
    __syncthreads();
  }
  const double bfix = (bias_exact || !sp.use_bias) ? 1.0 : sp.bias / (double)bias;
#pragma unroll
  for (int n = 0; n < 3; n++) {
    const uint32_t col = fb + wf + n * 16 + li;
    if (col >= nfun) continue;
    const double sc = (col == nfe) ? bfix : 1.0;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t o = o0 + m * 16 + lk + 4 * r;
        if (o < n_out) slab[((uint64_t)bz * n_out + o) * nfun + col] = acc[m][n][r] * sc;
      }
  }
}

template <int HAS_XROW, int NW, int KC, int F32, int MT>
static void launch_expf_mfma_one(hipStream_t st, dim3 grid, size_t sm, const double* A, uint32_t n_out, const float* X, uint32_t F,
                                 const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                                 uint64_t rows_per_chunk, double* slab, uint32_t o_base, uint32_t o_step = 16 * EM_MTW(F32, NW)) {
  constexpr int DB = NW == 8 ? 1 : 0;   // the 8-wave workgroup has its CU to itself: room for a second image pair
  static const bool db_off = getenv("SCRF_EXPF_DB") && atoi(getenv("SCRF_EXPF_DB")) == 0;   // A/B knob
  constexpr int MTW = EM_MTW(F32, NW);
  static const bool ws_off = getenv("SCRF_EXPF_MFMA_WS") && atoi(getenv("SCRF_EXPF_MFMA_WS")) == 0;   // A/B knob
  if (NW == 8 && !F32 && KC == EW_KC && MTW * 16 == EW_NO && !ws_off) {
    constexpr int MTC = MT > 4 ? 4 : MT;
    hipFuncSetAttribute((const void*)k_expf_mfma_ws<HAS_XROW, MTC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * EW_IMG));
    hipLaunchKernelGGL((k_expf_mfma_ws<HAS_XROW, MTC>), dim3(grid.x * grid.y * grid.z), dim3(768), 2 * EW_IMG, st, A, n_out, X, F, xrow, n_rows, lay, sp,
                       rows_per_chunk, slab, o_base, grid.x, grid.y, o_step);
    return;
  }
  if (DB && !db_off) {
    hipFuncSetAttribute((const void*)k_expf_mfma<HAS_XROW, NW, KC, F32, 0, MT, DB, MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * sm));
    hipLaunchKernelGGL((k_expf_mfma<HAS_XROW, NW, KC, F32, 0, MT, DB, MTW>), dim3(grid.x * grid.y * grid.z), dim3(64 * NW), 2 * sm, st, A, n_out, X, F, xrow,
                       n_rows, lay, sp, rows_per_chunk, slab, o_base, grid.x, grid.y, o_step);
    return;
  }
  hipFuncSetAttribute((const void*)k_expf_mfma<HAS_XROW, NW, KC, F32, 0, MT, 0, MTW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL((k_expf_mfma<HAS_XROW, NW, KC, F32, 0, MT, 0, MTW>), dim3(grid.x * grid.y * grid.z), dim3(64 * NW), sm, st, A, n_out, X, F, xrow,
                     n_rows, lay, sp, rows_per_chunk, slab, o_base, grid.x, grid.y, o_step);
}
// (a mixed tiling of the outputs -- 200 = 64 + 3 x 48 instead of 3 x 64 + a thin 8-output launch -- was measured at config 5:
// 81.8 against 80.0 ms; the thin launch costs less than the thirteenth M-tile)
template <int HAS_XROW, int NW, int KC, int F32>
static void launch_expf_mfma_x(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                               const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                               uint64_t rows_per_chunk, uint32_t n_chunks, double* slab) {
  const uint32_t nfun = sp.nfun();
  const uint32_t gx = (nfun + 48 * NW - 1) / (48 * NW);
  constexpr int MTW = EM_MTW(F32, NW);
  constexpr uint32_t NO = 16 * MTW;
  const size_t sm = sizeof(double) * KC * em_rss((int)NO) + sizeof(float) * KC * (48 * NW + 16);
  const uint32_t n_full = n_out / NO, rem = n_out % NO;
  if (n_full)
    launch_expf_mfma_one<HAS_XROW, NW, KC, F32, MTW>(st, dim3(gx, n_full, n_chunks), sm, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, slab, 0u);
  // the outputs past the last full tile: workgroups that carry only the M-tiles holding outputs
  if (MTW > 3 && rem > 48)
    launch_expf_mfma_one<HAS_XROW, NW, KC, F32, MTW>(st, dim3(gx, 1, n_chunks), sm, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, slab, n_full * NO);
  else if (rem > 32)
    launch_expf_mfma_one<HAS_XROW, NW, KC, F32, 3>(st, dim3(gx, 1, n_chunks), sm, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, slab, n_full * NO);
  else if (rem > 16)
    launch_expf_mfma_one<HAS_XROW, NW, KC, F32, 2>(st, dim3(gx, 1, n_chunks), sm, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, slab, n_full * NO);
  else if (rem > 0)
    launch_expf_mfma_one<HAS_XROW, NW, KC, F32, 1>(st, dim3(gx, 1, n_chunks), sm, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, slab, n_full * NO);
}
template <int NW, int KC, int F32>
static void launch_expf_mfma_nw(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                                const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                                uint64_t rows_per_chunk, uint32_t n_chunks, double* slab) {
  if (xrow) launch_expf_mfma_x<1, NW, KC, F32>(st, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, n_chunks, slab);
  else launch_expf_mfma_x<0, NW, KC, F32>(st, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, n_chunks, slab);
}

// few feature functions (<= 48) and many outputs: 4 wavefronts share the feature tile and split 192 outputs
template <int F32>
static void launch_expf_mfma_split(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                                   const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                                   uint64_t rows_per_chunk, uint32_t n_chunks, double* slab) {
  constexpr int NW = 4, KC = 16;
  dim3 grid(1, (n_out + EM_SPLIT_NO * NW - 1) / (EM_SPLIT_NO * NW), n_chunks);
  const size_t sm = sizeof(double) * KC * em_rss(EM_SPLIT_NO * NW) + sizeof(float) * KC * (48 + 16);
  if (xrow)
    hipLaunchKernelGGL((k_expf_mfma<1, NW, KC, F32, 1>), dim3(grid.x * grid.y * grid.z), dim3(64 * NW), sm, st, A, n_out, X, F, xrow, n_rows, lay, sp,
                       rows_per_chunk, slab, 0u, grid.x, grid.y, (uint32_t)(EM_SPLIT_NO * NW));
  else
    hipLaunchKernelGGL((k_expf_mfma<0, NW, KC, F32, 1>), dim3(grid.x * grid.y * grid.z), dim3(64 * NW), sm, st, A, n_out, X, F, xrow, n_rows, lay, sp,
                       rows_per_chunk, slab, 0u, grid.x, grid.y, (uint32_t)(EM_SPLIT_NO * NW));
}
uint32_t expf_mfma_wide_tiles(uint32_t n_out, uint32_t nfun, int f32) {
  const uint32_t tiles = (nfun + 47) / 48;
  if (tiles <= 4) return 0;   // split and narrow forms: several workgroups per CU, no round structure to fit
  const uint32_t gx = (nfun + 48 * 8 - 1) / (48 * 8);
  const uint32_t NO = 16 * (uint32_t)EM_MTW(f32, 8);
  return gx * ((n_out + NO - 1) / NO);
}

void launch_expf_mfma(hipStream_t st, const double* A, uint32_t n_out, const float* X, uint32_t F,
                      const uint64_t* xrow, uint64_t n_rows, const ScrfLayout& lay, const ScrfGemmSpec& sp,
                      uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, int f32) {
  if (n_rows == 0 || n_chunks == 0) return;
  const uint32_t nfun = sp.nfun();
  const uint32_t tiles = (nfun + 47) / 48;  // 48-column wave tiles needed
#define EXPF_ARGS st, A, n_out, X, F, xrow, n_rows, lay, sp, rows_per_chunk, n_chunks, slab
  if (tiles <= 1 && n_out >= 4 * EM_SPLIT_NO) {
    if (f32) launch_expf_mfma_split<1>(EXPF_ARGS);
    else launch_expf_mfma_split<0>(EXPF_ARGS);
    return;
  }
  if (f32) {
    if (tiles <= 1) launch_expf_mfma_nw<1, 32, 1>(EXPF_ARGS);
    else if (tiles <= 2) launch_expf_mfma_nw<2, 64, 1>(EXPF_ARGS);
    else if (tiles <= 3) launch_expf_mfma_nw<3, 64, 1>(EXPF_ARGS);
    else if (tiles <= 4) launch_expf_mfma_nw<4, 64, 1>(EXPF_ARGS);
    else launch_expf_mfma_nw<8, 32, 1>(EXPF_ARGS);
  } else {
    if (tiles <= 1) launch_expf_mfma_nw<1, 32, 0>(EXPF_ARGS);
    else if (tiles <= 2) launch_expf_mfma_nw<2, 64, 0>(EXPF_ARGS);
    else if (tiles <= 3) launch_expf_mfma_nw<3, 64, 0>(EXPF_ARGS);
    else if (tiles <= 4) launch_expf_mfma_nw<4, 64, 0>(EXPF_ARGS);
    else {
      // (64-row chunks for the 8-wave form were tried: 256 VGPRs + 336 bytes of spills)
      // (two 4-wave workgroups per CU with 64-row chunks instead: 18.1 ms against 14.4 at the TIMIT transition counts)
      launch_expf_mfma_nw<8, 32, 0>(EXPF_ARGS);
    }
  }
#undef EXPF_ARGS
}
