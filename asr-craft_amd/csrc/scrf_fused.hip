// scrf_fused.hip -- the two state contractions with the segment-window synthesis fused in.
//
// Why: the window image X is 9.7 MB per config-2 utterance (7200 windows x 337 floats) against
// 47 KB of raw frames.  Materialising it costs 12.5 ms per 4096 utterances to write, as much again
// per contraction to read back, and 337-deep fp64 MFMA contractions of 12 ms each.  The recipe
// (io/CRF_InFtrStream_SeqMultiWindow.cpp:556-884) has structure:
//
//   x(t,d) = [ F[b+s_0(d)] .. F[b+s_4(d)] | avg | max | min | onehot(d) ],  b = t-d+1,
//            s_k(d) = ceil(float(0.1 d) * (2k+1)) - 1
//
// * the five sampled blocks are COPIES of raw frames, so their share of the score is a sum of
//   per-frame projections P[f][k][o] = F[f] . W_k[o] (one small MFMA contraction over frames), and
//   their share of the expected counts is Z_k^T F with Z_k[f][o] = sum of R over the windows whose
//   k-th sample is frame f.  Both are exact re-associations of the same fp64 products.
// * avg / max / min are rebuilt per row tile in LDS from the raw frames with the reference's float
//   arithmetic (running sum from the last frame backwards, divided by the length) and contracted
//   on the MFMA there; the one-hot duration block and the bias are an epilogue add in the score
//   kernel and two more column groups of the expected-count kernel.
//
// X never exists in HBM and the dense depth drops from 8W+D to 3W (scores) / 3W+D+1 (counts).
//
// Kernels: k_pframe (P = F W_k), k_scores_fused (dense part + gather of P + exp epilogue),
// k_post_z (R = Y - gamma and Z in one walk; k_lin_z is the Z-only form), k_expf_fused (dense part
// of the expected counts), k_ztf (Z^T F).  DESIGN.md 4.1 has the data flow.
#include "scrf_dp_common.h"

#include <math.h>
#include <string.h>
#include <utility>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));

#define FU_GC 40        // columns per dense chunk of the score kernel
#define FU_XS 42        // float row stride of the chunk image (== 2 mod 4: conflict-free A fragments)
#define FU_WS 49        // row stride of the transposed lambda image
#define FU_ROWS SCRF_FUSED_ROWS_SCORES
#define FE_ROWS SCRF_FUSED_ROWS_EXPF
#define FE_RSF 80       // float row stride of the R image in the f32 form: == 16 (mod 32), the b32 counterpart
#define FE_RS 48        // double row stride of the R image: 96 dwords = 32 (mod 64), so the two k-rows a 32-lane half of a
                        // ds_read_b64 A-fragment covers fall on disjoint banks (50 overlapped four of them)

__host__ __device__ inline uint32_t fu_sample_step(uint32_t d, int k) {
  const float ot = (float)((double)d * 0.1);
  return (uint32_t)ceilf(ot * (float)(2 * k + 1)) - 1u;
}
__device__ __forceinline__ uint32_t fu_div(uint32_t i, uint32_t magic) { return magic ? __umulhi(i, magic) : i; }   // magic 0: d == 1 (2^32 does not fit)
__device__ __forceinline__ uint32_t fu_magic(uint32_t d) { return d < 2 ? 0u : 0xffffffffu / d + 1u; }   // ceil(2^32 / d) in 32-bit arithmetic
// FU_PROF (debug builds only): wave 0 of every workgroup stamps the phase boundaries with s_memtime and adds the
// differences to fu_prof[phase]; fu_prof[15] counts tiles.  Read with scrf_debug_fused_prof (tools/fused_phases.py).
#ifndef FU_PROF
#define FU_PROF 0
#endif
#if FU_PROF
__device__ unsigned long long fu_prof[16];
#define FU_STAMP(i)                                                                                      \
  do {                                                                                                   \
    if (threadIdx.x == 0) {                                                                              \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                      \
      atomicAdd(&fu_prof[i], now_ - stamp_);                                                             \
      stamp_ = now_;                                                                                     \
    }                                                                                                    \
  } while (0)
extern "C" int scrf_debug_fused_prof(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(fu_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(fu_prof), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define FU_STAMP(i) do {} while (0)
#endif

// A row tile (ScrfTileDesc, built on the host with the batch): rows [r0, r0+nrows) of an utterance,
// touching frames t0 .. t0+nfr-1; raw frames are staged from f0 = t0 - back.
struct FuTile {
  uint32_t t0, nfr, nrows, f0, r0;
  uint64_t row0;  // chunk-relative row of the tile's first window
  uint64_t fr0;   // chunk-relative index of frame f0
};
__device__ __forceinline__ FuTile fu_tile(const ScrfFusedArgs& fa, const ScrfTileDesc& q) {
  FuTile ft;
  ft.t0 = q.t0; ft.nfr = q.nfr; ft.nrows = q.nrows; ft.f0 = q.t0 - q.back; ft.r0 = q.r0;
  ft.row0 = q.row_abs - fa.row_base;
  ft.fr0 = q.fr_abs - fa.frame_base;
  return ft;
}

// exp(x) for x <~ 0 (Cody-Waite + degree-13 Horner, ~1 ulp; flushes to 0 below -745).  The
// coefficients live in constant memory so that they are read into scalar registers once instead of
// being re-materialised as 64-bit literals (two v_mov each) at every use.
__constant__ double FU_EXPC[18] = {
    1.4426950408889634, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
    1.6059043836821613e-10, 2.0876756987868100e-09, 2.5052108385441720e-08, 2.7557319223985888e-07,
    2.7557319223985893e-06, 2.4801587301587302e-05, 1.9841269841269841e-04, 1.3888888888888889e-03,
    8.3333333333333332e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5, 1.0, 1.0, -1000.0};
struct FuExpC { double c[18]; };
__device__ __forceinline__ FuExpC fu_exp_consts() {
  FuExpC k;
#pragma unroll
  for (int i = 0; i < 18; i++) k.c[i] = FU_EXPC[i];
  return k;
}
__device__ __forceinline__ double fu_exp(double x, const FuExpC& k) {
  x = fmax(x, k.c[17]);
  const double n = rint(x * k.c[0]);
  double r = fma(n, k.c[1], x);
  r = fma(n, k.c[2], r);
  double p = k.c[3];
#pragma unroll
  for (int i = 4; i <= 16; i++) p = fma(p, r, k.c[i]);
  return ldexp(p, (int)n);
}

// The same through a 256-entry table of 2^(j/256) in LDS: x = (256 e + j) ln2/256 + r, |r| <= ln2/512, so a degree-4
// polynomial is exact to 4e-17 and the whole thing is 14 vector instructions and one LDS read instead of 21.  The
// reduction uses ln2/256 as one constant: its rounding error times n is below 1e-14 for every x above -100 (what lies
// further down is below e^-100 of the row maximum).
#define FU_EXPT_N 256
__device__ __forceinline__ double fu_exp_tab(double x, const double* tab) {
  x = fmax(x, -1000.0);
  const double n = rint(x * 369.32993046756268);             // 256 / ln 2
  const double r = fma(n, -2.7076061740622863e-3, x);        // ln 2 / 256
  const int ni = (int)n;
  const double t = tab[ni & (FU_EXPT_N - 1)];
  double q = fma(r, 4.1666666666666664e-02, 1.6666666666666666e-01);
  q = fma(r, q, 0.5);
  q = fma(r, q, 1.0);
  return ldexp(fma(t * r, q, t), ni >> 8);
}

// Window statistics of one (frame, column): the values F[t-j][c], j = 0 .. nd-1, are read from LDS
// in one batch (no load sits inside the dependent chain), then walked with compile-time durations.
// The window average is sum / d in float: q0 = sum * y, r = fma(-q0, d, sum), q = fma(r, y, q0) with
// y the correctly rounded 1/d (a compile-time constant here) is the correctly rounded quotient
// (Markstein), i.e. the same float as the reference's division, at a third of the instructions.
// Results for durations outside [d_lo, d_hi] go to `dump` (a spare LDS row) instead of a branch.
template <int DMAX>
__device__ __forceinline__ void fu_load_vals(const float* last, uint32_t W, uint32_t nd, float (&v)[DMAX]) {
#pragma unroll
  for (int j = 0; j < DMAX; j++) v[j] = *(last - min((uint32_t)j, nd - 1) * W);
}
template <int DMAX>
__device__ __forceinline__ void fu_scan_avg(const float (&v)[DMAX], float* o, uint32_t stride, uint32_t d_lo,
                                            uint32_t d_hi, float* dump) {
  float a = 0.0f;
#pragma unroll
  for (int j = 0; j < DMAX; j++) {
    a = __fadd_rn(a, v[j]);
    const float df = (float)(j + 1), y = 1.0f / df;
    const float q0 = __fmul_rn(a, y);
    const float q = __fmaf_rn(__fmaf_rn(-q0, df, a), y, q0);
    float* w = ((uint32_t)(j + 1) >= d_lo && (uint32_t)(j + 1) <= d_hi) ? o + j * stride : dump;
    *w = q;
  }
}
// running extremum as one instruction: fmaxf / fminf make the compiler canonicalise both operands first (values
// loaded from memory are not known to be quiet), tripling the count.  For finite data v_max_f32 / v_min_f32 give
// what the reference's `if (v > a) a = v` gives.
__device__ __forceinline__ float fu_vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fu_vmaxd(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float fu_vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int DMAX, int IS_MAX>
__device__ __forceinline__ void fu_scan_ext(const float (&v)[DMAX], float* o, uint32_t stride, uint32_t d_lo,
                                            uint32_t d_hi, float* dump) {
  float a = v[0];
#pragma unroll
  for (int j = 0; j < DMAX; j++) {
    a = IS_MAX ? fu_vmax(a, v[j]) : fu_vmin(a, v[j]);
    float* w = ((uint32_t)(j + 1) >= d_lo && (uint32_t)(j + 1) <= d_hi) ? o + j * stride : dump;
    *w = a;
  }
}
// Steady-state forms (every duration 1 .. DMAX exists and lies inside the tile): no clamped loads, no select per
// store -- the LDS offsets are immediates.
template <int DMAX>
__device__ __forceinline__ void fu_load_vals_full(const float* img, uint32_t at, uint32_t W, float (&v)[DMAX]) {
  // a running index into the LDS image (one subtraction per load; j * W would be a quarter-rate integer multiply
  // each).  The opaque value is the INDEX, not the pointer: an opaque pointer loses its address space and the loads
  // become flat loads.
  // byte offsets in 32 bits: for an image in memory the loads then take the scalar-base + 32-bit-lane-offset form
  uint32_t atb = at * 4u;
  const uint32_t wb = W * 4u;
#pragma unroll
  for (int j = 0; j < DMAX; j++) {
    v[j] = *(const float*)((const char*)img + atb);
    atb -= wb;
    asm volatile("" : "+v"(atb));
  }
}
template <int DMAX, uint32_t STRIDE>
__device__ __forceinline__ void fu_scan_avg_full(const float (&v)[DMAX], float* o) {
  float a = 0.0f;
#pragma unroll
  for (int j = 0; j < DMAX; j++) {
    a = __fadd_rn(a, v[j]);
    const float df = (float)(j + 1), y = 1.0f / df;
    const float q0 = __fmul_rn(a, y);
    o[j * STRIDE] = __fmaf_rn(__fmaf_rn(-q0, df, a), y, q0);
  }
}
template <int DMAX, int IS_MAX, uint32_t STRIDE>
__device__ __forceinline__ void fu_scan_ext_full(const float (&v)[DMAX], float* o) {
  float a = v[0];
#pragma unroll
  for (int j = 0; j < DMAX; j++) {
    a = IS_MAX ? fu_vmax(a, v[j]) : fu_vmin(a, v[j]);
    o[j * STRIDE] = a;
  }
}

// ------------------------------------------------------------------------------------------
// k_scores_fused: S[row][o] = dense(avg|max|min) + sum_k P[b+s_k(d)][k][o] + lambda_dur[d][o] + bias.
// Workgroup (512 threads) = the windows of TB = 256/D whole frames (<= 256 rows); wave w owns rows
// [32w, 32w+32) x 48 outputs (2 x 3 MFMA tiles).  The three dense groups are rebuilt 40 columns at
// a time (one thread per (frame, column)); the next chunk's lambda slice is fetched under the MFMAs.
// After the last chunk the tile's rows of P are staged over the dead operand images for the gather.
// ------------------------------------------------------------------------------------------
#define FU_NT 512
// experiment switches (defaults = the shipped configuration)
#ifndef FU_PHI
#define FU_PHI 1        // 1: a lane's outputs are the pairs 8j + 2lk + {0,1}: the 4 lanes of a row store 64 contiguous bytes
#endif                  //    per instruction; 0: 12 consecutive outputs per lane (96-byte pieces 96 bytes apart)
#ifndef FU_EXPTAB
#define FU_EXPTAB 0     // 1: table-based exp in the epilogue (14 instructions + a gathered LDS read instead of 21; measured
                        // equal -- the epilogue is not instruction-bound -- and the table costs 2 KB of LDS and bank conflicts)
#endif
#ifndef FU_FULLSCAN
#define FU_FULLSCAN 1   // steady-state scan without clamps and dumps
#endif
#ifndef FU_PRIO
#define FU_PRIO 1       // waves in their vector-only phases (staging, scans, epilogue) outrank waves inside an MFMA loop
#endif
#ifndef FU_PRIO_HI
#define FU_PRIO_HI 1
#endif
#if FU_PRIO
#ifdef FU_PRIO_INV      // experiment: the other way round
#define FU_SETPRIO(p) __builtin_amdgcn_s_setprio((p) ? 0 : FU_PRIO_HI)
#else
#define FU_SETPRIO(p) __builtin_amdgcn_s_setprio((p) ? FU_PRIO_HI : 0)
#endif
#else
#define FU_SETPRIO(p) do {} while (0)
#endif
#ifndef FU_ABL
#define FU_ABL 0        // ablations (wrong results): 1 = no global loads in the P staging, 2 = nor in the frame staging;
                        // count kernel: 3 = no consumer MFMAs, 4 = no producer builds, 5 = no R loads, 6 = no window scans
#endif
#define FU_DS 50        // double row stride of the P image and of the duration-weight table: 25 16-byte slots, odd, so
                        // that the rows of consecutive frames / durations a ds_read_b128 lane group gathers start on
                        // distinct slots (48 would put them on two)
typedef double v2f64 __attribute__((ext_vector_type(2)));
typedef float v2f32 __attribute__((ext_vector_type(2)));
// smax != nullptr (n_out <= 48 only): the epilogue writes exp(S - smax[row]) instead of S, with
// smax[row] the float-rounded row maximum, and the labelled windows' scores to s_true -- the inputs
// of the linear-domain recursion (scrf_dplin.hip), saving a read-modify-write pass over S.
// DEC (decode, F32 == 0 only): write float(-1 * score) and the list of entries to recompute
// (ScrfDecodeOut, scrf_common.h) instead of S.
//
// The MFMAs compute the TRANSPOSED tile (A = lambda^T, B = X^T): a lane then holds 12 outputs of ONE window row
// (o = 12 lk + 4 n + r) instead of 3 outputs of 8 rows, so everything the epilogue does per row -- the five gather
// offsets, the duration weights, the row maximum, the label test -- is paid once per 12 outputs, the gathers and the
// stores are 16-byte accesses, and the row maximum needs two lane swaps.
// rows of the staged P image: block k holds the frames sample position k can reach, TB + off_k(D) of them
__host__ __device__ inline uint32_t fu_p_rows(uint32_t D, uint32_t TB) {
  uint32_t n = 0;
  for (int k = 0; k < 5; k++) n += TB + (D - 1 - fu_sample_step(D, k));
  return n;
}
#define FU_NPQ 12       // P rows per thread held in registers between their loads and their LDS stores (slot + 10 q)
// LA (linear window average, SCRF_PREC_FASTLIN): the avg block leaves the dense contraction.  The average is linear in
// the frames, so its share of the score is (C[t] - C[t-d]) / d with C the per-utterance prefix sum of the per-frame
// projection Q[f][o] = F[f] . W_avg[o] (group 5 of k_pframe, summed in place by k_avg_prefix).  The reference rounds the
// running float sum and the quotient to float (io/CRF_InFtrStream_SeqMultiWindow.cpp:609-646); this form does not, which
// makes it a precision tier of its own (measured 3e-8 relative on the gradient against the reference arithmetic; contract 1e-4).
// Block 5 of the staged P image holds C for the frames t0 - D .. t0 + nfr - 1 (zero rows before the utterance).
__host__ __device__ inline uint32_t fu_p_rows_la(uint32_t D, uint32_t TB) { return fu_p_rows(D, TB) + TB + D; }
#define FU_NPQ_LA 15

template <int DMAX, int F32, int DEC, int LA>
__global__ __launch_bounds__(FU_NT, 4) void k_scores_fused(ScrfFusedArgs fa, ScrfLayout lay,
                                                           const double* __restrict__ lambda,
                                                           const double* __restrict__ P, uint32_t n_out,
                                                           double* __restrict__ S, double* __restrict__ smax,
                                                           double* __restrict__ s_true,
                                                           const uint32_t* __restrict__ labels, ScrfDecodeOut dz) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const uint32_t W = fa.W, D = lay.D;
  const uint32_t nfmax = fa.TB + D - 1;
  // operand images (live until the last MFMA) and, over them, the P rows the epilogue gathers: block k = the frames
  // sample position k can reach (off_k(D) = D - 1 - s_k(D) before the tile's first frame), rows of FU_DS doubles
  float* Xg = (float*)fsm;                                              // [256 + dump row][FU_XS]
  double* Wg = (double*)(fsm + sizeof(float) * (FU_ROWS + 1) * FU_XS);  // [40][FU_WS] (floats when F32)
  float* fr = (float*)(Wg + FU_GC * FU_WS);                             // [nfmax][W]
  double* Pl = (double*)fsm;                                            // [fu_p_rows][FU_DS]: 48 used
  size_t opn = sizeof(float) * (FU_ROWS + 1) * FU_XS + sizeof(double) * FU_GC * FU_WS + sizeof(float) * nfmax * W;
  const uint32_t nprmax = fu_p_rows_la(D, fa.TB);   // the tile size is planned for the larger (LA) image
  const size_t pb = sizeof(double) * nprmax * FU_DS;
  if (pb > opn) opn = pb;
  // outside the union (staged with the raw frames, read by the epilogue):
  double* Dt = (double*)(fsm + ((opn + 15) & ~(size_t)15));             // [D][FU_DS] duration weight + bias term; [.][48] = 1/d
  // per-row record (16 bytes, one ds_read_b128 in the epilogue): the five gather offsets into Pl (in doubles), the
  // duration, the frame inside the tile, and the output whose score is the labelled window's (0xffff: none)
  uint4* recs = (uint4*)(Dt + D * FU_DS);                               // [FU_ROWS]
  uint16_t* rbase = (uint16_t*)(recs + FU_ROWS);                        // [TB] first row of each frame
  uint16_t* rowmap = rbase + ((fa.TB + 3) & ~3u);                       // [fu_p_rows] P image row -> 5 * (frame - f0) + k
#if FU_EXPTAB
  double* etab = (double*)(rowmap + ((nprmax + 3) & ~3u));              // [256] 2^(j/256) (exp epilogue only)
#endif
  const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t o0 = blockIdx.y * 48;
  // the tiles of an utterance are neighbours in the tile list and gather the same P rows and raw frames: with one
  // output block (L <= 48) the XCD-aware order keeps them on one XCD's L2
  const uint32_t tix = gridDim.y == 1 ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  const FuTile ft = fu_tile(fa, fa.tiles[fa.tile0 + tix]);
  float* Wgf = (float*)Wg;
  const uint32_t cpg = (W + FU_GC - 1) / FU_GC;  // chunks per group
  // MFMA row li of output tile n carries output phi(n, li), chosen so that the accumulators a lane ends up with
  // (rows lk + 4r of the f64 tile, 4 lk + r of the f32 tile) are its 12 consecutive outputs 12 lk + 4 n + r
#if FU_PHI
  // o = 16 n + 8 (r >> 1) + 2 lk + (r & 1) for accumulator r of tile n: value c = 4n + r of a lane is output
  // 8 (c >> 1) + 2 lk + (c & 1), so the lane's values come in the adjacent pairs (8j + 2lk, 8j + 2lk + 1), j = 0..5
  // (decode keeps the untransposed product -- rows of the tile on the MFMA's row axis, phi = identity: its epilogue has
  // the rounding screen on top of the gathers and ran out of registers in the 12-outputs-per-lane form)
  const uint32_t phi0 = DEC ? li
                            : (F32 ? 8 * ((li & 3) >> 1) + 2 * (li >> 2) + (li & 1) : 8 * (li >> 3) + 2 * (li & 3) + ((li >> 2) & 1));
#else
  const uint32_t phi0 = F32 ? (li >> 2) * 12 + (li & 3) : (li & 3) * 12 + (li >> 2);
#endif
#if FU_PROF
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) atomicAdd(&fu_prof[15], 1ull);
#endif
  // lambda chunk prefetch registers: element e = tid + 512*q of the [48][40] chunk
  double wp[4];
  auto load_w = [&](uint32_t ci) {
    const uint32_t ty = ci / cpg, c0 = (ci % cpg) * FU_GC;
    const uint32_t nc = min((uint32_t)FU_GC, W - c0);
    const uint32_t woff = (5 + ty) * W + c0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t e = tid + FU_NT * q, ol = e / FU_GC, c = e % FU_GC;
      wp[q] = (e < 48 * FU_GC && o0 + ol < n_out && c < nc) ? lambda[lay.state_idx(o0 + ol) + woff + c] : 0.0;
    }
  };
  FU_SETPRIO(1);
  load_w(LA ? cpg : 0);   // LA: the dense groups are max and min only
#if FU_PROF
  asm volatile("" :: "v"(ft.t0));   // the descriptor has arrived
  FU_STAMP(5);
#endif

  // geometry of the P image: block k starts at image row prow0[k] with frame pf0[k] (relative to f0)
  const uint32_t back = ft.t0 - ft.f0, nf = back + ft.nfr;
  uint32_t prow0[5], pf0[5], nprows = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t reach = D - 1 - fu_sample_step(D, k);
    pf0[k] = back > reach ? back - reach : 0;
    prow0[k] = nprows;
    nprows += nf - pf0[k];
  }
  const uint32_t crow0 = nprows;          // LA: block 5 = the prefix sums C of frames t0 - D .. t0 + nfr - 1
  if (LA) nprows += ft.nfr + D;
  // stage raw frames f0 .. t0+nfr-1 and the duration weights (loads batched ahead of the LDS stores), decode rows
  {
    auto dur_w = [&](uint32_t i) {
      const uint32_t dd = i / 48, oo = i % 48;
      double v = -1e300;   // outputs past n_out: never the row maximum, exp -> 0, masked at the stores
      if (o0 + oo < n_out) {
        const uint32_t base = lay.state_idx(o0 + oo) + 8 * W;
        v = lambda[base + dd];
        if (lay.use_sb) v += lambda[base + D] * lay.sbv;
      }
      return v;
    };
    // fa.dtab (k_dur_table, once per launch): the [D][FU_DS] table of this output block, copied with coalesced loads --
    // built per tile it costs two loads per element from 48 different weight rows (lambda[state(o) + 8W + d] and the bias)
    const double* dtab = fa.dtab ? fa.dtab + (size_t)blockIdx.y * D * FU_DS : nullptr;
    double dtv[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const uint32_t i = tid + FU_NT * q;
      dtv[q] = dtab ? (i < D * FU_DS ? dtab[i] : 0.0) : (i < D * 48 ? dur_w(i) : 0.0);
    }
    const float* src = fa.frames + (fa.frame_base + ft.fr0) * (uint64_t)W;
    const uint32_t n = nf * W;
    for (uint32_t i0 = 0; i0 < n; i0 += 4 * FU_NT) {
      float tmp[4];
#pragma unroll
      for (int q = 0; q < 4; q++) tmp[q] = (FU_ABL < 2 && i0 + tid + FU_NT * q < n) ? src[i0 + tid + FU_NT * q] : 0.5f;
#pragma unroll
      for (int q = 0; q < 4; q++) if (i0 + tid + FU_NT * q < n) fr[i0 + tid + FU_NT * q] = tmp[q];
    }
    if (dtab) {
#pragma unroll
      for (int q = 0; q < 3; q++) if (tid + FU_NT * q < D * FU_DS) Dt[tid + FU_NT * q] = dtv[q];
      for (uint32_t i = tid + 3 * FU_NT; i < D * FU_DS; i += FU_NT) Dt[i] = dtab[i];   // D > 30 only
    } else {
#pragma unroll
      for (int q = 0; q < 3; q++) if (tid + FU_NT * q < D * 48) Dt[((tid + FU_NT * q) / 48) * FU_DS + (tid + FU_NT * q) % 48] = dtv[q];
      for (uint32_t i = tid + 3 * FU_NT; i < D * 48; i += FU_NT) Dt[(i / 48) * FU_DS + i % 48] = dur_w(i);   // D > 32 only
      if (LA && tid < D) Dt[tid * FU_DS + 48] = 1.0 / (double)(tid + 1);
    }
    FU_STAMP(6);   // frames and duration weights in LDS
    // steady-state tiles (every frame has all D durations, TB frames) share their row records, row bases and rowmap:
    // copied from fa.rtab (k_tile_tables, once per launch), only the label slot is the tile's own
    const bool std_tile = fa.rtab && ft.t0 >= D && ft.nfr == fa.TB;
    if (std_tile) {
      const uint4* rg = (const uint4*)fa.rtab;
      const uint16_t* bg = (const uint16_t*)(rg + FU_ROWS);
      const uint16_t* mg = bg + ((fa.TB + 7) & ~7u);
      for (uint32_t i = tid; i < ft.nfr * D; i += FU_NT) {
        uint4 rec = rg[i];
        if (labels) {
          const uint32_t d = (rec.z >> 16) & 0xffu, tl = rec.z >> 24;
          const uint32_t lab = labels[fa.frame_base + ft.fr0 + back + tl];
          const uint32_t rel = lab - (d - 1) * n_out;
          if (lab != SCRF_LAB_BAD && rel < n_out) rec.w = rel;
        }
        recs[i] = rec;
      }
      for (uint32_t tl = tid; tl < ft.nfr; tl += FU_NT) rbase[tl] = bg[tl];
      for (uint32_t r = tid; r < nprows; r += FU_NT) rowmap[r] = mg[r];
    } else {
    const uint32_t mD = fu_magic(D);
    for (uint32_t i = tid; i < ft.nfr * D; i += FU_NT) {
      const uint32_t tl = fu_div(i, mD), d = i - tl * D + 1;
      const uint32_t t = ft.t0 + tl;
      if (d <= scrf_node_max_dur(t, D)) {
        const uint32_t row = (uint32_t)(scrf_seg_base(t, D) - ft.r0) + d - 1;
        const uint32_t b0 = t - d + 1 - ft.f0;
        uint32_t q[5];
#pragma unroll
        for (int k = 0; k < 5; k++) q[k] = (prow0[k] - pf0[k] + b0 + fu_sample_step(d, k)) * FU_DS;
        uint32_t mine_o = 0xffffu;
        if (labels) {
          // label = n_out * (duration - 1) + phone: this row's iff it lies in [n_out (d-1), n_out d)
          const uint32_t lab = labels[fa.frame_base + ft.fr0 + back + tl];
          const uint32_t rel = lab - (d - 1) * n_out;
          if (lab != SCRF_LAB_BAD && rel < n_out) mine_o = rel;
        }
        recs[row] = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (d << 16) | (tl << 24), mine_o);
      }
    }
    FU_STAMP(7);   // row records
    for (uint32_t tl = tid; tl < ft.nfr; tl += FU_NT) rbase[tl] = (uint16_t)(scrf_seg_base(ft.t0 + tl, D) - ft.r0);
    // image row -> NG * (frame - f0 + 1) + k (LA: NG = 6 and frame t0 - D may lie one before f0; 0xffff: a zero row)
    for (uint32_t r = tid; r < nprows; r += FU_NT) {
      if (LA && r >= crow0) {
        const int32_t fq = (int32_t)ft.t0 - (int32_t)D + (int32_t)(r - crow0);   // frame inside the utterance
        rowmap[r] = fq < 0 ? (uint16_t)0xffffu : (uint16_t)(6 * ((uint32_t)fq - ft.f0 + 1) + 5);
        continue;
      }
      uint32_t k = 0;
#pragma unroll
      for (int kk = 1; kk < 5; kk++) k += r >= prow0[kk] ? 1u : 0u;
      uint32_t base = prow0[0], f0k = pf0[0];
#pragma unroll
      for (int kk = 1; kk < 5; kk++) if (k == (uint32_t)kk) { base = prow0[kk]; f0k = pf0[kk]; }
      rowmap[r] = (uint16_t)((LA ? 6 : 5) * (f0k + r - base + 1) + k);
    }
    }
#if FU_EXPTAB
    if (!DEC && smax && tid < FU_EXPT_N) etab[tid] = exp2((double)tid * (1.0 / FU_EXPT_N));
#endif
    // The image region held the previous occupant's P rows (any bit pattern).  What the MFMAs read and the scans do
    // not write must be finite: the pad columns [min(W, FU_GC), FU_GC) of every row (their lambda rows are zero) --
    // rows past the tile are masked at the stores, so their garbage only has to stay in its own row, which a matrix
    // product guarantees; and the whole image before an edge tile, whose scans skip durations.
    {
      const uint32_t ncf = min(W, (uint32_t)FU_GC);
      if (ft.t0 + 1 >= D && ncf < FU_GC) {
        for (uint32_t i = tid; i < FU_ROWS * (FU_GC - ncf); i += FU_NT) Xg[(i / (FU_GC - ncf)) * FU_XS + ncf + i % (FU_GC - ncf)] = 0.0f;
      } else {
        for (uint32_t i = tid; i < (FU_ROWS + 1) * FU_XS; i += FU_NT) Xg[i] = 0.0f;
      }
    }
  }
  __syncthreads();
  FU_STAMP(0);   // stage

  v4f64 acc[F32 ? 1 : 2][F32 ? 1 : 3];
  v4f32 acc32[F32 ? 2 : 1][F32 ? 3 : 1];
#pragma unroll
  for (int m = 0; m < (F32 ? 1 : 2); m++)
#pragma unroll
    for (int n = 0; n < (F32 ? 1 : 3); n++) acc[m][n] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int m = 0; m < (F32 ? 2 : 1); m++)
#pragma unroll
    for (int n = 0; n < (F32 ? 3 : 1); n++) acc32[m][n] = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};

  // P rows travel through registers after the last MFMA: thread (slot = tid / 48, ol = tid % 48) takes image rows
  // slot + 10 q, all loads of a batch in flight together.  (Requesting the batch BEFORE the last chunk's MFMAs hides
  // its latency -- the staging phase fell from 20 k to 4 k cycles per tile -- but the 24 registers it holds across
  // the loop spill the scans: 10.2 -> 11.7 ms.  Measured, not kept.)
  constexpr int NPQ = LA ? FU_NPQ_LA : FU_NPQ;
  constexpr uint32_t NG = LA ? 6 : 5;   // groups per frame of P
  double pr[NPQ];
  const uint32_t pslot = tid / 48, pol = tid % 48;
  const bool plive = tid < 480 && o0 + pol < n_out;
  auto p_load = [&](uint32_t r0) {
#pragma unroll
    for (int q = 0; q < NPQ; q++) {
      const uint32_t r = r0 + pslot + 10 * q;
      const uint32_t rm = rowmap[r < nprows ? r : 0];
      // (frame f0 - 1 exists whenever a row refers to it: t0 >= D then)
      pr[q] = (FU_ABL < 1 && plive && r < nprows && !(LA && rm == 0xffffu)) ? P[((ft.fr0 - 1) * NG + rm) * (uint64_t)n_out + o0 + pol] : 0.0;
    }
  };
  auto p_store = [&](uint32_t r0) {
#pragma unroll
    for (int q = 0; q < NPQ; q++) {
      const uint32_t r = r0 + pslot + 10 * q;
      if (tid < 480 && r < nprows) Pl[r * FU_DS + pol] = pr[q];
    }
  };

  float* dump = Xg + FU_ROWS * FU_XS + FU_GC;   // pad column of the spare row: never an operand
  // steady state: every frame of the tile has all DMAX durations (no clamped loads, no dumped stores)
  const bool full = FU_FULLSCAN && (D == (uint32_t)DMAX) && ft.t0 + 1 >= D;
  const uint32_t ncw = min(W, (uint32_t)FU_GC), mncw = fu_magic(ncw);   // width of a full chunk
  const uint32_t nchunks = 3 * cpg;
  for (uint32_t ci = LA ? cpg : 0; ci < nchunks; ci++) {
    const uint32_t ty = ci / cpg, c0 = (ci % cpg) * FU_GC;   // 0 avg, 1 max, 2 min
    const uint32_t nc = min((uint32_t)FU_GC, W - c0);
    // a narrower last chunk of a group leaves stale columns behind: clear them
    if (nc < FU_GC && cpg > 1)
      for (uint32_t i = tid; i < ft.nrows * (FU_GC - nc); i += FU_NT) {
        const uint32_t row = i / (FU_GC - nc), c = nc + i % (FU_GC - nc);
        Xg[row * FU_XS + c] = 0.0f;
      }
    const uint32_t mnc = nc == ncw ? mncw : fu_magic(nc);
    for (uint32_t i = tid; i < ft.nfr * nc; i += FU_NT) {
      const uint32_t tl = fu_div(i, mnc), c = i - tl * nc;
      const uint32_t t = ft.t0 + tl;
      float v[DMAX];
      const uint32_t at = (t - ft.f0) * W + c0 + c;
      const float* last = fr + at;
      if (full) {
        float* o = Xg + tl * (DMAX * FU_XS) + c;
        fu_load_vals_full<DMAX>(fr, at, W, v);
        if (ty == 0) fu_scan_avg_full<DMAX, FU_XS>(v, o);
        else if (ty == 1) fu_scan_ext_full<DMAX, 1, FU_XS>(v, o);
        else fu_scan_ext_full<DMAX, 0, FU_XS>(v, o);
      } else {
        const uint32_t nd = scrf_node_max_dur(t, D);
        fu_load_vals<DMAX>(last, W, nd, v);
        float* o = Xg + rbase[tl] * FU_XS + c;
        if (ty == 0) fu_scan_avg<DMAX>(v, o, FU_XS, 1, nd, dump);
        else if (ty == 1) fu_scan_ext<DMAX, 1>(v, o, FU_XS, 1, nd, dump);
        else fu_scan_ext<DMAX, 0>(v, o, FU_XS, 1, nd, dump);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t e = tid + FU_NT * q, ol = e / FU_GC, c = e % FU_GC;
      if (e < 48 * FU_GC) {
        if (F32) Wgf[c * (2 * FU_WS) + ol] = (float)wp[q];
        else Wg[c * FU_WS + ol] = wp[q];
      }
    }
    __syncthreads();
    FU_STAMP(1);   // scans (+ barrier)
    if (ci + 1 < nchunks) load_w(ci + 1);   // lands under the MFMAs
    FU_SETPRIO(0);
#pragma unroll
    for (int ks = 0; ks < FU_GC / 4; ks++) {
      if (F32) {
        float wv[3];
#pragma unroll
        for (int n = 0; n < 3; n++) wv[n] = Wgf[(ks * 4 + lk) * (2 * FU_WS) + n * (FU_PHI ? 16 : 4) + phi0];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const float x = Xg[(wave * 32 + m * 16 + li) * FU_XS + ks * 4 + lk];
#pragma unroll
          for (int n = 0; n < 3; n++)
            acc32[F32 ? m : 0][F32 ? n : 0] =
                __builtin_amdgcn_mfma_f32_16x16x4f32(wv[n], x, acc32[F32 ? m : 0][F32 ? n : 0], 0, 0, 0);
        }
      } else {
        double wv[3];
#pragma unroll
        for (int n = 0; n < 3; n++) wv[n] = Wg[(ks * 4 + lk) * FU_WS + n * (FU_PHI ? 16 : 4) + phi0];
#pragma unroll
        for (int m = 0; m < 2; m++) {
          const double x = (double)Xg[(wave * 32 + m * 16 + li) * FU_XS + ks * 4 + lk];
#pragma unroll
          for (int n = 0; n < 3; n++)
            acc[F32 ? 0 : m][F32 ? 0 : n] = DEC ? __builtin_amdgcn_mfma_f64_16x16x4f64(x, wv[n], acc[F32 ? 0 : m][F32 ? 0 : n], 0, 0, 0)
                                                : __builtin_amdgcn_mfma_f64_16x16x4f64(wv[n], x, acc[F32 ? 0 : m][F32 ? 0 : n], 0, 0, 0);
        }
      }
    }
    FU_SETPRIO(1);
    __syncthreads();
    FU_STAMP(2);   // MFMA loops (+ barrier)
  }
  // the P image over the dead operand images
  for (uint32_t r0 = 0; r0 < nprows; r0 += 10 * NPQ) {   // one batch unless the tile is wider than the registers hold
    p_load(r0);
    p_store(r0);
  }
  __syncthreads();
  FU_STAMP(3);   // P staging
  // epilogue: + sampled-frame projections + (one-hot duration weight + bias) (all fp64), write S.
  // Branch-free per row: every lane gathers (rows past the tile read row 0's valid table entries and are masked at
  // the stores), so the 36 16-byte LDS reads of a row are in flight together.
  // value c of a lane is output o0 + OL(c); pair j = (c >> 1) is adjacent in memory at o0 + OB(j)
#if FU_PHI
#define FU_OB(j) (8 * (j) + 2 * lk)
#else
#define FU_OB(j) (lk * 12 + 2 * (j))
#endif
#define FU_OL(c) (FU_OB((c) >> 1) + ((c) & 1))
  const bool vec_ok = (n_out & 1) == 0;             // 16-byte stores of fp64 pairs need even row lengths
  const bool all_out = vec_ok && o0 + 48 <= n_out;  // every lane's 12 outputs exist
#if !FU_EXPTAB
  const FuExpC ek = fu_exp_consts();
#endif
  if (DEC) {
    // untransposed tile: lane (li, lk) holds rows lk + 4r of outputs n * 16 + li; per row five 8-byte gathers per output
    // tile, the float weight, the rounding screen, a 64-byte store per (row, output tile)
    const double xm = (double)dz.xm_f[ft.fr0 + back] * dz.bound_scale;   // a tile lies in one utterance
#pragma unroll
    for (int m = 0; m < 2; m++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t rl = wave * 32 + m * 16 + lk + 4 * r;
        const bool valid = rl < ft.nrows;
        const uint4 rec = recs[valid ? rl : 0];
        const uint32_t d = (rec.z >> 16) & 0xffu;
        const uint64_t grow = ft.row0 + rl;
#pragma unroll
        for (int n = 0; n < 3; n++) {
          const uint32_t ol = n * 16 + li, o = o0 + ol;
          const double lin = (((Pl[(rec.x & 0xffffu) + ol] + Pl[(rec.x >> 16) + ol]) + Pl[(rec.y & 0xffffu) + ol]) + Pl[(rec.y >> 16) + ol]) + Pl[(rec.z & 0xffffu) + ol];
          const double v = -1 * ((acc[F32 ? 0 : m][F32 ? 0 : n][r] + lin) + Dt[(d - 1) * FU_DS + ol]);
          const float w = (float)v;
          if (valid && o < n_out) {
            const double B = xm * dz.w1[o];
            if ((float)(v - B) != w || (float)(v + B) != w) {
              const uint32_t at = atomicAdd(dz.cnt, 1u);
              if (at < dz.cap) dz.list[at] = (grow << 16) | o;
            }
            dz.wneg[grow * n_out + o] = w;
          }
        }
      }
    return;
  }
#pragma unroll
  for (int m = 0; m < 2; m++) {
    const uint32_t rl = wave * 32 + m * 16 + li;
    const bool valid = rl < ft.nrows;
    const uint4 rec = recs[valid ? rl : 0];
    const uint32_t d = (rec.z >> 16) & 0xffu, tl = rec.z >> 24;
    const double* p0 = Pl + (rec.x & 0xffffu);
    const double* p1 = Pl + (rec.x >> 16);
    const double* p2 = Pl + (rec.y & 0xffffu);
    const double* p3 = Pl + (rec.y >> 16);
    const double* p4 = Pl + (rec.z & 0xffffu);
    const double* dp = Dt + (d - 1) * FU_DS;
    // LA: prefix sums at the window's last frame and at the frame before its first
    const double* pct = Pl + (crow0 + D + tl) * FU_DS;
    const double* pcb = pct - d * FU_DS;
    const double invd = LA ? dp[48] : 0.0;
    double sv[12];
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const uint32_t ob = FU_OB(j);
      const v2f64 a0 = *(const v2f64*)(p0 + ob), a1 = *(const v2f64*)(p1 + ob), a2 = *(const v2f64*)(p2 + ob),
                  a3 = *(const v2f64*)(p3 + ob), a4 = *(const v2f64*)(p4 + ob), dw = *(const v2f64*)(dp + ob);
      v2f64 ct = (v2f64){0.0, 0.0}, cb = (v2f64){0.0, 0.0};
      if (LA) { ct = *(const v2f64*)(pct + ob); cb = *(const v2f64*)(pcb + ob); }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int c = 2 * j + h, n = c >> 2, r = c & 3;
        const double lin = (((a0[h] + a1[h]) + a2[h]) + a3[h]) + a4[h];
        const double v = F32 ? (double)acc32[F32 ? m : 0][F32 ? n : 0][r] : acc[F32 ? 0 : m][F32 ? 0 : n][r];
        sv[c] = (v + lin) + dw[h];
        if (LA) sv[c] = fma(ct[h] - cb[h], invd, sv[c]);
      }
    }
    const uint64_t grow = ft.row0 + rl;
    if (smax) {
      // row maximum as a float: over the lane's 12 outputs, then over the 4 lanes (li, lk = 0..3) that share the row --
      // two half-swaps (v_permlane32_swap / v_permlane16_swap, CDNA4)
      double mxd = sv[0];
#pragma unroll
      for (int c = 1; c < 12; c++) mxd = fu_vmaxd(mxd, sv[c]);
      float mx = (float)mxd;
      {
        const auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fu_vmax(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
        const auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fu_vmax(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
      }
      const double ref = (double)mx;
      // the labelled window's score: value cm of this lane, if the output is one of the lane's
#if FU_PHI
      const uint32_t tm = rec.w - o0 - 2 * lk;
      const bool has = valid && tm < 48u && (tm & 6u) == 0;
      const uint32_t cm = 2 * (tm >> 3) + (tm & 1u);
#else
      const uint32_t cm = rec.w - o0 - lk * 12;
      const bool has = valid && cm < 12u;
#endif
      if (__any(has)) {
        if (has) {
          double v = sv[0];
#pragma unroll
          for (int c = 1; c < 12; c++) v = (cm == (uint32_t)c) ? sv[c] : v;
          s_true[ft.fr0 + back + tl] = v;
        }
      }
      double ev[12];
#pragma unroll
      for (int c = 0; c < 12; c++) {
#if FU_EXPTAB
        ev[c] = fu_exp_tab(sv[c] - ref, etab);
#else
        ev[c] = fu_exp(sv[c] - ref, ek);
#endif
      }
      double* Srow = S + grow * n_out + o0;
      if (all_out) {
        if (valid) {
#pragma unroll
          for (int j = 0; j < 6; j++) __builtin_nontemporal_store((v2f64){ev[2 * j], ev[2 * j + 1]}, (v2f64*)(Srow + FU_OB(j)));
        }
      } else if (vec_ok) {
#pragma unroll
        for (int j = 0; j < 6; j++)
          if (valid && o0 + FU_OB(j) < n_out) __builtin_nontemporal_store((v2f64){ev[2 * j], ev[2 * j + 1]}, (v2f64*)(Srow + FU_OB(j)));
      } else {
#pragma unroll
        for (int c = 0; c < 12; c++) if (valid && o0 + FU_OL(c) < n_out) __builtin_nontemporal_store(ev[c], &Srow[FU_OL(c)]);
      }
      if (valid && lk == 0) smax[grow] = ref;
    } else {
      double* Srow = S + grow * n_out + o0;
      if (vec_ok) {
#pragma unroll
        for (int j = 0; j < 6; j++)
          if (valid && o0 + FU_OB(j) < n_out) __builtin_nontemporal_store((v2f64){sv[2 * j], sv[2 * j + 1]}, (v2f64*)(Srow + FU_OB(j)));
      } else {
#pragma unroll
        for (int c = 0; c < 12; c++) if (valid && o0 + FU_OL(c) < n_out) __builtin_nontemporal_store(sv[c], &Srow[FU_OL(c)]);
      }
    }
  }
  FU_STAMP(4);   // epilogue of wave 0
#undef FU_OB
#undef FU_OL
}

static size_t fused_scores_smem_tb(uint32_t W, uint32_t D, uint32_t TB) {
  const uint32_t nfmax = TB + D - 1;
  size_t opn = sizeof(float) * (FU_ROWS + 1) * FU_XS + sizeof(double) * FU_GC * FU_WS + sizeof(float) * nfmax * W;
  const uint32_t npr = fu_p_rows_la(D, TB);   // one tile plan for every form of the kernel
  const size_t pb = sizeof(double) * npr * FU_DS;
  if (pb > opn) opn = pb;
  opn = (opn + 15) & ~(size_t)15;
  return opn + sizeof(double) * D * FU_DS + sizeof(uint4) * FU_ROWS + sizeof(uint16_t) * ((TB + 3) & ~3u) +
         sizeof(uint16_t) * ((npr + 3) & ~3u) + (FU_EXPTAB ? sizeof(double) * FU_EXPT_N : 0) + 16;
}
// frames per score tile: as many whole frames as give <= 256 rows and keep the workgroup's LDS
// (the staged P rows grow with TB + D - 1) within 80 KB, i.e. two workgroups per CU; 0 = no fit
uint32_t fused_scores_tb(uint32_t W, uint32_t D) {
  for (uint32_t TB = FU_ROWS / D; TB >= 1; TB--)
    if (fused_scores_smem_tb(W, D, TB) <= 80 * 1024) return TB;
  // long durations with wide frames (D = 32 at W = 39): one workgroup per CU, as many frames as fit
  for (uint32_t TB = FU_ROWS / D; TB >= 1; TB--)
    if (fused_scores_smem_tb(W, D, TB) <= 156 * 1024) return TB;
  return 0;
}
static size_t fused_scores_smem(uint32_t W, uint32_t D) { return fused_scores_smem_tb(W, D, fused_scores_tb(W, D)); }

// dtab[y][d][FU_DS]: one-hot duration weight + bias term of output 48 y + o (slots 0..47; -1e300 past n_out), 1/(d+1) in
// slot 48 -- what k_scores_fused stages per tile, built once per launch
__global__ void k_dur_table(ScrfLayout lay, uint32_t W, const double* __restrict__ lambda, double* __restrict__ dtab) {
  const uint32_t D = lay.D, n_out = lay.L;
  const uint32_t o0 = blockIdx.x * 48;
  for (uint32_t i = threadIdx.x; i < D * FU_DS; i += blockDim.x) {
    const uint32_t dd = i / FU_DS, oo = i % FU_DS;
    double v = 0.0;
    if (oo == 48) v = 1.0 / (double)(dd + 1);
    else if (oo < 48) {
      v = -1e300;
      if (o0 + oo < n_out) {
        const uint32_t base = lay.state_idx(o0 + oo) + 8 * W;
        v = lambda[base + dd];
        if (lay.use_sb) v += lambda[base + D] * lay.sbv;
      }
    }
    dtab[(size_t)blockIdx.x * D * FU_DS + i] = v;
  }
}
// rtab: the row records (label slot 0xffff), row bases and rowmap of a steady-state score tile (t0 >= D, TB whole frames):
// [FU_ROWS] uint4 | [(TB + 7) & ~7] uint16 | [fu_p_rows_la] uint16 -- the same arithmetic as the kernel's own (edge) path
__global__ void k_tile_tables(uint32_t D, uint32_t TB, int la, unsigned char* __restrict__ rtab) {
  uint4* rg = (uint4*)rtab;
  uint16_t* bg = (uint16_t*)(rg + FU_ROWS);
  uint16_t* mg = bg + ((TB + 7) & ~7u);
  const uint32_t back = D - 1, nf = back + TB;
  uint32_t prow0[5], pf0[5], nprows = 0;
  for (int k = 0; k < 5; k++) {
    const uint32_t reach = D - 1 - fu_sample_step(D, k);
    pf0[k] = back > reach ? back - reach : 0;
    prow0[k] = nprows;
    nprows += nf - pf0[k];
  }
  const uint32_t crow0 = nprows;
  if (la) nprows += TB + D;
  for (uint32_t i = threadIdx.x; i < TB * D; i += blockDim.x) {
    const uint32_t tl = i / D, d = i - tl * D + 1;
    const uint32_t b0 = tl + D - d;   // t - d + 1 - f0 with f0 = t0 - (D - 1)
    uint32_t q[5];
    for (int k = 0; k < 5; k++) q[k] = (prow0[k] - pf0[k] + b0 + fu_sample_step(d, k)) * FU_DS;
    rg[i] = make_uint4(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (d << 16) | (tl << 24), 0xffffu);
  }
  for (uint32_t tl = threadIdx.x; tl < TB; tl += blockDim.x) bg[tl] = (uint16_t)(tl * D);
  for (uint32_t r = threadIdx.x; r < nprows; r += blockDim.x) {
    if (la && r >= crow0) {   // frame t0 - D + (r - crow0), one before f0 at most: 6 * (frame - f0 + 1) + 5
      mg[r] = (uint16_t)(6 * (r - crow0) + 5);
      continue;
    }
    uint32_t k = 0;
    for (int kk = 1; kk < 5; kk++) k += r >= prow0[kk] ? 1u : 0u;
    mg[r] = (uint16_t)((la ? 6 : 5) * (pf0[k] + r - prow0[k] + 1) + k);
  }
}
size_t fused_tile_table_bytes(uint32_t D, uint32_t TB) {
  return sizeof(uint4) * FU_ROWS + sizeof(uint16_t) * (((TB + 7) & ~7u) + ((fu_p_rows_la(D, TB) + 7) & ~7u)) + 64;
}
void launch_tile_tables(hipStream_t st, uint32_t D, uint32_t TB, int la, void* rtab) {
  hipLaunchKernelGGL(k_tile_tables, dim3(1), dim3(256), 0, st, D, TB, la, (unsigned char*)rtab);
}
size_t fused_dur_table_doubles(const ScrfLayout& lay) { return (size_t)((lay.L + 47) / 48) * lay.D * FU_DS; }
void launch_dur_table(hipStream_t st, const ScrfLayout& lay, uint32_t W, const double* lambda, double* dtab) {
  hipLaunchKernelGGL(k_dur_table, dim3((lay.L + 47) / 48), dim3(256), 0, st, lay, W, lambda, dtab);
}

template <int DMAX, int F32, int DEC, int LA>
static void launch_scores_fused_t(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* lambda,
                                  const double* P, uint64_t n_tiles, double* S, double* smax, double* s_true,
                                  const uint32_t* labels, const ScrfDecodeOut& dz) {
  const size_t sm = fused_scores_smem(fa.W, lay.D);
  dim3 grid((uint32_t)n_tiles, (lay.L + 47) / 48);
  hipFuncSetAttribute((const void*)k_scores_fused<DMAX, F32, DEC, LA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL((k_scores_fused<DMAX, F32, DEC, LA>), grid, dim3(FU_NT), sm, st, fa, lay, lambda, P, lay.L, S, smax,
                     s_true, labels, dz);
}

// la (SCRF_PREC_FASTLIN, fp64 only): P carries 6 groups per frame, the sixth summed along the utterance (k_avg_prefix)
void launch_scores_fused(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* lambda,
                         const double* P, uint64_t n_tiles, double* S, int f32, double* smax, double* s_true,
                         const uint32_t* labels, int la) {
  if (n_tiles == 0) return;
  if (lay.L > 48) smax = nullptr;   // a row spans several workgroups: the caller runs k_exp_rows instead
  ScrfDecodeOut off;
  memset(&off, 0, sizeof(off));
#define FS_GO(N)                                                                        \
  do {                                                                                  \
    if (la) launch_scores_fused_t<N, 0, 0, 1>(st, fa, lay, lambda, P, n_tiles, S, smax, s_true, labels, off); \
    else if (f32) launch_scores_fused_t<N, 1, 0, 0>(st, fa, lay, lambda, P, n_tiles, S, smax, s_true, labels, off);   \
    else launch_scores_fused_t<N, 0, 0, 0>(st, fa, lay, lambda, P, n_tiles, S, smax, s_true, labels, off);       \
  } while (0)
  if (lay.D <= 12) FS_GO(12);
  else if (lay.D <= 25) FS_GO(25);
  else FS_GO(40);
#undef FS_GO
}

void launch_scores_fused_decode(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* lambda,
                                const double* P, uint64_t n_tiles, const ScrfDecodeOut& dz) {
  if (n_tiles == 0) return;
  if (lay.D <= 12) launch_scores_fused_t<12, 0, 1, 0>(st, fa, lay, lambda, P, n_tiles, nullptr, nullptr, nullptr, nullptr, dz);
  else if (lay.D <= 25) launch_scores_fused_t<25, 0, 1, 0>(st, fa, lay, lambda, P, n_tiles, nullptr, nullptr, nullptr, nullptr, dz);
  else launch_scores_fused_t<40, 0, 1, 0>(st, fa, lay, lambda, P, n_tiles, nullptr, nullptr, nullptr, nullptr, dz);
}

// ------------------------------------------------------------------------------------------
// k_lin_z: Z[f][k][o] = sum_d R[(t = f + off_k(d), d)][o], off_k(d) = d - 1 - s_k(d): the windows
// whose k-th sampled frame is f.  Thread = (utterance, output o): walks the utterance once.  off_k(d) is a compile-time table, so the partial sums of
// the frames still open live in REGISTERS: a window of off_k(D)+1 doubles per k that slides by one
// frame per step (no LDS, no atomics).  Sum order: t ascending, then d ascending.
// ------------------------------------------------------------------------------------------
constexpr int fu_step_c(int d, int k) {
  const float ot = (float)((double)d * 0.1);
  const float x = ot * (float)(2 * k + 1);
  int c = (int)x;
  if ((float)c < x) c++;
  return c - 1;
}
constexpr int fu_off_c(int d, int k) { return d - 1 - fu_step_c(d, k); }
template <int K, int DMAX>
struct LzWin {
  static constexpr int MO = fu_off_c(DMAX, K);   // frames a window of length <= DMAX can reach back
  double v[MO + 1];                              // v[j]: partial sum of frame t - j
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int j = 0; j <= MO; j++) v[j] = 0.0;
  }
  // the indices are template constants (a pack expansion): the window is a set of named registers
  template <int... D0>
  __device__ __forceinline__ void add_seq(const double (&r)[DMAX], std::integer_sequence<int, D0...>) {
    ((v[std::integral_constant<int, fu_off_c(D0 + 1, K)>::value] += r[D0]), ...);
  }
  __device__ __forceinline__ void add(const double (&r)[DMAX]) { add_seq(r, std::make_integer_sequence<int, DMAX>{}); }
  // the same from an LDS image of the frame's rows, rows[d0 * 64 + lane] (k_lin_z5)
  template <int... D0>
  __device__ __forceinline__ void add_lds_seq(const double* rows, std::integer_sequence<int, D0...>) {
    ((v[std::integral_constant<int, fu_off_c(D0 + 1, K)>::value] += rows[D0 * 64]), ...);
  }
  __device__ __forceinline__ void add_lds(const double* rows) { add_lds_seq(rows, std::make_integer_sequence<int, DMAX>{}); }
  // after frame t: frame t - MO is final; slide by one frame
  __device__ __forceinline__ void retire(double* Zk, int t, size_t zstride) {
    if (t >= MO) __builtin_nontemporal_store(v[MO], &Zk[(size_t)(t - MO) * zstride]);
#pragma unroll
    for (int j = MO; j >= 1; j--) v[j] = v[j - 1];
    v[0] = 0.0;
  }
  // after the last frame T-1 (already slid): v[j] holds frame T - j, j = 1 .. MO
  __device__ __forceinline__ void flush(double* Zk, int T, size_t zstride) {
#pragma unroll
    for (int j = 1; j <= MO; j++)
      if (T - j >= 0) __builtin_nontemporal_store(v[j], &Zk[(size_t)(T - j) * zstride]);
  }
  // The same for a walk over the frame range of ONE SEGMENT of the utterance (k_post_z's split form): frames lo <= f < hi
  // get windows from this segment only and are stored; the others are shared with the neighbouring segment and added
  // into the zeroed array (two contributions at most: the sum does not depend on their order).
  __device__ __forceinline__ void retire_seg(double* Zk, int t, size_t zstride, int lo, int hi) {
    if (t >= MO) {
      const int f = t - MO;
      if (f >= lo && f < hi) __builtin_nontemporal_store(v[MO], &Zk[(size_t)f * zstride]);
      else unsafeAtomicAdd(&Zk[(size_t)f * zstride], v[MO]);
    }
#pragma unroll
    for (int j = MO; j >= 1; j--) v[j] = v[j - 1];
    v[0] = 0.0;
  }
  __device__ __forceinline__ void flush_seg(double* Zk, int fb, size_t zstride, int lo, int hi) {
#pragma unroll
    for (int j = 1; j <= MO; j++) {
      const int f = fb - j;
      if (f < 0) continue;
      if (f >= lo && f < hi) __builtin_nontemporal_store(v[j], &Zk[(size_t)f * zstride]);
      else unsafeAtomicAdd(&Zk[(size_t)f * zstride], v[j]);
    }
  }
};

template <int DMAX>
__device__ __forceinline__ void lz_load(const double* Ru, uint32_t L, uint32_t D, int t, int T, double (&r)[DMAX]) {
  const uint32_t nd = (t < T) ? scrf_node_max_dur((uint32_t)t, D) : 0;
  const double* Rt = Ru + scrf_seg_base((uint32_t)(t < T ? t : 0), D) * (uint64_t)L;
#pragma unroll
  for (int d0 = 0; d0 < DMAX; d0++) r[d0] = ((uint32_t)d0 < nd) ? Rt[(uint64_t)d0 * L] : 0.0;
}

// one wavefront per (utterance, 64 outputs): every R row is loaded exactly once and feeds the five
// sample positions; the next frame's rows are in flight under the adds of the current one
template <int DMAX>
__global__ __launch_bounds__(64, 2) void k_lin_z(ScrfLayout lay, ScrfBatchView bv, uint32_t u0,
                                                 const double* __restrict__ R, double* __restrict__ Z) {
  const uint32_t D = lay.D, L = lay.L;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint32_t o = blockIdx.y * 64 + threadIdx.x;
  if (o >= L) return;
  const double* Ru = R + (bv.seg_off[u] - bv.seg_off[u0]) * (uint64_t)L + o;
  double* Zu = Z + (bv.frame_off[u] - bv.frame_off[u0]) * (uint64_t)(5 * L) + o;
  LzWin<0, DMAX> w0; LzWin<1, DMAX> w1; LzWin<2, DMAX> w2; LzWin<3, DMAX> w3; LzWin<4, DMAX> w4;
  w0.clear(); w1.clear(); w2.clear(); w3.clear(); w4.clear();
  const size_t zs = (size_t)5 * L;
  double r[DMAX], rn[DMAX];
  lz_load<DMAX>(Ru, L, D, 0, T, r);
#pragma unroll 1
  for (int t = 0; t < T; t++) {
    lz_load<DMAX>(Ru, L, D, t + 1, T, rn);
    w0.add(r); w1.add(r); w2.add(r); w3.add(r); w4.add(r);
    w0.retire(Zu, t, zs);
    w1.retire(Zu + L, t, zs);
    w2.retire(Zu + 2 * (size_t)L, t, zs);
    w3.retire(Zu + 3 * (size_t)L, t, zs);
    w4.retire(Zu + 4 * (size_t)L, t, zs);
#pragma unroll
    for (int d0 = 0; d0 < DMAX; d0++) r[d0] = rn[d0];
  }
  w0.flush(Zu, T, zs);
  w1.flush(Zu + L, T, zs);
  w2.flush(Zu + 2 * (size_t)L, T, zs);
  w3.flush(Zu + 3 * (size_t)L, T, zs);
  w4.flush(Zu + 4 * (size_t)L, T, zs);
}

// ------------------------------------------------------------------------------------------
// k_lin_z5 (hybrid path, any D up to 40 and any L): the per-frame sums Z_k[f] of R for the five sample positions AND the
// per-duration sums of R (the counts of the one-hot duration columns and of the bias) in one walk over R.  k_lin_z keeps
// the five sliding windows of a wavefront in registers -- 2 x 5 x ~D/2 of them, which spills at D = 40 (46.8 ms at
// BASELINE config 5).  Here a workgroup of six wavefronts owns (utterance, 64 outputs, frame segment): the frame's rows are
// staged once in LDS (double buffered, loads a frame ahead), wavefront k < 5 slides sample position k's window, wavefront
// 5 adds the rows into per-duration registers.  Frame segments as in k_post_z's split form (boundary frames added from
// both sides into the zeroed Z).  dslab[(utterance, segment)][o][D + bias] is reduced like the fused count kernel's.
// ------------------------------------------------------------------------------------------
// One role's frame loop (K < 5: sample position K's window; K = 5: the per-duration sums).  Every role runs the same loop
// with the same two barriers per frame and its share of the staging; the roles are separate functions so that a
// wavefront's registers hold ITS window only (one loop with six branches keeps all of them live: 300 registers, spills).
template <int K, int DMAX>
__device__ __forceinline__ void lz5_role(double* rows, const double* __restrict__ Ru, double* __restrict__ Zk, double* __restrict__ dout,
                                         const ScrfLayout& lay, int T, int fa, int fb, uint32_t ocols, uint32_t o, bool act, int lane) {
  const uint32_t D = lay.D, L = lay.L;
  const size_t zs = (size_t)5 * L;
  const int own_lo = fa, own_hi = (fb == T) ? 0x7fffffff : fb - (int)D + 1;
  // staging: thread -> elements e = tid + 384 j of the frame's [DMAX][64] image (row e / 64, output e % 64)
  constexpr int NE = (DMAX * 64 + 383) / 384;
  auto fetch = [&](int t, double (&x)[NE]) {
    const uint32_t nd = t < T ? scrf_node_max_dur((uint32_t)t, D) : 0;
    const double* Rt = Ru + scrf_seg_base((uint32_t)(t < T ? t : 0), D) * (uint64_t)L;
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const uint32_t e = threadIdx.x + 384 * j, d0 = e >> 6, c = e & 63;
      x[j] = (d0 < nd && c < ocols) ? __builtin_nontemporal_load(&Rt[(uint64_t)d0 * L + c]) : 0.0;
    }
  };
  auto stash = [&](int buf, const double (&x)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; j++) {
      const uint32_t e = threadIdx.x + 384 * j;
      if (e < (uint32_t)DMAX * 64) rows[buf * DMAX * 64 + e] = x[j];
    }
  };
  double x[NE];
  fetch(fa, x);
  stash(0, x);
  LzWin<(K < 5 ? K : 0), DMAX> w;
  double cd[K == 5 ? DMAX : 1];
  if (K < 5) w.clear();
  else {
#pragma unroll
    for (int d0 = 0; d0 < (K == 5 ? DMAX : 1); d0++) cd[d0] = 0.0;
  }
  for (int t = fa; t < fb; t++) {
    const int cur = (t - fa) & 1;
    if (t + 1 < fb) fetch(t + 1, x);                       // in flight under this frame's adds
    __syncthreads();                                       // frame t's image is complete
    const double* img = rows + cur * DMAX * 64 + lane;
    if (K < 5) {
      w.add_lds(img);
      w.retire_seg(Zk, act ? t : -1000000, zs, own_lo, own_hi);
    } else {
#pragma unroll
      for (int d0 = 0; d0 < (K == 5 ? DMAX : 1); d0++) cd[d0] += img[d0 * 64];
    }
    if (t + 1 < fb) stash(cur ^ 1, x);                     // the other image: last read before the previous frame's second barrier
    __syncthreads();
  }
  if (!act) return;
  if (K < 5) w.flush_seg(Zk, fb, zs, own_lo, own_hi);
  else {
    const uint32_t nd1 = D + (lay.use_sb ? 1 : 0);
    double bsum = 0.0;
#pragma unroll
    for (int d0 = 0; d0 < (K == 5 ? DMAX : 1); d0++) {
      if ((uint32_t)d0 < D) { dout[(uint64_t)o * nd1 + d0] = cd[d0]; bsum += cd[d0]; }
    }
    if (lay.use_sb) dout[(uint64_t)o * nd1 + D] = bsum * lay.sbv;
  }
}
template <int DMAX>
__global__ __launch_bounds__(384) void k_lin_z5(ScrfLayout lay, ScrfBatchView bv, uint32_t u0, const double* __restrict__ R,
                                                double* __restrict__ Z, double* __restrict__ dslab, int seg_len) {
  __shared__ double rows[2 * DMAX * 64];
  const uint32_t D = lay.D, L = lay.L;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const uint32_t o = blockIdx.y * 64 + lane;
  const bool act = o < L;
  const int fa = (int)blockIdx.z * seg_len;
  const int fb = fa + seg_len < T ? fa + seg_len : T;
  const uint32_t nd1 = D + (lay.use_sb ? 1 : 0);
  double* dout = dslab + ((uint64_t)blockIdx.x * gridDim.z + blockIdx.z) * L * nd1;
  if (fa >= T) {   // an empty segment still owns its slab block
    if (wave == 5 && act) for (uint32_t j = 0; j < nd1; j++) dout[(uint64_t)o * nd1 + j] = 0.0;
    return;
  }
  const double* Ru = R + (bv.seg_off[u] - bv.seg_off[u0]) * (uint64_t)L + blockIdx.y * 64;
  double* Zu = Z + (bv.frame_off[u] - bv.frame_off[u0]) * (uint64_t)(5 * L) + o;
  const uint32_t ocols = L - blockIdx.y * 64 < 64 ? L - blockIdx.y * 64 : 64;
  if (wave == 0) lz5_role<0, DMAX>(rows, Ru, Zu, dout, lay, T, fa, fb, ocols, o, act, lane);
  else if (wave == 1) lz5_role<1, DMAX>(rows, Ru, Zu + L, dout, lay, T, fa, fb, ocols, o, act, lane);
  else if (wave == 2) lz5_role<2, DMAX>(rows, Ru, Zu + 2 * (size_t)L, dout, lay, T, fa, fb, ocols, o, act, lane);
  else if (wave == 3) lz5_role<3, DMAX>(rows, Ru, Zu + 3 * (size_t)L, dout, lay, T, fa, fb, ocols, o, act, lane);
  else if (wave == 4) lz5_role<4, DMAX>(rows, Ru, Zu + 4 * (size_t)L, dout, lay, T, fa, fb, ocols, o, act, lane);
  else lz5_role<5, DMAX>(rows, Ru, Zu, dout, lay, T, fa, fb, ocols, o, act, lane);
}
// frame segments per utterance so that the launch has about 1024 workgroups, none shorter than 2 D frames
uint32_t lin_z5_segments(uint32_t n_utts, uint32_t L, uint32_t D, uint32_t t_max, int* seg_len) {
  const uint64_t wgs = (uint64_t)n_utts * ((L + 63) / 64);
  uint32_t want = wgs >= 1024 ? 1 : (uint32_t)((1024 + wgs - 1) / wgs);
  int sl = (int)((t_max + want - 1) / (want ? want : 1));
  if (sl < (int)(2 * D)) sl = (int)(2 * D);
  if (sl < 1) sl = 1;
  *seg_len = sl;
  const uint32_t nz = (t_max + sl - 1) / sl;
  return nz ? nz : 1;
}
void launch_lin_z5(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint32_t t_max, uint64_t n_frames,
                   const double* R, double* Z, double* dslab) {
  if (n_utts == 0) return;
  int seg_len = 0;
  const uint32_t nz = lin_z5_segments(n_utts, lay.L, lay.D, t_max, &seg_len);
  if (nz > 1) hipMemsetAsync(Z, 0, sizeof(double) * n_frames * 5 * lay.L, st);
  dim3 grid(n_utts, (lay.L + 63) / 64, nz);
#define LZ5_GO(N) hipLaunchKernelGGL(k_lin_z5<N>, grid, dim3(384), 0, st, lay, bv, u0, R, Z, dslab, seg_len)
  if (lay.D <= 8) LZ5_GO(8);
  else if (lay.D <= 16) LZ5_GO(16);
  else if (lay.D <= 25) LZ5_GO(25);
  else if (lay.D <= 32) LZ5_GO(32);
  else LZ5_GO(40);
#undef LZ5_GO
}

// k_add_p_exp (hybrid path: materialised dense statistics, L <= 256): the sampled blocks' share of the scores,
//   S[(t, d)][l] += sum_k P[t - d + 1 + s_k(d)][k][l],   s_k(d) = ceil(float(0.1 d) (2k + 1)) - 1
// (io/CRF_InFtrStream_SeqMultiWindow.cpp:556-884), and in the same pass what k_true_scores and k_exp_rows do for the
// linear-domain recursion: the labelled window's score, the row maximum, S -> exp(S - smax[row]).  One workgroup per
// frame, one wavefront per window row (lanes over the labels, L / 64 values per lane); the P rows of neighbouring
// frames overlap almost completely, so the gathers are L2 hits.
__global__ __launch_bounds__(256) void k_add_p_exp(ScrfLayout lay, ScrfBatchView bv, const uint32_t* __restrict__ frame_u, uint32_t u0,
                                                   uint64_t n_frames, const double* __restrict__ P, double* __restrict__ S, double* __restrict__ smax,
                                                   double* __restrict__ s_true) {
  const uint32_t L = lay.L, D = lay.D;
  // Workgroups go round-robin to the 8 XCDs, each with its own L2: XCD x walks the x-th eighth of the frames, so that the
  // frames in flight on one XCD are neighbours and their P rows (D frames back, 5 L doubles each) stay in ITS L2.  With
  // frame = blockIdx every XCD's 256 frames in flight span a whole utterance: 83 GB of HBM traffic per step at config 5
  // instead of 32 + P.
  const uint64_t per = (n_frames + 7) / 8;
  const uint64_t fi = (uint64_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if ((blockIdx.x >> 3) >= per || fi >= n_frames) return;
  const uint64_t gf = bv.frame_off[u0] + fi;
  const uint32_t u = frame_u[gf];
  const uint32_t t = (uint32_t)(gf - bv.frame_off[u]);
  const uint64_t fb = bv.frame_off[u] - bv.frame_off[u0];
  const uint32_t nd = scrf_node_max_dur(t, D);
  const uint64_t row0 = (bv.seg_off[u] - bv.seg_off[u0]) + scrf_seg_base(t, D);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t lab = bv.labels ? bv.labels[gf] : SCRF_LAB_BAD;
  uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD;
  if (lab != SCRF_LAB_BAD && lab < L * D) { al = lab % L; ld = lab / L + 1; }
  if (threadIdx.x == 0 && (ld == SCRF_LAB_BAD || ld > nd)) s_true[fi] = 0.0;
  for (uint32_t d0 = wave; d0 < nd; d0 += 4) {
    const float ot = (float)((double)(d0 + 1) * 0.1);
    double* Sr = S + (row0 + d0) * L;
    const double* Pb = P + (fb + t - d0) * (uint64_t)(5 * L);
    uint32_t step[5];
#pragma unroll
    for (int k = 0; k < 5; k++) step[k] = (uint32_t)ceilf(ot * (float)(2 * k + 1)) - 1u;
    double v[4];
    double m = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t l = lane + 64 * j;
      v[j] = -INFINITY;
      if (l < L) {
        double acc = Sr[l];
#pragma unroll
        for (int k = 0; k < 5; k++) acc += Pb[(uint64_t)step[k] * (5 * L) + (uint64_t)k * L + l];
        v[j] = acc;
        m = fmax(m, acc);
        if (d0 + 1 == ld && l == al) s_true[fi] = acc;
      }
    }
    m = wave_max_f64_dpp(m);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t l = lane + 64 * j;
      if (l < L) Sr[l] = exp_nonpos(v[j] - m);
    }
    if (lane == 0) smax[row0 + d0] = m;
  }
}
void launch_add_p_exp(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, const uint32_t* frame_u, uint32_t u0, uint64_t n_frames,
                      const double* P, double* S, double* smax, double* s_true) {
  if (n_frames == 0) return;
  hipLaunchKernelGGL(k_add_p_exp, dim3((uint32_t)(8 * ((n_frames + 7) / 8))), dim3(256), 0, st, lay, bv, frame_u, u0, n_frames, P, S, smax, s_true);
}

void launch_lin_z(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                  const double* R, double* Z) {
  if (n_utts == 0) return;
  dim3 grid(n_utts, (lay.L + 63) / 64);
#define LZ_GO(N) hipLaunchKernelGGL(k_lin_z<N>, grid, dim3(64), 0, st, lay, bv, u0, R, Z)
  if (lay.D <= 8) LZ_GO(8);
  else if (lay.D <= 16) LZ_GO(16);
  else if (lay.D <= 25) LZ_GO(25);
  else if (lay.D <= 32) LZ_GO(32);
  else LZ_GO(40);
#undef LZ_GO
}

// ------------------------------------------------------------------------------------------
// The two per-frame contractions of the sampled blocks (K = raw frame width), as wavefront-
// autonomous MFMA kernels without LDS or barriers (operands straight from memory, like k_atb):
//   k_pframe: P[f][k*L + l] = sum_c F[f][c] * lambda[state(l) + k*W + c]      (M = frames, N = 5L)
//   k_ztf   : slab[z][k*L + l][c] = sum_{f in chunk z} Z[f][k*L + l] * F[f][c]  (M = 5L, N = W, K = frames)
// A wavefront owns 80 outputs (5 MFMA tiles).  k_pframe keeps its slice of lambda in registers and
// streams 16-frame tiles; k_ztf streams its frame range four frames at a time.
// ------------------------------------------------------------------------------------------
#define PF_MT 4      // output tiles per wavefront (64 outputs)
template <int KS>   // k-steps: ceil(W / 4)
__global__ __launch_bounds__(64) void k_pframe(const float* __restrict__ F, uint32_t W, uint64_t n_frames,
                                               uint64_t frames_per_wave, const double* __restrict__ lambda,
                                               ScrfLayout lay, uint32_t n_out, double* __restrict__ P) {
  const uint32_t lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const uint32_t og = blockIdx.y * (16 * PF_MT);
  const uint64_t f_begin = (uint64_t)blockIdx.x * frames_per_wave;
  const uint64_t f_end = min(n_frames, f_begin + frames_per_wave);
  // B[k = c][j = o]: lane (li = o, lk = c within the k-step)
  double bw[PF_MT][KS];
#pragma unroll
  for (int nt = 0; nt < PF_MT; nt++) {
    const uint32_t o = og + nt * 16 + li;
    const uint32_t wo = (o < n_out) ? lay.state_idx(o % lay.L) + (o / lay.L) * W : 0;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const uint32_t c = ks * 4 + lk;
      bw[nt][ks] = (o < n_out && c < W) ? lambda[wo + c] : 0.0;
    }
  }
  // A[i = frame][k = c]: lane (li = frame, lk = c), kept as floats until used; two 16-frame tiles are in
  // flight while a third is multiplied and stored (loads and stores share one in-order counter on gfx9)
  float a1[KS], a2[KS];
  auto load_a = [&](uint64_t f0, float (&a_n)[KS]) {
    const uint64_t f = f0 + li;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const uint32_t c = ks * 4 + lk;
      a_n[ks] = (f < f_end && c < W) ? F[f * W + c] : 0.0f;
    }
  };
  auto tile = [&](uint64_t f0, const float (&av)[KS]) {
    v4f64 acc[PF_MT];
#pragma unroll
    for (int nt = 0; nt < PF_MT; nt++) acc[nt] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
      const double a = (double)av[ks];
#pragma unroll
      for (int nt = 0; nt < PF_MT; nt++) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bw[nt][ks], acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < PF_MT; nt++) {
      const uint32_t o = og + nt * 16 + li;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint64_t fr_ = f0 + lk + 4 * r;
        if (fr_ < f_end && o < n_out) __builtin_nontemporal_store((double)acc[nt][r], &P[fr_ * n_out + o]);
      }
    }
  };
  load_a(f_begin, a1);
  load_a(f_begin + 16, a2);
  for (uint64_t f0 = f_begin; f0 < f_end; f0 += 32) {
    float a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[ks] = a1[ks];
    load_a(f0 + 32, a1);
    tile(f0, a);
    if (f0 + 16 < f_end) {
#pragma unroll
      for (int ks = 0; ks < KS; ks++) a[ks] = a2[ks];
      load_a(f0 + 48, a2);
      tile(f0 + 16, a);
    }
  }
}

void launch_pframe(hipStream_t st, const float* F, uint32_t W, uint64_t n_frames, const double* lambda,
                   const ScrfLayout& lay, uint32_t n_out, double* P) {
  if (n_frames == 0) return;
  // about 2048 wavefronts per output group, whole pairs of 16-frame tiles each
  uint64_t fpw = ((n_frames + 2047) / 2048 + 31) & ~31ull;
  if (fpw < 64) fpw = 64;
  dim3 grid((uint32_t)((n_frames + fpw - 1) / fpw), (n_out + 16 * PF_MT - 1) / (16 * PF_MT));
  const uint32_t ks = (W + 3) / 4;
#define PF_GO(N) hipLaunchKernelGGL(k_pframe<N>, grid, dim3(64), 0, st, F, W, n_frames, fpw, lambda, lay, n_out, P)
  if (ks <= 4) PF_GO(4);
  else if (ks <= 10) PF_GO(10);
  else if (ks <= 16) PF_GO(16);
  else PF_GO(20);
#undef PF_GO
}
int pframe_supported(uint32_t W) { return W <= 80; }

// k_avg_prefix (SCRF_PREC_FASTLIN): group 5 of P (the projection of every frame on the avg weights) summed in place along
// each utterance, C[f][o] = sum_{f' <= f} Q[f'][o].  One wavefront per (utterance, 64 outputs), eight frames' loads in
// flight per step; the additions run in frame order (fixed association).
__global__ __launch_bounds__(64) void k_avg_prefix(ScrfBatchView bv, uint32_t u0, uint32_t L, double* __restrict__ P) {
  const uint32_t u = u0 + blockIdx.x, o = blockIdx.y * 64 + threadIdx.x;
  if (o >= L) return;
  const uint32_t T = bv.T[u];
  const size_t zs = (size_t)6 * L;
  double* q = P + (bv.frame_off[u] - bv.frame_off[u0]) * zs + (size_t)5 * L + o;
  double c = 0.0;
  for (uint32_t t0 = 0; t0 < T; t0 += 8) {
    double v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (t0 + j < T) ? q[(size_t)(t0 + j) * zs] : 0.0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      c += v[j];
      if (t0 + j < T) q[(size_t)(t0 + j) * zs] = c;
    }
  }
}
void launch_avg_prefix(hipStream_t st, ScrfBatchView bv, uint32_t u0, uint32_t n_utts, uint32_t L, double* P) {
  if (n_utts == 0) return;
  hipLaunchKernelGGL(k_avg_prefix, dim3(n_utts, (L + 63) / 64), dim3(64), 0, st, bv, u0, L, P);
}

// ZT_MT = output tiles per wavefront: 4 (64 outputs; leaves registers for a second prefetch stage) when the kernel has
// the chip to itself, 1 (74 registers: fits beside the two 216-register wavefronts per SIMD of k_expf_fused_ws) when it
// runs on the side stream under the expected-count kernel
template <int NT, int ZT_MT>   // NT = column tiles: ceil(W / 16)
__global__ __launch_bounds__(64) void k_ztf(const double* __restrict__ Zm, uint32_t n_out, const float* __restrict__ F,
                                            uint32_t W, uint64_t n_frames, uint64_t rows_per_chunk,
                                            double* __restrict__ slab) {
  const uint32_t lane = threadIdx.x, li = lane & 15, lk = lane >> 4;
  const uint32_t og = blockIdx.y * (16 * ZT_MT);
  const uint64_t r_begin = (uint64_t)blockIdx.x * rows_per_chunk;
  const uint64_t r_end = min(n_frames, r_begin + rows_per_chunk);
  v4f64 acc[ZT_MT][NT];
#pragma unroll
  for (int i = 0; i < ZT_MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  // two groups of four frames are in flight while a third is multiplied
  double a1[ZT_MT], a2[ZT_MT];
  float b1[NT], b2[NT];
  auto load = [&](uint64_t f0, double (&a_n)[ZT_MT], float (&b_n)[NT]) {
    const uint64_t f = f0 + lk;
#pragma unroll
    for (int i = 0; i < ZT_MT; i++) {
      const uint32_t o = og + i * 16 + li;
      a_n[i] = (f < r_end && o < n_out) ? Zm[f * n_out + o] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const uint32_t c = j * 16 + li;
      b_n[j] = (f < r_end && c < W) ? F[f * W + c] : 0.0f;
    }
  };
  load(r_begin, a1, b1);
  load(r_begin + 4, a2, b2);
  for (uint64_t f0 = r_begin; f0 < r_end; f0 += 8) {
    double a[ZT_MT], b[NT];
#pragma unroll
    for (int i = 0; i < ZT_MT; i++) a[i] = a1[i];
#pragma unroll
    for (int j = 0; j < NT; j++) b[j] = (double)b1[j];
    load(f0 + 8, a1, b1);
#pragma unroll
    for (int i = 0; i < ZT_MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < ZT_MT; i++) a[i] = a2[i];
#pragma unroll
    for (int j = 0; j < NT; j++) b[j] = (double)b2[j];
    load(f0 + 12, a2, b2);
#pragma unroll
    for (int i = 0; i < ZT_MT; i++)
#pragma unroll
      for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  double* out = slab + (size_t)blockIdx.x * n_out * W;
#pragma unroll
  for (int i = 0; i < ZT_MT; i++)
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t o = og + i * 16 + lk + 4 * r, c = j * 16 + li;
        if (o < n_out && c < W) out[(size_t)o * W + c] = acc[i][j][r];
      }
}

// slab: [n_chunks][n_out][W]
void launch_ztf(hipStream_t st, const double* Zm, uint32_t n_out, const float* F, uint32_t W, uint64_t n_frames,
                uint64_t rows_per_chunk, uint32_t n_chunks, double* slab, int narrow) {
  if (n_frames == 0 || n_chunks == 0) return;
  const uint32_t mt = narrow ? 1 : 4;
  dim3 grid(n_chunks, (n_out + 16 * mt - 1) / (16 * mt));
  const uint32_t nt = (W + 15) / 16;
#define ZT_GO(N)                                                                                                           \
  do {                                                                                                                     \
    if (narrow) hipLaunchKernelGGL((k_ztf<N, 1>), grid, dim3(64), 0, st, Zm, n_out, F, W, n_frames, rows_per_chunk, slab);  \
    else hipLaunchKernelGGL((k_ztf<N, 4>), grid, dim3(64), 0, st, Zm, n_out, F, W, n_frames, rows_per_chunk, slab);         \
  } while (0)
  if (nt <= 1) ZT_GO(1);
  else if (nt == 2) ZT_GO(2);
  else if (nt == 3) ZT_GO(3);
  else if (nt == 4) ZT_GO(4);
  else ZT_GO(5);
#undef ZT_GO
}

// ------------------------------------------------------------------------------------------
// k_post_z: the posterior pass and k_lin_z in one walk (linear-domain path, scrf_dplin.hip):
//   R[(t,d)][o] = Y - p[t-d][o] * es[(t,d)][o] * b[t][o] * exp(gp[t-d] + smax[(t,d)] + gb[t] - Zx)
// overwrites es in place (computeExpF :673-702) and is folded into the five register windows of
// Z on the spot, so R is written once and not read back for the per-frame sums.  One wavefront
// per (utterance, 64 outputs); the last D alpha-plus-trans vectors sit in an LDS ring; the D
// scale factors of a frame are computed by lanes 0..D-1 and broadcast through LDS.
// Also produces the per-frame numerator terms (gradbuilder :388-469).
// ------------------------------------------------------------------------------------------
// LA (SCRF_PREC_FASTLIN): a sixth group Z_avg[f][o] = sum over the windows (t, d) that contain frame f of R / d -- the
// expected-count side of the linear window average (its counts are Z_avg^T F, one more group of k_ztf).  A window reaches
// D - 1 frames back, so the open sums of the last D frames sit in an LDS ring (the register file is full of the five
// sample windows): at frame t the suffix sums u_j = sum_{d > j} R(t, d) / d are added to the slots of frames t - j, and
// frame t - D + 1 retires.  RW = lanes of the rings that exist (48 when L <= 48: 8 wavefronts per CU keep their LDS).
// SPLIT: gridDim.z wavefronts per (utterance, 64 outputs), each walking `seg_len` frames (>= 2 D): a launch of few
// utterances is a handful of wavefronts whose frame step is bare load latency (BASELINE config 3, 256 utterances: one
// wavefront per CU, 4 us per frame).  A segment warms its ring of alpha-plus-trans vectors from the stored ones; the
// per-frame sums of frames near a segment boundary get windows from both sides and are ADDED into the zeroed Z (LzWin).
template <int DMAX, int LA, int RW, int SPLIT>
__global__ __launch_bounds__(64, 2) void k_post_z(ScrfLayout lay, ScrfBatchView bv, uint32_t u0,
                                                  const uint32_t* __restrict__ next_lab,
                                                  const double* __restrict__ s_true, const double* __restrict__ M,
                                                  int m_per_frame, double* __restrict__ ES,
                                                  const double* __restrict__ smax, ScrfDpLin o_,
                                                  const double* __restrict__ zx, double* __restrict__ numer_f,
                                                  int* __restrict__ status, double* __restrict__ Z,
                                                  double* __restrict__ mass_s, int seg_len) {
  __shared__ double pring[DMAX * RW];
  __shared__ double zring[LA ? DMAX * RW : 1];
  __shared__ double fsb[DMAX < 64 ? DMAX : 64];
  const uint32_t D = lay.D, L = lay.L;
  const uint32_t u = u0 + blockIdx.x;
  const int T = (int)bv.T[u];
  const uint32_t lane = threadIdx.x;
  const uint32_t o = blockIdx.y * 64 + lane;
  const bool act = o < L;
  const uint32_t oc = act ? o : L - 1;
  const uint64_t f_base = bv.frame_off[u] - bv.frame_off[u0], s_base = bv.seg_off[u] - bv.seg_off[u0];
  const uint64_t gf0 = bv.frame_off[u];
  double* ESu = ES + s_base * L + oc;
  const double* smu = smax + s_base;
  const size_t zs = (size_t)(LA ? 6 : 5) * L;
  double* Zu = Z + f_base * (uint64_t)zs + oc;
  const double Zx = zx[u];
  const double LN_MAX = 709.782712893384;
  LzWin<0, DMAX> w0; LzWin<1, DMAX> w1; LzWin<2, DMAX> w2; LzWin<3, DMAX> w3; LzWin<4, DMAX> w4;
  w0.clear(); w1.clear(); w2.clear(); w3.clear(); w4.clear();
  const uint32_t rl = lane < (uint32_t)RW ? lane : RW - 1;   // ring column (lanes past RW are never active)
  const bool rw_ok = lane < (uint32_t)RW;
  if (LA) {
#pragma unroll
    for (int j = 0; j < DMAX; j++) if (rw_ok) zring[j * RW + lane] = 0.0;
  }
  int err = 0;
  const int fa = SPLIT ? (int)blockIdx.z * seg_len : 0;           // this wavefront's frames: fa .. fb - 1
  if (SPLIT && fa >= T) return;
  const int fb = SPLIT ? (fa + seg_len < T ? fa + seg_len : T) : T;
  // frames whose sums this segment owns alone: from fa on, and (unless it is the last one) D frames short of its end
  const int own_lo = fa, own_hi = (fb == T) ? 0x7fffffff : fb - (int)D + 1;
  int slot = SPLIT ? fa % (int)D : 0;   // ring slot that will receive p[t] (= t mod D)
  if (SPLIT && fa > 0) {
    for (int j = 1; j <= (int)D && j <= fa; j++)
      if (rw_ok) pring[((fa - j) % (int)D) * RW + lane] = o_.p[(f_base + fa - j) * L + oc];
  }
#pragma unroll 1
  for (int t = fa; t < fb; t++) {
    const uint32_t nd = scrf_node_max_dur((uint32_t)t, D), np = scrf_num_prev((uint32_t)t, D);
    const uint64_t row0 = scrf_seg_base((uint32_t)t, D);
    double r[DMAX];
    // streamed once each way: non-temporal loads and stores (5.85 -> 5.34 ms on one box; the stores are what counts)
#pragma unroll
    for (int d0 = 0; d0 < DMAX; d0++) r[d0] = ((uint32_t)d0 < nd) ? __builtin_nontemporal_load(&ESu[(row0 + d0) * L]) : 0.0;
    const double b = o_.b[(f_base + t) * L + oc];
    const double pnew = (t + 1 < T) ? o_.p[(f_base + t) * L + oc] : 0.0;
    // scale factor of duration d0 = lane
    {
      double x = -INFINITY;
      if (lane < nd) {
        x = ((lane < np) ? o_.gp[f_base + t - 1 - lane] : 0.0) + smu[row0 + lane] + o_.gb[f_base + t] - Zx;
        if (x >= LN_MAX) err = SCRF_ERR_NUMERIC;
      }
      if (lane < (DMAX < 64 ? DMAX : 64)) fsb[lane] = (lane < nd) ? exp(x) : 0.0;
    }
    const uint32_t lab = bv.labels ? bv.labels[gf0 + t] : SCRF_LAB_BAD;
    uint32_t al = SCRF_LAB_BAD, ld = SCRF_LAB_BAD;
    if (lab != SCRF_LAB_BAD) {
      if (lab >= L * D) err = SCRF_ERR_BAD_LABEL;
      al = lab % L;
      ld = lab / L + 1;
    }
    double gs = 0.0;
#pragma unroll
    for (int d0 = 0; d0 < DMAX; d0++) {
      int ps = slot - 1 - d0;            // ring slot of p[t-1-d0]
      if (ps < 0) ps += (int)D;
      const double pv = ((uint32_t)d0 < np) ? pring[(((uint32_t)d0 < np) ? ps : 0) * RW + rl] : 1.0;
      const double g = (pv * r[d0]) * (b * fsb[d0]);
      const double y = (o == al && (uint32_t)d0 + 1 == ld) ? 1.0 : 0.0;
      const double rv = ((uint32_t)d0 < nd) ? y - g : 0.0;
      if (act && (uint32_t)d0 < nd) __builtin_nontemporal_store(rv, &ESu[(row0 + d0) * L]);
      r[d0] = rv;
      gs += g;   // r[d0] = 0 and fsb[d0] = 0 past nd
    }
    {  // state posterior mass of the node (k_mass_check compares it with the transition mass)
      const double m = wave_sum_f64_dpp(act ? gs : 0.0);
      if (lane == 0) {
        if (gridDim.y == 1) mass_s[f_base + t] = m;
        else atomicAdd(&mass_s[f_base + t], m);   // L > 64: one partial sum per 64 outputs (buffer zeroed by the caller)
      }
    }
    if (LA) {
      // u runs over the suffix sums (durations descending); slot of frame t - d0 = (slot - d0) mod D
      double usum = 0.0;
#pragma unroll
      for (int d0 = DMAX - 1; d0 >= 0; d0--) {
        if ((uint32_t)d0 >= D) continue;
        usum = fma(r[d0], 1.0 / (double)(d0 + 1), usum);   // r is 0 past the node's durations
        int zsl = slot - d0;
        if (zsl < 0) zsl += (int)D;
        // LDS atomic without return (ds_add_f64): no read-add-write round trip in the frame's dependent chain; one
        // wavefront owns the ring, so the additions still happen in program order
        if (rw_ok) __hip_atomic_fetch_add(&zring[zsl * RW + lane], usum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
      // frame t - D + 1 has seen its last window
      int zf = slot + 1 == (int)D ? 0 : slot + 1;
      const double zv = zring[zf * RW + rl];
      if (act && t + 1 >= (int)D) {
        const int f = t + 1 - (int)D;
        double* q = &Zu[(size_t)f * zs + 5 * (size_t)L];
        if (!SPLIT || (f >= own_lo && f < own_hi)) __builtin_nontemporal_store(zv, q);
        else unsafeAtomicAdd(q, zv);
      }
      if (rw_ok) zring[zf * RW + lane] = 0.0;
    }
    w0.add(r); w1.add(r); w2.add(r); w3.add(r); w4.add(r);
    if (act && SPLIT) {
      w0.retire_seg(Zu, t, zs, own_lo, own_hi);
      w1.retire_seg(Zu + L, t, zs, own_lo, own_hi);
      w2.retire_seg(Zu + 2 * (size_t)L, t, zs, own_lo, own_hi);
      w3.retire_seg(Zu + 3 * (size_t)L, t, zs, own_lo, own_hi);
      w4.retire_seg(Zu + 4 * (size_t)L, t, zs, own_lo, own_hi);
    } else if (act) {
      w0.retire(Zu, t, zs);
      w1.retire(Zu + L, t, zs);
      w2.retire(Zu + 2 * (size_t)L, t, zs);
      w3.retire(Zu + 3 * (size_t)L, t, zs);
      w4.retire(Zu + 4 * (size_t)L, t, zs);
    } else {
      w0.retire(Zu, -1000000, zs); w1.retire(Zu, -1000000, zs); w2.retire(Zu, -1000000, zs);
      w3.retire(Zu, -1000000, zs); w4.retire(Zu, -1000000, zs);
    }
    if (rw_ok) pring[slot * RW + lane] = pnew;
    slot = (slot + 1 == (int)D) ? 0 : slot + 1;
    if (lane == 0 && blockIdx.y == 0) {
      double nodeLi = 0.0;
      if (lab != SCRF_LAB_BAD && err == 0) {
        if (ld <= nd) nodeLi += s_true[f_base + t];
        const uint32_t nl = next_lab[gf0 + t];
        if (t + 1 < T && nl != SCRF_LAB_BAD) {
          if (nl >= L * D) err = SCRF_ERR_BAD_LABEL;
          else {
            const double* Mn = M + (m_per_frame ? (f_base + t + 1) * (size_t)L * L : 0);
            nodeLi += Mn[(size_t)al * L + nl % L];
          }
        }
      }
      numer_f[f_base + t] = nodeLi;
    }
  }
  if (act && SPLIT) {
    w0.flush_seg(Zu, fb, zs, own_lo, own_hi);
    w1.flush_seg(Zu + L, fb, zs, own_lo, own_hi);
    w2.flush_seg(Zu + 2 * (size_t)L, fb, zs, own_lo, own_hi);
    w3.flush_seg(Zu + 3 * (size_t)L, fb, zs, own_lo, own_hi);
    w4.flush_seg(Zu + 4 * (size_t)L, fb, zs, own_lo, own_hi);
    if (LA) {
      for (int j = 1; j < (int)D && j <= fb; j++) {
        const int f = fb - j;
        double* q = &Zu[(size_t)f * zs + 5 * (size_t)L];
        const double zv = zring[(f % (int)D) * RW + rl];
        if (f >= own_lo && f < own_hi) *q = zv;
        else unsafeAtomicAdd(q, zv);
      }
    }
  } else if (act) {
    w0.flush(Zu, T, zs);
    w1.flush(Zu + L, T, zs);
    w2.flush(Zu + 2 * (size_t)L, T, zs);
    w3.flush(Zu + 3 * (size_t)L, T, zs);
    w4.flush(Zu + 4 * (size_t)L, T, zs);
    if (LA) {
      // frames T - D + 1 .. T - 1 are still open in the ring
      for (int j = 1; j < (int)D && j <= T; j++) {
        const int f = T - j;
        Zu[(size_t)f * zs + 5 * (size_t)L] = zring[(f % (int)D) * RW + rl];
      }
    }
  }
  if (__any(err != 0) && lane == 0) atomicMax(&status[u], err > 0 ? err : SCRF_ERR_NUMERIC);
}

void launch_post_z(hipStream_t st, const ScrfLayout& lay, ScrfBatchView bv, uint32_t u0, uint32_t n_utts,
                   const uint32_t* next_lab, const double* s_true, const double* M, int m_per_frame, double* ES,
                   const double* smax, const ScrfDpLin& o, const double* zx, double* numer_f, int* status, double* Z,
                   double* mass_s, int la, uint32_t t_max, uint64_t n_frames) {
  if (n_utts == 0) return;
  dim3 grid(n_utts, (lay.L + 63) / 64);
  // few wavefronts (a small minibatch): several segments per utterance, about four wavefronts per CU in all (half the
  // kernel's occupancy), no segment shorter than 2 D frames (SCRF_POSTZ_SPLIT=0: never).  Config 3, 256 utterances:
  // 1.21 -> 0.50 ms; config 2's shape at 64 utterances: 1.90 -> 0.38 ms
  static const bool split_ok = !(getenv("SCRF_POSTZ_SPLIT") && atoi(getenv("SCRF_POSTZ_SPLIT")) == 0);
  int seg_len = 0;
  uint32_t nz = 1;
  const uint64_t waves = (uint64_t)grid.x * grid.y;
  if (split_ok && waves <= 2 * 256 && t_max >= 4 * lay.D) {
    const uint32_t want = (uint32_t)((4 * 256 + waves - 1) / waves);
    seg_len = (int)((t_max + want - 1) / want);
    if (seg_len < (int)(2 * lay.D)) seg_len = (int)(2 * lay.D);
    nz = (uint32_t)((t_max + seg_len - 1) / seg_len);
  }
  if (nz > 1) {
    grid.z = nz;
    hipMemsetAsync(Z, 0, sizeof(double) * n_frames * (size_t)(la ? 6 : 5) * lay.L, st);
  }
#define PZ_GO3(N, A, R)                                                                                                           \
  do {                                                                                                                            \
    if (nz > 1) hipLaunchKernelGGL((k_post_z<N, A, R, 1>), grid, dim3(64), 0, st, lay, bv, u0, next_lab, s_true, M, m_per_frame, ES, smax, o, zx, numer_f, status, Z, mass_s, seg_len); \
    else hipLaunchKernelGGL((k_post_z<N, A, R, 0>), grid, dim3(64), 0, st, lay, bv, u0, next_lab, s_true, M, m_per_frame, ES, smax, o, zx, numer_f, status, Z, mass_s, 0); \
  } while (0)
#define PZ_GO(N)                                              \
  do {                                                        \
    if (la && lay.L <= 48) PZ_GO3(N, 1, 48);                  \
    else if (la) PZ_GO3(N, 1, 64);                            \
    else PZ_GO3(N, 0, 64);                                    \
  } while (0)
  if (lay.D <= 8) PZ_GO(8);
  else if (lay.D <= 16) PZ_GO(16);
  else if (lay.D <= 25) PZ_GO(25);
  else if (lay.D <= 32) PZ_GO(32);
  else PZ_GO(40);
#undef PZ_GO
#undef PZ_GO3
}

// ------------------------------------------------------------------------------------------
// k_expf_fused: slab[block][o][f] = sum_rows R[row][o] * x[row][f] over the dense column groups
// f in [avg | max | min | onehot(d) | bias]   (M = outputs, N = columns, K = rows).
// Persistent workgroups (256 threads, two per CU) walk 64-row tiles (any 64 consecutive windows of
// an utterance); the dense columns of the tile are rebuilt in LDS, R is staged through LDS once
// per tile.  The next tile's R and raw frames are fetched into registers under the MFMAs (issued
// after the barrier: a barrier drains outstanding loads on gfx9).  The 3 x n_ct output tiles are
// dealt round-robin to the 4 waves (slot q = wave + 4j -> output tile q % 3, column tile q / 3);
// the partial sums stay in registers for the whole launch.  The k-loop is software pipelined by
// hand: operands of step ks+1 are in flight while the MFMAs of step ks issue.
// ------------------------------------------------------------------------------------------
#define FE_NT 256
#define FE_NRP 15       // R elements per thread in prefetch registers (76*48 / 256, rounded up)
#define FE_NFP 6        // raw-frame floats per thread held in prefetch registers

// one wave's share of a tile: slot j = output tile (wave + j) % 3 (== (wave + 4j) % 3), column tile
// (wave + 4j) / 3.  The three R^T fragments are loaded in the wave's rotation, so slot j always
// multiplies register a[j % 3]; every LDS address is a per-slot base + a compile-time offset.
template <int NT, int F32, uint32_t XS, int NKS>
__device__ __forceinline__ void fe_mfma_tile(const double* Rs, const float* Xs, uint32_t wave, uint32_t lk,
                                             uint32_t li, uint32_t n_ot, v4f64* acc, v4f32* acc32) {
  const float* ap32[3];
  const double* ap[3];
  const float* bp[NT];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const uint32_t n = (wave + i) % 3;
    ap[i] = Rs + lk * FE_RS + n * 16 + li;
    ap32[i] = (const float*)Rs + lk * FE_RSF + n * 16 + li;
  }
  // slots past the last output tile repeat its column tile (their sums are never written)
#pragma unroll
  for (int j = 0; j < NT; j++) bp[j] = Xs + lk * XS + li + (min(wave + 4 * j, n_ot - 1) / 3) * 16;
  // software pipeline, two steps deep: while the MFMAs of step ks issue, the B operands of step
  // ks+1 are converted fp32 -> fp64 and the raw operands of step ks+2 are loaded.  The group
  // barriers pin the interleaving (one MFMA, one conversion, one or two LDS reads) so that the
  // matrix pipe does not idle behind a block of conversions and loads at every step.
  double a_c[3], a_n[3], bd_c[F32 ? 1 : NT], bd_n[F32 ? 1 : NT];
  float af_c[3], af_n[3], b_c[F32 ? NT : 1], b_r[NT];
#pragma unroll
  for (int i = 0; i < 3; i++) { if (F32) af_c[i] = *ap32[i]; else a_c[i] = *ap[i]; }
#pragma unroll
  for (int j = 0; j < NT; j++) { const float x = *bp[j]; if (F32) b_c[F32 ? j : 0] = x; else bd_c[F32 ? 0 : j] = (double)x; }
#pragma unroll
  for (int i = 0; i < 3; i++) { if (F32) af_n[i] = ap32[i][4 * FE_RSF]; else a_n[i] = ap[i][4 * FE_RS]; }
#pragma unroll
  for (int j = 0; j < NT; j++) b_r[j] = bp[j][4 * XS];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) {
    double a_nn[3];
    float af_nn[3];
#pragma unroll
    for (int j = 0; j < NT; j++) {
      if (F32) acc32[F32 ? j : 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af_c[j % 3], b_c[F32 ? j : 0], acc32[F32 ? j : 0], 0, 0, 0);
      else acc[F32 ? 0 : j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_c[j % 3], bd_c[F32 ? 0 : j], acc[F32 ? 0 : j], 0, 0, 0);
      // operand j of step ks+1: convert; of step ks+2: load
      if (!F32) bd_n[F32 ? 0 : j] = (double)b_r[j];
      const float held = b_r[j];
      if (ks + 2 < NKS) b_r[j] = bp[j][(ks + 2) * 4 * XS];
      if (F32) b_c[F32 ? j : 0] = held;   // (consumed by the next step's MFMA j, after this step's)
      if (j < 3 && ks + 2 < NKS) {
        if (F32) af_nn[j] = ap32[j][(ks + 2) * 4 * FE_RSF]; else a_nn[j] = ap[j][(ks + 2) * 4 * FE_RS];
      }
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
      if (!F32) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // one conversion
      if (j < 3) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);  // its LDS reads
      else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 3; i++) { a_c[i] = a_n[i]; af_c[i] = af_n[i]; a_n[i] = a_nn[i]; af_n[i] = af_nn[i]; }
    if (!F32) {
#pragma unroll
      for (int j = 0; j < NT; j++) bd_c[F32 ? 0 : j] = bd_n[F32 ? 0 : j];
    }
  }
}

template <int NT, int DMAX, int F32, int NKS>
__global__ __launch_bounds__(FE_NT, 2) void k_expf_fused(ScrfFusedArgs fa, ScrfLayout lay, const double* __restrict__ R,
                                                         uint32_t n_out, uint64_t n_tiles, uint32_t n_ct,
                                                         uint32_t nfmax, double* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const uint32_t W = fa.W, D = lay.D;
  // row stride of the column image: 80 / 144 / 208 == 16 (mod 64), so the 4 k-rows of a B fragment
  // hit disjoint banks, and compile-time, so every operand address is base + immediate
  constexpr uint32_t xs = (NT == 4 ? 5 : NT == 7 ? 9 : 13) * 16;
  float* Xs = (float*)fsm;                                    // [76 + dump row][xs]
  double* Rs = (double*)(Xs + (FE_ROWS + 1) * xs);            // [76][FE_RS] (floats when F32)
  float* Rsf = (float*)Rs;
  float* fr = (float*)(Rs + FE_ROWS * FE_RS);                 // [nfmax][W]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t o0 = blockIdx.y * 48;
  const uint32_t mW = fu_magic(W);
  const uint32_t ncol = 3 * W + D + (lay.use_sb ? 1 : 0);
  const uint32_t n_ot = 3 * n_ct;

  for (uint32_t i = tid; i < (FE_ROWS + 1) * xs; i += FE_NT) Xs[i] = 0.0f;

  v4f64 acc[F32 ? 1 : NT];
  v4f32 acc32[F32 ? NT : 1];
#pragma unroll
  for (int j = 0; j < (F32 ? 1 : NT); j++) acc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int j = 0; j < (F32 ? NT : 1); j++) acc32[j] = (v4f32){0.0f, 0.0f, 0.0f, 0.0f};

  // prefetch registers: R element e = tid + 256*q of the [76][48] tile, raw-frame float tid + 256*q
  double rp[FE_NRP];
  float fp[FE_NFP];
  const uint32_t nfw = nfmax * W;
  auto prefetch = [&](const FuTile& t) {
#pragma unroll
    for (int q = 0; q < FE_NRP; q++) {
      const uint32_t e = tid + FE_NT * q, row = e / 48, ol = e % 48;
      rp[q] = (row < t.nrows && o0 + ol < n_out) ? R[(t.row0 + row) * n_out + o0 + ol] : 0.0;
    }
    const float* src = fa.frames + (fa.frame_base + t.fr0) * (uint64_t)W;
    const uint32_t n = (t.t0 + t.nfr - t.f0) * W;
#pragma unroll
    for (int q = 0; q < FE_NFP; q++) fp[q] = (tid + FE_NT * q < n) ? src[tid + FE_NT * q] : 0.0f;
  };
  // tiles k*G .. (k+1)*G - 1 are in flight together; the XCD-aware slot order puts neighbours (tiles of one utterance,
  // whose raw-frame ranges overlap) on one XCD
  uint64_t tile = gridDim.y == 1 ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  FuTile ft, nft;
  ScrfTileDesc dn;   // descriptor of the tile after next
  if (tile < n_tiles) {
    ft = fu_tile(fa, fa.tiles[fa.tile0 + tile]);
    prefetch(ft);
    if (tile + gridDim.x < n_tiles) dn = fa.tiles[fa.tile0 + tile + gridDim.x];
  }
  uint32_t dur_prev = 0;   // thread tid < 64: duration whose one-hot column is set in row tid
  float* dump = Xs + FE_ROWS * xs;
  FU_SETPRIO(1);
  __syncthreads();
#if FU_PROF
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#endif
  while (tile < n_tiles) {
#if FU_PROF
    if (threadIdx.x == 0) atomicAdd(&fu_prof[14], 1ull);
#endif
    // stage: raw frames f0 .. t0+nfr-1, R, the rows' one-hot duration and bias columns
    {
#pragma unroll
      for (int q = 0; q < FE_NFP; q++) if (tid + FE_NT * q < nfw) fr[tid + FE_NT * q] = fp[q];
      if (nfw > FE_NT * FE_NFP) {   // very wide streams: the rest comes straight from memory
        const float* src = fa.frames + (fa.frame_base + ft.fr0) * (uint64_t)W;
        const uint32_t n = (ft.t0 + ft.nfr - ft.f0) * W;
        for (uint32_t i = tid + FE_NT * FE_NFP; i < n; i += FE_NT) fr[i] = src[i];
      }
#pragma unroll
      for (int q = 0; q < FE_NRP; q++) {
        const uint32_t e = tid + FE_NT * q, row = e / 48, ol = e % 48;
        if (row >= FE_ROWS) continue;   // 15 x 256 elements cover 80 rows
        if (F32) Rsf[row * FE_RSF + ol] = (float)rp[q];
        else Rs[row * FE_RS + ol] = rp[q];
      }
      if (tid < ft.nrows) {
        const uint32_t r = ft.r0 + tid;
        uint32_t t = ft.t0;
        while ((uint32_t)scrf_seg_base(t + 1, D) <= r) t++;
        const uint32_t d = r - (uint32_t)scrf_seg_base(t, D) + 1;
        float* xr = Xs + tid * xs + 3 * W;
        if (dur_prev) xr[dur_prev - 1] = 0.0f;
        xr[d - 1] = 1.0f;
        dur_prev = d;
        if (lay.use_sb) xr[D] = 1.0f;
      }
    }
    __syncthreads();
    FU_STAMP(8);    // expf: stage
    // rebuild avg | max | min of the tile's rows: task = (statistic, frame, column), statistic-major, so that the
    // tile's 3 nfr W tasks fill the workgroup's four waves and a wave runs one statistic's code (two at a seam).
    // Tiles are whole frames; in the steady state (every duration exists) the LDS offsets are immediates.
    {
      const uint32_t nfW = ft.nfr * W, mfW = fu_magic(nfW);
      const bool full = FU_FULLSCAN && D == (uint32_t)DMAX && ft.t0 + 1 >= D;
      for (uint32_t i = tid; i < 3 * nfW; i += FE_NT) {
        const uint32_t st = fu_div(i, mfW), rem = i - st * nfW;
        const uint32_t tl = fu_div(rem, mW), c = rem - tl * W;
        const uint32_t t = ft.t0 + tl;
        const uint32_t at = (t - ft.f0) * W + c;
        float v[DMAX];
        if (full) {
          float* o = Xs + tl * (DMAX * xs) + st * W + c;
          fu_load_vals_full<DMAX>(fr, at, W, v);
          if (st == 0) fu_scan_avg_full<DMAX, xs>(v, o);
          else if (st == 1) fu_scan_ext_full<DMAX, 1, xs>(v, o);
          else fu_scan_ext_full<DMAX, 0, xs>(v, o);
        } else {
          const uint32_t nd = scrf_node_max_dur(t, D);
          const int32_t lbase = (int32_t)scrf_seg_base(t, D) - (int32_t)ft.r0;
          const uint32_t d_lo = lbase < 0 ? (uint32_t)(1 - lbase) : 1u;
          const uint32_t d_hi = min(nd, (uint32_t)((int32_t)ft.nrows - lbase));
          fu_load_vals<DMAX>(fr + at, W, nd, v);
          float* o = Xs + lbase * (int32_t)xs + st * W + c;   // row of d = 1 (may lie before the tile: dumped)
          float* dmp = dump + st * W + c;
          if (st == 0) fu_scan_avg<DMAX>(v, o, xs, d_lo, d_hi, dmp);
          else if (st == 1) fu_scan_ext<DMAX, 1>(v, o, xs, d_lo, d_hi, dmp);
          else fu_scan_ext<DMAX, 0>(v, o, xs, d_lo, d_hi, dmp);
        }
      }
    }
    __syncthreads();
    FU_STAMP(9);    // expf: scans
    // next tile's R and frames, and the descriptor after that, are fetched under the MFMAs
    const uint64_t ntile = tile + gridDim.x;
    if (ntile < n_tiles) {
      nft = fu_tile(fa, dn);
      prefetch(nft);
      if (ntile + gridDim.x < n_tiles) {
        // a vector load on purpose: a scalar load would share lgkmcnt with the LDS operand reads
        // below and stall the first MFMA for a full memory round trip
        uint64_t ti = fa.tile0 + ntile + gridDim.x;
        asm volatile("" : "+v"(ti));
        dn = fa.tiles[ti];
      }
    }
    FU_SETPRIO(0);
    fe_mfma_tile<NT, F32, xs, NKS>(Rs, Xs, wave, lk, li, n_ot, acc, acc32);   // depth 4 NKS rows (fused_expf_nks)
    FU_SETPRIO(1);
    __syncthreads();
    FU_STAMP(10);   // expf: prefetch issue + MFMAs
    tile = ntile;
    ft = nft;
  }
  // slab[blockIdx.x][o][f]
  double* out = slab + (uint64_t)blockIdx.x * n_out * ncol;
#pragma unroll
  for (int j = 0; j < NT; j++) {
    const uint32_t q = wave + 4 * j;
    if (q >= n_ot) continue;
    const uint32_t n = q % 3, ct = q / 3;
    const uint32_t f = ct * 16 + li;
    if (f >= ncol) continue;
    const double sc = (lay.use_sb && f == 3 * W + D) ? lay.sbv : 1.0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t o = o0 + n * 16 + (F32 ? 4 * lk + r : lk + 4 * r);
      if (o >= n_out) continue;
      const double v = F32 ? (double)acc32[F32 ? j : 0][r] : acc[F32 ? 0 : j][r];
      out[(uint64_t)o * ncol + f] = v * sc;
    }
  }
}

// ------------------------------------------------------------------------------------------
// k_expf_fused_ws: the same contraction with the workgroup's waves SPECIALISED.  k_expf_fused alternates three
// barrier-separated phases per tile (stage, scans, MFMAs), and a wave that stages or scans shares its SIMD's issue
// with a wave of the other workgroup that is inside its MFMA loop: the vector phases stretch to several times their
// instruction time and the matrix pipe idles in between (in-kernel stamps: stage + scans = 37 % of a tile).  Here one
// 512-thread workgroup owns the CU: waves 0-3 (one per SIMD) do nothing but MFMAs on one LDS image pair while waves
// 4-7 (one per SIMD, s_setprio 1) build the next tile's pair -- R from memory, avg | max | min scanned straight from
// the raw frames in memory (L2-resident; no staged frame image, so the producers need no barrier among themselves),
// one-hot and bias columns -- and the two halves meet at ONE barrier per tile.
// LDS: 2 x ([76+1][xs] floats + [76][48] doubles) = 147 KB at config 2.
// ------------------------------------------------------------------------------------------
// Round 4: the one-hot duration and bias columns are no longer multiplied.  Their counts are plain sums of R over the
// rows of one duration, sum_t R[(t, d)][o]: producer thread (output ol, slot ds) reads the finished R image of the tile
// the consumers are working on and keeps the sums of the durations ds + 1, ds + 6, ... in registers for the whole launch
// (15 LDS reads per tile at config 2) -- an exact re-association, and 26 of the 143 dense columns less for the MFMAs.
// G0 = first dense group: 0 = [avg | max | min], 1 = [max | min] (SCRF_PREC_FASTLIN: the average goes through Z_avg).
// Outputs: slab[block][o][(3 - G0) W] and dslab[block][o][D + bias].
#define FW_NT 512
#define FW_DSL 5        // duration slots: producer threads 0..239 = (ol = ptid % 48, ds = ptid / 48)
// ROWS = window rows per tile (76, or 100 where the two image pairs still fit 160 KB: fewer tiles, fewer barriers and
// producer round trips per row); NKS = ceil(rows used / 4).
// (register cap 208 for the narrow forms: two such wavefronts per SIMD leave 96 registers per lane, which is what lets
// the narrow k_ztf of the side stream -- 64 registers -- be resident on the same SIMD)
template <int NT, int DMAX, int NKS, int G0, int ROWS>
__device__ __forceinline__ void expf_fused_ws_body(const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* __restrict__ R,
                                                   uint32_t n_out, uint64_t n_tiles, uint32_t n_ct,
                                                   double* __restrict__ slab, double* __restrict__ dslab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const uint32_t W = fa.W, D = lay.D;
  constexpr uint32_t xs = (NT == 4 ? 5 : NT <= 7 ? 9 : 13) * 16;
  constexpr uint32_t XB = (ROWS + 1) * xs;   // floats per column image (row ROWS: dump row)
  constexpr uint32_t RB = ROWS * FE_RS;
  constexpr int NRP = (ROWS * 48 + 255) / 256;   // R elements per producer thread      // doubles per R image
  float* Xs0 = (float*)fsm;                     // [2][XB]
  double* Rs0 = (double*)(Xs0 + 2 * XB);        // [2][RB]
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  const uint32_t li = lane & 15, lk = lane >> 4;
  const uint32_t o0 = blockIdx.y * 48;
  // gridDim.z > 1: one dense statistic per workgroup column (wide streams: (3 - G0) W columns do not fit one LDS image
  // pair / 13 column tiles) -- this workgroup builds and contracts statistic st0 only and writes columns
  // [blockIdx.z W, (blockIdx.z + 1) W) of the slab rows; the duration sums are taken by the z = 0 column
  const uint32_t st0 = G0 + (gridDim.z > 1 ? blockIdx.z : 0u);
  const uint32_t ngrp = gridDim.z > 1 ? 1u : (uint32_t)(3 - G0);
  const uint32_t ncol = ngrp * W, ncol_all = (3 - G0) * W, zoff = gridDim.z > 1 ? blockIdx.z * W : 0u;
  const uint32_t n_ot = 3 * n_ct;
  for (uint32_t i = tid; i < 2 * XB; i += FW_NT) Xs0[i] = 0.0f;
  for (uint32_t i = tid; i < 2 * RB; i += FW_NT) Rs0[i] = 0.0;

  v4f64 acc[NT];
  v4f32 acc32_unused[1];
#pragma unroll
  for (int j = 0; j < NT; j++) acc[j] = (v4f64){0.0, 0.0, 0.0, 0.0};

  const uint64_t first = (gridDim.y == 1 && gridDim.z == 1) ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  const uint64_t G = gridDim.x;
  // ---- producer state
  const uint32_t ptid = tid - 256;
  const uint32_t mW = fu_magic(W);
  ScrfTileDesc dn;                         // descriptor of the tile the producers build next
  if (producer && first < n_tiles) dn = fa.tiles[fa.tile0 + first];
  // per-duration sums of R (one-hot duration counts): durations dsl + 1 + FW_DSL q of output o0 + dol
  constexpr int NDA = (DMAX + FW_DSL - 1) / FW_DSL;
  double da[NDA];
#pragma unroll
  for (int q = 0; q < NDA; q++) da[q] = 0.0;
  const uint32_t dol = ptid % 48, dsl = ptid / 48;
  uint32_t ct0 = 0, cnfr = 0, cr0 = 0;     // frames and first row of the tile whose images the consumers hold
  auto dur_sums = [&](uint32_t buf) {
    if (ptid >= 48 * FW_DSL || blockIdx.z != 0) return;
    const double* Rs = Rs0 + buf * RB + dol;
    for (uint32_t tl = 0; tl < cnfr; tl++) {
      const uint32_t t = ct0 + tl, nd = scrf_node_max_dur(t, D);
      const double* Rt = Rs + ((uint32_t)scrf_seg_base(t, D) - cr0) * FE_RS;
#pragma unroll
      for (int q = 0; q < NDA; q++) {
        const uint32_t d0 = dsl + FW_DSL * q;
        if (d0 < nd) da[q] += Rt[d0 * FE_RS];
      }
    }
  };
  auto build = [&](uint32_t buf, uint64_t tile_next, uint32_t sum_buf) {
    // fills image pair `buf` with the tile whose descriptor sits in dn; then fetches tile_next's descriptor
    float* Xs = Xs0 + buf * XB;
    double* Rs = Rs0 + buf * RB;
    // the descriptor came through a vector load; everything derived from it is wave-uniform, and saying so keeps the
    // tile's base addresses in scalar registers (global loads with a 32-bit lane offset instead of 64-bit lane
    // addresses: with those the allocator ran out of registers and serialised the loads behind s_waitcnt vmcnt)
    FuTile ft = fu_tile(fa, dn);
    ft.t0 = __builtin_amdgcn_readfirstlane(ft.t0); ft.nfr = __builtin_amdgcn_readfirstlane(ft.nfr);
    ft.nrows = __builtin_amdgcn_readfirstlane(ft.nrows); ft.f0 = __builtin_amdgcn_readfirstlane(ft.f0);
    ft.r0 = __builtin_amdgcn_readfirstlane(ft.r0);
    ft.row0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(ft.row0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)ft.row0);
    ft.fr0 = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(ft.fr0 >> 32)) << 32) | __builtin_amdgcn_readfirstlane((uint32_t)ft.fr0);
    if (tile_next < n_tiles) {
      uint64_t ti = fa.tile0 + tile_next;
      asm volatile("" : "+v"(ti));   // a vector load: a scalar one would wait with the LDS traffic below
      dn = fa.tiles[ti];
    }
    double rp[NRP];
#pragma unroll
    for (int q = 0; q < NRP; q++) {
      const uint32_t e = ptid + 256 * q, row = e / 48, ol = e % 48;
      const double* Rt = R + ft.row0 * n_out + o0;   // uniform base, 32-bit lane offset
      const uint32_t offb = (row * n_out + ol) * 8u;
      rp[q] = (FU_ABL != 5 && row < ft.nrows && o0 + ol < n_out) ? __builtin_nontemporal_load((const double*)((const char*)Rt + offb)) : 0.0;
    }
    // the duration sums of the tile the consumers hold run under the loads just requested
    if (sum_buf < 2) dur_sums(sum_buf);
    // avg | max | min: task = (statistic, frame, column), values straight from the raw frames in memory.  A thread
    // takes tasks ptid and ptid + 256 together: both tasks' loads are in flight at once (one memory round trip per
    // tile instead of two).  (Interleaving the two scan chains by hand, so that the wave always has a second
    // instruction ready beside the MFMA wave, was measured too: 9.5 -> 11.3 ms.)
    {
      const float* gfr = fa.frames + (fa.frame_base + ft.fr0) * (uint64_t)W;
      const uint32_t nfW = ft.nfr * W, mfW = fu_magic(nfW);
      const bool full = FU_FULLSCAN && D == (uint32_t)DMAX && ft.t0 + 1 >= D;
      float* dump = Xs + ROWS * xs;
      // st = statistic (0 avg, 1 max, 2 min); its columns start at (st - G0) W
      auto decode = [&](uint32_t i, uint32_t& st, uint32_t& tl, uint32_t& c) {
        const uint32_t g = fu_div(i, mfW);
        st = g + st0;
        const uint32_t rem = i - g * nfW;
        tl = fu_div(rem, mW);
        c = rem - tl * W;
      };
      auto scan_full = [&](const float (&v)[DMAX], uint32_t st, float* o) {
        if (G0 == 0 && st == 0) fu_scan_avg_full<DMAX, xs>(v, o);
        else if (st == 1) fu_scan_ext_full<DMAX, 1, xs>(v, o);
        else fu_scan_ext_full<DMAX, 0, xs>(v, o);
      };
      if (FU_ABL == 6) {
      } else if (G0 == 1 && ngrp == 2 && full) {
        // max and min of a (frame, column) from ONE set of loads; the tile's nfr W tasks are dealt evenly to the four
        // producer wavefronts (a 100-row tile has 156 of them: 39 lanes of every wavefront, one pass each, instead of
        // 312 single-statistic tasks that gave the first wavefront two passes)
        const uint32_t per = (nfW + 3) >> 2, pw = ptid >> 6, pl = ptid & 63;
        for (uint32_t j = pl; j < per; j += 64) {
          const uint32_t i = pw * per + j;
          if (i < nfW) {
            const uint32_t tl = fu_div(i, mW), c = i - tl * W;
            float v[DMAX];
            fu_load_vals_full<DMAX>(gfr, (ft.t0 + tl - ft.f0) * W + c, W, v);
            float* o = Xs + tl * (DMAX * xs) + c;
            float amx = v[0], amn = v[0];
#pragma unroll
            for (int jj = 0; jj < DMAX; jj++) {
              amx = fu_vmax(amx, v[jj]);
              amn = fu_vmin(amn, v[jj]);
              o[jj * xs] = amx;
              o[jj * xs + W] = amn;
            }
          }
        }
      } else
      for (uint32_t i0 = ptid; i0 < ngrp * nfW; i0 += 512) {
        const uint32_t i1 = i0 + 256;
        const bool two = i1 < ngrp * nfW;
        uint32_t st0_, tl0, c0, st1, tl1, c1;
        decode(i0, st0_, tl0, c0);
        decode(two ? i1 : i0, st1, tl1, c1);
        if (full) {
          float v0[DMAX], v1[DMAX];
          fu_load_vals_full<DMAX>(gfr, (ft.t0 + tl0 - ft.f0) * W + c0, W, v0);
          fu_load_vals_full<DMAX>(gfr, (ft.t0 + tl1 - ft.f0) * W + c1, W, v1);
          scan_full(v0, st0_, Xs + tl0 * (DMAX * xs) + (st0_ - st0) * W + c0);
          if (two) scan_full(v1, st1, Xs + tl1 * (DMAX * xs) + (st1 - st0) * W + c1);
        } else {
          for (int h = 0; h < (two ? 2 : 1); h++) {
            const uint32_t st = h ? st1 : st0_, tl = h ? tl1 : tl0, c = h ? c1 : c0;
            const uint32_t t = ft.t0 + tl;
            const uint32_t at = (t - ft.f0) * W + c;
            float v[DMAX];
            const uint32_t nd = scrf_node_max_dur(t, D);
            const int32_t lbase = (int32_t)scrf_seg_base(t, D) - (int32_t)ft.r0;
            const uint32_t d_lo = lbase < 0 ? (uint32_t)(1 - lbase) : 1u;
            const uint32_t d_hi = min(nd, (uint32_t)((int32_t)ft.nrows - lbase));
            fu_load_vals<DMAX>(gfr + at, W, nd, v);
            float* o = Xs + lbase * (int32_t)xs + (st - st0) * W + c;
            float* dmp = dump + (st - st0) * W + c;
            if (G0 == 0 && st == 0) fu_scan_avg<DMAX>(v, o, xs, d_lo, d_hi, dmp);
            else if (st == 1) fu_scan_ext<DMAX, 1>(v, o, xs, d_lo, d_hi, dmp);
            else fu_scan_ext<DMAX, 0>(v, o, xs, d_lo, d_hi, dmp);
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < NRP; q++) {
      const uint32_t e = ptid + 256 * q, row = e / 48, ol = e % 48;
      if (row < ROWS) Rs[row * FE_RS + ol] = rp[q];
    }
    return ft;
  };

  __syncthreads();   // the cleared images
  uint32_t nt0 = 0, nnfr = 0, nr0 = 0;     // the tile built last (becomes the consumers' tile at the next barrier)
  if (producer) {
    FU_SETPRIO(1);
    if (first < n_tiles) { const FuTile b0 = build(0, first + G, 2); nt0 = b0.t0; nnfr = b0.nfr; nr0 = b0.r0; }
  }
  __syncthreads();
  uint32_t k = 0;
#if FU_PROF
  unsigned long long stamp_ = __builtin_amdgcn_s_memtime();
#define FW_STAMP(who, i) do { if (threadIdx.x == (who)) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); atomicAdd(&fu_prof[i], n_ - stamp_); stamp_ = n_; } } while (0)
#else
#define FW_STAMP(who, i) do {} while (0)
#endif
  for (uint64_t tile = first; tile < n_tiles; tile += G, k++) {
    const uint32_t cur = k & 1;
#if FU_PROF
    if (threadIdx.x == 0) atomicAdd(&fu_prof[14], 1ull);
#endif
    if (producer) {
      ct0 = nt0; cnfr = nnfr; cr0 = nr0;
      if (FU_ABL != 4 && tile + G < n_tiles) { const FuTile bn = build(cur ^ 1, tile + 2 * G, cur); nt0 = bn.t0; nnfr = bn.nfr; nr0 = bn.r0; }
      else dur_sums(cur);
      FW_STAMP(256, 8);    // producers: building the next tile's images
    } else {
      if (FU_ABL != 3) fe_mfma_tile<NT, 0, xs, NKS>(Rs0 + cur * RB, Xs0 + cur * XB, wave, lk, li, n_ot, acc, acc32_unused);
      FW_STAMP(0, 10);     // consumers: MFMAs
    }
    __syncthreads();
    FW_STAMP(256, 9);      // producers: waiting at the barrier
    FW_STAMP(0, 11);       // consumers: waiting at the barrier
  }
  // dslab[blockIdx.x][o][d0]; the bias count is the sum over the durations (in duration order) times the bias value
  const uint32_t nd1 = D + (lay.use_sb ? 1 : 0);
  double* dout = dslab + (uint64_t)blockIdx.x * n_out * nd1;
  double* dl = Rs0;   // [D][48], over the dead R images (the loop's last barrier is behind every wavefront)
  if (producer) {
    if (ptid < 48 * FW_DSL && blockIdx.z == 0) {
#pragma unroll
      for (int q = 0; q < NDA; q++) {
        const uint32_t d0 = dsl + FW_DSL * q;
        if (d0 < D) {
          dl[d0 * 48 + dol] = da[q];
          if (o0 + dol < n_out) dout[(uint64_t)(o0 + dol) * nd1 + d0] = da[q];
        }
      }
    }
  } else {
    // slab[blockIdx.x][o][f]
    double* out = slab + (uint64_t)blockIdx.x * n_out * ncol_all + zoff;
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const uint32_t q = wave + 4 * j;
      if (q >= n_ot) continue;
      const uint32_t n = q % 3, ct = q / 3;
      const uint32_t f = ct * 16 + li;
      if (f >= ncol) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t o = o0 + n * 16 + lk + 4 * r;
        if (o >= n_out) continue;
        out[(uint64_t)o * ncol_all + f] = acc[j][r];
      }
    }
  }
  __syncthreads();
  if (producer && blockIdx.z == 0 && lay.use_sb && ptid < 48 && o0 + ptid < n_out) {
    double bsum = 0.0;
    for (uint32_t d0 = 0; d0 < D; d0++) bsum += dl[d0 * 48 + ptid];
    dout[(uint64_t)(o0 + ptid) * nd1 + D] = bsum * lay.sbv;
  }
}

template <int NT, int DMAX, int NKS, int G0, int ROWS>
__global__ __launch_bounds__(FW_NT, 2) void k_expf_fused_ws(ScrfFusedArgs fa, ScrfLayout lay, const double* __restrict__ R, uint32_t n_out,
                                                            uint64_t n_tiles, uint32_t n_ct, double* __restrict__ slab,
                                                            double* __restrict__ dslab) {
  expf_fused_ws_body<NT, DMAX, NKS, G0, ROWS>(fa, lay, R, n_out, n_tiles, n_ct, slab, dslab);
}
// the same with 208 registers per wavefront (the attribute wants a literal, hence the second entry point): two such
// wavefronts per SIMD leave 96 registers per lane, which lets the narrow k_ztf of the side stream (64) be resident on
// the same SIMD while this kernel owns the CU
template <int NT, int DMAX, int NKS, int G0, int ROWS>
__global__ __launch_bounds__(FW_NT) __attribute__((amdgpu_num_vgpr(208))) void k_expf_fused_ws_n(
    ScrfFusedArgs fa, ScrfLayout lay, const double* __restrict__ R, uint32_t n_out, uint64_t n_tiles, uint32_t n_ct,
    double* __restrict__ slab, double* __restrict__ dslab) {
  expf_fused_ws_body<NT, DMAX, NKS, G0, ROWS>(fa, lay, R, n_out, n_tiles, n_ct, slab, dslab);
}

// column tiles needed, and the row stride of the kernel instantiation that serves them (NCT * 16)
static uint32_t fused_expf_xs(const ScrfLayout& lay, uint32_t W, uint32_t* n_ct) {
  const uint32_t ncol = 3 * W + lay.D + (lay.use_sb ? 1 : 0);
  *n_ct = (ncol + 15) / 16;
  return (*n_ct <= 5 ? 5 : *n_ct <= 9 ? 9 : 13) * 16;
}
// the wave-specialised kernel: dense groups only (g0 = 1: without the average); zb: one statistic per workgroup column
static uint32_t fused_expf_ws_xs(uint32_t W, int g0, int zb, uint32_t* n_ct) {
  const uint32_t ncol = zb ? W : (3 - g0) * W;
  *n_ct = (ncol + 15) / 16;
  return (*n_ct <= 5 ? 5 : *n_ct <= 9 ? 9 : 13) * 16;
}

// Expected-count tiles are whole frames: as many as give <= 76 rows (what two workgroups per CU can hold at config 2);
// the depth the kernel runs is the next of 52 / 64 / 76 rows (rows past the tile are staged as zeros).
// D = 25: 3 frames, 75 rows in 76; D = 10: 7 frames, 70 in 76.
uint32_t fused_expf_frames(uint32_t D) { return D >= FE_ROWS ? 1u : FE_ROWS / D; }
static uint32_t fused_expf_nks(uint32_t D) {
  const uint32_t rows = fused_expf_frames(D) * D;
  return rows <= 52 ? 13u : rows <= 64 ? 16u : 19u;
}
// most raw frames a tile can need (t_last - f0 + 1)
static uint32_t fused_expf_nfmax(uint32_t D) { return fused_expf_frames(D) + D - 1; }
static size_t fused_expf_smem(const ScrfLayout& lay, uint32_t W) {
  uint32_t n_ct;
  const uint32_t xs = fused_expf_xs(lay, W, &n_ct);
  return sizeof(float) * (FE_ROWS + 1) * xs + sizeof(double) * FE_ROWS * FE_RS +
         sizeof(float) * fused_expf_nfmax(lay.D) * W + 64;
}
// the single-role count kernel (all dense columns + one-hot + bias in one image; FAST32, SCRF_EXPF_WS=0)
static bool fused_expf_plain_fits(const ScrfLayout& lay, uint32_t W) {
  uint32_t n_ct;
  fused_expf_xs(lay, W, &n_ct);
  return n_ct <= 13 && fused_expf_smem(lay, W) <= 156 * 1024;   // above 80 KB: one workgroup per CU instead of two
}

// the wave-specialised kernel (one 512-thread workgroup per CU, two image pairs in LDS) serves the fp64 form whenever
// its LDS fits; SCRF_EXPF_WS=0 switches it off (A/B measurements).  Rows per tile: 100 when g0 = 1 and the narrower
// column image lets two pairs of that height into 160 KB (its own tile list, built with the batch), else 76.
// Wide streams (config 3: W = 144, config 5: W = 123) do not fit (3 - g0) W columns into 13 column tiles / 160 KB: the
// launch is then z-blocked, one statistic per workgroup column (W <= 208).
#define FW_ROWS_BIG 100
static size_t fused_expf_ws_smem_rows(uint32_t W, int g0, int zb, uint32_t rows) {
  uint32_t n_ct;
  const uint32_t xs = fused_expf_ws_xs(W, g0, zb, &n_ct);
  return 2 * (sizeof(float) * (rows + 1) * xs + sizeof(double) * rows * FE_RS);
}
static bool fused_expf_ws_fits(uint32_t W, int g0, int zb) {
  uint32_t n_ct;
  fused_expf_ws_xs(W, g0, zb, &n_ct);
  return n_ct <= 13 && fused_expf_ws_smem_rows(W, g0, zb, FE_ROWS) <= 160 * 1024;
}
static int fused_expf_ws_zb(uint32_t W, int g0) { return !fused_expf_ws_fits(W, g0, 0) && fused_expf_ws_fits(W, g0, 1) ? 1 : 0; }
static uint32_t fused_expf_ws_rows(uint32_t W, int g0) {
  // 100-row tiles take 232 registers per wavefront, 76-row tiles 212: only the latter leaves room on a SIMD for the
  // side stream's narrow k_ztf (64), and with that overlap the 76-row form is the faster step (29.22 against 29.46 ms;
  // without the side stream 29.54 against 29.43).  So: tall tiles only when the side stream is off.
  static const bool side = !(getenv("SCRF_SIDE") && atoi(getenv("SCRF_SIDE")) == 0);
  static const bool big = getenv("SCRF_EXPF_BIG") ? atoi(getenv("SCRF_EXPF_BIG")) != 0 : !side;
  uint32_t n_ct;
  fused_expf_ws_xs(W, g0, 0, &n_ct);
  return (big && g0 == 1 && n_ct <= 5 && !fused_expf_ws_zb(W, g0) && fused_expf_ws_smem_rows(W, g0, 0, FW_ROWS_BIG) <= 160 * 1024) ? FW_ROWS_BIG : FE_ROWS;
}
static bool fused_expf_ws(const ScrfLayout& lay, uint32_t W, int f32, int g0) {
  static const bool on = !(getenv("SCRF_EXPF_WS") && atoi(getenv("SCRF_EXPF_WS")) == 0);
  return on && !f32 && (fused_expf_ws_fits(W, g0, 0) || fused_expf_ws_fits(W, g0, 1));
}

// f32: FAST32 runs the single-role count kernel only
int fused_supported(const ScrfLayout& lay, uint32_t W, int f32) {
  if (lay.D < 2 || lay.D > 40 || W < 1) return 0;
  if (!fused_expf_plain_fits(lay, W) && !fused_expf_ws(lay, W, f32, 0)) return 0;
  // wide streams with many labels (config 5: W = 123, L = 200, D = 40) are served better by the general path: every
  // 48-output block of the fused kernels repeats the window scans, the D = 40 forms spill, k_post_z walks four label
  // groups (measured 253.6 ms fused against 206.4 ms materialised per 128 utterances).  The z-blocked count kernel is for
  // one label group.
  if (!fused_expf_plain_fits(lay, W) && lay.L > 64) return 0;
  // several label groups with the D = 40 forms (k_post_z<40> and the count kernel spill): the hybrid path is faster
  // (L = 200, D = 40, W = 40, 128 x 1000 frames: 66.8 ms fused, 49.6 ms hybrid; at D <= 25 the fused kernels win: 38.5 vs
  // 46.5 ms at L = 200, W = 39)
  if (lay.L > 64 && lay.D > 32) return 0;
  return fused_scores_tb(W, lay.D) >= 1;
}
// SCRF_PREC_FASTLIN needs the wave-specialised count kernel (the only one without the avg group); other shapes run the
// FAST kernels (the reference's float average) under that precision.  (L > 64: k_post_z keeps one Z_avg ring per
// 64-output group, which its LDS no longer holds two of per SIMD -- left to FAST.)
int fused_la_supported(const ScrfLayout& lay, uint32_t W) {
  return fused_supported(lay, W, 0) && lay.L <= 64 && fused_expf_ws(lay, W, 0, 1);
}
// layout of the count slabs: {dense columns, first dense group, separate duration slab?}, the tile list and its height
ScrfFusedExpfPlan fused_expf_plan(const ScrfLayout& lay, uint32_t W, int f32, int la) {
  ScrfFusedExpfPlan p;
  p.ws = fused_expf_ws(lay, W, f32, la ? 1 : 0) ? 1 : 0;
  p.g0 = (p.ws && la) ? 1 : 0;
  p.ncol = p.ws ? (3 - p.g0) * W : 3 * W + lay.D + (lay.use_sb ? 1 : 0);
  p.ndur = p.ws ? lay.D + (lay.use_sb ? 1 : 0) : 0;
  p.rows = p.ws ? fused_expf_ws_rows(W, p.g0) : FE_ROWS;
  p.tile_list = p.rows == FE_ROWS ? 1 : 2;
  p.frames = lay.D >= p.rows ? 1u : p.rows / lay.D;
  p.nz = (p.ws && fused_expf_ws_zb(W, p.g0)) ? (uint32_t)(3 - p.g0) : 1u;
  return p;
}
uint32_t fused_expf_blocks(const ScrfLayout& lay, uint32_t W, int f32, uint64_t n_tiles, int la) {
  static const uint32_t nb = getenv("SCRF_EXPF_BLOCKS") ? (uint32_t)atoi(getenv("SCRF_EXPF_BLOCKS")) : 512u;  // experiment knob (<= 512)
  const uint32_t cap = fused_expf_ws(lay, W, f32, la ? 1 : 0) ? std::min(nb, 256u) : nb;   // one workgroup per CU there
  return (uint32_t)(n_tiles < cap ? n_tiles : cap);
}

template <int NT, int DMAX, int NKS, int G0, int ROWS>
static void launch_expf_ws_t(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* R, uint64_t n_tiles,
                             uint32_t n_ct, double* slab, double* dslab) {
  const int zb = fused_expf_ws_zb(fa.W, G0);
  dim3 grid(fused_expf_blocks(lay, fa.W, 0, n_tiles, G0), (lay.L + 47) / 48, zb ? 3 - G0 : 1);
  const size_t smw = fused_expf_ws_smem_rows(fa.W, G0, zb, ROWS);
  if (NT <= 4) {   // the narrow forms leave room for the side stream's kernels
    hipFuncSetAttribute((const void*)k_expf_fused_ws_n<NT, DMAX, NKS, G0, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw);
    hipLaunchKernelGGL((k_expf_fused_ws_n<NT, DMAX, NKS, G0, ROWS>), grid, dim3(FW_NT), smw, st, fa, lay, R, lay.L, n_tiles, n_ct, slab, dslab);
    return;
  }
  hipFuncSetAttribute((const void*)k_expf_fused_ws<NT, DMAX, NKS, G0, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smw);
  hipLaunchKernelGGL((k_expf_fused_ws<NT, DMAX, NKS, G0, ROWS>), grid, dim3(FW_NT), smw, st, fa, lay, R, lay.L, n_tiles, n_ct, slab, dslab);
}

template <int NT, int DMAX, int F32, int NKS>
static void launch_expf_fused_t(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* R,
                                uint64_t n_tiles, uint32_t n_ct, size_t sm, double* slab) {
  dim3 grid(fused_expf_blocks(lay, fa.W, F32, n_tiles, 0), (lay.L + 47) / 48);
  hipFuncSetAttribute((const void*)k_expf_fused<NT, DMAX, F32, NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
  hipLaunchKernelGGL((k_expf_fused<NT, DMAX, F32, NKS>), grid, dim3(FE_NT), sm, st, fa, lay, R, lay.L, n_tiles, n_ct,
                     fused_expf_nfmax(lay.D), slab);
}

// slab: [fused_expf_blocks(n_tiles)][L][plan.ncol]; dslab (plan.ndur > 0): [blocks][L][D + bias]
void launch_expf_fused(hipStream_t st, const ScrfFusedArgs& fa, const ScrfLayout& lay, const double* R,
                       uint64_t n_tiles, double* slab, double* dslab, int f32, int la) {
  if (n_tiles == 0) return;
  const uint32_t nks = fused_expf_nks(lay.D);
  // the depths that occur per duration class: D <= 12 always fills more than 64 rows, 13..25 never stops at 52
#define FE_GO(N)                                                   \
  do {                                                             \
    if (lay.D <= 12) FE_GO3(N, 12, 19);                            \
    else if (lay.D <= 25) {                                        \
      if (nks == 16) FE_GO3(N, 25, 16); else FE_GO3(N, 25, 19);    \
    } else {                                                       \
      if (nks == 13) FE_GO3(N, 40, 13);                            \
      else if (nks == 16) FE_GO3(N, 40, 16);                       \
      else FE_GO3(N, 40, 19);                                      \
    }                                                              \
  } while (0)
  const ScrfFusedExpfPlan plan = fused_expf_plan(lay, fa.W, f32, la);
  if (plan.ws) {
    uint32_t n_ct;
    fused_expf_ws_xs(fa.W, plan.g0, plan.nz > 1 ? 1 : 0, &n_ct);
    if (plan.rows == FW_ROWS_BIG) {
      // 100-row tiles (g0 = 1, n_ct <= 5): k-steps = ceil(frames * D / 4) rounded up to 21 or 25
      const uint32_t used = plan.frames * lay.D;
#define FB_GO(DM)                                                                                                 \
  do {                                                                                                            \
    if (used <= 84) launch_expf_ws_t<4, DM, 21, 1, FW_ROWS_BIG>(st, fa, lay, R, n_tiles, n_ct, slab, dslab);      \
    else launch_expf_ws_t<4, DM, 25, 1, FW_ROWS_BIG>(st, fa, lay, R, n_tiles, n_ct, slab, dslab);                 \
  } while (0)
      if (lay.D <= 12) FB_GO(12);
      else if (lay.D <= 25) FB_GO(25);
      else FB_GO(40);
#undef FB_GO
      return;
    }
#define FE_GO3(N, DM, KS)                                                                                   \
  do {                                                                                                      \
    if (plan.g0) launch_expf_ws_t<N, DM, KS, 1, FE_ROWS>(st, fa, lay, R, n_tiles, n_ct, slab, dslab);       \
    else launch_expf_ws_t<N, DM, KS, 0, FE_ROWS>(st, fa, lay, R, n_tiles, n_ct, slab, dslab);               \
  } while (0)
    // slots per consumer wavefront = ceil(3 n_ct / 4)
    if (n_ct <= 5) FE_GO(4);
    else if (n_ct <= 8) FE_GO(6);
    else if (n_ct <= 9) FE_GO(7);
    else FE_GO(10);
#undef FE_GO3
    return;
  }
  uint32_t n_ct;
  fused_expf_xs(lay, fa.W, &n_ct);
  const size_t sm = fused_expf_smem(lay, fa.W);
#define FE_GO3(N, DM, KS)                                                                             \
  do {                                                                                                \
    if (f32) launch_expf_fused_t<N, DM, 1, KS>(st, fa, lay, R, n_tiles, n_ct, sm, slab);              \
    else launch_expf_fused_t<N, DM, 0, KS>(st, fa, lay, R, n_tiles, n_ct, sm, slab);                  \
  } while (0)
  if (n_ct <= 5) FE_GO(4);
  else if (n_ct <= 9) FE_GO(7);
  else FE_GO(10);
#undef FE_GO
#undef FE_GO3
}
