// scrf_common.h -- shared host/device definitions of the MI355X segmental-CRF engine.
// gfx950 only: 64-wide wavefronts are assumed throughout.
#ifndef SCRF_COMMON_H_
#define SCRF_COMMON_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scrf_abi.h"

#define SCRF_WAVE 64

// Lambda layout and feature ranges, closed form of CRF_StdFeatureMap::recalc /
// computeStateFeatureIdx / computeTransFeatureIdx for numStates == 1
// (ftrmaps/CRF_StdFeatureMap.cpp:280-320,355-410,472-517): per current label c a block
// [nsf state weights (features ascending, bias last)][for p: ntf weights of (p->c)].
struct ScrfLayout {
  uint32_t L, D, F;
  int32_t use_sf, use_tf, use_sb, use_tb;
  uint32_t sfs, sfe, tfs, tfe;
  double sbv, tbv;
  uint32_t nsf, ntf;   // numStateFuncs, numTransFuncs (bias included)
  uint32_t nsfe, ntfe; // feature counts excluding bias
  uint32_t stride;     // nsf + L*ntf
  uint32_t lambda_len;
  uint32_t K;          // states per label (numStates); 1 except for the n-state frame model (scrf_nstate.hip)

  // one state per label: the closed form every kernel of the dense models uses
  __host__ __device__ inline uint32_t state_idx(uint32_t c) const { return c * stride; }
  __host__ __device__ inline uint32_t trans_idx(uint32_t p, uint32_t c) const { return c * stride + nsf + p * ntf; }
  // any K (scrf_nstate.hip and the layout hooks).  K > 1 (ftrmaps/CRF_StdFeatureMap.cpp:293-312, :367-407): a label's
  // block is its state functions, its self transition, then for a phone's start state the transitions from every
  // phone's end state, for the other states the one from the state before; transitions outside that topology have no
  // weights (0xffffffff)
  __host__ __device__ inline uint32_t state_idx_k(uint32_t c) const {
    if (K <= 1) return c * stride;
    const uint32_t ns = (c + K - 1) / K;   // start states among the labels before c
    return c * nsf + ntf * (ns * (L / K + 1) + (c - ns) * 2);
  }
  __host__ __device__ inline uint32_t trans_idx_k(uint32_t p, uint32_t c) const {
    if (K <= 1) return c * stride + nsf + p * ntf;
    uint32_t v = state_idx_k(c) + nsf;
    if (p == c) return v;
    v += ntf;
    if (c % K == 0) return ((p + 1) % K == 0) ? v + (p / K) * ntf : 0xffffffffu;
    return p == c - 1 ? v : 0xffffffffu;
  }
};

// Which slice of the window vector / weight block a contraction covers.
//   kind 0: outputs = labels,            woff(o) = state_idx(o) + wadd
//   kind 1: outputs = (p,c) pairs,       woff(o) = trans_idx(o / L, o % L) + wadd
//   kind 2: outputs = (group k, label),  woff(o) = state_idx(o % L) + (o / L) * rw + wadd   (per-frame projections of the sampled blocks)
struct ScrfGemmSpec {
  uint32_t kind;
  uint32_t fs;        // first X column used
  uint32_t nfe;       // number of X columns
  uint32_t use_bias;  // append a bias column of value `bias`
  double bias;
  uint32_t wadd;
  uint32_t rw;
  __host__ __device__ inline uint32_t nfun() const { return nfe + (use_bias ? 1 : 0); }
  __host__ __device__ inline uint32_t woff(const ScrfLayout& l, uint32_t o) const {
    if (kind == 0) return l.state_idx(o) + wadd;
    if (kind == 1) return l.trans_idx(o / l.L, o % l.L) + wadd;
    return l.state_idx(o % l.L) + (o / l.L) * rw + wadd;
  }
};
inline ScrfGemmSpec scrf_spec_state(const ScrfLayout& l) { return ScrfGemmSpec{0, l.sfs, l.nsfe, (uint32_t)l.use_sb, l.sbv, 0, 0}; }
inline ScrfGemmSpec scrf_spec_trans(const ScrfLayout& l) { return ScrfGemmSpec{1, l.tfs, l.ntfe, (uint32_t)l.use_tb, l.tbv, 0, 0}; }

// windows ending at frame t: min(t+1, D) (gradbuilder :243-251)
__host__ __device__ inline uint32_t scrf_node_max_dur(uint32_t t, uint32_t D) {
  return (t + 1 <= D) ? t + 1 : D;
}
// number of previous nodes linked to node t (gradbuilder :258-266)
__host__ __device__ inline uint32_t scrf_num_prev(uint32_t t, uint32_t D) {
  return (t + 1 <= D) ? t : D;
}
// row of window d=1 of frame t inside its utterance; scrf_seg_base(T) = N_seg
__host__ __device__ inline uint64_t scrf_seg_base(uint32_t t, uint32_t D) {
  return (t < D) ? (uint64_t)t * (t + 1) / 2
                 : (uint64_t)D * (D + 1) / 2 + (uint64_t)(t - D) * D;
}
// lattice bookkeeping (decoders/...WithoutSegTransFtr.h:248-330): first state id of node t
__host__ __device__ inline int32_t scrf_node_start_state(uint32_t t, uint32_t L) {
  return (t == 0) ? 1 : (int32_t)(1 + L + (t - 1) * 2 * L);
}
// index of the first arc emitted while visiting node t
__host__ __device__ inline uint64_t scrf_arc_base(uint32_t t, uint32_t L, uint32_t D) {
  return (t == 0) ? 0 : (uint64_t)L + (uint64_t)(t - 1) * L * L + (uint64_t)L * (scrf_seg_base(t, D) - 1);
}

// Posterior-mass self-checks of the reference's computeExpF, per node t:
//   segmental (nodes/CRF_StdSegStateNode_WithoutDurLab_WithoutSegTransFtr.cpp:917-947): state mass
//   sum_{l,d} gamma <= 1 + 1e-6 and >= -1e-6, the same for the transition mass, and the two equal
//   within 1e-6 (the last node compares with 1.0);
//   frame model (nodes/CRF_StdStateNode.cpp:252-275): both within [0.9, 1.1].
// The transition mass sum_{c,n} xi[t][c][n] equals sum_c exp(alpha[t][c] + beta[t][c] - Zx) (beta is
// the log-sum over n of M + sd), which is how the kernels form it -- from arrays they already hold.
__host__ __device__ inline bool scrf_mass_ok(double state_mass, double trans_mass, bool last, bool frame_model) {
  if (last) trans_mass = 1.0;   // :902-907 (no following transition)
  if (frame_model) return state_mass <= 1.1 && state_mass >= 0.9 && trans_mass <= 1.1 && trans_mass >= 0.9;
  if (!(state_mass <= 1.000001) || !(state_mass >= -0.000001)) return false;
  if (!(trans_mass <= 1.000001) || !(trans_mass >= -0.000001)) return false;
  const double d = state_mass - trans_mass;
  return d <= 0.000001 && d >= -0.000001;
}

// Device view of a packed batch (all arrays in HBM).
struct ScrfBatchView {
  uint32_t U;
  const uint32_t* T;          // [U]
  const uint64_t* frame_off;  // [U+1]
  const uint64_t* seg_off;    // [U+1]
  const uint64_t* arc_off;    // [U+1]
  const uint32_t* labels;     // [sum T] or nullptr
};

// Outputs of the scaled linear-domain recursion (scrf_dplin.hip): mantissa vectors [frames][L] and
// per-frame log-scales [frames].
struct ScrfDpLin {
  double* a;    double* ga;    // exp(alpha[t][l])  = a * exp(ga)
  double* p;    double* gp;    // exp(aPT[t][l])    = p * exp(gp)   (alpha plus transition)
  double* b;    double* gb;    // exp(beta[t][l])   = b * exp(gb)
  double* sd;   double* gsd;   // exp(sd[t][l])     = sd * exp(gsd) (sum over next durations)
};

// Fused window synthesis (scrf_fused.hip).  Row tiles are described once per batch on the host:
// score tiles are the windows of TB = 256/D whole frames, expected-count tiles any 64 consecutive
// windows of an utterance.
struct ScrfTileDesc {
  uint64_t row_abs;   // batch row of the tile's first window (seg_off[u] + r0)
  uint64_t fr_abs;    // batch index of the first raw frame staged (frame_off[u] + t0 - back)
  uint32_t r0;        // first row inside the utterance
  uint32_t t0;        // first frame whose windows the tile touches
  uint16_t back;      // min(t0, D-1): frames staged before t0
  uint16_t nfr;       // frames touched
  uint16_t nrows;
  uint16_t pad_;
};
struct ScrfFusedArgs {
  const float* frames;        // stream-0 raw frames of the batch, [sum T][W]
  const ScrfTileDesc* tiles;  // all tiles of the batch
  uint64_t tile0;             // first tile of the chunk
  uint64_t row_base;          // seg_off[u0], frame_off[u0] of the chunk
  uint64_t frame_base;
  uint32_t TB, W;
  const double* dtab;         // k_dur_table's output ([output block][D][50]), or nullptr: the score kernel builds it per tile
  const void* rtab;           // k_tile_tables' output (records, row bases, rowmap of a steady-state tile), or nullptr
};

// Decode mode of the fused score kernel: instead of the fp64 scores it writes the float arc
// weights float(-1 * score) the lattice / Viterbi consume, and lists every entry whose float
// rounding the fp64-MFMA evaluation cannot guarantee to equal the reference-order evaluation's:
// |fused - reference order| <= (gamma_n + gamma_m) * sum_f |x_f lambda_f| <= bound_scale * xm * w1[o]
// =: B (constants in scrf_engine.cpp, run_scores), so an entry is safe when float(v - B) == float(v + B).
// The listed entries (a few per 10^5) are recomputed
// in reference order by k_decode_fixup; everything else is bit-identical by the bound.
struct ScrfDecodeOut {
  float* wneg;          // [rows][n_out]; nullptr = mode off
  const double* w1;     // [n_out] sum_f |lambda_f| over the label's state block, bias weight included
  const float* xm_f;    // [frames of the chunk] max(|x|, 1, |bias value|) over the frame's utterance
  uint32_t* cnt;        // entries appended to `list` (may exceed cap = overflow)
  uint64_t* list;       // (chunk-relative row << 16) | output
  uint32_t cap;
  double bound_scale;   // gamma_n + gamma_m, see above
};

// XCD-aware workgroup order (cdna_hip_programming.md T1): workgroup ids go round-robin over the 8 XCDs, each with its own
// L2, so neighbours in launch order do not share a cache.  swz gives every XCD a contiguous run of the logical order
// (bijective for any workgroup count); the kernels below order their tiles so that a run holds the tiles that read the
// same rows of X.
__host__ __device__ inline uint32_t xcd_swizzle(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg / 8, r = nwg % 8, xcd = bid % 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}


#endif  // SCRF_COMMON_H_
