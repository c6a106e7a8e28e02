// scrf_engine.cpp -- host side of libscrf_amd.so: the C ABI of include/scrf_abi.h on top of
// the HIP kernels.  One engine = one GPU, one HIP stream; batches live in HBM; scratch is a
// single device arena carved per chunk of utterances.  There is NO CPU compute path here:
// without a HIP device scrf_create fails with SCRF_ERR_NO_DEVICE.
#include <dlfcn.h>
#include <time.h>
#include <unistd.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <unordered_map>
#include <vector>

#include "scrf_kernels.h"

// ---------------------------------------------------------------------------------------------
// RCCL, bound lazily (so that loading the library never drags a second RCCL into a process
// that already has one, e.g. under torch.distributed)
// ---------------------------------------------------------------------------------------------
typedef struct { char internal[128]; } scrf_nccl_uid;
typedef void* scrf_nccl_comm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(scrf_nccl_uid*) = nullptr;
  int (*CommInitRank)(scrf_nccl_comm*, int, scrf_nccl_uid, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, scrf_nccl_comm, hipStream_t) = nullptr;
  int (*CommDestroy)(scrf_nccl_comm) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommAbort)(scrf_nccl_comm) = nullptr;
  int (*CommGetAsyncError)(scrf_nccl_comm, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static bool rccl_load(std::string* why) {
  if (g_rccl.lib) return true;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_rccl.lib) break;
  }
  if (!g_rccl.lib) {
    *why = std::string("cannot load RCCL: ") + dlerror();
    return false;
  }
  g_rccl.GetUniqueId = (int (*)(scrf_nccl_uid*))dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(scrf_nccl_comm*, int, scrf_nccl_uid, int))dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, scrf_nccl_comm, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(scrf_nccl_comm))dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.GroupStart = (int (*)())dlsym(g_rccl.lib, "ncclGroupStart");
  g_rccl.GroupEnd = (int (*)())dlsym(g_rccl.lib, "ncclGroupEnd");
  g_rccl.CommAbort = (int (*)(scrf_nccl_comm))dlsym(g_rccl.lib, "ncclCommAbort");
  g_rccl.CommGetAsyncError = (int (*)(scrf_nccl_comm, int*))dlsym(g_rccl.lib, "ncclCommGetAsyncError");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) {
    *why = "RCCL symbols missing";
    return false;
  }
  return true;
}
enum { SCRF_NCCL_DOUBLE = 8, SCRF_NCCL_SUM = 0 };  // ncclFloat64, ncclSum (rccl.h)

// ---------------------------------------------------------------------------------------------
struct scrf_engine_s {
  scrf_config cfg;
  ScrfLayout lay;
  // K states per label on a segmental model (nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp): the
  // kernels run the dense one-state layout `lay` with the bias of every transition the topology lacks pinned at
  // log 0 (mask_w); the callers see the reference's compact layout `xlay` (CRF_StdFeatureMap.cpp:293-312), and every
  // weight-length vector crossing the ABI is scattered / gathered through s2d (compact index -> dense index)
  bool shadow = false;
  ScrfLayout xlay;
  std::vector<uint32_t> s2d, masked;
  double mask_w = 0.0;
  std::vector<double> xbuf;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  double* d_lambda = nullptr;
  double* d_lambda_acc = nullptr;
  double* d_gsa = nullptr;
  double* d_grad = nullptr;
  bool own_grad = true;
  double* d_m0 = nullptr;     // [L*L] time-invariant transition scores (bias-only transitions)
  double* d_e0 = nullptr;     // exp(M0 - shift), its transpose and the shift (linear-domain DP)
  double* d_et0 = nullptr;
  double* d_msh0 = nullptr;
  bool m0_valid = false;
  double* d_sums = nullptr;   // {numer, zx, n_utts, active}
  // a batch's gradient and sums are built in a staging buffer and committed to d_grad / d_sums by a kernel
  // that is a no-op when an utterance of the batch failed (d_latch): a failed scrf_fb_batch leaves the
  // gradient as it was, and a NUMERIC failure of the linear-domain recursion can be redone in the log domain
  double* d_stage = nullptr;
  double* d_sums_stage = nullptr;
  int* d_latch = nullptr;     // {status code, utterance} of the first failed utterance of the batch in flight
  int* h_latch = nullptr;     // pinned copy, valid once ev_status has completed
  double* h_sums = nullptr;   // pinned image of the batch sums (scrf_queue_batch_sums / scrf_take_batch_sums)
  hipEvent_t ev_sums = nullptr;
  bool sums_queued = false;
  hipEvent_t ev_status = nullptr;
  uint64_t n_lin_fallback = 0;   // batches redone through the log-domain recursion
  char* scratch = nullptr;
  size_t scratch_cap = 0;
  // second lane: alternate chunks of a batch run on a second stream so that the VALU-bound DP
  // kernels of one chunk overlap the MFMA-bound contractions of the other
  hipStream_t stream2 = nullptr;
  // Batch arrays (scrf_batch_create / scrf_batch_destroy) come from a pool and are uploaded on a stream of their own:
  // a trainer that makes one batch per minibatch used to synchronise the device at every destroy (hipFree) and to wait
  // for the previous minibatch's kernels at every upload.  A freed block carries the epoch of its destroy call; the
  // engine stream's event of that epoch says when the last kernel that could read it has finished, and only then is the
  // block handed out again (otherwise a fresh one is allocated).  SCRF_BATCH_POOL=0: hipMalloc / hipFree as before.
  struct PoolBlock { void* p; size_t cap; uint64_t epoch; };
  std::vector<PoolBlock> pool_free;
  std::unordered_map<void*, size_t> pool_cap;     // live blocks
  hipStream_t up_stream = nullptr;
  hipEvent_t pool_ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  uint64_t pool_epoch = 0, pool_done = 0;         // destroy calls so far / epochs known to be finished
  size_t pool_bytes = 0;                          // bytes sitting in pool_free
  bool pool_on = true;
  bool pool_up = true;     // uploads on their own stream (SCRF_BATCH_POOL=2: on the engine stream)
  char* scratch2 = nullptr;
  size_t scratch2_cap = 0;
  double* d_grad2 = nullptr;
  double* d_sums2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int n_lanes = 1;  // SCRF_LANES=2: alternate chunks on two streams (worth ~3 % at config 2; off by default)
  bool fuse_windows = true;
  // all-reduce / compute overlap (scrf_fb_batch_allreduce): inside the fused call the transition
  // contraction runs first and the all-reduce of the transition weights (98.7 % of the TIMIT-demo gradient) is issued on
  // the second stream as soon as they are committed, under the state contraction
  bool overlap_comm = false;  // set for the duration of a scrf_fb_batch_allreduce call
  bool early_done = false;    // the transition block of this step has been all-reduced already
  uint64_t n_collectives = 0, n_overlapped = 0;   // scrf_comm_stats
  bool comm_overlap_on = true;   // SCRF_COMM_OVERLAP=0: the fused call issues both blocks after the batch (same results)
  double* d_pack = nullptr;   // [L * nsf + 8]: the state weights of every label + the 8 scalars, one small message
  bool fuse_mixed = true;     // SCRF_FUSE_MIXED=0: two-stream batches keep the general path for the state part too
  bool dur_table = true;      // the score kernel copies its duration-weight table instead of building it per tile (SCRF_DTAB=0: off)
  bool side_stream = true;    // k_ztf + transition counts on the second stream under k_expf_fused_ws (SCRF_SIDE=0: off)
  void* d_rtab = nullptr;       // k_tile_tables: row records / row bases / rowmap of a steady-state score tile
  size_t rtab_bytes = 0;
  double* d_dtab = nullptr;     // k_dur_table: duration weight + bias per output block, for the fused score kernel
  double* d_sl_tab = nullptr;   // STDSEG, bias-only transitions: E, E^T (nLabs^2 each) and max M (scrf_stdseg_lin.hip)
  bool frame_mass = false;   // posterior-mass self-checks with the frame model's bounds (scrf_set_frame_mass_check)
  bool lin_dp = true;
  // the workgroup-per-utterance log-domain recursion (k_fb: column-wise max-shifted log-sum-exp, the
  // reference's LogMath) instead of the wavefront kernels, whose transition step works on exp(M - max M):
  // set for the automatic redo of a batch / hook call that raised SCRF_ERR_NUMERIC there
  bool force_fb = false;
  // decode from the fused score kernel's float arc weights + reference-order fix-ups (bit-identical
  // to the EXACT path by a rounding-error bound, ScrfDecodeOut); SCRF_FAST_DECODE=0 turns it off
  bool hybrid = true;   // SCRF_HYBRID=0: the general path contracts all 8 W + D columns of X
  bool hybrid_first = false;   // SCRF_HYBRID=2 (A/B runs): L > 64 takes the hybrid path even where the fused kernels fit
  bool fast_decode = true;
  double decode_bound_factor = 1.0;   // SCRF_DECODE_BOUND_SCALE (tests widen the screen with it)
  double* d_w1 = nullptr;
  uint64_t n_decode_fix = 0, n_decode_fallback = 0;   // entries recomputed / chunks sent back to the EXACT path
  // result buffers of scrf_viterbi_batch, kept across calls (a hipMalloc / hipFree pair per call is a device-wide
  // synchronisation each) and their pinned host images (the labels come back in one asynchronous copy)
  uint32_t *dec_lab = nullptr, *dec_n = nullptr, *dec_hlab = nullptr, *dec_hn = nullptr;
  float *dec_cost = nullptr, *dec_hcost = nullptr;
  uint64_t dec_cap_f = 0, dec_cap_u = 0;
  std::string err;
  // per-kernel HIP-event times of the last timed call (scrf_kernel_timing)
  struct KTime { std::string name; double ms; uint32_t n; };
  std::vector<KTime> ktimes;
  hipEvent_t kev[2] = {nullptr, nullptr};
  bool timing = false;
  hipEvent_t ev[SCRF_N_PHASES + 1][2];
  bool ev_ok = false;
  float ms[SCRF_N_PHASES];
  uint32_t nlaunch[SCRF_N_PHASES];
  scrf_nccl_comm comm = nullptr;
  int rank = 0, nranks = 1;
};

struct scrf_batch_s {
  uint32_t U = 0;
  int mode = 0;  // 0 = windows resident, 1 = frames + recipe
  std::vector<uint32_t> T;
  std::vector<uint64_t> frame_off, seg_off, arc_off;
  uint32_t* d_T = nullptr;
  uint64_t* d_frame_off = nullptr;
  uint64_t* d_seg_off = nullptr;
  uint64_t* d_arc_off = nullptr;
  uint32_t* d_labels = nullptr;
  uint32_t* d_next_lab = nullptr;      // [sum T] label of the next labelled frame (gradbuilder :436-444)
  uint32_t* d_prev_lab = nullptr;      // [sum T] label of the nearest EARLIER labelled frame (CRF_NewGradBuilder_StdSeg.cpp, STDSEG_NO_DUR)
  uint32_t* d_trans_counts = nullptr;  // [L*L] observed (c -> n) transitions of the whole batch
  uint32_t* d_frame_u = nullptr;       // [sum T] utterance of each frame
  float* d_xm_f = nullptr;             // [sum T] max(|x|, 1, |state bias value|) over the frame's utterance (fused path)
  float* d_windows = nullptr;
  uint32_t n_streams = 0;
  scrf_stream_recipe recipe[SCRF_MAX_STREAMS];
  uint32_t width[SCRF_MAX_STREAMS];
  float* d_frames[SCRF_MAX_STREAMS] = {nullptr, nullptr, nullptr};
  uint64_t* d_sframe_off[SCRF_MAX_STREAMS] = {nullptr, nullptr, nullptr};
  double* d_numer = nullptr;
  double* d_zx = nullptr;
  int* d_status = nullptr;
  // fused window synthesis: row tiles of the score kernel [0] and of the expected-count kernel [1]
  bool fused_ok = false;
  bool hybrid_ok = false;   // one segment-recipe stream the fused kernels do not take (L > 64): sampled blocks through P / Z, dense statistics materialised
  bool mixed = false;   // fused state part (stream 0) + materialised transition-feature streams (config 3's shape)
  std::vector<uint64_t> tile_off[3];   // 0: score tiles, 1: expected-count tiles (<= 76 rows), 2: <= 100 rows (FASTLIN)
  ScrfTileDesc* d_tiles[3] = {nullptr, nullptr, nullptr};
  ScrfBatchView view() const {
    ScrfBatchView v;
    v.U = U; v.T = d_T; v.frame_off = d_frame_off; v.seg_off = d_seg_off; v.arc_off = d_arc_off;
    v.labels = d_labels;
    return v;
  }
};

static std::string g_create_err;

static int fail(scrf_handle h, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) h->err = buf; else g_create_err = buf;
  return code;
}

#define HIPCHK(h, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail(h, SCRF_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

static uint32_t window_width(const scrf_stream_recipe& r, uint32_t D) {
  // io/CRF_InFtrStream_SeqMultiWindow.cpp:50-125
  if (D == 1) return (r.left_ctx + 1 + r.right_ctx) * r.in_width;
  if (r.extract_seg_ftr) return 8 * r.in_width + D + (r.left_ctx + r.right_ctx) * r.in_width;
  return (r.left_ctx + 1 + r.right_ctx) * r.in_width;
}

static int build_layout(const scrf_config& c, ScrfLayout* l, std::string* why) {
  if (c.abi_version != SCRF_ABI_VERSION) { *why = "abi_version mismatch"; return SCRF_ERR_INVALID; }
  if (c.num_states == 0) { *why = "num_states must be >= 1"; return SCRF_ERR_INVALID; }
  if (c.num_states > 1 && c.model_type != SCRF_STDFRAME && c.model_type != SCRF_STDSEG_NO_DUR_NO_TRANSFTR && c.model_type != SCRF_STDSEG_NO_DUR_NO_SEGTRANSFTR) {
    // nodes/CRF_StateNode.cpp:496-507: "CRF_StdSegNStateNode has not been implemented yet"
    *why = "crf_states > 1: CRF_StdSegNStateNode / CRF_StdSegNStateNode_WithoutDurLab have not been implemented (the reference has stdframe and stdseg_no_dur_no_segtransftr n-state nodes only)";
    return SCRF_ERR_INVALID;
  }
  if (c.num_states > 1 && c.model_type != SCRF_STDFRAME && (!c.use_trans_bias || !(c.trans_bias_val > 0.0f))) {
    *why = "crf_states > 1 on a segmental model needs the transition bias (crf_use_trans_bias, a positive bias value): the topology is held by it";
    return SCRF_ERR_INVALID;
  }
  if (c.num_states > 1 && c.num_labs % c.num_states != 0) { *why = "Invalid state/label combination while computing transitions"; return SCRF_ERR_INVALID; }  // CRF_StdFeatureMap.cpp:476-479
  if (c.num_labs == 0 || c.lab_max_dur == 0 || c.num_feas == 0) { *why = "num_labs, lab_max_dur, num_feas must be > 0"; return SCRF_ERR_INVALID; }
  if (c.model_type == SCRF_STDSEG && c.num_labs % c.lab_max_dur != 0) { *why = "stdseg: the number of all labels and the maximum duration of labels do not correspond (nLabs = nActualLabs * labMaxDur)"; return SCRF_ERR_INVALID; }
  if (c.model_type > SCRF_STDSEG_NO_DUR_NO_SEGTRANSFTR) { *why = "unknown model_type"; return SCRF_ERR_INVALID; }
  if (c.model_type == SCRF_STDFRAME && c.lab_max_dur != 1) { *why = "the maximum duration of labels must be 1 for \"stdframe\" CRF model."; return SCRF_ERR_INVALID; }  // CRFTrain/src/Main.cpp:574-578
  if (c.map_type > SCRF_STDTRANS) { *why = "only dense stdstate/stdtrans feature maps are built"; return SCRF_ERR_INVALID; }
  if (c.model_type == SCRF_STDSEG_NO_DUR_NO_TRANSFTR && (c.map_type != SCRF_STDSTATE || c.use_trans_ftrs)) {   // CRFTrain/src/Main.cpp:465-468
    *why = "crf_featuremap must be \"stdstate\" for \"stdseg_no_dur_no_transftr\" CRF model.";
    return SCRF_ERR_INVALID;
  }
  if (c.num_labs > 1024) { *why = "num_labs > 1024 unsupported"; return SCRF_ERR_INVALID; }
  memset(l, 0, sizeof(*l));
  l->L = c.num_labs; l->D = c.lab_max_dur; l->F = c.num_feas;
  l->use_sf = c.use_state_ftrs != 0; l->use_tf = c.use_trans_ftrs != 0;
  l->use_sb = c.use_state_bias != 0; l->use_tb = c.use_trans_bias != 0;
  l->sfs = c.state_fidx_start; l->sfe = c.state_fidx_end; l->tfs = c.trans_fidx_start; l->tfe = c.trans_fidx_end;
  l->sbv = c.state_bias_val; l->tbv = c.trans_bias_val;
  if (l->use_sf && (l->sfe < l->sfs || l->sfe >= l->F)) { *why = "state feature range outside the window vector"; return SCRF_ERR_INVALID; }
  if (l->use_tf && (l->tfe < l->tfs || l->tfe >= l->F)) { *why = "transition feature range outside the window vector"; return SCRF_ERR_INVALID; }
  l->nsfe = l->use_sf ? l->sfe - l->sfs + 1 : 0;
  l->ntfe = l->use_tf ? l->tfe - l->tfs + 1 : 0;
  l->nsf = l->nsfe + (l->use_sb ? 1 : 0);
  l->ntf = l->ntfe + (l->use_tb ? 1 : 0);
  l->stride = l->nsf + l->L * l->ntf;
  l->K = c.num_states;
  uint64_t ll = (uint64_t)l->L * l->stride;
  if (l->K > 1) {   // :480-485: end->start + self + next-state transitions
    const uint64_t P = l->L / l->K;
    ll = (uint64_t)l->L * l->nsf + (P * P + 2 * (uint64_t)l->L - P) * l->ntf;
  }
  if (ll == 0 || ll > 0xfffffff0ull) { *why = "lambda_len out of range"; return SCRF_ERR_INVALID; }
  l->lambda_len = (uint32_t)ll;
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------------------
extern "C" int scrf_create(const scrf_config* cfg, scrf_handle* out) {
  if (!cfg || !out) return fail(nullptr, SCRF_ERR_INVALID, "scrf_create: null argument");
  *out = nullptr;
  ScrfLayout lay;
  std::string why;
  int rc = build_layout(*cfg, &lay, &why);
  if (rc != SCRF_OK) return fail(nullptr, rc, "scrf_create: %s", why.c_str());
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, SCRF_ERR_NO_DEVICE, "scrf_create: no HIP device (%s); this library has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (cfg->device_id < 0 || cfg->device_id >= ndev)
    return fail(nullptr, SCRF_ERR_INVALID, "scrf_create: device_id %d out of range [0,%d)", cfg->device_id, ndev);
  scrf_handle h = new scrf_engine_s();
  h->cfg = *cfg;
  h->lay = lay;
  h->xlay = lay;
  if (lay.K > 1 && cfg->model_type != SCRF_STDFRAME) {
    const uint64_t dense = (uint64_t)lay.L * lay.stride;
    if (dense > 0xfffffff0ull) { delete h; return fail(nullptr, SCRF_ERR_INVALID, "scrf_create: lambda_len out of range"); }
    h->shadow = true;
    h->lay.K = 1;
    h->lay.lambda_len = (uint32_t)dense;
    lay = h->lay;
    h->mask_w = -1e30 / (double)cfg->trans_bias_val;
    h->s2d.assign(h->xlay.lambda_len, 0);
    for (uint32_t c = 0; c < lay.L; c++) {
      for (uint32_t f = 0; f < lay.nsf; f++) h->s2d[h->xlay.state_idx_k(c) + f] = lay.state_idx(c) + f;
      for (uint32_t p = 0; p < lay.L; p++) {
        const uint32_t x = h->xlay.trans_idx_k(p, c);
        if (x == 0xffffffffu) h->masked.push_back(lay.trans_idx(p, c) + lay.ntf - 1);   // the bias is the last transition function
        else for (uint32_t f = 0; f < lay.ntf; f++) h->s2d[x + f] = lay.trans_idx(p, c) + f;
      }
    }
  }
  h->device = cfg->device_id;
  if (h->cfg.scratch_bytes == 0) h->cfg.scratch_bytes = 8ull << 30;
  if (const char* e = getenv("SCRF_LANES")) h->n_lanes = atoi(e) > 1 ? 2 : 1;  // experiment knobs
  if (const char* e = getenv("SCRF_FUSE")) h->fuse_windows = atoi(e) != 0;
  if (const char* e = getenv("SCRF_SIDE")) h->side_stream = atoi(e) != 0;
  if (const char* e = getenv("SCRF_DTAB")) h->dur_table = atoi(e) != 0;
  if (const char* e = getenv("SCRF_FUSE_MIXED")) h->fuse_mixed = atoi(e) != 0;
  if (const char* e = getenv("SCRF_COMM_OVERLAP")) h->comm_overlap_on = atoi(e) != 0;
  if (const char* e = getenv("SCRF_LINDP")) h->lin_dp = atoi(e) != 0;
  if (const char* e = getenv("SCRF_FAST_DECODE")) h->fast_decode = atoi(e) != 0;
  if (const char* e = getenv("SCRF_BATCH_POOL")) { h->pool_on = atoi(e) != 0; h->pool_up = atoi(e) != 2; }
  if (const char* e = getenv("SCRF_HYBRID")) { h->hybrid = atoi(e) != 0; h->hybrid_first = atoi(e) == 2; }
  if (const char* e = getenv("SCRF_DECODE_BOUND_SCALE")) h->decode_bound_factor = std::max(1.0, atof(e));   // widening only: < 1 would void the bound
  memset(h->ms, 0, sizeof(h->ms));
  memset(h->nlaunch, 0, sizeof(h->nlaunch));
#define CRCHK(call)                                                                          \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      fail(nullptr, SCRF_ERR_HIP, "scrf_create: %s failed: %s", #call, hipGetErrorString(e_)); \
      delete h;                                                                              \
      return SCRF_ERR_HIP;                                                                   \
    }                                                                                        \
  } while (0)
  CRCHK(hipSetDevice(h->device));
  CRCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  h->own_stream = true;
  size_t nb = sizeof(double) * lay.lambda_len;
  CRCHK(hipMalloc((void**)&h->d_lambda, nb));
  CRCHK(hipMalloc((void**)&h->d_lambda_acc, nb));
  CRCHK(hipMalloc((void**)&h->d_gsa, nb));
  CRCHK(hipMalloc((void**)&h->d_grad, nb));
  CRCHK(hipMalloc((void**)&h->d_m0, sizeof(double) * lay.L * lay.L));
  CRCHK(hipMalloc((void**)&h->d_w1, sizeof(double) * lay.L));
  if (lay.D <= 40) CRCHK(hipMalloc((void**)&h->d_dtab, sizeof(double) * fused_dur_table_doubles(lay)));
  if (cfg->model_type == SCRF_STDSEG && !lay.use_tf) CRCHK(hipMalloc((void**)&h->d_sl_tab, sizeof(double) * (2 * (size_t)lay.L * lay.L + 8)));
  CRCHK(hipMalloc((void**)&h->d_e0, sizeof(double) * lay.L * lay.L));
  CRCHK(hipMalloc((void**)&h->d_et0, sizeof(double) * lay.L * lay.L));
  CRCHK(hipMalloc((void**)&h->d_msh0, sizeof(double)));
  CRCHK(hipMalloc((void**)&h->d_sums, sizeof(double) * 8));
  CRCHK(hipMalloc((void**)&h->d_stage, nb));
  CRCHK(hipMalloc((void**)&h->d_sums_stage, sizeof(double) * 4));
  CRCHK(hipMalloc((void**)&h->d_latch, sizeof(int) * 2));
  CRCHK(hipHostMalloc((void**)&h->h_latch, sizeof(int) * 2, hipHostMallocDefault));
  h->h_latch[0] = h->h_latch[1] = 0;
  CRCHK(hipHostMalloc((void**)&h->h_sums, sizeof(double) * 4, hipHostMallocDefault));
  CRCHK(hipEventCreateWithFlags(&h->ev_sums, hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&h->ev_status, hipEventDisableTiming));
  CRCHK(hipEventCreate(&h->kev[0]));
  CRCHK(hipEventCreate(&h->kev[1]));
  CRCHK(hipMalloc((void**)&h->d_grad2, nb));
  CRCHK(hipMalloc((void**)&h->d_sums2, sizeof(double) * 4));
  CRCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
  CRCHK(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
  CRCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  CRCHK(hipMemsetAsync(h->d_lambda, 0, nb, h->stream));
  CRCHK(hipMemsetAsync(h->d_lambda_acc, 0, nb, h->stream));
  CRCHK(hipMemsetAsync(h->d_gsa, 0, nb, h->stream));
  CRCHK(hipMemsetAsync(h->d_grad, 0, nb, h->stream));
  CRCHK(hipMemsetAsync(h->d_sums, 0, sizeof(double) * 8, h->stream));
  for (int i = 0; i <= SCRF_N_PHASES; i++) {
    CRCHK(hipEventCreate(&h->ev[i][0]));
    CRCHK(hipEventCreate(&h->ev[i][1]));
  }
  h->ev_ok = true;
  CRCHK(hipStreamSynchronize(h->stream));
  if (h->shadow) {
    std::vector<double> z(h->xlay.lambda_len, 0.0);
    if (scrf_set_lambda(h, z.data(), h->xlay.lambda_len) != SCRF_OK) { g_create_err = h->err; scrf_destroy(h); return SCRF_ERR_HIP; }
  }
#undef CRCHK
  *out = h;
  return SCRF_OK;
}

extern "C" int scrf_destroy(scrf_handle h) {
  if (!h) return SCRF_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  hipFree(h->d_lambda); hipFree(h->d_lambda_acc); hipFree(h->d_gsa);
  if (h->own_grad) hipFree(h->d_grad);
  hipFree(h->d_grad2); hipFree(h->d_sums2); hipFree(h->scratch2);
  hipFree(h->d_stage); hipFree(h->d_sums_stage); hipFree(h->d_latch);
  if (h->h_latch) hipHostFree(h->h_latch);
  if (h->h_sums) hipHostFree(h->h_sums);
  if (h->ev_sums) hipEventDestroy(h->ev_sums);
  if (h->ev_status) hipEventDestroy(h->ev_status);
  if (h->kev[0]) hipEventDestroy(h->kev[0]);
  if (h->kev[1]) hipEventDestroy(h->kev[1]);
  if (h->stream2) { hipStreamSynchronize(h->stream2); hipStreamDestroy(h->stream2); }
  if (h->up_stream) { hipStreamSynchronize(h->up_stream); hipStreamDestroy(h->up_stream); }
  for (auto& bl : h->pool_free) hipFree(bl.p);   // (the engine stream was synchronised above)
  for (auto& kv : h->pool_cap) hipFree(kv.first);
  for (int i = 0; i < 8; i++) if (h->pool_ev[i]) hipEventDestroy(h->pool_ev[i]);
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  hipFree(h->dec_lab); hipFree(h->dec_n); hipFree(h->dec_cost);
  if (h->dec_hlab) hipHostFree(h->dec_hlab);
  if (h->dec_hn) hipHostFree(h->dec_hn);
  if (h->dec_hcost) hipHostFree(h->dec_hcost);
  hipFree(h->d_w1); hipFree(h->d_dtab); hipFree(h->d_rtab); hipFree(h->d_pack); hipFree(h->d_sl_tab); hipFree(h->d_m0); hipFree(h->d_e0); hipFree(h->d_et0); hipFree(h->d_msh0); hipFree(h->d_sums); hipFree(h->scratch);
  if (h->ev_ok)
    for (int i = 0; i <= SCRF_N_PHASES; i++) { hipEventDestroy(h->ev[i][0]); hipEventDestroy(h->ev[i][1]); }
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return SCRF_OK;
}

extern "C" const char* scrf_last_error(scrf_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int scrf_set_stream(scrf_handle h, void* s) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->own_stream) { hipStreamDestroy(h->stream); h->own_stream = false; }
  if (s) {
    h->stream = (hipStream_t)s;
  } else {
    HIPCHK(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return SCRF_OK;
}

extern "C" int scrf_synchronize(scrf_handle h) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream2));
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// layout hooks / model state
// ---------------------------------------------------------------------------------------------
extern "C" int scrf_lambda_len(scrf_handle h, uint32_t* n) { if (!h || !n) return SCRF_ERR_INVALID; *n = h->xlay.lambda_len; return SCRF_OK; }
extern "C" int scrf_num_state_funcs(scrf_handle h, uint32_t* n) { if (!h || !n) return SCRF_ERR_INVALID; *n = h->lay.nsf; return SCRF_OK; }
extern "C" int scrf_num_trans_funcs(scrf_handle h, uint32_t* n) { if (!h || !n) return SCRF_ERR_INVALID; *n = h->lay.ntf; return SCRF_OK; }
extern "C" int scrf_state_idx(scrf_handle h, uint32_t clab, uint32_t fno, uint32_t* idx) {
  if (!h || !idx) return SCRF_ERR_INVALID;
  if (clab >= h->lay.L) return fail(h, SCRF_ERR_INVALID, "scrf_state_idx: label %u >= %u", clab, h->lay.L);
  *idx = h->xlay.state_idx_k(clab) + fno;  // getStateFeatureIdx :421-423
  return SCRF_OK;
}
extern "C" int scrf_trans_idx(scrf_handle h, uint32_t plab, uint32_t clab, uint32_t fno, uint32_t* idx) {
  if (!h || !idx) return SCRF_ERR_INVALID;
  if (clab >= h->lay.L || plab >= h->lay.L) return fail(h, SCRF_ERR_INVALID, "scrf_trans_idx: label out of range");
  {
    const uint32_t v = h->xlay.trans_idx_k(plab, clab);   // getTransFeatureIdx :435-437; 0xffffffff: no such transition (n-state topology)
    *idx = v == 0xffffffffu ? v : v + fno;
  }
  return SCRF_OK;
}

static int vec_io(scrf_handle h, double* dev, const double* in, double* out, uint32_t n, const char* what) {
  if (!h || (!in && !out)) return SCRF_ERR_INVALID;
  if (n != h->xlay.lambda_len) return fail(h, SCRF_ERR_INVALID, "%s: length %u != lambda_len %u", what, n, h->xlay.lambda_len);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->shadow) {   // compact (caller) <-> dense (device)
    const uint32_t nd = h->lay.lambda_len;
    h->xbuf.assign(nd, 0.0);
    if (in) {
      for (uint32_t i = 0; i < n; i++) h->xbuf[h->s2d[i]] = in[i];
      if (dev == h->d_lambda) for (uint32_t m : h->masked) h->xbuf[m] = h->mask_w;
      HIPCHK(h, hipMemcpyAsync(dev, h->xbuf.data(), sizeof(double) * nd, hipMemcpyHostToDevice, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
    } else {
      HIPCHK(h, hipMemcpyAsync(h->xbuf.data(), dev, sizeof(double) * nd, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      for (uint32_t i = 0; i < n; i++) out[i] = h->xbuf[h->s2d[i]];
    }
    return SCRF_OK;
  }
  if (in) HIPCHK(h, hipMemcpyAsync(dev, in, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  else HIPCHK(h, hipMemcpyAsync(out, dev, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SCRF_OK;
}
extern "C" int scrf_set_lambda(scrf_handle h, const double* v, uint32_t n) {
  int rc = vec_io(h, h ? h->d_lambda : nullptr, v, nullptr, n, "scrf_set_lambda");
  if (rc == SCRF_OK) h->m0_valid = false;
  return rc;
}
extern "C" int scrf_get_lambda(scrf_handle h, double* v, uint32_t n) { return vec_io(h, h ? h->d_lambda : nullptr, nullptr, v, n, "scrf_get_lambda"); }
extern "C" int scrf_set_lambda_acc(scrf_handle h, const double* v, uint32_t n) { return vec_io(h, h ? h->d_lambda_acc : nullptr, v, nullptr, n, "scrf_set_lambda_acc"); }
extern "C" int scrf_get_lambda_acc(scrf_handle h, double* v, uint32_t n) { return vec_io(h, h ? h->d_lambda_acc : nullptr, nullptr, v, n, "scrf_get_lambda_acc"); }
extern "C" int scrf_set_grad_sqr_acc(scrf_handle h, const double* v, uint32_t n) { return vec_io(h, h ? h->d_gsa : nullptr, v, nullptr, n, "scrf_set_grad_sqr_acc"); }
extern "C" int scrf_get_grad_sqr_acc(scrf_handle h, double* v, uint32_t n) { return vec_io(h, h ? h->d_gsa : nullptr, nullptr, v, n, "scrf_get_grad_sqr_acc"); }
extern "C" int scrf_get_grad(scrf_handle h, double* v, uint32_t n) { return vec_io(h, h ? h->d_grad : nullptr, nullptr, v, n, "scrf_get_grad"); }

extern "C" int scrf_zero_grad(scrf_handle h) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemsetAsync(h->d_grad, 0, sizeof(double) * h->lay.lambda_len, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_sums, 0, sizeof(double) * 4, h->stream));
  return SCRF_OK;
}

static const char* k_shadow_devptr = "%s: with crf_states > 1 on a segmental model the device gradient is kept in the dense one-state layout; use scrf_get_grad / scrf_add_grad / scrf_allreduce_grad";
extern "C" int scrf_grad_device_ptr(scrf_handle h, void** p) {
  if (!h || !p) return SCRF_ERR_INVALID;
  if (h->shadow) return fail(h, SCRF_ERR_INVALID, k_shadow_devptr, "scrf_grad_device_ptr");
  *p = h->d_grad;
  return SCRF_OK;
}

extern "C" int scrf_set_grad_buffer(scrf_handle h, void* dptr) {
  // accumulate into caller-owned device memory (e.g. a torch tensor that torch.distributed
  // all-reduces over RCCL); NULL restores the engine's own buffer
  if (!h) return SCRF_ERR_INVALID;
  if (h->shadow) return fail(h, SCRF_ERR_INVALID, k_shadow_devptr, "scrf_set_grad_buffer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (dptr) {
    if (h->own_grad) hipFree(h->d_grad);
    h->d_grad = (double*)dptr;
    h->own_grad = false;
  } else if (!h->own_grad) {
    HIPCHK(h, hipMalloc((void**)&h->d_grad, sizeof(double) * h->lay.lambda_len));
    HIPCHK(h, hipMemsetAsync(h->d_grad, 0, sizeof(double) * h->lay.lambda_len, h->stream));
    h->own_grad = true;
  }
  return SCRF_OK;
}

extern "C" int scrf_get_batch_sums(scrf_handle h, double* sums3) {
  if (!h || !sums3) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpyAsync(sums3, h->d_sums, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SCRF_OK;
}

// The same without stopping the host: the copy is queued behind the work issued so far and read later (a trainer reads
// minibatch k's sums after it has issued minibatch k + 1, whose batch it can then prepare while k's count kernels run).
extern "C" int scrf_queue_batch_sums(scrf_handle h) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  // (a copy queued before and never taken -- a caller that gave up on its minibatch -- is superseded: the image is
  // written in stream order)
  HIPCHK(h, hipMemcpyAsync(h->h_sums, h->d_sums, sizeof(double) * 3, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_sums, h->stream));
  h->sums_queued = true;
  return SCRF_OK;
}
extern "C" int scrf_take_batch_sums(scrf_handle h, double* sums3) {
  if (!h || !sums3) return SCRF_ERR_INVALID;
  if (!h->sums_queued) return fail(h, SCRF_ERR_INVALID, "scrf_take_batch_sums: nothing queued");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventSynchronize(h->ev_sums));
  for (int i = 0; i < 3; i++) sums3[i] = h->h_sums[i];
  h->sums_queued = false;
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// scratch arena
// ---------------------------------------------------------------------------------------------
struct Arena {
  char* base;
  size_t cap, off;
  template <class Tp> Tp* take(size_t n) {
    size_t bytes = (n * sizeof(Tp) + 255) & ~(size_t)255;
    Tp* p = (Tp*)(base + off);
    off += bytes;
    return p;
  }
};
static size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

static int ensure_scratch(scrf_handle h, size_t bytes, int lane = 0) {
  char*& buf = lane ? h->scratch2 : h->scratch;
  size_t& cap = lane ? h->scratch2_cap : h->scratch_cap;
  if (bytes <= cap) return SCRF_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream2));
  if (buf) hipFree(buf);
  buf = nullptr;
  cap = 0;
  hipError_t e = hipMalloc((void**)&buf, bytes);
  if (e != hipSuccess) return fail(h, SCRF_ERR_HIP, "scratch allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e));
  cap = bytes;
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// batches
// ---------------------------------------------------------------------------------------------
// ---- the batch-array pool (see the handle) ----
static size_t pool_class(size_t bytes) {   // size classes: powers of two from 64 KB (a minibatch's arrays vary in size from step to step)
  size_t c = 64u << 10;
  while (c < bytes) c *= 2;
  return c;
}
static void pool_poll(scrf_handle h) {   // which destroy epochs have finished on the device
  while (h->pool_done < h->pool_epoch) {
    hipEvent_t ev = h->pool_ev[(h->pool_done + 1) % 8];
    if (!ev || hipEventQuery(ev) != hipSuccess) break;
    h->pool_done++;
  }
}
static int pool_alloc(scrf_handle h, size_t bytes, void** out) {
  *out = nullptr;
  if (!h->pool_on) { HIPCHK(h, hipMalloc(out, bytes)); return SCRF_OK; }
  const size_t cap = pool_class(bytes);
  pool_poll(h);
  for (size_t i = 0; i < h->pool_free.size(); i++) {
    const auto& bl = h->pool_free[i];
    if (bl.cap == cap && bl.epoch <= h->pool_done) {
      *out = bl.p;
      h->pool_bytes -= bl.cap;
      h->pool_free[i] = h->pool_free.back();
      h->pool_free.pop_back();
      h->pool_cap[*out] = cap;
      return SCRF_OK;
    }
  }
  HIPCHK(h, hipMalloc(out, cap));
  h->pool_cap[*out] = cap;
  return SCRF_OK;
}
static void pool_release(scrf_handle h, void* p) {   // inside a destroy call: the block carries the epoch being closed
  if (!p) return;
  auto it = h ? h->pool_cap.find(p) : decltype(h->pool_cap.find(p))();
  if (!h || !h->pool_on || it == h->pool_cap.end()) { hipFree(p); return; }
  h->pool_free.push_back({p, it->second, h->pool_epoch + 1});
  h->pool_bytes += it->second;
  h->pool_cap.erase(it);
}
static void pool_close_epoch(scrf_handle h) {   // after the releases of one destroy call
  const uint64_t e = h->pool_epoch + 1;
  hipEvent_t& ev = h->pool_ev[e % 8];
  if (ev) { hipEventSynchronize(ev); if (h->pool_done < e - 8 && e >= 8) h->pool_done = e - 8; }   // the slot's old epoch (e - 8) is long past
  else hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEventRecord(ev, h->stream);
  h->pool_epoch = e;
  // keep the pool bounded: beyond 8 GiB of idle blocks, give the finished ones back
  if (h->pool_bytes > ((size_t)8 << 30)) {
    pool_poll(h);
    for (size_t i = 0; i < h->pool_free.size();) {
      if (h->pool_free[i].epoch <= h->pool_done) {
        h->pool_bytes -= h->pool_free[i].cap;
        hipFree(h->pool_free[i].p);
        h->pool_free[i] = h->pool_free.back();
        h->pool_free.pop_back();
      } else i++;
    }
  }
}

template <class Tp>
static int upload(scrf_handle h, Tp** d, const Tp* src, size_t n) {
  *d = nullptr;
  if (n == 0) n = 1;
  void* p = nullptr;
  int rc = pool_alloc(h, sizeof(Tp) * n, &p);
  if (rc != SCRF_OK) return rc;
  *d = (Tp*)p;
  if (src) HIPCHK(h, hipMemcpyAsync(*d, src, sizeof(Tp) * n, hipMemcpyHostToDevice, (h->pool_on && h->pool_up) ? h->up_stream : h->stream));
  return SCRF_OK;
}

// K states per label on the segmental model: P*P + 2L - P boundary arcs per frame after the first instead of L*L
// (decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab_WithoutSegTransFtr.h:563-597)
static uint64_t shadow_num_arcs(scrf_handle h, uint32_t T) {
  const uint64_t L = h->xlay.L, P = L / h->xlay.K;
  return (uint64_t)(T - 1) * (P * P + 2 * L - P) + scrf_seg_base(T, h->xlay.D) * L + L;
}

extern "C" int scrf_batch_destroy(scrf_handle h, scrf_batch b) {
  if (!b) return SCRF_OK;
  if (h) hipSetDevice(h->device);
  if (h && !h->pool_on) hipStreamSynchronize(h->stream);
  // (pool: no synchronisation here -- the blocks wait in the pool until the engine stream has passed this point)
#define BFREE(p) pool_release(h, (void*)(p))
  BFREE(b->d_T); BFREE(b->d_frame_off); BFREE(b->d_seg_off); BFREE(b->d_arc_off);
  BFREE(b->d_labels); BFREE(b->d_next_lab); BFREE(b->d_prev_lab); BFREE(b->d_trans_counts); BFREE(b->d_windows); BFREE(b->d_frame_u); BFREE(b->d_xm_f);
  for (int s = 0; s < SCRF_MAX_STREAMS; s++) { BFREE(b->d_frames[s]); BFREE(b->d_sframe_off[s]); }
  BFREE(b->d_numer); BFREE(b->d_zx); BFREE(b->d_status);
  BFREE(b->d_tiles[0]); BFREE(b->d_tiles[1]); BFREE(b->d_tiles[2]);
#undef BFREE
  if (h && h->pool_on) pool_close_epoch(h);
  delete b;
  return SCRF_OK;
}

extern "C" int scrf_batch_create(scrf_handle h, const scrf_utt* utts, uint32_t n, uint32_t n_streams,
                                 const scrf_stream_recipe* recipes, scrf_batch* out) {
  if (!h || !out || (!utts && n)) return SCRF_ERR_INVALID;
  *out = nullptr;
  HIPCHK(h, hipSetDevice(h->device));
  const ScrfLayout& lay = h->lay;
  if (n == 0) return fail(h, SCRF_ERR_EMPTY, "scrf_batch_create: empty batch");
  if (n >= (1u << 23)) return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: at most 8388607 utterances per batch (the failure latch packs utterance and code into 31 bits)");
  const bool by_windows = utts[0].windows != nullptr;
  if (!by_windows) {
    if (n_streams == 0 || n_streams > SCRF_MAX_STREAMS || !recipes)
      return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: frame input needs 1..%d stream recipes", SCRF_MAX_STREAMS);
    uint32_t tot = 0;
    for (uint32_t s = 0; s < n_streams; s++) {
      if (recipes[s].in_width == 0) return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: stream %u has in_width 0", s);
      tot += window_width(recipes[s], lay.D);
    }
    if (tot != lay.F)
      return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: joined window width %u != num_feas %u", tot, lay.F);
  }
  scrf_batch b = new scrf_batch_s();
  b->U = n;
  b->mode = by_windows ? 0 : 1;
  b->n_streams = by_windows ? 0 : n_streams;
  b->T.resize(n);
  b->frame_off.assign(n + 1, 0); b->seg_off.assign(n + 1, 0); b->arc_off.assign(n + 1, 0);
  bool have_labels = utts[0].labels != nullptr;
  for (uint32_t u = 0; u < n; u++) {
    const scrf_utt& q = utts[u];
    if ((q.windows != nullptr) != by_windows) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: utterance %u mixes window and frame input", u); }
    if ((q.labels != nullptr) != have_labels) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: labels given for some utterances only"); }
    if (q.T == 0) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_EMPTY, "scrf_batch_create: utterance %u: No features read from this sentence.", u); }
    b->T[u] = q.T;
    b->frame_off[u + 1] = b->frame_off[u] + q.T;
    b->seg_off[u + 1] = b->seg_off[u] + scrf_seg_base(q.T, lay.D);
    uint64_t na = lay.K > 1 ? ns_num_arcs(q.T, lay.L, lay.K)
                  : (lay.D == 1 && h->cfg.model_type == SCRF_STDFRAME)
                      ? (uint64_t)lay.L + (uint64_t)(q.T - 1) * lay.L * lay.L + lay.L
                      : h->cfg.model_type == SCRF_STDSEG ? stdseg_num_arcs(q.T, lay.L / lay.D, lay.D)
                      : h->cfg.model_type == SCRF_STDSEG_NO_DUR ? segtrans_num_arcs(q.T, lay.L, lay.D)
                                                                : scrf_arc_base(q.T, lay.L, lay.D) + lay.L;
    b->arc_off[u + 1] = b->arc_off[u] + na;
  }
  const uint64_t NF = b->frame_off[n], NS = b->seg_off[n];
  int rc;
#define BCHK(x) do { rc = (x); if (rc != SCRF_OK) { scrf_batch_destroy(h, b); return rc; } } while (0)
#define BSYNC() do { hipError_t e_ = hipStreamSynchronize((h->pool_on && h->pool_up) ? h->up_stream : h->stream); if (e_ != hipSuccess) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_HIP, "scrf_batch_create: %s", hipGetErrorString(e_)); } } while (0)
  BCHK(upload(h, &b->d_T, b->T.data(), n));
  BCHK(upload(h, &b->d_frame_off, b->frame_off.data(), n + 1));
  BCHK(upload(h, &b->d_seg_off, b->seg_off.data(), n + 1));
  BCHK(upload(h, &b->d_arc_off, b->arc_off.data(), n + 1));
  {
    std::vector<uint32_t> fu(NF);
    for (uint32_t u = 0; u < n; u++) std::fill(fu.begin() + b->frame_off[u], fu.begin() + b->frame_off[u + 1], u);
    BCHK(upload(h, &b->d_frame_u, fu.data(), NF));
    BSYNC();
  }
  if (have_labels) {
    std::vector<uint32_t> lab(NF);
    for (uint32_t u = 0; u < n; u++) memcpy(&lab[b->frame_off[u]], utts[u].labels, sizeof(uint32_t) * utts[u].T);
    BCHK(upload(h, &b->d_labels, lab.data(), NF));
    std::vector<uint32_t> nxt(NF), cnt((size_t)lay.L * lay.L, 0);
    for (uint32_t u = 0; u < n; u++) {
      uint32_t cur = SCRF_LAB_BAD;
      for (uint32_t t = b->T[u]; t-- > 0;) {
        const uint64_t f = b->frame_off[u] + t;
        nxt[f] = cur;
        const uint32_t lb = lab[f];
        if (lb != SCRF_LAB_BAD) {
          // K states per label: a labelled transition the topology lacks matches none of the node's transition
          // terms (nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp:822-876) and adds nothing
          if (h->shadow && cur != SCRF_LAB_BAD && lb < lay.L * lay.D && cur < lay.L * lay.D &&
              h->xlay.trans_idx_k(lb % lay.L, cur % lay.L) == 0xffffffffu)
            nxt[f] = SCRF_LAB_BAD;
          else if (t + 1 < b->T[u] && cur != SCRF_LAB_BAD && lb < lay.L * lay.D && cur < lay.L * lay.D && h->cfg.model_type != SCRF_STDSEG)
            cnt[(size_t)(lb % lay.L) * lay.L + cur % lay.L]++;
          cur = lb;
        }
      }
    }
    BCHK(upload(h, &b->d_next_lab, nxt.data(), NF));
    if (h->cfg.model_type == SCRF_STDSEG_NO_DUR || h->cfg.model_type == SCRF_STDSEG) {
      std::vector<uint32_t> prv(NF);
      for (uint32_t u = 0; u < n; u++) {
        uint32_t cur = SCRF_LAB_BAD;
        for (uint32_t t = 0; t < b->T[u]; t++) {
          const uint64_t f = b->frame_off[u] + t;
          prv[f] = cur;
          if (lab[f] != SCRF_LAB_BAD) cur = lab[f];
        }
      }
      BCHK(upload(h, &b->d_prev_lab, prv.data(), NF));
    }
    BCHK(upload(h, &b->d_trans_counts, cnt.data(), cnt.size()));
    BSYNC();
  }
  if (by_windows) {
    BCHK(upload<float>(h, &b->d_windows, nullptr, NS * lay.F + 64));  // tail pad: wide loads may over-read 12 B
    for (uint32_t u = 0; u < n; u++) {
      hipError_t e = hipMemcpyAsync(b->d_windows + b->seg_off[u] * lay.F, utts[u].windows,
                                    sizeof(float) * (b->seg_off[u + 1] - b->seg_off[u]) * lay.F, hipMemcpyHostToDevice, (h->pool_on && h->pool_up) ? h->up_stream : h->stream);
      if (e != hipSuccess) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_HIP, "window upload failed: %s", hipGetErrorString(e)); }
    }
  } else {
    for (uint32_t s = 0; s < n_streams; s++) {
      b->recipe[s] = recipes[s];
      b->width[s] = window_width(recipes[s], lay.D);
      const uint32_t pad = recipes[s].left_ctx + recipes[s].right_ctx;
      std::vector<uint64_t> so(n + 1, 0);
      for (uint32_t u = 0; u < n; u++) {
        if (!utts[u].frames[s]) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_INVALID, "scrf_batch_create: utterance %u has no frames for stream %u", u, s); }
        so[u + 1] = so[u] + utts[u].T + pad;
      }
      BCHK(upload(h, &b->d_sframe_off[s], so.data(), n + 1));
      BCHK(upload<float>(h, &b->d_frames[s], nullptr, so[n] * recipes[s].in_width + 64));  // tail pad: wide loads may over-read 12 B
      BSYNC();
      for (uint32_t u = 0; u < n; u++) {
        hipError_t e = hipMemcpyAsync(b->d_frames[s] + so[u] * recipes[s].in_width, utts[u].frames[s],
                                      sizeof(float) * (so[u + 1] - so[u]) * recipes[s].in_width, hipMemcpyHostToDevice, (h->pool_on && h->pool_up) ? h->up_stream : h->stream);
        if (e != hipSuccess) { scrf_batch_destroy(h, b); return fail(h, SCRF_ERR_HIP, "frame upload failed: %s", hipGetErrorString(e)); }
      }
    }
  }
  // fused window synthesis: one segment-recipe stream without context whose window is exactly
  // the state feature range, no transition features
  const int f32_cfg = h->cfg.train_precision == SCRF_PREC_FAST32;
  const bool hybrid_first = h->hybrid_first && lay.L > 64 && n_streams == 1 && !lay.use_tf;
  const bool seg_stream0 = !hybrid_first && !by_windows && n_streams >= 1 && h->cfg.model_type != SCRF_STDSEG_NO_DUR && h->cfg.model_type != SCRF_STDSEG && lay.use_sf &&
                           recipes[0].extract_seg_ftr && !recipes[0].left_ctx && !recipes[0].right_ctx && lay.sfs == 0 &&
                           lay.nsfe == 8 * recipes[0].in_width + lay.D && fused_supported(lay, recipes[0].in_width, f32_cfg);
  // "mixed" (round 4, BASELINE config 3's shape): the state features are exactly stream 0's segment-recipe window and the
  // transition features live in the other streams' columns -- the state part takes the fused kernels, the transition
  // part keeps the materialised first-row windows and the dense contractions
  // "hybrid" (round 4, BASELINE config 5's shape): the same stream structure where the fused kernels' LDS images do not
  // fit (L > 64).  The window vectors stay materialised, but the five sampled blocks -- 5 W of the 8 W + D columns, copies
  // of raw frames -- leave the two dense contractions: scores get their share from the per-frame projections P (k_add_p),
  // counts through the per-frame sums Z (k_lin_z, Z^T F), exactly as on the fused path.
  b->hybrid_ok = !by_windows && n_streams == 1 && h->cfg.model_type != SCRF_STDSEG_NO_DUR && h->cfg.model_type != SCRF_STDSEG && lay.use_sf &&
                 !lay.use_tf && recipes[0].extract_seg_ftr && !recipes[0].left_ctx && !recipes[0].right_ctx && lay.sfs == 0 && lay.D >= 2 &&
                 lay.nsfe == 8 * recipes[0].in_width + lay.D && (hybrid_first || !fused_supported(lay, recipes[0].in_width, f32_cfg));
  const bool mixed_ok = seg_stream0 && n_streams >= 2 && lay.use_tf && lay.tfs >= lay.nsfe && h->fuse_mixed;
  if ((seg_stream0 && n_streams == 1 && !lay.use_tf && lay.nsfe == lay.F) || mixed_ok) {
    b->fused_ok = true;
    b->mixed = mixed_ok;
    {
      std::vector<float> xm(NF);
      const uint32_t W0 = recipes[0].in_width;
      for (uint32_t u = 0; u < n; u++) {
        float m = fmaxf(1.0f, (float)fabs(lay.sbv));
        const float* x = utts[u].frames[0];
        for (size_t i = 0; i < (size_t)utts[u].T * W0; i++) m = fmaxf(m, fabsf(x[i]));
        m = nextafterf(m, INFINITY);   // |sbv| was rounded to float
        std::fill(xm.begin() + b->frame_off[u], xm.begin() + b->frame_off[u + 1], m);
      }
      BCHK(upload(h, &b->d_xm_f, xm.data(), NF));
      BSYNC();
    }
    // score tiles: the windows of TB whole frames; expected-count tiles: those of TBE whole frames (<= 64 windows)
    const uint32_t D = lay.D, TB = fused_scores_tb(recipes[0].in_width, D), TBE = fused_expf_frames(D);
    // a third list when the engine trains with the linear window average and its count kernel walks taller tiles
    uint32_t TBL = 0;
    if (h->cfg.train_precision == SCRF_PREC_FASTLIN && fused_la_supported(lay, recipes[0].in_width)) {
      const ScrfFusedExpfPlan plan = fused_expf_plan(lay, recipes[0].in_width, 0, 1);
      if (plan.tile_list == 2) TBL = plan.frames;
    }
    for (int k = 0; k < (TBL ? 3 : 2); k++) {
      std::vector<ScrfTileDesc> td;
      b->tile_off[k].assign(n + 1, 0);
      for (uint32_t u = 0; u < n; u++) {
        const uint32_t T = b->T[u];
        const uint64_t nseg = scrf_seg_base(T, D);
        uint32_t t = 0;
        for (uint64_t r0 = 0; r0 < nseg;) {
          ScrfTileDesc q;
          memset(&q, 0, sizeof(q));
          uint32_t t_end;   // one past the last frame touched
          uint64_t r1;
          t_end = std::min(T, t + (k == 0 ? TB : k == 1 ? TBE : TBL));
          r1 = scrf_seg_base(t_end, D);
          q.r0 = (uint32_t)r0; q.t0 = t; q.back = (uint16_t)std::min(t, D - 1);
          q.nfr = (uint16_t)(t_end - t); q.nrows = (uint16_t)(r1 - r0);
          q.row_abs = b->seg_off[u] + r0;
          q.fr_abs = b->frame_off[u] + t - q.back;
          td.push_back(q);
          r0 = r1;
          t = t_end;
        }
        b->tile_off[k][u + 1] = td.size();
      }
      BCHK(upload(h, &b->d_tiles[k], td.data(), td.size()));
      BSYNC();
    }
  }
  BCHK(upload<double>(h, &b->d_numer, nullptr, n));
  BCHK(upload<double>(h, &b->d_zx, nullptr, n));
  BCHK(upload<int>(h, &b->d_status, nullptr, n));
#undef BCHK
  BSYNC();
#undef BSYNC
  *out = b;
  return SCRF_OK;
}

extern "C" int scrf_batch_info(scrf_handle h, scrf_batch b, uint32_t* n_utts, uint64_t* n_frames, uint64_t* n_segs, uint64_t* n_arcs) {
  if (!h || !b) return SCRF_ERR_INVALID;
  if (n_utts) *n_utts = b->U;
  if (n_frames) *n_frames = b->frame_off[b->U];
  if (n_segs) *n_segs = b->seg_off[b->U];
  if (n_arcs) {
    *n_arcs = b->arc_off[b->U];
    if (h->shadow) {
      *n_arcs = 0;
      for (uint32_t u = 0; u < b->U; u++) *n_arcs += shadow_num_arcs(h, b->T[u]);
    }
  }
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// chunk planning and the per-chunk pipeline
// ---------------------------------------------------------------------------------------------
enum { PH_WIN = 0, PH_SCORE = 1, PH_FB = 2, PH_EXPF = 3, PH_REDUCE = 4, PH_VIT = 5, PH_ALL = 6, PH_K_SCORE = 7, PH_K_DP = 8, PH_K_EXPF = 9 };
#define EXPF_ROWS_PER_CHUNK 4096ull
// split-K plan of the state expected-count contraction: about 1024 K-chunks (2 per CU-slot
// of the 512-thread MFMA kernel), at least 4096 rows each
static uint64_t expf_rows_per_chunk(uint64_t nseg) {
  uint64_t rpc = (nseg + 1023) / 1024;
  rpc = (rpc + 31) & ~31ull;
  return rpc < EXPF_ROWS_PER_CHUNK ? EXPF_ROWS_PER_CHUNK : rpc;
}

// K-chunks of the per-window transition contraction (STDSEG_NO_DUR: [N_seg][L*L] posteriors x a few feature functions):
// one per 2048 rows while the partial-sum slabs stay within 128 MB (at least 4)
static uint32_t segtrans_chunks(uint64_t nseg, size_t LL, uint32_t ntf) {
  const uint64_t cap = std::max<uint64_t>(4, (128ull << 20) / (LL * ntf * sizeof(double)));
  return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(cap, (nseg + 2047) / 2048));
}

// K-chunks (row chunks) of the per-frame transition contraction (stdtrans: [frames][L*L] posteriors x the transition
// feature functions).  The wide form of k_expf_mfma holds one 8-wavefront workgroup per CU, so the launch runs in rounds
// of 256 workgroups: with the TIMIT demo's 240 tiles per chunk, 4 chunks are 3.75 rounds (the last one three quarters
// full), 16 chunks are 15 whole rounds.  Smallest count from 4 up to 16 (at least 1024 rows each) whose last round is >= 97 %
// full, else the fullest; SCRF_TRANS_CHUNKS overrides (A/B runs).
static uint32_t transframe_chunks(uint64_t nfr, uint32_t n_out, uint32_t nfun, int f32) {
  const uint32_t base = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(4, (nfr + 2047) / 2048));
  static const int forced = getenv("SCRF_TRANS_CHUNKS") ? atoi(getenv("SCRF_TRANS_CHUNKS")) : 0;
  if (forced > 0) return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)forced, std::max<uint64_t>(1, nfr / 64)));
  const uint32_t tiles = expf_mfma_wide_tiles(n_out, nfun, f32);
  if (tiles == 0 || base < 4) return base;
  const uint32_t cus = 256;
  uint32_t best = base;
  double best_fill = 0.0;
  // every chunk owns a slab of partial sums [n_out][nfun]: no more chunks than keep the slabs within 1 GB (the TIMIT
  // demo: 34.5 MB each)
  const uint64_t slab_bytes = (uint64_t)n_out * nfun * sizeof(double);
  const uint32_t n_max = (uint32_t)std::min<uint64_t>(16, std::max<uint64_t>(base, (1ull << 30) / std::max<uint64_t>(1, slab_bytes)));
  for (uint32_t n = base; n <= n_max && nfr / n >= 1024; n++) {
    const uint64_t wg = (uint64_t)tiles * n;
    const double fill = (double)wg / (double)(((wg + cus - 1) / cus) * cus);
    if (fill > best_fill + 1e-9) { best_fill = fill; best = n; }
    if (fill >= 0.97) { best = n; break; }
  }
  return best;
}

// K-chunks of the transition-bias contraction: one wavefront each, about 4096 of them (four per SIMD: a wavefront's
// next operands are only one 576-cycle group of MFMAs ahead, the others cover the rest of the load latency)
static uint64_t atb_rows_per_chunk(uint64_t nfr) {
  uint64_t rpc = ((nfr + 4095) / 4096 + 3) & ~3ull;
  return rpc < 64 ? 64 : rpc;
}

struct ChunkBufs {
  hipStream_t st = nullptr;  // stream of the lane this chunk runs on
  double* grad = nullptr;    // gradient / batch sums this lane accumulates into
  double* sums = nullptr;
  float* X = nullptr;        // window vectors of the chunk (scratch, or a view into the batch)
  double* S = nullptr;       // [nseg][L]
  double* M = nullptr;       // [nfr][L*L] or engine M0
  int m_per_frame = 0;
  double* AD = nullptr;      // [nseg][L]
  double* alpha = nullptr;   // [nfr][L]
  double* beta = nullptr;    // [nfr][L] (parity hooks only)
  double* XI = nullptr;      // [nfr][L*L]
  double* xi_acc = nullptr;  // [nutt][L*L]
  uint64_t* xrow_cur = nullptr;
  uint64_t* xrow_next = nullptr;
  double* slab_s = nullptr;
  double* slab_t = nullptr;
  uint16_t* bp_b = nullptr;
  uint16_t* bp_e = nullptr;
  // fast decode: float arc weights of the fused score kernel + the entries to recompute
  float* Wn = nullptr;       // [nseg][L]
  uint64_t* fix_list = nullptr;
  uint32_t* fix_cnt = nullptr;
  uint32_t fix_cap = 0;
  uint32_t nch_s = 0, nch_t = 0;
  uint64_t rpc_s = 0, rpc_t = 0;
  // wavefront-per-utterance DP
  bool wave = false;
  double* E = nullptr;       // exp(M - shift) per frame, or the engine's E0
  double* ET = nullptr;
  double* msh = nullptr;
  double* sd = nullptr;      // [nfr][L]
  double* fA = nullptr;      // [nfr][L] xi factors
  double* fB = nullptr;
  double* numer_f = nullptr; // [nfr]
  double* mass_s = nullptr;  // [nfr] state posterior mass per node (reference self-check, computeExpF :917-947)
  // scaled linear-domain recursion (training path of the wavefront DP)
  bool lin = false;
  bool es_ready = false;     // cb.S already holds exp(S - smax) (written by the fused score kernel)
  bool z_ready = false;      // cb.Z already holds the per-frame sums of R (written by k_post_z)
  ScrfDpLin dl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double* smax = nullptr;    // [nseg] row maxima of the scores
  double* s_true = nullptr;  // [nfr] score of the labelled window ending at each frame
  double* R = nullptr;       // [nseg][L] Y - gamma: cb.S (linear path) or cb.AD
  double* slab_atb = nullptr;
  uint32_t nch_atb = 0;
  uint64_t rpc_atb = 0;
  bool fused = false;        // window synthesis fused into the contractions (no X)
  double* P = nullptr;       // [nfr][5L] per-frame projections of the sampled blocks
  double* Z = nullptr;       // [nfr][5L] per-frame sums of R over the windows sampling the frame
  double* slab_l = nullptr;
  uint32_t nch_l = 0;
  uint64_t rpc_l = 0;
  bool hybrid = false;       // materialised X, but the sampled blocks through P / Z (scrf_batch::hybrid_ok)
  bool la = false;           // SCRF_PREC_FASTLIN: linear window average (6 groups in P / Z, no avg group in the dense parts)
  double* slab_d = nullptr;  // duration + bias counts of the wave-specialised count kernel (behind slab_s)
  int expf_tiles = 1;        // tile list the fused count kernel walks
};

struct Need { bool fb, post, beta, vit; bool fused = false; bool vitfast = false; bool la = false; bool hybrid = false; };

// entries the decode screen may list per chunk before the chunk falls back to the EXACT path
static uint32_t decode_fix_cap(uint64_t nseg, uint32_t L) {
  const uint64_t c = nseg * L / 64 + 4096;
  return (uint32_t)std::min<uint64_t>(c, 1u << 26);
}

// does the recursion run on the wavefront kernels?  L <= 64: always (log-domain kernels for the
// hooks, linear-domain for training); 64 < L <= 256: the multi-wavefront linear-domain kernel,
// training path only
// STDSEG_NO_DUR: one transition matrix per window (scrf_segtrans.hip); its own workgroup recursion
static bool segtrans(scrf_handle h) { return h->cfg.model_type == SCRF_STDSEG_NO_DUR; }

static bool wave_path(scrf_handle h, bool post) {
  if (h->force_fb || segtrans(h)) return false;
  return dp_wave_supported(h->lay) || (post && h->lin_dp && (dplin_mw_supported(h->lay) || dplin_supported(h->lay)));
}

// fused path: the five sampled blocks as per-frame projections (outputs (k, label), k < 5), and
// the dense column groups [avg | max | min | onehot(d)] + bias
static ScrfGemmSpec spec_samples(uint32_t W) { return ScrfGemmSpec{2, 0, W, 0, 0.0, 0, W}; }
static ScrfGemmSpec spec_dense(const ScrfLayout& l, uint32_t W) {
  return ScrfGemmSpec{0, 0, 3 * W + l.D, (uint32_t)l.use_sb, l.sbv, 5 * W, 0};
}
// hybrid path: the materialised row holds [avg | max | min | onehot(d)] only (k_windows without the sampled blocks), padded
// to whole 16-byte groups
static uint32_t hybrid_row_floats(const ScrfLayout& l, uint32_t W) { return (3 * W + l.D + 3) & ~3u; }
static ScrfGemmSpec spec_stats_x(const ScrfLayout& l, uint32_t W) {   // [avg | max | min] of that row
  (void)l;
  return ScrfGemmSpec{0, 0, 3 * W, 0, 0.0, 5 * W, 0};
}
static ScrfGemmSpec spec_dense_x(const ScrfLayout& l, uint32_t W) {   // the whole row + bias
  return ScrfGemmSpec{0, 0, 3 * W + l.D, (uint32_t)l.use_sb, l.sbv, 5 * W, 0};
}

static size_t chunk_bytes(scrf_handle h, scrf_batch b, uint64_t nutt, uint64_t nfr, uint64_t nseg, const Need& nd) {
  const ScrfLayout& l = h->lay;
  const size_t LL = (size_t)l.L * l.L;
  size_t tot = 0;
  const uint32_t W0 = b->mode == 1 ? b->recipe[0].in_width : 0;
  if (nd.fused) {
    const size_t ng = nd.la ? 6 : 5;
    tot += pad256(nfr * ng * l.L * sizeof(double));                // P (scores) / Z (counts)
    if (nd.post) tot += pad256((size_t)512 * ng * l.L * W0 * sizeof(double));
    if (b->mixed) tot += pad256(nseg * l.F * sizeof(float));     // the transition streams' windows
  } else if (b->mode == 1) tot += pad256(nseg * (nd.hybrid ? hybrid_row_floats(l, W0) : l.F) * sizeof(float));
  if (nd.hybrid) tot += pad256(nfr * 5 * l.L * sizeof(double)) + pad256((size_t)512 * 5 * l.L * W0 * sizeof(double)) +   // P / Z, slab_l
                        pad256((size_t)(nutt + 1024) * l.L * (l.D + 1) * sizeof(double));                                 // per-duration sums
  if (nd.vitfast) tot += pad256(nseg * l.L * sizeof(float)) + pad256((size_t)decode_fix_cap(nseg, l.L) * 8) + 256;  // Wn, list, count
  else tot += pad256(nseg * l.L * sizeof(double));                  // S
  if (segtrans(h)) tot += pad256(nseg * LL * sizeof(double));               // M2: one matrix per window
  else if (l.use_tf) tot += pad256(nfr * LL * sizeof(double)) + pad256(nfr * 8);  // M, xrow_cur
  if (nd.fb) {
    const bool wave = wave_path(h, nd.post);
    if (wave && nd.post && h->lin_dp) {
      // scaled linear-domain recursion: no alpha-with-duration array
      tot += pad256(nseg * sizeof(double)) + pad256(nfr * sizeof(double));               // smax, s_true
      tot += 4 * pad256(nfr * l.L * sizeof(double)) + 4 * pad256(nfr * sizeof(double));  // a, p, b, sd + log-scales
    } else {
      tot += pad256(nseg * l.L * sizeof(double));                     // AD
      tot += pad256(nfr * l.L * sizeof(double));                      // alpha
      if (nd.beta || wave || segtrans(h)) tot += pad256(nfr * l.L * sizeof(double));
      if (wave) tot += pad256(nfr * l.L * sizeof(double));            // sd
      if (wave && nd.post) tot += 2 * pad256(nfr * l.L * sizeof(double));  // A, B
    }
    if (wave) {
      if (l.use_tf) tot += 2 * pad256(nfr * LL * sizeof(double)) + pad256(nfr * sizeof(double));  // E, ET, shift
      if (nd.post) {
        tot += pad256(nfr * sizeof(double));                          // numer_f
        if (!l.use_tf) tot += pad256(((nfr + atb_rows_per_chunk(nfr) - 1) / atb_rows_per_chunk(nfr)) * LL * sizeof(double));
      }
    }
    if (nd.post) {
      tot += pad256(nfr * sizeof(double));                            // mass_s
      if (segtrans(h)) {
        tot += pad256(nseg * LL * sizeof(double));                    // XI2
        tot += pad256((size_t)segtrans_chunks(nseg, LL, l.ntf) * LL * l.ntf * sizeof(double));
      } else if (l.use_tf) {
        tot += pad256(nfr * LL * sizeof(double)) + pad256(nfr * 8);  // XI, xrow_next
        uint32_t nch_t = transframe_chunks(nfr, (uint32_t)LL, scrf_spec_trans(l).nfun(), h->cfg.train_precision == SCRF_PREC_FAST32);
        tot += pad256((size_t)nch_t * LL * l.ntf * sizeof(double));
      } else if (!wave_path(h, nd.post)) {
        tot += pad256(nutt * LL * sizeof(double));
      }
      const uint64_t rpc_s = expf_rows_per_chunk(nseg);
      uint64_t nch_s = nd.fused ? 512 : (nseg + rpc_s - 1) / rpc_s;
      tot += pad256(nch_s * l.L * l.nsf * sizeof(double));
    }
  }
  if (nd.vit) tot += 2 * pad256(nfr * l.L * sizeof(uint16_t));
  return tot + 4096;
}

// largest u1 > u0 whose chunk fits the budget (always at least one utterance)
static uint32_t plan_chunk(scrf_handle h, scrf_batch b, uint32_t u0, const Need& nd) {
  uint32_t u1 = u0 + 1;
  const uint32_t max_utts = 65535;  // grid.x of the per-frame kernels stays small enough anyway
  while (u1 < b->U && u1 - u0 < max_utts) {
    uint64_t nfr = b->frame_off[u1 + 1] - b->frame_off[u0], nseg = b->seg_off[u1 + 1] - b->seg_off[u0];
    if (chunk_bytes(h, b, u1 + 1 - u0, nfr, nseg, nd) > h->cfg.scratch_bytes) break;
    if (nfr > 0x7fffffffull) break;
    u1++;
  }
  return u1;
}

static int carve(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, const Need& nd, ChunkBufs* cb, int lane = 0) {
  const ScrfLayout& l = h->lay;
  const size_t LL = (size_t)l.L * l.L;
  const uint64_t nutt = u1 - u0, nfr = b->frame_off[u1] - b->frame_off[u0], nseg = b->seg_off[u1] - b->seg_off[u0];
  size_t need = chunk_bytes(h, b, nutt, nfr, nseg, nd);
  int rc = ensure_scratch(h, need, lane);
  if (rc != SCRF_OK) return rc;
  Arena a{lane ? h->scratch2 : h->scratch, lane ? h->scratch2_cap : h->scratch_cap, 0};
  cb->st = lane ? h->stream2 : h->stream;
  cb->grad = lane ? h->d_grad2 : h->d_grad;
  cb->sums = lane ? h->d_sums2 : h->d_sums;
  if (nd.fused) {
    const uint32_t W0 = b->recipe[0].in_width;
    cb->X = b->mixed ? a.take<float>(nseg * l.F) : nullptr;
    cb->la = nd.la;
    const size_t ng = nd.la ? 6 : 5;
    cb->P = a.take<double>(nfr * ng * l.L);
    cb->Z = cb->P;  // the projections are dead once the scores exist
    if (nd.post) {
      cb->rpc_l = ((nfr + 511) / 512 + 31) & ~31ull;   // <= 512 K-chunks of whole 4-frame groups
      cb->nch_l = (uint32_t)((nfr + cb->rpc_l - 1) / cb->rpc_l);
      cb->slab_l = a.take<double>((size_t)512 * ng * l.L * W0);
    }
  } else if (b->mode == 1) cb->X = a.take<float>(nseg * (nd.hybrid ? hybrid_row_floats(l, b->recipe[0].in_width) : l.F));
  else cb->X = b->d_windows + b->seg_off[u0] * l.F;
  if (nd.hybrid) {
    cb->hybrid = true;
    cb->P = a.take<double>(nfr * 5 * l.L);
    cb->Z = cb->P;  // the projections are dead once they are added to the scores
    cb->rpc_l = ((nfr + 511) / 512 + 31) & ~31ull;
    cb->nch_l = (uint32_t)((nfr + cb->rpc_l - 1) / cb->rpc_l);
    cb->slab_l = a.take<double>((size_t)512 * 5 * l.L * b->recipe[0].in_width);
    cb->slab_d = a.take<double>((size_t)(nutt + 1024) * l.L * (l.D + 1));
  }
  cb->fused = nd.fused;
  if (nd.vitfast) {
    cb->Wn = a.take<float>(nseg * l.L);
    cb->fix_cap = decode_fix_cap(nseg, l.L);
    cb->fix_list = a.take<uint64_t>(cb->fix_cap);
    cb->fix_cnt = a.take<uint32_t>(64);
  } else {
    cb->S = a.take<double>(nseg * l.L);
  }
  if (segtrans(h)) {
    cb->M = a.take<double>(nseg * LL);
    cb->m_per_frame = 2;   // per window
  } else if (l.use_tf) {
    cb->M = a.take<double>(nfr * LL);
    cb->xrow_cur = a.take<uint64_t>(nfr);
    cb->m_per_frame = 1;
  } else {
    cb->M = h->d_m0;
    cb->m_per_frame = 0;
  }
  if (nd.fb) {
    cb->wave = wave_path(h, nd.post);
    cb->lin = cb->wave && nd.post && h->lin_dp;
    if (cb->lin) {
      cb->smax = a.take<double>(nseg);
      cb->s_true = a.take<double>(nfr);
      cb->dl.a = a.take<double>(nfr * l.L);  cb->dl.ga = a.take<double>(nfr);
      cb->dl.p = a.take<double>(nfr * l.L);  cb->dl.gp = a.take<double>(nfr);
      cb->dl.b = a.take<double>(nfr * l.L);  cb->dl.gb = a.take<double>(nfr);
      cb->dl.sd = a.take<double>(nfr * l.L); cb->dl.gsd = a.take<double>(nfr);
      cb->R = cb->S;              // R = Y - gamma overwrites the exponentiated scores
      cb->fA = cb->dl.a;          // xi factors: a, and sd rescaled in place
      cb->fB = cb->dl.sd;
    } else {
      cb->AD = a.take<double>(nseg * l.L);
      cb->R = cb->AD;
      cb->alpha = a.take<double>(nfr * l.L);
      if (nd.beta || cb->wave || segtrans(h)) cb->beta = a.take<double>(nfr * l.L);
      if (cb->wave) cb->sd = a.take<double>(nfr * l.L);
      if (cb->wave && nd.post) {
        cb->fA = a.take<double>(nfr * l.L);
        cb->fB = a.take<double>(nfr * l.L);
      }
    }
    if (cb->wave) {
      if (l.use_tf) {
        cb->E = a.take<double>(nfr * LL);
        cb->ET = a.take<double>(nfr * LL);
        cb->msh = a.take<double>(nfr);
      } else {
        cb->E = h->d_e0; cb->ET = h->d_et0; cb->msh = h->d_msh0;
      }
      if (nd.post) {
        cb->numer_f = a.take<double>(nfr);
        if (!l.use_tf) {
          cb->rpc_atb = atb_rows_per_chunk(nfr);
          cb->nch_atb = (uint32_t)((nfr + cb->rpc_atb - 1) / cb->rpc_atb);
          cb->slab_atb = a.take<double>((size_t)cb->nch_atb * LL);
        }
      }
    }
    if (nd.post) {
      cb->mass_s = a.take<double>(nfr);
      if (segtrans(h)) {
        cb->XI = a.take<double>(nseg * LL);
        cb->nch_t = segtrans_chunks(nseg, LL, l.ntf);
        cb->rpc_t = (nseg + cb->nch_t - 1) / cb->nch_t;
        cb->slab_t = a.take<double>((size_t)cb->nch_t * LL * l.ntf);
      } else if (l.use_tf) {
        cb->XI = a.take<double>(nfr * LL);
        cb->xrow_next = a.take<uint64_t>(nfr);
        cb->nch_t = transframe_chunks(nfr, (uint32_t)LL, scrf_spec_trans(l).nfun(), h->cfg.train_precision == SCRF_PREC_FAST32);
        cb->rpc_t = (nfr + cb->nch_t - 1) / cb->nch_t;
        cb->slab_t = a.take<double>((size_t)cb->nch_t * LL * l.ntf);
      } else if (!cb->wave) {
        cb->xi_acc = a.take<double>(nutt * LL);
      }
      cb->rpc_s = expf_rows_per_chunk(nseg);
      cb->nch_s = (uint32_t)((nseg + cb->rpc_s - 1) / cb->rpc_s);
      ScrfFusedExpfPlan plan;
      if (nd.fused) {
        plan = fused_expf_plan(l, b->recipe[0].in_width, h->cfg.train_precision == SCRF_PREC_FAST32, nd.la);
        cb->expf_tiles = plan.tile_list;
        cb->nch_s = fused_expf_blocks(l, b->recipe[0].in_width, h->cfg.train_precision == SCRF_PREC_FAST32,
                                      b->tile_off[plan.tile_list][u1] - b->tile_off[plan.tile_list][u0], nd.la);
      }
      cb->slab_s = a.take<double>((size_t)(nd.fused ? 512 : cb->nch_s) * l.L * l.nsf);
      // dense columns + durations + bias <= nsf: the duration slab fits behind the dense one
      if (nd.fused) cb->slab_d = cb->slab_s + (size_t)cb->nch_s * l.L * plan.ncol;
    }
  }
  if (nd.vit) {
    cb->bp_b = a.take<uint16_t>(nfr * l.L);
    cb->bp_e = a.take<uint16_t>(nfr * l.L);
  }
  if (a.off > a.cap) return fail(h, SCRF_ERR_INVALID, "internal: scratch arena overflow");
  return SCRF_OK;
}

struct PhaseTimer {
  scrf_handle h;
  int ph;
  hipStream_t st;
  PhaseTimer(scrf_handle h_, int p, hipStream_t s_ = nullptr) : h(h_), ph(p), st(s_ ? s_ : h_->stream) {
    if (h->timing) hipEventRecord(h->ev[ph][0], st);
  }
  void stop(uint32_t launches) {
    if (!h->timing) return;
    hipEventRecord(h->ev[ph][1], st);
    hipEventSynchronize(h->ev[ph][1]);
    float ms = 0;
    hipEventElapsedTime(&ms, h->ev[ph][0], h->ev[ph][1]);
    h->ms[ph] += ms;
    h->nlaunch[ph] += launches;
  }
};

// HIP-event time of one kernel launch (or a few belonging together), accumulated by name; only when
// timing is enabled (the stop waits for the kernel, so it serialises the host with the device)
struct KernelTimer {
  scrf_handle h;
  const char* name;
  hipStream_t st;
  KernelTimer(scrf_handle h_, const char* n, hipStream_t s_) : h(h_), name(n), st(s_) {
    if (h->timing) hipEventRecord(h->kev[0], st);
  }
  void stop(uint32_t launches = 1) {
    if (!h->timing) return;
    hipEventRecord(h->kev[1], st);
    hipEventSynchronize(h->kev[1]);
    float ms = 0;
    hipEventElapsedTime(&ms, h->kev[0], h->kev[1]);
    for (auto& k : h->ktimes)
      if (k.name == name) { k.ms += ms; k.n += launches; return; }
    h->ktimes.push_back({name, ms, launches});
  }
};
#define KT_RUN(name, st, ...) do { KernelTimer kt_(h, name, st); __VA_ARGS__; kt_.stop(); } while (0)

static ScrfFusedArgs fused_args(scrf_handle h, scrf_batch b, uint32_t u0, int which) {
  ScrfFusedArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.frames = b->d_frames[0];
  fa.tiles = b->d_tiles[which];
  fa.tile0 = b->tile_off[which][u0];
  fa.row_base = b->seg_off[u0];
  fa.frame_base = b->frame_off[u0];
  fa.W = b->recipe[0].in_width;
  fa.TB = fused_scores_tb(fa.W, h->lay.D);
  return fa;
}

// windows + exact scores of a chunk (both training and decode start here)
static int run_scores(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, ChunkBufs& cb, bool fast = false, int f32 = 0) {
  const ScrfLayout& l = h->lay;
  const uint64_t nfr = b->frame_off[u1] - b->frame_off[u0], nseg = b->seg_off[u1] - b->seg_off[u0];
  ScrfBatchView bv = b->view();
  if (cb.fused) {
    PhaseTimer tm(h, PH_SCORE, cb.st);
    const uint32_t W0 = b->recipe[0].in_width;
    ScrfFusedArgs fa = fused_args(h, b, u0, 0);
    if (h->d_dtab && h->dur_table) {
      launch_dur_table(cb.st, l, W0, h->d_lambda, h->d_dtab);
      fa.dtab = h->d_dtab;
      const size_t nb = fused_tile_table_bytes(l.D, fa.TB);
      if (nb > h->rtab_bytes) {
        hipFree(h->d_rtab); h->d_rtab = nullptr; h->rtab_bytes = 0;
        HIPCHK(h, hipMalloc(&h->d_rtab, nb));
        h->rtab_bytes = nb;
      }
      launch_tile_tables(cb.st, l.D, fa.TB, (cb.la && !cb.Wn) ? 1 : 0, h->d_rtab);
      static const bool rtab_on = !(getenv("SCRF_RTAB") && atoi(getenv("SCRF_RTAB")) == 0);   // A/B knob
      if (rtab_on) fa.rtab = h->d_rtab;
    }
    // per-frame projections of the five sampled blocks, then the dense part + gather
    if (pframe_supported(W0)) {
      KT_RUN("k_pframe", cb.st, launch_pframe(cb.st, b->d_frames[0] + b->frame_off[u0] * W0, W0, nfr, h->d_lambda, l, (cb.la ? 6 : 5) * l.L, cb.P));
      if (cb.la) KT_RUN("k_avg_prefix", cb.st, launch_avg_prefix(cb.st, bv, u0, u1 - u0, l.L, cb.P));
    } else {
      KT_RUN("k_scores_mfma(samples)", cb.st, launch_scores_mfma(cb.st, b->d_frames[0] + b->frame_off[u0] * W0, W0, nullptr, nfr, h->d_lambda, l,
                         spec_samples(W0), (cb.la ? 6 : 5) * l.L, cb.P));
      if (cb.la) KT_RUN("k_avg_prefix", cb.st, launch_avg_prefix(cb.st, bv, u0, u1 - u0, l.L, cb.P));
    }
    if (cb.Wn) {
      // decode: float arc weights + the rounding screen
      launch_state_l1(cb.st, h->d_lambda, l, h->d_w1);
      HIPCHK(h, hipMemsetAsync(cb.fix_cnt, 0, sizeof(uint32_t), cb.st));
      ScrfDecodeOut dz;
      dz.wneg = cb.Wn; dz.w1 = h->d_w1; dz.xm_f = b->d_xm_f + b->frame_off[u0];
      dz.cnt = cb.fix_cnt; dz.list = cb.fix_list; dz.cap = cb.fix_cap;
      // |fused - reference order| <= (gamma_n + gamma_m) * sum_f |x_f lambda_f|, u = 2^-53: n = nsfe + 1
      // separately rounded products added one by one; m <= 3 W + 8 roundings on any term's way through
      // the fused evaluation (MFMA accumulation of the 3W dense columns or a W-term P dot, gather and
      // epilogue adds).  1 % covers gamma's denominator and the rounding of the bound itself.
      dz.bound_scale = h->decode_bound_factor * 1.01 * 0x1p-53 * (double)(l.nsfe + 1 + 3 * W0 + 8);
      PhaseTimer tk(h, PH_K_SCORE, cb.st);
      KT_RUN("k_scores_fused(decode)", cb.st, launch_scores_fused_decode(cb.st, fa, l, h->d_lambda, cb.P, b->tile_off[0][u1] - b->tile_off[0][u0], dz));
      tk.stop(1);
      tm.stop(3);
      HIPCHK(h, hipGetLastError());
      return SCRF_OK;
    }
    // on the linear-domain path the epilogue already exponentiates the rows (L <= 48)
    cb.es_ready = cb.lin && l.L <= 48;
    {
      PhaseTimer tk(h, PH_K_SCORE, cb.st);
      KT_RUN("k_scores_fused", cb.st, launch_scores_fused(cb.st, fa, l, h->d_lambda, cb.P, b->tile_off[0][u1] - b->tile_off[0][u0], cb.S, f32,
                          cb.es_ready ? cb.smax : nullptr, cb.s_true, b->d_labels, cb.la ? 1 : 0));
      tk.stop(1);
    }
    tm.stop(2);
    HIPCHK(h, hipGetLastError());
    if (!b->mixed) return SCRF_OK;
    // mixed: the transition part below (windows of the other streams, per-frame transition scores)
  }
  if (b->mode == 1) {
    PhaseTimer tm(h, PH_WIN, cb.st);
    uint32_t col = 0;
    for (uint32_t s = 0; s < b->n_streams; s++) {
      const scrf_stream_recipe& r = b->recipe[s];
      if (cb.fused && s == 0) { col += b->width[s]; continue; }   // stream 0 is synthesised inside the fused kernels
      // FAST training path with per-frame transition features: a stream whose columns hold no state feature is read through
      // the frame rows (k_frame_rows: the node's first window) only -- its other window rows are not written (the TIMIT
      // demo's +-6-frame context stream: 5.8 GB of the 9.3 GB window image).  The window hook (scrf_windows) and EXACT
      // precision materialise every row.
      const bool first_only = fast && l.use_tf && !segtrans(h) && l.D > 1 && (col > l.sfe || col + b->width[s] <= l.sfs);
      KT_RUN("k_windows", cb.st, launch_windows(cb.st, b->d_frames[s], b->d_sframe_off[s], bv, u0, u1, nfr, r.in_width, l.D, r.left_ctx,
                     r.right_ctx, r.extract_seg_ftr, cb.X, cb.hybrid ? hybrid_row_floats(l, r.in_width) : l.F, col,
                     (first_only ? 1 : 0) | (cb.hybrid ? 2 : 0)));
      col += b->width[s];
    }
    tm.stop(b->n_streams);
  }
  PhaseTimer tm(h, PH_SCORE, cb.st);
  uint32_t nl = 1;
  if (!cb.fused && cb.hybrid) {
    // dense statistics [avg | max | min | onehot(d)] + bias from X (3 W + D of its 8 W + D columns), the five sampled blocks
    // as per-frame projections added by row
    const uint32_t W0 = b->recipe[0].in_width;
    if (pframe_supported(W0)) KT_RUN("k_pframe", cb.st, launch_pframe(cb.st, b->d_frames[0] + b->frame_off[u0] * W0, W0, nfr, h->d_lambda, l, 5 * l.L, cb.P));
    else KT_RUN("k_scores_mfma(samples)", cb.st, launch_scores_mfma(cb.st, b->d_frames[0] + b->frame_off[u0] * W0, W0, nullptr, nfr, h->d_lambda, l,
                                                                     spec_samples(W0), 5 * l.L, cb.P, f32));
    PhaseTimer tk(h, PH_K_SCORE, cb.st);
    KT_RUN("k_scores_mfma(state)", cb.st, launch_scores_mfma(cb.st, cb.X, hybrid_row_floats(l, W0), nullptr, nseg, h->d_lambda, l, spec_dense_x(l, W0), l.L, cb.S, f32));
    // + the labelled windows' scores, the row maxima and exp(S - smax) for the linear-domain recursion (cb.lin)
    KT_RUN("k_add_p_exp", cb.st, launch_add_p_exp(cb.st, l, bv, b->d_frame_u, u0, nfr, cb.P, cb.S, cb.smax, cb.s_true));
    cb.es_ready = true;
    tk.stop(2);
    nl += 2;
  } else if (!cb.fused) {
    PhaseTimer tk(h, PH_K_SCORE, cb.st);
    if (fast) KT_RUN("k_scores_mfma(state)", cb.st, launch_scores_mfma(cb.st, cb.X, l.F, nullptr, nseg, h->d_lambda, l, scrf_spec_state(l), l.L, cb.S, f32));
    else KT_RUN("k_scores_exact(state)", cb.st, launch_scores_exact(cb.st, cb.X, l.F, nullptr, nseg, h->d_lambda, l, 0, l.L, cb.S));
    tk.stop(1);
  }
  if (segtrans(h)) {
    // one transition matrix per window: the same contraction over every row of X; the rows of the
    // utterance-initial segments (no predecessor) are zeroed like the reference leaves them unused
    if (fast) KT_RUN("k_scores_mfma(trans)", cb.st, launch_scores_mfma(cb.st, cb.X, l.F, nullptr, nseg, h->d_lambda, l, scrf_spec_trans(l), l.L * l.L, cb.M, f32));
    else KT_RUN("k_scores_exact(trans)", cb.st, launch_scores_exact(cb.st, cb.X, l.F, nullptr, nseg, h->d_lambda, l, 1, l.L * l.L, cb.M));
    launch_zero_initial_rows(cb.st, bv, b->d_frame_u, u0, nfr, l.D, l.L, cb.M);
    nl += 2;
  } else if (l.use_tf) {
    launch_frame_rows(cb.st, bv, u0, u1, l.D, nfr, cb.xrow_cur, 0);
    if (fast) KT_RUN("k_scores_mfma(trans)", cb.st, launch_scores_mfma(cb.st, cb.X, l.F, cb.xrow_cur, nfr, h->d_lambda, l, scrf_spec_trans(l), l.L * l.L, cb.M, f32));
    else KT_RUN("k_scores_exact(trans)", cb.st, launch_scores_exact(cb.st, cb.X, l.F, cb.xrow_cur, nfr, h->d_lambda, l, 1, l.L * l.L, cb.M));
    nl += 2;
  } else if (!h->m0_valid && !segtrans(h)) {
    // transition scores carry only the bias: one L x L matrix for every frame
    launch_scores_exact(cb.st, cb.X, l.F, nullptr, 1, h->d_lambda, l, 1, l.L * l.L, h->d_m0);
    launch_exp_m(cb.st, h->d_m0, l.L, 1, h->d_e0, h->d_et0, h->d_msh0);
    h->m0_valid = true;
    nl += 2;
  }
  tm.stop(nl);
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

// forward + backward (+ posteriors when `post`): wavefront-per-utterance kernels for L <= 64,
// the workgroup-per-utterance kernel otherwise.  Leaves R = Y - gamma in cb.AD (post) or
// alpha-with-duration (no post), alpha in cb.alpha, beta in cb.beta (wave path or nd.beta).
static int run_dp(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, ChunkBufs& cb, bool post, uint32_t* nl_out) {
  const ScrfLayout& l = h->lay;
  const uint64_t nutt = u1 - u0, nfr = b->frame_off[u1] - b->frame_off[u0];
  // (only the posterior-mass self-checks look at it in the recursion kernels)
  const int frame_model = h->cfg.model_type == SCRF_STDFRAME || h->frame_mass;
  ScrfBatchView bv = b->view();
  uint32_t nl = 0;
  if (segtrans(h)) {
    if (fb_segtrans_smem_bytes(l, fb_block_threads(l)) > 160 * 1024)
      return fail(h, SCRF_ERR_INVALID, "labels x maximum duration = %u x %u is too large for the stdseg_no_dur recursion "
                  "(its three [D][L] rings must fit 160 KB of LDS)", l.L, l.D);
    KT_RUN("k_fb_segtrans", cb.st, launch_fb_segtrans(cb.st, l, bv, u0, (uint32_t)nutt, b->d_prev_lab, cb.S, cb.M, cb.AD, cb.alpha, cb.beta,
                       post ? cb.XI : nullptr, b->d_numer, b->d_zx, b->d_status, post ? 1 : 0));
    nl = 1;
  } else if (!cb.wave) {
    if (fb_smem_bytes(l, fb_block_threads(l)) > 160 * 1024)
      return fail(h, SCRF_ERR_INVALID, "labels x maximum duration = %u x %u is too large for the workgroup-per-utterance recursion "
                  "(its two [D][L] rings must fit 160 KB of LDS; the wavefront kernels cover L <= 256 with D <= 40)", l.L, l.D);
    if (post && cb.xi_acc) HIPCHK(h, hipMemsetAsync(cb.xi_acc, 0, sizeof(double) * nutt * l.L * l.L, cb.st));
    KT_RUN("k_fb", cb.st, launch_fb(cb.st, l, bv, u0, (uint32_t)nutt, cb.S, cb.M, cb.m_per_frame, cb.AD, cb.alpha, cb.beta, post ? cb.XI : nullptr,
              post ? cb.xi_acc : nullptr, b->d_numer, b->d_zx, b->d_status, post ? 1 : 0, frame_model));
    nl = 1;
  } else {
    if (cb.m_per_frame) { KT_RUN("k_exp_m", cb.st, launch_exp_m(cb.st, cb.M, l.L, nfr, cb.E, cb.ET, cb.msh)); nl++; }
    if (cb.lin) {
      const uint64_t nseg = b->seg_off[u1] - b->seg_off[u0];
      if (!cb.es_ready) {
        KT_RUN("k_true_scores", cb.st, launch_true_scores(cb.st, l, bv, b->d_frame_u, u0, nfr, cb.S, cb.s_true));
        KT_RUN("k_exp_rows", cb.st, launch_exp_rows(cb.st, cb.S, nseg, l.L, cb.smax));
        nl += 2;
      }
      {
        PhaseTimer tk(h, PH_K_DP, cb.st);
        KT_RUN(l.L > 64 ? "k_dp_lin_mw" : "k_dp_lin", cb.st, launch_dp_lin(cb.st, l, bv, u0, (uint32_t)nutt, cb.S, cb.smax, cb.E, cb.ET, cb.msh, cb.m_per_frame, cb.dl,
                      b->d_zx, b->d_status));
        tk.stop(1);
      }
      if (cb.fused) {
        // posterior pass and the per-frame sums of R in one walk (R is not read back for Z)
        if (l.L > 64) HIPCHK(h, hipMemsetAsync(cb.mass_s, 0, sizeof(double) * nfr, cb.st));   // summed over the 64-output groups
        uint32_t t_max = 0;   // the longest utterance of the chunk: a launch of few utterances splits each into segments
        for (uint64_t u = u0; u < u1; u++) t_max = std::max(t_max, b->T[u]);
        KT_RUN("k_post_z", cb.st, launch_post_z(cb.st, l, bv, u0, (uint32_t)nutt, b->d_next_lab, cb.s_true, cb.M, cb.m_per_frame, cb.S, cb.smax,
                      cb.dl, b->d_zx, cb.numer_f, b->d_status, cb.Z, cb.mass_s, cb.la ? 1 : 0, t_max, nfr));
        cb.z_ready = true;
      } else {
        KT_RUN("k_post_lin", cb.st, launch_post_lin(cb.st, l, bv, b->d_frame_u, u0, nfr, b->d_next_lab, cb.s_true, cb.M, cb.m_per_frame, cb.S,
                        cb.smax, cb.dl, b->d_zx, cb.numer_f, b->d_status, cb.mass_s));
      }
      // the reference's posterior-mass self-checks (computeExpF :917-947), before sd / gsd are rescaled below
      KT_RUN("k_mass_check", cb.st, launch_mass_check(cb.st, bv, b->d_frame_u, u0, nfr, l.L, frame_model, 1, cb.dl.a, cb.dl.ga, cb.dl.b, cb.dl.gb,
                        b->d_zx, cb.mass_s, b->d_status));
      KT_RUN("k_numer_reduce", cb.st, launch_numer_reduce(cb.st, bv, u0, (uint32_t)nutt, cb.numer_f, b->d_numer));
      // transition posteriors: with transition features the rescaled sd rows are needed as an array; with
      // bias-only transitions only their per-frame factor is, applied inside the A^T B contraction
      if (l.use_tf) KT_RUN("k_xi_lin", cb.st, launch_xi_lin(cb.st, l, bv, b->d_frame_u, u0, nfr, cb.dl, b->d_zx));
      else KT_RUN("k_xi_scale", cb.st, launch_xi_scale(cb.st, bv, b->d_frame_u, u0, nfr, cb.dl, b->d_zx));
      nl += 5;
      if (l.use_tf) {
        KT_RUN("k_xi_full", cb.st, launch_xi_full(cb.st, l, bv, u0, u1, nfr, b->d_next_lab, cb.fA, cb.fB, cb.E, cb.msh, cb.XI));
        nl++;
      }
      if (nl_out) *nl_out = nl;
      HIPCHK(h, hipGetLastError());
      return SCRF_OK;
    }
    {
      PhaseTimer tk(h, PH_K_DP, cb.st);
      KT_RUN("k_dp_wave", cb.st, launch_dp_wave(cb.st, l, bv, u0, (uint32_t)nutt, cb.S, cb.E, cb.ET, cb.msh, cb.m_per_frame, cb.AD, cb.alpha,
                     cb.beta, cb.sd, b->d_zx, b->d_status));
      tk.stop(1);
    }
    nl++;
    if (post) {
      KT_RUN("k_post_state", cb.st, launch_post_state(cb.st, l, bv, u0, u1, nfr, b->d_next_lab, cb.S, cb.M, cb.m_per_frame, cb.AD, cb.beta,
                        b->d_zx, cb.numer_f, b->d_status, cb.mass_s));
      KT_RUN("k_mass_check", cb.st, launch_mass_check(cb.st, bv, b->d_frame_u, u0, nfr, l.L, frame_model, 0, cb.alpha, nullptr, cb.beta, nullptr,
                        b->d_zx, cb.mass_s, b->d_status));
      KT_RUN("k_numer_reduce", cb.st, launch_numer_reduce(cb.st, bv, u0, (uint32_t)nutt, cb.numer_f, b->d_numer));
      KT_RUN("k_xi_factors", cb.st, launch_xi_factors(cb.st, l, bv, u0, u1, nfr, cb.alpha, cb.sd, b->d_zx, cb.fA, cb.fB));
      nl += 4;
      if (l.use_tf) {
        KT_RUN("k_xi_full", cb.st, launch_xi_full(cb.st, l, bv, u0, u1, nfr, b->d_next_lab, cb.fA, cb.fB, cb.E, cb.msh, cb.XI));
        nl++;
      }
    }
  }
  if (nl_out) *nl_out = nl;
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// status of a batch: first failed utterance -> {code, utterance} (d_latch), copied to pinned host memory;
// the commit of the staged gradient is a no-op when the latch is set
// ---------------------------------------------------------------------------------------------
// latch[0]: non-zero iff an utterance failed (what k_commit tests); latch[1]: min over the failed utterances of
// (utterance << 8 | code), so the LOWEST failing utterance is the one reported, whichever thread gets there first --
// the order in which the reference's per-utterance loop would have thrown.  fb_latch_reset arms it.
#define SCRF_LATCH_IDLE 0x7fffffff
__global__ void k_latch_status(const int* __restrict__ status, uint32_t n, int* __restrict__ latch) {
  const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= n) return;
  const int s = status[u];
  if (s != 0) {
    atomicMin(&latch[1], (int)((u << 8) | ((uint32_t)s & 0xffu)));
    atomicOr(&latch[0], 1);
  }
}
__global__ void k_latch_reset(int* __restrict__ latch) {
  latch[0] = 0;
  latch[1] = SCRF_LATCH_IDLE;
}
// {code, utterance} from the pinned copy of the latch
static void latch_decode(const int* h_latch, int out[2]) {
  out[0] = out[1] = 0;
  if (h_latch[0] != 0 && h_latch[1] != SCRF_LATCH_IDLE) {
    out[0] = h_latch[1] & 0xff;
    out[1] = (int)((uint32_t)h_latch[1] >> 8);
  }
}
__global__ void k_commit(double* __restrict__ grad, const double* __restrict__ stage, uint32_t n,
                         double* __restrict__ sums, const double* __restrict__ sums_stage,
                         const int* __restrict__ latch) {
  if (latch[0] != 0) return;   // a failed batch contributes nothing
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) grad[i] += stage[i];
  if (i < 3) sums[i] += sums_stage[i];
}

// the same for one block of the weight vector: part 1 = the transition weights of every label ([nsf, stride) of its
// block), part 0 = the state weights ([0, nsf)) and the batch sums
__global__ void k_commit_part(double* __restrict__ grad, const double* __restrict__ stage, uint32_t n, uint32_t stride, uint32_t nsf,
                              int part, double* __restrict__ sums, const double* __restrict__ sums_stage,
                              const int* __restrict__ latch) {
  if (latch[0] != 0) return;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && ((i % stride >= nsf) ? 1 : 0) == part) grad[i] += stage[i];
  if (part == 0 && i < 3) sums[i] += sums_stage[i];
}

static const char* status_text(int code) {
  return code == SCRF_ERR_BAD_LABEL ? "the label is larger than nActualLabs*labMaxDur"
         : code == SCRF_ERR_EMPTY   ? "No features read from this sentence."
                                    : "overflow / NaN / log of zero in the recursion, or posterior-mass check failed";
}

static int queue_status(scrf_handle h, scrf_batch b) {
  hipLaunchKernelGGL(k_latch_status, dim3((b->U + 255) / 256), dim3(256), 0, h->stream, b->d_status, b->U, h->d_latch);
  HIPCHK(h, hipMemcpyAsync(h->h_latch, h->d_latch, sizeof(int) * 2, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipEventRecord(h->ev_status, h->stream));
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// STDSEG (duration-labelled, scrf_stdseg.hip): its own small pipeline -- windows, scores, workgroup recursion,
// posteriors, per-weight gradient -- in chunks of utterances that fit the scratch budget.  h->lay is the layout over
// the FULL labels (lay.L = nLabs); La = nActualLabs.
// ---------------------------------------------------------------------------------------------
static bool stdseg(scrf_handle h) { return h->cfg.model_type == SCRF_STDSEG; }
static uint32_t stdseg_La(scrf_handle h) { return h->lay.L / h->lay.D; }

struct StdsegBufs {
  float* X; uint32_t *row_t, *row_d, *row_u;
  double *S, *MX, *alpha, *beta, *G, *XI, *mass_s, *mass_t;
};
static size_t stdseg_chunk_bytes(scrf_handle h, scrf_batch b, uint64_t nfr, uint64_t nseg, bool post) {
  const ScrfLayout& l = h->lay;
  const uint32_t La = stdseg_La(h);
  size_t tot = 0;
  if (b->mode == 1) tot += pad256(nseg * l.F * sizeof(float));
  tot += 3 * pad256(nseg * sizeof(uint32_t));
  tot += 3 * pad256(nseg * La * sizeof(double)) + pad256(nseg * (size_t)l.L * La * sizeof(double));   // S, alpha, beta, MX
  if (post) tot += pad256(nseg * La * sizeof(double)) + pad256(nseg * (size_t)l.L * La * sizeof(double)) + 2 * pad256(nfr * sizeof(double));
  return tot;
}
// ---- STDSEG, bias-only transitions, FAST precisions, training path: scrf_stdseg_lin.hip
static bool stdseg_lin(scrf_handle h) {
  static const bool on = !(getenv("SCRF_STDSEG_LIN") && atoi(getenv("SCRF_STDSEG_LIN")) == 0);
  return on && h->cfg.train_precision != SCRF_PREC_EXACT && h->lay.use_sf && stdseg_lin_supported(h->lay, stdseg_La(h));
}
static uint64_t sl_rows_per_chunk(uint64_t nfr) {   // K-chunks of the count contractions: <= 256 of them, multiples of 32 frames
  uint64_t rpc = ((nfr + 255) / 256 + 31) & ~31ull;
  return rpc < 64 ? 64 : rpc;
}
static size_t stdseg_lin_chunk_bytes(scrf_handle h, scrf_batch b, uint64_t nfr, uint64_t nseg) {
  const ScrfLayout& l = h->lay;
  const uint32_t La = stdseg_La(h);
  const size_t node = pad256((size_t)l.D * nfr * La * sizeof(double));
  const uint64_t nch = (nfr + sl_rows_per_chunk(nfr) - 1) / sl_rows_per_chunk(nfr);
  size_t tot = 0;
  if (b->mode == 1) tot += pad256(nseg * l.F * sizeof(float));
  tot += pad256((size_t)l.D * nfr * sizeof(uint64_t));                // xrow
  tot += 5 * node;                                                    // Sd, Ad, Bd, Rd, Bp
  tot += pad256(nfr * (size_t)l.L * sizeof(double)) + 2 * pad256(nfr * sizeof(double));   // Am, ga, numer_f
  tot += pad256(nch * (size_t)La * l.nsf * sizeof(double));           // state-count slabs (one duration at a time)
  tot += pad256(nch * (size_t)l.L * l.L * sizeof(double)) + pad256((size_t)l.L * l.L * sizeof(double));   // transition slabs, observed
  return tot + 4096;
}
static int stdseg_lin_run_chunk(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, double* grad) {
  const ScrfLayout& l = h->lay;
  const uint32_t La = stdseg_La(h);
  const uint64_t nfr = b->frame_off[u1] - b->frame_off[u0], nseg = b->seg_off[u1] - b->seg_off[u0];
  int rc = ensure_scratch(h, stdseg_lin_chunk_bytes(h, b, nfr, nseg));
  if (rc != SCRF_OK) return rc;
  Arena a{h->scratch, h->scratch_cap, 0};
  ScrfBatchView bv = b->view();
  hipStream_t st = h->stream;
  const int f32 = h->cfg.train_precision == SCRF_PREC_FAST32;
  float* X;
  if (b->mode == 1) {
    X = a.take<float>(nseg * l.F);
    uint32_t col = 0;
    for (uint32_t s = 0; s < b->n_streams; s++) {
      const scrf_stream_recipe& r = b->recipe[s];
      KT_RUN("k_windows", st, launch_windows(st, b->d_frames[s], b->d_sframe_off[s], bv, u0, u1, nfr, r.in_width, l.D, r.left_ctx, r.right_ctx,
                                             r.extract_seg_ftr, X, l.F, col));
      col += b->width[s];
    }
  } else X = b->d_windows + b->seg_off[u0] * l.F;
  const size_t nn = (size_t)l.D * nfr * La;
  uint64_t* xrow = a.take<uint64_t>((size_t)l.D * nfr);
  double* Sd = a.take<double>(nn); double* Ad = a.take<double>(nn); double* Bd = a.take<double>(nn);
  double* Rd = a.take<double>(nn); double* Bp = a.take<double>(nn);
  double* Am = a.take<double>(nfr * (size_t)l.L);
  double* ga = a.take<double>(nfr); double* numer_f = a.take<double>(nfr);
  const uint64_t rpc = sl_rows_per_chunk(nfr);
  const uint32_t nch = (uint32_t)((nfr + rpc - 1) / rpc);
  double* slab_s = a.take<double>((size_t)nch * La * l.nsf);
  double* slab_t = a.take<double>((size_t)nch * l.L * l.L);
  double* obs = a.take<double>((size_t)l.L * l.L);
  if (a.off > a.cap) return fail(h, SCRF_ERR_INVALID, "internal: scratch arena overflow");
  // the transition table and its exponential (once per call: lambda may have changed)
  double* E = h->d_sl_tab; double* ET = E + (size_t)l.L * l.L; double* mmax = ET + (size_t)l.L * l.L;
  launch_sl_tables(st, l, h->d_lambda, E, ET, mmax);
  launch_sl_rows(st, bv, b->d_frame_u, u0, nfr, l.D, xrow);
  {
    KernelTimer kt(h, "k_scores_mfma(state, per duration)", st);
    for (uint32_t d0 = 0; d0 < l.D; d0++) {
      const ScrfGemmSpec sp{0, l.sfs, l.nsfe, (uint32_t)l.use_sb, l.sbv, d0 * La * l.stride, 0};
      launch_scores_mfma(st, X, l.F, xrow + (size_t)d0 * nfr, nfr, h->d_lambda, l, sp, La, Sd + (size_t)d0 * nfr * La, f32);
    }
    kt.stop(l.D);
  }
  KT_RUN("k_sl_fb", st, launch_sl_fb(st, l, La, bv, u0, u1 - u0, nfr, Sd, E, ET, mmax, Ad, Bd, Am, ga, b->d_zx, b->d_status));
  KT_RUN("k_sl_post", st, launch_sl_post(st, l, La, bv, b->d_frame_u, u0, u1 - u0, nfr, b->d_prev_lab, h->d_lambda, Sd, Ad, Bd, ga, b->d_zx, Rd, Bp,
                                         numer_f, b->d_numer, b->d_status));
  {
    KernelTimer kt(h, "k_expf_mfma(state, per duration)", st);
    for (uint32_t d0 = 0; d0 < l.D; d0++) {
      const ScrfGemmSpec sp{0, l.sfs, l.nsfe, (uint32_t)l.use_sb, l.sbv, d0 * La * l.stride, 0};
      launch_expf_mfma(st, Rd + (size_t)d0 * nfr * La, La, X, l.F, xrow + (size_t)d0 * nfr, nfr, l, sp, rpc, nch, slab_s, f32);
      launch_reduce_slabs(st, slab_s, nch, La, l, sp, grad);
    }
    kt.stop(2 * l.D);
  }
  HIPCHK(h, hipMemsetAsync(obs, 0, sizeof(double) * l.L * l.L, st));
  KT_RUN("k_sl_atb", st, launch_sl_trans_counts(st, l, La, bv, b->d_frame_u, u0, nfr, b->d_prev_lab, rpc, nch, Am, Bp, E, mmax, slab_t, obs, grad));
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

// chunk [u0, u1): scores and recursion (and, with post, posteriors, numerators, the gradient into `grad`)
static int stdseg_run_chunk(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, bool post, double* grad, StdsegBufs* out) {
  if (post && !out && stdseg_lin(h)) return stdseg_lin_run_chunk(h, b, u0, u1, grad);
  const ScrfLayout& l = h->lay;
  const uint32_t La = stdseg_La(h);
  const uint64_t nfr = b->frame_off[u1] - b->frame_off[u0], nseg = b->seg_off[u1] - b->seg_off[u0];
  const size_t need = stdseg_chunk_bytes(h, b, nfr, nseg, post);
  int rc = ensure_scratch(h, need);
  if (rc != SCRF_OK) return rc;
  Arena a{h->scratch, h->scratch_cap, 0};
  StdsegBufs sb;
  memset(&sb, 0, sizeof(sb));
  ScrfBatchView bv = b->view();
  hipStream_t st = h->stream;
  if (b->mode == 1) {
    sb.X = a.take<float>(nseg * l.F);
    uint32_t col = 0;
    for (uint32_t s = 0; s < b->n_streams; s++) {
      const scrf_stream_recipe& r = b->recipe[s];
      launch_windows(st, b->d_frames[s], b->d_sframe_off[s], bv, u0, u1, nfr, r.in_width, l.D, r.left_ctx, r.right_ctx, r.extract_seg_ftr,
                     sb.X, l.F, col);
      col += b->width[s];
    }
  } else {
    sb.X = b->d_windows + b->seg_off[u0] * l.F;
  }
  sb.row_t = a.take<uint32_t>(nseg); sb.row_d = a.take<uint32_t>(nseg); sb.row_u = a.take<uint32_t>(nseg);
  sb.S = a.take<double>(nseg * La); sb.alpha = a.take<double>(nseg * La); sb.beta = a.take<double>(nseg * La);
  sb.MX = a.take<double>(nseg * (size_t)l.L * La);
  launch_stdseg_rowinfo(st, bv, b->d_frame_u, u0, nfr, l.D, sb.row_t, sb.row_d, sb.row_u);
  KT_RUN("k_stdseg_scores", st, launch_stdseg_scores(st, l, La, sb.X, nseg, sb.row_t, sb.row_d, h->d_lambda, sb.S, sb.MX));
  KT_RUN("k_stdseg_fb", st, launch_stdseg_fb(st, l, La, bv, u0, u1 - u0, sb.S, sb.MX, sb.alpha, sb.beta, b->d_zx, b->d_status));
  if (post) {
    sb.G = a.take<double>(nseg * La); sb.XI = a.take<double>(nseg * (size_t)l.L * La);
    sb.mass_s = a.take<double>(nfr); sb.mass_t = a.take<double>(nfr);
    HIPCHK(h, hipMemsetAsync(sb.mass_s, 0, sizeof(double) * nfr, st));
    HIPCHK(h, hipMemsetAsync(sb.mass_t, 0, sizeof(double) * nfr, st));
    KT_RUN("k_stdseg_post", st, launch_stdseg_post(st, l, La, bv, u0, u1 - u0, nseg, sb.row_t, sb.row_d, sb.row_u, b->d_prev_lab, sb.S, sb.MX, sb.alpha, sb.beta,
                                                   b->d_zx, sb.G, sb.XI, sb.mass_s, sb.mass_t, b->d_numer, b->d_status));
    KT_RUN("k_stdseg_expf", st, launch_stdseg_expf(st, l, La, bv, u0, u1 - u0, nseg, sb.row_t, sb.row_d, sb.row_u, b->d_prev_lab, sb.X, sb.G, sb.XI, grad));
  }
  HIPCHK(h, hipGetLastError());
  if (out) *out = sb;
  return SCRF_OK;
}
static uint32_t stdseg_plan_chunk(scrf_handle h, scrf_batch b, uint32_t u0, bool post) {
  uint32_t u1 = u0 + 1;
  if (post && stdseg_lin(h)) {
    while (u1 < b->U && u1 - u0 < 32767 &&
           stdseg_lin_chunk_bytes(h, b, b->frame_off[u1 + 1] - b->frame_off[u0], b->seg_off[u1 + 1] - b->seg_off[u0]) <= h->cfg.scratch_bytes) u1++;
    return u1;
  }
  while (u1 < b->U && stdseg_chunk_bytes(h, b, b->frame_off[u1 + 1] - b->frame_off[u0], b->seg_off[u1 + 1] - b->seg_off[u0], post) <= h->cfg.scratch_bytes) u1++;
  return u1;
}

// ---------------------------------------------------------------------------------------------
// n-state frame model (crf_states > 1, scrf_nstate.hip): the same small pipeline shape as STDSEG
// ---------------------------------------------------------------------------------------------
static bool nstate(scrf_handle h) { return h->lay.K > 1; }
struct NstateBufs {
  float* X;
  double *S, *TD, *TO, *TE, *alpha, *beta, *G, *XD, *XO, *XE, *mass_s, *mass_t;
};
static size_t nstate_chunk_bytes(scrf_handle h, scrf_batch b, uint64_t nfr, bool post) {
  const ScrfLayout& l = h->lay;
  const uint64_t P = l.L / l.K;
  size_t tot = 0;
  if (b->mode == 1) tot += pad256(nfr * l.F * sizeof(float));
  tot += 5 * pad256(nfr * l.L * sizeof(double)) + pad256(nfr * P * P * sizeof(double));              // S, TD, TO, alpha, beta, TE
  if (post) tot += 3 * pad256(nfr * l.L * sizeof(double)) + pad256(nfr * P * P * sizeof(double)) + 2 * pad256(nfr * sizeof(double)) +
                   pad256((size_t)ns_expf_slices(nfr) * l.lambda_len * sizeof(double));   // + the gradient's frame-slice partials
  return tot;
}
static int nstate_run_chunk(scrf_handle h, scrf_batch b, uint32_t u0, uint32_t u1, bool post, double* grad, NstateBufs* out) {
  const ScrfLayout& l = h->lay;
  const uint64_t P = l.L / l.K;
  const uint64_t nfr = b->frame_off[u1] - b->frame_off[u0];
  int rc = ensure_scratch(h, nstate_chunk_bytes(h, b, nfr, post));
  if (rc != SCRF_OK) return rc;
  Arena a{h->scratch, h->scratch_cap, 0};
  NstateBufs nb;
  memset(&nb, 0, sizeof(nb));
  ScrfBatchView bv = b->view();
  hipStream_t st = h->stream;
  if (b->mode == 1) {
    nb.X = a.take<float>(nfr * l.F);
    uint32_t col = 0;
    for (uint32_t s = 0; s < b->n_streams; s++) {
      const scrf_stream_recipe& r = b->recipe[s];
      launch_windows(st, b->d_frames[s], b->d_sframe_off[s], bv, u0, u1, nfr, r.in_width, l.D, r.left_ctx, r.right_ctx, r.extract_seg_ftr,
                     nb.X, l.F, col);
      col += b->width[s];
    }
  } else {
    nb.X = b->d_windows + b->seg_off[u0] * l.F;
  }
  nb.S = a.take<double>(nfr * l.L); nb.TD = a.take<double>(nfr * l.L); nb.TO = a.take<double>(nfr * l.L);
  nb.alpha = a.take<double>(nfr * l.L); nb.beta = a.take<double>(nfr * l.L);
  nb.TE = a.take<double>(nfr * P * P);
  KT_RUN("k_ns_scores", st, launch_ns_scores(st, l, nb.X, nfr, h->d_lambda, nb.S, nb.TD, nb.TO, nb.TE));
  KT_RUN("k_ns_fb", st, launch_ns_fb(st, l, bv, u0, u1 - u0, nb.S, nb.TD, nb.TO, nb.TE, nb.alpha, nb.beta, b->d_zx, b->d_status));
  if (post) {
    nb.G = a.take<double>(nfr * l.L); nb.XD = a.take<double>(nfr * l.L); nb.XO = a.take<double>(nfr * l.L);
    nb.XE = a.take<double>(nfr * P * P);
    nb.mass_s = a.take<double>(nfr); nb.mass_t = a.take<double>(nfr);
    HIPCHK(h, hipMemsetAsync(nb.mass_s, 0, sizeof(double) * nfr, st));
    HIPCHK(h, hipMemsetAsync(nb.mass_t, 0, sizeof(double) * nfr, st));
    KT_RUN("k_ns_post", st, launch_ns_post(st, l, bv, b->d_frame_u, u0, u1 - u0, nfr, nb.S, nb.TD, nb.TO, nb.TE, nb.alpha, nb.beta, b->d_zx, nb.G, nb.XD, nb.XO,
                                           nb.XE, nb.mass_s, nb.mass_t, b->d_numer, b->d_status));
    double* slab = a.take<double>((size_t)ns_expf_slices(nfr) * l.lambda_len);
    KT_RUN("k_ns_expf", st, launch_ns_expf(st, l, bv, b->d_frame_u, u0, nfr, nb.X, nb.G, nb.XD, nb.XO, nb.XE, slab, grad));
  }
  HIPCHK(h, hipGetLastError());
  if (out) *out = nb;
  return SCRF_OK;
}
static uint32_t nstate_plan_chunk(scrf_handle h, scrf_batch b, uint32_t u0, bool post) {
  uint32_t u1 = u0 + 1;
  while (u1 < b->U && nstate_chunk_bytes(h, b, b->frame_off[u1 + 1] - b->frame_off[u0], post) <= h->cfg.scratch_bytes) u1++;
  return u1;
}

// One pass of the forward-backward pipeline over the batch into the staging gradient.  latch[2] receives
// {status code, utterance} of the first failed utterance (0 = clean, gradient committed); *used_lin tells
// whether any chunk ran the linear-domain recursion.
static int allreduce_block(scrf_handle h, hipStream_t st, int part);
static bool two_block_reduce(scrf_handle h) { return h->comm && h->lay.use_tf && h->cfg.model_type != SCRF_STDSEG_NO_DUR && h->cfg.model_type != SCRF_STDSEG && h->lay.K <= 1 && !h->shadow; }

static int fb_run(scrf_handle h, scrf_batch b, int latch[2], bool* used_lin) {
  const ScrfLayout& l = h->lay;
  HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
  hipLaunchKernelGGL(k_latch_reset, dim3(1), dim3(1), 0, h->stream, h->d_latch);
  HIPCHK(h, hipMemsetAsync(h->d_stage, 0, sizeof(double) * l.lambda_len, h->stream));
  HIPCHK(h, hipMemsetAsync(h->d_sums_stage, 0, sizeof(double) * 4, h->stream));
  *used_lin = false;
  if (stdseg(h) || nstate(h)) {
    for (uint32_t u0 = 0; u0 < b->U;) {
      const uint32_t u1 = nstate(h) ? nstate_plan_chunk(h, b, u0, true) : stdseg_plan_chunk(h, b, u0, true);
      int rc = nstate(h) ? nstate_run_chunk(h, b, u0, u1, true, h->d_stage, nullptr) : stdseg_run_chunk(h, b, u0, u1, true, h->d_stage, nullptr);
      if (rc != SCRF_OK) return rc;
      launch_stdseg_sums(h->stream, b->d_numer, b->d_zx, u0, u1 - u0, h->d_sums_stage);
      u0 = u1;
    }
    int rc = queue_status(h, b);
    if (rc != SCRF_OK) return rc;
    hipLaunchKernelGGL(k_commit, dim3((l.lambda_len + 255) / 256), dim3(256), 0, h->stream, h->d_grad, h->d_stage, l.lambda_len,
                       h->d_sums, h->d_sums_stage, h->d_latch);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventSynchronize(h->ev_status));
    latch_decode(h->h_latch, latch);
    return SCRF_OK;
  }
  Need nd{true, true, false, false};
  ScrfBatchView bv = b->view();
  const bool fast = h->cfg.train_precision >= SCRF_PREC_FAST;
  const int f32 = h->cfg.train_precision == SCRF_PREC_FAST32;
  nd.fused = fast && b->fused_ok && h->fuse_windows;
  // the linear window average needs the fused kernels, the linear-domain recursion (k_post_z builds Z_avg) and a
  // shape its kernels take; anything else runs as FAST
  nd.hybrid = fast && !nd.fused && b->hybrid_ok && h->hybrid && h->fuse_windows && h->lin_dp && !h->force_fb && wave_path(h, true) && !f32;
  nd.la = nd.fused && h->cfg.train_precision == SCRF_PREC_FASTLIN && h->lin_dp && !h->force_fb && wave_path(h, true) &&
          fused_la_supported(l, b->recipe[0].in_width);

  // plan the chunks first: each must fit the scratch budget; with two lanes a batch is cut into
  // at least four chunks so that both streams always have work
  std::vector<uint32_t> cuts(1, 0);
  const bool two_lanes = h->n_lanes > 1 && !h->timing && b->U >= 64;
  {
    const uint32_t cap = two_lanes ? (b->U + 3) / 4 : b->U;
    for (uint32_t u0 = 0; u0 < b->U;) {
      uint32_t u1 = plan_chunk(h, b, u0, nd);
      if (u1 - u0 > cap) u1 = u0 + cap;
      cuts.push_back(u1);
      u0 = u1;
    }
  }
  const size_t n_chunks = cuts.size() - 1;
  const bool use2 = two_lanes && n_chunks >= 2;
  if (!h->m0_valid && !l.use_tf && !segtrans(h)) {
    // transition scores carry only the bias: one L x L matrix (and its exp) for every frame;
    // computed before the lanes fork
    launch_scores_exact(h->stream, nullptr, l.F, nullptr, 1, h->d_lambda, l, 1, l.L * l.L, h->d_m0);
    launch_exp_m(h->stream, h->d_m0, l.L, 1, h->d_e0, h->d_et0, h->d_msh0);
    h->m0_valid = true;
  }
  if (use2) {
    HIPCHK(h, hipMemsetAsync(h->d_grad2, 0, sizeof(double) * l.lambda_len, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_sums2, 0, sizeof(double) * 4, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
    HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
  }
  for (size_t ci = 0; ci < n_chunks; ci++) {
    const uint32_t u0 = cuts[ci], u1 = cuts[ci + 1];
    const int lane = use2 ? (int)(ci & 1) : 0;
    ChunkBufs cb;
    int rc = carve(h, b, u0, u1, nd, &cb, lane);
    if (rc != SCRF_OK) return rc;
    if (!lane) { cb.grad = h->d_stage; cb.sums = h->d_sums_stage; }
    if (cb.lin) *used_lin = true;
    const uint64_t nutt = u1 - u0, nfr = b->frame_off[u1] - b->frame_off[u0], nseg = b->seg_off[u1] - b->seg_off[u0];
    rc = run_scores(h, b, u0, u1, cb, fast, f32);
    if (rc != SCRF_OK) return rc;
    {
      PhaseTimer tm(h, PH_FB, cb.st);
      uint32_t nl = 0;
      rc = run_dp(h, b, u0, u1, cb, true, &nl);
      if (rc != SCRF_OK) return rc;
      tm.stop(nl);
    }
    // every per-utterance status of the batch is final once the last chunk's posterior kernels are queued:
    // its copy to the host travels under the expected-count kernels
    if (!use2 && ci + 1 == n_chunks) {
      rc = queue_status(h, b);
      if (rc != SCRF_OK) return rc;
    }
    // all-reduce / compute overlap (fused call only): transition counts first; once the last chunk's are reduced they
    // are committed and their all-reduce goes to the second stream, under the state contraction.  A NUMERIC failure of
    // the recursion is retried with the log-domain kernels by the caller: that has to be known BEFORE a collective is
    // issued (the peers issue theirs exactly once), so the host waits for the status here.
    const bool ov = h->overlap_comm && h->comm_overlap_on && two_block_reduce(h) && !use2 && !h->timing;
    bool trans_done = false;   // the transition counts of this chunk are already in the staged gradient
    if (ov) {
      if (ci + 1 == n_chunks) {
        HIPCHK(h, hipEventSynchronize(h->ev_status));
        latch_decode(h->h_latch, latch);
        if (latch[0] == SCRF_ERR_NUMERIC && wave_path(h, true)) return SCRF_OK;   // the caller reruns; nothing committed, nothing sent
      }
      launch_frame_rows(cb.st, bv, u0, u1, l.D, nfr, cb.xrow_next, 1);
      if (fast) launch_expf_mfma(cb.st, cb.XI, l.L * l.L, cb.X, l.F, cb.xrow_next, nfr, l, scrf_spec_trans(l), cb.rpc_t, cb.nch_t, cb.slab_t, f32);
      else launch_expf_gemm(cb.st, cb.XI, l.L * l.L, cb.X, l.F, cb.xrow_next, nfr, l, 1, cb.rpc_t, cb.nch_t, cb.slab_t);
      launch_reduce_slabs(cb.st, cb.slab_t, cb.nch_t, l.L * l.L, l, scrf_spec_trans(l), cb.grad);
      if (ci + 1 == n_chunks) {
        hipLaunchKernelGGL(k_commit_part, dim3((l.lambda_len + 255) / 256), dim3(256), 0, h->stream, h->d_grad, h->d_stage, l.lambda_len,
                           l.stride, l.nsf, 1, h->d_sums, h->d_sums_stage, h->d_latch);
        HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));
        HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
        rc = allreduce_block(h, h->stream2, 1);
        if (rc != SCRF_OK) return rc;
        HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));
        h->early_done = true;
      }
      trans_done = true;   // the state part (fused or dense) follows below, under the transition block's all-reduce
    }
    bool side = false;   // part of the count work runs on the second stream (below)
    {
      PhaseTimer tm(h, PH_EXPF, cb.st);
      uint32_t nl = 1;
      if (cb.fused) {
        const uint32_t W0 = b->recipe[0].in_width;
        ScrfFusedArgs fa = fused_args(h, b, u0, cb.expf_tiles);
        // Side stream (round 4): the per-frame count contraction k_ztf, its reduction and the transition counts A^T B
        // need Z / the recursion's vectors, not R's contraction -- they run on the engine's second stream UNDER
        // k_expf_fused_ws, whose one workgroup per CU leaves 80 registers per SIMD lane and the wave slots for a narrow
        // k_ztf (one output tile per wavefront).  The two sides add into disjoint weights; the streams join before the
        // commit.  Off while kernels are timed one by one (scrf_enable_timing) and with SCRF_SIDE=0.
        side = h->side_stream && !use2 && !h->timing && cb.z_ready && pframe_supported(W0) && !l.use_tf && !segtrans(h) && cb.wave &&
               fused_expf_plan(l, W0, f32, cb.la ? 1 : 0).ws;
        if (side) {
          HIPCHK(h, hipEventRecord(h->ev_fork, cb.st));
          HIPCHK(h, hipStreamWaitEvent(h->stream2, h->ev_fork, 0));
          launch_ztf(h->stream2, cb.Z, (cb.la ? 6 : 5) * l.L, b->d_frames[0] + b->frame_off[u0] * W0, W0, nfr, cb.rpc_l, cb.nch_l, cb.slab_l, 1);
          launch_reduce_slabs(h->stream2, cb.slab_l, cb.nch_l, (cb.la ? 6 : 5) * l.L, l, spec_samples(W0), cb.grad);
          launch_atb(h->stream2, l, cb.fA, cb.fB, nfr, cb.rpc_atb, cb.nch_atb, cb.slab_atb, h->d_m0, cb.grad, cb.lin ? cb.dl.gsd : nullptr);
          HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));
        }
        {
          PhaseTimer tk(h, PH_K_EXPF, cb.st);
          KT_RUN("k_expf_fused", cb.st, launch_expf_fused(cb.st, fa, l, cb.R, b->tile_off[cb.expf_tiles][u1] - b->tile_off[cb.expf_tiles][u0], cb.slab_s, cb.slab_d, f32, cb.la ? 1 : 0));
          tk.stop(1);
        }
        if (!cb.z_ready) KT_RUN("k_lin_z", cb.st, launch_lin_z(cb.st, l, bv, u0, (uint32_t)nutt, cb.R, cb.Z));
        if (side) {
        } else if (pframe_supported(W0))
          KT_RUN("k_ztf", cb.st, launch_ztf(cb.st, cb.Z, (cb.la ? 6 : 5) * l.L, b->d_frames[0] + b->frame_off[u0] * W0, W0, nfr, cb.rpc_l, cb.nch_l, cb.slab_l));
        else
          KT_RUN("k_expf_mfma(samples)", cb.st, launch_expf_mfma(cb.st, cb.Z, (cb.la ? 6 : 5) * l.L, b->d_frames[0] + b->frame_off[u0] * W0, W0, nullptr, nfr, l,
                           spec_samples(W0), cb.rpc_l, cb.nch_l, cb.slab_l));
        nl += 2;
      } else if (cb.hybrid) {
        const uint32_t W0 = b->recipe[0].in_width;
        {
          PhaseTimer tk(h, PH_K_EXPF, cb.st);
          // [avg | max | min] only: the one-hot duration and bias counts are sums of R (k_lin_z5), and 3 W columns are one
          // 384-column tile of the count kernel where 3 W + D + 1 were two
          KT_RUN("k_expf_mfma(state)", cb.st, launch_expf_mfma(cb.st, cb.R, l.L, cb.X, hybrid_row_floats(l, W0), nullptr, nseg, l, spec_stats_x(l, W0), cb.rpc_s, cb.nch_s, cb.slab_s, f32));
          tk.stop(1);
        }
        uint32_t t_max = 0;
        for (uint64_t u = u0; u < u1; u++) t_max = std::max(t_max, b->T[u]);
        KT_RUN("k_lin_z5", cb.st, launch_lin_z5(cb.st, l, bv, u0, (uint32_t)nutt, t_max, nfr, cb.R, cb.Z, cb.slab_d));
        if (pframe_supported(W0))
          KT_RUN("k_ztf", cb.st, launch_ztf(cb.st, cb.Z, 5 * l.L, b->d_frames[0] + b->frame_off[u0] * W0, W0, nfr, cb.rpc_l, cb.nch_l, cb.slab_l));
        else
          KT_RUN("k_expf_mfma(samples)", cb.st, launch_expf_mfma(cb.st, cb.Z, 5 * l.L, b->d_frames[0] + b->frame_off[u0] * W0, W0, nullptr, nfr, l,
                           spec_samples(W0), cb.rpc_l, cb.nch_l, cb.slab_l));
        nl += 2;
      } else {
        PhaseTimer tk(h, PH_K_EXPF, cb.st);
        if (fast) KT_RUN("k_expf_mfma(state)", cb.st, launch_expf_mfma(cb.st, cb.R, l.L, cb.X, l.F, nullptr, nseg, l, scrf_spec_state(l), cb.rpc_s, cb.nch_s, cb.slab_s, f32));
        else KT_RUN("k_expf_gemm(state)", cb.st, launch_expf_gemm(cb.st, cb.R, l.L, cb.X, l.F, nullptr, nseg, l, 0, cb.rpc_s, cb.nch_s, cb.slab_s));
        tk.stop(1);
      }
      if (segtrans(h)) {
        // transition counts of the segment's own window: XI2 rows are windows, no row map
        if (fast) KT_RUN("k_expf_mfma(trans)", cb.st, launch_expf_mfma(cb.st, cb.XI, l.L * l.L, cb.X, l.F, nullptr, nseg, l, scrf_spec_trans(l), cb.rpc_t, cb.nch_t, cb.slab_t, f32));
        else KT_RUN("k_expf_gemm(trans)", cb.st, launch_expf_gemm(cb.st, cb.XI, l.L * l.L, cb.X, l.F, nullptr, nseg, l, 1, cb.rpc_t, cb.nch_t, cb.slab_t));
        nl += 1;
      } else if (l.use_tf && !trans_done) {
        launch_frame_rows(cb.st, bv, u0, u1, l.D, nfr, cb.xrow_next, 1);
        if (fast) KT_RUN("k_expf_mfma(trans)", cb.st, launch_expf_mfma(cb.st, cb.XI, l.L * l.L, cb.X, l.F, cb.xrow_next, nfr, l, scrf_spec_trans(l), cb.rpc_t, cb.nch_t, cb.slab_t, f32));
        else KT_RUN("k_expf_gemm(trans)", cb.st, launch_expf_gemm(cb.st, cb.XI, l.L * l.L, cb.X, l.F, cb.xrow_next, nfr, l, 1, cb.rpc_t, cb.nch_t, cb.slab_t));
        nl += 2;
      }
      tm.stop(nl);
    }
    {
      PhaseTimer tm(h, PH_REDUCE, cb.st);
      KernelTimer kt(h, "reductions (k_reduce_slabs, k_atb, k_batch_sums)", cb.st);
      if (cb.fused) {
        const uint32_t W0 = b->recipe[0].in_width;
        const ScrfFusedExpfPlan plan = fused_expf_plan(l, W0, f32, cb.la ? 1 : 0);
        if (plan.ndur) {
          // dense groups [avg |] max | min at columns (5 + g0) W ..; one-hot duration + bias counts from their own slab
          launch_reduce_slabs(cb.st, cb.slab_s, cb.nch_s, l.L, l, ScrfGemmSpec{0, 0, plan.ncol, 0, 0.0, (5 + plan.g0) * W0, 0}, cb.grad);
          launch_reduce_slabs(cb.st, cb.slab_d, cb.nch_s, l.L, l, ScrfGemmSpec{0, 0, l.D, (uint32_t)l.use_sb, l.sbv, 8 * W0, 0}, cb.grad);
        } else launch_reduce_slabs(cb.st, cb.slab_s, cb.nch_s, l.L, l, spec_dense(l, W0), cb.grad);
        if (!side) launch_reduce_slabs(cb.st, cb.slab_l, cb.nch_l, (cb.la ? 6 : 5) * l.L, l, spec_samples(W0), cb.grad);
      } else if (cb.hybrid) {
        const uint32_t W0 = b->recipe[0].in_width;
        uint32_t t_max = 0;
        for (uint64_t u = u0; u < u1; u++) t_max = std::max(t_max, b->T[u]);
        int seg_len = 0;
        const uint32_t nblk_d = (uint32_t)nutt * lin_z5_segments((uint32_t)nutt, l.L, l.D, t_max, &seg_len);
        launch_reduce_slabs(cb.st, cb.slab_s, cb.nch_s, l.L, l, spec_stats_x(l, W0), cb.grad);
        launch_reduce_slabs(cb.st, cb.slab_d, nblk_d, l.L, l, ScrfGemmSpec{0, 0, l.D, (uint32_t)l.use_sb, l.sbv, 8 * W0, 0}, cb.grad);
        launch_reduce_slabs(cb.st, cb.slab_l, cb.nch_l, 5 * l.L, l, spec_samples(W0), cb.grad);
      } else launch_reduce_slabs(cb.st, cb.slab_s, cb.nch_s, l.L, l, scrf_spec_state(l), cb.grad);
      if (side) HIPCHK(h, hipStreamWaitEvent(cb.st, h->ev_join, 0));   // the side stream's weights are in
      else if (trans_done) {}
      else if (l.use_tf || segtrans(h)) launch_reduce_slabs(cb.st, cb.slab_t, cb.nch_t, l.L * l.L, l, scrf_spec_trans(l), cb.grad);
      else if (cb.wave) launch_atb(cb.st, l, cb.fA, cb.fB, nfr, cb.rpc_atb, cb.nch_atb, cb.slab_atb, h->d_m0, cb.grad,
                                   cb.lin ? cb.dl.gsd : nullptr);
      else launch_reduce_xiacc(cb.st, cb.xi_acc, (uint32_t)nutt, l, cb.grad);
      launch_batch_sums(cb.st, b->d_numer + u0, b->d_zx + u0, (uint32_t)nutt, cb.sums);
      kt.stop(4);
      tm.stop(3);
    }
    HIPCHK(h, hipGetLastError());
  }
  if (use2) {
    // join: lane 1's partial gradient and sums are added once, in a fixed order
    HIPCHK(h, hipEventRecord(h->ev_join, h->stream2));
    HIPCHK(h, hipStreamWaitEvent(h->stream, h->ev_join, 0));
    launch_add(h->stream, h->d_stage, h->d_grad2, l.lambda_len);
    launch_add(h->stream, h->d_sums_stage, h->d_sums2, 3);
    int rc = queue_status(h, b);
    if (rc != SCRF_OK) return rc;
  }
  if (!l.use_tf && wave_path(h, true)) launch_add_trans_counts(h->stream, b->d_trans_counts, l, h->d_stage);
  if (h->early_done)   // the transition block was committed (and sent) before the state contraction
    hipLaunchKernelGGL(k_commit_part, dim3((l.lambda_len + 255) / 256), dim3(256), 0, h->stream, h->d_grad, h->d_stage, l.lambda_len,
                       l.stride, l.nsf, 0, h->d_sums, h->d_sums_stage, h->d_latch);
  else
  hipLaunchKernelGGL(k_commit, dim3((l.lambda_len + 255) / 256), dim3(256), 0, h->stream, h->d_grad, h->d_stage, l.lambda_len,
                     h->d_sums, h->d_sums_stage, h->d_latch);
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventSynchronize(h->ev_status));
  latch_decode(h->h_latch, latch);
  return SCRF_OK;
}

extern "C" int scrf_fb_batch(scrf_handle h, scrf_batch b, double* numer, double* zx) {
  if (!h || !b) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  if (!b->d_labels) return fail(h, SCRF_ERR_INVALID, "scrf_fb_batch: the batch carries no labels");
  if (h->timing) {
    memset(h->ms, 0, sizeof(h->ms)); memset(h->nlaunch, 0, sizeof(h->nlaunch)); h->ktimes.clear();
    hipEventRecord(h->ev[SCRF_N_PHASES][0], h->stream);
  }
  int latch[2] = {0, 0};
  bool used_lin = false;
  int rc = fb_run(h, b, latch, &used_lin);
  if (rc != SCRF_OK) return rc;
  if (latch[0] == SCRF_ERR_NUMERIC && wave_path(h, true)) {
    // the wavefront recursions take their transition step on exp(M - max M) (and the linear-domain one
    // flushes what lies ~700 nats below a frame's maximum); where that empties a whole vector or breaks a
    // posterior-mass check, the batch is redone with the workgroup kernel, a column-wise max-shifted
    // log-sum-exp like the reference's LogMath -- the staged gradient of the first pass was dropped
    h->n_lin_fallback++;
    h->force_fb = true;
    rc = fb_run(h, b, latch, &used_lin);
    h->force_fb = false;
    if (rc != SCRF_OK) return rc;
  }
  if (h->timing) {
    hipEventRecord(h->ev[SCRF_N_PHASES][1], h->stream);
    hipEventSynchronize(h->ev[SCRF_N_PHASES][1]);
    hipEventElapsedTime(&h->ms[PH_ALL], h->ev[SCRF_N_PHASES][0], h->ev[SCRF_N_PHASES][1]);
  }
  if (latch[0] != 0) return fail(h, latch[0], "utterance %d: %s", latch[1], status_text(latch[0]));
  if (numer || zx) {
    if (numer) HIPCHK(h, hipMemcpyAsync(numer, b->d_numer, sizeof(double) * b->U, hipMemcpyDeviceToHost, h->stream));
    if (zx) HIPCHK(h, hipMemcpyAsync(zx, b->d_zx, sizeof(double) * b->U, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  }
  return SCRF_OK;
}

extern "C" int scrf_add_grad(scrf_handle h, const double* g, uint32_t n) {
  if (!h || !g) return SCRF_ERR_INVALID;
  if (n != h->xlay.lambda_len) return fail(h, SCRF_ERR_INVALID, "scrf_add_grad: length mismatch");
  HIPCHK(h, hipSetDevice(h->device));
  if (h->shadow) {
    h->xbuf.assign(h->lay.lambda_len, 0.0);
    for (uint32_t i = 0; i < n; i++) h->xbuf[h->s2d[i]] = g[i];
    g = h->xbuf.data();
    n = h->lay.lambda_len;
  }
  int rc = ensure_scratch(h, sizeof(double) * n);
  if (rc != SCRF_OK) return rc;
  HIPCHK(h, hipMemcpyAsync(h->scratch, g, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  launch_add(h->stream, h->d_grad, (const double*)h->scratch, n);
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// parity hooks (single utterance)
// ---------------------------------------------------------------------------------------------
static int check_u(scrf_handle h, scrf_batch b, uint32_t u, const char* fn) {
  if (!h || !b) return SCRF_ERR_INVALID;
  if (u >= b->U) return fail(h, SCRF_ERR_INVALID, "%s: utterance %u >= %u", fn, u, b->U);
  return SCRF_OK;
}

extern "C" int scrf_windows(scrf_handle h, scrf_batch b, uint32_t u, float* out) {
  int rc = check_u(h, b, u, "scrf_windows");
  if (rc != SCRF_OK) return rc;
  if (!out) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  Need nd{false, false, false, false};
  ChunkBufs cb;
  rc = carve(h, b, u, u + 1, nd, &cb);
  if (rc != SCRF_OK) return rc;
  rc = run_scores(h, b, u, u + 1, cb);
  if (rc != SCRF_OK) return rc;
  uint64_t nseg = b->seg_off[u + 1] - b->seg_off[u];
  HIPCHK(h, hipMemcpyAsync(out, cb.X, sizeof(float) * nseg * h->lay.F, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return SCRF_OK;
}

static int copy_M(scrf_handle h, const ChunkBufs& cb, uint32_t T, double* M) {
  const size_t LL = (size_t)h->lay.L * h->lay.L;
  if (cb.m_per_frame == 2) {   // STDSEG_NO_DUR: one matrix per window
    HIPCHK(h, hipMemcpyAsync(M, cb.M, sizeof(double) * scrf_seg_base(T, h->lay.D) * LL, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  } else if (cb.m_per_frame) {
    HIPCHK(h, hipMemcpyAsync(M, cb.M, sizeof(double) * T * LL, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
  } else {
    HIPCHK(h, hipMemcpyAsync(M, cb.M, sizeof(double) * LL, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (uint32_t t = 1; t < T; t++) memcpy(M + t * LL, M, sizeof(double) * LL);
  }
  return SCRF_OK;
}

extern "C" int scrf_scores(scrf_handle h, scrf_batch b, uint32_t u, double* S, double* M) {
  int rc = check_u(h, b, u, "scrf_scores");
  if (rc != SCRF_OK) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  if (nstate(h)) {   // S [T][nLabs]; M [T][2*nLabs + P*P]: self transitions | c -> c+1 | end state of p -> start state of q
    NstateBufs nb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = nstate_run_chunk(h, b, u, u + 1, false, nullptr, &nb);
    if (rc != SCRF_OK) return rc;
    const uint64_t T = b->T[u], L = h->lay.L, P = L / h->lay.K, w = 2 * L + P * P;
    if (S) HIPCHK(h, hipMemcpyAsync(S, nb.S, sizeof(double) * T * L, hipMemcpyDeviceToHost, h->stream));
    if (M) {
      HIPCHK(h, hipMemcpy2DAsync(M, sizeof(double) * w, nb.TD, sizeof(double) * L, sizeof(double) * L, T, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemcpy2DAsync(M + L, sizeof(double) * w, nb.TO, sizeof(double) * L, sizeof(double) * L, T, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemcpy2DAsync(M + 2 * L, sizeof(double) * w, nb.TE, sizeof(double) * P * P, sizeof(double) * P * P, T, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SCRF_OK;
  }
  if (stdseg(h)) {   // S [N_seg][nActualLabs], M [N_seg][nLabs][nActualLabs]
    StdsegBufs sb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = stdseg_run_chunk(h, b, u, u + 1, false, nullptr, &sb);
    if (rc != SCRF_OK) return rc;
    const uint64_t ns = b->seg_off[u + 1] - b->seg_off[u];
    const uint32_t La = stdseg_La(h);
    if (S) HIPCHK(h, hipMemcpyAsync(S, sb.S, sizeof(double) * ns * La, hipMemcpyDeviceToHost, h->stream));
    if (M) HIPCHK(h, hipMemcpyAsync(M, sb.MX, sizeof(double) * ns * h->lay.L * La, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return SCRF_OK;
  }
  Need nd{false, false, false, false};
  ChunkBufs cb;
  rc = carve(h, b, u, u + 1, nd, &cb);
  if (rc != SCRF_OK) return rc;
  rc = run_scores(h, b, u, u + 1, cb);
  if (rc != SCRF_OK) return rc;
  uint64_t nseg = b->seg_off[u + 1] - b->seg_off[u];
  if (S) HIPCHK(h, hipMemcpyAsync(S, cb.S, sizeof(double) * nseg * h->lay.L, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (M) return copy_M(h, cb, b->T[u], M);
  return SCRF_OK;
}

extern "C" int scrf_forward_backward(scrf_handle h, scrf_batch b, uint32_t u, uint32_t prec, double* alpha_dur,
                                     double* alpha, double* beta, double* zx) {
  int rc = check_u(h, b, u, "scrf_forward_backward");
  if (rc != SCRF_OK) return rc;
  if (prec != SCRF_PREC_EXACT) return fail(h, SCRF_ERR_INVALID, "scrf_forward_backward: the node-value hook runs at SCRF_PREC_EXACT only");
  HIPCHK(h, hipSetDevice(h->device));
  if (nstate(h)) {   // alpha, beta: [T][nLabs]; alpha_dur is not written
    NstateBufs nb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = nstate_run_chunk(h, b, u, u + 1, false, nullptr, &nb);
    if (rc != SCRF_OK) return rc;
    const uint64_t n = (uint64_t)b->T[u] * h->lay.L;
    int st = 0;
    if (alpha) HIPCHK(h, hipMemcpyAsync(alpha, nb.alpha, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    if (beta) HIPCHK(h, hipMemcpyAsync(beta, nb.beta, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    if (zx) HIPCHK(h, hipMemcpyAsync(zx, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&st, b->d_status + u, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (st != SCRF_OK) return fail(h, st, "utterance %u: numeric failure in forward-backward", u);
    return SCRF_OK;
  }
  if (stdseg(h)) {   // alpha_dur and beta: the nodes' alpha / beta over full labels, [N_seg][nActualLabs]; `alpha` is not written
    StdsegBufs sb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = stdseg_run_chunk(h, b, u, u + 1, false, nullptr, &sb);
    if (rc != SCRF_OK) return rc;
    const uint64_t ns = b->seg_off[u + 1] - b->seg_off[u];
    const uint32_t La = stdseg_La(h);
    int st = 0;
    if (alpha_dur) HIPCHK(h, hipMemcpyAsync(alpha_dur, sb.alpha, sizeof(double) * ns * La, hipMemcpyDeviceToHost, h->stream));
    if (beta) HIPCHK(h, hipMemcpyAsync(beta, sb.beta, sizeof(double) * ns * La, hipMemcpyDeviceToHost, h->stream));
    if (zx) HIPCHK(h, hipMemcpyAsync(zx, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(&st, b->d_status + u, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (st != SCRF_OK) return fail(h, st, "utterance %u: numeric failure in forward-backward", u);
    return SCRF_OK;
  }
  const ScrfLayout& l = h->lay;
  Need nd{true, false, true, false};
  uint64_t nseg = b->seg_off[u + 1] - b->seg_off[u];
  uint32_t T = b->T[u];
  // second attempt: the workgroup kernel (reference LogMath) when the wavefront recursion gave up
  for (int attempt = 0; attempt < 2; attempt++) {
    ChunkBufs cb;
    rc = carve(h, b, u, u + 1, nd, &cb);
    if (rc == SCRF_OK) rc = run_scores(h, b, u, u + 1, cb);
    if (rc == SCRF_OK && hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream) != hipSuccess) rc = fail(h, SCRF_ERR_HIP, "scrf_forward_backward: memset failed");
    if (rc == SCRF_OK) rc = run_dp(h, b, u, u + 1, cb, false, nullptr);
    int st = 0;
    if (rc == SCRF_OK && hipMemcpyAsync(&st, b->d_status + u, sizeof(int), hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(h, SCRF_ERR_HIP, "scrf_forward_backward: status copy failed");
    if (rc == SCRF_OK && hipStreamSynchronize(h->stream) != hipSuccess) rc = fail(h, SCRF_ERR_HIP, "scrf_forward_backward: synchronize failed");
    if (rc != SCRF_OK) { h->force_fb = false; return rc; }
    if (st == SCRF_ERR_NUMERIC && !h->force_fb && wave_path(h, false)) { h->force_fb = true; continue; }
    h->force_fb = false;
    if (st != SCRF_OK && st != SCRF_ERR_BAD_LABEL) return fail(h, st, "utterance %u: numeric failure in forward-backward", u);
    if (alpha_dur) HIPCHK(h, hipMemcpyAsync(alpha_dur, cb.AD, sizeof(double) * nseg * l.L, hipMemcpyDeviceToHost, h->stream));
    if (alpha) HIPCHK(h, hipMemcpyAsync(alpha, cb.alpha, sizeof(double) * T * l.L, hipMemcpyDeviceToHost, h->stream));
    if (beta) HIPCHK(h, hipMemcpyAsync(beta, cb.beta, sizeof(double) * T * l.L, hipMemcpyDeviceToHost, h->stream));
    if (zx) HIPCHK(h, hipMemcpyAsync(zx, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    break;
  }
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// decode
// ---------------------------------------------------------------------------------------------
extern "C" int scrf_lattice_arcs(scrf_handle h, scrf_batch b, uint32_t u, int norm, scrf_arc* arcs, uint64_t* n_arcs,
                                 uint32_t* n_states, int32_t* final_state) {
  int rc = check_u(h, b, u, "scrf_lattice_arcs");
  if (rc != SCRF_OK) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  if (nstate(h)) {   // decoders/CRF_LatticeBuilder.h nStateBuildLattice
    const uint32_t T = b->T[u], L = h->lay.L;
    const uint64_t na = b->arc_off[u + 1] - b->arc_off[u];
    if (n_arcs) *n_arcs = na;
    if (n_states) *n_states = L * T + 2;
    if (final_state) *final_state = (int32_t)(L * T + 1);
    if (!arcs) return SCRF_OK;
    NstateBufs nb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = nstate_run_chunk(h, b, u, u + 1, false, nullptr, &nb);
    if (rc != SCRF_OK) return rc;
    float final_w = 0.0f;
    if (norm) {
      double asum = 0;
      HIPCHK(h, hipMemcpyAsync(&asum, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      final_w = (float)(-1 * asum);
    }
    scrf_arc* d_arcs = nullptr;
    HIPCHK(h, hipMalloc((void**)&d_arcs, sizeof(scrf_arc) * na));
    launch_ns_arcs(h->stream, h->lay, T, nb.S, nb.TD, nb.TO, nb.TE, final_w, d_arcs);
    hipError_t e = hipMemcpyAsync(arcs, d_arcs, sizeof(scrf_arc) * na, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_arcs);
    if (e != hipSuccess) return fail(h, SCRF_ERR_HIP, "scrf_lattice_arcs: %s", hipGetErrorString(e));
    return SCRF_OK;
  }
  if (stdseg(h)) {   // decoders/CRF_LatticeBuilder_StdSeg.h: one state per (node, available full label)
    const uint32_t La = stdseg_La(h), T = b->T[u];
    const uint64_t ns = b->seg_off[u + 1] - b->seg_off[u], na = b->arc_off[u + 1] - b->arc_off[u];
    if (n_arcs) *n_arcs = na;
    if (n_states) *n_states = (uint32_t)(2 + ns * La);
    if (final_state) *final_state = (int32_t)(1 + ns * La);
    if (!arcs) return SCRF_OK;
    StdsegBufs sb;
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = stdseg_run_chunk(h, b, u, u + 1, false, nullptr, &sb);
    if (rc != SCRF_OK) return rc;
    float final_w = -0.0f;
    if (norm) {
      double asum = 0;
      HIPCHK(h, hipMemcpyAsync(&asum, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      const double Zx = -1 * asum;
      final_w = (float)(-Zx);
    }
    std::vector<uint64_t> off;
    stdseg_row_arc_offsets(T, La, h->lay.D, &off);
    uint64_t* d_off = nullptr;
    scrf_arc* d_arcs = nullptr;
    hipError_t e = hipMalloc((void**)&d_off, sizeof(uint64_t) * off.size());
    if (e == hipSuccess) e = hipMalloc((void**)&d_arcs, sizeof(scrf_arc) * na);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off.data(), sizeof(uint64_t) * off.size(), hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) {
      launch_stdseg_arcs(h->stream, h->lay, La, T, ns, d_off, sb.S, sb.MX, final_w, d_arcs);
      e = hipMemcpyAsync(arcs, d_arcs, sizeof(scrf_arc) * na, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    hipFree(d_off); hipFree(d_arcs);
    if (e != hipSuccess) return fail(h, SCRF_ERR_HIP, "scrf_lattice_arcs: %s", hipGetErrorString(e));
    return SCRF_OK;
  }
  const ScrfLayout& l = h->lay;
  const uint32_t T = b->T[u];
  const bool frame_model = h->cfg.model_type == SCRF_STDFRAME;
  const uint64_t na = b->arc_off[u + 1] - b->arc_off[u];
  if (n_arcs) *n_arcs = h->shadow ? shadow_num_arcs(h, T) : na;
  // STDSEG_NO_DUR: one state per (node, label) like the frame lattice (decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h)
  if (n_states) *n_states = (frame_model || segtrans(h)) ? l.L * T + 2 : (uint32_t)scrf_node_start_state(T, l.L) + 1;
  if (final_state) *final_state = (frame_model || segtrans(h)) ? (int32_t)(l.L * T + 1) : scrf_node_start_state(T, l.L);
  if (!arcs) return SCRF_OK;
  Need nd{norm != 0, false, false, false};
  ChunkBufs cb;
  rc = carve(h, b, u, u + 1, nd, &cb);
  if (rc != SCRF_OK) return rc;
  rc = run_scores(h, b, u, u + 1, cb);
  if (rc != SCRF_OK) return rc;
  float final_w = frame_model ? 0.0f : -0.0f;
  if (norm) {
    // Zx = -computeAlphaSum(): final arcs carry -Zx (segmental, :371,397) / Zx (frame, CRF_LatticeBuilder.h:191-198)
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    rc = run_dp(h, b, u, u + 1, cb, false, nullptr);
    if (rc != SCRF_OK) return rc;
    double asum = 0;
    HIPCHK(h, hipMemcpyAsync(&asum, b->d_zx + u, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    double Zx = -1 * asum;
    final_w = frame_model ? (float)Zx : (float)(-Zx);
  }
  scrf_arc* d_arcs = nullptr;
  HIPCHK(h, hipMalloc((void**)&d_arcs, sizeof(scrf_arc) * na));
  if (segtrans(h)) launch_arcs_segtrans(h->stream, l, T, cb.S, cb.M, final_w, d_arcs);
  else launch_arcs(h->stream, l, T, frame_model, cb.S, cb.M, cb.m_per_frame, final_w, d_arcs);
  std::vector<scrf_arc> dense;
  if (h->shadow) dense.resize(na);
  hipError_t e = hipMemcpyAsync(h->shadow ? dense.data() : arcs, d_arcs, sizeof(scrf_arc) * na, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  hipFree(d_arcs);
  if (e != hipSuccess) return fail(h, SCRF_ERR_HIP, "scrf_lattice_arcs: %s", hipGetErrorString(e));
  if (h->shadow) {
    // nStateBuildLattice :563-597: of the L*L boundary arcs of a frame (state-major, previous label ascending) a
    // phone's start state keeps those from the end states, ascending, and then its own; any other state the one from
    // the state before it and then its own.  States, segment arcs and final arcs are the dense lattice's.
    const uint32_t L = l.L, K = h->xlay.K;
    const scrf_arc* in = dense.data();
    scrf_arc* out = arcs;
    for (uint32_t t = 0; t < T; t++) {
      if (t > 0) {
        for (uint32_t lab = 0; lab < L; lab++, in += L) {
          if (lab % K == 0) { for (uint32_t e2 = K - 1; e2 < L; e2 += K) *out++ = in[e2]; }
          else *out++ = in[lab - 1];
          *out++ = in[lab];
        }
      }
      const uint64_t nseg = (uint64_t)L * scrf_node_max_dur(t, l.D);
      memcpy(out, in, sizeof(scrf_arc) * nseg);
      out += nseg; in += nseg;
    }
    memcpy(out, in, sizeof(scrf_arc) * L);
    out += L; in += L;
    if ((uint64_t)(out - arcs) != shadow_num_arcs(h, T) || (uint64_t)(in - dense.data()) != na)
      return fail(h, SCRF_ERR_INVALID, "scrf_lattice_arcs: arc count mismatch (internal)");
  }
  return SCRF_OK;
}

extern "C" int scrf_viterbi_batch(scrf_handle h, scrf_batch b, uint32_t* seg_labels, uint64_t max_labels,
                                  uint64_t* lab_off, float* best_cost) {
  if (!h || !b || !seg_labels || !lab_off) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const ScrfLayout& l = h->lay;
  const bool frame_model = h->cfg.model_type == SCRF_STDFRAME;
  if (h->timing) { memset(h->ms, 0, sizeof(h->ms)); memset(h->nlaunch, 0, sizeof(h->nlaunch)); h->ktimes.clear(); hipEventRecord(h->ev[SCRF_N_PHASES][0], h->stream); }
  const uint64_t NF = b->frame_off[b->U];
  if (NF > h->dec_cap_f) {
    hipFree(h->dec_lab); h->dec_lab = nullptr;
    if (h->dec_hlab) hipHostFree(h->dec_hlab);
    h->dec_hlab = nullptr; h->dec_cap_f = 0;
    HIPCHK(h, hipMalloc((void**)&h->dec_lab, sizeof(uint32_t) * NF));
    HIPCHK(h, hipHostMalloc((void**)&h->dec_hlab, sizeof(uint32_t) * NF, hipHostMallocDefault));
    h->dec_cap_f = NF;
  }
  if (b->U > h->dec_cap_u) {
    hipFree(h->dec_n); hipFree(h->dec_cost); h->dec_n = nullptr; h->dec_cost = nullptr;
    if (h->dec_hn) hipHostFree(h->dec_hn);
    if (h->dec_hcost) hipHostFree(h->dec_hcost);
    h->dec_hn = nullptr; h->dec_hcost = nullptr; h->dec_cap_u = 0;
    HIPCHK(h, hipMalloc((void**)&h->dec_n, sizeof(uint32_t) * b->U));
    HIPCHK(h, hipMalloc((void**)&h->dec_cost, sizeof(float) * b->U));
    HIPCHK(h, hipHostMalloc((void**)&h->dec_hn, sizeof(uint32_t) * b->U, hipHostMallocDefault));
    HIPCHK(h, hipHostMalloc((void**)&h->dec_hcost, sizeof(float) * b->U, hipHostMallocDefault));
    h->dec_cap_u = b->U;
  }
  uint32_t *d_lab = h->dec_lab, *d_n = h->dec_n;
  float* d_cost = h->dec_cost;
  Need nd{false, false, false, true};
  // fast decode: the fused score kernel writes the float arc weights and lists the entries whose
  // rounding it cannot guarantee; those are recomputed in reference order.  A chunk whose list
  // overflows (pathological cancellation) goes through the EXACT path instead.
  Need ndf = nd;
  ndf.fused = ndf.vitfast = true;
  const bool fast = h->fast_decode && b->fused_ok && !b->mixed && h->fuse_windows && !frame_model && l.L <= 0xffff;
  int rc = SCRF_OK;
  if (nstate(h)) {
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    for (uint32_t u0 = 0; u0 < b->U && rc == SCRF_OK;) {
      const uint32_t u1 = nstate_plan_chunk(h, b, u0, true);
      NstateBufs nb;
      rc = nstate_run_chunk(h, b, u0, u1, false, nullptr, &nb);
      if (rc != SCRF_OK) break;
      const uint64_t nfr = b->frame_off[u1] - b->frame_off[u0];
      float* vc = nullptr;
      uint16_t* bp = nullptr;
      hipError_t e = hipMalloc((void**)&vc, sizeof(float) * nfr * l.L);
      if (e == hipSuccess) e = hipMalloc((void**)&bp, sizeof(uint16_t) * nfr * l.L);
      if (e == hipSuccess) {
        launch_ns_viterbi(h->stream, l, b->view(), u0, u1 - u0, nb.S, nb.TD, nb.TO, nb.TE, vc, bp, d_lab, d_n, d_cost);
        e = hipStreamSynchronize(h->stream);
      }
      hipFree(vc); hipFree(bp);
      if (e != hipSuccess) { rc = fail(h, SCRF_ERR_HIP, "scrf_viterbi_batch: %s", hipGetErrorString(e)); break; }
      u0 = u1;
    }
  }
  if (stdseg(h)) {
    const uint32_t La = stdseg_La(h);
    HIPCHK(h, hipMemsetAsync(b->d_status, 0, sizeof(int) * b->U, h->stream));
    for (uint32_t u0 = 0; u0 < b->U && rc == SCRF_OK;) {
      // the chunk's scores in the scratch arena, path costs and back pointers behind them
      uint32_t u1 = stdseg_plan_chunk(h, b, u0, true);   // `post` sizing leaves room for the two decode arrays
      StdsegBufs sb;
      rc = stdseg_run_chunk(h, b, u0, u1, false, nullptr, &sb);
      if (rc != SCRF_OK) break;
      const uint64_t nseg = b->seg_off[u1] - b->seg_off[u0];
      float* vc = nullptr;
      uint16_t* bp = nullptr;
      hipError_t e = hipMalloc((void**)&vc, sizeof(float) * nseg * La);
      if (e == hipSuccess) e = hipMalloc((void**)&bp, sizeof(uint16_t) * nseg * La);
      if (e == hipSuccess) {
        launch_stdseg_viterbi(h->stream, h->lay, La, b->view(), u0, u1 - u0, sb.S, sb.MX, vc, bp, d_lab, d_n, d_cost);
        e = hipStreamSynchronize(h->stream);
      }
      hipFree(vc); hipFree(bp);
      if (e != hipSuccess) { rc = fail(h, SCRF_ERR_HIP, "scrf_viterbi_batch: %s", hipGetErrorString(e)); break; }
      u0 = u1;
    }
  }
  for (uint32_t u0 = 0; u0 < b->U && rc == SCRF_OK && !stdseg(h) && !nstate(h);) {
    uint32_t u_end = u0;
    if (fast) {
      const uint32_t u1 = plan_chunk(h, b, u0, ndf);
      ChunkBufs cb;
      rc = carve(h, b, u0, u1, ndf, &cb);
      if (rc != SCRF_OK) break;
      if (!h->m0_valid) {  // transition scores carry only the bias: one L x L matrix for every frame
        launch_scores_exact(cb.st, nullptr, l.F, nullptr, 1, h->d_lambda, l, 1, l.L * l.L, h->d_m0);
        launch_exp_m(cb.st, h->d_m0, l.L, 1, h->d_e0, h->d_et0, h->d_msh0);
        h->m0_valid = true;
      }
      rc = run_scores(h, b, u0, u1, cb);
      if (rc != SCRF_OK) break;
      uint32_t n_fix = 0;
      HIPCHK(h, hipMemcpyAsync(&n_fix, cb.fix_cnt, sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->n_decode_fix += n_fix;
      if (n_fix <= cb.fix_cap) {
        PhaseTimer tm(h, PH_VIT);
        launch_decode_fixup(h->stream, b->d_frames[0], b->recipe[0].in_width, b->view(), u0, u1, h->d_lambda, l, cb.fix_cnt,
                            cb.fix_list, std::min(n_fix, cb.fix_cap), cb.Wn);
        if (viterbi_fast_supported(l))
          launch_viterbi_fast(h->stream, l, b->view(), u0, u1 - u0, cb.Wn, cb.M, cb.bp_b, cb.bp_e, d_lab, d_n, d_cost);
        else
          launch_viterbi(h->stream, l, b->view(), u0, u1 - u0, nullptr, cb.M, cb.m_per_frame, 0, cb.bp_b, cb.bp_e, d_lab, d_n,
                         d_cost, cb.Wn);
        tm.stop(2);
        u0 = u1;
        continue;
      }
      h->n_decode_fallback++;
      u_end = u1;  // this range again, through the EXACT path
    }
    do {
      const uint32_t u1 = std::min(fast ? u_end : b->U, plan_chunk(h, b, u0, nd));
      ChunkBufs cb;
      rc = carve(h, b, u0, u1, nd, &cb);
      if (rc != SCRF_OK) break;
      rc = run_scores(h, b, u0, u1, cb);
      if (rc != SCRF_OK) break;
      PhaseTimer tm(h, PH_VIT);
      if (segtrans(h)) launch_viterbi_segtrans(h->stream, l, b->view(), u0, u1 - u0, cb.S, cb.M, cb.bp_b, cb.bp_e, d_lab, d_n, d_cost);
      else launch_viterbi(h->stream, l, b->view(), u0, u1 - u0, cb.S, cb.M, cb.m_per_frame, frame_model, cb.bp_b, cb.bp_e,
                          d_lab, d_n, d_cost);
      tm.stop(1);
      u0 = u1;
    } while (fast && u0 < u_end && rc == SCRF_OK);
  }
  const uint32_t *lab = h->dec_hlab, *cnt = h->dec_hn;
  if (rc == SCRF_OK) {
    hipError_t e = hipMemcpyAsync(h->dec_hlab, d_lab, sizeof(uint32_t) * NF, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h->dec_hn, d_n, sizeof(uint32_t) * b->U, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && best_cost) e = hipMemcpyAsync(h->dec_hcost, d_cost, sizeof(float) * b->U, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) rc = fail(h, SCRF_ERR_HIP, "scrf_viterbi_batch: %s", hipGetErrorString(e));
    else if (best_cost) memcpy(best_cost, h->dec_hcost, sizeof(float) * b->U);
  }
  if (rc != SCRF_OK) return rc;
  if (h->timing) {
    hipEventRecord(h->ev[SCRF_N_PHASES][1], h->stream);
    hipEventSynchronize(h->ev[SCRF_N_PHASES][1]);
    hipEventElapsedTime(&h->ms[PH_ALL], h->ev[SCRF_N_PHASES][0], h->ev[SCRF_N_PHASES][1]);
  }
  uint64_t pos = 0;
  for (uint32_t u = 0; u < b->U; u++) {
    lab_off[u] = pos;
    if (pos + cnt[u] > max_labels) return fail(h, SCRF_ERR_INVALID, "scrf_viterbi_batch: seg_labels capacity %llu too small", (unsigned long long)max_labels);
    memcpy(seg_labels + pos, &lab[b->frame_off[u]], sizeof(uint32_t) * cnt[u]);
    pos += cnt[u];
  }
  lab_off[b->U] = pos;
  return SCRF_OK;
}

extern "C" int scrf_batch_is_fused(scrf_handle h, scrf_batch b, int* fused) {
  if (!h || !b || !fused) return SCRF_ERR_INVALID;
  *fused = b->fused_ok && h->fuse_windows ? 1 : 0;
  // 3: hybrid (materialised dense statistics, sampled blocks through the per-frame projections)
  if (!*fused && b->hybrid_ok && h->hybrid && h->fuse_windows && h->cfg.train_precision >= SCRF_PREC_FAST && h->cfg.train_precision != SCRF_PREC_FAST32 &&
      h->lin_dp && wave_path(h, true)) *fused = 3;
  // 2: training runs with the linear window average (SCRF_PREC_FASTLIN on a shape its kernels take)
  if (*fused && h->cfg.train_precision == SCRF_PREC_FASTLIN && h->lin_dp && wave_path(h, true) &&
      fused_la_supported(h->lay, b->recipe[0].in_width)) *fused = 2;
  return SCRF_OK;
}

extern "C" int scrf_comm_stats(scrf_handle h, uint64_t* n_collectives, uint64_t* n_overlapped) {
  if (!h) return SCRF_ERR_INVALID;
  if (n_collectives) *n_collectives = h->n_collectives;
  if (n_overlapped) *n_overlapped = h->n_overlapped;
  return SCRF_OK;
}

extern "C" int scrf_set_frame_mass_check(scrf_handle h, int on) {
  if (!h) return SCRF_ERR_INVALID;
  h->frame_mass = on != 0;
  return SCRF_OK;
}

extern "C" int scrf_decode_stats(scrf_handle h, uint64_t* n_recomputed, uint64_t* n_fallback_chunks) {
  if (!h) return SCRF_ERR_INVALID;
  if (n_recomputed) *n_recomputed = h->n_decode_fix;
  if (n_fallback_chunks) *n_fallback_chunks = h->n_decode_fallback;
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// minibatch reduce + optimizer
// ---------------------------------------------------------------------------------------------
extern "C" int scrf_comm_unique_id(void* id128) {
  std::string why;
  if (!id128) return SCRF_ERR_INVALID;
  if (!rccl_load(&why)) return fail(nullptr, SCRF_ERR_COMM, "%s", why.c_str());
  scrf_nccl_uid id;
  int r = g_rccl.GetUniqueId(&id);
  if (r != 0) return fail(nullptr, SCRF_ERR_COMM, "ncclGetUniqueId failed: %d", r);
  memcpy(id128, &id, 128);
  return SCRF_OK;
}

extern "C" int scrf_comm_init(scrf_handle h, const void* id128, int rank, int n_ranks) {
  if (!h || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SCRF_ERR_INVALID;
  std::string why;
  if (!rccl_load(&why)) return fail(h, SCRF_ERR_COMM, "%s", why.c_str());
  HIPCHK(h, hipSetDevice(h->device));
  scrf_nccl_uid id;
  memcpy(&id, id128, 128);
  int r = g_rccl.CommInitRank(&h->comm, n_ranks, id, rank);
  if (r != 0) return fail(h, SCRF_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  h->rank = rank;
  h->nranks = n_ranks;
  return SCRF_OK;
}

__global__ void k_set_tail(double* sums8, int active, double e0, double e1, double e2, double e3) {
  sums8[3] = (double)active;
  sums8[4] = e0; sums8[5] = e1; sums8[6] = e2; sums8[7] = e3;
}
__global__ void k_div_by_active(double* __restrict__ g, uint32_t n, const double* __restrict__ sums4) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  double a = sums4[3];
  if (i < n && a > 0.0) g[i] = __ddiv_rn(g[i], a);  // grad[i] /= nStreams_active (:306-308)
}
__global__ void k_gauss_prior(double* __restrict__ g, uint32_t n, double inv) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] = __dsub_rn(g[i], __dmul_rn(g[i], inv));   // grad[i] -= grad[i] * invSquareVar
}

// ncclCommAbort: frees the communicator without waiting for outstanding collectives and makes the PEERS' pending
// operations fail (their watchdog below sees the asynchronous error) instead of leaving them in RCCL for ever.  The host
// calls it when a rank cannot enter or finish the per-step collective, then exits non-zero (a restart is a fresh process).
extern "C" int scrf_comm_abort(scrf_handle h) {
  if (!h) return SCRF_ERR_INVALID;
  if (h->comm) {
    if (g_rccl.CommAbort) g_rccl.CommAbort(h->comm);
    h->comm = nullptr;
  }
  return SCRF_OK;
}

// Bounded wait for the engine stream while a collective is on it: polls the stream and the communicator's asynchronous
// error state; after SCRF_COMM_TIMEOUT_S seconds (default 300) without completion, or on an asynchronous error (a peer
// aborted or died), the communicator is aborted and SCRF_ERR_COMM returned -- the caller ends the process.
static int wait_collective(scrf_handle h, const char* what) {
  const char* ts = getenv("SCRF_COMM_TIMEOUT_S");
  const double limit = ts && atof(ts) > 0 ? atof(ts) : 300.0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (;;) {
    const hipError_t q = hipStreamQuery(h->stream);
    if (q == hipSuccess) return SCRF_OK;
    if (q != hipErrorNotReady) { scrf_comm_abort(h); return fail(h, SCRF_ERR_HIP, "%s: %s", what, hipGetErrorString(q)); }
    if (h->comm && g_rccl.CommGetAsyncError) {
      int ae = 0;
      if (g_rccl.CommGetAsyncError(h->comm, &ae) == 0 && ae != 0) {
        scrf_comm_abort(h);
        return fail(h, SCRF_ERR_COMM, "%s: rank %d: the communicator reports an asynchronous error (%s): a peer failed or aborted", what,
                    h->rank, g_rccl.GetErrorString ? g_rccl.GetErrorString(ae) : "?");
      }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
      scrf_comm_abort(h);
      return fail(h, SCRF_ERR_COMM, "%s: rank %d: the collective did not complete within %.0f s (SCRF_COMM_TIMEOUT_S): a peer died or never "
                  "joined; communicator aborted", what, h->rank, limit);
    }
    usleep(200);
  }
}

// ---- the per-step collective in two blocks (models with transition features, e.g. the TIMIT demo: 4.37 M weights) ----
// The weight vector interleaves, per label, [nsf state weights | L * ntf transition weights]; 98.7 % of the TIMIT-demo
// gradient are transition weights.  Block 1 = the transition weights of every label (L contiguous pieces, one RCCL
// group), block 0 = the state weights of every label packed with the 8 scalars into one small message.  Every rank
// issues block 1 then block 0, whatever it did before (active, exhausted or failed): the sequence is part of the protocol.
// Inside scrf_fb_batch_allreduce block 1 is issued early, on the second stream, under the state contraction.
__global__ void k_pack_state(const double* __restrict__ grad, const double* __restrict__ sums8, uint32_t L, uint32_t stride,
                             uint32_t nsf, double* __restrict__ pack) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L * nsf) pack[i] = grad[(size_t)(i / nsf) * stride + i % nsf];
  else if (i < L * nsf + 8) pack[i] = sums8[i - L * nsf];
}
__global__ void k_unpack_state(const double* __restrict__ pack, uint32_t L, uint32_t stride, uint32_t nsf, double* __restrict__ grad,
                               double* __restrict__ sums8) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < L * nsf) grad[(size_t)(i / nsf) * stride + i % nsf] = pack[i];
  else if (i < L * nsf + 8) sums8[i - L * nsf] = pack[i];
}
static int allreduce_block(scrf_handle h, hipStream_t st, int part) {
  const ScrfLayout& l = h->lay;
  int r = 0;
  if (part == 1) {
    if (g_rccl.GroupStart && g_rccl.GroupEnd) r = g_rccl.GroupStart();
    for (uint32_t c = 0; c < l.L && r == 0; c++) {
      double* p = h->d_grad + (size_t)c * l.stride + l.nsf;
      r = g_rccl.AllReduce(p, p, (size_t)l.L * l.ntf, SCRF_NCCL_DOUBLE, SCRF_NCCL_SUM, h->comm, st);
    }
    if (g_rccl.GroupStart && g_rccl.GroupEnd) { const int r2 = g_rccl.GroupEnd(); if (r == 0) r = r2; }
  } else {
    const uint32_t n = l.L * l.nsf + 8;
    if (!h->d_pack && hipMalloc((void**)&h->d_pack, sizeof(double) * n) != hipSuccess) { scrf_comm_abort(h); return fail(h, SCRF_ERR_HIP, "hipMalloc of the pack buffer failed"); }
    hipLaunchKernelGGL(k_pack_state, dim3((n + 255) / 256), dim3(256), 0, st, h->d_grad, h->d_sums, l.L, l.stride, l.nsf, h->d_pack);
    r = g_rccl.AllReduce(h->d_pack, h->d_pack, n, SCRF_NCCL_DOUBLE, SCRF_NCCL_SUM, h->comm, st);
    hipLaunchKernelGGL(k_unpack_state, dim3((n + 255) / 256), dim3(256), 0, st, h->d_pack, l.L, l.stride, l.nsf, h->d_grad, h->d_sums);
  }
  if (r != 0) {
    scrf_comm_abort(h);
    return fail(h, SCRF_ERR_COMM, "ncclAllReduce (block %d) failed: %s", part, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  }
  return SCRF_OK;
}

extern "C" int scrf_allreduce_grad_ex(scrf_handle h, int active, const double* extra_in, uint32_t n_extra, double* sums4,
                                      double* extra_out) {
  if (!h || n_extra > 4 || (n_extra && !extra_in)) return SCRF_ERR_INVALID;
  // a rank that fails on its way into or out of the collective must not leave the others waiting in RCCL: every error
  // path below aborts the communicator first (the peers' wait_collective then ends with an error, not a hang)
#define ARCHK(call)                                                                                        \
  do {                                                                                                     \
    hipError_t e_ = (call);                                                                                \
    if (e_ != hipSuccess) { scrf_comm_abort(h); h->early_done = false; return fail(h, SCRF_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } \
  } while (0)
  ARCHK(hipSetDevice(h->device));
  const uint32_t n = h->lay.lambda_len;
  double e[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < n_extra; i++) e[i] = extra_in[i];
  hipLaunchKernelGGL(k_set_tail, dim3(1), dim3(1), 0, h->stream, h->d_sums, active ? 1 : 0, e[0], e[1], e[2], e[3]);
  ARCHK(hipGetLastError());
  if (h->comm) {
    h->n_collectives++;
    if (two_block_reduce(h)) {
      const bool early = h->early_done;
      h->early_done = false;
      if (early) h->n_overlapped++;
      if (early) ARCHK(hipStreamWaitEvent(h->stream, h->ev_join, 0));   // block 1 went out on the second stream already
      else { const int rc = allreduce_block(h, h->stream, 1); if (rc != SCRF_OK) return rc; }
      const int rc = allreduce_block(h, h->stream, 0);
      if (rc != SCRF_OK) return rc;
    } else {
      int r = g_rccl.AllReduce(h->d_grad, h->d_grad, n, SCRF_NCCL_DOUBLE, SCRF_NCCL_SUM, h->comm, h->stream);
      if (r == 0) r = g_rccl.AllReduce(h->d_sums, h->d_sums, 8, SCRF_NCCL_DOUBLE, SCRF_NCCL_SUM, h->comm, h->stream);
      if (r != 0) {
        scrf_comm_abort(h);
        return fail(h, SCRF_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
      }
    }
  }
  hipLaunchKernelGGL(k_div_by_active, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_grad, n, h->d_sums);
  ARCHK(hipGetLastError());
  if (sums4 || extra_out || h->comm) {
    double s8[8];
    ARCHK(hipMemcpyAsync(s8, h->d_sums, sizeof(double) * 8, hipMemcpyDeviceToHost, h->stream));
    if (h->comm) {
      const int rc = wait_collective(h, "scrf_allreduce_grad");
      if (rc != SCRF_OK) return rc;
    } else ARCHK(hipStreamSynchronize(h->stream));
    if (sums4) memcpy(sums4, s8, sizeof(double) * 4);
    if (extra_out) memcpy(extra_out, s8 + 4, sizeof(double) * n_extra);
  }
#undef ARCHK
  return SCRF_OK;
}

// scrf_fb_batch + scrf_allreduce_grad_ex in one call, for a host whose rank runs exactly one batch per step (the
// distributed minibatch accumulator): the all-reduce of the transition block overlaps the state contraction (above).
// The batch's own failure is reported through *fb_status (and extra[fail_slot] is raised, `active` cleared) while the
// collective still runs -- the peers learn it from the summed flag; the return value is the collective's.
extern "C" int scrf_fb_batch_allreduce(scrf_handle h, scrf_batch b, int active, const double* extra_in, uint32_t n_extra,
                                       uint32_t fail_slot, double* sums4, double* extra_out, int* fb_status) {
  if (!h || !b || n_extra > 4 || (n_extra && !extra_in) || (n_extra && fail_slot >= n_extra)) return SCRF_ERR_INVALID;
  h->overlap_comm = true;
  h->early_done = false;
  const int frc = scrf_fb_batch(h, b, nullptr, nullptr);
  h->overlap_comm = false;
  if (fb_status) *fb_status = frc;
  if (frc == SCRF_ERR_HIP || frc == SCRF_ERR_COMM) {   // the device or the communicator is gone: no collective can follow
    scrf_comm_abort(h);
    h->early_done = false;
    return frc;
  }
  const std::string batch_err = h->err;
  double e[4] = {0, 0, 0, 0};
  for (uint32_t i = 0; i < n_extra; i++) e[i] = extra_in[i];
  if (frc != SCRF_OK && n_extra) e[fail_slot] = 1.0;
  const int rc = scrf_allreduce_grad_ex(h, frc == SCRF_OK ? active : 0, e, n_extra, sums4, extra_out);
  if (rc == SCRF_OK && frc != SCRF_OK) h->err = batch_err;   // scrf_last_error keeps the batch's message
  return rc;
}
extern "C" int scrf_allreduce_grad(scrf_handle h, int active, double* sums4) {
  return scrf_allreduce_grad_ex(h, active, nullptr, 0, sums4, nullptr);
}

extern "C" int scrf_gauss_prior(scrf_handle h, float inv_square_var) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  const uint32_t n = h->lay.lambda_len;
  hipLaunchKernelGGL(k_gauss_prior, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d_grad, n, (double)inv_square_var);
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

extern "C" int scrf_div_grad(scrf_handle h, double d) {
  if (!h || d == 0.0) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  launch_scale(h->stream, h->d_grad, h->lay.lambda_len, d, 1);
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

extern "C" int scrf_scale_grad(scrf_handle h, double s) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  launch_scale(h->stream, h->d_grad, h->lay.lambda_len, s, 0);
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

extern "C" int scrf_sgd_step(scrf_handle h, double lr_or_eta, int use_adagrad, double eps) {
  if (!h) return SCRF_ERR_INVALID;
  HIPCHK(h, hipSetDevice(h->device));
  launch_sgd_step(h->stream, h->d_lambda, h->d_lambda_acc, h->d_gsa, h->d_grad, h->lay.lambda_len, lr_or_eta,
                  use_adagrad, eps);
  HIPCHK(h, hipMemsetAsync(h->d_sums, 0, sizeof(double) * 4, h->stream));
  h->m0_valid = false;
  HIPCHK(h, hipGetLastError());
  return SCRF_OK;
}

// ---------------------------------------------------------------------------------------------
// measurement
// ---------------------------------------------------------------------------------------------
extern "C" int scrf_kernel_timing(scrf_handle h, char* buf, size_t cap) {
  if (!h || !buf || cap == 0) return SCRF_ERR_INVALID;
  std::string out;
  char line[256];
  for (const auto& k : h->ktimes) {
    snprintf(line, sizeof line, "%s\t%.6f\t%u\n", k.name.c_str(), k.ms, k.n);
    out += line;
  }
  if (out.size() + 1 > cap) return fail(h, SCRF_ERR_INVALID, "scrf_kernel_timing: buffer of %zu bytes too small (%zu needed)", cap, out.size() + 1);
  memcpy(buf, out.c_str(), out.size() + 1);
  return SCRF_OK;
}
extern "C" int scrf_train_stats(scrf_handle h, uint64_t* n_lin_fallback) {
  if (!h) return SCRF_ERR_INVALID;
  if (n_lin_fallback) *n_lin_fallback = h->n_lin_fallback;
  return SCRF_OK;
}
extern "C" int scrf_enable_timing(scrf_handle h, int on) { if (!h) return SCRF_ERR_INVALID; h->timing = on != 0; return SCRF_OK; }
extern "C" int scrf_last_timing(scrf_handle h, float* ms, uint32_t* n_launch) {
  if (!h) return SCRF_ERR_INVALID;
  if (ms) memcpy(ms, h->ms, sizeof(h->ms));
  if (n_launch) memcpy(n_launch, h->nlaunch, sizeof(h->nlaunch));
  return SCRF_OK;
}
