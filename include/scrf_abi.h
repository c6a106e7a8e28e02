/*
 * scrf_abi.h -- C ABI of the MI355X-native segmental-CRF DP engine (libscrf_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of OSU-slatelab/ASR-CRaFT: per-utterance
 * segment/transition scores, forward-backward with expected-feature-count gradient,
 * Viterbi / lattice-arc emission, the minibatch gradient reduce and the SGD/AdaGrad step.
 * The reference has no FFI: the path sits behind in-process C++ virtual interfaces.  Each
 * entry point below names the reference interface it replaces (file:line relative to
 * /root/reference/CRF/src/).  C++ adaptor classes with the reference's own names
 * (CRF_Model, CRF_StdFeatureMap, CRF_StateNode accessors, CRF_GradBuilder,
 * CRF_Minibatch_GradAccumulator, CRF_LatticeBuilder_*) sit on top of this ABI in
 * asr-craft_amd/host/; INTEGRATION.md shows how a maintainer wires them in.
 *
 * Conventions
 *  - plain pointers and sizes, no C++/torch types; every call returns SCRF_OK (0) or an
 *    error code, scrf_last_error(h) gives the message the adaptor re-throws as
 *    std::runtime_error (the reference throws everywhere; SURVEY 8b "Errors").
 *  - one handle per process and GPU, calls serialized per handle (reference: one builder
 *    per pthread, model shared read-only; SURVEY 8b "Threading").
 *  - the engine fails loudly (SCRF_ERR_NO_DEVICE) when no gfx950 device is present: there
 *    is NO CPU fallback in this library.
 *  - labels: per frame t either SCRF_LAB_BAD or nActualLabs*(dur-1)+phone of the true
 *    segment ENDING at t (trainers/gradbuilders/CRF_NewGradBuilder_StdSeg_NoDur_NoTrans.cpp:216-231).
 *  - window order: for frame t the windows d=1..min(t+1,D) ending at t, each num_feas
 *    floats (what CRF_FeatureStream::read(bunch_size=min(t+1,D)) returns, :166).
 */
#ifndef SCRF_ABI_H_
#define SCRF_ABI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCRF_ABI_VERSION 1
#define SCRF_LAB_BAD 0xffffffffu /* CRF_LAB_BAD */

enum scrf_status {
  SCRF_OK = 0,
  SCRF_ERR_INVALID = 1,    /* bad argument / unsupported configuration */
  SCRF_ERR_NO_DEVICE = 2,  /* no HIP device: the product path never falls back to CPU */
  SCRF_ERR_HIP = 3,        /* a HIP runtime call or kernel failed */
  SCRF_ERR_NUMERIC = 4,    /* overflow/NaN/log(0) as thrown by utils/CRF_LogMath.cpp:192-224, or a failed
                              posterior-mass self-check of computeExpF (nodes/...WithoutSegTransFtr.cpp:917-947:
                              state mass and transition mass of a node within [-1e-6, 1+1e-6] and equal within
                              1e-6; frame model nodes/CRF_StdStateNode.cpp:252-275: both within [0.9, 1.1]) */
  SCRF_ERR_BAD_LABEL = 5,  /* label >= nActualLabs*labMaxDur (:627-631) */
  SCRF_ERR_EMPTY = 6,      /* "No features read from this sentence" (gradbuilder :331-335) */
  SCRF_ERR_COMM = 7        /* RCCL failure */
};

/* modeltype, CRF.h:50 */
enum scrf_model_type {
  SCRF_STDFRAME = 0,
  SCRF_STDSEG = 1,                       /* duration-labelled: num_labs = nActualLabs * lab_max_dur */
  SCRF_STDSEG_NO_DUR = 2,                /* transition features from the segment's own window (one L x L matrix per
                                            window): training, parity hooks, lattice arcs and best path (scrf_segtrans.hip) */
  SCRF_STDSEG_NO_DUR_NO_TRANSFTR = 3,    /* served by the same engine (bias-only transitions) */
  SCRF_STDSEG_NO_DUR_NO_SEGTRANSFTR = 4  /* the TIMIT-demo model */
};

/* ftrmaptype, CRF.h:40 (dense maps only) */
enum scrf_map_type { SCRF_STDSTATE = 0, SCRF_STDTRANS = 1 };

/* Arithmetic policy of the score / expected-count contractions.
 *  EXACT : fp64, ascending feature order, unfused multiply-then-add == the reference's
 *          CRF_StdFeatureMap::computeStateArrayValue bit for bit (needed for bit-identical
 *          lattice arcs / Viterbi).  Always used by the decode entry points.
 *  FAST  : fp64 MFMA (v_mfma_f64_16x16x4_f64); sums reordered, results within 1e-4 rel (measured
 *          <= 1e-9).  When the batch is one segment-recipe stream without context frames and the
 *          model has no transition features, the window vectors are never materialised: the five
 *          sampled-frame blocks are contracted per frame (an exact re-association of the same
 *          fp64 products) and avg/max/min are rebuilt in LDS with the reference's float
 *          arithmetic (scrf_fused.hip).
 *  FAST32 : like FAST but the two dense contractions run on the f32 MFMA (exact f32 FMA chain,
 *          twice the f64-MFMA rate of the chip): lambda and the posteriors are rounded to
 *          f32 as operands, scores sum in f32 over the feature axis, expected counts sum in f32
 *          inside 32/64-row chunks and in f64 across chunks.  The DP recursion, log-partition and
 *          posteriors stay f64.  Measured deviation <= ~1e-6 relative on gradients (contract 1e-4).
 *          Not the default of anything; opt-in.
 *  FASTLIN : FAST (fp64 throughout), except that the window AVERAGE of the segment recipe is taken as the exact
 *          mean (sum of the frames / length, linear in the frames) instead of the reference's float arithmetic
 *          (running float sum, float division: io/CRF_InFtrStream_SeqMultiWindow.cpp:609-646).  Linear means the
 *          average no longer has to be rebuilt and multiplied per window: its score share is a difference of
 *          prefix sums of one more per-frame projection, its expected counts one more per-frame sum -- a third
 *          of the dense matrix work of the fused kernels less.  The feature values differ from the reference's
 *          by their float rounding (<= 2^-24 relative per addition): measured 3e-8 relative on the gradient,
 *          1e-12 on the log-partition (contract 1e-4).  What bench.py runs.  Batches the fused kernels do not
 *          take (several streams, transition features, L > 64) run as FAST.  Decode never uses it. */
enum scrf_precision { SCRF_PREC_EXACT = 0, SCRF_PREC_FAST = 1, SCRF_PREC_FAST32 = 2, SCRF_PREC_FASTLIN = 3 };

/* Mirrors CRF_FeatureMap_config (ftrmaps/CRF_FeatureMap.h:24-47) plus the model fields
 * CRFTrain sets on CRF_Model (CRFTrain/src/Main.cpp:539-597). */
typedef struct scrf_config {
  uint32_t abi_version;     /* SCRF_ABI_VERSION */
  uint32_t model_type;      /* scrf_model_type */
  uint32_t map_type;        /* scrf_map_type */
  uint32_t num_labs;        /* numLabs == nActualLabs (crf_label_size) */
  uint32_t num_feas;        /* numFeas: floats per window (joined streams) */
  uint32_t num_states;      /* crf_states: 1, or K > 1 (n-state topology over num_labs = phones * K labels) with SCRF_STDFRAME or
                               SCRF_STDSEG_NO_DUR_NO_(SEG)TRANSFTR; the latter needs use_trans_bias, keeps the hook shapes of the
                               one-state model and refuses scrf_grad_device_ptr / scrf_set_grad_buffer */
  uint32_t lab_max_dur;     /* label_maximum_duration; 1 for STDFRAME */
  int32_t use_state_ftrs;
  uint32_t state_fidx_start, state_fidx_end; /* inclusive */
  int32_t use_trans_ftrs;
  uint32_t trans_fidx_start, trans_fidx_end; /* inclusive */
  int32_t use_state_bias, use_trans_bias;
  double state_bias_val, trans_bias_val;
  int32_t device_id;        /* HIP device ordinal */
  uint32_t train_precision; /* scrf_precision for scrf_fb_* (decode is always EXACT) */
  uint64_t scratch_bytes;   /* per-chunk device scratch budget, 0 = default (8 GiB) */
} scrf_config;

/* One input stream of CRFTrain (ftr1/ftr2/ftr3 flags, CRFTrain/src/Main.cpp:146-256):
 * raw frames plus the window recipe of io/CRF_InFtrStream_SeqMultiWindow.cpp.  The frame
 * array of an utterance has (T + left_ctx + right_ctx) rows: the caller supplies the
 * context padding (the demo reads it from a padded pfile). */
typedef struct scrf_stream_recipe {
  uint32_t in_width;        /* floats per raw frame */
  uint32_t left_ctx, right_ctx;
  int32_t extract_seg_ftr;  /* 1: [5 samples,avg,max,min,one-hot dur]; 0: boundary context */
} scrf_stream_recipe;

#define SCRF_MAX_STREAMS 3

/* One utterance.  Either `windows` (materialised window vectors, any feature pipeline) or
 * `frames[s]` for every stream of the batch recipe.  Host pointers. */
typedef struct scrf_utt {
  uint32_t T;                             /* frames */
  const float* windows;                   /* [N_seg(T)][num_feas] or NULL */
  const float* frames[SCRF_MAX_STREAMS];  /* [(T+lctx+rctx)][in_width] per stream, or NULL */
  const uint32_t* labels;                 /* [T] or NULL (decode only) */
} scrf_utt;

/* StdArc as OpenFST lays it out: int32 ilabel, olabel; float weight; int32 nextstate, plus
 * the source state the reference passes to AddArc. */
typedef struct scrf_arc {
  int32_t src, ilabel, olabel;
  float w;
  int32_t dst;
} scrf_arc;

typedef struct scrf_engine_s* scrf_handle;
typedef struct scrf_batch_s* scrf_batch;

/* ---- lifetime ------------------------------------------------------------------------ */
/* replaces: new CRF_Model + CRF_FeatureMap::createFeatureMap (CRF_Model.cpp:75,
 * ftrmaps/CRF_FeatureMap.cpp, CRFTrain/src/Main.cpp:539-597) */
int scrf_create(const scrf_config* cfg, scrf_handle* out);
int scrf_destroy(scrf_handle h);
const char* scrf_last_error(scrf_handle h); /* h may be NULL: error of the last failed create */
/* run everything on this hipStream_t (e.g. torch's current stream); NULL = engine's own */
int scrf_set_stream(scrf_handle h, void* hip_stream);
int scrf_synchronize(scrf_handle h);

/* ---- lambda layout (parity hooks for CRF_StdFeatureMap::recalc/get*Idx, :421-517) ------ */
int scrf_lambda_len(scrf_handle h, uint32_t* n);
int scrf_num_state_funcs(scrf_handle h, uint32_t* n);
int scrf_num_trans_funcs(scrf_handle h, uint32_t* n);
int scrf_state_idx(scrf_handle h, uint32_t clab, uint32_t fno, uint32_t* idx);
int scrf_trans_idx(scrf_handle h, uint32_t plab, uint32_t clab, uint32_t fno, uint32_t* idx);

/* ---- model state (CRF_Model::setLambda/getLambda/getLambdaAcc/getGradSqrAcc) ----------- */
int scrf_set_lambda(scrf_handle h, const double* lambda, uint32_t n);
int scrf_get_lambda(scrf_handle h, double* lambda, uint32_t n);
int scrf_set_lambda_acc(scrf_handle h, const double* v, uint32_t n);
int scrf_get_lambda_acc(scrf_handle h, double* v, uint32_t n);
int scrf_set_grad_sqr_acc(scrf_handle h, const double* v, uint32_t n);
int scrf_get_grad_sqr_acc(scrf_handle h, double* v, uint32_t n);

/* ---- batches: utterances resident in HBM ------------------------------------------------ */
/* replaces: CRF_FeatureStream views + CRF_InFtrStream_SeqMultiWindow / CRF_InLabStream_
 * SeqMultiWindow (io/, rows 25-27): packs and uploads n utterances once; all compute entry
 * points then run from HBM.  n_streams/recipes describe the frame inputs (ignored for
 * utterances given as windows). */
int scrf_batch_create(scrf_handle h, const scrf_utt* utts, uint32_t n, uint32_t n_streams,
                      const scrf_stream_recipe* recipes, scrf_batch* out);
int scrf_batch_destroy(scrf_handle h, scrf_batch b);
int scrf_batch_info(scrf_handle h, scrf_batch b, uint32_t* n_utts, uint64_t* n_frames,
                    uint64_t* n_segs, uint64_t* n_arcs);
/* Which training path the batch takes under the handle's precision:
 *   0  general (window vectors materialised, dense contractions over all of their columns);
 *   1  the FAST / decode kernels synthesise stream 0's windows on the fly (a segment-recipe stream without context whose
 *      shape fits the kernels' LDS budget; transition features, if any, on further streams);
 *   2  the same with the linear window average (SCRF_PREC_FASTLIN on a shape its kernels take);
 *   3  hybrid: one such stream with more labels than the fused kernels take -- windows materialised without their sampled
 *      blocks, which go through per-frame projections and per-frame sums.
 * Nonzero = no full window image in HBM. */
int scrf_batch_is_fused(scrf_handle h, scrf_batch b, int* fused);

/* ---- hot path: forward-backward + gradient ------------------------------------------------ */
/* replaces: CRF_NewGradBuilder_StdSeg_NoDur_NoTrans::buildGradient (trainers/gradbuilders/
 * ...NoDur_NoTrans.cpp:65-492; frame model: CRF_NewGradBuilder.cpp:48-382) for every
 * utterance of the batch.  The engine's device gradient accumulates (+=) the sum over
 * utterances of (observed - expected) counts; numer[u]/zx[u] (host, may be NULL) receive the
 * per-utterance numerator and log-partition (caller forms logLi = numer - zx,
 * trainers/CRF_SGTrainer.cpp:274).
 * Errors are reported by THIS call, also without numer/zx: it returns once the batch's recursion and
 * posterior kernels have finished and their per-utterance status has reached the host (the expected-count
 * kernels are still in flight then; the call is asynchronous from there on).  SCRF_ERR_BAD_LABEL /
 * SCRF_ERR_NUMERIC name the first failed utterance in scrf_last_error, as the reference's exceptions do, and
 * a failed batch contributes NOTHING to the gradient or the batch sums: its gradient is built in a staging
 * buffer and committed by a kernel that is a no-op when an utterance failed.  The training path runs a
 * scaled linear-domain recursion; if that raises SCRF_ERR_NUMERIC (a frame vector flushed
 * to zero: scores spread over more than ~700 nats) the batch is redone automatically with the log-domain
 * kernels, which follow the reference's LogMath, and only their verdict is reported (scrf_train_stats
 * counts these batches). */
int scrf_fb_batch(scrf_handle h, scrf_batch b, double* numer, double* zx);
/* batches scrf_fb_batch had to redo through the log-domain recursion since scrf_create */
int scrf_train_stats(scrf_handle h, uint64_t* n_lin_fallback);
int scrf_zero_grad(scrf_handle h);
int scrf_get_grad(scrf_handle h, double* grad, uint32_t n);      /* device -> host copy */
int scrf_add_grad(scrf_handle h, const double* grad, uint32_t n); /* grad += host vector */
int scrf_grad_device_ptr(scrf_handle h, void** dptr);            /* for an external all-reduce */
/* accumulate into caller-owned device memory of lambda_len doubles (e.g. a tensor that
 * torch.distributed all-reduces over RCCL); NULL restores the engine's own buffer */
int scrf_set_grad_buffer(scrf_handle h, void* dptr);
/* sums over the batch kept on device: {sum numer, sum zx, n_utts} */
int scrf_get_batch_sums(scrf_handle h, double* sums3);
/* The same in two halves, for a trainer that does not want to stop at every minibatch: `queue` copies the sums to a
 * pinned image behind the work issued so far (before the next scrf_zero_grad resets them), `take` waits for that copy
 * only.  One image: a second `queue` before `take` supersedes the first. */
int scrf_queue_batch_sums(scrf_handle h);
int scrf_take_batch_sums(scrf_handle h, double* sums3);

/* ---- parity hooks: the node accessors of nodes/CRF_StateNode.h:67-115 ---------------------- */
/* getStateValue(lab,dur) / getTransValue(p,c): S[N_seg][L], M[T][L*L] of utterance u (EXACT).  For
 * SCRF_STDSEG_NO_DUR M is [N_seg][L*L]: getTransValue(p,c,dur) of every window, the rows of utterance-initial
 * windows (no predecessor) zero (nodes/CRF_StdSegStateNode_WithoutDurLab.cpp:69-110, :570-580).  For SCRF_STDSEG
 * (La = num_labs / lab_max_dur) S is [N_seg][La] -- row (t,dur), phone = the node's stateArray[(dur-1)*La + phone] -- and
 * M is [N_seg][num_labs][La] = transMatrix[plab*num_labs + clab] (nodes/CRF_StdSegStateNode.cpp:83-127).  With num_states
 * = K > 1 on SCRF_STDFRAME (P = num_labs / K) S is [T][num_labs] and M is [T][2*num_labs + P*P]: the node's diagTransMatrix | offDiagTransMatrix
 * (entry c = transition c -> c+1) | denseTransMatrix (entry p*P + q = end state of phone p -> start state of phone q).
 * With num_states = K > 1 on SCRF_STDSEG_NO_DUR_NO_(SEG)TRANSFTR the shapes are the one-state model's; M holds -1e30 (log 0)
 * for the transitions the topology lacks (the reference's getTransValue throws there) */
int scrf_scores(scrf_handle h, scrf_batch b, uint32_t u, double* S, double* M);
/* window synthesis of utterance u: [N_seg][num_feas] (io/CRF_InFtrStream_SeqMultiWindow.cpp) */
int scrf_windows(scrf_handle h, scrf_batch b, uint32_t u, float* windows);
/* getAlpha / alphaArray_WithDur / getBeta / computeAlphaSum from the log-domain recursion over the EXACT
 * scores (`prec` must be SCRF_PREC_EXACT: the hook exists to compare node values with the reference's):
 * alpha_dur [N_seg][L], alpha [T][L], beta [T][L] (any may be NULL).  SCRF_STDSEG: alpha_dur and beta receive the nodes'
 * alpha / beta over full labels as [N_seg][La]; `alpha` is not written */
int scrf_forward_backward(scrf_handle h, scrf_batch b, uint32_t u, uint32_t prec,
                          double* alpha_dur, double* alpha, double* beta, double* zx);

/* ---- decode -------------------------------------------------------------------------------- */
/* replaces: buildLattice<StdArc> (decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab_
 * WithoutSegTransFtr.h:30-407; frame model decoders/CRF_LatticeBuilder.h:97-204): arcs in
 * the reference's AddArc order, weights float(-double).  Call with arcs==NULL to get counts.
 * num_states = K > 1: nStateBuildLattice (decoders/CRF_LatticeBuilder.h:715-840 for SCRF_STDFRAME; ...StdSeg_WithoutDurLab_
 * WithoutSegTransFtr.h:409-690 for the segmental model): the topology's arcs only, in that function's AddArc order */
int scrf_lattice_arcs(scrf_handle h, scrf_batch b, uint32_t u, int norm, scrf_arc* arcs,
                      uint64_t* n_arcs, uint32_t* n_states, int32_t* final_state);
/* replaces: buildLattice + ShortestPath/Project/RmEpsilon/TopSort + (olabel-1)
 * (CRFFstDecode/src/Main.cpp:804-889) for every utterance: seg_labels receives the labels
 * back to back, lab_off[u]..lab_off[u+1] (lab_off has n_utts+1 entries), best_cost[u] the
 * float path weight.  max_labels = capacity of seg_labels (sum of T is always enough). */
int scrf_viterbi_batch(scrf_handle h, scrf_batch b, uint32_t* seg_labels, uint64_t max_labels,
                       uint64_t* lab_off, float* best_cost);
/* scrf_viterbi_batch on a batch of raw frames takes its float arc weights from the fused fp64-MFMA
 * score kernel and recomputes, in the reference's operation order, every weight whose float
 * rounding a rounding-error bound cannot guarantee -- so labels and costs equal
 * the EXACT path's bit for bit.  Counters since scrf_create: weights recomputed, chunks whose
 * list overflowed and went through the EXACT path.  SCRF_FAST_DECODE=0 disables the fast path. */
int scrf_decode_stats(scrf_handle h, uint64_t* n_recomputed, uint64_t* n_fallback_chunks);
/* Posterior-mass self-checks of the training path with the FRAME model's bounds ([0.9, 1.1], nodes/CRF_StdStateNode.cpp
 * :252-275) on an engine whose model type is segmental (1e-6, ...WithoutSegTransFtr.cpp:917-947).  For a host that runs
 * the n-state frame model as the n-state segmental model with maximum duration 1 (the same function): an
 * utterance the reference's frame node accepts must not abort training because the stricter segmental check is applied. */
int scrf_set_frame_mass_check(scrf_handle h, int on);

/* ---- minibatch reduce + optimizer ------------------------------------------------------------ */
/* replaces the join/sum/average of CRF_Minibatch_GradAccumulator::accumulateGradient
 * (trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:277-312): all-reduce (sum) of the
 * device gradient and of {numer, zx, n_utts, active} over the communicator's ranks, then
 * grad /= number of active ranks.  Without a communicator (single GPU) it only divides by
 * `active` (1).  sums4 (host, may be NULL) = reduced {numer, zx, n_utts, n_active}. */
int scrf_comm_unique_id(void* id128);                       /* ncclGetUniqueId, 128 bytes */
int scrf_comm_init(scrf_handle h, const void* id128, int rank, int n_ranks);
int scrf_allreduce_grad(scrf_handle h, int active, double* sums4);
/* the same with up to 4 caller scalars riding in the scalar all-reduce (summed over the ranks): a C++ host
 * sends its "my stream is exhausted" flag this way, whose sum decides the end of the epoch (:248-250, :312).
 * extra_out (host, n_extra doubles) and sums4 are read back together; either may be NULL. */
int scrf_allreduce_grad_ex(scrf_handle h, int active, const double* extra_in, uint32_t n_extra, double* sums4,
                           double* extra_out);
/* scrf_fb_batch followed by scrf_allreduce_grad_ex, as ONE call -- for a host whose rank runs exactly one batch per step
 * (rank r = stream r of CRF_Minibatch_GradAccumulator).  With transition features (the TIMIT demo: 4.37 M weights, 98.7 % of
 * them transition weights) the collective runs in two blocks, transition weights then state weights + scalars, and inside
 * this call the transition contraction comes first so that its block is all-reduced on a second stream UNDER the state
 * contraction (SCRF_COMM_OVERLAP=0: after it; results identical).  The block sequence is the same in
 * scrf_allreduce_grad[_ex], so ranks that call either (an exhausted rank has no batch) still match.
 * A failure of the batch itself does not stop the collective: *fb_status gets scrf_fb_batch's code, extra[fail_slot] is
 * raised and `active` cleared before the scalars are reduced (the peers see the summed flag), scrf_last_error keeps the
 * batch's message.  The return value is the collective's (errors as described below). */
int scrf_fb_batch_allreduce(scrf_handle h, scrf_batch b, int active, const double* extra_in, uint32_t n_extra,
                            uint32_t fail_slot, double* sums4, double* extra_out, int* fb_status);
/* Failure behaviour of the collective (the reference's join is a pthread_join and cannot hang on a dead peer; a
 * collective can).  With a communicator, scrf_allreduce_grad[_ex] waits for completion under a watchdog: it polls the
 * stream and ncclCommGetAsyncError and gives up after SCRF_COMM_TIMEOUT_S seconds (default 300) -- in both cases, and
 * on any error of its own on the way into the collective, it aborts the communicator (ncclCommAbort) and returns
 * SCRF_ERR_COMM / SCRF_ERR_HIP.  scrf_comm_abort does the same on request: a host whose rank cannot enter the
 * collective calls it before exiting non-zero, so that its peers fail fast instead of waiting for the timeout.  After
 * an abort the handle has no communicator; a restart is a fresh process. */
int scrf_comm_abort(scrf_handle h);
/* per-step collectives run under a communicator since scrf_create, and how many of them had their transition block
 * all-reduced early, under the state contraction (scrf_fb_batch_allreduce) */
int scrf_comm_stats(scrf_handle h, uint64_t* n_collectives, uint64_t* n_overlapped);
/* the reference's Gaussian-prior step as written (trainers/CRF_SGTrainer.cpp:300-303): grad[i] -= grad[i] *
 * inv_square_var on the device gradient (it scales the gradient, not lambda -- kept as is) */
int scrf_gauss_prior(scrf_handle h, float inv_square_var);
/* grad *= s  (used with an external all-reduce: s = 1/n_active) */
int scrf_scale_grad(scrf_handle h, double s);
/* grad[i] /= d on the device: the `/ nStreams_active` of accumulateGradient (:306-308) for a host that
 * runs several streams on ONE device and keeps the summed gradient there */
int scrf_div_grad(scrf_handle h, double d);
/* replaces CRF_SGTrainer::sgtrainMinibatch's update (trainers/CRF_SGTrainer.cpp:299-325):
 * lambda += lr*g | AdaGrad; lambdaAcc += lambda; g = 0. */
int scrf_sgd_step(scrf_handle h, double lr_or_eta, int use_adagrad, double eps);

/* ---- measurement ------------------------------------------------------------------------------ */
/* HIP-event time (ms) of the phases of the last scrf_fb_batch / scrf_viterbi_batch on the
 * engine stream: [0] windows, [1] scores, [2] forward-backward+posteriors, [3] expected
 * counts (ExpF GEMM), [4] reduce, [5] viterbi, [6] whole call; n_launch[i] = kernel launches.
 * [7] [8] [9] time single kernels inside phases 1-3, events recorded on the stream the kernel is
 * launched on: the state score kernel, the forward/backward recursion, the state expected-count
 * kernel (the three that dominate a step). */
#define SCRF_N_PHASES 10
int scrf_last_timing(scrf_handle h, float* ms, uint32_t* n_launch);
/* HIP-event time of every kernel of the last timed scrf_fb_batch / scrf_viterbi_batch, one line per kernel
 * name: "name\tmilliseconds\tlaunches\n" (events recorded on the stream the kernel is launched on) */
int scrf_kernel_timing(scrf_handle h, char* buf, size_t cap);
int scrf_enable_timing(scrf_handle h, int on);

#ifdef __cplusplus
}
#endif
#endif /* SCRF_ABI_H_ */
