/*
 * scrf_oracle.h -- CPU restatement of ASR-CRaFT's segmental-CRF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported baseline.  The product (asr-craft_amd/) never
 * links or calls it.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   a1  (CRF_LogMath)        PINNED: bit-checked against the reference's own
 *                            CRF_LogMath.cpp compiled from /root/reference into
 *                            oracle/_ref/ (tests/test_oracle_logmath.py).
 *   a2..a19 (everything else) PARITY UNPINNED: the reference's remaining sources
 *                            need QuickNet3 / OpenFST / cblas headers that this image
 *                            lacks, the reference ships no tests or golden outputs,
 *                            so these functions are a line-by-line restatement of the
 *                            cited reference code (same loop order, fp64, unfused
 *                            multiply-then-add) cross-checked by brute-force
 *                            enumeration and finite differences, not by the
 *                            reference binary.
 *
 * All paths "file:line" below are relative to /root/reference/CRF/src/.
 * Build: -O2 -ffp-contract=off (the reference builds -O2 without -march, i.e. no FMA
 * contraction; configure.ac:12).
 */
#ifndef SCRF_ORACLE_H_
#define SCRF_ORACLE_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_LAB_BAD 0xffffffffu /* CRF_LAB_BAD, io/CRF_FeatureStream.h */

/* status codes (the reference throws std::runtime_error; we return codes) */
enum {
  ORC_OK = 0,
  ORC_ERR_LOG_ZERO = 1,     /* utils/CRF_LogMath.cpp:195 */
  ORC_ERR_LOG_NAN = 2,      /* utils/CRF_LogMath.cpp:200 */
  ORC_ERR_EXP_OVERFLOW = 3, /* utils/CRF_LogMath.cpp:214,219 */
  ORC_ERR_PROB_SUM = 4,     /* nodes/...WithoutSegTransFtr.cpp:917-947, nodes/CRF_StdStateNode.cpp:252-275 */
  ORC_ERR_BAD_LABEL = 5,    /* nodes/...WithoutSegTransFtr.cpp:627-631 */
  ORC_ERR_CONFIG = 6,
  ORC_ERR_EMPTY = 7         /* trainers/gradbuilders/...NoDur_NoTrans.cpp:331-335 */
};

/* modeltype, CRF.h:50 */
enum {
  ORC_STDFRAME = 0,
  ORC_STDSEG = 1,
  ORC_STDSEG_NO_DUR = 2,
  ORC_STDSEG_NO_DUR_NO_TRANSFTR = 3,
  ORC_STDSEG_NO_DUR_NO_SEGTRANSFTR = 4
};

/* Mirror of CRF_FeatureMap_config (ftrmaps/CRF_FeatureMap.h:24-47) + model fields
 * of CRF_Model (CRF_Model.h). */
typedef struct {
  uint32_t model_type;
  uint32_t num_labs;      /* numLabs == nActualLabs for the NO_DUR models and STDFRAME */
  uint32_t lab_max_dur;   /* D; 1 for STDFRAME */
  uint32_t num_feas;      /* floats per window vector (nFtrsPerSeg) */
  int32_t use_state_ftrs;
  uint32_t state_fidx_start, state_fidx_end; /* inclusive */
  int32_t use_trans_ftrs;
  uint32_t trans_fidx_start, trans_fidx_end; /* inclusive */
  int32_t use_state_bias, use_trans_bias;
  double state_bias_val, trans_bias_val;
  uint32_t num_states;    /* numStates (0 and 1 both mean one state per label); > 1: the n-state topology of
                             nodes/CRF_StdNStateNode.cpp over num_labs = phones * num_states labels, STDFRAME only */
} orc_config;

/* lambda index layout (ftrmaps/CRF_StdFeatureMap.cpp:280-320,355-410,472-517) */
typedef struct {
  uint32_t num_state_funcs; /* numStateFuncs */
  uint32_t num_trans_funcs; /* numTransFuncs */
  uint32_t lambda_len;      /* numFtrFuncs */
  uint32_t* state_idx;      /* [L]   stateFeatureIdxCache */
  uint32_t* trans_idx;      /* [L*L] transFeatureIdxCache[plab*L+clab] */
} orc_layout;

/* ---- a1: CRF_LogMath ---------------------------------------------------- */
extern const double ORC_LOG0; /* -DBL_MAX, utils/CRF_LogMath.h:26 */
double orc_expE(double a, int* err);
double orc_logE(double a, int* err);
double orc_logadd2(double a, double b, int* err);
double orc_logadd_n(const double* R, int n, int* err);
double orc_logadd_max_n(const double* R, double max, int n, int* err);

/* ---- a6: layout ---------------------------------------------------------- */
int orc_layout_init(const orc_config* cfg, orc_layout* lay);
void orc_layout_free(orc_layout* lay);

/* ---- a2..a5: feature map ------------------------------------------------- */
double orc_state_value(const orc_config* cfg, const orc_layout* lay, const float* x,
                       const double* lambda, uint32_t clab);
double orc_trans_value(const orc_config* cfg, const orc_layout* lay, const float* x,
                       const double* lambda, uint32_t plab, uint32_t clab);
double orc_state_expf(const orc_config* cfg, const orc_layout* lay, const float* x,
                      const double* lambda, double* ExpF, double* grad, double alpha_beta,
                      uint32_t t_clab, uint32_t clab);
double orc_trans_expf(const orc_config* cfg, const orc_layout* lay, const float* x,
                      const double* lambda, double* ExpF, double* grad, double alpha_beta,
                      uint32_t t_plab, uint32_t t_clab, uint32_t plab, uint32_t clab);

/* ---- inputs: segment windows and labels (rows 26/27, adjacent) ------------ */
/* number of windows ending at frame t, and total over an utterance */
uint32_t orc_node_max_dur(uint32_t t, uint32_t D);
uint64_t orc_num_segs(uint32_t T, uint32_t D);
uint64_t orc_seg_base(uint32_t t, uint32_t D); /* index of window d=1 of frame t */

/* width of one output window, io/CRF_InFtrStream_SeqMultiWindow.cpp:50-125 */
uint32_t orc_window_width(uint32_t in_width, uint32_t max_win_len, uint32_t lctx, uint32_t rctx,
                          int extract_seg_ftr);
/* Windows of all frames of one utterance, in read() order: for t, for d=1..min(t+1,D).
 * frames: [(T + lctx + rctx)][in_width], i.e. already padded by the caller with lctx
 * leading and rctx trailing frames (the reference reads those from a padded pfile,
 * demo/segmental-timit-demo.cfg.in:60).  out: [N_seg][out_stride], writes `width`
 * floats per window at column out_col. io/CRF_InFtrStream_SeqMultiWindow.cpp:325-470. */
void orc_windows(const float* frames, uint32_t T, uint32_t in_width, uint32_t D, uint32_t lctx,
                 uint32_t rctx, int extract_seg_ftr, float* out, uint32_t out_stride,
                 uint32_t out_col);
/* frame labels -> per-end-frame segment labels L*(dur-1)+phone or ORC_LAB_BAD;
 * io/CRF_InLabStream_SeqMultiWindow.cpp:51-110,119-183 + gradbuilder :216-231. */
void orc_group_labels(const uint32_t* frame_labs, uint32_t T, uint32_t D, uint32_t L,
                      uint32_t* seg_labels);

/* ---- a7..a12: segmental model #9 ----------------------------------------- */
/* S: [N_seg][L] (stateArray of node t at rows seg_base(t)..), M: [T][L*L] (transMatrix
 * of node t, p*L+l). segftrs: [N_seg][num_feas]. */
void orc_seg_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                    const float* segftrs, uint32_t T, double* S, double* M);
/* alpha_dur: [N_seg][L] (value for (t,d,l) at (seg_base(t)+d-1)*L+l; the reference
 * stores it transposed per node as alphaArray_WithDur[l*nodeMaxDur+d-1]),
 * alpha: [T][L], apt: [T][L] (alphaPlusTrans of node t, computed with M[t+1]; row T-1 unused) */
int orc_seg_forward(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                    double* alpha_dur, double* alpha, double* apt, double* Zx);
/* beta: [T][L], sd: [T][L] (tmpBetaArray_nextBetasPlusNextStateValue_sumOverDur; row T-1 unused) */
int orc_seg_backward(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                     double* beta, double* sd);
/* Full CRF_NewGradBuilder_StdSeg_NoDur_NoTrans::buildGradient: grad += counts - ExpF,
 * returns numerator via *numer, Zx via *Zx. labels: [T]. */
int orc_seg_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                           const float* segftrs, const uint32_t* labels, uint32_t T,
                           double* grad, double* numer, double* Zx);
/* the same with the worker's arrays kept across utterances (bench_cpu.c: one workspace per thread, like the reference's
 * per-thread CRF_StateVector and builder, nodes/CRF_StateVector.cpp:37-67); a zeroed struct is an empty workspace */
#define ORC_WS_SLOTS 10
typedef struct { void* p[ORC_WS_SLOTS]; size_t cap[ORC_WS_SLOTS]; } orc_workspace;
void orc_workspace_free(orc_workspace* w);
int orc_seg_build_gradient_ws(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                              const float* segftrs, const uint32_t* labels, uint32_t T,
                              double* grad, double* numer, double* Zx_out, orc_workspace* w);
/* posteriors for tests: gamma [N_seg][L], xi [T][L*L] (row T-1 unused) */
int orc_seg_posteriors(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                       double* gamma, double* xi, double* Zx);

/* ---- f3: n-state frame model (nodes/CRF_StdNStateNode.cpp, decoders/CRF_LatticeBuilder.h nStateBuildLattice):
 * cfg->num_states = K > 1, P = num_labs / K phones; label c is state c % K of phone c / K.  Transitions allowed: self
 * (c -> c), next state inside a phone (c-1 -> c), end state of any phone -> start state of any phone.
 * S [T][nLabs]; TD [T][nLabs] self transitions; TO [T][nLabs] entry c = transition c -> c+1 (unused for end states);
 * TE [T][P*P] entry p*P + q = end state of phone p -> start state of phone q.  The layout (orc_layout_init with
 * num_states > 1) marks transitions outside the topology with trans_idx = 0xffffffff. */
void orc_nstate_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* ftrs, uint32_t T,
                       double* S, double* TD, double* TO, double* TE);
int orc_nstate_forward(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                       uint32_t T, double* alpha, double* Zx);
int orc_nstate_backward(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                        uint32_t T, double* beta);
int orc_nstate_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* ftrs,
                              const uint32_t* labels, uint32_t T, double* grad, double* numer, double* Zx_out);
/* ---- f3: STDSEG (duration-labelled; cfg->num_labs = nLabs = nActualLabs * lab_max_dur) ------------------------
 * S, alpha, beta: [N_seg][nActualLabs] (row (t,dur), phone = the node's entry (dur-1)*nActualLabs + phone);
 * MX: [N_seg][nLabs][nActualLabs] = transMatrix[plab*nLabs + clab] of the window's node */
void orc_stdseg_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* segftrs,
                       uint32_t T, double* S, double* MX);
int orc_stdseg_forward(const orc_config* cfg, const double* S, const double* MX, uint32_t T, double* alpha, double* Zx);
int orc_stdseg_backward(const orc_config* cfg, const double* S, const double* MX, uint32_t T, double* beta);
int orc_stdseg_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* segftrs,
                              const uint32_t* labels, uint32_t T, double* grad, double* numer, double* Zx_out);
/* ---- f3: STDSEG_NO_DUR, segment-dependent transition features ----------------
 * nodes/CRF_StdSegStateNode_WithoutDurLab.cpp (+ trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp,
 * which passes the PREVIOUS label).  The transition score of a segment depends on its own window:
 * M2[(t,d)][p*L+l] = computeTransMatrixValue(window d of node t, p, l) for the durations that have a
 * predecessor node (d <= numPrevNodes(t)); rows of utterance-initial segments are unused (zero).
 *   alpha_d[t][l][d] = LSE_p(alpha[t-d][p] + M2[(t,d)][p][l]) + S[(t,d)][l]        (:132-190)
 *   beta[t][c]       = LSE_{d,l}(M2[(t+d,d)][c][l] + beta[t+d][l] + S[(t+d,d)][l])  (:248-310)
 *   gamma, xi[(t,d)][p][l] = exp(alpha[t-d][p] + M2 + S + beta[t][l] - Zx)         (:455-500)
 * S [N_seg][L], M2 [N_seg][L*L], alpha_dur [N_seg][L], alpha / beta [T][L]. */
void orc_segtrans_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                         const float* segftrs, uint32_t T, double* S, double* M2);
int orc_segtrans_forward(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                         double* alpha_dur, double* alpha, double* Zx);
int orc_segtrans_backward(const orc_config* cfg, const double* S, const double* M2, uint32_t T, double* beta);
int orc_segtrans_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                                const float* segftrs, const uint32_t* labels, uint32_t T,
                                double* grad, double* numer, double* Zx);
/* posteriors for tests: gamma [N_seg][L], xi [N_seg][L*L] (rows of initial segments zero) */
int orc_segtrans_posteriors(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                            double* gamma, double* xi, double* Zx);

/* ---- a15: frame-level chain ----------------------------------------------- */
int orc_frame_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                             const float* ftrs, const uint32_t* labels, uint32_t T,
                             double* grad, double* numer, double* Zx);

/* ---- a13/a14: minibatch accumulation + optimizer --------------------------- */
/* grad = (sum_s sgrad[s]) / n_active, trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:296-308 */
void orc_minibatch_reduce(const double* sgrad, uint32_t n_streams, const int32_t* active,
                          uint32_t lambda_len, double* grad);
/* trainers/CRF_SGTrainer.cpp:299-327 (useGvar off) */
void orc_sgd_step(double* lambda, double* lambda_acc, double* grad_sqr_acc, double* grad,
                  uint32_t n, double lr_or_eta, int use_adagrad, double eps);

/* ---- a16..a18: lattice + best path ---------------------------------------- */
typedef struct {
  int32_t src, ilabel, olabel;
  float w;
  int32_t dst;
} orc_arc;
uint64_t orc_seg_lattice_num_arcs(uint32_t T, uint32_t L, uint32_t D);
uint64_t orc_seg_lattice_num_arcs_k(uint32_t T, uint32_t L, uint32_t D, uint32_t K);
uint32_t orc_seg_lattice_num_states(uint32_t T, uint32_t L);
/* arcs in chronological AddArc order; decoders/...WithoutSegTransFtr.h:30-407 */
/* norm=0: final arcs carry (float)(-0.0); norm!=0: (float)(-(-alpha_sum)) as :371,397 */
uint64_t orc_seg_lattice_arcs(const orc_config* cfg, const double* S, const double* M,
                              uint32_t T, int norm, double alpha_sum, orc_arc* arcs,
                              uint32_t* n_states, int32_t* final_state);
/* STDSEG_NO_DUR: decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h (S [N_seg][L], M2 [N_seg][L*L]) */
uint64_t orc_segtrans_lattice_num_arcs(uint32_t T, uint32_t L, uint32_t D);
uint64_t orc_nstate_lattice_num_arcs(uint32_t T, uint32_t nLabs, uint32_t K);
uint64_t orc_nstate_lattice_arcs(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                                 uint32_t T, int norm, double alpha_sum, orc_arc* arcs, uint32_t* n_states, int32_t* final_state);
uint64_t orc_stdseg_lattice_num_arcs(uint32_t T, uint32_t La, uint32_t D);
uint64_t orc_stdseg_lattice_arcs(const orc_config* cfg, const double* S, const double* MX, uint32_t T, int norm,
                                 double alpha_sum, orc_arc* arcs, uint32_t* n_states, int32_t* final_state);
uint64_t orc_segtrans_lattice_arcs(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                                   int norm, double alpha_sum, orc_arc* arcs, uint32_t* n_states,
                                   int32_t* final_state);
uint64_t orc_frame_lattice_num_arcs(uint32_t T, uint32_t L);
/* decoders/CRF_LatticeBuilder.h:97-204; norm=0: final arcs carry +0.0f, else (float)(-alpha_sum) */
uint64_t orc_frame_lattice_arcs(const orc_config* cfg, const double* S, const double* M,
                                uint32_t T, int norm, double alpha_sum, orc_arc* arcs,
                                uint32_t* n_states, int32_t* final_state);
/* ShortestPath(n=1) + Project(OUTPUT) + RmEpsilon + TopSort + (olabel-1):
 * CRFFstDecode/src/Main.cpp:838-889.  Float tropical semiring, relaxation in state
 * order, arcs of a state in insertion order, strict-improvement update (first relaxed
 * wins ties).  OpenFST is absent from the image: tie-breaking is parity-unpinned.
 * Returns number of labels written (<= max_out), or -1 if no path. */
int64_t orc_best_path(const orc_arc* arcs, uint64_t n_arcs, uint32_t n_states, int32_t start,
                      int32_t final_state, uint32_t* out_labels, uint64_t max_out,
                      float* best_cost);

/* ---- a19: CRFDecode's Viterbi decoder, free phone loop ------------------------------------ */
/* decoders/CRF_ViterbiDecoder_StdSeg_NoSegTransFtr.cpp with lm_fst == NULL (createFreePhoneLmFst
 * :1270-1350), one state per phone, no pruning (input_beam <= 0), in the reference's own "push"
 * form: at node t every surviving (phone) hypothesis of node t-1 is expanded across LM arcs to
 * every OTHER phone (crossStateTransUpdate :435-500, weight old + float(-M[t][p][l]), marks a phone
 * start) and then to itself (internalStateTransUpdate :246-300, old + float(-M[t][l][l])); each
 * expansion enters the hypothesis lists of nodes t..t+D-1 under the key (phone, dur), the smaller
 * weight winning with strict < (CRF_ViterbiNode::addNonEpsVtbState); stateValueUpdate (:116-160)
 * adds float(-S[t][dur][l]); choose_nState_Best_Seg keeps per phone the best duration in list
 * order (earliest-entered, i.e. longest, first; strict <).  Node 0 is entered from the LM start
 * state with weight 0 (:442-445).  Backtrace :2204-2290: one arc per segment, weight
 * float(-S) for the first and float(-(M + S)) taken from the segment's END node for the others
 * (getFullTransValue :2262), olabel phone+1 where the phone starts, else 0.
 * The decoder itself cannot be compiled here (OpenFST absent): parity UNPINNED, ties included.
 * S [N_seg][L], M [T][L][L].  Outputs hold up to T segments.  Returns 0, or -1 if T == 0. */
int orc_free_phone_decode(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                          uint32_t* seg_phone, uint32_t* seg_dur, float* seg_weight,
                          int32_t* seg_phone_start, uint32_t* n_segs, float* best_weight);

#ifdef __cplusplus
}
#endif
/* ---- bench_cpu.c: the reference's threaded accumulator as the CPU baseline (a13; bench.py's cpu_baseline leg) ---- */
void orc_bench_set_cpus(const int* cpus, int n);   /* worker s is pinned to cpus[s % n]; n = 0: not pinned */
void orc_bench_phases(double* out5);               /* featLoad, transMat, alpha, beta, expF of the last call, us over all workers */
int orc_bench_fb(const orc_config* cfg, const double* lambda, const float* frames, const uint32_t* labels,
                 const uint64_t* frame_off, uint32_t U, uint32_t in_width, uint32_t n_threads, double* grad_out,
                 double* numer, double* zx, double* seconds);
/* with a second stream of in_width2-wide frames padded by ctx2 frames each side (no segment recipe: its window is the
 * 2 ctx2 + 1 frames around the node), joined behind the first stream's columns */
int orc_bench_fb2(const orc_config* cfg, const double* lambda, const float* frames, const float* frames2,
                  uint32_t in_width2, uint32_t ctx2, const uint32_t* labels, const uint64_t* frame_off, uint32_t U,
                  uint32_t in_width, uint32_t n_threads, double* grad_out, double* numer, double* zx, double* seconds);

#endif /* SCRF_ORACLE_H_ */
