/*
 * scrf_oracle.c -- CPU restatement of ASR-CRaFT's segmental-CRF hot path (plain C99).
 *
 * TEST INFRASTRUCTURE ONLY (see scrf_oracle.h for the parity status of each part).
 * Every function cites the reference file:line (relative to /root/reference/CRF/src/)
 * whose arithmetic and loop order it follows.  Canonical arithmetic: fp64,
 * float->double promotion of features, unfused multiply then add, ascending feature
 * index, bias last, logAdd = max-shift with the sum in index order.
 *
 * Build with -O2 -ffp-contract=off.
 */
#include "scrf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

const double ORC_LOG0 = -1 * DBL_MAX; /* utils/CRF_LogMath.h:26 */

static void set_err(int* err, int code) {
  if (err && *err == ORC_OK) *err = code;
}

/* ------------------------------------------------------------------ a1 ---- */

/* utils/CRF_LogMath.cpp:211-224 */
double orc_expE(double a, int* err) {
  const double ln_max = log(DBL_MAX); /* CRF_DBL_LN_MAX, CRF_LogMath.h:25 */
  if (a >= ln_max) {
    set_err(err, ORC_ERR_EXP_OVERFLOW);
    return NAN;
  }
  double b = exp(a);
  if (isnan(b) || isinf(b)) {
    set_err(err, ORC_ERR_EXP_OVERFLOW);
    return NAN;
  }
  return b;
}

/* utils/CRF_LogMath.cpp:192-205 */
double orc_logE(double a, int* err) {
  if (a == 0) {
    set_err(err, ORC_ERR_LOG_ZERO);
    return NAN;
  }
  double b = log(a);
  if (isnan(b) || isinf(b)) {
    set_err(err, ORC_ERR_LOG_NAN);
    return NAN;
  }
  return b;
}

/* utils/CRF_LogMath.cpp:41-64 */
double orc_logadd2(double loga, double logb, int* err) {
  double logx = loga;
  double logy = logb;
  if (logy > logx) {
    logy = loga;
    logx = logb;
  }
  double negDiff = logy - logx;
  return logx + orc_logE(1.0 + orc_expE(negDiff, err), err);
}

/* utils/CRF_LogMath.cpp:102-125 */
double orc_logadd_max_n(const double* R, double max, int n, int* err) {
  double sum = 0.0;
  for (int i = 0; i < n; i++) sum += orc_expE(R[i] - max, err);
  double lsum = orc_logE(sum, err);
  return max + lsum;
}

/* utils/CRF_LogMath.cpp:69-96 */
double orc_logadd_n(const double* R, int n, int* err) {
  double max = R[0];
  for (int i = 1; i < n; i++) {
    if (R[i] > max) max = R[i];
  }
  return orc_logadd_max_n(R, max, n, err);
}

/* ------------------------------------------------------------------ a6 ---- */

/* ftrmaps/CRF_StdFeatureMap.cpp:472-517 (recalc), :280-320, :355-410 with numStates==1 */
int orc_layout_init(const orc_config* cfg, orc_layout* lay) {
  const uint32_t L = cfg->num_labs;
  const uint32_t K = cfg->num_states > 1 ? cfg->num_states : 1;
  memset(lay, 0, sizeof(*lay));
  if (L == 0) return ORC_ERR_CONFIG;
  const uint32_t P = L / K; /* numActualLabels :474 */
  if (P * K != L) return ORC_ERR_CONFIG; /* "Invalid state/label combination" :476-479 */
  /* :480-485: end->start transitions + diagonal self transitions + off-diagonal transitions */
  uint32_t trans_mult = K == 1 ? L * L : P * P + L + L - P;
  uint32_t nff = 0, nsf = 0, ntf = 0;
  if (cfg->use_state_ftrs) {
    if (cfg->state_fidx_end < cfg->state_fidx_start) return ORC_ERR_CONFIG;
    nff += (cfg->state_fidx_end - cfg->state_fidx_start + 1) * L;
    nsf += (cfg->state_fidx_end - cfg->state_fidx_start + 1);
  }
  if (cfg->use_state_bias) {
    nff += L;
    nsf += 1;
  }
  if (cfg->use_trans_ftrs) {
    if (cfg->trans_fidx_end < cfg->trans_fidx_start) return ORC_ERR_CONFIG;
    nff += (cfg->trans_fidx_end - cfg->trans_fidx_start + 1) * trans_mult;
    ntf += (cfg->trans_fidx_end - cfg->trans_fidx_start + 1);
  }
  if (cfg->use_trans_bias) {
    nff += trans_mult;
    ntf += 1;
  }
  lay->num_state_funcs = nsf;
  lay->num_trans_funcs = ntf;
  lay->lambda_len = nff;
  lay->state_idx = (uint32_t*)malloc(sizeof(uint32_t) * L);
  lay->trans_idx = (uint32_t*)malloc(sizeof(uint32_t) * L * L);
  if (!lay->state_idx || !lay->trans_idx) return ORC_ERR_CONFIG;
  if (K == 1) {
    for (uint32_t clab = 0; clab < L; clab++) {
      /* computeStateFeatureIdx :280-320 */
      lay->state_idx[clab] = (clab == 0) ? 0 : clab * (nsf + L * ntf);
      for (uint32_t plab = 0; plab < L; plab++) {
        /* computeTransFeatureIdx :365 */
        lay->trans_idx[plab * L + clab] = clab * (nsf + L * ntf) + nsf + plab * ntf;
      }
    }
    return ORC_OK;
  }
  /* n-state blocks (:293-312): a label's state functions, its self transition, then for a phone's start state the
   * transitions from every phone's end state, for the other states the one from the state before */
  uint32_t at = 0;
  for (uint32_t clab = 0; clab < L; clab++) {
    lay->state_idx[clab] = at;
    at += nsf + ((clab % K == 0) ? (P + 1) * ntf : 2 * ntf);
  }
  for (uint32_t clab = 0; clab < L; clab++)
    for (uint32_t plab = 0; plab < L; plab++) { /* computeTransFeatureIdx :367-407 */
      uint32_t v = lay->state_idx[clab] + nsf;
      if (plab != clab) {
        v += ntf;
        if (clab % K == 0) {
          if ((plab + 1) % K != 0) v = 0xffffffffu;
          else v += (plab / K) * ntf;
        } else if (plab != clab - 1) {
          v = 0xffffffffu;
        }
      }
      lay->trans_idx[plab * L + clab] = v;
    }
  return ORC_OK;
}

void orc_layout_free(orc_layout* lay) {
  free(lay->state_idx);
  free(lay->trans_idx);
  memset(lay, 0, sizeof(*lay));
}

/* ------------------------------------------------------------- a2 .. a5 ---- */

/* ftrmaps/CRF_StdFeatureMap.cpp:65-81 */
double orc_state_value(const orc_config* cfg, const orc_layout* lay, const float* x,
                       const double* lambda, uint32_t clab) {
  double v = 0.0;
  uint32_t lc = lay->state_idx[clab];
  if (cfg->use_state_ftrs) {
    for (uint32_t f = cfg->state_fidx_start; f <= cfg->state_fidx_end; f++) {
      v += x[f] * lambda[lc];
      lc++;
    }
  }
  if (cfg->use_state_bias) {
    v += lambda[lc] * cfg->state_bias_val;
    lc++;
  }
  return v;
}

/* ftrmaps/CRF_StdFeatureMap.cpp:94-110 */
double orc_trans_value(const orc_config* cfg, const orc_layout* lay, const float* x,
                       const double* lambda, uint32_t plab, uint32_t clab) {
  double v = 0.0;
  uint32_t lc = lay->trans_idx[plab * cfg->num_labs + clab];
  if (cfg->use_trans_ftrs) {
    for (uint32_t f = cfg->trans_fidx_start; f <= cfg->trans_fidx_end; f++) {
      v += x[f] * lambda[lc];
      lc++;
    }
  }
  if (cfg->use_trans_bias) {
    v += lambda[lc] * cfg->trans_bias_val;
    lc++;
  }
  return v;
}

/* ftrmaps/CRF_StdFeatureMap.cpp:130-175 (compute_grad == true) */
double orc_state_expf(const orc_config* cfg, const orc_layout* lay, const float* x,
                      const double* lambda, double* ExpF, double* grad, double alpha_beta,
                      uint32_t t_clab, uint32_t clab) {
  double logLi = 0.0;
  uint32_t lc = lay->state_idx[clab];
  if (cfg->use_state_ftrs) {
    for (uint32_t f = cfg->state_fidx_start; f <= cfg->state_fidx_end; f++) {
      ExpF[lc] += alpha_beta * x[f];
      if (t_clab == clab) {
        grad[lc] += x[f];
        logLi += lambda[lc] * x[f];
      }
      lc++;
    }
  }
  if (cfg->use_state_bias) {
    ExpF[lc] += alpha_beta * cfg->state_bias_val;
    if (t_clab == clab) {
      grad[lc] += cfg->state_bias_val;
      logLi += lambda[lc] * cfg->state_bias_val;
    }
    lc++;
  }
  return logLi;
}

/* ftrmaps/CRF_StdFeatureMap.cpp:197-223 (compute_grad == true) */
double orc_trans_expf(const orc_config* cfg, const orc_layout* lay, const float* x,
                      const double* lambda, double* ExpF, double* grad, double alpha_beta,
                      uint32_t t_plab, uint32_t t_clab, uint32_t plab, uint32_t clab) {
  double logLi = 0.0;
  uint32_t lc = lay->trans_idx[plab * cfg->num_labs + clab];
  if (cfg->use_trans_ftrs) {
    for (uint32_t f = cfg->trans_fidx_start; f <= cfg->trans_fidx_end; f++) {
      ExpF[lc] += alpha_beta * x[f];
      if ((clab == t_clab) && (plab == t_plab)) {
        grad[lc] += x[f];
        logLi += lambda[lc] * x[f];
      }
      lc++;
    }
  }
  if (cfg->use_trans_bias) {
    ExpF[lc] += alpha_beta * cfg->trans_bias_val;
    if ((clab == t_clab) && (plab == t_plab)) {
      grad[lc] += cfg->trans_bias_val;
      logLi += lambda[lc] * cfg->trans_bias_val;
    }
    lc++;
  }
  return logLi;
}

/* ------------------------------------------------- windows and labels ------- */

/* trainers/gradbuilders/CRF_NewGradBuilder_StdSeg_NoDur_NoTrans.cpp:243-251 */
uint32_t orc_node_max_dur(uint32_t t, uint32_t D) { return (t + 1 <= D) ? t + 1 : D; }

uint64_t orc_seg_base(uint32_t t, uint32_t D) {
  if (t < D) return (uint64_t)t * (t + 1) / 2;
  return (uint64_t)D * (D + 1) / 2 + (uint64_t)(t - D) * D;
}

uint64_t orc_num_segs(uint32_t T, uint32_t D) { return orc_seg_base(T, D); }

/* io/CRF_InFtrStream_SeqMultiWindow.cpp:50-125 (boundary-delta variant out of scope) */
uint32_t orc_window_width(uint32_t in_width, uint32_t max_win_len, uint32_t lctx, uint32_t rctx,
                          int extract_seg_ftr) {
  if (max_win_len == 1) return (lctx + 1 + rctx) * in_width; /* :53 */
  if (extract_seg_ftr) return 8 * in_width + max_win_len + (lctx + rctx) * in_width; /* :77-78 */
  return (lctx + 1 + rctx) * in_width; /* :114 */
}

/* io/CRF_InFtrStream_SeqMultiWindow.cpp:413-455 and the helpers it calls:
 * first_frame_left_ctx_ftrs :897-925, sample_ftrs :556-598, avg_ftrs :609-646,
 * max_ftrs :657-697, min_ftrs :708-748, dur_ftrs :850-884, first_frame_ftrs :1017-1038,
 * first_frame_right_ctx_ftrs :937-965, last_frame_right_ctx_ftrs :977-1005. */
void orc_windows(const float* frames, uint32_t T, uint32_t in_width, uint32_t D, uint32_t lctx,
                 uint32_t rctx, int extract_seg_ftr, float* out, uint32_t out_stride,
                 uint32_t out_col) {
  const uint32_t W = in_width;
  float* acc_sum = (float*)malloc(sizeof(float) * W);
  float* acc_max = (float*)malloc(sizeof(float) * W);
  float* acc_min = (float*)malloc(sizeof(float) * W);
  uint64_t row = 0;
  for (uint32_t t = 0; t < T; t++) {
    const uint32_t avail = orc_node_max_dur(t, D);
    const float* last = frames + (size_t)(lctx + t) * W; /* current (last) frame */
    for (uint32_t j = 0; j < W; j++) {
      acc_sum[j] = 0.0f;
      acc_max[j] = last[j];
      acc_min[j] = last[j];
    }
    for (uint32_t w = 1; w <= avail; w++, row++) {
      const float* first = last - (size_t)(w - 1) * W; /* first frame of the window */
      float* o = out + row * out_stride + out_col;
      /* left context of the window's first frame */
      const float* lbase = first - (size_t)lctx * W;
      for (uint32_t c = 0; c < lctx; c++)
        for (uint32_t j = 0; j < W; j++) *o++ = lbase[(size_t)c * W + j];
      if (D == 1 || !extract_seg_ftr) {
        for (uint32_t j = 0; j < W; j++) *o++ = first[j];
      } else {
        /* sample frames at 10/30/50/70/90 % (:566-580) */
        float one_tenth_win_len = (float)(w * 0.1);
        for (int i = 1; i < 10; i += 2) {
          float prod = one_tenth_win_len * (float)i;
          uint32_t frameStep = (uint32_t)ceilf(prod) - 1;
          const float* s = first + (size_t)frameStep * W;
          for (uint32_t j = 0; j < W; j++) *o++ = s[j];
        }
        /* running sum / max / min from the last frame backwards (:621-631,:669-682,:720-733) */
        for (uint32_t j = 0; j < W; j++) {
          acc_sum[j] += first[j];
          *o++ = acc_sum[j] / (float)w;
        }
        for (uint32_t j = 0; j < W; j++) {
          if (first[j] > acc_max[j]) acc_max[j] = first[j];
          *o++ = acc_max[j];
        }
        for (uint32_t j = 0; j < W; j++) {
          if (first[j] < acc_min[j]) acc_min[j] = first[j];
          *o++ = acc_min[j];
        }
        for (uint32_t k = 1; k <= D; k++) *o++ = (k == w) ? 1.0f : 0.0f;
      }
      /* right context: after the LAST frame for segment features (:449-451),
       * after the FIRST frame otherwise (:453) */
      const float* rbase = (extract_seg_ftr) ? last : first;
      for (uint32_t c = 1; c <= rctx; c++)
        for (uint32_t j = 0; j < W; j++) *o++ = rbase[(size_t)c * W + j];
    }
  }
  free(acc_sum);
  free(acc_max);
  free(acc_min);
}

/* io/CRF_InLabStream_SeqMultiWindow.cpp:51-110 (groupLabels), :141-180 (run detection),
 * :284-299 (label emitted at the window's end frame), gradbuilder :216-231 (phone-dur id) */
void orc_group_labels(const uint32_t* frame_labs, uint32_t T, uint32_t D, uint32_t L,
                      uint32_t* seg_labels) {
  for (uint32_t t = 0; t < T; t++) seg_labels[t] = ORC_LAB_BAD;
  uint32_t start = 0;
  while (start < T) {
    uint32_t lab = frame_labs[start];
    uint32_t next = start + 1;
    while (next < T && frame_labs[next] == lab) next++;
    if (lab != ORC_LAB_BAD) {
      uint32_t dur = next - start;
      if (dur <= D) {
        seg_labels[next - 1] = L * (dur - 1) + lab;
      } else {
        uint32_t numPieces = (dur % D == 0) ? dur / D : dur / D + 1;
        uint32_t pieceDur = dur / numPieces;
        uint32_t remainder = dur % numPieces;
        uint32_t pieceStart = start;
        for (uint32_t r = 0; r < numPieces; r++) {
          uint32_t pd = (r < remainder) ? pieceDur + 1 : pieceDur;
          uint32_t pieceEnd = pieceStart + pd - 1;
          seg_labels[pieceEnd] = L * (pd - 1) + lab;
          pieceStart = pieceEnd + 1;
        }
      }
    }
    start = next;
  }
}

/* ------------------------------------------------------------ a7 .. a12 ---- */

/* nodes/CRF_StdSegStateNode_WithoutDurLab_WithoutSegTransFtr.cpp:39-114 for every node */
void orc_seg_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                    const float* segftrs, uint32_t T, double* S, double* M) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur, F = cfg->num_feas;
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const float* ftrBuf = segftrs + base * F;
    double* Mt = M + (size_t)t * L * L;
    for (uint32_t lab = 0; lab < L; lab++)
      for (uint32_t plab = 0; plab < L; plab++)
        Mt[plab * L + lab] = (cfg->num_states > 1 && lay->trans_idx[plab * L + lab] == 0xffffffffu)
                                 ? 0.0 /* no such transition in the n-state topology: never read */
                                 : orc_trans_value(cfg, lay, ftrBuf, lambda, plab, lab);
    const uint32_t nd = orc_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++) {
      const float* x = ftrBuf + (size_t)(dur - 1) * F;
      double* Sd = S + (base + dur - 1) * L;
      for (uint32_t lab = 0; lab < L; lab++) Sd[lab] = orc_state_value(cfg, lay, x, lambda, lab);
    }
  }
}

static uint32_t num_prev(uint32_t t, uint32_t D) { return (t + 1 <= D) ? t : D; } /* gradbuilder :258-266 */

/* n-state topology (cfg->num_states = K > 1, nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp): the labels a
 * transition INTO `lab` may come from, in the order the node visits them -- itself, then every phone's end state
 * ascending (lab a start state) or the state before (:1177-1213); and the labels a transition OUT OF `lab` may go to
 * -- itself, then every phone's start state ascending (lab an end state) or the state after (:215-234).
 * Returns the count; K <= 1: all labels ascending. */
static uint32_t ns_preds(const orc_config* cfg, uint32_t lab, uint32_t* out) {
  const uint32_t L = cfg->num_labs, K = cfg->num_states;
  uint32_t n = 0;
  if (K <= 1) { for (uint32_t p = 0; p < L; p++) out[n++] = p; return n; }
  out[n++] = lab;
  if (lab % K == 0) { for (uint32_t e = K - 1; e < L; e += K) out[n++] = e; }
  else out[n++] = lab - 1;
  return n;
}
static int ns_allowed(const orc_config* cfg, uint32_t from, uint32_t to) {
  const uint32_t K = cfg->num_states;
  if (K <= 1 || from == to) return 1;
  if (to % K == 0) return (from + 1) % K == 0;
  return from + 1 == to;
}
static uint32_t ns_succs(const orc_config* cfg, uint32_t lab, uint32_t* out) {
  const uint32_t L = cfg->num_labs, K = cfg->num_states;
  uint32_t n = 0;
  if (K <= 1) { for (uint32_t c = 0; c < L; c++) out[n++] = c; return n; }
  out[n++] = lab;
  if ((lab + 1) % K == 0) { for (uint32_t b = 0; b < L; b += K) out[n++] = b; }
  else out[n++] = lab + 1;
  return n;
}

/* computeFirstAlpha :335-382, computeAlphaPlusTrans :1077-1108, computeAlpha :123-245,
 * computeAlphaSum nodes/CRF_StdSegStateNode.cpp:447-462 */
int orc_seg_forward(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                    double* alpha_dur, double* alpha, double* apt, double* Zx) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  uint32_t accn = (L > D ? L : D) + 2;
  double* tmp = (double*)malloc(sizeof(double) * accn);
  uint32_t* plist = (uint32_t*)malloc(sizeof(uint32_t) * (L + 2));
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    if (t == 0) {
      for (uint32_t lab = 0; lab < L; lab++) {
        alpha_dur[lab] = S[lab];
        alpha[lab] = S[lab];
      }
      continue;
    }
    /* node t-1: alphaPlusTrans with node t's transition matrix (:156) */
    const double* Mt = M + (size_t)t * L * L;
    const double* aprev = alpha + (size_t)(t - 1) * L;
    for (uint32_t next_lab = 0; next_lab < L; next_lab++) {
      const uint32_t np_ = ns_preds(cfg, next_lab, plist);
      for (uint32_t i = 0; i < np_; i++) tmp[i] = aprev[plist[i]] + Mt[plist[i] * L + next_lab];
      apt[(size_t)(t - 1) * L + next_lab] = orc_logadd_n(tmp, (int)np_, &err);
    }
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    for (uint32_t clab = 0; clab < L; clab++) {
      uint32_t id = 0;
      for (uint32_t dur = 1; dur <= np; dur++) {
        double v = apt[(size_t)(t - dur) * L + clab] + S[(base + dur - 1) * L + clab];
        tmp[id++] = v;
        alpha_dur[(base + dur - 1) * L + clab] = v;
      }
      for (uint32_t dur = np + 1; dur <= nd; dur++) {
        double v = S[(base + dur - 1) * L + clab];
        tmp[id++] = v;
        alpha_dur[(base + dur - 1) * L + clab] = v;
      }
      alpha[(size_t)t * L + clab] = orc_logadd_n(tmp, (int)id, &err);
    }
  }
  *Zx = orc_logadd_n(alpha + (size_t)(T - 1) * L, (int)L, &err);
  free(tmp); free(plist);
  return err;
}

/* one node of computeBeta :395-466 / setTailBeta nodes/CRF_StdStateNode.cpp:198-204 */
static void seg_beta_node(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                          uint32_t t, double* beta, double* sd, double* tmp, int* err) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D; /* gradbuilder :390-398 */
  double* bt = beta + (size_t)t * L;
  if (nn == 0) {
    for (uint32_t l = 0; l < L; l++) bt[l] = 0.0;
    return;
  }
  double* sdt = sd + (size_t)t * L;
  for (uint32_t nextlab = 0; nextlab < L; nextlab++) {
    uint32_t id = 0;
    for (uint32_t dur = 1; dur <= nn; dur++) {
      const uint64_t nb = orc_seg_base(t + dur, D);
      tmp[id++] = S[(nb + dur - 1) * L + nextlab] + beta[(size_t)(t + dur) * L + nextlab];
    }
    sdt[nextlab] = orc_logadd_n(tmp, (int)id, err);
  }
  const double* Mn = M + (size_t)(t + 1) * L * L;
  uint32_t* slist = (uint32_t*)malloc(sizeof(uint32_t) * (L + 2));
  for (uint32_t clab = 0; clab < L; clab++) {
    const uint32_t ns_ = ns_succs(cfg, clab, slist);
    for (uint32_t i = 0; i < ns_; i++) tmp[i] = Mn[clab * L + slist[i]] + sdt[slist[i]];
    bt[clab] = orc_logadd_n(tmp, (int)ns_, err);
  }
  free(slist);
}

int orc_seg_backward(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                     double* beta, double* sd) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  uint32_t accn = (L > D ? L : D) + 2;
  double* tmp = (double*)malloc(sizeof(double) * accn);
  for (uint32_t t = T; t-- > 0;) seg_beta_node(cfg, S, M, T, t, beta, sd, tmp, &err);
  free(tmp);
  return err;
}

/* CRF_NewGradBuilder_StdSeg_NoDur_NoTrans::buildGradient,
 * trainers/gradbuilders/CRF_NewGradBuilder_StdSeg_NoDur_NoTrans.cpp:65-492, with
 * computeExpF nodes/...WithoutSegTransFtr.cpp:616-949 inlined in its loop order. */
/* Phase timers of the reference's buildGradient (gradbuilder :155-157, :481-488: featLoadTime, transMatTime,
 * alphaTime, betaTime, expFTime, microseconds).  A thread that wants them points orc_phase_us at five doubles
 * (bench_cpu.c does; slot 0, the feature load, is the caller's); NULL = no timing. */
__thread double* orc_phase_us = NULL;
static double phase_now_us(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}
#define PHASE_BEGIN() double ph_t0_ = orc_phase_us ? phase_now_us() : 0.0
#define PHASE_END(i) do { if (orc_phase_us) { const double n_ = phase_now_us(); orc_phase_us[i] += n_ - ph_t0_; ph_t0_ = n_; } } while (0)

/* A worker's arrays, kept across utterances the way the reference keeps its node vector (nodes/CRF_StateVector.cpp:37-67:
 * set() resets an existing node and only constructs one when the utterance is longer than any before) and its builder
 * keeps ExpF (gradbuilder ctor :30-34, zeroed per utterance :99-101). */
static void* ws_take(void** p, size_t* cap, size_t need) {
  if (need > *cap) {
    free(*p);
    *p = malloc(need);
    *cap = *p ? need : 0;
  }
  return *p;
}
void orc_workspace_free(orc_workspace* w) {
  if (!w) return;
  for (int i = 0; i < ORC_WS_SLOTS; i++) { free(w->p[i]); w->p[i] = NULL; w->cap[i] = 0; }
}

int orc_seg_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                           const float* segftrs, const uint32_t* labels, uint32_t T,
                           double* grad, double* numer, double* Zx_out) {
  orc_workspace w;
  memset(&w, 0, sizeof(w));
  const int rc = orc_seg_build_gradient_ws(cfg, lay, lambda, segftrs, labels, T, grad, numer, Zx_out, &w);
  orc_workspace_free(&w);
  return rc;
}

int orc_seg_build_gradient_ws(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                              const float* segftrs, const uint32_t* labels, uint32_t T,
                              double* grad, double* numer, double* Zx_out, orc_workspace* w) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur, F = cfg->num_feas;
  if (T == 0) return ORC_ERR_EMPTY;
  const uint64_t nseg = orc_num_segs(T, D);
  int err = ORC_OK;
  uint32_t accn = (L > D ? L : D) + 2;
  double* ExpF = (double*)ws_take(&w->p[0], &w->cap[0], sizeof(double) * lay->lambda_len);
  double* S = (double*)ws_take(&w->p[1], &w->cap[1], sizeof(double) * nseg * L);
  double* M = (double*)ws_take(&w->p[2], &w->cap[2], sizeof(double) * (size_t)T * L * L);
  double* ad = (double*)ws_take(&w->p[3], &w->cap[3], sizeof(double) * nseg * L);
  double* alpha = (double*)ws_take(&w->p[4], &w->cap[4], sizeof(double) * (size_t)T * L);
  double* apt = (double*)ws_take(&w->p[5], &w->cap[5], sizeof(double) * (size_t)T * L);
  double* beta = (double*)ws_take(&w->p[6], &w->cap[6], sizeof(double) * (size_t)T * L);
  double* sd = (double*)ws_take(&w->p[7], &w->cap[7], sizeof(double) * (size_t)T * L);
  double* tmp = (double*)ws_take(&w->p[8], &w->cap[8], sizeof(double) * accn);
  uint32_t* plist = (uint32_t*)ws_take(&w->p[9], &w->cap[9], sizeof(uint32_t) * (L + 2));
  if (!ExpF || !S || !M || !ad || !alpha || !apt || !beta || !sd || !tmp || !plist) return ORC_ERR_EMPTY;
  memset(ExpF, 0, sizeof(double) * lay->lambda_len); /* :99-101 */
  double logLi = 0.0, Zx = 0.0;

  PHASE_BEGIN();
  orc_seg_scores(cfg, lay, lambda, segftrs, T, S, M);       /* :284 */
  PHASE_END(1);
  err = orc_seg_forward(cfg, S, M, T, ad, alpha, apt, &Zx); /* :296,307,343 */
  PHASE_END(2);

  for (uint32_t t = T; t-- > 0 && err == ORC_OK;) { /* :388-469 */
    PHASE_END(4);   /* the expected-count part of the previous node (nothing before the first) */
    seg_beta_node(cfg, S, M, T, t, beta, sd, tmp, &err);
    PHASE_END(3);
    /* label of the next node that carries one (:436-444) */
    uint32_t next_lab = ORC_LAB_BAD;
    for (uint32_t u = t + 1; u < T; u++) {
      next_lab = labels[u];
      if (next_lab != ORC_LAB_BAD) break;
    }
    /* computeExpF :616-949 */
    uint32_t actualLab = labels[t], labDur = ORC_LAB_BAD;
    uint32_t actualNextLab = next_lab;
    if (actualLab != ORC_LAB_BAD) {
      if (actualLab >= L * D) { err = ORC_ERR_BAD_LABEL; break; }
      labDur = labels[t] / L + 1;
      actualLab = labels[t] % L;
    }
    if (actualNextLab != ORC_LAB_BAD) {
      if (actualNextLab >= L * D) { err = ORC_ERR_BAD_LABEL; break; }
      actualNextLab = next_lab % L;
    }
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D);
    const float* ftrBuf = segftrs + base * F;
    double ab_tot = 0.0, ab_trans_tot = 0.0, nodeLi = 0.0;
    for (uint32_t lab = 0; lab < L; lab++) {
      for (uint32_t dur = 1; dur <= nd; dur++) {
        const float* x = ftrBuf + (size_t)(dur - 1) * F;
        double ab = orc_expE(ad[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
        ab_tot += ab;
        int match = (lab == actualLab && dur == labDur);
        nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, match ? actualLab : ORC_LAB_BAD, lab);
      }
    }
    uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D;
    if (nn > 0) {
      const double* Mn = M + (size_t)(t + 1) * L * L;
      const float* xn = segftrs + orc_seg_base(t + 1, D) * F; /* next node's window 1 (:789) */
      for (uint32_t nl = 0; nl < L; nl++) {
        const uint32_t np_ = ns_preds(cfg, nl, plist);   /* n-state: itself, then ends / the state before (:318-353) */
        for (uint32_t i = 0; i < np_; i++) {
          const uint32_t clab = plist[i];
          double ab = orc_expE(alpha[(size_t)t * L + clab] + Mn[clab * L + nl] + sd[(size_t)t * L + nl] - Zx, &err);
          ab_trans_tot += ab;
          int match = (clab == actualLab && nl == actualNextLab);
          nodeLi += orc_trans_expf(cfg, lay, xn, lambda, ExpF, grad, ab,
                                   match ? actualLab : ORC_LAB_BAD,
                                   match ? actualNextLab : ORC_LAB_BAD, clab, nl);
        }
      }
    } else {
      ab_trans_tot = 1.0;
    }
    /* self checks :917-947 */
    if (ab_tot > 1.000001 || ab_tot < -0.000001 || ab_trans_tot > 1.000001 ||
        ab_trans_tot < -0.000001 || ab_tot - ab_trans_tot > 0.000001 ||
        ab_tot - ab_trans_tot < -0.000001) {
      /* the tail node sets trans_tot=1 while its state mass is 1 too, so the
       * equality check holds there as well */
      set_err(&err, ORC_ERR_PROB_SUM);
    }
    logLi += nodeLi; /* :453 */
  }
  PHASE_END(4);
  for (uint32_t i = 0; i < lay->lambda_len; i++) grad[i] -= ExpF[i]; /* :471-473 */
  *Zx_out = Zx;
  *numer = logLi;
  return err;
}

int orc_seg_posteriors(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                       double* gamma, double* xi, double* Zx_out) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  if (T == 0) return ORC_ERR_EMPTY;
  const uint64_t nseg = orc_num_segs(T, D);
  double* ad = (double*)malloc(sizeof(double) * nseg * L);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* apt = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* beta = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* sd = (double*)malloc(sizeof(double) * (size_t)T * L);
  double Zx = 0.0;
  int err = orc_seg_forward(cfg, S, M, T, ad, alpha, apt, &Zx);
  if (err == ORC_OK) err = orc_seg_backward(cfg, S, M, T, beta, sd);
  for (uint32_t t = 0; t < T && err == ORC_OK; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++)
      for (uint32_t lab = 0; lab < L; lab++)
        gamma[(base + dur - 1) * L + lab] =
            orc_expE(ad[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
    if (t + 1 < T) {
      const double* Mn = M + (size_t)(t + 1) * L * L;
      for (uint32_t c = 0; c < L; c++)
        for (uint32_t n = 0; n < L; n++)
          xi[(size_t)t * L * L + c * L + n] =
              (cfg->num_states > 1 && !ns_allowed(cfg, c, n)) ? 0.0 :
              orc_expE(alpha[(size_t)t * L + c] + Mn[c * L + n] + sd[(size_t)t * L + n] - Zx, &err);
    }
  }
  *Zx_out = Zx;
  free(ad); free(alpha); free(apt); free(beta); free(sd);
  return err;
}

/* ------------------------------------------------------------------ a15 ---- */

/* ======================================================================================
 * f3: STDSEG (nodes/CRF_StdSegStateNode.cpp): labels carry the duration, clab = (dur-1)*nActualLabs + phone;
 * cfg->num_labs is the FULL label count nLabs (the feature map and the weight layout are over full labels),
 * nActualLabs = nLabs / labMaxDur (:40).  Node values are kept per window row: alpha / beta / S at
 * [row(t,dur)][phone] is the node's entry clab; MX[row(t,dur)][plab][phone] is transMatrix[plab*nLabs + clab] with
 * plab a full label of node t-dur (entries with plab >= that node's numAvailLabs are never read).
 * ====================================================================================== */
static uint32_t stdseg_actual_labs(const orc_config* cfg) { return cfg->num_labs / cfg->lab_max_dur; }

void orc_stdseg_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* segftrs,
                       uint32_t T, double* S, double* MX) {
  /* computeTransMatrix :83-127 */
  const uint32_t NL = cfg->num_labs, D = cfg->lab_max_dur, L = stdseg_actual_labs(cfg), F = cfg->num_feas;
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++) {
      const float* x = segftrs + (base + dur - 1) * F;
      double* Mrow = MX + (base + dur - 1) * (size_t)NL * L;
      const uint32_t pavail = dur <= np ? L * orc_node_max_dur(t - dur, D) : 0;
      for (uint32_t lab = 0; lab < L; lab++) {
        const uint32_t clab = (dur - 1) * L + lab;
        S[(base + dur - 1) * L + lab] = orc_state_value(cfg, lay, x, lambda, clab);
        for (uint32_t plab = 0; plab < NL; plab++)
          Mrow[(size_t)plab * L + lab] = plab < pavail ? orc_trans_value(cfg, lay, x, lambda, plab, clab) : 0.0;
      }
    }
  }
}

int orc_stdseg_forward(const orc_config* cfg, const double* S, const double* MX, uint32_t T, double* alpha, double* Zx) {
  /* computeFirstAlpha :189-198, computeAlpha :136-180, computeAlphaSum :436-447 */
  const uint32_t NL = cfg->num_labs, D = cfg->lab_max_dur, L = stdseg_actual_labs(cfg);
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* acc = (double*)malloc(sizeof(double) * NL);
  for (uint32_t t = 0; t < T && err == ORC_OK; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    for (uint32_t dur = 1; dur <= np; dur++) {
      const uint64_t pbase = orc_seg_base(t - dur, D);
      const uint32_t pavail = L * orc_node_max_dur(t - dur, D);
      const double* pa = alpha + pbase * L;   /* the previous node's alpha over its full labels: rows are contiguous */
      const double* Mrow = MX + (base + dur - 1) * (size_t)NL * L;
      for (uint32_t lab = 0; lab < L; lab++) {
        acc[0] = pa[0] + Mrow[lab];
        double maxv = acc[0];
        for (uint32_t plab = 1; plab < pavail; plab++) {
          acc[plab] = pa[plab] + Mrow[(size_t)plab * L + lab];
          if (acc[plab] > maxv) maxv = acc[plab];
        }
        double v = orc_logadd_max_n(acc, maxv, (int)pavail, &err);
        v += S[(base + dur - 1) * L + lab];
        alpha[(base + dur - 1) * L + lab] = v;
      }
    }
    for (uint32_t dur = np + 1; dur <= nd; dur++)
      for (uint32_t lab = 0; lab < L; lab++) alpha[(base + dur - 1) * L + lab] = S[(base + dur - 1) * L + lab];
  }
  if (err == ORC_OK)
    *Zx = orc_logadd_n(alpha + orc_seg_base(T - 1, D) * L, (int)(L * orc_node_max_dur(T - 1, D)), &err);
  free(acc);
  return err;
}

int orc_stdseg_backward(const orc_config* cfg, const double* S, const double* MX, uint32_t T, double* beta) {
  /* computeBeta :211-260; setTailBeta nodes/CRF_StdStateNode.cpp:198-204 */
  const uint32_t NL = cfg->num_labs, D = cfg->lab_max_dur, L = stdseg_actual_labs(cfg);
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* tb = (double*)malloc(sizeof(double) * NL);
  double* acc = (double*)malloc(sizeof(double) * NL);
  {
    const uint64_t base = orc_seg_base(T - 1, D);
    for (uint32_t i = 0; i < L * orc_node_max_dur(T - 1, D); i++) beta[base * L + i] = 0.0;
  }
  for (uint32_t t = T - 1; t-- > 0 && err == ORC_OK;) {
    const uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D;
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t avail = L * orc_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= nn; dur++) {
      const uint64_t row = orc_seg_base(t + dur, D) + dur - 1;
      for (uint32_t lab = 0; lab < L; lab++) tb[(dur - 1) * L + lab] = beta[row * L + lab] + S[row * L + lab];
    }
    for (uint32_t clab = 0; clab < avail; clab++) {
      double maxv = MX[(orc_seg_base(t + 1, D) + 0) * (size_t)NL * L + (size_t)clab * L + 0] + tb[0];
      uint32_t n = 0;
      for (uint32_t dur = 1; dur <= nn; dur++) {
        const double* Mrow = MX + (orc_seg_base(t + dur, D) + dur - 1) * (size_t)NL * L;
        for (uint32_t lab = 0; lab < L; lab++) {
          acc[n] = Mrow[(size_t)clab * L + lab] + tb[n];
          if (acc[n] > maxv) maxv = acc[n];
          n++;
        }
      }
      beta[base * L + clab] = orc_logadd_max_n(acc, maxv, (int)n, &err);
    }
  }
  free(tb); free(acc);
  return err;
}

int orc_stdseg_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* segftrs,
                              const uint32_t* labels, uint32_t T, double* grad, double* numer, double* Zx_out) {
  /* trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp (shared with STDSEG_NO_DUR) + computeExpF :345-424 */
  const uint32_t NL = cfg->num_labs, D = cfg->lab_max_dur, L = stdseg_actual_labs(cfg), F = cfg->num_feas;
  if (T == 0) return ORC_ERR_EMPTY;
  const uint64_t nseg = orc_num_segs(T, D);
  int err = ORC_OK;
  double* ExpF = (double*)calloc(lay->lambda_len, sizeof(double));
  double* S = (double*)malloc(sizeof(double) * nseg * L);
  double* MX = (double*)malloc(sizeof(double) * nseg * (size_t)NL * L);
  double* alpha = (double*)malloc(sizeof(double) * nseg * L);
  double* beta = (double*)malloc(sizeof(double) * nseg * L);
  double logLi = 0.0, Zx = 0.0;
  orc_stdseg_scores(cfg, lay, lambda, segftrs, T, S, MX);
  err = orc_stdseg_forward(cfg, S, MX, T, alpha, &Zx);
  if (err == ORC_OK) err = orc_stdseg_backward(cfg, S, MX, T, beta);
  for (uint32_t t = T; t-- > 0 && err == ORC_OK;) {
    uint32_t prev_lab = ORC_LAB_BAD;   /* the nearest EARLIER node that carries a label */
    for (uint32_t u = t; u > 0; u--) {
      prev_lab = labels[u - 1];
      if (prev_lab != ORC_LAB_BAD) break;
    }
    const uint32_t label = labels[t];
    if (label != ORC_LAB_BAD && label >= NL) { err = ORC_ERR_BAD_LABEL; break; }
    if (prev_lab != ORC_LAB_BAD && prev_lab >= NL) { err = ORC_ERR_BAD_LABEL; break; }
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    double ab_tot = 0.0, ab_trans_tot = 0.0, nodeLi = 0.0;
    for (uint32_t dur = 1; dur <= nd; dur++) {
      const float* x = segftrs + (base + dur - 1) * F;
      const double* Mrow = MX + (base + dur - 1) * (size_t)NL * L;
      for (uint32_t lab = 0; lab < L; lab++) {
        const uint32_t clab = (dur - 1) * L + lab;
        const size_t at = (base + dur - 1) * L + lab;
        double ab = orc_expE(alpha[at] + beta[at] - Zx, &err);
        ab_tot += ab;
        nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, label, clab);
        if (dur > np) continue;
        const double* pa = alpha + orc_seg_base(t - dur, D) * L;
        const uint32_t pavail = L * orc_node_max_dur(t - dur, D);
        for (uint32_t plab = 0; plab < pavail; plab++) {
          ab = orc_expE(pa[plab] + Mrow[(size_t)plab * L + lab] + S[at] + beta[at] - Zx, &err);
          ab_trans_tot += ab;
          nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, prev_lab, label, plab, clab);
        }
      }
    }
    if (np == 0) ab_trans_tot = 1.0;
    if (ab_tot > 1.000001 || ab_tot < -0.000001 || ab_trans_tot > 1.000001 || ab_trans_tot < -0.000001)
      set_err(&err, ORC_ERR_PROB_SUM); /* :402-421 */
    logLi += nodeLi;
  }
  for (uint32_t i = 0; i < lay->lambda_len; i++) grad[i] -= ExpF[i];
  *Zx_out = Zx;
  *numer = logLi;
  free(ExpF); free(S); free(MX); free(alpha); free(beta);
  return err;
}

/* ======================================================================================
 * f3: STDSEG_NO_DUR (nodes/CRF_StdSegStateNode_WithoutDurLab.cpp)
 * ====================================================================================== */
void orc_segtrans_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                         const float* segftrs, uint32_t T, double* S, double* M2) {
  /* computeTransMatrix :69-110: for dur <= numPrevNodes state AND transition values of window dur,
   * for the remaining (utterance-initial) duration the state value only */
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur, F = cfg->num_feas;
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++) {
      const float* x = segftrs + (base + dur - 1) * F;
      double* Mrow = M2 + (base + dur - 1) * (size_t)L * L;
      for (uint32_t lab = 0; lab < L; lab++) {
        S[(base + dur - 1) * L + lab] = orc_state_value(cfg, lay, x, lambda, lab);
        for (uint32_t plab = 0; plab < L; plab++)
          Mrow[plab * L + lab] = dur <= np ? orc_trans_value(cfg, lay, x, lambda, plab, lab) : 0.0;
      }
    }
  }
}

int orc_segtrans_forward(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                         double* alpha_dur, double* alpha, double* Zx) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* acc = (double*)malloc(sizeof(double) * (L > D ? L : D));
  double* wd = (double*)malloc(sizeof(double) * D);
  for (uint32_t l = 0; l < L; l++) { alpha_dur[l] = S[l]; alpha[l] = S[l]; } /* computeFirstAlpha :200-208 */
  for (uint32_t t = 1; t < T && err == ORC_OK; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    for (uint32_t lab = 0; lab < L; lab++) { /* computeAlpha :132-190 */
      for (uint32_t dur = 1; dur <= np; dur++) {
        const double* pa = alpha + (size_t)(t - dur) * L;
        const double* Mrow = M2 + (base + dur - 1) * (size_t)L * L;
        acc[0] = pa[0] + Mrow[0 * L + lab];
        double maxv = acc[0];
        for (uint32_t plab = 1; plab < L; plab++) {
          acc[plab] = pa[plab] + Mrow[plab * L + lab];
          if (acc[plab] > maxv) maxv = acc[plab];
        }
        double v = orc_logadd_max_n(acc, maxv, (int)L, &err);
        v += S[(base + dur - 1) * L + lab];
        alpha_dur[(base + dur - 1) * L + lab] = v;
        wd[dur - 1] = v;
      }
      for (uint32_t dur = np + 1; dur <= nd; dur++) {
        alpha_dur[(base + dur - 1) * L + lab] = S[(base + dur - 1) * L + lab];
        wd[dur - 1] = S[(base + dur - 1) * L + lab];
      }
      alpha[(size_t)t * L + lab] = orc_logadd_n(wd, (int)nd, &err);
    }
  }
  if (err == ORC_OK) *Zx = orc_logadd_n(alpha + (size_t)(T - 1) * L, (int)L, &err); /* computeAlphaSum */
  free(acc); free(wd);
  return err;
}

int orc_segtrans_backward(const orc_config* cfg, const double* S, const double* M2, uint32_t T, double* beta) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* tb = (double*)malloc(sizeof(double) * (size_t)D * L);
  double* acc = (double*)malloc(sizeof(double) * (size_t)D * L);
  for (uint32_t l = 0; l < L; l++) beta[(size_t)(T - 1) * L + l] = 0.0; /* setTailBeta */
  for (uint32_t t = T - 1; t-- > 0 && err == ORC_OK;) { /* computeBeta :248-310 */
    const uint32_t nn = (T - 1 - t <= D) ? T - 1 - t : D;
    for (uint32_t dur = 1; dur <= nn; dur++) {
      const uint64_t row = orc_seg_base(t + dur, D) + dur - 1;
      for (uint32_t lab = 0; lab < L; lab++) tb[(dur - 1) * L + lab] = beta[(size_t)(t + dur) * L + lab] + S[row * L + lab];
    }
    for (uint32_t clab = 0; clab < L; clab++) {
      double maxv = M2[(orc_seg_base(t + 1, D) + 0) * (size_t)L * L + clab * L + 0] + tb[0];
      uint32_t n = 0;
      for (uint32_t dur = 1; dur <= nn; dur++) {
        const double* Mrow = M2 + (orc_seg_base(t + dur, D) + dur - 1) * (size_t)L * L;
        for (uint32_t lab = 0; lab < L; lab++) {
          acc[n] = Mrow[clab * L + lab] + tb[(dur - 1) * L + lab];
          if (acc[n] > maxv) maxv = acc[n];
          n++;
        }
      }
      beta[(size_t)t * L + clab] = orc_logadd_max_n(acc, maxv, (int)n, &err);
    }
  }
  free(tb); free(acc);
  return err;
}

int orc_segtrans_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                                const float* segftrs, const uint32_t* labels, uint32_t T,
                                double* grad, double* numer, double* Zx_out) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur, F = cfg->num_feas;
  if (T == 0) return ORC_ERR_EMPTY;
  const uint64_t nseg = orc_num_segs(T, D);
  int err = ORC_OK;
  double* ExpF = (double*)calloc(lay->lambda_len, sizeof(double));
  double* S = (double*)malloc(sizeof(double) * nseg * L);
  double* M2 = (double*)malloc(sizeof(double) * nseg * L * L);
  double* ad = (double*)malloc(sizeof(double) * nseg * L);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* beta = (double*)malloc(sizeof(double) * (size_t)T * L);
  double logLi = 0.0, Zx = 0.0;
  orc_segtrans_scores(cfg, lay, lambda, segftrs, T, S, M2);
  err = orc_segtrans_forward(cfg, S, M2, T, ad, alpha, &Zx);
  if (err == ORC_OK) err = orc_segtrans_backward(cfg, S, M2, T, beta);
  for (uint32_t t = T; t-- > 0 && err == ORC_OK;) {
    /* CRF_NewGradBuilder_StdSeg.cpp: the nearest EARLIER node that carries a label */
    uint32_t prev_lab = ORC_LAB_BAD;
    for (uint32_t u = t; u > 0; u--) {
      prev_lab = labels[u - 1];
      if (prev_lab != ORC_LAB_BAD) break;
    }
    /* computeExpF :422-560 */
    uint32_t actualLab = labels[t], labDur = ORC_LAB_BAD, actualPLab = prev_lab;
    if (actualLab != ORC_LAB_BAD) {
      if (actualLab >= L * D) { err = ORC_ERR_BAD_LABEL; break; }
      labDur = labels[t] / L + 1;
      actualLab = labels[t] % L;
    }
    if (actualPLab != ORC_LAB_BAD) {
      if (actualPLab >= L * D) { err = ORC_ERR_BAD_LABEL; break; }
      actualPLab = prev_lab % L;
    }
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    double ab_tot = 0.0, ab_trans_tot = 0.0, nodeLi = 0.0;
    for (uint32_t dur = 1; dur <= np; dur++) {
      const float* x = segftrs + (base + dur - 1) * F;
      const double* pa = alpha + (size_t)(t - dur) * L;
      const double* Mrow = M2 + (base + dur - 1) * (size_t)L * L;
      for (uint32_t lab = 0; lab < L; lab++) {
        double ab = orc_expE(ad[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
        ab_tot += ab;
        int match = (lab == actualLab && dur == labDur);
        nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, match ? actualLab : ORC_LAB_BAD, lab);
        for (uint32_t plab = 0; plab < L; plab++) {
          ab = orc_expE(pa[plab] + Mrow[plab * L + lab] + S[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
          ab_trans_tot += ab;
          match = (lab == actualLab && dur == labDur && plab == actualPLab);
          nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, match ? actualPLab : ORC_LAB_BAD,
                                   match ? actualLab : ORC_LAB_BAD, plab, lab);
        }
      }
    }
    for (uint32_t dur = np + 1; dur <= nd; dur++) {
      const float* x = segftrs + (base + dur - 1) * F;
      for (uint32_t lab = 0; lab < L; lab++) {
        double ab = orc_expE(ad[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
        ab_tot += ab;
        int match = (lab == actualLab && dur == labDur);
        nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, match ? actualLab : ORC_LAB_BAD, lab);
      }
    }
    if (np == 0) ab_trans_tot = 1.0;
    if (ab_tot > 1.000001 || ab_tot < -0.000001 || ab_trans_tot > 1.000001 || ab_trans_tot < -0.000001)
      set_err(&err, ORC_ERR_PROB_SUM); /* :530-545 (no state == trans check in this node) */
    logLi += nodeLi;
  }
  for (uint32_t i = 0; i < lay->lambda_len; i++) grad[i] -= ExpF[i];
  *Zx_out = Zx;
  *numer = logLi;
  free(ExpF); free(S); free(M2); free(ad); free(alpha); free(beta);
  return err;
}

int orc_segtrans_posteriors(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                            double* gamma, double* xi, double* Zx_out) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  const uint64_t nseg = orc_num_segs(T, D);
  int err = ORC_OK;
  double* ad = (double*)malloc(sizeof(double) * nseg * L);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* beta = (double*)malloc(sizeof(double) * (size_t)T * L);
  double Zx = 0.0;
  err = orc_segtrans_forward(cfg, S, M2, T, ad, alpha, &Zx);
  if (err == ORC_OK) err = orc_segtrans_backward(cfg, S, M2, T, beta);
  for (uint32_t t = 0; t < T && err == ORC_OK; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t nd = orc_node_max_dur(t, D), np = num_prev(t, D);
    for (uint32_t dur = 1; dur <= nd; dur++)
      for (uint32_t lab = 0; lab < L; lab++) {
        gamma[(base + dur - 1) * L + lab] = orc_expE(ad[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err);
        for (uint32_t plab = 0; plab < L; plab++)
          xi[(base + dur - 1) * (size_t)L * L + plab * L + lab] =
              dur <= np ? orc_expE(alpha[(size_t)(t - dur) * L + plab] + M2[(base + dur - 1) * (size_t)L * L + plab * L + lab] +
                                       S[(base + dur - 1) * L + lab] + beta[(size_t)t * L + lab] - Zx, &err)
                        : 0.0;
      }
  }
  *Zx_out = Zx;
  free(ad); free(alpha); free(beta);
  return err;
}

/* CRF_NewGradBuilder::buildGradient trainers/gradbuilders/CRF_NewGradBuilder.cpp:48-382 with
 * the CRF_StdStateNode methods nodes/CRF_StdStateNode.cpp:58-299 inlined. ftrs: [T][F]. */
int orc_frame_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda,
                             const float* ftrs, const uint32_t* labels, uint32_t T,
                             double* grad, double* numer, double* Zx_out) {
  const uint32_t L = cfg->num_labs, F = cfg->num_feas;
  if (T == 0) return ORC_ERR_EMPTY;
  int err = ORC_OK;
  double* ExpF = (double*)calloc(lay->lambda_len, sizeof(double));
  double* S = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* M = (double*)malloc(sizeof(double) * (size_t)T * L * L);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* beta = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* acc = (double*)malloc(sizeof(double) * L);
  double* tempBeta = (double*)malloc(sizeof(double) * L);
  double logLi = 0.0;
  for (uint32_t t = 0; t < T; t++) {
    const float* x = ftrs + (size_t)t * F;
    /* computeTransMatrix :58-72 */
    for (uint32_t clab = 0; clab < L; clab++) {
      S[(size_t)t * L + clab] = orc_state_value(cfg, lay, x, lambda, clab);
      for (uint32_t plab = 0; plab < L; plab++)
        M[(size_t)t * L * L + plab * L + clab] = orc_trans_value(cfg, lay, x, lambda, plab, clab);
    }
    if (t == 0) { /* computeFirstAlpha :118-128 */
      for (uint32_t clab = 0; clab < L; clab++) alpha[clab] = S[clab];
    } else { /* computeAlpha :81-110 */
      const double* pa = alpha + (size_t)(t - 1) * L;
      const double* Mt = M + (size_t)t * L * L;
      for (uint32_t clab = 0; clab < L; clab++) {
        acc[0] = pa[0] + Mt[0 + clab];
        double maxv = acc[0];
        for (uint32_t plab = 1; plab < L; plab++) {
          acc[plab] = pa[plab] + Mt[plab * L + clab];
          if (acc[plab] > maxv) maxv = acc[plab];
        }
        double a = orc_logadd_max_n(acc, maxv, (int)L, &err);
        a += S[(size_t)t * L + clab];
        alpha[(size_t)t * L + clab] = a;
      }
    }
  }
  double Zx = orc_logadd_n(alpha + (size_t)(T - 1) * L, (int)L, &err); /* :286-299 */
  for (uint32_t t = T; t-- > 0 && err == ORC_OK;) {
    double* bt = beta + (size_t)t * L;
    if (t == T - 1) {
      for (uint32_t c = 0; c < L; c++) bt[c] = 0.0; /* setTailBeta */
    } else { /* node t+1 ->computeBeta(beta of node t) :140-173 */
      const double* bn = beta + (size_t)(t + 1) * L;
      const double* Sn = S + (size_t)(t + 1) * L;
      const double* Mn = M + (size_t)(t + 1) * L * L;
      for (uint32_t c = 0; c < L; c++) tempBeta[c] = bn[c] + Sn[c];
      for (uint32_t plab = 0; plab < L; plab++) {
        acc[0] = Mn[plab * L + 0] + tempBeta[0];
        double maxv = acc[0];
        for (uint32_t c = 1; c < L; c++) {
          acc[c] = Mn[plab * L + c] + tempBeta[c];
          if (acc[c] > maxv) maxv = acc[c];
        }
        bt[plab] = orc_logadd_max_n(acc, maxv, (int)L, &err);
      }
    }
    /* computeExpF :221-277 */
    const float* x = ftrs + (size_t)t * F;
    const double* pa = (t > 0) ? alpha + (size_t)(t - 1) * L : NULL;
    uint32_t prev_lab = (t > 0) ? labels[t - 1] : L + 1; /* :299-307 */
    uint32_t label = labels[t];
    const double* Mt = M + (size_t)t * L * L;
    double ab_tot = 0.0, ab_trans_tot = 0.0, nodeLi = 0.0;
    for (uint32_t clab = 0; clab < L; clab++) {
      double ab = orc_expE(alpha[(size_t)t * L + clab] + bt[clab] - Zx, &err);
      ab_tot += ab;
      nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, label, clab);
      if (prev_lab > L) {
        ab_trans_tot = 1.0;
      } else {
        for (uint32_t plab = 0; plab < L; plab++) {
          ab = orc_expE(pa[plab] + Mt[plab * L + clab] + S[(size_t)t * L + clab] + bt[clab] - Zx, &err);
          ab_trans_tot += ab;
          nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, prev_lab, label, plab, clab);
        }
      }
    }
    if (ab_tot > 1.1 || ab_tot < 0.9 || ab_trans_tot > 1.1 || ab_trans_tot < 0.9)
      set_err(&err, ORC_ERR_PROB_SUM);
    logLi += nodeLi;
  }
  for (uint32_t i = 0; i < lay->lambda_len; i++) grad[i] -= ExpF[i];
  *Zx_out = Zx;
  *numer = logLi;
  free(ExpF); free(S); free(M); free(alpha); free(beta); free(acc); free(tempBeta);
  return err;
}

/* ------------------------------------------------------------ a13 / a14 ---- */

/* trainers/accumulators/CRF_Minibatch_GradAccumulator.cpp:217-219,296-298,306-308 */
void orc_minibatch_reduce(const double* sgrad, uint32_t n_streams, const int32_t* active,
                          uint32_t lambda_len, double* grad) {
  int n_active = 0;
  for (uint32_t i = 0; i < lambda_len; i++) grad[i] = 0.0;
  for (uint32_t s = 0; s < n_streams; s++) {
    if (!active[s]) continue;
    ++n_active;
    const double* g = sgrad + (size_t)s * lambda_len;
    for (uint32_t i = 0; i < lambda_len; ++i) grad[i] += g[i];
  }
  for (uint32_t i = 0; i < lambda_len; ++i) grad[i] /= n_active;
}

/* trainers/CRF_SGTrainer.cpp:299-325 (useGvar == false) */
void orc_sgd_step(double* lambda, double* lambda_acc, double* grad_sqr_acc, double* grad,
                  uint32_t n, double lr_or_eta, int use_adagrad, double eps) {
  for (uint32_t i = 0; i < n; i++) {
    if (use_adagrad) {
      grad_sqr_acc[i] += grad[i] * grad[i];
      lambda[i] += lr_or_eta / (sqrt(grad_sqr_acc[i]) + eps) * grad[i];
    } else {
      lambda[i] += lr_or_eta * grad[i];
    }
    lambda_acc[i] += lambda[i];
    grad[i] = 0.0;
  }
}

/* ------------------------------------------------------------ a16 .. a18 ---- */

uint64_t orc_seg_lattice_num_arcs(uint32_t T, uint32_t L, uint32_t D) {
  if (T == 0) return 0;
  return (uint64_t)(T - 1) * L * L + orc_num_segs(T, D) * L + L;
}
/* K states per phone: P*P + 2L - P boundary arcs per frame after the first (P = L / K) */
uint64_t orc_seg_lattice_num_arcs_k(uint32_t T, uint32_t L, uint32_t D, uint32_t K) {
  if (T == 0) return 0;
  if (K <= 1) return orc_seg_lattice_num_arcs(T, L, D);
  const uint64_t P = L / K;
  return (uint64_t)(T - 1) * (P * P + 2 * (uint64_t)L - P) + orc_num_segs(T, D) * L + L;
}
uint32_t orc_seg_lattice_num_states(uint32_t T, uint32_t L) {
  if (T == 0) return 1;
  return 1 + L + (T - 1) * 2 * L + 1;
}

/* decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab_WithoutSegTransFtr.h:30-407 */
uint64_t orc_seg_lattice_arcs(const orc_config* cfg, const double* S, const double* M,
                              uint32_t T, int norm, double alpha_sum, orc_arc* arcs,
                              uint32_t* n_states, int32_t* final_state) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur, K = cfg->num_states;
  const int startState = 0;
  int next_state = 1; /* AddState() for the start state */
  uint64_t na = 0;
  int* nss = (int*)malloc(sizeof(int) * ((size_t)T + 2)); /* nodeStartStates */
  uint32_t* plist = (uint32_t*)malloc(sizeof(uint32_t) * (L + 2));
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    if (t == 0) nss[0] = startState + 1; /* :248-250 */
    uint32_t num_new_states = 0;
    if (np > 0) { /* boundary states :260-296 */
      const double* Mt = M + (size_t)t * L * L;
      for (uint32_t lab = 0; lab < L; lab++) {
        int cur_state = next_state++;
        num_new_states++;
        /* one state per label: every previous label ascending (:275-293); with K states per phone
         * (nStateBuildLattice :563-597) a start state takes the end states ascending and then itself, any other
         * state the one before it and then itself */
        uint32_t npl = 0;
        if (K <= 1) for (uint32_t p = 0; p < L; p++) plist[npl++] = p;
        else if (lab % K == 0) { for (uint32_t e = K - 1; e < L; e += K) plist[npl++] = e; plist[npl++] = lab; }
        else { plist[npl++] = lab - 1; plist[npl++] = lab; }
        for (uint32_t i = 0; i < npl; i++) {
          const uint32_t prev_lab = plist[i];
          float value = -1 * Mt[prev_lab * L + lab];
          int prev_state = (t == 1) ? nss[t - 1] + (int)prev_lab : nss[t - 1] + (int)L + (int)prev_lab;
          orc_arc a = {prev_state, 0, 0, value, cur_state};
          arcs[na++] = a;
        }
      }
    }
    for (uint32_t lab = 0; lab < L; lab++) { /* end states :300-325 */
      int cur_state = next_state++;
      num_new_states++;
      int cur_lab = (int)lab;
      for (uint32_t dur = 1; dur <= np; dur++) {
        float value = -1 * S[(base + dur - 1) * L + lab];
        int prev_state = nss[t - dur + 1] + (int)lab;
        orc_arc a = {prev_state, cur_lab + 1, cur_lab + 1, value, cur_state};
        arcs[na++] = a;
        cur_lab += (int)L;
      }
      for (uint32_t dur = np + 1; dur <= nd; dur++) {
        float value = -1 * S[(base + dur - 1) * L + lab];
        orc_arc a = {startState, cur_lab + 1, cur_lab + 1, value, cur_state};
        arcs[na++] = a;
        cur_lab += (int)L;
      }
    }
    nss[t + 1] = nss[t] + (int)num_new_states; /* :327-330 */
  }
  int fin = -1;
  if (T > 0) { /* :366-399 */
    double Zx = 0;
    if (norm) Zx = -1 * alpha_sum;
    fin = next_state++;
    for (uint32_t prev_lab = 0; prev_lab < L; prev_lab++) {
      int prev_state = (T == 1) ? nss[T - 1] + (int)prev_lab : nss[T - 1] + (int)L + (int)prev_lab;
      float w = -Zx;
      orc_arc a = {prev_state, 0, 0, w, fin};
      arcs[na++] = a;
    }
  }
  *n_states = (uint32_t)next_state;
  *final_state = fin;
  free(nss); free(plist);
  return na;
}

/* decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h: node t owns L states (one per label) from 1 + t*L;
 * per label, for dur = 1..numPrev and every previous label an arc from state(t-dur, p) with
 * float(-1 * getFullTransValue(p, lab, dur)) = float(-(M2 + S)), then the utterance-initial duration from
 * the start state with float(-S); labels lab + L*(dur-1) + 1 on both tapes; L epsilon arcs of weight -Zx
 * (= -0.0 without norm) into the final state. */
uint64_t orc_segtrans_lattice_num_arcs(uint32_t T, uint32_t L, uint32_t D) {
  uint64_t n = 0;
  for (uint32_t t = 0; t < T; t++) {
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    n += (uint64_t)L * ((uint64_t)np * L + (nd - np));
  }
  return T ? n + L : 0;
}
uint64_t orc_segtrans_lattice_arcs(const orc_config* cfg, const double* S, const double* M2, uint32_t T,
                                   int norm, double alpha_sum, orc_arc* arcs, uint32_t* n_states,
                                   int32_t* final_state) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  uint64_t na = 0;
  int next_state = 1;
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    for (uint32_t lab = 0; lab < L; lab++) {
      const int cur_state = next_state++;
      int cur_lab = (int)lab;
      for (uint32_t dur = 1; dur <= np; dur++) {
        for (uint32_t prev_lab = 0; prev_lab < L; prev_lab++) {
          float value = -1 * (M2[(base + dur - 1) * (size_t)L * L + prev_lab * L + lab] + S[(base + dur - 1) * L + lab]);
          orc_arc a = {1 + (int)((t - dur) * L + prev_lab), cur_lab + 1, cur_lab + 1, value, cur_state};
          arcs[na++] = a;
        }
        cur_lab += (int)L;
      }
      for (uint32_t dur = np + 1; dur <= nd; dur++) {
        float value = -1 * S[(base + dur - 1) * L + lab];
        orc_arc a = {0, cur_lab + 1, cur_lab + 1, value, cur_state};
        arcs[na++] = a;
        cur_lab += (int)L;
      }
    }
  }
  int fin = -1;
  if (T > 0) {
    double Zx = 0;
    if (norm) Zx = -1 * alpha_sum;
    fin = next_state++;
    for (uint32_t prev_lab = 0; prev_lab < L; prev_lab++) {
      float w = -Zx;
      orc_arc a = {1 + (int)((T - 1) * L + prev_lab), 0, 0, w, fin};
      arcs[na++] = a;
    }
  }
  *n_states = (uint32_t)next_state;
  *final_state = fin;
  return na;
}

/* ======================================================================================
 * f3: n-state frame model (nodes/CRF_StdNStateNode.cpp + trainers/gradbuilders/CRF_NewGradBuilder.cpp)
 * ====================================================================================== */
void orc_nstate_scores(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* ftrs, uint32_t T,
                       double* S, double* TD, double* TO, double* TE) {
  /* computeTransMatrix :68-90 */
  const uint32_t L = cfg->num_labs, K = cfg->num_states, P = L / K, F = cfg->num_feas;
  for (uint32_t t = 0; t < T; t++) {
    const float* x = ftrs + (size_t)t * F;
    for (uint32_t clab = 0; clab < L; clab++) {
      S[(size_t)t * L + clab] = orc_state_value(cfg, lay, x, lambda, clab);
      TD[(size_t)t * L + clab] = orc_trans_value(cfg, lay, x, lambda, clab, clab);
      if (clab % K == 0) {
        for (uint32_t plab = 0; plab < P; plab++)
          TE[(size_t)t * P * P + plab * P + clab / K] = orc_trans_value(cfg, lay, x, lambda, plab * K + K - 1, clab);
      } else {
        TO[(size_t)t * L + clab - 1] = orc_trans_value(cfg, lay, x, lambda, clab - 1, clab);
      }
    }
    for (uint32_t e = K - 1; e < L; e += K) TO[(size_t)t * L + e] = 0.0; /* end states have no next state */
  }
}

int orc_nstate_forward(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                       uint32_t T, double* alpha, double* Zx) {
  /* computeFirstAlpha :148-157, computeAlpha :100-137, computeAlphaSum :376-383 */
  const uint32_t L = cfg->num_labs, K = cfg->num_states, P = L / K;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* acc = (double*)malloc(sizeof(double) * P);
  for (uint32_t c = 0; c < L; c++) alpha[c] = S[c];
  for (uint32_t t = 1; t < T && err == ORC_OK; t++) {
    const double* pa = alpha + (size_t)(t - 1) * L;
    double* a = alpha + (size_t)t * L;
    for (uint32_t clab = 0; clab < L; clab++) {
      double v = pa[clab] + TD[(size_t)t * L + clab];
      if (clab % K == 0) {
        const uint32_t q = clab / K;
        acc[0] = pa[K - 1] + TE[(size_t)t * P * P + q];
        double maxv = acc[0];
        for (uint32_t plab = 1; plab < P; plab++) {
          acc[plab] = pa[plab * K + K - 1] + TE[(size_t)t * P * P + plab * P + q];
          if (acc[plab] > maxv) maxv = acc[plab];
        }
        const double ls = orc_logadd_max_n(acc, maxv, (int)P, &err);
        v = orc_logadd2(v, ls, &err);
      } else {
        v = orc_logadd2(v, pa[clab - 1] + TO[(size_t)t * L + clab - 1], &err);
      }
      a[clab] = v + S[(size_t)t * L + clab];
    }
  }
  if (err == ORC_OK) *Zx = orc_logadd_n(alpha + (size_t)(T - 1) * L, (int)L, &err);
  free(acc);
  return err;
}

int orc_nstate_backward(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                        uint32_t T, double* beta) {
  /* setTailBeta :262-267; node t+1's computeBeta(beta of node t) :170-207 */
  const uint32_t L = cfg->num_labs, K = cfg->num_states, P = L / K;
  int err = ORC_OK;
  if (T == 0) return ORC_ERR_EMPTY;
  double* acc = (double*)malloc(sizeof(double) * P);
  double* tb = (double*)malloc(sizeof(double) * L);
  for (uint32_t c = 0; c < L; c++) beta[(size_t)(T - 1) * L + c] = 0.0;
  for (uint32_t t = T - 1; t-- > 0 && err == ORC_OK;) {
    const uint32_t n = t + 1;
    for (uint32_t c = 0; c < L; c++) tb[c] = beta[(size_t)n * L + c] + S[(size_t)n * L + c];
    for (uint32_t plab = 0; plab < L; plab++) {
      double v = tb[plab] + TD[(size_t)n * L + plab];
      if ((plab + 1) % K == 0) {
        const uint32_t ip = (plab + 1) / K - 1;
        acc[0] = TE[(size_t)n * P * P + ip * P] + tb[0];
        double maxv = acc[0];
        for (uint32_t q = 1; q < P; q++) {
          acc[q] = TE[(size_t)n * P * P + ip * P + q] + tb[q * K];
          if (acc[q] > maxv) maxv = acc[q];
        }
        const double ls = orc_logadd_max_n(acc, maxv, (int)P, &err);
        v = orc_logadd2(v, ls, &err);
      } else {
        v = orc_logadd2(v, TO[(size_t)n * L + plab] + tb[plab + 1], &err);
      }
      beta[(size_t)t * L + plab] = v;
    }
  }
  free(acc); free(tb);
  return err;
}

int orc_nstate_build_gradient(const orc_config* cfg, const orc_layout* lay, const double* lambda, const float* ftrs,
                              const uint32_t* labels, uint32_t T, double* grad, double* numer, double* Zx_out) {
  /* CRF_NewGradBuilder::buildGradient (frame level) with computeExpF :273-331 */
  const uint32_t L = cfg->num_labs, K = cfg->num_states, P = L / K, F = cfg->num_feas;
  if (T == 0) return ORC_ERR_EMPTY;
  int err = ORC_OK;
  double* ExpF = (double*)calloc(lay->lambda_len, sizeof(double));
  double* S = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* TD = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* TO = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* TE = (double*)malloc(sizeof(double) * (size_t)T * P * P);
  double* alpha = (double*)malloc(sizeof(double) * (size_t)T * L);
  double* beta = (double*)malloc(sizeof(double) * (size_t)T * L);
  double logLi = 0.0, Zx = 0.0;
  orc_nstate_scores(cfg, lay, lambda, ftrs, T, S, TD, TO, TE);
  err = orc_nstate_forward(cfg, S, TD, TO, TE, T, alpha, &Zx);
  if (err == ORC_OK) err = orc_nstate_backward(cfg, S, TD, TO, TE, T, beta);
  for (uint32_t t = T; t-- > 0 && err == ORC_OK;) {
    const float* x = ftrs + (size_t)t * F;
    const double* pa = (t > 0) ? alpha + (size_t)(t - 1) * L : NULL;
    const uint32_t prev_lab = (t > 0) ? labels[t - 1] : L + 1;
    const uint32_t label = labels[t];
    const double* a = alpha + (size_t)t * L;
    const double* b = beta + (size_t)t * L;
    const double* St = S + (size_t)t * L;
    double ab_tot = 0.0, ab_trans_tot = 0.0, nodeLi = 0.0;
    for (uint32_t clab = 0; clab < L; clab++) {
      double ab = orc_expE(a[clab] + b[clab] - Zx, &err);
      ab_tot += ab;
      nodeLi += orc_state_expf(cfg, lay, x, lambda, ExpF, grad, ab, label, clab);
      if (prev_lab > L) {
        ab_trans_tot = 1.0;
        continue;
      }
      ab = orc_expE(pa[clab] + TD[(size_t)t * L + clab] + St[clab] + b[clab] - Zx, &err);
      ab_trans_tot += ab;
      nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, prev_lab, label, clab, clab);
      if (clab % K == 0) {
        const uint32_t q = clab / K;
        for (uint32_t plab = 0; plab < P; plab++) {
          const uint32_t rp = plab * K + K - 1;
          ab = orc_expE(pa[rp] + TE[(size_t)t * P * P + plab * P + q] + St[clab] + b[clab] - Zx, &err);
          ab_trans_tot += ab;
          nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, prev_lab, label, rp, clab);
        }
      } else {
        ab = orc_expE(pa[clab - 1] + TO[(size_t)t * L + clab - 1] + St[clab] + b[clab] - Zx, &err);
        ab_trans_tot += ab;
        nodeLi += orc_trans_expf(cfg, lay, x, lambda, ExpF, grad, ab, prev_lab, label, clab - 1, clab);
      }
    }
    if (ab_tot > 1.1 || ab_tot < 0.9 || ab_trans_tot > 1.1 || ab_trans_tot < 0.9) set_err(&err, ORC_ERR_PROB_SUM); /* :333-349 */
    logLi += nodeLi;
  }
  for (uint32_t i = 0; i < lay->lambda_len; i++) grad[i] -= ExpF[i];
  *Zx_out = Zx;
  *numer = logLi;
  free(ExpF); free(S); free(TD); free(TO); free(TE); free(alpha); free(beta);
  return err;
}

/* decoders/CRF_LatticeBuilder.h:715-840 nStateBuildLattice: state 0 = start, state of (t, c) = 1 + t*nLabs + c; node 0 from
 * the start with float(-1*stateArray); node t: a phone's start state from every phone's END state (ascending) and then
 * from itself, the other states from the state before and then from themselves, weight float(-1*getFullTransValue);
 * the final state from EVERY label of the last node with weight Zx (= -alpha sum with norm, else 0). */
uint64_t orc_nstate_lattice_num_arcs(uint32_t T, uint32_t nLabs, uint32_t K) {
  if (T == 0) return 0;
  const uint32_t P = nLabs / K;
  return (uint64_t)nLabs + (uint64_t)(T - 1) * ((uint64_t)P * (P + 1) + (uint64_t)(nLabs - P) * 2) + nLabs;
}

uint64_t orc_nstate_lattice_arcs(const orc_config* cfg, const double* S, const double* TD, const double* TO, const double* TE,
                                 uint32_t T, int norm, double alpha_sum, orc_arc* arcs, uint32_t* n_states, int32_t* final_state) {
  const uint32_t L = cfg->num_labs, K = cfg->num_states, P = L / K;
  uint64_t na = 0;
  int next_state = 1;
  for (uint32_t t = 0; t < T; t++) {
    for (uint32_t c = 0; c < L; c++) {
      const int cur_state = next_state++;
      const double sv = S[(size_t)t * L + c];
      if (t == 0) {
        float value = -1 * sv;
        orc_arc a = {0, (int)c + 1, (int)c + 1, value, cur_state};
        arcs[na++] = a;
        continue;
      }
      const int pbase = 1 + (int)((t - 1) * L);
      if (c % K == 0) {
        for (uint32_t pl = K - 1; pl < L; pl += K) {
          float value = -1 * (TE[(size_t)t * P * P + (pl / K) * P + c / K] + sv);
          orc_arc a = {pbase + (int)pl, (int)c + 1, (int)c + 1, value, cur_state};
          arcs[na++] = a;
        }
        float value = -1 * (TD[(size_t)t * L + c] + sv);
        orc_arc a = {pbase + (int)c, (int)c + 1, (int)c + 1, value, cur_state};
        arcs[na++] = a;
      } else {
        float value = -1 * (TO[(size_t)t * L + c - 1] + sv);
        orc_arc a = {pbase + (int)c - 1, (int)c + 1, (int)c + 1, value, cur_state};
        arcs[na++] = a;
        value = -1 * (TD[(size_t)t * L + c] + sv);
        orc_arc a2 = {pbase + (int)c, (int)c + 1, (int)c + 1, value, cur_state};
        arcs[na++] = a2;
      }
    }
  }
  int fin = -1;
  if (T > 0) {
    double Zx = 0;
    if (norm) Zx = -1 * alpha_sum;
    fin = next_state++;
    for (uint32_t pl = 0; pl < L; pl++) {
      float w = Zx;
      orc_arc a = {1 + (int)((T - 1) * L + pl), 0, 0, w, fin};
      arcs[na++] = a;
    }
  }
  *n_states = (uint32_t)next_state;
  *final_state = fin;
  return na;
}

/* decoders/CRF_LatticeBuilder_StdSeg.h:40-590 (STDSEG).  State 0 = start; node t owns one state per available full
 * label, numbered from nodeStartStates[t] = 1 + La * seg_base(t) (:300-305, :425-427, :452-454), i.e. state of
 * (t, clab) = 1 + row(t,dur)*La + phone; arcs per node: dur ascending, phone ascending, then previous full label
 * ascending with weight float(-1 * getFullTransValue(plab, clab)) = float(-(transMatrix + stateArray)), the
 * utterance-initial durations one arc from the start state with float(-1 * stateArray[clab]); labels clab + 1;
 * final state last, reached from every available label of the last node by an epsilon arc of weight -Zx
 * (Zx = -alpha sum with norm, else 0). */
uint64_t orc_stdseg_lattice_num_arcs(uint32_t T, uint32_t La, uint32_t D) {
  uint64_t na = 0;
  for (uint32_t t = 0; t < T; t++) {
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    for (uint32_t dur = 1; dur <= np; dur++) na += (uint64_t)La * La * orc_node_max_dur(t - dur, D);
    na += (uint64_t)(nd - np) * La;
  }
  if (T > 0) na += (uint64_t)La * orc_node_max_dur(T - 1, D);
  return na;
}

uint64_t orc_stdseg_lattice_arcs(const orc_config* cfg, const double* S, const double* MX, uint32_t T, int norm,
                                 double alpha_sum, orc_arc* arcs, uint32_t* n_states, int32_t* final_state) {
  const uint32_t NL = cfg->num_labs, D = cfg->lab_max_dur, La = NL / D;
  uint64_t na = 0;
  int next_state = 1;
  for (uint32_t t = 0; t < T; t++) {
    const uint64_t base = orc_seg_base(t, D);
    const uint32_t np = num_prev(t, D), nd = orc_node_max_dur(t, D);
    int cur_lab = 0;
    for (uint32_t dur = 1; dur <= nd; dur++) {
      for (uint32_t lab = 0; lab < La; lab++) {
        const int cur_state = next_state++;
        const size_t at = (base + dur - 1) * La + lab;
        if (dur <= np) {
          const uint32_t pavail = La * orc_node_max_dur(t - dur, D);
          const int pstart = 1 + (int)(orc_seg_base(t - dur, D) * La);
          for (uint32_t prev_lab = 0; prev_lab < pavail; prev_lab++) {
            float value = -1 * (MX[((base + dur - 1) * (size_t)NL + prev_lab) * La + lab] + S[at]);
            orc_arc a = {pstart + (int)prev_lab, cur_lab + 1, cur_lab + 1, value, cur_state};
            arcs[na++] = a;
          }
        } else {
          float value = -1 * S[at];
          orc_arc a = {0, cur_lab + 1, cur_lab + 1, value, cur_state};
          arcs[na++] = a;
        }
        cur_lab++;
      }
    }
  }
  int fin = -1;
  if (T > 0) {
    double Zx = 0;
    if (norm) Zx = -1 * alpha_sum;
    fin = next_state++;
    const int pstart = 1 + (int)(orc_seg_base(T - 1, D) * La);
    for (uint32_t prev_lab = 0; prev_lab < La * orc_node_max_dur(T - 1, D); prev_lab++) {
      float w = -Zx;
      orc_arc a = {pstart + (int)prev_lab, 0, 0, w, fin};
      arcs[na++] = a;
    }
  }
  *n_states = (uint32_t)next_state;
  *final_state = fin;
  return na;
}

uint64_t orc_frame_lattice_num_arcs(uint32_t T, uint32_t L) {
  if (T == 0) return L;
  return (uint64_t)L + (uint64_t)(T - 1) * L * L + L;
}

/* decoders/CRF_LatticeBuilder.h:97-204. S: [T][L], M: [T][L*L] */
uint64_t orc_frame_lattice_arcs(const orc_config* cfg, const double* S, const double* M,
                                uint32_t T, int norm, double alpha_sum, orc_arc* arcs,
                                uint32_t* n_states, int32_t* final_state) {
  const uint32_t L = cfg->num_labs;
  const int startState = 0;
  int next_state = 1;
  uint64_t na = 0;
  for (uint32_t t = 0; t < T; t++) {
    if (t == 0) {
      for (uint32_t cur_lab = 0; cur_lab < L; cur_lab++) {
        float value = -1 * S[cur_lab];
        int cur_state = next_state++;
        orc_arc a = {startState, (int)cur_lab + 1, (int)cur_lab + 1, value, cur_state};
        arcs[na++] = a;
      }
    } else {
      int cur_time = (int)t + 1;
      for (uint32_t cur_lab = 0; cur_lab < L; cur_lab++) {
        int cur_state = next_state++;
        for (uint32_t prev_lab = 0; prev_lab < L; prev_lab++) {
          /* getFullTransValue nodes/CRF_StdStateNode.cpp:328-334 */
          float value = -1 * (M[(size_t)t * L * L + prev_lab * L + cur_lab] + S[(size_t)t * L + cur_lab]);
          int prev_state = (int)L * (cur_time - 2) + ((int)prev_lab + 1);
          orc_arc a = {prev_state, (int)cur_lab + 1, (int)cur_lab + 1, value, cur_state};
          arcs[na++] = a;
        }
      }
    }
  }
  double Zx = 0;
  if (norm) Zx = -1 * alpha_sum;
  int fin = next_state++;
  for (uint32_t prev_lab = 0; prev_lab < L; prev_lab++) {
    int prev_state = (int)L * ((int)T - 1) + ((int)prev_lab + 1);
    float w = Zx;
    orc_arc a = {prev_state, 0, 0, w, fin};
    arcs[na++] = a;
  }
  *n_states = (uint32_t)next_state;
  *final_state = fin;
  return na;
}

/* CRFFstDecode/src/Main.cpp:838-889: ShortestPath(lat,1) -> Project(OUTPUT) -> RmEpsilon ->
 * TopSort -> olabel-1.  OpenFST is not in /root/reference (un-vendored, unpinned; a stale
 * comment names OpenFst-1.3.4): its published single-source shortest path over the
 * tropical semiring is restated here -- relax states in state-id order (the lattices are
 * built top-sorted, so OpenFST's AutoQueue selects StateOrderQueue), arcs of a state in
 * insertion order, update only on strict improvement.  PARITY UNPINNED (tie-breaking). */
int64_t orc_best_path(const orc_arc* arcs, uint64_t n_arcs, uint32_t n_states, int32_t start,
                      int32_t final_state, uint32_t* out_labels, uint64_t max_out,
                      float* best_cost) {
  if (final_state < 0 || (uint32_t)final_state >= n_states) return -1;
  uint64_t* first = (uint64_t*)calloc((size_t)n_states + 1, sizeof(uint64_t));
  uint64_t* order = (uint64_t*)malloc(sizeof(uint64_t) * (n_arcs ? n_arcs : 1));
  float* dist = (float*)malloc(sizeof(float) * n_states);
  int64_t* parent = (int64_t*)malloc(sizeof(int64_t) * n_states);
  int64_t ret = -1;
  for (uint64_t a = 0; a < n_arcs; a++) {
    if (arcs[a].dst <= arcs[a].src) { ret = -2; goto done; } /* not top-sorted */
    first[arcs[a].src + 1]++;
  }
  for (uint32_t s = 0; s < n_states; s++) first[s + 1] += first[s];
  {
    uint64_t* fill = (uint64_t*)malloc(sizeof(uint64_t) * n_states);
    memcpy(fill, first, sizeof(uint64_t) * n_states);
    for (uint64_t a = 0; a < n_arcs; a++) order[fill[arcs[a].src]++] = a; /* stable */
    free(fill);
  }
  for (uint32_t s = 0; s < n_states; s++) {
    dist[s] = INFINITY; /* TropicalWeight::Zero() */
    parent[s] = -1;
  }
  dist[start] = 0.0f; /* One() */
  for (uint32_t s = 0; s < n_states; s++) {
    if (dist[s] == INFINITY) continue; /* never enqueued */
    for (uint64_t k = first[s]; k < first[s + 1]; k++) {
      const orc_arc* a = &arcs[order[k]];
      float nd = dist[s] + a->w; /* Times */
      if (nd < dist[a->dst]) {   /* d != Plus(d, nd) */
        dist[a->dst] = nd;
        parent[a->dst] = (int64_t)order[k];
      }
    }
  }
  if (dist[final_state] == INFINITY) goto done;
  if (best_cost) *best_cost = dist[final_state] + 0.0f; /* Times(d, Final = One) */
  {
    /* backtrace, then emit non-epsilon output labels start -> final */
    uint64_t n = 0;
    for (int32_t s = final_state; s != start;) {
      const orc_arc* a = &arcs[parent[s]];
      if (a->olabel != 0) n++;
      s = a->src;
    }
    ret = (int64_t)n;
    uint64_t pos = n;
    for (int32_t s = final_state; s != start;) {
      const orc_arc* a = &arcs[parent[s]];
      if (a->olabel != 0) {
        pos--;
        if (pos < max_out) out_labels[pos] = (uint32_t)(a->olabel - 1);
      }
      s = a->src;
    }
  }
done:
  free(first); free(order); free(dist); free(parent);
  return ret;
}

/* ---- a19: free-phone-loop Viterbi decoder (see scrf_oracle.h) ------------------------------- */
int orc_free_phone_decode(const orc_config* cfg, const double* S, const double* M, uint32_t T,
                          uint32_t* seg_phone, uint32_t* seg_dur, float* seg_weight,
                          int32_t* seg_phone_start, uint32_t* n_segs, float* best_weight) {
  const uint32_t L = cfg->num_labs, D = cfg->lab_max_dur;
  *n_segs = 0;
  *best_weight = 99999.0f;
  if (T == 0) return -1;
  const float BIG = 99999.0f; /* the decoder's "infinity" */
  /* hypothesis lists of the nodes: key (phone l, dur d) -> weight, back pointer, start flag */
  const size_t HN = (size_t)T * L * D;
  float* hw = (float*)malloc(sizeof(float) * HN);
  int32_t* hp = (int32_t*)malloc(sizeof(int32_t) * HN);
  int8_t* hs = (int8_t*)malloc(HN);
  int8_t* hset = (int8_t*)calloc(HN, 1);
  /* per node and phone after choose_nState_Best_Seg: weight, chosen dur */
  float* bw = (float*)malloc(sizeof(float) * (size_t)T * L);
  uint32_t* bd = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)T * L);
#define H(t, l, d) (((size_t)(t) * L + (l)) * D + ((d)-1))
  for (uint32_t t = 0; t < T; t++) {
    /* expansions entering at node t, pushed to nodes t .. t+D-1 */
    for (uint32_t l = 0; l < L; l++) {
      for (int pass = 0; pass < 2; pass++) {            /* cross (all p != l, list order), then internal */
        const uint32_t p0 = pass == 0 ? 0 : l, p1 = pass == 0 ? L : l + 1;
        for (uint32_t p = p0; p < p1; p++) {
          float w;
          int32_t ptr;
          int8_t start;
          if (t == 0) {
            if (pass == 1 || p != 0) continue;          /* one expansion from the LM start state */
            w = 0.0f + 0.0f;
            ptr = -1;
            start = 1;
          } else {
            if (pass == 0 && p == l) continue;
            if (bw[(size_t)(t - 1) * L + p] >= BIG) continue;
            const float trans_wt = (float)(-1 * M[((size_t)t * L + p) * L + l]);
            w = bw[(size_t)(t - 1) * L + p] + trans_wt;
            ptr = (int32_t)p;
            start = pass == 0;
          }
          for (uint32_t d = 1; d <= D && t + d - 1 < T; d++) {
            const size_t k = H(t + d - 1, l, d);
            if (!hset[k] || w < hw[k]) { hset[k] = 1; hw[k] = w; hp[k] = ptr; hs[k] = start; }
          }
        }
      }
    }
    /* stateValueUpdate + choose_nState_Best_Seg at node t */
    const uint64_t base = orc_seg_base(t, D);
    for (uint32_t l = 0; l < L; l++) {
      float best = BIG;
      uint32_t arg = 0;
      for (uint32_t d = D; d >= 1; d--) {               /* list order: entered at t-d+1, earliest first */
        const size_t k = H(t, l, d);
        if (d > t + 1 || !hset[k]) continue;
        const float phnStateVal = (float)(-1 * S[(base + d - 1) * L + l]);
        hw[k] = hw[k] + phnStateVal;
        if (arg == 0 || hw[k] < best) { best = hw[k]; arg = d; }
      }
      bw[(size_t)t * L + l] = arg ? best : BIG;
      bd[(size_t)t * L + l] = arg;
    }
  }
  /* final: best end hypothesis in list order (:2122-2133) */
  float min_weight = BIG;
  int32_t bl = -1;
  for (uint32_t l = 0; l < L; l++)
    if (bd[(size_t)(T - 1) * L + l] && bw[(size_t)(T - 1) * L + l] < min_weight) { min_weight = bw[(size_t)(T - 1) * L + l]; bl = (int32_t)l; }
  int rc = -1;
  if (bl >= 0) {
    uint32_t n = 0;
    int64_t te = (int64_t)T - 1;
    int32_t l = bl;
    while (te >= 0) {
      const uint32_t d = bd[(size_t)te * L + l];
      const size_t k = H(te, l, d);
      const int64_t ts = te + 1 - (int64_t)d;
      const double sv = S[(orc_seg_base((uint32_t)te, D) + d - 1) * L + l];
      seg_phone[n] = (uint32_t)l;
      seg_dur[n] = d;
      if (ts == 0) {
        seg_weight[n] = (float)(-1 * sv);
        seg_phone_start[n] = 1;
      } else {
        const int32_t p = hp[k];
        seg_weight[n] = (float)(-1 * (M[((size_t)te * L + p) * L + l] + sv)); /* END node's matrix */
        seg_phone_start[n] = hs[k];
        l = p;
      }
      n++;
      te = ts - 1;
    }
    for (uint32_t i = 0; i < n / 2; i++) { /* first to last */
      uint32_t a = seg_phone[i]; seg_phone[i] = seg_phone[n - 1 - i]; seg_phone[n - 1 - i] = a;
      a = seg_dur[i]; seg_dur[i] = seg_dur[n - 1 - i]; seg_dur[n - 1 - i] = a;
      float w = seg_weight[i]; seg_weight[i] = seg_weight[n - 1 - i]; seg_weight[n - 1 - i] = w;
      int32_t b = seg_phone_start[i]; seg_phone_start[i] = seg_phone_start[n - 1 - i]; seg_phone_start[n - 1 - i] = b;
    }
    *n_segs = n;
    *best_weight = min_weight;
    rc = 0;
  }
#undef H
  free(hw); free(hp); free(hs); free(hset); free(bw); free(bd);
  return rc;
}
