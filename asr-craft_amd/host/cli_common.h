// cli_common.h -- `name=value` argument table and ascii feature/label readers shared by the
// CRFTrain / CRFFstDecode front-ends (the reference parses the same flag names with QuickNet's
// QN_initargs, CRFTrain/src/Main.cpp:146-256, CRFFstDecode/src/Main.cpp).  Only the flags that
// reach the hot path are honoured; the others are accepted and ignored with a notice.
// Feature files: binary pfile (the reference's default format) or the "ascii" layout, one line
// per frame `sent frame v0 v1 ...` (the reference's bundled CRFTrain/test*.ascii fixtures); label
// files: binary ILAB or ascii `sent frame label` (qn_files.h).
#ifndef CLI_COMMON_H_
#define CLI_COMMON_H_

#include <stdlib.h>

#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "crf_amd.h"
#include "ftr_files.h"

struct Args {
  std::map<std::string, std::string> kv;
  Args(int argc, char** argv) {
    for (int i = 1; i < argc; i++) {
      std::string a(argv[i]);
      size_t k = a.find('=');
      if (k == std::string::npos) { std::cerr << "argument '" << a << "' is not name=value" << std::endl; exit(1); }
      kv[a.substr(0, k)] = a.substr(k + 1);
    }
  }
  bool has(const std::string& k) const { return kv.count(k) && !kv.at(k).empty(); }
  std::string str(const std::string& k, const std::string& d = "") const { return has(k) ? kv.at(k) : d; }
  long num(const std::string& k, long d) const { return has(k) ? atol(kv.at(k).c_str()) : d; }
  double real(const std::string& k, double d) const { return has(k) ? atof(kv.at(k).c_str()) : d; }
};

struct CliModel {
  CRF_FeatureMap_config fmap;
  std::vector<scrf_stream_recipe> recipes;
  uint32_t D = 1, L = 0, F = 0;
  modeltype mtype = STDFRAME;
};

inline modeltype parse_model_type(const std::string& s) {
  if (s == "stdseg") return STDSEG;
  if (s == "stdseg_no_dur") return STDSEG_NO_DUR;
  if (s == "stdseg_no_dur_no_transftr") return STDSEG_NO_DUR_NO_TRANSFTR;
  if (s == "stdseg_no_dur_no_segtransftr") return STDSEG_NO_DUR_NO_SEGTRANSFTR;
  return STDFRAME;
}

// Flags of the reference's argument tables that would change the numbers and that this build does not
// implement are refused instead of ignored (the remaining unknown names are accepted silently, like
// logging and scheduling options)
inline void refuse_unbuilt_flags(const Args& a, uint32_t D) {
  auto die = [](const std::string& m) { std::cerr << m << std::endl; exit(1); };
  for (int k = 1; k <= 3; k++) {
    const std::string p = "ftr" + std::to_string(k) + "_";
    if (a.num(p + "delta_order", 0) != 0) die(p + "delta_order: delta features are not built");
    if (a.has(p + "norm_file")) die(p + "norm_file: feature normalisation files are not built");
    if (a.num(p + "window_offset", 0) != 0) die(p + "window_offset must be 0");
    const long wl = a.num(p + "window_len", 1);
    if (wl != 1 && wl != (long)D) die(p + "window_len must be 1 or label_maximum_duration (the segment windows are synthesised from the raw frames)");
    if (a.num(p + "use_boundary_delta_ftr", 0) != 0) die(p + "use_boundary_delta_ftr is not built");
  }
  const long we = a.num("window_extent", 1);
  if (we != 1 && we != (long)D) die("window_extent must be 1 or label_maximum_duration");
  if (a.num("hardtarget_window_offset", 0) != 0) die("hardtarget_window_offset must be 0");
  if (a.num("use_broken_class_label", 0) != 0) die("use_broken_class_label: broken-class labels are not built");
  if (a.has("crf_objective_function") && a.str("crf_objective_function") != "expf") die("crf_objective_function=" + a.str("crf_objective_function") + " is not built (expf only)");
  if (a.has("crf_featuremap_file")) die("crf_featuremap_file: file-defined feature maps are not built");
}

// set_fmap_config of CRFTrain/src/Main.cpp:372-430: the feature-map configuration from the flags and the
// joined window width m->F
inline void set_fmap_config(const Args& a, CliModel* m) {
  CRF_FeatureMap_config& c = m->fmap;
  const std::string fm = a.str("crf_featuremap", "stdstate");
  c.map_type = fm == "stdtrans" ? STDTRANS : STDSTATE;
  if (fm != "stdstate" && fm != "stdtrans") { std::cerr << "crf_featuremap=" << fm << " is not built" << std::endl; exit(1); }
  if (m->mtype == STDSEG_NO_DUR_NO_TRANSFTR && fm != "stdstate") {   // CRFTrain/src/Main.cpp:465-468, exit code of its catch block
    std::cerr << "Exception: main() in CRFTrain caught exception: crf_featuremap must be \"stdstate\" for \"stdseg_no_dur_no_transftr\" CRF model." << std::endl;
    exit(-1);
  }
  c.numLabs = m->L;
  c.numFeas = m->F;
  c.numStates = (QNUInt32)a.num("crf_states", 1);
  c.useStateFtrs = true;
  c.stateFidxStart = (QNUInt32)a.num("crf_stateftr_start", 0);
  long e = a.num("crf_stateftr_end", -1);
  c.stateFidxEnd = e >= 0 ? (QNUInt32)e : m->F - 1;
  c.useTransFtrs = c.map_type == STDTRANS;
  c.transFidxStart = (QNUInt32)a.num("crf_transftr_start", 0);
  e = a.num("crf_transftr_end", -1);
  c.transFidxEnd = e >= 0 ? (QNUInt32)e : m->F - 1;
  c.useStateBias = a.num("crf_use_state_bias", 1) != 0;
  c.useTransBias = a.num("crf_use_trans_bias", 1) != 0;
  c.stateBiasVal = a.real("crf_state_bias_value", 1.0);
  c.transBiasVal = a.real("crf_trans_bias_value", 1.0);
  c.maxDur = m->D;
  c.durFtrStart = (QNUInt32)a.num("dur_ftr_start", 0);
  c.nActualLabs = (QNUInt32)a.num("num_actual_labs", m->L);
}

// crf_precision=exact|fast|fastlin|fast32: arithmetic of the training contractions (scrf_precision, scrf_abi.h).
// fast (default): fp64 MFMA, sums re-associated (<= 1e-9 relative on gradients; the contract is 1e-4);
// fastlin: fast with the segment recipe's window average taken as the exact mean (linear in the frames: it leaves the dense
// contractions; <= 1e-6, measured 3e-8; what bench.py runs); exact: the reference's operation order everywhere.
// Decode entry points are always exact.
inline uint32_t parse_precision(const Args& a) {
  const std::string p = a.str("crf_precision", "fast");
  if (p == "exact") return SCRF_PREC_EXACT;
  if (p == "fast") return SCRF_PREC_FAST;
  if (p == "fastlin") return SCRF_PREC_FASTLIN;
  if (p == "fast32") return SCRF_PREC_FAST32;
  std::cerr << "crf_precision=" << p << " (exact|fast|fastlin|fast32)" << std::endl;
  exit(1);
}
inline long env_num(const char* name, long d) { const char* v = getenv(name); return v && *v ? atol(v) : d; }

// streams (ftr1/ftr2/ftr3) + set_fmap_config of CRFTrain/src/Main.cpp:372-430
inline std::vector<FtrData> load_streams(const Args& a, CliModel* m) {
  std::vector<FtrData> data;
  m->D = (uint32_t)a.num("label_maximum_duration", 1);
  m->L = (uint32_t)a.num("crf_label_size", 0);
  m->mtype = parse_model_type(a.str("crf_model_type", "stdframe"));
  if (m->L == 0) { std::cerr << "crf_label_size is required" << std::endl; exit(1); }
  refuse_unbuilt_flags(a, m->D);
  m->F = 0;
  for (int k = 1; k <= 3; k++) {
    std::string p = "ftr" + std::to_string(k) + "_";
    if (!a.has(p + "file")) break;
    const std::string fmt = a.str(p + "format", "pfile");  // the reference's default
    const uint32_t f0 = (uint32_t)a.num(p + "ftr_start", 0), fc = (uint32_t)a.num(p + "ftr_count", 0);
    if (fmt == "pfile") {
      try { data.push_back(read_pfile_ftrs(a.str(p + "file"), f0, fc)); }
      catch (std::exception& e) { std::cerr << e.what() << std::endl; exit(1); }
    } else if (fmt == "ascii") {
      if (f0 || fc) { std::cerr << p << "ftr_start/ftr_count need " << p << "format=pfile" << std::endl; exit(1); }
      try { data.push_back(read_ascii_ftrs(a.str(p + "file"))); }
      catch (std::exception& e) { std::cerr << e.what() << std::endl; exit(1); }
    } else {
      std::cerr << p << "format=" << fmt << " is not built (pfile|ascii)" << std::endl;
      exit(1);
    }
    if (data.back().utts.size() != data[0].utts.size()) { std::cerr << p << "file holds " << data.back().utts.size() << " sentences, ftr1_file " << data[0].utts.size() << std::endl; exit(1); }
    const size_t w = data.back().width;
    scrf_stream_recipe r;
    r.in_width = (uint32_t)w;
    r.left_ctx = (uint32_t)a.num(p + "left_context_len", 0);
    r.right_ctx = (uint32_t)a.num(p + "right_context_len", 0);
    r.extract_seg_ftr = (int32_t)a.num(p + "extract_seg_ftr", 0);
    m->recipes.push_back(r);
    m->F += (m->D == 1) ? (r.left_ctx + 1 + r.right_ctx) * r.in_width
            : r.extract_seg_ftr ? 8 * r.in_width + m->D + (r.left_ctx + r.right_ctx) * r.in_width
                                : (r.left_ctx + 1 + r.right_ctx) * r.in_width;
  }
  if (data.empty()) { std::cerr << "ftr1_file is required" << std::endl; exit(1); }
  set_fmap_config(a, m);
  return data;
}

// sentences a run works on (train_sent_range / crf_eval_range, QN_Range syntax, qn_files.h)
inline std::vector<uint32_t> select_sents(const Args& a, const std::string& key, size_t n_sents) {
  return qn::parse_range(a.str(key, "all"), (uint32_t)n_sents);
}

#endif  // CLI_COMMON_H_
