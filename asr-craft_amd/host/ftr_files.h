// ftr_files.h -- feature / label file readers shared by CRF_FeatureStreamManager (crf_amd.cpp) and the
// front-ends: binary pfile (the reference's default, QN_build_ftrstream(format="pfile"),
// io/CRF_FeatureStreamManager.cpp:138) or the "ascii" layout of the reference's bundled fixtures, one line
// per frame `sent frame v0 v1 ...`; labels as QuickNet ILAB (QN_InLabStream_ILab, :285) or ascii
// `sent frame label`.  Errors are std::runtime_error.
#ifndef FTR_FILES_H_
#define FTR_FILES_H_

#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "qn_files.h"

// one input stream: sentence u = T x width floats, row-major.  ascii files are held whole; a pfile is
// read sentence by sentence on demand (a run over `crf_eval_range=0-399` touches those 400 only)
struct FtrData {
  size_t width = 0;
  std::vector<std::vector<float> > utts;      // ascii: everything; pfile: filled by get()
  std::shared_ptr<qn::PFileReader> pfile;
  uint32_t ftr_start = 0;
  size_t size() const { return utts.size(); }
  const std::vector<float>& get(size_t u) {
    if (pfile && utts[u].empty() && pfile->num_frames((uint32_t)u) > 0) pfile->read_sent((uint32_t)u, &utts[u], nullptr, ftr_start, (uint32_t)width);
    return utts[u];
  }
  void drop(size_t u) { if (pfile) std::vector<float>().swap(utts[u]); }  // the stream keeps its own copy
};

inline FtrData read_ascii_ftrs(const std::string& path) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) throw std::runtime_error("cannot open feature file " + path);
  FtrData d;
  std::string line;
  while (getline(f, line)) {
    std::istringstream is(line);
    long s, t;
    if (!(is >> s >> t)) continue;
    if ((size_t)s >= d.utts.size()) d.utts.resize(s + 1);
    size_t n = 0;
    float x;
    while (is >> x) { d.utts[s].push_back(x); n++; }
    if (d.width == 0) d.width = n;
    if (n != d.width) throw std::runtime_error(path + ": ragged feature line");
  }
  return d;
}

// pfile stream (QN_build_ftrstream(format="pfile"), io/CRF_FeatureStreamManager.cpp:138), columns
// ftr_start .. ftr_start+ftr_count (ftr_count 0 = the rest); sentences are read when first asked for
inline FtrData read_pfile_ftrs(const std::string& path, uint32_t ftr_start, uint32_t ftr_count) {
  FtrData d;
  d.pfile.reset(new qn::PFileReader(path));
  const qn::PFileInfo& info = d.pfile->info();
  if (ftr_start > info.n_ftrs) qn::fail(path, "ftr_start beyond the file's width");
  d.width = ftr_count ? ftr_count : info.n_ftrs - ftr_start;
  if (ftr_start + d.width > info.n_ftrs) qn::fail(path, "ftr_start + ftr_count beyond the file's width");
  d.ftr_start = ftr_start;
  d.utts.resize(d.pfile->num_sents());
  return d;
}

// hardtarget_file: QuickNet ILAB (QN_InLabStream_ILab, io/CRF_FeatureStreamManager.cpp:285) or the
// ascii `sent frame label` layout of the bundled fixture -- told apart by the magic
inline std::vector<std::vector<uint32_t> > read_labs(const std::string& path) {
  if (qn::is_ilab(path)) return qn::read_ilab(path).labels;
  std::ifstream f(path.c_str());
  if (!f.is_open()) throw std::runtime_error("cannot open label file " + path);
  std::vector<std::vector<uint32_t> > utts;
  long s, t, l;
  while (f >> s >> t >> l) {
    if ((size_t)s >= utts.size()) utts.resize(s + 1);
    utts[s].push_back((uint32_t)l);
  }
  return utts;
}

#endif  // FTR_FILES_H_
