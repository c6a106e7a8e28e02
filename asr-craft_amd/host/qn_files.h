// qn_files.h -- readers and writers for the two on-disk formats the reference's front-ends take
// their inputs from: ICSI/QuickNet "pfile" feature files (what
// io/CRF_FeatureStreamManager.cpp:138 opens through QN_build_ftrstream(format="pfile")) and
// QuickNet "ILAB" run-length label files (QN_InLabStream_ILab, CRF_FeatureStreamManager.cpp:285).
// Both formats are defined by QuickNet, which is not part of the reference tree:
//   * ILAB is pinned by the reference's own data file demo/timit-aux/timit_train.48labs.ilab
//     (3696 sentences, 1 124 823 frames; kept as tests/golden/timit_train.48labs.ilab): header,
//     record and index layout below were read off that file and every redundancy in it (frame
//     totals, per-sentence counts, record terminators, index offsets) is checked on read.
//   * pfile follows the published pfile(5) layout (version 0); the reference holds no pfile, so its
//     parity against QuickNet's reader is UNPINNED -- covered by write/read round trips and by an
//     independent parser in tests/test_qn_files.py.
// Pure host code, no GPU.
#ifndef QN_FILES_H_
#define QN_FILES_H_

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace qn {

inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline void put_be32(unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }
inline void fail(const std::string& path, const std::string& what) { throw std::runtime_error(path + ": " + what); }

struct File {
  FILE* f;
  std::string path;
  File(const std::string& p, const char* mode) : f(fopen(p.c_str(), mode)), path(p) {
    if (!f) fail(p, mode[0] == 'r' ? "cannot open for reading" : "cannot open for writing");
  }
  ~File() { if (f) fclose(f); }
  File(const File&) = delete;
  File& operator=(const File&) = delete;
  void seek(uint64_t off) { if (fseeko(f, (off_t)off, SEEK_SET)) fail(path, "seek failed"); }
  void read(void* dst, size_t n) { if (n && fread(dst, 1, n, f) != n) fail(path, "truncated file"); }
  void write(const void* src, size_t n) { if (n && fwrite(src, 1, n, f) != n) fail(path, "write failed"); }
  uint64_t size() {
    off_t at = ftello(f);
    fseeko(f, 0, SEEK_END);
    off_t e = ftello(f);
    fseeko(f, at, SEEK_SET);
    return (uint64_t)e;
  }
};

// ---------------------------------------------------------------------------------------------
// pfile: 32768-byte ascii header of `-key value...` lines ending with `-end`, then big-endian
// 32-bit words: the data section, nrow rows of ncol words `sentence frame features.. labels..`
// (features IEEE float32, everything else int32), then the sentence index, num_sentences+1 row
// numbers (the last one = num_frames).  `size` / `offset` of a section are in words from the end
// of the header.
// ---------------------------------------------------------------------------------------------
struct PFileInfo {
  uint32_t header_bytes = 32768, n_sents = 0, n_frames = 0, n_ftrs = 0, n_labs = 0, first_ftr_col = 2, first_lab_col = 2, n_cols = 2;
  uint64_t data_off = 0;             // words
  int64_t sent_table_off = -1;       // words, -1 if the file carries no index
};

class PFileReader {
 public:
  explicit PFileReader(const std::string& path) : fh_(path, "rb") {
    char head[64] = {0};
    fh_.read(head, 48);
    unsigned long hsize = 0;
    int version = -1;
    if (sscanf(head, "-pfile_header version %d size %lu", &version, &hsize) != 2) fail(path, "not a pfile (no -pfile_header line)");
    if (version != 0) fail(path, "pfile version " + std::to_string(version) + " is not supported (only 0)");
    if (hsize < 64 || hsize > (1u << 24)) fail(path, "implausible pfile header size");
    info_.header_bytes = (uint32_t)hsize;
    std::vector<char> hdr(hsize + 1, 0);
    fh_.seek(0);
    fh_.read(hdr.data(), hsize);
    std::istringstream is(std::string(hdr.data(), strnlen(hdr.data(), hsize)));
    std::string line;
    bool ended = false, have_data = false;
    std::string format;
    while (getline(is, line)) {
      std::istringstream ls(line);
      std::string key;
      ls >> key;
      if (key == "-end") { ended = true; break; }
      if (key == "-num_sentences") ls >> info_.n_sents;
      else if (key == "-num_frames") ls >> info_.n_frames;
      else if (key == "-first_feature_column") ls >> info_.first_ftr_col;
      else if (key == "-num_features") ls >> info_.n_ftrs;
      else if (key == "-first_label_column") ls >> info_.first_lab_col;
      else if (key == "-num_labels") ls >> info_.n_labs;
      else if (key == "-format") ls >> format;
      else if (key == "-data" || key == "-sent_table_data") {
        std::map<std::string, uint64_t> kv;
        std::string k;
        uint64_t v;
        while (ls >> k >> v) kv[k] = v;
        if (key == "-data") {
          have_data = true;
          info_.data_off = kv["offset"];
          if (kv.count("ncol")) info_.n_cols = (uint32_t)kv["ncol"];
          if (kv.count("nrow") && kv["nrow"] != info_.n_frames && info_.n_frames) fail(path, "-data nrow disagrees with -num_frames");
        } else {
          info_.sent_table_off = (int64_t)kv["offset"];
          if (kv["size"] != (uint64_t)info_.n_sents + 1) fail(path, "-sent_table_data size is not num_sentences+1");
        }
      }
    }
    if (!ended || !have_data) fail(path, "pfile header has no -data / -end line");
    if (info_.n_cols != 2 + info_.n_ftrs + info_.n_labs) fail(path, "pfile ncol is not 2 + num_features + num_labels");
    if (info_.n_ftrs && info_.first_ftr_col != 2) fail(path, "pfile features must start at column 2");
    if (info_.n_labs && info_.first_lab_col != 2 + info_.n_ftrs) fail(path, "pfile labels must follow the features");
    if (!format.empty()) {
      std::string want = "dd" + std::string(info_.n_ftrs, 'f') + std::string(info_.n_labs, 'd');
      if (format != want) fail(path, "pfile -format '" + format + "' is not sentence,frame,floats,ints");
    }
    const uint64_t need = (uint64_t)info_.header_bytes + 4 * (info_.data_off + (uint64_t)info_.n_frames * info_.n_cols);
    if (fh_.size() < need) fail(path, "pfile is shorter than its header says");
    load_index();
  }
  const PFileInfo& info() const { return info_; }
  uint32_t num_sents() const { return info_.n_sents; }
  uint32_t num_frames(uint32_t s) const { return start_.at(s + 1) - start_.at(s); }
  // features [T][n_ftrs] (columns first..first+count) and labels [T][n_labs] of sentence s
  void read_sent(uint32_t s, std::vector<float>* ftrs, std::vector<uint32_t>* labs, uint32_t first = 0, uint32_t count = ~0u) {
    if (s >= info_.n_sents) fail(fh_.path, "sentence " + std::to_string(s) + " out of range");
    if (first > info_.n_ftrs) fail(fh_.path, "first feature beyond the file's width");
    if (count == ~0u) count = info_.n_ftrs - first;
    if (first + count > info_.n_ftrs) fail(fh_.path, "feature range beyond the file's width");
    const uint32_t T = num_frames(s), C = info_.n_cols;
    buf_.resize((size_t)T * C * 4);
    fh_.seek((uint64_t)info_.header_bytes + 4 * (info_.data_off + (uint64_t)start_[s] * C));
    fh_.read(buf_.data(), buf_.size());
    if (ftrs) ftrs->resize((size_t)T * count);
    if (labs) labs->resize((size_t)T * info_.n_labs);
    for (uint32_t t = 0; t < T; t++) {
      const unsigned char* row = buf_.data() + (size_t)t * C * 4;
      if (be32(row) != s || be32(row + 4) != t) fail(fh_.path, "row " + std::to_string(start_[s] + t) + " is not (sentence " + std::to_string(s) + ", frame " + std::to_string(t) + ")");
      if (ftrs)
        for (uint32_t k = 0; k < count; k++) {
          const uint32_t w = be32(row + 4 * (2 + first + k));
          memcpy(&(*ftrs)[(size_t)t * count + k], &w, 4);
        }
      if (labs)
        for (uint32_t k = 0; k < info_.n_labs; k++) (*labs)[(size_t)t * info_.n_labs + k] = be32(row + 4 * (2 + info_.n_ftrs + k));
    }
  }

 private:
  void load_index() {
    start_.assign(info_.n_sents + 1, 0);
    if (info_.sent_table_off >= 0) {
      std::vector<unsigned char> raw((size_t)(info_.n_sents + 1) * 4);
      fh_.seek((uint64_t)info_.header_bytes + 4 * (uint64_t)info_.sent_table_off);
      fh_.read(raw.data(), raw.size());
      for (uint32_t s = 0; s <= info_.n_sents; s++) start_[s] = be32(raw.data() + 4 * s);
    } else {  // no index: one pass over the sentence column
      const uint32_t C = info_.n_cols;
      std::vector<unsigned char> row((size_t)C * 4);
      fh_.seek((uint64_t)info_.header_bytes + 4 * info_.data_off);
      uint32_t cur = 0;
      for (uint32_t r = 0; r < info_.n_frames; r++) {
        fh_.read(row.data(), row.size());
        const uint32_t s = be32(row.data());
        if (s >= info_.n_sents || s < cur) fail(fh_.path, "sentence column is not ascending");
        while (cur < s) start_[++cur] = r;
      }
      while (cur < info_.n_sents) start_[++cur] = info_.n_frames;
    }
    if (start_[0] != 0 || start_[info_.n_sents] != info_.n_frames) fail(fh_.path, "sentence index does not span the data");
    for (uint32_t s = 0; s < info_.n_sents; s++)
      if (start_[s + 1] < start_[s]) fail(fh_.path, "sentence index is not ascending");
  }
  File fh_;
  PFileInfo info_;
  std::vector<uint32_t> start_;
  std::vector<unsigned char> buf_;
};

class PFileWriter {
 public:
  PFileWriter(const std::string& path, uint32_t n_ftrs, uint32_t n_labs) : fh_(path, "wb"), n_ftrs_(n_ftrs), n_labs_(n_labs) {
    std::vector<char> z(kHeader, 0);
    fh_.write(z.data(), z.size());
    start_.push_back(0);
  }
  void write_sent(const float* ftrs, const uint32_t* labs, uint32_t T) {
    const uint32_t C = 2 + n_ftrs_ + n_labs_, s = (uint32_t)start_.size() - 1;
    std::vector<unsigned char> row((size_t)C * 4);
    for (uint32_t t = 0; t < T; t++) {
      put_be32(row.data(), s);
      put_be32(row.data() + 4, t);
      for (uint32_t k = 0; k < n_ftrs_; k++) {
        uint32_t w;
        memcpy(&w, &ftrs[(size_t)t * n_ftrs_ + k], 4);
        put_be32(row.data() + 4 * (2 + k), w);
      }
      for (uint32_t k = 0; k < n_labs_; k++) put_be32(row.data() + 4 * (2 + n_ftrs_ + k), labs[(size_t)t * n_labs_ + k]);
      fh_.write(row.data(), row.size());
    }
    start_.push_back(start_.back() + T);
  }
  void close() {
    if (closed_) return;
    closed_ = true;
    const uint32_t C = 2 + n_ftrs_ + n_labs_, S = (uint32_t)start_.size() - 1, N = start_.back();
    std::vector<unsigned char> idx((size_t)(S + 1) * 4);
    for (uint32_t s = 0; s <= S; s++) put_be32(idx.data() + 4 * s, start_[s]);
    fh_.write(idx.data(), idx.size());
    std::ostringstream h;
    h << "-pfile_header version 0 size " << kHeader << "\n"
      << "-num_sentences " << S << "\n-num_frames " << N << "\n"
      << "-first_feature_column 2\n-num_features " << n_ftrs_ << "\n"
      << "-first_label_column " << 2 + n_ftrs_ << "\n-num_labels " << n_labs_ << "\n"
      << "-format dd" << std::string(n_ftrs_, 'f') << std::string(n_labs_, 'd') << "\n"
      << "-data size " << (uint64_t)N * C << " offset 0 ndim 2 nrow " << N << " ncol " << C << "\n"
      << "-sent_table_data size " << S + 1 << " offset " << (uint64_t)N * C << " ndim 1\n-end\n";
    const std::string hs = h.str();
    if (hs.size() >= kHeader) fail(fh_.path, "pfile header overflow (too many columns)");
    fh_.seek(0);
    fh_.write(hs.data(), hs.size());
    fflush(fh_.f);
  }
  ~PFileWriter() { try { close(); } catch (...) {} }

 private:
  static const uint32_t kHeader = 32768;
  File fh_;
  uint32_t n_ftrs_, n_labs_;
  std::vector<uint32_t> start_;
  bool closed_ = false;
};

// ---------------------------------------------------------------------------------------------
// ILAB (as found in demo/timit-aux/timit_train.48labs.ilab), all integers big-endian:
//   bytes 0..3   "ILAB"
//   7 words      version 19990304 | 28 (bytes of this block = offset of the first record from
//                byte 4) | file offset of the index (= 32 + record bytes) | label width in bits
//                (8) | sentences | frames | 0
//   records      per sentence: (count, label u8) runs, then a 0 count byte followed by the
//                32-bit number of the NEXT sentence (s+1; the last record ends with the 0 alone).  count: one byte 1..127, or two bytes
//                `0x80|hi, lo` for 128..32767 (sentence 563 of the fixture opens with 80 a5 00 =
//                165 frames of label 0)
//   index        sentences words: offset of each record from byte 4; then sentences words: frames
//                of each sentence
// Only the 8-bit label width is pinned by a fixture (with labels 0..47); other widths are refused.
// ---------------------------------------------------------------------------------------------
struct ILabFile {
  uint32_t version = 19990304, label_bits = 8, n_sents = 0, n_frames = 0;
  std::vector<std::vector<uint32_t> > labels;  // [sentence][frame]
};

inline ILabFile read_ilab(const std::string& path) {
  File fh(path, "rb");
  const uint64_t size = fh.size();
  if (size < 32) fail(path, "not an ILAB file (too short)");
  std::vector<unsigned char> d(size);
  fh.read(d.data(), size);
  if (memcmp(d.data(), "ILAB", 4)) fail(path, "not an ILAB file (bad magic)");
  ILabFile out;
  out.version = be32(&d[4]);
  const uint32_t hdr = be32(&d[8]), idx_off = be32(&d[12]);
  out.label_bits = be32(&d[16]);
  out.n_sents = be32(&d[20]);
  out.n_frames = be32(&d[24]);
  if (out.version != 19990304) fail(path, "ILAB version " + std::to_string(out.version) + " is not supported");
  if (hdr != 28) fail(path, "unexpected ILAB header length");
  if (out.label_bits != 8) fail(path, "ILAB label width " + std::to_string(out.label_bits) + " bits is not supported (only 8)");
  const uint64_t rec0 = 4 + (uint64_t)hdr, idx0 = idx_off;
  if (idx0 < rec0 || idx0 + 8ull * out.n_sents != size) fail(path, "ILAB section sizes do not add up to the file size");
  out.labels.resize(out.n_sents);
  uint64_t at = rec0, total = 0;
  for (uint32_t s = 0; s < out.n_sents; s++) {
    if (be32(&d[idx0 + 4ull * s]) + 4ull != at) fail(path, "ILAB index offset of sentence " + std::to_string(s) + " is wrong");
    const uint32_t want = be32(&d[idx0 + 4ull * (out.n_sents + s)]);
    std::vector<uint32_t>& v = out.labels[s];
    v.reserve(want);
    for (;;) {
      if (at + 1 > idx0) fail(path, "ILAB record runs past the record section");
      uint32_t cnt = d[at];
      if (cnt == 0) {
        if (s + 1 == out.n_sents) { at += 1; break; }  // the last record ends with the bare 0
        if (at + 5 > idx0 || be32(&d[at + 1]) != s + 1) fail(path, "ILAB record of sentence " + std::to_string(s) + " has a bad terminator");
        at += 5;
        break;
      }
      if (cnt & 0x80) {
        if (at + 3 > idx0) fail(path, "ILAB record runs past the record section");
        cnt = ((cnt & 0x7f) << 8) | d[at + 1];
        at++;
      }
      if (at + 2 > idx0) fail(path, "ILAB record runs past the record section");
      v.insert(v.end(), cnt, (uint32_t)d[at + 1]);
      at += 2;
    }
    if (v.size() != want) fail(path, "ILAB sentence " + std::to_string(s) + " decodes to " + std::to_string(v.size()) + " frames, index says " + std::to_string(want));
    total += want;
  }
  if (at != idx0) fail(path, "ILAB records do not fill the record section");
  if (total != out.n_frames) fail(path, "ILAB frame total disagrees with the header");
  return out;
}

inline void write_ilab(const std::string& path, const std::vector<std::vector<uint32_t> >& labels) {
  std::vector<unsigned char> rec;
  std::vector<uint32_t> off, cnt;
  uint64_t total = 0;
  for (size_t s = 0; s < labels.size(); s++) {
    off.push_back(28 + (uint32_t)rec.size());
    cnt.push_back((uint32_t)labels[s].size());
    total += labels[s].size();
    for (size_t t = 0; t < labels[s].size();) {
      if (labels[s][t] > 255) fail(path, "label does not fit the 8-bit ILAB width");
      size_t e = t;
      while (e < labels[s].size() && labels[s][e] == labels[s][t] && e - t < 32767) e++;
      if (e - t > 127) rec.push_back((unsigned char)(0x80 | ((e - t) >> 8)));
      rec.push_back((unsigned char)((e - t) & 0xff));
      rec.push_back((unsigned char)labels[s][t]);
      t = e;
    }
    unsigned char term[5] = {0};
    put_be32(term + 1, (uint32_t)s + 1);
    rec.insert(rec.end(), term, term + (s + 1 == labels.size() ? 1 : 5));
  }
  unsigned char hdr[32];
  memcpy(hdr, "ILAB", 4);
  const uint32_t w[7] = {19990304, 28, 32 + (uint32_t)rec.size(), 8, (uint32_t)labels.size(), (uint32_t)total, 0};
  for (int i = 0; i < 7; i++) put_be32(hdr + 4 + 4 * i, w[i]);
  File fh(path, "wb");
  fh.write(hdr, 32);
  fh.write(rec.data(), rec.size());
  std::vector<unsigned char> idx(8 * labels.size());
  for (size_t s = 0; s < labels.size(); s++) {
    put_be32(&idx[4 * s], off[s]);
    put_be32(&idx[4 * (labels.size() + s)], cnt[s]);
  }
  fh.write(idx.data(), idx.size());
}

inline bool is_ilab(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  char m[4] = {0};
  const size_t n = fread(m, 1, 4, f);
  fclose(f);
  return n == 4 && !memcmp(m, "ILAB", 4);
}

// ---------------------------------------------------------------------------------------------
// Sentence ranges (`train_sent_range`, `cv_sent_range`: "QN_Range(3) format",
// CRFTrain/src/Main.cpp:209-210).  The subset the reference's scripts use: `all`, `nil`/`none`,
// comma- or space-separated terms `n`, `a:b` / `a-b` (inclusive), `a:step:b`, with `^k` counting
// from the last sentence (`^0`).  QuickNet's parser is not in the tree: parity UNPINNED.
// ---------------------------------------------------------------------------------------------
inline std::vector<uint32_t> parse_range(const std::string& spec, uint32_t n) {
  std::vector<uint32_t> out;
  std::string s = spec;
  for (char& c : s)
    if (c == ',' || c == ';') c = ' ';
  std::istringstream is(s);
  std::string term;
  auto endpoint = [&](const std::string& t) -> long {
    if (t.empty()) throw std::runtime_error("bad range term in '" + spec + "'");
    const bool from_end = t[0] == '^';
    char* e = nullptr;
    const long v = strtol(t.c_str() + (from_end ? 1 : 0), &e, 10);
    if (*e) throw std::runtime_error("bad range term '" + t + "' in '" + spec + "'");
    return from_end ? (long)n - 1 - v : v;
  };
  while (is >> term) {
    if (term == "all") { for (uint32_t i = 0; i < n; i++) out.push_back(i); continue; }
    if (term == "nil" || term == "none") continue;
    std::vector<std::string> parts;
    size_t at = 0;
    for (size_t i = 1; i <= term.size(); i++)  // a '-' or ':' after the first character separates
      if (i == term.size() || term[i] == ':' || (term[i] == '-' && term[i - 1] != ':')) { parts.push_back(term.substr(at, i - at)); at = i + 1; }
    long a, b, step = 1;
    if (parts.size() == 1) a = b = endpoint(parts[0]);
    else if (parts.size() == 2) { a = endpoint(parts[0]); b = endpoint(parts[1]); }
    else if (parts.size() == 3) { a = endpoint(parts[0]); step = endpoint(parts[1]); b = endpoint(parts[2]); }
    else throw std::runtime_error("bad range term '" + term + "' in '" + spec + "'");
    if (step <= 0) throw std::runtime_error("range step must be positive in '" + spec + "'");
    for (long v = a; v <= b; v += step) {
      if (v < 0 || v >= (long)n) throw std::runtime_error("range '" + spec + "' selects sentence " + std::to_string(v) + " of " + std::to_string(n));
      out.push_back((uint32_t)v);
    }
  }
  return out;
}

}  // namespace qn
#endif  // QN_FILES_H_
