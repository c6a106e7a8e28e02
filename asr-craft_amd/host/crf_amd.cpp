// crf_amd.cpp -- implementation of the reference-named host classes on the MI355X engine.
#include "crf_amd.h"

#include <math.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <iterator>
#include <limits>
#include <memory>
#include <random>
#include <chrono>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

#include "ftr_files.h"
#include "lbfgs.h"

using std::runtime_error;
using std::string;

// ------------------------------------------------------------------------------------------
// CRF_FeatureMap: lambda layout of the dense maps, one state or n states per label
// (same closed form as ScrfLayout; ftrmaps/CRF_StdFeatureMap.cpp:280-320,355-410,472-517)
// ------------------------------------------------------------------------------------------
CRF_FeatureMap::CRF_FeatureMap(CRF_FeatureMap_config* cnf) : config(cnf) { recalc(); }

CRF_FeatureMap* CRF_FeatureMap::createFeatureMap(CRF_FeatureMap_config* cnf) {
  if (cnf->map_type != STDSTATE && cnf->map_type != STDTRANS)
    throw runtime_error("createFeatureMap: only the dense stdstate/stdtrans maps are built");
  return new CRF_FeatureMap(cnf);
}

QNUInt32 CRF_FeatureMap::recalc() {
  const QNUInt32 L = config->numLabs, K = config->numStates;
  if (K == 0) throw runtime_error("CRF_StdFeatureMap created exception: numStates is 0");
  numActualLabels = L / K;
  if (numActualLabels * K != L) throw runtime_error("CRF_StdFeatureMap created exception: Invalid state/label combination while computing transitions");
  numStateFuncs = numTransFuncs = 0;
  if (config->useStateFtrs) numStateFuncs += config->stateFidxEnd - config->stateFidxStart + 1;
  if (config->useStateBias) numStateFuncs += 1;
  if (config->useTransFtrs) numTransFuncs += config->transFidxEnd - config->transFidxStart + 1;
  if (config->useTransBias) numTransFuncs += 1;
  // end->start transitions + diagonal self transitions + off-diagonal transitions (CRF_StdFeatureMap.cpp:480-485)
  const QNUInt32 transMult = K == 1 ? L * L : numActualLabels * numActualLabels + L + L - numActualLabels;
  numFtrFuncs = L * numStateFuncs + transMult * numTransFuncs;
  return numFtrFuncs;
}

QNUInt32 CRF_FeatureMap::getStateFeatureIdx(QNUInt32 clab, QNUInt32 fno) {
  const QNUInt32 K = config->numStates;
  if (K == 1) return clab * (numStateFuncs + config->numLabs * numTransFuncs) + fno;
  const QNUInt32 ns = (clab + K - 1) / K;   // start states among the labels before clab (:293-312)
  return clab * numStateFuncs + numTransFuncs * (ns * (numActualLabels + 1) + (clab - ns) * 2) + fno;
}

QNUInt32 CRF_FeatureMap::getTransFeatureIdx(QNUInt32 clab, QNUInt32 plab, QNUInt32 fno) {
  const QNUInt32 K = config->numStates;
  if (K == 1) return clab * (numStateFuncs + config->numLabs * numTransFuncs) + numStateFuncs + plab * numTransFuncs + fno;
  QNUInt32 v = getStateFeatureIdx(clab) + numStateFuncs;   // :367-407; (QNUInt32)-1 = no such transition
  if (plab != clab) {
    v += numTransFuncs;
    if (clab % K == 0) {
      if ((plab + 1) % K != 0) return (QNUInt32)-1;
      v += (plab / K) * numTransFuncs;
    } else if (plab != clab - 1) {
      return (QNUInt32)-1;
    }
  }
  return v + fno;
}

// ------------------------------------------------------------------------------------------
// crf_amd::Engine
// ------------------------------------------------------------------------------------------
namespace crf_amd {

Engine::Engine(const scrf_config& cfg) {
  int rc = scrf_create(&cfg, &h);
  if (rc != SCRF_OK) throw runtime_error(string("scrf_create: ") + scrf_last_error(nullptr));
  scrf_lambda_len(h, &lambda_len);
}
Engine::~Engine() { scrf_destroy(h); }
void Engine::check(int rc, const char* what) {
  if (rc != SCRF_OK) throw runtime_error(string(what) + " caught exception: " + scrf_last_error(h));
}

// a training-only engine takes the n-state frame model through the masked dense layout (CRF_Model::setTrainingOnly)
bool frameAsSegmental(CRF_Model* crf, uint32_t precision) {
  const CRF_FeatureMap_config* c = crf->getFeatureMap()->getConfig();
  return crf->trainingOnly() && precision != SCRF_PREC_EXACT && crf->getModelType() == STDFRAME && c->numStates > 1 && crf->getLabMaxDur() == 1;
}

scrf_config makeConfig(CRF_Model* crf, int device, uint32_t precision) {
  CRF_FeatureMap* fm = crf->getFeatureMap();
  if (!fm) throw runtime_error("CRF_Model has no feature map");
  const CRF_FeatureMap_config* c = fm->getConfig();
  scrf_config g;
  memset(&g, 0, sizeof(g));
  g.abi_version = SCRF_ABI_VERSION;
  g.model_type = (uint32_t)crf->getModelType();
  g.map_type = c->map_type == STDTRANS ? SCRF_STDTRANS : SCRF_STDSTATE;
  g.num_labs = c->numLabs;
  g.num_feas = c->numFeas;
  g.num_states = c->numStates;
  g.lab_max_dur = crf->getLabMaxDur();
  g.use_state_ftrs = c->useStateFtrs;
  g.state_fidx_start = c->stateFidxStart;
  g.state_fidx_end = c->stateFidxEnd;
  g.use_trans_ftrs = c->useTransFtrs;
  g.trans_fidx_start = c->transFidxStart;
  g.trans_fidx_end = c->transFidxEnd;
  g.use_state_bias = c->useStateBias;
  g.use_trans_bias = c->useTransBias;
  g.state_bias_val = c->stateBiasVal;
  g.trans_bias_val = c->transBiasVal;
  g.device_id = device;
  g.train_precision = precision;
  if (frameAsSegmental(crf, precision)) g.model_type = SCRF_STDSEG_NO_DUR_NO_SEGTRANSFTR;
  return g;
}

}  // namespace crf_amd

// ------------------------------------------------------------------------------------------
// CRF_Model
// ------------------------------------------------------------------------------------------
CRF_Model::CRF_Model(QNUInt32 num_labs) : nlabs(num_labs), nActualLabs(num_labs) {}
CRF_Model::~CRF_Model() { delete featureMap; }

void CRF_Model::setFeatureMap(CRF_FeatureMap* map) {
  delete featureMap;
  featureMap = map;
  const QNUInt32 n = map->getNumFtrFuncs();
  lambda.assign(n, 0.0);
  lambdaAcc.assign(n, 0.0);
  gradSqrAcc.assign(n, 0.0);
  eng.reset();
}

void CRF_Model::setLambda(double* v, QNUInt32 n) {
  if (n != lambda.size()) throw runtime_error("CRF_Model::setLambda: length mismatch");
  std::copy(v, v + n, lambda.begin());
}
void CRF_Model::resetLambda() { std::fill(lambda.begin(), lambda.end(), 0.0); }

// one value per line in the stream default format (6 significant digits, "%g"), as the reference's
// `ofile << lambda[i] << endl` (CRF_Model.cpp); buffered -- a stdtrans TIMIT model has 4.4 M lines
// (round 4: the text of large vectors is formatted by a few threads, slice by slice, and written in order -- the same
// bytes as one fprintf per value; the TIMIT demo wrote its three 4.4 M-line files per epoch in 1.06 s of a 2.8 s epoch)
static bool write_vec(const char* fname, const double* v, size_t n) {
  FILE* o = fopen(fname, "w");
  if (!o) throw runtime_error(string("CRF_Model::writeToFile() caught exception: cannot open the file:\n") + fname);
  std::vector<char> buf(1 << 20);
  setvbuf(o, buf.data(), _IOFBF, buf.size());
  bool ok = true;
  if (n >= (1u << 16)) {
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 1 : (nt > 8 ? 8 : nt);
    const size_t per = (n + nt - 1) / nt;
    std::vector<std::string> part(nt);
    std::vector<std::thread> th;
    for (unsigned k = 0; k < nt; k++)
      th.emplace_back([&, k]() {
        const size_t a = k * per, b = std::min(n, a + per);
        std::string& out = part[k];
        out.reserve((b > a ? b - a : 0) * 14);
        char line[64];
        for (size_t i = a; i < b; i++) {
          const int m = snprintf(line, sizeof line, "%g\n", v[i]);
          out.append(line, (size_t)m);
        }
      });
    for (auto& t : th) t.join();
    for (unsigned k = 0; k < nt && ok; k++) ok = part[k].empty() || fwrite(part[k].data(), 1, part[k].size(), o) == part[k].size();
  } else {
    for (size_t i = 0; i < n && ok; i++) ok = fprintf(o, "%g\n", v[i]) > 0;
  }
  ok = fclose(o) == 0 && ok;
  if (!ok) throw runtime_error(string("CRF_Model::writeToFile() caught exception: errors when writing the weights to the file:\n") + fname);
  return true;
}
static bool read_vec(const char* fname, double* v, size_t n, double scale) {
  FILE* f = fopen(fname, "r");
  if (!f) return false;
  char line[512];
  for (size_t i = 0; i < n; i++) {
    double x = 0.0;  // a missing or unparsable line reads as 0, like `iss >> v[i]` on a failed stream
    if (fgets(line, sizeof line, f)) {
      char* e = nullptr;
      x = strtod(line, &e);
      if (e == line) x = 0.0;
    }
    v[i] = scale != 1.0 ? x * scale : x;
  }
  fclose(f);
  return true;
}
bool CRF_Model::writeToFile(const char* fname) { return write_vec(fname, lambda.data(), lambda.size()); }
bool CRF_Model::writeToFile(const char* fname, double* lam, QNUInt32 ll) { return write_vec(fname, lam, ll); }
bool CRF_Model::readFromFile(const char* fname) { return read_vec(fname, lambda.data(), lambda.size(), 1.0); }
bool CRF_Model::readAverageFromFile(const char* fname, int present) {
  init_present = present;
  return read_vec(fname, lambdaAcc.data(), lambdaAcc.size(), present > 0 ? (double)present : 1.0);
}
bool CRF_Model::readGradSqrAccFromFile(const char* fname) { return read_vec(fname, gradSqrAcc.data(), gradSqrAcc.size(), 1.0); }

// ---- one process per GPU: the RCCL unique id goes from rank 0 to the others through a file ----
// The file holds the 128-byte id followed by a LAUNCH TOKEN, and a rank accepts only a file that carries its own
// token: a leftover of an earlier run is never mistaken for this launch's id, whatever the clocks say and however far
// apart the ranks start.  The token is SCRF_LAUNCH_TOKEN when the launcher provides one, else what one launch's
// processes share on a node: MASTER_ADDR, MASTER_PORT, TORCHELASTIC_RUN_ID, TORCHELASTIC_RESTART_COUNT and the
// launcher's process id (the parent of every rank) -- the parent only when none of the launcher variables is set: ranks
// started by per-node daemons, wrapper scripts or on several nodes do not share a parent (there, MASTER_PORT / the run
// id identify the launch, or SCRF_LAUNCH_TOKEN must be given).  Waiting time: SCRF_COMM_TIMEOUT_S seconds (120).
namespace {
double wall_now() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
string launch_token() {
  if (const char* t = getenv("SCRF_LAUNCH_TOKEN")) return string("token:") + t;
  string t = "launch";
  bool named = false;
  for (const char* k : {"MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT"}) {
    const char* v = getenv(k);
    t += string(":") + (v ? v : "-");
    if (v && (string(k) == "MASTER_PORT" || string(k) == "TORCHELASTIC_RUN_ID")) named = true;
  }
  return named ? t : t + ":ppid" + std::to_string((long)getppid());
}

void exchange_comm_id(int rank, const string& path, unsigned char id[128]) {
  const string token = launch_token();
  if (rank == 0) {
    remove(path.c_str());   // whatever an earlier run left behind
    if (scrf_comm_unique_id(id) != SCRF_OK) throw runtime_error(string("scrf_comm_unique_id: ") + scrf_last_error(nullptr));
    const string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, 128, f) != 128 || fwrite(token.data(), 1, token.size(), f) != token.size() || fclose(f) != 0)
      throw runtime_error("cannot write the communicator id file " + tmp);
    if (rename(tmp.c_str(), path.c_str()) != 0) throw runtime_error("cannot publish the communicator id file " + path);
    return;
  }
  const char* to = getenv("SCRF_COMM_TIMEOUT_S");
  const double deadline = wall_now() + (to && atof(to) > 0 ? atof(to) : 120.0);
  std::vector<char> buf(128 + token.size() + 1);
  string seen;   // the token of the last file that did not match (for the message)
  while (wall_now() < deadline) {
    FILE* f = fopen(path.c_str(), "rb");
    if (f) {
      const size_t n = fread(buf.data(), 1, buf.size(), f);   // one byte more than expected: a longer token is a mismatch
      fclose(f);
      if (n == 128 + token.size() && memcmp(buf.data() + 128, token.data(), token.size()) == 0) {
        memcpy(id, buf.data(), 128);
        return;
      }
      if (n > 128) seen.assign(buf.data() + 128, n - 128);
    }
    usleep(20000);
  }
  throw runtime_error("timed out waiting for rank 0's communicator id file " + path + " (this rank's launch token: " + token +
                      (seen.empty() ? "; no file with a token was found" : "; the file there carries: " + seen + (seen.size() > token.size() ? "..." : "")) +
                      "; set SCRF_LAUNCH_TOKEN to the same value on every rank when the ranks do not share MASTER_PORT / TORCHELASTIC_RUN_ID or a parent process)");
}
}  // namespace

void CRF_Model::setDistributed(int rank, int world, const string& id_file) {
  if (world < 1 || rank < 0 || rank >= world) throw runtime_error("CRF_Model::setDistributed: rank outside [0, world)");
  if (id_file.empty()) throw runtime_error("CRF_Model::setDistributed: a communicator id file is required");
  if (eng) throw runtime_error("CRF_Model::setDistributed: call it before the engine is first used");
  dist_on = true;
  dist_rank = rank;
  dist_world = world;
  dist_id_file = id_file;
}

crf_amd::Engine* CRF_Model::engine() {
  if (!eng) {
    eng.reset(new crf_amd::Engine(crf_amd::makeConfig(this, device, precision)));
    // the recast frame model keeps the frame node's posterior-mass bounds (CRF_StdStateNode.cpp:252-275)
    if (crf_amd::frameAsSegmental(this, precision)) eng->check(scrf_set_frame_mass_check(eng->h, 1), "scrf_set_frame_mass_check");
    if (dist_on) {
      unsigned char id[128];
      exchange_comm_id(dist_rank, dist_id_file, id);
      eng->check(scrf_comm_init(eng->h, id, dist_rank, dist_world), "scrf_comm_init");   // collective: every rank has the id now
      if (dist_rank == 0) remove(dist_id_file.c_str());
    }
  }
  return eng.get();
}
void CRF_Model::pushLambda() {
  crf_amd::Engine* e = engine();
  const QNUInt32 n = (QNUInt32)lambda.size();
  e->check(scrf_set_lambda(e->h, lambda.data(), n), "CRF_Model::pushLambda");
  e->check(scrf_set_lambda_acc(e->h, lambdaAcc.data(), n), "CRF_Model::pushLambda");
  e->check(scrf_set_grad_sqr_acc(e->h, gradSqrAcc.data(), n), "CRF_Model::pushLambda");
}
void CRF_Model::pullLambda() {
  crf_amd::Engine* e = engine();
  const QNUInt32 n = (QNUInt32)lambda.size();
  e->check(scrf_get_lambda(e->h, lambda.data(), n), "CRF_Model::pullLambda");
  e->check(scrf_get_lambda_acc(e->h, lambdaAcc.data(), n), "CRF_Model::pullLambda");
  e->check(scrf_get_grad_sqr_acc(e->h, gradSqrAcc.data(), n), "CRF_Model::pullLambda");
}

// ------------------------------------------------------------------------------------------
// CRF_MemoryFeatureStream
// ------------------------------------------------------------------------------------------
static uint32_t recipe_width(const scrf_stream_recipe& r, uint32_t D) {
  if (D == 1) return (r.left_ctx + 1 + r.right_ctx) * r.in_width;
  if (r.extract_seg_ftr) return 8 * r.in_width + D + (r.left_ctx + r.right_ctx) * r.in_width;
  return (r.left_ctx + 1 + r.right_ctx) * r.in_width;
}

CRF_MemoryFeatureStream::CRF_MemoryFeatureStream(std::vector<scrf_stream_recipe> recipes, QNUInt32 max_dur,
                                                 QNUInt32 n_actual_labs)
    : store_(new Store()) {
  (void)n_actual_labs;   // labels are formed for the model's nActualLabs when a batch is assembled
  store_->recipes = recipes;
  store_->D = max_dur;
  for (const auto& r : recipes) width_ += recipe_width(r, max_dur);
}
CRF_MemoryFeatureStream::~CRF_MemoryFeatureStream() { if (win_eng_) scrf_destroy(win_eng_); }

void CRF_MemoryFeatureStream::addUtterance(const std::vector<std::vector<float> >& frames,
                                           const std::vector<uint32_t>& fl) {
  Store& s = *store_;
  if (frames.size() != s.recipes.size()) throw runtime_error("addUtterance: one frame matrix per stream expected");
  const scrf_stream_recipe& r0 = s.recipes[0];
  const uint32_t T = (uint32_t)(frames[0].size() / r0.in_width) - r0.left_ctx - r0.right_ctx;
  for (size_t k = 0; k < frames.size(); k++) {
    const scrf_stream_recipe& r = s.recipes[k];
    if (frames[k].size() != (size_t)(T + r.left_ctx + r.right_ctx) * r.in_width)
      throw runtime_error("addUtterance: stream lengths disagree");
  }
  s.T.push_back(T);
  s.frames.push_back(frames);
  // frame labels -> (phone, start) at segment end frames; runs longer than D are split evenly
  // (io/CRF_InLabStream_SeqMultiWindow.cpp:51-110)
  std::vector<uint32_t> ph(T, CRF_LAB_BAD), st(T, CRF_LAB_BAD);
  if (!fl.empty()) {
    if (fl.size() != T) throw runtime_error("addUtterance: one label per frame expected");
    const uint32_t D = s.D;
    for (uint32_t a = 0; a < T;) {
      uint32_t b = a + 1;
      while (b < T && fl[b] == fl[a]) b++;
      if (fl[a] != CRF_LAB_BAD) {
        const uint32_t dur = b - a;
        const uint32_t pieces = dur <= D ? 1 : (dur % D == 0 ? dur / D : dur / D + 1);
        const uint32_t pd = dur / pieces, rem = dur % pieces;
        uint32_t ps = a;
        for (uint32_t p = 0; p < pieces; p++) {
          const uint32_t d = p < rem ? pd + 1 : pd;
          ph[ps + d - 1] = fl[a];
          st[ps + d - 1] = ps;
          ps += d;
        }
      }
      a = b;
    }
  }
  s.seg_phone.push_back(ph);
  s.seg_start.push_back(st);
  end_ = s.T.size();
}

void CRF_MemoryFeatureStream::addPlaceholder() {
  Store& s = *store_;
  s.T.push_back(0);
  s.frames.push_back(std::vector<std::vector<float> >(s.recipes.size()));
  s.seg_phone.push_back(std::vector<uint32_t>());
  s.seg_start.push_back(std::vector<uint32_t>());
  end_ = s.T.size();
}

static int g_view_rank = 0, g_view_world = 1;
void crf_amd::setProcessView(int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) throw runtime_error("crf_amd::setProcessView: rank outside [0, world)");
  g_view_rank = rank;
  g_view_world = world;
}

void CRF_MemoryFeatureStream::join(const CRF_MemoryFeatureStream& other) {
  Store& s = *store_;
  const Store& o = *other.store_;
  if (o.T.size() != s.T.size() || o.D != s.D) throw runtime_error("CRF_FeatureStream::join: the streams hold different utterances");
  for (size_t u = 0; u < s.T.size(); u++) {
    if (o.T[u] != s.T[u]) throw runtime_error("CRF_FeatureStream::join: utterance lengths differ between the streams");
    for (const auto& f : o.frames[u]) s.frames[u].push_back(f);
  }
  for (const auto& r : o.recipes) { s.recipes.push_back(r); width_ += recipe_width(r, s.D); }
  if (s.recipes.size() > SCRF_MAX_STREAMS) throw runtime_error("CRF_FeatureStream::join: more than three input streams");
}

CRF_MemoryFeatureStream* CRF_MemoryFeatureStream::makeView(size_t start, size_t count) {
  CRF_MemoryFeatureStream* v = new CRF_MemoryFeatureStream(*this);
  v->win_eng_ = nullptr;   // the helper engine is not shared
  v->win_utt_ = -1;
  v->view(start, count);
  return v;
}

void CRF_MemoryFeatureStream::view(size_t start, size_t count) {
  begin_ = begin_ + start;
  end_ = (count == CRF_UINT32_MAX || count > end_) ? end_ : std::min(end_, begin_ + count);
  if (begin_ > end_) begin_ = end_;
  cur_ = -1;
  if (mode_ != SEQUENTIAL) { epoch_ = 0; rewind(); }   // its own order over its own range
}

void CRF_MemoryFeatureStream::setPresentation(seqtype type, QNUInt32 seed) {
  mode_ = type;
  seed_ = seed;
  epoch_ = 0;
  rewind();   // the RandPresent constructor rewinds once (:38)
}

QN_SegID CRF_MemoryFeatureStream::nextseg() {
  if (mode_ != SEQUENTIAL) {
    if (pos_ >= order_.size()) { cur_ = (long)end_; return QN_SEGID_BAD; }
    cur_ = (long)(begin_ + order_[pos_++]);
    frame_ = 0;
    return (QN_SegID)(cur_ - (long)begin_);
  }
  long nxt = cur_ < 0 ? (long)begin_ : cur_ + 1;
  if (nxt >= (long)end_) { cur_ = (long)end_; return QN_SEGID_BAD; }
  cur_ = nxt;
  frame_ = 0;
  return (QN_SegID)(cur_ - (long)begin_);
}
int CRF_MemoryFeatureStream::rewind() {
  cur_ = -1;
  frame_ = 0;
  if (mode_ != SEQUENTIAL) {
    epoch_++;
    const size_t n = end_ - begin_;
    std::mt19937_64 gen(12345ull * epoch_ + seed_);
    order_.resize(n);
    if (mode_ == RANDOM_NO_REPLACE) {
      for (size_t i = 0; i < n; i++) order_[i] = i;
      for (size_t i = n; i > 1; i--) std::swap(order_[i - 1], order_[gen() % i]);   // Fisher-Yates, spelled out
    } else {
      for (size_t i = 0; i < n; i++) order_[i] = n ? gen() % n : 0;
    }
    pos_ = 0;
  }
  return 0;
}

bool CRF_MemoryFeatureStream::currentUtterance(Utterance* u) {
  if (cur_ < (long)begin_ || cur_ >= (long)end_) return false;
  const Store& s = *store_;
  u->T = s.T[cur_];
  u->frames.clear();
  for (const auto& f : s.frames[cur_]) u->frames.push_back(f.data());
  u->windows = nullptr;
  u->labels = nullptr;
  u->phones = s.seg_phone[cur_].data();
  u->starts = s.seg_start[cur_].data();
  return true;
}

// read(): the window vectors of the current utterance come from the engine's window kernel (k_windows, the
// reference recipe of io/CRF_InFtrStream_SeqMultiWindow.cpp bit for bit) through a private handle whose
// only job is scrf_windows; the host holds no restatement of the recipe.
void CRF_MemoryFeatureStream::fetchWindows() {
  if (win_utt_ == cur_) return;
  const Store& s = *store_;
  if (!win_eng_) {
    scrf_config g;
    memset(&g, 0, sizeof(g));
    g.abi_version = SCRF_ABI_VERSION;
    g.model_type = s.D == 1 ? SCRF_STDFRAME : SCRF_STDSEG_NO_DUR_NO_SEGTRANSFTR;
    g.map_type = SCRF_STDSTATE;
    g.num_labs = 1; g.num_feas = (uint32_t)width_; g.num_states = 1; g.lab_max_dur = s.D;
    g.use_state_ftrs = 1; g.state_fidx_start = 0; g.state_fidx_end = (uint32_t)width_ - 1;
    g.use_state_bias = 1; g.use_trans_bias = 1; g.state_bias_val = 1.0; g.trans_bias_val = 1.0;
    if (scrf_create(&g, &win_eng_) != SCRF_OK) throw runtime_error(string("CRF_MemoryFeatureStream::read: ") + scrf_last_error(nullptr));
  }
  scrf_utt q;
  memset(&q, 0, sizeof(q));
  q.T = s.T[cur_];
  for (size_t k = 0; k < s.frames[cur_].size() && k < SCRF_MAX_STREAMS; k++) q.frames[k] = s.frames[cur_][k].data();
  scrf_batch b = nullptr;
  if (scrf_batch_create(win_eng_, &q, 1, (uint32_t)s.recipes.size(), s.recipes.data(), &b) != SCRF_OK)
    throw runtime_error(string("CRF_MemoryFeatureStream::read: ") + scrf_last_error(win_eng_));
  uint64_t n_segs = 0;
  scrf_batch_info(win_eng_, b, nullptr, nullptr, &n_segs, nullptr);
  win_cache_.resize((size_t)n_segs * width_);
  const int rc = scrf_windows(win_eng_, b, 0, win_cache_.data());
  scrf_batch_destroy(win_eng_, b);
  if (rc != SCRF_OK) throw runtime_error(string("CRF_MemoryFeatureStream::read: ") + scrf_last_error(win_eng_));
  win_utt_ = cur_;
}

// windows ending at the current frame, d = 1..bunch, joined over the streams (what
// io/CRF_InFtrStream_SeqMultiWindow produces) + the 4 label words of the segment ending there
size_t CRF_MemoryFeatureStream::read(size_t bunch, float* out, QNUInt32* lab_buf) {
  if (cur_ < (long)begin_ || cur_ >= (long)end_) return 0;
  const Store& s = *store_;
  const uint32_t T = s.T[cur_], D = s.D, t = frame_;
  if (t >= T) return 0;
  const uint32_t avail = t + 1 <= D ? t + 1 : D;
  if (bunch != avail) throw runtime_error("CRF_MemoryFeatureStream::read: the number of windows has to be min(t+1, max_dur) at the current frame");
  fetchWindows();
  const size_t row0 = t < D ? (size_t)t * (t + 1) / 2 : (size_t)D * (D + 1) / 2 + (size_t)(t - D) * D;
  memcpy(out, &win_cache_[row0 * width_], sizeof(float) * avail * width_);
  if (lab_buf) {
    const uint32_t ph = s.seg_phone[cur_][t];
    if (ph == CRF_LAB_BAD) {
      lab_buf[0] = lab_buf[1] = lab_buf[2] = lab_buf[3] = CRF_LAB_BAD;
    } else {
      lab_buf[0] = ph;
      lab_buf[1] = s.seg_start[cur_][t];
      lab_buf[2] = t;
      lab_buf[3] = 0;
    }
  }
  frame_++;
  return avail;
}

// ------------------------------------------------------------------------------------------
// CRF_FeatureStreamManager
// ------------------------------------------------------------------------------------------
CRF_FeatureStreamManager::CRF_FeatureStreamManager(int debug, const char* debug_name, char* ftr_fname, const char* ftr_file_fmt,
                                                   char* ht_fname, size_t ht_offset, size_t ftr_width, size_t first_ftr,
                                                   size_t num_ftrs, size_t win_ext, size_t win_off, size_t win_len,
                                                   size_t left_ctx_len, size_t right_ctx_len, bool extract_seg_ftr,
                                                   bool use_bdy_delta_ftr, int delta_o, int delta_w, char* trn_rng, char* cv_rng,
                                                   FILE* nfile, int n_mode, double n_am, double n_av, seqtype ts, QNUInt32 rseed,
                                                   size_t n_threads)
    : nthreads(n_threads ? n_threads : 1) {
  (void)debug; (void)n_mode; (void)n_am; (void)n_av;
  const string who = string("CRF_FeatureStreamManager(") + (debug_name ? debug_name : "") + "): ";
  if (!ftr_fname || !*ftr_fname) throw runtime_error(who + "no feature file");
  if (nfile) throw runtime_error(who + "feature normalisation files are not built");
  if (delta_o != 0) throw runtime_error(who + "delta features are not built");
  (void)delta_w;
  if (use_bdy_delta_ftr) throw runtime_error(who + "boundary delta features are not built");
  if (win_off != 0 || ht_offset != 0) throw runtime_error(who + "window offsets must be 0");
  if (win_off + win_len > win_ext) throw runtime_error("CRF_FeatureStreamManager::create() caught exception: this->window_offset + this->window_len > this->window_extent.");
  if (win_len != win_ext) throw runtime_error(who + "window_len must equal window_extent (the label maximum duration)");
  const string fmt = ftr_file_fmt ? ftr_file_fmt : "pfile";
  FtrData data;
  if (fmt == "pfile") data = read_pfile_ftrs(ftr_fname, (uint32_t)first_ftr, (uint32_t)num_ftrs);
  else if (fmt == "ascii") {
    if (first_ftr || num_ftrs) throw runtime_error(who + "first_ftr / num_ftrs need format pfile");
    data = read_ascii_ftrs(ftr_fname);
  } else throw runtime_error(who + "format " + fmt + " is not built (pfile|ascii)");
  if (ftr_width != 0 && fmt == "ascii" && ftr_width != data.width) throw runtime_error(who + "ftr_width does not match the file");
  std::vector<std::vector<uint32_t> > labs;
  if (ht_fname && *ht_fname) labs = read_labs(ht_fname);
  scrf_stream_recipe r;
  r.in_width = (uint32_t)data.width;
  r.left_ctx = (uint32_t)left_ctx_len;
  r.right_ctx = (uint32_t)right_ctx_len;
  r.extract_seg_ftr = extract_seg_ftr ? 1 : 0;
  const QNUInt32 D = (QNUInt32)win_len;
  auto fill = [&](const char* rng, std::unique_ptr<CRF_MemoryFeatureStream>* dst, bool my_view_only) {
    const std::vector<uint32_t> sents = qn::parse_range(rng ? rng : "all", (uint32_t)data.size());
    dst->reset(new CRF_MemoryFeatureStream(std::vector<scrf_stream_recipe>(1, r), D));
    // crf_amd::setProcessView: this process is stream g_view_rank of nthreads -- the child view it walks (below)
    size_t lo = 0, hi = sents.size();
    if (my_view_only) {
      const size_t per = sents.size() / nthreads;
      lo = (size_t)g_view_rank * per;
      hi = (size_t)g_view_rank == nthreads - 1 ? sents.size() : lo + per;
    }
    for (size_t i = 0; i < sents.size(); i++) {
      const uint32_t u = sents[i];
      if (i < lo || i >= hi) { (*dst)->addPlaceholder(); continue; }
      std::vector<std::vector<float> > fr(1, data.get(u));
      (*dst)->addUtterance(fr, u < labs.size() ? labs[u] : std::vector<uint32_t>());
    }
  };
  // the training stream reads the train range; the CV stream (built like the reference's, read by no SG
  // trainer) the CV range
  fill(trn_rng, &trn, g_view_world > 1 && (size_t)g_view_world == nthreads);
  if (cv_rng && *cv_rng && string(cv_rng) != "nil" && string(cv_rng) != "none") fill(cv_rng, &cv, false);
  for (size_t u = 0; u < data.size(); u++) data.drop(u);
  if (ts != SEQUENTIAL) trn->setPresentation(ts, rseed);
  trn_stream = trn.get();
  cv_stream = cv.get();
  if (nthreads > 1) {
    // child i views [i*floor(n/N), (i+1)*floor(n/N)), the last one to the end (create() :425-464)
    const size_t nseg = trn->numUtterances(), per = nseg / nthreads;
    for (size_t i = 0; i < nthreads; i++) {
      std::unique_ptr<CRF_FeatureStreamManager> c(new CRF_FeatureStreamManager());
      c->trn.reset(trn->makeView(i * per, i == nthreads - 1 ? CRF_UINT32_MAX : per));
      c->trn_stream = c->trn.get();
      children.push_back(std::move(c));
    }
  }
}
CRF_FeatureStreamManager::~CRF_FeatureStreamManager() {}

void CRF_FeatureStreamManager::join(CRF_FeatureStreamManager* other) {
  // feature concatenation of the two managers' streams, children pairwise (:108-122)
  if (!other || !other->trn) throw runtime_error("CRF_FeatureStreamManager::join: nothing to join");
  old_trn_stream = trn_stream;
  trn->join(*other->trn);
  if (cv && other->cv) cv->join(*other->cv);
  // the children share the parent's storage: they see the joined frames, only their width has to follow
  if (children.size() != other->children.size()) throw runtime_error("CRF_FeatureStreamManager::join: thread counts differ");
  for (size_t i = 0; i < children.size(); i++) {
    const size_t nseg = trn->numUtterances(), per = nseg / nthreads;
    children[i]->trn.reset(trn->makeView(i * per, i == nthreads - 1 ? CRF_UINT32_MAX : per));
    children[i]->trn_stream = children[i]->trn.get();
  }
}
size_t CRF_FeatureStreamManager::getNumFtrs() { return trn ? trn->num_ftrs() : 0; }
void CRF_FeatureStreamManager::rewindAllChildrenTrn() { for (auto& c : children) c->trn_stream->rewind(); }
void CRF_FeatureStreamManager::display() {
  std::cout << "feature stream: " << (trn ? trn->numUtterances() : 0) << " utterances, " << getNumFtrs() << " features per window, "
            << nthreads << " child stream(s)" << std::endl;
}

// ------------------------------------------------------------------------------------------
// helpers: one engine batch from a list of utterances
// ------------------------------------------------------------------------------------------
namespace {

struct HeldUtt {
  CRF_FeatureStream::Utterance u;
  std::vector<float> windows;     // when the stream only offers read()
  std::vector<uint32_t> labels;
};

// pull the stream's current utterance, through the fast path or the read() protocol
void grab(CRF_FeatureStream* strm, CRF_Model* crf, HeldUtt* h) {
  if (strm->currentUtterance(&h->u)) {
    if (!h->u.labels && h->u.phones && h->u.starts) {
      // nActualLabs*(dur-1)+phone of the segment ending at each frame (gradbuilder :216-231)
      const uint32_t L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs();
      h->labels.assign(h->u.T, CRF_LAB_BAD);
      for (uint32_t t = 0; t < h->u.T; t++)
        if (h->u.phones[t] != CRF_LAB_BAD) h->labels[t] = L * (t - h->u.starts[t]) + h->u.phones[t];
      h->u.labels = h->labels.data();
    }
    return;
  }
  const uint32_t D = crf->getLabMaxDur(), L = crf->getNActualLabs();
  const size_t F = strm->num_ftrs(), LW = strm->num_labs();
  std::vector<float> buf(F * D);
  std::vector<QNUInt32> lab(LW ? LW : 1);
  size_t bunch = 1;
  uint32_t t = 0;
  while (true) {
    size_t n = strm->read(bunch, buf.data(), LW ? lab.data() : nullptr);
    if (n == 0) break;
    h->windows.insert(h->windows.end(), buf.begin(), buf.begin() + n * F);
    uint32_t lb = CRF_LAB_BAD;
    if (LW && lab[0] != CRF_LAB_BAD) lb = L * (lab[2] - lab[1]) + lab[0];  // nActualLabs*(dur-1)+phone
    h->labels.push_back(lb);
    t++;
    if (bunch < D) bunch++;
  }
  if (t == 0) throw runtime_error("CRF_NewGradBuilder_StdSeg::buildGradient() caught exception: No features read from this sentence.");
  h->u.T = t;
  h->u.windows = h->windows.data();
  h->u.labels = h->labels.data();
  h->u.frames.clear();
}

// SCRF_HOST_TIMING=1: where the trainer's wall time goes (cumulative seconds per step of the loop, printed once per epoch)
struct HostClock {
  bool on = getenv("SCRF_HOST_TIMING") && atoi(getenv("SCRF_HOST_TIMING")) != 0;
  double t[6] = {0, 0, 0, 0, 0, 0};   // grab, batch, forward-backward, sums, update, epoch end (pull + files)
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void report(int iter) {
    if (!on) return;
    std::cout << "host timing, iteration " << iter << " (s): utterances " << t[0] << " batch " << t[1] << " forward-backward " << t[2]
              << " sums " << t[3] << " update " << t[4] << " epoch end " << t[5] << std::endl;
  }
};
HostClock g_clock;
const double g_process_t0 = HostClock::now();   // library load
struct ClockScope {
  int k; double t0;
  explicit ClockScope(int k_) : k(k_), t0(g_clock.on ? HostClock::now() : 0.0) {}
  ~ClockScope() { if (g_clock.on) g_clock.t[k] += HostClock::now() - t0; }
};

struct BatchGuard {
  crf_amd::Engine* e;
  scrf_batch b = nullptr;
  ~BatchGuard() { if (b) scrf_batch_destroy(e->h, b); }
};

void make_batch(crf_amd::Engine* e, CRF_FeatureStream* strm, const std::vector<HeldUtt>& utts, BatchGuard* g) {
  std::vector<scrf_utt> su(utts.size());
  const auto& rec = strm->recipes();
  for (size_t i = 0; i < utts.size(); i++) {
    memset(&su[i], 0, sizeof(scrf_utt));
    su[i].T = utts[i].u.T;
    su[i].windows = utts[i].u.windows;
    for (size_t s = 0; s < utts[i].u.frames.size() && s < SCRF_MAX_STREAMS; s++) su[i].frames[s] = utts[i].u.frames[s];
    su[i].labels = utts[i].u.labels;
  }
  e->check(scrf_batch_create(e->h, su.data(), (uint32_t)su.size(), (uint32_t)rec.size(), rec.empty() ? nullptr : rec.data(), &g->b),
           "scrf_batch_create");
}

}  // namespace

// ------------------------------------------------------------------------------------------
// CRF_GradBuilder
// ------------------------------------------------------------------------------------------
CRF_GradBuilder* CRF_GradBuilder::create(CRF_Model* crf, objfunctype ofunc) {
  if (ofunc != EXPF) throw runtime_error("CRF_GradBuilder::create: only the EXPF objective is built (the reference disables the others)");
  return new CRF_GradBuilder(crf);
}

double CRF_GradBuilder::buildGradient(CRF_FeatureStream* ftr_strm, double* grad, double* Zx_out) {
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  e->check(scrf_zero_grad(e->h), "buildGradient");
  double numer = 0, zx = 0;
  e->check(scrf_fb_batch(e->h, g.b, &numer, &zx), "CRF_NewGradBuilder_StdSeg_NoDur_NoTrans::buildGradient()");
  std::vector<double> gd(e->lambda_len);
  e->check(scrf_get_grad(e->h, gd.data(), e->lambda_len), "buildGradient");
  for (uint32_t i = 0; i < e->lambda_len; i++) grad[i] += gd[i];
  *Zx_out = zx;
  return numer;
}

// ------------------------------------------------------------------------------------------
// CRF_Minibatch_GradAccumulator
// ------------------------------------------------------------------------------------------
CRF_Minibatch_GradAccumulator::CRF_Minibatch_GradAccumulator(CRF_Model* c, std::vector<CRF_FeatureStream*> s)
    : crf(c), ftrStrms(s), segids(s.size(), QN_SEGID_BAD) {}

// stream i = the manager's child i (its trn_stream), or the manager's own stream for one thread
// (CRF_Minibatch_GradAccumulator.cpp:110-150)
CRF_Minibatch_GradAccumulator::CRF_Minibatch_GradAccumulator(CRF_Model* myCrf, CRF_FeatureStreamManager* mgr, QNUInt32 myNStreams)
    : crf(myCrf) {
  if (!mgr) throw runtime_error("CRF_Minibatch_GradAccumulator: no feature stream manager");
  if (myNStreams <= 1) {
    if (!mgr->trn_stream) throw runtime_error("CRF_Minibatch_GradAccumulator() Error: feature stream is NULL for thread 0");
    ftrStrms.push_back(mgr->trn_stream);
  } else {
    for (QNUInt32 i = 0; i < myNStreams; i++) {
      CRF_FeatureStreamManager* child = mgr->getChild(i);
      if (!child) throw runtime_error("CRF_Minibatch_GradAccumulator() Error: feature stream manager is NULL for thread " + std::to_string(i));
      if (!child->trn_stream) throw runtime_error("CRF_Minibatch_GradAccumulator() Error: feature stream is NULL for thread " + std::to_string(i));
      ftrStrms.push_back(child->trn_stream);
    }
  }
  segids.assign(ftrStrms.size(), QN_SEGID_BAD);
}

void CRF_Minibatch_GradAccumulator::setObjectiveFunction(objfunctype ofunc) {
  if (ofunc != EXPF) throw runtime_error("CRF_Minibatch_GradAccumulator: only the EXPF objective is built (the reference disables the others, CRF_GradBuilder.cpp:130-135)");
}

void CRF_Minibatch_GradAccumulator::setMinibatch(QNUInt32 mb) {
  if (mb != 0 && mb < ftrStrms.size())
    throw runtime_error("CRF_Minibatch_GradAccumulator::setMinibatch() Error: minibatch size is less than the number of threads.");
  minibatch = mb == 0 ? CRF_UINT32_MAX : mb;
}

void CRF_Minibatch_GradAccumulator::rewindAllAndNextSegs() {
  for (size_t s = 0; s < ftrStrms.size(); s++) {
    ftrStrms[s]->rewind();
    segids[s] = ftrStrms[s]->nextseg();
  }
}

double CRF_Minibatch_GradAccumulator::accumulateGradientOnDevice(double* Zx_out, QNUInt32* uttCount, bool* isEndOfIter) {
  return accumulate(nullptr, Zx_out, uttCount, isEndOfIter);
}
double CRF_Minibatch_GradAccumulator::accumulateGradient(double* grad, double* Zx_out, QNUInt32* uttCount,
                                                         bool* isEndOfIter) {
  return accumulate(grad, Zx_out, uttCount, isEndOfIter);
}

// stream s of N: its share of a minibatch (CRF_Minibatch_GradAccumulator.cpp:229-241,257) and its
// contiguous utterance view (io/CRF_FeatureStreamManager.cpp:425-464) -- exported for the CPU tests
extern "C" uint32_t crf_amd_minibatch_share(uint32_t minibatch, uint32_t n_streams, uint32_t s) {
  if (minibatch == CRF_UINT32_MAX) return CRF_UINT32_MAX;
  return minibatch / n_streams + (s < minibatch % n_streams ? 1 : 0);
}
extern "C" void crf_amd_view_range(uint32_t n_utts, uint32_t n_streams, uint32_t s, uint32_t* lo, uint32_t* hi) {
  const uint32_t per = n_utts / n_streams;
  *lo = s * per;
  *hi = s == n_streams - 1 ? n_utts : (s + 1) * per;
}

// The per-step collective itself failed on this rank (a device error around it, the watchdog's timeout, a peer's abort):
// the communicator is aborted -- which ends the peers' wait with an error instead of a hang -- and the exception takes
// the process down with a non-zero exit (mains catch, print and exit(-1)); a restart is a fresh process.
struct CollectiveFailure : public runtime_error {
  explicit CollectiveFailure(const string& m) : runtime_error(m) {}
};
static void collective(crf_amd::Engine* e, int rc, const char* what) {
  if (rc == SCRF_OK) return;
  const string msg = string(what) + ": " + scrf_last_error(e->h);
  scrf_comm_abort(e->h);
  throw CollectiveFailure(msg + " [communicator aborted; this rank exits]");
}

// One process: grad != nullptr -- per-stream gradients come to the host and are summed there in stream
// order, like the reference's join; grad == nullptr -- the streams accumulate into the device gradient
// one after the other (same order) and it is divided there.
// Distributed (CRF_Model::setDistributed): this process is stream `rank`; the sum over streams and the
// division by the active ones happen in scrf_allreduce_grad_ex on every rank's device gradient.
double CRF_Minibatch_GradAccumulator::accumulate(double* grad, double* Zx_out, QNUInt32* uttCount, bool* isEndOfIter) {
  crf_amd::Engine* e = crf->engine();
  const QNUInt32 n = e->lambda_len, N = (QNUInt32)ftrStrms.size();
  *uttCount = 0;
  *Zx_out = 0.0;
  const bool dist = crf->distributed();
  if (dist && (QNUInt32)crf->distWorld() != N)
    throw runtime_error("CRF_Minibatch_GradAccumulator: " + std::to_string(N) + " streams but " + std::to_string(crf->distWorld()) + " ranks (threads must equal WORLD_SIZE)");
  string local_failure;   // distributed: a rank that fails must still meet its peers in the collective (below)
  try {   // (inside the net as well: a device error here would otherwise skip the collective the peers are entering)
    if (grad && !dist) for (QNUInt32 i = 0; i < n; i++) grad[i] = 0.0;
    else e->check(scrf_zero_grad(e->h), "accumulateGradient");
  } catch (const std::exception& ex) {
    if (!dist) throw;
    local_failure = ex.what();
  }
  int nEnd = 0, nActive = 0;
  for (QNUInt32 s = 0; s < N; s++) if (segids[s] == QN_SEGID_BAD) ++nEnd;
  if (nEnd == (int)N) throw runtime_error("All feature streams are at the end! You don't have any utterances or you forget to rewind all the streams.");
  double totNumer = 0.0;
  std::vector<double> sgrad(grad && !dist ? n : 0);
  bool collective_done = false;              // distributed: the fused batch + all-reduce call has run
  double dist_sums4[4] = {0, 0, 0, 0}, dist_flags[2] = {0.0, 0.0};
  for (QNUInt32 s = 0; s < N; s++) try {  // stream order == the reference's join/sum order
    if (!local_failure.empty()) break;
    if (dist && (int)s != crf->distRank()) continue;
    if (segids[s] == QN_SEGID_BAD) continue;
    const QNUInt32 share = crf_amd_minibatch_share(minibatch, N, s);
    std::vector<HeldUtt> utts;
    {
      ClockScope cs(0);
      do {  // thread run loop: at least one utterance, then until the share is reached or the view ends
        utts.emplace_back();
        grab(ftrStrms[s], crf, &utts.back());
        segids[s] = ftrStrms[s]->nextseg();
      } while (utts.size() < share && segids[s] != QN_SEGID_BAD);
    }
    BatchGuard g{e};
    { ClockScope cs(1); make_batch(e, ftrStrms[s], utts, &g); }
    if (grad && !dist) e->check(scrf_zero_grad(e->h), "accumulateGradient");
    if (dist) {
      // this rank's one batch of the step and the collective in one call: the engine reduces the transition block
      // under the state contraction (scrf_fb_batch_allreduce).  The batch's own failure travels as the flag.
      const double fl[2] = {segids[s] == QN_SEGID_BAD ? 1.0 : 0.0, 0.0};
      int fb_rc = SCRF_OK;
      collective(e, scrf_fb_batch_allreduce(e->h, g.b, 1, fl, 2, 1, dist_sums4, dist_flags, &fb_rc), "accumulateGradient (all-reduce)");
      collective_done = true;
      *uttCount += (QNUInt32)utts.size();
      if (fb_rc != SCRF_OK) local_failure = string("CRF_Minibatch_GradAccumulator::accumulateGradient() caught exception: ") + scrf_last_error(e->h);
      continue;
    }
    { ClockScope cs(2); e->check(scrf_fb_batch(e->h, g.b, nullptr, nullptr), "CRF_Minibatch_GradAccumulator::accumulateGradient()"); }
    if (grad && !dist) {   // per-stream gradient and sums to the host (they restart with every scrf_zero_grad)
      double sums[3] = {0, 0, 0};
      e->check(scrf_get_batch_sums(e->h, sums), "accumulateGradient");
      e->check(scrf_get_grad(e->h, sgrad.data(), n), "accumulateGradient");
      totNumer += sums[0];
      *Zx_out += sums[1];
    }
    ++nActive;
    *uttCount += (QNUInt32)utts.size();
    if (segids[s] == QN_SEGID_BAD) ++nEnd;
    if (grad && !dist) for (QNUInt32 i = 0; i < n; i++) grad[i] += sgrad[i];
  } catch (const CollectiveFailure&) {
    throw;   // the collective itself failed (the fused call): not a failure of the share -- no second collective can follow
  } catch (const std::exception& ex) {
    if (!dist) throw;
    local_failure = ex.what();
  }
  if (dist) {
    // the other streams' state is only known through the collective: every rank sends "my stream is
    // exhausted after this step" and "I failed" along with {numerator, Zx, utterances, active}.  A rank that threw
    // (a bad label, a numeric failure in its share) would otherwise leave its peers waiting in RCCL for ever; this
    // way all of them see the flag in the same step and end together with a non-zero exit (the reference's
    // trainer dies with the exception of whichever thread threw).
    const int r = crf->distRank();
    const double flags_in[2] = {segids[r] == QN_SEGID_BAD ? 1.0 : 0.0, local_failure.empty() ? 0.0 : 1.0};
    double sums4[4] = {0, 0, 0, 0}, flags_out[2] = {0.0, 0.0};
    if (collective_done) {
      for (int i = 0; i < 4; i++) sums4[i] = dist_sums4[i];
      flags_out[0] = dist_flags[0]; flags_out[1] = dist_flags[1];
    } else   // no batch on this rank (exhausted view, or it failed before one existed): the same block sequence
      collective(e, scrf_allreduce_grad_ex(e->h, nActive, flags_in, 2, sums4, flags_out), "accumulateGradient (all-reduce)");
    if (flags_out[1] > 0.5) {
      if (!local_failure.empty()) throw runtime_error(local_failure);
      throw runtime_error("CRF_Minibatch_GradAccumulator: " + std::to_string((int)(flags_out[1] + 0.5)) + " other rank(s) failed in this minibatch; rank " + std::to_string(r) + " stops with them");
    }
    const double ended_out = flags_out[0];
    if (grad) e->check(scrf_get_grad(e->h, grad, n), "accumulateGradient");
    totNumer = sums4[0];
    *Zx_out = sums4[1];
    *uttCount = (QNUInt32)(sums4[2] + 0.5);
    *isEndOfIter = (int)(ended_out + 0.5) == (int)N;
    // mirror the other streams' end state, so that the all-at-end guard above and rewinds stay meaningful
    if (*isEndOfIter) for (QNUInt32 s = 0; s < N; s++) segids[s] = QN_SEGID_BAD;
    return totNumer;
  }
  // averaged over ACTIVE STREAMS (reference quirk, :306-308)
  if (grad) {
    for (QNUInt32 i = 0; i < n; i++) grad[i] /= nActive;
  } else if (deferSums) {
    ClockScope cs(3);
    if (sumsQueued) {   // the minibatch before: its copy finished long ago (this minibatch's recursion ran behind it)
      double sums[3] = {0, 0, 0};
      e->check(scrf_take_batch_sums(e->h, sums), "accumulateGradient");
      prevNumer = sums[0]; prevZx = sums[1]; prevReady = true;
    }
    e->check(scrf_queue_batch_sums(e->h), "accumulateGradient");
    sumsQueued = true;
    if (nActive > 1) e->check(scrf_div_grad(e->h, (double)nActive), "accumulateGradient");
  } else {
    ClockScope cs(3);
    double sums[3] = {0, 0, 0};
    e->check(scrf_get_batch_sums(e->h, sums), "accumulateGradient");
    totNumer = sums[0];
    *Zx_out = sums[1];
    if (nActive > 1) e->check(scrf_div_grad(e->h, (double)nActive), "accumulateGradient");
  }
  *isEndOfIter = nEnd == (int)N;
  return totNumer;
}

bool CRF_Minibatch_GradAccumulator::takePreviousSums(double* numer, double* Zx) {
  if (!prevReady) return false;
  *numer = prevNumer; *Zx = prevZx;
  prevReady = false;
  return true;
}
void CRF_Minibatch_GradAccumulator::takeCurrentSums(double* numer, double* Zx) {
  if (!sumsQueued) throw runtime_error("CRF_Minibatch_GradAccumulator::takeCurrentSums: no minibatch is outstanding");
  double sums[3] = {0, 0, 0};
  crf->engine()->check(scrf_take_batch_sums(crf->engine()->h, sums), "accumulateGradient");
  *numer = sums[0]; *Zx = sums[1];
  sumsQueued = false;
}

// ------------------------------------------------------------------------------------------
// CRF_Trainer / CRF_SGTrainer
// ------------------------------------------------------------------------------------------
static string dir_of(const string& p) {
  size_t k = p.find_last_of('/');
  return k == string::npos ? string(".") : p.substr(0, k);
}
static bool touch(const string& f) {
  std::ofstream o(f.c_str());
  if (!o.is_open()) { std::cerr << "ERROR: cannot touch the done file " << f << std::endl; return false; }
  return true;
}

CRF_Trainer::CRF_Trainer(CRF_Model* crf_in, CRF_FeatureStreamManager* ftr_str_mgr, char* wt_fname)
    : crf_ptr(crf_in), ftr_strm_mgr(ftr_str_mgr), weight_fname(wt_fname ? wt_fname : ""), weight_dir(dir_of(weight_fname)) {}
void CRF_Trainer::train() { throw runtime_error("CRF_Trainer::train: abstract trainer"); }
void CRF_Trainer::setObjectiveFunction(objfunctype ofunc) {
  if (ofunc != EXPF) throw runtime_error("CRF_Trainer::setObjectiveFunction: only the EXPF objective is built (the reference disables the others, CRF_GradBuilder.cpp:130-135)");
  objective = ofunc;
}
bool CRF_Trainer::touchDoneFileIter(int iter) { return touch(weight_dir + "/.done.train.i" + std::to_string(iter)); }
bool CRF_Trainer::touchDoneFileFinal() { return touch(weight_dir + "/.done.train"); }

CRF_SGTrainer::CRF_SGTrainer(CRF_Model* crf_in, CRF_FeatureStreamManager* mgr, char* wt_fname) : CRF_Trainer(crf_in, mgr, wt_fname) {}
CRF_SGTrainer::CRF_SGTrainer(CRF_Model* crf, std::vector<CRF_FeatureStream*> s, const char* wf)
    : CRF_Trainer(crf, nullptr, const_cast<char*>(wf)), own_streams(s) { nThreads = (int)s.size(); }

void CRF_SGTrainer::train() { sgtrainMinibatch(); }   // both branches of the reference's train() (:58-65)

namespace {
// The three weight-sized text files of an iteration (and its done marker, last) are written by a thread of their own
// while the next iteration's minibatches run: at the TIMIT demo's size an iteration is 0.45 s of minibatches and 0.4 s
// of files.  The thread works on copies; the next iteration end, the final writes and every exit path wait for it, and
// an error in it surfaces there.  SCRF_ASYNC_WRITE=0: written in line, as before.
struct EpochWriter {
  std::thread th;
  std::string err;
  bool async = !(getenv("SCRF_ASYNC_WRITE") && atoi(getenv("SCRF_ASYNC_WRITE")) == 0);
  void wait() {
    if (th.joinable()) th.join();
    if (!err.empty()) { const std::string m = err; err.clear(); throw runtime_error(m); }
  }
  template <class F> void run(F&& job) {
    wait();
    if (!async) { job(); return; }
    th = std::thread([this, job]() {
      try { job(); } catch (const std::exception& ex) { err = ex.what(); }
    });
  }
  ~EpochWriter() { if (th.joinable()) th.join(); }
};
}  // namespace

void CRF_SGTrainer::sgtrainMinibatch() {
  crf_amd::Engine* e = crf_ptr->engine();
  EpochWriter writer;
  crf_ptr->pushLambda();
  const QNUInt32 n = crf_ptr->getLambdaLen();
  const bool chief = !crf_ptr->distributed() || crf_ptr->distRank() == 0;   // one writer of files and progress lines
  std::unique_ptr<CRF_Minibatch_GradAccumulator> gaccum(
      ftr_strm_mgr ? new CRF_Minibatch_GradAccumulator(crf_ptr, ftr_strm_mgr, (QNUInt32)nThreads)
                   : new CRF_Minibatch_GradAccumulator(crf_ptr, own_streams));
  gaccum->setMinibatch((QNUInt32)minibatch);
  gaccum->setUttReport((int)uttRpt);
  // one process: the sums of a minibatch are read one minibatch later (SCRF_ASYNC_SUMS=0: at once, as before round 4)
  const bool defer = !crf_ptr->distributed() && !(getenv("SCRF_ASYNC_SUMS") && atoi(getenv("SCRF_ASYNC_SUMS")) == 0);
  gaccum->setDeferSums(defer);
  QNUInt32 waitInc = 0, waitAfter = 0;
  std::vector<double> lambdaAvg(n, 0.0);
  int iCounter = (int)crf_ptr->getInitIter();
  QNUInt32 uCounter = 0;
  int accCnt = (int)crf_ptr->getPresentations();
  double totLogLi = 0.0;
  // a resumed run replays the presentation-order generator (CRF_SGTrainer.cpp:88-93)
  for (int i = 0; i < iCounter; i++) {
    if (ftr_strm_mgr) {
      ftr_strm_mgr->trn_stream->rewind();
      if (nThreads > 1) ftr_strm_mgr->rewindAllChildrenTrn();
    } else {
      for (CRF_FeatureStream* st_ : own_streams) st_->rewind();
    }
  }
  // the reference's prior step scales the gradient by (1 - 1/gvar) (:300-303); kept as written
  const float invSquareVar = useGvar ? 1 / gvar : 0.0f;
  gaccum->rewindAllAndNextSegs();
  if (chief && g_clock.on) std::cout << "host timing: " << HostClock::now() - g_process_t0 << " s from program start to the first minibatch (inputs, model, device)" << std::endl;
  bool start = true;
  while (iCounter < maxIters) {
    if (start && chief) {
      if (useAdagrad) std::cout << "Iteration: " << iCounter << " starting AdaGrad scaling factor (eta): " << eta << std::endl;
      else std::cout << "Iteration: " << iCounter << " starting LR: " << lr << std::endl;
    }
    start = false;
    QNUInt32 inc = 0;
    bool endOfIter = false;
    double Zx = 0.0;
    const double numer = gaccum->accumulateGradientOnDevice(&Zx, &inc, &endOfIter);
    uCounter += inc;
    // the minibatch's share of the log-likelihood and its progress line -- at once, or (deferred sums) one minibatch later,
    // in the same order and with the same text
    auto account = [&](double nm, double zx, QNUInt32 n_utts, QNUInt32 u_after) {
      const double logLi = nm - zx;
      totLogLi += logLi;
      if (chief && uttRpt > 0 && u_after % uttRpt == 0)
        std::cout << " Finished Utt: " << u_after - 1 << " Batch-Avg Numerator: " << nm / n_utts << " Batch-Avg Zx: " << zx / n_utts
                  << " Batch-Avg LogLi: " << logLi / n_utts << " Iter-Avg LogLi: " << totLogLi / u_after << std::endl;
    };
    if (defer) {
      double pn = 0.0, pz = 0.0;
      if (gaccum->takePreviousSums(&pn, &pz)) account(pn, pz, waitInc, waitAfter);
      waitInc = inc; waitAfter = uCounter;
      if (endOfIter) {   // the iteration's last minibatch: its numbers are needed now
        gaccum->takeCurrentSums(&pn, &pz);
        account(pn, pz, waitInc, waitAfter);
      }
    } else account(numer, Zx, inc, uCounter);
    if (useGvar) e->check(scrf_gauss_prior(e->h, invSquareVar), "sgtrainMinibatch");
    // update on the device: lambda += lr*g | AdaGrad; lambdaAcc += lambda; g = 0
    { ClockScope cs(4); e->check(scrf_sgd_step(e->h, useAdagrad ? eta : (double)lr, useAdagrad, eps), "sgtrainMinibatch"); }
    accCnt += (int)inc;
    if (endOfIter) {
      ClockScope cs_end(5);
      crf_ptr->pullLambda();
      const double* acc = crf_ptr->getLambdaAcc();
      for (QNUInt32 i = 0; i < n; i++) lambdaAvg[i] = acc[i] / (float)accCnt;
      if (chief) {
        const string base = weight_fname + ".i" + std::to_string(iCounter);
        std::cout << "Writing Iteration " << iCounter << " weights to file " << base << ".out" << std::endl;
        auto lam = std::make_shared<std::vector<double>>(crf_ptr->getLambda(), crf_ptr->getLambda() + n);
        auto avg = std::make_shared<std::vector<double>>(lambdaAvg);
        std::shared_ptr<std::vector<double>> gsa;
        if (useAdagrad) gsa = std::make_shared<std::vector<double>>(crf_ptr->getGradSqrAcc(), crf_ptr->getGradSqrAcc() + n);
        const string done = weight_dir + "/.done.train.i" + std::to_string(iCounter);
        writer.run([base, lam, avg, gsa, done]() {
          write_vec((base + ".out").c_str(), lam->data(), lam->size());
          write_vec((base + ".avg.out").c_str(), avg->data(), avg->size());
          if (gsa) write_vec((base + ".gradSqrAcc.out").c_str(), gsa->data(), gsa->size());
          touch(done);   // the marker last: a resumed run trusts the files it names
        });
      }
      gaccum->rewindAllAndNextSegs();
      if (chief) g_clock.report(iCounter);
      iCounter++;
      uCounter = 0;
      totLogLi = 0.0;
      start = true;
      if (!useAdagrad) {
        lr *= lr_decay_rate;
        if (chief) std::cout << "Learning rate is decayed by " << lr_decay_rate << " to " << lr << std::endl;
      }
    }
  }
  writer.wait();
  crf_ptr->pullLambda();
  if (chief) {
    std::cout << "Writing Final Iteration weights to file " << weight_fname << std::endl;
    crf_ptr->writeToFile(weight_fname.c_str());
    crf_ptr->writeToFile((weight_fname + ".avg.out").c_str(), lambdaAvg.data(), n);
    touchDoneFileFinal();
    if (crf_ptr->distributed()) {
      uint64_t nc = 0, no = 0;
      if (scrf_comm_stats(crf_ptr->engine()->h, &nc, &no) == SCRF_OK)
        std::cout << "Gradient all-reduces: " << nc << " (transition block overlapped with the state contraction in " << no << ")" << std::endl;
    }
  }
}

// ------------------------------------------------------------------------------------------
// CRF_GradAccumulator (full batch) / CRF_LBFGSTrainer
// ------------------------------------------------------------------------------------------
void CRF_GradAccumulator::setObjectiveFunction(objfunctype ofunc) {
  if (ofunc != EXPF) throw runtime_error("CRF_GradAccumulator: only the EXPF objective is built");
}

double CRF_GradAccumulator::accumulateGradient(CRF_FeatureStreamManager* mgr, int nStreams, double* grad, QNUInt32* uttCount) {
  crf_amd::Engine* e = crf->engine();
  const QNUInt32 n = e->lambda_len;
  const bool dist = crf->distributed();
  if (dist && crf->distWorld() != nStreams)
    throw runtime_error("CRF_GradAccumulator: " + std::to_string(nStreams) + " streams but " + std::to_string(crf->distWorld()) + " ranks");
  string local_failure;
  try {
    crf->pushLambda();
    e->check(scrf_zero_grad(e->h), "accumulateGradient");
  } catch (const std::exception& ex) {
    if (!dist) throw;
    local_failure = ex.what();
  }
  for (int s = 0; s < nStreams; s++) try {
    if (!local_failure.empty()) break;
    if (dist && s != crf->distRank()) continue;
    CRF_FeatureStream* strm = nStreams == 1 ? mgr->trn_stream : mgr->getChild((size_t)s)->trn_stream;
    strm->rewind();
    QN_SegID segid = strm->nextseg();
    if (segid == QN_SEGID_BAD) throw runtime_error("Feature stream contains no utterances!");
    while (segid != QN_SEGID_BAD) {   // the stream's whole view, in device batches
      std::vector<HeldUtt> utts;
      do {
        utts.emplace_back();
        grab(strm, crf, &utts.back());
        segid = strm->nextseg();
      } while (utts.size() < deviceBatch && segid != QN_SEGID_BAD);
      BatchGuard g{e};
      make_batch(e, strm, utts, &g);
      e->check(scrf_fb_batch(e->h, g.b, nullptr, nullptr), "CRF_GradAccumulator::accumulateGradient()");
    }
  } catch (const std::exception& ex) {
    if (!dist) throw;
    local_failure = ex.what();
  }
  double sums[4] = {0, 0, 0, 0};
  if (dist) {
    // sum over the ranks; the collective divides by the active ranks (all of them here): undone.  A failed rank
    // still joins, with a flag, so that every rank stops in the same evaluation (see the minibatch accumulator).
    const double failed_in = local_failure.empty() ? 0.0 : 1.0;
    double failed_out = 0.0;
    collective(e, scrf_allreduce_grad_ex(e->h, 1, &failed_in, 1, sums, &failed_out), "accumulateGradient (all-reduce)");
    if (failed_out > 0.5) throw runtime_error(!local_failure.empty() ? local_failure : "CRF_GradAccumulator: another rank failed in this evaluation; rank " + std::to_string(crf->distRank()) + " stops with it");
    e->check(scrf_scale_grad(e->h, sums[3]), "accumulateGradient");
  } else {
    e->check(scrf_get_batch_sums(e->h, sums), "accumulateGradient");
  }
  e->check(scrf_get_grad(e->h, grad, n), "accumulateGradient");
  *uttCount = (QNUInt32)(sums[2] + 0.5);
  return sums[0] - sums[1];
}

CRF_LBFGSTrainer::CRF_LBFGSTrainer(CRF_Model* crf_in, CRF_FeatureStreamManager* mgr, char* wt_fname) : CRF_Trainer(crf_in, mgr, wt_fname) {
  iCounter = (int)crf_ptr->getInitIter();
  for (int i = 0; i < iCounter; ++i) ftr_strm_mgr->trn_stream->rewind();
}

void CRF_LBFGSTrainer::train() {
  const QNUInt32 n = crf_ptr->getLambdaLen();
  const bool chief = !crf_ptr->distributed() || crf_ptr->distRank() == 0;
  CRF_GradAccumulator gaccum(crf_ptr, useLogspace != 0, (int)crf_ptr->getFeatureMap()->getNumStates());
  gaccum.setUttReport((int)uttRpt);
  gaccum.setObjectiveFunction(objective);
  std::vector<double> grad(n, 0.0), x(crf_ptr->getLambda(), crf_ptr->getLambda() + n);
  const double invSquareVar = useGvar ? 1 / gvar : 0.0;   // float division like the reference's (:48-50)
  const int nStreams = (int)ftr_strm_mgr->getNThreads();
  bool start = true;
  auto evaluate = [&](const double* lam, double* g, int len, double) -> double {   // evaluateGradient :70-165
    crf_ptr->setLambda(const_cast<double*>(lam), (QNUInt32)len);
    if (start) {
      start = false;
    } else if (chief) {
      std::stringstream ss;
      ss << weight_fname << ".i" << iCounter << ".out";
      std::cout << "Writing Iteration " << iCounter << " weights to file " << ss.str() << std::endl;
      crf_ptr->writeToFile(ss.str().c_str());
    }
    iCounter++;
    if (chief) std::cout << "Iteration: " << iCounter << std::endl;
    QNUInt32 uCounter = 0;
    double totLogLi = gaccum.accumulateGradient(ftr_strm_mgr, nStreams, grad.data(), &uCounter);
    if (useGvar) {
      for (int i = 0; i < len; i++) {
        g[i] = -grad[i] + lam[i] * invSquareVar;
        totLogLi -= ((lam[i] * lam[i]) * invSquareVar) / 2;
      }
    } else {
      for (int i = 0; i < len; i++) g[i] = -grad[i];
    }
    if (chief)
      std::cout << " End iteration: " << iCounter << " totLogLi: " << totLogLi << " Avg LogLi: " << totLogLi / uCounter
                << " ucounter: " << uCounter << std::endl;
    return -totLogLi;
  };
  auto progress = [&](const double*, const double*, double, double, double, double, int, int, int) -> int {   // :190-205
    if (chief) {
      std::cout << "PROGRESS called" << std::endl;
      std::cout << "Iteration: " << iCounter << " ending" << std::endl;
    }
    return iCounter >= maxIters ? 1 : 0;
  };
  double fx = 0.0;
  status = crf_amd::lbfgs_minimize((int)n, x.data(), &fx, evaluate, progress);
  // the reference writes the final weights only when lbfgs() returns 0 and complains otherwise (:63-68); a stop
  // requested by progress() (crf_epochs reached) is its non-zero LBFGSERR_CANCELED.  Here the last accepted point is
  // written in both cases -- losing crf_epochs evaluations of work to a return code helps nobody -- and the code is shown.
  crf_ptr->setLambda(x.data(), n);
  if (status != crf_amd::LBFGS_OK && chief) std::cerr << "LBFGS returned: " << status << (status == crf_amd::LBFGS_STOP ? " (stopped at crf_epochs evaluations)" : "") << std::endl;
  if (chief && (status == crf_amd::LBFGS_OK || status == crf_amd::LBFGS_STOP || status == crf_amd::LBFGS_ALREADY_MINIMIZED)) {
    std::cout << "Writing Final Iteration weights to file " << weight_fname << std::endl;
    crf_ptr->writeToFile(weight_fname.c_str());
    touchDoneFileFinal();
  }
}

// ------------------------------------------------------------------------------------------
// CRF_StateVector / CRF_StateNode: node values of one utterance from the engine's hooks
// ------------------------------------------------------------------------------------------
// the node view and the Viterbi decoder read getStateValue(lab, dur) / getTransValue(p, c) of the one-matrix-per-frame
// models; the other node types keep their values per window or in sparse tables (scrf_scores documents the shapes)
static void require_dense_model(CRF_Model* crf, const char* who) {
  const modeltype mt = crf->getModelType();
  const bool dense = mt == STDFRAME || mt == STDSEG_NO_DUR_NO_TRANSFTR || mt == STDSEG_NO_DUR_NO_SEGTRANSFTR;
  if (!dense || (crf->getFeatureMap() && crf->getFeatureMap()->getNumStates() != 1))
    throw runtime_error(string(who) + ": built for stdframe (one state per label), stdseg_no_dur_no_transftr and stdseg_no_dur_no_segtransftr");
}

CRF_StateVector::CRF_StateVector(CRF_FeatureStream* ftr_strm, CRF_Model* crf) {
  require_dense_model(crf, "CRF_StateVector");
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  const uint32_t T = utts[0].u.T, L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs(), D = crf->getLabMaxDur();
  uint64_t n_segs = 0;
  e->check(scrf_batch_info(e->h, g.b, nullptr, nullptr, &n_segs, nullptr), "CRF_StateVector");
  S.resize((size_t)n_segs * L); M.resize((size_t)T * L * L);
  AD.resize((size_t)n_segs * L); AL.resize((size_t)T * L); BE.resize((size_t)T * L);
  e->check(scrf_scores(e->h, g.b, 0, S.data(), M.data()), "CRF_StateNode::computeTransMatrix");
  e->check(scrf_forward_backward(e->h, g.b, 0, SCRF_PREC_EXACT, AD.data(), AL.data(), BE.data(), &zx), "CRF_StateNode::computeAlpha");
  nodes.resize(T);
  for (uint32_t t = 0; t < T; t++) {
    CRF_StateNode& nd = nodes[t];
    const size_t base = t < D ? (size_t)t * (t + 1) / 2 : (size_t)D * (D + 1) / 2 + (size_t)(t - D) * D;
    nd.alpha = &AL[(size_t)t * L];
    nd.beta = &BE[(size_t)t * L];
    nd.alpha_dur = &AD[base * L];
    nd.S = &S[base * L];
    nd.M = &M[(size_t)t * L * L];
    nd.nLabs = L;
    nd.nodeMaxDur = t + 1 <= D ? t + 1 : D;
    nd.label = utts[0].u.labels ? utts[0].u.labels[t] : CRF_LAB_BAD;
    nd.zx = zx;
    nd.last = t + 1 == T;
  }
}
double CRF_StateNode::computeAlphaSum() {
  if (!last) throw runtime_error("CRF_StateNode::computeAlphaSum: the node view answers it for the utterance's last node (Zx)");
  return zx;
}
double CRF_StateNode::getStateValue(QNUInt32 lab, QNUInt32 dur) {
  if (lab >= nLabs || dur < 1 || dur > nodeMaxDur) throw runtime_error("CRF_StateNode::getStateValue: label or duration out of range");
  return S[(size_t)(dur - 1) * nLabs + lab];
}
double CRF_StateNode::getTransValue(QNUInt32 prev_lab, QNUInt32 cur_lab) {
  if (prev_lab >= nLabs || cur_lab >= nLabs) throw runtime_error("CRF_StateNode::getTransValue: label out of range");
  return M[(size_t)prev_lab * nLabs + cur_lab];
}
double CRF_StateNode::getFullTransValue(QNUInt32 prev_lab, QNUInt32 cur_lab, QNUInt32 dur) {
  return getTransValue(prev_lab, cur_lab) + getStateValue(cur_lab, dur);
}

// ------------------------------------------------------------------------------------------
// lattice / best path
// ------------------------------------------------------------------------------------------
int CRF_LatticeBuilder::latticeArcs(bool norm, std::vector<scrf_arc>* arcs, uint32_t* n_states, int32_t* fin, std::vector<QNUInt32>* node_labels) {
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  if (node_labels) {   // the nodes' labels (frame labels, or phone-duration labels at segment ends, CRF_LAB_BAD elsewhere)
    node_labels->clear();
    if (utts[0].u.labels) node_labels->assign(utts[0].u.labels, utts[0].u.labels + utts[0].u.T);
  }
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  uint64_t na = 0;
  e->check(scrf_lattice_arcs(e->h, g.b, 0, norm, nullptr, &na, n_states, fin), "buildLattice");
  arcs->resize(na);
  e->check(scrf_lattice_arcs(e->h, g.b, 0, norm, arcs->data(), &na, n_states, fin), "buildLattice");
  return (int)utts[0].u.T;
}

// nStateDecode for the free phone loop: device Viterbi, then the arc weights of the backtrace
// from the node scores -- the reference reads them off the segment's END node
// (getFullTransValue(prev, cur, dur) = getTransValue + getStateValue of node seg_end, :2262)
int CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::decode() {
  require_dense_model(crf, "CRF_ViterbiDecoder_StdSeg_NoSegTransFtr");
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  const uint32_t T = utts[0].u.T, L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs(), D = crf->getLabMaxDur();
  segs.clear();
  zx = 0.0;
  best_weight = 0.0f;
  if (T == 0) return 0;
  std::vector<uint32_t> labs(T);
  uint64_t off[2] = {0, 0};
  e->check(scrf_viterbi_batch(e->h, g.b, labs.data(), labs.size(), off, &best_weight), "nStateDecode");
  labs.resize(off[1]);
  uint64_t n_frames = 0, n_segs = 0;
  uint32_t nu = 0;
  uint64_t n_arcs = 0;
  e->check(scrf_batch_info(e->h, g.b, &nu, &n_frames, &n_segs, &n_arcs), "nStateDecode");
  std::vector<double> S((size_t)n_segs * L), M((size_t)T * L * L);
  e->check(scrf_scores(e->h, g.b, 0, S.data(), M.data()), "nStateDecode");
  e->check(scrf_forward_backward(e->h, g.b, 0, SCRF_PREC_EXACT, nullptr, nullptr, nullptr, &zx), "nStateDecode");
  auto seg_row = [&](uint32_t t, uint32_t d) -> size_t {  // row of the window of length d ending at t
    const size_t base = t < D ? (size_t)t * (t + 1) / 2 : (size_t)D * (D + 1) / 2 + (size_t)(t - D) * D;
    return base + d - 1;
  };
  uint32_t at = 0, prev = 0;
  for (size_t i = 0; i < labs.size(); i++) {
    Segment sg;
    sg.phone = labs[i] % L;
    sg.dur = labs[i] / L + 1;
    sg.start = at;
    const uint32_t te = at + sg.dur - 1;
    if (te >= T) throw runtime_error("nStateDecode: best path runs past the utterance");
    const double sv = S[seg_row(te, sg.dur) * L + sg.phone];
    if (i == 0) {
      sg.weight = (float)(-1 * sv);
      sg.phone_start = true;
    } else {
      sg.weight = (float)(-1 * (M[((size_t)te * L + prev) * L + sg.phone] + sv));
      sg.phone_start = prev != sg.phone;
    }
    segs.push_back(sg);
    prev = sg.phone;
    at += sg.dur;
  }
  if (at != T) throw runtime_error("nStateDecode: best path does not cover the utterance");
  return (int)T;
}

void crf_amd::readFstText(const char* fname, crf_amd::ArcListFst* fst) {
  std::ifstream f(fname);
  if (!f.is_open()) throw runtime_error(string("readFstText: cannot open ") + fname);
  string line;
  int max_state = -1;
  bool first = true;
  while (getline(f, line)) {
    std::istringstream is(line);
    std::vector<string> tok;
    string t;
    while (is >> t) tok.push_back(t);
    if (tok.empty()) continue;
    if (tok.size() >= 4) {
      const int src = atoi(tok[0].c_str()), dst = atoi(tok[1].c_str());
      const float w = tok.size() >= 5 ? (float)atof(tok[4].c_str()) : 0.0f;
      if (first) { fst->start = src; first = false; }
      fst->arcs.push_back(scrf_arc{src, atoi(tok[2].c_str()), atoi(tok[3].c_str()), w, dst});
      max_state = std::max(max_state, std::max(src, dst));
    } else if (tok.size() <= 2) {
      const int st = atoi(tok[0].c_str());
      if (first) { fst->start = st; first = false; }
      fst->SetFinal(st, tok.size() == 2 ? (float)atof(tok[1].c_str()) : 0.0f);
      max_state = std::max(max_state, st);
    } else {
      throw runtime_error(string("readFstText: ") + fname + ": cannot parse line '" + line + "'");
    }
  }
  if (first) throw runtime_error(string("readFstText: ") + fname + " is empty");
  fst->n_states = max_state + 1;
}

namespace {
struct BinReader {
  std::vector<unsigned char> d;
  size_t at = 0;
  string name;
  void need(size_t n) { if (at + n > d.size()) throw runtime_error(name + ": truncated OpenFST file (print it with fstprint and use crf_lm_txt)"); }
  template <class Tp> Tp get() { need(sizeof(Tp)); Tp v; memcpy(&v, &d[at], sizeof(Tp)); at += sizeof(Tp); return v; }
  string str() {
    const int32_t n = get<int32_t>();
    if (n < 0 || n > 4096) throw runtime_error(name + ": implausible string length in the OpenFST header");
    need((size_t)n);
    string s((const char*)&d[at], (size_t)n);
    at += (size_t)n;
    return s;
  }
};
}  // namespace

void crf_amd::readFstBinary(const char* fname, crf_amd::ArcListFst* fst) {
  BinReader r;
  r.name = fname;
  {
    std::ifstream f(fname, std::ios::binary);
    if (!f.is_open()) throw runtime_error(string("readFstBinary: cannot open ") + fname);
    r.d.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  }
  const string hint = " (print it with fstprint and use crf_lm_txt)";
  if (r.get<int32_t>() != 2125659606) throw runtime_error(r.name + ": not an OpenFST binary file" + hint);
  const string fst_type = r.str(), arc_type = r.str();
  const int32_t version = r.get<int32_t>(), flags = r.get<int32_t>();
  (void)r.get<uint64_t>();  // properties
  const int64_t start = r.get<int64_t>(), n_states = r.get<int64_t>(), n_arcs = r.get<int64_t>();
  if (fst_type != "vector") throw runtime_error(r.name + ": FST type '" + fst_type + "' is not 'vector'" + hint);
  if (arc_type != "standard" && arc_type != "log") throw runtime_error(r.name + ": arc type '" + arc_type + "' is not standard/log" + hint);
  if (version < 1 || version > 2) throw runtime_error(r.name + ": unsupported vector FST version " + std::to_string(version) + hint);
  if (flags & 4) throw runtime_error(r.name + ": aligned FST files are not supported" + hint);
  if (n_states < 0 || n_arcs < 0 || n_states > (1 << 26) || start >= n_states) throw runtime_error(r.name + ": implausible state/arc counts" + hint);
  for (int side = 0; side < 2; side++) {   // embedded symbol tables (flags 1: input, 2: output) are skipped
    if (!(flags & (1 << side))) continue;
    if (r.get<int32_t>() != 2125658996) throw runtime_error(r.name + ": bad symbol table magic" + hint);
    (void)r.str();
    (void)r.get<int64_t>();
    const int64_t n = r.get<int64_t>();
    if (n < 0 || n > (1 << 26)) throw runtime_error(r.name + ": implausible symbol table size" + hint);
    for (int64_t i = 0; i < n; i++) { (void)r.str(); (void)r.get<int64_t>(); }
  }
  fst->n_states = (int)n_states;
  fst->start = (int)start;
  int64_t seen = 0;
  for (int64_t s = 0; s < n_states; s++) {
    const float fw = r.get<float>();
    const int64_t na = r.get<int64_t>();
    if (na < 0 || seen + na > n_arcs) throw runtime_error(r.name + ": arc counts do not add up" + hint);
    if (fw == fw && fw < 3.0e38f) fst->SetFinal((int)s, fw);   // +inf (Zero) marks a non-final state
    for (int64_t k = 0; k < na; k++) {
      scrf_arc a;
      a.src = (int32_t)s;
      a.ilabel = r.get<int32_t>();
      a.olabel = r.get<int32_t>();
      a.w = r.get<float>();
      a.dst = r.get<int32_t>();
      if (a.dst < 0 || a.dst >= n_states) throw runtime_error(r.name + ": arc to a state out of range" + hint);
      fst->arcs.push_back(a);
    }
    seen += na;
  }
  if (seen != n_arcs || r.at != r.d.size()) throw runtime_error(r.name + ": the file does not end where its header says" + hint);
}

// With `group` and a positive `beam` the search is pruned time-synchronously the way the reference's decoder prunes
// (decoders/CRF_ViterbiDecoder_StdSeg_NoSegTransFtr.cpp pruning() :976-1060 and its reads at :573, :760: a hypothesis of
// a node is kept, and expanded into the next nodes, iff its weight is < the node's minimum + beam): group[s] >= 0 names
// the node whose hypotheses lattice state s carries (the states of one group have consecutive ids and no arcs among
// themselves, so all of them are final when the first is reached); states with group -1 are never pruned.
bool crf_amd::composeShortestPath(const crf_amd::ArcListFst& lat, const crf_amd::ArcListFst& lm, crf_amd::ArcListFst* best, float* total,
                                  const std::vector<int>* group, double beam, uint64_t* n_expanded) {
  const int S = lat.n_states, Q = lm.n_states;
  if (group && (int)group->size() != S) throw runtime_error("composeShortestPath: one group entry per lattice state");
  if (S <= 0 || Q <= 0 || lat.start < 0 || lm.start < 0) throw runtime_error("composeShortestPath: a machine has no start state");
  if ((size_t)S * (size_t)Q > ((size_t)1 << 28)) throw runtime_error("composeShortestPath: lattice x LM too large for the dense product search");
  const float INF = std::numeric_limits<float>::infinity();
  // arcs by source state, in insertion order
  std::vector<std::vector<int> > lout(S), mout(Q);
  for (size_t i = 0; i < lat.arcs.size(); i++) {
    const scrf_arc& a = lat.arcs[i];
    if (a.src < 0 || a.src >= S || a.dst < 0 || a.dst >= S) throw runtime_error("composeShortestPath: lattice arc with a state out of range");
    if (a.dst <= a.src) throw runtime_error("composeShortestPath: the lattice's state ids are not a topological order");
    lout[a.src].push_back((int)i);
  }
  for (size_t i = 0; i < lm.arcs.size(); i++) {
    const scrf_arc& a = lm.arcs[i];
    if (a.src < 0 || a.src >= Q || a.dst < 0 || a.dst >= Q) throw runtime_error("composeShortestPath: LM arc with a state out of range");
    mout[a.src].push_back((int)i);
  }
  std::vector<float> lfin(S, INF), mfin(Q, INF);
  for (const auto& f : lat.finals) if (f.first >= 0 && f.first < S) lfin[f.first] = std::min(lfin[f.first], f.second);
  for (const auto& f : lm.finals) if (f.first >= 0 && f.first < Q) mfin[f.first] = std::min(mfin[f.first], f.second);
  const size_t N = (size_t)S * Q;
  std::vector<float> dist(N, INF);
  struct Back { int32_t prev; int32_t la, ma; };   // previous product state, lattice arc (-1: none), LM arc (-1: none)
  std::vector<Back> back(N, Back{-1, -1, -1});
  dist[(size_t)lat.start * Q + lm.start] = 0.0f;
  std::vector<int> work;
  float best_w = INF;
  long best_at = -1;
  int cur_group = -1;
  uint64_t n_exp = 0;
  for (int s = 0; s < S; s++) {
    float* ds = &dist[(size_t)s * Q];
    if (group && beam > 0 && (*group)[s] >= 0 && (*group)[s] != cur_group) {
      // first state of a node: its hypotheses are final; keep those below the node's minimum + beam
      cur_group = (*group)[s];
      int s1 = s;
      float gmin = INF;
      for (; s1 < S && (*group)[s1] == cur_group; s1++)
        for (int q = 0; q < Q; q++) gmin = std::min(gmin, dist[(size_t)s1 * Q + q]);
      for (int x = s; x < s1; x++)
        for (int q = 0; q < Q; q++) {
          float& dq = dist[(size_t)x * Q + q];
          if (dq < INF && !(dq < gmin + (float)beam)) dq = INF;
        }
    }
    for (int q = 0; q < Q; q++) if (ds[q] < INF) n_exp++;
    // epsilon-input closure of the LM at this lattice state
    work.clear();
    for (int q = 0; q < Q; q++) if (ds[q] < INF) work.push_back(q);
    size_t guard = 0;
    for (size_t k = 0; k < work.size(); k++) {
      const int q = work[k];
      for (int ai : mout[q]) {
        const scrf_arc& m = lm.arcs[ai];
        if (m.ilabel != 0) continue;
        const float w = ds[q] + (0.0f + m.w);
        if (w < ds[m.dst]) {
          ds[m.dst] = w;
          back[(size_t)s * Q + m.dst] = Back{(int32_t)((size_t)s * Q + q), -1, ai};
          work.push_back(m.dst);
          if (++guard > (size_t)Q * Q + 16) throw runtime_error("composeShortestPath: the LM has an epsilon cycle of negative weight");
        }
      }
    }
    if (lfin[s] < INF)
      for (int q = 0; q < Q; q++)
        if (ds[q] < INF && mfin[q] < INF) {
          const float w = ds[q] + (lfin[s] + mfin[q]);
          if (w < best_w) { best_w = w; best_at = (long)((size_t)s * Q + q); }
        }
    for (int li : lout[s]) {
      const scrf_arc& a = lat.arcs[li];
      float* dd = &dist[(size_t)a.dst * Q];
      if (a.olabel == 0) {   // the lattice moves alone
        for (int q = 0; q < Q; q++) {
          if (ds[q] >= INF) continue;
          const float w = ds[q] + (a.w + 0.0f);
          if (w < dd[q]) { dd[q] = w; back[(size_t)a.dst * Q + q] = Back{(int32_t)((size_t)s * Q + q), li, -1}; }
        }
        continue;
      }
      for (int q = 0; q < Q; q++) {
        if (ds[q] >= INF) continue;
        for (int ai : mout[q]) {
          const scrf_arc& m = lm.arcs[ai];
          if (m.ilabel != a.olabel) continue;
          const float w = ds[q] + (a.w + m.w);
          if (w < dd[m.dst]) { dd[m.dst] = w; back[(size_t)a.dst * Q + m.dst] = Back{(int32_t)((size_t)s * Q + q), li, ai}; }
        }
      }
    }
  }
  if (total) *total = best_w;
  if (best_at < 0) return false;
  // backtrace, then the chain without label-free arcs
  struct Step { int il, ol; float w; };
  std::vector<Step> steps;
  for (long at = best_at; back[at].prev >= 0; at = back[at].prev) {
    const Back& b = back[at];
    const float wl = b.la >= 0 ? lat.arcs[b.la].w : 0.0f, wm = b.ma >= 0 ? lm.arcs[b.ma].w : 0.0f;
    steps.push_back(Step{b.la >= 0 ? lat.arcs[b.la].ilabel : 0, b.ma >= 0 ? lm.arcs[b.ma].olabel : 0, wl + wm});
  }
  std::reverse(steps.begin(), steps.end());
  if (best) {
    *best = crf_amd::ArcListFst();
    int cur = best->AddState();
    best->SetStart(cur);
    float carry = 0.0f;
    for (const Step& st : steps) {
      if (st.il == 0 && st.ol == 0) { carry += st.w; continue; }
      const int nxt = best->AddState();
      best->AddArc(cur, crf_amd::ArcListFst::Arc(st.il, st.ol, carry + st.w, nxt));
      carry = 0.0f;
      cur = nxt;
    }
    const int sf = (int)(best_at / Q), qf = (int)(best_at % Q);
    best->SetFinal(cur, carry + (lfin[sf] + mfin[qf]));
  }
  if (n_expanded) *n_expanded = n_exp;
  return true;
}

void crf_amd::composeFst(const crf_amd::ArcListFst& A, const crf_amd::ArcListFst& B, crf_amd::ArcListFst* out, size_t max_states, bool sequence_filter) {
  if (A.n_states <= 0 || B.n_states <= 0 || A.start < 0 || B.start < 0) throw runtime_error("composeFst: a machine has no start state");
  const float INF = std::numeric_limits<float>::infinity();
  std::vector<std::vector<int> > aout(A.n_states), bout(B.n_states);
  for (size_t i = 0; i < A.arcs.size(); i++) {
    const scrf_arc& x = A.arcs[i];
    if (x.src < 0 || x.src >= A.n_states || x.dst < 0 || x.dst >= A.n_states) throw runtime_error("composeFst: arc with a state out of range (left machine)");
    aout[x.src].push_back((int)i);
  }
  for (size_t i = 0; i < B.arcs.size(); i++) {
    const scrf_arc& x = B.arcs[i];
    if (x.src < 0 || x.src >= B.n_states || x.dst < 0 || x.dst >= B.n_states) throw runtime_error("composeFst: arc with a state out of range (right machine)");
    bout[x.src].push_back((int)i);
  }
  std::vector<float> afin(A.n_states, INF), bfin(B.n_states, INF);
  for (const auto& f : A.finals) if (f.first >= 0 && f.first < A.n_states) afin[f.first] = std::min(afin[f.first], f.second);
  for (const auto& f : B.finals) if (f.first >= 0 && f.first < B.n_states) bfin[f.first] = std::min(bfin[f.first], f.second);
  *out = crf_amd::ArcListFst();
  // a state of the result: (state of A, state of B, filter state); the filter state is 1 after `B` has moved alone and
  // until the next label match -- `A` may not move alone there (always 0 without the filter)
  struct Triple { int sa, sb, fs; bool operator<(const Triple& o) const { return sa != o.sa ? sa < o.sa : (sb != o.sb ? sb < o.sb : fs < o.fs); } };
  std::map<Triple, int> id;
  std::vector<Triple> states;
  auto state_of = [&](int sa, int sb, int fs) -> int {
    const Triple t{sa, sb, fs};
    auto it = id.find(t);
    if (it != id.end()) return it->second;
    if (states.size() >= max_states) throw runtime_error("composeFst: more than " + std::to_string(max_states) + " state pairs");
    const int n = out->AddState();
    id[t] = n;
    states.push_back(t);
    return n;
  };
  out->SetStart(state_of(A.start, B.start, 0));
  for (size_t k = 0; k < states.size(); k++) {
    const int sa = states[k].sa, sb = states[k].sb, fs = states[k].fs, me = (int)k;
    if (afin[sa] < INF && bfin[sb] < INF) out->SetFinal(me, afin[sa] + bfin[sb]);
    for (int ai : aout[sa]) {
      const scrf_arc& x = A.arcs[ai];
      if (x.olabel == 0) {
        if (fs == 0) out->AddArc(me, crf_amd::ArcListFst::Arc(x.ilabel, 0, x.w + 0.0f, state_of(x.dst, sb, 0)));
        continue;
      }
      for (int bi : bout[sb]) {
        const scrf_arc& y = B.arcs[bi];
        if (y.ilabel != x.olabel) continue;
        out->AddArc(me, crf_amd::ArcListFst::Arc(x.ilabel, y.olabel, x.w + y.w, state_of(x.dst, y.dst, 0)));
      }
    }
    for (int bi : bout[sb]) {
      const scrf_arc& y = B.arcs[bi];
      if (y.ilabel != 0) continue;
      out->AddArc(me, crf_amd::ArcListFst::Arc(0, y.olabel, 0.0f + y.w, state_of(sa, y.dst, sequence_filter ? 1 : 0)));
    }
  }
}

// Plus of OpenFST's LogWeight on float values: -log(e^-a + e^-b)
static inline float log_plus(float a, float b) {
  const float INF = std::numeric_limits<float>::infinity();
  if (a == INF) return b;
  if (b == INF) return a;
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return (float)((double)lo - std::log(1.0 + std::exp(-(double)(hi - lo))));
}

void crf_amd::rmEpsilonLog(crf_amd::ArcListFst* fst) {
  const int n = fst->n_states;
  if (n <= 0 || fst->start < 0) return;
  const float INF = std::numeric_limits<float>::infinity();
  std::vector<std::vector<int> > eps(n), lab(n);
  std::vector<int> indeg(n, 0);
  for (size_t i = 0; i < fst->arcs.size(); i++) {
    const scrf_arc& a = fst->arcs[i];
    if (a.src < 0 || a.src >= n || a.dst < 0 || a.dst >= n) throw runtime_error("rmEpsilonLog: arc with a state out of range");
    if (a.ilabel == 0 && a.olabel == 0) { eps[a.src].push_back((int)i); indeg[a.dst]++; }
    else lab[a.src].push_back((int)i);
  }
  std::vector<float> fin(n, INF);
  for (const auto& f : fst->finals) if (f.first >= 0 && f.first < n) fin[f.first] = std::min(fin[f.first], f.second);
  // topological position of every state in the epsilon subgraph (Kahn); a state left over sits on an epsilon cycle
  std::vector<int> pos(n, -1), ready;
  for (int s = 0; s < n; s++) if (indeg[s] == 0) ready.push_back(s);
  int npos = 0;
  for (size_t k = 0; k < ready.size(); k++) {
    const int s = ready[k];
    pos[s] = npos++;
    for (int ai : eps[s]) if (--indeg[fst->arcs[ai].dst] == 0) ready.push_back(fst->arcs[ai].dst);
  }
  if (npos != n) throw runtime_error("rmEpsilonLog: the machine has an epsilon cycle (its log-semiring closure is a series: not built)");
  // states that stay reachable: the start state and every target of a labelled arc
  std::vector<char> keep(n, 0);
  keep[fst->start] = 1;
  for (const scrf_arc& a : fst->arcs) if (!(a.ilabel == 0 && a.olabel == 0)) keep[a.dst] = 1;
  crf_amd::ArcListFst res;
  std::vector<float> dist(n, INF);
  std::vector<std::pair<int, int> > heap;   // (topological position, state): min-heap, so a state pops after all its closure predecessors
  std::vector<int> touched;
  struct Key { int il, ol, dst; bool operator<(const Key& o) const { return il != o.il ? il < o.il : (ol != o.ol ? ol < o.ol : dst < o.dst); } };
  std::vector<std::vector<scrf_arc> > new_arcs(n);
  std::vector<float> new_fin(n, INF);
  for (int p = 0; p < n; p++) {
    if (!keep[p]) continue;
    touched.clear();
    heap.clear();
    dist[p] = 0.0f;
    touched.push_back(p);
    heap.push_back(std::make_pair(pos[p], p));
    std::map<Key, size_t> slot;
    std::vector<scrf_arc>& outp = new_arcs[p];
    float fw = INF;
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), std::greater<std::pair<int, int> >());
      const int q = heap.back().second;
      heap.pop_back();
      const float d = dist[q];
      for (int ai : eps[q]) {
        const scrf_arc& a = fst->arcs[ai];
        if (dist[a.dst] == INF) {
          touched.push_back(a.dst);
          heap.push_back(std::make_pair(pos[a.dst], a.dst));
          std::push_heap(heap.begin(), heap.end(), std::greater<std::pair<int, int> >());
        }
        dist[a.dst] = log_plus(dist[a.dst], d + a.w);
      }
      for (int ai : lab[q]) {
        const scrf_arc& a = fst->arcs[ai];
        const Key k{a.ilabel, a.olabel, a.dst};
        auto it = slot.find(k);
        if (it == slot.end()) { slot[k] = outp.size(); outp.push_back(scrf_arc{p, a.ilabel, a.olabel, d + a.w, a.dst}); }
        else outp[it->second].w = log_plus(outp[it->second].w, d + a.w);
      }
      if (fin[q] < INF) fw = log_plus(fw, d + fin[q]);
    }
    new_fin[p] = fw;
    for (int t : touched) dist[t] = INF;
  }
  // drop what the start state no longer reaches; ids keep their relative order
  std::vector<char> seen(n, 0);
  std::vector<int> work(1, fst->start);
  seen[fst->start] = 1;
  for (size_t k = 0; k < work.size(); k++)
    for (const scrf_arc& a : new_arcs[work[k]]) if (!seen[a.dst]) { seen[a.dst] = 1; work.push_back(a.dst); }
  std::vector<int> renum(n, -1);
  for (int s = 0; s < n; s++) if (seen[s]) renum[s] = res.AddState();
  res.SetStart(renum[fst->start]);
  for (int s = 0; s < n; s++) {
    if (!seen[s]) continue;
    for (const scrf_arc& a : new_arcs[s]) res.AddArc(renum[s], crf_amd::ArcListFst::Arc(a.ilabel, a.olabel, a.w, renum[a.dst]));
    if (new_fin[s] < INF) res.SetFinal(renum[s], new_fin[s]);
  }
  *fst = res;
}

bool crf_amd::topSortFst(crf_amd::ArcListFst* fst) {
  const int n = fst->n_states;
  if (n <= 0) return true;
  std::vector<std::vector<int> > out(n);
  for (size_t i = 0; i < fst->arcs.size(); i++) {
    const scrf_arc& a = fst->arcs[i];
    if (a.src < 0 || a.src >= n || a.dst < 0 || a.dst >= n) throw runtime_error("topSortFst: arc with a state out of range");
    out[a.src].push_back(a.dst);
  }
  // iterative depth-first search from the start state, then from every state not yet seen; reverse post-order
  std::vector<char> colour(n, 0);   // 0 unseen, 1 on the stack, 2 finished
  std::vector<int> finish;
  std::vector<std::pair<int, size_t> > stack;
  auto visit = [&](int root) -> bool {
    if (colour[root]) return true;
    stack.push_back(std::make_pair(root, (size_t)0));
    colour[root] = 1;
    while (!stack.empty()) {
      const int s = stack.back().first;
      if (stack.back().second < out[s].size()) {
        const int d = out[s][stack.back().second++];
        if (colour[d] == 1) return false;   // back arc: a cycle
        if (colour[d] == 0) { colour[d] = 1; stack.push_back(std::make_pair(d, (size_t)0)); }
      } else {
        colour[s] = 2;
        finish.push_back(s);
        stack.pop_back();
      }
    }
    return true;
  };
  if (fst->start >= 0 && !visit(fst->start)) return false;
  for (int s = 0; s < n; s++) if (!visit(s)) return false;
  std::vector<int> renum(n, -1);
  for (int k = 0; k < n; k++) renum[finish[n - 1 - k]] = k;
  for (scrf_arc& a : fst->arcs) { a.src = renum[a.src]; a.dst = renum[a.dst]; }
  for (auto& f : fst->finals) f.first = renum[f.first];
  if (fst->start >= 0) fst->start = renum[fst->start];
  if (fst->final_state >= 0) fst->final_state = renum[fst->final_state];
  return true;
}

void crf_amd::pruneFst(const crf_amd::ArcListFst& in, crf_amd::ArcListFst* out, float threshold) {
  crf_amd::ArcListFst m = in;
  *out = crf_amd::ArcListFst();
  if (m.n_states <= 0 || m.start < 0) return;
  if (!crf_amd::topSortFst(&m)) throw runtime_error("pruneFst: the machine has a cycle (acyclic lattices only)");
  const int n = m.n_states;
  const float INF = std::numeric_limits<float>::infinity();
  std::vector<std::vector<int> > outa(n);
  for (size_t i = 0; i < m.arcs.size(); i++) outa[m.arcs[i].src].push_back((int)i);
  std::vector<float> fin(n, INF), fd(n, INF), bd(n, INF);
  for (const auto& f : m.finals) fin[f.first] = std::min(fin[f.first], f.second);
  fd[m.start] = 0.0f;
  for (int s = 0; s < n; s++) {       // ids are a topological order now
    if (fd[s] == INF) continue;
    for (int ai : outa[s]) { const scrf_arc& a = m.arcs[ai]; fd[a.dst] = std::min(fd[a.dst], fd[s] + a.w); }
  }
  for (int s = n - 1; s >= 0; s--) {
    bd[s] = fin[s];
    for (int ai : outa[s]) { const scrf_arc& a = m.arcs[ai]; if (bd[a.dst] < INF) bd[s] = std::min(bd[s], bd[a.dst] + a.w); }
  }
  if (bd[m.start] == INF) return;      // no successful path: nothing survives
  const float limit = bd[m.start] + threshold;
  std::vector<char> seen(n, 0);
  std::vector<int> work(1, m.start);
  seen[m.start] = 1;
  std::vector<char> arc_ok(m.arcs.size(), 0);
  for (size_t k = 0; k < work.size(); k++) {
    const int s = work[k];
    for (int ai : outa[s]) {
      const scrf_arc& a = m.arcs[ai];
      if (bd[a.dst] == INF) continue;
      const float w = (fd[s] + a.w) + bd[a.dst];
      if (limit < w) continue;
      arc_ok[ai] = 1;
      if (!seen[a.dst]) { seen[a.dst] = 1; work.push_back(a.dst); }
    }
  }
  std::vector<int> renum(n, -1);
  for (int s = 0; s < n; s++) if (seen[s]) renum[s] = out->AddState();
  out->SetStart(renum[m.start]);
  for (size_t i = 0; i < m.arcs.size(); i++)
    if (arc_ok[i]) { const scrf_arc& a = m.arcs[i]; out->AddArc(renum[a.src], crf_amd::ArcListFst::Arc(a.ilabel, a.olabel, a.w, renum[a.dst])); }
  for (int s = 0; s < n; s++)
    if (seen[s] && fin[s] < INF && !(limit < fd[s] + fin[s])) out->SetFinal(renum[s], fin[s]);
}

// ------------------------------------------------------------------------------------------
// CRF_MLFManager (io/CRF_MLFManager.cpp:19-153)
// ------------------------------------------------------------------------------------------
CRF_MLFManager::CRF_MLFManager(const char* mlffile, const char* olist, const std::map<string, long>* st) : symTab(st) {
  (void)olist;   // the reference takes it and does not read it either (:19-22)
  readMLF(mlffile);
}

string CRF_MLFManager::getKey(const string& fname) {   // :36-40, npos arithmetic included
  const int lastc = (int)fname.rfind("."), firstc = (int)fname.rfind("/");
  if (lastc - firstc - 1 < 0) return "";
  return fname.substr((size_t)(firstc + 1), (size_t)(lastc - firstc - 1));
}

void CRF_MLFManager::readMLF(const char* mlffile) {
  std::ifstream ifile(mlffile);
  if (symTab == nullptr) throw runtime_error("SymbolTable undefined in CRF_MLFManager");
  if (!ifile.is_open()) throw runtime_error(string("Unable to read MLF from file ") + mlffile);
  bool isMLF = false;
  int count = 0;
  string s;
  while (getline(ifile, s)) {
    if (s == "#!MLF!#") { isMLF = true; continue; }
    if (!isMLF) throw runtime_error(string("File ") + mlffile + " is not a wellformed MLF");
    if (s.empty()) continue;
    if (s[0] == '"' && s[s.size() - 1] == '"') {
      const string key = getKey(s);
      if (key.empty()) throw runtime_error("CRF_MLFManager error finding key in string: " + s);
      transcripts.push_back(std::vector<int>());
      fnameTable[key] = count;
    } else if (s[0] == '.' && s.size() <= 1) {
      count++;
    } else {
      if ((size_t)count >= transcripts.size()) throw runtime_error(string("File ") + mlffile + ": a label line before the first entry name");
      auto it = symTab->find(s);
      transcripts[count].push_back(it == symTab->end() ? -1 : (int)it->second);   // SymbolTable::Find: -1 when missing
    }
  }
}

void CRF_MLFManager::getFst(const string& fname, crf_amd::ArcListFst* fst) {
  const string key = getKey(fname);
  if (key.empty()) throw runtime_error("Unable to acquire key from filename " + fname);
  auto it = fnameTable.find(key);
  // the reference indexes the table with operator[] (a missing key silently becomes transcript 0, :134); an error here
  if (it == fnameTable.end()) throw runtime_error("CRF_MLFManager: no MLF entry with key '" + key + "' (from " + fname + ")");
  const std::vector<int>& tr = transcripts.at((size_t)it->second);
  if (tr.empty()) throw runtime_error("CRF_MLFManager: the MLF entry '" + key + "' is empty");   // the reference would SetFinal an unset state
  *fst = crf_amd::ArcListFst();
  int prev = fst->AddState();
  fst->SetStart(prev);
  for (int sym : tr) {
    const int cur = fst->AddState();
    fst->AddArc(prev, crf_amd::ArcListFst::Arc(sym, sym, 0.0f, cur));
    prev = cur;
  }
  fst->SetFinal(prev, 0.0f);
}

void crf_amd::writeFstBinary(const char* fname, const crf_amd::ArcListFst& fst, const char* arc_type) {
  std::ofstream f(fname, std::ios::binary);
  if (!f.is_open()) throw runtime_error(string("writeFstBinary: cannot open ") + fname);
  auto put = [&](const void* p, size_t n) { f.write((const char*)p, (std::streamsize)n); };
  auto put_str = [&](const string& s) { const int32_t n = (int32_t)s.size(); put(&n, 4); put(s.data(), s.size()); };
  const int32_t magic = 2125659606, version = 2, flags = 0;
  const uint64_t props = 0;
  const int64_t start = fst.start, ns = fst.n_states, na = (int64_t)fst.arcs.size();
  put(&magic, 4); put_str("vector"); put_str(arc_type); put(&version, 4); put(&flags, 4); put(&props, 8);
  put(&start, 8); put(&ns, 8); put(&na, 8);
  std::vector<std::vector<const scrf_arc*> > out(fst.n_states);
  for (const scrf_arc& a : fst.arcs) out[a.src].push_back(&a);
  std::vector<float> fin(fst.n_states, std::numeric_limits<float>::infinity());
  for (const auto& fw : fst.finals) fin[fw.first] = fw.second;
  for (int s = 0; s < fst.n_states; s++) {
    put(&fin[s], 4);
    const int64_t n = (int64_t)out[s].size();
    put(&n, 8);
    for (const scrf_arc* a : out[s]) { put(&a->ilabel, 4); put(&a->olabel, 4); put(&a->w, 4); put(&a->dst, 4); }
  }
}

int CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::decodeNState(const crf_amd::ArcListFst* lm, double beam, crf_amd::ArcListFst* result_fst) {
  const modeltype mt = crf->getModelType();
  if (mt != STDFRAME && mt != STDSEG_NO_DUR_NO_TRANSFTR && mt != STDSEG_NO_DUR_NO_SEGTRANSFTR)
    throw runtime_error("nStateDecode: crf_states > 1 is built for stdframe and stdseg_no_dur_no_segtransftr");
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  const uint32_t T = utts[0].u.T, L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs();
  const uint32_t K = crf->getFeatureMap()->getNumStates(), P = L / K;
  segs.clear();
  zx = 0.0;
  best_weight = 0.0f;
  if (T == 0) return 0;
  uint64_t na = 0;
  uint32_t n_states = 0;
  int32_t fin = -1;
  e->check(scrf_lattice_arcs(e->h, g.b, 0, 0, nullptr, &na, &n_states, &fin), "nStateDecode");
  std::vector<scrf_arc> arcs(na);
  e->check(scrf_lattice_arcs(e->h, g.b, 0, 0, arcs.data(), &na, &n_states, &fin), "nStateDecode");
  e->check(scrf_forward_backward(e->h, g.b, 0, SCRF_PREC_EXACT, nullptr, nullptr, nullptr, &zx), "nStateDecode");
  // every state but the start (0) and the final one belongs to a label: (state - 1) % L in the frame lattice
  // (1 + t*L + c) and in the segmental one (boundary and segment states of a node are two runs of L)
  crf_amd::ArcListFst lat;
  lat.n_states = (int)n_states;
  lat.start = 0;
  lat.arcs.reserve(arcs.size());
  for (const scrf_arc& a : arcs) {
    if (a.dst == fin) {   // leave the utterance from a phone's end state
      const uint32_t p = (uint32_t)(a.src - 1) % L;
      if ((p + 1) % K == 0) lat.finals.push_back(std::make_pair((int)a.src, a.w));
      continue;
    }
    const uint32_t c = (uint32_t)(a.dst - 1) % L;
    scrf_arc x = a;
    x.olabel = 0;
    if (a.src == 0) {     // the utterance opens with a phone's start state
      if (c % K != 0) continue;
      x.olabel = (int32_t)(c / K) + 1;
    } else {
      const uint32_t p = (uint32_t)(a.src - 1) % L;
      if (p != c && c % K == 0) x.olabel = (int32_t)(c / K) + 1;   // end state of a phone -> start state of the next
    }
    lat.arcs.push_back(x);
  }
  crf_amd::ArcListFst loop;
  if (!lm) {   // createFreePhoneLmFst: one state, every phone, weight 0
    const int s0 = loop.AddState();
    loop.SetStart(s0);
    for (uint32_t p = 0; p < P; p++) loop.AddArc(s0, crf_amd::ArcListFst::Arc((int)p + 1, (int)p + 1, 0.0f, s0));
    loop.SetFinal(s0, 0.0f);
    lm = &loop;
  }
  // The reference's time-synchronous beam (pruning() :976-1060): the hypotheses of a node are its states a segment can
  // END in -- every state of a frame in the frame lattice (1 + t L + c), the second run of L states of a node in the
  // segmental one (boundary states first: decoders/...WithoutSegTransFtr.h:248-330) -- kept iff below the node's
  // minimum + beam.  beam <= 0: exhaustive.
  std::vector<int> group;
  if (beam > 0) {
    group.assign(n_states, -1);
    if (mt == STDFRAME) {
      for (uint32_t t = 0; t < T; t++) for (uint32_t c = 0; c < L; c++) group[1 + t * L + c] = (int)t;
    } else {
      for (uint32_t c = 0; c < L; c++) group[1 + c] = 0;
      for (uint32_t t = 1; t < T; t++) {
        const uint32_t s0 = 1 + L + (t - 1) * 2 * L;
        for (uint32_t c = 0; c < L; c++) group[s0 + L + c] = (int)t;
      }
    }
  }
  float total = 0.0f;
  uint64_t nexp = 0;
  const bool found = crf_amd::composeShortestPath(lat, *lm, result_fst, &total, beam > 0 ? &group : nullptr, beam, &nexp);
  n_hyps = nexp;
  if (!found) {   // "Could not reach end of utterance" (:2141-2147)
    *result_fst = crf_amd::ArcListFst();
    const int s0 = result_fst->AddState(), s1 = result_fst->AddState();
    result_fst->SetStart(s0);
    result_fst->AddArc(s0, crf_amd::ArcListFst::Arc(0, 0, 8, s1));
    result_fst->SetFinal(s1, (float)zx);
    return (int)T;
  }
  best_weight = total;
  result_fst->final_weight += (float)zx;
  for (auto& f : result_fst->finals) f.second += (float)zx;
  return (int)T;
}

int CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::decodeFull(const crf_amd::ArcListFst* lm, double beam, crf_amd::ArcListFst* result_fst, crf_amd::ArcListFst* out_full_fst) {
  if (lm != nullptr) return decodeLm(*lm, beam, result_fst, out_full_fst);
  // the free phone loop (createFreePhoneLmFst, one state per label): state 0 -> phone p on p+1:p+1, phone -> any OTHER
  // phone, every phone state final
  const uint32_t L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs();
  crf_amd::ArcListFst loop;
  const int s0 = loop.AddState();
  loop.SetStart(s0);
  for (uint32_t p = 0; p < L; p++) {
    const int sp = loop.AddState();
    loop.AddArc(s0, crf_amd::ArcListFst::Arc((int)p + 1, (int)p + 1, 0.0f, sp));
    loop.SetFinal(sp, 0.0f);
  }
  for (uint32_t p = 0; p < L; p++)
    for (uint32_t n = 0; n < L; n++)
      if (n != p) loop.AddArc(s0 + 1 + (int)p, crf_amd::ArcListFst::Arc((int)n + 1, (int)n + 1, 0.0f, s0 + 1 + (int)n));
  return decodeLm(loop, beam, result_fst, out_full_fst);
}

int CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::decodeLm(const crf_amd::ArcListFst& lm, double beam, crf_amd::ArcListFst* result_fst, crf_amd::ArcListFst* out_full_fst) {
  require_dense_model(crf, "CRF_ViterbiDecoder_StdSeg_NoSegTransFtr");
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  const uint32_t T = utts[0].u.T, L = crf->getNActualLabs() ? crf->getNActualLabs() : crf->getNLabs(), D = crf->getLabMaxDur();
  segs.clear();
  n_hyps = 0;
  uint64_t n_frames = 0, n_segs = 0, n_arcs = 0;
  uint32_t nu = 0;
  e->check(scrf_batch_info(e->h, g.b, &nu, &n_frames, &n_segs, &n_arcs), "nStateDecode");
  std::vector<double> S((size_t)n_segs * L), M((size_t)T * L * L);
  e->check(scrf_scores(e->h, g.b, 0, S.data(), M.data()), "nStateDecode");
  e->check(scrf_forward_backward(e->h, g.b, 0, SCRF_PREC_EXACT, nullptr, nullptr, nullptr, &zx), "nStateDecode");
  const float INF = 99999.0f;   // the decoder's "infinity"
  const int Q = lm.n_states;
  if (Q <= 0 || lm.start < 0) throw runtime_error("nStateDecode: the LM FST has no start state");
  if ((size_t)Q * L > (size_t)1 << 24) throw runtime_error("nStateDecode: LM too large for the dense (state, phone) search");
  // LM arcs per state; labels above the phone inventory are ignored like disambiguation symbols
  std::vector<std::vector<int> > out(Q);
  for (size_t i = 0; i < lm.arcs.size(); i++) {
    const scrf_arc& a = lm.arcs[i];
    if (a.src < 0 || a.src >= Q || a.dst < 0 || a.dst >= Q) throw runtime_error("nStateDecode: LM arc with a state out of range");
    out[a.src].push_back((int)i);
  }
  std::vector<float> fin(Q, INF);
  for (const auto& fw : lm.finals) if (fw.first >= 0 && fw.first < Q) fin[fw.first] = std::min(fin[fw.first], fw.second);
  // epsilon-input closure of every state: (state reached, summed weight, arcs taken), cheapest first found
  struct Eps { int state; float w; std::vector<int> path; };
  std::vector<std::vector<Eps> > closure(Q);
  for (int q = 0; q < Q; q++) {
    std::vector<Eps>& c = closure[q];
    c.push_back(Eps{q, 0.0f, {}});
    for (size_t k = 0; k < c.size(); k++) {
      const Eps cur = c[k];
      for (int ai : out[cur.state]) {
        const scrf_arc& a = lm.arcs[ai];
        if (a.ilabel != 0) continue;
        const float w = cur.w + a.w;
        bool found = false;
        for (Eps& x : c) if (x.state == a.dst) { found = true; if (w < x.w) { x.w = w; x.path = cur.path; x.path.push_back(ai); } }
        if (!found) { Eps n{a.dst, w, cur.path}; n.path.push_back(ai); c.push_back(n); }
        if (c.size() > 4096) throw runtime_error("nStateDecode: epsilon closure of the LM does not terminate");
      }
    }
  }
  auto seg_row = [&](uint32_t t, uint32_t d) -> size_t {
    const size_t base = t < D ? (size_t)t * (t + 1) / 2 : (size_t)D * (D + 1) / 2 + (size_t)(t - D) * D;
    return base + d - 1;
  };
  // E[t][q][l]: best weight of starting a segment of phone l at frame t in LM state q; F[t][q][l]: of a
  // segment of phone l ending at t.  Back pointers: for E the previous (q, p) and the LM arc taken (-1:
  // internal, -2: from the start); for F the duration.
  const size_t QL = (size_t)Q * L;
  std::vector<float> E((size_t)T * QL, INF), F((size_t)T * QL, INF);
  std::vector<int32_t> Eprev((size_t)T * QL, -1), Earc((size_t)T * QL, -1), Eeps((size_t)T * QL, -1);
  std::vector<uint16_t> Fdur((size_t)T * QL, 0);
  for (uint32_t t = 0; t < T; t++) {
    float* Et = &E[(size_t)t * QL];
    if (t == 0) {
      const std::vector<Eps>& c = closure[lm.start];
      for (size_t k = 0; k < c.size(); k++)
        for (int ai : out[c[k].state]) {
          const scrf_arc& a = lm.arcs[ai];
          if (a.ilabel <= 0 || a.ilabel > (int)L) continue;
          const size_t idx = (size_t)a.dst * L + (a.ilabel - 1);
          const float w = (0.0f + c[k].w) + a.w;
          if (w < Et[idx]) { Et[idx] = w; Eprev[idx] = -1; Earc[idx] = ai; Eeps[idx] = (int32_t)k; }
        }
    } else {
      const float* Fp = &F[(size_t)(t - 1) * QL];
      const double* Mt = &M[(size_t)t * L * L];
      float best_prev = INF;
      if (beam > 0) for (size_t i = 0; i < QL; i++) best_prev = std::min(best_prev, Fp[i]);
      for (int q = 0; q < Q; q++)
        for (uint32_t p = 0; p < L; p++) {
          const float old = Fp[(size_t)q * L + p];
          if (old >= INF || (beam > 0 && !(old < best_prev + beam))) continue;   // pruning(): kept iff weight < min + beam (:1006)
          n_hyps++;
          const size_t base = (size_t)t * QL;
          {  // internal: the phone continues, no LM move
            const size_t idx = (size_t)q * L + p;
            const float w = old + (float)(-1 * Mt[(size_t)p * L + p]);
            if (w < Et[idx]) { Et[idx] = w; Eprev[base + idx] = (int32_t)((size_t)q * L + p); Earc[base + idx] = -1; Eeps[base + idx] = -1; }
          }
          const std::vector<Eps>& c = closure[q];
          for (size_t k = 0; k < c.size(); k++)
            for (int ai : out[c[k].state]) {
              const scrf_arc& a = lm.arcs[ai];
              if (a.ilabel <= 0 || a.ilabel > (int)L) continue;
              const uint32_t l = (uint32_t)a.ilabel - 1;
              const size_t idx = (size_t)a.dst * L + l;
              const float w = ((old + c[k].w) + a.w) + (float)(-1 * Mt[(size_t)p * L + l]);
              if (w < Et[idx]) { Et[idx] = w; Eprev[base + idx] = (int32_t)((size_t)q * L + p); Earc[base + idx] = ai; Eeps[base + idx] = (int32_t)k; }
            }
        }
    }
    // segments ending at t: durations in list order (longest = earliest entered first), strict <
    const uint32_t nd = t + 1 < D ? t + 1 : D;
    float* Ft = &F[(size_t)t * QL];
    for (size_t idx = 0; idx < QL; idx++) {
      const uint32_t l = (uint32_t)(idx % L);
      float best = INF;
      uint16_t arg = 0;
      for (uint32_t d = nd; d >= 1; d--) {
        const float w0 = E[(size_t)(t - d + 1) * QL + idx];
        if (w0 >= INF) continue;
        const float w = w0 + (float)(-1 * S[seg_row(t, d) * L + l]);
        if (arg == 0 || w < best) { best = w; arg = (uint16_t)d; }
      }
      Ft[idx] = arg ? best : INF;
      Fdur[(size_t)t * QL + idx] = arg;
    }
  }
  // final: LM final states through the epsilon closure
  float min_weight = INF, fin_w = 0.0f;
  long best_idx = -1;
  int fin_eps = -1;
  if (T > 0) {
    const float* Fl = &F[(size_t)(T - 1) * QL];
    for (int q = 0; q < Q; q++)
      for (size_t k = 0; k < closure[q].size(); k++) {
        const Eps& c = closure[q][k];
        if (fin[c.state] >= INF) continue;
        for (uint32_t l = 0; l < L; l++) {
          const float f0 = Fl[(size_t)q * L + l];
          if (f0 >= INF) continue;
          const float w = (f0 + c.w) + fin[c.state];
          if (w < min_weight) { min_weight = w; best_idx = (long)((size_t)q * L + l); fin_eps = (int)k; fin_w = c.w + fin[c.state]; }
        }
      }
  }
  best_weight = min_weight;
  if (out_full_fst != nullptr) {
    // the search lattice: every transition the loops above expanded, for every duration, between hypothesis states
    crf_amd::ArcListFst full;
    const int fs0 = full.AddState();
    full.SetStart(fs0);
    // Lattice states are hypotheses (end frame, LM state, phone).  The reference's output_full_fst keys its states by
    // (end frame, LM state) (stateValueUpdate_onOutputFullFst): the same state set whenever an LM state is entered on
    // one phone only (its own free phone loop, phone-history LMs).  An LM with a state entered on several phones (a
    // unigram / back-off state) gives more states and arcs here than there: said once per run.
    {
      static bool noted = false;
      if (!noted) {
        std::vector<int> in_label(Q, 0);
        bool multi = false;
        for (const scrf_arc& a : lm.arcs)
          if (a.ilabel > 0 && a.ilabel <= (int)L) {
            if (in_label[a.dst] != 0 && in_label[a.dst] != a.ilabel) multi = true;
            in_label[a.dst] = a.ilabel;
          }
        if (multi) std::cerr << "NOTE: crf_if_output_full_lat: the LM has states that are entered on several phones; the full lattice keeps one state per "
                                "(end frame, LM state, phone) where the reference keeps one per (end frame, LM state): its state and arc counts differ "
                                "from the reference's for this LM (the paths and their weights are those of the search)" << std::endl;
        noted = true;
      }
    }
    // (a hash map: the dense T x Q x L table would be gigabytes for a large LM)
    std::unordered_map<uint64_t, int32_t> sid;
    auto state_of = [&](uint32_t t, size_t idx) -> int {
      const uint64_t key = (uint64_t)t * QL + idx;
      auto it = sid.find(key);
      if (it != sid.end()) return it->second;
      const int32_t s_ = full.AddState();
      sid.emplace(key, s_);
      return s_;
    };
    auto word_of = [&](const Eps& c, const scrf_arc& a) -> int {   // one word per arc, as the reference's wrdId: the phone arc's, else the last on the epsilon path
      if (a.olabel != 0) return a.olabel;
      for (size_t i = c.path.size(); i-- > 0;) if (lm.arcs[c.path[i]].olabel != 0) return lm.arcs[c.path[i]].olabel;
      return 0;
    };
    auto fan_out = [&](int from, uint32_t t, size_t idx, float trans, int word) {
      const uint32_t l = (uint32_t)(idx % L);
      for (uint32_t d = 1; d <= D && t + d - 1 < T; d++) {
        const uint32_t te = t + d - 1;
        full.AddArc(from, crf_amd::ArcListFst::Arc((int)l + 1, word, trans + (float)(-1 * S[seg_row(te, d) * L + l]), state_of(te, idx)));
      }
    };
    for (uint32_t t = 0; t < T; t++) {
      if (t == 0) {
        const std::vector<Eps>& c = closure[lm.start];
        for (size_t k = 0; k < c.size(); k++)
          for (int ai : out[c[k].state]) {
            const scrf_arc& a = lm.arcs[ai];
            if (a.ilabel <= 0 || a.ilabel > (int)L) continue;
            fan_out(fs0, 0, (size_t)a.dst * L + (a.ilabel - 1), (0.0f + c[k].w) + a.w, word_of(c[k], a));
          }
        continue;
      }
      const float* Fp = &F[(size_t)(t - 1) * QL];
      const double* Mt = &M[(size_t)t * L * L];
      float best_prev = INF;
      if (beam > 0) for (size_t i = 0; i < QL; i++) best_prev = std::min(best_prev, Fp[i]);
      for (int q = 0; q < Q; q++)
        for (uint32_t p = 0; p < L; p++) {
          const float old = Fp[(size_t)q * L + p];
          if (old >= INF || (beam > 0 && !(old < best_prev + beam))) continue;
          const int from = state_of(t - 1, (size_t)q * L + p);
          fan_out(from, t, (size_t)q * L + p, (float)(-1 * Mt[(size_t)p * L + p]), 0);
          const std::vector<Eps>& c = closure[q];
          for (size_t k = 0; k < c.size(); k++)
            for (int ai : out[c[k].state]) {
              const scrf_arc& a = lm.arcs[ai];
              if (a.ilabel <= 0 || a.ilabel > (int)L) continue;
              const uint32_t l = (uint32_t)a.ilabel - 1;
              fan_out(from, t, (size_t)a.dst * L + l, (c[k].w + a.w) + (float)(-1 * Mt[(size_t)p * L + l]), word_of(c[k], a));
            }
        }
    }
    if (T > 0) {
      const float* Fl = &F[(size_t)(T - 1) * QL];
      float best_last = INF;
      if (beam > 0) for (size_t i = 0; i < QL; i++) best_last = std::min(best_last, Fl[i]);
      for (size_t idx = 0; idx < QL; idx++) {
        if (Fl[idx] >= INF || (beam > 0 && Fl[idx] > best_last + beam)) continue;   // pruneFinal :947-970 erases weight > min + beam
        { auto it = sid.find((uint64_t)(T - 1) * QL + idx); if (it != sid.end()) full.SetFinal(it->second, (float)zx); }
      }
    }
    // Connect: keep the states that lie on a path from the start state to a final state
    const int NS = full.n_states;
    std::vector<std::vector<int> > rin(NS);
    for (const scrf_arc& a : full.arcs) rin[a.dst].push_back(a.src);
    std::vector<char> live(NS, 0);
    std::vector<int> work;
    for (const auto& fw : full.finals) if (!live[fw.first]) { live[fw.first] = 1; work.push_back(fw.first); }
    for (size_t k = 0; k < work.size(); k++)
      for (int s_ : rin[work[k]]) if (!live[s_]) { live[s_] = 1; work.push_back(s_); }
    *out_full_fst = crf_amd::ArcListFst();
    if (live[fs0]) {
      std::vector<int> renum(NS, -1);
      for (int s_ = 0; s_ < NS; s_++) if (live[s_]) renum[s_] = out_full_fst->AddState();
      out_full_fst->SetStart(renum[fs0]);
      for (const scrf_arc& a : full.arcs)
        if (live[a.src] && live[a.dst]) out_full_fst->AddArc(renum[a.src], crf_amd::ArcListFst::Arc(a.ilabel, a.olabel, a.w, renum[a.dst]));
      for (const auto& fw : full.finals) out_full_fst->SetFinal(renum[fw.first], fw.second);
    }
  }
  int cur = result_fst->AddState();
  result_fst->SetStart(cur);
  if (best_idx < 0) {  // "Could not reach end of utterance" (:2141-2147)
    int fs = result_fst->AddState();
    result_fst->AddArc(cur, crf_amd::ArcListFst::Arc(0, 0, 8, fs));
    result_fst->SetFinal(fs, (float)zx);
    return (int)T;
  }
  // backtrace, last segment first
  struct Step { uint32_t phone, dur, start; int arc, eps, q_from, prev_phone; };
  std::vector<Step> steps;
  long idx = best_idx;
  for (long te = (long)T - 1; te >= 0;) {
    const uint32_t d = Fdur[(size_t)te * QL + idx];
    const uint32_t ts = (uint32_t)(te + 1 - d);
    const size_t eidx = (size_t)ts * QL + idx;
    Step st{(uint32_t)(idx % L), d, ts, Earc[eidx], Eeps[eidx], -1, -1};
    if (Eprev[eidx] >= 0) { st.q_from = (int)(Eprev[eidx] / L); st.prev_phone = (int)(Eprev[eidx] % L); }
    else st.q_from = lm.start;
    steps.push_back(st);
    if (ts == 0) break;
    idx = Eprev[eidx];
    te = (long)ts - 1;
  }
  std::reverse(steps.begin(), steps.end());
  for (const Step& st : steps) {
    const uint32_t te = st.start + st.dur - 1;
    const double sv = S[seg_row(te, st.dur) * L + st.phone];
    Segment sg;
    sg.phone = st.phone; sg.dur = st.dur; sg.start = st.start;
    sg.phone_start = st.arc >= 0;
    float lmw = 0.0f;
    int olabel = 0;
    if (st.arc >= 0) {
      const Eps& c = closure[st.q_from][st.eps];
      for (int ai : c.path) {  // epsilon-input LM arcs on the way: words and back-off weights
        const scrf_arc& a = lm.arcs[ai];
        int nxt = result_fst->AddState();
        result_fst->AddArc(cur, crf_amd::ArcListFst::Arc(0, a.olabel, a.w, nxt));
        cur = nxt;
      }
      lmw = lm.arcs[st.arc].w;
      olabel = lm.arcs[st.arc].olabel;
    }
    const float ac = st.prev_phone < 0 ? (float)(-1 * sv) : (float)(-1 * (M[((size_t)te * L + st.prev_phone) * L + st.phone] + sv));
    sg.weight = ac + lmw;
    segs.push_back(sg);
    int nxt = result_fst->AddState();
    result_fst->AddArc(cur, crf_amd::ArcListFst::Arc((int)st.phone + 1, olabel, sg.weight, nxt));
    cur = nxt;
  }
  {
    const Eps& c = closure[(int)(best_idx / L)][fin_eps];
    for (int ai : c.path) {
      const scrf_arc& a = lm.arcs[ai];
      int nxt = result_fst->AddState();
      result_fst->AddArc(cur, crf_amd::ArcListFst::Arc(0, a.olabel, a.w, nxt));
      cur = nxt;
    }
    (void)fin_w;
    result_fst->SetFinal(cur, (float)zx + fin[c.state]);
  }
  return (int)T;
}

std::vector<uint32_t> crf_amd_best_path(CRF_FeatureStream* ftr_strm, CRF_Model* crf, float* cost) {
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts(1);
  grab(ftr_strm, crf, &utts[0]);
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  std::vector<uint32_t> labs(utts[0].u.T);
  uint64_t off[2] = {0, 0};
  float c = 0;
  e->check(scrf_viterbi_batch(e->h, g.b, labs.data(), labs.size(), off, &c), "ShortestPath");
  labs.resize(off[1]);
  if (cost) *cost = c;
  return labs;
}

size_t crf_amd_best_paths(CRF_FeatureStream* ftr_strm, CRF_Model* crf, size_t max_utts,
                          std::vector<std::vector<uint32_t> >* labels, std::vector<float>* costs, bool* at_end) {
  crf_amd::Engine* e = crf->engine();
  crf->pushLambda();
  std::vector<HeldUtt> utts;
  utts.reserve(max_utts);
  *at_end = false;
  size_t frames = 0;
  while (utts.size() < max_utts) {
    utts.emplace_back();
    grab(ftr_strm, crf, &utts.back());
    frames += utts.back().u.T;
    if (ftr_strm->nextseg() == QN_SEGID_BAD) { *at_end = true; break; }
  }
  BatchGuard g{e};
  make_batch(e, ftr_strm, utts, &g);
  std::vector<uint32_t> labs(frames);
  std::vector<uint64_t> off(utts.size() + 1, 0);
  costs->assign(utts.size(), 0.0f);
  e->check(scrf_viterbi_batch(e->h, g.b, labs.data(), labs.size(), off.data(), costs->data()), "ShortestPath");
  labels->resize(utts.size());
  for (size_t u = 0; u < utts.size(); u++) (*labels)[u].assign(labs.begin() + off[u], labs.begin() + off[u + 1]);
  return utts.size();
}
