// crf_amd.h -- C++ host side above the C ABI (include/scrf_abi.h): the reference's plugin /
// operator interface for the hot path, with the reference's own names, argument meaning and
// error behaviour (std::runtime_error), implemented on the MI355X engine.  Written from the
// interface descriptions in SURVEY.md section 8b; nothing here is reference code.
//
//   CRF_FeatureMap_config / CRF_FeatureMap / CRF_StdFeatureMap   ftrmaps/CRF_FeatureMap.h:24-96
//   CRF_Model                                                    CRF_Model.h
//   CRF_FeatureStream (abstract) + CRF_MemoryFeatureStream       io/CRF_FeatureStream.h:54-66
//   CRF_GradBuilder::buildGradient                               trainers/gradbuilders/CRF_GradBuilder.h:40
//   CRF_Minibatch_GradAccumulator::accumulateGradient            trainers/accumulators/...h:66-67
//   CRF_FeatureStreamManager                                     io/CRF_FeatureStreamManager.h:60-100
//   CRF_Trainer / CRF_SGTrainer                                  trainers/CRF_Trainer.h, CRF_SGTrainer.h:45-48
//   CRF_StateNode / CRF_StateVector (read-only node view)        nodes/CRF_StateNode.h:67-115
//   CRF_LatticeBuilder_* ::buildLattice<Fst>                     decoders/...WithoutSegTransFtr.h:26
//
// QuickNet3 and OpenFST are not dependencies: feature streams are an abstract interface the
// caller implements (a QuickNet-backed one is a 30-line subclass), and
// buildLattice is a template over any FST type with AddState/SetStart/AddArc/SetFinal
// (fst::VectorFst<fst::StdArc> fits; crf_amd::ArcListFst is provided for OpenFST-less builds).
#ifndef CRF_AMD_H_
#define CRF_AMD_H_

#include <stdint.h>
#include <stdio.h>

#include <memory>
#include <stdexcept>
#include <map>
#include <string>
#include <vector>

#include "scrf_abi.h"

typedef uint32_t QNUInt32;
typedef int32_t QNInt32;
typedef long QN_SegID;
#define QN_SEGID_BAD (-1L)
#define CRF_LAB_BAD SCRF_LAB_BAD
enum seqtype { SEQUENTIAL, RANDOM_NO_REPLACE, RANDOM_REPLACE };   // CRF.h:38
#define CRF_UINT32_MAX (0xffffffffu)

enum ftrmaptype { STDSTATE, STDTRANS, STDSPARSE, STDSPARSETRANS, INFILE };                       // CRF.h:40
enum modeltype { STDFRAME, STDSEG, STDSEG_NO_DUR, STDSEG_NO_DUR_NO_TRANSFTR, STDSEG_NO_DUR_NO_SEGTRANSFTR };  // CRF.h:50
enum objfunctype { EXPF, EXPFSOFT, FERR };                                                       // CRF.h:44

struct CRF_FeatureMap_config {
  ftrmaptype map_type = STDSTATE;
  QNUInt32 numLabs = 0, numFeas = 0, numStates = 1;
  bool useStateFtrs = true;
  QNUInt32 stateFidxStart = 0, stateFidxEnd = 0;
  bool useTransFtrs = false;
  QNUInt32 transFidxStart = 0, transFidxEnd = 0;
  bool useStateBias = true, useTransBias = true;
  double stateBiasVal = 1.0, transBiasVal = 1.0;
  QNUInt32 maxDur = 1, durFtrStart = 0, nActualLabs = 0;
};

// lambda index layout of the dense maps (host-only arithmetic, same closed form the kernels use)
class CRF_FeatureMap {
 public:
  explicit CRF_FeatureMap(CRF_FeatureMap_config* cnf);
  virtual ~CRF_FeatureMap() {}
  static CRF_FeatureMap* createFeatureMap(CRF_FeatureMap_config* cnf);
  virtual QNUInt32 getNumFtrFuncs() { return numFtrFuncs; }
  virtual QNUInt32 getNumStates() { return config->numStates; }
  virtual QNUInt32 getNumStateFuncs(QNUInt32) { return numStateFuncs; }
  virtual QNUInt32 getNumTransFuncs(QNUInt32, QNUInt32) { return numTransFuncs; }
  virtual QNUInt32 getStateFeatureIdx(QNUInt32 clab, QNUInt32 fno = 0);
  virtual QNUInt32 getTransFeatureIdx(QNUInt32 clab, QNUInt32 plab, QNUInt32 fno = 0);
  virtual QNUInt32 getStateBiasIdx(QNUInt32 clab) { return getStateFeatureIdx(clab, numStateFuncs - 1); }
  virtual QNUInt32 getTransBiasIdx(QNUInt32 clab, QNUInt32 plab) { return getTransFeatureIdx(clab, plab, numTransFuncs - 1); }
  virtual QNUInt32 recalc();
  CRF_FeatureMap_config* getConfig() { return config; }

 protected:
  CRF_FeatureMap_config* config;
  QNUInt32 numFtrFuncs = 0, numStateFuncs = 0, numTransFuncs = 0, numActualLabels = 0;
};
typedef CRF_FeatureMap CRF_StdFeatureMap;

namespace crf_amd { class Engine; }

// lambda / lambdaAcc / gradSqrAcc container + the text weight-file format (one value per line,
// default ostream precision = 6 significant digits, CRF_Model.cpp:205-233,290)
class CRF_Model {
 public:
  explicit CRF_Model(QNUInt32 num_labs);
  virtual ~CRF_Model();
  QNUInt32 getNLabs() { return nlabs; }
  virtual void setFeatureMap(CRF_FeatureMap* map);  // takes ownership; sizes lambda
  virtual CRF_FeatureMap* getFeatureMap() { return featureMap; }
  virtual double* getLambda() { return lambda.data(); }
  virtual QNUInt32 getLambdaLen() { return (QNUInt32)lambda.size(); }
  virtual double* getLambdaAcc() { return lambdaAcc.data(); }
  virtual double* getGradSqrAcc() { return gradSqrAcc.data(); }
  virtual QNUInt32 getPresentations() { return init_present; }
  virtual void setLambda(double* v, QNUInt32 n);
  virtual void resetLambda();
  virtual bool writeToFile(const char* fname);
  virtual bool writeToFile(const char* fname, double* lam, QNUInt32 ll);
  virtual bool readFromFile(const char* fname);
  virtual bool readAverageFromFile(const char* fname, int present);
  virtual bool readGradSqrAccFromFile(const char* fname);
  virtual void setLabMaxDur(QNUInt32 d) { lab_max_dur = d; }
  virtual QNUInt32 getLabMaxDur() { return lab_max_dur; }
  virtual void setNActualLabs(QNUInt32 n) { nActualLabs = n; }
  virtual QNUInt32 getNActualLabs() { return nActualLabs; }
  virtual void setModelType(modeltype m) { model_type = m; }
  virtual modeltype getModelType() { return model_type; }
  virtual void setInitIter(QNUInt32 i) { init_iter = i; }
  virtual QNUInt32 getInitIter() { return init_iter; }
  // the engine bound to this model, created lazily on setDevice()'s GPU with setTrainPrecision()'s
  // arithmetic policy (crf_device= / crf_precision= of the front-ends); lambda is pushed before use
  crf_amd::Engine* engine();
  void setDevice(int d) { device = d; }
  int getDevice() const { return device; }
  void setTrainPrecision(uint32_t p) { precision = p; }   // scrf_precision; decode entry points stay EXACT
  // This model's engine will only ever train (CRFTrain).  With FAST / FAST32 precision the n-state FRAME model
  // (stdframe, crf_states > 1) is then run as the n-state segmental model with maximum duration 1 -- the same function
  // (nodes/CRF_StdNStateNode.cpp against nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp at one frame
  // per segment: gradient, numerator and Zx agree to 1e-12 at the TIMIT shape, tools/experiments/r03_nstate_frame_vs_seg.py),
  // the same weight layout, but on the dense MFMA kernels over the masked layout instead of the
  // reference-order n-state kernels: 10.5 -> 3.7 ms per 256 utterances of 300 frames at 48 phones x 3 states.  Lattices
  // and decoding keep the n-state kernels (their arc order is the frame lattice builder's), hence "training only".
  void setTrainingOnly(bool on) { training_only = on; }
  bool trainingOnly() const { return training_only; }
  // One process per GPU (RANK / WORLD_SIZE of the launcher): rank r is the reference's stream (thread) r.
  // The RCCL communicator is created together with the engine; its 128-byte unique id travels from rank 0
  // to the others through `id_file` (rank 0 writes it, the others poll; removed after the collective
  // initialisation).  world == 1 with a non-empty id_file exercises the same path on one GPU.
  void setDistributed(int rank, int world, const std::string& id_file);
  int distRank() const { return dist_rank; }
  int distWorld() const { return dist_world; }
  bool distributed() const { return dist_on; }
  void pushLambda();   // host lambda/lambdaAcc/gradSqrAcc -> device
  void pullLambda();   // device -> host

 protected:
  QNUInt32 nlabs;
  std::vector<double> lambda, lambdaAcc, gradSqrAcc;
  CRF_FeatureMap* featureMap = nullptr;
  QNUInt32 init_present = 0, lab_max_dur = 1, nActualLabs = 0, init_iter = 0;
  modeltype model_type = STDFRAME;
  int device = 0;
  uint32_t precision = SCRF_PREC_FAST;
  bool training_only = false;
  bool dist_on = false;
  int dist_rank = 0, dist_world = 1;
  std::string dist_id_file;
  std::unique_ptr<crf_amd::Engine> eng;
};

// The stream interface the hot path consumes (io/CRF_FeatureStream.h:54-66).  read() returns
// `bunch` windows ending at the current frame, each num_ftrs() floats, and the 4 label words
// {label, start, end, broken} of the segment ending there (CRF_LAB_BAD x4 if none).
class CRF_FeatureStream {
 public:
  virtual ~CRF_FeatureStream() {}
  virtual QN_SegID nextseg() = 0;
  virtual size_t read(size_t bunch, float* ftr_buf, QNUInt32* lab_buf) = 0;
  virtual int rewind() = 0;
  virtual size_t num_ftrs() = 0;
  virtual size_t num_labs() = 0;
  // restrict the stream to `count` utterances from `start` (io/CRF_FeatureStream.cpp:345; what
  // CRF_FeatureStreamManager::create does for child i, :458-460); count == CRF_UINT32_MAX (QN_ALL): to the end
  virtual void view(size_t start, size_t count) { (void)start; (void)count; throw std::runtime_error("CRF_FeatureStream::view: not supported by this stream"); }
  // Engine fast path: whole utterance at once.  frames[s] = raw frames of stream s incl. context
  // padding (empty when only windows are available); either labels = per-end-frame segment labels
  // nActualLabs*(dur-1)+phone, or phones + starts (phone id and first frame of the segment ending at each
  // frame, CRF_LAB_BAD where none ends) from which the caller forms them for its model.
  struct Utterance {
    uint32_t T = 0;
    std::vector<const float*> frames;
    const float* windows = nullptr;
    const uint32_t* labels = nullptr;
    const uint32_t* phones = nullptr;
    const uint32_t* starts = nullptr;
  };
  virtual bool currentUtterance(Utterance* u) { (void)u; return false; }
  virtual const std::vector<scrf_stream_recipe>& recipes() { static std::vector<scrf_stream_recipe> e; return e; }
};

// In-memory stream over raw frames (what CRF_FeatureStreamManager builds from pfiles):
// per utterance one frame matrix per input stream + frame-level phone labels.
class CRF_MemoryFeatureStream : public CRF_FeatureStream {
 public:
  CRF_MemoryFeatureStream(std::vector<scrf_stream_recipe> recipes, QNUInt32 max_dur, QNUInt32 n_actual_labs = 0);
  ~CRF_MemoryFeatureStream() override;
  // frames[s]: (T + lctx_s + rctx_s) x in_width_s, frame_labels: T phone ids (or empty)
  void addUtterance(const std::vector<std::vector<float> >& frames, const std::vector<uint32_t>& frame_labels);
  CRF_MemoryFeatureStream* makeView(size_t start, size_t count);  // new child stream over a contiguous range
  void view(size_t start, size_t count) override;                 // the same in place
  // feature concatenation (CRF_FeatureStream::join, used by CRF_FeatureStreamManager::join): the other
  // stream's input streams become further streams of this one; same utterances, same lengths
  void join(const CRF_MemoryFeatureStream& other);
  // presentation order of the utterances (io/CRF_InFtrStream_RandPresent.cpp): a new order at every
  // rewind(), generator seeded with 12345 * epoch + seed as there (:125-128); RANDOM_REPLACE draws
  // numUtterances() utterances with replacement, RANDOM_NO_REPLACE a permutation.  The generator is
  // std::mt19937_64 with a plain modulo draw -- QuickNet's QN_SeqGen_* are not in the tree, so the
  // ORDER differs from the reference's for the same seed.  Views made afterwards inherit the mode.
  void setPresentation(seqtype type, QNUInt32 seed);
  size_t numUtterances() const { return end_ - begin_; }
  QN_SegID nextseg() override;
  size_t read(size_t bunch, float* ftr_buf, QNUInt32* lab_buf) override;
  int rewind() override;
  size_t num_ftrs() override { return width_; }
  size_t num_labs() override { return 4; }
  bool currentUtterance(Utterance* u) override;
  const std::vector<scrf_stream_recipe>& recipes() override { return store_->recipes; }

 private:
  struct Store {
    std::vector<scrf_stream_recipe> recipes;
    QNUInt32 D;
    std::vector<uint32_t> T;
    std::vector<std::vector<std::vector<float> > > frames;  // [utt][stream]
    std::vector<std::vector<uint32_t> > seg_phone;          // [utt][T] phone of the segment ending at t (CRF_LAB_BAD: none)
    std::vector<std::vector<uint32_t> > seg_start;          // [utt][T] its first frame
  };
  void fetchWindows();   // read(): the current utterance's window vectors, synthesised on the GPU
 public:
  // an utterance this process never reads (another rank's share, crf_amd::setProcessView): keeps the numbering, holds no frames
  void addPlaceholder();
 private:
  std::shared_ptr<Store> store_;
  size_t begin_ = 0, end_ = 0, width_ = 0;
  long cur_ = -1;
  uint32_t frame_ = 0;
  seqtype mode_ = SEQUENTIAL;
  QNUInt32 seed_ = 0, epoch_ = 0;
  std::vector<size_t> order_;   // utterance offsets inside [begin_, end_) in presentation order
  size_t pos_ = 0;
  // read() protocol: the window vectors come from the engine's k_windows through a private handle
  // (one utterance at a time); the host never restates the recipe
  scrf_handle win_eng_ = nullptr;
  long win_utt_ = -1;
  std::vector<float> win_cache_;
};

// The reference's stream factory (io/CRF_FeatureStreamManager.{h,cpp}): one feature file (pfile or ascii)
// + the hard-target label file -> trn_stream (and cv_stream when a CV range is given), `n_threads` children
// whose trn_stream views the contiguous utterance range [i*floor(n/N), ...) (the last child takes the
// remainder, create() :425-464), join() for feature concatenation.  Same constructor arguments as the
// reference.  Built here: formats pfile / ascii; ftr_width is checked against the file; win_len is the
// label maximum duration (window_extent == win_len, offsets 0); deltas, norm files and boundary-delta
// features are refused.  The utterance selection is QN_Range syntax (qn_files.h).
class CRF_FeatureStreamManager {
 public:
  CRF_FeatureStreamManager(int debug, const char* debug_name, char* ftr_fname, const char* ftr_file_fmt, char* ht_fname,
                           size_t ht_offset, size_t ftr_width, size_t first_ftr, size_t num_ftrs, size_t win_ext,
                           size_t win_off, size_t win_len, size_t left_ctx_len, size_t right_ctx_len, bool extract_seg_ftr,
                           bool use_bdy_delta_ftr, int delta_o, int delta_w, char* trn_rng, char* cv_rng, FILE* nfile,
                           int n_mode, double n_am, double n_av, seqtype ts, QNUInt32 rseed = 0, size_t n_threads = 1);
  virtual ~CRF_FeatureStreamManager();
  void join(CRF_FeatureStreamManager* other);
  size_t getNumFtrs();
  CRF_FeatureStreamManager* getChild(size_t child) { return child < children.size() ? children[child].get() : nullptr; }
  size_t getNThreads() { return nthreads; }
  void rewindAllChildrenTrn();
  void display();

  CRF_FeatureStream* trn_stream = nullptr;
  CRF_FeatureStream* cv_stream = nullptr;
  CRF_FeatureStream* old_trn_stream = nullptr;

 protected:
  CRF_FeatureStreamManager() {}
  size_t nthreads = 1;
  std::unique_ptr<CRF_MemoryFeatureStream> trn, cv;
  std::vector<std::unique_ptr<CRF_FeatureStreamManager> > children;
};

namespace crf_amd {

// RAII over scrf_handle; every failure becomes std::runtime_error(scrf_last_error)
class Engine {
 public:
  Engine(const scrf_config& cfg);
  ~Engine();
  scrf_handle h = nullptr;
  uint32_t lambda_len = 0;
  void check(int rc, const char* what);
};

scrf_config makeConfig(CRF_Model* crf, int device, uint32_t precision);
bool frameAsSegmental(CRF_Model* crf, uint32_t precision);   // the n-state frame model recast as the n-state segmental model with maximum duration 1
// One process per GPU (CRFTrain under RANK / WORLD_SIZE): rank r only ever walks child view r of the training stream
// (io/CRF_FeatureStreamManager.cpp:425-464: [r floor(n/N), ...)), so a CRF_FeatureStreamManager built afterwards with N
// threads keeps the frames of that view alone and placeholders for the rest -- N ranks hold the training data once between
// them instead of N times.  The file is still parsed by every rank (pfiles carry no per-sentence seek table worth trusting).
void setProcessView(int rank, int world);

// minimal FST recorder with the four calls buildLattice needs
struct ArcListFst {
  struct Arc { int ilabel, olabel; float weight; int nextstate; Arc(int i, int o, float w, int n) : ilabel(i), olabel(o), weight(w), nextstate(n) {} };
  int AddState() { return n_states++; }
  void SetStart(int s) { start = s; }
  void AddArc(int s, const Arc& a) { arcs.push_back(scrf_arc{s, a.ilabel, a.olabel, a.weight, a.nextstate}); }
  void SetFinal(int s, float w) { final_state = s; final_weight = w; finals.push_back(std::make_pair(s, w)); }
  int n_states = 0, start = -1, final_state = -1;
  float final_weight = 0;
  std::vector<scrf_arc> arcs;
  std::vector<std::pair<int, float> > finals;   // every SetFinal call (an LM has many final states)
};
// OpenFST text format (what `fstprint` writes with numeric labels): arc lines `src dst ilabel olabel
// [weight]`, final-state lines `state [weight]`; the start state is the source of the first line.
void readFstText(const char* fname, ArcListFst* fst);
// OpenFST binary `vector` FST over the standard (tropical) or log arc type -- what `crf_lm_bin` names
// (CRFDecode/src/Main.cpp:844-855 reads a VectorFst<LogArc> and maps it to the tropical semiring with the
// identity on the float values).  Layout as written by OpenFST 1.x (FstHeader::Write, VectorFst::Write):
// int32 magic 2125659606 | string fst type | string arc type | int32 version | int32 flags | uint64
// properties | int64 start | int64 states | int64 arcs | [symbol tables when flags say so] | per state:
// float final weight (+inf = not final), int64 arc count, arcs {int32 ilabel, int32 olabel, float
// weight, int32 nextstate}; strings are int32 length + bytes; little-endian.  OpenFST is not in the tree,
// so this layout is UNPINNED: every redundancy is checked (types, counts, exact file length) and anything
// unexpected is an error that points at `fstprint` + crf_lm_txt.  writeFstBinary emits the same layout.
void readFstBinary(const char* fname, ArcListFst* fst);
// ShortestPath(Compose(ArcSort(lat, olabel), lm), 1) + RmEpsilon + TopSort of CRFFstDecode's MLF path
// (CRFFstDecode/src/Main.cpp:1002-1023) in one pass, for an ACYCLIC `lat` whose state ids are a topological
// order (the lattices buildLattice emits are) and any `lm` (epsilon-input arcs allowed, no negative epsilon
// cycles): tropical semiring on float; a composed arc weighs Times(lat arc, lm arc) = w_lat + w_lm and a
// path accumulates start -> end, d + (w_lat + w_lm); lattice arcs with an epsilon OUTPUT advance the
// lattice alone, LM arcs with an epsilon INPUT the LM alone; strict-improvement relaxation, lattice states
// ascending, lattice arcs in insertion order, LM arcs in file order (first relaxed wins ties -- OpenFST is
// not in the tree, so tie order against its Compose / ShortestPath is UNPINNED).
// `best` receives the path as a chain (state i -> i+1): one arc per composed arc that carries a label on
// either side (ilabel = the lattice arc's ilabel, olabel = the LM arc's olabel); the weights of label-free
// arcs are folded into the next arc (the final weight at the end), as RmEpsilon does on a linear path.
// Returns false when no path reaches a final state of both machines.
bool composeShortestPath(const ArcListFst& lat, const ArcListFst& lm, ArcListFst* best, float* total,
                         const std::vector<int>* group = nullptr, double beam = 0.0, uint64_t* n_expanded = nullptr);
// Compose(a, b) on the tropical semiring (CRFFstDecode/src/Main.cpp:898-955 chains ComposeFst over the phone lattice,
// the dictionary, the alignment acceptor and the LM): states are the reachable pairs, numbered in discovery order
// (breadth first from the start pair); a pair of arcs with a.olabel == b.ilabel != 0 moves both machines (ilabel of a,
// olabel of b, weight a.w + b.w), an arc of `a` with an epsilon OUTPUT moves `a` alone, an arc of `b` with an epsilon
// INPUT moves `b` alone; a pair is final when both states are, with the sum of the final weights.  Without the filter
// the interleavings of a's and b's epsilon moves stay as parallel paths of equal weight, which a shortest-path search
// on the result does not care about (only the machine's size differs).
// Throws when the result would pass `max_states` pairs.
// `sequence_filter`: OpenFST's sequencing epsilon filter (the default of its ComposeFst) -- between two label matches
// the epsilon-output moves of `a` all come before the epsilon-input moves of `b`, so every (path of a, path of b) pair
// shows up as exactly ONE path of the result.  Needed when a log-semiring operation follows (path weights are summed:
// the phone-penalty stage, rmEpsilonLog below); the result's states are then (a state, b state, filter state) triples.
void composeFst(const ArcListFst& a, const ArcListFst& b, ArcListFst* out, size_t max_states = (size_t)1 << 22, bool sequence_filter = false);
// RmEpsilon on the LOG semiring over the float weights, as CRFFstDecode's phone-penalty stage runs it between
// Map(StdToLogMapper) and Map(LogToStdMapper) (CRFFstDecode/src/Main.cpp:904-914): arcs with epsilon on BOTH sides are
// removed; state p gets, for every state q of its epsilon closure (distance d(p,q) = log-sum over all epsilon paths),
// q's labelled arcs with weight d(p,q) + w and q's final weight; arcs of p that end up with the same (ilabel, olabel,
// next state) are merged by log-addition -min(a,b) - log(1 + exp(-|a-b|)) -- so where several epsilon paths lead to the
// same labelled arc the tropical weight that comes out is their log-sum, not their minimum.  States that are no longer
// reachable are dropped (ids keep their relative order).  An epsilon CYCLE throws (its closure is a series, which
// OpenFST truncates at a delta; the lattices and phone machines of this path have none).
void rmEpsilonLog(ArcListFst* fst);
// Prune(in, out, threshold) on the tropical semiring (CRFFstDecode/src/Main.cpp:915-919, :935-939): keeps the arcs and
// final weights that lie on a successful path of weight <= best + threshold, (d(start,src) + w) + d(dst,final) compared
// with the limit as OpenFST's Prune does; acyclic input only (throws otherwise).  An empty result has no states.
void pruneFst(const ArcListFst& in, ArcListFst* out, float threshold);
// TopSort: renumbers the states along a depth-first reverse post-order from the start state (arc order within a state
// kept); returns false and leaves the machine alone when it has a cycle.
bool topSortFst(ArcListFst* fst);
void writeFstBinary(const char* fname, const ArcListFst& fst, const char* arc_type = "standard");

}  // namespace crf_amd

// per-utterance gradient (CRF_GradBuilder::buildGradient, factory create())
class CRF_GradBuilder {
 public:
  static CRF_GradBuilder* create(CRF_Model* crf, objfunctype ofunc);
  explicit CRF_GradBuilder(CRF_Model* crf_in) : crf(crf_in) {}
  virtual ~CRF_GradBuilder() {}
  // grad += observed - expected counts of the stream's CURRENT utterance; *Zx_out = log partition;
  // returns the numerator (caller forms logLi = numerator - Zx)
  virtual double buildGradient(CRF_FeatureStream* ftr_strm, double* grad, double* Zx_out);

 protected:
  CRF_Model* crf;
};

// data-parallel layer: N streams, minibatch split, sum / n_active
// (trainers/accumulators/CRF_Minibatch_GradAccumulator.{h,cpp}).  In one process the streams run one after
// the other on the model's GPU (the gradient sum stays on the device).  With CRF_Model::setDistributed
// every process is ONE of the N streams (rank r == stream r == the manager's child r): it runs its share
// of the minibatch, then the device gradients and {numerator, Zx, utterances, active, ended} are
// all-reduced over RCCL (scrf_allreduce_grad_ex) and divided by the number of active streams -- the
// reference's join / sum / average (:277-312).
class CRF_Minibatch_GradAccumulator {
 public:
  CRF_Minibatch_GradAccumulator(CRF_Model* myCrf, CRF_FeatureStreamManager* myFtrStrmMgr, QNUInt32 myNStreams);
  CRF_Minibatch_GradAccumulator(CRF_Model* crf, std::vector<CRF_FeatureStream*> streams);   // caller-made streams
  virtual ~CRF_Minibatch_GradAccumulator() {}
  void setMinibatch(QNUInt32 mb);
  void setUttReport(int r) { uttReport = r; }
  void setObjectiveFunction(objfunctype ofunc);
  QNUInt32 getNStreams() { return (QNUInt32)ftrStrms.size(); }
  void rewindAllAndNextSegs();
  virtual double accumulateGradient(double* grad, double* Zx_out, QNUInt32* uttCount, bool* isEndOfIter);
  // the same minibatch, but the summed and averaged gradient STAYS in the engine's device buffer
  // (what CRF_SGTrainer uses: no 2 x lambda_len doubles over PCIe per minibatch)
  double accumulateGradientOnDevice(double* Zx_out, QNUInt32* uttCount, bool* isEndOfIter);
  // Deferred sums (one process, gradient on the device): accumulateGradientOnDevice then returns 0 / *Zx_out = 0 and only
  // QUEUES the copy of {numerator, Zx} behind the minibatch's kernels; the values of the minibatch BEFORE, read in
  // passing, wait in takePreviousSums, and takeCurrentSums waits for the one just issued (end of an iteration).  The
  // trainer prepares minibatch k + 1's batch while k's count kernels run instead of stopping for k's sums first.
  void setDeferSums(bool on) { deferSums = on; }
  bool takePreviousSums(double* numer, double* Zx);
  void takeCurrentSums(double* numer, double* Zx);

 protected:
  double accumulate(double* grad, double* Zx_out, QNUInt32* uttCount, bool* isEndOfIter);
  bool deferSums = false, sumsQueued = false, prevReady = false;
  double prevNumer = 0.0, prevZx = 0.0;
  CRF_Model* crf;
  std::vector<CRF_FeatureStream*> ftrStrms;
  std::vector<QN_SegID> segids;
  QNUInt32 minibatch = CRF_UINT32_MAX;
  int uttReport = 0;
};

// trainers/CRF_Trainer.{h,cpp}
class CRF_Trainer {
 public:
  CRF_Trainer(CRF_Model* crf_in, CRF_FeatureStreamManager* ftr_str_mgr, char* wt_fname);
  virtual ~CRF_Trainer() {}
  virtual void train();
  virtual void setMaxIters(int n) { maxIters = n; }
  virtual void setLR(float v) { lr = v; }
  virtual void setUttRpt(QNUInt32 r) { uttRpt = r; }
  virtual void setLogSpace(int v) { useLogspace = v; }
  virtual void setGaussVar(float gvar_in) { gvar = gvar_in; useGvar = true; }
  virtual void setLabelMask(bool useMask) { useLabelMask = useMask; }
  virtual void setObjectiveFunction(objfunctype ofunc);
  virtual void setLRDecayRate(float v) { lr_decay_rate = v; }
  virtual std::string getWeightDir() { return weight_dir; }
  virtual bool touchDoneFileIter(int iter);
  virtual bool touchDoneFileFinal();

 protected:
  CRF_Model* crf_ptr;
  CRF_FeatureStreamManager* ftr_strm_mgr;
  std::string weight_fname, weight_dir;
  int maxIters = 10;
  float lr = 0.008f, lr_decay_rate = 1.0f;
  QNUInt32 uttRpt = 100;
  int useLogspace = 1;
  float gvar = 0.0f;
  bool useGvar = false, useLabelMask = false;
  objfunctype objective = EXPF;
};

// trainers/CRF_SGTrainer.{h,cpp}: minibatch SGD / AdaGrad, weight averaging, per-iteration files and
// .done.train markers.  Under CRF_Model::setDistributed only rank 0 writes files and progress lines.
class CRF_SGTrainer : public CRF_Trainer {
 public:
  CRF_SGTrainer(CRF_Model* crf_in, CRF_FeatureStreamManager* ftr_str_mgr, char* wt_fname);
  CRF_SGTrainer(CRF_Model* crf, std::vector<CRF_FeatureStream*> streams, const char* weight_fname);   // caller-made streams
  void train() override;
  void setNThreads(int n) { nThreads = n; }
  void setMinibatch(int m) { minibatch = m; }
  void setEta(double e) { eta = e; }
  void setUseAdagrad(double b) { useAdagrad = b != 0.0; }

 protected:
  void sgtrainMinibatch();
  std::vector<CRF_FeatureStream*> own_streams;   // second constructor
  int nThreads = 1, minibatch = 1;
  bool useAdagrad = false;
  double eta = 1.0, eps = 1e-12;
};

// Full-batch accumulation for L-BFGS (trainers/accumulators/CRF_Pthread_GradAccumulator.cpp:130-232; the
// reference's single-thread CRF_GradAccumulator is the same sum): every stream walks its WHOLE view, grad = the plain
// sum over streams (no averaging), the return value is the summed log-likelihood (numerator - Zx), *uttCount the
// utterances.  One process: the streams run one after the other on the model's GPU in device batches;
// CRF_Model::setDistributed: rank r walks stream r and the sums are all-reduced.
class CRF_GradAccumulator {
 public:
  CRF_GradAccumulator(CRF_Model* myCrf, bool myLogspace, int myNStates) : crf(myCrf) { (void)myLogspace; (void)myNStates; }
  virtual ~CRF_GradAccumulator() {}
  virtual double accumulateGradient(CRF_FeatureStreamManager* ftr_str_mgr, int nStreams, double* grad, QNUInt32* uttCount);
  void setUttReport(int u) { uttReport = u; }
  void setObjectiveFunction(objfunctype ofunc);
  void setDeviceBatch(QNUInt32 n) { deviceBatch = n ? n : 1; }   // utterances per scrf_fb_batch call

 protected:
  CRF_Model* crf;
  int uttReport = 0;
  QNUInt32 deviceBatch = 256;
};
typedef CRF_GradAccumulator CRF_Pthread_GradAccumulator;

// trainers/CRF_LBFGSTrainer.{h,cpp}: L-BFGS over the full-batch gradient.  As there: the optimiser runs with the
// library's DEFAULT parameters (the reference fills a parameter struct and then passes NULL, :55-62), the objective is
// -(summed log-likelihood) with the optional Gaussian prior lambda^2 / (2 gvar) (:150-160, a real prior here, unlike
// the SG trainer's), every evaluation after the first writes <out>.i<k>.out, progress stops at crf_epochs evaluations,
// the final weights go to <out> when the optimiser returns 0.  The optimiser itself is host/lbfgs.h.
class CRF_LBFGSTrainer : public CRF_Trainer {
 public:
  CRF_LBFGSTrainer(CRF_Model* crf_in, CRF_FeatureStreamManager* ftr_str_mgr, char* wt_fname);
  void train() override;
  int lastStatus() const { return status; }

 protected:
  int iCounter = 0, status = 0;
};

// io/CRF_MLFManager.{h,cpp}: the transcripts of an HTK master label file as linear acceptors (`crf_align_mlffile`).
// As there: `#!MLF!#` first; an entry starts with a quoted name whose key is the text between the last '/' and the last
// '.' of the WHOLE line (quotes included when the name has no '/'); every other non-empty line up to the single '.' is
// one symbol, looked up as a whole in the symbol table (-1 when it is not there); getFst(name) looks the key of `name`
// up and returns start -> ... -> final with one arc `id:id / 0` per symbol.  The symbol table here is name -> id.
class CRF_MLFManager {
 public:
  CRF_MLFManager(const char* mlffile, const char* olist, const std::map<std::string, long>* symTab);
  virtual ~CRF_MLFManager() {}
  void readMLF(const char* mlffile);
  void getFst(const std::string& fname, crf_amd::ArcListFst* fst);

 private:
  std::string getKey(const std::string& fname);
  const std::map<std::string, long>* symTab;
  std::vector<std::vector<int> > transcripts;
  std::map<std::string, int> fnameTable;
};

// Read-only view of the per-frame DP nodes of ONE utterance (nodes/CRF_StateNode.h:67-115), backed by the
// engine's parity hooks (scrf_scores, scrf_forward_backward): the values a node of the reference holds
// after the gradbuilder's forward and backward sweeps.  The compute* virtuals of the reference are steps of
// a per-node recursion; here the whole utterance is evaluated on the GPU when the vector is loaded, so they
// are no-ops kept for source compatibility.  Custom CRF_StateNode / CRF_FeatureMap subclasses cannot plug
// into the device recursion.
class CRF_StateVector;
class CRF_StateNode {
 public:
  virtual ~CRF_StateNode() {}
  virtual double computeTransMatrix() { return 0.0; }
  virtual double computeAlpha() { return 0.0; }
  virtual double computeFirstAlpha() { return 0.0; }
  virtual double computeBeta(double scale = 1.0) { (void)scale; return 0.0; }
  virtual void setTailBeta() {}
  virtual double computeAlphaSum();                     // Zx on the utterance's last node
  virtual double* getAlpha() { return alpha; }          // [nActualLabs]
  virtual double* getBeta() { return beta; }
  virtual double* getAlphaWithDur() { return alpha_dur; }   // [nodeMaxDur][nActualLabs]: row d-1 = duration d
  virtual double getStateValue(QNUInt32 lab, QNUInt32 dur = 1);
  virtual double getTransValue(QNUInt32 prev_lab, QNUInt32 cur_lab);
  virtual double getFullTransValue(QNUInt32 prev_lab, QNUInt32 cur_lab, QNUInt32 dur = 1);
  virtual QNUInt32 getLabel() { return label; }
  virtual QNUInt32 getNodeMaxDur() { return nodeMaxDur; }
  virtual QNUInt32 getNumAvailLabs() { return nLabs; }

 protected:
  friend class CRF_StateVector;
  double *alpha = nullptr, *beta = nullptr, *alpha_dur = nullptr, *S = nullptr, *M = nullptr;
  QNUInt32 nLabs = 0, nodeMaxDur = 0, label = CRF_LAB_BAD;
  double zx = 0.0;
  bool last = false;
};
class CRF_StateVector {
 public:
  // evaluates the stream's CURRENT utterance under the model's lambda (EXACT scores, log-domain recursion)
  CRF_StateVector(CRF_FeatureStream* ftr_strm, CRF_Model* crf);
  size_t getNodeCount() { return nodes.size(); }
  CRF_StateNode* at(size_t t) { return &nodes.at(t); }
  double getZx() { return zx; }

 private:
  std::vector<CRF_StateNode> nodes;
  std::vector<double> S, M, AD, AL, BE;
  double zx = 0.0;
};

// lattice builders: same call as the reference's templates; the arcs come from the engine in
// AddArc order and are replayed into the caller's FST object
class CRF_LatticeBuilder {
 public:
  CRF_LatticeBuilder(CRF_FeatureStream* ftr_strm_in, CRF_Model* crf_in) : ftr_strm(ftr_strm_in), crf(crf_in) {}
  virtual ~CRF_LatticeBuilder() {}
  template <class Fst>
  int buildLattice(Fst* fst, bool align = false, Fst* alignFst = nullptr, bool norm = true) {
    std::vector<scrf_arc> arcs;
    uint32_t n_states = 0;
    int32_t fin = -1;
    std::vector<QNUInt32> node_labels;
    int seq_len = latticeArcs(norm, &arcs, &n_states, &fin, align ? &node_labels : nullptr);
    if (align) {
      // the label acceptor of decoders/CRF_LatticeBuilder.h:172-184 (every builder has the same block): one state per
      // run of equal node labels, entered by lab:lab / 0 and carrying a self loop lab:lab / 0, lab = node label + 1
      // (an unlabelled node is CRF_LAB_BAD + 1 == 0 there: an epsilon run)
      if (!alignFst) throw std::runtime_error("buildLattice: align mode needs the label FST");
      if (node_labels.empty()) throw std::runtime_error("CRF_LatticeBuilder::buildLattice() caught exception: The label stream is found NULL or the label width is found 0 under the align mode.");
      typedef typename Fst::Arc LArc;
      int cur = alignFst->AddState();
      alignFst->SetStart(cur);
      bool first = true;
      QNUInt32 cur_lab = 0;
      for (QNUInt32 nl : node_labels) {
        const QNUInt32 lab = nl + 1;   // unsigned wrap for CRF_LAB_BAD, as there
        if (first || lab != cur_lab) {
          const int prev = cur;
          cur = alignFst->AddState();
          alignFst->AddArc(prev, LArc((int)lab, (int)lab, 0, cur));
          alignFst->AddArc(cur, LArc((int)lab, (int)lab, 0, cur));
          first = false;
          cur_lab = lab;
        }
      }
      alignFst->SetFinal(cur, 0);
    }
    for (uint32_t s = 0; s < n_states; s++) fst->AddState();
    fst->SetStart(0);
    typedef typename Fst::Arc Arc;
    for (const scrf_arc& a : arcs) fst->AddArc(a.src, Arc(a.ilabel, a.olabel, a.w, a.dst));
    if (fin >= 0) fst->SetFinal(fin, 0);
    return seq_len;
  }

 protected:
  int latticeArcs(bool norm, std::vector<scrf_arc>* arcs, uint32_t* n_states, int32_t* final_state, std::vector<QNUInt32>* node_labels = nullptr);
  CRF_FeatureStream* ftr_strm;
  CRF_Model* crf;
};
typedef CRF_LatticeBuilder CRF_LatticeBuilder_StdSeg_WithoutDurLab_WithoutSegTransFtr;

// Viterbi decoder of CRFDecode (decoders/CRF_ViterbiDecoder_StdSeg_NoSegTransFtr.{h,cpp}; used for
// STDFRAME and STDSEG_NO_DUR_NO_SEGTRANSFTR models, CRFDecode/src/Main.cpp:1059-1077) for the case
// the reference builds its own free-phone-loop LM (lm_fst == NULL, createFreePhoneLmFst :1270-1350).
// That LM forbids the same phone twice in a row at the LM level, but the decoder lets a phone
// continue over several segments through its "internal" transition (l -> l, :246-300), so the
// hypothesis space is every segmentation x labelling and the weights are the float sums
// (old + float(-M)) + float(-S) (:286-290, :143-146): the same search as the dense device
// Viterbi over the lattice.  The per-node hypothesis vectors (viterbiPhnIds / viterbiPointers /
// viterbiDurs / isPhoneStartBoundary, nodes/CRF_StateNode.h:60-65) exist here in their dense form
// on the device (best weight, back pointer and duration per frame x label) and bestSegments()
// is the backtrace (:2204-2290).  The search is exhaustive: a positive beam cannot lose the best
// path here, whereas the reference's pruned search can.  With an LM (lm_fst != NULL): decodeLm below; with
// several states per label: decodeNState.
class CRF_ViterbiDecoder_StdSeg_NoSegTransFtr {
 public:
  CRF_ViterbiDecoder_StdSeg_NoSegTransFtr(CRF_FeatureStream* ftr_strm_in, CRF_Model* crf_in) : ftr_strm(ftr_strm_in), crf(crf_in) {}
  virtual ~CRF_ViterbiDecoder_StdSeg_NoSegTransFtr() {}
  void setIfOutputFullFst(bool ifFull) { if_output_full_fst = ifFull; }
  // one segment of the best path, first to last
  struct Segment {
    uint32_t phone, dur, start;  // frames start .. start+dur-1
    float weight;                // float(-getStateValue) for the first, float(-getFullTransValue) after (:2237,:2262)
    bool phone_start;            // isPhoneStartBoundary: the previous segment carries another phone
  };
  // result_fst: the best path as a chain, one arc per segment, StdArc(phone+1, phone+1 if the phone
  // starts here else 0, weight, next), final weight Zx (:1967-1971, :2204-2290, after Compose with
  // the free-phone LM :2294).  Returns the number of frames.
  template <class Fst>
  int nStateDecode(Fst* result_fst, Fst* lm_fst, Fst* out_full_fst, double input_beam, unsigned min_hyps = 0, unsigned max_hyps = 0, float beam_inc = 0.05f) {
    (void)min_hyps; (void)max_hyps; (void)beam_inc;
    if (if_output_full_fst) {
      if (out_full_fst == nullptr) throw std::runtime_error("CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::nStateDecode: setIfOutputFullFst(true) needs an out_full_fst to fill");
      if (crf->getFeatureMap() && crf->getFeatureMap()->getNumStates() > 1)
        throw std::runtime_error("CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::nStateDecode: the full output lattice is built for crf_states = 1 only");
      return decodeFull(lm_fst, input_beam, result_fst, out_full_fst);
    }
    if (crf->getFeatureMap() && crf->getFeatureMap()->getNumStates() > 1) return decodeNState(lm_fst, input_beam, result_fst);
    if (lm_fst != nullptr) return decodeLm(*lm_fst, input_beam, result_fst);
    const int T = decode();
    typedef typename Fst::Arc Arc;
    int cur = result_fst->AddState();
    result_fst->SetStart(cur);
    if (segs.empty()) {  // "Could not reach end of utterance" (:2141-2147)
      int fin = result_fst->AddState();
      result_fst->AddArc(cur, Arc(0, 0, 8, fin));
      result_fst->SetFinal(fin, (float)zx);
      return T;
    }
    for (const Segment& g : segs) {
      int nxt = result_fst->AddState();
      result_fst->AddArc(cur, Arc((int)g.phone + 1, g.phone_start ? (int)g.phone + 1 : 0, g.weight, nxt));
      cur = nxt;
    }
    result_fst->SetFinal(cur, (float)zx);
    return T;
  }
  const std::vector<Segment>& bestSegments() const { return segs; }
  double getZx() const { return zx; }
  float getBestWeight() const { return best_weight; }  // total weight of the best hypothesis (:2133)

  // LM-constrained decode (lm_fst != NULL), host-side search over the device's node scores.  Same
  // hypothesis space and float arithmetic as the reference's time-synchronous search (:246-735):
  // a hypothesis is (LM state, phone); a phone continues over segments through its internal transition
  // (old + float(-M[t][l][l]), no LM move); a new phone follows an LM arc with ilabel phone+1 after any
  // epsilon-input arcs ((old + LM weights) + float(-M[t][p][l]), :452-456); float(-S[t][d][l]) is added
  // at the segment's end (:143-146); strict-improvement updates.  `beam` > 0 keeps, like pruning()
  // (:976-1040), the hypotheses of a frame whose weight is < the frame's minimum + beam (min_hyps /
  // max_hyps / beam_inc are accepted and unused there as well); <= 0 searches exhaustively.  Result: one arc per LM epsilon arc that
  // carries a word (0 : olabel, LM weight), one per segment (phone+1 : LM arc's olabel where the phone
  // starts else 0, float(-(M+S)) from the END node + the LM arc's weight); final weight Zx + the LM's
  // final weight.  OpenFST itself is not in the tree: parity with Compose/ShortestPath ordering UNPINNED.
  template <class Fst> int decodeLm(const Fst&, double, Fst*) {
    throw std::runtime_error("nStateDecode: an LM FST must be a crf_amd::ArcListFst");
  }
  int decodeLm(const crf_amd::ArcListFst& lm, double beam, crf_amd::ArcListFst* result_fst, crf_amd::ArcListFst* out_full_fst = nullptr);
  // setIfOutputFullFst(true): the search lattice next to the best path (output_full_fst, :1163-1215 insertArcToOutputFullFst,
  // :168-222 stateValueUpdate_onOutputFullFst, :1974-1990 final states).  States: one start state (the reference's time -1)
  // and one per hypothesis (end frame, LM state, phone) that a segment reaches -- the reference keys its states by (end
  // frame, LM state) alone; with its own free phone loop and with LMs whose states are entered on one phone the two are
  // the same thing.  Arcs: one per transition the search expands (a kept hypothesis x [its internal continuation | an LM
  // arc behind the epsilon closure] x duration), StdArc(phone + 1, word if the LM moved else 0, (LM weights + float(-M)) +
  // float(-S of the segment), target); every hypothesis kept at the last frame is final with weight Zx (LM final weights
  // are NOT in it, as there); then Connect (states off every successful path dropped).  Without an LM the search runs
  // against the free phone loop of createFreePhoneLmFst (:1270-1350) built here.  crf_states = 1 only.
  template <class Fst> int decodeFull(const Fst*, double, Fst*, Fst*) {
    throw std::runtime_error("nStateDecode: the full output lattice needs crf_amd::ArcListFst machines");
  }
  int decodeFull(const crf_amd::ArcListFst* lm, double beam, crf_amd::ArcListFst* result_fst, crf_amd::ArcListFst* out_full_fst);
  // crf_states = K > 1 (CRF_ViterbiDecoder::nStateDecode, decoders/CRF_ViterbiDecoder.cpp:398-960, and the n-state
  // branches of this class's :246-735): a phone is its K states in order, each held for one or more frames (segments);
  // a hypothesis enters a phone at its start state along an LM arc with ilabel phone+1, leaves it from its end state,
  // an utterance starts at a start state and ends at an end state (:452-453, :899-906).  Searched here WITHOUT pruning,
  // as the shortest path of [the n-state lattice of the utterance, with the phone's LM token on every arc that enters a
  // start state from outside its phone and the final arcs of the end states only] composed with the LM (or the free
  // phone loop when lm_fst == NULL): the reference's beam search returns the same path whenever its beam does not
  // prune it (ties: unpinned).  Result: one arc per lattice arc that carries a label and per LM word, StdArc(state
  // label + 1 [+ nLabs*(dur-1) for a segmental model], word where a phone starts else 0, weight, next); final weight
  // Zx + the LM's final weight.
  // `beam` > 0: the reference's time-synchronous pruning over the n-state lattice's nodes (kept iff below the node's
  // minimum + beam, pruning() :976-1060); <= 0: exhaustive.
  template <class Fst> int decodeNState(const Fst*, double, Fst*) {
    throw std::runtime_error("nStateDecode: with crf_states > 1 the result and LM FSTs must be crf_amd::ArcListFst");
  }
  int decodeNState(const crf_amd::ArcListFst* lm, double beam, crf_amd::ArcListFst* result_fst);
  size_t lastNumHyps() const { return n_hyps; }   // hypotheses kept, summed over frames (beam diagnostics)

 protected:
  int decode();
  size_t n_hyps = 0;
  CRF_FeatureStream* ftr_strm;
  CRF_Model* crf;
  bool if_output_full_fst = false;
  std::vector<Segment> segs;
  double zx = 0.0;
  float best_weight = 0.0f;
};

// best path of CRFFstDecode (ShortestPath -> Project(OUTPUT) -> RmEpsilon -> TopSort -> olabel-1)
std::vector<uint32_t> crf_amd_best_path(CRF_FeatureStream* ftr_strm, CRF_Model* crf, float* cost);
// the same for the stream's CURRENT utterance and up to max_utts - 1 following ones in one device
// batch (the stream is advanced with nextseg(); *at_end is set when it ran out).  Labels and costs
// are what crf_amd_best_path returns utterance by utterance.
size_t crf_amd_best_paths(CRF_FeatureStream* ftr_strm, CRF_Model* crf, size_t max_utts,
                          std::vector<std::vector<uint32_t> >* labels, std::vector<float>* costs, bool* at_end);

// the split arithmetic of the data-parallel layer (exported with C linkage for the CPU tests): share of
// stream s of a minibatch (CRF_Minibatch_GradAccumulator.cpp:229-241,257), contiguous utterance view of
// stream s (io/CRF_FeatureStreamManager.cpp:425-464)
extern "C" uint32_t crf_amd_minibatch_share(uint32_t minibatch, uint32_t n_streams, uint32_t s);
extern "C" void crf_amd_view_range(uint32_t n_utts, uint32_t n_streams, uint32_t s, uint32_t* lo, uint32_t* hi);

#endif  // CRF_AMD_H_
