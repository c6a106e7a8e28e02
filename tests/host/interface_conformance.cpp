// interface_conformance.cpp -- does asr-craft_amd/host/crf_amd.h offer the interfaces a caller of the reference
// uses on the training path?  Two parts, both written from the signatures crf_amd.h declares (SURVEY.md 8b lists where
// each one comes from in the reference):
//   1. compile-time: static_asserts on constructor argument lists and member-function pointer types;
//   2. run-time: a small caller of our own (not the reference's main) that trains the bundled fixture through the
//      classes and prints the node view of utterance 0.  tests/test_gpu_cli.py compares the weight files it writes
//      with bin/CRFTrain's for the same settings, byte for byte, and the node values with the oracle.
//
//   interface_conformance <golden dir> <out_weight_file> <threads> <bunch> <epochs>
#include <unistd.h>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>
#include <type_traits>

#include "crf_amd.h"

// ---- 1. the shapes of the interfaces -----------------------------------------------------------------------
template <class Sig, Sig> struct has_member {};   // instantiating it checks name + exact type
#define MEMBER(cls, name, ...) static_assert(sizeof(has_member<__VA_ARGS__, &cls::name>) > 0, #cls "::" #name)

// stream factory: the 27-argument constructor (io/CRF_FeatureStreamManager.h), join, children, the two streams
static_assert(std::is_constructible<CRF_FeatureStreamManager, int, const char*, char*, const char*, char*, size_t, size_t,
                                    size_t, size_t, size_t, size_t, size_t, size_t, size_t, bool, bool, int, int, char*,
                                    char*, FILE*, int, double, double, seqtype, QNUInt32, size_t>::value,
              "CRF_FeatureStreamManager(debug, name, file, format, hardtarget, ..., seqtype, seed, threads)");
static_assert(std::is_constructible<CRF_FeatureStreamManager, int, const char*, char*, const char*, char*, size_t, size_t,
                                    size_t, size_t, size_t, size_t, size_t, size_t, size_t, bool, bool, int, int, char*,
                                    char*, FILE*, int, double, double, seqtype>::value,
              "seed and thread count are optional");
MEMBER(CRF_FeatureStreamManager, join, void (CRF_FeatureStreamManager::*)(CRF_FeatureStreamManager*));
MEMBER(CRF_FeatureStreamManager, getNumFtrs, size_t (CRF_FeatureStreamManager::*)());
MEMBER(CRF_FeatureStreamManager, getChild, CRF_FeatureStreamManager* (CRF_FeatureStreamManager::*)(size_t));
static_assert(std::is_same<decltype(CRF_FeatureStreamManager::trn_stream), CRF_FeatureStream*>::value, "trn_stream");
static_assert(std::is_same<decltype(CRF_FeatureStreamManager::cv_stream), CRF_FeatureStream*>::value, "cv_stream");
// the stream protocol the hot path consumes (io/CRF_FeatureStream.h:54-66)
MEMBER(CRF_FeatureStream, nextseg, QN_SegID (CRF_FeatureStream::*)());
MEMBER(CRF_FeatureStream, read, size_t (CRF_FeatureStream::*)(size_t, float*, QNUInt32*));
MEMBER(CRF_FeatureStream, rewind, int (CRF_FeatureStream::*)());
MEMBER(CRF_FeatureStream, num_ftrs, size_t (CRF_FeatureStream::*)());
// model + feature map (CRF_Model.h, ftrmaps/CRF_FeatureMap.h:61-96)
static_assert(std::is_constructible<CRF_Model, QNUInt32>::value, "CRF_Model(nlabs)");
MEMBER(CRF_Model, setFeatureMap, void (CRF_Model::*)(CRF_FeatureMap*));
MEMBER(CRF_Model, getLambda, double* (CRF_Model::*)());
MEMBER(CRF_Model, getLambdaLen, QNUInt32 (CRF_Model::*)());
MEMBER(CRF_Model, setLabMaxDur, void (CRF_Model::*)(QNUInt32));
MEMBER(CRF_Model, setNActualLabs, void (CRF_Model::*)(QNUInt32));
MEMBER(CRF_Model, setModelType, void (CRF_Model::*)(modeltype));
MEMBER(CRF_Model, readFromFile, bool (CRF_Model::*)(const char*));
static_assert(std::is_same<decltype(&CRF_FeatureMap::createFeatureMap), CRF_FeatureMap* (*)(CRF_FeatureMap_config*)>::value,
              "static factory createFeatureMap(config)");
MEMBER(CRF_FeatureMap, getStateFeatureIdx, QNUInt32 (CRF_FeatureMap::*)(QNUInt32, QNUInt32));
MEMBER(CRF_FeatureMap, getTransFeatureIdx, QNUInt32 (CRF_FeatureMap::*)(QNUInt32, QNUInt32, QNUInt32));
MEMBER(CRF_FeatureMap, getNumFtrFuncs, QNUInt32 (CRF_FeatureMap::*)());
MEMBER(CRF_FeatureMap, recalc, QNUInt32 (CRF_FeatureMap::*)());
// per-utterance and per-minibatch operations (SURVEY 8b iii, iv)
static_assert(std::is_same<decltype(&CRF_GradBuilder::create), CRF_GradBuilder* (*)(CRF_Model*, objfunctype)>::value, "create");
MEMBER(CRF_GradBuilder, buildGradient, double (CRF_GradBuilder::*)(CRF_FeatureStream*, double*, double*));
static_assert(std::is_constructible<CRF_Minibatch_GradAccumulator, CRF_Model*, CRF_FeatureStreamManager*, QNUInt32>::value, "accumulator");
MEMBER(CRF_Minibatch_GradAccumulator, accumulateGradient, double (CRF_Minibatch_GradAccumulator::*)(double*, double*, QNUInt32*, bool*));
// trainers (trainers/CRF_Trainer.h, CRF_SGTrainer.h, CRF_LBFGSTrainer.h)
static_assert(std::is_constructible<CRF_SGTrainer, CRF_Model*, CRF_FeatureStreamManager*, char*>::value, "CRF_SGTrainer(crf, mgr, file)");
static_assert(std::is_constructible<CRF_LBFGSTrainer, CRF_Model*, CRF_FeatureStreamManager*, char*>::value, "CRF_LBFGSTrainer(crf, mgr, file)");
static_assert(std::is_base_of<CRF_Trainer, CRF_SGTrainer>::value && std::is_base_of<CRF_Trainer, CRF_LBFGSTrainer>::value, "trainer hierarchy");
MEMBER(CRF_Trainer, train, void (CRF_Trainer::*)());
MEMBER(CRF_Trainer, setMaxIters, void (CRF_Trainer::*)(int));
MEMBER(CRF_Trainer, setLR, void (CRF_Trainer::*)(float));
MEMBER(CRF_Trainer, setLRDecayRate, void (CRF_Trainer::*)(float));
MEMBER(CRF_Trainer, setUttRpt, void (CRF_Trainer::*)(QNUInt32));
MEMBER(CRF_Trainer, setGaussVar, void (CRF_Trainer::*)(float));
MEMBER(CRF_Trainer, setObjectiveFunction, void (CRF_Trainer::*)(objfunctype));
MEMBER(CRF_Trainer, getWeightDir, std::string (CRF_Trainer::*)());
MEMBER(CRF_SGTrainer, setNThreads, void (CRF_SGTrainer::*)(int));
MEMBER(CRF_SGTrainer, setMinibatch, void (CRF_SGTrainer::*)(int));
MEMBER(CRF_SGTrainer, setEta, void (CRF_SGTrainer::*)(double));
MEMBER(CRF_SGTrainer, setUseAdagrad, void (CRF_SGTrainer::*)(double));
// node view (nodes/CRF_StateNode.h:67-115)
MEMBER(CRF_StateNode, computeAlphaSum, double (CRF_StateNode::*)());
MEMBER(CRF_StateNode, computeBeta, double (CRF_StateNode::*)(double));
MEMBER(CRF_StateNode, getAlpha, double* (CRF_StateNode::*)());
MEMBER(CRF_StateNode, getBeta, double* (CRF_StateNode::*)());
MEMBER(CRF_StateNode, getStateValue, double (CRF_StateNode::*)(QNUInt32, QNUInt32));
MEMBER(CRF_StateNode, getTransValue, double (CRF_StateNode::*)(QNUInt32, QNUInt32));
MEMBER(CRF_StateNode, getFullTransValue, double (CRF_StateNode::*)(QNUInt32, QNUInt32, QNUInt32));

// ---- 2. a caller -------------------------------------------------------------------------------------------
namespace {

struct Run {
  std::string dir, out;
  int threads = 1, bunch = 1, epochs = 2;
};

// one ascii feature file of the bundled fixture as a stream manager: no context, no deltas, window of one frame
std::unique_ptr<CRF_FeatureStreamManager> open_features(const Run& r, const char* tag, const std::string& file) {
  std::string path = r.dir + "/" + file, labels = r.dir + "/crftrain_test.lab.ascii", everything = "all";
  return std::unique_ptr<CRF_FeatureStreamManager>(new CRF_FeatureStreamManager(
      /*debug*/ 1, tag, &path[0], "ascii", &labels[0], /*ht_offset*/ 0, /*width, first, count*/ 0, 0, 0,
      /*window extent, offset, length*/ 1, 0, 1, /*context*/ 0, 0, /*segment ftrs, boundary deltas*/ false, false,
      /*delta order, window*/ 0, 9, &everything[0], /*cv range*/ nullptr, /*norm*/ nullptr, 0, 0.0, 0.0, SEQUENTIAL,
      /*seed*/ 0, (size_t)r.threads));
}

CRF_FeatureMap_config g_map;   // the map keeps a pointer to its configuration

void describe_frame_model(CRF_Model& crf, QNUInt32 n_features) {
  crf.setLabMaxDur(1);
  crf.setNActualLabs(crf.getNLabs());
  crf.setModelType(STDFRAME);
  g_map.map_type = STDSTATE;                       // state features over every input, bias on both kinds
  g_map.numLabs = crf.getNLabs();
  g_map.numFeas = n_features;
  g_map.numStates = 1;
  g_map.useStateFtrs = true;  g_map.stateFidxStart = 0;  g_map.stateFidxEnd = n_features - 1;
  g_map.useTransFtrs = false;
  g_map.useStateBias = g_map.useTransBias = true;
  g_map.stateBiasVal = g_map.transBiasVal = 1.0;
  g_map.maxDur = 1;
  g_map.nActualLabs = crf.getNLabs();
  crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&g_map));
}

void print_node_view(CRF_FeatureStreamManager& mgr, CRF_Model& crf) {
  mgr.trn_stream->rewind();
  mgr.trn_stream->nextseg();
  CRF_StateVector nodes(mgr.trn_stream, &crf);
  const size_t n = nodes.getNodeCount();
  std::cout.precision(17);
  std::cout << "NODES " << n << " ZX " << nodes.at(n - 1)->computeAlphaSum() << std::endl;
  for (size_t t = 0; t < n; t++) {
    CRF_StateNode* nd = nodes.at(t);
    std::cout << "NODE " << t << " label " << nd->getLabel() << " state0 " << nd->getStateValue(0, 1) << " state3 "
              << nd->getStateValue(3, 1) << " trans12 " << nd->getTransValue(1, 2) << " full12 " << nd->getFullTransValue(1, 2, 1)
              << " alpha2 " << nd->getAlpha()[2] << " beta2 " << nd->getBeta()[2] << std::endl;
  }
}

int run(const Run& r) {
  auto first = open_features(r, "ftr1_file", "crftrain_test.ascii");
  auto second = open_features(r, "ftr2_file", "crftrain_test.ftr2.ascii");
  first->join(second.get());                       // feature concatenation: 3 + 3 inputs per frame
  CRF_Model crf(48);
  describe_frame_model(crf, (QNUInt32)first->getNumFtrs());
  std::cout << "labels " << crf.getNLabs() << ", weights " << crf.getLambdaLen() << std::endl;

  std::string out = r.out;
  std::unique_ptr<CRF_SGTrainer> sg(new CRF_SGTrainer(&crf, first.get(), &out[0]));
  sg->setObjectiveFunction(EXPF);
  sg->setUseAdagrad(0);
  sg->setEta(1.0);
  sg->setNThreads(r.threads);
  sg->setMinibatch(r.bunch);
  CRF_Trainer* trainer = sg.get();                 // the rest goes through the base interface
  trainer->setMaxIters(r.epochs);
  trainer->setLR(0.1f);
  trainer->setLRDecayRate(1.0f);
  trainer->setUttRpt(1);
  if (access((trainer->getWeightDir() + "/.done.train").c_str(), F_OK) == 0) {
    std::cout << "already trained" << std::endl;
    return 0;
  }
  trainer->train();
  print_node_view(*first, crf);
  return 0;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 6) {
    std::cerr << "usage: interface_conformance <golden dir> <out_weight_file> <threads> <bunch> <epochs>" << std::endl;
    return 2;
  }
  Run r;
  r.dir = argv[1]; r.out = argv[2];
  r.threads = atoi(argv[3]); r.bunch = atoi(argv[4]); r.epochs = atoi(argv[5]);
  try {
    return run(r);
  } catch (const std::exception& e) {   // the reference's mains print the exception and exit non-zero
    std::cerr << "Exception: " << e.what() << std::endl;
    return 255;
  }
}
