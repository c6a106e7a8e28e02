// CRFTrain -- training front-end with the reference's `name=value` surface (CRFTrain/src/Main.cpp)
// on the MI355X engine: SGD / AdaGrad over minibatches, `threads` data-parallel streams with
// contiguous utterance views, per-iteration checkpoints and .done.train markers.
#include "cli_common.h"

int main(int argc, char** argv) {
  Args a(argc, argv);
  try {
    CliModel m;
    auto data = load_streams(a, &m);
    if (!a.has("hardtarget_file")) { std::cerr << "hardtarget_file is required" << std::endl; return 1; }
    if (!a.has("out_weight_file")) { std::cerr << "out_weight_file is required" << std::endl; return 1; }
    auto labs = read_labs(a.str("hardtarget_file"));
    const std::vector<uint32_t> sents = select_sents(a, "train_sent_range", data[0].size());
    if (a.has("cv_sent_range") && a.str("cv_sent_range") != "nil" && a.str("cv_sent_range") != "none")
      std::cout << "NOTE: cv_sent_range=" << a.str("cv_sent_range") << " ignored: the cross-validation pass is not built" << std::endl;
    if (a.str("crf_train_method", "sg") != "sg") { std::cerr << "only crf_train_method=sg is built" << std::endl; return 1; }
    // presentation order: seq (default HERE; the reference defaults to random) | random | noreplace
    const std::string order = a.str("crf_train_order", "seq");
    if (order != "seq" && order != "random" && order != "noreplace") { std::cerr << "crf_train_order=" << order << " (seq|random|noreplace)" << std::endl; return 1; }
    if (order != "seq") std::cout << "NOTE: crf_train_order=" << order << ": orders come from std::mt19937_64 seeded like the reference (12345*epoch+seed); QuickNet's generator is not reproducible here, so the sequence differs from the reference's" << std::endl;

    CRF_Model crf(m.L);
    crf.setLabMaxDur(m.D);
    crf.setNActualLabs(m.fmap.nActualLabs);
    crf.setModelType(m.mtype);
    std::cout << "LABELS: " << crf.getNLabs() << std::endl;
    std::cout << "LABEL_MAXIMUM_DURATION: " << crf.getLabMaxDur() << std::endl;
    crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&m.fmap));
    std::cout << "FEATURES: " << crf.getLambdaLen() << std::endl;
    // resume flags, nested as in CRFTrain/src/Main.cpp:599-621: the average and AdaGrad accumulators are
    // only read next to an initial weight file and a positive presentation count
    if (a.has("init_weight_file")) {
      if (!crf.readFromFile(a.str("init_weight_file").c_str())) { std::cerr << "ERROR! File " << a.str("init_weight_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
      if (a.num("avg_weight_present", 0) > 0) {
        if (a.has("avg_weight_file") && !crf.readAverageFromFile(a.str("avg_weight_file").c_str(), (int)a.num("avg_weight_present", 0))) {
          std::cerr << "ERROR! File " << a.str("avg_weight_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
        if (a.num("crf_use_adagrad", 0) != 0 && a.has("grad_sqr_acc_file") && !crf.readGradSqrAccFromFile(a.str("grad_sqr_acc_file").c_str())) {
          std::cerr << "ERROR! File " << a.str("grad_sqr_acc_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
      }
      crf.setInitIter((QNUInt32)a.num("init_iter", 0));
    }
    if (a.real("crf_gauss_var", 0.0) != 0.0) { std::cerr << "crf_gauss_var: the Gaussian prior is not built" << std::endl; return 1; }

    CRF_MemoryFeatureStream all(m.recipes, m.D, m.fmap.nActualLabs);
    const seqtype trn_seq = order == "seq" ? SEQUENTIAL : (order == "noreplace" ? RANDOM_NO_REPLACE : RANDOM_REPLACE);
    const size_t U = sents.size();
    for (size_t i = 0; i < U; i++) {
      const uint32_t u = sents[i];
      std::vector<std::vector<float> > fr(data.size());
      for (size_t s = 0; s < data.size(); s++) { fr[s] = data[s].get(u); data[s].drop(u); }
      all.addUtterance(fr, u < labs.size() ? labs[u] : std::vector<uint32_t>());
    }
    if (trn_seq != SEQUENTIAL) all.setPresentation(trn_seq, (QNUInt32)a.num("crf_random_seed", 0));
    // `threads` child streams over contiguous ranges (io/CRF_FeatureStreamManager.cpp:425-464)
    const size_t N = (size_t)std::max(1L, a.num("threads", 1));
    std::vector<std::unique_ptr<CRF_MemoryFeatureStream> > views;
    std::vector<CRF_FeatureStream*> streams;
    for (size_t s = 0; s < N; s++) {
      const size_t per = U / N, lo = s * per, cnt = s == N - 1 ? U - lo : per;
      views.emplace_back(all.view(lo, cnt));
      streams.push_back(views.back().get());
    }
    CRF_SGTrainer tr(&crf, streams, a.str("out_weight_file").c_str());
    std::cout << "MINIBATCH SIZE: " << a.num("crf_bunch_size", 1) << std::endl;
    std::cout << "NUMBER OF THREADS: " << N << std::endl;
    tr.setMaxIters((int)a.num("crf_epochs", 10));
    tr.setLR((float)a.real("crf_lr", 0.008));
    tr.setLRDecayRate((float)a.real("crf_lr_decay_rate", 1.0));
    tr.setMinibatch((QNUInt32)a.num("crf_bunch_size", 1));
    tr.setUseAdagrad(a.num("crf_use_adagrad", 0) != 0);
    tr.setEta(a.real("crf_adagrad_eta", 1.0));
    tr.setUttRpt((QNUInt32)a.num("crf_utt_rpt", 100));
    {  // a finished run is not repeated (Main.cpp:676-682)
      const std::string wf = a.str("out_weight_file");
      const size_t k = wf.find_last_of('/');
      const std::string done_file = (k == std::string::npos ? std::string(".") : wf.substr(0, k)) + "/.done.train";
      if (std::ifstream(done_file.c_str()).good()) {
        std::cout << "The done file has already existed: " << done_file << std::endl;
        std::cout << "Finished." << std::endl;
        return 0;
      }
    }
    tr.train();
  } catch (std::exception& e) {
    std::cerr << "Exception: " << e.what() << std::endl;
    return -1;
  }
  return 0;
}
