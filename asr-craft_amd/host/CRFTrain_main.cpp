// CRFTrain -- training front-end with the reference's `name=value` surface (CRFTrain/src/Main.cpp) on the
// MI355X engine.  The object sequence is the reference's (Main.cpp:508-684): CRF_FeatureStreamManager per
// input file (+ join), CRF_Model, CRF_FeatureMap::createFeatureMap, resume files, CRF_SGTrainer with its
// setters, the .done.train guard, train().
//
// Data parallelism: `threads=N` are N streams with contiguous utterance views, as in the reference.
//   * one process: the N streams run one after the other on one GPU (crf_device=, default 0);
//   * one process per GPU (launched with RANK / WORLD_SIZE / LOCAL_RANK in the environment, e.g. by
//     `python -m torch.distributed.run --nproc-per-node N .../bin/CRFTrain ...` with --no-python): rank r IS
//     stream r on GPU LOCAL_RANK, `threads` is WORLD_SIZE, and the minibatch gradient is all-reduced over
//     RCCL/xGMI every SGD step (scrf_allreduce_grad_ex) -- the reference's join / sum / average.  Rank 0
//     hands the RCCL unique id to the others through <out_weight_file>.rccl_id and is the only writer of
//     weight files, markers and progress lines.  crf_force_comm=1 initialises the communicator for one rank.
// crf_precision=exact|fast|fastlin|fast32 selects the arithmetic of the training contractions (default fast).
#include "cli_common.h"

int main(int argc, char** argv) {
  // ranks are read before anything touches the GPU
  const int rank = (int)env_num("RANK", 0), world = (int)env_num("WORLD_SIZE", 1), local_rank = (int)env_num("LOCAL_RANK", 0);
  Args a(argc, argv);
  try {
    if (world < 1 || rank < 0 || rank >= world) { std::cerr << "RANK=" << rank << " outside [0, WORLD_SIZE=" << world << ")" << std::endl; return 1; }
    if (!a.has("ftr1_file")) { std::cerr << "ftr1_file is required" << std::endl; return 1; }
    if (!a.has("hardtarget_file")) { std::cerr << "hardtarget_file is required" << std::endl; return 1; }
    if (!a.has("out_weight_file")) { std::cerr << "out_weight_file is required" << std::endl; return 1; }
    const std::string method = a.str("crf_train_method", "sg");
    if (method != "sg" && method != "lbfgs") { std::cerr << "crf_train_method=" << method << " is not built (sg|lbfgs)" << std::endl; return 1; }
    CliModel m;
    m.D = (uint32_t)a.num("label_maximum_duration", 1);
    m.L = (uint32_t)a.num("crf_label_size", 0);
    m.mtype = parse_model_type(a.str("crf_model_type", "stdframe"));
    if (m.L == 0) { std::cerr << "crf_label_size is required" << std::endl; return 1; }
    refuse_unbuilt_flags(a, m.D);
    // presentation order: seq (default HERE; the reference defaults to random) | random | noreplace
    const std::string order = a.str("crf_train_order", "seq");
    if (order != "seq" && order != "random" && order != "noreplace") { std::cerr << "crf_train_order=" << order << " (seq|random|noreplace)" << std::endl; return 1; }
    if (order != "seq" && rank == 0) std::cout << "NOTE: crf_train_order=" << order << ": orders come from std::mt19937_64 seeded like the reference (12345*epoch+seed); QuickNet's generator is not reproducible here, so the sequence differs from the reference's" << std::endl;
    const seqtype trn_seq = order == "seq" ? SEQUENTIAL : (order == "noreplace" ? RANDOM_NO_REPLACE : RANDOM_REPLACE);
    if (a.has("cv_sent_range") && a.str("cv_sent_range") != "nil" && a.str("cv_sent_range") != "none" && rank == 0)
      std::cout << "NOTE: cv_sent_range=" << a.str("cv_sent_range") << ": the CV stream is built, but like the reference's SG trainer (CRF_SGTrainer.cpp:73-441) nothing reads it" << std::endl;

    long threads = std::max(1L, a.num("threads", 1));
    if (world > 1) {
      if (a.has("threads") && threads != 1 && threads != world) { std::cerr << "threads=" << threads << " but WORLD_SIZE=" << world << ": with one process per GPU every rank is one stream" << std::endl; return 1; }
      threads = world;
      crf_amd::setProcessView(rank, world);   // the managers below keep this rank's child view of the training data only
    }

    // ---- Main.cpp:508-537: one manager per input file, joined
    std::vector<std::unique_ptr<CRF_FeatureStreamManager> > strs;
    for (int k = 1; k <= 3; k++) {
      const std::string p = "ftr" + std::to_string(k) + "_";
      if (!a.has(p + "file")) break;
      std::string file = a.str(p + "file"), fmt = a.str(p + "format", "pfile"), ht = a.str("hardtarget_file");
      std::string trn = a.str("train_sent_range", "all"), cv = a.str("cv_sent_range", "");
      strs.emplace_back(new CRF_FeatureStreamManager(
          1, (p + "file").c_str(), &file[0], fmt.c_str(), &ht[0], (size_t)a.num("hardtarget_window_offset", 0),
          (size_t)a.num(p + "width", 0), (size_t)a.num(p + "ftr_start", 0), (size_t)a.num(p + "ftr_count", 0),
          (size_t)m.D, (size_t)a.num(p + "window_offset", 0), (size_t)m.D, (size_t)a.num(p + "left_context_len", 0),
          (size_t)a.num(p + "right_context_len", 0), a.num(p + "extract_seg_ftr", 0) != 0,
          a.num(p + "use_boundary_delta_ftr", 0) != 0, (int)a.num(p + "delta_order", 0), (int)a.num(p + "delta_win", 0),
          &trn[0], cv.empty() ? nullptr : &cv[0], nullptr, 0, 0, 0, trn_seq, (QNUInt32)a.num("crf_random_seed", 0), (size_t)threads));
      if (k > 1) strs[0]->join(strs.back().get());
    }
    CRF_FeatureStreamManager& str1 = *strs[0];

    // ---- Main.cpp:539-597: the model
    CRF_Model my_crf(m.L);
    if (rank == 0) std::cout << "LABELS: " << my_crf.getNLabs() << std::endl;
    my_crf.setLabMaxDur(m.D);
    m.F = (uint32_t)str1.getNumFtrs();
    set_fmap_config(a, &m);
    my_crf.setNActualLabs(m.fmap.nActualLabs);
    if (rank == 0) std::cout << "LABEL_MAXIMUM_DURATION: " << my_crf.getLabMaxDur() << std::endl;
    my_crf.setModelType(m.mtype);
    my_crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&m.fmap));
    if (rank == 0) std::cout << "FEATURES: " << my_crf.getLambdaLen() << std::endl;
    my_crf.setDevice((int)a.num("crf_device", world > 1 ? local_rank : 0));
    my_crf.setTrainPrecision(parse_precision(a));
    my_crf.setTrainingOnly(true);   // this process never builds lattices: the n-state frame model may take the dense kernels
    if (rank == 0 && m.mtype == STDFRAME && m.fmap.numStates > 1 && parse_precision(a) != SCRF_PREC_EXACT)
      std::cout << "NOTE: crf_states=" << m.fmap.numStates << " on the frame model trains through the dense kernels (as the n-state segmental model with maximum duration 1: the same function and weight layout); crf_precision=exact keeps the reference-order n-state kernels" << std::endl;
    if (world > 1 || a.num("crf_force_comm", 0) != 0) my_crf.setDistributed(rank, world, a.str("out_weight_file") + ".rccl_id");

    // ---- Main.cpp:599-631: resume flags, nested as there -- the average and AdaGrad accumulators are only
    // read next to an initial weight file and a positive presentation count
    if (a.has("init_weight_file")) {
      if (!my_crf.readFromFile(a.str("init_weight_file").c_str())) { std::cerr << "ERROR! File " << a.str("init_weight_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
      if (a.num("avg_weight_present", 0) > 0) {
        if (a.has("avg_weight_file") && !my_crf.readAverageFromFile(a.str("avg_weight_file").c_str(), (int)a.num("avg_weight_present", 0))) {
          std::cerr << "ERROR! File " << a.str("avg_weight_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
        if (a.num("crf_use_adagrad", 0) != 0 && a.has("grad_sqr_acc_file") && !my_crf.readGradSqrAccFromFile(a.str("grad_sqr_acc_file").c_str())) {
          std::cerr << "ERROR! File " << a.str("grad_sqr_acc_file") << " unable to be opened for reading.  ABORT!" << std::endl; return -1; }
      }
      my_crf.setInitIter((QNUInt32)a.num("init_iter", 0));
    }

    // ---- Main.cpp:632-684: the trainer
    std::string wf = a.str("out_weight_file");
    CRF_Trainer* my_trainer;
    if (method == "lbfgs") {
      my_trainer = new CRF_LBFGSTrainer(&my_crf, &str1, &wf[0]);
      ((CRF_LBFGSTrainer*)my_trainer)->setObjectiveFunction(EXPF);
    } else {
      my_trainer = new CRF_SGTrainer(&my_crf, &str1, &wf[0]);
      ((CRF_SGTrainer*)my_trainer)->setObjectiveFunction(EXPF);
      ((CRF_SGTrainer*)my_trainer)->setUseAdagrad((double)a.num("crf_use_adagrad", 0));
      ((CRF_SGTrainer*)my_trainer)->setEta(a.real("crf_adagrad_eta", 1.0));
      ((CRF_SGTrainer*)my_trainer)->setNThreads((int)threads);
      ((CRF_SGTrainer*)my_trainer)->setMinibatch((int)a.num("crf_bunch_size", 1));
    }
    if (rank == 0) {
      std::cout << "MINIBATCH SIZE: " << a.num("crf_bunch_size", 1) << std::endl;
      std::cout << "NUMBER OF THREADS: " << threads << std::endl;
      if (world > 1) std::cout << "RANKS: " << world << " (one process per GPU, gradient all-reduce over RCCL)" << std::endl;
    }
    my_trainer->setMaxIters((int)a.num("crf_epochs", 10));
    my_trainer->setLR((float)a.real("crf_lr", 0.008));
    my_trainer->setLRDecayRate((float)a.real("crf_lr_decay_rate", 1.0));
    my_trainer->setUttRpt((QNUInt32)a.num("crf_utt_rpt", 100));
    if (a.real("crf_gauss_var", 0.0) != 0.0) my_trainer->setGaussVar((float)a.real("crf_gauss_var", 0.0));
    {  // a finished run is not repeated (Main.cpp:676-682)
      const std::string done_file = my_trainer->getWeightDir() + "/.done.train";
      if (std::ifstream(done_file.c_str()).good()) {
        if (rank == 0) {
          std::cout << "The done file has already existed: " << done_file << std::endl;
          std::cout << "Finished." << std::endl;
        }
        return 0;
      }
    }
    my_trainer->train();
    delete my_trainer;
  } catch (std::exception& e) {
    std::cerr << "Exception: " << e.what() << std::endl;
    return -1;
  }
  return 0;
}
