// CRFFstDecode -- lattice decode front-end (CRFFstDecode/src/Main.cpp): per utterance build the
// lattice, take the best path (ShortestPath/Project/RmEpsilon/TopSort), write `sent pos label`
// lines (the ILAB content as ascii); crf_lat_outdir additionally dumps the arc list as text.
// LM / dictionary composition needs OpenFST and stays host-side future work.
#include "cli_common.h"

int main(int argc, char** argv) {
  Args a(argc, argv);
  CliModel m;
  auto data = load_streams(a, &m);
  if (!a.has("weight_file")) { std::cerr << "weight_file is required" << std::endl; return 1; }
  CRF_Model crf(m.L);
  crf.setLabMaxDur(m.D);
  crf.setNActualLabs(m.fmap.nActualLabs);
  crf.setModelType(m.mtype);
  crf.setDevice((int)a.num("crf_device", 0));
  try {
    crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&m.fmap));
  } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  if (!crf.readFromFile(a.str("weight_file").c_str())) { std::cerr << "ERROR! File " << a.str("weight_file") << " unable to be opened for reading" << std::endl; return -1; }
  CRF_MemoryFeatureStream strm(m.recipes, m.D, m.fmap.nActualLabs);
  std::vector<uint32_t> sents;
  try {
    sents = select_sents(a, "crf_eval_range", data[0].size());
  } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  for (uint32_t u : sents) {
    std::vector<std::vector<float> > fr(data.size());
    for (size_t s = 0; s < data.size(); s++) { fr[s] = data[s].get(u); data[s].drop(u); }
    strm.addUtterance(fr, std::vector<uint32_t>());
  }
  // crf_output_format=ilab writes QuickNet ILAB (CRFFstDecode/src/Main.cpp:203); the default here
  // is the same content as ascii `sent pos label` lines
  const std::string ofmt = a.str("crf_output_format", "ascii");
  if (ofmt != "ascii" && ofmt != "ilab") { std::cerr << "crf_output_format=" << ofmt << " is not built (ascii|ilab)" << std::endl; return 1; }
  if (ofmt == "ilab" && !a.has("crf_output_labelfile")) { std::cerr << "crf_output_format=ilab needs crf_output_labelfile" << std::endl; return 1; }
  std::vector<std::vector<uint32_t> > all_labs;
  std::ofstream out;
  if (a.has("crf_output_labelfile") && ofmt == "ascii") out.open(a.str("crf_output_labelfile").c_str());
  std::ostream& os = out.is_open() ? (std::ostream&)out : std::cout;
  strm.rewind();
  size_t u = 0;
  auto emit = [&](const std::vector<uint32_t>& labs) {
    if (ofmt == "ascii")
      for (size_t i = 0; i < labs.size(); i++) os << u << " " << i << " " << labs[i] << "\n";
    all_labs.push_back(labs);
    u++;
  };
  if (!a.has("crf_lat_outdir")) {
    // best paths only: whole device batches of utterances (crf_bunch_size of them, default 256)
    const size_t bunch = (size_t)std::max(1L, a.num("crf_bunch_size", 256));
    bool at_end = strm.nextseg() == QN_SEGID_BAD;
    while (!at_end) {
      try {
        std::vector<std::vector<uint32_t> > labs;
        std::vector<float> costs;
        crf_amd_best_paths(&strm, &crf, bunch, &labs, &costs, &at_end);
        for (const auto& l : labs) emit(l);
      } catch (std::exception& e) {
        std::cerr << "Exception: " << e.what() << std::endl;
        return -1;
      }
    }
  } else {
    while (strm.nextseg() != QN_SEGID_BAD) {
      try {  // the reference prints the exception and continues with the next utterance (:1052-1054)
        crf_amd::ArcListFst fst;
        CRF_LatticeBuilder lb(&strm, &crf);
        lb.buildLattice(&fst, false, (crf_amd::ArcListFst*)nullptr, false);
        // the reference writes the lattice as an OpenFST binary, fst.<n>.final.fst (:832-837); here that file
        // (layout unpinned, crf_amd.h) plus the same arcs as text
        crf_amd::writeFstBinary((a.str("crf_lat_outdir") + "/fst." + std::to_string(u) + ".final.fst").c_str(), fst);
        std::ofstream lf((a.str("crf_lat_outdir") + "/fst." + std::to_string(u) + ".txt").c_str());
        for (const scrf_arc& c : fst.arcs) lf << c.src << " " << c.dst << " " << c.ilabel << " " << c.olabel << " " << c.w << "\n";
        lf << fst.final_state << "\n";
        float cost = 0;
        emit(crf_amd_best_path(&strm, &crf, &cost));
      } catch (std::exception& e) {
        std::cerr << "Exception: " << e.what() << std::endl;
        emit(std::vector<uint32_t>());
      }
    }
  }
  if (ofmt == "ilab") {
    try {
      qn::write_ilab(a.str("crf_output_labelfile"), all_labs);
    } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  }
  return 0;
}
