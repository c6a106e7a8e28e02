// CRFDecode -- Viterbi decode front-end (CRFDecode/src/Main.cpp): against the reference's own
// free-phone-loop LM (no LM given) or against an LM FST in OpenFST text format (crf_lm_txt): per utterance the best path from
// CRF_ViterbiDecoder_StdSeg_NoSegTransFtr::nStateDecode, written as an HTK MLF
// (crf_output_mlffile + crf_olist + crf_osymbols, Main.cpp:803-835,1176-1330) and, with
// crf_lat_outdir, as a text arc list `src dst ilabel olabel weight` + `final weight` per
// utterance (the reference writes OpenFST binaries there).  crf_if_output_full_lat: the search lattice instead of the
// best path goes to crf_lat_outdir (Main.cpp:1094-1141); htk_lat_outdir: that lattice (or the best path) as an HTK SLF
// file <name from crf_olist>.slf (Main.cpp:1143-1170, htk_lattice.h).  crf_lm_arpa is refused.
#include "cli_common.h"
#include "htk_lattice.h"

// OpenFST text symbol table: `symbol id` per line
static std::map<long, std::string> read_symbols(const std::string& path) {
  std::ifstream f(path.c_str());
  if (!f.is_open()) { std::cerr << "ERROR: Failed opening file: " << path << std::endl; exit(-1); }
  std::map<long, std::string> m;
  std::string sym;
  long id;
  while (f >> sym >> id) m[id] = sym;
  return m;
}

int main(int argc, char** argv) {
  Args a(argc, argv);
  if (a.has("crf_lm_arpa")) { std::cerr << "crf_lm_arpa: ARPA language models are not built; compile the LM to an FST and pass it as crf_lm_bin, or its `fstprint` text as crf_lm_txt" << std::endl; return 1; }
  if (!a.has("crf_output_labelfile") && !a.has("crf_output_mlffile")) { std::cerr << "At least one of crf_output_labelfile or crf_output_mlffile must be assigned" << std::endl; return -1; }
  if (!a.has("weight_file")) { std::cerr << "weight_file is required" << std::endl; return 1; }
  if (!a.has("crf_olist")) { std::cerr << "crf_olist required currently." << std::endl; return -1; }  // Main.cpp:1022-1025
  if (a.str("crf_decode_mode", "decode") != "decode") { std::cerr << "crf_decode_mode=" << a.str("crf_decode_mode") << " is not built" << std::endl; return 1; }
  CliModel m;
  std::vector<FtrData> data;
  try {
    data = load_streams(a, &m);
  } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  std::vector<std::string> olist;
  {
    std::ifstream f(a.str("crf_olist").c_str());
    if (!f.is_open()) { std::cerr << "ERROR: Failed opening file: " << a.str("crf_olist") << std::endl; return -1; }
    std::string s;
    while (getline(f, s)) olist.push_back(s);
  }
  // language model: OpenFST text format (tropical weights), numeric labels: ilabel = phone + 1, olabel = word
  crf_amd::ArcListFst lm;
  const bool have_lm = a.has("crf_lm_txt") || a.has("crf_lm_bin");
  if (have_lm) {
    try {
      if (a.has("crf_lm_txt")) crf_amd::readFstText(a.str("crf_lm_txt").c_str(), &lm);
      else {
        std::cout << "Reading in LM fst from file: " << a.str("crf_lm_bin") << std::endl;
        crf_amd::readFstBinary(a.str("crf_lm_bin").c_str(), &lm);
      }
    } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
    if (a.has("crf_disambig")) {  // disambiguation symbols become epsilons on the input side (Main.cpp:857-893)
      std::ifstream df(a.str("crf_disambig").c_str());
      std::string ln;
      std::vector<int> ids;
      while (getline(df, ln))
        if (!ln.empty()) {
          const int id = atoi(ln.c_str());
          if (id == 0) { std::cerr << "ERROR: invalid disambiguation ID (" << ln << ") from " << a.str("crf_disambig") << std::endl; return -1; }
          ids.push_back(id);
        }
      for (scrf_arc& c : lm.arcs)
        for (int id : ids)
          if (c.ilabel == id) c.ilabel = 0;
    } else {
      std::cout << "crf_disambig is not set: no disambiguation symbols for the decoding graph." << std::endl;
    }
    std::cout << "LM: " << lm.n_states << " states, " << lm.arcs.size() << " arcs, " << lm.finals.size() << " final" << std::endl;
  }
  std::map<long, std::string> osym;
  const bool have_osym = a.has("crf_osymbols");
  if (have_osym) osym = read_symbols(a.str("crf_osymbols"));
  std::ofstream mlf;
  if (a.has("crf_output_mlffile")) {
    mlf.open(a.str("crf_output_mlffile").c_str());
    mlf << "#!MLF!#" << std::endl;
  }
  if (a.has("crf_output_labelfile")) { std::ofstream touch(a.str("crf_output_labelfile").c_str()); }  // opened, never written (Main.cpp:794-801)
  const bool out_frames = a.num("crf_mlf_output_frames", 0) != 0;

  CRF_Model crf(m.L);
  crf.setLabMaxDur(m.D);
  crf.setNActualLabs(m.fmap.nActualLabs);
  crf.setModelType(m.mtype);
  crf.setDevice((int)a.num("crf_device", 0));
  std::cout << "LABELS: " << crf.getNLabs() << std::endl;
  std::cout << "LABEL_MAXIMUM_DURATION: " << crf.getLabMaxDur() << std::endl;
  std::cout << "ACTUAL_LABELS: " << crf.getNActualLabs() << std::endl;
  std::vector<uint32_t> sents;
  try {
    if (m.mtype == STDFRAME && m.D != 1) throw std::runtime_error("the maximum duration of labels must be 1 for \"stdframe\" CRF model.");
    if (m.mtype != STDFRAME && m.mtype != STDSEG_NO_DUR_NO_SEGTRANSFTR)
      throw std::runtime_error("CRF_ViterbiDecoder for CRF models other than \"stdframe\" and \"stdseg_no_dur_no_segtransftr\" have not been implmented.");
    crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&m.fmap));
    sents = select_sents(a, "crf_eval_range", data[0].size());
  } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  if (!crf.readFromFile(a.str("weight_file").c_str())) { std::cerr << "ERROR: Failed opening file: " << a.str("weight_file") << std::endl; return -1; }
  CRF_MemoryFeatureStream strm(m.recipes, m.D, m.fmap.nActualLabs);
  for (uint32_t u : sents) {
    std::vector<std::vector<float> > fr(data.size());
    for (size_t s = 0; s < data.size(); s++) { fr[s] = data[s].get(u); data[s].drop(u); }
    strm.addUtterance(fr, std::vector<uint32_t>());
  }
  strm.rewind();
  size_t count = 0;
  while (strm.nextseg() != QN_SEGID_BAD) {
    const uint32_t u = sents[count];
    try {
      if (u >= olist.size()) throw std::runtime_error("main() in CRFDecode caught exception: eval sentence range goes out of the olist size.");
      std::cout << "Processing file: " << olist[u] << " (" << u << " in the olist) (" << count << " in the current test set)" << std::endl;
      CRF_ViterbiDecoder_StdSeg_NoSegTransFtr vd(&strm, &crf);
      const bool full_lat = a.num("crf_if_output_full_lat", 0) != 0;
      vd.setIfOutputFullFst(full_lat);
      crf_amd::ArcListFst best_lat, out_full_lat;
      vd.nStateDecode(&best_lat, have_lm ? &lm : (crf_amd::ArcListFst*)nullptr, &out_full_lat, a.real("crf_decode_beam", 0.0),
                      (unsigned)a.num("crf_decode_min_hyp", 0), (unsigned)a.num("crf_decode_max_hyp", 0), (float)a.real("crf_decode_hyp_inc", 0.05));
      const crf_amd::ArcListFst& dump_lat = full_lat ? out_full_lat : best_lat;   // what the lattice directories receive (:1131-1139, :1156-1168)
      std::cout << "Acoustic model weight (negative log potential) = " << vd.getBestWeight() << ", -Z(X) = " << -1 * vd.getZx()
                << ", language model weight (negative log probability) = " << 0 << std::endl;
      if (a.has("crf_lat_outdir")) {
        crf_amd::writeFstBinary((a.str("crf_lat_outdir") + "/" + olist[u] + ".fst").c_str(), dump_lat);   // Main.cpp:1126-1136
        const std::string fn = a.str("crf_lat_outdir") + "/" + olist[u] + ".fst.txt";
        std::ofstream lf(fn.c_str());
        if (!lf.is_open()) throw std::runtime_error("cannot write " + fn);
        char buf[64];
        for (const scrf_arc& c : dump_lat.arcs) {
          snprintf(buf, sizeof buf, "%.9g", (double)c.w);
          lf << c.src << " " << c.dst << " " << c.ilabel << " " << c.olabel << " " << buf << "\n";
        }
        for (const auto& fw : dump_lat.finals) {
          snprintf(buf, sizeof buf, "%.9g", (double)fw.second);
          lf << fw.first << " " << buf << "\n";
        }
      }
      if (a.has("htk_lat_outdir")) {   // Main.cpp:1143-1170
        const std::string slf = a.str("htk_lat_outdir") + "/" + olist[u] + ".slf";
        try {
          FST2HTK_lat to_htk;
          to_htk.convert(dump_lat);
          to_htk.Write(slf, olist[u], have_osym ? &osym : (const std::map<long, std::string>*)nullptr);
        } catch (HtkLatticeError& e) {   // the reference prints these and exits with -1 (:566-592, :619-621, :686-689)
          std::cerr << e.what() << std::endl;
          return -1;
        }
      }
      if (mlf.is_open()) {
        // the arc walk of Main.cpp:1176-1330 on the (already linear, epsilon-free on the input side)
        // best path: `current` counts arcs, a label is printed where a phone starts
        if (have_osym) { mlf << "\"" << olist[u] << "\"" << std::endl; std::cout << "\"" << olist[u] << "\" : "; }
        int phnStateStart = 0, current = 0;
        for (const scrf_arc& c : best_lat.arcs) {
          if (c.olabel == 0 && c.ilabel != 0) {
            current++;
          } else if (c.olabel != 0) {
            if (c.ilabel != 0) current++;
            if (have_osym) {
              if (out_frames) mlf << phnStateStart << "\t" << current << "\t";  // start inclusive, end exclusive
              const std::string w = osym.count(c.olabel) ? osym[c.olabel] : std::string();
              mlf << w << std::endl;
              std::cout << w << " ";
              phnStateStart = current;
            }
          }
        }
        if (have_osym) { mlf << "." << std::endl; std::cout << "." << std::endl; }
      }
    } catch (std::exception& e) {
      std::cerr << "Exception: " << e.what() << std::endl;
      return 0;  // the reference exits with 0 here (Main.cpp:1371-1374)
    }
    count++;
  }
  return 0;
}
