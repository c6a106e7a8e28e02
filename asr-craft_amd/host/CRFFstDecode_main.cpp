// CRFFstDecode -- lattice decode front-end (CRFFstDecode/src/Main.cpp): per utterance build the
// lattice, take the best path (ShortestPath/Project/RmEpsilon/TopSort), write `sent pos label`
// lines (the ILAB content as ascii); crf_lat_outdir additionally dumps the arc list as text.
// crf_output_mlffile (+ crf_olist, crf_osymbols): the MLF path of the reference (Main.cpp:891-1044) -- the
// utterance's lattice, composed with an LM FST when crf_lm_txt (OpenFST text) or crf_lm_bin is given
// (Compose + ShortestPath + RmEpsilon + TopSort in one host pass, crf_amd::composeShortestPath; the LM reads
// the lattice's OUTPUT labels, phone + L*(dur-1) + 1, as the reference's does), best path written as HTK MLF
// lines (the LM's output symbols; without an LM the lattice's own output labels).  crf_mlf_output_frames adds
// the first and last frame of each label's segments.  crf_dict_bin / crf_dict_txt: the dictionary FST composed in between
// (lattice o dict o LM, Main.cpp:929-941); crf_align_mlffile: the utterance's transcript from an HTK MLF as a linear
// acceptor over the output symbols, composed in before the LM (:943-952, CRF_MLFManager); crf_mlf_output_states (+
// crf_isymbols): the phone label of every segment in front of the words.  crf_phn_bin / crf_phn_txt: the phone-penalty
// FST (:614-640 reads a log-arc VectorFst, maps it to the tropical semiring, sorts its arcs), composed onto the lattice
// first, epsilons removed on the LOG semiring, then Prune(crf_phn_wt) when that weight is not 0 (:898-926; the Minimize
// behind the Prune changes neither the paths nor their weights and is not run).  crf_dict_wt: Prune of lattice o dictionary
// (:935-940).  With a phone FST or a pruning weight the chain is materialised left to right as the reference does it
// (a Prune does not commute with the compositions behind it); otherwise everything right of the lattice is one machine.
// crf_pre_phn_wt, crf_lm_wt and crf_lm_arpa are declared by the reference (:118-128) and read by nothing there: accepted, no effect.
#include "cli_common.h"

#include <limits>
#include <set>

static std::map<long, std::string> read_symbols(const std::string& path) {   // OpenFST text symbol table: `symbol id`
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
  CliModel m;
  auto data = load_streams(a, &m);
  if (!a.has("weight_file")) { std::cerr << "weight_file is required" << std::endl; return 1; }
  for (const char* k : {"crf_pre_phn_wt", "crf_lm_wt", "crf_lm_arpa"})
    if (a.has(k)) std::cerr << "NOTE: " << k << " is declared by the reference's CRFFstDecode and read by nothing there: no effect" << std::endl;
  const float phn_wt = (float)a.real("crf_phn_wt", 0.0), dict_wt = (float)a.real("crf_dict_wt", 0.0);
  // crf_decode_mode=align (Main.cpp:464-471, :841-848): the best path of lattice o label acceptor -- the labels of
  // hardtarget_file in their order, every run of equal node labels stretched or shrunk to fit -- written to the label file
  const std::string mode = a.str("crf_decode_mode", "decode");
  if (mode != "decode" && mode != "align") { std::cerr << "crf_decode_mode=" << mode << " (decode|align)" << std::endl; return 1; }
  const bool align_mode = mode == "align";
  if (align_mode && !a.has("hardtarget_file")) { std::cerr << "hardtarget_file required when crf_decode_mode=align" << std::endl; return -1; }
  std::vector<std::vector<uint32_t> > hard_labs;
  if (align_mode) {
    try { hard_labs = read_labs(a.str("hardtarget_file")); }
    catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
  }
  const bool want_mlf = a.has("crf_output_mlffile");
  crf_amd::ArcListFst lm, dict, phn, chain0;   // chain0: dict o lm, composed once when no per-utterance acceptor sits between them
  const bool have_lm = a.has("crf_lm_txt") || a.has("crf_lm_bin");
  const bool have_dict = a.has("crf_dict_txt") || a.has("crf_dict_bin");
  const bool have_phn = a.has("crf_phn_txt") || a.has("crf_phn_bin");
  // a log-semiring epsilon removal or a Prune between the compositions: the chain is built left to right, machine by machine
  const bool staged = have_phn || (have_dict && dict_wt != 0.0f);
  const bool have_align = a.has("crf_align_mlffile");
  const bool out_states = a.num("crf_mlf_output_states", 0) != 0;
  std::vector<std::string> olist;
  std::map<long, std::string> osym, isym;
  std::map<std::string, long> osym_ids;
  std::unique_ptr<CRF_MLFManager> mlf_mgr;
  std::ofstream mlf;
  if (want_mlf) {
    if (!a.has("crf_olist")) { std::cerr << "crf_olist required with crf_output_mlffile." << std::endl; return -1; }
    std::ifstream f(a.str("crf_olist").c_str());
    if (!f.is_open()) { std::cerr << "ERROR: Failed opening file: " << a.str("crf_olist") << std::endl; return -1; }
    std::string ln;
    while (getline(f, ln)) olist.push_back(ln);
    if (a.has("crf_osymbols")) osym = read_symbols(a.str("crf_osymbols"));
    if (a.has("crf_isymbols")) isym = read_symbols(a.str("crf_isymbols"));
    for (const auto& kv : osym) osym_ids[kv.second] = kv.first;
    if (out_states && !a.has("crf_isymbols")) { std::cerr << "crf_isymbols required with crf_mlf_output_states" << std::endl; return -1; }
    try {
      if (a.has("crf_lm_txt")) crf_amd::readFstText(a.str("crf_lm_txt").c_str(), &lm);
      else if (a.has("crf_lm_bin")) crf_amd::readFstBinary(a.str("crf_lm_bin").c_str(), &lm);
      if (a.has("crf_dict_txt")) crf_amd::readFstText(a.str("crf_dict_txt").c_str(), &dict);
      else if (a.has("crf_dict_bin")) crf_amd::readFstBinary(a.str("crf_dict_bin").c_str(), &dict);
      if (a.has("crf_phn_txt")) crf_amd::readFstText(a.str("crf_phn_txt").c_str(), &phn);
      else if (a.has("crf_phn_bin")) crf_amd::readFstBinary(a.str("crf_phn_bin").c_str(), &phn);   // log arcs: the float values as they are (:616-623)
      if (have_align) {
        if (!a.has("crf_osymbols")) { std::cerr << "crf_osymbols required with crf_align_mlffile" << std::endl; return -1; }
        std::string mf = a.str("crf_align_mlffile"), ol = a.str("crf_olist");
        mlf_mgr.reset(new CRF_MLFManager(mf.c_str(), ol.c_str(), &osym_ids));
      }
      if (have_dict && have_lm && !have_align && !staged) crf_amd::composeFst(dict, lm, &chain0);
    } catch (std::exception& e) { std::cerr << "Exception: " << e.what() << std::endl; return -1; }
    if (have_lm) std::cout << "LM: " << lm.n_states << " states, " << lm.arcs.size() << " arcs, " << lm.finals.size() << " final" << std::endl;
    if (have_dict) std::cout << "Dictionary: " << dict.n_states << " states, " << dict.arcs.size() << " arcs" << std::endl;
    if (have_phn) std::cout << "Phone FST: " << phn.n_states << " states, " << phn.arcs.size() << " arcs" << std::endl;
    if (chain0.n_states) std::cout << "Dictionary o LM: " << chain0.n_states << " states, " << chain0.arcs.size() << " arcs" << std::endl;
    mlf.open(a.str("crf_output_mlffile").c_str());
    mlf << "#!MLF!#" << std::endl;
  } else if (have_lm || have_dict || have_align || have_phn) { std::cerr << "crf_lm_* / crf_dict_* / crf_phn_* / crf_align_mlffile need crf_output_mlffile (the label file holds the lattice's own best path)" << std::endl; return 1; }
  const bool out_frames = a.num("crf_mlf_output_frames", 0) != 0;
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
    if (align_mode) {
      if (u >= hard_labs.size() || hard_labs[u].size() != fr[0].size() / m.recipes[0].in_width) {
        std::cerr << "hardtarget_file: sentence " << u << ": one label per frame expected" << std::endl;
        return -1;
      }
      strm.addUtterance(fr, hard_labs[u]);
    } else {
      strm.addUtterance(fr, std::vector<uint32_t>());
    }
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
  // MLF path (Main.cpp:891-1044) of the lattice `fst` of sentence `sent`
  auto write_mlf = [&](const crf_amd::ArcListFst& fst, uint32_t sent) {
    if (sent >= olist.size()) throw std::runtime_error("eval sentence range goes out of the olist size.");
    crf_amd::ArcListFst best;
    float total = 0;
    bool ok;
    if (staged) {
      // Main.cpp:896-1006 machine by machine: working = lattice; o phone FST, RmEpsilon on the log semiring, Prune;
      // o dictionary, Prune; o transcript; then the shortest path of working o LM
      crf_amd::ArcListFst w = fst, t;
      if (have_phn) {
        crf_amd::composeFst(w, phn, &t, (size_t)1 << 22, true);
        crf_amd::rmEpsilonLog(&t);
        w = t;
        if (phn_wt != 0.0f) { crf_amd::pruneFst(w, &t, phn_wt); w = t; }
      }
      if (have_dict && w.n_states) {
        crf_amd::composeFst(w, dict, &t);
        w = t;
        if (dict_wt != 0.0f) { crf_amd::pruneFst(w, &t, dict_wt); w = t; }
      }
      if (have_align && w.n_states) {
        crf_amd::ArcListFst al;
        mlf_mgr->getFst(olist[sent], &al);
        crf_amd::composeFst(w, al, &t);
        w = t;
      }
      crf_amd::ArcListFst id;   // no LM: an acceptor of every output label the chain can emit
      if (!have_lm) {
        id.n_states = 1; id.start = 0; id.SetFinal(0, 0.0f);
        std::set<int> labs;
        for (const scrf_arc& c : w.arcs) if (c.olabel != 0) labs.insert(c.olabel);
        for (int l : labs) id.arcs.push_back(scrf_arc{0, l, l, 0.0f, 0});
      }
      if (w.n_states && !crf_amd::topSortFst(&w)) throw std::runtime_error("the composed lattice has a cycle (an epsilon-input loop in the phone FST or the dictionary)");
      total = std::numeric_limits<float>::infinity();
      ok = w.n_states && !w.finals.empty() && crf_amd::composeShortestPath(w, have_lm ? lm : id, &best, &total);
    } else if (have_lm || have_dict || have_align) {
      // everything right of the lattice as ONE machine: ((dict o transcript) o LM), then the product search over
      // the lattice (Compose is associative; the reference nests ComposeFst left to right, Main.cpp:929-1006)
      const crf_amd::ArcListFst* mach = chain0.n_states ? &chain0 : (have_dict ? &dict : (have_lm && !have_align ? &lm : nullptr));
      crf_amd::ArcListFst t1, t2;
      if (have_align) {
        crf_amd::ArcListFst al;
        mlf_mgr->getFst(olist[sent], &al);
        if (have_dict) { crf_amd::composeFst(dict, al, &t1); mach = &t1; }
        else { t1 = al; mach = &t1; }
        if (have_lm) { crf_amd::composeFst(*mach, lm, &t2); mach = &t2; }
      }
      ok = crf_amd::composeShortestPath(fst, *mach, &best, &total);
    } else {   // ShortestPath on the lattice alone: an LM that accepts every label and copies it
      crf_amd::ArcListFst id;
      id.n_states = 1; id.start = 0; id.SetFinal(0, 0.0f);
      for (uint32_t l = 1; l <= m.L * m.D; l++) id.arcs.push_back(scrf_arc{0, (int)l, (int)l, 0.0f, 0});
      ok = crf_amd::composeShortestPath(fst, id, &best, &total);
    }
    mlf << "\"" << olist[sent] << "\"" << std::endl;
    std::cout << "\"" << olist[sent] << "\" : ";
    if (!ok) std::cerr << "WARNING: no path through lattice and LM for " << olist[sent] << std::endl;
    uint32_t frame = 0, seg_start = 0;
    bool open_seg = false;
    for (const scrf_arc& c : best.arcs) {
      if (c.ilabel != 0) {   // a segment of the lattice: phone + L*(dur-1) + 1 (frame model: one frame)
        const uint32_t dur = (uint32_t)(c.ilabel - 1) / m.L + 1;
        if (!open_seg) { seg_start = frame; open_seg = true; }
        if (out_states) {   // the segment's phone, with its own frame span, ahead of the word it belongs to
          if (out_frames) mlf << frame << "\t" << (frame + dur - 1) << "\t";
          mlf << (isym.count(c.ilabel) ? isym[c.ilabel] : std::to_string(c.ilabel)) << std::endl;
        }
        frame += dur;
      }
      if (c.olabel != 0) {
        if (out_frames) mlf << seg_start << "\t" << (frame ? frame - 1 : 0) << "\t";
        const std::string w = osym.count(c.olabel) ? osym[c.olabel] : std::to_string(c.olabel);
        mlf << w << std::endl;
        std::cout << w << " ";
        open_seg = false;
      }
    }
    mlf << "." << std::endl;
    std::cout << ". (weight " << total << ")" << std::endl;
  };
  if (!a.has("crf_lat_outdir") && !want_mlf && !align_mode) {
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
        crf_amd::ArcListFst fst, lab_fst;
        CRF_LatticeBuilder lb(&strm, &crf);
        lb.buildLattice(&fst, align_mode, align_mode ? &lab_fst : (crf_amd::ArcListFst*)nullptr, false);
        if (a.has("crf_lat_outdir")) {
          // the reference writes the lattice as an OpenFST binary, fst.<n>.final.fst (:832-837); here that file
          // (layout unpinned, crf_amd.h) plus the same arcs as text
          crf_amd::writeFstBinary((a.str("crf_lat_outdir") + "/fst." + std::to_string(u) + ".final.fst").c_str(), fst);
          std::ofstream lf((a.str("crf_lat_outdir") + "/fst." + std::to_string(u) + ".txt").c_str());
          for (const scrf_arc& c : fst.arcs) lf << c.src << " " << c.dst << " " << c.ilabel << " " << c.olabel << " " << c.w << "\n";
          lf << fst.final_state << "\n";
        }
        if (want_mlf) write_mlf(fst, sents[u]);
        if (align_mode) {   // ShortestPath(Compose(lattice, labels)), Project(output), RmEpsilon: olabel - 1 per arc
          crf_amd::ArcListFst best;
          float total = 0;
          std::vector<uint32_t> labs;
          if (crf_amd::composeShortestPath(fst, lab_fst, &best, &total)) {
            for (const scrf_arc& c : best.arcs) { if (c.olabel != 0) labs.push_back((uint32_t)(c.olabel - 1)); }
          } else {
            std::cerr << "WARNING: the labels of sentence " << sents[u] << " do not fit its lattice" << std::endl;
          }
          emit(labs);
          continue;
        }
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
