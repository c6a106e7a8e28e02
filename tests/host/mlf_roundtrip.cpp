// mlf_roundtrip.cpp -- reads an HTK master label file through CRF_MLFManager (asr-craft_amd/host/crf_amd.h, after
// io/CRF_MLFManager.cpp of the reference) with a `name id` symbol list, asks for every entry's transcript acceptor and
// writes the MLF again from what comes back.  tests/test_ref_data_pins.py compares the output with the input file
// (the reference's own demo/timit-aux/timit_test39.mlf) byte for byte.
//   mlf_roundtrip <mlf> <symbol list: optional count line, then `name id` lines>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

#include "crf_amd.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  std::map<std::string, long> sym;
  std::map<long, std::string> name_of;
  {
    std::ifstream f(argv[2]);
    std::string ln;
    while (std::getline(f, ln)) {
      std::istringstream is(ln);
      std::string nm;
      long id;
      if (is >> nm >> id) { sym[nm] = id; name_of[id] = nm; }
    }
  }
  try {
    CRF_MLFManager mlf(argv[1], nullptr, &sym);
    std::ifstream f(argv[1]);
    std::string ln;
    std::cout << "#!MLF!#\n";
    while (std::getline(f, ln)) {
      if (ln.empty() || ln[0] != '"') continue;
      crf_amd::ArcListFst fst;
      mlf.getFst(ln, &fst);     // the entry line itself: the manager derives the same key from it as when it read it
      std::cout << ln << "\n";
      int at = fst.start;
      for (const scrf_arc& a : fst.arcs) {     // a chain: arc i leaves state i
        if (a.src != at || a.ilabel != a.olabel || a.w != 0.0f) { std::cerr << "not a plain linear acceptor" << std::endl; return 1; }
        std::cout << (name_of.count(a.ilabel) ? name_of[a.ilabel] : std::string("<") + std::to_string(a.ilabel) + ">") << "\n";
        at = a.dst;
      }
      if (at != fst.final_state) { std::cerr << "the chain does not end in the final state" << std::endl; return 1; }
      std::cout << ".\n";
    }
  } catch (const std::exception& e) {
    std::cerr << "Exception: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
