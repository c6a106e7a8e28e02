// Test translation unit (CPU only):
//   compose_chain chain <lattice.txt> <m1.txt> <m2.txt>  -> crf_amd::composeFst(m1, m2), then composeShortestPath(lattice, .):
//                                                           "states <n> arcs <m>" of the composed machine, then the
//                                                           lines compose_best_path prints
//   compose_chain mlf <file.mlf> <symbols.txt> <name>     -> CRF_MLFManager(file, ., symbols).getFst(name) as text arcs
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <iostream>
#include <map>
#include <string>

#include "crf_amd.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  try {
    if (!strcmp(argv[1], "chain")) {
      crf_amd::ArcListFst lat, m1, m2, m12, best;
      crf_amd::readFstText(argv[2], &lat);
      crf_amd::readFstText(argv[3], &m1);
      crf_amd::readFstText(argv[4], &m2);
      crf_amd::composeFst(m1, m2, &m12);
      printf("states %d arcs %zu\n", m12.n_states, m12.arcs.size());
      float total = 0;
      if (m12.finals.empty() || !crf_amd::composeShortestPath(lat, m12, &best, &total)) { printf("nopath\n"); return 0; }
      printf("total %.9g\n", (double)total);
      for (const scrf_arc& a : best.arcs) printf("%d %d %.9g\n", a.ilabel, a.olabel, (double)a.w);
      printf("final %.9g\n", (double)best.final_weight);
    } else {
      std::map<std::string, long> sym;
      std::ifstream f(argv[3]);
      std::string s;
      long id;
      while (f >> s >> id) sym[s] = id;
      CRF_MLFManager mgr(argv[2], nullptr, &sym);
      crf_amd::ArcListFst fst;
      mgr.getFst(argv[4], &fst);
      printf("start %d states %d\n", fst.start, fst.n_states);
      for (const scrf_arc& a : fst.arcs) printf("%d %d %d %d %.9g\n", a.src, a.dst, a.ilabel, a.olabel, (double)a.w);
      printf("final %d %.9g\n", fst.final_state, (double)fst.final_weight);
    }
  } catch (std::exception& e) {
    fprintf(stderr, "Exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
