// Test translation unit (CPU only): crf_amd::composeShortestPath on two OpenFST-text machines.
//   compose_best_path <lattice.txt> <lm.txt>   ->  "total <w>" then one line per arc "ilabel olabel weight", then "final <w>"
#include <stdio.h>

#include <iostream>

#include "crf_amd.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  try {
    crf_amd::ArcListFst lat, lm, best;
    crf_amd::readFstText(argv[1], &lat);
    crf_amd::readFstText(argv[2], &lm);
    float total = 0;
    const bool ok = crf_amd::composeShortestPath(lat, lm, &best, &total);
    if (!ok) { printf("nopath\n"); return 0; }
    printf("total %.9g\n", (double)total);
    for (const scrf_arc& a : best.arcs) printf("%d %d %.9g\n", a.ilabel, a.olabel, (double)a.w);
    printf("final %.9g\n", (double)best.final_weight);
  } catch (std::exception& e) {
    fprintf(stderr, "Exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
