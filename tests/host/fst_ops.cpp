// Test translation unit (CPU only): the FST operations of CRFFstDecode's phone-penalty / pruning stages.
//   fst_ops compose <a.txt> <b.txt> <filter 0|1>   -> crf_amd::composeFst(a, b, ., ., filter) as text
//   fst_ops rmeps <a.txt>                          -> crf_amd::rmEpsilonLog
//   fst_ops prune <a.txt> <threshold>              -> crf_amd::pruneFst
//   fst_ops topsort <a.txt>                        -> crf_amd::topSortFst ("cyclic" when it refuses)
// output: "start <s> states <n>", one line "src dst ilabel olabel weight" per arc, one line "final <s> <w>" per final state
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>

#include "crf_amd.h"

static void dump(const crf_amd::ArcListFst& f) {
  printf("start %d states %d\n", f.start, f.n_states);
  for (const scrf_arc& a : f.arcs) printf("%d %d %d %d %.9g\n", a.src, a.dst, a.ilabel, a.olabel, (double)a.w);
  for (const auto& x : f.finals) printf("final %d %.9g\n", x.first, (double)x.second);
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  try {
    crf_amd::ArcListFst a, b, out;
    crf_amd::readFstText(argv[2], &a);
    if (!strcmp(argv[1], "compose")) {
      crf_amd::readFstText(argv[3], &b);
      crf_amd::composeFst(a, b, &out, (size_t)1 << 20, atoi(argv[4]) != 0);
      dump(out);
    } else if (!strcmp(argv[1], "rmeps")) {
      crf_amd::rmEpsilonLog(&a);
      dump(a);
    } else if (!strcmp(argv[1], "prune")) {
      crf_amd::pruneFst(a, &out, (float)atof(argv[3]));
      dump(out);
    } else if (!strcmp(argv[1], "topsort")) {
      if (!crf_amd::topSortFst(&a)) { printf("cyclic\n"); return 0; }
      dump(a);
    } else return 2;
  } catch (std::exception& e) {
    fprintf(stderr, "Exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
