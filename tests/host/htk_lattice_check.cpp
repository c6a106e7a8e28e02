// Test translation unit (CPU only): htk_lattice_check <fst.txt> <symbols.txt | -> <utterance name> <out.slf>
//   crf_amd::readFstText, FST2HTK_lat::convert + Write (host/htk_lattice.h, the converter behind CRFDecode's htk_lat_outdir);
//   prints "nodes <N> arcs <L>"; a conversion error prints the message and exits with 3
#include <stdio.h>
#include <string.h>

#include <fstream>
#include <iostream>

#include "crf_amd.h"
#include "htk_lattice.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  try {
    crf_amd::ArcListFst fst;
    crf_amd::readFstText(argv[1], &fst);
    std::map<long, std::string> sym;
    const bool have_sym = strcmp(argv[2], "-") != 0;
    if (have_sym) {
      std::ifstream f(argv[2]);
      std::string s;
      long id;
      while (f >> s >> id) sym[id] = s;
    }
    FST2HTK_lat conv;
    conv.convert(fst);
    conv.Write(argv[4], argv[3], have_sym ? &sym : nullptr);
    printf("nodes %zu arcs %zu\n", conv.numNodes(), conv.numArcs());
  } catch (HtkLatticeError& e) {
    fprintf(stderr, "%s\n", e.what());
    return 3;
  } catch (std::exception& e) {
    fprintf(stderr, "Exception: %s\n", e.what());
    return 1;
  }
  return 0;
}
