// weight_file_text.cpp -- CRF_Model::writeToFile on a large vector (the slices of which are formatted by several threads)
// against the reference's way of writing it, `ofile << lambda[i] << endl` with the stream's default format
// (CRF_Model.cpp of the reference): the two files must be the same bytes.   weight_file_text <out file> <n>
#include <cmath>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "crf_amd.h"

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const size_t n = (size_t)atol(argv[2]);
  std::vector<double> v(n);
  unsigned long long s = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0 - 0.5;
    const int e = (int)((s >> 3) % 41) - 20;
    v[i] = (i % 97 == 0) ? 0.0 : u * std::pow(10.0, e);
  }
  if (n > 10) { v[1] = 1e-310; v[2] = -0.0; v[3] = 123456.5; v[4] = 1234567.0; v[5] = 1e300; v[6] = std::numeric_limits<double>::infinity(); v[7] = 0.1; v[8] = 100000; v[9] = 999999.5; }
  try {
    CRF_Model m(2);
    m.writeToFile(argv[1], v.data(), (QNUInt32)n);
  } catch (const std::exception& ex) {
    std::cerr << ex.what() << std::endl;
    return 1;
  }
  std::ostringstream want;
  for (size_t i = 0; i < n; i++) want << v[i] << "\n";
  std::ifstream f(argv[1], std::ios::binary);
  std::stringstream got;
  got << f.rdbuf();
  if (got.str() != want.str()) { std::cerr << "weight file text differs from the stream rendering" << std::endl; return 1; }
  std::cout << "same " << want.str().size() << " bytes" << std::endl;
  return 0;
}
