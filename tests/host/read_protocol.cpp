// Test translation unit: a CRF_FeatureStream that offers ONLY the reference's read() protocol
// (io/CRF_FeatureStream.h:54-66), fed from CRF_MemoryFeatureStream::read -- whose window vectors come from
// the engine's window kernel -- drives CRF_GradBuilder::buildGradient; the same utterance through the
// whole-utterance fast path must give the same windows (bit for bit) and the same gradient.
#include <math.h>

#include <iostream>
#include <vector>

#include "crf_amd.h"

struct ReadOnlyStream : CRF_FeatureStream {   // hides currentUtterance(): forces the read() protocol
  CRF_MemoryFeatureStream* in;
  explicit ReadOnlyStream(CRF_MemoryFeatureStream* s) : in(s) {}
  QN_SegID nextseg() override { return in->nextseg(); }
  size_t read(size_t bunch, float* f, QNUInt32* l) override { return in->read(bunch, f, l); }
  int rewind() override { return in->rewind(); }
  size_t num_ftrs() override { return in->num_ftrs(); }
  size_t num_labs() override { return in->num_labs(); }
};

int main() {
  const uint32_t L = 4, D = 3, W = 2, T = 7;
  try {
    scrf_stream_recipe r{W, 0, 0, 1};
    CRF_MemoryFeatureStream mem(std::vector<scrf_stream_recipe>(1, r), D);
    std::vector<std::vector<float> > fr(1, std::vector<float>(T * W));
    for (uint32_t t = 0; t < T; t++)
      for (uint32_t c = 0; c < W; c++) fr[0][t * W + c] = (float)(((t * 7 + c * 3) % 11) / 11.0);
    mem.addUtterance(fr, std::vector<uint32_t>{0, 0, 1, 1, 1, 1, 2});
    CRF_Model crf(L);
    crf.setLabMaxDur(D);
    crf.setNActualLabs(L);
    crf.setModelType(STDSEG_NO_DUR_NO_SEGTRANSFTR);
    crf.setTrainPrecision(SCRF_PREC_EXACT);
    CRF_FeatureMap_config fc;
    fc.map_type = STDSTATE; fc.numLabs = L; fc.numFeas = 8 * W + D; fc.stateFidxEnd = 8 * W + D - 1; fc.maxDur = D; fc.nActualLabs = L;
    crf.setFeatureMap(CRF_FeatureMap::createFeatureMap(&fc));
    std::vector<double> lam(crf.getLambdaLen());
    for (size_t i = 0; i < lam.size(); i++) lam[i] = (double)((long)((i * 37) % 19) - 9) / 50.0;
    crf.setLambda(lam.data(), (QNUInt32)lam.size());
    CRF_GradBuilder* gb = CRF_GradBuilder::create(&crf, EXPF);
    // (1) through read()
    ReadOnlyStream ro(&mem);
    ro.rewind(); ro.nextseg();
    std::vector<double> g1(lam.size(), 0.0), g2(lam.size(), 0.0);
    double zx1 = 0, zx2 = 0;
    const double n1 = gb->buildGradient(&ro, g1.data(), &zx1);
    // (2) through the whole-utterance fast path
    mem.rewind(); mem.nextseg();
    const double n2 = gb->buildGradient(&mem, g2.data(), &zx2);
    bool same = n1 == n2 && zx1 == zx2;
    double gs = 0;
    for (size_t i = 0; i < g1.size(); i++) { same = same && g1[i] == g2[i]; gs += fabs(g1[i]); }
    std::cout.precision(17);
    std::cout << "numer=" << n1 << " zx=" << zx1 << " gsum=" << gs << " windows_equal=" << (same ? 1 : 0) << std::endl;
    delete gb;
  } catch (std::exception& e) {
    std::cerr << "Exception: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
