// qn_filetool -- host-only converter/inspector for the pfile and ILAB formats of qn_files.h
// (the jobs QuickNet's pfile_create / pfile_print / labcat do for the reference's scripts,
// demo/kaldi-mods/mylocal/htk_to_pf.sh).  No GPU.
//   qn_filetool pfile_info  in.pfile
//   qn_filetool pfile2ascii in.pfile out.ascii        lines `sent frame f0 f1 .. l0 ..`
//   qn_filetool ascii2pfile in.ascii out.pfile [n_labels=0]
//   qn_filetool ilab2ascii  in.ilab out.ascii         lines `sent frame label`
//   qn_filetool ascii2ilab  in.ascii out.ilab
//   qn_filetool range       "spec" n_sentences
#include <fstream>
#include <iostream>

#include "qn_files.h"

static int usage() {
  std::cerr << "usage: qn_filetool pfile_info|pfile2ascii|ascii2pfile|ilab2ascii|ascii2ilab|range ..." << std::endl;
  return 2;
}

int main(int argc, char** argv) {
  if (argc < 3) return usage();
  const std::string cmd = argv[1];
  try {
    if (cmd == "pfile_info") {
      qn::PFileReader r(argv[2]);
      const qn::PFileInfo& i = r.info();
      std::cout << "sentences " << i.n_sents << " frames " << i.n_frames << " features " << i.n_ftrs << " labels " << i.n_labs << std::endl;
    } else if (cmd == "pfile2ascii" && argc >= 4) {
      qn::PFileReader r(argv[2]);
      FILE* o = fopen(argv[3], "w");
      if (!o) qn::fail(argv[3], "cannot open for writing");
      std::vector<float> f;
      std::vector<uint32_t> l;
      const uint32_t W = r.info().n_ftrs, NL = r.info().n_labs;
      for (uint32_t s = 0; s < r.num_sents(); s++) {
        r.read_sent(s, &f, &l);
        for (uint32_t t = 0; t < r.num_frames(s); t++) {
          fprintf(o, "%u %u", s, t);
          for (uint32_t k = 0; k < W; k++) fprintf(o, " %.9g", (double)f[(size_t)t * W + k]);
          for (uint32_t k = 0; k < NL; k++) fprintf(o, " %u", l[(size_t)t * NL + k]);
          fputc('\n', o);
        }
      }
      fclose(o);
    } else if (cmd == "ascii2pfile" && argc >= 4) {
      const uint32_t NL = argc >= 5 ? (uint32_t)atoi(argv[4]) : 0;
      std::ifstream in(argv[2]);
      if (!in.is_open()) qn::fail(argv[2], "cannot open for reading");
      std::string line;
      long cur = -1;
      uint32_t W = 0;
      bool first = true;
      std::vector<float> f;
      std::vector<uint32_t> l;
      qn::PFileWriter* w = nullptr;
      auto flush = [&]() { if (w && cur >= 0) w->write_sent(f.data(), l.data(), W + NL ? (uint32_t)((f.size() + l.size()) / (W + NL)) : 0); f.clear(); l.clear(); };
      while (getline(in, line)) {
        std::istringstream is(line);
        long s, t;
        if (!(is >> s >> t)) continue;
        std::vector<double> v;
        double x;
        while (is >> x) v.push_back(x);
        if (first) {
          if (v.size() < NL) qn::fail(argv[2], "fewer columns than labels");
          W = (uint32_t)v.size() - NL;
          w = new qn::PFileWriter(argv[3], W, NL);
          first = false;
        }
        if (v.size() != W + NL) qn::fail(argv[2], "ragged line");
        if (s != cur) {
          flush();
          if (s != cur + 1) qn::fail(argv[2], "sentence numbers must ascend by one");
          cur = s;
        }
        for (uint32_t k = 0; k < W; k++) f.push_back((float)v[k]);
        for (uint32_t k = 0; k < NL; k++) l.push_back((uint32_t)v[W + k]);
      }
      flush();
      if (!w) qn::fail(argv[2], "no data lines");
      w->close();
      delete w;
    } else if (cmd == "ilab2ascii" && argc >= 4) {
      qn::ILabFile f = qn::read_ilab(argv[2]);
      FILE* o = fopen(argv[3], "w");
      if (!o) qn::fail(argv[3], "cannot open for writing");
      for (size_t s = 0; s < f.labels.size(); s++)
        for (size_t t = 0; t < f.labels[s].size(); t++) fprintf(o, "%zu %zu %u\n", s, t, f.labels[s][t]);
      fclose(o);
      std::cout << "sentences " << f.n_sents << " frames " << f.n_frames << " label_bits " << f.label_bits << std::endl;
    } else if (cmd == "ascii2ilab" && argc >= 4) {
      std::ifstream in(argv[2]);
      if (!in.is_open()) qn::fail(argv[2], "cannot open for reading");
      std::vector<std::vector<uint32_t> > labs;
      long s, t, l;
      while (in >> s >> t >> l) {
        if (s < 0 || l < 0) qn::fail(argv[2], "negative field");
        if ((size_t)s >= labs.size()) labs.resize(s + 1);
        if ((size_t)t != labs[s].size()) qn::fail(argv[2], "frame numbers must ascend by one");
        labs[s].push_back((uint32_t)l);
      }
      qn::write_ilab(argv[3], labs);
    } else if (cmd == "range" && argc >= 4) {
      std::vector<uint32_t> r = qn::parse_range(argv[2], (uint32_t)atoi(argv[3]));
      for (size_t i = 0; i < r.size(); i++) std::cout << (i ? " " : "") << r[i];
      std::cout << std::endl;
    } else {
      return usage();
    }
  } catch (std::exception& e) {
    std::cerr << "Exception: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
