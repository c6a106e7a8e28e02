// htk_lattice.h -- a decoder lattice (FST) as an HTK Standard Lattice Format file: what CRFDecode writes into
// htk_lat_outdir (CRFDecode/src/Main.cpp:432-720, classes HTK_word_arc / FST_lat_node / FST2HTK_lat; used at :1143-1170
// on the best lattice or, with crf_if_output_full_lat, on the full search lattice).
//
// The conversion walks the FST breadth first from its start state (every state once, arcs in their order) and turns
// runs of arcs into WORD arcs of the HTK lattice:
//   - an arc with an output label starts a word at its source state: the source becomes an HTK node (numbered when first
//     needed), the target state remembers "a word arc that left HTK node n, weight so far -w";
//   - an arc without an output label stays inside the word: the target state inherits a copy of every word arc pending at
//     the source, each extended by -w;
//   - when a state starts a new word, or has no arcs at all, the word arcs pending AT it end there: it becomes an HTK
//     node and the arcs are written out (once per state).
// A state carries the word it lies in (the start state: none, written `!NULL`) and a time: its distance from the start
// state in ARCS (the start state: -1), written as t = 0.01 (time + 1) -- the frame clock of the frame-level decoder;
// the reference uses the same count for segmental lattices, so this does too.  Two paths that reach one state inside
// different words, or after a different number of arcs, stop the conversion (the reference's findOrInsertFstNode
// errors, :577-592), and so does an epsilon INPUT label (:619-621).  All weights count as acoustic; the LM weight of
// every arc is -1 * 0.0 (:640-652), which prints as `-0`.
#pragma once
#include <deque>
#include <fstream>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "crf_amd.h"

// raised where the reference prints an error and exits with -1
struct HtkLatticeError : public std::runtime_error {
  explicit HtkLatticeError(const std::string& what) : std::runtime_error(what) {}
};

class FST2HTK_lat {
 public:
  void convert(const crf_amd::ArcListFst& fst) {
    if (fst.start < 0 || fst.n_states <= 0) return;   // empty fst
    std::vector<std::vector<int> > out(fst.n_states);
    for (size_t i = 0; i < fst.arcs.size(); i++) out[fst.arcs[i].src].push_back((int)i);
    std::deque<int> todo(1, fst.start);
    std::set<int> queued;
    place(fst.start, kNoWord, kNoTime, "insertFstNode");
    while (!todo.empty()) {
      const int s = todo.front();
      todo.pop_front();
      for (int ai : out[s]) {
        const scrf_arc& a = fst.arcs[ai];
        if (a.ilabel == 0) throw HtkLatticeError("FST2HTK_lat::convert() ERROR: currently doesn't support epsilon input labels on any fst arc.");
        if (queued.insert(a.dst).second) todo.push_back(a.dst);
        const double ac = -1 * (double)a.w, lm = -1 * 0.0;
        if (a.olabel == 0) {   // inside the word: the pending word arcs of `s` move on, extended
          State& to = place(a.dst, at(s).word, at(s).time + 1, nullptr);
          const std::vector<Pending> from = at(s).pending;   // (a copy: `to` may be `s` in a degenerate machine)
          for (const Pending& p : from) to.pending.push_back(Pending{p.from, p.ac + ac, p.lm + lm});
        } else {               // a word begins at `s`
          State& to = place(a.dst, a.olabel, at(s).time + 1, nullptr);
          to.pending.push_back(Pending{htk_index(s), ac, lm});
          close(s);
        }
      }
      if (out[s].empty()) close(s);
    }
  }

  // osym == nullptr: no output symbol table (an error as soon as a node lies in a word)
  void Write(const std::string& filename, const std::string& uttname, const std::map<long, std::string>* osym) const {
    std::ofstream f(filename.c_str());
    f << "VERSION=1.0" << std::endl;
    f << "UTTERANCE=" << uttname << std::endl;
    f << "lmscale=1.00  wdpenalty=0.00" << std::endl;
    f << "prscale=1.00" << std::endl;
    f << "acscale=1.00" << std::endl;
    f << "N=" << htk_nodes.size() << " L=" << htk_arcs.size() << std::endl;
    for (size_t i = 0; i < htk_nodes.size(); i++) {
      const State& n = states.find(htk_nodes[i])->second;
      std::string label;
      if (n.word == kNoWord) label = "!NULL";
      else if (osym != nullptr && f.is_open()) { auto it = osym->find(n.word); label = it == osym->end() ? std::string() : it->second; }
      else throw HtkLatticeError("CRFDecode ERROR: output symbol table has not been set or htk_lat_stream is already closed.");
      f << "I=" << i << " t=" << kSecPerFrame * (n.time + 1) << " W=" << label;
      if (n.word != kNoWord) f << " v=1";
      f << std::endl;
    }
    for (size_t j = 0; j < htk_arcs.size(); j++)
      f << "J=" << j << " S=" << htk_arcs[j].from << " E=" << htk_arcs[j].to << " a=" << htk_arcs[j].ac << " l=" << htk_arcs[j].lm << " r=0.00" << std::endl;
  }

  size_t numNodes() const { return htk_nodes.size(); }
  size_t numArcs() const { return htk_arcs.size(); }

 private:
  static constexpr int kNoWord = -1, kNoTime = -1, kNoNode = -1;
  static constexpr double kSecPerFrame = 0.01;
  struct Pending { int from; double ac, lm; };          // a word arc under way: the HTK node it left, weights so far
  struct WordArc { int from, to; double ac, lm; };
  struct State {
    int word = kNoWord, time = kNoTime, node = kNoNode;
    bool closed = false;                                // its incoming word arcs are in the HTK lattice already
    std::vector<Pending> pending;
  };
  std::map<int, State> states;                           // by FST state
  std::vector<int> htk_nodes;                            // FST state of every HTK node
  std::vector<WordArc> htk_arcs;

  State& at(int s) { return states.find(s)->second; }
  // the state's record: made with (word, time) on first sight, checked against them afterwards
  State& place(int s, int word, int time, const char* must_be_new) {
    auto it = states.find(s);
    if (it == states.end()) {
      State& n = states[s];
      n.word = word;
      n.time = time;
      return n;
    }
    if (must_be_new) throw HtkLatticeError("FST2HTK_lat::insertFstNode() ERROR: state " + std::to_string(s) + " is already in the fst_node_map.");
    State& n = it->second;
    if (n.word != word)
      throw HtkLatticeError("FST2HTK_lat::findOrInsertFstNode() ERROR: two incoming arcs going through the fst state " + std::to_string(s) +
                            " with different word labels: " + std::to_string(n.word) + " and " + std::to_string(word));
    if (n.time != time)
      throw HtkLatticeError("FST2HTK_lat::findOrInsertFstNode() ERROR: two paths reach the fst state " + std::to_string(s) +
                            " at different time frame: " + std::to_string(n.time) + " and " + std::to_string(time));
    return n;
  }
  int htk_index(int s) {
    State& n = at(s);
    if (n.node == kNoNode) { n.node = (int)htk_nodes.size(); htk_nodes.push_back(s); }
    return n.node;
  }
  void close(int s) {
    State& n = at(s);
    if (n.closed) return;
    const int me = htk_index(s);
    for (const Pending& p : n.pending) htk_arcs.push_back(WordArc{p.from, me, p.ac, p.lm});
    n.closed = true;
  }
};
