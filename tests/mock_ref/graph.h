// tests/mock_ref/graph.h -- TEST-ONLY declaration mock of the few members of the reference's graph.h that
// include/gaml_hip_prob_calculator.h touches. Written for this repository (nothing is copied from the reference);
// it exists so that the adapter header is COMPILED, LINKED and RUN by the test suite -- a syntax / link / plumbing
// check, NOT a parity statement about graph.cc (the reference itself needs Boost, which the image lacks).
//
// What the adapter needs (reference graph.h line numbers for the real declarations):
//   Graph::nodes[i]->s                                   graph.h:74-110, 233-306
//   ReadSet{filename_, mismatch_prob_}                   graph.h:344-442 (private there: the adapter is a friend)
//   PacbioReadSet{filename_, mismatch_prob_ (logdouble), aligment_cache_, anchors_cache_, FilterReads,
//                 PacbioAligment{position, position_end, read_id, prob}}   graph.h:444-600
//   ScoringState                                         graph.h:612-619
//   using namespace std (the reference's headers do)
#ifndef MOCK_REF_GRAPH_H__
#define MOCK_REF_GRAPH_H__
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "logdouble.hpp"

using namespace std;

namespace std {
template <> struct hash<vector<int>> {  // the reference hashes int vectors too (its own functor)
  size_t operator()(const vector<int>& v) const { size_t h = v.size(); for (int x : v) h = h * 1000003u + (unsigned)x; return h; }
};
}

struct Node { int id; string s; };

class Graph {
 public:
  vector<Node*> nodes;
  bool Load(const string& file) {  // Velvet LastGraph: header, then per node one skipped line + forward + reverse sequence
    ifstream f(file.c_str());
    if (!f.is_open()) return false;
    string l;
    getline(f, l);
    int n = atoi(l.c_str());
    for (int i = 0; i < n; i++) {
      getline(f, l);
      for (int k = 0; k < 2; k++) { Node* nd = new Node(); nd->id = 2 * i + k; getline(f, nd->s); nodes.push_back(nd); }
    }
    return true;
  }
};

class ProbCalculator;

class ReadSet {
 public:
  ReadSet(const string& name, const string& filename, double match_prob, double mismatch_prob)
      : name_(name), filename_(filename), match_prob_(match_prob), mismatch_prob_(mismatch_prob) {}
 private:
  friend class ProbCalculator;  // the two-line patch of INTEGRATION.md
  string name_, filename_;
  double match_prob_;
  double mismatch_prob_;
};

class PacbioReadSet {
 public:
  struct PacbioAligment {
    int position, position_end, read_id;
    logdouble prob;
  };
  PacbioReadSet(const string& name, const string& filename, double match_prob, double mismatch_prob)
      : name_(name), filename_(filename), match_prob_(match_prob), mismatch_prob_(mismatch_prob) {}
  void FilterReads(string out_filename, const unordered_set<int>& filter) {  // keep the listed reads (4-line FASTQ records)
    ifstream in(filename_.c_str());
    ofstream out(out_filename.c_str());
    string a, b, c, d;
    int id = 0;
    while (getline(in, a) && getline(in, b) && getline(in, c) && getline(in, d)) { if (filter.count(id)) out << a << "\n" << b << "\n" << c << "\n" << d << "\n"; id++; }
  }
 private:
  friend class ProbCalculator;
  string name_, filename_;
  logdouble match_prob_, mismatch_prob_;
  unordered_map<vector<int>, vector<PacbioAligment> > aligment_cache_;
  unordered_map<int, unordered_set<int> > anchors_cache_;
};

struct ScoringState {
  vector<vector<int> > old_paths;
  int bad_bases;
  vector<double> probs;
  ScoringState() : bad_bases(0) {}
};
#endif
