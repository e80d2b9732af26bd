// gaml_host.h -- host-side C++ mirror of the reference's ProbCalculator / read-set / config
// surface for the likelihood path, implemented over the C ABI of libgaml_hip.so.
//
// Same names, argument meaning and error behaviour as the reference so that code written against
// gaml.cc's flow reads the same:
//   LoadConfig, PrepareReadSetFromConfig, PrepareReads   gaml.cc:748-909
//   Graph / LoadGraph                                     graph.h:233-306, graph.cc:52-106
//   ReadSet / PacbioReadSet (the public bits callers use) graph.h:344-395, 444-495
//   SingleReadConfig, PairedReadConfig, ProbCalculator    prob_calculator.h:7-124
// What is deliberately NOT here: the optimiser, the move generators, reachability tables, output
// writers, BLASR/bowtie drivers (SURVEY.md section 2 marks them out of scope).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <iterator>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/gaml_hip.h"

namespace gaml_host {

using std::pair;
using std::string;
using std::unordered_map;
using std::vector;

// ---- graph (only what the likelihood path reads: node sequences) ---------------------------
struct Graph {
  vector<string> nodes;  // nodes[i] = sequence of node i; twin of i is i^1
  string file;
};
inline bool LoadGraph(const string& filename, Graph& gr) {  // graph.cc:52-106
  std::ifstream f(filename.c_str());
  if (!f.is_open()) return false;
  string l;
  std::getline(f, l);
  int n = atoi(l.c_str());
  gr.nodes.assign(2 * (size_t)n, string());
  for (int i = 0; i < n; i++) {
    std::getline(f, l);
    std::getline(f, gr.nodes[2 * i]);
    std::getline(f, gr.nodes[2 * i + 1]);
  }
  gr.file = filename;
  printf("Loaded %d nodes\n", n);
  return true;
}

// ---- read sets ---------------------------------------------------------------------------------
class ReadSet {  // graph.h:344-395
 public:
  ReadSet(const string& name, const string& filename, double match_prob, double mismatch_prob)
      : match_prob_(match_prob), mismatch_prob_(mismatch_prob), name_(name), filename_(filename) {}
  void LoadAligments() {}     // on-disk cache: disabled for short reads in the reference (graph.cc:1036)
  void PreprocessReads() {}   // done by the library when the read set is added
  void PrepareReadIndex() {}
  const string& GetName() const { return name_; }
  const string& filename() const { return filename_; }
  int GetNumberOfReads() const { return reads_num_; }
  double match_prob_, mismatch_prob_;
  int reads_num_ = 0;  // filled when the context is built
 private:
  string name_, filename_;
};

class PacbioReadSet {  // graph.h:444-495
 public:
  PacbioReadSet(const string& name, const string& filename, double match_prob, double mismatch_prob)
      : match_prob_(match_prob), mismatch_prob_(mismatch_prob), name_(name), filename_(filename) {}
  void LoadAligments() {}
  void PreprocessReads() {}
  void NormalizeCache(const Graph&) {}
  void ComputeAnchors(const Graph&) {}
  const string& filename() const { return filename_; }
  int GetNumberOfReads() const { return reads_num_; }
  double match_prob_, mismatch_prob_;
  int reads_num_ = 0;
  // alignment records from outside (BLASR + banded DP in the reference, graph.cc:2650-2795):
  // one line per record: "<n> <node ids...> <position> <position_end> <read_id> <logprob>"
  string records_file;
 private:
  string name_, filename_;
};

// ---- configs (prob_calculator.h:7-35) ---------------------------------------------------------
struct SingleReadConfig {
  SingleReadConfig() {}
  SingleReadConfig(double pc, double s, double mp, double mps, double w, bool a)
      : penalty_constant(pc), step(s), min_prob_per_base(mp), min_prob_start(mps), weight(w), advice(a) {}
  double penalty_constant = 0, step = 50, min_prob_per_base = -0.7, min_prob_start = -10, weight = 1;
  bool advice = false;
};
struct PairedReadConfig {
  PairedReadConfig() {}
  PairedReadConfig(double pc, double s, double im, double is, double mp, double mps, double w, bool a)
      : penalty_constant(pc), step(s), insert_mean(im), insert_std(is), min_prob_per_base(mp), min_prob_start(mps),
        weight(w), advice(a) {}
  double penalty_constant = 0, step = 0, insert_mean = 0, insert_std = 0, min_prob_per_base = -0.7, min_prob_start = -10,
         weight = 1;
  bool advice = false;
};

// ---- config file (gaml.cc:32-51, 737-872) -------------------------------------------------------
inline double ExtractDouble(const string& key, unordered_map<string, string>& cfg, double def) {
  return cfg.count(key) ? atof(cfg[key].c_str()) : def;
}
inline bool LoadConfig(const string& config_file, unordered_map<string, string>& configs,
                       unordered_map<string, unordered_map<string, string>>& read_set_configs) {
  std::ifstream fi(config_file);
  if (fi.fail()) { printf("Failed to open config file\n"); return false; }
  string current, l;
  while (std::getline(fi, l)) {
    if (l.empty()) continue;
    if (l[0] == '[') current = l.substr(1, l.size() - 2);
    else if (l[0] >= 'a' && l[0] <= 'z') {
      size_t eq = l.find('=');
      if (eq == string::npos) { printf("Bad line in config file:\n%s\n", l.c_str()); return false; }
      if (current.empty()) configs[l.substr(0, eq)] = l.substr(eq + 1);
      else read_set_configs[current][l.substr(0, eq)] = l.substr(eq + 1);
    }
  }
  return true;
}
inline void PrepareReadSetFromConfig(unordered_map<string, unordered_map<string, string>>& read_set_configs,
                                     vector<pair<SingleReadConfig, ReadSet*>>& single_reads,
                                     vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>>& paired_reads,
                                     vector<pair<SingleReadConfig, PacbioReadSet*>>& pacbio_reads) {
  for (auto& e : read_set_configs) {  // hash order, as the reference (gaml.cc:788)
    auto& kv = e.second;
    string cache_prefix = kv.count("cache_prefix") ? kv["cache_prefix"] : e.first;
    if (!kv.count("type")) { fprintf(stderr, "No type for read set %s, ignoring...\n", e.first.c_str()); continue; }
    double weight = ExtractDouble("weight", kv, 1);
    bool advice = kv.count("advice") > 0;
    if (kv["type"] == "single" || kv["type"] == "pacbio") {
      if (!kv.count("filename")) { fprintf(stderr, "Missing filename for read set %s, ignoring...\n", e.first.c_str()); continue; }
      double mismatch = ExtractDouble("mismatch_prob", kv, 0.01), match = 1.0 - 4 * mismatch;
      SingleReadConfig cfg(ExtractDouble("penalty_constant", kv, 0), ExtractDouble("penalty_step", kv, 50),
                           ExtractDouble("min_prob_per_base", kv, -0.7), ExtractDouble("min_prob_start", kv, -10), weight, advice);
      if (kv["type"] == "single") single_reads.push_back({cfg, new ReadSet(cache_prefix, kv["filename"], match, mismatch)});
      else {
        auto* rs = new PacbioReadSet(cache_prefix, kv["filename"], match, mismatch);
        if (kv.count("records")) rs->records_file = kv["records"];  // extension: alignment records file
        pacbio_reads.push_back({cfg, rs});
      }
    } else if (kv["type"] == "paired") {
      const char* need[] = {"filename1", "filename2", "insert_mean", "insert_std"};
      bool ok = true;
      for (const char* k : need) if (!kv.count(k)) { fprintf(stderr, "Missing %s for read set %s, ignoring...\n", k, e.first.c_str()); ok = false; break; }
      if (!ok) continue;
      double im = atof(kv["insert_mean"].c_str()), is = atof(kv["insert_std"].c_str());
      double mismatch = ExtractDouble("mismatch_prob", kv, 0.01), match = 1.0 - 4 * mismatch;
      // sic: paired sets read min_prob_pre_base (gaml.cc:855); step = insert_mean - penalty_step (gaml.cc:860)
      PairedReadConfig cfg(ExtractDouble("penalty_constant", kv, 0), im - ExtractDouble("penalty_step", kv, 50), im, is,
                           ExtractDouble("min_prob_pre_base", kv, -0.7), ExtractDouble("min_prob_start", kv, -10), weight, advice);
      paired_reads.push_back({cfg, {new ReadSet(cache_prefix + "1", kv["filename1"], match, mismatch),
                                    new ReadSet(cache_prefix + "2", kv["filename2"], match, mismatch)}});
    } else {
      fprintf(stderr, "Unknown type %s for read set %s, ignoring...\n", kv["type"].c_str(), e.first.c_str());
    }
  }
}

// gaml.cc:30,84: where the BLASR binary lives (config key blasr_path)
inline string& BlasrPath() { static string p = "blasr/alignment/bin"; return p; }

// ---- ProbCalculator (prob_calculator.h:37-124) --------------------------------------------------
class ProbCalculator {
 public:
  // device list from the environment (GAML_HIP_DEVICES=all | 0,1,...; unset: device 0): several devices = the reads
  // sharded over them inside the library, one RCCL all-reduce per CalcProb (gaml_hip_create_from_env)
  static constexpr int kDevicesFromEnv = -2;
  ProbCalculator(const vector<pair<SingleReadConfig, ReadSet*>>& single_reads,
                 const vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>>& paired_reads,
                 const vector<pair<SingleReadConfig, PacbioReadSet*>>& pacbio_reads, Graph& gr, int device = kDevicesFromEnv)
      : single_reads(single_reads), paired_reads(paired_reads), pacbio_reads(pacbio_reads), gr(gr), device_(device) {}
  ~ProbCalculator() { if (ctx_) gaml_hip_destroy(ctx_); }

  double CalcProb(vector<vector<int>>& paths, vector<pair<int, int>>& zeros, int& total_len) {
    if (!ctx_ && !Build()) { fprintf(stderr, "gaml_hip: %s\n", err_.c_str()); exit(1); }
    if (!FillPacbioCache(paths)) { fprintf(stderr, "gaml_hip: %s\n", err_.c_str()); exit(1); }
    vector<int32_t> flat;
    vector<int64_t> offs(1, 0);
    for (auto& p : paths) { flat.insert(flat.end(), p.begin(), p.end()); offs.push_back((int64_t)flat.size()); }
    int32_t dummy = 0;
    double prob = 0;
    vector<int32_t> z(2 * (single_reads.size() + paired_reads.size() + pacbio_reads.size()) + 2);
    int32_t tl = 0;
    int rc = gaml_hip_calc_prob(ctx_, flat.empty() ? &dummy : flat.data(), offs.data(), (int32_t)paths.size(), &prob, z.data(), &tl);
    if (rc != GAML_HIP_OK) { fprintf(stderr, "gaml_hip_calc_prob: %s\n", gaml_hip_last_error(ctx_)); exit(1); }
    zeros.clear();
    size_t n = single_reads.size() + paired_reads.size() + pacbio_reads.size();
    for (size_t i = 0; i < n; i++) zeros.push_back({z[2 * i], z[2 * i + 1]});
    total_len = tl;
    return prob;
  }
  double CalcProb(vector<vector<int>>& paths, int& total_len) { vector<pair<int, int>> zeros; return CalcProb(paths, zeros, total_len); }
  double CalcProb(vector<vector<int>>& paths) { int tl; return CalcProb(paths, tl); }

  vector<pair<SingleReadConfig, ReadSet*>> single_reads;
  vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>> paired_reads;
  vector<pair<SingleReadConfig, PacbioReadSet*>> pacbio_reads;
  Graph& gr;
  gaml_hip_ctx* context() { return ctx_; }

 private:
  // built lazily on the first CalcProb: the reference constructs ProbCalculator BEFORE the reads are
  // prepared (gaml.cc:1010 vs 1017)
  bool Build() {
    const int rc = device_ == kDevicesFromEnv ? gaml_hip_create_from_env(&ctx_) : gaml_hip_create(&ctx_, device_);
    if (rc != GAML_HIP_OK) { ctx_ = nullptr; err_ = "no HIP device (the likelihood path has no CPU implementation)"; return false; }
    string bases;
    vector<int64_t> offs(1, 0);
    for (auto& s : gr.nodes) { bases += s; offs.push_back((int64_t)bases.size()); }
    if (gaml_hip_set_graph(ctx_, (int32_t)gr.nodes.size(), bases.data(), offs.data())) return Fail();
    for (auto& e : single_reads) {
      gaml_single_cfg c{e.first.penalty_constant, e.first.step, e.first.min_prob_per_base, e.first.min_prob_start, e.first.weight, e.second->mismatch_prob_};
      int h = gaml_hip_add_single_fastq(ctx_, &c, e.second->filename().c_str());
      if (h < 0) return Fail();
      e.second->reads_num_ = (int)gaml_hip_readset_reads(ctx_, h);
    }
    for (auto& e : paired_reads) {
      gaml_paired_cfg c{e.first.penalty_constant, e.first.step, e.first.insert_mean, e.first.insert_std, e.first.min_prob_per_base,
                        e.first.min_prob_start, e.first.weight, e.second.first->mismatch_prob_};
      int h = gaml_hip_add_paired_fastq(ctx_, &c, e.second.first->filename().c_str(), e.second.second->filename().c_str());
      if (h < 0) return Fail();
      e.second.first->reads_num_ = e.second.second->reads_num_ = (int)gaml_hip_readset_reads(ctx_, h);
    }
    for (auto& e : pacbio_reads) {
      gaml_single_cfg c{e.first.penalty_constant, e.first.step, e.first.min_prob_per_base, e.first.min_prob_start, e.first.weight, e.second->mismatch_prob_};
      int h = gaml_hip_add_pacbio_fastq(ctx_, &c, e.second->filename().c_str());
      if (h < 0) return Fail();
      pacbio_handles_.push_back(h);
      e.second->reads_num_ = (int)gaml_hip_readset_reads(ctx_, h);
      if (!e.second->records_file.empty() && !LoadPacbioRecords(h, e.second->records_file)) return false;
    }
    return true;
  }
  bool LoadPacbioRecords(int h, const string& file) {
    std::ifstream f(file);
    if (!f.is_open()) { err_ = "cannot open " + file; return false; }
    int n;
    while (f >> n) {
      vector<int32_t> walk(n);
      for (auto& x : walk) f >> x;
      gaml_pacbio_aligment r{};
      f >> r.position >> r.position_end >> r.read_id >> r.logprob;
      if (gaml_hip_put_pacbio_records(ctx_, h, walk.data(), n, &r, 1)) return Fail();
    }
    return true;
  }
  // The cache-miss side of the PacBio scorer (GetReadProbabilities graph.cc:2438-2478 ->
  // GetReadProbabilitiesSlow :2650-2795): every stretch of a path with an uncached sub-walk is
  // written out, BLASR aligns the reads to it, and the SAM lines become cached records. The
  // library says which stretches (gaml_hip_pacbio_missing) and turns SAM text into records
  // (gaml_hip_pacbio_ingest_sam: ParseAligment + AligmentProbability on the GPU + filing rule);
  // running the external aligner stays here, with the reference's command line (:2705-2715).
  bool FillPacbioCache(const vector<vector<int>>& paths) {
    for (size_t k = 0; k < pacbio_handles_.size(); k++) {
      const int h = pacbio_handles_[k];
      for (auto& path : paths) {
        if (path.empty()) continue;
        vector<int32_t> p(path.begin(), path.end()), ranges(2 * path.size() + 2);
        int32_t n = gaml_hip_pacbio_missing(ctx_, h, p.data(), (int32_t)p.size(), ranges.data(), (int32_t)ranges.size() / 2);
        if (n < 0) return Fail();
        for (int32_t r = 0; r < n; r++) {
          vector<int32_t> sub(p.begin() + ranges[2 * r], p.begin() + ranges[2 * r + 1] + 1);
          string sam;
          if (!RunBlasr(pacbio_reads[k].second->filename(), sub, sam)) return false;
          int64_t filed = 0;
          if (gaml_hip_pacbio_ingest_sam(ctx_, h, sub.data(), (int32_t)sub.size(), sam.data(), (int64_t)sam.size(), &filed)) return Fail();
          printf("pb slow: %d nodes, %lld records filed\n", (int)sub.size(), (long long)filed);
        }
      }
    }
    return true;
  }
  bool RunBlasr(const string& reads_file, const vector<int32_t>& sub, string& sam) {
    // (the reference uses tmpnam, graph.cc:2653-2658; a private directory avoids its race)
    char dir[] = "/tmp/gaml_hip_pbXXXXXX";
    if (!mkdtemp(dir)) { err_ = "mkdtemp failed"; return false; }
    const string fas = string(dir) + "/path.fas", out = string(dir) + "/blasr.sam";
    FILE* f = fopen(fas.c_str(), "w");
    if (!f) { err_ = "cannot write " + fas; rmdir(dir); return false; }
    fprintf(f, ">tmp\n");
    for (int32_t x : sub) {
      if (x < 0) for (int j = 0; j < -x; j++) fputc('N', f);
      else fputs(gr.nodes[x].c_str(), f);
    }
    fputc('\n', f);
    fclose(f);
    string cmd = BlasrPath() + "/blasr " + reads_file + " " + fas +
                 " -sam -sdpTupleSize 8 -guidedAlignBandSize 100 -nCandidates 50 -minMatch 11 -nproc 16 >" + out;
    printf("command %s\n", cmd.c_str());
    int rc = system(cmd.c_str());
    std::ifstream fi(out);
    bool ok = rc == 0 && fi.is_open();
    if (ok) sam.assign(std::istreambuf_iterator<char>(fi), std::istreambuf_iterator<char>());
    else err_ = "the aligner command failed: " + cmd;
    fi.close();
    remove(fas.c_str());
    remove(out.c_str());
    rmdir(dir);
    return ok;
  }
  bool Fail() { err_ = gaml_hip_last_error(ctx_); return false; }
  vector<int> pacbio_handles_;
  gaml_hip_ctx* ctx_ = nullptr;
  int device_;
  string err_;
};

// walks file as the reference writes it (Graph::OutputPathC graph.cc:277-291):
// ">tmp<cid>-<node>(<pos>)-<node>(<pos>)..." one walk per line; gap entries are negative numbers
inline bool LoadWalks(const string& file, vector<vector<int>>& paths) {
  std::ifstream f(file);
  if (!f.is_open()) return false;
  string l;
  while (std::getline(f, l)) {
    if (l.empty() || l[0] != '>') continue;
    size_t p = l.find('-');
    vector<int> w;
    while (p != string::npos && p + 1 < l.size()) {
      size_t open = l.find('(', p + 1);
      if (open == string::npos) break;
      w.push_back(atoi(l.substr(p + 1, open - p - 1).c_str()));
      size_t close = l.find(')', open);
      if (close == string::npos || close + 1 >= l.size()) break;
      p = close + 1;  // the '-' separator
    }
    paths.push_back(w);
  }
  return true;
}

}  // namespace gaml_host
