// gaml_hip_prob_calculator.h -- drop-in replacement for the reference's prob_calculator.h.
//
// Same struct names and fields, same constructor, same three CalcProb overloads, same public
// data members (reference prob_calculator.h:7-124); gaml.cc and moves.cc compile against it
// unchanged. CalcProb forwards to libgaml_hip.so (include/gaml_hip.h) instead of calling
// CalcScoreForPaths / CalcScoreForPathsNew / CalcScoreForPacbio (graph.cc:1650, 1952, 3171).
//
// The one change outside this file: ReadSet and PacbioReadSet (graph.h:344, 444) keep their FASTQ
// path in a private member, so add
//       friend class ProbCalculator;
// to both classes (INTEGRATION.md shows the two-line patch). Nothing else in GAML is touched.
//
// Several GPUs: the device list comes from the environment (GAML_HIP_DEVICES=all | 0,1,2,...; unset: device 0) --
// gaml_hip_create_from_env. With more than one device the context shards the reads over them inside the library
// (one host thread per device, one RCCL all-reduce of the per-readset sums per CalcProb); this header does not change.
//
// Compile-tested: tests/test_gpu_adapter.py builds tests/mock_ref/adapter_driver.cc, which includes THIS header over a
// test-only declaration mock of the graph.h members used below (tests/mock_ref/graph.h), and runs all three CalcProb
// overloads against the ctypes value. The reference's own graph.h needs Boost, which the image lacks (DESIGN.md
// "Oracle"), so the real gaml.cc is not built here; gaml_amd/host/gaml_host.h is the same logic over stand-alone
// mirrors of these classes.
#ifndef PROB_CALCULATOR_H__
#define PROB_CALCULATOR_H__

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <fstream>
#include <iterator>

#include "gaml_hip.h"
#include "graph.h"    // the reference's
#include "utility.h"  // the reference's

struct SingleReadConfig {
  SingleReadConfig() {}
  SingleReadConfig(double pc, double s, double mp, double mps, double w, bool a)
      : penalty_constant(pc), step(s), min_prob_per_base(mp), min_prob_start(mps), weight(w), advice(a) {}
  double penalty_constant;
  double step;
  double min_prob_per_base;
  double min_prob_start;
  double weight;
  bool advice;
};

struct PairedReadConfig {
  PairedReadConfig() {}
  PairedReadConfig(double pc, double s, double im, double is, double mp, double mps, double w, bool a)
      : penalty_constant(pc), step(s), insert_mean(im), insert_std(is), min_prob_per_base(mp), min_prob_start(mps),
        weight(w), advice(a) {}
  double penalty_constant;
  double step;
  double insert_mean;
  double insert_std;
  double min_prob_per_base;
  double min_prob_start;
  double weight;
  bool advice;
};

class ProbCalculator {
 public:
  ProbCalculator(const vector<pair<SingleReadConfig, ReadSet*>>& single_reads,
                 const vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>>& paired_reads,
                 const vector<pair<SingleReadConfig, PacbioReadSet*>>& pacbio_reads, Graph& gr)
      : single_reads(single_reads), paired_reads(paired_reads), pacbio_reads(pacbio_reads), gr(gr), ctx_(NULL) {
    paired_scoring_states.resize(paired_reads.size());  // kept for source compatibility; unused
  }
  ~ProbCalculator() { if (ctx_) gaml_hip_destroy(ctx_); }

  double CalcProb(vector<vector<int>>& pathso, vector<pair<int, int>>& zeros, int& total_len) {
    // the reference constructs ProbCalculator before PrepareReads (gaml.cc:1010 vs 1017): the
    // device context is built on first use, when the FASTQ files are known to be final
    if (!ctx_) Build();
    FillPacbioCache(pathso);
    vector<int32_t> flat;
    vector<int64_t> offs(1, 0);
    for (size_t i = 0; i < pathso.size(); i++) {
      flat.insert(flat.end(), pathso[i].begin(), pathso[i].end());
      offs.push_back((int64_t)flat.size());
    }
    int32_t none = 0, tl = 0;
    double prob = 0;
    vector<int32_t> z(2 * (single_reads.size() + paired_reads.size() + pacbio_reads.size()) + 2);
    if (gaml_hip_calc_prob(ctx_, flat.empty() ? &none : &flat[0], &offs[0], (int32_t)pathso.size(), &prob, &z[0], &tl) != GAML_HIP_OK)
      Die("gaml_hip_calc_prob");
    zeros.clear();
    for (size_t i = 0; i < single_reads.size() + paired_reads.size() + pacbio_reads.size(); i++)
      zeros.push_back(make_pair(z[2 * i], z[2 * i + 1]));
    total_len = tl;
    return prob;
  }
  double CalcProb(vector<vector<int> >& paths, int& total_len) {
    vector<pair<int, int>> zeros;
    return CalcProb(paths, zeros, total_len);
  }
  double CalcProb(vector<vector<int> >& paths) {
    int tl;
    return CalcProb(paths, tl);
  }

  vector<pair<SingleReadConfig, ReadSet*>> single_reads;
  vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>> paired_reads;
  vector<pair<SingleReadConfig, PacbioReadSet*>> pacbio_reads;
  vector<ScoringState> paired_scoring_states;
  Graph& gr;

 private:
  void Die(const char* what) {
    fprintf(stderr, "%s failed: %s\n", what, ctx_ ? gaml_hip_last_error(ctx_) : "no context");
    exit(1);  // the reference's convention for unrecoverable errors (assert / exit code)
  }
  void Build() {
    if (gaml_hip_create_from_env(&ctx_) != GAML_HIP_OK) { ctx_ = NULL; Die("gaml_hip_create_from_env (no HIP device? GAML_HIP_DEVICES?)"); }
    string bases;
    vector<int64_t> offs(1, 0);
    for (size_t i = 0; i < gr.nodes.size(); i++) { bases += gr.nodes[i]->s; offs.push_back((int64_t)bases.size()); }
    if (gaml_hip_set_graph(ctx_, (int32_t)gr.nodes.size(), bases.data(), &offs[0])) Die("gaml_hip_set_graph");
    for (size_t i = 0; i < single_reads.size(); i++) {
      const SingleReadConfig& c = single_reads[i].first;
      ReadSet* rs = single_reads[i].second;
      gaml_single_cfg g = {c.penalty_constant, c.step, c.min_prob_per_base, c.min_prob_start, c.weight, rs->mismatch_prob_};
      if (gaml_hip_add_single_fastq(ctx_, &g, rs->filename_.c_str()) < 0) Die("gaml_hip_add_single_fastq");  // friend access
    }
    for (size_t i = 0; i < paired_reads.size(); i++) {
      const PairedReadConfig& c = paired_reads[i].first;
      ReadSet* r1 = paired_reads[i].second.first;
      ReadSet* r2 = paired_reads[i].second.second;
      gaml_paired_cfg g = {c.penalty_constant, c.step, c.insert_mean, c.insert_std, c.min_prob_per_base, c.min_prob_start, c.weight,
                           r1->mismatch_prob_};
      if (gaml_hip_add_paired_fastq(ctx_, &g, r1->filename_.c_str(), r2->filename_.c_str()) < 0) Die("gaml_hip_add_paired_fastq");
    }
    for (size_t i = 0; i < pacbio_reads.size(); i++) {
      const SingleReadConfig& c = pacbio_reads[i].first;
      PacbioReadSet* rs = pacbio_reads[i].second;
      gaml_single_cfg g = {c.penalty_constant, c.step, c.min_prob_per_base, c.min_prob_start, c.weight, exp(rs->mismatch_prob_.logval)};
      int h = gaml_hip_add_pacbio_fastq(ctx_, &g, rs->filename_.c_str());
      if (h < 0) Die("gaml_hip_add_pacbio_fastq");
      pacbio_handles_.push_back(h);
      // hand over what BLASR + AligmentProbability already left in the PacBio cache (graph.h:587)
      for (auto it = rs->aligment_cache_.begin(); it != rs->aligment_cache_.end(); ++it) {
        vector<gaml_pacbio_aligment> recs;
        for (size_t k = 0; k < it->second.size(); k++) {
          gaml_pacbio_aligment r = {it->second[k].position, it->second[k].position_end, it->second[k].read_id, 0, it->second[k].prob.logval};
          recs.push_back(r);
        }
        vector<int32_t> walk(it->first.begin(), it->first.end());
        if (gaml_hip_put_pacbio_records(ctx_, h, &walk[0], (int32_t)walk.size(), recs.empty() ? NULL : &recs[0], (int64_t)recs.size()))
          Die("gaml_hip_put_pacbio_records");
      }
    }
  }
  // Cache-miss side of the PacBio scorer. The reference's GetReadProbabilities (graph.cc:2438-2478)
  // hands every stretch of a path with an uncached sub-walk to GetReadProbabilitiesSlow
  // (:2650-2795), which writes the stretch to a file, runs BLASR, and turns each SAM line into a
  // cached record on the CPU. Here the library names the stretches (gaml_hip_pacbio_missing) and
  // turns the SAM text into records (gaml_hip_pacbio_ingest_sam: ParseAligment, AligmentProbability
  // on the GPU, filing rule); running BLASR stays here, with the reference's files, read filter
  // (anchors) and command line (:2652-2715).
  void FillPacbioCache(const vector<vector<int> >& paths) {
    for (size_t k = 0; k < pacbio_handles_.size(); k++) {
      PacbioReadSet* rs = pacbio_reads[k].second;
      for (size_t pi = 0; pi < paths.size(); pi++) {
        if (paths[pi].empty()) continue;
        vector<int32_t> p(paths[pi].begin(), paths[pi].end()), ranges(2 * p.size() + 2);
        int32_t n = gaml_hip_pacbio_missing(ctx_, pacbio_handles_[k], &p[0], (int32_t)p.size(), &ranges[0], (int32_t)ranges.size() / 2);
        if (n < 0) Die("gaml_hip_pacbio_missing");
        for (int32_t r = 0; r < n; r++) {
          vector<int32_t> sub(p.begin() + ranges[2 * r], p.begin() + ranges[2 * r + 1] + 1);
          string sam = RunBlasr(rs, sub);
          int64_t filed = 0;
          if (gaml_hip_pacbio_ingest_sam(ctx_, pacbio_handles_[k], &sub[0], (int32_t)sub.size(), sam.data(), (int64_t)sam.size(), &filed))
            Die("gaml_hip_pacbio_ingest_sam");
        }
      }
    }
  }
  string RunBlasr(PacbioReadSet* rs, const vector<int32_t>& sub) {
    extern string gBlasrPath;  // gaml.cc:30
    // (the reference uses tmpnam, graph.cc:2653-2658; a private directory avoids its race and is removed whole)
    char dir[] = "/tmp/gaml_hip_pbXXXXXX";
    if (!mkdtemp(dir)) Die("mkdtemp");
    const string fas = string(dir) + "/path.fas", fq = string(dir) + "/reads.fq", out = string(dir) + "/blasr.sam";
    FILE* f = fopen(fas.c_str(), "w");
    if (!f) Die("fopen of the path file");
    fprintf(f, ">tmp\n");
    for (size_t i = 0; i < sub.size(); i++) {
      if (sub[i] < 0) for (int j = 0; j < -sub[i]; j++) fputc('N', f);
      else fputs(gr.nodes[sub[i]]->s.c_str(), f);
    }
    fputc('\n', f);
    fclose(f);
    string reads_filename = rs->filename_;
    unordered_set<int> read_filter;  // reads anchored on the stretch's nodes (graph.cc:2690-2701)
    for (size_t i = 0; i < sub.size(); i++)
      if (sub[i] >= 0 && rs->anchors_cache_.count(sub[i]))
        for (auto it = rs->anchors_cache_[sub[i]].begin(); it != rs->anchors_cache_[sub[i]].end(); ++it) read_filter.insert(*it);
    if (!read_filter.empty()) { rs->FilterReads(fq, read_filter); reads_filename = fq; }
    string cmd = gBlasrPath + "/blasr " + reads_filename + " " + fas +
                 " -sam -sdpTupleSize 8 -guidedAlignBandSize 100 -nCandidates 50 -minMatch 11 -nproc 16 >" + out;
    const int rc = system(cmd.c_str());
    string sam;
    {
      ifstream fi(out.c_str());
      sam.assign((std::istreambuf_iterator<char>(fi)), std::istreambuf_iterator<char>());
    }
    remove(fas.c_str());
    remove(fq.c_str());
    remove(out.c_str());
    rmdir(dir);
    if (rc != 0) Die("blasr");
    return sam;
  }
  vector<int> pacbio_handles_;
  gaml_hip_ctx* ctx_;
};

#endif
