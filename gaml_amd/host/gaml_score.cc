// gaml_score -- score a set of walks with the reference's config format, the way gaml.cc's main
// does up to the first CalcProb (gaml.cc:935-1022, 91-110), on the GPU.
//   gaml_score <config file> [<walks file>]
// Without a walks file the starting state of the reference is scored: every even node longer than
// long_contig_threshold as a one-node walk (gaml.cc:1002-1005).
#include <cstdio>

#include "gaml_host.h"

using namespace gaml_host;

int main(int argc, char** argv) {
  if (argc < 2) { printf("Missing config file!\nSyntax:\n./gaml_score <config file> [<walks file>]\n"); return 1; }
  unordered_map<string, string> configs;
  unordered_map<string, unordered_map<string, string>> read_set_configs;
  if (!LoadConfig(argv[1], configs, read_set_configs)) { printf("Load config failed\n"); return 1; }
  if (!configs.count("graph")) { fprintf(stderr, "Missing graph in config\n"); return 1; }
  if (configs.count("blasr_path")) BlasrPath() = configs["blasr_path"];  // gaml.cc:84
  vector<pair<SingleReadConfig, ReadSet*>> single_reads;
  vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*>>> paired_reads;
  vector<pair<SingleReadConfig, PacbioReadSet*>> pacbio_reads;
  PrepareReadSetFromConfig(read_set_configs, single_reads, paired_reads, pacbio_reads);
  Graph gr;
  if (!LoadGraph(configs["graph"], gr)) { printf("Load graph failed\n"); return 1; }
  vector<vector<int>> paths;
  if (argc > 2) {
    if (!LoadWalks(argv[2], paths)) { printf("Load walks failed\n"); return 1; }
  } else {
    int threshold = configs.count("long_contig_threshold") ? atoi(configs["long_contig_threshold"].c_str()) : 500;
    for (int i = 0; i < (int)gr.nodes.size(); i += 2)
      if ((int)gr.nodes[i].size() > threshold) paths.push_back(vector<int>({i}));
  }
  ProbCalculator pc(single_reads, paired_reads, pacbio_reads, gr);
  vector<pair<int, int>> zeros;
  int total_len = 0;
  double prob = pc.CalcProb(paths, zeros, total_len);
  printf("gaml_hip: %d device shard(s), exchange %s\n", gaml_hip_num_shards(pc.context()),
         gaml_hip_num_shards(pc.context()) > 1 ? (gaml_hip_get_exchange(pc.context()) == GAML_HIP_EXCHANGE_RCCL ? "rccl" : "host") : "none");
  printf("start prob %.17g len %d low prob reads", prob, total_len);  // gaml.cc:106-110 (more digits)
  for (auto& e : zeros) printf("%d/%d ", e.first, e.second);
  printf("\n");
  return 0;
}
