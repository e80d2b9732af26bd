// tests/mock_ref/adapter_driver.cc -- compiles include/gaml_hip_prob_calculator.h (the drop-in replacement for the
// reference's prob_calculator.h) against the declaration mock in this directory, constructs a ProbCalculator the way
// gaml.cc:1010 does and calls all three CalcProb overloads (prob_calculator.h:63-118).
//   adapter_driver <LastGraph> <fastq1> <fastq2> <insert_mean> <insert_std> [<single fastq>]
// Scores the genome as one walk (all even nodes in order) and as its two halves; prints one line per call.
#include <cstdio>
#include <cstdlib>

#include "gaml_hip_prob_calculator.h"

string gBlasrPath = "blasr/alignment/bin";  // gaml.cc:30

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: adapter_driver LastGraph fq1 fq2 mean std [single.fq]\n"); return 2; }
  Graph gr;
  if (!gr.Load(argv[1])) { fprintf(stderr, "cannot load %s\n", argv[1]); return 1; }
  const double mean = atof(argv[4]), sd = atof(argv[5]);
  vector<pair<SingleReadConfig, ReadSet*> > single_reads;
  vector<pair<PairedReadConfig, pair<ReadSet*, ReadSet*> > > paired_reads;
  vector<pair<SingleReadConfig, PacbioReadSet*> > pacbio_reads;
  // gaml.cc:851-864: penalty_constant 0, step = insert_mean - penalty_step(50), min_prob_pre_base -0.7, min_prob_start -10
  paired_reads.push_back(make_pair(PairedReadConfig(0, mean - 50, mean, sd, -0.7, -10, 1, false),
                                   make_pair(new ReadSet("a1", argv[2], 0.96, 0.01), new ReadSet("a2", argv[3], 0.96, 0.01))));
  if (argc > 6) single_reads.push_back(make_pair(SingleReadConfig(0, 50, -0.7, -10, 0.5, false), new ReadSet("s", argv[6], 0.96, 0.01)));
  ProbCalculator pc(single_reads, paired_reads, pacbio_reads, gr);
  vector<vector<int> > whole(1), halves(2);
  for (int i = 0; i < (int)gr.nodes.size(); i += 2) whole[0].push_back(i);
  for (int i = 0; i < (int)whole[0].size(); i++) halves[i < (int)whole[0].size() / 2 ? 0 : 1].push_back(whole[0][i]);
  vector<pair<int, int> > zeros;
  int tl = 0;
  double p = pc.CalcProb(whole, zeros, tl);
  printf("whole %.17g len %d zeros", p, tl);
  for (size_t i = 0; i < zeros.size(); i++) printf(" %d/%d", zeros[i].first, zeros[i].second);
  printf("\n");
  int tl2 = 0;
  printf("halves %.17g", pc.CalcProb(halves, tl2));
  printf(" len %d\n", tl2);
  printf("whole_again %.17g\n", pc.CalcProb(whole));
  return 0;
}
