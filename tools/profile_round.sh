#!/bin/bash
# Profiles of one round, run ON THE GPU BOX from the repository root:
#   bash tools/profile_round.sh r04a
# 1. rocprofv3 --kernel-trace --stats of `bench.py --no-cpu-baseline`: the headline steps AND the annealing pattern (small-batch
#    aligner, filing, delta maintenance), the repeat-rich block (the GEN instantiation of paired_score_kernel), the jumping library (coverage sweep),
#    the aligner block and the long annealing run (table builds beside the evaluations) -- every kernel of the path gets a row
# 1b. the same trace of the headline steps alone (`--no-extras`: 2,100 launches of the one workload `roofline` is quoted on): the
#    average duration of paired_score_kernel there is the figure bench.py's live `roofline.kernel_us` has to agree with
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with traces) of the headline steps alone
# 3. the same three for --workload cfg3x8 (the one size at which the scoring launch is HBM-bound: 295 MB > the Infinity Cache)
# Summaries are made by tools/pmc_summary.py (copied into profiles/ by tools/round_evidence.sh).
set -u
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "== kernel trace" > "$out/log.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py --no-cpu-baseline >> "$out/log.txt" 2>&1
echo "== kernel trace of the headline steps alone" >> "$out/log.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/headline" -o headline -- python3 bench.py --no-cpu-baseline --no-extras >> "$out/log.txt" 2>&1
cp "$out/headline/headline_kernel_stats.csv" "$out/${tag}_headline_kernel_stats.csv"
echo "== pmc FETCH_SIZE" >> "$out/log.txt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- python3 bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
echo "== pmc WRITE_SIZE" >> "$out/log.txt"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- python3 bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
python3 tools/pmc_summary.py "$tag" >> "$out/log.txt" 2>&1
if [ "${GAML_PROFILE_X8:-1}" = 1 ]; then
  x8="--workload cfg3x8 --no-cpu-baseline --no-sa --no-repeats --no-long"
  echo "== cfg3x8: kernel trace" >> "$out/log.txt"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/x8_trace" -o trace -- python3 bench.py $x8 --steps 300 --warmup 30 > "$out/${tag}_cfg3x8_bench.json" 2>> "$out/log.txt"
  echo "== cfg3x8: pmc FETCH_SIZE" >> "$out/log.txt"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/x8_fetch" -o fetch -- python3 bench.py $x8 --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
  echo "== cfg3x8: pmc WRITE_SIZE" >> "$out/log.txt"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/x8_write" -o write -- python3 bench.py $x8 --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
  python3 tools/pmc_summary.py "$tag" cfg3x8 >> "$out/log.txt" 2>&1
fi
ls -R "$out" | grep -c csv
