#!/bin/bash
# Everything profiles/<tag>_* holds, in one run ON THE GPU BOX from the repository root:
#   bash tools/round_evidence.sh r02c
# 1. tools/profile_round.sh (kernel trace + the two PMC passes) and its summaries into profiles/ -- BEFORE the bench, so
#    that bench.py finds a PMC summary whose source hash matches and attaches `traffic`
# 2. the default bench.py line                                  -> <tag>_bench.json
# 3. the N > 1 code path with one rank (in-library RCCL)        -> <tag>_bench_force_rccl_1rank.json
# 4. two ranks sharing the one GPU (host shared-memory exchange) -> <tag>_bench_n2_one_gpu_rehearsal.json
# 5. tools/config_report.py: all five BASELINE configurations          -> <tag>_configs.jsonl
# Copies land in gpurun_out/evidence_<tag>/ as well (profiles/ on the box does not travel back).
set -u
tag=${1:-r02}
ev=gpurun_out/evidence_$tag
mkdir -p "$ev"
bash tools/profile_round.sh "$tag" || exit 1
cp "gpurun_out/prof_$tag/${tag}_pmc_traffic.json" "gpurun_out/prof_$tag/${tag}_kernel_stats.csv" profiles/ || exit 1
cp "gpurun_out/prof_$tag/${tag}_pmc_traffic.json" "gpurun_out/prof_$tag/${tag}_kernel_stats.csv" "gpurun_out/prof_$tag/${tag}_headline_kernel_stats.csv" "$ev/"
# the HBM-bound size: kernel stats, PMC traffic and the bench line of --workload cfg3x8 on the same sources
cp gpurun_out/prof_$tag/${tag}_cfg3x8_* "$ev/" 2> /dev/null
echo "== bench" && python3 bench.py > "$ev/${tag}_bench.json" 2> "$ev/bench.err" || exit 1
echo "== one rank through the communicator" && python3 bench.py --force-dist --no-cpu-baseline > "$ev/${tag}_bench_force_rccl_1rank.json" 2> "$ev/force.err" || exit 1
echo "== two ranks on the one GPU" && GAML_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29577 bench.py --gpus 2 --no-inproc > "$ev/${tag}_bench_n2_one_gpu_rehearsal.json" 2> "$ev/n2.err" || exit 1
echo "== all five BASELINE configurations against the oracle" && python3 tools/config_report.py > "$ev/${tag}_configs.jsonl" 2> "$ev/configs.err" || exit 1
ls -l "$ev"
