#!/bin/bash
# Profiles of one round, run ON THE GPU BOX from the repository root:
#   bash tools/profile_round.sh r01d
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/trace
# 2. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with traces) -> .../fetch, .../write
# Summaries are made by tools/pmc_summary.py (copied into profiles/ by hand afterwards).
set -u
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
echo "== kernel trace" > "$out/log.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py --no-cpu-baseline --no-sa >> "$out/log.txt" 2>&1
echo "== pmc FETCH_SIZE" >> "$out/log.txt"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- python3 bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
echo "== pmc WRITE_SIZE" >> "$out/log.txt"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- python3 bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 24 >> "$out/log.txt" 2>&1
python3 tools/pmc_summary.py "$tag" >> "$out/log.txt" 2>&1
ls -R "$out" | grep -c csv
