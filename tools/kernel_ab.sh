#!/bin/bash
# A/B of build_ab/ variants of the library (tools/build_variant.sh NAME -D...) on ONE box: launch duration (events attached
# to every dispatch) and step time over the bench's 8 rotating path sets.   bash tools/kernel_ab.sh base cmp1 cmp0 base
cd "$(dirname "$0")/.."
DEFAULT_SWEEP="[{}]"
for L in "$@"; do
  printf "%-6s " $L
  GAML_HIP_LIB=$PWD/build_ab/libgaml_hip_$L.so SWEEP="${SWEEP:-$DEFAULT_SWEEP}" timeout -k 10 200 python3 tools/static_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-30,118- || exit 1
done
