#!/bin/bash
# What bounds paired_score_kernel at BASELINE config 3 -- every probe in ONE run on ONE box, from the sources of this tree:
#   bash tools/floor_probes.sh > gpurun_out/<tag>_floor_probes.txt      (copied into profiles/ afterwards)
# 1. the empty grid: a kernel of the scoring launch's shape whose waves do nothing / spin 5 us (tools/xcd_start_probe.hip)
# 2. the pure stream of the compact class's bytes with the scoring launch's fixed costs added one by one (tools/stream_floor2.hip)
# 3. the scoring kernel with parts switched off (timing builds, results wrong; tools/build_variant.sh -DGAML_STATIC_X=...):
#      x15 static class only, records + arithmetic only (no occurrence lookups, no values, no stores): its floor
#      x8  static class only, complete                 x16 everything BUT the static class
# 4. the complete kernel (this tree's development build) and the round-2 kernel (build_ab/libgaml_hip_r02.so) beside it
# 5. in-kernel stage stamps per class (tools/kernel_timeline.py: the instantiation with wall-clock stamps)
cd "$(dirname "$0")/.."
echo "== source hash: $(python3 -c 'import bench; print(bench.source_hash())')"
for p in xcd_start_probe stream_floor2; do
  if [ ! -x build_ab/$p ]; then /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o build_ab/$p tools/$p.hip || exit 1; fi
done
echo; echo "== 1. empty grid / spinning grid (962 x 256 threads), events attached to the dispatch"
./build_ab/xcd_start_probe 2>&1 | grep -v amdgpu.ids
echo; echo "== 2. pure stream + fixed costs"
./build_ab/stream_floor2 2>&1 | grep -v amdgpu.ids
echo; echo "== 3 / 4. scoring kernel, parts switched off (launch = events attached to every dispatch, 800 launches, 8 rotating path sets)"
for L in r02 main x15 x8 x16 main r02; do
  LIB=build_ab/libgaml_hip_$L.so; [ $L = main ] && LIB=gaml_amd/libgaml_hip_dev.so
  [ -f $LIB ] || { echo "($LIB missing)"; continue; }
  printf "%-5s " $L; GAML_HIP_LIB=$LIB SWEEP="[{}]" python3 tools/static_ab.py 2>&1 | grep -v amdgpu.ids | cut -c1-9,118-
done
echo; echo "== 5. in-kernel stage stamps (development build, kernel instantiation with stamps: slower than the product kernel)"
python3 tools/kernel_timeline.py cfg3 2>&1 | grep -v amdgpu.ids | grep -A24 "evaluation 2"
