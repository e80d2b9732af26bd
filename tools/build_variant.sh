#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...]: an A/B build of the library into build_ab/libgaml_hip_NAME.so (load it with
# GAML_HIP_LIB=...), with the register / scratch use of the paired scoring kernels printed: a variant that spills to
# scratch is not worth a GPU run.
set -e
NAME=$1; shift
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/gaml_amd/csrc
T=/tmp/gaml_variant_$NAME; mkdir -p $T $HERE/build_ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-sign-compare"
# (variants are development builds: the probes that load them use knobs and debug entry points)
(cd $SRC && /opt/rocm/bin/hipcc $FL -DGAML_HIP_DEV "$@" -Rpass-analysis=kernel-resource-usage -c gaml_hip.hip -o $T/gaml_hip.o > $T/usage.txt 2>&1) || { tail -20 $T/usage.txt; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $HERE/build_ab/libgaml_hip_$NAME.so $SRC/obj_dev/host_model.o $T/gaml_hip.o $SRC/obj_dev/multi.o $SRC/obj_dev/version.o -ldl -lpthread
python3 - $T/usage.txt $NAME <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for b in re.split(r"remark: Function Name: ", t)[1:]:
    name = b.split()[0]
    if "paired_score_kernel" not in name: continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    print(sys.argv[2], name[22:60], "VGPR", g("VGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "spillV", g("VGPRs Spill"), "spillS", g("SGPRs Spill"))
PY
