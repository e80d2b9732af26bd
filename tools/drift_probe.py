"""bench.py's incremental_drift block on its own.  python tools/drift_probe.py [iterations] [sample pairs]"""
import json, os, sys
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS["cfg3"]
genome, g = wl.build()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
sample = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
b1, o1 = synth.pack_reads(pr.mate1); b2, o2 = synth.pack_reads(pr.mate2)
print(json.dumps(bench.drift_block(api, synth, 0, g, b1, o1, b2, o2, wl.read_len, (wl.insert_mean, wl.insert_std), iters, sample), indent=1))
