"""Where one slow call of the annealing pattern spends its time: aligner stage split before / after the call.
python tools/sa_call_probe.py 20"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
which = [int(a) for a in sys.argv[1:]] or [20]
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, max(which) + 1)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *reads)
ctx.debug_set_knob(9, 1)
ctx.calc_prob(start)
os.environ["GAML_HIP_TRACE_ALIGNER"] = "1"
for k, p in enumerate(seq):
    f = api.FlatPaths(p)
    if k in which:
        print(f"before call {k}:", flush=True); ctx.aligner_stats()
    t = time.perf_counter(); ctx.score(f); dt = (time.perf_counter() - t) * 1e6
    if k in which:
        print(f"call {k}: {dt:.0f} us, phases {np.round(ctx.debug_profile(), 1)}", flush=True); ctx.aligner_stats()
        nodes = sorted(set(abs(x) for q in p for x in q) - set(abs(x) for q in (seq[k - 1] if k else start) for x in q))
        print("   nodes new to the path set:", [(n, int(g.lens[n]) if hasattr(g, "lens") else None) for n in nodes][:10])
ctx.close()
ctx.close()
