"""Host cost of the event timing modes on the blocking step (same process, interleaved rounds)."""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
[ctx.score(v) for v in variants]; ctx.compact_tables(); [ctx.score(v) for v in variants]
import gc; gc.disable()
res = {0: [], 8: [], 1: []}
for rnd in range(5):
    for mode in (0, 8, 1):
        ctx.set_event_timing(mode); ctx.kernel_stats(reset=True)
        for i in range(50): ctx.score(variants[i % 8])
        t = time.perf_counter()
        for i in range(400): ctx.score(variants[i % 8])
        res[mode].append((time.perf_counter() - t) / 400 * 1e6)
        ctx.kernel_stats(reset=True)
for m in res: print("event timing mode", m, "step us median %.1f min %.1f" % (np.median(res[m]), min(res[m])))
