import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS[sys.argv[1]]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
ts = []
for i in range(260):
    if i == 28: ctx.set_event_timing(True)
    t = time.perf_counter(); ctx.eval_begin(variants[i % 8]); p = ctx.eval_finish(); ts.append((time.perf_counter() - t) * 1e6)
ts = np.array(ts)
print("median us", np.median(ts[30:]), "slow steps (>500us):", [(i, round(x)) for i, x in enumerate(ts) if x > 500])
print(ctx.debug_table_stats(rs))
