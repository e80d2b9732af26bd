import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, '/root/repo')
import numpy as np
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, 300)
flat = [api.FlatPaths(p) for p in seq]
def run(tag, prime):
    ctx = api.Context(device=0); ctx.set_graph(*g.packed()); rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    if prime:
        vs = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
        [ctx.score(v) for v in vs]
        if prime > 1: ctx.compact_tables(); ctx.score(vs[0])
    ctx.calc_prob(start)
    for f in flat: ctx.score(f)
    prof = []
    for f in flat[-150:]: ctx.score(f); prof.append(ctx.debug_profile())
    ctx.set_event_timing(1); ctx.kernel_stats(reset=True)
    for f in flat[-150:]: ctx.score(f)
    ks = ctx.kernel_stats(reset=True); ctx.set_event_timing(0)
    print(tag, "score kernel avg us", ks["device_us"] / max(1, ks["launches"]), end=" ")
    print("wait median", np.median(np.array(prof)[:, 7]), "classes", ctx.debug_class_counts(rs), ctx.debug_table_stats(rs), flush=True)
    ctx.close()
run("fresh", 0); run("primed with genome walks", 1); run("primed + compact", 2)
