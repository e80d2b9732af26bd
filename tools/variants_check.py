import sys, os
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
for rep in range(3):
    print(rep, [round(ctx.calc_prob(v)[0], 9) for v in variants], flush=True)
import random
order = list(range(8)); random.Random(1).shuffle(order)
print("shuffled", order, [round(ctx.calc_prob(variants[i])[0], 9) for i in order])
