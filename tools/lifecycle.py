import os, sys, resource
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gaml_amd import synth, api
genome = synth.make_genome(300_000, 5)
g = synth.make_graph(genome, synth.cut_lengths(300_000, 5))
pr = synth.make_paired_reads(genome, 40_000, 150, 300.0, 30.0, 0.01, 5)
walk = synth.genome_walk(g)
free0 = None
for it in range(25):
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    ctx.add_paired(api.paired_cfg(300.0, 30.0, penalty_constant=0.0003 if it % 2 else 0.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    v = [ctx.calc_prob([walk])[0], ctx.calc_prob([walk[:9], walk[9:]])[0]]
    ctx.calc_prob_batch([[walk], [walk[:5], walk[5:]]])
    del ctx
    import gc; gc.collect()
    free, total = torch.cuda.mem_get_info()
    if it == 2: free0 = free
    if it % 6 == 0 or it == 24: print(it, v[0], "device free MB", free >> 20, "RSS MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10)
print("device memory drift MB since iteration 2:", (free0 - free) >> 20)
