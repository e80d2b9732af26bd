"""One context, tables taken over from the worker 40 evaluations after its start (the take-over test's first context)."""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
G, n, seed = 200_000, 36_000, 23
genome = synth.plant_repeats(synth.make_genome(G, seed), 3, 800, seed)
g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
c = api.Context(device=0)
c.set_graph(*g.packed())
c.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
c.debug_set_knob(14, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
start, seq = synth.sa_sequence(g, 500, seed=9, threshold=400)
for k, ps in enumerate([start] + seq):
    v = c.calc_prob(ps)
    if k % 20 == 0: print(k, v[0], c.debug_table_stats(0), flush=True)
c.close()
