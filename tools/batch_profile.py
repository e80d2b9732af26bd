"""Candidate batches (8 single-edit candidates of one ~900-path assembly) at cfg3: one-pass kernel vs one launch per set,
and the same candidates one blocking call at a time.  python tools/batch_profile.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, 200, seed=11)
base = seq[-1]
rng = np.random.default_rng(23)
batches, flat_sets = [], []
for _ in range(40):
    cands = [synth.sa_move(rng, base, g) for _ in range(8)]
    batches.append(api.BatchPaths(cands)); flat_sets.append([api.FlatPaths(c) for c in cands])
    if rng.random() < 0.6: base = cands[int(rng.integers(0, 8))]
ctx.calc_prob(base)
for b in batches: ctx.calc_prob_batch(b)
for knob11 in (0, 1, 0):
    ctx.debug_set_knob(11, knob11)
    for b in batches[:5]: ctx.calc_prob_batch(b)
    t = time.perf_counter()
    for b in batches: ctx.calc_prob_batch(b)
    dt = time.perf_counter() - t
    print(f"candidate batches, knob11={knob11}: {dt / 320 * 1e6:.1f} us per set")
t = time.perf_counter()
for fs in flat_sets:
    for f in fs: ctx.score(f)
dt = time.perf_counter() - t
print(f"the same candidates, one blocking call each: {dt / 320 * 1e6:.1f} us per set")
prof = []
for fs in flat_sets:
    for f in fs: ctx.score(f); prof.append(ctx.debug_profile())
print("median phases [pass1, tables_host, -, write, sync, launch, bytes, wait]:", np.round(np.median(np.array(prof), axis=0), 1))
print(ctx.debug_table_occurrences(rs, 0)[1], ctx.debug_table_stats(rs))
ctx.close()
