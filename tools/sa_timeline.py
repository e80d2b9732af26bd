"""In-kernel timeline (knob 3 = 8) of the scoring kernel for annealing-pattern path sets on tables that were built for
other path sets (delta pairs present): which class of blocks ends last.  python tools/sa_timeline.py"""
import os, sys
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
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
if len(sys.argv) < 2 or sys.argv[1] != "fresh":
    [ctx.score(v) for v in variants]
start, seq = synth.sa_sequence(g, int(os.environ.get("SA_STEPS", "300")))
ctx.calc_prob(start)
flat = [api.FlatPaths(p) for p in seq]
for f in flat: ctx.score(f)
print(ctx.debug_table_stats(rs), ctx.debug_class_counts(rs))
ctx.debug_set_knob(3, 8)
for f in flat[-3:]:
    ctx.score(f)
    t = ctx.debug_timeline(rs).astype(np.int64)
    nb = len(t) // 4
    blocks = t[: nb * 4].reshape(nb, 4, 8)
    ok = blocks[:, 0, 0] > 0
    t0 = blocks[ok][:, :, 0].min()
    ent = (blocks[:, :, 0].min(axis=1) - t0) / 100
    end = (blocks[:, :, 6].max(axis=1) - t0) / 100
    cls = blocks[:, 0, 7]
    print(f"last block reduced at {end[ok].max():.2f} us; blocks {nb}")
    # physical block b runs logical block total-1-b: delta / regs classes first
    for c in (0, 1, 2):
        m = ok & (cls == c)
        if m.any():
            print(f"  class tag {c}: {m.sum()} blocks, enter {np.median(ent[m]):.2f} (max {ent[m].max():.2f}), done median {np.median(end[m]):.2f} p90 {np.percentile(end[m], 90):.2f} max {end[m].max():.2f}, duration median {np.median(end[m] - ent[m]):.2f} max {(end[m] - ent[m]).max():.2f}")
    m2 = ok & (cls == 2)
    if m2.any():
        dur = end - ent
        worst = np.argsort(-(dur * m2))[:6]
        print("  slowest blocks outside the two big classes (logical block = blocks - 1 - dispatch index; classes", ctx.debug_class_counts(rs), "):",
              [(int(nb - 1 - b), round(float(dur[b]), 1)) for b in worst])
