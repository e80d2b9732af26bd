"""Replay one soak seed and report the first mismatch in detail.  python tools/soak_debug.py SEED [knob=value ...]"""
import os, sys
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaml_amd import api, synth
import oracle_py as op
from test_gpu_sa_pattern import _moves

seed = int(sys.argv[1]); steps = 300
knobs = dict((int(a.split("=")[0]), int(a.split("=")[1])) for a in sys.argv[2:])
rng = np.random.default_rng(seed)
G = int(rng.integers(60_000, 160_000)); n = int(rng.integers(3000, 30000)); L = int(rng.choice([75, 100, 150]))
penalty = float(rng.choice([0.0, 0.0003]))
genome = synth.plant_repeats(synth.make_genome(G, seed), int(rng.integers(1, 5)), int(rng.integers(300, 1200)), seed)
g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 4000), short_rng=(20, 340)))
mean = float(rng.choice([220.0, 300.0, 400.0]))
pr = synth.make_paired_reads(genome, n, L, mean, mean / 10, 0.01, seed)
args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
ctx = api.Context(device=0); ctx.set_graph(*g.packed())
for k, v in knobs.items(): ctx.debug_set_knob(k, v)
rs = ctx.add_paired(api.paired_cfg(mean, mean / 10, penalty_constant=penalty), *args)
orc = op.Oracle(); orc.set_graph(*g.packed())
ors = orc.add_paired(*args, 0.01, op.paired_cfg(mean, mean / 10, penalty_constant=penalty))
walk = synth.genome_walk(g)
cur = [[x] for x in walk if g.node_len(x) > 500] if seed % 2 else [walk]
print(f"seed {seed}: G={G} pairs={n} L={L} penalty={penalty} mean={mean} knobs={knobs}")
def report(it, what, paths, gv, wv):
    print(f"MISMATCH at it {it} ({what}): got {gv[0]!r} zeros {gv[1].tolist()} tl {gv[2]} | want {wv[0]!r} zeros {wv[1].tolist()} tl {wv[2]}")
    g2 = ctx.calc_prob(paths); w2 = orc.calc_prob(paths, fresh=True)
    print(f"  re-evaluated alone: got {g2[0]!r} zeros {g2[1].tolist()} | want {w2[0]!r} zeros {w2[1].tolist()}")
    p, q = ctx.read_probs(rs), orc.paired_probs(ors)[0]
    bad = np.nonzero(np.abs(p - q) > 1e-15 * np.abs(q))[0]
    print(f"  reads with different probability: {len(bad)} of {len(p)}; first {bad[:10].tolist()}; got {p[bad[:5]].tolist()} want {q[bad[:5]].tolist()}")
    print("  classes", ctx.debug_class_counts(rs), "tables", ctx.debug_table_stats(rs))
    sys.exit(1)
for it in range(steps):
    new = _moves(rng, cur, g)
    if it % 7 == 3:
        cands = [_moves(rng, cur, g) for _ in range(3)] + [new]
        got = ctx.calc_prob_batch(cands)
        for ci, (c, gv) in enumerate(zip(cands, got)):
            wv = orc.calc_prob(c, fresh=True)
            if not (gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist()): report(it, f"batch candidate {ci}", c, gv, wv)
    else:
        gv = ctx.calc_prob(new); wv = orc.calc_prob(new, fresh=True)
        if not (gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist()): report(it, "single", new, gv, wv)
    if rng.random() < 0.6: cur = new
print("no mismatch")
