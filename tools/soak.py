"""Randomised soak of the scoring path against the oracle (not part of the test suite: minutes of oracle
time). Several seeds x long annealing-style walks over path sets with planted repeats, gaps, duplicated
nodes and reversed paths; single calls, batch calls and re-evaluations interleaved; every value, floored
count, bad_bases and (periodically) every per-read probability compared with the oracle evaluated from
scratch.   python tools/soak.py [seeds] [steps]"""
import os
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaml_amd import api, synth  # noqa: E402
import oracle_py as op  # noqa: E402
from test_gpu_sa_pattern import _moves  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
t_all = time.time()
worst = 0.0
for seed in range(200, 200 + n_seeds):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(60_000, 160_000))
    n = int(rng.integers(3000, 30000))
    L = int(rng.choice([75, 100, 150]))
    penalty = float(rng.choice([0.0, 0.0003]))
    genome = synth.plant_repeats(synth.make_genome(G, seed), int(rng.integers(1, 5)), int(rng.integers(300, 1200)), seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 4000), short_rng=(20, 340)))
    mean = float(rng.choice([220.0, 300.0, 400.0]))
    pr = synth.make_paired_reads(genome, n, L, mean, mean / 10, 0.01, seed)
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    ctx = api.Context(device=0)
    # the table machinery at sizes where it would otherwise never run: rebuilds when the delta lists pass pairs / d, taking over
    # k evaluations after their start (1: on the calling stream), delta maintenance by one-block / multi-block launches
    knobs = {18: int(rng.choice([0, 16, 64])), 14: int(rng.choice([0, 1, 8, 24])), 22: int(rng.choice([0, 1]))}
    for k, v in knobs.items():
        ctx.debug_set_knob(k, v)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(mean, mean / 10, penalty_constant=penalty), *args)
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    ors = orc.add_paired(*args, 0.01, op.paired_cfg(mean, mean / 10, penalty_constant=penalty))
    walk = synth.genome_walk(g)
    cur = [[x] for x in walk if g.node_len(x) > 500] if seed % 2 else [walk]
    pend = []
    for it in range(steps):
        new = _moves(rng, cur, g)
        if it % 7 == 3:  # a batch of speculative candidates
            cands = [_moves(rng, cur, g) for _ in range(3)] + [new]
            got = ctx.calc_prob_batch(cands)
            for c, gv in zip(cands, got):
                wv = orc.calc_prob(c, fresh=True)
                assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), (seed, it)
                worst = max(worst, abs(gv[0] - wv[0]) / abs(wv[0]))
        else:
            gv = ctx.calc_prob(new)
            wv = orc.calc_prob(new, fresh=True)
            assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), (seed, it)
            _, wbad = orc.paired_probs(ors)
            assert ctx.bad_bases(rs) == (wbad if penalty > 0 else 0), (seed, it)
            worst = max(worst, abs(gv[0] - wv[0]) / abs(wv[0]))
            if it % 25 == 0:
                np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=1e-15, atol=0)  # a read with several terms: a few ulp (order of its sum)
        if rng.random() < 0.6:
            cur = new
    st = ctx.debug_table_stats(rs)
    print(f"seed {seed}: G={G} pairs={n} L={L} penalty={penalty} knobs={knobs} ok; worst rel delta so far {worst:.2e}; tables {st}", flush=True)
assert worst <= 1e-9, worst
print(f"soak passed: {n_seeds} seeds x {steps} steps in {time.time() - t_all:.0f} s, worst relative difference {worst:.2e}")
