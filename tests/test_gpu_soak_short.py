"""A short version of tools/soak.py inside the suite: two random graphs, 90 annealing-style edits each with
batch calls interleaved, every value / floored count / bad_bases against the oracle evaluated from scratch."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [311, 312])
def test_random_walk_with_batches_matches_oracle(seed):
    from gaml_amd import api
    import oracle_py as op
    from test_gpu_sa_pattern import _moves
    rng = np.random.default_rng(seed)
    G = int(rng.integers(60_000, 120_000))
    n = int(rng.integers(3000, 12000))
    L = int(rng.choice([75, 100, 150]))
    penalty = 0.0003 if seed % 2 else 0.0
    genome = synth.plant_repeats(synth.make_genome(G, seed), int(rng.integers(1, 4)), int(rng.integers(300, 1000)), seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 4000), short_rng=(20, 340)))
    pr = synth.make_paired_reads(genome, n, L, 260.0, 26.0, 0.01, seed)
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(260.0, 26.0, penalty_constant=penalty), *args)
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    ors = orc.add_paired(*args, 0.01, op.paired_cfg(260.0, 26.0, penalty_constant=penalty))
    walk = synth.genome_walk(g)
    cur = [[x] for x in walk if g.node_len(x) > 500] if seed % 2 else [walk]
    worst = 0.0
    for it in range(90):
        new = _moves(rng, cur, g)
        if it % 6 == 2:
            cands = [_moves(rng, cur, g) for _ in range(2)] + [new]
            for c, gv in zip(cands, ctx.calc_prob_batch(cands)):
                wv = orc.calc_prob(c, fresh=True)
                assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), (seed, it)
                worst = max(worst, abs(gv[0] - wv[0]) / abs(wv[0]))
        else:
            gv = ctx.calc_prob(new)
            wv = orc.calc_prob(new, fresh=True)
            assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), (seed, it)
            assert ctx.bad_bases(rs) == (orc.paired_probs(ors)[1] if penalty > 0 else 0), (seed, it)
            worst = max(worst, abs(gv[0] - wv[0]) / abs(wv[0]))
        if rng.random() < 0.6:
            cur = new
    assert worst <= 1e-9


def test_dirty_slot_whose_other_mate_sits_in_a_repeated_window():
    # tools/soak.py seed 205 up to its iteration 90: a class-0 slot moved to the delta lists (dirty mark in mate
    # 1's table) whose mate-2 record lies in a window that occurs several times. The four-wide compact body once
    # noted such slots for the general pass as well: scored twice, per-read probability overwritten with 0.
    from gaml_amd import api
    import oracle_py as op
    from test_gpu_sa_pattern import _moves
    seed = 205
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
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(mean, mean / 10, penalty_constant=penalty), *args)
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    ors = orc.add_paired(*args, 0.01, op.paired_cfg(mean, mean / 10, penalty_constant=penalty))
    walk = synth.genome_walk(g)
    cur = [[x] for x in walk if g.node_len(x) > 500] if seed % 2 else [walk]
    for it in range(94):
        new = _moves(rng, cur, g)
        if it % 7 == 3:
            cands = [_moves(rng, cur, g) for _ in range(3)] + [new]
            got = ctx.calc_prob_batch(cands)
            if it >= 80:
                for c, gv in zip(cands, got):
                    wv = orc.calc_prob(c, fresh=True)
                    assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), it
        else:
            gv = ctx.calc_prob(new)
            if it >= 80:  # the oracle is the slow side: only around the iteration that failed
                wv = orc.calc_prob(new, fresh=True)
                assert gv[2] == wv[2] and gv[1].tolist() == wv[1].tolist(), it
                assert abs(gv[0] - wv[0]) <= 1e-9 * abs(wv[0])
                np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=1e-15, atol=0)
        if rng.random() < 0.6:
            cur = new
