"""GPU parity of the paired-end scorer (through the C ABI) against the oracle."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _setup(G, n, seed, repeats=0, penalty=0.0, err=0.01, min_prob_per_base=-0.7, min_prob_start=-10.0):
    from gaml_amd import api
    import oracle_py as op
    genome = synth.make_genome(G, seed)
    if repeats:
        genome = synth.plant_repeats(genome, repeats, 700, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed))
    pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, err, seed)
    gb, go = g.packed()
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(300.0, 30.0, penalty_constant=penalty, min_prob_per_base=min_prob_per_base,
                                       min_prob_start=min_prob_start), b1, o1, b2, o2)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(300.0, 30.0, penalty_constant=penalty,
                                                             min_prob_per_base=min_prob_per_base,
                                                             min_prob_start=min_prob_start))
    return g, ctx, rs, orc, ors


def _check(ctx, rs, orc, ors, paths, rel_tol=1e-9, prob_rtol=4e-16):
    got, zeros, tl = ctx.calc_prob(paths)
    want, wzeros, wtl = orc.calc_prob(paths, fresh=True)
    assert tl == wtl
    assert zeros.tolist() == wzeros.tolist()
    probs = ctx.read_probs(rs)
    wprobs, wbad = orc.paired_probs(ors)
    # per-read probabilities: same products, same tables; reads with one term are bit-identical,
    # reads with several terms may differ in summation order (<= a few ulp)
    np.testing.assert_allclose(probs, wprobs, rtol=prob_rtol, atol=0)
    assert ctx.bad_bases(rs) == wbad or orc is None
    assert abs(got - want) <= rel_tol * abs(want), (got, want)
    return got, want


def test_true_genome_one_walk():
    g, ctx, rs, orc, ors = _setup(120_000, 6_000, 3)
    _check(ctx, rs, orc, ors, [synth.genome_walk(g)])


def test_singleton_start_state_and_back():
    g, ctx, rs, orc, ors = _setup(120_000, 6_000, 4)
    walk = synth.genome_walk(g)
    singles = [[i] for i in walk if g.node_len(i) > 500]
    _check(ctx, rs, orc, ors, singles)
    _check(ctx, rs, orc, ors, [walk])  # activates further windows: the device order of the pairs settles
    a, _ = _check(ctx, rs, orc, ors, singles)
    b, _ = _check(ctx, rs, orc, ors, [walk])
    c, _ = _check(ctx, rs, orc, ors, singles)
    # pure function of (paths, cache): bit-identical whatever was scored in between, as long as no
    # new window got activated in between (that re-sorts the pairs and with them the summation order)
    assert a == c


def test_gaps_reversed_and_repeats():
    g, ctx, rs, orc, ors = _setup(150_000, 8_000, 5, repeats=4)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    inv = [x ^ 1 for x in reversed(walk[k:2 * k])]
    paths = [walk[:k] + [-137] + walk[k + 2:2 * k], inv, walk[2 * k:], walk[3:9]]  # gap, twin walk, duplicate nodes
    _check(ctx, rs, orc, ors, paths)
    _check(ctx, rs, orc, ors, [[]] + paths[:2])  # an empty path
    _check(ctx, rs, orc, ors, [[-50] + walk[:k]])  # leading gap


def test_coverage_penalty_bad_bases():
    g, ctx, rs, orc, ors = _setup(150_000, 2_500, 6, penalty=0.0001)  # thin coverage -> uncovered stretches
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    for paths in ([walk], [walk[:k], walk[k:]], [walk[:k] + [-300] + walk[k + 4:]]):
        got, zeros, tl = ctx.calc_prob(paths)
        want, wzeros, wtl = orc.calc_prob(paths, fresh=True)
        _, wbad = orc.paired_probs(ors)
        assert ctx.bad_bases(rs) == wbad
        assert wbad > 0
        assert abs(got - want) <= 1e-9 * abs(want)


def test_no_reads_align():
    g, ctx, rs, orc, ors = _setup(60_000, 500, 7)
    # a path of only short nodes: nothing aligns well -> every read floored
    walk = synth.genome_walk(g)
    paths = [[walk[1]]]
    got, zeros, tl = ctx.calc_prob(paths)
    want, wzeros, wtl = orc.calc_prob(paths, fresh=True)
    assert zeros.tolist() == wzeros.tolist()
    assert abs(got - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("copies", [10, 40])
def test_windows_occurring_many_times(copies):
    # the same walk `copies` times over: every window's occurrence list is longer than the eight entries the general route
    # requests together (10: two rounds), and a pair with two records per mate has more candidates than that route stages
    # (40: 80 > kGenCands -- the plain loop takes over; class 0 pairs go through compact_general's loop form, lists > 8)
    g, ctx, rs, orc, ors = _setup(40_000, 3_000, 33, repeats=1)
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    many = [list(walk) for _ in range(copies)] + [walk[:k], walk[k:]]
    tol = dict(prob_rtol=copies * 1e-16)  # (a read's probability is a sum of `copies` + 1 terms here: the order of additions shows)
    a, _ = _check(ctx, rs, orc, ors, many, **tol)
    _check(ctx, rs, orc, ors, [walk])
    b, _ = _check(ctx, rs, orc, ors, many, **tol)
    assert a == b
    res = ctx.calc_prob_batch([many, [walk], many])
    assert res[0][0] == a and res[2][0] == a


@pytest.mark.parametrize("penalty", [0.0, 0.5])
def test_every_window_occurs_several_times(penalty):
    # the whole genome walk three times (twice forward, once as the twin walk) plus stretches of it: every
    # window occurs several times, so every pair of the table classes takes the general pass
    # (GEN instantiation); then the same paths once each, and back -- nothing of the general route must stick
    g, ctx, rs, orc, ors = _setup(90_000, 6_000, 21, repeats=2, penalty=penalty)
    walk = synth.genome_walk(g)
    inv = [x ^ 1 for x in reversed(walk)]  # the twin walk
    k = len(walk) // 3
    many = [walk, list(walk), inv, walk[k:2 * k], walk[:k] + walk[:k]]
    a, _ = _check(ctx, rs, orc, ors, many)
    _check(ctx, rs, orc, ors, [walk])
    b, _ = _check(ctx, rs, orc, ors, many)
    assert a == b
    if penalty == 0.0:  # the batch form sums the partials on the device (finisher kernel): same values
        res = ctx.calc_prob_batch([many, [walk], many])
        assert res[0][0] == a and res[2][0] == a
