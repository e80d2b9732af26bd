"""Edge cases of the paired path against the oracle: ragged read lengths, non-ACGT bases, empty
inputs, very many paths (path ids beyond the compact 15-bit field), long single-node windows,
unusual scoring parameters."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _pack_ragged(reads):
    offs = np.zeros(len(reads) + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.ascontiguousarray(np.concatenate(reads) if reads else np.zeros(0, np.uint8)), offs


def _both(gb, go, r1, r2, cfg_kw, mean, sd):
    from gaml_amd import api
    import oracle_py as op
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(mean, sd, **cfg_kw), *r1, *r2)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(*r1, *r2, cfg_kw.get("mismatch_prob", 0.01), op.paired_cfg(mean, sd, **{k: v for k, v in cfg_kw.items() if k != "mismatch_prob"}))
    return ctx, rs, orc, ors


def _agree(ctx, rs, orc, ors, paths, tol=1e-9):
    got, zeros, tl = ctx.calc_prob(paths)
    want, wz, wtl = orc.calc_prob(paths, fresh=True)
    assert tl == wtl and zeros.tolist() == wz.tolist()
    wprobs, wbad = orc.paired_probs(ors)
    np.testing.assert_allclose(ctx.read_probs(rs), wprobs, rtol=4e-16, atol=0)
    assert abs(got - want) <= tol * abs(want)


def test_ragged_read_lengths_and_non_acgt_bases():
    G, n, seed = 70_000, 3000, 101
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(G, seed)
    genome[30_000:30_040] = ord("N")  # a run of N inside a node
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 4000)))
    pr = synth.make_paired_reads(genome, n, 120, 300.0, 30.0, 0.01, seed)
    # trim reads to ragged lengths (the index assumes the LAST read's length, graph.cc:1286), sprinkle N
    m1 = [pr.mate1[i, : int(rng.integers(90, 121))].copy() for i in range(n)]
    m2 = [pr.mate2[i, : int(rng.integers(90, 121))].copy() for i in range(n)]
    for lst in (m1, m2):
        for i in rng.integers(0, n, 60):
            lst[i][int(rng.integers(0, len(lst[i])))] = ord("N")
    ctx, rs, orc, ors = _both(*g.packed(), _pack_ragged(m1), _pack_ragged(m2), {}, 300.0, 30.0)
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    for paths in ([walk], [walk[:k], walk[k:]], [[x] for x in walk]):
        _agree(ctx, rs, orc, ors, paths)
    assert ctx.aligner_stats()["windows"] > 0  # the GPU aligner handled the ragged set


def test_more_than_300_distinct_length_combinations():
    """Beyond 256 (L1, L2) combinations pairs leave the compact class; results must not change."""
    G, n, seed = 40_000, 2500, 103
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    pr = synth.make_paired_reads(genome, n, 150, 320.0, 30.0, 0.01, seed)
    m1 = [pr.mate1[i, : 100 + (i % 37)].copy() for i in range(n)]
    m2 = [pr.mate2[i, : 100 + (i % 41)].copy() for i in range(n)]
    m1[-1] = pr.mate1[-1, :136].copy(); m2[-1] = pr.mate2[-1, :140].copy()
    ctx, rs, orc, ors = _both(*g.packed(), _pack_ragged(m1), _pack_ragged(m2), {}, 320.0, 30.0)
    _agree(ctx, rs, orc, ors, [synth.genome_walk(g)])
    cls = ctx.debug_class_counts(rs)
    assert cls[0] > 0 and cls[1] > 0


def test_many_paths_beyond_compact_path_id():
    """> 32767 paths: occurrence entries whose path id does not fit 15 bits take the general route."""
    G, n, seed = 60_000, 1500, 105
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, 0.01, seed)
    ctx, rs, orc, ors = _both(*g.packed(), synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2), {}, 250.0, 25.0)
    walk = synth.genome_walk(g)
    paths = [[]] * 33_000 + [walk]  # the interesting path has index 33000
    _agree(ctx, rs, orc, ors, paths)
    # and a window that occurs in 300 paths at once (lists longer than the 128-candidate LDS staging)
    short = [x for x in walk if g.node_len(x) < 200][0]
    _agree(ctx, rs, orc, ors, [[short]] * 300 + [walk])


def test_long_single_node_windows_and_one_node_graph():
    G, n, seed = 150_000, 4000, 107
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, [100_000, 50_000])  # two very long nodes, no short ones
    pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, 0.01, seed)
    ctx, rs, orc, ors = _both(*g.packed(), synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2), {}, 300.0, 30.0)
    for paths in ([[0, 2]], [[0], [2]], [[3, 1]], [[2, 0]]):
        _agree(ctx, rs, orc, ors, paths)


def test_empty_inputs():
    from gaml_amd import api
    G, seed = 20_000, 109
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    pr = synth.make_paired_reads(genome, 400, 100, 250.0, 25.0, 0.01, seed)
    ctx, rs, orc, ors = _both(*g.packed(), synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2), {}, 250.0, 25.0)
    for paths in ([], [[]], [[], []], [[-100]], [[-5], []]):
        _agree(ctx, rs, orc, ors, paths, tol=1e-12)
    # a context without read sets scores 0
    c2 = api.Context(device=0)
    c2.set_graph(*g.packed())
    v, z, tl = c2.calc_prob([synth.genome_walk(g)])
    assert v == 0.0 and len(z) == 0 and tl == G
    with pytest.raises(api.GamlHipError):
        c2.calc_prob([[10_000_000]])  # node outside the graph
    # a paired set with zero reads
    c3 = api.Context(device=0)
    c3.set_graph(*g.packed())
    e = np.zeros(0, np.uint8), np.zeros(1, np.int64)
    c3.add_paired(api.paired_cfg(250.0, 25.0), *e, *e)
    part, _ = c3.calc_partials([synth.genome_walk(g)])
    assert part[0].tolist() == [0.0, 0.0, 0.0, 0.0]


@pytest.mark.parametrize("kw,mean,sd", [
    (dict(min_prob_per_base=0.0, min_prob_start=-80.0, penalty_constant=0.00013, penalty_step=3000.0), 3700.0, 350.0),  # example.cfg rs2
    (dict(mismatch_prob=0.05, min_prob_per_base=-1.0, weight=0.3), 220.0, 10.0),
    (dict(min_prob_start=80.0, min_prob_per_base=0.0), 180.0, 20.0),  # example.cfg rs1: floor above every probability
])
def test_unusual_scoring_parameters(kw, mean, sd):
    G, n, seed = 80_000, 3000, 111
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 5000)))
    pr = synth.make_paired_reads(genome, n, 100, mean, sd, kw.get("mismatch_prob", 0.01), seed)
    ctx, rs, orc, ors = _both(*g.packed(), synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2), kw, mean, sd)
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    for paths in ([walk], [walk[:k] + [-500] + walk[k + 3:]], [walk[:k], walk[k:]]):
        got, zeros, tl = ctx.calc_prob(paths)
        want, wz, wtl = orc.calc_prob(paths, fresh=True)
        _, wbad = orc.paired_probs(ors)
        assert zeros.tolist() == wz.tolist() and tl == wtl
        if kw.get("penalty_constant", 0) > 0:
            assert ctx.bad_bases(rs) == wbad
        assert abs(got - want) <= 1e-9 * abs(want)


def test_compact_tables_and_sampled_event_timing():
    """Maintenance and measurement entry points: gaml_hip_compact_tables folds the delta store at the next
    evaluation without changing values beyond the order of the final sum; gaml_hip_set_event_timing(k) times
    every k-th scoring launch and gaml_hip_kernel_stats then describes exactly the timed launches."""
    from gaml_amd import api
    genome = synth.make_genome(150_000, 97)
    g = synth.make_graph(genome, synth.cut_lengths(150_000, 97, long_rng=(600, 4000)))
    pr = synth.make_paired_reads(genome, 40_000, 100, 240.0, 24.0, 0.01, 97)
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    sets = [[walk], [walk[:9], walk[9:]], [walk[:20], walk[20:]], [walk[:5], walk[5:30], walk[30:]]]
    first = [ctx.calc_prob(s)[0] for s in sets]          # later sets activate junction windows: delta pairs
    st = ctx.debug_table_stats(rs)
    assert st["dirty_pairs"] > 0 and st["delta_updates"] > 0
    ctx.compact_tables()
    again = [ctx.calc_prob(s)[0] for s in sets]
    st2 = ctx.debug_table_stats(rs)
    assert st2["dirty_pairs"] == 0 and st2["full_rebuilds"] == st["full_rebuilds"] + 1
    for a, b in zip(first, again):
        assert abs(a - b) <= 1e-13 * abs(a)
    # sampled timing
    ctx.set_event_timing(4)
    ctx.kernel_stats(reset=True)
    for i in range(20):
        ctx.calc_prob(sets[i % 4])
    ks = ctx.kernel_stats(reset=True)
    assert ks["launches"] == 5 and ks["device_us"] > 0 and ks["algo_bytes"] > 0
    ctx.set_event_timing(1)
    for i in range(6):
        ctx.calc_prob(sets[i % 4])
    ks = ctx.kernel_stats(reset=True)
    assert ks["launches"] == 6
    per_launch_us = ks["device_us"] / ks["launches"]
    assert 1.0 < per_launch_us < 1000.0
    ctx.set_event_timing(0)
    for i in range(3):
        ctx.calc_prob(sets[i % 4])
    assert ctx.kernel_stats(reset=True)["device_us"] == 0.0


def test_floor_that_underflows_to_zero():
    """min_prob_per_base so low that exp(c + k * (L1 + L2)) is 0.0 in f64: a pair without a scoring alignment has
    probability 0, which is NOT below a floor of 0, so the reference adds log(0) = -inf and counts no floored read
    (graph.cc:1504-1513). The memo and the "probability 0 is floored" shortcuts must stand down."""
    G, n, seed = 60_000, 2500, 131
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 5000)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, 0.01, seed)
    kw = dict(min_prob_per_base=-5.0, min_prob_start=-10.0)  # exp(-10 - 5 * 200) = 0
    ctx, rs, orc, ors = _both(*g.packed(), synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2), kw, 250.0, 25.0)
    walk = synth.genome_walk(g)
    for paths in ([walk], [walk[: len(walk) // 2]]):  # the half walk leaves pairs without any alignment
        got, zeros, tl = ctx.calc_prob(paths)
        want, wz, wtl = orc.calc_prob(paths, fresh=True)
        assert tl == wtl and zeros.tolist() == wz.tolist()
        np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=4e-16, atol=0)
        assert got == want or abs(got - want) <= 1e-9 * abs(want)  # -inf == -inf on the half walk
    assert want == -np.inf


def test_upload_paths_agree():
    """The per-call tables are written by the host straight into device memory (large BAR: fine-grained allocation behind
    the PCIe BAR); knob 8 = 1 selects the pinned staging slot + hipMemcpyAsync route, knob 8 = 2 staging + copy kernel
    (what a device without a large BAR gets). Same values bit for bit -- also for a batch."""
    from gaml_amd import api
    genome = synth.make_genome(90_000, 141)
    g = synth.make_graph(genome, synth.cut_lengths(90_000, 141, long_rng=(600, 4000)))
    pr = synth.make_paired_reads(genome, 9_000, 100, 240.0, 24.0, 0.01, 141)
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    ctx.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    sets = [[walk], [walk[:9], walk[9:]], [walk[:20] + [-40] + walk[22:]]]
    [ctx.calc_prob(s) for s in sets]
    a = [ctx.calc_prob(s)[0] for s in sets]
    ab = [b[0] for b in ctx.calc_prob_batch(sets)]
    for knob in (1, 2, 0):
        ctx.debug_set_knob(8, knob)
        assert [ctx.calc_prob(s)[0] for s in sets] == a, knob
        assert [b[0] for b in ctx.calc_prob_batch(sets)] == ab, knob
    assert ab == a  # one pass over the records for all sets: the same lanes, the same sums


def test_records_that_are_always_overwritten_stay_out_of_the_tables():
    """A read in the last 300 bases of a long node is aligned through the node's window and through the junction window
    that follows it; wherever both occur the node's record overwrites the junction's (graph.cc:583-592, 563-566). The
    table build leaves such junction records out (host_model.cc dominated_records): every per-pair value must be the
    one computed with them (knob 16 = 1) and the oracle's, for path sets that use the junctions, cut them and drop them."""
    G, n, seed = 120_000, 30_000, 4242
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 3000)))
    pr = synth.make_paired_reads(genome, n, 100, 260.0, 26.0, 0.01, seed)
    r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    sets = [[walk], [walk[:k], walk[k:]], [walk[:k] + walk[k + 2:]], [walk[k:2 * k], walk[:k]], [walk]]
    ctx, rs, orc, ors = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    keep, rs_k, _, _ = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    keep.debug_set_knob(16, 1)
    for rnd in range(2):  # second round: tables rebuilt with every window of the sets on the device
        for paths in sets:
            got, kept = ctx.calc_prob(paths), keep.calc_prob(paths)
            assert np.array_equal(ctx.read_probs(rs), keep.read_probs(rs_k))
            assert got[1].tolist() == kept[1].tolist() and abs(got[0] - kept[0]) <= 1e-13 * abs(kept[0])
            _agree(ctx, rs, orc, ors, paths)
        ctx.compact_tables()
        keep.compact_tables()
    ctx.calc_prob(sets[0])
    keep.calc_prob(sets[0])
    st, st_k = ctx.debug_table_stats(rs), keep.debug_table_stats(rs_k)
    assert min(st["records_left_out"]) > 0 and st_k["records_left_out"] == [0, 0]
    c0, c1 = ctx.debug_class_counts(rs), keep.debug_class_counts(rs_k)
    assert c0[0] > c1[0] and sum(c0) == sum(c1) == n
    # the same rule for junction windows that join the device tables LATER (delta lists): tables built for one path per
    # node, then the nodes joined
    ctx2, rs2, orc2, ors2 = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    keep2, rs_k2, _, _ = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    keep2.debug_set_knob(16, 1)
    for paths in ([[x] for x in walk], [walk[:k]] + [[x] for x in walk[k:]], [walk], [walk[:k], walk[k:]], [[x] for x in walk]):
        got, kept = ctx2.calc_prob(paths), keep2.calc_prob(paths)
        assert np.array_equal(ctx2.read_probs(rs2), keep2.read_probs(rs_k2))
        assert got[1].tolist() == kept[1].tolist() and abs(got[0] - kept[0]) <= 1e-13 * abs(kept[0])
        _agree(ctx2, rs2, orc2, ors2, paths)
    st2, st_k2 = ctx2.debug_table_stats(rs2), keep2.debug_table_stats(rs_k2)
    assert st2["delta_records_left_out"] > 0 and st_k2["delta_records_left_out"] == 0


@pytest.mark.parametrize("ragged", [False, True])
def test_static_memo_indices_change_no_value(ragged):
    """A compact-class pair whose two records sit in the same window gets its memo index when the tables are built
    (host_model.cc build_pair_tables: orientation rule and insert distance, graph.cc:1864-1882, do not depend on where the
    window sits); the scoring launch then only asks the occurrence tables WHETHER the pair scores. Knob 19 = 1 resolves
    every pair per call instead. Same per-read probabilities bit for bit -- over path sets that use, cut, drop, invert and
    repeat the windows (general path), for one and for several read-length combinations, single calls and batches -- and
    both agree with the oracle."""
    G, n, seed = 150_000, 36_000, 777
    rng = np.random.default_rng(seed)
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 3500)))
    pr = synth.make_paired_reads(genome, n, 100, 260.0, 26.0, 0.01, seed)
    if ragged:  # several (L1, L2) combinations: the length-code tables instead of the one-combination fast path
        m1 = [pr.mate1[i, : (100 if i % 3 else 96)].copy() for i in range(n)]
        m2 = [pr.mate2[i, : (100 if i % 5 else 92)].copy() for i in range(n)]
        m1[-1] = pr.mate1[-1].copy(); m2[-1] = pr.mate2[-1].copy()  # the index assumes the LAST read's length (graph.cc:1286)
        r1, r2 = _pack_ragged(m1), _pack_ragged(m2)
    else:
        r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    inv = [x ^ 1 for x in reversed(walk[:k])]
    sets = [[walk], [walk[:k], walk[k:]], [walk[:k] + walk[k + 2:]], [walk[k:2 * k], walk[:k]], [inv, walk[k:]],
            [walk[:k] + [-35] + walk[k:]], [walk, walk[k:2 * k]], [walk[:k] + walk[:k]], [], [walk]]
    ctx, rs, orc, ors = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    dyn, rs_d, _, _ = _both(*g.packed(), r1, r2, {}, 260.0, 26.0)
    dyn.debug_set_knob(19, 1)
    for rnd in range(2):  # second round: tables rebuilt with every window of the sets on the device
        for paths in sets:
            got, want = ctx.calc_prob(paths), dyn.calc_prob(paths)
            assert np.array_equal(ctx.read_probs(rs), dyn.read_probs(rs_d)), (rnd, len(paths))
            assert got[1].tolist() == want[1].tolist() and got[2] == want[2]
            assert got[0] == want[0] or abs(got[0] - want[0]) <= 1e-13 * abs(want[0])
            _agree(ctx, rs, orc, ors, paths)
        b1, b2 = ctx.calc_prob_batch(sets[:8]), dyn.calc_prob_batch(sets[:8])
        for s_, x, y in zip(sets[:8], b1, b2):
            assert x[1].tolist() == y[1].tolist() and (x[0] == y[0] or abs(x[0] - y[0]) <= 1e-13 * abs(y[0]))
            one = ctx.calc_prob(s_)
            assert one[0] == x[0], "a batch gives what the calls give one by one, bit for bit"
        ctx.compact_tables()
        dyn.compact_tables()
    ctx.calc_prob(sets[0])
    dyn.calc_prob(sets[0])
    st, st_d = ctx.debug_table_stats(rs), dyn.debug_table_stats(rs_d)
    c0 = ctx.debug_class_counts(rs)
    assert st_d["static_index_pairs"] == 0 and 0.5 * c0[0] < st["static_index_pairs"] <= c0[0]
    assert list(c0) == list(dyn.debug_class_counts(rs_d))
