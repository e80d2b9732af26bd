"""The call pattern of the simulated-annealing loop (reference gaml.cc:148-339 and the CalcProb calls
inside moves.cc): a long sequence of slightly different path sets, new junction windows appearing
all the time, speculative evaluations interleaved with re-evaluations of earlier sets. Every value
is checked against the oracle evaluated with a fresh ScoringState on the same cache history."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


_moves = synth.sa_move  # one random edit of the kind GAML's move generators make (shared with bench.py / tools)


@pytest.mark.parametrize("penalty", [0.0, 0.0002])
def test_long_random_walk_of_path_sets_matches_oracle(penalty):
    from gaml_amd import api
    import oracle_py as op
    G, n, seed = 120_000, 6000, 91
    genome = synth.plant_repeats(synth.make_genome(G, seed), 3, 800, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    gb, go = g.packed()
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(240.0, 24.0, penalty_constant=penalty), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(240.0, 24.0, penalty_constant=penalty))
    rng = np.random.default_rng(5)
    walk = synth.genome_walk(g)
    cur = [[x] for x in walk if g.node_len(x) > 500]  # the reference's start state (gaml.cc:1002-1005)
    history = []
    worst = 0.0
    for it in range(120):
        new = _moves(rng, cur, g)
        got, zeros, tl = ctx.calc_prob(new)
        want, wz, wtl = orc.calc_prob(new, fresh=True)
        assert tl == wtl and zeros.tolist() == wz.tolist(), it
        _, wbad = orc.paired_probs(ors)
        assert ctx.bad_bases(rs) == (wbad if penalty > 0 else 0), it
        worst = max(worst, abs(got - want) / abs(want))
        history.append((new, got))
        if rng.random() < 0.6:  # accept
            cur = new
        if it % 10 == 9:  # a move generator re-evaluating an earlier candidate (moves.cc:705-793)
            old_paths, old_val = history[int(rng.integers(0, len(history)))]
            again, _, _ = ctx.calc_prob(old_paths)
            want_again, _, _ = orc.calc_prob(old_paths, fresh=True)
            assert abs(again - want_again) <= 1e-9 * abs(want_again)
    assert worst <= 1e-9
    # per-read state of the last evaluation too
    np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=4e-16, atol=0)
    assert ctx.window_count(rs, 0) == orc.L.orc_window_count(orc.h, ors, 0)


def test_delta_list_equals_full_rebuild():
    """Newly activated windows go to a delta list instead of rebuilding the device tables; both
    routes must give the same per-read probabilities and likelihoods."""
    from gaml_amd import api
    G, n, seed = 150_000, 40_000, 93
    genome = synth.plant_repeats(synth.make_genome(G, seed), 2, 700, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    ctxs = []
    for no_delta in (0, 1):
        c = api.Context(device=0)
        c.debug_set_knob(6, no_delta)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        ctxs.append(c)
    rng = np.random.default_rng(11)
    cur = [synth.genome_walk(g)]
    for it in range(60):
        cur = _moves(rng, cur, g)
        a = ctxs[0].calc_prob(cur)
        b = ctxs[1].calc_prob(cur)
        assert a[1].tolist() == b[1].tolist() and a[2] == b[2]
        assert abs(a[0] - b[0]) <= 1e-13 * abs(b[0]), it
        np.testing.assert_allclose(ctxs[0].read_probs(0), ctxs[1].read_probs(0), rtol=4e-16, atol=0)
    st0, st1 = ctxs[0].debug_table_stats(0), ctxs[1].debug_table_stats(0)
    assert st0["delta_updates"] > 5 and st0["full_rebuilds"] < st1["full_rebuilds"]
    assert st1["delta_updates"] == 0 and st1["dirty_pairs"] == 0


def test_incremental_planning_and_resident_tables_change_no_bit():
    """Three contexts walk the same annealing-style sequence: (a) the default -- paths diffed against the previous call,
    the resident device copy of the occurrence tables patched in place; (b) every set planned from scratch (knob 12);
    (c) whole tables through the ring for every call (knob 13). Values, floored counts and per-read probabilities are
    equal bit for bit at every step (the tables describe the same occurrences; slots and ranks only ever enter through
    equality / order within one path)."""
    from gaml_amd import api
    G, n, seed = 150_000, 8000, 17
    genome = synth.plant_repeats(synth.make_genome(G, seed), 3, 800, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    ctxs = []
    for knob in (None, 12, 13):
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        if knob:
            c.debug_set_knob(knob, 1)
        ctxs.append(c)
    start, seq = synth.sa_sequence(g, 300, seed=5, threshold=400)
    rng = np.random.default_rng(1)
    sets = [start]
    for s in seq:
        sets.append(s)
        if rng.random() < 0.1:
            sets.append(sets[int(rng.integers(0, len(sets)))])
    n_inc = 0
    for k, ps in enumerate(sets):
        vals = [c.calc_prob(ps) for c in ctxs]
        for v in vals[1:]:
            assert v[0] == vals[0][0] and v[1].tolist() == vals[0][1].tolist() and v[2] == vals[0][2], (k, vals)
        if k % 25 == 0:
            p0 = ctxs[0].read_probs(0)
            assert np.array_equal(p0, ctxs[1].read_probs(0)) and np.array_equal(p0, ctxs[2].read_probs(0))
        n_inc += ctxs[0].debug_table_occurrences(0, 0)[1]["incremental"]
    assert n_inc > len(sets) // 2


def test_rebuild_on_the_worker_thread_takes_over_mid_walk():
    """When the delta lists pass pairs / 8 the record tables are rebuilt by a worker thread from a private copy while
    evaluations go on; a later call swaps the new tables in and re-bases the pairs touched since the snapshot. Against
    a context that rebuilds on the calling thread (knob 14) and one that never uses delta lists (knob 6 = 1): same
    floored counts, values and per-read probabilities within re-association noise."""
    from gaml_amd import api
    G, n, seed = 200_000, 36_000, 23
    genome = synth.plant_repeats(synth.make_genome(G, seed), 3, 800, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    ctxs = []
    # the new tables take over 40 evaluations after the worker started (default 1152). The last context is the first one
    # again, but slowed down: its worker is done long before the take-over, so the lists that go with the new tables are
    # prepared in other slices than the first context's -- and every value must still be equal bit for bit (the point
    # of a take-over at a fixed evaluation count)
    for knobs in ({14: 40}, {14: 1}, {6: 1}, {14: 40}):
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        for k, v in knobs.items():
            c.debug_set_knob(k, v)
        ctxs.append(c)
    start, seq = synth.sa_sequence(g, 500, seed=9, threshold=400)
    import time
    slow_ctx = ctxs.pop()
    first = []
    for k, ps in enumerate([start] + seq):
        vals = [c.calc_prob(ps) for c in ctxs]
        first.append(vals[0][0])
        for v in vals[1:]:
            assert v[1].tolist() == vals[0][1].tolist() and v[2] == vals[0][2]
            assert v[0] == vals[0][0] or abs(v[0] - vals[0][0]) <= 1e-12 * abs(vals[0][0]), (k, v[0], vals[0][0])
        if k % 50 == 0:
            p0 = ctxs[0].read_probs(0)
            # (a pair with several terms adds them in table order: list vs rebuilt table may differ in the last bit)
            np.testing.assert_allclose(p0, ctxs[1].read_probs(0), rtol=4e-16, atol=0)
            np.testing.assert_allclose(p0, ctxs[2].read_probs(0), rtol=4e-16, atol=0)
    for k, ps in enumerate([start] + seq):
        assert slow_ctx.calc_prob(ps)[0] == first[k], k
        time.sleep(0.001)
    assert slow_ctx.debug_table_stats(0)["worker_rebuilds"] == ctxs[0].debug_table_stats(0)["worker_rebuilds"]
    st = ctxs[0].debug_table_stats(0)
    assert st["worker_rebuilds"] >= 1, st
    assert ctxs[1].debug_table_stats(0)["worker_rebuilds"] == 0 and ctxs[1].debug_table_stats(0)["full_rebuilds"] >= 2


def test_worker_take_overs_at_the_large_table_scale_against_the_oracle():
    """The rebuild machinery at the scale where build_pair_tables takes its multi-threaded branch (>= 2^16 pairs): 90,000
    pairs, tables rebuilt when the delta lists pass pairs / 64 (knob 18) and taken over 24 evaluations after the worker
    started (knob 14), so that several take-overs -- snapshot in slices, second delta store, re-basing of what was
    activated meanwhile, windows retired while a snapshot is copied -- happen within a few hundred annealing steps.
    Value, floored count and per-read probabilities against the ORACLE (fresh scoring state) at every step."""
    import oracle_py as op
    from gaml_amd import api
    G, n, seed = 600_000, 90_000, 31
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(800, 5000)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(240.0, 24.0), *r1, *r2)
    ctx.debug_set_knob(14, 24)
    ctx.debug_set_knob(18, 64)
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    ors = orc.add_paired(*r1, *r2, 0.01, op.paired_cfg(240.0, 24.0))
    start, seq = synth.sa_sequence(g, 260, seed=13, threshold=400)
    seen = 0
    for k, ps in enumerate([start] + seq):
        got = ctx.calc_prob(ps)
        want, wz, wtl = orc.calc_prob(ps, fresh=True)
        assert got[2] == wtl and got[1].tolist() == wz.tolist(), k
        assert abs(got[0] - want) <= 1e-9 * abs(want), (k, got[0], want)
        w = ctx.debug_table_stats(rs)["worker_rebuilds"]
        if w != seen or k % 40 == 0:  # right after a take-over, and now and then
            np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=4e-16, atol=0)
            seen = w
    st = ctx.debug_table_stats(rs)
    assert st["worker_rebuilds"] >= 3 and st["delta_updates"] > 20, st


@pytest.mark.parametrize("phase", ["snapshot", "worker"])
def test_destroy_and_compact_while_a_rebuild_is_in_flight(phase):
    """Lifecycle around the rebuild worker (paired_launch.hip.h: paired_start_async_rebuild .. paired_finish_async_rebuild):
    a context is DESTROYED, and another one asked to compact its tables (a rebuild on the calling thread), while a worker
    rebuild is between "decided" and "taken over" -- during the sliced private copy (state 4: 140,000 pairs hold more
    active records than one slice copies) and while / after the worker thread builds (state 1 / 2, take-over far away).
    Nothing may abort (a joinable std::thread destroyed, buffers released under the worker), and the compacted context
    must go on giving the values of a context that never had a worker."""
    from gaml_amd import api
    G, n, seed = 900_000, 140_000, 57
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(1000, 6000)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    k = len(walk) // 2

    def make(knobs):
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *r1, *r2)
        for kk, v in knobs.items():
            c.debug_set_knob(kk, v)
        return c

    def into_flight(c):
        """tables built over the single-node windows, then every junction window activates at once: the delta lists pass
        the threshold and a worker rebuild is decided in that call"""
        c.calc_prob([[x] for x in walk])
        c.compact_tables()
        c.calc_prob([[x] for x in walk])
        before = c.debug_table_stats(0)
        v = c.calc_prob([walk])
        if phase == "worker":  # two more calls: the sliced copy completes, the worker thread starts (and may finish)
            c.calc_prob([walk[:k], walk[k:]])
            c.calc_prob([walk])
        st = c.debug_table_stats(0)
        assert st["dirty_pairs"] > 4096 and st["worker_rebuilds"] == before["worker_rebuilds"]  # decided, not taken over
        return v

    ref = make({14: 1})  # rebuilds on the calling thread only
    want = [ref.calc_prob(p) for p in ([[x] for x in walk], [walk], [walk[:k], walk[k:]], [walk])]
    # (1) destroy in flight, several times (the worker is at a different point each time)
    for _ in range(3):
        c = make({14: 100000})
        into_flight(c)
        c.close()
    # (2) compact in flight: the calling thread joins the worker, discards or uses its tables, rebuilds -- values unchanged
    c = make({14: 100000})
    v = into_flight(c)
    assert v[1].tolist() == want[1][1].tolist() and abs(v[0] - want[1][0]) <= 1e-12 * abs(want[1][0])
    c.compact_tables()
    for p, w in zip(([walk], [walk[:k], walk[k:]], [[x] for x in walk]), (want[1], want[2], want[0])):
        got = c.calc_prob(p)
        assert got[1].tolist() == w[1].tolist() and got[2] == w[2] and abs(got[0] - w[0]) <= 1e-12 * abs(w[0])
    assert c.debug_table_stats(0)["dirty_pairs"] < 4096  # (the lists were folded in; what the calls above activated is new)
    # (3) ... and a second worker rebuild right after that one (the thread object is reused)
    c.calc_prob([[x ^ 1 for x in reversed(walk)]])  # the twin walk: as many new windows again
    c.calc_prob([walk])
    c.compact_tables()
    got = c.calc_prob([walk])
    assert abs(got[0] - want[1][0]) <= 1e-12 * abs(want[1][0])
    c.close()
    ref.close()
