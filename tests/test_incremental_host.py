"""Incremental path-set planning on the host (no GPU): a call that shares a prefix / suffix of paths with the previous
one only looks at the paths in between. After every step of a long annealing-style walk of path sets
  * the occurrence TABLES (the host image the device copy mirrors) describe exactly the occurrences the planner's memos
    list for the set, and
  * a second context that plans every set from scratch (knob 12) has the same occurrences, windows and records."""
import numpy as np
import pytest

from gaml_amd import synth


def _canon_flat(flat):
    """debug_occurrences (path = position, rank global in visiting order) -> path-local ranks, table clamping"""
    out = flat.copy()
    for p in np.unique(flat[:, 3]):
        m = flat[:, 3] == p
        out[m, 4] = flat[m, 4] - flat[m, 4].min()
    return out


def _same_tables(ctx, rs):
    for mate in (0, 1):
        flat = _canon_flat(ctx.debug_occurrences(rs, mate))
        tab, info = ctx.debug_table_occurrences(rs, mate)
        a = flat[np.lexsort((flat[:, 0], flat[:, 4], flat[:, 3]))]
        b = tab[np.lexsort((tab[:, 0], tab[:, 4], tab[:, 3]))]
        assert a.shape == b.shape, (mate, a.shape, b.shape)
        # the 8-byte table form clamps the filter threshold at -32768 (positions are >= 0: filters like the exact value)
        a[:, 2] = np.maximum(a[:, 2], -32768); b[:, 2] = np.maximum(b[:, 2], -32768)
        assert np.array_equal(a, b), mate
    return info


@pytest.mark.parametrize("seed", [3, 11])
def test_incremental_tables_equal_planning_from_scratch(built, seed):
    from gaml_amd import api
    G, n = 90_000, 2500
    genome = synth.plant_repeats(synth.make_genome(G, seed), 3, 700, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 3500), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    inc, ref = api.Context(device=-1), api.Context(device=-1)
    ref.debug_set_knob(12, 1)  # every set planned from scratch
    for c in (inc, ref):
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *reads)
    rs = 0
    start, seq = synth.sa_sequence(g, 260, seed=seed, threshold=400)
    rng = np.random.default_rng(seed)
    sets = [start]
    for s in seq:
        sets.append(s)
        if rng.random() < 0.15:
            sets.append(sets[int(rng.integers(0, len(sets)))])  # jump back to an earlier set (a rejected move)
        if rng.random() < 0.05:
            sets.append([])  # an empty assembly
    used_incremental = 0
    for k, ps in enumerate(sets):
        inc.debug_prepare(ps)
        ref.debug_prepare(ps)
        info = _same_tables(inc, rs)
        info_ref = _same_tables(ref, rs)
        assert not info_ref["incremental"]
        used_incremental += info["incremental"]
        for mate in (0, 1):
            assert np.array_equal(_canon_flat(inc.debug_occurrences(rs, mate)), _canon_flat(ref.debug_occurrences(rs, mate))), (k, mate)
            assert inc.window_count(rs, mate) == ref.window_count(rs, mate)
    assert used_incremental > len(sets) // 2  # the walk really exercised the incremental path
    # same windows with the same records on both sides
    for mate in (0, 1):
        for wid in range(0, inc.window_count(rs, mate), 7):
            w = inc.debug_window_walk(rs, mate, wid)
            assert w == ref.debug_window_walk(rs, mate, wid)
            assert np.array_equal(inc.window_records(rs, mate, w), ref.window_records(rs, mate, w))


def test_candidates_of_one_assembly_stay_incremental(built):
    """Two candidates of one assembly differ from each other in BOTH their edits, with most of the set between the two
    (what a move generator's batch looks like, and every step of an annealing walk whose previous move was rejected):
    the diff is an edit script with several edits, not one prefix / suffix pair -- the planner must not fall back to
    planning the whole set, and the tables must equal planning from scratch."""
    from gaml_amd import api
    G, n, seed = 120_000, 2500, 19
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(500, 1500), short_rng=(25, 200)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    inc, ref = api.Context(device=-1), api.Context(device=-1)
    ref.debug_set_knob(12, 1)
    for c in (inc, ref):
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(240.0, 24.0), *reads)
    start, seq = synth.sa_sequence(g, 40, seed=seed, threshold=400)
    base = seq[-1]
    assert len(base) >= 40
    rng = np.random.default_rng(seed)
    def edit(at):  # break path `at` in two (one path out, two in), or drop it when it cannot be broken
        p = base[at]
        return base[:at] + ([p[: len(p) // 2], p[len(p) // 2:]] if len(p) >= 2 else []) + base[at + 1:]
    cands = [edit(2), edit(len(base) - 3), edit(len(base) // 2), edit(5), base, edit(len(base) - 2), edit(1)]
    inc.debug_prepare(base); ref.debug_prepare(base)
    for k, ps in enumerate(cands):
        inc.debug_prepare(ps); ref.debug_prepare(ps)
        info = _same_tables(inc, 0)
        assert info["incremental"], k
        for mate in (0, 1):
            assert np.array_equal(_canon_flat(inc.debug_occurrences(0, mate)), _canon_flat(ref.debug_occurrences(0, mate))), (k, mate)


def test_a_bad_path_leaves_the_planner_usable(built):
    from gaml_amd import api
    genome = synth.make_genome(30_000, 5)
    g = synth.make_graph(genome, synth.cut_lengths(30_000, 5, long_rng=(700, 2500)))
    pr = synth.make_paired_reads(genome, 400, 100, 240.0, 24.0, 0.01, 5)
    c = api.Context(device=-1)
    c.set_graph(*g.packed())
    c.add_paired(api.paired_cfg(240.0, 24.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    base = [walk[:4], walk[4:8], walk[8:12], walk[12:16], walk[16:]]
    c.debug_prepare(base)
    with pytest.raises(api.GamlHipError):
        c.debug_prepare(base[:2] + [walk[8:10] + [10_000_000]] + base[3:])  # node outside the graph, in the middle of a diff
    c.debug_prepare(base)
    assert not _same_tables(c, 0)["incremental"]  # after the error: from scratch
    c.debug_prepare(base[:2] + [walk[8:10], walk[10:12]] + base[3:])
    assert _same_tables(c, 0)["incremental"]
