"""Host logic of the product (window registration, the library's own aligner, occurrence
tables) against the oracle -- no GPU needed: the product side runs on a host-only context."""
import numpy as np
import pytest

import oracle_py as op
from gaml_amd import synth


def _pair_setup(G=80_000, n=4000, seed=21, repeats=3, L=100):
    from gaml_amd import api
    genome = synth.plant_repeats(synth.make_genome(G, seed), repeats, 600, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 4000), short_rng=(30, 330)))
    pr = synth.make_paired_reads(genome, n, L, 250.0, 25.0, 0.01, seed)
    gb, go = g.packed()
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    ctx = api.Context(device=-1)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(250.0, 25.0), b1, o1, b2, o2)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(250.0, 25.0))
    return g, ctx, rs, orc, ors


def _path_sets(g):
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    return [
        [walk],
        [walk[:k], walk[k:2 * k], walk[2 * k:]],
        [walk[:k] + [-200] + walk[k + 3:2 * k], [x ^ 1 for x in reversed(walk[2 * k:])]],
        [[x] for x in walk if g.node_len(x) > 500],
        [walk[5:9], walk[5:9], []],
        [[-30] + walk[:4] + [-10, -20] + walk[6:10]],
    ]


def test_same_windows_cached_with_identical_records(built):
    """After the same sequence of CalcProb calls both sides hold the same window cache (graph.cc:447-533)
    and every window holds identical Aligment records (graph.cc:839-899, 753-837)."""
    g, ctx, rs, orc, ors = _pair_setup()
    for paths in _path_sets(g):
        orc.calc_prob(paths, fresh=True)
        ctx.debug_prepare(paths)
        for mate in (0, 1):
            keys = orc.window_keys(ors, mate)
            assert ctx.window_count(rs, mate) == len(keys)
            for key in keys:
                ref = orc.window_records(ors, mate, key)
                got = ctx.window_records(rs, mate, key)
                assert got is not None, key
                assert got.shape == ref.shape and (got == ref).all(), key


def _assemble(ctx, rs, mate, path_index):
    """Numpy emulation of what the kernel does with the occurrence list: shift, filter by min_pos,
    later record at the same position overwrites (kept in first-seen slot)."""
    occ = ctx.debug_occurrences(rs, mate)
    per_read = {}
    for wid, shift, min_pos, path, rank in occ[np.argsort(occ[:, 4], kind="stable")]:
        if path != path_index:
            continue
        recs = ctx.window_records(rs, mate, ctx.debug_window_walk(rs, mate, int(wid)))
        for pos, edit, read, orient in recs:
            if pos < min_pos:
                continue
            lst = per_read.setdefault(int(read), [])
            for e in lst:
                if e[0] == pos + shift:
                    e[1], e[3] = int(edit), int(orient)
                    break
            else:
                lst.append([int(pos + shift), int(edit), int(read), int(orient)])
    return sorted([tuple(e) for lst in per_read.values() for e in lst])


def test_occurrence_tables_reproduce_GetPositionsOnlyPath(built):
    """Occurrence list + records == the reference's assembled positions, including the
    max_pos - 5 filter, which the product evaluates as a prefix maximum (graph.cc:535-598)."""
    g, ctx, rs, orc, ors = _pair_setup(G=60_000, n=3000, seed=22)
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    for ctg in (walk, walk[:k], walk[k:], [x ^ 1 for x in reversed(walk[:k])], walk[3:4]):
        orc.calc_prob([ctg], fresh=True)
        ctx.debug_prepare([ctg])
        for mate in (0, 1):
            want = sorted(tuple(int(v) for v in r) for r in orc.positions_only_path(ors, mate, ctg, 0))
            got = _assemble(ctx, rs, mate, 0)
            assert got == want


def test_occurrences_with_gaps_use_path_coordinates(built):
    g, ctx, rs, orc, ors = _pair_setup(G=40_000, n=1500, seed=23)
    walk = synth.genome_walk(g)
    path = walk[:6] + [-77] + walk[8:14]
    ctx.debug_prepare([path, walk[:3]])
    occ = ctx.debug_occurrences(rs, 0)
    assert set(occ[:, 3].tolist()) == {0, 1}
    second = sum(g.node_len(x) for x in walk[:6]) + 77
    shifts0 = occ[occ[:, 3] == 0][:, 1]
    assert shifts0.min() == 0 and second in shifts0.tolist()
    assert (np.diff(occ[:, 4]) == 1).all()  # ranks follow visiting order


def test_single_end_host_logic(built):
    from gaml_amd import api
    G, seed = 50_000, 31
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    reads = synth.make_single_reads(genome, 2500, 100, 0.01, seed)
    gb, go = g.packed()
    b, o = synth.pack_reads(reads)
    ctx = api.Context(device=-1)
    ctx.set_graph(gb, go)
    rs = ctx.add_single(api.single_cfg(), b, o)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_single(b, o, 0.01, op.single_cfg())
    walk = synth.genome_walk(g)
    for paths in ([walk], [walk[:7], walk[7:]], [[x] for x in walk]):
        orc.single_detail(ors, paths)
        ctx.debug_prepare(paths)
        keys = orc.window_keys(ors, 0)
        assert ctx.window_count(rs, 0) == len(keys)
        for key in keys:
            assert (ctx.window_records(rs, 0, key) == orc.window_records(ors, 0, key)).all()
    occ = ctx.debug_occurrences(rs, 0)
    # paths are 1,000,000 apart (graph.cc:1685): one singleton path per node here
    slots = set((occ[:, 1] // 1_000_000).tolist())
    assert (occ[:, 1] >= 0).all() and slots <= set(range(len(walk))) and len(slots) > len(walk) // 3


def test_put_window_records_is_an_alternative_to_the_internal_aligner(built):
    """Records handed in from outside (the reference's external-aligner branch, graph.cc:924-1033)
    are stored sorted by (position, read) and are not re-aligned."""
    from gaml_amd import api
    g, ctx, rs, orc, ors = _pair_setup(G=30_000, n=800, seed=24)
    walk = synth.genome_walk(g)
    key = walk[:1]
    recs = np.zeros(3, api.ALIGMENT)
    recs["position"], recs["edit_dist"], recs["read_id"], recs["orientation"] = [50, 7, 7], [1, 0, 2], [5, 9, 3], [0, 1, 0]
    ctx.put_window_records(rs, 0, key, recs)
    assert ctx.window_records(rs, 0, key).tolist() == [[7, 2, 3, 0], [7, 0, 9, 1], [50, 1, 5, 0]]
    ctx.debug_prepare([walk])
    assert ctx.window_records(rs, 0, key).tolist() == [[7, 2, 3, 0], [7, 0, 9, 1], [50, 1, 5, 0]]
    with pytest.raises(api.GamlHipError):
        ctx.put_window_records(rs, 0, key, recs)


def test_put_pacbio_records_rejects_what_the_sweep_cannot_take(built):
    """PacBio records from outside: a read id outside the set, or an alignment that ends before it begins (the
    reference's coverage sweep would close an interval that is not open, graph.cc:3229-3231)."""
    from gaml_amd import api
    genome = synth.make_genome(20_000, 3)
    g = synth.make_graph(genome, synth.cut_lengths(20_000, 3, long_rng=(700, 2500)))
    ctx = api.Context(device=-1)
    ctx.set_graph(*g.packed())
    rs = ctx.add_pacbio(api.single_cfg(mismatch_prob=0.15), np.full(10, 800, np.int32))
    walk = synth.genome_walk(g)[:2]
    ctx.put_pacbio_records(rs, walk, [(5, 700, 3), (40, 40, 4)], [-300.0, -280.0])  # (an empty interval is allowed)
    assert len(ctx.pacbio_records(rs, walk)) == 2
    for bad in ([(5, 700, 10)], [(5, 700, -1)], [(700, 5, 3)]):
        with pytest.raises(api.GamlHipError):
            ctx.put_pacbio_records(rs, walk, bad, [-300.0])
    assert len(ctx.pacbio_records(rs, walk)) == 2


def test_sharded_contexts_partition_the_records(built):
    from gaml_amd import api
    G, n, seed = 40_000, 2000, 25
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, 0.01, seed)
    gb, go = g.packed()
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    whole = api.Context(device=-1)
    whole.set_graph(gb, go)
    rs = whole.add_paired(api.paired_cfg(250.0, 25.0), *args)
    whole.debug_prepare([walk])
    shards = []
    for r in range(3):
        c = api.Context(device=-1, rank=r, world=3)
        c.set_graph(gb, go)
        c.add_paired(api.paired_cfg(250.0, 25.0), *args)
        c.debug_prepare([walk])
        shards.append(c)
    for wid in range(0, whole.window_count(rs, 0), 7):
        key = whole.debug_window_walk(rs, 0, wid)
        full = whole.window_records(rs, 0, key)
        parts = [c.window_records(0, 0, key) for c in shards]
        merged = np.concatenate([p for p in parts if p is not None and len(p)] or [np.zeros((0, 4), np.int32)])
        merged = merged[np.lexsort((merged[:, 2], merged[:, 0]))]
        assert (merged == full).all()
        for r, p in enumerate(parts):
            if p is not None and len(p):
                assert p[:, 2].min() >= n * r // 3 and p[:, 2].max() < n * (r + 1) // 3


def test_record_tables_leave_out_only_records_that_are_always_overwritten(built):
    """The record tables drop a junction window's record when the first node's own window (on the device itself) holds
    a record of the same read at the same position: that one is looked up right behind it at the same path position
    and overwrites it (graph.cc:563-566, 583-592). Host-only check over path sets that activate the windows in
    different orders: with the rule, every pair's records are the records without it minus exactly such records."""
    g, ctx, rs, _, _ = _pair_setup(G=120_000, n=12_000, seed=77, repeats=2)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    seen = []
    for paths in ([[x] for x in walk[:k]], [walk[:k]], [walk], [walk[k:], walk[:k]], [[x ^ 1 for x in reversed(walk)]]):
        ctx.debug_prepare(paths)
        st = ctx.debug_fold_check(rs)
        assert st["violations"] == 0 and st["records_checked"] > 0
        seen.append(st)
    assert seen[0]["records_left_out"] == [0, 0]                    # one-node paths: no junction window is in use
    assert min(seen[2]["records_left_out"]) > 0                      # the whole walk: every junction is
    assert seen[2]["compact_pairs"][0] > seen[2]["compact_pairs"][1]  # ... and pairs move to the one-record class


def test_static_memo_indices_follow_from_the_window_cache(built):
    """A compact-class pair whose two records sit in windows with the same node walk carries its memo index in the record
    tables (orientation rule and insert distance, graph.cc:1864-1876, on window positions: both alignments get the
    window's shift wherever it occurs); a pair with an unaligned mate carries "never scores". Host-only: every index
    recomputed from the window cache, every pair without one has a reason, over path sets that activate the windows in
    different orders (incl. the inverted walk, whose windows are other walks)."""
    g, ctx, rs, _, _ = _pair_setup(G=120_000, n=12_000, seed=78, repeats=2)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    seen = []
    for paths in ([[x] for x in walk[:k]], [walk[:k]], [walk], [walk[k:], walk[:k]], [[x ^ 1 for x in reversed(walk)]]):
        ctx.debug_prepare(paths)
        st = ctx.debug_static_check(rs)
        assert st["violations"] == 0
        seen.append(st)
    whole = seen[2]
    assert whole["static_pairs"] > 10 * whole["other_pairs"] > 0
    assert whole["other_pairs"] == whole["different_windows"] + whole["orientation"] + whole["distance"] + whole["edits_or_code"]
