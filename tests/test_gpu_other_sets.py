"""GPU parity of the single-end and PacBio scorers, mixed read sets (ProbCalculator's sum),
golden fixtures through the C ABI, and read sharding on one GPU."""
import json
import math
import os
import sys

import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_pins.json")))


def h(x):
    return float.fromhex(x)


def _graph(G, seed, **kw):
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, **kw))
    return genome, g


def test_single_end_parity():
    from gaml_amd import api
    import oracle_py as op
    genome, g = _graph(60_000, 51, long_rng=(700, 3000))
    reads = synth.make_single_reads(genome, 4000, 100, 0.01, 51)
    gb, go = g.packed()
    b, o = synth.pack_reads(reads)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_single(api.single_cfg(), b, o)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_single(b, o, 0.01, op.single_cfg())
    walk = synth.genome_walk(g)
    for paths in ([walk], [walk[:9], walk[9:]], [[x] for x in walk], [walk[:5] + [-60] + walk[7:]], [walk[2:5], walk[2:5]]):
        got, zeros, tl = ctx.calc_prob(paths)
        want, wprobs, o3 = orc.single_detail(ors, paths)
        assert zeros.tolist() == [[int(o3[0]), 4000]] and tl == int(o3[1])
        np.testing.assert_allclose(ctx.read_probs(rs), wprobs, rtol=4e-16, atol=0)
        assert ctx.bad_bases(rs) == int(o3[2]) == 0
        assert abs(got - want) <= 1e-12 * abs(want)


def _pacbio_pair(g, walk, n_reads, cfg_kw, seed=4, read_len=2000):
    from gaml_amd import api
    import oracle_py as op
    gb, go = g.packed()
    pb = synth.make_pacbio_records(g, walk, n_reads, read_len, 0.15, seed)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_pacbio(api.single_cfg(mismatch_prob=0.15, **cfg_kw), pb.lens)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    okw = dict(cfg_kw)
    if "penalty_step" in okw:
        okw["step"] = okw.pop("penalty_step")
    ors = orc.add_pacbio(pb.lens, 0.15, op.single_cfg(**okw))
    for wk, rec, lp in zip(pb.walks, pb.recs, pb.logps):
        ctx.put_pacbio_records(rs, wk, rec, lp)
        orc.pacbio_put(ors, wk, rec, lp)
    return ctx, rs, orc, ors, pb


def test_pacbio_parity_and_golden():
    import make_golden as mg
    genome, g, pr = mg.tiny_case()
    walk = synth.genome_walk(g)
    for tag in ("pacbio", "pacbio_sparse"):
        c = PINS[tag]
        ctx, rs, orc, ors, pb = _pacbio_pair(g, walk, c["n_reads"], dict(penalty_constant=0.0001, min_prob_per_base=-1.06))
        got, zeros, tl = ctx.calc_prob([walk])
        want, wlp, o3 = orc.pacbio_detail(ors, [walk])
        assert zeros.tolist() == [[c["zeros"], c["n_reads"]]] and tl == c["total_len"]
        assert ctx.bad_bases(rs) == c["bad_bases"] == int(o3[2])
        lp = ctx.read_probs(rs)
        fin = np.isfinite(wlp)
        assert (np.isfinite(lp) == fin).all()
        np.testing.assert_allclose(lp[fin], wlp[fin], rtol=1e-13)
        assert abs(got - h(c["prob"])) <= 1e-12 * abs(h(c["prob"]))
        assert abs(got - want) <= 1e-12 * abs(want)


@pytest.mark.parametrize("step", [30.0, 0.5, 777.25])
def test_pacbio_coverage_sweep_on_crafted_intervals(step):
    """The device sweep (sort + running maximum + binary search, pacbio_sweep.hip.h) against the oracle's restatement of
    the reference's event / multiset loop (graph.cc:3198-3250) on intervals chosen to hit its corners: nested and
    identical intervals, intervals that touch (one ends where the next begins), begins equal to node boundaries, long
    uncovered stretches, records below GetMinReadProb (not counted), several contigs, a gap in a contig."""
    from gaml_amd import api
    import oracle_py as op
    genome, g = _graph(120_000, 97, long_rng=(2500, 6000))
    walk = synth.genome_walk(g)
    gb, go = g.packed()
    rng = np.random.default_rng(5)
    n_reads = 400
    lens = np.full(n_reads, 1500, np.int32)
    cfg = dict(penalty_constant=0.002, min_prob_per_base=-1.2)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_pacbio(api.single_cfg(mismatch_prob=0.15, penalty_step=step, **cfg), lens)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_pacbio(lens, 0.15, op.single_cfg(step=step, **cfg))
    node_len = lambda x: int(go[x + 1] - go[x])
    good_lp, bad_lp = -900.0, -5000.0  # GetMinReadProb(1500 bases at 15 %) is about -1130: the second kind is not counted
    rid = 0
    for sub in synth.all_subwalks_for_pacbio(g, walk, 1500):
        span = sum(node_len(x) for x in sub)
        recs, lps = [], []
        k = int(rng.integers(0, 5))
        anchors = rng.integers(0, max(1, span - 10), size=k)
        for a in anchors:
            a = int(a)
            e = int(min(span + 40, a + rng.integers(1, 1800)))
            for rep in range(int(rng.integers(1, 3))):  # identical copies
                recs.append((a, e, rid % n_reads)); lps.append(good_lp if rng.random() < 0.8 else bad_lp); rid += 1
            if rng.random() < 0.5:  # one that starts exactly where this one ends, one nested inside
                recs.append((e, e + int(rng.integers(1, 900)), rid % n_reads)); lps.append(good_lp); rid += 1
                if e - a > 4:
                    recs.append((a + 1, e - 1, rid % n_reads)); lps.append(good_lp); rid += 1
        rec = np.array(recs, np.int32).reshape(-1, 3)
        lp = np.array(lps, np.float64)
        ctx.put_pacbio_records(rs, sub, rec, lp)
        orc.pacbio_put(ors, sub, rec, lp)
    for paths in ([walk], [walk[:6], walk[6:]], [walk[:4] + [-700] + walk[5:11], walk[11:], walk[2:3]]):
        got, zeros, tl = ctx.calc_prob(paths)
        want, wlp, o3 = orc.pacbio_detail(ors, paths)
        assert int(o3[2]) > 0 or step < 1  # (int(begin + 0.5) never passes the position: nothing is bad)
        assert ctx.bad_bases(rs) == int(o3[2]), (step, paths)
        assert abs(got - want) <= 1e-12 * abs(want)


def test_pacbio_many_alignments_per_read_wave_lse():
    """A read with hundreds of candidate positions exercises the strided wave-level log-sum-exp."""
    from gaml_amd import api
    import oracle_py as op
    genome, g = _graph(40_000, 53, long_rng=(900, 2500))
    gb, go = g.packed()
    walk = synth.genome_walk(g)[:6]
    rng = np.random.default_rng(5)
    lens = np.array([1500, 1500, 1800], np.int32)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    rs = ctx.add_pacbio(api.single_cfg(mismatch_prob=0.15, min_prob_per_base=-2.0), lens)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_pacbio(lens, 0.15, op.single_cfg(min_prob_per_base=-2.0))
    for sub in synth.all_subwalks_for_pacbio(g, walk, int(lens.max())):
        k = int(rng.integers(0, 200))
        rec = np.stack([rng.integers(0, 500, k), rng.integers(1500, 2000, k), rng.integers(0, 2, k)], axis=1).astype(np.int32)
        lp = rng.uniform(-2400, -1700, k)
        ctx.put_pacbio_records(rs, sub, rec, lp)
        orc.pacbio_put(ors, sub, rec, lp)
    got, zeros, tl = ctx.calc_prob([walk, walk[1:4]])
    want, wlp, o3 = orc.pacbio_detail(ors, [walk, walk[1:4]])
    lp = ctx.read_probs(rs)
    assert math.isinf(lp[2]) and math.isinf(wlp[2])  # read 2 has no alignment: floored
    np.testing.assert_allclose(lp[:2], wlp[:2], rtol=1e-12)
    assert zeros.tolist() == [[int(o3[0]), 3]]
    assert abs(got - want) <= 1e-12 * abs(want)


def test_mixed_read_sets_sum_like_prob_calculator():
    """Two readsets as BASELINE config 4: paired (weight 1) + PacBio (weight 0.5, mismatch 0.15)."""
    from gaml_amd import api
    import oracle_py as op
    genome, g = _graph(90_000, 55, long_rng=(900, 4000))
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    pr = synth.make_paired_reads(genome, 5000, 150, 300.0, 30.0, 0.01, 55)
    pb = synth.make_pacbio_records(g, walk, 120, 5000, 0.15, 55)
    sr = synth.make_single_reads(genome, 1500, 100, 0.01, 56)
    ctx = api.Context(device=0)
    ctx.set_graph(gb, go)
    orc = op.Oracle()
    orc.set_graph(gb, go)
    # creation order pacbio, paired, single: the value and `zeros` must come out single, paired, pacbio
    crs = ctx.add_pacbio(api.single_cfg(mismatch_prob=0.15, weight=0.5, min_prob_per_base=-1.1), pb.lens)
    ors = orc.add_pacbio(pb.lens, 0.15, op.single_cfg(weight=0.5, min_prob_per_base=-1.1))
    for wk, rec, lp in zip(pb.walks, pb.recs, pb.logps):
        ctx.put_pacbio_records(crs, wk, rec, lp)
        orc.pacbio_put(ors, wk, rec, lp)
    ctx.add_paired(api.paired_cfg(300.0, 30.0, weight=1.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    orc.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(300.0, 30.0, weight=1.0))
    ctx.add_single(api.single_cfg(weight=0.25), *synth.pack_reads(sr))
    orc.add_single(*synth.pack_reads(sr), 0.01, op.single_cfg(weight=0.25))
    k = len(walk) // 2
    for paths in ([walk], [walk[:k], walk[k:]]):
        got, zeros, tl = ctx.calc_prob(paths)
        want, wz, wtl = orc.calc_prob(paths, fresh=True)
        assert zeros[:, 1].tolist() == [1500, 5000, 120]
        # sub-walks no synthetic record was filed under are cache misses on both sides (BLASR territory)
        assert zeros.tolist() == wz.tolist() and tl == wtl
        assert abs(got - want) <= 1e-9 * abs(want)


def test_paired_golden_through_c_abi():
    import make_golden as mg
    from gaml_amd import api
    genome, g, pr = mg.tiny_case()
    gb, go = g.packed()
    for pen_name, pen in (("nopenalty", 0.0), ("penalty", 0.0002)):
        ctx = api.Context(device=0)
        ctx.set_graph(gb, go)
        rs = ctx.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=pen), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        for name, c in PINS[f"paired_{pen_name}"]["cases"].items():
            got, zeros, tl = ctx.calc_prob(c["paths"])
            assert zeros.tolist() == c["zeros"] and tl == c["total_len"], name
            assert ctx.bad_bases(rs) == (c["bad_bases"] if pen > 0 else 0)
            assert abs(got - h(c["prob"])) <= 1e-12 * abs(h(c["prob"])), name
            probs = ctx.read_probs(rs)
            assert int((probs > 0).sum()) == c["probs_nonzero"]
            assert abs(float(probs.sum()) - h(c["probs_sum"])) <= 1e-13 * h(c["probs_sum"])


def test_fastq_and_lastgraph_files_end_to_end(tmp_path):
    """Same inputs as files in the reference's formats (LastGraph, FASTQ) vs arrays."""
    from gaml_amd import api
    genome, g = _graph(40_000, 57, long_rng=(900, 3000))
    pr = synth.make_paired_reads(genome, 1500, 100, 250.0, 25.0, 0.01, 57)
    synth.write_lastgraph(str(tmp_path / "LastGraph"), g)
    synth.write_fastq(str(tmp_path / "r_1.fastq"), pr.mate1, "p", 1)
    synth.write_fastq(str(tmp_path / "r_2.fastq"), pr.mate2, "p", 2)
    a = api.Context(device=0)
    a.load_graph(str(tmp_path / "LastGraph"))
    a.add_paired_fastq(api.paired_cfg(250.0, 25.0), str(tmp_path / "r_1.fastq"), str(tmp_path / "r_2.fastq"))
    b = api.Context(device=0)
    b.set_graph(*g.packed())
    b.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    assert a.num_nodes() == b.num_nodes() == g.n_nodes
    assert a.calc_prob([walk]) [0] == b.calc_prob([walk])[0]


def test_read_sharding_on_one_gpu_sums_to_the_whole():
    """world=3 shards scored one after the other on the same GPU: partial sums add up to the
    unsharded partials (the only cross-rank step is a sum)."""
    from gaml_amd import api
    genome, g = _graph(80_000, 59, long_rng=(900, 4000))
    pr = synth.make_paired_reads(genome, 5001, 150, 300.0, 30.0, 0.01, 59)
    gb, go = g.packed()
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    paths = [walk[:7], walk[7:]]
    whole = api.Context(device=0)
    whole.set_graph(gb, go)
    whole.add_paired(api.paired_cfg(300.0, 30.0), *args)
    wp, tl = whole.calc_partials(paths)
    want, wz, _ = whole.calc_prob(paths)
    acc = np.zeros(4)
    probs = []
    shards = []
    for r in range(3):
        c = api.Context(device=0, rank=r, world=3)
        c.set_graph(gb, go)
        c.add_paired(api.paired_cfg(300.0, 30.0), *args)
        shards.append(c)
    # cold evaluation: windows get aligned, so the shards must exchange each new window's largest
    # record position (the all-reduce(max) a multi-process run does over RCCL)
    pend = [c.eval_begin(paths) for c in shards]
    assert len({p for p, _ in pend}) == 1 and pend[0][0] > 0 and all(t == tl for _, t in pend)
    with pytest.raises(api.GamlHipError):
        shards[0].eval_finish()  # refuses to score with unexchanged maxima
    reduced = np.maximum.reduce([c.eval_pending_maxpos() for c in shards])
    for r, c in enumerate(shards):
        c.eval_apply_maxpos(reduced)
        p = c.eval_finish()
        acc += p[0]
        probs.append(c.read_probs(0)[: 5001 * (r + 1) // 3 - 5001 * r // 3])
        last = c
    # warm cache: nothing to exchange, the one-shot call works on a sharded context
    for c in shards:
        assert c.eval_begin(paths)[0] == 0
        c.eval_finish()
    p2, _ = shards[1].calc_partials(paths)
    assert acc[1] == wp[0][1] and acc[3] == wp[0][3] == 5001
    assert abs(acc[0] - wp[0][0]) <= 1e-12 * abs(wp[0][0])
    assert np.array_equal(np.concatenate(probs), whole.read_probs(0))
    acc[2] = 0.0
    got, z = last.combine_partials(acc, tl)
    assert z.tolist() == wz.tolist() and abs(got - want) <= 1e-12 * abs(want)


def test_sharded_context_refuses_coverage_penalty():
    from gaml_amd import api
    genome, g = _graph(30_000, 61, long_rng=(900, 3000))
    pr = synth.make_paired_reads(genome, 600, 100, 250.0, 25.0, 0.01, 61)
    c = api.Context(device=0, rank=0, world=2)
    c.set_graph(*g.packed())
    c.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=0.001), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    pending, _ = c.eval_begin([synth.genome_walk(g)])
    c.eval_apply_maxpos(c.eval_pending_maxpos())
    with pytest.raises(api.GamlHipError) as e:
        c.eval_finish()
    assert e.value.code == api.ESTATE and "coverage" in str(e.value)


@pytest.mark.parametrize("stream_kind", ["explicit", "default"])
def test_sharded_coverage_penalty_merges_the_ranks_maps(stream_kind):
    """penalty_constant > 0 on a sharded paired set (the one non-separable piece): bad_bases
    depends on the union of all ranks' coverage marks. Three shards on one GPU play the ranks; the
    all-gather a multi-process run does over RCCL is a torch.cat here. Low coverage, so that every
    shard alone sees gaps the union does not have."""
    import torch
    from gaml_amd import api
    genome, g = _graph(60_000, 67, long_rng=(900, 4000))
    pr = synth.make_paired_reads(genome, 1500, 100, 250.0, 25.0, 0.01, 67)
    gb, go = g.packed()
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    paths = [walk[:9], walk[9:]]
    cfg = dict(penalty_constant=0.0007, penalty_step=40.0)
    whole = api.Context(device=0)
    whole.set_graph(gb, go)
    whole.add_paired(api.paired_cfg(250.0, 25.0, **cfg), *args)
    want, wz, tl = whole.calc_prob(paths)
    bad_whole = whole.bad_bases(0)
    assert bad_whole > 0
    shards = []
    for r in range(3):
        c = api.Context(device=0, rank=r, world=3)
        c.set_graph(gb, go)
        c.add_paired(api.paired_cfg(250.0, 25.0, **cfg), *args)
        shards.append(c)
    pend = [c.eval_begin(paths) for c in shards]
    reduced = np.maximum.reduce([c.eval_pending_maxpos() for c in shards])
    for c in shards:
        c.eval_apply_maxpos(reduced)
    # ONE stream for the library's work and for torch's: an explicit one, or torch's default stream -- whose handle is 0,
    # the legacy default stream, which the library uses as such. (Until round 3 a null handle made every context fall
    # back to its OWN private stream, unordered against torch's copies -- the three "ranks" share one process here --
    # and this test read 2771 bad bases instead of 694.)
    ts = torch.cuda.Stream() if stream_kind == "explicit" else torch.cuda.default_stream()
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    assert (stream == 0) == (stream_kind == "default")
    parts = [torch.zeros(4, dtype=torch.float64, device="cuda") for _ in shards]
    maps = []
    for c, p in zip(shards, parts):
        assert c.eval_score_async(p.data_ptr(), stream) == 1
        nbytes = c.eval_coverage_bytes(0)
        m = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
        assert c.eval_coverage_export_async(0, m.data_ptr(), nbytes, stream) == nbytes
        maps.append(m)
    assert len({m.numel() for m in maps}) == 1
    gathered = torch.cat(maps)  # what all_gather_into_tensor leaves on every rank
    for r, (c, p) in enumerate(zip(shards, parts)):
        c.eval_coverage_finish_async(0, gathered.data_ptr(), 3, r == 0, stream)
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    own_bad = [float(p[2]) for p in parts]
    assert own_bad[0] == bad_whole and own_bad[1] == 0.0 and own_bad[2] == 0.0
    acc = torch.stack(parts).sum(0).cpu().numpy()  # the all-reduce(sum)
    got, z = shards[2].combine_partials(acc, tl)
    assert z.tolist() == wz.tolist() and abs(got - want) <= 1e-12 * abs(want)
    # each shard alone would have counted more uncovered bases
    lone = api.Context(device=0)
    lone.set_graph(gb, go)
    n = pr.mate1.shape[0]
    lone.add_paired(api.paired_cfg(250.0, 25.0, **cfg), *synth.pack_reads(pr.mate1[: n // 3]), *synth.pack_reads(pr.mate2[: n // 3]))
    lone.calc_prob(paths)
    assert lone.bad_bases(0) > bad_whole
    # without the exchange the sharded context still refuses
    c = shards[0]
    c.eval_begin(paths)
    with pytest.raises(api.GamlHipError):
        c.eval_finish()


@pytest.mark.parametrize("stream_kind", ["explicit", "default"])
def test_sharded_pacbio_penalty_merges_the_ranks_intervals(stream_kind):
    """A PacBio set with penalty_constant > 0 on sharded contexts: bad_bases (graph.cc:3198-3250) sweeps the
    alignment intervals of ALL reads, so the ranks exchange their interval lists (device memory). Three shards on one GPU
    play the ranks; sparse long reads leave uncovered stretches that only the union closes."""
    import torch
    from gaml_amd import api
    genome, g = _graph(60_000, 77, long_rng=(1500, 5000))
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, 45, 2500, 77, secondary=0.0)
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    cfg = api.single_cfg(penalty_constant=0.003, penalty_step=30.0, min_prob_per_base=-1.0, mismatch_prob=0.15)
    paths = [walk[:7], walk[7:]]

    def make(rank=0, world=1):
        c = api.Context(device=0, rank=rank, world=world)
        c.set_graph(*g.packed())
        rs = c.add_pacbio_reads(cfg, rb, ro, ps.names)
        c.pacbio_ingest_sam(rs, walk, ps.sam)  # every rank files its own reads' records
        return c, rs
    whole, wrs = make()
    want, wz, tl = whole.calc_prob(paths)
    bad_whole = whole.bad_bases(wrs)
    assert bad_whole > 0
    shards = [make(r, 3)[0] for r in range(3)]
    ts = torch.cuda.Stream() if stream_kind == "explicit" else torch.cuda.default_stream()  # (see the coverage test above)
    torch.cuda.set_stream(ts)
    stream = ts.cuda_stream
    parts = [torch.zeros(4, dtype=torch.float64, device="cuda") for _ in shards]
    own, counts = [], []
    for c, p in zip(shards, parts):
        c.eval_begin(paths)
        assert c.eval_score_async(p.data_ptr(), stream) == 0 and c.eval_pacbio_pending() == 1
        n = c.eval_pacbio_intervals(0)  # known on the host; the intervals themselves stay on the device
        t = torch.zeros(4 * max(1, n), dtype=torch.int32, device="cuda")
        c.eval_pacbio_export_async(0, t.data_ptr(), max(1, n), stream)
        own.append(t[: 4 * n])
        counts.append(n)
    assert sum(counts) > 0
    merged = torch.cat(own).contiguous()  # what the all-gather leaves on every rank (device memory)
    iv = merged.view(-1, 4).cpu().numpy()
    assert set(iv[:, 0].tolist()) <= {0, 1} and (iv[:, 2] > iv[:, 1]).all() and (iv[:, 3] == 0).all()
    for r, c in enumerate(shards):
        c.eval_pacbio_finish_async(0, merged.data_ptr(), sum(counts), r == 1, stream)  # any one rank may contribute
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    assert [float(p[2]) for p in parts] == [0.0, float(bad_whole), 0.0]
    acc = torch.stack(parts).sum(0).cpu().numpy()
    got, z = shards[0].combine_partials(acc, tl)
    assert z.tolist() == wz.tolist() and abs(got - want) <= 1e-12 * abs(want)
    c = shards[1]
    c.eval_begin(paths)
    with pytest.raises(api.GamlHipError):
        c.eval_finish()  # without the exchange the sharded context refuses


def test_external_aligner_records_score_like_the_internal_aligner():
    """The reference can take its per-window alignments from an external aligner (Bowtie2 branch,
    graph.cc:924-1033); gaml_hip_put_window_records is that entry. A context fed every window's records from
    outside (here: copied from a context that aligned them itself) must score identically and align nothing."""
    from gaml_amd import api
    genome, g = _graph(90_000, 83, long_rng=(800, 4000))
    pr = synth.make_paired_reads(genome, 7000, 100, 250.0, 25.0, 0.01, 83)
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    walk = synth.genome_walk(g)
    paths = [walk[:12], walk[12:]]
    own = api.Context(device=0)
    own.set_graph(*g.packed())
    rs = own.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=0.0002), *args)
    want = own.calc_prob(paths)
    fed = api.Context(device=0)
    fed.set_graph(*g.packed())
    rs2 = fed.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=0.0002), *args)
    n_put = 0
    for mate in (0, 1):
        for wid in range(own.window_count(rs, mate)):
            key = own.debug_window_walk(rs, mate, wid)
            r = own.window_records(rs, mate, key)
            recs = np.zeros(len(r), api.ALIGMENT)
            if len(r):
                recs["position"], recs["edit_dist"], recs["read_id"], recs["orientation"] = r[:, 0], r[:, 1], r[:, 2], r[:, 3]
            fed.put_window_records(rs2, mate, key, recs)
            n_put += 1
    assert n_put > 20
    got = fed.calc_prob(paths)
    assert fed.aligner_stats()["windows"] == 0  # nothing was aligned by the library
    assert got[2] == want[2] and got[1].tolist() == want[1].tolist()
    assert abs(got[0] - want[0]) <= 1e-13 * abs(want[0])
    assert fed.bad_bases(rs2) == own.bad_bases(rs)
    np.testing.assert_array_equal(fed.read_probs(rs2), own.read_probs(rs))
