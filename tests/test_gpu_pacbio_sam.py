"""PacBio cache-miss side on the GPU: SAM text -> banded DP -> filed records (reference
graph.cc:2650-2795, 2175-2297, 2945-3021) against the oracle's restatement of the same steps.

Tolerance: the DP's log probabilities agree to 1e-9 relative (the GPU solves the left-neighbour
recurrence of a row by a scan, which re-associates the logdouble additions); positions, read ids
and the set of cached sub-walks are exact."""
import numpy as np
import pytest

from gaml_amd import api, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu
MISMATCH = 0.15
RTOL = 1e-9
PER_BASE = -1.0  # floor c + k L below a true alignment (about -0.76 per base at 15 % errors), above a wrong one


def _world(genome_len=30000, seed=9, n_reads=120, read_len=900):
    gen = synth.make_genome(genome_len, seed)
    g = synth.make_graph(gen, synth.cut_lengths(genome_len, seed, long_rng=(1500, 4000)))
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, n_reads, read_len, seed)
    return g, walk, ps


def _both(g, ps, world=None, rank=0, penalty=0.0):
    bases, offs = g.packed()
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    orc = O.Oracle()
    orc.set_graph(bases, offs)
    ors = orc.add_pacbio_reads(rb, ro, ps.names, MISMATCH, O.single_cfg(penalty_constant=penalty, min_prob_per_base=PER_BASE))
    ctx = api.Context(device=0, rank=rank, world=world) if world else api.Context()
    ctx.set_graph(bases, offs)
    prs = ctx.add_pacbio_reads(api.single_cfg(penalty_constant=penalty, min_prob_per_base=PER_BASE, mismatch_prob=MISMATCH), rb, ro, ps.names)
    return orc, ors, ctx, prs


def _compare_caches(orc, ors, ctx, prs):
    keys = orc.pacbio_keys(ors)
    n = 0
    for k in keys:
        rec, lp = orc.pacbio_records(ors, list(k))
        got = ctx.pacbio_records(prs, list(k))
        assert got is not None, k
        assert len(got) == len(rec), k
        # the reference files records in SAM order; so do both sides
        assert np.array_equal(got["position"], rec[:, 0]) and np.array_equal(got["position_end"], rec[:, 1])
        assert np.array_equal(got["read_id"], rec[:, 2])
        np.testing.assert_allclose(got["logprob"], lp, rtol=RTOL, atol=0)
        n += len(rec)
    return len(keys), n


@pytest.mark.parametrize("penalty", [0.0, 0.02])
def test_ingest_matches_oracle_and_scores_equal(penalty):
    g, walk, ps = _world()
    orc, ors, ctx, prs = _both(g, ps, penalty=penalty)
    assert ctx.pacbio_missing(prs, walk) == [(0, len(walk) - 1)]
    filed_o = orc.pacbio_ingest_sam(ors, walk, ps.sam)
    filed_p = ctx.pacbio_ingest_sam(prs, walk, ps.sam)
    assert filed_p == filed_o and filed_o > 0.8 * ps.n_records
    nk, nrec = _compare_caches(orc, ors, ctx, prs)
    assert nrec == filed_o and nk > len(walk)
    assert ctx.pacbio_missing(prs, walk) == []
    st = ctx.pacbio_dp_stats(prs)
    assert st["jobs"] == filed_p and st["records"] == ps.n_records
    # the likelihood of the path from the freshly filled cache
    for paths in ([walk], [walk[:5], walk[5:]]):
        po = orc.calc_prob(paths)
        pp = ctx.calc_prob(paths)
        assert abs(pp[0] - po[0]) <= RTOL * abs(po[0])
        assert np.array_equal(pp[1], po[1]) and pp[2] == po[2]


def test_second_ingest_leaves_cached_subwalks_alone():
    g, walk, ps = _world(seed=4, n_reads=60)
    orc, ors, ctx, prs = _both(g, ps)
    head = walk[:6]
    # BLASR run on the first half only: records outside are not filed (their sub-walk is unknown)
    ps_head = synth.make_pacbio_sam(g, head, 30, 900, 21)
    # read names must exist in the set: reuse the generator's names r0..r29 with the main reads
    ps_head_sam = ps_head.sam
    rb_ok = all(n in ps.names for n in ps_head.names)
    assert rb_ok
    a = orc.pacbio_ingest_sam(ors, head, ps_head_sam)
    b = ctx.pacbio_ingest_sam(prs, head, ps_head_sam)
    assert a == b
    miss = ctx.pacbio_missing(prs, walk)
    assert miss and miss[0][0] > 0  # the head is cached now; what is missing starts later
    a2 = orc.pacbio_ingest_sam(ors, walk, ps.sam)
    b2 = ctx.pacbio_ingest_sam(prs, walk, ps.sam)
    assert a2 == b2 and a2 < ps.n_records  # records spanning head-only sub-walks are not saved twice
    _compare_caches(orc, ors, ctx, prs)
    po, pp = orc.calc_prob([walk]), ctx.calc_prob([walk])
    assert abs(pp[0] - po[0]) <= RTOL * abs(po[0])


def test_path_with_gap_and_tiny_nodes():
    g, walk, ps0 = _world(seed=13, n_reads=10)
    gapped = walk[:3] + [-57] + walk[4:8]
    ps = synth.make_pacbio_sam(g, gapped, 40, 700, 5)
    orc, ors, ctx, prs = _both(g, ps)
    assert orc.pacbio_ingest_sam(ors, gapped, ps.sam) == ctx.pacbio_ingest_sam(prs, gapped, ps.sam)
    _compare_caches(orc, ors, ctx, prs)
    po, pp = orc.calc_prob([gapped]), ctx.calc_prob([gapped])
    assert abs(pp[0] - po[0]) <= RTOL * abs(po[0])


QUIRKS = [
    "q/1\t0\tp\t10\t1\t5S10M\t*\t0\t10\tACGTACGTAC\t*",
    "q/1\t16\tp\t10\t1\t3I4M2D3M\t*\t0\t9\tACGTACGTAC\t*\tNM:i:3",
    "q/1\t0\tp\t3\t1\t10I\t*\t0\t0\tACGTACGTAC\t*",
    "q/1\t0\tp\t3\t1\t*\t*\t0\t0\tACGTACGTAC\t*",
    "q/1\t0\tp\t2\t1\t5M\t*\t0\t5\tACGTA\t*\tXS:i:300\tXE:i:305\tXQ:i:700",
    "q/1\t16\tp\t100\t1\t5M0D3M\t*\t0\t8\tACGTACGT\t*\tXS:i:4\tXE:i:12\tXQ:i:250",
    "q/1\t0\tp\t100\t1\t2M250I3M\t*\t0\t5\tACGTA\t*",
    "q/1\t0\tp\t0\t1\t8M\t*\t0\t8\tACGTACGT\t*",            # row 1 at the first base
    "q/1\t0\tp\t395\t1\t8M\t*\t0\t8\tACGTACGT\t*",          # runs into the separator
    "q/1\t16\tp\t0\t1\t8M\t*\t0\t8\tACGTACGT\t*",           # mirrored to the very end of the string
    "q/1\t0\tp\t50\t1\t4M3D4M\t*\t0\t11\tACGTACGT\t*\tXS:i:1\tXE:i:9\tXQ:i:8",
]


def test_dp_kernel_on_odd_records():
    rng = np.random.default_rng(3)
    half = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, 400))
    target = half + "\n" + synth.revcomp_str(half)
    ctx = api.Context()
    for line in QUIRKS:
        f = line.split("\t")
        n = 700 if "XQ:i:700" in line else 250 if "XQ:i:250" in line else 255 if "250I" in line else len(f[9])
        read = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))
        want = O.sam_alignment_logprob(line, target, read, MISMATCH)
        got, lo, hi = ctx.debug_sam_logprob(target, read, line, MISMATCH, with_band=True)
        # the band the kernel derives from the run-length CIGAR is the oracle's cell set
        _, _, olo, ohi = O.sam_band(line, len(target))
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), line
        if np.isinf(want):
            assert got == want, line
        else:
            assert abs(got - want) <= RTOL * abs(want), (line, got, want)
    # a read cut from the target with a true edit script, both strands, both lane widths
    g = synth.make_graph(np.frombuffer(half.encode(), np.uint8), [400])
    ps = synth.make_pacbio_sam(g, [0], 30, 300, 8)
    rd = dict(zip(ps.names, ps.reads))
    for line in ps.sam.split("\n")[1:-1]:
        read = rd[line.split("\t")[0].split("/")[0]]
        want = O.sam_alignment_logprob(line, target, read, MISMATCH)
        got, lo, hi = ctx.debug_sam_logprob(target, read, line, MISMATCH, with_band=True)
        _, _, olo, ohi = O.sam_band(line, len(target))
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi), line
        assert abs(got - want) <= RTOL * abs(want), (line, got, want)


def test_sharded_ingest_files_each_read_once():
    g, walk, ps = _world(seed=2, n_reads=90)
    orc, ors, full, prs = _both(g, ps)
    orc.pacbio_ingest_sam(ors, walk, ps.sam)
    filed_full = full.pacbio_ingest_sam(prs, walk, ps.sam)
    want = full.calc_prob([walk])
    total, parts = 0, []
    for rank in range(3):
        _, _, ctx, rs = _both(g, ps, world=3, rank=rank)
        total += ctx.pacbio_ingest_sam(rs, walk, ps.sam)
        parts.append(ctx.calc_partials([walk])[0])
    assert total == filed_full
    prob, zeros = full.combine_partials(np.sum(parts, axis=0), want[2])
    assert abs(prob - want[0]) <= 1e-12 * abs(want[0]) and np.array_equal(zeros, want[1])


def test_errors_are_loud():
    g, walk, ps = _world(seed=6, n_reads=5)
    orc, ors, ctx, prs = _both(g, ps)
    with pytest.raises(api.GamlHipError):
        ctx.pacbio_ingest_sam(prs, walk, "nosuchread/0_5\t0\tp\t5\t1\t5M\t*\t0\t5\tACGTA\t*\n")
    with pytest.raises(api.GamlHipError):
        ctx.pacbio_ingest_sam(prs, walk, "r0/0_5\t0\tp\t5\n")
    lens_only = ctx.add_pacbio(api.single_cfg(mismatch_prob=MISMATCH), [100, 100])
    with pytest.raises(api.GamlHipError):
        ctx.pacbio_ingest_sam(lens_only, walk, ps.sam)


def test_golden_pins_through_the_c_abi():
    """The committed fixtures of tests/golden/pacbio_sam_pins.json, product side: SAM fields and band exact,
    log probabilities to 1e-9, filed records exact."""
    import json
    import os
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pacbio_sam_pins.json")))
    target = pins["target"]
    ctx = api.Context()
    for c in pins["lines"]:
        f, r0, lo, hi = api.debug_sam_band(c["sam"], len(target))
        assert f == c["fields"] and r0 == c["row0"] and len(lo) == c["rows"]
        assert int(lo.sum()) == c["lo_sum"] and int(hi.sum()) == c["hi_sum"]
        got, dlo, dhi = ctx.debug_sam_logprob(target, c["read"], c["sam"], pins["mismatch_prob"], with_band=True)
        assert np.array_equal(dlo, lo) and np.array_equal(dhi, hi)
        if c["logprob"] == "-inf":
            assert got == -np.inf
        else:
            want = float.fromhex(c["logprob"])
            assert abs(got - want) <= RTOL * abs(want)
    ing = pins["ingest"]
    genome = synth.make_genome(ing["genome_len"], ing["genome_seed"])
    g = synth.make_graph(genome, synth.cut_lengths(ing["genome_len"], ing["genome_seed"], long_rng=tuple(ing["long_rng"])))
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, ing["n_reads"], ing["read_len"], ing["sam_seed"])
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    c2 = api.Context()
    c2.set_graph(*g.packed())
    rs = c2.add_pacbio_reads(api.single_cfg(min_prob_per_base=-1.0, mismatch_prob=0.15), rb, ro, ps.names)
    assert c2.pacbio_ingest_sam(rs, walk, ps.sam) == ing["filed"]
    for key, want in ing["records"].items():
        got = c2.pacbio_records(rs, [int(x) for x in key.split()])
        assert np.array_equal(np.stack([got["position"], got["position_end"], got["read_id"]], 1), np.array(want["rec"]))
        np.testing.assert_allclose(got["logprob"], [float.fromhex(x) for x in want["logp"]], rtol=RTOL, atol=0)
    v, z, tl = c2.calc_prob([walk])
    assert abs(v - float.fromhex(ing["prob"])) <= RTOL * abs(v) and z.tolist() == ing["zeros"] and tl == ing["total_len"]
