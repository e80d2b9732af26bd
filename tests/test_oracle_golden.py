"""Oracle regression pins (tests/golden/oracle_pins.json, made by tests/golden/make_golden.py
from THIS oracle -- not reference output, see the header of oracle/gaml_oracle.hpp) plus
known-answer checks that follow directly from the reference's formulas."""
import json
import math
import os
import sys

import numpy as np

import oracle_py as op
from gaml_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import make_golden as mg  # noqa: E402

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_pins.json")))


def h(x):
    return float.fromhex(x)


def test_extend_hit_pins():
    for c in PINS["extend_hit"]:
        assert list(op.extend_hit(c["win_pos"], c["read_pos"], c["read"], c["win"])) == c["out"], c["name"]


def test_extend_hit_known_answers():
    # exact match: 0 errors, begin/end = the read's span (graph.cc:771-775, 808-811)
    win = "ACGTTGCAAGGCTTACGGATCCATGCAATTGGCCAAGGTTACGTACGATCGATTTGCACGTAGCTAGCTAGGATCCGATCGAACCGGTTAACGT"
    read = win[20:70]
    assert op.extend_hit(30, 10, read, win) == (0, 20, 69)
    # one substitution before and one after the seed
    r2 = list(read)
    r2[2] = "A" if r2[2] != "A" else "C"
    r2[40] = "A" if r2[40] != "A" else "C"
    assert op.extend_hit(30, 10, "".join(r2), win) == (2, 20, 69)
    # seed at the very start of the window: begin stays -1, errors = read_pos (graph.cc:797-798)
    assert op.extend_hit(0, 0, win[:50], win) == (0, -1, 49)
    assert op.extend_hit(0, 4, "TTTT" + win[:46], win)[0] == 4
    assert op.extend_hit(0, 6, "TTTTTT" + win[:44], win)[0] == -1


def test_insert_prob_pins_and_closed_form():
    for c in PINS["insert_prob"]:
        assert op.lib().orc_insert_prob(float(c["len"]), c["mean"], c["sd"]) == h(c["p"])
    # GetInsertProbability at the mean is 1 / (sd * sqrt(2 pi)) (graph.cc:1593-1598)
    assert op.lib().orc_insert_prob(300.0, 300.0, 30.0) == 1.0 / (math.sqrt(2 * math.pi) * 30.0)


def test_window_hash_pins():
    w = PINS["window_hashes"]
    assert [[str(a), b] for a, b in op.window_hashes(w["seq"], w["read_len"])] == w["pairs"]


def test_window_hash_emission_rule():
    # every emitted hash is the maximum scrambled 15-mer code of some read-length span and the
    # position is the LAST base of that 15-mer (graph.cc:1289-1323)
    seq = PINS["window_hashes"]["seq"]
    L = 100
    code = {"G": 0, "A": 1, "T": 2, "C": 3}

    def hv(i):  # 15-mer ending at i
        v = 0
        for ch in seq[i - 14:i + 1]:
            v = (v << 2) | code[ch]
        return v ^ 0x2204abcd
    pairs = op.window_hashes(seq, L)
    assert pairs, "no spans"
    spans = [max(hv(i) for i in range(e - L + 15, e + 1)) for e in range(L - 1, len(seq))]
    expect = [spans[0]] + [spans[k] for k in range(1, len(spans)) if spans[k] != spans[k - 1]]
    assert [p[0] for p in pairs] == expect
    for hsh, pos in pairs:
        assert hv(pos) == hsh


def _tiny():
    genome, g, pr = mg.tiny_case()
    gb, go = g.packed()
    return g, gb, go, synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)


def test_paired_pins_fresh_and_incremental():
    g, gb, go, (b1, o1), (b2, o2) = _tiny()
    for pen_name, pen in (("nopenalty", 0.0), ("penalty", 0.0002)):
        o = op.Oracle()
        o.set_graph(gb, go)
        rs = o.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(250.0, 25.0, penalty_constant=pen))
        pins = PINS[f"paired_{pen_name}"]
        for name, c in pins["cases"].items():
            v, z, tl = o.calc_prob(c["paths"], fresh=True)
            probs, bad = o.paired_probs(rs)
            assert v == h(c["prob"]), name
            assert z.tolist() == c["zeros"] and tl == c["total_len"] and bad == c["bad_bases"]
            assert float(np.sum(probs)) == h(c["probs_sum"]) and int((probs > 0).sum()) == c["probs_nonzero"]
        for step in pins["incremental"]:
            v, z, _ = o.calc_prob(pins["cases"][step["set"]]["paths"], fresh=False)
            assert v == h(step["prob"]) and z.tolist() == step["zeros"]
        if pen == 0.0:
            for key, recs in PINS["window_records_mate0"].items():
                assert o.window_records(rs, 0, [int(x) for x in key.split(",")]).tolist() == recs


def test_twin_walk_scores_like_the_forward_walk():
    # a walk and its InvertPath describe the same sequence on the other strand: same likelihood
    c = PINS["paired_nopenalty"]["cases"]
    assert c["twin"]["prob"] == c["one_walk"]["prob"] and c["twin"]["zeros"] == c["one_walk"]["zeros"]


def test_incremental_state_equals_fresh_state_on_a_warm_cache():
    # with the window cache warm and no drift-provoking erase/add cycles the
    # incremental ScoringState and a fresh one give the same value
    g, gb, go, (b1, o1), (b2, o2) = _tiny()
    o = op.Oracle()
    o.set_graph(gb, go)
    o.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(250.0, 25.0))
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    sets = [[walk], [walk[:k], walk[k:]], [walk]]
    for s in sets:  # warm the cache
        o.calc_prob(s, fresh=True)
    fresh = [o.calc_prob(s, fresh=True)[0] for s in sets]
    o.calc_prob([], fresh=True)
    inc = [o.calc_prob(s, fresh=False)[0] for s in sets]
    assert np.allclose(inc, fresh, rtol=1e-12, atol=0)


def test_all_reads_floored_known_answer():
    # a path where nothing aligns: every read takes the floor exp(c + k (L1 + L2)) and the score
    # is the mean of log(floor) (graph.cc:1504-1515)
    g, gb, go, (b1, o1), (b2, o2) = _tiny()
    o = op.Oracle()
    o.set_graph(gb, go)
    o.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(250.0, 25.0))
    short = [i for i in range(0, g.n_nodes, 2) if g.node_len(i) < 100][:1]
    v, z, tl = o.calc_prob([short], fresh=True)
    assert z.tolist() == [[1500, 1500]]
    assert v == math.log(math.exp(-10 - 0.7 * 200))
    assert tl == g.node_len(short[0])


def test_single_and_pacbio_pins():
    genome, g, pr = mg.tiny_case()
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    path_sets = {"one_walk": [walk], "two_walks": [walk[:k], walk[k:]], "gap": [walk[:k] + [-120] + walk[k + 2:]],
                 "singletons": [[x] for x in walk if g.node_len(x) > 500]}
    sr = synth.make_single_reads(genome, 1200, 100, 0.01, 9)
    sb, so = synth.pack_reads(sr)
    o = op.Oracle()
    o.set_graph(gb, go)
    ss = o.add_single(sb, so, 0.01, op.single_cfg())
    for name, c in PINS["single"].items():
        v, probs, o3 = o.single_detail(ss, path_sets[name])
        assert v == h(c["prob"]) and int(o3[0]) == c["zeros"] and int(o3[1]) == c["total_len"] and int(o3[2]) == c["bad_bases"]
        assert float(probs.sum()) == h(c["probs_sum"])
    for tag in ("pacbio", "pacbio_sparse"):
        c = PINS[tag]
        pb = synth.make_pacbio_records(g, walk, c["n_reads"], 2000, 0.15, 4)
        o = op.Oracle()
        o.set_graph(gb, go)
        ps = o.add_pacbio(pb.lens, 0.15, op.single_cfg(penalty_constant=0.0001, min_prob_per_base=-1.06))
        for wk, rec, lp in zip(pb.walks, pb.recs, pb.logps):
            o.pacbio_put(ps, wk, rec, lp)
        for sub in synth.all_subwalks_for_pacbio(g, walk, int(pb.lens.max())):
            o.pacbio_put(ps, sub, np.zeros((0, 3), np.int32), np.zeros(0))
        v, lp, o3 = o.pacbio_detail(ps, [walk])
        assert v == h(c["prob"]) and int(o3[0]) == c["zeros"] and int(o3[2]) == c["bad_bases"]
        assert o.pacbio_misses(ps) == 0


def test_pacbio_single_read_known_answer():
    # one read, one alignment with log-probability lp: score = max(lp, floor) - log(2 T)
    # (graph.cc:3062-3088)
    genome, g, pr = mg.tiny_case()
    gb, go = g.packed()
    walk = synth.genome_walk(g)[:3]
    o = op.Oracle()
    o.set_graph(gb, go)
    ps = o.add_pacbio(np.array([1000], np.int32), 0.15, op.single_cfg(min_prob_per_base=-3.0))
    for sub in synth.all_subwalks_for_pacbio(g, walk, 1000):
        o.pacbio_put(ps, sub, np.zeros((0, 3), np.int32), np.zeros(0))
    o.pacbio_put(ps, walk[:1], np.array([[10, 1010, 0]], np.int32), np.array([-1234.5]))
    v, lp, o3 = o.pacbio_detail(ps, [walk])
    T = sum(g.node_len(x) for x in walk)
    assert lp[0] == -1234.5
    assert v == -1234.5 - math.log(2 * T)
    # the same alignment twice: logdouble sum = lp + log(2)
    o.pacbio_put(ps, walk[:1], np.array([[300, 1300, 0]], np.int32), np.array([-1234.5]))
    v2, lp2, _ = o.pacbio_detail(ps, [walk])
    assert lp2[0] == -1234.5 + math.log1p(math.exp(0.0))


def test_config_quirks(tmp_path):
    # gaml.cc:748-872: paired sets read min_prob_pre_base (sic), step = insert_mean - penalty_step,
    # sections iterate in hash order, a set without type is ignored
    cfg = tmp_path / "a.cfg"
    cfg.write_text("graph=g\nt0=0.02\n\n[rs1]\ntype=paired\nfilename1=a\nfilename2=b\ninsert_mean=180\ninsert_std=20\n"
                   "penalty_step=30\npenalty_constant=0.00007\nmin_prob_per_base=0\nmin_prob_start=80\n\n"
                   "[rs2]\ntype=paired\nfilename1=c\nfilename2=d\ninsert_mean=3700\ninsert_std=350\nmin_prob_pre_base=-0.5\n"
                   "weight=0.5\n\n[bad]\nfilename=x\n")
    import ctypes as C
    vals = np.zeros(8)
    assert op.lib().orc_config_paired_values(str(cfg).encode(), b"rs1", vals) == 0
    assert vals.tolist() == [0.00007, 150.0, 180.0, 20.0, -0.7, 80.0, 1.0, 0.01]  # documented key silently ignored
    assert op.lib().orc_config_paired_values(str(cfg).encode(), b"rs2", vals) == 0
    assert vals.tolist() == [0.0, 3650.0, 3700.0, 350.0, -0.5, -10.0, 0.5, 0.01]
    buf = C.create_string_buffer(64)
    op.lib().orc_config_order(str(cfg).encode(), buf, 64)
    assert sorted(buf.value.decode().strip(",").split(",")) == ["rs1", "rs2"]


def test_alignment_dp_against_explicit_path_enumeration():
    """AligmentProbability (reference graph.cc:2175-2297) sums, over every monotone path through the banded
    cell set that starts in column 0 and ends in column |read|, the product of MatchProbability along the
    path. Enumerate those paths one by one (exponential, so tiny inputs) and compare with the oracle's
    dynamic programme -- an answer that does not share the DP's bookkeeping."""
    import math
    from oracle_py import sam_band, sam_alignment_logprob
    mism = 0.15
    match = 1.0 - 4 * mism

    def pm(a, b):
        if a == "\n" or b == "\n":
            return 0.0
        return match if a == b else mism

    half = "ACGTTGCAAGCT"
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    target = half + "\n" + "".join(comp[ch] for ch in reversed(half))
    cases = [
        ("q/1\t0\tp\t3\t1\t4M\t*\t0\t4\tTTGC\t*", "TTGC"),
        ("q/1\t0\tp\t2\t1\t2M1I2M\t*\t0\t4\tGTATG\t*", "GTATG"),
        ("q/1\t0\tp\t4\t1\t2M1D2M\t*\t0\t5\tTGAA\t*", "TGAA"),
        ("q/1\t16\tp\t2\t1\t3M\t*\t0\t3\tAGC\t*", "GCT"),           # mirrored into the reverse-complement half
        ("q/1\t0\tp\t10\t1\t4M\t*\t0\t4\tGCTA\t*", "GCTA"),          # runs into the separator: those rows contribute nothing
        ("q/1\t0\tp\t5\t1\t2M\t*\t0\t2\tGC\t*\tXS:i:2\tXE:i:4\tXQ:i:4", "AGCT"),  # soft clips on both ends
    ]
    for line, read in cases:
        f, r0, lo, hi = sam_band(line, len(target))
        n = len(read)
        cells = {(r0 + i, c) for i in range(len(lo)) for c in range(int(lo[i]), int(hi[i]) + 1)}

        def usable(r, c):  # a cell the DP computes (graph.cc:2246-2255)
            gi = r + f["posstart"] - 1
            return (r, c) in cells and 1 <= c <= n and 0 <= gi < len(target)

        total = 0.0

        def walk(r, c, w):
            nonlocal total
            if c == n:
                total += w  # every computed cell of the last column is summed (graph.cc:2279-2281)
            for dr, dc in ((1, 1), (1, 0), (0, 1)):
                rr, cc = r + dr, c + dc
                if not usable(rr, cc):
                    continue
                g = target[rr + f["posstart"] - 1]
                step = pm(g, read[cc - 1]) if (dr, dc) == (1, 1) else pm(g, "-") if (dr, dc) == (1, 0) else pm("-", read[cc - 1])
                if step > 0.0:
                    walk(rr, cc, w * step)

        for (r, c) in sorted(cells):
            if c == 0:  # free start in column 0: value 1, never recomputed
                for dr, dc in ((1, 1), (0, 1)):
                    rr, cc = r + dr, c + dc
                    if usable(rr, cc):
                        g = target[rr + f["posstart"] - 1]
                        step = pm(g, read[cc - 1]) if dr else pm("-", read[cc - 1])
                        if step > 0.0:
                            walk(rr, cc, step)
        got = sam_alignment_logprob(line, target, read, mism)
        assert total > 0.0 and math.isfinite(got), line
        assert abs(got - math.log(total)) <= 1e-10 * abs(got), (line, got, math.log(total))


def test_pacbio_sam_pins():
    """tests/golden/pacbio_sam_pins.json (regression pins made by tests/golden/make_golden.py sam): parsed
    fields, band and log probability per SAM line, and what one ingest files."""
    import json
    import os
    import oracle_py as op2
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pacbio_sam_pins.json")))
    target = pins["target"]
    for c in pins["lines"]:
        f, r0, lo, hi = op2.sam_band(c["sam"], len(target))
        assert f == c["fields"] and r0 == c["row0"] and len(lo) == c["rows"]
        assert [int(x) for x in lo[:8]] == c["lo_head"] and [int(x) for x in hi[:8]] == c["hi_head"]
        assert int(lo.sum()) == c["lo_sum"] and int(hi.sum()) == c["hi_sum"]
        lp = op2.sam_alignment_logprob(c["sam"], target, c["read"], pins["mismatch_prob"])
        if c["logprob"] == "-inf":
            assert lp == -np.inf
        else:
            want = float.fromhex(c["logprob"])
            assert abs(lp - want) <= 1e-13 * abs(want)
    ing = pins["ingest"]
    genome = synth.make_genome(ing["genome_len"], ing["genome_seed"])
    g = synth.make_graph(genome, synth.cut_lengths(ing["genome_len"], ing["genome_seed"], long_rng=tuple(ing["long_rng"])))
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, ing["n_reads"], ing["read_len"], ing["sam_seed"])
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    o = op2.Oracle()
    o.set_graph(*g.packed())
    rs = o.add_pacbio_reads(rb, ro, ps.names, 0.15, op2.single_cfg(min_prob_per_base=-1.0))
    assert o.pacbio_ingest_sam(rs, walk, ps.sam) == ing["filed"]
    assert len(o.pacbio_keys(rs)) == ing["n_keys"]
    for key, want in ing["records"].items():
        rec, lp = o.pacbio_records(rs, [int(x) for x in key.split()])
        assert rec.tolist() == want["rec"]
        np.testing.assert_allclose(lp, [float.fromhex(x) for x in want["logp"]], rtol=1e-13, atol=0)
    v, z, tl = o.calc_prob([walk])
    assert abs(v - float.fromhex(ing["prob"])) <= 1e-13 * abs(v) and z.tolist() == ing["zeros"] and tl == ing["total_len"]


def _slow_vs_incremental(err, long_rng, seed=5, G=30_000, n=4000):
    import oracle_py as op
    from gaml_amd import synth
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=long_rng, short_rng=(40, 120)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, err, seed)
    o = op.Oracle()
    o.set_graph(*g.packed())
    rs = o.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(250.0, 25.0))
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    out = []
    for paths in ([walk], [walk[:k], walk[k:]], [walk[:k]], [[x ^ 1 for x in reversed(walk)]]):
        fast = o.calc_prob(paths, fresh=True)
        fast_probs = o.paired_probs(rs)[0].copy()
        slow = o.paired_slow(rs, paths)
        out.append((fast, fast_probs, slow))
    return out


def test_slow_paired_scorer_agrees_with_the_incremental_one_where_the_definitions_coincide(built):
    """The reference carries a second paired scorer, the slow non-incremental CalcScoreForPaths (graph.cc:1991-2127),
    whose comparison with CalcScoreForPathsNew is commented out at prob_calculator.h:80-95. Restated in the oracle as a
    differently structured opinion on the paired value (single-end style position assembly, all-against-all pairing
    over absolute coordinates). The two definitions coincide when every node is at most kMinSubpathLength long (the slow
    scorer knows no whole-node windows), there are no gaps and no penalty, and a read has ONE distinct alignment per mate
    (a second one a base or two away -- short tandem repeats allow it even without sequencing errors -- is kept, dropped
    or overwritten by different rules: graph.cc:577, 583-592 vs 633-644). On those reads: bit for bit; the others are few."""
    for fast, fast_probs, slow in _slow_vs_incremental(err=0.0, long_rng=(180, 300)):
        one = slow[3] == 0
        assert one.sum() >= 0.99 * len(one)
        assert np.array_equal(fast_probs[one], slow[1][one])
        assert fast[2] == int(slow[2][1]) and abs(fast[1][0][0] - int(slow[2][0])) <= int((~one).sum())
        assert abs(fast[0] - slow[0]) <= 1e-3 * abs(fast[0])
    # with sequencing errors the lossy aligner finds a few reads in a later window only, at a position the incremental
    # scorer's filter (max_pos - 5, graph.cc:577) has already passed: dropped there, kept by AddPositions
    for fast, fast_probs, slow in _slow_vs_incremental(err=0.01, long_rng=(180, 300)):
        one = slow[3] == 0
        differ = one & (fast_probs != slow[1])
        assert one.sum() >= 0.98 * len(one) and differ.sum() <= 4
        assert np.all(fast_probs[differ] == 0.0) and np.all(slow[1][differ] > 0.0)


def test_slow_paired_scorer_where_the_definitions_part(built):
    """With sequencing errors a few reads get a second, shifted alignment a base or two away; the incremental scorer's
    position filter (graph.cc:577) and overwrite rule (:583-592) then keep another subset than AddPositions (:633-644):
    all but a handful of reads still agree bit for bit, the likelihood to 1e-3. With nodes longer than
    kMinSubpathLength the slow scorer misses every read in a node's interior (junction windows only, the first node cut to
    its last 300 bases, graph.cc:846-848) -- which is presumably why the reference stopped calling it."""
    for fast, fast_probs, slow in _slow_vs_incremental(err=0.01, long_rng=(180, 300)):
        assert abs(fast[0] - slow[0]) <= 1e-3 * abs(fast[0])
    for fast, fast_probs, slow in _slow_vs_incremental(err=0.0, long_rng=(900, 2500)):
        assert int(slow[2][0]) > fast[1][0][0] + len(fast_probs) // 5  # far more floored reads
        assert np.all(slow[1] <= fast_probs + 1e-300)        # it only ever loses alignments here
