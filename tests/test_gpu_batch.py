"""gaml_hip_calc_prob_batch: several path sets in one call give the values of as many
single calls in the same order -- including on a cold cache, where set i's newly aligned windows are
visible to set i+1 exactly as in a sequence of CalcProb calls -- and the oracle's values."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _setup(penalty=0.0):
    from gaml_amd import api
    genome = synth.make_genome(120_000, 83)
    g = synth.make_graph(genome, synth.cut_lengths(120_000, 83, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, 9000, 150, 300.0, 30.0, 0.01, 83)
    sr = synth.make_single_reads(genome, 2000, 100, 0.01, 83)
    walk = synth.genome_walk(g)
    rng = np.random.default_rng(5)
    sets = []
    for k in range(9):  # speculative edits of one assembly: cuts, a dropped node, a reversed piece
        cut = int(rng.integers(2, len(walk) - 2))
        if k % 3 == 0:
            sets.append([walk[:cut], walk[cut:]])
        elif k % 3 == 1:
            sets.append([walk[:cut] + walk[cut + 1:]])
        else:
            sets.append([walk[:cut], [x ^ 1 for x in reversed(walk[cut:])]])
    sets.append([])  # an empty assembly is a valid argument too

    def make():
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(300.0, 30.0, penalty_constant=penalty), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        c.add_single(api.single_cfg(), *synth.pack_reads(sr))
        return c
    return g, pr, sr, sets, make


@pytest.mark.parametrize("penalty", [0.0, 0.0005])
def test_batch_equals_single_calls_cold_and_warm(penalty):
    g, pr, sr, sets, make = _setup(penalty)
    one, many = make(), make()
    want = [one.calc_prob(s) for s in sets]          # cold: windows get aligned along the way
    got = many.calc_prob_batch(sets)
    assert len(got) == len(sets)
    for w, b in zip(want, got):
        assert b[2] == w[2] and b[1].tolist() == w[1].tolist()
        assert abs(b[0] - w[0]) <= 1e-13 * abs(w[0])
    for a, b in zip(many.calc_prob_batch(sets), [one.calc_prob(s) for s in sets]):  # warm
        assert a[2] == b[2] and a[1].tolist() == b[1].tolist() and abs(a[0] - b[0]) <= 1e-13 * abs(b[0])
    assert many.calc_prob_batch([]) == []


def test_batch_against_the_oracle():
    import oracle_py as op
    g, pr, sr, sets, make = _setup()
    o = op.Oracle()
    o.set_graph(*g.packed())
    o.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(300.0, 30.0))
    o.add_single(*synth.pack_reads(sr), 0.01, op.single_cfg())
    got = make().calc_prob_batch(sets[:4])
    for s, b in zip(sets[:4], got):
        w = o.calc_prob(s)
        assert b[1].tolist() == w[1].tolist() and b[2] == w[2]
        assert abs(b[0] - w[0]) <= 1e-9 * abs(w[0])  # LL tolerance of the north star: 1e-6 relative


def _pack(reads):
    offs = np.zeros(len(reads) + 1, np.int64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.concatenate([np.asarray(r, np.uint8) for r in reads]), offs


def _paired_only(mixed_lengths=False, n=9000, seed=83):
    from gaml_amd import api
    genome = synth.plant_repeats(synth.make_genome(120_000, seed), 2, 700, seed)
    g = synth.make_graph(genome, synth.cut_lengths(120_000, seed, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, 0.01, seed)
    m1, m2 = list(pr.mate1), list(pr.mate2)
    if mixed_lengths:  # trimmed reads: several (L1, L2) combinations -> the length-code tables of the compact class
        rng = np.random.default_rng(3)
        for i in range(0, n, 3):
            m1[i] = m1[i][: int(rng.integers(110, 150))]
        for i in range(1, n, 5):
            m2[i] = m2[i][: int(rng.integers(120, 150))]
    walk = synth.genome_walk(g)
    rng = np.random.default_rng(9)
    sets = []
    for k in range(19):  # more than one chunk of 8
        cut = int(rng.integers(2, len(walk) - 2))
        kind = k % 5
        if kind == 0:
            sets.append([walk[:cut], walk[cut:]])
        elif kind == 1:
            sets.append([walk[:cut] + walk[cut + 1:]])
        elif kind == 2:
            sets.append([walk[:cut], [x ^ 1 for x in reversed(walk[cut:])]])
        elif kind == 3:  # a duplicated stretch: its windows occur several times (the kernels' GEN instantiation)
            sets.append([walk[:cut] + walk[max(0, cut - 3):cut] + walk[cut:]])
        else:
            sets.append([walk[:cut] + [-int(rng.integers(10, 300))] + walk[cut + 2:], walk[2:9]])
    sets.insert(7, [])

    def make():
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(300.0, 30.0), *_pack(m1), *_pack(m2))
        return c
    return g, (m1, m2), sets, make


@pytest.mark.parametrize("mixed_lengths", [False, True])
def test_one_pass_batch_equals_single_calls(mixed_lengths):
    """Contexts of paired sets only take the one-pass kernel (paired_score_multi_kernel): every set's tables resolved
    against records that are loaded once. Cold (windows aligned along the way, delta lists), warm, after a table
    rebuild; chunks of 8; sets with repeated windows; an empty set. Values, floored counts and the per-read
    probabilities left behind are those of the single calls."""
    g, reads, sets, make = _paired_only(mixed_lengths)
    one, many = make(), make()
    want = [one.calc_prob(s) for s in sets]
    got = many.calc_prob_batch(sets)
    assert len(got) == len(sets)
    for w, b in zip(want, got):
        assert b[2] == w[2] and b[1].tolist() == w[1].tolist()
        assert abs(b[0] - w[0]) <= 1e-13 * abs(w[0]), (b[0], w[0])
    assert np.array_equal(many.read_probs(0), one.read_probs(0))  # those of the last set
    one.compact_tables(); many.compact_tables()
    for rnd in range(2):
        want = [one.calc_prob(s) for s in sets]
        got = many.calc_prob_batch(sets)
        for w, b in zip(want, got):
            assert b[2] == w[2] and b[1].tolist() == w[1].tolist() and abs(b[0] - w[0]) <= 1e-13 * abs(w[0])
    # same device state on both sides now: bit for bit
    assert [b[0] for b in many.calc_prob_batch(sets[:8])] == [many.calc_prob(s)[0] for s in sets[:8]]
    many.debug_set_knob(11, 1)  # the sequential path (one launch per set)
    assert [b[0] for b in many.calc_prob_batch(sets[:8])] == [many.calc_prob(s)[0] for s in sets[:8]]


@pytest.mark.parametrize("two_sets", [False, True])
def test_candidate_batches_build_their_tables_on_the_device(two_sets):
    """Candidates of one assembly (each a single edit away from the current one): the per-set occurrence tables are
    built on the device from the resident copy + a few patched entries, and pairs whose windows no later set changed
    are finished from their first set's result. Every route must give the same bits: the device-built tables, whole
    tables per set (knob 11 = 2), no capture (knob 11 = 3), one launch per set (knob 11 = 1), and single calls."""
    from gaml_amd import api
    G, seed = 400_000, 31
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 2500), short_rng=(25, 250)))
    pr = synth.make_paired_reads(genome, 40_000, 100, 250.0, 25.0, 0.01, seed)
    args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    pr2 = synth.make_paired_reads(genome, 15_000, 100, 400.0, 40.0, 0.01, seed + 1)
    args2 = (*synth.pack_reads(pr2.mate1), *synth.pack_reads(pr2.mate2))
    start, seq = synth.sa_sequence(g, 60, seed=seed, threshold=400)
    base = seq[-1]
    rng = np.random.default_rng(seed)
    batches = []
    for _ in range(12):
        cands = [synth.sa_move(rng, base, g) for _ in range(8)]
        batches.append(cands)
        if rng.random() < 0.6:
            base = cands[int(rng.integers(0, 8))]
    ctxs = []
    for knob in (0, 2, 3, 1, None):  # None: single calls
        c = api.Context(device=0)
        c.set_graph(*g.packed())
        c.add_paired(api.paired_cfg(250.0, 25.0), *args)
        if two_sets:  # a second library over the same assembly: its own tables, patches and launches
            c.add_paired(api.paired_cfg(400.0, 40.0, weight=0.5), *args2)
        if knob:
            c.debug_set_knob(11, knob)
        c.calc_prob(seq[-1])
        ctxs.append((knob, c))
    for rnd in range(2):  # cold (windows aligned along the way), then warm
        for cands in batches:
            vals = []
            for knob, c in ctxs:
                if knob is None:
                    vals.append([(c.calc_prob(s)[0], c.calc_prob(s)[1].tolist()) for s in cands])
                else:
                    vals.append([(b[0], b[1].tolist()) for b in c.calc_prob_batch(cands)])
            for v in vals[1:]:
                if rnd == 0:  # the contexts' delta lists fill in different orders while windows get aligned: last bits of the sum
                    assert all(a[1] == b[1] and abs(a[0] - b[0]) <= 1e-13 * abs(b[0]) for a, b in zip(v, vals[0]))
                else:
                    assert v == vals[0]
        if rnd == 0:  # the same device state everywhere: everything folded into the record tables
            for knob, c in ctxs:
                c.compact_tables()
                c.calc_prob(base)
    st = ctxs[0][1].debug_table_stats(0)
    assert st["batches_patched"] >= 20 and st["batches_patched"] > 5 * st["batches_full"], st
    assert ctxs[1][1].debug_table_stats(0)["batches_patched"] == 0
    # a blocking call after a batch finds the resident tables in step with the planner
    last = batches[-1][-1]
    assert ctxs[0][1].calc_prob(last)[0] == ctxs[4][1].calc_prob(last)[0]


def test_one_pass_batch_against_the_oracle():
    import oracle_py as op
    g, (m1, m2), sets, make = _paired_only(True, n=5000, seed=21)
    o = op.Oracle()
    o.set_graph(*g.packed())
    o.add_paired(*_pack(m1), *_pack(m2), 0.01, op.paired_cfg(300.0, 30.0))
    pick = [sets[0], sets[3], sets[4], sets[2], sets[7]]
    got = make().calc_prob_batch(pick)
    for s, b in zip(pick, got):
        w = o.calc_prob(s, fresh=True)
        assert b[1].tolist() == w[1].tolist() and b[2] == w[2]
        assert (b[0] == w[0]) or abs(b[0] - w[0]) <= 1e-9 * abs(w[0])


def test_batch_refused_on_sharded_context():
    from gaml_amd import api
    c = api.Context(device=0, rank=0, world=2)
    with pytest.raises(api.GamlHipError) as e:
        c.calc_prob_batch([[[0]]])
    assert e.value.code == api.ESTATE
