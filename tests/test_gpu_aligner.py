"""GPU window aligner (cold path) against the library's host aligner and the oracle: identical
Aligment records for every window (reference AlignSubpathInternal graph.cc:839-899)."""
import time

import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _ctx(api, gb, go, pr, host_aligner, mean=300.0, sd=30.0):
    ctx = api.Context(device=0)
    ctx.debug_set_knob(5, 1 if host_aligner else 0)
    ctx.set_graph(gb, go)
    ctx.add_paired(api.paired_cfg(mean, sd), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    return ctx


def _all_windows(ctx, mate):
    return [tuple(ctx.debug_window_walk(0, mate, w)) for w in range(ctx.window_count(0, mate))]


@pytest.mark.parametrize("L,err,repeats", [(150, 0.01, 0), (100, 0.03, 4), (254, 0.005, 2)])
def test_gpu_records_equal_host_aligner_and_oracle(L, err, repeats):
    from gaml_amd import api
    import oracle_py as op
    G, n, seed = 90_000, 5000, 81 + L
    genome = synth.make_genome(G, seed)
    if repeats:
        genome = synth.plant_repeats(genome, repeats, 900, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 5000), short_rng=(20, 340)))
    pr = synth.make_paired_reads(genome, n, L, 2.2 * L, 0.2 * L, err, seed)
    gb, go = g.packed()
    gpu = _ctx(api, gb, go, pr, False, 2.2 * L, 0.2 * L)
    cpu = _ctx(api, gb, go, pr, True, 2.2 * L, 0.2 * L)
    walk = synth.genome_walk(g)
    k = len(walk) // 3
    sets = [[walk], [walk[:k], [x ^ 1 for x in reversed(walk[k:2 * k])], walk[2 * k:]], [[x] for x in walk], [walk[4:9] + [-33] + walk[11:15]]]
    for paths in sets:
        a = gpu.calc_prob(paths)
        b = cpu.calc_prob(paths)
        assert a[0] == b[0] and a[1].tolist() == b[1].tolist()
    st = gpu.aligner_stats()
    assert st["windows"] > 0 and st["candidates"] > 0 and cpu.aligner_stats()["windows"] == 0
    for mate in (0, 1):
        wa, wb = _all_windows(gpu, mate), _all_windows(cpu, mate)
        assert wa == wb and len(wa) == st["windows"] // 2 or wa == wb
        for key in wa:
            ra, rb = gpu.window_records(0, mate, list(key)), cpu.window_records(0, mate, list(key))
            assert ra.shape == rb.shape and (ra == rb).all(), key
    # and the oracle's deque-based 0-1 BFS on a sample of windows
    orc = op.Oracle()
    orc.set_graph(gb, go)
    orc.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(2.2 * L, 0.2 * L))
    for key in _all_windows(gpu, 0)[::5]:
        orc.align_window(0, 0, list(key))
        assert (orc.window_records(0, 0, list(key)) == gpu.window_records(0, 0, list(key))).all(), key


def test_gpu_aligner_speed_and_parity_at_cfg2():
    from gaml_amd import api
    wl = synth.WORKLOADS["cfg2"]
    genome = synth.make_genome(wl.genome_len, wl.seed)
    g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    out = {}
    for name, host in (("gpu", False), ("cpu", True)):
        ctx = _ctx(api, gb, go, pr, host)
        t0 = time.time()
        out[name] = (ctx.calc_prob([walk]), time.time() - t0, ctx)
    assert out["gpu"][0][0] == out["cpu"][0][0] and out["gpu"][0][1].tolist() == out["cpu"][0][1].tolist()
    # > 100,000 seed candidates per mate: the hits were ordered on the device (two radix sorts); same records
    assert out["gpu"][2].aligner_stats()["candidates"] > 300_000
    for w in ([walk[0]], [walk[4]], walk[0:3], walk[6:9], [walk[-1]]):
        for mate in (0, 1):
            a, b = out["gpu"][2].window_records(0, mate, w), out["cpu"][2].window_records(0, mate, w)
            assert (a is None) == (b is None)
            if a is not None:
                assert a.tobytes() == b.tobytes()
    print(f"cold CalcProb cfg2: gpu aligner {out['gpu'][1]:.3f} s ({out['gpu'][2].aligner_stats()}), host aligner {out['cpu'][1]:.3f} s")
    assert out["gpu"][1] < out["cpu"][1]


def test_small_batch_pipeline_equals_the_general_route():
    """Small batches of new windows (what an annealing move brings) take a one-wait pipeline on the library's stream:
    both mates in one (aln_pair_small), or one per mate side by side (knob 5 = 4); knob 5 = 3 forces the general route.
    Same records, same values, along an annealing-style walk."""
    from gaml_amd import api
    G, n, seed = 120_000, 6000, 77
    genome = synth.plant_repeats(synth.make_genome(G, seed), 2, 700, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, 0.01, seed)
    gb, go = g.packed()
    fast, per_mate, general = _ctx(api, gb, go, pr, False), _ctx(api, gb, go, pr, False), _ctx(api, gb, go, pr, False)
    per_mate.debug_set_knob(5, 4)
    general.debug_set_knob(5, 3)
    start, seq = synth.sa_sequence(g, 120, seed=3, threshold=400)
    for ps in [start] + seq:
        a, b, c = fast.calc_prob(ps), general.calc_prob(ps), per_mate.calc_prob(ps)
        assert a[0] == b[0] == c[0] and a[1].tolist() == b[1].tolist() == c[1].tolist()
    assert fast.aligner_stats()["windows"] == general.aligner_stats()["windows"] == per_mate.aligner_stats()["windows"] > 0
    for mate in (0, 1):
        wa = _all_windows(fast, mate)
        assert wa == _all_windows(general, mate) == _all_windows(per_mate, mate)
        for key in wa[::3]:
            rec = fast.window_records(0, mate, list(key)).tobytes()
            assert rec == general.window_records(0, mate, list(key)).tobytes() == per_mate.window_records(0, mate, list(key)).tobytes()
