"""The record tables are built and maintained ON THE DEVICE (table_build.hip.h, delta_dev.hip.h): the read-major join the
reference does per call through hash maps (graph.cc:535-598) over its window cache (graph.cc:911-922). The device build
is compared with the host restatement entry by entry; the delta lists against tables rebuilt from scratch and the oracle."""
import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _ctx(g, pr, insert=(240.0, 24.0), knobs=None):
    from gaml_amd import api
    c = api.Context(device=0)
    for k, v in (knobs or {}).items():
        c.debug_set_knob(k, v)
    c.set_graph(*g.packed())
    rs = c.add_paired(api.paired_cfg(*insert), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    return c, rs


@pytest.mark.parametrize("n,G,repeats", [(3000, 60_000, 0), (40_000, 150_000, 3), (90_000, 600_000, 0)])
def test_device_table_build_equals_the_host_restatement(n, G, repeats):
    seed = 7 + n
    genome = synth.make_genome(G, seed)
    if repeats:
        genome = synth.plant_repeats(genome, repeats, 800, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    walk = synth.genome_walk(g)
    for fold in (0, 1):
        c, rs = _ctx(g, pr, knobs={16: fold})
        c.calc_prob([walk])
        r = c.debug_tables_check(rs)
        assert r["mismatches"] == 0 and r["pairs"] == n and r["compared"] > 4 * n, r
        # ... and again with the twin walk's windows and single nodes active as well (more records per pair)
        c.calc_prob([[x ^ 1 for x in reversed(walk)]])
        c.calc_prob([[x] for x in walk])
        r2 = c.debug_tables_check(rs)
        assert r2["mismatches"] == 0 and r2["compared"] > r["compared"] // 2, r2
        c.close()


def test_two_builds_of_one_state_are_bit_equal():
    """Equal inputs give equal tables and equal values (the annealing loop compares likelihoods with strict >)."""
    G, n, seed = 200_000, 30_000, 3
    genome = synth.plant_repeats(synth.make_genome(G, seed), 2, 900, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(600, 4000), short_rng=(25, 330)))
    pr = synth.make_paired_reads(genome, n, 100, 240.0, 24.0, 0.01, seed)
    start, seq = synth.sa_sequence(g, 150, seed=4, threshold=400)
    vals = []
    for rep in range(2):
        c, rs = _ctx(g, pr, knobs={14: 16, 18: 64})
        vals.append([c.calc_prob(ps)[0] for ps in [start] + seq])
        assert c.debug_table_stats(rs)["worker_rebuilds"] >= 1
        c.close()
    assert vals[0] == vals[1]
