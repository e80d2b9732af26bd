"""Several device shards behind ONE context (gaml_hip_create_multi) and the in-library RCCL communicator.

The GPU test box has one GPU: shards that share it exchange through host memory (RCCL refuses two ranks on one
device); the RCCL path itself is exercised with one rank (single-shard multi context, and gaml_hip_comm_init_rank with
world = 1). Everything is compared with the plain single-device context on the same inputs, which the other GPU tests
pin to the oracle."""
import os
import re
import subprocess

import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(G=90_000, n=6001, seed=83):
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, 0.01, seed)
    return genome, g, (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))


def _rel(a, b):
    return abs(a - b) / max(1e-300, abs(b))


def test_shards_in_one_process_equal_the_whole():
    from gaml_amd import api
    genome, g, reads = _case()
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    sets = [[walk], [walk[:9], walk[9:]], [walk[:9], walk[11:30], [x ^ 1 for x in reversed(walk[30:])]], [walk[:20] + [-77] + walk[22:]]]
    whole = api.Context(device=0)
    whole.set_graph(gb, go)
    rs = whole.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    multi = api.Context(devices=[0, 0, 0])
    assert multi.num_shards() == 3 and multi.exchange() == "host"
    multi.set_graph(gb, go)
    assert multi.add_paired(api.paired_cfg(300.0, 30.0), *reads) == rs
    for rnd in range(2):  # cold (windows aligned per shard, maxima merged in the library), then warm
        for ps in sets:
            want, wz, wtl = whole.calc_prob(ps)
            got, gz, gtl = multi.calc_prob(ps)
            assert gtl == wtl and np.array_equal(gz, wz)
            assert _rel(got, want) <= 1e-12, (rnd, got, want)
            assert np.array_equal(multi.read_probs(rs), whole.read_probs(rs))  # per-read values: bit for bit
    # batch entry point == the calls one by one
    bp = api.BatchPaths(sets)
    for (p, z, tl), ps in zip(multi.calc_prob_batch(bp), sets):
        want, wz, wtl = whole.calc_prob(ps)
        assert tl == wtl and np.array_equal(z, wz) and _rel(p, want) <= 1e-12
    # partials come back reduced over the shards
    part, tl = multi.calc_partials(sets[1])
    wpart, _ = whole.calc_partials(sets[1])
    assert part[0][1] == wpart[0][1] and part[0][3] == wpart[0][3] == 6001 and _rel(part[0][0], wpart[0][0]) <= 1e-12
    multi.close()


def test_shards_with_coverage_penalties_and_long_reads():
    """penalty_constant > 0: bad_bases is a function of the union of all shards' coverage marks (paired,
    graph.cc:1893-1919) / interval events (PacBio, graph.cc:3226-3250) -- merged inside the library."""
    from gaml_amd import api
    genome, g, reads = _case(G=70_000, n=1500, seed=97)  # thin coverage: uncovered stretches exist
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    rng = np.random.default_rng(5)
    lens = rng.integers(900, 2500, 300).astype(np.int32)
    sr = synth.make_single_reads(genome, 700, 100, 0.01, 3)
    cached = []  # synthetic long-read alignments on two sub-walks (the records BLASR + the DP would have cached)
    for sub in (walk[0:3], walk[4:7]):
        span = sum(g.node_len(x) for x in sub)
        ids = rng.permutation(300)[:120]
        rec3 = [(int(p), int(min(span, p + lens[i])), int(i)) for i, p in zip(ids, rng.integers(0, max(1, span - 900), 120))]
        cached.append((sub, rec3, -rng.uniform(200, 900, 120)))
    ctxs = []
    for kw in (dict(device=0), dict(devices=[0, 0]), dict(device=0, comm=True)):
        comm = kw.pop("comm", False)
        c = api.Context(**kw)
        if comm:  # one process per GPU, world = 1: the RCCL form of the exchanges (all-gather of maps / interval lists)
            c.comm_init_rank(api.comm_unique_id(), 0, 1)
        c.set_graph(gb, go)
        c.add_paired(api.paired_cfg(300.0, 30.0, penalty_constant=0.001, penalty_step=20.0), *reads)
        c.add_single(api.single_cfg(weight=0.5), *synth.pack_reads(sr))
        pb = c.add_pacbio(api.single_cfg(penalty_constant=0.002, penalty_step=60.0, mismatch_prob=0.15, weight=0.5), lens)
        for sub, rec3, logp in cached:
            c.put_pacbio_records(pb, sub, rec3, logp)
        ctxs.append(c)
    whole, multi, solo = ctxs
    for ps in ([walk], [walk[:8], walk[8:]]):
        want, wz, wtl = whole.calc_prob(ps)
        assert whole.bad_bases(0) > 0 and whole.bad_bases(2) > 0
        for c in (multi, solo):
            got, gz, gtl = c.calc_prob(ps)
            assert gtl == wtl and np.array_equal(gz, wz)
            assert c.bad_bases(0) == whole.bad_bases(0)
            assert c.bad_bases(2) == whole.bad_bases(2)
            assert _rel(got, want) <= 1e-12, (got, want)
    multi.close()


def test_one_shard_over_rccl_and_comm_init_rank():
    """The RCCL exchange with ONE rank: the same code path N ranks take (finisher kernel -> ncclAllReduce on the
    library's stream -> fetch), checked against the blocking single-device path."""
    from gaml_amd import api
    genome, g, reads = _case(n=4000, seed=29)
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    sets = [[walk], [walk[:15], walk[15:]]]
    plain = api.Context(device=0)
    plain.set_graph(gb, go)
    plain.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    # (a) a multi context with one device: ncclCommInitAll over [0]
    multi = api.Context(devices=[0])
    assert multi.exchange() == "rccl", multi.last_error()
    multi.set_graph(gb, go)
    multi.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    # (b) one process per GPU: gaml_hip_comm_unique_id / gaml_hip_comm_init_rank, world = 1
    solo = api.Context(device=0)
    solo.comm_init_rank(api.comm_unique_id(), 0, 1)
    solo.set_graph(gb, go)
    solo.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    for ps in sets + sets:
        want, wz, wtl = plain.calc_prob(ps)
        for c in (multi, solo):
            got, gz, gtl = c.calc_prob(ps)
            assert gtl == wtl and np.array_equal(gz, wz) and _rel(got, want) <= 1e-13
            assert np.array_equal(c.read_probs(0), plain.read_probs(0))
    for c in (multi, solo):
        for (p, z, tl), ps in zip(c.calc_prob_batch(api.BatchPaths(sets)), sets):
            assert _rel(p, plain.calc_prob(ps)[0]) <= 1e-13
    # the three carriers of the hot-path exchange: ranks' partials gathered and added up in rank order by every rank (the
    # default), ncclAllReduce(sum), the calling thread adding the shards' pinned-host partials in rank order. The first and
    # the last add the same doubles in the same order: bit for bit the same value (ties in the annealing loop
    # must be stable); with one rank the all-reduce has nothing to re-order either.
    vals = {}
    for mode in ("rccl", "rccl-allreduce", "host", "rccl"):
        multi.set_exchange(mode)
        assert multi.exchange() == mode
        vals[mode] = [multi.calc_prob(ps)[0] for ps in sets] + [b[0] for b in multi.calc_prob_batch(api.BatchPaths(sets))]
    assert vals["rccl"] == vals["host"] == vals["rccl-allreduce"]
    solo.set_exchange("rccl-allreduce")
    assert [solo.calc_prob(ps)[0] for ps in sets] == vals["rccl"][:2]
    solo.set_exchange("rccl")
    # a failure on a rank BEFORE the exchange reaches every rank through it (here: the one rank; a bad node id)
    for c in (multi, solo):
        with pytest.raises(api.GamlHipError):
            c.calc_prob([[walk[0], 10 ** 6]])
        assert _rel(c.calc_prob(sets[0])[0], plain.calc_prob(sets[0])[0]) <= 1e-13  # ... and the context goes on working
    multi.close()
    solo.close()


def test_cpp_caller_over_two_shards(tmp_path):
    """The C++ host mirror (GAML config file -> ProbCalculator -> C ABI) with GAML_HIP_DEVICES=0,0: two in-process
    shards, same value as one device."""
    from test_gpu_cli import CLI, _write_case
    d = str(tmp_path)
    _write_case(d)
    outs = {}
    for devs in ("0", "0,0"):
        env = dict(os.environ, GAML_HIP_DEVICES=devs)
        out = subprocess.run([CLI, os.path.join(d, "run.cfg")], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr
        m = re.search(r"start prob (\S+) len (\d+) low prob reads(.*)", out.stdout)
        outs[devs] = (float(m.group(1)), int(m.group(2)), m.group(3).strip())
        assert ("2 device shard(s), exchange host" in out.stdout) == (devs == "0,0")
    assert outs["0"][1:] == outs["0,0"][1:]
    assert _rel(outs["0,0"][0], outs["0"][0]) <= 1e-12
