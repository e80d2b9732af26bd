"""gaml_amd.dist.ShardedScorer over a real RCCL process group (one rank: the pool's boxes have one GPU;
the driver runs N = 2, 4, 8 at round end). The context is told that a peer exists (presharded=2), so
the whole protocol runs: maxima exchange on the cold path, coverage-map all-gather for the penalty,
all-reduce of the partials. With the peer holding no reads the result must equal the plain context's."""
import os
import socket

import numpy as np
import pytest

from gaml_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_sharded_scorer_single_rank_group():
    import torch
    import torch.distributed as dist
    from gaml_amd import api
    from gaml_amd.dist import ShardedScorer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        genome = synth.make_genome(60_000, 71)
        g = synth.make_graph(genome, synth.cut_lengths(60_000, 71, long_rng=(900, 4000)))
        pr = synth.make_paired_reads(genome, 1200, 100, 250.0, 25.0, 0.01, 71)
        args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        walk = synth.genome_walk(g)
        for penalty in (0.0, 0.0007):
            cfg = api.paired_cfg(250.0, 25.0, penalty_constant=penalty, penalty_step=40.0)
            plain = api.Context(device=0)
            plain.set_graph(*g.packed())
            plain.add_paired(cfg, *args)
            ctx = api.Context(device=0, presharded=2)
            ctx.set_graph(*g.packed())
            ctx.add_paired(cfg, *args)
            scorer = ShardedScorer(ctx)
            for paths in ([walk], [walk[:6], walk[6:]], [walk[3:]]):
                want, wz, tl = plain.calc_prob(paths)
                got, z, tl2 = scorer.calc_prob(paths)
                assert tl == tl2 and z.tolist() == wz.tolist()
                assert abs(got - want) <= 1e-12 * abs(want), (penalty, got, want)
            if penalty:
                assert plain.bad_bases(0) > 0
            sets = [[walk], [walk[:6], walk[6:]], [walk[3:]], [walk]]
            single = [plain.calc_prob(p) for p in sets]
            for got, want in zip(scorer.calc_prob_batch(sets), single):
                assert got[2] == want[2] and got[1].tolist() == want[1].tolist()
                assert abs(got[0] - want[0]) <= 1e-12 * abs(want[0])
    finally:
        dist.destroy_process_group()


def _long_reads(g, walk):
    ps = synth.make_pacbio_sam(g, walk, 31, 2500, 79, secondary=0.0)
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    return ps, rb, ro


def _two_rank_worker(rank, world, port, out_dir, penalty):
    """One of `world` processes sharing the single GPU of the test box. Collectives run over gloo (RCCL
    refuses two ranks on one device); everything else -- one context per process holding its shard,
    maxima exchange, coverage maps, partial sums -- is the multi-GPU path as it runs on a node."""
    import json
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gaml_amd import api, synth
    from gaml_amd.dist import ShardedScorer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        genome = synth.make_genome(60_000, 73)
        g = synth.make_graph(genome, synth.cut_lengths(60_000, 73, long_rng=(900, 4000)))
        pr = synth.make_paired_reads(genome, 1501, 100, 250.0, 25.0, 0.01, 73)  # odd: uneven shards
        walk = synth.genome_walk(g)
        ctx = api.Context(device=0, rank=rank, world=world)
        ctx.set_graph(*g.packed())
        ctx.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=penalty, penalty_step=40.0),
                       *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        ps, rb, ro = _long_reads(g, walk)
        pb = ctx.add_pacbio_reads(api.single_cfg(penalty_constant=4 * penalty, penalty_step=30.0, min_prob_per_base=-1.0, weight=0.5,
                                                 mismatch_prob=0.15), rb, ro, ps.names)
        ctx.pacbio_ingest_sam(pb, walk, ps.sam)
        scorer = ShardedScorer(ctx)
        sets = [[walk], [walk[:6], walk[6:]], [walk[3:]], [walk[:6], [x ^ 1 for x in reversed(walk[6:])]]]
        single = [scorer.calc_prob(p) for p in sets]          # cold: windows aligned, maxima exchanged
        batch = scorer.calc_prob_batch(sets)                  # warm, one all-reduce for the four sets
        res = {"single": [[v[0], v[1].tolist(), v[2]] for v in single], "batch": [[v[0], v[1].tolist(), v[2]] for v in batch]}
        if penalty == 0.0:
            # the single-node exchange: blocking evaluation per rank + sum through shared memory (gaml_hip_shm_*)
            host = ShardedScorer(ctx, host_exchange=f"/gaml_test_{port}")
            res["host_exchange"] = [[v[0], v[1].tolist(), v[2]] for v in (host.calc_prob(p) for p in sets)]
            host.close()
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("penalty", [0.0, 0.0007])
def test_two_processes_share_the_reads(tmp_path, penalty):
    import json
    import torch.multiprocessing as mp
    from gaml_amd import api
    world, port = 2, _free_port()
    mp.spawn(_two_rank_worker, args=(world, port, str(tmp_path), penalty), nprocs=world, join=True)
    ranks = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert ranks[0] == ranks[1]  # every rank ends up with the same values
    if penalty == 0.0:  # two ranks: a + b either way, so the shared-memory exchange gives the collective's bits
        assert ranks[0]["host_exchange"] == ranks[0]["single"]
    # the unsharded context on the same inputs
    genome = synth.make_genome(60_000, 73)
    g = synth.make_graph(genome, synth.cut_lengths(60_000, 73, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, 1501, 100, 250.0, 25.0, 0.01, 73)
    walk = synth.genome_walk(g)
    plain = api.Context(device=0)
    plain.set_graph(*g.packed())
    plain.add_paired(api.paired_cfg(250.0, 25.0, penalty_constant=penalty, penalty_step=40.0),
                     *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    ps, rb, ro = _long_reads(g, walk)
    pb = plain.add_pacbio_reads(api.single_cfg(penalty_constant=4 * penalty, penalty_step=30.0, min_prob_per_base=-1.0, weight=0.5,
                                               mismatch_prob=0.15), rb, ro, ps.names)
    plain.pacbio_ingest_sam(pb, walk, ps.sam)
    sets = [[walk], [walk[:6], walk[6:]], [walk[3:]], [walk[:6], [x ^ 1 for x in reversed(walk[6:])]]]
    for k, p in enumerate(sets):
        want, wz, tl = plain.calc_prob(p)
        for kind in ("single", "batch"):
            got, z, tl2 = ranks[0][kind][k]
            assert tl2 == tl and z == wz.tolist()
            assert abs(got - want) <= 1e-12 * abs(want), (kind, k, got, want)


def _cold_alignment_worker(rank, world, port, out_dir, repeats):
    """Two processes on the one GPU, `repeats` fresh contexts each: the cold evaluation aligns every window through the
    small-batch pipeline while the peer does the same (what exposed the hits' publication race of round 3)."""
    import json
    import sys
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from gaml_amd import api, synth
    from gaml_amd.dist import ShardedScorer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        genome = synth.make_genome(60_000, 73)
        g = synth.make_graph(genome, synth.cut_lengths(60_000, 73, long_rng=(900, 4000)))
        pr = synth.make_paired_reads(genome, 1501, 100, 250.0, 25.0, 0.01, 73)
        walk = synth.genome_walk(g)
        out = []
        for _ in range(repeats):
            ctx = api.Context(device=0, rank=rank, world=world)
            ctx.set_graph(*g.packed())
            ctx.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
            scorer = ShardedScorer(ctx)
            v = scorer.calc_prob([walk])
            out.append([v[0], v[1].tolist(), v[2]])
            dist.barrier()
            ctx.close()
        with open(os.path.join(out_dir, f"cold{rank}.json"), "w") as f:
            json.dump(out, f)
    finally:
        dist.destroy_process_group()


def test_two_processes_cold_alignment_repeated(tmp_path):
    import json
    import torch.multiprocessing as mp
    from gaml_amd import api
    repeats, world, port = 4, 2, _free_port()
    mp.spawn(_cold_alignment_worker, args=(world, port, str(tmp_path), repeats), nprocs=world, join=True)
    ranks = [json.load(open(tmp_path / f"cold{r}.json")) for r in range(world)]
    genome = synth.make_genome(60_000, 73)
    g = synth.make_graph(genome, synth.cut_lengths(60_000, 73, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, 1501, 100, 250.0, 25.0, 0.01, 73)
    plain = api.Context(device=0)
    plain.set_graph(*g.packed())
    plain.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    want, wz, tl = plain.calc_prob([synth.genome_walk(g)])
    for k in range(repeats):
        assert ranks[0][k] == ranks[1][k]
        got, z, tl2 = ranks[0][k]
        assert tl2 == tl and z == wz.tolist(), (k, z, wz.tolist())  # every hit of every block arrived: the floored counts are exact
        assert abs(got - want) <= 1e-12 * abs(want)
