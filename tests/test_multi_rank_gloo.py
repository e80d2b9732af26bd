"""N > 1 path on CPU: 2 processes (gloo), each owning one shard of the reads.

The kernels need a GPU, so here each rank's per-shard partials {sum of log-probabilities, floored
reads, bad_bases, reads} are produced by the oracle over the rank's own reads; what is under test
is the product's sharding contract: gaml_hip_set_shard's read partition, ONE all-reduce(sum) of
the 4 doubles per read set, and gaml_hip_combine_partials turning the reduced partials into
CalcProb's value -- which must equal the unsharded oracle value."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_py as op
    from gaml_amd import api, synth

    G, n, seed = 50_000, 2501, 41  # odd read count: uneven shards
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(700, 3000)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, 0.01, seed)
    gb, go = g.packed()
    walk = synth.genome_walk(g)
    paths = [walk[:9], walk[9:]]
    lo, hi = n * rank // world, n * (rank + 1) // world

    # product side: a sharded (host-only) context -- checks the partition the library makes
    ctx = api.Context(device=-1, rank=rank, world=world)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(250.0, 25.0, weight=0.75), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    # cold path of a sharded run: exchange the largest record position of every new window
    pending, tl_ctx = ctx.eval_begin(paths)
    mine = torch.from_numpy(ctx.eval_pending_maxpos().copy())
    assert pending == mine.numel() > 0
    red = mine.clone()
    dist.all_reduce(red, op=dist.ReduceOp.MAX)
    assert (red >= mine).all() and (red > mine).any()  # the other shard really knows larger positions
    ctx.eval_apply_maxpos(red.numpy())
    assert ctx.eval_begin(paths)[0] == 0  # warm: nothing left to exchange
    ids = np.concatenate([ctx.window_records(rs, 0, ctx.debug_window_walk(rs, 0, w))[:, 2] for w in range(ctx.window_count(rs, 0))])
    assert ids.size and ids.min() >= lo and ids.max() < hi

    # per-shard partials from the oracle restricted to this rank's reads
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(*synth.pack_reads(pr.mate1[lo:hi]), *synth.pack_reads(pr.mate2[lo:hi]), 0.01, op.paired_cfg(250.0, 25.0))
    mean_log, zeros, tl = orc.calc_prob(paths, fresh=True)
    part = torch.tensor([mean_log * (hi - lo), float(zeros[0][0]), 0.0, float(hi - lo)], dtype=torch.float64)
    mine_part = part.clone()
    dist.all_reduce(part, op=dist.ReduceOp.SUM)  # the single collective of the path
    prob, z = ctx.combine_partials(part.numpy(), tl)
    # the single-node carrier of the same sum: POSIX shared memory (gaml_hip_shm_*), rank 0 opens first
    name = f"/gaml_cpu_test_{port}"
    if rank == 0:
        ctx.shm_exchange_open(name, 0, world, 4)
    dist.barrier()
    if rank != 0:
        ctx.shm_exchange_open(name, rank, world, 4)
    dist.barrier()
    everyone = [torch.empty_like(mine_part) for _ in range(world)]
    dist.all_gather(everyone, mine_part)
    for step in range(300):  # many rounds: the two parities of the block get reused, ranks run ahead of each other
        v = (mine_part + step).numpy().copy()
        ctx.shm_allreduce_sum(v.ctypes.data, 4)
        want = torch.zeros(4, dtype=torch.float64)
        for r in range(world):  # rank order, as the exchange adds them
            want = want + (everyone[r] + step)
        assert v.tolist() == want.tolist(), (rank, step)
    dist.barrier()
    ctx.shm_exchange_close(unlink_name=rank == 0)

    if rank == 0:
        full = op.Oracle()
        full.set_graph(gb, go)
        full.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(250.0, 25.0, weight=0.75))
        want, wz, wtl = full.calc_prob(paths, fresh=True)
        ok = abs(prob - want) <= 1e-12 * abs(want) and z.tolist() == wz.tolist() and tl == wtl
        with open(os.path.join(out_dir, "result.txt"), "w") as f:
            f.write(f"{int(ok)} {prob!r} {want!r} {z.tolist()} {wz.tolist()}\n")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_calc_prob(built, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    line = open(tmp_path / "result.txt").read().split()
    assert line[0] == "1", line
