#!/usr/bin/env python3
"""bench.py -- reads/s scored by the MI355X likelihood path (BASELINE.json metric).

A STEP is one ProbCalculator::CalcProb over the whole resident read shard: host builds the
window-occurrence tables for the path set, the GPU rescoring every read pair from scratch
(paired_score_kernel), the value comes back to the host (CalcProb is a blocking call in the
reference, gaml.cc:284). Consecutive steps score DIFFERENT path sets (the genome walk broken at
rotating points, as simulated-annealing moves do) so nothing can be reused from the previous
step; the alignment-window cache is warm (all windows aligned before the timed region), which
is the "warm from-scratch" evaluation SURVEY.md 8d defines as what the kernels replace.

N > 1: one process per GPU (torch.distributed / RCCL). Every rank holds its own shard of
reads (weak scaling: 833,333 pairs per GPU), no data-path collective, one all-reduce(sum) of
the 4 partial doubles per step.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def path_variants(walk, k=8):
    """k path sets: the whole walk, and the walk broken in two at k-1 rotating points."""
    out = [[list(walk)]]
    n = len(walk)
    for i in range(1, k):
        cut = (n * i) // k
        cut -= cut % 2  # keep the long/short alternation aligned
        cut = max(1, min(n - 1, cut))
        out.append([list(walk[:cut]), list(walk[cut:])])
    return out


def cpu_baseline(gb, go, b1, o1, b2, o2, sample_pairs, read_len, variants, budget_s=10.0):
    """Oracle (CPU restatement of the reference, 1 thread) timed on a bounded sample of the same
    workload: the first `sample_pairs` pairs against the full graph, warm from-scratch CalcProb
    (fresh ScoringState, window cache hot)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    orc = op.Oracle()
    orc.set_graph(gb, go)
    nb = sample_pairs * read_len
    rs = orc.add_paired(b1[:nb], o1[:sample_pairs + 1], b2[:nb], o2[:sample_pairs + 1], 0.01, op.paired_cfg(300.0, 30.0))
    t0 = time.time()
    vals = [orc.calc_prob(v, fresh=True)[0] for v in variants]  # cold: aligns every window
    cold_s = time.time() - t0
    n_eval, t_warm = 0, 0.0
    while t_warm < budget_s and n_eval < 2000:  # ~10 s of CPU work (+ the cold pass): a bounded sample, not the full set
        v = variants[n_eval % len(variants)]
        t0 = time.time()
        orc.calc_prob(v, fresh=True)
        t_warm += time.time() - t0
        n_eval += 1
    reads_per_s = 2.0 * sample_pairs * n_eval / t_warm
    return {"value": reads_per_s, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"first {sample_pairs} pairs of rank 0's reads vs the full graph, {n_eval} warm from-scratch "
                      f"CalcProb calls ({t_warm:.1f} s); cold pass incl. window alignment {cold_s:.1f} s "
                      f"({2.0 * sample_pairs * len(variants) / cold_s:.0f} reads/s)"}, vals, rs, orc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "tiny"])
    ap.add_argument("--cpu-sample-pairs", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-GPU code path (process group, side stream, "
                    "all-reduce) even with one rank -- rehearsal of the N > 1 path on a 1-GPU box")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from gaml_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # Rehearsal of N > 1 on a box with fewer GPUs than ranks (never used by the driver): GAML_BENCH_SHARE_GPU=1
    # puts every rank on cuda:0 and runs the collectives over gloo (RCCL refuses two ranks on one device).
    share_gpu = os.environ.get("GAML_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = synth.WORKLOADS[args.workload]
    # same genome + graph on every rank; each rank draws its own reads (its shard of the N x larger read set)
    genome = synth.make_genome(wl.genome_len, wl.seed)
    g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed + 1000 * rank)
    gb, go = g.packed()
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    variants = path_variants(walk)

    ctx = api.Context(device=local_rank, presharded=world)  # every rank hands over only its own reads
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), b1, o1, b2, o2)
    n_pairs_rank = wl.n_pairs

    # N > 1: gaml_amd.dist.ShardedScorer runs the kernels on a torch side stream (a real, non-null HIP
    # stream) so that the all-reduce and the D2H copy are ordered after them
    scorer = None
    in_loop = [False]
    if use_dist:
        from gaml_amd.dist import ShardedScorer
        # one node (the driver's launch): the partials are summed through shared memory after a blocking evaluation;
        # GAML_BENCH_EXCHANGE=rccl keeps them on the device (finisher kernel -> RCCL all-reduce -> fetch)
        single_node = int(os.environ.get("LOCAL_WORLD_SIZE", world)) == world
        exchange = os.environ.get("GAML_BENCH_EXCHANGE", "shm" if single_node else "rccl")
        scorer = ShardedScorer(ctx, host_exchange=("/gaml_bench_%s" % os.environ.get("MASTER_PORT", "0")) if exchange == "shm" else None)
    torch.cuda.synchronize()

    def step(paths):
        if scorer is not None:
            if in_loop[0]:  # the side stream is current: the lean form (cold path: maxima exchange; every step: one all-reduce(sum) of 4 f64)
                return scorer.score(paths), None
            prob, zeros, _ = scorer.calc_prob(paths)
            return prob, zeros
        # ONE ABI call, like the reference's CalcProb(paths): registration, kernels, the per-block partials
        # land in pinned host memory, the library adds them up and returns the value (blocking)
        return ctx.score(paths), None

    # prime: align every window any variant needs (cold path, untimed), then warm-up steps
    t0 = time.time()
    variants_py = variants
    variants = [api.FlatPaths(v) for v in variants]  # the ABI's form, built once (as a C++ caller holds it)
    vals = [step(v)[0] for v in variants]
    ctx.compact_tables()  # steady state: fold what the priming calls aligned into the device tables (the library would after 64 quiet calls)
    step(variants[0])     # ... which happens at the next evaluation: keep it out of the warm-up / timed steps
    prime_s = time.time() - t0
    for i in range(args.warmup):
        step(variants[i % len(variants)])

    ctx.set_event_timing(8)  # every 8th launch: events attached to a dispatch cost ~4 us of host time each
    ctx.kernel_stats(reset=True)
    # a step is ~70 us: one generation-2 pass of Python's garbage collector over the interpreter's
    # (torch-sized) object graph costs ~40 ms, i.e. hundreds of steps -- keep it out of the timed loop
    import gc
    gc.collect()
    gc.disable()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    import contextlib
    # N > 1: the scorer's side stream is made current once for the whole loop (not once per step)
    loop_ctx = torch.cuda.stream(scorer.stream) if scorer is not None else contextlib.nullcontext()
    t0 = time.perf_counter()
    last = None
    stamps = [0.0] * (args.steps + 1)
    stamps[0] = t0
    with loop_ctx:
        in_loop[0] = True
        for i in range(args.steps):
            last = step(variants[i % len(variants)])
            stamps[i + 1] = time.perf_counter()
        in_loop[0] = False
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    ks = ctx.kernel_stats(reset=True)
    ctx.set_event_timing(False)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_reads = 2.0 * n_pairs_rank * world
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_reads * args.steps / elapsed
        launches = max(1, ks["launches"])
        kern_us = ks["device_us"] / launches
        bytes_per_launch = ks["algo_bytes"] / launches
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        # HBM traffic per launch: not measurable from inside the process; taken from the committed rocprofv3
        # --pmc passes of this same command (profiles/*pmc_traffic.json), FETCH_SIZE doubled per the guide
        traffic, traffic_src = None, None
        try:
            import glob
            f = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))[-1]
            tj = json.load(open(f))
            if args.workload == "cfg3":
                traffic = tj["kernels"]["paired_score_kernel"]["hbm_bytes_per_launch"]
                traffic_src = os.path.relpath(f, ROOT)
        except Exception:
            pass
        out = {
            "metric": "reads_per_sec_scored", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name + f"; {n_pairs_rank} pairs per GPU, whole true genome as 1-2 walks, "
                                             "8 rotating path sets, window cache warm", "pairs_per_gpu": n_pairs_rank,
                       "genome_bp": wl.genome_len, "parallelism": f"reads sharded over {world} GPU(s), 1 all-reduce of 32 B/step"
                       + ("" if scorer is None else (" through shared memory (one node; blocking evaluation per rank)" if scorer._host_exchange else " over RCCL"))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "paired_score_kernel", "kernel_us": kern_us, "algo_bytes_per_launch": bytes_per_launch,
                         "timing": "HIP events attached to the dispatch (hipExtLaunchKernelGGL, on the launch stream) of every 8th "
                                   "paired_score_kernel launch of the timed region: the kernel's own begin/end stamps, as in "
                                   "rocprofv3's kernel trace",
                         "timed_launches": int(launches),
                         "traffic_source": traffic_src},
            "step_us_distribution": (lambda d: {"p50": float(np.percentile(d, 50)), "p90": float(np.percentile(d, 90)),
                                                "p99": float(np.percentile(d, 99)), "max": float(d.max())})(
                np.diff(np.array(stamps)) * 1e6),
            "log_likelihood": last[0], "prime_s": prime_s,
            "pair_classes_le1_le2_le4_more": [int(x) for x in ctx.debug_class_counts(rs)],
            "timing_last_step_us": ctx.last_timing(),
        }
        if not use_dist:
            # extra, outside the timed region: the same 8 path sets through gaml_hip_calc_prob_batch
            #  -- what a move generator gets that compares several candidate assemblies
            bp = api.BatchPaths(variants_py)
            bvals = [b[0] for b in ctx.calc_prob_batch(bp)]
            assert all(abs(a - b) <= 1e-12 * abs(b) for a, b in zip(bvals, vals)), (bvals, vals)
            calls = max(4, args.steps // (8 * 4))
            gc.disable()
            tb = time.perf_counter()
            for _ in range(calls):
                ctx.calc_prob_batch(bp)
            tb = time.perf_counter() - tb
            gc.enable()
            out["batched"] = {"api": "gaml_hip_calc_prob_batch", "sets_per_call": len(variants_py), "calls": calls,
                              "ms_per_set": 1e3 * tb / (calls * len(variants_py)),
                              "reads_per_sec": total_reads * calls * len(variants_py) / tb}
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is taken on rank 0 at N = 1 only
            sample = min(args.cpu_sample_pairs, n_pairs_rank)
            cb, cpu_vals, _, _ = cpu_baseline(gb, go, b1, o1, b2, o2, sample, wl.read_len, variants_py)
            out["cpu_baseline"] = cb
            # log-likelihood delta vs the CPU reference restatement on the same sample of reads
            c2 = api.Context(device=local_rank)
            c2.set_graph(gb, go)
            nb = sample * wl.read_len
            c2.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), b1[:nb], o1[:sample + 1], b2[:nb], o2[:sample + 1])
            gpu_vals = [c2.calc_prob(v)[0] for v in variants]
            out["ll_max_rel_delta_vs_cpu"] = max(abs(a - b) / abs(b) for a, b in zip(gpu_vals, cpu_vals))
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        if scorer is not None:
            scorer.close()  # unlinks the shared-memory name (rank 0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
