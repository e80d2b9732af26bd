#!/usr/bin/env python3
"""bench.py -- reads/s scored by the MI355X likelihood path (BASELINE.json metric).

A STEP is one ProbCalculator::CalcProb over the whole read set: the host builds the window-occurrence tables for the
path set, the GPU(s) rescore every read pair from scratch (paired_score_kernel), the value comes back to the host
(CalcProb is a blocking call in the reference, gaml.cc:284). Consecutive steps score DIFFERENT path sets (the genome
walk broken at rotating points, as simulated-annealing moves do) so nothing can be reused from the previous step; the
alignment-window cache is warm (all windows aligned before the timed region): the "warm from-scratch" evaluation
SURVEY.md 8d defines as what the kernels replace.

N = 1: BASELINE config 3's read set (5 Mbp, 833,333 pairs 2x150, insert 300+-30) on one GPU -- the configuration the
north star's target is quoted on.
N > 1 (one process per GPU, torch.distributed.run): BASELINE config 3 as stated -- the SAME 833,333 pairs (same seed as
N = 1) split over the N GPUs by gaml_hip_set_shard ("scaling": "strong"), one in-library RCCL all-reduce(sum) of 4 f64
per step on the library's stream (gaml_hip_comm_init_rank; the id travels over torch.distributed). `--scaling weak`
(833,333 fresh pairs per GPU) and the other exchanges (GAML_BENCH_EXCHANGE=rccl-torch|shm) are explicitly labelled
alternatives.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import contextlib
import gc
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (~6.3 TB/s achievable)


def source_hash() -> str:
    """What the committed rocprofv3 summaries must have been taken on to describe this run: the kernel sources + this file."""
    h = hashlib.sha256()
    rels = subprocess.check_output(["make", "-s", "-C", os.path.join(ROOT, "gaml_amd", "csrc"), "print-srcs"], text=True).split()
    for rel in ["bench.py"] + rels:  # the Makefile's list: every file in csrc/ and include/
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


@contextlib.contextmanager
def stdout_to_stderr():
    """RCCL prints a version banner on the process's stdout when a communicator is made; the contract is ONE JSON line there."""
    sys.stdout.flush()
    saved = os.dup(1)
    try:
        os.dup2(2, 1)
        yield
    finally:
        sys.stdout.flush()
        try:  # the banner goes through C stdio: out of libc's buffer while fd 1 still points at stderr
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(saved, 1)
        os.close(saved)


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


def cpu_baseline(gb, go, b1, o1, b2, o2, pairs, read_len, variants, cfg, budget_s=10.0):
    """Oracle (CPU restatement of the reference, 1 thread) on the same workload: `pairs` pairs (default: all of them)
    against the full graph; one cold pass per path set (aligns every window), then warm from-scratch CalcProb calls
    (fresh ScoringState, window cache hot -- what the GPU step replaces) for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    orc = op.Oracle()
    orc.set_graph(gb, go)
    nb = pairs * read_len
    orc.add_paired(b1[:nb], o1[:pairs + 1], b2[:nb], o2[:pairs + 1], 0.01, op.paired_cfg(*cfg))
    t0 = time.time()
    vals = [orc.calc_prob(v, fresh=True)[0] for v in variants]  # cold: aligns every window
    cold_s = time.time() - t0
    n_eval, t_warm = 0, 0.0
    while (t_warm < budget_s or n_eval < 3) and n_eval < 2000:
        v = variants[n_eval % len(variants)]
        t0 = time.time()
        orc.calc_prob(v, fresh=True)
        t_warm += time.time() - t0
        n_eval += 1
    return {"value": 2.0 * pairs * n_eval / t_warm, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{pairs} pairs vs the full graph, {n_eval} warm from-scratch CalcProb calls in {t_warm:.1f} s "
                      f"({1e3 * t_warm / n_eval:.0f} ms each); cold pass over the {len(variants)} path sets incl. window "
                      f"alignment {cold_s:.1f} s"}, vals


def cpu_shard_child(args):
    """`--cpu-shard-child R,N`: one of N CPU processes of the courtesy row (no GPU, no torch): the oracle on reads
    [pairs R/N, pairs (R+1)/N) of the workload, cold pass over the 8 path sets, then warm from-scratch CalcProb calls for
    the budget. Prints one JSON line."""
    from gaml_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    r, n = (int(x) for x in args.cpu_shard_child.split(","))
    wl = synth.WORKLOADS[args.workload]
    genome, g = wl.build()
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    lo, hi = wl.n_pairs * r // n, wl.n_pairs * (r + 1) // n
    b1, o1 = synth.pack_reads(pr.mate1[lo:hi])
    b2, o2 = synth.pack_reads(pr.mate2[lo:hi])
    variants = path_variants(synth.genome_walk(g))
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    orc.add_paired(b1, o1, b2, o2, wl.err, op.paired_cfg(wl.insert_mean, wl.insert_std))
    t0 = time.time()
    for v in variants:
        orc.calc_prob(v, fresh=True)
    cold_s = time.time() - t0
    n_eval, t_warm = 0, 0.0
    while t_warm < args.cpu_shard_budget or n_eval < 3:
        t0 = time.time()
        orc.calc_prob(variants[n_eval % len(variants)], fresh=True)
        t_warm += time.time() - t0
        n_eval += 1
    print("CPUSHARD " + json.dumps({"rank": r, "pairs": hi - lo, "evals": n_eval, "warm_s": t_warm, "cold_s": cold_s}), flush=True)


def cpu_baseline_sharded(workload, n_pairs, budget_s=8.0):
    """The courtesy row of BASELINE.md's plan: the same CPU restatement read-sharded over the box's cores, one process
    per core (the reference itself is single-threaded: its native mode is the 1-core row). Cores = this process's CPU
    affinity, at most 16 (a GPU box's share per GPU). Each process holds a contiguous shard of the reads against the full
    graph and re-scores it from scratch for `budget_s` seconds; the shards' rates add up."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-shard-child", f"{r},{cores}", "--workload", workload,
                               "--cpu-shard-budget", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(cores)]
    rate, evals, cold, done = 0.0, 0, 0.0, 0
    for p in procs:
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            continue
        line = [l for l in out.splitlines() if l.startswith("CPUSHARD ")]
        if p.returncode == 0 and line:
            d = json.loads(line[-1][9:])
            rate += 2.0 * d["pairs"] * d["evals"] / d["warm_s"]
            evals += d["evals"]
            cold = max(cold, d["cold_s"])
            done += 1
    return {"value": rate, "unit": "reads/s", "cores": cores, "kind": "port", "processes_finished": done,
            "sample": f"{n_pairs} pairs split into {cores} contiguous shards, one oracle process per core against the full graph, "
                      f"{evals} warm from-scratch CalcProb calls in ~{budget_s:.0f} s each process (cold pass incl. window alignment: {cold:.1f} s, slowest shard); "
                      "each shard's position filter sees its own reads' window maxima only: a throughput row, not a likelihood"}


def sa_pattern(ctx, rs, g, iters, api, synth):
    """BASELINE config 5's call pattern on this GPU (untimed for the headline): the reference's start state, then
    `iters` edited path sets, one blocking CalcProb each; new junction windows get aligned on the fly."""
    start, seq = synth.sa_sequence(g, iters)
    flat = [api.FlatPaths(p) for p in seq]
    t0 = time.perf_counter()
    ctx.calc_prob(start)
    first_s = time.perf_counter() - t0
    a0 = ctx.aligner_stats()
    t_stats0 = ctx.table_stats(rs)
    per = np.zeros(len(flat))
    aligned = np.zeros(len(flat), bool)  # calls that brought new windows (one cheap counter read per call, outside the call's clock)
    seen_windows = a0["windows"]
    gc.disable()
    t0 = time.perf_counter()
    for i, f in enumerate(flat):
        t1 = time.perf_counter()
        ctx.score(f)
        per[i] = time.perf_counter() - t1
        w = ctx.aligner_stats()["windows"]
        aligned[i] = w != seen_windows
        seen_windows = w
    total = float(per.sum())  # (the counter reads between the calls are not part of it)
    gc.enable()
    a1 = ctx.aligner_stats()
    t_stats1 = ctx.table_stats(rs)
    per *= 1e6
    # where a warm call of this pattern goes (replay of the last 200 sets: everything cached), library-side phases
    prof = []
    for f in flat[-200:]:
        ctx.score(f)
        prof.append(ctx.last_phases())
    ph = np.median(np.array(prof), axis=0)
    return {"iterations": iters, "paths_at_start": len(start), "total_s": total, "first_call_cold_s": first_s,
            "us_median": float(np.median(per)), "us_p90": float(np.percentile(per, 90)), "us_p99": float(np.percentile(per, 99)),
            "us_max": float(per.max()), "table_rebuilds": t_stats1["full_rebuilds"] - t_stats0["full_rebuilds"],
            "delta_updates": t_stats1["delta_updates"] - t_stats0["delta_updates"],
            "worker_rebuilds": t_stats1["worker_rebuilds"] - t_stats0["worker_rebuilds"],
            "windows_aligned": a1["windows"] - a0["windows"], "aligner_ms": (a1["us"] - a0["us"]) * 1e-3,
            # the p90 of 1000 calls is the 100th slowest: with ~90 calls that align new windows it IS what such a call costs
            "aligning_calls": int(aligned.sum()), "aligning_call_us_median": float(np.median(per[aligned])) if aligned.any() else None,
            "other_calls_us_median": float(np.median(per[~aligned])), "other_calls_us_p99": float(np.percentile(per[~aligned], 99)),
            "warm_call_phases_us": {"planning": float(ph[0]), "tables": float(ph[1]), "write": float(ph[3]), "launch": float(ph[5]),
                                    "wait": float(ph[7]), "bytes_written": float(ph[6])},
            "recipe": "gaml_amd.synth.sa_sequence(seed 7): BreakPath / join / reverse / LocalChange / duplicate / trim edits, 60 % accepted"}


def weak_scaling_pass(args, api, synth, dist, torch, wl, g, gb, go, cfg, variants_py, rank, world, local_rank, share_gpu, exchange):
    """The N > 1 step with per-GPU work fixed: every rank draws its OWN read set of the workload's size (seed + 1000 * rank),
    hands it over as its shard (presharded context), one exchange of the partials per step through the same carrier as
    the headline. Collective: every rank calls it. Returns the block for the JSON line (the same on every rank)."""
    import gc
    genome = wl.build()[0]  # (seeded: the same genome on every rank)
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed + 1000 * rank)
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    ctx = api.Context(device=local_rank, presharded=world)
    ctx.set_graph(gb, go)
    ctx.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
    scorer = None
    if exchange == "rccl":
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            idt = torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8).clone()
        if not share_gpu:
            idt = idt.cuda()
        with stdout_to_stderr():
            dist.broadcast(idt, src=0)
            ctx.comm_init_rank(bytes(idt.cpu().numpy().tobytes()), rank, world)
            torch.cuda.synchronize()
        if os.environ.get("GAML_BENCH_RCCL_FORM", "gather") == "allreduce":
            ctx.set_exchange("rccl-allreduce")
    else:
        from gaml_amd.dist import ShardedScorer
        shm = exchange == "shm"
        scorer = ShardedScorer(ctx, host_exchange=("/gaml_bench_w_%s" % os.environ.get("MASTER_PORT", "0")) if shm else None)
    in_loop = [False]

    def step(fp):
        if scorer is not None:
            return scorer.score(fp) if in_loop[0] else scorer.calc_prob(fp)[0]
        return ctx.score(fp)

    variants = [api.FlatPaths(v) for v in variants_py]
    for v in variants:
        step(v)
    ctx.compact_tables()
    step(variants[0])
    for v in variants:
        step(v)
    steps, warm = min(args.steps, 1000), min(args.warmup, 50)
    gc.collect()
    gc.disable()
    for i in range(warm):
        step(variants[i % len(variants)])
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    loop_ctx = torch.cuda.stream(scorer.stream) if (scorer is not None and not scorer._host_exchange) else contextlib.nullcontext()
    last = None
    t0 = time.perf_counter()
    with loop_ctx:
        in_loop[0] = True
        for i in range(steps):
            last = step(variants[i % len(variants)])
        in_loop[0] = False
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if scorer is not None:
        scorer.close()
    ctx.close()
    pairs_total = wl.n_pairs * world
    return {"scaling": "weak", "value": 2.0 * pairs_total * steps / elapsed, "unit": "reads/s", "ms_per_step": 1e3 * elapsed / steps, "steps": steps,
            "warmup": warm, "pairs_total": pairs_total, "pairs_per_gpu": wl.n_pairs, "log_likelihood_of_all_ranks_reads": last,
            "what": f"every rank scores a read set of its own of the workload's size ({wl.n_pairs} pairs; seeds differ per rank) against the same "
                    "graph and path sets; one exchange of the partials per step through the headline's carrier; barrier + synchronize on "
                    "both sides of the timed steps, MAX over ranks"}


def batched_candidates(ctx, g, api, synth, n_batches=40, per_batch=8):
    """What the move generators do (moves.cc:107-113, 694-800, 1156-1305): several single-edit candidates of ONE current
    assembly, scored together, the best kept. Base = a state of the annealing walk (~900 paths); every batch holds
    `per_batch` different edits of the current base; 60 % of the batches move the base on."""
    start, seq = synth.sa_sequence(g, 200, seed=11)
    base = seq[-1]
    rng = np.random.default_rng(23)
    batches = []
    for _ in range(n_batches):
        cands = [synth.sa_move(rng, base, g) for _ in range(per_batch)]
        batches.append(api.BatchPaths(cands))
        if rng.random() < 0.6:
            base = cands[int(rng.integers(0, per_batch))]
    ctx.calc_prob(base)
    for b in batches:  # cold pass: windows the candidates need get aligned
        ctx.calc_prob_batch(b)
    ctx.compact_tables()  # steady state, as for the headline: what the cold pass aligned is folded into the device tables
    for b in batches:     # ... at the next evaluation; one more untimed pass
        ctx.calc_prob_batch(b)
    gc.disable()
    t0 = time.perf_counter()
    for b in batches:
        ctx.calc_prob_batch(b)
    dt = time.perf_counter() - t0
    gc.enable()
    return {"api": "gaml_hip_calc_prob_batch", "sets_per_call": per_batch, "calls": n_batches, "paths_per_set": len(base),
            "ms_per_set": 1e3 * dt / (n_batches * per_batch), "pattern": "8 single-edit candidates of one ~900-path assembly per call"}


def drift_block(api, synth, device, g, b1, o1, b2, o2, read_len, cfg, iters=1000, sample=50_000):
    """The reference keeps its per-read probabilities across calls and updates them in place (probs[i] -= old term,
    += new term: graph.cc:1936-1950); the GPU path rescores every read from scratch. Over BASELINE config 5's call
    pattern, on the first `sample` pairs: the CPU oracle in the reference's INCREMENTAL mode against the GPU value along
    the same `iters` path sets -- how far the long-running state drifts, how many accept decisions `new > cur`
    (gaml.cc:286) come out differently, and how many reads end up scored from a rounding residue instead of the floor
    ("Incremental FP drift"). The oracle from scratch at checkpoints separates drift from GPU-vs-CPU delta."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    nb = sample * read_len
    sb1, so1, sb2, so2 = b1[:nb], o1[:sample + 1], b2[:nb], o2[:sample + 1]
    gb, go = g.packed()
    start, seq, parent = synth.sa_sequence_parents(g, iters)
    ctx = api.Context(device=device)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(*cfg), sb1, so1, sb2, so2)
    gpu0 = ctx.calc_prob(start)[0]
    gpu = [ctx.calc_prob(p)[0] for p in seq]
    gpu_probs_end = ctx.read_probs(rs)
    ctx.close()
    inc = op.Oracle()
    inc.set_graph(gb, go)
    irs = inc.add_paired(sb1, so1, sb2, so2, 0.01, op.paired_cfg(*cfg))
    t0 = time.time()
    ref0 = inc.calc_prob(start, fresh=False)[0]
    ref = [inc.calc_prob(p, fresh=False)[0] for p in seq]
    inc_s = time.time() - t0
    inc_probs_end = inc.paired_probs(irs)[0].copy()
    checks = sorted({iters - 1, iters // 2, iters // 10})
    fresh_delta = 0.0
    for k in checks:
        want = inc.calc_prob(seq[k], fresh=True)[0]  # (a fresh state on the same window cache; resets the incremental one)
        fresh_delta = max(fresh_delta, abs(gpu[k] - want) / abs(want))
    rel = np.array([abs(a - b) / abs(b) for a, b in zip(gpu, ref)])
    cur_gpu = [gpu0 if p < 0 else gpu[p] for p in parent]
    cur_ref = [ref0 if p < 0 else ref[p] for p in parent]
    margin = np.array([abs(b - cb) / abs(cb) for b, cb in zip(ref, cur_ref)])
    differ = np.array([(a > ca) != (b > cb) for a, ca, b, cb in zip(gpu, cur_gpu, ref, cur_ref)])
    flips = int(differ.sum())
    # a move that leaves the likelihood unchanged up to rounding (a path reversed, a gap trimmed) is a coin toss in BOTH
    # implementations: the reference's own result for it depends on the order of its += / -= updates
    flips_real = int((differ & (margin > 1e-12)).sum())
    residue = int(np.count_nonzero((gpu_probs_end == 0.0) & (inc_probs_end != 0.0)))
    neg = int(np.count_nonzero(inc_probs_end < 0.0))
    return {"pairs": sample, "iterations": iters, "what": "CPU oracle with the reference's incremental ScoringState (graph.cc:1936-1950) vs the GPU value, same path sets",
            "ll_rel_delta_max": float(rel.max()), "ll_rel_delta_mean": float(rel.mean()), "ll_rel_delta_last": float(rel[-1]),
            "accept_decisions_that_differ": flips, "of_which_with_a_margin_above_1e-12": flips_real,
            "moves_within_1e-12_of_a_tie": int((margin <= 1e-12).sum()),  # (most edits of the one-node start state touch no pair of the sample)
            "smallest_rel_margin_above_1e-12": float(margin[margin > 1e-12].min()) if (margin > 1e-12).any() else 0.0,
            "reads_scored_from_a_residue_at_the_end": residue, "reads_with_negative_probability_at_the_end": neg,
            "gpu_vs_fresh_oracle_rel_delta_max_at_checkpoints": fresh_delta, "checkpoints": checks, "oracle_incremental_s": inc_s}


def repeats_block(api, synth, device, sa_iters=5000, oracle=True):
    """Repeat-rich assemblies, untimed for the headline ("optionally with planted repeats"; GAML's repeat moves,
    moves.cc:1156-1305, exist for them): (1) cfg3r = config 3's recipe with 2 % of the genome in COLLAPSED 5-copy repeat
    families -- one node each, visited five times by the true walk, so its windows occur several times in the path set
    (the kernels' GEN instantiation scores their pairs where a lane meets them, by function calls) -- scored through the same 8 rotating path sets, with the likelihood against
    the CPU oracle on ALL pairs; (2) the late state of a long synthetic annealing walk at config 3 (duplicated nodes
    pile up: 60 % of the moves are accepted whatever they do)."""
    wl = synth.WORKLOADS["cfg3r"]
    genome, g = wl.build()
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    gb, go = g.packed()
    r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    ctx = api.Context(device=device)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *r1, *r2)
    walk = synth.genome_walk(g)
    variants_py = path_variants(walk)
    variants = [api.FlatPaths(v) for v in variants_py]
    vals = [ctx.score(v) for v in variants]
    ctx.compact_tables()
    for _ in range(2):
        vals = [ctx.score(v) for v in variants]
    gc.disable()
    t0 = time.perf_counter()
    n = 400
    for i in range(n):
        ctx.score(variants[i % 8])
    step_us = (time.perf_counter() - t0) / n * 1e6
    gc.enable()
    ctx.set_event_timing(1)
    ctx.kernel_stats(reset=True)
    for i in range(128):
        ctx.score(variants[i % 8])
    ks = ctx.kernel_stats(reset=True)
    ctx.set_event_timing(False)
    out = {"workload": wl.name, "pairs": wl.n_pairs, "walk_nodes": len(walk), "distinct_nodes": len(set(walk)),
           "pair_classes_le1_le2_le4_more": [int(x) for x in ctx.pair_classes(rs)],
           "step_us": step_us, "reads_per_sec": 2.0 * wl.n_pairs / (step_us * 1e-6),
           "scoring_kernel_us": ks["device_us"] / max(1, ks["launches"]),
           "launches_per_step": 1,  # (rounds 2-3: a second launch for the pairs on repeated windows, 20-25 us)
           "algo_bytes_per_launch": ks["algo_bytes"] / max(1, ks["launches"])}
    ctx.close()
    if oracle:
        # the CPU oracle on all pairs, two path sets (the whole walk; the walk in two pieces): cold, incl. window alignment
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as op
        orc = op.Oracle()
        orc.set_graph(gb, go)
        orc.add_paired(*r1, *r2, wl.err, op.paired_cfg(wl.insert_mean, wl.insert_std))
        t0 = time.time()
        want = [orc.calc_prob(v, fresh=True)[0] for v in variants_py[:2]]
        out["ll_max_rel_delta_vs_cpu"] = max(abs(a - b) / abs(b) for a, b in zip(vals[:2], want))
        out["ll_delta_pairs"] = wl.n_pairs
        out["cpu_oracle_s"] = time.time() - t0
        del orc
    # (2) late in a long annealing walk at config 3
    wl3 = synth.WORKLOADS["cfg3"]
    genome, g = wl3.build()
    pr = synth.make_paired_reads(genome, wl3.n_pairs, wl3.read_len, wl3.insert_mean, wl3.insert_std, wl3.err, wl3.seed)
    ctx = api.Context(device=device)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(wl3.insert_mean, wl3.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    start, seq = synth.sa_sequence(g, sa_iters)
    flat = [api.FlatPaths(p) for p in seq]
    ctx.calc_prob(start)
    per = np.zeros(len(flat))
    gc.disable()
    for k, f in enumerate(flat):
        t1 = time.perf_counter()
        ctx.score(f)
        per[k] = time.perf_counter() - t1
    gc.enable()
    ctx.set_event_timing(1)
    ctx.kernel_stats(reset=True)
    for f in flat[-200:]:
        ctx.score(f)
    ks = ctx.kernel_stats(reset=True)
    out["late_annealing_walk"] = {"iterations": sa_iters, "paths_at_end": len(seq[-1]), "call_us_median_last_1000": float(np.median(per[-1000:]) * 1e6),
                                  "pair_classes_le1_le2_le4_more": [int(x) for x in ctx.pair_classes(rs)],
                                  "delta_pairs": ctx.table_stats(rs)["dirty_pairs"],
                                  "scoring_kernel_us": ks["device_us"] / max(1, ks["launches"])}
    ctx.close()
    return out


def jumping_block(api, synth, device, oracle_sets=2, oracle=True):
    """The reference's own example configuration has a jumping library (example.cfg:20-29: insert 3700 +- 350,
    penalty_constant 0.00013, penalty_step 3000, min_prob_start -80). Untimed for the headline: pair classes, the static
    share of the compact class, kernel and step time through the 8 rotating path sets, and the likelihood + bad_bases
    against the CPU oracle on ALL pairs."""
    wl = synth.WORKLOADS["cfg3j"]
    genome, g = wl.build()
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    gb, go = g.packed()
    r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
    kw = dict(penalty_constant=0.00013, penalty_step=3000.0, min_prob_start=-80.0)  # (min_prob_per_base stays -0.7: gaml.cc:855 reads another key)
    ctx = api.Context(device=device)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std, **kw), *r1, *r2)
    walk = synth.genome_walk(g)
    variants_py = path_variants(walk)
    variants = [api.FlatPaths(v) for v in variants_py]
    t0 = time.perf_counter()
    vals = [ctx.calc_prob(v) for v in variants_py]
    cold_s = time.perf_counter() - t0
    bad = []
    for v in variants_py[:oracle_sets]:
        ctx.calc_prob(v)
        bad.append(ctx.bad_bases(rs))
    ctx.compact_tables()
    for _ in range(2):
        for v in variants:
            ctx.score(v)
    gc.disable()
    t0 = time.perf_counter()
    n = 400
    for i in range(n):
        ctx.score(variants[i % 8])
    step_us = (time.perf_counter() - t0) / n * 1e6
    gc.enable()
    ctx.set_event_timing(1)
    ctx.kernel_stats(reset=True)
    prof = []
    for i in range(128):
        ctx.score(variants[i % 8])
        prof.append(ctx.last_phases())
    ks = ctx.kernel_stats(reset=True)
    ctx.set_event_timing(False)
    ph = np.median(np.array(prof), axis=0)
    classes = [int(x) for x in ctx.pair_classes(rs)]
    st = ctx.table_stats(rs)
    out = {"workload": wl.name, "pairs": wl.n_pairs, "config": dict(kw, insert_mean=wl.insert_mean, insert_std=wl.insert_std, min_prob_per_base=-0.7),
           "pair_classes_le1_le2_le4_more": classes, "static_pairs": st["static_index_pairs"],
           "static_fraction_of_compact_class": st["static_index_pairs"] / max(1, classes[0]),
           "cold_8_sets_s": cold_s, "step_us": step_us, "reads_per_sec": 2.0 * wl.n_pairs / (step_us * 1e-6),
           "scoring_kernel_us": ks["device_us"] / max(1, ks["launches"]), "algo_bytes_per_launch": ks["algo_bytes"] / max(1, ks["launches"]),
           "step_phases_us": {"planning": float(ph[0]), "tables": float(ph[1]), "write": float(ph[3]), "launch": float(ph[5]), "wait": float(ph[7])},
           "note": "penalty_constant > 0: every call plans the whole path set, marks coverage (atomics into a bitmap) and runs coverage_sweep_kernel behind the scoring launch"}
    ctx.close()
    if not oracle:
        return out
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    orc = op.Oracle()
    orc.set_graph(gb, go)
    ors = orc.add_paired(*r1, *r2, wl.err, op.paired_cfg(wl.insert_mean, wl.insert_std, **kw))
    t0 = time.time()
    worst, bad_ok = 0.0, True
    for k, v in enumerate(variants_py[:oracle_sets]):
        want, wz, wtl = orc.calc_prob(v, fresh=True)
        worst = max(worst, abs(vals[k][0] - want) / abs(want))
        bad_ok = bad_ok and vals[k][1].tolist() == wz.tolist() and bad[k] == orc.paired_probs(ors)[1]
    out["ll_max_rel_delta_vs_cpu"] = worst
    out["floored_counts_and_bad_bases_equal_cpu"] = bool(bad_ok)
    out["ll_delta_pairs"] = wl.n_pairs
    out["cpu_oracle_s"] = time.time() - t0
    return out


def aligner_block(api, synth, device, g, gb, go, b1, o1, b2, o2, cfg, read_len, sa_iters=300):
    """SURVEY 8(d)'s separate figure for GPU window alignment: L read bytes + a 16-byte record per seed candidate. A fresh
    context: the cold batch (every window of the start assembly, both mates) and the small batches of an annealing walk
    (a move's new junction windows, both mates in one pipeline). Times are the host's clock around the aligner (window
    strings, launches, the device pipeline, the wait); the kernels' own durations are in profiles/ (rocprofv3 rows
    span_maxima_kernel / candidates_kernel / extend_kernel: cold; span_cands_kernel / extend_pair2_kernel / aln_file_small_kernel: small)."""
    ctx = api.Context(device=device)
    ctx.set_graph(gb, go)
    ctx.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
    start, seq = synth.sa_sequence(g, sa_iters)
    t0 = time.perf_counter()
    ctx.calc_prob(start)
    cold_call_s = time.perf_counter() - t0
    a0 = ctx.aligner_stats()
    s0 = ctx.aligner_stages()
    for p in seq:
        ctx.score(api.FlatPaths(p))
    a1 = ctx.aligner_stats()
    s1 = ctx.aligner_stages()
    ctx.close()
    per_cand = read_len + 16
    def row(w, k, us, batches):
        return {"windows": w, "candidates": k, "batches": batches, "aligner_us": us, "us_per_batch": us / max(1, batches),
                "candidates_per_sec": k / max(1e-9, us * 1e-6), "algo_bytes": k * per_cand,
                "GB_per_s": k * per_cand / max(1e-9, us * 1e-6) / 1e9, "frac_of_hbm_peak": k * per_cand / max(1e-9, us * 1e-6) / 1e9 / HBM_PEAK_GBS}
    cold = row(a0["windows"], a0["candidates"], a0["us"], s0["batches"])
    cold["stages_us"] = {k: s0[k] for k in ("strings_upload", "spans_candidates", "extension", "hits_d2h", "sort_file")}
    cold["first_call_cold_s"] = cold_call_s
    small = row(a1["windows"] - a0["windows"], a1["candidates"] - a0["candidates"], a1["us"] - a0["us"], s1["batches"] - s0["batches"])
    small["stages_us"] = {k: s1[k] - s0[k] for k in ("strings_upload", "spans_candidates", "extension", "hits_d2h", "sort_file")}
    return {"bytes_per_candidate": per_cand, "formula": "L read bytes + 16-byte record per seed candidate (SURVEY 8d); window bases staged in LDS, amortised",
            "cold_batch": cold, "small_batches": small,
            "note": "latency-bound, not bandwidth-bound: a small batch lasts as long as its slowest candidate's chain of LDS steps (extension) plus three dispatches; "
                    "the records stay in HBM (filed on the device), the host receives the windows' headers"}


def sa_long_block(api, synth, device, g, gb, go, b1, o1, b2, o2, cfg, read_len, iters=10000, sample=50_000, t0_temp=0.008, oracle=True):
    """A long annealing run with the reference's accept rule (example.cfg:3 max_iterations 10000): a move is accepted iff the
    GPU value improves; a worse one only after BreakPath, with probability exp((new - cur) / T), T = t0 / log(it + 1)
    (gaml.cc:274, 286, 304-311). Per 1,000 calls: median / p90 / max of the call, pairs on the delta lists, take-overs.
    At the end the final assembly's likelihood on the first 50,000 pairs against the CPU oracle (fresh state)."""
    ctx = api.Context(device=device)
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
    walk = synth.genome_walk(g)
    cur = [[x] for x in walk if g.node_len(x) > 500]
    cur_val = ctx.calc_prob(cur)[0]
    rng = np.random.default_rng(11)
    per = np.zeros(iters)
    accepted = 0
    windows = []
    st_prev = ctx.table_stats(rs)
    gc.disable()
    for it in range(iters):
        new, kind = synth.sa_move_kind(rng, cur, g)
        fp = api.FlatPaths(new)
        t1 = time.perf_counter()
        v = ctx.score(fp)
        per[it] = time.perf_counter() - t1
        temp = t0_temp / np.log(it + 2.0)
        if v > cur_val or (kind == 0 and rng.random() < np.exp((v - cur_val) / temp)):
            cur, cur_val = new, v
            accepted += 1
        if (it + 1) % 1000 == 0:
            w = per[it - 999:it + 1] * 1e6
            st = ctx.table_stats(rs)
            windows.append({"calls": [it - 999, it + 1], "us_median": float(np.median(w)), "us_p90": float(np.percentile(w, 90)), "us_max": float(w.max()),
                            "delta_pairs": st["dirty_pairs"], "take_overs": st["worker_rebuilds"] - st_prev["worker_rebuilds"],
                            "table_rebuilds": st["full_rebuilds"] - st_prev["full_rebuilds"], "paths": len(cur)})
            st_prev = st
    gc.enable()
    final_val = ctx.calc_prob(cur)[0]
    out = {"iterations": iters, "accept_rule": "new > cur; Metropolis exp((new - cur) / T), T = 0.008 / log(it + 1), after BreakPath only (gaml.cc:274, 286, 304-311)",
           "accepted": accepted, "paths_at_end": len(cur), "total_s": float(per.sum()), "windows": windows,
           "last_over_first_median": windows[-1]["us_median"] / windows[0]["us_median"], "log_likelihood_start_end": [float(ctx.calc_prob([[x] for x in walk if g.node_len(x) > 500])[0]), float(final_val)]}
    ctx.close()
    if not oracle:
        return out
    # the final assembly on the first `sample` pairs: GPU (a context of its own) against the CPU oracle
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as op
    nb = sample * read_len
    c2 = api.Context(device=device)
    c2.set_graph(gb, go)
    c2.add_paired(api.paired_cfg(*cfg), b1[:nb], o1[:sample + 1], b2[:nb], o2[:sample + 1])
    got = c2.calc_prob(cur)
    c2.close()
    orc = op.Oracle()
    orc.set_graph(gb, go)
    orc.add_paired(b1[:nb], o1[:sample + 1], b2[:nb], o2[:sample + 1], 0.01, op.paired_cfg(*cfg))
    want, wz, wtl = orc.calc_prob(cur, fresh=True)
    out["end_state_ll_rel_delta_vs_cpu"] = abs(got[0] - want) / abs(want)
    out["end_state_floored_equal"] = bool(got[1].tolist() == wz.tolist() and got[2] == wtl)
    out["end_state_pairs_checked"] = sample
    return out


def inproc_child(args):
    """`--inproc-devices 0,1,..`: ONE process, one context over those devices (gaml_hip_create_multi) -- what a gaml.cc
    linked against the adapter header runs. Same read set and steps as the headline; prints its own JSON line."""
    from gaml_amd import api, synth
    devs = [int(x) for x in args.inproc_devices.split(",")]
    wl = synth.WORKLOADS[args.workload]
    genome, g = wl.build()
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    ctx = api.Context(devices=devs)
    ctx.set_graph(*g.packed())
    ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    variants = [api.FlatPaths(v) for v in path_variants(synth.genome_walk(g))]
    out = {"devices": devs, "shards": ctx.num_shards(), "note": ctx.last_error()}
    for mode in (["rccl", "host"] if ctx.exchange() == "rccl" else ["host"]):
        ctx.set_exchange(mode)
        vals = [ctx.score(v) for v in variants]
        ctx.compact_tables()
        for i in range(args.warmup + 1):
            ctx.score(variants[i % len(variants)])
        gc.disable()
        t0 = time.perf_counter()
        for i in range(args.steps):
            ctx.score(variants[i % len(variants)])
        el = time.perf_counter() - t0
        gc.enable()
        out[mode] = {"ms_per_step": 1e3 * el / args.steps, "reads_per_sec": 2.0 * wl.n_pairs * args.steps / el, "log_likelihood": vals[0]}
    print("INPROC " + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "tiny", "cfg3r", "tinyr", "cfg3x8"])
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = BASELINE config 3's read set split over the GPUs (default); weak = a fresh read set of the same size per GPU")
    ap.add_argument("--cpu-sample-pairs", type=int, default=0, help="pairs the CPU baseline scores (0 = the whole read set)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sa", action="store_true", help="skip the annealing-pattern block (N = 1)")
    ap.add_argument("--no-repeats", action="store_true", help="skip the repeat-rich block (N = 1)")
    ap.add_argument("--no-long", action="store_true", help="skip the jumping-library, aligner and long-annealing blocks (N = 1)")
    ap.add_argument("--sa-long-iters", type=int, default=10000)
    ap.add_argument("--no-extras", action="store_true", help="timed region only (no kernel-timing pass, batch or annealing block): "
                    "what the rocprofv3 --pmc passes run, so that the last --steps dispatches are the timed steps")
    ap.add_argument("--sa-iters", type=int, default=1000)
    ap.add_argument("--kernel-samples", type=int, default=256, help="launches of the separate kernel-timing pass")
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path (process group, communicator, all-reduce) "
                    "with one rank -- rehearsal on a 1-GPU box")
    ap.add_argument("--no-inproc", action="store_true", help="N > 1: skip the single-process multi-device run on rank 0")
    ap.add_argument("--inproc-devices", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-shard-child", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-shard-budget", type=float, default=8.0, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-sharded", action="store_true", help="skip the N-process CPU courtesy row")
    args = ap.parse_args()
    if args.inproc_devices:
        return inproc_child(args)
    if args.cpu_shard_child:
        return cpu_shard_child(args)

    import torch
    import torch.distributed as dist
    from gaml_amd import api, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    # Rehearsal of N > 1 on a box with fewer GPUs than ranks (never used by the driver): GAML_BENCH_SHARE_GPU=1 puts every
    # rank on cuda:0; RCCL refuses two ranks on one device, so the collectives run over gloo / shared memory there.
    share_gpu = os.environ.get("GAML_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        with stdout_to_stderr():  # (a process group with a device id makes its communicator -- and prints RCCL's banner -- here)
            if share_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    wl = synth.WORKLOADS[args.workload]
    strong = args.scaling == "strong"
    genome, g = wl.build()
    # strong: every rank draws the SAME read set (the N = 1 seed) and keeps its contiguous share; weak: its own set
    read_seed = wl.seed if (strong or world == 1) else wl.seed + 1000 * rank
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, read_seed)
    gb, go = g.packed()
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    variants_py = path_variants(walk)
    cfg = (wl.insert_mean, wl.insert_std)

    if strong and world > 1:
        ctx = api.Context(device=local_rank, rank=rank, world=world)  # gaml_hip_set_shard: reads [N r/w, N (r+1)/w)
    else:
        ctx = api.Context(device=local_rank, presharded=world)  # every rank hands over only its own reads
    ctx.set_graph(gb, go)
    rs = ctx.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
    total_pairs = wl.n_pairs if (strong or world == 1) else wl.n_pairs * world

    # ---- the exchange of the N > 1 path
    scorer, exchange, fallback, comm_setup_s = None, "none", None, None
    if use_dist:
        want = os.environ.get("GAML_BENCH_EXCHANGE", "shm" if share_gpu else "rccl")
        if want == "rccl":
            # the library's own communicator: rank 0 makes the id, torch.distributed carries the 128 bytes
            watchdog = None
            try:
                # a stuck ncclCommInitRank (a peer that never arrives, a fabric problem) must end the run, not hang the
                # driver: the whole setup is on a clock, and a process that misses it exits non-zero
                import threading
                limit = float(os.environ.get("GAML_BENCH_COMM_SETUP_TIMEOUT_S", "180"))
                def _give_up():
                    sys.stderr.write(f"bench.py rank {rank}: communicator setup did not finish within {limit:.0f} s -- giving up\n")
                    sys.stderr.flush()
                    os._exit(3)
                watchdog = threading.Timer(limit, _give_up)
                watchdog.daemon = True
                watchdog.start()
                t_comm = time.time()
                idt = torch.zeros(128, dtype=torch.uint8)
                if rank == 0:
                    idt = torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8).clone()
                if not share_gpu:
                    idt = idt.cuda()
                with stdout_to_stderr():
                    dist.broadcast(idt, src=0)
                    ctx.comm_init_rank(bytes(idt.cpu().numpy().tobytes()), rank, world)
                    torch.cuda.synchronize()
                want_sub = os.environ.get("GAML_BENCH_RCCL_FORM", "gather")  # gather (rank-order sum, default) | allreduce
                if want_sub == "allreduce":
                    ctx.set_exchange("rccl-allreduce")
                watchdog.cancel()
                comm_setup_s = time.time() - t_comm
                exchange = "rccl"
            except Exception as e:  # keep the run alive on the well-trodden path, and say so
                fallback = repr(e)
                want = "rccl-torch"
            finally:
                if watchdog is not None:
                    watchdog.cancel()  # (also when setup RAISED: the fallback run must not be killed 180 s later with the setup's message)
        if want in ("rccl-torch", "shm", "gloo"):
            from gaml_amd.dist import ShardedScorer
            shm = want == "shm"
            scorer = ShardedScorer(ctx, host_exchange=("/gaml_bench_%s" % os.environ.get("MASTER_PORT", "0")) if shm else None)
            exchange = "shm" if shm else ("gloo" if share_gpu else "rccl-torch")
    torch.cuda.synchronize()

    in_loop = [False]

    def step(fp):
        if scorer is not None:
            return scorer.score(fp) if in_loop[0] else scorer.calc_prob(fp)[0]
        # ONE ABI call, like the reference's CalcProb(paths): registration, kernels, (N > 1: the RCCL all-reduce on the
        # library's stream,) the value back on the host (blocking)
        return ctx.score(fp)

    # prime: align every window any variant needs (cold path, untimed), then warm-up steps
    t0 = time.time()
    variants = [api.FlatPaths(v) for v in variants_py]  # the ABI's form, built once (as a C++ caller holds it)
    vals = [step(v) for v in variants]
    ctx.compact_tables()  # steady state: fold what the priming calls aligned into the device tables (the library would after 64 quiet calls)
    step(variants[0])     # ... which happens at the next evaluation: keep it out of the warm-up / timed steps
    for v in variants:    # ... and every path set once more: the first use of a memoised path after a rebuild re-marks its
        step(v)           # windows as in use (the rebuild's bookkeeping, 50-60 us once per path set) -- with --warmup < 8
    prime_s = time.time() - t0  # those calls would otherwise land in the timed steps
    # (the collector runs BEFORE the warm-up steps: a millisecond of host work between the last warm-up step and the
    # first timed one lets the GPU drop its clocks, and a short run -- the driver's --steps 20 -- pays for that step)
    gc.collect()
    gc.disable()
    for i in range(args.warmup):
        step(variants[i % len(variants)])

    # ---- the timed region: exactly --steps steps, no event timing, no garbage collector
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    loop_ctx = torch.cuda.stream(scorer.stream) if (scorer is not None and not scorer._host_exchange) else contextlib.nullcontext()
    stamps = np.zeros(args.steps + 1)
    last = None
    t0 = time.perf_counter()
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
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- separate pass: HIP events attached to EVERY scoring dispatch (not part of the throughput number)
    ks = {"launches": 0, "device_us": 0.0, "algo_bytes": 0.0}
    if not args.no_extras:
        ctx.set_event_timing(1)
        ctx.kernel_stats(reset=True)
        with loop_ctx:
            in_loop[0] = True
            for i in range(args.kernel_samples):
                step(variants[i % len(variants)])
            in_loop[0] = False
        torch.cuda.synchronize()
        ks = ctx.kernel_stats(reset=True)
        ctx.set_event_timing(False)

    # ---- N > 1: the sharded value against the unsharded one (rank 0 scores the whole read set on its GPU)
    ll_delta_n1 = None
    if use_dist and (strong or world == 1) and not args.no_extras:
        sharded = [step(v) for v in variants]
        if rank == 0:
            c1 = api.Context(device=local_rank)
            c1.set_graph(gb, go)
            c1.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
            whole = [c1.score(v) for v in variants]
            ll_delta_n1 = max(abs(a - b) / abs(b) for a, b in zip(sharded, whole))
            c1.close()
        dist.barrier()

    # ---- N > 1: the same read set through ONE process that owns all N devices (what gaml.cc + the adapter header runs)
    inproc = None
    if world > 1 and rank == 0 and not args.no_inproc and not share_gpu:
        try:
            cmd = [sys.executable, os.path.abspath(__file__), "--inproc-devices", ",".join(str(i) for i in range(world)),
                   "--workload", args.workload, "--steps", str(min(args.steps, 1000)), "--warmup", str(min(args.warmup, 50))]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                                    "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("INPROC ")]
            inproc = json.loads(line[-1][7:]) if line else {"error": (r.stderr or r.stdout)[-400:]}
        except Exception as e:
            inproc = {"error": repr(e)}
    if world > 1:
        dist.barrier()

    # ---- N > 1 over the library's communicator: what SCALES -- work per call. The same 8 path sets through
    # gaml_hip_calc_prob_batch (every rank calls it: ONE exchange per batch), and candidate batches of one assembly
    batched_dist, cands_dist = None, None
    if use_dist and exchange == "rccl" and not args.no_extras:
        bp = api.BatchPaths(variants_py)
        bvals = [b[0] for b in ctx.calc_prob_batch(bp)]
        calls = max(4, args.steps // (8 * 4))
        dist.barrier()
        gc.disable()
        tb = time.perf_counter()
        for _ in range(calls):
            ctx.calc_prob_batch(bp)
        dist.barrier()
        tb = time.perf_counter() - tb
        gc.enable()
        batched_dist = {"api": "gaml_hip_calc_prob_batch over the communicator (one exchange per batch)", "sets_per_call": len(variants_py), "calls": calls,
                        "ms_per_set": 1e3 * tb / (calls * len(variants_py)), "reads_per_sec": 2.0 * total_pairs * calls * len(variants_py) / tb,
                        "ll_rel_delta_vs_single_calls": max(abs(a - b) / abs(b) for a, b in zip(bvals, vals))}
        if not args.no_sa:
            cands_dist = batched_candidates(ctx, g, api, synth)  # (seeded: every rank builds the same batches)

    # ---- N > 1, strong scaling is the headline (BASELINE config 3 as stated: ONE 50x read set split over the GPUs): the same
    # step with the per-GPU work FIXED next to it -- every rank holds a 50x read set of its own (N x 833,333 pairs in all),
    # one exchange per step. This is the quantity that can grow ~N (see `expected_to_scale`).
    weak_block = None
    if (world > 1 or args.force_dist) and strong and use_dist and not args.no_extras:  # (--force-dist: the code path with one rank, rehearsal)
        try:
            weak_block = weak_scaling_pass(args, api, synth, dist, torch, wl, g, gb, go, cfg, variants_py, rank, world, local_rank, share_gpu, exchange)
        except Exception as e:  # never at the expense of the headline line
            weak_block = {"error": repr(e)}
        dist.barrier()

    if rank == 0:
        total_reads = 2.0 * total_pairs
        ms_per_step = 1e3 * elapsed / args.steps
        value = total_reads * args.steps / elapsed
        launches = max(1, ks["launches"])
        kern_us = ks["device_us"] / launches
        bytes_per_launch = ks["algo_bytes"] / launches
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9 if kern_us > 0 else 0.0
        # HBM traffic per launch comes from rocprofv3 --pmc passes of this same command (counters cannot be read from
        # inside the process): only a profile taken on THESE sources counts, otherwise null + traffic_stale
        traffic, traffic_src, traffic_stale = None, None, None
        try:
            import glob
            # (the newest summary taken for THIS workload: <tag>_pmc_traffic.json is cfg3's, <tag>_cfg3x8_pmc_traffic.json the HBM-bound size's)
            cands = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json")))
                     if json.load(open(f)).get("workload", "cfg3") == args.workload]
            f = cands[-1]
            tj = json.load(open(f))
            traffic_src = os.path.relpath(f, ROOT)
            traffic_stale = tj.get("source_hash") != source_hash()
            if world == 1 and not traffic_stale:
                traffic = tj["kernels"]["paired_score_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        d = np.diff(stamps) * 1e6
        out = {
            "metric": "reads_per_sec_scored", "value": value, "unit": "reads/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if (strong or world == 1) else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl.name + f"; {total_pairs} pairs in all, whole true genome as 1-2 walks, 8 rotating path sets, "
                                             "window cache warm", "pairs_total": total_pairs, "pairs_per_gpu": total_pairs // world,
                       "genome_bp": wl.genome_len,
                       "parallelism": f"reads sharded over {world} GPU(s)" + ("" if not use_dist else
                                      {"rccl": ", one in-library RCCL all-reduce(sum) of 32 B per step on the scoring stream",
                                       "rccl-torch": ", one torch.distributed (RCCL) all-reduce of 32 B per step",
                                       "shm": ", partials summed through POSIX shared memory (one node; blocking evaluation per rank)",
                                       "gloo": ", gloo all-reduce (rehearsal: ranks share one GPU)"}[exchange])},
            "exchange": exchange, "rccl_ranks": world if exchange in ("rccl", "rccl-torch") else 0,
            "exchange_form": (ctx.exchange() if exchange == "rccl" else None), "comm_setup_s": comm_setup_s,
            "ll_rel_delta_vs_n1": ll_delta_n1,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "paired_score_kernel", "kernel_us": kern_us, "algo_bytes_per_launch": bytes_per_launch,
                         "timing": "HIP events attached to the dispatch (hipExtLaunchKernelGGL, on the launch stream) of EVERY "
                                   "paired_score_kernel launch of a separate pass after the timed region: the kernel's own begin/end "
                                   "stamps, as in rocprofv3's kernel trace",
                         "timed_launches": int(launches), "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "source_hash": source_hash()},
            "step_us_distribution": {"p50": float(np.percentile(d, 50)), "p90": float(np.percentile(d, 90)),
                                     "p99": float(np.percentile(d, 99)), "max": float(d.max())},
            "log_likelihood": last, "prime_s": prime_s,
            "pair_classes_le1_le2_le4_more": [int(x) for x in ctx.pair_classes(rs)],
            "timing_last_step_us": ctx.last_timing(),
        }
        if use_dist:
            out["expected_to_scale"] = (
                "value (blocking single calls, strong scaling) is NOT expected to grow with N at this size: a step is 30-40 us of which "
                "~12 us of host planning are repeated on every rank and ~8 us are launch / completion latency; sharding shortens only "
                "the ~5 us of a wave's memory chain and adds the exchange. What scales: batched.reads_per_sec (one exchange per "
                "batch of 8 path sets) and weak_scaling.value in this line (per-GPU work fixed, every rank a 50x read set of its own: should grow ~N; "
                "--scaling weak makes it the headline)" if strong else
                "weak scaling: per-GPU work is fixed, value should grow ~N (minus the exchange's latency per step)")
        if use_dist and batched_dist is None and not args.no_extras:
            out["batched"] = {"note": "needs the in-library communicator (exchange rccl): one exchange per batch"}
        if weak_block is not None:
            out["weak_scaling"] = weak_block
        if batched_dist is not None:
            out["batched"] = batched_dist
        if cands_dist is not None:
            out["batched_candidates"] = cands_dist
        if fallback:
            out["exchange_fallback_reason"] = fallback
        if inproc is not None:
            out["inproc_multi"] = inproc
        if not use_dist and not args.no_extras:
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
            if not args.no_sa:
                # BASELINE config 5's call pattern in a context of its own, as a GAML run has it: ProbCalculator made, start
                # assembly scored, annealing. (In the headline's context -- tables built over the 8 rotating path sets -- the start
                # state's ~1000 windows arrive as delta pairs, a worker rebuild is under way from the first call on, and the
                # walk measures that bookkeeping as well: p90 93-106 us there against 85-93 here; `warm_context` has it.)
                ctx_sa = api.Context(device=local_rank)
                ctx_sa.set_graph(gb, go)
                rs_sa = ctx_sa.add_paired(api.paired_cfg(*cfg), b1, o1, b2, o2)
                out["sa_pattern"] = sa_pattern(ctx_sa, rs_sa, g, args.sa_iters, api, synth)
                out["sa_pattern"]["context"] = "fresh (its own; first call = the start assembly, cold: window alignment + table build)"
                ctx_sa.close()
                warm = sa_pattern(ctx, rs, g, args.sa_iters, api, synth)
                out["sa_pattern"]["warm_context"] = {k: warm[k] for k in ("total_s", "us_median", "us_p90", "us_p99", "us_max", "aligning_call_us_median", "other_calls_us_p99")}
                out["batched_candidates"] = batched_candidates(ctx, g, api, synth)
        if not use_dist and not args.no_extras and not args.no_long and args.workload == "cfg3":
            out["aligner"] = aligner_block(api, synth, local_rank, g, gb, go, b1, o1, b2, o2, cfg, wl.read_len)
            out["sa_long"] = sa_long_block(api, synth, local_rank, g, gb, go, b1, o1, b2, o2, cfg, wl.read_len, iters=args.sa_long_iters, oracle=not args.no_cpu_baseline)
        if not use_dist and not args.no_extras and not args.no_repeats and args.workload == "cfg3":
            # (with --no-cpu-baseline -- the profiling runs -- the GPU sides of these blocks still run, without their oracle legs)
            ctx.close()  # (the headline context's tables: ~100 MB of device memory back before two more read sets are built)
            out["repeats"] = repeats_block(api, synth, local_rank, oracle=not args.no_cpu_baseline)
            if not args.no_long:
                out["jumping"] = jumping_block(api, synth, local_rank, oracle=not args.no_cpu_baseline)
            if not args.no_cpu_baseline:
                out["incremental_drift"] = drift_block(api, synth, local_rank, g, b1, o1, b2, o2, wl.read_len, cfg)
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is taken on rank 0 at N = 1 only
            pairs = min(args.cpu_sample_pairs or wl.n_pairs, wl.n_pairs)
            cb, cpu_vals = cpu_baseline(gb, go, b1, o1, b2, o2, pairs, wl.read_len, variants_py, cfg)
            out["cpu_baseline"] = cb
            # log-likelihood delta vs the CPU restatement of the reference on the same reads, every path set
            if pairs == wl.n_pairs:
                gpu_vals = vals
            else:
                c2 = api.Context(device=local_rank)
                c2.set_graph(gb, go)
                nb = pairs * wl.read_len
                c2.add_paired(api.paired_cfg(*cfg), b1[:nb], o1[:pairs + 1], b2[:nb], o2[:pairs + 1])
                gpu_vals = [c2.calc_prob(v)[0] for v in variants]
            out["ll_max_rel_delta_vs_cpu"] = max(abs(a - b) / abs(b) for a, b in zip(gpu_vals, cpu_vals))
            out["ll_delta_pairs"] = pairs
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
            if not args.no_cpu_sharded and not args.no_extras:
                out["cpu_baseline_all_cores"] = cpu_baseline_sharded(args.workload, wl.n_pairs)
        print(json.dumps(out), flush=True)
    if use_dist:
        if scorer is not None:
            scorer.close()  # unlinks the shared-memory name (rank 0)
        ctx.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
