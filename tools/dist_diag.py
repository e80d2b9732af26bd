"""Which aligner route gives two processes sharing one GPU the plain context's floored-read counts?
  python tools/dist_diag.py            (GPU box; prints one line per knob-5 value: 0 = default, 4 = per-mate small, 3 = general)
Per rank also: the number of records per window, against the plain context's records restricted to that rank's reads."""
import json
import os
import socket
import sys
import tempfile

os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, out_dir, knob5, stagger):
    import time
    import torch
    import torch.distributed as dist
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
        ctx = api.Context(device=0, rank=rank, world=world)
        if knob5:
            ctx.debug_set_knob(5, knob5)
        ctx.set_graph(*g.packed())
        ctx.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
        if stagger:  # as tests/test_gpu_dist.py: a PacBio read set ingested before the first evaluation
            from test_gpu_dist import _long_reads
            ps, rb, ro = _long_reads(g, walk)
            pb = ctx.add_pacbio_reads(api.single_cfg(penalty_constant=0.0, penalty_step=30.0, min_prob_per_base=-1.0, weight=0.5, mismatch_prob=0.15), rb, ro, ps.names)
            ctx.pacbio_ingest_sam(pb, walk, ps.sam)
        scorer = ShardedScorer(ctx)
        sets = [[walk], [walk[:6], walk[6:]], [walk[3:]], [walk[:6], [x ^ 1 for x in reversed(walk[6:])]]]
        single = [scorer.calc_prob(p) for p in sets]
        wins = {}
        for mate in (0, 1):
            for w in range(ctx.window_count(0, mate)):
                key = list(ctx.debug_window_walk(0, mate, w))
                rec = ctx.window_records(0, mate, key)
                wins[f"{mate}:{key}"] = None if rec is None else [int(len(rec)), rec.tobytes().hex()]
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump({"single": [[v[0], v[1].tolist(), v[2]] for v in single], "wins": wins, "stats": ctx.aligner_stats()}, f)
    finally:
        dist.destroy_process_group()


def main():
    import numpy as np
    import torch.multiprocessing as mp
    from gaml_amd import api, synth
    genome = synth.make_genome(60_000, 73)
    g = synth.make_graph(genome, synth.cut_lengths(60_000, 73, long_rng=(900, 4000)))
    pr = synth.make_paired_reads(genome, 1501, 100, 250.0, 25.0, 0.01, 73)
    walk = synth.genome_walk(g)
    sets = [[walk], [walk[:6], walk[6:]], [walk[3:]], [walk[:6], [x ^ 1 for x in reversed(walk[6:])]]]
    plain = api.Context(device=0)
    plain.set_graph(*g.packed())
    plain.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    want = [plain.calc_prob(p) for p in sets]
    print("plain zeros:", [w[1].tolist() for w in want], flush=True)
    pw = {}
    for mate in (0, 1):
        for w in range(plain.window_count(0, mate)):
            key = list(plain.debug_window_walk(0, mate, w))
            pw[f"{mate}:{key}"] = plain.window_records(0, mate, key)
    for knob5, pacbio in [(k, pb) for pb in (True, False) for k in [int(x) for x in os.environ.get("DIAG_KNOBS", "0,4,3").split(",")] for _ in range(int(os.environ.get("DIAG_REPS", "2")))]:
        with tempfile.TemporaryDirectory() as d:
            mp.spawn(worker, args=(2, _free_port(), d, knob5, pacbio), nprocs=2, join=True)
            ranks = [json.load(open(os.path.join(d, f"rank{r}.json"))) for r in range(2)]
        print(f"knob5={knob5} pacbio={pacbio}: zeros", [v[1][0][0] for v in ranks[0]["single"]], "stats", ranks[0]["stats"], ranks[1]["stats"], flush=True)
        # per window: records of the two ranks together against the plain context's
        bad = 0
        for key, rec in pw.items():
            n_plain = 0 if rec is None else len(rec)
            got = [ranks[r]["wins"].get(key) for r in range(2)]
            n_got = sum(0 if x is None else x[0] for x in got)
            if n_got != n_plain:
                bad += 1
                if bad <= 6:
                    print(f"   window {key[:60]}: plain {n_plain} records, ranks {[None if x is None else x[0] for x in got]}")
        print(f"   windows whose record counts differ: {bad} of {len(pw)}", flush=True)


if __name__ == "__main__":
    main()
