import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, "/root/repo")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist
from gaml_amd import synth, api
from gaml_amd.dist import ShardedScorer
import bench
dist.init_process_group("nccl", rank=0, world_size=1)
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0); ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
sc = ShardedScorer(ctx)
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
with torch.cuda.stream(sc.stream):
    for v in variants: sc.score(v)
    ctx.compact_tables()
    for v in variants: sc.score(v)
    def old(fp):
        tl = sc._enqueue(fp, sc.d_part); sc._all_reduce(sc.d_part, dist.ReduceOp.SUM)
        sc.h_part.copy_(sc.d_part, non_blocking=True); sc.stream.synchronize()
        return sc.ctx.combine_fast(sc._h_ptr, tl)
    for rnd in range(3):
        for name, f in (("fetch", sc.score), ("copy+sync", old)):
            t0 = time.perf_counter()
            for i in range(1000): f(variants[i % 8])
            print(name, "%.1f us/step" % ((time.perf_counter() - t0) / 1000 * 1e6))
dist.destroy_process_group()
