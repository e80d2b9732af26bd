"""Where the time of one sharded (N > 1 code path) step goes, single rank over RCCL."""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from gaml_amd import synth, api
from gaml_amd.dist import ShardedScorer
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0, presharded=2)
ctx.set_graph(*g.packed())
ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
sc = ShardedScorer(ctx)
[sc.calc_prob(v) for v in variants]; ctx.compact_tables(); [sc.calc_prob(v) for v in variants]
seg = np.zeros(5); n = 0
import gc; gc.disable()
with torch.cuda.stream(sc.stream):
    sp = sc.stream.cuda_stream
    for i in range(600):
        v = variants[i % 8]
        t0 = time.perf_counter(); pending, tl = ctx.eval_begin(v)
        t1 = time.perf_counter(); ctx.eval_score_async(sc.d_part.data_ptr(), sp)
        t2 = time.perf_counter(); dist.all_reduce(sc.d_part, op=dist.ReduceOp.SUM)
        t3 = time.perf_counter(); sc.ctx.fetch_async(sc._d_ptr, sc._n_part, sc.stream.cuda_stream)
        t4 = time.perf_counter(); sc.ctx.fetch_wait(sc._h_ptr, sc._n_part)
        t5 = time.perf_counter()
        if i >= 100:
            seg += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]; n += 1
print("us per step: eval_begin %.1f, eval_score_async %.1f, all_reduce enqueue %.1f, fetch enqueue %.1f, fetch wait %.1f, total %.1f" % (*(seg / n * 1e6), seg.sum() / n * 1e6))
dist.destroy_process_group()
