"""Long annealing-pattern run at cfg3 size: stability of the per-call time and of the process memory over many
iterations (planner memos, window cache, delta store, table rebuilds).  python tools/sa_long.py [iterations]"""
import os, sys, time, resource
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gaml_amd import synth, api
from test_gpu_sa_pattern import _moves
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
del pr
walk = synth.genome_walk(g)
cur = [[x] for x in walk if g.node_len(x) > 500]
ctx.calc_prob(cur)
rng = np.random.default_rng(7)
t0 = time.time(); per = []; prof = []
for it in range(iters):
    new = _moves(rng, cur, g)
    fp = api.FlatPaths(new)
    t = time.perf_counter(); v = ctx.score(fp); per.append(time.perf_counter() - t); prof.append(ctx.debug_profile())
    assert np.isfinite(v)
    if rng.random() < 0.6:
        cur = new
    if (it + 1) % 2500 == 0:
        p = np.array(per[-2500:]) * 1e6
        print(f"iteration {it + 1}: last 2500 calls median {np.median(p):.0f} us, p90 {np.percentile(p, 90):.0f} us, total {p.sum() / 1e6:.2f} s; "
              f"paths {len(cur)}, windows {ctx.window_count(rs, 0)}, tables {ctx.debug_table_stats(rs)}, "
              f"max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6:.2f} GB; "
              f"median profile [pass1, tables, ovf, pack, h2d, launch, bytes, wait] {np.round(np.median(np.array(prof[-2500:]), axis=0), 0).tolist()}", flush=True)
print(f"{iters} calls in {time.time() - t0:.1f} s (incl. move generation in Python)")
