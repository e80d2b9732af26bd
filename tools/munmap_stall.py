"""Does freeing a large host buffer (munmap) stall the process's GPU queues? A warm context scores the same path set in a loop;
between calls a numpy array of SIZE bytes is allocated, touched and freed.   python tools/munmap_stall.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS["tiny"] if "tiny" in synth.WORKLOADS else synth.WORKLOADS["cfg2"]
genome, g = wl.build()
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
fp = api.FlatPaths([synth.genome_walk(g)])
for _ in range(50): ctx.score(fp)
def timed(n=200, between=None):
    ts = []
    for _ in range(n):
        if between: between()
        t = time.perf_counter(); ctx.score(fp); ts.append((time.perf_counter() - t) * 1e6)
    return np.median(ts), np.percentile(ts, 90), max(ts)
print("plain loop: median %.1f p90 %.1f max %.1f us" % timed())
for size in (64 << 10, 1 << 20, 8 << 20, 64 << 20):
    def churn():
        a = np.empty(size, np.uint8); a[::4096] = 1; del a
    print("alloc + touch + free of %8d KB between calls: median %.1f p90 %.1f max %.1f us" % ((size >> 10,) + timed(100, churn)))
ctx.close()
