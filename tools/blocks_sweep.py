"""cfg3, warm steps: launch duration and step time over the number of compact-class blocks (knob 0) and <=2-record
class blocks (knob 10).  python tools/blocks_sweep.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
walk = synth.genome_walk(g)
cut = (len(walk) // 2) & ~1
variants = [api.FlatPaths([list(walk)]), api.FlatPaths([list(walk[:cut]), list(walk[cut:])])]
for v in variants: ctx.score(v)
ctx.compact_tables()
paths = variants[0]
ctx.score(paths)
print("class counts", ctx.debug_class_counts(rs))
ctx.set_event_timing(True)
for k0, k10 in eval(os.environ.get('SWEEP', '((0, 0), (0, 128), (0, 160), (0, 256), (776, 0), (850, 0), (0, 0))')):
    ctx.debug_set_knob(0, k0 % 10000); ctx.debug_set_knob(10, k10); ctx.debug_set_knob(17, k0 // 10000)  # k0 = order * 10000 + blocks
    for _ in range(50): ctx.score(paths)
    ctx.kernel_stats(reset=True)
    t = time.perf_counter()
    for _ in range(400): ctx.score(paths)
    dt = (time.perf_counter() - t) / 400 * 1e6
    st = ctx.kernel_stats()
    print(f"knob0={k0:5d} knob10={k10:4d}: launch {st['device_us'] / max(1, st['launches']):6.2f} us, step {dt:6.2f} us", flush=True)
ctx.close()
