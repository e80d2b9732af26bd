"""Does it matter to the scoring launch whether the occurrence tables sit in fine-grained device memory (written by
the host through the BAR) or in ordinary device memory (staged copy)?  cfg3 warm steps, knob 8.  python tools/finegrained_probe.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
walk = synth.genome_walk(g)
paths = api.FlatPaths([list(walk)])
ctx.score(paths); ctx.compact_tables(); ctx.score(paths)
ctx.set_event_timing(True)
for k8, k13 in ((0, 0), (1, 0), (0, 1), (0, 0), (1, 0)):
    ctx.debug_set_knob(8, k8); ctx.debug_set_knob(13, k13)
    for _ in range(50): ctx.score(paths)
    ctx.kernel_stats(reset=True)
    t = time.perf_counter()
    for _ in range(400): ctx.score(paths)
    dt = (time.perf_counter() - t) / 400 * 1e6
    st = ctx.kernel_stats()
    print(f"knob8={k8} knob13={k13}: launch {st['device_us'] / max(1, st['launches']):6.2f} us, step {dt:6.2f} us", flush=True)
ctx.close()
