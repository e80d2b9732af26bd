"""Where one blocking CalcProb goes (cfg3 by default), per upload route (knob 8: 0 = host writes device memory through the
BAR, 1 = staging + hipMemcpyAsync, 2 = staging + copy kernel): medians of gaml_hip_debug_profile over 400 steps.
  python tools/phase_profile.py [cfg3|cfg2] [batch]"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
import bench
wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
vp = bench.path_variants(synth.genome_walk(g))
variants = [api.FlatPaths(v) for v in vp]
[ctx.score(v) for v in variants]; ctx.compact_tables(); ctx.score(variants[0])
names = ["pass1", "tables_host", "-", "pack/write", "table_sync", "launch", "bytes", "wait"]
for knob in (0, 2, 1, 0):
    ctx.debug_set_knob(8, knob)
    for i in range(50): ctx.score(variants[i % 8])
    prof, ts = [], []
    for i in range(400):
        t = time.perf_counter(); ctx.score(variants[i % 8]); ts.append((time.perf_counter() - t) * 1e6); prof.append(ctx.debug_profile())
    med = np.median(np.array(prof), axis=0)
    print(f"knob8={knob}: step median {np.median(ts):.1f} us p90 {np.percentile(ts, 90):.1f} | " + ", ".join(f"{n} {v:.1f}" for n, v in zip(names, med) if n != "-"))
if len(sys.argv) > 2:
    bp = api.BatchPaths(vp)
    for knob11 in (0, 1, 0):
        ctx.debug_set_knob(11, knob11)
        for _ in range(5): ctx.calc_prob_batch(bp)
        t = time.perf_counter()
        for _ in range(50): ctx.calc_prob_batch(bp)
        dt = (time.perf_counter() - t) / 50
        print(f"batch of 8, knob11={knob11} ({'one pass' if knob11 == 0 else 'sequential'}): {dt * 1e6:.1f} us per call, {dt * 1e6 / 8:.1f} us per set")
    ctx.set_event_timing(1); ctx.kernel_stats(reset=True)
    ctx.debug_set_knob(11, 0)
    for _ in range(20): ctx.calc_prob_batch(bp)
    print("multi kernel:", ctx.kernel_stats(reset=True))
