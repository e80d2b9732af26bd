"""Kernel tuning harness: one cfg3 context, interleaved rounds over knob settings (guide 5.4 rule 24)."""
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
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
[ctx.calc_prob(v) for v in variants]  # prime: activates every window
ctx.compact_tables()  # fold the delta lists into the device tables (device order settles)
ref = [ctx.calc_prob(v)[0] for v in variants]
settings = eval(sys.argv[2]) if len(sys.argv) > 2 else [{}]
ctx.set_event_timing(True)
res = {i: [] for i in range(len(settings))}
wall = {i: [] for i in range(len(settings))}
for rnd in range(6):
    for si, st in enumerate(settings):
        for k in range(12):
            ctx.debug_set_knob(k, st.get(k, 0))
        ctx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for i in range(40):
            v = ctx.calc_prob(variants[i % 8])[0]
            assert st.get(3, 0) or abs(v - ref[i % 8]) <= 1e-12 * abs(v), (st, v, ref[i % 8])
        wall[si].append((time.perf_counter() - t0) / 40 * 1e6)
        ks = ctx.kernel_stats(reset=True)
        res[si].append(ks["device_us"] / max(1, ks["launches"]))
prof = np.zeros(8)
for i in range(40):
    ctx.calc_prob(variants[i % 8]); prof += ctx.debug_profile()
print('profile us [pass1, tables, ovf+occ8, pack, h2d_enq, launch, bytes, wait]:', np.round(prof / 40, 1))
for si, st in enumerate(settings):
    print(st, "kernel_us median %.2f min %.2f | step_us median %.1f min %.1f" % (np.median(res[si]), min(res[si]), np.median(wall[si]), min(wall[si])))
print("class counts [0: <=1 record/mate, 1: <=2, 2: <=4, 3: overflow]:", ctx.debug_class_counts(rs), "table stats:", ctx.debug_table_stats(rs))
