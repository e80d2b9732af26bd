"""A/B of the record tables with and without the records that are always overwritten (knob 16, host_model.cc
dominated_records): one cfg3 context, tables rebuilt between the legs, interleaved rounds."""
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
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
ref = [ctx.calc_prob(v)[0] for v in variants]
ctx.set_event_timing(True)
out = {0: [], 1: []}
for rnd in range(4):
    for keep in (1, 0):
        ctx.debug_set_knob(16, keep)
        ctx.compact_tables()
        for v in variants:
            ctx.calc_prob(v)
        ctx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for i in range(160):
            v = ctx.calc_prob(variants[i % 8])[0]
            assert abs(v - ref[i % 8]) <= 1e-12 * abs(v), (keep, v, ref[i % 8])
        wall = (time.perf_counter() - t0) / 160 * 1e6
        ks = ctx.kernel_stats(reset=True)
        out[keep].append((ks["device_us"] / max(1, ks["launches"]), wall))
        if rnd == 0:
            print("kept" if keep else "left out", "classes", ctx.debug_class_counts(rs), ctx.debug_table_stats(rs), flush=True)
for keep in (1, 0):
    a = np.array(out[keep])
    print("%-8s kernel us median %.2f min %.2f | step us median %.1f min %.1f" % ("kept" if keep else "left out", np.median(a[:, 0]), a[:, 0].min(),
                                                                                    np.median(a[:, 1]), a[:, 1].min()))
ctx.close()
