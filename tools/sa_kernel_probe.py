"""Annealing pattern at cfg3: how long is the scoring launch itself, call by call, and what do the tables look like?
  python tools/sa_kernel_probe.py [iterations]"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
start, seq = synth.sa_sequence(g, n_it)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
ctx.calc_prob(start)
print("after the start state: classes", ctx.debug_class_counts(rs), ctx.debug_table_stats(rs), flush=True)
flat = [api.FlatPaths(p) for p in seq]
ctx.set_event_timing(True)
for rnd in range(2):
    ks, wall, prof = [], [], []
    for f in flat:
        ctx.kernel_stats(reset=True)
        t = time.perf_counter(); ctx.score(f); wall.append((time.perf_counter() - t) * 1e6)
        st = ctx.kernel_stats(reset=True)
        ks.append(st["device_us"] / max(1, st["launches"]))
        prof.append(ctx.debug_profile())
    ks, wall, prof = np.array(ks), np.array(wall), np.array(prof)
    print(f"round {rnd}: kernel us median {np.median(ks):.2f} p90 {np.percentile(ks, 90):.2f}; call median {np.median(wall):.1f} (event timing on); "
          f"phases median [plan {np.median(prof[:, 0]):.1f}, tables {np.median(prof[:, 1]):.1f}, write {np.median(prof[:, 3]):.2f}, sync {np.median(prof[:, 4]):.1f}, "
          f"launch {np.median(prof[:, 5]):.1f}, wait {np.median(prof[:, 7]):.1f}]")
    print("   classes", ctx.debug_class_counts(rs), ctx.debug_table_stats(rs), flush=True)
    ctx.compact_tables()
ctx.close()
