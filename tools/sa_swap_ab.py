"""Annealing pattern over several table take-overs: total time, tail and the take-over calls for different take-over
delays (knob 14).  python tools/sa_swap_ab.py [iterations] [delays...]"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
delays = [int(a) for a in sys.argv[2:]] or [0, 768]
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, n_it)
flat = [api.FlatPaths(p) for p in seq]
for d in delays:
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *reads)
    if d > 0: ctx.debug_set_knob(14, d)
    if d < 0: ctx.debug_set_knob(18, -d)  # negative: the rebuild threshold as pairs / -d instead
    ctx.calc_prob(start)
    per, wr, at, prof = [], 0, [], []
    for k, f in enumerate(flat):
        t = time.perf_counter(); ctx.score(f); per.append((time.perf_counter() - t) * 1e6)
        prof.append(ctx.debug_profile())
        if k % 16 == 0:
            w = ctx.debug_table_stats(rs)["worker_rebuilds"]
            if w != wr: at.append(k); wr = w
    per = np.array(per)
    slow = np.argsort(-per)[:5]
    print(f"delay {d or 'default'}: total {per.sum() / 1e3:.1f} ms, median {np.median(per):.1f}, p90 {np.percentile(per, 90):.1f}, p99 {np.percentile(per, 99):.1f}, max {per.max():.0f} us; "
          f"take-overs seen by call {at}; slowest calls {[(int(k), int(per[k])) for k in slow]}; {ctx.debug_table_stats(rs)}", flush=True)
    ctx.set_event_timing(True); ctx.kernel_stats(reset=True)
    for f in flat[-200:]: ctx.score(f)
    print("   last 200 paths again, event timing:", ctx.kernel_stats(), "general kernel", ctx.debug_general_stats() if hasattr(ctx, "debug_general_stats") else None, "classes", ctx.debug_class_counts(rs), "general windows / multi:", [len(ctx.debug_table_occurrences(rs, m)[0]) for m in (0, 1)], flush=True)
    ctx.set_event_timing(False)
    prof = np.array(prof)
    for b in range(0, n_it, 1000):
        seg = slice(b, min(n_it, b + 1000))
        print(f"   calls {b}..: median {np.median(per[seg]):.1f} us; phases median [plan {np.median(prof[seg, 0]):.1f}, tables {np.median(prof[seg, 1]):.1f}, write {np.median(prof[seg, 3]):.2f}, "
              f"sync {np.median(prof[seg, 4]):.1f}, launch {np.median(prof[seg, 5]):.1f}, wait {np.median(prof[seg, 7]):.1f}], paths {len(seq[seg.stop - 1])}", flush=True)
    ctx.close()
