"""Tail of the annealing pattern at cfg3: per-call times with / without the quiet-spell table fold (knob 6 = 2), where
the slow calls come from, aligner stage breakdown.  python tools/sa_tail.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, int(os.environ.get("SA_ITERS", "1000")))
flat = [api.FlatPaths(p) for p in seq]
for knob6 in (0,):
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *reads)
    ctx.debug_set_knob(6, knob6)
    ctx.debug_set_knob(9, 1)
    if os.environ.get("SA_WARM"):  # as bench.py: the context has scored the 8 rotating path sets of the headline (tables built over THEIR windows) first
        import bench
        variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
        for v in variants: ctx.score(v)
        ctx.compact_tables(); ctx.score(variants[0])
        for i in range(200): ctx.score(variants[i % 8])
    ctx.calc_prob(start)
    os.environ["GAML_HIP_TRACE_ALIGNER"] = "1"; print("   after the cold call:", end=" ", flush=True); ctx.aligner_stats(); del os.environ["GAML_HIP_TRACE_ALIGNER"]
    per, prof, kinds = [], [], []
    a_prev = ctx.aligner_stats()["windows"]
    for f in flat:
        t = time.perf_counter(); ctx.score(f); per.append((time.perf_counter() - t) * 1e6)
        prof.append(ctx.debug_profile())
        a = ctx.aligner_stats()["windows"]; kinds.append(a - a_prev); a_prev = a
    per = np.array(per); kinds = np.array(kinds); prof = np.array(prof)
    al = kinds > 0
    print(f"knob6={knob6}: total {per.sum() / 1e3:.1f} ms, median {np.median(per):.1f}, p90 {np.percentile(per, 90):.1f}, p99 {np.percentile(per, 99):.1f}, max {per.max():.0f} us; "
          f"calls that aligned windows: {al.sum()} (median {np.median(per[al]):.0f} us, p90 {np.percentile(per[al], 90):.0f}); others median {np.median(per[~al]):.1f} p99 {np.percentile(per[~al], 99):.1f}")
    print("   aligning calls, median phases [pass1, tables_host, align (in pass1), write, sync(delta), launch, bytes, wait]:", np.round(np.median(prof[al], axis=0), 1))
    print("   other calls,    median phases:", np.round(np.median(prof[~al], axis=0), 1))
    order = np.argsort(-per)[:8]
    for k in order: print(f"   slow call {k}: {per[k]:.0f} us, windows aligned {kinds[k]}, phases", np.round(prof[k], 1))
    print("   last 200 calls median", np.median(per[-200:]), ctx.debug_table_stats(rs))
    os.environ["GAML_HIP_TRACE_ALIGNER"] = "1"; print("   at the end:", end=" ", flush=True); ctx.aligner_stats(); del os.environ["GAML_HIP_TRACE_ALIGNER"]
    ctx.close()

# calls that aligned nothing and still took long: what they did (delta-list moves of windows aligned earlier)
if os.environ.get("SA_TAIL_OTHERS"):
    slow = [k for k in np.argsort(-per) if not al[k] and per[k] > 60.0]
    print(f"   calls without alignment above 60 us: {len(slow)}; above 90 us: {sum(per[k] > 90 for k in slow)}; above 130 us: {sum(per[k] > 130 for k in slow)}")
    for k in slow[::max(1, len(slow) // 16)]:
        print(f"      call {k}: {per[k]:.0f} us, phases", np.round(prof[k], 1))
