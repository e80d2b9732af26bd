"""A repeat-rich assembly (collapsed 5-copy repeat families: synth.make_repeat_graph) on the GPU: pair classes, the
scoring launch (its GEN instantiation: pairs on windows that occur several times), step time, and
-- ORACLE=1 -- the likelihood against the CPU oracle on all pairs.    python tools/repeats_probe.py [cfg3r|tinyr]
python tools/repeats_probe.py late [iterations]: the late state of a long synthetic annealing walk at cfg3 (duplicated nodes pile up)."""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
import bench

def late_walk(iters):
    wl = synth.WORKLOADS["cfg3"]
    genome, g = wl.build()
    pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
    ctx = api.Context(device=0)
    for kv in filter(None, os.environ.get("KNOBS", "").split(",")):
        ctx.debug_set_knob(*map(int, kv.split("=")))
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    start, seq = synth.sa_sequence(g, iters)
    flat = [api.FlatPaths(p) for p in seq]
    ctx.calc_prob(start)
    per = np.zeros(len(flat))
    for k, f in enumerate(flat):
        t1 = time.perf_counter(); ctx.score(f); per[k] = time.perf_counter() - t1
    ctx.set_event_timing(True)
    ctx.kernel_stats(reset=True)
    kus = []
    tot = {"launches": 0, "device_us": 0.0}
    for f in flat[-200:]:
        ctx.score(f); k1 = ctx.kernel_stats(reset=False)
        kus.append(k1["device_us"] - tot["device_us"]); tot = k1
    ks = ctx.kernel_stats()
    print("   scoring launch us over the last 200 calls: min %.1f p10 %.1f median %.1f p90 %.1f max %.1f" % (min(kus), np.percentile(kus, 10), np.median(kus), np.percentile(kus, 90), max(kus)))
    print(f"late walk, {iters} iterations: last 1000 calls median {np.median(per[-1000:]) * 1e6:.1f} us; scoring launch {ks['device_us'] / max(1, ks['launches']):.2f} us; "
          f"classes {list(map(int, ctx.debug_class_counts(rs)))}; {ctx.debug_table_stats(rs)}", flush=True)
    ctx.close()

if len(sys.argv) > 1 and sys.argv[1] == "late":
    late_walk(int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
    sys.exit(0)
wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3r"]
genome, g = wl.build()
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
gb, go = g.packed()
r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
ctx = api.Context(device=0)
for kv in filter(None, os.environ.get("KNOBS", "").split(",")):  # e.g. KNOBS=10=128
    ctx.debug_set_knob(*map(int, kv.split("=")))
ctx.set_graph(gb, go)
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *r1, *r2)
walk = synth.genome_walk(g)
variants_py = bench.path_variants(walk)
variants = [api.FlatPaths(v) for v in variants_py]
t0 = time.time()
vals = [ctx.score(v) for v in variants]
print(f"{wl.name}: walk of {len(walk)} nodes, {len(set(walk))} distinct; cold pass {time.time() - t0:.2f} s", flush=True)
ctx.compact_tables()
vals = [ctx.score(v) for v in variants]
vals = [ctx.score(v) for v in variants]
print("classes", list(ctx.debug_class_counts(rs)), ctx.debug_table_stats(rs), flush=True)
ctx.set_event_timing(True)
for i in range(64):
    ctx.score(variants[i % 8])
ctx.kernel_stats(reset=True)
t = time.perf_counter()
n = 400
for i in range(n):
    ctx.score(variants[i % 8])
dt = (time.perf_counter() - t) / n * 1e6
ks = ctx.kernel_stats()
print(f"scoring launch {ks['device_us'] / max(1, ks['launches']):.2f} us, "
      f"step {dt:.2f} us, algorithmic {ks['algo_bytes'] / max(1, ks['launches']) / 1e6:.2f} MB", flush=True)
ctx.set_event_timing(False)
prof = []
for i in range(200):
    ctx.score(variants[i % 8])
    prof.append(ctx.debug_profile())
ph = np.median(np.array(prof), axis=0)
print("phases us: planning %.1f, tables %.1f, align %.1f, write %.1f, sync %.1f, launch %.1f, bytes %.0f, wait %.1f" % tuple(ph), flush=True)
if os.environ.get("ORACLE"):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle_py as op
    orc = op.Oracle()
    orc.set_graph(gb, go)
    orc.add_paired(*r1, *r2, wl.err, op.paired_cfg(wl.insert_mean, wl.insert_std))
    k = int(os.environ.get("ORACLE_SETS", "2"))
    t0 = time.time()
    want = [orc.calc_prob(v, fresh=True)[0] for v in variants_py[:k]]
    print(f"oracle: {k} path sets in {time.time() - t0:.1f} s; max rel delta {max(abs(a - b) / abs(b) for a, b in zip(vals, want)):.2e}", flush=True)
ctx.close()
