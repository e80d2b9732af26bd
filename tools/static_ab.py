"""cfg3, the bench's 8 rotating path sets: launch duration (events attached to the dispatch) and step time for a list
of knob settings, e.g. static memo indices on / off (knob 19; takes effect at a table build, so the tables are rebuilt
after every change), blocks of the compact class's two parts (knobs 0 / 20), of the two-record class (knob 10).
  python tools/static_ab.py [workload]      SWEEP='[{}, {19: 1}, {20: 64}]'"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
import bench

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
genome, g = wl.build()
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
ref = [ctx.score(v) for v in variants]
sweep = eval(os.environ.get("SWEEP", "[{}, {19: 1}, {}]"))
cur = {}
ctx.set_event_timing(True)
for knobs in sweep:
    for k in set(cur) | set(knobs):
        ctx.debug_set_knob(k, knobs.get(k, 0))
    cur = dict(knobs)
    ctx.compact_tables()
    vals = [ctx.score(v) for v in variants]
    vals = [ctx.score(v) for v in variants]
    st = ctx.debug_table_stats(rs)
    for i in range(64):
        ctx.score(variants[i % 8])
    ctx.kernel_stats(reset=True)
    t = time.perf_counter()
    for i in range(800):
        ctx.score(variants[i % 8])
    dt = (time.perf_counter() - t) / 800 * 1e6
    ks = ctx.kernel_stats()
    rel = max(abs(a - b) / abs(b) for a, b in zip(vals, ref))
    print(f"knobs {str(knobs):28s} classes {list(ctx.debug_class_counts(rs))} static {st.get('static_index_pairs', -1):7d}: launch {ks['device_us'] / max(1, ks['launches']):6.2f} us, "
          f"step {dt:6.2f} us, LL delta vs first {rel:.1e}", flush=True)
ctx.close()
