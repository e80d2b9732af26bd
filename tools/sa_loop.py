"""BASELINE config 5 rehearsal: the simulated-annealing call pattern at full size on one GPU.
The optimiser itself (move generators, gaml.cc:91-343) is out of scope; what is driven here is its
use of CalcProb: one evaluation per iteration on a slightly edited path set, new junction windows
appearing all the time. Prints per-iteration cost and where it goes; optionally the CPU oracle on a
read sample drives the same sequence (incremental ScoringState, like the reference)."""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from gaml_amd import synth, api
from test_gpu_sa_pattern import _moves

wl = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cpu_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 0
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
if os.environ.get("GAML_HIP_TRACE_ALIGNER"):
    ctx.debug_set_knob(9, 1)  # separate the extension kernel from its D2H copy in the stage timings
ctx.set_graph(*g.packed())
_args = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
_t = time.time()
rs = ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *_args)
print(f'add_paired (read index of both mates): {time.time() - _t:.2f} s', flush=True)
walk = synth.genome_walk(g)
start = [[x] for x in walk if g.node_len(x) > 500]  # gaml.cc:1002-1005
t0 = time.time(); v0 = ctx.calc_prob(start); t_first = time.time() - t0
print('after the first call:', ctx.aligner_stats(), flush=True)
rng = np.random.default_rng(7)
seq, cur = [], start
for it in range(iters):
    new = _moves(rng, cur, g)
    seq.append(new)
    if rng.random() < 0.6:
        cur = new
flat = [api.FlatPaths(p) for p in seq]
t0 = time.time(); vals = []
per = []
prof = []
for f in flat:
    t1 = time.perf_counter(); vals.append(ctx.calc_prob(f)[0]); per.append(time.perf_counter() - t1); prof.append(ctx.debug_profile())
t_gpu = time.time() - t0
per = np.array(per) * 1e3
print("median profile us [pass1, tables, ovf+occ8, pack, h2d_enq, launch, bytes, wait]:", np.round(np.median(np.array(prof), axis=0), 1))
print(f"first CalcProb (start state, cold): {t_first:.2f} s  value {v0[0]:.6f}")
print(f"{iters} SA-pattern CalcProb calls: {t_gpu:.2f} s total, per call median {np.median(per):.3f} ms, p90 {np.percentile(per, 90):.3f} ms, max {per.max():.1f} ms")
print("aligner:", ctx.aligner_stats(), "tables:", ctx.debug_table_stats(rs), "windows:", ctx.window_count(rs, 0), "paths now:", len(cur))
if cpu_pairs:
    import oracle_py as op
    orc = op.Oracle(); orc.set_graph(*g.packed())
    orc.add_paired(*synth.pack_reads(pr.mate1[:cpu_pairs]), *synth.pack_reads(pr.mate2[:cpu_pairs]), 0.01, op.paired_cfg(wl.insert_mean, wl.insert_std))
    t0 = time.time(); orc.calc_prob(start, fresh=False); t_c0 = time.time() - t0
    n_cpu = min(iters, 100)
    t0 = time.time()
    for p in seq[:n_cpu]:
        orc.calc_prob(p, fresh=False)  # incremental ScoringState, as the reference runs it
    t_cpu = time.time() - t0
    print(f"CPU oracle (1 core, incremental state, {cpu_pairs} of {wl.n_pairs} pairs): first {t_c0:.2f} s, {n_cpu} calls {t_cpu:.2f} s = {1e3 * t_cpu / n_cpu:.2f} ms per call "
          f"-> scaled to all pairs ~{1e3 * t_cpu / n_cpu * wl.n_pairs / cpu_pairs:.1f} ms per call")
