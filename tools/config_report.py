"""All five BASELINE.json configurations on one MI355X, one JSON object per line (-> profiles/<round>_configs.jsonl).
Each line: what was run, GPU wall time per evaluation, reads/s, and the log-likelihood difference against the
CPU restatement of the reference (oracle/) on the same inputs (on a read sample where the oracle would take
minutes). Not the driver's benchmark (that is bench.py); a record of the other configs at their stated sizes.
  python tools/config_report.py [out.jsonl]"""
import json
import os
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gaml_amd import api, synth  # noqa: E402
import oracle_py as op  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else None
lines = []


def emit(d):
    print(json.dumps(d), flush=True)
    lines.append(d)


def timed(f, reps):
    f()
    t = time.perf_counter()
    for _ in range(reps):
        f()
    return (time.perf_counter() - t) / reps


# ---- config 1: 50 kbp graph, 10,000 x 100 bp single reads --------------------------------------
G, n, seed = 50_000, 10_000, 101
genome = synth.make_genome(G, seed)
g = synth.make_graph(genome, synth.cut_lengths(G, seed))
sr = synth.make_single_reads(genome, n, 100, 0.01, seed)
walk = synth.genome_walk(g)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
ctx.add_single(api.single_cfg(), *synth.pack_reads(sr))
fp = api.FlatPaths([walk])
v = ctx.score(fp)
dt = timed(lambda: ctx.score(fp), 200)
o = op.Oracle()
o.set_graph(*g.packed())
o.add_single(*synth.pack_reads(sr), 0.01, op.single_cfg())
want = o.calc_prob([walk])[0]
t = time.perf_counter(); o.calc_prob([walk]); t_cpu = time.perf_counter() - t
emit({"config": 1, "what": "single-end: 50 kbp, 10,000 x 100 bp; the whole genome as one walk", "gpu_us_per_eval": dt * 1e6,
      "gpu_reads_per_s": n / dt, "cpu_oracle_reads_per_s": n / t_cpu, "log_likelihood": v, "rel_delta_vs_oracle": abs(v - want) / abs(want)})

# ---- config 2: 1 Mbp, 100,000 pairs 2x150 --------------------------------------------------------
wl = synth.WORKLOADS["cfg2"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
walk = synth.genome_walk(g)
pargs = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *pargs)
fp = api.FlatPaths([walk])
t = time.perf_counter(); v = ctx.score(fp); t_cold = time.perf_counter() - t
ctx.compact_tables()
dt = timed(lambda: ctx.score(fp), 500)
o = op.Oracle()
o.set_graph(*g.packed())
o.add_paired(*pargs, 0.01, op.paired_cfg(wl.insert_mean, wl.insert_std))
t = time.perf_counter(); want = o.calc_prob([walk])[0]; t_cpu_cold = time.perf_counter() - t
t = time.perf_counter(); o.calc_prob([walk]); t_cpu = time.perf_counter() - t
emit({"config": 2, "what": "paired: 1 Mbp, 100,000 pairs 2x150, insert 300+-30; one walk", "gpu_us_per_eval": dt * 1e6,
      "gpu_reads_per_s": 2 * wl.n_pairs / dt, "gpu_cold_first_eval_s": t_cold, "cpu_oracle_reads_per_s": 2 * wl.n_pairs / t_cpu,
      "cpu_oracle_cold_s": t_cpu_cold, "log_likelihood": v, "rel_delta_vs_oracle": abs(v - want) / abs(want)})

# ---- config 4: config 2's pairs (weight 1) + 5 kbp PacBio reads (weight 0.5, mismatch 0.15) ---------
ps = synth.make_pacbio_sam(g, walk, 400, 5000, 17)
rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
ro = np.zeros(len(ps.reads) + 1, np.int64)
ro[1:] = np.cumsum([len(r) for r in ps.reads])
pb = ctx.add_pacbio_reads(api.single_cfg(weight=0.5, mismatch_prob=0.15, min_prob_per_base=-1.0), rb, ro, ps.names)
t = time.perf_counter(); filed = ctx.pacbio_ingest_sam(pb, walk, ps.sam); t_ing = time.perf_counter() - t
st = ctx.pacbio_dp_stats(pb)
v = ctx.score(fp)
dt = timed(lambda: ctx.score(fp), 300)
ob = o.add_pacbio_reads(rb, ro, ps.names, 0.15, op.single_cfg(weight=0.5, min_prob_per_base=-1.0))
t = time.perf_counter(); o.pacbio_ingest_sam(ob, walk, ps.sam); t_cpu_ing = time.perf_counter() - t
want = o.calc_prob([walk])[0]
emit({"config": 4, "what": "config 2's pairs (weight 1) + 400 PacBio reads of ~4 kbp (weight 0.5, mismatch_prob 0.15) from SAM text",
      "gpu_us_per_eval": dt * 1e6, "sam_records": int(st["records"]), "filed": int(filed), "gpu_ingest_ms": t_ing * 1e3,
      "gpu_dp_kernel_ms": st["kernel_ms"], "cpu_oracle_ingest_s": t_cpu_ing, "log_likelihood": v, "rel_delta_vs_oracle": abs(v - want) / abs(want)})

# ---- config 3: 5 Mbp, 833,333 pairs; LL against the oracle on the first 50,000 pairs ------------------
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
walk = synth.genome_walk(g)
pargs = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
t = time.perf_counter(); ctx.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *pargs); t_setup = time.perf_counter() - t
fp = api.FlatPaths([walk])
t = time.perf_counter(); v = ctx.score(fp); t_cold = time.perf_counter() - t
ctx.compact_tables()
dt = timed(lambda: ctx.score(fp), 1000)
ns = 50_000
sub = api.Context(device=0)
sub.set_graph(*g.packed())
sargs = (pargs[0][: ns * wl.read_len], pargs[1][: ns + 1], pargs[2][: ns * wl.read_len], pargs[3][: ns + 1])
sub.add_paired(api.paired_cfg(wl.insert_mean, wl.insert_std), *sargs)
vs = sub.score(fp)
o = op.Oracle()
o.set_graph(*g.packed())
o.add_paired(*sargs, 0.01, op.paired_cfg(wl.insert_mean, wl.insert_std))
want = o.calc_prob([walk])[0]
emit({"config": 3, "what": "paired: 5 Mbp, 833,333 pairs 2x150 on ONE GPU (the 8-GPU form is the driver's scaling run); one walk",
      "gpu_us_per_eval": dt * 1e6, "gpu_reads_per_s": 2 * wl.n_pairs / dt, "setup_s": t_setup, "gpu_cold_first_eval_s": t_cold,
      "log_likelihood": v, "oracle_sample_pairs": ns, "rel_delta_vs_oracle_on_sample": abs(vs - want) / abs(want)})

# ---- config 5: the annealing call pattern on config 3's context ------------------------------------
from test_gpu_sa_pattern import _moves  # noqa: E402
cur = [[x] for x in walk if g.node_len(x) > 500]
ctx.score(api.FlatPaths(cur))
rng = np.random.default_rng(7)
seq = []
for it in range(1000):
    new = _moves(rng, cur, g)
    seq.append(api.FlatPaths(new))
    if rng.random() < 0.6:
        cur = new
per = []
t0 = time.perf_counter()
for f in seq:
    t = time.perf_counter(); ctx.score(f); per.append(time.perf_counter() - t)
tot = time.perf_counter() - t0
per = np.array(per) * 1e6
# the oracle with the reference's incremental ScoringState on a sample of the reads, same sequence
o2 = op.Oracle()
o2.set_graph(*g.packed())
o2.add_paired(*sargs, 0.01, op.paired_cfg(wl.insert_mean, wl.insert_std))
o2.calc_prob([[x] for x in walk if g.node_len(x) > 500], fresh=False)
t = time.perf_counter()
for f in seq[:100]:
    paths = [list(f.flat[f.offs[i]:f.offs[i + 1]]) for i in range(f.n)]
    o2.calc_prob(paths, fresh=False)
t_cpu = (time.perf_counter() - t) / 100
emit({"config": 5, "what": "1000 annealing-pattern evaluations (edited path sets, ~900 paths, new junction windows aligned on the fly) on config 3's reads, ONE GPU",
      "gpu_total_s": tot, "gpu_us_per_eval_median": float(np.median(per)), "gpu_us_per_eval_p90": float(np.percentile(per, 90)),
      "gpu_us_per_eval_max": float(per.max()), "table_stats": ctx.debug_table_stats(0),
      "cpu_oracle_incremental_ms_per_eval_on_50000_pairs": t_cpu * 1e3,
      "cpu_oracle_incremental_ms_per_eval_scaled_to_all_pairs": t_cpu * 1e3 * wl.n_pairs / ns})
if out_path:
    with open(out_path, "w") as f:
        for d in lines:
            f.write(json.dumps(d) + "\n")
