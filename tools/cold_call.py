"""The cold first evaluation at cfg3 (start assembly of the annealing pattern: ~990 paths, 3,952 windows to align, first table
build): wall time and host phases, twice in one process (the second context finds the runtime warm).  python tools/cold_call.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, 4)
for rnd in range(2):
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    t = time.perf_counter(); rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *reads); t_add = time.perf_counter() - t
    ctx.debug_set_knob(9, 1)
    t = time.perf_counter(); ctx.calc_prob(start); dt = time.perf_counter() - t
    ph = ctx.debug_profile()
    print(f"context {rnd}: add_paired {t_add * 1e3:.1f} ms; cold call {dt * 1e3:.1f} ms; phases us [pass1, tables_host, align (in pass1), write, sync(tables), launch, bytes, wait]:", np.round(ph, 0), flush=True)
    os.environ["GAML_HIP_TRACE_ALIGNER"] = "1"; ctx.aligner_stats(); del os.environ["GAML_HIP_TRACE_ALIGNER"]
    print("   aligner stages us:", ctx.aligner_stages(), flush=True)
    ctx.close()
