"""The cold first evaluation at cfg3 (every window aligned, first table build): where it goes.  python tools/cold_probe.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, 3)
os.environ["GAML_HIP_TRACE_HOST"] = "1"
os.environ["GAML_HIP_TRACE_ALIGNER"] = "1"
for rep in range(2):
    ctx = api.Context(device=0)
    ctx.set_graph(*g.packed())
    t = time.perf_counter(); rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *reads); t_add = time.perf_counter() - t
    ctx.debug_set_knob(9, 1)
    t = time.perf_counter(); ctx.calc_prob(start); t_cold = time.perf_counter() - t
    print(f"run {rep}: add_paired {t_add * 1e3:.1f} ms, cold evaluation {t_cold * 1e3:.1f} ms", flush=True)
    ctx.aligner_stats()
    ctx.close()
