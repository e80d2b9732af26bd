"""Contexts created and destroyed in a loop: no device-memory drift, same values. Exercises every buffer a context can
own: a path set with a repeated window (general pass: gen_bits), the stream-ordered route + fetch kernel (mapped pinned
fetch buffer), a batch (ring arena, multi-set partials), the resident tables, the aligner's small-batch buffers, a
multi-device context with its worker threads and -- one rank -- an RCCL communicator.   python tools/lifecycle.py"""
import os, sys, resource, gc
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaml_amd import synth, api
genome = synth.make_genome(300_000, 5)
g = synth.make_graph(genome, synth.cut_lengths(300_000, 5))
pr = synth.make_paired_reads(genome, 40_000, 150, 300.0, 30.0, 0.01, 5)
reads = (*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
walk = synth.genome_walk(g)
repeated = [walk[:12] + walk[8:12] + walk[12:]]  # windows of walk[8:12] occur twice
free0 = None
d_part = torch.zeros(4, dtype=torch.float64, device="cuda")
for it in range(25):
    kind = it % 3
    ctx = api.Context(devices=[0, 0]) if kind == 1 else api.Context(device=0)
    if kind == 2:
        ctx.comm_init_rank(api.comm_unique_id(), 0, 1)
    ctx.set_graph(*g.packed())
    ctx.add_paired(api.paired_cfg(300.0, 30.0, penalty_constant=0.0003 if it % 2 else 0.0), *reads)
    v = [ctx.calc_prob([walk])[0], ctx.calc_prob([walk[:9], walk[9:]])[0], ctx.calc_prob(repeated)[0]]
    ctx.calc_prob_batch([[walk], [walk[:5], walk[5:]], repeated])
    if kind == 0:
        ctx.calc_partials_async([walk[:7], walk[7:]], d_part.data_ptr())
        ctx.fetch_async(d_part.data_ptr(), 4); out = np.zeros(4); ctx.fetch_wait(out.ctypes.data, 4)
    ctx.close()
    del ctx
    gc.collect()
    free, total = torch.cuda.mem_get_info()
    if it == 5: free0 = free
    if it % 6 == 0 or it == 24: print(it, v, "device free MB", free >> 20, "RSS MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10, flush=True)
print("device memory drift MB since iteration 5:", (free0 - free) >> 20)
