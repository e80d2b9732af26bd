"""Time the PacBio cache-miss side (SAM -> banded DP -> records): GPU through the C ABI vs the
CPU oracle on a sample.  python tools/pacbio_dp_bench.py [n_reads] [read_len]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from gaml_amd import api, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
G = 1_000_000
gen = synth.make_genome(G, 1)
g = synth.make_graph(gen, synth.cut_lengths(G, 1))
walk = synth.genome_walk(g)
t = time.time()
ps = synth.make_pacbio_sam(g, walk, n_reads, read_len, 1)
print(f"synth {time.time() - t:.1f} s, {ps.n_records} records", flush=True)
bases, offs = g.packed()
rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
ro = np.zeros(len(ps.reads) + 1, np.int64)
ro[1:] = np.cumsum([len(r) for r in ps.reads])
sam_bytes = ps.sam.encode()  # what reading BLASR's output file gives
ctx = api.Context()
ctx.set_graph(bases, offs)
rs = ctx.add_pacbio_reads(api.single_cfg(min_prob_per_base=-1.0, mismatch_prob=0.15), rb, ro, ps.names)
for rep in range(2):  # second round: fresh context state is not needed, re-ingest under a new set
    if rep:
        rs = ctx.add_pacbio_reads(api.single_cfg(min_prob_per_base=-1.0, mismatch_prob=0.15), rb, ro, ps.names)
    t = time.time()
    filed = ctx.pacbio_ingest_sam(rs, walk, sam_bytes)
    dt = time.time() - t
    st = ctx.pacbio_dp_stats(rs)
    print(f"gpu ingest #{rep}: {dt * 1e3:.1f} ms wall, filed {filed}; kernel {st['kernel_ms']:.2f} ms, host prepare {st['host_prepare_ms']:.1f} ms, "
          f"device leg {st['device_ms']:.1f} ms, rows {st['rows']:.0f}, cells {st['cells']:.0f}, scratch {st['scratch_bytes'] / 1e6:.1f} MB, "
          f"{st['cells'] / st['kernel_ms'] / 1e6:.2f} Gcell/s", flush=True)
# CPU oracle on a bounded sample of the same records
lines = ps.sam.split("\n")
sample = "\n".join(lines[:1 + min(60, ps.n_records)]) + "\n"
orc = O.Oracle()
orc.set_graph(bases, offs)
ors = orc.add_pacbio_reads(rb, ro, ps.names, 0.15, O.single_cfg(min_prob_per_base=-1.0))
t = time.time()
n = orc.pacbio_ingest_sam(ors, walk, sample)
dt = time.time() - t
print(f"cpu oracle: {n} records in {dt:.2f} s = {dt / max(1, n) * 1e3:.2f} ms per record (1 core)")
