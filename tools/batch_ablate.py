"""Which class of blocks bounds paired_score_multi_kernel: candidate batches at cfg3 with classes of blocks left out
(knob 11 = 32 + mask; results are wrong, only the time is looked at).  python tools/batch_ablate.py"""
import os, sys, time
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaml_amd import synth, api
wl = synth.WORKLOADS["cfg3"]
genome = synth.make_genome(wl.genome_len, wl.seed)
g = synth.make_graph(genome, synth.cut_lengths(wl.genome_len, wl.seed))
pr = synth.make_paired_reads(genome, wl.n_pairs, wl.read_len, wl.insert_mean, wl.insert_std, wl.err, wl.seed)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
start, seq = synth.sa_sequence(g, 200, seed=11)
base = seq[-1]
rng = np.random.default_rng(23)
batches = []
for _ in range(40):
    cands = [synth.sa_move(rng, base, g) for _ in range(8)]
    batches.append(api.BatchPaths(cands))
    if rng.random() < 0.6: base = cands[int(rng.integers(0, 8))]
ctx.calc_prob(base)
ctx.set_event_timing(True)
for b in batches: ctx.calc_prob_batch(b)
for mask, what in ((0, "everything"), (1, "without compact"), (2, "without <=2"), (4, "without <=4"), (8, "without delta"), (16, "without wave-per-pair"), (30, "compact only"), (32 + 30, "compact only, no capture"), (32, "everything, no capture"), (0, "everything")):
    ctx.debug_set_knob(11, 32 + mask)
    for b in batches[:5]: ctx.calc_prob_batch(b)
    ctx.kernel_stats(reset=True)
    t = time.perf_counter()
    for b in batches: ctx.calc_prob_batch(b)
    dt = time.perf_counter() - t
    st = ctx.kernel_stats()
    print(f"{what:24s}: {dt / 320 * 1e6:5.1f} us per set; scoring launches {st['launches']}, {st['device_us'] / max(1, st['launches']):.1f} us each", flush=True)
print(ctx.debug_class_counts(rs), ctx.debug_table_stats(rs))
# launch duration over the number of sets in one launch (batches of up to 4 sets go out in one launch)
ctx.debug_set_knob(11, 0)
for n in (2, 3, 4):
    small = []
    rng = np.random.default_rng(5)
    for _ in range(40):
        small.append(api.BatchPaths([synth.sa_move(rng, base, g) for _ in range(n)]))
    for knob in (0, 64):
        ctx.debug_set_knob(11, knob)
        for b in small[:10]: ctx.calc_prob_batch(b)
        ctx.kernel_stats(reset=True)
        for b in small: ctx.calc_prob_batch(b)
        st = ctx.kernel_stats()
        print(f"{n} sets per launch, {'capture' if knob == 0 else 'every set resolves'}: {st['device_us'] / max(1, st['launches']):.1f} us per launch", flush=True)
ctx.close()
