"""Where does a wave of the scoring kernel spend its time?  Ablation 8 stamps the wall clock (10 ns units) at the
stage boundaries of every wave; this prints per class the distribution of each stage, relative to the first wave's
entry.  python tools/kernel_timeline.py [cfg3]"""
import os, sys
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
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
variants = [api.FlatPaths(v) for v in bench.path_variants(synth.genome_walk(g))]
[ctx.calc_prob(v) for v in variants]
ctx.compact_tables()
[ctx.calc_prob(v) for v in variants]
for k, v in eval(sys.argv[2]).items() if len(sys.argv) > 2 else []:
    ctx.debug_set_knob(k, v)
ctx.debug_set_knob(3, 8)
names = ["entry", "tables->LDS", "records in", "occurrences in", "memo in", "stores out", "block reduced"]
for rep in range(3):
    for i in range(8):
        ctx.calc_prob(variants[0 if os.environ.get("SAME_PATHS") else i])  # SAME_PATHS=1: 2T constant, so no memo rebuild before the launch
    t = ctx.debug_timeline(rs).astype(np.int64)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    print(f"--- evaluation {rep}: {len(t)} waves, last block reduced at {(t[:, 6].max() - t0) / 100:.2f} us after the first entry")
    for cls, label in ((0, "compact"), (1, "2-record"), (2, "4-record")):
        w = t[t[:, 7] == cls]
        if not len(w):
            continue
        print(f"  class {label}: {len(w)} waves")
        cols = range(7) if cls == 0 else ((0, 2, 3, 5, 6) if cls == 1 else (0, 6))  # the two-record class stamps: records in, candidates in, first pair finished
        prev = None
        for c in cols:
            abs_us = (w[:, c] - t0) / 100
            line = f"    {names[c]:>15}: at {np.median(abs_us):6.2f} us (min {abs_us.min():5.2f}, p90 {np.percentile(abs_us, 90):5.2f}, max {abs_us.max():5.2f})"
            if prev is not None:
                d = (w[:, c] - w[:, prev]) / 100
                line += f"   stage {np.median(d):5.2f} us (p90 {np.percentile(d, 90):5.2f})"
            print(line)
            prev = c
    if True:  # which blocks are late, evaluation after evaluation? (lb = blocks - 1 - blockIdx: position in the pair order)
        fw = ctx.debug_timeline(rs).astype(np.int64)
        okc = (fw[:, 0] > 0) & (fw[:, 7] == 0)
        dn = (fw[:, 6] - t0) / 100
        nblk = len(fw) // 4
        late = sorted(set((np.arange(len(fw))[okc & (dn > np.percentile(dn[okc], 97))] // 4).tolist()))
        rec = {b: float(((fw[b * 4:(b + 1) * 4, 2] - fw[b * 4:(b + 1) * 4, 1]) / 100).max()) for b in late}
        print("  late blocks (done beyond p97), blockIdx:records stage us:", ", ".join(f"{b}:{rec[b]:.1f}" for b in late))
    if rep == 2:  # dispatch order: when does block b enter?
        full = ctx.debug_timeline(rs).astype(np.int64)
        nb = len(full) // 4
        ent = (full[: nb * 4, 0].reshape(nb, 4).min(axis=1) - t0) / 100
        print("  entry time by blockIdx:", ", ".join(f"{b}: {ent[b]:.2f}" for b in (0, 1, 2, 3, 8, 16, 64, 128, 256, 512, 768, nb - 2, nb - 1) if b < nb))
        order = np.argsort(ent)
        print("  first blocks to enter:", order[:16].tolist(), " last:", order[-8:].tolist())
        # the stragglers: which compact-class waves finish last, and in which stage do they lose their time?
        wave_id = np.arange(len(full))
        ok = (full[:, 0] > 0) & (full[:, 7] == 0)
        wv, ids = full[ok], wave_id[ok]
        done = (wv[:, 6] - t0) / 100
        med = [np.median((wv[:, c] - wv[:, c - 1]) / 100) for c in range(1, 7)]
        print("  median stage durations (tables, records, occurrences, memo, stores, reduce):", " ".join(f"{x:.2f}" for x in med))
        print("  slowest compact-class waves: block.wave, entry, then stage durations, done")
        for k in np.argsort(done)[-16:][::-1]:
            st = [(wv[k, c] - wv[k, c - 1]) / 100 for c in range(1, 7)]
            print(f"    {ids[k] // 4:4d}.{ids[k] % 4}  entry {(wv[k, 0] - t0) / 100:5.2f}  " + " ".join(f"{x:5.2f}" for x in st) + f"  done {done[k]:5.2f}")
        blk_done = np.array([done[ids // 4 == b].max() for b in np.unique(ids // 4)])
        print("  blocks done after 9 us:", int((blk_done > 9).sum()), "of", len(blk_done), "; after 8.5:", int((blk_done > 8.5).sum()), "; after 8:", int((blk_done > 8).sum()))
ctx.close()
