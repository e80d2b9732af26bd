#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.

Two kinds of fixture, kept apart on purpose:
  * ref_*.json    -- produced by the REFERENCE's own code: oracle/_ref/libref_logdouble.so is our
                     driver TU compiled against /root/reference/logdouble.hpp and utility.h in place
                     (the only reference files that build without Boost). These pin the oracle.
  * oracle_*.json -- produced by OUR oracle (oracle/gaml_oracle.cc) on seeded synthetic inputs.
                     They are regression pins for the oracle and the product, NOT reference
                     output: graph.cc cannot be built here (Boost absent, stand-ins not allowed) and
                     the reference has no fixtures of its own, so these stages are "parity unpinned".
Run in the build container (needs /root/reference for the ref_* part): python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_py as op  # noqa: E402
from gaml_amd import synth  # noqa: E402


def hexf(x):
    return float(x).hex()


def ref_logdouble():
    R = op.ref_lib()
    assert R is not None, "oracle/_ref/libref_logdouble.so missing: run make -C oracle with /root/reference present"
    ninf = -math.inf
    vals = [ninf, -745.2, -300.5, -36.04365338911715, -10.0, -0.7, -1e-9, 0.0, 0.3, 5.0, 700.0]
    rows = []
    for a in vals:
        for b in vals:
            rows.append({"a": hexf(a), "b": hexf(b), "add": hexf(R.ref_ld_add(a, b)), "add_assign": hexf(R.ref_ld_add_assign(a, b)),
                         "mul": hexf(R.ref_ld_mul(a, b)), "div": hexf(R.ref_ld_div(a, b)) if not (a == ninf and b == ninf) else "nan",
                         "lt": int(R.ref_ld_lt(a, b)), "gt": int(R.ref_ld_gt(a, b))})
    lin = [0.0, 1e-300, 1e-35, 1e-15, 0.01, 0.15, 0.4, 0.96, 1.0, 2.0, 1e10]
    ctor = [{"x": hexf(x), "log": hexf(R.ref_ld_from_linear(x))} for x in lin]
    pows = [{"a": hexf(a), "e": hexf(e), "pow": hexf(R.ref_ld_pow(a, e))}
            for a in [math.log(0.01), math.log(0.96), math.log(0.15), math.log(0.4), 0.0]
            for e in [0.0, 0.25 * 5000, 0.75 * 5000, 150.0, 37.5]]
    walks = [[], [0], [1], [4, 7, 2], [0, -50, 3], [-7], [10, 11, -1, -300, 6, 6]]
    inv = []
    for w in walks:
        a = np.array(w if w else [0], np.int32)
        out = np.zeros(max(1, len(w)), np.int32)
        n = R.ref_invert_path(a, len(w), out)
        rev = a.copy()
        R.ref_reverse_path(rev, len(w))
        inv.append({"walk": w, "invert": [int(x) for x in out[:n]], "reverse": [int(x) for x in rev[:len(w)]]})
    return {"source": "reference logdouble.hpp + utility.h compiled in place (oracle/_ref)", "default": hexf(R.ref_ld_default()),
            "binary": rows, "ctor": ctor, "pow": pows, "paths": inv}


def tiny_case(seed=5, G=30_000, n=1500, penalty=0.0):
    genome = synth.plant_repeats(synth.make_genome(G, seed), 2, 500, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(900, 2500), short_rng=(40, 120)))
    pr = synth.make_paired_reads(genome, n, 100, 250.0, 25.0, 0.01, seed)
    return genome, g, pr


def oracle_pins():
    out = {"source": "oracle/gaml_oracle.cc on seeded synthetic input (regression pin, NOT reference output)"}
    # seed extension cases (ProcessHit): (win_pos, read_pos, read, window) -> (errs, begin, end)
    rng = np.random.default_rng(3)
    w = "".join("ACGT"[i] for i in rng.integers(0, 4, 400))
    cases = []

    def case(name, wp, rp, read, win):
        cases.append({"name": name, "win_pos": wp, "read_pos": rp, "read": read, "win": win, "out": list(op.extend_hit(wp, rp, read, win))})

    r = w[100:200]
    case("exact", 130, 30, r, w)
    for k, pos in enumerate([5, 60, 95]):
        rr = r[:pos] + ("A" if r[pos] != "A" else "C") + r[pos + 1:]
        case(f"sub{k}", 130, 30, rr, w)
    rr = r[:70] + r[71:] + w[200]
    case("del_in_read", 130, 30, rr, w)
    rr = r[:70] + "G" + r[70:-1]
    case("ins_in_read", 130, 30, rr, w)
    rr = list(r)
    for pos in (50, 55, 60, 65):
        rr[pos] = "A" if rr[pos] != "A" else "C"
    case("four_subs_fwd_fail", 130, 30, "".join(rr), w)
    case("window_start_seed0", 0, 0, w[0:100], w)
    case("window_start_overhang3", 0, 3, "TTT" + w[0:97] if w[0:3] != "TTT" else "GGG" + w[0:97], w)
    case("window_start_overhang6_fail", 0, 6, "TTTTTT" + w[0:94], w)
    case("window_end_overhang2", 385 - 83, 0, w[302:400] + "AC", w)
    out["extend_hit"] = cases
    out["insert_prob"] = [{"len": d, "mean": m, "sd": s, "p": hexf(op.lib().orc_insert_prob(float(d), m, s))}
                          for (m, s) in [(300.0, 30.0), (180.0, 20.0), (3700.0, 350.0)] for d in [0, 1, 150, 299, 300, 301, 449, 450, 1000, 1500, 6000]]
    seq = w[:260]
    out["window_hashes"] = {"seq": seq, "read_len": 100, "pairs": [[str(h), p] for h, p in op.window_hashes(seq, 100)]}

    # miniature end-to-end cases
    genome, g, pr = tiny_case()
    gb, go = g.packed()
    b1, o1 = synth.pack_reads(pr.mate1)
    b2, o2 = synth.pack_reads(pr.mate2)
    walk = synth.genome_walk(g)
    k = len(walk) // 2
    path_sets = {"one_walk": [walk], "two_walks": [walk[:k], walk[k:]], "gap": [walk[:k] + [-120] + walk[k + 2:]],
                 "twin": [[x ^ 1 for x in reversed(walk)]], "singletons": [[x] for x in walk if g.node_len(x) > 500]}
    for pen_name, pen in (("nopenalty", 0.0), ("penalty", 0.0002)):
        o = op.Oracle()
        o.set_graph(gb, go)
        rs = o.add_paired(b1, o1, b2, o2, 0.01, op.paired_cfg(250.0, 25.0, penalty_constant=pen))
        res = {}
        for name, ps in path_sets.items():
            v, z, tl = o.calc_prob(ps, fresh=True)
            probs, bad = o.paired_probs(rs)
            res[name] = {"paths": ps, "prob": hexf(v), "zeros": z.tolist(), "total_len": tl, "bad_bases": bad,
                         "probs_sum": hexf(float(np.sum(probs))), "probs_nonzero": int((probs > 0).sum())}
        # incremental sequence (ScoringState carried over), as the SA loop drives it
        inc = []
        for name in ("one_walk", "two_walks", "one_walk", "gap", "singletons", "one_walk"):
            v, z, tl = o.calc_prob(path_sets[name], fresh=False)
            inc.append({"set": name, "prob": hexf(v), "zeros": z.tolist()})
        out[f"paired_{pen_name}"] = {"cases": res, "incremental": inc}
        if pen == 0.0:
            wins = {}
            for key in sorted(o.window_keys(rs, 0))[:6]:
                wins[",".join(map(str, key))] = o.window_records(rs, 0, key).tolist()
            out["window_records_mate0"] = wins
    # single-end (cfg1 flavour)
    sr = synth.make_single_reads(genome, 1200, 100, 0.01, 9)
    sb, so = synth.pack_reads(sr)
    o = op.Oracle()
    o.set_graph(gb, go)
    ss = o.add_single(sb, so, 0.01, op.single_cfg())
    res = {}
    for name in ("one_walk", "two_walks", "gap", "singletons"):
        v, probs, o3 = o.single_detail(ss, path_sets[name])
        res[name] = {"prob": hexf(v), "zeros": int(o3[0]), "total_len": int(o3[1]), "bad_bases": int(o3[2]), "probs_sum": hexf(float(probs.sum()))}
    out["single"] = res
    # pacbio (synthetic records): dense coverage, and sparse coverage (junctions no read spans -> bad_bases)
    for tag, n_reads in (("pacbio", 60), ("pacbio_sparse", 9)):
        pb = synth.make_pacbio_records(g, walk, n_reads, 2000, 0.15, 4)
        o = op.Oracle()
        o.set_graph(gb, go)
        ps_ = o.add_pacbio(pb.lens, 0.15, op.single_cfg(penalty_constant=0.0001, min_prob_per_base=-1.06))
        for wk, rec, lp in zip(pb.walks, pb.recs, pb.logps):
            o.pacbio_put(ps_, wk, rec, lp)
        for sub in synth.all_subwalks_for_pacbio(g, walk, int(pb.lens.max())):
            o.pacbio_put(ps_, sub, np.zeros((0, 3), np.int32), np.zeros(0))
        v, lp, o3 = o.pacbio_detail(ps_, [walk])
        finite = lp[np.isfinite(lp)]
        out[tag] = {"n_reads": n_reads, "prob": hexf(v), "zeros": int(o3[0]), "total_len": int(o3[1]), "bad_bases": int(o3[2]),
                    "logprob_sum_finite": hexf(float(finite.sum())), "n_finite": int(len(finite))}
    return out


def pacbio_sam_pins():
    """Regression pins (NOT reference output) for the PacBio cache-miss side: parsed SAM fields, the DP band
    and the alignment log probability of hand-made and synthetic SAM lines, and the records one ingest files."""
    half = "".join("ACGT"[int(x)] for x in np.random.default_rng(3).integers(0, 4, 400))
    target = half + "\n" + synth.revcomp_str(half)
    rng = np.random.default_rng(4)
    lines = [
        "q/1\t0\tp\t10\t1\t5S10M\t*\t0\t10\tACGTACGTAC\t*",
        "q/1\t16\tp\t10\t1\t3I4M2D3M\t*\t0\t9\tACGTACGTAC\t*\tNM:i:3",
        "q/1\t0\tp\t3\t1\t10I\t*\t0\t0\tACGTACGTAC\t*",
        "q/1\t0\tp\t2\t1\t5M\t*\t0\t5\tACGTA\t*\tXS:i:300\tXE:i:305\tXQ:i:700",
        "q/1\t16\tp\t100\t1\t5M0D3M\t*\t0\t8\tACGTACGT\t*\tXS:i:4\tXE:i:12\tXQ:i:250",
        "q/1\t0\tp\t395\t1\t8M\t*\t0\t8\tACGTACGT\t*",
    ]
    g1 = synth.make_graph(np.frombuffer(half.encode(), np.uint8), [400])
    ps1 = synth.make_pacbio_sam(g1, [0], 6, 300, 8)
    rd1 = dict(zip(ps1.names, ps1.reads))
    cases = []
    for line in lines:
        n = 700 if "XQ:i:700" in line else 250 if "XQ:i:250" in line else len(line.split("\t")[9])
        cases.append((line, "".join("ACGT"[int(x)] for x in rng.integers(0, 4, n))))
    for line in ps1.sam.split("\n")[1:-1]:
        cases.append((line, rd1[line.split("\t")[0].split("/")[0]]))
    out = {"target": target, "mismatch_prob": 0.15, "lines": []}
    for line, read in cases:
        f, r0, lo, hi = op.sam_band(line, len(target))
        lp = op.sam_alignment_logprob(line, target, read, 0.15)
        out["lines"].append({"sam": line, "read": read, "fields": f, "row0": r0, "rows": len(lo), "lo_head": [int(x) for x in lo[:8]],
                             "hi_head": [int(x) for x in hi[:8]], "lo_sum": int(lo.sum()), "hi_sum": int(hi.sum()),
                             "logprob": hexf(lp) if np.isfinite(lp) else "-inf"})
    # one ingest on a small graph: which sub-walks get entries, and what is filed under them
    genome = synth.make_genome(20_000, 12)
    g = synth.make_graph(genome, synth.cut_lengths(20_000, 12, long_rng=(1200, 3000)))
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, 12, 700, 12)
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    o = op.Oracle()
    o.set_graph(*g.packed())
    rs = o.add_pacbio_reads(rb, ro, ps.names, 0.15, op.single_cfg(min_prob_per_base=-1.0))
    filed = o.pacbio_ingest_sam(rs, walk, ps.sam)
    recs = {}
    for k in sorted(o.pacbio_keys(rs)):
        rec, lp = o.pacbio_records(rs, list(k))
        if len(rec):
            recs[" ".join(map(str, k))] = {"rec": rec.tolist(), "logp": [hexf(float(x)) for x in lp]}
    v, z, tl = o.calc_prob([walk])
    out["ingest"] = {"genome_seed": 12, "genome_len": 20_000, "long_rng": [1200, 3000], "n_reads": 12, "read_len": 700, "sam_seed": 12,
                     "filed": int(filed), "n_keys": len(o.pacbio_keys(rs)), "records": recs, "prob": hexf(v), "zeros": z.tolist(), "total_len": int(tl)}
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sam":  # only the PacBio SAM pins (the other fixtures stay as they are)
        with open(os.path.join(HERE, "pacbio_sam_pins.json"), "w") as f:
            json.dump(pacbio_sam_pins(), f, indent=0)
    else:
        with open(os.path.join(HERE, "ref_logdouble.json"), "w") as f:
            json.dump(ref_logdouble(), f, indent=0)
        with open(os.path.join(HERE, "oracle_pins.json"), "w") as f:
            json.dump(oracle_pins(), f, indent=0)
        with open(os.path.join(HERE, "pacbio_sam_pins.json"), "w") as f:
            json.dump(pacbio_sam_pins(), f, indent=0)
    print("wrote", os.listdir(HERE))
