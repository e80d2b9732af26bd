"""Host side of the PacBio cache-miss path (no GPU): the library's SAM parser, DP band and the
run-length CIGAR it hands to the kernel, against the oracle's restatement of ParseAligment
(reference graph.cc:2945-3021) and of the cell list of AligmentProbability (graph.cc:2183-2221)."""
import numpy as np

from gaml_amd import api, synth
from oracle import oracle_py as O

QUIRKS = [
    "q/1\t0\tp\t10\t1\t5S10M\t*\t0\t10\tACGTACGTAC\t*",                       # junk operation folded into the next count
    "q\t0\tp\t10\t1\t10M\t*\t0\t10\tACGTACGTAC\t*",                            # no '/': empty name
    "q/1\t16\tp\t10\t1\t3I4M2D3M\t*\t0\t9\tACGTACGTAC\t*\tNM:i:3",             # leading insertions, reverse strand
    "q/1\t0\tp\t3\t1\t10I\t*\t0\t0\tACGTACGTAC\t*",                            # only insertions: no clip boxes
    "q/1\t0\tp\t3\t1\t*\t*\t0\t0\tACGTACGTAC\t*",                              # empty CIGAR
    "q/1\t0\tp\t2\t1\t5M\t*\t0\t5\tACGTA\t*\tXS:i:300\tXE:i:305\tXQ:i:700",    # clips longer than the boxes, head > posstart
    "q/1\t16\tp\t100\t1\t5M0D3M\t*\t0\t8\tACGTACGT\t*\tXS:i:4\tXE:i:12\tXQ:i:250",
    "q/1\t0\tp\t100\t1\t2M250I3M\t*\t0\t5\tACGTA\t*",                          # a long insertion run inside
    "q/a/b/0_5\t0\tp\t7\t1\t1M\t*\t0\t1\tA\t*\tXS\tX\tXE:i:\tNM:i:x",          # short / empty tags
    "q/1\t0\tp\t9\t1\t3M2I2I1D0I4M\t*\t0\t8\tACGTACGTACG\t*",                  # adjacent insertion operations
]


def _lines():
    gen = synth.make_genome(20000, 5)
    g = synth.make_graph(gen, synth.cut_lengths(20000, 5))
    ps = synth.make_pacbio_sam(g, synth.genome_walk(g), 150, 800, 3)
    return ps.sam.split("\n")[1:-1] + QUIRKS, 2 * 20000 + 1


def test_parser_and_band_match_the_oracle():
    lines, total = _lines()
    for l in lines:
        fo, r0o, loo, hio = O.sam_band(l, total)
        fp, r0p, lop, hip = api.debug_sam_band(l, total)
        assert fo == fp, l
        assert r0o == r0p and np.array_equal(loo, lop) and np.array_equal(hio, hip), l


def test_kernel_inputs_describe_the_same_band():
    """Replay of the kernel's band derivation (5-row sliding window over the run-length CIGAR) in
    Python: same rows and columns as the materialised band, and max_width bounds every row."""
    lines, total = _lines()
    for l in lines:
        _, r0, lo, hi = api.debug_sam_band(l, total)
        sh, ops = api.debug_sam_shape(l, total)
        assert all(n > 0 for n, _ in ops) and all(not (a[1] == "I" and b[1] == "I") for a, b in zip(ops, ops[1:])), l
        assert sum(n for n, c in ops if c != "I") == sh["row_f"] and sum(n for n, c in ops if c != "D") == sh["col_f"]
        r_first = -sh["bl"] if sh["bl"] > 0 else 0
        r_last = max(sh["row_f"], sh["row_f"] + sh["el"] - 1, 2 if sh["bl"] > 0 else 0)
        assert r0 == r_first - 2 and len(lo) == r_last - r_first + 5, l
        assert int((hi - lo + 1).max()) <= sh["max_width"], (l, int((hi - lo + 1).max()), sh)
        # first-pass rows from the run-length CIGAR
        k, used, col = 0, 0, 0
        first = {}
        for r in range(r_first, r_last + 1):
            a, b = [], []
            if r == 0:
                a.append(0); b.append(0)
            if sh["bl"] > 0 and -sh["bl"] <= r <= 2:
                a.append(0); b.append(sh["bl"] - 1)
            if 0 <= r <= sh["row_f"]:
                enter = col
                if k < len(ops) and ops[k][1] == "I":
                    col += ops[k][0]; k += 1
                a.append(enter); b.append(col)
                if r < sh["row_f"]:
                    assert ops[k][1] != "I"
                    if ops[k][1] == "M":
                        col += 1
                    used += 1
                    if used == ops[k][0]:
                        k += 1; used = 0
            if sh["row_f"] <= r < sh["row_f"] + sh["el"]:
                a.append(sh["col_f"] - sh["el"]); b.append(sh["col_f"])
            first[r] = (min(a), max(b))
        for i in range(len(lo)):
            r = r0 + i
            win = [first[q] for q in range(r - 2, r + 3) if q in first]
            assert lo[i] == min(w[0] for w in win) - 2 and hi[i] == max(w[1] for w in win) + 2, (l, r)
