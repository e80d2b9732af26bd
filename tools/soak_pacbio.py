"""Fuzz of the PacBio alignment DP: random CIGARs (runs of M/I/D, zero-length and junk operations), random
positions (including alignments that run past the end of the target or into the separator), both strands,
random soft clips (XS/XE/XQ) -- GPU kernel through the C ABI against the oracle: band equal row by row,
log probability to 1e-9 (or both minus infinity).   python tools/soak_pacbio.py [cases]"""
import os
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")  # tools look inside the library: the development build
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gaml_amd import api, synth  # noqa: E402
import oracle_py as op  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(77)
ctx = api.Context()
worst, n_inf = 0.0, 0
for k in range(cases):
    if k % 50 == 0:
        half = "".join("ACGTN"[int(x)] for x in rng.choice(5, int(rng.integers(200, 900)), p=[0.245, 0.245, 0.245, 0.245, 0.02]))
        target = half + "\n" + synth.revcomp_str(half)
    ops = []
    for _ in range(int(rng.integers(1, 40))):
        ops.append((int(rng.integers(0, 25)) if rng.random() < 0.95 else int(rng.integers(25, 320)), ("MIDMMMS" if rng.random() < 0.1 else "MIDMMMM")[int(rng.integers(0, 7))]))
    cigar = "".join(f"{n}{c}" for n, c in ops) or "*"
    aligned = sum(n for n, c in ops if c in "MI")
    span = sum(n for n, c in ops if c in "MD")
    clip_l = int(rng.integers(0, 260)) if rng.random() < 0.35 else 0
    clip_r = int(rng.integers(0, 260)) if rng.random() < 0.35 else 0
    slen = clip_l + aligned + clip_r
    if slen == 0:
        continue
    read = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, slen))
    if rng.random() < 0.75 and span + 10 < len(half):  # inside the target; the rest: anywhere, also past its end
        pos = int(rng.integers(0, len(half) - span - 5))
    else:
        pos = int(rng.integers(0, len(half) + 20))
    flag = 16 if rng.random() < 0.5 else 0
    tags = []
    if clip_l or clip_r or rng.random() < 0.2:
        tags = [f"XS:i:{clip_l + 1}", f"XE:i:{clip_l + aligned + 1}", f"XQ:i:{slen}"]
    line = "\t".join([f"r{k}/0_{slen}", str(flag), "p", str(pos), "254", cigar, "*", "0", str(span), "A" * (aligned if tags else slen), "*"] + tags)
    f, r0, lo, hi = op.sam_band(line, len(target))
    want = op.sam_alignment_logprob(line, target, read, 0.15)
    got, dlo, dhi = ctx.debug_sam_logprob(target, read, line, 0.15, with_band=True)
    assert np.array_equal(dlo, lo) and np.array_equal(dhi, hi), line
    if np.isinf(want) or np.isinf(got):
        assert got == want, (line, got, want)
        n_inf += 1
    else:
        worst = max(worst, abs(got - want) / abs(want))
        assert abs(got - want) <= 1e-9 * abs(want), (line, got, want)
print(f"pacbio soak passed: {cases} random SAM lines, {n_inf} with probability zero on both sides, worst relative difference {worst:.2e}")
