"""The PRODUCT library (libgaml_hip.so: no debug surface, every A/B knob compiled in at its default) through the parity
checks -- the rest of the suite loads the development build. One library per process, so this runs in a child process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from gaml_amd import api, synth
import oracle_py as op
assert " dev " not in api.version() and not hasattr(api._lib, "gaml_hip_debug_set_knob"), api.version()
genome, g = synth.make_repeat_graph(300_000, 99, frac=0.03)
n = 50_000
pr = synth.make_paired_reads(genome, n, 150, 300.0, 30.0, 0.01, 99)
r1, r2 = synth.pack_reads(pr.mate1), synth.pack_reads(pr.mate2)
walk = synth.genome_walk(g)
k = len(walk) // 2
sets = [[walk], [walk[:k], walk[k:]], [[x] for x in walk], [walk[:k] + [-77] + walk[k + 1:]], [[x ^ 1 for x in reversed(walk)]]]
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
rs = ctx.add_paired(api.paired_cfg(300.0, 30.0), *r1, *r2)
orc = op.Oracle()
orc.set_graph(*g.packed())
ors = orc.add_paired(*r1, *r2, 0.01, op.paired_cfg(300.0, 30.0))
for rnd in range(2):
    for paths in sets:
        got = ctx.calc_prob(paths)
        want = orc.calc_prob(paths, fresh=True)
        assert got[2] == want[2] and got[1].tolist() == want[1].tolist()
        np.testing.assert_allclose(ctx.read_probs(rs), orc.paired_probs(ors)[0], rtol=4e-16, atol=0)
        assert abs(got[0] - want[0]) <= 1e-9 * abs(want[0])
    one = [ctx.calc_prob(s)[0] for s in sets]
    assert [b[0] for b in ctx.calc_prob_batch(sets)] == one
    ctx.compact_tables()
assert sum(ctx.pair_classes(rs)) == n and ctx.table_stats(rs)["static_index_pairs"] > 0  # (the twin walk gave most reads a second record)
print("release library OK", api.version())
"""


def test_product_library_passes_the_parity_checks():
    env = dict(os.environ, GAML_HIP_FLAVOUR="release")
    env.pop("GAML_HIP_LIB", None)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "release library OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


KERNARG_CHILD = r"""
import os, sys
sys.path.insert(0, ROOT)
import numpy as np
from gaml_amd import api, synth
import ctypes
_getenv = ctypes.CDLL(None).getenv
_getenv.restype = ctypes.c_char_p
assert _getenv(b"HIP_FORCE_DEV_KERNARG") == WANT.encode()  # (set by the library when it was loaded, unless the caller had set it; os.environ is a snapshot)
genome = synth.make_genome(1_000_000, 5)
g = synth.make_graph(genome, synth.cut_lengths(1_000_000, 5))
pr = synth.make_paired_reads(genome, 100_000, 150, 300.0, 30.0, 0.01, 5)
ctx = api.Context(device=0)
ctx.set_graph(*g.packed())
ctx.add_paired(api.paired_cfg(300.0, 30.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
walk = synth.genome_walk(g)
sets = [api.FlatPaths([walk[:k], walk[k:]]) for k in (100, 200, 300, 400)]
for s in sets: ctx.score(s)
ctx.set_event_timing(True)
ctx.kernel_stats(reset=True)
for i in range(400): ctx.score(sets[i % 4])
st = ctx.kernel_stats()
print("KERNEL_US", st["device_us"] / st["launches"])
"""


def _kernel_us(env, want):
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\nWANT = %r\n" % (ROOT, want) + KERNARG_CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "KERNEL_US" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    return float(r.stdout.split("KERNEL_US")[1].split()[0]), r.stderr


def test_kernel_arguments_live_in_device_memory_without_help_from_the_environment():
    """The scoring launch's ~600-byte argument block must sit in device memory (HIP_FORCE_DEV_KERNARG=1; host memory costs
    every launch ~4 us of PCIe round trips). A process that does not export the variable gets it from the library itself --
    set when the library is loaded, before the process's first HIP call -- and runs the kernel as fast as one that does; a
    process that explicitly switches it off is told so once."""
    base = dict(os.environ, GAML_HIP_FLAVOUR="release")
    base.pop("GAML_HIP_LIB", None)
    with_env, _ = _kernel_us(dict(base, HIP_FORCE_DEV_KERNARG="1"), "1")
    plain = dict(base)
    plain.pop("HIP_FORCE_DEV_KERNARG", None)
    without_env, _ = _kernel_us(plain, "1")
    assert without_env <= 1.10 * with_env + 0.3, (without_env, with_env)
    off, err = _kernel_us(dict(base, HIP_FORCE_DEV_KERNARG="0"), "0")
    assert "HIP_FORCE_DEV_KERNARG is not 1" in err and off > with_env, (off, with_env, err[-500:])
