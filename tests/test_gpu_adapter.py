"""include/gaml_hip_prob_calculator.h -- the drop-in replacement for the reference's prob_calculator.h -- compiled,
linked and run. The reference's graph.h needs Boost (absent), so the header is built over a TEST-ONLY declaration mock
of the few graph.h members it touches (tests/mock_ref/): a syntax / link / plumbing check of the adapter, not a parity
statement about graph.cc. The values must equal the ctypes path on the same files."""
import os
import re
import subprocess

import pytest

from gaml_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "mock_ref", "_build", "adapter_driver")


def test_adapter_header_compiles_and_links(built):
    """CPU side: the build produced the driver (g++ -std=c++0x over the mock), and without a GPU it fails loudly."""
    assert os.path.exists(DRIVER)
    try:
        import torch
        if torch.cuda.is_available():
            return
    except ImportError:
        pass
    out = subprocess.run([DRIVER, "/nonexistent", "a", "b", "300", "30"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["", "0,0"])
def test_adapter_calc_prob_overloads_equal_the_ctypes_value(tmp_path, devices):
    from gaml_amd import api
    d = str(tmp_path)
    G, seed = 50_000, 41
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(800, 3000)))
    synth.write_lastgraph(os.path.join(d, "LastGraph"), g)
    pr = synth.make_paired_reads(genome, 3000, 100, 250.0, 25.0, 0.01, seed)
    sr = synth.make_single_reads(genome, 800, 100, 0.01, seed)
    f1, f2, fs = (os.path.join(d, n) for n in ("a_1.fastq", "a_2.fastq", "s.fastq"))
    synth.write_fastq(f1, pr.mate1, "p", 1)
    synth.write_fastq(f2, pr.mate2, "p", 2)
    synth.write_fastq(fs, sr, "s", None)
    env = dict(os.environ)
    env.pop("GAML_HIP_DEVICES", None)
    if devices:
        env["GAML_HIP_DEVICES"] = devices
    out = subprocess.run([DRIVER, os.path.join(d, "LastGraph"), f1, f2, "250", "25", fs], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr
    m1 = re.search(r"whole (\S+) len (\d+) zeros (\d+)/(\d+) (\d+)/(\d+)", out.stdout)
    m2 = re.search(r"halves (\S+) len (\d+)", out.stdout)
    m3 = re.search(r"whole_again (\S+)", out.stdout)
    assert m1 and m2 and m3, out.stdout

    ctx = api.Context(device=0)
    ctx.load_graph(os.path.join(d, "LastGraph"))
    ctx.add_single_fastq(api.single_cfg(weight=0.5), fs)                   # the adapter adds single sets first (ctor order)
    ctx.add_paired_fastq(api.paired_cfg(250.0, 25.0), f1, f2)
    whole = [list(range(0, g.n_nodes, 2))]
    n = len(whole[0])
    halves = [whole[0][: n // 2], whole[0][n // 2:]]
    want, wz, wtl = ctx.calc_prob(whole)
    tol = 1e-12 if devices else 1e-14
    assert abs(float(m1.group(1)) - want) <= tol * abs(want) and int(m1.group(2)) == wtl
    assert [int(m1.group(k)) for k in (3, 4, 5, 6)] == wz.reshape(-1).tolist()   # zeros: single set, then paired set
    want2, _, wtl2 = ctx.calc_prob(halves)
    assert abs(float(m2.group(1)) - want2) <= tol * abs(want2) and int(m2.group(2)) == wtl2
    assert float(m3.group(1)) == float(m1.group(1))
