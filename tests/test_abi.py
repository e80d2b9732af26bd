"""The C-ABI library loads without a GPU, exports every symbol include/gaml_hip.h declares, and
refuses to score without a device (there is no CPU scoring path in the product)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="gaml_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gaml_hip_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    """The product library exports exactly the drop-in surface (include/gaml_hip.h) and NOTHING of the test / tuning
    surface (include/gaml_hip_debug.h: entry points, A/B knobs, in-kernel time stamps); the development build, the same
    sources with -DGAML_HIP_DEV, exports both."""
    import subprocess
    names = declared_symbols()
    assert len(names) >= 30
    assert not [n for n in names if "_debug_" in n]  # the drop-in header carries no debug surface
    dbg = declared_symbols("gaml_hip_debug.h")
    assert len(dbg) >= 10 and all("_debug_" in n for n in dbg)
    exported = {}
    for lib in ("libgaml_hip.so", "libgaml_hip_dev.so"):
        out = subprocess.check_output(["nm", "-D", "--defined-only", os.path.join(ROOT, "gaml_amd", lib)], text=True)
        exported[lib] = {ln.split()[-1] for ln in out.splitlines() if ln.split()}
    release, dev = exported["libgaml_hip.so"], exported["libgaml_hip_dev.so"]
    assert not [n for n in names if n not in release], [n for n in names if n not in release]
    assert not [n for n in names + dbg if n not in dev], [n for n in names + dbg if n not in dev]
    assert not [n for n in release if "debug" in n], [n for n in release if "debug" in n]
    # ... and no kernel instantiation with in-kernel time stamps (paired_score_kernel<false, false, true>) in the product
    blob = open(os.path.join(ROOT, "gaml_amd", "libgaml_hip.so"), "rb").read()
    assert b"paired_score_kernelILb0ELb0ELb1E" not in blob
    assert b"paired_score_kernelILb0ELb0ELb1E" in open(os.path.join(ROOT, "gaml_amd", "libgaml_hip_dev.so"), "rb").read()
    lib = ctypes.CDLL(os.path.join(ROOT, "gaml_amd", "libgaml_hip.so"))
    assert not [n for n in names if not hasattr(lib, n)]


def test_library_was_built_from_this_tree(built):
    """gaml_hip_version() carries the hash of the sources the .so was built from (csrc/Makefile: SRC_HASH): the GPU box
    receives the prebuilt library, this is how a stale one is noticed."""
    import hashlib
    import subprocess
    from gaml_amd import api
    # the ONE list of sources: the Makefile's (everything in csrc/ and include/, so a new header cannot be left out)
    rel = subprocess.check_output(["make", "-s", "-C", os.path.join(ROOT, "gaml_amd", "csrc"), "print-srcs"], text=True).split()
    assert "gaml_amd/csrc/pacbio_sweep.hip.h" in rel and "include/gaml_hip_debug.h" in rel and len(rel) >= 14
    h = hashlib.sha256()
    for r in rel:
        h.update(open(os.path.join(ROOT, r), "rb").read())
    assert api.version().endswith("src " + h.hexdigest()[:16]), api.version()
    assert " dev " in api.version()  # tests load the development build (tests/conftest.py)
    rel = ctypes.CDLL(os.path.join(ROOT, "gaml_amd", "libgaml_hip.so"))
    rel.gaml_hip_version.restype = ctypes.c_char_p
    v = rel.gaml_hip_version().decode()
    assert v.endswith("src " + h.hexdigest()[:16]) and " dev " not in v, v


def test_record_layouts_match_the_reference_structs(built):
    from gaml_amd import api
    assert api.ALIGMENT.itemsize == 16        # Aligment = 4 x int32 (graph.h:211-231)
    assert api.PACBIO_ALIGMENT.itemsize == 24  # PacbioAligment = 3 x int32 + logdouble (graph.h:516-535)
    assert ctypes.sizeof(api.PairedCfg) == 64 and ctypes.sizeof(api.SingleCfg) == 48


def test_host_only_context_refuses_to_score(built):
    from gaml_amd import api, synth
    genome = synth.make_genome(5000, 1)
    g = synth.make_graph(genome, synth.cut_lengths(5000, 1, long_rng=(900, 1500)))
    pr = synth.make_paired_reads(genome, 50, 100, 250.0, 25.0, 0.01, 1)
    ctx = api.Context(device=-1)
    ctx.set_graph(*g.packed())
    rs = ctx.add_paired(api.paired_cfg(250.0, 25.0), *synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2))
    assert ctx.num_readsets() == 1 and ctx.readset_reads(rs) == 50 and ctx.readset_kind(rs) == 1
    assert ctx.num_nodes() == g.n_nodes and ctx.node_len(0) == g.node_len(0)
    with pytest.raises(api.GamlHipError) as e:
        ctx.calc_prob([synth.genome_walk(g)])
    assert e.value.code == api.ENODEVICE
    with pytest.raises(api.GamlHipError):
        ctx.calc_partials([synth.genome_walk(g)])
    with pytest.raises(api.GamlHipError):
        ctx.read_probs(rs)


def test_argument_errors(built):
    from gaml_amd import api
    ctx = api.Context(device=-1)
    with pytest.raises(api.GamlHipError):  # odd node count: twin of i is i^1
        ctx.set_graph(np.frombuffer(b"ACGT", np.uint8), np.array([0, 2, 3, 4], np.int64))
    with pytest.raises(api.GamlHipError):
        ctx.load_graph("/nonexistent/LastGraph")
    with pytest.raises(api.GamlHipError):
        ctx.add_paired_fastq(api.paired_cfg(300, 30), "/nonexistent/a.fq", "/nonexistent/b.fq")
    with pytest.raises(api.GamlHipError):  # device ordinal that does not exist
        api.Context(device=4096)
    assert "gaml_hip" in api.version()
