"""The C++ host mirror (gaml_amd/host: LoadConfig / PrepareReadSetFromConfig / ProbCalculator over
the C ABI) driven by a GAML config file, against the oracle driven by the same file."""
import os
import re
import subprocess

import numpy as np
import pytest

from gaml_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "gaml_amd", "host", "gaml_score")


def _write_case(d, two_sets=True):
    G, seed = 60_000, 71
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(800, 3500)))
    synth.write_lastgraph(os.path.join(d, "LastGraph"), g)
    pr = synth.make_paired_reads(genome, 2500, 100, 250.0, 25.0, 0.01, seed)
    synth.write_fastq(os.path.join(d, "a_1.fastq"), pr.mate1, "p", 1)
    synth.write_fastq(os.path.join(d, "a_2.fastq"), pr.mate2, "p", 2)
    sets = [dict(name="rs1", type="paired", filename1=os.path.join(d, "a_1.fastq"), filename2=os.path.join(d, "a_2.fastq"),
                 insert_mean=250, insert_std=25, penalty_step=30, penalty_constant=0.00007, min_prob_per_base=0, min_prob_start=-20)]
    if two_sets:
        sr = synth.make_single_reads(genome, 900, 100, 0.01, seed)
        synth.write_fastq(os.path.join(d, "s.fastq"), sr, "s", None)
        sets.append(dict(name="zz", type="single", filename=os.path.join(d, "s.fastq"), weight=0.5, min_prob_per_base=-0.6))
    synth.write_config(os.path.join(d, "run.cfg"), os.path.join(d, "LastGraph"), sets, extra={"t0": 0.02, "long_contig_threshold": 700})
    return g


@pytest.mark.gpu
def test_gaml_score_matches_oracle_from_the_same_config(tmp_path):
    import oracle_py as op
    d = str(tmp_path)
    g = _write_case(d)
    walk = synth.genome_walk(g)
    with open(os.path.join(d, "x.walks"), "w") as f:  # Graph::OutputPathC format (graph.cc:277-291)
        for cid, w in enumerate([walk[:11] + [-140] + walk[13:], [x ^ 1 for x in reversed(walk[:4])]]):
            pos, parts = 0, []
            for x in w:
                parts.append(f"{x}({pos})")
                pos += g.node_len(x) if x >= 0 else -x
            f.write(f">tmp{cid}-" + "-".join(parts) + "\n")
    paths = [walk[:11] + [-140] + walk[13:], [x ^ 1 for x in reversed(walk[:4])]]
    orc = op.Oracle()
    assert orc.load_config(os.path.join(d, "run.cfg")) == 2
    for args, ps in (([os.path.join(d, "x.walks")], paths), ([], [[i] for i in range(0, g.n_nodes, 2) if g.node_len(i) > 700])):
        out = subprocess.run([CLI, os.path.join(d, "run.cfg")] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        m = re.search(r"start prob (\S+) len (\d+) low prob reads(.*)", out.stdout)
        got, tl = float(m.group(1)), int(m.group(2))
        zeros = [[int(a), int(b)] for a, b in re.findall(r"(\d+)/(\d+)", m.group(3))]
        want, wz, wtl = orc.calc_prob(ps, fresh=True)
        assert tl == wtl and zeros == wz.tolist()
        assert abs(got - want) <= 1e-9 * abs(want)


@pytest.mark.gpu
def test_cfg1_at_its_stated_size_through_files(tmp_path):
    """BASELINE config 1 as stated: a 50 kbp synthetic Velvet LastGraph and 10,000 x 100 bp single reads, through the
    files a GAML user has -- LastGraph (graph.cc:52-106), FASTQ, GAML config (gaml.cc:748-872) -- into the C++ host
    mirror's ProbCalculator, against the oracle reading the same config. (BASELINE runs this configuration on the CPU
    only; here it is the plumbing check of the file readers at full size.)"""
    import oracle_py as op
    d = str(tmp_path)
    G, n, seed = 50_000, 10_000, 101
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed))
    synth.write_lastgraph(os.path.join(d, "LastGraph"), g)
    sr = synth.make_single_reads(genome, n, 100, 0.01, seed)
    synth.write_fastq(os.path.join(d, "reads.fastq"), sr, "s", None)
    synth.write_config(os.path.join(d, "run.cfg"), os.path.join(d, "LastGraph"),
                       [dict(name="single1", type="single", filename=os.path.join(d, "reads.fastq"))], extra={"long_contig_threshold": 500})
    walk = synth.genome_walk(g)
    with open(os.path.join(d, "genome.walks"), "w") as f:
        pos, parts = 0, []
        for x in walk:
            parts.append(f"{x}({pos})")
            pos += g.node_len(x)
        f.write(">tmp0-" + "-".join(parts) + "\n")
    orc = op.Oracle()
    assert orc.load_config(os.path.join(d, "run.cfg")) == 1
    start = [[i] for i in range(0, g.n_nodes, 2) if g.node_len(i) > 500]  # gaml.cc:1002-1005
    for args, ps in (([os.path.join(d, "genome.walks")], [walk]), ([], start)):
        out = subprocess.run([CLI, os.path.join(d, "run.cfg")] + args, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        assert f"Loaded {g.n_nodes // 2} nodes" in out.stdout
        m = re.search(r"start prob (\S+) len (\d+) low prob reads(.*)", out.stdout)
        got, tl = float(m.group(1)), int(m.group(2))
        zeros = [[int(a), int(b)] for a, b in re.findall(r"(\d+)/(\d+)", m.group(3))]
        want, wz, wtl = orc.calc_prob(ps, fresh=True)
        assert tl == wtl and zeros == wz.tolist() and zeros[0][1] == n
        assert abs(got - want) <= 1e-9 * abs(want)


def test_gaml_score_fails_loudly_without_a_gpu(tmp_path, built):
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    d = str(tmp_path)
    _write_case(d, two_sets=False)
    out = subprocess.run([CLI, os.path.join(d, "run.cfg")], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert "no HIP device" in out.stderr or "HIP" in out.stderr
    assert "start prob" not in out.stdout


@pytest.mark.gpu
def test_cfg4_style_run_with_an_external_aligner(tmp_path):
    """BASELINE config 4 in miniature: a paired set (weight 1) + PacBio reads (weight 0.5, mismatch_prob
    0.15) whose alignment cache is empty, so the first CalcProb has to run the external aligner exactly
    like the reference (graph.cc:2705-2715). `blasr_path` points at a stand-in that prints a prepared
    SAM file (parity at the BLASR boundary is pinned by fixing the SAM text); the oracle is
    fed the same text."""
    import stat
    import oracle_py as op
    d = str(tmp_path)
    G, seed = 40_000, 91
    genome = synth.make_genome(G, seed)
    g = synth.make_graph(genome, synth.cut_lengths(G, seed, long_rng=(1200, 4000)))
    synth.write_lastgraph(os.path.join(d, "LastGraph"), g)
    pr = synth.make_paired_reads(genome, 2000, 150, 300.0, 30.0, 0.01, seed)
    synth.write_fastq(os.path.join(d, "a_1.fastq"), pr.mate1, "p", 1)
    synth.write_fastq(os.path.join(d, "a_2.fastq"), pr.mate2, "p", 2)
    walk = synth.genome_walk(g)
    ps = synth.make_pacbio_sam(g, walk, 80, 1500, seed)
    with open(os.path.join(d, "pb.fastq"), "w") as f:
        for name, read in zip(ps.names, ps.reads):
            f.write(f"@{name} extra words\n{read}\n+\n{'I' * len(read)}\n")
    tool = os.path.join(d, "tools")
    os.makedirs(tool)
    with open(os.path.join(tool, "prepared.sam"), "w") as f:
        f.write(ps.sam)
    with open(os.path.join(tool, "blasr"), "w") as f:
        f.write('#!/bin/sh\ncat "$(dirname "$0")/prepared.sam"\n')
    os.chmod(os.path.join(tool, "blasr"), os.stat(os.path.join(tool, "blasr")).st_mode | stat.S_IEXEC)
    sets = [dict(name="il", type="paired", filename1=os.path.join(d, "a_1.fastq"), filename2=os.path.join(d, "a_2.fastq"),
                 insert_mean=300, insert_std=30),
            dict(name="pb", type="pacbio", filename=os.path.join(d, "pb.fastq"), weight=0.5, mismatch_prob=0.15, min_prob_per_base=-1.0)]
    synth.write_config(os.path.join(d, "run.cfg"), os.path.join(d, "LastGraph"), sets, extra={"blasr_path": tool})
    with open(os.path.join(d, "x.walks"), "w") as f:
        pos, parts = 0, []
        for x in walk:
            parts.append(f"{x}({pos})")
            pos += g.node_len(x)
        f.write(">tmp0-" + "-".join(parts) + "\n")
    out = subprocess.run([CLI, os.path.join(d, "run.cfg"), os.path.join(d, "x.walks")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "records filed" in out.stdout
    m = re.search(r"start prob (\S+) len (\d+) low prob reads(.*)", out.stdout)
    got, tl = float(m.group(1)), int(m.group(2))
    zeros = [[int(a), int(b)] for a, b in re.findall(r"(\d+)/(\d+)", m.group(3))]
    orc = op.Oracle()
    orc.set_graph(*g.packed())
    orc.add_paired(*synth.pack_reads(pr.mate1), *synth.pack_reads(pr.mate2), 0.01, op.paired_cfg(300.0, 30.0))
    rb = np.frombuffer("".join(ps.reads).encode(), np.uint8)
    ro = np.zeros(len(ps.reads) + 1, np.int64)
    ro[1:] = np.cumsum([len(r) for r in ps.reads])
    ors = orc.add_pacbio_reads(rb, ro, ps.names, 0.15, op.single_cfg(min_prob_per_base=-1.0, weight=0.5))
    assert orc.pacbio_ingest_sam(ors, walk, ps.sam) > 0
    want, wz, wtl = orc.calc_prob([walk])
    assert tl == wtl and zeros == wz.tolist()
    assert wz[1][0] < wz[1][1] // 2  # most long reads score above the floor: the cache really got filled
    assert abs(got - want) <= 1e-9 * abs(want)
