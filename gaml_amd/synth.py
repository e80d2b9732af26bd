"""Seeded synthetic inputs for the GAML likelihood path.

Everything here is ours (the reference ships no data): a uniform random genome, cut into
alternating long / short nodes written as a Velvet ``LastGraph`` (the format
``LoadGraph`` parses, reference graph.cc:52-106), paired / single FASTQ with substitution
errors, PacBio-like alignment records, and a GAML config file (reference gaml.cc:748-872,
README.md:40-95).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def revcomp(a: np.ndarray) -> np.ndarray:
    """Reverse complement along the last axis (reference graph.h:58-72)."""
    return _COMP[a[..., ::-1]]


def make_genome(length: int, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return _ACGT[rng.integers(0, 4, size=length, dtype=np.uint8)]


def plant_repeats(genome: np.ndarray, n_copies: int, rep_len: int, seed: int) -> np.ndarray:
    """Copy one segment to n_copies other places (raises alignments per read)."""
    rng = np.random.default_rng(seed + 7)
    g = genome.copy()
    src = int(rng.integers(0, len(g) - rep_len))
    for _ in range(n_copies):
        dst = int(rng.integers(0, len(g) - rep_len))
        g[dst:dst + rep_len] = g[src:src + rep_len]
    return g


def cut_lengths(total: int, seed: int, long_rng=(2000, 8000), short_rng=(40, 120)) -> list[int]:
    """Alternating long / short piece lengths summing to ``total``."""
    rng = np.random.default_rng(seed + 1)
    out, left, want_long = [], total, True
    while left > 0:
        lo, hi = long_rng if want_long else short_rng
        n = int(rng.integers(lo, hi + 1))
        if left - n < short_rng[0]:
            n = left
        out.append(n)
        left -= n
        want_long = not want_long
    return out


@dataclass
class Graph:
    """2n node sequences; node 2i is piece i forward, node 2i+1 its twin (reverse complement)."""
    seqs: list  # list[np.ndarray uint8]
    arcs: list = field(default_factory=list)  # (src velvet id, dst velvet id), 1-based signed
    true_walk: list | None = None  # the genome as a walk when it is not "every forward piece once" (collapsed repeats)

    @property
    def n_nodes(self) -> int:
        return len(self.seqs)

    def node_len(self, i: int) -> int:
        return len(self.seqs[i])

    def packed(self):
        offs = np.zeros(len(self.seqs) + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(s) for s in self.seqs])
        bases = np.concatenate(self.seqs) if self.seqs else np.zeros(0, np.uint8)
        return np.ascontiguousarray(bases), offs


def make_graph(genome: np.ndarray, cuts: list[int]) -> Graph:
    seqs, arcs, pos = [], [], 0
    for i, n in enumerate(cuts):
        piece = genome[pos:pos + n]
        seqs.append(np.ascontiguousarray(piece))
        seqs.append(np.ascontiguousarray(revcomp(piece)))
        if i > 0:
            arcs.append((i, i + 1))
        pos += n
    return Graph(seqs, arcs)


def write_lastgraph(path: str, g: Graph) -> None:
    n = g.n_nodes // 2
    with open(path, "wb") as f:
        f.write(b"%d\t0\t31\t1\n" % n)
        for i in range(n):
            f.write(b"NODE\t%d\t%d\t0\t0\t0\t0\n" % (i + 1, g.node_len(2 * i)))
            f.write(g.seqs[2 * i].tobytes() + b"\n")
            f.write(g.seqs[2 * i + 1].tobytes() + b"\n")
        for s, d in g.arcs:
            f.write(b"ARC\t%d\t%d\t1\n" % (s, d))


def genome_walk(g: Graph) -> list[int]:
    """The true genome as one walk: all forward pieces in order (a graph with collapsed repeats carries its own)."""
    return list(g.true_walk) if g.true_walk is not None else list(range(0, g.n_nodes, 2))


def make_repeat_graph(length: int, seed: int, frac: float = 0.02, rep_rng=(1000, 3000), copies: int = 5,
                      long_rng=(2000, 8000), short_rng=(40, 120)):
    """A genome with COLLAPSED repeats, the case GAML's repeat moves exist for (FixBigReps / FixRepForNode2,
    moves.cc:1156-1305): `frac` of the genome lies in repeat families of `copies` identical copies, rep_rng bases each.
    A family is ONE node of the graph, as in a de Bruijn assembly graph; the true genome walk visits it `copies` times, so
    its windows occur several times in the path set and its reads align once but sit at several path positions.
    Returns (genome sequence, graph); unique pieces alternate long / short as in make_graph, every repeat occurrence is
    followed by a short unique piece of its own."""
    rng = np.random.default_rng(seed + 11)
    cuts = cut_lengths(length, seed, long_rng, short_rng)
    n_fam = max(1, int(round(frac * length / (copies * 0.5 * (rep_rng[0] + rep_rng[1])))))
    fams = [_ACGT[rng.integers(0, 4, size=int(rng.integers(rep_rng[0], rep_rng[1] + 1)), dtype=np.uint8)] for _ in range(n_fam)]
    uniq = make_genome(length, seed)
    # unique pieces, then the places (before a long piece, behind a short one) where repeat occurrences go
    pieces, pos = [], 0
    for n in cuts:
        pieces.append(uniq[pos:pos + n]); pos += n
    slots = [i for i in range(2, len(pieces), 2)]  # index of a long piece with a short one before it
    take = rng.choice(len(slots), size=min(len(slots), n_fam * copies), replace=False)
    occ_at = {}
    for k, t in enumerate(sorted(take.tolist())):
        occ_at[slots[t]] = k % n_fam  # families interleaved along the genome
    seqs, walk, genome_parts, node_of_fam = [], [], [], {}
    def add_node(seq):
        seqs.append(np.ascontiguousarray(seq)); seqs.append(np.ascontiguousarray(revcomp(seq)))
        return len(seqs) - 2
    for i, piece in enumerate(pieces):
        if i in occ_at:
            f = occ_at[i]
            if f not in node_of_fam:
                node_of_fam[f] = add_node(fams[f])
            walk.append(node_of_fam[f]); genome_parts.append(fams[f])
            spacer = _ACGT[rng.integers(0, 4, size=int(rng.integers(short_rng[0], short_rng[1] + 1)), dtype=np.uint8)]
            walk.append(add_node(spacer)); genome_parts.append(spacer)
        walk.append(add_node(piece)); genome_parts.append(piece)
    arcs = sorted({(a // 2 + 1, b // 2 + 1) for a, b in zip(walk, walk[1:])})
    return np.concatenate(genome_parts), Graph(seqs, arcs, true_walk=walk)


def _mutate(reads: np.ndarray, err: float, rng) -> np.ndarray:
    if err <= 0:
        return reads
    hit = rng.random(reads.shape) < err
    # substitute with one of the three other bases
    code = np.zeros(256, dtype=np.uint8)
    code[list(b"ACGT")] = [0, 1, 2, 3]
    c = code[reads]
    c = np.where(hit, (c + rng.integers(1, 4, size=reads.shape, dtype=np.uint8)) & 3, c)
    return _ACGT[c]


@dataclass
class PairedReads:
    mate1: np.ndarray  # [n, L] uint8
    mate2: np.ndarray
    frag_start: np.ndarray
    frag_len: np.ndarray
    swapped: np.ndarray

    @property
    def n(self) -> int:
        return self.mate1.shape[0]


def make_paired_reads(genome: np.ndarray, n_pairs: int, read_len: int, mean: float, sd: float,
                      err: float, seed: int, chunk: int = 1 << 18) -> PairedReads:
    """Innie pairs: mate1 = forward prefix of the fragment, mate2 = reverse complement of its
    suffix; mates are swapped on odd ids."""
    rng = np.random.default_rng(seed + 2)
    G = len(genome)
    flen = np.maximum(read_len + 10, np.rint(rng.normal(mean, sd, n_pairs))).astype(np.int64)
    flen = np.minimum(flen, G)
    start = (rng.random(n_pairs) * (G - flen + 1)).astype(np.int64)
    m1 = np.empty((n_pairs, read_len), np.uint8)
    m2 = np.empty((n_pairs, read_len), np.uint8)
    ar = np.arange(read_len, dtype=np.int64)
    for lo in range(0, n_pairs, chunk):
        hi = min(n_pairs, lo + chunk)
        s, fl = start[lo:hi, None], flen[lo:hi, None]
        a = genome[s + ar]
        b = revcomp(genome[s + fl - read_len + ar])
        m1[lo:hi] = _mutate(a, err, rng)
        m2[lo:hi] = _mutate(b, err, rng)
    swapped = (np.arange(n_pairs) & 1).astype(bool)
    t = m1[swapped].copy()
    m1[swapped] = m2[swapped]
    m2[swapped] = t
    return PairedReads(m1, m2, start, flen, swapped)


def make_single_reads(genome: np.ndarray, n: int, read_len: int, err: float, seed: int) -> np.ndarray:
    rng = np.random.default_rng(seed + 3)
    G = len(genome)
    start = (rng.random(n) * (G - read_len + 1)).astype(np.int64)
    reads = genome[start[:, None] + np.arange(read_len, dtype=np.int64)]
    flip = rng.random(n) < 0.5
    reads[flip] = revcomp(reads[flip])
    return _mutate(reads, err, rng)


def pack_reads(reads: np.ndarray):
    """[n, L] matrix -> (bases, int64 offsets[n+1]) as the C ABI takes them."""
    n, L = reads.shape
    return np.ascontiguousarray(reads).reshape(-1), (np.arange(n + 1, dtype=np.int64) * L)


def write_fastq(path: str, reads: np.ndarray, prefix: str, mate: int | None) -> None:
    n, L = reads.shape
    qual = b"I" * L
    with open(path, "wb") as f:
        for i in range(n):
            name = b"@%s%d" % (prefix.encode(), i)
            if mate is not None:
                name += b"/%d" % mate
            f.write(name + b"\n" + reads[i].tobytes() + b"\n+\n" + qual + b"\n")


def write_config(path: str, graph_file: str, readsets: list[dict], extra: dict | None = None) -> None:
    """GAML config (reference gaml.cc:748-780): global key=value lines, then [name] sections."""
    with open(path, "w") as f:
        f.write(f"graph={graph_file}\n")
        for k, v in (extra or {}).items():
            f.write(f"{k}={v}\n")
        for rs in readsets:
            f.write(f"\n[{rs['name']}]\n")
            for k, v in rs.items():
                if k != "name":
                    f.write(f"{k}={v}\n")


@dataclass
class PacbioRecords:
    """Synthetic stand-in for what BLASR + the banded DP leave in the PacBio cache
    (reference graph.cc:2776-2782): per sub-walk, (position, position_end, read_id, logprob)."""
    lens: np.ndarray
    walks: list  # list[list[int]]
    recs: list   # list[np.ndarray [k,3] int32]
    logps: list  # list[np.ndarray [k] float64]


def make_pacbio_records(g: Graph, walk: list[int], n_reads: int, read_len: int, mismatch: float,
                        seed: int, extra_per_read: float = 0.3) -> PacbioRecords:
    """Place reads uniformly on ``walk``; each read gets a record under the sub-walk that the
    reference would file it under -- the run of nodes overlapping [tstart-5, tstart+len+5]
    (reference graph.cc:2761-2776) -- with a log-probability near L*(0.85*log M + 0.15*log m),
    plus a few weaker secondary records."""
    rng = np.random.default_rng(seed + 4)
    node_len = np.array([g.node_len(x) for x in walk], dtype=np.int64)
    ends = np.cumsum(node_len)
    begins = ends - node_len
    total = int(ends[-1])
    lens = np.full(n_reads, read_len, dtype=np.int32)
    lens = np.minimum(lens, total - 20).astype(np.int32)
    M, m = 1.0 - 4 * mismatch, mismatch
    by_walk: dict[tuple, list] = {}

    def file_record(rid, tstart, tlen, logp):
        ib = int(np.searchsorted(ends, max(0, tstart - 5), side="left"))
        ie = int(np.searchsorted(ends, min(tstart + tlen + 5, total), side="left"))
        ie = min(ie, len(walk) - 1)
        key = tuple(walk[ib:ie + 1])
        pb = int(begins[ib])
        by_walk.setdefault(key, []).append((tstart - pb, tstart + tlen - pb, rid, logp))

    for rid in range(n_reads):
        L = int(lens[rid])
        tstart = int(rng.integers(0, total - L))
        good = int(round(L * 0.85))
        logp = good * np.log(M) + (L - good) * np.log(m) + float(rng.normal(0, 3.0))
        file_record(rid, tstart, L, logp)
        if rng.random() < extra_per_read:
            t2 = int(rng.integers(0, total - L))
            file_record(rid, t2, L, logp - float(rng.uniform(5, 400)))
    walks, recs, logps = [], [], []
    for key, lst in by_walk.items():
        lst.sort(key=lambda t: t[0])
        walks.append(list(key))
        recs.append(np.array([[a, b, c] for a, b, c, _ in lst], dtype=np.int32))
        logps.append(np.array([d for *_, d in lst], dtype=np.float64))
    return PacbioRecords(lens, walks, recs, logps)


def revcomp_str(s: str) -> str:
    return revcomp(np.frombuffer(s.encode(), np.uint8)).tobytes().decode()


def walk_string(g: Graph, walk: list[int]) -> str:
    """Path string as the PacBio code builds it (reference graph.cc:2662-2686): gaps become N runs."""
    return "".join(g.seqs[x].tobytes().decode() if x >= 0 else "N" * (-x) for x in walk)


@dataclass
class PacbioSam:
    reads: list       # fastq orientation
    names: list       # fastq names ("r<i>"); the SAM QNAME is "<name>/0_<len>" like BLASR writes it
    sam: str          # SAM text (header line + one line per alignment)
    n_records: int


def make_pacbio_sam(g: Graph, walk: list[int], n_reads: int, read_len: int, seed: int, sub=0.05, ins=0.06, dele=0.04,
                    clip_frac=0.3, rev_frac=0.5, secondary=0.25, max_clip=60) -> PacbioSam:
    """Synthetic stand-in for BLASR's SAM output on ``walk`` (parity at the BLASR boundary
    is pinned by fixing the SAM text).  Every read is cut from the path string (either strand),
    mutated with substitutions / insertions / deletions, optionally padded with unaligned ends
    (reported through the XS/XE/XQ tags like BLASR's soft clips); the CIGAR is the true edit
    script.  Coordinates follow the reference's reading of the file (ParseAligment
    graph.cc:2945-3021: POS is used as a 0-based offset, column 9 is the reference span, reverse
    strand records are mirrored into the second half of path + separator + reverse complement).
    A fraction of reads gets a second, wrong-place record (low probability) as BLASR's extra
    candidates would."""
    rng = np.random.default_rng(seed + 11)
    seq = walk_string(g, walk)
    both = seq + "\n" + revcomp_str(seq)
    total_len = len(both)
    S = len(seq)
    acgt = "ACGT"
    reads, names, lines = [], [], ["@HD\tVN:1.0"]

    def mutate(tpl: str):
        out, ops = [], []
        for ch in tpl:
            while rng.random() < ins:
                out.append(acgt[int(rng.integers(4))]); ops.append("I")
            u = rng.random()
            if u < dele:
                ops.append("D")
            else:
                out.append(acgt[int(rng.integers(4))] if u < dele + sub else ch); ops.append("M")
        # a local alignment starts and ends with a matched column
        shift = 0  # path bases dropped in front
        while ops and ops[0] != "M":
            if ops[0] == "I":
                out.pop(0)
            else:
                shift += 1
            ops.pop(0)
        while ops and ops[-1] != "M":
            if ops[-1] == "I":
                out.pop()
            ops.pop()
        return "".join(out), ops, shift

    def rle(ops):
        res, i = [], 0
        while i < len(ops):
            j = i
            while j < len(ops) and ops[j] == ops[i]:
                j += 1
            res.append(f"{j - i}{ops[i]}"); i = j
        return res

    def emit(rid, name, read, clip_l, clip_r, ops, start_in_both):
        """One SAM line for `read` whose aligned part (read[clip_l:len-clip_r]) follows `ops` from
        both[start_in_both]."""
        span = sum(1 for o in ops if o != "I")
        aligned_len = len(read) - clip_l - clip_r
        if start_in_both <= S:  # forward half
            flag, pos, cig = 0, start_in_both, rle(ops)
        else:                   # mirrored: posstart' = total_len - posend
            flag, pos, cig = 16, total_len - start_in_both - span, list(reversed(rle(ops)))
        tags = [f"NM:i:{sum(1 for o in ops if o != 'M')}"]
        if clip_l or clip_r:
            tags += [f"XS:i:{clip_l + 1}", f"XE:i:{clip_l + aligned_len + 1}", f"XQ:i:{len(read)}"]
        seq_col = "A" * (aligned_len if (clip_l or clip_r) else len(read))  # only its length is read
        lines.append("\t".join([f"{name}/0_{len(read)}", str(flag), "path", str(pos), "254", "".join(cig), "*", "0",
                                 str(span), seq_col, "*"] + tags))

    n_rec = 0
    for rid in range(n_reads):
        L = int(min(read_len, S - 50))
        L = max(20, int(L * rng.uniform(0.6, 1.0)))
        rev = rng.random() < rev_frac
        t0 = int(rng.integers(0, S - L))
        start = t0 + (S + 1 if rev else 0)
        core, ops, lead_trim = mutate(both[start:start + L])
        clip_l = int(rng.integers(1, max_clip)) if rng.random() < clip_frac else 0
        clip_r = int(rng.integers(1, max_clip)) if rng.random() < clip_frac else 0
        read = "".join(acgt[int(x)] for x in rng.integers(0, 4, clip_l)) + core + "".join(acgt[int(x)] for x in rng.integers(0, 4, clip_r))
        name = f"r{rid}"
        reads.append(read); names.append(name)
        emit(rid, name, read, clip_l, clip_r, ops, start + lead_trim)
        n_rec += 1
        if rng.random() < secondary:
            # a wrong-place candidate with the same edit script shape (low probability)
            span = sum(1 for o in ops if o != "I")
            t1 = int(rng.integers(0, S - span - 1))
            emit(rid, name, read, clip_l, clip_r, ops, t1 + (S + 1 if rng.random() < 0.5 else 0))
            n_rec += 1
    return PacbioSam(reads, names, "\n".join(lines) + "\n", n_rec)


def all_subwalks_for_pacbio(g: Graph, walk: list[int], max_read_len: int) -> list[list[int]]:
    """Every sub-walk the PacBio scorer looks up for ``walk`` (reference graph.cc:2438-2454)."""
    node_len = [g.node_len(x) if x >= 0 else -x for x in walk]
    ends = np.cumsum(node_len)
    begins = ends - np.array(node_len)
    out = []
    for i in range(len(walk)):
        for j in range(i, len(walk)):
            out.append(walk[i:j + 1])
            if (ends[j] - begins[i]) - (ends[i] - begins[i]) > max_read_len:
                break
    return out


@dataclass
class Workload:
    name: str
    genome_len: int
    n_pairs: int
    read_len: int = 150
    insert_mean: float = 300.0
    insert_std: float = 30.0
    err: float = 0.01
    seed: int = 20260301
    repeat_frac: float = 0.0   # > 0: collapsed repeats (make_repeat_graph): that share of the genome in 5-copy families of 1-3 kbp

    def build(self):
        """(genome sequence, graph) of the workload."""
        if self.repeat_frac > 0:
            return make_repeat_graph(self.genome_len, self.seed, self.repeat_frac)
        genome = make_genome(self.genome_len, self.seed)
        return genome, make_graph(genome, cut_lengths(self.genome_len, self.seed))


# BASELINE.json configs (SURVEY.md 8d sizes)
WORKLOADS = {
    "cfg2": Workload("cfg2: 1 Mbp, 30x 2x150 paired, insert 300+-30", 1_000_000, 100_000),
    "cfg3": Workload("cfg3: 5 Mbp, 50x 2x150 paired, insert 300+-30", 5_000_000, 833_333),
    "tiny": Workload("tiny: 60 kbp, 2x150 paired", 60_000, 3_000),
    # "optionally with planted repeats" -- config 3's recipe with 2 % of the genome in collapsed 5-copy repeat
    # families of 1-3 kbp: the true walk visits those nodes five times (windows that occur several times: general path)
    "cfg3r": Workload("cfg3r: 5 Mbp + 2 % collapsed 5-copy repeats of 1-3 kbp, 50x 2x150 paired, insert 300+-30", 5_000_000, 850_000, repeat_frac=0.02),
    "tinyr": Workload("tinyr: 120 kbp + 5 % collapsed repeats, 2x150 paired", 120_000, 20_000, repeat_frac=0.05),
    # the reference's own example configuration (example.cfg:20-29) has a JUMPING library next to the short-insert one: insert
    # 3700 +- 350, penalty_constant 0.00013, penalty_step 3000, min_prob_start -80 (its min_prob_per_base=0 is silently
    # ignored for paired sets -- the reader looks for "min_prob_pre_base", gaml.cc:855 -- so -0.7 applies). cfg3's genome and
    # graph, 2x150 reads at 20x: both mates of a pair rarely share a 2-8 kbp node window, the coverage sweep runs every call
    "cfg3j": Workload("cfg3j: 5 Mbp, 20x 2x150 paired, JUMPING library insert 3700+-350 (example.cfg:20-29)", 5_000_000, 333_333, insert_mean=3700.0, insert_std=350.0),
    # not a BASELINE config: the cfg3 recipe at 8x the size, to see where the scoring kernel's bandwidth levels off
    "cfg3x8": Workload("cfg3x8: 40 Mbp, 50x 2x150 paired, insert 300+-30", 40_000_000, 6_666_664),
}


def sa_move(rng, paths, g):
    """One random edit of the kind GAML's move generators make (moves.cc: BreakPath, ExtendPaths, LocalChange, ...):
    the CalcProb call pattern of the annealing loop (gaml.cc:148-339), not the optimiser itself."""
    return sa_move_kind(rng, paths, g)[0]


def sa_move_kind(rng, paths, g):
    """sa_move plus the kind of edit it drew (0 BreakPath, 1 join, 2 reverse, 3 LocalChange, 4 duplicate, 5 trim): the
    annealing loop accepts a worse assembly only after BreakPath (gaml.cc:286, 304-311)."""
    paths = [list(p) for p in paths]
    kind = rng.integers(0, 6)
    i = int(rng.integers(0, len(paths)))
    p = paths[i]
    if kind == 0 and len(p) > 3:  # BreakPath
        c = int(rng.integers(1, len(p) - 1))
        paths[i:i + 1] = [p[:c], p[c:]]
    elif kind == 1 and len(paths) > 1:  # join two paths, sometimes over a gap (ExtendPaths / FixGapLength)
        j = int(rng.integers(0, len(paths)))
        if j != i:
            gap = [-int(rng.integers(20, 400))] if rng.random() < 0.5 else []
            q = paths[j]
            paths[i] = p + gap + q
            del paths[j]
    elif kind == 2:  # reverse a path (the same sequence on the other strand)
        paths[i] = [x ^ 1 if x >= 0 else x for x in reversed(p)]
    elif kind == 3 and len(p) > 6:  # LocalChange: drop a short stretch, bridge it by a gap
        a = int(rng.integers(1, len(p) - 4))
        b = a + int(rng.integers(1, 3))
        removed = sum(g.node_len(x) if x >= 0 else -x for x in p[a:b])
        paths[i] = p[:a] + [-max(1, removed)] + p[b:]
    elif kind == 4 and len(p) > 4:  # duplicate a node (repeat resolution attempts)
        a = int(rng.integers(1, len(p) - 1))
        if p[a] >= 0:
            paths[i] = p[:a] + [p[a]] + p[a:]
    elif kind == 5 and len(p) > 2:  # trim an end
        paths[i] = p[1:] if rng.random() < 0.5 else p[:-1]
    return [q for q in paths if q], int(kind)


def sa_sequence(g, iters: int, seed: int = 7, threshold: int = 500):
    """BASELINE config 5's call pattern: the reference's start state (every long node a one-node walk,
    gaml.cc:1002-1005) and `iters` edited path sets, 60 % of the edits accepted."""
    start, seq, _ = sa_sequence_parents(g, iters, seed, threshold)
    return start, seq


def sa_sequence_parents(g, iters: int, seed: int = 7, threshold: int = 500):
    """sa_sequence plus, per edited path set, the index of the set it was derived from (-1: the start state) -- the
    "current" assembly whose likelihood the annealing loop compares the new one with (gaml.cc:286)."""
    walk = genome_walk(g)
    start = [[x] for x in walk if g.node_len(x) > threshold]
    rng = np.random.default_rng(seed)
    seq, parent, cur, cur_idx = [], [], start, -1
    for k in range(iters):
        new = sa_move(rng, cur, g)
        seq.append(new)
        parent.append(cur_idx)
        if rng.random() < 0.6:
            cur, cur_idx = new, k
    return start, seq, parent


def flatten_paths(paths: list[list[int]]):
    flat = np.array([x for p in paths for x in p], dtype=np.int32)
    offs = np.zeros(len(paths) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(p) for p in paths])
    return flat, offs


def ensure_dir(d: str) -> str:
    os.makedirs(d, exist_ok=True)
    return d
