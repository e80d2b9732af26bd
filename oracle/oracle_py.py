"""ctypes wrapper over oracle/_build/libgaml_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (gaml_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libgaml_oracle.so")
_REF = os.path.join(_HERE, "_ref", "libref_logdouble.so")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> None:
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def _load():
    build()
    L = C.CDLL(_LIB)
    L.orc_new.restype = C.c_void_p
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_load_graph.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_set_graph.argtypes = [C.c_void_p, C.c_int, _u8p, _i64p]
    L.orc_num_nodes.argtypes = [C.c_void_p]
    L.orc_node_len.argtypes = [C.c_void_p, C.c_int]
    L.orc_normalize_node.argtypes = [C.c_void_p, C.c_int]
    L.orc_add_single_fastq.argtypes = [C.c_void_p, C.c_char_p, C.c_double, _f64p]
    L.orc_add_paired_fastq.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_double, _f64p]
    L.orc_add_single.argtypes = [C.c_void_p, C.c_int, _u8p, _i64p, C.c_double, _f64p]
    L.orc_add_paired.argtypes = [C.c_void_p, C.c_int, _u8p, _i64p, _u8p, _i64p, C.c_double, _f64p]
    L.orc_add_pacbio.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_double, _f64p]
    L.orc_pacbio_put.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int, _i32p, _f64p, C.c_int]
    L.orc_add_pacbio_reads.argtypes = [C.c_void_p, C.c_int, _u8p, _i64p, C.c_char_p, C.c_double, _f64p]
    L.orc_pacbio_ingest_sam.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int, C.c_char_p]
    L.orc_pacbio_records.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int, _i32p, _f64p, C.c_int]
    L.orc_pacbio_keys.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_long]
    L.orc_pacbio_keys.restype = C.c_long
    L.orc_pacbio_misses.argtypes = [C.c_void_p, C.c_int]
    L.orc_pacbio_misses.restype = C.c_long
    L.orc_num_sets.argtypes = [C.c_void_p]
    L.orc_set_kind.argtypes = [C.c_void_p, C.c_int]
    L.orc_set_reads.argtypes = [C.c_void_p, C.c_int]
    L.orc_calc_prob.argtypes = [C.c_void_p, _i32p, _i64p, C.c_int, C.c_int, _i32p, _i32p]
    L.orc_calc_prob.restype = C.c_double
    L.orc_paired_probs.argtypes = [C.c_void_p, C.c_int, _f64p, _i32p]
    L.orc_single_detail.argtypes = [C.c_void_p, C.c_int, _i32p, _i64p, C.c_int, _f64p, _i32p]
    L.orc_single_detail.restype = C.c_double
    L.orc_paired_slow.argtypes = [C.c_void_p, C.c_int, _i32p, _i64p, C.c_int, _f64p, _i32p, np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")]
    L.orc_paired_slow.restype = C.c_double
    L.orc_pacbio_detail.argtypes = [C.c_void_p, C.c_int, _i32p, _i64p, C.c_int, _f64p, _i32p]
    L.orc_pacbio_detail.restype = C.c_double
    L.orc_window_count.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_windows_aligned.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_windows_aligned.restype = C.c_long
    L.orc_window_records.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, _i32p, C.c_int]
    L.orc_align_window.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int]
    L.orc_window_keys.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_long]
    L.orc_window_keys.restype = C.c_long
    L.orc_window_string.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_char_p, C.c_int, _i32p]
    L.orc_positions_only_path.argtypes = [C.c_void_p, C.c_int, C.c_int, _i32p, C.c_int, C.c_int, _i32p, C.c_int]
    L.orc_score_path_paired.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int, _f64p, _i32p, _i64p]
    L.orc_extend_hit.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_char_p, _i32p]
    L.orc_insert_prob.argtypes = [C.c_double] * 3
    L.orc_insert_prob.restype = C.c_double
    L.orc_max_hash.argtypes = [C.c_char_p]
    L.orc_max_hash.restype = C.c_uint64
    L.orc_window_hashes.argtypes = [C.c_char_p, C.c_int, _u64p, _i32p, C.c_int]
    for f in ("orc_ld_add", "orc_ld_mul", "orc_ld_pow", "orc_ld_div"):
        getattr(L, f).argtypes = [C.c_double, C.c_double]
        getattr(L, f).restype = C.c_double
    L.orc_ld_from_linear.argtypes = [C.c_double]
    L.orc_ld_from_linear.restype = C.c_double
    L.orc_invert_walk.argtypes = [_i32p, C.c_int, _i32p]
    L.orc_sam_alignment_logprob.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_double, C.c_int, _i32p]
    L.orc_sam_alignment_logprob.restype = C.c_double
    L.orc_sam_band.argtypes = [C.c_char_p, C.c_int, _i32p, _i32p, _i32p, _i32p, C.c_int]
    L.orc_load_config.argtypes = [C.c_void_p, C.c_char_p]
    L.orc_config_order.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    L.orc_config_paired_values.argtypes = [C.c_char_p, C.c_char_p, _f64p]
    return L


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def ref_lib():
    """The reference's own logdouble.hpp / utility.h compiled in place (oracle/_ref)."""
    if not os.path.exists(_REF):
        return None
    R = C.CDLL(_REF)
    for f in ("ref_ld_add", "ref_ld_add_assign", "ref_ld_mul", "ref_ld_mul_assign", "ref_ld_pow", "ref_ld_div"):
        getattr(R, f).argtypes = [C.c_double, C.c_double]
        getattr(R, f).restype = C.c_double
    R.ref_ld_from_linear.argtypes = [C.c_double]
    R.ref_ld_from_linear.restype = C.c_double
    R.ref_ld_default.restype = C.c_double
    R.ref_ld_lt.argtypes = [C.c_double, C.c_double]
    R.ref_ld_gt.argtypes = [C.c_double, C.c_double]
    R.ref_invert_path.argtypes = [_i32p, C.c_int, _i32p]
    R.ref_reverse_path.argtypes = [_i32p, C.c_int]
    return R


def _flat(paths):
    flat = np.array([x for p in paths for x in p], dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    offs = np.zeros(len(paths) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(p) for p in paths])
    return flat, offs


def single_cfg(penalty_constant=0.0, step=50.0, min_prob_per_base=-0.7, min_prob_start=-10.0, weight=1.0):
    return np.array([penalty_constant, step, min_prob_per_base, min_prob_start, weight], dtype=np.float64)


def paired_cfg(insert_mean, insert_std, penalty_constant=0.0, penalty_step=50.0, min_prob_per_base=-0.7,
               min_prob_start=-10.0, weight=1.0):
    # step = insert_mean - penalty_step (reference gaml.cc:860)
    return np.array([penalty_constant, insert_mean - penalty_step, insert_mean, insert_std, min_prob_per_base,
                     min_prob_start, weight], dtype=np.float64)


class Oracle:
    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_new())

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- set-up
    def load_graph(self, path):
        n = self.L.orc_load_graph(self.h, path.encode())
        if n < 0:
            raise IOError(path)
        return n

    def set_graph(self, bases, offs):
        return self.L.orc_set_graph(self.h, len(offs) - 1, bases, offs)

    def load_config(self, path):
        n = self.L.orc_load_config(self.h, path.encode())
        if n < 0:
            raise IOError(f"orc_load_config({path}) -> {n}")
        return n

    def add_single(self, bases, offs, mismatch, cfg):
        return self.L.orc_add_single(self.h, len(offs) - 1, bases, offs, mismatch, cfg)

    def add_paired(self, b1, o1, b2, o2, mismatch, cfg):
        return self.L.orc_add_paired(self.h, len(o1) - 1, b1, o1, b2, o2, mismatch, cfg)

    def add_single_fastq(self, f, mismatch, cfg):
        r = self.L.orc_add_single_fastq(self.h, f.encode(), mismatch, cfg)
        if r < 0:
            raise IOError(f)
        return r

    def add_paired_fastq(self, f1, f2, mismatch, cfg):
        r = self.L.orc_add_paired_fastq(self.h, f1.encode(), f2.encode(), mismatch, cfg)
        if r < 0:
            raise IOError((f1, f2, r))
        return r

    def add_pacbio(self, lens, mismatch, cfg):
        lens = np.ascontiguousarray(lens, np.int32)
        return self.L.orc_add_pacbio(self.h, len(lens), lens, mismatch, cfg)

    def pacbio_put(self, rs, walk, rec3, logp):
        walk = np.ascontiguousarray(walk, np.int32)
        rec3 = np.ascontiguousarray(rec3, np.int32).reshape(-1, 3)
        logp = np.ascontiguousarray(logp, np.float64)
        return self.L.orc_pacbio_put(self.h, rs, walk, len(walk), rec3.reshape(-1) if rec3.size else np.zeros(3, np.int32),
                                     logp if logp.size else np.zeros(1), len(logp))

    def add_pacbio_reads(self, bases, offs, names, mismatch, cfg):
        return self.L.orc_add_pacbio_reads(self.h, len(offs) - 1, bases, offs, ("\n".join(names) + "\n").encode(), mismatch, cfg)

    def pacbio_ingest_sam(self, rs, path, sam_text: str):
        path = np.ascontiguousarray(path, np.int32)
        return self.L.orc_pacbio_ingest_sam(self.h, rs, path, len(path), sam_text.encode())

    def pacbio_keys(self, rs):
        need = self.L.orc_pacbio_keys(self.h, rs, np.zeros(1, np.int32), 0)
        buf = np.zeros(max(1, need), np.int32)
        self.L.orc_pacbio_keys(self.h, rs, buf, need)
        keys, i = [], 0
        while i < need:
            n = int(buf[i]); keys.append(tuple(int(x) for x in buf[i + 1:i + 1 + n])); i += 1 + n
        return keys

    def pacbio_records(self, rs, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        n = self.L.orc_pacbio_records(self.h, rs, walk, len(walk), np.zeros(3, np.int32), np.zeros(1), 0)
        if n < 0:
            return None
        rec = np.zeros(3 * max(1, n), np.int32); lp = np.zeros(max(1, n))
        self.L.orc_pacbio_records(self.h, rs, walk, len(walk), rec, lp, n)
        return rec.reshape(-1, 3)[:n], lp[:n]

    def pacbio_misses(self, rs):
        return self.L.orc_pacbio_misses(self.h, rs)

    # ---- scoring
    def num_sets(self):
        return self.L.orc_num_sets(self.h)

    def set_reads(self, rs):
        return self.L.orc_set_reads(self.h, rs)

    def set_kind(self, rs):
        return self.L.orc_set_kind(self.h, rs)

    def calc_prob(self, paths, fresh=True):
        flat, offs = _flat(paths)
        zeros = np.zeros(2 * max(1, self.num_sets()), np.int32)
        tl = np.zeros(1, np.int32)
        v = self.L.orc_calc_prob(self.h, flat, offs, len(paths), 1 if fresh else 0, zeros, tl)
        return v, zeros.reshape(-1, 2)[: self.num_sets()].copy(), int(tl[0])

    def paired_probs(self, rs):
        out = np.zeros(self.set_reads(rs), np.float64)
        bb = np.zeros(1, np.int32)
        self.L.orc_paired_probs(self.h, rs, out, bb)
        return out, int(bb[0])

    def paired_slow(self, rs, paths):
        """The reference's slow non-incremental paired scorer (graph.cc:1991-2127): value, per-read probabilities,
        {zero_reads, total_len, bad_bases}, and per read whether a mate had more than one distinct alignment."""
        flat, offs = _flat(paths)
        probs = np.zeros(self.set_reads(rs), np.float64)
        o3 = np.zeros(3, np.int32)
        several = np.zeros(self.set_reads(rs), np.uint8)
        v = self.L.orc_paired_slow(self.h, rs, flat, offs, len(paths), probs, o3, several)
        return v, probs, o3, several

    def single_detail(self, rs, paths):
        flat, offs = _flat(paths)
        probs = np.zeros(self.set_reads(rs), np.float64)
        o3 = np.zeros(3, np.int32)
        v = self.L.orc_single_detail(self.h, rs, flat, offs, len(paths), probs, o3)
        return v, probs, o3

    def pacbio_detail(self, rs, paths):
        flat, offs = _flat(paths)
        lp = np.zeros(self.set_reads(rs), np.float64)
        o3 = np.zeros(3, np.int32)
        v = self.L.orc_pacbio_detail(self.h, rs, flat, offs, len(paths), lp, o3)
        return v, lp, o3

    def score_path_paired(self, rs, path):
        path = np.ascontiguousarray(path, np.int32)
        probs = np.zeros(self.set_reads(rs), np.float64)
        bb = np.zeros(1, np.int32)
        nt = np.zeros(1, np.int64)
        self.L.orc_score_path_paired(self.h, rs, path, len(path), probs, bb, nt)
        return probs, int(bb[0]), int(nt[0])

    # ---- stage access
    def window_records(self, rs, mate, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        n = self.L.orc_window_records(self.h, rs, mate, walk, len(walk), np.zeros(4, np.int32), 0)
        if n < 0:
            return None
        out = np.zeros(4 * max(1, n), np.int32)
        self.L.orc_window_records(self.h, rs, mate, walk, len(walk), out, n)
        return out.reshape(-1, 4)[:n]

    def align_window(self, rs, mate, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        return self.L.orc_align_window(self.h, rs, mate, walk, len(walk))

    def window_keys(self, rs, mate):
        need = self.L.orc_window_keys(self.h, rs, mate, np.zeros(1, np.int32), 0)
        buf = np.zeros(max(1, need), np.int32)
        self.L.orc_window_keys(self.h, rs, mate, buf, need)
        keys, i = [], 0
        while i < need:
            n = int(buf[i])
            keys.append(tuple(int(x) for x in buf[i + 1:i + 1 + n]))
            i += 1 + n
        return keys

    def window_string(self, rs, mate, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        off = np.zeros(1, np.int32)
        n = self.L.orc_window_string(self.h, rs, mate, walk, len(walk), None, 0, off)
        buf = C.create_string_buffer(n + 1)
        self.L.orc_window_string(self.h, rs, mate, walk, len(walk), buf, n + 1, off)
        return buf.value.decode(), int(off[0])

    def positions_only_path(self, rs, mate, ctg, st=0):
        ctg = np.ascontiguousarray(ctg, np.int32)
        n = self.L.orc_positions_only_path(self.h, rs, mate, ctg, len(ctg), st, np.zeros(4, np.int32), 0)
        out = np.zeros(4 * max(1, n), np.int32)
        self.L.orc_positions_only_path(self.h, rs, mate, ctg, len(ctg), st, out, n)
        return out.reshape(-1, 4)[:n]

    def windows_aligned(self, rs, mate):
        return self.L.orc_windows_aligned(self.h, rs, mate)


def extend_hit(win_pos, read_pos, read: str, win: str):
    out = np.zeros(3, np.int32)
    lib().orc_extend_hit(win_pos, read_pos, read.encode(), win.encode(), out)
    return tuple(int(x) for x in out)


def window_hashes(seq: str, read_len: int):
    cap = max(1, len(seq))
    h = np.zeros(cap, np.uint64)
    p = np.zeros(cap, np.int32)
    n = lib().orc_window_hashes(seq.encode(), read_len, h, p, cap)
    return [(int(h[i]), int(p[i])) for i in range(n)]


def sam_band(sam_line: str, total_len: int):
    """(fields dict, row0, lo[], hi[]) of one SAM line as the oracle parses / bands it."""
    f = np.zeros(10, np.int32); r0 = np.zeros(1, np.int32)
    n = lib().orc_sam_band(sam_line.encode(), total_len, f, r0, np.zeros(1, np.int32), np.zeros(1, np.int32), 0)
    lo = np.zeros(n, np.int32); hi = np.zeros(n, np.int32)
    lib().orc_sam_band(sam_line.encode(), total_len, f, r0, lo, hi, n)
    keys = ["flags", "len", "posstart", "posend", "sstart", "send", "slen", "tstart", "tend", "edit_dist"]
    return dict(zip(keys, (int(x) for x in f))), int(r0[0]), lo, hi


def sam_alignment_logprob(sam_line: str, target_all: str, read: str, mismatch: float, band: int = 2):
    return float(lib().orc_sam_alignment_logprob(sam_line.encode(), target_all.encode(), read.encode(), mismatch, band, np.zeros(2, np.int32)))
