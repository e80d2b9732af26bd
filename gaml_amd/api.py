"""ctypes binding of libgaml_hip.so (C ABI in include/gaml_hip.h).

The library is the product: there is no Python or CPU scoring path behind this module. If
the shared object is missing, import fails; if no HIP device is present, every scoring call
raises GamlHipError(ENODEVICE).
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two flavours of the library (gaml_amd/csrc/Makefile): libgaml_hip.so, the product, and libgaml_hip_dev.so, the same
# sources with the gaml_hip_debug_* entry points, A/B knobs and in-kernel time stamps compiled in. The product is what
# gets loaded unless GAML_HIP_FLAVOUR=dev (tests/conftest.py and tools/ set it) or GAML_HIP_LIB names a file (A/B builds).
FLAVOUR = os.environ.get("GAML_HIP_FLAVOUR", "release")
LIB_PATH = os.environ.get("GAML_HIP_LIB") or os.path.join(_HERE, "libgaml_hip_dev.so" if FLAVOUR == "dev" else "libgaml_hip.so")

OK, EINVAL, ENODEVICE, EHIP, ESTATE = 0, -1, -2, -3, -4

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")

ALIGMENT = np.dtype([("position", np.int32), ("edit_dist", np.int32), ("read_id", np.int32), ("orientation", np.int32)])
PACBIO_ALIGMENT = np.dtype([("position", np.int32), ("position_end", np.int32), ("read_id", np.int32), ("pad_", np.int32),
                            ("logprob", np.float64)])


class SingleCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("penalty_constant", "step", "min_prob_per_base", "min_prob_start", "weight", "mismatch_prob")]


class PairedCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("penalty_constant", "step", "insert_mean", "insert_std", "min_prob_per_base", "min_prob_start", "weight",
                 "mismatch_prob")]


def single_cfg(penalty_constant=0.0, penalty_step=50.0, min_prob_per_base=-0.7, min_prob_start=-10.0, weight=1.0,
               mismatch_prob=0.01) -> SingleCfg:
    """Defaults of the reference's config reader (gaml.cc:812-819)."""
    return SingleCfg(penalty_constant, penalty_step, min_prob_per_base, min_prob_start, weight, mismatch_prob)


def paired_cfg(insert_mean, insert_std, penalty_constant=0.0, penalty_step=50.0, min_prob_per_base=-0.7,
               min_prob_start=-10.0, weight=1.0, mismatch_prob=0.01) -> PairedCfg:
    """Defaults of the reference's config reader (gaml.cc:851-862); step = insert_mean - penalty_step."""
    return PairedCfg(penalty_constant, insert_mean - penalty_step, insert_mean, insert_std, min_prob_per_base,
                     min_prob_start, weight, mismatch_prob)


class BatchPaths:
    """Several path sets in the ABI's concatenated form (built once by callers that re-score them)."""

    def __init__(self, path_sets):
        allp = [p for ps in path_sets for p in ps]
        self.flat = np.array([x for p in allp for x in p], dtype=np.int32) if allp else np.zeros(0, np.int32)
        if self.flat.size == 0:
            self.flat = np.zeros(1, np.int32)
        self.offs = np.zeros(len(allp) + 1, np.int64)
        self.offs[1:] = np.cumsum([len(p) for p in allp])
        self.set_offs = np.zeros(len(path_sets) + 1, np.int32)
        self.set_offs[1:] = np.cumsum([len(ps) for ps in path_sets])
        self.n_sets = len(path_sets)


class GamlHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gaml_hip error {code}: {msg}")
        self.code = code


def _bind_to_torch_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel bundles its own libamdhip64 / libhsa-runtime64;
    libgaml_hip.so links the system ones.  Loaded after torch, the dynamic linker resolves our
    dependency to torch's copy (same SONAME) and all is well; loaded BEFORE torch, the process ends
    up with two runtimes and the second one to initialise finds no device.  So when PyTorch is
    installed, load its runtime first (without importing torch) and let ours bind to it.
    GAML_HIP_SYSTEM_RUNTIME=1 skips this."""
    if os.environ.get("GAML_HIP_SYSTEM_RUNTIME") == "1" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def _load():
    _bind_to_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no fallback implementation)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.gaml_hip_create.argtypes = [C.POINTER(vp), C.c_int]
    L.gaml_hip_create_multi.argtypes = [C.POINTER(vp), _i32p, C.c_int32]
    L.gaml_hip_create_from_env.argtypes = [C.POINTER(vp)]
    L.gaml_hip_num_shards.argtypes = [vp]
    L.gaml_hip_set_exchange.argtypes = [vp, C.c_int32]
    L.gaml_hip_get_exchange.argtypes = [vp]
    L.gaml_hip_comm_unique_id.argtypes = [vp]
    L.gaml_hip_comm_init_rank.argtypes = [vp, vp, C.c_int32, C.c_int32]
    L.gaml_hip_destroy.argtypes = [vp]
    L.gaml_hip_destroy.restype = None
    L.gaml_hip_last_error.argtypes = [vp]
    L.gaml_hip_last_error.restype = C.c_char_p
    L.gaml_hip_version.restype = C.c_char_p
    L.gaml_hip_set_graph.argtypes = [vp, C.c_int32, _u8p, _i64p]
    L.gaml_hip_load_graph.argtypes = [vp, C.c_char_p]
    L.gaml_hip_add_single.argtypes = [vp, C.POINTER(SingleCfg), C.c_int32, _u8p, _i64p]
    L.gaml_hip_add_paired.argtypes = [vp, C.POINTER(PairedCfg), C.c_int32, _u8p, _i64p, _u8p, _i64p]
    L.gaml_hip_add_pacbio.argtypes = [vp, C.POINTER(SingleCfg), C.c_int32, _i32p]
    L.gaml_hip_add_single_fastq.argtypes = [vp, C.POINTER(SingleCfg), C.c_char_p]
    L.gaml_hip_add_paired_fastq.argtypes = [vp, C.POINTER(PairedCfg), C.c_char_p, C.c_char_p]
    L.gaml_hip_add_pacbio_fastq.argtypes = [vp, C.POINTER(SingleCfg), C.c_char_p]
    L.gaml_hip_set_shard.argtypes = [vp, C.c_int32, C.c_int32]
    L.gaml_hip_set_presharded.argtypes = [vp, C.c_int32]
    L.gaml_hip_put_window_records.argtypes = [vp, C.c_int, C.c_int, _i32p, C.c_int32, vp, C.c_int64]
    L.gaml_hip_put_pacbio_records.argtypes = [vp, C.c_int, _i32p, C.c_int32, vp, C.c_int64]
    L.gaml_hip_add_pacbio_reads.argtypes = [vp, C.POINTER(SingleCfg), C.c_int32, _u8p, _i64p, C.c_char_p]
    L.gaml_hip_pacbio_missing.argtypes = [vp, C.c_int, _i32p, C.c_int32, _i32p, C.c_int32]
    L.gaml_hip_pacbio_missing.restype = C.c_int32
    L.gaml_hip_pacbio_ingest_sam.argtypes = [vp, C.c_int, _i32p, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]
    L.gaml_hip_pacbio_records.argtypes = [vp, C.c_int, _i32p, C.c_int32, vp, C.c_int64]
    L.gaml_hip_pacbio_records.restype = C.c_int64
    L.gaml_hip_pacbio_dp_stats.argtypes = [vp, C.c_int, _f64p]
    if hasattr(L, "gaml_hip_debug_sam_logprob"):  # development build only
        L.gaml_hip_debug_sam_logprob.argtypes = [vp, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int64, C.c_double, C.POINTER(C.c_double), vp, vp, C.c_int32]
    if hasattr(L, "gaml_hip_debug_sam_shape"):  # development build only
        L.gaml_hip_debug_sam_shape.argtypes = [C.c_char_p, C.c_int64, C.c_int32, _i32p, C.c_void_p, C.c_int32]
    if hasattr(L, "gaml_hip_debug_sam_band"):  # development build only
        L.gaml_hip_debug_sam_band.argtypes = [C.c_char_p, C.c_int64, C.c_int32, _i32p, _i32p, _i32p, _i32p, C.c_int32]
        L.gaml_hip_debug_sam_band.restype = C.c_int32
    L.gaml_hip_calc_prob.argtypes = [vp, _i32p, _i64p, C.c_int32, C.POINTER(C.c_double), _i32p, C.POINTER(C.c_int32)]
    L.gaml_hip_calc_prob_batch.argtypes = [vp, C.c_int32, _i32p, _i64p, _i32p, _f64p, vp, vp]
    L.gaml_hip_calc_partials.argtypes = [vp, _i32p, _i64p, C.c_int32, _f64p, C.POINTER(C.c_int32)]
    L.gaml_hip_combine_partials.argtypes = [vp, _f64p, C.c_int32, C.POINTER(C.c_double), _i32p]
    L.gaml_hip_calc_partials_async.argtypes = [vp, _i32p, _i64p, C.c_int32, vp, vp, C.POINTER(C.c_int32)]
    L.gaml_hip_eval_begin.argtypes = [vp, _i32p, _i64p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    L.gaml_hip_eval_pending_maxpos.argtypes = [vp, vp, C.c_int64]
    L.gaml_hip_eval_pending_maxpos.restype = C.c_int64
    L.gaml_hip_eval_apply_maxpos.argtypes = [vp, _i32p, C.c_int64]
    L.gaml_hip_eval_finish.argtypes = [vp, _f64p]
    L.gaml_hip_eval_finish_async.argtypes = [vp, vp, vp]
    L.gaml_hip_sync.argtypes = [vp]
    L.gaml_hip_compact_tables.argtypes = [vp]
    L.gaml_hip_eval_score_async.argtypes = [vp, vp, vp]
    L.gaml_hip_eval_score_async.restype = C.c_int32
    L.gaml_hip_eval_coverage_export_async.argtypes = [vp, C.c_int32, vp, C.c_int64, C.POINTER(C.c_int64), vp]
    L.gaml_hip_eval_coverage_finish_async.argtypes = [vp, C.c_int32, vp, C.c_int32, C.c_int32, vp]
    L.gaml_hip_eval_pacbio_pending.argtypes = [vp]
    L.gaml_hip_eval_pacbio_pending.restype = C.c_int32
    L.gaml_hip_eval_pacbio_intervals.argtypes = [vp, C.c_int32]
    L.gaml_hip_eval_pacbio_intervals.restype = C.c_int64
    L.gaml_hip_eval_pacbio_export_async.argtypes = [vp, C.c_int32, vp, C.c_int64, vp]
    L.gaml_hip_eval_pacbio_finish_async.argtypes = [vp, C.c_int32, vp, C.c_int64, C.c_int32, vp]
    L.gaml_hip_num_readsets.argtypes = [vp]
    L.gaml_hip_readset_kind.argtypes = [vp, C.c_int]
    L.gaml_hip_readset_reads.argtypes = [vp, C.c_int]
    L.gaml_hip_readset_reads.restype = C.c_int64
    L.gaml_hip_num_nodes.argtypes = [vp]
    L.gaml_hip_node_len.argtypes = [vp, C.c_int32]
    L.gaml_hip_read_probs.argtypes = [vp, C.c_int, _f64p, C.c_int64]
    L.gaml_hip_bad_bases.argtypes = [vp, C.c_int, C.POINTER(C.c_int64)]
    L.gaml_hip_window_count.argtypes = [vp, C.c_int, C.c_int]
    L.gaml_hip_window_count.restype = C.c_int64
    L.gaml_hip_window_records.argtypes = [vp, C.c_int, C.c_int, _i32p, C.c_int32, vp, C.c_int64]
    L.gaml_hip_window_records.restype = C.c_int64
    L.gaml_hip_align_window.argtypes = [vp, C.c_int, C.c_int, _i32p, C.c_int32]
    L.gaml_hip_align_window.restype = C.c_int64
    if hasattr(L, "gaml_hip_debug_prepare"):  # development build only
        L.gaml_hip_debug_prepare.argtypes = [vp, _i32p, _i64p, C.c_int32]
    if hasattr(L, "gaml_hip_debug_occurrences"):  # development build only
        L.gaml_hip_debug_occurrences.argtypes = [vp, C.c_int, C.c_int, _i32p, C.c_int64]
    if hasattr(L, "gaml_hip_debug_occurrences"):  # development build only
        L.gaml_hip_debug_occurrences.restype = C.c_int64
    if hasattr(L, "gaml_hip_debug_table_occurrences"):  # development build only
        L.gaml_hip_debug_table_occurrences.argtypes = [vp, C.c_int, C.c_int, _i32p, C.c_int64, _i64p]
    if hasattr(L, "gaml_hip_debug_table_occurrences"):  # development build only
        L.gaml_hip_debug_table_occurrences.restype = C.c_int64
    if hasattr(L, "gaml_hip_debug_window_walk"):  # development build only
        L.gaml_hip_debug_window_walk.argtypes = [vp, C.c_int, C.c_int, C.c_int32, _i32p, C.c_int32]
    for new, old in (("gaml_hip_pair_classes", "gaml_hip_debug_class_counts"), ("gaml_hip_last_phases", "gaml_hip_debug_profile"),
                     ("gaml_hip_table_stats", "gaml_hip_debug_table_stats")):
        if not hasattr(L, new) and hasattr(L, old):  # an older A/B build loaded through GAML_HIP_LIB (tools/)
            setattr(L, new, getattr(L, old))
    L.gaml_hip_pair_classes.argtypes = [vp, C.c_int, _i64p]
    if hasattr(L, "gaml_hip_debug_fold_check"):  # development build only
        L.gaml_hip_debug_fold_check.argtypes = [vp, C.c_int, _i64p]
    if hasattr(L, "gaml_hip_debug_radix_sort"):  # development build only
        L.gaml_hip_debug_radix_sort.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    if hasattr(L, "gaml_hip_debug_tables_check"):
        L.gaml_hip_debug_tables_check.argtypes = [vp, C.c_int, _i64p]
    if hasattr(L, "gaml_hip_debug_static_check"):  # absent from older A/B builds loaded through GAML_HIP_LIB
        L.gaml_hip_debug_static_check.argtypes = [vp, C.c_int, _i64p]
    if hasattr(L, "gaml_hip_debug_set_knob"):  # development build only
        L.gaml_hip_debug_set_knob.argtypes = [vp, C.c_int, C.c_int]
    if hasattr(L, "gaml_hip_shm_exchange_open"):
        L.gaml_hip_shm_exchange_open.argtypes = [vp, C.c_char_p, C.c_int32, C.c_int32, C.c_int32]
        L.gaml_hip_shm_allreduce_sum.argtypes = [vp, C.c_void_p, C.c_int32]
        L.gaml_hip_shm_exchange_close.argtypes = [vp, C.c_int32]
    if hasattr(L, "gaml_hip_fetch_async"):
        L.gaml_hip_fetch_async.argtypes = [vp, C.c_void_p, C.c_int32, C.c_void_p]
        L.gaml_hip_fetch_wait.argtypes = [vp, C.c_void_p, C.c_int32]
    if hasattr(L, "gaml_hip_debug_timeline"):  # absent from older A/B builds loaded through GAML_HIP_LIB
        L.gaml_hip_debug_timeline.argtypes = [vp, C.c_int, C.c_void_p, C.c_int64]
    L.gaml_hip_last_phases.argtypes = [vp, _f64p]
    L.gaml_hip_table_stats.argtypes = [vp, C.c_int, _i64p]
    L.gaml_hip_aligner_stats.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.gaml_hip_aligner_stages.argtypes = [vp, _f64p]
    L.gaml_hip_last_timing.argtypes = [vp, _f64p]
    L.gaml_hip_set_event_timing.argtypes = [vp, C.c_int]
    L.gaml_hip_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    return L


_lib = _load()

# every symbol include/gaml_hip.h declares (checked by tests/test_abi.py against the header text)
def lib():
    return _lib


def version() -> str:
    return _lib.gaml_hip_version().decode()


class FlatPaths:
    """Paths already in the ABI's form (flattened int32 ids + int64 offsets): build once, score often."""

    def __init__(self, paths):
        self.flat, self.offs = _flat(paths)
        self.n = len(paths)
        self.flat_ptr, self.offs_ptr = self.flat.ctypes.data, self.offs.ctypes.data  # for Context.score

    def __len__(self):
        return self.n


def _flat(paths):
    if isinstance(paths, FlatPaths):
        return paths.flat, paths.offs
    flat = np.array([x for p in paths for x in p], dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, np.int32)
    offs = np.zeros(len(paths) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(p) for p in paths])
    return flat, offs


def comm_unique_id() -> bytes:
    """128 bytes that rank 0 sends to every other rank before Context.comm_init_rank (ncclGetUniqueId)."""
    buf = C.create_string_buffer(128)
    rc = _lib.gaml_hip_comm_unique_id(C.cast(buf, C.c_void_p))
    if rc != OK:
        raise GamlHipError(rc, "gaml_hip_comm_unique_id failed (RCCL not loadable?)")
    return buf.raw


class Context:
    """One graph + its read sets + their device state (one per process / GPU)."""

    def __init__(self, device: int = 0, rank: int = 0, world: int = 1, presharded: int = 1, devices=None):
        """devices: a list of HIP ordinals -> ONE context over that many device shards in this process
        (gaml_hip_create_multi); every method then acts on the whole read set."""
        self._h = C.c_void_p()
        if devices is not None:
            devs = np.ascontiguousarray(devices, np.int32)
            rc = _lib.gaml_hip_create_multi(C.byref(self._h), devs, len(devs))
            if rc != OK:
                raise GamlHipError(rc, f"gaml_hip_create_multi(devices={list(devs)}) failed")
            device = int(devs[0])
        else:
            rc = _lib.gaml_hip_create(C.byref(self._h), device)
            if rc != OK:
                raise GamlHipError(rc, f"gaml_hip_create(device={device}) failed")
        self.device = device
        if world != 1:
            self._check(_lib.gaml_hip_set_shard(self._h, rank, world))
        elif presharded != 1:
            self._check(_lib.gaml_hip_set_presharded(self._h, presharded))
        self.rank, self.world = rank, world
        self._fast = None
        self._fast_begin = None
        self._fast_combine = None

    def close(self):
        if self._h:
            _lib.gaml_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise GamlHipError(rc, _lib.gaml_hip_last_error(self._h).decode())
        return rc

    # ---- several GPUs
    def num_shards(self) -> int:
        return _lib.gaml_hip_num_shards(self._h)

    def set_exchange(self, mode: str):
        """'rccl' (all-gather of the shards' partials on their streams, summed in rank order by every shard: the default
        when every shard has its own GPU), 'rccl-allreduce' (ncclAllReduce(sum): RCCL's order of additions) or 'host'
        (pinned-host partials summed in rank order by the calling thread)."""
        self._check(_lib.gaml_hip_set_exchange(self._h, {"host": 0, "rccl": 1, "rccl-allreduce": 2}[mode]))

    def exchange(self) -> str:
        return {0: "host", 1: "rccl", 2: "rccl-allreduce"}[self._check(_lib.gaml_hip_get_exchange(self._h))]

    def last_error(self) -> str:
        return _lib.gaml_hip_last_error(self._h).decode()

    def comm_init_rank(self, comm_id: bytes, rank: int, world: int):
        """One process per GPU: join the library's RCCL communicator (id from comm_unique_id() of rank 0)."""
        buf = C.create_string_buffer(bytes(comm_id), 128)
        self._check(_lib.gaml_hip_comm_init_rank(self._h, C.cast(buf, C.c_void_p), rank, world))

    # ---- inputs
    def set_graph(self, bases: np.ndarray, offs: np.ndarray):
        self._check(_lib.gaml_hip_set_graph(self._h, len(offs) - 1, np.ascontiguousarray(bases, np.uint8),
                                            np.ascontiguousarray(offs, np.int64)))

    def load_graph(self, path: str):
        self._check(_lib.gaml_hip_load_graph(self._h, path.encode()))

    def add_single(self, cfg: SingleCfg, bases, offs) -> int:
        return self._check(_lib.gaml_hip_add_single(self._h, C.byref(cfg), len(offs) - 1, bases, offs))

    def add_paired(self, cfg: PairedCfg, b1, o1, b2, o2) -> int:
        return self._check(_lib.gaml_hip_add_paired(self._h, C.byref(cfg), len(o1) - 1, b1, o1, b2, o2))

    def add_pacbio(self, cfg: SingleCfg, lens) -> int:
        lens = np.ascontiguousarray(lens, np.int32)
        return self._check(_lib.gaml_hip_add_pacbio(self._h, C.byref(cfg), len(lens), lens))

    def add_pacbio_reads(self, cfg: SingleCfg, bases, offs, names) -> int:
        bases = np.ascontiguousarray(bases, np.uint8); offs = np.ascontiguousarray(offs, np.int64)
        joined = ("\n".join(names) + "\n").encode()
        return self._check(_lib.gaml_hip_add_pacbio_reads(self._h, C.byref(cfg), len(offs) - 1, bases, offs, joined))

    def pacbio_missing(self, rs, path):
        path = np.ascontiguousarray(path, np.int32)
        n = self._check(_lib.gaml_hip_pacbio_missing(self._h, rs, path, len(path), np.zeros(2, np.int32), 0))
        out = np.zeros(2 * max(1, n), np.int32)
        self._check(_lib.gaml_hip_pacbio_missing(self._h, rs, path, len(path), out, n))
        return [(int(out[2 * i]), int(out[2 * i + 1])) for i in range(n)]

    def pacbio_ingest_sam(self, rs, path, sam_text) -> int:
        """sam_text: the SAM file's text (str, or bytes as read from the file -- no copy then)"""
        path = np.ascontiguousarray(path, np.int32)
        raw = sam_text.encode() if isinstance(sam_text, str) else bytes(sam_text) if not isinstance(sam_text, bytes) else sam_text
        filed = C.c_int64(0)
        self._check(_lib.gaml_hip_pacbio_ingest_sam(self._h, rs, path, len(path), raw, len(raw), C.byref(filed)))
        return filed.value

    def pacbio_records(self, rs, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        n = _lib.gaml_hip_pacbio_records(self._h, rs, walk, len(walk), None, 0)
        if n == -1:
            return None
        self._check(int(n))
        recs = np.zeros(max(1, n), PACBIO_ALIGMENT)
        _lib.gaml_hip_pacbio_records(self._h, rs, walk, len(walk), recs.ctypes.data, n)
        return recs[:n]

    def debug_sam_logprob(self, target: str, read: str, sam_line: str, mismatch: float, with_band=False):
        t, r, l = target.encode(), read.encode(), sam_line.encode()
        out = C.c_double(0)
        n = self._check(_lib.gaml_hip_debug_sam_logprob(self._h, t, len(t), r, len(r), l, len(l), mismatch, C.byref(out), None, None, 0))
        if not with_band:
            return out.value
        lo = np.zeros(n, np.int32); hi = np.zeros(n, np.int32)
        self._check(_lib.gaml_hip_debug_sam_logprob(self._h, t, len(t), r, len(r), l, len(l), mismatch, C.byref(out), lo.ctypes.data,
                                                    hi.ctypes.data, n))
        return out.value, lo, hi

    def pacbio_dp_stats(self, rs):
        out = np.zeros(8)
        self._check(_lib.gaml_hip_pacbio_dp_stats(self._h, rs, out))
        keys = ["records", "jobs", "rows", "cells", "kernel_ms", "host_prepare_ms", "device_ms", "scratch_bytes"]
        return dict(zip(keys, (float(x) for x in out)))

    def add_single_fastq(self, cfg, f) -> int:
        return self._check(_lib.gaml_hip_add_single_fastq(self._h, C.byref(cfg), f.encode()))

    def add_paired_fastq(self, cfg, f1, f2) -> int:
        return self._check(_lib.gaml_hip_add_paired_fastq(self._h, C.byref(cfg), f1.encode(), f2.encode()))

    def add_pacbio_fastq(self, cfg, f) -> int:
        return self._check(_lib.gaml_hip_add_pacbio_fastq(self._h, C.byref(cfg), f.encode()))

    def put_window_records(self, rs, mate, walk, recs: np.ndarray):
        walk = np.ascontiguousarray(walk, np.int32)
        recs = np.ascontiguousarray(recs, ALIGMENT)
        self._check(_lib.gaml_hip_put_window_records(self._h, rs, mate, walk, len(walk), recs.ctypes.data, len(recs)))

    def put_pacbio_records(self, rs, walk, rec3, logp):
        walk = np.ascontiguousarray(walk, np.int32)
        rec3 = np.asarray(rec3, np.int32).reshape(-1, 3)
        recs = np.zeros(len(rec3), PACBIO_ALIGMENT)
        recs["position"], recs["position_end"], recs["read_id"] = rec3[:, 0], rec3[:, 1], rec3[:, 2]
        recs["logprob"] = logp
        self._check(_lib.gaml_hip_put_pacbio_records(self._h, rs, walk, len(walk), recs.ctypes.data, len(recs)))

    # ---- hot path
    def calc_prob(self, paths):
        flat, offs = _flat(paths)
        prob = C.c_double()
        tl = C.c_int32()
        zeros = np.zeros(2 * max(1, self.num_readsets()), np.int32)
        self._check(_lib.gaml_hip_calc_prob(self._h, flat, offs, len(paths), C.byref(prob), zeros, C.byref(tl)))
        return prob.value, zeros.reshape(-1, 2)[: self.num_readsets()].copy(), tl.value

    def score(self, fp: "FlatPaths") -> float:
        """gaml_hip_calc_prob with the least Python around it (what a C++ caller's `CalcProb(paths)` costs):
        prebuilt FlatPaths, raw pointers, preallocated outputs. zeros / total_len of the call stay in
        self.last_zeros / self.last_total_len."""
        if self._fast is None:
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p)
            self._fast = proto(("gaml_hip_calc_prob", _lib))
            self._prob, self._tl = C.c_double(), C.c_int32()
            self._zeros = np.zeros(2 * max(1, self.num_readsets()), np.int32)
            self._out_ptrs = (C.addressof(self._prob), self._zeros.ctypes.data, C.addressof(self._tl))
        rc = self._fast(self._h, fp.flat_ptr, fp.offs_ptr, fp.n, *self._out_ptrs)
        if rc < 0:
            self._check(rc)
        return self._prob.value

    # lean forms of the two-phase calls for tight loops (gaml_amd.dist): raw pointers, preallocated outputs
    def eval_begin_fast(self, fp: "FlatPaths"):
        if self._fast_begin is None:
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)
            self._fast_begin = proto(("gaml_hip_eval_begin", _lib))
            self._pending, self._tl2 = C.c_int64(), C.c_int32()
            self._begin_out = (C.addressof(self._pending), C.addressof(self._tl2))
        rc = self._fast_begin(self._h, fp.flat_ptr, fp.offs_ptr, fp.n, *self._begin_out)
        if rc < 0:
            self._check(rc)
        return self._pending.value, self._tl2.value

    def shm_exchange_open(self, name: str, rank: int, world: int, cap_doubles: int):
        """gaml_hip_shm_exchange_open: single-node sum of host-resident partials through POSIX shared memory."""
        self._check(_lib.gaml_hip_shm_exchange_open(self._h, name.encode(), rank, world, cap_doubles))

    def shm_allreduce_sum(self, ptr: int, n_doubles: int):
        rc = _lib.gaml_hip_shm_allreduce_sum(self._h, ptr, n_doubles)
        if rc < 0:
            self._check(rc)

    def shm_exchange_close(self, unlink_name: bool = False):
        self._check(_lib.gaml_hip_shm_exchange_close(self._h, 1 if unlink_name else 0))

    def eval_finish_fast(self, out_ptr: int):
        """gaml_hip_eval_finish (blocking) into a host buffer given by address."""
        if getattr(self, "_fast_finish", None) is None:
            self._fast_finish = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)(("gaml_hip_eval_finish", _lib))
        rc = self._fast_finish(self._h, out_ptr)
        if rc < 0:
            self._check(rc)

    def fetch_async(self, d_ptr: int, n_doubles: int, stream: int = 0):
        """gaml_hip_fetch_async: device doubles -> the context's pinned block, behind everything enqueued on `stream`."""
        rc = _lib.gaml_hip_fetch_async(self._h, d_ptr, n_doubles, stream)
        if rc < 0:
            self._check(rc)

    def fetch_wait(self, out_ptr: int, n_doubles: int):
        rc = _lib.gaml_hip_fetch_wait(self._h, out_ptr, n_doubles)
        if rc < 0:
            self._check(rc)

    def combine_fast(self, partials_ptr: int, total_len: int) -> float:
        """gaml_hip_combine_partials on a host buffer given by address; zeros stay in self.last_zeros."""
        if self._fast_combine is None:
            proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p)
            self._fast_combine = proto(("gaml_hip_combine_partials", _lib))
            if self._fast is None:
                self._prob, self._tl = C.c_double(), C.c_int32()
                self._zeros = np.zeros(2 * max(1, self.num_readsets()), np.int32)
            self._combine_out = (C.addressof(self._prob), self._zeros.ctypes.data)
        rc = self._fast_combine(self._h, partials_ptr, total_len, *self._combine_out)
        if rc < 0:
            self._check(rc)
        self._tl.value = total_len
        return self._prob.value

    @property
    def last_zeros(self):
        return self._zeros.reshape(-1, 2)[: self.num_readsets()].copy()

    @property
    def last_total_len(self):
        return self._tl.value

    def calc_prob_batch(self, path_sets):
        """[(prob, zeros, total_len)] of several path sets in one call (gaml_hip_calc_prob_batch)."""
        if isinstance(path_sets, BatchPaths):
            b = path_sets
        else:
            b = BatchPaths(path_sets)
        ns = max(1, self.num_readsets())
        probs = np.zeros(max(1, b.n_sets))
        zeros = np.zeros(max(1, b.n_sets) * 2 * ns, np.int32)
        tls = np.zeros(max(1, b.n_sets), np.int32)
        self._check(_lib.gaml_hip_calc_prob_batch(self._h, b.n_sets, b.flat, b.offs, b.set_offs, probs, zeros.ctypes.data, tls.ctypes.data))
        z = zeros.reshape(-1, ns, 2)[:, : self.num_readsets()]
        return [(float(probs[i]), z[i].copy(), int(tls[i])) for i in range(b.n_sets)]

    def calc_partials(self, paths):
        flat, offs = _flat(paths)
        tl = C.c_int32()
        part = np.zeros(4 * max(1, self.num_readsets()), np.float64)
        self._check(_lib.gaml_hip_calc_partials(self._h, flat, offs, len(paths), part, C.byref(tl)))
        return part.reshape(-1, 4)[: self.num_readsets()].copy(), tl.value

    def calc_partials_async(self, paths, d_partials_ptr: int, stream_ptr: int = 0):
        flat, offs = _flat(paths)
        tl = C.c_int32()
        self._check(_lib.gaml_hip_calc_partials_async(self._h, flat, offs, len(paths), C.c_void_p(d_partials_ptr),
                                                      C.c_void_p(stream_ptr), C.byref(tl)))
        return tl.value

    # two-phase form (sharded contexts: exchange window maxima between begin and finish)
    def eval_begin(self, paths):
        flat, offs = _flat(paths)
        pending, tl = C.c_int64(), C.c_int32()
        self._check(_lib.gaml_hip_eval_begin(self._h, flat, offs, len(paths), C.byref(pending), C.byref(tl)))
        return pending.value, tl.value

    def eval_pending_maxpos(self) -> np.ndarray:
        n = _lib.gaml_hip_eval_pending_maxpos(self._h, None, 0)
        out = np.zeros(max(1, n), np.int32)
        _lib.gaml_hip_eval_pending_maxpos(self._h, out.ctypes.data, n)
        return out[:n]

    def eval_apply_maxpos(self, reduced):
        reduced = np.ascontiguousarray(reduced, np.int32)
        self._check(_lib.gaml_hip_eval_apply_maxpos(self._h, reduced if reduced.size else np.zeros(1, np.int32), reduced.size))

    def eval_finish(self):
        part = np.zeros(4 * max(1, self.num_readsets()), np.float64)
        self._check(_lib.gaml_hip_eval_finish(self._h, part))
        return part.reshape(-1, 4)[: self.num_readsets()].copy()

    def eval_finish_async(self, d_partials_ptr: int, stream_ptr: int = 0):
        self._check(_lib.gaml_hip_eval_finish_async(self._h, C.c_void_p(d_partials_ptr), C.c_void_p(stream_ptr)))

    def sync(self):
        self._check(_lib.gaml_hip_sync(self._h))

    def compact_tables(self):
        self._check(_lib.gaml_hip_compact_tables(self._h))

    # sharded evaluation with a coverage penalty: scoring, then the coverage maps of all ranks, then the sweeps
    def eval_score_async(self, d_partials_ptr: int, stream_ptr: int = 0) -> int:
        return self._check(_lib.gaml_hip_eval_score_async(self._h, C.c_void_p(d_partials_ptr), C.c_void_p(stream_ptr)))

    def eval_coverage_bytes(self, i: int) -> int:
        n = C.c_int64(0)
        self._check(_lib.gaml_hip_eval_coverage_export_async(self._h, i, None, 0, C.byref(n), None))
        return n.value

    def eval_coverage_export_async(self, i: int, dst_ptr: int, cap: int, stream_ptr: int = 0) -> int:
        n = C.c_int64(0)
        self._check(_lib.gaml_hip_eval_coverage_export_async(self._h, i, C.c_void_p(dst_ptr), cap, C.byref(n), C.c_void_p(stream_ptr)))
        return n.value

    def eval_pacbio_pending(self) -> int:
        return self._check(_lib.gaml_hip_eval_pacbio_pending(self._h))

    def eval_pacbio_intervals(self, i: int) -> int:
        """How many alignment intervals (16 bytes each, device memory) this rank's reads contribute to the sweep."""
        return self._check(int(_lib.gaml_hip_eval_pacbio_intervals(self._h, i)))

    def eval_pacbio_export_async(self, i: int, dst_ptr: int, cap: int, stream_ptr: int = 0):
        self._check(_lib.gaml_hip_eval_pacbio_export_async(self._h, i, C.c_void_p(dst_ptr), cap, C.c_void_p(stream_ptr)))

    def eval_pacbio_finish_async(self, i: int, intervals_ptr: int, n_intervals: int, contribute: bool, stream_ptr: int = 0):
        """intervals_ptr: all ranks' gathered intervals in DEVICE memory (int32 {contig, begin, end, 0} each)."""
        self._check(_lib.gaml_hip_eval_pacbio_finish_async(self._h, i, C.c_void_p(intervals_ptr), n_intervals, 1 if contribute else 0, C.c_void_p(stream_ptr)))

    def eval_coverage_finish_async(self, i: int, maps_ptr: int, n_maps: int, contribute: bool, stream_ptr: int = 0):
        self._check(_lib.gaml_hip_eval_coverage_finish_async(self._h, i, C.c_void_p(maps_ptr), n_maps, 1 if contribute else 0,
                                                             C.c_void_p(stream_ptr)))

    def combine_partials(self, partials, total_len):
        part = np.ascontiguousarray(partials, np.float64).reshape(-1)
        prob = C.c_double()
        zeros = np.zeros(2 * max(1, self.num_readsets()), np.int32)
        self._check(_lib.gaml_hip_combine_partials(self._h, part, total_len, C.byref(prob), zeros))
        return prob.value, zeros.reshape(-1, 2)[: self.num_readsets()].copy()

    # ---- introspection
    def num_readsets(self) -> int:
        return _lib.gaml_hip_num_readsets(self._h)

    def readset_reads(self, rs) -> int:
        return _lib.gaml_hip_readset_reads(self._h, rs)

    def readset_kind(self, rs) -> int:
        return _lib.gaml_hip_readset_kind(self._h, rs)

    def num_nodes(self) -> int:
        return _lib.gaml_hip_num_nodes(self._h)

    def node_len(self, i) -> int:
        return _lib.gaml_hip_node_len(self._h, i)

    def read_probs(self, rs) -> np.ndarray:
        n = self.readset_reads(rs)
        out = np.zeros(max(1, n), np.float64)
        got = self._check(_lib.gaml_hip_read_probs(self._h, rs, out, len(out)))
        return out[:got]

    def bad_bases(self, rs) -> int:
        v = C.c_int64()
        self._check(_lib.gaml_hip_bad_bases(self._h, rs, C.byref(v)))
        return v.value

    def window_count(self, rs, mate=0) -> int:
        return _lib.gaml_hip_window_count(self._h, rs, mate)

    def window_records(self, rs, mate, walk):
        walk = np.ascontiguousarray(walk, np.int32)
        n = _lib.gaml_hip_window_records(self._h, rs, mate, walk, len(walk), None, 0)
        if n < 0:
            return None
        out = np.zeros(max(1, n), ALIGMENT)
        _lib.gaml_hip_window_records(self._h, rs, mate, walk, len(walk), out.ctypes.data, n)
        o = out[:n]
        return np.stack([o["position"], o["edit_dist"], o["read_id"], o["orientation"]], axis=1)

    def align_window(self, rs, mate, walk) -> int:
        walk = np.ascontiguousarray(walk, np.int32)
        return _lib.gaml_hip_align_window(self._h, rs, mate, walk, len(walk))

    def debug_prepare(self, paths):
        flat, offs = _flat(paths)
        self._check(_lib.gaml_hip_debug_prepare(self._h, flat, offs, len(paths)))

    def debug_occurrences(self, rs, mate=0) -> np.ndarray:
        n = _lib.gaml_hip_debug_occurrences(self._h, rs, mate, np.zeros(5, np.int32), 0)
        out = np.zeros(5 * max(1, n), np.int32)
        _lib.gaml_hip_debug_occurrences(self._h, rs, mate, out, n)
        return out.reshape(-1, 5)[:n]

    def debug_table_occurrences(self, rs, mate=0):
        """(entries [n, 5] = window, shift, min_pos, path position, path-local rank; info = {incremental, incremental_calls, full_calls})"""
        info = np.zeros(3, np.int64)
        n = _lib.gaml_hip_debug_table_occurrences(self._h, rs, mate, np.zeros(5, np.int32), 0, info)
        out = np.zeros(5 * max(1, n), np.int32)
        _lib.gaml_hip_debug_table_occurrences(self._h, rs, mate, out, n, info)
        return out.reshape(-1, 5)[:n], {"incremental": bool(info[0]), "incremental_calls": int(info[1]), "full_calls": int(info[2])}

    def debug_window_walk(self, rs, mate, wid) -> list:
        buf = np.zeros(64, np.int32)
        n = _lib.gaml_hip_debug_window_walk(self._h, rs, mate, wid, buf, 64)
        if n > 64:
            buf = np.zeros(n, np.int32)
            _lib.gaml_hip_debug_window_walk(self._h, rs, mate, wid, buf, n)
        return [int(x) for x in buf[:n]]

    def table_stats(self, rs):
        out = np.zeros(10, np.int64)
        self._check(_lib.gaml_hip_table_stats(self._h, rs, out))
        # ("worker_rebuilds": the name of rounds 1-3, when a host thread built the tables; now = rebuilds beside the evaluations, on a stream of their own)
        return {"full_rebuilds": int(out[0]), "delta_updates": int(out[1]), "dirty_pairs": int(out[2]), "worker_rebuilds": int(out[3]), "side_stream_rebuilds": int(out[3]),
                "batches_patched": int(out[4]), "batches_full": int(out[5]), "records_left_out": [int(out[6]), int(out[7])], "delta_records_left_out": int(out[8]),
                "static_index_pairs": int(out[9])}

    def aligner_stats(self):
        w, k, us = C.c_int64(), C.c_int64(), C.c_double()
        _lib.gaml_hip_aligner_stats(self._h, C.byref(w), C.byref(k), C.byref(us))
        return {"windows": w.value, "candidates": k.value, "us": us.value}

    def aligner_stages(self):
        """Host-clock time of the GPU aligner by stage (us, cumulative) and the number of batches."""
        out = np.zeros(6, np.float64)
        self._check(_lib.gaml_hip_aligner_stages(self._h, out))
        return {"strings_upload": out[0], "spans_candidates": out[1], "extension": out[2], "hits_d2h": out[3], "sort_file": out[4], "batches": int(out[5])}

    def last_phases(self):
        out = np.zeros(8, np.float64)
        _lib.gaml_hip_last_phases(self._h, out)
        return out

    def debug_timeline(self, rs: int, cap_waves: int = 1 << 16) -> np.ndarray:
        """[waves, 8] wall-clock stamps (10 ns units) of the last evaluation run with knob 3 = 8."""
        out = np.zeros((cap_waves, 8), np.uint64)
        n = self._check(_lib.gaml_hip_debug_timeline(self._h, rs, out.ctypes.data, cap_waves))
        return out[:n]

    def debug_set_knob(self, knob, value):
        self._check(_lib.gaml_hip_debug_set_knob(self._h, knob, value))

    def debug_fold_check(self, rs):
        """Host-only: record tables with / without the always-overwritten records, compared pair by pair."""
        out = np.zeros(6, np.int64)
        self._check(_lib.gaml_hip_debug_fold_check(self._h, rs, out))
        return {"records_left_out": [int(out[0]), int(out[1])], "compact_pairs": [int(out[2]), int(out[3])], "records_checked": int(out[4]),
                "violations": int(out[5])}

    def debug_radix_sort(self, keys, vals=None, begin_bit=0, end_bit=64, running_max=False):
        """The library's own stable radix sort on bits [begin_bit, end_bit) (gaml_amd/csrc/radix_sort.hip.h): returns the
        sorted keys, the payload in the keys' order (or None) and, if asked for, the inclusive running maximum of the
        sorted payload (of the sorted keys without one)."""
        k = np.ascontiguousarray(keys, np.uint64).copy()
        v = None if vals is None else np.ascontiguousarray(vals, np.uint64).copy()
        m = np.zeros(len(k), np.uint64) if running_max else None
        self._check(_lib.gaml_hip_debug_radix_sort(self._h, k.ctypes.data, None if v is None else v.ctypes.data, len(k), begin_bit, end_bit,
                                                   None if m is None else m.ctypes.data))
        return k, v, m

    def debug_static_check(self, rs):
        """Host-only: the compact class's static memo indices recomputed from the window cache."""
        out = np.zeros(8, np.int64)
        self._check(_lib.gaml_hip_debug_static_check(self._h, rs, out))
        return {"static_pairs": int(out[0]), "other_pairs": int(out[1]), "violations": int(out[2]), "no_record": int(out[3]),
                "different_windows": int(out[4]), "orientation": int(out[5]), "distance": int(out[6]), "edits_or_code": int(out[7])}

    def debug_tables_check(self, rs):
        """The device table build against the host restatement, entry by entry (development build)."""
        out = np.zeros(8, np.int64)
        rc = _lib.gaml_hip_debug_tables_check(self._h, rs, out)
        res = {"pairs": int(out[0]), "compact": int(out[1]), "static": int(out[2]), "two": int(out[3]), "four": int(out[4]), "more": int(out[5]),
               "compared": int(out[6]), "mismatches": int(out[7])}
        if rc != 0 and res["compared"] == 0:
            self._check(rc)
        return res

    def debug_block_partials(self, rs, set_index=0):
        sums, zeros, lay = np.zeros(8192, np.float64), np.zeros(8192, np.int32), np.zeros(8, np.int32)
        _lib.gaml_hip_debug_block_partials.argtypes = [C.c_void_p, C.c_int, C.c_int32, _f64p, _i32p, C.c_int32, _i32p]
        n = _lib.gaml_hip_debug_block_partials(self._h, rs, set_index, sums, zeros, 8192, lay)
        return sums[:max(n, 0)].copy(), zeros[:max(n, 0)].copy(), lay.tolist()

    def pair_classes(self, rs):
        out = np.zeros(4, np.int64)
        self._check(_lib.gaml_hip_pair_classes(self._h, rs, out))
        return out

    # (the names these carried while they lived in gaml_hip_debug.h)
    debug_table_stats, debug_profile, debug_class_counts = table_stats, last_phases, pair_classes

    def last_timing(self):
        out = np.zeros(3, np.float64)
        _lib.gaml_hip_last_timing(self._h, out)
        return {"host_us": out[0], "device_wall_us": out[1], "kernel_us": out[2]}

    def set_event_timing(self, on):
        """False / 0: off; True / 1: every scoring launch; k > 1: every k-th launch."""
        _lib.gaml_hip_set_event_timing(self._h, int(on))

    def kernel_stats(self, reset=False):
        n, us, b = C.c_int64(), C.c_double(), C.c_double()
        _lib.gaml_hip_kernel_stats(self._h, 1 if reset else 0, C.byref(n), C.byref(us), C.byref(b))
        return {"launches": n.value, "device_us": us.value, "algo_bytes": b.value}


def debug_sam_band(sam_line: str, total_len: int):
    """Host-only: (fields dict, row0, lo[], hi[]) of one SAM line as the library parses / bands it."""
    raw = sam_line.encode()
    f = np.zeros(10, np.int32); r0 = np.zeros(1, np.int32)
    n = _lib.gaml_hip_debug_sam_band(raw, len(raw), total_len, f, r0, np.zeros(1, np.int32), np.zeros(1, np.int32), 0)
    if n < 0:
        raise GamlHipError(n, "malformed SAM line")
    lo = np.zeros(n, np.int32); hi = np.zeros(n, np.int32)
    _lib.gaml_hip_debug_sam_band(raw, len(raw), total_len, f, r0, lo, hi, n)
    keys = ["flags", "len", "posstart", "posend", "sstart", "send", "slen", "tstart", "tend", "edit_dist"]
    return dict(zip(keys, (int(x) for x in f))), int(r0[0]), lo, hi


def debug_sam_shape(sam_line: str, total_len: int):
    """Host-only: the per-alignment inputs of the DP kernel (shape dict, run-length CIGAR as (len, code) pairs)."""
    raw = sam_line.encode()
    f = np.zeros(6, np.int32)
    n = _lib.gaml_hip_debug_sam_shape(raw, len(raw), total_len, f, None, 0)
    if n < 0:
        raise GamlHipError(n, "malformed SAM line")
    ops = np.zeros(max(1, n), np.uint32)
    _lib.gaml_hip_debug_sam_shape(raw, len(raw), total_len, f, ops.ctypes.data, n)
    keys = ["n_ops", "row_f", "col_f", "bl", "el", "max_width"]
    return dict(zip(keys, (int(x) for x in f))), [(int(o) >> 2, "MID"[int(o) & 3]) for o in ops[:n]]
