"""Python binding (ctypes) of the C-ABI in include/bwagpu.h.

This is plumbing for tests and bench.py: every call goes straight into libbwagpu.so, the HIP library built by
`bwa_amd.build`.  There is no CPU implementation behind it; if the library is missing or no GPU is visible the
constructor raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

try:   # torch bundles its own HIP runtime (libamdhip64); loading it first keeps one runtime per process
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

from .structs import ALNREG_DTYPE, MemOpt

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "csrc", "libbwagpu.so")

INTV3_DTYPE = np.dtype([("x0", "<u8"), ("x2", "<u8"), ("info", "<u8")])
GCHAIN_DTYPE = np.dtype([("n_seeds", "<i4"), ("rid", "<i4"), ("w", "<i4"), ("kept", "<i4"), ("is_alt", "<i4"),
                         ("frac_rep", "<f4"), ("pos", "<i8")])
GSEED_DTYPE = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("_pad", "<i4")])
assert GCHAIN_DTYPE.itemsize == 32 and GSEED_DTYPE.itemsize == 24


MATESW_DTYPE = np.dtype([("read", "<i4"), ("r", "<i4"), ("anchor_rb", "<i8"), ("anchor_rid", "<i4"), ("score", "<i4"), ("te", "<i4"), ("qe", "<i4"),
                         ("score2", "<i4"), ("te2", "<i4"), ("tb", "<i4"), ("qb", "<i4"), ("pad_", "<i4"), ("pad2_", "<i4")])   # bwagpu_matesw_t (56 bytes with its tail padding)
assert MATESW_DTYPE.itemsize == 56
PES_DTYPE = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad_", "<i4")])               # bwagpu_pes_t
CIGAR_DTYPE = np.dtype([("score", "<i4"), ("n_cigar", "<i4"), ("cigar", "<u4", (6,)), ("nm", "<i4"), ("md_len", "<i4"), ("md", "<u8")])   # bwagpu_cigar_t
assert CIGAR_DTYPE.itemsize == 48
DP_CASE_DTYPE = np.dtype([("q_off", "<i4"), ("q_len", "<i4"), ("t_off", "<i4"), ("t_len", "<i4"), ("w", "<i4"), ("h0", "<i4"), ("end_bonus", "<i4"), ("flags", "<i4")])   # bwagpu_dp_case_t
assert DP_CASE_DTYPE.itemsize == 32


class Stats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in (
        "n_reads", "n_bases", "n_intv", "n_seeds", "n_chains", "n_regs_raw", "n_regs", "n_occ_blocks", "n_lf_steps",
        "n_ext_calls", "n_ext_cells", "n_glb_calls", "n_glb_cells", "ref_bases", "n_sw_calls", "n_sw_cells")] + [
        (n, C.c_float) for n in ("ms_seed", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup", "ms_total")] + [
        ("n_retries", C.c_int32), ("ms_publish", C.c_float), ("n_tab_lookups", C.c_int64), ("n_bt_nodes", C.c_int64), ("n_chain_recs", C.c_int64),
        ("n_chain_deferred", C.c_int64), ("n_ext_fast", C.c_int64), ("n_chain_deferred2", C.c_int64),
        ("ms_pack", C.c_float), ("ms_download_copy", C.c_float), ("ms_cigar_kernels", C.c_float), ("ms_cigar_copy", C.c_float),
        ("n_cig_cells", C.c_int64), ("n_cig_dp", C.c_int64), ("retry_mask", C.c_int32), ("reserved_", C.c_int32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved_"}


EXPORTS = [
    "bwagpu_create", "bwagpu_create_from_files", "bwagpu_destroy", "bwagpu_strerror", "bwagpu_last_error", "bwagpu_version",
    "bwagpu_index_info", "bwagpu_densify_sa", "bwagpu_set_stats", "bwagpu_get_stats", "bwagpu_align_bseq", "bwagpu_align_flat",
    "bwagpu_free", "bwagpu_batch_upload", "bwagpu_batch_run", "bwagpu_batch_download", "bwagpu_set_taps", "bwagpu_tap_intervals",
    "bwagpu_tap_chains", "bwagpu_tap_regs_raw", "bwagpu_index_buffers", "bwagpu_index_export", "bwagpu_clone", "bwagpu_index_ready",
    "bwagpu_batch_cigars", "bwagpu_batch_cigar_ops", "bwagpu_debug_phase", "bwagpu_batch_matesw", "bwagpu_clone_to_device", "bwagpu_index_build", "bwagpu_built_free", "bwagpu_abi_sizes", "bwagpu_debug_prof", "bwagpu_debug_hist", "bwagpu_debug_seed_x2", "bwagpu_debug_chain_hist", "bwagpu_debug_dp", "bwagpu_set_cigar_filter", "bwagpu_batch_reserve", "bwagpu_batch_footprint", "bwagpu_mem_info",
    "bwagpu_trim", "bwagpu_set_option", "bwagpu_get_option", "bwagpu_set_default_option", "bwagpu_clear_default_options", "bwagpu_option_name",
]


class IndexDesc(C.Structure):
    _fields_ = [("bwt", C.c_void_p), ("bwt_size", C.c_uint64), ("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64),
                ("sa", C.c_void_p), ("n_sa", C.c_uint64), ("sa_intv", C.c_int), ("pac", C.c_void_p), ("l_pac", C.c_int64),
                ("n_seqs", C.c_int32), ("ctg_offset", C.c_void_p), ("ctg_len", C.c_void_p), ("ctg_is_alt", C.c_void_p)]


class BwaGpuError(RuntimeError):
    pass


def load_library(path: str | None = None) -> C.CDLL:
    path = path or DEFAULT_LIB
    if not os.path.exists(path):
        raise BwaGpuError(f"{path} not found: build the HIP library first (python -m bwa_amd.build); there is no CPU fallback")
    L = C.CDLL(path)
    L.bwagpu_strerror.restype = C.c_char_p
    L.bwagpu_last_error.restype = C.c_char_p
    L.bwagpu_last_error.argtypes = [C.c_void_p]
    L.bwagpu_version.restype = C.c_char_p
    L.bwagpu_create_from_files.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_int]
    L.bwagpu_destroy.argtypes = [C.c_void_p]
    L.bwagpu_index_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_densify_sa.argtypes = [C.c_void_p, C.c_int]
    L.bwagpu_set_stats.argtypes = [C.c_void_p, C.c_int]
    L.bwagpu_set_taps.argtypes = [C.c_void_p, C.c_int]
    L.bwagpu_get_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.bwagpu_align_flat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_batch_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.bwagpu_batch_run.argtypes = [C.c_void_p, C.c_void_p]
    L.bwagpu_batch_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_batch_matesw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_batch_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_batch_cigar_ops.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_tap_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_tap_chains.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_tap_regs_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_free.argtypes = [C.c_void_p]
    L.bwagpu_create.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_int]
    L.bwagpu_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    L.bwagpu_index_ready.argtypes = [C.c_void_p]
    L.bwagpu_index_buffers.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.bwagpu_index_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bwagpu_debug_dp.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.bwagpu_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
    L.bwagpu_get_option.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    L.bwagpu_set_default_option.argtypes = [C.c_char_p, C.c_longlong]
    L.bwagpu_clear_default_options.restype = None
    L.bwagpu_option_name.argtypes = [C.c_int, C.c_void_p]
    return L


def option_names(L: C.CDLL) -> list:
    """The library's option list (bwa_amd/csrc/bwagpu_config.h), as bwagpu_option_name enumerates it."""
    out, i, p = [], 0, C.c_char_p()
    while L.bwagpu_option_name(i, C.byref(p)) == 0:
        out.append(p.value.decode()); i += 1
    return out


import threading
_CREATE_LOCK = threading.Lock()


class BwaGpu:
    """One handle = one GPU with the index resident in HBM."""

    def __init__(self, prefix: str, device: int = 0, lib_path: str | None = None, options: dict | None = None):
        """options: {name: integer} of include/bwagpu.h's option list -- given to the library as defaults for the handle being created (so
        that the ones shaping what is derived from the index at load time -- occ32, occ32_sb_shift, ptab_m -- apply) and forgotten again."""
        self.L = load_library(lib_path)
        self.h = C.c_void_p()
        with _CREATE_LOCK:      # (the defaults are process-wide: two handles created from two threads must not see each other's, or have theirs cleared)
            try:
                for k, v in (options or {}).items():
                    if self.L.bwagpu_set_default_option(k.encode(), int(v)) != 0:
                        raise BwaGpuError(f"unknown option {k!r}")
                rc = self.L.bwagpu_create_from_files(C.byref(self.h), prefix.encode(), device)
            finally:
                if options:
                    self.L.bwagpu_clear_default_options()
        if rc != 0:
            raise BwaGpuError(f"bwagpu_create_from_files({prefix}) failed: {self.L.bwagpu_strerror(rc).decode()}")
        self.set_taps(True)     # the library's default is off (a second region arena per batch); the tests read the stage taps, bench.py turns them off

    @classmethod
    def empty(cls, meta: dict, device: int = 0, lib_path: str | None = None):
        """Receiving side of an index broadcast: allocate the HBM buffers described by `meta` without uploading."""
        self = cls.__new__(cls)
        self.L = load_library(lib_path)
        self.h = C.c_void_p()
        d = IndexDesc()
        for k in ("bwt_size", "primary", "seq_len", "n_sa", "sa_intv", "l_pac", "n_seqs"):
            setattr(d, k, meta[k])
        for i in range(5):
            d.L2[i] = meta["L2"][i]
        off = np.ascontiguousarray(meta["ctg_offset"], dtype=np.int64)
        ln = np.ascontiguousarray(meta["ctg_len"], dtype=np.int32)
        alt = np.ascontiguousarray(meta["ctg_is_alt"], dtype=np.int32)
        d.ctg_offset, d.ctg_len, d.ctg_is_alt = off.ctypes.data, ln.ctypes.data, alt.ctypes.data
        rc = self.L.bwagpu_create(C.byref(self.h), C.byref(d), device)
        if rc != 0:
            raise BwaGpuError(f"bwagpu_create(empty) failed: {self.L.bwagpu_strerror(rc).decode()}")
        return self

    def index_ready(self):
        self._chk(self.L.bwagpu_index_ready(self.h))

    def clone(self):
        """A second handle sharing this one's resident index, with its own stream and arenas (for a second host thread)."""
        other = BwaGpu.__new__(BwaGpu)
        other.L = self.L
        other.h = C.c_void_p()
        self._chk(self.L.bwagpu_clone(self.h, C.byref(other.h)))
        return other

    def index_meta(self) -> dict:
        d = IndexDesc()
        self._chk(self.L.bwagpu_index_export(self.h, C.byref(d), None, None, None))
        off = np.zeros(d.n_seqs, dtype=np.int64); ln = np.zeros(d.n_seqs, dtype=np.int32); alt = np.zeros(d.n_seqs, dtype=np.int32)
        self._chk(self.L.bwagpu_index_export(self.h, C.byref(d), off.ctypes.data, ln.ctypes.data, alt.ctypes.data))
        return {"bwt_size": d.bwt_size, "primary": d.primary, "L2": [int(x) for x in d.L2], "seq_len": d.seq_len, "n_sa": d.n_sa,
                "sa_intv": d.sa_intv, "l_pac": d.l_pac, "n_seqs": d.n_seqs, "ctg_offset": off, "ctg_len": ln, "ctg_is_alt": alt}

    def index_buffers(self):
        """[(device pointer, bytes)] of the BWT/Occ blocks, the SA and the pac -- the payload of the index broadcast."""
        p = [C.c_void_p() for _ in range(3)]
        n = [C.c_uint64() for _ in range(3)]
        self._chk(self.L.bwagpu_index_buffers(self.h, C.byref(p[0]), C.byref(n[0]), C.byref(p[1]), C.byref(n[1]), C.byref(p[2]), C.byref(n[2])))
        return [(p[i].value, n[i].value) for i in range(3)]

    def _chk(self, rc):
        if rc != 0:
            raise BwaGpuError(f"{self.L.bwagpu_strerror(rc).decode()}: {self.L.bwagpu_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.L.bwagpu_destroy(self.h)
            self.h = C.c_void_p()

    def set_stats(self, on=True):
        self._chk(self.L.bwagpu_set_stats(self.h, int(on)))

    def set_option(self, name: str, value: int):
        """bwagpu_set_option: one of the handle's tuning / test options (bwa_amd/csrc/bwagpu_config.h), between batches."""
        self._chk(self.L.bwagpu_set_option(self.h, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        v = C.c_longlong()
        self._chk(self.L.bwagpu_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def set_taps(self, on=True):
        self._chk(self.L.bwagpu_set_taps(self.h, int(on)))

    def densify_sa(self, intv: int):
        self._chk(self.L.bwagpu_densify_sa(self.h, intv))

    def stats(self) -> dict:
        s = Stats()
        self._chk(self.L.bwagpu_get_stats(self.h, C.byref(s)))
        return s.as_dict()

    # -- hot path -------------------------------------------------------------------------------------------
    def upload(self, seqs: np.ndarray, off: np.ndarray):
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        self._n = off.shape[0] - 1
        self._chk(self.L.bwagpu_batch_upload(self.h, self._n, seqs.ctypes.data, off.ctypes.data))

    def run(self, opt: MemOpt):
        self._chk(self.L.bwagpu_batch_run(self.h, C.byref(opt)))

    def _take(self, ptr, n, dtype):
        out = np.frombuffer(C.string_at(ptr, n * dtype.itemsize), dtype=dtype).copy() if n else np.zeros(0, dtype=dtype)
        self.L.bwagpu_free(ptr)
        return out

    def download(self):
        counts = np.zeros(self._n, dtype=np.int32)
        p = C.c_void_p()
        n = C.c_int64()
        self._chk(self.L.bwagpu_batch_download(self.h, counts.ctypes.data, C.byref(p), C.byref(n)))
        return counts, self._take(p, n.value, ALNREG_DTYPE)

    def cigars(self, opt: MemOpt):
        """bwagpu_batch_cigars: one CIGAR_DTYPE record per region of the last download(), in its order."""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_batch_cigars(self.h, C.byref(opt), C.byref(p), C.byref(n)))
        return self._take(p, n.value, CIGAR_DTYPE)

    def cigar_ops(self):
        """bwagpu_batch_cigar_ops: the operation array of the last cigars() call (records with 7..64 operations point into it)."""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_batch_cigar_ops(self.h, C.byref(p), C.byref(n)))
        return self._take(p, n.value, np.dtype("<u4"))

    def matesw(self, opt: MemOpt, pes: np.ndarray):
        """bwagpu_batch_matesw: precomputed mate-rescue alignments (MATESW_DTYPE) for the last download(); pes = PES_DTYPE[4]."""
        pes = np.ascontiguousarray(pes, dtype=PES_DTYPE)
        assert pes.shape == (4,)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_batch_matesw(self.h, C.byref(opt), pes.ctypes.data, C.byref(p), C.byref(n)))
        return self._take(p, n.value, MATESW_DTYPE)

    def debug_dp(self, opt: MemOpt, kind: int, cases: np.ndarray, seqs: np.ndarray) -> np.ndarray:
        """bwagpu_debug_dp: one wavefront of a device DP routine per case (DP_CASE_DTYPE) -> int32[n_cases, 72]."""
        cases = np.ascontiguousarray(cases, dtype=DP_CASE_DTYPE)
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        out = np.zeros((cases.shape[0], 72), dtype=np.int32)
        self._chk(self.L.bwagpu_debug_dp(self.h, C.byref(opt), kind, cases.shape[0], cases.ctypes.data, seqs.ctypes.data, seqs.shape[0], out.ctypes.data))
        return out

    def align(self, opt: MemOpt, seqs: np.ndarray, off: np.ndarray):
        self.upload(seqs, off)
        self.run(opt)
        return self.download()

    # -- taps -------------------------------------------------------------------------------------------------
    def tap_intervals(self):
        counts = np.zeros(self._n, dtype=np.int32)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_tap_intervals(self.h, counts.ctypes.data, C.byref(p), C.byref(n)))
        return counts, self._take(p, n.value, INTV3_DTYPE)

    def tap_chains(self):
        counts = np.zeros(self._n, dtype=np.int32)
        pc, nc, ps, ns = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_tap_chains(self.h, counts.ctypes.data, C.byref(pc), C.byref(nc), C.byref(ps), C.byref(ns)))
        return counts, self._take(pc, nc.value, GCHAIN_DTYPE), self._take(ps, ns.value, GSEED_DTYPE)

    def tap_regs_raw(self):
        counts = np.zeros(self._n, dtype=np.int32)
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.bwagpu_tap_regs_raw(self.h, counts.ctypes.data, C.byref(p), C.byref(n)))
        return counts, self._take(p, n.value, ALNREG_DTYPE)
