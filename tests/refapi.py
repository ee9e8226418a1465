"""ctypes access to oracle/_ref/libbwaref.so -- the compiled, unmodified reference (test oracle only)."""
import ctypes as C
import os
import subprocess
import numpy as np

from bwa_amd.structs import MemOpt, ALNREG_DTYPE, INTV_DTYPE, SEED_DTYPE, CHAIN_HDR_DTYPE, MemPestat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DIR = os.path.join(ROOT, "oracle", "_ref")
REF_SO = os.path.join(REF_DIR, "libbwaref.so")
REF_BWA = os.path.join(REF_DIR, "bwa")
DATA = os.path.join(ROOT, "tests", "_data")


def have_ref() -> bool:
    return os.path.exists(REF_SO) and os.path.exists(REF_BWA)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(REF_SO)
        L.refshim_idx_load.restype = C.c_void_p
        L.refshim_idx_load.argtypes = [C.c_char_p]
        L.refshim_idx_destroy.argtypes = [C.c_void_p]
        L.refshim_idx_info.argtypes = [C.c_void_p, C.c_void_p]
        for f in ("refshim_idx_bwt", "refshim_idx_bns", "refshim_idx_pac"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.refshim_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.refshim_align.restype = C.c_int64
        L.refshim_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.refshim_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.refshim_chains.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.refshim_regs_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.refshim_process_seqs.restype = C.c_void_p
        L.refshim_process_seqs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refshim_regs2sam.restype = C.c_void_p
        L.refshim_regs2sam.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.refshim_free.argtypes = [C.c_void_p]
        sz = (C.c_int32 * 16)()
        L.refshim_sizes(sz)
        assert sz[0] == C.sizeof(MemOpt) and sz[1] == ALNREG_DTYPE.itemsize and sz[2] == INTV_DTYPE.itemsize
        assert sz[3] == SEED_DTYPE.itemsize and sz[11] == CHAIN_HDR_DTYPE.itemsize and sz[6] == C.sizeof(MemPestat)
        _lib = L
    return _lib


def build_index(fasta: str) -> str:
    """Run the reference's `bwa index` (cached: skipped when the .sa is newer than the fasta)."""
    if not (os.path.exists(fasta + ".sa") and os.path.getmtime(fasta + ".sa") >= os.path.getmtime(fasta)):
        subprocess.run([REF_BWA, "index", fasta], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return fasta


class RefIndex:
    def __init__(self, prefix: str):
        self.h = lib().refshim_idx_load(prefix.encode())
        assert self.h, "reference failed to load index " + prefix
        info = np.zeros(8, dtype=np.int64)
        lib().refshim_idx_info(self.h, info.ctypes.data)
        self.l_pac, self.n_seqs, self.seq_len, self.primary, self.sa_intv, self.n_sa, self.bwt_size, self.n_holes = [int(x) for x in info]

    def close(self):
        if self.h:
            lib().refshim_idx_destroy(self.h)
            self.h = None

    def set_alt(self, rid, flag=1):
        lib().refshim_set_alt(self.h, rid, flag)

    def align(self, opt: MemOpt, seqs: np.ndarray, off: np.ndarray):
        n = off.shape[0] - 1
        counts = np.zeros(n, dtype=np.int32)
        cap = max(4 * n, 1024)
        while True:
            out = np.zeros(cap, dtype=ALNREG_DTYPE)
            tot = lib().refshim_align(self.h, C.byref(opt), n, seqs.ctypes.data, off.ctypes.data, counts.ctypes.data, out.ctypes.data, cap)
            if tot <= cap:
                return counts, out[:tot]
            cap = int(tot)

    def intervals(self, opt, seq: np.ndarray):
        cap = 4096
        out = np.zeros(cap, dtype=INTV_DTYPE)
        n = lib().refshim_intervals(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def chains(self, opt, seq: np.ndarray, stage: int):
        capc, caps = 1 << 14, 1 << 17
        hdr = np.zeros(capc, dtype=CHAIN_HDR_DTYPE)
        seeds = np.zeros(caps, dtype=SEED_DTYPE)
        ns = C.c_int32(0)
        n = lib().refshim_chains(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, stage, hdr.ctypes.data, capc, seeds.ctypes.data, caps, C.byref(ns))
        assert n <= capc and ns.value <= caps
        return hdr[:n], seeds[:ns.value]

    def regs_stage(self, opt, seq: np.ndarray, stage: int):
        cap = 1 << 14
        out = np.zeros(cap, dtype=ALNREG_DTYPE)
        n = lib().refshim_regs_stage(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, stage, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def process_seqs(self, opt, names, seqs_ascii: bytes, quals: bytes, off: np.ndarray, n_processed=0, pes0=None) -> bytes:
        n = off.shape[0] - 1
        nm = b"".join(x.encode() + b"\0" for x in names)
        ln = C.c_int64(0)
        p = lib().refshim_process_seqs(self.h, C.byref(opt), n_processed, n, nm, seqs_ascii, quals, off.ctypes.data, pes0, C.byref(ln))
        s = C.string_at(p, ln.value)
        lib().refshim_free(p)
        return s

    def regs2sam(self, opt, names, seqs_nt4: bytes, quals: bytes, off, counts, regs, n_processed=0, pes0=None) -> bytes:
        n = off.shape[0] - 1
        nm = b"".join(x.encode() + b"\0" for x in names)
        ln = C.c_int64(0)
        regs = np.ascontiguousarray(regs)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        p = lib().refshim_regs2sam(self.h, C.byref(opt), n_processed, n, nm, seqs_nt4, quals, off.ctypes.data, counts.ctypes.data, regs.ctypes.data, pes0, C.byref(ln))
        s = C.string_at(p, ln.value)
        lib().refshim_free(p)
        return s
