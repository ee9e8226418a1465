"""ctypes access to oracle/liboracle.so -- our plain-C restatement (test oracle only)."""
import ctypes as C
import os
import subprocess
import numpy as np

from bwa_amd.structs import MemOpt, ALNREG_DTYPE, INTV_DTYPE, SEED_DTYPE, CHAIN_HDR_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORC_SO):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True, stdout=subprocess.DEVNULL)
        L = C.CDLL(ORC_SO)
        L.orc_index_load.restype = C.c_void_p
        L.orc_index_load.argtypes = [C.c_char_p]
        L.orc_index_free.argtypes = [C.c_void_p]
        L.orc_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_occ4.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_sa.restype = C.c_uint64
        L.orc_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_align.restype = C.c_int64
        L.orc_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_intervals.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_chains.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_regs_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        sz = (C.c_int32 * 8)()
        L.orc_sizes(sz)
        assert sz[0] == C.sizeof(MemOpt) and sz[1] == ALNREG_DTYPE.itemsize and sz[2] == INTV_DTYPE.itemsize
        assert sz[3] == SEED_DTYPE.itemsize and sz[4] == CHAIN_HDR_DTYPE.itemsize
        _lib = L
    return _lib


class OrcIndex:
    def __init__(self, prefix: str):
        self.h = lib().orc_index_load(prefix.encode())
        assert self.h, "oracle failed to load index " + prefix

    def close(self):
        if self.h:
            lib().orc_index_free(self.h)
            self.h = None

    def set_alt(self, rid, flag=1):
        lib().orc_set_alt(self.h, rid, flag)

    def occ4(self, k):
        out = np.zeros(4, dtype=np.uint64)
        lib().orc_occ4(self.h, C.c_uint64(k & 0xFFFFFFFFFFFFFFFF), out.ctypes.data)
        return out

    def sa(self, k):
        return lib().orc_sa(self.h, k)

    def align(self, opt, seqs, off):
        n = off.shape[0] - 1
        counts = np.zeros(n, dtype=np.int32)
        cap = max(4 * n, 1024)
        while True:
            out = np.zeros(cap, dtype=ALNREG_DTYPE)
            tot = lib().orc_align(self.h, C.byref(opt), n, seqs.ctypes.data, off.ctypes.data, counts.ctypes.data, out.ctypes.data, cap)
            if tot <= cap:
                return counts, out[:tot]
            cap = int(tot)

    def intervals(self, opt, seq):
        cap = 4096
        out = np.zeros(cap, dtype=INTV_DTYPE)
        n = lib().orc_intervals(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]

    def chains(self, opt, seq, stage):
        capc, caps = 1 << 14, 1 << 17
        hdr = np.zeros(capc, dtype=CHAIN_HDR_DTYPE)
        seeds = np.zeros(caps, dtype=SEED_DTYPE)
        ns = C.c_int32(0)
        n = lib().orc_chains(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, stage, hdr.ctypes.data, capc, seeds.ctypes.data, caps, C.byref(ns))
        assert n <= capc and ns.value <= caps
        return hdr[:n], seeds[:ns.value]

    def regs_stage(self, opt, seq, stage):
        cap = 1 << 14
        out = np.zeros(cap, dtype=ALNREG_DTYPE)
        n = lib().orc_regs_stage(self.h, C.byref(opt), seq.shape[0], seq.ctypes.data, stage, out.ctypes.data, cap)
        assert n <= cap
        return out[:n]
