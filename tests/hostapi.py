"""ctypes access to bwa_amd/csrc/host/libbwamem_host.so -- the product's host finalize code (no GPU needed)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_DIR = os.path.join(ROOT, "bwa_amd", "csrc", "host")
HOST_SO = os.path.join(HOST_DIR, "libbwamem_host.so")
_lib = None


def build():
    if os.environ.get("BWA_AMD_HOST_LIB"):       # (a sanitizer build of the same sources: tools/sanitize_mock.sh)
        return os.environ["BWA_AMD_HOST_LIB"]
    srcs = [os.path.join(HOST_DIR, f) for f in sorted(os.listdir(HOST_DIR)) if f.endswith((".cpp", ".h")) and not f.startswith("main_")]
    if os.path.exists(HOST_SO) and all(os.path.getmtime(HOST_SO) >= os.path.getmtime(s) for s in srcs):
        return HOST_SO
    cpp = [s for s in srcs if s.endswith(".cpp")]
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off"] + cpp + ["-o", HOST_SO, "-lpthread"], check=True)
    return HOST_SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.bwamem_host_create.restype = C.c_void_p
        L.bwamem_host_create.argtypes = [C.c_char_p]
        L.bwamem_host_destroy.argtypes = [C.c_void_p]
        L.bwamem_host_set_alt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.bwamem_host_regs2sam.restype = C.c_void_p
        L.bwamem_host_regs2sam.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_char_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.bwamem_host_pestat.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.bwamem_host_matesw_records.restype = C.c_int64
        L.bwamem_host_matesw_records.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.bwamem_host_free.argtypes = [C.c_void_p]
        L.bwamem_host_region_cigars.restype = C.c_int64
        L.bwamem_host_region_cigars.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        _lib = L
    return _lib


class HostFinalize:
    def __init__(self, prefix):
        self.h = lib().bwamem_host_create(prefix.encode())
        assert self.h

    def set_alt(self, rid, flag=1):
        lib().bwamem_host_set_alt(self.h, rid, flag)

    def regs2sam(self, opt, names, seqs_nt4: np.ndarray, quals: bytes, off, counts, regs, n_processed=0, pes0=None, n_threads=4, cigs=None, msw=None, cig_ops=None) -> bytes:
        n = off.shape[0] - 1
        nm = b"".join(x.encode() + b"\0" for x in names)
        ln = C.c_int64(0)
        regs = np.ascontiguousarray(regs)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        seqs_nt4 = np.ascontiguousarray(seqs_nt4, dtype=np.uint8)
        p = lib().bwamem_host_regs2sam(self.h, C.byref(opt), n_processed, n, nm, seqs_nt4.ctypes.data, quals, off.ctypes.data, counts.ctypes.data, regs.ctypes.data, pes0, n_threads, C.byref(ln),
                                       None if cigs is None else np.ascontiguousarray(cigs).ctypes.data,
                                       None if msw is None else np.ascontiguousarray(msw).ctypes.data, 0 if msw is None else int(msw.shape[0]),
                                       None if cig_ops is None else np.ascontiguousarray(cig_ops, dtype=np.uint32).ctypes.data)
        s = C.string_at(p, ln.value)
        lib().bwamem_host_free(p)
        return s

    def region_cigars(self, opt, seqs_nt4, off, counts, regs, with_ops=False):
        """Host-computed bwagpu_cigar_t records (the reference for the device's bwagpu_batch_cigars); with_ops: also the operation
        array that records with 7..64 operations point into (else such regions are reported unserved)."""
        from bwa_amd.api import CIGAR_DTYPE
        regs = np.ascontiguousarray(regs); counts = np.ascontiguousarray(counts, dtype=np.int32); seqs_nt4 = np.ascontiguousarray(seqs_nt4, dtype=np.uint8)
        out = np.zeros(regs.shape[0], dtype=CIGAR_DTYPE)
        cap = 330 * regs.shape[0] + 64
        ops = np.zeros(cap, dtype=np.uint32) if with_ops else None
        n = lib().bwamem_host_region_cigars(self.h, C.byref(opt), off.shape[0] - 1, seqs_nt4.ctypes.data, off.ctypes.data, counts.ctypes.data, regs.ctypes.data, out.ctypes.data,
                                            ops.ctypes.data if with_ops else None, cap)
        return (out, ops[:n]) if with_ops else out

    PESTAT_DTYPE = np.dtype([("low", "<i4"), ("high", "<i4"), ("failed", "<i4"), ("pad_", "<i4"), ("avg", "<f8"), ("std", "<f8")])   # hostmem::Pestat

    def pestat(self, opt, counts, regs):
        """mem_pestat of the batch: Pestat[4] (low, high, failed, avg, std)."""
        regs = np.ascontiguousarray(regs); counts = np.ascontiguousarray(counts, dtype=np.int32)
        pes = np.zeros(4, dtype=self.PESTAT_DTYPE)
        lib().bwamem_host_pestat(self.h, C.byref(opt), counts.shape[0], counts.ctypes.data, regs.ctypes.data, pes.ctypes.data)
        return pes

    def matesw_records(self, opt, seqs_nt4, off, counts, regs, pes):
        """Host-computed bwagpu_matesw_t records (the reference for the device's bwagpu_batch_matesw)."""
        from bwa_amd.api import MATESW_DTYPE
        regs = np.ascontiguousarray(regs); counts = np.ascontiguousarray(counts, dtype=np.int32); seqs_nt4 = np.ascontiguousarray(seqs_nt4, dtype=np.uint8)
        cap = 2 * counts.shape[0] + 1024
        out = np.zeros(cap, dtype=MATESW_DTYPE)
        n = lib().bwamem_host_matesw_records(self.h, C.byref(opt), counts.shape[0], seqs_nt4.ctypes.data, off.ctypes.data, counts.ctypes.data, regs.ctypes.data,
                                             np.ascontiguousarray(pes).ctypes.data, out.ctypes.data, cap)
        return out[:n]

    def close(self):
        if self.h:
            lib().bwamem_host_destroy(self.h)
            self.h = None


def decode_cigars(cigs, ops):
    """bwagpu_cigar_t records as [(score, n_cigar, (op, ...), nm, md)]: records with more than 6 operations or an MD string of more than 8
    characters are looked up in the operation array, whose order differs between producers (the device appends in the order its waves finish)."""
    out = []
    raw = ops.tobytes() if ops is not None else b""
    for c in cigs:
        n = int(c["n_cigar"])
        if n > 6:
            at = int(c["cigar"][1]) << 32 | int(c["cigar"][0])
            assert at + n <= ops.shape[0], (at, n, ops.shape[0])
            body = tuple(int(x) for x in ops[at:at + n])
        else:
            body = tuple(int(x) for x in c["cigar"][:max(n, 0)])
        ml = int(c["md_len"])
        if n < 0:
            md = b""
        elif ml <= 8:
            md = int(c["md"]).to_bytes(8, "little")[:ml]
        else:
            at = int(c["md"])
            assert at * 4 + ml <= len(raw), (at, ml, len(raw))
            md = raw[at * 4: at * 4 + ml]
        out.append((int(c["score"]), n, body, int(c["nm"]), md))
    return out
