"""Index construction on the device: writes <prefix>.{bwt,sa,pac,ann,amb} byte-identical to what the
reference's `bwa index` produces (bwtindex.c:255-323), in seconds instead of ~0.5-0.7 s/Mbp.

The work is done by `bwagpu_index_build` (include/bwagpu.h, bwa_amd/csrc/bwagpu_index.hip): a suffix sort of the
forward + reverse-complement text in HBM (bucketed 64-bit radix sort of 32-mers, then prefix doubling restricted to the
still-unsorted suffixes), followed by streaming kernels that lay out the BWT, its Occ checkpoints and the sampled SA the way
bwt_bwtupdate_core / bwt_cal_sa leave them.  This module is the ctypes binding plus the writers of the reference's on-disk
formats:
  .bwt  primary, L2[1..4], then bwt_t::bwt (bwt_dump_bwt, bwt.c:385-393)
  .sa   primary, L2[1..4], sa_intv, seq_len, then SA[k] for k = sa_intv, 2 sa_intv, ... (bwt_dump_sa, bwt.c:396-407)
  .pac  2 bits per base of the forward strand, tail bytes as bns_fasta2bntseq writes them (bntseq.c:314-323)
  .ann / .amb  text (bns_dump, bntseq.c:65-95)
There is no CPU implementation: without libbwagpu.so and a GPU the call raises (the CPU test-suite runs the same HIP source
under the mock runtime of tests/hostsim by passing its library as `lib_path`).
Equality with `bwa index` output is asserted in tests/test_index_build.py (CPU, mock runtime) and tests/test_gpu_index.py.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .api import BwaGpuError, load_library


class Built(C.Structure):
    """bwagpu_built_t"""
    _fields_ = [("bwt", C.POINTER(C.c_uint32)), ("bwt_size", C.c_uint64), ("sa", C.POINTER(C.c_uint64)), ("n_sa", C.c_uint64),
                ("sa_intv", C.c_int), ("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("build_ms", C.c_float)]


def pack_pac(codes: np.ndarray) -> np.ndarray:
    """2-bit packing of the forward strand, base l in byte l >> 2 at bits (3 - l % 4) * 2 (bntseq.c:229); l_pac/4 + 1 bytes."""
    l_pac = int(codes.shape[0])
    # an ambiguity code left in the array would be OR-ed into its neighbours' bits: the caller replaces Ns first (bns_fasta2bntseq
    # does, bntseq.c:266,295-296) and lists them as holes
    if l_pac and int(codes.max()) > 3:
        raise ValueError("pack_pac: base codes must be 0..3 (replace ambiguity codes first and pass them to write_pac_ann_amb as holes)")
    out = np.zeros(l_pac // 4 + 1, dtype=np.uint8)
    full = l_pac // 4 * 4
    if full:
        c4 = codes[:full].reshape(-1, 4)
        out[: full // 4] = (c4[:, 0] << 6) | (c4[:, 1] << 4) | (c4[:, 2] << 2) | c4[:, 3]
    for k in range(full, l_pac):
        out[k >> 2] |= int(codes[k]) << ((3 - (k & 3)) * 2)
    return out


def build_arrays(pac: np.ndarray, l_pac: int, sa_intv: int = 32, device: int = 0, lib_path: str | None = None):
    """Run the device builder; returns (Built struct, library) -- the caller frees with L.bwagpu_built_free."""
    L = load_library(lib_path)
    L.bwagpu_index_build.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_char_p, C.c_size_t]
    L.bwagpu_built_free.argtypes = [C.c_void_p]
    pac = np.ascontiguousarray(pac, dtype=np.uint8)
    assert pac.shape[0] >= l_pac // 4 + 1
    b = Built()
    err = C.create_string_buffer(512)
    rc = L.bwagpu_index_build(pac.ctypes.data, l_pac, sa_intv, device, C.byref(b), err, 512)
    if rc != 0:
        raise BwaGpuError(f"bwagpu_index_build failed: {L.bwagpu_strerror(rc).decode()} {err.value.decode()}")
    return b, L


def write_pac_ann_amb(prefix: str, codes_or_pac: np.ndarray, l_pac: int, contigs, packed: bool = False, holes=()):
    pac = codes_or_pac if packed else pack_pac(codes_or_pac)
    with open(prefix + ".pac", "wb") as f:          # file length is l_pac/4 + 1 (+1 when l_pac % 4 == 0) + the count byte (bntseq.c:314-323)
        f.write(pac[: (l_pac + 3) // 4].tobytes())
        if l_pac % 4 == 0:
            f.write(b"\0")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write(f"{l_pac} {len(contigs)} 11\n")
        off = 0
        for name, ln in contigs:
            n_amb = sum(1 for h in holes if off <= h[0] < off + int(ln))
            f.write(f"0 {name} (null)\n{off} {int(ln)} {n_amb}\n")
            off += int(ln)
    with open(prefix + ".amb", "w") as f:
        f.write(f"{l_pac} {len(contigs)} {len(holes)}\n")
        for o, ln, ch in holes:
            f.write(f"{o} {ln} {ch}\n")


def build_index(prefix: str, codes: np.ndarray, contigs, device: int = 0, sa_intv: int = 32, lib_path: str | None = None) -> dict:
    """codes: uint8 array of the whole genome (values 0..3, contigs concatenated); contigs: [(name, length), ...]."""
    l_pac = int(codes.shape[0])
    assert sum(int(l) for _, l in contigs) == l_pac
    pac = pack_pac(codes)
    b, L = build_arrays(pac, l_pac, sa_intv, device, lib_path)
    try:
        hdr = np.array([b.primary, b.L2[1], b.L2[2], b.L2[3], b.L2[4]], dtype=np.uint64)
        with open(prefix + ".bwt", "wb") as f:
            f.write(hdr.tobytes())
            np.ctypeslib.as_array(b.bwt, shape=(int(b.bwt_size),)).tofile(f)
        with open(prefix + ".sa", "wb") as f:
            f.write(hdr.tobytes())
            f.write(np.array([b.sa_intv, b.seq_len], dtype=np.uint64).tobytes())
            np.ctypeslib.as_array(b.sa, shape=(int(b.n_sa),))[1:].tofile(f)       # sa[0] = -1 is not stored (bwt.c:404)
        info = {"l_pac": l_pac, "seq_len": int(b.seq_len), "primary": int(b.primary), "n_sa": int(b.n_sa), "build_ms": float(b.build_ms)}
    finally:
        L.bwagpu_built_free(C.byref(b))
    write_pac_ann_amb(prefix, pac, l_pac, contigs, packed=True)
    return info
