"""Plain-C struct mirrors of the reference's boundary types (ctypes + numpy views).

These are the exact layouts the C-ABI in include/bwagpu.h exchanges; they replicate
mem_opt_t (bwamem.h:52-84, 168 B), mem_alnreg_t (bwamem.h:86-104, 88 B), bwtintv_t (bwt.h:62, 32 B),
mem_pestat_t (bwamem.h:108-112, 32 B).  Sizes are asserted against the compiled reference in tests.
"""
import ctypes as C
import numpy as np


class MemOpt(C.Structure):
    _fields_ = [
        ("a", C.c_int), ("b", C.c_int),
        ("o_del", C.c_int), ("e_del", C.c_int), ("o_ins", C.c_int), ("e_ins", C.c_int),
        ("pen_unpaired", C.c_int), ("pen_clip5", C.c_int), ("pen_clip3", C.c_int),
        ("w", C.c_int), ("zdrop", C.c_int),
        ("max_mem_intv", C.c_uint64),
        ("T", C.c_int), ("flag", C.c_int), ("min_seed_len", C.c_int), ("min_chain_weight", C.c_int),
        ("max_chain_extend", C.c_int), ("split_factor", C.c_float), ("split_width", C.c_int),
        ("max_occ", C.c_int), ("max_chain_gap", C.c_int), ("n_threads", C.c_int), ("chunk_size", C.c_int),
        ("mask_level", C.c_float), ("drop_ratio", C.c_float), ("XA_drop_ratio", C.c_float),
        ("mask_level_redun", C.c_float), ("mapQ_coef_len", C.c_float), ("mapQ_coef_fac", C.c_int),
        ("max_ins", C.c_int), ("max_matesw", C.c_int), ("max_XA_hits", C.c_int), ("max_XA_hits_alt", C.c_int),
        ("mat", C.c_int8 * 25),
    ]


assert C.sizeof(MemOpt) == 168

MEM_F_PE = 0x2

ALNREG_DTYPE = np.dtype([
    ("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"),
    ("truesc", "<i4"), ("sub", "<i4"), ("alt_sc", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"),
    ("seedcov", "<i4"), ("secondary", "<i4"), ("secondary_all", "<i4"), ("seedlen0", "<i4"),
    ("ncomp_isalt", "<u4"),  # n_comp:30 (low bits), is_alt:2 (high bits)
    ("frac_rep", "<f4"), ("hash", "<u8"),
])
assert ALNREG_DTYPE.itemsize == 88

INTV_DTYPE = np.dtype([("x0", "<u8"), ("x1", "<u8"), ("x2", "<u8"), ("info", "<u8")])
SEED_DTYPE = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("_pad", "<i4")])
CHAIN_HDR_DTYPE = np.dtype([("n", "<i4"), ("rid", "<i4"), ("w", "<i4"), ("kept", "<i4"), ("is_alt", "<i4"),
                            ("first", "<i4"), ("frac_rep", "<f4"), ("seed_off", "<i4"), ("pos", "<i8")])
assert SEED_DTYPE.itemsize == 24 and CHAIN_HDR_DTYPE.itemsize == 40


class MemPestat(C.Structure):
    _fields_ = [("low", C.c_int), ("high", C.c_int), ("failed", C.c_int), ("avg", C.c_double), ("std", C.c_double)]


assert C.sizeof(MemPestat) == 32


def default_opt() -> MemOpt:
    """mem_opt_init() defaults (bwamem.c:74-110) incl. bwa_fill_scmat (bwa.c:136-145)."""
    o = MemOpt()
    o.a, o.b = 1, 4
    o.o_del = o.o_ins = 6
    o.e_del = o.e_ins = 1
    o.w, o.T, o.zdrop = 100, 30, 100
    o.pen_unpaired = 17
    o.pen_clip5 = o.pen_clip3 = 5
    o.max_mem_intv = 20
    o.min_seed_len, o.split_width, o.max_occ = 19, 10, 500
    o.max_chain_gap = o.max_ins = 10000
    o.mask_level = o.drop_ratio = 0.5
    o.XA_drop_ratio, o.split_factor = 0.8, 1.5
    o.chunk_size, o.n_threads = 10000000, 1
    o.max_XA_hits, o.max_XA_hits_alt, o.max_matesw = 5, 200, 50
    o.mask_level_redun = 0.95
    o.min_chain_weight, o.max_chain_extend = 0, 1 << 30
    o.mapQ_coef_len, o.mapQ_coef_fac = 50.0, 3
    fill_scmat(o)
    return o


def fill_scmat(o: MemOpt):
    k = 0
    for i in range(4):
        for j in range(4):
            o.mat[k] = o.a if i == j else -o.b
            k += 1
        o.mat[k] = -1
        k += 1
    for _ in range(5):
        o.mat[k] = -1
        k += 1


def pacbio_opt() -> MemOpt:
    """`-x pacbio` preset (fastmap.c:337-345) followed by update_a with a == 1 (no rescale)."""
    o = default_opt()
    o.o_del = o.o_ins = 1
    o.e_del = o.e_ins = 1
    o.b = 1
    o.split_factor = 10.0
    o.pen_clip5 = o.pen_clip3 = 0
    o.min_seed_len = 17
    o.min_chain_weight = 40
    fill_scmat(o)
    return o
