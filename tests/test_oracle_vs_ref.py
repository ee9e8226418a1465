"""CPU: the plain-C oracle against the compiled, unmodified reference (oracle/_ref/libbwaref.so).
Skipped where oracle/_ref has not been built (it needs /root/reference)."""
import ctypes as C
import numpy as np
import pytest

import refapi
import testdata
from cmputil import assert_regs_equal, golden_opts
from bwa_amd import simdata
from bwa_amd.structs import default_opt, pacbio_opt

pytestmark = pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def pair():
    import orcapi
    fa, g = testdata.medium_index()
    ref, orc = refapi.RefIndex(fa), orcapi.OrcIndex(fa)
    yield ref, orc, g
    ref.close()
    orc.close()


def test_struct_sizes():
    refapi.lib()   # asserts sizeof(mem_opt_t) == 168, sizeof(mem_alnreg_t) == 88, ... against the compiled reference


def test_occ4_and_sa(pair):
    ref, orc, _ = pair
    L = refapi.lib()
    L.bwt_occ4.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
    L.bwt_sa.restype = C.c_uint64
    L.bwt_sa.argtypes = [C.c_void_p, C.c_uint64]
    bwt = L.refshim_idx_bwt(ref.h)
    rng = np.random.default_rng(1)
    ks = list(rng.integers(0, ref.seq_len + 1, size=20000)) + [0, 1, ref.primary - 1, ref.primary, ref.primary + 1, ref.seq_len - 1, ref.seq_len, 2 ** 64 - 1]
    out = np.zeros(4, dtype=np.uint64)
    for k in ks:
        L.bwt_occ4(bwt, C.c_uint64(int(k)), out.ctypes.data)
        assert np.array_equal(out, orc.occ4(int(k))), f"occ4({k})"
    for k in list(rng.integers(0, ref.seq_len + 1, size=3000)) + [0, ref.primary, ref.seq_len]:
        assert L.bwt_sa(bwt, int(k)) == orc.sa(int(k)), f"sa({k})"


@pytest.mark.parametrize("name,n,kw,oname", [
    ("se150", 6000, dict(seed=11, n_frac=0.002), "default"),
    ("se100_noisy", 2000, dict(length=100, seed=12, sub=0.04, dele=0.006, ins=0.006), "default"),
    ("se250_odd", 1500, dict(length=250, seed=13, sub=0.03, dele=0.005, ins=0.005), "odd"),
    ("se30", 2000, dict(length=30, seed=14), "default"),
])
def test_regs_short_reads(pair, name, n, kw, oname):
    ref, orc, g = pair
    reads = simdata.make_reads_se(g, n, **kw)
    seqs, off = testdata.flat(reads)
    opt = golden_opts()[oname]
    assert_regs_equal(*ref.align(opt, seqs, off), *orc.align(opt, seqs, off), name)


def test_regs_long_reads_and_ragged(pair):
    ref, orc, g = pair
    lr = simdata.make_reads_long(g, 25, length=5000, seed=15)
    seqs, off = testdata.flat(lr)
    assert_regs_equal(*ref.align(pacbio_opt(), seqs, off), *orc.align(pacbio_opt(), seqs, off), "pacbio 5 kb")
    rng = np.random.default_rng(16)
    base = simdata.make_reads_se(g, 300, length=400, seed=17)
    rag = [r[: int(rng.integers(1, 400))] for r in base] + [np.zeros(0, dtype=np.uint8), np.full(50, 4, dtype=np.uint8)]
    seqs, off = testdata.ragged(rag)
    assert_regs_equal(*ref.align(default_opt(), seqs, off), *orc.align(default_opt(), seqs, off), "ragged")


def test_regs_alt_contig(pair):
    ref, orc, g = pair
    ref.set_alt(2, 1); orc.set_alt(2, 1)
    try:
        seqs, off = testdata.flat(simdata.make_reads_se(g, 4000, seed=18))
        assert_regs_equal(*ref.align(default_opt(), seqs, off), *orc.align(default_opt(), seqs, off), "alt")
    finally:
        ref.set_alt(2, 0); orc.set_alt(2, 0)


def test_ksw_extend2_fuzz():
    import orcapi
    R, O = refapi.lib(), orcapi.lib()
    rng = np.random.default_rng(3)
    from bwa_amd.structs import default_opt, pacbio_opt
    mats = [default_opt(), pacbio_opt()]
    outs_r = [C.c_int() for _ in range(5)]
    outs_o = [C.c_int() for _ in range(5)]
    for it in range(4000):
        o = mats[it & 1]
        qlen, tlen = int(rng.integers(1, 200)), int(rng.integers(1, 260))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = t[:qlen].copy() if qlen <= tlen else np.concatenate([t, rng.integers(0, 4, size=qlen - tlen).astype(np.uint8)])
        m = rng.random(qlen) < rng.choice([0.02, 0.1, 0.3])
        q[m] = rng.integers(0, 5, size=int(m.sum())).astype(np.uint8)
        if rng.random() < 0.5 and qlen > 10:   # an indel
            p = int(rng.integers(1, qlen - 1)); q = np.concatenate([q[:p], q[p + int(rng.integers(1, 6)):], rng.integers(0, 4, size=8).astype(np.uint8)])[:qlen]
        q = np.ascontiguousarray(q); qlen = len(q)
        w, h0, zd, eb = int(rng.integers(1, 201)), int(rng.integers(1, 300)), int(rng.choice([0, 20, 100])), int(rng.choice([0, 5]))
        args = lambda outs: (qlen, q.ctypes.data_as(C.c_void_p), tlen, t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, w, eb, zd, h0) + tuple(C.byref(x) for x in outs)
        a = R.ksw_extend2(*args(outs_r))
        b = O.orc_ksw_extend2(*args(outs_o))
        assert a == b and [x.value for x in outs_r] == [x.value for x in outs_o], f"case {it}: {a} {b} {[x.value for x in outs_r]} {[x.value for x in outs_o]}"


def test_ksw_global2_fuzz():
    import orcapi
    R, O = refapi.lib(), orcapi.lib()
    rng = np.random.default_rng(4)
    from bwa_amd.structs import default_opt, pacbio_opt
    mats = [default_opt(), pacbio_opt()]
    for it in range(1500):
        o = mats[it & 1]
        tlen = int(rng.integers(1, 150))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = t.copy()
        m = rng.random(tlen) < 0.08
        q[m] = rng.integers(0, 4, size=int(m.sum())).astype(np.uint8)
        if tlen > 12 and rng.random() < 0.6:
            p = int(rng.integers(1, tlen - 8)); q = np.concatenate([q[:p], q[p + int(rng.integers(1, 5)):]])
        if rng.random() < 0.4:
            p = int(rng.integers(0, len(q))); q = np.concatenate([q[:p], rng.integers(0, 4, size=int(rng.integers(1, 5))).astype(np.uint8), q[p:]])
        q = np.ascontiguousarray(q); qlen = len(q)
        w = abs(tlen - qlen) + 3 + int(rng.integers(0, 30))
        nr, no = C.c_int(), C.c_int()
        cr, co = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        base = (qlen, q.ctypes.data_as(C.c_void_p), tlen, t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, w)
        a = R.ksw_global2(*base, C.byref(nr), C.byref(cr))
        b = O.orc_ksw_global2(*base, C.byref(no), C.byref(co))
        assert a == b and nr.value == no.value and [cr[i] for i in range(nr.value)] == [co[i] for i in range(no.value)], f"case {it}"
        assert a == R.ksw_global2(*base, None, None) == O.orc_ksw_global2(*base, None, None)
