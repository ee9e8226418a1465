"""CPU: the product's from-scratch host finalize (bwa_amd/csrc/host: mark-primary, mapQ, CIGAR/NM/MD, XA, SAM records,
insert-size statistics, mate rescue, pairing) against the compiled reference.  Regions come from the reference's own
mem_align1_core, so this isolates the host code: its SAM text must equal mem_process_seqs' byte for byte."""
import ctypes as C
import numpy as np
import pytest

import refapi
import testdata
from bwa_amd import simdata
from bwa_amd.structs import default_opt, pacbio_opt

pytestmark = pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


@pytest.fixture(scope="module")
def pair():
    import hostapi
    fa, g = testdata.medium_index()
    ref, host = refapi.RefIndex(fa), hostapi.HostFinalize(fa)
    yield ref, host, g
    ref.close(); host.close()


def _check(ref, host, opt, reads, what, n_processed=0, ragged=None):
    seqs, off = ragged if ragged is not None else testdata.flat(reads)
    n = off.shape[0] - 1
    names = [f"q{i >> 1}" if (opt.flag & 2) else f"q{i}" for i in range(n)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    want = ref.process_seqs(opt, names, ASCII[seqs].tobytes(), quals, off, n_processed=n_processed)
    counts, regs = ref.align(opt, seqs, off)
    got = host.regs2sam(opt, names, seqs, quals, off, counts, regs, n_processed=n_processed)
    if got != want:
        for a, b in zip(want.split(b"\n"), got.split(b"\n")):
            assert a == b, f"{what}: first differing SAM line\nwant {a[:300]!r}\ngot  {b[:300]!r}"
    assert got == want, what


def _interleave(r1, r2):
    out = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8)
    out[0::2], out[1::2] = r1, r2
    return out


@pytest.mark.parametrize("flag,kw", [
    (0, dict(seed=301, n_frac=0.003)),
    (0x8, dict(seed=302)),                                   # -a  (MEM_F_ALL)
    (0x200 | 0x2000, dict(seed=303)),                        # -Y, XB tags
    (0x800 | 0x10, dict(seed=304, length=250, sub=0.03, dele=0.01, ins=0.01)),   # -5 -M
    (0x1000 | 0x100, dict(seed=305, length=100, sub=0.05, dele=0.008, ins=0.008)),   # -q -V
])
def test_single_end_sam(pair, flag, kw):
    ref, host, g = pair
    opt = default_opt(); opt.flag |= flag
    _check(ref, host, opt, simdata.make_reads_se(g, 6000, **kw), f"SE flag {flag:#x}", n_processed=12345)


def test_single_end_alt_contig_and_thresholds(pair):
    ref, host, g = pair
    ref.set_alt(2, 1); host.set_alt(2, 1)
    try:
        opt = default_opt(); opt.T = 20; opt.max_XA_hits = 3
        _check(ref, host, opt, simdata.make_reads_se(g, 6000, seed=306), "SE with ALT contig")
        opt.flag |= 0x2
        _check(ref, host, opt, _interleave(*simdata.make_reads_pe(g, 3000, seed=307)), "PE with ALT contig")
    finally:
        ref.set_alt(2, 0); host.set_alt(2, 0)


def test_single_end_long_and_ragged(pair):
    ref, host, g = pair
    _check(ref, host, pacbio_opt(), simdata.make_reads_long(g, 30, length=4000, seed=308), "pacbio 4 kb")
    rng = np.random.default_rng(309)
    base = simdata.make_reads_se(g, 400, length=300, seed=310)
    rag = [r[: int(rng.integers(1, 300))] for r in base] + [np.full(40, 4, dtype=np.uint8), base[0][:10]]
    _check(ref, host, default_opt(), None, "ragged", ragged=testdata.ragged(rag))


@pytest.mark.parametrize("flag,kw", [
    (0x2, dict(seed=311)),
    (0x2, dict(seed=312, sub=0.06, dele=0.01, ins=0.01)),                      # noisy: many ends unmapped -> mate rescue
    (0x2, dict(seed=313, ins_mean=230.0, ins_sd=60.0)),                         # overlapping mates, wide distribution
    (0x2 | 0x20, dict(seed=314, sub=0.04)),                                     # -S  no rescue
    (0x2 | 0x4, dict(seed=315)),                                                # -P  no pairing
    (0x2 | 0x8, dict(seed=316, sub=0.03)),                                      # -a
])
def test_paired_end_sam(pair, flag, kw):
    ref, host, g = pair
    opt = default_opt(); opt.flag |= flag
    _check(ref, host, opt, _interleave(*simdata.make_reads_pe(g, 4000, **kw)), f"PE flag {flag:#x}", n_processed=8000)


@pytest.mark.parametrize("n_processed", [(1 << 24) - 2000, (1 << 24) + 6, (1 << 25) + (1 << 24) + 1000, 20_000_000])
def test_paired_end_pair_ids_beyond_2_to_23(pair, n_processed):
    """BASELINE configs[2]/[3] hold 10-100 M pairs: mem_pair hashes `id << 8` with a 32-bit int id (bwamem_pair.c:208,248), which
    overflows -- and is sign-extended into the 64-bit XOR -- from pair id 2^23 on.  Batches that start just below, at, and far
    beyond that id (bit 31 of id << 8 set and clear) must give the reference's SAM: the tie-breaking hash decides between
    equally good pairings, which the repeat-rich genome provides."""
    ref, host, g = pair
    opt = default_opt(); opt.flag |= 0x2
    _check(ref, host, opt, _interleave(*simdata.make_reads_pe(g, 3000, seed=351, sub=0.02)), f"PE, n_processed {n_processed}", n_processed=n_processed)


def test_paired_end_mixed_orientations_and_chimeras(pair):
    """Pairs whose second mate is not reverse-complemented (FF), swapped, or taken from elsewhere: exercises the four
    orientation branches of mate rescue and the no-pairing fallback."""
    ref, host, g = pair
    r1, r2 = simdata.make_reads_pe(g, 3000, seed=317)
    rng = np.random.default_rng(318)
    kind = rng.integers(0, 4, size=r1.shape[0])
    r2 = r2.copy()
    ff = kind == 1
    r2[ff] = (3 - r2[ff])[:, ::-1]                     # same strand as mate 1
    chim = kind == 2
    r2[chim] = simdata.make_reads_se(g, int(chim.sum()), seed=319)
    opt = default_opt(); opt.flag |= 0x2
    _check(ref, host, opt, _interleave(r1, r2), "PE mixed")


def test_cigar_hints_do_not_change_the_sam(pair):
    """The finalize code may be handed precomputed region alignments (bwagpu_cigar_t records, normally from the device).  With
    host-computed records for every region the SAM text is the reference's, SE and PE; most regions are served by a record."""
    ref, host, g = pair
    for pe in (False, True):
        opt = default_opt()
        if pe:
            opt.flag |= 2
            r1, r2 = simdata.make_reads_pe(g, 3000, seed=331, sub=0.02, dele=0.003, ins=0.003)
            reads = _interleave(r1, r2)
        else:
            reads = simdata.make_reads_se(g, 6000, seed=330, sub=0.02, dele=0.003, ins=0.003)
        seqs, off = testdata.flat(reads)
        n = off.shape[0] - 1
        names = [f"q{i >> 1}" if pe else f"q{i}" for i in range(n)]
        quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
        want = ref.process_seqs(opt, names, ASCII[seqs].tobytes(), quals, off)
        counts, regs = ref.align(opt, seqs, off)
        cigs, ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
        assert (cigs["n_cigar"][regs["score"] >= opt.T] >= 0).mean() > 0.8
        assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, cig_ops=ops) == want
        # without the operation array only records that are complete in themselves (at most 6 operations, MD of at most 8 characters) are usable
        few = host.region_cigars(opt, seqs, off, counts, regs)
        assert 0.05 < (few["n_cigar"][regs["score"] >= opt.T] >= 0).mean() < 0.8
        assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=few) == want


def _noisy_mate_pairs(g, n_pairs, seed):
    """Pairs whose second mate carries ~9 % substitutions: it often has no acceptable hit of its own and must be rescued."""
    r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=seed)
    rng = np.random.default_rng(seed + 1)
    hit = rng.random(r2.shape) < 0.09
    r2 = np.where(hit, (r2 + rng.integers(1, 4, r2.shape)) % 4, r2).astype(np.uint8)
    return _interleave(r1, r2)


def test_matesw_hints_do_not_change_the_sam(pair):
    """mem_matesw may be handed precomputed ksw_align2 results (bwagpu_matesw_t records, normally from the device).  With the
    host-computed records of every task the initial region lists call for, the SAM text is the reference's; the data make
    rescue frequent."""
    ref, host, g = pair
    opt = default_opt(); opt.flag |= 2
    reads = _noisy_mate_pairs(g, 3000, seed=341)
    seqs, off = testdata.flat(reads)
    n = off.shape[0] - 1
    names = [f"q{i >> 1}" for i in range(n)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    want = ref.process_seqs(opt, names, ASCII[seqs].tobytes(), quals, off)
    counts, regs = ref.align(opt, seqs, off)
    pes = host.pestat(opt, counts, regs)
    recs = host.matesw_records(opt, seqs, off, counts, regs, pes)
    assert (recs["r"] >= 0).sum() > 300, "too few rescue alignments to mean anything"
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, msw=recs) == want
    cigs, ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, msw=recs, cig_ops=ops) == want


def test_ksw_align2_fuzz():
    """ksw_align2 (striped SSE2 in the reference) vs the host restatement: score, te, qe, score2, te2, tb, qb."""
    import hostapi
    R, H = refapi.lib(), hostapi.lib()

    class Kswr(C.Structure):
        _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]
    R.ksw_align2.restype = Kswr
    rng = np.random.default_rng(320)
    opts = [default_opt(), pacbio_opt()]
    out = (C.c_int * 7)()
    for it in range(3000):
        o = opts[it % 2]
        qlen = int(rng.integers(20, 260)); tlen = int(rng.integers(qlen, qlen + 500))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        p = int(rng.integers(0, tlen - qlen + 1))
        q = t[p:p + qlen].copy()
        m = rng.random(qlen) < rng.choice([0.03, 0.1, 0.21])
        q[m] = rng.integers(0, 5, size=int(m.sum())).astype(np.uint8)
        if rng.random() < 0.5:
            c = int(rng.integers(5, qlen - 5)); q = np.concatenate([q[:c], q[c + int(rng.integers(1, 4)):], rng.integers(0, 4, size=4).astype(np.uint8)])[:qlen]
        if rng.random() < 0.3:   # a second, weaker copy of the query elsewhere in the target -> score2/te2
            p2 = int(rng.integers(0, tlen - qlen + 1)); half = qlen // 2
            t[p2:p2 + half] = q[:half] % 4
        q = np.ascontiguousarray(q); t = np.ascontiguousarray(t)
        xtra = 0x40000 | 0x80000 | (0x10000 if qlen * o.a < 250 else 0) | (19 * o.a)
        if it % 7 == 0:
            xtra &= ~0x10000
        a = R.ksw_align2(qlen, q.ctypes.data_as(C.c_void_p), tlen, t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, xtra, None)
        H.bwamem_host_ksw_align2(qlen, q.ctypes.data_as(C.c_void_p), tlen, t.ctypes.data_as(C.c_void_p), C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, xtra, out)
        want = [a.score, a.te, a.qe, a.score2, a.te2, a.tb, a.qb]
        assert want == list(out), f"case {it}: qlen {qlen} tlen {tlen} xtra {xtra:#x}: reference {want} host {list(out)}"


def test_paired_end_reads_with_a_base_of_code_5(pair):
    """A '-' among the bases is code 5 (nst_nt4_table), whose letter in the SAM record is the NUL that ends "ACGTN": the reference fputs() a
    read's records, so that read's text stops there.  The chunked finalize writes a chunk's records into one buffer and falls back to
    read-by-read text only for chunks that hold such a read: same bytes as the reference, for either mate, both mates and neither."""
    ref, host, g = pair
    opt = default_opt(); opt.flag |= 0x2
    reads = _interleave(*simdata.make_reads_pe(g, 600, seed=331))
    rng = np.random.default_rng(332)
    for i in rng.choice(reads.shape[0], size=40, replace=False):
        reads[i, int(rng.integers(0, reads.shape[1]))] = 5
    reads[100, 7] = 5; reads[101, 140] = 5                     # (both mates of one pair)
    seqs, off = testdata.flat(reads)
    names = [f"q{i >> 1}" for i in range(reads.shape[0])]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    counts, regs = ref.align(opt, seqs, off)
    want = ref.regs2sam(opt, names, seqs.tobytes(), quals, off, counts, regs)      # (the reference's worker2 on the same regions and base codes)
    for threads in (1, 4):
        got = host.regs2sam(opt, names, seqs, quals, off, counts, regs, n_threads=threads)
        assert got == want, f"{threads} thread(s)"
    assert want.count(b"\n") < 2 * 600 and b"\0" not in want    # (the cut records have no line end of their own)
