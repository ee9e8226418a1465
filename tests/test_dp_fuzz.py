"""Differential fuzz of the device DP routines through bwagpu_debug_dp (one wavefront per case, set up as the product kernels set
them up) against the compiled reference's exported ksw_extend2 (ksw.c:416), ksw_global2 (ksw.c:540) and ksw_align2 (ksw.c:379):
k_extend_wave's wave_ksw_extend2 in both modes (incl. the diagonal shortcut, the one- and two-column row forms, multi-pass rows and
the stale-cell rule of SURVEY App. A.10), k_cigar's wave_ksw_global2, k_dedup_wave's score-only ring form, k_matesw_sw's
ksw_align2 restatement, and the long-segment traceback kernel.

CPU: a few hundred cases per routine on the mock runtime (tests/hostsim).  -m gpu: 5000 per routine on the device."""
import ctypes as C
import os

import numpy as np
import pytest

import refapi
import testdata
from bwa_amd.api import BwaGpu, DP_CASE_DTYPE
from bwa_amd.structs import default_opt, pacbio_opt

pytestmark = pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")

KSW_XBYTE, KSW_XSTOP, KSW_XSUBO, KSW_XSTART = 0x10000, 0x20000, 0x40000, 0x80000


class Kswr(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


def _opts():
    a = default_opt()
    b = pacbio_opt()
    c = default_opt()     # asymmetric gap costs, other clipping penalties
    c.o_del, c.e_del, c.o_ins, c.e_ins, c.zdrop = 8, 2, 5, 1, 40
    d = default_opt()     # scaled scores (-A 2), no z-drop
    d.a, d.b, d.o_del, d.o_ins, d.e_del, d.e_ins, d.zdrop = 2, 8, 12, 12, 2, 2, 0
    for i in range(4):
        for j in range(4):
            d.mat[i * 5 + j] = 2 if i == j else -8
    return [a, b, c, d]


def _mutate(rng, t, sub, indel):
    q = t.copy()
    m = rng.random(len(q)) < sub
    q[m] = rng.integers(0, 5, size=int(m.sum())).astype(np.uint8)       # incl. N
    out, i = [], 0
    while i < len(q):
        r = rng.random()
        if r < indel:
            i += int(rng.integers(1, 6))                                # deletion from the query
            continue
        if r < 2 * indel:
            out.extend(rng.integers(0, 4, size=int(rng.integers(1, 6))).tolist())
        out.append(int(q[i])); i += 1
    return np.asarray(out if out else [0], dtype=np.uint8)


def _seen(t, flags):
    s = t
    if flags & 4:
        s = (3 - s[::-1]).astype(np.uint8)
    if flags & 2:
        s = s[::-1]
    return np.ascontiguousarray(s)


class CaseSet:
    def __init__(self):
        self.seqs, self.cases, self.pos, self.py = [], [], 0, []

    def add(self, q, t, w, h0, eb, flags):
        q = np.ascontiguousarray(q, dtype=np.uint8); t = np.ascontiguousarray(t, dtype=np.uint8)
        self.cases.append((self.pos, len(q), self.pos + len(q), len(t), w, h0, eb, flags))
        self.seqs += [q, t]; self.pos += len(q) + len(t)
        self.py.append((q[::-1].copy() if flags & 1 else q, _seen(t, flags), w, h0, eb))

    def arrays(self):
        return np.array(self.cases, dtype=DP_CASE_DTYPE), np.concatenate(self.seqs)


def extend_cases(rng, n, max_len):
    cs = CaseSet()
    ws = [1, 5, 100, 400]
    for it in range(n):
        kind = it % 5
        tlen = int(rng.integers(1, max_len))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        if kind == 4:      # the stale-cell family: large h0, small band, a run of mismatches first, then a good diagonal
            h0 = int(rng.integers(40, 250)); w = int(rng.integers(3, 30))
            tlen = max(tlen, 60); t = rng.integers(0, 4, size=tlen).astype(np.uint8)
            q = _mutate(rng, t, 0.02, 0.0)[: max_len - 1]
            k = min(len(q) - 1, max(1, (h0 - int(rng.integers(1, 9))) // 4))
            q[:k] = (t[:k] + 1 + rng.integers(0, 3, size=k)) % 4
            cs.add(q, t, w, h0, int(rng.choice([0, 5])), 0)
            continue
        q = _mutate(rng, t, float(rng.choice([0.0, 0.02, 0.1, 0.3])), float(rng.choice([0.0, 0.005, 0.03])))[: max_len - 1]
        if kind == 3:      # unrelated sequences: the band collapses early
            q = rng.integers(0, 5, size=int(rng.integers(1, max_len))).astype(np.uint8)
        w = int(rng.choice(ws)) if kind != 2 else int(rng.integers(1, 201))
        cs.add(q, t, w, int(rng.integers(1, 300)), int(rng.choice([0, 5, 9])), int(rng.integers(0, 8)))
    return cs


def wide_band_cases(rng, n):
    """Rows of exactly 64 and exactly 128 live columns (and their neighbours): a large h0 keeps every column of the first-row ramp alive, so
    with beg = 0 the band grows by one column per row from w + 1 to qlen and passes through every width on the way -- the widths at which
    the kernel changes its row form (one column per lane, two columns per lane, several passes)."""
    cs = CaseSet()
    for it in range(n):
        wide = it % 4 != 0
        qlen = int(rng.integers(128, 135)) if wide else int(rng.integers(64, 70))
        w = int(rng.choice([100, 127, 110])) if wide else int(rng.choice([40, 63, 50]))
        if it % 4 == 1:
            qlen, w = 129, int(rng.choice([100, 90, 64]))     # the row after the 128-column one reads column 128 as the query's last
        tlen = int(rng.integers(qlen - 10, qlen + 20))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = (rng.integers(0, 4, size=qlen).astype(np.uint8) if it % 3 else _mutate(rng, np.resize(t, qlen), 0.25, 0.02)[:qlen])
        cs.add(q, t, w, int(rng.integers(qlen + 80, qlen + 200)), int(rng.choice([0, 5])), int(rng.integers(0, 8)))
    return cs


def window_cases(rng, n, max_len):
    """Long, good diagonals for the register-resident row forms of the short-read kernel (dev_extw.h, "window rows"): hundreds of rows, so that a
    window of 64 (or 128) columns is re-based several times, the parked rows are evaluated every 32 rows and at the 64-row boundaries of the
    reference bases, and the last rows touch the query's end (to-end score).  Narrow bands (w 4..40) make the band's clamps move it by exactly one
    column per row; h0 and the divergence set how wide the live band is, so the rows drift between the one-column and the two-column form; a
    diverged stretch in the middle lets the zero-trimming bite and, with the option sets that have a small zdrop, ends the extension inside a
    run of parked rows."""
    cs = CaseSet()
    for it in range(n):
        qlen = int(rng.integers(max_len // 2, max_len - 1))
        tlen = qlen + int(rng.integers(-30, 60))
        t = rng.integers(0, 4, size=max(tlen, 40)).astype(np.uint8)
        q = _mutate(rng, t, float(rng.choice([0.0, 0.01, 0.03, 0.06])), float(rng.choice([0.0, 0.004, 0.015])))
        q = np.resize(q, qlen) if len(q) < qlen else q[:qlen]
        if it % 3 == 1 and qlen > 90:        # an unrelated stretch: scores decay, the band shrinks, possibly z-drop
            a = int(rng.integers(30, qlen - 50)); b = a + int(rng.integers(8, 40))
            q[a:b] = rng.integers(0, 4, size=b - a)
        w = int(rng.choice([4, 8, 20, 33, 40, 64, 100]))
        h0 = int(rng.choice([15, 25, 40, 70, 120]))
        cs.add(q, t, w, h0, int(rng.choice([0, 5, 9])), int(rng.integers(0, 8)))
    return cs


def very_wide_band_cases(rng, n):
    """The same for the ring form's multi-pass rows at widths beyond 256 columns: the band
    grows through 255..257 (and, for the longest queries, 511..513) live columns; every other case has an unrelated stretch in the middle,
    so that the zero-trimming cuts the band inside a pass and the stale-cell rule meets slots of a block the band has left."""
    cs = CaseSet()
    for it in range(n):
        qlen = int(rng.choice([250, 256, 257, 258, 300, 513, 530]))
        w = int(rng.choice([130, 200, 255, 256, 257, 300]))
        tlen = int(rng.integers(qlen - 10, qlen + 30))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = _mutate(rng, np.resize(t, qlen), float(rng.choice([0.02, 0.1, 0.25])), 0.02)[:qlen]
        if it % 2:
            p0 = int(rng.integers(40, len(q) - 60)); q = q.copy(); q[p0:p0 + 40] = rng.integers(0, 4, size=40)
        h0 = int(rng.integers(len(q) + 80, len(q) + 200)) if it % 3 else int(rng.integers(20, 120))
        cs.add(q, t, w, h0, int(rng.choice([0, 5])), int(rng.integers(0, 8)))
    return cs


def ref_extend(o, q, t, w, h0, eb):
    R = refapi.lib()
    outs = [C.c_int() for _ in range(5)]
    sc = R.ksw_extend2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins,
                       w, eb, o.zdrop, h0, *[C.byref(x) for x in outs])
    qle, tle, gtle, gscore, max_off = [x.value for x in outs]
    return [sc, qle, tle, gtle, gscore, max_off]


def py_extend_stale_differs(o, q, t, w, h0, eb):
    """Does this case read a never-written column beyond every `end` reached so far with a non-zero first-row value?  (A restatement of
    ksw.c:430-505's band bookkeeping only: enough to know that the stale-cell rule is exercised, not a checker of results.)"""
    qlen, tlen = len(q), len(t)
    oe_ins = o.o_ins + o.e_ins
    ramp = [h0, max(h0 - oe_ins, 0)]
    for j in range(2, qlen + 1):
        ramp.append(ramp[-1] - o.e_ins if ramp[-1] > o.e_ins else 0)
    mat = [o.mat[i] for i in range(25)]
    mx = max(mat)
    lim = int((qlen * mx + eb - o.o_ins) / o.e_ins + 1.); w = min(w, max(lim, 1))
    lim = int((qlen * mx + eb - o.o_del) / o.e_del + 1.); w = min(w, max(lim, 1))
    H = ramp[:] + [0]; E = [0] * (qlen + 2); written = [False] * (qlen + 2)
    beg, end, mxs, max_i, max_j = 0, qlen, h0, -1, -1
    hit = False
    for i in range(tlen):
        f, m, mj = 0, 0, -1
        beg = max(beg, i - w); end = min(end, i + w + 1, qlen)
        h1 = max(h0 - (o.o_del + o.e_del * (i + 1)), 0) if beg == 0 else 0
        for j in range(beg, end):
            if i > 0 and not written[j] and H[j] != 0:
                hit = True
            M, e = H[j], E[j]
            H[j] = h1; written[j] = True
            M = M + mat[t[i] * 5 + q[j]] if M else 0
            h = max(M, e, f); h1 = h
            if h >= m:
                m, mj = h, j
            E[j] = max(e - o.e_del, M - (o.o_del + o.e_del), 0)
            f = max(f - o.e_ins, M - oe_ins, 0)
        H[end] = h1; E[end] = 0; written[end] = True
        if m == 0:
            break
        if m > mxs:
            mxs, max_i, max_j = m, i, mj
        elif o.zdrop > 0:
            di, dj = i - max_i, mj - max_j
            if (di > dj and mxs - m - (di - dj) * o.e_del > o.zdrop) or (di <= dj and mxs - m - (dj - di) * o.e_ins > o.zdrop):
                break
        j = beg
        while j < end and H[j] == 0 and E[j] == 0:
            j += 1
        beg = j
        j = end
        while j >= beg and H[j] == 0 and E[j] == 0:
            j -= 1
        end = min(j + 2, qlen)
    return hit


def run_extend(dev, kind, n, max_len, seed, need_stale, very_wide=0):
    rng = np.random.default_rng(seed)
    fast = stale = 0
    for oi, o in enumerate(_opts()):
        cs = extend_cases(rng, n // 4, max_len)
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, kind, cases, seqs)
        for k, (q, t, w, h0, eb) in enumerate(cs.py):
            exp = ref_extend(o, q, t, w, h0, eb)
            assert out[k, :6].tolist() == exp, f"kind {kind} opt {oi} case {k}: device {out[k, :8].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} w {w} h0 {h0} eb {eb} flags {cases['flags'][k]})"
            if need_stale and k % 5 == 4 and k < 400:
                stale += py_extend_stale_differs(o, q, t, w, h0, eb)
        fast += int(out[:, 6].sum())
        cs = wide_band_cases(rng, max(16, n // 12))
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, kind, cases, seqs)
        for k, (q, t, w, h0, eb) in enumerate(cs.py):
            exp = ref_extend(o, q, t, w, h0, eb)
            assert out[k, :6].tolist() == exp, f"kind {kind} opt {oi} wide-band case {k}: device {out[k, :8].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} w {w} h0 {h0} eb {eb} flags {cases['flags'][k]})"
        cs = window_cases(rng, max(24, n // 8), max_len)
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, kind, cases, seqs)
        for k, (q, t, w, h0, eb) in enumerate(cs.py):
            exp = ref_extend(o, q, t, w, h0, eb)
            assert out[k, :6].tolist() == exp, f"kind {kind} opt {oi} window case {k}: device {out[k, :8].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} w {w} h0 {h0} eb {eb} flags {cases['flags'][k]})"
        if very_wide:
            cs = very_wide_band_cases(rng, very_wide)
            cases, seqs = cs.arrays()
            out = dev.debug_dp(o, kind, cases, seqs)
            for k, (q, t, w, h0, eb) in enumerate(cs.py):
                exp = ref_extend(o, q, t, w, h0, eb)
                assert out[k, :6].tolist() == exp, f"kind {kind} opt {oi} very wide case {k}: device {out[k, :8].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} w {w} h0 {h0} eb {eb} flags {cases['flags'][k]})"
    assert fast > 0, "no case took the diagonal shortcut"
    if need_stale:
        assert stale > 0, "no case exercised the stale-cell rule"


def pack_cases(rng, n, cols):
    """Cases for the packed extension routine of k_ext_pack (dev_extp.h): queries of up to cols - 1 bases (and a few beyond: the routine must say
    'not mine'), the bands mem_chain2aln asks for (w = 100, the default) and odd ones, h0 from a seed's length; targets longer than the query
    by up to 120 rows, so that the rows after the query's end -- deletion tails that leave the band on the left one column per row -- are many;
    unrelated stretches (the zero-trimming), exact copies (the diagonal rule), N bases, both strands and directions."""
    cs = CaseSet()
    for it in range(n):
        kind = it % 6
        qlen = int(rng.integers(1, cols + (6 if it % 17 == 0 else 0)))
        tlen = max(1, qlen + int(rng.integers(-12, 121)))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        src = np.resize(t, qlen) if tlen < qlen else t[:qlen]
        if kind == 0:
            q = src.copy()
        elif kind == 1:
            q = rng.integers(0, 5, size=qlen).astype(np.uint8)
        else:
            q = _mutate(rng, src, float(rng.choice([0.01, 0.03, 0.1, 0.25])), float(rng.choice([0.0, 0.01, 0.04])))
            q = np.resize(q, qlen) if len(q) < qlen else q[:qlen]
        if kind == 5 and qlen > 24:
            a = int(rng.integers(4, qlen - 12)); q = q.copy(); q[a:a + 10] = rng.integers(0, 4, size=10)
        w = 100 if it % 3 else int(rng.choice([1, 3, 10, 31, 63, 64, 127, 200, 400]))
        h0 = int(rng.integers(19, 140)) if it % 4 else int(rng.integers(1, 300))
        cs.add(q, t, w, h0, int(rng.choice([5, 5, 5, 9, 0])), int(rng.integers(0, 8)))      # (end bonus 5 = mem_chain2aln's pen_clip: with 0 the band limit of ksw.c:436-443 falls below qlen - 1 and the routine declines)
    return cs


def run_extend_pack(dev, kind, n, seed):
    """kind 6 / 7: out[7] says whether the routine vouches for its result; where it does, the result is ksw_extend2's.  (Where it does not the
    product runs the one-wave routine, kind 0.)"""
    rng = np.random.default_rng(seed)
    cols = 64 if kind == 6 else 128
    vouched = total = fast = 0
    for oi, o in enumerate(_opts()):
        cs = pack_cases(rng, n // 4, cols)
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, kind, cases, seqs)
        for k, (q, t, w, h0, eb) in enumerate(cs.py):
            total += 1
            if len(q) > cols - 1:
                assert out[k, 7] == 0, f"kind {kind} opt {oi} case {k}: a query of {len(q)} bases was accepted"
            if not out[k, 7]:
                continue
            vouched += 1
            exp = ref_extend(o, q, t, w, h0, eb)
            assert out[k, :6].tolist() == exp, f"kind {kind} opt {oi} case {k}: device {out[k, :8].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} w {w} h0 {h0} eb {eb} flags {cases['flags'][k]})"
        fast += int(out[:, 6].sum())
    assert vouched > total * 0.3, f"only {vouched} of {total} cases were answered"
    assert fast > 0, "no case took the diagonal shortcut"


def global_cases(rng, n, max_len, max_cols, big_gaps=False, wide=()):
    cs = CaseSet()
    for it in range(n):
        tlen = int(rng.integers(1, max_len))
        t = rng.integers(0, 4, size=tlen).astype(np.uint8)
        q = _mutate(rng, t, float(rng.choice([0.0, 0.03, 0.12])), float(rng.choice([0.0, 0.01, 0.04])))[:max_len]
        if big_gaps and it % 3 == 0 and len(q) > 150:      # a gap of 20-70 columns: the traceback leaves a tile row's window
            p, g = int(rng.integers(40, len(q) - 80)), int(rng.integers(20, 70))
            q = np.concatenate([q[:p], q[p + g:]]) if it % 2 else np.concatenate([q[:p], rng.integers(0, 4, size=g).astype(np.uint8), q[p:]])
        dl = abs(len(q) - tlen)
        w = dl + 3 + int(rng.integers(0, 40)) if it % 3 else dl + 3
        if it % 5 == 0 and not big_gaps:     # bands of exactly 63 .. 65, 127 .. 129, 191 .. 193 columns: the widths at which a row takes another pass of the wave
            w = max(w, int(rng.choice([31, 32, 63, 64, 95, 96])))
        if wide and it % 2 == 1:             # (the four-columns-per-lane form takes another pass every 256 columns)
            w = max(w, int(rng.choice(wide)))
        if min(len(q), 2 * w + 1) > max_cols:
            w = max(dl + 3, (max_cols - 1) // 2)
        if min(len(q), 2 * w + 1) > max_cols:
            q = q[:max_cols]; w = abs(len(q) - tlen) + 3
        cs.add(q, t, w, 0, 0, int(rng.integers(0, 8)))
    return cs


def ref_global(o, q, t, w):
    R = refapi.lib()
    n = C.c_int(); cg = C.POINTER(C.c_uint32)()
    sc = R.ksw_global2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, w, C.byref(n), C.byref(cg))
    ops = [cg[i] for i in range(n.value)]
    refapi.lib().refshim_free(cg)
    return sc, ops


def run_global(dev, kind, n, max_len, max_cols, seed, wide=()):
    rng = np.random.default_rng(seed)
    served = 0
    for oi, o in enumerate(_opts()):
        cs = global_cases(rng, n // 4, max_len, max_cols, big_gaps=kind == 5, wide=wide)
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, kind, cases, seqs)
        for k, (q, t, w, _, _) in enumerate(cs.py):
            sc, ops = ref_global(o, q, t, w)
            if kind == 3:
                if out[k, 1] == -2:
                    continue
                assert out[k, 0] == sc, f"kind 3 opt {oi} case {k}: {out[k, 0]} vs {sc}"
                served += 1
                continue
            if kind == 5:      # the long-segment kernel: any number of operations (the entry returns the first 70)
                if out[k, 1] == -2:
                    continue
                m = min(70, len(ops))
                assert out[k, 0] == sc and out[k, 1] == len(ops) and out[k, 2:2 + m].astype(np.uint32).tolist() == ops[:m], \
                    f"kind 5 opt {oi} case {k}: device {out[k, :2 + m].tolist()} reference {sc} {len(ops)} {ops[:m]} (qlen {len(q)} tlen {len(t)} w {w})"
                served += 1
                continue
            if out[k, 1] == -2 or (out[k, 1] == -1 and len(ops) > 64):
                continue
            assert out[k, 0] == sc and out[k, 1] == len(ops) and out[k, 2:2 + len(ops)].astype(np.uint32).tolist() == ops, \
                f"kind {kind} opt {oi} case {k}: device {out[k, :2 + max(0, out[k, 1])].tolist()} reference {sc} {ops} (qlen {len(q)} tlen {len(t)} w {w})"
            served += 1
    assert served > n * 0.6, f"only {served} of {n} cases served by the kernel"


def run_align2(dev, n, seed):
    R = refapi.lib()
    R.ksw_align2.restype = Kswr
    rng = np.random.default_rng(seed)
    n_sub = n_start = 0
    for oi, o in enumerate(_opts()):
        cs = CaseSet(); xs = []
        for it in range(n // 4):
            qlen = int(rng.integers(20, 260 if it % 7 else 500))
            if it % 5 == 0:      # padded query lengths around the two instances' column counts and around multiples of the wave
                qlen = int(rng.choice([63, 64, 65, 127, 128, 129, 175, 176, 177, 191, 192, 193, 255, 256, 257, 496, 500]))
            tlen = int(rng.integers(qlen // 2 + 1, 900))
            t = rng.integers(0, 4, size=tlen).astype(np.uint8)
            p = int(rng.integers(0, max(1, tlen - qlen)))
            q = _mutate(rng, t[p:p + qlen], float(rng.choice([0.03, 0.1, 0.21])), float(rng.choice([0.0, 0.01, 0.03])))[:500]
            if it % 5 == 0 and len(q) != qlen:
                q = np.concatenate([q, rng.integers(0, 4, size=qlen).astype(np.uint8)])[:qlen]
            if it % 6 == 0 and tlen > 2 * qlen + 40:      # a second, weaker copy: score2 / te2
                t[-qlen:] = np.minimum(_mutate(rng, t[p:p + qlen], 0.08, 0.0), 3)
            if it % 9 == 0:
                q = rng.integers(0, 5, size=len(q)).astype(np.uint8)
            minsc = int(rng.choice([19, 25, 40])) * o.a
            xtra = KSW_XSUBO | KSW_XSTART | (KSW_XBYTE if len(q) * o.a < 250 else 0) | minsc
            if it % 11 == 0:
                xtra &= ~KSW_XSTART
            cs.add(q, t, 0, xtra, 0, int(rng.integers(0, 8)))
            xs.append(xtra)
        cases, seqs = cs.arrays()
        out = dev.debug_dp(o, 4, cases, seqs)
        for k, (q, t, _, xtra, _) in enumerate(cs.py):
            r = R.ksw_align2(len(q), q.ctypes.data_as(C.c_void_p), len(t), t.ctypes.data_as(C.c_void_p), 5, C.cast(o.mat, C.c_void_p), o.o_del, o.e_del, o.o_ins, o.e_ins, xtra, None)
            exp = [r.score, r.te, r.qe, r.score2, r.te2, r.tb, r.qb]
            assert out[k, :7].tolist() == exp, f"align2 opt {oi} case {k}: device {out[k, :7].tolist()} reference {exp} (qlen {len(q)} tlen {len(t)} xtra {xtra:#x})"
            n_sub += r.score2 > 0; n_start += r.tb >= 0
    assert n_sub > 0 and n_start > 0


# ---- CPU: the product source under the mock runtime --------------------------------------------------------------------------
@pytest.fixture(scope="module")
def sim():
    import hostsim_build
    prefix, _ = testdata.small_index()
    s = BwaGpu(prefix, lib_path=hostsim_build.build())
    yield s
    s.close()


def test_sim_extend_fuzz(sim):
    run_extend(sim, 0, 240, 160, 11, need_stale=True)


def test_sim_extend_ring_fuzz(sim):
    """Ring mode (long reads) in both row forms: a row's band in one pass with four columns per lane (option ext_blk, the default since round 6) and one pass
    per 64 columns; bands of 200-255 columns and an unrelated stretch in the middle take the one-pass form through its widest rows and its trimming."""
    assert sim.get_option("ext_blk") == -1
    run_extend(sim, 1, 120, 160, 12, need_stale=False)
    run_extend(sim, 1, 60, 400, 17, need_stale=False, very_wide=6)
    sim.set_option("ext_blk", 0)
    try:
        run_extend(sim, 1, 120, 160, 12, need_stale=False)
    finally:
        sim.set_option("ext_blk", -1)


def test_sim_ring_global_both_forms(sim):
    """The score-only ring aligner of k_dedup_wave.  Default (option dedup_blk): four adjacent columns per lane, one scan and one ordering
    point per 256 columns of a row; dedup_blk = 0: one column per lane, a pass per 64 columns.  Same scores as the reference's ksw_global2 either
    way -- short segments, bands at the widths where a row takes a second and a third pass (127..129, 255..257 columns on either side),
    segments of up to 900 bases, reverse-strand geometry and N bases included."""
    assert sim.get_option("dedup_blk") == -1
    run_global(sim, 3, 80, 150, 1 << 30, seed=31)
    run_global(sim, 3, 48, 900, 1 << 30, seed=32, wide=(63, 64, 126, 127, 128, 129, 130, 200, 255, 256, 257, 300))
    sim.set_option("dedup_blk", 0)
    try:
        run_global(sim, 3, 40, 150, 1 << 30, seed=33)
        run_global(sim, 3, 16, 900, 1 << 30, seed=34, wide=(63, 64, 127, 128, 129, 255, 256, 257))
    finally:
        sim.set_option("dedup_blk", -1)


def test_sim_ring_extension_wide_bands(sim):
    """The ring-mode extension (long reads) with bands that grow through 255..257 and 511..513 live columns -- five to nine passes of 64
    columns per row, the zero-trimming cutting the band inside a pass, stale slots of passes the band has left."""
    run_extend(sim, 1, 60, 400, seed=41, need_stale=False, very_wide=10)


def test_sim_extend_pack_fuzz(sim):
    """The packed extension routine (four extensions per wavefront, a DPP row of 16 lanes each; four and eight columns per lane)."""
    run_extend_pack(sim, 6, 240, seed=51)
    run_extend_pack(sim, 7, 200, seed=52)


def test_sim_global_fuzz(sim):
    run_global(sim, 2, 160, 150, 192, 13)
    run_global(sim, 3, 80, 150, 1 << 30, 14)
    run_global(sim, 5, 48, 420, 1900, 16)


def test_sim_align2_fuzz(sim):
    run_align2(sim, 40, 15)


# ---- GPU ----------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def gpu():
    prefix, _ = testdata.small_index()
    g = BwaGpu(prefix)
    yield g
    g.close()


@pytest.mark.gpu
def test_gpu_extend_fuzz(gpu):
    run_extend(gpu, 0, 5000, 400, 21, need_stale=True)
    run_extend(gpu, 0, 1000, 1000, 22, need_stale=False)


@pytest.mark.gpu
def test_gpu_extend_pack_fuzz(gpu):
    run_extend_pack(gpu, 6, 8000, seed=61)
    run_extend_pack(gpu, 7, 8000, seed=62)


@pytest.mark.gpu
def test_gpu_extend_ring_fuzz(gpu):
    run_extend(gpu, 1, 5000, 400, 23, need_stale=True)
    gpu.set_option("ext_blk", 0)
    try:
        run_extend(gpu, 1, 2000, 400, 28, need_stale=True)
    finally:
        gpu.set_option("ext_blk", -1)


@pytest.mark.gpu
def test_gpu_global_fuzz(gpu):
    run_global(gpu, 2, 5000, 320, 192, 24)
    run_global(gpu, 3, 5000, 600, 1 << 30, 25)
    run_global(gpu, 5, 2000, 2500, 1900, 27)


@pytest.mark.gpu
def test_gpu_align2_fuzz(gpu):
    run_align2(gpu, 5000, 26)


@pytest.mark.gpu
def test_gpu_ring_global_both_forms(gpu):
    """k_dedup_wave's score-only aligner on hardware in both forms -- four columns per lane (the product's default since round 4) and one column
    per lane (option dedup_blk = 0) -- against the reference's ksw_global2, with bands through every width at which a row takes another pass.
    (Round 3 ran the four-column form as a non-strict expected failure in a child process; it is the product now and gates the suite.)"""
    run_global(gpu, 3, 3000, 1000, 1 << 30, 44, wide=(63, 64, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385, 500))
    gpu.set_option("dedup_blk", 0)
    try:
        run_global(gpu, 3, 2000, 1000, 1 << 30, 46, wide=(63, 64, 127, 128, 129, 255, 256, 257, 383, 384, 385))
    finally:
        gpu.set_option("dedup_blk", -1)


@pytest.mark.gpu
def test_gpu_ring_extension_wide_bands(gpu):
    run_extend(gpu, 1, 2000, 400, 43, need_stale=True, very_wide=40)
