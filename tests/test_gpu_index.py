"""GPU (-m gpu): the device index builder (bwagpu_index_build) against the reference's `bwa index`, and the whole hot path on
an index whose seq_len exceeds 2^32 (GRCh38 has seq_len ~ 6.2e9; bwtint_t is 64-bit, bwt.h:46): packed 37-bit interval records,
prefix tables, SA values, B-tree keys and region coordinates all carry high bits there.

The > 2^32 index cannot come from `bwa index` inside a test (about an hour of CPU); it is built by our builder -- proved
byte-identical to `bwa index` at 64 Mbp right here and on small genomes in test_index_build.py -- and then checked for internal
consistency with the *reference's own* bwt_sa/bwt_occ (LF walks over our BWT must land on suffixes in lexicographic order)
before the reference aligner and the GPU are compared on it."""
import ctypes as C
import filecmp
import os
import shutil

import numpy as np
import pytest

import refapi
import testdata
from cmputil import assert_regs_equal
from bwa_amd import simdata
from bwa_amd.index import build_index
from bwa_amd.structs import default_opt

pytestmark = pytest.mark.gpu

TMP = os.environ.get("BWA_AMD_TEST_TMP", "/tmp/bwa_amd_test_index")


def need_ref():
    # on a GPU box a missing oracle/_ref is an error, not a reason to skip: the parity claims rest on it
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) did not travel with the snapshot"


def test_gpu_built_index_equals_bwa_index_64mbp():
    need_ref()
    os.makedirs(TMP, exist_ok=True)
    g, lens = simdata.make_genome_large(64_000_000, n_contigs=5, seed=64)
    g[1_000_000:1_003_000] = np.tile(g[1_000_000:1_000_060], 50)      # a long exact tandem repeat: many doubling rounds
    fa = os.path.join(TMP, "g64m.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)                                             # the reference's bwtsw route (l_pac > 50 Mbp, bwtindex.c:275)
    prefix = os.path.join(TMP, "g64m_gpu")
    info = build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)])
    print(f"device build of 64 Mbp: {info['build_ms']:.0f} ms")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, fa + "." + ext, shallow=False), ext
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        os.remove(prefix + "." + ext); os.remove(fa + "." + ext)
    os.remove(fa)


L_BIG = 2_200_000_000          # seq_len = 4.4e9 > 2^32


@pytest.fixture(scope="module")
def big():
    need_ref()
    from bwa_amd.api import BwaGpu
    os.makedirs(TMP, exist_ok=True)
    g, lens = simdata.make_genome_large(L_BIG, n_contigs=24, seed=2200)
    prefix = os.path.join(TMP, "g2200m")
    info = build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)])
    print(f"device build of {L_BIG / 1e6:.0f} Mbp: {info['build_ms']:.0f} ms")
    assert info["seq_len"] == 2 * L_BIG and info["seq_len"] > (1 << 32)
    ref = refapi.RefIndex(prefix)
    gpu = BwaGpu(prefix)
    yield gpu, ref, g, info
    gpu.close(); ref.close()
    shutil.rmtree(TMP, ignore_errors=True)


def text_window(g, i, n_bases):
    """T[i : i + n_bases] of T = forward + reverse complement (values 0..3); shorter at the end of the text."""
    l = g.shape[0]
    idx = np.arange(i, min(i + n_bases, 2 * l), dtype=np.int64)
    fwd = idx < l
    out = np.empty(idx.shape[0], dtype=np.uint8)
    out[fwd] = g[idx[fwd]]
    out[~fwd] = 3 - g[2 * l - 1 - idx[~fwd]]
    return out


def test_big_index_is_consistent_under_the_references_lf_walk(big):
    """bwt_sa (bwt.c:86-96) of the reference, run over our .bwt/.sa: adjacent rows must hold lexicographically adjacent suffixes."""
    gpu, ref, g, info = big
    L = refapi.lib()
    L.bwt_sa.restype = C.c_uint64
    L.bwt_sa.argtypes = [C.c_void_p, C.c_uint64]
    bwt = L.refshim_idx_bwt(ref.h)
    n = info["seq_len"]
    rng = np.random.default_rng(3)
    rows = np.concatenate([rng.integers(1, n, size=1500), rng.integers(1 << 32, n, size=1500), [1, n - 1, info["primary"] - 1, info["primary"]]])
    n_hi = 0
    for k in rows.tolist():
        if k + 1 > n:
            continue
        a, b = L.bwt_sa(bwt, k), L.bwt_sa(bwt, k + 1)
        assert a < n and b < n and a != b
        n_hi += a >= (1 << 32) or b >= (1 << 32)
        wa, wb = text_window(g, a, 400), text_window(g, b, 400)
        m = min(len(wa), len(wb))
        d = np.nonzero(wa[:m] != wb[:m])[0]
        if d.size:
            assert wa[d[0]] < wb[d[0]], f"rows {k},{k + 1}: suffixes {a},{b} out of order"
        else:
            assert len(wa) < len(wb) or m == 400, f"rows {k},{k + 1}: suffix {a} is not a proper prefix of {b}"
    assert n_hi > 100, "the sample never produced suffix positions beyond 2^32"
    assert L.bwt_sa(bwt, info["primary"]) == 0


def test_big_index_hot_path_equals_compiled_reference(big):
    gpu, ref, g, info = big
    opt = default_opt()
    # reads from everywhere, plus reads from the first 60 Mbp: their reverse-strand hits have coordinates 2*l_pac-1-p > 2^32
    r_all = simdata.make_reads_se(g, 3000, seed=51, n_frac=0.001)
    r_low = simdata.make_reads_se(g[:60_000_000], 3000, seed=52)
    r_noisy = simdata.make_reads_se(g, 1000, length=250, seed=53, sub=0.03, dele=0.005, ins=0.005)
    for name, reads in (("all", r_all), ("first 60 Mbp", r_low), ("250 bp noisy", r_noisy)):
        seqs, off = testdata.flat(reads)
        gpu.set_taps(True)
        cg, rg = gpu.align(opt, seqs, off)
        cr, rr = ref.align(opt, seqs, off)
        assert_regs_equal(cr, rr, cg, rg, f"seq_len > 2^32, {name}")
        if name == "first 60 Mbp":
            assert int((rg["rb"] >= (1 << 32)).sum()) > 500, "no region beyond 2^32"
        if name == "all":    # stage taps: SA intervals with rows beyond 2^32 equal the reference's (packed 37-bit records, prefix tables)
            n_iv, iv = gpu.tap_intervals()
            assert int((iv["x0"] >= (1 << 32)).sum()) > 100
            k = 0
            for i in range(200):
                want = ref.intervals(opt, reads[i])
                got = iv[k:k + n_iv[i]]
                k += n_iv[i]
                assert n_iv[i] == want.shape[0] and np.array_equal(got["x0"], want["x0"]) and np.array_equal(got["x2"], want["x2"]) and np.array_equal(got["info"], want["info"]), i
    # paired-end flag set (mates interleaved) and the dense SA: same regions
    r1, r2 = simdata.make_reads_pe(g, 1500, seed=54)
    inter = np.empty((3000, 150), dtype=np.uint8); inter[0::2] = r1; inter[1::2] = r2
    seqs, off = testdata.flat(inter)
    o2 = default_opt(); o2.flag |= 0x2
    want = ref.align(o2, seqs, off)
    assert_regs_equal(*want, *gpu.align(o2, seqs, off), "seq_len > 2^32, PE mates")
    gpu.densify_sa(4)
    assert_regs_equal(*want, *gpu.align(o2, seqs, off), "seq_len > 2^32, PE mates, SA densified to 4")


def test_big_index_cli_sam_equals_bwa_mem(big, tmp_path):
    """`bwa-amd mem` against `bwa mem` on the seq_len > 2^32 index (GRCh38-scale SAM parity lived in bench.py only): 60 000 pairs in eight
    batches over the command line's three device handles -- mem_pestat per batch, device CIGARs and mate rescue with reference coordinates beyond 2^32
    (reverse-strand hits of reads from the first contigs), the dense suffix array sized against free HBM."""
    import subprocess
    from bwa_amd import build as b
    need_ref()
    gpu, ref, g, info = big
    prefix = os.path.join(TMP, "g2200m")
    _, cli = b.build_host(verbose=False)
    r1, r2 = simdata.make_reads_pe(g[:400_000_000], 60000, seed=56)
    f1, f2 = str(tmp_path / "b1.fq"), str(tmp_path / "b2.fq")
    simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
    outs = []
    for binary in (refapi.REF_BWA, cli):
        p = subprocess.run([binary, "mem", "-K", "2400000", "-t", "16", "-v", "3", prefix, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr.decode()[-1000:]
        if binary == cli:
            assert p.stderr.count(b"[M::process] read ") >= 6, p.stderr.decode()[-600:]
        outs.append(b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")))
    assert outs[0] == outs[1], "SAM on the seq_len > 2^32 index"
    assert outs[0].count(b"\n") >= 120000
