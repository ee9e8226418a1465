"""GPU (-m gpu): parity of the HIP path, called through the C-ABI of libbwagpu.so, against
  * the committed reference outputs (tests/golden/, generated from the compiled reference),
  * the plain-C oracle on larger seeded inputs (and the compiled reference itself where oracle/_ref is present),
  * size-independent properties at benchmark scale.
Bit-exact: all quantities on this path are integers (scores, coordinates) plus one float that is a pure function
of integers (frac_rep); records are compared byte for byte."""
import os
import numpy as np
import pytest

import testdata
from cmputil import assert_regs_equal, golden_opts, golden_sets
from bwa_amd import simdata
from bwa_amd.structs import ALNREG_DTYPE, default_opt, pacbio_opt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from bwa_amd.api import BwaGpu
    import orcapi
    prefix, g = testdata.small_index()
    gpu, orc = BwaGpu(prefix), orcapi.OrcIndex(prefix)
    yield gpu, orc, g
    gpu.close(); orc.close()


@pytest.fixture(scope="module")
def medium():
    """2 Mb repeat-rich genome; needs the reference's `bwa index` (oracle/_ref/bwa travels with the snapshot)."""
    import refapi
    # on the GPU box the compiled reference must have travelled with the snapshot: a silent skip here would turn the parity suite green
    # with a handful of tests (oracle/_ref is git-ignored, not gpurun-ignored)
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    from bwa_amd.api import BwaGpu
    import orcapi
    fa, g = testdata.medium_index()
    gpu, orc, ref = BwaGpu(fa), orcapi.OrcIndex(fa), refapi.RefIndex(fa)
    yield gpu, orc, ref, g
    gpu.close(); orc.close(); ref.close()


def test_native_library_is_loaded(small):
    gpu = small[0]
    assert b"gfx950" in gpu.L.bwagpu_version()
    maps = open("/proc/self/maps").read()
    assert "libbwagpu.so" in maps and "hostsim" not in maps


def test_golden_regs(small):
    gpu = small[0]
    opts = golden_opts()
    for name, oname, reads, counts, regs in golden_sets(os.path.join(testdata.GOLDEN, "golden_regs.npz")):
        seqs, off = testdata.flat(reads)
        c, r = gpu.align(opts[oname], seqs, off)
        assert_regs_equal(counts, regs.astype(ALNREG_DTYPE), c, r, f"golden {name}")


def test_golden_stage_taps(small):
    gpu = small[0]
    z = np.load(os.path.join(testdata.GOLDEN, "golden_stages.npz"))
    seqs, off = testdata.flat(z["reads"])
    gpu.align(golden_opts()["default"], seqs, off)
    n, iv = gpu.tap_intervals()
    assert np.array_equal(n, z["intv_n"])
    for f in ("x0", "x2", "info"):
        assert np.array_equal(iv[f], z["intv"][f])
    cn, ch, cs = gpu.tap_chains()
    assert np.array_equal(cn, z["chain_n"])
    for f, g in (("n_seeds", "n"), ("rid", "rid"), ("w", "w"), ("kept", "kept"), ("is_alt", "is_alt"), ("frac_rep", "frac_rep"), ("pos", "pos")):
        assert np.array_equal(ch[f], z["chain_hdr"][g]), f
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(cs[f], z["chain_seeds"][f]), f
    rn, rr = gpu.tap_regs_raw()
    assert np.array_equal(rn, z["raw_n"]) and rr.tobytes() == z["raw_regs"].astype(ALNREG_DTYPE).tobytes()


def test_edge_cases(small):
    gpu, orc, g = small
    opt = default_opt()
    c, r = gpu.align(opt, np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
    assert len(c) == 0 and len(r) == 0
    rng = np.random.default_rng(5)
    base = simdata.make_reads_se(g, 300, length=300, seed=21)
    rag = [r_[: int(rng.integers(1, 300))] for r_ in base] + [np.zeros(0, dtype=np.uint8), np.full(60, 4, dtype=np.uint8), base[0][:18]]
    rag.append(simdata.make_reads_long(g, 1, length=1200, seed=22, sub=0.01, dele=0.005, ins=0.005)[0])
    seqs, off = testdata.ragged(rag)
    assert_regs_equal(*orc.align(opt, seqs, off), *gpu.align(opt, seqs, off), "ragged")
    rep = simdata.make_reads_se(g, 64, seed=23)
    seqs, off = testdata.flat(np.concatenate([np.tile(rep[:1], (64, 1)), rep]))
    o2 = default_opt(); o2.max_occ = 2000
    assert_regs_equal(*orc.align(o2, seqs, off), *gpu.align(o2, seqs, off), "arena growth")


def _n_rich_reads(g, n, max_len, seed):
    """Ragged reads of max_len/3..max_len bases with Ns sprinkled in: single Ns, runs, Ns at either end and on the 16-base word
    boundaries of the seeding kernel's LDS copy, plus an all-N, an empty and a too-short read."""
    rng = np.random.default_rng(seed)
    base = simdata.make_reads_se(g, n, length=max_len, seed=seed, sub=0.01)
    out = []
    for i, r_ in enumerate(base):
        r_ = r_[: int(rng.integers(max_len // 3, max_len + 1))].copy()
        kind = i % 5
        if kind == 1: r_[int(rng.integers(0, len(r_)))] = 4
        elif kind == 2: a = int(rng.integers(0, len(r_) - 6)); r_[a:a + 5] = 4
        elif kind == 3: r_[0] = 4; r_[-1] = 4
        elif kind == 4 and len(r_) > 40: r_[16] = 4; r_[31] = 4; r_[32] = 4
        out.append(r_)
    return out + [np.zeros(0, dtype=np.uint8), np.full(40, 4, dtype=np.uint8), base[0][:max_len], base[1][:17]]


@pytest.mark.gpu
def test_short_read_batches_with_ns_and_seed_length_options(small):
    """Batches whose longest read is at most 256 bases take the seeding kernel with the reads in LDS (8, 12 or 16 words per lane); reads
    holding an N fetch their bases from global memory.  Same regions as the oracle for ragged, N-rich reads at three batch shapes, and
    for -k at, below and above the prefix tables' depth (which switches the bit-mask representation of short stack entries)."""
    gpu, orc, g = small
    for max_len in (120, 150, 250):
        seqs, off = testdata.ragged(_n_rich_reads(g, 2000, max_len, seed=300 + max_len))
        assert_regs_equal(*orc.align(default_opt(), seqs, off), *gpu.align(default_opt(), seqs, off), f"N-rich ragged reads up to {max_len} bp")
    seqs, off = testdata.flat(simdata.make_reads_se(g, 3000, seed=77, sub=0.04))
    for k in (9, 10, 11, 14):
        opt = default_opt(); opt.min_seed_len = k
        assert_regs_equal(*orc.align(opt, seqs, off), *gpu.align(opt, seqs, off), f"min_seed_len {k}")


@pytest.mark.parametrize("name,n,kw,oname", [
    ("se150", 30000, dict(seed=31, n_frac=0.002), "default"),
    ("se100_noisy", 8000, dict(length=100, seed=32, sub=0.04, dele=0.006, ins=0.006), "default"),
    ("se250_odd", 6000, dict(length=250, seed=33, sub=0.03, dele=0.005, ins=0.005), "odd"),
    ("se36", 8000, dict(length=36, seed=34), "default"),
])
def test_medium_short_reads_vs_oracle_and_reference(medium, name, n, kw, oname):
    gpu, orc, ref, g = medium
    seqs, off = testdata.flat(simdata.make_reads_se(g, n, **kw))
    opt = golden_opts()[oname]
    cg, rg = gpu.align(opt, seqs, off)
    assert_regs_equal(*orc.align(opt, seqs, off), cg, rg, name + " vs oracle")
    assert_regs_equal(*ref.align(opt, seqs, off), cg, rg, name + " vs compiled reference")


def test_medium_paired_and_long_reads(medium):
    gpu, orc, ref, g = medium
    r1, r2 = simdata.make_reads_pe(g, 8000, seed=35)
    inter = np.empty((16000, 150), dtype=np.uint8); inter[0::2] = r1; inter[1::2] = r2   # mates interleaved (bwamem.h:146-150)
    seqs, off = testdata.flat(inter)
    opt = default_opt(); opt.flag |= 0x2
    assert_regs_equal(*ref.align(opt, seqs, off), *gpu.align(opt, seqs, off), "PE mates")
    seqs, off = testdata.flat(simdata.make_reads_long(g, 48, length=4000, seed=36))
    assert_regs_equal(*ref.align(pacbio_opt(), seqs, off), *gpu.align(pacbio_opt(), seqs, off), "pacbio 4 kb")


def test_dense_sa_gives_identical_results(medium):
    from bwa_amd.api import BwaGpu
    gpu, orc, ref, g = medium
    fa, _ = testdata.medium_index()
    seqs, off = testdata.flat(simdata.make_reads_se(g, 5000, seed=37))
    base = gpu.align(default_opt(), seqs, off)
    g2 = BwaGpu(fa)
    g2.densify_sa(1)
    assert_regs_equal(*base, *g2.align(default_opt(), seqs, off), "sa_intv 1")
    g2.close()


def test_index_access_variants_give_identical_results(medium):
    """Seeding, SA look-ups, SA densification and the prefix tables read the 32-byte layout of the BWT by default; option occ32 = 0 keeps them
    on the reference-format 64-byte blocks; prefix tables off or shallow; the one-round-trip seeding kernel (seed_mrg = 2, the long-read
    default) on short reads: same regions and interval taps in every variant, with the SA at the reference's interval and densified."""
    from bwa_amd.api import BwaGpu
    gpu, orc, ref, g = medium
    fa, _ = testdata.medium_index()
    seqs, off = testdata.flat(simdata.make_reads_se(g, 8000, seed=39, sub=0.02))
    base = gpu.align(default_opt(), seqs, off)
    n0, iv0 = gpu.tap_intervals()
    assert_regs_equal(*ref.align(default_opt(), seqs, off), *base, "32-byte blocks vs compiled reference")
    for env in ({"occ32": 0}, {"occ32": 0, "ptab_m": 0}, {"ptab_m": 0}, {"ptab_m": 7}, {"seed_mrg": 0}, {"seed_mrg": 2}, {"seed_mrg": 2, "seed_lds_ent": 2}):
        g2 = BwaGpu(fa, options=env)
        assert_regs_equal(*base, *g2.align(default_opt(), seqs, off), f"{env}, sa_intv 32")
        n1, iv1 = g2.tap_intervals()
        assert np.array_equal(n0, n1) and iv0.tobytes() == iv1.tobytes(), env
        g2.densify_sa(2)
        assert_regs_equal(*base, *g2.align(default_opt(), seqs, off), f"{env}, sa_intv 2")
        g2.close()


def test_properties_at_scale(medium):
    """200 k reads: results are independent of batch composition (a permuted batch gives the permuted results) and
    equal the oracle on a sample; every region lies within one contig strand."""
    gpu, orc, ref, g = medium
    n = 200000
    reads = simdata.make_reads_se(g, n, seed=38)
    seqs, off = testdata.flat(reads)
    opt = default_opt()
    c, r = gpu.align(opt, seqs, off)
    starts = np.concatenate([[0], np.cumsum(c)])
    perm = np.random.default_rng(39).permutation(n)
    cp, rp = gpu.align(opt, *testdata.flat(reads[perm]))
    assert np.array_equal(cp, c[perm])
    sp = np.concatenate([[0], np.cumsum(cp)])
    for k in range(0, n, 997):
        i = perm[k]
        assert r[starts[i]:starts[i + 1]].tobytes() == rp[sp[k]:sp[k + 1]].tobytes()
    sub = np.arange(0, n, 41)
    co, ro = orc.align(opt, *testdata.flat(reads[sub]))
    assert np.array_equal(co, c[sub])
    so = np.concatenate([[0], np.cumsum(co)])
    for k, i in enumerate(sub):
        assert ro[so[k]:so[k + 1]].tobytes() == r[starts[i]:starts[i + 1]].tobytes()
    l_pac = ref.l_pac
    assert ((r["rb"] < r["re"]) & (r["qb"] < r["qe"]) & (r["re"] <= 2 * l_pac) & ~((r["rb"] < l_pac) & (r["re"] > l_pac))).all()


def test_device_cigars_match_the_reference_sam(medium):
    """bwagpu_batch_cigars (mem_reg2aln's banded global alignment + traceback on the device).  Fed to the host
    finalize code as hints, the SAM text must equal the compiled reference's mem_process_seqs output byte for byte, for
    single-end and paired-end input with indels on both strands; most regions must be served by the device."""
    import hostapi
    gpu, orc, ref, g = medium
    host = hostapi.HostFinalize(testdata.medium_index()[0])
    ascii_ = np.frombuffer(b"ACGTN", dtype=np.uint8)
    for pe, noisy in ((False, 1), (True, 1), (False, 0), (False, 2)):
        opt = default_opt()
        kw = (dict(), dict(sub=0.03, dele=0.004, ins=0.004), dict(sub=0.02, dele=0.03, ins=0.03))[noisy]   # 2: gap-rich, many regions with 7..64 operations
        if pe:
            opt.flag |= 2
            r1, r2 = simdata.make_reads_pe(g, 10000, seed=611, **kw)
            reads = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
        else:
            reads = simdata.make_reads_se(g, 20000, seed=610, **kw)
        seqs, off = testdata.flat(reads)
        n = off.shape[0] - 1
        names = [f"q{i >> 1}" if pe else f"q{i}" for i in range(n)]
        quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
        counts, regs = gpu.align(opt, seqs, off)
        cigs, ops = gpu.cigars(opt), gpu.cigar_ops()
        assert cigs.shape[0] == regs.shape[0]
        hc, hops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
        assert (cigs["score"] == hc["score"]).all() and (cigs["n_cigar"] == hc["n_cigar"]).all() and ops.shape[0] == hops.shape[0]
        assert (cigs["nm"] == hc["nm"]).all() and (cigs["md_len"] == hc["md_len"]).all()
        few = (cigs["n_cigar"] <= 6) & (cigs["md_len"] <= 8)
        assert cigs[few].tobytes() == hc[few].tobytes(), "device records differ from the host's"
        assert hostapi.decode_cigars(cigs[~few], ops) == hostapi.decode_cigars(hc[~few], hops), "device operation arrays (CIGARs, MD strings) differ from the host's"
        want = ref.process_seqs(opt, names, ascii_[seqs].tobytes(), quals, off)
        got = host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, cig_ops=ops)
        if got != want:
            for a, b in zip(want.split(b"\n"), got.split(b"\n")):
                assert a == b, f"first differing SAM line (pe={pe})\nwant {a[:300]!r}\ngot  {b[:300]!r}"
        assert got == want
        ok = regs["score"] >= opt.T
        served = (cigs["n_cigar"][ok] >= 0).mean()      # the rest has more than 64 CIGAR operations or exceeds the kernel's band limits, and stays on the host
        assert served > 0.95, f"device served too few regions: {served:.3f}"
        assert (cigs["n_cigar"] > 1).sum() > 100, "too few gapped alignments to exercise the traceback"
        if noisy == 2:
            assert (cigs["n_cigar"] > 6).sum() > 1000, "too few long alignments to exercise the operation array"
    host.close()


def test_device_matesw_records_match_the_host(medium):
    """bwagpu_batch_matesw on pairs whose second mate is too noisy to map on its own -- the device's task list and
    ksw_align2 results equal the host code's (same records, any order), and the SAM text produced with them is the reference's."""
    import ctypes as C
    import hostapi
    from bwa_amd.api import MATESW_DTYPE, PES_DTYPE
    gpu, orc, ref, g = medium
    host = hostapi.HostFinalize(testdata.medium_index()[0])
    r1, r2 = simdata.make_reads_pe(g, 10000, seed=98)
    rng = np.random.default_rng(99)
    r2 = np.where(rng.random(r2.shape) < 0.10, (r2 + rng.integers(1, 4, r2.shape)) % 4, r2).astype(np.uint8)
    reads = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
    seqs, off = testdata.flat(reads)
    opt = default_opt(); opt.flag |= 2
    counts, regs = gpu.align(opt, seqs, off)
    pes = host.pestat(opt, counts, regs)
    dpes = np.zeros(4, dtype=PES_DTYPE)
    for k in ("low", "high", "failed"):
        dpes[k] = pes[k]
    got = gpu.matesw(opt, dpes)
    want = host.matesw_records(opt, seqs, off, counts, regs, pes)
    key = lambda a: np.sort(np.frombuffer(a.tobytes(), dtype=f"V{MATESW_DTYPE.itemsize}"))
    assert got.shape == want.shape and (key(got) == key(want)).all(), "device mate-rescue records differ from the host's"
    assert (got["r"] >= 0).sum() > 1000
    names = [f"q{i >> 1}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    ascii_ = np.frombuffer(b"ACGTN", dtype=np.uint8)
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, msw=got, cigs=gpu.cigars(opt)) == ref.process_seqs(opt, names, ascii_[seqs].tobytes(), quals, off)
    host.close()


@pytest.mark.parametrize("n_processed", [(1 << 24) - 3000, (1 << 25) + (1 << 24) + 7000])
def test_pair_ids_beyond_2_to_23_with_device_records(medium, n_processed):
    """mem_pair / mem_matesw hash the pair's number n_processed + i >> 1 through an `id << 8` that wraps at 2^23 pairs (bwamem_pair.c:208,248):
    a batch deep inside a run -- n_processed >= 2^24 reads, crossing the wrap inside the batch in the first case -- with the DEVICE's regions, CIGAR records and
    mate-rescue alignments attached must still give the reference's mem_process_seqs text for that n_processed."""
    import hostapi
    from bwa_amd.api import PES_DTYPE
    gpu, orc, ref, g = medium
    host = hostapi.HostFinalize(testdata.medium_index()[0])
    r1, r2 = simdata.make_reads_pe(g, 6000, seed=355, sub=0.02)
    rng = np.random.default_rng(356)
    noisy = rng.random(r2.shape[0]) < 0.3                      # a third of the second mates too noisy to map on their own: mate rescue has work
    r2 = np.where(noisy[:, None] & (rng.random(r2.shape) < 0.10), (r2 + rng.integers(1, 4, r2.shape)) % 4, r2).astype(np.uint8)
    reads = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
    seqs, off = testdata.flat(reads)
    opt = default_opt(); opt.flag |= 2
    counts, regs = gpu.align(opt, seqs, off)
    pes = host.pestat(opt, counts, regs)
    dpes = np.zeros(4, dtype=PES_DTYPE)
    for k in ("low", "high", "failed"):
        dpes[k] = pes[k]
    msw = gpu.matesw(opt, dpes)
    cigs, ops = gpu.cigars(opt), gpu.cigar_ops()
    assert (msw["r"] >= 0).sum() > 300 and (cigs["n_cigar"] > 0).sum() > 5000
    names = [f"q{i >> 1}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    ascii_ = np.frombuffer(b"ACGTN", dtype=np.uint8)
    want = ref.process_seqs(opt, names, ascii_[seqs].tobytes(), quals, off, n_processed=n_processed)
    got = host.regs2sam(opt, names, seqs, quals, off, counts, regs, n_processed=n_processed, msw=msw, cigs=cigs, cig_ops=ops)
    if got != want:
        for a, b in zip(want.split(b"\n"), got.split(b"\n")):
            assert a == b, f"first differing SAM line (n_processed {n_processed})\nwant {a[:300]!r}\ngot  {b[:300]!r}"
    assert got == want
    assert want != ref.process_seqs(opt, names, ascii_[seqs].tobytes(), quals, off, n_processed=0), "the pair ids made no difference: the test does not reach the hash"
    host.close()


def test_index_broadcast_over_rccl_single_rank(small):
    """The multi-GPU start-up path on one GPU: a world_size-1 RCCL group, the index buffers wrapped as device tensors
    (CUDA array interface) and broadcast; the handle must align exactly like one created directly."""
    import torch
    import torch.distributed as dist
    from bwa_amd import dist as bdist
    gpu, orc, g = small
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29631")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        prefix, _ = testdata.small_index()
        g2 = bdist.broadcast_index(prefix, device=0, src=0)
        ptr, nbytes = g2.index_buffers()[0]
        t = bdist._as_tensor(ptr, nbytes, True)
        assert t.is_cuda and t.numel() == nbytes
        ref_bytes = np.fromfile(prefix + ".bwt", dtype=np.uint8)[40:]
        assert np.array_equal(t[: len(ref_bytes)].cpu().numpy(), ref_bytes)       # the device buffer really is the .bwt payload
        e = bdist.BwaGpu.empty(g2.index_meta())                                    # receiving side: allocate, then fill by copy
        for (dp, dn), (sp, sn) in zip(e.index_buffers(), g2.index_buffers()):
            bdist._as_tensor(dp, dn, True).copy_(bdist._as_tensor(sp, sn, True))
        torch.cuda.synchronize()
        e.index_ready()
        seqs, off = testdata.flat(simdata.make_reads_se(g, 3000, seed=81))
        want = gpu.align(default_opt(), seqs, off)
        assert_regs_equal(*want, *g2.align(default_opt(), seqs, off), "broadcast handle")
        assert_regs_equal(*want, *e.align(default_opt(), seqs, off), "received handle")
        g2.close(); e.close()
    finally:
        dist.destroy_process_group()


def test_alt_contigs_on_the_device(medium, tmp_path):
    """An index with a .alt file (bntseq.c:185-205): the ALT flag reaches mem_chain (chain records), mem_chain_flt's overlap rule
    (bwamem.c:377), mem_sort_dedup_patch and the regions' is_alt (bwamem.c:1113-1114) on the device.  Regions equal the compiled
    reference's (which loads the same .alt), and the SAM of `bwa-amd mem` equals `bwa mem`'s for single- and paired-end reads."""
    import subprocess
    import refapi
    from bwa_amd.api import BwaGpu
    from bwa_amd import build as b
    _, _, _, g = medium
    fa = testdata.medium_index()[0]
    new = str(tmp_path / "alt_idx")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        os.symlink(os.path.abspath(fa + "." + ext), new + "." + ext)
    with open(new + ".alt", "w") as f:
        f.write("chr3\t0\tchr1\t1\t60\t100M\t*\t0\t0\t*\t*\nchr2\n")
    gpu, ref = BwaGpu(new), refapi.RefIndex(new)
    lens = simdata.make_genome(**testdata.MEDIUM)[1]
    lo = lens[0]
    reads = np.concatenate([simdata.make_reads_se(g[lo:], 6000, seed=61), simdata.make_reads_se(g, 6000, seed=62)])
    seqs, off = testdata.flat(reads)
    opt = default_opt()
    cg, rg = gpu.align(opt, seqs, off)
    assert_regs_equal(*ref.align(opt, seqs, off), cg, rg, "ALT contigs vs compiled reference")
    assert int((rg["ncomp_isalt"] >> 30).sum()) > 1000
    gpu.close(); ref.close()
    _, cli = b.build_host(verbose=False)
    f1, f2 = str(tmp_path / "a1.fq"), str(tmp_path / "a2.fq")
    r1, r2 = simdata.make_reads_pe(g, 5000, seed=63)
    simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
    for files in ([f1], [f1, f2]):
        outs = []
        for binary in (refapi.REF_BWA, cli):
            p = subprocess.run([binary, "mem", "-K", "100000000", "-t", "4", new] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-1000:]
            outs.append(b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")))
        assert outs[0] == outs[1], f"SAM with ALT contigs, {len(files)} file(s)"
        assert b"AH:*" in outs[0] and b"\tpa:f:" in outs[0]


def test_pacbio_10kb_reads(medium):
    """BASELINE configs[4] shape: 200 reads of 10 kb, PacBio-CLR-like errors, -x pacbio: long-read seeding, mem_flt_chained_seeds
    (wave-per-read re-scoring), ring-mode extension incl. its wide-band fallback, wave-per-read patching -- regions equal the
    compiled reference's."""
    gpu, orc, ref, g = medium
    reads = simdata.make_reads_long(g, 200, length=10000, seed=64)
    seqs, off = testdata.flat(reads)
    cg, rg = gpu.align(pacbio_opt(), seqs, off)
    assert_regs_equal(*ref.align(pacbio_opt(), seqs, off), cg, rg, "pacbio 10 kb x 200")
    o2 = pacbio_opt(); o2.w = 700                       # bands wider than the LDS ring: the lane-per-read extension kernel (k_extend)
    sub = reads[:24]
    seqs, off = testdata.flat(sub)
    assert_regs_equal(*ref.align(o2, seqs, off), *gpu.align(o2, seqs, off), "pacbio 10 kb, w = 700 (k_extend fallback)")


def _ref_align_threads(ref, opt, reads, n_threads=16):
    """ref.align over slices of `reads` on host threads (the compiled reference's call is per read and ctypes drops the GIL)."""
    import threading
    n = reads.shape[0]
    cuts = np.linspace(0, n, min(n_threads, n) + 1).astype(int)
    parts = [None] * (len(cuts) - 1)

    def work(i):
        parts[i] = ref.align(opt, *testdata.flat(reads[cuts[i]:cuts[i + 1]]))
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
    [t.start() for t in th]; [t.join() for t in th]
    return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])


def test_pacbio_10kb_reads_at_scale_under_the_long_read_defaults(medium):
    """600 reads of 10 kb, -x pacbio, in two batches of 300 on one handle (arenas, chunk-task tables and scratch re-used), under the kernel
    forms that became the long-read defaults in round 4 -- pass 1 of the seeding by chunks of 256 bases with one memory round trip per
    iteration, workgroup-per-read interval sort, seed re-scoring in LDS, four columns per lane in the patch alignments: regions equal the
    compiled reference's mem_align1_core, read by read.  A third batch runs the forms they replaced on the same handle (options set
    through the API between batches): the same regions again."""
    gpu, orc, ref, g = medium
    for k in ("seed_mrg", "seed_tasks", "publish_blk", "seedsw_lds", "dedup_blk"):
        assert gpu.get_option(k) == -1, f"{k} is not on its automatic setting"
    reads = simdata.make_reads_long(g, 600, length=10000, seed=66)
    want = _ref_align_threads(ref, pacbio_opt(), reads)
    got = [gpu.align(pacbio_opt(), *testdata.flat(reads[a:a + 300])) for a in (0, 300)]
    assert_regs_equal(*want, np.concatenate([x[0] for x in got]), np.concatenate([x[1] for x in got]), "pacbio 10 kb x 600, long-read defaults, two batches")
    assert int(want[0].sum()) > 600
    old = {"seed_mrg": 0, "seed_tasks": 0, "publish_blk": 0, "seedsw_lds": 0, "dedup_blk": 0}
    try:
        for k, v in old.items():
            gpu.set_option(k, v)
        c3, r3 = gpu.align(pacbio_opt(), *testdata.flat(reads[:120]))
    finally:
        for k in old:
            gpu.set_option(k, -1)
    n120 = int(want[0][:120].sum())
    assert_regs_equal(want[0][:120], want[1][:n120], c3, r3, "pacbio 10 kb x 120, round-3 kernel forms")


def _heavy():
    """tests/heavycase.py's genome, index and hard reads, built once per session."""
    import heavycase
    global _HEAVY
    try:
        _HEAVY
    except NameError:
        _HEAVY = heavycase.build()
    return _HEAVY


@pytest.mark.parametrize("regs,flt_lds", [(2, 256), (1, 256), (0, 256), (2, 0), (2, 40)])
def test_gpu_wave_chaining_heavy_reads(regs, flt_lds):
    """The device twin of tests/test_hostsim.py::test_hostsim_wave_chaining_heavy_reads (round 5's chaining forms had their directed test on
    the mock runtime only -- and DPP lane shifts are what a fibre emulation can get differently from silicon): reads with hundreds of chains, a few
    dozen, duplicate chain positions in one-node and larger trees, through every (chain_regs, chain_flt_lds) form of k_chain_wave; the histogram of
    forms used is asserted (register form / tree form / four-array form), chains and seeds equal the COMPILED REFERENCE's mem_chain + mem_chain_flt
    (refshim_chains), regions its mem_align1_core."""
    import heavycase
    import refapi
    from bwa_amd.api import BwaGpu
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    fa, orc, reads = _heavy()
    ref = refapi.RefIndex(fa)
    gpu = BwaGpu(fa, options={"chain_regs": regs, "chain_flt_lds": flt_lds})
    heavycase.check(gpu, ref, reads, regs)
    # and the packed-extension switch on the same hard reads (reads with more chains than a wave's task list holds)
    gpu.set_option("ext_pack", 1)
    seqs, off = testdata.flat(reads)
    assert_regs_equal(*ref.align(default_opt(), seqs, off), *gpu.align(default_opt(), seqs, off), "heavy reads, ext_pack = 1")
    gpu.close(); ref.close()


def test_ext_pack_gives_identical_results(medium):
    """Option ext_pack (k_ext_pack, dev_extp.h: the chains' first extensions four to a wavefront, replayed by k_extend_wave; off by default since it
    measured slower): same regions as the compiled reference at both occupancies, single-end, paired-end flag, odd options (band 20: the routine declines
    most tasks), 250 bp reads (queries beyond its 127 columns)."""
    from bwa_amd.api import BwaGpu
    import ctypes as C
    gpu, orc, ref, g = medium
    fa, _ = testdata.medium_index()
    odd = golden_opts()["odd"]
    sets = [("se150", simdata.make_reads_se(g, 8000, seed=71), default_opt()), ("se150 odd options", simdata.make_reads_se(g, 3000, seed=72, sub=0.03), odd),
            ("se250 noisy", simdata.make_reads_se(g, 2000, length=250, seed=73, sub=0.03, dele=0.005, ins=0.005), default_opt())]
    for ep in (1, 5):
        g2 = BwaGpu(fa, options={"ext_pack": ep})
        g2.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
        for name, reads, opt in sets:
            seqs, off = testdata.flat(reads)
            assert_regs_equal(*ref.align(opt, seqs, off), *g2.align(opt, seqs, off), f"ext_pack = {ep}, {name}")
            if name == "se150":
                prof = (C.c_ulonglong * 16)()
                g2.L.bwagpu_debug_prof(g2.h, prof)
                assert prof[8] > reads.shape[0], f"k_ext_pack answered only {prof[8]} extensions of {reads.shape[0]} reads"
        g2.close()


@pytest.mark.parametrize("heavy_min,stage,big,net", [(0, -1, -1, -1), (2, 16, -1, 12), (2, 16, 32, 0), (8, 0, -1, -1), (-1, 512, 0, 4)])
def test_dedup_list_gives_identical_results(medium, heavy_min, stage, big, net):
    """k_dedup lists the reads with several regions for k_dedup_wave<.., LIST> (dedup_read_par, dev_dedupp.h: operands in LDS, the lanes over the
    regions), options dedup_heavy / dedup_stage / dedup_big (3 regions, 128, what 64 KB hold by default, which the other tests run): with no list; with
    nearly every read listed, small arrays in the first launch and the rest in the second; the same with reads beyond the second launch's arrays (done in
    place); a middling bound with every listed read in place; one launch with large arrays.  The regions equal the compiled reference's -- ordinary reads,
    odd options, 250 bp reads, and the repeat-rich genome's reads with hundreds of regions."""
    from bwa_amd.api import BwaGpu
    gpu, orc, ref, g = medium
    fa, _ = testdata.medium_index()
    odd = golden_opts()["odd"]
    sets = [("se150", simdata.make_reads_se(g, 8000, seed=81), default_opt()), ("se150 odd options", simdata.make_reads_se(g, 3000, seed=82, sub=0.03), odd),
            ("se250 noisy", simdata.make_reads_se(g, 2000, length=250, seed=83, sub=0.03, dele=0.005, ins=0.005), default_opt())]
    g2 = BwaGpu(fa, options={"dedup_heavy": heavy_min, "dedup_stage": stage, "dedup_big": big, "dedup_net": net})
    for name, reads, opt in sets:
        seqs, off = testdata.flat(reads)
        assert_regs_equal(*ref.align(opt, seqs, off), *g2.align(opt, seqs, off), f"dedup_heavy = {heavy_min}, dedup_stage = {stage}, dedup_big = {big}, dedup_net = {net}, {name}")
    g2.close()
    import refapi
    hfa, horc, hreads = _heavy()
    hg = simdata.make_genome(500_000, n_contigs=2, seed=5, n_interspersed=2000, divergence=0.04)[0]      # (heavycase.build's genome)
    rep_reads = simdata.make_reads_se(hg, 1500, seed=84, sub=0.05)
    g3, href = BwaGpu(hfa, options={"dedup_heavy": heavy_min, "dedup_stage": stage, "dedup_big": big, "dedup_net": net}), refapi.RefIndex(hfa)
    for name, reads in (("hard reads", hreads), ("repeat-rich genome", rep_reads)):
        seqs, off = testdata.flat(reads)
        assert_regs_equal(*href.align(default_opt(), seqs, off), *g3.align(default_opt(), seqs, off), f"dedup_heavy = {heavy_min}, {name}")
    g3.close(); href.close()


@pytest.mark.parametrize("heavy_min", [-1, 2, 0])
def test_dedup_patch_joins(medium, heavy_min):
    """The device twin of tests/test_hostsim.py::test_hostsim_dedup_patch_joins, 4000 reads against the compiled reference: regions that mem_patch_reg
    joins (several per read, in sequence), through k_dedup alone and through the listed reads' dedup_read_par."""
    import heavycase
    from bwa_amd.api import BwaGpu
    gpu, orc, ref, g = medium
    fa, _ = testdata.medium_index()
    reads, opt = heavycase.patch_reads(g, 4000, seed=9), heavycase.patch_opt()
    seqs, off = testdata.flat(reads)
    want = ref.align(opt, seqs, off)
    assert int(((want[1]["ncomp_isalt"] & 0x3fffffff) > 1).sum()) >= 1000, "no joined regions in the test's reads"
    g2 = BwaGpu(fa, options={"dedup_heavy": heavy_min})
    g2.set_stats(True)
    assert_regs_equal(*want, *g2.align(opt, seqs, off), f"patch joins, dedup_heavy = {heavy_min}")
    assert g2.stats()["n_glb_calls"] >= 10000
    g2.close()


def test_many_contigs_with_alt_at_scale(tmp_path):
    """An hs38DH-shaped contig table: 24 primary contigs and 3000 ALT contigs of 1 kb that are diverged copies of primary sequence, listed in a
    .alt file -- bns_pos2rid's bisection over thousands of contigs (bntseq.c:354-368), seeds and chains that bridge short contigs (rid -1), the ALT rules of
    mem_chain_flt (bwamem.c:377), mem_sort_dedup_patch and mem_mark_primary_se, XA / pa tags.  Regions equal the compiled reference's, `bwa-amd mem` SAM
    equals `bwa mem`'s (single-end, paired-end in several batches)."""
    import subprocess
    import refapi
    from bwa_amd.api import BwaGpu
    from bwa_amd import build as b
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    rng = np.random.default_rng(77)
    gp, lens_p = simdata.make_genome(3_000_000, n_contigs=24, seed=78)
    n_alt, alt_len = 3000, 1000
    starts = rng.integers(0, gp.shape[0] - alt_len, size=n_alt)
    alts = np.stack([gp[s:s + alt_len] for s in starts]).copy()
    mut = rng.random(alts.shape) < 0.02
    alts[mut] = (alts[mut] + rng.integers(1, 4, size=int(mut.sum()))) % 4
    g = np.concatenate([gp, alts.reshape(-1)]).astype(np.uint8)
    lens = [int(x) for x in lens_p] + [alt_len] * n_alt
    fa = str(tmp_path / "alt3k.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    with open(fa + ".alt", "w") as f:
        for k in range(n_alt):
            f.write(f"chr{len(lens_p) + 1 + k}\t0\tchr1\t1\t60\t{alt_len}M\t*\t0\t0\t*\t*\n")
    gpu, ref = BwaGpu(fa), refapi.RefIndex(fa)
    # reads from the primary sequence under the ALT copies, from the ALT contigs themselves (many cross a 1 kb contig's end), and from everywhere
    under = np.concatenate([gp[s:s + alt_len] for s in starts[:400]])
    reads = np.concatenate([simdata.make_reads_se(under, 4000, seed=81), simdata.make_reads_se(g[gp.shape[0]:], 4000, seed=82), simdata.make_reads_se(g, 4000, seed=83)])
    seqs, off = testdata.flat(reads)
    opt = default_opt()
    cg, rg = gpu.align(opt, seqs, off)
    assert_regs_equal(*ref.align(opt, seqs, off), cg, rg, "3000 ALT contigs vs compiled reference")
    assert int((rg["ncomp_isalt"] >> 30).sum()) > 3000 and int((rg["rid"] >= len(lens_p)).sum()) > 3000
    gpu.close(); ref.close()
    _, cli = b.build_host(verbose=False)
    f1, f2 = str(tmp_path / "a1.fq"), str(tmp_path / "a2.fq")
    r1, r2 = simdata.make_reads_pe(under, 6000, seed=84)
    simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
    for files, K in (([f1], "100000000"), ([f1, f2], "400000")):
        outs = []
        for binary in (refapi.REF_BWA, cli):
            p = subprocess.run([binary, "mem", "-K", K, "-t", "4", fa] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, p.stderr.decode()[-1000:]
            outs.append(b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")))
        assert outs[0] == outs[1], f"SAM with 3000 ALT contigs, {len(files)} file(s)"
        assert outs[0].count(b"AH:*") == n_alt and b"\tpa:f:" in outs[0] and b"\tXA:Z:" in outs[0]
