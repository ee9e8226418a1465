"""The chaining stage's hard reads, shared by the mock-runtime test (tests/test_hostsim.py) and its twin on the device (tests/test_gpu_parity.py):
a repeat-rich 500 kb genome (500 copies per repeat family), its index, and the reads with the most chains, a few middling ones, an ordinary one and
reads that put two chains on one reference position; and the checker: histogram of the forms k_chain_wave used (bwagpu_debug_chain_hist), chains and
seeds against a checker index (the oracle, or the compiled reference: both have .chains and .align), regions."""
import ctypes as C
import os
import tempfile

import numpy as np

import orcapi
import refapi
import testdata
from bwa_amd import simdata
from bwa_amd.structs import default_opt
from cmputil import assert_regs_equal


def build():
    import pytest
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the genome)")
    d = tempfile.mkdtemp()
    g, lens = simdata.make_genome(500_000, n_contigs=2, seed=5, n_interspersed=2000, divergence=0.04)
    fa = os.path.join(d, "rep.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    orc = orcapi.OrcIndex(fa)
    opt = default_opt()
    cand = simdata.make_reads_se(g, 1200, seed=97, sub=0.05)        # noisy reads: the home copy of a repeat is not much better than the others
    n_chains = np.array([orc.chains(opt, r, 0)[0].shape[0] for r in cand])
    top = np.argsort(n_chains)[::-1][:30]
    n_kept = np.array([orc.chains(opt, cand[i], 1)[0].shape[0] for i in top])
    pick = list(top[:2]) + [int(top[i]) for i in np.argsort(n_kept)[::-1][:2]]
    assert n_chains[pick[0]] > 200 and n_kept.max() > 100, (n_chains[top], n_kept)
    mid = [int(i) for i in np.nonzero((n_chains >= 36) & (n_chains <= 60))[0][:2]]    # a few dozen chains: B-trees of two levels
    mid += [int(i) for i in np.nonzero((n_chains >= 70) & (n_chains <= 250))[0][:2]]  # more than one lane-array of chains, fewer than four
    assert len(mid) == 4, np.sort(n_chains)[::-1][:40]
    # Reads that put two chains on the same reference position (duplicate keys in the chain tree): a stretch of the genome, more than the band width of
    # other sequence, and the same stretch again -- the second copy's seeds start where the first copy's chains do and cannot join them (bwamem.c:229).
    rng = np.random.default_rng(11)
    L = cand.shape[1]
    twice, few, many = [], False, False
    for p0 in rng.integers(0, g.shape[0] - 200, size=60):
        rd = np.concatenate([g[p0:p0 + 40], rng.integers(0, 4, size=L - 80).astype(np.uint8), g[p0:p0 + 40]])
        if (rd > 3).any():
            continue
        hdr = orc.chains(opt, rd, 0)[0]
        dup = hdr.shape[0] > np.unique(hdr["pos"]).shape[0]
        if dup and hdr.shape[0] <= 9 and not few:
            few = True; twice.append(rd)
        elif dup and 9 < hdr.shape[0] <= 64 and not many:
            many = True; twice.append(rd)
    assert few and many, "no read with two chains on one position found (in a one-node tree, in a larger one)"
    reads = np.concatenate([cand[pick[:1]], cand[pick[2:3]], cand[mid], cand[:1], np.stack(twice)])   # most chains, most kept chains, four middling ones, an ordinary read, duplicate keys
    return fa, orc, reads


def check(dev, chk, reads, regs):
    """dev: a BwaGpu handle created with chain_regs = regs (mock runtime or device); chk: the index whose chains and regions are the truth."""
    opt = default_opt()
    dev.set_taps(True); dev.set_stats(True)
    seqs, off = testdata.flat(reads)
    c, r = dev.align(opt, seqs, off)
    hist = (C.c_ulonglong * 192)()
    dev.L.bwagpu_debug_chain_hist.argtypes = [C.c_void_p, C.c_void_p]
    assert dev.L.bwagpu_debug_chain_hist(dev.h, hist) == 0
    in_regs, in_tree = sum(hist[0:32]), sum(hist[64:96])
    assert in_regs + in_tree == reads.shape[0]
    assert (in_regs >= 4 and in_tree >= 2) if regs else in_regs == 0, (in_regs, in_tree)
    wide = sum(hist[4:32])           # reads that finished in the register form with 64 chains or more: the four-array form
    assert wide >= 2 if regs == 2 else wide == 0, list(hist[0:32])
    cn, ch, cs = dev.tap_chains()
    kc = ks = 0
    for i, rd in enumerate(reads):
        hdr, seeds = chk.chains(opt, rd, 1)
        assert cn[i] == hdr.shape[0], i
        got_h, got_s = ch[kc:kc + cn[i]], cs[ks:ks + int(hdr["n"].sum())]
        for f, gname in (("n_seeds", "n"), ("rid", "rid"), ("w", "w"), ("kept", "kept"), ("is_alt", "is_alt"), ("frac_rep", "frac_rep"), ("pos", "pos")):
            assert np.array_equal(got_h[f], hdr[gname]), (i, f)
        for f in ("rbeg", "qbeg", "len"):
            assert np.array_equal(got_s[f], seeds[f]), (i, f)
        kc += cn[i]; ks += int(hdr["n"].sum())
    assert_regs_equal(*chk.align(opt, seqs, off), c, r, "heavy chaining")
    dev.set_stats(False)


def patch_reads(g, n, seed=5):
    """Reads whose alignment breaks into several regions on one diagonal, which mem_patch_reg (bwamem.c:432-461) then joins again: 250 bp with two to
    four clusters of three adjacent mismatches, to be aligned with zdrop = 8 (patch_opt) so that an extension stops at a cluster.  Several joins per
    read, in sequence (a joined region goes on to meet the next one), and joins that fail the 0.9 score test."""
    rng = np.random.default_rng(seed + 2)
    reads = simdata.make_reads_se(g, n, length=250, seed=seed)
    for rd in reads:
        for pos in rng.choice(np.arange(40, 210, 45), size=int(rng.integers(2, 5)), replace=False):
            for k in range(3):
                rd[pos + k] = (rd[pos + k] + 1 + rng.integers(0, 3)) % 4
    return reads


def patch_opt():
    opt = default_opt()
    opt.zdrop = 8
    return opt
