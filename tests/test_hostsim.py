"""CPU: the product source (host orchestration + lane-serial device routines) run under the mock HIP runtime
(tests/hostsim) and compared with the committed reference outputs and with the oracle.  This is a logic check of
the HIP code path without a GPU; the real-device parity tests are in test_gpu_parity.py (-m gpu)."""
import os
import numpy as np
import pytest

import hostsim_build
import orcapi
import testdata
from cmputil import assert_regs_equal, golden_opts, golden_sets
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import ALNREG_DTYPE, default_opt


@pytest.fixture(scope="module")
def sim():
    prefix, _ = testdata.small_index()
    s = BwaGpu(prefix, lib_path=hostsim_build.build())
    yield s
    s.close()


def test_hostsim_regs_match_golden(sim):
    opts = golden_opts()
    for name, oname, reads, counts, regs in golden_sets(os.path.join(testdata.GOLDEN, "golden_regs.npz")):
        seqs, off = testdata.flat(reads)
        c, r = sim.align(opts[oname], seqs, off)
        assert_regs_equal(counts, regs.astype(ALNREG_DTYPE), c, r, f"golden {name}")


def test_hostsim_stage_taps_match_golden(sim):
    z = np.load(os.path.join(testdata.GOLDEN, "golden_stages.npz"))
    seqs, off = testdata.flat(z["reads"])
    sim.align(golden_opts()["default"], seqs, off)
    n, iv = sim.tap_intervals()
    assert np.array_equal(n, z["intv_n"])
    for f, g in (("x0", "x0"), ("x2", "x2"), ("info", "info")):
        assert np.array_equal(iv[f], z["intv"][g])
    cn, ch, cs = sim.tap_chains()
    assert np.array_equal(cn, z["chain_n"])
    for f, g in (("n_seeds", "n"), ("rid", "rid"), ("w", "w"), ("kept", "kept"), ("is_alt", "is_alt"), ("frac_rep", "frac_rep"), ("pos", "pos")):
        assert np.array_equal(ch[f], z["chain_hdr"][g]), f
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(cs[f], z["chain_seeds"][f]), f
    rn, rr = sim.tap_regs_raw()
    assert np.array_equal(rn, z["raw_n"]) and rr.tobytes() == z["raw_regs"].astype(ALNREG_DTYPE).tobytes()


def test_hostsim_edge_cases_and_arena_growth(sim):
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    opt = default_opt()
    # empty batch
    c, r = sim.align(opt, np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
    assert len(c) == 0 and len(r) == 0
    # ragged: empty read, all-N read, reads shorter than the seed length, one long read
    rng = np.random.default_rng(5)
    base = simdata.make_reads_se(g, 200, length=300, seed=21)
    rag = [r_[: int(rng.integers(1, 300))] for r_ in base] + [np.zeros(0, dtype=np.uint8), np.full(60, 4, dtype=np.uint8), base[0][:18]]
    rag.append(simdata.make_reads_long(g, 1, length=1200, seed=22, sub=0.01, dele=0.005, ins=0.005)[0])
    seqs, off = testdata.ragged(rag)
    assert_regs_equal(*orc.align(opt, seqs, off), *sim.align(opt, seqs, off), "ragged")
    # a batch made of copies of a repeat element: far more seeds per read than the first arena guess -> growth + rerun
    rep = simdata.make_reads_se(g, 64, seed=23)
    hot = np.tile(rep[:1], (64, 1))
    seqs, off = testdata.flat(np.concatenate([hot, rep]))
    o2 = default_opt(); o2.max_occ = 2000
    assert_regs_equal(*orc.align(o2, seqs, off), *sim.align(o2, seqs, off), "arena growth")
    orc.close()
