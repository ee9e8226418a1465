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
from bwa_amd.structs import ALNREG_DTYPE, default_opt, pacbio_opt


@pytest.fixture(scope="module")
def sim():
    prefix, _ = testdata.small_index()
    s = BwaGpu(prefix, lib_path=hostsim_build.build())
    yield s
    s.close()


def sim_handle(prefix, **options):
    """A fresh handle on the mock runtime with the given library options (bwagpu_set_option's names, bwa_amd/csrc/bwagpu_config.h): the tests
    select kernel forms and test hooks through the API, not through the environment."""
    return BwaGpu(prefix, lib_path=hostsim_build.build(), options=options)


# The mock runs every lane as a fiber and every wave collective as a rendezvous, so the wave-cooperative extension
# kernel costs ~0.1 s per read here; the CPU suite therefore checks a prefix of each golden set (the GPU suite
# checks all of it).
N_SIM = {"se150": 48, "se150_N": 32, "se100_noisy": 32, "se250_odd": 16, "se40": 48, "se17": 50, "long2k_pacbio": 1, "long1500_default": 1}


def test_hostsim_regs_match_golden(sim):
    opts = golden_opts()
    for name, oname, reads, counts, regs in golden_sets(os.path.join(testdata.GOLDEN, "golden_regs.npz")):
        k = N_SIM.get(name, 16)
        seqs, off = testdata.flat(reads[:k])
        c, r = sim.align(opts[oname], seqs, off)
        assert_regs_equal(counts[:k], regs.astype(ALNREG_DTYPE)[: int(counts[:k].sum())], c, r, f"golden {name}")


def test_hostsim_long_read_dedup_ring_sizes():
    """k_dedup_wave's ring of {H,E} columns is sized by the batch's longest read, and its workgroups shrink to two waves (2048 columns) or one
    (4096) to stay within a workgroup's LDS; a ring too small for a patch alignment's band sends that alignment to the one-lane fall-back
    (256 columns): same regions in every case (2048 columns / two waves: the GPU suite's long-read tests).  The first run takes the round-3
    kernel forms (lane-per-read seeding with the compiler's load schedule, one-lane interval sort, one column per lane in the patch alignments),
    the second the long-read defaults: the one-round-trip seeding kernel (seed_mrg = 2: stack entries and read windows fetched a step ahead),
    the workgroup-per-read interval sort and SA-row expansion, four columns per lane."""
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    reads = simdata.make_reads_long(g, 1, length=1800, seed=23)
    seqs, off = testdata.flat(reads)
    want = orc.align(pacbio_opt(), seqs, off)
    for ring, mrg in (("256", "0"), ("4096", "2")):
        s2 = sim_handle(prefix, dedup_ring=int(ring), **({"seed_mrg": 0, "publish_blk": 0, "dedup_blk": 0, "seed_tasks": 0, "seedsw_lds": 0} if mrg == "0" else {"seed_tasks": 0}))
        assert s2.get_option("seed_mrg") == (0 if mrg == "0" else -1)
        s2.set_stats(True)
        assert_regs_equal(*want, *s2.align(pacbio_opt(), seqs, off), f"1.8 kb -x pacbio read, dedup ring {ring}, seeding variant {mrg}")
        if mrg == "2":       # a read too long for an LDS copy: its 16-base windows arrive one step ahead, with the index blocks
            import ctypes as C
            prof = (C.c_ulonglong * 16)()
            s2.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
            s2.L.bwagpu_debug_prof(s2.h, prof)
            assert prof[9] >= 1800 // 16, f"only {prof[9]} read windows were fetched ahead"
        s2.close()
    orc.close()


def test_hostsim_publish_per_workgroup():
    """Option publish_blk (long-read batches; their default): one workgroup per read sorts the interval list (bitonic network in LDS; equal keys are
    identical intervals), counts the SA rows and expands them from a block-wide prefix sum.  With -k 9 on the repeat-rich 2 Mb genome a
    1.3 kb read leaves ~800 intervals -- several 256-interval chunks -- next to a read with a handful and one with none: interval lists
    (order included), slot counts and regions equal the one-lane-per-read kernels'."""
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    reads = simdata.make_reads_long(g, 2, length=1300, seed=5)
    seqs, off = testdata.ragged([reads[1], reads[0][:40], np.full(300, 4, dtype=np.uint8), reads[0][:1150]])
    opt = pacbio_opt(); opt.min_seed_len = 9; opt.max_occ = 7
    got = {}
    for blk in ("0", "1"):
        s2 = sim_handle(prefix, publish_blk=int(blk))
        s2.set_stats(True)
        c, r = s2.align(opt, seqs, off)
        ic, iv = s2.tap_intervals()
        st = s2.stats()
        got[blk] = (c.tobytes(), r.tobytes(), ic.tobytes(), iv.tobytes(), st["n_seeds"], st["n_intv"])
        assert ic.max() > 512 and ic.min() == 0
        s2.close()
    assert got["0"] == got["1"]


@pytest.mark.parametrize("scale", [1, 2])
def test_hostsim_seed_rescoring_in_lds(scale):
    """Option seedsw_lds (long-read batches; their default): mem_seed_sw's local alignments (bwamem.c:597-621) with their DP rows, the window's query bases
    and the scoring matrix in LDS / registers -- 8-bit cells for a = 1, 16-bit cells for scaled scores: the seeds' scores after
    mem_flt_chained_seeds, the surviving seeds and the regions equal the HBM-scratch form's."""
    from bwa_amd.structs import fill_scmat
    prefix, g = testdata.small_index()
    reads = simdata.make_reads_long(g, 2, length=1400, seed=5)
    seqs, off = testdata.ragged([reads[0], reads[1][:1250], reads[0][:60]])
    opt = pacbio_opt()
    if scale == 2:
        opt.a = 2; opt.b = 2
        for x in ("o_del", "o_ins", "e_del", "e_ins", "zdrop", "pen_clip5", "pen_clip3"):
            setattr(opt, x, getattr(opt, x) * 2)
        fill_scmat(opt)
    got = {}
    for lds in ("0", "1"):
        s2 = sim_handle(prefix, seedsw_lds=int(lds))
        s2.set_stats(True)
        c, r = s2.align(opt, seqs, off)
        cc, ch, sd = s2.tap_chains()
        st = s2.stats()
        assert st["n_sw_cells"] > 100000 and (sd["score"] > 0).any()
        got[lds] = (c.tobytes(), r.tobytes(), cc.tobytes(), ch.tobytes(), sd.tobytes(), st["n_sw_cells"])
        s2.close()
    assert got["0"] == got["1"]


def test_hostsim_seeding_variants_same_intervals():
    """Seeding alone, in bulk: with a chain-weight threshold nothing passes, the stages after seeding have no work, so a few thousand reads
    of the repeat-rich 2 Mb genome (substitutions, indels, Ns) cost seconds on the mock runtime.  The interval lists (order included) of the
    one-round-trip kernels -- seed_mrg = 2, the same on a two-entry LDS stack, and without the LDS copy of the reads -- equal the default
    kernel's, whose intervals the golden-fixture tests pin to the reference."""
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    reads = list(simdata.make_reads_se(g, 700, seed=301, sub=0.02, dele=0.003, ins=0.003, n_frac=0.002))
    reads += [r_[: 40 + (i * 7) % 110] for i, r_ in enumerate(simdata.make_reads_se(g, 300, seed=302, sub=0.05))]
    reads += list(simdata.make_reads_se(g, 150, length=250, seed=303, sub=0.01))
    seqs, off = testdata.ragged(reads)
    opt = default_opt(); opt.min_chain_weight = 1 << 20
    got = {}
    for name, options in (("default", {}), ("mrg0", {"seed_mrg": 0}), ("mrg2", {"seed_mrg": 2}),
                          ("mrg2 small stack", {"seed_mrg": 2, "seed_lds_ent": 2}), ("mrg2 no read copy", {"seed_mrg": 2, "seed_rd_lds": 0})):
        s2 = sim_handle(prefix, **options)
        c, r = s2.align(opt, seqs, off)
        ic, iv = s2.tap_intervals()
        assert int(c.sum()) == 0 and int(ic.sum()) > 5 * len(reads)
        got[name] = (ic.tobytes(), iv.tobytes())
        s2.close()
    for name in got:
        assert got[name] == got["default"], name


def test_hostsim_long_read_pass1_by_tasks():
    """Option seed_tasks (long-read batches; their default): pass 1 of mem_collect_intv as independent tasks -- one bwt_smem1 search per read and
    min_seed_len-th position, each reporting the matches that START within the min_seed_len positions up to its own (so that every match of at
    least min_seed_len bases is reported exactly once, by the first multiple it covers), appended to the read's list through an atomic count --
    then the lane-per-read kernel from pass 2 on (k_seed's LR modes).  Noisy 3 kb reads, reads with N runs (on and off the task positions),
    nearly exact reads whose matches span dozens of task positions and feed pass 2, a read shorter than min_seed_len, an all-N and an empty read;
    several -k; with and without the one-round-trip fetch; with task stacks so small that most tasks are redone by the second launch; and with
    interval lists so short that the batch is redone: the interval lists equal the lane-per-read chain's (after the sort; equal keys are
    identical intervals), and so do the counts of index blocks' users downstream (slots)."""
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    reads = list(simdata.make_reads_long(g, 3, length=3000, seed=5))
    rng = np.random.default_rng(3)
    reads[1] = reads[1][:1700].copy(); reads[1][rng.integers(0, 1700, 12)] = 4; reads[1][500:530] = 4; reads[1][17 * 40] = 4; reads[1][17 * 41 - 1] = 4
    reads[2] = reads[2][:1151]
    reads += [reads[0][:90], reads[0][:12], np.full(200, 4, dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
    reads += list(simdata.make_reads_se(g, 2, length=1300, seed=9, sub=0.002))
    seqs, off = testdata.ragged(reads)
    for k in (17, 12, 23):
        opt = pacbio_opt(); opt.min_seed_len = k; opt.min_chain_weight = 1 << 20          # (nothing passes the chain filter: the stages after seeding have no work)
        if k == 23:
            opt.split_factor = 1.5                         # pass 2 in force: the nearly exact reads' long matches are re-seeded
        got = {}
        for name, options in (("chain", {"seed_tasks": 0, "seed_mrg": 0}), ("tasks", {"seed_mrg": 0}), ("tasks + one trip", {}), ("tiny stacks", {"seed_task_stack": 2, "seed_lds_ent": 3}),
                              ("short lists", {"mem_cap": 24})):
            if k != 17 and name in ("tasks", "short lists"):
                continue
            s2 = sim_handle(prefix, **options)
            s2.set_stats(True)
            c, r = s2.align(opt, seqs, off)
            ic, iv = s2.tap_intervals()
            st = s2.stats()
            got[name] = (ic.tobytes(), iv.tobytes(), st["n_seeds"])
            assert int(ic.max()) > 40
            if name == "short lists":
                assert st["n_retries"] >= 1
            import ctypes as C
            prof = (C.c_ulonglong * 16)()
            s2.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
            s2.L.bwagpu_debug_prof(s2.h, prof)
            assert (prof[8] > 20) if name == "tiny stacks" else (prof[8] == 0), (name, prof[8])      # tasks handed to the second launch
            s2.close()
        for name in got:
            assert got[name] == got["chain"], (k, name)


def test_hostsim_stage_taps_match_golden(sim):
    z = np.load(os.path.join(testdata.GOLDEN, "golden_stages.npz"))
    k = 40
    seqs, off = testdata.flat(z["reads"][:k])
    sim.align(golden_opts()["default"], seqs, off)
    n, iv = sim.tap_intervals()
    assert np.array_equal(n, z["intv_n"][:k])
    ni = int(z["intv_n"][:k].sum())
    for f, g in (("x0", "x0"), ("x2", "x2"), ("info", "info")):
        assert np.array_equal(iv[f], z["intv"][g][:ni])
    cn, ch, cs = sim.tap_chains()
    assert np.array_equal(cn, z["chain_n"][:k])
    nc = int(z["chain_n"][:k].sum())
    for f, g in (("n_seeds", "n"), ("rid", "rid"), ("w", "w"), ("kept", "kept"), ("is_alt", "is_alt"), ("frac_rep", "frac_rep"), ("pos", "pos")):
        assert np.array_equal(ch[f], z["chain_hdr"][g][:nc]), f
    ns = int(z["chain_hdr"]["n"][:nc].sum())
    for f in ("rbeg", "qbeg", "len", "score"):
        assert np.array_equal(cs[f], z["chain_seeds"][f][:ns]), f
    rn, rr = sim.tap_regs_raw()
    nr = int(z["raw_n"][:k].sum())
    assert np.array_equal(rn, z["raw_n"][:k]) and rr.tobytes() == z["raw_regs"].astype(ALNREG_DTYPE)[:nr].tobytes()


def test_hostsim_edge_cases_and_arena_growth(sim):
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    opt = default_opt()
    # empty batch
    c, r = sim.align(opt, np.zeros(0, dtype=np.uint8), np.zeros(1, dtype=np.int64))
    assert len(c) == 0 and len(r) == 0
    # ragged: empty read, all-N read, reads shorter than the seed length, one long read
    rng = np.random.default_rng(5)
    base = simdata.make_reads_se(g, 24, length=300, seed=21)
    rag = [r_[: int(rng.integers(1, 300))] for r_ in base] + [np.zeros(0, dtype=np.uint8), np.full(60, 4, dtype=np.uint8), base[0][:18]]
    seqs, off = testdata.ragged(rag)
    assert_regs_equal(*orc.align(opt, seqs, off), *sim.align(opt, seqs, off), "ragged")
    # one read longer than the wave kernel's LDS limit: the batch takes the lane-per-read extension kernel
    rag.append(simdata.make_reads_long(g, 1, length=1300, seed=22, sub=0.01, dele=0.005, ins=0.005)[0])
    seqs, off = testdata.ragged(rag)
    assert_regs_equal(*orc.align(opt, seqs, off), *sim.align(opt, seqs, off), "ragged + long")
    # a batch made of copies of a repeat element: far more seeds per read than the first arena guess -> growth + rerun
    rep = simdata.make_reads_se(g, 8, seed=23)
    hot = np.tile(rep[:1], (24, 1))
    seqs, off = testdata.flat(np.concatenate([hot, rep]))
    o2 = default_opt(); o2.max_occ = 2000
    assert_regs_equal(*orc.align(o2, seqs, off), *sim.align(o2, seqs, off), "arena growth")
    orc.close()


def _n_rich_reads(g, n, max_len, seed):
    """Ragged reads of 1..max_len bases with Ns sprinkled in: single Ns, runs, Ns at either end, an all-N and an empty read."""
    rng = np.random.default_rng(seed)
    base = simdata.make_reads_se(g, n, length=max_len, seed=seed, sub=0.01)
    out = []
    for i, r_ in enumerate(base):
        r_ = r_[: int(rng.integers(max_len // 3, max_len + 1))].copy()
        kind = i % 5
        if kind == 1: r_[int(rng.integers(0, len(r_)))] = 4
        elif kind == 2: a = int(rng.integers(0, len(r_) - 6)); r_[a:a + 5] = 4
        elif kind == 3: r_[0] = 4; r_[-1] = 4
        elif kind == 4 and len(r_) > 40: r_[16] = 4; r_[31] = 4; r_[32] = 4       # on and around the 16-base words of the LDS copy
        out.append(r_)
    out += [np.zeros(0, dtype=np.uint8), np.full(40, 4, dtype=np.uint8), base[0][:max_len], base[1][:17]]
    return out


@pytest.mark.parametrize("max_len,env", [(150, {}), (250, {}), (120, {"seed_no_virt": 1}), (150, {"seed_rd_lds": 0}),
                                         (150, {"ptab_m": 5, "seed_lds_ent": 2}), (150, {"occ32": 0}), (150, {"occ32": 0, "ptab_m": 0}), (150, {"ptab_m": 0}),
                                         (150, {"seed_mrg": 2}), (250, {"seed_mrg": 2, "seed_lds_ent": 2}),
                                         (150, {"seed_mrg": 2, "seed_rd_lds": 0, "ptab_m": 5}), (150, {"seed_mrg": 2, "seed_no_virt": 1, "seed_lds_ent": 1})])
def test_hostsim_seeding_paths_with_n_reads(max_len, env):
    """The seeding kernel's read copy in LDS (2 bits per base; 8, 12 or 16 words per lane by the batch's longest read; reads with an N
    take their bases from global memory), the short stack entries kept as a bit mask (off with option seed_no_virt, and narrower with
    shallow prefix tables), a two-entry LDS stack that spills almost everything, and the two ways of reading the index -- the 32-byte
    layout (the default) and the reference-format 64-byte blocks (occ32 = 0) -- with and without prefix tables; and the one-round-trip form
    of the 32-byte layout (seed_mrg = 2: table entries and whole blocks through range-checked buffer loads, the next interval-stack entry
    fetched a step ahead, with LDS stacks so small that nearly every backward step takes a prefetched entry): same regions as the oracle for
    ragged reads with Ns.  (Options are given through the API: bwagpu_set_default_option before the handle is created.)"""
    prefix, g = testdata.small_index()
    s2 = sim_handle(prefix, **env)
    orc = orcapi.OrcIndex(prefix)
    seqs, off = testdata.ragged(_n_rich_reads(g, 14, max_len, seed=300 + max_len))
    assert_regs_equal(*orc.align(default_opt(), seqs, off), *s2.align(default_opt(), seqs, off), f"N-rich ragged reads up to {max_len} bp, {env}")
    s2.close(); orc.close()


def test_hostsim_occ32_layout_across_superblocks():
    """The 32-byte block layout on the 2 Mb genome with superblocks of 2^12 bases (a thousand of them, so the relative counts and the superblock
    table are both in play; the default of 2^32 bases gives a genome this size one): same regions as the oracle, with the prefix tables
    (filled through the new layout) and without."""
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    orc = orcapi.OrcIndex(prefix)
    seqs, off = testdata.flat(simdata.make_reads_se(g, 16, seed=88, sub=0.02))
    want = orc.align(default_opt(), seqs, off)
    for m, shift, mrg in (("6", "12", "0"), ("0", "12", "0"), ("6", "32", "0"), ("6", "12", "2"), ("6", "32", "2")):      # (seed_mrg = 2: the per-symbol form of the superblock table)
        s2 = sim_handle(prefix, occ32=1, ptab_m=int(m), occ32_sb_shift=int(shift), seed_mrg=int(mrg))
        assert s2.get_option("occ32_sb_shift") == int(shift)
        assert_regs_equal(*want, *s2.align(default_opt(), seqs, off), f"32-byte blocks, prefix tables {m}, superblock shift {shift}")
        s2.close()
    orc.close()


@pytest.mark.parametrize("env", [{}, {"occ32": 0}])
def test_hostsim_densified_sa_equals_the_walk(env):
    """bwagpu_densify_sa fills the new samples from one LF walk per OLD sample (k_densify); every kept row must hold what the oracle's
    bwt_sa returns for it (bwt.c:91-103), including row 0 (-1), the primary row and the last row, at intervals 8, 2 and 1."""
    import ctypes as C
    prefix, _ = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    for intv in (8, 2, 1):
        s2 = sim_handle(prefix, **env)
        meta0 = s2.index_meta()
        s2.densify_sa(intv)
        meta = s2.index_meta()
        assert meta["sa_intv"] == intv and meta["n_sa"] == (meta["seq_len"] + intv) // intv and meta0["sa_intv"] == 32
        ptr, nbytes = s2.index_buffers()[1]
        assert nbytes == meta["n_sa"] * 8
        sa = np.frombuffer((C.c_uint64 * meta["n_sa"]).from_address(ptr), dtype=np.uint64)     # (mock runtime: device memory is host memory)
        assert sa[0] == np.uint64(2 ** 64 - 1)
        rows = np.arange(1, meta["n_sa"], max(1, meta["n_sa"] // 4000)).tolist() + [meta["n_sa"] - 1, meta["primary"] // intv, meta["primary"] // intv + 1]
        for r in rows:
            if 0 < r < meta["n_sa"]:
                assert int(sa[r]) == orc.sa(r * intv), f"interval {intv}, row {r * intv}"
        s2.close()
    orc.close()


def test_hostsim_min_seed_len_around_the_table_depth(sim):
    """-k at, below and just above the prefix tables' depth (10): at or below it short matches can be reported, so every stack
    entry is stored; just above it the shortest stored entry is the one a backward row has just grown to the tables' depth."""
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    seqs, off = testdata.flat(simdata.make_reads_se(g, 8, seed=77, sub=0.04))
    for k in (9, 10, 11):
        opt = default_opt(); opt.min_seed_len = k
        assert_regs_equal(*orc.align(opt, seqs, off), *sim.align(opt, seqs, off), f"min_seed_len {k}")
    orc.close()


def test_hostsim_cloned_handle_shares_index(sim):
    """bwagpu_clone: a second handle on the same resident index gives the same results; both stay usable and are destroyed
    independently (the mock runtime is single-threaded, so the two are driven one after the other here)."""
    prefix, g = testdata.small_index()
    seqs, off = testdata.flat(simdata.make_reads_se(g, 24, seed=91))
    opt = default_opt()
    a = sim.align(opt, seqs, off)
    other = sim.clone()
    b = other.align(opt, seqs, off)
    assert_regs_equal(*a, *b, "clone")
    other.close()
    assert_regs_equal(*a, *sim.align(opt, seqs, off), "original after the clone is destroyed")


@pytest.mark.parametrize("results", [1, 0])
def test_hostsim_batch_reserve(results):
    """bwagpu_batch_reserve ahead of the first batch (device buffers of a batch of that shape and, unless reserve_results = 0, the page-locked result blocks,
    handed to the pool the downloads draw from): the batch that follows -- smaller, and then larger than what was reserved -- gives the same regions as a
    handle that reserved nothing."""
    prefix, g = testdata.small_index()
    opt = default_opt()
    plain = sim_handle(prefix)
    h = sim_handle(prefix, reserve_results=results, pinned_min_kb=0)
    assert h.L.bwagpu_batch_reserve(h.h, 64, 64 * 150, 150) == 0
    assert h.L.bwagpu_batch_reserve(h.h, 0, 0, 150) != 0            # (an empty shape is an argument error, and leaves the handle usable)
    for n, seed in ((40, 61), (200, 62)):
        seqs, off = testdata.flat(simdata.make_reads_se(g, n, seed=seed))
        assert_regs_equal(*plain.align(opt, seqs, off), *h.align(opt, seqs, off), f"after bwagpu_batch_reserve, {n} reads")
    h.close(); plain.close()


def test_hostsim_interval_list_overflow_retries():
    """A deliberately tiny per-read interval capacity: the seeding kernel flags the overflow and the batch is re-run with a
    larger capacity; results are unchanged."""
    import hostsim_build
    prefix, g = testdata.small_index()
    s2 = sim_handle(prefix, mem_cap=3)
    seqs, off = testdata.flat(simdata.make_reads_se(g, 24, seed=92))
    orc = orcapi.OrcIndex(prefix)
    assert_regs_equal(*orc.align(default_opt(), seqs, off), *s2.align(default_opt(), seqs, off), "interval overflow")
    assert s2.stats()["n_retries"] >= 1
    s2.close(); orc.close()


@pytest.mark.parametrize("ent,mrg", [("1", "0"), ("3", "0"), ("1", "2"), ("3", "2")])
def test_hostsim_lds_stack_ring_eviction(ent, mrg):
    """A tiny LDS interval stack: the forward sweep's ring wraps and evicts to the HBM spill area, and the backward sweep
    reads deep entries back from it; the seeds (and everything downstream) are unchanged."""
    prefix, g = testdata.small_index()
    s2 = sim_handle(prefix, seed_lds_ent=int(ent), seed_mrg=int(mrg))   # (2: deep entries arrive one step ahead of their use)
    seqs, off = testdata.flat(simdata.make_reads_se(g, 14, seed=93))
    orc = orcapi.OrcIndex(prefix)
    s2.set_stats(True)
    assert_regs_equal(*orc.align(default_opt(), seqs, off), *s2.align(default_opt(), seqs, off), f"LDS stack of {ent}")
    import ctypes as C
    prof = (C.c_ulonglong * 16)()
    s2.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
    s2.L.bwagpu_debug_prof(s2.h, prof)
    assert prof[10] > 20, "the test's reads no longer spill their interval stacks"
    assert prof[11] == (prof[10] if mrg == "2" else 0), "every deep entry of a backward row is fetched one step ahead (and only with seed_mrg = 2)"
    s2.close(); orc.close()


def test_hostsim_device_cigars_give_the_same_sam(sim, monkeypatch):
    """bwagpu_batch_cigars (banded global alignment + traceback on the device): feeding its records to the
    host finalize code as hints yields exactly the SAM text the host produces when it runs every DP itself, and a good
    share of the regions is actually served by the device."""
    import hostapi
    from bwa_amd.api import CIGAR_DTYPE
    prefix, g = testdata.small_index()
    host = hostapi.HostFinalize(prefix)
    reads = simdata.make_reads_se(g, 40, seed=95, sub=0.03, dele=0.004, ins=0.004)
    seqs, off = testdata.flat(reads)
    opt = default_opt()
    counts, regs = sim.align(opt, seqs, off)
    cigs = sim.cigars(opt)
    assert cigs.dtype == CIGAR_DTYPE and cigs.shape[0] == regs.shape[0]
    names = [f"q{i}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    ops = sim.cigar_ops()
    want, want_ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
    assert hostapi.decode_cigars(cigs, ops) == hostapi.decode_cigars(want, want_ops), "device records differ from the host's"
    assert ops.shape[0] == want_ops.shape[0]
    plain = host.regs2sam(opt, names, seqs, quals, off, counts, regs)
    hinted = host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, cig_ops=ops)
    assert hinted == plain
    served = int((cigs["n_cigar"] >= 0).sum())
    assert served >= regs.shape[0] // 2, (served, regs.shape[0])
    gapped = int(((cigs["n_cigar"] > 1)).sum())
    assert gapped > 0, "no gapped alignment exercised the traceback"
    # gap-rich reads: alignments of 7..64 operations go through the operation array
    reads = simdata.make_reads_se(g, 12, seed=96, sub=0.02, dele=0.03, ins=0.03)
    seqs, off = testdata.flat(reads)
    counts, regs = sim.align(opt, seqs, off)
    cigs, ops = sim.cigars(opt), sim.cigar_ops()
    want, want_ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
    assert hostapi.decode_cigars(cigs, ops) == hostapi.decode_cigars(want, want_ops)
    assert int((cigs["n_cigar"] > 6).sum()) >= 4, cigs["n_cigar"]
    sim.set_option("cig_ops_cap", 10)       # an operation array that is too small: the pass is redone with the size it asked for
    cigs2, ops2 = sim.cigars(opt), sim.cigar_ops()
    assert hostapi.decode_cigars(cigs2, ops2) == hostapi.decode_cigars(want, want_ops) and ops2.shape[0] == want_ops.shape[0]
    sim.set_option("cig_ops_cap", 0)
    names = [f"q{i}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, cig_ops=ops) == host.regs2sam(opt, names, seqs, quals, off, counts, regs)
    host.close()


def test_hostsim_pooled_result_buffers(sim, monkeypatch):
    """The large per-batch results come from a pool of page-locked blocks that bwagpu_free refills (a threshold of 0 KiB sends the small
    results of this test through it): the same records as with plain malloc, over several batches of different sizes, blocks re-used."""
    prefix, g = testdata.small_index()
    opt = default_opt()
    sets = [testdata.flat(simdata.make_reads_se(g, n, seed=170 + n, sub=0.03, dele=0.004, ins=0.004)) for n in (10, 4, 10)]
    want = []
    sim.set_option("pinned_results", 0)
    for seqs, off in sets:
        c, r = sim.align(opt, seqs, off)
        want.append((c.copy(), r.copy(), sim.cigars(opt).copy(), sim.cigar_ops().copy()))
    sim.set_option("pinned_results", 1); sim.set_option("pinned_min_kb", 0)
    for (seqs, off), (c0, r0, g0, o0) in zip(sets, want):
        c, r = sim.align(opt, seqs, off)
        cg, ops = sim.cigars(opt), sim.cigar_ops()
        assert np.array_equal(c, c0) and r.tobytes() == r0.tobytes() and cg.tobytes() == g0.tobytes() and ops.tobytes() == o0.tobytes()
    # bwagpu_alloc_host: input buffers from the same pool -- a block comes back after bwagpu_free, and a batch whose base codes lie in one
    # aligns like any other
    import ctypes as C
    L = sim.L
    L.bwagpu_alloc_host.restype = C.c_void_p; L.bwagpu_alloc_host.argtypes = [C.c_size_t]; L.bwagpu_free.argtypes = [C.c_void_p]
    p1 = L.bwagpu_alloc_host(3 << 20)
    assert p1
    L.bwagpu_free(p1)
    p2 = L.bwagpu_alloc_host(3 << 20)
    assert p2 == p1, "a freed block of the pool is handed out again"
    seqs, off = sets[0]
    C.memmove(p2, seqs.ctypes.data, seqs.nbytes)
    pinned = np.ctypeslib.as_array(C.cast(p2, C.POINTER(C.c_uint8)), shape=(seqs.nbytes,))
    c, r = sim.align(opt, pinned, off)
    assert np.array_equal(c, want[0][0]) and r.tobytes() == want[0][1].tobytes()
    L.bwagpu_free(p2)
    sim.set_option("pinned_min_kb", 1024)


def test_hostsim_long_segment_cigars(sim, monkeypatch):
    """The third tier of bwagpu_batch_cigars (k_cigar_long: columns in an LDS ring, direction bytes in HBM, tiled traceback, operations
    and MD strings of any length): for noisy 1.3 kb -x pacbio reads -- hundreds of operations per alignment -- the records equal the host
    code's and the hinted SAM equals the plain one; every region is served by the device."""
    import hostapi
    prefix, g = testdata.small_index()
    host = hostapi.HostFinalize(prefix)
    reads = simdata.make_reads_long(g, 2, length=1300, seed=97)
    seqs, off = testdata.flat(reads)
    opt = pacbio_opt()
    counts, regs = sim.align(opt, seqs, off)
    cigs, ops = sim.cigars(opt), sim.cigar_ops()
    want, want_ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
    assert hostapi.decode_cigars(cigs, ops) == hostapi.decode_cigars(want, want_ops), "device records differ from the host's"
    ok = regs["score"] >= opt.T
    assert (cigs["n_cigar"][ok] > 64).sum() >= 2 and (cigs["n_cigar"][ok] >= 0).all(), cigs["n_cigar"]
    budget = sim.get_option("cigl_mib")
    sim.set_option("cigl_mib", 0)       # a scratch budget below one direction matrix: one workgroup takes the tier's whole work list
    cigs1, ops1 = sim.cigars(opt), sim.cigar_ops()
    assert hostapi.decode_cigars(cigs1, ops1) == hostapi.decode_cigars(cigs, ops)
    sim.set_option("cigl_mib", budget)
    names = [f"q{i}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, cigs=cigs, cig_ops=ops) == host.regs2sam(opt, names, seqs, quals, off, counts, regs)
    host.close()


def test_hostsim_long_reads_ring_extension(sim):
    """Reads beyond the short-read limit take the wave extension kernel in ring mode ({H,E} columns of the band only, lazily
    initialised) -- same regions as the oracle for noisy 1.8 kb reads with the pacbio preset and a 1.5 kb read with defaults (the GPU suite runs 4-5 kb)."""
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    for n, opt, kw in ((2, pacbio_opt(), dict(length=1800, sub=0.05, dele=0.04, ins=0.04)), (1, default_opt(), dict(length=1500, sub=0.02, dele=0.005, ins=0.005))):
        seqs, off = testdata.flat(simdata.make_reads_se(g, n, seed=97, **kw))
        assert_regs_equal(*orc.align(opt, seqs, off), *sim.align(opt, seqs, off), f"long reads {kw}")
    orc.close()


def test_hostsim_device_matesw_records_match_the_host(sim):
    """bwagpu_batch_matesw: for a paired batch whose second mates are too noisy to map on their own, the
    device's task list and ksw_align2 results equal the host code's (same records, any order), and feeding them to the
    finalize stage leaves the SAM text unchanged."""
    import hostapi
    from bwa_amd.api import MATESW_DTYPE, PES_DTYPE
    prefix, g = testdata.small_index()
    host = hostapi.HostFinalize(prefix)
    r1, r2 = simdata.make_reads_pe(g, 40, seed=98)
    rng = np.random.default_rng(99)
    r2 = np.where(rng.random(r2.shape) < 0.14, (r2 + rng.integers(1, 4, r2.shape)) % 4, r2).astype(np.uint8)
    reads = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
    seqs, off = testdata.flat(reads)
    opt = default_opt(); opt.flag |= 2
    counts, regs = sim.align(opt, seqs, off)
    pes = host.pestat(opt, counts, regs)
    if (pes["failed"] != 0).all():          # tiny batch: give the FR orientation the simulated library's window
        pes["failed"][1] = 0; pes["low"][1] = 200; pes["high"][1] = 600
    dpes = np.zeros(4, dtype=PES_DTYPE)
    for k in ("low", "high", "failed"):
        dpes[k] = pes[k]
    got = sim.matesw(opt, dpes)
    want = host.matesw_records(opt, seqs, off, counts, regs, pes)
    assert got.dtype == MATESW_DTYPE and got.shape == want.shape
    key = lambda a: np.sort(np.frombuffer(a.tobytes(), dtype=f"V{MATESW_DTYPE.itemsize}"))
    assert (key(got) == key(want)).all(), "device mate-rescue records differ from the host's"
    assert (got["r"] >= 0).sum() >= 8, "too few rescue alignments to mean anything"
    names = [f"q{i >> 1}" for i in range(off.shape[0] - 1)]
    quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
    import ctypes as C
    p0 = pes.ctypes.data_as(C.c_void_p)
    plain = host.regs2sam(opt, names, seqs, quals, off, counts, regs, pes0=p0)
    assert host.regs2sam(opt, names, seqs, quals, off, counts, regs, pes0=p0, msw=got) == plain
    host.close()


@pytest.fixture(scope="module")
def heavy_case():
    import heavycase
    fa, orc, reads = heavycase.build()
    yield fa, orc, reads
    orc.close()


@pytest.mark.parametrize("regs,flt_lds", [(2, 256), (1, 256), (0, 256), (2, 0), (2, 40)])
def test_hostsim_wave_chaining_heavy_reads(monkeypatch, heavy_case, regs, flt_lds):
    """k_chain_wave on reads with many chains (repeat-rich genome).  Register form (chains one per lane, exact for distinct positions and for one-node
    trees): lane shifts on insertion, kbtree's rules for equal positions, the hand-over to the tree form at the 65th chain or at a duplicate position in a
    larger tree.  Tree form (all of it with chain_regs = 0): multi-level B-trees with splits, duplicate keys, the root node cached in registers across
    look-ups and insertions.  Then the weight sort (quicksort passes by one lane, the stable finish by all), the 64-wide chain filter with its arrays in LDS
    or (chain_flt_lds below a read's chain count) in the read's HBM region, and the flattening of hundreds of chains.  All must reproduce the oracle's chains exactly.
    (The same on hardware, against the compiled reference: tests/test_gpu_parity.py::test_gpu_wave_chaining_heavy_reads.)"""
    import heavycase
    fa, orc, reads = heavy_case
    s2 = sim_handle(fa, chain_regs=regs, chain_flt_lds=flt_lds)
    heavycase.check(s2, orc, reads, regs)
    s2.close()


def dedup_list_reads(dev, n_reads):
    """Reads the wave-per-read de-duplication kernel did (its stats histogram, bwagpu_debug_hist out[160..192))."""
    import ctypes as C
    hist = (C.c_ulonglong * 256)()
    dev.L.bwagpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
    assert dev.L.bwagpu_debug_hist(dev.h, hist) == 0
    return sum(hist[160:192])


@pytest.mark.parametrize("heavy_min,stage,big,net", [(0, -1, -1, -1), (-1, -1, -1, -1), (2, 16, -1, 12), (2, 16, 32, 0), (8, 0, -1, -1), (-1, 512, 0, 4)])
def test_hostsim_dedup_hands_heavy_reads_to_the_wave_kernel(heavy_case, heavy_min, stage, big, net):
    """k_dedup keeps the reads with few regions (one lane each) and lists the others for k_dedup_wave<.., LIST> (option dedup_heavy: 0 = none,
    auto = 3 regions), which runs dedup_read_par (dev_dedupp.h: operands in LDS, the lanes over the regions -- the stable finish of both sorts, the
    redundancy scan 64 regions at a time, compactions by prefix counts) in two launches: reads of up to dedup_stage regions (auto 128; 0 = every
    listed read in place in HBM, the long reads' routine), then those of up to dedup_big (auto: what 64 KB hold; beyond it in place).  The regions equal
    the oracle's whichever way a read went, and the list is used when it should be.  dedup_net: the reads whose sorts are finished by the sorting network
    (auto: 129 regions and more, i.e. none of these reads; 12 / 4: most of the listed ones; 0: none)."""
    fa, orc, reads = heavy_case
    opt = default_opt()
    s2 = sim_handle(fa, dedup_heavy=heavy_min, dedup_stage=stage, dedup_big=big, dedup_net=net)
    s2.set_stats(True); s2.set_taps(True)
    more = simdata.make_reads_se(simdata.make_genome(500_000, n_contigs=2, seed=5, n_interspersed=2000, divergence=0.04)[0], 40, seed=85, sub=0.05)   # (heavycase.build's genome: reads of 1..80 regions)
    seqs, off = testdata.flat(np.concatenate([reads, more]))
    c, r = s2.align(opt, seqs, off)
    assert_regs_equal(*orc.align(opt, seqs, off), c, r, f"dedup_heavy={heavy_min} dedup_stage={stage} dedup_big={big} dedup_net={net}")
    n_raw = s2.tap_regs_raw()[0]
    done = dedup_list_reads(s2, reads.shape[0])
    lo = 3 if heavy_min < 0 else heavy_min
    assert done == (int((n_raw >= lo).sum()) if heavy_min else 0), (done, n_raw)
    assert heavy_min == 0 or (done >= 6 and ((n_raw >= lo) & (n_raw <= 16)).sum() >= 2), n_raw
    assert stage != 16 or ((n_raw > 16) & (n_raw <= 32)).sum() >= 2 and (n_raw > 32).sum() >= 2, n_raw          # (reads for the second launch, and beyond its arrays when dedup_big = 32)
    s2.close()


@pytest.mark.parametrize("heavy_min", [-1, 2, 0])
def test_hostsim_dedup_patch_joins(heavy_min):
    """mem_patch_reg in the de-duplication kernels: reads that break into regions on one diagonal and are joined again (tests/heavycase.py::patch_reads),
    through k_dedup alone (dedup_heavy = 0) and through the listed reads' dedup_read_par, where a patch alignment is the one event of the redundancy
    scan that is handled in sequence -- the scan goes on behind it with the changed region.  Regions equal the oracle's; joins did happen."""
    import heavycase
    prefix, g = testdata.small_index()
    orc = orcapi.OrcIndex(prefix)
    reads, opt = heavycase.patch_reads(g, 24), heavycase.patch_opt()
    seqs, off = testdata.flat(reads)
    want = orc.align(opt, seqs, off)
    assert int(((want[1]["ncomp_isalt"] & 0x3fffffff) > 1).sum()) >= 8, "no joined regions in the test's reads"
    s2 = sim_handle(prefix, dedup_heavy=heavy_min)
    s2.set_stats(True)
    assert_regs_equal(*want, *s2.align(opt, seqs, off), f"patch joins, dedup_heavy={heavy_min}")
    assert s2.stats()["n_glb_calls"] >= 100
    assert dedup_list_reads(s2, reads.shape[0]) >= (20 if heavy_min else 0)
    s2.close(); orc.close()


def _alt_prefix(tmp_path, prefix, alt_names):
    """The same index under a new prefix, plus a .alt file (first column = contig name, bntseq.c:185-205)."""
    new = str(tmp_path / "alt_idx")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        os.symlink(os.path.abspath(prefix + "." + ext), new + "." + ext)
    with open(new + ".alt", "w") as f:
        f.write("@SQ\tSN:ignored header line\n")
        for nme in alt_names:
            f.write(f"{nme}\t0\tchr1\t1\t60\t100M\t*\t0\t0\t*\t*\n")
    return new


def test_hostsim_alt_contigs(tmp_path):
    """ALT contigs on the device path (is_alt from the .alt file -> mem_chain's chain records, mem_chain_flt's ALT-aware overlap rule
    bwamem.c:377, mem_sort_dedup_patch, the regions' is_alt, bwamem.c:1113-1114): regions equal the oracle's with the same contig
    flagged, and differ from the run without the .alt file (so the flag demonstrably reaches the kernels)."""
    prefix, g = testdata.small_index()
    alt = _alt_prefix(tmp_path, prefix, ["chr3"])
    orc = orcapi.OrcIndex(prefix); orc.set_alt(2, 1)
    s_alt, s_plain = BwaGpu(alt, lib_path=hostsim_build.build()), BwaGpu(prefix, lib_path=hostsim_build.build())
    # reads from the ALT contig and from the repeats the contigs share
    lens = testdata.small_genome()[1]
    lo = sum(lens[:2])
    reads = np.concatenate([simdata.make_reads_se(g[lo:], 30, seed=98), simdata.make_reads_se(g, 30, seed=99)])
    seqs, off = testdata.flat(reads)
    opt = default_opt()
    c, r = s_alt.align(opt, seqs, off)
    assert_regs_equal(*orc.align(opt, seqs, off), c, r, "ALT contig")
    assert int((r["ncomp_isalt"] >> 30).sum()) > 0, "no region on the ALT contig"
    c0, r0 = s_plain.align(opt, seqs, off)
    assert r.tobytes() != r0.tobytes()
    s_alt.close(); s_plain.close(); orc.close()


def test_hostsim_long_read_four_columns_per_lane():
    """k_dedup_wave's score-only patch alignments (mem_patch_reg -> ksw_global2) with four adjacent columns per lane (wave_global2_score_ring_blk,
    the default) and with one (option dedup_blk = 0): a 7 kb -x pacbio read drifts out of the extension's band, so its regions are merged by a
    patch alignment of ~7000 x 750 cells (three passes of 256 columns per row, or twelve of 64).  The regions equal the compiled reference's
    either way, and the batch did run a patch alignment."""
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    ref = refapi.RefIndex(prefix)
    reads = simdata.make_reads_long(g, 1, length=7000, seed=78)
    seqs, off = testdata.flat(reads)
    want = ref.align(pacbio_opt(), seqs, off)
    for env in ({}, {"dedup_blk": 0}):
        s2 = sim_handle(prefix, ptab_m=6, **env)
        s2.set_stats(True)
        assert_regs_equal(*want, *s2.align(pacbio_opt(), seqs, off), f"7 kb -x pacbio read, {env}")
        st = s2.stats()
        assert st["n_glb_calls"] >= 1 and st["n_glb_cells"] > 4_000_000 and st["n_ext_cells"] > 1_000_000, st
        s2.close()


def test_hostsim_heavy_reads_seeded_by_tasks():
    """Short-read batches: a read on which a lane of the lane-per-read seeding kernel has spent more than option seed_budget iterations (4096 by
    default) is given up there and listed; the listed reads take pass 1 as independent searches at every min_seed_len-th position and pass 2 as
    one task per qualifying pass-1 entry.  With a budget of 150 iterations every read that is not trivially cheap takes that route (repeat-rich 2 Mb
    genome; ragged reads with Ns, reads shorter than the seed length, -k 19 and -k 11 with -r 1 so that pass 2 is busy): interval lists, regions
    and slot counts equal the lane-per-read route's and the oracle's; a pass-2 task list of two entries forces the retry."""
    import ctypes as C
    import refapi
    if not refapi.have_ref():
        pytest.skip("oracle/_ref not built (needed to index the 2 Mb genome)")
    prefix, g = testdata.medium_index()
    orc = orcapi.OrcIndex(prefix)
    reads = list(simdata.make_reads_se(g, 60, seed=601, sub=0.02, dele=0.003, ins=0.003, n_frac=0.002))
    reads += [r_[: 30 + (i * 11) % 120] for i, r_ in enumerate(simdata.make_reads_se(g, 30, seed=602, sub=0.04))]
    reads += [reads[0][:12], np.full(60, 4, dtype=np.uint8), np.zeros(0, dtype=np.uint8)]
    seqs, off = testdata.ragged(reads)
    o11 = default_opt(); o11.min_seed_len = 11; o11.split_factor = 1.0
    for name, opt in (("-k 19", default_opt()), ("-k 11 -r 1", o11)):
        want = orc.align(opt, seqs, off)
        got = {}
        # (last: an index that counts as beyond a buffer descriptor's reach -- option idx_desc_max_mb, the 2 Mb genome's 32-byte blocks are 2 MiB --
        # has no task kernels to hand reads to: the budget must be off with it, whatever seed_budget says, and the plain-load kernels run)
        for cfg_name, options in (("lane per read", {"seed_budget": 0}), ("tasks", {"seed_budget": 150}), ("tasks, tiny pass-2 list", {"seed_budget": 150, "seed_p2_cap": 2}),
                                  ("index beyond a descriptor", {"seed_budget": 150, "idx_desc_max_mb": 1})):
            s2 = sim_handle(prefix, **options)
            s2.set_stats(True)
            c, r = s2.align(opt, seqs, off)
            assert_regs_equal(*want, c, r, f"{name}, {cfg_name}")
            ic, iv = s2.tap_intervals()
            st = s2.stats()
            got[cfg_name] = (ic.tobytes(), iv.tobytes(), st["n_seeds"])
            prof = (C.c_ulonglong * 16)()
            s2.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
            s2.L.bwagpu_debug_prof(s2.h, prof)
            if cfg_name in ("lane per read", "index beyond a descriptor"):
                assert prof[0] == 0, (name, cfg_name, prof[0])
            else:
                assert prof[0] >= 5 and prof[1] >= 5, (name, cfg_name, prof[0], prof[1])      # heavy reads, pass-2 tasks
            if cfg_name == "tasks, tiny pass-2 list":
                assert st["n_retries"] >= 1
            s2.close()
        for k in got:
            assert got[k] == got["lane per read"], (name, k)
    orc.close()


def test_hostsim_option_api_semantics(monkeypatch):
    """include/bwagpu.h's option calls: the names enumerate (bwagpu_option_name) and match bwagpu_config.h's list; an unknown name is
    BWAGPU_EINVAL for all three setters/getters; a handle takes compiled-in default <- BWAGPU_<NAME> in the environment AT CREATION <-
    bwagpu_set_default_option; a clone copies its parent's values; bwagpu_set_option changes one handle only; and the environment is not
    read again after creation (no batch call may depend on it)."""
    import ctypes as C
    lib = hostsim_build.build()
    prefix, g = testdata.small_index()
    a = BwaGpu(prefix, lib_path=lib)
    L = a.L
    L.bwagpu_option_name.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
    names = []
    for i in range(200):
        p = C.c_char_p()
        if L.bwagpu_option_name(i, C.byref(p)) != 0:
            break
        names.append(p.value.decode())
    import re
    listed = re.findall(r"^\s*X\((\w+),", open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bwa_amd", "csrc", "bwagpu_config.h")).read(), flags=re.M)
    assert names == listed and "seed_budget" in names and len(set(names)) == len(names)
    v = C.c_longlong()
    assert L.bwagpu_get_option(a.h, b"no_such_option", C.byref(v)) != 0
    assert L.bwagpu_set_option(a.h, b"no_such_option", 1) != 0
    assert L.bwagpu_set_default_option(b"no_such_option", 1) != 0
    assert a.get_option("seed_budget") == -1 and a.get_option("ext_occ") == 6
    # environment: read when a handle is created, never afterwards
    monkeypatch.setenv("BWAGPU_EXT_OCC", "4")
    assert a.get_option("ext_occ") == 6
    b = BwaGpu(prefix, lib_path=lib)
    assert b.get_option("ext_occ") == 4
    monkeypatch.delenv("BWAGPU_EXT_OCC")
    assert b.get_option("ext_occ") == 4
    # defaults given through the API win over the environment's absence, reach later handles only, and can be forgotten
    assert L.bwagpu_set_default_option(b"seed_budget", 1234) == 0
    c = BwaGpu(prefix, lib_path=lib)
    assert c.get_option("seed_budget") == 1234 and a.get_option("seed_budget") == -1
    L.bwagpu_clear_default_options()
    d = BwaGpu(prefix, lib_path=lib)
    assert d.get_option("seed_budget") == -1
    # per handle; clones copy
    c.set_option("dedup_ring", 512)
    c2 = c.clone()
    assert c2.get_option("dedup_ring") == 512 and c2.get_option("seed_budget") == 1234 and d.get_option("dedup_ring") == 0
    c2.set_option("dedup_ring", 1024)
    assert c.get_option("dedup_ring") == 512
    for h_ in (c2, c, d, b, a):
        h_.close()
