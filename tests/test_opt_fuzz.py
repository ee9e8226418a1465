"""Option-space fuzz: random mem_opt_t draws (the fields fastmap.c:159-259 sets and the presets of fastmap.c:330-358) through the
device path, regions compared bit for bit with the compiled reference's mem_align1_core; and random `bwa mem` command lines
(incl. the host-side options -U -m -S -P -a -M -Y -5 -q -h -z -T) through `bwa-amd mem` vs `bwa mem`, SAM compared byte for byte.

CPU: a handful of draws on the mock runtime.  -m gpu: 40 draws x 2000 reads (long-read presets: 200 x 1.5 kb) and 16 command lines."""
import os
import subprocess

import numpy as np
import pytest

import refapi
import testdata
from cmputil import assert_regs_equal
from bwa_amd import simdata
from bwa_amd.structs import default_opt, fill_scmat

pytestmark = pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")


def preset(name):
    """fastmap.c:330-358 applied to mem_opt_init's defaults (no option given on the command line)"""
    o = default_opt()
    if name == "intractg":
        o.o_del = o.o_ins = 16; o.b = 9; o.pen_clip5 = o.pen_clip3 = 5
    elif name in ("pacbio", "ont2d"):
        o.o_del = o.e_del = o.o_ins = o.e_ins = 1; o.b = 1; o.split_factor = 10.
        o.min_chain_weight = 20 if name == "ont2d" else 40
        o.min_seed_len = 14 if name == "ont2d" else 17
        o.pen_clip5 = o.pen_clip3 = 0
    fill_scmat(o)
    return o


def random_opt(rng, k):
    base = ("default", "default", "default", "intractg", "pacbio", "ont2d")[k % 6]
    o = preset(base)
    long_reads = base in ("pacbio", "ont2d")
    n_mod = int(rng.integers(2, 7)) if k >= 6 else 0          # the first six draws are the plain presets
    fields = rng.permutation(["gap", "clip", "w", "zdrop", "y", "r", "s", "c", "D", "W", "N", "G", "X", "k", "AB"])[:n_mod]
    for f in fields:
        if f == "gap":
            o.o_del, o.o_ins = int(rng.integers(1, 17)), int(rng.integers(1, 17))
            o.e_del, o.e_ins = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        elif f == "clip":
            o.pen_clip5, o.pen_clip3 = int(rng.integers(0, 13)), int(rng.integers(0, 13))
        elif f == "w":
            o.w = int(rng.choice([5, 20, 60, 100, 150]))
        elif f == "zdrop":
            o.zdrop = int(rng.choice([0, 20, 100, 200]))
        elif f == "y":
            o.max_mem_intv = int(rng.choice([0, 5, 20, 100]))
        elif f == "r":
            o.split_factor = float(rng.choice([0.8, 1.5, 3.0, 10.0]))
        elif f == "s":
            o.split_width = int(rng.choice([1, 10, 30]))
        elif f == "c":
            o.max_occ = int(rng.choice([5, 50, 500]))
        elif f == "D":
            o.drop_ratio = float(rng.choice([0.3, 0.5, 0.8]))
        elif f == "W":
            o.min_chain_weight = int(rng.choice([0, 10, 40]))
        elif f == "N":
            o.max_chain_extend = int(rng.choice([1, 3, 1 << 30]))
        elif f == "G":
            o.max_chain_gap = int(rng.choice([100, 1000, 10000]))
        elif f == "X":
            o.mask_level = float(rng.choice([0.3, 0.5, 0.9]))
        elif f == "k":
            o.min_seed_len = int(rng.integers(8, 29))
        elif f == "AB" and not long_reads:
            o.a = int(rng.integers(1, 4)); o.b = int(rng.integers(2, 10))
            for x in ("o_del", "o_ins", "e_del", "e_ins", "zdrop", "pen_clip5", "pen_clip3"):   # update_a (fastmap.c:125-139)
                setattr(o, x, getattr(o, x) * o.a)
    fill_scmat(o)
    return o, long_reads, base, [str(f) for f in fields]


def describe(o):
    return {n: (getattr(o, n) if n != "mat" else None) for n, _ in o._fields_ if n != "mat"}


def run_region_fuzz(dev, ref, g, draws, n_short, n_long, long_len, seed):
    rng = np.random.default_rng(seed)
    for k in range(draws):
        o, long_reads, base, fields = random_opt(rng, k)
        if long_reads:
            reads = simdata.make_reads_long(g, n_long, length=long_len, seed=seed * 1000 + k)
        else:
            length = int(rng.choice([100, 150, 250]))
            sub = float(rng.choice([0.01, 0.03]))
            reads = simdata.make_reads_se(g, n_short, length=length, seed=seed * 1000 + k, sub=sub, dele=0.002, ins=0.002)
        seqs, off = testdata.flat(reads)
        assert_regs_equal(*ref.align(o, seqs, off), *dev.align(o, seqs, off), f"draw {k} ({base} + {fields}): {describe(o)}")


# ---- command lines ---------------------------------------------------------------------------------------------------------------
CLI_SETS = [
    ["-x", "intractg"],
    ["-U", "9", "-m", "10"],
    ["-S"],
    ["-P"],
    ["-S", "-P", "-a"],
    ["-L", "3,8", "-O", "4,9", "-E", "2,1"],
    ["-y", "5", "-r", "3", "-c", "50", "-D", "0.3", "-W", "10"],
    ["-M", "-Y", "-T", "20", "-h", "2,8", "-z", "0.6"],
    ["-5", "-q", "-a"],
    ["-k", "14", "-w", "20", "-d", "30", "-N", "3", "-G", "500", "-s", "3", "-X", "0.7"],
    ["-A", "2", "-B", "5", "-U", "30"],
    ["-u", "-V", "-j"],
    ["-I", "350,60,800,50", "-m", "3"],
    ["-Q", "0"],
]
CLI_LONG = [["-x", "ont2d"], ["-x", "pacbio", "-k", "15"]]


def _body(sam: bytes) -> bytes:
    return b"\n".join(l for l in sam.split(b"\n") if not l.startswith(b"@PG"))


def _run(binary, args, env=None):
    p = subprocess.run([binary, "mem"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, f"{binary} {' '.join(args)}: {p.stderr.decode()[-800:]}"
    return _body(p.stdout)


def run_cli_fuzz(cli, fa, g, tmp_path, sets, long_sets, n_pairs, n_long, long_len, env=None):
    f1, f2 = str(tmp_path / "f1.fq"), str(tmp_path / "f2.fq")
    r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=931, sub=0.02)
    # every fourth mate is unrelated sequence so that mate rescue and the unpaired branches have work
    rng = np.random.default_rng(932)
    bad = np.arange(n_pairs) % 4 == 3
    r2[bad] = rng.integers(0, 4, size=(int(bad.sum()), r2.shape[1])).astype(np.uint8)
    simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
    K = ["-K", "100000000", "-t", "4"]
    for s in sets:
        assert _run(refapi.REF_BWA, K + s + [fa, f1, f2]) == _run(cli, K + s + [fa, f1, f2], env), f"paired-end {' '.join(s)}"
    if long_sets:
        fl = str(tmp_path / "long.fq")
        simdata.write_fastq(fl, simdata.make_reads_long(g, n_long, length=long_len, seed=933))
        for s in long_sets:
            assert _run(refapi.REF_BWA, K + s + [fa, fl]) == _run(cli, K + s + [fa, fl], env), f"long reads {' '.join(s)}"


# ---- CPU: mock runtime --------------------------------------------------------------------------------------------------------------
def test_sim_option_fuzz():
    import hostsim_build
    from bwa_amd.api import BwaGpu
    prefix, g = testdata.small_index()
    sim, ref = BwaGpu(prefix, lib_path=hostsim_build.build()), refapi.RefIndex(prefix)
    try:
        run_region_fuzz(sim, ref, g, draws=14, n_short=14, n_long=1, long_len=1200, seed=51)
    finally:
        sim.close(); ref.close()


def test_sim_option_fuzz_kernel_forms():
    """The same draws' kind through the other kernel forms.  First: seeding with one memory round trip per iteration for short reads as well
    (seed_mrg = 2, on LDS stacks of two entries so that nearly every backward step takes a prefetched entry) and pass 1 of the long reads by
    chunks of 128 bases instead of 256.  Second: the round-3 forms the long-read defaults replaced (lane-per-read seeding, one-lane interval
    sort, HBM-scratch seed re-scoring, one column per lane in the patch alignments)."""
    import hostsim_build
    from bwa_amd.api import BwaGpu
    prefix, g = testdata.small_index()
    for seed, options in ((53, {"seed_mrg": 2, "seed_lds_ent": 2, "seed_task_stack": 4}), (54, {"seed_mrg": 0, "seed_tasks": 0, "publish_blk": 0, "seedsw_lds": 0, "dedup_blk": 0})):
        sim, ref = BwaGpu(prefix, lib_path=hostsim_build.build(), options=options), refapi.RefIndex(prefix)
        try:
            run_region_fuzz(sim, ref, g, draws=6, n_short=10, n_long=1, long_len=1200, seed=seed)
        finally:
            sim.close(); ref.close()


def test_sim_cli_option_fuzz(tmp_path):
    import hostsim_build
    import test_cli
    prefix, g = testdata.small_index()
    env = dict(os.environ, BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6")
    run_cli_fuzz(test_cli._sim_cli(), prefix, g, tmp_path, [CLI_SETS[1], CLI_SETS[4], CLI_SETS[5], CLI_SETS[7]], [], 10, 0, 0, env)


# ---- GPU ------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_gpu_option_fuzz():
    from bwa_amd.api import BwaGpu
    fa, g = testdata.medium_index()
    gpu, ref = BwaGpu(fa), refapi.RefIndex(fa)
    try:
        run_region_fuzz(gpu, ref, g, draws=42, n_short=2000, n_long=200, long_len=1500, seed=52)
    finally:
        gpu.close(); ref.close()


@pytest.mark.gpu
def test_gpu_option_fuzz_kernel_forms():
    """On hardware, strictly: the kernel forms that are not the default for their class of batch -- one-trip seeding for short reads and small
    chunks for long ones; and the round-3 long-read forms (lane-per-read seeding, one-lane sort, HBM re-scoring, one column per lane)."""
    from bwa_amd.api import BwaGpu
    fa, g = testdata.medium_index()
    for seed, options in ((55, {"seed_mrg": 2, "seed_lds_ent": 2, "seed_task_stack": 4}), (56, {"seed_mrg": 0, "seed_tasks": 0, "publish_blk": 0, "seedsw_lds": 0, "dedup_blk": 0})):
        gpu, ref = BwaGpu(fa, options=options), refapi.RefIndex(fa)
        try:
            run_region_fuzz(gpu, ref, g, draws=10, n_short=1000, n_long=150, long_len=1500, seed=seed)
        finally:
            gpu.close(); ref.close()


@pytest.mark.gpu
def test_gpu_cli_option_fuzz(tmp_path):
    from bwa_amd import build as b
    _, cli = b.build_host(verbose=False)
    fa, g = testdata.medium_index()
    run_cli_fuzz(cli, fa, g, tmp_path, CLI_SETS, CLI_LONG, 3000, 60, 3000)
