"""Whole-SAM parity through the drop-in boundary: regions computed by the HIP path (or, on CPU, by the product source
under the mock runtime) are handed to the *reference's own* finalize code (mem_mark_primary_se / mem_reg2sam /
mem_pestat / mem_sam_pe, bwamem.c:1217-1233 via oracle/_ref/libbwaref.so) and the SAM text must equal, byte for byte,
what the reference's mem_process_seqs produces on its own.  This is exactly what replacing the worker1 loop means."""
import numpy as np
import pytest

import refapi
import testdata
from bwa_amd import simdata
from bwa_amd.structs import default_opt

needs_ref = pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


def _sam_pair(ref, engine, opt, reads, n_processed=0):
    seqs, off = testdata.flat(reads)
    names = [f"r{i >> 1}" if (opt.flag & 2) else f"r{i}" for i in range(reads.shape[0])]
    quals = b"I" * seqs.shape[0]
    want = ref.process_seqs(opt, names, ASCII[seqs].tobytes(), quals, off, n_processed=n_processed)
    counts, regs = engine.align(opt, seqs, off)
    got = ref.regs2sam(opt, names, seqs.tobytes(), quals, off, counts, regs, n_processed=n_processed)
    return want, got


def _interleave(r1, r2):
    out = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8)
    out[0::2], out[1::2] = r1, r2
    return out


@needs_ref
def test_sam_identical_hostsim():
    import hostsim_build
    from bwa_amd.api import BwaGpu
    fa, g = testdata.medium_index()
    ref = refapi.RefIndex(fa)
    sim = BwaGpu(fa, lib_path=hostsim_build.build())
    want, got = _sam_pair(ref, sim, default_opt(), simdata.make_reads_se(g, 60, seed=51), n_processed=1000)
    assert got == want and got.count(b"\n") >= 60
    opt = default_opt(); opt.flag |= 0x2
    want, got = _sam_pair(ref, sim, opt, _interleave(*simdata.make_reads_pe(g, 40, seed=52)), n_processed=2000)
    assert got == want
    sim.close(); ref.close()


@needs_ref
@pytest.mark.gpu
def test_sam_identical_gpu():
    from bwa_amd.api import BwaGpu
    fa, g = testdata.medium_index()
    ref = refapi.RefIndex(fa)
    gpu = BwaGpu(fa)
    want, got = _sam_pair(ref, gpu, default_opt(), simdata.make_reads_se(g, 20000, seed=53, n_frac=0.001), n_processed=12345)
    assert got == want and got.count(b"\n") >= 20000
    opt = default_opt(); opt.flag |= 0x2
    want, got = _sam_pair(ref, gpu, opt, _interleave(*simdata.make_reads_pe(g, 10000, seed=54)), n_processed=4000)
    assert got == want
    gpu.close(); ref.close()


@needs_ref
@pytest.mark.gpu
def test_dropin_binary_sam_identical(tmp_path):
    """oracle/_ref/bwa_gpu = the unmodified reference program whose mem_process_seqs is redirected (ld --wrap) to
    integration/mem_process_seqs_gpu.c -> libbwagpu.so.  `bwa_gpu mem` must write the same SAM as `bwa mem`
    (all lines except @PG, whose CL: field names the binary)."""
    import os
    import subprocess
    bwa_gpu = os.path.join(refapi.REF_DIR, "bwa_gpu")
    if not os.path.exists(bwa_gpu):
        pytest.skip("oracle/_ref/bwa_gpu not built")
    fa, g = testdata.medium_index()
    r1, r2 = simdata.make_reads_pe(g, 30000, seed=61)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)

    def run(binary, args):
        p = subprocess.run([binary, "mem", "-t", "8", "-K", "100000000"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG"))

    assert run(refapi.REF_BWA, [fa, f1]) == run(bwa_gpu, [fa, f1])               # single-end
    assert run(refapi.REF_BWA, [fa, f1, f2]) == run(bwa_gpu, [fa, f1, f2])       # paired-end (mem_pestat + mem_sam_pe on the host)


def _example_lines(binary, fa, fq, env=None):
    import subprocess
    p = subprocess.run([binary, fa, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1200)
    assert p.returncode == 0, p.stderr[-500:]
    return p.stdout


def _example_inputs(tmp_path, g, n, seed):
    reads = simdata.make_reads_se(g, n, seed=seed, sub=0.02, n_frac=0.002)
    fq = str(tmp_path / "lite.fq")
    simdata.write_fastq(fq, reads)
    return fq


@needs_ref
def test_mem_align1_binding_hostsim(tmp_path):
    """mem_align1 (bwamem_extra.c:102-112) through the drop-in boundary: oracle/_ref/example_gpu is the reference's example.c
    (bwamem-lite: mem_align1 + mem_reg2aln per read, position / mapQ / CIGAR / NM printed) with mem_align1 redirected (ld --wrap) to
    integration/mem_process_seqs_gpu.c -> bwagpu_align_bseq of one read + the reference's own mem_mark_primary_se.  Here the device
    library is the mock-runtime build (found first through LD_LIBRARY_PATH); same lines as the stock example."""
    import os
    import hostsim_build
    ex, ex_gpu = os.path.join(refapi.REF_DIR, "example"), os.path.join(refapi.REF_DIR, "example_gpu")
    if not (os.path.exists(ex) and os.path.exists(ex_gpu)):
        pytest.skip("oracle/_ref/example[_gpu] not built (make -C oracle example)")
    fa, g = testdata.small_index()
    fq = _example_inputs(tmp_path, g, 12, seed=71)
    os.symlink(hostsim_build.build(), str(tmp_path / "libbwagpu.so"))
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    want, got = _example_lines(ex, fa, fq), _example_lines(ex_gpu, fa, fq, env=env)
    assert got == want and want.count(b"\n") >= 10


@needs_ref
@pytest.mark.gpu
def test_mem_align1_binding_gpu(tmp_path):
    """The same on the device: every read is a batch of one through bwagpu_align_bseq."""
    import os
    ex, ex_gpu = os.path.join(refapi.REF_DIR, "example"), os.path.join(refapi.REF_DIR, "example_gpu")
    if not (os.path.exists(ex) and os.path.exists(ex_gpu)):
        pytest.skip("oracle/_ref/example[_gpu] not built (make -C oracle example)")
    fa, g = testdata.medium_index()
    fq = _example_inputs(tmp_path, g, 300, seed=72)
    want, got = _example_lines(ex, fa, fq), _example_lines(ex_gpu, fa, fq)
    assert got == want and want.count(b"\n") >= 280
