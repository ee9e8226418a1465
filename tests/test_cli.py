"""The stand-alone command line `bwa-amd mem` (bwa_amd/csrc/host/main_mem.cpp) against the reference's `bwa mem`:
same FASTQ/FASTA input, same options, same -K -> the SAM must be identical except for the @PG line.
CPU variant: the CLI linked against the mock-runtime build of the device library (tests/hostsim), small inputs.
GPU variant (-m gpu): the real binary, larger inputs."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import refapi
import testdata
from bwa_amd import simdata

ROOT = testdata.ROOT


def _body(sam: bytes) -> bytes:
    return b"\n".join(l for l in sam.split(b"\n") if not l.startswith(b"@PG"))


def _run(binary, args, env=None):
    p = subprocess.run([binary, "mem"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    return _body(p.stdout)


def _sim_cli():
    if os.environ.get("BWA_AMD_SIM_CLI"):        # (a sanitizer build of the same sources: tools/sanitize_mock.sh)
        return os.environ["BWA_AMD_SIM_CLI"]
    import hostsim_build
    from bwa_amd import build as b
    hostsim_build.build()
    sim = os.path.join(ROOT, "tests", "hostsim")
    out = os.path.join(sim, "bwa-amd-sim")
    host = os.path.join(ROOT, "bwa_amd", "csrc", "host")
    srcs = [os.path.join(host, f) for f in sorted(os.listdir(host)) if f.endswith(".cpp")]
    deps = srcs + [os.path.join(host, f) for f in os.listdir(host) if f.endswith(".h")] + [os.path.join(sim, "libbwagpu_hostsim.so")]
    if not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(d) for d in deps):
        subprocess.run(["g++"] + b.HOST_FLAGS + srcs + ["-o", out, "-L" + sim, "-lbwagpu_hostsim", "-Wl,-rpath," + sim, "-lz", "-lpthread"], check=True)
    return out


def _write_inputs(tmp_path, g, n_pairs, seed):
    r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=seed)
    f1, f2 = str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq")
    simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
    # interleaved + gzip + comments + /1 /2 suffixes, plus a few unpaired records in between
    inter = str(tmp_path / "inter.fq.gz")
    with gzip.open(inter, "wb") as f:
        a = simdata._ASCII
        for i in range(n_pairs):
            for k, r in ((1, r1), (2, r2)):
                f.write(f"@p{i}/{k} BC:Z:ACGT{i}\n".encode() + a[r[i]].tobytes() + b"\n+\n" + b"F" * r.shape[1] + b"\n")
            if i % 7 == 3:
                f.write(f"@single{i} XX:i:{i}\n".encode() + a[r1[i][::-1] % 4].tobytes() + b"\n+\n" + b"5" * r1.shape[1] + b"\n")
    fasta = str(tmp_path / "reads.fa")
    with open(fasta, "wb") as f:
        for i in range(min(n_pairs, 40)):
            s = simdata._ASCII[r1[i]].tobytes()
            f.write(f">fa{i}\n".encode() + s[:70] + b"\n" + s[70:] + b"\n")
    # awkward but legal input: CRLF line ends, multi-line FASTQ records, lower case and IUPAC codes, blank lines, a record
    # without qualities in between, a comment after the name
    weird = str(tmp_path / "weird.fq")
    with open(weird, "wb") as f:
        a = simdata._ASCII
        for i in range(min(n_pairs, 12)):
            s = a[r1[i]].tobytes()
            q = bytes(33 + (j * 7 + i) % 40 for j in range(len(s)))
            if i % 4 == 0:
                f.write(f"@w{i} some comment\r\n".encode() + s + b"\r\n+\r\n" + q + b"\r\n")
            elif i % 4 == 1:
                f.write(f"@w{i}\n".encode() + s[:60] + b"\n" + s[60:] + b"\n+w\n" + q[:100] + b"\n" + q[100:] + b"\n\n")
            elif i % 4 == 2:
                f.write(f">w{i}\n".encode() + s.lower()[:75] + b"\n" + s[75:].replace(b"A", b"R", 1) + b"\n")
            else:   # (a '-' among the bases: nst_nt4_table gives it code 5, not 4, bntseq.c:46-63)
                f.write(f"@w{i}\n".encode() + s[:40] + b"-" + s[41:] + b"\n+\n" + q + b"\n")
    return f1, f2, inter, fasta


def _compare_all(cli, fa, f1, f2, inter, fasta, env=None, full=True):
    """full=False (the mock runtime, seconds per invocation): the option sets that only change host-side behaviour already covered
    by another run are left to the GPU test."""
    K = ["-K", "100000000", "-t", "4"]
    assert _run(refapi.REF_BWA, K + [fa, f1]) == _run(cli, K + [fa, f1], env), "single-end"
    assert _run(refapi.REF_BWA, K + [fa, f1, f2]) == _run(cli, K + [fa, f1, f2], env), "paired-end, two files"
    x = ["-p", "-C", "-R", "@RG\\tID:grp1\\tSM:s1", "-Y", "-M"]
    assert _run(refapi.REF_BWA, K + x + [fa, inter]) == _run(cli, K + x + [fa, inter], env), "smart pairing, comments, read group"
    y = ["-a", "-k", "17", "-A", "2", "-T", "40", "-h", "3,10", "-I", "400,50", "-5"]
    assert _run(refapi.REF_BWA, K + y + [fa, f1, f2]) == _run(cli, K + y + [fa, f1, f2], env), "scaled scores (-A 2), -I, -a, -5"
    if full:   # (the -H/-o run below reads the same file)
        assert _run(refapi.REF_BWA, K + [fa, fasta]) == _run(cli, K + [fa, fasta], env), "multi-line FASTA input"
    weird = os.path.join(os.path.dirname(f1), "weird.fq")
    assert _run(refapi.REF_BWA, K + ["-C", fa, weird]) == _run(cli, K + ["-C", fa, weird], env), "CRLF, multi-line FASTQ, lower case / IUPAC, mixed FASTA records"
    # the second file ends early: both programs stop at the last complete pair (bseq_read warns, bwa.c:96-99)
    short2 = os.path.join(os.path.dirname(f1), "r2_short.fq")
    with open(f2, "rb") as fi, open(short2, "wb") as fo:
        fo.write(b"".join(fi.readlines()[: 4 * 7]))
    if full:
        assert _run(refapi.REF_BWA, K + [fa, f1, short2]) == _run(cli, K + [fa, f1, short2], env), "second file shorter than the first"
    # a record whose quality string is shorter than its sequence (kseq_read returns -2, kseq.h:219): the record is dropped, the
    # batch ends there, and reading resumes after whatever the quality loop consumed -- in the middle of the file and at its end
    lines = open(f1, "rb").read().split(b"\n")
    for name, rec in (("trunc_mid.fq", 3), ("trunc_last.fq", len(lines) // 4 - 1))[: 2 if full else 1]:
        t = os.path.join(os.path.dirname(f1), name)
        ll = list(lines[: 4 * (rec + 1 if name == "trunc_last.fq" else len(lines) // 4)])
        ll[4 * rec + 3] = ll[4 * rec + 3][:60]
        with open(t, "wb") as f:
            f.write(b"\n".join(ll) + (b"" if name == "trunc_last.fq" else b"\n"))
        assert _run(refapi.REF_BWA, K + [fa, t]) == _run(cli, K + [fa, t], env), name
    # header lines from a file (-H), output to a file (-o)
    hdr = os.path.join(os.path.dirname(f1), "hdr.txt")
    with open(hdr, "w") as f:
        f.write("@CO\tfirst comment\nnot a header line\n@CO\tsecond\\tcomment\n")
    outs = []
    for binary, e in ((refapi.REF_BWA, None), (cli, env)):
        o = os.path.join(os.path.dirname(f1), "out.sam")
        p = subprocess.run([binary, "mem"] + K + ["-H", hdr, "-o", o, fa, fasta], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
        assert p.returncode == 0 and p.stdout == b"", p.stderr.decode()[-500:]
        outs.append(_body(open(o, "rb").read()))
    assert outs[0] == outs[1], "-H <file> and -o <file>"
    assert _run(refapi.REF_BWA, ["-K", "3000", "-t", "2", fa, f1, f2]) == _run(cli, ["-K", "3000", "-t", "2", fa, f1, f2], env), "many small batches (-K 3000)"


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim(tmp_path):
    prefix, g = testdata.small_index()
    # the reference binary wants <prefix>.bwt etc.; both programs take the same prefix
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 14, seed=401)
    # the mock HIP runtime keeps its lane/block state in globals: two device threads, but one device call at a time
    _compare_all(_sim_cli(), prefix, f1, f2, inter, fasta, env=dict(os.environ, BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6"), full=False)   # (small prefix tables: a million emulated lanes per handle otherwise)


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim_reader_buffer_ends(tmp_path):
    """The input stage parses out of a 1 MiB buffer that a read-ahead thread refills; the test files fit one buffer, so the paths that
    handle a record (a name, a CRLF, a multi-line sequence, a '+' line, a gzip member) straddling buffer ends only run on big inputs.
    With buffers of 13 and 100 bytes every record of the awkward inputs straddles several: same SAM as `bwa mem`."""
    prefix, g = testdata.small_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 8, seed=405)
    weird = os.path.join(os.path.dirname(f1), "weird.fq")
    cli = _sim_cli()
    K = ["-K", "100000000", "-t", "2"]
    runs = [(["-C", prefix, weird], "awkward FASTQ/FASTA mix"), (["-p", "-C", prefix, inter], "interleaved gzip"), ([prefix, fasta], "multi-line FASTA"), ([prefix, f1, f2], "two files")]
    for args, what in runs:
        want = _run(refapi.REF_BWA, K + args)
        for cap in ("13", "100"):
            env = dict(os.environ, BWAGPU_CLI_STREAMS="1", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6", BWAGPU_CLI_BUF=cap)
            assert want == _run(cli, K + args, env), f"{what}, {cap}-byte buffers"


@pytest.mark.gpu
def test_cli_gpu(tmp_path):
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    from bwa_amd import build as b
    _, cli = b.build_host(verbose=False)
    fa, g = testdata.medium_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 20000, seed=402)
    _compare_all(cli, fa, f1, f2, inter, fasta)


@pytest.mark.gpu
def test_cli_gpu_two_devices(tmp_path):
    """BWAGPU_DEVICES on hardware: when the box has at least two GPUs, `bwa-amd mem` with every batch split over devices
    0 and 1 (index copied device to device with hipMemcpyPeer, bwagpu_clone_to_device) must reproduce `bwa mem` -- paired-end (one
    mem_pestat over the gathered shards), single-end, and many small batches."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"this box has {n} GPU(s): the multi-device path of bwa-amd mem (BWAGPU_DEVICES, hipMemcpyPeer) needs two and was NOT exercised on hardware by this run")
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    from bwa_amd import build as b
    _, cli = b.build_host(verbose=False)
    fa, g = testdata.medium_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 20000, seed=405)
    env = dict(os.environ, BWAGPU_DEVICES="0,1")
    K = ["-K", "100000000", "-t", "4"]
    p = subprocess.run([cli, "mem"] + K + ["-v", "3", fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and b"index copied to 2 devices" in p.stderr, p.stderr.decode()[-800:]
    assert _run(refapi.REF_BWA, K + [fa, f1, f2]) == _body(p.stdout), "paired-end, 2 GPUs"
    assert _run(refapi.REF_BWA, K + [fa, f1]) == _run(cli, K + [fa, f1], env), "single-end, 2 GPUs"
    assert _run(refapi.REF_BWA, ["-K", "300000", "-t", "4", fa, f1, f2]) == _run(cli, ["-K", "300000", "-t", "4", fa, f1, f2], env), "20 batches, 2 GPUs"
    if n >= 3:
        env3 = dict(os.environ, BWAGPU_DEVICES=",".join(str(i) for i in range(min(n, 8))))
        assert _run(refapi.REF_BWA, K + [fa, f1, f2]) == _run(cli, K + [fa, f1, f2], env3), f"paired-end, {min(n, 8)} GPUs"


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim_two_devices(tmp_path):
    """Multi-GPU in the product: with BWAGPU_DEVICES=0,1 every batch is split into two contiguous ranges of whole
    pairs, each range runs on its own device (hot path, device CIGARs, mate-rescue alignments), regions are gathered on the host and
    ONE mem_pestat covers the whole batch (bwamem.c:1258) -- the SAM must be the reference's, hence the single-device SAM, for
    single-end, paired-end, smart-pairing and many-small-batches input.  Two devices of the mock runtime on the CPU."""
    prefix, g = testdata.small_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 18, seed=403)
    cli = _sim_cli()
    env = dict(os.environ, BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", MOCK_HIP_DEVICES="2", BWAGPU_DEVICES="0,1", BWAGPU_PTAB_M="6")
    K = ["-K", "100000000", "-t", "4"]
    p = subprocess.run([cli, "mem"] + K + ["-v", "3", prefix, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and b"index copied to 2 devices" in p.stderr, p.stderr.decode()[-800:]
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _body(p.stdout), "paired-end, 2 devices"
    assert _run(refapi.REF_BWA, K + [prefix, f1]) == _run(cli, K + [prefix, f1], env), "single-end, 2 devices"
    x = ["-p", "-C"]
    assert _run(refapi.REF_BWA, K + x + [prefix, inter]) == _run(cli, K + x + [prefix, inter], env), "smart pairing, 2 devices"
    assert _run(refapi.REF_BWA, ["-K", "9000", "-t", "2", prefix, f1, f2]) == _run(cli, ["-K", "9000", "-t", "2", prefix, f1, f2], env), "small batches, 2 devices"
    # gap-rich reads: many alignments of 7..64 CIGAR operations, whose records point into each device's operation array -- the
    # offsets of the second device's records must move with its part of the merged array
    noisy = str(tmp_path / "noisy.fq")
    simdata.write_fastq(noisy, simdata.make_reads_se(g, 16, seed=404, sub=0.02, dele=0.03, ins=0.03))
    assert _run(refapi.REF_BWA, K + [prefix, noisy]) == _run(cli, K + [prefix, noisy], env), "gap-rich reads, 2 devices"
    # the 32-byte block layout of the BWT is rebuilt on every device the index is copied to (here with small superblocks, so that the
    # superblock table is in play); and the same run on the reference-format blocks, fetched quad-cooperatively
    env_occ = dict(env, BWAGPU_OCC32="1", BWAGPU_OCC32_SB_SHIFT="10")
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], env_occ), "paired-end, 2 devices, 32-byte blocks with 2^10-base superblocks"
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], dict(env, BWAGPU_OCC32="0", BWAGPU_SEED_COOP="1")), "paired-end, 2 devices, 64-byte blocks fetched quad-cooperatively"
    env3 = dict(env, MOCK_HIP_DEVICES="3", BWAGPU_DEVICES="0,1,2")
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], env3), "paired-end, 3 devices"
