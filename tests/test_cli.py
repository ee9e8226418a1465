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
        subprocess.run(["g++"] + b.HOST_FLAGS + ["-DBWAGPU_CLI_TEST_HOOKS"] + srcs + ["-o", out, "-L" + sim, "-lbwagpu_hostsim", "-Wl,-rpath," + sim, "-lz", "-lpthread"], check=True)
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
def test_cli_gpu_block_parallel_input(tmp_path):
    """`bwa-amd mem` with the block-parallel input stage (BWAGPU_CLI_PARSE_THREADS=4, blocks of 64 KiB so that the 20 000-pair files are
    several hundred blocks each) on hardware: same SAM as `bwa mem`, one batch and many batches."""
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    from bwa_amd import build as b
    _, cli = b.build_host(verbose=False)
    fa, g = testdata.medium_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 20000, seed=406)
    env = dict(os.environ, BWAGPU_CLI_PARSE_THREADS="4", BWAGPU_CLI_PAR_BLOCK="65536")
    for K in (["-K", "100000000", "-t", "4"], ["-K", "300000", "-t", "4"]):
        assert _run(refapi.REF_BWA, K + [fa, f1, f2]) == _run(cli, K + [fa, f1, f2], env), f"paired-end, {K[1]} bases per batch"
    weird = os.path.join(os.path.dirname(f1), "weird.fq")
    assert _run(refapi.REF_BWA, ["-C", fa, weird]) == _run(cli, ["-C", fa, weird], env), "awkward file"


def _pair_id_run(cli, fa, g, tmp_path, n_pairs, K, n0, threads, env_extra, seed):
    """(SAM records of `cli` started at read number n0, the compiled reference's mem_process_seqs batch by batch with each batch's n_processed)"""
    from bwa_amd.structs import default_opt
    r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=seed, sub=0.02)
    f1, f2 = str(tmp_path / "p1.fq"), str(tmp_path / "p2.fq")
    simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
    args = ["-K", str(K), "-t", str(threads), fa, f1, f2]
    got = _run(cli, args, dict(os.environ, BWAGPU_CLI_N_PROCESSED0=str(n0), **env_extra))
    ref = refapi.RefIndex(fa)
    opt = default_opt(); opt.flag |= 2; opt.n_threads = threads
    L = r1.shape[1]
    per = -(-K // L); per += per & 1
    want = b""
    inter = np.empty((2 * r1.shape[0], L), dtype=np.uint8); inter[0::2], inter[1::2] = r1, r2
    for lo in range(0, inter.shape[0], per):
        rd = inter[lo:lo + per]
        names = [f"r{(lo + i) >> 1}" for i in range(rd.shape[0])]
        want += ref.process_seqs(opt, names, simdata._ASCII[rd].tobytes(), b"I" * (rd.shape[0] * L), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * L, n_processed=n0 + lo)
    ref.close()
    body = lambda t: b"".join(l for l in t.splitlines(True) if not l.startswith(b"@"))
    return body(got), want, args, body


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim_run_starting_at_a_read_number(tmp_path):
    """BWAGPU_CLI_N_PROCESSED0 (the number of the run's first read) reaches mem_pair / mem_matesw through every batch: four batches of the mock-runtime
    command line just below pair id 2^23 equal the reference's mem_process_seqs called with the same numbers."""
    prefix, g = testdata.small_index()
    got, want, _, _ = _pair_id_run(_sim_cli(), prefix, g, tmp_path, 40, 3000, (1 << 24) - 30, 2, dict(BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6"), seed=408)
    assert got == want


@pytest.mark.gpu
def test_cli_gpu_pair_ids_across_2_to_23(tmp_path):
    """`bwa-amd mem` deep inside a long run: the run's first read is given the number 2^24 - 9000 (BWAGPU_CLI_N_PROCESSED0), so that the pair ids cross 2^23 -- where
    mem_pair's `id << 8` wraps (bwamem_pair.c:208,248) -- inside the second of five batches, with device CIGARs and mate rescue on.  The reference
    side is its own mem_process_seqs, called batch by batch (the batches `-K 2000000` forms) with each batch's n_processed."""
    assert refapi.have_ref(), "oracle/_ref (the compiled reference) is missing on the GPU box"
    from bwa_amd import build as b
    _, cli = b.build_host(verbose=False)
    fa, g = testdata.medium_index()
    got, want, args, body = _pair_id_run(cli, fa, g, tmp_path, 30000, 2000000, (1 << 24) - 9000, 4, {}, seed=407)
    assert got == want, "bwa-amd mem with pair ids across 2^23 differs from mem_process_seqs"
    assert body(_run(cli, args)) != want, "the pair ids made no difference: the test does not reach the hash"


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
    assert _run(refapi.REF_BWA, ["-K", "300000", "-t", "4", fa, f1, f2]) == _run(cli, ["-K", "300000", "-t", "4", fa, f1, f2], dict(env, BWAGPU_CLI_MULTI="split")), "20 batches, each split over 2 GPUs"
    # whole batches handed to the devices in turn (the default at this -K: a device's share of a split batch would be far below what fills it)
    p = subprocess.run([cli, "mem", "-K", "300000", "-t", "4", "-v", "3", fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and b"whole batches go to the devices in turn" in p.stderr, p.stderr.decode()[-800:]
    assert _run(refapi.REF_BWA, ["-K", "300000", "-t", "4", fa, f1, f2]) == _body(p.stdout), "20 whole batches over 2 GPUs"
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
    assert _run(refapi.REF_BWA, ["-K", "9000", "-t", "2", prefix, f1, f2]) == _run(cli, ["-K", "9000", "-t", "2", prefix, f1, f2], dict(env, BWAGPU_CLI_MULTI="split")), "small batches, each split over 2 devices"
    # whole batches to the devices in turn (BWAGPU_CLI_MULTI=batch; the default when a device's share of a split batch is small): four device
    # threads -- two slots on each device -- take the batches as they come; the batch, hence mem_pestat, is what a single device sees
    p = subprocess.run([cli, "mem", "-K", "9000", "-t", "2", "-v", "3", prefix, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    assert p.returncode == 0 and b"whole batches go to the devices in turn (4 device threads)" in p.stderr, p.stderr.decode()[-800:]
    assert _run(refapi.REF_BWA, ["-K", "9000", "-t", "2", prefix, f1, f2]) == _body(p.stdout), "small whole batches over 2 devices"
    assert _run(refapi.REF_BWA, ["-K", "9000", "-t", "2"] + x + [prefix, inter]) == _run(cli, ["-K", "9000", "-t", "2"] + x + [prefix, inter], dict(env, BWAGPU_CLI_MULTI="batch")), "smart pairing, whole batches over 2 devices"
    # gap-rich reads: many alignments of 7..64 CIGAR operations, whose records point into each device's operation array -- the
    # offsets of the second device's records must move with its part of the merged array
    noisy = str(tmp_path / "noisy.fq")
    simdata.write_fastq(noisy, simdata.make_reads_se(g, 16, seed=404, sub=0.02, dele=0.03, ins=0.03))
    assert _run(refapi.REF_BWA, K + [prefix, noisy]) == _run(cli, K + [prefix, noisy], env), "gap-rich reads, 2 devices"
    # the 32-byte block layout of the BWT is rebuilt on every device the index is copied to (here with small superblocks, so that the
    # superblock table is in play); and the same run on the reference-format blocks
    env_occ = dict(env, BWAGPU_OCC32="1", BWAGPU_OCC32_SB_SHIFT="10")
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], env_occ), "paired-end, 2 devices, 32-byte blocks with 2^10-base superblocks"
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], dict(env, BWAGPU_OCC32="0")), "paired-end, 2 devices, reference-format 64-byte blocks"
    env3 = dict(env, MOCK_HIP_DEVICES="3", BWAGPU_DEVICES="0,1,2")
    assert _run(refapi.REF_BWA, K + [prefix, f1, f2]) == _run(cli, K + [prefix, f1, f2], env3), "paired-end, 3 devices"


def _parse_dump(cli, prefix, files, env, K="100000000"):
    p = subprocess.run([cli, "mem", "-K", K, prefix] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(env, BWAGPU_CLI_PARSE_ONLY="2"))
    assert p.returncode == 0, p.stderr.decode()[-1000:]
    return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@"))


def test_cli_block_parallel_input_delivers_the_same_records(tmp_path):
    """BWAGPU_CLI_PARSE_THREADS: plain FASTQ files are cut into blocks at record boundaries and parsed by a pool of threads; anything the
    fast path declines, or a cut that was not a boundary, sends the rest of the file through the streaming reader.  The records and the
    batch boundaries the input stage delivers (BWAGPU_CLI_PARSE_ONLY=2) must not depend on the number of threads or on where the cuts fall:
    clean files, files that turn awkward in the middle (CRLF, multi-line records, FASTA records, quality lines that start with '@',
    a truncated record, no newline at the end), two files in lockstep, an empty file, many small batches."""
    prefix, g = testdata.small_index()
    cli = _sim_cli()
    rng = np.random.default_rng(77)
    a = simdata._ASCII
    r1, r2 = simdata.make_reads_pe(g, 300, seed=409)

    def rec(name, bases, qual=None, comment=b""):
        q = qual if qual is not None else bytes(33 + int(x) for x in rng.integers(0, 41, size=len(bases)))
        return b"@" + name + (b" " + comment if comment else b"") + b"\n" + bases + b"\n+\n" + q + b"\n"

    files = {}
    clean1 = b"".join(rec(b"p%d/1" % i, a[r1[i]].tobytes(), comment=b"BC:Z:%d" % i if i % 3 == 0 else b"") for i in range(300))
    clean2 = b"".join(rec(b"p%d/2" % i, a[r2[i][: 100 + i % 50]].tobytes()) for i in range(300))      # (other lengths than file 1: the batch boundary depends on both)
    files["clean1"], files["clean2"] = clean1, clean2
    # quality strings that start with '@' and with '+', names that look like bases
    tricky = b"".join(rec(b"q%d" % i, a[r1[i]].tobytes(), qual=(b"@" if i % 2 else b"+") + b"I" * (r1.shape[1] - 1)) for i in range(120))
    files["tricky"] = tricky
    s = a[r1[5]].tobytes()
    awkward = [rec(b"w0", s, comment=b"a comment").replace(b"\n", b"\r\n"),
               b"@w1\n" + s[:60] + b"\n" + s[60:] + b"\n+w1\n" + b"5" * 100 + b"\n" + b"5" * (len(s) - 100) + b"\n\n",
               b">w2 fasta\n" + s.lower()[:75] + b"\n" + s[75:] + b"\n",
               rec(b"w3", s[:40] + b"-" + s[41:]),
               rec(b"w4", s[:40] + b" " + s[41:])]
    for k, piece in enumerate(awkward):      # 150 clean records, one awkward one, 100 more clean ones
        files[f"mid{k}"] = clean1[: len(clean1) // 2] + piece + b"".join(rec(b"t%d" % i, a[r2[i]].tobytes()) for i in range(100))
    files["no_final_newline"] = clean1[:-1]
    files["truncated_qual"] = clean1[: len(clean1) // 3] + b"@bad\n" + s + b"\n+\n" + b"I" * 60 + b"\n" + clean2[: len(clean2) // 3]
    files["empty"] = b""
    files["one"] = rec(b"only", s)
    paths = {}
    for k, v in files.items():
        paths[k] = str(tmp_path / (k + ".fq"))
        open(paths[k], "wb").write(v)
    base = dict(os.environ, BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6")
    cases = [([paths[k]], "100000000") for k in files] + [([paths["clean1"], paths["clean2"]], "100000000"), ([paths["clean1"], paths["clean2"]], "7000"),
             ([paths["mid1"], paths["clean2"]], "20000"), ([paths["clean1"], paths["truncated_qual"]], "100000000"), ([paths["tricky"]], "3000")]
    for fl, K in cases:
        want = _parse_dump(cli, prefix, fl, dict(base, BWAGPU_CLI_PARSE_THREADS="0"), K)
        if fl == [paths["clean1"]]:
            assert want.count(b"\n") >= 300 and b"p7\tBC:Z:" not in want and b"p6\tBC:Z:6\t" in want       # (the dump is what it should be: /1 trimmed, comments kept)
        for threads, blk in (("1", "4194304"), ("3", "700"), ("2", "1111"), ("3", "4096"), ("2", "50000")):
            got = _parse_dump(cli, prefix, fl, dict(base, BWAGPU_CLI_PARSE_THREADS=threads, BWAGPU_CLI_PAR_BLOCK=blk), K)
            assert got == want, f"{[os.path.basename(f) for f in fl]} -K {K}: {threads} parser threads, blocks of {blk} bytes"


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim_block_parallel_input(tmp_path):
    """The whole command line with the block-parallel input stage: same SAM as `bwa mem` (two files, many small batches; the awkward file)."""
    prefix, g = testdata.small_index()
    f1, f2, inter, fasta = _write_inputs(tmp_path, g, 10, seed=407)
    weird = os.path.join(os.path.dirname(f1), "weird.fq")
    cli = _sim_cli()
    env = dict(os.environ, BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6", BWAGPU_CLI_PARSE_THREADS="3", BWAGPU_CLI_PAR_BLOCK="1500")
    for args, what in ((["-K", "3000", "-t", "2", prefix, f1, f2], "two files, -K 3000"), (["-K", "100000000", "-t", "2", "-C", prefix, weird], "awkward file")):
        assert _run(refapi.REF_BWA, args) == _run(cli, args, env), what


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
def test_cli_hostsim_dashes_among_the_bases(tmp_path):
    """'-' is base code 5 (nst_nt4_table): it indexes the scoring matrix one past a row's end in the reference's query profile and its
    letter in the SAM record is a NUL.  Reads with one to three of them at random positions, single-end and paired: same SAM as `bwa mem`."""
    prefix, g = testdata.small_index()
    cli = _sim_cli()
    rng = np.random.default_rng(5)
    r1, r2 = simdata.make_reads_pe(g, 30, seed=77)
    files = []
    for k, r in ((1, r1), (2, r2)):
        files.append(str(tmp_path / f"d{k}.fq"))
        with open(files[-1], "wb") as f:
            for i in range(r.shape[0]):
                s = bytearray(simdata._ASCII[r[i]].tobytes())
                for _ in range(int(rng.integers(1, 4))):
                    s[int(rng.integers(0, len(s)))] = ord("-")
                f.write(b"@d%d/%d\n" % (i, k) + bytes(s) + b"\n+\n" + b"I" * len(s) + b"\n")
    env = dict(os.environ, BWAGPU_CLI_STREAMS="2", BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6")
    K = ["-K", "100000000", "-t", "2"]
    for args in ([files[0]], files):
        assert _run(refapi.REF_BWA, K + [prefix] + args) == _run(cli, K + [prefix] + args, env), f"{len(args)} file(s)"


def test_cli_input_file_that_shrinks_under_the_mapping(tmp_path):
    """Plain FASTQ files are mapped by the block-parallel input stage; a file truncated while it is being read raises SIGBUS in whichever parser
    thread touches the missing pages.  The command line must say so and leave with EX_IOERR instead of dying silently.  The test
    hook BWAGPU_CLI_TEST_SHRINK cuts the file in half right after it has been mapped."""
    prefix, g = testdata.small_index()
    cli = _sim_cli()
    a = simdata._ASCII
    rng = np.random.default_rng(5)
    fq = os.path.join(str(tmp_path), "big.fq")
    with open(fq, "wb") as f:
        for i in range(3000):
            seq = bytes(a[rng.integers(0, 4, 150)])
            f.write(b"@r%d\n" % i + seq + b"\n+\n" + b"I" * 150 + b"\n")
    assert os.path.getsize(fq) > 4 * 8192
    env = dict(os.environ, BWAGPU_CLI_PARSE_ONLY="1", BWAGPU_CLI_PARSE_THREADS="2", BWAGPU_CLI_PAR_BLOCK="65536", BWAGPU_CLI_TEST_SHRINK="1")
    p = subprocess.run([cli, "mem", prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=120)
    assert p.returncode == 74, (p.returncode, p.stderr.decode()[-500:])
    assert b"SIGBUS" in p.stderr and b"shrank" in p.stderr


def test_cli_sa_interval_leaves_room_for_the_batches(tmp_path):
    """`bwa-amd mem` expands the suffix array on the device to the SMALLEST interval that still leaves room for its handles' batch arenas
    (bwagpu_mem_info / bwagpu_batch_footprint: ADVICE r5 -- the full array fitted, the first batch then failed, on a device with less free memory
    than an idle MI355X).  The mock device's free memory is what MOCK_HIP_FREE_MB says; the SAM is the same whatever interval is taken."""
    cli = _sim_cli()
    prefix, g = testdata.small_index()
    r = simdata.make_reads_se(g, 12, seed=5)
    fq = str(tmp_path / "r.fq")
    simdata.write_fastq(fq, r)
    import ctypes as C
    from bwa_amd.api import BwaGpu
    import hostsim_build
    s = BwaGpu(prefix, lib_path=hostsim_build.build())
    s.set_taps(False)                    # (as the command line does: the tap copies are part of the footprint)
    s.L.bwagpu_batch_footprint.restype = C.c_int64
    s.L.bwagpu_batch_footprint.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int]
    K = 200000
    per_handle = int(s.L.bwagpu_batch_footprint(s.h, K // 100 + 1024, K + (1 << 20), 256))
    assert per_handle > 0
    assert int(s.L.bwagpu_batch_footprint(s.h, 0, 10, 10)) == -1
    fre, tot = C.c_uint64(), C.c_uint64()
    assert s.L.bwagpu_mem_info(s.h, C.byref(fre), C.byref(tot)) == 0 and 0 < fre.value <= tot.value
    seq_len = 2 * g.shape[0]
    s.close()
    need_mb = (per_handle * 3 * 1.15 + 2e9) / (1 << 20)
    outs = {}
    for label, free_mb, want in (("plenty", need_mb + 8.0 * seq_len / (1 << 20) + 64, "interval 1 "), ("tight", need_mb + 8.0 * seq_len / 2 / (1 << 20) + 0.7, "interval 2 "), ("none", need_mb * 0.5, "kept at the index's own interval")):
        p = subprocess.run([cli, "mem", "-v", "3", "-K", str(K), prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6", MOCK_HIP_FREE_MB=str(int(free_mb))))
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert want in p.stderr.decode(), (label, p.stderr.decode()[-1500:])
        outs[label] = _body(p.stdout)
    assert outs["plenty"] == outs["tight"] == outs["none"] and outs["plenty"].count(b"\n") >= 12
    # the limit a user can set does the same as a fuller device
    p = subprocess.run([cli, "mem", "-v", "3", "-K", str(K), prefix, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6", BWAGPU_CLI_HBM_LIMIT_MB=str(int(need_mb * 0.5))))
    assert p.returncode == 0 and "kept at the index's own interval" in p.stderr.decode()


def test_option_ranges_apply_to_defaults_and_environment(monkeypatch):
    """The range test of bwagpu_set_option also guards bwagpu_set_default_option (refused) and the environment (ignored with a warning): seed_lds_ent
    above 16 entries per lane would exceed a workgroup's dynamic LDS and fail the launch instead of the call."""
    import ctypes as C
    from bwa_amd.api import BwaGpu
    import hostsim_build
    prefix, _ = testdata.small_index()
    monkeypatch.setenv("BWAGPU_SEED_LDS_ENT", "40")
    monkeypatch.setenv("BWAGPU_EXT_OCC", "5")
    s = BwaGpu(prefix, lib_path=hostsim_build.build())
    assert s.get_option("seed_lds_ent") == -1 and s.get_option("ext_occ") == 6
    s.L.bwagpu_set_default_option.argtypes = [C.c_char_p, C.c_longlong]
    assert s.L.bwagpu_set_default_option(b"seed_lds_ent", 17) != 0
    assert s.L.bwagpu_set_default_option(b"seed_lds_ent", 16) == 0
    s.L.bwagpu_clear_default_options()
    with pytest.raises(Exception):
        s.set_option("seed_lds_ent", 17)
    s.close()


def _write_bgzf(path, data: bytes, rng, eof_marker=True, empty_block_at=None):
    """BGZF as bgzip writes it (SAM spec 4.1): gzip members of at most 64 KiB with the extra field BC = member length - 1; block sizes drawn at random so that
    records straddle blocks everywhere; optionally an empty block in the middle and the 28-byte end-of-file marker."""
    import struct
    import zlib

    def block(chunk: bytes) -> bytes:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    out, o, k = [], 0, 0
    while o < len(data):
        n = int(rng.integers(1, 65281)) if k % 3 else int(rng.integers(1, 400))
        out.append(block(data[o:o + n])); o += n; k += 1
        if empty_block_at is not None and k == empty_block_at:
            out.append(block(b""))
    if eof_marker:
        out.append(block(b""))
    with open(path, "wb") as f:
        f.write(b"".join(out))
    return len(out)


def test_cli_bgzf_input_is_inflated_in_parallel(tmp_path):
    """BGZF (bgzip) FASTQ files: the blocks are inflated by the input pool side by side (BgzfPipe, main_mem.cpp) and the parser must see exactly the byte
    stream gzread delivers (bwa.c:79-112 over kseq.h): the records of the BGZF files equal the records of the plain files, whatever the grouping of
    blocks and the parser's buffer size; the SAM equals `bwa mem`'s on the same .gz files; a plain gzip file keeps the single-stream reader; a damaged
    block is an error, not a silently shorter input."""
    import shutil
    cli = _sim_cli()
    prefix, g = testdata.small_index()
    rng = np.random.default_rng(9)
    r1, r2 = simdata.make_reads_pe(g, 3000, seed=410)
    f1, f2 = str(tmp_path / "b1.fq"), str(tmp_path / "b2.fq")
    simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
    n_blk = _write_bgzf(f1 + ".gz", open(f1, "rb").read(), rng, empty_block_at=3)
    _write_bgzf(f2 + ".gz", open(f2, "rb").read(), rng, eof_marker=False)
    assert n_blk > 20

    def dump(files, **env):
        p = subprocess.run([cli, "mem", "-v", "3", "-K", "90000", prefix] + files, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6", BWAGPU_CLI_PARSE_ONLY="2", **env))
        assert p.returncode == 0, p.stderr.decode()[-800:]
        return b"\n".join(l for l in p.stdout.split(b"\n") if not l.startswith(b"@"))      # (the SAM header comes first: @PG holds the file names)
    want = dump([f1, f2])
    assert want.count(b"\n") > 6000
    for env in ({}, {"BWAGPU_CLI_BGZF_GROUP": "1"}, {"BWAGPU_CLI_BUF": "4099"}, {"BWAGPU_CLI_PARSE_THREADS": "1"}, {"BWAGPU_CLI_NO_BGZF": "1"}, {"BWAGPU_CLI_PARSE_THREADS": "0"}):
        assert dump([f1 + ".gz", f2 + ".gz"], **env) == want, env
    # the same through the aligner, against `bwa mem` reading the same BGZF files with gzread
    if refapi.have_ref():
        s1, s2 = str(tmp_path / "s1.fq"), str(tmp_path / "s2.fq")
        simdata.write_fastq(s1, r1[:24], suffix="/1"); simdata.write_fastq(s2, r2[:24], suffix="/2")
        _write_bgzf(s1 + ".gz", open(s1, "rb").read(), rng); _write_bgzf(s2 + ".gz", open(s2, "rb").read(), rng)
        assert _run(refapi.REF_BWA, [prefix, s1 + ".gz", s2 + ".gz"]) == _run(cli, [prefix, s1 + ".gz", s2 + ".gz"], dict(os.environ, BWAGPU_CLI_SERIALIZE="1", BWAGPU_PTAB_M="6"))
    # a flipped byte inside a block's compressed data
    bad = str(tmp_path / "bad.fq.gz")
    raw = bytearray(open(f1 + ".gz", "rb").read())
    raw[len(raw) // 2] ^= 0x5a
    open(bad, "wb").write(bytes(raw))
    p = subprocess.run([cli, "mem", "-v", "3", prefix, bad], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, BWAGPU_CLI_PARSE_ONLY="1"))
    assert p.returncode != 0 and b"BGZF" in p.stderr, p.stderr.decode()[-400:]
