#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X BWA-MEM hot path on BASELINE.json's metric: 2x150 bp paired-end reads against a
GRCh38-scale index, SAM bit-identical to `bwa mem`.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run with one
rank per GPU.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[2] layout, the configuration the metric is quoted on): per GPU and step one batch of
`--reads` (default 1 M = 500 k pairs, mates interleaved as mem_process_seqs takes them, bwamem.h:146-150) synthetic 2x150 bp
reads against a seeded synthetic stand-in for GRCh38 of `--genome-mbp` (default 3100 -> seq_len = 6.2e9 > 2^32; no genome
data exists offline).  The index is built on the GPU at start-up by our own builder (bwagpu_index_build, byte-identical to
`bwa index`, ~5 s for 3.1 Gbp).  One "step" = one pass of the hot path (mem_align1_core for every read: seed -> SA -> chain ->
extend -> dedup) over a batch that is already resident in HBM: `value` is the RESIDENT HOT PATH rate and excludes PCIe and the
host finalize; the end-to-end `bwa-amd mem` rates (FASTQ in -> SAM out) are reported beside it.  Reads shard across ranks with
no data-path collective ("weak" scaling: every rank aligns its own `--reads`); the index reaches ranks > 0 by RCCL broadcast.

Parity gate: the unmodified reference (`oracle/_ref/bwa mem`, also the CPU baseline) and the product command line
(`bwa-amd mem`) align the same samples with the same -K -- a single-end sample in one batch and a paired-end sample in TWELVE
batches (four per device handle: arenas, learnt sizes and packed buffers are re-used from batch to batch) -- and their
SAM must be byte-identical apart from @PG, otherwise the run exits non-zero.  With --gpus N > 1 (`python bench.py --gpus N` starts the N
ranks itself) rank 0 runs the gate on device 0 as ever, then the paired-end sample and the end-to-end run once more over all N devices
(BWAGPU_DEVICES) -- reported as parity.pe_all_devices / end_to_end_pe, not part of the exit code: that path has never met hardware.

Extra objects on the JSON line:
  roofline       the longest kernel's algorithmic bytes / its measured duration (HIP events on the library's stream)
  cpu_baseline   the reference on the paired-end sample (and the single-end one under "se"), on this box's host cores
  parity         result of the gate (se, pe, multibatch)
  end_to_end_pe  FASTQ -> SAM rate of `bwa-amd mem` on a few million pairs, with per-stage microseconds per read
  longread       BASELINE configs[4]: 10 kb reads with -x pacbio (hot-path rate, GCUPS, reference on a prefix, SAM parity on it)
"""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the best measured streaming copy


_T0 = time.time()


def log(*a):
    print(f"[{time.time() - _T0:7.1f}s]", *a, file=sys.stderr, flush=True)


def build_or_load_index(genome_mbp: float, cache: str, rank: int, barrier, lib_path=None):
    """Seeded synthetic genome + reference-format index files, built on the GPU by bwagpu_index_build and cached on the box.
    Rank 0 builds; the genome's base codes are shared with the other ranks through a memory-mapped file."""
    from bwa_amd import simdata
    total = int(genome_mbp * 1_000_000)
    prefix = os.path.join(cache, f"g{total}_s42")
    info = {}
    if rank == 0 and not (os.path.exists(prefix + ".sa") and os.path.exists(prefix + ".codes.npy")):
        from bwa_amd.index import build_index
        os.makedirs(cache, exist_ok=True)
        t = time.time()
        g, lens = simdata.make_genome_large(total, n_contigs=24, seed=42, threads=4)
        t_gen = time.time() - t
        t = time.time()
        info = build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], lib_path=lib_path)
        t_idx = time.time() - t
        np.save(prefix + ".codes.npy", g)
        del g
        log(f"[bench] {genome_mbp:g} Mbp genome generated in {t_gen:.1f}s; index built on the device in {info['build_ms'] / 1e3:.2f}s "
            f"({t_idx:.1f}s with packing, D2H and file output)")
    barrier()
    return prefix, np.load(prefix + ".codes.npy", mmap_mode="r"), info


def sam_body_digest(path: str):
    """sha256 over a SAM file without its @PG lines (the only lines allowed to differ: program name and command line)."""
    h = hashlib.sha256()
    n = 0
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b"@PG"):
                continue
            h.update(line)
            n += line[:1] != b"@"
    return h.hexdigest(), n


def run_reference(prefix: str, files, threads: int, out_sam: str, K: int = 100000000, extra=(), timeout: float = 240.0):
    """Unmodified reference `bwa mem -t threads -K <K>`; reads/s from its own per-batch timing lines
    (bwamem.c:1263: '[M::mem_process_seqs] Processed N reads in X CPU sec, Y real sec') and the whole-run real time (main.c:126)."""
    bwa = os.path.join(ROOT, "oracle", "_ref", "bwa")
    t = time.time()
    log(f"[bench] reference: bwa mem -t {threads} -K {K} {' '.join(extra)} on {[os.path.basename(f) for f in files]}")
    try:
        with open(out_sam, "wb") as fo:
            p = subprocess.run([bwa, "mem", "-t", str(threads), "-K", str(K), "-v", "3"] + list(extra) + [prefix] + list(files), stdout=fo, stderr=subprocess.PIPE, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        log(f"[bench] reference bwa mem did not finish within {timeout:.0f} s")
        return None
    wall = time.time() - t
    n = tot = 0.0
    for m in re.finditer(r"Processed (\d+) reads in ([\d.]+) CPU sec, ([\d.]+) real sec", p.stderr):
        n += int(m.group(1)); tot += float(m.group(3))
    if p.returncode != 0 or tot <= 0:
        log("[bench] reference bwa mem failed:", p.stderr[-500:])
        return None
    return {"reads_per_s": n / tot, "n": int(n), "wall_s": wall}


def run_instrumented(prefix: str, files, threads: int):
    """The counter-instrumented reference build (oracle/_ref/bwa_instr, oracle/make_instr.py): SURVEY.md 8(d)'s N_blk, N_lf, N_sa,
    W_ref and DP cells, counted by the reference's own bwt_2occ4 / bwt_sa / mem_chain2aln / ksw_* on the same reads."""
    exe = os.path.join(ROOT, "oracle", "_ref", "bwa_instr")
    if not os.path.exists(exe):
        return None
    log("[bench] instrumented reference (work counters)")
    try:
        p = subprocess.run([exe, "mem", "-t", str(threads), "-K", "100000000", "-v", "3", prefix] + list(files), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=120)
    except subprocess.TimeoutExpired:
        return None
    m = re.search(r"\[orc_instr\] n_2occ4 (\d+) N_blk (\d+) N_sa (\d+) N_lf (\d+) W_ref (\d+) ext_calls (\d+) ext_cells (\d+) glb_cells (\d+)", p.stderr)
    n = sum(int(x) for x in re.findall(r"Processed (\d+) reads", p.stderr))
    if p.returncode != 0 or not m or n == 0:
        return None
    k = [int(x) for x in m.groups()]
    return {"n_reads": n, "n_2occ4": k[0] / n, "N_blk": k[1] / n, "N_sa": k[2] / n, "N_lf": k[3] / n, "W_ref": k[4] / n, "ext_calls": k[5] / n, "ext_cells": k[6] / n, "glb_cells": k[7] / n}


def run_product(prefix: str, files, threads: int, out_sam: str | None, streams: int | None = None, K: int = 100000000, extra=(), devices=None, timeout: float = 240.0, out_from_batch: int = 0):
    """The stand-alone `bwa-amd mem` (FASTQ in -> device hot path + device CIGARs / mate rescue -> host finalize -> SAM text).
    devices: device ids for BWAGPU_DEVICES (every batch is split over them), None = device 0."""
    cli = os.path.join(ROOT, "bwa_amd", "bwa-amd")
    cmd = [cli, "mem", "-t", str(threads), "-K", str(K), "-v", "3"] + list(extra) + (["-o", out_sam] if out_sam else []) + [prefix] + list(files)
    env = dict(os.environ, BWAGPU_CLI_TRACE="1")      # (per-batch timings of the device stage on stderr, averaged below)
    if streams:
        env["BWAGPU_CLI_STREAMS"] = str(streams)
    if out_from_batch > 0:
        env["BWAGPU_CLI_OUT_FROM_BATCH"] = str(out_from_batch)
    if devices and len(devices) > 1:
        env["BWAGPU_DEVICES"] = ",".join(str(d) for d in devices)
    t = time.time()
    log(f"[bench] product: bwa-amd mem -t {threads} -K {K} {' '.join(extra)} on {[os.path.basename(f) for f in files]}" + (f" devices {devices}" if devices and len(devices) > 1 else ""))
    try:
        p = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env, timeout=timeout)
    except subprocess.TimeoutExpired:
        log(f"[bench] bwa-amd mem did not finish within {timeout:.0f} s")
        return None
    wall = time.time() - t
    m = re.search(r"\[M::main_mem\] (\d+) reads in ([\d.]+) sec .*: (\d+) reads/s", p.stderr)
    if p.returncode != 0 or not m:
        log("[bench] bwa-amd mem failed:", p.stderr[-800:])
        return None
    busy = re.search(r"stage busy time: (.*)", p.stderr)
    n_reads = int(m.group(1))
    stage_us = {}
    if busy and n_reads:     # busy seconds of each pipeline stage -> microseconds per read (a stage's share of one host core, or of the device)
        for name, sec in re.findall(r"([a-z+]+) ([\d.]+) s", busy.group(1)):
            stage_us[name] = round(float(sec) / n_reads * 1e6, 3)
    retries = [int(x) for x in re.findall(r"\(retries (\d+)", p.stderr)]
    cpu = re.search(r"stage CPU time: total ([\d.]+) s = ([\d.]+) us per read; read ([\d.]+) s, encode ([\d.]+) s, device threads ([\d.]+) s, finalize\+pestat pools ([\d.]+) s, write ([\d.]+) s; one process on (\d+) threads tops out near ([\d.]+) Mreads/s .* single reader thread near ([\d.]+)", p.stderr)
    cpu_us = {}
    if cpu and n_reads:
        names = ("total", "read", "encode", "device_threads", "finalize_pestat_pools", "write")
        vals = (cpu.group(1),) + cpu.groups()[2:7]
        cpu_us = {k_: round(float(v_) / n_reads * 1e6, 4) for k_, v_ in zip(names, vals)}
        cpu_us["one_process_ceiling_Mreads_s"] = float(cpu.group(9)); cpu_us["single_reader_ceiling_Mreads_s"] = float(cpu.group(10)); cpu_us["threads"] = int(cpu.group(8))
    hm = re.search(r"over (\d+) handles", p.stderr)
    dev = re.findall(r"\[D::device_sub\] (\d+) reads .*: upload ([\d.]+) run ([\d.]+) download\+cigars ([\d.]+) pestat\+matesw ([\d.]+) s \(pestat ([\d.]+), (\d+) mate", p.stderr)
    dev_ms = {}
    if dev:
        full = max(int(d[0]) for d in dev)
        rows = [d for d in dev if int(d[0]) >= 0.9 * full] or dev       # (full-size batches: the last one is short)
        for k_, name in enumerate(("upload", "hot_path", "download_cigars", "pestat_matesw", "pestat")):
            dev_ms[name] = round(sum(float(d[k_ + 1]) for d in rows) / len(rows) * 1e3, 1)
        dev_ms["matesw_alignments"] = int(sum(int(d[6]) for d in rows) / len(rows))
        dev_ms["reads_per_batch"] = full
    after = re.findall(r"after the hot path, ms: pack ([\d.]+) download copy ([\d.]+) cigar kernels ([\d.]+) cigar copies ([\d.]+)", p.stderr)
    if after and dev_ms:      # download_cigars split by the library's own HIP events (kernels vs copies), averaged over the batches
        for k_, name in enumerate(("pack_kernel", "download_copy", "cigar_kernels", "cigar_copies")):
            dev_ms[name] = round(sum(float(a_[k_]) for a_ in after) / len(after), 1)
    steady = steady_state_from_trace(p.stderr, dev_ms.get("reads_per_batch"))
    return {"reads_per_s": float(m.group(3)), "n": n_reads, "wall_s": wall, "stages": busy.group(1) if busy else "", "stage_us_per_read": stage_us, "handles": int(hm.group(1)) if hm else None, "device_stage_ms_per_batch": dev_ms, "steady_state": steady,
            "n_batches": len(re.findall(r"\[M::process\] read \d+ sequences", p.stderr)), "retries": sum(retries) if retries else 0, "cpu_us_per_read": cpu_us}


def steady_state_from_trace(stderr: str, reads_per_batch):
    """`bwa-amd mem`'s timeline trace (BWAGPU_CLI_TRACE: `[D::timeline] batch N device a .. b`) -> the rate at which full-size batches leave the device
    stage between the pipeline's fill and its drain; None when the run is too short to have such a part."""
    ends = sorted(float(x) for x in re.findall(r"\[D::timeline\] batch \d+ device [\d.]+ \.\. ([\d.]+)", stderr))
    if len(ends) < 8 or not reads_per_batch:
        return None
    body = ends[3:-1]          # (after the handles' first batches; the last batch is short)
    gap = (body[-1] - body[0]) / (len(body) - 1)
    if gap <= 0:
        return None
    return {"Mreads_s": round(reads_per_batch / gap / 1e6, 3), "ms_per_batch": round(gap * 1e3, 1), "first_batch_out_s": round(ends[0], 3),
            "what": "full-size batches leaving the device stage between the pipeline's fill and its drain (time between the 4th and the last full batch / batches): the rate a longer input approaches; `value` is the whole run including fill and drain"}


def measure_traffic(prefix: str, batch_file: str, dense_sa: int, layout: str, cache: str, limit_s: float = 45.0):
    """HBM-side traffic of every hot-path kernel, measured IN THIS RUN: two child processes under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` /
    `--pmc WRITE_SIZE` (separate passes: the TCC block cannot count both at once) over one solo batch of the headline's reads (tools/pmc_child.py), each under
    `timeout -s KILL` (rocprofv3 7.2 can hang in its own finalisation).  Returns (tools/pmc_summary.py's per-kernel dict, seconds) or (None, reason)."""
    import shutil
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary
    t = time.time()
    dirs = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(cache, "pmc_" + ctr.lower())
        shutil.rmtree(d, ignore_errors=True)
        cmd = ["timeout", "-s", "KILL", str(int(limit_s)), prof, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "c", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_child.py"), "--prefix", prefix, "--batch", batch_file, "--dense-sa", str(dense_sa), "--layout", layout]
        try:
            p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=limit_s + 15, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        except subprocess.TimeoutExpired:
            return None, f"{ctr} pass did not end within {limit_s + 15:.0f} s"
        if p.returncode != 0 or "pmc_child done" not in p.stdout:
            return None, f"{ctr} pass failed (rc {p.returncode}): {p.stdout[-200:]}"
        dirs[ctr] = d
    try:
        res = pmc_summary.summarize("bench.py in-run passes", dirs["FETCH_SIZE"], dirs["WRITE_SIZE"])
    except Exception as e:
        return None, "summary failed: " + repr(e)
    for d in dirs.values():
        shutil.rmtree(d, ignore_errors=True)
    return res, round(time.time() - t, 1)


def kernel_traffic(pj, k, alg_bytes):
    """Counter traffic of kernel k (raw: FETCH_SIZE counts 64 bytes per request whatever its size) and how many times its algorithmic bytes that is."""
    if not pj:
        return {}
    names = [k] + (["k_seed3"] if k == "k_seed" else [])
    f = sum(pj.get(n, {}).get("FETCH_SIZE_KB", 0.0) for n in names) * 1024.0
    w = sum(pj.get(n, {}).get("WRITE_SIZE_KB", 0.0) for n in names) * 1024.0
    if f + w <= 0:
        return {}
    return {"traffic_raw_GB": round((f + w) / 1e9, 3), "fetch_raw_GB": round(f / 1e9, 3), "write_GB": round(w / 1e9, 3), "wasted_traffic": round((f + w) / alg_bytes, 2) if alg_bytes > 0 else None}


def effective_cpus() -> int:
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU boxes expose 256 hardware threads but
    cap the container at 16 CPUs), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def interleave(r1: np.ndarray, r2: np.ndarray) -> np.ndarray:
    out = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8)
    out[0::2] = r1; out[1::2] = r2
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step (pairs x 2 when --layout pe)")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--layout", choices=("pe", "se"), default="pe", help="pe: BASELINE configs[2] (2x150 bp pairs, the metric's layout); se: configs[1]")
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--cache", default=os.environ.get("BWA_AMD_CACHE", "/tmp/bwa_amd_bench"))
    ap.add_argument("--streams", type=int, default=3, help="batches in flight per GPU (handles sharing the index)")
    ap.add_argument("--dense-sa", type=int, default=1, help="densify the SA on the device to this interval (0 = keep the reference's 32)")
    ap.add_argument("--cpu-sample", type=int, default=200_000, help="reads of the single-end parity / CPU-baseline sample; the paired-end one has this many reads too")
    ap.add_argument("--e2e-reads", type=int, default=20_000_000, help="reads of the end-to-end runs: 10 M pairs, BASELINE configs[2] at its stated size (pipeline fill and drain cost ~0.5 s whatever the length)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the reference runs (and with them the parity gate)")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 counter passes (roofline.traffic then comes from profiles/pmc_latest.json and says so)")
    ap.add_argument("--long-extra", default="10000,20000", help="further long-read batch sizes whose hot-path rate is reported (10 000 reads of 10 kb = the reference's -K of 100 Mbase); '' = none")
    ap.add_argument("--tail-batches", type=int, default=3, help="full-size batches at the END of the end-to-end run whose SAM is compared with the reference's (arena re-use at depth); 0 = skip")
    ap.add_argument("--parity-pairs", type=int, default=800_000, help="pairs of the multi-batch paired-end parity / CPU-baseline sample (twelve batches)")
    ap.add_argument("--no-longread", action="store_true", help="skip the BASELINE configs[4] leg (10 kb reads, -x pacbio)")
    ap.add_argument("--long-reads", type=int, default=6000)
    ap.add_argument("--long-len", type=int, default=10000)
    ap.add_argument("--long-steps", type=int, default=2)
    ap.add_argument("--long-sample", type=int, default=2000, help="reads of the long-read CPU-baseline / parity prefix (the reference does ~200 reads/s on 16 threads: ~10 s); "
                    "the product aligns them in >= 9 batches, three per device handle")
    ap.add_argument("--e2e-handles", type=int, default=0, help="a second FASTQ->SAM run of the paired-end sample with this many batches in flight (BWAGPU_CLI_STREAMS; the default run uses 3); 0 = skip (the default since round 6: 4 and 5 handles measured 5.1-6.0 and 4.7 against 5.9-6.2 Mreads/s with 3, gpurun_out/s5)")
    ap.add_argument("--variants", default="ext_pack=1;seed_mrg=0;ext_occ=4;chain_regs=0", help="';'-separated library option settings (bwagpu_set_option names) to A/B against the defaults in a child process (tools/variant_probe.py); '' = none")
    ap.add_argument("--timed-sample", type=int, default=20000, help="reads of timed batch 0 whose regions are compared with the compiled reference's mem_align1_core (parity.timed_batch); 0 = skip")
    ap.add_argument("--instr-pairs", type=int, default=30000, help="pairs the counter-instrumented reference runs on (b_alg_per_read)")
    ap.add_argument("--variants-timeout", type=float, default=70.0, help="seconds for the short-read child process (the long-read one gets 0.8 of it)")
    ap.add_argument("--wall-budget", type=float, default=330.0, help="seconds since start after which no variants child may still run: the two legs share what is left "
                    "when the line's own measurements are done (they end first; a leg that gets less than 20 s is skipped)")
    args = ap.parse_args()

    # --gpus N > 1 means N ranks, one per GPU.  Launched bare (`python bench.py --gpus N`, no WORLD_SIZE in the environment) the script
    # re-executes itself under torch.distributed.run with N ranks on this node; launched by the driver's own torch.distributed.run line it finds
    # WORLD_SIZE == N.  Anything else -- a world size that is not --gpus -- is an error: a line that says n_gpus N must have run on N ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        log(f"[bench] --gpus {args.gpus} without WORLD_SIZE: re-launching as {' '.join(cmd[1:8])} ...")
        sys.stdout.flush(); sys.stderr.flush()
        os.execv(sys.executable, cmd)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: launch `python bench.py --gpus N` bare, or under torch.distributed.run --nproc-per-node N")
    # test hooks (tests/test_bench_ranks.py runs this very entry on a CPU box): the collective backend and the library that stands for the device
    backend = os.environ.get("BWA_AMD_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm; "gloo" only with the mock-runtime library below
    lib_path = os.environ.get("BWA_AMD_BENCH_LIB") or None             # default: bwa_amd/csrc/libbwagpu.so
    on_gpu = backend == "nccl"
    if not on_gpu and not lib_path:
        sys.exit("bench.py: a CPU collective backend needs BWA_AMD_BENCH_LIB (the mock-runtime build); the product has no CPU path")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if on_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend)
    dev_id = local if on_gpu else 0

    def barrier():
        if dist is not None:
            dist.barrier()
        if on_gpu:
            torch.cuda.synchronize()

    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt

    t_all = time.time()
    prefix, g, idx_info = build_or_load_index(args.genome_mbp, args.cache, rank, barrier, lib_path)
    bcast_s = None
    if dist is not None:                     # rank 0 loads + uploads, the others receive the index over RCCL/xGMI
        from bwa_amd import dist as bdist
        t_b = time.perf_counter()
        gpu = bdist.broadcast_index(prefix, device=dev_id, src=0, lib_path=lib_path)
        barrier()
        bcast_s = time.perf_counter() - t_b
        if rank == 0:
            log(f"[bench] index loaded on rank 0 and broadcast to {world} ranks over RCCL in {bcast_s:.2f}s")
    else:
        gpu = BwaGpu(prefix, device=dev_id, lib_path=lib_path)   # index resident in this GPU's HBM (no CPU fallback: raises without a GPU)
    if args.dense_sa:
        gpu.densify_sa(args.dense_sa)
    gpu.set_taps(False)
    opt = default_opt()
    pe = args.layout == "pe"
    if pe:
        opt.flag |= 0x2                      # MEM_F_PE (bwamem.h:43): mates interleaved; the hot path aligns them independently (bwamem.c:1209-1213)

    def make_batch(seed):
        if pe:
            r1, r2 = simdata.make_reads_pe(g, args.reads // 2, length=args.read_len, seed=seed)
            return interleave(r1, r2)
        return simdata.make_reads_se(g, args.reads, length=args.read_len, seed=seed)

    # S batches in flight per GPU: S handles share the resident index (bwagpu_clone), each with its own stream and arenas and
    # driven by its own host thread, so the latency-bound tails of one batch overlap the throughput-bound kernels of another.
    import threading
    S = max(1, args.streams)
    handles = [gpu] + [gpu.clone() for _ in range(S - 1)]
    batches = []
    for si, hdl in enumerate(handles):
        rd = make_batch(1000 + rank * 64 + si)            # this rank's shard(s)
        hdl.set_taps(False)
        hdl.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * args.read_len)   # resident in HBM
        batches.append(rd)
    n_batch = batches[0].shape[0]
    variant_files = []
    if world == 1 and ((args.variants.strip() and not args.no_cpu_baseline) or not args.no_pmc):       # the variants leg and the counter passes (child processes at the end) run on these very batches
        for si, rd in enumerate(batches):
            if si == 0 or (args.variants.strip() and not args.no_cpu_baseline):
                variant_files.append(os.path.join(os.path.dirname(prefix), f"variant_batch{si}.npy"))
                np.save(variant_files[-1], rd)

    log(f"[bench] index resident, {S} batches of {n_batch} reads uploaded")
    # one untimed instrumented solo pass: algorithmic work counters of batch 0 (roofline numerator) and solo kernel times
    # (the algorithm's own work: with the seeding kernel's iteration budget off no read is given up half-way and seeded again by the task kernels;
    # the product's counters, rework included, are reported beside it)
    gpu.set_option("share", 100)      # (the solo passes: every kernel with the whole chip, whatever the handle count makes the default -- bwagpu_config.h)
    gpu.set_stats(True)
    gpu.set_option("seed_budget", 0)
    gpu.run(opt)
    work = gpu.stats()
    gpu.set_option("seed_budget", -1)
    gpu.run(opt)
    work_product = gpu.stats()
    try:        # k_seed's stats instance counts the lane steps that fetch an interval-stack entry back from its HBM spill area (bwagpu_debug_prof[10])
        import ctypes as C_
        prof_ = (C_.c_ulonglong * 16)()
        gpu.L.bwagpu_debug_prof.argtypes = [C_.c_void_p, C_.c_void_p]
        if gpu.L.bwagpu_debug_prof(gpu.h, prof_) == 0:
            work_product["n_stack_spill"] = int(prof_[10])
    except Exception:
        pass
    gpu.set_stats(False)
    gpu.run(opt)
    solo = gpu.stats()
    stage_keys = ("ms_seed", "ms_publish", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup", "ms_total")
    stage_ms = {k: solo[k] for k in stage_keys}
    gpu.set_option("share", -1)

    def worker(hdl, n_pass):
        for _ in range(n_pass):
            hdl.run(opt)                     # blocks until this batch's kernels have finished (stream sync inside)

    share = [args.steps // S + (1 if i < args.steps % S else 0) for i in range(S)]
    for _ in range(args.warmup):
        th = [threading.Thread(target=worker, args=(h_, 1)) for h_ in handles]
        [t.start() for t in th]; [t.join() for t in th]
    barrier()
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(handles[i], share[i])) for i in range(S) if share[i]]
    [t.start() for t in th]; [t.join() for t in th]
    barrier()
    dt = time.perf_counter() - t0
    ranks_seen, per_rank = 1, None
    if dist is not None:
        tdev = "cuda" if on_gpu else "cpu"
        dt_own = dt
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, dtype=torch.int64, device=tdev)            # the ranks the collective library itself reports: a sum of ones
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones.item())
        mine = torch.tensor([batches[0].shape[0] * args.steps / dt_own / 1e6], dtype=torch.float64, device=tdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [round(float(x.item()), 4) for x in allr]
        if ranks_seen != world or dist.get_world_size() != world:
            sys.exit(f"bench.py: the collective reports {ranks_seen} ranks (world size {dist.get_world_size()}), expected {world}")
    counts, regs = gpu.download()
    digest = hashlib.sha256(counts.tobytes() + regs.tobytes()).hexdigest()[:16]
    cigar_stage = None
    try:       # the CIGAR stage of the same batch, alone on the chip (bwagpu_batch_cigars with the command line's XA-aware filter): kernels' time, DP cells
        import ctypes as C_
        gpu.L.bwagpu_set_cigar_filter.argtypes = [C_.c_void_p, C_.c_int]
        gpu.L.bwagpu_set_cigar_filter(gpu.h, 1)
        gpu.set_stats(True); cg = gpu.cigars(opt); st_c = gpu.stats(); gpu.set_stats(False)
        ms_c = []
        for _ in range(2):
            cg = gpu.cigars(opt); ms_c.append(gpu.stats()["ms_cigar_kernels"])
        gpu.L.bwagpu_set_cigar_filter(gpu.h, 0)
        cigar_stage = {"kernels_ms": round(min(ms_c), 3), "regions": int(regs.shape[0]), "served": int((cg["n_cigar"] > 0).sum()), "filtered": int(((cg["n_cigar"] < 0) & (cg["score"] == 1)).sum()),
                       "left_to_host": int(((cg["n_cigar"] < 0) & (cg["score"] != 1)).sum()), "dp_fills": int(st_c["n_cig_dp"]), "cigar_cells_per_read": round(st_c["n_cig_cells"] / max(1, n_batch), 1),
                       "gcups": round(st_c["n_cig_cells"] / (min(ms_c) * 1e-3) / 1e9, 1) if min(ms_c) > 0 else None,
                       "what": "bwagpu_batch_cigars on timed batch 0 alone (k_cigar's two tiers + k_cigar_long, NM/MD included), with `bwa-amd mem`'s filter: regions that can appear neither as a record nor in an XA tag are skipped"}
        del cg
    except Exception as e:
        cigar_stage = {"error": repr(e)}
    for hdl in handles[1:]:
        hdl.close()
    if dist is not None:
        # everything below is host-side work of rank 0 (reference runs, the command-line product over all N devices): the other ranks
        # release their GPUs and leave now, so that no collective is left waiting for minutes
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            gpu.close()
            sys.exit(0)

    rc_exit = 0
    log(f"[bench] hot path timed: {dt / args.steps * 1e3:.1f} ms/step; solo stages {stage_ms}")
    total_reads = n_batch * world * args.steps
    value = total_reads / dt / 1e6
    nr = float(work["n_reads"])
    occ32 = os.environ.get("BWAGPU_OCC32", "1") != "0"
    blk_bytes = 32.0 if occ32 else 64.0
    # algorithmic bytes per launch: one index block per Occ lookup / LF step (64 bytes in the reference's layout; the
    # device's default layout answers the same query from a 32-byte block), 16 bytes per prefix-table entry, 8 bytes per SA sample, the
    # read bases; chaining: the slot records it reads (20 B) and the 160-byte B-tree nodes and 64-byte chain records it touches (counted
    # by the instrumented pass); extension: packed reference window + read + regions
    alg = {
        "k_seed": 64.0 * work["n_occ_blocks"] + 16.0 * work["n_tab_lookups"] + work["n_bases"] / 2,
        "k_sa": 64.0 * work["n_lf_steps"] + 8.0 * work["n_seeds"],
        "k_chain": 20.0 * work["n_seeds"] + 160.0 * work.get("n_bt_nodes", 0) + 64.0 * work.get("n_chain_recs", 0),
        "k_extend_wave": work["ref_bases"] / 4 + work["n_bases"] + 88.0 * work["n_regs_raw"],
        "k_dedup": 2 * 88.0 * work["n_regs_raw"],
    }
    dur = {"k_seed": stage_ms["ms_seed"], "k_sa": stage_ms["ms_sa"], "k_chain": stage_ms["ms_chain"], "k_extend_wave": stage_ms["ms_extend"], "k_dedup": stage_ms["ms_dedup"]}
    roof_k = max(dur, key=lambda k: dur[k])           # the longest kernel, whatever it is
    achieved = alg[roof_k] / (dur[roof_k] * 1e-3) / 1e9
    # HBM-side traffic of that kernel from the PMC counters of an earlier profiling run of the same workload (tools/profile_round.sh; separate
    # --pmc passes).  FETCH_SIZE counts 64 bytes per request whatever its size on this chip (profiles/r03_fetch_calibration.md: 64-byte reads
    # 1.00x, 32-byte reads 2.00x), so for a kernel whose fetches are 32-byte index blocks and 16-byte table entries the raw figure over-counts:
    # `traffic` is the corrected one (fetch / 2 + write for the 32-byte layout), the raw counters and the request count are beside it.
    traffic, traffic_src, traffic_detail = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    pj, pmc_note = None, None
    if world == 1 and not args.no_pmc and variant_files:
        log("[bench] counter passes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over one solo batch (two child processes)")
        pj, pmc_note = measure_traffic(prefix, variant_files[0], args.dense_sa, args.layout, os.path.dirname(prefix))
        if pj is None:
            log("[bench] counter passes failed:", pmc_note)
    measured_here = pj is not None
    if pj is None and os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
        except Exception:
            pj = None
    if pj is not None:
        try:
            names = [roof_k] + (["k_seed3"] if roof_k == "k_seed" else [])     # the seeding stage is two kernels since round 2 (pass 3 runs first)
            fetch = sum(pj.get(k_, {}).get("FETCH_SIZE_KB", 0.0) for k_ in names) * 1024.0       # (tools/pmc_summary.py's figures are per launch already)
            write = sum(pj.get(k_, {}).get("WRITE_SIZE_KB", 0.0) for k_ in names) * 1024.0
            if fetch > 0:
                factor = 0.5 if (roof_k in ("k_seed", "k_sa") and occ32) else 1.0
                traffic = fetch * factor + write
                traffic_detail = {"raw_bytes": fetch + write, "fetch_raw_bytes": fetch, "write_bytes": write, "fetch_correction": factor,
                                  "fetch_requests": fetch / 64.0, "corrected_over_device_layout_alg": None,
                                  "why": "FETCH_SIZE counts 64 bytes per request; this kernel's requests are 32-byte blocks and 16-byte entries (profiles/r03_fetch_calibration.md)" if factor != 1.0 else "64-byte requests: counter taken as is"}
            if traffic_detail:
                dev_alg = alg[roof_k] - (64.0 - blk_bytes) * work["n_occ_blocks"] if roof_k == "k_seed" else alg[roof_k]
                traffic_detail["corrected_over_device_layout_alg"] = round(traffic / dev_alg, 3) if dev_alg > 0 else None
            if measured_here:
                traffic_src = f"measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in two child processes over one solo batch of the headline's reads (tools/pmc_child.py, {pmc_note} s)"
            else:
                traffic_src = ("NOT measured in this run (" + str(pmc_note or "--no-pmc") + "): read from profiles/pmc_latest.json (" + str(pj.get("_meta", {}).get("what", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes")) +
                               ", taken " + str(pj.get("_meta", {}).get("date", "in an earlier run of the same workload")) + ")")
        except Exception:
            traffic = None
    layout = f"{n_batch // 2} pairs of 2x{args.read_len} bp (mates interleaved)" if pe else f"{n_batch} single-end {args.read_len} bp reads"
    # the chip's measured ceiling for lane-private dependent random reads (tools/randbw4.hip, profiles/r05_randbw4.md): the memory pipeline charges
    # 19 ps per DISTINCT block a lane asks for -- 52.5e9 requests/s -- whatever the request's size (16 or 32 bytes) or the number of load instructions.
    # (Rounds 2-4 quoted 48.3 / 37.8 / 23.0e9 by request size: that benchmark's own 64-bit modulo, not the memory system.)
    REQ_CEIL = 52.5e9
    n_blk, n_tab = work["n_occ_blocks"], work["n_tab_lookups"]
    spill_per_read = round(work_product["n_stack_spill"] / nr, 1) if work_product.get("n_stack_spill") is not None else None     # interval-stack entries the lanes fetched back from HBM
    seed_s = stage_ms["ms_seed"] * 1e-3
    out = {
        "metric": "Mreads/s (whole job) 2x150 bp vs GRCh38-scale index; SAM bit-identical to bwa mem (gate: `parity`).  `value` = the hot path (mem_align1_core of every read) on batches resident in HBM, "
                  "as the bench contract times it; the SAM-producing FASTQ->SAM rate of the same build is `end_to_end_pe.value` (also in `summary`)",
        "value": round(value, 4), "unit": "Mreads/s", "n_gpus": ranks_seen, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32/u64 (integer DP + FM-index ranks)", "data": "synthetic",
        "config": {"workload": f"{layout} per GPU per step vs seeded synthetic {args.genome_mbp:g} Mbp genome (GRCh38 stand-in, seq_len {2 * int(args.genome_mbp * 1e6):.3g}; BASELINE configs[{2 if pe else 1}] layout)",
                   "reads_per_gpu": n_batch, "read_len": args.read_len, "layout": args.layout, "genome_mbp": args.genome_mbp, "sa_intv": args.dense_sa or 32,
                   "occ_block_bytes": int(blk_bytes),
                   "sharding": f"reads x{world}, no collective", "batches_in_flight": S, "result_sha256_16": digest,
                   "timed": "kernels of the hot path on batches resident in HBM (no PCIe, no host finalize); see end_to_end_* for FASTQ->SAM"},
        "roofline": {"bound": "hbm", "kernel": roof_k + (" (+ k_seed3: the seeding stage)" if roof_k == "k_seed" else ""), "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_detail": traffic_detail, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": alg[roof_k], "kernel_ms": round(dur[roof_k], 3),
                     "rework": {"N_blk_per_read_algorithm": round(work["n_occ_blocks"] / nr, 1), "N_blk_per_read_product": round(work_product["n_occ_blocks"] / nr, 1),
                                "what": "the lane-per-read seeding kernel gives up reads that exceed its iteration budget (option seed_budget) and the task kernels seed them again: "
                                        "blocks touched twice are counted in the product's figure, not in the algorithmic bytes of `achieved`"},
                     "alg_bytes_note": "SURVEY 8(d)'s unit: 64 bytes per Occ block of the REFERENCE layout (N_blk counted per rank-query pair exactly as bwt_2occ4 does, 1 if k and l share a 128-base block else 2), "
                                       f"16 bytes per prefix-table entry, l_seq/2 for the read; the device's own index blocks are {int(blk_bytes)} bytes, see achieved_device_layout",
                     "achieved_device_layout": round((alg["k_seed"] - (64.0 - blk_bytes) * work["n_occ_blocks"]) / (dur["k_seed"] * 1e-3) / 1e9, 2) if roof_k == "k_seed" and dur["k_seed"] > 0 else None,
                     "per_kernel": {k: dict({"ms": round(dur[k], 3), "alg_GB": round(alg[k] / 1e9, 3), "GB/s": round(alg[k] / (dur[k] * 1e-3) / 1e9, 1) if dur[k] > 0 else None}, **kernel_traffic(pj, k, alg[k])) for k in dur},
                     # the same kernel against the two yardsticks that fit it better than SURVEY 8(d)'s 64-byte unit: the bytes of the device's own layout, and requests
                     "frac_device_bytes": round((alg["k_seed"] - (64.0 - blk_bytes) * n_blk) / seed_s / 1e9 / HBM_PEAK_GBS, 5) if seed_s > 0 else None,
                     "frac_requests": round((n_blk + n_tab) / REQ_CEIL / seed_s, 4) if seed_s > 0 else None,
                     "random_request_ceiling": {"requests_per_s": REQ_CEIL,
                                                "source": "tools/randbw4.hip on MI355X (profiles/r05_randbw4.md): 19 ps per distinct random block a lane asks for, whatever its size",
                                                # the seeding STAGE's algorithmic requests (index blocks + prefix-table entries of the budget-free algorithm; the interval stacks'
                                                # spill requests, the task kernels' rework and the interval lists are the kernel's own making and are NOT counted as useful work)
                                                "seeding_alg_requests_per_launch": n_blk + n_tab,
                                                "seeding_alg_requests_per_s": round((n_blk + n_tab) / seed_s, 0) if seed_s > 0 else None,
                                                "spill_requests_per_read": spill_per_read,
                                                "seeding_frac": round((n_blk + n_tab) / REQ_CEIL / seed_s, 4) if seed_s > 0 else None},
                     "ext_gcups": round(work["n_ext_cells"] / (stage_ms["ms_extend"] * 1e-3) / 1e9, 1) if stage_ms["ms_extend"] > 0 else None},
        "stage_ms_solo": {k: round(v, 3) for k, v in stage_ms.items()},
        "cigar_stage": cigar_stage,
        "work_per_read": {"N_blk": round(work["n_occ_blocks"] / nr, 1), "N_tab": round(work["n_tab_lookups"] / nr, 1), "N_lf": round(work["n_lf_steps"] / nr, 1),
                          "N_sa": round(work["n_seeds"] / nr, 2), "ext_cells": round(work["n_ext_cells"] / nr, 0), "ext_calls": round(work["n_ext_calls"] / nr, 2),
                          "ext_fast": round(work["n_ext_fast"] / nr, 2), "regs": round(work["n_regs"] / nr, 3)},
        "index_build": {"device_s": round(idx_info.get("build_ms", 0) / 1e3, 3) if idx_info else None,
                        "what": "bwagpu_index_build: suffix sort of forward+reverse text in HBM, BWT/Occ/SA in the reference's layout (byte-identical to bwa index)"},
    }
    if per_rank is not None:
        out["ranks"] = {"n": ranks_seen, "backend": backend + (" (RCCL)" if on_gpu else ""), "per_rank_Mreads_s": per_rank,
                        "what": "n = all-reduce of ones over the collective library; every rank's own hot-path rate (its reads x steps / its own time); `value` = all ranks' reads / the slowest rank's time"}
    if bcast_s is not None:
        out["index_broadcast"] = {"ranks": world, "seconds": round(bcast_s, 3), "what": "rank 0 loads the index files and uploads; RCCL broadcast of .bwt/.sa/.pac buffers over xGMI; every rank then builds its 32-byte blocks and prefix tables"}
    gpu.close()
    devices = list(range(world))              # the command-line product splits every batch over these (BWAGPU_DEVICES)
    if not args.no_cpu_baseline:
        threads = effective_cpus()
        cache = os.path.dirname(prefix)
        n_s = min(args.cpu_sample, n_batch) // 2 * 2
        # ---- single-end sample: reference vs product, same reads, same -K ----
        se_reads = batches[0][:n_s] if not pe else simdata.make_reads_se(g, n_s, length=args.read_len, seed=4001)
        fq = os.path.join(cache, "sample_se.fq")
        simdata.write_fastq(fq, se_reads)
        ref_se = run_reference(prefix, [fq], threads, os.path.join(cache, "ref_se.sam"))
        our_se = run_product(prefix, [fq], threads, os.path.join(cache, "our_se.sam"))       # (the gate: on device 0, the configuration every round has run on hardware)
        # ---- paired-end sample, MANY BATCHES: -K small enough that every one of the product's three device handles sees four batches (a
        # handle re-uses arenas, learnt sizes and packed buffers from batch to batch; mem_pestat depends on the batching, so -K is the
        # same on both sides, fastmap.c:394, bwamem.c:1258) ----
        n_mb = max(args.parity_pairs, 8) // 2 * 2
        K_mb = max(1, (n_mb * 2 * args.read_len) // 12 // 1_000_000) * 1_000_000           # twelve batches: four for each of the command line's three device handles
        p1, p2 = simdata.make_reads_pe(g, n_mb, length=args.read_len, seed=4002)
        f1, f2 = os.path.join(cache, "sample_1.fq"), os.path.join(cache, "sample_2.fq")
        simdata.write_fastq(f1, p1, suffix="/1"); simdata.write_fastq(f2, p2, suffix="/2")
        ref_pe = run_reference(prefix, [f1, f2], threads, os.path.join(cache, "ref_pe.sam"), K=K_mb)
        our_pe = run_product(prefix, [f1, f2], threads, os.path.join(cache, "our_pe.sam"), K=K_mb)
        par = {"se": False, "pe": False, "multibatch": False, "n_se": n_s, "n_pairs": n_mb,
               "how": f"sha256 of the SAM text minus @PG lines: oracle/_ref/bwa mem vs bwa-amd mem on the same FASTQ; single-end sample with -K 100000000 (one batch), paired-end sample with -K {K_mb} on both sides"}
        if ref_se and our_se:
            a, b = sam_body_digest(os.path.join(cache, "ref_se.sam")), sam_body_digest(os.path.join(cache, "our_se.sam"))
            par["se"] = a == b and a[1] >= n_s
            par["se_records"] = a[1]
        if ref_pe and our_pe:
            a, b = sam_body_digest(os.path.join(cache, "ref_pe.sam")), sam_body_digest(os.path.join(cache, "our_pe.sam"))
            par["pe"] = a == b and a[1] >= 2 * n_mb
            par["pe_records"] = a[1]
            par["pe_batches"] = our_pe["n_batches"]; par["pe_handles"] = our_pe["handles"]
            par["multibatch"] = bool(par["pe"] and our_pe["n_batches"] >= 4 * (our_pe["handles"] or 2))
            par["pe_product_Mreads_s"] = round(our_pe["reads_per_s"] / 1e6, 4)
            if world > 1:
                # ... and the same sample over all N devices (BWAGPU_DEVICES: the one-process multi-device path of `bwa-amd mem`, which no round could run on hardware).
                # Reported, not part of the exit code: a failure here must not take the N-rank hot-path measurement with it.
                try:
                    our_all = run_product(prefix, [f1, f2], threads, os.path.join(cache, "our_pe_all.sam"), K=K_mb, devices=devices, timeout=400)
                    par["pe_all_devices"] = {"devices": devices, "ok": bool(our_all and sam_body_digest(os.path.join(cache, "our_pe_all.sam")) == a),
                                             "Mreads_s": round(our_all["reads_per_s"] / 1e6, 4) if our_all else None}
                except Exception as e:
                    par["pe_all_devices"] = {"devices": devices, "ok": False, "error": repr(e)}
        if args.timed_sample > 0:
            log(f"[bench] parity of the timed batch: {args.timed_sample} reads of batch 0 vs the compiled reference's mem_align1_core")
            try:
                par["timed_batch"] = timed_batch_parity(prefix, opt, batches[0], counts, regs, args.timed_sample, threads)
            except Exception as e:
                par["timed_batch"] = {"ok": False, "error": repr(e)}
        out["parity"] = par
        # B_alg per read: the reference's own counts on a prefix of the paired-end sample next to what the device path does for
        # the same result (prefix tables replace short-match steps: N_blk falls, 16-byte look-ups appear; the SA is denser: N_lf falls)
        n_i = min(n_mb, max(1000, args.instr_pairs))
        fi1, fi2 = os.path.join(cache, "instr_1.fq"), os.path.join(cache, "instr_2.fq")
        simdata.write_fastq(fi1, p1[:n_i], suffix="/1"); simdata.write_fastq(fi2, p2[:n_i], suffix="/2")
        ins = run_instrumented(prefix, [fi1, fi2], threads)
        l_seq = args.read_len
        regs_pr = work["n_regs"] / nr
        dev_b = (alg["k_seed"] + alg["k_sa"]) / nr + (work["ref_bases"] / 4 + work["n_bases"] + 88.0 * work["n_regs"]) / nr
        out["b_alg_per_read"] = {"device": {"bytes": round(dev_b, 0), "N_blk": round(work["n_occ_blocks"] / nr, 1), "N_tab_16B": round(work["n_tab_lookups"] / nr, 1),
                                            "N_lf": round(work["n_lf_steps"] / nr, 1), "N_sa": round(work["n_seeds"] / nr, 2), "W_ref": round(work["ref_bases"] / nr, 0),
                                            "ext_cells": round(work["n_ext_cells"] / nr, 0), "sa_intv": args.dense_sa or 32}}
        if ins:
            ref_b = 64.0 * ins["N_blk"] + 64.0 * ins["N_lf"] + 8.0 * ins["N_sa"] + ins["W_ref"] / 4 + l_seq + 88.0 * regs_pr
            out["b_alg_per_read"]["reference"] = {"bytes": round(ref_b, 0), **{k_: round(v_, 2) for k_, v_ in ins.items() if k_ != "n_reads"}, "sa_intv": 32,
                                                  "how": f"oracle/_ref/bwa_instr (counters patched into a scratch copy of bwt.c/ksw.c/bwamem.c at build time) on {ins['n_reads']} reads of the paired-end sample; "
                                                         "B_alg = 64 N_blk + 64 N_lf + 8 N_sa + W_ref/4 + l_seq + 88 n_regs"}
        if not (par["se"] and par["pe"] and par["multibatch"]) or (par.get("timed_batch") is not None and not par["timed_batch"].get("ok")):
            rc_exit = 3
            log("[bench] PARITY GATE FAILED:", par)
        note = f"`bwa mem -t {threads}`, whole mem_process_seqs incl. SAM text, rate from its own per-batch real-time lines; the box exposes {os.cpu_count()} hardware threads but its cgroup quota is {threads} CPUs"
        if ref_pe:
            out["cpu_baseline"] = {"value": round(ref_pe["reads_per_s"] / 1e6, 4), "unit": "Mreads/s", "cores": threads, "kind": "reference",
                                   "sample": f"{n_mb} pairs of 2x{args.read_len} bp of the benchmark's read model (seed 4002), -K {K_mb}; {note}; {ref_pe['wall_s']:.1f}s wall incl. index load"}
            if ref_se:
                out["cpu_baseline"]["se"] = {"value": round(ref_se["reads_per_s"] / 1e6, 4), "sample": f"{n_s} single-end reads, -K 100000000"}
        if not args.no_e2e:
            # a few million reads, so that the figure reflects the pipeline's steady state rather than its fill and drain
            n_e = max(args.e2e_reads, n_batch) // 2 * 2
            r1, r2 = simdata.make_reads_pe(g, n_e // 2, length=args.read_len, seed=77)
            simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
            # (only the last batches' reads are needed again, by the tail gate: the 3 GB of the rest go back before the timed run -- inside this process the
            # command line ran 5-15 % below the stand-alone tools/e2e_bench.py, whose only difference is what the parent holds)
            per_batch_ = -(-100_000_000 // args.read_len); per_batch_ += per_batch_ & 1
            lo_tail = max(0, -(-n_e // per_batch_) - max(0, args.tail_batches)) * per_batch_
            r1_tail, r2_tail = r1[lo_tail // 2:].copy(), r2[lo_tail // 2:].copy()
            del r1, r2
            import gc
            gc.collect()
            e2e = run_product(prefix, [f1, f2], threads, None, devices=devices, timeout=400)
            if not e2e and world > 1:       # (the multi-device run failed: the line still gets the single-device figure, and says so)
                e2e = run_product(prefix, [f1, f2], threads, None)
                if e2e:
                    e2e["fallback_single_device"] = True
            if e2e:
                out["end_to_end_pe"] = {"value": round(e2e["reads_per_s"] / 1e6, 4), "unit": "Mreads/s", "n_gpus": 1 if e2e.get("fallback_single_device") else world, "stages": e2e["stages"], "stage_us_per_read": e2e["stage_us_per_read"],
                                        "device_stage_ms_per_batch": e2e["device_stage_ms_per_batch"], "steady_state": e2e.get("steady_state"), "handles": e2e["handles"], "retries": e2e["retries"], "cpu_us_per_read": e2e["cpu_us_per_read"],
                                        "what": f"`bwa-amd mem -t {threads}` on {n_e // 2} pairs as two FASTQ files (the BASELINE metric's layout, SAM discarded; the same command's SAM is what parity.pe compares): parsing + H2D + "
                                                f"device hot path + device CIGARs and mate-rescue alignments + D2H + mem_pestat/pairing/SAM text on the host, batches of 100 Mbp"
                                                + (f", every batch split over devices {devices} (BWAGPU_DEVICES)" if world > 1 else "") + "; wall time after the index is loaded"}
                if ref_pe:
                    out["end_to_end_pe"]["vs_cpu_baseline"] = round(e2e["reads_per_s"] / ref_pe["reads_per_s"], 1)
                if args.tail_batches > 0 and world == 1:
                    try:
                        out["parity"]["e2e_tail"] = e2e_tail_parity(args, prefix, [f1, f2], r1_tail, r2_tail, lo_tail, n_e, threads, cache)
                    except Exception as e:
                        out["parity"]["e2e_tail"] = {"ok": False, "error": repr(e)}
                    if not out["parity"]["e2e_tail"].get("ok"):
                        rc_exit = 3
                        log("[bench] PARITY GATE FAILED (tail of the end-to-end run):", out["parity"]["e2e_tail"])
                if world == 1 and args.e2e_handles > 0 and args.e2e_handles != e2e["handles"]:
                    # the same command with more batches in flight: a handle's share of the chip idles while its batch is in the download /
                    # mem_pestat / mate-rescue part of the device stage, which more handles fill (a measurement next to the default, not the default)
                    alt = run_product(prefix, [f1, f2], threads, None, streams=args.e2e_handles, timeout=120)
                    if alt:
                        out["end_to_end_pe"]["more_handles"] = {"handles": alt["handles"], "value": round(alt["reads_per_s"] / 1e6, 4), "stages": alt["stages"],
                                                                "device_stage_ms_per_batch": alt["device_stage_ms_per_batch"], "what": f"BWAGPU_CLI_STREAMS={args.e2e_handles}, otherwise the same command"}
            if world == 1:
                try:
                    out["ingest_gz"] = ingest_gz(prefix, [f1, f2], threads, cache)
                except Exception as e:
                    out["ingest_gz"] = {"error": repr(e)}
            if world == 1:
                e2e = run_product(prefix, [f1], threads, None)          # (the first file alone, as single-end reads: no second copy of the sample to write)
                if e2e:
                    out["end_to_end_se"] = {"value": round(e2e["reads_per_s"] / 1e6, 4), "unit": "Mreads/s", "stages": e2e["stages"], "stage_us_per_read": e2e["stage_us_per_read"], "what": f"the first FASTQ file alone: {n_e // 2} single-end reads"}
            del r1_tail, r2_tail
        if world == 1 and not args.no_longread:
            try:
                out["longread"] = longread_bench(args, prefix, g, threads, cache)
                if out["longread"].get("parity") is False or out["longread"].get("multibatch") is False:
                    rc_exit = 3
                    log("[bench] LONG-READ PARITY GATE FAILED")
            except Exception as e:   # (the long-read leg must not take the headline line with it)
                out["longread"] = {"error": repr(e)}
    if world == 1 and args.variants.strip() and not args.no_cpu_baseline:      # (a full run only: the profiling runs pass --no-cpu-baseline)
        # the A/B tables go to a side file (they were 15 KB of the line); the line keeps one row per configuration
        var = run_variants(args, prefix, variant_files, max(0.0, args.wall_budget - (time.time() - _T0)))
        side = os.path.join(ROOT, "gpurun_out", "bench_variants.json")
        try:
            os.makedirs(os.path.dirname(side), exist_ok=True)
            json.dump(var, open(side, "w"))
        except OSError:
            side = None
        brief = {"file": os.path.relpath(side, ROOT) if side else None, "what": var.get("what")}
        for leg_name, leg in var.items():
            if isinstance(leg, dict) and "runs" in leg:
                brief[leg_name] = {"rc": leg.get("rc"), "wall_s": leg.get("wall_s"),
                                   "runs": [dict({k_: r_.get(k_) for k_ in ("config", "ms_per_step", "ms_per_pass", "same_result_as_defaults", "error") if k_ in r_},
                                                 **({"ms_total_solo": r_["stage_ms_solo"].get("ms_total")} if isinstance(r_.get("stage_ms_solo"), dict) else {})) for r_ in leg["runs"]]}
            elif isinstance(leg, dict):
                brief[leg_name] = leg
        out["variants"] = brief
    out["bench_wall_s"] = round(time.time() - t_all, 1)
    # last on the line (a truncated tail of stdout still shows it) and once more on stderr: the numbers and the gates in one small object
    par_ = out.get("parity", {}); lr_ = out.get("longread", {}) if isinstance(out.get("longread"), dict) else {}
    e2e_v = out.get("end_to_end_pe", {}).get("value")
    out["summary"] = {"fastq_to_sam_Mreads_s": e2e_v, "what": "fastq_to_sam = the BASELINE metric (bit-identical SAM out of FASTQ, `end_to_end_pe`); value_hot_path = the resident hot path the bench contract times (`value`)",
                      "value_hot_path_Mreads_s": out["value"], "hot_path_over_fastq_to_sam": round(out["value"] / e2e_v, 3) if e2e_v else None,
                      "end_to_end_pe_Mreads_s": e2e_v, "roofline_frac": out["roofline"]["frac"],
                      "cpu_baseline_Mreads_s": out.get("cpu_baseline", {}).get("value"),
                      "parity": {"se": par_.get("se"), "pe": par_.get("pe"), "multibatch": par_.get("multibatch"), "timed_batch": (par_.get("timed_batch") or {}).get("ok"),
                                 "e2e_tail": (par_.get("e2e_tail") or {}).get("ok"),
                                 "long": lr_.get("parity"), "long_reads": lr_.get("parity_reads"), "long_multibatch": lr_.get("multibatch")},
                      "longread_reads_s": lr_.get("reads_per_s"), "rc": rc_exit}
    log("[bench] SUMMARY " + json.dumps(out["summary"]))
    sys.stdout.flush()
    # The line stays under 8 KB (the driver keeps a tail of stdout): explanatory strings longer than 150 characters are cut there and kept whole, with
    # everything else, in gpurun_out/bench_full.json (and on stderr).
    full = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(full), exist_ok=True)
        json.dump(out, open(full, "w"))
    except OSError:
        full = None
    log("[bench] FULL " + json.dumps(out))

    def slim(o, limit):
        if isinstance(o, dict):
            return {k_: slim(v_, limit) for k_, v_ in o.items()}
        if isinstance(o, list):
            return [slim(v_, limit) for v_ in o]
        if isinstance(o, str) and len(o) > limit:
            return o[:limit - 3] + "..."
        return o
    line = slim(out, 120)
    line["full_text"] = os.path.relpath(full, ROOT) if full else None
    # ... and, while it is still too long, secondary tables go to the file alone (least important first; the contract's fields, `roofline`,
    # `cpu_baseline`, `parity`, `end_to_end_pe` and `summary` always stay)
    for path in (("variants",), ("roofline", "per_kernel"), ("longread", "by_batch_size"), ("b_alg_per_read",), ("end_to_end_pe", "more_handles"), ("roofline", "traffic_detail"),
                 ("end_to_end_se",), ("roofline", "rework"), ("longread", "end_to_end"), ("cigar_stage",), ("end_to_end_pe", "device_stage_ms_per_batch"), ("longread", "dp_cells_per_read"),
                 ("roofline", "random_request_ceiling"), ("work_per_read",), ("index_build",), ("longread", "stage_ms")):
        if len(json.dumps(line)) < 7900:
            break
        o_ = line
        for k_ in path[:-1]:
            o_ = o_.get(k_) if isinstance(o_, dict) else None
        if isinstance(o_, dict) and path[-1] in o_:
            o_[path[-1]] = "in full_text"
    print(json.dumps(line), flush=True)      # the one JSON line, last thing on stdout (RCCL prints a version banner of its own at start-up)
    sys.exit(rc_exit)


def write_bgzf(path: str, data: bytes, level: int = 4):
    """The bytes as a BGZF file (bgzip's format, SAM spec 4.1: gzip members of <= 64 KiB whose extra field BC holds the member's length) -- no bgzip here."""
    import struct
    import zlib
    with open(path, "wb") as f:
        for o in list(range(0, len(data), 65280)) + [len(data)]:          # (the last, empty block is the end-of-file marker)
            chunk = data[o:o + 65280] if o < len(data) else b""
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            body = c.compress(chunk) + c.flush()
            f.write(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(body) + 8 - 1) + body + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))


def ingest_gz(prefix, files, threads, cache, n_pairs=500_000):
    """Compressed input: the input stage alone (`bwa-amd mem` with BWAGPU_CLI_PARSE_ONLY: read + inflate + parse, no alignment) on the first
    n_pairs pairs of the end-to-end sample as a pair of plain gzip files (one zlib stream each: one inflating thread per file is all there can be) and as a
    pair of BGZF files (blocks inflated by the input pool side by side)."""
    import shutil
    cli = os.path.join(ROOT, "bwa_amd", "bwa-amd")
    res = {"pairs": n_pairs, "what": "input stage alone (read + inflate + parse; BWAGPU_CLI_PARSE_ONLY=1), Mreads/s: a pair of FASTQ files plain / gzip (one stream per file) / BGZF (blocks inflated in parallel)"}
    subs = []
    for k, f in enumerate(files):
        with open(f, "rb") as fi:
            data = b"".join(fi.readline() for _ in range(4 * n_pairs))
        sub = os.path.join(cache, f"ingest_{k}.fq")
        open(sub, "wb").write(data)
        if shutil.which("gzip"):
            subprocess.run(f"gzip -4 -c {sub} > {sub}.gz", shell=True, check=True)
        write_bgzf(sub + ".bgzf.gz", data)
        subs.append(sub)
        del data
    for name, ext in (("plain", ""), ("gzip", ".gz"), ("bgzf", ".bgzf.gz")):
        fl = [s_ + ext for s_ in subs]
        if not all(os.path.exists(x) for x in fl):
            continue
        p = subprocess.run([cli, "mem", "-t", str(threads), "-v", "3", prefix] + fl, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=dict(os.environ, BWAGPU_CLI_PARSE_ONLY="1"), timeout=120)
        m = re.search(r"parsed (\d+) records \(\d+ bp\) in ([\d.]+) s", p.stderr)
        if p.returncode == 0 and m and float(m.group(2)) > 0:
            res[name + "_Mreads_s"] = round(int(m.group(1)) / float(m.group(2)) / 1e6, 3)
    for s_ in subs:
        for ext in ("", ".gz", ".bgzf.gz"):
            try:
                os.remove(s_ + ext)
            except OSError:
                pass
    return res


def e2e_tail_parity(args, prefix, files, r1, r2, lo_tail, n_e, threads, cache):      # (r1, r2: the pairs from read number lo_tail on)
    """Arena re-use at depth, at full batch size: the SAM of the LAST batches of the end-to-end input (every handle's last batch: its arenas, learnt sizes and
    packed buffers have been through five or six batches by then) against the compiled reference's mem_process_seqs on exactly those batches.  The reference is
    called per batch through tests/refapi.py with the batch's own n_processed -- mem_pair's tie-breaking hash takes the pair's number in the whole run
    (bwamem_pair.c:208,248), so `bwa mem` on the tail's reads alone would not do -- and with the batches `bwa mem -K 100000000` forms (fastmap.c:394, bwa.c:bseq_read)."""
    import refapi
    from bwa_amd import simdata
    from bwa_amd.structs import default_opt
    if not refapi.have_ref():
        return {"ok": False, "error": "oracle/_ref is missing"}
    L = args.read_len
    per_batch = -(-100_000_000 // L); per_batch += per_batch & 1
    n_b = -(-n_e // per_batch)
    b0 = max(0, n_b - args.tail_batches)
    t = time.time()
    our = run_product(prefix, files, threads, os.path.join(cache, "our_tail.sam"), out_from_batch=b0)
    res = {"batches": n_b - b0, "of": n_b, "reads": int(n_e - b0 * per_batch), "ok": False}
    if not our:
        return res
    h = hashlib.sha256(); got_n = 0
    with open(os.path.join(cache, "our_tail.sam"), "rb") as f:
        for line in f:
            if line[:1] != b"@":
                h.update(line); got_n += 1
    os.remove(os.path.join(cache, "our_tail.sam"))
    ref = refapi.RefIndex(prefix)
    opt = default_opt(); opt.flag |= 0x2; opt.n_threads = threads
    hw = hashlib.sha256(); want_n = 0
    for bi in range(b0, n_b):
        lo, hi = bi * per_batch, min(n_e, (bi + 1) * per_batch)
        rd = interleave(r1[(lo - lo_tail) // 2: (hi - lo_tail) // 2], r2[(lo - lo_tail) // 2: (hi - lo_tail) // 2])
        names = [f"r{(lo + i) >> 1}" for i in range(hi - lo)]
        txt = ref.process_seqs(opt, names, simdata._ASCII[rd].tobytes(), b"I" * (rd.shape[0] * L), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * L, n_processed=lo)
        hw.update(txt); want_n += txt.count(b"\n")
        del rd, names, txt
    ref.close()
    res["ok"] = bool(h.digest() == hw.digest() and want_n >= res["reads"]); res["records"] = want_n; res["product_records"] = got_n; res["seconds"] = round(time.time() - t, 1)
    res["how"] = (f"`bwa-amd mem` on all {n_e // 2} pairs writing only the records of its last {n_b - b0} batches (BWAGPU_CLI_OUT_FROM_BATCH={b0}) vs the compiled reference's "
                  "mem_process_seqs called batch by batch with each batch's own n_processed (the pair number enters mem_pair's tie-breaking hash): sha256 of the SAM records")
    return res


def timed_batch_parity(prefix, opt, reads, counts, regs, n_s, threads):
    """The regions the TIMED loop left for the first n_s reads of batch 0 against the compiled reference's mem_align1_core on the same reads
    (tests/refapi.py -> oracle/_ref/libbwaref.so; the reference's call is per read, so the slices run on `threads` host threads)."""
    import threading
    import refapi
    if not refapi.have_ref() or n_s <= 0:
        return None
    t = time.time()
    ref = refapi.RefIndex(prefix)
    t_load = time.time() - t
    n_s = min(n_s, reads.shape[0])
    L = reads.shape[1]
    cuts = np.linspace(0, n_s, max(1, min(threads, n_s // 64 or 1)) + 1).astype(np.int64)
    parts = [None] * (len(cuts) - 1)

    def work(i):
        a, b = int(cuts[i]), int(cuts[i + 1])
        sub = np.ascontiguousarray(reads[a:b].reshape(-1))
        parts[i] = ref.align(opt, sub, np.arange(0, b - a + 1, dtype=np.int64) * L)
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
    [x.start() for x in th]; [x.join() for x in th]
    ref.close()
    rc = np.concatenate([p_[0] for p_ in parts]); rr = np.concatenate([p_[1] for p_ in parts])
    n_regs = int(counts[:n_s].sum())
    same = bool(np.array_equal(rc, counts[:n_s]) and rr.tobytes() == regs[:n_regs].tobytes())
    return {"ok": same, "reads": int(n_s), "regions": n_regs, "seconds": round(time.time() - t, 1), "index_load_s": round(t_load, 1),
            "how": "first reads of timed batch 0: counts + 88-byte mem_alnreg_t records downloaded after the timed loop == the compiled reference's mem_align1_core (refshim_align) on the same reads, byte for byte"}


def run_variants(args, prefix, batch_files=(), wall_left=1e9):
    """Kernel variants that sit behind library options (bwagpu_set_option), A/B'd against the defaults on the headline's workload by tools/variant_probe.py
    in a CHILD process with a time limit: solo stage times, step time with the same batches in flight, and a digest of the regions that
    must equal the defaults'.  Informational -- `value` above is always the default configuration's; a variant that faults or hangs costs
    this object its entries and nothing else."""
    cfgs = [c.strip() for c in args.variants.split(";") if c.strip()]
    probe = [sys.executable, os.path.join(ROOT, "tools", "variant_probe.py"), "--prefix", prefix, "--codes", prefix + ".codes.npy", "--dense-sa", str(args.dense_sa)]
    res = {"what": "tools/variant_probe.py in child processes: each configuration's solo stage times (ms per batch), step time with --streams batches in flight (short reads) or "
                   "time per pass (long reads), and whether its regions equal the default configuration's; `value` is never taken from here"}
    legs = [("short_reads", ["--reads", str(args.reads), "--read-len", str(args.read_len), "--streams", str(args.streams), "--steps", "6"] +
             (["--batch-files", ",".join(batch_files)] if batch_files else []), args.variants_timeout)]
    if not args.no_longread:
        long_file = os.path.join(os.path.dirname(prefix), "long_reads.npy")       # (left there by the long-read leg)
        legs.append(("long_reads", ["--long-reads", str(args.long_reads), "--long-len", str(args.long_len), "--passes", "1"] +
                     (["--long-file", long_file] if os.path.exists(long_file) else []), args.variants_timeout * 0.8))
    t_legs = time.time()
    for k_leg, (name, extra, limit) in enumerate(legs):
        left = wall_left - (time.time() - t_legs)
        limit = min(limit, left * 0.55 if k_leg + 1 < len(legs) else left)      # (the bench line must come out within --wall-budget whatever the children do)
        if limit < 20.0:
            res[name] = {"skipped": f"{left:.0f} s of --wall-budget left"}
            continue
        if name == "long_reads":
            # the long-read defaults of round 4 against the forms they replaced: all of round 3's together (what BENCH_r03's `longread` ran), then each alone
            alone = ["seed_mrg=0", "seed_tasks=0", "publish_blk=0", "seedsw_lds=0", "dedup_blk=0"]
            cfgs = [" ".join(alone)] + alone          # (all together first: the entry to have if the leg runs out of its time)
        log(f"[bench] variants, {name} (child process, <= {limit:.0f} s): {cfgs}")
        t = time.time()
        leg = {"runs": []}
        text, leg["rc"] = run_child(probe + extra + cfgs, limit, leg)
        for line in text.splitlines():
            try:
                leg["runs"].append(json.loads(line))
            except ValueError:
                pass
        leg["wall_s"] = round(time.time() - t, 1)
        res[name] = leg
    return res


def run_child(cmd, limit, leg):
    """Run `cmd` in a session of its own for at most `limit` seconds; returns (stdout so far, return code | "timeout").  The child's
    output goes to files, not pipes, and a child that does not die within 10 s of SIGKILL (a process stuck on a wedged device) is left
    behind rather than waited for: the bench line must still come out."""
    import signal
    import tempfile
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        p = subprocess.Popen(cmd, stdout=fo, stderr=fe, text=True, start_new_session=True)
        try:
            rc = p.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            rc = "timeout"
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                leg["unreaped"] = True
        fo.seek(0); fe.seek(0)
        if rc != 0:
            leg["stderr_tail"] = fe.read()[-400:]
        return fo.read(), rc


def longread_bench(args, prefix, g, threads, cache):
    """BASELINE configs[4]: 10 kb PacBio-like reads (SURVEY 8d error model: 1.5 % substitutions, 4 % deletions, 9 % insertions) with
    `-x pacbio` against the same index.  Resident hot path rate of one batch, DP-cell throughput (extension + patch alignments +
    seed re-scoring), the reference on a prefix of the reads, and a SAM parity gate on that prefix through `bwa-amd mem -x pacbio`."""
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import pacbio_opt
    L, n = args.long_len, args.long_reads
    log(f"[bench] long-read leg: {n} reads of {L} bp")
    reads = simdata.make_reads_long(g, n, length=L, seed=7)
    np.save(os.path.join(cache, "long_reads.npy"), reads)         # (the variants leg aligns the same batch)
    gpu = BwaGpu(prefix)
    if args.dense_sa:
        gpu.densify_sa(args.dense_sa)
    gpu.set_taps(False)
    opt = pacbio_opt()
    gpu.upload(np.ascontiguousarray(reads.reshape(-1)), np.arange(0, n + 1, dtype=np.int64) * L)
    gpu.set_stats(True); gpu.run(opt); work = gpu.stats()          # (also the warm-up: arenas learn their sizes)
    gpu.set_stats(False)
    ms, st = [], None
    for _ in range(args.long_steps):
        t = time.perf_counter(); gpu.run(opt); ms.append((time.perf_counter() - t) * 1e3); st = gpu.stats()
    best = min(ms)
    log(f"[bench] long-read hot path: {best:.1f} ms per pass; stages {st}")
    by_size = {}
    for m_ in [int(x) for x in args.long_extra.split(",") if x.strip()]:      # other batch sizes (10 000 reads of 10 kb = the reference's -K of 100 Mbase, fastmap.c:394)
        if m_ == n or time.time() - _T0 > args.wall_budget + 120:
            continue
        try:
            rd2 = simdata.make_reads_long(g, m_, length=L, seed=7 + m_)
            gpu.upload(np.ascontiguousarray(rd2.reshape(-1)), np.arange(0, m_ + 1, dtype=np.int64) * L)
            del rd2
            gpu.run(opt)
            t = time.perf_counter(); gpu.run(opt); d_ = (time.perf_counter() - t) * 1e3
            s2 = gpu.stats()
            by_size[str(m_)] = {"reads_per_s": round(m_ / d_ * 1e3, 1), "ms_per_pass": round(d_, 1), "stage_ms": {k: round(s2[k], 1) for k in ("ms_seed", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup")}}
            log(f"[bench] long-read hot path at {m_} reads per batch: {d_:.0f} ms per pass")
        except Exception as e:
            by_size[str(m_)] = {"error": repr(e)}
    gpu.close()
    cells = work["n_ext_cells"] + work["n_glb_cells"] + work["n_sw_cells"]
    dp_ms = st["ms_extend"] + st["ms_dedup"] + st["ms_seedsw"]
    res = {"what": f"BASELINE configs[4] layout: {n} reads of {L} bp (1.5 % sub, 4 % del, 9 % ins), -x pacbio, same index; one batch resident in HBM, best of {args.long_steps} passes of the hot path",
           "reads_per_s": round(n / best * 1e3, 1), "Mbp_per_s": round(n * L / best / 1e3, 2), "ms_per_pass": round(best, 2),
           "stage_ms": {k: round(st[k], 2) for k in ("ms_seed", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup", "ms_total")},
           "dp_cells_per_read": {"extend": round(work["n_ext_cells"] / n), "global_score": round(work["n_glb_cells"] / n), "seed_sw": round(work["n_sw_cells"] / n)},
           "gcups": round(cells / (dp_ms * 1e-3) / 1e9, 1) if dp_ms > 0 else None,
           "by_batch_size": by_size,
           "gcups_note": "extension + patch (score-only global) + seed re-scoring cells / the three kernels' time; the final CIGARs' cells are counted under end_to_end"}
    # ---- reference and product command lines on a prefix ----
    # (-K a tenth of the prefix's bases: the product's three device handles each see three or four batches -- arenas, learnt sizes, the long CIGAR
    # tier's scratch re-used from batch to batch; single-end SAM does not depend on the batching, the reference gets the same -K anyway)
    n_p = min(args.long_sample, n)
    K_long = max(L + 1, n_p * L // 10)
    fq = os.path.join(cache, "long_sample.fq")
    simdata.write_fastq(fq, reads[:n_p])
    ref = run_reference(prefix, [fq], threads, os.path.join(cache, "ref_long.sam"), K=K_long, extra=["-x", "pacbio"], timeout=150)
    our = run_product(prefix, [fq], threads, os.path.join(cache, "our_long.sam"), K=K_long, extra=["-x", "pacbio"], timeout=150)
    if ref:
        res["cpu_baseline"] = {"value": round(ref["reads_per_s"], 1), "unit": "reads/s", "cores": threads, "kind": "reference",
                               "sample": f"first {n_p} reads, `bwa mem -x pacbio -t {threads} -K 100000000`, rate from its own per-batch real-time lines; {ref['wall_s']:.1f}s wall"}
    # ---- the product on the whole batch: FASTQ -> SAM rate with the CIGARs, NM and MD of the long alignments computed on the device ----
    fq_all = os.path.join(cache, "long_all.fq")
    simdata.write_fastq(fq_all, reads)
    e2e_sam = os.path.join(cache, "our_long_all.sam")
    e2e = run_product(prefix, [fq_all], threads, e2e_sam, extra=["-x", "pacbio"], timeout=200, K=max(1000000, n * L // 3 + L))      # one batch per device handle
    if e2e:
        res["end_to_end"] = {"reads_per_s": round(e2e["reads_per_s"], 1), "Mbp_per_s": round(e2e["reads_per_s"] * L / 1e6, 2), "stages": e2e["stages"],
                             "device_stage_ms_per_batch": e2e["device_stage_ms_per_batch"],
                             "what": f"`bwa-amd mem -x pacbio -t {threads}` on all {n} reads, FASTQ -> SAM (SAM discarded; the prefix's SAM is what `parity` compares)"}
    if ref and our:
        a, b = sam_body_digest(os.path.join(cache, "ref_long.sam")), sam_body_digest(os.path.join(cache, "our_long.sam"))
        res["parity"] = bool(a == b and a[1] >= n_p)
        res["parity_records"] = a[1]; res["parity_reads"] = n_p
        res["parity_batches"] = our["n_batches"]; res["parity_handles"] = our["handles"]
        res["multibatch"] = bool(res["parity"] and our["n_batches"] >= 3 * (our["handles"] or 3))
        res["parity_how"] = f"sha256 of the SAM text minus @PG: oracle/_ref/bwa mem -x pacbio vs bwa-amd mem -x pacbio on the first {n_p} reads, -K {K_long} on both sides"
        if e2e and res["parity"]:
            # the whole batch's FASTQ -> SAM run (three batches of n/3 reads, full-size arenas): its records for the first n_p reads must be the reference's too
            def body(path, limit):
                out_ = []
                with open(path, "rb") as f:
                    for line in f:
                        if line[:1] != b"@":
                            out_.append(line)
                            if len(out_) >= limit:
                                break
                return out_
            want = body(os.path.join(cache, "ref_long.sam"), 1 << 60)
            got = body(e2e_sam, len(want))
            res["parity_full_run_prefix"] = bool(want == got)
            if not res["parity_full_run_prefix"]:
                res["parity"] = False
    if e2e:
        try:
            os.remove(e2e_sam)
        except OSError:
            pass
    return res


if __name__ == "__main__":
    main()
