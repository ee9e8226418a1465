#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X BWA-MEM hot path (mem_align1_core for every read of a batch).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by torch.distributed.run
with one rank per GPU.  One "step" = one pass of the hot path (seed -> SA -> chain -> extend -> dedup) over one batch
of synthetic reads that is already resident in HBM.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): `--reads` (default 1 M) synthetic 150 bp single-end reads per GPU against a
seeded synthetic stand-in for GRCh38 (`--genome-mbp`, default below; no genome data exists offline and a 3.1 Gbp
index cannot be built inside the GPU-time budget -- see DESIGN.md section 6).  Reads shard across ranks with no
data-path collective ("weak" scaling: every rank aligns its own `--reads`).

Extra objects on the JSON line:
  roofline     dominant kernel's algorithmic bytes / its measured duration (HIP events on the library's stream)
  cpu_baseline the unmodified reference (`oracle/_ref/bwa mem -t C`) on a bounded sample of the same reads
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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_or_load_index(genome_mbp: float, cache: str, rank: int, barrier):
    """Seeded synthetic genome + reference-format index, built on the GPU by bwa_amd.index (byte-identical to `bwa index`
    output, tests/test_index_build.py) and cached on the box."""
    from bwa_amd import simdata
    total = int(genome_mbp * 1_000_000)
    prefix = os.path.join(cache, f"g{total}_s42")
    g, lens = simdata.make_genome(total, n_contigs=8, seed=42)
    if rank == 0 and not os.path.exists(prefix + ".sa"):
        from bwa_amd.index import build_index
        os.makedirs(cache, exist_ok=True)
        t = time.time()
        build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)])
        import torch
        torch.cuda.empty_cache()
        log(f"[bench] built {genome_mbp} Mbp index on the device in {time.time() - t:.1f}s")
    barrier()
    return prefix, g


def cpu_baseline(fa: str, reads: np.ndarray, threads: int):
    """Unmodified reference `bwa mem -t threads` on a bounded sample; reads/s from its own per-batch timing lines
    (bwamem.c:1263: '[M::mem_process_seqs] Processed N reads in X CPU sec, Y real sec')."""
    from bwa_amd import simdata
    bwa = os.path.join(ROOT, "oracle", "_ref", "bwa")
    fq = os.path.join(os.path.dirname(fa), f"sample_{reads.shape[0]}.fq")
    simdata.write_fastq(fq, reads)
    p = subprocess.run([bwa, "mem", "-t", str(threads), "-K", "100000000", "-v", "3", fa, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    n = tot = 0.0
    for m in re.finditer(r"Processed (\d+) reads in ([\d.]+) CPU sec, ([\d.]+) real sec", p.stderr):
        n += int(m.group(1)); tot += float(m.group(3))
    if p.returncode != 0 or tot <= 0:
        return None
    return n / tot


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


def end_to_end(fa: str, files, threads: int, streams: int = 3):
    """The stand-alone `bwa-amd mem` (FASTQ in -> device hot path + device CIGARs -> host finalize -> SAM text out) on FASTQ
    files; whole-run reads/s as the program reports it after the index is loaded (input parsing and output included)."""
    cli = os.path.join(ROOT, "bwa_amd", "bwa-amd")
    if not (os.path.exists(cli) and all(os.path.exists(f) for f in files)):
        return None
    p = subprocess.run([cli, "mem", "-t", str(threads), "-K", "100000000", "-v", "3", fa] + list(files), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, BWAGPU_CLI_STREAMS=str(streams)))
    m = re.search(r"\[M::main_mem\] (\d+) reads in ([\d.]+) sec .*: (\d+) reads/s", p.stderr)
    if p.returncode != 0 or not m:
        return None
    return float(m.group(3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=1_000_000, help="reads per GPU per step")
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--genome-mbp", type=float, default=512.0)
    ap.add_argument("--cache", default=os.environ.get("BWA_AMD_CACHE", "/tmp/bwa_amd_bench"))
    ap.add_argument("--streams", type=int, default=3, help="batches in flight per GPU (handles sharing the index)")
    ap.add_argument("--dense-sa", type=int, default=4, help="densify the SA on the device to this interval (0 = keep the reference's 32)")
    ap.add_argument("--cpu-sample", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))   # "nccl" is RCCL on ROCm
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt

    fa, g = build_or_load_index(args.genome_mbp, args.cache, rank, barrier)
    if dist is not None:                     # rank 0 loads + uploads, the others receive the index over RCCL/xGMI
        from bwa_amd import dist as bdist
        t_b = time.perf_counter()
        gpu = bdist.broadcast_index(fa, device=local, src=0)
        if rank == 0:
            log(f"[bench] index broadcast to {world} GPUs in {time.perf_counter() - t_b:.2f}s")
    else:
        gpu = BwaGpu(fa, device=local)       # index resident in this GPU's HBM (no CPU fallback: raises without a GPU)
    if args.dense_sa:
        gpu.densify_sa(args.dense_sa)
    gpu.set_taps(False)
    opt = default_opt()

    # S batches in flight per GPU: S handles share the resident index (bwagpu_clone), each with its own stream and arenas and
    # driven by its own host thread, so the latency-bound tails of one batch overlap the throughput-bound kernels of another.
    import threading
    S = max(1, args.streams)
    handles = [gpu] + [gpu.clone() for _ in range(S - 1)]
    batches = []
    for si, hdl in enumerate(handles):
        rd = simdata.make_reads_se(g, args.reads, length=args.read_len, seed=1000 + rank * 64 + si)   # this rank's shard(s)
        hdl.set_taps(False)
        hdl.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, args.reads + 1, dtype=np.int64) * args.read_len)   # resident in HBM
        batches.append(rd)
    reads = batches[0]

    # one untimed instrumented solo pass: algorithmic work counters of batch 0 (roofline numerator) and solo kernel times
    gpu.set_stats(True)
    gpu.run(opt)
    work = gpu.stats()
    gpu.set_stats(False)
    gpu.run(opt)
    solo = gpu.stats()
    stage_keys = ("ms_seed", "ms_publish", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup", "ms_total")
    stage_ms = {k: solo[k] for k in stage_keys}

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
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    counts, regs = gpu.download()
    digest = hashlib.sha256(counts.tobytes() + regs.tobytes()).hexdigest()[:16]
    for hdl in handles[1:]:
        hdl.close()

    if rank == 0:
        total_reads = args.reads * world * args.steps
        value = total_reads / dt / 1e6
        # algorithmic bytes per launch of each index-bound kernel: one 64-byte block per Occ lookup /
        # LF step, 8 bytes per SA sample, plus the read bases the seeding kernel consumes
        alg = {
            "k_seed": 64.0 * work["n_occ_blocks"] + 16.0 * work["n_tab_lookups"] + work["n_bases"] / 2,
            "k_sa": 64.0 * work["n_lf_steps"] + 8.0 * work["n_seeds"],
        }
        dur = {"k_seed": stage_ms["ms_seed"], "k_sa": stage_ms["ms_sa"]}
        dom = max(("k_seed", "k_sa", "k_extend", "k_chain", "k_dedup"), key=lambda k: {"k_extend": stage_ms["ms_extend"], "k_chain": stage_ms["ms_chain"], "k_dedup": stage_ms["ms_dedup"], **dur}[k])
        roof_k = dom if dom in alg else "k_seed"
        achieved = alg[roof_k] / (dur[roof_k] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(roof_k, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "Mreads/s (whole job), hot path mem_align1_core, regs bit-identical to bwa mem",
            "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32/u64 (integer DP + FM-index ranks)", "data": "synthetic",
            "config": {"workload": f"{args.reads} synthetic {args.read_len}bp SE reads per GPU vs seeded synthetic {args.genome_mbp} Mbp genome (GRCh38 stand-in, BASELINE configs[1])",
                       "reads_per_gpu": args.reads, "read_len": args.read_len, "genome_mbp": args.genome_mbp, "sa_intv": args.dense_sa or 32,
                       "sharding": f"reads x{world}, no collective", "batches_in_flight": S, "result_sha256_16": digest},
            "roofline": {"bound": "hbm", "kernel": roof_k, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "alg_bytes_per_launch": alg[roof_k], "kernel_ms": round(dur[roof_k], 3),
                         "blocks_64B_per_s": round(alg[roof_k] / 64.0 / (dur[roof_k] * 1e-3), 0),
                         "random_64B_ceiling": {"GB/s": 1670.0, "blocks_per_s": 26.0e9, "frac": round(achieved / 1670.0, 4),
                                                "source": "tools/randbw.hip on MI355X, profiles/r01_randbw_microbench.md: random 64-byte reads from HBM saturate at 26e9/s"}},
            "stage_ms_solo": {k: round(v, 3) for k, v in stage_ms.items()},
            "work_per_read": {"N_blk": round(work["n_occ_blocks"] / work["n_reads"], 1), "N_tab": round(work["n_tab_lookups"] / work["n_reads"], 1), "N_lf": round(work["n_lf_steps"] / work["n_reads"], 1),
                              "N_sa": round(work["n_seeds"] / work["n_reads"], 2), "ext_cells": round(work["n_ext_cells"] / work["n_reads"], 0),
                              "regs": round(work["n_regs"] / work["n_reads"], 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            threads = effective_cpus()
            n_s = min(args.cpu_sample, args.reads)
            t = time.time()
            rps = cpu_baseline(fa, reads[:n_s], threads)
            if rps:
                out["cpu_baseline"] = {"value": round(rps / 1e6, 4), "unit": "Mreads/s", "cores": threads, "kind": "reference",
                                       "sample": f"first {n_s} reads of the same batch, `bwa mem -t {threads} -K 100000000` (whole mem_process_seqs incl. SAM text; "
                                                 f"the box exposes {os.cpu_count()} hardware threads but its cgroup quota is {threads} CPUs), {time.time() - t:.1f}s wall"}
            gpu.close()
            cache = os.path.dirname(fa)
            fq = os.path.join(cache, "e2e_se.fq")
            # a few million reads, so that the figure reflects the pipeline's steady state rather than its fill and drain
            all_reads = np.concatenate(batches + [simdata.make_reads_se(g, max(0, 6 * args.reads - len(batches) * args.reads), length=args.read_len, seed=4242)])
            simdata.write_fastq(fq, all_reads)
            e2e = end_to_end(fa, [fq], threads, streams=2)
            if e2e:
                out["end_to_end"] = {"value": round(e2e / 1e6, 4), "unit": "Mreads/s",
                                     "what": f"`bwa-amd mem -t {threads}` on {all_reads.shape[0]} reads as FASTQ: parsing + H2D + device hot path + device CIGARs + D2H + "
                                             f"host finalize + SAM text, pipelined over batches of 100 Mbp with 2 in flight; wall time after the index is loaded"}
            n_pairs = 2 * args.reads
            r1, r2 = simdata.make_reads_pe(g, n_pairs, length=args.read_len, seed=77)
            f1, f2 = os.path.join(cache, "e2e_1.fq"), os.path.join(cache, "e2e_2.fq")
            simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
            e2e = end_to_end(fa, [f1, f2], threads, streams=2)
            if e2e:
                out["end_to_end_pe"] = {"value": round(e2e / 1e6, 4), "unit": "Mreads/s",
                                        "what": f"same, {n_pairs} pairs of 2x{args.read_len} bp (BASELINE metric's read layout): adds mem_pestat and pairing on the host, mate-rescue alignments on the device"}
        print(json.dumps(out), flush=True)
    gpu.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
