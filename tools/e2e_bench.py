#!/usr/bin/env python3
"""End-to-end throughput of the stand-alone `bwa-amd mem` (FASTQ in -> SAM out) on the bench genome; GPU box only.

usage: e2e_bench.py [--genome-mbp 512] [--reads 1000000] [--threads 64,128] [--streams 1,2,3] [--pe]
Prints one line per configuration: wall reads/s as reported by the program after the index was loaded."""
import argparse, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bwa_amd import simdata


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=512.0)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--threads", default="64,128")
    ap.add_argument("--streams", default="1,2,3")
    ap.add_argument("--chunk", default="100000000")
    ap.add_argument("--pe", action="store_true")
    ap.add_argument("--cache", default=os.environ.get("BWA_AMD_CACHE", "/tmp/bwa_amd_bench"))
    a = ap.parse_args()
    import torch
    fa, g = bench.build_or_load_index(a.genome_mbp, a.cache, 0, lambda: torch.cuda.synchronize())
    t = time.time()
    if a.pe:
        r1, r2 = simdata.make_reads_pe(g, a.reads // 2, seed=77)
        f1, f2 = os.path.join(a.cache, "e2e_1.fq"), os.path.join(a.cache, "e2e_2.fq")
        simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
        files = [f1, f2]
    else:
        f1 = os.path.join(a.cache, "e2e_se.fq")
        simdata.write_fastq(f1, simdata.make_reads_se(g, a.reads, seed=77))
        files = [f1]
    print(f"[e2e] inputs written in {time.time() - t:.1f}s", flush=True)
    cli = os.path.join(ROOT, "bwa_amd", "bwa-amd")
    for th in a.threads.split(","):
        for st in a.streams.split(","):
            env = dict(os.environ, BWAGPU_CLI_STREAMS=st)
            t = time.time()
            p = subprocess.run([cli, "mem", "-t", th, "-K", a.chunk, "-v", "3", fa] + files, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
            m = re.search(r"\[M::main_mem\] (\d+) reads in ([\d.]+) sec .*: (\d+) reads/s", p.stderr)
            print(f"[e2e] {'PE' if a.pe else 'SE'} -t {th} streams {st} -K {a.chunk}: rc={p.returncode} {m.group(0) if m else p.stderr[-300:]}  (process wall {time.time() - t:.1f}s)", flush=True)
            m2 = re.search(r"stage busy time: .*", p.stderr)
            if m2:
                print("[e2e]    " + m2.group(0), flush=True)


if __name__ == "__main__":
    main()
