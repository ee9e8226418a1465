#!/usr/bin/env python3
"""End-to-end throughput of the stand-alone `bwa-amd mem` (FASTQ in -> SAM out) on the bench genome; GPU box only.

usage: e2e_bench.py [--genome-mbp 3100] [--reads 1000000] [--threads 16] [--streams 3] [--pe] [--env "NAME=VALUE ...;NAME=VALUE ..."]
Prints one line per configuration (threads x streams x environment set): wall reads/s as reported by the program after the index was
loaded, its stage busy times, and the device stage's mean step times per batch."""
import argparse, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bwa_amd import simdata


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--threads", default="16")
    ap.add_argument("--streams", default="3")
    ap.add_argument("--env", default="", help="environment sets to compare, separated by ';' (an empty set = defaults)")
    ap.add_argument("--chunk", default="100000000")
    ap.add_argument("--pe", action="store_true")
    ap.add_argument("--hold", type=int, default=0, help="1: this (parent) process first opens a device context the way bench.py has one by the time it runs the command line -- torch.cuda, a handle on the index, closed again -- and keeps it; 2: ... and keeps the handle open")
    ap.add_argument("--cache", default=os.environ.get("BWA_AMD_CACHE", "/tmp/bwa_amd_bench"))
    a = ap.parse_args()
    import torch
    fa, g, _ = bench.build_or_load_index(a.genome_mbp, a.cache, 0, lambda: torch.cuda.synchronize())
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
    held = None
    if a.hold:
        from bwa_amd.api import BwaGpu
        torch.zeros(1, device="cuda")
        held = BwaGpu(fa)
        held.densify_sa(1)
        if a.hold == 1:
            held.close(); held = None
        print(f"[e2e] parent holds a device context (--hold {a.hold})", flush=True)
    cli = os.path.join(ROOT, "bwa_amd", "bwa-amd")
    for th in a.threads.split(","):
        for st in a.streams.split(","):
            for es in a.env.split(";"):
                env = dict(os.environ, BWAGPU_CLI_STREAMS=st, BWAGPU_CLI_TRACE="1")
                env.update(dict(kv.split("=", 1) for kv in es.split()))
                t = time.time()
                p = subprocess.run([cli, "mem", "-t", th, "-K", a.chunk, "-v", "3", fa] + files, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
                m = re.search(r"\[M::main_mem\] (\d+) reads in ([\d.]+) sec .*: (\d+) reads/s", p.stderr)
                print(f"[e2e] {'PE' if a.pe else 'SE'} -t {th} streams {st} [{es.strip() or 'defaults'}]: rc={p.returncode} {m.group(0) if m else p.stderr[-300:]}  (process wall {time.time() - t:.1f}s)", flush=True)
                m2 = re.search(r"stage busy time: .*", p.stderr)
                if m2:
                    print("[e2e]    " + m2.group(0), flush=True)
                steps = re.findall(r"upload ([\d.]+) run ([\d.]+) download\+cigars ([\d.]+) pestat\+matesw ([\d.]+) s", p.stderr)
                after = re.findall(r"after the hot path, ms: pack ([\d.]+) download copy ([\d.]+) cigar kernels ([\d.]+) cigar copies ([\d.]+)", p.stderr)
                for ln in re.findall(r"\[D::device_sub\] \d+ reads.*", p.stderr)[:6]:
                    print("[e2e]    " + ln, flush=True)
                for ln in re.findall(r"\[D::timeline\].*", p.stderr):
                    print("[e2e]    " + ln, flush=True)
                for ln in re.findall(r"\[D::device_sub\] stage ms:.*", p.stderr):
                    print("[e2e]    " + ln, flush=True)
                if after:
                    print("[e2e]    after the hot path (ms per batch, HIP events, mean of %d): pack %.1f, download copy %.1f, CIGAR kernels %.1f, CIGAR copies %.1f" %
                          ((len(after),) + tuple(sum(float(x[i]) for x in after) / len(after) for i in range(4))), flush=True)
                if steps:
                    k = len(steps)
                    print("[e2e]    device stage per batch (ms, mean of %d): upload %.1f, hot path %.1f, download + CIGARs %.1f, pestat + mate rescue %.1f" %
                          ((k,) + tuple(1e3 * sum(float(x[i]) for x in steps) / k for i in range(4))), flush=True)


if __name__ == "__main__":
    main()
