#!/usr/bin/env python3
"""Diagnostics (GPU): stage times of `bwa-amd mem` on paired-end FASTQ (BWAGPU_CLI_TRACE prints the device stage's steps)."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
mbp = float(os.environ.get("MBP", "512"))
n_pairs = int(os.environ.get("PAIRS", "2000000"))
prefix, g, _ = bench.build_or_load_index(mbp, "/tmp/bwa_amd_bench", 0, lambda: None)
r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=77)
f1, f2 = "/tmp/bwa_amd_bench/e1.fq", "/tmp/bwa_amd_bench/e2.fq"
simdata.write_fastq(f1, r1, suffix="/1"); simdata.write_fastq(f2, r2, suffix="/2")
for streams in os.environ.get("STREAMS", "2,3").split(","):
    env = dict(os.environ, BWAGPU_CLI_STREAMS=streams, BWAGPU_CLI_TRACE="1")
    t = time.time()
    p = subprocess.run([os.path.join(ROOT, "bwa_amd", "bwa-amd"), "mem", "-t", "16", "-K", "100000000", "-v", "3", prefix, f1, f2], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    lines = [l for l in p.stderr.split("\n") if "device_sub" in l or "main_mem" in l]
    print(f"--- streams={streams} wall {time.time() - t:.1f}s rc={p.returncode}")
    print("\n".join(lines[:4] + lines[-3:]))
