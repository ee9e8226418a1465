#!/usr/bin/env python3
"""A/B of library options on one resident long-read batch, alternating within one process: tools/longread_ab.py [--reads 6000] [--rounds 4] "opt=val ..." "opt=val ..." ...
Prints min / median of the named stage times per configuration (the long-read kernels vary by +-8 % from run to run: single runs cannot be compared)."""
import argparse, os, sys, statistics
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import pacbio_opt
ap = argparse.ArgumentParser()
ap.add_argument("configs", nargs="+")
ap.add_argument("--reads", type=int, default=6000)
ap.add_argument("--read-len", type=int, default=10000)
ap.add_argument("--rounds", type=int, default=4)
a = ap.parse_args()
fa, g, _ = bench.build_or_load_index(3100.0, "/tmp/bwa_amd_bench", 0, lambda: None)
gpu = BwaGpu(fa); gpu.densify_sa(1); gpu.set_taps(False)
rd = simdata.make_reads_long(g, a.reads, length=a.read_len, seed=7)
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, a.reads + 1, dtype=np.int64) * a.read_len)
opt = pacbio_opt()
gpu.run(opt)
res = {c: [] for c in a.configs}
for _ in range(a.rounds):
    for c in a.configs:
        sets = dict(kv.split("=", 1) for kv in c.split())
        old = {k: gpu.get_option(k) for k in sets}
        for k, v in sets.items():
            gpu.set_option(k, int(v))
        gpu.run(opt); res[c].append(gpu.stats())
        for k, v in old.items():
            gpu.set_option(k, v)
for c in a.configs:
    out = []
    for key in ("ms_seed", "ms_extend", "ms_dedup", "ms_total"):
        v = [s[key] for s in res[c]]
        out.append(f"{key[3:]} min {min(v):.0f} med {statistics.median(v):.0f}")
    print(f"[{a.reads} reads] {c or 'defaults'}: " + "; ".join(out), flush=True)
gpu.close()
