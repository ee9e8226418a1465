#!/usr/bin/env python3
"""Diagnostics (GPU): where do the solo kernel times of one batch go -- bulk throughput or the heaviest reads?
Runs the benchmark batch (1 M reads, PE layout) whole, then its lightest 99 % / heaviest 1 % / heaviest 0.1 % alone,
and prints per-stage device times plus the distribution of seeds (SA look-ups) per read."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt

ap = argparse.ArgumentParser()
ap.add_argument("--genome-mbp", type=float, default=3100.0)
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--cache", default="/tmp/bwa_amd_bench")
a = ap.parse_args()
prefix, g, _ = bench.build_or_load_index(a.genome_mbp, a.cache, 0, lambda: None)
gpu = BwaGpu(prefix); gpu.densify_sa(4)
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, a.reads // 2, seed=1000)
rd = bench.interleave(r1, r2)
L = rd.shape[1]

def run(reads, label, taps=False):
    gpu.set_taps(taps); gpu.set_stats(False)
    gpu.upload(np.ascontiguousarray(reads.reshape(-1)), np.arange(0, reads.shape[0] + 1, dtype=np.int64) * L)
    gpu.run(opt); gpu.run(opt)
    s = gpu.stats()
    print(f"{label:28s} n={reads.shape[0]:8d} seed {s['ms_seed']:7.2f} publish {s['ms_publish']:5.2f} sa {s['ms_sa']:5.2f} chain {s['ms_chain']:7.2f} ext {s['ms_extend']:7.2f} dedup {s['ms_dedup']:6.2f} total {s['ms_total']:7.2f}", flush=True)
    return s

run(rd, "whole batch", taps=True)
n_iv, iv = gpu.tap_intervals()
x2 = iv["x2"].astype(np.int64)
occ = opt.max_occ
step = np.where(x2 > occ, x2 // occ, 1)
cnt = np.minimum((x2 + step - 1) // step, occ)
owner = np.repeat(np.arange(rd.shape[0]), n_iv)
ns = np.bincount(owner, weights=cnt, minlength=rd.shape[0]).astype(np.int64)
q = [50, 90, 99, 99.9, 99.99, 100]
print("seeds/read percentiles", dict(zip(q, np.percentile(ns, q).astype(int).tolist())), "mean", ns.mean(), "sum", ns.sum())
print("intervals/read percentiles", dict(zip(q, np.percentile(n_iv, q).astype(int).tolist())), "mean", n_iv.mean())
order = np.argsort(ns, kind="stable")
n = rd.shape[0]
for frac, lab in ((0.99, "lightest 99%"), (0.999, "lightest 99.9%")):
    k = int(n * frac)
    run(rd[np.sort(order[:k])], lab)
for frac, lab in ((0.01, "heaviest 1%"), (0.001, "heaviest 0.1%"), (0.0001, "heaviest 0.01%")):
    k = max(1, int(n * frac))
    sel = np.sort(order[-k:])
    run(rd[sel], lab)
    print("   seeds in this subset:", int(ns[sel].sum()), "max", int(ns[sel].max()))
run(rd[order[-1:]], "the heaviest read alone")
run(rd[order[-64:]], "64 heaviest reads")
