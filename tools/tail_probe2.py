#!/usr/bin/env python3
"""Diagnostics (GPU): how the lane-per-read kernels scale with the number of (identical / different) heavy reads in a wave."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
prefix, g, _ = bench.build_or_load_index(3100.0, "/tmp/bwa_amd_bench", 0, lambda: None)
gpu = BwaGpu(prefix); gpu.densify_sa(4)
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000)
rd = bench.interleave(r1, r2); L = 150
def run(reads, label, taps=False, stats=False):
    gpu.set_taps(taps); gpu.set_stats(stats)
    gpu.upload(np.ascontiguousarray(reads.reshape(-1)), np.arange(0, reads.shape[0] + 1, dtype=np.int64) * L)
    gpu.run(opt); gpu.run(opt)
    s = gpu.stats()
    print(f"{label:34s} n={reads.shape[0]:8d} seed {s['ms_seed']:7.2f} chain {s['ms_chain']:7.2f} ext {s['ms_extend']:7.2f} dedup {s['ms_dedup']:6.2f}" + (f"  N_blk {s['n_occ_blocks']} tab {s['n_tab_lookups']} seeds {s['n_seeds']} chains {s['n_chains']} regs_raw {s['n_regs_raw']} ext_calls {s['n_ext_calls']} ext_cells {s['n_ext_cells']}" if stats else ""), flush=True)
    return s
run(rd, "whole", taps=True)
n_iv, iv = gpu.tap_intervals()
x2 = iv["x2"].astype(np.int64); occ = opt.max_occ
step = np.where(x2 > occ, x2 // occ, 1); cnt = np.minimum((x2 + step - 1) // step, occ)
ns = np.bincount(np.repeat(np.arange(rd.shape[0]), n_iv), weights=cnt, minlength=rd.shape[0]).astype(np.int64)
order = np.argsort(ns, kind="stable")
top = rd[order[::-1][:64]]
for k in (1, 2, 4, 8, 16, 32, 64):
    run(top[:k], f"{k} heaviest (different) reads", stats=(k in (1, 64)))
for k in (2, 8, 64, 256, 4096):
    run(np.tile(top[:1], (k, 1)), f"{k} copies of the heaviest read")
med = rd[order[len(order) // 2]][None, :]
for k in (1, 64, 4096, 65536):
    run(np.tile(med, (k, 1)), f"{k} copies of a median read", stats=(k == 1))
# per-read stage times of each of the 64 heaviest, alone
for j in range(0, 64, 4):
    run(top[j:j + 1], f"heavy read #{j} alone (ns={int(ns[order[::-1][j]])})", stats=True)
