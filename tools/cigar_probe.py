#!/usr/bin/env python3
"""Diagnostics (GPU): the CIGAR stage alone (bwagpu_batch_cigars: k_cigar's two LDS tiers, k_cigar_long) on one resident batch of the bench workload --
its kernels' time by the library's HIP events, the DP cells of its fills, how many regions each path served.  READS=666668 python tools/cigar_probe.py ["opt=val ..."]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
prefix, g, _ = bench.build_or_load_index(float(os.environ.get("MBP", "3100")), "/tmp/bwa_amd_bench", 0, lambda: None)
n = int(os.environ.get("READS", "666668")) // 2 * 2
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, n // 2, seed=1000)
rd = bench.interleave(r1, r2)
gpu = BwaGpu(prefix); gpu.densify_sa(1); gpu.set_taps(False)
gpu.L.bwagpu_set_cigar_filter.argtypes = [__import__("ctypes").c_void_p, __import__("ctypes").c_int]
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, n + 1, dtype=np.int64) * rd.shape[1])
gpu.run(opt); gpu.run(opt)
counts, regs = gpu.download()
for cfg in (sys.argv[1:] or [""]):
    for kv in cfg.split():
        k, v = kv.split("=", 1); gpu.set_option(k, int(v))
    for filt in (1, 0):
        gpu.L.bwagpu_set_cigar_filter(gpu.h, filt)
        gpu.set_stats(True)
        cg = gpu.cigars(opt); st = gpu.stats()
        gpu.set_stats(False)
        ms = []
        for _ in range(3):
            t = time.perf_counter(); cg = gpu.cigars(opt); wall = (time.perf_counter() - t) * 1e3
            s2 = gpu.stats(); ms.append((s2["ms_cigar_kernels"], s2["ms_cigar_copy"], wall))
        best = min(ms)
        served = int((cg["n_cigar"] > 0).sum()); skipped = int(((cg["n_cigar"] < 0) & (cg["score"] == 1)).sum()); left = int(((cg["n_cigar"] < 0) & (cg["score"] != 1)).sum())
        cells, ndp = st["n_cig_cells"], st["n_cig_dp"]
        print(f"[{cfg or 'defaults'}] filter {filt}: {n} reads, {regs.shape[0]} regions: served {served}, filtered {skipped}, left to the host {left}; kernels {best[0]:.2f} ms, copy {best[1]:.2f} ms (wall {best[2]:.1f}); "
              f"{ndp} fills, {cells / max(n, 1):.0f} cells per read, {cells / max(ndp, 1):.0f} per fill, {cells / (best[0] * 1e-3) / 1e9:.1f} GCUPS", flush=True)
gpu.close()
