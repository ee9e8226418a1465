#!/usr/bin/env python3
"""Diagnostics (GPU): cycle counters of the profiling build of k_chain_wave (libbwagpu_prof.so, -DBWAGPU_PROFILE)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
prefix, g, _ = bench.build_or_load_index(3100.0, "/tmp/bwa_amd_bench", 0, lambda: None)
gpu = BwaGpu(prefix, lib_path=os.path.join(ROOT, "bwa_amd", "csrc", "libbwagpu_prof.so")); gpu.densify_sa(4); gpu.set_taps(False)
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000)
rd = bench.interleave(r1, r2)
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150)
gpu.run(opt); gpu.run(opt)
s = gpu.stats()
out = (C.c_ulonglong * 16)()
gpu.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
gpu.L.bwagpu_debug_prof(gpu.h, out)
names = ["0 initial loads", "1 storage setup + seed cache fill", "2 chunk load of seeds", "3 readlanes", "4 lower()", "5 record + merge", "6 new chain + insert",
         "7 frac_rep", "8 inorder", "9 weights", "10 sort+filter+kept", "11 publish", "12 reservation"]
tot = sum(out)
print(f"chain {s['ms_chain']:.2f} ms; total counted cycles {tot:.3e} (100 MHz constant clock: {tot / 1e8 * 1e3:.0f} wave-ms)")
for i, nme in enumerate(names):
    print(f"  {nme:36s} {out[i]:14d}  {100.0 * out[i] / max(tot, 1):5.1f}%")
print("seeds", s["n_seeds"], "deferred", s["n_chain_deferred"], s["n_chain_deferred2"])
