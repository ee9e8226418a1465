#!/usr/bin/env python3
"""Diagnostics (GPU): wave iterations of k_seed, how many ran the bookkeeping code, lanes extending per iteration."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
prefix, g, _ = bench.build_or_load_index(float(os.environ.get("MBP", "3100")), "/tmp/bwa_amd_bench", 0, lambda: None)
gpu = BwaGpu(prefix); gpu.densify_sa(4); gpu.set_taps(False)
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000)
rd = bench.interleave(r1, r2)
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150)
gpu.set_stats(True); gpu.run(opt)
s = gpu.stats()
out = (C.c_ulonglong * 16)()
gpu.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
gpu.L.bwagpu_debug_prof(gpu.h, out)
it, slow, ext = out[13], out[14], out[15]
print(f"k_seed {s['ms_seed']:.1f} ms: wave iterations {it:.4g}, with bookkeeping {slow:.4g} ({100.0 * slow / it:.1f}%), lanes extending per iteration {ext / it:.1f} of 64; "
      f"lane steps {s['n_occ_blocks']} blocks + {s['n_tab_lookups']} table look-ups")
