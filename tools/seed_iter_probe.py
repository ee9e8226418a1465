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
opt = default_opt(); opt.flag |= 2
r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000)
rd = bench.interleave(r1, r2)
flat = np.ascontiguousarray(rd.reshape(-1)); off = np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150
# every argument is one configuration: space-separated NAME=VALUE environment settings ("" = defaults)
for cfg in (sys.argv[1:] or [""]):
    sets = dict(kv.split("=", 1) for kv in cfg.split())
    for k, v in sets.items():
        os.environ[k] = v
    gpu = BwaGpu(prefix); gpu.densify_sa(4); gpu.set_taps(False)
    gpu.upload(flat, off)
    gpu.set_stats(False); gpu.run(opt)
    runs = []
    for _ in range(4):
        gpu.run(opt); runs.append(gpu.stats())
    ms_plain = {k: min(r[k] for r in runs) for k in ("ms_seed", "ms_chain", "ms_extend", "ms_dedup", "ms_total")}
    gpu.set_stats(True); gpu.run(opt)
    s = gpu.stats()
    out = (C.c_ulonglong * 16)()
    gpu.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
    gpu.L.bwagpu_debug_prof(gpu.h, out)
    it, slow, ext, deep = out[13], out[14], out[15], out[12]
    print(f"[{cfg or 'defaults'}] k_seed {ms_plain['ms_seed']:.1f} ms (with counters {s['ms_seed']:.1f}), chain {ms_plain['ms_chain']:.1f} extend {ms_plain['ms_extend']:.1f} dedup {ms_plain['ms_dedup']:.1f} total {ms_plain['ms_total']:.1f}: "
          f"wave iterations {it:.4g}, reading the stack from HBM {deep:.4g} ({100.0 * deep / max(it, 1):.1f}%), with bookkeeping {slow:.4g} ({100.0 * slow / max(it, 1):.1f}%), "
          f"lanes extending per iteration {ext / max(it, 1):.1f} of 64; lane steps {s['n_occ_blocks']} blocks + {s['n_tab_lookups']} table look-ups; regs {s['n_regs']}; "
          f"extension calls {s['n_ext_calls']} (diagonal rule {s['n_ext_fast']}); cells {s['n_ext_cells']}", flush=True)
    gpu.close()
    for k in sets:
        del os.environ[k]
