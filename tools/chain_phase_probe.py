#!/usr/bin/env python3
"""Diagnostics (GPU): time of k_chain_wave when every read returns after phase k (BWAGPU_CHAIN_STOP)."""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import bench
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    prefix, g, _ = bench.build_or_load_index(3100.0, "/tmp/bwa_amd_bench", 0, lambda: None)
    gpu = BwaGpu(prefix); gpu.densify_sa(4); gpu.set_taps(False)
    opt = default_opt(); opt.flag |= 2
    r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000)
    rd = bench.interleave(r1, r2)
    gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150)
    for _ in range(2):
        try:
            gpu.run(opt)
        except Exception as e:
            print("run failed", e); break
    s = gpu.stats()
    print(f"stop={os.environ.get('BWAGPU_CHAIN_STOP', '0')} lane={os.environ.get('BWAGPU_CHAIN_LANE', '-')}: chain {s['ms_chain']:.2f} ms  (seed {s['ms_seed']:.1f})", flush=True)
else:
    for stop in ("1", "2", "3", "4", "5", "0"):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, BWAGPU_CHAIN_STOP=stop))
