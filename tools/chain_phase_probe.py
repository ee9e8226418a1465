#!/usr/bin/env python3
"""Diagnostics (GPU): one solo batch of the benchmark workload (two passes); run as `chain_phase_probe.py x` under rocprofv3 for per-kernel
times and counters (tools/profile_round.sh uses bench.py instead)."""
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
    print(f"seed {s['ms_seed']:.1f} chain {s['ms_chain']:.2f} extend {s['ms_extend']:.1f} dedup {s['ms_dedup']:.1f} ms", flush=True)
else:
    subprocess.run([sys.executable, __file__, "x"])
