#!/usr/bin/env python3
"""The workload of bench.py's in-run counter passes: ONE solo batch of the headline's reads through the hot path (index from bench.py's cache, no stats
instance, no timing), run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a second process, `--pmc WRITE_SIZE` (the TCC block cannot count both at
once, MI355X_MICROARCH.md).  usage: pmc_child.py --prefix P --batch reads.npy [--dense-sa 4]"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--prefix", required=True)
ap.add_argument("--batch", required=True)
ap.add_argument("--dense-sa", type=int, default=1)
ap.add_argument("--layout", default="pe")
a = ap.parse_args()
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
rd = np.load(a.batch)
gpu = BwaGpu(a.prefix)
if a.dense_sa:
    gpu.densify_sa(a.dense_sa)
gpu.set_taps(False)
opt = default_opt()
if a.layout == "pe":
    opt.flag |= 0x2
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * rd.shape[1])
gpu.run(opt)          # arenas learn their sizes (a first run may redo the batch)
gpu.run(opt)          # the launch the counters are read from (per kernel: the launch with the largest FETCH_SIZE, tools/pmc_summary.py)
print("pmc_child done", gpu.stats()["ms_total"], flush=True)
gpu.close()
