#!/usr/bin/env python3
"""What k_dedup's reads look like: the headline's batch (1 M paired-end reads of 150 bp on the 3.1 Gbp stand-in), regions per read before and after
mem_sort_dedup_patch, and the share of the regions that sit in reads of each size (tools/gpu_session.sh runs it; index and genome from bench.py's cache)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prefix", required=True)
    ap.add_argument("--codes", required=True)
    ap.add_argument("--reads", type=int, default=1_000_000)
    args = ap.parse_args()
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    g = np.load(args.codes, mmap_mode="r")
    r1, r2 = simdata.make_reads_pe(g, args.reads // 2, length=150, seed=1000)
    rd = np.empty((2 * r1.shape[0], 150), dtype=np.uint8); rd[0::2] = r1; rd[1::2] = r2
    opt = default_opt(); opt.flag |= 0x2
    gpu = BwaGpu(args.prefix)
    gpu.set_taps(True); gpu.set_stats(True)
    counts, regs = gpu.align(opt, np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150)
    n_raw = gpu.tap_regs_raw()[0]
    st = gpu.stats()
    edges = [0, 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 65, 129, 1 << 30]
    out = {"reads": int(n_raw.shape[0]), "regions_raw": int(n_raw.sum()), "regions_kept": int(counts.sum()), "patch_alignments": int(st.get("glb_calls", -1)), "patch_cells": int(st.get("glb_cells", -1)), "by_raw_regions": []}
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (n_raw >= lo) & (n_raw < hi)
        out["by_raw_regions"].append({"n": f"{lo}..{hi - 1}" if hi < (1 << 30) else f"{lo}+", "reads": int(m.sum()), "regions": int(n_raw[m].sum()), "pairs_n2": int((n_raw[m].astype(np.int64) ** 2).sum())})
    w = n_raw.reshape(-1, 64)
    out["wave_of_64_reads"] = {"mean_of_max": float(w.max(axis=1).mean()), "mean_of_mean": float(w.mean()), "mean_of_sum": float(w.sum(axis=1).mean()), "p99_of_max": float(np.percentile(w.max(axis=1), 99))}
    import ctypes as C
    hist = (C.c_ulonglong * 256)()
    gpu.L.bwagpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
    gpu.L.bwagpu_debug_hist(gpu.h, hist)
    out["wave_kernel_reads_by_time"] = [{"us": f"{(1 << (b - 1)) / 100 if b else 0:.2f}..{(1 << b) / 100:.2f}", "reads": int(hist[160 + b]), "patch_calls": int(hist[192 + b]), "kcells": int(hist[224 + b])} for b in range(32) if hist[160 + b]]
    out["stats"] = {k: v for k, v in st.items() if "glb" in k or "dedup" in k or "regs" in k}
    print(json.dumps(out))
    gpu.close()


if __name__ == "__main__":
    main()
