#!/usr/bin/env python3
"""GPU-box measurement of the long-read path (BASELINE configs[4] shape: 10 kb reads, -x pacbio) -- a parity-test
configuration, not the bench line; prints reads/s and the stage times of one batch."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import pacbio_opt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-mbp", type=float, default=3100.0)
    ap.add_argument("--reads", type=int, default=20000)
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--cigars", action="store_true")
    ap.add_argument("--options", default="", help="library options for this run, e.g. 'ext_phased=0'")
    ap.add_argument("--passes", type=int, default=1, help="timed passes (the best is reported)")
    ap.add_argument("--cache", default=os.environ.get("BWA_AMD_CACHE", "/tmp/bwa_amd_bench"))
    a = ap.parse_args()
    import torch
    fa, g, _ = bench.build_or_load_index(a.genome_mbp, a.cache, 0, lambda: torch.cuda.synchronize())
    gpu = BwaGpu(fa); gpu.densify_sa(1); gpu.set_taps(False)
    for kv in a.options.split():
        k, v = kv.split("=", 1); gpu.set_option(k, int(v))
    rd = simdata.make_reads_long(g, a.reads, length=a.read_len, seed=7)      # SURVEY 8d's PacBio-like model: 1.5 % sub, 4 % del, 9 % ins
    off = np.arange(0, a.reads + 1, dtype=np.int64) * a.read_len
    opt = pacbio_opt()
    gpu.upload(np.ascontiguousarray(rd.reshape(-1)), off)
    gpu.set_stats(True)
    t = time.time(); gpu.run(opt); dt0 = time.time() - t
    w = gpu.stats()
    print(f"[longread] first pass {dt0:.2f}s (retries {w['n_retries']}); per read: intervals {w['n_intv'] / a.reads:.0f}, seeds {w['n_seeds'] / a.reads:.0f}, chains kept {w['n_chains'] / a.reads:.1f}, "
          f"seed-SW calls {w['n_sw_calls'] / a.reads:.0f} ({w['n_sw_cells'] / a.reads / 1e6:.2f} M cells), extension calls {w['n_ext_calls'] / a.reads:.1f} ({w['n_ext_cells'] / a.reads / 1e6:.2f} M cells), "
          f"patch alignments {w['n_glb_calls'] / a.reads:.1f} ({w['n_glb_cells'] / a.reads / 1e6:.2f} M cells), regions {w['n_regs_raw'] / a.reads:.1f} -> {w['n_regs'] / a.reads:.1f}; "
          "stage ms: " + ", ".join(f"{k[3:]} {w[k]:.0f}" for k in ("ms_seed", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup")), flush=True)
    import ctypes as C
    hist = (C.c_ulonglong * 256)()
    gpu.L.bwagpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
    gpu.L.bwagpu_debug_hist(gpu.h, hist)
    for name, base in (("k_extend_wave", 64), ("k_dedup_wave", 160)):      # reads by the time their wave spent on them, with the DP calls and cells of each class
        rows = [(b, hist[base + b], hist[base + 32 + b], hist[base + 64 + b]) for b in range(32) if hist[base + b]]
        print(f"[longread] {name}, reads by wave time: " + "; ".join(f"<{(1 << b) / 1e5:.3g} ms: {n} reads, {c / n:.1f} DP calls and {x * 1024 / n / 1e6:.2f} M cells each" for b, n, c, x in rows), flush=True)
    gpu.set_stats(False)
    dt, st = 1e9, None
    for _ in range(max(1, a.passes)):
        t = time.time(); gpu.run(opt); d_ = time.time() - t
        if d_ < dt:
            dt, st = d_, gpu.stats()
    counts, regs = gpu.download()
    if a.cigars:       # the CIGAR stage as `bwa-amd mem` drives it; BWAGPU_CIG_TRACE=1 splits it into its launches on stderr
        for k in range(2):
            t = time.time(); cg = gpu.cigars(opt); ops = gpu.cigar_ops(); dtc = time.time() - t
            served = int((cg["n_cigar"] >= 0).sum())
            print(f"[longread] cigars pass {k}: {dtc:.2f}s for {cg.shape[0]} regions ({served} with a device CIGAR, {ops.shape[0] / 1e6:.1f} M extension entries)", flush=True)
    print(f"[longread] {a.reads} x {a.read_len} bp -x pacbio: first pass {dt0:.2f}s, second {dt:.2f}s -> {a.reads / dt:.0f} reads/s ({a.reads * a.read_len / dt / 1e6:.1f} Mbp/s); "
          f"regions {regs.shape[0]}, retries {st['n_retries']}, stage ms: " + ", ".join(f"{k[3:]} {st[k]:.0f}" for k in ("ms_seed", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup")), flush=True)


if __name__ == "__main__":
    main()
