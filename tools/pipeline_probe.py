#!/usr/bin/env python3
"""Diagnostics (GPU): ms per step of the pipelined hot path (S batches in flight, as bench.py times it) for several environment
configurations in one process.  usage: pipeline_probe.py [--streams S] [--steps K] "NAME=VALUE ..." ..."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
argv = sys.argv[1:]
S, K = 3, 6
while argv and argv[0].startswith("--"):
    if argv[0] == "--streams": S = int(argv[1])
    if argv[0] == "--steps": K = int(argv[1])
    argv = argv[2:]
prefix, g, _ = bench.build_or_load_index(float(os.environ.get("MBP", "3100")), "/tmp/bwa_amd_bench", 0, lambda: None)
opt = default_opt(); opt.flag |= 2
gpu = BwaGpu(prefix)
_t = time.perf_counter(); gpu.densify_sa(4); print(f"[densify 32 -> 4] {time.perf_counter() - _t:.3f} s", flush=True)
gpu.set_taps(False)
handles = [gpu] + [gpu.clone() for _ in range(S - 1)]
for si, h in enumerate(handles):
    r1, r2 = simdata.make_reads_pe(g, 500_000, seed=1000 + si)
    rd = bench.interleave(r1, r2)
    h.set_taps(False)
    h.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * 150)

def worker(h, n):
    for _ in range(n):
        h.run(opt)

for cfg in (argv or [""]):
    sets = dict(kv.split("=", 1) for kv in cfg.split())
    for k, v in sets.items():
        os.environ[k] = v
    best = None
    for rep in range(2):
        th = [threading.Thread(target=worker, args=(h, 1)) for h in handles]
        [t.start() for t in th]; [t.join() for t in th]
        share = [K // S + (1 if i < K % S else 0) for i in range(S)]
        t0 = time.perf_counter()
        th = [threading.Thread(target=worker, args=(handles[i], share[i])) for i in range(S) if share[i]]
        [t.start() for t in th]; [t.join() for t in th]
        dt = (time.perf_counter() - t0) / K * 1e3
        best = dt if best is None or dt < best else best
    gpu.run(opt); st = gpu.stats()
    print(f"[{cfg or 'defaults'}] {best:.1f} ms/step over {S} batches in flight ({1e3 / best:.2f} Mreads/s); solo: seed {st['ms_seed']:.1f} chain {st['ms_chain']:.1f} extend {st['ms_extend']:.1f} dedup {st['ms_dedup']:.1f} total {st['ms_total']:.1f}", flush=True)
    for k in sets:
        del os.environ[k]
