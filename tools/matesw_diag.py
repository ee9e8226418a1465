#!/usr/bin/env python3
"""GPU-box check of bwagpu_batch_matesw against the host code (records as sets) + timing; noisy second mates."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import testdata, hostapi
from bwa_amd import simdata
from bwa_amd.api import BwaGpu, MATESW_DTYPE, PES_DTYPE
from bwa_amd.structs import default_opt
fa, g = testdata.medium_index()
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=98)
rng = np.random.default_rng(99)
r2 = np.where(rng.random(r2.shape) < 0.10, (r2 + rng.integers(1, 4, r2.shape)) % 4, r2).astype(np.uint8)
reads = np.empty((2 * n_pairs, r1.shape[1]), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
seqs, off = testdata.flat(reads)
opt = default_opt(); opt.flag |= 2
gpu, host = BwaGpu(fa), hostapi.HostFinalize(fa)
counts, regs = gpu.align(opt, seqs, off)
pes = host.pestat(opt, counts, regs)
dpes = np.zeros(4, dtype=PES_DTYPE)
for k in ("low", "high", "failed"):
    dpes[k] = pes[k]
t = time.time(); got = gpu.matesw(opt, dpes); dt = time.time() - t
t = time.time(); want = host.matesw_records(opt, seqs, off, counts, regs, pes); dth = time.time() - t
key = lambda a: np.sort(np.frombuffer(a.tobytes(), dtype=f"V{MATESW_DTYPE.itemsize}"))
same = got.shape == want.shape and bool((key(got) == key(want)).all())
print(f"[matesw] {n_pairs} pairs: {got.shape[0]} tasks ({int((got['r'] >= 0).sum())} alignments), device {dt * 1e3:.1f} ms, host 1 thread {dth * 1e3:.0f} ms, records identical: {same}", flush=True)
