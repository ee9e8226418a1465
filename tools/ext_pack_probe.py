#!/usr/bin/env python3
"""Diagnostics (GPU): the extension stage with and without the packed first extensions (option ext_pack: 0 = every extension by a wavefront of its own,
1 = k_ext_pack at 4 waves per SIMD, 5 = at 5): solo stage times of one batch of the bench's reads, the digest of the regions (must not move), and how many
of the stage's ksw_extend2 calls k_ext_pack answered (bwagpu_debug_prof[8]) against the calls the one-wave path makes (stats run)."""
import ctypes as C, hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
prefix, g, _ = bench.build_or_load_index(float(os.environ.get("MBP", "3100")), "/tmp/bwa_amd_bench", 0, lambda: None)
opt = default_opt(); opt.flag |= 2
n = int(os.environ.get("READS", "1000000"))
r1, r2 = simdata.make_reads_pe(g, n // 2, seed=1000)
rd = bench.interleave(r1, r2)
gpu = BwaGpu(prefix, lib_path=os.environ.get("LIB") or None); gpu.densify_sa(1)      # (LIB: e.g. a -DBWAGPU_FAKE_DP build, whose extension stage is control only); gpu.set_taps(False)
gpu.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
gpu.upload(np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * rd.shape[1])
gpu.set_stats(True); gpu.run(opt); st = gpu.stats(); gpu.set_stats(False)
print(f"stats run: ext_calls {st['n_ext_calls']} ({st['n_ext_calls'] / n:.2f} per read), answered by the diagonal rule {st['n_ext_fast']}, cells {st['n_ext_cells'] / n:.0f} per read, regions (raw) {st['n_regs_raw'] / n:.2f} per read", flush=True)
base = None
for cfg in (sys.argv[1:] or ["0", "1", "5", "0", "1"]):
    gpu.set_option("ext_pack", int(cfg))
    gpu.run(opt)
    runs = []
    for _ in range(3):
        gpu.run(opt); runs.append(gpu.stats())
    prof = (C.c_ulonglong * 16)()
    gpu.L.bwagpu_debug_prof(gpu.h, prof)
    counts, regs = gpu.download()
    dig = hashlib.sha256(counts.tobytes() + regs.tobytes()).hexdigest()[:16]
    base = base or dig
    ms = {k: round(min(r[k] for r in runs), 2) for k in ("ms_chain", "ms_extend", "ms_dedup", "ms_total")}
    print(f"ext_pack={cfg}: {ms}  answered {prof[8]} ({prof[8] / max(1, st['n_ext_calls']):.3f} of the calls)  digest {dig} {'same' if dig == base else 'DIFFERENT'}", flush=True)
