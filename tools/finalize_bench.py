#!/usr/bin/env python3
"""CPU-only timing of the host finalize stage (regions -> SAM text) as `bwa-amd mem` runs it: regions from the compiled reference's
mem_align1_core, CIGAR/NM/MD and mate-rescue records from the host restatement of the device kernels (the same records the device
delivers), then bwamem_host_regs2sam on N threads.  usage: finalize_bench.py [n_pairs] [threads] [repeats]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostapi, refapi, testdata
from bwa_amd import simdata
from bwa_amd.structs import default_opt

n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
fa, g = testdata.medium_index()
ref, host = refapi.RefIndex(fa), hostapi.HostFinalize(fa)
opt = default_opt(); opt.flag |= 2      # MEM_F_PE
r1, r2 = simdata.make_reads_pe(g, n_pairs, seed=77)
rd = np.empty((2 * n_pairs, r1.shape[1]), dtype=np.uint8); rd[0::2] = r1; rd[1::2] = r2
seqs, off = testdata.flat(rd)
cache = os.path.join(os.environ.get("BWA_AMD_CACHE", "/tmp"), f"finalize_bench_{n_pairs}.npz")    # (the inputs take 20 s to make: kept between runs)
if os.path.exists(cache):
    z = np.load(cache); counts, regs, cigs, ops, pes, msw = (z[k] for k in ("counts", "regs", "cigs", "ops", "pes", "msw"))
    regs, cigs, msw, pes = regs.view(hostapi.REG_DTYPE) if hasattr(hostapi, "REG_DTYPE") else regs, cigs, msw, pes
else:
    t = time.time(); counts, regs = ref.align(opt, seqs, off); print(f"regions: {regs.shape[0] / (2 * n_pairs):.2f} per read ({time.time() - t:.1f}s)", flush=True)
    t = time.time(); cigs, ops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True); print(f"cigar records {time.time() - t:.1f}s", flush=True)
    pes = host.pestat(opt, counts, regs)
    t = time.time(); msw = host.matesw_records(opt, seqs, off, counts, regs, pes); print(f"mate-rescue records: {msw.shape[0]} ({time.time() - t:.1f}s)", flush=True)
    np.savez(cache, counts=counts, regs=regs, cigs=cigs, ops=ops, pes=pes, msw=msw)
names = [f"q{i >> 1}" for i in range(2 * n_pairs)]
quals = bytes((33 + (np.arange(seqs.shape[0]) % 40)).astype(np.uint8))
import ctypes as C
pes_buf = pes.ctypes.data_as(C.c_void_p)
for variant, kw in (("hints: cigars + mate rescue, pestat given", dict(cigs=cigs, cig_ops=ops, msw=msw, pes0=pes_buf)), ("hints: cigars + mate rescue", dict(cigs=cigs, cig_ops=ops, msw=msw)), ("no hints", {})):
    best = None
    for _ in range(reps if not (os.environ.get("FB_FIRST_ONLY") and variant != "hints: cigars + mate rescue, pestat given") else 0):
        mc = C.CDLL(os.environ["MC_LIB"]) if os.environ.get("MC_LIB") else None      # (optional: a malloc-counting preload library)
        if mc: mc.mc_count.restype = C.c_ulong; m0 = mc.mc_count()
        t = time.time(); sam = host.regs2sam(opt, names, seqs, quals, off, counts, regs, n_threads=threads, **kw); dt = time.time() - t
        best = dt if best is None or dt < best else best
        if mc: print(f"  mallocs per read: {(mc.mc_count() - m0) / (2 * n_pairs):.2f}")
    if os.environ.get("FB_FIRST_ONLY") and variant != "hints: cigars + mate rescue, pestat given": continue
    print(f"{variant}: {best:.3f} s on {threads} threads = {best / (2 * n_pairs) * 1e6:.3f} us/read wall, {len(sam) / 1e6:.0f} MB of SAM", flush=True)
