#!/usr/bin/env python3
"""Diagnostics (GPU): where k_seed's lane-slots go.  Per configuration (arguments: space-separated option settings, "" = defaults) and per
batch size (READS=250000,500000,1000000,2000000): the kernel's time without and with its work counters, wave iterations, lanes extending per
iteration, and the lane-slots that do NOT extend split into lanes that have run out of reads (the kernel's tail), lanes waiting in a bookkeeping
state for the wave to run that code, and lanes running it; the iteration at which a wave's first lane runs out of reads against the wave's length."""
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
sizes = [int(x) for x in os.environ.get("READS", "1000000").split(",")]
r1, r2 = simdata.make_reads_pe(g, max(sizes) // 2, seed=1000)
rd_all = bench.interleave(r1, r2)
gpu = BwaGpu(prefix); gpu.densify_sa(1); gpu.set_taps(False)
gpu.L.bwagpu_debug_prof.argtypes = [C.c_void_p, C.c_void_p]
for cfg in (sys.argv[1:] or [""]):
    sets = {}
    for kv in cfg.split():
        k, v = kv.split("=", 1); k = k.lower(); sets[k[7:] if k.startswith("bwagpu_") else k] = int(v)
    old = {k: gpu.get_option(k) for k in sets}
    for k, v in sets.items():
        gpu.set_option(k, v)
    for n in sizes:
        rd = rd_all[:n]
        flat = np.ascontiguousarray(rd.reshape(-1)); off = np.arange(0, rd.shape[0] + 1, dtype=np.int64) * rd.shape[1]
        gpu.upload(flat, off)
        gpu.set_stats(False); gpu.run(opt)
        runs = []
        for _ in range(3):
            gpu.run(opt); runs.append(gpu.stats())
        ms = {k: min(r[k] for r in runs) for k in ("ms_seed", "ms_publish", "ms_sa", "ms_chain", "ms_extend", "ms_dedup", "ms_total")}
        gpu.set_stats(True); gpu.run(opt)
        s = gpu.stats()
        out = (C.c_ulonglong * 16)()
        gpu.L.bwagpu_debug_prof(gpu.h, out)
        it, slow, ext, deep = out[13], out[14], out[15], out[12]
        done_l, wait_l, run_l, first_done, longest, waves = out[2], out[3], out[4], out[5], out[6], max(out[7], 1)
        slots = 64.0 * max(it, 1)
        print(f"[{cfg or 'defaults'}] {n} reads: k_seed(+k_seed3) {ms['ms_seed']:.1f} ms (with counters {s['ms_seed']:.1f}), sa {ms['ms_sa']:.1f} chain {ms['ms_chain']:.1f} extend {ms['ms_extend']:.1f} dedup {ms['ms_dedup']:.1f} "
              f"total {ms['ms_total']:.1f} | wave iterations {it:.4g} over {waves} waves (mean {it / waves:.0f}, longest {longest}; first lane out of reads at {first_done / waves:.0f} on average), "
              f"with bookkeeping {100.0 * slow / max(it, 1):.1f}%, stack from HBM {100.0 * deep / max(it, 1):.1f}% | lane-slots: extending {100.0 * ext / slots:.1f}% ({ext / max(it, 1):.1f} of 64), "
              f"out of reads {100.0 * done_l / slots:.1f}%, waiting for bookkeeping {100.0 * wait_l / slots:.1f}%, in bookkeeping {100.0 * run_l / slots:.1f}% | "
              f"{s['n_occ_blocks'] / n:.0f} blocks + {s['n_tab_lookups'] / n:.0f} table look-ups per read", flush=True)
        hist = (C.c_ulonglong * 256)()
        gpu.L.bwagpu_debug_hist.argtypes = [C.c_void_p, C.c_void_p]
        gpu.L.bwagpu_debug_hist(gpu.h, hist)
        tot_r = max(sum(hist[:32]), 1); tot_i = max(sum(hist[32:]), 1)
        ch = (C.c_ulonglong * 192)()
        gpu.L.bwagpu_debug_chain_hist.argtypes = [C.c_void_p, C.c_void_p]
        gpu.L.bwagpu_debug_chain_hist(gpu.h, ch)
        for t in range(2):      # row 0: reads chained in registers, row 1: in the B-tree
            tot = max(sum(ch[t * 64: t * 64 + 32]), 1)
            print(f"    chaining, {('register form', 'tree form')[t]}: {tot} reads; by chains (x16): " + " ".join(f"{b}:{ch[t * 64 + b]}" for b in range(32) if ch[t * 64 + b])
                  + " | by seeds (x32): " + " ".join(f"{b}:{ch[t * 64 + 32 + b]}" for b in range(32) if ch[t * 64 + 32 + b]), flush=True)
        ph = [ch[128 + i] for i in range(9)]
        names = ["register-form seed loop", "tree-form seed loop", "repeat fraction + in-order list", "weights", "sort", "pairwise filter", "publishing"]
        tot = max(sum(ph[:7]), 1)
        print("    chaining, wave time by phase: " + ", ".join(f"{nm} {100.0 * v / tot:.1f}%" for nm, v in zip(names, ph)) + f"; {tot / 1e5:.1f} wave-ms in all over {ph[8]} reads that kept chains past the weight filter, longest read {ph[7] / 100.0:.0f} us", flush=True)
        x2 = (C.c_ulonglong * 8)()
        gpu.L.bwagpu_debug_seed_x2.argtypes = [C.c_void_p, C.c_void_p]
        gpu.L.bwagpu_debug_seed_x2(gpu.h, x2)
        print(f"    index-block steps per read: forward {x2[0] / n:.1f} ({x2[2] / n:.1f} on one-row intervals, in {x2[4] / n:.2f} runs), backward {x2[1] / n:.1f} "
              f"({x2[3] / n:.1f} on one-row intervals, in {x2[5] / n:.2f} runs); one-row share of all block steps {100.0 * (x2[2] + x2[3]) / max(x2[0] + x2[1], 1):.1f}%", flush=True)
        print(f"    interval-stack entries taken from HBM scratch: {out[10] / n:.1f} lane steps per read ({out[11] / n:.1f} of them served by the entry fetched a step ahead); "
              f"wave iterations that read the stack from HBM: {out[12] / max(it, 1) * 100:.1f}%", flush=True)
        print("    iterations per read: " + ", ".join(f"<{1 << b}: {100.0 * hist[b] / tot_r:.1f}% of reads / {100.0 * hist[32 + b] / tot_i:.1f}% of iterations" for b in range(32) if hist[b]), flush=True)
    for k, v in old.items():
        gpu.set_option(k, v)
gpu.close()
