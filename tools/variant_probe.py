#!/usr/bin/env python3
"""A/B runs of kernel variants that sit behind library options (bwagpu_set_option, bwa_amd/csrc/bwagpu_config.h) on the GPU.  bench.py starts this as a child process with a time limit
and copies what it prints into the `variants` object of its JSON line -- a variant that faults or hangs takes this process with it,
never the headline measurement.

Every argument is one configuration: space-separated name=value option settings, e.g. "seed_mrg=2 seed_lds_ent=4" ("" = the library's
defaults, always run first; the round-3 spelling BWAGPU_SEED_MRG=2 is accepted too).  For each
one: the resident hot path of one batch alone on the chip (stage times from the library's HIP events, best of `--passes`), the
step time with `--streams` batches in flight (the headline's timing loop), and a digest of the batch's regions, which must equal the
default configuration's -- a variant may only change when things are computed, never what.

Prints one JSON object per configuration, one per line, as it goes (so that a time-out keeps the lines already out).
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*")
    ap.add_argument("--prefix", required=True, help="index files (bench.py's cache)")
    ap.add_argument("--codes", required=True, help="the genome's base codes (.npy), to draw the reads from")
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--streams", type=int, default=3)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--dense-sa", type=int, default=1)
    ap.add_argument("--lib", default=os.environ.get("BWA_AMD_PROBE_LIB"), help="(tests) the mock-runtime build of the library")
    ap.add_argument("--long-reads", type=int, default=0, help="long-read mode (BASELINE configs[4]): this many reads of --long-len bases, -x pacbio, one batch, one handle")
    ap.add_argument("--long-len", type=int, default=10000)
    ap.add_argument("--batch-files", default="", help="comma-separated .npy files (reads x bases, nt4 codes) to use as the batches instead of drawing them (bench.py hands over its own)")
    ap.add_argument("--long-file", default="", help="the same for the long-read batch")
    args = ap.parse_args()

    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt

    g = np.load(args.codes, mmap_mode="r")
    if args.long_reads:
        return long_mode(args, g)
    opt = default_opt()
    opt.flag |= 0x2
    S = max(1, args.streams)
    batches = []
    files = [f for f in args.batch_files.split(",") if f]
    for si in range(S):
        if files:
            rd = np.load(files[si % len(files)])
        else:
            r1, r2 = simdata.make_reads_pe(g, args.reads // 2, length=args.read_len, seed=1000 + si)      # (bench.py's rank-0 batches)
            rd = np.empty((2 * r1.shape[0], r1.shape[1]), dtype=np.uint8)
            rd[0::2] = r1; rd[1::2] = r2
        batches.append((np.ascontiguousarray(rd.reshape(-1)), np.arange(0, rd.shape[0] + 1, dtype=np.int64) * rd.shape[1]))
    n_batch = batches[0][1].shape[0] - 1

    base_digest = None
    shared = None
    for cfg in [""] + [c for c in args.configs if c.strip()]:
        sets = parse_config(cfg)
        t0 = time.time()
        res = {"config": cfg or "defaults"}
        restore = {}
        try:
            gpu, shared = open_index(args, sets, shared)
            restore = apply_options(gpu, sets)
            handles = [gpu] + [gpu.clone() for _ in range(S - 1)]
            for hdl, (flat, off) in zip(handles, batches):
                hdl.set_taps(False)
                hdl.upload(flat, off)
            if "share" not in sets:
                gpu.set_option("share", 100)                      # (the solo passes: every kernel with the whole chip; the default would be half of it with three handles on the index)
            gpu.run(opt)                                          # warm-up: arenas learn their sizes
            solo = []
            for _ in range(args.passes):
                gpu.run(opt)
                solo.append(gpu.stats())
            if "share" not in sets:
                gpu.set_option("share", -1)
            keys = ("ms_seed", "ms_publish", "ms_sa", "ms_chain", "ms_extend", "ms_dedup", "ms_total")
            res["stage_ms_solo"] = {k: round(min(s[k] for s in solo), 3) for k in keys}
            counts, regs = gpu.download()
            res["result_sha256_16"] = hashlib.sha256(counts.tobytes() + regs.tobytes()).hexdigest()[:16]
            if base_digest is None:
                base_digest = res["result_sha256_16"]
            res["same_result_as_defaults"] = res["result_sha256_16"] == base_digest

            def worker(hdl, n):
                for _ in range(n):
                    hdl.run(opt)
            th = [threading.Thread(target=worker, args=(h_, 1)) for h_ in handles]      # warm-up of the other handles
            [t.start() for t in th]; [t.join() for t in th]
            share = [args.steps // S + (1 if i < args.steps % S else 0) for i in range(S)]
            t1 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(handles[i], share[i])) for i in range(S) if share[i]]
            [t.start() for t in th]; [t.join() for t in th]
            dt = time.perf_counter() - t1
            res["ms_per_step"] = round(dt / args.steps * 1e3, 3)
            res["Mreads_s"] = round(n_batch * args.steps / dt / 1e6, 4)
            res["steps"] = args.steps; res["streams"] = S
            for hdl in handles[1:]:
                hdl.close()
            if gpu is not shared:
                gpu.close()
            else:
                apply_options(gpu, restore)
        except Exception as e:       # (a configuration the library refuses must not take the others with it)
            res["error"] = repr(e)
        res["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(res), flush=True)


LAYOUT_KEYS = ("occ32", "occ32_sb_shift", "ptab_m")      # options that shape what is derived from the index at load time: such a configuration gets a handle of its own


def parse_config(cfg):
    """'seed_mrg=2 BWAGPU_PTAB_M=8' -> {'seed_mrg': 2, 'ptab_m': 8}"""
    out = {}
    for kv in cfg.split():
        k, v = kv.split("=", 1)
        k = k.lower()
        out[k[7:] if k.startswith("bwagpu_") else k] = int(v)
    return out


def apply_options(gpu, sets):
    """Set the per-batch options on the handle; returns the values they had (to put back on a shared handle)."""
    old = {}
    for k, v in sets.items():
        if k in LAYOUT_KEYS:
            continue
        old[k] = gpu.get_option(k)
        gpu.set_option(k, v)
    return old


def open_index(args, sets, shared):
    """The handle a configuration runs on: one shared by all configurations whose options apply per batch, a fresh one otherwise."""
    from bwa_amd.api import BwaGpu
    own = any(k in LAYOUT_KEYS for k in sets)
    if not own and shared is not None:
        return shared, shared
    gpu = BwaGpu(args.prefix, lib_path=args.lib, options={k: v for k, v in sets.items() if k in LAYOUT_KEYS})
    if args.dense_sa:
        gpu.densify_sa(args.dense_sa)
    gpu.set_taps(False)
    return gpu, (shared if own else gpu)


def long_mode(args, g):
    """BASELINE configs[4]'s layout (bench.py's long-read leg): one resident batch of long reads, -x pacbio, stage times of the hot path."""
    from bwa_amd import simdata
    from bwa_amd.structs import pacbio_opt
    n, L = args.long_reads, args.long_len
    if args.long_file:
        reads = np.load(args.long_file); n, L = reads.shape
    else:
        reads = simdata.make_reads_long(g, n, length=L, seed=7)
    flat, off = np.ascontiguousarray(reads.reshape(-1)), np.arange(0, n + 1, dtype=np.int64) * L
    opt = pacbio_opt()
    base_digest, shared = None, None
    for cfg in [""] + [c for c in args.configs if c.strip()]:
        sets = parse_config(cfg)
        t0 = time.time()
        res = {"config": cfg or "defaults", "mode": f"{n} reads of {L} bp, -x pacbio"}
        try:
            gpu, shared = open_index(args, sets, shared)
            restore = apply_options(gpu, sets)
            gpu.upload(flat, off)
            gpu.run(opt)                                          # warm-up: arenas learn their sizes
            runs = []
            for _ in range(args.passes):
                t1 = time.perf_counter(); gpu.run(opt); ms = (time.perf_counter() - t1) * 1e3
                st = gpu.stats(); st["ms_pass"] = ms; runs.append(st)
            best = min(runs, key=lambda s_: s_["ms_pass"])
            res["stage_ms"] = {k: round(best[k], 2) for k in ("ms_seed", "ms_publish", "ms_sa", "ms_chain", "ms_seedsw", "ms_extend", "ms_dedup", "ms_total")}
            res["ms_per_pass"] = round(best["ms_pass"], 2); res["reads_per_s"] = round(n / best["ms_pass"] * 1e3, 1)
            counts, regs = gpu.download()
            res["result_sha256_16"] = hashlib.sha256(counts.tobytes() + regs.tobytes()).hexdigest()[:16]
            if base_digest is None:
                base_digest = res["result_sha256_16"]
            res["same_result_as_defaults"] = res["result_sha256_16"] == base_digest
            if gpu is not shared:
                gpu.close()
            else:
                apply_options(gpu, restore)
        except Exception as e:
            res["error"] = repr(e)
        res["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
