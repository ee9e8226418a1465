#!/usr/bin/env python3
"""One small batch through a chosen build of the library with chosen options: prints a digest of the regions (debugging aid for the de-duplication kernels;
tools/gpu_session.sh runs it under a time limit).  usage: dedup_debug.py LIB "name=value ..." [n_reads]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import testdata
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    lib = sys.argv[1] if sys.argv[1] != "-" else None
    sets = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in sys.argv[2].split()}
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    prefix, g = testdata.small_index()
    reads = simdata.make_reads_se(g, n, seed=5)
    seqs, off = testdata.flat(reads)
    gpu = BwaGpu(prefix, lib_path=lib, options=sets)
    print("handle up", sets, flush=True)
    c, r = gpu.align(default_opt(), seqs, off)
    print("OK", sets, "regions", int(c.sum()), "max per read", int(c.max()), hashlib.sha256(c.tobytes() + r.tobytes()).hexdigest()[:16], flush=True)
    gpu.close()


if __name__ == "__main__":
    main()
