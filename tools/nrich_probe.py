#!/usr/bin/env python3
"""Diagnostics (GPU): the cases of test_short_read_batches_with_ns_and_seed_length_options one per subprocess, so that a crash names its case."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "case":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import testdata, orcapi
    from test_gpu_parity import _n_rich_reads, assert_regs_equal
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    kind, val = sys.argv[2], int(sys.argv[3])
    prefix, g = testdata.small_index()
    gpu, orc = BwaGpu(prefix), orcapi.OrcIndex(prefix)
    opt = default_opt()
    if kind == "len":
        seqs, off = testdata.ragged(_n_rich_reads(g, 2000, val, seed=300 + val))
    else:
        seqs, off = testdata.flat(simdata.make_reads_se(g, 3000, seed=77, sub=0.04)); opt.min_seed_len = val
    want = orc.align(opt, seqs, off)
    print("oracle done", flush=True)
    got = gpu.align(opt, seqs, off)
    assert_regs_equal(*want, *got, f"{kind} {val}")
    print("OK", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "seq":      # the whole sequence on one handle, as the test runs it
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import faulthandler; faulthandler.enable()
    import numpy as np
    import testdata, orcapi
    from test_gpu_parity import _n_rich_reads, assert_regs_equal
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    prefix, g = testdata.small_index()
    gpu, orc = BwaGpu(prefix), orcapi.OrcIndex(prefix)
    for max_len in (120, 150, 250):
        seqs, off = testdata.ragged(_n_rich_reads(g, 2000, max_len, seed=300 + max_len))
        want = orc.align(default_opt(), seqs, off); print("oracle", max_len, flush=True)
        gpu.upload(seqs, off); print("uploaded", flush=True)
        gpu.run(default_opt()); print("ran", gpu.L.bwagpu_debug_phase(gpu.h), flush=True)
        got = gpu.download(); print("downloaded", flush=True)
        assert_regs_equal(*want, *got, f"{max_len}"); print("ok", max_len, flush=True)
    seqs, off = testdata.flat(simdata.make_reads_se(g, 3000, seed=77, sub=0.04))
    for k in (9, 10, 11, 14):
        opt = default_opt(); opt.min_seed_len = k
        assert_regs_equal(*orc.align(opt, seqs, off), *gpu.align(opt, seqs, off), f"k {k}"); print("ok k", k, flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "seq2":      # one ragged N-rich batch of <= argv[2] bases, then 3000 x 150 bp with -k argv[3], on one handle
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import testdata, orcapi
    from test_gpu_parity import _n_rich_reads, assert_regs_equal
    from bwa_amd import simdata
    from bwa_amd.api import BwaGpu
    from bwa_amd.structs import default_opt
    prefix, g = testdata.small_index()
    gpu, orc = BwaGpu(prefix), orcapi.OrcIndex(prefix)
    first, k, nrd = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    if first:
        seqs, off = testdata.ragged(_n_rich_reads(g, 2000, first, seed=300 + first))
        gpu.align(default_opt(), seqs, off); print("first ok", flush=True)
    seqs, off = testdata.flat(simdata.make_reads_se(g, nrd, seed=77, sub=0.04))
    opt = default_opt(); opt.min_seed_len = k
    gpu.set_stats(True)
    got = gpu.align(opt, seqs, off); print("second ran", gpu.stats()["n_retries"], flush=True)
    assert_regs_equal(*orc.align(opt, seqs, off), *got, "second"); print("OK", flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "runseq2":
    for first, k, nrd, env in ((250, 9, 3000, {}), (250, 19, 3000, {}), (150, 9, 3000, {}), (250, 9, 1500, {}), (250, 9, 3000, {"BWAGPU_SEED_RD_LDS": "0"}), (250, 9, 3000, {"BWAGPU_SEED_NO_VIRT": "1"}), (120, 9, 3000, {}), (250, 11, 3000, {})):
        p = subprocess.run([sys.executable, __file__, "seq2", str(first), str(k), str(nrd)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        err = [l for l in p.stderr.decode().splitlines() if any(w in l for w in ("fault", "Fatal", "Error", "error", "assert", "HSA", "Abort"))]
        print(first, k, nrd, env, "rc", p.returncode, p.stdout.decode().split("\n")[-3:], err[:2], flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "runseq":
    for env in ({}, {"BWAGPU_SEED_NO_VIRT": "1"}):
        p = subprocess.run([sys.executable, __file__, "seq"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        err = [l for l in p.stderr.decode().splitlines() if any(w in l for w in ("fault", "Fatal", "[bwagpu]"))]
        print(env, "rc", p.returncode, p.stdout.decode().split()[-4:], err[-6:], flush=True)
    sys.exit(0)
cases = [("len", 120, {}), ("len", 150, {}), ("len", 250, {}), ("k", 9, {}), ("k", 10, {}), ("k", 11, {}), ("k", 14, {}),
         ("len", 150, {"BWAGPU_SEED_RD_LDS": "0"}), ("len", 150, {"BWAGPU_SEED_NO_VIRT": "1"}), ("k", 9, {"BWAGPU_SEED_RD_LDS": "0"})]
for kind, val, env in cases:
    p = subprocess.run([sys.executable, __file__, "case", kind, str(val)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    out = p.stdout.decode().split()
    err = [l for l in p.stderr.decode().splitlines() if any(w in l for w in ("fault", "Fatal", "Error", "error", "assert", "HSA", "Abort"))]
    print(kind, val, env, "rc", p.returncode, out[-2:], err[:4], flush=True)
