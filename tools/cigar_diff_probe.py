"""Diagnostics: device CIGAR records vs the host's on the medium test genome; prints the first differing records."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import testdata, hostapi
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt

prefix, g = testdata.medium_index()
gpu = BwaGpu(prefix); host = hostapi.HostFinalize(prefix)
for kw in (dict(sub=0.03, dele=0.004, ins=0.004), dict(sub=0.02, dele=0.03, ins=0.03)):
    opt = default_opt()
    seqs, off = testdata.flat(simdata.make_reads_se(g, 20000, seed=610, **kw))
    counts, regs = gpu.align(opt, seqs, off)
    cigs, ops = gpu.cigars(opt), gpu.cigar_ops()
    hc, hops = host.region_cigars(opt, seqs, off, counts, regs, with_ops=True)
    bad = np.flatnonzero((cigs["score"] != hc["score"]) | (cigs["n_cigar"] != hc["n_cigar"]))
    print(kw, "regions", regs.shape[0], "ops", ops.shape[0], hops.shape[0], "differing", bad.shape[0], "long", int((hc["n_cigar"] > 6).sum()))
    for b in bad[:8]:
        r = regs[b]
        print("  reg", b, "dev", cigs[b]["score"], cigs[b]["n_cigar"], "host", hc[b]["score"], hc[b]["n_cigar"], "qb/qe", r["qb"], r["qe"], "rlen", r["re"] - r["rb"], "score", r["score"], "truesc", r["truesc"], "w", r["w"])
    d, h = hostapi.decode_cigars(cigs, ops), hostapi.decode_cigars(hc, hops)
    nb = [i for i in range(len(d)) if d[i] != h[i]]
    print("  decoded differing", len(nb))
    for i in nb[:4]:
        print("   ", i, d[i], h[i])
