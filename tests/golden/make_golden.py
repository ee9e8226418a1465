"""Generates the committed fixtures of tests/golden/ from the *compiled reference* (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the build container:  python tests/golden/make_golden.py

  g200k.{bwt,sa,pac,ann,amb}  index of the seeded 200 kb genome (testdata.SMALL), from `bwa index`
  golden_regs.npz             for several read sets / option sets: the reads and the reference's
                              mem_align1_core output (per-read counts + mem_alnreg_t records, 88 B each)
  golden_stages.npz           for one read set: the reference's SA intervals, chains (after mem_chain_flt) and
                              pre-dedup regions, so that each stage of the oracle / HIP path can be pinned

The reference ships no golden vectors of its own; these are its outputs.
"""
import os
import shutil
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from bwa_amd import simdata  # noqa: E402
from bwa_amd.structs import default_opt, pacbio_opt  # noqa: E402
import refapi  # noqa: E402
import testdata  # noqa: E402


def option_sets():
    odd = default_opt()
    odd.max_occ, odd.min_seed_len, odd.w, odd.zdrop, odd.max_chain_extend, odd.min_chain_weight = 50, 15, 20, 30, 3, 25
    return {"default": default_opt(), "pacbio": pacbio_opt(), "odd": odd}


def read_sets(g):
    n_reads = simdata.make_reads_se(g, 400, seed=101, n_frac=0.003)
    return {
        "se150": ("default", simdata.make_reads_se(g, 600, seed=100)),
        "se150_N": ("default", n_reads),
        "se100_noisy": ("default", simdata.make_reads_se(g, 300, length=100, seed=102, sub=0.04, dele=0.006, ins=0.006)),
        "se250_odd": ("odd", simdata.make_reads_se(g, 300, length=250, seed=103, sub=0.02, dele=0.004, ins=0.004)),
        "se40": ("default", simdata.make_reads_se(g, 200, length=40, seed=104)),
        "se17": ("default", simdata.make_reads_se(g, 50, length=17, seed=105)),
        "long2k_pacbio": ("pacbio", simdata.make_reads_long(g, 12, length=2000, seed=106)),
        "long1500_default": ("default", simdata.make_reads_long(g, 12, length=1500, seed=107, sub=0.01, dele=0.01, ins=0.01)),
    }


def main():
    g, lens = testdata.small_genome()
    tmp = os.path.join(refapi.DATA, "golden_build")
    os.makedirs(tmp, exist_ok=True)
    fa = os.path.join(tmp, "g200k.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        shutil.copy(fa + "." + ext, os.path.join(testdata.GOLDEN, "g200k." + ext))
    idx = refapi.RefIndex(fa)
    opts = option_sets()
    out = {}
    for name, (oname, reads) in read_sets(g).items():
        seqs, off = testdata.flat(reads)
        counts, regs = idx.align(opts[oname], seqs, off)
        out[name + "/reads"] = reads
        out[name + "/opt"] = np.array(oname)
        out[name + "/counts"] = counts
        out[name + "/regs"] = regs
        print(name, oname, reads.shape, int(counts.sum()), int(counts.max()))
    np.savez_compressed(os.path.join(testdata.GOLDEN, "golden_regs.npz"), **out)
    # stage-level fixtures for one read set
    reads = simdata.make_reads_se(g, 200, seed=108)
    st = {"reads": reads}
    ivn, iv, chn, chh, chs_n, chs, rrn, rr = [], [], [], [], [], [], [], []
    for r in reads:
        a = idx.intervals(opts["default"], r); ivn.append(len(a)); iv.append(a)
        h, s = idx.chains(opts["default"], r, 1); chn.append(len(h)); chh.append(h); chs.append(s)
        x = idx.regs_stage(opts["default"], r, 0); rrn.append(len(x)); rr.append(x)
    st["intv_n"] = np.array(ivn, dtype=np.int32); st["intv"] = np.concatenate(iv)
    st["chain_n"] = np.array(chn, dtype=np.int32); st["chain_hdr"] = np.concatenate(chh); st["chain_seeds"] = np.concatenate(chs)
    st["raw_n"] = np.array(rrn, dtype=np.int32); st["raw_regs"] = np.concatenate(rr)
    np.savez_compressed(os.path.join(testdata.GOLDEN, "golden_stages.npz"), **st)
    print("stages:", st["intv"].shape, st["chain_hdr"].shape, st["chain_seeds"].shape, st["raw_regs"].shape)


if __name__ == "__main__":
    main()
