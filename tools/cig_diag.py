import sys, os, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import testdata
from bwa_amd import simdata
from bwa_amd.api import BwaGpu
from bwa_amd.structs import default_opt
fa, g = testdata.medium_index()
r1, r2 = simdata.make_reads_pe(g, 20000, seed=402)
idx = [i for i in range(20000) if i % 7 == 3]
singles = np.stack([r1[i][::-1] % 4 for i in idx])
s = BwaGpu(fa)
opt = default_opt()
seqs, off = testdata.flat(singles)
counts, regs = s.align(opt, seqs, off); print("aligned", regs.shape[0], flush=True)
t = time.time(); cigs = s.cigars(opt); print("cigars done in %.2fs" % (time.time() - t), np.unique(cigs["n_cigar"], return_counts=True), flush=True)
