"""The device-side index builder (bwa_amd/index.py, torch prefix-doubling suffix array) must write the same five files as the
reference's `bwa index` (CPU here; the same code runs on the GPU in bench.py)."""
import filecmp
import os
import numpy as np
import pytest

import refapi
import testdata
from bwa_amd import simdata
from bwa_amd.index import build_index


def test_small_index_equals_committed_reference_index(tmp_path):
    g, lens = testdata.small_genome()
    prefix = str(tmp_path / "mine")
    build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], device="cpu")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, os.path.join(testdata.GOLDEN, "g200k." + ext), shallow=False), ext


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("total,seed", [(1003, 5), (65536, 6), (300001, 7)])
def test_index_equals_bwa_index(tmp_path, total, seed):
    g, lens = simdata.make_genome(total, n_contigs=2 if total > 5000 else 1, seed=seed, repeats=total > 5000)
    if total < 5000:
        lens = [total]
        g[100:400] = np.tile(g[100:130], 10)       # an exact tandem repeat: deep prefix doubling
    fa = str(tmp_path / "ref.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    prefix = str(tmp_path / "mine")
    build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], device="cpu")
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, fa + "." + ext, shallow=False), ext
