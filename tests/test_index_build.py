"""The device-side index builder (bwagpu_index_build in bwa_amd/csrc/bwagpu_index.hip, bound by bwa_amd/index.py) must write
the same five files as the reference's `bwa index`.  CPU here: the unmodified HIP source under the mock runtime of
tests/hostsim (rocPRIM's sort/scan replaced by std:: stand-ins); the same code on the GPU in tests/test_gpu_index.py."""
import filecmp
import os
import numpy as np
import pytest

import hostsim_build
import refapi
import testdata
from bwa_amd import simdata
from bwa_amd.index import build_index


def test_small_index_equals_committed_reference_index(tmp_path):
    g, lens = testdata.small_genome()
    prefix = str(tmp_path / "mine")
    build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], lib_path=hostsim_build.build())
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
    build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], lib_path=hostsim_build.build())
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, fa + "." + ext, shallow=False), ext


@pytest.mark.skipif(not refapi.have_ref(), reason="oracle/_ref not built")
@pytest.mark.parametrize("env", [{"BWAGPU_INDEX_BUCKET_BASES": "2"}, {"BWAGPU_INDEX_BUCKET_BASES": "3", "BWAGPU_INDEX_SPLIT_SORT": "1"}])
def test_index_bucketed_and_split_sort_paths(tmp_path, monkeypatch, env):
    """Large-genome code paths forced on a small genome: several first-pass buckets, and the two-sort form of a doubling
    round (used when group number + rank exceed 64 bits)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g, lens = simdata.make_genome(50_000, n_contigs=2, seed=11)
    g[7000:9000] = np.tile(g[7000:7050], 40)          # long exact tandem repeat
    g[-300:] = 0                                       # poly-A up to the end of the text: exercises the terminator ordering
    fa = str(tmp_path / "ref.fa")
    simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    prefix = str(tmp_path / "mine")
    build_index(prefix, g, [(f"chr{i + 1}", l) for i, l in enumerate(lens)], lib_path=hostsim_build.build())
    for ext in ("bwt", "sa", "pac", "ann", "amb"):
        assert filecmp.cmp(prefix + "." + ext, fa + "." + ext, shallow=False), ext
