"""CPU: the plain-C oracle (oracle/orc_*.c) against the committed reference outputs in tests/golden/."""
import os
import numpy as np
import pytest

import orcapi
import testdata
from cmputil import assert_regs_equal, golden_opts, golden_sets
from bwa_amd.structs import ALNREG_DTYPE, INTV_DTYPE, SEED_DTYPE


@pytest.fixture(scope="module")
def orc():
    prefix, _ = testdata.small_index()
    o = orcapi.OrcIndex(prefix)
    yield o
    o.close()


def test_genome_matches_committed_pac():
    prefix, g = testdata.small_index()
    pac = np.fromfile(prefix + ".pac", dtype=np.uint8)
    nb = (len(g) + 3) // 4
    codes = np.stack([(pac[:nb] >> (6 - 2 * k)) & 3 for k in range(4)], axis=1).reshape(-1)
    assert np.array_equal(codes[: len(g)], g)


def test_oracle_regs_match_reference_golden(orc):
    opts = golden_opts()
    for name, oname, reads, counts, regs in golden_sets(os.path.join(testdata.GOLDEN, "golden_regs.npz")):
        seqs, off = testdata.flat(reads)
        c, r = orc.align(opts[oname], seqs, off)
        assert_regs_equal(counts, regs.astype(ALNREG_DTYPE), c, r, f"golden {name}")


def test_oracle_stages_match_reference_golden(orc):
    z = np.load(os.path.join(testdata.GOLDEN, "golden_stages.npz"))
    opt = golden_opts()["default"]
    io = co = so = ro = 0
    for i, read in enumerate(z["reads"]):
        iv = orc.intervals(opt, read)
        n = int(z["intv_n"][i])
        assert len(iv) == n and iv.tobytes() == z["intv"][io:io + n].astype(INTV_DTYPE).tobytes(), f"intervals of read {i}"
        io += n
        h, s = orc.chains(opt, read, 1)
        n = int(z["chain_n"][i])
        gh = z["chain_hdr"][co:co + n]
        assert len(h) == n
        for f in ("n", "rid", "w", "kept", "is_alt", "frac_rep", "pos"):
            assert np.array_equal(h[f], gh[f]), f"chain field {f} of read {i}"
        ns = int(gh["n"].sum())
        for f in ("rbeg", "qbeg", "len", "score"):
            assert np.array_equal(s[f], z["chain_seeds"][so:so + ns][f]), f"seed field {f} of read {i}"
        co += n
        so += ns
        rr = orc.regs_stage(opt, read, 0)
        n = int(z["raw_n"][i])
        assert len(rr) == n and rr.tobytes() == z["raw_regs"][ro:ro + n].astype(ALNREG_DTYPE).tobytes(), f"raw regs of read {i}"
        ro += n
