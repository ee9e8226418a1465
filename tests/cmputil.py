import numpy as np


def assert_regs_equal(c_a, r_a, c_b, r_b, what=""):
    """Bit-exact comparison of (counts, mem_alnreg_t records) with a readable report on mismatch."""
    assert np.array_equal(c_a, c_b), f"{what}: per-read region counts differ at reads {np.nonzero(c_a != c_b)[0][:10]}"
    if r_a.tobytes() == r_b.tobytes():
        return
    for f in r_a.dtype.names:
        d = np.nonzero(r_a[f] != r_b[f])[0]
        if len(d):
            raise AssertionError(f"{what}: field {f} differs in {len(d)} regions, first {d[:5]}: {r_a[f][d[:5]]} vs {r_b[f][d[:5]]}")
    raise AssertionError(f"{what}: records differ in padding bytes only")


def golden_opts():
    from bwa_amd.structs import default_opt, pacbio_opt
    odd = default_opt()
    odd.max_occ, odd.min_seed_len, odd.w, odd.zdrop, odd.max_chain_extend, odd.min_chain_weight = 50, 15, 20, 30, 3, 25
    return {"default": default_opt(), "pacbio": pacbio_opt(), "odd": odd}


def golden_sets(path):
    z = np.load(path)
    names = sorted({k.split("/")[0] for k in z.files})
    return [(n, str(z[n + "/opt"]), z[n + "/reads"], z[n + "/counts"], z[n + "/regs"]) for n in names]
