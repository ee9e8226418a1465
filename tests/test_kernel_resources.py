"""CPU: the default kernels' register and scratch use, read from the code object's metadata (tools/isa_resources.py: hipcc -S, no GPU), against bounds a
little above what the tree has.  A change that doubles a kernel's registers or sends its arrays to scratch still passes every parity test and costs a
GPU session to notice: round 6's list launch of k_dedup_wave went from 160 VGPRs / 784 bytes of scratch to 248 / 2336 when the in-place routine was
inlined at a second call site, and from 0.7 to 69 ms (profiles/r06_dedup.md)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# kernel (as tools/isa_resources.py prints it): (VGPRs, scratch bytes per lane, spilled VGPRs) it may use at most
BOUNDS = {
    "k_seed<true, false, 1, 4, 2, 0>": (128, 64, 8),          # the headline's seeding kernel: four waves per SIMD
    "k_seed3<1, false>": (96, 0, 0),
    "k_seed<false, false, 1, 3, 2, 1>": (168, 0, 0),          # the heavy reads' task kernels: three waves per SIMD
    "k_seed<false, false, 1, 3, 2, 3>": (168, 0, 0),
    "k_publish": (80, 512, 0),
    "k_sa": (64, 0, 0),
    "k_chain_wave": (96, 544, 16),                             # five waves per SIMD
    "k_extend_wave<false, 6>": (80, 496, 4),                   # six waves per SIMD
    "k_extend_wave<true, 4>": (96, 496, 0),
    "k_dedup": (112, 592, 0),
    "k_dedup_wave<true, true>": (170, 800, 0),                 # three waves per SIMD (512 / 3 = 170 registers; ten one-wave workgroups per CU by LDS)
    "k_dedup_wave<true, false>": (144, 592, 0),
    "k_cigar<true>": (96, 0, 0),
    "k_cigar<false>": (144, 0, 0),
    "k_matesw_sw": (96, 0, 0),
}


def test_default_kernels_stay_within_their_register_and_scratch_budgets():
    import isa_resources
    from bwa_amd import build
    if not os.path.exists(build.HIPCC):
        pytest.skip("hipcc is not installed")
    rows = {r[0]: r for r in isa_resources.kernels(os.path.join(build.CSRC, "bwagpu.hip"))}
    missing = [k for k in BOUNDS if k not in rows]
    assert not missing, f"kernels not in the code object (renamed? update BOUNDS): {missing}"
    over = []
    for k, (vgpr, scratch, spill) in BOUNDS.items():
        r = rows[k]
        got = (int(r[1]), int(r[4]), int(r[6]))
        if got[0] > vgpr or got[1] > scratch or got[2] > spill:
            over.append(f"{k}: vgpr/scratch/spill {got} > {(vgpr, scratch, spill)}")
    assert not over, "; ".join(over)
