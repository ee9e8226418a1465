"""Builds tests/hostsim/libbwagpu_hostsim.so: the unmodified product source bwa_amd/csrc/bwagpu.hip compiled with g++
against the mock HIP runtime of tests/hostsim/hip/hip_runtime.h (serial lane emulation).  TEST INFRASTRUCTURE ONLY --
this lets the CPU-only suite exercise the host orchestration and the lane-serial device routines without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "hostsim")
OUT = os.path.join(SIM, "libbwagpu_hostsim.so")


def build():
    if os.environ.get("BWA_AMD_HOSTSIM_LIB"):       # e.g. a build of the same sources with -fsanitize=address,undefined (tools/sanitize_mock.sh)
        return os.environ["BWA_AMD_HOSTSIM_LIB"]
    csrc = os.path.join(ROOT, "bwa_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))] + [
        os.path.join(SIM, "hip", "hip_runtime.h"), os.path.join(SIM, "mock_globals.cpp"), os.path.join(ROOT, "include", "bwagpu.h"),
        os.path.join(SIM, "rocprim", "functional.hpp")]
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    subprocess.run(["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-x", "c++", "-I", SIM,
                    os.path.join(csrc, "bwagpu.hip"), os.path.join(csrc, "bwagpu_index.hip"), os.path.join(SIM, "mock_globals.cpp"), "-o", OUT], check=True)
    return OUT
