"""Build libbwagpu.so (HIP, gfx950) in-tree:  python -m bwa_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(CSRC, "libbwagpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the reference's float/double threshold arithmetic (bwamem.c:381,442-458,649-650) must not be
# fused into FMAs, or borderline comparisons could flip.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h"))] + [
        os.path.join(os.path.dirname(HERE), "include", "bwagpu.h")]


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = sources()
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, "bwagpu.hip"), "-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
