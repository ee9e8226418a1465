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
    # two translation units (the index builder pulls in rocPRIM's sort templates and rarely changes): objects are rebuilt
    # only when one of their own sources is newer
    objs = []
    for tu, deps in (("bwagpu.hip", [s for s in srcs if not s.endswith("bwagpu_index.hip")]), ("bwagpu_index.hip", [os.path.join(CSRC, "bwagpu_index.hip"), srcs[-1]])):
        obj = os.path.join(CSRC, tu.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + ["-c", os.path.join(CSRC, tu), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


HOST = os.path.join(CSRC, "host")
HOST_SO = os.path.join(HOST, "libbwamem_host.so")
CLI = os.path.join(HERE, "bwa-amd")
HOST_FLAGS = ["-O2", "-g", "-std=c++17", "-fPIC", "-Wall", "-Wno-misleading-indentation", "-ffp-contract=off"]


def build_host(force: bool = False, verbose: bool = True):
    """Host finalize library (g++, no HIP) and the stand-alone `bwa-amd` command line (links libbwagpu.so)."""
    srcs = [os.path.join(HOST, f) for f in sorted(os.listdir(HOST)) if f.endswith((".cpp", ".h"))]
    lib_cpp = [s for s in srcs if s.endswith(".cpp") and not os.path.basename(s).startswith("main_")]
    newest = max(os.path.getmtime(s) for s in srcs)
    if force or not os.path.exists(HOST_SO) or os.path.getmtime(HOST_SO) < newest:
        cmd = ["g++"] + HOST_FLAGS + ["-shared"] + lib_cpp + ["-o", HOST_SO, "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if force or not os.path.exists(CLI) or os.path.getmtime(CLI) < max(newest, os.path.getmtime(OUT)):
        cmd = ["g++"] + HOST_FLAGS + lib_cpp + [os.path.join(HOST, "main_mem.cpp"), "-o", CLI, "-L" + CSRC, "-lbwagpu",
                                               "-Wl,-rpath,$ORIGIN/csrc", "-Wl,-rpath,/opt/rocm/lib", "-lz", "-lpthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_SO, CLI


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    build_host(force="--force" in sys.argv)
