#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel of libbwagpu.so, from the code object's metadata (CPU only: hipcc -S --cuda-device-only).
A default kernel that starts to spill after a change shows up here before it reaches a GPU box:  python tools/isa_resources.py [substring ...]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwa_amd import build  # noqa: E402


def kernels(src):
    with tempfile.TemporaryDirectory() as td:
        s = os.path.join(td, "k.s")
        subprocess.run([build.HIPCC] + [f for f in build.FLAGS if f not in ("-shared", "-Wall")] + ["--cuda-device-only", "-S", src, "-o", s], check=True, stderr=subprocess.DEVNULL)
        txt = open(s).read()
    meta = txt[txt.find("amdhsa.kernels"):]
    n_ins = {}
    for m in re.finditer(r"\n(_Z\w+|\w+):\s+; @\1\n(.*?)\n\s*s_endpgm", txt, flags=re.S):        # static instruction count of each kernel's body
        n_ins[m.group(1)] = sum(1 for ln in m.group(2).split("\n") if ln.startswith("\t") and not ln.strip().startswith((".", ";")))
    rows = []
    for k in re.split(r"\n  - ", meta)[1:]:
        def g(key):
            m = re.search(r"\." + key + r":\s*(\S+)", k)
            return m.group(1) if m else "?"
        rows.append([g("name")] + [g(x) for x in ("vgpr_count", "agpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size", "vgpr_spill_count", "max_flat_workgroup_size")] + [str(n_ins.get(g("name"), "?"))])
    names = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, names):
        r[0] = re.sub(r"^void ", "", re.sub(r"\(.*", "", d))
    return rows


if __name__ == "__main__":
    pats = sys.argv[1:]
    print(f"{'kernel':70s} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7} {'spill':>6} {'wg':>5} {'instrs':>7}")
    for r in kernels(os.path.join(build.CSRC, "bwagpu.hip")):
        if not pats or any(p in r[0] for p in pats):
            print(f"{r[0][:70]:70s} {r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>8} {r[5]:>7} {r[6]:>6} {r[7]:>5} {r[8]:>7}")
