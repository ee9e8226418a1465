#!/usr/bin/env python3
"""Fold a rocprofv3 --pmc SQ_* counter_collection CSV into a markdown table: per kernel, the launch with the most vector instructions.
usage: sq_summary.py <dir> <out.md> [note]"""
import csv, glob, os, sys


def main():
    d, dst = sys.argv[1:3]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            rows.setdefault((k, r.get("Dispatch_Id", "0")), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    best = {}
    for (k, _), c in rows.items():
        if k not in best or c.get("SQ_INSTS_VALU", 0) > best[k].get("SQ_INSTS_VALU", 0):
            best[k] = c
    names = sorted({n for c in best.values() for n in c})
    with open(dst, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --pmc {' '.join(names)} (one pass), per launch\n\n{note}\n\n")
        o.write("| kernel | " + " | ".join(names) + " |\n|---|" + "---:|" * len(names) + "\n")
        for k in sorted(best, key=lambda k: -best[k].get("SQ_INSTS_VALU", 0)):
            if k.startswith("k_"):
                o.write(f"| `{k}` | " + " | ".join(f"{best[k].get(n, 0):.3g}" for n in names) + " |\n")
    print(open(dst).read())


if __name__ == "__main__":
    main()
