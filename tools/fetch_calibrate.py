#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE for random reads of 32 and 64 bytes: `rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/randbw3 <MiB>`
issues, per variant and occupancy, a warm-up launch of 100 iterations and a timed one of 1500, every lane reading one block per iteration
(tools/randbw3.hip) -- a known byte count per launch.  usage: fetch_calibrate.py <rocprof output dir> <out.md>"""
import csv, glob, os, sys


def main():
    d, dst = sys.argv[1:3]
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE":
                rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], int(r.get("Grid_Size", 0) or 0), float(r["Counter_Value"])))
    rows.sort()
    out = ["# FETCH_SIZE against a known byte count: dependent random reads of 32 / 64 bytes from a table in HBM (`tools/randbw3.hip`)", "",
           "`FETCH_SIZE` is in KiB.  Expected bytes = lanes x iterations x block bytes (every lane reads one block per iteration; the table is far larger than",
           "the caches, so nearly every block comes from HBM).  Largest launch (1500 iterations) of each variant and occupancy.", "",
           "| kernel | lanes | block bytes | expected GB | FETCH_SIZE GB | ratio |", "|---|---:|---:|---:|---:|---:|"]
    best = {}
    for disp, name, grid, val in rows:
        if not ("k_lane" in name or "k_coop" in name):
            continue
        nb = 32 if "<32>" in name else 64
        key = (name.split("(")[0], grid)
        if key not in best or val > best[key][0]:
            best[key] = (val, nb)
    for (name, grid), (val, nb) in sorted(best.items()):
        exp = grid * 1500 * nb
        out.append(f"| `{name}` | {grid} | {nb} | {exp / 1e9:.1f} | {val * 1024 / 1e9:.1f} | {val * 1024 / exp:.2f} |")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
