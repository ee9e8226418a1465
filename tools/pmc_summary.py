#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter) into profiles/pmc_latest.json.

usage: pmc_summary.py <note> <fetch_dir> <write_dir> [out.json]
Per kernel: the launch with the largest FETCH_SIZE (the instrumented solo pass and the timed step run the same
batch; warm-up-free runs have one or two launches per kernel).  bytes = (FETCH_SIZE + WRITE_SIZE) KiB * 1024.
"""
import csv, glob, json, os, sys


def load(d, name):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"].split("(")[0]
            if k.startswith("void "):
                k = k[5:]
            out.setdefault(k, []).append(float(r["Counter_Value"]))
    return out


def summarize(note, fd, wd):
    """{kernel: {FETCH_SIZE_KB, WRITE_SIZE_KB, ...}} from the two passes' output directories (bench.py's in-run passes use this too)."""
    F, W = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    import datetime
    res = {"_note": note, "_meta": {"what": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes; raw counters (FETCH_SIZE counts 64 bytes per request whatever its size, profiles/r03_fetch_calibration.md)", "date": datetime.date.today().isoformat()}}
    for k in sorted(F, key=lambda k: -max(F[k])):
        if not k.startswith("k_"):
            continue
        f, w = max(F[k]), max(W.get(k, [0.0]))
        res[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "launches_seen": len(F[k]),
                  "hbm_bytes_per_launch": (f + w) * 1024.0}
    # the chaining stage under its bench name (rounds 2-4: three template instances, summed)
    tiers = [k for k in res if k.startswith("k_chain_wave")]
    if tiers:
        res["k_chain"] = {"FETCH_SIZE_KB": sum(res[k]["FETCH_SIZE_KB"] for k in tiers), "WRITE_SIZE_KB": sum(res[k]["WRITE_SIZE_KB"] for k in tiers),
                          "launches_seen": min(res[k]["launches_seen"] for k in tiers), "hbm_bytes_per_launch": sum(res[k]["hbm_bytes_per_launch"] for k in tiers)}
    for k in list(res):
        if k.startswith("k_extend_wave<"):
            res["k_extend_wave"] = res[k]
    # the seeding stage's lane-per-read kernel has an instance with work counters (one launch per bench run) and one without (the timed
    # launches); since round 4 the reads it gives up are seeded by task kernels (other instances of the same template: RD = false).  k_seed = the
    # counter-free lane-per-read instance + the counter-free task instances; k_seed3 = pass 3
    def targs(k):
        return [a.strip() for a in k[k.index("<") + 1:-1].split(",")]
    plain = [k for k in res if k.startswith("k_seed<") and targs(k)[1:2] == ["false"]]
    if plain:
        res["k_seed"] = {"FETCH_SIZE_KB": sum(res[k]["FETCH_SIZE_KB"] for k in plain), "WRITE_SIZE_KB": sum(res[k]["WRITE_SIZE_KB"] for k in plain),
                         "launches_seen": min(res[k]["launches_seen"] for k in plain), "hbm_bytes_per_launch": sum(res[k]["hbm_bytes_per_launch"] for k in plain), "instances": plain}
    for k in list(res):
        if k.startswith("k_seed3<") and "k_seed3" not in res:
            res["k_seed3"] = res[k]
    return res


def main():
    note, fd, wd = sys.argv[1:4]
    dst = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(__file__), "..", "profiles", "pmc_latest.json")
    res = summarize(note, fd, wd)
    json.dump(res, open(dst, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:20s} fetch {v['FETCH_SIZE_KB']*1024/1e9:8.2f} GB  write {v['WRITE_SIZE_KB']*1024/1e9:8.2f} GB  x{v['launches_seen']}")


if __name__ == "__main__":
    main()
