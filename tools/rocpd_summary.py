#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_stats.csv) as a markdown table.
usage: tools/rocpd_summary.py <dir-or-db> [title]"""
import glob
import os
import sqlite3
import sys


def from_db(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    q = (f"select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6, max(d.end-d.start)/1e6, "
         f"max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x), max(d.workgroup_size_x) "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
    return list(cur.execute(q))


def main():
    src = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else src
    dbs = [src] if src.endswith(".db") else glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    rows = []
    for d in dbs:
        rows += from_db(d)
    tot = sum(r[2] for r in rows) or 1.0
    print(f"# {title}\n")
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % | VGPR | SGPR | LDS B | scratch B | grid | block |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        name = r[0].replace(".kd", "")
        print(f"| `{name}` | {r[1]} | {r[2]:.3f} | {r[3]:.3f} | {r[4]:.3f} | {r[5]:.3f} | {100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")


if __name__ == "__main__":
    main()
