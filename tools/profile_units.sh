#!/bin/bash
# Which unit a kernel waits on: TA / TCP / SQ wait counters of the hot-path kernels of a solo 1 M-read batch, one rocprofv3 --pmc pass per group
# (gpurun allows --pmc only together with --kernel-trace).   tools/profile_units.sh <tag>
tag=${1:-units}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e --no-longread"
i=0
for grp in "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN2_sum" \
           "FETCH_SIZE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o c -- $B > $out/p$i.log 2>&1
done
python - $out <<'P'
import csv, glob, os, sys
out = sys.argv[1]
vals = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if not k.startswith(("k_seed<true", "k_extend_wave", "k_chain_wave", "k_dedup", "k_seed3", "k_sa")):
            continue
        key = (k, r["Counter_Name"])
        vals[key] = max(vals.get(key, 0.0), float(r["Counter_Value"]))
names = sorted({c for _, c in vals})
kernels = sorted({k for k, _ in vals})
with open(os.path.join(out, "units.md"), "w") as o:
    o.write("# per launch (the largest value over a kernel's launches), rocprofv3 --kernel-trace --pmc, one pass per counter group; solo batch of 1 M reads (bench.py --steps 1 --warmup 0 --streams 1)\n\n")
    o.write("| counter | " + " | ".join(f"`{k}`" for k in kernels) + " |\n|---|" + "---:|" * len(kernels) + "\n")
    for n in names:
        o.write(f"| {n} | " + " | ".join(f"{vals.get((k, n), 0):.3g}" for k in kernels) + " |\n")
print(open(os.path.join(out, "units.md")).read())
P
