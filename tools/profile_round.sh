#!/bin/bash
# Profiles of one round on the GPU box (run through gpurun): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in passes of
# their own (the TCC block cannot count both at once, MI355X_MICROARCH.md), all on `bench.py --steps 1 --warmup 0 --streams 1`
# (default workload: 500 k pairs of 2x150 bp vs the 3.1 Gbp stand-in).  Results land under gpurun_out/<tag>/; copy what is to be
# kept into profiles/.  Kernel variants are chosen through the environment as everywhere else, e.g.
#   BWAGPU_SEED_MRG=2 tools/profile_round.sh r04_mrg2
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e --no-longread --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > $out/write.log 2>&1
python tools/pmc_summary.py "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes on: $B (500 k pairs of 2x150 bp vs the 3.1 Gbp stand-in, full SA); per launch; bytes = (FETCH_SIZE + WRITE_SIZE) KiB * 1024; Infinity-Cache hits are included" $out/fetch $out/write $out/pmc.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -o s -- $B > $out/sq.log 2>&1
python tools/sq_summary.py $out/sq $out/sq_counters.md "solo batch of 1 M reads (500 k pairs of 2x150 bp) vs the 3.1 Gbp stand-in; $B"
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
ls $out
