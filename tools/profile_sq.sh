#!/bin/bash
# The instruction-mix half of tools/profile_round.sh alone (kernel trace + SQ counters of a solo 1 M-read batch): what a change of a kernel's inner
# loop is judged by between full profiles.   tools/profile_sq.sh <tag>
tag=${1:-sq}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e --no-longread --no-pmc"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -o s -- $B > $out/sq.log 2>&1
python tools/sq_summary.py $out/sq $out/sq_counters.md "solo batch of 1 M reads (500 k pairs of 2x150 bp) vs the 3.1 Gbp stand-in; $B"
find $out -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
head -30 $out/kernel_stats.csv | cut -c1-200
cat $out/sq_counters.md | cut -c1-260
