#!/bin/bash
# Profiles of the long-read batch (BASELINE configs[4]: 6000 reads of 10 kb, -x pacbio, 3.1 Gbp stand-in) on the GPU box (run through gpurun):
# kernel trace + stats, SQ instruction counters, FETCH_SIZE / WRITE_SIZE in passes of their own.  Needs bench.py's index cache on the box
# (run bench.py, or tools/longread_bench.py once, earlier in the same gpurun call).  Results land under gpurun_out/<tag>/; copy what is to be
# kept into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/profile_longread.sh r04_longread'
tag=${1:-r04_longread}; reads=${2:-6000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python tools/longread_bench.py --reads $reads --cigars"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- $B > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq -o s -- $B > $out/sq.log 2>&1
python tools/sq_summary.py $out/sq $out/sq_counters.md "one batch of $reads reads of 10 kb, -x pacbio, vs the 3.1 Gbp stand-in; $B"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- $B > $out/write.log 2>&1
python tools/pmc_summary.py "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes on: $B; per launch; raw counters (FETCH_SIZE counts 64 bytes per request)" $out/fetch $out/write $out/pmc.json
grep "longread" $out/trace.log | tail -4
head -30 $out/kernel_stats.csv
rm -rf $out/trace $out/sq $out/fetch $out/write      # (the raw per-dispatch tables are tens of MB; the summaries are what is kept)
ls $out
