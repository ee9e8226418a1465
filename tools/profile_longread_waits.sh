#!/bin/bash
# Where the long-read DP kernels' wave cycles go: SQ wait / active counters and resident waves, one rocprofv3 --pmc pass over one batch.   tools/profile_longread_waits.sh <tag> <reads>
tag=${1:-longwaits}; reads=${2:-6000}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python tools/longread_bench.py --reads $reads"
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/p1 -o c -- $B > $out/p1.log 2>&1; echo "pass rc $?"
python tools/sq_summary.py $out/p1 $out/waits_$reads.md "one batch of $reads reads of 10 kb, -x pacbio; $B" 2>&1 | grep "k_extend_wave\|k_dedup_wave\|kernel\|---" | cut -c1-300
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_MISC --output-format csv -d $out/p2 -o c -- $B > $out/p2.log 2>&1; echo "pass rc $?"
python tools/sq_summary.py $out/p2 $out/pipes_$reads.md "one batch of $reads reads of 10 kb, -x pacbio; $B" 2>&1 | grep "k_extend_wave\|k_dedup_wave\|kernel\|---" | cut -c1-300
rm -rf $out/p1 $out/p2
