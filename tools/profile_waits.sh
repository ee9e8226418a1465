#!/bin/bash
# Where the hot-path kernels' wave cycles go (SQ wait / active counters), one rocprofv3 --pmc pass over a solo 1 M-read batch.  The profiler is run under
# `timeout -s KILL`: with some counter groups (TA_*) rocprofv3 7.2 aborts at start-up and then hangs in its own finalisation.   tools/profile_waits.sh <tag>
tag=${1:-waits}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e --no-longread --no-pmc"
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM --output-format csv -d $out/p1 -o c -- $B > $out/p1.log 2>&1; echo "pass rc $?"
tail -3 $out/p1.log | cut -c1-200
python tools/sq_summary.py $out/p1 $out/waits.md "solo batch of 1 M reads; $B" 2>&1 | grep "k_seed<true\|k_extend_wave\|k_chain_wave\|k_dedup\|kernel\|---" | cut -c1-300
