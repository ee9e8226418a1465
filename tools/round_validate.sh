#!/bin/bash
# Full validation of a round on the GPU box (run through gpurun, ~8 box-minutes): the -m gpu suite, the default bench line, kernel trace + PMC
# of the short-read batch (tools/profile_round.sh) and of the long-read batch (tools/profile_longread.sh).  Copy what is to be kept into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2400 -- "bash tools/round_validate.sh"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r05}; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc $?"; grep SUMMARY $out/bench.log
bash tools/profile_round.sh ${1:-r05}/short > $out/profile_short.log 2>&1; tail -5 $out/profile_short.log
bash tools/profile_longread.sh ${1:-r05}/longread > $out/profile_longread.log 2>&1; grep "longread\]" gpurun_out/${1:-r05}/longread/trace.log | tail -4; head -12 gpurun_out/${1:-r05}/longread/kernel_stats.csv | cut -c1-140
ls $out $out/short
