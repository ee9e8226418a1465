#!/bin/bash
# round 4, first GPU call: suite, bench (new defaults + new gates), long-read profiles, kernel stats of the FASTQ->SAM run
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04a; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc $?"; grep SUMMARY $out/bench.log; tail -c 600 $out/bench.json; echo
bash tools/profile_longread.sh r04a/longread > $out/profile_longread.log 2>&1; tail -40 $out/profile_longread.log
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
if [ -n "$P" ] && [ -e $C/sample_1.fq ]; then
  BWAGPU_CLI_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e2e_trace -o e -- bwa_amd/bwa-amd mem -t 16 -K 100000000 $P $C/sample_1.fq $C/sample_2.fq > /dev/null 2> $out/e2e_trace.log
  find $out/e2e_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/e2e_kernel_stats.csv
  rm -rf $out/e2e_trace
  grep "device_sub" $out/e2e_trace.log | tail -4
  head -25 $out/e2e_kernel_stats.csv
fi
ls $out
