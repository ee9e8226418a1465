#!/bin/bash
# Kernel trace of the FASTQ -> SAM run (`bwa-amd mem` on 6 M reads of the bench genome): what the CIGAR, mate-rescue and packing kernels cost next to the
# hot path, under three batches in flight.   tools/profile_e2e.sh <tag>
tag=${1:-e2e}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 400 python tools/e2e_bench.py --pe --reads 6000000 > $out/e2e_bench.log 2>&1; tail -3 $out/e2e_bench.log | cut -c1-400
C=/tmp/bwa_amd_bench; P=$(ls $C/*.bwt | head -1); P=${P%.bwt}
BWAGPU_CLI_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o e -- bwa_amd/bwa-amd mem -t 16 -K 100000000 $P $C/e2e_1.fq $C/e2e_2.fq > /dev/null 2> $out/trace.log
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/e2e_kernel_stats.csv
grep "device_sub" $out/trace.log | tail -4 | cut -c1-400
head -16 $out/e2e_kernel_stats.csv | cut -c1-150
READS=1000000 timeout 200 python tools/seed_iter_probe.py "" > $out/seed_lane_slots.log 2>&1; tail -3 $out/seed_lane_slots.log | cut -c1-900
