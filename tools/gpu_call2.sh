#!/bin/bash
# round 4, second GPU call: task-based pass 1 of long reads (parity + time), where k_seed's lane-slots go (and how its time scales with the batch)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04b; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_opt_fuzz.py -m gpu -x -q -k "pacbio or paired_and_long or option_fuzz" > $out/pytest_long.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_long.log
timeout 300 python tools/longread_bench.py --reads 6000 > $out/longread.log 2>&1; grep "longread\]" $out/longread.log | tail -3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/longread_bench.py --reads 6000 > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/longread_kernel_stats.csv; rm -rf $out/trace
head -14 $out/longread_kernel_stats.csv | cut -c1-150
READS=250000,500000,1000000,2000000 timeout 400 python tools/seed_iter_probe.py "" > $out/seed_iter.log 2>&1; grep "reads:" $out/seed_iter.log
ls $out
