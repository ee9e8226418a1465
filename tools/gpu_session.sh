# Session r6-10: the slots' warm-up batch without the result-sized stages, on / off (three runs each, interleaved).
mkdir -p gpurun_out/s10
export TMPDIR=/tmp
(timeout 900 python tools/e2e_bench.py --pe --reads 20000000 --streams 3 --env ";BWAGPU_CLI_WARMUP_READS=0;;BWAGPU_CLI_WARMUP_READS=0;;BWAGPU_CLI_WARMUP_READS=0" > gpurun_out/s10/warm_ab.log 2>&1; echo "rc $?" >> gpurun_out/s10/warm_ab.log)
grep "reads/s" gpurun_out/s10/warm_ab.log
