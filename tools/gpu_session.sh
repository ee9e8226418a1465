# Scratch script of a GPU session (rewritten per session).  Session r6-1: the bench line of the tree as it stands, then FASTQ->SAM A/Bs
# (slots ahead of the finalize stage 0 vs 2; 12 M vs 20 M reads), then the parity suite.
mkdir -p gpurun_out/s1
export TMPDIR=/tmp
(timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; echo "rc $?" >> gpurun_out/s1/bench.err)
tail -n 2 gpurun_out/s1/bench.err
(timeout 600 python tools/e2e_bench.py --pe --reads 20000000 --env ";BWAGPU_CLI_AHEAD=0;BWAGPU_CLI_AHEAD=2;BWAGPU_CLI_AHEAD=0;BWAGPU_CLI_AHEAD=4" > gpurun_out/s1/e2e_ab.log 2>&1; echo "rc $?" >> gpurun_out/s1/e2e_ab.log)
grep "reads/s" gpurun_out/s1/e2e_ab.log
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s1/pytest_gpu.log)
tail -n 3 gpurun_out/s1/pytest_gpu.log
