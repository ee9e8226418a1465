# Session r6-5: the GPU suite with the new tests; FASTQ->SAM at 20 M reads with 3 / 4 / 5 handles now that the result pool holds 16 GiB.
mkdir -p gpurun_out/s5
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s5/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s5/pytest_gpu.log)
tail -n 4 gpurun_out/s5/pytest_gpu.log
(timeout 700 python tools/e2e_bench.py --pe --reads 20000000 --streams 3,4,5,3,4 --env "BWAGPU_CLI_AHEAD=0" > gpurun_out/s5/e2e_streams.log 2>&1; echo "rc $?" >> gpurun_out/s5/e2e_streams.log)
grep "reads/s" gpurun_out/s5/e2e_streams.log
