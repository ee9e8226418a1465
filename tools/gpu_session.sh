# Session r6-8: why is FASTQ->SAM slower inside bench.py than from the stand-alone tool?  The tool with and without a device context in the parent process.
mkdir -p gpurun_out/s8
export TMPDIR=/tmp
(timeout 600 python tools/e2e_bench.py --pe --reads 20000000 --streams 3,3 > gpurun_out/s8/plain.log 2>&1; echo "rc $?" >> gpurun_out/s8/plain.log)
grep "reads/s" gpurun_out/s8/plain.log
(timeout 600 python tools/e2e_bench.py --pe --reads 20000000 --streams 3,3 --hold 1 > gpurun_out/s8/hold1.log 2>&1; echo "rc $?" >> gpurun_out/s8/hold1.log)
grep "reads/s" gpurun_out/s8/hold1.log
