# Session r6-7: spread of the FASTQ->SAM rate -- the stand-alone tool three times, then the bench line again.
mkdir -p gpurun_out/s7
export TMPDIR=/tmp
(timeout 700 python tools/e2e_bench.py --pe --reads 20000000 --streams 3,3,3 > gpurun_out/s7/e2e3.log 2>&1; echo "rc $?" >> gpurun_out/s7/e2e3.log)
grep "reads/s" gpurun_out/s7/e2e3.log
(timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/s7/bench.json 2> gpurun_out/s7/bench.err; echo "rc $?" >> gpurun_out/s7/bench.err)
grep -a "SUMMARY\|^rc" gpurun_out/s7/bench.err | tail -2 | cut -c1-400
cp gpurun_out/bench_full.json gpurun_out/s7/bench_full.json 2>/dev/null
