# Session r6-43: FASTQ -> SAM on 10 M pairs, the final tree, five runs in a row on one box (the figure's run-to-run spread).
mkdir -p gpurun_out/s43
export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  (timeout -s KILL 300 python tools/e2e_bench.py --reads 20000000 --pe > gpurun_out/s43/e2e_$i.log 2>&1; echo "rc $?" >> gpurun_out/s43/e2e_$i.log)
  grep "reads/s" gpurun_out/s43/e2e_$i.log | cut -c1-200
done
