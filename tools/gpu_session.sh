# Session r6-44: the CIGAR stage's two LDS tiers: how many regions the first leaves to the second, and what each costs.
mkdir -p gpurun_out/s44
export TMPDIR=/tmp
(READS=1000000 timeout -s KILL 500 python tools/cigar_probe.py "" "cig_tiers=1" > gpurun_out/s44/cig.log 2>&1; echo "rc $?" >> gpurun_out/s44/cig.log)
grep "filter" gpurun_out/s44/cig.log | cut -c1-330
