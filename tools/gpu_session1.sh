mkdir -p gpurun_out/s1
export TMPDIR=/tmp
(timeout 120 tools/randbw4 > gpurun_out/s1/randbw4.log 2>&1; echo "rc $?" >> gpurun_out/s1/randbw4.log)
(timeout 600 python tools/seed_iter_probe.py "" "seed_budget=0" > gpurun_out/s1/seed_probe.log 2>&1; echo "rc $?" >> gpurun_out/s1/seed_probe.log)
(timeout 400 python tools/e2e_bench.py --pe --reads 6000000 > gpurun_out/s1/e2e.log 2>&1; echo "rc $?" >> gpurun_out/s1/e2e.log)
tail -3 gpurun_out/s1/*.log
