mkdir -p gpurun_out/s2
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s2/pytest_gpu.log)
(timeout 400 python tools/e2e_bench.py --pe --reads 6000000 > gpurun_out/s2/e2e.log 2>&1; echo "rc $?" >> gpurun_out/s2/e2e.log)
(timeout 400 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 6 "ptab_m=11" "ptab_m=12" "seed_lds_ent=7" > gpurun_out/s2/variants.log 2>&1; echo "rc $?" >> gpurun_out/s2/variants.log)
tail -n 3 gpurun_out/s2/pytest_gpu.log gpurun_out/s2/e2e.log gpurun_out/s2/variants.log
