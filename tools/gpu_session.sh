# scratch script of the current GPU session (rewritten per session)
mkdir -p gpurun_out/s21
export TMPDIR=/tmp
(timeout 400 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 9 "seed_lds_ent=7" "seed_lds_ent=9" > gpurun_out/s21/variants.log 2>&1; echo "rc $?" >> gpurun_out/s21/variants.log)
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s21/pytest_parity.log 2>&1; echo "rc $?" >> gpurun_out/s21/pytest_parity.log)
tail -n 3 gpurun_out/s21/pytest_parity.log
