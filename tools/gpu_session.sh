# scratch script of the current GPU session (rewritten per session)
mkdir -p gpurun_out/s12
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s12/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s12/pytest_gpu.log)
(timeout 400 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 6 "ext_occ=4" > gpurun_out/s12/variants.log 2>&1; echo "rc $?" >> gpurun_out/s12/variants.log)
tail -n 4 gpurun_out/s12/pytest_gpu.log; cut -c1-330 gpurun_out/s12/variants.log
