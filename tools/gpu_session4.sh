mkdir -p gpurun_out/s4
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_dp_fuzz.py -m gpu -x -q -k "extend_ring or ring_extension" > gpurun_out/s4/pytest_ring.log 2>&1; echo "rc $?" >> gpurun_out/s4/pytest_ring.log)
(timeout 600 python tools/longread_bench.py --reads 6000 > gpurun_out/s4/longread6000.log 2>&1; echo "rc $?" >> gpurun_out/s4/longread6000.log)
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "long or pacbio" > gpurun_out/s4/pytest_long.log 2>&1; echo "rc $?" >> gpurun_out/s4/pytest_long.log)
tail -n 4 gpurun_out/s4/*.log
