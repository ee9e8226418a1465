# Session r6-13 (the cheaper test: running scalars, every second row): rows that cannot change the result are not computed (k_extend_wave's window rows) -- stage times, digest, DP fuzz and parity on the device.
mkdir -p gpurun_out/s13
export TMPDIR=/tmp
(timeout 600 python tools/ext_pack_probe.py 0 0 > gpurun_out/s13/probe.log 2>&1; echo "rc $?" >> gpurun_out/s13/probe.log)
grep -a "ext_pack\|stats run\|rc " gpurun_out/s13/probe.log
(timeout 900 python -m pytest tests/test_dp_fuzz.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/s13/pytest.log 2>&1; echo "rc $?" >> gpurun_out/s13/pytest.log)
tail -n 3 gpurun_out/s13/pytest.log
(timeout 500 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 9 > gpurun_out/s13/step.log 2>&1; echo "rc $?" >> gpurun_out/s13/step.log)
tail -n 2 gpurun_out/s13/step.log | cut -c1-400
