# Session r6-11: long reads -- the one-pass row form of the ring-mode extension (ext_blk) against the pass-per-64-columns form; DP fuzz and long-read parity on the device.
mkdir -p gpurun_out/s11
export TMPDIR=/tmp
(timeout 900 python tools/longread_ab.py --reads 6000 --rounds 4 "ext_blk=0" "ext_blk=1" > gpurun_out/s11/ab6000.log 2>&1; echo "rc $?" >> gpurun_out/s11/ab6000.log)
tail -n 6 gpurun_out/s11/ab6000.log
(timeout 900 python -m pytest tests/test_dp_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "ring or pacbio or long" > gpurun_out/s11/pytest_long.log 2>&1; echo "rc $?" >> gpurun_out/s11/pytest_long.log)
tail -n 3 gpurun_out/s11/pytest_long.log
(timeout 600 python tools/longread_ab.py --reads 10000 --rounds 2 "ext_blk=0" "ext_blk=1" > gpurun_out/s11/ab10000.log 2>&1; echo "rc $?" >> gpurun_out/s11/ab10000.log)
tail -n 4 gpurun_out/s11/ab10000.log
