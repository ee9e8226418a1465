# Scratch script of a GPU session (rewritten per session, run as `gpurun -- 'python tools/seed_iter_probe.py > /dev/null 2>&1; bash tools/gpu_session.sh'`:
# the probe builds the 3.1 Gbp index under /tmp/bwa_amd_bench, which the commands below reuse).  The shape of a typical one -- an A/B of library options
# with result digests, then the parity suite:
mkdir -p gpurun_out/sNN
export TMPDIR=/tmp
(timeout 500 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 9 "chain_regs=0" > gpurun_out/sNN/variants.log 2>&1; echo "rc $?" >> gpurun_out/sNN/variants.log)
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/sNN/pytest_parity.log 2>&1; echo "rc $?" >> gpurun_out/sNN/pytest_parity.log)
tail -n 3 gpurun_out/sNN/pytest_parity.log
