# scratch script of the current GPU session (rewritten per session)
mkdir -p gpurun_out/s36
export TMPDIR=/tmp
python tools/e2e_bench.py --reads 12000000 --pe --threads 16 --streams 3 --env ";BWAGPU_SHARE=50;" > gpurun_out/s36/e2e.log 2>&1
(timeout 300 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 9 "share=50" > gpurun_out/s36/variants.log 2>&1; echo "rc $?" >> gpurun_out/s36/variants.log)
