# scratch script of the current GPU session (rewritten per session)
mkdir -p gpurun_out/s14
export TMPDIR=/tmp
for d in 4 2 1; do
(timeout 300 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 6 --dense-sa $d 2>&1 | grep config | sed "s/^/[dense-sa $d] /" >> gpurun_out/s14/dense.log)
done
cut -c1-360 gpurun_out/s14/dense.log
