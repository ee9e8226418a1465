# scratch script of the current GPU session (rewritten per session)
mkdir -p gpurun_out/s16
export TMPDIR=/tmp
(timeout 700 python tools/variant_probe.py --prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 18 "share=50" "" "share=50" "share=40" "share=60" "share=34" "share=50 seed_grid=512" "" > gpurun_out/s16/variants.log 2>&1; echo "rc $?" >> gpurun_out/s16/variants.log)
