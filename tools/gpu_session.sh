# Session r6-15: with the extension faster, where is the best split of the chip between the batches in flight?  `share` (percent of a chip-filling launch per
# persistent kernel) and the number of batches in flight, step time over 12 steps each.
mkdir -p gpurun_out/s15
export TMPDIR=/tmp
python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 12"
(timeout 500 python tools/variant_probe.py $P --streams 3 "share=100" "share=70" "share=50" "share=35" > gpurun_out/s15/share3.log 2>&1; echo "rc $?" >> gpurun_out/s15/share3.log)
(timeout 500 python tools/variant_probe.py $P --streams 2 "share=100" "share=60" > gpurun_out/s15/share2.log 2>&1; echo "rc $?" >> gpurun_out/s15/share2.log)
(timeout 500 python tools/variant_probe.py $P --streams 4 "share=50" "share=35" "share=25" > gpurun_out/s15/share4.log 2>&1; echo "rc $?" >> gpurun_out/s15/share4.log)
python - <<'PY'
import json
for f in ("share3","share2","share4"):
    for ln in open(f"gpurun_out/s15/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); print(f, d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"))
PY
