# Session r6-37: share of the chip per persistent kernel with three batches in flight, after the de-duplication rewrite; and four batches in flight.
mkdir -p gpurun_out/s37
export TMPDIR=/tmp
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 18"
(timeout -s KILL 600 python tools/variant_probe.py $P --streams 3 "share=35" "share=50" "share=65" "share=80" "share=100" > gpurun_out/s37/share3.log 2>&1; echo "rc $?" >> gpurun_out/s37/share3.log)
(timeout -s KILL 400 python tools/variant_probe.py $P --streams 4 "share=35" "share=50" > gpurun_out/s37/share4.log 2>&1; echo "rc $?" >> gpurun_out/s37/share4.log)
(timeout -s KILL 400 python tools/variant_probe.py $P --streams 2 "share=50" "share=100" > gpurun_out/s37/share2.log 2>&1; echo "rc $?" >> gpurun_out/s37/share2.log)
python - <<'PY'
import json
for f in ("share3","share4","share2"):
    for ln in open(f"gpurun_out/s37/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); print(f, d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"))
PY
