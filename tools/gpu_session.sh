# Session r6-35: the sorts of reads with hundreds of regions finished by a bitonic network instead of by counting (option dedup_net): parity, solo stage time, step.
mkdir -p gpurun_out/s35
export TMPDIR=/tmp
(timeout -s KILL 60 python -u tools/dedup_debug.py - "dedup_net=4" > gpurun_out/s35/dflt.log 2>&1; echo "rc $?" >> gpurun_out/s35/dflt.log); tail -2 gpurun_out/s35/dflt.log | cut -c1-200
grep -q "^OK" gpurun_out/s35/dflt.log || exit 0
(timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -x -k "dedup or golden or heavy_reads" > gpurun_out/s35/pytest.log 2>&1; echo "rc $?" >> gpurun_out/s35/pytest.log); tail -3 gpurun_out/s35/pytest.log
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 600 python tools/variant_probe.py $P --steps 18 --streams 3 "dedup_net=0" "dedup_net=65" "dedup_net=33" "dedup_net=0" > gpurun_out/s35/net.log 2>&1; echo "rc $?" >> gpurun_out/s35/net.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s35/net.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_dedup"), d.get("error"))
PY
