# Session r6-32: k_publish's list of heavy reads for k_publish_blk<LIST> (option publish_heavy): parity, solo stage time, step time.
mkdir -p gpurun_out/s32
export TMPDIR=/tmp
(timeout -s KILL 60 python -u tools/dedup_debug.py - "" > gpurun_out/s32/dflt.log 2>&1; echo "rc $?" >> gpurun_out/s32/dflt.log); tail -2 gpurun_out/s32/dflt.log | cut -c1-200
grep -q "^OK" gpurun_out/s32/dflt.log || exit 0
(timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -x -k "publish or golden or medium_short or heavy_reads" > gpurun_out/s32/pytest.log 2>&1; echo "rc $?" >> gpurun_out/s32/pytest.log); tail -3 gpurun_out/s32/pytest.log
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 600 python tools/variant_probe.py $P --steps 18 --streams 3 "publish_heavy=0" "publish_heavy=16" "publish_heavy=64" "publish_heavy=0" > gpurun_out/s32/pub.log 2>&1; echo "rc $?" >> gpurun_out/s32/pub.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s32/pub.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_publish"), d.get("error"))
PY
