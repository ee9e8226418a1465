# Session r6-34: the seeding kernels' interval-stack entries in LDS at 12 instead of 16 bytes (no x1): parity, solo seeding time for several depths, step time.
mkdir -p gpurun_out/s34
export TMPDIR=/tmp
(timeout -s KILL 60 python -u tools/dedup_debug.py - "" > gpurun_out/s34/dflt.log 2>&1; echo "rc $?" >> gpurun_out/s34/dflt.log); tail -2 gpurun_out/s34/dflt.log | cut -c1-200
grep -q "^OK" gpurun_out/s34/dflt.log || exit 0
(timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or medium_short or short_read_batches or edge" > gpurun_out/s34/pytest.log 2>&1; echo "rc $?" >> gpurun_out/s34/pytest.log); tail -3 gpurun_out/s34/pytest.log
timeout -s KILL 300 python tools/seed_iter_probe.py > gpurun_out/s34/probe.log 2>&1
grep -n "interval-stack\|k_seed(+k_seed3)" gpurun_out/s34/probe.log | cut -c1-330
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 900 python tools/variant_probe.py $P --steps 18 --streams 3 "seed_lds_ent=7" "seed_lds_ent=8" "seed_lds_ent=10" "seed_lds_ent=11" "seed_lds_ent=13" > gpurun_out/s34/lds.log 2>&1; echo "rc $?" >> gpurun_out/s34/lds.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s34/lds.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_seed"), d.get("error"))
PY
