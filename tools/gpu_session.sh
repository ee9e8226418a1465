# Session r6-26: dedup_read_par with 16-byte sort keys: does it run on the device, parity, then timing.  Every command under its own time limit.
mkdir -p gpurun_out/s26
export TMPDIR=/tmp
(timeout -s KILL 40 python -u tools/dedup_debug.py - "" > gpurun_out/s26/dflt.log 2>&1; echo "rc $?" >> gpurun_out/s26/dflt.log); tail -2 gpurun_out/s26/dflt.log | cut -c1-200
(timeout -s KILL 40 python -u tools/dedup_debug.py - "dedup_heavy=0" > gpurun_out/s26/none.log 2>&1; echo "rc $?" >> gpurun_out/s26/none.log); tail -2 gpurun_out/s26/none.log | cut -c1-200
grep -q "^OK" gpurun_out/s26/dflt.log || exit 0
(timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -q -x -k "dedup or golden_regs or medium_short or medium_paired or heavy_reads" > gpurun_out/s26/pytest.log 2>&1; echo "rc $?" >> gpurun_out/s26/pytest.log); tail -3 gpurun_out/s26/pytest.log
grep -q "rc 0" gpurun_out/s26/pytest.log || exit 0
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 300 python tools/dedup_hist_probe.py $P > gpurun_out/s26/hist.json 2> gpurun_out/s26/hist.err; echo "rc $?" >> gpurun_out/s26/hist.err)
(timeout -s KILL 600 python tools/variant_probe.py $P --steps 12 --streams 3 "dedup_heavy=0" "dedup_heavy=2" "dedup_heavy=5" "dedup_stage=64" "dedup_stage=256" "dedup_big=0" > gpurun_out/s26/dd.log 2>&1; echo "rc $?" >> gpurun_out/s26/dd.log)
python - <<'PY'
import json
try:
    d=json.load(open("gpurun_out/s26/hist.json"))
    for r in d["wave_kernel_reads_by_time"]: print(r)
    print(d["stats"])
except Exception as e: print("hist:", e)
for ln in open("gpurun_out/s26/dd.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_dedup"), d.get("error"))
PY
