# Session r6-29: validation of the tree with the new de-duplication path: A/B first, then the -m gpu suite, smoke, the default bench line, kernel trace + PMC.
mkdir -p gpurun_out/s29
export TMPDIR=/tmp
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 300 python tools/variant_probe.py $P --steps 12 --streams 3 "dedup_heavy=0" > gpurun_out/s29/dd.log 2>&1; echo "rc $?" >> gpurun_out/s29/dd.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s29/dd.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_dedup"), d.get("error"))
PY
(timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s29/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s29/pytest_gpu.log); tail -3 gpurun_out/s29/pytest_gpu.log
(timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s29/smoke.log 2>&1; echo "rc $?" >> gpurun_out/s29/smoke.log); tail -2 gpurun_out/s29/smoke.log
(timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s29/bench.json 2> gpurun_out/s29/bench.log; echo "rc $?" >> gpurun_out/s29/bench.log); tail -1 gpurun_out/s29/bench.log; cp gpurun_out/bench_full.json gpurun_out/s29/bench_full.json; cp gpurun_out/bench_variants.json gpurun_out/s29/ 2>/dev/null
timeout -s KILL 900 bash tools/profile_round.sh s29/short > gpurun_out/s29/profile_short.log 2>&1
head -16 gpurun_out/s29/short/kernel_stats.csv | cut -c1-150
