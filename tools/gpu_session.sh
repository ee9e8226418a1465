# Session r6-42: validation of the final tree: the -m gpu suite, smoke, the default bench line, kernel trace + PMC of the short-read batch.
mkdir -p gpurun_out/s42
export TMPDIR=/tmp
(timeout -s KILL 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s42/pytest_gpu.log 2>&1; echo "rc $?" >> gpurun_out/s42/pytest_gpu.log); grep -n "passed\|failed" gpurun_out/s42/pytest_gpu.log | tail -2
(timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/s42/smoke.log 2>&1; echo "rc $?" >> gpurun_out/s42/smoke.log); tail -2 gpurun_out/s42/smoke.log
(timeout -s KILL 900 python bench.py --steps 20 --warmup 5 > gpurun_out/s42/bench.json 2> gpurun_out/s42/bench.log; echo "rc $?" >> gpurun_out/s42/bench.log); tail -1 gpurun_out/s42/bench.log; cp gpurun_out/bench_full.json gpurun_out/s42/bench_full.json; cp gpurun_out/bench_variants.json gpurun_out/s42/ 2>/dev/null
timeout -s KILL 900 bash tools/profile_round.sh s42/short > gpurun_out/s42/profile_short.log 2>&1
grep -n "k_seed" gpurun_out/s42/short/kernel_stats.csv | cut -c1-70,100-200
python - <<'PY'
import json
d=json.load(open('gpurun_out/s42/bench_full.json'))
print(d['value'], d['ms_per_step'], d['bench_wall_s'], d['stage_ms_solo'])
print(d['roofline']['frac'], d['roofline']['frac_requests'], d['roofline']['kernel_ms'])
print(json.dumps(d['summary'])[:600])
PY
