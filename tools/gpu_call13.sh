set -x
mkdir -p gpurun_out/r04m
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r04m/pytest.log 2>&1; echo "pytest rc $?" | tee -a gpurun_out/r04m/pytest.log
tail -3 gpurun_out/r04m/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-e2e --no-longread --steps 12 --warmup 3 --variants "seed_mrg=0" > gpurun_out/r04m/bench_1m.json 2> gpurun_out/r04m/bench_1m.log; echo "bench rc $?"
python - <<'P'
import json
d=json.load(open('gpurun_out/r04m/bench_1m.json'))
print('value', d['value'], d['ms_per_step'], d['stage_ms_solo'])
print(json.dumps(d.get('variants'))[:1200])
print(json.dumps(d.get('roofline'))[:800])
P
