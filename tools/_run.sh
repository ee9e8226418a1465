cd $GRAFT_REPO_ROOT
for s in 2 4 6; do
python bench.py --no-e2e --no-cpu-baseline --streams $s --steps 12 > gpurun_out/r02_bench_s$s.json 2> gpurun_out/r02_bench_s$s.err
python - <<PY
import json
d=json.load(open('gpurun_out/r02_bench_s$s.json'))
print('streams',$s, d['value'], d['ms_per_step'])
PY
done
