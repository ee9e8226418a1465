cd $GRAFT_REPO_ROOT
for mode in prio noprio; do
if [ $mode = noprio ]; then export BWAGPU_SEED_PRIO=0; fi
python bench.py --no-e2e --no-cpu-baseline --steps 9 > gpurun_out/r02_bench_$mode.json 2> gpurun_out/r02_bench_$mode.err; echo "rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r02_bench_$mode.json'))
print('$mode', d['value'], d['ms_per_step'], d['stage_ms_solo'])
PY
done
