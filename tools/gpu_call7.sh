#!/bin/bash
# round 4, seventh GPU call: iteration budget of the lane-per-read seeding kernel (parity, A/B over budgets)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_opt_fuzz.py -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-longread --steps 12 --warmup 3 > $out/bench_1m.json 2> $out/bench_1m.log; python -c "
import json; d=json.loads(open('$out/bench_1m.json').read().strip().split(chr(10))[-1]); print('1M/step:', d['value'], d['ms_per_step'], d['stage_ms_solo'], d['roofline']['frac'])"
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
timeout 400 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --steps 9 "seed_budget=0" "seed_budget=2048" "seed_budget=3072" "seed_budget=8192" > $out/variants_short.jsonl 2> $out/variants_short.log; python - <<PY
import json
for l in open('$out/variants_short.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms_solo'), d.get('ms_per_step'), d.get('same_result_as_defaults'), d.get('error'))
PY
READS=250000,1000000 timeout 300 python tools/seed_iter_probe.py "" > $out/seed_iter.log 2>&1; grep -A1 "reads:" $out/seed_iter.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --steps 1 --warmup 0 --streams 1 --no-cpu-baseline --no-e2e --no-longread > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; rm -rf $out/trace
grep "k_seed" $out/kernel_stats.csv | cut -c1-170
ls $out
