#!/bin/bash
# round 4, third GPU call: chain-parallel extension + error-aware read order (parity, A/B), batch-size effect on the headline
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04c; mkdir -p $out
timeout 1200 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-longread --steps 12 --warmup 3 > $out/bench_1m.json 2> $out/bench_1m.log; python -c "
import json; d=json.loads(open('$out/bench_1m.json').read().strip().split(chr(10))[-1]); print('1M/step:', d['value'], d['ms_per_step'], d['stage_ms_solo'])"
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
timeout 300 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --steps 9 "seed_w_err=0" "ext_par=0" "seed_w_err=0 ext_par=0" > $out/variants_short.jsonl 2> $out/variants_short.log; python - <<PY
import json
for l in open('$out/variants_short.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms_solo'), d.get('ms_per_step'), d.get('same_result_as_defaults'), d.get('error'))
PY
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-longread --reads 2000000 --steps 9 --warmup 3 > $out/bench_2m.json 2> $out/bench_2m.log; python -c "
import json; d=json.loads(open('$out/bench_2m.json').read().strip().split(chr(10))[-1]); print('2M/step:', d['value'], d['ms_per_step'], d['stage_ms_solo'])"
timeout 300 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --long-reads 6000 --passes 2 "ext_par=0" "ext_par=2" "ext_par=6" > $out/variants_long.jsonl 2> $out/variants_long.log; python - <<PY
import json
for l in open('$out/variants_long.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms'), d.get('ms_per_pass'), d.get('same_result_as_defaults'), d.get('error'))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/longread_bench.py --reads 6000 > $out/trace.log 2>&1
find $out/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/longread_kernel_stats.csv; rm -rf $out/trace
head -12 $out/longread_kernel_stats.csv | cut -c1-150
READS=1000000 timeout 200 python tools/seed_iter_probe.py "" "seed_w_err=0" > $out/seed_iter.log 2>&1; grep "reads:" $out/seed_iter.log
ls $out
