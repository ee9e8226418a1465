#!/bin/bash
# round 4, fourth GPU call: per-chain hot spots of the extension removed (parity, A/B with and without the chain-parallel route), k_seed's iterations per read, FASTQ->SAM split
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04d; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_opt_fuzz.py tests/test_dp_fuzz.py -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-longread --steps 12 --warmup 3 > $out/bench_1m.json 2> $out/bench_1m.log; python -c "
import json; d=json.loads(open('$out/bench_1m.json').read().strip().split(chr(10))[-1]); print('1M/step:', d['value'], d['ms_per_step'], d['stage_ms_solo'])"
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
timeout 300 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --steps 9 "ext_par=0" "ext_par=64" > $out/variants_short.jsonl 2> $out/variants_short.log; python - <<PY
import json
for l in open('$out/variants_short.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms_solo'), d.get('ms_per_step'), d.get('same_result_as_defaults'), d.get('error'))
PY
timeout 300 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --long-reads 6000 --passes 2 "ext_par=0" "ext_par=8" > $out/variants_long.jsonl 2> $out/variants_long.log; python - <<PY
import json
for l in open('$out/variants_long.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms'), d.get('ms_per_pass'), d.get('same_result_as_defaults'), d.get('error'))
PY
READS=1000000 timeout 200 python tools/seed_iter_probe.py "" > $out/seed_iter.log 2>&1; grep -A1 "reads:" $out/seed_iter.log
timeout 300 python tools/e2e_bench.py --pe --reads 8000000 > $out/e2e.log 2>&1; grep "e2e\]" $out/e2e.log | tail -6
ls $out
