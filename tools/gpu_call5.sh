#!/bin/bash
# round 4, fifth GPU call: k_seed's read order by 12-mer repetitiveness + errors (A/B), per-read wave times of the long-read DP kernels
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04e; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_opt_fuzz.py -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $out/pytest_gpu.log
timeout 300 python bench.py --no-cpu-baseline --no-e2e --no-longread --steps 12 --warmup 3 > $out/bench_1m.json 2> $out/bench_1m.log; python -c "
import json; d=json.loads(open('$out/bench_1m.json').read().strip().split(chr(10))[-1]); print('1M/step:', d['value'], d['ms_per_step'], d['stage_ms_solo'])"
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
timeout 300 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --steps 9 "seed_w_err=0" > $out/variants_short.jsonl 2> $out/variants_short.log; python - <<PY
import json
for l in open('$out/variants_short.jsonl'):
    d=json.loads(l); print(d.get('config'), d.get('stage_ms_solo'), d.get('ms_per_step'), d.get('same_result_as_defaults'), d.get('error'))
PY
READS=1000000 timeout 200 python tools/seed_iter_probe.py "" "seed_w_err=0" > $out/seed_iter.log 2>&1; grep -A1 "reads:" $out/seed_iter.log
timeout 300 python tools/longread_bench.py --reads 6000 > $out/longread.log 2>&1; grep "longread\]" $out/longread.log
ls $out
