# Session r6-40: the seeding kernel's iteration budget (reads given up to the task kernels after that many iterations): solo stage time and step time.
mkdir -p gpurun_out/s40
export TMPDIR=/tmp
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 30"
(timeout -s KILL 900 python tools/variant_probe.py $P --streams 3 "seed_budget=10240" "seed_budget=8192" "seed_budget=10240" "seed_budget=9216" "seed_budget=11264" "seed_budget=8192" "seed_budget=10240" > gpurun_out/s40/budget.log 2>&1; echo "rc $?" >> gpurun_out/s40/budget.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s40/budget.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_seed"), d.get("error"))
PY
