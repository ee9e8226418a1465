# Session r6-28: the list launch of k_dedup_wave takes 69 ms inside bench.py and 0.7 ms inside tools/variant_probe.py: what is different?
mkdir -p gpurun_out/s28
export TMPDIR=/tmp
B="--steps 3 --warmup 1 --streams 1 --no-cpu-baseline --no-e2e --no-longread --no-pmc"
getd() { python - "$1" <<'PY'
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d=json.loads(ln); print(sys.argv[1], d.get("config") if isinstance(d.get("config"),str) else "", "ms_dedup", (d.get("stage_ms_solo") or {}).get("ms_dedup"), "ms/step", d.get("ms_per_step"))
PY
}
(timeout -s KILL 400 python bench.py $B --variants "" > gpurun_out/s28/a.json 2> gpurun_out/s28/a.log); getd gpurun_out/s28/a.json
(BWAGPU_DEDUP_HEAVY=0 timeout -s KILL 300 python bench.py $B --variants "" > gpurun_out/s28/b.json 2> gpurun_out/s28/b.log); getd gpurun_out/s28/b.json
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 3 --streams 1"
(timeout -s KILL 300 python tools/variant_probe.py $P > gpurun_out/s28/c.json 2> gpurun_out/s28/c.log); getd gpurun_out/s28/c.json
(timeout -s KILL 300 python -c "import torch, runpy, sys; torch.cuda.init(); x = torch.zeros(8, device='cuda'); sys.argv = ['variant_probe.py'] + '$P'.split(); runpy.run_path('tools/variant_probe.py', run_name='__main__')" > gpurun_out/s28/d.json 2> gpurun_out/s28/d.log); getd gpurun_out/s28/d.json
(timeout -s KILL 300 python bench.py $B --variants "" --dense-sa 0 > gpurun_out/s28/e.json 2> gpurun_out/s28/e.log); getd gpurun_out/s28/e.json
(timeout -s KILL 300 python tools/variant_probe.py $P --dense-sa 0 > gpurun_out/s28/f.json 2> gpurun_out/s28/f.log); getd gpurun_out/s28/f.json
