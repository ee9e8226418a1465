# Session r6-45: k_chain_wave at four / five (the tree) / six waves per SIMD (__launch_bounds__): solo stage time and step time, alternate-library builds.
mkdir -p gpurun_out/s45
export TMPDIR=/tmp
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy --steps 24 --streams 3"
for v in cwocc4 tree cwocc6 tree cwocc4; do
  L=""; [ $v != tree ] && L="--lib bwa_amd/csrc/libbwagpu_$v.so"
  (timeout -s KILL 300 python tools/variant_probe.py $P $L > gpurun_out/s45/$v.log 2>&1; echo "rc $?" >> gpurun_out/s45/$v.log)
  python - $v <<'PY'
import json,sys
for ln in open(f"gpurun_out/s45/{sys.argv[1]}.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(sys.argv[1], d.get("ms_per_step"), d.get("Mreads_s"), d.get("result_sha256_16"), d.get("stage_ms_solo",{}).get("ms_chain"), d.get("error"))
PY
done
