# Session r6-31: raised issue priority for the list launches of k_dedup_wave (option dedup_prio): step time with three batches in flight, and the stage's
# time batch by batch inside the FASTQ -> SAM run, with and without.
mkdir -p gpurun_out/s31
export TMPDIR=/tmp
timeout -s KILL 300 python tools/seed_iter_probe.py > /dev/null 2>&1
P="--prefix /tmp/bwa_amd_bench/g3100000000_s42 --codes /tmp/bwa_amd_bench/g3100000000_s42.codes.npy"
(timeout -s KILL 400 python tools/variant_probe.py $P --steps 18 --streams 3 "dedup_prio=0" "dedup_prio=1" "dedup_prio=0" > gpurun_out/s31/dd.log 2>&1; echo "rc $?" >> gpurun_out/s31/dd.log)
python - <<'PY'
import json
for ln in open("gpurun_out/s31/dd.log"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"], d.get("ms_per_step"), d.get("Mreads_s"), d.get("same_result_as_defaults"), d.get("stage_ms_solo",{}).get("ms_dedup"), d.get("error"))
PY
for e in "BWAGPU_CLI_TRACE=1" "BWAGPU_CLI_TRACE=1 BWAGPU_DEDUP_PRIO=0" "BWAGPU_CLI_TRACE=1" "BWAGPU_CLI_TRACE=1 BWAGPU_DEDUP_PRIO=0"; do
  (timeout -s KILL 300 python tools/e2e_bench.py --reads 20000000 --pe --env "$e" > gpurun_out/s31/e2e.log 2>&1; echo "rc $?" >> gpurun_out/s31/e2e.log)
  echo "== $e"; grep "reads/s" gpurun_out/s31/e2e.log | cut -c1-200
  grep "stage ms" gpurun_out/s31/e2e.log | awk '{for(i=1;i<=NF;i++) if($i=="dedup") d[NR]=$(i+1); } END {n=0;s=0;m=0; for(k in d){n++; s+=d[k]; if(d[k]>m)m=d[k]} print "dedup under sharing: batches", n, "mean", s/n, "max", m}'
done
