#!/bin/bash
# First GPU call of a round (run through gpurun, ~20 box-minutes): what was built behind switches without a GPU at hand gets its
# numbers in one go.  Results land under gpurun_out/<tag>/; copy what is to be kept into profiles/.
#   /usr/local/graft/bin/gpurun --timeout 2100 -- 'bash tools/round_start.sh r05'
# 0. if tools/quick_gpu_prepare.py has been run on the CPU box: the defaults and every switchable kernel form against `bwa mem` digests (seconds)
# 1. the -m gpu suite (parity first: a failing test ends the script -- fix that before anything is timed)
# 2. the full default bench line; its `variants` object A/Bs option settings (bwagpu_set_option names) against the defaults on the same batches
#    (short reads: seed_mrg=0, ext_occ=4; long reads: the round-3 kernel forms together and each alone) with a digest that must equal the defaults'
# 3. kernel trace + FETCH_SIZE / WRITE_SIZE + SQ counters of the default configuration (tools/profile_round.sh); to profile a
#    variant that won in step 2, run e.g.  BWAGPU_SEED_MRG=0 bash tools/profile_round.sh r05_mrg0  in a later call
# 4. kernel stats of `bwa-amd mem` itself on the bench's FASTQ files (hot path + CIGAR + mate-rescue kernels side by side)
# Further configurations for step 2's probe can be given as extra arguments, e.g.  "seed_mrg=0 ptab_m=12".
tag=${1:-r05}; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
if [ -e tests/_data/quick/expected.txt ]; then bash tools/quick_gpu_check.sh > $out/quick.log 2>&1; bash tools/quick_gpu_variants.sh >> $out/quick.log 2>&1; grep -c OK $out/quick.log; grep MISMATCH $out/quick.log; fi
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; echo "GPU SUITE FAILED"; exit 1; }
tail -3 $out/pytest_gpu.log
timeout 800 python bench.py > $out/bench.json 2> $out/bench.log; echo "bench rc $?"; tail -c 800 $out/bench.json; echo
C=/tmp/bwa_amd_bench
P=$(ls $C/*.bwt 2>/dev/null | head -1); P=${P%.bwt}
if [ -n "$P" ] && [ $# -gt 0 ]; then
  timeout 420 python tools/variant_probe.py --prefix $P --codes $P.codes.npy --batch-files $C/variant_batch0.npy,$C/variant_batch1.npy,$C/variant_batch2.npy --steps 6 "$@" > $out/variants_extra.jsonl 2> $out/variants_extra.log
  cat $out/variants_extra.jsonl
fi
# 4. the kernels of the FASTQ->SAM run (never profiled so far: what do the CIGAR and mate-rescue kernels cost next to the hot path?)
if [ -n "$P" ] && [ -e $C/sample_1.fq ]; then
  BWAGPU_CLI_TRACE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/e2e_trace -o e -- bwa_amd/bwa-amd mem -t 16 -K 100000000 $P $C/sample_1.fq $C/sample_2.fq > /dev/null 2> $out/e2e_trace.log
  find $out/e2e_trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/e2e_kernel_stats.csv
  grep "device_sub" $out/e2e_trace.log | tail -5
fi
bash tools/profile_round.sh $tag > $out/profile.log 2>&1
ls $out
