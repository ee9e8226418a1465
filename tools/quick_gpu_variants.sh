#!/bin/bash
# Companion of quick_gpu_check.sh: the library's options (bwagpu_config.h; a handle takes BWAGPU_<NAME> from the environment when it is created) on
# hardware, each setting against the same `bwa mem` digests (seconds per run).
#   bash tools/quick_gpu_variants.sh            -> the round's switches;  extra arguments: "pe:ENV=1 ENV2=2" / "long:ENV=1"
Q=tests/_data/quick; P=tests/golden/g200k; rc=0
body() { grep -av '^@PG' | sha256sum | cut -d' ' -f1; }
E1=$(sed -n 1p $Q/expected.txt); E2=$(sed -n 2p $Q/expected.txt)
run() {  # leg, env settings
  local leg=$1; shift
  if [ $leg = pe ]; then d=$(env "$@" timeout 30 bwa_amd/bwa-amd mem -t 8 -K 1500000 $P $Q/r1.fq $Q/r2.fq 2>$Q/v.err | body); e=$E1
  else d=$(env "$@" timeout 30 bwa_amd/bwa-amd mem -t 8 -x pacbio $P $Q/long.fq 2>$Q/v.err | body); e=$E2; fi
  t=$(grep -o "device [0-9.]* s" $Q/v.err | tail -1)
  [ "$d" = "$e" ] && echo "$leg $* OK ($t)" || { echo "$leg $* MISMATCH $d"; tail -3 $Q/v.err; rc=1; }
}
if [ $# -gt 0 ]; then for a in "$@"; do run ${a%%:*} ${a#*:}; done; exit $rc; fi
run pe BWAGPU_SEED_MRG=0
run pe BWAGPU_OCC32=0
run pe BWAGPU_DEDUP_WAVE=1
run pe BWAGPU_EXT_OCC=4
run pe BWAGPU_SEED_BUDGET=200
run pe BWAGPU_SEED_BUDGET=0
run long BWAGPU_SEED_MRG=0
run long BWAGPU_SEED_TASKS=0
run long BWAGPU_PUBLISH_BLK=0
run long BWAGPU_SEEDSW_LDS=0
run long BWAGPU_DEDUP_BLK=0
run long BWAGPU_SEED_TASK_STACK=2
run long BWAGPU_SEED_MRG=0 BWAGPU_SEED_TASKS=0 BWAGPU_PUBLISH_BLK=0 BWAGPU_SEEDSW_LDS=0 BWAGPU_DEDUP_BLK=0
exit $rc; fi
run pe BWAGPU_SEED_MRG=1
run pe BWAGPU_SEED_MRG=2
run pe BWAGPU_OCC32=0
run pe BWAGPU_OCC32=0 BWAGPU_SEED_COOP=1
run pe BWAGPU_DEDUP_WAVE=1
run long BWAGPU_SEED_MRG=2
run long BWAGPU_SEED_CHUNK=256
run long BWAGPU_PUBLISH_BLK=1
run long BWAGPU_LONG_QLDS=1
run long BWAGPU_SEEDSW_LDS=1
run long BWAGPU_DEDUP_BLK=1
run long BWAGPU_EXT_BLK=1
run long BWAGPU_EXT_BLK=1 BWAGPU_LONG_QLDS=1
run long BWAGPU_SEED_MRG=2 BWAGPU_SEED_CHUNK=256 BWAGPU_PUBLISH_BLK=1 BWAGPU_LONG_QLDS=1 BWAGPU_SEEDSW_LDS=1 BWAGPU_DEDUP_BLK=1 BWAGPU_EXT_BLK=1
exit $rc
