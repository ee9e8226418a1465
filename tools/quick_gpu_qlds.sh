#!/bin/bash
# Where does BWAGPU_LONG_QLDS=1 lose its time (profiles/r03_quick_hw_check.md: 1.38 s against 0.22 s for 40 reads of 5 kb)?  Stage times by the
# library's HIP events for the switch alone, with chunked seeding, and for all five long-read switches; then the short-read stages of one
# 667 k-read batch on one handle (200 kb genome: the HBM-bound stages are not representative, the issue-bound ones are).
Q=tests/_data/quick; P=tests/golden/g200k
for e in BWAGPU_X=0 BWAGPU_LONG_QLDS=1 "BWAGPU_LONG_QLDS=1 BWAGPU_SEED_CHUNK=256" "BWAGPU_LONG_QLDS=1 BWAGPU_SEEDSW_LDS=1" "BWAGPU_SEED_MRG=2 BWAGPU_SEED_CHUNK=256 BWAGPU_PUBLISH_BLK=1 BWAGPU_LONG_QLDS=1 BWAGPU_SEEDSW_LDS=1"; do
  env $e BWAGPU_CLI_TRACE=1 timeout 20 bwa_amd/bwa-amd mem -t 8 -x pacbio $P $Q/long.fq 2>&1 >/dev/null | grep -o "stage ms.*" | sed "s/^/long [$e] /"
done
for i in $(seq 35); do cat $Q/r1.fq $Q/r2.fq; done > /tmp/quick_big.fq
BWAGPU_CLI_STREAMS=1 BWAGPU_CLI_TRACE=1 timeout 30 bwa_amd/bwa-amd mem -t 16 -K 100000000 $P /tmp/quick_big.fq 2>&1 >/dev/null | grep -o "device_sub\].*" | head -4
