#!/bin/bash
# Full-size batches on hardware without a large index: the quick check's 40 000 reads, 35 times over, as ONE single-end file (1.4 M reads,
# 450 MB, made on the box with cat).  Single-end SAM does not depend on the batch size, so the run with 667 k-read batches (pooled
# page-locked buffers, arenas of the bench's size) must equal the run with 10 k-read batches, the regime quick_gpu_check.sh compares with
# `bwa mem`; five batches in flight and the block-parallel input stage must equal them too.
Q=tests/_data/quick; P=tests/golden/g200k; B=/tmp/quick_big.fq; rc=0
for i in $(seq 35); do cat $Q/r1.fq $Q/r2.fq; done > $B
body() { grep -av '^@PG' | sha256sum | cut -d' ' -f1; }
run() { local name=$1; shift; d=$(env "$@" timeout 40 bwa_amd/bwa-amd mem -t 16 $KARG $P $B 2>/tmp/quick_big.err | body); echo "$name $d $(grep -o 'reads in [0-9.]* sec' /tmp/quick_big.err | tail -1) | $(grep -o 'stage busy time.*' /tmp/quick_big.err | tail -1)"; eval "D_$name=$d"; }
KARG="-K 1500000";   run small BWAGPU_X=0
KARG="-K 100000000"; run full BWAGPU_X=0
KARG="-K 100000000"; run five BWAGPU_CLI_STREAMS=5
KARG="-K 100000000"; run par BWAGPU_CLI_PARSE_THREADS=4
for n in full five par; do v=D_$n; [ "${!v}" = "$D_small" ] && echo "$n == small OK" || { echo "$n MISMATCH"; rc=1; }; done
exit $rc
