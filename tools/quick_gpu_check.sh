#!/bin/bash
# A hardware check that fits a minute of box time (no Python, no torch import): `bwa-amd mem` on the committed 200 kb index against digests of
# `bwa mem`'s SAM made beforehand on the CPU box (tests/_data/quick/, see the here-document in DESIGN §5 "quick check"):
#   20 000 pairs of 2x150 bp in batches of 1.5 Mbp (8 batches: upload, hot path, CIGARs, mate rescue, pestat, finalize) and 40 reads of 5 kb
#   with -x pacbio (the long-read tiers).  Prints OK / MISMATCH per leg; exit code 1 on any mismatch.
Q=tests/_data/quick; P=tests/golden/g200k; rc=0
body() { grep -av '^@PG' | sha256sum | cut -d' ' -f1; }
t0=$(date +%s%N)
d1=$(timeout 40 bwa_amd/bwa-amd mem -t 8 -K 1500000 $P $Q/r1.fq $Q/r2.fq 2>$Q/pe.err | body)
d2=$(timeout 40 bwa_amd/bwa-amd mem -t 8 -x pacbio $P $Q/long.fq 2>$Q/long.err | body)
t1=$(date +%s%N)
[ "$d1" = "$(sed -n 1p $Q/expected.txt)" ] && echo "pe OK $d1" || { echo "pe MISMATCH $d1"; tail -5 $Q/pe.err; rc=1; }
[ "$d2" = "$(sed -n 2p $Q/expected.txt)" ] && echo "long OK $d2" || { echo "long MISMATCH $d2"; tail -5 $Q/long.err; rc=1; }
echo "elapsed $(( (t1 - t0) / 1000000 )) ms"
mkdir -p gpurun_out/quick; cp $Q/pe.err $Q/long.err gpurun_out/quick/ 2>/dev/null
exit $rc
