#!/bin/bash
# The mem_align1 binding (oracle/_ref/example_gpu: the reference's example.c on bwagpu_align_bseq, a batch of one per read) and the
# block-parallel input stage on an awkward file, on hardware, without Python (inputs: the here-document in profiles/r03_quick_hw_check.md).
Q=tests/_data/quick; FA=tests/_data/g2m.fa
timeout 6 oracle/_ref/example_gpu $FA $Q/lite.fq > /tmp/ex_gpu.out 2>/tmp/ex_gpu.err; echo "example_gpu rc $? $(wc -l < /tmp/ex_gpu.out) lines"
oracle/_ref/example $FA $Q/lite.fq > /tmp/ex.out 2>/dev/null
cmp -s /tmp/ex.out /tmp/ex_gpu.out && echo "mem_align1 binding OK" || { echo "mem_align1 binding MISMATCH"; tail -3 /tmp/ex_gpu.err; }
d=$(BWAGPU_CLI_PARSE_THREADS=4 BWAGPU_CLI_PAR_BLOCK=65536 timeout 5 bwa_amd/bwa-amd mem -C $FA $Q/w/weird.fq 2>/dev/null | grep -av '^@PG' | sha256sum | cut -d' ' -f1)
[ "$d" = "$(cat $Q/expected_weird.txt)" ] && echo "weird.fq OK" || echo "weird.fq MISMATCH $d"
