#!/usr/bin/env python3
"""Inputs of tools/quick_gpu_check.sh / quick_gpu_variants.sh (run on the CPU box, where oracle/_ref/bwa exists; tests/_data/ travels with gpurun).

20 000 pairs of 2x150 bp and 40 reads of 5 kb from the committed 200 kb genome, plus tests/_data/quick/expected.txt: sha256 of
`bwa mem`'s SAM minus @PG lines for `-t 8 -K 1500000` (pairs) and `-t 8 -x pacbio` (long reads).
"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bwa_amd import simdata  # noqa: E402
import testdata  # noqa: E402


def digest(cmd):
    p = subprocess.run(cmd, shell=True, capture_output=True, check=True, cwd=ROOT)
    body = b"".join(l for l in p.stdout.splitlines(True) if not l.startswith(b"@PG"))
    return hashlib.sha256(body).hexdigest()


def main():
    prefix, g = testdata.small_index()
    q = os.path.join(ROOT, "tests", "_data", "quick")
    os.makedirs(q, exist_ok=True)
    r1, r2 = simdata.make_reads_pe(g, 20000, seed=991)[:2]
    simdata.write_fastq(f"{q}/r1.fq", r1, suffix="/1")
    simdata.write_fastq(f"{q}/r2.fq", r2, suffix="/2")
    lr = simdata.make_reads_long(g, 40, length=5000, seed=5)
    simdata.write_fastq(f"{q}/long.fq", lr[0] if isinstance(lr, tuple) else lr)
    bwa = os.path.join(ROOT, "oracle", "_ref", "bwa")
    d = [digest(f"{bwa} mem -t 8 -K 1500000 {prefix} {q}/r1.fq {q}/r2.fq 2>/dev/null"),
         digest(f"{bwa} mem -t 8 -x pacbio {prefix} {q}/long.fq 2>/dev/null")]
    open(f"{q}/expected.txt", "w").write("\n".join(d) + "\n")
    print("\n".join(d))


if __name__ == "__main__":
    main()
