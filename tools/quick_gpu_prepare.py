#!/usr/bin/env python3
"""Inputs of tools/quick_gpu_check.sh / quick_gpu_variants.sh (run on the CPU box, where oracle/_ref/bwa exists; tests/_data/ travels with gpurun).

20 000 pairs of 2x150 bp and 40 reads of 5 kb from the committed 200 kb genome, plus tests/_data/quick/expected.txt: sha256 of
`bwa mem`'s SAM minus @PG lines for `-t 8 -K 1500000` (pairs) and `-t 8 -x pacbio` (long reads).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bwa_amd import simdata  # noqa: E402
import testdata  # noqa: E402


def digest(cmd):
    """sha256 of the command's SAM minus @PG lines, through the very pipeline the check scripts use on the box (grep adds the final newline
    that the reference's output lacks when the last FASTQ record has none)."""
    p = subprocess.run(f"({cmd}) | grep -av '^@PG' | sha256sum | cut -d' ' -f1", shell=True, capture_output=True, check=True, cwd=ROOT, text=True)
    return p.stdout.strip()


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
    # quick_gpu_bindings.sh: 300 reads for the reference's example program (mem_align1 binding) and the CLI tests' awkward file, on the 2 Mb genome
    import pathlib
    import test_cli
    import test_sam_via_reference
    fa, g2 = testdata.medium_index()
    test_sam_via_reference._example_inputs(pathlib.Path(q), g2, 300, seed=72)
    os.makedirs(f"{q}/w", exist_ok=True)
    test_cli._write_inputs(pathlib.Path(q) / "w", g2, 2000, seed=406)
    dw = digest(f"{bwa} mem -C {os.path.relpath(fa, ROOT)} {os.path.relpath(q, ROOT)}/w/weird.fq 2>/dev/null")
    open(f"{q}/expected_weird.txt", "w").write(dw + "\n")
    d.append(dw)
    print("\n".join(d))


if __name__ == "__main__":
    main()
