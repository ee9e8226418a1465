"""Shared test inputs: seeded genomes + their reference-format indices.

small_index(): 200 kb genome whose index files are committed under tests/golden/ (built once by the reference's
`bwa index`, see tests/golden/make_golden.py) -- usable everywhere, including boxes without oracle/_ref.
medium_index(): 2 Mb repeat-rich genome indexed on demand with oracle/_ref/bwa into tests/_data/ (needs oracle/_ref).
"""
import os
import numpy as np

from bwa_amd import simdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "_data")

SMALL = dict(total_len=200_000, n_contigs=3, seed=7)
MEDIUM = dict(total_len=2_000_000, n_contigs=3, seed=43)


def small_genome():
    return simdata.make_genome(**SMALL)


def small_index():
    g, _ = small_genome()
    return os.path.join(GOLDEN, "g200k"), g


def medium_index():
    import refapi
    os.makedirs(DATA, exist_ok=True)
    g, lens = simdata.make_genome(**MEDIUM)
    fa = os.path.join(DATA, "g2m.fa")
    if not os.path.exists(fa):
        simdata.write_fasta(fa, g, lens)
    refapi.build_index(fa)
    return fa, g


def flat(reads: np.ndarray):
    n, length = reads.shape
    return np.ascontiguousarray(reads.reshape(-1)), np.arange(0, n + 1, dtype=np.int64) * length


def ragged(read_list):
    off = np.zeros(len(read_list) + 1, dtype=np.int64)
    for i, r in enumerate(read_list):
        off[i + 1] = off[i] + len(r)
    seqs = np.concatenate([np.asarray(r, dtype=np.uint8) for r in read_list]) if read_list else np.zeros(0, dtype=np.uint8)
    return seqs, off
