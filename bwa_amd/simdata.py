"""Seeded synthetic genomes and reads (configs C1-C5).

There is no genome data and no network on the build or GPU boxes, so every benchmark and parity
test runs on data generated here.  Everything is a pure function of its seed (numpy PCG64), so the
GPU box regenerates byte-identical inputs.

Genome recipe: `n_contigs` contigs of uniform-random ACGT, then (optionally) repeat families are
stamped in: `n_interspersed` copies of a few 300 bp elements, each copy mutated at `divergence`,
and `n_tandem` tandem arrays (unit 20-60 bp, 5-40 copies, lightly mutated).  The repeats make the
max_occ / frac_rep / multi-chain paths of BWA-MEM fire (bwamem.c:291-309), which a purely random
genome never does.

Read model (same as BASELINE.md section 2): uniform start, random strand, per-base 1 % substitution,
0.15 % deletion, 0.15 % insertion; pairs are FR with insert ~ N(400, 40) clipped at >= 200.
"""
from __future__ import annotations

import numpy as np

_ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


def make_genome(total_len: int, n_contigs: int = 4, seed: int = 42, n_interspersed: int | None = None,
                n_tandem: int | None = None, divergence: float = 0.10, repeats: bool = True):
    """Return (codes uint8[total_len] in 0..3, contig_lengths list)."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=total_len, dtype=np.uint8)
    # contig lengths: roughly geometric so that there are big and small contigs
    w = np.array([0.5 ** min(i, 6) for i in range(n_contigs)], dtype=np.float64)
    lens = np.maximum((w / w.sum() * total_len).astype(np.int64), 1000)
    lens[0] += total_len - lens.sum()
    assert lens.sum() == total_len and (lens > 0).all()
    if repeats:
        if n_interspersed is None:
            n_interspersed = max(8, total_len // 1400)   # ~ the 1500 copies / 2 Mb of the SURVEY probe
        if n_tandem is None:
            n_tandem = max(2, total_len // 40000)
        n_fam = 4
        fams = [rng.integers(0, 4, size=300, dtype=np.uint8) for _ in range(n_fam)]
        pos = rng.integers(0, total_len - 400, size=n_interspersed)
        fam = rng.integers(0, n_fam, size=n_interspersed)
        strand = rng.integers(0, 2, size=n_interspersed)
        for p, f, s in zip(pos.tolist(), fam.tolist(), strand.tolist()):
            e = fams[f].copy()
            m = rng.random(300) < divergence
            e[m] = (e[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            if s:
                e = (3 - e)[::-1]
            g[p:p + 300] = e
        for _ in range(n_tandem):
            unit = rng.integers(0, 4, size=int(rng.integers(20, 61)), dtype=np.uint8)
            ncopy = int(rng.integers(5, 41))
            arr = np.tile(unit, ncopy)
            m = rng.random(arr.size) < 0.02
            arr[m] = (arr[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
            p = int(rng.integers(0, total_len - arr.size - 1))
            g[p:p + arr.size] = arr
    return g, [int(x) for x in lens]


def make_genome_large(total_len: int, n_contigs: int = 24, seed: int = 42, divergence: float = 0.10, threads: int = 8):
    """GRCh38-scale stand-in: the recipe of make_genome (uniform-random contigs + 4 families of 300 bp interspersed repeats,
    one copy per 1400 bp, + one tandem array per 40 kb), restructured so that 3.1 Gbp take tens of seconds: the random
    background is drawn in 16 MB chunks, each from its own PCG64 stream spawned from `seed` (so the result does not depend
    on the number of threads), and copy mutations come from byte draws.  Contig lengths fall off linearly (the longest is
    ~8 % of the genome, like chr1); every contig stays below 2^31 (bntann1_t::len is int32).
    Not byte-compatible with make_genome for the same seed: fixtures keep using make_genome."""
    from concurrent.futures import ThreadPoolExecutor
    CH = 1 << 24
    n_ch = (total_len + CH - 1) // CH
    ss = np.random.SeedSequence(seed)
    kids = ss.spawn(n_ch + 1)
    g = np.empty(total_len, dtype=np.uint8)

    def fill(k):
        lo, hi = k * CH, min(total_len, (k + 1) * CH)
        g[lo:hi] = np.frombuffer(np.random.Generator(np.random.PCG64(kids[k])).bytes(hi - lo), dtype=np.uint8) & 3

    with ThreadPoolExecutor(max_workers=max(1, threads)) as ex:
        list(ex.map(fill, range(n_ch)))
    rng = np.random.Generator(np.random.PCG64(kids[n_ch]))
    w = np.linspace(1.0, 0.2, n_contigs)
    lens = np.maximum((w / w.sum() * total_len).astype(np.int64), 1000)
    lens[0] += total_len - lens.sum()
    assert lens.sum() == total_len and (lens > 0).all() and lens.max() < (1 << 31)
    n_fam = 4
    fams = rng.integers(0, 4, size=(n_fam, 300), dtype=np.uint8)
    n_inter = max(8, total_len // 1400)
    thr = int(round(divergence * 256))
    for lo in range(0, n_inter, 200_000):                 # chunks bound the temporaries (200k copies x 300 bases)
        c = min(200_000, n_inter - lo)
        pos = rng.integers(0, total_len - 400, size=c)
        e = fams[rng.integers(0, n_fam, size=c)]
        m = rng.integers(0, 256, size=(c, 300), dtype=np.uint8) < thr
        e = np.where(m, (e + rng.integers(1, 4, size=(c, 300), dtype=np.uint8)) & 3, e).astype(np.uint8)
        rev = rng.integers(0, 2, size=c).astype(bool)
        e[rev] = (3 - e[rev])[:, ::-1]
        for p, row in zip(pos.tolist(), e):               # slice copies (~0.6 us each) beat a 60 M-element fancy-index scatter
            g[p:p + 300] = row
    n_tandem = max(2, total_len // 40000)
    ulen = rng.integers(20, 61, size=n_tandem).tolist(); ncopy = rng.integers(5, 41, size=n_tandem).tolist()
    tpos = rng.integers(0, total_len - 60 * 41 - 1, size=n_tandem).tolist()
    units = rng.integers(0, 4, size=(n_tandem, 60), dtype=np.uint8)
    mut = rng.integers(0, 256, size=(n_tandem, 64), dtype=np.uint8)     # per array: up to 32 (offset, delta) mutation draws
    for k in range(n_tandem):
        arr = np.tile(units[k, : ulen[k]], ncopy[k])
        n_mut = arr.size // 50                                           # ~2 % of the array's bases
        for j in range(min(n_mut, 32)):
            o = (int(mut[k, 2 * j]) * 257 + j * 7919) % arr.size
            arr[o] = (arr[o] + 1 + (mut[k, 2 * j + 1] % 3)) & 3
        g[tpos[k]: tpos[k] + arr.size] = arr
    return g, [int(x) for x in lens]


def write_fasta(path: str, g: np.ndarray, lens, prefix: str = "chr"):
    off = 0
    with open(path, "wb") as f:
        for i, ln in enumerate(lens):
            f.write(f">{prefix}{i + 1}\n".encode())
            f.write(_ASCII[g[off:off + ln]].tobytes())
            f.write(b"\n")
            off += ln


def _mutate(ref_codes: np.ndarray, starts: np.ndarray, length: int, rng, sub: float, dele: float, ins: float):
    """Vectorised read synthesis: returns uint8[n, length] codes sampled from ref_codes at `starts`."""
    n = starts.shape[0]
    is_ins = rng.random((n, length)) < ins
    is_del = rng.random((n, length)) < dele
    is_ins[:, 0] = False
    is_del[:, 0] = False
    # reference offset consumed before output position j
    consumed = np.arange(length, dtype=np.int64)[None, :] - np.cumsum(is_ins, axis=1) + np.cumsum(is_del, axis=1)
    idx = starts[:, None] + consumed
    np.clip(idx, 0, ref_codes.shape[0] - 1, out=idx)
    reads = ref_codes[idx]
    rnd = rng.integers(0, 4, size=(n, length), dtype=np.uint8)
    reads = np.where(is_ins, rnd, reads)
    m = rng.random((n, length)) < sub
    reads = np.where(m, (reads + rng.integers(1, 4, size=(n, length), dtype=np.uint8)) & 3, reads).astype(np.uint8)
    return reads


def _revcomp(a: np.ndarray) -> np.ndarray:
    out = (3 - a)[:, ::-1]
    return np.ascontiguousarray(out)


def make_reads_se(g: np.ndarray, n: int, length: int = 150, seed: int = 43, sub: float = 0.01, dele: float = 0.0015,
                  ins: float = 0.0015, n_frac: float = 0.0):
    """n single-end reads as uint8[n, length] nt4 codes (0..3, 4 = N)."""
    rng = np.random.default_rng(seed)
    span = length + 40
    starts = rng.integers(0, g.shape[0] - span, size=n)
    reads = _mutate(g, starts, length, rng, sub, dele, ins)
    rev = rng.random(n) < 0.5
    reads[rev] = _revcomp(reads[rev])
    if n_frac > 0:
        m = rng.random((n, length)) < n_frac
        reads[m] = 4
    return reads


def make_reads_pe(g: np.ndarray, n_pairs: int, length: int = 150, seed: int = 44, ins_mean: float = 400.0,
                  ins_sd: float = 40.0, sub: float = 0.01, dele: float = 0.0015, ins: float = 0.0015):
    """n_pairs FR pairs -> (r1, r2) each uint8[n_pairs, length]."""
    rng = np.random.default_rng(seed)
    isz = np.maximum(rng.normal(ins_mean, ins_sd, size=n_pairs).astype(np.int64), max(200, length + 10))
    starts = rng.integers(0, g.shape[0] - isz.max() - 80, size=n_pairs)
    r1 = _mutate(g, starts, length, rng, sub, dele, ins)
    # read 2: reverse complement of the fragment's far end
    far = starts + isz - length
    r2f = _mutate(g, far, length, rng, sub, dele, ins)
    r2 = _revcomp(r2f)
    flip = rng.random(n_pairs) < 0.5  # fragment from the reverse strand: swap roles
    a = np.where(flip[:, None], r2, r1)
    b = np.where(flip[:, None], r1, r2)
    return np.ascontiguousarray(a), np.ascontiguousarray(b)


def make_reads_long(g: np.ndarray, n: int, length: int = 10000, seed: int = 7, sub: float = 0.015, dele: float = 0.04,
                    ins: float = 0.09):
    """PacBio-CLR-like long reads (config C5)."""
    rng = np.random.default_rng(seed)
    starts = rng.integers(0, g.shape[0] - int(length * 1.2) - 10, size=n)
    reads = _mutate(g, starts, length, rng, sub, dele, ins)
    rev = rng.random(n) < 0.5
    reads[rev] = _revcomp(reads[rev])
    return reads


def write_fastq(path: str, reads: np.ndarray, name_prefix: str = "r", suffix: str = "", start: int = 0):
    """FASTQ with names <prefix><i><suffix>, constant quality 'I'.  Records are assembled as byte matrices, one per group of
    equal name length (a Python loop per read costs ~2 us x millions of reads)."""
    n, length = reads.shape
    pre, suf = name_prefix.encode(), suffix.encode()
    with open(path, "wb") as f:
        lo = 0
        while lo < n:
            nd = len(str(start + lo))
            hi = min(n, 10 ** nd - start)                     # reads lo..hi-1 have nd-digit numbers
            m = hi - lo
            w = 1 + len(pre) + nd + len(suf) + 1 + length + 3 + length + 1
            rec = np.empty((m, w), dtype=np.uint8)
            k = 0
            rec[:, k] = ord("@"); k += 1
            rec[:, k:k + len(pre)] = np.frombuffer(pre, dtype=np.uint8); k += len(pre)
            idx = np.arange(start + lo, start + hi, dtype=np.int64)
            for d in range(nd):
                rec[:, k + d] = (idx // 10 ** (nd - 1 - d)) % 10 + 48
            k += nd
            if suf:
                rec[:, k:k + len(suf)] = np.frombuffer(suf, dtype=np.uint8); k += len(suf)
            rec[:, k] = 10; k += 1
            rec[:, k:k + length] = _ASCII[reads[lo:hi]]; k += length
            rec[:, k:k + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8); k += 3
            rec[:, k:k + length] = ord("I"); k += length
            rec[:, k] = 10
            rec.tofile(f)
            lo = hi
