"""Index construction on the device: writes <prefix>.{bwt,sa,pac,ann,amb} byte-identical to what the
reference's `bwa index` produces (bwtindex.c:255-323) for an N-free genome, in seconds instead of ~0.3-0.7 s/Mbp.

The suffix array of T = forward + reverse-complement strand (bwtindex.c:305-311 builds the BWT of exactly this text) is
computed by prefix doubling with `torch.sort` (HBM-resident radix sorts; PyTorch is plumbing here: device memory + sort),
then the reference's on-disk layouts are assembled:
  .bwt  primary, L2[1..4], then per 128 symbols 4 x u64 running Occ counts + 8 x u32 of 2-bit BWT symbols, plus a final
        counts record (bwt_bwtupdate_core, bwtindex.c:150-172; bwt_dump_bwt, bwt.c:385-393)
  .sa   primary, L2[1..4], sa_intv, seq_len, then SA[k] for k = 32, 64, ... (bwt_cal_sa bwt.c:62-84; bwt_dump_sa :396-407)
  .pac  2 bits per base of the forward strand, tail bytes as bns_fasta2bntseq writes them (bntseq.c:314-323)
  .ann / .amb  text (bns_dump, bntseq.c:65-95)
Equality with `bwa index` output is asserted in tests/test_index_build.py.
"""
from __future__ import annotations

import numpy as np
import torch


def _shift(x: torch.Tensor, s: int) -> torch.Tensor:
    """x[i+s] with zeros past the end."""
    n = x.numel()
    if s >= n:
        return torch.zeros_like(x)
    out = torch.zeros_like(x)
    out[: n - s] = x[s:]
    return out


def suffix_array(T: torch.Tensor) -> torch.Tensor:
    """Suffix array of the 2-bit text T (uint8 tensor, values 0..3) with an implicit smallest terminator."""
    n = T.numel()
    dev = T.device
    c = T.to(torch.int64)
    k = 1
    while k < 24 and k * 2 <= 16:           # 16-mers by doubling, then 24-mers
        c = c * (1 << (2 * k)) + _shift(c, k)
        k *= 2
    if k == 16:
        c8 = T.to(torch.int64)
        for kk in (1, 2, 4):
            c8 = c8 * (1 << (2 * kk)) + _shift(c8, kk)
        c = c * (1 << 16) + _shift(c8, 16)
        k = 24
        del c8
    v = torch.clamp(n - torch.arange(n, device=dev, dtype=torch.int64), max=k)   # shorter (terminated) suffixes sort first
    key = c * (k + 1) + v
    del c, v
    h = k
    while True:
        skey, perm = torch.sort(key)
        del key
        neq = (skey[1:] != skey[:-1]).to(torch.int64)
        del skey
        rank_sorted = torch.zeros(n, dtype=torch.int64, device=dev)
        torch.cumsum(neq, 0, out=rank_sorted[1:])
        del neq
        if int(rank_sorted[-1]) == n - 1:
            return perm
        rank = torch.empty(n, dtype=torch.int64, device=dev)
        rank[perm] = rank_sorted
        del rank_sorted, perm
        key = rank * (n + 1) + _shift(rank + 1, h)     # (rank of suffix i, rank of suffix i+h); 0 = past the end = smallest
        del rank
        h *= 2


def build_index(prefix: str, codes: np.ndarray, contigs, device: str | None = None, sa_intv: int = 32) -> dict:
    """codes: uint8 array of the whole genome (values 0..3, contigs concatenated); contigs: [(name, length), ...]."""
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    l_pac = int(codes.shape[0])
    assert sum(int(l) for _, l in contigs) == l_pac and codes.max() <= 3
    fwd = torch.from_numpy(np.ascontiguousarray(codes)).to(device)
    T = torch.cat([fwd, (3 - fwd).flip(0)])
    n = 2 * l_pac
    assert (n + 1) * (n + 1) < (1 << 63), "prefix doubling packs two ranks into one int64 key"
    sa = suffix_array(T)                                          # n entries: suffix starts in sorted order
    sa_full = torch.cat([torch.tensor([n], dtype=torch.int64, device=sa.device), sa])   # row 0 = the terminator suffix
    del sa
    primary = int(torch.nonzero(sa_full == 0)[0, 0])
    prev = sa_full - 1
    prev[primary] = 0
    bw = T[prev]                                                  # BWT symbol of every row (row `primary` is the terminator)
    keep = torch.ones(n + 1, dtype=torch.bool, device=bw.device); keep[primary] = False
    B = bw[keep]                                                  # the stored, $-less BWT string (bwt.c:114,177)
    del bw, keep, prev
    cnt = torch.bincount(T.to(torch.int64), minlength=4).cpu().numpy().astype(np.uint64)
    L2 = np.zeros(5, dtype=np.uint64); L2[1:] = np.cumsum(cnt)
    # sampled SA: SA[k] for k % sa_intv == 0, k = 0..n ; SA[0] is stored as -1 by the loader, not in the file
    n_sa = (n + sa_intv) // sa_intv
    sa_s = sa_full[::sa_intv][:n_sa].cpu().numpy().astype(np.uint64)
    del sa_full
    # 2-bit packing, 16 symbols per u32, symbol i in bits (15 - i%16)*2 (bwt.h:74-80)
    n_words = (n + 15) // 16
    n_blk = (n + 127) // 128
    Bp = torch.zeros(n_blk * 128, dtype=torch.int64, device=B.device); Bp[:n] = B.to(torch.int64)
    sh = torch.tensor([30 - 2 * j for j in range(16)], dtype=torch.int64, device=B.device)
    words = (Bp.view(-1, 16) << sh).sum(1)                        # [n_blk * 8]
    blk_id = torch.arange(n, device=B.device, dtype=torch.int64) >> 7
    per_blk = torch.bincount(blk_id * 4 + B.to(torch.int64), minlength=n_blk * 4).view(n_blk, 4)
    occ = torch.cumsum(per_blk, 0) - per_blk                      # counts before each block
    total = per_blk.sum(0)
    del Bp, blk_id, per_blk, B
    occ_np = occ.cpu().numpy().astype(np.uint64)                  # [n_blk, 4] u64 -> 8 u32 each
    words_np = words.cpu().numpy().astype(np.uint32).reshape(n_blk, 8)
    inter = np.empty((n_blk, 16), dtype=np.uint32)
    inter[:, :8] = occ_np.view(np.uint32).reshape(n_blk, 8)
    inter[:, 8:] = words_np
    flat = inter.reshape(-1)
    pad_words = n_blk * 8 - n_words                               # the last block holds only the words that exist
    if pad_words:
        flat = flat[: flat.shape[0] - pad_words]
    flat = np.concatenate([flat, total.cpu().numpy().astype(np.uint64).view(np.uint32)])
    with open(prefix + ".bwt", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes()); f.write(L2[1:].tobytes()); f.write(flat.tobytes())
    with open(prefix + ".sa", "wb") as f:
        f.write(np.array([primary], dtype=np.uint64).tobytes()); f.write(L2[1:].tobytes())
        f.write(np.array([sa_intv, n], dtype=np.uint64).tobytes()); f.write(sa_s[1:].tobytes())
    # .pac: 4 bases per byte, base l in bits (3 - l%4)*2; file length is always l_pac/4 + 2 (bntseq.c:314-323)
    nb = (l_pac + 3) // 4
    cp = np.zeros(nb * 4, dtype=np.uint8); cp[:l_pac] = codes
    pac = (cp[0::4] << 6) | (cp[1::4] << 4) | (cp[2::4] << 2) | cp[3::4]
    with open(prefix + ".pac", "wb") as f:
        f.write(pac.astype(np.uint8).tobytes())
        if l_pac % 4 == 0:
            f.write(b"\0")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write(f"{l_pac} {len(contigs)} 11\n")
        off = 0
        for name, ln in contigs:
            f.write(f"0 {name} (null)\n{off} {int(ln)} 0\n")
            off += int(ln)
    with open(prefix + ".amb", "w") as f:
        f.write(f"{l_pac} {len(contigs)} 0\n")
    return {"l_pac": l_pac, "seq_len": n, "primary": primary, "n_sa": int(n_sa)}
