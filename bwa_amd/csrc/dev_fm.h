// dev_fm.h -- FM-index primitives on the device (rank, bidirectional extension, SA lookup).
#pragma once
#include "dev_common.h"

struct OccBlock { uint4 c01, c23, w0, w1; };

DEVFN OccBlock load_block(const DevIndex &ix, u64 blk)
{
	const uint4 *p = ix.bwt + blk * 4;
	OccBlock b;
	b.c01 = p[0]; b.c23 = p[1]; b.w0 = p[2]; b.w1 = p[3];
	return b;
}

// symbols 1,2,3 among the top n (<=32) bases of a 32-base pair; base i of a u32 word sits in bits (15-i)*2
// (bwt.h:74-80), so two consecutive words form a big-endian run of 32 bases.
DEVFN void count_pair(u32 whi, u32 wlo, int n, u32 &c1, u32 &c2, u32 &c3)
{
	if (n <= 0) return;
	u64 p = (u64)whi << 32 | wlo;
	u64 mask = 0x5555555555555555ull;
	if (n < 32) mask &= ~0ull << (64 - 2 * n);
	u64 lo = p & mask, hi = (p >> 1) & mask;
	c3 += __popcll(hi & lo);
	c2 += __popcll(hi & ~lo);
	c1 += __popcll(lo & ~hi);
}

// Occ counts of a block up to in-block offset `o` (0..127) inclusive == bwt_occ4 (bwt.c:169-186) for an
// already primary-adjusted k.
DEVFN void block_occ4(const OccBlock &b, int o, u64 cnt[4])
{
	int n = o + 1;
	u32 c1 = 0, c2 = 0, c3 = 0;
	count_pair(b.w0.x, b.w0.y, n, c1, c2, c3);
	count_pair(b.w0.z, b.w0.w, n - 32, c1, c2, c3);
	count_pair(b.w1.x, b.w1.y, n - 64, c1, c2, c3);
	count_pair(b.w1.z, b.w1.w, n - 96, c1, c2, c3);
	cnt[0] = ((u64)b.c01.y << 32 | b.c01.x) + (u32)(n - c1 - c2 - c3);
	cnt[1] = ((u64)b.c01.w << 32 | b.c01.z) + c1;
	cnt[2] = ((u64)b.c23.y << 32 | b.c23.x) + c2;
	cnt[3] = ((u64)b.c23.w << 32 | b.c23.z) + c3;
}

// symbols 1,2,3 among the top n bases of a 32-base pair, branch-free (n may be <= 0 or >= 32)
DEVFN void count_pair_bf(u32 whi, u32 wlo, int n, u32 &c1, u32 &c2, u32 &c3)
{
	int ne = n < 0 ? 0 : (n > 32 ? 32 : n);
	u64 p = (u64)whi << 32 | wlo;
	u64 keep = ne >= 32 ? ~0ull : ~(~0ull >> (2 * ne));        // top 2*ne bits
	u64 mask = 0x5555555555555555ull & keep;
	u64 lo = p & mask, hi = (p >> 1) & mask;
	c3 += __popcll(hi & lo);
	c2 += __popcll(hi & ~lo);
	c1 += __popcll(lo & ~hi);
}
DEVFN void block_occ4_bf(const OccBlock &b, int o, u64 cnt[4])
{
	int n = o + 1;
	u32 c1 = 0, c2 = 0, c3 = 0;
	count_pair_bf(b.w0.x, b.w0.y, n, c1, c2, c3);
	count_pair_bf(b.w0.z, b.w0.w, n - 32, c1, c2, c3);
	count_pair_bf(b.w1.x, b.w1.y, n - 64, c1, c2, c3);
	count_pair_bf(b.w1.z, b.w1.w, n - 96, c1, c2, c3);
	cnt[0] = ((u64)b.c01.y << 32 | b.c01.x) + (u32)(n - c1 - c2 - c3);
	cnt[1] = ((u64)b.c01.w << 32 | b.c01.z) + c1;
	cnt[2] = ((u64)b.c23.y << 32 | b.c23.x) + c2;
	cnt[3] = ((u64)b.c23.w << 32 | b.c23.z) + c3;
}

// Single-child bwt_extend for the seeding kernel: only ok[c] is produced, and the whole routine is one straight-line
// instruction stream (both Occ blocks are always fetched -- the same block twice when k and l share it, an L1 hit --
// and every data-dependent choice is a select), so that all lanes of a wave run it together whatever their state.
// Returns the number of distinct 64-byte blocks touched (N_blk of SURVEY.md 8d).
// Occ counts up to position kk (primary-adjusted, inclusive) from the 32-byte layout: == block_occ4_bf on the 64-byte one.
DEVFN void occ32_counts(const uint4 &rel, const uint4 &w, const uint4 &sb01, const uint4 &sb23, int o, u64 cnt[4])
{
	const int n = o + 1;
	u32 c1 = 0, c2 = 0, c3 = 0;
	count_pair_bf(w.x, w.y, n, c1, c2, c3);
	count_pair_bf(w.z, w.w, n - 32, c1, c2, c3);
	cnt[0] = ((u64)sb01.y << 32 | sb01.x) + rel.x + (u32)(n - c1 - c2 - c3);
	cnt[1] = ((u64)sb01.w << 32 | sb01.z) + rel.y + c1;
	cnt[2] = ((u64)sb23.y << 32 | sb23.x) + rel.z + c2;
	cnt[3] = ((u64)sb23.w << 32 | sb23.z) + rel.w + c3;
}

// The two rank queries of one extension on the 32-byte layout, in three pieces -- positions, loads, arithmetic -- so that a caller
// can put other loads of its own between "issue" and "finish" and pay ONE memory round trip for all of them (ext_one_trip,
// dev_seed.h).  Left to itself the compiler (a) loads a block's first count word only in the lanes whose symbol is A, in a block of
// its own AFTER the other 28 bytes have arrived -- a second dependent round trip per extension -- and (b) waits for whatever a
// conditional block loaded before leaving that block.  Here the loads are BUFFER loads (V#: base, size; 32-bit byte offset per
// lane): a lane that does not need one passes an offset beyond the buffer's size -- the hardware's range check then returns zeros
// without a memory request -- so there is no branch around any load and the whole set issues back to back.  dev_keep() is an
// empty asm statement that reads and writes the registers: every load issued before it has to have landed there, none can be
// narrowed or sunk past it, and it costs no instruction.
// The routine also asks less of the superblock table: an extension by symbol c needs, per position, the superblock's count of c and
// the summed counts of the symbols above c (they only ever enter a difference) -- DevIndex::occ_sbx holds exactly that pair per
// (superblock, symbol), so a position costs one 16-byte load where the row form costs two, and eight registers fewer stay live.
#define BUF_OOB 0xFFFFFF00u        // an offset no buffer reaches (sizes are capped at BUF_MAX_BYTES; +32 of instruction offset cannot wrap)
#define BUF_MAX_BYTES 0xFFFFFE00ull
DEVFN BufRsrc buf_rsrc(const void *p, u64 bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(u32)(bytes > BUF_MAX_BYTES ? 0 : bytes), 0x00020000); }   // (raw buffer, 32-bit elements; a table too large for a V# reads as empty)
DEVFN uint2 buf_load8(BufRsrc r, u32 off) { const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0); return make_uint2(v[0], v[1]); }
DEVFN uint4 buf_load16(BufRsrc r, u32 off) { const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); return make_uint4(v[0], v[1], v[2], v[3]); }
struct Occ32Pos { u64 kk, ll; };
struct Occ32Data { uint4 rk, wk, rl, wl, sk, sl; };
DEVFN void dev_keep(uint4 &v) { DEV_KEEP(v.x); DEV_KEEP(v.y); DEV_KEEP(v.z); DEV_KEEP(v.w); }
DEVFN void occ32_keep(Occ32Data &d) { dev_keep(d.rk); dev_keep(d.wk); dev_keep(d.rl); dev_keep(d.wl); dev_keep(d.sk); dev_keep(d.sl); }
DEVFN Occ32Pos occ32_pos(const DevIndex &ix, const BiIntv &ik, int is_back)
{
	const u64 a = is_back ? ik.x0 : ik.x1;
	const u64 k = a - 1, l = a - 1 + ik.x2;                    // a >= 1 always (intervals start at L2[c]+1)
	Occ32Pos p; p.kk = k - (k >= ix.primary); p.ll = l - (l >= ix.primary);
	return p;
}
struct Occ32Bufs { BufRsrc occ, sbx; };
DEVFN Occ32Bufs occ32_bufs(const DevIndex &ix)
{
	Occ32Bufs b; b.occ = buf_rsrc(ix.occ32, ix.occ32_bytes); b.sbx = buf_rsrc(ix.occ_sbx, ix.occ_sbx_bytes);
	return b;
}
// `need` false: the lane issues the same six instructions and moves no data.
// (Round 4 also range-checked away the l side of a pair whose two positions share a block -- most steps of a unique match -- and copied it from the k
// side after the wait.  The fabric never saw those duplicates: FETCH_SIZE per launch 91.4 -> 93.6 GB raw, i.e. unchanged, the cache in front of it
// merges the two misses; the kernel's time did not move either, 57.9 vs 56.8-58.9 ms.  Deleted.)
DEVFN void occ32_issue(const DevIndex &ix, const Occ32Bufs &bf, bool need, const Occ32Pos &p, int c, Occ32Data &d)
{
	const u32 ok = need ? (u32)(p.kk >> 6) << 5 : BUF_OOB, ol = need ? (u32)(p.ll >> 6) << 5 : BUF_OOB;
	const u32 sk = need ? ((u32)(p.kk >> ix.occ_sb_shift) * 4 + (u32)c) << 4 : BUF_OOB, sl = need ? ((u32)(p.ll >> ix.occ_sb_shift) * 4 + (u32)c) << 4 : BUF_OOB;
	d.rk = buf_load16(bf.occ, ok); d.wk = buf_load16(bf.occ, ok + 16); d.rl = buf_load16(bf.occ, ol); d.wl = buf_load16(bf.occ, ol + 16);
	d.sk = buf_load16(bf.sbx, sk); d.sl = buf_load16(bf.sbx, sl);
}
// per-symbol counts of one position relative to its superblock: block-relative counts + the symbols of the block up to offset o
DEVFN void occ32_rel(const uint4 &rel, const uint4 &w, int o, u64 v[4])
{
	const int n = o + 1;
	u32 c1 = 0, c2 = 0, c3 = 0;
	count_pair_bf(w.x, w.y, n, c1, c2, c3);
	count_pair_bf(w.z, w.w, n - 32, c1, c2, c3);
	v[0] = (u64)rel.x + (u32)(n - c1 - c2 - c3); v[1] = (u64)rel.y + c1; v[2] = (u64)rel.z + c2; v[3] = (u64)rel.w + c3;
}
// bwt_extend for one child (bwt.c:262-275) from the loaded data -- the arithmetic of fm_extend1 below with every count split into
// its superblock part and its part relative to the superblock (sums and differences modulo 2^64, so the results are the same
// numbers); returns N_blk in the reference's 128-base units
DEVFN int occ32_finish(const DevIndex &ix, const BiIntv &ik, int c, int is_back, const Occ32Pos &p, const Occ32Data &d, BiIntv &out)
{
	const u64 a = is_back ? ik.x0 : ik.x1, other = is_back ? ik.x1 : ik.x0;
	u64 vk[4], vl[4];
	occ32_rel(d.rk, d.wk, (int)(p.kk & 63), vk);
	occ32_rel(d.rl, d.wl, (int)(p.ll & 63), vl);
	const u64 sbk = (u64)d.sk.y << 32 | d.sk.x, abk = (u64)d.sk.w << 32 | d.sk.z, sbl = (u64)d.sl.y << 32 | d.sl.x, abl = (u64)d.sl.w << 32 | d.sl.z;
	const u64 d1 = vl[1] - vk[1], d2 = vl[2] - vk[2], d3 = vl[3] - vk[3];
	u64 o = other + (a <= ix.primary && a + ik.x2 - 1 >= ix.primary) + (abl - abk);
	o += c < 3 ? d3 : 0; o += c < 2 ? d2 : 0; o += c < 1 ? d1 : 0;
	const u64 tkc = sbk + (c == 0 ? vk[0] : c == 1 ? vk[1] : c == 2 ? vk[2] : vk[3]);
	const u64 tlc = sbl + (c == 0 ? vl[0] : c == 1 ? vl[1] : c == 2 ? vl[2] : vl[3]);
	const u64 L2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	const u64 na = L2c + 1 + tkc;
	out.x2 = tlc - tkc;
	out.x0 = is_back ? na : o;
	out.x1 = is_back ? o : na;
	return (p.kk >> 7) == (p.ll >> 7) ? 1 : 2;
}

// O32: 1 = the 32-byte layout, 0 = the reference-format blocks, -1 = whichever the handle has (decided at run time)
template <int O32 = -1> DEVFN int fm_extend1(const DevIndex &ix, const BiIntv &ik, int c, int is_back, BiIntv &out)
{
	const u64 a = is_back ? ik.x0 : ik.x1, other = is_back ? ik.x1 : ik.x0;
	const u64 k = a - 1, l = a - 1 + ik.x2;                    // a >= 1 always (intervals start at L2[c]+1)
	const u64 kk = k - (k >= ix.primary), ll = l - (l >= ix.primary);
	u64 tk[4], tl[4];
	int nblk;
	if (O32 < 0 ? ix.occ32 != nullptr : O32 > 0) {             // (wave-uniform)
		const uint4 *bk = ix.occ32 + (kk >> 6) * 2, *bl = ix.occ32 + (ll >> 6) * 2;
		const uint4 *sk = (const uint4*)(ix.occ_sb + (kk >> ix.occ_sb_shift) * 4), *sl = (const uint4*)(ix.occ_sb + (ll >> ix.occ_sb_shift) * 4);
		const uint4 rk = bk[0], wk = bk[1], rl = bl[0], wl = bl[1];
		const uint4 sk0 = sk[0], sk1 = sk[1], sl0 = sl[0], sl1 = sl[1];
		occ32_counts(rk, wk, sk0, sk1, (int)(kk & 63), tk);
		occ32_counts(rl, wl, sl0, sl1, (int)(ll & 63), tl);
		nblk = (kk >> 7) == (ll >> 7) ? 1 : 2;                     // (counted in the reference's 128-base blocks, SURVEY 8d's N_blk, whatever the layout read)
	} else {
		const OccBlock bk = load_block(ix, kk >> 7), bl = load_block(ix, ll >> 7);
		block_occ4_bf(bk, (int)(kk & 127), tk);
		block_occ4_bf(bl, (int)(ll & 127), tl);
		nblk = (kk >> 7) == (ll >> 7) ? 1 : 2;
	}
	const u64 d1 = tl[1] - tk[1], d2 = tl[2] - tk[2], d3 = tl[3] - tk[3];
	u64 o = other + (a <= ix.primary && a + ik.x2 - 1 >= ix.primary);
	o += c < 3 ? d3 : 0; o += c < 2 ? d2 : 0; o += c < 1 ? d1 : 0;
	const u64 tkc = c == 0 ? tk[0] : c == 1 ? tk[1] : c == 2 ? tk[2] : tk[3];
	const u64 tlc = c == 0 ? tl[0] : c == 1 ? tl[1] : c == 2 ? tl[2] : tl[3];
	const u64 L2c = c == 0 ? ix.L2[0] : c == 1 ? ix.L2[1] : c == 2 ? ix.L2[2] : ix.L2[3];
	const u64 na = L2c + 1 + tkc;
	out.x2 = tlc - tkc;
	out.x0 = is_back ? na : o;
	out.x1 = is_back ? o : na;
	return nblk;
}

DEVFN void fm_init(const DevIndex &ix, int c, BiIntv &ik)
{	// bwt_set_intv (bwt.h:82)
	ik.x0 = ix.L2[c] + 1; ik.x2 = ix.L2[c + 1] - ix.L2[c]; ik.x1 = ix.L2[3 - c] + 1; ik.info = 0;
}

// One bwt_invPsi step (bwt.c:53-59) for k != primary: the row of the suffix one base to the left.  Reads one index block of whichever
// layout the handle has (wave-uniform choice).
DEVFN u64 fm_lf(const DevIndex &ix, u64 k)
{
	const u64 x = k - (k > ix.primary);        // position in the $-less BWT string (note: '>' here); k and x never straddle a block edge
	u64 cnt[4]; int c;                           // because occ(k) adjusts k the same way when k > primary
	if (ix.occ32 != nullptr) {
		const uint4 *b = ix.occ32 + (x >> 6) * 2;
		const uint4 *sb = (const uint4*)(ix.occ_sb + (x >> ix.occ_sb_shift) * 4);
		const uint4 rel = b[0], w = b[1], s0 = sb[0], s1 = sb[1];
		const int o = (int)(x & 63);
		const u32 word = o < 32 ? (o < 16 ? w.x : w.y) : (o < 48 ? w.z : w.w);
		c = (word >> ((~o & 15) << 1)) & 3;
		occ32_counts(rel, w, s0, s1, o, cnt);
	} else {
		const OccBlock b = load_block(ix, x >> 7);
		const int o = (int)(x & 127);
		const u32 word = o < 64 ? (o < 32 ? (o < 16 ? b.w0.x : b.w0.y) : (o < 48 ? b.w0.z : b.w0.w))
								: (o < 96 ? (o < 80 ? b.w1.x : b.w1.y) : (o < 112 ? b.w1.z : b.w1.w));
		c = (word >> ((~o & 15) << 1)) & 3;
		block_occ4_bf(b, o, cnt);              // == bwt_occ(k, c): count of c in BWT[0..x]
	}
	return c == 0 ? ix.L2[0] + cnt[0] : c == 1 ? ix.L2[1] + cnt[1] : c == 2 ? ix.L2[2] + cnt[2] : ix.L2[3] + cnt[3];
}

// bwt_sa (bwt.c:86-96) with bwt_invPsi (bwt.c:53-59): walk LF until a sampled row.  *steps += walk length.
DEVFN u64 fm_sa(const DevIndex &ix, u64 k, u32 *steps)
{
	u64 sa = 0;
	while (k & ix.sa_mask) {
		++sa;
		k = k == ix.primary ? 0 : fm_lf(ix, k);
	}
	*steps += (u32)sa;
	return sa + ix.sa[k >> ix.sa_shift];
}

// ---- contig lookup ---------------------------------------------------------------------------------------
DEVFN int dev_pos2rid(const DevIndex &ix, i64 pos_f)
{	// bns_pos2rid (bntseq.c:354-368)
	if (pos_f >= ix.l_pac) return -1;
	int lo = 0, hi = ix.n_seqs;
	while (hi - lo > 1) {
		int mid = (lo + hi) >> 1;
		if (ix.ctg_off[mid] <= pos_f) lo = mid; else hi = mid;
	}
	return lo;
}
DEVFN i64 dev_depos(const DevIndex &ix, i64 pos, int *is_rev)
{	// bns_depos (bntseq.h:87-90)
	*is_rev = pos >= ix.l_pac;
	return *is_rev ? (ix.l_pac << 1) - 1 - pos : pos;
}
DEVFN int dev_intv2rid(const DevIndex &ix, i64 rb, i64 re)
{	// bns_intv2rid (bntseq.c:370-379)
	int r;
	if (rb < ix.l_pac && re > ix.l_pac) return -2;
	int a = dev_pos2rid(ix, dev_depos(ix, rb, &r));
	int b = rb < re ? dev_pos2rid(ix, dev_depos(ix, re - 1, &r)) : a;
	return a == b ? a : -1;
}

