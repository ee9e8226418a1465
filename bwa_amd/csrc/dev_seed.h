// dev_seed.h -- SMEM seeding (mem_collect_intv, bwamem.c:140-188) and the SA-lookup kernel (bwt_sa).
//
// One lane per read.  The search is a chain of dependent 64-byte index reads (~1000 per 150 bp read), so
// throughput comes from having several hundred thousand independent chains in flight, not from
// parallelising one chain; each lane keeps its bi-interval in registers and fetches a whole Occ block
// (4 x dwordx4) per rank query.  Measured: the kernel runs at the chip's ceiling for random memory
// requests, so everything else a lane touches is kept out of HBM -- its read (2 bits per base) and the top of its interval
// stack live in LDS, matches shorter than the prefix tables' depth are a bit mask in a register (SeedLane::smask), and
// whatever is left of the stack spills to a per-lane scratch area.
#pragma once
#include "dev_fm.h"
#include "dev_sort.h"

#define PTAB_MAX 12      // deepest prefix table

struct SeedEmit {        // the MEM list of the read being seeded (lane-private scratch)
	Intv3 *intv; int r, n, cap; bool overflow;      // the list is intv[r * cap ..] (the address is rebuilt on use: two registers less per lane)
	int min_seed_len;
	i32 *shared_n;  // k_seed<LR = 1>: several lanes (tasks) append to read r's list at once -- the entry's index comes from an atomic add on the read's count (null elsewhere)
	DEVFN Intv3 *mem() const { return intv + (size_t)r * (size_t)cap; }
	// pass 2 (bwamem.c:160-168) walks pass 1's SMEMs looking for the long and rare ones; re-reading the list costs a memory round
	// trip per entry with the whole wave waiting, so the test is made here and remembered: bit k = entry base + k qualifies
	int split_len, base; u64 split_width, cand;
	DEVFN void add(u64 x0, u64 x2, int start, int end) {
		if (end - start < min_seed_len) return;
		Intv3 v; v.x0 = x0; v.x2 = x2; v.info = (u64)start << 32 | (u32)end;
		if (shared_n) {
			const int k = atomicAdd(&shared_n[r], 1);
			if (k >= cap) overflow = true; else mem()[k] = v;      // (the count keeps running past the capacity: the kernel that follows sees that and flags the batch)
			return;
		}
		if (n == cap) { overflow = true; return; }
		if (end - start >= split_len && x2 <= split_width && n - base < 64) cand |= 1ull << (n - base);
		mem()[n++] = v;
	}
};

struct IntvInfoLess { DEVFN bool operator()(const Intv3 &a, const Intv3 &b) const { return a.info < b.info; } };

#define BT_NODE_INTS 40   // n, internal, 9 chain indices, 10 children, (pad), 9 x i64 positions = 160 bytes

// ---- seeding as a per-lane state machine ---------------------------------------------------------------------------
// mem_collect_intv (bwamem.c:140-188) is three passes of nested, data-dependent loops around one expensive primitive,
// the FM extension (two 64-byte block reads + ~300 integer ops).  Written as nested loops, the lanes of a wave spread
// over five different extension sites and loop depths and the wave executes them one after the other (measured: ~12 %
// lane utilisation).  Here every lane runs the same loop: a cheap switch advances its private state
// (pass / forward sweep / backward row / re-seeding / LAST-like pass) up to the point where it needs an extension, all
// lanes then perform their extension together, and a second switch consumes the result.  Lanes pull reads from a
// global counter, so a lane that finishes a read early starts the next one instead of idling.
enum { SS_FETCH = 0, SS_PASS1, SS_PASS2, SS_PASS3, SS_FWD, SS_BWD, SS_STRAT, SS_FINAL, SS_DONE };

struct SeedLane {
	int st, len, x, k2, old_n, pass;      // (the read's number is em.r)
	u64 qoff, win;            // the read's offset in the packed base array; the 16-base window last fetched from it
	u32 win_w;                // index of that window (~0u: none)
	u64 win2; u32 win2_w;     // k_seed<RD = false, MRG = 2>: a second window -- the one the sweep is about to walk into, fetched a step ahead with the index blocks
	const u32 *rd;            // k_seed<true>: the block's LDS copy of its lanes' reads (2 bits per base), [word][lane]
	int rd_on;                // ... in use for the current read (it holds no N)
	const u8 *raw; const i64 *off;   // ... and where a read with an N finds its bases (Batch::seq, Batch::off)
	// current SMEM search (bwt_smem1a, bwt.c:289-351)
	u64 min_intv, last_x2;
	int sx, i, n0, nprev, nc, j, c, ret, last_start;
	int lo;                   // the backward sweep runs rows i >= lo (-1: all of them, bwt.c:326; a pass-1 task of a long read stops where the next task to the left reports)
	bool any;
	BiIntv ik;
	u32 code;                 // prefix-table window: forward sweep, q[sx..sx+ptab_m); backward sweep, q[i..i+ptab_m) (window_code)
	// Matches shorter than the prefix tables' depth are not kept in the stack at all: whatever is done with such an entry -- extend it
	// by a base, compare interval sizes -- is answered by the table for the string it stands for, so its end position is all there is to
	// know about it.  They are the shortest entries (the deep end of the stack): bit k of smask = there is an entry whose match is k+1
	// bases long.  An entry that grows to the tables' depth in a backward row is written to the stack then.  (Only with min_seed_len
	// above that depth: no match that short is ever reported.)
	u32 smask, srem, snew;    // the row's short entries; those not yet visited; those surviving into the next row
	int ncl;                  // survivors of the row written to the stack (nc counts the short ones too)
	int top;                  // index of the longest match in the interval stack (prev[j] = the entry j below the top)
	int slot;                 // forward sweep: ring position of the next push; backward sweep: ring position of the top entry
	SeedEmit em;
};

// The interval stack of one lane (bwt_smem1a's curr/prev vectors, bwt.c:292-300).  One array suffices: the backward sweep
// reads prev[j] for increasing j and appends at most one survivor per entry read, so survivors are written in place over the
// consumed part.  The Batch::seed_lds_ent entries nearest the top (the longest matches: the ones every backward row touches,
// and after a dozen rows the only ones left) live in LDS, packed to 16 bytes ({x0,x1,x2} < 2^37, end < 2^16) and laid out
// [slot][lane] so that a wave's accesses are conflict-free.  The forward sweep fills the LDS slots as a ring and evicts the
// oldest (shortest) entry to HBM scratch when it wraps; the backward sweep addresses entries by their depth below the top.
// Keeping the stack out of HBM matters because the kernel runs at the chip's random-request ceiling (profiles/r01_randbw_*).
struct SeedStack {
	uint4 *lds_base;  // the block's LDS array [slot][lane] (stride blockDim.x entries)
	BiIntv *glob_base; int glob_cap;   // spill areas of all lanes, glob_cap entries each; a lane's is indexed by entry and holds packed uint4 records when n_lds > 0
	// (per-lane addresses are rebuilt on use instead of being carried in registers)
	DEVFN uint4 *lds_col() const { return lds_base + threadIdx.x; }
	DEVFN BiIntv *glob_col() const { return glob_base + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * (size_t)glob_cap + PTAB_MAX; }   // (a backward row may add one entry at the deep end, see SeedLane::smask)
	int stride, n_lds;   // n_lds == 0: intervals not packable for this batch, everything in glob (unpacked)
	int virt_m;          // matches of fewer bases live in SeedLane::smask instead (0: none do)
	// forward sweep: an entry of the change-point list
	DEVFN void push_fw(SeedLane &L, const BiIntv &v) const {
		const int len = (int)v.info - L.sx;
		if (len < virt_m) L.smask |= 1u << (len - 1); else push(L, v);
	}
	static DEVFN uint4 pack(const BiIntv &v) {
		uint4 w;
		w.x = (u32)v.x0; w.y = (u32)v.x1; w.z = (u32)v.x2;
		w.w = (u32)(v.x0 >> 32) | (u32)(v.x1 >> 32) << 5 | (u32)(v.x2 >> 32) << 10 | (u32)v.info << 16;
		return w;
	}
	static DEVFN BiIntv unpack(const uint4 &w) {
		BiIntv v;
		v.x0 = (u64)(w.w & 31) << 32 | w.x; v.x1 = (u64)(w.w >> 5 & 31) << 32 | w.y; v.x2 = (u64)(w.w >> 10 & 31) << 32 | w.z;
		v.info = w.w >> 16;
		return v;
	}
	// forward sweep: append entry number L.n0; L.slot is the ring position it goes to
	DEVFN void push(SeedLane &L, const BiIntv &v) const {
		if (n_lds == 0) glob_col()[L.n0] = v;
		else {
			if (L.n0 >= n_lds) ((uint4*)glob_col())[L.n0 - n_lds] = lds_col()[L.slot * stride];   // the ring is full: evict the oldest entry
			lds_col()[L.slot * stride] = pack(v);
			L.slot = L.slot + 1 == n_lds ? 0 : L.slot + 1;
		}
		++L.n0;
	}
	// backward sweep: the entry `d` below the top (L.top = index of the top entry, L.slot = its ring position)
	DEVFN int ring(const SeedLane &L, int d) const { int s = L.slot - d; return s < 0 ? s + n_lds : s; }
	DEVFN void store(const SeedLane &L, int d, const BiIntv &v) const {
		if (d < n_lds) lds_col()[ring(L, d) * stride] = pack(v);
		else if (n_lds) ((uint4*)glob_col())[L.top - d] = pack(v);
		else glob_col()[L.top - d] = v;
	}
	DEVFN BiIntv load(const SeedLane &L, int d) const {
		if (d < n_lds) return unpack(lds_col()[ring(L, d) * stride]);
		if (n_lds) return unpack(((const uint4*)glob_col())[L.top - d]);
		return glob_col()[L.top - d];
	}
};

// Base i of the lane's read (0..3, 4 = N).  A lane walks its read sequentially, forward from x and backward from x-1, and asks
// for a base at every step; byte loads from the raw read (a different 64-byte line per lane, ~2 MB per XCD of resident
// lanes against a 4 MB L2 that the index blocks stream through) showed up as memory requests of their own.  k_pack_reads
// packs the batch to 4 bits per base and each lane keeps the 16-base word it is in; a sweep reloads it every 16 steps.
// In k_seed that reload stalled the wave almost every iteration (with 64 lanes some lane always crosses a word, and the
// index blocks cannot be requested before the base is known to the compiler's satisfaction): there the read lives in LDS at 2
// bits per base, copied once when the lane takes it, and only reads holding an N use the 4-bit array.
DEVFN int seed_q(SeedLane &L, const u64 *nib, int i)
{
	if (L.rd_on) {
		const u32 w = (u32)i >> 4;
		if (w != L.win_w) { L.win = L.rd[w * blockDim.x + threadIdx.x]; L.win_w = w; }
		return (int)((u32)L.win >> (((u32)i & 15) << 1)) & 3;
	}
	if (nib == nullptr) { const u8 c = L.raw[L.off[L.em.r] + i]; return c > 3 ? 4 : (int)c; }   // k_seed<true>, a read with an N (rare): byte by byte
	const u64 g = L.qoff + (u64)i;
	const u32 w = (u32)(g >> 4);
	if (w != L.win_w) {
		if (w == L.win2_w) { const u64 t = L.win; L.win = L.win2; L.win2 = t; L.win2_w = L.win_w; }   // (win2_w stays ~0u where nothing prefetches)
		else L.win = nib[w];
		L.win_w = w;
	}
	return (int)(L.win >> (((u32)g & 15) << 2)) & 15;
}

// 4-bit packing of the batch's bases for seed_q: word w holds bases [16w, 16w+16) of the flat read array, codes > 3 become 4
__global__ void __launch_bounds__(256) k_pack_reads(Batch B, u64 n_words)
{
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (u64)gridDim.x * blockDim.x) {
		const uint4 v = ((const uint4*)B.seq)[w];      // B.seq is padded to a multiple of 16 bytes
		const u32 d[4] = { v.x, v.y, v.z, v.w };
		u64 o = 0;
		for (int k = 0; k < 4; ++k)
			for (int b = 0; b < 4; ++b) {
				u32 c = d[k] >> (8 * b) & 255;
				o |= (u64)(c > 3 ? 4 : c) << ((k * 4 + b) * 4);
			}
		B.seq_nib[w] = o;
	}
}

// 2-bit copy of every read from a word boundary of its own: word j of read r holds its bases [16j, 16j+16), N and positions past
// the end as 0; a read with an N is flagged (its lane uses the 4-bit array instead)
__global__ void __launch_bounds__(256) k_pack_reads2b(Batch B)
{
	const u64 total = (u64)B.n_reads * (u64)B.rd_words;
	for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (u64)gridDim.x * blockDim.x) {
		const int r = (int)(t / (u32)B.rd_words), j = (int)(t % (u32)B.rd_words);
		const i64 b0 = B.off[r] + 16 * (i64)j, left = B.off[r + 1] - b0;
		u32 o = 0; bool any_n = false;
		for (int k = 0; k < 16 && k < left; ++k) {
			const u32 c = B.seq[b0 + k];
			if (c > 3) any_n = true; else o |= c << (2 * k);
		}
		B.seq_2b[t] = o;
		if (any_n) B.seq_flags[r] = 1;
	}
}

DEVFN void smem_finish(SeedLane &L) { if (L.pass == 1) { L.x = L.ret; L.st = SS_PASS1; } else L.st = SS_PASS2; }

// start backward row i (bwt.c:326-345); rows without a usable base (i < 0 or N) need no extension at all
DEVFN void bwd_begin_row(SeedLane &L, const SeedStack &S, const u64 *nib, int m)
{
	for (;;) {
		if (L.i < L.lo) { smem_finish(L); return; }
		L.c = L.i < 0 ? -1 : seed_q(L, nib, L.i);
		if (L.c > 3) L.c = -1;
		L.j = 0; L.nc = 0; L.ncl = 0; L.last_x2 = 0; L.srem = L.smask; L.snew = 0;
		if (L.c >= 0) { if (m > 0) L.code = (u32)L.c << (2 * (m - 1)) | L.code >> 2; L.st = SS_BWD; return; }
		// every interval stops here; only the longest one (first in prev[]) can be a new MEM
		if (!L.any || L.i + 1 < L.last_start) {
			if (L.nprev > 0) { BiIntv p = S.load(L, 0); L.em.add(p.x0, p.x2, L.i + 1, (int)p.info); }   // (a short entry on top: too short to be reported)
			L.any = true; L.last_start = L.i + 1;
		}
		smem_finish(L);
		return;
	}
}

DEVFN void fwd_finish(SeedLane &L, const SeedStack &S, const u64 *nib, int m)
{	// forward sweep done: stack[0 .. n0) holds the change points, longest match on top
	L.top = L.n0 - 1; L.nprev = L.n0;
	L.slot = (L.slot == 0 ? S.n_lds : L.slot) - 1;
	L.any = false; L.last_start = 0;
	L.i = L.sx - 1;
	bwd_begin_row(L, S, nib, m);
}

// Bi-interval of the first j bases of the m-mer with 2-bit code `w` (first base most significant) from the prefix tables.
// The tables are filled at start-up by the same fm_extend1 the sweep uses (k_ptab_level), so the values are those the
// reference's step-by-step extension would produce.  A look-up is one 16-byte entry instead of the two 64-byte index
// blocks of an extension, and the look-ups of consecutive steps fall into the same 16*m-byte record.
DEVFN void ptab_load(const DevIndex &ix, int j, u32 w, BiIntv &out)
{
	out = SeedStack::unpack(ix.ptab[(u64)w * (u32)ix.ptab_m + (u32)(j - 1)]);
}

// One memory round trip per extension step (k_seed<.., MRG>, k_seed3<.., true>; 32-byte index layout).  A wave's lanes are a mix
// of table look-ups (short matches) and block look-ups (long ones); written as `if (short) table else blocks` each side waits for
// its own loads before the other side issues, and the block side itself comes in two dependent parts (dev_fm.h, Occ32Data).  Here
// every lane issues the same eight range-checked loads -- one table entry, two blocks and their superblock entries, and (pf_off)
// the interval-stack entry of its NEXT step -- with out-of-range offsets for the ones it does not need, and the wave waits once.
// Returns N_blk (0 for a table look-up).
struct SeedBufs { Occ32Bufs occ; BufRsrc ptab, stk, nib; };
template <bool WIN = false> DEVFN int ext_one_trip(const DevIndex &ix, const SeedBufs &bf, bool ext, bool blocks, const BiIntv &src, int c, int back, int tl, u32 code, BiIntv &ok, u32 pf_off, uint4 &pf, u32 win_off = BUF_OOB, u64 *win = nullptr)
{
	const Occ32Pos pp = occ32_pos(ix, src, back);      // (lanes that do not extend hold src = 0: the arithmetic is harmless, nothing is loaded)
	Occ32Data od;
	uint4 te = buf_load16(bf.ptab, ext && !blocks ? (code * (u32)ix.ptab_m + (u32)(tl - 1)) << 4 : BUF_OOB);
	occ32_issue(ix, bf.occ, blocks, pp, c, od);
	pf = buf_load16(bf.stk, pf_off);                   // (BUF_OOB: zeros, i.e. "no entry", SeedStack::pack never yields w == 0)
	uint2 wv = make_uint2(0, 0);
	if (WIN) { wv = buf_load8(bf.nib, win_off); DEV_KEEP(wv.x); DEV_KEEP(wv.y); *win = (u64)wv.y << 32 | wv.x; }      // (the 16 bases the sweep walks into next)
	occ32_keep(od); dev_keep(te); dev_keep(pf);
	int nb = 0;
	if (ext) {
		if (!blocks) ok = SeedStack::unpack(te);
		else nb = occ32_finish(ix, src, c, back, pp, od, ok);
	}
	return nb;
}

// 2-bit code of q[x..x+m), first base most significant; N and positions past the read's end read as 0 (no match that the
// tables are asked about extends over them)
DEVFN u32 window_code(SeedLane &L, const u64 *nib, int x, int m)
{
	u32 rc = 0;
	for (int k = 0; k < m; ++k) { const int g = x + k; rc = rc << 2 | (u32)((g < L.len ? seed_q(L, nib, g) : 0) & 3); }
	return rc;
}


DEVFN void smem_start(const DevIndex &ix, SeedLane &L, const SeedStack &S, const u64 *nib, int x, u64 min_intv, int pass)
{
	L.pass = pass; L.sx = x; L.min_intv = min_intv < 1 ? 1 : min_intv;
	const int c0 = seed_q(L, nib, x);
	if (c0 > 3) { L.ret = x + 1; smem_finish(L); return; }   // bwt.c:296
	fm_init(ix, c0, L.ik); L.ik.info = (u64)(x + 1);
	L.i = x + 1; L.n0 = 0; L.slot = 0; L.smask = 0;
	L.code = window_code(L, nib, x, ix.ptab_m);
	if (L.i >= L.len || seed_q(L, nib, L.i) > 3) {                          // nothing (more) to extend: push and go backward
		S.push_fw(L, L.ik); L.ret = (int)L.ik.info;
		fwd_finish(L, S, nib, ix.ptab_m);
	} else L.st = SS_FWD;
}

// After seeding: per read, sort the intervals (bwamem.c:187; equal keys are identical intervals, so tie order is immaterial),
// count its SA lookups (mem_chain's inner loop bounds, bwamem.c:304-305) and reserve its slot range and B-tree nodes.  Kept out
// of k_seed so that no lane of the seeding state machine ever waits for another lane's sort.
__global__ void __launch_bounds__(256) k_publish(bwagpu_opt_t opt, Batch B)
{
	u64 nintv = 0;
	for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < B.n_reads; r += gridDim.x * blockDim.x) {
		int n = B.intv_n[r];
		if (n > B.mem_cap) { atomicOr(&B.ctr->overflow, 16ull); B.intv_n[r] = 0; n = 0; }      // (seeding tasks count past a full list: the batch is redone with longer lists)
		B.intv_off[r] = (i64)r * B.mem_cap;
		B.seed_n[r] = 0; B.seed_off[r] = 0; B.node_off[r] = 0;
		if (n == 0) continue;
		Intv3 *iv = B.intv + (size_t)r * B.mem_cap;
		dev_introsort(iv, n, IntvInfoLess());
		i64 ns = 0;
		for (int i = 0; i < n; ++i) {
			const u64 x2 = iv[i].x2;
			u64 step = x2 > (u64)opt.max_occ ? x2 / opt.max_occ : 1;
			u64 cnt = (x2 + step - 1) / step;
			ns += (i64)(cnt < (u64)opt.max_occ ? cnt : (u64)opt.max_occ);
		}
		u64 soff = atomicAdd(&B.ctr->seed_used, (unsigned long long)ns);
		u64 nnode = (u64)ns / 4 + 2;
		u64 noff = atomicAdd(&B.ctr->node_used, (unsigned long long)nnode);
		if (soff + ns > (u64)B.slot_cap) { atomicOr(&B.ctr->overflow, 2ull); B.intv_n[r] = 0; continue; }
		if (noff + nnode > (u64)B.node_cap) { atomicOr(&B.ctr->overflow, 4ull); B.intv_n[r] = 0; continue; }
		B.seed_n[r] = (i32)ns; B.seed_off[r] = (i64)soff; B.node_off[r] = (i64)noff;
		nintv += (u64)n;
		// ... and its SA rows (mem_chain's k-loop, bwamem.c:304-305) straight away: the intervals are in this lane's cache lines now (k_expand was a
		// launch of its own re-reading them, 2 ms per million reads)
		i64 sl_ = (i64)soff;
		for (int i = 0; i < n; ++i) {
			const Intv3 p = iv[i];
			const int step = p.x2 > (u64)opt.max_occ ? (int)(p.x2 / opt.max_occ) : 1;
			int count = 0;
			const i32 qb = (i32)(p.info >> 32), ln = (i32)((u32)p.info - (u32)(p.info >> 32));
			for (i64 k = 0; (u64)k < p.x2 && count < opt.max_occ; k += step, ++count, ++sl_) {
				B.slot_pos[sl_] = p.x0 + (u64)k;
				B.slot_qbeg[sl_] = qb;
				B.slot_len[sl_] = ln;
			}
		}
	}
	if (B.stats) atomicAdd(&B.ctr->n_intv, (unsigned long long)nintv);
}

// k_publish + k_expand for long reads: one workgroup per read.  A 10 kb read leaves a few thousand intervals and as many SA rows; sorted
// and expanded by one lane (24-byte records in HBM, a dependent access per comparison) that was 0.17 s per batch whatever its size --
// a floor, not a rate.  Here the keys are sorted in LDS (bitonic network over {info, index}; equal keys are identical intervals, so any
// sorting order gives the reference's result), the records are permuted through the block's share of the seeding scratch area, the slot
// counts are a block reduction and every interval writes its SA rows from its own prefix sum.  A read with more than PUB_MAX intervals
// takes the serial path inside this kernel.  (Round 3 launched it with one workgroup per 256 lanes of the lane-per-read seeding kernel -- 24 workgroups
// for a 6000-read batch, 48 ms; it needs 96 KB of scratch per workgroup, not a seeding block's 80 MB: up to 2048 workgroups now.)
#define PUB_MAX 4096
__global__ void __launch_bounds__(256) k_publish_blk(bwagpu_opt_t opt, Batch B)
{
	__shared__ u64 s_key[PUB_MAX];
	__shared__ unsigned short s_idx[PUB_MAX];
	__shared__ u32 s_part[256];
	__shared__ u64 s_bcast[2];
	const int tid = threadIdx.x;
	Intv3 *tmp = (Intv3*)B.tmp_intv + (size_t)blockIdx.x * PUB_MAX;       // PUB_MAX records of the seeding kernels' spill area (free since k_seed ended) per workgroup: the launch is sized for that
	const size_t tmp_recs = PUB_MAX;
	u64 nintv = 0;
	for (int r = blockIdx.x; r < B.n_reads; r += gridDim.x) {
		int n = B.intv_n[r];
		if (n > B.mem_cap) { n = 0; if (tid == 0) atomicOr(&B.ctr->overflow, 16ull); }      // (seeding tasks count past a full list: the batch is redone with longer lists)
		__syncthreads();
		if (tid == 0) { B.intv_off[r] = (i64)r * B.mem_cap; B.seed_n[r] = 0; B.seed_off[r] = 0; B.node_off[r] = 0; if (n == 0) B.intv_n[r] = 0; }
		if (n == 0) continue;                      // (uniform over the block)
		Intv3 *iv = B.intv + (size_t)r * B.mem_cap;
		if (n > PUB_MAX || (size_t)n > tmp_recs) { if (tid == 0) dev_introsort(iv, n, IntvInfoLess()); }
		else if (n > 1) {
			int N = 2; while (N < n) N <<= 1;
			for (int i = tid; i < N; i += blockDim.x) { s_key[i] = i < n ? iv[i].info : ~0ull; s_idx[i] = (unsigned short)i; }
			__syncthreads();
			for (int k = 2; k <= N; k <<= 1)
				for (int j = k >> 1; j > 0; j >>= 1) {
					for (int i = tid; i < N; i += blockDim.x) {
						const int p = i ^ j;
						if (p > i) {
							const u64 a = s_key[i], b = s_key[p];
							const bool up = (i & k) == 0;
							if (up ? a > b : a < b) { s_key[i] = b; s_key[p] = a; const unsigned short t = s_idx[i]; s_idx[i] = s_idx[p]; s_idx[p] = t; }
						}
					}
					__syncthreads();
				}
			for (int i = tid; i < n; i += blockDim.x) tmp[i] = iv[s_idx[i]];
			__syncthreads();
			for (int i = tid; i < n; i += blockDim.x) iv[i] = tmp[i];
		}
		__syncthreads();
		// SA rows per interval (mem_chain's k-loop bounds, bwamem.c:304-305): block-wide sum, then one reservation
		u32 mine = 0;
		for (int i = tid; i < n; i += blockDim.x) {
			const u64 x2 = iv[i].x2;
			const u64 step = x2 > (u64)opt.max_occ ? x2 / opt.max_occ : 1;
			const u64 cnt = (x2 + step - 1) / step;
			mine += (u32)(cnt < (u64)opt.max_occ ? cnt : (u64)opt.max_occ);
		}
		s_part[tid] = mine;
		__syncthreads();
		if (tid == 0) {
			u64 ns = 0;
			for (int t = 0; t < (int)blockDim.x; ++t) ns += s_part[t];
			const u64 soff = atomicAdd(&B.ctr->seed_used, (unsigned long long)ns);
			const u64 nnode = ns / 4 + 2;
			const u64 noff = atomicAdd(&B.ctr->node_used, (unsigned long long)nnode);
			u64 okv = 1;
			if (soff + ns > (u64)B.slot_cap) { atomicOr(&B.ctr->overflow, 2ull); B.intv_n[r] = 0; okv = 0; }
			else if (noff + nnode > (u64)B.node_cap) { atomicOr(&B.ctr->overflow, 4ull); B.intv_n[r] = 0; okv = 0; }
			else { B.seed_n[r] = (i32)ns; B.seed_off[r] = (i64)soff; B.node_off[r] = (i64)noff; nintv += (u64)n; }
			s_bcast[0] = okv; s_bcast[1] = soff;
		}
		__syncthreads();
		if (s_bcast[0]) {
			// k_expand's loop (bwamem.c:304-305): an interval's first slot is the sum of the counts of all intervals before it in sorted
			// order -- a running prefix over chunks of 256 intervals
			const u64 soff = s_bcast[1];
			__syncthreads();
			u32 base = 0;
			for (int i0 = 0; i0 < n; i0 += blockDim.x) {
				const int i = i0 + tid;
				Intv3 p; p.x0 = p.x2 = p.info = 0; u32 cnt = 0; int step = 1;
				if (i < n) {
					p = iv[i];
					step = p.x2 > (u64)opt.max_occ ? (int)(p.x2 / opt.max_occ) : 1;
					const u64 c64 = (p.x2 + (u64)step - 1) / (u64)step;
					cnt = (u32)(c64 < (u64)opt.max_occ ? c64 : (u64)opt.max_occ);
				}
				s_part[tid] = cnt;
				__syncthreads();
				if (tid == 0) { u32 run = 0; for (int t = 0; t < (int)blockDim.x; ++t) { const u32 v = s_part[t]; s_part[t] = run; run += v; } s_bcast[0] = run; }
				__syncthreads();
				if (i < n) {
					const i32 qb = (i32)(p.info >> 32), sl = (i32)((u32)p.info - (u32)(p.info >> 32));
					u64 s = soff + base + s_part[tid];
					i64 k = 0;
					for (u32 c = 0; c < cnt; ++c, k += step, ++s) { B.slot_pos[s] = p.x0 + (u64)k; B.slot_qbeg[s] = qb; B.slot_len[s] = sl; }
				}
				base += (u32)s_bcast[0];
				__syncthreads();
			}
		}
		__syncthreads();
	}
	if (B.stats && tid == 0) atomicAdd(&B.ctr->n_intv, (unsigned long long)nintv);
}

// RD: the batch's reads are short enough for an LDS copy (Batch::rd_words > 0)
// BLK: how the extension reads the index -- 1 (default): the 32-byte layout (DevIndex::occ32), each lane fetching its own blocks; 0: the
//      reference-format 64-byte blocks, per lane (option occ32 = 0).  A compile-time choice, so that no path pays for another's registers.
//      (A third form -- the 64-byte blocks fetched by quads of lanes -- raised the micro-benchmark's request ceiling 2.2x and made this kernel
//      1.6x slower, profiles/r03_seed_variants.md; deleted in round 4.)
// OCC: waves per SIMD the register allocation aims at
// MRG (BLK == 1 only): memory round trips per wave iteration.  0: as the compiler schedules them -- interval-stack entry from HBM scratch,
//      then the table look-ups, then the index blocks, then the blocks' first words for the lanes extending by A: up to four dependent
//      trips.  2: table entries and whole blocks are issued together and waited for once (ext_one_trip), and a backward row's next
//      interval-stack entry, when it lives in HBM scratch, is fetched in the same trip, one step ahead: one trip per iteration.  Measured
//      (BENCH_r03 variants): short reads 83.3 -> 87.6 ms (the kernel is bound by the request rate of its live lanes, not by trips per
//      iteration), long reads 515 -> 451 ms (few lanes, every trip exposed): the default for long-read batches only.  (The intermediate
//      form without the prefetch, MRG = 1, was slower than both and is gone.)
// LR (long-read batches, option seed_tasks): pass 1 of a 10 kb read (bwamem.c:147-157) is a chain of ~500 searches x -> ret(x), ~40 000 dependent index
//      look-ups, and a batch has fewer reads than the chip has SIMDs.  Round 3 cut the chain into chunks whose chains were to merge with the
//      read's own; on noisy reads they almost never do (two chains meet only where a match's suffix is already unique: measured 38 ms of chunk
//      workers + 316 ms of a lane-per-read kernel recomputing nearly everything, profiles/r04_longread_kernel_stats.csv).  The chain is not needed:
//      bwt_smem1(x) with min_intv 1 returns exactly the matches through x that can be extended neither way (each change point of the forward
//      sweep that survives the backward rows down to its own left end, bwt.c:326-345) -- a property of the read, whoever asks -- and the chain only
//      makes sure that every such match is asked for once.  A match of at least min_seed_len bases covers a multiple of s = min_seed_len, so:
//      1 = TASK (read, position g = k s): the search at g, reporting the matches whose start lies in (g - s, g] -- each match is reported by
//          the first multiple of s it covers, once -- which also ends the backward sweep after s rows.  Tasks append to the read's list through an
//          atomic count (k_publish sorts; equal keys are identical intervals).  Their interval stacks are small (Batch::vr_room entries; a forward
//          sweep with more change points than that hands its task to a second launch of this instance on full-size stacks, Batch::vr_ovf_run).
//      2 = the ordinary lane-per-read kernel entered at pass 2 (bwamem.c:160-168), over the entries the tasks left behind pass 3's (k_seed3 runs first).
//      3 = TASK (read, pass-1 entry): ONE search of pass 2 -- bwt_smem1 from the middle of that entry with min_intv = its occurrences + 1 -- every
//          match it returns appended like a pass-1 task's (pass 2's searches depend on pass 1's list, not on one another).
//      Exact by construction, no stitching; shorter matches are dropped by the length filter as ever.
//      SHORT-read batches use 1 and 3 for their HEAVY reads only (Batch::task_tpr, heavy_list).  Measured at 1 M reads
//      (profiles/r04_seed_iterations_per_read.log): 19 % of the reads take under 512 iterations of the lane-per-read kernel, 2.5 % over 4096, one 42 000
//      -- a read inside a tandem array, where EVERY forward step changes the interval size and every backward row walks ~150 stack entries -- and
//      that one read's dependent chain was the kernel's 87 ms, while a million reads' worth of requests keep the chip busy for ~35 ms.  No weight
//      k_seed3 can compute from its forward walk found those reads (12-mer repetitiveness put the 42 000-iteration read outside the top 1.5 %:
//      profiles/r04_seed_heavy_by_weight_ab.jsonl), so the lane-per-read kernel finds them itself: a lane that has spent Batch::seed_budget
//      iterations on a read gives it up -- nothing of it has been published: a read's count is written when it is done -- and lists it; the
//      listed reads' pass 1 then runs as tasks (critical path: one search with at most min_seed_len backward rows) and their pass 2 as tasks.
template<bool RD, bool STATS, int BLK, int OCC, int MRG = 0, int LR = 0>
__global__ void __launch_bounds__(256, OCC) k_seed(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	HIP_DYNAMIC_SHARED(uint4, seed_lds)
	const int cap = B.seed_stack_cap > 0 ? B.seed_stack_cap : B.max_len + 1 + PTAB_MAX;      // entries of a lane's spill area (tasks of long reads: small ones, see LR)
	const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
	SeedLane L;
	SeedStack S;
	S.lds_base = seed_lds; S.stride = blockDim.x; S.n_lds = B.seed_lds_ent;
	S.glob_base = B.tmp_intv; S.glob_cap = cap;
	S.virt_m = (ix.ptab_m >= 2 && opt.min_seed_len > ix.ptab_m && !B.seed_no_virt) ? ix.ptab_m : 0;
	L.smask = L.srem = L.snew = 0; L.ncl = 0;
	L.em.intv = B.intv; L.em.r = -1; L.em.cap = B.mem_cap; L.em.min_seed_len = opt.min_seed_len;
	L.em.split_len = split_len; L.em.split_width = (u64)opt.split_width; L.em.cand = 0; L.em.base = 0; L.em.shared_n = LR == 1 ? B.intv_n : nullptr;
	L.lo = -1;
	int vr_task = -1, vr_run = 0;       // LR == 1: the lane's task and whether its search has been started
	const unsigned long long n_tasks = LR == 1 ? (B.vr_ovf_run ? B.ctr->n_vr_ovf : B.task_tpr > 0 ? B.ctr->n_heavy * (unsigned long long)B.task_tpr : (unsigned long long)B.n_vreads)
									 : LR == 3 ? (B.ctr->n_p2_tasks < (unsigned long long)B.p2_cap ? B.ctr->n_p2_tasks : (unsigned long long)B.p2_cap) : 0;
	if (LR == 3) L.em.shared_n = B.intv_n;
	int p2_entry = 0;                   // LR == 3: the pass-1 entry the task re-seeds
	L.st = SS_FETCH; L.len = 0; L.qoff = 0; L.win = 0; L.win_w = ~0u; L.win2 = 0; L.win2_w = ~0u;
	u32 *rd_lds = (u32*)(seed_lds + (size_t)(B.seed_lds_ent ? B.seed_lds_ent : 1) * blockDim.x);   // (after the stacks)
	L.rd = rd_lds; L.rd_on = 0; L.raw = B.seq; L.off = B.off;
	const u64 *nib = RD ? nullptr : B.seq_nib;
	u32 nblk = 0, ntab = 0;
	// Reads are drawn from the batch counter 64 at a time into a pool of the wave, and lanes that finish a read take the pool's
	// next one: a per-lane atomicAdd would be 10^6 same-address atomics per batch, which alone take ~13 ms on this chip.
	int pool_base = 0, pool_cnt = 0, pool_r = 0;
	// The bookkeeping between extensions (next read, next search of a pass, publishing a read's intervals) is a few hundred
	// instructions that a wave executes whenever ANY of its lanes needs them -- with 64 lanes, in nine iterations out of ten, for one
	// or two lanes each time (measured: 700 VALU instructions per iteration, 280 of them the extension).  Lanes therefore wait in
	// their bookkeeping state until eight of them have gathered (or three iterations have passed, or nobody can extend), and the
	// wave then runs that code once for all of them.
	int deferred = 0;
	u32 n_iter = 0, n_slow = 0, n_ext_lanes = 0, n_deep = 0, n_deep_l = 0, n_pf_l = 0, n_win_l = 0;
	u32 n_done_l = 0, n_wait_l = 0, n_slowrun_l = 0, n_first_done = 0;      // STATS: where the lane-slots that do not extend go (prof[2..7])
	u32 n_x2[6] = { 0, 0, 0, 0, 0, 0 }; bool run_f = false, run_b = false;   // STATS: Counters::seed_x2
	u32 my_iter = 0;                                                        // iterations this lane has spent on its current read (the budget of short-read batches; STATS: Counters::seed_hist)
	// MRG == 2: the stack entry this lane's next backward step will read, fetched a step ahead (pf.w != 0: valid -- an entry's `info`, its
	// match's end position >= 1, sits in the top half of w).  A backward step that is not the last of its row is always followed by the
	// step for entry j + 1 of the same row, and the steps in between write survivors at depths <= j only (SeedStack::store).
	uint4 pf = make_uint4(0, 0, 0, 0);
	SeedBufs bf;
	if (MRG && BLK == 1) {
		bf.occ = occ32_bufs(ix); bf.ptab = buf_rsrc(ix.ptab, ix.ptab_bytes);
		// (the workgroup's share of the spill area: the descriptor's base is a per-workgroup scalar, so the whole area may exceed 4 GiB)
		bf.stk = buf_rsrc((const u8*)B.tmp_intv + (size_t)blockIdx.x * blockDim.x * (size_t)cap * sizeof(BiIntv), MRG == 2 ? (u64)blockDim.x * (u64)cap * sizeof(BiIntv) : 0);
		bf.nib = buf_rsrc(B.seq_nib, MRG == 2 && !RD ? B.seq_nib_bytes : 0);
	}
	while (__ballot(L.st != SS_DONE)) {
		if (STATS) ++n_iter;
		if (LR == 0 && (STATS || B.seed_budget > 0)) {
			if (L.st != SS_DONE && L.st != SS_FETCH) ++my_iter;
			if (B.seed_budget > 0 && my_iter > (u32)B.seed_budget && L.st != SS_DONE && L.st != SS_FETCH && L.st != SS_FINAL) {
				// this read is one of the batch's heavy ones: given up here (its interval count still says "pass 3 only"), finished by the task kernels
				const unsigned long long k = atomicAdd(&B.ctr->n_heavy, 1ull);
				B.heavy_list[k] = L.em.r;
				if (STATS) { atomicAdd(&B.ctr->seed_hist[31], 1ull); atomicAdd(&B.ctr->seed_hist[63], (unsigned long long)my_iter); }
				my_iter = 0; L.st = SS_FETCH;
			}
		}
		const bool slow = L.st < SS_FWD || L.st == SS_FINAL;
		const u64 sm = __ballot(slow);
		bool run_slow = false;
		if (sm) {
			const u64 am = __ballot(1);
			run_slow = __popcll(sm) >= 8 || sm == am || ++deferred >= 3;
		}
		if (STATS) {
			const u64 dm = __ballot(L.st == SS_DONE);
			n_done_l += (u32)__popcll(dm);
			if (dm && n_first_done == 0) n_first_done = n_iter;
			if (run_slow) n_slowrun_l += (u32)__popcll(sm); else n_wait_l += (u32)__popcll(sm);
		}
		if (run_slow) {
			deferred = 0; if (STATS) ++n_slow;
			if (L.st == SS_FINAL) {
				if (STATS && LR == 0) { const int bin = my_iter ? 32 - __clz((int)my_iter) : 0; atomicAdd(&B.ctr->seed_hist[bin & 31], 1ull); atomicAdd(&B.ctr->seed_hist[32 + (bin & 31)], (unsigned long long)my_iter); }
				my_iter = 0;
				if (L.em.overflow) atomicOr(&B.ctr->overflow, 16ull);
				else if (LR != 1 && LR != 3) B.intv_n[L.em.r] = L.em.n;       // (tasks counted their entries as they went)
				L.st = SS_FETCH;
			}
			const bool want = L.st == SS_FETCH;
			const u64 wm = __ballot(want);
			if (wm) {
				if (pool_cnt == 0) {
					const int first = __ffsll((unsigned long long)__ballot(1)) - 1;        // lane 0 may already have left the loop
					const unsigned long long old = atomicAdd(LR == 1 ? (B.vr_ovf_run ? &B.ctr->next_vovf : &B.ctr->next_vread) : LR == 3 ? &B.ctr->next_p2 : &B.ctr->next_read, (threadIdx.x & 63) == first ? 64ull : 0ull);
					pool_base = __shfl((int)old, first); pool_cnt = 64;
					// lane l looks up the pool's read number l now: a lane taking a read later gets it from a register of the wave
					// instead of a memory round trip of its own in front of the reads of the read's data
					{ const int idx = pool_base + (int)(threadIdx.x & 63); pool_r = LR == 1 || LR == 3 ? idx : (idx < B.n_reads ? (B.seed_order ? B.seed_order[idx] : idx) : 0); }
					// the waves holding the (predicted) heaviest reads get issue priority: a lane's long chain of dependent extensions then
					// advances at the pace of the wave alone on its SIMD instead of a quarter of it
					if (B.seed_order && B.seed_prio) { if (pool_base < B.n_reads / 32) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
				}
				const int rank = __popcll(wm & ((1ull << (threadIdx.x & 63)) - 1));
				// k_seed3 ran first: it left the read's LAST-like seeds in the list and a repetitiveness weight by which the reads were
				// ordered heaviest first (a read inside a repeat family is a chain of 10-20 k dependent blocks -- started last, it alone
				// kept the kernel running for another 15 ms)
				const int r = __shfl(pool_r, (64 - pool_cnt + rank) & 63);
				if (want && rank < pool_cnt) {
					const int idx = pool_base + rank;
					if (LR == 1 || LR == 3 ? (unsigned long long)idx >= n_tasks : idx >= B.n_reads) L.st = SS_DONE;
					else if (LR == 3) {       // one search of pass 2: from the middle of pass-1 entry p2_entry of its read, every match of at least its occurrences + 1
						const i64 t = B.p2_tasks[r];
						const int rr = (int)(t >> 32);
						p2_entry = (int)(u32)t;
						L.em.r = rr; L.qoff = (u64)B.off[rr]; L.len = (int)(B.off[rr + 1] - B.off[rr]);
						L.win_w = ~0u; L.win2_w = ~0u;
						L.em.overflow = false;
						vr_task = r; vr_run = 0; L.lo = -1;
						L.st = SS_PASS1;
					} else if (LR == 1) {       // a task: its read by bisection of the reads' first tasks (or, short reads, by its rank among the heavy ones), its position from its rank among the read's tasks
						const int t = B.vr_ovf_run ? B.vr_ovf[r] : r;
						int lo = 0, hi = B.n_reads;
						if (B.task_tpr > 0) lo = B.heavy_list[t / B.task_tpr];
						else while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (B.vr_first[mid] <= t) lo = mid; else hi = mid; }
						const int g = B.task_tpr > 0 ? (t % B.task_tpr) * B.task_step : (t - B.vr_first[lo]) * B.task_step;
						L.em.r = lo; L.qoff = (u64)B.off[lo]; L.len = (int)(B.off[lo + 1] - B.off[lo]);
						L.win_w = ~0u; L.win2_w = ~0u;
						L.em.overflow = false;
						vr_task = t; vr_run = 0;
						L.lo = g - B.task_step > -1 ? g - B.task_step : -1;
						L.x = g; L.st = L.len >= opt.min_seed_len ? SS_PASS1 : SS_FINAL;       // (mem_chain returns at once for shorter reads, bwamem.c:286)
					} else {
						L.em.r = r; L.qoff = (u64)B.off[r]; L.len = (int)(B.off[r + 1] - B.off[r]);
						L.win_w = ~0u;
						if (RD) {
							L.rd_on = !B.seq_flags[r];      // the read's bases into LDS (2 bits each; at most four 16-byte pieces)
							const uint4 *src = (const uint4*)(B.seq_2b + (size_t)r * (u32)B.rd_words);
							u32 *dst = rd_lds + threadIdx.x;
							const int n4 = B.rd_words >> 2;
#pragma unroll
							for (int q = 0; q < 4; ++q)
								if (q < n4) {
									const uint4 v = src[q];
									dst[(4 * q + 0) * blockDim.x] = v.x; dst[(4 * q + 1) * blockDim.x] = v.y; dst[(4 * q + 2) * blockDim.x] = v.z; dst[(4 * q + 3) * blockDim.x] = v.w;
								}
						}
						if (B.seed_pass3_inline) B.intv_n[r] = 0;
						L.em.n = B.seed_pass3_inline ? 0 : B.intv_n[r]; L.em.base = L.em.n; L.em.overflow = false;   // (base: the entries after pass 3's)
						L.em.cand = 0;
						if (LR == 2) {       // pass 2 only: the tasks' entries follow pass 3's; every one of them is looked at (no emitter's bit mask: base - 64)
							const int n3 = B.intv_n3[r];
							if (L.em.n > L.em.cap) { atomicOr(&B.ctr->overflow, 16ull); L.em.n = n3; B.intv_n[r] = n3; }      // (a task ran out of room: the batch is redone with longer lists)
							L.em.base = n3 - 64; L.old_n = L.em.n; L.k2 = n3;
							if (L.len >= opt.min_seed_len) L.st = SS_PASS2;
						} else
						if (L.len >= opt.min_seed_len) { L.x = 0; L.st = SS_PASS1; }   // else mem_chain returns at once (bwamem.c:286): draw the next read
					}
				}
				const int took = __popcll(wm) < pool_cnt ? __popcll(wm) : pool_cnt;
				pool_base += took; pool_cnt -= took;
			}
			// ---- advance the lane's state up to its next extension (a few steps: skipped bases, searches that end at once) ----
			for (int rep = 0; rep < 3; ++rep) {
				switch (L.st) {
				case SS_PASS1:   // pass 1: all SMEMs, left to right (bwamem.c:147-157)
					if (LR == 1) {       // a task is ONE search, at its own position (nothing to do where the read holds an N: no match covers it)
						if (vr_run || L.x >= L.len || seed_q(L, nib, L.x) > 3) L.st = SS_FINAL;
						else { vr_run = 1; smem_start(ix, L, S, nib, L.x, 1, 1); }
						break;
					}
					if (LR == 3) {       // (bwamem.c:163-167)
						if (vr_run) L.st = SS_FINAL;
						else {
							vr_run = 1;
							const Intv3 p = L.em.mem()[p2_entry];
							const int start = (int)(p.info >> 32), end = (int)(u32)p.info;
							smem_start(ix, L, S, nib, (start + end) >> 1, p.x2 + 1, 1);       // (pass "1": when the search is over the lane comes back here, not to SS_PASS2)
						}
						break;
					}
					while (L.x < L.len && seed_q(L, nib, L.x) > 3) ++L.x;
					if (L.x >= L.len) { L.old_n = L.em.n; L.k2 = L.em.base; L.st = SS_PASS2; }   // pass 2 re-seeds pass 1's SMEMs (the entries after pass 3's)
					else smem_start(ix, L, S, nib, L.x, 1, 1);
					break;
				case SS_PASS2: { // pass 2: re-seed from the middle of long, rare SMEMs (bwamem.c:160-168)
					bool started = false;
					while (L.k2 < L.old_n && !started) {
						const int kk = L.k2 - L.em.base;          // skip the entries the emitter already found not to qualify
						if (kk < 64) {
							const u64 rest = L.em.cand >> kk;
							if (rest == 0) { L.k2 = L.em.base + 64 < L.old_n ? L.em.base + 64 : L.old_n; continue; }
							L.k2 += __ffsll((unsigned long long)rest) - 1;
							if (L.k2 >= L.old_n) break;
						}
						Intv3 p = L.em.mem()[L.k2++];
						int start = (int)(p.info >> 32), end = (int)(u32)p.info;
						if (end - start >= split_len && p.x2 <= (u64)opt.split_width) { smem_start(ix, L, S, nib, (start + end) >> 1, p.x2 + 1, 2); started = true; }
					}
					if (!started) { L.x = 0; L.st = (opt.max_mem_intv > 0 && B.seed_pass3_inline) ? SS_PASS3 : SS_FINAL; }
					break; }
				case SS_PASS3:   // pass 3: LAST-like seeds (bwamem.c:170-185, bwt_seed_strategy1 bwt.c:358-379)
					while (L.x < L.len && seed_q(L, nib, L.x) > 3) ++L.x;
					if (L.x >= L.len) L.st = SS_FINAL;
					else {
						const int c0 = seed_q(L, nib, L.x);
						fm_init(ix, c0, L.ik); L.sx = L.x; L.i = L.x + 1;
						L.code = window_code(L, nib, L.x, ix.ptab_m);
						if (L.i >= L.len) { L.x = L.len; }
						else if (seed_q(L, nib, L.i) > 3) { L.x = L.i + 1; }
						else L.st = SS_STRAT;
					}
					break;
				default: break;
				}
			}
		}
		// ---- the one expensive, convergent step: an FM extension ---------------------------------------------------------
		const int st = L.st;
		if (STATS) n_ext_lanes += (u32)__popcll(__ballot(st == SS_FWD || st == SS_BWD || st == SS_STRAT));
		if (STATS && __ballot(st == SS_BWD && L.j >= S.n_lds && L.j < L.nprev)) ++n_deep;
		const bool ext = st == SS_FWD || st == SS_BWD || st == SS_STRAT;
		BiIntv ok, src, p;        // p: the stack entry a backward step extends (it lives for this iteration only)
		p.x0 = p.x1 = p.x2 = p.info = 0;
		const int back = st == SS_BWD;
		bool short_ent = false;
		int cb = 0, tl = 0;
		ok.x0 = ok.x1 = ok.x2 = ok.info = 0; src = ok;
		if (ext) {
			if (back) {
				if (L.j < L.nprev) {
					if (STATS && L.j >= S.n_lds) { ++n_deep_l; if (MRG == 2 && pf.w != 0) ++n_pf_l; }
					if (MRG == 2 && pf.w != 0) p = SeedStack::unpack(pf); else p = S.load(L, L.j);
				} else {                                 // the stack's entries are done: the short ones, longest first
					const int len = 32 - __clz((int)L.srem);
					L.srem &= ~(1u << (len - 1));
					p.x0 = p.x1 = p.x2 = 0; p.info = (u64)(L.i + 1 + len); short_ent = true;
				}
			}
			src.x0 = back ? p.x0 : L.ik.x0; src.x1 = back ? p.x1 : L.ik.x1; src.x2 = back ? p.x2 : L.ik.x2; src.info = 0;
			const int qi = back ? 0 : seed_q(L, nib, L.i);
			cb = back ? L.c : 3 - qi;
			tl = back ? (int)p.info - L.i : L.i - L.sx + 1;    // length of the extended match
		}
		// A match no longer than ptab_m bases has its bi-interval in the prefix tables (filled by the same extension routine at
		// start-up, k_ptab_level; a bi-interval is a function of the string, whichever way it was extended): one 16-byte
		// entry instead of two index blocks.  That covers the first steps of every forward search and, in the first
		// backward rows, the short change-point intervals, whose match q[i..end) is still short.
		const bool blocks = ext && tl > ix.ptab_m;
		if (MRG && BLK == 1) {
			u32 pf_off = BUF_OOB;       // byte offset of the lane's next stack entry in the spill area (SeedStack::glob_col, packed entries)
			if (MRG == 2 && back && !short_ent && S.n_lds && L.j + 1 < L.nprev && L.j + 1 >= S.n_lds)
				pf_off = ((u32)threadIdx.x * (u32)S.glob_cap + PTAB_MAX) * (u32)sizeof(BiIntv) + (u32)(L.top - (L.j + 1)) * (u32)sizeof(uint4);
			u32 nb;
			if (MRG == 2 && !RD) {
				// Reads too long for an LDS copy take their bases from the packed array in HBM, 16 per fetch -- a dependent round trip of its
				// own whenever some lane of the wave walks into a new word, i.e. in most iterations.  The word a lane will need next
				// (forward sweep: base i + 1, looked at when this step's result is in; backward row: base i - 1, at the row's end) comes
				// with this step's loads instead.
				u32 win_off = BUF_OOB, ww = 0; u64 wnew = 0;
				const int want = back ? L.i - 1 : L.i + 1;
				if (ext && want >= 0 && want < L.len) {
					ww = (u32)((L.qoff + (u64)want) >> 4);
					if (ww != L.win_w && ww != L.win2_w) win_off = ww << 3;
				}
				nb = (u32)ext_one_trip<true>(ix, bf, ext, blocks, src, cb, back, tl, L.code, ok, pf_off, pf, win_off, &wnew);
				if (win_off != BUF_OOB && B.seq_nib_bytes <= BUF_MAX_BYTES) { L.win2 = wnew; L.win2_w = ww; if (STATS) ++n_win_l; }
			} else nb = (u32)ext_one_trip(ix, bf, ext, blocks, src, cb, back, tl, L.code, ok, pf_off, pf);
			if (STATS && ext) { if (blocks) nblk += nb; else ++ntab; }
		}
		if (STATS && ext && blocks) {      // how many of the index look-ups are on a unique match (an interval of one row)
			const bool one = src.x2 == 1;
			++n_x2[back ? 1 : 0]; if (one) ++n_x2[back ? 3 : 2];
			if (one && !(back ? run_b : run_f)) ++n_x2[back ? 5 : 4];
			if (back) run_b = one; else run_f = one;
		}
		if (ext) {
			if (MRG && BLK == 1) ;
			else if (!blocks) { ptab_load(ix, tl, L.code, ok); if (STATS) ++ntab; }
			else { const u32 nb = fm_extend1<BLK>(ix, src, cb, back, ok); if (STATS) nblk += nb; }
			if (LR == 1 && st == SS_FWD && L.n0 >= B.vr_room) {
				// the task's interval stack is full (a forward sweep with hundreds of change points: tandem repeats): nothing has been reported yet --
				// reports come from the backward rows -- so the task is handed to the second launch, which has full-size stacks, as it is
				const unsigned long long k = atomicAdd(&B.ctr->n_vr_ovf, 1ull);
				B.vr_ovf[k] = vr_task;
				L.st = SS_FINAL;
			} else
			if (st == SS_FWD) {           // forward sweep of bwt_smem1a (bwt.c:304-320)
				bool stop = false;
				if (ok.x2 != L.ik.x2) {
					S.push_fw(L, L.ik); L.ret = (int)L.ik.info;
					if (ok.x2 < L.min_intv) stop = true;
				}
				if (!stop) {
					ok.info = (u64)(L.i + 1); L.ik = ok; ++L.i;
					if (L.i >= L.len || seed_q(L, nib, L.i) > 3) { S.push_fw(L, L.ik); L.ret = (int)L.ik.info; stop = true; }
				}
				if (stop) fwd_finish(L, S, nib, ix.ptab_m);
			} else if (st == SS_BWD) {    // one interval of one backward row (bwt.c:328-342)
				if (ok.x2 < L.min_intv) {
					if (L.nc == 0 && (!L.any || L.i + 1 < L.last_start)) {
						L.em.add(p.x0, p.x2, L.i + 1, (int)p.info); L.any = true; L.last_start = L.i + 1;
					}
				} else if (L.nc == 0 || ok.x2 != L.last_x2) {
					ok.info = p.info;
					const int nl = (int)p.info - L.i;      // bases of the extended match
					if (nl < S.virt_m) L.snew |= 1u << (nl - 1);
					else { S.store(L, L.ncl, ok); ++L.ncl; }   // in place: ncl <= j, or one past the deep end for a short entry that has grown up
					++L.nc; L.last_x2 = ok.x2;
				}
				if (!short_ent) ++L.j;
				if (L.j >= L.nprev && L.srem == 0) {
					if (L.nc == 0) smem_finish(L);
					else { L.nprev = L.ncl; L.smask = L.snew; --L.i; bwd_begin_row(L, S, nib, ix.ptab_m); }
				}
			} else {                      // bwt_seed_strategy1 (bwt.c:364-377)
				if (ok.x2 < opt.max_mem_intv && L.i - L.sx >= opt.min_seed_len) {
					if (ok.x2 > 0) L.em.add(ok.x0, ok.x2, L.sx, L.i + 1);
					L.x = L.i + 1; L.st = SS_PASS3;
				} else {
					L.ik = ok; ++L.i;
					if (L.i >= L.len) { L.x = L.len; L.st = SS_PASS3; }
					else if (seed_q(L, nib, L.i) > 3) { L.x = L.i + 1; L.st = SS_PASS3; }
				}
			}
		}
	}
	if (STATS) {
		atomicAdd(&B.ctr->occ_blocks, (unsigned long long)nblk); atomicAdd(&B.ctr->tab_lookups, (unsigned long long)ntab);
		for (int k = 0; k < 6; ++k) atomicAdd(&B.ctr->seed_x2[k], (unsigned long long)n_x2[k]);
		atomicAdd(&B.ctr->prof[9], (unsigned long long)n_win_l); atomicAdd(&B.ctr->prof[10], (unsigned long long)n_deep_l); atomicAdd(&B.ctr->prof[11], (unsigned long long)n_pf_l);
		if ((threadIdx.x & 63) == 0) { atomicAdd(&B.ctr->prof[12], (unsigned long long)n_deep); atomicAdd(&B.ctr->prof[13], (unsigned long long)n_iter); atomicAdd(&B.ctr->prof[14], (unsigned long long)n_slow); atomicAdd(&B.ctr->prof[15], (unsigned long long)n_ext_lanes);
			// lane-slots of lanes that have run out of reads, of lanes waiting in a bookkeeping state for the wave to run that code, of lanes running it; the
			// iteration at which a wave's first lane ran out (summed), the longest wave, and the waves that did any work
			atomicAdd(&B.ctr->prof[2], (unsigned long long)n_done_l); atomicAdd(&B.ctr->prof[3], (unsigned long long)n_wait_l); atomicAdd(&B.ctr->prof[4], (unsigned long long)n_slowrun_l);
			atomicAdd(&B.ctr->prof[5], (unsigned long long)(n_first_done ? n_first_done : n_iter)); atomicMax(&B.ctr->prof[6], (unsigned long long)n_iter); if (n_iter > 1) atomicAdd(&B.ctr->prof[7], 1ull); }
	}
}

// Pass 2 of the heavy reads of a short-read batch as tasks: one per pass-1 entry that qualifies (bwamem.c:163-164) -- the entries the pass-1 tasks
// left behind pass 3's.  One lane per heavy read; the searches themselves are k_seed<LR = 3>'s.
__global__ void __launch_bounds__(256) k_seed_p2_tasks(bwagpu_opt_t opt, Batch B)
{
	const int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
	const unsigned long long nh = B.ctr->n_heavy;
	for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < nh; k += (unsigned long long)gridDim.x * blockDim.x) {
		const int r = B.heavy_list[k];
		const int n3 = B.intv_n3[r], n = B.intv_n[r];
		if (n > B.mem_cap) { atomicOr(&B.ctr->overflow, 16ull); B.intv_n[r] = n3; continue; }      // (a pass-1 task ran out of room: the batch is redone with longer lists)
		const Intv3 *iv = B.intv + (size_t)r * (size_t)B.mem_cap;
		for (int e = n3; e < n; ++e) {
			const Intv3 p = iv[e];
			const int start = (int)(p.info >> 32), end = (int)(u32)p.info;
			if (end - start < split_len || p.x2 > (u64)opt.split_width) continue;
			const unsigned long long t = atomicAdd(&B.ctr->n_p2_tasks, 1ull);
			if ((long long)t < B.p2_cap) B.p2_tasks[t] = (i64)r << 32 | (u32)e; else atomicOr(&B.ctr->overflow, 32ull);      // (bit 5: the task list is redone larger)
		}
	}
}

// Pass 3 of mem_collect_intv (bwamem.c:170-185) -- the LAST-like seeds of bwt_seed_strategy1 (bwt.c:358-379) -- as a kernel of its
// own, run BEFORE passes 1-2 (k_seed): besides its seeds it leaves, per read, the summed occurrence counts of the seed-length matches it
// walked through -- a measure of how repetitive the read is, by which k_seed's reads are then ordered heaviest first.  The pass does not depend on passes 1-2 (it only appends to the read's interval list, which k_publish sorts afterwards), and it
// is a plain forward extension loop: run inside k_seed's state machine each of its steps paid for that machine's whole divergent
// iteration (~1060 VALU instructions); here an iteration is the extension plus a dozen instructions of control.  One lane per
// read, reads drawn from a per-wave pool.
template <int BLK, bool MRG = false> __global__ void __launch_bounds__(256, 3) k_seed3(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	SeedLane L;                   // only the read window (qoff, win, win_w, len), x, sx, i, ik, code and the emitter are used
	L.em.cap = B.mem_cap; L.em.min_seed_len = opt.min_seed_len; L.em.intv = B.intv; L.em.r = -1; L.em.n = 0; L.em.overflow = false;
	L.em.split_len = 0x7fffffff; L.em.split_width = 0; L.em.cand = 0; L.em.base = 0; L.em.shared_n = nullptr;
	L.len = 0; L.qoff = 0; L.win = 0; L.win_w = ~0u; L.win2 = 0; L.win2_w = ~0u; L.x = 0; L.sx = 0; L.i = 0; L.code = 0; L.rd = nullptr; L.rd_on = 0; L.raw = nullptr; L.off = nullptr;
	L.ik.x0 = L.ik.x1 = L.ik.x2 = L.ik.info = 0;
	const u64 *nib = B.seq_nib;
	const int lane = threadIdx.x & 63;
	enum { T_FETCH = 0, T_START, T_EXT, T_DONE };
	int st = T_FETCH, pool_base = 0, pool_cnt = 0;
	u32 nblk = 0, ntab = 0, weight = 0;
	SeedBufs bf;
	if (MRG && BLK == 1) { bf.occ = occ32_bufs(ix); bf.ptab = buf_rsrc(ix.ptab, ix.ptab_bytes); bf.stk = buf_rsrc(nullptr, 0); bf.nib = bf.stk; }
	while (__ballot(st != T_DONE)) {
		const u64 wm = __ballot(st == T_FETCH);
		if (wm) {
			if (pool_cnt == 0) {
				const int first = __ffsll((unsigned long long)__ballot(1)) - 1;
				const unsigned long long old = atomicAdd(&B.ctr->next_read3, lane == first ? 64ull : 0ull);
				pool_base = __shfl((int)old, first); pool_cnt = 64;
			}
			const int rank = __popcll(wm & ((1ull << lane) - 1));
			if (st == T_FETCH && rank < pool_cnt) {
				const int r = pool_base + rank;
				if (r >= B.n_reads) st = T_DONE;
				else {
					L.em.r = r; L.qoff = (u64)B.off[r]; L.len = (int)(B.off[r + 1] - B.off[r]);
					L.em.n = 0; L.em.overflow = false;
					B.intv_n[r] = 0; B.seed_w[r] = 0; weight = 0;
					if (B.intv_n3) B.intv_n3[r] = 0;
					if (L.len >= opt.min_seed_len && opt.max_mem_intv > 0) { L.x = 0; st = T_START; }   // mem_chain returns at once for shorter reads (bwamem.c:286)
				}
			}
			const int took = __popcll(wm) < pool_cnt ? __popcll(wm) : pool_cnt;
			pool_base += took; pool_cnt -= took;
		}
		if (st == T_START) {
			while (L.x < L.len && seed_q(L, nib, L.x) > 3) ++L.x;
			if (L.x >= L.len) {
				if (L.em.overflow) atomicOr(&B.ctr->overflow, 16ull); else { B.intv_n[L.em.r] = L.em.n; if (B.intv_n3) B.intv_n3[L.em.r] = L.em.n; }
				B.seed_w[L.em.r] = (i32)(weight > 0x3fffffffu ? 0x3fffffffu : weight);
				st = T_FETCH;
			} else {
				fm_init(ix, seed_q(L, nib, L.x), L.ik); L.sx = L.x; L.i = L.x + 1;
				L.code = window_code(L, nib, L.x, ix.ptab_m);
				if (L.i >= L.len) L.x = L.len;
				else if (seed_q(L, nib, L.i) > 3) L.x = L.i + 1;
				else st = T_EXT;
			}
		}
		const bool ext = st == T_EXT;        // one forward extension (bwt.c:364-377)
		const int tl = L.i - L.sx + 1;
		const bool blocks = ext && tl > ix.ptab_m;
		const int cb = blocks ? 3 - seed_q(L, nib, L.i) : 0;
		BiIntv ok; ok.x0 = ok.x1 = ok.x2 = ok.info = 0;
		if (MRG && BLK == 1) {
			uint4 none;
			BiIntv srcv = L.ik; if (!ext) srcv.x0 = srcv.x1 = srcv.x2 = 0;
			const int nb = ext_one_trip(ix, bf, ext, blocks, srcv, cb, 0, tl, L.code, ok, BUF_OOB, none);
			if (ext) { if (blocks) nblk += nb; else ++ntab; }
		}
		if (ext) {
			if (MRG && BLK == 1) ;
			else if (!blocks) { ptab_load(ix, tl, L.code, ok); ++ntab; }
			else nblk += fm_extend1<BLK>(ix, L.ik, cb, 0, ok);
			// occurrences of the seed-length match: the read's repetitiveness.  (Round 4 tried two richer weights -- 12-mer occurrences relative to chance,
			// then the number of seed-length stretches without occurrence, i.e. read errors -- to start the costly reads earlier: neither moved the
			// kernel, profiles/r04_seed_order_ab.log; what did is the iteration budget, k_seed's LR comment.)
			if (tl == opt.min_seed_len) weight += (u32)(ok.x2 > 65535 ? 65535 : ok.x2);
			if (ok.x2 < opt.max_mem_intv && L.i - L.sx >= opt.min_seed_len) {
				if (ok.x2 > 0) L.em.add(ok.x0, ok.x2, L.sx, L.i + 1);
				L.x = L.i + 1; st = T_START;
			} else {
				L.ik = ok; ++L.i;
				if (L.i >= L.len) { L.x = L.len; st = T_START; }
				else if (seed_q(L, nib, L.i) > 3) { L.x = L.i + 1; st = T_START; }
			}
		}
	}
	if (B.stats) { atomicAdd(&B.ctr->occ_blocks, (unsigned long long)nblk); atomicAdd(&B.ctr->tab_lookups, (unsigned long long)ntab); }
}

// SA lookups (bwt_sa, bwt.c:86-96): ~31 dependent LF steps each with the reference's sa_intv = 32, a single read when the
// SA has been densified.  The walk length is geometric, so lanes are persistent: every lane owns one lookup at a time and
// takes the next slot as soon as its walk reaches a sampled row -- all lanes execute the same LF step every iteration.
__global__ void __launch_bounds__(256) k_sa(DevIndex ix, Batch B)
{
	u64 n = B.ctr->seed_used;
	if (n > (u64)B.slot_cap) n = (u64)B.slot_cap;
	const u64 stride = (u64)gridDim.x * blockDim.x;
	u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
	u32 steps = 0;
	u64 k = 0, sa = 0; bool have = false;
	for (;;) {
		if (!have) {
			if (s >= n) break;
			k = B.slot_pos[s]; sa = 0; have = true;
			// When the slot arena overflows, k_publish drops the reads whose range does not fit and the batch is re-run with a larger
			// arena -- but this kernel still walks every slot below the capacity, including the never-written head of a dropped read's
			// range.  Whatever those bytes held before must not become an index address.
			if (k > ix.seq_len) k = 0;
		}
		if (k & ix.sa_mask) {           // one bwt_invPsi step (bwt.c:53-59)
			++sa;
			k = k == ix.primary ? 0 : fm_lf(ix, k);
		} else {                        // sampled row reached: finish this lookup, move to the lane's next slot
			u64 rbeg = sa + ix.sa[k >> ix.sa_shift];
			steps += (u32)sa;
			B.slot_pos[s] = rbeg;
			B.slot_rid[s] = dev_intv2rid(ix, (i64)rbeg, (i64)rbeg + B.slot_len[s]);   // bns_intv2rid of mem_chain (bwamem.c:312)
			s += stride; have = false;
		}
	}
	if (B.stats) atomicAdd(&B.ctr->lf_steps, (unsigned long long)steps);
}
