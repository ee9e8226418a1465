// dev_extw.h -- wave-cooperative seed extension: one wavefront (64 lanes) per read.
//
// ksw_extend2 (ksw.c:416-515) opens gaps from the diagonal term M, not from H, so within one DP row
//     F(i,j) = max_{k<j} ( max(M(i,k) - oe_ins, 0) - (j-1-k) * e_ins )
// is a max-plus prefix scan over M(i,.) and the whole row can be computed lane-parallel: lanes own columns,
// rows stay sequential, and the reference's row-by-row band trimming (beg/end), its tie rules and its z-drop
// test are reproduced exactly with ballots and wave reductions.  The per-column state {H(i-1,j-1), E(i,j)} and
// the query profile live in LDS (one private region per wave); columns the band does not touch keep their old
// contents exactly like the reference's never-cleared eh[] array (the "stale cell" rule).
// Since round 4 the rows of short reads whose band fits 127 columns -- nearly all of them -- hold that state in
// registers instead, a lane owning one or two columns of a window that follows the band ("window rows", below),
// and eh[] is only what they are loaded from and written back to when the window moves or a wider row follows.
// The order-dependent control logic of mem_chain2aln (bwamem.c:658-812) runs wave-uniformly; lane 0 does the
// global stores.
#pragma once
#include "dev_ext.h"

#define W_NEG (-0x3fffffff)

DEVFN void wave_sync()
{	// intra-wave ordering point for LDS/global data handed from one lane to another
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	__builtin_amdgcn_wave_barrier();
}
// The lane number as a value the optimizer cannot see through.  Everything a routine derives from threadIdx -- `seeds + lane`, `srt + lane`, LDS slots --
// is loop-invariant for the kernel's outer loops, so LLVM hoists the address arithmetic of every inlined routine into the kernel's prologue and keeps the
// results alive (in this kernel: spilled to scratch, 64-bit values a lane apiece) until the routine's next turn, reads later.  Behind this fence the
// arithmetic stays where it is used: two instructions there instead of a scratch round trip.
DEVFN int opaque_lane() { int l = (int)(threadIdx.x & 63); DEV_KEEP(l); return l; }
// ---- cross-lane primitives on DPP (no LDS round trip) ----------------------------------------------------------
// gfx9-family DPP controls: row_shr:n = 0x110+n (within a row of 16 lanes), row_bcast15 = 0x142 (lane 15 of each row
// to the next row), row_bcast31 = 0x143 (lane 31 to rows 2-3), wave_shr:1 = 0x138 (whole-wave shift by one lane).
// With bound_ctrl = 0 a lane without a valid source keeps `old`.
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_WAVE_SHR1 0x138

DEVFN int imax(int a, int b) { return a > b ? a : b; }
// Sums and differences over columns a lane merely reads along with the live ones (the {H,E} slots past the band's end hold whatever the
// LDS held before): the result is discarded, but it must be a defined one -- two's-complement wrap-around, the instruction the signed
// forms compile to anyway (found by UBSan on the mock runtime in the multi-pass rows: stale slots of INT_MIN; the code object is the same
// instruction for instruction.  The single-pass and two-column rows keep the signed forms: there the unsigned ones change the hot rows'
// code, and no run has met such a slot in them).
DEVFN int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
DEVFN int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
// Inclusive prefix maximum over the 64 lanes; lane 63 ends up with the wave maximum.  `old` is INT_MIN, the identity
// of signed max, which lets the compiler fold each mov_dpp + max pair into a single v_max_i32_dpp.
#define I32_MIN ((int)0x80000000)
DEVFN int wave_incl_scan_max(int v)
{
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(2), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(4), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(8), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_BCAST15, 0xa, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_BCAST31, 0xc, 0xf, false));
	return v;
}
// Inclusive prefix SUM over the 64 lanes, the same six DPP steps (lanes without a source add `old` = 0); lane 63 ends up with the wave's total.
// (The shuffle forms -- __shfl_up / __shfl_xor loops -- are ds_bpermute with a computed address per step: six address registers that LLVM hoists out of
// the kernel's loops and, in k_extend_wave, spills.)
DEVFN int wave_incl_scan_add(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST15, 0xa, 0xf, false);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_BCAST31, 0xc, 0xf, false);
	return v;
}
DEVFN int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_add(v), 63); }
// value of the lane below (lane 0 receives `fill`)
DEVFN int wave_shift_up1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHR1, 0xf, 0xf, false); }
DEVFN int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }   // pin a wave-uniform value into an SGPR
// __ballot() of a comparison costs two extra vector instructions (the predicate is turned into 0/1 and compared again); the w64 builtin hands back the
// compare's own mask
#ifdef __HIP_DEVICE_COMPILE__
DEVFN u64 wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
#else
DEVFN u64 wave_ballot(bool p) { return __ballot(p); }
#endif
DEVFN i64 uni64(i64 v)
{
	int lo = __builtin_amdgcn_readfirstlane((int)(u32)(u64)v), hi = __builtin_amdgcn_readfirstlane((int)(u32)((u64)v >> 32));
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}
DEVFN i64 readlane_i64_(i64 v, int l)
{
	int lo = __builtin_amdgcn_readlane((int)(u32)(u64)v, l), hi = __builtin_amdgcn_readlane((int)(u32)((u64)v >> 32), l);
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}
DEVFN i64 lane0_i64(i64 v)       // broadcast lane 0's value
{
	int lo = __builtin_amdgcn_readlane((int)(u32)(u64)v, 0), hi = __builtin_amdgcn_readlane((int)(u32)((u64)v >> 32), 0);
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}
// Wave-wide work fetch: the wave draws the next item from a global counter and every lane receives it.  Every lane takes
// part in the atomic (lane 0 adds one, the others zero) instead of guarding it with `if (lane == 0)`: with the guard, the
// optimizer unswitched the work loop on the (loop-invariant, but lane-dependent) test, lanes 1..63 got a loop copy of their
// own without the atomic, and a wave whose item took a `continue` spun in it forever.
DEVFN long long wave_fetch(unsigned long long *ctr)
{
	const unsigned long long old = atomicAdd(ctr, (unsigned long long)((threadIdx.x & 63) == 0));
	return (long long)lane0_i64((i64)old);
}
// The same for n consecutive items.
DEVFN long long wave_fetch_n(unsigned long long *ctr, int n)
{
	const unsigned long long old = atomicAdd(ctr, (unsigned long long)((threadIdx.x & 63) == 0 ? n : 0));
	return (long long)lane0_i64((i64)old);
}
// Work queue of one wave over items [0, n) of a heaviest-first list: the first WQ_SINGLE items are drawn one at a time (a wave should
// not sit on several of the heaviest reads), the rest in chunks of WQ_CHUNK, which cuts the atomics on the counter eightfold.
#define WQ_SINGLE 16384
#define WQ_CHUNK 8
struct WaveQueue { int cur, end, step; };      // (item numbers: a batch has fewer than 2^31 reads or regions)
DEVFN void wq_init(WaveQueue &q) { q.cur = q.end = 0; q.step = 1; }
DEVFN bool wq_next(WaveQueue &q, unsigned long long *ctr, long long n, long long &k)
{
	if (q.cur >= q.end) {
		const long long b = wave_fetch_n(ctr, q.step);
		if (b >= n) return false;
		q.cur = uni((int)b); q.end = uni((int)(b + q.step < n ? b + q.step : n));
		if (b >= WQ_SINGLE) q.step = WQ_CHUNK;
	}
	k = q.cur++;
	return true;
}
DEVFN bwagpu_seed_t uni_seed(bwagpu_seed_t s) { s.rbeg = uni64(s.rbeg); s.qbeg = uni(s.qbeg); s.len = uni(s.len); s.score = uni(s.score); return s; }

// LDS of one wave.  Short reads: eh[] holds every query column and qp[] the 5 x qlen query profile.  Long reads (RING): a row of
// ksw_extend2 only touches columns i-w .. i+w+1, columns left of the band are dead and columns right of it still hold their
// first-row values, so eh[] is a ring of ring_mask+1 columns that is initialised lazily as the band advances, and scores come
// from a 25-entry copy of the matrix (mat) and the query bases in global memory.
// (Round 3 also had an optional per-wave LDS copy of a long read and a four-columns-per-lane row form for long reads behind switches; on hardware the copy
// made both long-read DP kernels 14-18 % slower -- its LDS halves the resident waves -- and the row form gained nothing, BENCH_r03 variants: both deleted.)
struct WaveLds { int2 *eh; int8_t *qp; int qstride; int ring_mask; const int8_t *mat; int blk; /* RING: rows of up to 255 columns in one pass, four columns per lane (option ext_blk) */
	u32 *stat; /* stats runs: the wave's work counters in LDS -- [0..1] DP cells, [2] calls, [3] calls answered by the diagonal rule, [4..5] reference bases --; null otherwise.
	              (As u64 registers threaded through the call chain they were ten registers live across every extension of a kernel that spills.) */ };
DEVFN void ext_stat_add(const WaveLds &L, u32 calls, u64 cells, u32 fast, u64 refb)
{
	if (L.stat && (threadIdx.x & 63) == 0) {
		u64 c = (u64)L.stat[1] << 32 | L.stat[0]; c += cells; L.stat[0] = (u32)c; L.stat[1] = (u32)(c >> 32);
		L.stat[2] += calls; L.stat[3] += fast;
		u64 r = (u64)L.stat[5] << 32 | L.stat[4]; r += refb; L.stat[4] = (u32)r; L.stat[5] = (u32)(r >> 32);
	}
}

template <bool RING> __device__ ExtRes wave_ksw_extend2(const DevIndex &ix, const bwagpu_opt_t &opt, int mat_max, const u8 *q, int q0, const int qdir, int qlen,
								   i64 t0, const int tdir, int tlen, int w, int end_bonus, int h0, const WaveLds &L, u64 &cells, u64 &fast)
{
#ifdef BWAGPU_FAKE_DP      // measurement builds only (tools/ext_control_floor.sh): every extension "answered" at once -- what is left of the kernel is mem_chain2aln's control
	{ ExtRes r_; r_.score = h0 + qlen; r_.qle = qlen; r_.tle = qlen < tlen ? qlen : tlen; r_.gtle = r_.tle; r_.gscore = h0 + qlen; r_.max_off = 0; return r_; }
#endif
	const int lane = opaque_lane();
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = opt.zdrop;
	int2 *eh = L.eh; int8_t *qp = L.qp; const int qs = L.qstride;
	// every argument is wave-uniform: keep it in SGPRs so that the row loop's control flow and address arithmetic are scalar
	q0 = uni(q0); qlen = uni(qlen); tlen = uni(tlen); w = uni(w); h0 = uni(h0); end_bonus = uni(end_bonus); t0 = uni64(t0); mat_max = uni(mat_max);
	// ---- the extension that stays on the diagonal needs no DP ------------------------------------------------------------
	// Let P be the score the first qlen diagonal cells lose against an all-match diagonal (mismatches, Ns).  Any cell (i,j) off
	// the diagonal lies on a path with a gap, hence H(i,j) <= h0 + mat_max * (min(i,j)+1) - (o + e*|i-j|), while the diagonal
	// cell of the same row has H(i,i) >= h0 + mat_max * (i+1) - P.  With P < o_del+e_del and P < o_ins+e_ins every diagonal
	// cell is therefore the strict maximum of its row and equals h0 + its prefix sum of scores (E and F, which come out of
	// gaps, cannot reach it; it stays positive, so the band's zero-trimming never cuts it off).  Everything ksw_extend2 returns
	// then follows from the prefix sums V_i: score = max(h0, max V_i) at the first i that attains it (strict update, ksw.c:491),
	// qle = tle = that i + 1, max_off = 0; rows beyond the diagonal's end only hold gapped cells, all below it, so the
	// to-end score is gscore = V_{qlen-1} at gtle = qlen (ksw.c:486-489), and neither m == 0 nor the z-drop test (which
	// would need max - m > zdrop > P) ends the loop before that row.  Needs tlen >= qlen.  At 1 % substitutions this covers
	// most extensions of a read's true locus: a couple of wave steps instead of ~60 rows.
	// score of reference base b against column j's query base: the read's profile (built once per read, indexed in the read's own
	// coordinates whichever way the extension runs), or for long reads the matrix copy and the base itself
	#define QBASE(idx) ((int)q[idx])
	#define SCORE_AT(b, j) (RING ? (int)L.mat[(b) * 5 + QBASE(q0 + (j) * qdir)] : (int)qp[(b) * qs + q0 + (j) * qdir])
	if (tlen >= qlen && qlen > 0) {
		const int oe_min = oe_del < oe_ins ? oe_del : oe_ins;
		int P = 0;
		for (int b = 0; b < qlen && P < oe_min; b += 64) {
			const int j = b + lane;
			int loss = 0;
			if (j < qlen) loss = mat_max - SCORE_AT(ref_base(ix, t0 + (i64)j * tdir), j);
			P += wave_sum(loss);
		}
		if (P < oe_min && (zdrop <= 0 || P < zdrop) && h0 > P) {
			int best = h0, best_i = -1, run = h0;
			for (int b = 0; b < qlen; b += 64) {
				const int j = b + lane;
				int sc = 0;
				if (j < qlen) sc = SCORE_AT(ref_base(ix, t0 + (i64)j * tdir), j);
				const int inc = wave_incl_scan_add(sc);        // inclusive prefix sum over the wave
				const int v = run + inc;
				// first lane of this chunk whose value exceeds everything before it: maximum of (v << 6 | 63 - lane)
				const int key = wave_incl_scan_max(j < qlen ? (v << 6 | (63 - lane)) : I32_MIN);
				const int kmax = __builtin_amdgcn_readlane(key, 63);
				if ((kmax >> 6) > best) { best = kmax >> 6; best_i = b + 63 - (kmax & 63); }
				const int nact = qlen - b < 64 ? qlen - b : 64;
				run = __builtin_amdgcn_readlane(v, nact - 1);
			}
			ExtRes r; r.score = best; r.qle = best_i + 1; r.tle = best_i + 1; r.gtle = qlen; r.gscore = run; r.max_off = 0;
			++fast;
			return r;
		}
	}
	// first row (ksw.c:430-433: H(-1,-1) = h0, then an insertion ramp); the query profile (ksw.c:425-428) is the read's, see ext_read_wave
	#define EHI(j) (RING ? ((j) & L.ring_mask) : (j))
	const int v1 = h0 > oe_ins ? h0 - oe_ins : 0;
	if (!RING) {
		for (int j = lane; j <= qlen; j += 64) {
			int hv = j == 0 ? h0 : v1 - (j - 1) * e_ins;
			eh[j] = make_int2(hv > 0 ? hv : 0, 0);
		}
		wave_sync();
	}
	int init_hi = -1;                                  // RING: columns 0..init_hi hold valid (initial or computed) values
	int q_pre = 4, q_pre_beg = -1;                     // RING: the query bases of columns q_pre_beg + lane, asked for a row ahead (see the multi-pass rows)
	int qb_bb = -8; u32 qb_w = 0, qb_n = 0;            // RING, four columns per lane: the block the packed bases qb_w belong to, and the next block's
	int lim = trunc_div_add(qlen * mat_max + end_bonus - o_ins, e_ins, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;
	lim = trunc_div_add(qlen * mat_max + end_bonus - o_del, e_del, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;
	w = uni(w);
	u32 cells32 = 0;
	int beg = 0, end = qlen, max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, treg = 0;
	const bool two_col_ok = h0 + qlen * mat_max < (1 << 23);      // (score << 7 | column) must fit the scan's 31 bits
	const int lane_e = lane * e_ins;
	// The single-pass rows ("window rows", the form 150 bp reads spend most of their rows in).  The kernel is bound by instruction issue, the
	// scalar unit first (round 3: 85 scalar and 62 vector instructions per row, one scalar unit per CU against four vector units), so the form is
	// written for few of both:
	//  * lanes keep their columns.  While the band fits the 64 columns [base, base + 64) lane L owns column base + L and keeps its {H, E} slot
	//    in two registers from row to row -- the slot is read and written by that lane only (H(i,j) reaches column j + 1 by a lane shift) -- so a
	//    row touches LDS once, for its score.  Lanes outside the band leave their registers alone, which IS the reference's stale-cell rule;
	//    when the band leaves the window (it drifts right by about a column a row) the registers go back to eh[] and the window is re-based.
	//  * the band is a pair of lane numbers [lo, hi) and the trimming (ksw.c:502-505) two bit scans of ONE ballot: bit L = column base + L holds
	//    a non-zero {h, e}, the column `end` (lane hi, which stores {H(i,end-1), 0}, ksw.c:485) included.
	//  * F's offsets j * e_ins shrink to lane * e_ins: the row's common term cancels between the scan's input and output.  Lanes below the band
	//    feed the scan its identity, so the first column's F is far below zero instead of the reference's 0 -- H = max(M, E, F) is the same
	//    number, because E >= 0 everywhere.  Lanes outside the band shift out H = 0, the reference's h1 for beg > 0.
	//  * the row maximum is taken over max(M, E) instead of H = max(M, E, F): F(i,j) <= max(0, max_k M(i,k) - oe_ins) lies strictly below a
	//    positive row maximum, so neither the maximum nor the "last column wins" choice (ksw.c:473-474) changes, and the two six-step DPP
	//    chains interleave instead of idling between dependent steps.  The scan's key carries the ABSOLUTE column (score << 11 | column).
	//  * deferred bookkeeping.  What a row contributes to the running maximum (ksw.c:491-493), to the z-drop test (:494-500) and to the to-end
	//    score (:486-489) are two numbers -- the key of its maximum, and H(i, end-1) when the band touches the query's end -- and neither
	//    steers the next row (the band's trimming does, and stays in the row).  A row parks them in lane `hist_n` of two vector registers,
	//    and every 32 rows -- or when the extension ends, or before a row of another form -- the parked rows are evaluated
	//    together, one lane per row: prefix maxima give every row the (max, max_i, max_j) it would have seen, the first z-drop hit ends the
	//    extension there (the rows computed past it are simply not counted; the next call re-initialises the columns they wrote), and the
	//    survivors update the state.
	const bool win_ok = !RING && qlen < 2048 && h0 + qlen * mat_max < (1 << 20);      // (score << 11 | column) must fit the scan's 31 bits
	int hist_k = 0, hist_h1 = -1, hist_n = 0, hist_row0 = 0;
	auto hist_flush = [&]() -> bool {       // returns true when a parked row ended the extension by z-drop
		const int cnt = hist_n;
		if (cnt == 0) return false;
		const bool val = lane < cnt;
		const int r = hist_row0 + lane;
		const int hm = hist_k >> 11, hj = hist_k & 2047;                                                  // this lane's row: its maximum and that maximum's column
		const int pm = imax(wave_shift_up1(wave_incl_scan_max(val ? hm : I32_MIN), I32_MIN), max);     // the maximum before row r
		const bool imp = val && hm > pm;                                                               // row r raises it
		const int lk = wave_shift_up1(wave_incl_scan_max(imp ? ((lane + 1) << 16 | hj) : 0), 0);      // the latest raising row before r (0: none among the parked ones)
		bool zb = false;
		if (zdrop > 0 && val && !imp) {
			const int pmi = lk ? hist_row0 + (lk >> 16) - 1 : max_i, pmj = lk ? (lk & 0xffff) : max_j;
			const int di = r - pmi, dj = hj - pmj;
			zb = di > dj ? pm - hm - (di - dj) * e_del > zdrop : pm - hm - (dj - di) * e_ins > zdrop;
		}
		const u64 zm = __ballot(zb);
		const int last = zm ? __builtin_ctzll(zm) : cnt - 1;                 // the last row that counts
		const bool use = lane <= last;
		const u64 im = wave_ballot(imp) & wave_ballot(use);
		if (im) {
			const int l = 63 - __builtin_clzll(im);                          // the raises are strictly increasing: the last one holds the maximum
			max = __builtin_amdgcn_readlane(hm, l); max_i = hist_row0 + l; max_j = __builtin_amdgcn_readlane(hj, l);
			int off = hj - r; off = off < 0 ? -off : off;
			const int mo = __builtin_amdgcn_readlane(wave_incl_scan_max(imp && use ? off : 0), 63);
			max_off = mo > max_off ? mo : max_off;
		}
		const int g = __builtin_amdgcn_readlane(wave_incl_scan_max(use && hist_h1 >= 0 ? (hist_h1 << 6 | lane) : -1), 63);   // the best to-end score, latest row on ties
		if (g >= 0 && (g >> 6) >= gscore) { max_ie = hist_row0 + (g & 63); gscore = g >> 6; }
		hist_row0 += cnt; hist_n = 0; hist_h1 = -1;
		return zm != 0;
	};
	// ---- rows that cannot change the result are not computed (round 6) ------------------------------------------------------------------------
	// ksw_extend2 stops at m == 0, at a z-drop or at the target's end; after the best alignment has run into the query's end that is tlen - qlen ~ qlen
	// further rows of decaying deletion tails -- 42 % of all DP rows on 2x150 bp reads (counted with an instrumented reference).  None of them changes what
	// the call returns once the following holds.  Give a state cell -- eh[j].h = H(i,j-1), the diagonal source of column j, or eh[j].e = E(i+1,j) -- the
	// potential  P = value + mat_max * (columns still to its right),  qlen - j for the h slot, qlen - 1 - j for the e slot.  Every value a later row can
	// hold is reached from a live state cell by diagonal steps (one column on, at most + mat_max: P does not grow), deletions (same column, value falls)
	// and insertions (columns on, value falls), so it is at most B = max P over the band's live cells (H(i,-1) is eh[0].h and is in the band while it
	// matters; zero cells are dead, ksw.c:469; with gscore >= 0 the band has touched the query's end, so no column still holds an unvisited first-row
	// value).  Later rows then have m <= B and h1 <= B:  B <= max  means no row raises max / max_i / max_j / max_off (ksw.c:491: strict), and
	// B < gscore  means none updates gscore / max_ie (ksw.c:486-489: h1 >= gscore).  The remaining exits (m == 0, z-drop) return the same values.
	// The test needs max and gscore, which the window rows only bring up to date every 32 rows (parked bookkeeping).  Two scalars stand in: m_run >= the
	// largest row maximum and g_run >= the largest to-end score of ALL rows computed so far.  Without a z-drop among the parked rows they equal max and gscore;
	// with one, the extension has already ended at that row and whatever is skipped after it was never part of the result (the flush after the loop cuts the
	// parked rows there as always).  Tested every second row from eight rows before the query's end on, and every 32 rows before; the DP fuzz holds every
	// returned field to the reference's.
	const int tail_from = qlen - 8;
	int m_run = h0, g_run = -1;
	auto tail_done = [&](int cand) -> bool {
		const int B = __builtin_amdgcn_readlane(wave_incl_scan_max(cand), 63);
		return g_run >= 0 && B <= m_run && B < g_run;
	};
	int i = 0;
	while (i < tlen) {
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		beg = uni(beg); end = uni(end);
		bool stop = false;
		if (win_ok && end > beg && end - beg <= 63) {
			const int base = beg, jcol = base + lane;
			int sth, ste; { const int2 t = eh[jcol]; sth = t.x; ste = t.y; }      // this lane's {H, E} slot (the LDS region is padded by 64 columns)
			const int8_t *qcol = qp + q0 + jcol * qdir;       // column jcol's score is qcol[tb * qs] (lanes past the band read the profile's padding or its neighbourhood, never past the wave's LDS)
			const int e_lane = lane == 0 ? W_NEG : e_ins - lane_e;           // (lane 0 has no column to its left: the shift hands it 0, this makes its F the scan's identity)
			const int qhi = qlen - base;
			// H(i,-1) = max(h0 - (o_del + e_del * (i + 1)), 0) (ksw.c:452-455), the value column 0's left neighbour hands over: lane 0 of a window that starts at
			// column 0 counts it down; every other lane, and every lane of a window further right, shifts in H = 0
			int hdl = lane == 0 && base == 0 ? h0 - (o_del + e_del * (i + 1)) : W_NEG;
			const int edel0 = lane == 0 ? e_del : 0;
			int lo = 0, hi = end - base, lo_min = i - w - base, hi_max = i + w + 1 - base;
			hi = uni(hi);                                     // (without the pin the compiler carries the row loop's control in vector registers and branches on exec masks)
			int why = 0;                                      // 1: the extension ends (m == 0 or z-drop), 2: the band has left the window
			do {
				// a segment: rows up to the next multiple of 64 (the reference bases in treg), the 32nd parked row, or the last row
				if ((i & 63) == 0) { const int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
				int i_end = (i | 63) + 1; if (i_end > tlen) i_end = tlen; if (i_end > i + 32 - hist_n) i_end = i + 32 - hist_n;
				if (i >= tail_from && i_end > i + 2) i_end = i + 2;
				i_end = uni(i_end);
				int sc_next = qcol[__builtin_amdgcn_readlane(treg, i & 63) * qs];
				int key;
				for (;;) {
					const int sc = sc_next;
					sc_next = qcol[__builtin_amdgcn_readlane(treg, (i + 1) & 63) * qs];   // the next row's score, in flight while this row computes (past the segment's end: read and dropped)
					const int nact = hi - lo;
					const unsigned rel = (unsigned)(lane - lo);
					const bool act = rel < (unsigned)nact, wr = rel <= (unsigned)nact;
					const int M = sth ? sth + sc : 0;          // ksw.c:469: a dead diagonal cell stays dead
					const int hme = imax(M, ste);
					const int kmax = wave_incl_scan_max(act ? (hme << 11 | jcol) : -1);
					const int inc = wave_incl_scan_max(act ? imax(M - oe_ins, 0) + lane_e : W_NEG);
					const int exc = __builtin_amdgcn_update_dpp(0, inc, DPP_WAVE_SHR1, 0xf, 0xf, true);     // best insertion start left of this column
					const int h = act ? imax(hme, exc + e_lane) : 0;             // H(i,j) = max(M, E, F), ksw.c:470-471
					const int e_new = act ? imax(imax(ste - e_del, M - oe_del), 0) : 0;   // E(i+1,j), ksw.c:475-479
					const int hleft = imax(__builtin_amdgcn_update_dpp(0, h, DPP_WAVE_SHR1, 0xf, 0xf, true), hdl);   // eh[j].h after this row = H(i,j-1)
					hdl -= edel0;
					sth = wr ? hleft : sth; ste = wr ? e_new : ste;
					const u64 nz = wave_ballot((hleft | e_new) != 0) & wave_ballot(wr);      // (two compare masks and a scalar AND: the ballot of a conjunction is rebuilt from 0/1 values)
					key = __builtin_amdgcn_readlane(kmax, 63);
					hist_k = lane == hist_n ? key : hist_k;
					m_run = imax(m_run, key >> 11);
					if (hi == qhi) {                                  // H(i, end-1) feeds the to-end score (ksw.c:486-489); rows that do not touch the query's end leave the -1 of the last flush
						const int h1 = __builtin_amdgcn_readlane(hleft, hi);
						hist_h1 = lane == hist_n ? h1 : hist_h1;
						g_run = imax(g_run, h1);
					}
					++hist_n; ++i; cells32 += (u32)nact;
					if (key < 2048) break;                           // m == 0 (ksw.c:490)
					// band for the next row (ksw.c:502-505): skip leading / trailing columns whose {h,e} are both zero (m > 0, so nz != 0), then its clamps
					lo = __builtin_ctzll(nz | 1ull << hi);
					hi = 65 - __builtin_clzll(nz); if (hi > qhi) hi = qhi;
					++lo_min; ++hi_max;
					if (lo < lo_min) lo = lo_min;
					if (hi > hi_max) hi = hi_max;
					if (hi > 63) break;
					if (hi <= lo) break;
					if (i == i_end) break;
				}
				if (key < 2048) why = 1;
				else {
					if (hi > 63 || hi <= lo) why = 2;
					if (hist_n == 32 && hist_flush()) why = 1;
					if (why == 0 && g_run >= 0 && (hist_n == 0 || i >= tail_from)) {
						{
							const bool inb = (unsigned)(lane - lo) <= (unsigned)(hi - lo);       // the band's columns and column `end`
							const int pot = (qlen - jcol) * mat_max;
							if (tail_done(inb ? imax(sth > 0 ? sth + pot : 0, ste > 0 ? ste + pot - mat_max : 0) : 0)) why = 1;
						}
					}
				}
			} while (why == 0 && i < tlen);
			if (why == 1) break;
			eh[jcol] = make_int2(sth, ste);
			wave_sync();
			beg = base + lo; end = base + hi;
			continue;
		}
		if (win_ok && end - beg > 63 && end - beg <= 127) {
			// The same with two adjacent columns per lane (64..127 columns: the longer half of a 150 bp read's extensions): lane L owns columns base + 2L ("A") and
			// base + 2L + 1 ("B"), so a row still takes ONE prefix scan for F and one for the row maximum.  H(i, A) reaches column B inside the lane, H(i, B) reaches
			// the next lane's A by the shift.  The band's bits come as two ballots (even and odd columns); the trimming reads them lane-wise.
			const int base = beg, jA = base + 2 * lane, jB = jA + 1;
			int shA = 0, seA = 0, shB = 0, seB = 0;          // the lane's two {H, E} slots (columns past the query's end: never in a band, neither loaded nor stored)
			if (jA <= qlen) { const int2 t = eh[jA]; shA = t.x; seA = t.y; }
			if (jB <= qlen) { const int2 t = eh[jB]; shB = t.x; seB = t.y; }
			const int8_t *qcA = qp + q0 + (jA < qlen ? jA : qlen - 1) * qdir, *qcB = qp + q0 + (jB < qlen ? jB : qlen - 1) * qdir;
			const int lane_e2 = 2 * lane_e;
			const int e_laneA = lane == 0 ? W_NEG : e_ins - lane_e2;
			const int qhi = qlen - base;
			int hdl = lane == 0 && base == 0 ? h0 - (o_del + e_del * (i + 1)) : W_NEG;
			const int edel0 = lane == 0 ? e_del : 0;
			int lo = 0, hi = end - base, lo_min = i - w - base, hi_max = i + w + 1 - base;
			hi = uni(hi);
			int why = 0;
			do {
				if ((i & 63) == 0) { const int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
				int i_end = (i | 63) + 1; if (i_end > tlen) i_end = tlen; if (i_end > i + 32 - hist_n) i_end = i + 32 - hist_n;
				if (i >= tail_from && i_end > i + 2) i_end = i + 2;
				i_end = uni(i_end);
				int scA_next, scB_next; { const int o = __builtin_amdgcn_readlane(treg, i & 63) * qs; scA_next = qcA[o]; scB_next = qcB[o]; }
				int key;
				for (;;) {
					const int scA = scA_next, scB = scB_next;
					{ const int o = __builtin_amdgcn_readlane(treg, (i + 1) & 63) * qs; scA_next = qcA[o]; scB_next = qcB[o]; }
					const int nact = hi - lo;
					const unsigned relA = (unsigned)(2 * lane - lo), relB = relA + 1;
					const bool actA = relA < (unsigned)nact, wrA = relA <= (unsigned)nact, actB = relB < (unsigned)nact, wrB = relB <= (unsigned)nact;
					const int MA = shA ? shA + scA : 0, MB = shB ? shB + scB : 0;          // ksw.c:469
					const int hmA = imax(MA, seA), hmB = imax(MB, seB);
					const int kmax = wave_incl_scan_max(imax(actA ? (hmA << 11 | jA) : -1, actB ? (hmB << 11 | jB) : -1));   // last column wins ties (ksw.c:473-474)
					const int aA = actA ? imax(MA - oe_ins, 0) + lane_e2 : W_NEG, aB = actB ? imax(MB - oe_ins, 0) + lane_e2 + e_ins : W_NEG;
					const int exc = __builtin_amdgcn_update_dpp(0, wave_incl_scan_max(imax(aA, aB)), DPP_WAVE_SHR1, 0xf, 0xf, true);   // best insertion start among the columns of the lanes below
					const int hA = actA ? imax(hmA, exc + e_laneA) : 0;                         // ksw.c:470-471
					const int hB = actB ? imax(hmB, imax(exc, aA) - lane_e2) : 0;
					const int eA = actA ? imax(imax(seA - e_del, MA - oe_del), 0) : 0, eB = actB ? imax(imax(seB - e_del, MB - oe_del), 0) : 0;   // ksw.c:475-479
					const int hleftA = imax(__builtin_amdgcn_update_dpp(0, hB, DPP_WAVE_SHR1, 0xf, 0xf, true), hdl);   // H(i, A - 1): the lane below's column B
					hdl -= edel0;
					shA = wrA ? hleftA : shA; seA = wrA ? eA : seA;
					shB = wrB ? hA : shB; seB = wrB ? eB : seB;
					const u64 mwA = wave_ballot(wrA), mwB = wave_ballot(wrB);
					const u64 nzA = wave_ballot((hleftA | eA) != 0) & mwA, nzB = wave_ballot((hA | eB) != 0) & mwB;
					// column `end` (in one of the two write masks, in neither set of live columns) stands in for "none" in the search from the left
					const u64 xA = nzA | (mwA & ~wave_ballot(actA)), xB = nzB | (mwB & ~wave_ballot(actB));
					key = __builtin_amdgcn_readlane(kmax, 63);
					hist_k = lane == hist_n ? key : hist_k;
					m_run = imax(m_run, key >> 11);
					if (hi == qhi) {                                  // H(i, end-1), what column `end` has just stored (ksw.c:485-489)
						const int h1 = (hi & 1) ? __builtin_amdgcn_readlane(hA, hi >> 1) : __builtin_amdgcn_readlane(hleftA, hi >> 1);
						hist_h1 = lane == hist_n ? h1 : hist_h1;
						g_run = imax(g_run, h1);
					}
					++hist_n; ++i; cells32 += (u32)nact;
					if (key < 2048) break;                           // m == 0 (ksw.c:490)
					// band for the next row (ksw.c:502-505), lane-wise: the first lane with a non-zero column (or column `end`) and whether that is its A, the last
					// lane with a non-zero column and whether that is its B (m > 0, so there is one)
					{
						const int lf = __builtin_ctzll(xA | xB), ll = 63 - __builtin_clzll(nzA | nzB);
						lo = 2 * lf + 1 - (int)(xA >> lf & 1);
						hi = 2 * ll + (int)(nzB >> ll & 1) + 2; if (hi > qhi) hi = qhi;
					}
					++lo_min; ++hi_max;
					if (lo < lo_min) lo = lo_min;
					if (hi > hi_max) hi = hi_max;
					if (hi > 127) break;
					if (hi - lo <= 63) break;                        // (narrow enough for a column per lane, or empty)
					if (i == i_end) break;
				}
				if (key < 2048) why = 1;
				else {
					if (hi > 127 || hi - lo <= 63) why = 2;
					if (hist_n == 32 && hist_flush()) why = 1;
					if (why == 0 && g_run >= 0 && (hist_n == 0 || i >= tail_from)) {
						{
							const bool inA = (unsigned)(2 * lane - lo) <= (unsigned)(hi - lo), inB = (unsigned)(2 * lane + 1 - lo) <= (unsigned)(hi - lo);
							const int potA = (qlen - jA) * mat_max, potB = potA - mat_max;
							const int cA = inA ? imax(shA > 0 ? shA + potA : 0, seA > 0 ? seA + potA - mat_max : 0) : 0;
							const int cB = inB ? imax(shB > 0 ? shB + potB : 0, seB > 0 ? seB + potB - mat_max : 0) : 0;
							if (tail_done(imax(cA, cB))) why = 1;
						}
					}
				}
			} while (why == 0 && i < tlen);
			if (why == 1) break;
			if (jA <= qlen) eh[jA] = make_int2(shA, seA);
			if (jB <= qlen) eh[jB] = make_int2(shB, seB);
			wave_sync();
			beg = base + lo; end = base + hi;
			continue;
		}
		if ((i & 63) == 0) { int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		const int8_t *qrow = qp + tb * qs + q0;        // column j's score is qrow[j * qdir]
		if (RING) {      // first-row values (ksw.c:430-433) for the columns this row can reach for the first time
			int hi = i + w + 2 < qlen ? i + w + 2 : qlen;
			if (hi > init_hi) {
				for (int j = init_hi + 1 + lane; j <= hi; j += 64) {
					int hv = j == 0 ? h0 : v1 - (j - 1) * e_ins;
					eh[EHI(j)] = make_int2(hv > 0 ? hv : 0, 0);
				}
				init_hi = hi;
				wave_sync();
			}
		}
		int h1_init = 0;
		if (beg == 0) { h1_init = h0 - (o_del + e_del * (i + 1)); if (h1_init < 0) h1_init = 0; }
		cells32 += (u32)(end > beg ? end - beg : 0);
		{
		if (hist_flush()) break;                         // (rows of the other forms keep their bookkeeping per row: bring the state up to date first)
		hist_row0 = i + 1;
		int m = 0, mj = -1, carry = W_NEG, hprev = h1_init, first_nz = -1, last_nz = -1, bnd = 0;
		if (!RING && end - beg <= 127 && two_col_ok) {     // (127: column `end` needs a slot of its own, like the spare lane of the single-pass form)
			// 65..127 columns (the longer half of a 150 bp read's extensions): each lane owns two adjacent columns, so the row still takes ONE
			// prefix scan for F and one for the row maximum instead of two passes of the loop below with their carries.  A lane's two
			// {H,E} slots are read and written by that lane only -- H(i,j) reaches the owner of column j+1 through a lane shift, not LDS.
			const int nact = end - beg;
			const int jA = beg + 2 * lane, jB = jA + 1;
			const bool actA = jA < end, actB = jB < end;
			const int2 oA = eh[jA], oB = eh[jB];               // (the LDS region is padded by 64 columns past the longest read)
			const int scA = qrow[(actA ? jA : beg) * qdir], scB = qrow[(actB ? jB : beg) * qdir];
			const int MA = oA.x ? oA.x + scA : 0, MB = oB.x ? oB.x + scB : 0;      // ksw.c:469
			const int aA = actA ? imax(MA - oe_ins, 0) + 2 * lane_e : W_NEG;           // (F's offsets without the row's common term beg * e_ins, as above)
			const int aB = actB ? imax(MB - oe_ins, 0) + 2 * lane_e + e_ins : W_NEG;
			const int hmA = imax(MA, oA.y), hmB = imax(MB, oB.y);
			const int kA = actA ? (hmA << 7 | 2 * lane) : -1, kB = actB ? (hmB << 7 | (2 * lane + 1)) : -1;   // last column wins ties (ksw.c:473-474); F never holds the row maximum
			const int kmax = wave_incl_scan_max(imax(kA, kB));
			const int exc = wave_shift_up1(wave_incl_scan_max(imax(aA, aB)), W_NEG);   // best insertion start among the columns of the lanes below
			const int fA = lane == 0 ? 0 : exc - 2 * lane_e + e_ins;
			const int fB = imax(exc, aA) - 2 * lane_e;
			const int hA = imax(hmA, fA), hB = imax(hmB, fB);                          // ksw.c:470-471
			const int eA = imax(imax(oA.y - e_del, MA - oe_del), 0), eB = imax(imax(oB.y - e_del, MB - oe_del), 0);   // ksw.c:475-479
			const int hleftA = wave_shift_up1(hB, h1_init);                       // H(i, jA-1): the lane below's second column
			if (jA <= end) eh[jA] = make_int2(hleftA, actA ? eA : 0);                 // (column `end` gets {h1, 0}, ksw.c:485)
			if (jB <= end) eh[jB] = make_int2(hA, actB ? eB : 0);
			const u64 nzA = wave_ballot((hleftA | eA) != 0) & wave_ballot(actA), nzB = wave_ballot((hA | eB) != 0) & wave_ballot(actB);
			if (nzA | nzB) {
				const int fa = nzA ? 2 * __builtin_ctzll(nzA) : 1 << 20, fb = nzB ? 2 * __builtin_ctzll(nzB) + 1 : 1 << 20;
				const int la = nzA ? 2 * (63 - __builtin_clzll(nzA)) : -1, lb = nzB ? 2 * (63 - __builtin_clzll(nzB)) + 1 : -1;
				first_nz = beg + (fa < fb ? fa : fb); last_nz = beg + (la > lb ? la : lb);
			}
			const int key = __builtin_amdgcn_readlane(kmax, 63);
			if (key >= 0) { m = key >> 7; mj = beg + (key & 127); }
			const int lastc = nact - 1;
			hprev = (lastc & 1) ? __builtin_amdgcn_readlane(hB, lastc >> 1) : __builtin_amdgcn_readlane(hA, lastc >> 1);
			wave_sync();
		} else {
			// RING: a lane's query base comes from the batch's array in HBM -- a dependent global load (an L1/L2 hit: 200-500 cycles) in front of every
			// pass of every row, at one wave per SIMD where nothing hides it (measured: ~100 ms of wave time per 10 kb read, profiles/r04_longread_wave_time_per_read.log).
			// The addresses are known long before: the base of pass b + 64 is asked for when pass b starts, the first pass's base when the ROW BEFORE
			// starts (for the usual case that the band moves on by one column; anything else loads it here).  (Measured on 6000 x 10 kb reads: the stage's
			// time did not move, 361-411 ms either way -- the row's dependent DPP/LDS chain at one wave per SIMD is the longer wait; kept because it is never slower.)
			int qc_next = 4;
			if (RING && L.blk && beg < end && end - (beg & ~3) <= 255) {
				// Four adjacent columns per lane: the whole band of a long read's row (2 w + 1 <= 255 columns with the presets' w = 100) in ONE pass -- one pair of
				// prefix scans, one round of ballots and lane reads per row instead of one per 64 columns.  The scalar pipe is the busy unit of this kernel
				// (profiles/r05_longread_experiments.md): the pass loop below spends ~30 scalar instructions per pass, three to four passes per row.  A lane
				// reads and writes its own four ring slots only (two 16-byte LDS accesses each way); H(i,j) reaches column j+1 inside the lane or, for a lane's
				// first column, by a lane shift.  Slots outside [beg, end] are written back as they were read: the stale-cell rule.  The lane's four query bases are
				// kept packed in a register while the band's first block stays where it is (beg & ~3 moves every ~4 rows; the next block's bases are
				// loaded one move ahead).
				const int bb = beg & ~3, j0 = bb + 4 * lane, pr = j0 & L.ring_mask;
				auto qload = [&](int jb) { u32 v = 0; for (int c = 0; c < 4; ++c) { const int j = jb + c; v |= (u32)(j < qlen ? QBASE(q0 + j * qdir) : 4) << (8 * c); } return v; };
				if (bb != qb_bb) {
					if (bb == qb_bb + 4 && qb_bb >= 0) qb_w = qb_n; else qb_w = qload(j0);
					qb_n = qload(j0 + 4); qb_bb = bb;
				}
				const int4 a0 = *(const int4*)&eh[pr], a1 = *(const int4*)&eh[pr + 2];
				const int ox[4] = { a0.x, a0.z, a1.x, a1.z }, oy[4] = { a0.y, a0.w, a1.y, a1.w };
				const int8_t *mrow = L.mat + tb * 5;
				int M4[4], pre[4], run = W_NEG;
				#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int j = j0 + c; const bool act = j >= beg && j < end;
					const int sc = mrow[(qb_w >> (8 * c)) & 255u];
					M4[c] = ox[c] ? wadd(ox[c], sc) : 0;                       // ksw.c:469
					pre[c] = run;
					run = imax(run, act ? imax(wsub(M4[c], oe_ins), 0) + j * e_ins : W_NEG);
				}
				const int inc = wave_incl_scan_max(run);
				const int exl = wave_shift_up1(inc, W_NEG);                     // best insertion start among the columns of the lanes below
				int hv[4], en[4], key = -1; u32 nzb = 0;
				#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int j = j0 + c; const bool act = j >= beg && j < end;
					const int f = j == beg ? 0 : imax(exl, pre[c]) - (j - 1) * e_ins;
					hv[c] = imax(imax(M4[c], oy[c]), f);                       // ksw.c:470-471 (used for active columns only)
					en[c] = imax(imax(wsub(oy[c], e_del), wsub(M4[c], oe_del)), 0);   // ksw.c:475-479
					key = imax(key, act ? (hv[c] << 8 | (4 * lane + c)) : -1);  // last column wins ties (ksw.c:473-474)
				}
				const int hv3 = hv[3];
				const int hl = wave_shift_up1(hv3, h1_init);                   // H(i, j0 - 1): the lane below's last column
				int nx[4], ny[4];
				#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int j = j0 + c; const bool act = j >= beg && j < end;
					const int hleft = c ? hv[c - 1] : hl;
					nx[c] = j == beg ? h1_init : (j > beg && j <= end ? hleft : ox[c]);
					ny[c] = act ? en[c] : (j == end ? 0 : oy[c]);
					if (act && (nx[c] | ny[c]) != 0) nzb |= 1u << c;
				}
				*(int4*)&eh[pr] = make_int4(nx[0], ny[0], nx[1], ny[1]);
				*(int4*)&eh[pr + 2] = make_int4(nx[2], ny[2], nx[3], ny[3]);
				const u64 any = wave_ballot(nzb != 0);
				if (any) {
					const int fl = __builtin_ctzll(any), ll = 63 - __builtin_clzll(any);
					const int lo_c = __builtin_ctz(nzb | 16u), hi_c = 31 - __builtin_clz(nzb | 1u);       // the lane's first / last non-zero slot (meaningful where nzb != 0)
					first_nz = bb + 4 * fl + __builtin_amdgcn_readlane(lo_c, fl);
					last_nz = bb + 4 * ll + __builtin_amdgcn_readlane(hi_c, ll);
				}
				const int kk = __builtin_amdgcn_readlane(wave_incl_scan_max(key), 63);
				if (kk >= 0) { m = kk >> 8; mj = bb + (kk & 255); }
				{	// H(i, end-1): what the reference's column loop leaves in h1
					const int le = end - 1 - bb, se = le & 3;
					const int hs = se == 0 ? hv[0] : (se == 1 ? hv[1] : (se == 2 ? hv[2] : hv[3]));
					hprev = __builtin_amdgcn_readlane(hs, le >> 2);
				}
				wave_sync();
			} else {
			if (RING) {
				const int j0 = beg + lane;
				qc_next = (beg == q_pre_beg) ? q_pre : (j0 < qlen ? QBASE(q0 + j0 * qdir) : 4);
				const int jn = beg + 1 + lane;                      // (the next row's first pass, if its band starts one column on)
				q_pre = jn < qlen ? QBASE(q0 + jn * qdir) : 4; q_pre_beg = beg + 1;
			}
			for (int b = beg; b < end; b += 64) {
				const int j = b + lane; const bool act = j < end;
				// the LDS region is padded by 64 columns, so inactive lanes may read (never write) past `end`
				int2 old = eh[EHI(j)];
				int sc;
				if (RING) {
					const int qc = qc_next;
					const int j2 = j + 64;                          // (the next pass's base: in flight while this pass computes)
					qc_next = (b + 64 < end && j2 < qlen) ? QBASE(q0 + j2 * qdir) : 4;
					sc = L.mat[tb * 5 + qc];
				}
				else sc = qrow[(act ? j : beg) * qdir];
				const int bnd_next = eh[EHI(b + 64)].x;             // next pass's diagonal for its lane 0, before lane 63 overwrites it
				if (b != beg && lane == 0) old.x = bnd;
				wave_sync();
				const int M = old.x ? wadd(old.x, sc) : 0;     // ksw.c:469: a dead diagonal cell stays dead
				const int a = act ? imax(M - oe_ins, 0) + j * e_ins : W_NEG;
				const int inc = wave_incl_scan_max(a);
				const int exc = imax(wave_shift_up1(inc, W_NEG), carry);
				const int f = j == beg ? 0 : exc - (j - 1) * e_ins;       // F(i,j): best insertion ending left of column j
				const int h = imax(imax(M, old.y), f);                    // H(i,j) = max(M, E, F), ksw.c:470-471
				const int e_new = imax(imax(wsub(old.y, e_del), wsub(M, oe_del)), 0); // E(i+1,j), ksw.c:475-479
				if (act) {
					eh[EHI(j)].y = e_new;
					eh[EHI(j + 1)].x = h;                                  // becomes the diagonal of column j+1 in row i+1
				}
				if (j == beg) eh[EHI(j)].x = h1_init;
				const int hleft = wave_shift_up1(h, hprev);               // eh[j].h after this row = H(i,j-1)
				const u64 nzm = wave_ballot((hleft | e_new) != 0) & wave_ballot(act);
				if (nzm) { if (first_nz < 0) first_nz = b + __ffsll((unsigned long long)nzm) - 1; last_nz = b + 63 - __clzll((long long)nzm); }
				// row maximum with "last column wins ties" (ksw.c:473-474): one scan over (h << 6 | lane)
				const int key = __builtin_amdgcn_readlane(wave_incl_scan_max(act ? (h << 6 | lane) : -1), 63);
				if ((key >> 6) >= m) { m = key >> 6; mj = b + (key & 63); }
				carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
				const int nact = end - b < 64 ? end - b : 64;
				hprev = __builtin_amdgcn_readlane(h, nact - 1);
				bnd = bnd_next;
				wave_sync();
			}
			if (lane == 0) { eh[EHI(end)].x = beg < end ? hprev : h1_init; eh[EHI(end)].y = 0; }
			wave_sync();
			}
		}
		const int h1 = beg < end ? hprev : h1_init;      // H(i, end-1) as left in h1 by the reference's column loop
		const int jfin = beg < end ? end : beg;
		if (jfin == qlen) { if (h1 >= gscore) max_ie = i; if (h1 > gscore) gscore = h1; g_run = imax(g_run, h1); }
		m_run = imax(m_run, m);
		stop = m == 0;
		if (m > max) {
			int off = mj - i; if (off < 0) off = -off;
			max = m; max_i = i; max_j = mj;
			if (off > max_off) max_off = off;
		} else if (zdrop > 0 && !stop) {
			const int di = i - max_i, dj = mj - max_j;
			if (di > dj) stop = max - m - (di - dj) * e_del > zdrop;
			else stop = max - m - (dj - di) * e_ins > zdrop;
		}
		// band for the next row (ksw.c:502-505): skip leading / trailing columns whose {h,e} are both zero
		const int nbeg = first_nz >= 0 ? first_nz : end;
		const int jl = h1 != 0 ? end : (last_nz >= 0 ? last_nz : nbeg - 1);
		beg = nbeg;
		end = jl + 2 < qlen ? jl + 2 : qlen;
		}
		if (stop) break;
		++i;
	}
	hist_flush();
	#undef EHI
	#undef SCORE_AT
	#undef QBASE
	cells += cells32;
	ExtRes r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

// minimum / maximum of a 64-bit value over the wave (every lane receives it)
template <int CTRL, int RM> DEVFN i64 dpp_i64(i64 v)       // the 64-bit value of the DPP source lane (a lane without one, or outside the row mask: its own)
{
	const int lo = (int)(u32)(u64)v, hi = (int)(u32)((u64)v >> 32);
	const int l2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, RM, 0xf, false), h2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, RM, 0xf, false);
	return (i64)((u64)(u32)h2 << 32 | (u32)l2);
}
DEVFN i64 wave_min_i64(i64 v)
{
	i64 t;
	t = dpp_i64<DPP_ROW_SHR(1), 0xf>(v); v = t < v ? t : v; t = dpp_i64<DPP_ROW_SHR(2), 0xf>(v); v = t < v ? t : v;
	t = dpp_i64<DPP_ROW_SHR(4), 0xf>(v); v = t < v ? t : v; t = dpp_i64<DPP_ROW_SHR(8), 0xf>(v); v = t < v ? t : v;
	t = dpp_i64<DPP_ROW_BCAST15, 0xa>(v); v = t < v ? t : v; t = dpp_i64<DPP_ROW_BCAST31, 0xc>(v); v = t < v ? t : v;
	return readlane_i64_(v, 63);
}
DEVFN i64 wave_max_i64(i64 v)
{
	i64 t;
	t = dpp_i64<DPP_ROW_SHR(1), 0xf>(v); v = t > v ? t : v; t = dpp_i64<DPP_ROW_SHR(2), 0xf>(v); v = t > v ? t : v;
	t = dpp_i64<DPP_ROW_SHR(4), 0xf>(v); v = t > v ? t : v; t = dpp_i64<DPP_ROW_SHR(8), 0xf>(v); v = t > v ? t : v;
	t = dpp_i64<DPP_ROW_BCAST15, 0xa>(v); v = t > v ? t : v; t = dpp_i64<DPP_ROW_BCAST31, 0xc>(v); v = t > v ? t : v;
	return readlane_i64_(v, 63);
}

// Ascending sort of n DISTINCT 64-bit keys by the whole wave: a bitonic network in its all-ascending form (the first step of every merge stage
// pairs i with i ^ (2k - 1), the others i with i ^ j), so that positions past n behave as +infinity and pairs reaching there are skipped.  For
// mem_chain2aln's seed order (bwamem.c:684-685: score << 32 | index, unique keys) any correct sort gives what ks_introsort gives; one lane's
// introsort of a long read's ~2500 keys in global memory was tens of milliseconds per chain.
DEVFN void wave_sort_u64(u64 *a, int n)
{
	const int lane = opaque_lane();
	int N = 1; while (N < n) N <<= 1;
	for (int k = 2; k <= N; k <<= 1) {
		for (int j = k >> 1; j > 0; j >>= 1) {
			const int mask = j == (k >> 1) ? k - 1 : j;
			for (int i = lane; i < n; i += 64) {
				const int p = i ^ mask;
				if (p > i && p < n) { const u64 x = a[i], y = a[p]; if (x > y) { a[i] = y; a[p] = x; } }
			}
			wave_sync();
		}
	}
}

// "is the seed already covered by an earlier alignment of this read?" (bwamem.c:697-713) is an existence query -- the reference only uses
// whether its scan stopped early -- so 64 earlier regions are tested per step
DEVFN bool wave_seed_covered(const bwagpu_opt_t &opt, const bwagpu_seed_t &s, int l_query, const bwagpu_alnreg_t *av, int n_av)
{
	const int lane = opaque_lane();
	bool covered = false;
	for (int base = 0; base < n_av && !covered; base += 64) {
		const int ii = base + lane;
		bool hit = false;
		if (ii < n_av) {
			const bwagpu_alnreg_t &p = av[ii];
			if (!(s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) && !(s.len - p.seedlen0 > .1 * l_query)) {
				i64 rd; int qd, w, mg;
				qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
				mg = dev_max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p.w ? mg : p.w;
				if (qd - rd < w && rd - qd < w) hit = true;
				else {
					qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
					mg = dev_max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p.w ? mg : p.w;
					if (qd - rd < w && rd - qd < w) hit = true;
				}
			}
		}
		covered = __ballot(hit) != 0;
	}
	return covered;
}

// mem_chain2aln (bwamem.c:658-812) for ONE chain of a read: regions are appended at av[n_av ..]; "is the seed already covered" is asked of av[0 .. n_av).
// (Round 4 also extended the chains of reads with many chains by a wave each, ahead of their turn, and replayed the order-dependent decisions over
// the results -- exact, tested, and a loss on hardware: for 1 M short reads the kernel is throughput-bound (49.6 ms without, 55-61 ms with the two
// extra launches and the serial replay of the 640-chain read), and the long-read kernel's slow reads have ONE chain with many extended seeds
// (348 ms without, 800 ms with: profiles/r04_chain_parallel_*.jsonl).  Deleted; what stayed are the wave-parallel forms of the chain's serial steps.)
// The reference window a chain may reach (bwamem.c:669-683, bns_fetch_seq's clamp to the contig of seeds[0]: bntseq.c:426-451) and its seeds ordered by
// score (bwamem.c:684-685): the two wave-cooperative steps mem_chain2aln starts with.  k_ext_pack (dev_extp.h) runs them ahead of the extensions and leaves
// the results in the chain's ExtPlan; ext_chain_wave then finds them there.
DEVFN void chain_window_wave(const DevIndex &ix, const bwagpu_opt_t &opt, int l_query, const bwagpu_seed_t *seeds, int n, i64 &rmax0_, i64 &rmax1_)
{
	const int lane = opaque_lane();
	const i64 l_pac = ix.l_pac;
	i64 rmax0 = l_pac << 1, rmax1 = 0;
	for (int i = lane; i < n; i += 64) {      // (lanes stride over the seeds -- a long read's chain has thousands)
		const bwagpu_seed_t t = seeds[i];
		i64 b = t.rbeg - (t.qbeg + dev_max_gap(opt, t.qbeg));
		i64 e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + dev_max_gap(opt, l_query - t.qbeg - t.len));
		if (b < rmax0) rmax0 = b;
		if (e > rmax1) rmax1 = e;
	}
	rmax0 = wave_min_i64(rmax0); rmax1 = wave_max_i64(rmax1);
	if (rmax0 < 0) rmax0 = 0;
	if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
	if (rmax0 < l_pac && l_pac < rmax1) { if (seeds[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
	{
		int is_rev; int rid = dev_pos2rid(ix, dev_depos(ix, seeds[0].rbeg, &is_rev));
		i64 fb = ix.ctg_off[rid], fe = fb + ix.ctg_len[rid];
		if (is_rev) { i64 t = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t; }
		if (rmax0 < fb) rmax0 = fb;
		if (rmax1 > fe) rmax1 = fe;
	}
	rmax0_ = uni64(rmax0); rmax1_ = uni64(rmax1);
}
// the seeds by score: keys score << 32 | index are distinct, so the order is the keys' own whatever sorts them
DEVFN void chain_sort_wave(const bwagpu_seed_t *seeds, u64 *srt, int n)
{
	const int lane = opaque_lane();
	if (n <= 32) {
		if (lane == 0) {
			for (int i = 0; i < n; ++i) srt[i] = (u64)seeds[i].score << 32 | (u32)i;
			dev_introsort(srt, n, U64Less());
		}
		wave_sync();
	} else {
		for (int i = lane; i < n; i += 64) srt[i] = (u64)seeds[i].score << 32 | (u32)i;
		wave_sync();
		wave_sort_u64(srt, n);
	}
}

struct ExtPlan;
// an extension k_ext_pack has answered ahead of time (dev_extp.h; null / false: none)
DEVFN bool plan_result(const ExtPlan *plan, bool right, ExtRes &x);
DEVFN bool plan_window(const ExtPlan *plan, i64 &rmax0, i64 &rmax1);

template <bool RING> __device__ void ext_chain_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const WaveLds &L, const u8 *query, int l_query, int mat_max,
															   const bwagpu_chain_t &c, const bwagpu_seed_t *seeds, u64 *srt, int n, bwagpu_alnreg_t *av, int &n_av, const ExtPlan *plan)
{
	const int lane = opaque_lane();
	i64 rmax0, rmax1;
	if (RING || !plan_window(plan, rmax0, rmax1)) {
		chain_window_wave(ix, opt, l_query, seeds, n, rmax0, rmax1);
		chain_sort_wave(seeds, srt, n);
	}
	ext_stat_add(L, 0, 0, 0, (u64)(rmax1 - rmax0));
	// The seeds of this chain extended so far (not skipped: srt[] != 0 among the entries behind k), KS * 64 of them in registers, lane by lane.
	// The "other diagonal" rule below asks whether ANY of them overlaps the seed at hand: the reference walks srt[k+1 .. n) and skips the
	// zeroed entries (bwamem.c:716-717) -- for a long read's chain of ~2500 seeds, nearly all of them skipped, that walk was n^2 / 64 steps
	// of two dependent global loads each, most of the long-read kernel's time; the extended ones are a handful.  More than KS * 64 of them: the
	// chain falls back to the walk (srt[] is kept up to date for that).
	constexpr int KS = RING ? 2 : 1;
	int kq[KS], kl[KS]; i64 kr[KS]; int n_kept = 0; bool walk = false;
	#pragma unroll
	for (int t_ = 0; t_ < KS; ++t_) { kq[t_] = 0; kl[t_] = 0; kr[t_] = 0; }
	for (int k = n - 1; k >= 0; --k) {
		bwagpu_seed_t s = uni_seed(seeds[(u32)srt[k]]);
		const bool covered = wave_seed_covered(opt, s, l_query, av, n_av);
		if (covered) {   // extend anyway only if an overlapping seed sits on another diagonal (bwamem.c:714-732): also an existence query
			bool other = false;
			if (!walk) {
				bool hit = false;
				#pragma unroll
				for (int t_ = 0; t_ < KS; ++t_)
					if (t_ * 64 + lane < n_kept && !(kl[t_] < s.len * .95)) {
						if (s.qbeg <= kq[t_] && s.qbeg + s.len - kq[t_] >= s.len >> 2 && kq[t_] - s.qbeg != kr[t_] - s.rbeg) hit = true;
						else if (kq[t_] <= s.qbeg && kq[t_] + kl[t_] - s.qbeg >= s.len >> 2 && s.qbeg - kq[t_] != s.rbeg - kr[t_]) hit = true;
					}
				other = __ballot(hit) != 0;
			} else
			for (int base = k + 1; base < n && !other; base += 64) {
				const int i = base + lane;
				bool hit = false;
				if (i < n && srt[i] != 0) {
					bwagpu_seed_t t = seeds[(u32)srt[i]];
					if (!(t.len < s.len * .95)) {
						if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) hit = true;
						else if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) hit = true;
					}
				}
				other = __ballot(hit) != 0;
			}
			if (!other) {
				if (walk) wave_sync();             // every lane has finished reading srt[k..] before it is modified
				if (lane == 0) srt[k] = 0;
				if (walk) wave_sync();
				continue;
			}
		}
		if (!walk) {       // this seed is extended: it joins the list (its srt[] entry stays non-zero)
			if (n_kept < KS * 64) {
				#pragma unroll
				for (int t_ = 0; t_ < KS; ++t_) if (n_kept == t_ * 64 + lane) { kq[t_] = s.qbeg; kl[t_] = s.len; kr[t_] = s.rbeg; }
				++n_kept;
			} else { walk = true; wave_sync(); }       // (lane 0's zero stores so far are visible to the walks from here on)
		}
		bwagpu_alnreg_t a;
		a.rb = a.re = 0; a.qb = a.qe = 0; a.rid = c.rid; a.score = a.truesc = -1; a.sub = a.alt_sc = a.csub = a.sub_n = 0;
		a.w = opt.w; a.seedcov = 0; a.secondary = a.secondary_all = 0; a.seedlen0 = 0; a.n_comp = 0; a.is_alt = 0;
		a.frac_rep = 0.f; a.hash = 0;
		int aw0 = opt.w, aw1 = opt.w;
		if (s.qbeg) {
			ExtRes x; x.qle = x.tle = x.gtle = 0; x.gscore = -1; x.max_off = 0; x.score = -1;
			int tl = (int)(s.rbeg - rmax0);
			if (!RING && k == n - 1 && plan_result(plan, false, x)) a.score = x.score;      // (answered by k_ext_pack with the band opt.w, and not a result the loop below would re-run)
			else
			for (int i = 0; i < 2; ++i) {
				int prev = a.score;
				aw0 = opt.w << i;
				u64 cells_ = 0, fast_ = 0;
				x = wave_ksw_extend2<RING>(ix, opt, mat_max, query, s.qbeg - 1, -1, s.qbeg, s.rbeg - 1, -1, tl, aw0, opt.pen_clip5, s.len * opt.a, L, cells_, fast_);
				ext_stat_add(L, 1, cells_, (u32)fast_, 0);
				a.score = x.score;
				if (a.score == prev || x.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
			}
			if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip5) { a.qb = s.qbeg - x.qle; a.rb = s.rbeg - x.tle; a.truesc = a.score; }
			else { a.qb = 0; a.rb = s.rbeg - x.gtle; a.truesc = x.gscore; }
		} else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }
		if (s.qbeg + s.len != l_query) {
			ExtRes x; x.qle = x.tle = x.gtle = 0; x.gscore = -1; x.max_off = 0; x.score = -1;
			int sc0 = a.score, qe = s.qbeg + s.len;
			i64 re = s.rbeg + s.len;
			if (!RING && k == n - 1 && plan_result(plan, true, x)) a.score = x.score;
			else
			for (int i = 0; i < 2; ++i) {
				int prev = a.score;
				aw1 = opt.w << i;
				u64 cells_ = 0, fast_ = 0;
				x = wave_ksw_extend2<RING>(ix, opt, mat_max, query, qe, 1, l_query - qe, re, 1, (int)(rmax1 - re), aw1, opt.pen_clip3, sc0, L, cells_, fast_);
				ext_stat_add(L, 1, cells_, (u32)fast_, 0);
				a.score = x.score;
				if (a.score == prev || x.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
			}
			if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip3) { a.qe = qe + x.qle; a.re = re + x.tle; a.truesc += a.score - sc0; }
			else { a.qe = l_query; a.re = re + x.gtle; a.truesc += x.gscore - sc0; }
		} else { a.qe = l_query; a.re = s.rbeg + s.len; }
		int cov = 0;
		for (int i = opaque_lane(); i < n; i += 64) {   // seedcov (bwamem.c:801-805): lanes stride over the chain's seeds
			bwagpu_seed_t t = seeds[i];
			if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
		}
		a.seedcov = wave_sum(cov);
		a.w = aw0 > aw1 ? aw0 : aw1;
		a.seedlen0 = s.len;
		a.frac_rep = c.frac_rep;
		if (lane == 0) av[n_av] = a;
		++n_av;
		wave_sync();
	}
}

template <bool RING> __device__ void ext_read_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, const WaveLds &L0)
{
	const int lane = opaque_lane();
	WaveLds L = L0;
	r = uni(r);
	int n_ch = uni(B.chain_n[r]);
	if (n_ch == 0) { if (lane == 0) B.reg_n_raw[r] = 0; return; }
	const i64 qoff = uni64(B.off[r]);
	const u8 *query = B.seq + qoff;
	int l_query = uni((int)(B.off[r + 1] - qoff));
	i64 so = uni64(B.seed_off[r]);
	const RegionView R = region_of(B.slot_blob, so, uni(B.seed_n[r]));
	const bwagpu_chain_t *chains = R.cchain;
	const bwagpu_seed_t *seeds_all = R.cseed;
	u64 *srt_all = R.srt;
	bwagpu_alnreg_t *av = B.regs + uni64(B.reg_off[r]);
	int n_av = 0, sbeg = 0, mat_max = opt_mat_max(opt);
	if (!RING) {     // query profile of the whole read (ksw.c:425-428 builds one per call): every extension of the read indexes into it
		for (int j = lane; j < l_query; j += 64) {
			const int qc = query[j];
			for (int k = 0; k < 5; ++k) L.qp[k * L.qstride + j] = L.mat[k * 5 + qc];
		}
		wave_sync();
	}
	for (int ci = 0; ci < n_ch; ++ci) {
		bwagpu_chain_t c = chains[ci];
		c.rid = uni(c.rid); c.n_seeds = uni(c.n_seeds);      // (wave-uniform values loaded by a vector load: into scalar registers)
		const bwagpu_seed_t *seeds = seeds_all + sbeg;
		u64 *srt = srt_all + sbeg;
		int n = uni(c.n_seeds);
		sbeg += n;
		if (n == 0) continue;
		ext_chain_wave<RING>(ix, opt, L, query, l_query, mat_max, c, seeds, srt, n, av, n_av, B.ext_plan ? (const ExtPlan*)((const u8*)R.chain + (size_t)ci * 64) : nullptr);
	}
	if (lane == 0) B.reg_n_raw[r] = n_av;
}

// One wavefront per read, 4 waves per workgroup; dynamic LDS = 4 private regions of
// 8*(max_len+2+64) bytes of {H,E} columns + 5*qstride bytes of query profile, or (RING, long reads) of ring_cols*8 + 32 bytes.
template <bool RING, int OCC> __global__ void __launch_bounds__(256, OCC) k_extend_wave(DevIndex ix, bwagpu_opt_t opt, Batch B, int lds_per_wave, int ring_cols)
{
	HIP_DYNAMIC_SHARED(unsigned char, dyn_lds)
	const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
	WaveLds L;
	unsigned char *base = dyn_lds + (size_t)wave_in_blk * lds_per_wave;
	L.eh = (int2*)base;                                      // max_len + 2 columns + 64 of read-only padding
	if (RING) {
		int8_t *m = (int8_t*)(base + (size_t)8 * ring_cols);
		if (lane < 25) m[lane] = opt.mat[lane];
		L.mat = m; L.ring_mask = ring_cols - 1; L.qp = nullptr; L.qstride = 0;
	} else {
		L.qstride = (B.max_len + 64 + 3) & ~3;
		L.qp = (int8_t*)(base + (size_t)8 * (B.max_len + 2 + 64));
		int8_t *m = L.qp + 5 * L.qstride;                  // (the scoring matrix: dynamic indexing of the kernel argument would go through scratch memory)
		if (lane < 25) m[lane] = opt.mat[lane];
		L.mat = m; L.ring_mask = 0;
	}
	// (the mat copy takes 25 of the 32 bytes behind the columns / the ring; the stats counters the 32 after that)
	L.blk = RING ? B.ext_blk : 0;
	L.stat = B.stats ? (u32*)((unsigned char*)L.mat + 32) : nullptr;
	if (B.stats && lane < 8) ((u32*)((unsigned char*)L.mat + 32))[lane] = 0;
	wave_sync();
	u64 nraw = 0;
	// Reads are handed out heaviest first from a global counter: a wave that drew light reads simply draws more of them, and
	// the launch needs no particular relation between its grid and the number of resident workgroups.
	WaveQueue wq; wq_init(wq);                       // (the heaviest reads one at a time, the bulk in chunks: an eighth of the atomics on the counter)
	for (;;) {
		long long k;
		if (!wq_next(wq, &B.ctr->next_ext, B.n_reads, k)) break;
		const int r = B.order[k];
		long long t_0 = 0; u64 c_0 = 0, x_0 = 0;
		if (B.stats) { t_0 = wall_clock64(); c_0 = L.stat[2]; x_0 = (u64)L.stat[1] << 32 | L.stat[0]; }
		ext_read_wave<RING>(ix, opt, B, r, L);
		wave_sync();
		if (B.stats) {      // where the kernel's time goes, read by read (bwagpu_debug_hist)
			const long long dt = wall_clock64() - t_0;
			const int bin = dt > 0 ? (64 - __clzll(dt) < 31 ? 64 - __clzll(dt) : 31) : 0;
			const u64 calls = L.stat[2], cells = (u64)L.stat[1] << 32 | L.stat[0];
			if (lane == 0) { atomicAdd(&B.ctr->wave_hist[0][bin], 1ull); atomicAdd(&B.ctr->wave_hist[0][32 + bin], (unsigned long long)(calls - c_0)); atomicAdd(&B.ctr->wave_hist[0][64 + bin], (unsigned long long)((cells - x_0) >> 10)); }
			nraw += B.reg_n_raw[r];
		}
	}
	if (B.stats && lane == 0) {
		atomicAdd(&B.ctr->ext_calls, (unsigned long long)L.stat[2]);
		atomicAdd(&B.ctr->ext_cells, (unsigned long long)((u64)L.stat[1] << 32 | L.stat[0]));
		atomicAdd(&B.ctr->ext_fast, (unsigned long long)L.stat[3]);
		atomicAdd(&B.ctr->ref_bases, (unsigned long long)((u64)L.stat[5] << 32 | L.stat[4]));
		atomicAdd(&B.ctr->n_regs_raw, (unsigned long long)nraw);
	}
}
