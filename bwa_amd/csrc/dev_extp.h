// dev_extp.h -- packed seed extension: FOUR ksw_extend2 problems per wavefront, one per DPP row of 16 lanes.
//
// k_extend_wave (dev_extw.h) gives a whole wavefront to one extension; a 150 bp read's extensions have 25-100 live columns and a row costs
// ~90 instructions whatever the band holds, so the kernel is bound by instruction issue at ~10 % lane efficiency.  Here a
// 16-lane row owns one extension and every lane CPL adjacent query columns (CPL = 4: queries up to 63 bases, CPL = 8: up to 127): ALL of the
// query's columns {H(i-1,j-1), E(i,j)} live in registers -- the reference's eh[] array itself (ksw.c:416-515), no window, no LDS copy --, one
// instruction stream advances four extensions, and a group that finishes takes the next task from the wave's list while the others carry on.
//
// What makes the band cheap (proofs in the comments of pack_row):
//  * LEFT of the band nothing needs masking.  Columns the reference has trimmed (eh[j] = {0,0}, ksw.c:502-503) or that the band limit i - w has
//    passed keep computing zeros once their E is cleared in the row they leave the band, and contribute nothing to F or to the row maximum; `beg`
//    is therefore not tracked at all.
//  * RIGHT of the band (columns >= end) the registers hold {0,0} by the same trimming rule (given qlen <= w + 1: the limit i + w + 1 never
//    binds, so no column keeps its first-row value unvisited): E(i+1,j) = max(E - e_del, M - oe_del, 0) is 0 there whatever F says, and H is
//    handed on from the MASKED h of the column before, so one compare and one select per column is all the band costs.
//  * the new `end` is the last column with h > 0, plus three (e_new <= h, so a column's E can only be non-zero where its h is).
//
// A task that does not fit these conditions -- query longer than 16 * CPL - 1, qlen > w + 1, scores beyond the key packing, or a result that
// mem_chain2aln would re-run with a doubled band (bwamem.c:742-752) -- is simply left unanswered (valid = 0) and k_extend_wave computes it as before.
#pragma once
#include "dev_extw.h"

#define DPP_ROW_ROR(n) (0x120 + (n))
#define XP_NEG (-0x3fffffff)
typedef uint16_t u16;
#define XP_LIST 192        // tasks a wave plans ahead per list (the left and the right extension of a chain are never in a list at the same time)

// every lane of a 16-lane DPP row receives the row's maximum / bitwise OR / sum (rotations within the row: no lane is without a source).  `old` is the
// operation's identity in every step, which lets the compiler fold each mov_dpp + op pair into one v_max_i32_dpp / v_or_b32_dpp / v_add_u32_dpp
// (with old = v it emitted a copy, the DPP move and the operation: three instructions and a wait state per step).
DEVFN int row_allmax(int v)
{
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_ROR(8), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_ROR(4), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_ROR(2), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_ROR(1), 0xf, 0xf, false));
	return v;
}
DEVFN int row_allor(int v)
{
	v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(8), 0xf, 0xf, true);
	v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(4), 0xf, 0xf, true);
	v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(2), 0xf, 0xf, true);
	v |= __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(1), 0xf, 0xf, true);
	return v;
}
DEVFN int row_allsum(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(8), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(4), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(2), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_ROR(1), 0xf, 0xf, true);
	return v;
}
// lane L of a row: maximum over the row's lanes below L (XP_NEG for the row's first lane; v > XP_NEG everywhere)
DEVFN int row_excl_scan_max(int v)
{
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(1), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(2), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(4), 0xf, 0xf, false));
	v = imax(v, __builtin_amdgcn_update_dpp(I32_MIN, v, DPP_ROW_SHR(8), 0xf, 0xf, false));
	return __builtin_amdgcn_update_dpp(XP_NEG, v, DPP_ROW_SHR(1), 0xf, 0xf, false);
}
// ... sum over the row's lanes below L (0 for the first)
DEVFN int row_excl_scan_add(int v)
{
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(2), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(4), 0xf, 0xf, true);
	v += __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(8), 0xf, 0xf, true);
	return __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHR(1), 0xf, 0xf, true);
}
// the value lane `src` (0..15) of this lane's row holds
DEVFN int row_get(int v, int src) { return __shfl(v, (int)((threadIdx.x & 48) | (unsigned)src)); }

DEVFN int imax3(int a, int b, int c) { return imax(imax(a, b), c); }
DEVFN int imin(int a, int b) { return a < b ? a : b; }

// What the kernels keep per chain between the planning step, the packed extensions and k_extend_wave's replay: 64 bytes in the read's private
// region where the chaining stage's chain pool was (RegionView::chain -- dead once k_chain_wave has published the kept chains, and not used
// again before the de-duplication stage takes it for its sort keys).
struct ExtPlan {
	i64 rmax0, rmax1;      // the chain's reference window (bwamem.c:669-683)
	i64 rbeg;              // the chain's best seed (srt[n - 1], bwamem.c:684-690) ...
	u16 qbeg, len;         // ... (short reads only: the packed path is not built for long ones)
	u32 flags;             // bit 0: window and seed order are in place; bit 1 / 2: the left / right extension's result below is valid
	i32 l_score, l_gscore; u16 l_qle, l_tle, l_gtle, l_maxoff;
	i32 r_score, r_gscore; u16 r_qle, r_tle, r_gtle, r_maxoff;
};
static_assert(sizeof(ExtPlan) == 64, "layout");
#define XPF_PLANNED 1u
#define XPF_LEFT 2u
#define XPF_RIGHT 4u

// One extension as ksw_extend2 takes it (dev_ext.h: query j -> q[q0 + j*qdir], target i -> ref_base(t0 + i*tdir)).
struct PackTask { const u8 *q; int q0, qdir, qlen; i64 t0; int tdir, tlen, w, end_bonus, h0; };

// Per-wave constants of the scoring scheme (scalar registers).
struct PackConst { int o_del, e_del, o_ins, e_ins, oe_del, oe_ins, zdrop, mat_max; const int8_t *mat; int8_t *prof; };

template <int CPL> struct PackState {
	int H[CPL], E[CPL];            // this lane's columns lb .. lb + CPL - 1 of eh[]: {H(i-1,j-1), E(i,j)}
	int run;                       // the group's DP is under way
	int done;                      // ... or has a result to be collected (run == 0)
	int valid;                     // the result is ksw_extend2's (0: outside this routine's conditions)
	int i, end, qlen, tlen, w, h0;
	int max, max_i, max_j, max_ie, gscore, max_off;
	int tdir; i64 t0;
	u32 tpack; int tnext;          // reference bases of rows (i & ~15) .. + 15, 2 bits each; this lane's base of the NEXT sixteen rows (loaded a segment ahead)
	int lq, sq;                    // lane (0..15) and register of column qlen - 1, the source of the to-end score (ksw.c:486-489)
};

// ---- start of a task ---------------------------------------------------------------------------------------------------------------------
// ALL 64 lanes run this (its row-wide steps are at one place in the instruction stream for every lane -- the mock runtime of the CPU tests needs
// that, and the hardware loses nothing); the lanes of the group that takes the task (`mine`) commit, the others leave their state alone.  T is
// uniform within the group.  The group comes back with S.run = 1 (DP rows to do), or with S.done = 1 and its result in S: answered by the
// diagonal rule of dev_extw.h or by tlen <= 0, or valid = 0 -- outside this routine's conditions.
template <int CPL> DEVFN void pack_init(const DevIndex &ix, const PackConst &C, const PackTask &T, PackState<CPL> &S, bool mine, u64 &n_fast)
{
	const int gl = (int)(threadIdx.x & 15), grp = (int)((threadIdx.x >> 4) & 3), lb = gl * CPL;
	const int qlen = T.qlen, tlen = T.tlen, h0 = T.h0;
	int w = T.w;
	int lim = trunc_div_add(qlen * C.mat_max + T.end_bonus - C.o_ins, C.e_ins, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;      // ksw.c:436-443
	lim = trunc_div_add(qlen * C.mat_max + T.end_bonus - C.o_del, C.e_del, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;
	const bool ok = qlen >= 1 && qlen <= 16 * CPL - 1 && qlen <= w + 1 && h0 > 0 && (i64)h0 + (i64)qlen * C.mat_max < (1 << 22);
	const bool go = mine && ok && tlen > 0;
	// the query's bases of this lane's columns, and the reference's along the diagonal
	int qb[CPL], rb[CPL], sc[CPL];
	const int nd = qlen < tlen ? qlen : tlen;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int col = lb + c;
		qb[c] = go && col < qlen ? (int)T.q[T.q0 + col * T.qdir] : 4;
		rb[c] = go && col < nd ? ref_base(ix, T.t0 + (i64)col * T.tdir) : 0;
	}
	// ---- the extension that stays on the diagonal needs no DP (the rule and its proof: wave_ksw_extend2, dev_extw.h) ----
	const int oe_min = C.oe_del < C.oe_ins ? C.oe_del : C.oe_ins;
	int loss = 0, tot = 0;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) { sc[c] = lb + c < qlen ? (int)C.mat[rb[c] * 5 + qb[c]] : 0; loss += lb + c < qlen ? C.mat_max - sc[c] : 0; tot += sc[c]; }
	const int P = row_allsum(loss);
	const bool fast = go && tlen >= qlen && P < oe_min && (C.zdrop <= 0 || P < C.zdrop) && h0 > P;
	int run = h0 + row_excl_scan_add(tot), key = -1;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) { run += sc[c]; if (lb + c < qlen) key = imax(key, run << 7 | (127 - (lb + c))); }      // the FIRST column that attains the maximum (strict update, ksw.c:491)
	const int kmax = row_allmax(fast ? key : -1);
	const int total = h0 + row_allsum(tot);
#ifdef XP_TRACE
	if (getenv("XP_ROWS") && qlen == 15 && tlen == 89 && (threadIdx.x & 63) >= 56) fprintf(stderr, "[pre] lane %d mine %d ok %d go %d fast %d P %d kmax %d total %d\n", (int)(threadIdx.x & 63), (int)mine, (int)ok, (int)go, (int)fast, P, kmax, total);
#endif
	if (!mine) return;
	S.run = 0; S.done = 1; S.valid = ok ? 1 : 0;
	S.qlen = qlen; S.tlen = tlen; S.h0 = h0; S.t0 = T.t0; S.tdir = T.tdir; S.w = w;
	S.max = h0; S.max_i = S.max_j = S.max_ie = -1; S.gscore = -1; S.max_off = 0; S.i = 0; S.end = qlen;
	S.lq = qlen >= 1 ? (qlen - 1) / CPL : 0; S.sq = qlen >= 1 ? (qlen - 1) % CPL : 0;
	if (!go) return;             // not this routine's, or no row at all: score h0, qle = tle = gtle = 0, gscore -1 (the initial values above)
	if (fast) {
		if ((kmax >> 7) > h0) { S.max = kmax >> 7; S.max_i = S.max_j = 127 - (kmax & 127); }
		S.max_ie = qlen - 1; S.gscore = total;
		++n_fast;
		return;
	}
	// ---- DP: the query profile of the group (four reference bases x 16 * CPL columns, a byte each), the first row (ksw.c:430-433) ----
	int8_t *prof = C.prof + grp * (4 * 16 * CPL) + lb;
	#pragma unroll
	for (int b = 0; b < 4; ++b) {
		#pragma unroll
		for (int c = 0; c < CPL; ++c) prof[b * 16 * CPL + c] = lb + c < qlen ? C.mat[b * 5 + qb[c]] : (int8_t)0;
	}
	const int v1 = h0 > C.oe_ins ? h0 - C.oe_ins : 0;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int col = lb + c;
		int hv = col == 0 ? h0 : v1 - (col - 1) * C.e_ins;
		S.H[c] = col <= qlen && hv > 0 ? hv : 0; S.E[c] = 0;
	}
	S.tnext = gl < tlen ? ref_base(ix, T.t0 + (i64)gl * T.tdir) : 0;      // rows 0..15
	S.tpack = 0;
	S.run = 1; S.done = 0;
#ifdef XP_TRACE
	if (getenv("XP_ROWS") && qlen == 15) fprintf(stderr, "[init] lane %d gl %d tnext %d tlen %d t0 %lld tdir %d\n", (int)(threadIdx.x & 63), gl, S.tnext, tlen, (long long)T.t0, T.tdir);
#endif
}

// ---- one DP row for every group of the wave that is running (ALL 64 lanes execute this; groups that are not running compute on stale state and
// commit nothing) ---------------------------------------------------------------------------------------------------------------------------
template <int CPL, bool KEEP> DEVFN void pack_row(const DevIndex &ix, const PackConst &C, PackState<CPL> &S)
{
	// KEEP: a group that is not running keeps its result (the test kernel collects the four results at the end; k_ext_pack collects a group's result
	// before the next row, so there every update is unconditional -- selects instead of branches either way: the compiler turned `if (S.run) { ... }` into
	// a dozen register copies per row around a divergent region)
	const int gl = (int)(threadIdx.x & 15), grp = (int)((threadIdx.x >> 4) & 3), lb = gl * CPL;
	const int e_ins = C.e_ins, e_del = C.e_del, oe_del = C.oe_del;
	const bool run = S.run != 0;
	// reference bases: sixteen rows per segment, gathered from the lanes' loads; the next segment's loads are in flight meanwhile
	{
		const bool seg = run && (S.i & 15) == 0;
		if (__ballot(seg)) {
			const u32 tp = (u32)row_allor((int)((u32)S.tnext << (2 * gl)));
			if (seg) {
				S.tpack = tp;
				const int ii = S.i + 16 + gl;
				S.tnext = ii < S.tlen ? ref_base(ix, S.t0 + (i64)ii * S.tdir) : 0;
			}
		}
	}
	const int i = S.i;
	const int tb = (int)(S.tpack >> (2 * (i & 15))) & 3;
	const int8_t *sp = C.prof + grp * (4 * 16 * CPL) + tb * (16 * CPL) + lb;
	const int lane_e = lb * e_ins;
	int M[CPL], u[CPL], en[CPL], p[CPL];
	#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int sc = sp[c];
		M[c] = __mul24(sc, imin(S.H[c], 1)) + S.H[c];             // ksw.c:469: a dead diagonal cell stays dead (H >= 0)
		u[c] = M[c] + lane_e + (c * e_ins - C.oe_ins);             // F's seeds, each with its column's offset: F(j) = max_{k<j} u_k - (j-1) e_ins (ksw.c:480-483; not floored at 0 -- E >= 0 makes H the same)
		en[c] = imax3(S.E[c] - e_del, M[c] - oe_del, 0);           // E(i+1,j), ksw.c:475-479
		p[c] = c ? imax(p[c - 1], u[c]) : u[c];
	}
	const int X = row_excl_scan_max(p[CPL - 1]);
	const int de = S.end - lb;                                      // column lb + c is in the band iff c < de
	const int dw = lb + S.w - i;                                    // ... and stays in it for row i + 1 iff c + dw > 0 (ksw.c:452: beg >= i + 1 - w)
	int hk[CPL], key = -1, lp = -1;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) {
		const int pm = c ? imax(X, p[c - 1]) : X;
		const int F = pm - lane_e - (c - 1) * e_ins;
		const int h = imax3(M[c], S.E[c], F);                      // ksw.c:470-471
		hk[c] = c < de ? h : 0;
		key = imax(key, hk[c] << 8 | c);                           // row maximum, the last column on ties (ksw.c:473-474)
		lp = imax(lp, imin(hk[c], 1) << 8 | c);                    // the last column with h > 0
		en[c] = c + dw > 0 ? en[c] : 0;
	}
	const int kmax = row_allmax(key + lb), lpm = row_allmax(lp + lb);
	// H(i,j-1) moves one column up: within the lane, and from the lane below; column 0 receives H(i,-1) = h0 - (o_del + e_del (i+1)) while it is in the band (ksw.c:453-456)
	int hin = __builtin_amdgcn_update_dpp(0, hk[CPL - 1], DPP_ROW_SHR(1), 0xf, 0xf, true);
	{
		const int hdl = S.h0 - (C.o_del + e_del * (i + 1));
		hin = gl == 0 ? (i < S.w ? imax(hdl, 0) : 0) : hin;
	}
	// the to-end score's source: h of column qlen - 1 (what the reference leaves in eh[qlen].h when the band reaches the query's end)
	int h1 = hk[0];
	#pragma unroll
	for (int c = 1; c < CPL; ++c) h1 = S.sq == c ? hk[c] : h1;
	h1 = row_get(h1, S.lq);
	#pragma unroll
	for (int c = CPL - 1; c > 0; --c) S.H[c] = hk[c - 1];
	S.H[0] = hin;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) S.E[c] = en[c];
	const int m = kmax >> 8, mj = kmax & 255;
	const bool upd = KEEP ? run : true;
	{	// ksw.c:486-489 (before the m == 0 test, as there)
		const bool at_end = upd && S.end == S.qlen;
		S.max_ie = at_end && h1 >= S.gscore ? i : S.max_ie;
		S.gscore = at_end ? imax(S.gscore, h1) : S.gscore;
	}
	const bool nm = m > S.max;                                      // (m == 0 never is: max >= h0 > 0, so ksw.c:490's break needs no test here)
	bool stop = m == 0;
	{
		const int d = (i - S.max_i) - (mj - S.max_j);
		const int pen = d > 0 ? __mul24(d, e_del) : __mul24(-d, e_ins);
		stop = stop || (C.zdrop > 0 && !nm && S.max - m - pen > C.zdrop);      // ksw.c:494-500
		int off = mj - i; off = imax(off, -off);
		const bool nmu = nm && upd;
		S.max_off = nmu ? imax(S.max_off, off) : S.max_off;
		S.max_i = nmu ? i : S.max_i; S.max_j = nmu ? mj : S.max_j; S.max = nmu ? m : S.max;
	}
	// band for the next row (ksw.c:502-505): the last non-zero column is the one after the last h > 0 (m > 0: there is one)
	S.end = imin((lpm & 255) + 3, S.qlen);
	S.i = i + 1;
	const bool fin = stop || i + 1 >= S.tlen;
	S.done = KEEP ? (run ? (int)fin : S.done) : (int)(run && fin);
	S.run = run && !fin;
}

// the finished group's result as ksw_extend2 returns it
template <int CPL> DEVFN ExtRes pack_result(const PackState<CPL> &S)
{
	ExtRes r; r.score = S.max; r.qle = S.max_j + 1; r.tle = S.max_i + 1; r.gtle = S.max_ie + 1; r.gscore = S.gscore; r.max_off = S.max_off;
	return r;
}

// ---- what k_extend_wave's replay reads back (declared in dev_extw.h) ------------------------------------------------------------------------
DEVFN bool plan_window(const ExtPlan *plan, i64 &rmax0, i64 &rmax1)
{
	if (!plan) return false;
	if (!((u32)uni((int)plan->flags) & XPF_PLANNED)) return false;
	rmax0 = uni64(plan->rmax0); rmax1 = uni64(plan->rmax1);
	return true;
}
DEVFN bool plan_result(const ExtPlan *plan, bool right, ExtRes &x)
{
	if (!plan) return false;
	const u32 f = (u32)uni((int)plan->flags);
	if (!(f & XPF_PLANNED) || !(f & (right ? XPF_RIGHT : XPF_LEFT))) return false;
	if (right) { x.score = uni(plan->r_score); x.gscore = uni(plan->r_gscore); x.qle = uni((int)plan->r_qle); x.tle = uni((int)plan->r_tle); x.gtle = uni((int)plan->r_gtle); x.max_off = uni((int)plan->r_maxoff); }
	else { x.score = uni(plan->l_score); x.gscore = uni(plan->l_gscore); x.qle = uni((int)plan->l_qle); x.tle = uni((int)plan->l_tle); x.gtle = uni((int)plan->l_gtle); x.max_off = uni((int)plan->l_maxoff); }
	return true;
}

// ---- the wave's task lists (LDS) -----------------------------------------------------------------------------------------------------------
// entry: read | (chain << 1 | side) << 32, side 0 = the left extension of the chain's best seed, 1 = the right one.  list4: queries of up to 63
// bases (four columns per lane), list8: up to 127 (eight).  Counters are wave-uniform scalars; one lane stores.
struct PackLists { u64 *l4, *l8; int n4, n8; };
DEVFN void pack_push(PackLists &Q, int r, int ci, int side, int qlen)
{
	const u64 e = (u64)(u32)r | (u64)(u32)(ci << 1 | side) << 32;
	if (qlen <= 63) { if ((threadIdx.x & 63) == 0) Q.l4[Q.n4] = e; ++Q.n4; }
	else if (qlen <= 127) { if ((threadIdx.x & 63) == 0) Q.l8[Q.n8] = e; ++Q.n8; }
	wave_sync();      // (one lane's store, every lane's load: in lockstep on the device, an ordering point for the lane-serial mock runtime)
}

// mem_chain2aln's first two steps for every chain of read r (window, seed order), left in the chains' ExtPlan records; the best seed's first
// extension -- the left one, or the right one when the seed starts the read -- joins the lists while there is room for `room` tasks.
// Returns the number of tasks pushed.
DEVFN int pack_plan_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, PackLists &Q, int room)
{
	const int lane = opaque_lane();
	r = uni(r);
	const int n_ch = uni(B.chain_n[r]);
	if (n_ch == 0) return 0;
	const i64 qoff = uni64(B.off[r]);
	const int l_query = uni((int)(B.off[r + 1] - qoff));
	const RegionView R = region_of(B.slot_blob, uni64(B.seed_off[r]), uni(B.seed_n[r]));
	int sbeg = 0, pushed = 0;
	for (int ci = 0; ci < n_ch; ++ci) {
		const int n = uni(R.cchain[ci].n_seeds);
		const bwagpu_seed_t *seeds = R.cseed + sbeg;
		u64 *srt = R.srt + sbeg;
		sbeg += n;
		ExtPlan *pl = (ExtPlan*)((u8*)R.chain + (size_t)ci * 64);
		if (n == 0) { if (lane == 0) pl->flags = 0; continue; }
		i64 rmax0, rmax1;
		chain_window_wave(ix, opt, l_query, seeds, n, rmax0, rmax1);
		chain_sort_wave(seeds, srt, n);
		const bwagpu_seed_t s = uni_seed(seeds[(u32)srt[n - 1]]);
		if (lane == 0) { pl->rmax0 = rmax0; pl->rmax1 = rmax1; pl->rbeg = s.rbeg; pl->qbeg = (u16)s.qbeg; pl->len = (u16)s.len; pl->flags = XPF_PLANNED; }
		if (pushed < room) {
			const int before = Q.n4 + Q.n8;
			if (s.qbeg > 0) pack_push(Q, r, ci, 0, s.qbeg);
			else if (s.qbeg + s.len != l_query) pack_push(Q, r, ci, 1, l_query - (s.qbeg + s.len));
			pushed += Q.n4 + Q.n8 - before;
		}
	}
	wave_sync();
	return pushed;
}

// Works through one of the lists: idle groups take tasks, all running groups advance a row per turn, finished groups leave their result in the
// chain's plan and, after a left extension, queue the right one (which may belong to the other list).
template <int CPL> __device__ void pack_phase(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, const PackConst &C, PackLists &Q, unsigned long long &n_done)
{
	const int lane = (int)(threadIdx.x & 63), grp = lane >> 4;
	PackState<CPL> S;
	#pragma unroll
	for (int c = 0; c < CPL; ++c) { S.H[c] = 0; S.E[c] = 0; }
	S.run = 0; S.done = 0; S.valid = 0; S.i = 0; S.end = 0; S.qlen = 1; S.tlen = 0; S.w = 0; S.h0 = 1; S.max = 0; S.max_i = S.max_j = S.max_ie = -1; S.gscore = -1; S.max_off = 0;
	S.tdir = 1; S.t0 = 0; S.tpack = 0; S.tnext = 0; S.lq = 0; S.sq = 0;
	int t_r = 0, t_cs = 0, t_lq = 0;       // the group's task: read, chain << 1 | side, the read's length
	const int w34 = (opt.w >> 1) + (opt.w >> 2);
	u64 n_fast = 0;
	int &n_list = CPL == 4 ? Q.n4 : Q.n8;
	u64 *list = CPL == 4 ? Q.l4 : Q.l8;
	for (;;) {
		// finished groups: the result goes to the chain's plan; a left extension queues the right one.  Idle groups: the next task of the list.
		const u64 dm = __ballot(S.done != 0);
		u64 rm = __ballot(S.run != 0);
		if (dm || (n_list > 0 && (rm | dm) != ~0ull)) {
		for (int g = 0; g < 4; ++g) {
			const int l0 = g * 16;
			bool fin = (dm >> l0) & 1;
			if (!fin && !((rm >> l0) & 1) && n_list > 0) {
				// an idle group: the next task of the list
				--n_list;
				const u64 e = list[n_list];
				const int r = uni((int)(u32)e), cs = uni((int)(u32)(e >> 32)), ci = cs >> 1, side = cs & 1;
				const i64 qoff = uni64(B.off[r]);
				const int l_query = uni((int)(B.off[r + 1] - qoff));
				const ExtPlan *pl = (const ExtPlan*)((const u8*)region_of(B.slot_blob, uni64(B.seed_off[r]), uni(B.seed_n[r])).chain + (size_t)ci * 64);
				const i64 rbeg = uni64(pl->rbeg);
				const int qbeg = uni((int)pl->qbeg), len = uni((int)pl->len);
				PackTask T;
				T.q = B.seq + qoff; T.w = opt.w;
				if (side == 0) { T.q0 = qbeg - 1; T.qdir = -1; T.qlen = qbeg; T.t0 = rbeg - 1; T.tdir = -1; T.tlen = (int)(rbeg - uni64(pl->rmax0)); T.end_bonus = opt.pen_clip5; T.h0 = len * opt.a; }
				else {
					const int qe = qbeg + len; const i64 re = rbeg + len;
					T.q0 = qe; T.qdir = 1; T.qlen = l_query - qe; T.t0 = re; T.tdir = 1; T.tlen = (int)(uni64(pl->rmax1) - re); T.end_bonus = opt.pen_clip3;
					T.h0 = ((u32)uni((int)pl->flags) & XPF_LEFT) ? uni(pl->l_score) : len * opt.a;      // sc0 of bwamem.c:758-760
				}
#ifdef XP_TRACE
				if (getenv("XP_ROWS") && (lane == 59 || lane == 60 || lane == 0)) fprintf(stderr, "[asg] lane %d g %d n_list %d r %d cs %d qlen %d tlen %d h0 %d list %p\n", lane, g, n_list, r, cs, T.qlen, T.tlen, T.h0, (void*)list);
#endif
				if (grp == g) { t_r = r; t_cs = cs; t_lq = l_query; }
				pack_init<CPL>(ix, C, T, S, grp == g, n_fast);
				fin = __builtin_amdgcn_readlane(S.done, l0) != 0;
			}
			if (fin) {
				const int r = __builtin_amdgcn_readlane(t_r, l0), cs = __builtin_amdgcn_readlane(t_cs, l0), ci = cs >> 1, side = cs & 1;
				const int l_query = __builtin_amdgcn_readlane(t_lq, l0);
				ExtPlan *pl = (ExtPlan*)((u8*)region_of(B.slot_blob, uni64(B.seed_off[r]), uni(B.seed_n[r])).chain + (size_t)ci * 64);
				const int sc = __builtin_amdgcn_readlane(S.max, l0), gs = __builtin_amdgcn_readlane(S.gscore, l0);
				const int qle = __builtin_amdgcn_readlane(S.max_j, l0) + 1, tle = __builtin_amdgcn_readlane(S.max_i, l0) + 1, gtle = __builtin_amdgcn_readlane(S.max_ie, l0) + 1;
				const int mo = __builtin_amdgcn_readlane(S.max_off, l0);
				// usable iff it is ksw_extend2's and mem_chain2aln would not run it again with the band doubled (bwamem.c:742-752: prev is -1 in the first round)
				const bool ok = __builtin_amdgcn_readlane(S.valid, l0) != 0 && mo < w34;
#ifdef XP_TRACE
				{ const int a_ = __builtin_amdgcn_readlane(S.qlen, l0), b_ = __builtin_amdgcn_readlane(S.tlen, l0), c_ = __builtin_amdgcn_readlane(S.h0, l0), d_ = __builtin_amdgcn_readlane(S.w, l0), e_ = __builtin_amdgcn_readlane(S.i, l0);
				if (lane == 0) fprintf(stderr, "[xp] CPL %d g %d r %d ci %d side %d -> score %d qle %d tle %d gtle %d gscore %d maxoff %d valid %d (qlen %d tlen %d h0 %d w %d rows %d)\n", CPL, g, r, ci, side, sc, qle, tle, gtle, gs, mo, (int)ok, a_, b_, c_, d_, e_); }
#endif
				if (ok) {
					if (lane == l0) {
						if (side == 0) { pl->l_score = sc; pl->l_gscore = gs; pl->l_qle = (u16)qle; pl->l_tle = (u16)tle; pl->l_gtle = (u16)gtle; pl->l_maxoff = (u16)mo; pl->flags |= XPF_LEFT; }
						else { pl->r_score = sc; pl->r_gscore = gs; pl->r_qle = (u16)qle; pl->r_tle = (u16)tle; pl->r_gtle = (u16)gtle; pl->r_maxoff = (u16)mo; pl->flags |= XPF_RIGHT; }
					}
					++n_done;
					if (side == 0) {
						wave_sync();      // (the right task's start reads l_score and the flag back)
						const int qe = uni((int)pl->qbeg) + uni((int)pl->len);
						if (qe != l_query) pack_push(Q, r, ci, 1, l_query - qe);
					}
				}
				if (grp == g) { S.done = 0; S.run = 0; }
			}
		}
		rm = __ballot(S.run != 0);
		}
		if (!rm) { if (n_list == 0) break; continue; }
		pack_row<CPL, false>(ix, C, S);
	}
}

#define XP_LDS_BYTES (32 + 4 * 4 * 16 * 8 + 2 * XP_LIST * 8)
// One pass over the batch's reads ahead of k_extend_wave: per wave, plan a few dozen reads, then run their chains' first extensions four at a time.
template <int OCC> __global__ void __launch_bounds__(256, OCC) k_ext_pack(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	HIP_DYNAMIC_SHARED(unsigned char, dyn_lds)
	const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned char *base = dyn_lds + (size_t)wave_in_blk * XP_LDS_BYTES;
	int8_t *mat = (int8_t*)base;
	if (lane < 25) mat[lane] = opt.mat[lane];
	PackConst C;
	C.o_del = opt.o_del; C.e_del = opt.e_del; C.o_ins = opt.o_ins; C.e_ins = opt.e_ins; C.oe_del = opt.o_del + opt.e_del; C.oe_ins = opt.o_ins + opt.e_ins;
	C.zdrop = opt.zdrop; C.mat = mat; C.prof = (int8_t*)(base + 32);
	PackLists Q; Q.l4 = (u64*)(base + 32 + 4 * 4 * 16 * 8); Q.l8 = Q.l4 + XP_LIST; Q.n4 = Q.n8 = 0;
	wave_sync();
	C.mat_max = opt_mat_max(opt);
	WaveQueue wq; wq_init(wq);
	long long pending = -1;
	bool more = true;
	unsigned long long n_done = 0;
	for (;;) {
		int added = 0;
		while (more && added < XP_LIST - 16) {
			long long k;
			if (pending >= 0) { k = pending; pending = -1; }
			else if (!wq_next(wq, &B.ctr->next_pack, B.n_reads, k)) { more = false; break; }
			const int r = uni(B.order[k]);
			const int n_ch = uni(B.chain_n[r]);
			if (added > 0 && added + n_ch > XP_LIST) { pending = k; break; }
			added += pack_plan_read(ix, opt, B, r, Q, XP_LIST - added);
		}
		if (Q.n4 == 0 && Q.n8 == 0) { if (!more && pending < 0) break; continue; }
		while (Q.n4 > 0 || Q.n8 > 0) {
			if (Q.n4 > 0) pack_phase<4>(ix, opt, B, C, Q, n_done);
			if (Q.n8 > 0) pack_phase<8>(ix, opt, B, C, Q, n_done);
		}
	}
	if (lane == 0 && n_done) atomicAdd(&B.ctr->prof[8], n_done);      // (bwagpu_debug_prof: extensions answered here)
}
