// dev_cigar.h -- banded global alignment with traceback for the final regions: what mem_reg2aln (bwamem.c:1119-1152)
// obtains from bwa_gen_cigar2 (bwa.c:148-234) / ksw_global2 (ksw.c:540-642), one wavefront per region.
//
// ksw_global2 opens gaps from the diagonal term M as ksw_extend2 does, so a DP row is again a max-plus prefix scan over
// M(i,.): lanes own the columns of the band (64 per pass, up to CIG_MAX_COLS), rows are sequential.  The per-column state
// {H(i-1,j-1), E(i,j)}, the query profile and the direction bytes live in LDS; lane 0 walks the direction bytes back.
// All integers are those of the scalar recurrence (including the -2^30 "minus infinity" terms), so direction bytes,
// score and CIGAR are identical to the host's.  The kernel runs in two tiers that differ in the LDS they reserve for
// direction nibbles: most regions need a narrow band and run at high occupancy, the few with a wide band are redone by a
// second launch.  Regions outside the limits (band wider than CIG_MAX_COLS, more than CIG_Z_BIG cells, more than
// CIG_MAX_OPS operations, windows spanning the forward/reverse boundary) are flagged n_cigar = -1 and left to the caller's
// own bwa_gen_cigar2.
#pragma once
#include "dev_extw.h"

#define CIG_NEG_INF (-0x40000000)
#define CIG_MAX_LEN 320        // longest query / target segment handled here
#define CIG_Z_SMALL 6144       // DP cells (direction nibbles, two rows per byte) per wave: first tier ...
#define CIG_Z_BIG 28672        // ... and second tier
#define CIG_MAX_COLS 192       // widest band (columns per row)
#define CIG_MAX_OPS 6          // operations kept per record (bwagpu_cigar_t)
#define CIG_TMP_OPS 64
#define CIG_MD_CAP 1024        // longest MD string kept (bytes of LDS per wave)

struct CigLds { i32 *hd, *e; int8_t *qp; u8 *z; u32 *ops; u8 *md; int qstride, z_cells; };
#define CIG_LDS_BYTES_DIAG(zc) ((5 * (CIG_MAX_LEN + 64) + (zc) / 2 + CIG_MAX_COLS + CIG_TMP_OPS * 4 + CIG_MD_CAP + 15) & ~15)   // per wave, first tier (diag form only)
#define CIG_LDS_BYTES(zc) ((2 * (CIG_MAX_LEN + 2 + 64) * 4 + 5 * (CIG_MAX_LEN + 64) + (zc) / 2 + CIG_MAX_COLS + CIG_TMP_OPS * 4 + CIG_MD_CAP + 15) & ~15)   // per wave

// The query profile of a region's query segment (ksw.c:552-556), qp[b * qstride + j] = score of reference base b against column j: the same for
// every band-doubling attempt of the region, so it is built once per region.
DEVFN void cig_build_profile(const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, const CigLds &L)
{
	const int lane = threadIdx.x & 63;
	for (int j = lane; j < qlen; j += 64) {
		const int qc = q[q0 + j * qdir];
		for (int k = 0; k < 5; ++k) L.qp[k * L.qstride + j] = opt.mat[k * 5 + qc];
	}
	wave_sync();
}

// ---- ksw_global2's matrix fill and traceback (ksw.c:566-639) in two forms ------------------------------------------------------------------------
// "diag": bands of up to 64 columns (2 w + 1 <= 64: nearly every region of a 150 bp read, whose band comes from infer_bw, bwamem.c:818-825).  Lane L owns
//   the band's DIAGONAL L = j - i + w.  The diagonal term H(i-1,j-1) is then the lane's own H of the row before -- it never moves --, E(i,j) comes
//   from the lane above (E(i,j) was computed in row i-1 by the lane that owned column j there), F is the usual max-plus prefix scan over the row, and the
//   band is a pair of lane numbers that changes by at most one per row.  No {H,E} arrays, no LDS round trip and no wave fence per row: the only LDS
//   traffic is the row's score and, every eighth row, one word of direction nibbles per lane (a lane packs its eight rows in a register; round 4's
//   form read and wrote hd[], e[] and a direction byte per cell, with two fences per 64-column pass).  The traceback reads those words by diagonal:
//   a run of matches stays on its diagonal, so lane 0 takes a whole word's worth of "came from the diagonal" steps at once.
// "lds": wider bands (up to CIG_MAX_COLS columns), the round-2 form: columns in LDS, 64 per pass, direction nibbles two rows per byte.
// Both leave the operations in L.ops in traceback (reversed) order, run-length merged; all integers are the scalar recurrence's.
DEVFN bool cig_diag_form(int qlen, int tlen, int w) { return 2 * w + 1 <= 64 && qlen > 0 && tlen > 0; }
// direction nibbles of the diag form: word (i >> 3) * (2 w + 1) + L holds rows 8 (i >> 3) .. + 7 of diagonal L
DEVFN int cig_diag_cells(int tlen, int w) { return ((tlen + 7) & ~7) * (2 * w + 1); }

#define DPP_WAVE_SHL1 0x130
// value of the lane above (lane 63 receives `fill`)
DEVFN int wave_shift_down1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, DPP_WAVE_SHL1, 0xf, 0xf, false); }

__device__ int wave_global2_fill_diag(const DevIndex &ix, const bwagpu_opt_t &opt, int qlen, i64 t0, int tdir, int tlen, int w, const CigLds &L, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	qlen = uni(qlen); tlen = uni(tlen); w = uni(w); t0 = uni64(t0);
	const int ncl = 2 * w + 1, qs = L.qstride;
	const int8_t *qp = L.qp; u32 *zw = (u32*)L.z;
	const int lane_e = lane * e_ins;
	int j = lane - w;                                     // this lane's column in the current row (one more every row)
	int hcur = j == 0 ? 0 : -(o_ins + e_ins * j);          // H(i-1, j-1): the first row's values (ksw.c:566-570; only the lanes of row 0's band, 0 <= j <= w, use theirs)
	int eout = CIG_NEG_INF;                               // E(i, .) as this lane left it in the row before: the lane below reads it
	int lo = w, hi = qlen + w < ncl ? qlen + w : ncl;     // the row's live lanes [lo, hi): columns max(0, i - w) .. min(qlen, i + w + 1) - 1
	u32 zacc = 0, cells32 = 0;
	int treg = 0;
	{ treg = lane < tlen ? ref_base(ix, t0 + (i64)lane * tdir) : 0; }
	auto score_at = [&](int tb, int jj) -> int { const int jc = jj < 0 ? 0 : (jj < qlen ? jj : qlen - 1); return (int)qp[tb * qs + jc]; };      // (lanes outside the band read a neighbour's score and drop it)
	int sc_next = score_at(__builtin_amdgcn_readlane(treg, 0), j);
	for (int i = 0; i < tlen; ++i) {
		const int sc = sc_next;
		if (i + 1 < tlen) {                               // the next row's score, in flight while this row computes
			if (((i + 1) & 63) == 0) { const int ii = i + 1 + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
			sc_next = score_at(__builtin_amdgcn_readlane(treg, (i + 1) & 63), j + 1);
		}
		const bool act = lane >= lo && lane < hi;
		const int dg = (j == 0 && i > 0) ? -(o_del + e_del * i) : hcur;        // column 0's diagonal term is H(i-1,-1) (ksw.c:578)
		const int ec = wave_shift_down1(eout, CIG_NEG_INF);                      // (past the band's right edge: -inf, ksw.c:619)
		const int m = dg + sc;
		const int a = act ? m - oe_ins + lane_e : I32_MIN;
		const int exc = wave_shift_up1(wave_incl_scan_max(a), I32_MIN);
		int f = CIG_NEG_INF - (lane - lo) * e_ins;                             // F(i, beg) = -inf, then f <- max(f - e, m - oe) (ksw.c:596-599)
		if (lane > lo && act) f = imax(f, exc - (lane_e - e_ins));
		int d = m >= ec ? 0 : 1, h = m >= ec ? m : ec;                          // ksw.c:587-590
		if (h < f) { d = 2; h = f; }
		int t = m - oe_del, en = ec - e_del;
		if (en > t) d |= 4; else en = t;                                       // E continues (ksw.c:592-595)
		t = m - oe_ins;
		if (f - e_ins > t) d |= 8;                                            // F continues (ksw.c:596-599)
		hcur = act ? h : hcur;
		eout = act ? en : CIG_NEG_INF;
		const int sh = (i & 7) << 2;
		zacc = sh == 0 ? (u32)d : zacc | (u32)d << sh;
		if ((sh == 28 || i == tlen - 1) && lane < ncl) zw[(i >> 3) * ncl + lane] = zacc;
		cells32 += (u32)(hi > lo ? hi - lo : 0);
		++j;
		lo = lo > 0 ? lo - 1 : 0;
		{ const int nh = qlen + w - (i + 1); hi = nh < ncl ? nh : ncl; }
	}
	cells += cells32;
	wave_sync();                                                               // the direction words are read back by lane 0
	// H(tlen-1, qlen-1), if the last row's band holds that cell (it does whenever w >= |tlen - qlen|); else what eh[qlen].h still holds: -inf (ksw.c:570,619)
	const int ls = uni((qlen - 1) - (tlen - 1) + w), lo_l = w - (tlen - 1) > 0 ? w - (tlen - 1) : 0, hi_l = qlen + w - (tlen - 1) < ncl ? qlen + w - (tlen - 1) : ncl;
	return ls >= lo_l && ls < hi_l ? __builtin_amdgcn_readlane(hcur, ls) : CIG_NEG_INF;
}

// traceback of the diag form (ksw.c:624-639), lane 0; *n_ops < 0: more than CIG_TMP_OPS operations
__device__ void wave_global2_trace_diag(int qlen, int tlen, int w, const CigLds &L, int *n_ops)
{
	const int lane = threadIdx.x & 63;
	int n = 0;
	if (lane == 0) {
		const int ncl = 2 * w + 1;
		const u32 *zw = (const u32*)L.z;
		u32 *ops = L.ops;
		int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
		auto push = [&](int op, int len) {
			if (n > 0 && (int)(ops[n - 1] & 0xf) == op) ops[n - 1] += (u32)len << 4;
			else if (n < CIG_TMP_OPS) ops[n++] = (u32)len << 4 | (u32)op;
			else n = CIG_TMP_OPS + 1;
		};
		int cw_row = -1, cw_l = -1; u32 cw = 0;           // the cached word: rows 8 cw_row .. + 7 of diagonal cw_l
		while (i >= 0 && k >= 0 && n <= CIG_TMP_OPS) {
			const int ld = k - i + w;
			if ((i >> 3) != cw_row || ld != cw_l) { cw_row = i >> 3; cw_l = ld; cw = (ld >= 0 && ld < ncl) ? zw[cw_row * ncl + ld] : 0; }
			const int p = i & 7;
			if (which == 0) {
				// in H: rows p, p-1, .. of this word whose H came from the diagonal (low two bits 0) are match steps on this very diagonal: take them at once
				const u32 low = cw & 0x33333333u & (p == 7 ? ~0u : (1u << ((p + 1) << 2)) - 1u);
				int run = low ? p - ((31 - __clz((int)low)) >> 2) : p + 1;
				if (run > k + 1) run = k + 1;
				if (run > 0) { push(0, run); i -= run; k -= run; continue; }
			}
			const u32 nib = cw >> (p << 2) & 15u;
			// states: 0 = in H (take its source), 1 = in E (continue the deletion?), 2 = in F (continue the insertion?)
			which = which == 0 ? (int)(nib & 3) : which == 1 ? (int)(nib >> 2 & 1) : (int)(nib >> 3 & 1) << 1;
			if (which == 0) { push(0, 1); --i; --k; }
			else if (which == 1) { push(2, 1); --i; }
			else { push(1, 1); --k; }
		}
		if (n <= CIG_TMP_OPS && i >= 0) push(2, i + 1);
		if (n <= CIG_TMP_OPS && k >= 0) push(1, k + 1);
	}
	n = __builtin_amdgcn_readlane(n, 0);
	wave_sync();
	*n_ops = n > CIG_TMP_OPS ? -1 : n;
}

// the lds form's matrix fill (ksw.c:566-619); returns the score
__device__ int wave_global2_fill_lds(const DevIndex &ix, const bwagpu_opt_t &opt, int qlen, i64 t0, int tdir, int tlen, int w, const CigLds &L, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	i32 *hd = L.hd, *e_ = L.e; int8_t *qp = L.qp; u8 *z = L.z; const int qs = L.qstride;
	for (int j = lane; j <= qlen; j += 64) {      // first row (ksw.c:566-570)
		hd[j] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : CIG_NEG_INF);
		e_[j] = CIG_NEG_INF;
	}
	wave_sync();
	int treg = 0;
	u32 cells32 = 0;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) { int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : CIG_NEG_INF;
		cells32 += (u32)(end > beg ? end - beg : 0);
		int carry = I32_MIN, bnd = 0;
		for (int b = beg; b < end; b += 64) {
			const int j = b + lane; const bool act = j < end;
			int dg = hd[j]; const int ec = e_[j];          // padded arrays: inactive lanes read, never write
			const int sc = qp[tb * qs + j];
			const int bnd_next = hd[b + 64];                // the next pass's first diagonal, before this pass's lane 63 overwrites it
			if (b != beg && lane == 0) dg = bnd;
			wave_sync();
			const int m = dg + sc;
			// F(i,j) = max( -inf - (j-beg) e_ins , max_{beg<=k<j} m_k - oe_ins - (j-1-k) e_ins ): the scalar chain f <- max(f - e, m - oe)
			const int a = act ? m - oe_ins + j * e_ins : I32_MIN;
			const int inc = wave_incl_scan_max(a);
			const int exc = imax(wave_shift_up1(inc, I32_MIN), carry);
			int f = CIG_NEG_INF - (j - beg) * e_ins;
			if (j > beg && act) f = imax(f, exc - (j - 1) * e_ins);
			int d = m >= ec ? 0 : 1, h = m >= ec ? m : ec;
			if (h < f) { d = 2; h = f; }
			int t = m - oe_del, en = ec - e_del;
			if (en > t) d |= 1 << 2; else en = t;
			t = m - oe_ins; const int fn = f - e_ins;
			if (fn > t) d |= 2 << 4;
			if (act) {
				e_[j] = en;
				hd[j + 1] = h;
				// direction nibble {H source (2 bits), E continues, F continues}; rows 2r and 2r+1 of a band column share a byte and
				// are written by the same lane
				const u32 nib = (u32)(d & 7) | (u32)(d >> 5 & 1) << 3;
				u8 *zb = z + (i >> 1) * n_col + (j - beg);
				// (an odd row replaces the high nibble rather than OR-ing into it: while the band still widens, its last column has no even-row
				// partner that would have cleared the byte, and the LDS holds whatever the previous region left there)
				*zb = (i & 1) ? (u8)((*zb & 15u) | nib << 4) : (u8)nib;
			}
			if (b == beg && lane == 0) hd[beg] = h1_init;
			carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
			bnd = bnd_next;
			wave_sync();
		}
		if (lane == 0) e_[end] = CIG_NEG_INF;            // hd[end] = H(i,end-1) was written by the last active lane
		wave_sync();
	}
	cells += cells32;
	return hd[qlen];
}

// ... and its traceback (ksw.c:624-639), lane 0; the other lanes wait at the barrier below
__device__ void wave_global2_trace_lds(int qlen, int tlen, int w, const CigLds &L, int *n_ops)
{
	const int lane = threadIdx.x & 63;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	const u8 *z = L.z;
	int n = 0;
	if (lane == 0) {
		u32 *ops = L.ops;
		int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
		auto push = [&](int op, int len) {
			if (n > 0 && (int)(ops[n - 1] & 0xf) == op) ops[n - 1] += (u32)len << 4;
			else if (n < CIG_TMP_OPS) ops[n++] = (u32)len << 4 | (u32)op;
			else n = CIG_TMP_OPS + 1;
		};
		while (i >= 0 && k >= 0 && n <= CIG_TMP_OPS) {
			const u32 nib = z[(i >> 1) * n_col + (k - (i > w ? i - w : 0))] >> ((i & 1) << 2) & 15;
			// states: 0 = in H (take its source), 1 = in E (continue the deletion?), 2 = in F (continue the insertion?)
			which = which == 0 ? (int)(nib & 3) : which == 1 ? (int)(nib >> 2 & 1) : (int)(nib >> 3 & 1) << 1;
			if (which == 0) { push(0, 1); --i; --k; }
			else if (which == 1) { push(2, 1); --i; }
			else { push(1, 1); --k; }
		}
		if (n <= CIG_TMP_OPS && i >= 0) push(2, i + 1);
		if (n <= CIG_TMP_OPS && k >= 0) push(1, k + 1);
	}
	n = __builtin_amdgcn_readlane(n, 0);
	wave_sync();
	*n_ops = n > CIG_TMP_OPS ? -1 : n;
}

// fill + traceback in whichever form the band takes; the profile of the query segment must be in place (cig_build_profile)
DEVFN int wave_global2_fill(const DevIndex &ix, const bwagpu_opt_t &opt, int qlen, i64 t0, int tdir, int tlen, int w, const CigLds &L, u64 &cells)
{
	return cig_diag_form(qlen, tlen, w) ? wave_global2_fill_diag(ix, opt, qlen, t0, tdir, tlen, w, L, cells) : wave_global2_fill_lds(ix, opt, qlen, t0, tdir, tlen, w, L, cells);
}
DEVFN void wave_global2_trace(int qlen, int tlen, int w, const CigLds &L, int *n_ops)
{
	if (cig_diag_form(qlen, tlen, w)) wave_global2_trace_diag(qlen, tlen, w, L, n_ops); else wave_global2_trace_lds(qlen, tlen, w, L, n_ops);
}
// ksw_global2 (ksw.c:540-642) as one call (bwagpu_debug_dp's entry).  Returns the score; *n_ops < 0 when the traceback does not fit CIG_TMP_OPS.
__device__ int wave_ksw_global2(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, i64 t0, int tdir, int tlen,
								int w, const CigLds &L, int *n_ops)
{
	u64 cells = 0;
	cig_build_profile(opt, q, q0, qdir, qlen, L);
	const int score = wave_global2_fill(ix, opt, qlen, t0, tdir, tlen, w, L, cells);
	wave_global2_trace(qlen, tlen, w, L, n_ops);
	return score;
}

// NM and MD (bwa.c:196-226) of an alignment whose operations lie in ops[0 .. n_ops) in traceback (reversed) order, over the sequences as
// the DP saw them (both reversed for a reverse-strand hit, in which case MD's letters are complemented: fwd == false).  The bases of an
// operation are compared 64 at a time; the characters are written by lane 0 (the deleted bases of a D by their lanes).  Returns NM;
// *md_len may exceed md_cap, in which case the string is incomplete and the caller gives the region up.
__device__ int wave_nm_md(const DevIndex &ix, const u8 *q, int q0, int qdir, i64 t0, int tdir, bool fwd, const u32 *ops, int n_ops, u8 *md, int md_cap, int *md_len)
{
	const int lane = threadIdx.x & 63;
	int x = 0, y = 0, u = 0, nm = 0, n = 0;
	auto put = [&](int c) { if (lane == 0 && n < md_cap) md[n] = (u8)c; ++n; };
	auto put_int = [&](int v) { int d = 1; while (d * 10 <= v) d *= 10; for (; d > 0; d /= 10) put('0' + v / d % 10); };
	auto letter = [&](int b) -> int { return (int)(0x54474341u >> (8 * (fwd ? b : 3 - b)) & 255u); };   // "ACGT"[b], or its complement
	int ocache = 0, obase = -64;                      // the operations, fetched 64 at a time (the long-segment kernel keeps them in HBM)
	for (int k = 0; k < n_ops; ++k) {
		if (k - obase >= 64) { obase = k; const int idx = n_ops - 1 - (k + lane); ocache = idx >= 0 ? (int)ops[idx] : 0; }
		const u32 o = (u32)__builtin_amdgcn_readlane(ocache, k - obase);
		const int op = (int)(o & 15u), len = (int)(o >> 4);
		if (op == 0) {
			for (int b = 0; b < len; b += 64) {
				const int i = b + lane; const bool in = i < len;
				const int rb = in ? ref_base(ix, t0 + (i64)(y + i) * tdir) : 0;
				const int qb = in ? (int)q[q0 + (x + i) * qdir] : 0;
				u64 mm = __ballot(in && qb != rb);
				int cur = 0;
				while (mm) {                                  // (uniform: one turn per mismatch)
					const int p = __builtin_ctzll(mm); mm &= mm - 1;
					u += p - cur; put_int(u); put(letter(__builtin_amdgcn_readlane(rb, p))); u = 0; cur = p + 1; ++nm;
				}
				u += (len - b < 64 ? len - b : 64) - cur;
			}
			x += len; y += len;
		} else if (op == 2) {
			if (k > 0 && k < n_ops - 1) {                     // (a deletion at either end is squeezed out later and does not count, bwa.c:214)
				put_int(u); put('^');
				for (int i = lane; i < len; i += 64) if (n + i < md_cap) md[n + i] = (u8)letter(ref_base(ix, t0 + (i64)(y + i) * tdir));
				n += len; u = 0; nm += len;
			}
			y += len;
		} else { x += len; nm += len; }
	}
	put_int(u);
	wave_sync();
	*md_len = n;
	return nm;
}

DEVFN int dev_infer_bw(int l1, int l2, int score, int a, int q, int r)
{	// infer_bw (bwamem.c:818-825)
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	int w = trunc_div_add((l1 < l2 ? l1 : l2) * a - score - q, r, 2);   // (int)((double)x / r + 2.), bwamem.c:822
	const int d = l1 > l2 ? l1 - l2 : l2 - l1;
	if (w < d) w = d;
	return w;
}

// One region: the band-doubling loop of mem_reg2aln (bwamem.c:1143-1152) around bwa_gen_cigar2 (bwa.c:148-195).
// DIAG_ONLY (the first tier): only the diag form is compiled in -- no {H,E} arrays in LDS, fewer registers, more waves per CU --; a band of more
// than 64 columns is deferred to the second tier like one whose directions do not fit.
template <bool DIAG_ONLY> __device__ void cigar_region(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *query, const bwagpu_alnreg_t &p, const CigLds &L, bwagpu_cigar_t *out,
							 u32 *ext, unsigned long long *ext_used, i64 ext_cap, u64 &cells, u64 &n_dp)
{
	const int lane = threadIdx.x & 63;
	const i64 rb = uni64(p.rb), re = uni64(p.re), l_pac = ix.l_pac;
	const int qb = uni(p.qb), qe = uni(p.qe), truesc = uni(p.truesc), pw = uni(p.w);
	const int l_query = qe - qb;
	int res_score = 2, res_n = -1;          // unserved records carry a reason in `score`: 1 below T, 2 shape/limits, 3 too many operations
	const bool ok_shape = l_query > 0 && rb < re && !(rb < l_pac && re > l_pac) && l_query <= CIG_MAX_LEN && re - rb <= CIG_MAX_LEN;
	if (ok_shape) {
		const int rlen = (int)(re - rb);
		const bool rev = rb >= l_pac;            // both sequences reversed so that gaps are left-aligned on the forward strand (bwa.c:163-170)
		const int q0 = rev ? qe - 1 : qb, qdir = rev ? -1 : 1;
		const i64 t0 = rev ? re - 1 : rb; const int tdir = rev ? -1 : 1;
		int tmp = dev_infer_bw(l_query, rlen, truesc, opt.a, opt.o_del, opt.e_del);
		int w2 = dev_infer_bw(l_query, rlen, truesc, opt.a, opt.o_ins, opt.e_ins);
		w2 = w2 > tmp ? w2 : tmp;
		if (w2 > opt.w) w2 = w2 < pw ? w2 : pw;
		int i = 0, score = 0, last_sc = -(1 << 30), n_ops = -1;
		bool give_up = false, defer = false, have_qp = false;
		int w_fill = -1;                                  // band of the last matrix fill, whose traceback is still owed (the reference traces every attempt back and keeps the last one's CIGAR)
		do {
			w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
			w_fill = -1;
			if (l_query == rlen && w2 == 0) {     // no gap possible: one M run, score by direct comparison (bwa.c:171-174)
				int s = 0;
				for (int j = lane; j < l_query; j += 64) s += opt.mat[ref_base(ix, t0 + (i64)j * tdir) * 5 + query[q0 + j * qdir]];
				s = wave_sum(s);
				score = s; n_ops = 1;
				if (lane == 0) L.ops[0] = (u32)l_query << 4;
				wave_sync();
			} else {
				int max_ins = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
				int max_del = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
				int max_gap = max_ins > max_del ? max_ins : max_del;
				const int dl = rlen > l_query ? rlen - l_query : l_query - rlen;
				max_gap = max_gap > 1 ? max_gap : 1;
				int w = (max_gap + dl + 1) >> 1; w = w < w2 ? w : w2;
				const int min_w = dl + 3; w = w > min_w ? w : min_w;
				const int n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
				const bool diag = cig_diag_form(l_query, rlen, w);
				const int z_need = diag ? cig_diag_cells(rlen, w) : n_col * ((rlen + 1) & ~1);      // direction nibbles of this fill
				if (!diag && n_col > CIG_MAX_COLS) { give_up = true; break; }
				if (z_need > CIG_Z_BIG) { give_up = true; break; }
				if (z_need > L.z_cells || (DIAG_ONLY && !diag)) { defer = true; break; }     // needs the second tier's LDS (or its other form)
				if (!have_qp) { cig_build_profile(opt, query, q0, qdir, l_query, L); have_qp = true; }
				score = DIAG_ONLY ? wave_global2_fill_diag(ix, opt, l_query, t0, tdir, rlen, w, L, cells) : wave_global2_fill(ix, opt, l_query, t0, tdir, rlen, w, L, cells);
				++n_dp;
				w_fill = w;
			}
			if (score == last_sc || w2 == opt.w << 2) break;
			last_sc = score;
			w2 <<= 1;
		} while (++i < 3 && score < truesc - opt.a);
		if (!give_up && !defer && w_fill >= 0) {
			if (DIAG_ONLY) wave_global2_trace_diag(l_query, rlen, w_fill, L, &n_ops); else wave_global2_trace(l_query, rlen, w_fill, L, &n_ops);
			if (n_ops < 0) give_up = true;
		}
		if (defer) res_n = -2;
		else if (!give_up) {
			int md_len = 0;
			const int nm = wave_nm_md(ix, query, q0, qdir, t0, tdir, !rev, L.ops, n_ops, L.md, CIG_MD_CAP, &md_len);
			// 7 .. CIG_TMP_OPS operations and MD strings of more than 8 characters go to the batch's operation array; the record holds their offsets
			const int n_ext_ops = n_ops > CIG_MAX_OPS ? n_ops : 0, n_ext_md = md_len > 8 ? (md_len + 3) >> 2 : 0;
			unsigned long long at = 0;
			bool fits = md_len <= CIG_MD_CAP;
			if (fits && n_ext_ops + n_ext_md > 0) {
				if (lane == 0) at = atomicAdd(ext_used, (unsigned long long)(n_ext_ops + n_ext_md));
				at = (unsigned long long)lane0_i64((i64)at);
				fits = ext && (i64)(at + n_ext_ops + n_ext_md) <= ext_cap;
			}
			if (fits) {
				for (int k = lane; k < n_ext_ops; k += 64) ext[at + k] = L.ops[n_ops - 1 - k];   // traceback order reversed
				auto md4 = [&](int w) -> u32 { u32 v = 0; for (int b = 0; b < 4; ++b) if (4 * w + b < md_len) v |= (u32)L.md[4 * w + b] << (8 * b); return v; };
				for (int w = lane; w < n_ext_md; w += 64) ext[at + n_ext_ops + w] = md4(w);
				if (lane == 0) {
					out->score = score; out->n_cigar = n_ops;
					if (n_ext_ops) { out->cigar[0] = (u32)at; out->cigar[1] = (u32)(at >> 32); for (int k = 2; k < CIG_MAX_OPS; ++k) out->cigar[k] = 0; }
					else for (int k = 0; k < CIG_MAX_OPS; ++k) out->cigar[k] = k < n_ops ? L.ops[n_ops - 1 - k] : 0;
					out->nm = nm; out->md_len = md_len;
					out->md = n_ext_md ? (u64)(at + n_ext_ops) : ((u64)md4(1) << 32 | md4(0));
				}
				wave_sync();
				return;
			}
			res_score = 3;
		}
	}
	if (lane == 0) {
		out->score = res_score; out->n_cigar = res_n; out->nm = -1; out->md_len = 0; out->md = 0;
		for (int k = 0; k < CIG_MAX_OPS; ++k) out->cigar[k] = 0;
	}
	wave_sync();
}

// ---- long segments (BASELINE configs[4]: 10 kb reads) --------------------------------------------------------------------------------
// The same recurrence for segments of any length: the band's {H, E} columns live in an LDS ring (as in k_dedup_wave's score-only form), scores
// come from the 25-entry matrix and the query bases, and the direction codes -- one byte per cell, rows padded to 16 bytes -- go to a
// scratch area of the wave in HBM (ksw.c:548-549 keeps the same matrix in host memory).  The traceback (ksw.c:624-639) is a chain of
// ~2 (qlen + tlen) dependent one-byte look-ups; left to one lane reading HBM it would cost a memory round trip per step, so the wave
// fetches the path's neighbourhood a tile at a time -- the 48 bytes around the diagonal's column in each of 64 rows -- into LDS and lane 0
// walks the tile; a path that drifts out of a row's window (more than 16 gap columns within 64 rows) just ends the tile early.  Operations
// are pushed to the wave's HBM scratch as they complete, in traceback order.
#define CIGL_MAX_COLS 1900      // widest band (columns per row)
#define CIGL_RING 2048          // ring of {H,E} columns: >= 2 w + 132 for every band the kernel takes (2 w + 1 <= CIGL_MAX_COLS)
#define CIGL_MAX_OPS 32768
#define CIGL_MD_CAP 98304
#define CIGL_TILE_W 48
#define CIGL_QCAP 16384         // query bases kept in LDS, in alignment order (longer segments read theirs from the batch)
#define CIGL_LDS_BYTES (2 * CIGL_RING * 4 + 32 + 64 * CIGL_TILE_W + CIGL_QCAP + 16)
struct CigLongLds { i32 *hd, *e; int8_t *mat; u8 *tile; u8 *qs; };
struct CigLongScratch { u8 *z; i64 z_cap; u32 *ops; u8 *md; };

DEVFN void cigl_lds_setup(unsigned char *lds, CigLongLds &L)
{
	L.hd = (i32*)lds; L.e = L.hd + CIGL_RING;
	L.mat = (int8_t*)(L.e + CIGL_RING);
	L.tile = (u8*)(lds + 2 * CIGL_RING * 4 + 32);
	L.qs = L.tile + 64 * CIGL_TILE_W;
}

// The segment's query bases into LDS, in alignment order (column j's base at qs[j]); false: too long, the DP reads the batch's array.
DEVFN bool cigl_stage_query(const CigLongLds &L, const u8 *q, int q0, int qdir, int qlen)
{
	if (qlen > CIGL_QCAP) return false;
	const int lane = threadIdx.x & 63;
	for (int j = lane; j < qlen; j += 64) L.qs[j] = q[q0 + j * qdir];
	wave_sync();
	return true;
}

// The matrix fill (ksw.c:566-619): direction bytes of every band cell to S.z, the score returned.  q_lds: the query is staged (cigl_stage_query).
__device__ int wave_ksw_global2_long_fill(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, i64 t0, int tdir, int tlen,
										  int w, const CigLongLds &L, const CigLongScratch &S, bool q_lds)
{
	const int lane = threadIdx.x & 63;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1, zs = (n_col + 15) & ~15;       // row stride of the direction matrix
	i32 *hd = L.hd, *e_ = L.e; const int rm = CIGL_RING - 1;
	int init_hi = -1, treg = 0;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) { int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		{	// first-row values (ksw.c:566-570) of the columns this row can reach for the first time
			const int hi = i + w + 2 < qlen ? i + w + 2 : qlen;
			if (hi > init_hi) {
				for (int j = init_hi + 1 + lane; j <= hi; j += 64) {
					hd[j & rm] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : CIG_NEG_INF);
					e_[j & rm] = CIG_NEG_INF;
				}
				init_hi = hi;
				wave_sync();
			}
		}
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : CIG_NEG_INF;
		int carry = I32_MIN, bnd = 0;
		u8 *zrow = S.z + (i64)i * zs;
		for (int b = beg; b < end; b += 64) {
			const int j = b + lane; const bool act = j < end;
			int dg = hd[j & rm]; const int ec = e_[j & rm];
			const int qc = j < qlen ? (q_lds ? (int)L.qs[j] : (int)q[q0 + j * qdir]) : 4;
			const int sc = L.mat[tb * 5 + qc];
			const int bnd_next = hd[(b + 64) & rm];
			if (b != beg && lane == 0) dg = bnd;
			wave_sync();
			const int m = dg + sc;
			const int a = act ? m - oe_ins + j * e_ins : I32_MIN;
			const int inc = wave_incl_scan_max(a);
			const int exc = imax(wave_shift_up1(inc, I32_MIN), carry);
			int f = CIG_NEG_INF - (j - beg) * e_ins;
			if (j > beg && act) f = imax(f, exc - (j - 1) * e_ins);
			int d = m >= ec ? 0 : 1, h = m >= ec ? m : ec;             // ksw.c:587-590
			if (h < f) { d = 2; h = f; }
			int t = m - oe_del, en = ec - e_del;
			if (en > t) d |= 4; else en = t;                          // E continues (ksw.c:592-595)
			t = m - oe_ins;
			if (f - e_ins > t) d |= 8;                                // F continues (ksw.c:596-599)
			if (act) { e_[j & rm] = en; hd[(j + 1) & rm] = h; zrow[j - beg] = (u8)d; }
			if (b == beg && lane == 0) hd[beg & rm] = h1_init;
			carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
			bnd = bnd_next;
			wave_sync();
		}
		if (lane == 0) e_[end & rm] = CIG_NEG_INF;
		wave_sync();
	}
	const int score = hd[qlen & rm];
	__threadfence();                                              // the direction bytes are read back by the traceback through other lanes
	wave_sync();
	return score;
}

// The traceback (ksw.c:624-639) over the direction bytes the fill left in S.z for the same (qlen, tlen, w); operations to S.ops in traceback order.
__device__ void wave_ksw_global2_long_trace(int qlen, int tlen, int w, const CigLongLds &L, const CigLongScratch &S, int *n_ops)
{
	const int lane = threadIdx.x & 63;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1, zs = (n_col + 15) & ~15;
	int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0, n = 0;
	int cur_op = -1, cur_len = 0;                                 // the run being built (lane 0)
	while (i >= 0 && k >= 0 && n <= CIGL_MAX_OPS) {
		// tile: row i - r for lane r, the CIGL_TILE_W bytes from the 16-byte boundary at or below (column the diagonal through (i,k) has in that row) - 16
		{
			const int ii = i - lane;
			if (ii >= 0) {
				const int bi = ii > w ? ii - w : 0;
				int ws = ((k - lane) - bi - 16) & ~15; if (ws < 0) ws = 0;
				const uint4 *src = (const uint4*)(S.z + (i64)ii * zs + ws);
				uint4 *dst = (uint4*)(L.tile + lane * CIGL_TILE_W);
#pragma unroll
				for (int p = 0; p < CIGL_TILE_W / 16; ++p) if (ws + 16 * p < zs) dst[p] = src[p];
			}
		}
		wave_sync();
		if (lane == 0) {
			const int i0 = i, k0 = k;
			while (i >= 0 && k >= 0 && i0 - i < 64 && n <= CIGL_MAX_OPS) {
				const int r = i0 - i, bi = i > w ? i - w : 0;
				int ws = ((k0 - r) - bi - 16) & ~15; if (ws < 0) ws = 0;
				const int c = (k - bi) - ws;
				if (c < 0 || c >= CIGL_TILE_W) break;                  // drifted out of this row's window: start a new tile here
				const u32 dd = L.tile[r * CIGL_TILE_W + c];
				which = which == 0 ? (int)(dd & 3) : which == 1 ? (int)(dd >> 2 & 1) : (int)(dd >> 3 & 1) << 1;
				const int op = which == 0 ? 0 : (which == 1 ? 2 : 1);
				if (op == cur_op) ++cur_len;
				else { if (cur_op >= 0) { if (n < CIGL_MAX_OPS) S.ops[n] = (u32)cur_len << 4 | (u32)cur_op; ++n; } cur_op = op; cur_len = 1; }
				if (which == 0) { --i; --k; } else if (which == 1) --i; else --k;
			}
		}
		i = __builtin_amdgcn_readlane(i, 0); k = __builtin_amdgcn_readlane(k, 0); n = __builtin_amdgcn_readlane(n, 0);
		wave_sync();
	}
	if (lane == 0) {
		auto push = [&](int op, int len) {
			if (op == cur_op) cur_len += len;
			else { if (cur_op >= 0) { if (n < CIGL_MAX_OPS) S.ops[n] = (u32)cur_len << 4 | (u32)cur_op; ++n; } cur_op = op; cur_len = len; }
		};
		if (i >= 0) push(2, i + 1);
		if (k >= 0) push(1, k + 1);
		if (cur_op >= 0) { if (n < CIGL_MAX_OPS) S.ops[n] = (u32)cur_len << 4 | (u32)cur_op; ++n; }
	}
	n = __builtin_amdgcn_readlane(n, 0);
	__threadfence();
	wave_sync();
	*n_ops = n > CIGL_MAX_OPS ? -1 : n;
}

__device__ int wave_ksw_global2_long(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, i64 t0, int tdir, int tlen,
									 int w, const CigLongLds &L, const CigLongScratch &S, int *n_ops)
{
	const bool q_lds = cigl_stage_query(L, q, q0, qdir, qlen);
	const int score = wave_ksw_global2_long_fill(ix, opt, q, q0, qdir, qlen, t0, tdir, tlen, w, L, S, q_lds);
	wave_ksw_global2_long_trace(qlen, tlen, w, L, S, n_ops);
	return score;
}

// One region of the long tier: mem_reg2aln's band-doubling loop (bwamem.c:1143-1152) around the routine above, then NM / MD and the record.
__device__ void cigar_region_long(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *query, const bwagpu_alnreg_t &p, const CigLongLds &L, const CigLongScratch &S,
								  bwagpu_cigar_t *out, u32 *ext, unsigned long long *ext_used, i64 ext_cap)
{
	const int lane = threadIdx.x & 63;
	const i64 rb = uni64(p.rb), re = uni64(p.re), l_pac = ix.l_pac;
	const int qb = uni(p.qb), qe = uni(p.qe), truesc = uni(p.truesc), pw = uni(p.w);
	const int l_query = qe - qb;
	if (!(l_query > 0 && rb < re && !(rb < l_pac && re > l_pac) && re - rb < (1 << 24) && l_query < (1 << 24))) return;     // (the record keeps its "not computed" state)
	const int rlen = (int)(re - rb);
	const bool rev = rb >= l_pac;
	const int q0 = rev ? qe - 1 : qb, qdir = rev ? -1 : 1;
	const i64 t0 = rev ? re - 1 : rb; const int tdir = rev ? -1 : 1;
	int tmp = dev_infer_bw(l_query, rlen, truesc, opt.a, opt.o_del, opt.e_del);
	int w2 = dev_infer_bw(l_query, rlen, truesc, opt.a, opt.o_ins, opt.e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt.w) w2 = w2 < pw ? w2 : pw;
	int i = 0, score = 0, last_sc = -(1 << 30), n_ops = -1;
	int w_fill = -1;                                              // band of the last matrix fill whose traceback is still owed
	const bool q_lds = cigl_stage_query(L, query, q0, qdir, l_query);
	do {
		w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
		w_fill = -1;
		if (l_query == rlen && w2 == 0) {
			int s = 0;
			for (int j = lane; j < l_query; j += 64) s += opt.mat[ref_base(ix, t0 + (i64)j * tdir) * 5 + query[q0 + j * qdir]];
			s = wave_sum(s);
			score = s; n_ops = 1;
			if (lane == 0) S.ops[0] = (u32)l_query << 4;
			__threadfence();
			wave_sync();
		} else {
			int max_ins = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
			int max_del = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
			int max_gap = max_ins > max_del ? max_ins : max_del;
			const int dl = rlen > l_query ? rlen - l_query : l_query - rlen;
			max_gap = max_gap > 1 ? max_gap : 1;
			int w = (max_gap + dl + 1) >> 1; w = w < w2 ? w : w2;
			const int min_w = dl + 3; w = w > min_w ? w : min_w;
			const int n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
			if (n_col > CIGL_MAX_COLS || (i64)rlen * ((n_col + 15) & ~15) > S.z_cap) return;
			score = wave_ksw_global2_long_fill(ix, opt, query, q0, qdir, l_query, t0, tdir, rlen, w, L, S, q_lds);
			w_fill = w;                                           // (the reference traces every attempt back; only the last one's CIGAR is used)
		}
		if (score == last_sc || w2 == opt.w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < truesc - opt.a);
	if (w_fill >= 0) {
		wave_ksw_global2_long_trace(l_query, rlen, w_fill, L, S, &n_ops);
		if (n_ops < 0) return;
	}
	int md_len = 0;
	const int nm = wave_nm_md(ix, query, q0, qdir, t0, tdir, !rev, S.ops, n_ops, S.md, CIGL_MD_CAP, &md_len);
	if (md_len > CIGL_MD_CAP) return;
	__threadfence();
	wave_sync();
	const int n_ext_ops = n_ops > CIG_MAX_OPS ? n_ops : 0, n_ext_md = md_len > 8 ? (md_len + 3) >> 2 : 0;
	unsigned long long at = 0;
	bool fits = true;
	if (n_ext_ops + n_ext_md > 0) {
		if (lane == 0) at = atomicAdd(ext_used, (unsigned long long)(n_ext_ops + n_ext_md));
		at = (unsigned long long)lane0_i64((i64)at);
		fits = ext && (i64)(at + n_ext_ops + n_ext_md) <= ext_cap;
	}
	if (!fits) { if (lane == 0) out->score = 3; wave_sync(); return; }
	for (int k = lane; k < n_ext_ops; k += 64) ext[at + k] = S.ops[n_ops - 1 - k];
	auto md4 = [&](int w_) -> u32 { u32 v = 0; for (int b = 0; b < 4; ++b) if (4 * w_ + b < md_len) v |= (u32)S.md[4 * w_ + b] << (8 * b); return v; };
	for (int w_ = lane; w_ < n_ext_md; w_ += 64) ext[at + n_ext_ops + w_] = md4(w_);
	if (lane == 0) {
		out->score = score; out->n_cigar = n_ops;
		if (n_ext_ops) { out->cigar[0] = (u32)at; out->cigar[1] = (u32)(at >> 32); for (int k = 2; k < CIG_MAX_OPS; ++k) out->cigar[k] = 0; }
		else for (int k = 0; k < CIG_MAX_OPS; ++k) out->cigar[k] = k < n_ops ? S.ops[n_ops - 1 - k] : 0;
		out->nm = nm; out->md_len = md_len;
		out->md = n_ext_md ? (u64)(at + n_ext_ops) : ((u64)md4(1) << 32 | md4(0));
	}
	wave_sync();
}

// Sizing pass of the third tier: list[0 .. plan[0]) = the regions the LDS tiers left uncomputed, plan[1] = the largest direction matrix (bytes)
// any of them can ask for -- the widest band cigar_region_long's loop can reach is max(min((max_gap + dl + 1) / 2, 4 opt.w), dl + 3).
__global__ void __launch_bounds__(256) k_cigar_long_plan(DevIndex ix, bwagpu_opt_t opt, i64 n_regs, const bwagpu_alnreg_t *regs, const bwagpu_cigar_t *out, unsigned long long *plan, i32 *list)
{
	unsigned long long need = 0;
	for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n_regs; g += (i64)gridDim.x * blockDim.x) {
		if (!(out[g].n_cigar < 0 && out[g].score != 1)) continue;
		const bwagpu_alnreg_t &p = regs[g];
		const int l_query = p.qe - p.qb;
		if (!(l_query > 0 && p.rb < p.re && !(p.rb < ix.l_pac && p.re > ix.l_pac) && p.re - p.rb < (1 << 24) && l_query < (1 << 24))) continue;
		const int rlen = (int)(p.re - p.rb);
		int max_ins = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
		int max_del = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
		int max_gap = max_ins > max_del ? max_ins : max_del; max_gap = max_gap > 1 ? max_gap : 1;
		const int dl = rlen > l_query ? rlen - l_query : l_query - rlen;
		int w = (max_gap + dl + 1) >> 1; w = w < opt.w << 2 ? w : opt.w << 2;
		w = w > dl + 3 ? w : dl + 3;
		int n_col = l_query < 2 * w + 1 ? l_query : 2 * w + 1;
		list[atomicAdd(plan, 1ull)] = (i32)g;                     // the tier's work list (a handful per short-read batch, every region of a long-read one)
		if (n_col > CIGL_MAX_COLS) n_col = (dl + 3) * 2 + 1 <= CIGL_MAX_COLS ? CIGL_MAX_COLS : 0;      // (attempts over the limit leave the region to the host)
		const unsigned long long b = (unsigned long long)rlen * (unsigned long long)((n_col + 15) & ~15);
		need = b > need ? b : need;
	}
	if (need) atomicMax(plan + 1, need);
}

// Third tier of bwagpu_batch_cigars: one wavefront per region the LDS tiers left uncomputed for their limits (segments over CIG_MAX_LEN bases,
// bands over CIG_MAX_COLS columns, more than CIG_TMP_OPS operations), drawn one at a time from k_cigar_long_plan's list.  64 threads per
// workgroup; scratch: one CigLongScratch per workgroup.
__global__ void __launch_bounds__(64) k_cigar_long(DevIndex ix, bwagpu_opt_t opt, Batch B, i64 n_list, const i32 *list, const bwagpu_alnreg_t *regs, const i32 *reg_read, bwagpu_cigar_t *out,
													unsigned long long *next, u8 *z_all, i64 z_cap, u32 *ops_all, u8 *md_all, u32 *ext, unsigned long long *ext_used, i64 ext_cap)
{
	HIP_DYNAMIC_SHARED(unsigned char, cigl_lds)
	const int lane = threadIdx.x & 63;
	CigLongLds L;
	cigl_lds_setup(cigl_lds, L);
	if (lane < 25) L.mat[lane] = opt.mat[lane];
	CigLongScratch S;
	S.z = z_all + (i64)blockIdx.x * z_cap; S.z_cap = z_cap; S.ops = ops_all + (size_t)blockIdx.x * CIGL_MAX_OPS; S.md = md_all + (size_t)blockIdx.x * CIGL_MD_CAP;
	wave_sync();
	for (;;) {
		const long long k = wave_fetch(next);
		if (k >= n_list) break;
		const long long gg = uni(list[k]);
		const bwagpu_alnreg_t p = regs[gg];
		cigar_region_long(ix, opt, B.seq + B.off[reg_read[gg]], p, L, S, out + gg, ext, ext_used, ext_cap);
	}
}

// One wavefront per packed region (bwagpu_batch_download's order); regions below the output threshold T are skipped.
// tier 0 visits every region and defers (n_cigar = -2) those whose band needs more LDS than z_cells; tier 1 redoes exactly those.
// best_of: null, or for every read the index of its first (best-scoring) packed region: with it, regions that overlap the read's best region
// and score below XA_drop_ratio times its score are left uncomputed (reason 1) -- the finalize stage hardly ever asks for them (they are
// neither a line of their own nor within reach of an XA list, bwamem_extra.c:118-134), and should it ask, it computes them itself.  They are the
// expensive ones: diverged repeat copies whose low score means a wide band (bwamem.c:818-825).
template <bool DIAG_ONLY> __global__ void __launch_bounds__(256, DIAG_ONLY ? 5 : 2) k_cigar(DevIndex ix, bwagpu_opt_t opt, Batch B, i64 n_regs, const bwagpu_alnreg_t *regs, const i32 *reg_read, bwagpu_cigar_t *out,
											   unsigned long long *next, int z_cells, int tier, u32 *ext, unsigned long long *ext_used, i64 ext_cap, const i64 *best_of)
{
	HIP_DYNAMIC_SHARED(unsigned char, cig_lds)
	const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned char *base = cig_lds + (size_t)wave_in_blk * (DIAG_ONLY ? CIG_LDS_BYTES_DIAG(z_cells) : CIG_LDS_BYTES(z_cells));
	CigLds L;
	L.hd = (i32*)base; L.e = L.hd + (CIG_MAX_LEN + 2 + 64);
	L.qstride = CIG_MAX_LEN + 64; L.z_cells = z_cells;
	L.qp = DIAG_ONLY ? (int8_t*)base : (int8_t*)(L.e + (CIG_MAX_LEN + 2 + 64));      // (the diag form keeps no {H,E} columns in LDS)
	L.z = (u8*)(L.qp + 5 * L.qstride);
	L.ops = (u32*)(L.z + z_cells / 2 + CIG_MAX_COLS);
	L.md = (u8*)(L.ops + CIG_TMP_OPS);
	WaveQueue wq; wq_init(wq);
	u64 cells = 0, n_dp = 0;
	for (;;) {
		long long g;
		if (!wq_next(wq, next, n_regs, g)) break;
		if (tier > 0 && uni(out[g].n_cigar) != -2) continue;
		const bwagpu_alnreg_t p = regs[g];
		if (p.score < opt.T) { if (lane == 0) { out[g].score = 1; out[g].n_cigar = -1; out[g].nm = -1; out[g].md_len = 0; out[g].md = 0; for (int k = 0; k < CIG_MAX_OPS; ++k) out[g].cigar[k] = 0; } continue; }
		const int r = reg_read[g];
		if (best_of) {
			const bwagpu_alnreg_t &b = regs[best_of[r]];
			const int lo = p.qb > b.qb ? p.qb : b.qb, hi = p.qe < b.qe ? p.qe : b.qe, ml = p.qe - p.qb < b.qe - b.qb ? p.qe - p.qb : b.qe - b.qb;
			if (hi > lo && hi - lo >= ml * opt.mask_level && p.score < b.score * opt.XA_drop_ratio) {
				if (lane == 0) { out[g].score = 1; out[g].n_cigar = -1; out[g].nm = -1; out[g].md_len = 0; out[g].md = 0; for (int k = 0; k < CIG_MAX_OPS; ++k) out[g].cigar[k] = 0; }
				continue;
			}
		}
		cigar_region<DIAG_ONLY>(ix, opt, B.seq + B.off[r], p, L, out + g, ext, ext_used, ext_cap, cells, n_dp);
		if (tier > 0 && lane == 0 && out[g].n_cigar == -2) out[g].n_cigar = -1;
	}
	if (B.stats && lane == 0) { atomicAdd(&B.ctr->glb_cells, (unsigned long long)cells); atomicAdd(&B.ctr->glb_calls, (unsigned long long)n_dp); }      // (bwagpu_batch_cigars zeroes the two counters first: the batch's run is over)
}
