/* oracle/orc_ksw.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * Scalar restatements of the reference's banded DP kernels. */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

/* ksw_extend2 (ksw.c:416-515): banded affine-gap extension from score h0 with z-drop.
 * Row i = target base, column j = query base.  Hd[j] holds H(i-1,j-1) and E[j] holds E(i,j) when row i
 * starts.  Gaps open from the diagonal term M, not from H (ksw.c:469-483).  The two arrays are never
 * cleared between rows, which is what makes the "stale cell" rule of SURVEY.md App. A.10 come out. */
int orc_ksw_extend2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
					int o_del, int e_del, int o_ins, int e_ins, int w, int end_bonus, int zdrop, int h0,
					int *qle, int *tle, int *gtle, int *gscore_, int *max_off_)
{
	int32_t *Hd = (int32_t*)calloc(qlen + 2, 4), *E = (int32_t*)calloc(qlen + 2, 4);
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int i, j, k, beg = 0, end = qlen, max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, mmax = 0, lim;
	Hd[0] = h0;
	if (qlen >= 1) Hd[1] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && Hd[j-1] > e_ins; ++j) Hd[j] = Hd[j-1] - e_ins;
	for (k = 0; k < m * m; ++k) if (mat[k] > mmax) mmax = mat[k];
	lim = (int)((double)(qlen * mmax + end_bonus - o_ins) / e_ins + 1.); if (lim < 1) lim = 1; if (w > lim) w = lim;
	lim = (int)((double)(qlen * mmax + end_bonus - o_del) / e_del + 1.); if (lim < 1) lim = 1; if (w > lim) w = lim;
	for (i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * m;
		int f = 0, h1, rowmax = 0, rowmax_j = -1;
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
		for (j = beg; j < end; ++j) {
			int M = Hd[j], e = E[j], h, t;
			Hd[j] = h1;
			M = M ? M + srow[query[j]] : 0;
			h = M > e ? M : e; if (f > h) h = f;
			h1 = h;
			if (h >= rowmax) { rowmax_j = j; rowmax = h; } /* last column wins ties (ksw.c:473-474) */
			t = M - oe_del; if (t < 0) t = 0; e -= e_del; E[j] = e > t ? e : t;
			t = M - oe_ins; if (t < 0) t = 0; f -= e_ins; if (t > f) f = t;
		}
		Hd[end] = h1; E[end] = 0;
		if (j == qlen) { if (h1 >= gscore) max_ie = i; if (h1 > gscore) gscore = h1; }
		if (rowmax == 0) break;
		if (rowmax > max) {
			int off = rowmax_j - i; if (off < 0) off = -off;
			max = rowmax; max_i = i; max_j = rowmax_j;
			if (off > max_off) max_off = off;
		} else if (zdrop > 0) {
			int di = i - max_i, dj = rowmax_j - max_j;
			if (di > dj) { if (max - rowmax - (di - dj) * e_del > zdrop) break; }
			else if (max - rowmax - (dj - di) * e_ins > zdrop) break;
		}
		for (j = beg; j < end && Hd[j] == 0 && E[j] == 0; ++j) {}
		beg = j;
		for (j = end; j >= beg && Hd[j] == 0 && E[j] == 0; --j) {}
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	free(Hd); free(E);
	if (qle) *qle = max_j + 1;
	if (tle) *tle = max_i + 1;
	if (gtle) *gtle = max_ie + 1;
	if (gscore_) *gscore_ = gscore;
	if (max_off_) *max_off_ = max_off;
	return max;
}

#define NEG_INF (-0x40000000)

static void cigar_push(uint32_t **c, int *n, int *cap, int op, int len)
{
	if (*n && ((*c)[*n - 1] & 0xf) == (uint32_t)op) { (*c)[*n - 1] += (uint32_t)len << 4; return; }
	if (*n == *cap) { *cap = *cap ? *cap << 1 : 4; *c = (uint32_t*)realloc(*c, *cap * 4); }
	(*c)[(*n)++] = (uint32_t)len << 4 | op;
}

/* ksw_global2 (ksw.c:540-642): banded Needleman-Wunsch, |i-j| <= w, optional traceback.
 * Direction byte per cell: bits 0-1 = source of H (0 diag, 1 E, 2 F), bit 2 = E continues a deletion,
 * bit 5 = F continues an insertion (ksw.c:587-600).  Ties prefer M over E over F. */
int orc_ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
					int o_del, int e_del, int o_ins, int e_ins, int w, int *n_cigar_, uint32_t **cigar_)
{
	int32_t *Hd = (int32_t*)malloc((qlen + 2) * 4), *E = (int32_t*)malloc((qlen + 2) * 4);
	int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, i, j, score;
	int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1, want_tb = n_cigar_ && cigar_;
	uint8_t *z = want_tb ? (uint8_t*)malloc((size_t)n_col * tlen + 1) : 0;
	if (n_cigar_) *n_cigar_ = 0;
	Hd[0] = 0; E[0] = NEG_INF;
	for (j = 1; j <= qlen && j <= w; ++j) { Hd[j] = -(o_ins + e_ins * j); E[j] = NEG_INF; }
	for (; j <= qlen; ++j) Hd[j] = E[j] = NEG_INF;
	for (i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * m;
		int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int32_t f = NEG_INF, h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		uint8_t *zi = z ? z + (size_t)i * n_col : 0;
		for (j = beg; j < end; ++j) {
			int32_t mm = Hd[j] + srow[query[j]], e = E[j], h, t; uint8_t d;
			Hd[j] = h1;
			d = mm >= e ? 0 : 1; h = mm >= e ? mm : e;
			if (h < f) { d = 2; h = f; }
			h1 = h;
			t = mm - oe_del; e -= e_del; if (e > t) d |= 1 << 2; else e = t; E[j] = e;
			t = mm - oe_ins; f -= e_ins; if (f > t) d |= 2 << 4; else f = t;
			if (zi) zi[j - beg] = d;
		}
		Hd[end] = h1; E[end] = NEG_INF;
	}
	score = Hd[qlen];
	if (want_tb) {
		uint32_t *cg = 0; int n = 0, cap = 0, which = 0, k;
		i = tlen - 1; k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { cigar_push(&cg, &n, &cap, 0, 1); --i; --k; }
			else if (which == 1) { cigar_push(&cg, &n, &cap, 2, 1); --i; }
			else { cigar_push(&cg, &n, &cap, 1, 1); --k; }
		}
		if (i >= 0) cigar_push(&cg, &n, &cap, 2, i + 1);
		if (k >= 0) cigar_push(&cg, &n, &cap, 1, k + 1);
		for (i = 0; i < n >> 1; ++i) { uint32_t t = cg[i]; cg[i] = cg[n-1-i]; cg[n-1-i] = t; }
		*n_cigar_ = n; *cigar_ = cg;
	}
	free(Hd); free(E); free(z);
	return score;
}

/* Score of ksw_align2(..., KSW_XSTART, 0) as used by mem_seed_sw (bwamem.c:619): plain Gotoh local
 * alignment with gaps opened from H and everything clamped at 0; equal to the striped SSE2 kernel
 * ksw_i16 (ksw.c:255-377) in score. */
int orc_ksw_local_score(int qlen, const uint8_t *query, int tlen, const uint8_t *target, int m, const int8_t *mat,
						int o_del, int e_del, int o_ins, int e_ins)
{
	int32_t *H = (int32_t*)calloc(qlen + 1, 4), *E = (int32_t*)calloc(qlen + 1, 4);
	int i, j, best = 0, oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	for (i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * m;
		int32_t f = 0, hdiag = 0; /* H(i-1,-1) = 0 */
		for (j = 0; j < qlen; ++j) {
			int32_t h = hdiag + srow[query[j]], e = E[j], t;
			hdiag = H[j];
			if (h < e) h = e;
			if (h < f) h = f;
			if (h < 0) h = 0;
			H[j] = h;
			if (h > best) best = h;
			t = h - oe_del; if (t < 0) t = 0; e -= e_del; if (e < 0) e = 0; E[j] = e > t ? e : t;
			t = h - oe_ins; if (t < 0) t = 0; f -= e_ins; if (f < 0) f = 0; if (t > f) f = t;
		}
	}
	free(H); free(E);
	return best;
}
