// host_ksw.cpp -- DP kernels of the host finalize stage, restated from the reference's semantics.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "bwamem_host.h"

namespace hostmem {

static const int NEG_INF = -0x40000000;

static inline void push_op(std::vector<uint32_t> &c, int op, int len)
{
	if (!c.empty() && (c.back() & 0xf) == (uint32_t)op) c.back() += (uint32_t)len << 4;
	else c.push_back((uint32_t)len << 4 | (uint32_t)op);
}

// ksw_global2 (ksw.c:540-642): banded Needleman-Wunsch |i-j| <= w.  Direction byte: bits 0-1 source of H (0 diagonal,
// 1 E, 2 F; ties prefer M then E), bit 2 = E continues a deletion, bit 5 = F continues an insertion.  Traceback state
// machine at ksw.c:624-639.  CIGAR ops M=0 I=1 D=2.
int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w, std::vector<uint32_t> *cigar)
{
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int n_col = qlen < 2 * w + 1 ? qlen : 2 * w + 1;
	std::vector<int32_t> Hd(qlen + 2), E(qlen + 2);
	std::vector<uint8_t> z;
	if (cigar) { cigar->clear(); z.resize((size_t)n_col * tlen + 1); }
	Hd[0] = 0; E[0] = NEG_INF;
	int j;
	for (j = 1; j <= qlen && j <= w; ++j) { Hd[j] = -(o_ins + e_ins * j); E[j] = NEG_INF; }
	for (; j <= qlen; ++j) Hd[j] = E[j] = NEG_INF;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int32_t f = NEG_INF, h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : NEG_INF;
		uint8_t *zi = cigar ? &z[(size_t)i * n_col] : nullptr;
		for (j = beg; j < end; ++j) {
			int32_t m = Hd[j] + srow[query[j]], e = E[j], h, t; uint8_t d;
			Hd[j] = h1;
			d = m >= e ? 0 : 1; h = m >= e ? m : e;
			if (h < f) { d = 2; h = f; }
			h1 = h;
			t = m - oe_del; e -= e_del; if (e > t) d |= 1 << 2; else e = t; E[j] = e;
			t = m - oe_ins; f -= e_ins; if (f > t) d |= 2 << 4; else f = t;
			if (zi) zi[j - beg] = d;
		}
		Hd[end] = h1; E[end] = NEG_INF;
	}
	const int score = Hd[qlen];
	if (cigar) {
		int i = tlen - 1, k = (i + w + 1 < qlen ? i + w + 1 : qlen) - 1, which = 0;
		while (i >= 0 && k >= 0) {
			which = z[(size_t)i * n_col + (k - (i > w ? i - w : 0))] >> (which << 1) & 3;
			if (which == 0) { push_op(*cigar, 0, 1); --i; --k; }
			else if (which == 1) { push_op(*cigar, 2, 1); --i; }
			else { push_op(*cigar, 1, 1); --k; }
		}
		if (i >= 0) push_op(*cigar, 2, i + 1);
		if (k >= 0) push_op(*cigar, 1, k + 1);
		std::reverse(cigar->begin(), cigar->end());
	}
	return score;
}

// ---- ksw_align2 (ksw.c:379-400) over ksw_u8 (:122-248) / ksw_i16 (:255-377) -----------------------------------------
// The reference evaluates a Gotoh local alignment (gaps open from H, cells clamped at 0) with Farrar's striped SSE2 layout.
// What survives of that layout in the *results* and is reproduced here:
//   * the query is padded to a multiple of the vector width p (16 cells for the byte kernel, 8 for the 16-bit one); pad
//     columns score 0 against everything, so they echo H diagonally and take part in the per-row maximum that drives the
//     second-best bookkeeping (array b[], ksw.c:215-223) -- they can never raise the best score;
//   * te = first target row reaching the best score, qe = smallest query index holding it in that row (:237-239);
//   * score2/te2 = best row-run maximum farther than ceil(score/max_mat) rows from te (:241-248);
//   * byte kernel: stop once best + shift >= 255 (:228) and report 255;
//   * KSW_XSTOP: stop once best >= threshold;  KSW_XSTART: second pass on the reversed prefixes (:392-399).
// The lazy-F evaluation order does not change H (every insertion->deletion path has an equal-score deletion->insertion twin).
enum { XBYTE = 0x10000, XSTOP = 0x20000, XSUBO = 0x40000, XSTART = 0x80000 };

static KswResult sw_core(int size, int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	const int p = size == 1 ? 16 : 8, slen = (qlen + p - 1) / p, qpad = slen * p;
	int mn = 127, mx = 0;
	for (int a = 0; a < 25; ++a) { if (mat[a] < mn) mn = mat[a]; if (mat[a] > mx) mx = mat[a]; }
	const int shift = (uint8_t)(256 - mn);                       // kswq_t::shift is a uint8_t (ksw.c:91)
	const int minsc = (xtra & XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & XSTOP) ? (xtra & 0xffff) : 0x10000;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	const int cap = size == 1 ? 255 : 32767;
	std::vector<int> H(qpad, 0), E(qpad, 0), Hn(qpad, 0), Hmax(qpad, 0);
	std::vector<std::pair<int, int>> b;                            // (row maximum, row)
	KswResult r = {0, -1, -1, -1, -1, -1, -1};
	int gmax = 0, te = -1;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = mat + target[i] * 5;
		int f = 0, hdiag = 0, imax = 0;
		for (int j = 0; j < qpad; ++j) {
			int h = hdiag + (j < qlen ? srow[query[j]] : 0), e = E[j], t;
			if (h < 0) h = 0;
			if (h > cap) h = cap;
			hdiag = H[j];
			if (h < e) h = e;
			if (h < f) h = f;
			Hn[j] = h;
			if (h > imax) imax = h;
			e -= e_del; if (e < 0) e = 0; t = h - oe_del; if (t < 0) t = 0; E[j] = e > t ? e : t;
			f -= e_ins; if (f < 0) f = 0; t = h - oe_ins; if (t < 0) t = 0; if (t > f) f = t;
		}
		if (imax >= minsc) {
			if (b.empty() || b.back().second + 1 != i) b.push_back(std::make_pair(imax, i));
			else if (b.back().first < imax) b.back() = std::make_pair(imax, i);
		}
		H.swap(Hn);
		if (imax > gmax) {
			gmax = imax; te = i; Hmax = H;
			if ((size == 1 && gmax + shift >= 255) || gmax >= endsc) break;
		}
	}
	r.score = (size == 1 && gmax + shift >= 255) ? 255 : gmax;
	r.te = te;
	if (!(size == 1 && r.score == 255)) {
		int best = -1;
		for (int j = 0; j < qpad; ++j) if (Hmax[j] > best) { best = Hmax[j]; r.qe = j; }   // smallest index holding the maximum
		if (!b.empty()) {
			int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
			for (auto &x : b) if ((x.second < low || x.second > high) && x.first > r.score2) { r.score2 = x.first; r.te2 = x.second; }
		}
	}
	return r;
}

KswResult ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra)
{
	const int size = (xtra & XBYTE) ? 1 : 2;
	KswResult r = sw_core(size, qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, xtra);
	if (!(xtra & XSTART) || ((xtra & XSUBO) && r.score < (xtra & 0xffff))) return r;
	// start positions: align the reversed prefixes until the same score is reached (ksw.c:392-399)
	std::vector<uint8_t> q2(query, query + r.qe + 1), t2(target, target + tlen);
	std::reverse(q2.begin(), q2.end());
	std::reverse(t2.begin(), t2.begin() + r.te + 1);
	KswResult rr = sw_core(size, r.qe + 1, q2.data(), tlen, t2.data(), mat, o_del, e_del, o_ins, e_ins, XSTOP | r.score);
	if (r.score == rr.score) { r.tb = r.te - rr.te; r.qb = r.qe - rr.qe; }
	return r;
}

}  // namespace hostmem
