// host_finalize.cpp -- single-end finalize: regions -> primary/secondary marking -> mapQ -> CIGAR/NM/MD -> SAM text.
// Re-written from the behaviour of the reference's worker2 path (bwamem.c:519-584, 814-1079, 1119-1189; bwa.c:148-234;
// bwamem_extra.c:118-172).  Every arithmetic expression that feeds an integer truncation keeps the reference's operand
// types and order (float vs double, where the +.499 sits), because the SAM must come out byte-identical.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include "bwamem_host.h"
#include "host_sort.h"

namespace hostmem {

uint64_t hash_64(uint64_t key)
{
	key += ~(key << 32); key ^= (key >> 22); key += ~(key << 13); key ^= (key >> 8);
	key += (key << 3); key ^= (key >> 15); key += ~(key << 27); key ^= (key >> 31);
	return key;
}

template <class S> static inline void put_int(S &s, long long v)
{	// decimal text of v (what kputw/kputl/ksprintf("%d") emit), without a trip through snprintf
	char buf[24]; int k = 24;
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	do { buf[--k] = (char)('0' + u % 10); u /= 10; } while (u);
	if (v < 0) buf[--k] = '-';
	s.append(buf + k, (size_t)(24 - k));
}

// SamText forms: digits written in place (no temporary, no variable-size memcpy call), literals copied with their size known at compile time
static inline void put_int(SamText &s, long long v)
{
	char *w = s.need(21);
	unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
	if (v < 0) { *w++ = '-'; ++s.n; }
	int nd = 1;
	for (unsigned long long t = u; t >= 10; t /= 10) ++nd;
	for (int i = nd - 1; i >= 0; --i) { w[i] = (char)('0' + u % 10); u /= 10; }
	s.n += (size_t)nd;
}
template <size_t N> static inline void lit(SamText &s, const char (&z)[N]) { memcpy(s.need(N - 1), z, N - 1); s.n += N - 1; }

// ---- primary / secondary marking (bwamem.c:519-584) -------------------------------------------------------------------
static void mark_core(const bwagpu_opt_t &opt, int n, bwagpu_alnreg_t *a, std::vector<int> &z)
{
	int tmp = opt.a + opt.b;
	tmp = opt.o_del + opt.e_del > tmp ? opt.o_del + opt.e_del : tmp;
	tmp = opt.o_ins + opt.e_ins > tmp ? opt.o_ins + opt.e_ins : tmp;
	z.clear(); z.push_back(0);
	for (int i = 1; i < n; ++i) {
		size_t k;
		for (k = 0; k < z.size(); ++k) {
			int j = z[k];
			int b_max = a[j].qb > a[i].qb ? a[j].qb : a[i].qb, e_min = a[j].qe < a[i].qe ? a[j].qe : a[i].qe;
			if (e_min > b_max) {
				int min_l = a[i].qe - a[i].qb < a[j].qe - a[j].qb ? a[i].qe - a[i].qb : a[j].qe - a[j].qb;
				if (e_min - b_max >= min_l * opt.mask_level) {
					if (a[j].sub == 0) a[j].sub = a[i].score;
					if (a[j].score - a[i].score <= tmp && (a[j].is_alt || !a[i].is_alt)) ++a[j].sub_n;
					break;
				}
			}
		}
		if (k == z.size()) z.push_back(i); else a[i].secondary = z[k];
	}
}

struct HashLess {    // alnreg_hlt (bwamem.c:423)
	bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const {
		return a.score > b.score || (a.score == b.score && (a.is_alt < b.is_alt || (a.is_alt == b.is_alt && a.hash < b.hash)));
	}
};
struct HashLess2 {   // alnreg_hlt2 (bwamem.c:426)
	bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const {
		return a.is_alt < b.is_alt || (a.is_alt == b.is_alt && (a.score > b.score || (a.score == b.score && a.hash < b.hash)));
	}
};

int mark_primary_se(const bwagpu_opt_t &opt, Regs &av, int64_t id)
{
	int n = (int)av.size(), n_pri = 0;
	if (n == 0) return 0;
	bwagpu_alnreg_t *a = av.data();
	thread_local std::vector<int> z, map;      // (scratch kept from read to read: the stage is allocation-bound otherwise)
	for (int i = 0; i < n; ++i) {
		a[i].sub = a[i].alt_sc = 0; a[i].secondary = a[i].secondary_all = -1; a[i].hash = hash_64((uint64_t)(id + i));
		if (!a[i].is_alt) ++n_pri;
	}
	introsort(a, n, HashLess());
	mark_core(opt, n, a, z);
	for (int i = 0; i < n; ++i) {
		a[i].secondary_all = i;
		if (!a[i].is_alt && a[i].secondary >= 0 && a[a[i].secondary].is_alt) a[i].alt_sc = a[a[i].secondary].score;
	}
	if (n_pri >= 0 && n_pri < n) {
		map.resize(n);
		if (n_pri > 0) introsort(a, n, HashLess2());
		for (int i = 0; i < n; ++i) map[a[i].secondary_all] = i;
		for (int i = 0; i < n; ++i) {
			if (a[i].secondary >= 0) { a[i].secondary_all = map[a[i].secondary]; if (a[i].is_alt) a[i].secondary = INT_MAX; }
			else a[i].secondary_all = -1;
		}
		if (n_pri > 0) {
			for (int i = 0; i < n_pri; ++i) { a[i].sub = 0; a[i].secondary = -1; }
			mark_core(opt, n_pri, a, z);
		}
	} else for (int i = 0; i < n; ++i) a[i].secondary_all = a[i].secondary;
	return n_pri;
}

void reorder_primary5(int T, Regs &av)
{
	int n = (int)av.size(), n_pri = 0, left_st = INT_MAX, left_k = -1;
	bwagpu_alnreg_t *a = av.data();
	for (int k = 0; k < n; ++k) if (a[k].secondary < 0 && !a[k].is_alt && a[k].score >= T) ++n_pri;
	if (n_pri <= 1) return;
	for (int k = 0; k < n; ++k) {
		if (a[k].secondary >= 0 || a[k].is_alt || a[k].score < T) continue;
		if (a[k].qb < left_st) { left_st = a[k].qb; left_k = k; }
	}
	if (left_k == 0) return;
	std::swap(a[0], a[left_k]);
	for (int k = 1; k < n; ++k) {
		if (a[k].secondary == 0) a[k].secondary = left_k; else if (a[k].secondary == left_k) a[k].secondary = 0;
		if (a[k].secondary_all == 0) a[k].secondary_all = left_k; else if (a[k].secondary_all == left_k) a[k].secondary_all = 0;
	}
}

// ---- mapQ (bwamem.c:982-1006) ----------------------------------------------------------------------------------------
int approx_mapq_se(const bwagpu_opt_t &opt, const bwagpu_alnreg_t &a)
{
	int mapq, l, sub = a.sub ? a.sub : opt.min_seed_len * opt.a;
	double identity;
	sub = a.csub > sub ? a.csub : sub;
	if (sub >= a.score) return 0;
	l = a.qe - a.qb > a.re - a.rb ? a.qe - a.qb : (int)(a.re - a.rb);
	identity = 1. - (double)(l * opt.a - a.score) / (opt.a + opt.b) / l;
	if (a.score == 0) mapq = 0;
	else if (opt.mapQ_coef_len > 0) {
		double tmp = l < opt.mapQ_coef_len ? 1. : opt.mapQ_coef_fac / log(l);
		tmp *= identity * identity;
		mapq = (int)(6.02 * (a.score - sub) / opt.a * tmp * tmp + .499);
	} else {
		mapq = (int)(30 * (1. - (double)sub / a.score) * log(a.seedcov) + .499);   // MEM_MAPQ_COEF 30.0 (bwamem.c:40)
		mapq = identity < 0.95 ? (int)(mapq * identity * identity + .499) : mapq;
	}
	if (a.sub_n > 0) mapq -= (int)(4.343 * log(a.sub_n + 1) + .499);
	if (mapq > 60) mapq = 60;
	if (mapq < 0) mapq = 0;
	mapq = (int)(mapq * (1. - a.frac_rep) + .499);
	return mapq;
}

// NM and MD from a CIGAR over the (possibly reversed) query and reference segments (bwa.c:196-226)
static void nm_md(bool fwd, const uint8_t *query, const uint8_t *rseq, const std::vector<uint32_t> &cigar, int *NM, std::string &md)
{
	const char *int2base = fwd ? "ACGTN" : "TGCAN";
	int x = 0, y = 0, u = 0, n_mm = 0, n_gap = 0, n_cigar = (int)cigar.size();
	md.clear();
	for (int k = 0; k < n_cigar; ++k) {
		int op = cigar[k] & 0xf, len = cigar[k] >> 4;
		if (op == 0) {
			for (int i = 0; i < len; ++i) {
				if (query[x + i] != rseq[y + i]) { put_int(md, u); md += int2base[rseq[y + i]]; ++n_mm; u = 0; }
				else ++u;
			}
			x += len; y += len;
		} else if (op == 2) {
			if (k > 0 && k < n_cigar - 1) {
				put_int(md, u); md += '^';
				for (int i = 0; i < len; ++i) md += int2base[rseq[y + i]];
				u = 0; n_gap += len;
			}
			y += len;
		} else if (op == 1) { x += len; n_gap += len; }
	}
	put_int(md, u);
	*NM = n_mm + n_gap;
}

// ---- CIGAR + NM + MD (bwa_gen_cigar2, bwa.c:148-234) -------------------------------------------------------------------
// returns false when the reference would return a NULL cigar; *score is set whenever DP/ungapped scoring ran
static bool gen_cigar2(const bwagpu_opt_t &opt, const RefSeqs &ref, int w_, int l_query, const uint8_t *query_, int64_t rb, int64_t re,
					   int *score, std::vector<uint32_t> &cigar, int *NM, std::string &md)
{
	const int64_t l_pac = ref.l_pac;
	cigar.clear(); md.clear(); *NM = -1;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return false;
	std::vector<uint8_t> rseq, query(query_, query_ + l_query);
	ref.get_seq(rb, re, rseq);
	if ((int64_t)rseq.size() != re - rb) return false;
	const int rlen = (int)rseq.size();
	if (rb >= l_pac) { std::reverse(query.begin(), query.end()); std::reverse(rseq.begin(), rseq.end()); }   // left-align gaps on the forward strand
	if (l_query == rlen && w_ == 0) {
		cigar.push_back((uint32_t)l_query << 4);
		int s = 0;
		for (int i = 0; i < l_query; ++i) s += opt.mat[rseq[i] * 5 + query[i]];
		*score = s;
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins) / opt.e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del) / opt.e_del + 1.);
		int max_gap = max_ins > max_del ? max_ins : max_del, w, min_w, dl = abs(rlen - l_query);
		max_gap = max_gap > 1 ? max_gap : 1;
		w = (max_gap + dl + 1) >> 1; w = w < w_ ? w : w_;
		min_w = dl + 3; w = w > min_w ? w : min_w;
		*score = ksw_global2(l_query, query.data(), rlen, rseq.data(), opt.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, w, &cigar);
	}
	nm_md(rb < l_pac, query.data(), rseq.data(), cigar, NM, md);
	return true;
}

static inline int infer_bw(int l1, int l2, int score, int a, int q, int r)
{	// bwamem.c:818-825
	int w;
	if (l1 == l2 && l1 * a - score < (q + r - a) << 1) return 0;
	w = (int)((double)((l1 < l2 ? l1 : l2) * a - score - q) / r + 2.);
	if (w < abs(l1 - l2)) w = abs(l1 - l2);
	return w;
}

// ---- region -> alignment (mem_reg2aln, bwamem.c:1119-1189) -------------------------------------------------------------
// a usable device result for this region, if any (see CigHints)
static const bwagpu_cigar_t *find_hint(const CigHints *h, const bwagpu_alnreg_t &a)
{
	if (!h) return nullptr;
	for (int k = 0; k < h->n; ++k) {
		const bwagpu_alnreg_t &r = h->regs[k];
		if (r.rb == a.rb && r.re == a.re && r.qb == a.qb && r.qe == a.qe && r.truesc == a.truesc && r.w == a.w)
			return h->cigs[k].n_cigar >= 0 && ((h->cigs[k].n_cigar <= 6 && h->cigs[k].md_len <= 8) || h->ops) ? &h->cigs[k] : nullptr;
	}
	return nullptr;
}

// What bwagpu_batch_cigars delivers for one region, computed here on the host (tests: the device records must equal these;
// they also let the CPU suite exercise the hint path on thousands of reads).
void host_region_cigar(const bwagpu_opt_t &opt, const RefSeqs &ref, const uint8_t *query, const bwagpu_alnreg_t &ar, bwagpu_cigar_t *out, std::vector<uint32_t> *ext)
{
	out->score = 2; out->n_cigar = -1; out->nm = -1; out->md_len = 0; out->md = 0; for (int k = 0; k < 6; ++k) out->cigar[k] = 0;
	if (ar.score < opt.T) { out->score = 1; return; }
	const int qb = ar.qb, qe = ar.qe; const int64_t rb = ar.rb, re = ar.re;
	if (qe - qb <= 0 || rb >= re || (rb < ref.l_pac && re > ref.l_pac)) return;
	int score = 0, last_sc = -(1 << 30), NM = -1, i = 0;
	int tmp = infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_del, opt.e_del);
	int w2 = infer_bw(qe - qb, (int)(re - rb), ar.truesc, opt.a, opt.o_ins, opt.e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt.w) w2 = w2 < ar.w ? w2 : ar.w;
	std::vector<uint32_t> cigar; std::string md;
	do {
		w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
		gen_cigar2(opt, ref, w2, qe - qb, query + qb, rb, re, &score, cigar, &NM, md);
		if (score == last_sc || w2 == opt.w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar.truesc - opt.a);
	// MD strings of more than 8 characters and 7..64 operations go to the operation array, like the device's records (operations first)
	const bool ext_ops = cigar.size() > 6, ext_md = md.size() > 8;
	if ((ext_ops || ext_md) && !ext) { out->score = 3; return; }
	if (cigar.size() > 32768 || md.size() > 98304) { out->score = 3; return; }      // (the device's limits, k_cigar_long)
	auto md4 = [&](size_t w) { uint32_t v = 0; for (size_t b = 0; b < 4; ++b) if (4 * w + b < md.size()) v |= (uint32_t)(unsigned char)md[4 * w + b] << (8 * b); return v; };
	out->score = score; out->n_cigar = (int)cigar.size(); out->nm = NM; out->md_len = (int)md.size();
	if (ext_ops) {
		const uint64_t at = ext->size();
		ext->insert(ext->end(), cigar.begin(), cigar.end());
		out->cigar[0] = (uint32_t)at; out->cigar[1] = (uint32_t)(at >> 32);
	} else for (size_t k = 0; k < cigar.size(); ++k) out->cigar[k] = cigar[k];
	if (ext_md) {
		out->md = ext->size();
		for (size_t w = 0; w < (md.size() + 3) / 4; ++w) ext->push_back(md4(w));
	} else out->md = (uint64_t)md4(1) << 32 | md4(0);
}

Aln reg2aln(const bwagpu_opt_t &opt, const RefSeqs &ref, int l_query, const uint8_t *query, const bwagpu_alnreg_t *ar, const CigHints *hints)
{
	Aln a;
	if (ar == 0 || ar->rb < 0 || ar->re < 0) { a.rid = -1; a.pos = -1; a.flag |= 0x4; return a; }
	int qb = ar->qb, qe = ar->qe, score = 0, last_sc = -(1 << 30), NM = -1, i = 0;
	int64_t rb = ar->rb, re = ar->re;
	a.mapq = ar->secondary < 0 ? approx_mapq_se(opt, *ar) : 0;
	if (ar->secondary >= 0) a.flag |= 0x100;
	int tmp = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt.a, opt.o_del, opt.e_del);
	int w2 = infer_bw(qe - qb, (int)(re - rb), ar->truesc, opt.a, opt.o_ins, opt.e_ins);
	w2 = w2 > tmp ? w2 : tmp;
	if (w2 > opt.w) w2 = w2 < ar->w ? w2 : ar->w;
	if (const bwagpu_cigar_t *pc = find_hint(hints, *ar)) {   // the loop below already ran on the device, NM and MD (bwa.c:196-226) included
		if (pc->n_cigar <= 6) a.cigar.assign(pc->cigar, pc->cigar + pc->n_cigar);
		else { const uint32_t *o = hints->ops + ((uint64_t)pc->cigar[1] << 32 | pc->cigar[0]); a.cigar.assign(o, o + pc->n_cigar); }
		NM = pc->nm;
		if (pc->md_len <= 8) { char b[8]; memcpy(b, &pc->md, 8); a.md.assign(b, (size_t)pc->md_len); }
		else a.md.assign((const char*)(hints->ops + pc->md), (size_t)pc->md_len);      // (four characters per entry, first in the low byte: the bytes in order on this little-endian host)
	} else do {
		w2 = w2 < opt.w << 2 ? w2 : opt.w << 2;
		gen_cigar2(opt, ref, w2, qe - qb, query + qb, rb, re, &score, a.cigar, &NM, a.md);
		if (score == last_sc || w2 == opt.w << 2) break;
		last_sc = score;
		w2 <<= 1;
	} while (++i < 3 && score < ar->truesc - opt.a);
	a.NM = NM;
	bool is_rev = (rb < ref.l_pac ? rb : re - 1) >= ref.l_pac;
	int64_t p0 = rb < ref.l_pac ? rb : re - 1;
	int64_t pos = is_rev ? (ref.l_pac << 1) - 1 - p0 : p0;          // bns_depos
	a.is_rev = is_rev;
	if (!a.cigar.empty()) {   // squeeze out a leading or trailing deletion
		if ((a.cigar[0] & 0xf) == 2) { pos += a.cigar[0] >> 4; a.cigar.erase(a.cigar.begin()); }
		else if ((a.cigar.back() & 0xf) == 2) a.cigar.pop_back();
	}
	if (qb != 0 || qe != l_query) {   // clipping
		int clip5 = is_rev ? l_query - qe : qb, clip3 = is_rev ? qb : l_query - qe;
		if (clip5) a.cigar.insert(a.cigar.begin(), (uint32_t)clip5 << 4 | 3);
		if (clip3) a.cigar.push_back((uint32_t)clip3 << 4 | 3);
	}
	a.rid = ref.pos2rid(pos);
	a.pos = pos - ref.ctg[a.rid].offset;
	a.score = ar->score; a.sub = ar->sub > ar->csub ? ar->sub : ar->csub;
	a.is_alt = ar->is_alt; a.alt_sc = ar->alt_sc;
	return a;
}

// ---- XA strings (mem_gen_alt, bwamem_extra.c:118-172) ------------------------------------------------------------------
static inline int pri_idx(double ratio, const bwagpu_alnreg_t *a, int i)
{
	int k = a[i].secondary_all;
	if (k >= 0 && a[i].score >= a[k].score * ratio) return k;
	return -1;
}

// returns false when no XA exists for any region (the reference's NULL)
static bool gen_alt(const bwagpu_opt_t &opt, const RefSeqs &ref, const Regs &av, int l_query, const uint8_t *query, std::vector<std::string> &xa, std::vector<char> &has, const CigHints *hints)
{
	int n = (int)av.size(), tot = 0;
	const bwagpu_alnreg_t *a = av.data();
	thread_local std::vector<int> cnt; thread_local std::vector<char> has_alt;
	cnt.assign(n, 0); has_alt.assign(n, 0);
	for (int i = 0; i < n; ++i) {
		int r = pri_idx(opt.XA_drop_ratio, a, i);
		if (r >= 0) { ++cnt[r]; ++tot; if (a[i].is_alt) has_alt[r] = 1; }
	}
	if (tot == 0) return false;
	xa.assign(n, std::string()); has.assign(n, 0);
	for (int i = 0; i < n; ++i) {
		int r = pri_idx(opt.XA_drop_ratio, a, i);
		if (r < 0) continue;
		if (cnt[r] > opt.max_XA_hits_alt || (!has_alt[r] && cnt[r] > opt.max_XA_hits)) continue;
		Aln t = reg2aln(opt, ref, l_query, query, &a[i], hints);
		std::string &s = xa[r];
		s += ref.ctg[t.rid].name; s += ','; s += "+-"[t.is_rev]; put_int(s, t.pos + 1); s += ',';
		for (uint32_t c : t.cigar) { put_int(s, c >> 4); s += "MIDSHN"[c & 0xf]; }
		s += ','; put_int(s, t.NM);
		if (opt.flag & F_XB) { s += ','; put_int(s, t.score); s += ','; put_int(s, t.mapq); }
		s += ';';
		has[r] = 1;
	}
	return true;
}

bool gen_alt_for_pe(const bwagpu_opt_t &opt, const RefSeqs &ref, const Regs &av, int l_query, const uint8_t *query, std::vector<std::string> &xa, std::vector<char> &has, const CigHints *hints)
{
	return gen_alt(opt, ref, av, l_query, query, xa, has, hints);
}

// ---- SAM record (mem_aln2sam, bwamem.c:851-976) ------------------------------------------------------------------------
static int get_rlen(const std::vector<uint32_t> &c)
{
	int l = 0;
	for (uint32_t x : c) { int op = x & 0xf; if (op == 0 || op == 2) l += x >> 4; }
	return l;
}

static void add_cigar(const bwagpu_opt_t &opt, const std::vector<uint32_t> &cigar, bool is_alt, SamText &s, int which)
{	// bwamem.c:838-849
	if (!cigar.empty()) {
		for (uint32_t x : cigar) {
			int c = x & 0xf;
			if (!(opt.flag & F_SOFTCLIP) && !is_alt && (c == 3 || c == 4)) c = which ? 4 : 3;
			put_int(s, x >> 4); s += "MIDSH"[c];
		}
	} else s += '*';
}

// SEQ and QUAL columns: letters of the base codes ("ACGTN"[c]; code 5 gives the literal's NUL, as in the reference's table), reversed and
// complemented for a reverse-strand record, qualities reversed.  Sixteen bytes per step with SSSE3 byte shuffles where the CPU has them.
static void bases_fwd(const uint8_t *s, int n, char *d);
static void bases_rc(const uint8_t *s, int n, char *d);
static void bytes_rev(const char *s, int n, char *d);
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("ssse3"))) static void bases_fwd_v(const uint8_t *s, int n, char *d)
{
	const __m128i lut = _mm_setr_epi8('A', 'C', 'G', 'T', 'N', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
	int i = 0;
	for (; i + 16 <= n; i += 16) _mm_storeu_si128((__m128i*)(d + i), _mm_shuffle_epi8(lut, _mm_loadu_si128((const __m128i*)(s + i))));
	for (; i < n; ++i) d[i] = "ACGTN"[s[i]];
}
__attribute__((target("ssse3"))) static void bases_rc_v(const uint8_t *s, int n, char *d)
{
	const __m128i lut = _mm_setr_epi8('T', 'G', 'C', 'A', 'N', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
	int k = 0;
	for (; k + 16 <= n; k += 16) _mm_storeu_si128((__m128i*)(d + k), _mm_shuffle_epi8(lut, _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(s + n - 16 - k)), rev)));
	for (; k < n; ++k) d[k] = "TGCAN"[s[n - 1 - k]];
}
__attribute__((target("ssse3"))) static void bytes_rev_v(const char *s, int n, char *d)
{
	const __m128i rev = _mm_setr_epi8(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
	int k = 0;
	for (; k + 16 <= n; k += 16) _mm_storeu_si128((__m128i*)(d + k), _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)(s + n - 16 - k)), rev));
	for (; k < n; ++k) d[k] = s[n - 1 - k];
}
static const bool g_ssse3 = __builtin_cpu_supports("ssse3") && !getenv("BWAGPU_CLI_NO_AVX2");
#else
static const bool g_ssse3 = false;
static void bases_fwd_v(const uint8_t*, int, char*) {} static void bases_rc_v(const uint8_t*, int, char*) {} static void bytes_rev_v(const char*, int, char*) {}
#endif
static void bases_fwd(const uint8_t *s, int n, char *d) { if (g_ssse3) bases_fwd_v(s, n, d); else for (int i = 0; i < n; ++i) d[i] = "ACGTN"[s[i]]; }
static void bases_rc(const uint8_t *s, int n, char *d) { if (g_ssse3) bases_rc_v(s, n, d); else for (int k = 0; k < n; ++k) d[k] = "TGCAN"[s[n - 1 - k]]; }
static void bytes_rev(const char *s, int n, char *d) { if (g_ssse3) bytes_rev_v(s, n, d); else for (int k = 0; k < n; ++k) d[k] = s[n - 1 - k]; }

void aln2sam(const bwagpu_opt_t &opt, const RefSeqs &ref, SamText &str, const Read &s, const std::vector<Aln> &list, int which, const Aln *m_, const char *rg_id)
{
	// The reference works on copies of the record and of the mate (bwamem.c:859-872) because it borrows coordinates from the
	// mapped end for an unmapped one; only a few scalars and "has no CIGAR any more" change, so those are copied here.
	const Aln &src = list[which];
	struct View { int flag, rid, mapq, NM, score, sub, alt_sc; int64_t pos; bool is_rev, is_alt, has_xa; const std::vector<uint32_t> *cigar; const std::string *md, *xa; };
	static const std::vector<uint32_t> no_cigar;
	auto view = [](const Aln &a) { View v; v.flag = a.flag; v.rid = a.rid; v.mapq = a.mapq; v.NM = a.NM; v.score = a.score; v.sub = a.sub; v.alt_sc = a.alt_sc; v.pos = a.pos;
		v.is_rev = a.is_rev; v.is_alt = a.is_alt; v.has_xa = a.has_xa; v.cigar = &a.cigar; v.md = &a.md; v.xa = &a.xa; return v; };
	View p = view(src), mtmp; View *m = 0;
	const int n = (int)list.size();
	if (m_) { mtmp = view(*m_); m = &mtmp; }
	p.flag |= m ? 0x1 : 0;
	p.flag |= p.rid < 0 ? 0x4 : 0;
	p.flag |= m && m->rid < 0 ? 0x8 : 0;
	if (p.rid < 0 && m && m->rid >= 0) { p.rid = m->rid; p.pos = m->pos; p.is_rev = m->is_rev; p.cigar = &no_cigar; }
	if (m && m->rid < 0 && p.rid >= 0) { m->rid = p.rid; m->pos = p.pos; m->is_rev = p.is_rev; m->cigar = &no_cigar; }
	p.flag |= p.is_rev ? 0x10 : 0;
	p.flag |= m && m->is_rev ? 0x20 : 0;
	str += s.name; str += '\t';
	put_int(str, (p.flag & 0xffff) | (p.flag & 0x10000 ? 0x100 : 0)); str += '\t';
	if (p.rid >= 0) {
		str += ref.ctg[p.rid].name; str += '\t';
		put_int(str, p.pos + 1); str += '\t';
		put_int(str, p.mapq); str += '\t';
		add_cigar(opt, *p.cigar, p.is_alt, str, which);
	} else lit(str, "*\t0\t0\t*");
	str += '\t';
	if (m && m->rid >= 0) {
		if (p.rid == m->rid) str += '='; else str += ref.ctg[m->rid].name;
		str += '\t';
		put_int(str, m->pos + 1); str += '\t';
		if (p.rid == m->rid) {
			int64_t p0 = p.pos + (p.is_rev ? get_rlen(*p.cigar) - 1 : 0);
			int64_t p1 = m->pos + (m->is_rev ? get_rlen(*m->cigar) - 1 : 0);
			if (m->cigar->empty() || p.cigar->empty()) str += '0';
			else put_int(str, -(p0 - p1 + (p0 > p1 ? 1 : p0 < p1 ? -1 : 0)));
		} else str += '0';
	} else lit(str, "*\t0\t0");
	str += '\t';
	if (p.flag & 0x100) lit(str, "*\t*");
	else {
		int qb = 0, qe = s.l_seq;
		const bool trim = !p.cigar->empty() && which && !(opt.flag & F_SOFTCLIP) && !p.is_alt;
		if (!p.is_rev) {
			if (trim) {
				if (((*p.cigar)[0] & 0xf) == 4 || ((*p.cigar)[0] & 0xf) == 3) qb += (*p.cigar)[0] >> 4;
				if ((p.cigar->back() & 0xf) == 4 || (p.cigar->back() & 0xf) == 3) qe -= p.cigar->back() >> 4;
			}
			{ const size_t k = (size_t)(qe > qb ? qe - qb : 0); bases_fwd(s.seq + qb, (int)k, str.need(k)); str.n += k; }   // (written in place: a checked append per base was a fifth of the stage)
			str += '\t';
			if (s.qual) str.append(s.qual + qb, s.qual + qe); else str += '*';
		} else {
			if (trim) {
				if (((*p.cigar)[0] & 0xf) == 4 || ((*p.cigar)[0] & 0xf) == 3) qe -= (*p.cigar)[0] >> 4;
				if ((p.cigar->back() & 0xf) == 4 || (p.cigar->back() & 0xf) == 3) qb += p.cigar->back() >> 4;
			}
			{ const size_t k = (size_t)(qe > qb ? qe - qb : 0); bases_rc(s.seq + qb, (int)k, str.need(k)); str.n += k; }
			str += '\t';
			if (s.qual) { const size_t k = (size_t)(qe > qb ? qe - qb : 0); bytes_rev(s.qual + qb, (int)k, str.need(k)); str.n += k; } else str += '*';
		}
	}
	if (!p.cigar->empty()) { lit(str, "\tNM:i:"); put_int(str, p.NM); lit(str, "\tMD:Z:"); str += *p.md; }
	if (m && !m->cigar->empty()) { lit(str, "\tMC:Z:"); add_cigar(opt, *m->cigar, m->is_alt, str, which); }
	if (m) { lit(str, "\tMQ:i:"); put_int(str, m->mapq); }
	if (p.score >= 0) { lit(str, "\tAS:i:"); put_int(str, p.score); }
	if (p.sub >= 0) { lit(str, "\tXS:i:"); put_int(str, p.sub); }
	if (rg_id && rg_id[0]) { lit(str, "\tRG:Z:"); str += rg_id; }
	if (!(p.flag & 0x100)) {
		int i;
		for (i = 0; i < n; ++i) if (i != which && !(list[i].flag & 0x100)) break;
		if (i < n) {
			lit(str, "\tSA:Z:");
			for (i = 0; i < n; ++i) {
				const Aln &r = list[i];
				if (i == which || (r.flag & 0x100)) continue;
				str += ref.ctg[r.rid].name; str += ','; put_int(str, r.pos + 1); str += ','; str += "+-"[r.is_rev]; str += ',';
				for (uint32_t c : r.cigar) { put_int(str, c >> 4); str += "MIDSH"[c & 0xf]; }
				str += ','; put_int(str, r.mapq); str += ','; put_int(str, r.NM); str += ';';
			}
		}
		if (p.alt_sc > 0) { char buf[64]; snprintf(buf, sizeof buf, "\tpa:f:%.3f", (double)p.score / p.alt_sc); str += buf; }
	}
	if (p.has_xa) { str += (opt.flag & F_XB) ? "\tXB:Z:" : "\tXA:Z:"; str += *p.xa; }
	if (s.comment) { str += '\t'; str += s.comment; }
	if ((opt.flag & F_REF_HDR) && p.rid >= 0 && !ref.ctg[p.rid].anno.empty()) {
		lit(str, "\tXR:Z:");
		for (char c : ref.ctg[p.rid].anno) str += c == '\t' ? ' ' : c;
	}
	str += '\n';
}

// ---- all records of one read (mem_reg2sam, bwamem.c:1033-1079) -----------------------------------------------------------
void reg2sam(const bwagpu_opt_t &opt, const RefSeqs &ref, SamText &out, const Read &s, Regs &av, int extra_flag, const Aln *m, const char *rg_id)
{
	std::vector<std::string> xa; std::vector<char> has;
	bool have_xa = false;
	out.reserve(out.size() + 2 * (size_t)s.l_seq + 320);
	if (!(opt.flag & F_ALL)) have_xa = gen_alt(opt, ref, av, s.l_seq, s.seq, xa, has, s.hints);
	std::vector<Aln> aa;
	int l = 0;
	const int n = (int)av.size();
	for (int k = 0; k < n; ++k) {
		const bwagpu_alnreg_t &p = av[k];
		if (p.score < opt.T) continue;
		if (p.secondary >= 0 && (p.is_alt || !(opt.flag & F_ALL))) continue;
		if (p.secondary >= 0 && p.secondary < INT_MAX && p.score < av[p.secondary].score * opt.drop_ratio) continue;
		Aln q = reg2aln(opt, ref, s.l_seq, s.seq, &p, s.hints);
		if (have_xa && has[k]) { q.has_xa = true; q.xa = xa[k]; }
		q.flag |= extra_flag;
		if (p.secondary >= 0) q.sub = -1;
		if (l && p.secondary < 0) q.flag |= (opt.flag & F_NO_MULTI) ? 0x10000 : 0x800;
		if (!(opt.flag & F_KEEP_SUPP_MAPQ) && l && !p.is_alt && q.mapq > aa[0].mapq) q.mapq = aa[0].mapq;
		aa.push_back(std::move(q));
		++l;
	}
	if (aa.empty()) {
		std::vector<Aln> one(1, reg2aln(opt, ref, s.l_seq, s.seq, 0));
		one[0].flag |= extra_flag;
		aln2sam(opt, ref, out, s, one, 0, m, rg_id);
	} else for (int k = 0; k < (int)aa.size(); ++k) aln2sam(opt, ref, out, s, aa, k, m, rg_id);
}

}  // namespace hostmem
