/* oracle/orc_aln.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * Chain -> alignment regions (mem_chain2aln), region de-duplication / patching, and the per-read driver. */
#include <stdlib.h>
#include <string.h>
#include <assert.h>
#include "orc.h"

static inline orc_reg_t *push_reg(orc_reg_v *v)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 8; v->a = (orc_reg_t*)realloc(v->a, v->m * sizeof(orc_reg_t)); }
	return &v->a[v->n++];
}

/* cal_max_gap (bwamem.c:647-654) */
static int max_gap(const orc_opt_t *opt, int qlen)
{
	int l_del = (int)((double)(qlen * opt->a - opt->o_del) / opt->e_del + 1.);
	int l_ins = (int)((double)(qlen * opt->a - opt->o_ins) / opt->e_ins + 1.);
	int l = l_del > l_ins ? l_del : l_ins;
	if (l < 1) l = 1;
	return l < opt->w << 1 ? l : opt->w << 1;
}

static int u64_lt(const void *a, const void *b) { return *(const uint64_t*)a < *(const uint64_t*)b; }

/* mem_chain2aln (bwamem.c:658-812) */
void orc_chain2aln(const orc_opt_t *opt, const orc_index_t *ix, int l_query, const uint8_t *query, const orc_chain_t *c, orc_reg_v *av)
{
	int64_t l_pac = ix->l_pac, rmax0 = l_pac << 1, rmax1 = 0;
	int i, k, rid; uint8_t *rseq; uint64_t *srt;
	if (c->n == 0) return;
	for (i = 0; i < c->n; ++i) { /* widest reference window any seed of the chain could reach */
		const orc_seed_t *t = &c->seeds[i];
		int64_t b = t->rbeg - (t->qbeg + max_gap(opt, t->qbeg));
		int64_t e = t->rbeg + t->len + ((l_query - t->qbeg - t->len) + max_gap(opt, l_query - t->qbeg - t->len));
		if (b < rmax0) rmax0 = b;
		if (e > rmax1) rmax1 = e;
	}
	if (rmax0 < 0) rmax0 = 0;
	if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
	if (rmax0 < l_pac && l_pac < rmax1) { if (c->seeds[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
	rseq = orc_fetch_seq(ix, &rmax0, c->seeds[0].rbeg, &rmax1, &rid);
	assert(rid == c->rid);
	srt = (uint64_t*)malloc(c->n * 8);
	for (i = 0; i < c->n; ++i) srt[i] = (uint64_t)c->seeds[i].score << 32 | (uint32_t)i;
	orc_introsort(srt, c->n, 8, u64_lt);
	for (k = c->n - 1; k >= 0; --k) { /* best-scoring seed first */
		const orc_seed_t *s = &c->seeds[(uint32_t)srt[k]];
		orc_reg_t *a; int aw0, aw1, moff;
		size_t ii;
		for (ii = 0; ii < av->n; ++ii) { /* is the seed already inside an earlier alignment of this read? */
			const orc_reg_t *p = &av->a[ii]; int64_t rd; int qd, w, mg;
			if (s->rbeg < p->rb || s->rbeg + s->len > p->re || s->qbeg < p->qb || s->qbeg + s->len > p->qe) continue;
			if (s->len - p->seedlen0 > .1 * l_query) continue;
			qd = s->qbeg - p->qb; rd = s->rbeg - p->rb;
			mg = max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p->w ? mg : p->w;
			if (qd - rd < w && rd - qd < w) break;
			qd = p->qe - (s->qbeg + s->len); rd = p->re - (s->rbeg + s->len);
			mg = max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p->w ? mg : p->w;
			if (qd - rd < w && rd - qd < w) break;
		}
		if (ii < av->n) { /* contained: extend anyway only if an overlapping, differently-placed seed exists */
			for (i = k + 1; i < c->n; ++i) {
				const orc_seed_t *t;
				if (srt[i] == 0) continue;
				t = &c->seeds[(uint32_t)srt[i]];
				if (t->len < s->len * .95) continue;
				if (s->qbeg <= t->qbeg && s->qbeg + s->len - t->qbeg >= s->len >> 2 && t->qbeg - s->qbeg != t->rbeg - s->rbeg) break;
				if (t->qbeg <= s->qbeg && t->qbeg + t->len - s->qbeg >= s->len >> 2 && s->qbeg - t->qbeg != s->rbeg - t->rbeg) break;
			}
			if (i == c->n) { srt[k] = 0; continue; }
		}
		a = push_reg(av);
		memset(a, 0, sizeof(*a));
		a->w = aw0 = aw1 = opt->w; a->score = a->truesc = -1; a->rid = c->rid;
		if (s->qbeg) { /* left extension on reversed prefixes */
			int qle, tle, gtle, gscore; int64_t tl = s->rbeg - rmax0;
			uint8_t *qs = (uint8_t*)malloc(s->qbeg), *rs = (uint8_t*)malloc(tl + 1);
			for (i = 0; i < s->qbeg; ++i) qs[i] = query[s->qbeg - 1 - i];
			for (i = 0; i < tl; ++i) rs[i] = rseq[tl - 1 - i];
			for (i = 0; i < 2; ++i) {
				int prev = a->score;
				aw0 = opt->w << i;
				a->score = orc_ksw_extend2(s->qbeg, qs, (int)tl, rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw0, opt->pen_clip5, opt->zdrop, s->len * opt->a, &qle, &tle, &gtle, &gscore, &moff);
				if (a->score == prev || moff < (aw0 >> 1) + (aw0 >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip5) { a->qb = s->qbeg - qle; a->rb = s->rbeg - tle; a->truesc = a->score; }
			else { a->qb = 0; a->rb = s->rbeg - gtle; a->truesc = gscore; }
			free(qs); free(rs);
		} else { a->score = a->truesc = s->len * opt->a; a->qb = 0; a->rb = s->rbeg; }
		if (s->qbeg + s->len != l_query) { /* right extension */
			int qle, tle, gtle, gscore, sc0 = a->score, qe = s->qbeg + s->len; int64_t re = s->rbeg + s->len - rmax0;
			for (i = 0; i < 2; ++i) {
				int prev = a->score;
				aw1 = opt->w << i;
				a->score = orc_ksw_extend2(l_query - qe, query + qe, (int)(rmax1 - rmax0 - re), rseq + re, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, aw1, opt->pen_clip3, opt->zdrop, sc0, &qle, &tle, &gtle, &gscore, &moff);
				if (a->score == prev || moff < (aw1 >> 1) + (aw1 >> 2)) break;
			}
			if (gscore <= 0 || gscore <= a->score - opt->pen_clip3) { a->qe = qe + qle; a->re = rmax0 + re + tle; a->truesc += a->score - sc0; }
			else { a->qe = l_query; a->re = rmax0 + re + gtle; a->truesc += gscore - sc0; }
		} else { a->qe = l_query; a->re = s->rbeg + s->len; }
		for (i = 0, a->seedcov = 0; i < c->n; ++i) {
			const orc_seed_t *t = &c->seeds[i];
			if (t->qbeg >= a->qb && t->qbeg + t->len <= a->qe && t->rbeg >= a->rb && t->rbeg + t->len <= a->re) a->seedcov += t->len;
		}
		a->w = aw0 > aw1 ? aw0 : aw1;
		a->seedlen0 = s->len;
		a->frac_rep = c->frac_rep;
	}
	free(srt); free(rseq);
}

/* Score-only use of bwa_gen_cigar2 (bwa.c:148-234) as made by mem_patch_reg (bwamem.c:454).
 * *ok = 0 when the reference would return without touching *score. */
int orc_global_score(const orc_opt_t *opt, const orc_index_t *ix, int w_, int l_query, uint8_t *query, int64_t rb, int64_t re, int *ok)
{
	int64_t rlen, l_pac = ix->l_pac; uint8_t *rseq, *q2 = 0; int i, score = 0;
	*ok = 0;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	rseq = orc_get_seq(ix, rb, re, &rlen);
	if (re - rb != rlen) { free(rseq); return 0; }
	if (rb >= l_pac) { /* reverse both so that gaps are left-aligned on the forward strand */
		q2 = (uint8_t*)malloc(l_query);
		for (i = 0; i < l_query; ++i) q2[i] = query[l_query - 1 - i];
		for (i = 0; i < rlen >> 1; ++i) { uint8_t t = rseq[i]; rseq[i] = rseq[rlen-1-i]; rseq[rlen-1-i] = t; }
		query = q2;
	}
	if (l_query == re - rb && w_ == 0) {
		for (i = 0; i < l_query; ++i) score += opt->mat[rseq[i] * 5 + query[i]];
	} else {
		int max_ins = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_ins) / opt->e_ins + 1.);
		int max_del = (int)((double)(((l_query + 1) >> 1) * opt->mat[0] - opt->o_del) / opt->e_del + 1.);
		int mg = max_ins > max_del ? max_ins : max_del, w, min_w, dl = abs((int)rlen - l_query);
		if (mg < 1) mg = 1;
		w = (mg + dl + 1) >> 1; if (w > w_) w = w_;
		min_w = dl + 3; if (w < min_w) w = min_w;
		score = orc_ksw_global2(l_query, query, (int)rlen, rseq, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins, w, 0, 0);
	}
	*ok = 1;
	free(rseq); free(q2);
	return score;
}

/* mem_patch_reg (bwamem.c:432-461) */
static int patch_reg(const orc_opt_t *opt, const orc_index_t *ix, uint8_t *query, const orc_reg_t *a, const orc_reg_t *b, int *w_out)
{
	int w, score = 0, q_s, r_s, ok; double r;
	if (a->rb < ix->l_pac && b->rb >= ix->l_pac) return 0;
	if (a->qb >= b->qb || a->qe >= b->qe || a->re >= b->re) return 0;
	w = (int)((a->re - b->rb) - (a->qe - b->qb)); if (w < 0) w = -w;
	r = (double)(a->re - b->rb) / (b->re - a->rb) - (double)(a->qe - b->qb) / (b->qe - a->qb); if (r < 0.) r = -r;
	if (a->re < b->rb || a->qe < b->qb) { if (w > opt->w << 1 || r >= 0.05f) return 0; }
	else if (w > opt->w << 2 || r >= 0.05f * 2) return 0;
	w += a->w + b->w;
	if (w > opt->w << 2) w = opt->w << 2;
	score = orc_global_score(opt, ix, w, b->qe - a->qb, query + a->qb, a->rb, b->re, &ok);
	q_s = (int)((double)(b->qe - a->qb) / ((b->qe - b->qb) + (a->qe - a->qb)) * (b->score + a->score) + .499);
	r_s = (int)((double)(b->re - a->rb) / ((b->re - b->rb) + (a->re - a->rb)) * (b->score + a->score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	*w_out = w;
	return score;
}

static int reg_re_lt(const void *a, const void *b) { return ((const orc_reg_t*)a)->re < ((const orc_reg_t*)b)->re; }
static int reg_best_lt(const void *a_, const void *b_)
{
	const orc_reg_t *a = (const orc_reg_t*)a_, *b = (const orc_reg_t*)b_;
	return a->score > b->score || (a->score == b->score && (a->rb < b->rb || (a->rb == b->rb && a->qb < b->qb)));
}
#define SET_NCOMP(p, v) ((p)->ncomp_isalt = ((p)->ncomp_isalt & 0xc0000000u) | ((uint32_t)(v) & 0x3fffffffu))
#define GET_NCOMP(p) ((int)((p)->ncomp_isalt << 2) >> 2)

/* mem_sort_dedup_patch (bwamem.c:463-515) */
int orc_sort_dedup_patch(const orc_opt_t *opt, const orc_index_t *ix, uint8_t *query, int n, orc_reg_t *a)
{
	int m, i, j;
	if (n <= 1) return n;
	orc_introsort(a, n, sizeof(orc_reg_t), reg_re_lt);
	for (i = 0; i < n; ++i) SET_NCOMP(&a[i], 1);
	for (i = 1; i < n; ++i) {
		orc_reg_t *p = &a[i];
		if (p->rid != a[i-1].rid || p->rb >= a[i-1].re + opt->max_chain_gap) continue;
		for (j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt->max_chain_gap; --j) {
			orc_reg_t *q = &a[j]; int64_t or_, oq, mr, mq; int score, w;
			if (q->qe == q->qb) continue;
			or_ = q->re - p->rb;
			oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (or_ > opt->mask_level_redun * mr && oq > opt->mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			} else if (q->rb < p->rb && (score = patch_reg(opt, ix, query, q, p, &w)) > 0) {
				SET_NCOMP(p, GET_NCOMP(p) + GET_NCOMP(q) + 1);
				if (q->seedcov > p->seedcov) p->seedcov = q->seedcov;
				if (q->sub > p->sub) p->sub = q->sub;
				if (q->csub > p->csub) p->csub = q->csub;
				p->qb = q->qb; p->rb = q->rb;
				p->truesc = p->score = score;
				p->w = w;
				q->qb = q->qe;
			}
		}
	}
	for (i = 0, m = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	orc_introsort(a, n, sizeof(orc_reg_t), reg_best_lt);
	for (i = 1; i < n; ++i)
		if (a[i].score == a[i-1].score && a[i].rb == a[i-1].rb && a[i].qb == a[i-1].qb) a[i].qe = a[i].qb;
	for (i = 1, m = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	return m;
}

/* mem_align1_core (bwamem.c:1081-1117); seq must already be nt4 codes 0..4 */
orc_reg_v orc_align1_core(const orc_opt_t *opt, const orc_index_t *ix, int l_seq, uint8_t *seq)
{
	orc_chain_v chn = orc_chain(opt, ix, l_seq, seq);
	orc_reg_v regs = {0,0,0}; size_t i;
	chn.n = orc_chain_flt(opt, (int)chn.n, chn.a);
	orc_flt_chained_seeds(opt, ix, l_seq, seq, (int)chn.n, chn.a);
	for (i = 0; i < chn.n; ++i) {
		orc_chain2aln(opt, ix, l_seq, seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	regs.n = orc_sort_dedup_patch(opt, ix, seq, (int)regs.n, regs.a);
	for (i = 0; i < regs.n; ++i)
		if (regs.a[i].rid >= 0 && ix->ctg[regs.a[i].rid].is_alt)
			regs.a[i].ncomp_isalt = (regs.a[i].ncomp_isalt & 0x3fffffffu) | (1u << 30);
	return regs;
}
