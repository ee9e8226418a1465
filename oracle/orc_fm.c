/* oracle/orc_fm.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * FM-index queries of the seeding stage: rank, bidirectional extension, SMEMs, re-seeding, SA lookup. */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static inline void push_intv(orc_intv_v *v, const orc_intv_t *p)
{
	if (v->n == v->m) { v->m = v->m ? v->m << 1 : 16; v->a = (orc_intv_t*)realloc(v->a, v->m * sizeof(orc_intv_t)); }
	v->a[v->n++] = *p;
}

/* number of symbols == c among the 2-bit fields of w selected by the 0x55555555-style mask `sel` */
static inline int cnt2(uint32_t w, int c, uint32_t sel)
{
	uint32_t lo = (c & 1) ? w : ~w, hi = (c & 2) ? w >> 1 : ~(w >> 1);
	return __builtin_popcount(lo & hi & sel);
}

/* bwt_occ4 (bwt.c:169-186): cnt[c] = #c in BWT[0..k] inclusive; k == -1 -> 0.
 * A 64-byte block (bwtindex.c:150-172) = 4 x u64 running counts + 8 x u32 of 16 bases each,
 * base i of a word in bits (15-i)*2 (bwt.h:74-80).  `$` is not stored: k -= (k >= primary). */
void orc_occ4(const orc_index_t *ix, uint64_t k, uint64_t cnt[4])
{
	const uint32_t *blk; int c, j, nfull, ntop; uint32_t sel;
	if (k == (uint64_t)-1) { cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0; return; }
	k -= (k >= ix->primary);
	blk = ix->bwt + ((k >> 7) << 4);
	memcpy(cnt, blk, 32);
	nfull = (int)(k & 127) >> 4;      /* whole words before the one holding k */
	ntop = (int)(k & 15) + 1;         /* bases of the last word that count */
	sel = 0x55555555u & (0xffffffffu << (32 - 2 * ntop));
	for (c = 0; c < 4; ++c) {
		int x = 0;
		for (j = 0; j < nfull; ++j) x += cnt2(blk[8 + j], c, 0x55555555u);
		x += cnt2(blk[8 + nfull], c, sel);
		cnt[c] += x;
	}
}

/* bwt_extend (bwt.c:262-275) */
void orc_extend(const orc_index_t *ix, const orc_intv_t *ik, orc_intv_t ok[4], int is_back)
{
	uint64_t tk[4], tl[4], a = is_back ? ik->x0 : ik->x1, other = is_back ? ik->x1 : ik->x0, s = ik->x2, o[4];
	int c;
	orc_occ4(ix, a - 1, tk);
	orc_occ4(ix, a - 1 + s, tl);
	for (c = 0; c < 4; ++c) ok[c].x2 = tl[c] - tk[c];
	o[3] = other + (a <= ix->primary && a + s - 1 >= ix->primary);
	o[2] = o[3] + ok[3].x2; o[1] = o[2] + ok[2].x2; o[0] = o[1] + ok[1].x2;
	for (c = 0; c < 4; ++c) {
		uint64_t na = ix->L2[c] + 1 + tk[c];
		if (is_back) ok[c].x0 = na, ok[c].x1 = o[c]; else ok[c].x1 = na, ok[c].x0 = o[c];
	}
}

static inline void init_intv(const orc_index_t *ix, int c, orc_intv_t *ik)
{	/* bwt_set_intv (bwt.h:82) */
	ik->x0 = ix->L2[c] + 1; ik->x2 = ix->L2[c+1] - ix->L2[c]; ik->x1 = ix->L2[3-c] + 1; ik->info = 0;
}

static void reverse(orc_intv_v *v)
{
	size_t i;
	for (i = 0; i < v->n >> 1; ++i) { orc_intv_t t = v->a[i]; v->a[i] = v->a[v->n-1-i]; v->a[v->n-1-i] = t; }
}

/* bwt_smem1a with max_intv == 0 (bwt.c:289-351) */
int orc_smem1(const orc_index_t *ix, int len, const uint8_t *q, int x, int min_intv, orc_intv_v *mem)
{
	orc_intv_v va = {0,0,0}, vb = {0,0,0}, *prev = &va, *curr = &vb, *sw;
	orc_intv_t ik, ok[4];
	int i, ret; size_t j;
	mem->n = 0;
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	init_intv(ix, q[x], &ik); ik.info = x + 1;
	for (i = x + 1; i < len; ++i) { /* forward: remember the interval each time its size is about to change */
		if (q[i] < 4) {
			int c = 3 - q[i];
			orc_extend(ix, &ik, ok, 0);
			if (ok[c].x2 != ik.x2) {
				push_intv(curr, &ik);
				if (ok[c].x2 < (uint64_t)min_intv) break;
			}
			ik = ok[c]; ik.info = i + 1;
		} else { push_intv(curr, &ik); break; }
	}
	if (i == len) push_intv(curr, &ik);
	reverse(curr);
	ret = (int)curr->a[0].info;
	sw = curr; curr = prev; prev = sw;
	for (i = x - 1; i >= -1; --i) { /* backward: extend every surviving interval by q[i] */
		int c = i < 0 ? -1 : q[i] < 4 ? q[i] : -1;
		curr->n = 0;
		for (j = 0; j < prev->n; ++j) {
			orc_intv_t *p = &prev->a[j];
			if (c >= 0) orc_extend(ix, p, ok, 1);
			if (c < 0 || ok[c].x2 < (uint64_t)min_intv) {
				if (curr->n == 0 && (mem->n == 0 || (uint64_t)(i + 1) < mem->a[mem->n-1].info >> 32)) {
					ik = *p; ik.info |= (uint64_t)(i + 1) << 32;
					push_intv(mem, &ik);
				}
			} else if (curr->n == 0 || ok[c].x2 != curr->a[curr->n-1].x2) {
				ok[c].info = p->info;
				push_intv(curr, &ok[c]);
			}
		}
		if (curr->n == 0) break;
		sw = curr; curr = prev; prev = sw;
	}
	reverse(mem);
	free(va.a); free(vb.a);
	return ret;
}

/* bwt_seed_strategy1 (bwt.c:358-379) */
int orc_seed_strategy1(const orc_index_t *ix, int len, const uint8_t *q, int x, int min_len, int max_intv, orc_intv_t *mem)
{
	orc_intv_t ik, ok[4]; int i;
	memset(mem, 0, sizeof(*mem));
	if (q[x] > 3) return x + 1;
	init_intv(ix, q[x], &ik);
	for (i = x + 1; i < len; ++i) {
		int c;
		if (q[i] > 3) return i + 1;
		c = 3 - q[i];
		orc_extend(ix, &ik, ok, 0);
		if (ok[c].x2 < (uint64_t)max_intv && i - x >= min_len) {
			*mem = ok[c];
			mem->info = (uint64_t)x << 32 | (uint64_t)(i + 1);
			return i + 1;
		}
		ik = ok[c];
	}
	return len;
}

static int intv_lt(const void *a, const void *b) { return ((const orc_intv_t*)a)->info < ((const orc_intv_t*)b)->info; }

/* mem_collect_intv (bwamem.c:140-188) */
void orc_collect_intv(const orc_opt_t *opt, const orc_index_t *ix, int len, const uint8_t *seq, orc_intv_v *out)
{
	orc_intv_v m1 = {0,0,0};
	int x = 0, split_len = (int)(opt->min_seed_len * opt->split_factor + .499);
	size_t i, k, old_n;
	out->n = 0;
	while (x < len) { /* pass 1: all SMEMs */
		if (seq[x] < 4) {
			x = orc_smem1(ix, len, seq, x, 1, &m1);
			for (i = 0; i < m1.n; ++i)
				if ((int)((uint32_t)m1.a[i].info - (m1.a[i].info >> 32)) >= opt->min_seed_len) push_intv(out, &m1.a[i]);
		} else ++x;
	}
	old_n = out->n;
	for (k = 0; k < old_n; ++k) { /* pass 2: re-seed inside long, rare SMEMs */
		orc_intv_t p = out->a[k];
		int start = (int)(p.info >> 32), end = (int32_t)p.info;
		if (end - start < split_len || p.x2 > (uint64_t)opt->split_width) continue;
		orc_smem1(ix, len, seq, (start + end) >> 1, (int)(p.x2 + 1), &m1);
		for (i = 0; i < m1.n; ++i)
			if ((int)((uint32_t)m1.a[i].info - (m1.a[i].info >> 32)) >= opt->min_seed_len) push_intv(out, &m1.a[i]);
	}
	if (opt->max_mem_intv > 0) { /* pass 3: LAST-like */
		x = 0;
		while (x < len) {
			if (seq[x] < 4) {
				orc_intv_t m;
				x = orc_seed_strategy1(ix, len, seq, x, opt->min_seed_len, (int)opt->max_mem_intv, &m);
				if (m.x2 > 0) push_intv(out, &m);
			} else ++x;
		}
	}
	orc_introsort(out->a, out->n, sizeof(orc_intv_t), intv_lt);
	free(m1.a);
}

/* bwt_sa / bwt_invPsi (bwt.c:53-59, 86-96) */
uint64_t orc_sa(const orc_index_t *ix, uint64_t k)
{
	uint64_t sa = 0, mask = (uint64_t)ix->sa_intv - 1;
	while (k & mask) {
		uint64_t x = k - (k > ix->primary), cnt[4]; int c;
		++sa;
		if (k == ix->primary) { k = 0; continue; }
		c = ix->bwt[((x >> 7) << 4) + 8 + ((x & 127) >> 4)] >> ((~x & 15) << 1) & 3;
		orc_occ4(ix, k, cnt);
		k = ix->L2[c] + cnt[c];
	}
	return sa + ix->sa[k / ix->sa_intv];
}
