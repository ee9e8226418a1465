/* oracle/orc_chain.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * Seeds -> chains (mem_chain), chain weights and the chain filter (mem_chain_flt). */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc.h"

/* ---- B-tree keyed by chain position, duplicate keys allowed -------------------------------------
 * Restates the behaviour of klib's kbtree as instantiated at bwamem.c:212-213 (KB_DEFAULT_SIZE 512,
 * 40-byte keys -> minimum degree t = 5, at most 9 keys per node, kbtree.h:52).  Equal positions can
 * coexist, and where a duplicate lands depends on node layout, so the node
 * structure is reproduced literally.  Keys are indices into the chain array. */
#define BT_T 5
#define BT_MAXK (2*BT_T-1)
typedef struct { int n, internal; int key[BT_MAXK]; int child[BT_MAXK+1]; } bt_node_t;
typedef struct { bt_node_t *nd; int n_nodes, m_nodes, root; const orc_chain_t *chains; } bt_t;

static int bt_new(bt_t *b, int internal)
{
	if (b->n_nodes == b->m_nodes) { b->m_nodes = b->m_nodes ? b->m_nodes << 1 : 16; b->nd = (bt_node_t*)realloc(b->nd, b->m_nodes * sizeof(bt_node_t)); }
	memset(&b->nd[b->n_nodes], 0, sizeof(bt_node_t));
	b->nd[b->n_nodes].internal = internal;
	return b->n_nodes++;
}

/* __kb_getp_aux (kbtree.h:117-131): first key >= pos; *r = 0 if equal, <0 if pos is smaller than that key
 * (then the index before it is returned), 1 if every key is smaller (index n-1 returned). */
static int bt_search(const bt_t *b, const bt_node_t *x, int64_t pos, int *r)
{
	int lo = 0, hi = x->n;
	if (x->n == 0) return -1;
	while (lo < hi) {
		int mid = (lo + hi) >> 1;
		if (b->chains[x->key[mid]].pos < pos) lo = mid + 1; else hi = mid;
	}
	if (lo == x->n) { *r = 1; return x->n - 1; }
	*r = pos < b->chains[x->key[lo]].pos ? -1 : 0;
	return *r < 0 ? lo - 1 : lo;
}

/* kb_intervalp, lower bound only (kbtree.h:152-168) */
static int bt_lower(const bt_t *b, int64_t pos)
{
	int x = b->root, lower = -1;
	for (;;) {
		const bt_node_t *nd = &b->nd[x]; int r = 0, i = bt_search(b, nd, pos, &r);
		if (i >= 0 && r == 0) return nd->key[i];
		if (i >= 0) lower = nd->key[i];
		if (!nd->internal) return lower;
		x = nd->child[i + 1];
	}
}

/* __kb_split (kbtree.h:173-190): child y = x->child[i] is full; move its upper half into a new right sibling */
static void bt_split(bt_t *b, int xi, int i, int yi)
{
	int zi = bt_new(b, b->nd[yi].internal), j;
	bt_node_t *x = &b->nd[xi], *y = &b->nd[yi], *z = &b->nd[zi];
	z->n = BT_T - 1;
	for (j = 0; j < BT_T - 1; ++j) z->key[j] = y->key[j + BT_T];
	if (y->internal) for (j = 0; j < BT_T; ++j) z->child[j] = y->child[j + BT_T];
	y->n = BT_T - 1;
	for (j = x->n; j > i; --j) x->child[j + 1] = x->child[j];
	x->child[i + 1] = zi;
	for (j = x->n - 1; j >= i; --j) x->key[j + 1] = x->key[j];
	x->key[i] = y->key[BT_T - 1];
	++x->n;
}

/* kb_putp / __kb_putp_aux (kbtree.h:191-224) */
static void bt_insert(bt_t *b, int k)
{
	int64_t pos = b->chains[k].pos; int xi, r;
	if (b->nd[b->root].n == BT_MAXK) {
		int s = bt_new(b, 1);
		b->nd[s].child[0] = b->root;
		bt_split(b, s, 0, b->root);
		b->root = s;
	}
	xi = b->root;
	for (;;) {
		bt_node_t *x = &b->nd[xi]; int i;
		if (!x->internal) {
			int j;
			i = bt_search(b, x, pos, &r);
			for (j = x->n - 1; j > i; --j) x->key[j + 1] = x->key[j];
			x->key[i + 1] = k; ++x->n;
			return;
		}
		i = bt_search(b, x, pos, &r) + 1;
		if (b->nd[x->child[i]].n == BT_MAXK) {
			bt_split(b, xi, i, x->child[i]);
			x = &b->nd[xi]; /* node array may have moved */
			if (pos > b->chains[x->key[i]].pos) ++i;
		}
		xi = x->child[i];
	}
}

static void bt_inorder(const bt_t *b, int xi, int *out, int *n)
{
	const bt_node_t *x = &b->nd[xi]; int i;
	for (i = 0; i < x->n; ++i) {
		if (x->internal) bt_inorder(b, x->child[i], out, n);
		out[(*n)++] = x->key[i];
	}
	if (x->internal) bt_inorder(b, x->child[x->n], out, n);
}

/* test_and_merge (bwamem.c:216-237) */
static int try_merge(const orc_opt_t *opt, int64_t l_pac, orc_chain_t *c, const orc_seed_t *p, int seed_rid)
{
	const orc_seed_t *last = &c->seeds[c->n-1], *first = &c->seeds[0];
	int64_t qend = last->qbeg + last->len, rend = last->rbeg + last->len, x, y;
	if (seed_rid != c->rid) return 0;
	if (p->qbeg >= first->qbeg && p->qbeg + p->len <= qend && p->rbeg >= first->rbeg && p->rbeg + p->len <= rend) return 1;
	if ((last->rbeg < l_pac || first->rbeg < l_pac) && p->rbeg >= l_pac) return 0;
	x = p->qbeg - last->qbeg; y = p->rbeg - last->rbeg;
	if (y >= 0 && x - y <= opt->w && y - x <= opt->w && x - last->len < opt->max_chain_gap && y - last->len < opt->max_chain_gap) {
		if (c->n == c->m) { c->m <<= 1; c->seeds = (orc_seed_t*)realloc(c->seeds, c->m * sizeof(orc_seed_t)); }
		c->seeds[c->n++] = *p;
		return 1;
	}
	return 0;
}

/* mem_chain (bwamem.c:277-342) */
orc_chain_v orc_chain(const orc_opt_t *opt, const orc_index_t *ix, int len, const uint8_t *seq)
{
	orc_chain_v out = {0,0,0}, pool = {0,0,0};
	orc_intv_v iv = {0,0,0};
	bt_t bt; size_t i; int b, e, l_rep, *order, n_ord = 0;
	if (len < opt->min_seed_len) return out;
	orc_collect_intv(opt, ix, len, seq, &iv);
	for (i = 0, b = e = l_rep = 0; i < iv.n; ++i) { /* length of the query covered by over-abundant seeds */
		int sb = (int)(iv.a[i].info >> 32), se = (int)(uint32_t)iv.a[i].info;
		if (iv.a[i].x2 <= (uint64_t)opt->max_occ) continue;
		if (sb > e) { l_rep += e - b; b = sb; e = se; }
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	memset(&bt, 0, sizeof bt);
	bt.root = bt_new(&bt, 0);
	for (i = 0; i < iv.n; ++i) {
		const orc_intv_t *p = &iv.a[i];
		int slen = (int)((uint32_t)p->info - (p->info >> 32)), count = 0;
		int step = p->x2 > (uint64_t)opt->max_occ ? (int)(p->x2 / opt->max_occ) : 1;
		int64_t k;
		for (k = 0; (uint64_t)k < p->x2 && count < opt->max_occ; k += step, ++count) {
			orc_seed_t s; int rid, lower, add = 1;
			s.rbeg = (int64_t)orc_sa(ix, p->x0 + k);
			s.qbeg = (int32_t)(p->info >> 32); s.score = s.len = slen; s.pad_ = 0;
			rid = orc_intv2rid(ix, s.rbeg, s.rbeg + s.len);
			if (rid < 0) continue;
			if (pool.n) {
				orc_chain_t probe; probe.pos = s.rbeg;
				/* the tree compares against positions stored in pool.a; a probe is looked up by value */
				bt.chains = pool.a;
				lower = bt_lower(&bt, probe.pos);
				if (lower >= 0 && try_merge(opt, ix->l_pac, &pool.a[lower], &s, rid)) add = 0;
			}
			if (add) {
				orc_chain_t c; memset(&c, 0, sizeof c);
				c.n = 1; c.m = 4; c.seeds = (orc_seed_t*)calloc(c.m, sizeof(orc_seed_t)); c.seeds[0] = s;
				c.rid = rid; c.pos = s.rbeg; c.is_alt = !!ix->ctg[rid].is_alt;
				if (pool.n == pool.m) { pool.m = pool.m ? pool.m << 1 : 16; pool.a = (orc_chain_t*)realloc(pool.a, pool.m * sizeof(orc_chain_t)); }
				pool.a[pool.n++] = c;
				bt.chains = pool.a;
				bt_insert(&bt, (int)pool.n - 1);
			}
		}
	}
	order = (int*)malloc((pool.n + 1) * sizeof(int));
	bt.chains = pool.a;
	bt_inorder(&bt, bt.root, order, &n_ord);
	out.n = out.m = pool.n;
	out.a = (orc_chain_t*)malloc((pool.n + 1) * sizeof(orc_chain_t));
	for (i = 0; i < pool.n; ++i) { out.a[i] = pool.a[order[i]]; out.a[i].frac_rep = (float)l_rep / len; }
	free(order); free(pool.a); free(bt.nd); free(iv.a);
	return out;
}

/* mem_chain_weight (bwamem.c:239-258) */
int orc_chain_weight(const orc_chain_t *c)
{
	int64_t end; int j, w = 0, wq;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->qbeg >= end) w += s->len;
		else if (s->qbeg + s->len > end) w += s->qbeg + s->len - end;
		if (s->qbeg + s->len > end) end = s->qbeg + s->len;
	}
	wq = w; w = 0;
	for (j = 0, end = 0; j < c->n; ++j) {
		const orc_seed_t *s = &c->seeds[j];
		if (s->rbeg >= end) w += s->len;
		else if (s->rbeg + s->len > end) w += s->rbeg + s->len - end;
		if (s->rbeg + s->len > end) end = s->rbeg + s->len;
	}
	if (wq < w) w = wq;
	return w < 1<<30 ? w : (1<<30) - 1;
}

static int chain_w_gt(const void *a, const void *b) { return ((const orc_chain_t*)a)->w > ((const orc_chain_t*)b)->w; }
#define CBEG(c) ((c).seeds[0].qbeg)
#define CEND(c) ((c).seeds[(c).n-1].qbeg + (c).seeds[(c).n-1].len)

/* mem_chain_flt (bwamem.c:353-411) */
int orc_chain_flt(const orc_opt_t *opt, int n_chn, orc_chain_t *a)
{
	int i, k, *kept_idx, n_kept = 0;
	if (n_chn == 0) return 0;
	for (i = k = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		c->first = -1; c->kept = 0; c->w = orc_chain_weight(c);
		if (c->w < opt->min_chain_weight) free(c->seeds); else a[k++] = *c;
	}
	n_chn = k;
	if (n_chn == 0) return 0; /* NB: the reference would touch a[0] here; with min_chain_weight filtering everything it reads freed data but keeps nothing */
	orc_introsort(a, n_chn, sizeof(orc_chain_t), chain_w_gt);
	kept_idx = (int*)malloc(n_chn * sizeof(int));
	a[0].kept = 3; kept_idx[n_kept++] = 0;
	for (i = 1; i < n_chn; ++i) {
		int large_ovlp = 0;
		for (k = 0; k < n_kept; ++k) {
			int j = kept_idx[k];
			int b_max = CBEG(a[j]) > CBEG(a[i]) ? CBEG(a[j]) : CBEG(a[i]);
			int e_min = CEND(a[j]) < CEND(a[i]) ? CEND(a[j]) : CEND(a[i]);
			if (e_min > b_max && (!a[j].is_alt || a[i].is_alt)) {
				int li = CEND(a[i]) - CBEG(a[i]), lj = CEND(a[j]) - CBEG(a[j]), min_l = li < lj ? li : lj;
				if (e_min - b_max >= min_l * opt->mask_level && min_l < opt->max_chain_gap) {
					large_ovlp = 1;
					if (a[j].first < 0) a[j].first = i;
					if (a[i].w < a[j].w * opt->drop_ratio && a[j].w - a[i].w >= opt->min_seed_len << 1) break;
				}
			}
		}
		if (k == n_kept) { kept_idx[n_kept++] = i; a[i].kept = large_ovlp ? 2 : 3; }
	}
	for (i = 0; i < n_kept; ++i) {
		orc_chain_t *c = &a[kept_idx[i]];
		if (c->first >= 0) a[c->first].kept = 1;
	}
	free(kept_idx);
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0 || a[i].kept == 3) continue;
		if (++k >= opt->max_chain_extend) break;
	}
	for (; i < n_chn; ++i) if (a[i].kept < 3) a[i].kept = 0;
	for (i = k = 0; i < n_chn; ++i) {
		if (a[i].kept == 0) free(a[i].seeds); else a[k++] = a[i];
	}
	return k;
}

/* mem_flt_chained_seeds / mem_seed_sw (bwamem.c:597-645) -- only active for long reads */
void orc_flt_chained_seeds(const orc_opt_t *opt, const orc_index_t *ix, int l_query, const uint8_t *query, int n_chn, orc_chain_t *a)
{
	double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log(l_query);
	int i, j, k, min_hsp = (int)(opt->a * min_l + .499);
	if (min_l > 0.05f * l_query) return;
	for (i = 0; i < n_chn; ++i) {
		orc_chain_t *c = &a[i];
		for (j = k = 0; j < c->n; ++j) {
			orc_seed_t *s = &c->seeds[j];
			int sc = -1;
			if (s->len < 200) {
				int qb = s->qbeg - 50, qe = s->qbeg + s->len + 50; int64_t rb = s->rbeg - 50, re = s->rbeg + s->len + 50, mid = (s->rbeg + s->rbeg + s->len) >> 1;
				if (qb < 0) qb = 0;
				if (qe > l_query) qe = l_query;
				if (rb < 0) rb = 0;
				if (re > ix->l_pac << 1) re = ix->l_pac << 1;
				if (rb < ix->l_pac && ix->l_pac < re) { if (mid < ix->l_pac) re = ix->l_pac; else rb = ix->l_pac; }
				if (qe - qb < 200 && re - rb < 200) {
					int rid; uint8_t *rs = orc_fetch_seq(ix, &rb, mid, &re, &rid);
					sc = orc_ksw_local_score(qe - qb, query + qb, (int)(re - rb), rs, 5, opt->mat, opt->o_del, opt->e_del, opt->o_ins, opt->e_ins);
					free(rs);
				}
			}
			s->score = sc;
			if (s->score < 0 || s->score >= min_hsp) {
				s->score = s->score < 0 ? s->len * opt->a : s->score;
				c->seeds[k++] = *s;
			}
		}
		c->n = k;
	}
}
