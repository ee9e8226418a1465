// dev_chainw.h -- seeds -> chains (mem_chain, bwamem.c:277-342) and the chain filter (mem_chain_flt, bwamem.c:353-411),
// one wavefront per read.
//
// Why not one lane per read (dev_chain.h, round 1): every lane walked its own B-tree and chain records, so each wave-wide load
// touched 64 different cache lines (measured: 18 GB of HBM-side traffic per launch for < 1 GB of algorithmic bytes), the
// pairwise chain filter of a read with hundreds of equally good chains (a read inside a repeat family: 640 chains, 200 k pair
// tests) ran as one serial dependent-load loop -- 74 ms for a single read, the whole kernel's tail -- and the waves of a batch
// advanced at the pace of their heaviest lane.  Here the order-dependent part (one B-tree look-up / insertion per seed, literally
// kbtree's) runs wave-uniformly with the tree and the chain records in LDS: a node is fetched as one 160-byte
// row, its nine keys are compared by nine lanes and the lower bound is a popcount of a ballot; node splits and shifts move all
// entries at once.  Everything that is not order-dependent is lane-parallel: chain weights (lane per chain), the pairwise filter
// (64 kept chains per step; the reference's "stop at the first chain that drops this one" is a find-first-set on the ballot), the
// flattening of the kept chains.
// Storage: tree, chain records, seed links, sort keys in the read's own region of HBM (RegionView); LDS holds the traversal stack and the weight sort's
// pairs.  Rounds 2-4 ran three tiers of one template -- tree and chain records in LDS for reads that fit, HBM for the rest -- on the premise that the tree's
// memory is what a look-up costs.  It is not (the per-seed cost is the wave-uniform logic: 5-7 us per seed in every tier), and in round 5 the HBM form alone
// measured as fast as the three together (24.2 vs 24.9 ms per million reads, profiles/r05_chain_tiers.log), with the root node -- where every look-up
// starts -- held in registers (RootCache).  One kernel, one form.
#pragma once
#include "dev_chain.h"
#include "dev_extw.h"

#define CW_STACK_INTS 32
// LDS bytes of one wave: the traversal stack and the {weight, index} pairs of up to CW_PW_HBM_TIER chains (a serial sort in HBM costs ~1 us per step)
#define CW_PW_HBM_TIER 512
#define CW_LDS_BYTES (CW_STACK_INTS * 4 + CW_PW_HBM_TIER * 8)

DEVFN i64 *cw_pos(i32 *node) { return (i64*)(node + 22); }
DEVFN i64 readlane_i64(i64 v, int l)
{
	const int lo = __builtin_amdgcn_readlane((int)(u32)(u64)v, l), hi = __builtin_amdgcn_readlane((int)(u32)((u64)v >> 32), l);
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}

// a 64-bit field of a record held one int per lane: ints l and l + 1
DEVFN i64 readlane_i64x(i32 v, int l)
{
	const int lo = __builtin_amdgcn_readlane(v, l), hi = __builtin_amdgcn_readlane(v, l + 1);
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}

// A node as the wave holds it after ONE memory round trip (two loads in flight, whether the tree is in LDS or in HBM): lane l < 22
// has int l of the record -- n, internal flag, the nine chain indices, the ten children --, lanes 0..8 the nine positions.
struct NodeRegs { i32 hdr; i64 key; };
DEVFN NodeRegs cw_load(i32 *nd, int x, int lane)
{
	i32 *node = nd + x * BT_NODE_INTS;
	NodeRegs r;
	// (every lane loads: the lanes past the record's 22 header ints / nine positions repeat its last one -- nobody reads their copy, and a clamped index
	// costs one instruction where a guarded load costs an exec-mask region of six)
	r.hdr = node[lane < 22 ? lane : 21];
	r.key = cw_pos(node)[lane < BT_MAXK ? lane : BT_MAXK - 1];
	return r;
}
DEVFN int nr_n(const NodeRegs &r) { return __builtin_amdgcn_readlane(r.hdr, 0); }
DEVFN int nr_internal(const NodeRegs &r) { return __builtin_amdgcn_readlane(r.hdr, 1); }
DEVFN int nr_key(const NodeRegs &r, int i) { return __builtin_amdgcn_readlane(r.hdr, 2 + i); }
DEVFN int nr_child(const NodeRegs &r, int i) { return __builtin_amdgcn_readlane(r.hdr, 2 + BT_MAXK + i); }
// __kb_getp_aux (kbtree.h:117-131): nine lanes compare the nine keys, the lower bound is a popcount
DEVFN int cw_search(const NodeRegs &nr, i64 pos, int lane, int &r)
{
	const int n = nr_n(nr);
	const bool in = lane < n;
	const u64 m_in = wave_ballot(in);                         // (two compare masks and a scalar AND: the ballot of a conjunction is rebuilt from 0/1 values)
	const int lo = __popcll(wave_ballot(nr.key < pos) & m_in);    // keys are sorted: lower bound = number of smaller keys
	const bool eq = ((wave_ballot(nr.key == pos) & m_in) >> (lo & 63)) & 1;
	// straight-line selects instead of early returns (an empty node: lo == n == 0 gives -1, and r is not looked at)
	const bool past = lo == n;
	r = past ? 1 : eq ? 0 : -1;
	return past ? n - 1 : eq ? lo : lo - 1;
}
// kb_intervalp, lower side (kbtree.h:152-168).  The walk also remembers where it ended: when it reaches a leaf without meeting a
// full node, a following insertion of the same position (the seed did not merge into the chain found) would descend along exactly
// this path and split nothing, so it can be done on the leaf still held in registers (cw_insert_at) without a second descent.
struct LowerPath { int leaf, i; bool direct; NodeRegs nr; };
// The root node as the wave last loaded it.  Every look-up starts at the root, and in the HBM tier every node costs a dependent memory round trip (three to
// four per seed: root, inner node, leaf, chain record): the root's -- a third of the descent -- is saved as long as nobody has written to the root since.
// node = -1: nothing cached.  Writers call cw_touch() with the node they are about to modify.
struct RootCache { int node; NodeRegs nr; };
DEVFN void cw_touch(RootCache &rc, int x) { if (x == rc.node) rc.node = -1; }
DEVFN int cw_lower(i32 *nd, int root, i64 pos, int lane, u32 &visits, LowerPath &P, RootCache &rc)
{
	int x = root, low = -1, i;
	bool full = false, leaf;
	NodeRegs nr;
	for (;;) {
		if (x == rc.node) nr = rc.nr;
		else { nr = cw_load(nd, x, lane); if (x == root) { rc.node = x; rc.nr = nr; } }
		int r = 0;
		i = cw_search(nr, pos, lane, r);
		++visits;
		full = full || nr_n(nr) == BT_MAXK;
		leaf = !nr_internal(nr);
		if (i >= 0) low = nr_key(nr, i);
		if ((i >= 0 && r == 0) || leaf) break;             // an equal key, or the walk's end
		x = nr_child(nr, i + 1);
	}
	P.leaf = x; P.i = i; P.nr = nr; P.direct = leaf && !full;  // (the last node visited: only a leaf reached without a full node on the way serves the direct insertion)
	return low;
}
// the leaf step of __kb_putp_aux (kbtree.h:199-206) on a leaf already in registers: key k / position pos go in after entry i
DEVFN void cw_insert_at(i32 *nd, const LowerPath &P, int k, i64 pos, int lane, RootCache &rc)
{
	cw_touch(rc, P.leaf);
	i32 *X = nd + P.leaf * BT_NODE_INTS;
	const int i = P.i, n = nr_n(P.nr);
	if (lane > 2 + i && lane < 2 + n) X[lane + 1] = P.nr.hdr;
	if (lane > i && lane < n) cw_pos(X)[lane + 1] = P.nr.key;
	if (lane == 0) { X[2 + i + 1] = k; cw_pos(X)[i + 1] = pos; X[0] = n + 1; }
	wave_sync();
}
// __kb_split (kbtree.h:173-190): y = child i of x is full; its upper half moves to a new node z, its median key up into x
DEVFN void cw_split(i32 *nd, int &n_nodes, int x, int i, int y, int lane, RootCache &rc)
{
	cw_touch(rc, x); cw_touch(rc, y);
	const int z = n_nodes++;
	i32 *X = nd + x * BT_NODE_INTS, *Y = nd + y * BT_NODE_INTS, *Z = nd + z * BT_NODE_INTS;
	const NodeRegs xr = cw_load(nd, x, lane), yr = cw_load(nd, y, lane);
	const int yint = nr_internal(yr), xn = nr_n(xr);
	const i32 mkey = nr_key(yr, BT_T - 1);
	const i64 mpos = readlane_i64(yr.key, BT_T - 1);
	wave_sync();                                           // both nodes are in registers; now rewrite them
	// z: n, internal, keys 5..8 of y, (children 5..9 of y), positions 5..8 of y
	if (lane == 0) { Z[0] = BT_T - 1; Z[1] = yint; Y[0] = BT_T - 1; X[0] = xn + 1; X[2 + i] = mkey; cw_pos(X)[i] = mpos; X[2 + BT_MAXK + i + 1] = z; }
	if (lane >= 2 + BT_T && lane < 2 + BT_MAXK) Z[lane - BT_T] = yr.hdr;                                   // KEY(z, j) = KEY(y, j + t)
	if (yint && lane >= 2 + BT_MAXK + BT_T && lane < 2 + BT_MAXK + 2 * BT_T) Z[lane - BT_T] = yr.hdr;      // CH(z, j) = CH(y, j + t)
	if (lane >= BT_T && lane < BT_MAXK) cw_pos(Z)[lane - BT_T] = yr.key;
	// x: keys / positions i.. move up by one, children i+1.. move up by one
	if (lane >= 2 + i && lane < 2 + xn) X[lane + 1] = xr.hdr;
	if (lane >= i && lane < xn) cw_pos(X)[lane + 1] = xr.key;
	if (lane > 2 + BT_MAXK + i && lane <= 2 + BT_MAXK + xn) X[lane + 1] = xr.hdr;
	wave_sync();
}
// kb_putp / __kb_putp_aux (kbtree.h:191-224)
DEVFN void cw_insert(i32 *nd, int &n_nodes, int &root, int &height, int k, i64 pos, int lane, RootCache &rc)
{
	if (uni(nd[root * BT_NODE_INTS]) == BT_MAXK) {
		++height;
		const int s = n_nodes++;
		if (lane == 0) { nd[s * BT_NODE_INTS] = 0; nd[s * BT_NODE_INTS + 1] = 1; nd[s * BT_NODE_INTS + 2 + BT_MAXK] = root; }
		wave_sync();
		cw_split(nd, n_nodes, s, 0, root, lane, rc);
		root = s; rc.node = -1;
	}
	int x = root;
	for (;;) {
		i32 *X = nd + x * BT_NODE_INTS;
		NodeRegs xr = cw_load(nd, x, lane);
		int r = 0;
		if (!nr_internal(xr)) {
			const int i = cw_search(xr, pos, lane, r), n = nr_n(xr);
			cw_touch(rc, x);
			wave_sync();
			if (lane > 2 + i && lane < 2 + n) X[lane + 1] = xr.hdr;                  // keys i+1.. move up by one
			if (lane > i && lane < n) cw_pos(X)[lane + 1] = xr.key;
			if (lane == 0) { X[2 + i + 1] = k; cw_pos(X)[i + 1] = pos; X[0] = n + 1; }
			wave_sync();
			return;
		}
		int i = cw_search(xr, pos, lane, r) + 1;
		const int c = nr_child(xr, i);
		if (uni(nd[c * BT_NODE_INTS]) == BT_MAXK) {
			cw_split(nd, n_nodes, x, i, c, lane, rc);
			xr = cw_load(nd, x, lane);
			if (pos > readlane_i64(xr.key, i)) ++i;
		}
		x = nr_child(xr, i);
	}
}
// __kb_traverse (kbtree.h:336-358): in-order walk; the explicit stack lives in LDS (depth <= log_5 n + 1), a leaf's keys are
// written by as many lanes
DEVFN int cw_inorder(i32 *nd, int root, i32 *out, i32 *stk, int lane)
{
	int sp = 0, n = 0;
	if (lane == 0) { stk[0] = root; stk[1] = 0; }
	wave_sync();
	while (sp >= 0) {
		const int x = uni(stk[2 * sp]), st = uni(stk[2 * sp + 1]);
		i32 *X = nd + x * BT_NODE_INTS;
		const int xn = uni(X[0]);
		wave_sync();                                       // the stack entry has been read by every lane before lane 0 rewrites it
		if (!uni(X[1])) {
			if (lane < xn) out[n + lane] = X[2 + lane];
			n += xn; --sp;
			continue;
		}
		const int i = st >> 1;
		if (!(st & 1)) {                                  // descend into child i
			if (lane == 0) { stk[2 * sp + 1] = st | 1; stk[2 * sp + 2] = X[2 + BT_MAXK + i]; stk[2 * sp + 3] = 0; }
			++sp;
		} else if (i < xn) {                              // back from child i: emit key i, go on to child i + 1
			if (lane == 0) { out[n] = X[2 + i]; stk[2 * sp + 1] = 2 * (i + 1); }
			++n;
		} else --sp;
		wave_sync();
	}
	wave_sync();
	return n;
}

DEVFN int wave_excl_scan_add(int v, int lane) { (void)lane; return wave_incl_scan_add(v) - v; }      // (DPP steps, dev_extw.h: no shuffle addresses to keep in registers)

// One read.
__device__ void chain_read_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, unsigned char *lds, u64 &n_visits, u64 &n_recs, int &out_k, int &out_m)
{
	out_k = 0; out_m = 0;
	const int lane = threadIdx.x & 63;
	r = uni(r);
	const i64 off0 = B.off[r], off1 = B.off[r + 1];
	const int ns_ = B.seed_n[r], niv_ = B.intv_n[r];
	const i64 so_ = B.seed_off[r], ivo_ = B.intv_off[r], no_ = B.node_off[r];
	const int len = uni((int)(off1 - off0)), ns = uni(ns_), n_iv = uni(niv_);
	const i64 so = uni64(so_);
	if (ns == 0) {
		if (lane == 0) { B.chain_n[r] = 0; B.reg_off[r] = 0; B.reg_cap_r[r] = 0; B.reg_n_raw[r] = 0; B.reg_n[r] = 0; }
		return;
	}
	const RegionView R = region_of(B.slot_blob, so, ns);
	// ---- storage: the read's region of HBM; LDS for the traversal stack and (up to CW_PW_HBM_TIER chains) the weight sort's pairs ----
	i32 *nd = B.nodes + uni64(no_) * BT_NODE_INTS; ChainRec *ch = R.chain; int2 *pw = (int2*)R.srt; int4 *kinfo = R.kinfo; i32 *kept = R.kept, *ord = R.ord, *stk = (i32*)lds;
	if (ns <= CW_PW_HBM_TIER) pw = (int2*)(lds + CW_STACK_INTS * 4);   // (n <= n_ch <= ns)
	const i64 *pos = (const i64*)(B.slot_pos + so);
	const i32 *sqb = B.slot_qbeg + so, *sln = B.slot_len + so, *srid = B.slot_rid + so;
	i32 *next = R.next;
	// ---- mem_chain (bwamem.c:299-334): one B-tree look-up / insertion per seed, in seed order ----
	int n_nodes = 1, root = 0, n_ch = 0, height = 1;
	RootCache rc; rc.node = -1; rc.nr.hdr = 0; rc.nr.key = 0;
	if (lane == 0) { nd[0] = 0; nd[1] = 0; }
	wave_sync();
	u32 visits = 0, recs = 0;
	for (int base = 0; base < ns; base += 64) {
		const int li = base + lane;
		i64 v_rbeg = 0; int v_qb = 0, v_len = 0, v_rid = -1;
		if (li < ns) { v_rbeg = pos[li]; v_qb = sqb[li]; v_len = sln[li]; v_rid = srid[li]; }
		const int cnt = ns - base < 64 ? ns - base : 64;
		for (int t = 0; t < cnt; ++t) {
			const int s = base + t;
			const int rid = __builtin_amdgcn_readlane(v_rid, t);
			if (rid < 0) continue;
			const int qbeg = __builtin_amdgcn_readlane(v_qb, t), slen = __builtin_amdgcn_readlane(v_len, t);
			const i64 rbeg = readlane_i64(v_rbeg, t);
			bool add = true;
			LowerPath path; path.direct = false;
			if (n_ch) {
				const int lo = cw_lower(nd, root, rbeg, lane, visits, path, rc);
				if (lo >= 0) {   // test_and_merge (bwamem.c:216-237)
					ChainRec *c = ch + lo;
					++recs;
					// the record's sixteen ints by sixteen lanes, one load; its fields are then lane reads (ints 0-1 pos, 2-3 last_rbeg, 5 last, 6 first_qbeg,
					// 7 last_qbeg, 8 last_len, 10 rid)
					const i32 cv = ((const i32*)c)[lane & 15];
					const i64 c_pos = readlane_i64x(cv, 0), c_lrb = readlane_i64x(cv, 2);
					const int c_last = __builtin_amdgcn_readlane(cv, 5), c_fqb = __builtin_amdgcn_readlane(cv, 6), c_lqb = __builtin_amdgcn_readlane(cv, 7), c_ll = __builtin_amdgcn_readlane(cv, 8), c_rid = __builtin_amdgcn_readlane(cv, 10);
					wave_sync();                               // every lane holds the record before lane 0 may update it
					const i64 qend = c_lqb + c_ll, rend = c_lrb + c_ll;
					if (rid == c_rid) {
						if (qbeg >= c_fqb && qbeg + slen <= qend && rbeg >= c_pos && rbeg + slen <= rend) add = false;   // contained
						else if ((c_lrb < ix.l_pac || c_pos < ix.l_pac) && rbeg >= ix.l_pac) add = true;                 // other strand
						else {
							const i64 x = qbeg - c_lqb, y = rbeg - c_lrb;
							if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - c_ll < opt.max_chain_gap && y - c_ll < opt.max_chain_gap) {
								if (lane == 0) {
									next[c_last] = s; next[s] = -1;
									c->last = s; c->last_qbeg = qbeg; c->last_len = slen; c->last_rbeg = rbeg; ++c->n;
								}
								wave_sync();
								add = false;
							}
						}
					}
				}
			}
			if (add) {
				if (lane == 0) {
					ChainRec c;
					c.pos = rbeg; c.last_rbeg = rbeg; c.first = c.last = s; c.first_qbeg = c.last_qbeg = qbeg; c.last_len = slen;
					c.n = 1; c.rid = rid; c.w = 0; c.kept = 0; c.first_shadow = -1; c.is_alt = ix.ctg_alt[rid] ? 1 : 0;
					next[s] = -1;
					ch[n_ch] = c;
				}
				wave_sync();
				if (path.direct) cw_insert_at(nd, path, n_ch, rbeg, lane, rc);
				else cw_insert(nd, n_nodes, root, height, n_ch, rbeg, lane, rc);
				++n_ch; ++recs;
			}
		}
	}
	n_visits += visits; n_recs += recs;
	if (B.stats && lane == 0) {      // (diagnostics: reads by chains and by seeds -- bwagpu_debug_chain_hist)
		const int cb = n_ch / 16 < 31 ? n_ch / 16 : 31, sb = ns / 32 < 31 ? ns / 32 : 31;
		atomicAdd(&B.ctr->chain_hist[0][cb], 1ull); atomicAdd(&B.ctr->chain_seeds[0][sb], 1ull);
	}
	if (lane == 0) { B.chain_n[r] = 0; B.reg_off[r] = 0; B.reg_cap_r[r] = 0; B.reg_n_raw[r] = 0; B.reg_n[r] = 0; }
	if (n_ch == 0) return;
	// Fraction of the read covered by over-abundant seeds (bwamem.c:291-298).  The reference merges the intervals -- sorted by
	// start -- into runs and adds up the runs' lengths; that is the length of their union, and with M(k) the largest end before
	// interval k the union is the sum of max(0, end_k - max(start_k, M(k))): a prefix maximum, 64 intervals per step.
	float frac_rep;
	{
		const Intv3 *iv = B.intv + uni64(ivo_);
		int l_rep = 0, run_max = 0;
		for (int base = 0; base < n_iv; base += 64) {
			const int i = base + lane;
			int sb = 0, se = 0; bool use = false;
			if (i < n_iv) { const u64 info = iv[i].info; sb = (int)(info >> 32); se = (int)(u32)info; use = iv[i].x2 > (u64)opt.max_occ; }
			const int inc = wave_incl_scan_max(use ? se : 0);
			const int before = imax(wave_shift_up1(inc, 0), run_max);        // largest end among the earlier over-abundant intervals
			int add = 0;
			if (use) { const int from = sb > before ? sb : before; add = se > from ? se - from : 0; }
			l_rep += wave_sum(add);
			run_max = imax(run_max, __builtin_amdgcn_readlane(inc, 63));
		}
		frac_rep = (float)l_rep / len;
	}
	int n = cw_inorder(nd, root, ord, stk, lane);
	// ---- mem_chain_flt (bwamem.c:353-411): weights (lane per chain), drop light chains keeping the order ----
	int k = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		bool keep = false; int oi = 0;
		if (i < n) {
			oi = ord[i];
			ChainRec &c = ch[oi];
			i64 end = 0; int w = 0;                          // mem_chain_weight (bwamem.c:239-258)
			for (int s = c.first; s >= 0; s = next[s]) {
				const int qb = sqb[s], sl = sln[s];
				if (qb >= end) w += sl; else if (qb + sl > end) w += (int)(qb + sl - end);
				if (qb + sl > end) end = qb + sl;
			}
			const int wq = w; w = 0; end = 0;
			for (int s = c.first; s >= 0; s = next[s]) {
				const int sl = sln[s]; const i64 rb = pos[s];
				if (rb >= end) w += sl; else if (rb + sl > end) w += (int)(rb + sl - end);
				if (rb + sl > end) end = rb + sl;
			}
			if (wq < w) w = wq;
			if (w >= 1 << 30) w = (1 << 30) - 1;
			c.w = w; c.first_shadow = -1; c.kept = 0;
			keep = w >= opt.min_chain_weight;
		}
		const u64 m = __ballot(keep);
		wave_sync();                                       // the whole chunk has been read before its slots are overwritten
		if (keep) ord[k + __popcll(m & ((1ull << lane) - 1))] = oi;
		k += __popcll(m);
	}
	n = k;
	if (n == 0) return;
	wave_sync();
	// ks_introsort by weight (bwamem.c:367): {weight, index} pairs, literal comparison sequence, one lane
	for (int i = lane; i < n; i += 64) pw[i] = make_int2(ch[ord[i]].w, ord[i]);
	wave_sync();
	if (lane == 0) dev_introsort(pw, n, ChainWGreater());
	wave_sync();
	for (int i = lane; i < n; i += 64) ord[i] = pw[i].y;
	wave_sync();
	// pairwise filter (bwamem.c:369-393): chain i against the kept chains, 64 at a time
	int nk = 1;
	if (lane == 0) {
		ChainRec &c0 = ch[ord[0]];
		c0.kept = 3; kept[0] = 0; kinfo[0] = make_int4(c0.first_qbeg, c0.last_qbeg + c0.last_len, c0.w, c0.is_alt);
	}
	wave_sync();
	int4 v_ci = make_int4(0, 0, 0, 0); int v_oi = 0;      // {beg, end, weight, is_alt} and chain index of 64 chains at a time, one per lane
	for (int i = 1; i < n; ++i) {
		if (i == 1 || (i & 63) == 0) {
			const int ii = (i & ~63) + lane;
			if (ii < n) { v_oi = ord[ii]; const ChainRec &c = ch[v_oi]; v_ci = make_int4(c.first_qbeg, c.last_qbeg + c.last_len, c.w, c.is_alt); }
		}
		const int oi = __builtin_amdgcn_readlane(v_oi, i & 63);
		const int bi = __builtin_amdgcn_readlane(v_ci.x, i & 63), ei = __builtin_amdgcn_readlane(v_ci.y, i & 63);
		const int wi = __builtin_amdgcn_readlane(v_ci.z, i & 63), alti = __builtin_amdgcn_readlane(v_ci.w, i & 63);
		bool large_ovlp = false, dropped = false;
		for (int kb = 0; kb < nk && !dropped; kb += 64) {
			const int kk = kb + lane;
			bool ov = false, dr = false; int4 kj = make_int4(0, 0, 0, 0);
			if (kk < nk) {
				kj = kinfo[kk];
				const int bj = kj.x, ej = kj.y;
				const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
				if (e_min > b_max && (!(kj.w & 1) || alti)) {
					const int li = ei - bi, lj = ej - bj, min_l = li < lj ? li : lj;
					if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
						ov = true;
						dr = wi < kj.z * opt.drop_ratio && kj.z - wi >= opt.min_seed_len << 1;
					}
				}
			}
			const u64 mdr = __ballot(dr);
			const int limit = mdr ? __ffsll((unsigned long long)mdr) - 1 : 63;   // the reference stops at the first kept chain that drops chain i
			const bool eff = ov && lane <= limit;
			if (eff && !(kj.w & 2)) { ch[ord[kept[kk]]].first_shadow = i; kinfo[kk].w = kj.w | 2; }
			if (__ballot(eff)) large_ovlp = true;
			if (mdr) dropped = true;
		}
		if (!dropped) {
			if (lane == 0) { kept[nk] = i; kinfo[nk] = make_int4(bi, ei, wi, alti); ch[oi].kept = large_ovlp ? 2 : 3; }
			++nk;
		}
		wave_sync();
	}
	for (int i = lane; i < nk; i += 64) {
		const ChainRec &c = ch[ord[kept[i]]];
		if (c.first_shadow >= 0) ch[ord[c.first_shadow]].kept = 1;
	}
	wave_sync();
	{	// at most max_chain_extend chains of kind 1/2 are extended (bwamem.c:398-403)
		int cnt12 = 0, cut = n;
		for (int base = 0; base < n && cut == n; base += 64) {
			const int i = base + lane;
			const int kp = i < n ? ch[ord[i]].kept : 0;
			const bool f = kp == 1 || kp == 2;
			const u64 m = __ballot(f);
			const int before = cnt12 + __popcll(m & ((1ull << lane) - 1));
			const u64 hit = __ballot(f && before + 1 >= opt.max_chain_extend);
			if (hit) cut = base + __ffsll((unsigned long long)hit) - 1;
			cnt12 += __popcll(m);
		}
		for (int i = cut + lane; i < n; i += 64) if (ch[ord[i]].kept < 3) ch[ord[i]].kept = 0;
		wave_sync();
	}
	// ---- publish the kept chains: headers + seeds flattened chain by chain, in sorted order ----
	bwagpu_chain_t *oc = R.cchain;
	bwagpu_seed_t *os = R.cseed;
	int m_tot = 0; k = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		int kp = 0, cn = 0, oi = 0;
		if (i < n) { oi = ord[i]; kp = ch[oi].kept; cn = kp ? ch[oi].n : 0; }
		const u64 mk = __ballot(kp != 0);
		const int my_k = k + __popcll(mk & ((1ull << lane) - 1));
		const int my_m = m_tot + wave_excl_scan_add(cn, lane);
		if (kp) {
			const ChainRec &c = ch[oi];
			bwagpu_chain_t h;
			h.n_seeds = c.n; h.rid = c.rid; h.w = c.w; h.kept = c.kept; h.is_alt = c.is_alt; h.frac_rep = frac_rep; h.pos = c.pos;
			oc[my_k] = h;
			int m = my_m;
			for (int s = c.first; s >= 0; s = next[s]) {
				bwagpu_seed_t sd;
				sd.rbeg = pos[s]; sd.qbeg = sqb[s]; sd.len = sln[s]; sd.score = sd.len; sd.pad_ = 0;
				os[m++] = sd;
			}
		}
		k += __popcll(mk);
		m_tot = __builtin_amdgcn_readlane(my_m + cn, 63);
	}
	wave_sync();
	// the caller reserves the read's range of the region arena (one atomic per chunk of reads) and sets reg_off
	if (k && lane == 0) { B.chain_n[r] = k; B.reg_cap_r[r] = m_tot; }
	out_k = k; out_m = m_tot;
}

// Every read, heaviest first (B.order by seed count).  4 waves per workgroup, CW_LDS_BYTES of dynamic LDS per wave.
__global__ void __launch_bounds__(256) k_chain_wave(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	HIP_DYNAMIC_SHARED(unsigned char, cw_lds)
	const int lane = threadIdx.x & 63;
	unsigned char *lds = cw_lds + (size_t)(threadIdx.x >> 6) * CW_LDS_BYTES;
	u64 visits = 0, recs = 0, nch = 0;
	const long long n_items = (long long)B.n_reads;
	unsigned long long *cursor = &B.ctr->next_chain;
	const i32 *items = B.order;
	// Reads are drawn in chunks (one at a time at the heavy head of the list) and a read's range of the region arena is reserved once per chunk:
	// lane j keeps the j-th read's results.
	int step = 1;
	for (;;) {
		const long long b = wave_fetch_n(cursor, step);
		if (b >= n_items) break;
		const int cnt = (int)(b + step <= n_items ? step : n_items - b);
		if (b >= WQ_SINGLE) step = WQ_CHUNK;
		const int my_r = lane < cnt ? items[b + lane] : -1;
		int my_m = 0, my_k = 0;
		for (int j = 0; j < cnt; ++j) {
			const int r = __builtin_amdgcn_readlane(my_r, j);
			int kk = 0, mm = 0;
			chain_read_wave(ix, opt, B, r, lds, visits, recs, kk, mm);
			wave_sync();
			if (lane == j) { my_k = kk; my_m = mm; }
			nch += (u64)kk;
		}
		const int excl = wave_excl_scan_add(my_m, lane);
		const int total = __builtin_amdgcn_readlane(excl + my_m, 63);
		if (total > 0) {
			u64 roff = 0;
			if (lane == 0) roff = atomicAdd(&B.ctr->reg_used, (unsigned long long)total);
			roff = (u64)lane0_i64((i64)roff);
			if (my_k > 0) {
				if (roff + excl + my_m > (u64)B.reg_cap) { atomicOr(&B.ctr->overflow, 8ull); B.chain_n[my_r] = 0; }
				else B.reg_off[my_r] = (i64)(roff + excl);
			}
		}
	}
	if (B.stats && lane == 0) {
		atomicAdd(&B.ctr->n_chains, (unsigned long long)nch);
		atomicAdd(&B.ctr->bt_nodes, (unsigned long long)visits);
		atomicAdd(&B.ctr->chain_recs, (unsigned long long)recs);
	}
}
