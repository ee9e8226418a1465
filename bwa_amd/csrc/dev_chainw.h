// dev_chainw.h -- seeds -> chains (mem_chain, bwamem.c:277-342) and the chain filter (mem_chain_flt, bwamem.c:353-411),
// one wavefront per read, heaviest reads first, one kernel (k_chain_wave).
//
// Why not one lane per read (dev_chain.h, round 1): every lane walked its own B-tree and chain records, so each wave-wide load
// touched 64 different cache lines (measured: 18 GB of HBM-side traffic per launch for < 1 GB of algorithmic bytes), the
// pairwise chain filter of a read with hundreds of equally good chains (a read inside a repeat family: 640 chains, 200 k pair
// tests) ran as one serial dependent-load loop -- 74 ms for a single read, the whole kernel's tail -- and the waves of a batch
// advanced at the pace of their heaviest lane.
//
// The order-dependent part -- one look-up / insertion per seed, in seed order, with kbtree's semantics -- has two forms:
//   * register form (chain_seeds_lanes, round 5; 98.7 % of the bench batch's reads end here): a kbtree with distinct keys is a sorted set, so the read's
//     chains are a sorted array held one chain per lane, up to CL_SLOTS per lane; look-up = compare masks + popcount, test_and_merge on every lane's own
//     chain, insertion = a DPP shift of the lanes above the slot.  No memory access inside the loop but the seed-link stores.  See the comment there for
//     when it is exact (distinct keys, or a one-node tree) and when the read is handed on;
//   * tree form (the loop in chain_read_wave): kbtree literally, wave-uniform -- a node is fetched as one 160-byte row, its nine keys are compared by nine
//     lanes and the lower bound is a popcount of a ballot, node splits and shifts move all entries at once; tree and chain records in the read's own
//     region of HBM (RegionView), the root node -- where every look-up starts -- cached in registers (RootCache).
// Rounds 2-4 ran the tree form in three tiers (tree and records in LDS for reads that fit) on the premise that the tree's memory is what a look-up costs.
// It is not: the cost was the wave-uniform logic on the CU's one scalar unit, 5-7 us per seed in every tier, and the HBM form alone measured as fast as
// the three together (profiles/r05_chain_tiers.log); the register form is what moved it (24.9 -> 12.7 ms per million reads, r05_chain_regs_ab.log).
// Everything that is not order-dependent is lane-parallel: chain weights (lane per chain), the finish of the weight sort (every lane ranks its chain),
// the pairwise filter (64 kept chains per step; the reference's "stop at the first chain that drops this one" is a find-first-set on the ballot), the
// flattening of the kept chains.  LDS holds what the sort and the filter work on for reads of up to CW_FLT_LDS chains.
#pragma once
#include "dev_chain.h"
#include "dev_extw.h"

#define CW_STACK_INTS 32
// LDS bytes of one wave: the traversal stack; the chains in sorted order (up to CW_FLT_LDS of them); and an area that first holds the weight sort's
// {weight, index} pairs (up to CW_PW_LDS; a serial sort in HBM costs ~1 us per step), then the filter's arrays for up to CW_FLT_LDS chains: the kept
// chains' records (16 bytes), their first shadowed chain (4) and every chain's kind (4).  With these in LDS the filter loop -- one iteration per chain,
// each ending in a fence -- has no store to HBM to wait for.
#define CW_FLT_LDS 256
#define CW_AREA_BYTES (CW_FLT_LDS * 24)
#define CW_PW_LDS (CW_AREA_BYTES / 8)
#define CW_LDS_BYTES (CW_STACK_INTS * 4 + CW_FLT_LDS * 4 + CW_AREA_BYTES)

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
// The root node as the wave last loaded it.  Every look-up starts at the root, and in the tree form every node costs a dependent memory round trip (three to
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

// mem_chain (bwamem.c:299-334) with the "tree" in registers: the chains as an array sorted by position, ONE CHAIN PER LANE -- key and record both.
// A kbtree with distinct keys is a sorted set: kb_intervalp's lower bound is the largest key <= pos whatever the tree's shape, an insertion lands at the
// key's sorted place, and the in-order walk is the sorted order.  So a look-up is two compare masks and a popcount, test_and_merge is evaluated by every
// lane on its own chain (lane `li`'s answer is the one that counts), a merge is an exec-masked update in that lane, and an insertion shifts the lanes
// above the slot up by one (wave_shr DPP moves).  No node or chain record is read from memory inside the loop, and the work is vector work: the tree form
// spends 424 scalar instructions per seed on the one scalar unit the CU's four SIMDs share.
// With duplicate keys the tree's shape starts to matter -- which of two equal keys a look-up meets first, where the new one goes: while the tree is a
// single leaf (<= 9 keys, kbtree.h t = 5) that is still a sorted array with kbtree's rules (look-up: the first equal key; insertion: right after it), and
// beyond that the read is handed to the tree form below.
// More than 64 chains: CL_SLOTS arrays of 64 (sorted place p = lane p % 64 of slot p / 64).  The loop runs in the one-slot form while that holds the
// read's chains (most reads) and goes on in the CL_SLOTS form, state in place, when the 65th arrives; an insertion there shifts the slot it lands in
// above its lane and every slot above it whole, lane 63 of one slot moving into lane 0 of the next.
// A chain is seven registers: the position, and -- the read and its seed list being shorter than 65536 (the caller checks; longer ones take the tree form) --
// d = last_rbeg - pos (a merge needs rbeg >= last_rbeg and |x - y| <= w, so it grows by at most x + w per merge and stays below l_seq + n w, also checked),
// a = first_qbeg | last_qbeg << 16, b = last_len | n << 16, c = first | last << 16 (seed slots), rid.
#define CL_SLOTS 4
struct ChainLanes { i64 pos; u32 d, a, b, c; i32 rid; };
DEVFN i64 wave_shift_up1_i64(i64 v, i64 fill)
{
	const int lo = wave_shift_up1((int)(u32)(u64)v, (int)(u32)(u64)fill), hi = wave_shift_up1((int)(u32)((u64)v >> 32), (int)(u32)((u64)fill >> 32));
	return (i64)((u64)(u32)hi << 32 | (u32)lo);
}
DEVFN u32 wave_shift_up1_u32(u32 v, u32 fill) { return (u32)wave_shift_up1((int)v, (int)fill); }
DEVFN u32 readlane_u32(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
// returns 0: all seeds done; 1: out of room at seed s_next (state intact: go on with more slots, or start over in the tree form); -1: duplicate key in a tree of more than one node
template <int KS>
DEVFN int chain_seeds_lanes(const DevIndex &ix, const bwagpu_opt_t &opt, int ns, const i64 *pos, const i32 *sqb, const i32 *sln, const i32 *srid, i32 *next,
	int lane, u32 &recs, ChainLanes (&K)[CL_SLOTS], int &n_ch, bool &has_dup, int &s_next)
{
	const i64 l_pac = ix.l_pac;
	u64 m_in[KS];                                                    // the lanes of each array that hold a chain
#pragma unroll
	for (int j = 0; j < KS; ++j) { const int nj = n_ch - 64 * j; m_in[j] = nj >= 64 ? ~0ull : nj > 0 ? (1ull << nj) - 1 : 0ull; }
	for (int base = s_next & ~63; base < ns; base += 64) {
		const int li_ = base + lane;
		i64 v_rbeg = 0; int v_qb = 0, v_len = 0, v_rid = -1;
		if (li_ < ns) { v_rbeg = pos[li_]; v_qb = sqb[li_]; v_len = sln[li_]; v_rid = srid[li_]; }
		const int cnt = ns - base < 64 ? ns - base : 64;
		for (int t = s_next > base ? s_next - base : 0; t < cnt; ++t) {
			const int s = base + t;
			const int rid = __builtin_amdgcn_readlane(v_rid, t);
			if (rid < 0) continue;
			const int qbeg = __builtin_amdgcn_readlane(v_qb, t), slen = __builtin_amdgcn_readlane(v_len, t);
			const i64 rbeg = readlane_i64(v_rbeg, t);
			// lower bound: the number of smaller keys; eq: the key at that place equals rbeg -- the array is sorted, so that is the case when any key does
			int lo = 0; u64 any_eq = 0;
#pragma unroll
			for (int j = 0; j < KS; ++j) {
				lo += __popcll(wave_ballot(K[j].pos < rbeg) & m_in[j]);
				any_eq |= wave_ballot(K[j].pos == rbeg) & m_in[j];
			}
			const bool eq = any_eq != 0;
			const int li = eq ? lo : lo - 1;                             // the chain kb_intervalp hands to test_and_merge (-1: none)
			// test_and_merge (bwamem.c:216-237): the lanes of the chain's slot on their own chains, lane li % 64's answer counts
			bool hit = false;
#pragma unroll
			for (int j = 0; j < KS; ++j) {
				if (KS > 1 && (li >> 6) != j) continue;                  // (li = -1: no slot)
				ChainLanes &C = K[j];
				const bool me = lane == (li & 63) && li >= 0;
				const int c_fqb = (int)(C.a & 0xffff), c_lqb = (int)(C.a >> 16), c_ll = (int)(C.b & 0xffff);
				const i64 c_lrb = C.pos + (i64)C.d;
				const int qend = c_lqb + c_ll; const i64 rend = c_lrb + c_ll;
				const bool contained = qbeg >= c_fqb && qbeg + slen <= qend && rbeg >= C.pos && rbeg + slen <= rend;
				const bool strand = C.pos < l_pac && rbeg >= l_pac;      // (bwamem.c:227 also asks last_rbeg < l_pac: implied, last_rbeg >= pos)
				// bwamem.c:229: y >= 0, |x - y| <= w, x - last_len < max_chain_gap, y - last_len < max_chain_gap.  x is a difference of read offsets; a y that does
				// not fit 30 bits fails y - x <= w (the caller's check: w < 2^30 - 2^16), so from there on 32-bit arithmetic is exact
				const int x = qbeg - c_lqb; const i64 y64 = rbeg - c_lrb; const int y = (int)y64;
				const bool fits = (u64)y64 < (1ull << 30) && x - y <= opt.w && y - x <= opt.w && x - c_ll < opt.max_chain_gap && y - c_ll < opt.max_chain_gap;
				const bool mine = me && rid == C.rid;
				const bool hit_c = mine && contained, hit_m = mine && !contained && !strand && fits;
				if (hit_m) {
					next[C.c >> 16] = s; next[s] = -1;
					C.c = (C.c & 0xffff) | (u32)s << 16; C.a = (C.a & 0xffff) | (u32)qbeg << 16; C.b = ((C.b & 0xffff0000u) | (u32)slen) + 0x10000u; C.d = (u32)(u64)(rbeg - C.pos);
				}
				hit = wave_ballot(hit_c || hit_m) != 0;
			}
			recs += li >= 0;
			if (hit) continue;
			// a new chain, at sorted place li + 1
			if (n_ch >= BT_MAXK && (has_dup || eq)) return -1;
			if (n_ch == 64 * KS) { s_next = s; return 1; }
			has_dup = has_dup || eq;
#pragma unroll
			for (int j = 0; j < KS; ++j) if ((n_ch >> 6) == j) m_in[j] |= 1ull << (n_ch & 63);
			const int p = li + 1;
			const u32 new_a = (u32)qbeg | (u32)qbeg << 16, new_b = (u32)slen | 0x10000u, new_c = (u32)s | (u32)s << 16;
#pragma unroll
			for (int j = KS - 1; j >= 0; --j) {
				if (KS > 1 && ((p >> 6) > j || n_ch < 64 * j)) continue;   // slots below the place stay; slots above the last chain are empty
				ChainLanes &C = K[j];
				ChainLanes F; F.pos = 0; F.d = F.a = F.b = F.c = 0; F.rid = 0;   // what moves into lane 0: lane 63 of the slot below
				if (j > 0) {
					const ChainLanes &D = K[j > 0 ? j - 1 : 0];
					F.pos = readlane_i64(D.pos, 63); F.d = readlane_u32(D.d, 63); F.a = readlane_u32(D.a, 63); F.b = readlane_u32(D.b, 63); F.c = readlane_u32(D.c, 63);
					F.rid = __builtin_amdgcn_readlane(D.rid, 63);
				}
				ChainLanes U;
				U.pos = wave_shift_up1_i64(C.pos, F.pos); U.d = wave_shift_up1_u32(C.d, F.d); U.a = wave_shift_up1_u32(C.a, F.a); U.b = wave_shift_up1_u32(C.b, F.b);
				U.c = wave_shift_up1_u32(C.c, F.c); U.rid = wave_shift_up1(C.rid, F.rid);
				const bool here = (p >> 6) == j;                         // the new chain lands in this slot: lanes below its place stay
				const bool at = here && lane == (p & 63), up = !here || lane > (p & 63);
				C.pos = at ? rbeg : up ? U.pos : C.pos; C.d = at ? 0u : up ? U.d : C.d;
				C.a = at ? new_a : up ? U.a : C.a; C.b = at ? new_b : up ? U.b : C.b; C.c = at ? new_c : up ? U.c : C.c; C.rid = at ? rid : up ? U.rid : C.rid;
			}
			if (lane == 0) next[s] = -1;
			++n_ch; ++recs;
		}
	}
	s_next = ns;
	return 0;
}
// the seeds of one read through the register forms; the records the rest of the kernel works on are then written with chain index = sorted place, so the
// in-order list is the identity.  Returns the number of chains, or -1: nothing has been published, the caller starts over in the tree form.
__device__ int chain_seeds_regs(const DevIndex &ix, const bwagpu_opt_t &opt, int ns, const i64 *pos, const i32 *sqb, const i32 *sln, const i32 *srid, i32 *next,
	ChainRec *ch, i32 *ord, int lane, u32 &recs, int slots)
{
	ChainLanes K[CL_SLOTS];
#pragma unroll
	for (int j = 0; j < CL_SLOTS; ++j) { K[j].pos = 0; K[j].d = K[j].a = K[j].b = K[j].c = 0; K[j].rid = 0; }
	int n_ch = 0, s_next = 0; bool has_dup = false;
	int rc = chain_seeds_lanes<1>(ix, opt, ns, pos, sqb, sln, srid, next, lane, recs, K, n_ch, has_dup, s_next);
	if (rc == 1 && slots > 1) rc = chain_seeds_lanes<CL_SLOTS>(ix, opt, ns, pos, sqb, sln, srid, next, lane, recs, K, n_ch, has_dup, s_next);
	if (rc != 0) return -1;
#pragma unroll
	for (int j = 0; j < CL_SLOTS; ++j) {
		const int i = 64 * j + lane;
		if (i < n_ch) {
			const ChainLanes &C = K[j];
			ChainRec c;
			c.pos = C.pos; c.last_rbeg = C.pos + (i64)C.d; c.first = (i32)(C.c & 0xffff); c.last = (i32)(C.c >> 16); c.first_qbeg = (i32)(C.a & 0xffff); c.last_qbeg = (i32)(C.a >> 16);
			c.last_len = (i32)(C.b & 0xffff); c.n = (i32)(C.b >> 16); c.rid = C.rid; c.w = 0; c.kept = 0; c.first_shadow = -1; c.is_alt = ix.ctg_alt[C.rid] ? 1 : 0;
			ch[i] = c; ord[i] = i;
		}
	}
	wave_sync();
	return n_ch;
}

// stats runs: 10 ns ticks spent per phase, summed over the reads (row 2 of chain_hist: [0] register-form seed loop, [1] tree-form seed loop, [2] repeat fraction and
// in-order list, [3] weights, [4] sort, [5] pairwise filter, [6] publishing, [7] the longest read, [8] reads)
#define CW_PHASE(i) do { if (B.stats) { const long long t_ = wall_clock64(); ph[i] += (u64)(t_ - t_ph); t_ph = t_; } } while (0)
// One read.
__device__ void chain_read_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, unsigned char *lds, u64 &n_visits, u64 &n_recs, int &out_k, int &out_m, u64 (&ph)[9])
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
	// ---- storage: the read's region of HBM; LDS for the traversal stack and, for reads of up to CW_FLT_LDS chains, what the sort and the filter work on ----
	i32 *nd = B.nodes + uni64(no_) * BT_NODE_INTS; ChainRec *ch = R.chain; i32 *ord = R.ord, *stk = (i32*)lds;
	unsigned char *const area = lds + CW_STACK_INTS * 4 + CW_FLT_LDS * 4;
	const i64 *pos = (const i64*)(B.slot_pos + so);
	const i32 *sqb = B.slot_qbeg + so, *sln = B.slot_len + so, *srid = B.slot_rid + so;
	i32 *next = R.next;
	// ---- mem_chain (bwamem.c:299-334): one look-up / insertion per seed, in seed order; in registers while that is exact, else in the B-tree ----
	long long t_ph = B.stats ? wall_clock64() : 0; const long long t_read = t_ph;
	u32 visits = 0, recs = 0;
	// (the register form packs seed slots, read offsets and seed counts into 16 bits and a chain's reference span into 32)
	const bool packs = len < 65536 && ns < 65536 && opt.w < (1 << 30) - 65536 && (i64)ns * ((i64)opt.w + 1) + 65536 < (1ll << 31);
	int n_ch = B.chain_regs <= 0 || !packs ? -1 : chain_seeds_regs(ix, opt, ns, pos, sqb, sln, srid, next, ch, ord, lane, recs, B.chain_regs);
	const bool tree = n_ch < 0;
	CW_PHASE(0);
	int n_nodes = 1, root = 0, height = 1;
	if (tree) {
	n_ch = 0; recs = 0;
	RootCache rc; rc.node = -1; rc.nr.hdr = 0; rc.nr.key = 0;
	if (lane == 0) { nd[0] = 0; nd[1] = 0; }
	wave_sync();
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
	CW_PHASE(1);
	}
	n_visits += visits; n_recs += recs;
	if (B.stats && lane == 0) {      // (diagnostics: reads by chains and by seeds -- bwagpu_debug_chain_hist)
		const int cb = n_ch / 16 < 31 ? n_ch / 16 : 31, sb = ns / 32 < 31 ? ns / 32 : 31, f = tree ? 1 : 0;   // row 0: register form, row 1: tree form
		atomicAdd(&B.ctr->chain_hist[f][cb], 1ull); atomicAdd(&B.ctr->chain_seeds[f][sb], 1ull);
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
	int n = tree ? cw_inorder(nd, root, ord, stk, lane) : n_ch;
	CW_PHASE(2);
	// ---- mem_chain_flt (bwamem.c:353-411): weights (lane per chain), drop light chains keeping the order ----
	int2 *pw = n <= CW_PW_LDS && B.chain_flt_lds > 0 ? (int2*)area : (int2*)R.srt;
	int k = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		bool keep = false; int oi = 0, w_ = 0;
		if (i < n) {
			oi = ord[i];
			ChainRec &c = ch[oi];
			// mem_chain_weight (bwamem.c:239-258): the query-side and the reference-side coverage, one walk along the chain for both
			// (a chain of one seed weighs the seed's length: no walk -- in repeat-rich reads, where the chains are many, most are such)
			i64 end = 0, rend = 0; int wq = 0, w = 0;
			if (c.n == 1) wq = w = c.last_len;
			else for (int s = c.first; s >= 0; ) {
				const int qb = sqb[s], sl = sln[s]; const i64 rb = pos[s];
				s = next[s];
				if (qb >= end) wq += sl; else if (qb + sl > end) wq += (int)(qb + sl - end);
				if (qb + sl > end) end = qb + sl;
				if (rb >= rend) w += sl; else if (rb + sl > rend) w += (int)(rb + sl - rend);
				if (rb + sl > rend) rend = rb + sl;
			}
			if (wq < w) w = wq;
			if (w >= 1 << 30) w = (1 << 30) - 1;
			c.w = w; w_ = w;
			keep = w >= opt.min_chain_weight;
		}
		const u64 m = __ballot(keep);
		if (keep) pw[k + __popcll(m & ((1ull << lane) - 1))] = make_int2(w_, oi);   // what ks_introsort sorts (bwamem.c:367): {weight, index} pairs
		k += __popcll(m);
	}
	n = k;
	if (n == 0) return;
	wave_sync();
	CW_PHASE(3);
	// lane 0: the quicksort passes (dev_sort.h).  The insertion sort that ends ks_introsort is a stable sort of the arrangement they leave, and a stable
	// sort's result is its definition: a chain's place is the number of chains that are heavier, or equally heavy and ahead of it -- counted by every lane
	// for its own chain, the others' {weight, ~place} keys read lane by lane from registers.
	if (n >= 3) {
		if (lane == 0) dev_introsort<int2, ChainWGreater, false>(pw, n, ChainWGreater());
		wave_sync();
	}
	const bool flt_lds = n <= B.chain_flt_lds;         // (option chain_flt_lds: CW_FLT_LDS, or less to send ordinary reads down the HBM path in tests)
	i32 *sord = flt_lds ? (i32*)(lds + CW_STACK_INTS * 4) : ord;       // the chains in sorted order
	for (int x0 = 0; x0 < n; x0 += 64) {
		const int x = x0 + lane;
		const int2 e = x < n ? pw[x] : make_int2(0, 0);
		const u64 kx = (u64)(u32)e.x << 32 | (u32)~x;
		int place = 0;
		for (int c = 0; c < n; c += 64) {
			const int yy = c + lane;
			const u64 ky = yy < n ? ((u64)(u32)pw[yy].x << 32 | (u32)~yy) : 0ull;      // (0: greater than no key)
			const int cnt = n - c < 64 ? n - c : 64;
			for (int t = 0; t < cnt; ++t) place += (u64)readlane_i64((i64)ky, t) > kx;
		}
		if (x < n) sord[place] = e.y;
	}
	wave_sync();
	CW_PHASE(4);
	// pairwise filter (bwamem.c:369-393): chain i against the kept chains, 64 at a time.  kinfo: the kept chains' {beg, end, weight, is_alt | has a shadow << 1};
	// shadow: mem_chain_t::first of a kept chain, the first chain it shadows; kind: mem_chain_t::kept of every chain, by sorted place.
	int4 *kinfo = flt_lds ? (int4*)area : R.kinfo;
	i32 *shadow = flt_lds ? (i32*)(area + CW_FLT_LDS * 16) : (i32*)R.srt, *kind = flt_lds ? (i32*)(area + CW_FLT_LDS * 20) : R.kept;
	wave_sync();                                           // (the sort's pairs have been read)
	for (int i = lane; i < n; i += 64) kind[i] = i == 0 ? 3 : 0;
	int nk = 1;
	if (lane == 0) {
		const ChainRec &c0 = ch[sord[0]];
		kinfo[0] = make_int4(c0.first_qbeg, c0.last_qbeg + c0.last_len, c0.w, c0.is_alt); shadow[0] = -1;
	}
	wave_sync();
	int4 v_ci = make_int4(0, 0, 0, 0);                    // {beg, end, weight, is_alt} of 64 chains at a time, one per lane
	for (int i = 1; i < n; ++i) {
		if (i == 1 || (i & 63) == 0) {
			const int ii = (i & ~63) + lane;
			if (ii < n) { const ChainRec &c = ch[sord[ii]]; v_ci = make_int4(c.first_qbeg, c.last_qbeg + c.last_len, c.w, c.is_alt); }
		}
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
			if (eff && !(kj.w & 2)) { shadow[kk] = i; kinfo[kk].w = kj.w | 2; }
			if (__ballot(eff)) large_ovlp = true;
			if (mdr) dropped = true;
		}
		if (!dropped) {
			if (lane == 0) { kinfo[nk] = make_int4(bi, ei, wi, alti); shadow[nk] = -1; kind[i] = large_ovlp ? 2 : 3; }
			++nk;
		}
		wave_sync();
	}
	for (int i = lane; i < nk; i += 64) {
		const int fs = shadow[i];
		if (fs >= 0) kind[fs] = 1;
	}
	wave_sync();
	{	// at most max_chain_extend chains of kind 1/2 are extended (bwamem.c:398-403)
		int cnt12 = 0, cut = n;
		for (int base = 0; base < n && cut == n; base += 64) {
			const int i = base + lane;
			const int kp = i < n ? kind[i] : 0;
			const bool f = kp == 1 || kp == 2;
			const u64 m = __ballot(f);
			const int before = cnt12 + __popcll(m & ((1ull << lane) - 1));
			const u64 hit = __ballot(f && before + 1 >= opt.max_chain_extend);
			if (hit) cut = base + __ffsll((unsigned long long)hit) - 1;
			cnt12 += __popcll(m);
		}
		for (int i = cut + lane; i < n; i += 64) if (kind[i] < 3) kind[i] = 0;
		wave_sync();
	}
	CW_PHASE(5);
	// ---- publish the kept chains: headers + seeds flattened chain by chain, in sorted order ----
	bwagpu_chain_t *oc = R.cchain;
	bwagpu_seed_t *os = R.cseed;
	int m_tot = 0; k = 0;
	for (int base = 0; base < n; base += 64) {
		const int i = base + lane;
		int kp = 0, cn = 0, oi = 0;
		if (i < n) { oi = sord[i]; kp = kind[i]; cn = kp ? ch[oi].n : 0; }
		const u64 mk = __ballot(kp != 0);
		const int my_k = k + __popcll(mk & ((1ull << lane) - 1));
		const int my_m = m_tot + wave_excl_scan_add(cn, lane);
		if (kp) {
			const ChainRec &c = ch[oi];
			bwagpu_chain_t h;
			h.n_seeds = c.n; h.rid = c.rid; h.w = c.w; h.kept = kp; h.is_alt = c.is_alt; h.frac_rep = frac_rep; h.pos = c.pos;
			oc[my_k] = h;
			int m = my_m;
			if (c.n == 1) {                                    // the record says all there is to say about its one seed
				bwagpu_seed_t sd;
				sd.rbeg = c.pos; sd.qbeg = c.first_qbeg; sd.len = c.last_len; sd.score = sd.len; sd.pad_ = 0;
				os[m] = sd;
			} else for (int s = c.first; s >= 0; s = next[s]) {
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
	CW_PHASE(6);
	if (B.stats) { const u64 dt = (u64)(t_ph - t_read); ph[7] = dt > ph[7] ? dt : ph[7]; ++ph[8]; }
}

// Every read, heaviest first (B.order by seed count).  4 waves per workgroup, CW_LDS_BYTES of dynamic LDS per wave.
__global__ void __launch_bounds__(256, 5) k_chain_wave(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	HIP_DYNAMIC_SHARED(unsigned char, cw_lds)
	const int lane = threadIdx.x & 63;
	unsigned char *lds = cw_lds + (size_t)(threadIdx.x >> 6) * CW_LDS_BYTES;
	u64 visits = 0, recs = 0, nch = 0;
	u64 ph[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
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
			chain_read_wave(ix, opt, B, r, lds, visits, recs, kk, mm, ph);
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
#pragma unroll
		for (int i = 0; i < 9; ++i) { if (i == 7) atomicMax(&B.ctr->chain_hist[2][i], (unsigned long long)ph[i]); else atomicAdd(&B.ctr->chain_hist[2][i], (unsigned long long)ph[i]); }
	}
}
