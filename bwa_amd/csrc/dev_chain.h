// dev_chain.h -- seeds -> chains (mem_chain, bwamem.c:277-342) and the chain filter (mem_chain_flt,
// bwamem.c:353-411).  One lane per read; all state lives in the read's slot-space region.
#pragma once
#include "dev_fm.h"
#include "dev_sort.h"
#include "dev_seed.h"

// ---- B-tree over chain positions (kbtree.h as instantiated at bwamem.c:212-213: t = 5, <= 9 keys/node).
// Duplicate positions are legal and their in-order place depends on the node layout, so the structure is
// reproduced literally.  A node is a 160-byte record {n, internal, chain index[9], child[10],
// position[9]}: the positions are stored in the node so that one descent step costs one memory latency (all nine
// keys are fetched together and the lower bound is a branch-free count), not a pointer chase per comparison.
#define BT_T 5
#define BT_MAXK 9
struct BTree {
	i32 *nd;            // node pool of this read
	int n_nodes, root;
	DEVFN i32 &N(int x) { return nd[x * BT_NODE_INTS]; }
	DEVFN i32 &INT(int x) { return nd[x * BT_NODE_INTS + 1]; }
	DEVFN i32 &KEY(int x, int i) { return nd[x * BT_NODE_INTS + 2 + i]; }
	DEVFN i32 &CH(int x, int i) { return nd[x * BT_NODE_INTS + 2 + BT_MAXK + i]; }
	DEVFN i64 &POS(int x, int i) { return ((i64*)(nd + x * BT_NODE_INTS + 22))[i]; }
	DEVFN int alloc(int internal) { int x = n_nodes++; N(x) = 0; INT(x) = internal; return x; }
	// __kb_getp_aux (kbtree.h:117-131): index of the first key >= pos; *r = 0 if it equals pos, -1 if pos is smaller
	// (then the index before it is returned), 1 if every key is smaller (n-1 returned); -1 for an empty node.
	DEVFN int search(int x, i64 pos, int *r) {
		const i64 *pp = (const i64*)(nd + x * BT_NODE_INTS + 22);
		int n = N(x);
		i64 k0 = pp[0], k1 = pp[1], k2 = pp[2], k3 = pp[3], k4 = pp[4], k5 = pp[5], k6 = pp[6], k7 = pp[7], k8 = pp[8];
		if (n == 0) return -1;
		int lo = (n > 0 && k0 < pos) + (n > 1 && k1 < pos) + (n > 2 && k2 < pos) + (n > 3 && k3 < pos) + (n > 4 && k4 < pos)
			   + (n > 5 && k5 < pos) + (n > 6 && k6 < pos) + (n > 7 && k7 < pos) + (n > 8 && k8 < pos);   // keys are sorted: lower bound = #smaller
		if (lo == n) { *r = 1; return n - 1; }
		i64 kl = lo == 0 ? k0 : lo == 1 ? k1 : lo == 2 ? k2 : lo == 3 ? k3 : lo == 4 ? k4 : lo == 5 ? k5 : lo == 6 ? k6 : lo == 7 ? k7 : k8;
		*r = pos < kl ? -1 : 0;
		return *r < 0 ? lo - 1 : lo;
	}
	// kb_intervalp, lower side (kbtree.h:152-168)
	DEVFN int lower(i64 pos) {
		int x = root, low = -1;
		for (;;) {
			int r = 0, i = search(x, pos, &r);
			if (i >= 0 && r == 0) return KEY(x, i);
			if (i >= 0) low = KEY(x, i);
			if (!INT(x)) return low;
			x = CH(x, i + 1);
		}
	}
	// __kb_split (kbtree.h:173-190)
	DEVFN void split(int x, int i, int y) {
		int z = alloc(INT(y));
		N(z) = BT_T - 1;
		for (int j = 0; j < BT_T - 1; ++j) { KEY(z, j) = KEY(y, j + BT_T); POS(z, j) = POS(y, j + BT_T); }
		if (INT(y)) for (int j = 0; j < BT_T; ++j) CH(z, j) = CH(y, j + BT_T);
		N(y) = BT_T - 1;
		int xn = N(x);
		for (int j = xn; j > i; --j) CH(x, j + 1) = CH(x, j);
		CH(x, i + 1) = z;
		for (int j = xn - 1; j >= i; --j) { KEY(x, j + 1) = KEY(x, j); POS(x, j + 1) = POS(x, j); }
		KEY(x, i) = KEY(y, BT_T - 1); POS(x, i) = POS(y, BT_T - 1);
		N(x) = xn + 1;
	}
	// kb_putp / __kb_putp_aux (kbtree.h:191-224)
	DEVFN void insert(int k, i64 pos) {
		int r;
		if (N(root) == BT_MAXK) {
			int s = alloc(1);
			CH(s, 0) = root;
			split(s, 0, root);
			root = s;
		}
		int x = root;
		for (;;) {
			if (!INT(x)) {
				int i = search(x, pos, &r), n = N(x);
				for (int j = n - 1; j > i; --j) { KEY(x, j + 1) = KEY(x, j); POS(x, j + 1) = POS(x, j); }
				KEY(x, i + 1) = k; POS(x, i + 1) = pos; N(x) = n + 1;
				return;
			}
			int i = search(x, pos, &r) + 1;
			if (N(CH(x, i)) == BT_MAXK) {
				split(x, i, CH(x, i));
				if (pos > POS(x, i)) ++i;
			}
			x = CH(x, i);
		}
	}
	// __kb_traverse (kbtree.h:336-358): plain in-order walk with an explicit stack (depth <= log_5 n + 1)
	DEVFN int inorder(i32 *out) {
		int sx[24], si[24], sp = 0, n = 0;
		sx[0] = root; si[0] = 0;
		while (sp >= 0) {
			int x = sx[sp];
			if (!INT(x)) { for (int j = 0; j < N(x); ++j) out[n++] = KEY(x, j); --sp; continue; }
			int st = si[sp], i = st >> 1;
			if (!(st & 1)) { si[sp] = st | 1; ++sp; sx[sp] = CH(x, i); si[sp] = 0; continue; } // descend into child i
			if (i < N(x)) { out[n++] = KEY(x, i); si[sp] = 2 * (i + 1); } else --sp;             // back from child i
		}
		return n;
	}
};

struct ChainWGreater {   // flt_lt (bwamem.c:350): heavier chains first; elements are {weight, chain index} pairs so that the
	DEVFN bool operator()(const int2 &a, const int2 &b) const { return a.x > b.x; }   // sort touches no other memory
};

__device__ void chain_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r)
{
	int len = (int)(B.off[r + 1] - B.off[r]);
	int ns = B.seed_n[r], n_iv = B.intv_n[r];
	i64 so = B.seed_off[r];
	B.chain_n[r] = 0; B.reg_off[r] = 0; B.reg_cap_r[r] = 0; B.reg_n_raw[r] = 0; B.reg_n[r] = 0;
	if (ns == 0) return;
	const Intv3 *iv = B.intv + B.intv_off[r];
	// fraction of the read covered by over-abundant seeds (bwamem.c:291-298)
	int b = 0, e = 0, l_rep = 0;
	for (int i = 0; i < n_iv; ++i) {
		int sb = (int)(iv[i].info >> 32), se = (int)(u32)iv[i].info;
		if (iv[i].x2 <= (u64)opt.max_occ) continue;
		if (sb > e) { l_rep += e - b; b = sb; e = se; }
		else e = e > se ? e : se;
	}
	l_rep += e - b;
	float frac_rep = (float)l_rep / len;

	const RegionView R = region_of(B.slot_blob, so, ns);
	ChainRec *ch = R.chain;
	i32 *next = R.next;
	const u64 *pos = B.slot_pos + so;
	const i32 *sqb = B.slot_qbeg + so, *sln = B.slot_len + so, *srid = B.slot_rid + so;
	BTree bt; bt.nd = B.nodes + B.node_off[r] * BT_NODE_INTS; bt.n_nodes = 0;
	bt.root = bt.alloc(0);
	int n_ch = 0;
	// the per-seed inputs do not depend on the tree: fetch them one seed ahead of the (latency-bound) tree walk
	i64 nx_rbeg = (i64)pos[0]; int nx_qbeg = sqb[0], nx_len = sln[0], nx_rid = srid[0];
	for (int s = 0; s < ns; ++s) {
		const int qbeg = nx_qbeg, slen = nx_len, rid = nx_rid;
		const i64 rbeg = nx_rbeg;
		if (s + 1 < ns) { nx_rbeg = (i64)pos[s + 1]; nx_qbeg = sqb[s + 1]; nx_len = sln[s + 1]; nx_rid = srid[s + 1]; }
		if (rid < 0) continue;
		bool add = true;
		if (n_ch) {
			int lo = bt.lower(rbeg);
			if (lo >= 0) {   // test_and_merge (bwamem.c:216-237)
				ChainRec &c = ch[lo];
				i64 qend = c.last_qbeg + c.last_len, rend = c.last_rbeg + c.last_len;
				if (rid == c.rid) {
					if (qbeg >= c.first_qbeg && qbeg + slen <= qend && rbeg >= c.pos && rbeg + slen <= rend) add = false; // contained
					else if ((c.last_rbeg < ix.l_pac || c.pos < ix.l_pac) && rbeg >= ix.l_pac) add = true;            // other strand
					else {
						i64 x = qbeg - c.last_qbeg, y = rbeg - c.last_rbeg;
						if (y >= 0 && x - y <= opt.w && y - x <= opt.w && x - c.last_len < opt.max_chain_gap && y - c.last_len < opt.max_chain_gap) {
							next[c.last] = s; next[s] = -1;
							c.last = s; c.last_qbeg = qbeg; c.last_len = slen; c.last_rbeg = rbeg; ++c.n;
							add = false;
						}
					}
				}
			}
		}
		if (add) {
			ChainRec c;
			c.pos = rbeg; c.last_rbeg = rbeg; c.first = c.last = s; c.first_qbeg = c.last_qbeg = qbeg; c.last_len = slen;
			c.n = 1; c.rid = rid; c.w = 0; c.kept = 0; c.first_shadow = -1; c.is_alt = ix.ctg_alt[rid] ? 1 : 0;
			next[s] = -1;
			ch[n_ch] = c;
			bt.insert(n_ch, rbeg);
			++n_ch;
		}
	}
	if (n_ch == 0) return;
	i32 *ord = R.ord, *kept = R.kept;
	int n = bt.inorder(ord);

	// ---- mem_chain_flt (bwamem.c:353-411) ----
	int k = 0;
	for (int i = 0; i < n; ++i) {
		ChainRec &c = ch[ord[i]];
		// mem_chain_weight (bwamem.c:239-258)
		i64 end = 0; int w = 0;
		for (int s = c.first; s >= 0; s = next[s]) {
			int qb = sqb[s], sl = sln[s];
			if (qb >= end) w += sl; else if (qb + sl > end) w += (int)(qb + sl - end);
			if (qb + sl > end) end = qb + sl;
		}
		int wq = w; w = 0; end = 0;
		for (int s = c.first; s >= 0; s = next[s]) {
			int sl = sln[s]; i64 rb = (i64)pos[s];
			if (rb >= end) w += sl; else if (rb + sl > end) w += (int)(rb + sl - end);
			if (rb + sl > end) end = rb + sl;
		}
		if (wq < w) w = wq;
		if (w >= 1 << 30) w = (1 << 30) - 1;
		c.w = w; c.first_shadow = -1; c.kept = 0;
		if (w >= opt.min_chain_weight) ord[k++] = ord[i];
	}
	n = k;
	if (n == 0) return;
	{	// ks_introsort moves whole chain records; sorting {weight, index} pairs performs the same comparisons and moves
		int2 *pw = (int2*)R.srt;
		for (int i = 0; i < n; ++i) pw[i] = make_int2(ch[ord[i]].w, ord[i]);
		dev_introsort(pw, n, ChainWGreater());
		for (int i = 0; i < n; ++i) ord[i] = pw[i].y;
	}
	int nk = 0;
	int4 *kinfo = R.kinfo;
	{
		ChainRec &c0 = ch[ord[0]];
		c0.kept = 3; kept[0] = 0; kinfo[0] = make_int4(c0.first_qbeg, c0.last_qbeg + c0.last_len, c0.w, c0.is_alt); nk = 1;
	}
	for (int i = 1; i < n; ++i) {
		ChainRec &ci = ch[ord[i]];
		const int bi = ci.first_qbeg, ei = ci.last_qbeg + ci.last_len, wi = ci.w, alti = ci.is_alt;
		bool large_ovlp = false; int kk;
		// the pairwise test against every kept chain is quadratic for reads in repeats (hundreds of chains of similar
		// weight are all kept): stream the packed {beg,end,w,flags} records instead of chasing chain records
		bool dropped = false;
		for (kk = 0; kk < nk && !dropped; kk += 4) {
			// four records per step (one 64-byte line): the loop is a chain of dependent L1 round trips otherwise.
			// Reading up to three records past nk stays inside the slot arena (it is allocated with slack); they are ignored.
			const int4 k4[4] = { kinfo[kk], kinfo[kk + 1], kinfo[kk + 2], kinfo[kk + 3] };
			for (int u = 0; u < 4 && kk + u < nk; ++u) {
				const int4 kj = k4[u];
				const int bj = kj.x, ej = kj.y;
				const int b_max = bj > bi ? bj : bi, e_min = ej < ei ? ej : ei;
				if (e_min > b_max && (!(kj.w & 1) || alti)) {
					const int li = ei - bi, lj = ej - bj, min_l = li < lj ? li : lj;
					if (e_min - b_max >= min_l * opt.mask_level && min_l < opt.max_chain_gap) {
						large_ovlp = true;
						if (!(kj.w & 2)) { ch[ord[kept[kk + u]]].first_shadow = i; kinfo[kk + u].w = kj.w | 2; }
						if (wi < kj.z * opt.drop_ratio && kj.z - wi >= opt.min_seed_len << 1) { dropped = true; break; }
					}
				}
			}
		}
		if (!dropped) { kept[nk] = i; kinfo[nk] = make_int4(bi, ei, wi, alti); ++nk; ci.kept = large_ovlp ? 2 : 3; }
	}
	for (int i = 0; i < nk; ++i) {
		ChainRec &c = ch[ord[kept[i]]];
		if (c.first_shadow >= 0) ch[ord[c.first_shadow]].kept = 1;
	}
	int i = 0;
	for (k = 0; i < n; ++i) {
		int kp = ch[ord[i]].kept;
		if (kp == 0 || kp == 3) continue;
		if (++k >= opt.max_chain_extend) break;
	}
	for (; i < n; ++i) if (ch[ord[i]].kept < 3) ch[ord[i]].kept = 0;
	// ---- publish the kept chains: headers + seeds flattened chain by chain ----
	bwagpu_chain_t *oc = R.cchain;
	bwagpu_seed_t *os = R.cseed;
	int m = 0; k = 0;
	for (i = 0; i < n; ++i) {
		ChainRec &c = ch[ord[i]];
		if (c.kept == 0) continue;
		bwagpu_chain_t h;
		h.n_seeds = c.n; h.rid = c.rid; h.w = c.w; h.kept = c.kept; h.is_alt = c.is_alt; h.frac_rep = frac_rep; h.pos = c.pos;
		oc[k++] = h;
		for (int s = c.first; s >= 0; s = next[s]) {
			bwagpu_seed_t sd;
			sd.rbeg = (i64)pos[s]; sd.qbeg = sqb[s]; sd.len = sln[s]; sd.score = sd.len; sd.pad_ = 0;
			os[m++] = sd;
		}
	}
	B.chain_n[r] = k;
	if (k == 0) return;
	u64 roff = atomicAdd(&B.ctr->reg_used, (unsigned long long)m);
	if (roff + m > (u64)B.reg_cap) { atomicOr(&B.ctr->overflow, 8ull); B.chain_n[r] = 0; return; }
	B.reg_off[r] = (i64)roff; B.reg_cap_r[r] = m;
}

// ---- heavy-first ordering for the wave-per-read extension kernel ------------------------------------------------------
// Reads are binned by floor(log2(weight)); bins are laid out heaviest first.  Per-block LDS histograms keep the global
// atomics down to one per bin per block.
DEVFN int order_bin(int w) { return w <= 0 ? 0 : 32 - __clz(w); }   // 0, then 1 + floor(log2 w)

__global__ void __launch_bounds__(256) k_order_count(Batch B, const i32 *weight)
{
	__shared__ u32 hist[ORDER_BINS];
	if (threadIdx.x < ORDER_BINS) hist[threadIdx.x] = 0;
	__syncthreads();
	for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < B.n_reads; r += gridDim.x * blockDim.x)
		atomicAdd(&hist[order_bin(weight[r])], 1u);
	__syncthreads();
	if (threadIdx.x < ORDER_BINS && hist[threadIdx.x]) atomicAdd(&B.bin_cnt[threadIdx.x], hist[threadIdx.x]);
}
__global__ void k_order_scan(Batch B)
{
	if (blockIdx.x == 0 && threadIdx.x == 0) {
		u32 acc = 0;
		for (int b = ORDER_BINS - 1; b >= 0; --b) { u32 c = B.bin_cnt[b]; B.bin_cnt[ORDER_BINS + b] = acc; acc += c; }   // heaviest bin first
	}
}
// each block handles one contiguous chunk of reads: local histogram -> one global reservation per bin -> scatter
__global__ void __launch_bounds__(256) k_order_fill(Batch B, const i32 *weight, int chunk)
{
	__shared__ u32 hist[ORDER_BINS], base[ORDER_BINS];
	if (threadIdx.x < ORDER_BINS) hist[threadIdx.x] = 0;
	__syncthreads();
	int lo = blockIdx.x * chunk, hi = lo + chunk < B.n_reads ? lo + chunk : B.n_reads;
	for (int r = lo + threadIdx.x; r < hi; r += blockDim.x) atomicAdd(&hist[order_bin(weight[r])], 1u);
	__syncthreads();
	if (threadIdx.x < ORDER_BINS) { base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&B.bin_cnt[ORDER_BINS + threadIdx.x], hist[threadIdx.x]) : 0; hist[threadIdx.x] = 0; }
	__syncthreads();
	for (int r = lo + threadIdx.x; r < hi; r += blockDim.x) { int b = order_bin(weight[r]); B.order[base[b] + atomicAdd(&hist[b], 1u)] = r; }
}

__global__ void __launch_bounds__(256) k_chain(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	u64 nch = 0;
	// reads are drawn from a counter (in input order): the grid need not match the number of resident lanes, and a lane
	// that finishes a light read takes the next one instead of idling behind its wave's heaviest
	for (;;) {
		const int r = (int)atomicAdd(&B.ctr->next_chain, 1ull);
		if (r >= B.n_reads) break;
		chain_read(ix, opt, B, r); nch += B.chain_n[r];
	}
	if (B.stats) atomicAdd(&B.ctr->n_chains, (unsigned long long)nch);
}
