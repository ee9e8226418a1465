// dev_dedupw.h -- mem_sort_dedup_patch for long reads, one wavefront per read.
//
// Same decisions as dedup_read (dev_dedup.h), but the score-only global alignments of mem_patch_reg span thousands of
// bases for 10 kb reads and there are only as many reads as a few hundred wavefronts have lanes, so the DP runs
// row-parallel over the wave (ksw_global2 opens gaps from M, dev_cigar.h) with the band's {H, E} columns in an LDS ring;
// the order-dependent control logic runs wave-uniformly and lane 0 performs the sorts and the stores.
#pragma once
#include "dev_dedup.h"
#include "dev_extw.h"

struct DedupLds { i32 *hd, *e; const int8_t *mat; int ring_mask; i32 *H, *E; /* lane 0's HBM scratch columns (dev_ksw_global2_score) for bands wider than the ring */
	u8 *qbuf; int qcap; /* four-columns-per-lane form (option dedup_blk, the default): room for a patch alignment's query segment in alignment order */
	struct DdHot *hot; struct DdKey *keys; i32 *ord, *ord2; int par_cap; bwagpu_alnreg_t *tmp; /* dedup_read_par (dev_dedupp.h): per-region arrays in LDS for reads of up to par_cap regions (0: none), and the wave's staging area in HBM */ };

// ksw_global2 without traceback (ksw.c:540-619), columns in a ring of ring_mask+1 entries, lazily initialised
__device__ int wave_global2_score_ring(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, i64 t0, int tdir, int tlen,
									   int w, const DedupLds &L, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	i32 *hd = L.hd, *e_ = L.e; const int rm = L.ring_mask;
	int init_hi = -1, treg = 0;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) { int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		{	// first-row values (ksw.c:566-570) of the columns this row can reach for the first time
			const int hi = i + w + 2 < qlen ? i + w + 2 : qlen;
			if (hi > init_hi) {
				for (int j = init_hi + 1 + lane; j <= hi; j += 64) {
					hd[j & rm] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : DEV_NEG_INF);
					e_[j & rm] = DEV_NEG_INF;
				}
				init_hi = hi;
				wave_sync();
			}
		}
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : DEV_NEG_INF;
		int carry = I32_MIN, bnd = 0;
		cells += (u64)(end > beg ? end - beg : 0);
		for (int b = beg; b < end; b += 64) {
			const int j = b + lane; const bool act = j < end;
			int dg = hd[j & rm]; const int ec = e_[j & rm];
			const int qc = j < qlen ? (int)q[q0 + j * qdir] : 4;
			const int sc = L.mat[tb * 5 + qc];
			const int bnd_next = hd[(b + 64) & rm];
			if (b != beg && lane == 0) dg = bnd;
			wave_sync();
			const int m = dg + sc;
			const int a = act ? m - oe_ins + j * e_ins : I32_MIN;
			const int inc = wave_incl_scan_max(a);
			const int exc = imax(wave_shift_up1(inc, I32_MIN), carry);
			int f = DEV_NEG_INF - (j - beg) * e_ins;
			if (j > beg && act) f = imax(f, exc - (j - 1) * e_ins);
			int h = m >= ec ? m : ec;
			if (h < f) h = f;
			const int t = m - oe_del; int en = ec - e_del; if (en < t) en = t;
			if (act) { e_[j & rm] = en; hd[(j + 1) & rm] = h; }
			if (b == beg && lane == 0) hd[beg & rm] = h1_init;
			carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
			bnd = bnd_next;
			wave_sync();
		}
		if (lane == 0) e_[end & rm] = DEV_NEG_INF;
		wave_sync();
	}
	const int score = hd[qlen & rm];
	wave_sync();
	return score;
}

// The same with FOUR adjacent columns per lane (option dedup_blk; the default since round 4: 381 -> 207 ms per 6000 x 10 kb reads, BENCH_r03 variants).  The patch alignments of a 10 kb read run in bands
// of 200-800 columns, and in the form above every 64 of them cost a pass -- two ordering points, an LDS round trip and a six-step scan, each
// waiting for the one before at one wave per SIMD.  Here a pass covers 256 columns: a lane computes its four diagonal terms, a local prefix
// maximum of their insertion starts, ONE wave scan over the lanes' totals, then F, H and E of its columns (ksw.c:587-603).  Blocks are
// aligned to multiples of four columns in the ring, so a lane reads {H(i-1, j-1)}, {E} as two 16-byte LDS reads and writes them back the
// same way: ring slot j holds the diagonal for column j, which after this row is H(i, j-1) -- the lane's own h shifted by one column, with
// the first slot filled from the lane below (a lane shift).  No lane touches another lane's slots within a row: one ordering point per row.
// Slots a block covers outside [beg, end] receive defined but meaningless values: left of the band they are dead (beg never decreases),
// right of `end` every slot is rewritten -- as H[end] = h1, E[end] = -inf of a later row, or by the lazy first-row initialisation -- before
// a row reads it, and the ring is wider than the band plus these margins (2 w + 132 columns), so they alias no live slot.
__device__ int wave_global2_score_ring_blk(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen, i64 t0, int tdir, int tlen,
										   int w, const DedupLds &L, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	i32 *hd = L.hd, *e_ = L.e; const int rm = L.ring_mask;
	int init_hi = -1, treg = 0;
	for (int j = lane; j < qlen; j += 64) L.qbuf[j] = q[q0 + j * qdir];       // the segment in alignment order (the caller checked L.qcap >= qlen)
	wave_sync();
	const u8 *qs = L.qbuf;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) { int ii = i + lane; treg = ii < tlen ? ref_base(ix, t0 + (i64)ii * tdir) : 0; }
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		const int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		{	// first-row values (ksw.c:566-570) of the columns this row can reach for the first time
			const int hi = i + w + 2 < qlen ? i + w + 2 : qlen;
			if (hi > init_hi) {
				for (int j = init_hi + 1 + lane; j <= hi; j += 64) {
					hd[j & rm] = j == 0 ? 0 : (j <= w ? -(o_ins + e_ins * j) : DEV_NEG_INF);
					e_[j & rm] = DEV_NEG_INF;
				}
				init_hi = hi;
				wave_sync();
			}
		}
		const int h1_init = beg == 0 ? -(o_del + e_del * (i + 1)) : DEV_NEG_INF;
		const int8_t *mrow = L.mat + tb * 5;
		int carry = I32_MIN, hprev = h1_init;
		cells += (u64)(end > beg ? end - beg : 0);
		for (int b = beg & ~3; b <= end; b += 256) {            // (<=: slot `end` is written too, ksw.c:605)
			const int j0 = b + 4 * lane, p = j0 & rm;          // the ring's size and j0 are multiples of four: the block is contiguous and 16-byte aligned
			const int4 dg4 = *(const int4*)&hd[p], e4 = *(const int4*)&e_[p];
			const u32 qw = j0 < qlen ? *(const u32*)&qs[j0] : 0x04040404u;     // (bytes of the word at or past qlen are not the segment's: masked below)
			const int dgv[4] = { dg4.x, dg4.y, dg4.z, dg4.w }, ev[4] = { e4.x, e4.y, e4.z, e4.w };
			int m[4], pre[4], run = I32_MIN;
			#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int j = j0 + c; const bool act = j >= beg && j < end;
				const int qc = j < qlen ? (int)((qw >> (8 * c)) & 255u) : 4;
				m[c] = wadd(dgv[c], mrow[qc]);                 // (slots outside the band hold anything: wrap-around arithmetic, results discarded)
				pre[c] = run;
				run = imax(run, act ? wadd(wsub(m[c], oe_ins), j * e_ins) : I32_MIN);
			}
			const int inc = wave_incl_scan_max(run);
			const int exl = imax(wave_shift_up1(inc, I32_MIN), carry);     // best insertion start among the columns of the lanes below and of earlier passes
			int hv[4], en[4];
			#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int j = j0 + c; const bool act = j >= beg && j < end;
				int f = DEV_NEG_INF - (j - beg) * e_ins;
				if (j > beg && act) f = imax(f, imax(exl, pre[c]) - (j - 1) * e_ins);
				int h = m[c] >= ev[c] ? m[c] : ev[c];
				if (h < f) h = f;
				const int t = wsub(m[c], oe_del); int e2 = wsub(ev[c], e_del); if (e2 < t) e2 = t;
				hv[c] = act ? h : h1_init;                     // (left of the band: slot `beg` receives h1, ksw.c:578)
				en[c] = act ? e2 : DEV_NEG_INF;                // (column `end`: E = -inf, ksw.c:605)
			}
			const int hl = wave_shift_up1(hv[3], hprev);         // H(i, j0 - 1): the lane below's last column; lane 0: the pass before, or h1
			if (j0 <= end) {
				*(int4*)&hd[p] = make_int4(hl, hv[0], hv[1], hv[2]);
				*(int4*)&e_[p] = make_int4(en[0], en[1], en[2], en[3]);
			}
			carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
			hprev = __builtin_amdgcn_readlane(hv[3], 63);
		}
		wave_sync();
	}
	const int score = hd[qlen & rm];
	wave_sync();
	return score;
}

// bwa_gen_cigar2 in score-only mode (bwa.c:148-194).  A band that does not fit the ring (rare: the length difference of the two
// segments exceeds 4 * opt.w) is computed by lane 0 alone with its columns in HBM scratch.
template <bool BLK = false> __device__ int wave_global_score(const DevIndex &ix, const bwagpu_opt_t &opt, int w_, int l_query, const u8 *query, i64 rb, i64 re,
								 const DedupLds &L, u64 &calls, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const i64 l_pac = ix.l_pac;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	const int rlen = (int)(re - rb), rev = rb >= l_pac;
	const int q0 = rev ? l_query - 1 : 0, qdir = rev ? -1 : 1; const i64 t0 = rev ? re - 1 : rb; const int tdir = rev ? -1 : 1;
	if (l_query == rlen && w_ == 0) {
		int s = 0;
		for (int i = lane; i < l_query; i += 64) s += opt.mat[ref_base(ix, t0 + (i64)i * tdir) * 5 + query[q0 + i * qdir]];
		s = wave_sum(s);
		return s;
	}
	int max_ins = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
	int max_del = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
	int mg = max_ins > max_del ? max_ins : max_del, dl = rlen - l_query;
	if (dl < 0) dl = -dl;
	if (mg < 1) mg = 1;
	int w = (mg + dl + 1) >> 1; if (w > w_) w = w_;
	if (w < dl + 3) w = dl + 3;
	++calls;
	if (2 * w + 4 + 128 > L.ring_mask + 1) {
		int sc = 0;
		if (lane == 0) sc = dev_ksw_global2_score(ix, opt, query, q0, qdir, l_query, t0, tdir, rlen, w, L.H, L.E, cells);
		sc = __builtin_amdgcn_readlane(sc, 0);
		wave_sync();
		return sc;
	}
	if (BLK && L.qcap >= l_query) return wave_global2_score_ring_blk(ix, opt, query, q0, qdir, l_query, t0, tdir, rlen, w, L, cells);
	return wave_global2_score_ring(ix, opt, query, q0, qdir, l_query, t0, tdir, rlen, w, L, cells);
}

// mem_patch_reg (bwamem.c:432-461); all arguments wave-uniform
template <bool BLK = false> __device__ int wave_patch_reg(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *query, const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b,
							  int *w_out, const DedupLds &L, u64 &calls, u64 &cells)
{
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb)); if (w < 0) w = -w;
	double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb); if (r < 0.) r = -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > opt.w << 1 || r >= 0.05f) return 0; }
	else if (w > opt.w << 2 || r >= 0.05f * 2) return 0;
	w += a.w + b.w;
	if (w > opt.w << 2) w = opt.w << 2;
	int score = wave_global_score<BLK>(ix, opt, w, b.qe - a.qb, query + a.qb, a.rb, b.re, L, calls, cells);
	int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	*w_out = w;
	return score;
}

// a wave-uniform copy of a region record (every lane loads the same address)
DEVFN bwagpu_alnreg_t uni_reg(const bwagpu_alnreg_t *p)
{
	bwagpu_alnreg_t r = *p;
	r.rb = uni64(r.rb); r.re = uni64(r.re); r.qb = uni(r.qb); r.qe = uni(r.qe); r.rid = uni(r.rid); r.score = uni(r.score); r.w = uni(r.w);
	return r;
}

// mem_sort_dedup_patch (bwamem.c:463-515) for one read.  Every store by lane 0 is bracketed by wave_sync: the other lanes read
// the same records to take the same branches.
template <bool BLK = false> __device__ void dedup_read_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, const DedupLds &L, u64 &calls, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	int n = uni(B.reg_n_raw[r]);
	bwagpu_alnreg_t *a = B.regs + uni64(B.reg_off[r]);
	const u8 *query = B.seq + uni64(B.off[r]);
	if (B.regs_raw) { for (int i = lane; i < n; i += 64) B.regs_raw[B.reg_off[r] + i] = a[i]; }
	if (n > 1) {
		wave_sync();
		if (lane == 0) {
			RegKey *keys = n >= DEDUP_KEYSORT_MIN ? (RegKey*)region_of(B.slot_blob, B.seed_off[r], B.seed_n[r]).chain : nullptr;     // (see dev_sort_regs_by_key)
			if (keys) dev_sort_regs_by_key(a, n, keys, true, KeyEndLess()); else dev_introsort(a, n, RegEndLess());
			for (int i = 0; i < n; ++i) a[i].n_comp = 1;
		}
		wave_sync();
		for (int i = 1; i < n; ++i) {
			bwagpu_alnreg_t p = uni_reg(&a[i]);
			{
				const bwagpu_alnreg_t pr = uni_reg(&a[i - 1]);
				if (p.rid != pr.rid || p.rb >= pr.re + opt.max_chain_gap) continue;
			}
			bool p_dirty = false;
			for (int j = i - 1; j >= 0; --j) {
				bwagpu_alnreg_t q = uni_reg(&a[j]);
				if (!(p.rid == q.rid && p.rb < q.re + opt.max_chain_gap)) break;
				i64 orr, oq, mr, mq; int score, w = 0;
				if (q.qe == q.qb) continue;
				orr = q.re - p.rb;
				oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
				mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				if (orr > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
					if (p.score < q.score) { p.qe = p.qb; p_dirty = true; break; }
					else { wave_sync(); if (lane == 0) a[j].qe = q.qb; wave_sync(); }
				} else if (q.rb < p.rb) {
					score = wave_patch_reg<BLK>(ix, opt, query, q, p, &w, L, calls, cells);
					if (score > 0) {
						p.n_comp += q.n_comp + 1;
						if (q.seedcov > p.seedcov) p.seedcov = q.seedcov;
						if (q.sub > p.sub) p.sub = q.sub;
						if (q.csub > p.csub) p.csub = q.csub;
						p.qb = q.qb; p.rb = q.rb;
						p.truesc = p.score = score;
						p.w = w;
						p_dirty = true;
						wave_sync();
						if (lane == 0) a[j].qb = q.qe;
						wave_sync();
					}
				}
			}
			if (p_dirty) { wave_sync(); if (lane == 0) a[i] = p; wave_sync(); }
		}
		wave_sync();
		if (lane == 0) {
			int m = 0;
			for (int i = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
			n = m;
			if (n >= DEDUP_KEYSORT_MIN) dev_sort_regs_by_key(a, n, (RegKey*)region_of(B.slot_blob, B.seed_off[r], B.seed_n[r]).chain, false, KeyBestLess()); else dev_introsort(a, n, RegBestLess());
			for (int i = 1; i < n; ++i)
				if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
			m = n > 0 ? 1 : 0;
			for (int i = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
			n = m;
		}
		n = __builtin_amdgcn_readlane(n, 0);
		wave_sync();
	}
	if (lane == 0) {
		for (int i = 0; i < n; ++i)   // bwamem.c:1111-1115
			if (a[i].rid >= 0 && ix.ctg_alt[a[i].rid]) a[i].is_alt = 1;
		B.reg_n[r] = n;
	}
	wave_sync();
}

#include "dev_dedupp.h"

// One wavefront per read.
// BLK (option dedup_blk, default on): patch alignments with four columns per lane
// LIST (short-read batches): the reads k_dedup handed over in Batch::dd_list -- list 0: from the front (n_dd_heavy reads of up to Batch::dd_stage_cap
// regions), list 1: from the back (n_dd_big reads with more) -- through dedup_read_par when the LDS arrays hold the read's regions (par_cap), one
// wave per workgroup.
#define DDW_PAR_BYTES(cap) ((size_t)(cap) * DDP_LDS_PER_REG)
template <bool BLK = false, bool LIST = false> __global__ void __launch_bounds__(256) k_dedup_wave(DevIndex ix, bwagpu_opt_t opt, Batch B, int ring_cols, int q_cap, int par_cap, int list, bwagpu_alnreg_t *par_tmp)
{
	HIP_DYNAMIC_SHARED(unsigned char, ddw_lds)
	const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
	unsigned char *base = ddw_lds + (size_t)wave_in_blk * (8 * ring_cols + 32 + q_cap + DDW_PAR_BYTES(par_cap));
	DedupLds L;
	L.hd = (i32*)base; L.e = L.hd + ring_cols; L.ring_mask = ring_cols - 1;
	int8_t *m = (int8_t*)(base + (size_t)8 * ring_cols);
	if (lane < 25) m[lane] = opt.mat[lane];
	L.mat = m;
	L.qbuf = base + (size_t)8 * ring_cols + 32; L.qcap = q_cap;
	L.hot = (DdHot*)(L.qbuf + q_cap); L.keys = (DdKey*)(L.hot + par_cap); L.ord = (i32*)(L.keys + par_cap); L.ord2 = L.ord + par_cap; L.par_cap = par_cap;      // (q_cap is a multiple of 16)
	L.tmp = par_tmp + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_blk) * par_cap;
	{
		const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_blk;
		L.H = B.dp_h + wave * (B.max_len + 2) * DPS; L.E = B.dp_e + wave * (B.max_len + 2) * DPS;
	}
	wave_sync();
	// The listed reads' waves are few, short and latency-bound (a chain of LDS round trips per region), and with three batches on the chip they share their
	// SIMDs with the throughput kernels of the other two: at raised issue priority they get their instructions in when they are ready (option dedup_prio).
	if (LIST && B.dd_prio) __builtin_amdgcn_s_setprio(3);
	u64 calls = 0, cells = 0, nreg = 0;
	for (;;) {
		const long long k = wave_fetch(!LIST ? &B.ctr->next_dedup : (list ? &B.ctr->next_dd_big : &B.ctr->next_dd_heavy));
		if (k >= (!LIST ? (long long)B.n_reads : (long long)(list ? B.ctr->n_dd_big : B.ctr->n_dd_heavy))) break;
		const int r = !LIST ? (int)k : uni(list ? B.dd_list[B.n_reads - 1 - k] : B.dd_list[k]);
		const long long t_0 = B.stats ? wall_clock64() : 0; const u64 c_0 = calls, x_0 = cells;
		bool done = false;
		if (LIST && uni(B.reg_n_raw[r]) <= L.par_cap) done = dedup_read_par<BLK>(ix, opt, B, r, L, calls, cells);
		if (!done) dedup_read_wave<BLK>(ix, opt, B, r, L, calls, cells);
		if (B.stats && lane == 0) {
			const long long dt = wall_clock64() - t_0;
			const int bin = dt > 0 ? (64 - __clzll(dt) < 31 ? 64 - __clzll(dt) : 31) : 0;
			atomicAdd(&B.ctr->wave_hist[1][bin], 1ull); atomicAdd(&B.ctr->wave_hist[1][32 + bin], (unsigned long long)(calls - c_0)); atomicAdd(&B.ctr->wave_hist[1][64 + bin], (unsigned long long)((cells - x_0) >> 10));
		}
		nreg += B.reg_n[r];
	}
	if (B.stats && lane == 0) {
		atomicAdd(&B.ctr->glb_calls, (unsigned long long)calls);
		atomicAdd(&B.ctr->glb_cells, (unsigned long long)cells);
		atomicAdd(&B.ctr->n_regs, (unsigned long long)nreg);
	}
}

// ---- mem_flt_chained_seeds (bwamem.c:624-645) for long reads: one wavefront per read, one lane per seed --------------------
// The local re-alignments of a read's seeds are independent of each other; only the compaction that follows is ordered.
// (seedsw_read, dev_seedsw.h, is the lane-per-read form of the same logic.)
// CB: 0 = DP rows in HBM scratch (H, E); 8 / 16 = everything in LDS (dev_local_score_lds: HE, Q, rows)
template <int CB> __device__ void seedsw_read_wave(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, i32 *H, i32 *E, void *HE, u8 *Q, const u64 *rows, u64 &calls, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int n_ch = uni(B.chain_n[r]);
	if (n_ch == 0) return;
	const int l_query = uni((int)(B.off[r + 1] - B.off[r]));
	const int min_hsp = uni(B.seedsw_minhsp[l_query]);
	if (min_hsp < 0) return;
	const u8 *query = B.seq + uni64(B.off[r]);
	const i64 so = uni64(B.seed_off[r]), l_pac = ix.l_pac;
	const RegionView R = region_of(B.slot_blob, so, uni(B.seed_n[r]));
	bwagpu_chain_t *chains = R.cchain;
	bwagpu_seed_t *seeds = R.cseed;
	int S = 0;
	for (int ci = 0; ci < n_ch; ++ci) S += uni(chains[ci].n_seeds);
	for (int t = lane; t < S; t += 64) {
		bwagpu_seed_t s = seeds[t];
		int sc = -1;
		if (s.len < 200) {
			int qb = s.qbeg - 50, qe = s.qbeg + s.len + 50;
			i64 rb = s.rbeg - 50, re = s.rbeg + s.len + 50, mid = (s.rbeg + s.rbeg + s.len) >> 1;
			if (qb < 0) qb = 0;
			if (qe > l_query) qe = l_query;
			if (rb < 0) rb = 0;
			if (re > l_pac << 1) re = l_pac << 1;
			if (rb < l_pac && l_pac < re) { if (mid < l_pac) re = l_pac; else rb = l_pac; }
			if (qe - qb < 200 && re - rb < 200) {
				int is_rev; int rid = dev_pos2rid(ix, dev_depos(ix, mid, &is_rev));   // bns_fetch_seq clamp
				i64 fb = ix.ctg_off[rid], fe = fb + ix.ctg_len[rid];
				if (is_rev) { i64 t2 = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t2; }
				if (rb < fb) rb = fb;
				if (re > fe) re = fe;
				if (CB) sc = dev_local_score_lds<CB ? CB : 8>(ix, opt, rows, query + qb, qe - qb, rb, (int)(re - rb), (typename SwCell<CB ? CB : 8>::T*)HE, Q, cells);
				else sc = dev_local_score(ix, opt, query + qb, qe - qb, rb, (int)(re - rb), H, E, cells);
				++calls;
			}
		}
		seeds[t].score = sc;
	}
	wave_sync();
	if (lane == 0) {
		int sbeg = 0, m = 0;
		for (int ci = 0; ci < n_ch; ++ci) {
			const int n = chains[ci].n_seeds; int k = 0;
			for (int j = 0; j < n; ++j) {
				bwagpu_seed_t s = seeds[sbeg + j];
				if (s.score < 0 || s.score >= min_hsp) {
					if (s.score < 0) s.score = s.len * opt.a;
					seeds[m + k] = s; ++k;
				}
			}
			sbeg += n; m += k;
			chains[ci].n_seeds = k;
		}
	}
	wave_sync();
}

// CB != 0: per wave, dynamic LDS of SEEDSW_LDS_COLS x 64 cells of 2 CB bits + as many query bytes (launched with one wave per workgroup)
template <int CB> __global__ void __launch_bounds__(256) k_seedsw_wave(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	HIP_DYNAMIC_SHARED(unsigned char, ssw_lds)
	const int wave_in_blk = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + wave_in_blk;
	i32 *H = B.dp_h + wave * (B.max_len + 2) * DPS + lane;
	i32 *E = B.dp_e + wave * (B.max_len + 2) * DPS + lane;
	const size_t cell_bytes = CB ? (size_t)CB / 4 : 0, per_wave = (size_t)SEEDSW_LDS_COLS * 64 * (cell_bytes + 1);
	unsigned char *base = ssw_lds + (size_t)wave_in_blk * per_wave;
	void *HE = CB ? (void*)(base + (size_t)lane * cell_bytes) : nullptr;
	u8 *Q = CB ? base + (size_t)SEEDSW_LDS_COLS * 64 * cell_bytes + lane : nullptr;
	u64 rows[5] = { 0, 0, 0, 0, 0 };
	if (CB) mat_rows(opt, rows);
	u64 calls = 0, cells = 0;
	for (;;) {
		const long long k = wave_fetch(&B.ctr->next_seedsw);
		if (k >= B.n_reads) break;
		seedsw_read_wave<CB>(ix, opt, B, (int)k, H, E, HE, Q, rows, calls, cells);
	}
	if (B.stats) { atomicAdd(&B.ctr->sw_calls, (unsigned long long)calls); atomicAdd(&B.ctr->sw_cells, (unsigned long long)cells); }
}
