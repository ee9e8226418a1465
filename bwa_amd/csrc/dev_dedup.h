// dev_dedup.h -- mem_sort_dedup_patch (bwamem.c:463-515) with mem_patch_reg (bwamem.c:432-461), whose global
// alignment is bwa_gen_cigar2 in score-only mode (bwa.c:148-194) -> ksw_global2 without traceback (ksw.c:604-619).
#pragma once
#include "dev_extw.h"

#define DEV_NEG_INF (-0x40000000)

// Score of the banded global alignment of query[0..qlen) (q[q0 + j*qdir]) against target ref_base(t0 + i*tdir).
__device__ int dev_ksw_global2_score(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int q0, int qdir, int qlen,
									 i64 t0, int tdir, int tlen, int w, i32 *H, i32 *E, u64 &cells)
{
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins;
	int j;
	H[0] = 0; E[0] = DEV_NEG_INF;
	for (j = 1; j <= qlen && j <= w; ++j) { H[j * DPS] = -(o_ins + e_ins * j); E[j * DPS] = DEV_NEG_INF; }
	for (; j <= qlen; ++j) H[j * DPS] = E[j * DPS] = DEV_NEG_INF;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = opt.mat + ref_base(ix, t0 + (i64)i * tdir) * 5;
		int beg = i > w ? i - w : 0, end = i + w + 1 < qlen ? i + w + 1 : qlen;
		int f = DEV_NEG_INF, h1 = beg == 0 ? -(o_del + e_del * (i + 1)) : DEV_NEG_INF;
		cells += (u64)(end > beg ? end - beg : 0);
		for (j = beg; j < end; ++j) {
			int m = H[j * DPS] + srow[q[q0 + j * qdir]], e = E[j * DPS], h, t;
			H[j * DPS] = h1;
			h = m >= e ? m : e; if (h < f) h = f;
			h1 = h;
			t = m - oe_del; e -= e_del; E[j * DPS] = e > t ? e : t;
			t = m - oe_ins; f -= e_ins; if (t > f) f = t;
		}
		H[end * DPS] = h1; E[end * DPS] = DEV_NEG_INF;
	}
	return H[qlen * DPS];
}

// bwa_gen_cigar2 (bwa.c:148-194), score only: both sequences are reversed for reverse-strand hits so that gaps
// end up left-aligned on the forward strand.
__device__ int dev_global_score(const DevIndex &ix, const bwagpu_opt_t &opt, int w_, int l_query, const u8 *query, i64 rb, i64 re,
								i32 *H, i32 *E, u64 &calls, u64 &cells)
{
	i64 l_pac = ix.l_pac;
	if (l_query <= 0 || rb >= re || (rb < l_pac && re > l_pac)) return 0;
	int rlen = (int)(re - rb), rev = rb >= l_pac;
	int q0 = rev ? l_query - 1 : 0, qdir = rev ? -1 : 1; i64 t0 = rev ? re - 1 : rb; int tdir = rev ? -1 : 1;
	if (l_query == rlen && w_ == 0) {
		int score = 0;
		for (int i = 0; i < l_query; ++i) score += opt.mat[ref_base(ix, t0 + (i64)i * tdir) * 5 + query[q0 + i * qdir]];
		return score;
	}
	int max_ins = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_ins, opt.e_ins, 1);
	int max_del = trunc_div_add(((l_query + 1) >> 1) * opt.mat[0] - opt.o_del, opt.e_del, 1);
	int mg = max_ins > max_del ? max_ins : max_del, dl = rlen - l_query;
	if (dl < 0) dl = -dl;
	if (mg < 1) mg = 1;
	int w = (mg + dl + 1) >> 1; if (w > w_) w = w_;
	if (w < dl + 3) w = dl + 3;
	++calls;
	return dev_ksw_global2_score(ix, opt, query, q0, qdir, l_query, t0, tdir, rlen, w, H, E, cells);
}

// mem_patch_reg (bwamem.c:432-461)
__device__ int dev_patch_reg(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *query, const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b,
							 int *w_out, i32 *H, i32 *E, u64 &calls, u64 &cells)
{
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return 0;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return 0;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb)); if (w < 0) w = -w;
	double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb); if (r < 0.) r = -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > opt.w << 1 || r >= 0.05f) return 0; }
	else if (w > opt.w << 2 || r >= 0.05f * 2) return 0;
	w += a.w + b.w;
	if (w > opt.w << 2) w = opt.w << 2;
	int score = dev_global_score(ix, opt, w, b.qe - a.qb, query + a.qb, a.rb, b.re, H, E, calls, cells);
	int q_s = (int)((double)(b.qe - a.qb) / ((b.qe - b.qb) + (a.qe - a.qb)) * (b.score + a.score) + .499);
	int r_s = (int)((double)(b.re - a.rb) / ((b.re - b.rb) + (a.re - a.rb)) * (b.score + a.score) + .499);
	if ((double)score / (q_s > r_s ? q_s : r_s) < 0.90f) return 0;
	*w_out = w;
	return score;
}

struct RegEndLess { DEVFN bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const { return a.re < b.re; } };
struct RegBestLess {
	DEVFN bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const {
		return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb)));
	}
};

// The two sorts of mem_sort_dedup_patch for a read with many regions.  ks_introsort is not stable, so the order of equal keys must be the
// introsort's own -- but that order is a function of the comparisons alone.  Sorting small key records {key fields, index} with the same
// routine and the same comparator, then moving every 88-byte region once along the permutation's cycles, leaves the array exactly as sorting
// the regions themselves does, for a fraction of the memory traffic (one lane's introsort of 900 regions was 10 of this kernel's 13 ms).
// The keys live in the read's chaining scratch (RegionView::chain, 64 bytes per seed slot, free since the chaining stage; a read has no more
// regions than seed slots).
struct RegKey { i64 a; i32 b, c; i32 idx; i32 pad_; };       // RegEndLess: a = re.  RegBestLess: b = score, a = rb, c = qb.
struct KeyEndLess { DEVFN bool operator()(const RegKey &x, const RegKey &y) const { return x.a < y.a; } };
struct KeyBestLess { DEVFN bool operator()(const RegKey &x, const RegKey &y) const { return x.b > y.b || (x.b == y.b && (x.a < y.a || (x.a == y.a && x.c < y.c))); } };
#define DEDUP_KEYSORT_MIN 24
template <class LT> DEVFN void dev_sort_regs_by_key(bwagpu_alnreg_t *a, int n, RegKey *k, bool by_end, LT lt)
{
	for (int i = 0; i < n; ++i) { k[i].a = by_end ? a[i].re : a[i].rb; k[i].b = a[i].score; k[i].c = a[i].qb; k[i].idx = i; k[i].pad_ = 0; }
	dev_introsort(k, n, lt);
	for (int i = 0; i < n; ++i) {          // position i takes the region that was at k[i].idx: follow each cycle once (idx < 0: already in place)
		if (k[i].idx < 0 || k[i].idx == i) { k[i].idx = -1; continue; }
		const bwagpu_alnreg_t first = a[i];
		int j = i;
		for (;;) {
			const int src = k[j].idx;
			k[j].idx = -1;
			if (src == i) { a[j] = first; break; }
			a[j] = a[src];
			j = src;
		}
	}
}

__device__ void dedup_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, i32 *H, i32 *E, u64 &calls, u64 &cells)
{
	int n = B.reg_n_raw[r];
	bwagpu_alnreg_t *a = B.regs + B.reg_off[r];
	const u8 *query = B.seq + B.off[r];
	if (B.regs_raw) for (int i = 0; i < n; ++i) B.regs_raw[B.reg_off[r] + i] = a[i];
	if (n > 1) {
		int m;
		RegKey *keys = n >= DEDUP_KEYSORT_MIN ? (RegKey*)region_of(B.slot_blob, B.seed_off[r], B.seed_n[r]).chain : nullptr;
		static_assert(sizeof(RegKey) <= sizeof(ChainRec), "a key record per seed slot must fit the chaining scratch");
		if (keys) dev_sort_regs_by_key(a, n, keys, true, KeyEndLess()); else dev_introsort(a, n, RegEndLess());
		for (int i = 0; i < n; ++i) a[i].n_comp = 1;
		for (int i = 1; i < n; ++i) {
			bwagpu_alnreg_t &p = a[i];
			if (p.rid != a[i - 1].rid || p.rb >= a[i - 1].re + opt.max_chain_gap) continue;
			for (int j = i - 1; j >= 0 && p.rid == a[j].rid && p.rb < a[j].re + opt.max_chain_gap; --j) {
				bwagpu_alnreg_t &q = a[j];
				i64 orr, oq, mr, mq; int score, w;
				if (q.qe == q.qb) continue;
				orr = q.re - p.rb;
				oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
				mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				if (orr > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
					if (p.score < q.score) { p.qe = p.qb; break; }
					else q.qe = q.qb;
				} else if (q.rb < p.rb && (score = dev_patch_reg(ix, opt, query, q, p, &w, H, E, calls, cells)) > 0) {
					p.n_comp += q.n_comp + 1;
					if (q.seedcov > p.seedcov) p.seedcov = q.seedcov;
					if (q.sub > p.sub) p.sub = q.sub;
					if (q.csub > p.csub) p.csub = q.csub;
					p.qb = q.qb; p.rb = q.rb;
					p.truesc = p.score = score;
					p.w = w;
					q.qb = q.qe;
				}
			}
		}
		m = 0;
		for (int i = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
		n = m;
		if (keys && n >= DEDUP_KEYSORT_MIN) dev_sort_regs_by_key(a, n, keys, false, KeyBestLess()); else dev_introsort(a, n, RegBestLess());
		for (int i = 1; i < n; ++i)
			if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
		m = n > 0 ? 1 : 0;
		for (int i = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
		n = m;
	}
	for (int i = 0; i < n; ++i)   // bwamem.c:1111-1115
		if (a[i].rid >= 0 && ix.ctg_alt[a[i].rid]) a[i].is_alt = 1;
	B.reg_n[r] = n;
}

__global__ void __launch_bounds__(256) k_dedup(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	int tid = blockIdx.x * blockDim.x + threadIdx.x;
	int wave = tid >> 6, lane = tid & 63;
	i32 *H = B.dp_h + (size_t)wave * (B.max_len + 2) * DPS + lane;
	i32 *E = B.dp_e + (size_t)wave * (B.max_len + 2) * DPS + lane;
	u64 calls = 0, cells = 0, nreg = 0;
	for (;;) {       // reads are drawn 64 at a time, one atomic per wave: a million same-address atomics are ~13 ms on this chip, the whole of this kernel's time
		const long long base = wave_fetch_n(&B.ctr->next_dedup, 64);
		if (base >= B.n_reads) break;
		const int r = (int)base + lane;
		// A read with many regions is a long serial job for one lane (sorts, the pair loop, the patch alignments) while the other 63 wait:
		// those go to a list that k_dedup_wave<.., LIST> works through with one wavefront per read.
		if (r < B.n_reads) {
			const int n_raw = B.dd_heavy_min > 0 ? B.reg_n_raw[r] : 0;
			if (n_raw > B.dd_stage_cap && n_raw >= B.dd_heavy_min) B.dd_list[B.n_reads - 1 - (long long)atomicAdd(&B.ctr->n_dd_big, 1ull)] = r;      // (more regions than the LDS copy holds: from the far end)
			else if (n_raw >= B.dd_heavy_min && B.dd_heavy_min > 0) B.dd_list[atomicAdd(&B.ctr->n_dd_heavy, 1ull)] = r;
			else { dedup_read(ix, opt, B, r, H, E, calls, cells); nreg += B.reg_n[r]; }
		}
	}
	if (B.stats) {
		atomicAdd(&B.ctr->glb_calls, (unsigned long long)calls);
		atomicAdd(&B.ctr->glb_cells, (unsigned long long)cells);
		atomicAdd(&B.ctr->n_regs, (unsigned long long)nreg);
	}
}
