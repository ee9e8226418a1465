// host_pair.cpp -- paired-end finalize: insert-size statistics, mate rescue, pairing, PE SAM records.
// Re-written from the behaviour of bwamem_pair.c (mem_pestat :72-135, mem_matesw :137-206, mem_pair :208-274,
// mem_sam_pe :276-419).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include "bwamem_host.h"
#include "host_sort.h"

namespace hostmem {

enum { XBYTE = 0x10000, XSUBO = 0x40000, XSTART = 0x80000 };

static inline int infer_dir(int64_t l_pac, int64_t b1, int64_t b2, int64_t *dist)
{	// mem_infer_dir (bwamem_pair.c:49-56)
	int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	int64_t p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

struct RegSpan {   // a read's regions inside a flat array
	const bwagpu_alnreg_t *a; size_t n;
	size_t size() const { return n; } bool empty() const { return n == 0; }
	const bwagpu_alnreg_t &operator[](size_t i) const { return a[i]; }
};

template <class R> static int cal_sub(const bwagpu_opt_t &opt, const R &r)
{	// bwamem_pair.c:58-70
	size_t j;
	for (j = 1; j < r.size(); ++j) {
		int b_max = r[j].qb > r[0].qb ? r[j].qb : r[0].qb, e_min = r[j].qe < r[0].qe ? r[j].qe : r[0].qe;
		if (e_min > b_max) {
			int min_l = r[j].qe - r[j].qb < r[0].qe - r[0].qb ? r[j].qe - r[j].qb : r[0].qe - r[0].qb;
			if (e_min - b_max >= min_l * opt.mask_level) break;
		}
	}
	return j < r.size() ? r[j].score : opt.min_seed_len * opt.a;
}

struct U64Less { bool operator()(uint64_t a, uint64_t b) const { return a < b; } };

// The insert sizes of a batch's uniquely placed pairs, per orientation (the filter of bwamem_pair.c:79-91).  The pairs are independent,
// so the batch is cut into one contiguous range per thread and the ranges' lists are concatenated; everything downstream sorts the
// lists first, so their order is immaterial.
template <class Get> static void collect_isizes(const bwagpu_opt_t &opt, int64_t l_pac, int n_pairs, Get get, int n_threads, std::vector<uint64_t> isize[4])
{
	const int T = n_threads < 1 ? 1 : (n_pairs < 20000 ? 1 : (n_threads > 16 ? 16 : n_threads));
	std::vector<std::vector<uint64_t>> part((size_t)T * 4);
	parallel_for(T, T, [&](long t) {
		const int lo = (int)((int64_t)n_pairs * t / T), hi = (int)((int64_t)n_pairs * (t + 1) / T);
		for (int i = lo; i < hi; ++i) {
			const auto r0 = get(i << 1), r1 = get(i << 1 | 1);
			int64_t is;
			if (r0.empty() || r1.empty() || r0[0].rid != r1[0].rid) continue;
			if (cal_sub(opt, r0) > 0.8 * r0[0].score || cal_sub(opt, r1) > 0.8 * r1[0].score) continue;
			const int dir = infer_dir(l_pac, r0[0].rb, r1[0].rb, &is);
			if (is && is <= opt.max_ins) part[(size_t)t * 4 + dir].push_back((uint64_t)is);
		}
	});
	for (int d = 0; d < 4; ++d) { isize[d].clear(); for (int t = 0; t < T; ++t) isize[d].insert(isize[d].end(), part[(size_t)t * 4 + d].begin(), part[(size_t)t * 4 + d].end()); }
}

template <class Get> static void pestat_impl(const bwagpu_opt_t &opt, int64_t l_pac, int n, Get get, Pestat pes[4], bool verbose, int n_threads)
{
	std::vector<uint64_t> isize[4];
	memset(pes, 0, 4 * sizeof(Pestat));
	collect_isizes(opt, l_pac, n >> 1, get, n_threads, isize);
	if (verbose) fprintf(stderr, "[M::%s] # candidate unique pairs for (FF, FR, RF, RR): (%ld, %ld, %ld, %ld)\n", "mem_pestat", (long)isize[0].size(), (long)isize[1].size(), (long)isize[2].size(), (long)isize[3].size());
	for (int d = 0; d < 4; ++d) {
		Pestat *r = &pes[d];
		std::vector<uint64_t> &q = isize[d];
		if (q.size() < 10) {
			if (verbose) fprintf(stderr, "[M::%s] skip orientation %c%c as there are not enough pairs\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
			r->failed = 1;
			continue;
		} else if (verbose) fprintf(stderr, "[M::%s] analyzing insert size distribution for orientation %c%c...\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
		// (only the sorted values matter below, and every one of them is at most max_ins: a batch's ~300 k insert sizes are sorted by counting --
		// the comparison sort was most of this function's 26 ms per batch)
		if (q.size() > 4096 && opt.max_ins > 0 && opt.max_ins <= (1 << 22)) {
			std::vector<uint32_t> cnt((size_t)opt.max_ins + 1, 0);
			for (uint64_t v : q) ++cnt[(size_t)v];
			size_t w = 0;
			for (size_t v = 0; v < cnt.size(); ++v) for (uint32_t c = cnt[v]; c > 0; --c) q[w++] = (uint64_t)v;
		} else introsort(q.data(), (long)q.size(), U64Less());
		int p25 = (int)q[(int)(.25 * q.size() + .499)], p50 = (int)q[(int)(.50 * q.size() + .499)], p75 = (int)q[(int)(.75 * q.size() + .499)];
		r->low = (int)(p25 - 2.0 * (p75 - p25) + .499);
		if (r->low < 1) r->low = 1;
		r->high = (int)(p75 + 2.0 * (p75 - p25) + .499);
		if (verbose) {
			fprintf(stderr, "[M::%s] (25, 50, 75) percentile: (%d, %d, %d)\n", "mem_pestat", p25, p50, p75);
			fprintf(stderr, "[M::%s] low and high boundaries for computing mean and std.dev: (%d, %d)\n", "mem_pestat", r->low, r->high);
		}
		int x = 0; r->avg = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) { r->avg += v; ++x; }
		r->avg /= x;
		r->std = 0;
		for (uint64_t v : q) if (v >= (uint64_t)r->low && v <= (uint64_t)r->high) r->std += (v - r->avg) * (v - r->avg);
		r->std = sqrt(r->std / x);
		if (verbose) fprintf(stderr, "[M::%s] mean and std.dev: (%.2f, %.2f)\n", "mem_pestat", r->avg, r->std);
		r->low = (int)(p25 - 3.0 * (p75 - p25) + .499);
		r->high = (int)(p75 + 3.0 * (p75 - p25) + .499);
		if (r->low > r->avg - 4.0 * r->std) r->low = (int)(r->avg - 4.0 * r->std + .499);
		if (r->high < r->avg + 4.0 * r->std) r->high = (int)(r->avg + 4.0 * r->std + .499);
		if (r->low < 1) r->low = 1;
		if (verbose) fprintf(stderr, "[M::%s] low and high boundaries for proper pairs: (%d, %d)\n", "mem_pestat", r->low, r->high);
	}
	size_t max = 0;
	for (int d = 0; d < 4; ++d) max = max > isize[d].size() ? max : isize[d].size();
	for (int d = 0; d < 4; ++d)
		if (pes[d].failed == 0 && isize[d].size() < max * 0.05) {
			pes[d].failed = 1;
			if (verbose) fprintf(stderr, "[M::%s] skip orientation %c%c\n", "mem_pestat", "FR"[d >> 1 & 1], "FR"[d & 1]);
		}
}

// mem_pestat on a batch's flat region array (regions of read i at all[roff[i] .. roff[i+1]))
void pestat_flat(const bwagpu_opt_t &opt, int64_t l_pac, int n, const bwagpu_alnreg_t *all, const int64_t *roff, Pestat pes[4], bool verbose, int n_threads)
{
	pestat_impl(opt, l_pac, n, [&](int i) { return RegSpan{all + roff[i], (size_t)(roff[i + 1] - roff[i])}; }, pes, verbose, n_threads);
}

// ---- mem_sort_dedup_patch with bns == 0: no patching, only redundancy removal and the final sort (bwamem.c:463-515) ----
struct RegEndLess { bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const { return a.re < b.re; } };
struct RegBestLess {
	bool operator()(const bwagpu_alnreg_t &a, const bwagpu_alnreg_t &b) const {
		return a.score > b.score || (a.score == b.score && (a.rb < b.rb || (a.rb == b.rb && a.qb < b.qb)));
	}
};

int sort_dedup_nopatch(const bwagpu_opt_t &opt, Regs &av)
{
	int n = (int)av.size(), m;
	if (n <= 1) return n;
	bwagpu_alnreg_t *a = av.data();
	introsort(a, n, RegEndLess());
	for (int i = 0; i < n; ++i) a[i].n_comp = 1;
	for (int i = 1; i < n; ++i) {
		bwagpu_alnreg_t *p = &a[i];
		if (p->rid != a[i - 1].rid || p->rb >= a[i - 1].re + opt.max_chain_gap) continue;
		for (int j = i - 1; j >= 0 && p->rid == a[j].rid && p->rb < a[j].re + opt.max_chain_gap; --j) {
			bwagpu_alnreg_t *q = &a[j];
			if (q->qe == q->qb) continue;
			int64_t orr = q->re - p->rb, oq = q->qb < p->qb ? q->qe - p->qb : p->qe - q->qb;
			int64_t mr = q->re - q->rb < p->re - p->rb ? q->re - q->rb : p->re - p->rb;
			int64_t mq = q->qe - q->qb < p->qe - p->qb ? q->qe - q->qb : p->qe - p->qb;
			if (orr > opt.mask_level_redun * mr && oq > opt.mask_level_redun * mq) {
				if (p->score < q->score) { p->qe = p->qb; break; }
				else q->qe = q->qb;
			}   // mem_patch_reg() returns 0 without a reference (bwamem.c:436)
		}
	}
	m = 0;
	for (int i = 0; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	n = m;
	introsort(a, n, RegBestLess());
	for (int i = 1; i < n; ++i) if (a[i].score == a[i - 1].score && a[i].rb == a[i - 1].rb && a[i].qb == a[i - 1].qb) a[i].qe = a[i].qb;
	m = 1;
	for (int i = 1; i < n; ++i) if (a[i].qe > a[i].qb) { if (m != i) a[m] = a[i]; ++m; }
	av.resize(m);
	return m;
}

// ---- mate rescue (mem_matesw, bwamem_pair.c:137-206) ----------------------------------------------------------------------
static int matesw(const bwagpu_opt_t &opt, const RefSeqs &ref, const Pestat pes[4], const bwagpu_alnreg_t &a, int l_ms, const uint8_t *ms, Regs &ma,
				  const bwagpu_matesw_t *msw, int n_msw)   // msw: device-computed alignments of this mate (bwagpu_batch_matesw), or none
{
	const int64_t l_pac = ref.l_pac;
	int skip[4], n = 0;
	for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
	for (size_t i = 0; i < ma.size(); ++i) {
		int64_t dist;
		int r = infer_dir(l_pac, a.rb, ma[i].rb, &dist);
		if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
	}
	if (skip[0] + skip[1] + skip[2] + skip[3] == 4) return 0;
	for (int r = 0; r < 4; ++r) {
		if (skip[r]) continue;
		const int is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
		std::vector<uint8_t> rev, refseq;
		const uint8_t *seq = ms;
		if (is_rev) {
			rev.resize(l_ms);
			for (int i = 0; i < l_ms; ++i) rev[l_ms - 1 - i] = ms[i] < 4 ? 3 - ms[i] : 4;
			seq = rev.data();
		}
		int64_t rb, re; int rid = -1; bool fetched = false;
		if (!is_rev) {
			rb = is_larger ? a.rb + pes[r].low : a.rb - pes[r].high;
			re = (is_larger ? a.rb + pes[r].high : a.rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? a.rb + pes[r].low : a.rb - pes[r].high) - l_ms;
			re = is_larger ? a.rb + pes[r].high : a.rb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		if (rb < re) { ref.fetch_seq(rb, (rb + re) >> 1, re, rid, refseq); fetched = true; }
		if (fetched && a.rid == rid && re - rb >= opt.min_seed_len) {
			int xtra = XSUBO | XSTART | (l_ms * opt.a < 250 ? XBYTE : 0) | (opt.min_seed_len * opt.a);
			KswResult aln;
			const bwagpu_matesw_t *hm = nullptr;
			for (int k = 0; k < n_msw; ++k) if (msw[k].r == r && msw[k].anchor_rb == a.rb && msw[k].anchor_rid == a.rid) { hm = &msw[k]; break; }
			if (hm) { aln.score = hm->score; aln.te = hm->te; aln.qe = hm->qe; aln.score2 = hm->score2; aln.te2 = hm->te2; aln.tb = hm->tb; aln.qb = hm->qb; }
			else aln = ksw_align2(l_ms, seq, (int)(re - rb), refseq.data(), opt.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra);
			if (aln.score >= opt.min_seed_len && aln.qb >= 0) {
				bwagpu_alnreg_t b; memset(&b, 0, sizeof b);
				b.rid = a.rid; b.is_alt = a.is_alt;
				b.qb = is_rev ? l_ms - (aln.qe + 1) : aln.qb;
				b.qe = is_rev ? l_ms - aln.qb : aln.qe + 1;
				b.rb = is_rev ? (l_pac << 1) - (rb + aln.te + 1) : rb + aln.tb;
				b.re = is_rev ? (l_pac << 1) - (rb + aln.tb) : rb + aln.te + 1;
				b.score = aln.score; b.csub = aln.score2; b.secondary = -1;
				b.seedcov = (int)((b.re - b.rb < b.qe - b.qb ? b.re - b.rb : b.qe - b.qb) >> 1);
				ma.push_back(b);
				size_t i, tmp;
				for (i = 0; i < ma.size() - 1; ++i) if (ma[i].score < b.score) break;   // keep ma sorted by score
				tmp = i;
				for (i = ma.size() - 1; i > tmp; --i) ma[i] = ma[i - 1];
				ma[i] = b;
			}
			++n;
		}
		if (n) sort_dedup_nopatch(opt, ma);
	}
	return n;
}

// What bwagpu_batch_matesw computes, on the host (tests): tasks enumerated from the initial region lists in read order, results by
// ksw_align2.  Returns the number of records written (at most cap).
int64_t host_matesw_records(const bwagpu_opt_t &opt, const RefSeqs &ref, int n, const uint8_t *seqs, const int64_t *off, const bwagpu_alnreg_t *all, const int64_t *roff,
							const Pestat pes[4], bwagpu_matesw_t *out, int64_t cap)
{
	const int64_t l_pac = ref.l_pac;
	int64_t nout = 0;
	for (int p = 0; p < n / 2; ++p)
		for (int i = 0; i < 2; ++i) {
			const int ri = 2 * p + i, rm = 2 * p + (1 - i);
			const int ni = (int)(roff[ri + 1] - roff[ri]), nm = (int)(roff[rm + 1] - roff[rm]);
			if (ni == 0) continue;
			const bwagpu_alnreg_t *a = all + roff[ri], *ma = all + roff[rm];
			int taken = 0;
			for (int j = 0; j < ni && taken < opt.max_matesw; ++j) {
				if (a[j].score < a[0].score - opt.pen_unpaired) continue;
				++taken;
				int skip[4];
				for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
				for (int k = 0; k < nm; ++k) { int64_t dist; int r = infer_dir(l_pac, a[j].rb, ma[k].rb, &dist); if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1; }
				for (int r = 0; r < 4; ++r) {
					if (skip[r] || nout >= cap) continue;
					bwagpu_matesw_t o; memset(&o, 0, sizeof o);
					o.read = rm; o.r = -1; o.anchor_rb = a[j].rb; o.anchor_rid = a[j].rid; o.te = o.qe = o.score2 = o.te2 = o.tb = o.qb = -1;
					const int l_ms = (int)(off[rm + 1] - off[rm]); const uint8_t *ms = seqs + off[rm];
					const int is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
					std::vector<uint8_t> rev, refseq; const uint8_t *seq = ms;
					if (is_rev) { rev.resize(l_ms); for (int q = 0; q < l_ms; ++q) rev[l_ms - 1 - q] = ms[q] < 4 ? 3 - ms[q] : 4; seq = rev.data(); }
					int64_t rb, re; int rid = -1;
					if (!is_rev) { rb = is_larger ? a[j].rb + pes[r].low : a[j].rb - pes[r].high; re = (is_larger ? a[j].rb + pes[r].high : a[j].rb - pes[r].low) + l_ms; }
					else { rb = (is_larger ? a[j].rb + pes[r].low : a[j].rb - pes[r].high) - l_ms; re = is_larger ? a[j].rb + pes[r].high : a[j].rb - pes[r].low; }
					if (rb < 0) rb = 0;
					if (re > l_pac << 1) re = l_pac << 1;
					if (rb < re && l_ms <= 512) {
						ref.fetch_seq(rb, (rb + re) >> 1, re, rid, refseq);
						if (a[j].rid == rid && re - rb >= opt.min_seed_len && re - rb <= 2048) {
							int xtra = XSUBO | XSTART | (l_ms * opt.a < 250 ? XBYTE : 0) | (opt.min_seed_len * opt.a);
							KswResult aln = ksw_align2(l_ms, seq, (int)(re - rb), refseq.data(), opt.mat, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra);
							o.r = r; o.score = aln.score; o.te = aln.te; o.qe = aln.qe; o.score2 = aln.score2; o.te2 = aln.te2; o.tb = aln.tb; o.qb = aln.qb;
						}
					}
					out[nout++] = o;
				}
			}
		}
	return nout;
}

// ---- pairing (mem_pair, bwamem_pair.c:208-274) --------------------------------------------------------------------------------
struct Pair64 { uint64_t x, y; };
struct Pair64Less { bool operator()(const Pair64 &a, const Pair64 &b) const { return a.x < b.x || (a.x == b.x && a.y < b.y); } };   // utils.c:44

static int pair_ends(const bwagpu_opt_t &opt, const RefSeqs &ref, const Pestat pes[4], const Regs a[2], int id, int *sub, int *n_sub, int z[2], const int n_pri[2])
{
	thread_local std::vector<Pair64> v, u;      // (scratch kept from pair to pair)
	v.clear(); u.clear();
	const int64_t l_pac = ref.l_pac;
	int y[4], ret;
	for (int r = 0; r < 2; ++r)
		for (int i = 0; i < n_pri[r]; ++i) {
			const bwagpu_alnreg_t &e = a[r][i];
			Pair64 key;
			key.x = e.rb < l_pac ? e.rb : (l_pac << 1) - 1 - e.rb;
			key.x = (uint64_t)e.rid << 32 | (key.x - ref.ctg[e.rid].offset);
			key.y = (uint64_t)e.score << 32 | (uint64_t)(int64_t)(i << 2) | (uint64_t)((e.rb >= l_pac) << 1) | (uint64_t)r;
			v.push_back(key);
		}
	introsort(v.data(), (long)v.size(), Pair64Less());
	y[0] = y[1] = y[2] = y[3] = -1;
	for (int i = 0; i < (int)v.size(); ++i) {
		for (int r = 0; r < 2; ++r) {
			int dir = r << 1 | (int)(v[i].y >> 1 & 1), which;
			if (pes[dir].failed) continue;
			which = r << 1 | (int)((v[i].y & 1) ^ 1);
			if (y[which] < 0) continue;
			for (int k = y[which]; k >= 0; --k) {
				if ((int)(v[k].y & 3) != which) continue;
				int64_t dist = (int64_t)v[i].x - (int64_t)v[k].x;
				if (dist > pes[dir].high) break;
				if (dist < pes[dir].low) continue;
				double ns = (dist - pes[dir].avg) / pes[dir].std;
				int q = (int)((v[i].y >> 32) + (v[k].y >> 32) + .721 * log(2. * erfc(fabs(ns) * M_SQRT1_2)) * opt.a + .499);
				if (q < 0) q = 0;
				Pair64 p;
				p.y = (uint64_t)k << 32 | (uint64_t)i;
				// the reference hashes p->y ^ id<<8 with `int id`: the shift is done in 32 bits and sign-extended (bwamem_pair.c:248)
				p.x = (uint64_t)q << 32 | (hash_64(p.y ^ (uint64_t)(int64_t)(int32_t)((uint32_t)id << 8)) & 0xffffffffU);
				u.push_back(p);
			}
		}
		y[v[i].y & 3] = i;
	}
	if (!u.empty()) {
		int tmp = opt.a + opt.b;
		tmp = tmp > opt.o_del + opt.e_del ? tmp : opt.o_del + opt.e_del;
		tmp = tmp > opt.o_ins + opt.e_ins ? tmp : opt.o_ins + opt.e_ins;
		introsort(u.data(), (long)u.size(), Pair64Less());
		int i = (int)(u.back().y >> 32), k = (int)(u.back().y << 32 >> 32);
		z[v[i].y & 1] = (int)(v[i].y << 32 >> 34);
		z[v[k].y & 1] = (int)(v[k].y << 32 >> 34);
		ret = (int)(u.back().x >> 32);
		*sub = u.size() > 1 ? (int)(u[u.size() - 2].x >> 32) : 0;
		*n_sub = 0;
		for (long j = (long)u.size() - 2; j >= 0; --j) if (*sub - (int)(u[j].x >> 32) <= tmp) ++*n_sub;
	} else { ret = 0; *sub = 0; *n_sub = 0; }
	return ret;
}

// from host_finalize.cpp
bool gen_alt_for_pe(const bwagpu_opt_t &opt, const RefSeqs &ref, const Regs &av, int l_query, const uint8_t *query, std::vector<std::string> &xa, std::vector<char> &has, const CigHints *hints);

static inline int raw_mapq(int diff, int a) { return (int)(6.02 * diff / a + .499); }

// ---- mem_sam_pe (bwamem_pair.c:276-419) ------------------------------------------------------------------------------------
int sam_pe(const bwagpu_opt_t &opt, const RefSeqs &ref, const Pestat pes[4], uint64_t id, const Read s[2], Regs a[2], SamText *const outp[2], const char *rg_id)
{
	int n = 0, z[2] = {0, 0}, o, subo, n_sub, extra_flag = 1, n_pri[2];
	Aln h[2];
	SamText &out0 = *outp[0], &out1 = *outp[1];
	out0.reserve(out0.size() + 2 * (size_t)s[0].l_seq + 320); out1.reserve(out1.size() + 2 * (size_t)s[1].l_seq + 320);      // one allocation instead of the five a growing string makes
	if (!(opt.flag & F_NO_RESCUE)) {   // mate rescue from the best hits of each end
		thread_local Regs b[2];
		b[0].clear(); b[1].clear();
		for (int i = 0; i < 2; ++i)
			for (size_t j = 0; j < a[i].size(); ++j)
				if (a[i][j].score >= a[i][0].score - opt.pen_unpaired) b[i].push_back(a[i][j]);
		for (int i = 0; i < 2; ++i)
			for (int j = 0; j < (int)b[i].size() && j < opt.max_matesw; ++j)
				n += matesw(opt, ref, pes, b[i][j], s[!i].l_seq, s[!i].seq, a[!i], s[!i].msw, s[!i].n_msw);
	}
	n_pri[0] = mark_primary_se(opt, a[0], (int64_t)(id << 1 | 0));
	n_pri[1] = mark_primary_se(opt, a[1], (int64_t)(id << 1 | 1));
	if (opt.flag & F_PRIMARY5) { reorder_primary5(opt.T, a[0]); reorder_primary5(opt.T, a[1]); }
	bool no_pairing = (opt.flag & F_NOPAIRING) != 0;
	if (!no_pairing) {
		if (n_pri[0] && n_pri[1] && (o = pair_ends(opt, ref, pes, a, (int)id, &subo, &n_sub, z, n_pri)) > 0) {
			int is_multi[2], q_pe, score_un, q_se[2];
			for (int i = 0; i < 2; ++i) {
				int j;
				for (j = 1; j < n_pri[i]; ++j) if (a[i][j].secondary < 0 && a[i][j].score >= opt.T) break;
				is_multi[i] = j < n_pri[i] ? 1 : 0;
			}
			if (is_multi[0] || is_multi[1]) no_pairing = true;
			else {
				score_un = a[0][0].score + a[1][0].score - opt.pen_unpaired;
				subo = subo > score_un ? subo : score_un;
				q_pe = raw_mapq(o - subo, opt.a);
				if (n_sub > 0) q_pe -= (int)(4.343 * log(n_sub + 1) + .499);
				if (q_pe < 0) q_pe = 0;
				if (q_pe > 60) q_pe = 60;
				q_pe = (int)(q_pe * (1. - .5 * (a[0][0].frac_rep + a[1][0].frac_rep)) + .499);
				if (o > score_un) {
					bwagpu_alnreg_t *c[2] = { &a[0][z[0]], &a[1][z[1]] };
					for (int i = 0; i < 2; ++i) {
						if (c[i]->secondary >= 0) { c[i]->sub = a[i][c[i]->secondary].score; c[i]->secondary = -2; }
						q_se[i] = approx_mapq_se(opt, *c[i]);
					}
					q_se[0] = q_se[0] > q_pe ? q_se[0] : q_pe < q_se[0] + 40 ? q_pe : q_se[0] + 40;
					q_se[1] = q_se[1] > q_pe ? q_se[1] : q_pe < q_se[1] + 40 ? q_pe : q_se[1] + 40;
					extra_flag |= 2;
					q_se[0] = q_se[0] < raw_mapq(c[0]->score - c[0]->csub, opt.a) ? q_se[0] : raw_mapq(c[0]->score - c[0]->csub, opt.a);
					q_se[1] = q_se[1] < raw_mapq(c[1]->score - c[1]->csub, opt.a) ? q_se[1] : raw_mapq(c[1]->score - c[1]->csub, opt.a);
				} else {
					z[0] = z[1] = 0;
					q_se[0] = approx_mapq_se(opt, a[0][0]);
					q_se[1] = approx_mapq_se(opt, a[1][0]);
				}
				for (int i = 0; i < 2; ++i) {
					int k = a[i][z[i]].secondary_all;
					if (k >= 0 && k < n_pri[i]) {   // swap primary and secondary roles if both are non-ALT
						for (size_t j = 0; j < a[i].size(); ++j) if (a[i][j].secondary_all == k || (int)j == k) a[i][j].secondary_all = z[i];
						a[i][z[i]].secondary_all = -1;
					}
				}
				std::vector<std::string> xa[2]; std::vector<char> has[2]; bool have[2] = {false, false};
				if (!(opt.flag & F_ALL)) for (int i = 0; i < 2; ++i) have[i] = gen_alt_for_pe(opt, ref, a[i], s[i].l_seq, s[i].seq, xa[i], has[i], s[i].hints);
				std::vector<Aln> aa[2];
				for (int i = 0; i < 2; ++i) {
					h[i] = reg2aln(opt, ref, s[i].l_seq, s[i].seq, &a[i][z[i]], s[i].hints);
					h[i].mapq = q_se[i];
					h[i].flag |= 0x40 << i | extra_flag;
					if (have[i] && has[i][z[i]]) { h[i].has_xa = true; h[i].xa = xa[i][z[i]]; }
					aa[i].push_back(h[i]);
					if (n_pri[i] < (int)a[i].size()) {   // the read has ALT hits
						const bwagpu_alnreg_t &p = a[i][n_pri[i]];
						if (p.score < opt.T || p.secondary >= 0 || !p.is_alt) continue;
						Aln g = reg2aln(opt, ref, s[i].l_seq, s[i].seq, &p, s[i].hints);
						g.flag |= 0x800 | 0x40 << i | extra_flag;
						if (have[i] && has[i][n_pri[i]]) { g.has_xa = true; g.xa = xa[i][n_pri[i]]; }
						aa[i].push_back(g);
					}
				}
				for (int i = 0; i < (int)aa[0].size(); ++i) aln2sam(opt, ref, out0, s[0], aa[0], i, &h[1], rg_id);
				for (int i = 0; i < (int)aa[1].size(); ++i) aln2sam(opt, ref, out1, s[1], aa[1], i, &h[0], rg_id);
				return n;
			}
		} else no_pairing = true;
	}
	// no_pairing (bwamem_pair.c:397-418)
	for (int i = 0; i < 2; ++i) {
		int which = -1;
		if (!a[i].empty()) {
			if (a[i][0].score >= opt.T) which = 0;
			else if (n_pri[i] < (int)a[i].size() && a[i][n_pri[i]].score >= opt.T) which = n_pri[i];
		}
		h[i] = which >= 0 ? reg2aln(opt, ref, s[i].l_seq, s[i].seq, &a[i][which], s[i].hints) : reg2aln(opt, ref, s[i].l_seq, s[i].seq, 0);
	}
	if (!(opt.flag & F_NOPAIRING) && h[0].rid == h[1].rid && h[0].rid >= 0) {
		int64_t dist;
		int d = infer_dir(ref.l_pac, a[0][0].rb, a[1][0].rb, &dist);
		if (!pes[d].failed && dist >= pes[d].low && dist <= pes[d].high) extra_flag |= 2;
	}
	reg2sam(opt, ref, out0, s[0], a[0], 0x41 | extra_flag, &h[1], rg_id);
	reg2sam(opt, ref, out1, s[1], a[1], 0x81 | extra_flag, &h[0], rg_id);
	return n;
}

}  // namespace hostmem
