// dev_ext.h -- seed extension: ksw_extend2 (ksw.c:416-515) driven by mem_chain2aln (bwamem.c:658-812).
//
// v1 layout: one lane per read.  The lane walks its chains in order (the containment test of bwamem.c:697-732
// makes seeds of a read order-dependent), and runs the banded DP row by row.  The two DP rows (H of the previous
// row shifted by one, and E) live in a per-wave scratch that is interleaved by lane ([column][lane]) so that the
// 64 lanes of a wave, which sweep their columns in near lock-step, touch the same cache lines.  The reference
// window is never materialised: target bases are decoded on the fly from the 2-bit pac (one byte load per row).
#pragma once
#include "dev_chain.h"

DEVFN int pac_base(const u8 *pac, i64 l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }   // _get_pac (bntseq.c:230)
DEVFN int ref_base(const DevIndex &ix, i64 p)
{	// base at forward+reverse-complement coordinate p (bns_get_seq, bntseq.c:403-424)
	return p < ix.l_pac ? pac_base(ix.pac, p) : 3 - pac_base(ix.pac, (ix.l_pac << 1) - 1 - p);
}

// (int)((double)x / e + k.) for integers x, k and e > 0, in integer arithmetic.  The reference computes several band limits this way
// (bwamem.c:649-650, ksw.c:436-443, bwa.c:180-181, bwamem.c:822).  The double quotient is the correctly rounded x/e, whose distance to the
// next integer is at least 1/e unless it is one, far more than the rounding errors of the division and of the addition for 32-bit
// operands, and the conversion truncates toward zero: exactly C's (x + k*e) / e.  A double division costs the device ~40 instructions, and
// with the usual gap extension penalty of 1 there is nothing to divide.
DEVFN int trunc_div_add(int x, int e, int k) { const int z = x + k * e; return e == 1 ? z : z / e; }

// cal_max_gap (bwamem.c:647-654)
DEVFN int dev_max_gap(const bwagpu_opt_t &opt, int qlen)
{
	int l_del = trunc_div_add(qlen * opt.a - opt.o_del, opt.e_del, 1);
	int l_ins = trunc_div_add(qlen * opt.a - opt.o_ins, opt.e_ins, 1);
	int l = l_del > l_ins ? l_del : l_ins;
	if (l < 1) l = 1;
	return l < opt.w << 1 ? l : opt.w << 1;
}

struct ExtRes { int score, qle, tle, gtle, gscore, max_off; };
#define DPS 64   // lane interleave of the DP scratch

// ksw_extend2.  query j -> q[q0 + j*qdir]; target i -> ref_base(t0 + i*tdir).  H[j*DPS] = H(i-1,j-1), E[j*DPS] = E(i,j).
// Gaps open from the diagonal term M, not from H (ksw.c:469-483); the arrays are never cleared between rows, which
// yields the reference's "stale cell" behaviour when the band regrows.
__device__ ExtRes dev_ksw_extend2(const DevIndex &ix, const bwagpu_opt_t &opt, int mat_max, const u8 *q, int q0, int qdir, int qlen,
								  i64 t0, int tdir, int tlen, int w, int end_bonus, int h0, i32 *H, i32 *E, u64 &cells)
{
	const int o_del = opt.o_del, e_del = opt.e_del, o_ins = opt.o_ins, e_ins = opt.e_ins;
	const int oe_del = o_del + e_del, oe_ins = o_ins + e_ins, zdrop = opt.zdrop;
	int beg = 0, end = qlen, max = h0, max_i = -1, max_j = -1, max_ie = -1, gscore = -1, max_off = 0, j;
	for (j = 0; j <= qlen; ++j) { H[j * DPS] = 0; E[j * DPS] = 0; }
	H[0] = h0;
	if (qlen >= 1) H[DPS] = h0 > oe_ins ? h0 - oe_ins : 0;
	for (j = 2; j <= qlen && H[(j - 1) * DPS] > e_ins; ++j) H[j * DPS] = H[(j - 1) * DPS] - e_ins;
	int lim = trunc_div_add(qlen * mat_max + end_bonus - o_ins, e_ins, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;
	lim = trunc_div_add(qlen * mat_max + end_bonus - o_del, e_del, 1); if (lim < 1) lim = 1; if (w > lim) w = lim;
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = opt.mat + ref_base(ix, t0 + (i64)i * tdir) * 5;
		int f = 0, h1, m = 0, mj = -1;
		if (beg < i - w) beg = i - w;
		if (end > i + w + 1) end = i + w + 1;
		if (end > qlen) end = qlen;
		if (beg == 0) { h1 = h0 - (o_del + e_del * (i + 1)); if (h1 < 0) h1 = 0; } else h1 = 0;
		cells += (u64)(end > beg ? end - beg : 0);
		for (j = beg; j < end; ++j) {
			int M = H[j * DPS], e = E[j * DPS], h, t;
			H[j * DPS] = h1;
			M = M ? M + srow[q[q0 + j * qdir]] : 0;
			h = M > e ? M : e; if (f > h) h = f;
			h1 = h;
			if (h >= m) { mj = j; m = h; }
			t = M - oe_del; if (t < 0) t = 0; e -= e_del; E[j * DPS] = e > t ? e : t;
			t = M - oe_ins; if (t < 0) t = 0; f -= e_ins; if (t > f) f = t;
		}
		H[end * DPS] = h1; E[end * DPS] = 0;
		if (j == qlen) { if (h1 >= gscore) max_ie = i; if (h1 > gscore) gscore = h1; }
		if (m == 0) break;
		if (m > max) {
			int off = mj - i; if (off < 0) off = -off;
			max = m; max_i = i; max_j = mj;
			if (off > max_off) max_off = off;
		} else if (zdrop > 0) {
			int di = i - max_i, dj = mj - max_j;
			if (di > dj) { if (max - m - (di - dj) * e_del > zdrop) break; }
			else if (max - m - (dj - di) * e_ins > zdrop) break;
		}
		for (j = beg; j < end && H[j * DPS] == 0 && E[j * DPS] == 0; ++j) {}
		beg = j;
		for (j = end; j >= beg && H[j * DPS] == 0 && E[j * DPS] == 0; --j) {}
		end = j + 2 < qlen ? j + 2 : qlen;
	}
	ExtRes r; r.score = max; r.qle = max_j + 1; r.tle = max_i + 1; r.gtle = max_ie + 1; r.gscore = gscore; r.max_off = max_off;
	return r;
}

struct U64Less { DEVFN bool operator()(const u64 &a, const u64 &b) const { return a < b; } };

DEVFN int opt_mat_max(const bwagpu_opt_t &opt)
{
	int m = 0;
	for (int k = 0; k < 25; ++k) if (opt.mat[k] > m) m = opt.mat[k];
	return m;
}

// mem_chain2aln over all chains of one read (the loop of mem_align1_core, bwamem.c:1096-1101)
__device__ void ext_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, i32 *H, i32 *E,
						 u64 &n_calls, u64 &n_cells, u64 &n_refb)
{
	int n_ch = B.chain_n[r];
	B.reg_n_raw[r] = 0;
	if (n_ch == 0) return;
	const u8 *query = B.seq + B.off[r];
	int l_query = (int)(B.off[r + 1] - B.off[r]);
	i64 so = B.seed_off[r], l_pac = ix.l_pac;
	const RegionView R = region_of(B.slot_blob, so, B.seed_n[r]);
	const bwagpu_chain_t *chains = R.cchain;
	const bwagpu_seed_t *seeds_all = R.cseed;
	u64 *srt_all = R.srt;
	bwagpu_alnreg_t *av = B.regs + B.reg_off[r];
	int n_av = 0, sbeg = 0, mat_max = opt_mat_max(opt);
	for (int ci = 0; ci < n_ch; ++ci) {
		const bwagpu_chain_t c = chains[ci];
		const bwagpu_seed_t *seeds = seeds_all + sbeg;
		u64 *srt = srt_all + sbeg;
		int n = c.n_seeds;
		sbeg += n;
		if (n == 0) continue;
		// reference window the chain may reach (bwamem.c:669-685)
		i64 rmax0 = l_pac << 1, rmax1 = 0;
		for (int i = 0; i < n; ++i) {
			bwagpu_seed_t t = seeds[i];
			i64 b = t.rbeg - (t.qbeg + dev_max_gap(opt, t.qbeg));
			i64 e = t.rbeg + t.len + ((l_query - t.qbeg - t.len) + dev_max_gap(opt, l_query - t.qbeg - t.len));
			if (b < rmax0) rmax0 = b;
			if (e > rmax1) rmax1 = e;
		}
		if (rmax0 < 0) rmax0 = 0;
		if (rmax1 > l_pac << 1) rmax1 = l_pac << 1;
		if (rmax0 < l_pac && l_pac < rmax1) { if (seeds[0].rbeg < l_pac) rmax1 = l_pac; else rmax0 = l_pac; }
		{	// bns_fetch_seq's clamp to the contig holding seeds[0] (bntseq.c:426-451)
			int is_rev; int rid = dev_pos2rid(ix, dev_depos(ix, seeds[0].rbeg, &is_rev));
			i64 fb = ix.ctg_off[rid], fe = fb + ix.ctg_len[rid];
			if (is_rev) { i64 t = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t; }
			if (rmax0 < fb) rmax0 = fb;
			if (rmax1 > fe) rmax1 = fe;
		}
		n_refb += (u64)(rmax1 - rmax0);
		for (int i = 0; i < n; ++i) srt[i] = (u64)seeds[i].score << 32 | (u32)i;
		dev_introsort(srt, n, U64Less());
		for (int k = n - 1; k >= 0; --k) {
			bwagpu_seed_t s = seeds[(u32)srt[k]];
			int ii;
			for (ii = 0; ii < n_av; ++ii) {   // already covered by an earlier alignment of this read? (bwamem.c:697-713)
				const bwagpu_alnreg_t &p = av[ii];
				i64 rd; int qd, w, mg;
				if (s.rbeg < p.rb || s.rbeg + s.len > p.re || s.qbeg < p.qb || s.qbeg + s.len > p.qe) continue;
				if (s.len - p.seedlen0 > .1 * l_query) continue;
				qd = s.qbeg - p.qb; rd = s.rbeg - p.rb;
				mg = dev_max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p.w ? mg : p.w;
				if (qd - rd < w && rd - qd < w) break;
				qd = p.qe - (s.qbeg + s.len); rd = p.re - (s.rbeg + s.len);
				mg = dev_max_gap(opt, qd < rd ? qd : (int)rd); w = mg < p.w ? mg : p.w;
				if (qd - rd < w && rd - qd < w) break;
			}
			if (ii < n_av) {   // contained: extend anyway only if an overlapping seed sits on another diagonal (bwamem.c:714-732)
				int i;
				for (i = k + 1; i < n; ++i) {
					if (srt[i] == 0) continue;
					bwagpu_seed_t t = seeds[(u32)srt[i]];
					if (t.len < s.len * .95) continue;
					if (s.qbeg <= t.qbeg && s.qbeg + s.len - t.qbeg >= s.len >> 2 && t.qbeg - s.qbeg != t.rbeg - s.rbeg) break;
					if (t.qbeg <= s.qbeg && t.qbeg + t.len - s.qbeg >= s.len >> 2 && s.qbeg - t.qbeg != s.rbeg - t.rbeg) break;
				}
				if (i == n) { srt[k] = 0; continue; }
			}
			bwagpu_alnreg_t a;
			a.rb = a.re = 0; a.qb = a.qe = 0; a.rid = c.rid; a.score = a.truesc = -1; a.sub = a.alt_sc = a.csub = a.sub_n = 0;
			a.w = opt.w; a.seedcov = 0; a.secondary = a.secondary_all = 0; a.seedlen0 = 0; a.n_comp = 0; a.is_alt = 0;
			a.frac_rep = 0.f; a.hash = 0;
			int aw0 = opt.w, aw1 = opt.w;
			if (s.qbeg) {   // left extension: reversed query prefix against the reversed reference prefix
				ExtRes x; x.qle = x.tle = x.gtle = 0; x.gscore = -1; x.max_off = 0; x.score = -1;
				int tl = (int)(s.rbeg - rmax0);
				for (int i = 0; i < 2; ++i) {
					int prev = a.score;
					aw0 = opt.w << i;
					x = dev_ksw_extend2(ix, opt, mat_max, query, s.qbeg - 1, -1, s.qbeg, s.rbeg - 1, -1, tl, aw0, opt.pen_clip5, s.len * opt.a, H, E, n_cells);
					++n_calls;
					a.score = x.score;
					if (a.score == prev || x.max_off < (aw0 >> 1) + (aw0 >> 2)) break;
				}
				if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip5) { a.qb = s.qbeg - x.qle; a.rb = s.rbeg - x.tle; a.truesc = a.score; }
				else { a.qb = 0; a.rb = s.rbeg - x.gtle; a.truesc = x.gscore; }
			} else { a.score = a.truesc = s.len * opt.a; a.qb = 0; a.rb = s.rbeg; }
			if (s.qbeg + s.len != l_query) {   // right extension
				ExtRes x; x.qle = x.tle = x.gtle = 0; x.gscore = -1; x.max_off = 0; x.score = -1;
				int sc0 = a.score, qe = s.qbeg + s.len;
				i64 re = s.rbeg + s.len;
				for (int i = 0; i < 2; ++i) {
					int prev = a.score;
					aw1 = opt.w << i;
					x = dev_ksw_extend2(ix, opt, mat_max, query, qe, 1, l_query - qe, re, 1, (int)(rmax1 - re), aw1, opt.pen_clip3, sc0, H, E, n_cells);
					++n_calls;
					a.score = x.score;
					if (a.score == prev || x.max_off < (aw1 >> 1) + (aw1 >> 2)) break;
				}
				if (x.gscore <= 0 || x.gscore <= a.score - opt.pen_clip3) { a.qe = qe + x.qle; a.re = re + x.tle; a.truesc += a.score - sc0; }
				else { a.qe = l_query; a.re = re + x.gtle; a.truesc += x.gscore - sc0; }
			} else { a.qe = l_query; a.re = s.rbeg + s.len; }
			int cov = 0;
			for (int i = 0; i < n; ++i) {
				bwagpu_seed_t t = seeds[i];
				if (t.qbeg >= a.qb && t.qbeg + t.len <= a.qe && t.rbeg >= a.rb && t.rbeg + t.len <= a.re) cov += t.len;
			}
			a.seedcov = cov;
			a.w = aw0 > aw1 ? aw0 : aw1;
			a.seedlen0 = s.len;
			a.frac_rep = c.frac_rep;
			av[n_av++] = a;
		}
	}
	B.reg_n_raw[r] = n_av;
}

__global__ void __launch_bounds__(256) k_extend(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
	int wave = tid >> 6, lane = tid & 63;
	i32 *H = B.dp_h + (size_t)wave * (B.max_len + 2) * DPS + lane;
	i32 *E = B.dp_e + (size_t)wave * (B.max_len + 2) * DPS + lane;
	u64 calls = 0, cells = 0, refb = 0, nraw = 0;
	for (int r = tid; r < B.n_reads; r += nth) { ext_read(ix, opt, B, r, H, E, calls, cells, refb); nraw += B.reg_n_raw[r]; }
	if (B.stats) {
		atomicAdd(&B.ctr->ext_calls, (unsigned long long)calls);
		atomicAdd(&B.ctr->ext_cells, (unsigned long long)cells);
		atomicAdd(&B.ctr->ref_bases, (unsigned long long)refb);
		atomicAdd(&B.ctr->n_regs_raw, (unsigned long long)nraw);
	}
}
