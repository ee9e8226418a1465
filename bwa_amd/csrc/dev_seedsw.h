// dev_seedsw.h -- mem_flt_chained_seeds / mem_seed_sw (bwamem.c:597-645): for long reads only, every seed
// shorter than 200 bp is re-scored by a local alignment of the seed +-50 bp (query) against +-50 bp (reference);
// seeds scoring below min_HSP_score are dropped.  Only the score of ksw_align2 is used (bwamem.c:619-621), and
// that equals the plain Gotoh local-alignment score (gaps opened from H, everything clamped at 0; SURVEY.md
// App. A.12), which is what is computed here.
#pragma once
#include "dev_ext.h"

__device__ int dev_local_score(const DevIndex &ix, const bwagpu_opt_t &opt, const u8 *q, int qlen, i64 t0, int tlen, i32 *H, i32 *E, u64 &cells)
{
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	int best = 0;
	for (int j = 0; j < qlen; ++j) { H[j * DPS] = 0; E[j * DPS] = 0; }
	for (int i = 0; i < tlen; ++i) {
		const int8_t *srow = opt.mat + ref_base(ix, t0 + i) * 5;
		int f = 0, hdiag = 0;
		for (int j = 0; j < qlen; ++j) {
			int h = hdiag + srow[q[j]], e = E[j * DPS], t;
			hdiag = H[j * DPS];
			if (h < e) h = e;
			if (h < f) h = f;
			if (h < 0) h = 0;
			H[j * DPS] = h;
			if (h > best) best = h;
			t = h - oe_del; if (t < 0) t = 0; e -= e_del; if (e < 0) e = 0; E[j * DPS] = e > t ? e : t;
			t = h - oe_ins; if (t < 0) t = 0; f -= e_ins; if (f < 0) f = 0; if (t > f) f = t;
		}
	}
	cells += (u64)qlen * tlen;
	return best;
}

// The same score with everything the cell loop touches in LDS (k_seedsw_wave<CB>, BWAGPU_SEEDSW_LDS=1).  The window is under 200 x 200
// (bwamem.c:611): per lane, 200 columns of {H, E} packed into one cell of 2 x CB bits (CB = 8 when every score stays below 256 -- a = 1, the
// presets' value -- else 16), [column][lane], and the window's query bases as bytes next to them; the five rows of the scoring matrix are
// five 64-bit words in scalar registers (mat_rows), a cell's score a shift of the row its target base selects.  In the form above every
// cell waits for four dependent memory round trips (its query base, the matrix entry, E and H from HBM scratch) in a kernel that runs
// one wave per SIMD; here it waits for two independent LDS reads.  H, E and F are clamped at 0 by the recurrence, so unsigned cells hold them.
#define SEEDSW_LDS_COLS 200
template <int CB> struct SwCell;
template <> struct SwCell<8> { typedef unsigned short T; };
template <> struct SwCell<16> { typedef u32 T; };
DEVFN void mat_rows(const bwagpu_opt_t &opt, u64 rows[5])
{
	for (int b = 0; b < 5; ++b) { u64 r = 0; for (int k = 0; k < 5; ++k) r |= (u64)(u8)opt.mat[b * 5 + k] << (8 * k); rows[b] = r; }
}
template <int CB> __device__ int dev_local_score_lds(const DevIndex &ix, const bwagpu_opt_t &opt, const u64 rows[5], const u8 *q, int qlen, i64 t0, int tlen,
													   typename SwCell<CB>::T *HE, u8 *Q, u64 &cells)
{
	typedef typename SwCell<CB>::T cell_t;
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	const u32 mask = (1u << CB) - 1;
	int best = 0;
	for (int j = 0; j < qlen; ++j) { HE[j * 64] = 0; Q[j * 64] = q[j]; }
	for (int i = 0; i < tlen; ++i) {
		const int tb = ref_base(ix, t0 + i);
		const u64 row = tb == 0 ? rows[0] : tb == 1 ? rows[1] : tb == 2 ? rows[2] : tb == 3 ? rows[3] : rows[4];
		int f = 0, hdiag = 0;
		for (int j = 0; j < qlen; ++j) {
			const u32 he = HE[j * 64];
			const int qc = Q[j * 64];
			int h = hdiag + (int)(int8_t)(u8)(row >> (8 * qc)), e = (int)(he >> CB), t;
			hdiag = (int)(he & mask);
			if (h < e) h = e;
			if (h < f) h = f;
			if (h < 0) h = 0;
			if (h > best) best = h;
			t = h - oe_del; if (t < 0) t = 0; e -= e_del; if (e < 0) e = 0; if (e < t) e = t;
			HE[j * 64] = (cell_t)((u32)e << CB | (u32)h);
			t = h - oe_ins; if (t < 0) t = 0; f -= e_ins; if (f < 0) f = 0; if (t > f) f = t;
		}
	}
	cells += (u64)qlen * tlen;
	return best;
}

__device__ void seedsw_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, i32 *H, i32 *E, u64 &calls, u64 &cells)
{
	int n_ch = B.chain_n[r];
	if (n_ch == 0) return;
	int l_query = (int)(B.off[r + 1] - B.off[r]);
	int min_hsp = B.seedsw_minhsp[l_query];
	if (min_hsp < 0) return;                       // "don't run the following for short reads" (bwamem.c:628)
	const u8 *query = B.seq + B.off[r];
	i64 so = B.seed_off[r], l_pac = ix.l_pac;
	const RegionView R = region_of(B.slot_blob, so, B.seed_n[r]);
	bwagpu_chain_t *chains = R.cchain;
	bwagpu_seed_t *seeds = R.cseed;
	int sbeg = 0, m = 0;
	for (int ci = 0; ci < n_ch; ++ci) {
		int n = chains[ci].n_seeds, k = 0;
		for (int j = 0; j < n; ++j) {
			bwagpu_seed_t s = seeds[sbeg + j];
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
					if (is_rev) { i64 t = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t; }
					if (rb < fb) rb = fb;
					if (re > fe) re = fe;
					sc = dev_local_score(ix, opt, query + qb, qe - qb, rb, (int)(re - rb), H, E, cells);
					++calls;
				}
			}
			s.score = sc;
			if (s.score < 0 || s.score >= min_hsp) {
				if (s.score < 0) s.score = s.len * opt.a;
				seeds[m + k] = s; ++k;
			}
		}
		sbeg += n; m += k;
		chains[ci].n_seeds = k;
	}
}

__global__ void __launch_bounds__(256) k_seedsw(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
	int wave = tid >> 6, lane = tid & 63;
	i32 *H = B.dp_h + (size_t)wave * (B.max_len + 2) * DPS + lane;
	i32 *E = B.dp_e + (size_t)wave * (B.max_len + 2) * DPS + lane;
	u64 calls = 0, cells = 0;
	for (int r = tid; r < B.n_reads; r += nth) seedsw_read(ix, opt, B, r, H, E, calls, cells);
	if (B.stats) { atomicAdd(&B.ctr->sw_calls, (unsigned long long)calls); atomicAdd(&B.ctr->sw_cells, (unsigned long long)cells); }
}
