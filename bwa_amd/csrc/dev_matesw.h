// dev_matesw.h -- mate-rescue local alignments: the ksw_align2 calls of mem_matesw (bwamem_pair.c:137-206).
//
// Whether a rescue is attempted depends on the mate's region list, which grows as rescues succeed, so the decision stays
// with the host.  What can be precomputed is the expensive part: for every (anchor region, orientation) that the *initial*
// region lists do not already satisfy, the window and the local alignment of the mate inside it.  The host's mem_matesw
// looks a result up by (mate read, anchor position, anchor contig, orientation) and falls back to its own ksw_align2
// when there is none; rescues only add regions, so the initial lists can only over-estimate what is needed.
//
// k_matesw_tasks: one lane per pair enumerates the tasks.  k_matesw_sw: one wavefront per task runs ksw_align2 as the host code
// restates it (host_ksw.cpp: Farrar's striped layout matters only through its pad columns, the saturation rule of the byte
// kernel and the score2 / end-position bookkeeping), row-parallel with its columns in registers (round 2's one-lane-per-task form with
// rows in HBM took 127 ms for a batch's 11.5 k alignments -- as long as the whole hot path -- because a lane's serial 75 k cells set the pace).
#pragma once
#include "dev_extw.h"

struct MateTask { i32 read, r; i64 anchor_rb; i32 anchor_rid, pad_; };

#define MSW_MAX_Q 512          // longest mate handled (query columns incl. padding: MSW_MAX_Q + 16)
#define MSW_MAX_T 2048         // longest window
#define MSW_QCOLS (MSW_MAX_Q + 16)

DEVFN int dev_infer_dir(i64 l_pac, i64 b1, i64 b2, i64 *dist)
{	// mem_infer_dir (bwamem_pair.c:49-56)
	const int r1 = b1 >= l_pac, r2 = b2 >= l_pac;
	const i64 p2 = r1 == r2 ? b2 : (l_pac << 1) - 1 - b2;
	*dist = p2 > b1 ? p2 - b1 : b1 - p2;
	return (r1 == r2 ? 0 : 1) ^ (p2 > b1 ? 0 : 3);
}

// One lane per pair: reads 2p and 2p+1 of the batch; regs/cnt/off are the packed regions of bwagpu_batch_download.
__global__ void __launch_bounds__(256) k_matesw_tasks(DevIndex ix, bwagpu_opt_t opt, int n_reads, const i32 *cnt, const i64 *off, const bwagpu_alnreg_t *regs,
													   const bwagpu_pes_t *pes, MateTask *tasks, unsigned long long *n_tasks, i64 task_cap)
{
	for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n_reads / 2; p += gridDim.x * blockDim.x) {
		for (int i = 0; i < 2; ++i) {
			const int ri = 2 * p + i, rm = 2 * p + (1 - i);
			const int ni = cnt[ri], nm = cnt[rm];
			if (ni == 0) continue;
			const bwagpu_alnreg_t *a = regs + off[ri], *ma = regs + off[rm];
			const int best = a[0].score;
			int taken = 0;
			for (int j = 0; j < ni && taken < opt.max_matesw; ++j) {
				if (a[j].score < best - opt.pen_unpaired) continue;
				++taken;
				int skip[4];
				for (int r = 0; r < 4; ++r) skip[r] = pes[r].failed ? 1 : 0;
				for (int k = 0; k < nm; ++k) {
					i64 dist;
					const int r = dev_infer_dir(ix.l_pac, a[j].rb, ma[k].rb, &dist);
					if (dist >= pes[r].low && dist <= pes[r].high) skip[r] = 1;
				}
				for (int r = 0; r < 4; ++r) {
					if (skip[r]) continue;
					const unsigned long long t = atomicAdd(n_tasks, 1ull);
					if ((i64)t < task_cap) { MateTask m; m.read = rm; m.r = r; m.anchor_rb = a[j].rb; m.anchor_rid = a[j].rid; m.pad_ = 0; tasks[t] = m; }
				}
			}
		}
	}
}

struct MswRes { int score, te, qe, score2, te2; };
enum { MSW_XBYTE = 0x10000, MSW_XSTOP = 0x20000, MSW_XSUBO = 0x40000, MSW_XSTART = 0x80000 };
#define MSW_RUN_INTS (MSW_MAX_T + 4)     // LDS ints per wave: the (row maximum, row) runs of the score2 bookkeeping, at most one per two rows

// ksw_u8 / ksw_i16 (ksw.c:122-377) as far as results go -- the restatement host_ksw.cpp's sw_core pins against the reference: plain Gotoh
// local alignment with gaps opened from H, Farrar's striped layout visible only through its pad columns (the query is padded to a multiple
// of 16 or 8 columns), the saturation of the byte kernel, the run-merging score2 / te2 bookkeeping (ksw.c:215-223) and the column snapshot
// at the best row from which qe is taken (:237-239).  One wavefront per alignment, lanes over the query's columns:
//     F(i,j) = max(0, max_{k<j} (max(M, E)(i,k) - oe_ins - (j-1-k) e_ins))
// because H >= F and o >= 0 make "open from H" and "open from max(M, E)" the same thing, so a row is a max-plus prefix scan as in the
// extension kernel.  A lane owns columns lane, lane + 64, ... (NP of them); H(i-1,.), E and the snapshot live in registers, the previous
// row's neighbour comes by a lane shift, and the only memory a row touches is the packed reference.
template <int NP, class QF, class TF>
__device__ MswRes wave_sw_core(int size, int qlen, QF Q, int tlen, TF T, const bwagpu_opt_t &opt, int xtra, i32 *runs)
{
	const int lane = threadIdx.x & 63;
	const int pw = size == 1 ? 16 : 8, qpad = (qlen + pw - 1) / pw * pw;
	int mn = 127, mx = 0;
	for (int a = 0; a < 25; ++a) { if (opt.mat[a] < mn) mn = opt.mat[a]; if (opt.mat[a] > mx) mx = opt.mat[a]; }
	const int shift = (int)(u8)(256 - mn);
	const int minsc = (xtra & MSW_XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & MSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	const int cap = size == 1 ? 255 : 32767;
	int pk[NP], Hp[NP], E[NP], Hm[NP];          // per owned column: scores against A,C,G,T (a byte each), H(i-1,j), E(i,j), H at the best row
#pragma unroll
	for (int p = 0; p < NP; ++p) {
		const int j = 64 * p + lane;
		const int qc = j < qlen ? Q(j) : -1;
		pk[p] = qc < 0 ? 0 : (int)((u32)(u8)opt.mat[qc] | (u32)(u8)opt.mat[5 + qc] << 8 | (u32)(u8)opt.mat[10 + qc] << 16 | (u32)(u8)opt.mat[15 + qc] << 24);
		Hp[p] = E[p] = Hm[p] = 0;
	}
	int nb = 0, gmax = 0, te = -1, last_v = 0, last_row = -2, treg = 0;
	for (int i = 0; i < tlen; ++i) {
		if ((i & 63) == 0) treg = i + lane < tlen ? T(i + lane) : 0;
		const int tb = __builtin_amdgcn_readlane(treg, i & 63);
		int carry = W_NEG, rowmax = 0, Hn[NP];
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const int j = 64 * p + lane;
			if (64 * p < qpad) {                           // (uniform)
				const bool act = j < qpad;
				const int dg = wave_shift_up1(Hp[p], p ? __builtin_amdgcn_readlane(Hp[p ? p - 1 : 0], 63) : 0);    // H(i-1, j-1)
				const int sc = j < qlen ? (int)(int8_t)(u8)((u32)pk[p] >> (8 * tb)) : 0;
				int M = dg + sc; M = M < 0 ? 0 : (M > cap ? cap : M);
				const int hme = imax(M, E[p]);
				const int inc = wave_incl_scan_max(act ? hme - oe_ins + j * e_ins : W_NEG);
				const int exc = imax(wave_shift_up1(inc, W_NEG), carry);
				const int h = imax(hme, imax(exc - (j - 1) * e_ins, 0));
				Hn[p] = act ? h : 0;
				rowmax = imax(rowmax, Hn[p]);
				E[p] = imax(imax(E[p] - e_del, h - oe_del), 0);
				carry = imax(carry, __builtin_amdgcn_readlane(inc, 63));
			} else Hn[p] = 0;
		}
		const int im = __builtin_amdgcn_readlane(wave_incl_scan_max(rowmax), 63);
		if (im >= minsc) {   // runs of consecutive rows reaching minsc keep their best row (ksw.c:215-223); the open run lives in registers
			if (nb == 0 || last_row + 1 != i) { if (nb > 0 && lane == 0) { runs[2 * (nb - 1)] = last_v; runs[2 * (nb - 1) + 1] = last_row; } ++nb; last_v = im; last_row = i; }
			else if (last_v < im) { last_v = im; last_row = i; }
		}
#pragma unroll
		for (int p = 0; p < NP; ++p) Hp[p] = Hn[p];
		if (im > gmax) {
			gmax = im; te = i;
#pragma unroll
			for (int p = 0; p < NP; ++p) Hm[p] = Hp[p];
			if ((size == 1 && gmax + shift >= 255) || gmax >= endsc) break;
		}
	}
	if (nb > 0 && lane == 0) { runs[2 * (nb - 1)] = last_v; runs[2 * (nb - 1) + 1] = last_row; }
	wave_sync();
	MswRes r; r.score = (size == 1 && gmax + shift >= 255) ? 255 : gmax; r.te = te; r.qe = -1; r.score2 = -1; r.te2 = -1;
	if (!(size == 1 && r.score == 255)) {
		// qe: the first column of the snapshot holding its maximum (ksw.c:237-239); the snapshot's maximum is the best row's
		int best = 0;
#pragma unroll
		for (int p = 0; p < NP; ++p) best = imax(best, (64 * p + lane < qpad) ? Hm[p] : 0);
		best = __builtin_amdgcn_readlane(wave_incl_scan_max(best), 63);
#pragma unroll
		for (int p = 0; p < NP; ++p) {
			const u64 m = __ballot(64 * p + lane < qpad && Hm[p] == best);
			if (m && r.qe < 0) r.qe = 64 * p + __builtin_ctzll(m);
		}
		if (nb) {
			const int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
			for (int k = 0; k < nb; ++k) {
				const int v = runs[2 * k], row = runs[2 * k + 1];
				if ((row < low || row > high) && v > r.score2) { r.score2 = v; r.te2 = row; }
			}
		}
	}
	wave_sync();
	return r;
}

// ksw_align2 (ksw.c:379-400): the forward pass and, with KSW_XSTART, the pass over the reversed prefixes that finds the start
// positions.  res = {score, te, qe, score2, te2, tb, qb}.
template <int NP, class QF, class TF>
__device__ void wave_align2(const bwagpu_opt_t &opt, int qlen, QF Qf, int tlen, TF Tf, int xtra, i32 *runs, int res[7])
{
	const int size = (xtra & MSW_XBYTE) ? 1 : 2;
	const MswRes a = wave_sw_core<NP>(size, qlen, Qf, tlen, Tf, opt, xtra, runs);
	res[0] = a.score; res[1] = a.te; res[2] = a.qe; res[3] = a.score2; res[4] = a.te2; res[5] = -1; res[6] = -1;
	if (((xtra & MSW_XSTART) == 0) || ((xtra & MSW_XSUBO) && a.score < (xtra & 0xffff))) return;
	const int qe = a.qe, te = a.te;
	auto Q2 = [&](int j) -> int { return Qf(qe - j); };
	auto T2 = [&](int i) -> int { return i <= te ? Tf(te - i) : Tf(i); };
	const MswRes b = wave_sw_core<NP>(size, qe + 1, Q2, tlen, T2, opt, MSW_XSTOP | a.score, runs);
	if (a.score == b.score) { res[5] = a.te - b.te; res[6] = a.qe - b.qe; }
}
// (two instances: up to 192 padded query columns -- every 150 bp mate -- and up to MSW_QCOLS)
template <class QF, class TF>
__device__ void msw_align2(const bwagpu_opt_t &opt, int qlen, QF Qf, int tlen, TF Tf, int xtra, i32 *runs, int res[7])
{
	if (qlen <= 176) wave_align2<3>(opt, qlen, Qf, tlen, Tf, xtra, runs, res);
	else wave_align2<(MSW_QCOLS + 63) / 64>(opt, qlen, Qf, tlen, Tf, xtra, runs, res);
}

// One wavefront per task (tasks drawn from a counter); out[t] is written for every task, r = -1 marking "no alignment was due".
__global__ void __launch_bounds__(256) k_matesw_sw(DevIndex ix, bwagpu_opt_t opt, Batch Bt, const bwagpu_pes_t *pes, const MateTask *tasks, i64 n_tasks,
													bwagpu_matesw_t *out, unsigned long long *next)
{
	__shared__ i32 msw_runs[4 * MSW_RUN_INTS];
	const int lane = threadIdx.x & 63;
	i32 *runs = msw_runs + (threadIdx.x >> 6) * MSW_RUN_INTS;
	const i64 l_pac = ix.l_pac;
	WaveQueue wq; wq_init(wq); wq.step = WQ_CHUNK;
	for (;;) {
		long long t;
		if (!wq_next(wq, next, n_tasks, t)) break;
		const MateTask k = tasks[t];
		bwagpu_matesw_t o;
		o.read = k.read; o.r = -1; o.anchor_rb = k.anchor_rb; o.anchor_rid = k.anchor_rid;
		o.score = 0; o.te = o.qe = o.score2 = o.te2 = o.tb = o.qb = -1; o.pad_ = 0; o.pad2_ = 0;
		const u8 *ms = Bt.seq + Bt.off[k.read];
		const int l_ms = uni((int)(Bt.off[k.read + 1] - Bt.off[k.read]));
		const int r = uni(k.r), is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
		const i64 arb = uni64(k.anchor_rb);
		i64 rb, re;
		if (!is_rev) {
			rb = is_larger ? arb + pes[r].low : arb - pes[r].high;
			re = (is_larger ? arb + pes[r].high : arb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? arb + pes[r].low : arb - pes[r].high) - l_ms;
			re = is_larger ? arb + pes[r].high : arb - pes[r].low;
		}
		if (rb < 0) rb = 0;
		if (re > l_pac << 1) re = l_pac << 1;
		bool due = rb < re && l_ms <= MSW_MAX_Q;
		int rid = -1;
		if (due) {   // bns_fetch_seq: clamp to the contig of the window's midpoint (bntseq.c:426-443)
			const i64 mid = (rb + re) >> 1;
			int mrev; rid = dev_pos2rid(ix, dev_depos(ix, mid, &mrev));
			i64 fb = ix.ctg_off[rid], fe = fb + ix.ctg_len[rid];
			if (mrev) { const i64 t2 = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t2; }
			if (rb < fb) rb = fb;
			if (re > fe) re = fe;
			due = uni(k.anchor_rid) == rid && re - rb >= opt.min_seed_len && re - rb <= MSW_MAX_T;
		}
		rb = uni64(rb); re = uni64(re);
		if (due) {
			const int tlen = (int)(re - rb);
			const int xtra = MSW_XSUBO | MSW_XSTART | (l_ms * opt.a < 250 ? MSW_XBYTE : 0) | (opt.min_seed_len * opt.a);
			auto Qf = [&](int j) -> int { return is_rev ? (ms[l_ms - 1 - j] < 4 ? 3 - ms[l_ms - 1 - j] : 4) : (int)ms[j]; };
			auto Tf = [&](int i) -> int { return ref_base(ix, rb + i); };
			int res[7];
			msw_align2(opt, l_ms, Qf, tlen, Tf, xtra, runs, res);
			o.r = r; o.score = res[0]; o.te = res[1]; o.qe = res[2]; o.score2 = res[3]; o.te2 = res[4]; o.tb = res[5]; o.qb = res[6];
		}
		if (lane == 0) out[t] = o;
	}
}
