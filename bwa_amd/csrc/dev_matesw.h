// dev_matesw.h -- mate-rescue local alignments: the ksw_align2 calls of mem_matesw (bwamem_pair.c:137-206).
//
// Whether a rescue is attempted depends on the mate's region list, which grows as rescues succeed, so the decision stays
// with the host.  What can be precomputed is the expensive part: for every (anchor region, orientation) that the *initial*
// region lists do not already satisfy, the window and the local alignment of the mate inside it.  The host's mem_matesw
// looks a result up by (mate read, anchor position, anchor contig, orientation) and falls back to its own ksw_align2
// when there is none; rescues only add regions, so the initial lists can only over-estimate what is needed.
//
// k_matesw_tasks: one lane per pair enumerates the tasks.  k_matesw_sw: one lane per task runs ksw_align2 as the host code
// restates it (host_ksw.cpp: Farrar's striped layout matters only through its pad columns, the saturation rule of the byte
// kernel and the score2 / end-position bookkeeping), with its rows in lane-interleaved HBM scratch.
#pragma once
#include "dev_ext.h"

struct MateTask { i32 read, r; i64 anchor_rb; i32 anchor_rid, pad_; };

#define MSW_MAX_Q 512          // longest mate handled (query columns incl. padding: MSW_MAX_Q + 16)
#define MSW_MAX_T 2048         // longest window
#define MSW_QCOLS (MSW_MAX_Q + 16)
#define MSW_LANE_INTS (4 * MSW_QCOLS + 2 * MSW_MAX_T)

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

// sw_core of host_ksw.cpp (== ksw_u8 / ksw_i16, ksw.c:122-377, as far as results go).  Q(j) / T(i) deliver query and target codes;
// S[] is this lane's scratch, element k at S[k * 64] (lane-interleaved within the wave).
template <class QF, class TF>
__device__ MswRes dev_sw_core(int size, int qlen, QF Q, int tlen, TF T, const bwagpu_opt_t &opt, int xtra, i32 *S)
{
	const int pw = size == 1 ? 16 : 8, slen = (qlen + pw - 1) / pw, qpad = slen * pw;
	int mn = 127, mx = 0;
	for (int a = 0; a < 25; ++a) { if (opt.mat[a] < mn) mn = opt.mat[a]; if (opt.mat[a] > mx) mx = opt.mat[a]; }
	const int shift = (int)(u8)(256 - mn);
	const int minsc = (xtra & MSW_XSUBO) ? (xtra & 0xffff) : 0x10000, endsc = (xtra & MSW_XSTOP) ? (xtra & 0xffff) : 0x10000;
	const int e_del = opt.e_del, e_ins = opt.e_ins, oe_del = opt.o_del + e_del, oe_ins = opt.o_ins + e_ins;
	const int cap = size == 1 ? 255 : 32767;
	i32 *H = S, *E = S + (size_t)MSW_QCOLS * 64, *Hn = S + (size_t)2 * MSW_QCOLS * 64, *Hmax = S + (size_t)3 * MSW_QCOLS * 64, *B = S + (size_t)4 * MSW_QCOLS * 64;
	for (int j = 0; j < qpad; ++j) { H[j * 64] = 0; E[j * 64] = 0; Hmax[j * 64] = 0; }
	int nb = 0, gmax = 0, te = -1;
	MswRes r; r.score = 0; r.te = -1; r.qe = -1; r.score2 = -1; r.te2 = -1;
	for (int i = 0; i < tlen; ++i) {
		const int tb = T(i);
		int f = 0, hdiag = 0, imax = 0;
		for (int j = 0; j < qpad; ++j) {
			int h = hdiag + (j < qlen ? (int)opt.mat[tb * 5 + Q(j)] : 0), e = E[j * 64], t;
			if (h < 0) h = 0;
			if (h > cap) h = cap;
			hdiag = H[j * 64];
			if (h < e) h = e;
			if (h < f) h = f;
			Hn[j * 64] = h;
			if (h > imax) imax = h;
			e -= e_del; if (e < 0) e = 0; t = h - oe_del; if (t < 0) t = 0; E[j * 64] = e > t ? e : t;
			f -= e_ins; if (f < 0) f = 0; t = h - oe_ins; if (t < 0) t = 0; if (t > f) f = t;
		}
		if (imax >= minsc) {   // runs of consecutive rows reaching minsc keep their best row (ksw.c:215-223)
			if (nb == 0 || B[(2 * (nb - 1) + 1) * 64] + 1 != i) { B[(2 * nb) * 64] = imax; B[(2 * nb + 1) * 64] = i; ++nb; }
			else if (B[(2 * (nb - 1)) * 64] < imax) { B[(2 * (nb - 1)) * 64] = imax; B[(2 * (nb - 1) + 1) * 64] = i; }
		}
		{ i32 *tmp = H; H = Hn; Hn = tmp; }
		if (imax > gmax) {
			gmax = imax; te = i;
			for (int j = 0; j < qpad; ++j) Hmax[j * 64] = H[j * 64];
			if ((size == 1 && gmax + shift >= 255) || gmax >= endsc) break;
		}
	}
	r.score = (size == 1 && gmax + shift >= 255) ? 255 : gmax;
	r.te = te;
	if (!(size == 1 && r.score == 255)) {
		int best = -1;
		for (int j = 0; j < qpad; ++j) if (Hmax[j * 64] > best) { best = Hmax[j * 64]; r.qe = j; }
		if (nb) {
			const int d = (r.score + mx - 1) / mx, low = te - d, high = te + d;
			for (int k = 0; k < nb; ++k) {
				const int v = B[(2 * k) * 64], row = B[(2 * k + 1) * 64];
				if ((row < low || row > high) && v > r.score2) { r.score2 = v; r.te2 = row; }
			}
		}
	}
	return r;
}

// ksw_align2 (ksw.c:379-400): the forward pass and, with KSW_XSTART, the pass over the reversed prefixes that finds the start
// positions.  res = {score, te, qe, score2, te2, tb, qb}.
template <class QF, class TF>
__device__ void msw_align2(const bwagpu_opt_t &opt, int qlen, QF Qf, int tlen, TF Tf, int xtra, i32 *S, int res[7])
{
	const int size = (xtra & MSW_XBYTE) ? 1 : 2;
	const MswRes a = dev_sw_core(size, qlen, Qf, tlen, Tf, opt, xtra, S);
	res[0] = a.score; res[1] = a.te; res[2] = a.qe; res[3] = a.score2; res[4] = a.te2; res[5] = -1; res[6] = -1;
	if (((xtra & MSW_XSTART) == 0) || ((xtra & MSW_XSUBO) && a.score < (xtra & 0xffff))) return;
	const int qe = a.qe, te = a.te;
	auto Q2 = [&](int j) -> int { return Qf(qe - j); };
	auto T2 = [&](int i) -> int { return i <= te ? Tf(te - i) : Tf(i); };
	const MswRes b = dev_sw_core(size, qe + 1, Q2, tlen, T2, opt, MSW_XSTOP | a.score, S);
	if (a.score == b.score) { res[5] = a.te - b.te; res[6] = a.qe - b.qe; }
}

// One lane per task (tasks drawn from a counter); out[t] is written for every task, r = -1 marking "no alignment was due".
__global__ void __launch_bounds__(256) k_matesw_sw(DevIndex ix, bwagpu_opt_t opt, Batch Bt, const bwagpu_pes_t *pes, const MateTask *tasks, i64 n_tasks,
													bwagpu_matesw_t *out, unsigned long long *next, i32 *scratch)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; const int lane = threadIdx.x & 63;
	i32 *S = scratch + wave * ((size_t)MSW_LANE_INTS * 64) + lane;
	const i64 l_pac = ix.l_pac;
	for (;;) {
		const i64 t = (i64)atomicAdd(next, 1ull);
		if (t >= n_tasks) break;
		const MateTask k = tasks[t];
		bwagpu_matesw_t o;
		o.read = k.read; o.r = -1; o.anchor_rb = k.anchor_rb; o.anchor_rid = k.anchor_rid;
		o.score = 0; o.te = o.qe = o.score2 = o.te2 = o.tb = o.qb = -1; o.pad_ = 0; o.pad2_ = 0;
		const u8 *ms = Bt.seq + Bt.off[k.read];
		const int l_ms = (int)(Bt.off[k.read + 1] - Bt.off[k.read]);
		const int r = k.r, is_rev = (r >> 1) != (r & 1), is_larger = !(r >> 1);
		i64 rb, re;
		if (!is_rev) {
			rb = is_larger ? k.anchor_rb + pes[r].low : k.anchor_rb - pes[r].high;
			re = (is_larger ? k.anchor_rb + pes[r].high : k.anchor_rb - pes[r].low) + l_ms;
		} else {
			rb = (is_larger ? k.anchor_rb + pes[r].low : k.anchor_rb - pes[r].high) - l_ms;
			re = is_larger ? k.anchor_rb + pes[r].high : k.anchor_rb - pes[r].low;
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
			due = k.anchor_rid == rid && re - rb >= opt.min_seed_len && re - rb <= MSW_MAX_T;
		}
		if (due) {
			const int tlen = (int)(re - rb);
			const int xtra = MSW_XSUBO | MSW_XSTART | (l_ms * opt.a < 250 ? MSW_XBYTE : 0) | (opt.min_seed_len * opt.a);
			auto Qf = [&](int j) -> int { return is_rev ? (ms[l_ms - 1 - j] < 4 ? 3 - ms[l_ms - 1 - j] : 4) : (int)ms[j]; };
			auto Tf = [&](int i) -> int { return ref_base(ix, rb + i); };
			int res[7];
			msw_align2(opt, l_ms, Qf, tlen, Tf, xtra, S, res);
			o.r = r; o.score = res[0]; o.te = res[1]; o.qe = res[2]; o.score2 = res[3]; o.te2 = res[4]; o.tb = res[5]; o.qb = res[6];
		}
		out[t] = o;
	}
}
