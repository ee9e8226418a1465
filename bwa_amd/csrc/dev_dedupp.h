// dev_dedupp.h -- mem_sort_dedup_patch (bwamem.c:463-515) for the short reads with several regions, one wavefront per read, the lanes over the regions.
// (Included by dev_dedupw.h, whose kernel k_dedup_wave<.., LIST = true> calls it and whose wave_patch_reg it uses.)
//
// The lane-per-read kernel (dev_dedup.h) is as slow as its heaviest lane: of an average 64 reads of the headline's batch one has 56 regions, the
// other 63 have 3.2 between them, and one read of the batch has 900 (profiles/r06_dedup.md); every step of its sorts and of its pair loop waits
// for HBM.  Here the decisions' operands -- six fields of a region -- sit in LDS, the 88-byte records stay where they are until the end, and
// what the reference does one element at a time is done 64 at a time wherever the order of the results is fixed by the reference's DEFINITION
// rather than by its sequence of steps:
//   * ks_introsort (ksort.h:176-226) = quicksort passes, then an insertion sort.  The passes (lane 0, on 24-byte keys in LDS) leave ranges of up
//     to 16 elements unsorted; the insertion sort is a stable sort of that arrangement and a stable sort's result is its definition, so every
//     lane counts how many keys go ahead of its own (as for the chain weights, dev_chainw.h).
//   * the redundancy scan (bwamem.c:470-497) compares region i with regions i-1, i-2, .. while they are within max_chain_gap.  What happens at j
//     depends on earlier j only through p: a redundant q with the lower score dies (q only), a redundant q with the higher score kills p and
//     ends the scan, a successful patch alignment changes p.  So 64 lanes evaluate 64 j, all the q that die before the first lane that ends
//     the scan or wants a patch alignment do so at once, and only that lane's event is handled in sequence (29 patch alignments in a million
//     reads; the scan goes on with the changed p behind it).
//   * the two compactions and the duplicate marks (bwamem.c:498-513) are prefix counts and neighbour compares.
// At the end the kept records are gathered in their final order into the wave's staging area and copied back over the read's records.
#pragma once

struct DdHot { i64 rb, re; i32 qb, qe, rid, score; };     // what the decisions read
// What the sorts move: 16 bytes, the order in the upper 96 bits, the region's index below (not compared: equal keys must stay equal, their order is
// ks_introsort's).  By end (bwamem.c:467): hi = re.  By score, rb, qb (bwamem.c:504): hi = ~score : rb[47:16], lo = rb[15:0] : qb : index -- which needs
// 0 <= rb < 2^48 and 0 <= qb < 2^16: the host sends a batch here only when the index and the reads are that small (every index there is; reads of
// short-read batches).
// (Sixteen bytes also because the 24-byte RegKey of dev_dedup.h did not survive here: lane 0's quicksort passes over RegKey records in LDS never came
// back on the device -- bisected with early-exit builds to dev_introsort<RegKey, KeyBestLess, false>, sessions r6-23..25; the mock runtime and a build
// with printf calls between the phases ran it correctly.  Its swaps are compiled as two overlapping 16-byte copies through a scratch temporary; these
// are single 16-byte LDS reads and writes.  Not understood further.)
struct alignas(16) DdKey { u64 hi, lo; };
struct DdKeyLessEnd { DEVFN bool operator()(const DdKey &x, const DdKey &y) const { return x.hi < y.hi; } };
struct DdKeyLessBest { DEVFN bool operator()(const DdKey &x, const DdKey &y) const { return x.hi < y.hi || (x.hi == y.hi && (x.lo >> 32) < (y.lo >> 32)); } };
DEVFN DdKey ddp_key_end(const DdHot &h, int idx) { DdKey k; k.hi = (u64)h.re ^ ((u64)1 << 63); k.lo = (u32)idx; return k; }
DEVFN DdKey ddp_key_best(const DdHot &h, int idx)
{
	DdKey k;
	k.hi = (u64)(u32)~((u32)h.score ^ 0x80000000u) << 32 | (u32)((u64)h.rb >> 16);
	k.lo = ((u64)h.rb & 0xffffu) << 48 | (u64)((u32)h.qb & 0xffffu) << 32 | (u32)idx;
	return k;
}
#define DDP_LDS_PER_REG (sizeof(DdHot) + sizeof(DdKey) + 8)       // + the order before and after a compaction
static_assert(sizeof(DdHot) == 32 && sizeof(DdKey) == 16, "LDS arrays of dedup_read_par");
static_assert(sizeof(bwagpu_alnreg_t) % 4 == 0, "records are moved word by word");
#define DDP_REG_WORDS ((int)(sizeof(bwagpu_alnreg_t) / 4))

// The stable sort that finishes ks_introsort: out[place of keys[x]] = keys[x]'s index.  BEST: the order is (hi, lo >> 32), else hi alone.
template <bool BEST> DEVFN void ddp_rank(const DdKey *keys, int n, i32 *out, int lane)
{
	for (int x0 = 0; x0 < n; x0 += 64) {
		const int x = x0 + lane;
		u64 hx = 0, lx = 0; int idx = 0;
		if (x < n) { const DdKey k = keys[x]; hx = k.hi; lx = k.lo >> 32; idx = (int)(u32)k.lo; }
		int place = 0;
		for (int c = 0; c < n; c += 64) {
			const int y = c + lane;
			u64 hy = 0; u32 ly = 0;
			if (y < n) { const DdKey k = keys[y]; hy = k.hi; ly = (u32)(k.lo >> 32); }
			const int cnt = n - c < 64 ? n - c : 64;
			for (int t = 0; t < cnt; ++t) {
				const u64 h = (u64)readlane_i64_((i64)hy, t);
				if (BEST) {
					const u64 l = (u32)__builtin_amdgcn_readlane((int)ly, t);
					place += (h < hx || (h == hx && (l < lx || (l == lx && c + t < x)))) ? 1 : 0;
				} else place += (h < hx || (h == hx && c + t < x)) ? 1 : 0;
			}
		}
		if (x < n) out[place] = idx;
	}
}

// The same finish for the reads with hundreds of regions, where counting is n^2 / 64 steps (a read of 900 regions: most of its 2 ms): with its place in
// the array as the last part of its key every element is distinct, the stable order is a total one, and any sorting network produces it -- a bitonic
// one here, over N = the power of two above n (the elements beyond n are "greater than all"; the caller has checked that the LDS behind keys[] holds N).
// The place goes where the key has room for it: END: lo = place : index; BEST: lo = {rb, qb} : place (16 bits) : index (16 bits).
template <bool BEST> DEVFN void ddp_bitonic(DdKey *keys, int n, int N, i32 *out, int lane)
{
	for (int x = lane; x < N; x += 64) {
		DdKey k; k.hi = ~0ull; k.lo = ~0ull;
		if (x < n) { k = keys[x]; k.lo = BEST ? (k.lo & 0xffffffff00000000ull) | (u64)(u32)x << 16 | (k.lo & 0xffffu) : (u64)(u32)x << 32 | (k.lo & 0xffffffffu); }
		keys[x] = k;
	}
	wave_sync();
	for (int k = 2; k <= N; k <<= 1)
		for (int j = k >> 1; j > 0; j >>= 1) {
			for (int t = lane; t < (N >> 1); t += 64) {
				const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;      // the t-th pair of this step: i has bit j clear
				const DdKey a = keys[i], b = keys[p];
				const bool a_gt_b = a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo);        // (lo as a whole: its low bits are the place, then the index)
				if (((i & k) == 0) == a_gt_b) { keys[i] = b; keys[p] = a; }
			}
			wave_sync();
		}
	for (int x = lane; x < n; x += 64) out[x] = (int)(BEST ? (u32)keys[x].lo & 0xffffu : (u32)keys[x].lo);
}

// mem_patch_reg's tests ahead of its alignment (bwamem.c:436-445), for a lane's own pair
DEVFN bool ddp_patch_may(const DevIndex &ix, const bwagpu_opt_t &opt, const DdHot &a, const DdHot &b)
{
	if (a.rb < ix.l_pac && b.rb >= ix.l_pac) return false;
	if (a.qb >= b.qb || a.qe >= b.qe || a.re >= b.re) return false;
	int w = (int)((a.re - b.rb) - (a.qe - b.qb)); if (w < 0) w = -w;
	double r = (double)(a.re - b.rb) / (b.re - a.rb) - (double)(a.qe - b.qb) / (b.qe - a.qb); if (r < 0.) r = -r;
	if (a.re < b.rb || a.qe < b.qb) { if (w > opt.w << 1 || r >= 0.05f) return false; }
	else if (w > opt.w << 2 || r >= 0.05f * 2) return false;
	return true;
}

// Returns false, with nothing changed, for a read whose coordinates do not fit the sort keys (the caller then takes the in-place routine).
template <bool BLK> __device__ bool dedup_read_par(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, const DedupLds &L, u64 &calls, u64 &cells)
{
	const int lane = threadIdx.x & 63;
	const int n = uni(B.reg_n_raw[r]);                   // (the caller checked n <= L.par_cap)
	bwagpu_alnreg_t *const ga = B.regs + uni64(B.reg_off[r]);
	const u8 *query = B.seq + uni64(B.off[r]);
	if (B.regs_raw) { for (int i = lane; i < n; i += 64) B.regs_raw[B.reg_off[r] + i] = ga[i]; }
	if (n <= 1) {
		if (lane == 0) {
			if (n == 1 && ga[0].rid >= 0 && ix.ctg_alt[ga[0].rid]) ga[0].is_alt = 1;
			B.reg_n[r] = n;
		}
		wave_sync();
		return true;
	}
	DdHot *hot = L.hot; DdKey *keys = L.keys; i32 *ord = L.ord, *ord2 = L.ord2;
	wave_sync();                                         // (the arrays' last readers: the read before)
	bool odd = false;                                    // a coordinate outside DdKey's fields (none is expected: 0 <= rb, re <= 2 l_pac < 2^48, 0 <= qb <= l_query)
	for (int i = lane; i < n; i += 64) {
		const bwagpu_alnreg_t &g = ga[i];
		DdHot h_; h_.rb = g.rb; h_.re = g.re; h_.qb = g.qb; h_.qe = g.qe; h_.rid = g.rid; h_.score = g.score;
		hot[i] = h_;
		keys[i] = ddp_key_end(h_, i);
		odd |= h_.rb < 0 || h_.rb >= ((i64)1 << 48) || h_.qb < 0 || h_.qb >= (1 << 16);
	}
	if (wave_ballot(odd)) { wave_sync(); return false; }   // ... then in place, with the full-width keys
	for (int i = lane; i < n; i += 64) ga[i].n_comp = 1;   // bwamem.c:468
	wave_sync();
	// ---- by end position (bwamem.c:467)
	if (lane == 0) dev_introsort<DdKey, DdKeyLessEnd, false>(keys, n, DdKeyLessEnd());
	wave_sync();
	const int cap_n = L.par_cap + L.par_cap / 2;          // elements of 16 bytes the LDS from keys[] on holds (keys, ord, ord2)
	{
		int N = 16; while (N < n) N <<= 1;
		if (B.dd_net > 0 && n >= B.dd_net && N <= cap_n) ddp_bitonic<false>(keys, n, N, ord, lane); else ddp_rank<false>(keys, n, ord, lane);
	}
	wave_sync();
	// ---- the redundancy scan (bwamem.c:470-497)
	const float mlr = opt.mask_level_redun; const int gap = opt.max_chain_gap;
	for (int i = 1; i < n; ++i) {
		const int pi = uni(ord[i]);
		DdHot p = hot[pi];
		p.rb = uni64(p.rb); p.re = uni64(p.re); p.qb = uni(p.qb); p.qe = uni(p.qe); p.rid = uni(p.rid); p.score = uni(p.score);
		bool p_dead = false;
		for (int jb = i - 1; jb >= 0 && !p_dead; jb -= 64) {
			int start = 0; bool more = false;
			for (;;) {                                   // (again from `start` after a patch alignment changed p)
				const int j = jb - lane; const bool valid = j >= 0;
				const int qi = valid ? ord[j] : 0;
				const DdHot q = hot[qi];
				const bool inwin = valid && q.rid == p.rid && p.rb < q.re + gap;
				const u64 out_m = wave_ballot(!inwin);
				const int nwin = out_m ? (int)__builtin_ctzll(out_m) : 64;           // the scan stops at the first region out of reach
				const bool act = lane >= start && lane < nwin && q.qe != q.qb;
				const i64 orr = q.re - p.rb;
				const i64 oq = q.qb < p.qb ? q.qe - p.qb : p.qe - q.qb;
				const i64 mr = q.re - q.rb < p.re - p.rb ? q.re - q.rb : p.re - p.rb;
				const i64 mq = q.qe - q.qb < p.qe - p.qb ? q.qe - q.qb : p.qe - p.qb;
				const bool red = orr > mlr * mr && oq > mlr * mq;
				const bool lose = act && red && p.score < q.score;
				const bool cand = act && !red && q.rb < p.rb && ddp_patch_may(ix, opt, q, p);
				const u64 lose_m = wave_ballot(lose), ev = lose_m | wave_ballot(cand);
				const int first = ev ? (int)__builtin_ctzll(ev) : nwin;
				if (act && red && lane < first) hot[qi].qe = q.qb;                 // the redundant regions with the lower score, up to the event
				wave_sync();
				if (!ev) { more = nwin == 64; break; }
				if ((lose_m >> first) & 1) {                                       // p is the redundant one: the scan ends
					if (lane == 0) hot[pi].qe = p.qb;
					p_dead = true;
					wave_sync();
					break;
				}
				const int jf = __builtin_amdgcn_readlane(qi, first);
				const bwagpu_alnreg_t qf = uni_reg(&ga[jf]), pf = uni_reg(&ga[pi]);   // (the records' rb, qb, score, w are kept up to date below)
				int w = 0;
				const int score = wave_patch_reg<BLK>(ix, opt, query, qf, pf, &w, L, calls, cells);
				if (score > 0) {
					if (lane == 0) {
						bwagpu_alnreg_t &gp = ga[pi]; const bwagpu_alnreg_t &gq = ga[jf];
						gp.n_comp += gq.n_comp + 1;
						if (gq.seedcov > gp.seedcov) gp.seedcov = gq.seedcov;
						if (gq.sub > gp.sub) gp.sub = gq.sub;
						if (gq.csub > gp.csub) gp.csub = gq.csub;
						gp.qb = gq.qb; gp.rb = gq.rb;
						gp.truesc = gp.score = score;
						gp.w = w;
						hot[pi].qb = qf.qb; hot[pi].rb = qf.rb; hot[pi].score = score;
						hot[jf].qb = qf.qe;
					}
					p.qb = qf.qb; p.rb = qf.rb; p.score = score;
					wave_sync();
				}
				start = first + 1;
			}
			if (!more) break;
		}
	}
	wave_sync();
	// ---- the regions left (bwamem.c:498-502), by score (bwamem.c:504)
	int m = 0;
	for (int x0 = 0; x0 < n; x0 += 64) {
		const int x = x0 + lane;
		const int id = x < n ? ord[x] : 0;
		const DdHot h_ = hot[id];
		const bool keep = x < n && h_.qe > h_.qb;
		const u64 km = wave_ballot(keep);
		if (keep) {
			keys[m + __popcll(km & (((u64)1 << lane) - 1))] = ddp_key_best(h_, id);
		}
		m += __popcll(km);
	}
	wave_sync();
	if (lane == 0) dev_introsort<DdKey, DdKeyLessBest, false>(keys, m, DdKeyLessBest());
	wave_sync();
	{
		int N = 16; while (N < m) N <<= 1;
		if (B.dd_net > 0 && m >= B.dd_net && N <= cap_n && m < 65536) ddp_bitonic<true>(keys, m, N, ord, lane); else ddp_rank<true>(keys, m, ord, lane);
	}
	wave_sync();
	// ---- identical hits (bwamem.c:505-513): every region that equals the one before it goes
	int nf = 0;
	for (int x0 = 0; x0 < m; x0 += 64) {
		const int x = x0 + lane;
		const int id = x < m ? ord[x] : 0, idp = x > 0 && x < m ? ord[x - 1] : 0;
		const DdHot h_ = hot[id], hp = hot[idp];
		const bool keep = x < m && (x == 0 || !(h_.score == hp.score && h_.rb == hp.rb && h_.qb == hp.qb));
		const u64 km = wave_ballot(keep);
		if (keep) ord2[nf + __popcll(km & (((u64)1 << lane) - 1))] = id;
		nf += __popcll(km);
	}
	wave_sync();
	// ---- the records in their final order: gathered into the staging area, flagged (bwamem.c:1111-1115), and back
	u32 *tw = (u32*)L.tmp; const u32 *gw = (const u32*)ga;
	for (int x = lane; x < nf * DDP_REG_WORDS; x += 64) {
		const int k = x / DDP_REG_WORDS, d = x - k * DDP_REG_WORDS;
		tw[x] = gw[(size_t)ord2[k] * DDP_REG_WORDS + d];
	}
	wave_sync();
	for (int k = lane; k < nf; k += 64) {
		const int rid = hot[ord2[k]].rid;
		if (rid >= 0 && ix.ctg_alt[rid]) L.tmp[k].is_alt = 1;
	}
	wave_sync();
	{
		u32 *ow = (u32*)ga;
		for (int x = lane; x < nf * DDP_REG_WORDS; x += 64) ow[x] = tw[x];
	}
	if (lane == 0) B.reg_n[r] = nf;
	wave_sync();
	return true;
}
