// dev_seed.h -- SMEM seeding (mem_collect_intv, bwamem.c:140-188) and the SA-lookup kernel (bwt_sa).
//
// One lane per read.  The search is a chain of dependent 64-byte index reads (~1000 per 150 bp read), so
// throughput comes from having several hundred thousand independent chains in flight, not from
// parallelising one chain; each lane keeps its bi-interval in registers and fetches a whole Occ block
// (4 x dwordx4) per rank query.
#pragma once
#include "dev_fm.h"
#include "dev_sort.h"

struct SeedEmit {        // the MEM list of the read being seeded (lane-private scratch)
	Intv3 *mem; int n, cap; bool overflow;
	int min_seed_len;
	DEVFN void add(u64 x0, u64 x2, int start, int end) {
		if (end - start < min_seed_len) return;
		if (n == cap) { overflow = true; return; }
		Intv3 v; v.x0 = x0; v.x2 = x2; v.info = (u64)start << 32 | (u32)end;
		mem[n++] = v;
	}
};

// bwt_smem1a with max_intv = 0 (bwt.c:289-351).  s0/s1: two lane-private stacks of `cap` entries.
// Forward sweep: remember the interval each time its size is about to change (stored top-down in s0 so that the
// longest match comes first without a reversal).  Backward sweep: extend every surviving interval by q[i];
// an interval that can no longer be extended is a MEM iff nothing longer survived this step and it is not
// contained in the previously emitted MEM.
__device__ int dev_smem1(const DevIndex &ix, const u8 *q, int len, int x, u64 min_intv, BiIntv *s0, BiIntv *s1, int cap,
						 SeedEmit &em, u32 &nblk)
{
	if (q[x] > 3) return x + 1;
	if (min_intv < 1) min_intv = 1;
	BiIntv ik; fm_init(ix, q[x], ik); ik.info = (u64)(x + 1);
	int n0 = 0, i;
	for (i = x + 1; i < len; ++i) {
		int b = q[i];
		if (b < 4) {
			BiIntv ok; nblk += fm_extend1(ix, ik, 3 - b, 0, ok);
			if (ok.x2 != ik.x2) {
				s0[cap - 1 - n0] = ik; ++n0;
				if (ok.x2 < min_intv) break;
			}
			ok.info = (u64)(i + 1); ik = ok;
		} else { s0[cap - 1 - n0] = ik; ++n0; break; }
	}
	if (i == len) { s0[cap - 1 - n0] = ik; ++n0; }
	BiIntv *prev = s0 + (cap - n0), *curr = s1;
	int nprev = n0, ret = (int)prev[0].info;
	bool any = false; int last_start = 0;
	for (i = x - 1; i >= -1; --i) {
		int c = i < 0 ? -1 : (q[i] < 4 ? (int)q[i] : -1);
		int nc = 0; u64 last_x2 = 0;
		for (int j = 0; j < nprev; ++j) {
			BiIntv p = prev[j], ok; ok.x0 = ok.x1 = ok.x2 = 0;
			if (c >= 0) nblk += fm_extend1(ix, p, c, 1, ok);
			if (c < 0 || ok.x2 < min_intv) {
				if (nc == 0 && (!any || i + 1 < last_start)) {
					em.add(p.x0, p.x2, i + 1, (int)p.info);
					any = true; last_start = i + 1;
				}
			} else if (nc == 0 || ok.x2 != last_x2) {
				ok.info = p.info; curr[nc++] = ok; last_x2 = ok.x2;
			}
		}
		if (nc == 0) break;
		prev = curr; nprev = nc; curr = (curr == s1) ? s0 : s1;
	}
	return ret;
}

// bwt_seed_strategy1 (bwt.c:358-379)
__device__ int dev_seed_strategy1(const DevIndex &ix, const u8 *q, int len, int x, int min_len, u64 max_intv, SeedEmit &em, u32 &nblk)
{
	if (q[x] > 3) return x + 1;
	BiIntv ik; fm_init(ix, q[x], ik);
	for (int i = x + 1; i < len; ++i) {
		int b = q[i];
		if (b > 3) return i + 1;
		BiIntv ok; nblk += fm_extend1(ix, ik, 3 - b, 0, ok);
		if (ok.x2 < max_intv && i - x >= min_len) {
			if (ok.x2 > 0) em.add(ok.x0, ok.x2, x, i + 1); // caller keeps it only if x[2] > 0 (bwamem.c:177)
			return i + 1;
		}
		ik = ok;
	}
	return len;
}

struct IntvInfoLess { DEVFN bool operator()(const Intv3 &a, const Intv3 &b) const { return a.info < b.info; } };

#define BT_NODE_INTS 40   // n, internal, 9 chain indices, 10 children, (pad), 9 x i64 positions = 160 bytes

// The seeding pass for one read + reservation of everything later stages need for it.
__device__ void seed_read(const DevIndex &ix, const bwagpu_opt_t &opt, const Batch &B, int r, int tslot, u32 &nblk)
{
	const u8 *q = B.seq + B.off[r];
	int len = (int)(B.off[r + 1] - B.off[r]);
	B.intv_n[r] = 0; B.intv_off[r] = 0; B.seed_n[r] = 0; B.seed_off[r] = 0; B.node_off[r] = 0;
	if (len < opt.min_seed_len) return;   // mem_chain (bwamem.c:286)
	int cap = B.max_len + 1;
	BiIntv *s0 = B.tmp_intv + (size_t)tslot * 2 * cap, *s1 = s0 + cap;
	SeedEmit em; em.mem = B.tmp_mem + (size_t)tslot * B.mem_cap; em.n = 0; em.cap = B.mem_cap; em.overflow = false;
	em.min_seed_len = opt.min_seed_len;
	int split_len = (int)(opt.min_seed_len * opt.split_factor + .499);
	// pass 1: all SMEMs
	int x = 0;
	while (x < len) {
		if (q[x] < 4) x = dev_smem1(ix, q, len, x, 1, s0, s1, cap, em, nblk);
		else ++x;
	}
	// pass 2: re-seed from the middle of long, rare SMEMs
	int old_n = em.n;
	for (int k = 0; k < old_n; ++k) {
		Intv3 p = em.mem[k];
		int start = (int)(p.info >> 32), end = (int)(u32)p.info;
		if (end - start < split_len || p.x2 > (u64)opt.split_width) continue;
		dev_smem1(ix, q, len, (start + end) >> 1, p.x2 + 1, s0, s1, cap, em, nblk);
	}
	// pass 3: LAST-like seeds
	if (opt.max_mem_intv > 0) {
		x = 0;
		while (x < len) {
			if (q[x] < 4) x = dev_seed_strategy1(ix, q, len, x, opt.min_seed_len, opt.max_mem_intv, em, nblk);
			else ++x;
		}
	}
	if (em.overflow) { atomicOr(&B.ctr->overflow, 16ull); return; }
	// sort by (start, end); equal keys are identical intervals, so the sort need not mimic ks_introsort's ties
	dev_introsort(em.mem, em.n, IntvInfoLess());
	// publish the intervals
	int n = em.n;
	if (n == 0) return;
	u64 ioff = atomicAdd(&B.ctr->intv_used, (unsigned long long)n);
	if (ioff + n > (u64)B.intv_cap) { atomicOr(&B.ctr->overflow, 1ull); return; }
	// number of SA lookups (mem_chain's inner loop bounds, bwamem.c:304-305)
	i64 ns = 0;
	for (int i = 0; i < n; ++i) {
		Intv3 p = em.mem[i];
		B.intv[ioff + i] = p;
		u64 step = p.x2 > (u64)opt.max_occ ? p.x2 / opt.max_occ : 1;
		u64 cnt = (p.x2 + step - 1) / step;
		ns += (i64)(cnt < (u64)opt.max_occ ? cnt : (u64)opt.max_occ);
	}
	B.intv_n[r] = n; B.intv_off[r] = (i64)ioff;
	u64 soff = atomicAdd(&B.ctr->seed_used, (unsigned long long)ns);
	u64 nnode = (u64)ns / 4 + 2;
	u64 noff = atomicAdd(&B.ctr->node_used, (unsigned long long)nnode);
	if (soff + ns > (u64)B.slot_cap) { atomicOr(&B.ctr->overflow, 2ull); return; }
	if (noff + nnode > (u64)B.node_cap) { atomicOr(&B.ctr->overflow, 4ull); return; }
	i64 s = 0;
	for (int i = 0; i < n; ++i) {
		Intv3 p = em.mem[i];
		int step = p.x2 > (u64)opt.max_occ ? (int)(p.x2 / opt.max_occ) : 1;
		int count = 0;
		for (i64 k = 0; (u64)k < p.x2 && count < opt.max_occ; k += step, ++count, ++s) {
			B.slot_pos[soff + s] = p.x0 + (u64)k;
			B.slot_qbeg[soff + s] = (i32)(p.info >> 32);
			B.slot_len[soff + s] = (i32)((u32)p.info - (u32)(p.info >> 32));
		}
	}
	B.seed_n[r] = (i32)ns; B.seed_off[r] = (i64)soff; B.node_off[r] = (i64)noff;
}

__global__ void __launch_bounds__(256, 4) k_seed(DevIndex ix, bwagpu_opt_t opt, Batch B)
{
	int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
	u32 nblk = 0; u64 nintv = 0;
	for (int r = tid; r < B.n_reads; r += nth) {
		seed_read(ix, opt, B, r, tid, nblk);
		nintv += B.intv_n[r];
	}
	if (B.stats) {
		atomicAdd(&B.ctr->occ_blocks, (unsigned long long)nblk);
		atomicAdd(&B.ctr->n_intv, (unsigned long long)nintv);
	}
}

// One lane per SA lookup: slot_pos[s] (an SA row) -> reference position.  ~31 dependent block reads each with
// the reference's sa_intv = 32; a single read when the SA has been densified.
__global__ void __launch_bounds__(256) k_sa(DevIndex ix, Batch B)
{
	u64 n = B.ctr->seed_used;
	if (n > (u64)B.slot_cap) n = (u64)B.slot_cap;
	u32 steps = 0;
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += (u64)gridDim.x * blockDim.x) {
		u64 rbeg = fm_sa(ix, B.slot_pos[s], &steps);
		B.slot_pos[s] = rbeg;
		B.slot_rid[s] = dev_intv2rid(ix, (i64)rbeg, (i64)rbeg + B.slot_len[s]);   // bns_intv2rid of mem_chain (bwamem.c:312)
	}
	if (B.stats) atomicAdd(&B.ctr->lf_steps, (unsigned long long)steps);
}
