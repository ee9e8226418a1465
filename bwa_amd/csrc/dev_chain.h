// dev_chain.h -- what the chaining stage (mem_chain, bwamem.c:277-342; mem_chain_flt, bwamem.c:353-411; kernels in dev_chainw.h) shares with
// the rest: B-tree geometry, the weight order, and the heavy-first ordering of reads used by the wave-per-read kernels.
#pragma once
#include "dev_fm.h"
#include "dev_sort.h"
#include "dev_seed.h"

// B-tree geometry of kbtree.h as instantiated at bwamem.c:212-213 (t = 5, at most 9 keys per node); the wave-per-read chaining kernel
// (dev_chainw.h) reproduces the structure literally.  A node is a 160-byte record {n, internal, chain index[9],
// child[10], position[9]}.  (The round-1 lane-per-read kernel that lived here is gone: the wave kernel's three tiers take every read.)
#define BT_T 5
#define BT_MAXK 9

struct ChainWGreater {   // flt_lt (bwamem.c:350): heavier chains first; elements are {weight, chain index} pairs so that the
	DEVFN bool operator()(const int2 &a, const int2 &b) const { return a.x > b.x; }   // sort touches no other memory
};

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
