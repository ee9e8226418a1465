// bwagpu_index.hip -- FM-index construction on the device: the BWT/SA half of `bwa index`.
//
// What it replaces in the reference: bwt_bwtgen2 / bwt_pac2bwt (bwtindex.c:64-120, bwt_gen.c: BWT of the forward +
// reverse-complement text by incremental block merging, ~0.5 s/Mbp on one core), bwt_bwtupdate_core (bwtindex.c:150-172:
// interleaving of the Occ checkpoints) and bwt_cal_sa (bwt.c:62-84: the sampled suffix array by an LF walk over the whole
// text).  The outputs are the very arrays those routines leave in bwt_t (bwt.h:48-60), so that bwt_dump_bwt / bwt_dump_sa
// (bwt.c:385-407) -- or our writer -- produce byte-identical .bwt/.sa files (tests/test_index_build.py, -m gpu tests).
//
// Method (not the reference's): the suffix array of T = forward + reverse complement is built outright, in HBM.
//   1. T is packed to 2 bits per base, 32 bases per big-endian u64, so that any 32-mer is two loads and a funnel shift.
//   2. Suffixes are partitioned by their first B bases into 4^B buckets (B chosen so that a bucket is ~10^8 suffixes),
//      and each bucket is radix-sorted (rocPRIM, 64-bit keys) by its next 29 bases + the suffix's valid length (the
//      terminator sorts below A).  After this one pass every suffix whose first B+29 bases are unique -- all but the
//      repeats of the genome -- has its final row.
//   3. The remaining groups are refined by prefix doubling restricted to the unsorted suffixes (Larsson-Sadakane style
//      discarding): a suffix's key is (its group, rank of the suffix h bases further on); one radix sort of the compacted
//      active list per round, h = K, 2K, 4K, ... until no group is left.
//   4. BWT symbols, Occ checkpoints every 128 symbols and the sampled SA are derived from the full SA by streaming kernels.
// Everything is integer work bounded by HBM bandwidth (the sorts) and random 8-byte reads (rank look-ups); no MFMA.
// GRCh38 scale (seq_len 6.2e9): ~110 GB of HBM for SA + rank, which is why this is a 288 GB-per-GPU design.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/functional.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <string>
#include <vector>
#include "../../include/bwagpu.h"

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef unsigned char u8;

#define IDX_DEVFN __device__ __forceinline__
#define IDX_BLOCK 256
#define IDX_KEY_BASES 29          // bases of a first-pass sort key (58 bits) + 6 bits of valid length
#define IDX_RANK_BITS 34          // rows < 2^34: l_pac up to 8.5 Gbp

namespace {

struct Buf {
	void *p = nullptr; size_t cap = 0;
	int ensure(size_t bytes) {
		if (bytes <= cap) return 0;
		release();
		if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) { p = nullptr; return -1; }
		cap = bytes;
		return 0;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T*)p; }
	~Buf() { release(); }
};

// base p of the forward strand from the reference's .pac (bntseq.c:229-230: base l in byte l>>2 at bits (3-(l&3))*2)
IDX_DEVFN int pac_get(const u8 *pac, u64 p) { return pac[p >> 2] >> ((~p & 3) << 1) & 3; }

// 32 bases of T starting at i, first base in the top two bits, 'A' (0) past the end; tw is padded by two zero words
IDX_DEVFN u64 text32(const u64 *tw, u64 i)
{
	const u64 w = i >> 5; const int o = (int)(i & 31) << 1;
	const u64 a = tw[w], b = tw[w + 1];
	return o ? a << o | b >> (64 - o) : a;
}
IDX_DEVFN int text1(const u64 *tw, u64 i) { return (int)(tw[i >> 5] >> ((~i & 31) << 1)) & 3; }

// T = forward strand followed by its reverse complement (bwtindex.c:305-311 / bns_fasta2bntseq with for_only = 0, bntseq.c:299-311)
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_text(const u8 *pac, u64 l_pac, u64 *tw, u64 n_words_pad)
{
	const u64 n = l_pac << 1;
	for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words_pad; w += (u64)gridDim.x * blockDim.x) {
		u64 acc = 0;
		for (int j = 0; j < 32; ++j) {
			const u64 p = (w << 5) + (u64)j;
			int c = 0;
			if (p < l_pac) c = pac_get(pac, p);
			else if (p < n) c = 3 - pac_get(pac, n - 1 - p);
			acc = acc << 2 | (u64)c;
		}
		tw[w] = acc;
	}
}

// ---- first pass: buckets by the first B bases ------------------------------------------------------------------------------
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_bucket_count(const u64 *tw, u64 n, int B, u64 *hist)
{
	__shared__ u32 lh[4096];
	const int nb = 1 << (2 * B);
	for (int k = threadIdx.x; k < nb; k += blockDim.x) lh[k] = 0;
	__syncthreads();
	const u64 per_block = (u64)IDX_BLOCK * 64;
	for (u64 base = (u64)blockIdx.x * per_block; base < n; base += (u64)gridDim.x * per_block) {
		for (int k = 0; k < 64; ++k) {
			const u64 i = base + (u64)k * IDX_BLOCK + threadIdx.x;
			if (i < n) atomicAdd(&lh[B ? (u32)(text32(tw, i) >> (64 - 2 * B)) : 0u], 1u);
		}
	}
	__syncthreads();
	for (int k = threadIdx.x; k < nb; k += blockDim.x) if (lh[k]) atomicAdd(&hist[k], (u64)lh[k]);
}

// positions of every bucket, in arbitrary order inside the bucket (the sort that follows orders them), written straight into
// the bucket's row range of the SA: rows[cursor[b] ...)
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_bucket_scatter(const u64 *tw, u64 n, int B, u64 *cursor, u64 *rows)
{
	__shared__ u32 lh[4096];
	__shared__ u64 lbase[4096];
	const int nb = 1 << (2 * B);
	const u64 per_block = (u64)IDX_BLOCK * 16;
	for (u64 base = (u64)blockIdx.x * per_block; base < n; base += (u64)gridDim.x * per_block) {
		for (int k = threadIdx.x; k < nb; k += blockDim.x) lh[k] = 0;
		__syncthreads();
		u32 bk[16], rk[16];
		for (int k = 0; k < 16; ++k) {
			const u64 i = base + (u64)k * IDX_BLOCK + threadIdx.x;
			bk[k] = ~0u; rk[k] = 0;
			if (i < n) { bk[k] = B ? (u32)(text32(tw, i) >> (64 - 2 * B)) : 0u; rk[k] = atomicAdd(&lh[bk[k]], 1u); }
		}
		__syncthreads();
		for (int k = threadIdx.x; k < nb; k += blockDim.x) lbase[k] = lh[k] ? atomicAdd(&cursor[k], (u64)lh[k]) : 0;
		__syncthreads();
		for (int k = 0; k < 16; ++k) {
			const u64 i = base + (u64)k * IDX_BLOCK + threadIdx.x;
			if (bk[k] != ~0u) rows[lbase[bk[k]] + rk[k]] = i;
		}
		__syncthreads();
	}
}

// sort key of suffix i inside its bucket: the 29 bases after the bucket prefix, then the suffix's valid length capped at
// K = B + 29.  Padding bases past the end read as A; among equal padded prefixes the shorter suffix is the smaller one
// (its terminator sorts below A), which is what the length field encodes.  Suffixes with v < K get a unique key.
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_keys(const u64 *tw, u64 n, int B, const u64 *pos, u64 m, u64 *keys)
{
	const u64 K = (u64)B + IDX_KEY_BASES;
	for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (u64)gridDim.x * blockDim.x) {
		const u64 i = pos[q];
		const u64 v = n - i < K ? n - i : K;
		keys[q] = (text32(tw, i + (u64)B) & ~63ull) | v;
	}
}

// hd[q] = q + 1 at the first element of every run of equal keys, else 0 (an inclusive max-scan turns it into "my run's head + 1")
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_heads(const u64 *keys, u64 m, u64 *hd)
{
	for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (u64)gridDim.x * blockDim.x)
		hd[q] = (q == 0 || keys[q] != keys[q - 1]) ? q + 1 : 0;
}

// After a bucket's sort: rows and ranks of its suffixes; af[q] = 1 for suffixes whose group has more than one member
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_bucket_finish(const u64 *vals, const u64 *hs, u64 m, u64 row0, u64 *sa, u64 *rank, u64 *af)
{
	for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (u64)gridDim.x * blockDim.x) {
		const u64 i = vals[q], h = hs[q];
		sa[row0 + q] = i;
		rank[i] = row0 + h - 1;
		const bool head = h == q + 1, next_head = q + 1 == m || hs[q + 1] == q + 2;
		af[q] = (head && next_head) ? 0 : 1;
	}
}

__global__ void __launch_bounds__(IDX_BLOCK) k_idx_bucket_compact(const u64 *vals, const u64 *hs, const u64 *af, const u64 *pos, u64 m, u64 row0,
																  u64 *act_i, u64 *act_head, u64 out0)
{
	for (u64 q = (u64)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += (u64)gridDim.x * blockDim.x)
		if (af[q]) { act_i[out0 + pos[q]] = vals[q]; act_head[out0 + pos[q]] = row0 + hs[q] - 1; }
}

// ---- prefix doubling over the unsorted suffixes ---------------------------------------------------------------------------
// active list -> group numbers: gf[a] = 1 where a new group starts
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_group_flags(const u64 *act_head, u64 A, u64 *gf)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x)
		gf[a] = (a == 0 || act_head[a] != act_head[a - 1]) ? 1 : 0;
}
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_group_init(const u64 *act_head, const u64 *gsum, u64 A, u64 *gid, u64 *ghead, u64 *gstart)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		const u64 g = gsum[a] - 1;
		gid[a] = g;
		if (a == 0 || act_head[a] != act_head[a - 1]) { ghead[g] = act_head[a]; gstart[g] = a; }
	}
}

// key of one active suffix in the round with step h: (group, rank of the suffix h bases on); the terminator's rank is 0
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_round_keys(const u64 *act_i, const u64 *gid, const u64 *rank, u64 A, u64 n, u64 h, u64 *keys)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		const u64 j = act_i[a] + h;
		keys[a] = gid[a] << IDX_RANK_BITS | (j < n ? rank[j] : 0ull);
	}
}
// fallback when the group number does not fit beside the rank: two stable sorts, by rank first, then by group
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_round_rank_keys(const u64 *act_i, const u64 *rank, u64 A, u64 n, u64 h, u64 *keys)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		const u64 j = act_i[a] + h;
		keys[a] = j < n ? rank[j] : 0ull;
	}
}

// After the round's sort (elements of one group are still contiguous and in the group's old slot range):
//   sf[a] - 1 = index of the first element with the same (group, rank) -> the element's new group head
// writes the new rows and ranks, and flags survivors (members of groups that still have > 1 element) and new group heads.
// `split` form (two-sort fallback): k2 holds the rank keys, gsorted the group numbers; otherwise the group is keys >> 34.
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_round_flags(const u64 *keys, const u64 *gsorted, u64 A, u64 *fk)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		bool head = a == 0 || keys[a] != keys[a - 1];
		if (gsorted && !head) head = gsorted[a] != gsorted[a - 1];
		fk[a] = head ? a + 1 : 0;
	}
}
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_round_apply(const u64 *keys, const u64 *gsorted, const u64 *vals, const u64 *sf, u64 A, const u64 *ghead, const u64 *gstart,
															   u64 *sa, u64 *rank, u64 *surv, u64 *nhead)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		const u64 g = gsorted ? gsorted[a] : keys[a] >> IDX_RANK_BITS;
		const u64 r0 = ghead[g], a0 = gstart[g], i = vals[a];
		sa[r0 + (a - a0)] = i;
		rank[i] = r0 + (sf[a] - 1 - a0);
		const bool head = sf[a] == a + 1, next_head = a + 1 == A || sf[a + 1] == a + 2;
		const u64 s = (head && next_head) ? 0 : 1;
		surv[a] = s; nhead[a] = (s && head) ? 1 : 0;
	}
}
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_round_compact(const u64 *keys, const u64 *gsorted, const u64 *vals, const u64 *sf, const u64 *surv, const u64 *pos, const u64 *ngsum, u64 A,
																 const u64 *ghead, const u64 *gstart, u64 *act_i2, u64 *gid2, u64 *ghead2, u64 *gstart2)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) {
		if (!surv[a]) continue;
		const u64 np = pos[a], ng = ngsum[a] - 1;
		act_i2[np] = vals[a]; gid2[np] = ng;
		if (sf[a] == a + 1) {
			const u64 g = gsorted ? gsorted[a] : keys[a] >> IDX_RANK_BITS;
			ghead2[ng] = ghead[g] + (a - gstart[g]); gstart2[ng] = np;
		}
	}
}

__global__ void __launch_bounds__(IDX_BLOCK) k_idx_iota(u64 *out, u64 A)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) out[a] = a;
}
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_gather(const u64 *src, const u64 *idx, u64 A, u64 *out)
{
	for (u64 a = (u64)blockIdx.x * blockDim.x + threadIdx.x; a < A; a += (u64)gridDim.x * blockDim.x) out[a] = src[idx[a]];
}

// ---- BWT, Occ checkpoints, sampled SA (bwtindex.c:150-172, bwt.c:62-84) --------------------------------------------------------
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_find_primary(const u64 *sa, u64 n_rows, u64 *primary)
{
	for (u64 r = (u64)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (u64)gridDim.x * blockDim.x)
		if (sa[r] == 0) *primary = r;
}
// word wi of the $-less BWT string: symbols x = 16 wi .. 16 wi + 15, symbol x = T[SA[row] - 1] with row = x + (x >= primary);
// symbol j of a word sits at bits (15 - j) * 2 (bwt.h:74-80)
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_bwt_words(const u64 *sa, const u64 *tw, u64 n, u64 primary, u32 *words, u64 n_words_pad)
{
	for (u64 wi = (u64)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words_pad; wi += (u64)gridDim.x * blockDim.x) {
		u32 acc = 0;
		for (int j = 0; j < 16; ++j) {
			const u64 x = (wi << 4) + (u64)j;
			u32 c = 0;
			if (x < n) { const u64 row = x + (x >= primary ? 1 : 0); c = (u32)text1(tw, sa[row] - 1); }
			acc = acc << 2 | c;
		}
		words[wi] = acc;
	}
}
// per 128-symbol block: number of A, C, G, T (the last block counts only the symbols that exist)
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_block_counts(const u32 *words, u64 n, u64 n_blk, u64 *c0, u64 *c1, u64 *c2, u64 *c3)
{
	for (u64 b = (u64)blockIdx.x * blockDim.x + threadIdx.x; b < n_blk; b += (u64)gridDim.x * blockDim.x) {
		u32 k1 = 0, k2 = 0, k3 = 0;
		for (int k = 0; k < 8; ++k) {
			const u32 w = words[b * 8 + k], lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
			k3 += __popc(hi & lo); k2 += __popc(hi & ~lo); k1 += __popc(lo & ~hi);
		}
		const u64 len = n - b * 128 < 128 ? n - b * 128 : 128;      // padding symbols are 0 and were never counted as 1..3
		c0[b] = len - k1 - k2 - k3; c1[b] = k1; c2[b] = k2; c3[b] = k3;
	}
}
// the reference's interleaved layout: per block 4 x u64 counts before the block + 8 x u32 symbols; one trailing counts record
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_interleave(const u32 *words, const u64 *o0, const u64 *o1, const u64 *o2, const u64 *o3, u64 n_blk, u32 *out)
{
	for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n_blk * 16; t += (u64)gridDim.x * blockDim.x) {
		const u64 b = t >> 4; const int k = (int)(t & 15);
		u32 v;
		if (k < 8) { const u64 c = (k >> 1) == 0 ? o0[b] : (k >> 1) == 1 ? o1[b] : (k >> 1) == 2 ? o2[b] : o3[b]; v = (k & 1) ? (u32)(c >> 32) : (u32)c; }
		else v = words[b * 8 + (k - 8)];
		out[t] = v;
	}
}
__global__ void __launch_bounds__(IDX_BLOCK) k_idx_sample_sa(const u64 *sa, u64 n_sa, int intv, u64 *out)
{
	for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_sa; j += (u64)gridDim.x * blockDim.x)
		out[j] = j == 0 ? ~0ull : sa[j * (u64)intv];          // bwt_cal_sa leaves sa[0] = -1 (bwt.c:79)
}

static unsigned grid_for(u64 items, u64 per_thread = 1)
{
	u64 b = (items + (u64)IDX_BLOCK * per_thread - 1) / ((u64)IDX_BLOCK * per_thread);
	if (b < 1) b = 1;
	if (b > 65536) b = 65536;
	return (unsigned)b;
}

struct Builder {
	hipStream_t st = nullptr;
	std::string err;
	Buf tmp;   // rocPRIM temporary storage
	int verbose = 0;

	int fail(const char *what, hipError_t e) { err = std::string(what) + ": " + hipGetErrorString(e); return BWAGPU_EHIP; }
	int sort_pairs(u64 *k_in, u64 *k_out, u64 *v_in, u64 *v_out, u64 m, int begin_bit, int end_bit)
	{
		size_t bytes = 0;
		hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k_in, k_out, v_in, v_out, (size_t)m, (unsigned)begin_bit, (unsigned)end_bit, st);
		if (e != hipSuccess) return fail("radix_sort_pairs(size)", e);
		if (tmp.ensure(bytes)) { err = "hipMalloc failed (sort scratch)"; return BWAGPU_ENOMEM; }
		e = rocprim::radix_sort_pairs(tmp.p, bytes, k_in, k_out, v_in, v_out, (size_t)m, (unsigned)begin_bit, (unsigned)end_bit, st);
		if (e != hipSuccess) return fail("radix_sort_pairs", e);
		return 0;
	}
	int scan_max(u64 *in, u64 *out, u64 m)
	{
		size_t bytes = 0;
		hipError_t e = rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)m, rocprim::maximum<u64>(), st);
		if (e != hipSuccess) return fail("inclusive_scan(size)", e);
		if (tmp.ensure(bytes)) { err = "hipMalloc failed (scan scratch)"; return BWAGPU_ENOMEM; }
		e = rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)m, rocprim::maximum<u64>(), st);
		return e == hipSuccess ? 0 : fail("inclusive_scan", e);
	}
	int scan_sum_incl(u64 *in, u64 *out, u64 m)
	{
		size_t bytes = 0;
		hipError_t e = rocprim::inclusive_scan(nullptr, bytes, in, out, (size_t)m, rocprim::plus<u64>(), st);
		if (e != hipSuccess) return fail("inclusive_scan(size)", e);
		if (tmp.ensure(bytes)) { err = "hipMalloc failed (scan scratch)"; return BWAGPU_ENOMEM; }
		e = rocprim::inclusive_scan(tmp.p, bytes, in, out, (size_t)m, rocprim::plus<u64>(), st);
		return e == hipSuccess ? 0 : fail("inclusive_scan", e);
	}
	int scan_sum_excl(u64 *in, u64 *out, u64 m)
	{
		size_t bytes = 0;
		hipError_t e = rocprim::exclusive_scan(nullptr, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), st);
		if (e != hipSuccess) return fail("exclusive_scan(size)", e);
		if (tmp.ensure(bytes)) { err = "hipMalloc failed (scan scratch)"; return BWAGPU_ENOMEM; }
		e = rocprim::exclusive_scan(tmp.p, bytes, in, out, (u64)0, (size_t)m, rocprim::plus<u64>(), st);
		return e == hipSuccess ? 0 : fail("exclusive_scan", e);
	}
	// enlarge a device array that already holds `used` bytes (geometric growth, contents kept)
	int grow(Buf &b, size_t need, size_t used)
	{
		if (need <= b.cap) return 0;
		size_t want = need + need / 2 + 4096;
		void *np = nullptr;
		if (hipMalloc(&np, want) != hipSuccess) {
			want = need;
			if (hipMalloc(&np, want) != hipSuccess) { err = "hipMalloc failed (list of unsorted suffixes)"; return BWAGPU_ENOMEM; }
		}
		if (used) { hipError_t e = hipMemcpyAsync(np, b.p, used, hipMemcpyDeviceToDevice, st); if (e == hipSuccess) e = hipStreamSynchronize(st); if (e != hipSuccess) { (void)hipFree(np); return fail("grow", e); } }
		b.release(); b.p = np; b.cap = want;
		return 0;
	}
	// last element of an exclusive scan + its input = total
	int total_of(const u64 *excl, const u64 *in, u64 m, u64 *out)
	{
		u64 a = 0, b = 0;
		if (m) {
			hipError_t e = hipMemcpyAsync(&a, excl + (m - 1), 8, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess) e = hipMemcpyAsync(&b, in + (m - 1), 8, hipMemcpyDeviceToHost, st);
			if (e == hipSuccess) e = hipStreamSynchronize(st);
			if (e != hipSuccess) return fail("read back scan total", e);
		}
		*out = a + b;
		return 0;
	}
};

static int bits_for(u64 v) { int b = 0; while (b < 64 && (v >> b)) ++b; return b; }

#define IDXCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { rc = bl.fail(#call, e_); goto done; } } while (0)
#define IDXRC(call) do { rc = (call); if (rc) goto done; } while (0)

}  // namespace

extern "C" void bwagpu_built_free(bwagpu_built_t *b)
{
	if (!b) return;
	free(b->bwt); free(b->sa);
	b->bwt = nullptr; b->sa = nullptr;
}

extern "C" int bwagpu_index_build(const uint8_t *pac, int64_t l_pac_, int sa_intv, int device, bwagpu_built_t *out, char *errbuf, size_t errlen)
{
	if (errbuf && errlen) errbuf[0] = 0;
	if (!pac || !out || l_pac_ <= 0 || sa_intv <= 0 || (sa_intv & (sa_intv - 1))) return BWAGPU_EINVAL;
	if ((u64)l_pac_ >= (1ull << (IDX_RANK_BITS - 1)) - 64) return BWAGPU_EUNSUP;
	memset(out, 0, sizeof *out);
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BWAGPU_ENODEV;
	if (hipSetDevice(device) != hipSuccess) return BWAGPU_ENODEV;
	Builder bl;
	bl.verbose = getenv("BWAGPU_INDEX_VERBOSE") ? atoi(getenv("BWAGPU_INDEX_VERBOSE")) : 0;
	if (hipStreamCreate(&bl.st) != hipSuccess) return BWAGPU_ENODEV;
	int rc = 0;
	const u64 l_pac = (u64)l_pac_, n = l_pac << 1, n_rows = n + 1;
	const u64 n_tw = (n + 31) / 32 + 3;                   // + padding words so that text32 may read past the end
	// bucket prefix: ~2^27 suffixes per bucket (the sort's throughput is flat from ~10^7 elements; its buffers stay small)
	int B = 0;
	if (getenv("BWAGPU_INDEX_BUCKET_BASES")) B = atoi(getenv("BWAGPU_INDEX_BUCKET_BASES"));   // test hook
	else while (B < 6 && (n >> (2 * B)) > (1ull << 27)) ++B;
	if (B < 0) B = 0; if (B > 6) B = 6;
	const int n_bk = 1 << (2 * B);
	const u64 K = (u64)B + IDX_KEY_BASES;
	Buf d_pac, d_tw, d_sa, d_rank, d_hist, d_cursor, d_keys, d_keys2, d_vals2, d_s1, d_s2, d_s3, d_s4;
	Buf d_act_i, d_act_i2, d_gid, d_gid2, d_ghead, d_ghead2, d_gstart, d_gstart2, d_scalar;
	Buf d_words, d_c[4], d_o[4], d_out, d_sas, d_x1, d_x2, d_x3;
	std::vector<u64> hist((size_t)n_bk), base((size_t)n_bk + 1, 0);
	u64 max_bucket = 0, A = 0, primary = 0, n_groups = 0;
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	(void)hipEventCreate(&ev0); (void)hipEventCreate(&ev1);
	(void)hipEventRecord(ev0, bl.st);

	if (d_pac.ensure((size_t)(l_pac / 4 + 1)) || d_tw.ensure(n_tw * 8) || d_sa.ensure(n_rows * 8) || d_rank.ensure(n_rows * 8) ||
		d_hist.ensure((size_t)n_bk * 8) || d_cursor.ensure((size_t)n_bk * 8) || d_scalar.ensure(64)) { bl.err = "hipMalloc failed (suffix array)"; rc = BWAGPU_ENOMEM; goto done; }
	IDXCHK(hipMemcpyAsync(d_pac.p, pac, (size_t)(l_pac / 4 + 1), hipMemcpyHostToDevice, bl.st));
	hipLaunchKernelGGL(k_idx_text, dim3(grid_for(n_tw)), dim3(IDX_BLOCK), 0, bl.st, d_pac.as<u8>(), l_pac, d_tw.as<u64>(), n_tw);
	IDXCHK(hipMemsetAsync(d_hist.p, 0, (size_t)n_bk * 8, bl.st));
	hipLaunchKernelGGL(k_idx_bucket_count, dim3(grid_for(n, 64)), dim3(IDX_BLOCK), 0, bl.st, d_tw.as<u64>(), n, B, d_hist.as<u64>());
	IDXCHK(hipMemcpyAsync(hist.data(), d_hist.p, (size_t)n_bk * 8, hipMemcpyDeviceToHost, bl.st));
	IDXCHK(hipStreamSynchronize(bl.st));
	for (int b = 0; b < n_bk; ++b) { base[b + 1] = base[b] + hist[b]; if (hist[b] > max_bucket) max_bucket = hist[b]; }
	if (base[n_bk] != n) { bl.err = "internal: bucket histogram does not add up"; rc = BWAGPU_EHIP; goto done; }
	{	// rows: row 0 is the terminator's suffix; bucket b owns rows [1 + base[b], 1 + base[b+1])
		std::vector<u64> cur((size_t)n_bk);
		for (int b = 0; b < n_bk; ++b) cur[b] = 1 + base[b];
		IDXCHK(hipMemcpyAsync(d_cursor.p, cur.data(), (size_t)n_bk * 8, hipMemcpyHostToDevice, bl.st));
		IDXCHK(hipStreamSynchronize(bl.st));
		hipLaunchKernelGGL(k_idx_bucket_scatter, dim3(grid_for(n, 16)), dim3(IDX_BLOCK), 0, bl.st, d_tw.as<u64>(), n, B, d_cursor.as<u64>(), d_sa.as<u64>());
		const u64 row0v[2] = { n, 0 };                       // SA[0] = n (the empty suffix); rank[n] = 0
		IDXCHK(hipMemcpyAsync(d_sa.p, &row0v[0], 8, hipMemcpyHostToDevice, bl.st));
		IDXCHK(hipMemcpyAsync(d_rank.as<u64>() + n, &row0v[1], 8, hipMemcpyHostToDevice, bl.st));
		IDXCHK(hipStreamSynchronize(bl.st));
	}
	if (d_keys.ensure(max_bucket * 8) || d_keys2.ensure(max_bucket * 8) || d_vals2.ensure(max_bucket * 8) || d_s1.ensure(max_bucket * 8) ||
		d_s2.ensure(max_bucket * 8) || d_s3.ensure(max_bucket * 8)) { bl.err = "hipMalloc failed (bucket sort buffers)"; rc = BWAGPU_ENOMEM; goto done; }
	for (int b = 0; b < n_bk; ++b) {
		const u64 m = hist[b], row0 = 1 + base[b];
		if (m == 0) continue;
		u64 *rows = d_sa.as<u64>() + row0;
		hipLaunchKernelGGL(k_idx_keys, dim3(grid_for(m)), dim3(IDX_BLOCK), 0, bl.st, d_tw.as<u64>(), n, B, rows, m, d_keys.as<u64>());
		IDXRC(bl.sort_pairs(d_keys.as<u64>(), d_keys2.as<u64>(), rows, d_vals2.as<u64>(), m, 0, 64));
		hipLaunchKernelGGL(k_idx_heads, dim3(grid_for(m)), dim3(IDX_BLOCK), 0, bl.st, d_keys2.as<u64>(), m, d_s1.as<u64>());
		IDXRC(bl.scan_max(d_s1.as<u64>(), d_s2.as<u64>(), m));                                          // s2 = hs
		hipLaunchKernelGGL(k_idx_bucket_finish, dim3(grid_for(m)), dim3(IDX_BLOCK), 0, bl.st, d_vals2.as<u64>(), d_s2.as<u64>(), m, row0, d_sa.as<u64>(), d_rank.as<u64>(), d_s3.as<u64>());
		IDXRC(bl.scan_sum_excl(d_s3.as<u64>(), d_s1.as<u64>(), m));                                     // s1 = position among the bucket's unsorted suffixes
		u64 na = 0;
		IDXRC(bl.total_of(d_s1.as<u64>(), d_s3.as<u64>(), m, &na));
		if (na == 0) continue;
		// the list of unsorted suffixes grows bucket by bucket (its final size is only known at the end)
		IDXRC(bl.grow(d_act_i, (A + na) * 8, A * 8)); IDXRC(bl.grow(d_s4, (A + na) * 8, A * 8));
		hipLaunchKernelGGL(k_idx_bucket_compact, dim3(grid_for(m)), dim3(IDX_BLOCK), 0, bl.st, d_vals2.as<u64>(), d_s2.as<u64>(), d_s3.as<u64>(), d_s1.as<u64>(), m, row0,
						   d_act_i.as<u64>(), d_s4.as<u64>(), A);
		A += na;
	}
	if (bl.verbose) fprintf(stderr, "[bwagpu_index] n=%llu B=%d buckets=%d max_bucket=%llu: %llu suffixes (%.2f%%) unsorted after %llu bases\n",
							n, B, n_bk, max_bucket, A, 100.0 * A / n, K);
	IDXCHK(hipStreamSynchronize(bl.st));
	d_keys.release(); d_keys2.release(); d_vals2.release(); d_s1.release(); d_s2.release(); d_s3.release();
	if (A) {
		// group numbers of the active list (d_s4 = head row of every active suffix, non-decreasing)
		if (d_keys.ensure(A * 8) || d_keys2.ensure(A * 8) || d_vals2.ensure(A * 8) || d_s1.ensure(A * 8) || d_s2.ensure(A * 8) || d_s3.ensure(A * 8) ||
			d_act_i2.ensure(A * 8) || d_gid.ensure(A * 8) || d_gid2.ensure(A * 8) || d_ghead.ensure((A / 2 + 1) * 8) || d_ghead2.ensure((A / 2 + 1) * 8) ||
			d_gstart.ensure((A / 2 + 1) * 8) || d_gstart2.ensure((A / 2 + 1) * 8)) { bl.err = "hipMalloc failed (prefix-doubling buffers)"; rc = BWAGPU_ENOMEM; goto done; }
		hipLaunchKernelGGL(k_idx_group_flags, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_s4.as<u64>(), A, d_s1.as<u64>());
		IDXRC(bl.scan_sum_incl(d_s1.as<u64>(), d_s2.as<u64>(), A));
		hipLaunchKernelGGL(k_idx_group_init, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_s4.as<u64>(), d_s2.as<u64>(), A, d_gid.as<u64>(), d_ghead.as<u64>(), d_gstart.as<u64>());
		IDXCHK(hipMemcpyAsync(&n_groups, d_s2.as<u64>() + (A - 1), 8, hipMemcpyDeviceToHost, bl.st));
		IDXCHK(hipStreamSynchronize(bl.st));
		u64 *act_i = d_act_i.as<u64>(), *act_i2 = d_act_i2.as<u64>(), *gid = d_gid.as<u64>(), *gid2 = d_gid2.as<u64>();
		u64 *ghead = d_ghead.as<u64>(), *ghead2 = d_ghead2.as<u64>(), *gstart = d_gstart.as<u64>(), *gstart2 = d_gstart2.as<u64>();
		int round = 0;
		for (u64 h = K; A > 0; h <<= 1, ++round) {
			if (h >= (n << 1)) { bl.err = "internal: prefix doubling did not terminate"; rc = BWAGPU_EHIP; goto done; }
			const int gbits = bits_for(n_groups ? n_groups - 1 : 0);
			const u64 *gsorted = nullptr;
			if (gbits + IDX_RANK_BITS <= 64 && !getenv("BWAGPU_INDEX_SPLIT_SORT")) {
				hipLaunchKernelGGL(k_idx_round_keys, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, act_i, gid, d_rank.as<u64>(), A, n, h, d_keys.as<u64>());
				IDXRC(bl.sort_pairs(d_keys.as<u64>(), d_keys2.as<u64>(), act_i, d_vals2.as<u64>(), A, 0, IDX_RANK_BITS + (gbits ? gbits : 1)));
			} else {
				// group number and rank do not fit one 64-bit key (> 2^30 groups): order the active indices by rank, then stably by
				// group, and gather suffixes, ranks and groups through the resulting permutation
				if (d_x1.ensure(A * 8) || d_x2.ensure(A * 8) || d_x3.ensure(A * 8)) { bl.err = "hipMalloc failed (split sort)"; rc = BWAGPU_ENOMEM; goto done; }
				hipLaunchKernelGGL(k_idx_round_rank_keys, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, act_i, d_rank.as<u64>(), A, n, h, d_keys.as<u64>());   // keys = rank
				hipLaunchKernelGGL(k_idx_iota, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_x1.as<u64>(), A);
				IDXRC(bl.sort_pairs(d_keys.as<u64>(), d_keys2.as<u64>(), d_x1.as<u64>(), d_x2.as<u64>(), A, 0, IDX_RANK_BITS));                   // x2 = indices by rank
				hipLaunchKernelGGL(k_idx_gather, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, gid, d_x2.as<u64>(), A, d_x1.as<u64>());             // x1 = their groups
				IDXRC(bl.sort_pairs(d_x1.as<u64>(), d_x3.as<u64>(), d_x2.as<u64>(), d_vals2.as<u64>(), A, 0, gbits ? gbits : 1));                  // x3 = groups sorted, vals2 = permutation
				hipLaunchKernelGGL(k_idx_gather, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_keys.as<u64>(), d_vals2.as<u64>(), A, d_keys2.as<u64>());   // keys2 = ranks in (group, rank) order
				hipLaunchKernelGGL(k_idx_gather, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, act_i, d_vals2.as<u64>(), A, d_x1.as<u64>());
				IDXCHK(hipMemcpyAsync(d_vals2.p, d_x1.p, A * 8, hipMemcpyDeviceToDevice, bl.st));                                                   // vals2 = suffixes in that order
				gsorted = d_x3.as<u64>();
			}
			hipLaunchKernelGGL(k_idx_round_flags, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_keys2.as<u64>(), gsorted, A, d_s1.as<u64>());
			IDXRC(bl.scan_max(d_s1.as<u64>(), d_s2.as<u64>(), A));                                       // s2 = sf
			hipLaunchKernelGGL(k_idx_round_apply, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_keys2.as<u64>(), gsorted, d_vals2.as<u64>(), d_s2.as<u64>(), A, ghead, gstart,
							   d_sa.as<u64>(), d_rank.as<u64>(), d_s1.as<u64>(), d_s3.as<u64>());          // s1 = surv, s3 = new heads
			IDXRC(bl.scan_sum_excl(d_s1.as<u64>(), d_keys.as<u64>(), A));                                // keys = pos
			u64 A2 = 0, G2 = 0;
			IDXRC(bl.total_of(d_keys.as<u64>(), d_s1.as<u64>(), A, &A2));
			if (A2) {
				IDXRC(bl.scan_sum_incl(d_s3.as<u64>(), d_s4.as<u64>(), A));                              // s4 = running count of new heads
				IDXCHK(hipMemcpyAsync(&G2, d_s4.as<u64>() + (A - 1), 8, hipMemcpyDeviceToHost, bl.st));
				IDXCHK(hipStreamSynchronize(bl.st));
				hipLaunchKernelGGL(k_idx_round_compact, dim3(grid_for(A)), dim3(IDX_BLOCK), 0, bl.st, d_keys2.as<u64>(), gsorted, d_vals2.as<u64>(), d_s2.as<u64>(), d_s1.as<u64>(),
								   d_keys.as<u64>(), d_s4.as<u64>(), A, ghead, gstart, act_i2, gid2, ghead2, gstart2);
				{ u64 *t = gid; gid = gid2; gid2 = t; }
				{ u64 *t = act_i; act_i = act_i2; act_i2 = t; }
				{ u64 *t = ghead; ghead = ghead2; ghead2 = t; }
				{ u64 *t = gstart; gstart = gstart2; gstart2 = t; }
			}
			if (bl.verbose) fprintf(stderr, "[bwagpu_index] round %d (h=%llu): %llu -> %llu unsorted suffixes in %llu groups\n", round, h, A, A2, G2);
			A = A2; n_groups = G2;
		}
	}
	IDXCHK(hipStreamSynchronize(bl.st));
	d_rank.release(); d_keys.release(); d_keys2.release(); d_vals2.release(); d_s1.release(); d_s2.release(); d_s3.release(); d_s4.release();
	d_act_i.release(); d_act_i2.release(); d_gid.release(); d_gid2.release(); d_ghead.release(); d_ghead2.release(); d_gstart.release(); d_gstart2.release(); d_x1.release(); d_x2.release(); d_x3.release();
	{	// ---- BWT + Occ checkpoints in the reference's layout, sampled SA --------------------------------------------------------
		const u64 n_blk = (n + 127) / 128, n_words = (n + 15) / 16;
		const u64 bwt_size = n_words + (n_blk + 1) * 8;                // bwtindex.c:154-156
		const u64 n_sa = (n + (u64)sa_intv) / (u64)sa_intv;             // bwt.c:70
		hipLaunchKernelGGL(k_idx_find_primary, dim3(grid_for(n_rows)), dim3(IDX_BLOCK), 0, bl.st, d_sa.as<u64>(), n_rows, d_scalar.as<u64>());
		IDXCHK(hipMemcpyAsync(&primary, d_scalar.p, 8, hipMemcpyDeviceToHost, bl.st));
		IDXCHK(hipStreamSynchronize(bl.st));
		if (d_words.ensure(n_blk * 8 * 4) || d_out.ensure((n_blk + 1) * 16 * 4) || d_sas.ensure(n_sa * 8)) { bl.err = "hipMalloc failed (bwt)"; rc = BWAGPU_ENOMEM; goto done; }
		for (int c = 0; c < 4; ++c) if (d_c[c].ensure(n_blk * 8) || d_o[c].ensure(n_blk * 8)) { bl.err = "hipMalloc failed (occ)"; rc = BWAGPU_ENOMEM; goto done; }
		hipLaunchKernelGGL(k_idx_bwt_words, dim3(grid_for(n_blk * 8)), dim3(IDX_BLOCK), 0, bl.st, d_sa.as<u64>(), d_tw.as<u64>(), n, primary, d_words.as<u32>(), n_blk * 8);
		hipLaunchKernelGGL(k_idx_block_counts, dim3(grid_for(n_blk)), dim3(IDX_BLOCK), 0, bl.st, d_words.as<u32>(), n, n_blk, d_c[0].as<u64>(), d_c[1].as<u64>(), d_c[2].as<u64>(), d_c[3].as<u64>());
		u64 tot[4];
		for (int c = 0; c < 4; ++c) {
			IDXRC(bl.scan_sum_excl(d_c[c].as<u64>(), d_o[c].as<u64>(), n_blk));
			IDXRC(bl.total_of(d_o[c].as<u64>(), d_c[c].as<u64>(), n_blk, &tot[c]));
		}
		if (tot[0] + tot[1] + tot[2] + tot[3] != n) { bl.err = "internal: BWT symbol counts do not add up"; rc = BWAGPU_EHIP; goto done; }
		hipLaunchKernelGGL(k_idx_interleave, dim3(grid_for(n_blk * 16)), dim3(IDX_BLOCK), 0, bl.st, d_words.as<u32>(), d_o[0].as<u64>(), d_o[1].as<u64>(), d_o[2].as<u64>(), d_o[3].as<u64>(),
						   n_blk, d_out.as<u32>());
		hipLaunchKernelGGL(k_idx_sample_sa, dim3(grid_for(n_sa)), dim3(IDX_BLOCK), 0, bl.st, d_sa.as<u64>(), n_sa, sa_intv, d_sas.as<u64>());
		out->bwt = (uint32_t*)malloc((size_t)bwt_size * 4);
		out->sa = (uint64_t*)malloc((size_t)n_sa * 8);
		if (!out->bwt || !out->sa) { bl.err = "malloc failed (host copies of the index)"; rc = BWAGPU_ENOMEM; goto done; }
		// the last block keeps only the words that exist; the trailing record holds the totals (bwtindex.c:158-169)
		const u64 body_words = n_blk * 16 - (n_blk * 8 - n_words);
		IDXCHK(hipMemcpyAsync(out->bwt, d_out.p, (size_t)body_words * 4, hipMemcpyDeviceToHost, bl.st));
		IDXCHK(hipMemcpyAsync(out->sa, d_sas.p, (size_t)n_sa * 8, hipMemcpyDeviceToHost, bl.st));
		IDXCHK(hipStreamSynchronize(bl.st));
		memcpy(out->bwt + body_words, tot, 32);
		out->bwt_size = bwt_size; out->n_sa = n_sa; out->sa_intv = sa_intv; out->primary = primary; out->seq_len = n;
		out->L2[0] = 0; for (int c = 0; c < 4; ++c) out->L2[c + 1] = out->L2[c] + tot[c];
	}
	(void)hipEventRecord(ev1, bl.st); (void)hipEventSynchronize(ev1);
	{ float ms = 0; if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) out->build_ms = ms; }
done:
	if (rc) { bwagpu_built_free(out); if (errbuf && errlen) snprintf(errbuf, errlen, "%s", bl.err.c_str()); }
	if (ev0) (void)hipEventDestroy(ev0);
	if (ev1) (void)hipEventDestroy(ev1);
	(void)hipStreamDestroy(bl.st);
	return rc;
}
