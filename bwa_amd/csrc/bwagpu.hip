// bwagpu.hip -- host side of libbwagpu.so: index upload, batch arenas, kernel sequencing, the C-ABI of
// include/bwagpu.h.  Device code lives in the dev_*.h headers included below (one translation unit so that the
// per-lane routines inline into their kernels).
//
// Pipeline of one batch (all on one HIP stream, no host round-trips between stages):
//   k_seed    one lane/read   SMEM seeding (3 passes) -> SA intervals; reserves slot space, B-tree nodes
//   k_sa      one lane/seed   suffix-array lookups (LF walk to a sampled row)
//   k_chain   one lane/read   B-tree chaining, chain weights, chain filter -> kept chains + flattened seeds
//   k_seedsw  one lane/read   (long reads only) mem_flt_chained_seeds: local SW re-scoring of short seeds
//   k_extend  one lane/read   banded extension of seeds (ksw_extend2) -> raw alignment regions
//   k_dedup   one lane/read   sort / redundancy removal / patching (ksw_global2 score) -> final regions
// Arenas are sized from the batch's base count and grown (whole batch re-run) if a stage reports overflow.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include <chrono>

#include "bwagpu_config.h"
#include "dev_common.h"
#include "dev_fm.h"
#include "dev_sort.h"
#include "dev_seed.h"
#include "dev_chain.h"
#include "dev_chainw.h"
#include "dev_ext.h"
#include "dev_extw.h"
#include "dev_extp.h"
#include "dev_dedup.h"
#include "dev_seedsw.h"
#include "dev_cigar.h"
#include "dev_dedupw.h"
#include "dev_matesw.h"
#include "dev_debug.h"

#define BWAGPU_VERSION "bwagpu 0.1 (gfx950)"

// bwagpu_batch_footprint: the allocation code itself run "dry" -- ensure() adds what it would allocate to *g_dry_sum and allocates nothing --, so that the
// estimate cannot drift away from what bwagpu_batch_reserve / alloc_batch really ask for.
static thread_local size_t *g_dry_sum = nullptr;
struct DevBuf {
	void *p = nullptr; size_t cap = 0;
	int ensure(size_t bytes) {
		if (bytes <= cap) return 0;
		if (g_dry_sum) { *g_dry_sum += bytes + (bytes >> 3) + 256 - cap; return 0; }
		if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
		size_t want = bytes + (bytes >> 3) + 256;
		if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return -1; }
		cap = want;
		return 0;
	}
	void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
	template <class T> T *as() const { return (T*)p; }
};

struct bwagpu_s {
	int device = 0;
	hipStream_t stream = nullptr;
	hipEvent_t ev[8] = {};
	void *reserved[3] = {nullptr, nullptr, nullptr};   // page-locked result blocks of bwagpu_batch_reserve, held until the handle's first download
	hipEvent_t ev_wait = nullptr;    // blocking-sync event: waiting for the stream must not spin on a host core (see wait_stream)
	std::string err;
	BwagpuConfig cfg;               // tuning and test options (bwagpu_config.h): environment read once at creation, then bwagpu_set_option
	// index
	DevIndex ix = {};
	struct IndexBufs {
		DevBuf d_bwt, d_sa, d_pac, d_ctg_off, d_ctg_len, d_ctg_alt, d_ptab, d_occ32, d_occ_sb; std::atomic<int> refs{1};
		std::atomic<int> busy{0};      // handles on this index that are inside a call that launches kernels (share auto)
		// per-base arena needs learnt by any handle on this index (a re-run for arena growth doubles a batch's device time, so a
		// cloned handle should not have to learn them again); written and read under `m`
		std::mutex m; double need_slot = 0, need_node = 0, need_reg = 0; int need_mem = 0;
		long batches = 0, mem_events = 0;      // batches run on this index; those that had to be redone with longer interval lists (see the end of bwagpu_batch_run)
	};
	IndexBufs *ibuf = nullptr;      // shared by bwagpu_clone()d handles
	i64 l_pac = 0; int n_seqs = 0; u64 seq_len = 0; int sa_intv = 0;
	u64 bwt_blocks = 0, bwt_bytes = 0, sa_bytes = 0, pac_bytes = 0, bwt_size = 0, n_sa = 0;
	std::vector<i64> h_ctg_off; std::vector<i32> h_ctg_len, h_ctg_alt;
	// batch
	int n_reads = 0, max_len = 0; i64 n_bases = 0;
	bool have_batch = false, ran = false;
	int cigar_filter = 0;           // bwagpu_set_cigar_filter
	int stats_on = 0, taps_on = 0;       // stage taps cost a second region arena and a copy per batch: off unless a test asks (bwagpu_set_taps)
	bwagpu_stats_t stats = {};
	volatile int phase = 0;       // progress marker for bwagpu_debug_phase (diagnostics of a stuck call)
	i64 packed_tot = -1;          // regions packed by the last bwagpu_batch_download (-1: none)
	DevBuf d_msw_tasks, d_msw_out, d_msw_pes, d_msw_scratch;
	i64 cigl_z_cap = 0;                          // bytes per direction matrix of the long CIGAR tier's scratch (grows with the batches)
	DevBuf d_cigl_z, d_cigl_ops, d_cigl_md, d_cigl_list;      // scratch of the long-segment CIGAR tier (k_cigar_long): direction matrices, operations, MD strings per workgroup
	DevBuf d_cig_ext; i64 cig_ext_n = -1;   // operation array of the last bwagpu_batch_cigars (records with 7..64 operations point into it)
	DevBuf d_dd_tmp;                            // short-read batches: staging areas of k_dedup_wave<.., LIST>'s waves (the kept records of a read in their final order)
	DevBuf d_heavy;                             // short-read batches: the reads the lane-per-read seeding kernel gave up (their passes 1-2 run as tasks)
	DevBuf d_p2_tasks; double p2_factor = 1.;   // pass-2 tasks of a short-read batch's heavy reads (grown on overflow bit 5)
	DevBuf d_vr_tab, d_vr_ovf, d_intv_n3;   // pass 1 of long-read batches as tasks (k_seed<LR>): the reads' first tasks, the list of tasks to redo on full-size stacks, pass 3's entries per read
	DevBuf d_seq_2b, d_seq_flags; int rd_words = 0;   // per-read 2-bit copies for k_seed's LDS (k_pack_reads2b)
	DevBuf d_pack_off, d_regs_packed, d_pack_read, d_cigs, d_seq, d_seq_nib, d_off, d_ctr, d_tmp_intv, d_intv_n, d_intv_off, d_intv, d_seed_n, d_seed_off;
	DevBuf d_slot_pos, d_slot_qbeg, d_slot_len, d_slot_rid, d_slot_blob;
	DevBuf d_order, d_bin_cnt, d_seed_w, d_seed_order, d_chain_n, d_node_off, d_nodes, d_reg_off, d_reg_cap_r, d_reg_n_raw, d_reg_n, d_regs, d_regs_raw, d_dp_h, d_dp_e, d_minhsp;
	i64 slot_cap = 0, node_cap = 0, reg_cap = 0; int mem_cap = 0;
	double need_slot = 0, need_node = 0, need_reg = 0; int need_mem = 0;   // per-base arena needs learnt from earlier batches of this handle
	std::vector<i64> h_off;
};

// Wait for the handle's stream without burning a host core: hipStreamSynchronize busy-waits, and a process that keeps several
// batches in flight from several host threads would spend that many cores spinning -- cores the finalize stage needs.
#define wait_stream(h) ((h)->ev_wait ? (hipEventRecord((h)->ev_wait, (h)->stream) == hipSuccess ? hipEventSynchronize((h)->ev_wait) : hipStreamSynchronize((h)->stream)) : hipStreamSynchronize((h)->stream))

#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (h)->err = std::string(#call) + ": " + hipGetErrorString(e_); return BWAGPU_EHIP; } } while (0)

extern "C" const char *bwagpu_version(void) { return BWAGPU_VERSION; }
// sizeof of the public structs as this library was compiled (bindings check their mirrors against it)
extern "C" void bwagpu_abi_sizes(int32_t out[8])
{
	out[0] = (int32_t)sizeof(bwagpu_opt_t); out[1] = (int32_t)sizeof(bwagpu_alnreg_t); out[2] = (int32_t)sizeof(bwagpu_stats_t); out[3] = (int32_t)sizeof(bwagpu_index_desc_t);
	out[4] = (int32_t)sizeof(bwagpu_cigar_t); out[5] = (int32_t)sizeof(bwagpu_matesw_t); out[6] = (int32_t)sizeof(bwagpu_bseq1_t); out[7] = (int32_t)sizeof(bwagpu_built_t);
}

extern "C" const char *bwagpu_strerror(int code)
{
	switch (code) {
	case BWAGPU_OK: return "ok";
	case BWAGPU_ENODEV: return "no usable HIP device";
	case BWAGPU_EINVAL: return "invalid argument";
	case BWAGPU_ENOMEM: return "out of memory";
	case BWAGPU_EIO: return "index files missing or inconsistent";
	case BWAGPU_EHIP: return "HIP runtime error";
	case BWAGPU_EUNSUP: return "unsupported option on the device path";
	default: return "unknown error";
	}
}
extern "C" const char *bwagpu_last_error(const bwagpu_t *h) { return h ? h->err.c_str() : ""; }
// ---- result buffers ------------------------------------------------------------------------------------------------
// The large per-batch results (regions, CIGAR records, operation array, mate-rescue records: ~300 MB per 667 k reads) are copied out of the
// device with hipMemcpyAsync.  Into malloc'ed memory the runtime stages such a copy through bounce buffers; into page-locked memory it is one
// DMA at the link's rate.  Page-locking itself costs more than the copy, so the blocks are kept: bwagpu_free puts a pooled block back,
// result_alloc hands out the smallest free block that fits (sizes are rounded up so that consecutive batches find each other's blocks).
// Results under 1 MiB (BWAGPU_PINNED_MIN_KB), and everything when BWAGPU_PINNED_RESULTS=0, come from malloc as before.  The pool is bounded (4 GiB page-locked).
namespace {
struct ResultPool {
	std::mutex m;
	std::map<void*, size_t> live;                   // blocks handed out -> capacity
	std::multimap<size_t, void*> idle;              // capacity -> block
	size_t pinned = 0;
	std::atomic<long long> on{1}, min_kb{1024}, cap_mb{16384};
	std::atomic<bool> warned{false};
	std::atomic<bool> inited{false};                // settings taken from a handle's options (init_config) or, for a block asked for before any handle exists, from the environment
	void *get(size_t bytes)
	{
		if (!inited.exchange(true)) { BwagpuConfig c; c.from_env(); on = c.pinned_results; min_kb = c.pinned_min_kb < 0 ? 0 : c.pinned_min_kb; cap_mb = c.pinned_cap_mb < 0 ? 0 : c.pinned_cap_mb; }
		const size_t cap_total = (size_t)cap_mb.load() << 20;
		const bool enabled = on.load() != 0;                       // (options pinned_results / pinned_min_kb: process-wide, set whenever a handle is created or the option is set)
		const size_t min_bytes = (size_t)min_kb.load() << 10;
		if (!enabled || bytes < min_bytes) return malloc(bytes ? bytes : 1);
		size_t step = (size_t)1 << 20;                 // block sizes: multiples of 1 MiB up to 8 MiB, then of a quarter of the power of two below
		while (step * 8 <= bytes) step <<= 1;
		const size_t want = (bytes + step - 1) / step * step;
		{
			std::lock_guard<std::mutex> l(m);
			auto it = idle.lower_bound(bytes);
			if (it != idle.end() && it->first <= want * 2) { void *p = it->second; live[p] = it->first; idle.erase(it); return p; }
			if (pinned + want > cap_total) {           // make room: drop idle blocks, smallest first
				while (!idle.empty() && pinned + want > cap_total) { auto b = idle.begin(); (void)hipHostFree(b->second); pinned -= b->first; idle.erase(b); }
				if (pinned + want > cap_total) {
					if (!warned.exchange(true)) fprintf(stderr, "[W::bwagpu] the page-locked result pool is full (%lld MiB, option pinned_cap_mb): further results are copied through pageable memory\n", cap_mb.load());
					return malloc(bytes);
				}
			}
			pinned += want;
		}
		void *p = nullptr;
		if (hipHostMalloc(&p, want, hipHostMallocPortable) != hipSuccess || !p) {     // (portable: a block may serve another device's handle next time)
			(void)hipGetLastError();
			std::lock_guard<std::mutex> l(m); pinned -= want;
			return malloc(bytes);
		}
		std::lock_guard<std::mutex> l(m);
		live[p] = want;
		return p;
	}
	// release the idle blocks (bwagpu_trim; also when the process's last handle is destroyed: an embedding program gets the page-locked
	// memory back with the device)
	void trim()
	{
		std::lock_guard<std::mutex> l(m);
		for (auto &b : idle) { (void)hipHostFree(b.second); pinned -= b.first; }
		idle.clear();
	}
	void put(void *p)
	{
		if (!p) return;
		{
			std::lock_guard<std::mutex> l(m);
			auto it = live.find(p);
			if (it != live.end()) { idle.emplace(it->second, p); live.erase(it); return; }
		}
		free(p);
	}
};
ResultPool g_results;
}
static void *result_alloc(size_t bytes) { return g_results.get(bytes); }
extern "C" void bwagpu_free(void *p) { g_results.put(p); }
extern "C" void *bwagpu_alloc_host(size_t bytes) { return g_results.get(bytes); }
extern "C" void bwagpu_trim(void) { g_results.trim(); }
static std::atomic<int> g_live_handles{0};

static int upload(bwagpu_t *h, DevBuf &b, const void *src, size_t bytes)
{
	if (b.ensure(bytes ? bytes : 16)) { h->err = "hipMalloc failed"; return BWAGPU_ENOMEM; }
	if (bytes) HIPCHK(h, hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice));
	return 0;
}

// ---- options (bwagpu_config.h) ----------------------------------------------------------------------------------------------
namespace {
std::mutex g_cfg_m;
std::map<std::string, long long> g_cfg_defaults;     // bwagpu_set_default_option: applied to handles created afterwards, on top of the environment
}
// Ranges of the options (one test for all three ways in -- bwagpu_set_option, bwagpu_set_default_option, the environment): sizes and counts that the
// batch calls cast to int or multiply, and the options with a fixed set of kernel instances behind them (-1 = auto where the list says so).
// seed_lds_ent: 16 entries x 16 B x 256 lanes is the whole 64 KiB a workgroup's dynamic LDS may take.
static bool option_in_range(const BwagpuConfig &c, const long long *f, long long value)
{
	auto in = [&](long long lo, long long hi) { return value >= lo && value <= hi; };
	if (f == &c.ext_occ) return value == 4 || value == 6;
	if (f == &c.seed_mrg) return value == -1 || value == 0 || value == 2;
	if (f == &c.share) return in(-1, 100);
	if (f == &c.seed_task_stack || f == &c.seed_p2_cap || f == &c.mem_cap || f == &c.seed_grid || f == &c.cig_ops_cap || f == &c.idx_desc_max_mb) return in(0, 0x3fffffff);
	if (f == &c.seed_budget || f == &c.dedup_heavy) return in(-1, 0x3fffffff);
	if (f == &c.dedup_stage || f == &c.dedup_big || f == &c.dedup_net) return in(-1, 1024);
	if (f == &c.seed_lds_ent) return in(-1, 16);
	if (f == &c.dedup_ring) return value == 0 || (in(256, 4096) && (value & (value - 1)) == 0);
	if (f == &c.cigl_mib) return in(0, 1 << 20);
	if (f == &c.cig_tiers || f == &c.chain_regs) return in(0, 2);
	if (f == &c.chain_flt_lds) return in(0, CW_FLT_LDS);
	if (f == &c.ptab_m) return in(0, PTAB_MAX);
	if (f == &c.occ32_sb_shift) return in(8, 32);
	return true;
}
static void init_config(BwagpuConfig &c)
{
	c = BwagpuConfig();
	const BwagpuConfig dflt;
	c.from_env();
	// a value from the environment that is out of range is dropped with a warning (the handle keeps the compiled-in default); bwagpu_set_default_option
	// refuses such values when they are set
#define X(name, d) if (!option_in_range(c, &c.name, c.name)) { fprintf(stderr, "[W::bwagpu] BWAGPU_%s=%lld is out of range: ignored\n", #name, c.name); c.name = dflt.name; }
	BWAGPU_OPTION_LIST(X)
#undef X
	std::lock_guard<std::mutex> l(g_cfg_m);
	for (auto &kv : g_cfg_defaults) if (long long *f = c.field(kv.first.c_str())) *f = kv.second;
	g_results.on = c.pinned_results; g_results.min_kb = c.pinned_min_kb < 0 ? 0 : c.pinned_min_kb; g_results.cap_mb = c.pinned_cap_mb < 0 ? 0 : c.pinned_cap_mb; g_results.inited = true;
}
extern "C" int bwagpu_set_default_option(const char *key, long long value)
{
	BwagpuConfig probe;
	const long long *f = probe.field(key);
	if (!f || !option_in_range(probe, f, value)) return BWAGPU_EINVAL;
	std::lock_guard<std::mutex> l(g_cfg_m);
	g_cfg_defaults[key] = value;
	return BWAGPU_OK;
}
extern "C" void bwagpu_clear_default_options(void) { std::lock_guard<std::mutex> l(g_cfg_m); g_cfg_defaults.clear(); }
extern "C" int bwagpu_set_option(bwagpu_t *h, const char *key, long long value)
{
	if (!h) return BWAGPU_EINVAL;
	long long *f = h->cfg.field(key);
	if (!f) return BWAGPU_EINVAL;
	// the index-side options shape what bwagpu_create / bwagpu_index_ready derive from the index: per handle they can only be set before that
	// happens (a handle created with NULL arrays, ahead of the broadcast); otherwise use bwagpu_set_default_option before creating the handle
	if ((f == &h->cfg.occ32 || f == &h->cfg.occ32_sb_shift || f == &h->cfg.ptab_m) && (h->ix.occ32 || h->ix.ptab) && *f != value) return BWAGPU_EINVAL;
	if (!option_in_range(h->cfg, f, value)) return BWAGPU_EINVAL;
	*f = value;
	if (f == &h->cfg.pinned_results) g_results.on = value;
	if (f == &h->cfg.pinned_min_kb) g_results.min_kb = value < 0 ? 0 : value;
	if (f == &h->cfg.pinned_cap_mb) g_results.cap_mb = value < 0 ? 0 : value;
	return BWAGPU_OK;
}
extern "C" int bwagpu_get_option(const bwagpu_t *h, const char *key, long long *value)
{
	if (!h || !value) return BWAGPU_EINVAL;
	const long long *f = const_cast<bwagpu_t*>(h)->cfg.field(key);
	if (!f) return BWAGPU_EINVAL;
	*value = *f;
	return BWAGPU_OK;
}
extern "C" int bwagpu_option_name(int i, const char **name)
{
	static const char *const names[] = {
#define X(n, d) #n,
		BWAGPU_OPTION_LIST(X)
#undef X
	};
	if (!name || i < 0 || i >= (int)(sizeof names / sizeof names[0])) return BWAGPU_EINVAL;
	*name = names[i];
	return BWAGPU_OK;
}

// ---- prefix tables (DevIndex::ptab): level j from level j-1 with the sweep's own extension ----------------------------------
__global__ void __launch_bounds__(256) k_ptab_level(DevIndex ix, u64 *tab, int j)
{
	const u64 n = (u64)1 << (2 * j);
	for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (u64)gridDim.x * blockDim.x) {
		BiIntv out;
		if (j == 1) fm_init(ix, (int)c, out);                          // bwt_set_intv (bwt.h:82)
		else {
			const u64 *pe = tab + ((((u64)1 << (2 * (j - 1))) - 4) / 3 + (c >> 2)) * 3;
			BiIntv par; par.x0 = pe[0]; par.x1 = pe[1]; par.x2 = pe[2]; par.info = 0;
			// forward extension by base c&3 (bwt.c:307-308).  Empty parents are extended too, by the same arithmetic:
			// bwt_seed_strategy1 keeps extending an interval that has become empty until min_seed_len is reached (bwt.c:364-377)
			fm_extend1(ix, par, 3 - (int)(c & 3), 0, out);
		}
		u64 *e = tab + ((n - 4) / 3 + c) * 3;
		e[0] = out.x0; e[1] = out.x1; e[2] = out.x2;
	}
}

// records: for every m-mer W the packed bi-intervals of its m prefixes, gathered from the level tables
__global__ void __launch_bounds__(256) k_ptab_records(const u64 *tab, uint4 *rec, int m)
{
	const u64 n = ((u64)1 << (2 * m)) * (u64)m;
	for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (u64)gridDim.x * blockDim.x) {
		const u64 w = t / (u64)m; const int j = (int)(t % (u64)m) + 1;
		const u64 *e = tab + ((((u64)1 << (2 * j)) - 4) / 3 + (w >> (2 * (m - j)))) * 3;
		BiIntv v; v.x0 = e[0]; v.x1 = e[1]; v.x2 = e[2]; v.info = 0;
		rec[t] = SeedStack::pack(v);
	}
}

// ---- the 32-byte block layout of the BWT (DevIndex::occ32) --------------------------------------------------------------------
// counts of the four symbols at the start of 64-base block j, from the reference-format block it is half of
DEVFN void occ32_start_counts(const DevIndex &ix, u64 j, u64 cnt[4])
{
	const uint4 *src = ix.bwt + (j >> 1) * 4;
	const uint4 c01 = src[0], c23 = src[1];
	cnt[0] = (u64)c01.y << 32 | c01.x; cnt[1] = (u64)c01.w << 32 | c01.z; cnt[2] = (u64)c23.y << 32 | c23.x; cnt[3] = (u64)c23.w << 32 | c23.z;
	if (j & 1) {                           // second half: add the first 64 bases of the block
		const uint4 w0 = src[2];
		u32 c1 = 0, c2 = 0, c3 = 0;
		count_pair(w0.x, w0.y, 32, c1, c2, c3); count_pair(w0.z, w0.w, 32, c1, c2, c3);
		cnt[0] += 64 - c1 - c2 - c3; cnt[1] += c1; cnt[2] += c2; cnt[3] += c3;
	}
}
__global__ void __launch_bounds__(256) k_occ32_sb(DevIndex ix, u64 *sb, u64 n_sb, u64 n_new, int sh /* log2 of 64-base blocks per superblock */)
{
	for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < n_sb; s += (u64)gridDim.x * blockDim.x) {
		u64 cnt[4] = { 0, 0, 0, 0 };
		if ((s << sh) < n_new) occ32_start_counts(ix, s << sh, cnt);   // (the table has one spare entry past the end)
		for (int k = 0; k < 4; ++k) sb[s * 4 + k] = cnt[k];
	}
}
// Second form of the superblock table for the one-trip rank routine (dev_fm.h, occ32x_*): entry [s][c] = { count of c, sum of the counts of
// the symbols above c } -- what an extension by c needs from a position's superblock in ONE 16-byte load instead of the whole 32-byte row
__global__ void __launch_bounds__(256) k_occ32_sbx(const u64 *sb, uint4 *out, u64 n_sb)
{
	for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n_sb * 4; t += (u64)gridDim.x * blockDim.x) {
		const u64 *row = sb + (t >> 2) * 4; const int c = (int)(t & 3);
		u64 above = 0;
		for (int i = c + 1; i < 4; ++i) above += row[i];
		out[t] = make_uint4((u32)row[c], (u32)(row[c] >> 32), (u32)above, (u32)(above >> 32));
	}
}
__global__ void __launch_bounds__(256) k_occ32_blocks(DevIndex ix, const u64 *sb, uint4 *out, u64 n_new, int sh)
{
	for (u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x; j < n_new; j += (u64)gridDim.x * blockDim.x) {
		u64 cnt[4];
		occ32_start_counts(ix, j, cnt);
		const u64 *base = sb + (j >> sh) * 4;
		uint4 rel; rel.x = (u32)(cnt[0] - base[0]); rel.y = (u32)(cnt[1] - base[1]); rel.z = (u32)(cnt[2] - base[2]); rel.w = (u32)(cnt[3] - base[3]);
		out[j * 2] = rel;
		out[j * 2 + 1] = ix.bwt[(j >> 1) * 4 + 2 + (j & 1)];
	}
}
// The seeding and SA kernels read this layout (default; BWAGPU_OCC32=0 keeps them on the reference-format blocks): built from the resident
// reference-format blocks, which stay the interchange format (files, index broadcast, bwagpu_index_buffers).  Measured at 3.1 Gbp
// (profiles/r03_seed_variants.md): k_seed 94.6 ms on the 64-byte blocks, 84.6 ms on these -- and 134.5 ms with the 64-byte blocks fetched
// quad-cooperatively (a round-3 variant, deleted in round 4), although that fetch pattern moves the chip's random-block ceiling from 22.9e9 to
// 51.3e9 per second (tools/randbw3.hip): the cooperative form adds instructions and registers (3 waves per SIMD) in front of every trip.
static int build_occ32(bwagpu_t *h)
{
	h->ix.occ32 = nullptr; h->ix.occ_sb = nullptr; h->ix.occ_sbx = nullptr; h->ix.occ_sb_shift = 32; h->ix.occ32_bytes = h->ix.occ_sbx_bytes = 0;
	if (h->cfg.occ32 == 0 || h->bwt_blocks == 0) return 0;   // (option occ32 = 0: keep to the reference-format blocks)
	int shift = (int)h->cfg.occ32_sb_shift;      // (tests: small superblocks on small genomes)
	if (shift < 8) shift = 8; if (shift > 32) shift = 32;
	const int sh = shift - 6;
	const u64 n_new = (u64)h->bwt_blocks * 2, n_sb = (n_new >> sh) + 1;
	if (h->ibuf->d_occ32.ensure((size_t)n_new * 32) || h->ibuf->d_occ_sb.ensure((size_t)n_sb * (32 + 64))) { h->err = "hipMalloc failed (32-byte blocks)"; return BWAGPU_ENOMEM; }   // (both forms of the superblock table, one after the other)
	hipLaunchKernelGGL(k_occ32_sb, dim3((unsigned)((n_sb + 255) / 256 < 1024 ? (n_sb + 255) / 256 : 1024)), dim3(256), 0, h->stream, h->ix, h->ibuf->d_occ_sb.as<u64>(), n_sb, n_new, sh);
	hipLaunchKernelGGL(k_occ32_blocks, dim3((unsigned)((n_new + 255) / 256 < 65536 ? (n_new + 255) / 256 : 65536)), dim3(256), 0, h->stream, h->ix, h->ibuf->d_occ_sb.as<u64>(), h->ibuf->d_occ32.as<uint4>(), n_new, sh);
	hipLaunchKernelGGL(k_occ32_sbx, dim3((unsigned)((n_sb * 4 + 255) / 256 < 1024 ? (n_sb * 4 + 255) / 256 : 1024)), dim3(256), 0, h->stream, h->ibuf->d_occ_sb.as<u64>(), (uint4*)(h->ibuf->d_occ_sb.as<u64>() + n_sb * 4), n_sb);
	hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(h->stream);
	HIPCHK(h, e1); HIPCHK(h, e2);
	h->ix.occ32 = h->ibuf->d_occ32.as<uint4>(); h->ix.occ_sb = h->ibuf->d_occ_sb.as<u64>(); h->ix.occ_sb_shift = shift;
	h->ix.occ_sbx = (const uint4*)(h->ibuf->d_occ_sb.as<u64>() + n_sb * 4);
	h->ix.occ32_bytes = n_new * 32; h->ix.occ_sbx_bytes = n_sb * 64;
	return 0;
}

static int build_prefix_tables(bwagpu_t *h, int m)
{
	// the packed entries hold 37-bit interval bounds (as do the LDS interval stacks)
	h->ix.ptab_bytes = 0;
	if (m < 2 || h->ix.seq_len >= ((u64)1 << 36)) { h->ix.ptab = nullptr; h->ix.ptab_m = 0; return 0; }
	if (m > PTAB_MAX) m = PTAB_MAX;
	size_t entries = ((((size_t)1 << (2 * (m + 1))) - 4) / 3);
	DevBuf levels;
	if (levels.ensure(entries * 24) || h->ibuf->d_ptab.ensure(((size_t)1 << (2 * m)) * m * sizeof(uint4))) { levels.release(); h->err = "hipMalloc failed (prefix tables)"; return BWAGPU_ENOMEM; }
	DevIndex ix = h->ix; ix.ptab = nullptr; ix.ptab_m = 0;
	for (int j = 1; j <= m; ++j) {
		u64 n = (u64)1 << (2 * j);
		unsigned nb = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
		hipLaunchKernelGGL(k_ptab_level, dim3(nb), dim3(256), 0, h->stream, ix, levels.as<u64>(), j);
	}
	const u64 n_rec = ((u64)1 << (2 * m)) * (u64)m;
	hipLaunchKernelGGL(k_ptab_records, dim3((unsigned)((n_rec + 255) / 256 < 8192 ? (n_rec + 255) / 256 : 8192)), dim3(256), 0, h->stream, levels.as<u64>(), h->ibuf->d_ptab.as<uint4>(), m);
	hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(h->stream);
	levels.release();
	HIPCHK(h, e1); HIPCHK(h, e2);
	h->ix.ptab = h->ibuf->d_ptab.as<uint4>(); h->ix.ptab_m = m; h->ix.ptab_bytes = n_rec * sizeof(uint4);
	return 0;
}

extern "C" int bwagpu_create(bwagpu_t **out, const bwagpu_index_desc_t *d, int device)
{
	if (!out || !d || d->n_seqs <= 0 || !d->ctg_offset || !d->ctg_len || !d->ctg_is_alt) return BWAGPU_EINVAL;
	const bool alloc_only = !d->bwt && !d->sa && !d->pac;   // receiving side of an index broadcast
	if (!alloc_only && (!d->bwt || !d->sa || !d->pac)) return BWAGPU_EINVAL;
	if (d->sa_intv <= 0 || (d->sa_intv & (d->sa_intv - 1))) return BWAGPU_EINVAL;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return BWAGPU_ENODEV;
	if (hipSetDevice(device) != hipSuccess) return BWAGPU_ENODEV;
	bwagpu_t *h = new bwagpu_s(); ++g_live_handles;
	h->device = device;
	h->ibuf = new bwagpu_s::IndexBufs();
	init_config(h->cfg);
	int rc;
	if (hipStreamCreate(&h->stream) != hipSuccess) { h->stream = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	for (int i = 0; i < 8; ++i) if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->ev[i] = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	if (hipEventCreateWithFlags(&h->ev_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) h->ev_wait = nullptr;
	// the last Occ record of the .bwt is a trailing 32-byte half block; pad the upload to whole 64-byte blocks
	u64 nblk = (d->bwt_size + 15) / 16;
	std::vector<uint32_t> padded;
	const uint32_t *src = d->bwt;
	if (!alloc_only && d->bwt_size % 16) { padded.assign(nblk * 16, 0); memcpy(padded.data(), d->bwt, d->bwt_size * 4); src = padded.data(); }
	h->bwt_bytes = nblk * 64; h->sa_bytes = d->n_sa * 8; h->pac_bytes = (u64)(d->l_pac / 4 + 1);
	h->bwt_size = d->bwt_size; h->n_sa = d->n_sa;
	if (alloc_only) {
		if (h->ibuf->d_bwt.ensure(h->bwt_bytes) || h->ibuf->d_sa.ensure(h->sa_bytes) || h->ibuf->d_pac.ensure(h->pac_bytes)) { rc = BWAGPU_ENOMEM; goto fail; }
	} else {
		if ((rc = upload(h, h->ibuf->d_bwt, src, nblk * 64))) goto fail;
		if ((rc = upload(h, h->ibuf->d_sa, d->sa, d->n_sa * 8))) goto fail;
		if ((rc = upload(h, h->ibuf->d_pac, d->pac, (size_t)(d->l_pac / 4 + 1)))) goto fail;
	}
	if ((rc = upload(h, h->ibuf->d_ctg_off, d->ctg_offset, (size_t)d->n_seqs * 8))) goto fail;
	if ((rc = upload(h, h->ibuf->d_ctg_len, d->ctg_len, (size_t)d->n_seqs * 4))) goto fail;
	if ((rc = upload(h, h->ibuf->d_ctg_alt, d->ctg_is_alt, (size_t)d->n_seqs * 4))) goto fail;
	h->bwt_blocks = nblk;
	h->ix.bwt = h->ibuf->d_bwt.as<uint4>();
	h->ix.primary = d->primary; for (int i = 0; i < 5; ++i) h->ix.L2[i] = d->L2[i];
	h->ix.seq_len = d->seq_len;
	h->ix.sa = h->ibuf->d_sa.as<u64>(); h->ix.sa_mask = (u64)d->sa_intv - 1;
	h->ix.sa_shift = 0; while ((1 << h->ix.sa_shift) < d->sa_intv) ++h->ix.sa_shift;
	h->ix.pac = h->ibuf->d_pac.as<u8>(); h->ix.l_pac = d->l_pac;
	h->ix.n_seqs = d->n_seqs; h->ix.ctg_off = h->ibuf->d_ctg_off.as<i64>(); h->ix.ctg_len = h->ibuf->d_ctg_len.as<i32>(); h->ix.ctg_alt = h->ibuf->d_ctg_alt.as<i32>();
	h->l_pac = d->l_pac; h->n_seqs = d->n_seqs; h->seq_len = d->seq_len; h->sa_intv = d->sa_intv;
	h->h_ctg_off.assign(d->ctg_offset, d->ctg_offset + d->n_seqs);
	h->h_ctg_len.assign(d->ctg_len, d->ctg_len + d->n_seqs);
	h->h_ctg_alt.assign(d->ctg_is_alt, d->ctg_is_alt + d->n_seqs);
	h->ix.ptab = nullptr; h->ix.ptab_m = 0; h->ix.ptab_bytes = 0; h->ix.occ32 = nullptr; h->ix.occ_sb = nullptr; h->ix.occ_sbx = nullptr; h->ix.occ32_bytes = h->ix.occ_sbx_bytes = 0;
	if (!alloc_only) {   // (a handle that receives its index by broadcast builds them in bwagpu_index_ready)
		if ((rc = build_occ32(h))) goto fail;
		if ((rc = build_prefix_tables(h, (int)h->cfg.ptab_m))) goto fail;
	}
	*out = h;
	return BWAGPU_OK;
fail:
	bwagpu_destroy(h);
	return rc;
}

extern "C" void bwagpu_destroy(bwagpu_t *h)
{
	if (!h) return;
	struct LastOut { ~LastOut() { if (--g_live_handles == 0) g_results.trim(); } } last_out;      // (after the handle's own buffers are gone)
	for (void *&p_ : h->reserved) { if (p_) bwagpu_free(p_); p_ = nullptr; }
	if (h->ibuf && --h->ibuf->refs == 0) {
		DevBuf *ib[] = { &h->ibuf->d_bwt, &h->ibuf->d_sa, &h->ibuf->d_pac, &h->ibuf->d_ctg_off, &h->ibuf->d_ctg_len, &h->ibuf->d_ctg_alt, &h->ibuf->d_ptab, &h->ibuf->d_occ32, &h->ibuf->d_occ_sb };
		for (DevBuf *b : ib) b->release();
		delete h->ibuf;
	}
	DevBuf *all[] = { &h->d_heavy, &h->d_dd_tmp, &h->d_p2_tasks, &h->d_vr_tab, &h->d_vr_ovf, &h->d_intv_n3, &h->d_cigl_list, &h->d_cigl_z, &h->d_cigl_ops, &h->d_cigl_md, &h->d_seq_2b, &h->d_seq_flags, &h->d_cig_ext, &h->d_msw_tasks, &h->d_msw_out, &h->d_msw_pes, &h->d_msw_scratch, &h->d_pack_off, &h->d_regs_packed, &h->d_pack_read, &h->d_cigs, &h->d_seq, &h->d_seq_nib, &h->d_off, &h->d_ctr, &h->d_tmp_intv,
		&h->d_intv_n, &h->d_intv_off, &h->d_intv, &h->d_seed_n, &h->d_seed_off, &h->d_slot_pos, &h->d_slot_qbeg, &h->d_slot_len, &h->d_slot_rid, &h->d_slot_blob, &h->d_chain_n, &h->d_node_off,
		&h->d_order, &h->d_bin_cnt, &h->d_seed_w, &h->d_seed_order, &h->d_nodes, &h->d_reg_off, &h->d_reg_cap_r, &h->d_reg_n_raw, &h->d_reg_n, &h->d_regs, &h->d_regs_raw, &h->d_dp_h, &h->d_dp_e, &h->d_minhsp };
	for (DevBuf *b : all) b->release();
	for (int i = 0; i < 8; ++i) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
	if (h->ev_wait) (void)hipEventDestroy(h->ev_wait);
	if (h->stream) (void)hipStreamDestroy(h->stream);
	delete h;
}

// ---- index files (formats: bwt.c:385-462, bntseq.c:97-211, bwa.c:300-312) -------------------------------------
static bool read_file(const std::string &fn, std::vector<char> &out)
{
	FILE *fp = fopen(fn.c_str(), "rb");
	if (!fp) return false;
	fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
	out.resize((size_t)sz);
	bool ok = sz == 0 || fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
	fclose(fp);
	return ok;
}

extern "C" int bwagpu_create_from_files(bwagpu_t **out, const char *prefix, int device)
{
	if (!out || !prefix) return BWAGPU_EINVAL;
	std::string pre(prefix);
	std::vector<char> fb, fs, fp;
	if (!read_file(pre + ".bwt", fb) || fb.size() < 40 || !read_file(pre + ".sa", fs) || fs.size() < 56 || !read_file(pre + ".pac", fp)) return BWAGPU_EIO;
	bwagpu_index_desc_t d; memset(&d, 0, sizeof d);
	const u64 *hb = (const u64*)fb.data();
	d.primary = hb[0]; d.L2[0] = 0; for (int i = 0; i < 4; ++i) d.L2[i + 1] = hb[1 + i];
	d.seq_len = d.L2[4];
	d.bwt = (const uint32_t*)(fb.data() + 40); d.bwt_size = (fb.size() - 40) / 4;
	const u64 *hs = (const u64*)fs.data();
	if (hs[0] != d.primary || hs[6] != d.seq_len) return BWAGPU_EIO;
	d.sa_intv = (int)hs[5];
	if (d.sa_intv <= 0) return BWAGPU_EIO;
	d.n_sa = (d.seq_len + d.sa_intv) / d.sa_intv;
	if (fs.size() < 56 + (d.n_sa - 1) * 8) return BWAGPU_EIO;
	std::vector<u64> sa(d.n_sa);
	sa[0] = (u64)-1;                                   // bwt_restore_sa forces sa[0] = -1 (bwt.c:437)
	memcpy(sa.data() + 1, fs.data() + 56, (d.n_sa - 1) * 8);
	d.sa = sa.data();
	// .ann: "l_pac n_seqs seed" then per contig "gi name [anno]" / "offset len n_ambs"
	FILE *fa = fopen((pre + ".ann").c_str(), "r");
	if (!fa) return BWAGPU_EIO;
	long long xx; int n_seqs; unsigned seed;
	if (fscanf(fa, "%lld%d%u", &xx, &n_seqs, &seed) != 3 || n_seqs <= 0) { fclose(fa); return BWAGPU_EIO; }
	d.l_pac = xx; d.n_seqs = n_seqs;
	std::vector<i64> coff(n_seqs); std::vector<i32> clen(n_seqs), calt(n_seqs, 0); std::vector<std::string> names(n_seqs);
	for (int i = 0; i < n_seqs; ++i) {
		char name[8192]; unsigned gi; int c, n_ambs, len;
		if (fscanf(fa, "%u%8191s", &gi, name) != 2) { fclose(fa); return BWAGPU_EIO; }
		names[i] = name;
		while ((c = fgetc(fa)) != '\n' && c != EOF) {}
		if (fscanf(fa, "%lld%d%d", &xx, &len, &n_ambs) != 3) { fclose(fa); return BWAGPU_EIO; }
		coff[i] = xx; clen[i] = len;
	}
	fclose(fa);
	if (FILE *fl = fopen((pre + ".alt").c_str(), "r")) {   // first column of each non-@ line names an ALT contig (bns_restore, bntseq.c:185-205)
		std::string name; int ch;
		while ((ch = fgetc(fl)) != EOF) {            // lines of any length; like the reference, a last line without a line end is not seen
			if (ch == '\t' || ch == '\n' || ch == '\r') {
				if (!name.empty() && name[0] != '@') for (int i = 0; i < n_seqs; ++i) if (names[i] == name) calt[i] = 1;
				while (ch != '\n' && ch != EOF) ch = fgetc(fl);
				name.clear();
			} else name += (char)ch;
		}
		fclose(fl);
	}
	if ((i64)fp.size() < d.l_pac / 4 + 1) return BWAGPU_EIO;
	d.pac = (const u8*)fp.data();
	d.ctg_offset = coff.data(); d.ctg_len = clen.data(); d.ctg_is_alt = calt.data();
	return bwagpu_create(out, &d, device);
}

// A second handle on the same device that shares the index already resident in HBM (no copy) but has its own stream and
// batch arenas: two handles driven from two host threads keep two batches in flight, so the latency-bound tails of one
// batch (chaining of repeat-rich reads) overlap with the throughput-bound kernels of the other.
extern "C" int bwagpu_clone(bwagpu_t *src, bwagpu_t **out)
{
	if (!src || !out) return BWAGPU_EINVAL;
	if (hipSetDevice(src->device) != hipSuccess) return BWAGPU_ENODEV;
	bwagpu_t *h = new bwagpu_s(); ++g_live_handles;
	h->device = src->device;
	h->ibuf = src->ibuf; ++h->ibuf->refs;     // (bwagpu_destroy drops the reference again on every failure path below)
	if (hipStreamCreate(&h->stream) != hipSuccess) { h->stream = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	for (int i = 0; i < 8; ++i) if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->ev[i] = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	if (hipEventCreateWithFlags(&h->ev_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) h->ev_wait = nullptr;
	h->ix = src->ix; h->l_pac = src->l_pac; h->n_seqs = src->n_seqs; h->seq_len = src->seq_len; h->sa_intv = src->sa_intv;
	h->bwt_blocks = src->bwt_blocks; h->bwt_bytes = src->bwt_bytes; h->sa_bytes = src->sa_bytes; h->pac_bytes = src->pac_bytes;
	h->bwt_size = src->bwt_size; h->n_sa = src->n_sa;
	h->h_ctg_off = src->h_ctg_off; h->h_ctg_len = src->h_ctg_len; h->h_ctg_alt = src->h_ctg_alt;
	h->stats_on = src->stats_on; h->taps_on = src->taps_on; h->cigar_filter = src->cigar_filter; h->cfg = src->cfg;
	*out = h;
	return BWAGPU_OK;
}

// A handle on ANOTHER device of the node holding a copy of src's resident index: the three index buffers, the
// (possibly densified) SA, the contig table and the prefix tables travel device to device with hipMemcpyPeer -- point to point
// over xGMI, the single-process counterpart of the RCCL broadcast that bwa_amd/dist.py does between processes.  The new handle
// owns its copy; bwagpu_clone() on it gives further streams on that device.
static int clone_to_device_impl(bwagpu_t *src, int device, bwagpu_t **out)
{
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return BWAGPU_ENODEV;
	if (device == src->device) return bwagpu_clone(src, out);
	if (hipSetDevice(device) != hipSuccess) return BWAGPU_ENODEV;
	bwagpu_t *h = new bwagpu_s(); ++g_live_handles;
	h->device = device;
	h->ibuf = new bwagpu_s::IndexBufs();
	h->cfg = src->cfg;
	if (hipStreamCreate(&h->stream) != hipSuccess) { h->stream = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	for (int i = 0; i < 8; ++i) if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->ev[i] = nullptr; bwagpu_destroy(h); return BWAGPU_ENODEV; }
	if (hipEventCreateWithFlags(&h->ev_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) h->ev_wait = nullptr;
	struct { DevBuf *dst; const DevBuf *from; size_t bytes; } parts[] = {
		{ &h->ibuf->d_bwt, &src->ibuf->d_bwt, (size_t)src->bwt_bytes }, { &h->ibuf->d_sa, &src->ibuf->d_sa, (size_t)src->sa_bytes },
		{ &h->ibuf->d_pac, &src->ibuf->d_pac, (size_t)src->pac_bytes }, { &h->ibuf->d_ctg_off, &src->ibuf->d_ctg_off, (size_t)src->n_seqs * 8 },
		{ &h->ibuf->d_ctg_len, &src->ibuf->d_ctg_len, (size_t)src->n_seqs * 4 }, { &h->ibuf->d_ctg_alt, &src->ibuf->d_ctg_alt, (size_t)src->n_seqs * 4 },
		{ &h->ibuf->d_ptab, &src->ibuf->d_ptab, src->ix.ptab ? ((size_t)1 << (2 * src->ix.ptab_m)) * src->ix.ptab_m * sizeof(uint4) : 0 } };
	for (auto &pt : parts) {
		if (pt.bytes == 0) continue;
		if (pt.dst->ensure(pt.bytes)) { h->err = "hipMalloc failed (index copy)"; bwagpu_destroy(h); return BWAGPU_ENOMEM; }
		if (hipMemcpyPeer(pt.dst->p, device, pt.from->p, src->device, pt.bytes) != hipSuccess) { bwagpu_destroy(h); return BWAGPU_EHIP; }
	}
	h->ix = src->ix;
	h->ix.bwt = h->ibuf->d_bwt.as<uint4>(); h->ix.sa = h->ibuf->d_sa.as<u64>(); h->ix.pac = h->ibuf->d_pac.as<u8>();
	h->ix.ctg_off = h->ibuf->d_ctg_off.as<i64>(); h->ix.ctg_len = h->ibuf->d_ctg_len.as<i32>(); h->ix.ctg_alt = h->ibuf->d_ctg_alt.as<i32>();
	h->ix.ptab = src->ix.ptab ? h->ibuf->d_ptab.as<uint4>() : nullptr;
	h->bwt_blocks = src->bwt_blocks;
	if (src->ix.occ32) { if (int rc = build_occ32(h)) { bwagpu_destroy(h); return rc; } } else { h->ix.occ32 = nullptr; h->ix.occ_sb = nullptr; h->ix.occ_sbx = nullptr; h->ix.occ32_bytes = h->ix.occ_sbx_bytes = 0; }   // (rebuilt here rather than copied)
	h->l_pac = src->l_pac; h->n_seqs = src->n_seqs; h->seq_len = src->seq_len; h->sa_intv = src->sa_intv;
	h->bwt_blocks = src->bwt_blocks; h->bwt_bytes = src->bwt_bytes; h->sa_bytes = src->sa_bytes; h->pac_bytes = src->pac_bytes;
	h->bwt_size = src->bwt_size; h->n_sa = src->n_sa;
	h->h_ctg_off = src->h_ctg_off; h->h_ctg_len = src->h_ctg_len; h->h_ctg_alt = src->h_ctg_alt;
	h->stats_on = src->stats_on; h->taps_on = src->taps_on; h->cigar_filter = src->cigar_filter;
	*out = h;
	return BWAGPU_OK;
}

extern "C" int bwagpu_clone_to_device(bwagpu_t *src, int device, bwagpu_t **out)
{
	if (!src || !out) return BWAGPU_EINVAL;
	int prev = -1;
	if (hipGetDevice(&prev) != hipSuccess) prev = -1;
	const int rc = clone_to_device_impl(src, device, out);
	if (prev >= 0) (void)hipSetDevice(prev);      // the calling thread's current device is left as it was
	return rc;
}

extern "C" int bwagpu_index_buffers(bwagpu_t *h, void **bwt, uint64_t *bwt_bytes, void **sa, uint64_t *sa_bytes, void **pac, uint64_t *pac_bytes)
{
	if (!h || !bwt || !bwt_bytes || !sa || !sa_bytes || !pac || !pac_bytes) return BWAGPU_EINVAL;
	*bwt = h->ibuf->d_bwt.p; *bwt_bytes = h->bwt_bytes; *sa = h->ibuf->d_sa.p; *sa_bytes = h->sa_bytes; *pac = h->ibuf->d_pac.p; *pac_bytes = h->pac_bytes;
	return BWAGPU_OK;
}

// call on a handle created with NULL arrays once its buffers have been filled by the broadcast
extern "C" int bwagpu_index_ready(bwagpu_t *h)
{
	if (!h) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	if (int rc = build_occ32(h)) return rc;
	return build_prefix_tables(h, (int)h->cfg.ptab_m);
}

extern "C" int bwagpu_index_export(const bwagpu_t *h, bwagpu_index_desc_t *d, int64_t *ctg_offset, int32_t *ctg_len, int32_t *ctg_is_alt)
{
	if (!h || !d) return BWAGPU_EINVAL;
	memset(d, 0, sizeof *d);
	d->bwt_size = h->bwt_size; d->primary = h->ix.primary; for (int i = 0; i < 5; ++i) d->L2[i] = h->ix.L2[i];
	d->seq_len = h->seq_len; d->n_sa = h->n_sa; d->sa_intv = h->sa_intv; d->l_pac = h->l_pac; d->n_seqs = h->n_seqs;
	if (ctg_offset) memcpy(ctg_offset, h->h_ctg_off.data(), (size_t)h->n_seqs * 8);
	if (ctg_len) memcpy(ctg_len, h->h_ctg_len.data(), (size_t)h->n_seqs * 4);
	if (ctg_is_alt) memcpy(ctg_is_alt, h->h_ctg_alt.data(), (size_t)h->n_seqs * 4);
	return BWAGPU_OK;
}

extern "C" int bwagpu_index_info(const bwagpu_t *h, int64_t *l_pac, int32_t *n_seqs, uint64_t *seq_len, int *sa_intv)
{
	if (!h) return BWAGPU_EINVAL;
	if (l_pac) *l_pac = h->l_pac;
	if (n_seqs) *n_seqs = h->n_seqs;
	if (seq_len) *seq_len = h->seq_len;
	if (sa_intv) *sa_intv = h->sa_intv;
	return BWAGPU_OK;
}

extern "C" int bwagpu_debug_phase(const bwagpu_t *h) { return h ? h->phase : -1; }
// diagnostic counters of the last batch_run (Counters::prof)
extern "C" int bwagpu_debug_prof(bwagpu_t *h, unsigned long long out[16])
{
	if (!h || !out || !h->d_ctr.p) return BWAGPU_EINVAL;
	Counters c;
	if (hipMemcpy(&c, h->d_ctr.p, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return BWAGPU_EHIP;
	for (int i = 0; i < 16; ++i) out[i] = c.prof[i];
	out[8] = c.n_vr_ovf + c.prof[8];       // long-read batches: tasks of pass 1 that were redone on full-size interval stacks; short-read batches: extensions k_ext_pack answered (the one is 0 where the other counts)
	out[0] = c.n_heavy; out[1] = c.n_p2_tasks;       // short-read batches: reads whose passes 1-2 ran as tasks, and their pass-2 searches
	return BWAGPU_OK;
}

// ... its index look-ups by interval size (Counters::seed_x2)
extern "C" int bwagpu_debug_seed_x2(bwagpu_t *h, unsigned long long out[8])
{
	if (!h || !out || !h->d_ctr.p) return BWAGPU_EINVAL;
	Counters c;
	if (hipMemcpy(&c, h->d_ctr.p, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return BWAGPU_EHIP;
	for (int i = 0; i < 8; ++i) out[i] = c.seed_x2[i];
	return BWAGPU_OK;
}
// ... and the histogram of k_seed's iterations per read (stats runs): out[b] = reads that took [2^(b-1), 2^b) iterations, out[32 + b] = their iterations summed
extern "C" int bwagpu_debug_hist(bwagpu_t *h, unsigned long long out[256])
{
	if (!h || !out || !h->d_ctr.p) return BWAGPU_EINVAL;
	Counters c;
	if (hipMemcpy(&c, h->d_ctr.p, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return BWAGPU_EHIP;
	for (int i = 0; i < 64; ++i) out[i] = c.seed_hist[i];
	for (int i = 0; i < 96; ++i) { out[64 + i] = c.wave_hist[0][i]; out[160 + i] = c.wave_hist[1][i]; }
	return BWAGPU_OK;
}
// ... and of the chaining tiers: out[t * 64 + b] = reads that finished in tier t with 16 b .. 16 b + 15 chains (b = 31: more), out[t * 64 + 32 + b] = with 32 b .. 32 b + 31 seeds
extern "C" int bwagpu_debug_chain_hist(bwagpu_t *h, unsigned long long out[192])
{
	if (!h || !out || !h->d_ctr.p) return BWAGPU_EINVAL;
	Counters c;
	if (hipMemcpy(&c, h->d_ctr.p, sizeof c, hipMemcpyDeviceToHost) != hipSuccess) return BWAGPU_EHIP;
	for (int t = 0; t < 3; ++t) for (int i = 0; i < 32; ++i) { out[t * 64 + i] = c.chain_hist[t][i]; out[t * 64 + 32 + i] = c.chain_seeds[t][i]; }
	return BWAGPU_OK;
}
extern "C" int bwagpu_set_stats(bwagpu_t *h, int enable) { if (!h) return BWAGPU_EINVAL; h->stats_on = enable ? 1 : 0; return BWAGPU_OK; }
extern "C" int bwagpu_set_cigar_filter(bwagpu_t *h, int enable) { if (!h) return BWAGPU_EINVAL; h->cigar_filter = enable ? 1 : 0; return BWAGPU_OK; }
extern "C" int bwagpu_set_taps(bwagpu_t *h, int enable) { if (!h) return BWAGPU_EINVAL; h->taps_on = enable ? 1 : 0; return BWAGPU_OK; }
extern "C" int bwagpu_get_stats(const bwagpu_t *h, bwagpu_stats_t *out) { if (!h || !out) return BWAGPU_EINVAL; *out = h->stats; return BWAGPU_OK; }

// ---- SA densification ---------------------------------------------------------------------------------------
// SA[next(k)] = SA[k] - 1 for next(k) = (k == primary ? 0 : LF(k)) (bwt_sa, bwt.c:91-103, read backwards), and `next` runs through all rows in
// one cycle.  So every row's value follows from ONE walk around that cycle: a lane starts at a row the old sampling holds, steps with LF and
// writes the rows on its way that the new sampling keeps, until it reaches the next row of the old sampling -- seq_len LF steps in all,
// where a walk per new sample (fm_sa) took old_intv - 1 steps on average for each of them (6.8x as many for 32 -> 4).  Row 0 is stored as
// -1 (bwtindex.c:bwt_cal_sa); the walk that starts there counts down from seq_len.  A lane takes its next start when its walk ends, not
// when the wave's longest one does: the loop is one LF step per lane and turn.
__global__ void __launch_bounds__(256) k_densify(DevIndex ix, u64 *out, u64 n_chains, int new_shift, u64 seq_len)
{
	const u64 stride = (u64)gridDim.x * blockDim.x, new_mask = ((u64)1 << new_shift) - 1;
	u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x, k = 0, v = 0;
	auto start = [&]() {
		k = c << ix.sa_shift;
		v = c == 0 ? seq_len : ix.sa[c];
		out[k >> new_shift] = c == 0 ? ~0ull : v;
	};
	if (c < n_chains) start();
	while (c < n_chains) {
		k = k == ix.primary ? 0 : fm_lf(ix, k);
		--v;
		if ((k & ix.sa_mask) == 0) { c += stride; if (c < n_chains) start(); }
		else if ((k & new_mask) == 0) out[k >> new_shift] = v;
	}
}

extern "C" int bwagpu_densify_sa(bwagpu_t *h, int new_intv)
{
	if (!h || new_intv <= 0 || (new_intv & (new_intv - 1)) || new_intv > h->sa_intv) return BWAGPU_EINVAL;
	if (new_intv == h->sa_intv) return BWAGPU_OK;
	if (h->ibuf->refs > 1) { h->err = "densify the SA before cloning the handle"; return BWAGPU_EINVAL; }
	HIPCHK(h, hipSetDevice(h->device));
	int sh = 0; while ((1 << sh) < new_intv) ++sh;
	u64 n_out = (h->seq_len + new_intv) / new_intv;
	DevBuf nb;
	if (nb.ensure(n_out * 8)) { h->err = "hipMalloc failed (dense SA)"; return BWAGPU_ENOMEM; }
	hipLaunchKernelGGL(k_densify, dim3((unsigned)((h->n_sa + 255) / 256 < 8192 ? (h->n_sa + 255) / 256 : 8192)), dim3(256), 0, h->stream, h->ix, nb.as<u64>(), (u64)h->n_sa, sh, (u64)h->seq_len);
	HIPCHK(h, hipGetLastError());
	HIPCHK(h, hipStreamSynchronize(h->stream));
	h->ibuf->d_sa.release();
	h->ibuf->d_sa = nb;
	h->ix.sa = h->ibuf->d_sa.as<u64>(); h->ix.sa_mask = (u64)new_intv - 1; h->ix.sa_shift = sh; h->sa_intv = new_intv;
	h->n_sa = n_out; h->sa_bytes = n_out * 8;
	return BWAGPU_OK;
}

static int sc_max_all(const bwagpu_opt_t *opt) { int m = 1; for (int k = 0; k < 25; ++k) if (opt->mat[k] > m) m = opt->mat[k]; return m; }

// ---- batches -------------------------------------------------------------------------------------------------
static const int BLOCK = 256;
static const int MAX_RESIDENT_THREADS = 256 * 2048;   // 256 CUs x 32 waves x 64 lanes
static const int WAVE_EXT_MAX_LEN = 1100;
static const int SEED_LDS_ENT = 10;                     // 10 x 16 B x 256 lanes = 40 KiB of LDS per block -> 4 blocks (16 waves) per CU; measured best of {4,7,10,15}               // 4 waves x (8+5) B/column must fit the 64 KiB dynamic-LDS limit

// first guess of a batch's arena sizes from its shape (n_reads, n_bases, max_len); grown on overflow
static void size_arenas(bwagpu_t *h)
{
	const int n = h->n_reads;
	i64 nb = h->n_bases > 1024 ? h->n_bases : 1024;
	h->slot_cap = nb / 3 + 4096;     // (a 3.1 Gbp repeat-rich genome needs ~0.26 slots and ~0.15 region records per base)
	h->node_cap = h->slot_cap / 4 + 2 * (i64)n + 64;
	h->reg_cap = nb / 5 + 4096;
	// capacity of one read's interval list: reads keep ~10-30 intervals whatever their length class; grown x4 on overflow
	// (256 at least: with 64, one read in ten thousand of a repeat-rich genome -- a 150 bp read inside a tandem array leaves well over a hundred
	// intervals -- made the first batch of every handle run twice, profiles/r04_e2e_split.log)
	h->mem_cap = h->max_len / 3 < 256 ? 256 : h->max_len / 3;
	// what earlier batches -- of this handle or of another handle on the same index -- turned out to need carries over
	{
		std::lock_guard<std::mutex> l(h->ibuf->m);
		if (h->ibuf->need_slot > h->need_slot) h->need_slot = h->ibuf->need_slot;
		if (h->ibuf->need_node > h->need_node) h->need_node = h->ibuf->need_node;
		if (h->ibuf->need_reg > h->need_reg) h->need_reg = h->ibuf->need_reg;
		if (h->ibuf->need_mem > h->need_mem) h->need_mem = h->ibuf->need_mem;
	}
	if ((i64)(h->need_slot * nb) > h->slot_cap) h->slot_cap = (i64)(h->need_slot * nb);
	if ((i64)(h->need_node * nb) > h->node_cap) h->node_cap = (i64)(h->need_node * nb);
	if ((i64)(h->need_reg * nb) > h->reg_cap) h->reg_cap = (i64)(h->need_reg * nb);
	if (h->need_mem > h->mem_cap) h->mem_cap = h->need_mem;
	if (h->cfg.mem_cap > 0) h->mem_cap = (int)h->cfg.mem_cap;   // test hook: force the overflow/retry path
}

// resident lanes of the lane-per-read kernels: enough to fill the chip, bounded by the seeding scratch budget (2 interval stacks per lane)
static int resident_threads(const bwagpu_t *h)
{
	const int BLOCK_ = 256;
	size_t per_lane = (size_t)(h->max_len + 1 + PTAB_MAX) * sizeof(BiIntv) + (size_t)2 * (h->max_len + 2) * 4;
	size_t budget = (size_t)12 << 30;
	i64 max_thr = (i64)(budget / per_lane);
	if (max_thr > 256 * 2048) max_thr = 256 * 2048;
	if (max_thr < BLOCK_) max_thr = BLOCK_;
	int n_threads = (int)(((i64)h->n_reads + BLOCK_ - 1) / BLOCK_ * BLOCK_);
	if (n_threads > max_thr) n_threads = (int)(max_thr / BLOCK_ * BLOCK_);
	return n_threads;
}

extern "C" int bwagpu_batch_upload(bwagpu_t *h, int n, const uint8_t *seqs, const int64_t *off)
{
	if (!h || n < 0 || (n > 0 && (!seqs || !off))) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	h->have_batch = false; h->ran = false; h->packed_tot = -1; h->cig_ext_n = -1; h->phase = 10;
	h->n_reads = n; h->max_len = 0; h->n_bases = n ? off[n] - off[0] : 0;
	if (n && off[0] != 0) return BWAGPU_EINVAL;
	for (int i = 0; i < n; ++i) {
		i64 l = off[i + 1] - off[i];
		if (l < 0 || l > 0x3fffffff) return BWAGPU_EINVAL;
		if (l > h->max_len) h->max_len = (int)l;
	}
	h->h_off.assign(off, off + n + 1);
	const u64 n_words = ((u64)h->n_bases + 15) / 16;
	h->phase = 11;
	if (h->d_seq.ensure((size_t)h->n_bases + 16) || h->d_seq_nib.ensure((size_t)(n_words + 1) * 8) || h->d_off.ensure((size_t)(n + 1) * 8)) { h->err = "hipMalloc failed (reads)"; return BWAGPU_ENOMEM; }
	if (n) {
		HIPCHK(h, hipMemcpyAsync(h->d_seq.p, seqs, (size_t)h->n_bases, hipMemcpyHostToDevice, h->stream));
		HIPCHK(h, hipMemcpyAsync(h->d_off.p, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, h->stream));
		h->phase = 12;
		Batch P = {}; P.seq = h->d_seq.as<u8>(); P.seq_nib = h->d_seq_nib.as<u64>();
		u64 pb = (n_words + BLOCK - 1) / BLOCK;
		hipLaunchKernelGGL(k_pack_reads, dim3((unsigned)(pb < 65536 ? pb : 65536)), dim3(BLOCK), 0, h->stream, P, n_words);
		HIPCHK(h, hipGetLastError());
		// reads of up to 256 bases: a 2-bit copy per read for the seeding kernel's LDS
		h->rd_words = (h->max_len <= 256 && h->cfg.seed_rd_lds != 0) ? (((h->max_len + 15) / 16 + 3) & ~3) : 0;
		if (h->rd_words) {
			if (h->d_seq_2b.ensure((size_t)n * h->rd_words * 4 + 64) || h->d_seq_flags.ensure((size_t)n + 16)) { h->err = "hipMalloc failed (reads)"; return BWAGPU_ENOMEM; }
			HIPCHK(h, hipMemsetAsync(h->d_seq_flags.p, 0, (size_t)n, h->stream));
			P.off = h->d_off.as<i64>(); P.n_reads = n; P.rd_words = h->rd_words; P.seq_2b = h->d_seq_2b.as<u32>(); P.seq_flags = h->d_seq_flags.as<u8>();
			const u64 tb = ((u64)n * h->rd_words + BLOCK - 1) / BLOCK;
			hipLaunchKernelGGL(k_pack_reads2b, dim3((unsigned)(tb < 65536 ? tb : 65536)), dim3(BLOCK), 0, h->stream, P);
			HIPCHK(h, hipGetLastError());
		}
		HIPCHK(h, wait_stream(h));
	}
	size_arenas(h);
	h->have_batch = true;
	return BWAGPU_OK;
}

// Waves that own a DP scratch region (dp_h / dp_e).  The lane-per-read kernels need one per wave of the seeding grid; the wave-per-read
// kernels of long-read batches (k_seedsw_wave, k_dedup_wave) should fill the chip whatever that grid is -- for 10 kb reads the seeding
// scratch limits it to a few dozen waves, which left 2000 reads to 32 waves (measured: 3.1 of a batch's 4.8 s) -- so they get up to 1024.
static int dp_wave_count(const bwagpu_t *h, int n_threads)
{
	int w = (n_threads + 63) / 64;
	if (h->max_len > WAVE_EXT_MAX_LEN) { int want = h->n_reads < 1024 ? h->n_reads : 1024; if (want > w) w = want; }
	return w;
}

static int alloc_batch(bwagpu_t *h, int n_threads, int seed_lanes)
{
	int n = h->n_reads; size_t sc = (size_t)h->slot_cap + 8;   // +8: chunked readers may touch a few slots past the last read's range
	int bad = 0;
	bad |= h->d_ctr.ensure(sizeof(Counters));
	bad |= h->d_tmp_intv.ensure((size_t)(seed_lanes > n_threads ? seed_lanes : n_threads) * (h->max_len + 1 + PTAB_MAX) * sizeof(BiIntv));
	bad |= h->d_intv_n.ensure((size_t)n * 4 + 16); bad |= h->d_intv_off.ensure((size_t)n * 8 + 16);
	bad |= h->d_intv.ensure(((size_t)n * h->mem_cap + 16) * sizeof(Intv3));
	bad |= h->d_seed_n.ensure((size_t)n * 4 + 16); bad |= h->d_seed_off.ensure((size_t)n * 8 + 16);
	bad |= h->d_slot_pos.ensure(sc * 8); bad |= h->d_slot_qbeg.ensure(sc * 4); bad |= h->d_slot_len.ensure(sc * 4); bad |= h->d_slot_rid.ensure(sc * 4); bad |= h->d_slot_blob.ensure(sc * SLOT_BLOB_BYTES);
	bad |= h->d_chain_n.ensure((size_t)n * 4 + 16); bad |= h->d_node_off.ensure((size_t)n * 8 + 16);
	bad |= h->d_nodes.ensure((size_t)h->node_cap * BT_NODE_INTS * 4);
	bad |= h->d_reg_off.ensure((size_t)n * 8 + 16); bad |= h->d_reg_cap_r.ensure((size_t)n * 4 + 16);
	bad |= h->d_reg_n_raw.ensure((size_t)n * 4 + 16); bad |= h->d_reg_n.ensure((size_t)n * 4 + 16);
	bad |= h->d_regs.ensure((size_t)h->reg_cap * sizeof(bwagpu_alnreg_t));
	if (h->taps_on) bad |= h->d_regs_raw.ensure((size_t)h->reg_cap * sizeof(bwagpu_alnreg_t));
	int n_waves = dp_wave_count(h, n_threads);
	bad |= h->d_dp_h.ensure((size_t)n_waves * (h->max_len + 2) * DPS * 4);
	bad |= h->d_dp_e.ensure((size_t)n_waves * (h->max_len + 2) * DPS * 4);
	bad |= h->d_minhsp.ensure((size_t)(h->max_len + 2) * 4);
	bad |= h->d_order.ensure((size_t)n * 4 + 16); bad |= h->d_bin_cnt.ensure(2 * ORDER_BINS * 4); bad |= h->d_seed_w.ensure((size_t)n * 4 + 16); bad |= h->d_seed_order.ensure((size_t)n * 4 + 16);
	if (bad) { h->err = "hipMalloc failed (batch arenas)"; return BWAGPU_ENOMEM; }
	return 0;
}

// heavy-first processing order of the batch by a per-read weight array (device)
static int order_reads(bwagpu_t *h, const Batch &B, const i32 *weight)
{
	HIPCHK(h, hipMemsetAsync(B.bin_cnt, 0, 2 * ORDER_BINS * 4, h->stream));
	int nb = (B.n_reads + BLOCK - 1) / BLOCK; if (nb > 1024) nb = 1024;
	int chunk = (B.n_reads + nb - 1) / nb;
	hipLaunchKernelGGL(k_order_count, dim3(nb), dim3(BLOCK), 0, h->stream, B, weight);
	hipLaunchKernelGGL(k_order_scan, dim3(1), dim3(64), 0, h->stream, B);
	hipLaunchKernelGGL(k_order_fill, dim3(nb), dim3(BLOCK), 0, h->stream, B, weight, chunk);
	return 0;
}

static void release_reserved(bwagpu_t *h) { for (void *&p : h->reserved) { if (p) bwagpu_free(p); p = nullptr; } }

// Allocate now what a batch of this shape will need -- read arrays, arenas, scratch, packed results: device buffers are only ever grown, so
// the first real batch of the handle finds them in place instead of spending ~0.4 s in hipMalloc inside the pipeline.
extern "C" int bwagpu_batch_reserve(bwagpu_t *h, int n_reads, int64_t n_bases, int max_len)
{
	if (!h || n_reads <= 0 || n_bases <= 0 || max_len <= 0 || max_len > 0x3fffffff) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	const int n0 = h->n_reads, m0 = h->max_len; const i64 b0 = h->n_bases; const bool have0 = h->have_batch, ran0 = h->ran;
	const i64 sc0 = h->slot_cap, nc0 = h->node_cap, rc0 = h->reg_cap; const int mc0 = h->mem_cap;
	h->n_reads = n_reads; h->n_bases = n_bases; h->max_len = max_len;
	size_arenas(h);
	int bad = alloc_batch(h, resident_threads(h), 0) != 0;
	const u64 n_words = ((u64)n_bases + 15) / 16;
	const int rdw = max_len <= 256 ? (((max_len + 15) / 16 + 3) & ~3) : 0;
	bad |= h->d_seq.ensure((size_t)n_bases + 16); bad |= h->d_seq_nib.ensure((size_t)(n_words + 1) * 8); bad |= h->d_off.ensure((size_t)(n_reads + 1) * 8);
	if (rdw) { bad |= h->d_seq_2b.ensure((size_t)n_reads * rdw * 4 + 64); bad |= h->d_seq_flags.ensure((size_t)n_reads + 16); }
	const i64 tot = (i64)n_reads * 4;        // (packed results: ~3.2 regions per read on a repeat-rich genome)
	bad |= h->d_pack_off.ensure((size_t)n_reads * 8); bad |= h->d_regs_packed.ensure((size_t)tot * sizeof(bwagpu_alnreg_t)); bad |= h->d_pack_read.ensure((size_t)tot * 4);
	bad |= h->d_cigs.ensure((size_t)tot * sizeof(bwagpu_cigar_t)); bad |= h->d_cig_ext.ensure((size_t)(tot * 4 + 65536) * 4);
	// ... and the page-locked blocks its results will be copied into (regions, CIGAR records, operation array: ~0.4 GB per 667 k reads; page-locking costs
	// more than the copy -- ~0.2 ms per MB -- and the pool keeps what it is given back, so the handle's first batch finds them)
	if (h->cfg.reserve_results && !g_dry_sum) {
		// (held by the handle until its first download asks for blocks: handles that reserve one after another would otherwise all be handed the same three idle blocks)
		release_reserved(h);
		h->reserved[0] = result_alloc((size_t)tot * sizeof(bwagpu_alnreg_t)); h->reserved[1] = result_alloc((size_t)tot * sizeof(bwagpu_cigar_t)); h->reserved[2] = result_alloc((size_t)tot * 2 * 4);
	}
	h->n_reads = n0; h->n_bases = b0; h->max_len = m0; h->have_batch = have0; h->ran = ran0;
	h->slot_cap = sc0; h->node_cap = nc0; h->reg_cap = rc0; h->mem_cap = mc0;
	if (bad) { h->err = "hipMalloc failed (reserve)"; return BWAGPU_ENOMEM; }
	return BWAGPU_OK;
}

// Device memory a handle's buffers would GROW by for a batch of this shape (bytes; what bwagpu_batch_reserve would allocate now), and the device's
// free / total memory: what a caller needs to decide how much HBM it can spend on a denser suffix array before the batches arrive.
extern "C" int64_t bwagpu_batch_footprint(bwagpu_t *h, int n_reads, int64_t n_bases, int max_len)
{
	if (!h || n_reads <= 0 || n_bases <= 0 || max_len <= 0 || max_len > 0x3fffffff) return -1;
	size_t sum = 0;
	g_dry_sum = &sum;
	const int rc = bwagpu_batch_reserve(h, n_reads, n_bases, max_len);
	g_dry_sum = nullptr;
	return rc == BWAGPU_OK ? (int64_t)sum : -1;
}
extern "C" int bwagpu_mem_info(bwagpu_t *h, uint64_t *free_bytes, uint64_t *total_bytes)
{
	if (!h) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	size_t f = 0, t = 0;
	HIPCHK(h, hipMemGetInfo(&f, &t));
	if (free_bytes) *free_bytes = f;
	if (total_bytes) *total_bytes = t;
	return BWAGPU_OK;
}

namespace { struct BusyGuard { std::atomic<int> &c; int others; explicit BusyGuard(std::atomic<int> &c_) : c(c_), others(c_.fetch_add(1)) {} ~BusyGuard() { --c; } }; }
extern "C" int bwagpu_batch_run(bwagpu_t *h, const bwagpu_opt_t *opt)
{
	if (!h || !opt || !h->have_batch) return BWAGPU_EINVAL;
	const BusyGuard busy(h->ibuf->busy);
	if (opt->e_del <= 0 || opt->e_ins <= 0 || opt->max_occ <= 0 || opt->min_seed_len <= 0 || opt->w < 0) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	memset(&h->stats, 0, sizeof h->stats);
	h->stats.n_reads = h->n_reads; h->stats.n_bases = h->n_bases;
	h->ran = false; h->packed_tot = -1; h->cig_ext_n = -1;      // regions packed by an earlier download (and their CIGAR operations) belong to the previous run
	if (h->n_reads == 0) { h->ran = true; return BWAGPU_OK; }
	int n = h->n_reads;
	// resident lanes: enough to fill the chip, bounded by the seeding scratch budget (2 interval stacks per lane)
	const int n_threads = resident_threads(h);
	// mem_flt_chained_seeds thresholds per read length (bwamem.c:626-628); log() stays on the host
	std::vector<i32> minhsp(h->max_len + 2, -1);
	bool any_seedsw = false;
	for (int l = 1; l <= h->max_len; ++l) {
		double min_l = opt->min_chain_weight ? 1.1f * opt->min_chain_weight : 5.5f * log((double)l);
		if (!(min_l > 0.05f * l)) { minhsp[l] = (int)(opt->a * min_l + .499); any_seedsw = true; }
	}
	const BwagpuConfig &cfg = h->cfg;            // (no batch call reads the environment: bwagpu_config.h)
	const bool dbg_sync = cfg.debug_sync != 0;   // diagnostics: wait and report after every stage
	// Long-read batches (a read beyond the short-read extension kernel's columns) default to the kernel forms BENCH_r03's `variants` measured
	// fastest for them, identical regions: seeding by chunks with one memory round trip per iteration, workgroup-per-read interval sort, LDS
	// seed re-scoring, four columns per lane in the patch alignments.  An explicit option (>= 0) overrides either way.
	const bool long_batch = h->max_len > WAVE_EXT_MAX_LEN;
	// Option share (percent; short-read batches): the persistent kernels of the hot path launch with that share of the workgroups that would fill
	// the chip.  A kernel whose workgroups live until its work runs out holds every slot it got: launched to fill the chip, the kernels of the
	// batches in flight take turns; launched for a share, kernels of different batches -- the memory-bound seeding of one, the issue-bound
	// extension of another -- run side by side.
	// auto (-1): half the chip per kernel when at least three handles share this index -- three batches in flight, the way `bwa-amd mem` and the bench's timed loop
	// drive a device -- and another of them has kernels in flight as this run starts; else all of it (a lone handle; the first batch of a pipeline, which has
	// the chip to itself while the reader is still parsing the second: 151 -> ~90 ms, `profiles/r05_e2e_reserve_results.log`).  Measured with three batches in flight (profiles/r05_share_ab.log): 108.3 ms per step at 100, 105.2-105.7 at 50 (105.1 at 40,
	// 107.1 at 60), i.e. -2.7 %, for +10 % on a batch that has the chip to itself (131.8 -> 144.8 ms) -- round 4 measured nothing; the kernels' balance has moved.
	const long long share_pct = cfg.share >= 0 ? cfg.share : (h->ibuf->refs.load() >= 3 && busy.others > 0 ? 50 : 100);
	auto share = [&](long long g) { if (long_batch || share_pct >= 100 || share_pct <= 0) return g; const long long v = g * share_pct / 100; return v < 1 ? 1ll : v; };
	auto pick = [&](long long v, long long dflt_long) { return v >= 0 ? v : (long_batch ? dflt_long : 0); };
	// Pass 1 of long-read batches as independent tasks (option seed_tasks; dev_seed.h, k_seed's LR): one task per read and min_seed_len-th
	// position.  The host only says where each read's tasks begin; a task finds its read by bisection.
	bool seed_tasks = pick(cfg.seed_tasks, 1) != 0;
	int n_vreads = 0, task_lanes = 0;
	const int TASK_STACK_CAP = PTAB_MAX + (cfg.seed_task_stack > 0 ? (int)((cfg.seed_task_stack + 1) / 2) : 128);          // BiIntv-sized entries of a task lane's spill area (default: 256 packed entries; a full stack hands the task to the second launch)
	if (!long_batch || h->max_len >= 65536 || h->seq_len >= ((u64)1 << 37) || h->ix.occ32 == nullptr || h->ix.ptab == nullptr || h->rd_words != 0 || opt->min_seed_len < 1 || cfg.seed_pass3_inline) seed_tasks = false;
	if (seed_tasks) {
		std::vector<i32> first((size_t)n + 1);
		i64 tot = 0;
		for (int r = 0; r < n; ++r) { first[r] = (i32)tot; tot += (h->h_off[r + 1] - h->h_off[r] + opt->min_seed_len - 1) / opt->min_seed_len; }
		first[n] = (i32)tot;
		if (tot <= 0 || tot > 0x7fffffff) seed_tasks = false;
		else {
			n_vreads = (int)tot;
			task_lanes = (int)((tot + BLOCK - 1) / BLOCK * BLOCK); if (task_lanes > 256 * 3 * BLOCK) task_lanes = 256 * 3 * BLOCK;   // persistent lanes (three workgroups per CU), tasks drawn from a counter
			if (h->d_vr_tab.ensure(((size_t)n + 1) * 4) || h->d_vr_ovf.ensure((size_t)n_vreads * 4 + 16) || h->d_intv_n3.ensure((size_t)n * 4 + 16)) { h->err = "hipMalloc failed (seeding tasks)"; return BWAGPU_ENOMEM; }
			HIPCHK(h, hipMemcpyAsync(h->d_vr_tab.p, first.data(), first.size() * 4, hipMemcpyHostToDevice, h->stream));   // (pageable source: staged before the call returns)
		}
	}
	if (dbg_sync) fprintf(stderr, "[bwagpu] pass 1 by tasks: %d tasks on %d lanes (max_len %d, step %d)\n", n_vreads, task_lanes, h->max_len, opt->min_seed_len);
	// Short-read batches: a read on which a lane of the lane-per-read kernel has spent more than option seed_budget iterations (default 8192: 0.15 % of
	// the bench's reads -- and the whole of that kernel's critical path; measured over budgets 2048..8192: profiles/r04_seed_budget_ab.jsonl) is given up there and seeded by the task kernels afterwards (dev_seed.h, LR).
	// Round 6 (profiles/r06_seed_budget.md): alone on the chip the stage is shortest at ~6 k iterations (50.5 ms against 55.5 at 8192, 59.8 at 12288: the main
	// kernel's tail of nearly empty waves ends sooner); with other batches' kernels on the chip that tail costs nothing -- they fill it -- and the step is best
	// with fewer reads handed to the chip-filling task kernels (86.3-87.6 ms at 12288 against 87.1-89.1 at 8192).  So auto follows the same signal as `share`.
	const int seed_budget = (int)(cfg.seed_budget < 0 ? (share_pct < 100 ? 12288 : 6144) : cfg.seed_budget);
	const int heavy_tpr = opt->min_seed_len > 0 ? (h->max_len + opt->min_seed_len - 1) / opt->min_seed_len : 0;
	// The one-trip seeding kernels (MRG = 2) address the index through buffer descriptors: 32-bit byte offsets into tables of less than 4 GiB.  The heavy
	// reads' task kernels exist in that form only, so an index beyond a descriptor's reach (occ32 above 4 GiB: a genome of more than ~4.3 Gbp; a deep
	// prefix table) keeps every read on the lane-per-read kernel, without a budget (option idx_desc_max_mb lowers the limit: test hook).
	const u64 desc_max = cfg.idx_desc_max_mb > 0 && ((u64)cfg.idx_desc_max_mb << 20) < BUF_MAX_BYTES ? (u64)cfg.idx_desc_max_mb << 20 : BUF_MAX_BYTES;
	const bool idx_in_desc = h->ix.occ32 != nullptr && h->ix.ptab != nullptr && h->ix.occ32_bytes <= desc_max && h->ix.ptab_bytes <= desc_max;
	const bool heavy_tasks = !long_batch && h->rd_words != 0 && seed_budget > 0 && heavy_tpr > 0 && idx_in_desc &&
							 h->seq_len < ((u64)1 << 37) && !cfg.seed_pass3_inline && opt->max_mem_intv > 0;
	const int heavy_lanes = heavy_tasks ? (int)(((i64)n * heavy_tpr + BLOCK - 1) / BLOCK * BLOCK < 256 * 3 * BLOCK ? ((i64)n * heavy_tpr + BLOCK - 1) / BLOCK * BLOCK : 256 * 3 * BLOCK) : 0;
	if (heavy_tasks && (h->d_intv_n3.ensure((size_t)n * 4 + 16) || h->d_heavy.ensure((size_t)n * 4 + 16))) { h->err = "hipMalloc failed (seeding tasks)"; return BWAGPU_ENOMEM; }
	// k_dedup's list of the reads it leaves to the wave-per-read kernel: the seeding kernels' list of heavy reads, free again by then
	const int dd_heavy_min = long_batch || cfg.dedup_wave || h->seq_len >= ((u64)1 << 47) || h->max_len >= (1 << 16) ? 0 : (int)(cfg.dedup_heavy < 0 ? 3 : cfg.dedup_heavy);     // (the bounds: DdKey, dev_dedupp.h)
	struct { int rc, q_cap, cap_m, cap_b; size_t lds0; long long blk_m, blk_b; } dd = { 0, 0, 0, 0, 0, 0, 0 };
	if (dd_heavy_min > 0) {
		int need = 8 * opt->w + 4 + 128; if (need < h->max_len / 4 + 132) need = h->max_len / 4 + 132;
		dd.rc = 256; while (dd.rc < need && dd.rc < 4096) dd.rc <<= 1;
		dd.q_cap = (h->max_len + 15) & ~15;
		if (8 * dd.rc + 32 + dd.q_cap > 65536) dd.q_cap = 0;
		dd.lds0 = (size_t)8 * dd.rc + 32 + dd.q_cap;
		const long long room = (65536 - (long long)dd.lds0) / (long long)DDP_LDS_PER_REG;       // regions a workgroup's LDS holds next to the ring
		dd.cap_m = (int)(cfg.dedup_stage < 0 ? 128 : cfg.dedup_stage); if (dd.cap_m > room) dd.cap_m = (int)(room > 0 ? room : 0);
		dd.cap_b = (int)(cfg.dedup_big < 0 ? room : cfg.dedup_big); if (dd.cap_b > room) dd.cap_b = (int)(room > 0 ? room : 0);
		const long long wpc_m = (160 * 1024) / (long long)(dd.lds0 + DDW_PAR_BYTES(dd.cap_m)), wpc_b = (160 * 1024) / (long long)(dd.lds0 + DDW_PAR_BYTES(dd.cap_b));
		dd.blk_m = share(256 * (wpc_m > 12 ? 12 : wpc_m)); dd.blk_b = share(256 * (wpc_b > 12 ? 12 : wpc_b));
		const long long dpw = dp_wave_count(h, n_threads), cap = dpw > 0 ? dpw : 1;       // (dp_h / dp_e hold one scratch region per wave)
		if (dd.blk_m > cap) dd.blk_m = cap; if (dd.blk_b > cap) dd.blk_b = cap;
		if (dd.blk_m > n) dd.blk_m = n; if (dd.blk_b > n) dd.blk_b = n;
		if (h->d_heavy.ensure((size_t)n * 4 + 16) || h->d_dd_tmp.ensure(((size_t)dd.blk_m * dd.cap_m + (size_t)dd.blk_b * dd.cap_b) * sizeof(bwagpu_alnreg_t) + 16)) { h->err = "hipMalloc failed (de-duplication lists)"; return BWAGPU_ENOMEM; }
	}
	for (int attempt = 0; attempt < 12; ++attempt) {
		h->phase = 20 + attempt * 100;
		int rc = alloc_batch(h, n_threads, seed_tasks ? (int)(((size_t)task_lanes * TASK_STACK_CAP + (size_t)(h->max_len + PTAB_MAX)) / (size_t)(h->max_len + 1 + PTAB_MAX))   // (the tasks' small spill areas, in units of a full-size one)
											: heavy_lanes);      // (the heavy reads' task lanes use the spill area after the lane-per-read kernel)
		if (rc) return rc;
		HIPCHK(h, hipMemcpyAsync(h->d_minhsp.p, minhsp.data(), minhsp.size() * 4, hipMemcpyHostToDevice, h->stream));
		HIPCHK(h, hipMemsetAsync(h->d_ctr.p, 0, sizeof(Counters), h->stream));
		Batch B; memset(&B, 0, sizeof B);
		B.n_reads = n; B.max_len = h->max_len; B.stats = h->stats_on;
		B.seq = h->d_seq.as<u8>(); B.seq_nib = h->d_seq_nib.as<u64>(); B.off = h->d_off.as<i64>(); B.ctr = h->d_ctr.as<Counters>();
		B.tmp_intv = h->d_tmp_intv.as<BiIntv>(); B.mem_cap = h->mem_cap;
		B.rd_words = h->rd_words; B.seq_2b = h->d_seq_2b.as<u32>(); B.seq_flags = h->d_seq_flags.as<u8>();
		// LDS per lane: 160 bytes at four blocks per CU -- the read's 2-bit copy first, interval-stack entries with the rest
		const int lane_lds = 160, ent_max = SEED_LDS_ENT;
		const int lds_ent_dflt = h->rd_words ? ((lane_lds - 4 * h->rd_words) / 16 < ent_max ? (lane_lds - 4 * h->rd_words) / 16 : ent_max) : ent_max;
		B.seed_lds_ent = (h->seq_len < ((u64)1 << 37) && h->max_len < 65536) ? (cfg.seed_lds_ent >= 0 ? (int)cfg.seed_lds_ent : lds_ent_dflt) : 0;
		if (((size_t)(B.seed_lds_ent ? B.seed_lds_ent : 1) * 16 + (size_t)B.rd_words * 4) * BLOCK > 65536) B.rd_words = 0;   // (an LDS_ENT override too large for both)
		B.intv_n = h->d_intv_n.as<i32>(); B.intv_off = h->d_intv_off.as<i64>(); B.intv = h->d_intv.as<Intv3>();
		B.seed_n = h->d_seed_n.as<i32>(); B.seed_off = h->d_seed_off.as<i64>(); B.slot_cap = h->slot_cap;
		B.slot_pos = h->d_slot_pos.as<u64>(); B.slot_qbeg = h->d_slot_qbeg.as<i32>(); B.slot_len = h->d_slot_len.as<i32>(); B.slot_rid = h->d_slot_rid.as<i32>(); B.slot_blob = h->d_slot_blob.as<u8>();
		B.chain_n = h->d_chain_n.as<i32>(); B.node_off = h->d_node_off.as<i64>(); B.nodes = h->d_nodes.as<i32>(); B.node_cap = h->node_cap;
		B.reg_off = h->d_reg_off.as<i64>(); B.reg_cap_r = h->d_reg_cap_r.as<i32>(); B.reg_n_raw = h->d_reg_n_raw.as<i32>(); B.reg_n = h->d_reg_n.as<i32>();
		B.regs = h->d_regs.as<bwagpu_alnreg_t>(); B.reg_cap = h->reg_cap;
		B.regs_raw = h->taps_on ? h->d_regs_raw.as<bwagpu_alnreg_t>() : nullptr;
		B.dp_h = h->d_dp_h.as<i32>(); B.dp_e = h->d_dp_e.as<i32>(); B.dp_waves = dp_wave_count(h, n_threads);
		B.seedsw_minhsp = h->d_minhsp.as<i32>();
		B.order = h->d_order.as<i32>(); B.bin_cnt = h->d_bin_cnt.as<u32>(); B.seed_w = h->d_seed_w.as<i32>(); B.seed_order = nullptr;
		B.seed_prio = cfg.seed_prio != 0;
		B.ext_blk = cfg.ext_blk != 0;
		B.seed_no_virt = cfg.seed_no_virt != 0;
		B.seed_pass3_inline = cfg.seed_pass3_inline != 0;
		B.task_step = opt->min_seed_len; B.n_vreads = n_vreads; B.vr_ovf_run = 0; B.vr_room = 0; B.seed_stack_cap = 0; B.intv_n3 = nullptr;
		if (seed_tasks) { B.vr_first = h->d_vr_tab.as<i32>(); B.vr_ovf = h->d_vr_ovf.as<i32>(); B.intv_n3 = h->d_intv_n3.as<i32>(); }
		B.task_tpr = 0; B.p2_tasks = nullptr; B.p2_cap = 0; B.heavy_list = nullptr; B.seed_budget = 0;
		if (heavy_tasks) {
			const i64 p2_cap = cfg.seed_p2_cap > 0 ? cfg.seed_p2_cap * (i64)h->p2_factor : (i64)((double)n * .5 * h->p2_factor) + 4096;      // (grown fourfold when a batch overflows it)
			B.heavy_list = h->d_heavy.as<i32>(); B.seed_budget = seed_budget;
			if (h->d_p2_tasks.ensure((size_t)p2_cap * 8)) { h->err = "hipMalloc failed (seeding tasks)"; return BWAGPU_ENOMEM; }
			B.intv_n3 = h->d_intv_n3.as<i32>(); B.p2_tasks = h->d_p2_tasks.as<i64>(); B.p2_cap = p2_cap;
		}
		// memory round trips per iteration of the seeding kernels (dev_seed.h, k_seed's MRG): 0 = as compiled, 2 = table entries, whole index
		// blocks and the next interval-stack entry in one trip (any other non-zero value selects 2 as well)
		int seed_mrg = (cfg.seed_mrg >= 0 ? cfg.seed_mrg : 2) ? 2 : 0;      // (auto: 2 for every batch since round 4 -- 2 % faster in step time for short reads as well, every time it was measured)
		if (!idx_in_desc) seed_mrg = 0;   // (its loads address 32-bit offsets into buffers of < 4 GiB: an index beyond that keeps the plain kernels, see idx_in_desc above)
		B.seq_nib_bytes = (((u64)h->n_bases + 15) / 16) * 8;
		dim3 grid(n_threads / BLOCK), block(BLOCK);
		HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "start", hipGetErrorString(e_)); }
		if (!B.seed_pass3_inline) {   // pass 3 first (cheap), then passes 1-2 on the reads ordered by the repetitiveness it measured
			if (h->ix.occ32 != nullptr && seed_mrg) hipLaunchKernelGGL((k_seed3<1, true>), grid, block, 0, h->stream, h->ix, *opt, B);
			else if (h->ix.occ32 != nullptr) hipLaunchKernelGGL(k_seed3<1>, grid, block, 0, h->stream, h->ix, *opt, B);
			else hipLaunchKernelGGL(k_seed3<0>, grid, block, 0, h->stream, h->ix, *opt, B);
			if (!cfg.seed_input_order) {
				i32 *keep = B.order; B.order = h->d_seed_order.as<i32>();
				if (int rc2 = order_reads(h, B, B.seed_w)) return rc2;
				B.seed_order = B.order; B.order = keep;
			}
		}
		const size_t seed_lds = ((size_t)(B.seed_lds_ent ? B.seed_lds_ent : 1) * sizeof(uint4) + (size_t)B.rd_words * 4) * BLOCK;
		dim3 sgrid = grid;                    // (option seed_grid, measurements: fewer resident workgroups of the seeding kernel; it stays within 5 % down to two per CU -- it is bound by memory requests, not by waves -- but step time with three batches in flight did not move either, nor did making the batches' seeding kernels take turns)
		if (cfg.seed_grid > 0 && (unsigned long long)cfg.seed_grid < grid.x) sgrid = dim3((unsigned)cfg.seed_grid);
		else if (!long_batch) sgrid = dim3((unsigned)share(grid.x));
		// (instances: with/without the LDS copy of the reads, the work counters -- which cost registers --, the two index layouts, one or several trips per iteration)
#define SEED_LAUNCH(RD_, ST_, B_, O_, M_) hipLaunchKernelGGL((k_seed<RD_, ST_, B_, O_, M_>), sgrid, block, seed_lds, h->stream, h->ix, *opt, B)
#define SEED_LAUNCH_B(RD_, ST_) do { if (blk == 1 && mrg == 2) SEED_LAUNCH(RD_, ST_, 1, (RD_ ? 4 : 3), 2);   /* (no LDS copy of the reads: long reads, few lanes -- registers instead of spills) */ \
		else if (blk == 1) SEED_LAUNCH(RD_, ST_, 1, 4, 0); else SEED_LAUNCH(RD_, ST_, 0, 4, 0); } while (0)
		{
			const bool rd = B.rd_words != 0, st = B.stats != 0;
			const int blk = h->ix.occ32 != nullptr ? 1 : 0;
			const int mrg = seed_mrg;
			if (seed_tasks && blk == 1 && !rd) {
				// long reads: pass 1 as independent tasks on small interval stacks; the few tasks whose stack filled up once more, on full-size
				// stacks (its work list is known on the device only: the launch is sized for the lane-per-read kernel and usually finds nothing);
				// then the lane-per-read kernel from pass 2 on
				Batch BA = B; BA.seed_order = nullptr; BA.seed_stack_cap = TASK_STACK_CAP; BA.vr_room = B.seed_lds_ent + 2 * (TASK_STACK_CAP - PTAB_MAX) - 2;
				Batch BO = B; BO.seed_order = nullptr; BO.vr_ovf_run = 1; BO.vr_room = 0x7fffffff;
				const dim3 agrid((unsigned)(task_lanes / BLOCK));
#define SEED_LAUNCH_LR(ST_, M_) do { hipLaunchKernelGGL((k_seed<false, ST_, 1, 3, M_, 1>), agrid, block, seed_lds, h->stream, h->ix, *opt, BA); \
					hipLaunchKernelGGL((k_seed<false, ST_, 1, 3, M_, 1>), sgrid, block, seed_lds, h->stream, h->ix, *opt, BO); \
					hipLaunchKernelGGL((k_seed<false, ST_, 1, 3, M_, 2>), sgrid, block, seed_lds, h->stream, h->ix, *opt, B); } while (0)
				if (mrg == 2) { if (st) SEED_LAUNCH_LR(true, 2); else SEED_LAUNCH_LR(false, 2); }
				else { if (st) SEED_LAUNCH_LR(true, 0); else SEED_LAUNCH_LR(false, 0); }
#undef SEED_LAUNCH_LR
			}
			else {
				if (rd) { if (st) SEED_LAUNCH_B(true, true); else SEED_LAUNCH_B(true, false); }
				else { if (st) SEED_LAUNCH_B(false, true); else SEED_LAUNCH_B(false, false); }
				if (heavy_tasks && blk == 1) {
					// the reads the kernel above gave up: pass-1 tasks, the list of their pass-2 searches, those searches (all sized on the device)
					Batch BT = B; BT.rd_words = 0; BT.seed_order = nullptr; BT.task_tpr = heavy_tpr; BT.vr_room = 0x7fffffff; BT.seed_stack_cap = 0; BT.seed_budget = 0;
					const size_t lds_t = (size_t)(B.seed_lds_ent ? B.seed_lds_ent : 1) * sizeof(uint4) * BLOCK;
					const dim3 hgrid((unsigned)(heavy_lanes / BLOCK));
					if (st) hipLaunchKernelGGL((k_seed<false, true, 1, 3, 2, 1>), hgrid, block, lds_t, h->stream, h->ix, *opt, BT);
					else hipLaunchKernelGGL((k_seed<false, false, 1, 3, 2, 1>), hgrid, block, lds_t, h->stream, h->ix, *opt, BT);
					hipLaunchKernelGGL(k_seed_p2_tasks, dim3((unsigned)(n < 256 * 64 ? (n + BLOCK - 1) / BLOCK : 64)), block, 0, h->stream, *opt, BT);
					if (st) hipLaunchKernelGGL((k_seed<false, true, 1, 3, 2, 3>), hgrid, block, lds_t, h->stream, h->ix, *opt, BT);
					else hipLaunchKernelGGL((k_seed<false, false, 1, 3, 2, 3>), hgrid, block, lds_t, h->stream, h->ix, *opt, BT);
				}
			}
		}
#undef SEED_LAUNCH_B
#undef SEED_LAUNCH
		HIPCHK(h, hipEventRecord(h->ev[7], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_seed3+k_seed", hipGetErrorString(e_)); }
		if (long_batch && pick(cfg.publish_blk, 1))    // long reads: one workgroup per read sorts, counts and expands (167 -> 47 ms per 6000 x 10 kb reads, BENCH_r03 variants)
		{
			size_t pb = h->d_tmp_intv.cap / ((size_t)PUB_MAX * sizeof(Intv3));      // workgroups the spill area has scratch for (PUB_MAX records each)
			if (pb > 2048) pb = 2048; if (pb > (size_t)n) pb = (size_t)n; if (pb < 1) pb = 1;
			hipLaunchKernelGGL(k_publish_blk, dim3((unsigned)pb), block, 0, h->stream, *opt, B);
		}
		else {
			hipLaunchKernelGGL(k_publish, grid, block, 0, h->stream, *opt, B);      // (sorts, reserves and expands: k_expand's loop is its last step)
		}
		HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "publish+expand", hipGetErrorString(e_)); }
		i64 sa_blocks = (h->slot_cap + BLOCK - 1) / BLOCK;
		if (sa_blocks > MAX_RESIDENT_THREADS / BLOCK) sa_blocks = MAX_RESIDENT_THREADS / BLOCK;
		hipLaunchKernelGGL(k_sa, dim3((unsigned)sa_blocks), block, 0, h->stream, h->ix, B);
		HIPCHK(h, hipEventRecord(h->ev[2], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_sa", hipGetErrorString(e_)); }
		B.chain_regs = (int)cfg.chain_regs; B.chain_flt_lds = (int)(cfg.chain_flt_lds < CW_FLT_LDS ? cfg.chain_flt_lds : CW_FLT_LDS);
		{	// wave per read, heaviest reads (most seeds) first
			if (int rc2 = order_reads(h, B, B.seed_n)) return rc2;
			i64 nblk = ((i64)n + 3) / 4, cap = share(256 * 5);   // (5 workgroups per CU: __launch_bounds__ of k_chain_wave)
			hipLaunchKernelGGL(k_chain_wave, dim3((unsigned)(nblk < cap ? nblk : cap)), block, (size_t)CW_LDS_BYTES * 4, h->stream, h->ix, *opt, B);
		}
		HIPCHK(h, hipEventRecord(h->ev[3], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_chain", hipGetErrorString(e_)); }
		if (any_seedsw && h->max_len > WAVE_EXT_MAX_LEN) {   // long reads: one wavefront per read, one lane per seed
			int sc_max = 0; for (int k = 0; k < 25; ++k) if (opt->mat[k] > sc_max) sc_max = opt->mat[k];
			const i64 wcap = B.dp_waves > 0 ? B.dp_waves : 1;
			if (pick(cfg.seedsw_lds, 1) && SEEDSW_LDS_COLS * sc_max < 65536) {
				// the cell loop's state in LDS (dev_local_score_lds), one wave per workgroup (165 -> 26 ms per 6000 x 10 kb reads, BENCH_r03 variants)
				if (SEEDSW_LDS_COLS * sc_max < 256) hipLaunchKernelGGL((k_seedsw_wave<8>), dim3((unsigned)(n < wcap ? n : wcap)), dim3(64), (size_t)SEEDSW_LDS_COLS * 64 * 3, h->stream, h->ix, *opt, B);
				else hipLaunchKernelGGL((k_seedsw_wave<16>), dim3((unsigned)(n < wcap ? n : wcap)), dim3(64), (size_t)SEEDSW_LDS_COLS * 64 * 5, h->stream, h->ix, *opt, B);
			} else {
				i64 nblk = ((i64)n + 3) / 4, cap = wcap / 4 > 0 ? wcap / 4 : 1;      // (one DP scratch region per wave)
				hipLaunchKernelGGL((k_seedsw_wave<0>), dim3((unsigned)(nblk < cap ? nblk : cap)), block, 0, h->stream, h->ix, *opt, B);
			}
		} else if (any_seedsw) hipLaunchKernelGGL(k_seedsw, grid, block, 0, h->stream, h->ix, *opt, B);
		HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_seedsw", hipGetErrorString(e_)); }
		if (int rc2 = order_reads(h, B, B.reg_cap_r)) return rc2;   // heaviest reads (most seeds in kept chains) first
		// wave-per-read extension with the DP columns in LDS; its row-max scan packs (score << 6 | lane) into 31 bits
		i64 max_score = (i64)h->max_len * (opt->a > 0 ? opt->a : 1) * 2 + 1024;
		int ring_cols = 256;                     // long reads: ring of {H,E} columns wide enough for the widest band (2 * opt.w, bwamem.c:742)
		while (ring_cols < 4 * opt->w + 4 + 128) ring_cols <<= 1;
		if (h->max_len <= WAVE_EXT_MAX_LEN && max_score < (1 << 24)) {
			int lds_wave = (8 * (h->max_len + 2 + 64) + 5 * ((h->max_len + 64 + 3) & ~3) + 64 + 15) & ~15;   // {H,E} columns, query profile, scoring matrix, the stats runs' counters
			i64 nblk = ((i64)n + 3) / 4, cap = share(256 * 6);       // (six workgroups per CU are resident at 6 waves per SIMD)
			const int occ = (int)cfg.ext_occ;   // waves per SIMD the register allocation aims at (measured at 3.1 Gbp in round 2: 4 -> 63 ms, 5 -> 58 ms, 6 -> 56 ms; round 4, with the window rows: 4 -> 37.9, 5 -> 37.0-38.3, 6 -> 35.4-36.1 ms; the instance for 5 is not built -- one state of the source made hipcc 7.2 fail on it with an unaligned 64-bit spill reload, and it never won)
			const dim3 g((unsigned)(nblk < cap ? nblk : cap));
			// the chains' first extensions -- most of the stage's DP cells -- four to a wavefront, ahead of the kernel that replays mem_chain2aln's order-dependent
			// logic over them (dev_extp.h).  Not in stats runs: the work counters stay the one-wave routine's.
			if (cfg.ext_pack && !h->stats_on && h->max_len < 65536 && opt->w > 0 && (i64)h->max_len * sc_max_all(opt) < (1 << 21)) {
				B.ext_plan = 1;
				const i64 pcap = share(256 * 5);
				if (cfg.ext_pack == 5) hipLaunchKernelGGL((k_ext_pack<5>), dim3((unsigned)(nblk < pcap ? nblk : pcap)), block, (size_t)XP_LDS_BYTES * 4, h->stream, h->ix, *opt, B);
				else hipLaunchKernelGGL((k_ext_pack<4>), dim3((unsigned)(nblk < share(256 * 4) ? nblk : share(256 * 4))), block, (size_t)XP_LDS_BYTES * 4, h->stream, h->ix, *opt, B);
			}
			if (occ == 4) hipLaunchKernelGGL((k_extend_wave<false, 4>), g, block, (size_t)lds_wave * 4, h->stream, h->ix, *opt, B, lds_wave, 0);
			else hipLaunchKernelGGL((k_extend_wave<false, 6>), g, block, (size_t)lds_wave * 4, h->stream, h->ix, *opt, B, lds_wave, 0);
		} else if (max_score < (1 << 24) && ring_cols <= 2048) {
			// the band's columns only: independent of the read length
			const int lds_wave = 8 * ring_cols + 64;      // (the ring, the scoring matrix, the stats runs' counters)
			int wpb = 4; while (wpb > 1 && lds_wave * wpb > 65536) wpb >>= 1;
			i64 nblk = ((i64)n + wpb - 1) / wpb, cap = 256 * 8 * (4 / wpb);
			hipLaunchKernelGGL((k_extend_wave<true, 4>), dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(64 * wpb), (size_t)lds_wave * wpb, h->stream, h->ix, *opt, B, lds_wave, ring_cols);
		} else                                  // very wide bands: lane-per-read scalar DP with the columns in HBM scratch
			hipLaunchKernelGGL(k_extend, grid, block, 0, h->stream, h->ix, *opt, B);
		HIPCHK(h, hipEventRecord(h->ev[5], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_extend", hipGetErrorString(e_)); }
		if (long_batch || cfg.dedup_wave) {     // long reads: few reads, long patch alignments -> one wavefront per read
			// Ring of {H,E} columns for the patch alignments' band: 2 w + 132 columns, where w = max(min(.., 4 opt.w), |rlen - l_query| + 3) (bwa.c:180-187)
			// and the length difference of two merged regions of a 10 kb read with 13 % indels runs to several hundred bases.  A band the ring
			// cannot hold falls back to one lane with its columns in HBM -- 10^7 cells at one lane's pace: measured 0.6 s per call, 58 s of a
			// 100-read batch's 59 (profiles/r03_longread_probe.md) -- so the ring is sized for 1/8 of the longest read and capped by what one
			// wave's LDS share allows; wider rings mean fewer waves per workgroup (the dynamic LDS of a workgroup is 64 KiB).
			int need = 8 * opt->w + 4 + 128; if (need < h->max_len / 4 + 132) need = h->max_len / 4 + 132;
			int rc_ = 256; while (rc_ < need && rc_ < 4096) rc_ <<= 1;
			if (cfg.dedup_ring > 0) rc_ = (int)cfg.dedup_ring;      // test hook: a power of two, 256..4096
			if (rc_ < 256 || rc_ > 4096 || (rc_ & (rc_ - 1))) rc_ = 1024;
			const bool dedup_blk = (cfg.dedup_blk >= 0 ? cfg.dedup_blk : 1) != 0;   // four columns per lane in the patch alignments (wave_global2_score_ring_blk: 381 -> 207 ms per 6000 x 10 kb reads, BENCH_r03 variants); needs the segment in LDS
			int q_cap = dedup_blk ? (h->max_len + 15) & ~15 : 0;      // room for a patch alignment's query segment next to the ring
			if (8 * rc_ + 32 + q_cap > 65536) q_cap = 0;
			int wpb = rc_ <= 1024 ? 4 : (rc_ <= 2048 ? 2 : 1);
			while (wpb > 1 && (8 * rc_ + 32 + q_cap) * wpb > 65536) wpb >>= 1;
			i64 nblk = ((i64)n + wpb - 1) / wpb, cap = B.dp_waves / wpb > 0 ? B.dp_waves / wpb : 1;   // dp_h/dp_e hold one scratch region per wave
			if (dedup_blk && q_cap) hipLaunchKernelGGL(k_dedup_wave<true>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(64 * wpb), (size_t)(8 * rc_ + 32 + q_cap) * wpb, h->stream, h->ix, *opt, B, rc_, q_cap, 0, 0, (bwagpu_alnreg_t*)nullptr);
			else hipLaunchKernelGGL(k_dedup_wave<false>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(64 * wpb), (size_t)(8 * rc_ + 32 + q_cap) * wpb, h->stream, h->ix, *opt, B, rc_, q_cap, 0, 0, (bwagpu_alnreg_t*)nullptr);
		} else {
			// Lane per read for the reads with one or two regions (nine in ten of the headline's); the others are listed by k_dedup and done by two more
			// launches, one wavefront per read with the decisions' operands in LDS (dedup_read_par, dev_dedupp.h): the reads of up to dedup_stage regions
			// (16.6 KB of LDS per wave, nine to a CU), then the few with more (up to dedup_big regions in 64 KB; beyond that, in place in HBM).
			B.dd_heavy_min = dd_heavy_min; B.dd_list = h->d_heavy.as<i32>(); B.dd_prio = cfg.dedup_prio != 0; B.dd_net = (int)(cfg.dedup_net < 0 ? 129 : cfg.dedup_net);
			B.dd_stage_cap = dd.cap_m > 0 ? dd.cap_m : 0x3fffffff;      // (no LDS arrays: one list, every read in place)
			hipLaunchKernelGGL(k_dedup, dim3((unsigned)share(grid.x)), block, 0, h->stream, h->ix, *opt, B);
			if (B.dd_heavy_min > 0) {
				bwagpu_alnreg_t *tmp = h->d_dd_tmp.as<bwagpu_alnreg_t>();
				const size_t lds_m = dd.lds0 + DDW_PAR_BYTES(dd.cap_m), lds_b = dd.lds0 + DDW_PAR_BYTES(dd.cap_b);
				if (dd.q_cap) hipLaunchKernelGGL((k_dedup_wave<true, true>), dim3((unsigned)dd.blk_m), dim3(64), lds_m, h->stream, h->ix, *opt, B, dd.rc, dd.q_cap, dd.cap_m, 0, tmp);
				else hipLaunchKernelGGL((k_dedup_wave<false, true>), dim3((unsigned)dd.blk_m), dim3(64), lds_m, h->stream, h->ix, *opt, B, dd.rc, dd.q_cap, dd.cap_m, 0, tmp);
				if (dd.cap_m > 0) {
					tmp += (size_t)dd.blk_m * dd.cap_m;
					if (dd.q_cap) hipLaunchKernelGGL((k_dedup_wave<true, true>), dim3((unsigned)dd.blk_b), dim3(64), lds_b, h->stream, h->ix, *opt, B, dd.rc, dd.q_cap, dd.cap_b, 1, tmp);
					else hipLaunchKernelGGL((k_dedup_wave<false, true>), dim3((unsigned)dd.blk_b), dim3(64), lds_b, h->stream, h->ix, *opt, B, dd.rc, dd.q_cap, dd.cap_b, 1, tmp);
				}
			}
		}
		HIPCHK(h, hipEventRecord(h->ev[6], h->stream));
		if (dbg_sync) { hipError_t e_ = hipStreamSynchronize(h->stream); fprintf(stderr, "[bwagpu] attempt %d: %s done (%s)\n", attempt, "k_dedup", hipGetErrorString(e_)); }
		HIPCHK(h, hipGetLastError());
		h->phase = 22 + attempt * 100;
		Counters c;
		HIPCHK(h, hipMemcpyAsync(&c, h->d_ctr.p, sizeof c, hipMemcpyDeviceToHost, h->stream));
		HIPCHK(h, wait_stream(h));
		if (c.overflow) {   // grow what overflowed and redo the batch; nothing of the failed attempt is kept
			// the bump counters kept counting past the arenas' ends: grow to what was asked for (+10 %) in one step rather than by doubling
			if (c.overflow & 2) { const i64 want = (i64)(c.seed_used + c.seed_used / 10) + 4096; h->slot_cap = want > h->slot_cap * 5 / 4 ? want : h->slot_cap * 2; }
			if (c.overflow & 6) {
				i64 want = h->slot_cap / 4 + 2 * (i64)n + 64; const i64 asked = (i64)(c.node_used + c.node_used / 10) + 64;
				if (asked > want) want = asked;
				h->node_cap = want > h->node_cap * 5 / 4 ? want : h->node_cap * 2;
			}
			if (c.overflow & 8) { const i64 want = (i64)(c.reg_used + c.reg_used / 10) + 4096; h->reg_cap = (c.overflow & 2) || want <= h->reg_cap * 5 / 4 ? h->reg_cap * 2 : want; }
			if (c.overflow & 16) h->mem_cap = h->mem_cap * 4;
			if (c.overflow & 32) h->p2_factor *= 4.;
			++h->stats.n_retries; h->stats.retry_mask |= (int32_t)c.overflow;
			continue;
		}
		{	// remember the sizes that sufficed, per base
			const double nbd = (double)(h->n_bases > 1024 ? h->n_bases : 1024);
			if (h->slot_cap / nbd > h->need_slot) h->need_slot = h->slot_cap / nbd;
			if (h->node_cap / nbd > h->need_node) h->need_node = h->node_cap / nbd;
			if (h->reg_cap / nbd > h->need_reg) h->need_reg = h->reg_cap / nbd;
			std::lock_guard<std::mutex> l(h->ibuf->m);
			// The interval lists' capacity is a stride (read r's list sits at r * mem_cap), so a longer one costs every read of every later batch: four times the
			// arena, and k_publish / k_chain_wave walking lists 24 KB apart (measured after one retry in a 20 M-read run: k_publish 2.5 -> 20 ms in the batches that
			// followed, 16 GB per handle).  One read in ~20 million of the bench's genome leaves more than 256 intervals; redoing that one batch is the cheaper
			// answer, so the longer lists are only kept once overflows recur (two or more, and more often than one batch in sixteen).
			++h->ibuf->batches;
			if (h->stats.retry_mask & 16) ++h->ibuf->mem_events;
			const bool keep_mem = h->ibuf->mem_events >= 2 && h->ibuf->mem_events * 16 > h->ibuf->batches;
			if (h->mem_cap > h->need_mem && cfg.mem_cap <= 0 && (keep_mem || !(h->stats.retry_mask & 16))) h->need_mem = h->mem_cap;
			if (h->need_slot > h->ibuf->need_slot) h->ibuf->need_slot = h->need_slot;
			if (h->need_node > h->ibuf->need_node) h->ibuf->need_node = h->need_node;
			if (h->need_reg > h->ibuf->need_reg) h->ibuf->need_reg = h->need_reg;
			if (h->need_mem > h->ibuf->need_mem) h->ibuf->need_mem = h->need_mem;
		}
		float ms[6];
		for (int i = 0; i < 6; ++i) HIPCHK(h, hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
		HIPCHK(h, hipEventElapsedTime(&h->stats.ms_publish, h->ev[7], h->ev[1]));
		ms[0] -= h->stats.ms_publish;
		h->stats.ms_seed = ms[0]; h->stats.ms_sa = ms[1]; h->stats.ms_chain = ms[2]; h->stats.ms_seedsw = ms[3]; h->stats.ms_extend = ms[4]; h->stats.ms_dedup = ms[5];
		HIPCHK(h, hipEventElapsedTime(&h->stats.ms_total, h->ev[0], h->ev[6]));
		h->stats.n_seeds = (i64)c.seed_used;
		h->stats.n_intv = (i64)c.n_intv;
		h->stats.n_chains = (i64)c.n_chains; h->stats.n_regs_raw = (i64)c.n_regs_raw; h->stats.n_regs = (i64)c.n_regs;
		h->stats.n_tab_lookups = (i64)c.tab_lookups; h->stats.n_bt_nodes = (i64)c.bt_nodes; h->stats.n_chain_recs = (i64)c.chain_recs;
		h->stats.n_occ_blocks = (i64)c.occ_blocks; h->stats.n_lf_steps = (i64)c.lf_steps;
		h->stats.n_ext_calls = (i64)c.ext_calls; h->stats.n_ext_cells = (i64)c.ext_cells; h->stats.n_ext_fast = (i64)c.ext_fast;
		h->stats.n_glb_calls = (i64)c.glb_calls; h->stats.n_glb_cells = (i64)c.glb_cells; h->stats.ref_bases = (i64)c.ref_bases;
		h->stats.n_sw_calls = (i64)c.sw_calls; h->stats.n_sw_cells = (i64)c.sw_cells;
		h->ran = true;
		return BWAGPU_OK;
	}
	h->err = "arena growth did not converge";
	return BWAGPU_ENOMEM;
}

// gather per-read variable-length device records into read order on the host
template <class T>
static int gather(bwagpu_t *h, const DevBuf &d_n, const DevBuf &d_off, const DevBuf &d_rec, i64 rec_cap, int32_t *counts, T **out, int64_t *n_out)
{
	int n = h->n_reads;
	std::vector<i32> cnt(n); std::vector<i64> off(n);
	if (n) {
		HIPCHK(h, hipMemcpy(cnt.data(), d_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIPCHK(h, hipMemcpy(off.data(), d_off.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	}
	i64 tot = 0, hi = 0;
	for (int i = 0; i < n; ++i) { tot += cnt[i]; if (cnt[i] && off[i] + cnt[i] > hi) hi = off[i] + cnt[i]; }
	if (hi > rec_cap) { h->err = "internal: record range beyond arena"; return BWAGPU_EHIP; }
	std::vector<T> all((size_t)hi);
	if (hi) HIPCHK(h, hipMemcpy(all.data(), d_rec.p, (size_t)hi * sizeof(T), hipMemcpyDeviceToHost));
	T *res = (T*)malloc((size_t)(tot ? tot : 1) * sizeof(T));
	if (!res) return BWAGPU_ENOMEM;
	i64 k = 0;
	for (int i = 0; i < n; ++i) {
		if (counts) counts[i] = cnt[i];
		if (cnt[i]) memcpy(res + k, all.data() + off[i], (size_t)cnt[i] * sizeof(T));
		k += cnt[i];
	}
	*out = res; *n_out = tot;
	return BWAGPU_OK;
}

// The region arena is sparse (every read owns a range sized for its worst case); pack the used records on the device so
// that only they cross PCIe.  One lane per read, 88-byte records copied as 11 x u64.
__global__ void __launch_bounds__(256) k_pack_regs(int n, const i32 *reg_n, const i64 *reg_off, const bwagpu_alnreg_t *regs, const i64 *dst_off, bwagpu_alnreg_t *dst, i32 *dst_read)
{
	static_assert(sizeof(bwagpu_alnreg_t) == 88, "layout");
	for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
		const int c = reg_n[r];
		const u64 *src = (const u64*)(regs + reg_off[r]); u64 *d = (u64*)(dst + dst_off[r]);
		for (int k = 0; k < c * 11; ++k) d[k] = src[k];
		for (int k = 0; k < c; ++k) dst_read[dst_off[r] + k] = r;
	}
}

extern "C" int bwagpu_batch_download(bwagpu_t *h, int32_t *counts, bwagpu_alnreg_t **regs_out, int64_t *n_regs_out)
{
	if (!h || !h->ran || !regs_out || !n_regs_out) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	const int n = h->n_reads;
	h->cig_ext_n = -1;
	if (n == 0) { *regs_out = (bwagpu_alnreg_t*)malloc(sizeof(bwagpu_alnreg_t)); *n_regs_out = 0; return BWAGPU_OK; }
	h->phase = 30;
	std::vector<i32> cnt((size_t)n); std::vector<i64> dst((size_t)n);
	HIPCHK(h, hipMemcpyAsync(cnt.data(), h->d_reg_n.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, wait_stream(h));
	i64 tot = 0;
	for (int i = 0; i < n; ++i) { dst[i] = tot; tot += cnt[i]; if (counts) counts[i] = cnt[i]; }
	release_reserved(h);      // (the blocks bwagpu_batch_reserve page-locked for this handle go back to the pool, where the next lines find them)
	bwagpu_alnreg_t *res = (bwagpu_alnreg_t*)result_alloc((size_t)(tot ? tot : 1) * sizeof(bwagpu_alnreg_t));
	if (!res) return BWAGPU_ENOMEM;
	h->phase = 31;
	if (tot) {
		if (h->d_pack_off.ensure((size_t)n * 8) || h->d_regs_packed.ensure((size_t)tot * sizeof(bwagpu_alnreg_t)) || h->d_pack_read.ensure((size_t)tot * 4)) { bwagpu_free(res); h->err = "hipMalloc failed (packed regions)"; return BWAGPU_ENOMEM; }
		h->phase = 32;
		(void)hipEventRecord(h->ev[0], h->stream);
		hipError_t e = hipMemcpyAsync(h->d_pack_off.p, dst.data(), (size_t)n * 8, hipMemcpyHostToDevice, h->stream);
		if (e == hipSuccess) {
			int nb = (n + BLOCK - 1) / BLOCK; if (nb > 8192) nb = 8192;
			hipLaunchKernelGGL(k_pack_regs, dim3(nb), dim3(BLOCK), 0, h->stream, n, h->d_reg_n.as<i32>(), h->d_reg_off.as<i64>(), h->d_regs.as<bwagpu_alnreg_t>(),
							   h->d_pack_off.as<i64>(), h->d_regs_packed.as<bwagpu_alnreg_t>(), h->d_pack_read.as<i32>());
			e = hipGetLastError();
		}
		(void)hipEventRecord(h->ev[1], h->stream);
		if (e == hipSuccess) e = hipMemcpyAsync(res, h->d_regs_packed.p, (size_t)tot * sizeof(bwagpu_alnreg_t), hipMemcpyDeviceToHost, h->stream);
		(void)hipEventRecord(h->ev[2], h->stream);
		if (e == hipSuccess) e = wait_stream(h);
		if (e != hipSuccess) { bwagpu_free(res); HIPCHK(h, e); }
		(void)hipEventElapsedTime(&h->stats.ms_pack, h->ev[0], h->ev[1]); (void)hipEventElapsedTime(&h->stats.ms_download_copy, h->ev[1], h->ev[2]);
	}
	h->packed_tot = tot; h->phase = 39;
	*regs_out = res; *n_regs_out = tot;
	return BWAGPU_OK;
}

extern "C" int bwagpu_batch_cigars(bwagpu_t *h, const bwagpu_opt_t *opt, bwagpu_cigar_t **out, int64_t *n_out)
{
	if (!h || !opt || !h->ran || h->packed_tot < 0 || !out || !n_out) return BWAGPU_EINVAL;
	const BusyGuard busy(h->ibuf->busy);
	if (opt->e_del <= 0 || opt->e_ins <= 0) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	const i64 tot = h->packed_tot;
	h->phase = 40; h->cig_ext_n = -1;
	static_assert(sizeof(bwagpu_cigar_t) == 48, "layout");
	bwagpu_cigar_t *res = (bwagpu_cigar_t*)result_alloc((size_t)(tot ? tot : 1) * sizeof(bwagpu_cigar_t));
	if (!res) return BWAGPU_ENOMEM;
	if (tot) {
		if (h->d_cigs.ensure((size_t)tot * sizeof(bwagpu_cigar_t)) || h->d_ctr.ensure(sizeof(Counters))) { bwagpu_free(res); h->err = "hipMalloc failed (cigars)"; return BWAGPU_ENOMEM; }
		Batch B = {}; B.seq = h->d_seq.as<u8>(); B.off = h->d_off.as<i64>(); B.n_reads = h->n_reads; B.max_len = h->max_len;
		B.ctr = h->d_ctr.as<Counters>(); B.stats = h->stats_on;      // (stats: the fills' DP cells, counted in glb_cells / glb_calls -- the batch's run is over, its counters have been read)
		unsigned long long *next = &h->d_ctr.as<Counters>()->next_ext, *ext_used = &h->d_ctr.as<Counters>()->cig_ext_used;
		const int zc[2] = { CIG_Z_SMALL, CIG_Z_BIG };
		const int n_tier = (int)h->cfg.cig_tiers;   // diagnostics
		// operation array: sized for a typical batch; one that needs more (gap-rich reads) reports the total it reserved and is
		// redone once with exactly that much
		i64 ext_cap = tot * 4 + 65536;
		if (h->max_len > CIG_MAX_LEN && h->n_bases / 2 + 65536 > ext_cap) ext_cap = h->n_bases / 2 + 65536;   // (long reads: ~0.37 entries per base at 13 % indels -- operations and MD characters)
		if (h->cfg.cig_ops_cap > 0) ext_cap = h->cfg.cig_ops_cap;   // (tests of the second attempt)
		if (ext_cap < 1) ext_cap = 1;
		// third tier (k_cigar_long): segments, bands and operation counts beyond the LDS tiers' limits; one 64-thread workgroup per region at a time,
		// each with a direction matrix of its own in HBM.  A sizing pass (k_cigar_long_plan) counts the regions the LDS tiers left and the largest
		// matrix any of them can ask for; the scratch is as many such matrices as there are regions, at most 1024 and at most option cigl_mib (32 GiB).
		const bool long_tier = h->cfg.cig_long != 0;
		const i64 cigl_budget = (i64)h->cfg.cigl_mib << 20;
		const bool cig_trace = h->cfg.cig_trace != 0;
		auto t_last = std::chrono::steady_clock::now();
		auto lap = [&](const char *what) {
			if (!cig_trace) return;
			(void)wait_stream(h);
			auto t = std::chrono::steady_clock::now();
			fprintf(stderr, "[bwagpu] cigars: %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count());
			t_last = t;
		};
		unsigned long long used = 0;
		hipError_t e = hipSuccess;
		(void)hipEventRecord(h->ev[0], h->stream);
		for (int attempt = 0; attempt < 2; ++attempt) {
			if (h->d_cig_ext.ensure((size_t)ext_cap * 4)) { bwagpu_free(res); h->err = "hipMalloc failed (cigars)"; return BWAGPU_ENOMEM; }
			e = hipMemsetAsync(ext_used, 0, sizeof(unsigned long long), h->stream);
			if (e == hipSuccess && h->stats_on) e = hipMemsetAsync(&h->d_ctr.as<Counters>()->glb_calls, 0, 2 * sizeof(unsigned long long), h->stream);      // (glb_calls, glb_cells: adjacent)
			for (int tier = 0; tier < n_tier && e == hipSuccess; ++tier) {   // narrow bands at high occupancy, then the deferred wide ones
				e = hipMemsetAsync(next, 0, sizeof(unsigned long long), h->stream);
				if (e != hipSuccess) break;
				h->phase = 41 + tier;
				// tier 0: the diag form only (bands of up to 64 columns whose direction nibbles fit CIG_Z_SMALL; no {H,E} columns in LDS: 6.3 KB per wave);
				// tier 1: both forms with CIG_Z_BIG nibbles, for what tier 0 deferred
				const int lds_wave = tier == 0 ? CIG_LDS_BYTES_DIAG(zc[tier]) : CIG_LDS_BYTES(zc[tier]);
				const int wpb = tier == 0 ? 4 : 2;                 // waves per workgroup: the wide tier stays below 64 KiB of LDS per group
				i64 nblk = (tot + wpb - 1) / wpb, cap = 256 * 6;
				if (tier == 0) hipLaunchKernelGGL(k_cigar<true>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(64 * wpb), (size_t)lds_wave * wpb, h->stream, h->ix, *opt, B, tot,
								   h->d_regs_packed.as<bwagpu_alnreg_t>(), h->d_pack_read.as<i32>(), h->d_cigs.as<bwagpu_cigar_t>(), next, zc[tier], tier,
								   h->d_cig_ext.as<u32>(), ext_used, ext_cap, h->cigar_filter ? h->d_pack_off.as<i64>() : (const i64*)nullptr);
				else hipLaunchKernelGGL(k_cigar<false>, dim3((unsigned)(nblk < cap ? nblk : cap)), dim3(64 * wpb), (size_t)lds_wave * wpb, h->stream, h->ix, *opt, B, tot,
								   h->d_regs_packed.as<bwagpu_alnreg_t>(), h->d_pack_read.as<i32>(), h->d_cigs.as<bwagpu_cigar_t>(), next, zc[tier], tier,
								   h->d_cig_ext.as<u32>(), ext_used, ext_cap, h->cigar_filter ? h->d_pack_off.as<i64>() : (const i64*)nullptr);
				e = hipGetLastError();
			}
			lap("LDS tiers");
			if (e == hipSuccess && long_tier) {
				unsigned long long *plan_d = h->d_ctr.as<Counters>()->cigl_plan, plan[2] = { 0, 0 };
				if (h->d_cigl_list.ensure((size_t)tot * 4)) { bwagpu_free(res); h->err = "hipMalloc failed (cigars)"; return BWAGPU_ENOMEM; }
				e = hipMemsetAsync(plan_d, 0, sizeof plan, h->stream);
				if (e == hipSuccess) {
					h->phase = 43;
					i64 nb = (tot + 255) / 256; if (nb > 2048) nb = 2048;
					hipLaunchKernelGGL(k_cigar_long_plan, dim3((unsigned)nb), dim3(256), 0, h->stream, h->ix, *opt, tot, h->d_regs_packed.as<bwagpu_alnreg_t>(), h->d_cigs.as<bwagpu_cigar_t>(), plan_d, h->d_cigl_list.as<i32>());
					e = hipGetLastError();
				}
				if (e == hipSuccess) e = hipMemcpyAsync(plan, plan_d, sizeof plan, hipMemcpyDeviceToHost, h->stream);
				if (e == hipSuccess) e = wait_stream(h);
				if (e == hipSuccess && plan[0] > 0 && plan[1] > 0) {
					// matrices a quarter larger than this batch's largest and a multiple of 1 MiB, so that the next batches rarely re-allocate
					i64 z_cap = (i64)plan[1] + (i64)plan[1] / 4; z_cap = (z_cap + ((i64)1 << 20) - 1) & ~(((i64)1 << 20) - 1);
					if (h->cigl_z_cap > z_cap) z_cap = h->cigl_z_cap;
					i64 n_long = cigl_budget / z_cap; if (n_long > 1024) n_long = 1024; if (n_long > (i64)plan[0]) n_long = (i64)plan[0]; if (n_long < 1) n_long = 1;
					if (h->d_cigl_z.ensure((size_t)z_cap * n_long) || h->d_cigl_ops.ensure((size_t)n_long * CIGL_MAX_OPS * 4) || h->d_cigl_md.ensure((size_t)n_long * CIGL_MD_CAP)) {
						bwagpu_free(res); h->err = "hipMalloc failed (cigars)"; return BWAGPU_ENOMEM; }
					h->cigl_z_cap = z_cap;
					if (cig_trace) fprintf(stderr, "[bwagpu] cigars: long tier: %llu regions, largest matrix %.1f MB, %lld workgroups x %.1f MB\n", plan[0], plan[1] / 1e6, (long long)n_long, z_cap / 1e6);
					lap("long tier plan + scratch");
					e = hipMemsetAsync(next, 0, sizeof(unsigned long long), h->stream);
					if (e == hipSuccess) {
						h->phase = 44;
						hipLaunchKernelGGL(k_cigar_long, dim3((unsigned)n_long), dim3(64), (size_t)CIGL_LDS_BYTES, h->stream, h->ix, *opt, B, (i64)plan[0], h->d_cigl_list.as<i32>(),
										   h->d_regs_packed.as<bwagpu_alnreg_t>(), h->d_pack_read.as<i32>(), h->d_cigs.as<bwagpu_cigar_t>(), next,
										   h->d_cigl_z.as<u8>(), z_cap, h->d_cigl_ops.as<u32>(), h->d_cigl_md.as<u8>(), h->d_cig_ext.as<u32>(), ext_used, ext_cap);
						e = hipGetLastError();
					}
					lap("k_cigar_long");
				}
			}
			if (e == hipSuccess) e = hipMemcpyAsync(&used, ext_used, sizeof used, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess) e = wait_stream(h);
			if (e != hipSuccess || (i64)used <= ext_cap) break;
			ext_cap = (i64)used;
		}
		h->phase = 45;
		(void)hipEventRecord(h->ev[1], h->stream);
		if (e == hipSuccess && h->stats_on) {
			unsigned long long cc[2] = { 0, 0 };
			e = hipMemcpyAsync(cc, &h->d_ctr.as<Counters>()->glb_calls, sizeof cc, hipMemcpyDeviceToHost, h->stream);
			if (e == hipSuccess) e = wait_stream(h);
			h->stats.n_cig_dp = (i64)cc[0]; h->stats.n_cig_cells = (i64)cc[1];
		}
		if (e == hipSuccess) e = hipMemcpyAsync(res, h->d_cigs.p, (size_t)tot * sizeof(bwagpu_cigar_t), hipMemcpyDeviceToHost, h->stream);
		(void)hipEventRecord(h->ev[2], h->stream);
		if (e == hipSuccess) e = wait_stream(h);
		if (e != hipSuccess) { bwagpu_free(res); h->cig_ext_n = -1; HIPCHK(h, e); }
		(void)hipEventElapsedTime(&h->stats.ms_cigar_kernels, h->ev[0], h->ev[1]); (void)hipEventElapsedTime(&h->stats.ms_cigar_copy, h->ev[1], h->ev[2]);     // (kernels: includes the plan's small D2H and, after an overflow, the second attempt)
		h->cig_ext_n = (i64)used < ext_cap ? (i64)used : ext_cap;
	}
	if (tot == 0) h->cig_ext_n = 0;
	*out = res; *n_out = tot;
	return BWAGPU_OK;
}

extern "C" int bwagpu_batch_cigar_ops(bwagpu_t *h, uint32_t **ops, int64_t *n_ops)
{
	if (!h || !ops || !n_ops || h->cig_ext_n < 0) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	const i64 n = h->cig_ext_n;
	uint32_t *res = (uint32_t*)result_alloc((size_t)(n ? n : 1) * 4);
	if (!res) return BWAGPU_ENOMEM;
	if (n) {
		(void)hipEventRecord(h->ev[0], h->stream);
		hipError_t e = hipMemcpyAsync(res, h->d_cig_ext.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream);
		(void)hipEventRecord(h->ev[1], h->stream);
		if (e == hipSuccess) e = wait_stream(h);
		if (e != hipSuccess) { bwagpu_free(res); HIPCHK(h, e); }
		float ms = 0; (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]); h->stats.ms_cigar_copy += ms;
	}
	*ops = res; *n_ops = n;
	return BWAGPU_OK;
}

extern "C" int bwagpu_batch_matesw(bwagpu_t *h, const bwagpu_opt_t *opt, const bwagpu_pes_t pes[4], bwagpu_matesw_t **out, int64_t *n_out)
{
	if (!h || !opt || !pes || !h->ran || h->packed_tot < 0 || !out || !n_out) return BWAGPU_EINVAL;
	if (opt->e_del <= 0 || opt->e_ins <= 0 || (h->n_reads & 1)) return BWAGPU_EINVAL;
	const BusyGuard busy(h->ibuf->busy);
	static_assert(sizeof(bwagpu_matesw_t) == 56 && sizeof(MateTask) == 24 && sizeof(bwagpu_pes_t) == 16, "layout");
	HIPCHK(h, hipSetDevice(h->device));
	const int n = h->n_reads;
	*out = nullptr; *n_out = 0;
	if (n == 0 || h->packed_tot == 0) { *out = (bwagpu_matesw_t*)malloc(sizeof(bwagpu_matesw_t)); return *out ? BWAGPU_OK : BWAGPU_ENOMEM; }
	const i64 task_cap = (i64)n * 2 + 1024;          // more candidates than this are simply left to the host
	const int waves = 1024;
	if (h->d_msw_tasks.ensure((size_t)task_cap * sizeof(MateTask)) || h->d_msw_out.ensure((size_t)task_cap * sizeof(bwagpu_matesw_t)) ||
		h->d_msw_pes.ensure(4 * sizeof(bwagpu_pes_t)) || h->d_ctr.ensure(sizeof(Counters))) {
		h->err = "hipMalloc failed (mate rescue)"; return BWAGPU_ENOMEM;
	}
	unsigned long long *n_tasks = &h->d_ctr.as<Counters>()->next_chain, *next = &h->d_ctr.as<Counters>()->next_dedup;   // idle counters at this point
	HIPCHK(h, hipMemsetAsync(n_tasks, 0, 8, h->stream));
	HIPCHK(h, hipMemsetAsync(next, 0, 8, h->stream));
	HIPCHK(h, hipMemcpyAsync(h->d_msw_pes.p, pes, 4 * sizeof(bwagpu_pes_t), hipMemcpyHostToDevice, h->stream));
	int nb = (n / 2 + BLOCK - 1) / BLOCK; if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
	hipLaunchKernelGGL(k_matesw_tasks, dim3(nb), dim3(BLOCK), 0, h->stream, h->ix, *opt, n, h->d_reg_n.as<i32>(), h->d_pack_off.as<i64>(), h->d_regs_packed.as<bwagpu_alnreg_t>(),
					   h->d_msw_pes.as<bwagpu_pes_t>(), h->d_msw_tasks.as<MateTask>(), n_tasks, task_cap);
	HIPCHK(h, hipGetLastError());
	unsigned long long nt = 0;
	HIPCHK(h, hipMemcpyAsync(&nt, n_tasks, 8, hipMemcpyDeviceToHost, h->stream));
	HIPCHK(h, wait_stream(h));
	if ((i64)nt > task_cap) nt = (unsigned long long)task_cap;
	bwagpu_matesw_t *res = (bwagpu_matesw_t*)result_alloc((size_t)(nt ? nt : 1) * sizeof(bwagpu_matesw_t));
	if (!res) return BWAGPU_ENOMEM;
	if (nt) {
		Batch B = {}; B.seq = h->d_seq.as<u8>(); B.off = h->d_off.as<i64>(); B.n_reads = n; B.max_len = h->max_len;
		const i64 want = ((i64)nt + 3) / 4;           // one wavefront per alignment, four per workgroup
		hipLaunchKernelGGL(k_matesw_sw, dim3((unsigned)(want < waves / 4 * 8 ? want : waves / 4 * 8)), dim3(BLOCK), 0, h->stream, h->ix, *opt, B, h->d_msw_pes.as<bwagpu_pes_t>(), h->d_msw_tasks.as<MateTask>(), (i64)nt,
						   h->d_msw_out.as<bwagpu_matesw_t>(), next);
		hipError_t e = hipGetLastError();
		if (e == hipSuccess) e = hipMemcpyAsync(res, h->d_msw_out.p, (size_t)nt * sizeof(bwagpu_matesw_t), hipMemcpyDeviceToHost, h->stream);
		if (e == hipSuccess) e = wait_stream(h);
		if (e != hipSuccess) { bwagpu_free(res); HIPCHK(h, e); }
	}
	*out = res; *n_out = (int64_t)nt;
	return BWAGPU_OK;
}

extern "C" int bwagpu_align_flat(bwagpu_t *h, const bwagpu_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off,
								 int32_t *counts, bwagpu_alnreg_t **regs_out, int64_t *n_regs_out)
{
	int rc;
	if ((rc = bwagpu_batch_upload(h, n, seqs, off))) return rc;
	if ((rc = bwagpu_batch_run(h, opt))) return rc;
	return bwagpu_batch_download(h, counts, regs_out, n_regs_out);
}

// nst_nt4_table (bntseq.c:46-63) as a function: A/a C/c G/g T/t -> 0..3, '-' -> 5, everything else 4
static inline uint8_t nt4(unsigned char c)
{
	switch (c) {
	case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3;
	case '-': return 5;
	default: return 4;
	}
}

extern "C" int bwagpu_align_bseq(bwagpu_t *h, const bwagpu_opt_t *opt, int n, bwagpu_bseq1_t *seqs, bwagpu_alnreg_v *regs)
{
	if (!h || !opt || n < 0 || (n > 0 && (!seqs || !regs))) return BWAGPU_EINVAL;
	std::vector<i64> off((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) off[i + 1] = off[i] + seqs[i].l_seq;
	std::vector<u8> flat((size_t)off[n] + 1);
	for (int i = 0; i < n; ++i) {   // in-place nt4 encoding, the contract of mem_align1_core (bwamem.c:1087-1088)
		char *s = seqs[i].seq;
		for (int j = 0; j < seqs[i].l_seq; ++j) { unsigned char c = (unsigned char)s[j]; s[j] = (char)(c < 4 ? c : nt4(c)); }
		memcpy(flat.data() + off[i], s, (size_t)seqs[i].l_seq);
	}
	std::vector<int32_t> counts((size_t)n);
	bwagpu_alnreg_t *all = nullptr; int64_t tot = 0;
	int rc = bwagpu_align_flat(h, opt, n, flat.data(), off.data(), counts.data(), &all, &tot);
	if (rc) return rc;
	i64 k = 0;
	for (int i = 0; i < n; ++i) {
		regs[i].n = regs[i].m = (size_t)counts[i];
		regs[i].a = (bwagpu_alnreg_t*)malloc((size_t)(counts[i] ? counts[i] : 1) * sizeof(bwagpu_alnreg_t));
		if (!regs[i].a) {   // leave no half-filled output behind
			for (int j = 0; j < i; ++j) { free(regs[j].a); regs[j].a = nullptr; regs[j].n = regs[j].m = 0; }
			regs[i].n = regs[i].m = 0;
			bwagpu_free(all);
			return BWAGPU_ENOMEM;
		}
		memcpy(regs[i].a, all + k, (size_t)counts[i] * sizeof(bwagpu_alnreg_t));
		k += counts[i];
	}
	bwagpu_free(all);
	return BWAGPU_OK;
}

// ---- differential tests of the DP routines (dev_debug.h) ---------------------------------------------------------------------
extern "C" int bwagpu_debug_dp(bwagpu_t *h, const bwagpu_opt_t *opt, int kind, int n_cases, const bwagpu_dp_case_t *cases, const uint8_t *seqs, int64_t n_seq_bytes, int32_t *out)
{
	if (!h || !opt || n_cases < 0 || kind < 0 || kind > 7 || (n_cases > 0 && (!cases || !seqs || !out)) || n_seq_bytes < 0) return BWAGPU_EINVAL;
	if (opt->e_del <= 0 || opt->e_ins <= 0) return BWAGPU_EINVAL;
	if (n_cases == 0) return BWAGPU_OK;
	static_assert(sizeof(bwagpu_dp_case_t) == 32, "layout");
	int max_q = 1, max_w = 1;
	for (int i = 0; i < n_cases; ++i) {
		const bwagpu_dp_case_t &c = cases[i];
		if (c.q_len < 0 || c.t_len < 0 || c.q_off < 0 || c.t_off < 0 || (i64)c.q_off + c.q_len > n_seq_bytes || (i64)c.t_off + c.t_len > n_seq_bytes || c.w < 0) return BWAGPU_EINVAL;
		if ((kind == 0 || kind == 1 || kind >= 6) && c.h0 <= 0) return BWAGPU_EINVAL;      // ksw_extend2 asserts h0 > 0 (ksw.c:420)
		if (c.q_len > max_q) max_q = c.q_len;
		if (c.w > max_w) max_w = c.w;
	}
	HIPCHK(h, hipSetDevice(h->device));
	DevBuf d_seq, d_pac, d_cases, d_out, d_scr, d_pac2;
	int rc = BWAGPU_OK;
	const int grid = n_cases < 2048 ? n_cases : 2048;
	const bool dbg_blk = (h->cfg.dedup_blk >= 0 ? h->cfg.dedup_blk : 1) != 0;      // kind 3 in its four-columns-per-lane form (the product's default) or, option dedup_blk = 0, one column per lane
	hipError_t e = hipSuccess;
	if (d_seq.ensure((size_t)n_seq_bytes + 16) || d_pac.ensure((size_t)n_seq_bytes / 4 + 16) || d_cases.ensure((size_t)n_cases * sizeof(bwagpu_dp_case_t)) || d_out.ensure((size_t)n_cases * DBG_OUT_INTS * 4)) { h->err = "hipMalloc failed (debug)"; rc = BWAGPU_ENOMEM; goto done; }
	e = hipMemcpyAsync(d_seq.p, seqs, (size_t)n_seq_bytes, hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemcpyAsync(d_cases.p, cases, (size_t)n_cases * sizeof(bwagpu_dp_case_t), hipMemcpyHostToDevice, h->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_out.p, 0, (size_t)n_cases * DBG_OUT_INTS * 4, h->stream);
	if (e == hipSuccess) {
		hipLaunchKernelGGL(k_debug_pack, dim3(256), dim3(256), 0, h->stream, d_seq.as<u8>(), (i64)n_seq_bytes, d_pac.as<u8>());
		DevIndex ix = h->ix; ix.pac = d_pac.as<u8>(); ix.l_pac = n_seq_bytes;
		if (kind == 0) {
			if (max_q > WAVE_EXT_MAX_LEN) { rc = BWAGPU_EINVAL; goto done; }
			const size_t lds = (8 * (size_t)(max_q + 2 + 64) + 5 * (size_t)((max_q + 64 + 3) & ~3) + 32 + 15) & ~(size_t)15;
			hipLaunchKernelGGL((k_debug_extend<false>), dim3(grid), dim3(64), lds, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), max_q, 0, d_out.as<i32>(), 0);
		} else if (kind == 1) {
			int ring_cols = 256; while (ring_cols < 2 * max_w + 4 + 128) ring_cols <<= 1;
			if (ring_cols > 4096) { rc = BWAGPU_EINVAL; goto done; }
			hipLaunchKernelGGL((k_debug_extend<true>), dim3(grid), dim3(64), (size_t)8 * ring_cols + 32, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), max_q, ring_cols, d_out.as<i32>(), (int)(h->cfg.ext_blk != 0));
		} else if (kind == 2) {
			hipLaunchKernelGGL(k_debug_global, dim3(grid), dim3(64), (size_t)CIG_LDS_BYTES(CIG_Z_BIG), h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), d_out.as<i32>());
		} else if (kind == 3) {
			int ring_cols = 256; while (ring_cols < 2 * max_w + 4 + 128) ring_cols <<= 1;
			if (ring_cols > 4096) { rc = BWAGPU_EINVAL; goto done; }
			const int q_cap = dbg_blk && 8 * ring_cols + 32 + ((max_q + 15) & ~15) <= 65536 ? (max_q + 15) & ~15 : 0;
			if (dbg_blk) hipLaunchKernelGGL(k_debug_global_ring<true>, dim3(grid), dim3(64), (size_t)8 * ring_cols + 32 + q_cap, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), ring_cols, d_out.as<i32>(), q_cap);
			else hipLaunchKernelGGL(k_debug_global_ring<false>, dim3(grid), dim3(64), (size_t)8 * ring_cols + 32 + q_cap, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), ring_cols, d_out.as<i32>(), q_cap);
		} else if (kind == 6 || kind == 7) {
			const int g4 = (n_cases + 3) / 4 < 2048 ? (n_cases + 3) / 4 : 2048;
			if (kind == 6) hipLaunchKernelGGL((k_debug_extpack<4>), dim3(g4), dim3(64), (size_t)XP_LDS_BYTES, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), d_out.as<i32>());
			else hipLaunchKernelGGL((k_debug_extpack<8>), dim3(g4), dim3(64), (size_t)XP_LDS_BYTES, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), d_out.as<i32>());
		} else if (kind == 5) {
			int max_t = 1; for (int i = 0; i < n_cases; ++i) if (cases[i].t_len > max_t) max_t = cases[i].t_len;
			i64 z_cap = ((i64)max_t + 16) * ((CIGL_MAX_COLS + 15) & ~15); z_cap = (z_cap + 15) & ~(i64)15;
			const int blocks = grid < 64 ? grid : 64;
			if (d_scr.ensure((size_t)blocks * z_cap) || d_pac2.ensure((size_t)blocks * CIGL_MAX_OPS * 4)) { h->err = "hipMalloc failed (debug)"; rc = BWAGPU_ENOMEM; goto done; }
			hipLaunchKernelGGL(k_debug_global_long, dim3(blocks), dim3(64), (size_t)CIGL_LDS_BYTES, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), d_scr.as<u8>(), z_cap, d_pac2.as<u32>(), d_out.as<i32>());
		} else {
			hipLaunchKernelGGL(k_debug_align2, dim3(grid), dim3(64), 0, h->stream, ix, *opt, n_cases, d_cases.as<bwagpu_dp_case_t>(), d_seq.as<u8>(), d_out.as<i32>());
		}
		e = hipGetLastError();
	}
	if (e == hipSuccess) e = hipMemcpyAsync(out, d_out.p, (size_t)n_cases * DBG_OUT_INTS * 4, hipMemcpyDeviceToHost, h->stream);
	if (e == hipSuccess) e = wait_stream(h);
	if (e != hipSuccess) { h->err = std::string("bwagpu_debug_dp: ") + hipGetErrorString(e); rc = BWAGPU_EHIP; }
done:
	(void)hipStreamSynchronize(h->stream);
	d_seq.release(); d_pac.release(); d_cases.release(); d_out.release(); d_scr.release(); d_pac2.release();
	return rc;
}

// ---- stage taps ----------------------------------------------------------------------------------------------
extern "C" int bwagpu_tap_intervals(bwagpu_t *h, int32_t *counts, bwagpu_intv_t **out, int64_t *n_out)
{
	if (!h || !h->ran || !out || !n_out) return BWAGPU_EINVAL;
	static_assert(sizeof(bwagpu_intv_t) == sizeof(Intv3), "layout");
	return gather<bwagpu_intv_t>(h, h->d_intv_n, h->d_intv_off, h->d_intv, (i64)h->n_reads * h->mem_cap, counts, out, n_out);
}

extern "C" int bwagpu_tap_chains(bwagpu_t *h, int32_t *counts, bwagpu_chain_t **chains, int64_t *n_chains, bwagpu_seed_t **seeds, int64_t *n_seeds)
{
	if (!h || !h->ran || !chains || !n_chains || !seeds || !n_seeds) return BWAGPU_EINVAL;
	HIPCHK(h, hipSetDevice(h->device));
	const int n = h->n_reads;
	std::vector<i32> cn((size_t)n), sn((size_t)n); std::vector<i64> off((size_t)n);
	if (n) {
		HIPCHK(h, hipMemcpy(cn.data(), h->d_chain_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIPCHK(h, hipMemcpy(sn.data(), h->d_seed_n.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		HIPCHK(h, hipMemcpy(off.data(), h->d_seed_off.p, (size_t)n * 8, hipMemcpyDeviceToHost));
	}
	std::vector<bwagpu_chain_t> vc; std::vector<bwagpu_seed_t> vs; std::vector<bwagpu_chain_t> tc; std::vector<bwagpu_seed_t> ts;
	for (int i = 0; i < n; ++i) {   // per-read regions: kept chain headers at +80n, their seeds at +112n (RegionView)
		if (counts) counts[i] = cn[i];
		if (cn[i] == 0) continue;
		const u8 *base = h->d_slot_blob.as<u8>() + off[i] * SLOT_BLOB_BYTES;
		tc.resize((size_t)cn[i]);
		HIPCHK(h, hipMemcpy(tc.data(), base + (size_t)80 * sn[i], (size_t)cn[i] * sizeof(bwagpu_chain_t), hipMemcpyDeviceToHost));
		int ks = 0; for (auto &c : tc) ks += c.n_seeds;
		ts.resize((size_t)ks);
		if (ks) HIPCHK(h, hipMemcpy(ts.data(), base + (size_t)112 * sn[i], (size_t)ks * sizeof(bwagpu_seed_t), hipMemcpyDeviceToHost));
		vc.insert(vc.end(), tc.begin(), tc.end()); vs.insert(vs.end(), ts.begin(), ts.end());
	}
	*chains = (bwagpu_chain_t*)malloc((vc.size() + 1) * sizeof(bwagpu_chain_t)); *seeds = (bwagpu_seed_t*)malloc((vs.size() + 1) * sizeof(bwagpu_seed_t));
	if (!*chains || !*seeds) return BWAGPU_ENOMEM;
	if (!vc.empty()) memcpy(*chains, vc.data(), vc.size() * sizeof(bwagpu_chain_t));
	if (!vs.empty()) memcpy(*seeds, vs.data(), vs.size() * sizeof(bwagpu_seed_t));
	*n_chains = (int64_t)vc.size(); *n_seeds = (int64_t)vs.size();
	return BWAGPU_OK;
}

extern "C" int bwagpu_tap_regs_raw(bwagpu_t *h, int32_t *counts, bwagpu_alnreg_t **out, int64_t *n_out)
{
	if (!h || !h->ran || !h->taps_on || !out || !n_out) return BWAGPU_EINVAL;
	return gather<bwagpu_alnreg_t>(h, h->d_reg_n_raw, h->d_reg_off, h->d_regs_raw, h->reg_cap, counts, out, n_out);
}
