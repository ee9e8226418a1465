// dev_common.h -- shared device-side types of the MI355X BWA-MEM core (gfx950, wave64).
//
// Data layout in HBM:
//   * FM-index blocks are kept exactly as in the reference's .bwt (bwtindex.c:150-172): one 64-byte block per
//     128 BWT symbols = 4 x u64 running Occ(A,C,G,T) followed by 8 x u32 of 2-bit symbols.  A block is one
//     64-byte HBM burst, fetched by a lane as 4 x dwordx4.
//   * reads: 1 byte per base (nt4 codes 0..4), concatenated, int64 offsets.
//   * per-read variable-size results live in bump-allocated arenas ("slot space"); a read reserves its slots
//     with one atomicAdd and never moves.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bwagpu.h"

typedef uint64_t u64;
typedef int64_t i64;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint8_t u8;

#define DEVFN __device__ __forceinline__
// Scheduling fence for a 32-bit register value (no instruction is emitted): the value has to be in its register here -- loads issued
// before have landed; none of them is narrowed, split or sunk past this point.  (The CPU test harness supplies its own no-op.)
#ifndef DEV_KEEP
#define DEV_KEEP(v) asm volatile("" : "+v"(v))
#endif
typedef __amdgpu_buffer_rsrc_t BufRsrc;   // a buffer descriptor (V#) for range-checked loads (dev_fm.h: buf_rsrc, buf_load16)

// The index as the kernels see it.
struct DevIndex {
	const uint4 *bwt;      // 64-byte blocks, 4 x uint4 each
	u64 primary, L2[5], seq_len;
	const u64 *sa;         // sampled (or densified) suffix array
	u64 sa_mask;           // sa_intv - 1
	int sa_shift;          // log2(sa_intv)
	const u8 *pac;
	i64 l_pac;
	int n_seqs;
	const i64 *ctg_off;
	const i32 *ctg_len;
	const i32 *ctg_alt;
	// prefix tables: one record per ptab_m-mer W, holding the bi-intervals of W's prefixes of length 1..ptab_m as packed
	// 16-byte entries (SeedStack::pack layout) -- every look-up a search makes while its window q[x..x+m) stays the same hits
	// the same 16*m-byte record
	const uint4 *ptab;
	int ptab_m;
	// optional second layout of the BWT for the seeding kernels (BWAGPU_OCC32=1): 32-byte blocks of 64 bases -- four 32-bit counts relative to
	// the block's superblock (2^occ_sb_shift bases, 2^32 unless a test asks for less; occ_sb holds four 64-bit counts per superblock: two or
	// three entries for a human genome, which never leave the L1 cache) and the 64 bases.
	// One rank query then costs one 32-byte memory request instead of a 64-byte one; the chip serves those 1.6 times as fast
	// (profiles/r02_experiments.md).  Null: not built.
	const uint4 *occ32;
	const u64 *occ_sb;
	const uint4 *occ_sbx;  // the same table per symbol: [superblock][c] = { count of c, summed counts of the symbols above c } (one-trip rank routine, dev_fm.h)
	u64 occ32_bytes, occ_sbx_bytes, ptab_bytes;   // sizes of occ32 / occ_sbx / ptab (buffer descriptors of the one-trip routine)
	int occ_sb_shift;
};

// SA interval kept by the seeding stage: {x0, x2, info}; the reverse-strand start x[1] of bwtintv_t (bwt.h:62)
// is not needed after seeding (mem_chain reads x[0], x[2] and info only, bwamem.c:299-309).
struct Intv3 { u64 x0, x2, info; };

// Full bi-interval used inside the SMEM search.
struct BiIntv { u64 x0, x1, x2; u64 info; };

// Chain record while chaining (per read, indexed in slot space).
struct ChainRec {
	i64 pos;               // mem_chain_t::pos = rbeg of the first seed
	i64 last_rbeg;         // rbeg of the last seed
	i32 first, last;       // first / last seed slot (linked through Batch::slot_next)
	i32 first_qbeg, last_qbeg, last_len;
	i32 n;                 // number of seeds
	i32 rid;
	i32 w;                 // weight (mem_chain_weight)
	i32 kept;              // (rounds 1-4: mem_chain_t::kept; since round 5 the filter keeps the kinds by sorted place -- dev_chainw.h `kind` -- and this stays 0)
	i32 first_shadow;      // (likewise mem_chain_t::first: now the filter's `shadow` array; stays -1)
	i32 is_alt;
};

#define ORDER_BINS 32
#define SLOT_BLOB_BYTES 160   // 64 chain record + 16 packed kept-chain record + 32 kept chain header + 24 seed + 8 sort key + 3 x 4 ints (+4 pad)

// Every wave of a kernel draws its work from these counters, and same-address atomics complete at only ~75 M/s on this chip
// (measured: 10^6 fetches = 13 ms, most of the round-1 chaining kernel's "bulk").  Hence: the hot counters sit on cache lines of
// their own (atomics of different counters then go to different L2 channels), and kernels draw reads in chunks (wave_fetch_n).
struct alignas(128) HotCounter { unsigned long long v; unsigned long long pad_[15]; };
struct Counters {          // device-side bump allocators + flags
	HotCounter seed_used_, node_used_, reg_used_;
	HotCounter next_read_, next_read3_;   // work counters of the seeding kernels (passes 1-2, pass 3)
	HotCounter next_vread_;               // ... and of the pass-1 tasks of long-read batches (k_seed<LR = 1>)
	HotCounter n_heavy_, n_p2_tasks_, next_p2_;   // short-read batches: the heavy reads -- given up by the lane-per-read kernel when they exceeded its iteration budget --; their pass-2 searches as tasks (k_seed<LR = 3>) and the work counter over them
	HotCounter n_vr_ovf_, next_vovf_;     // ... tasks whose interval stack outgrew the task lanes' small spill areas (redone on full-size stacks), and the work counter of that second launch
	HotCounter next_ext_;    // work counter of the wave extension kernel (position in Batch::order)
	HotCounter next_seedsw_; // work counter of the wave-per-read seed re-scoring kernel (long reads)
	HotCounter next_chain_, next_dedup_;       // work counters of the chaining and de-duplication kernels
	HotCounter next_pack_;                     // ... and of the packed-extension kernel (k_ext_pack)
	HotCounter n_dd_heavy_, n_dd_big_, next_dd_heavy_, next_dd_big_;    // short-read batches: the reads k_dedup left to the wave-per-read kernel (several regions; more regions than that kernel's LDS copy holds), and the work counter over both
	HotCounter cig_ext_used_;                  // operations written to the batch's CIGAR operation array (records with more than 6 operations)
	unsigned long long intv_used;
	unsigned long long overflow;   // bit0 intv, bit1 seed, bit2 node, bit3 reg, bit4 tmp-intv scratch
	// algorithmic work counters (bwagpu_stats_t)
	unsigned long long n_intv, n_chains, n_regs_raw, n_regs;
	unsigned long long occ_blocks, lf_steps, ext_calls, ext_cells, glb_calls, glb_cells, ref_bases, sw_calls, sw_cells, tab_lookups;
	unsigned long long prof[16];                   // diagnostics (bwagpu_debug_prof): k_seed's stats instance: [2] lane-slots of lanes out of reads, [3] of lanes waiting in a bookkeeping state, [4] of lanes running it, [5] sum over waves of the iteration at which the first lane ran out of reads, [6] iterations of the longest wave, [7] waves that did any work; [8] extensions answered by k_ext_pack; [9] read windows k_seed fetched a step ahead (MRG 2, reads without an LDS copy), [10] k_seed lane steps that take an interval-stack entry from HBM scratch, [11] those served by an entry fetched a step ahead (MRG 2), [12] k_seed iterations that read the interval stack from HBM, [13..15] its wave iterations, bookkeeping iterations, extending lanes (stats runs)
	unsigned long long ext_fast;                   // ksw_extend2 calls answered by the diagonal rule (no DP)
	unsigned long long bt_nodes, chain_recs;       // B-tree nodes visited by look-ups / chain records touched (k_chain's algorithmic bytes)
	unsigned long long wave_hist[2][96];           // stats runs of k_extend_wave [0] / k_dedup_wave [1]: reads by floor(log2(time the wave spent on the read, in 10 ns units)) + 1; then, per bin, the DP calls and the DP cells (>> 10) of those reads (bwagpu_debug_hist)
	unsigned long long chain_hist[3][32];          // stats runs of k_chain_wave: reads by min(31, chains before the filter / 16); row 0: chained in registers, row 1: in the B-tree, row 2: 10 ns ticks by phase (CW_PHASE, dev_chainw.h) (bwagpu_debug_chain_hist)
	unsigned long long chain_seeds[3][32];         // ... and by min(31, seeds / 32)
	unsigned long long seed_hist[64];              // k_seed's stats instance: reads by floor(log2(iterations spent on the read)) + 1, then the iterations summed per bin (bwagpu_debug_hist)
	unsigned long long seed_x2[8];                 // k_seed's stats instance: extension steps that read index blocks, [0] forward / [1] backward in all, [2] / [3] those on an interval of ONE row (a unique match: the step is a comparison with the next text base), [4] / [5] forward / backward runs of such steps (maximal, per search), [6] prefix-table steps (bwagpu_debug_seed_x2)
	unsigned long long cigl_plan[2];               // k_cigar_long_plan: regions left to the long CIGAR tier, bytes of the largest direction matrix among them
};
#define seed_used seed_used_.v
#define node_used node_used_.v
#define reg_used reg_used_.v
#define next_read next_read_.v
#define next_read3 next_read3_.v
#define next_vread next_vread_.v
#define n_vr_ovf n_vr_ovf_.v
#define n_heavy n_heavy_.v
#define n_p2_tasks n_p2_tasks_.v
#define next_p2 next_p2_.v
#define next_vovf next_vovf_.v
#define next_ext next_ext_.v
#define next_seedsw next_seedsw_.v
#define next_chain next_chain_.v
#define next_dedup next_dedup_.v
#define next_pack next_pack_.v
#define n_dd_heavy n_dd_heavy_.v
#define next_dd_heavy next_dd_heavy_.v
#define n_dd_big n_dd_big_.v
#define next_dd_big next_dd_big_.v
#define cig_ext_used cig_ext_used_.v

// Sub-arrays of one read's private region (n = its number of seed slots); offsets keep every array naturally aligned.
struct RegionView {
	ChainRec *chain;          // chain pool                                   [0, 64n)
	int4 *kinfo;              // packed {beg, end, weight, flags} of kept chains  [64n, 80n)
	bwagpu_chain_t *cchain;   // kept chains (headers)                        [80n, 112n)
	bwagpu_seed_t *cseed;     // seeds of the kept chains, chain by chain     [112n, 136n)
	u64 *srt;                 // sort keys (chain filter, mem_chain2aln)      [136n, 144n)
	i32 *next, *ord, *kept;   // seed links, chain order, kept list           [144n, 156n)
};
DEVFN RegionView region_of(u8 *blob, i64 seed_off, int n)
{
	u8 *b = blob + seed_off * SLOT_BLOB_BYTES;
	RegionView v;
	v.chain = (ChainRec*)b; v.kinfo = (int4*)(b + (size_t)64 * n); v.cchain = (bwagpu_chain_t*)(b + (size_t)80 * n);
	v.cseed = (bwagpu_seed_t*)(b + (size_t)112 * n); v.srt = (u64*)(b + (size_t)136 * n);
	v.next = (i32*)(b + (size_t)144 * n); v.ord = (i32*)(b + (size_t)148 * n); v.kept = (i32*)(b + (size_t)152 * n);
	return v;
}

// Everything one batch needs on the device.
struct Batch {
	int n_reads;
	int max_len;               // longest read of the batch
	int stats;                 // collect work counters
	const u8 *seq;             // concatenated nt4 codes
	u64 *seq_nib;              // the same bases at 4 bits each, 16 per word (k_pack_reads; read by the seeding kernel)
	u64 seq_nib_bytes;         // ... its size (buffer descriptor of k_seed<RD = false, MRG = 2>)
	const i64 *off;            // n_reads + 1
	u32 *seq_2b;               // [n_reads][rd_words]: every read's bases at 2 bits each from a word boundary of its own (k_pack_reads2b) ...
	u8 *seq_flags;             // ... and per read: 1 = the read holds an N (its lane then reads bases from seq_nib)
	int rd_words;              // words per read in seq_2b (a multiple of 4), 0: reads too long, no LDS copy
	Counters *ctr;
	// --- seeding scratch: per resident lane one interval stack (first entries in LDS, see SeedStack)
	BiIntv *tmp_intv;          // [n_seed_threads][max_len+1]: spill area of the lanes' interval stacks
	int seed_stack_cap;        // entries of a lane's spill area (0: max_len + 1 + PTAB_MAX, the worst case; the task launch of long-read batches: small)
	int seed_lds_ent;          // stack entries per lane kept in LDS (0 when seq_len >= 2^37 or max_len >= 2^16: the packing would not fit)
	int seed_no_virt;          // diagnostics: keep short matches in the stack as well (see SeedLane::smask)
	int mem_cap;               // capacity of one read's interval list
	// --- seeding results
	i32 *seed_w;               // per read: repetitiveness weight left by k_seed3 (sum of seed-length match occurrences)
	const i32 *seed_order;     // processing order of k_seed (heaviest first by seed_w), or null: input order
	i32 *intv_n;               // per read
	i64 *intv_off;             // per read, into intv[]
	Intv3 *intv;               // [n_reads][mem_cap]: read r's SA intervals at intv + r * mem_cap (intv_off[r] = r * mem_cap)
	// --- slot space (one slot per SA lookup / seed)
	i32 *seed_n;               // per read
	i64 *seed_off;             // per read
	i64 slot_cap;
	u64 *slot_pos;             // SA row before the lookup kernel, reference position (rbeg) after it
	i32 *slot_qbeg, *slot_len; // the seed's query start and length (copied from its interval by k_seed)
	i32 *slot_rid;             // contig id of the seed, -1/-2 if it bridges contigs/strands (filled by k_sa)
	// Per-read private work area of the chaining / extension stages: ONE contiguous block of SLOT_BLOB_BYTES * n_seeds bytes
	// per read at slot_blob + seed_off * SLOT_BLOB_BYTES, carved into the sub-arrays below by RegionView.  A lane (or wave)
	// working on a read then touches one compact region instead of a dozen arena-wide arrays (fewer TLB entries / DRAM pages
	// per read, and the next read of the lane starts in a fresh region).
	u8 *slot_blob;
	i32 *chain_n;              // per read: chains after filtering
	// --- pass 1 of long-read batches as independent tasks (k_seed<LR = 1>, option seed_tasks): task t of read r searches position (t - vr_first[r]) * task_step
	int task_step, n_vreads;        // min_seed_len; number of tasks of the batch
	int task_tpr;                   // > 0 (short-read batches, the heavy reads only): task t = position (t % task_tpr) * task_step of read heavy_list[t / task_tpr], for t < n_heavy * task_tpr
	i32 *heavy_list;                // the reads the lane-per-read kernel gave up after seed_budget iterations (n_heavy of them); their passes 1-2 run as tasks
	i32 *dd_list;                   // k_dedup: the reads with at least dd_heavy_min regions, left to k_dedup_wave<.., LIST = true>: n_dd_heavy of them from the front, and from the
	                                // back (entry n_reads - 1 downwards) the n_dd_big reads with more than dd_stage_cap regions
	int dd_heavy_min, dd_stage_cap; // dd_heavy_min 0 = k_dedup does every read itself
	int dd_net;                     // reads of at least this many regions finish their sorts by a sorting network instead of by counting (option dedup_net; 0: never)
	int dd_prio;                    // the list launches' waves run at raised issue priority (option dedup_prio)
	int seed_budget;                // ... that budget (0: none)
	i64 *p2_tasks; long long p2_cap; // pass-2 searches of the heavy reads as tasks (k_seed<LR = 3>): read << 32 | index of the pass-1 entry to re-seed
	const i32 *vr_first;            // per read: its first task (n_reads + 1 entries)
	i32 *vr_ovf;                    // tasks to be redone on full-size interval stacks (n_vr_ovf of them)
	int vr_ovf_run, vr_room;        // this launch: 1 = redo the tasks of vr_ovf; interval-stack entries a lane may hold before its task counts as overflowed
	i32 *intv_n3;                   // per read: the entries pass 3 (k_seed3, run first) left at the head of its interval list
	int seed_prio;             // waves holding the heaviest 3 % of k_seed's reads run at raised issue priority (option seed_prio = 0 turns it off)
	int seed_pass3_inline;     // A/B switch (option seed_pass3_inline = 1): pass 3 inside k_seed's state machine as in round 1, instead of k_seed3
	// --- B-tree nodes
	i64 *node_off;             // per read
	i32 *nodes; i64 node_cap;  // 21 ints per node
	// --- alignment regions
	i64 *reg_off;              // per read
	i32 *reg_cap_r;            // per read capacity (= seeds in kept chains)
	i32 *reg_n_raw;            // per read: regions after mem_chain2aln
	i32 *reg_n;                // per read: regions after mem_sort_dedup_patch
	bwagpu_alnreg_t *regs; i64 reg_cap;
	bwagpu_alnreg_t *regs_raw; // copy of the pre-dedup regions when taps are enabled (else null)
	// --- DP scratch: per resident wave, interleaved by lane: [(j * 64) + lane]
	i32 *dp_h, *dp_e;          // (max_len + 2) * 64 ints per wave each
	int dp_waves;
	// --- per-length table for mem_flt_chained_seeds: min HSP score, or -1 when the stage is off
	const i32 *seedsw_minhsp;
	// --- heavy-first processing order of the reads (k_order_*): reads binned by log2(weight), heaviest bin first, so
	// that the few reads with thousands of seeds start first and lanes of a wave get reads of similar cost
	int chain_flt_lds;         // chains up to which the chain filter's arrays live in LDS (<= CW_FLT_LDS)
	int chain_regs;            // option chain_regs: 0 = every read is chained in the B-tree form, 1 = register form up to 64 chains, 2 = up to 256
	int ext_blk;               // long reads (ring mode of k_extend_wave): rows of up to 255 columns in one pass, four columns per lane (option ext_blk)
	int ext_plan;              // k_ext_pack has run: every chain's ExtPlan record (dev_extp.h) is in place of the chain pool, k_extend_wave takes windows, seed orders and answered extensions from there
	i32 *order;                // [n_reads] permutation of read indices
	u32 *bin_cnt;              // [2 * ORDER_BINS]: counts, then fill cursors / starts
};
