// bwagpu_config.h -- the tuning and test options of a handle (host code; bwagpu_set_option / bwagpu_get_option / bwagpu_set_default_option,
// include/bwagpu.h).
//
// Until round 3 every one of these was a getenv() inside the batch calls -- a dozen per batch, and nothing a reference-side caller could bind.
// Now they are fields of a plain struct that every handle owns: filled when the handle is created -- compiled-in defaults, then the process
// environment (BWAGPU_<NAME IN CAPITALS>, read ONCE per bwagpu_create*; the tools' and tests' old switches keep working), then whatever
// bwagpu_set_default_option() was given -- copied by bwagpu_clone*(), and changed per handle with bwagpu_set_option().  No batch call reads the
// environment.  None of the options changes a result: they choose between kernel forms that the tests hold to the same output, size scratch
// areas, or force the overflow/retry paths.
//
// -1 ("auto") for the long-read kernel forms means: on for batches whose longest read exceeds the short-read extension kernel's limit
// (1100 bp), off otherwise -- the setting BENCH_r03's `variants` measured as the fastest for each class of batch.
#pragma once
#include <stdlib.h>
#include <string.h>
#include <ctype.h>
#include <string>

//      name               default   meaning
#define BWAGPU_OPTION_LIST(X) \
	X(occ32,             1)    /* index: kernels read the 32-byte block layout derived at load time (0: the reference-format 64-byte blocks)          */ \
	X(occ32_sb_shift,    32)   /* index: log2 bases per superblock of that layout (tests: small superblocks on small genomes)                           */ \
	X(ptab_m,            10)   /* index: depth of the prefix tables (0: none)                                                                           */ \
	X(idx_desc_max_mb,   0)    /* index: tables above this many MiB count as beyond a buffer descriptor's reach (0: the hardware's 4 GiB; test hook for the fallback kernels) */ \
	X(seed_mrg,          -1)   /* seeding: 0 = loads as the compiler schedules them, 2 = one memory round trip per iteration; auto: 2                    */ \
	X(seed_tasks,        -1)   /* seeding: pass 1 of long reads as independent tasks, one per min_seed_len-th position (0: the lane-per-read chain); auto: on for long reads */ \
	X(seed_budget,       -1)   /* seeding, short reads: iterations after which the lane-per-read kernel gives a read up to the task kernels; auto: 6144 for a batch alone on the chip, 12288 next to other batches (as `share`), 0: never           */ \
	X(seed_p2_cap,       0)    /* seeding: entries of the heavy reads' pass-2 task list (0: 16 per heavy read; tests: a tiny list forces the retry)                             */ \
	X(seed_task_stack,   0)    /* seeding: packed interval-stack entries a task lane may spill (0: 256; tests: tiny stacks force the second launch)      */ \
	X(publish_blk,       -1)   /* interval sort + SA-row expansion by one workgroup per read; auto: on for long reads                                   */ \
	X(seedsw_lds,        -1)   /* mem_flt_chained_seeds' local alignments with their state in LDS; auto: on for long reads                              */ \
	X(dedup_blk,         -1)   /* patch alignments of k_dedup_wave with four columns per lane; auto: on (the wave kernel only runs for long reads)      */ \
	X(seed_lds_ent,      -1)   /* seeding: interval-stack entries per lane kept in LDS; auto: 10, or what the read copy leaves                          */ \
	X(seed_rd_lds,       1)    /* seeding: short reads copied to LDS at 2 bits per base                                                                 */ \
	X(seed_no_virt,      0)    /* seeding: keep matches shorter than the prefix tables' depth in the stack too (diagnostics)                            */ \
	X(seed_prio,         1)    /* seeding: raised issue priority for the waves holding the heaviest reads                                               */ \
	X(seed_input_order,  0)    /* seeding: reads in input order instead of heaviest first (diagnostics)                                                 */ \
	X(seed_pass3_inline, 0)    /* seeding: pass 3 inside k_seed's state machine instead of k_seed3 (A/B)                                                */ \
	X(seed_grid,         0)    /* seeding: resident workgroups of k_seed (0: fill the chip; measurements)                                               */ \
	X(share,             -1)   /* short reads: percent of a chip-filling launch that the hot path's persistent kernels take (kernels of different batches side by side); auto: 50 when three or more handles share the index, else 100 */ \
	X(ext_pack,          0)    /* extension, short reads: 1 / 5 = the chains' first extensions four to a wavefront ahead of k_extend_wave (k_ext_pack at 4 / 5 waves per SIMD, dev_extp.h); measured slower than a wavefront per extension (profiles/r06_ext_pack.md): off, kept as an A/B switch */ \
	X(ext_blk,           -1)   /* extension, long reads: a row's band (up to 255 columns) in one pass, four columns per lane; auto: on (0: one pass per 64 columns, the round-3..5 form) */ \
	X(ext_occ,           6)    /* extension: waves per SIMD the short-read kernel's register allocation aims at (4 or 6)                                */ \
	X(chain_regs,        2)    /* chaining: the chains of a read kept in registers, one per lane, while that is exact: 2 = up to 256 chains (four per lane), 1 = up to 64, 0 = every read through the B-tree (A/B and tests) */ \
	X(chain_flt_lds,     256)  /* chaining: reads of up to this many chains keep the weight sort's and the chain filter's arrays in LDS (at most 256; tests: 0 sends every read down the HBM path) */ \
	X(dedup_heavy,       -1)   /* short reads: k_dedup leaves reads with at least this many regions to the wave-per-read kernel; 0 = none, auto: 3      */ \
	X(dedup_stage,       -1)   /* ... with the decisions' operands in LDS for reads of up to this many regions (first launch); 0 = every read in place in HBM, auto: 128 */ \
	X(dedup_big,         -1)   /* ... and for the reads with more, up to this many (second launch, 64 KB of LDS per wave); 0 = those in place, auto: what fits (893) */ \
	X(dedup_net,         -1)   /* ... and finishes the sorts of reads with at least this many regions by a bitonic network instead of by counting; 0 = never, auto: 129 */ \
	X(dedup_prio,        1)    /* ... whose waves run at raised issue priority (0: A/B)                                                              */ \
	X(dedup_wave,        0)    /* 1 = the wave-per-read de-duplication kernel for short reads as well (test hook)                                       */ \
	X(dedup_ring,        0)    /* ring columns of that kernel (0: from the batch; test hook: a power of two, 256..4096)                                 */ \
	X(mem_cap,           0)    /* capacity of a read's interval list (0: from the batch; test hook: a small value forces the retry path)                */ \
	X(cig_tiers,         2)    /* CIGAR stage: LDS tiers to run (diagnostics)                                                                           */ \
	X(cig_ops_cap,       0)    /* CIGAR stage: entries of the operation array (0: from the batch; test hook: forces the second attempt)                 */ \
	X(cig_long,          1)    /* CIGAR stage: the long-segment tier (k_cigar_long)                                                                     */ \
	X(cigl_mib,          32768)/* CIGAR stage: scratch budget of that tier in MiB (32 GiB: 1024 direction matrices of a 10 kb read's widest band)                                                                      */ \
	X(cig_trace,         0)    /* CIGAR stage: print the launches' times (waits for the stream after each)                                              */ \
	X(debug_sync,        0)    /* wait and report after every stage of bwagpu_batch_run                                                                 */ \
	X(reserve_results,   1)    /* bwagpu_batch_reserve also page-locks the result blocks of a batch of that shape (0: the first download does; A/B)      */ \
	X(pinned_results,    1)    /* PROCESS-WIDE (the result pool is shared by all handles): large results in pooled page-locked blocks (0: plain malloc)  */ \
	X(pinned_min_kb,     1024) /* PROCESS-WIDE: results below this size come from malloc (tests: 0 pools everything)                                    */ \
	X(pinned_cap_mb,     16384)/* PROCESS-WIDE: bound of the page-locked result pool in MiB (a handle's batch of 667 k reads holds ~0.4 GB of results; five handles two batches ahead outgrew the 4 GiB of round 5 and fell back to pageable copies) */

struct BwagpuConfig {
#define X(name, dflt) long long name = dflt;
	BWAGPU_OPTION_LIST(X)
#undef X
	long long *field(const char *key)
	{
		if (!key) return nullptr;
#define X(name, dflt) if (!strcmp(key, #name)) return &name;
		BWAGPU_OPTION_LIST(X)
#undef X
		return nullptr;
	}
	// BWAGPU_<NAME>=<integer> for every option present in the environment (BWAGPU_CIGL_GIB, a possibly fractional number of GiB, is
	// accepted for cigl_mib as well)
	void from_env()
	{
#define X(name, dflt) { std::string e = "BWAGPU_"; for (const char *p = #name; *p; ++p) e += (char)toupper((unsigned char)*p); if (const char *v = getenv(e.c_str())) if (*v) name = atoll(v); }
		BWAGPU_OPTION_LIST(X)
#undef X
		if (const char *v = getenv("BWAGPU_CIGL_GIB")) if (*v) cigl_mib = (long long)(atof(v) * 1024.);
	}
};
