/* include/bwagpu.h -- C-ABI of the MI355X-native BWA-MEM alignment core (libbwagpu.so).
 *
 * The library replaces exactly one thing in lh3/bwa: the first parallel loop of mem_process_seqs()
 *     kt_for(opt->n_threads, worker1, &w, n)                       (reference bwamem.c:1252)
 * i.e. "for every read i: regs[i] = mem_align1_core(opt, bwt, bns, pac, l_seq, seq, aux)"
 * (reference bwamem.c:1081-1117, 1203-1215).  Everything it computes on the way -- SMEM seeding over the
 * FM-index (bwt.c:189-379), suffix-array lookup (bwt.c:53-96), chaining and chain filtering
 * (bwamem.c:216-411), banded seed extension (ksw.c:416-515 via bwamem.c:658-812) and region
 * de-duplication/patching (bwamem.c:432-515, ksw.c:540-642) -- runs as HIP kernels on gfx950 with the index
 * resident in HBM.  Results (mem_alnreg_t records) are bit-identical to the reference's.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types cross this boundary;
 *   - every function returns 0 on success or a negative BWAGPU_E* code (the reference itself has no error
 *     returns on this path: it exit()s via err_fatal, utils.c:90-122; a drop-in wrapper prints
 *     bwagpu_strerror() and exits to mimic that);
 *   - a handle is single-caller / non-reentrant, like step 1 of the reference's kt_pipeline
 *     (kthread.c:93-104 guarantees one batch at a time in mem_process_seqs);
 *   - there is NO CPU fallback: without a HIP device bwagpu_create() fails with BWAGPU_ENODEV.
 *
 * Struct layouts below are byte-for-byte those of the reference (checked by tests against the compiled
 * reference): bwagpu_opt_t == mem_opt_t (bwamem.h:52-84, 168 B), bwagpu_alnreg_t == mem_alnreg_t
 * (bwamem.h:86-104, 88 B), bwagpu_alnreg_v == mem_alnreg_v (bwamem.h:106), bwagpu_bseq1_t == bseq1_t
 * (bwa.h:58-61, 48 B).  A reference-side caller may pass its own structs through a pointer cast.
 */
#ifndef BWAGPU_H
#define BWAGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BWAGPU_OK        0
#define BWAGPU_ENODEV   -1  /* no usable HIP device / HIP runtime error at start-up */
#define BWAGPU_EINVAL   -2  /* bad argument */
#define BWAGPU_ENOMEM   -3  /* host or device allocation failed */
#define BWAGPU_EIO      -4  /* index files missing or inconsistent */
#define BWAGPU_EHIP     -5  /* HIP runtime error during a batch (see bwagpu_last_error) */
#define BWAGPU_EUNSUP   -6  /* option combination not implemented on the device path yet */

typedef struct bwagpu_s bwagpu_t;

/* == mem_opt_t, reference bwamem.h:52-84 */
typedef struct {
	int a, b;
	int o_del, e_del;
	int o_ins, e_ins;
	int pen_unpaired;
	int pen_clip5, pen_clip3;
	int w;
	int zdrop;
	uint64_t max_mem_intv;
	int T;
	int flag;
	int min_seed_len;
	int min_chain_weight;
	int max_chain_extend;
	float split_factor;
	int split_width;
	int max_occ;
	int max_chain_gap;
	int n_threads;
	int chunk_size;
	float mask_level;
	float drop_ratio;
	float XA_drop_ratio;
	float mask_level_redun;
	float mapQ_coef_len;
	int mapQ_coef_fac;
	int max_ins;
	int max_matesw;
	int max_XA_hits, max_XA_hits_alt;
	int8_t mat[25];
} bwagpu_opt_t;

/* == mem_alnreg_t, reference bwamem.h:86-104 */
typedef struct {
	int64_t rb, re;
	int qb, qe;
	int rid;
	int score;
	int truesc;
	int sub;
	int alt_sc;
	int csub;
	int sub_n;
	int w;
	int seedcov;
	int secondary;
	int secondary_all;
	int seedlen0;
	int n_comp:30, is_alt:2;
	float frac_rep;
	uint64_t hash;
} bwagpu_alnreg_t;

/* Banded global alignment of one region as mem_reg2aln's band-doubling loop around bwa_gen_cigar2 leaves it
 * (bwamem.c:1143-1152, bwa.c:148-234): score, BAM-style CIGAR (len << 4 | op, op M=0 I=1 D=2) before clipping and before
 * the leading/trailing-deletion squeeze, and the NM / MD values computed from it (bwa.c:196-226).  n_cigar 1..6: the operations are
 * in cigar[]; n_cigar > 6 (up to 32768: a 10 kb read with 13 % indels has ~2500): they are entries [at, at + n_cigar) of the batch's
 * operation array (bwagpu_batch_cigar_ops), at = cigar[1] << 32 | cigar[0].  MD: md_len characters; up to 8 of them are the bytes of `md` (first character in the low byte),
 * longer strings are packed four to an entry (first character in the low byte) at entries [md, md + (md_len + 3) / 4) of the
 * operation array.  n_cigar == -1: not computed on the device (region below opt->T; a band of more than 2048 columns, more than 32768
 * operations or an MD string of more than 96 KiB; `score` then holds the reason 1/2/3, nm is -1) -- the caller runs bwa_gen_cigar2 itself. */
typedef struct {
	int32_t score;
	int32_t n_cigar;
	uint32_t cigar[6];
	int32_t nm;
	int32_t md_len;
	uint64_t md;
} bwagpu_cigar_t;

/* Insert-size window of one orientation as mem_matesw uses it: mem_pestat_t::low/high/failed (bwamem.h:108-112). */
typedef struct { int32_t low, high, failed, pad_; } bwagpu_pes_t;

/* One precomputed mate-rescue alignment: the kswr_t that mem_matesw's ksw_align2 call (bwamem_pair.c:177) returns for
 * aligning read `read` (or its reverse complement) inside the window that anchor position `anchor_rb` on contig
 * `anchor_rid` and orientation r imply.  r == -1: no alignment was due for this task (empty window, other contig, window
 * shorter than min_seed_len or beyond the kernel's limits). */
typedef struct {
	int32_t read, r;
	int64_t anchor_rb;
	int32_t anchor_rid;
	int32_t score, te, qe, score2, te2, tb, qb;
	int32_t pad_, pad2_;     /* zero */
} bwagpu_matesw_t;

/* == mem_alnreg_v, reference bwamem.h:106 */
typedef struct { size_t n, m; bwagpu_alnreg_t *a; } bwagpu_alnreg_v;

/* == bseq1_t, reference bwa.h:58-61 */
typedef struct {
	int l_seq, id;
	char *name, *comment, *seq, *qual, *sam;
} bwagpu_bseq1_t;

/* The reference index as plain arrays.  A reference-side caller fills it from bwt_t (bwt.h:48-60), bntseq_t
 * (bntseq.h:41-64) and the pac pointer of bwaidx_t (bwa.h:48-56); see INTEGRATION.md for the 15-line stub. */
typedef struct {
	/* FM-index: interleaved Occ/BWT words exactly as in bwt_t::bwt (bwtindex.c:150-172) */
	const uint32_t *bwt;     /* bwt_t::bwt      */
	uint64_t bwt_size;       /* bwt_t::bwt_size (number of uint32 words) */
	uint64_t primary;        /* bwt_t::primary  */
	uint64_t L2[5];          /* bwt_t::L2       */
	uint64_t seq_len;        /* bwt_t::seq_len  */
	/* sampled suffix array */
	const uint64_t *sa;      /* bwt_t::sa (sa[0] == (uint64_t)-1) */
	uint64_t n_sa;           /* bwt_t::n_sa     */
	int sa_intv;             /* bwt_t::sa_intv (power of two) */
	/* 2-bit packed forward reference */
	const uint8_t *pac;      /* bwaidx_t::pac, l_pac/4+1 bytes */
	int64_t l_pac;           /* bntseq_t::l_pac */
	/* contig table */
	int32_t n_seqs;          /* bntseq_t::n_seqs */
	const int64_t *ctg_offset;  /* anns[i].offset */
	const int32_t *ctg_len;     /* anns[i].len    */
	const int32_t *ctg_is_alt;  /* anns[i].is_alt */
} bwagpu_index_desc_t;

/* Counters of one batch (for the roofline denominator) and per-stage device times. */
typedef struct {
	int64_t n_reads, n_bases;
	int64_t n_intv;          /* SA intervals kept by mem_collect_intv          */
	int64_t n_seeds;         /* bwt_sa calls (N_sa)                            */
	int64_t n_chains;        /* chains after mem_chain_flt                     */
	int64_t n_regs_raw;      /* regions before mem_sort_dedup_patch            */
	int64_t n_regs;          /* regions returned                               */
	/* filled only when stats collection is enabled (bwagpu_set_stats): */
	int64_t n_occ_blocks;    /* N_blk: 64-byte index blocks touched by seeding */
	int64_t n_lf_steps;      /* N_lf : bwt_invPsi steps inside bwt_sa          */
	int64_t n_ext_calls;     /* ksw_extend2 calls                              */
	int64_t n_ext_cells;     /* ksw_extend2 DP cells                           */
	int64_t n_glb_calls;     /* ksw_global2 (score-only) calls                 */
	int64_t n_glb_cells;     /* ksw_global2 DP cells                           */
	int64_t ref_bases;       /* W_ref: reference bases covered by extension windows */
	int64_t n_sw_calls;      /* mem_seed_sw local alignments (long reads)      */
	int64_t n_sw_cells;
	/* device time per stage, milliseconds (HIP events on the library's stream) */
	float ms_seed, ms_sa, ms_chain, ms_seedsw, ms_extend, ms_dedup, ms_total;
	int32_t n_retries;       /* arena-growth reruns */
	float ms_publish;        /* k_publish + k_expand (interval sort, slot reservation); ms_seed is the k_seed kernel alone */
	int64_t n_tab_lookups;   /* 16-byte prefix-table entries read by seeding in place of index blocks (stats only) */
	int64_t n_bt_nodes;      /* B-tree nodes (160 B) visited by chaining's look-ups (stats only)               */
	int64_t n_chain_recs;    /* chain records (64 B) read or created by chaining (stats only)                  */
	int64_t n_chain_deferred;/* (rounds 2-4: reads whose chaining outgrew the first LDS tier; 0 since round 5 -- one chaining kernel, no tiers) */
	int64_t n_ext_fast;      /* ksw_extend2 calls answered without DP (diagonal rule, dev_extw.h)              */
	int64_t n_chain_deferred2;/* (likewise: always 0)                                                               */
	/* the calls after bwagpu_batch_run, filled by them (HIP events on the handle's stream; 0 until the call has run for this batch): */
	float ms_pack;           /* bwagpu_batch_download: packing the used region records on the device          */
	float ms_download_copy;  /* ... and their device-to-host copy                                              */
	float ms_cigar_kernels;  /* bwagpu_batch_cigars: all its launches (the three tiers, NM/MD)                 */
	float ms_cigar_copy;     /* ... and the copies of the records and (bwagpu_batch_cigar_ops) the operation array */
	int64_t n_cig_cells;     /* bwagpu_batch_cigars (stats only): DP cells of the ksw_global2 fills with traceback, every band-doubling attempt counted */
	int64_t n_cig_dp;        /* ... and the number of such fills (regions answered by the gap-free comparison, bwa.c:171-174, do not count) */
	int32_t retry_mask;      /* OR of the overflow bits that made bwagpu_batch_run redo the batch: 2 slots, 4 B-tree nodes, 8 regions, 16 interval lists, 32 pass-2 task list */
	int32_t reserved_;
} bwagpu_stats_t;

/* Diagnostics: a marker of the step the handle's current (or last) batch call has reached; safe to call from another
 * thread while a call is in progress.  10-12 upload, 20+100*attempt run launched, 22+100*attempt run waiting, 30-39 download,
 * 40-45 cigars. */
int bwagpu_debug_phase(const bwagpu_t *h);
/* Diagnostics: sixteen event counters of the last batch_run; with stats on, [13..15] = wave iterations of the seeding kernel, those that
 * ran its bookkeeping code, and the lanes extending summed over iterations (tools/seed_iter_probe.py). */
int bwagpu_debug_prof(bwagpu_t *h, unsigned long long out[16]);
/* Diagnostics (stats on): where three kernels' time goes, read by read.  out[0..64): the seeding kernel -- out[b] = reads that took [2^(b-1), 2^b)
 * wave iterations, out[32 + b] = their iterations summed (tools/seed_iter_probe.py).  out[64..160): the wave-per-read extension kernel -- reads by
 * the time their wave spent on them (bin b: [2^(b-1), 2^b) x 10 ns), then per bin the ksw_extend2 calls and the DP cells (>> 10) of those reads;
 * out[160..256): the same for the wave-per-read de-duplication kernel (long reads) and its patch alignments. */
int bwagpu_debug_hist(bwagpu_t *h, unsigned long long out[256]);
/* Diagnostics (stats on): the seeding kernel's index-block look-ups by interval size.  out[0] / out[1] = forward / backward extension steps that read
 * index blocks, out[2] / out[3] = those whose interval is a single row (a unique match), out[4] / out[5] = maximal runs of such steps. */
int bwagpu_debug_seed_x2(bwagpu_t *h, unsigned long long out[8]);
/* Diagnostics (stats on): the chaining kernel's reads by size and form.  out[t * 64 + b] = reads whose seeds were chained in form t (0: in registers,
 * 1: in the B-tree) with 16 b .. 16 b + 15 chains (before the chain filter) (b = 31: more); out[t * 64 + 32 + b] = with 32 b .. 32 b + 31 seeds.
 * out[128 .. 136] = the kernel's wave time in 10 ns ticks by phase (seed loop in registers, in the tree, repeat fraction + in-order list, weights, sort,
 * pairwise filter, publishing), the longest read, and the reads counted (tools/seed_iter_probe.py prints all of it). */
int bwagpu_debug_chain_hist(bwagpu_t *h, unsigned long long out[192]);

/* ---- differential tests of the device DP routines ----------------------------------------------------------------------- */
/* One case of bwagpu_debug_dp.  Sequences are nt4 codes in the call's `seqs` array: the query may hold 0..4, the target 0..3 (the
 * device reads targets from 2-bit packed reference text).  flags: bit 0 = present the query back to front, bit 1 = present the
 * target back to front, bit 2 = present the target's complement (read through the reverse-strand half of the text). */
typedef struct {
	int32_t q_off, q_len, t_off, t_len;
	int32_t w;               /* band width */
	int32_t h0;              /* kinds 0,1: ksw_extend2's h0; kind 4: ksw_align2's xtra word */
	int32_t end_bonus;       /* kinds 0,1 */
	int32_t flags;
} bwagpu_dp_case_t;
/* Runs one wavefront of a device DP routine per case, set up exactly as the product kernel sets it up, and returns 72 ints per
 * case.  kind 0: ksw_extend2 as k_extend_wave runs it for short reads (ksw.c:416; columns and read profile in LDS), kind 1: the
 * same in ring mode (long reads) -> {score, qle, tle, gtle, gscore, max_off, answered-without-DP, cells}; kind 2: ksw_global2 with
 * traceback as k_cigar runs it (ksw.c:540) -> {score, n_ops, ops...} (n_ops -1: more than 64 operations, -2: outside the kernel's
 * limits); kind 3: the score-only ksw_global2 of k_dedup_wave -> {score}; kind 4: ksw_align2 as k_matesw_sw runs it (ksw.c:379)
 * -> {score, te, qe, score2, te2, tb, qb}; kind 5: ksw_global2 with traceback as the long-segment kernel k_cigar_long runs it (columns
 * in an LDS ring, direction bytes in HBM, tiled traceback) -> {score, n_ops, up to 70 ops...} (n_ops may exceed 70: the first 70 are returned).  opt supplies the scoring (mat, gap costs, zdrop). */
int bwagpu_debug_dp(bwagpu_t *h, const bwagpu_opt_t *opt, int kind, int n_cases, const bwagpu_dp_case_t *cases, const uint8_t *seqs, int64_t n_seq_bytes, int32_t *out);

/* ---- optional widening past mem_process_seqs' first loop ---- */
/* After bwagpu_batch_download: one bwagpu_cigar_t per downloaded region, in the same order, computed on the device.  They
 * are what worker2's mem_reg2aln (bwamem.c:1119-1152) would compute on the host for that region; a finalize stage can use
 * them instead of calling bwa_gen_cigar2 (the records carry NM and the MD string as well, bwa.c:196-238).  Free with bwagpu_free. */
int bwagpu_batch_cigars(bwagpu_t *h, const bwagpu_opt_t *opt, bwagpu_cigar_t **out, int64_t *n_out);
/* Enable (1) / disable (0, default) a filter in bwagpu_batch_cigars: regions that overlap their read's best region (by mask_level, as
 * mem_mark_primary_se judges overlap, bwamem.c:519-545) and score below XA_drop_ratio times its score are not computed (reason 1).  Such a
 * region is neither printed nor listed in an XA tag (bwamem_extra.c:118-134) unless a third region stands between the two, so a finalize stage
 * that treats the records as hints loses nothing but the rare recomputation -- and the device skips its most expensive alignments. */
int bwagpu_set_cigar_filter(bwagpu_t *h, int enable);
/* The operation array of the last bwagpu_batch_cigars call (records with more than 6 operations point into it).  Free with bwagpu_free. */
int bwagpu_batch_cigar_ops(bwagpu_t *h, uint32_t **ops, int64_t *n_ops);

/* After bwagpu_batch_download of a paired batch (mates interleaved 2i, 2i+1): the local alignments mem_matesw
 * (bwamem_pair.c:137-206) would run for every (anchor region, orientation) that the downloaded region lists do not already
 * satisfy, given the batch's insert-size windows.  A finalize stage looks results up by (read, anchor_rb, anchor_rid, r)
 * and runs ksw_align2 itself where there is none.  Free with bwagpu_free. */
int bwagpu_batch_matesw(bwagpu_t *h, const bwagpu_opt_t *opt, const bwagpu_pes_t pes[4], bwagpu_matesw_t **out, int64_t *n_out);

/* ---- index construction on the device -------------------------------------------------------- */
/* The arrays `bwa index` leaves in bwt_t after bwt_bwtgen2/bwt_pac2bwt + bwt_bwtupdate_core + bwt_cal_sa
 * (bwtindex.c:64-120, 150-172; bwt.c:62-84), built from the 2-bit packed forward strand by a suffix sort in HBM
 * (bwagpu_index.hip).  bwt/sa are malloc()ed; free with bwagpu_built_free.  Written with the 40/56-byte headers of
 * bwt_dump_bwt / bwt_dump_sa (bwt.c:385-407) they are byte-identical to the reference's .bwt/.sa files. */
typedef struct {
	uint32_t *bwt;           /* bwt_t::bwt: Occ checkpoints interleaved with 2-bit symbols, bwt_size words */
	uint64_t bwt_size;
	uint64_t *sa;            /* bwt_t::sa: n_sa entries, sa[0] = (uint64_t)-1 */
	uint64_t n_sa;
	int sa_intv;
	uint64_t primary, L2[5], seq_len;
	float build_ms;          /* device time of the whole construction (HIP events) */
} bwagpu_built_t;
/* pac: the reference's .pac layout (bntseq.c:229-230), forward strand, l_pac bases, no ambiguity codes (bns_fasta2bntseq
 * replaces them before packing, bntseq.c:266,295-296).  sa_intv: power of two (the reference uses 32, bwtindex.c:316).
 * On failure a message is copied to errbuf (may be NULL). */
int bwagpu_index_build(const uint8_t *pac, int64_t l_pac, int sa_intv, int device, bwagpu_built_t *out, char *errbuf, size_t errlen);
void bwagpu_built_free(bwagpu_built_t *b);

/* ---- lifetime ------------------------------------------------------------------------------------------ */

/* Create a handle on HIP device `device` and upload the index once (replaces nothing in the reference; it is
 * the extra call a drop-in adds after bwa_idx_load, fastmap.c:362-368).  The descriptor's arrays are copied to
 * HBM; the caller keeps ownership of its host copies. */
int bwagpu_create(bwagpu_t **h, const bwagpu_index_desc_t *idx, int device);

/* Multi-GPU start-up: the index is uploaded once by one rank and broadcast to the others over
 * RCCL/xGMI.  A receiving rank calls bwagpu_create() with bwt = sa = pac = NULL (sizes, scalars and the small contig
 * table filled in): the HBM buffers are allocated but left for the collective to fill.  bwagpu_index_buffers() exposes
 * the three device buffers (pointer + byte size) on both sides; bwagpu_index_export() returns the scalars and contig
 * table of a loaded handle so that they can be sent to the other ranks. */
int bwagpu_index_buffers(bwagpu_t *h, void **bwt, uint64_t *bwt_bytes, void **sa, uint64_t *sa_bytes, void **pac, uint64_t *pac_bytes);
int bwagpu_index_export(const bwagpu_t *h, bwagpu_index_desc_t *scalars, int64_t *ctg_offset, int32_t *ctg_len, int32_t *ctg_is_alt);
/* After the broadcast has filled a receiving handle's buffers: derive the device-side acceleration tables from them. */
int bwagpu_index_ready(bwagpu_t *h);

/* Same, reading <prefix>.bwt/.sa/.pac/.ann/.amb/.alt from disk in the reference's on-disk formats
 * (replaces bwa_idx_load_from_disk(hint, BWA_IDX_ALL), bwa.c:289-321, for a stand-alone host). */
int bwagpu_create_from_files(bwagpu_t **h, const char *prefix, int device);

/* Another handle on the same GPU sharing the resident index (no copy), with its own stream and batch arenas.  Two handles
 * driven from two host threads keep two batches in flight (the kt_pipeline of the reference overlaps I/O the same way,
 * kthread.c:119).  Clone after bwagpu_densify_sa, not before.  Each handle is destroyed separately. */
int bwagpu_clone(bwagpu_t *src, bwagpu_t **out);
/* A handle on another device of the node with its own copy of src's index, copied device to device over xGMI (hipMemcpyPeer): the
 * single-process counterpart of the RCCL index broadcast between processes.  Densify the SA on src first. */
int bwagpu_clone_to_device(bwagpu_t *src, int device, bwagpu_t **out);
void bwagpu_destroy(bwagpu_t *h);
const char *bwagpu_strerror(int code);
const char *bwagpu_last_error(const bwagpu_t *h);
const char *bwagpu_version(void);
/* sizeof of bwagpu_opt_t, _alnreg_t, _stats_t, _index_desc_t, _cigar_t, _matesw_t, _bseq1_t, _built_t in this build of the library */
void bwagpu_abi_sizes(int32_t out[8]);

/* Index facts (for callers that loaded from files). */
int bwagpu_index_info(const bwagpu_t *h, int64_t *l_pac, int32_t *n_seqs, uint64_t *seq_len, int *sa_intv);

/* Optional: replace the sampled SA (interval sa_intv, ~31 dependent index reads per lookup, bwt.c:86-96) by a
 * denser one built on the device (new_intv in {1,2,4,8,16}; values identical by definition of the SA).  Spends
 * HBM capacity to delete dependent loads. */
int bwagpu_densify_sa(bwagpu_t *h, int new_intv);

/* ---- options ------------------------------------------------------------------------------------------------
 * Tuning and test options of a handle, by name (the list with defaults and meanings: bwa_amd/csrc/bwagpu_config.h; bwagpu_option_name(i)
 * enumerates it).  None of them changes a result -- they pick between kernel forms that the tests hold to identical output, size scratch
 * areas, or force the overflow/retry paths.  A handle's options are fixed when it is created: compiled-in defaults, then the environment
 * (BWAGPU_<NAME IN CAPITALS>, read once per bwagpu_create*), then bwagpu_set_default_option(); bwagpu_clone*() copies them;
 * bwagpu_set_option() changes one between batches.  No batch call reads the environment.  (The reference has no counterpart: its tuning
 * lives in mem_opt_t, which this library takes as it is.)
 *   bwagpu_set_default_option: for handles created afterwards by this process -- the way to set the options that shape what is derived from
 *   the index at load time (occ32, occ32_sb_shift, ptab_m); bwagpu_clear_default_options() forgets them all.
 *   Unknown name -> BWAGPU_EINVAL.  -1 means "automatic" for the options that have such a setting. */
int bwagpu_set_option(bwagpu_t *h, const char *name, long long value);
int bwagpu_get_option(const bwagpu_t *h, const char *name, long long *value);
int bwagpu_set_default_option(const char *name, long long value);
void bwagpu_clear_default_options(void);
int bwagpu_option_name(int i, const char **name);    /* i = 0, 1, ... until BWAGPU_EINVAL */

/* Enable (1) / disable (0) collection of the algorithmic work counters in bwagpu_stats_t. */
int bwagpu_set_stats(bwagpu_t *h, int enable);
int bwagpu_get_stats(const bwagpu_t *h, bwagpu_stats_t *out);

/* ---- the hot path -------------------------------------------------------------------------------------- */

/* Drop-in replacement for the worker1 loop (reference bwamem.c:1203-1215, 1252).
 *   seqs[i].seq holds ASCII bases or 0..4 codes; on return it holds 0..4 codes (the in-place mutation
 *   contract of mem_align1_core, bwamem.c:1087-1088).  regs[i] receives {n, m, a} with a malloc()ed array
 *   the caller free()s, exactly as worker2 does (bwamem.c:1227,1231).  With MEM_F_PE set in opt->flag nothing
 *   changes on this path: both mates are aligned independently (bwamem.c:1209-1213).
 *   With n = 1 it is also the device half of mem_align1 (bwamem_extra.c:102-112; example.c:40): the binding copies the sequence,
 *   calls this and then the reference's own mem_mark_primary_se (integration/mem_process_seqs_gpu.c, __wrap_mem_align1). */
int bwagpu_align_bseq(bwagpu_t *h, const bwagpu_opt_t *opt, int n, bwagpu_bseq1_t *seqs, bwagpu_alnreg_v *regs);

/* Flat form of the same call: reads are nt4 codes (0..4) concatenated in `seqs`, read i = seqs[off[i]..off[i+1]).
 *   counts[i] = number of regions of read i; *regs_out = array of all regions in read order
 *   (caller frees with bwagpu_free -- NOT free(): large results are page-locked blocks of a pool that bwagpu_free refills,
 *   BWAGPU_PINNED_RESULTS=0 turns that off); *n_regs_out = total. */
int bwagpu_align_flat(bwagpu_t *h, const bwagpu_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off,
					  int32_t *counts, bwagpu_alnreg_t **regs_out, int64_t *n_regs_out);
/* A host buffer for a batch's base codes (what bwagpu_batch_upload reads): page-locked when the runtime grants it, from the same pool as
 * the result arrays, so that the upload is one DMA instead of a staged copy; plain memory otherwise.  Release with bwagpu_free.
 * (No counterpart in the reference: bseq1_t::seq is malloc'ed by bseq_read, bwa.c:79-112.) */
void *bwagpu_alloc_host(size_t bytes);
void bwagpu_free(void *p);   /* releases any array an entry point of this library returned through an out-pointer (thread-safe) */
/* The pool behind bwagpu_alloc_host and the large result arrays keeps up to 4 GiB of page-locked blocks for re-use (options pinned_results,
 * pinned_min_kb).  bwagpu_trim() gives the idle ones back to the system; it happens by itself when the process's last handle is destroyed.
 * Blocks are pinned under the calling thread's current device (the batch calls set their handle's); they are portable across devices. */
void bwagpu_trim(void);

/* Split form for callers that overlap transfers with compute, and for measuring the device path with the batch
 * already resident in HBM: upload -> run (device only, asynchronous kernels + one final sync) -> download. */
int bwagpu_batch_upload(bwagpu_t *h, int n, const uint8_t *seqs, const int64_t *off);
/* Optional, any time: allocate the device buffers a batch of about this shape will need (they are only ever grown), so that the handle's
 * first batch does not pay for them inside a pipeline.  Unless option reserve_results is 0 it also page-locks the result blocks of such a batch
 * (about 150 bytes per region at 4 regions per read) and hands them to the pool bwagpu_free() feeds, where the batch's downloads find them. */
int bwagpu_batch_reserve(bwagpu_t *h, int n_reads, int64_t n_bases, int max_len);
/* Bytes of device memory the handle's buffers would grow by for a batch of this shape (what bwagpu_batch_reserve would allocate now; -1 on bad
 * arguments), and the device's free / total memory (hipMemGetInfo).  `bwa-amd mem` sizes the dense suffix array with them: it takes the smallest
 * SA interval that leaves room for its handles' batches. */
int64_t bwagpu_batch_footprint(bwagpu_t *h, int n_reads, int64_t n_bases, int max_len);
int bwagpu_mem_info(bwagpu_t *h, uint64_t *free_bytes, uint64_t *total_bytes);
int bwagpu_batch_run(bwagpu_t *h, const bwagpu_opt_t *opt);
int bwagpu_batch_download(bwagpu_t *h, int32_t *counts, bwagpu_alnreg_t **regs_out, int64_t *n_regs_out);

/* ---- stage taps (parity tests; device results of the last batch_run, read order) ------------------------ */
/* Taps are on by default; switching them off (0) skips the copy of the pre-dedup regions kept for
 * bwagpu_tap_regs_raw (a benchmark setting). */
int bwagpu_set_taps(bwagpu_t *h, int enable);
/* SA intervals after mem_collect_intv (bwamem.c:140-188): per read counts + records {x0, x2, info}. */
typedef struct { uint64_t x0, x2, info; } bwagpu_intv_t;
int bwagpu_tap_intervals(bwagpu_t *h, int32_t *counts, bwagpu_intv_t **out, int64_t *n_out);
/* Chains after mem_chain_flt (+ mem_flt_chained_seeds): per read chain counts, chain headers, flat seeds. */
typedef struct { int32_t n_seeds, rid, w, kept, is_alt; float frac_rep; int64_t pos; } bwagpu_chain_t;
typedef struct { int64_t rbeg; int32_t qbeg, len, score, pad_; } bwagpu_seed_t;
int bwagpu_tap_chains(bwagpu_t *h, int32_t *counts, bwagpu_chain_t **chains, int64_t *n_chains,
					  bwagpu_seed_t **seeds, int64_t *n_seeds);
/* Regions after mem_chain2aln, before mem_sort_dedup_patch. */
int bwagpu_tap_regs_raw(bwagpu_t *h, int32_t *counts, bwagpu_alnreg_t **out, int64_t *n_out);

#ifdef __cplusplus
}
#endif
#endif
