/* oracle/ref_shim.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin ctypes-friendly shim that is linked together with the *unmodified* reference objects
 * (compiled from /root/reference by oracle/Makefile) into oracle/_ref/libbwaref.so.  It lets the
 * test-suite run the reference's own stage functions (all exported by libbwa: bwamem.c / bwt.c /
 * ksw.c) read-by-read and dump their intermediate results, so that both the plain-C restatement in
 * oracle/orc_*.c and the HIP path can be compared with the real thing stage by stage.
 *
 * Nothing here is part of the product; nothing here re-implements reference logic.  The two private
 * structs of bwamem.c (mem_seed_t / mem_chain_t, bwamem.c:194-208; smem_aux_t, bwamem.c:119-121) are
 * re-declared because the reference does not export them in a header.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <stdio.h>
#include "bwa.h"
#include "bwamem.h"
#include "bwt.h"
#include "bntseq.h"
#include "ksw.h"
#include "kvec.h"

/* private to bwamem.c -- layouts re-declared (sizes checked in refshim_sizes) */
typedef struct { int64_t rbeg; int32_t qbeg, len; int score; } rs_seed_t;
typedef struct {
	int n, m, first, rid;
	uint32_t w:29, kept:2, is_alt:1;
	float frac_rep;
	int64_t pos;
	rs_seed_t *seeds;
} rs_chain_t;
typedef struct { size_t n, m; rs_chain_t *a; } rs_chain_v;
typedef struct { bwtintv_v mem, mem1, *tmpv[2]; } rs_aux_t;

/* exported by bwamem.c although not declared in bwamem.h */
extern rs_chain_v mem_chain(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, int len, const uint8_t *seq, void *buf);
extern int mem_chain_flt(const mem_opt_t *opt, int n_chn, rs_chain_t *a);
extern void mem_flt_chained_seeds(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, int n_chn, rs_chain_t *a);
extern void mem_chain2aln(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, int l_query, const uint8_t *query, const rs_chain_t *c, mem_alnreg_v *av);
extern int mem_sort_dedup_patch(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, uint8_t *query, int n, mem_alnreg_t *a);
extern mem_alnreg_v mem_align1_core(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int l_seq, char *seq, void *buf);
extern int mem_mark_primary_se(const mem_opt_t *opt, int n, mem_alnreg_t *a, int64_t id);
extern void mem_reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, mem_alnreg_v *a, int extra_flag, const mem_aln_t *m);
extern int mem_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], mem_alnreg_v a[2]);

/* flat chain header handed to python */
typedef struct {
	int32_t n, rid, w, kept, is_alt, first;
	float frac_rep;
	int32_t seed_off; /* offset of this chain's first seed in the flat seed array */
	int64_t pos;
} rs_chain_hdr_t;

void refshim_sizes(int32_t out[16])
{
	out[0] = sizeof(mem_opt_t); out[1] = sizeof(mem_alnreg_t); out[2] = sizeof(bwtintv_t);
	out[3] = sizeof(rs_seed_t); out[4] = sizeof(rs_chain_t); out[5] = sizeof(bseq1_t);
	out[6] = sizeof(mem_pestat_t); out[7] = sizeof(mem_aln_t); out[8] = sizeof(bwt_t);
	out[9] = sizeof(bntseq_t); out[10] = sizeof(bntann1_t); out[11] = sizeof(rs_chain_hdr_t);
}

void *refshim_idx_load(const char *prefix) { bwa_verbose = 1; return bwa_idx_load(prefix, BWA_IDX_ALL); }
void refshim_idx_destroy(void *idx) { bwa_idx_destroy((bwaidx_t*)idx); }
void refshim_idx_info(void *idx_, int64_t out[8])
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	out[0] = idx->bns->l_pac; out[1] = idx->bns->n_seqs; out[2] = idx->bwt->seq_len; out[3] = idx->bwt->primary;
	out[4] = idx->bwt->sa_intv; out[5] = idx->bwt->n_sa; out[6] = idx->bwt->bwt_size; out[7] = idx->bns->n_holes;
}
const void *refshim_idx_bwt(void *idx) { return ((bwaidx_t*)idx)->bwt; }
const void *refshim_idx_bns(void *idx) { return ((bwaidx_t*)idx)->bns; }
const void *refshim_idx_pac(void *idx) { return ((bwaidx_t*)idx)->pac; }
void refshim_set_alt(void *idx_, int rid, int is_alt) { ((bwaidx_t*)idx_)->bns->anns[rid].is_alt = is_alt; }

void *refshim_opt_init(void) { return mem_opt_init(); }
void refshim_free(void *p) { free(p); }

/* Full mem_align1_core for n reads given as nt4 codes (0..4), concatenated with offsets off[0..n]. */
int64_t refshim_align(void *idx_, const mem_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off,
					  int32_t *counts, mem_alnreg_t *out, int64_t cap)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	int64_t tot = 0; int i;
	for (i = 0; i < n; ++i) {
		int len = (int)(off[i+1] - off[i]);
		char *s = (char*)malloc(len + 1);
		mem_alnreg_v r;
		memcpy(s, seqs + off[i], len);
		r = mem_align1_core(opt, idx->bwt, idx->bns, idx->pac, len, s, 0);
		counts[i] = (int32_t)r.n;
		if (tot + (int64_t)r.n <= cap) memcpy(out + tot, r.a, r.n * sizeof(mem_alnreg_t));
		tot += r.n;
		free(r.a); free(s);
	}
	return tot;
}

/* Sorted SA intervals of one read (the content of smem_aux_t::mem after mem_collect_intv, bwamem.c:187). */
int refshim_intervals(void *idx_, const mem_opt_t *opt, int len, const uint8_t *seq, bwtintv_t *out, int cap)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	rs_aux_t *aux = (rs_aux_t*)calloc(1, sizeof(rs_aux_t));
	rs_chain_v chn; size_t i; int n;
	aux->tmpv[0] = (bwtintv_v*)calloc(1, sizeof(bwtintv_v));
	aux->tmpv[1] = (bwtintv_v*)calloc(1, sizeof(bwtintv_v));
	chn = mem_chain(opt, idx->bwt, idx->bns, len, seq, aux);
	n = (int)aux->mem.n;
	for (i = 0; i < aux->mem.n && (int)i < cap; ++i) out[i] = aux->mem.a[i];
	if (len < opt->min_seed_len) n = 0;
	for (i = 0; i < chn.n; ++i) free(chn.a[i].seeds);
	free(chn.a);
	free(aux->tmpv[0]->a); free(aux->tmpv[0]); free(aux->tmpv[1]->a); free(aux->tmpv[1]);
	free(aux->mem.a); free(aux->mem1.a); free(aux);
	return n;
}

/* Chains of one read at stage 0 (after mem_chain), 1 (after mem_chain_flt), 2 (after mem_flt_chained_seeds). */
int refshim_chains(void *idx_, const mem_opt_t *opt, int len, const uint8_t *seq, int stage,
				   rs_chain_hdr_t *hdr, int cap_chain, rs_seed_t *seeds, int cap_seed, int32_t *n_seed_out)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	rs_chain_v chn = mem_chain(opt, idx->bwt, idx->bns, len, seq, 0);
	size_t i; int j, ns = 0;
	if (stage >= 1) chn.n = mem_chain_flt(opt, chn.n, chn.a);
	if (stage >= 2) mem_flt_chained_seeds(opt, idx->bns, idx->pac, len, seq, chn.n, chn.a);
	for (i = 0; i < chn.n; ++i) {
		rs_chain_t *c = &chn.a[i];
		if ((int)i < cap_chain) {
			hdr[i].n = c->n; hdr[i].rid = c->rid; hdr[i].w = c->w; hdr[i].kept = c->kept; hdr[i].is_alt = c->is_alt;
			hdr[i].first = c->first; hdr[i].frac_rep = c->frac_rep; hdr[i].seed_off = ns; hdr[i].pos = c->pos;
		}
		for (j = 0; j < c->n; ++j, ++ns)
			if (ns < cap_seed) seeds[ns] = c->seeds[j];
		free(c->seeds);
	}
	*n_seed_out = ns;
	free(chn.a);
	return (int)chn.n;
}

/* Alignment regions of one read before (stage 0) or after (stage 1) mem_sort_dedup_patch. */
int refshim_regs_stage(void *idx_, const mem_opt_t *opt, int len, const uint8_t *seq_, int stage, mem_alnreg_t *out, int cap)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	uint8_t *seq = (uint8_t*)malloc(len + 1);
	rs_chain_v chn; mem_alnreg_v regs; size_t i; int n;
	memcpy(seq, seq_, len);
	chn = mem_chain(opt, idx->bwt, idx->bns, len, seq, 0);
	chn.n = mem_chain_flt(opt, chn.n, chn.a);
	mem_flt_chained_seeds(opt, idx->bns, idx->pac, len, seq, chn.n, chn.a);
	kv_init(regs);
	for (i = 0; i < chn.n; ++i) {
		mem_chain2aln(opt, idx->bns, idx->pac, len, seq, &chn.a[i], &regs);
		free(chn.a[i].seeds);
	}
	free(chn.a);
	if (stage >= 1) regs.n = mem_sort_dedup_patch(opt, idx->bns, idx->pac, seq, regs.n, regs.a);
	n = (int)regs.n;
	for (i = 0; i < regs.n && (int)i < cap; ++i) out[i] = regs.a[i];
	free(regs.a); free(seq);
	return n;
}

/* Reference mem_process_seqs over an in-memory batch -> concatenated SAM text (caller frees with refshim_free).
 * seqs are ASCII here (as bseq_read would deliver them); names are NUL-separated. */
char *refshim_process_seqs(void *idx_, const mem_opt_t *opt, int64_t n_processed, int n, const char *names,
						   const char *seqs, const char *quals, const int64_t *off, const mem_pestat_t *pes0, int64_t *out_len)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	bseq1_t *bs = (bseq1_t*)calloc(n, sizeof(bseq1_t));
	const char *nm = names; int i; size_t tot = 0, pos = 0; char *out;
	for (i = 0; i < n; ++i) {
		int len = (int)(off[i+1] - off[i]);
		bs[i].l_seq = len; bs[i].id = i;
		bs[i].name = strdup(nm); nm += strlen(nm) + 1;
		bs[i].seq = (char*)malloc(len + 1); memcpy(bs[i].seq, seqs + off[i], len); bs[i].seq[len] = 0;
		if (quals) { bs[i].qual = (char*)malloc(len + 1); memcpy(bs[i].qual, quals + off[i], len); bs[i].qual[len] = 0; }
	}
	mem_process_seqs(opt, idx->bwt, idx->bns, idx->pac, n_processed, n, bs, pes0);
	for (i = 0; i < n; ++i) tot += strlen(bs[i].sam);
	out = (char*)malloc(tot + 1);
	for (i = 0; i < n; ++i) {
		size_t l = strlen(bs[i].sam);
		memcpy(out + pos, bs[i].sam, l); pos += l;
		free(bs[i].name); free(bs[i].seq); free(bs[i].qual); free(bs[i].sam);
	}
	out[tot] = 0; *out_len = (int64_t)tot;
	free(bs);
	return out;
}

/* Finalize-only entry: given regs (as produced by any mem_align1_core implementation) run the reference's
 * worker2 logic (bwamem.c:1217-1233) and return SAM text.  Used to prove that regs from the HIP path drive the
 * reference's own SAM writer to byte-identical output. */
char *refshim_regs2sam(void *idx_, const mem_opt_t *opt, int64_t n_processed, int n, const char *names,
					   const char *seqs_nt4, const char *quals, const int64_t *off, const int32_t *counts,
					   const mem_alnreg_t *regs, const mem_pestat_t *pes0, int64_t *out_len)
{
	bwaidx_t *idx = (bwaidx_t*)idx_;
	bseq1_t *bs = (bseq1_t*)calloc(n, sizeof(bseq1_t));
	mem_alnreg_v *rv = (mem_alnreg_v*)calloc(n, sizeof(mem_alnreg_v));
	mem_pestat_t pes[4];
	const char *nm = names; int i; size_t tot = 0, pos = 0; char *out; int64_t roff = 0;
	for (i = 0; i < n; ++i) {
		int len = (int)(off[i+1] - off[i]);
		bs[i].l_seq = len; bs[i].id = i;
		bs[i].name = strdup(nm); nm += strlen(nm) + 1;
		bs[i].seq = (char*)malloc(len + 1); memcpy(bs[i].seq, seqs_nt4 + off[i], len); bs[i].seq[len] = 0;
		if (quals) { bs[i].qual = (char*)malloc(len + 1); memcpy(bs[i].qual, quals + off[i], len); bs[i].qual[len] = 0; }
		rv[i].n = rv[i].m = counts[i];
		rv[i].a = (mem_alnreg_t*)malloc((counts[i] + 1) * sizeof(mem_alnreg_t));
		memcpy(rv[i].a, regs + roff, counts[i] * sizeof(mem_alnreg_t)); roff += counts[i];
	}
	if (opt->flag & MEM_F_PE) {
		if (pes0) memcpy(pes, pes0, 4 * sizeof(mem_pestat_t));
		else mem_pestat(opt, idx->bns->l_pac, n, rv, pes);
		for (i = 0; i < n>>1; ++i)
			mem_sam_pe(opt, idx->bns, idx->pac, pes, (n_processed>>1) + i, &bs[i<<1], &rv[i<<1]);
	} else {
		for (i = 0; i < n; ++i) {
			mem_mark_primary_se(opt, rv[i].n, rv[i].a, n_processed + i);
			mem_reg2sam(opt, idx->bns, idx->pac, &bs[i], &rv[i], 0, 0);
		}
	}
	for (i = 0; i < n; ++i) tot += strlen(bs[i].sam);
	out = (char*)malloc(tot + 1);
	for (i = 0; i < n; ++i) {
		size_t l = strlen(bs[i].sam);
		memcpy(out + pos, bs[i].sam, l); pos += l;
		free(bs[i].name); free(bs[i].seq); free(bs[i].qual); free(bs[i].sam); free(rv[i].a);
	}
	out[tot] = 0; *out_len = (int64_t)tot;
	free(bs); free(rv);
	return out;
}
