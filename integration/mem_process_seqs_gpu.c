/* integration/mem_process_seqs_gpu.c -- the reference-side binding a bwa maintainer would add to run the
 * mem_align1_core loop on an MI355X through libbwagpu.so.  (Shown in INTEGRATION.md; built for real by
 * `make -C oracle bwa_gpu` into oracle/_ref/bwa_gpu = the unmodified reference program linked with
 * -Wl,--wrap=mem_process_seqs so that fastmap.c's calls land here.)
 *
 * It is written against the reference's own headers and keeps the reference's structure
 * (bwamem.c:1235-1264): only the first kt_for (worker1, bwamem.c:1252) is replaced by one call into the C-ABI;
 * mem_pestat and the worker2 loop stay the reference's own code.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "bwamem.h"
#include "bntseq.h"
#include "bwt.h"
#include "bwa.h"
#include "utils.h"
#include "bwagpu.h"

extern void kt_for(int n_threads, void (*func)(void*, long, int), void *data, long n);
extern int mem_mark_primary_se(const mem_opt_t *opt, int n, mem_alnreg_t *a, int64_t id);
extern void mem_reorder_primary5(int T, mem_alnreg_v *a);
extern void mem_reg2sam(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, bseq1_t *s, mem_alnreg_v *a, int extra_flag, const mem_aln_t *m);
extern int mem_sam_pe(const mem_opt_t *opt, const bntseq_t *bns, const uint8_t *pac, const mem_pestat_t pes[4], uint64_t id, bseq1_t s[2], mem_alnreg_v a[2]);

typedef struct {
	const mem_opt_t *opt; const bntseq_t *bns; const uint8_t *pac; const mem_pestat_t *pes;
	bseq1_t *seqs; mem_alnreg_v *regs; int64_t n_processed;
} fin_t;

static void finalize1(void *data, long i, int tid)   /* == worker2, bwamem.c:1217-1233 */
{
	fin_t *w = (fin_t*)data;
	if (!(w->opt->flag & MEM_F_PE)) {
		mem_mark_primary_se(w->opt, w->regs[i].n, w->regs[i].a, w->n_processed + i);
		if (w->opt->flag & MEM_F_PRIMARY5) mem_reorder_primary5(w->opt->T, &w->regs[i]);
		mem_reg2sam(w->opt, w->bns, w->pac, &w->seqs[i], &w->regs[i], 0, 0);
		free(w->regs[i].a);
	} else {
		mem_sam_pe(w->opt, w->bns, w->pac, w->pes, (w->n_processed >> 1) + i, &w->seqs[i<<1], &w->regs[i<<1]);
		free(w->regs[i<<1|0].a); free(w->regs[i<<1|1].a);
	}
}

static bwagpu_t *gpu_handle(const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac)
{
	static bwagpu_t *h = 0;
	if (h == 0) {   /* one-time upload of the index the reference has already loaded (bwa.c:289-321) */
		bwagpu_index_desc_t d; int i, rc;
		int64_t *off = (int64_t*)malloc(bns->n_seqs * 8); int32_t *len = (int32_t*)malloc(bns->n_seqs * 4), *alt = (int32_t*)malloc(bns->n_seqs * 4);
		memset(&d, 0, sizeof d);
		d.bwt = bwt->bwt; d.bwt_size = bwt->bwt_size; d.primary = bwt->primary; memcpy(d.L2, bwt->L2, sizeof d.L2); d.seq_len = bwt->seq_len;
		d.sa = bwt->sa; d.n_sa = bwt->n_sa; d.sa_intv = bwt->sa_intv;
		d.pac = pac; d.l_pac = bns->l_pac; d.n_seqs = bns->n_seqs;
		for (i = 0; i < bns->n_seqs; ++i) off[i] = bns->anns[i].offset, len[i] = bns->anns[i].len, alt[i] = bns->anns[i].is_alt;
		d.ctg_offset = off; d.ctg_len = len; d.ctg_is_alt = alt;
		rc = bwagpu_create(&h, &d, getenv("BWAGPU_DEVICE") ? atoi(getenv("BWAGPU_DEVICE")) : 0);
		free(off); free(len); free(alt);
		if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] %s\n", __func__, bwagpu_strerror(rc)); exit(EXIT_FAILURE); }   /* err_fatal-style, utils.c:90 */
	}
	return h;
}

void __wrap_mem_process_seqs(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int64_t n_processed, int n, bseq1_t *seqs, const mem_pestat_t *pes0)
{
	fin_t w; mem_pestat_t pes[4]; double ctime = cputime(), rtime = realtime(); int rc;
	bwagpu_t *h = gpu_handle(bwt, bns, pac);
	w.regs = (mem_alnreg_v*)malloc(n * sizeof(mem_alnreg_v));
	w.opt = opt; w.bns = bns; w.pac = pac; w.seqs = seqs; w.n_processed = n_processed; w.pes = &pes[0];
	/* was: kt_for(opt->n_threads, worker1, &w, ...)  -- every read through mem_align1_core (bwamem.c:1252) */
	rc = bwagpu_align_bseq(h, (const bwagpu_opt_t*)opt, n, (bwagpu_bseq1_t*)seqs, (bwagpu_alnreg_v*)w.regs);
	if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] %s: %s\n", __func__, bwagpu_strerror(rc), bwagpu_last_error(h)); exit(EXIT_FAILURE); }
	if (opt->flag & MEM_F_PE) {
		if (pes0) memcpy(pes, pes0, 4 * sizeof(mem_pestat_t));
		else mem_pestat(opt, bns->l_pac, n, w.regs, pes);
	}
	kt_for(opt->n_threads, finalize1, &w, (opt->flag & MEM_F_PE) ? n >> 1 : n);
	free(w.regs);
	if (bwa_verbose >= 3)
		fprintf(stderr, "[M::%s] Processed %d reads in %.3f CPU sec, %.3f real sec\n", "mem_process_seqs", n, cputime() - ctime, realtime() - rtime);
}

/* mem_align1 (bwamem_extra.c:102-112; the library entry point bwamem.h documents, caller example.c:40): one read through the same
 * device path.  Like the reference's it works on a copy of the sequence, marks primaries with a random id and returns a malloc'd array
 * (-Wl,--wrap=mem_align1: oracle/_ref/example_gpu is the reference's example.c linked this way). */
mem_alnreg_v __wrap_mem_align1(const mem_opt_t *opt, const bwt_t *bwt, const bntseq_t *bns, const uint8_t *pac, int l_seq, const char *seq_)
{
	bseq1_t s; mem_alnreg_v ar; int rc;
	bwagpu_t *h = gpu_handle(bwt, bns, pac);
	memset(&s, 0, sizeof s);
	s.l_seq = l_seq; s.seq = (char*)malloc(l_seq > 0 ? l_seq : 1);
	memcpy(s.seq, seq_, l_seq);
	ar.n = ar.m = 0; ar.a = 0;
	/* was: mem_align1_core(opt, bwt, bns, pac, l_seq, seq, 0)  (bwamem_extra.c:108) */
	rc = bwagpu_align_bseq(h, (const bwagpu_opt_t*)opt, 1, (bwagpu_bseq1_t*)&s, (bwagpu_alnreg_v*)&ar);
	if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] %s: %s\n", __func__, bwagpu_strerror(rc), bwagpu_last_error(h)); exit(EXIT_FAILURE); }
	mem_mark_primary_se(opt, ar.n, ar.a, lrand48());
	free(s.seq);
	return ar;
}

