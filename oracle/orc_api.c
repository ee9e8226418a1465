/* oracle/orc_api.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * Flat, ctypes-friendly entry points mirroring oracle/ref_shim.c one-to-one, so the same python test code can
 * drive either the compiled reference or this restatement. */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

int orc_intervals(const orc_index_t *ix, const orc_opt_t *opt, int len, const uint8_t *seq, orc_intv_t *out, int cap)
{
	orc_intv_v v = {0,0,0}; size_t i; int n;
	if (len < opt->min_seed_len) return 0;
	orc_collect_intv(opt, ix, len, seq, &v);
	for (i = 0; i < v.n && (int)i < cap; ++i) out[i] = v.a[i];
	n = (int)v.n; free(v.a);
	return n;
}

int orc_chains(const orc_index_t *ix, const orc_opt_t *opt, int len, const uint8_t *seq, int stage,
			   orc_chain_hdr_t *hdr, int cap_chain, orc_seed_t *seeds, int cap_seed, int32_t *n_seed_out)
{
	orc_chain_v chn = orc_chain(opt, ix, len, seq); size_t i; int j, ns = 0;
	if (stage >= 1) chn.n = orc_chain_flt(opt, (int)chn.n, chn.a);
	if (stage >= 2) orc_flt_chained_seeds(opt, ix, len, seq, (int)chn.n, chn.a);
	for (i = 0; i < chn.n; ++i) {
		orc_chain_t *c = &chn.a[i];
		if ((int)i < cap_chain) {
			hdr[i].n = c->n; hdr[i].rid = c->rid; hdr[i].w = c->w; hdr[i].kept = c->kept; hdr[i].is_alt = c->is_alt;
			hdr[i].first = c->first; hdr[i].frac_rep = c->frac_rep; hdr[i].seed_off = ns; hdr[i].pos = c->pos;
		}
		for (j = 0; j < c->n; ++j, ++ns) if (ns < cap_seed) seeds[ns] = c->seeds[j];
		free(c->seeds);
	}
	*n_seed_out = ns;
	free(chn.a);
	return (int)chn.n;
}

int orc_regs_stage(const orc_index_t *ix, const orc_opt_t *opt, int len, const uint8_t *seq_, int stage, orc_reg_t *out, int cap)
{
	uint8_t *seq = (uint8_t*)malloc(len + 1);
	orc_chain_v chn; orc_reg_v regs = {0,0,0}; size_t i; int n;
	memcpy(seq, seq_, len);
	chn = orc_chain(opt, ix, len, seq);
	chn.n = orc_chain_flt(opt, (int)chn.n, chn.a);
	orc_flt_chained_seeds(opt, ix, len, seq, (int)chn.n, chn.a);
	for (i = 0; i < chn.n; ++i) { orc_chain2aln(opt, ix, len, seq, &chn.a[i], &regs); free(chn.a[i].seeds); }
	free(chn.a);
	if (stage >= 1) regs.n = orc_sort_dedup_patch(opt, ix, seq, (int)regs.n, regs.a);
	n = (int)regs.n;
	for (i = 0; i < regs.n && (int)i < cap; ++i) out[i] = regs.a[i];
	free(regs.a); free(seq);
	return n;
}

int64_t orc_align(const orc_index_t *ix, const orc_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off,
				  int32_t *counts, orc_reg_t *out, int64_t cap)
{
	int64_t tot = 0; int i;
	for (i = 0; i < n; ++i) {
		int len = (int)(off[i+1] - off[i]);
		uint8_t *s = (uint8_t*)malloc(len + 1); orc_reg_v r;
		memcpy(s, seqs + off[i], len);
		r = orc_align1_core(opt, ix, len, s);
		counts[i] = (int32_t)r.n;
		if (tot + (int64_t)r.n <= cap) memcpy(out + tot, r.a, r.n * sizeof(orc_reg_t));
		tot += r.n;
		free(r.a); free(s);
	}
	return tot;
}

void orc_sizes(int32_t out[8])
{
	out[0] = sizeof(orc_opt_t); out[1] = sizeof(orc_reg_t); out[2] = sizeof(orc_intv_t); out[3] = sizeof(orc_seed_t);
	out[4] = sizeof(orc_chain_hdr_t);
}
