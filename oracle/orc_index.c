/* oracle/orc_index.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 * Loader for the reference's on-disk index and the contig / packed-reference queries of the hot path. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "orc.h"

static void *slurp(const char *fn, size_t skip, size_t *nbytes)
{
	FILE *fp = fopen(fn, "rb");
	long sz; void *buf;
	if (!fp) return 0;
	fseek(fp, 0, SEEK_END); sz = ftell(fp); fseek(fp, (long)skip, SEEK_SET);
	buf = malloc(sz - skip + 16);
	if (fread(buf, 1, sz - skip, fp) != (size_t)(sz - skip)) { free(buf); fclose(fp); return 0; }
	fclose(fp);
	*nbytes = sz - skip;
	return buf;
}

/* .bwt layout: bwt_dump_bwt / bwt_restore_bwt (bwt.c:385-393, 443-462)
 * .sa  layout: bwt_dump_sa / bwt_restore_sa (bwt.c:396-441); sa[0] is forced to -1
 * .ann/.alt  : bns_restore_core / bns_restore (bntseq.c:97-211); .pac: bwa_idx_load_from_disk (bwa.c:300-312) */
orc_index_t *orc_index_load(const char *prefix)
{
	char fn[4096]; size_t nb; uint64_t hdr[7]; FILE *fp; int i;
	orc_index_t *ix = (orc_index_t*)calloc(1, sizeof(orc_index_t));
	long long xx; unsigned seed;
	/* .bwt */
	snprintf(fn, sizeof fn, "%s.bwt", prefix);
	fp = fopen(fn, "rb"); if (!fp) goto fail;
	if (fread(hdr, 8, 5, fp) != 5) { fclose(fp); goto fail; }
	fclose(fp);
	ix->primary = hdr[0]; ix->L2[0] = 0;
	for (i = 0; i < 4; ++i) ix->L2[i+1] = hdr[1+i];
	ix->seq_len = ix->L2[4];
	ix->bwt = (uint32_t*)slurp(fn, 40, &nb); if (!ix->bwt) goto fail;
	ix->bwt_size = nb >> 2;
	/* .sa */
	snprintf(fn, sizeof fn, "%s.sa", prefix);
	fp = fopen(fn, "rb"); if (!fp) goto fail;
	if (fread(hdr, 8, 7, fp) != 7) { fclose(fp); goto fail; }
	if (hdr[0] != ix->primary || hdr[6] != ix->seq_len) { fclose(fp); goto fail; }
	ix->sa_intv = (int)hdr[5];
	ix->n_sa = (ix->seq_len + ix->sa_intv) / ix->sa_intv;
	ix->sa = (uint64_t*)calloc(ix->n_sa, 8);
	ix->sa[0] = (uint64_t)-1;
	if (fread(ix->sa + 1, 8, ix->n_sa - 1, fp) != ix->n_sa - 1) { fclose(fp); goto fail; }
	fclose(fp);
	/* .ann */
	snprintf(fn, sizeof fn, "%s.ann", prefix);
	fp = fopen(fn, "r"); if (!fp) goto fail;
	if (fscanf(fp, "%lld%d%u", &xx, &ix->n_seqs, &seed) != 3) { fclose(fp); goto fail; }
	ix->l_pac = xx;
	ix->ctg = (orc_contig_t*)calloc(ix->n_seqs, sizeof(orc_contig_t));
	for (i = 0; i < ix->n_seqs; ++i) {
		char name[8192]; unsigned gi; int c, n_ambs;
		if (fscanf(fp, "%u%8191s", &gi, name) != 2) { fclose(fp); goto fail; }
		ix->ctg[i].name = strdup(name);
		while ((c = fgetc(fp)) != '\n' && c != EOF) {}
		if (fscanf(fp, "%lld%d%d", &xx, &ix->ctg[i].len, &n_ambs) != 3) { fclose(fp); goto fail; }
		ix->ctg[i].offset = xx;
	}
	fclose(fp);
	/* .alt (optional): first column of every non-@ line names an ALT contig */
	snprintf(fn, sizeof fn, "%s.alt", prefix);
	if ((fp = fopen(fn, "r")) != 0) {
		char line[8192];
		while (fgets(line, sizeof line, fp)) {
			char *e = line;
			if (line[0] == '@') continue;
			while (*e && *e != '\t' && *e != '\n' && *e != '\r') ++e;
			*e = 0;
			for (i = 0; i < ix->n_seqs; ++i)
				if (strcmp(ix->ctg[i].name, line) == 0) ix->ctg[i].is_alt = 1;
		}
		fclose(fp);
	}
	/* .pac */
	snprintf(fn, sizeof fn, "%s.pac", prefix);
	ix->pac = (uint8_t*)slurp(fn, 0, &nb); if (!ix->pac) goto fail;
	if ((int64_t)nb < ix->l_pac / 4 + 1) goto fail;
	return ix;
fail:
	orc_index_free(ix);
	return 0;
}

void orc_index_free(orc_index_t *ix)
{
	int i;
	if (!ix) return;
	if (ix->ctg) for (i = 0; i < ix->n_seqs; ++i) free(ix->ctg[i].name);
	free(ix->ctg); free(ix->bwt); free(ix->sa); free(ix->pac); free(ix);
}

void orc_set_alt(orc_index_t *ix, int rid, int is_alt) { ix->ctg[rid].is_alt = is_alt; }

/* bns_pos2rid (bntseq.c:354-368): index of the contig whose [offset, offset+len) holds pos_f */
int orc_pos2rid(const orc_index_t *ix, int64_t pos_f)
{
	int lo = 0, hi = ix->n_seqs; /* answer in [lo, hi) */
	if (pos_f >= ix->l_pac) return -1;
	while (hi - lo > 1) {
		int mid = (lo + hi) >> 1;
		if (ix->ctg[mid].offset <= pos_f) lo = mid; else hi = mid;
	}
	return lo;
}

static inline int64_t depos(const orc_index_t *ix, int64_t pos, int *is_rev)
{	/* bns_depos (bntseq.h:87-90) */
	*is_rev = pos >= ix->l_pac;
	return *is_rev ? (ix->l_pac << 1) - 1 - pos : pos;
}

/* bns_intv2rid (bntseq.c:370-379) */
int orc_intv2rid(const orc_index_t *ix, int64_t rb, int64_t re)
{
	int r, a, b;
	if (rb < ix->l_pac && re > ix->l_pac) return -2;
	a = orc_pos2rid(ix, depos(ix, rb, &r));
	b = rb < re ? orc_pos2rid(ix, depos(ix, re - 1, &r)) : a;
	return a == b ? a : -1;
}

static inline int pac_base(const uint8_t *pac, int64_t l)
{	/* _get_pac (bntseq.c:230) */
	return pac[l >> 2] >> ((~l & 3) << 1) & 3;
}

/* bns_get_seq (bntseq.c:403-424): bases of [beg,end) in the forward+revcomp coordinate system */
uint8_t *orc_get_seq(const orc_index_t *ix, int64_t beg, int64_t end, int64_t *len)
{
	int64_t l_pac = ix->l_pac, k, n = 0; uint8_t *s = 0;
	if (end < beg) { int64_t t = beg; beg = end; end = t; }
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	*len = 0;
	if (beg >= l_pac || end <= l_pac) {
		*len = end - beg;
		s = (uint8_t*)malloc(end - beg + 1);
		if (beg >= l_pac) {
			for (k = (l_pac << 1) - 1 - beg; k > (l_pac << 1) - 1 - end; --k) s[n++] = 3 - pac_base(ix->pac, k);
		} else for (k = beg; k < end; ++k) s[n++] = pac_base(ix->pac, k);
	}
	return s;
}

/* bns_fetch_seq (bntseq.c:426-451): clamp [beg,end) to the contig that holds mid, then fetch */
uint8_t *orc_fetch_seq(const orc_index_t *ix, int64_t *beg, int64_t mid, int64_t *end, int *rid)
{
	int64_t fb, fe, len; int is_rev;
	if (*end < *beg) { int64_t t = *beg; *beg = *end; *end = t; }
	*rid = orc_pos2rid(ix, depos(ix, mid, &is_rev));
	fb = ix->ctg[*rid].offset; fe = fb + ix->ctg[*rid].len;
	if (is_rev) { int64_t t = fb; fb = (ix->l_pac << 1) - fe; fe = (ix->l_pac << 1) - t; }
	if (*beg < fb) *beg = fb;
	if (*end > fe) *end = fe;
	return orc_get_seq(ix, *beg, *end, &len);
}
