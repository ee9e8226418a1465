// host_index.cpp -- the reference-sequence side of the index for the host finalize code: .ann/.alt/.pac
// (formats: bntseq.c:97-211; loader bwa.c:300-312).
#include <stdio.h>
#include <string.h>
#include "bwamem_host.h"

namespace hostmem {

bool load_refseqs(const std::string &prefix, RefSeqs &out, std::string &err)
{
	FILE *fa = fopen((prefix + ".ann").c_str(), "r");
	if (!fa) { err = "cannot open " + prefix + ".ann"; return false; }
	long long xx; int n_seqs; unsigned seed;
	if (fscanf(fa, "%lld%d%u", &xx, &n_seqs, &seed) != 3 || n_seqs <= 0) { fclose(fa); err = "bad .ann header"; return false; }
	out.l_pac = xx;
	out.ctg.resize(n_seqs);
	for (int i = 0; i < n_seqs; ++i) {
		Contig &c = out.ctg[i];
		char name[8192], anno[8192]; int k = 0, ch;
		if (fscanf(fa, "%u%8191s", &c.gi, name) != 2) { fclose(fa); err = "bad .ann record"; return false; }
		c.name = name;
		while ((ch = fgetc(fa)) != '\n' && ch != EOF) if (k < 8190) anno[k++] = (char)ch;   // " anno" or " (null)"
		anno[k] = 0;
		c.anno = (k > 1 && strcmp(anno, " (null)") != 0) ? std::string(anno + 1) : std::string();   // bntseq.c:128-129
		if (fscanf(fa, "%lld%d%d", &xx, &c.len, &c.n_ambs) != 3) { fclose(fa); err = "bad .ann record"; return false; }
		c.offset = xx; c.is_alt = 0;
	}
	fclose(fa);
	if (FILE *fl = fopen((prefix + ".alt").c_str(), "r")) {   // bns_restore (bntseq.c:185-205): first column of every non-@ line
		std::string name; int ch;
		while ((ch = fgetc(fl)) != EOF) {            // lines of any length; like the reference, a last line without a line end is not seen
			if (ch == '\t' || ch == '\n' || ch == '\r') {
				if (!name.empty() && name[0] != '@') for (auto &c : out.ctg) if (c.name == name) c.is_alt = 1;
				while (ch != '\n' && ch != EOF) ch = fgetc(fl);
				name.clear();
			} else name += (char)ch;
		}
		fclose(fl);
	}
	FILE *fp = fopen((prefix + ".pac").c_str(), "rb");
	if (!fp) { err = "cannot open " + prefix + ".pac"; return false; }
	out.pac.resize((size_t)(out.l_pac / 4 + 1));
	size_t got = fread(out.pac.data(), 1, out.pac.size(), fp);
	fclose(fp);
	if (got != out.pac.size()) { err = ".pac too short"; return false; }
	return true;
}

int RefSeqs::pos2rid(int64_t pos_f) const
{
	if (pos_f >= l_pac) return -1;
	int lo = 0, hi = (int)ctg.size();
	while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (ctg[mid].offset <= pos_f) lo = mid; else hi = mid; }
	return lo;
}

static inline int pac_at(const uint8_t *pac, int64_t l) { return pac[l >> 2] >> ((~l & 3) << 1) & 3; }

// bns_get_seq (bntseq.c:403-423): bases [beg, end) of the forward + reverse-complement coordinate space; empty when the range
// bridges the two strands.  Four bases per packed byte are unpacked at a time through a 256-entry table (the per-base form --
// one shift/mask and a push_back per base -- was 15 % of the finalize stage).
struct PacLut { uint8_t f[256][4], r[256][4]; PacLut() { for (int b = 0; b < 256; ++b) for (int k = 0; k < 4; ++k) { f[b][k] = (uint8_t)(b >> ((3 - k) << 1) & 3); r[b][k] = (uint8_t)(3 - (b >> (k << 1) & 3)); } } };
static const PacLut g_paclut;

void RefSeqs::get_seq(int64_t beg, int64_t end, std::vector<uint8_t> &out) const
{
	out.clear();
	if (end < beg) std::swap(beg, end);
	if (end > l_pac << 1) end = l_pac << 1;
	if (beg < 0) beg = 0;
	if (!(beg >= l_pac || end <= l_pac)) return;
	const size_t n = (size_t)(end - beg);
	out.resize(n);
	uint8_t *o = out.data();
	const uint8_t *p = pac.data();
	if (beg < l_pac) {          // forward strand: bases beg .. end-1
		int64_t k = beg; size_t i = 0;
		for (; i < n && (k & 3); ++i, ++k) o[i] = (uint8_t)pac_at(p, k);
		for (; i + 4 <= n; i += 4, k += 4) memcpy(o + i, g_paclut.f[p[k >> 2]], 4);
		for (; i < n; ++i, ++k) o[i] = (uint8_t)pac_at(p, k);
	} else {                    // reverse strand: complement of forward bases 2 l_pac - 1 - beg downwards
		int64_t k = (l_pac << 1) - 1 - beg; size_t i = 0;
		for (; i < n && (k & 3) != 3; ++i, --k) o[i] = (uint8_t)(3 - pac_at(p, k));
		for (; i + 4 <= n; i += 4, k -= 4) memcpy(o + i, g_paclut.r[p[k >> 2]], 4);    // byte k>>2 holds bases k-3 .. k; reversed and complemented
		for (; i < n; ++i, --k) o[i] = (uint8_t)(3 - pac_at(p, k));
	}
}

bool RefSeqs::fetch_seq(int64_t &beg, int64_t mid, int64_t &end, int &rid, std::vector<uint8_t> &out) const
{
	if (end < beg) std::swap(beg, end);
	bool is_rev = mid >= l_pac;
	rid = pos2rid(is_rev ? (l_pac << 1) - 1 - mid : mid);
	int64_t fb = ctg[rid].offset, fe = fb + ctg[rid].len;
	if (is_rev) { int64_t t = fb; fb = (l_pac << 1) - fe; fe = (l_pac << 1) - t; }
	if (beg < fb) beg = fb;
	if (end > fe) end = fe;
	get_seq(beg, end, out);
	return (int64_t)out.size() == end - beg;
}

}  // namespace hostmem
