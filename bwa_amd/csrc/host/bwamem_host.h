// bwamem_host.h -- host side of the stand-alone aligner: the reference's mem_process_seqs() surface
// (bwamem.h:161, bwamem.c:1235-1264) re-written from scratch around libbwagpu.so.
//
//   worker1 loop  -> bwagpu_align_bseq()            (device, include/bwagpu.h)
//   mem_pestat    -> hostmem::pestat_flat()              (bwamem_pair.c:72-135)
//   worker2 loop  -> hostmem::finalize_se / finalize_pe
//                    mark-primary (bwamem.c:519-584), mapQ (:982-1006), CIGAR/NM/MD (bwa.c:148-234 + ksw.c:540-642),
//                    XA (bwamem_extra.c:124-172), SAM record (bwamem.c:851-976), mate rescue / pairing
//                    (bwamem_pair.c:137-419)
// Output must be byte-identical to the reference's SAM; tests/test_host_finalize.py checks that against
// oracle/_ref/libbwaref.so on the CPU (no GPU needed: regions come from the oracle).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <atomic>
#include <thread>
#include <vector>
#include "../../../include/bwagpu.h"

namespace hostmem {

// flags of mem_opt_t::flag (bwamem.h:40-50)
enum { F_PE = 0x2, F_NOPAIRING = 0x4, F_ALL = 0x8, F_NO_MULTI = 0x10, F_NO_RESCUE = 0x20, F_REF_HDR = 0x100, F_SOFTCLIP = 0x200,
	   F_SMARTPE = 0x400, F_PRIMARY5 = 0x800, F_KEEP_SUPP_MAPQ = 0x1000, F_XB = 0x2000 };

struct Contig { int64_t offset; int32_t len, n_ambs, is_alt; uint32_t gi; std::string name, anno; };

// what bntseq_t + pac give the finalize code (bntseq.h:41-64)
struct RefSeqs {
	int64_t l_pac = 0;
	std::vector<Contig> ctg;
	std::vector<uint8_t> pac;
	int pos2rid(int64_t pos_f) const;                              // bns_pos2rid (bntseq.c:354-368)
	void get_seq(int64_t beg, int64_t end, std::vector<uint8_t> &out) const;   // bns_get_seq (bntseq.c:403-424)
	bool fetch_seq(int64_t &beg, int64_t mid, int64_t &end, int &rid, std::vector<uint8_t> &out) const;   // bns_fetch_seq (:426-451)
};

struct Pestat { int low, high, failed; double avg, std; };          // == mem_pestat_t (bwamem.h:108-112)

// SAM text under construction (the reference's kstring_t, kstring.h): a plain growing byte buffer.  Not std::string: a record is ~25 short
// fields, and a checked append with a terminator written per field -- plus resize()'s zero fill ahead of the base and quality loops --
// was half of what a record costs.  need(k) makes room for k more bytes and returns where they go; the caller adds what it wrote to n.
struct SamText {
	char *d = nullptr; size_t n = 0, cap = 0;
	SamText() = default;
	SamText(const SamText&) = delete; SamText &operator=(const SamText&) = delete;
	~SamText() { free(d); }
	void clear() { n = 0; }
	size_t size() const { return n; }
	const char *data() const { return d ? d : ""; }
	void reserve(size_t c) { if (c > cap) grow(c); }
	char *need(size_t k) { if (n + k > cap) grow(n + k + (cap > 256 ? cap : 256)); return d + n; }
	void append(const char *p, size_t k) { memcpy(need(k), p, k); n += k; }
	void append(const char *b, const char *e) { append(b, (size_t)(e - b)); }
	SamText &operator+=(char c) { *need(1) = c; ++n; return *this; }
	SamText &operator+=(const char *z) { append(z, strlen(z)); return *this; }
	SamText &operator+=(const std::string &z) { append(z.data(), z.size()); return *this; }
private:
	void grow(size_t c) { char *p = (char*)realloc(d, c); if (!p) { fprintf(stderr, "[E::SamText] out of memory\n"); abort(); } d = p; cap = c; }
};

struct Aln {   // == the information of mem_aln_t (bwamem.h:114-126)
	int64_t pos = -1; int rid = -1, flag = 0; bool is_rev = false, is_alt = false; int mapq = 0, NM = 0;
	std::vector<uint32_t> cigar; std::string md; std::string xa; bool has_xa = false;
	int score = 0, sub = 0, alt_sc = 0;
};

// Device-computed global alignments of a read's regions (bwagpu_batch_cigars): regs[k] as downloaded, cigs[k] its result.
// reg2aln looks a region up by the fields its band-doubling loop depends on and skips the DP when it finds a usable entry;
// the result is the same either way.
struct CigHints { const bwagpu_alnreg_t *regs; const bwagpu_cigar_t *cigs; int n; const uint32_t *ops = nullptr; /* the batch's operation array: records with 7..64 operations (bwagpu_batch_cigar_ops) */ };

struct Read {   // == bseq1_t as the finalize code needs it
	const char *name; const char *comment; const uint8_t *seq /* nt4 codes */; const char *qual; int l_seq;
	const CigHints *hints = nullptr;
	const bwagpu_matesw_t *msw = nullptr; int n_msw = 0;   // device-computed mate-rescue alignments of this read (bwagpu_batch_matesw)
};

typedef std::vector<bwagpu_alnreg_t> Regs;

// kt_for-like (kthread.c:49): the per-read outputs are independent, so any work split gives the same result
template <class F> static inline void parallel_for(int n_threads, long n, F f)
{
	if (n_threads <= 1 || n <= 1) { for (long i = 0; i < n; ++i) f(i); return; }
	std::atomic<long> next(0);
	std::vector<std::thread> th;
	for (int t = 0; t < n_threads; ++t)
		th.emplace_back([&]() { for (;;) { long i = next.fetch_add(16); if (i >= n) break; long e = i + 16 < n ? i + 16 : n; for (; i < e; ++i) f(i); } });
	for (auto &t : th) t.join();
}

uint64_t hash_64(uint64_t key);                                     // utils.h:98-109
int mark_primary_se(const bwagpu_opt_t &opt, Regs &a, int64_t id);  // bwamem.c:547-584
void reorder_primary5(int T, Regs &a);                              // bwamem.c:1008-1030
int approx_mapq_se(const bwagpu_opt_t &opt, const bwagpu_alnreg_t &a);   // bwamem.c:982-1006
void host_region_cigar(const bwagpu_opt_t &opt, const RefSeqs &ref, const uint8_t *query, const bwagpu_alnreg_t &ar, bwagpu_cigar_t *out, std::vector<uint32_t> *ext);   // == one bwagpu_batch_cigars record (+ its entries of the operation array)
Aln reg2aln(const bwagpu_opt_t &opt, const RefSeqs &ref, int l_query, const uint8_t *query, const bwagpu_alnreg_t *ar, const CigHints *hints = nullptr);   // bwamem.c:1119-1189
void aln2sam(const bwagpu_opt_t &opt, const RefSeqs &ref, SamText &out, const Read &s, const std::vector<Aln> &list, int which, const Aln *mate, const char *rg_id);   // bwamem.c:851-976
void reg2sam(const bwagpu_opt_t &opt, const RefSeqs &ref, SamText &out, const Read &s, Regs &a, int extra_flag, const Aln *mate, const char *rg_id);   // bwamem.c:1033-1079
void pestat_flat(const bwagpu_opt_t &opt, int64_t l_pac, int n, const bwagpu_alnreg_t *all, const int64_t *roff, Pestat pes[4], bool verbose, int n_threads = 1);
void attach_matesw(int n, Read *reads, const bwagpu_matesw_t *recs, int64_t n_recs, std::vector<bwagpu_matesw_t> &sorted);
int64_t host_matesw_records(const bwagpu_opt_t &opt, const RefSeqs &ref, int n, const uint8_t *seqs, const int64_t *off, const bwagpu_alnreg_t *all, const int64_t *roff,
							const Pestat pes[4], bwagpu_matesw_t *out, int64_t cap);   // == bwagpu_batch_matesw, on the host   // bwamem_pair.c:72-135
int sam_pe(const bwagpu_opt_t &opt, const RefSeqs &ref, const Pestat pes[4], uint64_t id, const Read s[2], Regs a[2], SamText *const out[2] /* may be one buffer twice: the first read's records all precede the second's */, const char *rg_id);   // bwamem_pair.c:276-419

// DP kernels of the finalize stage
int ksw_global2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int w, std::vector<uint32_t> *cigar);   // ksw.c:540-642
struct KswResult { int score, te, qe, score2, te2, tb, qb; };
KswResult ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra);   // ksw.c:379-400
int sort_dedup_nopatch(const bwagpu_opt_t &opt, Regs &a);           // mem_sort_dedup_patch with bns == 0 (bwamem_pair.c:201)

template <class T, class LT> void introsort(T *a, long n, LT lt);   // ks_introsort (ksort.h:176-226), defined in host_sort.h

bool load_refseqs(const std::string &prefix, RefSeqs &out, std::string &err);

}  // namespace hostmem
