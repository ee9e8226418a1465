// host_api.cpp -- batch-level finalize (the reference's worker2 loop, bwamem.c:1217-1233, 1256-1260) over a thread pool,
// and a flat C entry point used by the tests to compare this host code with the reference on the CPU.
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <thread>
#include "bwamem_host.h"

namespace hostmem {

// the regions of all reads of a batch (flat, read i's at all[roff[i] .. roff[i+1])) -> SAM text per read (sam[i]); PE when opt.flag & F_PE (mates
// interleaved).  A worker copies a read's regions into a list of its own that it keeps from read to read (the stage's functions reorder, flag
// and extend that list): per-read lists built by one thread and released by another cost two allocator round trips per read, and
// the cross-thread releases contend -- a quarter of the stage's time on 8 threads.
void finalize_batch(const bwagpu_opt_t &opt, const RefSeqs &ref, int64_t n_processed, int n, const Read *reads, const bwagpu_alnreg_t *all, const int64_t *roff,
					const Pestat *pes0, int n_threads, const char *rg_id, std::vector<std::string> &sam, bool verbose)
{
	sam.assign(n, std::string());
	if (opt.flag & F_PE) {
		Pestat pes[4];
		if (pes0) memcpy(pes, pes0, sizeof pes); else pestat_flat(opt, ref.l_pac, n, all, roff, pes, verbose, n_threads < 4 ? n_threads : 4);
		parallel_for(n_threads, n >> 1, [&](long i) {
			thread_local Regs a[2];
			a[0].assign(all + roff[i << 1], all + roff[(i << 1) + 1]); a[1].assign(all + roff[(i << 1) + 1], all + roff[(i << 1) + 2]);
			thread_local SamText out[2];
			out[0].clear(); out[1].clear();
			SamText *each[2] = { &out[0], &out[1] };
			sam_pe(opt, ref, pes, (uint64_t)((n_processed >> 1) + i), &reads[i << 1], a, each, rg_id);
			sam[i << 1].assign(out[0].data(), out[0].size()); sam[i << 1 | 1].assign(out[1].data(), out[1].size());
		});
	} else {
		parallel_for(n_threads, n, [&](long i) {
			thread_local Regs a;
			a.assign(all + roff[i], all + roff[i + 1]);
			mark_primary_se(opt, a, n_processed + i);
			if (opt.flag & F_PRIMARY5) reorder_primary5(opt.T, a);
			thread_local SamText out;
			out.clear();
			reg2sam(opt, ref, out, reads[i], a, 0, 0, rg_id);
			sam[i].assign(out.data(), out.size());
		});
	}
}

// The same, delivering the batch's SAM text in input order as one string per chunk of `chunk` reads (an even number) instead of one per
// read: the per-read strings were two allocator round trips per read, the second of them a release on the writer's thread.  A read's text
// ends at its first NUL, as it does when the reference fputs() it (fastmap.c:116; the letter of base code 5 is a NUL).
void finalize_batch_chunks(const bwagpu_opt_t &opt, const RefSeqs &ref, int64_t n_processed, int n, const Read *reads, const bwagpu_alnreg_t *all, const int64_t *roff,
						   const Pestat *pes0, int n_threads, const char *rg_id, int chunk, std::vector<std::string> &text, bool verbose)
{
	if (chunk < 2) chunk = 2;
	chunk &= ~1;
	const long n_chunks = ((long)n + chunk - 1) / chunk;
	text.resize((size_t)n_chunks);                    // (strings a caller hands back in keep their capacity: no fresh pages to fault in)
	const bool pe = (opt.flag & F_PE) != 0;
	Pestat pes[4];
	if (pe) { if (pes0) memcpy(pes, pes0, sizeof pes); else pestat_flat(opt, ref.l_pac, n, all, roff, pes, verbose, n_threads < 4 ? n_threads : 4); }
	auto put = [](std::string &dst, const SamText &s) { dst.append(s.data(), strnlen(s.data(), s.size())); };
	std::atomic<long> next(0);
	auto work = [&]() {
		Regs a[2]; SamText out[2], buf;
		for (;;) {
			const long c = next.fetch_add(1);
			if (c >= n_chunks) break;
			const int lo = (int)(c * chunk), hi = lo + chunk < n ? lo + chunk : n;
			std::string &dst = text[(size_t)c];
			// The chunk's records are written one after the other into one buffer that stays in this thread's cache and reach the string in one
			// copy.  Only if some record holds a NUL (a base of code 5) is the chunk redone read by read, each read's text cut at its first NUL.
			buf.clear();
			if (pe) {
				SamText *both[2] = { &buf, &buf };          // (mem_sam_pe emits all records of the first read, then all of the second)
				for (int i = lo; i + 1 < hi; i += 2) {
					a[0].assign(all + roff[i], all + roff[i + 1]); a[1].assign(all + roff[i + 1], all + roff[i + 2]);
					sam_pe(opt, ref, pes, (uint64_t)((n_processed + i) >> 1), &reads[i], a, both, rg_id);
				}
			} else {
				for (int i = lo; i < hi; ++i) {
					a[0].assign(all + roff[i], all + roff[i + 1]);
					mark_primary_se(opt, a[0], n_processed + i);
					if (opt.flag & F_PRIMARY5) reorder_primary5(opt.T, a[0]);
					reg2sam(opt, ref, buf, reads[i], a[0], 0, 0, rg_id);
				}
			}
			if (!memchr(buf.data(), 0, buf.size())) { dst.assign(buf.data(), buf.size()); continue; }
			dst.clear();
			if (dst.capacity() < (size_t)(hi - lo) * 720) dst.reserve((size_t)(hi - lo) * 720);
			if (pe) {
				SamText *each[2] = { &out[0], &out[1] };
				for (int i = lo; i + 1 < hi; i += 2) {
					a[0].assign(all + roff[i], all + roff[i + 1]); a[1].assign(all + roff[i + 1], all + roff[i + 2]);
					out[0].clear(); out[1].clear();
					sam_pe(opt, ref, pes, (uint64_t)((n_processed + i) >> 1), &reads[i], a, each, rg_id);
					put(dst, out[0]); put(dst, out[1]);
				}
			} else {
				for (int i = lo; i < hi; ++i) {
					a[0].assign(all + roff[i], all + roff[i + 1]);
					mark_primary_se(opt, a[0], n_processed + i);
					if (opt.flag & F_PRIMARY5) reorder_primary5(opt.T, a[0]);
					out[0].clear();
					reg2sam(opt, ref, out[0], reads[i], a[0], 0, 0, rg_id);
					put(dst, out[0]);
				}
			}
		}
	};
	if (n_threads <= 1 || n_chunks <= 1) work();
	else {      // (never more threads than chunks: a 100-read batch under -t 64 is two chunks)
		const int nt = n_threads < n_chunks ? n_threads : n_chunks;
		std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back(work); for (auto &t : th) t.join();
	}
}

// group bwagpu_batch_matesw records by the read they align and attach the slices to the reads
void attach_matesw(int n, Read *reads, const bwagpu_matesw_t *recs, int64_t n_recs, std::vector<bwagpu_matesw_t> &sorted)
{
	std::vector<int64_t> start((size_t)n + 1, 0);
	for (int64_t k = 0; k < n_recs; ++k) if (recs[k].r >= 0 && recs[k].read >= 0 && recs[k].read < n) ++start[recs[k].read + 1];
	for (int i = 0; i < n; ++i) start[i + 1] += start[i];
	sorted.resize((size_t)start[n]);
	std::vector<int64_t> fill(start.begin(), start.end() - 1);
	for (int64_t k = 0; k < n_recs; ++k) if (recs[k].r >= 0 && recs[k].read >= 0 && recs[k].read < n) sorted[(size_t)fill[recs[k].read]++] = recs[k];
	for (int i = 0; i < n; ++i) { reads[i].msw = sorted.data() + start[i]; reads[i].n_msw = (int)(start[i + 1] - start[i]); }
}

}  // namespace hostmem

using namespace hostmem;

extern "C" {

void *bwamem_host_create(const char *prefix)
{
	RefSeqs *r = new RefSeqs();
	std::string err;
	if (!load_refseqs(prefix, *r, err)) { fprintf(stderr, "[E::%s] %s\n", __func__, err.c_str()); delete r; return 0; }
	return r;
}
void bwamem_host_destroy(void *h) { delete (RefSeqs*)h; }
void bwamem_host_set_alt(void *h, int rid, int flag) { ((RefSeqs*)h)->ctg[rid].is_alt = flag; }

// same shape as refshim_regs2sam (oracle/ref_shim.c): names NUL-separated, seqs are nt4 codes, regs flat in read order
char *bwamem_host_regs2sam(void *h, const bwagpu_opt_t *opt, int64_t n_processed, int n, const char *names, const uint8_t *seqs, const char *quals,
						   const int64_t *off, const int32_t *counts, const bwagpu_alnreg_t *regs, const Pestat *pes0, int n_threads, int64_t *out_len,
						   const bwagpu_cigar_t *cigs /* optional: bwagpu_batch_cigars output, parallel to regs */,
						   const bwagpu_matesw_t *msw /* optional: bwagpu_batch_matesw output */, int64_t n_msw,
						   const uint32_t *cig_ops /* optional: bwagpu_batch_cigar_ops output (records with more than 6 operations) */)
{
	const RefSeqs &ref = *(RefSeqs*)h;
	std::vector<Read> reads(n); std::vector<CigHints> hints(n); std::vector<int64_t> roffs((size_t)n + 1, 0);
	const char *nm = names; int64_t roff = 0;
	for (int i = 0; i < n; ++i) {
		reads[i].name = nm; nm += strlen(nm) + 1;
		reads[i].comment = 0; reads[i].seq = seqs + off[i]; reads[i].qual = quals ? quals + off[i] : 0; reads[i].l_seq = (int)(off[i + 1] - off[i]);
		if (cigs) { hints[i].regs = regs + roff; hints[i].cigs = cigs + roff; hints[i].n = counts[i]; hints[i].ops = cig_ops; reads[i].hints = &hints[i]; }
		roff += counts[i]; roffs[i + 1] = roff;
	}
	std::vector<bwagpu_matesw_t> msw_sorted;
	if (msw) attach_matesw(n, reads.data(), msw, n_msw, msw_sorted);
	std::vector<std::string> sam;
	const bool trace = getenv("BWAMEM_HOST_TRACE") != nullptr;      // diagnostics: time of the finalize stage proper
	const auto t0 = std::chrono::steady_clock::now();
	if (getenv("BWAMEM_HOST_BY_READ")) finalize_batch(*opt, ref, n_processed, n, reads.data(), regs, roffs.data(), pes0, n_threads, 0, sam, false);   // (the per-read form, which the command line uses for -p batches)
	else finalize_batch_chunks(*opt, ref, n_processed, n, reads.data(), regs, roffs.data(), pes0, n_threads, 0, 64, sam, false);
	if (trace) fprintf(stderr, "[host] finalize_batch: %d reads, %d threads, %.3f s\n", n, n_threads, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
	size_t tot = 0;
	for (auto &s : sam) tot += s.size();
	char *out = (char*)malloc(tot + 1); size_t p = 0;
	for (auto &s : sam) { memcpy(out + p, s.data(), s.size()); p += s.size(); }
	out[tot] = 0; *out_len = (int64_t)tot;
	return out;
}

void bwamem_host_free(void *p) { free(p); }

// mem_pestat of a batch given as flat regions (the insert-size windows bwagpu_batch_matesw needs); pes = Pestat[4]
void bwamem_host_pestat(void *h, const bwagpu_opt_t *opt, int n, const int32_t *counts, const bwagpu_alnreg_t *regs, Pestat *pes)
{
	std::vector<int64_t> roff((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) roff[i + 1] = roff[i] + counts[i];
	pestat_flat(*opt, ((RefSeqs*)h)->l_pac, n, regs, roff.data(), pes, false);
}

// the records bwagpu_batch_matesw should produce, computed on the host (order: by pair, end, anchor, orientation)
int64_t bwamem_host_matesw_records(void *h, const bwagpu_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off, const int32_t *counts, const bwagpu_alnreg_t *regs,
								   const Pestat *pes, bwagpu_matesw_t *out, int64_t cap)
{
	std::vector<int64_t> roff((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) roff[i + 1] = roff[i] + counts[i];
	return host_matesw_records(*opt, *(RefSeqs*)h, n, seqs, off, regs, roff.data(), pes, out, cap);
}

// one bwagpu_cigar_t per region, computed by the host code (reference for bwagpu_batch_cigars; see host_region_cigar)
// ops/ops_cap: optional operation array for records with 7..64 operations (returns the number of entries used; without it such
// regions are reported unserved, reason 3)
int64_t bwamem_host_region_cigars(void *h, const bwagpu_opt_t *opt, int n, const uint8_t *seqs, const int64_t *off, const int32_t *counts, const bwagpu_alnreg_t *regs, bwagpu_cigar_t *out,
								  uint32_t *ops, int64_t ops_cap)
{
	const RefSeqs &ref = *(RefSeqs*)h;
	std::vector<uint32_t> ext;
	int64_t k = 0;
	for (int i = 0; i < n; ++i)
		for (int j = 0; j < counts[i]; ++j, ++k) host_region_cigar(*opt, ref, seqs + off[i], regs[k], out + k, ops ? &ext : nullptr);
	if (ops && (int64_t)ext.size() <= ops_cap && !ext.empty()) memcpy(ops, ext.data(), ext.size() * 4);
	return (int64_t)ext.size();
}

void bwamem_host_ksw_align2(int qlen, const uint8_t *query, int tlen, const uint8_t *target, const int8_t *mat, int o_del, int e_del, int o_ins, int e_ins, int xtra, int out[7])
{
	KswResult r = ksw_align2(qlen, query, tlen, target, mat, o_del, e_del, o_ins, e_ins, xtra);
	out[0] = r.score; out[1] = r.te; out[2] = r.qe; out[3] = r.score2; out[4] = r.te2; out[5] = r.tb; out[6] = r.qb;
}

}  // extern "C"
