// main_mem.cpp -- `bwa-amd mem`: the stand-alone command line around libbwagpu.so + the host finalize code.
// Mirrors the reference's `bwa mem` (fastmap.c:141-406): same option letters and meaning, same batching rule
// (chunk_size * n_threads bases per batch unless -K, fastmap.c:394), same SAM header (bwa.c:407-439), so that for the same
// input and the same -K the output equals `bwa mem`'s except for the @PG line.
#include <signal.h>
#include <ctype.h>
#include <errno.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/mman.h>
#include <getopt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <functional>
#include <vector>
#include "bwamem_host.h"

namespace hostmem {
void finalize_batch(const bwagpu_opt_t &opt, const RefSeqs &ref, int64_t n_processed, int n, const Read *reads, const bwagpu_alnreg_t *all, const int64_t *roff,
					const Pestat *pes0, int n_threads, const char *rg_id, std::vector<std::string> &sam, bool verbose);
void finalize_batch_chunks(const bwagpu_opt_t &opt, const RefSeqs &ref, int64_t n_processed, int n, const Read *reads, const bwagpu_alnreg_t *all, const int64_t *roff,
						   const Pestat *pes0, int n_threads, const char *rg_id, int chunk, std::vector<std::string> &text, bool verbose);
}
using namespace hostmem;

static int g_verbose = 3;
static int g_device_matesw = 1;   // BWAGPU_CLI_MATESW=0: the host runs every mate-rescue alignment itself (same output)
static int g_device_cigars = 1;   // BWAGPU_CLI_CIGARS=0: the host computes every CIGAR itself (same output)

// ---- options -----------------------------------------------------------------------------------------------------------
static void fill_scmat(int a, int b, int8_t mat[25])
{	// bwa_fill_scmat (bwa.c:136-145)
	int k = 0;
	for (int i = 0; i < 4; ++i) { for (int j = 0; j < 4; ++j) mat[k++] = i == j ? a : -b; mat[k++] = -1; }
	for (int j = 0; j < 5; ++j) mat[k++] = -1;
}

static void opt_init(bwagpu_opt_t *o)
{	// mem_opt_init (bwamem.c:74-110)
	memset(o, 0, sizeof *o);
	o->a = 1; o->b = 4; o->o_del = o->o_ins = 6; o->e_del = o->e_ins = 1; o->w = 100; o->T = 30; o->zdrop = 100;
	o->pen_unpaired = 17; o->pen_clip5 = o->pen_clip3 = 5; o->max_mem_intv = 20; o->min_seed_len = 19; o->split_width = 10;
	o->max_occ = 500; o->max_chain_gap = 10000; o->max_ins = 10000; o->mask_level = 0.50f; o->drop_ratio = 0.50f;
	o->XA_drop_ratio = 0.80f; o->split_factor = 1.5f; o->chunk_size = 10000000; o->n_threads = 1; o->max_XA_hits = 5;
	o->max_XA_hits_alt = 200; o->max_matesw = 50; o->mask_level_redun = 0.95f; o->min_chain_weight = 0;
	o->max_chain_extend = 1 << 30; o->mapQ_coef_len = 50; o->mapQ_coef_fac = (int)log(o->mapQ_coef_len);
	fill_scmat(o->a, o->b, o->mat);
}

static std::string unescape(const char *s)
{	// bwa_escape (bwa.c:441-455)
	std::string r;
	for (const char *p = s; *p; ++p) {
		if (*p == '\\') { ++p; if (*p == 't') r += '\t'; else if (*p == 'n') r += '\n'; else if (*p == 'r') r += '\r'; else if (*p == '\\') r += '\\'; if (!*p) break; }
		else r += *p;
	}
	return r;
}

// ---- FASTA/FASTQ input (what kseq_read + bseq_read deliver, kseq.h:175-220, bwa.c:79-112) ----------------------------
// One record = four NUL-terminated strings in its batch's text arena (no per-record allocation: the reader is a single
// thread and sets the pace of the whole pipeline).
struct Seq { size_t name = 0, comment = 0, seq = 0, qual = 0; int l_name = 0, l_seq = 0, l_qual = 0; bool has_comment = false, has_qual = false;
	char *base = nullptr;   /* the text the offsets refer to when it is not the batch's own arena (a parsed block, see ParFile) */ };
typedef std::vector<char> Arena;

// does any of the n bytes at p hold a blank or control character (<= ' ')?  Eight bytes per step.
static bool has_blank(const char *p, size_t n) {
	const uint64_t ones = ~0ull / 255;
	size_t i = 0;
	for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); if ((w - ones * 0x21) & ~w & (ones * 0x80)) return true; }
	for (; i < n; ++i) if ((unsigned char)p[i] <= ' ') return true;
	return false;
}

// One plain four-line FASTQ record at b (the text ends at e): appended to A as name\0 comment\0 bases\0 qualities\0, its successor's
// address returned -- or null, nothing appended, when the record is not of that kind or not completely there (the general reader
// decides then).  Both the streaming reader and the block parsers below go through here, so they accept exactly the same records.
static const char *parse_fast_record(const char *b, const char *e, Seq &s, Arena &A)
{
	if (b >= e || *b != '@') return nullptr;
	const char *n1 = (const char*)memchr(b, '\n', (size_t)(e - b)); if (!n1 || n1 + 1 >= e) return nullptr;
	const char *n2 = (const char*)memchr(n1 + 1, '\n', (size_t)(e - n1 - 1)); if (!n2 || n2 + 1 >= e || n2[1] != '+') return nullptr;
	const char *n3 = (const char*)memchr(n2 + 1, '\n', (size_t)(e - n2 - 1)); if (!n3 || n3 + 1 >= e) return nullptr;
	const char *n4 = (const char*)memchr(n3 + 1, '\n', (size_t)(e - n3 - 1)); if (!n4) return nullptr;
	const char *sq = n1 + 1, *ql = n3 + 1;
	const size_t ls = (size_t)(n2 - sq), lq = (size_t)(n4 - ql);
	if (ls == 0 || ls != lq || n1[-1] == '\r' || n2[-1] == '\r' || n4[-1] == '\r') return nullptr;
	if (sq[0] == '>' || sq[0] == '+' || sq[0] == '@') return nullptr;              // the general reader treats these as record structure
	if (has_blank(sq, ls)) return nullptr;                                         // blanks / control characters: general path
	const char *h = b + 1, *he = n1, *ne = h;
	while (ne < he && !isspace((unsigned char)*ne)) ++ne;
	s = Seq();
	s.name = A.size(); s.l_name = (int)(ne - h);
	A.insert(A.end(), h, ne); A.push_back(0);
	s.comment = A.size();
	if (ne < he) { s.has_comment = true; A.insert(A.end(), ne + 1, he); }
	A.push_back(0);
	s.seq = A.size(); s.l_seq = (int)ls; A.insert(A.end(), sq, sq + ls); A.push_back(0);
	s.qual = A.size(); s.l_qual = (int)lq; s.has_qual = true; A.insert(A.end(), ql, ql + lq); A.push_back(0);
	return n4 + 1;
}

// A small pool of worker threads for the input stage: parsing blocks of plain FASTQ files (ParFile) and inflating the blocks of BGZF files (BgzfPipe).
// A task carries its owner, so that a file that is closed early can take its queued tasks back.
struct ParPool {
	struct Task { void *owner; std::function<void()> run; };
	std::mutex m; std::condition_variable cv; std::deque<Task> q; bool stop = false; std::vector<std::thread> th;
	explicit ParPool(int n) { for (int i = 0; i < n; ++i) th.emplace_back([this] { loop(); }); }
	~ParPool() { { std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all(); for (auto &t : th) t.join(); }
	void push(void *owner, std::function<void()> f) { { std::lock_guard<std::mutex> l(m); q.push_back(Task{owner, std::move(f)}); } cv.notify_one(); }
	int drop(void *owner) { std::lock_guard<std::mutex> l(m); int n = 0; for (auto it = q.begin(); it != q.end();) if (it->owner == owner) { it = q.erase(it); ++n; } else ++it; return n; }
	void loop() {
		for (;;) {
			Task t;
			{
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return stop || !q.empty(); });
				if (q.empty()) return;
				t = std::move(q.front()); q.pop_front();
			}
			t.run();
		}
	}
};

// ---- BGZF input (bgzip; the block-compressed gzip of htslib): independent blocks of at most 64 KiB, each a gzip member whose header says how long it is
// (extra field "BC", SAM spec 4.1), so the blocks of a file can be inflated side by side -- one zlib stream inflates ~210 MB/s of FASTQ on a host core,
// 0.7 M records/s per file, which is what `bwa-amd mem` delivered for ANY gzip input ("compressed input at the product's speed").  The file is
// mapped; one thread walks the block headers and queues groups of blocks (~0.5 MB compressed) to the pool; the reader takes the inflated groups in file
// order.  What comes out is the same byte stream gzread would deliver (bseq_read over kseq.h's gzread, bwa.c:79-112) and goes through the same parser.
// A plain gzip file (one stream, no block sizes) cannot be split and keeps its single inflating thread.
struct BgzfGroup { const unsigned char *src = nullptr; size_t src_len = 0; std::vector<char> out; size_t out_len = 0; bool done = false, bad = false; };
struct BgzfPipe {
	int fd = -1; const unsigned char *map = nullptr; size_t size = 0; ParPool *pool = nullptr; std::string path;
	std::mutex m; std::condition_variable cv;
	std::deque<std::shared_ptr<BgzfGroup>> inflight; bool io_done = false, stop = false, bad = false; int pending = 0;
	std::thread io; std::shared_ptr<BgzfGroup> cur; size_t cpos = 0;
	// total length of the BGZF block at p (header .. trailer), or 0 if p does not start one
	static size_t block_len(const unsigned char *p, size_t n) {
		if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return 0;
		const size_t xlen = p[10] | (size_t)p[11] << 8;
		if (12 + xlen > n) return 0;
		for (size_t o = 12; o + 4 <= 12 + xlen;) {
			const size_t sl = p[o + 2] | (size_t)p[o + 3] << 8;
			if (p[o] == 'B' && p[o + 1] == 'C' && sl == 2 && o + 6 <= 12 + xlen) { const size_t bs = (p[o + 4] | (size_t)p[o + 5] << 8) + 1; return bs >= 12 + xlen + 8 && bs <= n ? bs : 0; }
			o += 4 + sl;
		}
		return 0;
	}
	static bool is_bgzf(const char *fn) {
		const int f = ::open(fn, O_RDONLY); if (f < 0) return false;
		unsigned char h[64]; const ssize_t n = pread(f, h, sizeof h, 0); struct stat st; const bool reg = fstat(f, &st) == 0 && S_ISREG(st.st_mode);
		::close(f);
		if (!reg || n < 18) return false;
		// (block_len wants the whole block in view: here only its header is, so the length is read without the bound)
		if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
		const size_t xlen = h[10] | (size_t)h[11] << 8;
		for (size_t o = 12; o + 6 <= 12 + xlen && o + 6 <= (size_t)n;) { const size_t sl = h[o + 2] | (size_t)h[o + 3] << 8; if (h[o] == 'B' && h[o + 1] == 'C' && sl == 2) return true; o += 4 + sl; }
		return false;
	}
	~BgzfPipe() {
		{ std::lock_guard<std::mutex> l(m); stop = true; }
		cv.notify_all();
		if (io.joinable()) io.join();
		if (pool) { const int n = pool->drop(this); std::unique_lock<std::mutex> l(m); pending -= n; cv.wait(l, [&] { return pending == 0; }); }
		if (map) munmap((void*)map, size);
		if (fd >= 0) ::close(fd);
	}
	bool open(const char *fn, ParPool *pl) {
		fd = ::open(fn, O_RDONLY); if (fd < 0) return false;
		struct stat st; if (fstat(fd, &st) != 0 || st.st_size <= 0) return false;
		size = (size_t)st.st_size;
		void *mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
		if (mp == MAP_FAILED) return false;
		madvise(mp, size, MADV_SEQUENTIAL);
		map = (const unsigned char*)mp; pool = pl; path = fn;
		io = std::thread([this] { scan(); });
		return true;
	}
	static void inflate_group(BgzfGroup &G) {
		size_t want = 0;
		for (size_t o = 0; o < G.src_len;) { const size_t bl = block_len(G.src + o, G.src_len - o); if (!bl) { G.bad = true; return; } const unsigned char *t = G.src + o + bl - 4; want += t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24; o += bl; }
		if (G.out.size() < want) G.out.resize(want);
		z_stream z; memset(&z, 0, sizeof z);
		if (inflateInit2(&z, -15) != Z_OK) { G.bad = true; return; }
		size_t w = 0;
		for (size_t o = 0; o < G.src_len && !G.bad;) {
			const size_t bl = block_len(G.src + o, G.src_len - o);
			const unsigned char *b = G.src + o, *t = b + bl - 8;
			const size_t xlen = b[10] | (size_t)b[11] << 8, isize = t[4] | (size_t)t[5] << 8 | (size_t)t[6] << 16 | (size_t)t[7] << 24;
			const uint32_t crc = t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16 | (uint32_t)t[3] << 24;
			z.next_in = (Bytef*)(b + 12 + xlen); z.avail_in = (uInt)(bl - 12 - xlen - 8);
			Bytef spare[8];                                            // (an empty block -- bgzip's end-of-file marker -- still needs somewhere to "write")
			z.next_out = isize ? (Bytef*)G.out.data() + w : spare; z.avail_out = isize ? (uInt)isize : (uInt)sizeof spare;
			const int rc = inflate(&z, Z_FINISH);
			if (rc != Z_STREAM_END || z.total_out != isize || (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)G.out.data() + w, (uInt)isize) != crc) G.bad = true;
			w += isize; o += bl;
			inflateReset(&z);
		}
		inflateEnd(&z);
		G.out_len = w;
	}
	void scan() {
		size_t off = 0;
		const size_t group_bytes = getenv("BWAGPU_CLI_BGZF_GROUP") ? (size_t)atoll(getenv("BWAGPU_CLI_BGZF_GROUP")) : (size_t)512 << 10;   // (tests: a block per group)
		while (off < size) {
			{ std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return stop || inflight.size() < 16; }); if (stop) break; }
			size_t end = off;
			while (end < size && end - off < group_bytes) { const size_t bl = block_len(map + end, size - end); if (!bl) break; end += bl; }
			if (end == off) { std::lock_guard<std::mutex> l(m); bad = true; break; }       // not a BGZF block where one must start: a corrupt or truncated file
			std::shared_ptr<BgzfGroup> G(new BgzfGroup());
			G->src = map + off; G->src_len = end - off;
			BgzfGroup *gp = G.get();
			{ std::lock_guard<std::mutex> l(m); inflight.push_back(G); ++pending; }
			auto work = [this, gp] { inflate_group(*gp); std::lock_guard<std::mutex> l(m); gp->done = true; --pending; cv.notify_all(); };
			if (pool) pool->push(this, work); else work();
			off = end;
		}
		{ std::lock_guard<std::mutex> l(m); io_done = true; }
		cv.notify_all();
	}
	// the next bytes of the inflated stream (blocking; in file order): > 0 bytes, 0 at the end, -1 for a damaged file
	int read(char *dst, size_t cap) {
		for (;;) {
			if (cur && cpos < cur->out_len) { const size_t n = cur->out_len - cpos < cap ? cur->out_len - cpos : cap; memcpy(dst, cur->out.data() + cpos, n); cpos += n; return (int)n; }
			cur.reset();
			std::unique_lock<std::mutex> l(m);
			cv.wait(l, [&] { return (!inflight.empty() && inflight.front()->done) || (inflight.empty() && io_done); });
			if (inflight.empty()) return bad ? -1 : 0;
			cur = std::move(inflight.front()); inflight.pop_front(); cpos = 0;
			l.unlock(); cv.notify_all();
			if (cur->bad) return -1;
		}
	}
};

struct Reader {
	std::unique_ptr<BgzfPipe> bg;      // a BGZF file: blocks inflated by the pool (else fp / raw_fd)
	gzFile fp = nullptr; int raw_fd = -1; std::vector<char> buf; int pos = 0, len = 0; int last = 0; bool eof = false;
	// The file is read (and, for gzip input, inflated) by a thread of its own, one buffer ahead of the parser: inflating a FASTQ stream
	// costs several times what parsing it does, and for paired input the two files' streams then inflate side by side.
	std::thread ahead; std::mutex m; std::condition_variable cv;
	std::vector<char> nbuf; int nlen = 0; bool nfull = false, stop = false;
	~Reader() {
		if (ahead.joinable()) { { std::lock_guard<std::mutex> l(m); stop = true; } cv.notify_all(); ahead.join(); }      // (a read in flight completes: the pipe below is still alive)
		bg.reset();
		if (fp) gzclose(fp); else if (raw_fd >= 0) ::close(raw_fd);
	}
	// A regular file that does not start with the gzip magic is read with read(2) straight into the buffer: zlib's transparent mode
	// would copy every byte twice more.  Everything else (gzip files, stdin) goes through zlib as in the reference (kseq.h over gzread,
	// fastmap.c:357-372).
	bool open(const char *fn, ParPool *pool = nullptr) {
		size_t cap = 1 << 20;
		if (getenv("BWAGPU_CLI_BUF")) { cap = (size_t)atoll(getenv("BWAGPU_CLI_BUF")); if (cap < 8) cap = 8; }   // (tests: records that straddle buffer ends)
		buf.resize(cap); nbuf.resize(cap);
		if (strcmp(fn, "-")) {
			const int fd = ::open(fn, O_RDONLY);
			if (fd < 0) return false;
			unsigned char magic[2] = { 0, 0 };
			struct stat st;
			if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && pread(fd, magic, 2, 0) >= 0 && !(magic[0] == 0x1f && magic[1] == 0x8b)) { raw_fd = fd; return true; }
			if (pool && !getenv("BWAGPU_CLI_NO_BGZF") && BgzfPipe::is_bgzf(fn)) {
				bg.reset(new BgzfPipe());
				if (bg->open(fn, pool)) { ::close(fd); return true; }
				bg.reset();
			}
			fp = gzdopen(fd, "r");
			if (!fp) ::close(fd);
		} else fp = gzdopen(fileno(stdin), "r");
		if (fp) gzbuffer(fp, 1 << 20);
		return fp != nullptr;
	}
	int read_some(char *dst, size_t cap) {
		int n;
		if (bg) n = bg->read(dst, cap);
		else if (raw_fd >= 0) { do n = (int)::read(raw_fd, dst, cap); while (n < 0 && errno == EINTR); }
		else n = gzread(fp, dst, (unsigned)cap);
		return n;
	}
	void read_ahead() {
		for (;;) {
			{ std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !nfull || stop; }); if (stop) return; }
			const int n = read_some(nbuf.data(), nbuf.size());       // (nbuf belongs to this thread while !nfull)
			{ std::lock_guard<std::mutex> l(m); nlen = n; nfull = true; }
			cv.notify_all();
			if (n <= 0) return;
		}
	}
	bool fill() {
		if (eof) return false;
		if (!ahead.joinable()) ahead = std::thread([this] { read_ahead(); });
		{
			std::unique_lock<std::mutex> l(m);
			cv.wait(l, [&] { return nfull; });
			buf.swap(nbuf); len = nlen; nfull = false;
		}
		cv.notify_all();
		pos = 0;
		if (len < 0 && bg) { fprintf(stderr, "[E::%s] `%s' is a damaged or truncated BGZF file (a block does not inflate to its recorded size and checksum)\n", "main_mem", bg->path.c_str()); exit(EXIT_FAILURE); }
		if (len <= 0) { len = 0; eof = true; return false; }
		return true;
	}
	int getc_() { if (pos >= len && !fill()) return -1; return (unsigned char)buf[pos++]; }
	// append the bytes up to the next delimiter (newline, or any white space when `space`) to `out` and consume the delimiter;
	// returns the delimiter, or -1 at end of input.  Whole buffer spans are copied at once.
	int until(bool space, Arena *out) {
		for (;;) {
			if (pos >= len && !fill()) return -1;
			const char *b = buf.data() + pos; const int n = len - pos; int k;
			if (space) { for (k = 0; k < n && !isspace((unsigned char)b[k]); ++k) {} }
			else { const char *q = (const char*)memchr(b, '\n', (size_t)n); k = q ? (int)(q - b) : n; }
			if (out) out->insert(out->end(), b, b + k);
			pos += k;
			if (k < n) return (unsigned char)buf[pos++];
		}
	}
	// drop the characters of A[from..) for which `drop` holds
	template <class P> static void squeeze(Arena &A, size_t from, P drop) { size_t w = from; for (size_t i = from; i < A.size(); ++i) if (!drop((unsigned char)A[i])) A[w++] = A[i]; A.resize(w); }
	// The common case -- a four-line FASTQ record that lies completely in the buffer, without '\r' or blanks in the sequence --
	// located with four memchr calls and appended in bulk; anything else goes through the general reader below.
	bool read_fast(Seq &s, Arena &A) {
		if (last != 0 || pos >= len) return false;
		const char *next = parse_fast_record(buf.data() + pos, buf.data() + len, s, A);
		if (!next) return false;
		pos = (int)(next - buf.data());
		return true;
	}
	bool read(Seq &s, Arena &A) {
		if (read_fast(s, A)) return true;
		int c;
		if (last == 0) { while ((c = getc_()) != -1 && c != '>' && c != '@') {} if (c == -1) return false; last = c; }
		s = Seq();
		s.name = A.size();
		c = until(true, &A);
		size_t name_end = A.size();
		A.push_back(0);
		s.comment = A.size();
		if (c != '\n' && c != -1) { s.has_comment = true; until(false, &A); while (A.size() > s.comment && A.back() == '\r') A.pop_back(); }
		else while (name_end > s.name && A[name_end - 1] == '\r') A[--name_end] = 0;
		s.l_name = (int)(name_end - s.name);
		A.push_back(0);
		s.seq = A.size();
		while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			A.push_back((char)c);
			until(false, &A);
		}
		bool ws = false;
		for (size_t i = s.seq; i < A.size(); ++i) if (isspace((unsigned char)A[i])) { ws = true; break; }
		if (ws) squeeze(A, s.seq, [](unsigned char ch) { return isspace(ch) != 0; });
		s.l_seq = (int)(A.size() - s.seq);
		A.push_back(0);
		s.qual = A.size();
		if (c == '>' || c == '@') last = c; else last = 0;
		if (c != '+') { A.push_back(0); return true; }
		if (until(false, nullptr) == -1) { A.push_back(0); last = 0; return false; } // '+' line, then end of input: kseq_read's "no quality string" error (-2), the record is dropped
		s.has_qual = true;
		while (A.size() - s.qual < (size_t)s.l_seq) {
			const size_t before = A.size();
			c = until(false, &A);
			const bool got = A.size() > before;
			if (memchr(A.data() + before, '\r', A.size() - before)) squeeze(A, before, [](unsigned char ch) { return ch == '\r'; });
			if (c == -1 && !got) break;
		}
		s.l_qual = (int)(A.size() - s.qual);
		A.push_back(0);
		last = 0;
		// kseq_read returns -2 when the quality string and the sequence differ in length (kseq.h:219): bseq_read's loop ends there
		// (bwa.c:88), the record is dropped and the batch closed; the next batch resumes scanning for a header after the text the
		// quality loop has consumed.  Passing such a record on would also let the SAM writer read qual[0..l_seq) out of bounds.
		if (s.l_qual != s.l_seq) return false;
		return true;
	}
};

// ---- block-parallel input (BWAGPU_CLI_PARSE_THREADS=N; default 4) ---------------
// One thread parses ~8 M records per second; one MI355X takes half of that, a node of eight needs four times it.  A plain FASTQ file is
// therefore read in blocks that end on a record boundary, the blocks are parsed by a pool of threads (parse_fast_record, the streaming
// reader's own fast path) into arenas of their own, and the batch builder copies finished records out in file order -- one memcpy per
// record instead of the parse.  Exactness does not rest on finding the boundaries right: a block must be consumed to its last byte by
// records of the plain kind; the first byte that is not -- a record the fast path declines, or a cut that was not a record start, which
// leaves the block before it with an incomplete last record -- is, by induction from the file's first byte, the start of a record, and
// from there on the file goes through the streaming reader (lseek), whose general path has the reference's kseq semantics.  gzip
// input and pipes keep the streaming reader throughout.
struct ParBlock {
	const char *raw = nullptr; size_t len = 0; int64_t file_off = 0; bool last = false, whole = false;   // whole: the buffer held no cut (or the file ended) -- what follows the consumed part is not known to be a record start
	Arena text; std::vector<Seq> seqs; size_t end_at = 0;      // bytes of raw that became records
	bool parsed = false;
};
// blocks are shared: a batch keeps the blocks its records lie in (no copy of the text), the last owner hands a block back to this pool
static std::mutex g_blk_m; static std::vector<ParBlock*> g_blk_pool;
static std::shared_ptr<ParBlock> new_block()
{
	ParBlock *b = nullptr;
	{ std::lock_guard<std::mutex> l(g_blk_m); if (!g_blk_pool.empty()) { b = g_blk_pool.back(); g_blk_pool.pop_back(); } }
	if (!b) b = new ParBlock();
	return std::shared_ptr<ParBlock>(b, [](ParBlock *x) { std::lock_guard<std::mutex> l(g_blk_m); if (g_blk_pool.size() < 256) g_blk_pool.push_back(x); else delete x; });
}
struct ParFile {
	std::string path; int fd = -1; const char *map = nullptr; size_t size = 0; size_t blk = (size_t)4 << 20; ParPool *pool = nullptr;
	std::mutex m; std::condition_variable cv;
	std::deque<std::shared_ptr<ParBlock>> inflight;      // in file order; the front one is handed to the consumer once parsed
	int pending = 0;                                      // blocks queued in or being parsed by the pool
	bool io_done = false, stop = false;
	std::thread io;
	std::shared_ptr<ParBlock> cur; size_t ci = 0; long cur_batch = -1;   // consumer side (cur_batch: the batch that already holds cur)
	bool fallback = false; Reader ser; long n_par = 0;
	~ParFile() { shutdown(); if (map) munmap((void*)map, size); if (fd >= 0) ::close(fd); }
	void shutdown() {
		if (!pool) return;
		{ std::lock_guard<std::mutex> l(m); stop = true; }
		cv.notify_all();
		if (io.joinable()) io.join();
		const int taken = pool->drop(this);      // blocks of this file still waiting in the pool's queue are taken back; those being parsed are waited for
		std::unique_lock<std::mutex> l(m);
		pending -= taken;
		cv.wait(l, [&] { return pending == 0; });
	}
	// the last position in [1, n) that starts a line with '@' and whose next-but-one line starts with '+' (0: none)
	static size_t find_cut(const char *b, size_t n) {
		size_t p = n;
		while (p > 0) {
			const char *nl = (const char*)memrchr(b, '\n', p);      // the newest line start before p
			if (!nl) return 0;
			const size_t ls = (size_t)(nl - b) + 1;
			p = (size_t)(nl - b);
			if (ls >= n || b[ls] != '@') continue;
			const char *e1 = (const char*)memchr(b + ls, '\n', n - ls); if (!e1) continue;
			const char *e2 = (const char*)memchr(e1 + 1, '\n', (size_t)(b + n - e1 - 1)); if (!e2 || e2 + 1 >= b + n) continue;
			if (e2[1] == '+') return ls;
		}
		return 0;
	}
	void parse_block(ParBlock &B) {
		const char *b = B.raw, *e = b + B.len, *p = b;
		if (B.text.capacity() < B.len) B.text.reserve(B.len + 1024);
		Seq s;
		while (p < e) { const char *nx = parse_fast_record(p, e, s, B.text); if (!nx) break; B.seqs.push_back(s); p = nx; }
		B.end_at = (size_t)(p - b);
		std::lock_guard<std::mutex> l(m); B.parsed = true; --pending; cv.notify_all();   // (notified under the lock: shutdown() may let the file go as soon as pending is 0)
	}
	void read_blocks() {      // (cuts only: the file is mapped, its pages are first touched by the threads that parse them)
		size_t off = 0;
		for (;;) {
			{
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return stop || inflight.size() < 8; });
				if (stop) break;
			}
			std::shared_ptr<ParBlock> B = new_block();
			const size_t n = size - off < blk ? size - off : blk;
			const bool eof = off + n == size;
			B->raw = map + off; B->file_off = (int64_t)off; B->seqs.clear(); B->text.clear(); B->end_at = 0; B->parsed = false; B->last = eof; B->whole = eof;
			size_t cut = eof ? n : find_cut(map + off, n);
			if (!eof && cut == 0) { cut = n; B->whole = true; }       // no boundary in sight (records longer than a block, or not this kind of file)
			B->len = cut;
			off += cut;
			const bool end = eof || B->whole;
			ParBlock *raw_ptr = B.get();
			{ std::lock_guard<std::mutex> l(m); inflight.push_back(B); ++pending; }
			pool->push(this, [this, raw_ptr] { parse_block(*raw_ptr); });
			if (end) break;                                          // after a block without a cut the consumer goes to the streaming reader
		}
		{ std::lock_guard<std::mutex> l(m); io_done = true; }
		cv.notify_all();
	}
	// A mapped file that is truncated while it is being read raises SIGBUS in whichever thread touches the missing pages; without a handler that is a
	// silent crash of a long run.  The handler can only say so and leave (async-signal-safe calls only).
	// (Only a fault INSIDE one of the mapped inputs is reported that way: any other SIGBUS -- a driver mapping, a bug -- gets the default action back.)
	struct MappedRange { std::atomic<const char*> lo{nullptr}; std::atomic<size_t> len{0}; };
	static MappedRange *mapped_ranges() { static MappedRange r[8]; return r; }
	static void on_sigbus(int sig, siginfo_t *si, void *) {
		const char *a = si ? (const char*)si->si_addr : nullptr;
		bool ours = false;
		for (int k = 0; k < 8; ++k) { const char *lo = mapped_ranges()[k].lo.load(); const size_t n = mapped_ranges()[k].len.load(); if (lo && a >= lo && a < lo + n) ours = true; }
		if (!ours) { signal(sig, SIG_DFL); raise(sig); return; }
		static const char msg[] = "[bwa-amd] SIGBUS: an input file shrank while it was being read (plain FASTQ files are mapped); no SAM after this point is valid\n";
		if (::write(2, msg, sizeof msg - 1) < 0) {}
		_exit(74);                                                   // EX_IOERR
	}
	static void guard_mapped_input(const char *lo, size_t len) {
		static std::once_flag once;
		std::call_once(once, [] { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = on_sigbus; sa.sa_flags = SA_SIGINFO; sigemptyset(&sa.sa_mask); sigaction(SIGBUS, &sa, nullptr); });
		for (int k = 0; k < 8; ++k) { const char *none = nullptr; if (mapped_ranges()[k].lo.compare_exchange_strong(none, lo)) { mapped_ranges()[k].len = len; break; } }
	}
	bool open(const char *fn, ParPool *pl) {      // true: this file is read in blocks
		const int f = ::open(fn, O_RDONLY);
		if (f < 0) return false;
		unsigned char magic[2] = { 0, 0 }; struct stat st;
		if (!(fstat(f, &st) == 0 && S_ISREG(st.st_mode) && pread(f, magic, 2, 0) >= 0 && !(magic[0] == 0x1f && magic[1] == 0x8b))) { ::close(f); return false; }
		size = (size_t)st.st_size;
		if (size > 0) {
			void *mp = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, f, 0);
			if (mp == MAP_FAILED) { ::close(f); return false; }
			madvise(mp, size, MADV_SEQUENTIAL);
			map = (const char*)mp;
			guard_mapped_input(map, size);
#ifdef BWAGPU_CLI_TEST_HOOKS      // (the mock-runtime build of the tests only: the shipped program never touches its input files)
			if (getenv("BWAGPU_CLI_TEST_SHRINK") && truncate(fn, (off_t)(size / 8192 * 4096)) != 0) {}   // (the file loses its second half under the mapping)
#endif
		}
		fd = f; path = fn; pool = pl;
		if (getenv("BWAGPU_CLI_PAR_BLOCK")) { blk = (size_t)atoll(getenv("BWAGPU_CLI_PAR_BLOCK")); if (blk < 16) blk = 16; }   // (tests: cuts in every position)
		io = std::thread([this] { read_blocks(); });
		return true;
	}
	// the next record: in a parsed block (s.base set; the batch `hold` keeps the block) or, after the switch, appended to A
	bool read(Seq &s, Arena &A, std::vector<std::shared_ptr<ParBlock>> &hold, long batch_id) {
		for (;;) {
			if (fallback) return ser.read(s, A);
			if (!cur) {
				std::unique_lock<std::mutex> l(m);
				cv.wait(l, [&] { return (!inflight.empty() && inflight.front()->parsed) || (inflight.empty() && io_done); });
				if (inflight.empty()) return false;                  // (an empty file)
				cur = std::move(inflight.front()); inflight.pop_front(); ci = 0; cur_batch = -1;
				l.unlock(); cv.notify_all();
			}
			if (ci < cur->seqs.size()) {
				s = cur->seqs[ci++]; s.base = cur->text.data(); ++n_par;
				if (cur_batch != batch_id) { hold.push_back(cur); cur_batch = batch_id; }
				return true;
			}
			const bool clean = cur->end_at == cur->len && !cur->whole, at_eof = cur->last && cur->end_at == cur->len;
			const int64_t resume = cur->file_off + (int64_t)cur->end_at;
			cur.reset();
			if (clean) continue;
			if (at_eof) return false;
			// the rest of the file through the streaming reader, from the first byte that did not become a record
			shutdown();
			{ std::lock_guard<std::mutex> l(m); inflight.clear(); }
			if (!ser.open(path.c_str()) || ser.raw_fd < 0 || lseek(ser.raw_fd, (off_t)resume, SEEK_SET) < 0) { fprintf(stderr, "[E::%s] fail to re-open file `%s'.\n", "main_mem", path.c_str()); exit(EXIT_FAILURE); }
			if (getenv("BWAGPU_CLI_TRACE")) fprintf(stderr, "[D::input] %s: %ld records from parsed blocks, the streaming reader takes over at byte %ld\n", path.c_str(), n_par, (long)resume);
			fallback = true;
		}
	}
};
// an input file: read in blocks when it is a plain file and a pool is given, through the streaming reader otherwise
struct Source {
	Reader ser; std::unique_ptr<ParFile> par;
	bool open(const char *fn, ParPool *pool) {
		if (pool && strcmp(fn, "-")) { par.reset(new ParFile()); if (par->open(fn, pool)) return true; par.reset(); }
		return ser.open(fn, pool);
	}
	bool read(Seq &s, Arena &A, std::vector<std::shared_ptr<ParBlock>> &hold, long batch_id) { return par ? par->read(s, A, hold, batch_id) : ser.read(s, A); }
};

// "/1" and "/2" name suffixes are dropped (trim_readno, bwa.c:66-71)
static void trim_readno(Seq &s, Arena &A) { char *T = s.base ? s.base : A.data(); if (s.l_name > 2 && T[s.name + s.l_name - 2] == '/' && isdigit((unsigned char)T[s.name + s.l_name - 1])) { s.l_name -= 2; T[s.name + s.l_name] = 0; } }

struct Batch { Arena text; std::vector<Seq> seqs; std::vector<std::shared_ptr<ParBlock>> blocks;   /* parsed blocks the records with a base of their own lie in */
	const char *T(const Seq &q) const { return q.base ? q.base : text.data(); } };

static bool read_batch(Source &r1, Source *r2, int chunk, Batch &out)
{
	static long batch_no = 0;
	const long id = batch_no++;
	out.seqs.clear(); out.text.clear(); out.blocks.clear();
	if (out.text.capacity() == 0) { out.text.reserve((size_t)chunk * 5 / 2 + (1 << 20)); out.seqs.reserve((size_t)chunk / 64 + 1024); }   // ~2.3 text bytes per base
	long size = 0; Seq s, s2;
	while (r1.read(s, out.text, out.blocks, id)) {
		if (r2 && !r2->read(s2, out.text, out.blocks, id)) { fprintf(stderr, "[W::%s] the 2nd file has fewer sequences.\n", "bseq_read"); break; }
		trim_readno(s, out.text); size += s.l_seq; out.seqs.push_back(s);
		if (r2) { trim_readno(s2, out.text); size += s2.l_seq; out.seqs.push_back(s2); }
		if (size >= chunk && (out.seqs.size() & 1) == 0) break;
	}
	return !out.seqs.empty();
}

struct Nt4 {   // nst_nt4_table (bntseq.c:46-63)
	uint8_t t[256];
	Nt4() { memset(t, 4, sizeof t); t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3; t['-'] = 5; }
};
static const Nt4 g_nt4;

// nst_nt4_table over a run of bases.  32 bytes per step where the CPU has AVX2: A/C/G/T in either case are found by comparing the byte
// with bit 5 cleared, '-' by itself, everything else is 4 -- the same values as the table (checked for all 256 bytes at start-up).
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void nt4_encode_avx2(const unsigned char *src, uint8_t *d, int n)
{
	const __m256i up = _mm256_set1_epi8((char)0xDF), four = _mm256_set1_epi8(4);
	const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T'), cD = _mm256_set1_epi8('-');
	int j = 0;
	for (; j + 32 <= n; j += 32) {
		const __m256i c = _mm256_loadu_si256((const __m256i*)(src + j)), u = _mm256_and_si256(c, up);
		// 4 minus: 4 for A, 3 for C, 2 for G, 1 for T (the compare masks are 0xFF = -1, ANDed down to the amount), plus 1 for '-'
		__m256i r = four;
		r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(u, cA), four));
		r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(u, cC), _mm256_set1_epi8(3)));
		r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(u, cG), _mm256_set1_epi8(2)));
		r = _mm256_sub_epi8(r, _mm256_and_si256(_mm256_cmpeq_epi8(u, cT), _mm256_set1_epi8(1)));
		r = _mm256_sub_epi8(r, _mm256_cmpeq_epi8(c, cD));                       // (minus -1)
		_mm256_storeu_si256((__m256i*)(d + j), r);
	}
	for (; j < n; ++j) d[j] = g_nt4.t[src[j]];
}
static bool nt4_avx2_ok()
{	// the vector form is used only if this CPU has AVX2 and the form reproduces the table for every byte value
	if (!__builtin_cpu_supports("avx2")) return false;
	unsigned char in[256]; uint8_t out[256];
	for (int i = 0; i < 256; ++i) in[i] = (unsigned char)i;
	nt4_encode_avx2(in, out, 256);
	for (int i = 0; i < 256; ++i) if (out[i] != g_nt4.t[i]) return false;
	return true;
}
static const bool g_nt4_avx2 = !getenv("BWAGPU_CLI_NO_AVX2") && nt4_avx2_ok();
#else
static const bool g_nt4_avx2 = false;
static void nt4_encode_avx2(const unsigned char*, uint8_t*, int) {}
#endif
static inline void nt4_encode(const unsigned char *src, uint8_t *d, int n)
{
	if (g_nt4_avx2) nt4_encode_avx2(src, d, n);
	else for (int j = 0; j < n; ++j) d[j] = g_nt4.t[src[j]];
}

// ---- batches flow through a four-stage pipeline: read+encode | device (hot path) | finalize (host threads) | write -------------
// (the reference overlaps input, compute and output the same way with kt_pipeline, kthread.c:119; here the compute step is
// split once more so that the GPU works on batch i+1 while the host cores turn batch i's regions into SAM text)
// base codes of a batch: page-locked (bwagpu_alloc_host) so that the upload is one DMA; kept from batch to batch
struct HostBuf {
	uint8_t *p = nullptr; size_t cap = 0;
	HostBuf() = default;
	HostBuf(const HostBuf&) = delete; HostBuf &operator=(const HostBuf&) = delete;
	HostBuf(HostBuf &&o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
	HostBuf &operator=(HostBuf &&o) noexcept { if (this != &o) { bwagpu_free(p); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
	~HostBuf() { bwagpu_free(p); }
	uint8_t *data() const { return p; }
	void need(size_t n) {
		if (n <= cap) return;
		bwagpu_free(p);
		cap = n + n / 8;
		p = (uint8_t*)bwagpu_alloc_host(cap);
		if (!p) { fprintf(stderr, "[E::%s] out of memory\n", "mem_process_seqs"); exit(EXIT_FAILURE); }
	}
};
struct Sub {      // one mem_process_seqs call (bwamem.c:1235-1264) on the reads `idx` of its batch
	std::vector<int> idx; bwagpu_opt_t opt; int64_t n_processed = 0;
	HostBuf flat; std::vector<int64_t> off; std::vector<int32_t> counts;
	bwagpu_alnreg_t *all = nullptr; int64_t tot = 0;
	bwagpu_cigar_t *cigs = nullptr;           // device-side global alignments of the regions (bwagpu_batch_cigars)
	uint32_t *cig_ops = nullptr;              // ... and the operation array its records with more than 6 operations point into
	bwagpu_matesw_t *msw = nullptr; int64_t n_msw = 0;   // device-side mate-rescue alignments (bwagpu_batch_matesw)
	Pestat pes[4]; bool have_pes = false;     // insert-size windows, when they had to be computed before the finalize stage
	double t_dev = 0;
};
struct Work { long no = 0; Batch in; std::vector<Sub> subs; std::vector<std::string> out; bool by_read = false; /* SAM text in output order: one string per chunk of reads, or (by_read, smart pairing) per read */ };
typedef std::unique_ptr<Work> WorkP;

struct Chan {     // bounded FIFO between two stages
	std::mutex m; std::condition_variable cv; std::deque<WorkP> q; size_t cap; bool closed = false;
	explicit Chan(size_t c) : cap(c) {}
	void push(WorkP w) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return q.size() < cap; }); q.push_back(std::move(w)); cv.notify_all(); }
	bool pop(WorkP &w) { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return !q.empty() || closed; }); if (q.empty()) return false; w = std::move(q.front()); q.pop_front(); cv.notify_all(); return true; }
	void close() { std::lock_guard<std::mutex> l(m); closed = true; cv.notify_all(); }
};

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// CPU seconds of the calling thread / of the whole process so far (the -v 3 summary: what the host stages cost per read, as opposed to how long they were busy)
static double thread_cpu_s() { struct timespec ts; return clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec : 0.; }
static double process_cpu_s() { struct timespec ts; return clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts) == 0 ? (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec : 0.; }

static void encode_sub(const Batch &in, Sub &u)
{
	const int n = (int)u.idx.size();
	u.off.assign((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) u.off[i + 1] = u.off[i] + (int64_t)in.seqs[u.idx[i]].l_seq;
	u.flat.need((size_t)u.off[n] + 1);
	parallel_for(u.opt.n_threads < 4 ? u.opt.n_threads : 4, n, [&](long i) {
		const Seq &q = in.seqs[u.idx[i]];
		const unsigned char *src = (const unsigned char*)in.T(q) + q.seq; uint8_t *d = u.flat.data() + u.off[i];
		nt4_encode(src, d, q.l_seq);
	});
	u.counts.assign((size_t)n, 0);
}

// stage 2: the kt_for(worker1) of mem_process_seqs (bwamem.c:1252) on the device
static std::mutex g_dev_mutex;
static bool g_dev_serialize = false;   // BWAGPU_CLI_SERIALIZE=1: one device call at a time (the mock HIP runtime of the CPU tests is not thread-safe)

// (_exit: this runs on a device thread while sibling threads may be inside HIP calls -- no atexit handlers / static destructors under them)
static void device_fail(bwagpu_t *gpu, int rc, const char *what = nullptr)
{
	if (rc == BWAGPU_OK) fprintf(stderr, "[E::%s] device results inconsistent: %s\n", "mem_process_seqs", what ? what : "record count mismatch");   // (a call that returned BWAGPU_OK with the wrong number of records)
	else fprintf(stderr, "[E::%s] %s: %s\n", "mem_process_seqs", bwagpu_strerror(rc), bwagpu_last_error(gpu));
	fflush(stderr); fflush(stdout); _exit(EXIT_FAILURE);
}

// One mem_process_seqs call's device work on the GPUs of `gpus` (one handle per device).  With several devices
// the reads are split into contiguous ranges of whole pairs, every device runs the hot path -- and the device-side CIGARs and
// mate-rescue alignments -- on its range, and the results are concatenated in read order.  The one step that needs the whole
// batch, mem_pestat (bwamem.c:1258), runs on the host over the gathered regions between the two device phases, so the SAM is the
// single-device SAM whatever the number of devices.
static void device_sub(const std::vector<bwagpu_t*> &gpus, Sub &u, const RefSeqs &ref, const Pestat *pes0)
{
	std::unique_lock<std::mutex> serial(g_dev_mutex, std::defer_lock);
	if (g_dev_serialize) serial.lock();
	const double t0 = now_s();
	const int n = (int)u.idx.size();
	const bool trace = getenv("BWAGPU_CLI_TRACE") != nullptr;
	const bool pe = (u.opt.flag & F_PE) != 0;
	int D = (int)gpus.size();
	const int units = pe ? n / 2 : n, per = pe ? 2 : 1;
	if (D > units) D = units > 0 ? units : 1;
	struct Shard { int lo = 0, hi = 0; std::vector<int64_t> off; bwagpu_alnreg_t *all = nullptr; int64_t tot = 0; bwagpu_cigar_t *cigs = nullptr; uint32_t *ops = nullptr; int64_t n_ops = 0; bwagpu_matesw_t *msw = nullptr; int64_t n_msw = 0; };
	std::vector<Shard> sh((size_t)D);
	for (int d = 0; d < D; ++d) {
		sh[d].lo = (int)((int64_t)units * d / D) * per; sh[d].hi = d + 1 == D ? n : (int)((int64_t)units * (d + 1) / D) * per;
		sh[d].off.resize((size_t)(sh[d].hi - sh[d].lo) + 1);
		for (int i = sh[d].lo; i <= sh[d].hi; ++i) sh[d].off[(size_t)(i - sh[d].lo)] = u.off[i] - u.off[sh[d].lo];
	}
	auto on_devices = [&](const std::function<void(int)> &f) {   // f(d) for every device, concurrently (the calls block on their streams)
		if (D == 1 || g_dev_serialize) { for (int d = 0; d < D; ++d) f(d); return; }
		std::vector<std::thread> th;
		for (int d = 1; d < D; ++d) th.emplace_back(f, d);
		f(0);
		for (auto &t : th) t.join();
	};
	double t1 = t0, t2 = t0, t3 = t0, t4 = t0, t5 = t0;
	on_devices([&](int d) {
		Shard &s = sh[d];
		int rc = bwagpu_batch_upload(gpus[d], s.hi - s.lo, u.flat.data() + u.off[s.lo], s.off.data());
		if (d == 0) t1 = now_s();
		if (rc == BWAGPU_OK) rc = bwagpu_batch_run(gpus[d], &u.opt);
		if (d == 0) t2 = now_s();
		if (d == 0 && trace && rc == BWAGPU_OK) {      // the hot path's stages by the library's own HIP events
			bwagpu_stats_t st;
			if (bwagpu_get_stats(gpus[d], &st) == BWAGPU_OK)
				fprintf(stderr, "[D::device_sub] stage ms: seed %.2f publish %.2f sa %.2f chain %.2f seedsw %.2f extend %.2f dedup %.2f total %.2f (retries %d, overflow mask 0x%x)\n",
						st.ms_seed, st.ms_publish, st.ms_sa, st.ms_chain, st.ms_seedsw, st.ms_extend, st.ms_dedup, st.ms_total, st.n_retries, (unsigned)st.retry_mask);
		}
		if (rc == BWAGPU_OK) rc = bwagpu_batch_download(gpus[d], u.counts.data() + s.lo, &s.all, &s.tot);
		if (rc != BWAGPU_OK) device_fail(gpus[d], rc);
	});
	if (D == 1) { u.all = sh[0].all; u.tot = sh[0].tot; }
	else {   // gather the regions in read order
		u.tot = 0; for (auto &s : sh) u.tot += s.tot;
		u.all = (bwagpu_alnreg_t*)malloc((size_t)(u.tot ? u.tot : 1) * sizeof(bwagpu_alnreg_t));
		if (!u.all) { fprintf(stderr, "[E::%s] out of memory\n", "mem_process_seqs"); exit(EXIT_FAILURE); }
		int64_t k = 0;
		for (auto &s : sh) {
			if (s.tot) memcpy(u.all + k, s.all, (size_t)s.tot * sizeof(bwagpu_alnreg_t));
			k += s.tot;
			bwagpu_free(s.all); s.all = nullptr;
		}
	}
	// mem_pestat needs the whole batch's regions (bwamem.c:1258) and nothing else: it runs on host threads while the devices
	// produce the CIGARs of the same regions (it used to wait for them: 26 of a batch's 377 ms in this stage)
	const bool want_matesw = g_device_matesw && pe && !(u.opt.flag & F_NO_RESCUE) && u.tot > 0;
	double t_pes = 0;
	std::thread pes_thread;
	if (want_matesw) {
		if (pes0) memcpy(u.pes, pes0, sizeof u.pes);
		else pes_thread = std::thread([&] {
			const double tp = now_s();
			std::vector<int64_t> roff((size_t)n + 1, 0);
			for (int i = 0; i < n; ++i) roff[i + 1] = roff[i] + u.counts[i];
			pestat_flat(u.opt, ref.l_pac, n, u.all, roff.data(), u.pes, g_verbose >= 3, u.opt.n_threads < 4 ? u.opt.n_threads : 4);
			t_pes = now_s() - tp;
		});
		u.have_pes = true;
	}
	const bool have_cigs = g_device_cigars && u.tot > 0;
	if (have_cigs) on_devices([&](int d) {       // the DP of mem_reg2aln, NM and MD on the device as well; the host keeps the text
		Shard &s = sh[d];
		if (s.tot == 0) return;
		int64_t nc = 0;
		int rc = bwagpu_batch_cigars(gpus[d], &u.opt, &s.cigs, &nc);
		if (rc != BWAGPU_OK || nc != s.tot) device_fail(gpus[d], rc, "bwagpu_batch_cigars returned another number of records than bwagpu_batch_download");
		rc = bwagpu_batch_cigar_ops(gpus[d], &s.ops, &s.n_ops);
		if (rc != BWAGPU_OK) device_fail(gpus[d], rc);
	});
	t3 = now_s();
	if (D == 1) { u.cigs = sh[0].cigs; u.cig_ops = sh[0].ops; }
	else if (have_cigs) {
		u.cigs = (bwagpu_cigar_t*)malloc((size_t)u.tot * sizeof(bwagpu_cigar_t));
		int64_t n_ops = 0; for (auto &s : sh) n_ops += s.n_ops;
		u.cig_ops = (uint32_t*)malloc((size_t)(n_ops ? n_ops : 1) * 4);
		if (!u.cigs || !u.cig_ops) { fprintf(stderr, "[E::%s] out of memory\n", "mem_process_seqs"); exit(EXIT_FAILURE); }
		int64_t k = 0, ko = 0;
		for (auto &s : sh) {
			if (s.tot) {   // (a shard without regions has no records); offsets into the operation array move with the shard's part of it
				memcpy(u.cigs + k, s.cigs, (size_t)s.tot * sizeof(bwagpu_cigar_t));
				if (s.n_ops) memcpy(u.cig_ops + ko, s.ops, (size_t)s.n_ops * 4);
				if (ko) for (int64_t i = 0; i < s.tot; ++i) {
					bwagpu_cigar_t &c = u.cigs[k + i];
					if (c.n_cigar > 6) { const uint64_t at = ((uint64_t)c.cigar[1] << 32 | c.cigar[0]) + (uint64_t)ko; c.cigar[0] = (uint32_t)at; c.cigar[1] = (uint32_t)(at >> 32); }
					if (c.n_cigar >= 0 && c.md_len > 8) c.md += (uint64_t)ko;
				}
			}
			k += s.tot; ko += s.n_ops;
			bwagpu_free(s.cigs); s.cigs = nullptr; bwagpu_free(s.ops); s.ops = nullptr;
		}
	}
	t4 = now_s();
	if (pes_thread.joinable()) pes_thread.join();
	if (want_matesw) {
		bwagpu_pes_t dp[4];
		for (int d = 0; d < 4; ++d) { dp[d].low = u.pes[d].low; dp[d].high = u.pes[d].high; dp[d].failed = u.pes[d].failed; dp[d].pad_ = 0; }
		on_devices([&](int d) {
			if (sh[d].tot == 0) return;
			int rc = bwagpu_batch_matesw(gpus[d], &u.opt, dp, &sh[d].msw, &sh[d].n_msw);
			if (rc != BWAGPU_OK) device_fail(gpus[d], rc);
		});
		if (D == 1) { u.msw = sh[0].msw; u.n_msw = sh[0].n_msw; }
		else {
			u.n_msw = 0; for (auto &s : sh) u.n_msw += s.n_msw;
			u.msw = (bwagpu_matesw_t*)malloc((size_t)(u.n_msw ? u.n_msw : 1) * sizeof(bwagpu_matesw_t));
			if (!u.msw) { fprintf(stderr, "[E::%s] out of memory\n", "mem_process_seqs"); exit(EXIT_FAILURE); }
			int64_t k = 0;
			for (auto &s : sh) {
				for (int64_t i = 0; i < s.n_msw; ++i) { u.msw[k] = s.msw[i]; u.msw[k].read += s.lo; ++k; }   // read indices are relative to the device's range
				bwagpu_free(s.msw); s.msw = nullptr;
			}
		}
	}
	t5 = now_s();
	u.t_dev = t5 - t0;
	if (trace) fprintf(stderr, "[D::device_sub] %d reads on %d device(s) -> %ld regions (flag 0x%x): upload %.3f run %.3f download+cigars %.3f pestat+matesw %.3f s (pestat %.3f, %ld mate-rescue alignments)\n", n, D, (long)u.tot, u.opt.flag,
					   t1 - t0, t2 - t1, t3 - t2, t5 - t4, t_pes, (long)u.n_msw);
	if (trace) {       // the part after the hot path by the library's own HIP events (device 0's share)
		bwagpu_stats_t st;
		if (bwagpu_get_stats(gpus[0], &st) == BWAGPU_OK)
			fprintf(stderr, "[D::device_sub] after the hot path, ms: pack %.2f download copy %.2f cigar kernels %.2f cigar copies %.2f\n", st.ms_pack, st.ms_download_copy, st.ms_cigar_kernels, st.ms_cigar_copy);
	}
}

// stage 3: mem_pestat + kt_for(worker2) (bwamem.c:1254-1260) on the host cores
static void finalize_sub(const RefSeqs &ref, Work &w, Sub &u, const Pestat *pes0, const char *rg_id, bool copy_comment)
{
	const double t0 = now_s();
	const int n = (int)u.idx.size();
	std::vector<Read> reads((size_t)n); std::vector<CigHints> hints(u.cigs ? (size_t)n : 0);
	std::vector<int64_t> roff((size_t)n + 1, 0);
	for (int i = 0; i < n; ++i) roff[i + 1] = roff[i] + u.counts[i];
	parallel_for(u.opt.n_threads, n, [&](long i) {
		const Seq &q = w.in.seqs[u.idx[i]];
		const char *T = w.in.T(q);
		if (u.cigs) { hints[i].regs = u.all + roff[i]; hints[i].cigs = u.cigs + roff[i]; hints[i].n = u.counts[i]; hints[i].ops = u.cig_ops; reads[i].hints = &hints[i]; }
		reads[i].name = T + q.name;
		reads[i].comment = copy_comment && q.has_comment ? T + q.comment : nullptr;
		reads[i].seq = u.flat.data() + u.off[i]; reads[i].qual = q.has_qual ? T + q.qual : nullptr; reads[i].l_seq = q.l_seq;
	});
	if (u.opt.flag & F_PE) for (int i = 0; i + 1 < n; i += 2) if (strcmp(reads[i].name, reads[i + 1].name) != 0) { fprintf(stderr, "[mem_sam_pe] paired reads have different names: \"%s\", \"%s\"\n", reads[i].name, reads[i + 1].name); exit(EXIT_FAILURE); }
	std::vector<bwagpu_matesw_t> msw_sorted;
	if (u.msw) attach_matesw(n, reads.data(), u.msw, u.n_msw, msw_sorted);
	const Pestat *pes = (u.opt.flag & F_PE) ? (u.have_pes ? u.pes : pes0) : nullptr;
	if (!w.by_read)      // the batch is one mem_process_seqs call (always, without -p): reads are in output order, text comes by the chunk
		finalize_batch_chunks(u.opt, ref, u.n_processed, n, reads.data(), u.all, roff.data(), pes, u.opt.n_threads, rg_id, 64, w.out, g_verbose >= 3);
	else {
		std::vector<std::string> sam;
		finalize_batch(u.opt, ref, u.n_processed, n, reads.data(), u.all, roff.data(), pes, u.opt.n_threads, rg_id, sam, g_verbose >= 3);
		for (int i = 0; i < n; ++i) w.out[u.idx[i]].swap(sam[i]);
	}
	bwagpu_free(u.all); u.all = nullptr; bwagpu_free(u.cigs); u.cigs = nullptr; bwagpu_free(u.cig_ops); u.cig_ops = nullptr; bwagpu_free(u.msw); u.msw = nullptr;
	if (g_verbose >= 3) fprintf(stderr, "[M::%s] Processed %d reads in %.3f real sec\n", "mem_process_seqs", n, u.t_dev + (now_s() - t0));
}

static bool slurp(const std::string &fn, std::vector<char> &out)
{
	FILE *fp = fopen(fn.c_str(), "rb");
	if (!fp) return false;
	fseek(fp, 0, SEEK_END); long sz = ftell(fp); fseek(fp, 0, SEEK_SET);
	out.resize((size_t)sz);
	bool ok = fread(out.data(), 1, (size_t)sz, fp) == (size_t)sz;
	fclose(fp);
	return ok;
}

static int usage()
{
	fprintf(stderr, "\nUsage: bwa-amd mem [options] <idxbase> <in1.fq> [in2.fq]\n\n"
			"Options are those of `bwa mem` (0.7.19): -t -k -w -d -r -y -c -D -W -m -S -P -A -B -O -E -L -U -x -p -R -H -j -5 -q -K -v -T -h -z -a -C -V -Y -M -u -I -N -G -s -X -Q\n"
			"The FM-index seeding, chaining and seed extension of every read run on the GPU (libbwagpu.so); output equals bwa mem's.\n\n");
	return 1;
}

int main(int argc, char *argv[])
{
	if (argc < 2 || strcmp(argv[1], "mem") != 0) return usage();
	std::string pg = "@PG\tID:bwa-amd\tPN:bwa-amd\tVN:0.1\tCL:" + std::string(argv[0]);
	for (int i = 1; i < argc; ++i) { pg += ' '; pg += argv[i]; }
	--argc; ++argv;
	bwagpu_opt_t opt, opt0; opt_init(&opt); memset(&opt0, 0, sizeof opt0);
	Pestat pes[4]; memset(pes, 0, sizeof pes); for (int i = 0; i < 4; ++i) pes[i].failed = 1;
	const Pestat *pes0 = nullptr;
	const char *mode = nullptr; char *p;
	int c, fixed_chunk = -1, ignore_alt = 0, copy_comment = 0, device = getenv("BWAGPU_DEVICE") ? atoi(getenv("BWAGPU_DEVICE")) : 0;
	std::string hdr_line, rg_line, rg_id;
	while ((c = getopt(argc, argv, "51qpaMCSPVYjuk:c:v:s:r:t:R:A:B:O:E:U:w:L:d:T:Q:D:m:I:N:o:f:W:x:G:h:y:K:X:H:z:")) >= 0) {
		if (c == 'k') opt.min_seed_len = atoi(optarg), opt0.min_seed_len = 1;
		else if (c == '1') {}
		else if (c == 'x') mode = optarg;
		else if (c == 'w') opt.w = atoi(optarg), opt0.w = 1;
		else if (c == 'A') opt.a = atoi(optarg), opt0.a = 1;
		else if (c == 'B') opt.b = atoi(optarg), opt0.b = 1;
		else if (c == 'T') opt.T = atoi(optarg), opt0.T = 1;
		else if (c == 'U') opt.pen_unpaired = atoi(optarg), opt0.pen_unpaired = 1;
		else if (c == 't') opt.n_threads = atoi(optarg), opt.n_threads = opt.n_threads > 1 ? opt.n_threads : 1;
		else if (c == 'P') opt.flag |= F_NOPAIRING;
		else if (c == 'a') opt.flag |= F_ALL;
		else if (c == 'p') opt.flag |= F_PE | F_SMARTPE;
		else if (c == 'M') opt.flag |= F_NO_MULTI;
		else if (c == 'S') opt.flag |= F_NO_RESCUE;
		else if (c == 'Y') opt.flag |= F_SOFTCLIP;
		else if (c == 'V') opt.flag |= F_REF_HDR;
		else if (c == '5') opt.flag |= F_PRIMARY5 | F_KEEP_SUPP_MAPQ;
		else if (c == 'q') opt.flag |= F_KEEP_SUPP_MAPQ;
		else if (c == 'u') opt.flag |= F_XB;
		else if (c == 'c') opt.max_occ = atoi(optarg), opt0.max_occ = 1;
		else if (c == 'd') opt.zdrop = atoi(optarg), opt0.zdrop = 1;
		else if (c == 'v') g_verbose = atoi(optarg);
		else if (c == 'j') ignore_alt = 1;
		else if (c == 'r') opt.split_factor = (float)atof(optarg), opt0.split_factor = 1.f;
		else if (c == 'D') opt.drop_ratio = (float)atof(optarg), opt0.drop_ratio = 1.f;
		else if (c == 'm') opt.max_matesw = atoi(optarg), opt0.max_matesw = 1;
		else if (c == 's') opt.split_width = atoi(optarg), opt0.split_width = 1;
		else if (c == 'G') opt.max_chain_gap = atoi(optarg), opt0.max_chain_gap = 1;
		else if (c == 'N') opt.max_chain_extend = atoi(optarg), opt0.max_chain_extend = 1;
		else if (c == 'W') opt.min_chain_weight = atoi(optarg), opt0.min_chain_weight = 1;
		else if (c == 'y') opt.max_mem_intv = (uint64_t)atol(optarg), opt0.max_mem_intv = 1;
		else if (c == 'C') copy_comment = 1;
		else if (c == 'K') fixed_chunk = atoi(optarg);
		else if (c == 'X') opt.mask_level = (float)atof(optarg);
		else if (c == 'h') {
			opt0.max_XA_hits = opt0.max_XA_hits_alt = 1;
			opt.max_XA_hits = opt.max_XA_hits_alt = (int)strtol(optarg, &p, 10);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.max_XA_hits_alt = (int)strtol(p + 1, &p, 10);
		}
		else if (c == 'z') opt.XA_drop_ratio = (float)atof(optarg);
		else if (c == 'Q') { opt0.mapQ_coef_len = 1; opt.mapQ_coef_len = (float)atoi(optarg); opt.mapQ_coef_fac = opt.mapQ_coef_len > 0 ? (int)log(opt.mapQ_coef_len) : 0; }
		else if (c == 'O') { opt0.o_del = opt0.o_ins = 1; opt.o_del = opt.o_ins = (int)strtol(optarg, &p, 10); if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.o_ins = (int)strtol(p + 1, &p, 10); }
		else if (c == 'E') { opt0.e_del = opt0.e_ins = 1; opt.e_del = opt.e_ins = (int)strtol(optarg, &p, 10); if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.e_ins = (int)strtol(p + 1, &p, 10); }
		else if (c == 'L') { opt0.pen_clip5 = opt0.pen_clip3 = 1; opt.pen_clip5 = opt.pen_clip3 = (int)strtol(optarg, &p, 10); if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) opt.pen_clip3 = (int)strtol(p + 1, &p, 10); }
		else if (c == 'R') {   // bwa_set_rg (bwa.c:457-488)
			if (strstr(optarg, "@RG") != optarg) { fprintf(stderr, "[E::bwa_set_rg] the read group line is not started with @RG\n"); return 1; }
			if (strchr(optarg, '\t')) { fprintf(stderr, "[E::bwa_set_rg] the read group line contained literal <tab> characters -- replace with escaped tabs: \\t\n"); return 1; }
			rg_line = unescape(optarg);
			size_t q = rg_line.find("\tID:");
			if (q == std::string::npos) { fprintf(stderr, "[E::bwa_set_rg] no ID within the read group line\n"); return 1; }
			for (q += 4; q < rg_line.size() && rg_line[q] != '\t' && rg_line[q] != '\n'; ++q) rg_id += rg_line[q];
		}
		else if (c == 'H') {   // header lines given literally or in a file (fastmap.c:224-239; bwa_insert_header, bwa.c:490-505)
			auto add = [&](const char *line) { if (line[0] == '@') { if (!hdr_line.empty()) hdr_line += '\n'; hdr_line += unescape(line); } };
			if (optarg[0] == '@') add(optarg);
			else if (FILE *fp = fopen(optarg, "r")) {
				std::string line; int ch;
				while ((ch = fgetc(fp)) != EOF) { if (ch == '\n') { add(line.c_str()); line.clear(); } else line += (char)ch; }
				fclose(fp);
			}
		}
		else if (c == 'o' || c == 'f') { if (!freopen(optarg, "wb", stdout)) { fprintf(stderr, "[E::%s] fail to open '%s' for writing\n", "main_mem", optarg); return 1; } }
		else if (c == 'I') {
			pes0 = pes; pes[1].failed = 0; pes[1].avg = strtod(optarg, &p); pes[1].std = pes[1].avg * .1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes[1].std = strtod(p + 1, &p);
			pes[1].high = (int)(pes[1].avg + 4. * pes[1].std + .499); pes[1].low = (int)(pes[1].avg - 4. * pes[1].std + .499);
			if (pes[1].low < 1) pes[1].low = 1;
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes[1].high = (int)(strtod(p + 1, &p) + .499);
			if (*p != 0 && ispunct((unsigned char)*p) && isdigit((unsigned char)p[1])) pes[1].low = (int)(strtod(p + 1, &p) + .499);
		}
		else return 1;
	}
	if (!rg_line.empty()) { if (!hdr_line.empty()) hdr_line += '\n'; hdr_line += rg_line; }
	if (opt.n_threads < 1) opt.n_threads = 1;
	if (optind + 1 >= argc || optind + 3 < argc) return usage();
	if (mode) {   // presets (fastmap.c:330-358)
		if (strcmp(mode, "intractg") == 0) {
			if (!opt0.o_del) opt.o_del = 16; if (!opt0.o_ins) opt.o_ins = 16; if (!opt0.b) opt.b = 9;
			if (!opt0.pen_clip5) opt.pen_clip5 = 5; if (!opt0.pen_clip3) opt.pen_clip3 = 5;
		} else if (strcmp(mode, "pacbio") == 0 || strcmp(mode, "pbref") == 0 || strcmp(mode, "ont2d") == 0) {
			if (!opt0.o_del) opt.o_del = 1; if (!opt0.e_del) opt.e_del = 1; if (!opt0.o_ins) opt.o_ins = 1; if (!opt0.e_ins) opt.e_ins = 1;
			if (!opt0.b) opt.b = 1; if (opt0.split_factor == 0.) opt.split_factor = 10.;
			const bool ont = strcmp(mode, "ont2d") == 0;
			if (!opt0.min_chain_weight) opt.min_chain_weight = ont ? 20 : 40;
			if (!opt0.min_seed_len) opt.min_seed_len = ont ? 14 : 17;
			if (!opt0.pen_clip5) opt.pen_clip5 = 0; if (!opt0.pen_clip3) opt.pen_clip3 = 0;
		} else { fprintf(stderr, "[E::%s] unknown read type '%s'\n", "main_mem", mode); return 1; }
	} else if (opt0.a) {   // update_a (fastmap.c:125-139)
		if (!opt0.b) opt.b *= opt.a; if (!opt0.T) opt.T *= opt.a; if (!opt0.o_del) opt.o_del *= opt.a; if (!opt0.e_del) opt.e_del *= opt.a;
		if (!opt0.o_ins) opt.o_ins *= opt.a; if (!opt0.e_ins) opt.e_ins *= opt.a; if (!opt0.zdrop) opt.zdrop *= opt.a;
		if (!opt0.pen_clip5) opt.pen_clip5 *= opt.a; if (!opt0.pen_clip3) opt.pen_clip3 *= opt.a; if (!opt0.pen_unpaired) opt.pen_unpaired *= opt.a;
	}
	fill_scmat(opt.a, opt.b, opt.mat);

	// ---- index: host copy for the finalize code, device copy for the hot path ----
	const std::string prefix = argv[optind];
	RefSeqs ref; std::string err;
	// While the index is read and uploaded (seconds of file and device work on this thread), a helper thread page-locks the base arrays of the first batches
	// and touches the parse blocks' text arenas: the pipeline's first batches otherwise pay for them one after the other on the two threads that pace its
	// fill -- the encoder (hipHostMalloc of 100 MB: batch 0 took 145-180 ms there against 30-50 in the steady state) and the reader (90-130 ms against 25-45).
	struct Prewarm { std::vector<HostBuf> flats; std::thread th; ~Prewarm() { if (th.joinable()) th.join(); } } prewarm;
	if (!getenv("BWAGPU_CLI_NO_PREWARM")) {
		const int64_t chunk0 = fixed_chunk > 0 ? fixed_chunk : (int64_t)opt.chunk_size * opt.n_threads;
		const int n_flat = getenv("BWAGPU_CLI_STREAMS") ? (atoi(getenv("BWAGPU_CLI_STREAMS")) > 0 ? atoi(getenv("BWAGPU_CLI_STREAMS")) + 1 : 2) : 4;
		const bool par_blocks = !(getenv("BWAGPU_CLI_PARSE_THREADS") && atoi(getenv("BWAGPU_CLI_PARSE_THREADS")) <= 0);
		prewarm.th = std::thread([&prewarm, chunk0, n_flat, par_blocks] {
			if (chunk0 > 0 && chunk0 <= ((int64_t)1 << 31)) for (int k = 0; k < n_flat && k < 6; ++k) { HostBuf b; b.need((size_t)chunk0 + (1 << 20)); prewarm.flats.push_back(std::move(b)); }
			if (par_blocks) {
				std::vector<std::shared_ptr<ParBlock>> hold;
				for (int k = 0; k < 40; ++k) { hold.push_back(new_block()); Arena &t = hold.back()->text; t.reserve(((size_t)4 << 20) + 1024); t.resize(t.capacity()); for (size_t o = 0; o < t.size(); o += 4096) t[o] = 1; t.clear(); hold.back()->seqs.reserve(20000); }
			}      // (the blocks go back to the pool new_block() draws from)
		});
	}
	if (!load_refseqs(prefix, ref, err)) { fprintf(stderr, "[E::%s] fail to locate the index files: %s\n", "main_mem", err.c_str()); return 1; }
	if (ignore_alt) for (auto &ctg : ref.ctg) ctg.is_alt = 0;
	bwagpu_t *gpu = nullptr;
	{
		std::vector<char> fb, fs;
		if (!slurp(prefix + ".bwt", fb) || fb.size() < 40 || !slurp(prefix + ".sa", fs) || fs.size() < 56) { fprintf(stderr, "[E::%s] fail to read %s.bwt/.sa\n", "main_mem", prefix.c_str()); return 1; }
		bwagpu_index_desc_t d; memset(&d, 0, sizeof d);
		const uint64_t *hb = (const uint64_t*)fb.data(), *hs = (const uint64_t*)fs.data();
		d.primary = hb[0]; for (int i = 0; i < 4; ++i) d.L2[i + 1] = hb[1 + i]; d.seq_len = d.L2[4];
		d.bwt = (const uint32_t*)(fb.data() + 40); d.bwt_size = (fb.size() - 40) / 4;
		d.sa_intv = (int)hs[5]; d.n_sa = (d.seq_len + d.sa_intv) / d.sa_intv;
		if (hs[0] != d.primary || hs[6] != d.seq_len || fs.size() < 56 + (d.n_sa - 1) * 8) { fprintf(stderr, "[E::%s] SA-BWT inconsistency\n", "main_mem"); return 1; }
		std::vector<uint64_t> sa(d.n_sa); sa[0] = (uint64_t)-1; memcpy(sa.data() + 1, fs.data() + 56, (d.n_sa - 1) * 8);
		d.sa = sa.data(); d.pac = ref.pac.data(); d.l_pac = ref.l_pac; d.n_seqs = (int)ref.ctg.size();
		std::vector<int64_t> off(d.n_seqs); std::vector<int32_t> len(d.n_seqs), alt(d.n_seqs);
		for (int i = 0; i < d.n_seqs; ++i) { off[i] = ref.ctg[i].offset; len[i] = ref.ctg[i].len; alt[i] = ref.ctg[i].is_alt; }
		d.ctg_offset = off.data(); d.ctg_len = len.data(); d.ctg_is_alt = alt.data();
		int rc = bwagpu_create(&gpu, &d, device);
		if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] %s\n", "main_mem", bwagpu_strerror(rc)); return 1; }
	}
	bwagpu_set_taps(gpu, 0);
	bwagpu_set_cigar_filter(gpu, getenv("BWAGPU_CLI_CIGAR_FILTER") ? atoi(getenv("BWAGPU_CLI_CIGAR_FILTER")) : 1);   // (clones inherit it)
	{	// SA look-ups walk ~31 LF steps with the reference's interval of 32; HBM has room for the full array (same values): 8 bytes per text position, 50 GB for a
		// human genome, one look-up = one 8-byte read (k_sa 3.1 -> 0.9 ms per million reads against an interval of 4).  If that much is not to be had: the next intervals up.
		// The array must leave room for the batches: every handle (BWAGPU_CLI_STREAMS of them per device) grows arenas for a -K batch -- about 17 GB at the
		// default -K with 150 bp reads -- and a device with less free memory than an idle MI355X (smaller parts, several processes per GPU) could take the
		// full array and then fail its first batch.  So: the smallest interval whose array fits beside the handles' estimated footprint (the library's own
		// allocation arithmetic run dry, bwagpu_batch_footprint, for reads of >= 100 bp -- shorter reads mean more reads per -K batch; + 15 % + 2 GB for the
		// CIGAR / mate-rescue scratch that is sized by results).  BWAGPU_CLI_HBM_LIMIT_MB caps what the program takes the free memory to be.
		const int dense = getenv("BWAGPU_CLI_DENSE_SA") ? atoi(getenv("BWAGPU_CLI_DENSE_SA")) : 1;
		const int n_slots = getenv("BWAGPU_CLI_STREAMS") ? (atoi(getenv("BWAGPU_CLI_STREAMS")) > 1 ? atoi(getenv("BWAGPU_CLI_STREAMS")) : 1) : 3;
		const int64_t chunk_est = fixed_chunk > 0 ? fixed_chunk : (int64_t)opt.chunk_size * opt.n_threads;
		const bool long_est = mode && (strcmp(mode, "pacbio") == 0 || strcmp(mode, "pbref") == 0 || strcmp(mode, "ont2d") == 0);
		const int len_est = long_est ? 10000 : 100;
		const int64_t per_handle = bwagpu_batch_footprint(gpu, (int)(chunk_est / len_est) + 1024, chunk_est + (1 << 20), long_est ? 20000 : 256);
		uint64_t free_b = 0, total_b = 0, seq_len = 0;
		bwagpu_mem_info(gpu, &free_b, &total_b);
		bwagpu_index_info(gpu, nullptr, nullptr, &seq_len, nullptr);
		if (getenv("BWAGPU_CLI_HBM_LIMIT_MB")) { const uint64_t lim = (uint64_t)atoll(getenv("BWAGPU_CLI_HBM_LIMIT_MB")) << 20; if (lim < free_b) free_b = lim; }
		const double need = per_handle > 0 ? (double)per_handle * n_slots * 1.15 + 2e9 : 0.0;
		int chosen = 0;
		for (int dn = dense; dn > 0 && dn < 32; dn *= 2) {
			const double sa_bytes = 8.0 * ((double)seq_len / dn + 1);
			if (free_b > 0 && sa_bytes + need > (double)free_b) {
				if (g_verbose >= 3) fprintf(stderr, "[M::%s] SA interval %d needs %.1f GB beside %.1f GB for %d handles' batches; %.1f GB are free: trying the next interval\n", "main_mem", dn, sa_bytes / 1e9, need / 1e9, n_slots, free_b / 1e9);
				continue;
			}
			const int rc = bwagpu_densify_sa(gpu, dn);
			if (rc == BWAGPU_OK) { chosen = dn; break; }
			if (g_verbose >= 2) fprintf(stderr, "[W::%s] SA not densified to an interval of %d: %s\n", "main_mem", dn, bwagpu_strerror(rc));
			if (rc != BWAGPU_ENOMEM) break;
		}
		if (g_verbose >= 3) {
			if (chosen) fprintf(stderr, "[M::%s] suffix array expanded on the device to interval %d (%.1f GB; %.1f GB free before, %.1f GB set aside for %d handles' batches)\n", "main_mem", chosen, 8.0 * ((double)seq_len / chosen + 1) / 1e9, free_b / 1e9, need / 1e9, n_slots);
			else if (dense > 0 && dense < 32) fprintf(stderr, "[M::%s] suffix array kept at the index's own interval (no denser one fits beside the batches)\n", "main_mem");
		}
	}

	// input: BWAGPU_CLI_PARSE_THREADS parser threads for plain FASTQ files (0: the streaming reader alone).  Default 4 -- since round 5 for one device as well:
	// its device stage lets a 667 k-read batch go every 79 ms and the streaming reader delivers one every 71-77, so every hiccup of the reader was the device's;
	// with the blocks parsed by four threads the reader holds a batch 39-55 ms (6 M pairs: 5.67 / 5.83 -> 6.26 Mreads/s, profiles/r05_e2e_reserve_results.log)
	bool any_bgzf = false;
	for (int k = optind + 1; k < argc && k <= optind + 2; ++k) if (strcmp(argv[k], "-") && BgzfPipe::is_bgzf(argv[k])) any_bgzf = true;
	// (BGZF input: the pool inflates -- ~210 MB/s of FASTQ per thread --, so it gets up to eight threads, half of -t)
	const int n_parse = getenv("BWAGPU_CLI_PARSE_THREADS") ? atoi(getenv("BWAGPU_CLI_PARSE_THREADS")) : (any_bgzf ? (opt.n_threads / 2 > 8 ? 8 : (opt.n_threads / 2 < 4 ? 4 : opt.n_threads / 2)) : 4);
	std::unique_ptr<ParPool> parse_pool(n_parse > 0 ? new ParPool(n_parse) : nullptr);
	Source r1, r2; Source *pr2 = nullptr;
	if (!r1.open(argv[optind + 1], parse_pool.get())) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", "main_mem", argv[optind + 1]); return 1; }
	if (optind + 2 < argc) {
		if (opt.flag & F_PE) { if (g_verbose >= 2) fprintf(stderr, "[W::%s] when '-p' is in use, the second query file is ignored.\n", "main_mem"); }
		else { if (!r2.open(argv[optind + 2], parse_pool.get())) { fprintf(stderr, "[E::%s] fail to open file `%s'.\n", "main_mem", argv[optind + 2]); return 1; } pr2 = &r2; opt.flag |= F_PE; }
	}
	// SAM header (bwa_print_sam_hdr, bwa.c:407-439)
	{
		bool has_hd = hdr_line.rfind("@HD\t", 0) == 0 || hdr_line.find("\n@HD\t") != std::string::npos;
		bool has_sq = hdr_line.rfind("@SQ\t", 0) == 0 || hdr_line.find("\n@SQ\t") != std::string::npos;
		if (!has_hd) fputs("@HD\tVN:1.5\tSO:unsorted\tGO:query\n", stdout);
		if (!has_sq) for (auto &ctg : ref.ctg) { printf("@SQ\tSN:%s\tLN:%d", ctg.name.c_str(), ctg.len); fputs(ctg.is_alt ? "\tAH:*\n" : "\n", stdout); }
		if (!hdr_line.empty()) printf("%s\n", hdr_line.c_str());
		printf("%s\n", pg.c_str());
	}
	const int chunk = fixed_chunk > 0 ? fixed_chunk : opt.chunk_size * opt.n_threads;
	const bool long_preset = mode && (strcmp(mode, "pacbio") == 0 || strcmp(mode, "pbref") == 0 || strcmp(mode, "ont2d") == 0);
	const double t_start = now_s(), cpu_start = process_cpu_s();
	const bool tl_trace = getenv("BWAGPU_CLI_TRACE") != nullptr;      // (with the other trace lines: when each stage held each batch, seconds since here)
	if (getenv("BWAGPU_CLI_PARSE_ONLY")) {   // diagnostics: speed of the input stage alone
		Batch b; long n = 0, bp = 0;
		const bool dump = atoi(getenv("BWAGPU_CLI_PARSE_ONLY")) == 2;      // (tests: what the input stage delivers, batch by batch)
		const bool enc = atoi(getenv("BWAGPU_CLI_PARSE_ONLY")) == 3;       // (and the encoding stage: seconds, and a digest of the codes)
		double t_enc = 0; uint64_t dig = 1469598103934665603ull;
		while (read_batch(r1, pr2, chunk, b)) {
			n += (long)b.seqs.size(); for (auto &q : b.seqs) bp += q.l_seq;
			if (enc) {
				Sub u; u.idx.resize(b.seqs.size()); for (size_t i = 0; i < b.seqs.size(); ++i) u.idx[i] = (int)i;
				u.opt = opt;
				const double te = now_s(); encode_sub(b, u); t_enc += now_s() - te;
				for (int64_t i = 0; i < u.off[b.seqs.size()]; ++i) dig = (dig ^ u.flat.data()[i]) * 1099511628211ull;
			}
			if (dump) {
				printf("#batch %zu\n", b.seqs.size());
				for (auto &q : b.seqs) { const char *T = b.T(q); printf("%s\t%s\t%s\t%s\n", T + q.name, q.has_comment ? T + q.comment : "-", T + q.seq, q.has_qual ? T + q.qual : "-"); }
			}
		}
		fprintf(stderr, "[M::%s] parsed %ld records (%ld bp) in %.3f s\n", "main_mem", n, bp, now_s() - t_start);
		if (enc) fprintf(stderr, "[M::%s] encoded in %.3f s (%s), digest %016llx\n", "main_mem", t_enc, g_nt4_avx2 ? "AVX2" : "table", (unsigned long long)dig);
		return 0;
	}
	if (getenv("BWAGPU_CLI_SERIALIZE")) g_dev_serialize = atoi(getenv("BWAGPU_CLI_SERIALIZE")) != 0;
	if (getenv("BWAGPU_CLI_MATESW")) g_device_matesw = atoi(getenv("BWAGPU_CLI_MATESW"));
	if (getenv("BWAGPU_CLI_CIGARS")) g_device_cigars = atoi(getenv("BWAGPU_CLI_CIGARS"));
	int n_dev = getenv("BWAGPU_CLI_STREAMS") ? atoi(getenv("BWAGPU_CLI_STREAMS")) : 3;      // batches in flight on the device
	if (n_dev < 1) n_dev = 1;
	// devices: BWAGPU_DEVICES=0,1,... (default: the one of BWAGPU_DEVICE).  The index reaches the other devices by device-to-device copies over
	// xGMI (bwagpu_clone_to_device: one process drives all devices, so a peer copy is the direct route; the RCCL broadcast of bwa_amd/dist.py is
	// its counterpart between processes); handles[slot][device], one slot per batch in flight and device.
	// How the batches meet the devices (BWAGPU_CLI_MULTI): "split" cuts EVERY batch into one contiguous range of whole pairs per device (one
	// mem_pestat over the gathered regions, device_sub) -- right for large batches, but at the default -K a device's share of a batch is a
	// fraction of what fills it (k_seed alone has a ~35 ms floor per launch, every stage pays its heaviest read); "batch" hands WHOLE batches to
	// the devices round-robin, each device thread owning one device -- the batch, hence mem_pestat (bwamem.c:1258), is what a single device
	// sees, so the SAM is the same either way.  Default: whole batches unless a device's share of a split batch would still be >= 300 k reads.
	std::vector<int> dev_ids(1, device);
	if (const char *dl = getenv("BWAGPU_DEVICES")) {
		dev_ids.clear();
		for (const char *q = dl; *q;) { dev_ids.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
		if (dev_ids.empty() || dev_ids[0] != device) { fprintf(stderr, "[E::%s] BWAGPU_DEVICES must start with the device the index was loaded on (%d)\n", "main_mem", device); return 1; }
	}
	std::vector<std::vector<bwagpu_t*>> handles((size_t)n_dev);
	for (size_t di = 0; di < dev_ids.size(); ++di) {
		bwagpu_t *base = gpu;
		if (di > 0) { int rc = bwagpu_clone_to_device(gpu, dev_ids[di], &base); if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] device %d: %s\n", "main_mem", dev_ids[di], bwagpu_strerror(rc)); return 1; } bwagpu_set_taps(base, 0); }
		handles[0].push_back(base);
		for (int i = 1; i < n_dev; ++i) { bwagpu_t *h2 = nullptr; int rc = bwagpu_clone(base, &h2); if (rc != BWAGPU_OK) { fprintf(stderr, "[E::%s] %s\n", "main_mem", bwagpu_strerror(rc)); return 1; } bwagpu_set_taps(h2, 0); handles[(size_t)i].push_back(h2); }
	}
	bool whole_batches = dev_ids.size() > 1 && (int64_t)chunk / 150 / (int64_t)dev_ids.size() < 300000;
	if (const char *mm = getenv("BWAGPU_CLI_MULTI")) { if (!strcmp(mm, "batch")) whole_batches = dev_ids.size() > 1; else if (!strcmp(mm, "split")) whole_batches = false; }
	// the device threads: one per slot driving all devices (split), or one per slot and device driving that device alone (whole batches)
	std::vector<std::vector<bwagpu_t*>> workers;
	if (!whole_batches) workers = handles;
	else for (int i = 0; i < n_dev; ++i) for (size_t di = 0; di < dev_ids.size(); ++di) workers.push_back(std::vector<bwagpu_t*>(1, handles[(size_t)i][di]));   // (the first D threads: one per device)
	const int n_work = (int)workers.size();
	if (g_verbose >= 3 && dev_ids.size() > 1) fprintf(stderr, "[M::%s] index copied to %zu devices; %s (%d device threads)\n", "main_mem", dev_ids.size(),
													   whole_batches ? "whole batches go to the devices in turn" : "every batch is split over them", n_work);
	Chan to_enc(1), to_dev((size_t)(n_work / 4 > 2 ? n_work / 4 : 2)), to_out(2);
	// text arenas and base arrays of finished batches are handed back to the reader: re-using them saves a few hundred MB of
	// first-touch page faults per batch on the one thread that paces the pipeline
	std::mutex pool_m; std::vector<Batch> batch_pool; std::vector<HostBuf> flat_pool; std::vector<std::vector<std::string>> out_pool;   // (and the chunk strings of written batches)
	if (prewarm.th.joinable()) prewarm.th.join();
	for (auto &b_ : prewarm.flats) flat_pool.push_back(std::move(b_));
	prewarm.flats.clear();
	std::mutex dm; std::condition_variable dcv; std::map<long, WorkP> done; long next_fin = 0;   // device -> finalize, re-ordered
	std::atomic<long> n_works(-1), n_reads_total(0);
	double busy_read = 0, busy_enc = 0, busy_fin = 0, busy_write = 0; std::atomic<long> busy_dev_us(0);   // per-stage busy time (-v 3 summary)
	double cpu_read = 0, cpu_enc = 0, cpu_write = 0; std::atomic<long> cpu_dev_us(0);                    // ... and the CPU time of the single-thread stages (finalize = the process's rest)

	// watchdog (BWAGPU_CLI_WATCHDOG=<seconds>): if no stage makes progress for that long, say where everything is and give up
	std::atomic<long> progress(0); std::atomic<int> dev_no[16]; std::atomic<bool> all_done(false);
	for (auto &x : dev_no) x = -1;
	const int wd_secs = getenv("BWAGPU_CLI_WATCHDOG") ? atoi(getenv("BWAGPU_CLI_WATCHDOG")) : 0;
	std::thread watchdog([&] {
		long last = -1; int idle = 0;
		while (wd_secs > 0 && !all_done.load()) {
			std::this_thread::sleep_for(std::chrono::milliseconds(250));
			const long p = progress.load();
			if (p != last) { last = p; idle = 0; continue; }
			if (++idle < wd_secs * 4) continue;
			fprintf(stderr, "[E::%s] no progress for %d s: next batch to finalize %ld, batches read %ld, done-but-waiting %zu\n", "main_mem", wd_secs, next_fin, n_works.load(), done.size());
			for (int d = 0; d < n_work && d < 16; ++d) fprintf(stderr, "[E::%s]   device thread %d: batch %d, library phase %d\n", "main_mem", d, dev_no[d].load(), bwagpu_debug_phase(workers[(size_t)d][0]));
			_exit(3);
		}
	});

	std::thread reader([&] {      // stage 1: input, pairing classes
		// (BWAGPU_CLI_N_PROCESSED0: the number the run's first read gets -- mem_pair's tie-breaking hash takes the pair's number in the whole run, bwamem_pair.c:208,248, and
		// wraps at 2^23 pairs: the tests start a small input just below that)
		const int64_t n_processed0 = getenv("BWAGPU_CLI_N_PROCESSED0") ? atoll(getenv("BWAGPU_CLI_N_PROCESSED0")) : 0;
		if (n_processed0 != 0 && g_verbose >= 2) fprintf(stderr, "[W::%s] BWAGPU_CLI_N_PROCESSED0=%lld: reads are numbered from there (pairing ties break as they would that deep in a run)\n", "main_mem", (long long)n_processed0);
		int64_t n_processed = n_processed0; long no = 0;
		for (;;) {
			WorkP w(new Work()); w->no = no;
			{ std::lock_guard<std::mutex> l(pool_m); if (!batch_pool.empty()) { w->in = std::move(batch_pool.back()); batch_pool.pop_back(); } }
			const double tr = now_s();
			if (!read_batch(r1, pr2, chunk, w->in)) break;
			const int n = (int)w->in.seqs.size();
			long bp = 0; for (auto &q : w->in.seqs) bp += (long)q.l_seq;
			if (g_verbose >= 3) fprintf(stderr, "[M::%s] read %d sequences (%ld bp)...\n", "process", n, bp);
			if (opt.flag & F_SMARTPE) {   // -p: adjacent records with equal names are pairs (bseq_classify, bwa.c:114-130)
				Sub se, pe; bool has_last = true; int i;
				for (i = 1; i < n; ++i) {
					if (has_last) {
						if (strcmp(w->in.T(w->in.seqs[i]) + w->in.seqs[i].name, w->in.T(w->in.seqs[i - 1]) + w->in.seqs[i - 1].name) == 0) { pe.idx.push_back(i - 1); pe.idx.push_back(i); has_last = false; }
						else se.idx.push_back(i - 1);
					} else has_last = true;
				}
				if (has_last) se.idx.push_back(i - 1);
				se.opt = opt; se.opt.flag &= ~F_PE; se.n_processed = n_processed;
				pe.opt = opt; pe.opt.flag |= F_PE; pe.n_processed = n_processed + (int64_t)se.idx.size();
				if (!se.idx.empty()) w->subs.push_back(std::move(se));
				if (!pe.idx.empty()) w->subs.push_back(std::move(pe));
				if (w->subs.size() > 1) { w->out.assign((size_t)n, std::string()); w->by_read = true; }   // (two calls whose reads interleave in the output: text by the read)
			} else {
				Sub u; u.idx.resize((size_t)n); for (int i = 0; i < n; ++i) u.idx[i] = i;
				u.opt = opt; u.n_processed = n_processed;
				w->subs.push_back(std::move(u));
			}
			n_processed += n; ++no; ++progress;
			busy_read += now_s() - tr;
			if (tl_trace) fprintf(stderr, "[D::timeline] batch %ld read %.3f .. %.3f\n", no - 1, tr - t_start, now_s() - t_start);
			to_enc.push(std::move(w));
		}
		n_works = no; n_reads_total = (long)(n_processed - n_processed0);
		cpu_read = thread_cpu_s();
		to_enc.close();
	});

	// stage 1b: base codes (nst_nt4_table) of the batch's reads, flat.  A stage of its own: on the reader's thread it slowed the one stage
	// nothing can hide (round 2), on the device threads (round 3) its 27 ms per batch were time a handle's stream sat idle -- and with
	// three handles sharing one GPU the device stage is what bounds FASTQ -> SAM.
	std::thread encoder([&] {
		WorkP w;
		while (to_enc.pop(w)) {
			const double te = now_s();
			for (Sub &u : w->subs) {
				{ std::lock_guard<std::mutex> l(pool_m); if (!flat_pool.empty()) { u.flat = std::move(flat_pool.back()); flat_pool.pop_back(); } }
				encode_sub(w->in, u);
			}
			busy_enc += now_s() - te;
			++progress;
			if (tl_trace) fprintf(stderr, "[D::timeline] batch %ld encode %.3f .. %.3f\n", w->no, te - t_start, now_s() - t_start);
			to_dev.push(std::move(w));
		}
		cpu_enc = thread_cpu_s();
		to_dev.close();
		{ std::lock_guard<std::mutex> l(dm); dcv.notify_all(); }
	});

	// finished batches that may wait for the finalize stage beyond one per slot.  Round 5's last commit made it 2 without an A/B; round 6 measured, 20 M reads,
	// one session each (gpurun_out/s1, s5): 0 -> 6.15 / 6.17 / 6.19 / 5.92 Mreads/s, 2 -> 6.32 / 5.60, 4 -> 4.98: no gain on average and a wider spread, back to 0
	const long ahead = getenv("BWAGPU_CLI_AHEAD") ? atol(getenv("BWAGPU_CLI_AHEAD")) : 0;
	std::vector<std::thread> devs;
	for (int d = 0; d < n_work; ++d) devs.emplace_back([&, d] {      // stage 2: one host thread per slot (and, with whole batches, device)
		// while the reader parses the first batch: the arenas of a batch of -K bases of short reads (150 bp assumed; anything else grows them
		// later).  Not for the long-read presets: the short-read shape asks for ~4 GB per handle that a long-read run never uses and -- device
		// buffers only ever grow -- never gets back.
		if (!g_dev_serialize && !long_preset && !(getenv("BWAGPU_CLI_RESERVE") && atoi(getenv("BWAGPU_CLI_RESERVE")) == 0))      // (not under the mock runtime of the CPU tests)
			for (bwagpu_t *hh : workers[(size_t)d]) {
				const int rc = bwagpu_batch_reserve(hh, (int)((int64_t)chunk / 150 / (int64_t)workers[(size_t)d].size()) + 1024, (int64_t)chunk / (int64_t)workers[(size_t)d].size() + (1 << 20), 256);
				if (rc != BWAGPU_OK && g_verbose >= 2) fprintf(stderr, "[W::%s] could not reserve the batch arenas ahead of the first batch (%s: %s); they are grown batch by batch instead\n", "main_mem", bwagpu_strerror(rc), bwagpu_last_error(hh));
			}
		// (Round 6 also ran a small warm-up batch through every slot here, while the first batch is being read -- kernels' first launches, first staging
		// buffers.  Batch 0's hot path fell from 214 to 104 ms and the run as a whole did not move: 6.00 / 6.00 / 5.91 with, 6.21 / 6.09 / 5.90 Mreads/s without
		// (gpurun_out/s10) -- the three slots then start together and share the chip from the first millisecond; with the CIGAR and mate-rescue stages in the
		// warm-up it was worse, their result-sized buffers had to be grown by the first real batches and a grown buffer is a hipFree, a device-wide wait.  Removed.)
		WorkP w;
		while (to_dev.pop(w)) {
			// do not run far ahead of the host: a slot may start batch `no` while at most n_work + ahead batches before it are not finalized yet.  (Rounds 3-4: ahead = 0.
			// A pipeline's first finalize calls are its slowest -- 156, 75, 62 ms for batches 0..2 of the bench run -- and with no slack every slot that finished its
			// first batch waited for them: 69 + 121 + 80 ms of idle slots during the fill, profiles/r05_e2e_reserve_results.log.)
			{ std::unique_lock<std::mutex> l(dm); dcv.wait(l, [&] { return w->no - next_fin <= (long)n_work + ahead; }); }
			if (d < 16) dev_no[d] = (int)w->no;
			++progress;
			const double td = now_s();
			for (Sub &u : w->subs) { device_sub(workers[(size_t)d], u, ref, pes0); ++progress; busy_dev_us += (long)(u.t_dev * 1e6); }
			if (tl_trace) fprintf(stderr, "[D::timeline] batch %ld device %.3f .. %.3f (slot %d)\n", w->no, td - t_start, now_s() - t_start, d);
			std::lock_guard<std::mutex> l(dm);
			const long no = w->no;
			done[no] = std::move(w);
			dcv.notify_all();
		}
		cpu_dev_us += (long)(thread_cpu_s() * 1e6);
	});

	std::thread writer([&] {      // stage 4: output in input order
		WorkP w;
		if (getenv("BWAGPU_CLI_OUT_FROM_BATCH") && atol(getenv("BWAGPU_CLI_OUT_FROM_BATCH")) > 0 && g_verbose >= 2)
			fprintf(stderr, "[W::%s] BWAGPU_CLI_OUT_FROM_BATCH=%s: the records of the batches before that one are NOT written\n", "main_mem", getenv("BWAGPU_CLI_OUT_FROM_BATCH"));
		const long out_from = getenv("BWAGPU_CLI_OUT_FROM_BATCH") ? atol(getenv("BWAGPU_CLI_OUT_FROM_BATCH")) : 0;      // (bench.py's tail check: only the records of batches out_from.. are written)
		while (to_out.pop(w)) { const double tw = now_s(); if (w->no >= out_from) for (auto &t : w->out) fwrite(t.data(), 1, w->by_read ? strnlen(t.data(), t.size()) : t.size(), stdout);
			if (!w->by_read) { std::lock_guard<std::mutex> l(pool_m); if (out_pool.size() < 4) out_pool.push_back(std::move(w->out)); } /* (the reference fputs() a read's records, fastmap.c:116: a NUL -- the letter of base code 5, a '-' in the input -- ends them) */ busy_write += now_s() - tw; }
		cpu_write = thread_cpu_s();
	});

	for (;;) {                    // stage 3 (this thread drives the worker pool of finalize_batch)
		WorkP w;
		{
			std::unique_lock<std::mutex> l(dm);
			dcv.wait(l, [&] { return done.count(next_fin) || (n_works.load() >= 0 && next_fin >= n_works.load()); });
			if (!done.count(next_fin)) break;
			w = std::move(done[next_fin]); done.erase(next_fin);
		}
		const double tf = now_s();
		if (!w->by_read) { std::lock_guard<std::mutex> l(pool_m); if (!out_pool.empty()) { w->out = std::move(out_pool.back()); out_pool.pop_back(); } }
		for (Sub &u : w->subs) finalize_sub(ref, *w, u, pes0, rg_id.c_str(), copy_comment != 0);
		busy_fin += now_s() - tf;
		if (tl_trace) fprintf(stderr, "[D::timeline] batch %ld finalize %.3f .. %.3f\n", w->no, tf - t_start, now_s() - t_start);
		{
			std::lock_guard<std::mutex> l(pool_m);
			w->in.blocks.clear();                 // (the parsed blocks go back to their pool now, not when the batch is next used)
			if (batch_pool.size() < 4) batch_pool.push_back(std::move(w->in));
			for (Sub &u : w->subs) if (flat_pool.size() < 6) flat_pool.push_back(std::move(u.flat));
		}
		w->in = Batch(); w->subs.clear();
		to_out.push(std::move(w));
		{ std::lock_guard<std::mutex> l(dm); ++next_fin; ++progress; dcv.notify_all(); }
	}
	reader.join();
	encoder.join();
	for (auto &t : devs) t.join();
	to_out.close();
	writer.join();
	all_done = true; watchdog.join();
	if (g_verbose >= 3) { const double dt = now_s() - t_start; fprintf(stderr, "[M::%s] %ld reads in %.3f sec after the index was loaded: %.0f reads/s\n", "main_mem", n_reads_total.load(), dt, dt > 0 ? n_reads_total.load() / dt : 0.);
		fprintf(stderr, "[M::%s] stage busy time: read %.3f s, encode %.3f s, device %.3f s (over %d handles), finalize %.3f s, write %.3f s\n", "main_mem", busy_read, busy_enc, busy_dev_us.load() * 1e-6, n_work, busy_fin, busy_write);
		// what the stages cost in CPU time (a stage's busy time says how long it held the pipeline, not how many cores it used): the reader, encoder and writer are one
		// thread each, the device threads mostly sleep in event waits, everything else -- the finalize pool, mem_pestat's threads, the block parser -- is the rest
		const double cpu_all = process_cpu_s() - cpu_start, cpu_dev = cpu_dev_us.load() * 1e-6, cpu_rest = cpu_all - cpu_read - cpu_enc - cpu_write - cpu_dev;
		const long nr = n_reads_total.load() > 0 ? n_reads_total.load() : 1;
		fprintf(stderr, "[M::%s] stage CPU time: total %.3f s = %.3f us per read; read %.3f s, encode %.3f s, device threads %.3f s, finalize+pestat pools %.3f s, write %.3f s; one process on %d threads tops out near %.1f Mreads/s (threads / CPU time per read), its single reader thread near %.1f\n",
				"main_mem", cpu_all, cpu_all / nr * 1e6, cpu_read, cpu_enc, cpu_dev, cpu_rest > 0 ? cpu_rest : 0., cpu_write, opt.n_threads, cpu_all > 0 ? opt.n_threads / (cpu_all / nr * 1e6) : 0., cpu_read > 0 ? 1. / (cpu_read / nr * 1e6) : 0.); }
	fflush(stdout);
	for (auto &slot : handles) for (bwagpu_t *hh : slot) if (hh != gpu) bwagpu_destroy(hh);
	bwagpu_destroy(gpu);
	return 0;
}
