// dev_debug.h -- bwagpu_debug_dp: one wavefront of a device DP routine on caller-supplied sequences (differential tests).
//
// The whole-read parity tests feed the DP routines a narrow distribution of (h0, band, lengths); this entry lets a test drive
// wave_ksw_extend2 (k_extend_wave, both modes), wave_ksw_global2 (k_cigar), wave_global2_score_ring (k_dedup_wave) and the
// ksw_align2 restatement of k_matesw_sw with adversarial inputs and compare them with the reference's exported ksw_extend2
// (ksw.c:416), ksw_global2 (ksw.c:540) and ksw_align2 (ksw.c:379).  The routines read their target from 2-bit packed reference
// text (ref_base), so the caller's sequence array is packed into a scratch "pac" and the kernels run on a DevIndex that points
// at it; everything else is the product code, called exactly as the product kernels call it.
#pragma once
#include "dev_extw.h"
#include "dev_extp.h"
#include "dev_cigar.h"
#include "dev_dedupw.h"
#include "dev_matesw.h"

#define DBG_OUT_INTS 72

DEVFN void dbg_case_geometry(const bwagpu_dp_case_t &c, i64 l_pac, int &q0, int &qdir, i64 &t0, int &tdir)
{
	q0 = (c.flags & 1) ? c.q_len - 1 : 0; qdir = (c.flags & 1) ? -1 : 1;       // bit 0: the query is presented back to front
	if (c.flags & 4) {                                                        // bit 2: the target's complement, through the reverse-strand half of the text
		if (c.flags & 2) { t0 = (l_pac << 1) - 1 - c.t_off; tdir = -1; }       //   ... read back to front at that (complement of the reversed target)
		else { t0 = (l_pac << 1) - 1 - (c.t_off + c.t_len - 1); tdir = 1; }
	} else if (c.flags & 2) { t0 = c.t_off + c.t_len - 1; tdir = -1; }         // bit 1: the target is presented back to front
	else { t0 = c.t_off; tdir = 1; }
}

// kind 0 / 1: wave_ksw_extend2 as k_extend_wave sets it up (columns + read profile in LDS / ring mode)
template <bool RING> __global__ void __launch_bounds__(64) k_debug_extend(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs,
																			 int max_q, int ring_cols, i32 *out, int blk)
{
	HIP_DYNAMIC_SHARED(unsigned char, dbg_lds)
	const int lane = threadIdx.x & 63;
	WaveLds L;
	L.stat = nullptr; L.blk = RING ? blk : 0;
	L.eh = (int2*)dbg_lds;
	if (RING) {
		int8_t *m = (int8_t*)(dbg_lds + (size_t)8 * ring_cols);
		if (lane < 25) m[lane] = opt.mat[lane];
		L.mat = m; L.ring_mask = ring_cols - 1; L.qp = nullptr; L.qstride = 0;
	} else {
		L.qstride = (max_q + 64 + 3) & ~3;
		L.qp = (int8_t*)(dbg_lds + (size_t)8 * (max_q + 2 + 64));
		int8_t *m = L.qp + 5 * L.qstride;
		if (lane < 25) m[lane] = opt.mat[lane];
		L.mat = m; L.ring_mask = 0;
	}
	wave_sync();
	const int mat_max = opt_mat_max(opt);
	for (int k = blockIdx.x; k < n_cases; k += gridDim.x) {
		const bwagpu_dp_case_t c = cases[k];
		const u8 *q = seqs + c.q_off;
		int q0, qdir, tdir; i64 t0;
		dbg_case_geometry(c, ix.l_pac, q0, qdir, t0, tdir);
		if (!RING) {      // the read's profile (ext_read_wave builds it once per read)
			for (int j = lane; j < c.q_len; j += 64) { const int qc = q[j]; for (int b = 0; b < 5; ++b) L.qp[b * L.qstride + j] = L.mat[b * 5 + qc]; }
			wave_sync();
		}
		u64 cells = 0, fast = 0;
		const ExtRes r = wave_ksw_extend2<RING>(ix, opt, mat_max, q, q0, qdir, c.q_len, t0, tdir, c.t_len, c.w, c.end_bonus, c.h0, L, cells, fast);
		wave_sync();
		if (lane == 0) {
			i32 *o = out + (size_t)k * DBG_OUT_INTS;
			o[0] = r.score; o[1] = r.qle; o[2] = r.tle; o[3] = r.gtle; o[4] = r.gscore; o[5] = r.max_off; o[6] = (i32)fast; o[7] = (i32)cells;
		}
	}
}

// kind 6 / 7: the packed extension routine of k_ext_pack (dev_extp.h), four cases per wavefront, four / eight columns per lane.  out[0..5] as kind 0,
// out[6]: answered by the diagonal rule, out[7]: 1 = the routine vouches for the result (0: outside its conditions -- the product then runs kind 0's routine)
template <int CPL> __global__ void __launch_bounds__(64) k_debug_extpack(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs, i32 *out)
{
	HIP_DYNAMIC_SHARED(unsigned char, dbg_lds)
	const int lane = threadIdx.x & 63, grp = lane >> 4;
	int8_t *mat = (int8_t*)dbg_lds;
	if (lane < 25) mat[lane] = opt.mat[lane];
	wave_sync();
	PackConst C;
	C.o_del = opt.o_del; C.e_del = opt.e_del; C.o_ins = opt.o_ins; C.e_ins = opt.e_ins; C.oe_del = opt.o_del + opt.e_del; C.oe_ins = opt.o_ins + opt.e_ins;
	C.zdrop = opt.zdrop; C.mat = mat; C.prof = (int8_t*)(dbg_lds + 32); C.mat_max = opt_mat_max(opt);
	for (int base = blockIdx.x * 4; base < n_cases; base += gridDim.x * 4) {
		const int k = base + grp;
		PackState<CPL> S;
		#pragma unroll
		for (int c = 0; c < CPL; ++c) { S.H[c] = 0; S.E[c] = 0; }
		S.run = 0; S.done = 0; S.valid = 0; S.i = 0; S.end = 0; S.qlen = 1; S.tlen = 0; S.w = 0; S.h0 = 1; S.max = 0; S.max_i = S.max_j = S.max_ie = -1; S.gscore = -1; S.max_off = 0;
		S.tdir = 1; S.t0 = 0; S.tpack = 0; S.tnext = 0; S.lq = 0; S.sq = 0;
		u64 fast = 0;
		{
			const bwagpu_dp_case_t c = cases[k < n_cases ? k : n_cases - 1];
			PackTask T;
			T.q = seqs + c.q_off; T.qlen = c.q_len; T.tlen = c.t_len; T.w = c.w; T.end_bonus = c.end_bonus; T.h0 = c.h0;
			dbg_case_geometry(c, ix.l_pac, T.q0, T.qdir, T.t0, T.tdir);
			pack_init<CPL>(ix, C, T, S, k < n_cases, fast);
		}
		while (__ballot(S.run != 0)) pack_row<CPL, true>(ix, C, S);
		if (k < n_cases && (lane & 15) == 0) {
			const ExtRes r = pack_result(S);
			i32 *o = out + (size_t)k * DBG_OUT_INTS;
			o[0] = r.score; o[1] = r.qle; o[2] = r.tle; o[3] = r.gtle; o[4] = r.gscore; o[5] = r.max_off; o[6] = (i32)fast; o[7] = S.valid;
		}
		wave_sync();
	}
}

// kind 2: wave_ksw_global2 with traceback as k_cigar sets it up (second tier's LDS); out: score, n_ops (-1: more than 64, -2: outside
// the kernel's limits), then the operations in alignment order
__global__ void __launch_bounds__(64) k_debug_global(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs, i32 *out)
{
	HIP_DYNAMIC_SHARED(unsigned char, dbg_lds)
	const int lane = threadIdx.x & 63;
	CigLds L;
	L.hd = (i32*)dbg_lds; L.e = L.hd + (CIG_MAX_LEN + 2 + 64);
	L.qstride = CIG_MAX_LEN + 64; L.z_cells = CIG_Z_BIG;
	L.qp = (int8_t*)(L.e + (CIG_MAX_LEN + 2 + 64));
	L.z = (u8*)(L.qp + 5 * L.qstride);
	L.ops = (u32*)(L.z + CIG_Z_BIG / 2 + CIG_MAX_COLS);
	L.md = (u8*)(L.ops + CIG_TMP_OPS);
	for (int k = blockIdx.x; k < n_cases; k += gridDim.x) {
		const bwagpu_dp_case_t c = cases[k];
		const u8 *q = seqs + c.q_off;
		int q0, qdir, tdir; i64 t0;
		dbg_case_geometry(c, ix.l_pac, q0, qdir, t0, tdir);
		i32 *o = out + (size_t)k * DBG_OUT_INTS;
		const int n_col = c.q_len < 2 * c.w + 1 ? c.q_len : 2 * c.w + 1;
		if (c.q_len > CIG_MAX_LEN || c.t_len > CIG_MAX_LEN || n_col > CIG_MAX_COLS || n_col * ((c.t_len + 1) & ~1) > CIG_Z_BIG) { if (lane == 0) { o[0] = 0; o[1] = -2; } continue; }
		int n_ops = 0;
		const int score = wave_ksw_global2(ix, opt, q, q0, qdir, c.q_len, t0, tdir, c.t_len, c.w, L, &n_ops);
		if (lane == 0) { o[0] = score; o[1] = n_ops; for (int j = 0; j < n_ops; ++j) o[2 + j] = (i32)L.ops[n_ops - 1 - j]; }
		wave_sync();
	}
}

// kind 3: the score-only ring form of k_dedup_wave
template <bool BLK = false> __global__ void __launch_bounds__(64) k_debug_global_ring(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs, int ring_cols, i32 *out, int q_cap)
{
	HIP_DYNAMIC_SHARED(unsigned char, dbg_lds)
	const int lane = threadIdx.x & 63;
	DedupLds L;
	L.hd = (i32*)dbg_lds; L.e = L.hd + ring_cols; L.ring_mask = ring_cols - 1; L.H = nullptr; L.E = nullptr; L.qbuf = dbg_lds + (size_t)8 * ring_cols + 32; L.qcap = q_cap;
	int8_t *m = (int8_t*)(dbg_lds + (size_t)8 * ring_cols);
	if (lane < 25) m[lane] = opt.mat[lane];
	L.mat = m;
	wave_sync();
	for (int k = blockIdx.x; k < n_cases; k += gridDim.x) {
		const bwagpu_dp_case_t c = cases[k];
		int q0, qdir, tdir; i64 t0;
		dbg_case_geometry(c, ix.l_pac, q0, qdir, t0, tdir);
		i32 *o = out + (size_t)k * DBG_OUT_INTS;
		if (2 * c.w + 4 + 128 > ring_cols) { if (lane == 0) { o[0] = 0; o[1] = -2; } continue; }
		u64 cells = 0;
		const int score = BLK && L.qcap >= c.q_len ? wave_global2_score_ring_blk(ix, opt, seqs + c.q_off, q0, qdir, c.q_len, t0, tdir, c.t_len, c.w, L, cells)
													   : wave_global2_score_ring(ix, opt, seqs + c.q_off, q0, qdir, c.q_len, t0, tdir, c.t_len, c.w, L, cells);
		if (lane == 0) { o[0] = score; o[1] = 0; o[7] = (i32)cells; }
		wave_sync();
	}
}

// kind 4: ksw_align2 as k_matesw_sw runs it (one wavefront per case; h0 carries the xtra word); out: score, te, qe, score2, te2, tb, qb
__global__ void __launch_bounds__(64) k_debug_align2(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs, i32 *out)
{
	__shared__ i32 dbg_runs[MSW_RUN_INTS];
	const int lane = threadIdx.x & 63;
	for (int k = blockIdx.x; k < n_cases; k += gridDim.x) {
		const bwagpu_dp_case_t c = cases[k];
		i32 *o = out + (size_t)k * DBG_OUT_INTS;
		if (c.q_len > MSW_MAX_Q || c.t_len > MSW_MAX_T) { if (lane == 0) { o[0] = 0; o[1] = -2; } continue; }
		const u8 *q = seqs + c.q_off;
		int q0, qdir, tdir; i64 t0;
		dbg_case_geometry(c, ix.l_pac, q0, qdir, t0, tdir);
		auto Qf = [&](int j) -> int { return (int)q[q0 + j * qdir]; };
		auto Tf = [&](int i) -> int { return ref_base(ix, t0 + (i64)i * tdir); };
		int res[7];
		msw_align2(opt, c.q_len, Qf, c.t_len, Tf, c.h0, dbg_runs, res);
		if (lane == 0) for (int j = 0; j < 7; ++j) o[j] = res[j];
		wave_sync();
	}
}

// kind 5: wave_ksw_global2_long as k_cigar_long sets it up; out: score, n_ops (-1: too many, -2: outside the limits), the first 70 operations
__global__ void __launch_bounds__(64) k_debug_global_long(DevIndex ix, bwagpu_opt_t opt, int n_cases, const bwagpu_dp_case_t *cases, const u8 *seqs, u8 *z_all, i64 z_cap, u32 *ops_all, i32 *out)
{
	HIP_DYNAMIC_SHARED(unsigned char, dbg_lds)
	const int lane = threadIdx.x & 63;
	CigLongLds L;
	cigl_lds_setup(dbg_lds, L);
	if (lane < 25) L.mat[lane] = opt.mat[lane];
	CigLongScratch S;
	S.z = z_all + (i64)blockIdx.x * z_cap; S.z_cap = z_cap; S.ops = ops_all + (size_t)blockIdx.x * CIGL_MAX_OPS; S.md = nullptr;
	wave_sync();
	for (int k = blockIdx.x; k < n_cases; k += gridDim.x) {
		const bwagpu_dp_case_t c = cases[k];
		int q0, qdir, tdir; i64 t0;
		dbg_case_geometry(c, ix.l_pac, q0, qdir, t0, tdir);
		i32 *o = out + (size_t)k * DBG_OUT_INTS;
		const int n_col = c.q_len < 2 * c.w + 1 ? c.q_len : 2 * c.w + 1;
		if (n_col > CIGL_MAX_COLS || (i64)c.t_len * ((n_col + 15) & ~15) > z_cap) { if (lane == 0) { o[0] = 0; o[1] = -2; } continue; }
		int n_ops = 0;
		const int score = wave_ksw_global2_long(ix, opt, seqs + c.q_off, q0, qdir, c.q_len, t0, tdir, c.t_len, c.w, L, S, &n_ops);
		if (lane == 0) { o[0] = score; o[1] = n_ops; for (int j = 0; j < n_ops && j < DBG_OUT_INTS - 2; ++j) o[2 + j] = (i32)S.ops[n_ops - 1 - j]; }
		wave_sync();
	}
}

// 2-bit packing of the case array's bases (codes > 3 are stored as their low two bits: targets hold 0..3 only)
__global__ void __launch_bounds__(256) k_debug_pack(const u8 *seqs, i64 n, u8 *pac)
{
	for (i64 b = (i64)blockIdx.x * blockDim.x + threadIdx.x; b < (n + 3) / 4; b += (i64)gridDim.x * blockDim.x) {
		u32 v = 0;
		for (int k = 0; k < 4; ++k) { const i64 l = b * 4 + k; const u32 c = l < n ? seqs[l] & 3u : 0u; v |= c << ((3 - k) << 1); }   // _set_pac (bntseq.c:229)
		pac[b] = (u8)v;
	}
}
