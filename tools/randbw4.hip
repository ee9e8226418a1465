// tools/randbw4.hip -- micro-benchmark: what does one k_seed extension step cost the memory pipeline, piece by piece?
// A step of the one-trip seeding kernel (dev_seed.h ext_one_trip, dev_fm.h occ32_issue) is, per lane: two 32-byte index blocks as 2 x 2
// range-checked 16-byte buffer loads (the rank positions k-1 and k+s-1), two 16-byte superblock entries from a table of a few hundred bytes
// (always an L1 hit), a prefix-table entry and a stack prefetch that most lanes range-check away.  VERDICT r4 puts the kernel at ~90 % of the
// chip's ceiling for lane-private dependent random 32-byte reads; this program measures how that ceiling moves with each piece:
//   two        two random blocks per step (4 loads)                         -- a step on a large interval
//   two+sb     ... plus the two superblock loads (6 loads)                  -- what k_seed issues today
//   same       one random block, asked for twice (4 loads, 2 distinct)      -- a step on a one-row interval today
//   same+sb    ... plus the two superblock loads
//   one        one random block, asked for once (2 loads)
//   one+sb     ... plus one superblock load
//   one16      one random 16-byte entry (1 load)                            -- a prefix-table or dense-SA look-up
//   two/half   `two` with every other lane range-checked away (offset beyond the buffer): does an idle lane cost anything?
//   two/quarter three lanes of four range-checked away
// Every variant is a dependent chain per lane (the next address depends on the loaded data), 1500 steps, 256 x wps workgroups of 256.
//   hipcc --offload-arch=gfx950 -O3 tools/randbw4.hip -o tools/randbw4 && tools/randbw4
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

typedef __amdgpu_buffer_rsrc_t BufRsrc;
#define BUF_OOB 0xFFFFFF00u
__device__ __forceinline__ BufRsrc rsrc(const void *p, uint64_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(uint32_t)bytes, 0x00020000); }
__device__ __forceinline__ uint4 ld16(BufRsrc r, uint32_t off) { const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0); return make_uint4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ uint64_t next_rand(uint64_t &x) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
__device__ __forceinline__ void keep(uint4 &v) { asm volatile("" : "+v"(v.x)); asm volatile("" : "+v"(v.y)); asm volatile("" : "+v"(v.z)); asm volatile("" : "+v"(v.w)); }

// MODE: 0 two, 1 same, 2 one, 3 one16;  SB: superblock loads per block position;  LIVE: one lane in LIVE issues real offsets
template <int MODE, int SB, int LIVE> __global__ void __launch_bounds__(256) k_step(const uint4 *tab, uint64_t tab_bytes, const uint4 *sbt, uint32_t n_units, int iters, uint64_t *sink)
{
	const BufRsrc rt = rsrc(tab, tab_bytes), rs = rsrc(sbt, 256);
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
	const bool live = (threadIdx.x % LIVE) == 0;
	for (int it = 0; it < iters; ++it) {
		const uint32_t a = (uint32_t)((next_rand(x) + acc) % n_units), b = MODE == 0 ? (uint32_t)((next_rand(x) + acc) % n_units) : a;
		const uint32_t oa = live ? a << 5 : BUF_OOB, ob = live ? b << 5 : BUF_OOB;
		uint4 v0 = ld16(rt, oa), v1 = make_uint4(0, 0, 0, 0), v2 = v1, v3 = v1, s0 = v1, s1 = v1;
		if (MODE != 3) v1 = ld16(rt, oa + 16);
		if (MODE < 2) { v2 = ld16(rt, ob); v3 = ld16(rt, ob + 16); }
		if (SB >= 1) s0 = ld16(rs, live ? (a & 15) << 4 : BUF_OOB);
		if (SB >= 2) s1 = ld16(rs, live ? (b >> 4 & 15) << 4 : BUF_OOB);
		keep(v0); keep(v1); keep(v2); keep(v3); keep(s0); keep(s1);
		acc += v0.x + v1.w + v2.x + v3.w + s0.x + s1.y;
	}
	if (acc == 0xdeadbeef) *sink = acc;
}

template <class K> static void run(K kernel, const char *what, const uint4 *tab, size_t bytes, const uint4 *sbt, uint64_t *sink, int wps)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * wps, iters = 1500;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)bytes, sbt, (uint32_t)(bytes / 32), 100, sink);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)bytes, sbt, (uint32_t)(bytes / 32), iters, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)blocks * 256 * iters;
	printf("%-12s %d waves/SIMD: %7.2f G lane-steps/s  (%.3f ms; %.1f ps per lane-step chip-wide)\n", what, wps, n / ms / 1e6, ms, ms * 1e9 / n);
	fflush(stdout);
}

int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 3072;      // (a V# reaches < 4 GiB)
	size_t bytes = mib << 20;
	uint4 *tab, *sbt; uint64_t *sink;
	if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sbt, 256) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
	hipMemset(tab, 1, bytes); hipMemset(sbt, 1, 256);
	printf("table %zu MiB, 32-byte units, dependent random reads through range-checked buffer loads\n", mib);
	for (int wps = 4; wps <= 4; wps += 2) {
		run(k_step<0, 0, 1>, "two", tab, bytes, sbt, sink, wps);
		run(k_step<0, 2, 1>, "two+sb", tab, bytes, sbt, sink, wps);
		run(k_step<1, 0, 1>, "same", tab, bytes, sbt, sink, wps);
		run(k_step<1, 2, 1>, "same+sb", tab, bytes, sbt, sink, wps);
		run(k_step<2, 0, 1>, "one", tab, bytes, sbt, sink, wps);
		run(k_step<2, 1, 1>, "one+sb", tab, bytes, sbt, sink, wps);
		run(k_step<3, 0, 1>, "one16", tab, bytes, sbt, sink, wps);
		run(k_step<0, 0, 2>, "two/half", tab, bytes, sbt, sink, wps);
		run(k_step<0, 0, 4>, "two/quarter", tab, bytes, sbt, sink, wps);
		run(k_step<0, 2, 2>, "two+sb/half", tab, bytes, sbt, sink, wps);
	}
	return 0;
}
