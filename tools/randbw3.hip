// tools/randbw3.hip -- micro-benchmark: does the chip's ceiling for dependent random reads move when the lanes of a quad fetch ONE block
// together?  k_seed's idiom (dev_fm.h load_block) is "one lane fetches its whole block as 2 or 4 x dwordx4": every load instruction then
// touches 64 different lines for 16 bytes each.  The cooperative form transposes the fetch: in step s the lanes of a group load the
// 16-byte pieces of the block wanted by the group's lane s (one fully used 32- or 64-byte segment per group and instruction), reduce
// what they read over the group with DPP quad_perm and hand the result to that lane -- the same number of blocks, bytes and loads in
// flight per lane, a different shape per instruction.
//   hipcc --offload-arch=gfx950 -O3 tools/randbw3.hip -o tools/randbw3 && tools/randbw3 [table MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__device__ __forceinline__ uint64_t next_rand(uint64_t &x) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; }
template <int CTRL> __device__ __forceinline__ uint32_t qperm(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ uint64_t qperm64(uint64_t v) { return (uint64_t)qperm<CTRL>((uint32_t)(v >> 32)) << 32 | qperm<CTRL>((uint32_t)v); }

// the kernels' idiom: a lane fetches its own block
template <int NB> __global__ void __launch_bounds__(256) k_lane(const uint4 *tab, uint64_t n_units, int iters, uint64_t *sink)
{
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
	for (int it = 0; it < iters; ++it) {
		const uint4 *p = tab + ((next_rand(x) + acc) % n_units) * (NB / 16);
		uint4 v[NB / 16];
#pragma unroll
		for (int k = 0; k < NB / 16; ++k) v[k] = p[k];
#pragma unroll
		for (int k = 0; k < NB / 16; ++k) acc += v[k].x + v[k].w;
	}
	if (acc == 0xdeadbeef) *sink = acc;
}

// group-cooperative: NB / 16 adjacent lanes per block
template <int NB> __global__ void __launch_bounds__(256) k_coop(const uint4 *tab, uint64_t n_units, int iters, uint64_t *sink)
{
	constexpr int G = NB / 16;
	static_assert(G == 2 || G == 4, "quad_perm reaches four lanes");
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345, acc = 0;
	const int sub = threadIdx.x & (G - 1);
	for (int it = 0; it < iters; ++it) {
		const uint64_t idx = (next_rand(x) + acc) % n_units;
		uint64_t want[G]; uint4 piece[G];
		if (G == 4) { want[0] = qperm64<0x00>(idx); want[1] = qperm64<0x55>(idx); want[2] = qperm64<0xAA>(idx); want[3] = qperm64<0xFF>(idx); }
		else { want[0] = qperm64<0xA0>(idx); want[1] = qperm64<0xF5>(idx); }
#pragma unroll
		for (int s = 0; s < G; ++s) piece[s] = tab[want[s] * G + sub];       // G independent loads in flight, as in k_lane
#pragma unroll
		for (int s = 0; s < G; ++s) {
			uint32_t part = piece[s].x + piece[s].w;
			part += qperm<0xB1>(part);                                       // lanes 0<->1, 2<->3
			if (G == 4) part += qperm<0x4E>(part);                           // pairs 01<->23
			if (sub == s) acc += part;
		}
	}
	if (acc == 0xdeadbeef) *sink = acc;
}

template <class K> static void run(K kernel, const char *what, int nb, const uint4 *tab, size_t bytes, uint64_t *sink, int wps)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * wps, iters = 1500;
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)(bytes / nb), 100, sink);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)(bytes / nb), iters, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)blocks * 256 * iters;
	printf("table %zu MiB  %3d-byte blocks  %-26s %d waves/SIMD: %.2f G blocks/s = %.0f GB/s\n", bytes >> 20, nb, what, wps, n / ms / 1e6, n * nb / ms / 1e6);
}

int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 4096;
	size_t bytes = mib << 20;
	uint4 *tab; uint64_t *sink;
	if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
	hipMemset(tab, 1, bytes);
	for (int wps = 2; wps <= 4; wps += 2) {
		run(k_lane<32>, "lane fetches its block", 32, tab, bytes, sink, wps);
		run(k_coop<32>, "pair-cooperative", 32, tab, bytes, sink, wps);
		run(k_lane<64>, "lane fetches its block", 64, tab, bytes, sink, wps);
		run(k_coop<64>, "quad-cooperative", 64, tab, bytes, sink, wps);
	}
	return 0;
}
