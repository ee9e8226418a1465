// tools/randbw2.hip -- micro-benchmark: random reads of 16, 32, 64 and 128 bytes (aligned to their size) from a table in HBM.
// Question behind it: is the chip's ceiling for the FM-index access pattern a number of requests or a number of bytes?  (A 32-byte Occ
// block with 32-bit in-block counts is a possible index layout; it pays only if 32-byte requests come faster than 64-byte ones.)
//   hipcc --offload-arch=gfx950 -O3 tools/randbw2.hip -o tools/randbw2 && tools/randbw2 [table MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int NB> __global__ void __launch_bounds__(256) k_rand(const uint4 *tab, uint64_t n_units, int iters, uint64_t *sink)
{
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;
		const uint4 *p = tab + ((x + acc) % n_units) * (NB / 16);       // dependent on the previous read, like the index walk
		uint4 v[NB / 16];
#pragma unroll
		for (int k = 0; k < NB / 16; ++k) v[k] = p[k];
#pragma unroll
		for (int k = 0; k < NB / 16; ++k) acc += v[k].x + v[k].w;
	}
	if (acc == 0xdeadbeef) *sink = acc;
}

template <int NB> static void run(const uint4 *tab, size_t bytes, uint64_t *sink, int wps)
{
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int blocks = 256 * wps, iters = 1500;
	hipLaunchKernelGGL(k_rand<NB>, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)(bytes / NB), 100, sink);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(k_rand<NB>, dim3(blocks), dim3(256), 0, 0, tab, (uint64_t)(bytes / NB), iters, sink);
	hipEventRecord(e1, 0); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double n = (double)blocks * 256 * iters;
	printf("table %zu MiB  %3d-byte reads  %d waves/SIMD: %.2f G reads/s = %.0f GB/s\n", bytes >> 20, NB, wps, n / ms / 1e6, n * NB / ms / 1e6);
}

int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 4096;
	size_t bytes = mib << 20;
	uint4 *tab; uint64_t *sink;
	if (hipMalloc(&tab, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
	hipMemset(tab, 1, bytes);
	for (int wps = 2; wps <= 4; wps += 2) { run<16>(tab, bytes, sink, wps); run<32>(tab, bytes, sink, wps); run<64>(tab, bytes, sink, wps); run<128>(tab, bytes, sink, wps); }
	return 0;
}
