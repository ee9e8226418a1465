// tools/randbw.hip -- micro-benchmark: how many random, independent 64-byte block reads per second can an MI355X sustain?
// This is the access pattern of the FM-index kernels (one 64-byte Occ block per rank query, uniformly random addresses), so its
// result is the practical ceiling against which k_seed / k_sa are judged.
//   hipcc --offload-arch=gfx950 -O3 tools/randbw.hip -o tools/randbw && tools/randbw [table MiB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) k_rand(const uint4 *tab, uint64_t n_blocks, int iters, int dep, uint64_t *sink)
{
	uint64_t x = (uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
	uint64_t acc = 0;
	for (int it = 0; it < iters; ++it) {
		x ^= x << 13; x ^= x >> 7; x ^= x << 17;                      // xorshift: next random block
		uint64_t b = (x + (dep ? acc : 0)) % n_blocks;                // dep = 1: the address depends on the previous block's data
		const uint4 *p = tab + b * 4;
		uint4 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3];             // one 64-byte block as 4 x dwordx4, like load_block()
		acc += a0.x + a1.y + a2.z + a3.w;
	}
	if (acc == 0xdeadbeef) *sink = acc;
}

int main(int argc, char **argv)
{
	size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 4096;
	size_t bytes = mib << 20; uint64_t n_blocks = bytes / 64;
	uint4 *tab; uint64_t *sink;
	hipMalloc(&tab, bytes); hipMalloc(&sink, 8);
	hipMemset(tab, 1, bytes);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	const int waves_per_simd[] = {1, 2, 4, 8};
	for (int dep = 0; dep < 2; ++dep)
		for (int wi = 0; wi < 4; ++wi) {
			int blocks = 256 * waves_per_simd[wi];                        // 256 CUs x (waves/SIMD) blocks of 4 waves
			int iters = 2000;
			hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, 100, dep, sink);
			hipDeviceSynchronize();
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(k_rand, dim3(blocks), dim3(256), 0, 0, tab, n_blocks, iters, dep, sink);
			hipEventRecord(e1, 0); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			double n = (double)blocks * 256 * iters;
			printf("table %zu MiB  %s  %d waves/SIMD: %.2f G blocks/s  = %.0f GB/s  (lat x conc: %.2f us per dependent step)\n", mib,
				   dep ? "dependent  " : "independent", waves_per_simd[wi], n / ms / 1e6, n * 64 / ms / 1e6, ms * 1e3 / iters);
		}
	return 0;
}
