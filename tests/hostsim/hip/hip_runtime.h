// tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal *mock* of the HIP runtime so that the unmodified product source bwa_amd/csrc/bwagpu.hip can be
// compiled with g++ and exercised by the CPU-only test-suite (this container has no GPU).
//
// Execution model: hipLaunchKernelGGL runs the grid block by block; inside a block every lane is a ucontext
// fiber.  A fiber runs until it reaches a wave collective (__shfl*, __ballot, wave barrier) where it parks until
// all live lanes of its 64-lane wave have arrived -- i.e. lanes are *not* in lock-step between collectives, which
// is stricter than the hardware and catches missing intra-wave synchronisation.  Lane-serial kernels never
// yield.  Static/dynamic __shared__ memory is one host buffer (blocks run one at a time).
//
// The mock is never part of the product: libbwagpu.so is built by hipcc against the real runtime and has no CPU
// path.  The library built from this header is tests/hostsim/libbwagpu_hostsim.so.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint4 { uint32_t x, y, z, w; };
struct int2 { int x, y; };
struct uint2 { uint32_t x, y; };
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
struct int4 { int x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline int4 make_int4(int x, int y, int z, int w) { int4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline int2 make_int2(int x, int y) { int2 r; r.x = x; r.y = y; return r; }
extern dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorMock = 1 };
typedef struct mock_stream_s *hipStream_t;
struct mock_event_s { std::chrono::steady_clock::time_point t; };
typedef mock_event_s *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

static inline const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("MOCK_HIP_DEVICES"); *n = e ? atoi(e) : 1; return hipSuccess; }   // all mock devices share the host's memory
static inline hipError_t hipMemcpyPeer(void *d, int, const void *s, int, size_t n) { memcpy(d, s, n); return hipSuccess; }
extern int mock_cur_device;
static inline hipError_t hipSetDevice(int d) { mock_cur_device = d; return hipSuccess; }
static inline hipError_t hipGetDevice(int *d) { *d = mock_cur_device; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new mock_event_s(); return hipSuccess; }
#define hipEventBlockingSync 1
#define hipEventDisableTiming 2
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = new mock_event_s(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }   // (launches are synchronous here)
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
// device memory comes back filled with a poison pattern: code that consumes bytes nobody wrote (a stale arena slot, a flag never
// cleared) then fails here -- as a wild index or a wrong result -- the way it may on a GPU whose allocator recycles pages
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCB, n < ((size_t)32 << 20) ? (n ? n : 1) : ((size_t)32 << 20));   /* (the first 32 MiB: scratch areas sized for a whole chip are never touched otherwise) */ return *p ? hipSuccess : hipErrorMock; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
// (the mock's "device" reports a fixed amount of memory, MOCK_HIP_FREE_MB of it free -- tests of the callers' headroom arithmetic set it)
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { const char *e = getenv("MOCK_HIP_FREE_MB"); *t = (size_t)288 << 30; *f = e ? (size_t)atoll(e) << 20 : (size_t)280 << 30; return hipSuccess; }
#define hipHostMallocDefault 0
#define hipHostMallocPortable 1
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); if (*p) memset(*p, 0xCD, n < ((size_t)1 << 20) ? n : ((size_t)1 << 20)); return *p ? hipSuccess : hipErrorMock; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned int x) { return __builtin_popcount(x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
static inline unsigned int atomicAdd(unsigned int *p, unsigned int v) { unsigned int o = *p; *p += v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p += v; return o; }

// ---- wave collectives (implemented in mock_globals.cpp on top of the fiber scheduler) ----
void mock_exchange(uint64_t v, uint64_t out[64], uint64_t *active_mask);
void mock_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body);
extern unsigned char *mock_dyn_lds;

static inline int mock_lane() { return (int)(threadIdx.x & 63); }
static inline long long wall_clock64() { return (long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10; }   // (the device's constant 100 MHz counter)
static inline int __mul24(int a, int b) { return ((a << 8) >> 8) * ((b << 8) >> 8); }
static inline int __shfl(int v, int src) { uint64_t o[64], m; mock_exchange((uint32_t)v, o, &m); return (m >> (src & 63) & 1) ? (int)(uint32_t)o[src & 63] : v; }
static inline int __shfl_up(int v, int d) { uint64_t o[64], m; mock_exchange((uint32_t)v, o, &m); int s = mock_lane() - d; return (s >= 0 && (m >> s & 1)) ? (int)(uint32_t)o[s] : v; }
static inline int __shfl_down(int v, int d) { uint64_t o[64], m; mock_exchange((uint32_t)v, o, &m); int s = mock_lane() + d; return (s < 64 && (m >> s & 1)) ? (int)(uint32_t)o[s] : v; }
static inline int __shfl_xor(int v, int x) { uint64_t o[64], m; mock_exchange((uint32_t)v, o, &m); int s = mock_lane() ^ x; return (m >> s & 1) ? (int)(uint32_t)o[s] : v; }
static inline unsigned long long __ballot(int p) { uint64_t o[64], m; mock_exchange(p ? 1 : 0, o, &m); unsigned long long r = 0; for (int i = 0; i < 64; ++i) if ((m >> i & 1) && o[i]) r |= 1ull << i; return r; }
void mock_block_barrier();
#define __syncthreads() mock_block_barrier()
static inline void mock_wave_barrier() { uint64_t o[64], m; mock_exchange(0, o, &m); }
// DPP / readlane emulation (gfx9 semantics; bound_ctrl = 0 keeps `old` where the source lane is invalid or masked off)
static inline int mock_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
	uint64_t o[64], m; mock_exchange((uint32_t)src, o, &m);
	int l = mock_lane(), row = l >> 4, s = -1;
	if (!((row_mask >> row) & 1) || !((bank_mask >> ((l & 15) >> 2)) & 1)) return old;
	if (ctrl >= 0 && ctrl <= 0xff) s = (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3);      // quad_perm
	else if (ctrl >= 0x111 && ctrl <= 0x11f) { int n = ctrl - 0x110; s = (l & 15) >= n ? l - n : -1; }
	else if (ctrl >= 0x121 && ctrl <= 0x12f) { int n = ctrl - 0x120; s = (l & ~15) | (((l & 15) - n) & 15); }   // row_ror:n (rotation within the row of 16)
	else if (ctrl == 0x138) s = l >= 1 ? l - 1 : -1;
	else if (ctrl == 0x130) s = l < 63 ? l + 1 : -1;                                   // wave_shl:1 (the lane above)
	else if (ctrl == 0x142) s = row >= 1 ? row * 16 - 1 : -1;
	else if (ctrl == 0x143) s = row >= 2 ? 31 : -1;
	else abort();
	if (s < 0 || !((m >> s) & 1)) return bound_ctrl ? 0 : old;
	return (int)(uint32_t)o[s];
}
static inline int mock_readlane(int v, int src) { uint64_t o[64], m; mock_exchange((uint32_t)v, o, &m); return (int)(uint32_t)o[src & 63]; }
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) mock_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_readlane(v, l) mock_readlane((v), (l))
#define __builtin_amdgcn_readfirstlane(v) (v)   /* used only on wave-uniform values */
#define __builtin_amdgcn_fence(...) ((void)0)
#define __threadfence() ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)
#define __builtin_amdgcn_wave_barrier() mock_wave_barrier()
#define DEV_KEEP(v) ((void)0)            /* scheduling fence of the device code (dev_common.h): nothing to do here */
// buffer descriptors and range-checked 16-byte loads (gfx9 raw buffers: a lane whose offset + 16 exceeds the size reads zeros)
struct mock_v4u32 { uint32_t v[4]; uint32_t operator[](int i) const { return v[i]; } };
struct mock_rsrc_t { const unsigned char *p; uint32_t n; };
typedef mock_rsrc_t __amdgpu_buffer_rsrc_t;
struct mock_v2u32 { uint32_t v[2]; uint32_t operator[](int i) const { return v[i]; } };
static inline mock_v2u32 __builtin_amdgcn_raw_buffer_load_b64(mock_rsrc_t r, int voff, int soff, int aux)
{
	(void)aux; mock_v2u32 o; memset(&o, 0, sizeof o);
	const uint64_t off = (uint64_t)(uint32_t)voff + (uint32_t)soff;
	if (off + 8 <= r.n) memcpy(&o, r.p + off, 8);
	return o;
}
static inline mock_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void *p, short stride, int n, int flags) { (void)stride; (void)flags; mock_rsrc_t r; r.p = (const unsigned char*)p; r.n = (uint32_t)n; return r; }
static inline mock_v4u32 __builtin_amdgcn_raw_buffer_load_b128(mock_rsrc_t r, int voff, int soff, int aux)
{
	(void)aux; mock_v4u32 o; memset(&o, 0, sizeof o);
	const uint64_t off = (uint64_t)(uint32_t)voff + (uint32_t)soff;
	if (off + 16 <= r.n) memcpy(&o, r.p + off, 16);
	return o;
}
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type*)mock_dyn_lds;

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) mock_launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
