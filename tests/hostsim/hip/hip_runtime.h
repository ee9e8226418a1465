// tests/hostsim/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A minimal *mock* of the HIP runtime so that the unmodified product source bwa_amd/csrc/bwagpu.hip can be
// compiled with g++ and exercised by the CPU-only test-suite (this container has no GPU).  Kernels are run
// serially: hipLaunchKernelGGL loops over the grid and block and sets threadIdx/blockIdx for each "lane".
// That is valid for this code base because its v1 kernels are lane-serial (no __syncthreads, no shuffles).
// The mock is never part of the product: libbwagpu.so is built by hipcc against the real runtime and has no
// CPU path.  The library built here is tests/hostsim/libbwagpu_hostsim.so.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint4 { uint32_t x, y, z, w; };
extern thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorMock = 1 };
typedef struct mock_stream_s *hipStream_t;
struct mock_event_s { std::chrono::steady_clock::time_point t; };
typedef mock_event_s *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };

static inline const char *hipGetErrorString(hipError_t) { return "mock hip error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new mock_event_s(); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorMock; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p += v; return o; }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p |= v; return o; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) do { \
		dim3 g_ = (grid), b_ = (block); gridDim = g_; blockDim = b_; \
		for (unsigned bx_ = 0; bx_ < g_.x; ++bx_) for (unsigned tx_ = 0; tx_ < b_.x; ++tx_) { \
			blockIdx = dim3(bx_, 0, 0); threadIdx = dim3(tx_, 0, 0); kernel(__VA_ARGS__); } \
	} while (0)
