// tests/hostsim/rocprim/functional.hpp -- TEST INFRASTRUCTURE ONLY: CPU stand-ins for the three rocPRIM device primitives
// bwagpu_index.hip calls (radix_sort_pairs, inclusive_scan, exclusive_scan), so that the unmodified index-builder source can
// run under the mock HIP runtime.  Semantics follow rocPRIM's: a call with a null temporary-storage pointer only reports
// the size; the sort is stable and compares the key bits [begin_bit, end_bit) only.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <numeric>
#include <vector>
namespace rocprim {
template <class T> struct maximum { T operator()(const T &a, const T &b) const { return a < b ? b : a; } };
template <class T> struct plus { T operator()(const T &a, const T &b) const { return a + b; } };
template <class K, class V>
hipError_t radix_sort_pairs(void *tmp, size_t &bytes, K *ki, K *ko, V *vi, V *vo, size_t n, unsigned bb, unsigned eb, hipStream_t)
{
	if (!tmp) { bytes = 16; return hipSuccess; }
	const K mask = (eb - bb >= sizeof(K) * 8) ? ~(K)0 : (((K)1 << (eb - bb)) - 1);
	std::vector<size_t> idx(n);
	std::iota(idx.begin(), idx.end(), (size_t)0);
	std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((ki[a] >> bb) & mask) < ((ki[b] >> bb) & mask); });
	for (size_t i = 0; i < n; ++i) { ko[i] = ki[idx[i]]; vo[i] = vi[idx[i]]; }
	return hipSuccess;
}
template <class T, class Op>
hipError_t inclusive_scan(void *tmp, size_t &bytes, T *in, T *out, size_t n, Op op, hipStream_t)
{
	if (!tmp) { bytes = 16; return hipSuccess; }
	T acc = T();
	for (size_t i = 0; i < n; ++i) { acc = i ? op(acc, in[i]) : in[i]; out[i] = acc; }
	return hipSuccess;
}
template <class T, class Op>
hipError_t exclusive_scan(void *tmp, size_t &bytes, T *in, T *out, T init, size_t n, Op op, hipStream_t)
{
	if (!tmp) { bytes = 16; return hipSuccess; }
	T acc = init;
	for (size_t i = 0; i < n; ++i) { const T v = in[i]; out[i] = acc; acc = op(acc, v); }
	return hipSuccess;
}
}
