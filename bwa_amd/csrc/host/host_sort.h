// host_sort.h -- ks_introsort (ksort.h:176-226) restated for the host finalize code: the tie behaviour of this unstable
// sort decides the order of equal-score hits in the SAM output (bwamem.c:423-426, utils.c:46-47).
#pragma once
#include <cstddef>
#include <utility>

namespace hostmem {

template <class T, class LT> static inline void ins_sort(T *a, long lo, long hi, LT lt)
{
	for (long i = lo + 1; i < hi; ++i)
		for (long j = i; j > lo && lt(a[j], a[j - 1]); --j) std::swap(a[j], a[j - 1]);
}

template <class T, class LT> static void comb_sort(T *a, long n, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	long gap = n; bool swapped;
	do {
		if (gap > 2) { gap = (long)(gap / shrink); if (gap == 9 || gap == 10) gap = 11; }
		swapped = false;
		for (long i = 0; i + gap < n; ++i) if (lt(a[i + gap], a[i])) { std::swap(a[i], a[i + gap]); swapped = true; }
	} while (swapped || gap > 2);
	if (gap != 1) ins_sort(a, 0, n, lt);
}

template <class T, class LT> void introsort(T *a, long n, LT lt)
{
	struct Frame { long l, r; int d; } stack[128];
	int top = 0, d;
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) std::swap(a[0], a[1]); return; }
	for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
	d <<= 1;
	long s = 0, t = n - 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { comb_sort(a + s, t - s + 1, lt); t = s; continue; }
			long i = s, j = t, k = i + ((j - i) >> 1) + 1;
			if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
			else k = lt(a[j], a[i]) ? i : j;
			T piv = a[k];
			if (k != t) std::swap(a[k], a[t]);
			for (;;) {
				do ++i; while (lt(a[i], piv));
				do --j; while (i <= j && lt(piv, a[j]));
				if (j <= i) break;
				std::swap(a[i], a[j]);
			}
			std::swap(a[i], a[t]);
			if (i - s > t - i) {
				if (i - s > 16) { stack[top].l = s; stack[top].r = i - 1; stack[top].d = d; ++top; }
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) { stack[top].l = i + 1; stack[top].r = t; stack[top].d = d; ++top; }
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (top == 0) break;
			--top; s = stack[top].l; t = stack[top].r; d = stack[top].d;
		}
	}
	ins_sort(a, 0, n, lt);
}

}  // namespace hostmem
