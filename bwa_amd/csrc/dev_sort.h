// dev_sort.h -- device restatement of klib's ks_introsort (ksort.h:176-226) for one lane.
//
// BWA-MEM's results depend on the exact tie behaviour of this unstable sort (chains by weight bwamem.c:367,
// regions by end bwamem.c:467 and by (score,rb,qb) bwamem.c:504), so the comparison/swap sequence is
// reproduced step for step: single compare for n == 2; otherwise quicksort with the median of
// (first, mid+1, last) moved to the right end, Hoare scan, ranges <= 16 left for one final insertion sort,
// explicit stack with the larger side pushed, depth limit 2*ceil(log2 n) after which a range is comb-sorted
// (ksort.h:152-175).
#pragma once
#include "dev_common.h"

template <class T, class LT>
DEVFN void dev_insertion(T *a, int lo, int hi, LT lt)
{
	for (int i = lo + 1; i < hi; ++i)
		for (int j = i; j > lo && lt(a[j], a[j - 1]); --j) { T t = a[j]; a[j] = a[j - 1]; a[j - 1] = t; }
}

template <class T, class LT>
DEVFN void dev_combsort(T *a, int n, LT lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	int gap = n; bool swapped;
	do {
		if (gap > 2) {
			gap = (int)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = false;
		for (int i = 0; i + gap < n; ++i)
			if (lt(a[i + gap], a[i])) { T t = a[i]; a[i] = a[i + gap]; a[i + gap] = t; swapped = true; }
	} while (swapped || gap > 2);
	if (gap != 1) dev_insertion(a, 0, n, lt);
}

// FINISH = false: without the final insertion sort; the caller finishes with any STABLE sort of what the quicksort passes leave (the insertion
// sort is one, and the result of a stable sort does not depend on the method: a wave does it in parallel, dev_chainw.h).  Note that those passes
// do not leave tidy blocks: the scan from the left starts at the range's SECOND element, so the first one can stay on the wrong side of the pivot
// until the insertion sort carries it home -- the finishing sort has to be a full one.
template <class T, class LT, bool FINISH = true>
__device__ void dev_introsort(T *a, int n, LT lt)
{
	struct Frame { int l, r, d; } stack[40]; // the smaller side is iterated, the larger pushed: depth <= log2 n
	int top = 0, d;
	if (n < 1) return;
	if (n == 2) { if (lt(a[1], a[0])) { T t = a[0]; a[0] = a[1]; a[1] = t; } return; }
	for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
	d <<= 1;
	int s = 0, t = n - 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { dev_combsort(a + s, t - s + 1, lt); t = s; continue; }
			int i = s, j = t, k = i + ((j - i) >> 1) + 1;
			if (lt(a[k], a[i])) { if (lt(a[k], a[j])) k = j; }
			else k = lt(a[j], a[i]) ? i : j;
			T piv = a[k];
			if (k != t) { T x = a[k]; a[k] = a[t]; a[t] = x; }
			for (;;) {
				do ++i; while (lt(a[i], piv));
				do --j; while (i <= j && lt(piv, a[j]));
				if (j <= i) break;
				T x = a[i]; a[i] = a[j]; a[j] = x;
			}
			{ T x = a[i]; a[i] = a[t]; a[t] = x; }
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
	if (FINISH) dev_insertion(a, 0, n, lt);
}
