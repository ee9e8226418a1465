/* oracle/orc_sort.c -- TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * Restatement of klib's ks_introsort (ksort.h:176-226) as one byte-wise generic routine.  BWA-MEM
 * sorts chains and alignment regions with keys that tie (bwamem.c:367,467,504), and the sort is not
 * stable, so the *exact* sequence of comparisons and swaps decides which chain or hit survives:
 *   - n == 2 is a single compare-and-swap;
 *   - otherwise quicksort with a median-of-three pivot taken from (first, middle+1, last), pivot
 *     moved to the right end, Hoare-style scan, sub-ranges of <= 16 elements left unsorted, the
 *     larger side pushed on an explicit stack, depth limit 2*ceil(log2 n) after which the range is
 *     comb-sorted (ksort.h:152-175, shrink factor 1.2473309501039786540366528676643, gap 9/10 -> 11);
 *   - one final insertion sort over the whole array (ksort.h:143-151).
 */
#include <stdlib.h>
#include <string.h>
#include "orc.h"

#define EL(i) (b + (size_t)(i) * es)

static void swp(char *x, char *y, size_t es, char *tmp) { memcpy(tmp, x, es); memcpy(x, y, es); memcpy(y, tmp, es); }

static void insertion(char *b, long lo, long hi /* exclusive */, size_t es, orc_lt_f lt, char *tmp)
{
	long i, j;
	for (i = lo + 1; i < hi; ++i)
		for (j = i; j > lo && lt(EL(j), EL(j-1)); --j) swp(EL(j), EL(j-1), es, tmp);
}

static void combsort(char *b, long n, size_t es, orc_lt_f lt, char *tmp)
{
	const double shrink = 1.2473309501039786540366528676643;
	long gap = n, i; int swapped;
	do {
		if (gap > 2) {
			gap = (long)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = 0;
		for (i = 0; i + gap < n; ++i)
			if (lt(EL(i + gap), EL(i))) { swp(EL(i), EL(i + gap), es, tmp); swapped = 1; }
	} while (swapped || gap > 2);
	if (gap != 1) insertion(b, 0, n, es, lt, tmp);
}

void orc_introsort(void *base, size_t n_, size_t es, orc_lt_f lt)
{
	char *b = (char*)base, *tmp, *piv;
	long n = (long)n_, s, t, i, j, k;
	struct { long l, r; int d; } stack[128];
	int top = 0, d;
	if (n < 1) return;
	tmp = (char*)malloc(2 * es); piv = tmp + es;
	if (n == 2) {
		if (lt(EL(1), EL(0))) swp(EL(0), EL(1), es, tmp);
		free(tmp); return;
	}
	for (d = 2; (1ul << d) < (unsigned long)n; ++d) {}
	d <<= 1;
	s = 0; t = n - 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) { combsort(EL(s), t - s + 1, es, lt, tmp); t = s; continue; }
			i = s; j = t; k = i + ((j - i) >> 1) + 1;
			if (lt(EL(k), EL(i))) { if (lt(EL(k), EL(j))) k = j; }
			else k = lt(EL(j), EL(i)) ? i : j;
			memcpy(piv, EL(k), es);
			if (k != t) swp(EL(k), EL(t), es, tmp);
			for (;;) {
				do ++i; while (lt(EL(i), piv));
				do --j; while (i <= j && lt(piv, EL(j)));
				if (j <= i) break;
				swp(EL(i), EL(j), es, tmp);
			}
			swp(EL(i), EL(t), es, tmp);
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
	insertion(b, 0, n, es, lt, tmp);
	free(tmp);
}
