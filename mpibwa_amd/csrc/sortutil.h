// sortutil.h — the reference's sort is UNSTABLE and its tie order is visible in
// the SAM output (chains, regions, pair tables), so the exact comparison /
// swap sequence of ks_introsort (src/ksort.h:176-226: median-of-3 quicksort
// with an explicit stack, ranges of <= 16 left for a final insertion sort,
// comb sort when the depth budget 2*ceil(log2 n) runs out) is restated here.
#ifndef MBW_SORTUTIL_H
#define MBW_SORTUTIL_H
#include <cstddef>
#include <utility>
#include <vector>

namespace mbw {

template <class T, class Less>
inline void ks_insertion(T *s, T *t, Less lt)
{
	for (T *i = s + 1; i < t; ++i)
		for (T *j = i; j > s && lt(*j, *(j - 1)); --j) std::swap(*j, *(j - 1));
}

template <class T, class Less>
inline void ks_comb(size_t n, T *a, Less lt)
{
	const double shrink = 1.2473309501039786540366528676643;
	size_t gap = n;
	bool swapped;
	do {
		if (gap > 2) {
			gap = (size_t)(gap / shrink);
			if (gap == 9 || gap == 10) gap = 11;
		}
		swapped = false;
		for (T *i = a; i < a + n - gap; ++i) {
			T *j = i + gap;
			if (lt(*j, *i)) { std::swap(*i, *j); swapped = true; }
		}
	} while (swapped || gap > 2);
	if (gap != 1) ks_insertion(a, a + n, lt);
}

template <class T, class Less>
inline void ks_introsort(size_t n, T *a, Less lt)
{
	if (n < 1) return;
	if (n == 2) {
		if (lt(a[1], a[0])) std::swap(a[0], a[1]);
		return;
	}
	int d = 2;
	while ((1ul << d) < n) ++d;
	// the reference mallocs a stack of 8 d + 2 frames; only partitions of more than 16 elements are ever pushed and the
	// smaller side is always worked on first, so a few dozen frames cover any n — no heap traffic per sort
	struct Frame { T *left, *right; int depth; };
	struct FixedStack {
		Frame f[192];
		int n = 0;
		void push_back(const Frame &x) { f[n++] = x; }
		bool empty() const { return n == 0; }
		const Frame &back() const { return f[n - 1]; }
		void pop_back() { --n; }
	} stack;
	T *s = a, *t = a + (n - 1);
	d <<= 1;
	for (;;) {
		if (s < t) {
			if (--d == 0) {
				ks_comb((size_t)(t - s + 1), s, lt);
				t = s;
				continue;
			}
			T *i = s, *j = t, *k = i + ((j - i) >> 1) + 1;
			if (lt(*k, *i)) {
				if (lt(*k, *j)) k = j;
			} else k = lt(*j, *i) ? i : j;
			T pivot = *k;
			if (k != t) std::swap(*k, *t);
			for (;;) {
				do ++i; while (lt(*i, pivot));
				do --j; while (i <= j && lt(pivot, *j));
				if (j <= i) break;
				std::swap(*i, *j);
			}
			std::swap(*i, *t);
			if (i - s > t - i) {
				if (i - s > 16) stack.push_back({s, i - 1, d});
				s = t - i > 16 ? i + 1 : t;
			} else {
				if (t - i > 16) stack.push_back({i + 1, t, d});
				t = i - s > 16 ? i - 1 : s;
			}
		} else {
			if (stack.empty()) {
				ks_insertion(a, a + n, lt);
				return;
			}
			Frame f = stack.back();
			stack.pop_back();
			s = f.left; t = f.right; d = f.depth;
		}
	}
}

} // namespace mbw
#endif
