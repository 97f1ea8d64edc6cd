// libstdcxx_sort.h — step-exact emulation of libstdc++'s std::sort (bits/stl_algo.h: introsort loop with
// median-of-3 pivot to 16-element runs, heapsort fallback at depth 2*floor(log2 n), final insertion sort)
// that compiles for both host and gfx950 device code.
//
// Why: ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:708) sorts (count, node) pairs with
// a comparator (compareNodes, :569-582) that is a strict weak order on (count, UL.x) only.  Nodes tying on
// both keys leave std::sort in libstdc++'s *unstable* order, and the early `break` at :745-746 makes that
// order observable in the final keypoint set.  The quadtree kernel therefore
// reproduces the exact permutation instead of "a" sorted order.  No recursion: an explicit stack of
// sub-ranges (the sub-ranges are disjoint, so processing order does not change the result).
#ifndef ORBX_LIBSTDCXX_SORT_H
#define ORBX_LIBSTDCXX_SORT_H

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD inline
#endif

namespace orbx_sort {

template <class T, class Less>
ORBX_HD void push_heap(T *first, int holeIndex, int topIndex, T value, Less less)
{
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && less(first[parent], value)) {
        first[holeIndex] = first[parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    first[holeIndex] = value;
}

template <class T, class Less>
ORBX_HD void adjust_heap(T *first, int holeIndex, int len, T value, Less less)
{
    const int topIndex = holeIndex;
    int secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (less(first[secondChild], first[secondChild - 1])) secondChild--;
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    push_heap(first, holeIndex, topIndex, value, less);
}

// std::__partial_sort(first, last, last): make_heap + sort_heap (the heap_select loop over [middle,last) is empty)
template <class T, class Less>
ORBX_HD void heap_sort(T *first, int len, Less less)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            T value = first[parent];
            adjust_heap(first, parent, len, value, less);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        T value = first[last];
        first[last] = first[0];
        adjust_heap(first, 0, last, value, less);
    }
}

template <class T>
ORBX_HD void swap_elems(T &a, T &b)
{
    T t = a;
    a = b;
    b = t;
}

template <class T, class Less>
ORBX_HD void unguarded_linear_insert(T *a, int last, Less less)
{
    T val = a[last];
    int next = last - 1;
    while (less(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

template <class T, class Less>
ORBX_HD void insertion_sort(T *a, int first, int last, Less less)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (less(a[i], a[first])) {
            T val = a[i];
            for (int k = i; k > first; --k) a[k] = a[k - 1];
            a[first] = val;
        } else {
            unguarded_linear_insert(a, i, less);
        }
    }
}

// std::sort(a, a + n, less) — result (including the order of equivalent elements) identical to libstdc++.
template <class T, class Less>
ORBX_HD void sort(T *a, int n, Less less)
{
    if (n <= 1) return;
    const int kThreshold = 16;
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) ++lg;  // std::__lg
    // explicit stack of (first, last, depth_limit); depth <= 2*lg + 1 entries
    int stFirst[72], stLast[72], stDepth[72];
    int sp = 0;
    stFirst[0] = 0; stLast[0] = n; stDepth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stFirst[sp], last = stLast[sp], depth = stDepth[sp];
        while (last - first > kThreshold) {
            if (depth == 0) {
                heap_sort(a + first, last - first, less);
                break;
            }
            --depth;
            // __unguarded_partition_pivot: median of (first+1, mid, last-1) moved to first
            int mid = first + (last - first) / 2;
            int ia = first + 1, ib = mid, ic = last - 1;
            if (less(a[ia], a[ib])) {
                if (less(a[ib], a[ic])) swap_elems(a[first], a[ib]);
                else if (less(a[ia], a[ic])) swap_elems(a[first], a[ic]);
                else swap_elems(a[first], a[ia]);
            } else if (less(a[ia], a[ic])) swap_elems(a[first], a[ia]);
            else if (less(a[ib], a[ic])) swap_elems(a[first], a[ic]);
            else swap_elems(a[first], a[ib]);
            // __unguarded_partition(first + 1, last, pivot = first)
            int lo = first + 1, hi = last;
            for (;;) {
                while (less(a[lo], a[first])) ++lo;
                --hi;
                while (less(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                swap_elems(a[lo], a[hi]);
                ++lo;
            }
            int cut = lo;
            // recurse on [cut, last) with the decremented depth, loop on [first, cut)
            stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; ++sp;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > kThreshold) {
        insertion_sort(a, 0, kThreshold, less);
        for (int i = kThreshold; i != n; ++i) unguarded_linear_insert(a, i, less);
    } else {
        insertion_sort(a, 0, n, less);
    }
}

}  // namespace orbx_sort
#endif
