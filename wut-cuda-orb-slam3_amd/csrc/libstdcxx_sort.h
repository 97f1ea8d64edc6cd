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
//
// The algorithm is written against an accessor (get(i) / set(i, v)); the host tests and the kernels use the plain
// pointer accessor (LDS on the device).  An accessor that keeps the elements in lane registers (v_readlane /
// v_writelane with scalar indices) was measured and is no faster: a lone wave pays about as much for the ~30 scalar
// instructions of one element step as for one LDS round trip.  The sequence of comparisons and moves is identical
// for every accessor.
#ifndef ORBX_LIBSTDCXX_SORT_H
#define ORBX_LIBSTDCXX_SORT_H

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD inline
#endif

namespace orbx_sort {

template <class T>
struct PtrAccessor {
    T *p;
    ORBX_HD T get(int i) const { return p[i]; }
    ORBX_HD void set(int i, const T &v) const { p[i] = v; }
};

template <class A, class T, class Less>
ORBX_HD void push_heap(A &a, int first, int holeIndex, int topIndex, T value, Less less)
{
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex) {
        const T pv = a.get(first + parent);
        if (!less(pv, value)) break;
        a.set(first + holeIndex, pv);
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    a.set(first + holeIndex, value);
}

template <class A, class T, class Less>
ORBX_HD void adjust_heap(A &a, int first, int holeIndex, int len, T value, Less less)
{
    const int topIndex = holeIndex;
    int secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        T c1 = a.get(first + secondChild);
        const T c0 = a.get(first + secondChild - 1);
        if (less(c1, c0)) { secondChild--; c1 = c0; }
        a.set(first + holeIndex, c1);
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        a.set(first + holeIndex, a.get(first + secondChild - 1));
        holeIndex = secondChild - 1;
    }
    push_heap(a, first, holeIndex, topIndex, value, less);
}

// std::__partial_sort(first, last, last): make_heap + sort_heap (the heap_select loop over [middle,last) is empty)
template <class T, class A, class Less>
ORBX_HD void heap_sort_acc(A &a, int first, int len, Less less)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        for (;;) {
            const T value = a.get(first + parent);
            adjust_heap(a, first, parent, len, value, less);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        const T value = a.get(first + last);
        a.set(first + last, a.get(first));
        adjust_heap(a, first, 0, last, value, less);
    }
}

template <class T, class A, class Less>
ORBX_HD void unguarded_linear_insert(A &a, int last, Less less)
{
    const T val = a.get(last);
    int next = last - 1;
    for (;;) {
        const T nv = a.get(next);
        if (!less(val, nv)) break;
        a.set(last, nv);
        last = next;
        --next;
    }
    a.set(last, val);
}

template <class T, class A, class Less>
ORBX_HD void insertion_sort(A &a, int first, int last, Less less)
{
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        const T vi = a.get(i);
        if (less(vi, a.get(first))) {
            for (int k = i; k > first; --k) a.set(k, a.get(k - 1));
            a.set(first, vi);
        } else {
            unguarded_linear_insert<T>(a, i, less);
        }
    }
}

// std::sort over elements [0, n) of the accessor — result (including the order of equivalent elements) identical to libstdc++.
// `stack` = caller-provided storage for 3 * 72 ints (the explicit recursion stack); on the device it lives in LDS so
// that no dynamically indexed private array (scratch memory) sits in the middle of the latency chain.
template <class T, class A, class Less>
ORBX_HD void sort_acc(A &a, int n, Less less, int *stack)
{
    if (n <= 1) return;
    const int kThreshold = 16;
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) ++lg;  // std::__lg
    // explicit stack of (first, last, depth_limit); depth <= 2*lg + 1 entries
    int *stFirst = stack, *stLast = stack + 72, *stDepth = stack + 144;
    int sp = 0;
    stFirst[0] = 0; stLast[0] = n; stDepth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stFirst[sp], last = stLast[sp], depth = stDepth[sp];
        while (last - first > kThreshold) {
            if (depth == 0) {
                heap_sort_acc<T>(a, first, last - first, less);
                break;
            }
            --depth;
            // __unguarded_partition_pivot: median of (first+1, mid, last-1) moved to first
            const int mid = first + (last - first) / 2;
            const int ia = first + 1, ib = mid, ic = last - 1;
            const T vf = a.get(first), va = a.get(ia), vb = a.get(ib), vc = a.get(ic);
            int im;  // index whose element is swapped with *first
            if (less(va, vb)) {
                if (less(vb, vc)) im = ib;
                else if (less(va, vc)) im = ic;
                else im = ia;
            } else if (less(va, vc)) im = ia;
            else if (less(vb, vc)) im = ic;
            else im = ib;
            const T pivot = im == ia ? va : (im == ib ? vb : vc);
            a.set(im, vf);
            a.set(first, pivot);
            // __unguarded_partition(first + 1, last, pivot = *first); *first is not touched inside the loop
            int lo = first + 1, hi = last;
            for (;;) {
                T vlo = a.get(lo);
                while (less(vlo, pivot)) { ++lo; vlo = a.get(lo); }
                --hi;
                T vhi = a.get(hi);
                while (less(pivot, vhi)) { --hi; vhi = a.get(hi); }
                if (!(lo < hi)) break;
                a.set(lo, vhi);
                a.set(hi, vlo);
                ++lo;
            }
            const int cut = lo;
            // recurse on [cut, last) with the decremented depth, loop on [first, cut)
            stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; ++sp;
            last = cut;
        }
    }
    // __final_insertion_sort
    if (n > kThreshold) {
        insertion_sort<T>(a, 0, kThreshold, less);
        for (int i = kThreshold; i != n; ++i) unguarded_linear_insert<T>(a, i, less);
    } else {
        insertion_sort<T>(a, 0, n, less);
    }
}

template <class T, class Less>
ORBX_HD void heap_sort(T *p, int n, Less less);

// ---- the same result, data-parallel ------------------------------------------------------------------------------------------
// A formulation whose steps are independent over the elements (k_quadtree_mw runs them on the lanes of a wave / the threads of a
// workgroup; this plain C++ statement of it is what the host test pins against std::sort):
//   * __unguarded_partition(first + 1, last, pivot): the k-th swap exchanges the k-th "left stopper" (position, ascending, whose
//     element is not less than the pivot) with the k-th "right stopper" (position, descending, whose element the pivot is not
//     less than); K = number of k with L_k < R_k (a prefix: L ascends, R descends).  The scan that ends the partition stops at
//     the next original left stopper or at the position of the last swap's right element, whichever comes first:
//     cut = (K < nL && L_K < Rprev) ? L_K : Rprev with Rprev = K ? R_{K-1} : last.
//   * __final_insertion_sort is an insertion sort, i.e. a STABLE sort of whatever arrangement the introsort loop leaves:
//     rank(i) = #{j : e_j < e_i} + #{j < i : e_j equivalent to e_i}.
// `lp`, `rp`: scratch for n positions each; `tmp`: scratch for n elements.
template <class T, class Less>
inline void sort_pairing(T *p, int n, Less less, int *stack, int *lp, int *rp, T *tmp)
{
    if (n <= 1) return;
    const int kThreshold = 16;
    int lg = 0;
    for (unsigned v = (unsigned)n; v > 1; v >>= 1) ++lg;
    int *stFirst = stack, *stLast = stack + 72, *stDepth = stack + 144;
    int sp = 0;
    stFirst[0] = 0; stLast[0] = n; stDepth[0] = lg * 2; sp = 1;
    while (sp > 0) {
        --sp;
        int first = stFirst[sp], last = stLast[sp], depth = stDepth[sp];
        while (last - first > kThreshold) {
            if (depth == 0) { heap_sort(p + first, last - first, less); break; }
            --depth;
            const int mid = first + (last - first) / 2;
            const int ia = first + 1, ib = mid, ic = last - 1;
            const T vf = p[first], va = p[ia], vb = p[ib], vc = p[ic];
            int im;
            if (less(va, vb)) {
                if (less(vb, vc)) im = ib;
                else if (less(va, vc)) im = ic;
                else im = ia;
            } else if (less(va, vc)) im = ia;
            else if (less(vb, vc)) im = ic;
            else im = ib;
            const T pivot = im == ia ? va : (im == ib ? vb : vc);
            p[im] = vf;
            p[first] = pivot;
            int nL = 0, nR = 0;
            for (int q = first + 1; q < last; ++q) {  // independent per position
                if (!less(p[q], pivot)) lp[nL++] = q;
                if (!less(pivot, p[q])) rp[nR++] = q;  // ascending: R_k = rp[nR - 1 - k]
            }
            int K = 0;
            while (K < nL && K < nR && lp[K] < rp[nR - 1 - K]) ++K;  // (a prefix: evaluated for all k at once on the device)
            for (int k = 0; k < K; ++k) { const T t = p[lp[k]]; p[lp[k]] = p[rp[nR - 1 - k]]; p[rp[nR - 1 - k]] = t; }
            const int rprev = K > 0 ? rp[nR - K] : last;
            const int cut = (K < nL && lp[K] < rprev) ? lp[K] : rprev;
            stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; ++sp;
            last = cut;
        }
    }
    for (int i = 0; i < n; ++i) {  // independent per element
        int r = 0;
        for (int j = 0; j < n; ++j) r += (less(p[j], p[i]) || (!less(p[i], p[j]) && j < i)) ? 1 : 0;
        tmp[r] = p[i];
    }
    for (int i = 0; i < n; ++i) p[i] = tmp[i];
}

// pointer convenience wrappers (host tests, LDS fallback)
template <class T, class Less>
ORBX_HD void sort(T *p, int n, Less less, int *stack)
{
    PtrAccessor<T> a{p};
    sort_acc<T>(a, n, less, stack);
}
template <class T, class Less>
inline void sort(T *p, int n, Less less)  // host convenience
{
    int stack[216];
    sort(p, n, less, stack);
}
template <class T, class Less>
ORBX_HD void heap_sort(T *p, int n, Less less)
{
    PtrAccessor<T> a{p};
    heap_sort_acc<T>(a, 0, n, less);
}

}  // namespace orbx_sort
#endif
