// Pins wut-cuda-orb-slam3_amd/csrc/libstdcxx_sort.h against the real libstdc++ std::sort /
// std::partial_sort on inputs with many comparator ties (the case DistributeOctTree makes observable,
// reference src/ORBextractor.cc:569-582, 708).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../wut-cuda-orb-slam3_amd/csrc/libstdcxx_sort.h"

struct E { int cnt, ulx, id; };
static bool lessE(const E &a, const E &b) { return a.cnt < b.cnt || (a.cnt == b.cnt && a.ulx < b.ulx); }

int main()
{
    std::mt19937 rng(12345);
    long checked = 0;
    for (int trial = 0; trial < 20000; ++trial) {
        int n = 1 + rng() % (trial % 7 == 0 ? 3000 : 200);
        int kc = 1 + rng() % 12, kx = 1 + rng() % 6;
        std::vector<E> a(n);
        int mode = trial % 5;
        for (int i = 0; i < n; ++i) {
            a[i] = {(int)(rng() % kc) + 2, (int)(rng() % kx) * 37, i};
            if (mode == 1) a[i].cnt = i % kc;             // sawtooth
            if (mode == 2) a[i].cnt = n - i, a[i].ulx = 0;  // descending
            if (mode == 3) a[i].cnt = 5, a[i].ulx = (i % 2) * 10;  // almost all ties
        }
        std::vector<E> b = a, c = a;
        std::sort(b.begin(), b.end(), lessE);
        orbx_sort::sort(c.data(), n, lessE);
        for (int i = 0; i < n; ++i)
            if (b[i].id != c[i].id) { printf("MISMATCH sort trial %d n %d at %d\n", trial, n, i); return 1; }
        {   // the data-parallel formulation the multi-wave quadtree kernel runs (pairing partition + stable rank)
            std::vector<E> g = a, tmp(n);
            std::vector<int> lp(n), rp(n);
            int stack[216];
            orbx_sort::sort_pairing(g.data(), n, lessE, stack, lp.data(), rp.data(), tmp.data());
            for (int i = 0; i < n; ++i)
                if (b[i].id != g[i].id) { printf("MISMATCH pairing trial %d n %d at %d\n", trial, n, i); return 1; }
        }
        // heap fallback: std::partial_sort(first, last, last) == heap_sort
        std::vector<E> d = a, e = a;
        std::partial_sort(d.begin(), d.end(), d.end(), lessE);
        orbx_sort::heap_sort(e.data(), n, lessE);
        for (int i = 0; i < n; ++i)
            if (d[i].id != e[i].id) { printf("MISMATCH heap trial %d n %d at %d\n", trial, n, i); return 1; }
        checked += n;
    }
    // median-of-3 killer (forces the depth limit -> heapsort path inside sort)
    for (int n : {64, 200, 1000, 5000}) {
        std::vector<E> a(n);
        int k = n / 2;
        for (int i = 1; i <= k; ++i) {
            if (i % 2 == 1) { a[i - 1].cnt = i; a[i].cnt = k + i; }
            a[k + i - 1].cnt = 2 * i;
        }
        for (int i = 0; i < n; ++i) { a[i].ulx = 0; a[i].id = i; }
        std::vector<E> b = a, c = a;
        std::sort(b.begin(), b.end(), lessE);
        orbx_sort::sort(c.data(), n, lessE);
        for (int i = 0; i < n; ++i)
            if (b[i].id != c[i].id) { printf("MISMATCH killer n %d at %d\n", n, i); return 1; }
        std::vector<E> g = a, tmp(n);
        std::vector<int> lp(n), rp(n);
        int stack[216];
        orbx_sort::sort_pairing(g.data(), n, lessE, stack, lp.data(), rp.data(), tmp.data());
        for (int i = 0; i < n; ++i)
            if (b[i].id != g[i].id) { printf("MISMATCH pairing killer n %d at %d\n", n, i); return 1; }
    }
    printf("OK %ld elements\n", checked);
    return 0;
}
