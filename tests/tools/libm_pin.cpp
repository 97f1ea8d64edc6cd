// libm_pin.cpp - pins include/orbx_libm.h (what the GPU evaluates for logf / atanf / atan2f / tanf) against the REAL libm of the host.
//
//   g++ -O2 -std=c++17 -ffp-contract=off -pthread -I include tests/tools/libm_pin.cpp -o /tmp/libm_pin -lm
//   /tmp/libm_pin [stride] [atan2_pairs_millions] [threads]
//
// stride 1 = exhaustive: logf on every positive float (incl. subnormals, +0, inf), atanf on every float, tanf on every float; atan2f on `pairs` million random pairs per shape (uniform bit patterns; the KannalaBrandt8 shapes
// (sqrtf(x^2 + y^2), z) and (y, x) with camera-frame coordinates; near-equal exponents; the special-case grid).  A larger stride visits
// every stride-th bit pattern (the CPU test suite runs stride 509).  Prints one line per function and exits non-zero on any mismatch.
// NaN results compare equal to NaN results (payloads are not part of the contract).
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "orbx_libm.h"

static inline bool same(float a, float b)
{
    if (a != a && b != b) return true;
    return orbx_f2u(a) == orbx_f2u(b);
}

struct Rng {  // SplitMix64
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    float unit() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
};

template <class F>
static uint64_t sweep(uint64_t lo, uint64_t hi, uint64_t stride, int threads, F f, uint64_t *firstBad)
{
    std::atomic<uint64_t> bad{0}, first{~0ull};
    std::vector<std::thread> th;
    const uint64_t chunk = (hi - lo + threads - 1) / threads;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([&, t] {
            uint64_t b = 0, fb = ~0ull;
            const uint64_t a0 = lo + t * chunk, a1 = std::min(hi, a0 + chunk);
            uint64_t u = a0 + (stride - a0 % stride) % stride;
            for (; u < a1; u += stride)
                if (!f((uint32_t)u)) { ++b; if (fb == ~0ull) fb = u; }
            bad += b;
            uint64_t cur = first.load();
            while (fb < cur && !first.compare_exchange_weak(cur, fb)) {}
        });
    for (auto &t : th) t.join();
    *firstBad = first.load();
    return bad.load();
}

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const uint64_t pairsM = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1000;
    const int threads = argc > 3 ? atoi(argv[3]) : (int)std::max(1u, std::thread::hardware_concurrency());
    int rc = 0;
    uint64_t fb, bad;

    // logf: +0 .. +inf inclusive, and a sample of negative / NaN patterns
    bad = sweep(0, 0x7f800001ull, stride, threads, [](uint32_t u) { const float x = orbx_u2f(u); return same(orbx_glibc_logf(x), logf(x)); }, &fb);
    printf("logf   positive floats 0x00000000..0x7f800000 stride %llu: %llu mismatches", (unsigned long long)stride, (unsigned long long)bad);
    if (bad) { const float x = orbx_u2f((uint32_t)fb); printf(" (first 0x%08x: header %a libm %a)", (uint32_t)fb, orbx_glibc_logf(x), logf(x)); rc = 1; }
    printf("\n");
    bad = sweep(0x7f800001ull, 0x100000000ull, stride * 64 + 1, threads, [](uint32_t u) { const float x = orbx_u2f(u); return same(orbx_glibc_logf(x), logf(x)); }, &fb);
    printf("logf   NaN / negative patterns (sampled): %llu mismatches\n", (unsigned long long)bad);
    if (bad) rc = 1;

    bad = sweep(0, 0x100000000ull, stride, threads, [](uint32_t u) { const float x = orbx_u2f(u); return same(orbx_glibc_atanf(x), atanf(x)); }, &fb);
    printf("atanf  all floats stride %llu: %llu mismatches", (unsigned long long)stride, (unsigned long long)bad);
    if (bad) { const float x = orbx_u2f((uint32_t)fb); printf(" (first 0x%08x: header %a libm %a)", (uint32_t)fb, orbx_glibc_atanf(x), atanf(x)); rc = 1; }
    printf("\n");

    auto tanOk = [](uint32_t u) {
        const float x = orbx_u2f(u);
        return same(orbx_glibc_tanf(x), tanf(x));
    };
    bad = sweep(0, 0x100000000ull, stride, threads, tanOk, &fb);
    printf("tanf   all floats stride %llu: %llu mismatches", (unsigned long long)stride, (unsigned long long)bad);
    if (bad) { const float x = orbx_u2f((uint32_t)fb); printf(" (first 0x%08x: header %a libm %a)", (uint32_t)fb, orbx_glibc_tanf(x), tanf(x)); rc = 1; }
    printf("\n");

    // atan2f: pairs
    const uint64_t perThread = pairsM * 1000000ull / threads;
    const char *shapes[] = {"uniform bit patterns", "KB8 theta: (sqrtf(x*x + y*y), z), camera-frame points", "KB8 psi: (y, x)", "exponents within +-3, random mantissas",
                            "exponent difference 55..66 (the k > 60 / k < -60 branches)"};
    for (int shape = 0; shape < 5; ++shape) {
        std::atomic<uint64_t> nbad{0};
        float by = 0, bx = 0;
        std::vector<std::thread> th;
        for (int t = 0; t < threads; ++t)
            th.emplace_back([&, t] {
                Rng g{0x0123456789ABCDEFull * (uint64_t)(shape + 1) + (uint64_t)t * 0x1000003ull};
                uint64_t b = 0;
                for (uint64_t i = 0; i < perThread; ++i) {
                    float y, x;
                    const uint64_t r = g.next();
                    if (shape == 0) { y = orbx_u2f((uint32_t)r); x = orbx_u2f((uint32_t)(r >> 32)); }
                    else if (shape == 1 || shape == 2) {
                        const float range = (i & 1) ? 3.0f : 40.0f;
                        const float px = (g.unit() - 0.5f) * range, py = (g.unit() - 0.5f) * range, pz = (g.unit() - ((i & 2) ? 0.5f : 0.02f)) * range;
                        if (shape == 1) { y = sqrtf(px * px + py * py); x = pz; } else { y = py; x = px; }
                    } else if (shape == 3) {
                        const uint32_t e = 1 + (uint32_t)(r % 250);
                        y = orbx_u2f(((uint32_t)(r >> 8) & 0x807fffffu) | (e << 23));
                        x = orbx_u2f(((uint32_t)(r >> 32) & 0x807fffffu) | ((e + (uint32_t)((r >> 20) % 7) - 3 + 1) << 23));
                    } else {
                        const uint32_t d = 55 + (uint32_t)(r % 12), e = 1 + (uint32_t)((r >> 8) % (253 - d));
                        const uint32_t ey = (r & (1ull << 63)) ? e + d : e, ex = (r & (1ull << 63)) ? e : e + d;
                        y = orbx_u2f(((uint32_t)(r >> 8) & 0x807fffffu) | (ey << 23));
                        x = orbx_u2f(((uint32_t)(r >> 32) & 0x807fffffu) | (ex << 23));
                    }
                    if (!same(orbx_glibc_atan2f(y, x), atan2f(y, x))) { if (!b) { by = y; bx = x; } ++b; }
                }
                nbad += b;
            });
        for (auto &t : th) t.join();
        printf("atan2f %llu pairs, %s: %llu mismatches", (unsigned long long)(perThread * threads), shapes[shape], (unsigned long long)nbad.load());
        if (nbad.load()) { printf(" (e.g. y %a x %a: header %a libm %a)", by, bx, orbx_glibc_atan2f(by, bx), atan2f(by, bx)); rc = 1; }
        printf("\n");
    }
    {   // special-case grid: every combination of +-0, +-tiny, +-1, +-huge, +-inf, NaN, x = 1 exactly with a sweep of y
        const float sp[] = {0.0f, -0.0f, 1e-45f, -1e-45f, 1.17549435e-38f, -1.17549435e-38f, 1.0f, -1.0f, 0.5f, -0.5f, 2.0f, -2.0f, 3.4e38f, -3.4e38f,
                            INFINITY, -INFINITY, NAN, 1e-30f, -1e-30f, 1e30f, -1e30f};
        uint64_t b = 0, n = 0;
        for (float y : sp) for (float x : sp) { ++n; if (!same(orbx_glibc_atan2f(y, x), atan2f(y, x))) ++b; }
        for (uint64_t u = 0; u < 0x100000000ull; u += 65537 * stride) { const float y = orbx_u2f((uint32_t)u); ++n; if (!same(orbx_glibc_atan2f(y, 1.0f), atan2f(y, 1.0f))) ++b;
            if (!same(orbx_glibc_atan2f(1.0f, y), atan2f(1.0f, y))) ++b; }
        printf("atan2f special-case grid + (y, 1) / (1, x) sweeps, %llu cases: %llu mismatches\n", (unsigned long long)n, (unsigned long long)b);
        if (b) rc = 1;
    }
    printf(rc ? "FAILED\n" : "OK: include/orbx_libm.h equals this host's libm on everything visited\n");
    return rc;
}
