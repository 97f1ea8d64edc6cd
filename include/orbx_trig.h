/*
 * orbx_trig.h — glibc >= 2.28 cosf / sinf, restated (header-only, C99 / C++ / HIP device code).
 *
 * computeOrbDescriptor calls `cos(angle)` / `sin(angle)` on a float, unqualified (reference src/ORBextractor.cc:108).  When the
 * float overload is chosen (orbx_trig_mode 1) the reference's result is whatever the platform's libm returns for cosf / sinf;
 * on glibc that is the double-precision evaluation of sysdeps/ieee754/flt-32/s_cosf.c / s_sinf.c / sincosf.h (one-multiply range
 * reduction to [-pi/4, pi/4], odd / even minimax polynomial, one rounding to float).  This header evaluates the same arithmetic
 * for 0 <= y < 120 so that the GPU produces the same floats; tests/test_oracle_kat.py compiles it on the host and compares it
 * with the real cosf / sinf on every float in [0, 6.5] (the angle in radians is in [0, 2 pi]).  Un-fused and fused evaluation of
 * the double operations give identical floats on that range (both checked).
 */
#ifndef ORBX_TRIG_H
#define ORBX_TRIG_H
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD static inline
#endif

/* want_cos != 0: cosf(y); else sinf(y).  0 <= y < 120. */
ORBX_HD float orbx_glibc_sincosf(float y, int want_cos)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23; /* 2 / pi * 2^24 */
    const double hpi = 0x1.921FB54442D18p0;       /* pi / 2 */
    const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                 C4 = 0x1.99343027bf8c3p-16;
    const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
    uint32_t bits;
    memcpy(&bits, &y, sizeof bits);
    const uint32_t top = (bits >> 20) & 0x7ff;    /* abstop12 */
    double x = (double)y, sgn = 1.0;
    int n = 0;
    if (top < 0x3f4) {                            /* |y| < pi / 4 */
        if (top < 0x398) return want_cos ? 1.0f : y; /* |y| < 2^-12 */
    } else {                                      /* reduce_fast */
        const double r = x * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;        /* quadrant, rounded to nearest */
        x = x - (double)n * hpi;
        sgn = ((n + 1) & 2) ? -1.0 : 1.0;         /* sign[n & 3] = {1, -1, -1, 1} */
    }
    const int q = want_cos ? (n ^ 1) : n;         /* odd: the cosine polynomial */
    const double pc = (n & 2) ? -1.0 : 1.0;       /* __sincosf_table[1]: the cosine polynomial negated */
    const double xs = x * sgn, x2 = x * x;
    if ((q & 1) == 0) {
        const double x3 = xs * x2, s1 = S2 + x2 * S3, x7 = x3 * x2, s = xs + x3 * S1;
        return (float)(s + x7 * s1);
    } else {
        const double x4 = x2 * x2, c2 = pc * C3 + x2 * (pc * C4), c1 = pc * C0 + x2 * (pc * C1), x6 = x4 * x2,
                     c = c1 + x4 * (pc * C2);
        return (float)(c + x6 * c2);
    }
}
#endif
