/*
 * orbx_libm.h - glibc 2.35 logf / atanf / atan2f / tanf, restated (header-only, C99 / C++ / HIP device code).
 *
 * The reference calls the platform libm's FLOAT functions at four places of the path this library replaces:
 *   MapPoint::PredictScale       `ceil(log(ratio) / mfLogScaleFactor)` with float operands under `using namespace std`
 *                                (src/MapPoint.cc:514-546; the directive reaches the file through include/MapPoint.h:23 -> KeyFrame.h ->
 *                                ORBVocabulary.h -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36)  => std::log(float) = logf
 *   KannalaBrandt8::project      `atan2f(sqrtf(x2_plus_y2), v3D[2])`, `atan2f(v3D[1], v3D[0])` (src/CameraModels/KannalaBrandt8.cpp:69-70)
 *   KannalaBrandt8::unproject    `std::tan(theta)` with float theta (:139)                      => tanf
 * Their results are whatever glibc returns, and none of the three is correctly rounded, so a device libm (ocml) or a double-precision
 * evaluation narrowed to float differs from them on a measurable share of inputs.  This header evaluates glibc 2.35's algorithms
 *   sysdeps/ieee754/flt-32/e_logf.c + logf_data.c  (16-entry table, degree-3 polynomial in double, one rounding)
 *   sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c   (fdlibm, float arithmetic)
 *   sysdeps/ieee754/flt-32/s_tanf.c, k_tanf.c (fdlibm, float arithmetic) with e_rem_pio2f.c = reduce_fast / reduce_large of s_sincosf.h (double)
 * operation by operation so that the GPU produces the same floats.  PINNED against the real libm of this image (glibc 2.35):
 * tests/tools/libm_pin.cpp compares every one of them with logf / atanf / atan2f / tanf - logf on every positive float, atanf on every
 * float, tanf on every float, atan2f on >= 4e9 argument pairs (random, the KB8 shapes, exponent-difference and special
 * cases); the record of the exhaustive run is profiles/r05_libm_pin.txt, and tests/test_oracle_kat.py::test_glibc_libm_restatements runs a
 * strided subset of the same program in the CPU suite.
 *
 * Arithmetic rules: float operations un-fused (the sources that include this header are compiled with -ffp-contract=off; glibc builds its
 * generic float routines without FMA), division correctly rounded (ORBX_FDIV).  logf's double operations can also be evaluated with the
 * fma() contractions of glibc's x86-64 FMA build (sysdeps/x86_64/fpu/multiarch/e_logf.c, selected by ifunc on every CPU with FMA + AVX2):
 * ORBX_LOGF_FMA below - the result is the same float everywhere.
 */
#ifndef ORBX_LIBM_H
#define ORBX_LIBM_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef ORBX_HD
#ifdef __HIPCC__
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD static inline
#endif
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define ORBX_FDIV(a, b) __fdiv_rn((a), (b))
#define ORBX_FMA(a, b, c) __fma_rn((a), (b), (c))
#else
#define ORBX_FDIV(a, b) ((a) / (b))
#define ORBX_FMA(a, b, c) fma((a), (b), (c))
#endif

/* 1: logf as glibc's FMA ifunc variant computes it (every x86-64 CPU since Haswell / Zen); 0: the generic un-fused build.  Measured: the two
 * give the SAME float on every one of the 2,139,095,041 positive floats, and both equal this host's logf (profiles/r05_libm_pin.txt), so the
 * choice cannot be observed; the default is the un-fused form like the rest of the library. */
#ifndef ORBX_LOGF_FMA
#define ORBX_LOGF_FMA 0
#endif

ORBX_HD uint32_t orbx_f2u(float f) { uint32_t u; memcpy(&u, &f, sizeof u); return u; }
ORBX_HD float orbx_u2f(uint32_t u) { float f; memcpy(&f, &u, sizeof f); return f; }

/* ---- logf: e_logf.c.  LOGF_TABLE_BITS 4, LOGF_POLY_ORDER 4, OFF 0x3f330000 ---- */
ORBX_HD double orbx_logf_invc(int i)
{
    switch (i) {
    case 0: return 0x1.661ec79f8f3bep+0; case 1: return 0x1.571ed4aaf883dp+0; case 2: return 0x1.49539f0f010bp+0;
    case 3: return 0x1.3c995b0b80385p+0; case 4: return 0x1.30d190c8864a5p+0; case 5: return 0x1.25e227b0b8eap+0;
    case 6: return 0x1.1bb4a4a1a343fp+0; case 7: return 0x1.12358f08ae5bap+0; case 8: return 0x1.0953f419900a7p+0;
    case 9: return 0x1p+0; case 10: return 0x1.e608cfd9a47acp-1; case 11: return 0x1.ca4b31f026aap-1;
    case 12: return 0x1.b2036576afce6p-1; case 13: return 0x1.9c2d163a1aa2dp-1; case 14: return 0x1.886e6037841edp-1;
    default: return 0x1.767dcf5534862p-1;
    }
}
ORBX_HD double orbx_logf_logc(int i)
{
    switch (i) {
    case 0: return -0x1.57bf7808caadep-2; case 1: return -0x1.2bef0a7c06ddbp-2; case 2: return -0x1.01eae7f513a67p-2;
    case 3: return -0x1.b31d8a68224e9p-3; case 4: return -0x1.6574f0ac07758p-3; case 5: return -0x1.1aa2bc79c81p-3;
    case 6: return -0x1.a4e76ce8c0e5ep-4; case 7: return -0x1.1973c5a611cccp-4; case 8: return -0x1.252f438e10c1ep-5;
    case 9: return 0x0p+0; case 10: return 0x1.aa5aa5df25984p-5; case 11: return 0x1.c5e53aa362eb4p-4;
    case 12: return 0x1.526e57720db08p-3; case 13: return 0x1.bc2860d22477p-3; case 14: return 0x1.1058bc8a07ee1p-2;
    default: return 0x1.4043057b6ee09p-2;
    }
}

ORBX_HD float orbx_glibc_logf(float x)
{
    const double Ln2 = 0x1.62e42fefa39efp-1;
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = orbx_f2u(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) { /* x < 0x1p-126, inf or nan */
        if (ix * 2u == 0) return -INFINITY;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return orbx_u2f(0x7fc00000u) /* invalid: (x - x) / 0 */;
        ix = orbx_f2u(x * 0x1p23f); /* subnormal: normalise */
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & (0x1ffu << 23));
    const double invc = orbx_logf_invc(i), logc = orbx_logf_logc(i);
    const double z = (double)orbx_u2f(iz);
#if ORBX_LOGF_FMA
    const double r = ORBX_FMA(z, invc, -1.0);
    const double y0 = ORBX_FMA((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = ORBX_FMA(A1, r, A2);
    y = ORBX_FMA(A0, r2, y);
    y = ORBX_FMA(y, r2, y0 + r);
#else
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
#endif
    return (float)y;
}

/* ---- atanf: s_atanf.c ---- */
ORBX_HD float orbx_glibc_atanf(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    float hi = 0.0f, lo = 0.0f;
    const int32_t hx = (int32_t)orbx_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) { /* |x| >= 2^25 */
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? 1.5707962513e+00f + 7.5497894159e-08f : -1.5707962513e+00f - 7.5497894159e-08f;
    }
    if (ix < 0x3ee00000) { /* |x| < 0.4375 */
        if (ix < 0x31000000) return x; /* |x| < 2^-29 */
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {     /* |x| < 1.1875 */
            if (ix < 0x3f300000) { /* 7/16 <= |x| < 11/16 */
                id = 0; x = ORBX_FDIV(2.0f * x - 1.0f, 2.0f + x);
                hi = 4.6364760399e-01f; lo = 5.0121582440e-09f;
            } else {               /* 11/16 <= |x| < 19/16 */
                id = 1; x = ORBX_FDIV(x - 1.0f, x + 1.0f);
                hi = 7.8539812565e-01f; lo = 3.7748947079e-08f;
            }
        } else {
            if (ix < 0x401c0000) { /* |x| < 2.4375 */
                id = 2; x = ORBX_FDIV(x - 1.5f, 1.0f + 1.5f * x);
                hi = 9.8279368877e-01f; lo = 3.4473217170e-08f;
            } else {               /* 2.4375 <= |x| < 2^25 */
                id = 3; x = ORBX_FDIV(-1.0f, x);
                hi = 1.5707962513e+00f; lo = 7.5497894159e-08f;
            }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (id < 0) return x - x * (s1 + s2);
    z = hi - ((x * (s1 + s2) - lo) - x);
    return hx < 0 ? -z : z;
}

/* ---- atan2f: e_atan2f.c ---- */
ORBX_HD float orbx_glibc_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)orbx_f2u(x), hy = (int32_t)orbx_f2u(y);
    const int32_t ix = hx & 0x7fffffff, iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return orbx_glibc_atanf(y);
    const int32_t m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = orbx_glibc_atanf(fabsf(ORBX_FDIV(y, x)));
    switch (m) {
    case 0: return z;
    case 1: return orbx_u2f(orbx_f2u(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

/* ---- tanf: s_tanf.c + k_tanf.c (fdlibm, float) over the double-precision argument reduction of s_sincosf.h; every float ---- */
ORBX_HD float orbx_glibc_kernel_tanf(float x, float y, int iy)
{
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const float T0 = 3.3333334327e-01f, T1 = 1.3333334029e-01f, T2 = 5.3968254477e-02f, T3 = 2.1869488060e-02f, T4 = 8.8632395491e-03f,
                T5 = 3.5920790397e-03f, T6 = 1.4562094584e-03f, T7 = 5.8804126456e-04f, T8 = 2.4646313977e-04f, T9 = 7.8179444245e-05f,
                T10 = 7.1407252108e-05f, T11 = -1.8558637748e-05f, T12 = 2.5907305826e-05f;
    float z, r, v, w, s;
    const int32_t hx = (int32_t)orbx_f2u(x);
    const int32_t ix = hx & 0x7fffffff;
    if (ix < 0x39000000) { /* |x| < 2^-13 */
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return ORBX_FDIV(1.0f, fabsf(x));
            else if (iy == 1) return x;
            else return ORBX_FDIV(-1.0f, x);
        }
    }
    if (ix >= 0x3f2ca140) { /* |x| >= 0.6744 */
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w; y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (float)(1 - ((hx >> 30) & 2)) * (float)iy * (1.0f - 2.0f * (float)iy * x);
    }
    z = x * x;
    w = z * z;
    r = T1 + w * (T3 + w * (T5 + w * (T7 + w * (T9 + w * T11))));
    v = z * (T2 + w * (T4 + w * (T6 + w * (T8 + w * (T10 + w * T12)))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T0 * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (ORBX_FDIV(w * w, w + v) - r)));
    }
    if (iy == 1) return w;
    {
        float a, t;
        z = orbx_u2f(orbx_f2u(w) & 0xfffff000u);
        v = r - (z - x);
        t = a = ORBX_FDIV(-1.0f, w);
        t = orbx_u2f(orbx_f2u(t) & 0xfffff000u);
        s = 1.0f + t * z;
        return t + a * (s + t * v);
    }
}

/* __inv_pio4[i] (sysdeps/ieee754/flt-32/s_sincosf_data.c): 32-bit windows of the binary expansion of 4 / pi, one byte apart */
ORBX_HD uint32_t orbx_inv_pio4(int i)
{
    switch (i) {
    case 0: return 0xa2u; case 1: return 0xa2f9u; case 2: return 0xa2f983u; case 3: return 0xa2f9836eu;
    case 4: return 0xf9836e4eu; case 5: return 0x836e4e44u; case 6: return 0x6e4e4415u; case 7: return 0x4e441529u;
    case 8: return 0x441529fcu; case 9: return 0x1529fc27u; case 10: return 0x29fc2757u; case 11: return 0xfc2757d1u;
    case 12: return 0x2757d1f5u; case 13: return 0x57d1f534u; case 14: return 0xd1f534ddu; case 15: return 0xf534ddc0u;
    case 16: return 0x34ddc0dbu; case 17: return 0xddc0db62u; case 18: return 0xc0db6295u; case 19: return 0xdb629599u;
    case 20: return 0x6295993cu; case 21: return 0x95993c43u; case 22: return 0x993c4390u; default: return 0x3c439041u;
    }
}

/* rem_pio2f of glibc >= 2.28 (e_rem_pio2f.c on top of s_sincosf.h): reduce_fast below 120, reduce_large above, both in double; the
 * reduced argument leaves as a float pair y0 + y1.  Returns n (only its low bits matter). */
ORBX_HD int orbx_glibc_rem_pio2f(float x, float *y0, float *y1)
{
    const double hpi_inv = 0x1.45F306DC9C883p+23; /* 2 / pi * 2^24 */
    const double hpi = 0x1.921FB54442D18p0;       /* pi / 2 */
    const double pi63 = 0x1.921FB54442D18p-62;    /* 2 pi / 2^64 */
    const uint32_t xi0 = orbx_f2u(x);
    double dx = (double)x;
    int n;
    if (((xi0 >> 20) & 0x7ff) < 0x42f) { /* abstop12(x) < abstop12(120.0f): reduce_fast */
        const double r = dx * hpi_inv;
        n = ((int32_t)r + 0x800000) >> 24;
        dx = dx - (double)n * hpi;
    } else {                             /* reduce_large */
        const int base = (int)((xi0 >> 26) & 15);
        const int shift = (int)((xi0 >> 23) & 7);
        uint32_t xi = (xi0 & 0xffffffu) | 0x800000u;
        xi <<= shift;
        uint64_t res0 = (uint64_t)(uint32_t)(xi * orbx_inv_pio4(base));
        const uint64_t res1 = (uint64_t)xi * orbx_inv_pio4(base + 4);
        const uint64_t res2 = (uint64_t)xi * orbx_inv_pio4(base + 8);
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        dx = (double)(int64_t)res0 * pi63;
        n = (int)nn;
        if (xi0 >> 31) dx = -dx;
    }
    *y0 = (float)dx;
    *y1 = (float)(dx - (double)*y0);
    return n;
}

ORBX_HD float orbx_glibc_tanf(float x)
{
    const int32_t ix = (int32_t)orbx_f2u(x) & 0x7fffffff;
    if (ix <= 0x3f490fda) return orbx_glibc_kernel_tanf(x, 0.0f, 1);
    if (ix >= 0x7f800000) return orbx_u2f(0x7fc00000u); /* inf, NaN: x - x */
    float y0, y1;
    const int n = orbx_glibc_rem_pio2f(x, &y0, &y1);
    return orbx_glibc_kernel_tanf(y0, y1, 1 - ((n & 1) << 1));
}
#endif
