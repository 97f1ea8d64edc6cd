/*
 * orbx_synth.h — deterministic, integer-only synthetic mono frame generator (header-only, C99/C++).
 *
 * Implements the synthetic-input recipe of SURVEY.md §8(d): a smooth low-frequency background, K random
 * filled rectangles / discs (K = 300 at 640x480, scaled with area; intensity 0..255; size 4..40 px) and
 * uniform +-6 noise, from seed 0x4F524230 ("ORB0") + frame_index through SplitMix64 -> xoshiro256**.
 * Integer arithmetic only (no libm), so every harness (CPU oracle, GPU bench, tests) sees identical bytes.
 * The reference ships no dataset (Examples/, datasets/ are git-ignored), hence synthetic frames.
 */
#ifndef ORBX_SYNTH_H
#define ORBX_SYNTH_H
#include <stddef.h>
#include <stdint.h>

typedef struct { uint64_t s[4]; } orbx_rng;

static inline uint64_t orbx_splitmix64(uint64_t *x)
{
    uint64_t z = (*x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline void orbx_rng_seed(orbx_rng *r, uint64_t seed)
{
    uint64_t x = seed;
    for (int i = 0; i < 4; ++i) r->s[i] = orbx_splitmix64(&x);
}
static inline uint64_t orbx_rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }
static inline uint64_t orbx_rng_next(orbx_rng *r)
{
    uint64_t *s = r->s;
    const uint64_t result = orbx_rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = orbx_rotl64(s[3], 45);
    return result;
}
/* uniform integer in [0, n) (multiply-shift; n < 2^32) */
static inline uint32_t orbx_rng_below(orbx_rng *r, uint32_t n)
{
    return (uint32_t)(((orbx_rng_next(r) >> 32) * (uint64_t)n) >> 32);
}

#define ORBX_SYNTH_SEED_BASE 0x4F524230ull /* "ORB0" */

static inline void orbx_synth_frame_impl(uint64_t frame_index, int w, int h, uint8_t *out, ptrdiff_t stride)
{
    orbx_rng rng;
    orbx_rng_seed(&rng, ORBX_SYNTH_SEED_BASE + frame_index);
    if (w <= 0 || h <= 0) return;

    /* 1. smooth background: bilinear interpolation of a 32-px grid of values in 96 +- 40 */
    enum { G = 32, GMAX = 80 };
    int gw = w / G + 2, gh = h / G + 2;
    if (gw > GMAX) gw = GMAX;
    if (gh > GMAX) gh = GMAX;
    static const int kGridCap = GMAX * GMAX;
    uint8_t grid[GMAX * GMAX];
    (void)kGridCap;
    for (int i = 0; i < gw * gh; ++i) grid[i] = (uint8_t)(56 + orbx_rng_below(&rng, 81));
    for (int y = 0; y < h; ++y) {
        int gy = y / G, fy = y % G;
        if (gy > gh - 2) { gy = gh - 2; fy = G; }
        uint8_t *row = out + (ptrdiff_t)y * stride;
        for (int x = 0; x < w; ++x) {
            int gx = x / G, fx = x % G;
            if (gx > gw - 2) { gx = gw - 2; fx = G; }
            const uint8_t *g = grid + gy * gw + gx;
            int top = g[0] * (G - fx) + g[1] * fx;
            int bot = g[gw] * (G - fx) + g[gw + 1] * fx;
            row[x] = (uint8_t)((top * (G - fy) + bot * fy + 512) >> 10);
        }
    }

    /* 2. K filled shapes, later shapes overwrite earlier ones */
    long long area = (long long)w * h;
    int K = (int)((300ll * area + 153600) / 307200);
    if (K < 1) K = 1;
    for (int k = 0; k < K; ++k) {
        int disc = (int)(orbx_rng_next(&rng) >> 63);
        int cx = (int)orbx_rng_below(&rng, (uint32_t)w);
        int cy = (int)orbx_rng_below(&rng, (uint32_t)h);
        int sw = 4 + (int)orbx_rng_below(&rng, 37);
        int sh = 4 + (int)orbx_rng_below(&rng, 37);
        uint8_t val = (uint8_t)orbx_rng_below(&rng, 256);
        if (disc) {
            int r = sw / 2, r2 = r * r;
            for (int y = cy - r; y <= cy + r; ++y) {
                if (y < 0 || y >= h) continue;
                uint8_t *row = out + (ptrdiff_t)y * stride;
                int dy = y - cy;
                for (int x = cx - r; x <= cx + r; ++x) {
                    if (x < 0 || x >= w) continue;
                    int dx = x - cx;
                    if (dx * dx + dy * dy <= r2) row[x] = val;
                }
            }
        } else {
            int x0 = cx - sw / 2, y0 = cy - sh / 2;
            for (int y = y0; y < y0 + sh; ++y) {
                if (y < 0 || y >= h) continue;
                uint8_t *row = out + (ptrdiff_t)y * stride;
                for (int x = x0; x < x0 + sw; ++x) {
                    if (x < 0 || x >= w) continue;
                    row[x] = val;
                }
            }
        }
    }

    /* 3. uniform noise in [-6, 6], clamped */
    for (int y = 0; y < h; ++y) {
        uint8_t *row = out + (ptrdiff_t)y * stride;
        int x = 0;
        while (x < w) {
            uint64_t bits = orbx_rng_next(&rng);
            /* four 16-bit draws per 64-bit word */
            for (int k = 0; k < 4 && x < w; ++k, ++x) {
                int n = (int)((((bits >> (16 * k)) & 0xFFFFu) * 13u) >> 16) - 6;
                int v = row[x] + n;
                row[x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
    }
}

#endif /* ORBX_SYNTH_H */
