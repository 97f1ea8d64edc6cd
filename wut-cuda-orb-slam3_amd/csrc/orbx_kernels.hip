// orbx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ORB front-end.
//
// Pipeline per batch of frames (all levels, all frames per launch; everything stays in HBM):
//   k_pyr_fast   (x nlevels)  fused: level-l tile -> LDS; bilinear level l+1 (cv::resize fixed point);
//                             FAST-9/16 in the reference's two rounds (iniThFAST, then minThFAST in the empty
//                             cells); score; cell-local 3x3 NMS; ordered (cell-row-major, raster)
//                             compaction.  Replaces ComputePyramid (reference src/ORBextractor.cc:1309-1329) +
//                             tileCalcKeypoints (:867-950) [and the fork's tileCalcKeypoints_kernel].
//   k_quadtree                one wave per (frame, level): DistributeOctTree (:584-774) with exact list
//                             order and libstdc++ std::sort tie order, as level-synchronous wave-parallel sweeps.
//   k_blur3      (x nlevels)  cv::GaussianBlur 7x7 sigma 2, fixed point (:1270-1273), as int8 MFMA products
//                             (k_blur2: SWAR, k_blur: generic fallbacks).
//   k_pack                    border/octave/size (:1004-1017), pt *= scale, mono/stereo split (:1283-1303).
//   k_orient_desc             one wave per 8 keypoints: IC_Angle moments + fastAtan2 (Angle.cl:24-60) and the
//                             256-bit steered BRIEF (computeOrbDescriptor, :105-149), 64 bits per ballot.
//
// Integer / bitwise work on the VALU; the separable blur - a banded Toeplitz product - runs on the matrix cores in exact int8 /
// int32 arithmetic (k_blur3).  Floating point only where the reference has it, evaluated
// with explicitly rounded, un-fused operations (__fmul_rn/__fadd_rn/__fdiv_rn; file is also built with
// -ffp-contract=off) so results are bit-identical to the x86 CPU path.
#include "orbx_internal.h"
#include "libstdcxx_sort.h"
#include "orbx_pattern.inc"
#include "../../include/orbx_trig.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#define WAVE 64

// XCD-aware work mapping.  Workgroups are dealt round-robin over the 8 XCDs in linear block order, and each XCD has
// its own 4 MiB L2.  A 1-D grid of 8 * ceil(nframes / 8) * unitsPerFrame blocks is decoded so that block b works on
// frame (b % 8) + 8 * (b / (8 * units)) and unit (b / 8) % units: eight frames are in flight, one per XCD, and
// every unit (tile / keypoint chunk) of a frame runs on the same XCD back to back, so halo rows, overlapping
// keypoint patches and the level written by the previous launch are re-read from that XCD's L2, not from HBM.
__device__ __forceinline__ bool xcd_decode(int nframes, int units, int &frame, int &unit)
{
    const unsigned b = blockIdx.x;
    const unsigned per = 8u * (unsigned)units;
    const unsigned chunk = b / per, rem = b - chunk * per;
    frame = (int)(chunk * 8u + (rem & 7u));
    unit = (int)(rem >> 3);
    return frame < nframes;
}
static inline unsigned xcd_grid(int nframes, int units) { return (unsigned)((nframes + 7) / 8) * 8u * (unsigned)units; }
// The same decode with the division by `per` = 8 * units as a multiply-high by a host-computed reciprocal (xcd_magic): the generic 32-bit
// division is ~25 dependent scalar / vector instructions (v_rcp_iflag_f32 and two correction steps) in front of a workgroup's first load.
// For per not a power of two: m = floor(2^(32 + l) / per) + 1, l = floor(log2 per): floor(b / per) = mulhi(b, m) >> l for every b < 2^31.
struct XcdMagic { unsigned m; int l; unsigned per; };
static inline XcdMagic xcd_magic(int units)
{
    XcdMagic x;
    x.per = 8u * (unsigned)units;
    int l = 0;
    while ((2u << l) <= x.per) ++l;   // floor(log2 per)
    x.l = l;
    x.m = (x.per & (x.per - 1)) == 0 ? 0u : (unsigned)((((unsigned long long)1 << (32 + l)) / x.per) + 1ull);   // 0: power of two, shift only
    return x;
}
// floor(b / d) for b < 2^31 by the same construction, any divisor d >= 1
static inline XcdMagic div_magic(unsigned d)
{
    XcdMagic x;
    x.per = std::max(d, 1u);
    int l = 0;
    while ((2u << l) <= x.per) ++l;
    x.l = l;
    x.m = (x.per & (x.per - 1)) == 0 ? 0u : (unsigned)((((unsigned long long)1 << (32 + l)) / x.per) + 1ull);
    return x;
}
__device__ __forceinline__ unsigned div_by_magic(unsigned b, unsigned m, int l) { return m ? __umulhi(b, m) >> l : b >> l; }
__device__ __forceinline__ bool xcd_decode_magic(int nframes, unsigned m, int l, unsigned per, int &frame, int &unit)
{
    const unsigned b = blockIdx.x;
    const unsigned chunk = m ? __umulhi(b, m) >> l : b >> l;
    const unsigned rem = b - chunk * per;
    frame = (int)(chunk * 8u + (rem & 7u));
    unit = (int)(rem >> 3);
    return frame < nframes;
}

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    const unsigned lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// =====================================================================================================
// FAST-9/16 (cv::FAST TYPE_9_16)
// =====================================================================================================

// non-zero iff the 16-bit circular mask has >= 9 contiguous set bits
__device__ __forceinline__ unsigned ring_run9(unsigned m)
{
    m |= m << 16;
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return r;
}

// 3-input min / max (v_min3_i32 / v_max3_i32: one half-rate op instead of two; the compiler does not form them
// from nested min()/max() here)
__device__ __forceinline__ int min3i(int a, int b, int c)
{
    int r;
    asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int max3i(int a, int b, int c)
{
    int r;
    asm("v_max3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// (a * b) >> 32 for a, b < 2^24 (v_mul_hi_u32_u24: half rate; a full v_mul_lo_u32 / v_mul_hi_u32 issues at ~8 cycles on gfx950)
__device__ __forceinline__ uint32_t mul_hi_u24(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Number of set bits of a wave mask below this lane.
__device__ __forceinline__ unsigned mbcnt64(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// `total` (wave-uniform) slots of an LDS queue for the active lanes of this wave: the first active lane does the one atomic.
__device__ __forceinline__ unsigned wave_queue_alloc(unsigned *counter, unsigned total)
{
    unsigned base = 0;
    if (mbcnt64(__ballot(1)) == 0) base = atomicAdd(counter, total);
    return __builtin_amdgcn_readfirstlane(base);
}

// `n` queue slots for THIS lane from an LDS counter: the hardware's own same-address atomic (the active lanes' additions are serialised
// by the LDS, a few cycles each).  Written as atomicAdd() the compiler "optimises" it into a wave reduction over v_readlane; the
// ballot / v_mbcnt prefix counts of wave_queue_alloc cost ~12 vector instructions per queue and wave.  For queues whose order is free.
__device__ __forceinline__ unsigned lane_queue_alloc(unsigned *counter, unsigned n)
{
    unsigned r;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)counter;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(n) : "memory");
    return r;
}

// FAST corner score (cornerScore<16>: largest threshold for which the pixel is still a corner = max over the
// 16 arcs of 9 of the minimum one-sided difference, minus 1).  `c` points at the centre pixel inside an LDS tile
// with row pitch `tp`.  Only called for pixels that passed the ring test (sparse), so byte reads are fine here.
// min over 9 consecutive = min3 of three min3's: 16 + 16 ops per polarity, then a max3 tree.
__device__ __forceinline__ int fast_corner_score(const uint8_t *c, int tp)
{
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * tp];       d[1] = v - c[3 * tp + 1];   d[2] = v - c[2 * tp + 2];   d[3] = v - c[tp + 3];
    d[4] = v - c[3];            d[5] = v - c[-tp + 3];      d[6] = v - c[-2 * tp + 2];  d[7] = v - c[-3 * tp + 1];
    d[8] = v - c[-3 * tp];      d[9] = v - c[-3 * tp - 1];  d[10] = v - c[-2 * tp - 2]; d[11] = v - c[-tp - 3];
    d[12] = v - c[-3];          d[13] = v - c[tp - 3];      d[14] = v - c[2 * tp - 2];  d[15] = v - c[3 * tp - 1];
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn3[k] = min3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
        mx3[k] = max3i(d[k], d[(k + 1) & 15], d[(k + 2) & 15]);
    }
    int m9[16], M9[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        m9[k] = min3i(mn3[k], mn3[(k + 3) & 15], mn3[(k + 6) & 15]);
        M9[k] = max3i(mx3[k], mx3[(k + 3) & 15], mx3[(k + 6) & 15]);
    }
    int a5[5], b5[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        a5[k] = max3i(m9[3 * k], m9[3 * k + 1], m9[3 * k + 2]);
        b5[k] = min3i(M9[3 * k], M9[3 * k + 1], M9[3 * k + 2]);
    }
    const int a = max3i(max3i(a5[0], a5[1], a5[2]), max3i(a5[3], a5[4], m9[15]), -256);
    const int b = min3i(min3i(b5[0], b5[1], b5[2]), min3i(b5[3], b5[4], M9[15]), 256);
    return max(a, -b) - 1;
}

// FAST-9/16 ring test — byte-SWAR, 4 pixels (one "group") per 32-bit operation.
//
// Measured on gfx950 (profiles/r01_valu_microbench.txt): v_and/v_or/v_add/v_sub/v_bitop3/shifts issue at 2
// cycles per wave64, while SDWA forms, v_alignbit, min/max, v_pk_*, v_perm, v_bfi and v_mad issue at 4.  So the
// test is phrased in full-rate ops only (plus one v_alignbyte per ring position to fetch 4 unaligned bytes):
//   * a group = 4 adjacent pixels = one dword V of centres; ring position k of the group = the (unaligned) dword
//     P_k at byte offset (dx_k, dy_k) from it, fetched from the LDS tile as two aligned dwords + one v_alignbyte.
//   * per byte "P < LO" (dark) and "P > HI" (bright) with LO = v - th, HI = v + th use the borrow-free 7+1-bit
//     compare: t = (Pl + (0x80 - LOl)) has bit 7 = [Pl >= LOl]; flag = (~P7 & LO7) | (~(P7 ^ LO7) & ~t7).
//     LO / HI wrap modulo 256; pixels whose v - th < 0 (v + th > 255) cannot have dark (bright) ring pixels and
//     are masked once at the end.
//   * the flags stay transposed (one word per ring position, bit 7 of each byte = one pixel); ">= 9 contiguous"
//     = OR over the 16 arcs of the AND of 9 consecutive words, with 3-input ANDs (v_bitop3).
// Returns the flags of the 4 pixels in bit 7 of each byte.
#define SW_H 0x80808080u
#define SW_L 0x7F7F7F7Fu

// 3-input bitwise function by truth table (v_bitop3_b32): result bit = tt[(a << 2) | (b << 1) | c], i.e. a = 0xF0, b = 0xCC, c = 0xAA.
#define BITOP3(a, b, c, tt) __builtin_amdgcn_bitop3_b32((a), (b), (c), (tt))
// per byte, bit 7: [x < y] (unsigned) from x, y and a word t whose bit 7 says [low 7 bits of x >= low 7 bits of y]:
// (~x & y) | (~(x ^ y) & ~t) - ONE v_bitop3.  Written as that C expression the compiler does not keep it one instruction where x or y is
// itself an expression: it expands LO = tS ^ ((V ^ ~T4) & SW_H) and HI into their terms and re-associates, and every compare then
// costs two or three v_bitop3 plus a (half-rate) v_and_or - 9 instead of 6 vector instructions per ring position (round 4; ISA of round 3's
// kernel).  PF_SWAR_IMPL=0 keeps the round-3 formulation for tools/experiments/ring_rate.hip.
#define SW_LT(x, y, t) BITOP3((x), (y), (t), 0x4D)
#define SW_AND3(a, b, c) BITOP3((a), (b), (c), 0x80)
#define SW_OR3(a, b, c) BITOP3((a), (b), (c), 0xFE)
#ifndef PF_SWAR_IMPL
#define PF_SWAR_IMPL 1
#endif
// a value the optimiser cannot look into (no instruction)
__device__ __forceinline__ uint32_t sw_opaque(uint32_t x) { asm("" : "+v"(x)); return x; }

// The per-group constants of the two compares of a ring byte P against the centre byte v:
//   dark   P < LO = v - th :  td = (P & 0x7F) + Kd,  Kd = 0x80 - (LO & 0x7F):  bit 7 of td = [Pl >= LOl];  D = SW_LT(P, LO, td)
//   bright P > HI = v + th :  tb = HIh - (P & 0x7F), HIh = HI | 0x80:          bit 7 of tb = [HIl >= Pl];  B = SW_LT(HI, P, tb)
// LO / HI wrap modulo 256; pixels with v - th < 0 (v + th > 255) cannot have dark (bright) ring pixels: noDark / noBright.
struct SwarCentre {
    uint32_t LO, HI, Kd, HIh, noDark, noBright;
};
__device__ __forceinline__ SwarCentre swar_centre(uint32_t V, uint32_t T4, uint32_t C4)
{
    SwarCentre c;
    const uint32_t Vl = V & SW_L;
    const uint32_t tS = (V | SW_H) - (T4 & SW_L);
    const uint32_t LO = tS ^ ((V ^ ~T4) & SW_H);
    c.noDark = SW_LT(V, T4, tS);
    const uint32_t HI = (Vl + (T4 & SW_L)) ^ ((V ^ T4) & SW_H);
    const uint32_t tC = (C4 | SW_H) - Vl;
    c.noBright = SW_LT(C4, V, tC);
    c.LO = sw_opaque(LO);
    c.HI = sw_opaque(HI);
    c.Kd = sw_opaque(SW_H - (c.LO & SW_L));
    c.HIh = sw_opaque(c.HI | SW_H);
    return c;
}

// The group's pixels are bytes 4..7 of a 16-byte LDS window starting at `wp` (row y-3; 4-byte aligned).  Each ring dword
// pair is fetched from LDS on demand instead of holding a 7 x 16-byte window in registers: ~50 VGPRs instead of ~75,
// so the kernel reaches 8 waves/SIMD without scratch spills.
__device__ __forceinline__ uint32_t fast_group_flags_lds(const uint8_t *wp, int tp, uint32_t T4, uint32_t C4)
{
    const uint32_t V = *(const uint32_t *)(wp + 3 * tp + 4);
#if PF_SWAR_IMPL
    const SwarCentre c = swar_centre(V, T4, C4);
    const uint32_t LO = c.LO, HI = c.HI, Kd = c.Kd, HIh = c.HIh, ltVT = c.noDark, ltCV = c.noBright;
#else
    const uint32_t Vl = V & SW_L;
    const uint32_t tS = (V | SW_H) - (T4 & SW_L);
    const uint32_t LO = tS ^ ((V ^ ~T4) & SW_H);
    const uint32_t ltVT = (~V & T4) | (~(V ^ T4) & ~tS);
    const uint32_t HI = (Vl + (T4 & SW_L)) ^ ((V ^ T4) & SW_H);
    const uint32_t tC = (C4 | SW_H) - Vl;
    const uint32_t ltCV = (~C4 & V) | (~(C4 ^ V) & ~tC);
    const uint32_t Kd = SW_H - (LO & SW_L);
    const uint32_t HIh = HI | SW_H;
#endif
    uint32_t D[16], B[16];
#if PF_SWAR_IMPL
#define RING_CMP(k) D[k] = SW_LT(P, LO, td); B[k] = SW_LT(HI, P, tb);
#else
#define RING_CMP(k) D[k] = (~P & LO) | (~(P ^ LO) & ~td); B[k] = (~HI & P) | (~(HI ^ P) & ~tb);
#endif
#define RING(k, dx, dy)                                                                            \
    {                                                                                              \
        constexpr int o = 4 + (dx);                                                                \
        const uint32_t *q = (const uint32_t *)(wp + (3 + (dy)) * tp + (o & ~3));                   \
        const uint32_t P = (o & 3) == 0 ? q[0] : __builtin_amdgcn_alignbyte(q[1], q[0], o & 3);    \
        const uint32_t Pl = P & SW_L;                                                              \
        const uint32_t td = Pl + Kd;                                                               \
        const uint32_t tb = HIh - Pl;                                                              \
        RING_CMP(k)                                                                                \
    }
    RING(0, 0, 3) RING(1, 1, 3) RING(2, 2, 2) RING(3, 3, 1) RING(4, 3, 0) RING(5, 3, -1) RING(6, 2, -2) RING(7, 1, -3)
    RING(8, 0, -3) RING(9, -1, -3) RING(10, -2, -2) RING(11, -3, -1) RING(12, -3, 0) RING(13, -3, 1) RING(14, -2, 2) RING(15, -1, 3)
#undef RING
#undef RING_CMP
#if PF_SWAR_IMPL
    // ">= 9 contiguous of 16": runs of three, runs of nine out of three runs of three, then the OR over the 16 arcs - all 3-input
    uint32_t d3[16], b3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        d3[k] = SW_AND3(D[k], D[(k + 1) & 15], D[(k + 2) & 15]);
        b3[k] = SW_AND3(B[k], B[(k + 1) & 15], B[(k + 2) & 15]);
    }
    uint32_t d9[16], b9[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        d9[k] = SW_AND3(d3[k], d3[(k + 3) & 15], d3[(k + 6) & 15]);
        b9[k] = SW_AND3(b3[k], b3[(k + 3) & 15], b3[(k + 6) & 15]);
    }
    uint32_t rd = SW_OR3(d9[0], d9[1], d9[2]), rb = SW_OR3(b9[0], b9[1], b9[2]);
#pragma unroll
    for (int k = 3; k < 15; k += 2) {
        rd = SW_OR3(rd, d9[k], d9[k + 1]);
        rb = SW_OR3(rb, b9[k], b9[k + 1]);
    }
    // (rd | d9[15]) & ~noDark  |  (rb | b9[15]) & ~noBright
    rd = BITOP3(rd, d9[15], ltVT, 0x54);  // (a | b) & ~c
    rb = BITOP3(rb, b9[15], ltCV, 0x54);
    return (rd | rb) & SW_H;
#else
    uint32_t d3[16], b3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        d3[k] = D[k] & D[(k + 1) & 15] & D[(k + 2) & 15];
        b3[k] = B[k] & B[(k + 1) & 15] & B[(k + 2) & 15];
    }
    uint32_t rd = 0, rb = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        rd |= d3[k] & d3[(k + 3) & 15] & d3[(k + 6) & 15];
        rb |= b3[k] & b3[(k + 3) & 15] & b3[(k + 6) & 15];
    }
    return ((rd & ~ltVT) | (rb & ~ltCV)) & SW_H;
#endif
}

// Cheap necessary condition for the ring test of one 4-pixel group: a contiguous arc of 9 ring pixels contains two
// adjacent compass points, i.e. (k0 | k8) & (k4 | k12) must hold for one polarity (the first two of OpenCV's
// opposite-pair checks).  Needs only the rows y-3, y, y+3 of the strip window (w0, w3, w6).  On the benchmark
// frames 11-22 % of the groups pass at iniThFAST; only those go through the full 16-position test.
// V = the group's centre dword, Vlft / Vrgt its neighbours in the row, Pdn / Pup the dwords three rows below / above.
__device__ __forceinline__ uint32_t fast_group_pretest_v(uint32_t V, uint32_t Vlft, uint32_t Vrgt, uint32_t Pdn, uint32_t Pup, uint32_t T4, uint32_t C4)
{
#if PF_SWAR_IMPL
    const SwarCentre c = swar_centre(V, T4, C4);
    const uint32_t LO = c.LO, HI = c.HI, Kd = c.Kd, HIh = c.HIh, ltVT = c.noDark, ltCV = c.noBright;
#else
    const uint32_t Vl = V & SW_L;
    const uint32_t tS = (V | SW_H) - (T4 & SW_L);
    const uint32_t LO = tS ^ ((V ^ ~T4) & SW_H);
    const uint32_t ltVT = (~V & T4) | (~(V ^ T4) & ~tS);
    const uint32_t HI = (Vl + (T4 & SW_L)) ^ ((V ^ T4) & SW_H);
    const uint32_t tC = (C4 | SW_H) - Vl;
    const uint32_t ltCV = (~C4 & V) | (~(C4 ^ V) & ~tC);
    const uint32_t Kd = SW_H - (LO & SW_L);
    const uint32_t HIh = HI | SW_H;
#endif
    uint32_t D[4], B[4];
    const uint32_t P[4] = {Pdn, __builtin_amdgcn_alignbyte(Vrgt, V, 3), Pup, __builtin_amdgcn_alignbyte(V, Vlft, 1)};  // k = 0, 4, 8, 12
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t Pl = P[j] & SW_L;
        const uint32_t td = Pl + Kd;
        const uint32_t tb = HIh - Pl;
#if PF_SWAR_IMPL
        D[j] = SW_LT(P[j], LO, td);
        B[j] = SW_LT(HI, P[j], tb);
#else
        D[j] = (~P[j] & LO) | (~(P[j] ^ LO) & ~td);
        B[j] = (~HI & P[j]) | (~(HI ^ P[j]) & ~tb);
#endif
    }
#if PF_SWAR_IMPL
    // ((D0 | D2) & (D1 | D3) & ~noDark) | ((B0 | B2) & (B1 | B3) & ~noBright)
    const uint32_t dv = BITOP3(D[0], D[2], ltVT, 0x54), bv = BITOP3(B[0], B[2], ltCV, 0x54);  // (a | b) & ~c
    const uint32_t d = BITOP3(D[1], D[3], dv, 0xA8), b = BITOP3(B[1], B[3], bv, 0xA8);        // (a | b) & c
    return (d | b) & SW_H;
#else
    const uint32_t d = (D[0] | D[2]) & (D[1] | D[3]);
    const uint32_t b = (B[0] | B[2]) & (B[1] | B[3]);
    return ((d & ~ltVT) | (b & ~ltCV)) & SW_H;
#endif
}
template <int GRP>
__device__ __forceinline__ uint32_t fast_group_pretest(const uint32_t (&w0)[4], const uint32_t (&w3)[4], const uint32_t (&w6)[4],
                                                       uint32_t T4, uint32_t C4)
{
    return fast_group_pretest_v(w3[1 + GRP], w3[GRP], w3[2 + GRP], w6[1 + GRP], w0[1 + GRP], T4, C4);
}

// exclusive prefix sum over the wave (all 64 lanes active): Kogge-Stone inside the rows of 16 lanes with DPP row shifts, then the
// row totals with the two gfx9 row broadcasts - six full-rate VALU operations, no LDS crossbar round trips
__device__ __forceinline__ int wave_excl_scan(int v, int &total)
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1, 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(x, 63);
    return x - v;
}

struct PyrFastArgs {
    OrbxLevelGeom g;
    const uint8_t *src; int srcPitch; long long srcFrameStride; int srcAlign;  // 16, 4 or 1
    uint8_t *dst; int dw, dh, dstPitch; long long dstFrameStride;
    OrbxResizeTabs tabs;
    uint32_t *cand; long long candFrameStride;
    int *tileCount; int tilesPerFrame;
    int iniTh, minTh;
    int nframes;
    const OrbxPfTile *tiles;  // [g.nTiles] host-built tile records of this level
    int laneAlloc;  // 1: queue slots by per-lane LDS atomics (ORBX_PF_LANE_ALLOC), 0: ballot prefix counts
    unsigned xcdM; int xcdL; unsigned xcdPer;  // xcd_magic(g.nTiles): the workgroup -> (frame, tile) decode without a division
    int phases;  // debug/ablation mask (ORBX_PF_PHASES): 1 resize, 2 ring test+score, 4 NMS, 8 count+emit;
                 // bits 4..7 = stop point for instruction-count profiles (tools/pmc_phases.sh): return after 1 tile load,
                 // 2 LDS setup, 3 resize, 4 pre-test, 5 ring test, 6 score, 7 NMS (round 0)
};

#define PF_MAX_GROUP_CELLS 32
#ifndef PF_NT
#define PF_NT 512           // threads per workgroup: 8 waves x 3 workgroups/CU (LDS-limited); with the queue-based two-round
                            // FAST the kernel is barrier/latency-bound and 512 threads measured 3.79 -> 3.14 ms per batch
#endif
#define PF_NW (PF_NT / 64)
int orbxPyrFastThreads() { return PF_NT; }
#define PF_XO 16      // LDS column of image column ax0 (keeps every strip window inside the tile row)
#define PF_QCAP 1024  // corner queue entries (u16: row << 9 | interior column)

// score + cell-local NMS of one ring-test survivor at tile row r, LDS column c
// (`score` points at interior column 0 minus cInt0: it is indexed with LDS columns like the pixel tile, with its own pitch `sp`)
__device__ __forceinline__ void pf_score_one(const uint8_t *tile, uint8_t *score, int tp, int sp, int rowOff, int r, int c, int minEff)
{
    const int sc = fast_corner_score(tile + (rowOff + r) * tp + c, tp);
    score[(r + 1) * sp + c] = (uint8_t)(sc >= minEff ? sc : 0);
}
__device__ __forceinline__ void pf_nms_one(const uint8_t *score, const uint8_t *colInfo, uint32_t *bits, int *cnt, int tp /* score pitch */, int r, int c,
                                           int cInt0, int wCell, int hCell, int wpc)
{
    const uint8_t *s1p = score + (r + 1) * tp + c, *s0p = s1p - tp, *s2p = s1p + tp;
    const int sc = s1p[0];
    if (sc > 0) {
        const int xi = c - cInt0;
        const int ci = colInfo[xi];
        int mx = max(s0p[0], s2p[0]);
        if (!(ci & 0x80)) mx = max(mx, max(s0p[-1], max(s1p[-1], s2p[-1])));
        if (!(ci & 0x40)) mx = max(mx, max(s0p[1], max(s1p[1], s2p[1])));
        if (sc > mx) {
            const int cell = ci & 0x3F;
            const int cx = xi - cell * wCell;
            const uint32_t bit = 1u << (cx & 31);
            const uint32_t old = atomicOr(&bits[(cell * hCell + r) * wpc + (cx >> 5)], bit);
            if (!(old & bit)) atomicAdd(&cnt[cell], 1);  // the overflow path may visit a pixel twice: count a bit once
        }
    }
}

#ifdef PF_TIMING
// debug build (-DPF_TIMING): s_memtime stamps of wave 0 at the phase boundaries of the first 8192 workgroups of the last launch
// that also produced a pyramid level; read back with orbx_debug_pf_timing (tools/pf_timing.py runs a 2-level extractor)
__device__ unsigned long long g_pfTiming[8192][16];
extern "C" int orbx_debug_pf_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pfTiming), sizeof(g_pfTiming)); }
#define PF_STAMP(i) do { if (blockIdx.x < 8192 && threadIdx.x == 0 && K().dst) g_pfTiming[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PF_STAMP(i) do { } while (0)
#endif
// DBG: the ablation mask / stop points of PyrFastArgs::phases (ORBX_PF_PHASES) are compiled in.  They are seven more wave-uniform
// conditions that stay live through the FAST rounds (14 SGPRs, all spilled), so the production instance is built without them.
#ifndef PF_MIN_WAVES
#define PF_MIN_WAVES 8      // waves per SIMD the register budget is for (8: 64 VGPRs; 6: 80 - measured slower, DESIGN 6 "Round 4")
#endif
#ifndef PF_DUMMY_SALU
#define PF_DUMMY_SALU 0
#endif
#ifndef PF_RESIZE_PACK_SDWA
#define PF_RESIZE_PACK_SDWA 1
#endif
#ifndef PF_RESIZE_WINDOW
#define PF_RESIZE_WINDOW 1   // round 5: dword windows + v_perm_b32 + v_dot2_u32_u16 for the horizontal pass of the resize (0: the byte reads of round 4)
#endif
// ML (multi-level): FAST / NMS / ordered candidates of ALL levels of a (small) batch in one launch - the pyramid exists already
// (k_resize_level chain), the per-level argument blocks (the same PyrFastArgs the per-level launches take, with dst = NULL) lie in a
// device array and are read like the kernarg segment, through the constant address space (DESIGN 6 "One frame at a time").  The
// per-level kernel (ML = false) reads its kernarg segment exactly as before: handing the block to a common body as a pointer cost it 12
// spilled VGPRs and 0.23 ms per batch.
struct PyrFastLevelsArgs {
    const PyrFastArgs *levels;            // [nlevels]
    int tileStart[ORBX_MAX_LEVELS + 1];   // prefix of the levels' tile counts
    int nlevels, nframes;
};
template <bool ML> struct PyrFastKernArg { typedef PyrFastArgs type; };
template <> struct PyrFastKernArg<true> { typedef PyrFastLevelsArgs type; };

template <bool DBG, bool ML = false>
__global__ __launch_bounds__(PF_NT, PF_MIN_WAVES) void k_pyr_fast(typename PyrFastKernArg<ML>::type)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    // Scalars are kept PER PHASE.  The kernel has four phases (stage, resize, FAST rounds, emission) and ~60 wave-uniform values
    // (tile geometry, LDS sub-buffer offsets, loop bounds); with all of them computed in the prologue and live to the end the
    // 80-SGPR budget that 8 waves / SIMD leaves spilled ~50 of them to VGPR lanes - v_writelane / v_readlane are VALU
    // instructions, ~8 % of this VALU-bound kernel.  So: the geometry of a tile is a host-built record (OrbxPfTile, one per tile of
    // the level) and the argument block stays in the kernarg segment; every phase reads what it needs from both through pointers
    // the compiler cannot look through (`K()`, `T()`), with scalar loads, into registers that die with the phase.
    typedef const __attribute__((address_space(4))) PyrFastArgs KArgs;
    typedef const __attribute__((address_space(4))) OrbxPfTile TileK;
    int f, tileId;
    KArgs *levelArgs = nullptr;
    if constexpr (ML) {
        typedef const __attribute__((address_space(4))) PyrFastLevelsArgs LArgs;
        LArgs *la = (LArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        int unit;
        if (!xcd_decode(la->nframes, la->tileStart[la->nlevels], f, unit)) return;
        int level = 0;
        for (int l = 1; l < la->nlevels; ++l) level += unit >= la->tileStart[l] ? 1 : 0;
        tileId = unit - la->tileStart[level];
        levelArgs = (KArgs *)(unsigned long long)(la->levels + level);
    }
    auto K = [&]() -> KArgs & {
        KArgs *p;
        if constexpr (ML) p = levelArgs; else p = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return *p;
    };
    PF_STAMP(0);
    if constexpr (!ML) {
        KArgs &a = K();
        if (!xcd_decode_magic(a.nframes, a.xcdM, a.xcdL, a.xcdPer, f, tileId)) return;
    }
    auto T = [&]() -> TileK & { TileK *p = (TileK *)(unsigned long long)K().tiles + tileId; asm volatile("" : "+s"(p)); return *p; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // opaque copy of the thread index: stops the compiler from hoisting every phase's per-lane address arithmetic into the
    // kernel prologue, where it was being spilled to scratch at the 64-VGPR budget needed for 4 workgroups per CU
    auto otid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    __shared__ int cnt[PF_MAX_GROUP_CELLS];                 // keypoints per cell (LDS atomics in the NMS phase)
    __shared__ unsigned gqCountR[2], cqCountR[2];           // queue fill counters, one pair per round (no reset barrier)
    // LDS sub-buffers (byte offsets computed by the host, OrbxLevelGeom::o*): tile [tileRows][tp] | score [hCell + 2][tp], same column
    // indexing | bits [cell][row][wpc] NMS survivors | xt / yt resize tables of this tile | gq groups that passed the pre-test
    // (row << 9 | 2 * strip + grp) | cq ring-test survivors (row << 9 | interior column) | gqm | colInfo per interior column
    uint8_t *const tile = lds;

    // ---- phase 0: stage the tile: coalesced 16-byte loads of row-major image rows into LDS ------------------------
    // Two chains of global loads - the tile, and (two levels deep) the resize tables of this tile.  Both are ISSUED first (tile
    // vectors and table entries into registers), the LDS-only setup (zeroing, cell table) runs under their latency, and the
    // registers go to LDS last; issued one after the other the chains cost ~2500 cycles each.
    {
        KArgs &a = K();
        TileK &t = T();
        const int tp = a.g.tilePitch;
        const int ax0 = t.ax0, nV = t.nV, nVec = t.nVec;
        // Staged image rows [lyFirst, lyFirst + nRowsAll): the tile's own rows [ly0, ly0 + nRowsL) go to the pixel tile; the margin bands
        // of an edge tile (OrbxPfTile::bandTop / bandBot: rows beyond the 3-px halo, sources of the next level only) go to the bytes
        // of the score tile + bitmaps, which are first used - and for such tiles first zeroed - in the FAST rounds.
        const int bandTop = t.bandTop, nRowsL = t.nRowsL, nRowsAll = bandTop + nRowsL + t.bandBot, lyFirst = t.ly0 - bandTop;
        const bool hasBands = nRowsAll != nRowsL;  // (wave-uniform)
        const uint32_t oBand = (uint32_t)a.g.oScore;
        auto ldsRowOff = [&](uint32_t r) -> uint32_t {  // byte offset of staged row r in LDS
            if (!hasBands) return __umul24(r, (uint32_t)tp);
            const uint32_t rr = r - (uint32_t)bandTop;
            return rr < (uint32_t)nRowsL ? __umul24(rr, (uint32_t)tp) : oBand + __umul24(r < (uint32_t)bandTop ? r : r - (uint32_t)nRowsL, (uint32_t)tp);
        };
        const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
        const bool doResize = a.dst && (!DBG || (a.phases & 1));
        const int ndx = doResize ? t.dx1 - t.dx0 : 0, ndy = doResize ? t.dy1 - t.dy0 : 0, dx0 = t.dx0, dy0 = t.dy0;
        const float invNV = __builtin_amdgcn_rcpf((float)nV);  // 1-ulp reciprocal: the quotients below sit >= 0.5/n from an integer
        uint4 tv0 = make_uint4(0, 0, 0, 0), tv1 = tv0;  // (two named registers sets: an array here ends up in scratch)
        int tvOff0 = -1, tvOff1 = -1;
        if (a.srcAlign == 16) {
            // vector it = r * nV + c of the tile: r by a 20-bit reciprocal (the tile record's; exact below 1024), every product on the 24-bit
            // multiplier, and a 32-bit offset from the frame's uniform base (a frame is < 2 GB): converted through float with 32-bit
            // multiplies and a 64-bit address per lane this was ~60 issue cycles per load, in front of the kernel's first memory access
            const int it0 = otid(), it1 = it0 + PF_NT;
            const uint32_t nVM = (uint32_t)t.nVMagic, rowBase = (uint32_t)(lyFirst * a.srcPitch + ax0);
            if (it0 < nVec) {
                const uint32_t r = __umul24((uint32_t)it0, nVM) >> 20, c = (uint32_t)it0 - __umul24(r, (uint32_t)nV);
                tv0 = *(const uint4 *)(srcF + (rowBase + __umul24(r, (uint32_t)a.srcPitch) + c * 16));
                tvOff0 = (int)(ldsRowOff(r) + PF_XO + c * 16);
            }
            if (it1 < nVec) {
                const uint32_t r = __umul24((uint32_t)it1, nVM) >> 20, c = (uint32_t)it1 - __umul24(r, (uint32_t)nV);
                tv1 = *(const uint4 *)(srcF + (rowBase + __umul24(r, (uint32_t)a.srcPitch) + c * 16));
                tvOff1 = (int)(ldsRowOff(r) + PF_XO + c * 16);
            }
        }
        int xo_ = 0, xw_ = 0, yo_ = 0, yw_ = 0;  // first table entries of this thread
        if (doResize) {
            const int i = otid();
            if (i < ndx) { xo_ = a.tabs.xofs[dx0 + i]; xw_ = a.tabs.xab[dx0 + i]; }
            if (i < ndy) { yo_ = a.tabs.yofs[dy0 + i]; yw_ = a.tabs.yab[dy0 + i]; }
        }
        // zero the score tile and the per-cell NMS bitmaps (16-byte stores; the bitmaps follow the score tile) - unless margin bands are
        // about to be staged there (edge tiles: zeroed at the head of the FAST rounds instead)
        if (!hasBands) {
            uint4 z = make_uint4(0, 0, 0, 0);
            uint4 *zs = (uint4 *)(lds + a.g.oScore);
            const int nz = (a.g.oXt - a.g.oScore) >> 4;
#pragma nounroll
            for (int i = otid(); i < nz; i += PF_NT) zs[i] = z;
        }
        if (tid < PF_MAX_GROUP_CELLS) cnt[tid] = 0;
        if (tid < 2) { gqCountR[tid] = 0; cqCountR[tid] = 0; }
        // per interior column: cell index | bit7 first column of its cell | bit6 last column
        {
            uint8_t *colInfo = lds + a.g.oColInfo;
            const int wCell = a.g.wCell, TWi = t.TWi;
            const float invWC = __builtin_amdgcn_rcpf((float)wCell);
            // (at most two turns; unrolled / interleaved by eight the loop spilled two registers of the tile vectors that are in flight -
            // and a spill store of a load's destination waits for the load)
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
            for (int x = otid(); x < TWi; x += PF_NT) {
                int cell = (int)(((float)x + 0.5f) * invWC), cx = x - cell * wCell;
                int cw = min(wCell, TWi - cell * wCell);
                colInfo[x] = (uint8_t)(cell | (cx == 0 ? 0x80 : 0) | (cx == cw - 1 ? 0x40 : 0));
            }
        }
        // the tile
        if (a.srcAlign == 16) {
            if (tvOff0 >= 0) *(uint4 *)(tile + tvOff0) = tv0;
            if (tvOff1 >= 0) *(uint4 *)(tile + tvOff1) = tv1;
#pragma nounroll
            for (int it = otid() + 2 * PF_NT; it < nVec; it += PF_NT) {  // tiles of more than 1024 vectors
                const int r = (int)(((float)it + 0.5f) * invNV);
                const int c = it - r * nV;
                const uint4 v = *(const uint4 *)(srcF + (size_t)(lyFirst + r) * a.srcPitch + ax0 + c * 16);
                *(uint4 *)(tile + ldsRowOff((uint32_t)r) + PF_XO + c * 16) = v;
            }
        } else if (a.srcAlign == 4) {
            const int nDw = (t.lx1 - ax0 + 3) >> 2;
#pragma nounroll
            for (int r = wave; r < nRowsAll; r += PF_NW) {
                const uint32_t *srow = (const uint32_t *)(srcF + (size_t)(lyFirst + r) * a.srcPitch + ax0);
                uint32_t *trow = (uint32_t *)(tile + ldsRowOff((uint32_t)r) + PF_XO);
#pragma nounroll
                for (int c = otid() & 63; c < nDw; c += 64) trow[c] = srow[c];
            }
        } else {
            const int nB = t.lx1 - ax0;
#pragma nounroll
            for (int r = wave; r < nRowsAll; r += PF_NW) {
                const uint8_t *srow = srcF + (size_t)(lyFirst + r) * a.srcPitch + ax0;
                uint8_t *trow = tile + ldsRowOff((uint32_t)r) + PF_XO;
#pragma nounroll
                for (int c = otid() & 63; c < nB; c += 64) trow[c] = srow[c];
            }
        }
        PF_STAMP(1);
        if (DBG && (a.phases >> 4) == 1) return;
        // resize coefficient tables of this tile's destination rows / columns -> LDS
        if (doResize) {
            int2 *xt = (int2 *)(lds + a.g.oXt), *yt = (int2 *)(lds + a.g.oYt);
            const int colOf0 = t.colOf0, hm1 = a.g.h - 1;
            const uint32_t tileLdsA = (uint32_t)(size_t)(__attribute__((address_space(3))) uint8_t *)tile;  // LDS address of the tile (< 64 K)
            auto xEntry = [&](int sx, int w2) { return make_int2(sx + colOf0, w2); };  // the right tap is always the next LDS column: where
                                                                                      // cv::resize clamps sx to w - 1 its weight is 0
            auto yEntry = [&](int sy, int w2) {
                const int sy0 = min(max(sy, 0), hm1) - lyFirst, sy1 = min(max(sy + 1, 0), hm1) - lyFirst;
                // .x = the LDS addresses of the two rows (16 bits each; a margin band's rows lie in the score tile's bytes), .y = the first
                // row's weight b0 << 8, ready for v_mul_hi_u32_u24 (phase 1); b1 = 2048 - b0 (cv::resize: the two sum to 2048)
                return make_int2((int)((tileLdsA + ldsRowOff((uint32_t)sy0)) | ((tileLdsA + ldsRowOff((uint32_t)sy1)) << 16)), (w2 & 0xFFFF) << 8);
            };
            const int i0 = otid();
            if (i0 < ndx) xt[i0] = xEntry(xo_, xw_);
            if (i0 < ndy) yt[i0] = yEntry(yo_, yw_);
#pragma nounroll
            for (int i = i0 + PF_NT; i < ndx; i += PF_NT) xt[i] = xEntry(a.tabs.xofs[dx0 + i], a.tabs.xab[dx0 + i]);
#pragma nounroll
            for (int i = i0 + PF_NT; i < ndy; i += PF_NT) yt[i] = yEntry(a.tabs.yofs[dy0 + i], a.tabs.yab[dy0 + i]);
        }
    }
    __syncthreads();
    PF_STAMP(2);
    if (DBG && (K().phases >> 4) == 2) return;

    // ---- phase 1: next pyramid level: cv::resize INTER_LINEAR fixed point; 4 px per lane -------
    {
        KArgs &a = K();
        if (a.dst && (!DBG || (a.phases & 1))) {
            TileK &t = T();
            const int2 *xt = (const int2 *)(lds + a.g.oXt), *yt = (const int2 *)(lds + a.g.oYt);
            uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
            const int dstPitch = a.dstPitch;
            const int dx0 = t.dx0, dx1 = t.dx1, dy0 = t.dy0, dy1 = t.dy1;
            const int q0 = t.q0, nq = t.nq, RC = t.RC, q1 = q0 + nq;
            // One pixel group per lane, kept for the whole phase (its four column entries stay in registers); the 512 lanes are dealt
            // over (row class, group): lane t -> group t mod nq, rows dy0 + t / nq, + RC, + 2 RC ...  With the usual 45 - 53 groups per
            // row a wave-per-row mapping left 20 - 30 % of the lanes idle; here RC = 512 / nq row classes use all but 512 mod nq lanes.
            // (Rows wider than 2048 pixels fall back to several groups per lane.)
            const int tt = otid();
            const int rc = nq >= PF_NT ? 0 : (int)(__umul24((uint32_t)tt, (uint32_t)t.nqMagic) >> 20);  // tt / nq (the tile record's reciprocal)
            for (int q = q0 + (tt - rc * nq); q < q1 && rc < RC; q += PF_NT) {
                // Every 4-pixel group is whole: the host puts the boundaries between column groups at multiples of 4 (dx0 = 4 q0; the tile
                // stages the extra source columns, OrbxLevelGeom::rightHalo) and the last group of a row may run into the row's padding
                // (dx >= dx1 there: the entry of the last column again).  A lane with a partial group stored byte by byte under four exec
                // masks, ~17 scalar instructions in every turn of every lane's loop.
                int ca[4], wa[4], wb[4];
                uint32_t wp[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int2 e = xt[min(q * 4 + j, dx1 - 1) - dx0];
                    ca[j] = e.x;
                    wa[j] = (e.y & 0xFFFF) << 4;   // weights are in [0, 2048]; << 4: the horizontal sum comes out as h << 4 (< 2^24)
                    wb[j] = (e.y >> 16) << 4;
                    wp[j] = (uint32_t)e.y << 4;    // both weights << 4 as a u16 pair (2048 << 4 = 0x8000 still fits): the operand of v_dot2_u32_u16
                }
                const uint32_t two = sw_opaque(2u);  // (SDWA takes no inline constant)
#if PF_RESIZE_WINDOW
                // Round 5: the taps of two neighbouring pixels lie in ONE 8-byte window that starts at the dword of the first pixel's left
                // tap (sx advances by ceil(scale) <= 3 per pixel: scale factors up to 3; the wave falls back to byte reads otherwise).  Per
                // source row two ds_read2_b32 instead of eight ds_read_u8 - the byte reads made this phase the kernel's largest consumer of
                // LDS cycles (196 M of 530 M per launch set, half of them bank conflicts) -, and per (pixel, row) one v_perm_b32 that puts
                // the two taps into the halves of a dword (selector per pixel, constant over the phase) and one v_dot2_u32_u16 against
                // the weight pair: h << 4 = S[sx] * (a0 << 4) + S[sx + 1] * (a1 << 4), the same integer as before.
                const int b01 = ca[0] & ~3, b23 = ca[2] & ~3;
                const int o0 = ca[0] - b01, o1 = ca[1] - b01, o2 = ca[2] - b23, o3 = ca[3] - b23;
                if (__ballot(o1 > 6 || o3 > 6 || o1 < o0 || o3 < o2) == 0) {
                    // selector bytes: {0x0c = constant 0, o + 1, 0x0c, o} -> S[sx] | S[sx + 1] << 16 out of {hi dword, lo dword}
                    const uint32_t sl0 = (uint32_t)o0 * 0x00010001u + 0x0c010c00u, sl1 = (uint32_t)o1 * 0x00010001u + 0x0c010c00u;
                    const uint32_t sl2 = (uint32_t)o2 * 0x00010001u + 0x0c010c00u, sl3 = (uint32_t)o3 * 0x00010001u + 0x0c010c00u;
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t Dw2 __attribute__((ext_vector_type(2), aligned(4)));   // 4-byte aligned pair: ds_read2_b32
                    auto hsum = [](uint32_t hi, uint32_t lo, uint32_t sel, uint32_t w) {
                        return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(hi, lo, sel)), __builtin_bit_cast(us2, w), 0u, false);
                    };
#pragma nounroll
                    for (int dy = dy0 + rc; dy < dy1; dy += RC) {
                        const int2 ye = yt[dy - dy0];
                        const uint32_t b0s = (uint32_t)ye.y, b1s = (2048u << 8) - b0s;  // vertical weights << 8 (the two sum to 2048)
                        const uint32_t r0 = sw_opaque((uint32_t)ye.x & 0xFFFFu), r1 = sw_opaque((uint32_t)ye.x >> 16);
                        typedef const __attribute__((address_space(3))) Dw2 *LdsDw2;
                        const Dw2 A0 = *(LdsDw2)(uintptr_t)(r0 + (uint32_t)b01), C0 = *(LdsDw2)(uintptr_t)(r0 + (uint32_t)b23);
                        const Dw2 A1 = *(LdsDw2)(uintptr_t)(r1 + (uint32_t)b01), C1 = *(LdsDw2)(uintptr_t)(r1 + (uint32_t)b23);
                        const uint32_t h0[4] = {hsum(A0.y, A0.x, sl0, wp[0]), hsum(A0.y, A0.x, sl1, wp[1]), hsum(C0.y, C0.x, sl2, wp[2]), hsum(C0.y, C0.x, sl3, wp[3])};
                        const uint32_t h1[4] = {hsum(A1.y, A1.x, sl0, wp[0]), hsum(A1.y, A1.x, sl1, wp[1]), hsum(C1.y, C1.x, sl2, wp[2]), hsum(C1.y, C1.x, sl3, wp[3])};
                        uint32_t packed = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t s2 = mul_hi_u24(h0[j] & 0xFFFFFF00u, b0s) + mul_hi_u24(h1[j] & 0xFFFFFF00u, b1s) + 2u;  // <= 1022
                            if (j == 0) packed = s2 >> 2;
                            else if (j == 1) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
                            else if (j == 2) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
                            else asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
                        }
                        *(uint32_t *)(dstF + (uint32_t)(dy * dstPitch + q * 4)) = packed;
                    }
                    continue;
                }
#endif
#pragma nounroll
                for (int dy = dy0 + rc; dy < dy1; dy += RC) {
                    const int2 ye = yt[dy - dy0];
                    const uint32_t b0s = (uint32_t)ye.y, b1s = (2048u << 8) - b0s;  // vertical weights << 8 (the two sum to 2048)
                    // The 16 taps (2 rows x 4 pixels x 2 neighbouring bytes at an arbitrary address) as hand-written ds_read_u8 pairs with
                    // offset:0 / offset:1.  Left to the compiler, its load-store vectoriser merges a pair into ONE ds_read_u16 - which the
                    // LDS executes several times slower when the address is odd (the stage went from 1.66 to 3.56 ms when it started to).
                    // LDS addresses of the two source rows (opaque: folded into the eight address additions as SDWA word selects they
                    // make every one of them a slow-class instruction)
                    const uint32_t r0 = sw_opaque((uint32_t)ye.x & 0xFFFFu), r1 = sw_opaque((uint32_t)ye.x >> 16);
                    uint32_t t0[4], t1[4], u0[4], u1[4];   // row 0 left / right tap, row 1 left / right tap
                    asm volatile("ds_read_u8 %0, %16\n\tds_read_u8 %4, %16 offset:1\n\tds_read_u8 %8, %20\n\tds_read_u8 %12, %20 offset:1\n\t"
                                 "ds_read_u8 %1, %17\n\tds_read_u8 %5, %17 offset:1\n\tds_read_u8 %9, %21\n\tds_read_u8 %13, %21 offset:1\n\t"
                                 "ds_read_u8 %2, %18\n\tds_read_u8 %6, %18 offset:1\n\tds_read_u8 %10, %22\n\tds_read_u8 %14, %22 offset:1\n\t"
                                 "ds_read_u8 %3, %19\n\tds_read_u8 %7, %19 offset:1\n\tds_read_u8 %11, %23\n\tds_read_u8 %15, %23 offset:1\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : "=&v"(t0[0]), "=&v"(t0[1]), "=&v"(t0[2]), "=&v"(t0[3]), "=&v"(t1[0]), "=&v"(t1[1]), "=&v"(t1[2]), "=&v"(t1[3]),
                                   "=&v"(u0[0]), "=&v"(u0[1]), "=&v"(u0[2]), "=&v"(u0[3]), "=&v"(u1[0]), "=&v"(u1[1]), "=&v"(u1[2]), "=&v"(u1[3])
                                 : "v"(r0 + ca[0]), "v"(r0 + ca[1]), "v"(r0 + ca[2]), "v"(r0 + ca[3]), "v"(r1 + ca[0]), "v"(r1 + ca[1]),
                                   "v"(r1 + ca[2]), "v"(r1 + ca[3])
                                 : "memory");
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // cv::resize: ((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2 >> 2 with h = S[sx] * a0 + S[sx + 1] * a1.
                        // Every factor is below 2^24, so the 24-bit multipliers do: h << 4 directly from the pre-shifted weights
                        // (v_mul_u32_u24 + v_mad_u32_u24), and (b * (h >> 4)) >> 16 = ((b << 8) * ((h >> 4) << 8)) >> 32 = ONE v_mul_hi_u32_u24
                        // on (h << 4) & ~0xFF.  v_mul_lo_u32, which the compiler emits for the plain expression, issues at ~8 cycles.
                        const uint32_t h0 = __umul24(t0[j], (uint32_t)wa[j]) + __umul24(t1[j], (uint32_t)wb[j]);
                        const uint32_t h1 = __umul24(u0[j], (uint32_t)wa[j]) + __umul24(u1[j], (uint32_t)wb[j]);
                        // <= 255 after the shift: the weights of each axis sum to 2048
                        const uint32_t s2 = mul_hi_u24(h0 & 0xFFFFFF00u, b0s) + mul_hi_u24(h1 & 0xFFFFFF00u, b1s) + 2u;  // <= 1022
#if PF_RESIZE_PACK_SDWA
                        // byte j of `packed` = s2 >> 2, written in place by the shift itself (SDWA destination select, the other bytes
                        // preserved): one instruction per pixel instead of shift + shift-add + mask + or
                        if (j == 0) packed = s2 >> 2;
                        else if (j == 1) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
                        else if (j == 2) asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
                        else asm("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(packed) : "v"(two), "v"(s2));
#else
                        packed |= (s2 >> 2) << (8 * j);
#endif
                    }
                    *(uint32_t *)(dstF + (uint32_t)(dy * dstPitch + q * 4)) = packed;
                }
            }
        }
    }

    PF_STAMP(3);
    if (DBG && (K().phases >> 4) == 3) return;
#if PF_DUMMY_SALU
    {   // calibration (tools/experiments): what N more scalar ALU instructions per wave cost the kernel
        unsigned dummy = 0;
        asm volatile(".rept %1\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(dummy) : "n"(PF_DUMMY_SALU));
    }
#endif
    // ---- phase 2: FAST-9/16 + cell-local NMS, in the reference's two rounds -----------------------------------------------
    // round 0: threshold iniThFAST everywhere; round 1: threshold minThFAST, only in cells that produced no keypoint
    // in round 0 (tileCalcKeypoints :908-925).  Each round: (A) 4-point pre-test on 8-pixel strips -> LDS queue of
    // 4-pixel groups; (B) dense full 16-position SWAR ring test, one queued group per lane -> LDS queue of corners;
    // (C) dense corner score; (D) dense cell-local strict 3x3 NMS -> per-cell row bitmaps; (E) per-cell counts.
    {
        KArgs &a = K();
        TileK &t = T();
        const int tp = a.g.tilePitch, wCell = a.g.wCell, hCell = a.g.hCell, wpc = a.g.wpc;
        const int sp = a.g.scorePitch;
        uint8_t *score = lds + a.g.oScore - t.cInt0;  // interior column x of the tile = LDS column cInt0 + x
        uint32_t *bits = (uint32_t *)(lds + a.g.oBits);
        uint16_t *gq = (uint16_t *)(lds + a.g.oGq), *cq = (uint16_t *)(lds + a.g.oCq);
        uint8_t *colInfo = lds + a.g.oColInfo;
        const int phases = DBG ? a.phases : 15;
        // (the prefix-count allocation lives in the debug instance only: both paths in the production kernel cost it four spilled
        // registers - 16 bytes of scratch per lane, +0.3 GB of HBM writes per launch set)
        const bool laneAlloc = !DBG || a.laneAlloc != 0;
        const int minTh = min(max(a.minTh, 0), 255), iniTh = min(max(a.iniTh, 0), 255);
        const int cInt0 = t.cInt0, cInt1 = t.cInt1, nCells = t.nCells, THi = t.THi, rowOff = t.rowOff;
        const int s0 = t.s0, NS = t.NS;  // strips cover LDS columns [8s + 4, 8s + 12)
        // The group queue may hold fewer entries than the tile has groups (OrbxLevelGeom::gqCap: what keeps a fourth workgroup resident
        // per CU): entries beyond it land in the dump slot gq[gqCap], and a round whose count exceeds the capacity runs the ring
        // test over EVERY group of the tile instead of the queued ones (pathological corner density only).
        const unsigned gqCap = (unsigned)a.g.gqCap;
        if (t.bandTop + t.bandBot) {
            // an edge tile staged its margin bands where the score tile and the bitmaps live: once every wave is through the resize,
            // zero them here (the other tiles did under the latency of their tile load)
            __syncthreads();
            uint4 z = make_uint4(0, 0, 0, 0);
            uint4 *zs = (uint4 *)(lds + a.g.oScore);
            const int nz = (a.g.oXt - a.g.oScore) >> 4;
#pragma nounroll
            for (int i = otid(); i < nz; i += PF_NT) zs[i] = z;
        }
        // (1 / NS is formed where it is used, from a copy of NS the optimiser cannot see through: hoisted out of the round loop it stayed
        // live - and spilled - through both rounds)
        const int nItems = NS * THi;
        for (int round = 0; round < 2; ++round) {
            const int th = round == 0 ? iniTh : minTh;
            const int thEff = max(th, 1);
            unsigned eMask = 0;  // round 1: cells without a keypoint after round 0 (cnt is complete: barrier at the end of the NMS phase)
            if (round == 1) {
                // lane c looks at cell c (a scalar loop over the cells is unrolled by the compiler into a chain of wave-uniform
                // conditions that all stay live - and spilled - through the rounds)
                eMask = (unsigned)__ballot(lane < nCells && cnt[otid() & (PF_MAX_GROUP_CELLS - 1)] == 0);
                if (minTh >= iniTh || !eMask) break;  // uniform
            }
            const uint32_t T4 = (uint32_t)th * 0x01010101u, C4 = (uint32_t)(255 - th) * 0x01010101u;

            // (A) pre-test of one 8-pixel strip (row r, strip sI of the tile, `valid` = its candidate pixels)
            auto preTest = [&](int r, int sI, const uint8_t *wp) {  // wp = tile + (rowOff + r - 3) * tp + 8 * (s0 + sI)
                uint32_t w0[4], w3[4], w6[4];
                {
                    const uint2 a0 = *(const uint2 *)(wp), a1 = *(const uint2 *)(wp + 8);
                    const uint2 b0 = *(const uint2 *)(wp + 3 * tp), b1 = *(const uint2 *)(wp + 3 * tp + 8);
                    const uint2 c0 = *(const uint2 *)(wp + 6 * tp), c1 = *(const uint2 *)(wp + 6 * tp + 8);
                    w0[0] = a0.x; w0[1] = a0.y; w0[2] = a1.x; w0[3] = a1.y;
                    w3[0] = b0.x; w3[1] = b0.y; w3[2] = b1.x; w3[3] = b1.y;
                    w6[0] = c0.x; w6[1] = c0.y; w6[2] = c1.x; w6[3] = c1.y;
                }
                const uint32_t f0 = fast_group_pretest<0>(w0, w3, w6, T4, C4);
                const uint32_t f1 = fast_group_pretest<1>(w0, w3, w6, T4, C4);
                // (pixels outside the interior columns / the still-empty cells are not masked here: the ring test clips its result to
                // both anyway, and the mask cost every strip ~15 instructions to keep a few edge groups out of the queue)
                // queue slots: one LDS atomic per wave and ballot prefix counts (a per-lane atomicAdd on one address is expanded by
                // the compiler into a readlane loop over the active lanes: ~9 scalar instructions per passing strip)
                if (laneAlloc) {
                    // (one test of both flag words in front of the block, the first entry stored unconditionally inside it: two compares,
                    // two selects and an exec round trip less per strip than "g0 = f0 != 0, g1 = f1 != 0; if (g0 | g1) ... if (g0) ... if (g1)")
                    if ((f0 | f1) != 0) {
                        const unsigned n0 = f0 != 0 ? 1u : 0u, n1 = f1 != 0 ? 1u : 0u;
                        const unsigned pos = lane_queue_alloc(&gqCountR[round], n0 + n1);
                        const unsigned e = ((unsigned)r << 9) | (unsigned)(2 * sI);
                        gq[min(pos, gqCap)] = (uint16_t)(e | (n0 ^ 1u));           // the first passing group of the strip
                        if (n0 & n1) gq[min(pos + 1u, gqCap)] = (uint16_t)(e | 1u);
                    }
                    return;
                }
                const bool g0 = f0 != 0, g1 = f1 != 0;
                const unsigned long long bg0 = __ballot(g0), bg1 = __ballot(g1);
                if (bg0 | bg1) {
                    unsigned pos = wave_queue_alloc(&gqCountR[round], (unsigned)(__popcll(bg0) + __popcll(bg1))) + mbcnt64(bg0) + mbcnt64(bg1);
                    const unsigned e = ((unsigned)r << 9) | (unsigned)(2 * sI);
                    if (g0) { gq[min(pos, gqCap)] = (uint16_t)e; ++pos; }
                    if (g1) gq[min(pos, gqCap)] = (uint16_t)(e | 1u);
                }
            };
            if (!(phases & 2)) {
            } else if (round == 0) {
                // One strip COLUMN per lane for the whole round: the 512 lanes are dealt over (row class, strip) - lane t -> strip t mod NS,
                // rows t / NS, + dR, + 2 dR ... with dR = 512 / NS row classes (NS < 512: checked by the host) - like the resize phase.
                // The walk is a row step and an address step; as item it = r * NS + sI with (r, sI, address) as induction variables it
                // carried a wrap test with two selects per strip (11 against 3 vector instructions), and divided by NS through a
                // float reciprocal per item, as it first was, two v_cvt, four v_mul_lo_u32 and a dozen adds.
                const int it = otid();
                int nsO = NS;
                asm volatile("" : "+s"(nsO));
                const float invNS = __builtin_amdgcn_rcpf((float)nsO);
                int r = (int)(((float)it + 0.5f) * invNS);
                const int sI = it - r * NS;
                const int dR = (int)(((float)PF_NT + 0.5f) * invNS);   // wave-uniform
                const uint8_t *wp = tile + (rowOff + r - 3) * tp + 8 * (s0 + sI);
                const int stepW = dR * tp;
                if (r < dR)
                    for (; r < THi; r += dR, wp += stepW) preTest(r, sI, wp);
            } else {
                // only the strips of still-empty cells, one cell at a time (cell geometry in SGPRs).  A strip that straddles two
                // empty cells is tested once, with the left one.
                int visit = 0;
                for (unsigned em = eMask; em; em &= em - 1, ++visit) {
                    const int c = __ffs(em) - 1;
                    const bool prevEmpty = c > 0 && ((eMask >> (c - 1)) & 1u);
                    const int cc0 = cInt0 + c * wCell, cc1 = min(cc0 + wCell, cInt1);
                    int sF = (cc0 - 4) >> 3;
                    const int sL = (cc1 - 1 - 4) >> 3;
                    if (prevEmpty && 8 * sF + 4 < cc0) ++sF;
                    const int nS = sL - sF + 1;
                    if (nS <= 0) continue;
                    const float invS = __builtin_amdgcn_rcpf((float)nS);
                    // alternate the half of the workgroup that takes a cell's (usually < 256) strips
                    for (int it = (otid() + visit * (PF_NT / 2)) & (PF_NT - 1); it < nS * THi; it += PF_NT) {
                        const int r = (int)(((float)it + 0.5f) * invS);
                        const int sI = sF + (it - r * nS) - s0;
                        preTest(r, sI, tile + (rowOff + r - 3) * tp + 8 * (s0 + sI));
                    }
                }
            }
            __syncthreads();
            PF_STAMP(4 + 4 * round);
            if (DBG && (phases >> 4) == 4) return;

            // (B) full ring test of the queued groups -> corner queue
            const unsigned gn = gqCountR[round];
            // (queue overflow: every group of the tile, i -> (row, 2 * strip + half) in raster order - the same loop, so that the ring
            // test exists once in the kernel)
            const bool dense = gn > gqCap;
            const unsigned gnEff = dense ? (unsigned)(2 * nItems) : gn;
            for (unsigned i = otid(); i < gnEff; i += PF_NT) {
                unsigned e;
                if (dense) {
                    int ns2 = 2 * NS;
                    asm volatile("" : "+s"(ns2));
                    const unsigned rr = (unsigned)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)ns2));
                    e = (rr << 9) | (i - rr * (unsigned)ns2);
                } else {
                    e = gq[i];
                }
                const int r = e >> 9, sg = e & 511;
                const int cb = 8 * s0 + 4 * sg;                       // window start: the group's pixels are bytes 4..7
                const uint32_t fl = fast_group_flags_lds(tile + (rowOff + r - 3) * tp + cb, tp, T4, C4);
                const int c4 = cb + 4;
                const int plo = max(cInt0 - c4, 0), phi = min(cInt1 - c4, 4);
                unsigned m = ((fl * 0x00204081u) >> 28) & (0xFu >> (4 - phi)) & (0xFu << plo);
                if (round == 1 && m) {
                    // keep the pixels that lie in a still-empty cell; a group overlaps at most two cells (cells are >= 35 px wide)
                    const int x0 = c4 + plo - cInt0, n = phi - plo;
                    const int cl = colInfo[x0] & 0x3F, ch = colInfo[x0 + n - 1] & 0x3F;
                    const int b = min(max(ch * wCell - x0, 0), n);  // first pixel of cell ch (0 when cl == ch)
                    const unsigned all = (1u << n) - 1u, lo = (1u << b) - 1u;
                    const unsigned keep = (((eMask >> cl) & 1u) ? lo : 0u) | (((eMask >> ch) & 1u) ? all & ~lo : 0u);
                    m &= keep << plo;
                }
                const unsigned nc = __popc(m);  // 0..4
                if (laneAlloc) {
                    if (nc) {
                        unsigned pos = lane_queue_alloc(&cqCountR[round], nc);
                        while (m) {
                            const int k = __ffs(m) - 1;
                            m &= m - 1;
                            if (pos < PF_QCAP) cq[pos] = (uint16_t)((r << 9) | (c4 + k - cInt0));
                            else pf_score_one(tile, score, tp, sp, rowOff, r, c4 + k, thEff);  // queue overflow: score in place
                            ++pos;
                        }
                    }
                    continue;
                }
                const unsigned long long n1 = __ballot((nc & 1u) != 0), n2 = __ballot((nc & 2u) != 0), n4 = __ballot((nc & 4u) != 0);
                if (n1 | n2 | n4) {
                    const unsigned total = (unsigned)(__popcll(n1) + 2 * __popcll(n2) + 4 * __popcll(n4));
                    const unsigned base = wave_queue_alloc(&cqCountR[round], total);
                    unsigned pos = base + mbcnt64(n1) + 2u * mbcnt64(n2) + 4u * mbcnt64(n4);
                    if (base + total <= PF_QCAP) {
                        // (the usual case, wave-uniform) four predicated stores: a loop over the set bits runs as often as the lane
                        // with the most corners needs, ~24 vector instructions per turn
                        const int e0 = (r << 9) + (c4 - cInt0);  // (c4 - cInt0 may be negative for a group that straddles the left edge)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (m & (1u << k)) cq[pos++] = (uint16_t)(e0 + k);
                    } else {
                        while (m) {
                            const int k = __ffs(m) - 1;
                            m &= m - 1;
                            if (pos < PF_QCAP) cq[pos] = (uint16_t)((r << 9) | (c4 + k - cInt0));
                            else pf_score_one(tile, score, tp, sp, rowOff, r, c4 + k, thEff);  // queue overflow: score in place
                            ++pos;
                        }
                    }
                }
            }
            __syncthreads();
            PF_STAMP(5 + 4 * round);
            if (DBG && (phases >> 4) == 5) return;

            // (C) dense corner score (merging it into (B) measured slower: 3.26 vs 3.0 ms — divergent scoring loops)
            const unsigned qn = cqCountR[round];
            for (unsigned i = otid(); i < min(qn, (unsigned)PF_QCAP); i += PF_NT) {
                const unsigned e = cq[i];
                pf_score_one(tile, score, tp, sp, rowOff, e >> 9, (int)(e & 511) + cInt0, thEff);
            }
            __syncthreads();
            PF_STAMP(6 + 4 * round);
            if (DBG && (phases >> 4) == 6) return;

            // (D) cell-local NMS -> bitmaps; (E) per-cell counts by LDS atomics
            if (qn <= PF_QCAP) {
                if (phases & 4)
                for (unsigned i = otid(); i < qn; i += PF_NT) {
                    const unsigned e = cq[i];
                    pf_nms_one(score, colInfo, bits, cnt, sp, e >> 9, (int)(e & 511) + cInt0, cInt0, wCell, hCell, wpc);
                }
            } else {
                // corner queue overflow (pathological corner density): NMS over every scored pixel of the queued groups (of every group
                // when the group queue overflowed too)
                for (unsigned i = otid(); i < gnEff; i += PF_NT) {
                    unsigned e;
                    if (dense) {
                        int ns2 = 2 * NS;
                        asm volatile("" : "+s"(ns2));
                        const unsigned rr = (unsigned)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)ns2));
                        e = (rr << 9) | (i - rr * (unsigned)ns2);
                    } else {
                        e = gq[i];
                    }
                    const int r = e >> 9, c4 = 8 * s0 + 4 * (int)(e & 511) + 4;
                    for (int k = 0; k < 4; ++k)
                        if (c4 + k >= cInt0 && c4 + k < cInt1)
                            pf_nms_one(score, colInfo, bits, cnt, sp, r, c4 + k, cInt0, wCell, hCell, wpc);
                }
            }
            __syncthreads();
            PF_STAMP(7 + 4 * round);
            if (DBG && (phases >> 4) == 7) return;
        }
        if (!(phases & 8)) return;
    }

    // ---- phase 3: ordered emission: cells in order, raster inside a cell (== tileCalcKeypoints emission order) ----
    {
        KArgs &a = K();
        TileK &t = T();
        const int sp = a.g.scorePitch, wCell = a.g.wCell, wpc = a.g.wpc;
        const uint8_t *score = lds + a.g.oScore - t.cInt0;
        const uint32_t *bits = (const uint32_t *)(lds + a.g.oBits);
        const int nCells = t.nCells, THi = t.THi, cInt0 = t.cInt0;
        const int xBias = t.colOf0 + ORBX_BORDER, yBias = t.ty0 - ORBX_BORDER;
        uint32_t *slot = a.cand + (size_t)f * a.candFrameStride + a.g.candBase + (size_t)tileId * a.g.slotCap;
        // One (cell, row) pair per lane, pairs in emission order (cell-major): the exclusive prefix of the rows' keypoint counts over the
        // workgroup - a DPP scan per wave, the eight wave totals through LDS - is each row's offset in the slot.  (A wave per cell with a
        // scan per cell left one wave with two of a tile's nine cells and the others idle: the emission was 2 - 3 k of a workgroup's
        // 32 k cycles.)  The wave totals go where the group queue was (dead since the barrier behind the last NMS).
        const int hCell = a.g.hCell, nPairs = nCells * THi;
        int *wtot = (int *)(lds + a.g.oGq);
        int thO = THi;
        asm volatile("" : "+s"(thO));
        const float invTH = __builtin_amdgcn_rcpf((float)thO);
        int base = 0, chunk = 0;
#pragma nounroll
        for (int p0 = 0; p0 < nPairs; p0 += PF_NT, chunk ^= 1) {   // (one turn unless a tile has more than 512 cell rows)
            const int p = p0 + otid();
            const bool live = p < nPairs;
            const int c = live ? (int)(((float)p + 0.5f) * invTH) : 0, r = live ? p - c * THi : 0;
            const uint32_t *rbits = bits + (size_t)(c * hCell + r) * wpc;
            int n = 0;
            if (live)
                for (int k = 0; k < wpc; ++k) n += __popc(rbits[k]);
            int wsum;
            int o = wave_excl_scan(n, wsum);
            if (lane == 0) wtot[chunk * PF_NW + wave] = wsum;
            __syncthreads();
            int before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < PF_NW; ++w) {
                const int v = wtot[chunk * PF_NW + w];
                all += v;
                before += w < wave ? v : 0;
            }
            o += base + before;
            if (n) {
                const int cellC0 = cInt0 + c * wCell;
                for (int k = 0; k < wpc; ++k) {
                    uint32_t wd = rbits[k];
                    while (wd) {
                        const int b = __ffs(wd) - 1;
                        wd &= wd - 1;
                        const int cc = cellC0 + k * 32 + b;
                        slot[o++] = ORBX_PACK(cc - xBias, yBias + r, score[(r + 1) * sp + cc]);
                    }
                }
            }
            base += all;
        }
        if (tid == 0) a.tileCount[(size_t)f * a.tilesPerFrame + a.g.tileBase + tileId] = base;
    }
    PF_STAMP(12);
}

// =====================================================================================================
// k_pyr_fast_cw — the same fused pyramid level + tileCalcKeypoints, with the FAST rounds organised by CELL instead of by phase.
//
// tileCalcKeypoints (reference src/ORBextractor.cc:867-950) is independent per 35-px cell: cv::FAST on the cell's ROI, NMS inside the
// ROI, the retry with minThFAST if THAT cell found nothing (:908-925), emission cell by cell in raster order (:941-944).  k_pyr_fast runs
// every step of that as a phase of the whole workgroup (pre-test | ring test | score | NMS, with a barrier and shared LDS queues
// between them, a second set of phases for the empty cells, then an emission pass over per-cell bitmaps): the sparse phases keep 2 - 5
// of the 8 waves busy, and a workgroup lived 33 k cycles for ~1070 vector instructions per wave (tools/pf_timing.py).
// Here, once the tile and the resize tables are in LDS, nothing is shared any more:
//   * all waves resize (dense, as before), then wave c takes cell c - no barrier in between;
//   * a cell is ONE order-preserving stream through one wave: 4-pixel groups in raster order -> pre-test -> ring of survivors (wave
//     private, 128 entries) -> full ring test whenever 64 are queued (always a full wave) -> list of corners of the cell, still in
//     raster order -> scores (dense, 64 at a time) -> cell-local 3x3 NMS compacts the list in place.  Queue slots are ballot prefix
//     counts on wave-uniform counters: no LDS atomics, no bitmaps, no per-column cell table;
//   * an empty cell simply runs again with minThFAST on its own wave, while the other waves are done;
//   * what is left of the list IS the cell's keypoints in emission order: after the one barrier that makes the per-cell counts
//     visible, every wave copies its cells' lists to the tile's candidate slot at the cells' prefix offsets.
// Three barriers per workgroup instead of six (ten with a second round).  A cell with more corners than the list holds (cwClCap,
// pathological density) scores in place and takes NMS + emission by scanning its score rows (same order, same result).
// =====================================================================================================
#define PFC_GQ 128  // entries of a wave's ring of pre-test survivors (row << 9 | LDS column / 4)

// pre-test of ONE 4-pixel group: fast_group_pretest_v (centre dword V with its left / right neighbours in the row, the dwords three
// rows below (ring position 0) and above (8))
#define fast_group_pretest1 fast_group_pretest_v

// strict 3x3 maximum inside the cell's columns [cc0, cc1) (cv::FAST's nonmax suppression on the cell's ROI): score row r + 1 of the
// score tile is tile row r; rows 0 and THi + 1 stay zero
__device__ __forceinline__ bool pfc_is_max(const uint8_t *score, int sp, int r, int c, int cc0, int cc1)
{
    const uint8_t *s1p = score + (r + 1) * sp + c, *s0p = s1p - sp, *s2p = s1p + sp;
    const int sc = s1p[0];
    if (sc <= 0) return false;
    int mx = max(s0p[0], s2p[0]);
    if (c != cc0) mx = max(mx, max(s0p[-1], max(s1p[-1], s2p[-1])));
    if (c != cc1 - 1) mx = max(mx, max(s0p[1], max(s1p[1], s2p[1])));
    return sc > mx;
}

template <bool DBG>
__global__ __launch_bounds__(PF_NT, 8) void k_pyr_fast_cw(PyrFastArgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    typedef const __attribute__((address_space(4))) PyrFastArgs KArgs;
    typedef const __attribute__((address_space(4))) OrbxPfTile TileK;
    int f, tileId;
    auto K = [&]() -> KArgs & {
        KArgs *p = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(p));
        return *p;
    };
    {
        KArgs &a = K();
        if (!xcd_decode(a.nframes, a.g.nTiles, f, tileId)) return;
    }
    auto T = [&]() -> TileK & { TileK *p = (TileK *)(unsigned long long)K().tiles + tileId; asm volatile("" : "+s"(p)); return *p; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: everything per cell stays in SGPRs
    auto otid = [&]() { int t = tid; asm volatile("" : "+v"(t)); return t; };
    __shared__ int cnt[PF_MAX_GROUP_CELLS];  // keypoints per cell (bit 31: the cell took the scanning path), written by the cell's wave
    uint8_t *const tile = lds;

    // ---- phase 0: stage the tile and the resize tables (as k_pyr_fast) ------------------------------------------------------------
    {
        KArgs &a = K();
        TileK &t = T();
        const int tp = a.g.tilePitch;
        const int ly0 = t.ly0, nRowsL = t.nRowsL, ax0 = t.ax0, nV = t.nV, nVec = t.nVec;
        const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
        const bool doResize = a.dst != nullptr;
        const int ndx = doResize ? t.dx1 - t.dx0 : 0, ndy = doResize ? t.dy1 - t.dy0 : 0, dx0 = t.dx0, dy0 = t.dy0;
        const float invNV = __builtin_amdgcn_rcpf((float)nV);
        uint4 tv0 = make_uint4(0, 0, 0, 0), tv1 = tv0;
        int tvOff0 = -1, tvOff1 = -1;
        if (a.srcAlign == 16) {
            const int it0 = otid(), it1 = it0 + PF_NT;
            if (it0 < nVec) {
                const int r = (int)(((float)it0 + 0.5f) * invNV), c = it0 - r * nV;
                tv0 = *(const uint4 *)(srcF + (size_t)(ly0 + r) * a.srcPitch + ax0 + c * 16);
                tvOff0 = r * tp + PF_XO + c * 16;
            }
            if (it1 < nVec) {
                const int r = (int)(((float)it1 + 0.5f) * invNV), c = it1 - r * nV;
                tv1 = *(const uint4 *)(srcF + (size_t)(ly0 + r) * a.srcPitch + ax0 + c * 16);
                tvOff1 = r * tp + PF_XO + c * 16;
            }
        }
        int xo_ = 0, xw_ = 0, yo_ = 0, yw_ = 0;
        if (doResize) {
            const int i = otid();
            if (i < ndx) { xo_ = a.tabs.xofs[dx0 + i]; xw_ = a.tabs.xab[dx0 + i]; }
            if (i < ndy) { yo_ = a.tabs.yofs[dy0 + i]; yw_ = a.tabs.yab[dy0 + i]; }
        }
        {   // zero the score tile (under the latency of the loads above)
            uint4 z = make_uint4(0, 0, 0, 0);
            uint4 *zs = (uint4 *)(lds + a.g.oScore);
            const int nz = (a.g.cwXt - a.g.oScore) >> 4;
#pragma nounroll
            for (int i = otid(); i < nz; i += PF_NT) zs[i] = z;
        }
        if (a.srcAlign == 16) {
            if (tvOff0 >= 0) *(uint4 *)(tile + tvOff0) = tv0;
            if (tvOff1 >= 0) *(uint4 *)(tile + tvOff1) = tv1;
#pragma nounroll
            for (int it = otid() + 2 * PF_NT; it < nVec; it += PF_NT) {
                const int r = (int)(((float)it + 0.5f) * invNV);
                const int c = it - r * nV;
                const uint4 v = *(const uint4 *)(srcF + (size_t)(ly0 + r) * a.srcPitch + ax0 + c * 16);
                *(uint4 *)(tile + r * tp + PF_XO + c * 16) = v;
            }
        } else if (a.srcAlign == 4) {
            const int nDw = (t.lx1 - ax0 + 3) >> 2;
#pragma nounroll
            for (int r = wave; r < nRowsL; r += PF_NW) {
                const uint32_t *srow = (const uint32_t *)(srcF + (size_t)(ly0 + r) * a.srcPitch + ax0);
                uint32_t *trow = (uint32_t *)(tile + r * tp + PF_XO);
#pragma nounroll
                for (int c = otid() & 63; c < nDw; c += 64) trow[c] = srow[c];
            }
        } else {
            const int nB = t.lx1 - ax0;
#pragma nounroll
            for (int r = wave; r < nRowsL; r += PF_NW) {
                const uint8_t *srow = srcF + (size_t)(ly0 + r) * a.srcPitch + ax0;
                uint8_t *trow = tile + r * tp + PF_XO;
#pragma nounroll
                for (int c = otid() & 63; c < nB; c += 64) trow[c] = srow[c];
            }
        }
        if (DBG && (a.phases >> 4) == 1) return;
        if (doResize) {
            int2 *xt = (int2 *)(lds + a.g.cwXt), *yt = (int2 *)(lds + a.g.cwYt);
            const int colOf0 = t.colOf0, hm1 = a.g.h - 1;
            auto xEntry = [&](int sx, int w2) { return make_int2(sx + colOf0, w2); };
            auto yEntry = [&](int sy, int w2) {
                const int sy0 = min(max(sy, 0), hm1) - ly0, sy1 = min(max(sy + 1, 0), hm1) - ly0;
                return make_int2((sy0 * tp) | ((sy1 * tp) << 16), w2);
            };
            const int i0 = otid();
            if (i0 < ndx) xt[i0] = xEntry(xo_, xw_);
            if (i0 < ndy) yt[i0] = yEntry(yo_, yw_);
#pragma nounroll
            for (int i = i0 + PF_NT; i < ndx; i += PF_NT) xt[i] = xEntry(a.tabs.xofs[dx0 + i], a.tabs.xab[dx0 + i]);
#pragma nounroll
            for (int i = i0 + PF_NT; i < ndy; i += PF_NT) yt[i] = yEntry(a.tabs.yofs[dy0 + i], a.tabs.yab[dy0 + i]);
        }
    }
    __syncthreads();
    if (DBG && (K().phases >> 4) == 2) return;

    // ---- phase 1: next pyramid level (as k_pyr_fast) ------------------------------------------------------------------------------
    {
        KArgs &a = K();
        if (a.dst) {
            TileK &t = T();
            const int2 *xt = (const int2 *)(lds + a.g.cwXt), *yt = (const int2 *)(lds + a.g.cwYt);
            uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
            const int dstPitch = a.dstPitch;
            const int dx0 = t.dx0, dx1 = t.dx1, dy0 = t.dy0, dy1 = t.dy1;
            const int q0 = t.q0, nq = t.nq, RC = t.RC, q1 = q0 + nq;
            const int tt = otid();
            const int rc = nq >= PF_NT ? 0 : (int)(((float)tt + 0.5f) * __builtin_amdgcn_rcpf((float)nq));
            for (int q = q0 + (tt - rc * nq); q < q1 && rc < RC; q += PF_NT) {
                int ca[4], wa[4], wb[4];
                bool ok[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dx = q * 4 + j;
                    ok[j] = dx >= dx0 && dx < dx1;
                    const int2 e = xt[min(max(dx, dx0), dx1 - 1) - dx0];
                    ca[j] = e.x;
                    wa[j] = (short)(e.y & 0xFFFF);
                    wb[j] = e.y >> 16;
                }
                const bool full = ok[0] && ok[3];
#pragma nounroll
                for (int dy = dy0 + rc; dy < dy1; dy += RC) {
                    const int2 ye = yt[dy - dy0];
                    const int b0 = ye.y & 0xFFFF, b1 = (unsigned)ye.y >> 16;
                    const uint8_t *r0 = tile + (ye.x & 0xFFFF);
                    const uint8_t *r1 = tile + ((unsigned)ye.x >> 16);
                    uint32_t packed = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int h0 = r0[ca[j]] * wa[j] + r0[ca[j] + 1] * wb[j];
                        const int h1 = r1[ca[j]] * wa[j] + r1[ca[j] + 1] * wb[j];
                        const uint32_t v = (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
                        packed |= v << (8 * j);
                    }
                    uint8_t *drow = dstF + (size_t)dy * dstPitch + q * 4;
                    if (full) *(uint32_t *)drow = packed;
                    else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ok[j]) drow[j] = (uint8_t)(packed >> (8 * j));
                    }
                }
            }
        }
    }
    if (DBG && (K().phases >> 4) == 3) return;

    // ---- phase 2: FAST-9/16 + NMS, a wave per cell, both rounds ---------------------------------------------------------------------
    {
        KArgs &a = K();
        TileK &t = T();
        const int tp = a.g.tilePitch, sp = a.g.scorePitch, wCell = a.g.wCell, clCap = a.g.cwClCap;
        uint8_t *score = lds + a.g.oScore - t.cInt0;  // indexed with LDS columns like the pixel tile
        uint16_t *gq = (uint16_t *)(lds + a.g.cwGq) + wave * PFC_GQ;
        const int cInt0 = t.cInt0, cInt1 = t.cInt1, nCells = t.nCells, THi = t.THi, rowOff = t.rowOff;
        const int minTh = min(max(a.minTh, 0), 255), iniTh = min(max(a.iniTh, 0), 255);
        const int stop = DBG ? a.phases >> 4 : 0;
        for (int c = wave; c < nCells; c += PF_NW) {
            uint16_t *cl = (uint16_t *)(lds + a.g.cwCl) + c * clCap;
            const int cc0 = cInt0 + c * wCell, cc1 = min(cc0 + wCell, cInt1);
            const int gF = cc0 >> 2, nG = ((cc1 - 1) >> 2) - gF + 1;  // the cell's 4-pixel groups per row (LDS columns [4 g, 4 g + 4))
            const int nItems = nG * THi;
            const float invG = __builtin_amdgcn_rcpf((float)nG);
            int nk = 0;
            bool ovf = false;
            for (int round = 0; round < 2; ++round) {
                const int th = round == 0 ? iniTh : minTh;
                const int thEff = max(th, 1);
                const uint32_t T4 = (uint32_t)th * 0x01010101u, C4 = (uint32_t)(255 - th) * 0x01010101u;
                int it0 = 0, nc = 0, scored = 0;
                unsigned qh = 0, qt = 0;
                ovf = false;
                for (;;) {
                    // (A) pre-test of the next 64 groups (raster order) while fewer than 64 survivors are queued
                    while (it0 < nItems && (int)(qt - qh) < 64) {
                        const int it = it0 + lane;
                        bool pass = false;
                        unsigned e = 0;
                        if (it < nItems) {
                            const int r = (int)(((float)it + 0.5f) * invG), g = gF + (it - r * nG);
                            const uint8_t *wp = tile + (rowOff + r - 3) * tp + 4 * g;
                            const uint32_t *q = (const uint32_t *)(wp + 3 * tp);
                            pass = fast_group_pretest1(q[0], q[-1], q[1], *(const uint32_t *)(wp + 6 * tp), *(const uint32_t *)wp, T4, C4) != 0;
                            e = ((unsigned)r << 9) | (unsigned)g;
                        }
                        const unsigned long long b = __ballot(pass);
                        if (b) {
                            if (pass) gq[(qt + mbcnt64(b)) & (PFC_GQ - 1)] = (uint16_t)e;
                            qt += (unsigned)__popcll(b);
                        }
                        it0 += 64;
                    }
                    const int avail = (int)(qt - qh);
                    const bool last = it0 >= nItems;
                    // (B) full ring test of up to 64 queued groups -> corners of the cell, appended in raster order
                    if (avail >= 64 || (last && avail > 0)) {
                        const int m = min(avail, 64);
                        if (!(DBG && stop == 4)) {
                            unsigned msk = 0;
                            int r = 0, c4 = 0;
                            if (lane < m) {
                                const unsigned e = gq[(qh + lane) & (PFC_GQ - 1)];
                                r = e >> 9;
                                c4 = 4 * (int)(e & 511);
                                const uint32_t fl = fast_group_flags_lds(tile + (rowOff + r - 3) * tp + c4 - 4, tp, T4, C4);
                                const int plo = max(cc0 - c4, 0), phi = min(cc1 - c4, 4);
                                msk = ((fl * 0x00204081u) >> 28) & (0xFu >> (4 - phi)) & (0xFu << plo);
                            }
                            const unsigned ncn = __popc(msk);  // 0..4
                            const unsigned long long n1 = __ballot((ncn & 1u) != 0), n2 = __ballot((ncn & 2u) != 0), n4 = __ballot((ncn & 4u) != 0);
                            if (n1 | n2 | n4) {
                                int pos = nc + (int)(mbcnt64(n1) + 2u * mbcnt64(n2) + 4u * mbcnt64(n4));
                                nc += __popcll(n1) + 2 * __popcll(n2) + 4 * __popcll(n4);
                                while (msk) {
                                    const int k = __ffs(msk) - 1;
                                    msk &= msk - 1;
                                    if (pos < clCap) cl[pos] = (uint16_t)((r << 9) | (c4 + k - cInt0));
                                    else pf_score_one(tile, score, tp, sp, rowOff, r, c4 + k, thEff);  // list full: score in place
                                    ++pos;
                                }
                                if (nc > clCap) ovf = true;
                            }
                        }
                        qh += (unsigned)m;
                    }
                    // (C) scores of the listed corners, 64 at a time
                    const int listed = min(nc, clCap);
                    const bool drained = last && qt == qh;
                    if (listed - scored >= 64 || (drained && listed > scored)) {
                        const int m = min(listed - scored, 64);
                        if (!(DBG && (stop == 4 || stop == 5)) && lane < m) {
                            const unsigned e = cl[scored + lane];
                            pf_score_one(tile, score, tp, sp, rowOff, e >> 9, (int)(e & 511) + cInt0, thEff);
                        }
                        scored += m;
                    }
                    if (drained && scored >= listed) break;
                }
                // (D) cell-local NMS: the list shrinks to the cell's keypoints, order kept
                nk = 0;
                if (DBG && stop >= 4 && stop <= 6) break;
                if (!ovf) {
                    for (int i = 0; i < nc; i += 64) {
                        bool keep = false;
                        unsigned e = 0;
                        if (i + lane < nc) {
                            e = cl[i + lane];
                            keep = pfc_is_max(score, sp, e >> 9, (int)(e & 511) + cInt0, cc0, cc1);
                        }
                        const unsigned long long b = __ballot(keep);
                        if (keep) cl[nk + (int)mbcnt64(b)] = (uint16_t)e;  // (at or below every slot this iteration read)
                        nk += __popcll(b);
                    }
                } else {
                    for (int r = 0; r < THi; ++r)
                        for (int x0 = cc0; x0 < cc1; x0 += 64) {
                            const int cc = x0 + lane;
                            nk += __popcll(__ballot(cc < cc1 && pfc_is_max(score, sp, r, cc, cc0, cc1)));
                        }
                }
                if (nk > 0 || minTh >= iniTh || (DBG && stop)) break;  // the cell found nothing: once more with minThFAST (:925)
            }
            if (lane == 0) cnt[c] = nk | (ovf ? (int)0x80000000 : 0);
        }
        if (DBG && stop) return;
    }
    __syncthreads();

    // ---- phase 3: emission: cells in order, raster inside a cell (== tileCalcKeypoints' emission order) -----------------------------
    {
        KArgs &a = K();
        TileK &t = T();
        const int sp = a.g.scorePitch, wCell = a.g.wCell, clCap = a.g.cwClCap;
        const uint8_t *score = lds + a.g.oScore - t.cInt0;
        const int nCells = t.nCells, THi = t.THi, cInt0 = t.cInt0, cInt1 = t.cInt1;
        const int xBias = t.colOf0 + ORBX_BORDER, yBias = t.ty0 - ORBX_BORDER;
        uint32_t *slot = a.cand + (size_t)f * a.candFrameStride + a.g.candBase + (size_t)tileId * a.g.slotCap;
        const int mine = lane < nCells ? cnt[lane & (PF_MAX_GROUP_CELLS - 1)] : 0;
        int tot;
        const int cellBase = wave_excl_scan(mine & 0x7FFFFFFF, tot);
        if (tid == 0) a.tileCount[(size_t)f * a.tilesPerFrame + a.g.tileBase + tileId] = tot;
        for (int c = wave; c < nCells; c += PF_NW) {
            const int base = __builtin_amdgcn_readlane(cellBase, c), cw = __builtin_amdgcn_readlane(mine, c);
            const int n = cw & 0x7FFFFFFF;
            if (cw >= 0) {
                const uint16_t *cl = (const uint16_t *)(lds + a.g.cwCl) + c * clCap;
                for (int i = lane; i < n; i += 64) {
                    const unsigned e = cl[i];
                    const int r = e >> 9, cc = (int)(e & 511) + cInt0;
                    slot[base + i] = ORBX_PACK(cc - xBias, yBias + r, score[(r + 1) * sp + cc]);
                }
            } else {
                const int cc0 = cInt0 + c * wCell, cc1 = min(cc0 + wCell, cInt1);
                int o = base;
                for (int r = 0; r < THi; ++r)
                    for (int x0 = cc0; x0 < cc1; x0 += 64) {
                        const int cc = x0 + lane;
                        const bool keep = cc < cc1 && pfc_is_max(score, sp, r, cc, cc0, cc1);
                        const unsigned long long b = __ballot(keep);
                        if (keep) slot[o + (int)mbcnt64(b)] = ORBX_PACK(cc - xBias, yBias + r, score[(r + 1) * sp + cc]);
                        o += __popcll(b);
                    }
            }
        }
    }
}

// cv::resize INTER_LINEAR fixed point (the same arithmetic as the resize phase of k_pyr_fast) from level l to level l + 1
// in global memory: 4 destination pixels per lane.  Only used in front of k_pyr_fast_levels, where the launch chain - not the
// arithmetic - is the cost.
struct ResizeLevelArgs {
    const uint8_t *src; int srcPitch; long long srcFrameStride; int sh;
    uint8_t *dst; int dw, dh, dstPitch; long long dstFrameStride;
    const int *xofs, *xab, *yofs, *yab;
};
__global__ __launch_bounds__(256) void k_resize_level(ResizeLevelArgs a)
{
    const int q = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y, f = blockIdx.z;
    if (q * 4 >= a.dw) return;
    const int sy = a.yofs[dy], wy = a.yab[dy];
    const int b0 = wy & 0xFFFF, b1 = (unsigned)wy >> 16;
    const int hm1 = a.sh - 1;
    const uint8_t *r0 = a.src + (size_t)f * a.srcFrameStride + (size_t)min(max(sy, 0), hm1) * a.srcPitch;
    const uint8_t *r1 = a.src + (size_t)f * a.srcFrameStride + (size_t)min(max(sy + 1, 0), hm1) * a.srcPitch;
    uint32_t packed = 0;
    const int n = min(4, a.dw - q * 4);
    for (int j = 0; j < n; ++j) {
        const int dx = q * 4 + j;
        const int sx = a.xofs[dx], wx = a.xab[dx];
        const int wa = (short)(wx & 0xFFFF), wb = wx >> 16;
        // (where cv::resize clamps sx to w - 1 the right tap's weight is 0: it is not read)
        const int h0 = r0[sx] * wa + (wb ? r0[sx + 1] * wb : 0);
        const int h1 = r1[sx] * wa + (wb ? r1[sx + 1] * wb : 0);
        const uint32_t v = (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
        packed |= v << (8 * j);
    }
    uint8_t *d = a.dst + (size_t)f * a.dstFrameStride + (size_t)dy * a.dstPitch + q * 4;
    if (n == 4) *(uint32_t *)d = packed;
    else
        for (int j = 0; j < n; ++j) d[j] = (uint8_t)(packed >> (8 * j));
}

// The whole pyramid of a (small) batch in ONE launch, without any synchronisation between workgroups: a workgroup owns one tile of every
// level (the levels' pixels dealt over the same tile grid) and computes, level by level in LDS, the region of each level it needs for
// its own tiles of all further levels (host-computed `needed` rectangles: needed[l] = owned[l] + the sources of needed[l + 1]).  Halo
// pixels are computed by several workgroups - the arithmetic is exact integer, so every copy is the same value - and only owned pixels
// are written.  Seven dependent resize launches were 57 us of a single-frame call.
struct PyrChainArgs {
    const uint8_t *src; int srcPitch; long long srcFrameStride;
    uint8_t *dst[ORBX_MAX_LEVELS]; int dstPitch[ORBX_MAX_LEVELS]; long long dstFrameStride[ORBX_MAX_LEVELS];  // [1 .. nlevels)
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS];
    const int *xofs[ORBX_MAX_LEVELS], *xab[ORBX_MAX_LEVELS], *yofs[ORBX_MAX_LEVELS], *yab[ORBX_MAX_LEVELS];  // transition l - 1 -> l at [l]
    const int *rects;       // [nTiles][nlevels][8]: needed x0, x1, y0, y1 (exclusive ends), owned x0, x1, y0, y1
    int nTiles, nlevels, nframes, bufBytes, tabBytes;
};

__global__ __launch_bounds__(256) void k_pyr_chain(PyrChainArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t pc_lds[];
    int f, tile;
    if (!xcd_decode(a.nframes, a.nTiles, f, tile)) return;
    const int tid = threadIdx.x;
    const int *rt = a.rects + (size_t)tile * a.nlevels * 8;
    uint8_t *prev = pc_lds, *cur = pc_lds + a.bufBytes;
    int2 *tabs = (int2 *)(pc_lds + 2 * (size_t)a.bufBytes);  // per level: (xofs, xab) of the needed columns, then (yofs, yab) of the needed rows
    // all table slices and the level-0 region are requested at once: one global round trip for the whole chain
    {
        int toff = 0;
        for (int l = 1; l < a.nlevels; ++l) {
            const int *q = rt + l * 8;
            const int gx0 = q[0] & ~3, nX = (((q[1] + 3) >> 2) << 2) - gx0, nY = q[3] - q[2];
            const int *xofs = a.xofs[l], *xab = a.xab[l], *yofs = a.yofs[l], *yab = a.yab[l];
            const int wl = a.w[l];
            for (int i = tid; i < nX + nY; i += 256) {
                int2 e = make_int2(0, 0);
                if (i < nX) { const int dx = gx0 + i; if (dx < wl) e = make_int2(xofs[dx], xab[dx]); }
                else { const int dy = q[2] + (i - nX); e = make_int2(yofs[dy], yab[dy]); }
                tabs[toff + i] = e;
            }
            toff += nX + nY;
        }
    }
    int px0 = rt[0], px1 = rt[1], py0 = rt[2], py1 = rt[3];
    int pp = ((px1 - px0) + 3) & ~3;  // LDS pitch of the previous level's region
    {
        const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
        const int wr = px1 - px0, nr = py1 - py0;
        const float invW = __builtin_amdgcn_rcpf((float)max(wr, 1));
        for (int it = tid; it < wr * nr; it += 256) {
            const int r = (int)(((float)it + 0.5f) * invW), c = it - r * wr;
            prev[r * pp + c] = srcF[(size_t)(py0 + r) * a.srcPitch + px0 + c];
        }
    }
    __syncthreads();
    int toff = 0;
    for (int l = 1; l < a.nlevels; ++l) {
        const int *q = rt + l * 8;
        const int x0 = q[0], x1 = q[1], y0 = q[2], y1 = q[3], ox0 = q[4], ox1 = q[5], oy0 = q[6], oy1 = q[7];
        const int gx0 = x0 & ~3, ng = ((x1 + 3) >> 2) - (gx0 >> 2), nr = y1 - y0;
        const int cp = ng * 4;  // LDS pitch of this level's region (column gx0 at offset 0)
        const int2 *tx = tabs + toff, *ty = tx + cp;
        toff += cp + nr;
        const int hm1 = a.h[l - 1] - 1;
        uint8_t *dstF = a.dst[l] + (size_t)f * a.dstFrameStride[l];
        const float invG = __builtin_amdgcn_rcpf((float)max(ng, 1));
        for (int it = tid; it < ng * nr; it += 256) {
            const int r = (int)(((float)it + 0.5f) * invG), gq = it - r * ng;
            const int dy = y0 + r, dx0 = gx0 + 4 * gq;
            const int2 ye = ty[r];
            const int sy = ye.x, b0 = ye.y & 0xFFFF, b1 = (unsigned)ye.y >> 16;
            const uint8_t *r0 = prev + (min(max(sy, 0), hm1) - py0) * pp - px0;
            const uint8_t *r1 = prev + (min(max(sy + 1, 0), hm1) - py0) * pp - px0;
            uint32_t packed = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int dx = dx0 + j;
                if (dx >= x0 && dx < x1) {
                    const int2 xe = tx[4 * gq + j];
                    const int sx = xe.x, wa = (short)(xe.y & 0xFFFF), wb = xe.y >> 16;
                    const int h0 = r0[sx] * wa + (wb ? r0[sx + 1] * wb : 0);
                    const int h1 = r1[sx] * wa + (wb ? r1[sx + 1] * wb : 0);
                    packed |= (uint32_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) << (8 * j);
                }
            }
            *(uint32_t *)(cur + r * cp + 4 * gq) = packed;
            if (dy >= oy0 && dy < oy1) {
                uint8_t *d = dstF + (size_t)dy * a.dstPitch[l] + dx0;
                if (dx0 >= ox0 && dx0 + 4 <= ox1) *(uint32_t *)d = packed;
                else
                    for (int j = 0; j < 4; ++j)
                        if (dx0 + j >= ox0 && dx0 + j < ox1) d[j] = (uint8_t)(packed >> (8 * j));
            }
        }
        __syncthreads();
        uint8_t *t = prev; prev = cur; cur = t;
        px0 = gx0; py0 = y0; pp = cp;
    }
}

hipError_t orbxLaunchPyrChain(hipStream_t st, int nframes, int nlevels, const uint8_t *src, int srcPitch, long long srcFrameStride,
                              uint8_t *const *dst, const int *dstPitch, const long long *dstFrameStride, const int *w, const int *h,
                              const int *const *xofs, const int *const *xab, const int *const *yofs, const int *const *yab, const int *rects,
                              int nTiles, int bufBytes, int tabBytes)
{
    if (nframes <= 0 || nlevels < 2) return hipSuccess;
    if (nlevels > ORBX_MAX_LEVELS || 2 * (size_t)bufBytes + (size_t)tabBytes > 64 * 1024 || (bufBytes & 7)) return hipErrorInvalidValue;
    PyrChainArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    for (int l = 0; l < nlevels; ++l) {
        a.w[l] = w[l]; a.h[l] = h[l];
        if (l > 0) {
            if ((dstPitch[l] & 3) || (dstFrameStride[l] & 3) || ((uintptr_t)dst[l] & 3)) return hipErrorInvalidValue;
            a.dst[l] = dst[l]; a.dstPitch[l] = dstPitch[l]; a.dstFrameStride[l] = dstFrameStride[l];
            a.xofs[l] = xofs[l]; a.xab[l] = xab[l]; a.yofs[l] = yofs[l]; a.yab[l] = yab[l];
        }
    }
    a.rects = rects; a.nTiles = nTiles; a.nlevels = nlevels; a.nframes = nframes; a.bufBytes = bufBytes; a.tabBytes = tabBytes;
    hipLaunchKernelGGL(k_pyr_chain, dim3(xcd_grid(nframes, nTiles)), dim3(256), 2 * (size_t)bufBytes + (size_t)tabBytes, st, a);
    return hipGetLastError();
}

hipError_t orbxLaunchResizeLevel(hipStream_t st, int nframes, const uint8_t *src, int srcPitch, long long srcFrameStride, int sh, uint8_t *dst,
                                 int dw, int dh, int dstPitch, long long dstFrameStride, OrbxResizeTabs tabs)
{
    if (nframes <= 0) return hipSuccess;
    if ((dstPitch & 3) || (dstFrameStride & 3) || dh > 65535 || nframes > 65535) return hipErrorInvalidValue;
    ResizeLevelArgs a;
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride; a.sh = sh;
    a.dst = dst; a.dw = dw; a.dh = dh; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.xofs = tabs.xofs; a.xab = tabs.xab; a.yofs = tabs.yofs; a.yab = tabs.yab;
    hipLaunchKernelGGL(k_resize_level, dim3(((dw + 3) / 4 + 255) / 256, dh, nframes), dim3(256), 0, st, a);
    return hipGetLastError();
}

static void fillPyrFastArgs(PyrFastArgs &a, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch, long long srcFrameStride,
                            uint8_t *dst, int dw, int dh, int dstPitch, long long dstFrameStride, OrbxResizeTabs tabs, uint32_t *cand,
                            long long candFrameStride, int *tileCount, int tilesPerFrame, int iniTh, int minTh)
{
    memset(&a, 0, sizeof a);
    a.g = g;
    a.tiles = tabs.pfTiles;
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    {
        const uintptr_t bits = (uintptr_t)src | (uintptr_t)srcPitch | (uintptr_t)srcFrameStride;
        a.srcAlign = (bits & 15) == 0 ? 16 : ((bits & 3) == 0 ? 4 : 1);
    }
    a.dst = dst; a.dw = dw; a.dh = dh; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.tabs = tabs;
    a.cand = cand; a.candFrameStride = candFrameStride;
    a.tileCount = tileCount; a.tilesPerFrame = tilesPerFrame;
    a.iniTh = iniTh; a.minTh = minTh;
    a.nframes = nframes;
    a.phases = 15;
    static const int laneAlloc = [] { const char *e = getenv("ORBX_PF_LANE_ALLOC"); return e ? atoi(e) : 1; }();  // (0: A/B and tests)
    a.laneAlloc = laneAlloc;
    const XcdMagic xm = xcd_magic(g.nTiles);
    a.xcdM = xm.m; a.xcdL = xm.l; a.xcdPer = xm.per;
}

size_t orbxPyrFastArgsBytes() { return sizeof(PyrFastArgs); }

// argument block of level `g` for k_pyr_fast_levels (no destination level), written to host memory `out` (orbxPyrFastArgsBytes() bytes)
void orbxFillPyrFastLevelArgs(void *out, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch, long long srcFrameStride,
                              const OrbxPfTile *tiles, uint32_t *cand, long long candFrameStride, int *tileCount, int tilesPerFrame, int iniTh,
                              int minTh)
{
    OrbxResizeTabs tabs;
    memset(&tabs, 0, sizeof tabs);
    tabs.pfTiles = tiles;
    fillPyrFastArgs(*(PyrFastArgs *)out, nframes, g, src, srcPitch, srcFrameStride, nullptr, 0, 0, 0, 0, tabs, cand, candFrameStride, tileCount,
                    tilesPerFrame, iniTh, minTh);
}

hipError_t orbxLaunchPyrFastLevels(hipStream_t st, int nframes, const OrbxLevelGeom *geomsHost, int nlevels, const void *dLevelArgs)
{
    if (nlevels <= 0 || nlevels > ORBX_MAX_LEVELS) return hipErrorInvalidValue;
    PyrFastLevelsArgs a;
    memset(&a, 0, sizeof a);
    a.levels = (const PyrFastArgs *)dLevelArgs;
    a.nlevels = nlevels; a.nframes = nframes;
    int lds = 0;
    for (int l = 0; l < nlevels; ++l) {
        if (geomsHost[l].groupCells > PF_MAX_GROUP_CELLS) return hipErrorInvalidValue;
        a.tileStart[l + 1] = a.tileStart[l] + geomsHost[l].nTiles;
        lds = std::max(lds, geomsHost[l].ldsBytes);
    }
    hipLaunchKernelGGL((k_pyr_fast<false, true>), dim3(xcd_grid(nframes, a.tileStart[nlevels])), dim3(PF_NT), lds, st, a);
    return hipGetLastError();
}

// packed host frames [nf][h][w] (as they crossed the link in one linear copy) -> the pitched device layout k_pyr_fast reads
__global__ __launch_bounds__(256) void k_unpack_frames(const uint8_t *src, uint8_t *dst, int w, int h, long long pitch, long long frameStride)
{
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    const int c4 = (blockIdx.x * 256 + threadIdx.x) * 4, r = blockIdx.y, f = blockIdx.z;
    if (c4 >= w) return;
    const uint8_t *s = src + ((size_t)f * h + r) * w + c4;
    uint8_t *d = dst + (size_t)f * frameStride + (size_t)r * pitch + c4;
    if (c4 + 4 <= w) *(uint32_t *)d = *(const u32_unaligned *)s;
    else
        for (int k = 0; c4 + k < w; ++k) d[k] = s[k];
}

hipError_t orbxLaunchUnpackFrames(hipStream_t st, const uint8_t *packed, uint8_t *dst, int w, int h, long long pitch, long long frameStride, int nf)
{
    if (nf <= 0) return hipSuccess;
    if ((pitch & 3) || (frameStride & 3) || ((uintptr_t)dst & 3) || h > 65535 || nf > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_unpack_frames, dim3((w + 1023) / 1024, h, nf), dim3(256), 0, st, packed, dst, w, h, pitch, frameStride);
    return hipGetLastError();
}

hipError_t orbxLaunchPyrFast(hipStream_t st, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch,
                             long long srcFrameStride, uint8_t *dst, int dw, int dh, int dstPitch,
                             long long dstFrameStride, OrbxResizeTabs tabs, uint32_t *cand, long long candFrameStride,
                             int *tileCount, int tilesPerFrame, int iniTh, int minTh)
{
    PyrFastArgs a;
    if (!tabs.pfTiles) return hipErrorInvalidValue;
    fillPyrFastArgs(a, nframes, g, src, srcPitch, srcFrameStride, dst, dw, dh, dstPitch, dstFrameStride, tabs, cand, candFrameStride, tileCount,
                    tilesPerFrame, iniTh, minTh);
    {
        static int phases = -1;
        if (phases < 0) { const char *e = getenv("ORBX_PF_PHASES"); phases = e ? atoi(e) : 15; }
        a.phases = phases;
    }
    if (g.groupCells > PF_MAX_GROUP_CELLS) return hipErrorInvalidValue;
    if (getenv("ORBX_DEBUG")) {
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_pyr_fast<false>, PF_NT, g.ldsBytes);
        fprintf(stderr, "[orbx] k_pyr_fast level %dx%d: tiles/frame %d (%d groups x %d rows, %d cells/group), LDS %d B, %d threads, %d blocks/CU\n", g.w, g.h,
                g.nTiles, g.nGroups, g.ncy, g.groupCells, g.ldsBytes, PF_NT, nb);
    }
    dim3 grid(xcd_grid(nframes, g.nTiles));
    static const int ldsPad = [] { const char *e = getenv("ORBX_PF_LDS_PAD"); return e ? atoi(e) : 0; }();  // occupancy experiments
    // ORBX_PF_KERNEL: 1 (default) the phase-organised kernel; 2 a wave per cell (k_pyr_fast_cw: three barriers instead of six to ten, no
    // atomics, no bitmaps - and 16 % MORE vector instructions, because every sparse stage runs per cell at a fraction of a wave: 2.34
    // against 1.80 ms per 1024 VGA frames.  The kernel sits at the practical issue rate of its instruction mix either way
    // (tools/experiments/ring_rate.hip), so the instruction count decides; kept as the measured alternative)
    static const int pfKernel = [] { const char *e = getenv("ORBX_PF_KERNEL"); return e ? atoi(e) : 1; }();
    if (pfKernel == 2) {
        if (a.phases == 15) hipLaunchKernelGGL(k_pyr_fast_cw<false>, grid, dim3(PF_NT), g.cwLdsBytes + ldsPad, st, a);
        else hipLaunchKernelGGL(k_pyr_fast_cw<true>, grid, dim3(PF_NT), g.cwLdsBytes + ldsPad, st, a);
        return hipGetLastError();
    }
    if (a.phases == 15 && a.laneAlloc) hipLaunchKernelGGL(k_pyr_fast<false>, grid, dim3(PF_NT), g.ldsBytes + ldsPad, st, a);
    else hipLaunchKernelGGL(k_pyr_fast<true>, grid, dim3(PF_NT), g.ldsBytes + ldsPad, st, a);
    return hipGetLastError();
}

// =====================================================================================================
// GaussianBlur 7x7, sigma 2, BORDER_REFLECT_101, 8.8 fixed point
// =====================================================================================================
#define BL_TW 64
#define BL_TH 32

__device__ __forceinline__ int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

// The blurred levels are only ever read as 37 x 37 patches around keypoints (k_orient_desc), so they are stored TILED: 16 x 8 pixel
// tiles of 128 contiguous bytes (one L2 line), tiles in raster order.  A patch then touches ~18 lines instead of ~51 row
// segments of a row-major image - that kernel is bound by L1 line fills - and a wave of the blur writes whole lines.
__host__ __device__ __forceinline__ size_t orbx_blur_tiled_offset(int x, int y, int tilesX)
{
    return ((size_t)(y >> 3) * tilesX + (x >> 4)) * 128 + (size_t)((y & 7) * 16 + (x & 15));
}

struct BlurArgs {
    const uint8_t *src; int srcPitch; long long srcFrameStride;
    uint8_t *dst; int dstPitch; long long dstFrameStride;  // dst is tiled: dstPitch = tiles per tile row
    int w, h;
    int taps[7];
    int nframes, tilesX, tilesY;
    int srcAligned;  // k_blur3: source rows may be staged with aligned 16-byte loads
    uint32_t biasH, biasL, biasHi;  // k_blur3: constant-term operand bytes (bl3_bias_bytes)
    // k_blur3: workgroup -> (frame, unit) and unit -> (segment, column strip) without the two integer divisions a wave of ~600
    // instructions otherwise starts with (xcd_magic(tilesX * tilesY), div_magic(tilesX))
    unsigned xcdM; int xcdL; unsigned xcdPer; unsigned txM; int txL;
};

__global__ __launch_bounds__(256) void k_blur(BlurArgs a)
{
    __shared__ uint8_t in[BL_TH + 6][BL_TW + 8];
    __shared__ uint16_t hs[BL_TH + 6][BL_TW];
    int f, tileIdx;
    if (!xcd_decode(a.nframes, a.tilesX * a.tilesY, f, tileIdx)) return;
    const int x0 = (tileIdx % a.tilesX) * BL_TW, y0 = (tileIdx / a.tilesX) * BL_TH;
    const int tid = threadIdx.x;
    const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
    for (int i = tid; i < (BL_TH + 6) * (BL_TW + 6); i += 256) {
        int r = i / (BL_TW + 6), c = i - r * (BL_TW + 6);
        int gy = reflect101(y0 + r - 3, a.h), gx = reflect101(x0 + c - 3, a.w);
        in[r][c] = srcF[(size_t)gy * a.srcPitch + gx];
    }
    __syncthreads();
    for (int i = tid; i < (BL_TH + 6) * BL_TW; i += 256) {
        int r = i >> 6, c = i & 63;
        unsigned acc = 0;
#pragma unroll
        for (int k = 0; k < 7; ++k) acc += (unsigned)a.taps[k] * in[r][c + k];
        hs[r][c] = (uint16_t)min(acc, 65535u);
    }
    __syncthreads();
    uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
    for (int i = tid; i < BL_TH * BL_TW; i += 256) {
        int r = i >> 6, c = i & 63;
        int gx = x0 + c, gy = y0 + r;
        if (gx < a.w && gy < a.h) {
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < 7; ++k) acc += (unsigned)a.taps[k] * hs[r + k][c];
            unsigned v = (acc + 32768u) >> 16;
            dstF[orbx_blur_tiled_offset(gx, gy, a.dstPitch)] = (uint8_t)min(v, 255u);
        }
    }
}

// ---- v2: SWAR blur.  Tile = 128 px x 64 rows per workgroup.
// Horizontal pass: one lane = 4 adjacent output pixels; the 10 source bytes come from three aligned dwords of
// the global row; byte pairs (s[j], s[j+2]) are isolated with one v_alignbyte + one AND and multiplied as packed
// 2 x u16 lanes (sums stay below 2^16, so one 32-bit multiply-add serves two pixels); symmetric taps are folded.
// Vertical pass: one lane = the same 4-pixel column for 8 output rows, 14 rows of 16-bit sums in registers.
#define BL2_GW 32   // 4-px groups per tile row (128 px)
#define BL2_TH 64   // output rows per tile
#define BL2_RT 8    // output rows per lane in the vertical pass

__device__ __forceinline__ uint32_t bl_alignbyte(uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }

__global__ __launch_bounds__(256) void k_blur2(BlurArgs a)
{
    __shared__ uint2 T[BL2_TH + 6][BL2_GW];
    int f, tileIdx;
    if (!xcd_decode(a.nframes, a.tilesX * a.tilesY, f, tileIdx)) return;
    const int x0 = (tileIdx % a.tilesX) * (BL2_GW * 4), y0 = (tileIdx / a.tilesX) * BL2_TH;
    const int tid = threadIdx.x;
    const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
    const uint32_t k0 = a.taps[0], k1 = a.taps[1], k2 = a.taps[2], k3 = a.taps[3];
    const int rowsNeeded = min(BL2_TH, a.h - y0) + 6;

    // horizontal pass: one item = 8 adjacent output pixels (two 4-px groups sharing four aligned source dwords)
    auto group4 = [&](const uint8_t *row, int x, uint32_t &oe, uint32_t &oo) {  // generic 4-px group (image borders)
        if (x >= 4 && x + 7 < a.w) {
            const uint32_t *p = (const uint32_t *)(row + x);
            const uint32_t w0 = p[-1], w1 = p[0], w2 = p[1];
            const uint32_t M = 0x00FF00FFu;
            const uint32_t em3 = bl_alignbyte(w1, w0, 1) & M, em2 = bl_alignbyte(w1, w0, 2) & M, em1 = bl_alignbyte(w1, w0, 3) & M;
            const uint32_t e0 = w1 & M, e1 = bl_alignbyte(w2, w1, 1) & M, e2 = bl_alignbyte(w2, w1, 2) & M;
            const uint32_t e3 = bl_alignbyte(w2, w1, 3) & M, e4 = w2 & M;
            oe = k0 * (em3 + e3) + k1 * (em2 + e2) + k2 * (em1 + e1) + k3 * e0;
            oo = k0 * (em2 + e4) + k1 * (em1 + e3) + k2 * (e0 + e2) + k3 * e1;
        } else {
            uint32_t sv[10];  // REFLECT_101 per byte
#pragma unroll
            for (int j = 0; j < 10; ++j) sv[j] = row[reflect101(x - 3 + j, a.w)];
            uint32_t t4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                t4[i] = k0 * (sv[i] + sv[i + 6]) + k1 * (sv[i + 1] + sv[i + 5]) + k2 * (sv[i + 2] + sv[i + 4]) + k3 * sv[i + 3];
            oe = t4[0] | (t4[2] << 16);
            oo = t4[1] | (t4[3] << 16);
        }
    };
    for (int it = tid; it < rowsNeeded * (BL2_GW / 2); it += 256) {
        const int r = it >> 4, gp = it & 15;
        const int x = x0 + 8 * gp;
        uint32_t oeA = 0, ooA = 0, oeB = 0, ooB = 0;
        if (x < a.w) {
            const int gy = reflect101(y0 + r - 3, a.h);
            const uint8_t *row = srcF + (size_t)gy * a.srcPitch;
            if (x >= 4 && x + 11 < a.w) {
                const uint32_t *p = (const uint32_t *)(row + x);
                const uint32_t w0 = p[-1], w1 = p[0], w2 = p[1], w3 = p[2];
                const uint32_t M = 0x00FF00FFu;
                const uint32_t em3 = bl_alignbyte(w1, w0, 1) & M, em2 = bl_alignbyte(w1, w0, 2) & M, em1 = bl_alignbyte(w1, w0, 3) & M;
                const uint32_t e0 = w1 & M, e1 = bl_alignbyte(w2, w1, 1) & M, e2 = bl_alignbyte(w2, w1, 2) & M, e3 = bl_alignbyte(w2, w1, 3) & M;
                const uint32_t e4 = w2 & M, e5 = bl_alignbyte(w3, w2, 1) & M, e6 = bl_alignbyte(w3, w2, 2) & M, e7 = bl_alignbyte(w3, w2, 3) & M;
                const uint32_t e8 = w3 & M;
                oeA = k0 * (em3 + e3) + k1 * (em2 + e2) + k2 * (em1 + e1) + k3 * e0;
                ooA = k0 * (em2 + e4) + k1 * (em1 + e3) + k2 * (e0 + e2) + k3 * e1;
                oeB = k0 * (e1 + e7) + k1 * (e2 + e6) + k2 * (e3 + e5) + k3 * e4;
                ooB = k0 * (e2 + e8) + k1 * (e3 + e7) + k2 * (e4 + e6) + k3 * e5;
            } else {
                group4(row, x, oeA, ooA);
                if (x + 4 < a.w) group4(row, x + 4, oeB, ooB);
            }
        }
        T[r][2 * gp] = make_uint2(oeA, ooA);
        T[r][2 * gp + 1] = make_uint2(oeB, ooB);
    }
    __syncthreads();

    const int g = tid & 31, rb = tid >> 5;
    const int x = x0 + 4 * g;
    const int r0 = rb * BL2_RT;
    if (x >= a.w || y0 + r0 >= a.h) return;
    uint32_t t[BL2_RT + 6][4];
#pragma unroll
    for (int j = 0; j < BL2_RT + 6; ++j) {
        const uint2 v = T[min(r0 + j, rowsNeeded - 1)][g];
        t[j][0] = v.x & 0xFFFFu; t[j][2] = v.x >> 16; t[j][1] = v.y & 0xFFFFu; t[j][3] = v.y >> 16;
    }
    uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
#pragma unroll
    for (int i = 0; i < BL2_RT; ++i) {
        const int gy = y0 + r0 + i;
        if (gy < a.h) {
            uint32_t packed = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t acc = k0 * (t[i][c] + t[i + 6][c]) + k1 * (t[i + 1][c] + t[i + 5][c]) + k2 * (t[i + 2][c] + t[i + 4][c]) + k3 * t[i + 3][c];
                packed |= min((acc + 32768u) >> 16, 255u) << (8 * c);
            }
            *(uint32_t *)(dstF + orbx_blur_tiled_offset(x, gy, a.dstPitch)) = packed;  // (a tile row holds all 4 pixels)
        }
    }
}

// ---- v3: the separable 7-tap blur on the matrix cores (exact integer arithmetic, v_mfma_i32_16x16x32_i8).
// A 7-tap convolution of a 16-wide block is a banded 32 x 16 Toeplitz product, so one wave walks down a 16-column strip in
// chunks of 16 rows:
//   horizontal  H[16 rows][16 cols] = X[16 rows][32 cols] * Th          A = pixels - 128 (8 bytes of a row per lane, straight
//                                                                       from global memory), B = taps; accumulator preset to
//                                                                       128 * sum(taps), so H is the exact 16-bit row sum
//   vertical    V^T[16 cols][16 rows] = H^T[16 cols][32 rows] * Tv      the result layout of the first product (4 consecutive
//                                                                       rows of one column per lane) IS the A layout of the
//                                                                       second (summation over rows) - no lane movement, no LDS;
//                                                                       H is split into a low and a high byte plane (two
//                                                                       products, V = 256 * Vh + Vl); K = this chunk + the next
// and each lane ends up with 4 horizontally adjacent output pixels of one row: one dword store.  BORDER_REFLECT_101 is applied
// when the A operand is loaded (interior strips take aligned 8-byte loads, the strips at the left / right image border gather
// reflected bytes).  About 25 VALU + 3 MFMA instructions per 256 pixels, against ~150 VALU in the SWAR version.
#define BL3_SEG 4  // 16-row output blocks per wave (plus one extra chunk of rows) in the all-levels launch of small batches
typedef int bl3_v4i __attribute__((ext_vector_type(4)));

// The constant terms (128 * sum(taps) from the signed-byte operands, + 32768 rounding) ride in K slots whose taps are zero
// anyway - the last 8 window columns of the horizontal product, the last 4 "next chunk" rows of the vertical one, both held by
// lanes 48..63 - as sum(a_i * b_i) with b = {127, 127, 127, 1}; so the accumulators start from the inline constant 0 and no
// bias is added on the VALU.  Returns the a bytes for the constant X <= 4 * 127 * 127.
static uint32_t bl3_bias_bytes(int X)
{
    const int q = X / 127, r = X % 127;
    const int a0 = std::min(q, 127), a1 = std::min(std::max(q - 127, 0), 127), a2 = std::max(q - 254, 0);
    return (uint32_t)a0 | ((uint32_t)a1 << 8) | ((uint32_t)a2 << 16) | ((uint32_t)r << 24);
}
#define BL3_BIAS_B 0x017F7F7Fu

__device__ __forceinline__ int bl3_reflect(int p, int len)  // REFLECT_101 for -len < p < 2 * len - 1, clamped beyond (unused taps)
{
    int q = p < 0 ? -p : p;
    q = q >= len ? 2 * (len - 1) - q : q;
    return max(q, 0);
}

// Per-lane B operands of the two products (the tap matrices), computed once on the host and read from the kernarg segment: a lane's
// operands depend on its position in the wave and on the taps only, and building them from shifted tap words cost every wave ~75
// instructions.  K slot (g, jj) of the horizontal product is window column 8g + jj (image column x0 - 4 + 8g + jj), tap
// 8g + jj - n - 1 for output column n; of the vertical product it is row 4g + jj of the previous chunk (jj < 4), tap 4g + jj - n for
// output row n, or row 4g + jj - 4 of this one, tap 16 + 4g + jj - 4 - n.  Lanes 48..63 hold the constant terms (above).
struct BlurLaneTab { uint32_t v[64][4]; };  // {Bh low, Bh high, Bv low, Bv high}
typedef const __attribute__((address_space(4))) BlurLaneTab BlurLaneTabK;

static void fillBlurLaneTab(BlurLaneTab &t, const int taps[7])
{
    auto tapWin = [&](int k0) -> uint32_t {  // bytes tap[k0 .. k0 + 3], zero outside 0 .. 6
        uint32_t r = 0;
        for (int j = 0; j < 4; ++j)
            if (k0 + j >= 0 && k0 + j < 7) r |= (uint32_t)taps[k0 + j] << (8 * j);
        return r;
    };
    for (int lane = 0; lane < 64; ++lane) {
        const int n = lane & 15, g = lane >> 4;
        const bool biasLane = g == 3;
        t.v[lane][0] = biasLane ? BL3_BIAS_B : tapWin(8 * g - n - 1);
        t.v[lane][1] = biasLane ? 0u : tapWin(8 * g + 4 - n - 1);
        t.v[lane][2] = tapWin(4 * g - n);
        t.v[lane][3] = biasLane ? BL3_BIAS_B : tapWin(16 + 4 * g - n);
    }
}

// NW waves per workgroup = NW adjacent 16-column strips of one row segment of SEG 16-row blocks.  The source rows of the segment are
// staged in LDS with coalesced 16-byte loads (each 128-byte line is fetched once per workgroup; per-wave 8-byte gathers straight from
// global memory fetched every line ~5 times through a thrashing L1) and the A operands are read from there.
// The kernel is bound by instruction issue, not by the matrix cores or HBM (the products are 3 of ~40 instructions per block), so what
// a wave does once - decode, staging, operands - is kept short and spread over as many blocks as the LDS tile allows.
template <int NW, int SEG>
__device__ __forceinline__ void blur3_body(const BlurArgs &a, BlurLaneTabK *tab, const int f, const int unit)
{
    constexpr int TW = 16 * NW, NVEC = NW + 2, LP = 16 * (NW + 3), NT = 64 * NW, ROWS = 16 * (SEG + 1);
    constexpr int RPP = NT / NVEC, NIT = (ROWS + RPP - 1) / RPP;  // rows per staging pass (NVEC threads per row), passes
    __shared__ __attribute__((aligned(16))) uint8_t tile[ROWS * LP];  // image columns [tx0, tx0 + TW + 32), REFLECT_101 rows
    const int seg = (int)div_by_magic((unsigned)unit, a.txM, a.txL), sgx = unit - seg * a.tilesX;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tx0 = sgx * TW - 16;
    const int x0 = sgx * TW + wv * 16;
    const int n = lane & 15, g = lane >> 4;
    const int nBlk = (a.h + 15) >> 4;
    const int b0 = seg * SEG, b1 = min(b0 + SEG, nBlk);
    const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
    uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
    const int nRows = 16 * (b1 - b0 + 1);
    if (a.srcAligned) {
        // A thread keeps its 16-byte column and walks down the rows, RPP at a time.  All its loads are issued before the first one is
        // waited for (one memory latency per workgroup, not one per pass): unconditional loads from clamped, always valid addresses,
        // only the LDS writes are predicated.
        const int r0 = tid / NVEC, v = tid - r0 * NVEC;
        const int col = tx0 + 16 * v;
        const bool colOk = tid < RPP * NVEC && col >= 0 && col + 16 <= a.srcPitch;
        // (the frame's wave-uniform base + a 32-bit lane offset is one addressing mode of the load: a frame is < 2^31 bytes and rows x pitch
        // < 2^24; as a per-lane 64-bit pointer every load cost two v_mad_u64_u32 and a v_lshl_add_u64 - in a wave of ~600 instructions)
        const uint32_t colOff = (uint32_t)min(max(col, 0), a.srcPitch - 16);
        uint4 stage[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            stage[k] = *(const uint4 *)(srcF + (colOff + __umul24((uint32_t)bl3_reflect(16 * b0 - 3 + min(r0 + RPP * k, nRows - 1), a.h), (uint32_t)a.srcPitch)));
#pragma unroll
        for (int k = 0; k < NIT; ++k)  // (keeps the compiler from sinking each load into the predicated block of its LDS write)
            asm volatile("" : "+v"(stage[k].x), "+v"(stage[k].y), "+v"(stage[k].z), "+v"(stage[k].w));
        uint8_t *my = tile + r0 * LP + 16 * v;
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (colOk && r0 + RPP * k < nRows) *(uint4 *)(my + RPP * k * LP) = stage[k];
    } else {
        for (int it = tid; it < nRows * (TW + 32); it += NT) {
            const int r = it / (TW + 32), cx = it - r * (TW + 32);
            const int col = tx0 + cx;
            if (col >= 0 && col < a.w) tile[r * LP + cx] = srcF[(size_t)bl3_reflect(16 * b0 - 3 + r, a.h) * a.srcPitch + col];
        }
    }
    __syncthreads();
    if (x0 >= a.w) return;

    const __attribute__((address_space(4))) uint32_t *tv = &tab->v[lane][0];
    const long Bh = (long)(((unsigned long)tv[1] << 32) | tv[0]), Bv = (long)(((unsigned long)tv[3] << 32) | tv[2]);
    const bool biasLane = g == 3;
    const bool fast = x0 >= 4 && x0 + 28 <= a.w;  // the whole window lies inside the image (wave-uniform)
    // LDS byte offsets of this lane's 8 window columns (strips at the image border: REFLECT_101, clamped where the tap is 0)
    int colOff[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!fast) {  // only the strips at the left / right image border pay for the reflections
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) colOff[jj] = min(max(bl3_reflect(x0 - 4 + 8 * g + jj, a.w) - tx0, 0), TW + 31);
    }
    const uint8_t *myRow = tile + n * LP + (x0 - 4 + 8 * g - tx0);
    auto loadA = [&](int i) -> uint2 {  // rows 16 (b0 + i) - 3 + (lane & 15) of the extended image, pixels - 128
        uint2 v;
        if (fast) {
            const uint32_t *q = (const uint32_t *)(myRow + 16 * i * LP);
            v = make_uint2(q[0], q[1]);
        } else {
            const uint8_t *row = tile + (16 * i + n) * LP;
            uint32_t t[2] = {0, 0};
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) t[jj >> 2] |= (uint32_t)row[colOff[jj]] << (8 * (jj & 3));
            v = make_uint2(t[0], t[1]);
        }
        return biasLane ? make_uint2(a.biasH, 0) : make_uint2(v.x ^ 0x80808080u, v.y ^ 0x80808080u);
    };
    const bl3_v4i zero = {0, 0, 0, 0};
    const int xo = x0 + 4 * g;
    // output row n of block c lies 2 c tile rows below row n of block 0: one address per lane, a scalar step per block
    // (the frame's base + a 32-bit lane offset: the blurred frame is < 2^31 bytes)
    const uint32_t blockBytes = (uint32_t)a.dstPitch * 256u;
    const uint32_t dstLaneOff = ((uint32_t)(n >> 3) * (uint32_t)a.dstPitch + (uint32_t)(xo >> 4)) * 128u + (uint32_t)((n & 7) * 16 + (xo & 15)) + (uint32_t)b0 * blockBytes;
    const bool xOk = xo < a.w;
    // Straight-line code over the chunks (no early exit: blocks beyond the image compute on stale LDS rows and fail the `y < a.h` test
    // of the store), so that the LDS reads of all chunks are in flight before the first product and the products of a chunk overlap
    // the VALU work of the previous one.
    uint2 avAll[SEG + 1];
#pragma unroll
    for (int i = 0; i <= SEG; ++i) avAll[i] = loadA(i);
    uint32_t loPrev = 0, hiPrev = 0;
#pragma unroll
    for (int i = 0; i <= SEG; ++i) {
        const uint2 av = avAll[i];
        const long A = (long)(((unsigned long)av.y << 32) | av.x);
        const bl3_v4i H = __builtin_amdgcn_mfma_i32_16x16x32_i8(A, Bh, zero, 0, 0, 0);  // exact row sums, 0 .. 255 * S <= 65535
        const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)H[1], (uint32_t)H[0], 0x05010400u);
        const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)H[3], (uint32_t)H[2], 0x05010400u);
        const uint32_t lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;  // low bytes of rows 4g .. 4g + 3, - 128
        const uint32_t hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u) ^ 0x80808080u;  // high bytes, - 128
        if (i > 0) {
            const long Alo = (long)(((unsigned long)(biasLane ? a.biasL : lo) << 32) | loPrev);
            const long Ahi = (long)(((unsigned long)(biasLane ? a.biasHi : hi) << 32) | hiPrev);
            const bl3_v4i Dl = __builtin_amdgcn_mfma_i32_16x16x32_i8(Alo, Bv, zero, 0, 0, 0);
            const bl3_v4i Dh = __builtin_amdgcn_mfma_i32_16x16x32_i8(Ahi, Bv, zero, 0, 0, 0);  // includes the rounding term >> 8
            const int y = 16 * (b0 + i - 1) + n;
            if (y < a.h && xOk) {
                // X = Dh * 256 + Dl, pixel = min(X >> 16, 255) (the 4.5.0 taps sum to 257: 255 needs the clamp).  The high halves of two
                // X's as an i16 pair (v_perm_b32), both clamped and narrowed by ONE v_sat_pk_u8_i16: 9 instructions per 4 pixels where
                // shift / min / shift-or per pixel were 15 - in a wave of ~590.
                uint32_t x[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] = (uint32_t)((Dh[r] << 8) + Dl[r]);
                uint32_t q01, q23;
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(q01) : "v"(__builtin_amdgcn_perm(x[1], x[0], 0x07060302u)));
                asm("v_sat_pk_u8_i16 %0, %1" : "=v"(q23) : "v"(__builtin_amdgcn_perm(x[3], x[2], 0x07060302u)));
                const uint32_t packed = q01 | (q23 << 16);
                *(uint32_t *)(dstF + (dstLaneOff + (uint32_t)(i - 1) * blockBytes)) = packed;  // (a tile row holds all 4 pixels)
            }
        }
        loPrev = lo; hiPrev = hi;
    }
}

struct Blur3KArgs { BlurArgs a; BlurLaneTab t; };

template <int NW, int SEG>
__global__ __launch_bounds__(NW * 64) void k_blur3(Blur3KArgs ka)
{
    int f, unit;
    if (!xcd_decode_magic(ka.a.nframes, ka.a.xcdM, ka.a.xcdL, ka.a.xcdPer, f, unit)) return;
    const __attribute__((address_space(4))) char *kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    blur3_body<NW, SEG>(ka.a, (BlurLaneTabK *)(kp + offsetof(Blur3KArgs, t)), f, unit);
}

// every level of a (small) batch in one launch: eight launches of a few microseconds each were the longest branch of a single-frame call
struct BlurLevelsArgs {
    BlurArgs lv[ORBX_MAX_LEVELS];
    int tileStart[ORBX_MAX_LEVELS + 1];
    int nlevels, nframes;
    BlurLaneTab t;
};
__global__ __launch_bounds__(256) void k_blur3_levels(BlurLevelsArgs la)
{
    int f, unit;
    if (!xcd_decode(la.nframes, la.tileStart[la.nlevels], f, unit)) return;
    int level = 0;
    for (int l = 1; l < la.nlevels; ++l) level += unit >= la.tileStart[l] ? 1 : 0;
    const BlurArgs a = la.lv[level];
    const __attribute__((address_space(4))) char *kp = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
    blur3_body<4, BL3_SEG>(a, (BlurLaneTabK *)(kp + offsetof(BlurLevelsArgs, t)), f, unit - la.tileStart[level]);
}

static bool blur3Eligible(const int taps[7], int w, int h, const uint8_t *dst, long long dstFrameStride)
{
    static const int variant = [] { const char *e = getenv("ORBX_BLUR_KERNEL"); return e ? atoi(e) : 3; }();
    static const int nw = [] { const char *e = getenv("ORBX_BLUR_WAVES"); return e ? atoi(e) : 4; }();
    bool i8ok = true;
    for (int k = 0; k < 7; ++k) i8ok = i8ok && taps[k] >= 0 && taps[k] <= 127;
    const bool small = (taps[0] + taps[1] + taps[2]) * 2 + taps[3] <= 257;
    return variant == 3 && nw == 4 && small && i8ok && (((uintptr_t)dst | (uintptr_t)dstFrameStride) & 3) == 0 && w >= 32 && h >= 16;
}

static void fillBlur3Args(BlurArgs &a, int nframes, int w, int h, const uint8_t *src, int srcPitch, long long srcFrameStride, uint8_t *dst,
                          int dstPitch, long long dstFrameStride, const int taps[7])
{
    memset(&a, 0, sizeof a);
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    a.dst = dst; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.w = w; a.h = h;
    for (int k = 0; k < 7; ++k) a.taps[k] = taps[k];
    a.srcAligned = (((uintptr_t)src | (uintptr_t)srcPitch | (uintptr_t)srcFrameStride) & 15) == 0;
    const int S = taps[0] + taps[1] + taps[2] + taps[3] + taps[4] + taps[5] + taps[6];
    a.biasH = bl3_bias_bytes(128 * S);
    a.biasL = bl3_bias_bytes(128 * S);
    a.biasHi = bl3_bias_bytes(128 * S + 128);
    a.nframes = nframes; a.tilesY = ((h + 15) / 16 + BL3_SEG - 1) / BL3_SEG;
    a.tilesX = (w + 63) / 64;
    const XcdMagic xm = xcd_magic(a.tilesX * a.tilesY), tm = div_magic((unsigned)a.tilesX);
    a.xcdM = xm.m; a.xcdL = xm.l; a.xcdPer = xm.per; a.txM = tm.m; a.txL = tm.l;
}

// GaussianBlur of `nlevels` levels in one launch (k_blur3, 4 waves); hipErrorNotSupported when a level needs another kernel
hipError_t orbxLaunchBlurLevels(hipStream_t st, int nframes, int nlevels, const int *w, const int *h, const uint8_t *const *src, const int *srcPitch,
                                const long long *srcFrameStride, uint8_t *const *dst, const int *dstPitch, const long long *dstFrameStride,
                                const int taps[7])
{
    if (nlevels <= 0 || nlevels > ORBX_MAX_LEVELS) return hipErrorInvalidValue;
    BlurLevelsArgs la;
    memset(&la, 0, sizeof la);
    for (int l = 0; l < nlevels; ++l) {
        if (!blur3Eligible(taps, w[l], h[l], dst[l], dstFrameStride[l])) return hipErrorNotSupported;
        fillBlur3Args(la.lv[l], nframes, w[l], h[l], src[l], srcPitch[l], srcFrameStride[l], dst[l], dstPitch[l], dstFrameStride[l], taps);
        la.tileStart[l + 1] = la.tileStart[l] + la.lv[l].tilesX * la.lv[l].tilesY;
    }
    la.nlevels = nlevels; la.nframes = nframes;
    fillBlurLaneTab(la.t, taps);
    hipLaunchKernelGGL(k_blur3_levels, dim3(xcd_grid(nframes, la.tileStart[nlevels])), dim3(256), 0, st, la);
    return hipGetLastError();
}

hipError_t orbxLaunchBlur(hipStream_t st, int nframes, int w, int h, const uint8_t *src, int srcPitch,
                          long long srcFrameStride, uint8_t *dst, int dstPitch, long long dstFrameStride,
                          const int taps[7])
{
    BlurArgs a;
    memset(&a, 0, sizeof a);
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    a.dst = dst; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.w = w; a.h = h;
    for (int k = 0; k < 7; ++k) a.taps[k] = taps[k];
    const bool symmetric = taps[0] == taps[6] && taps[1] == taps[5] && taps[2] == taps[4];
    const bool small = (taps[0] + taps[1] + taps[2]) * 2 + taps[3] <= 257;  // packed 16-bit horizontal sums cannot overflow
    const uintptr_t bits = (uintptr_t)src | (uintptr_t)srcPitch | (uintptr_t)srcFrameStride | (uintptr_t)dst | (uintptr_t)dstFrameStride;
    static const int variant = [] { const char *e = getenv("ORBX_BLUR_KERNEL"); return e ? atoi(e) : 3; }();  // 2: SWAR, 1: generic (A/B timing)
    bool i8ok = true;
    for (int k = 0; k < 7; ++k) i8ok = i8ok && taps[k] >= 0 && taps[k] <= 127;
    const uintptr_t dbits = (uintptr_t)dst | (uintptr_t)dstFrameStride;
    a.srcAligned = (((uintptr_t)src | (uintptr_t)srcPitch | (uintptr_t)srcFrameStride) & 15) == 0;
    const int S = taps[0] + taps[1] + taps[2] + taps[3] + taps[4] + taps[5] + taps[6];
    if (variant == 3 && small && i8ok && (dbits & 3) == 0 && w >= 32 && h >= 16) {
        a.biasH = bl3_bias_bytes(128 * S);         // pixels - 128
        a.biasL = bl3_bias_bytes(128 * S);         // low byte plane - 128
        a.biasHi = bl3_bias_bytes(128 * S + 128);  // high byte plane - 128, + 32768 >> 8 for the final rounding
        static const int nw = [] { const char *e = getenv("ORBX_BLUR_WAVES"); return e ? atoi(e) : 4; }();
        // 16-row blocks per workgroup.  8 spreads what a wave does once over twice the blocks and is 11 % faster alone (0.61 against
        // 0.66 ms per 1024-frame batch), but its 16 KB tiles take the LDS from the quadtree next to it: within 1 % either way on the
        // four workloads (tools/experiments/blur_seg_sweep.sh), so the small tile stays.
        static const int segEnv = [] { const char *e = getenv("ORBX_BLUR_SEG"); return e ? atoi(e) : 0; }();
        const int seg = segEnv == 8 ? 8 : 4;
        Blur3KArgs ka;
        ka.a = a;
        fillBlurLaneTab(ka.t, taps);
        ka.a.nframes = nframes; ka.a.tilesY = ((h + 15) / 16 + seg - 1) / seg;
        ka.a.tilesX = nw == 8 ? (w + 127) / 128 : (w + 63) / 64;
        {
            const XcdMagic xm = xcd_magic(ka.a.tilesX * ka.a.tilesY), tm = div_magic((unsigned)ka.a.tilesX);
            ka.a.xcdM = xm.m; ka.a.xcdL = xm.l; ka.a.xcdPer = xm.per; ka.a.txM = tm.m; ka.a.txL = tm.l;
        }
        const dim3 grid(xcd_grid(nframes, ka.a.tilesX * ka.a.tilesY));
        if (nw == 8 && seg == 8) hipLaunchKernelGGL((k_blur3<8, 8>), grid, dim3(512), 0, st, ka);
        else if (nw == 8) hipLaunchKernelGGL((k_blur3<8, 4>), grid, dim3(512), 0, st, ka);
        else if (seg == 8) hipLaunchKernelGGL((k_blur3<4, 8>), grid, dim3(256), 0, st, ka);
        else hipLaunchKernelGGL((k_blur3<4, 4>), grid, dim3(256), 0, st, ka);
    } else if (variant >= 2 && symmetric && small && (bits & 3) == 0) {
        a.nframes = nframes; a.tilesX = (w + BL2_GW * 4 - 1) / (BL2_GW * 4); a.tilesY = (h + BL2_TH - 1) / BL2_TH;
        hipLaunchKernelGGL(k_blur2, dim3(xcd_grid(nframes, a.tilesX * a.tilesY)), dim3(256), 0, st, a);
    } else {
        a.nframes = nframes; a.tilesX = (w + BL_TW - 1) / BL_TW; a.tilesY = (h + BL_TH - 1) / BL_TH;
        hipLaunchKernelGGL(k_blur, dim3(xcd_grid(nframes, a.tilesX * a.tilesY)), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

// =====================================================================================================
// Quadtree keypoint distribution — ORBextractor::DistributeOctTree (reference src/ORBextractor.cc:584-774,
// DivideNode :515-567, compareNodes :569-582; step specification SURVEY App. D).
// One wave per (frame, level).  Control flow is wave-uniform (every lane evaluates the same scalars);
// the 64 lanes do the per-node stable 4-way key partitions with ballot + popcount, the ordered gather of
// the FAST candidates and the per-node arg-max.  List order (push_front / erase) is kept exactly with a
// doubly linked list in LDS; the std::sort tie order with the libstdc++ introsort emulation.
// =====================================================================================================
#define QT_LDS_KEYS 0      // candidates of one (frame, level) held in an LDS buffer behind the node arrays; 0: always in global scratch

typedef unsigned long long qt_u64;

struct QtArgs {
    const OrbxLevelGeom *geoms;  // device array [nlevels]
    const uint32_t *cand; long long candFrameStride;
    const int *tileCount; int tilesPerFrame;
    uint32_t *scratchA, *scratchB;  // [nframes][candFrameStride] spill space for levels with > QT_LDS_KEYS candidates
    uint32_t *sel; int selFrameStride;
    int *selCount;
    int nlevels, nframes;
    int levelBase, levelsRun;  // this launch covers levels [levelBase, levelBase + levelsRun)
    int pool;     // node pool size P
    int listCap;  // capacity of list-sized arrays (max quota + 8)
    int ldsKeys;  // candidates of one instance held in LDS (0: always in global scratch)
    int maxKeys;  // largest candidate set this launch of the level-synchronous kernel takes (ORBX_QT_MAX_KEYS / _KEYS2 lower it in tests)
    int pass2;    // 1: second launch of the level-synchronous kernel - only the instances the first one deferred, larger run-start bitmap
    int nodeBytes;  // bytes of the node arrays of one instance (in LDS the keys follow)
    uint8_t *nodeScratch;  // != nullptr: node arrays in global memory, nodeBytes per instance of this launch (quotas beyond LDS)
    int mwKeys;   // k_quadtree_mw: the candidate count its LDS layout (run-start bitmap, per-chunk arrays) is sized for
    int variant;  // 3: level-synchronous (default), 1: serial list walk (ORBX_QT_KERNEL; also taken for > QT3_MAX_KEYS candidates)
    int stop;     // debug (ORBX_QT_STOP): 1 return after the gather, 2 after the roots, 3 skip the sorted phase, 4 skip the sort only
};

// the kernels read the argument block in place from the kernarg segment (no private copy when it is passed on by reference)
typedef const __attribute__((address_space(4))) QtArgs QtArgsK;

// (count, UL.x, node id) of an expandable node packed as count << 32 | ulx << 16 | id; compareNodes (:569-582)
// orders by (count, UL.x) only = key >> 16
struct QtSizedLess {
    __host__ __device__ __forceinline__ bool operator()(qt_u64 a, qt_u64 b) const { return (a >> 16) < (b >> 16); }
};

// the workgroup is a single wave: ordering only has to hold inside the wave
template <bool LDSK>
__device__ __forceinline__ void qt_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (LDSK) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    else __builtin_amdgcn_s_waitcnt(0x0070);       // vmcnt(0) lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

#define QT_F_NOMORE 1
#define QT_F_ALIVE 2

struct QtCtx {
    int *gx, *gy, *beg, *cnt;           // node data [P]: UL.x | UR.x << 16, UL.y | BR.y << 16, key range
    unsigned short *fl;                 // [P] QT_F_*
    short *front, *freeL;               // [P] children pushed to the list front in the sorted phase; free node ids
    short *order, *tmpC, *tmpK;         // [L] list order; children / kept nodes of the running pass
    int *sortStack;                     // [216] explicit recursion stack of the std::sort emulation
    qt_u64 *sized, *prev;               // [L] expandable nodes of the running / previous pass
    int size, freeTop, bump, nSized, nFront, nC, nK, P;
};

__device__ __forceinline__ int qt_alloc(QtCtx &S)
{
    if (S.freeTop > 0) return S.freeL[--S.freeTop];
    return S.bump++;
}

// ExtractorNode::DivideNode (:515-567) + "add childs if they contain points" (:653-688 / :717-741) + erase.
// mode 0 = list walk of an outer pass (children collected in tmpC, the parent's id is recycled at once);
// mode 1 = sorted phase (children pushed on the front stack, the parent becomes a tombstone).
template <bool LDSK, class KP>
__device__ __forceinline__ void qt_divide(QtCtx &S, KP A, KP B, int node, int mode)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = lanemask_lt();
    const int gx = S.gx[node], gy = S.gy[node];
    const int ulx = gx & 0xFFFF, urx = gx >> 16, uly = gy & 0xFFFF, bry = gy >> 16;
    const int begin = S.beg[node], cnt = S.cnt[node];
    const int mx = ulx + ((urx - ulx + 1) >> 1);  // UL.x + ceil((UR.x - UL.x) / 2)
    const int my = uly + ((bry - uly + 1) >> 1);
    int c[4] = {0, 0, 0, 0};
    if (cnt <= WAVE) {
        const bool act = lane < cnt;
        const uint32_t key = act ? A[begin + lane] : 0u;
        const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
        const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
        unsigned long long m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = __ballot(act && q == k); c[k] = __popcll(m[k]); }
        const int o1 = c[0], o2 = o1 + c[1], o3 = o2 + c[2];
        const int offq = q == 0 ? 0 : (q == 1 ? o1 : (q == 2 ? o2 : o3));
        const unsigned long long mq = q == 0 ? m[0] : (q == 1 ? m[1] : (q == 2 ? m[2] : m[3]));
        qt_sync<LDSK>();
        if (act) A[begin + offq + __popcll(mq & lt)] = key;
    } else {
        for (int base = 0; base < cnt; base += WAVE) {
            const bool act = base + lane < cnt;
            const uint32_t key = act ? A[begin + base + lane] : 0u;
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] += __popcll(__ballot(act && q == k));
        }
        int run[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
        for (int base = 0; base < cnt; base += WAVE) {
            const bool act = base + lane < cnt;
            const uint32_t key = act ? A[begin + base + lane] : 0u;
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
            unsigned long long m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = __ballot(act && q == k);
            const int rq = q == 0 ? run[0] : (q == 1 ? run[1] : (q == 2 ? run[2] : run[3]));
            const unsigned long long mq = q == 0 ? m[0] : (q == 1 ? m[1] : (q == 2 ? m[2] : m[3]));
            if (act) B[begin + rq + __popcll(mq & lt)] = key;
#pragma unroll
            for (int k = 0; k < 4; ++k) run[k] += __popcll(m[k]);
        }
        qt_sync<false>();  // B is global scratch: wait for the stores (vmcnt) before other lanes read them back
        for (int i = lane; i < cnt; i += WAVE) A[begin + i] = B[begin + i];
    }
    // the parent's id is free for its own children in an outer pass (nothing refers to it any more)
    if (mode == 0) S.freeL[S.freeTop++] = (short)node;
    else S.fl[node] = 0;
    // children n1 (UL), n2 (UR), n3 (BL), n4 (BR) in push order
    int off = begin, nonEmpty = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c[k] > 0) {
            const int id = qt_alloc(S);
            const int cul = (k & 1) ? mx : ulx, cur = (k & 1) ? urx : mx;
            const int cuy = (k & 2) ? my : uly, cby = (k & 2) ? bry : my;
            S.gx[id] = cul | (cur << 16);
            S.gy[id] = cuy | (cby << 16);
            S.beg[id] = off;
            S.cnt[id] = c[k];
            S.fl[id] = (unsigned short)(QT_F_ALIVE | (c[k] == 1 ? QT_F_NOMORE : 0));
            if (mode == 0) S.tmpC[S.nC++] = (short)id;
            else S.front[S.nFront++] = (short)id;
            if (c[k] > 1) S.sized[S.nSized++] = ((qt_u64)(unsigned)c[k] << 32) | ((qt_u64)(unsigned)cul << 16) | (unsigned)id;
            off += c[k];
            ++nonEmpty;
        }
    }
    S.size += nonEmpty - 1;
    qt_sync<LDSK>();
}

// final list order = front stack reversed (alive entries) followed by the base order (alive entries) -> dst[0 .. size)
template <bool LDSK>
__device__ __forceinline__ int qt_flatten(QtCtx &S, int nOrder, short *dst)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = lanemask_lt();
    int out = 0;
    for (int base = 0; base < S.nFront; base += WAVE) {
        const int j = S.nFront - 1 - (base + lane);
        const int id = j >= 0 ? S.front[j] : 0;
        const bool alive = j >= 0 && (S.fl[id] & QT_F_ALIVE);
        const unsigned long long m = __ballot(alive);
        if (alive) dst[out + __popcll(m & lt)] = (short)id;
        out += __popcll(m);
    }
    for (int base = 0; base < nOrder; base += WAVE) {
        const int j = base + lane;
        const int id = j < nOrder ? S.order[j] : 0;
        const bool alive = j < nOrder && (S.fl[id] & QT_F_ALIVE);
        const unsigned long long m = __ballot(alive);
        if (alive) dst[out + __popcll(m & lt)] = (short)id;
        out += __popcll(m);
    }
    qt_sync<LDSK>();
    return out;
}

template <bool LDSK, bool NG, class KP>
__device__ __forceinline__ void qt_body(QtArgsK &a, const OrbxLevelGeom &g, int f, int level, uint8_t *qlds, KP A, KP B, int M,
                        const int *tc, uint32_t *selOut, int *selCnt)
{
    const int lane = threadIdx.x;
    const int P = a.pool, L = a.listCap;
    QtCtx S;
    S.gx = (int *)qlds; S.gy = S.gx + P; S.beg = S.gy + P; S.cnt = S.beg + P;
    S.sized = (qt_u64 *)(S.cnt + P); S.prev = S.sized + L;
    S.fl = (unsigned short *)(S.prev + L);
    S.front = (short *)(S.fl + P); S.freeL = S.front + P;
    S.order = S.freeL + P; S.tmpC = S.order + L; S.tmpK = S.tmpC + L;
    S.sortStack = (int *)(S.tmpK + L);
    S.P = P;

    // ---- ordered gather of the FAST candidates (tile slots in cell-row-major order) --------------------
    // flattened over the M candidates: lane e finds its tile by binary search in the prefix of the tile counts (LDS),
    // so all slot loads are independent and stay in flight together
    {
        int *tileOff = (int *)S.prev;  // nTiles + 1 prefix offsets; `prev` is free until the sorted phase
        int base = 0;
        for (int tb = 0; tb < g.nTiles; tb += WAVE) {
            const int t = tb + lane;
            const int myN = t < g.nTiles ? tc[t] : 0;
            int total;
            const int myOff = wave_excl_scan(myN, total);
            if (t < g.nTiles) tileOff[t] = base + myOff;
            base += total;
        }
        if (lane == 0) tileOff[g.nTiles] = base;
        qt_sync<!NG>();
        const uint32_t *slots = a.cand + (size_t)f * a.candFrameStride + g.candBase;
        for (int e = lane; e < M; e += WAVE) {
            int lo = 0, hi = g.nTiles;  // largest t with tileOff[t] <= e
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (tileOff[mid] <= e) lo = mid; else hi = mid;
            }
            A[e] = slots[(size_t)lo * g.slotCap + (e - tileOff[lo])];
        }
    }
    qt_sync<LDSK>();
    if (a.stop == 1) { if (lane == 0) *selCnt = 0; return; }

    // ---- roots (:589-626) -----------------------------------------------------------------------------
    const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
    const int N = g.quota;
    const int nIni = (int)roundf(__fdiv_rn((float)W, (float)H));
    const float hX = __fdiv_rn((float)W, (float)nIni);
    S.size = 0; S.freeTop = 0; S.bump = 0; S.nSized = 0; S.nFront = 0; S.nC = 0; S.nK = 0;
    int nOrder = 0;
    {
        int off = 0;
        for (int r = 0; r < nIni; ++r) {
            int cnt;
            if (nIni == 1) cnt = M;
            else {
                cnt = 0;  // stable compaction of the keys whose root is r, appended to B
                for (int base = 0; base < M; base += WAVE) {
                    const bool act = base + lane < M;
                    const uint32_t key = act ? A[base + lane] : 0u;
                    const bool mine = act && (int)__fdiv_rn((float)(key & 0xFFF), hX) == r;
                    const unsigned long long m = __ballot(mine);
                    if (mine) B[off + cnt + __popcll(m & lanemask_lt())] = key;
                    cnt += __popcll(m);
                }
            }
            if (cnt > 0) {
                const int id = S.bump++;
                S.gx[id] = (int)__fmul_rn(hX, (float)r) | ((int)__fmul_rn(hX, (float)(r + 1)) << 16);
                S.gy[id] = 0 | (H << 16);
                S.beg[id] = off; S.cnt[id] = cnt;
                S.fl[id] = (unsigned short)(QT_F_ALIVE | (cnt == 1 ? QT_F_NOMORE : 0));
                S.order[nOrder++] = (short)id;  // push_back
                ++S.size;
            }
            off += cnt;
        }
        if (nIni > 1) {
            qt_sync<false>();
            for (int i = lane; i < M; i += WAVE) A[i] = B[i];
        }
    }
    qt_sync<LDSK>();

    if (a.stop == 2) { if (lane == 0) *selCnt = 0; return; }
    // ---- subdivision (:630-752) -----------------------------------------------------------------------
    bool finish = false;
    while (!finish) {
        // one walk over the list: every node with more than one key is divided; its children go to the list front
        // in creation order, so the new list is reverse(children) followed by the untouched nodes
        const int prevSize = S.size;
        S.nSized = 0; S.nC = 0; S.nK = 0;
        // 64 list entries per step: the nodes that are final (one key) are appended to tmpK in bulk, the others are
        // divided one after the other in list order (their ids broadcast from the lanes' registers)
        for (int base = 0; base < nOrder; base += WAVE) {
            const int i = base + lane;
            const int node = i < nOrder ? S.order[i] : 0;
            const bool valid = i < nOrder;
            const bool noMore = valid && (S.fl[node] & QT_F_NOMORE);
            const unsigned long long mKeep = __ballot(noMore);
            unsigned long long mDiv = __ballot(valid && !noMore);
            if (noMore) S.tmpK[S.nK + __popcll(mKeep & lanemask_lt())] = (short)node;
            S.nK += __popcll(mKeep);
            while (mDiv) {
                const int b = __ffsll((long long)mDiv) - 1;
                mDiv &= mDiv - 1;
                qt_divide<LDSK>(S, A, B, __builtin_amdgcn_readlane(node, b), 0);
            }
        }
        qt_sync<LDSK>();
        for (int j = lane; j < S.nC; j += WAVE) S.order[j] = S.tmpC[S.nC - 1 - j];
        for (int j = lane; j < S.nK; j += WAVE) S.order[S.nC + j] = S.tmpK[j];
        nOrder = S.nC + S.nK;
        qt_sync<LDSK>();
        const int nToExpand = S.nSized;
        if (S.size >= N || S.size == prevSize) {
            finish = true;
        } else if (S.size + nToExpand * 3 > N) {
            // sorted phase: divide the largest nodes first until the quota is reached (:701-751)
            if (a.stop == 3) break;
            while (!finish) {
                const int prevSize2 = S.size;
                const int nPrev = S.nSized;
                S.nSized = 0;
                {
                    // (a variant with the elements in lane registers + v_readlane measured 4x slower than this LDS sort)
                    for (int i = lane; i < nPrev; i += WAVE) S.prev[i] = S.sized[i];
                    qt_sync<LDSK>();
                    if (lane == 0 && a.stop != 4) orbx_sort::sort(S.prev, nPrev, QtSizedLess(), S.sortStack);
                }
                qt_sync<LDSK>();
                for (int j = nPrev - 1; j >= 0; --j) {
                    // node ids are only recycled by a flatten: make room for 4 children first if the pool is nearly used up
                    if (S.P - S.bump + S.freeTop < 4) {
                        nOrder = qt_flatten<LDSK>(S, nOrder, S.tmpC);
                        for (int i = lane; i < nOrder; i += WAVE) S.order[i] = S.tmpC[i];
                        S.nFront = 0;
                        S.freeTop = 0;
                        for (int base = 0; base < S.bump; base += WAVE) {
                            const int id = base + lane;
                            const bool dead = id < S.bump && !(S.fl[id] & QT_F_ALIVE);
                            const unsigned long long m = __ballot(dead);
                            if (dead) S.freeL[S.freeTop + __popcll(m & lanemask_lt())] = (short)id;
                            S.freeTop += __popcll(m);
                        }
                        qt_sync<LDSK>();
                    }
                    qt_divide<LDSK>(S, A, B, (int)(S.prev[j] & 0xFFFF), 1);
                    if (S.size >= N) break;
                }
                if (S.size >= N || S.size == prevSize2) finish = true;
            }
        }
    }

    // ---- retain the best (first strictly greatest response) key of every node, in list order (:755-771) --
    const int nFinal = qt_flatten<LDSK>(S, nOrder, S.tmpC);
    for (int k = lane; k < nFinal; k += WAVE) {
        const int node = S.tmpC[k];
        const int b = S.beg[node], n = S.cnt[node];
        uint32_t best = A[b];
        for (int i = 1; i < n; ++i) {
            const uint32_t key = A[b + i];
            if ((key >> 24) > (best >> 24)) best = key;
        }
        selOut[k] = best;
    }
    if (lane == 0) *selCnt = nFinal;
}

// ---- v3: level-synchronous formulation.  The list is an ARRAY of nodes in list order and the keys are kept in list order
// too (a node's keys = one contiguous run, in candidate order), so one pass of the reference's list walk - or one round of its
// sorted phase - becomes a handful of wave-parallel sweeps instead of one serial DivideNode per node:
//   1. key sweep:   every key finds its node (popcount over a bitmap of run starts) and its quadrant; per-node child counts
//                   with ballots restricted to the node's lane range;
//   2. node sweep:  nodes to divide, in processing order t (list order, or the std::sort order from the largest count down);
//                   prefix sums over t give the list size before each divide (-> the reference's `break` at size >= N), the
//                   creation index of every child and its slot in vSizeAndPointerToNode.  push_front makes the new list
//                   reverse(children in creation order) ++ untouched nodes in their old order;
//   3. key sweep:   stable scatter of every key to its node's new run (children by quadrant, untouched nodes verbatim).
// Only the std::sort emulation (the tie order of libstdc++'s introsort is observable) still runs on one lane.
// A single wave runs these sweeps as chains of LDS round trips, so they keep several independent reads in flight: lists of
// up to 256 nodes are handled as 4 entries per lane (loads first, then the scans), up to KCH * 64 keys stay in registers for
// a whole pass (the scatter is then in place, in one LDS buffer), and counters advance with fire-and-forget LDS atomics.
#define QT3_MAX_KEYS 8192   // first launch: run-start bitmap of 1 KB in LDS.  Larger candidate sets (levels of 1080p images) are deferred
#define QT3_MAX_KEYS2 49152 // to a second launch of the same kernel with a 6 KB bitmap (key offsets are 16-bit, 0xFFFF is a sentinel);
                            // beyond that - and for everything when ORBX_QT_KERNEL=1 - the serial formulation takes over
// 64-key chunks held in registers (template parameter of the kernel).  32 chunks: 158 VGPRs = 3 waves per SIMD, 12 instances per CU;
// 8 chunks: 105 VGPRs = 4 per SIMD, 16 per CU where the level pair's LDS allows it (quotas up to ~170).  Candidate sets beyond the
// chunks are re-read from (L2-resident) scratch in every sweep, so the small variant is for launches whose largest quota is <= 300
// (c2: quadtree 0.91 -> 0.79 ms per batch; at 1500 features per frame, quota 326, it is 3 % slower).
#define QT3_KCH_LARGE 32
#define QT3_KCH_SMALL 8

struct Qt3 {
    qt_u64 *V;                 // [L] expandable nodes (count << 32 | UL.x << 16 | list position): vSizeAndPointerToNode
    uint32_t *gx[2], *gy[2];   // [L] node bounds by list position: UL.x | UR.x << 16, UL.y | BR.y << 16 (two generations)
    uint32_t *cc;              // [L][2] four 16-bit slots: child counts of the pass, then list positions, then write cursors
    uint32_t *aux;             // [L] creation index of the node's first child | index of its first entry in V << 16; ~0: untouched
    uint16_t *cnt[2], *beg[2]; // [L], [L + 2] keys of the node, exclusive prefix
    uint16_t *proc;            // [L] list positions of the nodes to divide, in processing order
    int *sortStack;            // [216]
    uint32_t *starts;          // [maxKeys / 32 + 2] bitmap: key e is the first key of its node
};
#define QT3_NONE 0xFFFFu

__host__ __device__ static inline size_t qt3NodeBytes(int L, int maxKeys = QT3_MAX_KEYS2) { return (((size_t)L * 46 + 8 + 216 * 4 + (size_t)(maxKeys / 32 + 2) * 4) + 15) & ~(size_t)15; }

// A / B: key buffers.  With M <= QT3_KCH * 64 they are the same (LDS) buffer, larger sets ping-pong between two global ones.
typedef __attribute__((address_space(1))) uint32_t *QtGlobalKeys;  // the key buffers A / B in global scratch, typed: global_load / global_store
// key `i` of a buffer: for the typed global buffers a wave-uniform base + an UNSIGNED 32-bit byte offset (one addressing mode of the
// load / store; with a signed index the compiler forms a per-lane 64-bit address: a shift, a sign extension and a 64-bit add in
// front of every access of a kernel that is nothing but dependent latencies)
__device__ __forceinline__ __attribute__((address_space(1))) uint32_t &qt_key(QtGlobalKeys base, int i)
{
    typedef __attribute__((address_space(1))) char *gbytes_t;
    return *(QtGlobalKeys)((gbytes_t)base + ((uint32_t)i << 2));
}
__device__ __forceinline__ uint32_t &qt_key(uint32_t *base, int i) { return base[i]; }
template <bool NG, int QT3_KCH, class KP = uint32_t *>  // NG: the node arrays live in global scratch (quotas too large for LDS), not in LDS;
                                 // QT3_KCH: 64-key chunks held in registers
__device__ __forceinline__ void qt3_body(QtArgsK &a, const OrbxLevelGeom &g, int f, uint8_t *qlds, KP A, KP B, int M,
                         const int *tc, uint32_t *selOut, int *selCnt)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = lanemask_lt();
    const int L = a.listCap;
    const bool inRegs = M <= QT3_KCH * WAVE;
    Qt3 S;
    S.V = (qt_u64 *)qlds;
    S.gx[0] = (uint32_t *)(S.V + L); S.gx[1] = S.gx[0] + L; S.gy[0] = S.gx[1] + L; S.gy[1] = S.gy[0] + L;
    S.cc = S.gy[1] + L; S.aux = S.cc + 2 * L;
    S.cnt[0] = (uint16_t *)(S.aux + L); S.cnt[1] = S.cnt[0] + L; S.beg[0] = S.cnt[1] + L; S.beg[1] = S.beg[0] + L + 2;
    S.proc = S.beg[1] + L + 2;
    S.sortStack = (int *)(S.proc + L);
    S.starts = (uint32_t *)(S.sortStack + 216);

    uint32_t kk[QT3_KCH];  // keys base + u * 64 + lane of the running batch of chunks
    // ---- ordered gather of the FAST candidates (tile slots in cell-row-major order) --------------------
    {
        int *tileOff = (int *)qlds;  // nTiles + 1 prefix offsets; the node arrays are not in use yet
        int base = 0;
        for (int tb = 0; tb < g.nTiles; tb += WAVE) {
            const int t = tb + lane;
            const int myN = t < g.nTiles ? tc[t] : 0;
            int total;
            const int myOff = wave_excl_scan(myN, total);
            if (t < g.nTiles) tileOff[t] = base + myOff;
            base += total;
        }
        if (lane == 0) tileOff[g.nTiles] = base;
        qt_sync<!NG>();
        const uint32_t *slots = a.cand + (size_t)f * a.candFrameStride + g.candBase;
        for (int e0 = 0; e0 < M; e0 += QT3_KCH * WAVE) {  // all slot loads of a batch are in flight together
#pragma unroll
            for (int u = 0; u < QT3_KCH; ++u) {
                const int e = e0 + u * WAVE + lane;
                if (e0 + u * WAVE < M) {
                    int lo = 0, hi = g.nTiles;  // largest t with tileOff[t] <= e
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (tileOff[mid] <= e) lo = mid; else hi = mid;
                    }
                    kk[u] = e < M ? slots[(size_t)lo * g.slotCap + (e - tileOff[lo])] : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < QT3_KCH; ++u) {
                const int e = e0 + u * WAVE + lane;
                if (e < M) qt_key(A, e) = kk[u];
            }
        }
    }
    qt_sync<false>();
    if (a.stop == 1) { if (lane == 0) *selCnt = 0; return; }

    // ---- roots (:589-626): push_back in order, keys stably grouped by root ---------------------------------------------
    const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
    const int N = g.quota;
    const int nIni = (int)roundf(__fdiv_rn((float)W, (float)H));
    const float hX = __fdiv_rn((float)W, (float)nIni);
    int cur = 0, size = 0;
    bool kkValid = inRegs;  // kk holds A[0 .. M)
    {
        KP R = inRegs ? (KP)(a.scratchB + (size_t)f * a.candFrameStride + g.candBase) : B;  // grouping target
        int off = 0;
        for (int r = 0; r < nIni; ++r) {
            int cnt;
            if (nIni == 1) cnt = M;
            else {
                cnt = 0;
                for (int base = 0; base < M; base += WAVE) {
                    const bool act = base + lane < M;
                    const uint32_t key = act ? qt_key(A, base + lane) : 0u;
                    const bool mine = act && (int)__fdiv_rn((float)(key & 0xFFF), hX) == r;
                    const unsigned long long m = __ballot(mine);
                    if (mine) qt_key(R, off + cnt + __popcll(m & lt)) = key;
                    cnt += __popcll(m);
                }
            }
            if (cnt > 0 && lane == 0) {
                S.gx[0][size] = (uint32_t)(int)__fmul_rn(hX, (float)r) | ((uint32_t)(int)__fmul_rn(hX, (float)(r + 1)) << 16);
                S.gy[0][size] = 0u | ((uint32_t)H << 16);
                S.cnt[0][size] = (uint16_t)cnt;
            }
            if (cnt > 0) ++size;
            off += cnt;
        }
        if (nIni > 1) {
            qt_sync<false>();
            if (inRegs) { for (int i = lane; i < M; i += WAVE) qt_key(A, i) = qt_key(R, i); kkValid = false; }
            else { KP t = A; A = B; B = t; }
        }
    }
    qt_sync<false>();
    if (a.stop == 2) { if (lane == 0) *selCnt = 0; return; }

    int nV = 0;  // entries of V after a pass
    // one pass / one sorted round.  nSorted < 0: divide every node with more than one key, in list order; otherwise divide
    // V[nSorted - 1], V[nSorted - 2], ... while the list holds fewer than N nodes.
    auto pass = [&](int nSorted) {
        // (the two generations of an array lie behind each other: generation c = base + c * length.  Indexing the pointer PAIRS with `cur`
        // made them generic pointers - the compiler no longer saw that they point into LDS -, and every access to them a flat load / store
        // that counts against both wait counters)
        const int nxt = cur ^ 1;
        uint32_t *gx = S.gx[0] + cur * L, *gy = S.gy[0] + cur * L;
        uint16_t *cn = S.cnt[0] + cur * L, *bg = S.beg[0] + cur * (L + 2);
        uint32_t *gx2 = S.gx[0] + nxt * L, *gy2 = S.gy[0] + nxt * L;
        uint16_t *cn2 = S.cnt[0] + nxt * L, *bg2 = S.beg[0] + nxt * (L + 2);
        // (a) run starts: exclusive prefix of the counts, bitmap of first keys; child counters to zero
        for (int i = lane; i < ((M + 63) >> 6) * 2 + 2; i += WAVE) S.starts[i] = 0;
        {
            int base = 0;
            for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
                int c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int p = pb0 + u * WAVE + lane; c[u] = p < size ? (int)cn[p] : 0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb0 + u * WAVE + lane;
                    int total;
                    const int o = base + wave_excl_scan(c[u], total);
                    if (p < size) {
                        bg[p] = (uint16_t)o;
                        atomicOr(&S.starts[o >> 5], 1u << (o & 31));
                        *(uint2 *)(S.cc + 2 * p) = make_uint2(0, 0);
                        S.aux[p] = ~0u;
                    }
                    base += total;
                }
            }
            if (lane == 0) bg[size] = (uint16_t)M;
        }
        qt_sync<!NG>();
        // key -> (node, quadrant, lane range of the node inside its 64-key chunk)
        struct KeyInfo { uint32_t key; int p, q, lo, e0; bool act, big; unsigned long long R; };
        auto classify = [&](int base, int p, uint32_t key, uint32_t ngx, uint32_t ngy, int nb, int nc) -> KeyInfo {
            KeyInfo k;
            const int e = base + lane;
            k.act = e < M; k.p = p; k.key = key;
            const int ulx = ngx & 0xFFFF, urx = ngx >> 16, uly = ngy & 0xFFFF, bry = ngy >> 16;
            const int mx = ulx + ((urx - ulx + 1) >> 1), my = uly + ((bry - uly + 1) >> 1);  // UL + ceil(extent / 2)
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            k.q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
            k.big = k.act && nc > 1;
            k.e0 = e - nb;
            k.lo = max(nb - base, 0);
            const int hi = min(nb + nc - base, 64);
            k.R = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << k.lo) - 1ull);
            return k;
        };
        // a sweep over all keys: batches of QT3_KCH chunks (keys -> registers), inside a batch 4 chunks per step (bitmap words,
        // then owners, then node fields), then `body(KeyInfo)` chunk by chunk in key order
        auto sweep = [&](bool reload, auto &&body) {
            int before = 0;
            for (int b0 = 0; b0 < M; b0 += QT3_KCH * WAVE) {
                if (reload) {
#pragma unroll
                    for (int u = 0; u < QT3_KCH; ++u) {
                        const int e = b0 + u * WAVE + lane;
                        if (b0 + u * WAVE < M) kk[u] = e < M ? qt_key(A, e) : 0u;
                    }
                }
#pragma unroll
                for (int u0 = 0; u0 < QT3_KCH; u0 += 4) {
                    if (b0 + u0 * WAVE < M) {  // (no `break`: the loop must unroll completely, kk is indexed statically)
                    unsigned long long w[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int base = b0 + (u0 + u) * WAVE;
                        const int wi = base >> 5;
                        w[u] = base < M ? ((unsigned long long)S.starts[wi] | ((unsigned long long)S.starts[wi + 1] << 32)) : 0ull;
                    }
                    int pp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int base = b0 + (u0 + u) * WAVE;
                        pp[u] = base + lane < M ? before + __popcll(w[u] & (lt | (1ull << lane))) - 1 : 0;
                        before += __popcll(w[u]);
                    }
                    uint32_t ngx[4], ngy[4]; int nb[4], nc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { ngx[u] = gx[pp[u]]; ngy[u] = gy[pp[u]]; nb[u] = (int)bg[pp[u]]; nc[u] = (int)cn[pp[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int base = b0 + (u0 + u) * WAVE;
                        if (base < M) body(classify(base, pp[u], kk[u0 + u], ngx[u], ngy[u], nb[u], nc[u]));
                    }
                    }
                }
            }
        };
        auto addSlots = [&](const KeyInfo &k, const unsigned long long (&m)[4]) {  // first lane of the node's run in the chunk
            const uint32_t lo2 = (uint32_t)__popcll(m[0] & k.R) | ((uint32_t)__popcll(m[1] & k.R) << 16);
            const uint32_t hi2 = (uint32_t)__popcll(m[2] & k.R) | ((uint32_t)__popcll(m[3] & k.R) << 16);
            if (lo2) atomicAdd(&S.cc[2 * k.p], lo2);
            if (hi2) atomicAdd(&S.cc[2 * k.p + 1], hi2);
        };
        // (b) child counts
        sweep(!kkValid, [&](const KeyInfo &k) {
            unsigned long long m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = __ballot(k.big && k.q == j);
            if (k.big && lane == k.lo) addSlots(k, m);
        });
        // (c) nodes to divide, in processing order
        int nProc = 0;
        if (nSorted < 0) {
            for (int pb = 0; pb < size; pb += WAVE) {
                const int p = pb + lane;
                const bool big = p < size && cn[p] > 1;
                const unsigned long long m = __ballot(big);
                if (big) S.proc[nProc + __popcll(m & lt)] = (uint16_t)p;
                nProc += __popcll(m);
            }
        } else {
            for (int t = lane; t < nSorted; t += WAVE) S.proc[t] = (uint16_t)(S.V[nSorted - 1 - t] & 0xFFFF);
            nProc = nSorted;
        }
        qt_sync<!NG>();
        // (d) prefix sums over the processing order: list size before each divide (a sorted round stops dividing once the
        //     list holds N nodes), creation index of the first child, index of the first new V entry
        int C = 0;
        nV = 0;
        {
            int sz = size;
            bool stop = false;
            for (int tb = 0; tb < nProc && !stop; tb += WAVE) {
                const int t = tb + lane;
                const bool val = t < nProc;
                const int p = val ? S.proc[t] : 0;
                const uint2 c2 = *(const uint2 *)(S.cc + 2 * p);
                const uint32_t c0 = c2.x & 0xFFFF, c1 = c2.x >> 16, c2_ = c2.y & 0xFFFF, c3 = c2.y >> 16;
                const int ne = val ? (c0 > 0) + (c1 > 0) + (c2_ > 0) + (c3 > 0) : 0;
                const int nx = val ? (c0 > 1) + (c1 > 1) + (c2_ > 1) + (c3 > 1) : 0;
                int totNe, totNx, totG;
                const int cb = wave_excl_scan(ne, totNe);
                const int vb = wave_excl_scan(nx, totNx);
                const int gEx = wave_excl_scan(val ? ne - 1 : 0, totG);
                bool take = val;
                if (nSorted >= 0) {
                    // the reference divides node t and THEN breaks if size >= N: node t is divided iff the size before it is < N
                    take = val && sz + gEx < N;
                    const int nTake = __popcll(__ballot(take));  // a prefix of the chunk: sizes never decrease
                    if (nTake < min(WAVE, nProc - tb)) stop = true;
                    const int last = max(nTake - 1, 0);
                    const int iNe = __shfl(cb + ne, last), iNx = __shfl(vb + nx, last), iG = __shfl(gEx + ne - 1, last);
                    totNe = nTake > 0 ? iNe : 0; totNx = nTake > 0 ? iNx : 0; totG = nTake > 0 ? iG : 0;
                }
                if (take) S.aux[p] = (uint32_t)(C + cb) | ((uint32_t)(nV + vb) << 16);
                C += totNe; nV += totNx; sz += totG;
            }
        }
        qt_sync<!NG>();
        // (e) new list: children (reverse creation order) then the untouched nodes; new V in creation order.  The slots of cc
        //     now hold the list position of the child (untouched node: its own new position in slot 0), QT3_NONE where there is none.
        int nKeep = 0;
        for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
            uint32_t ax[4], ogx[4], ogy[4], ocn[4]; uint2 c2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = min(pb0 + u * WAVE + lane, size - 1);
                ax[u] = S.aux[p]; ogx[u] = gx[p]; ogy[u] = gy[p]; ocn[u] = cn[p]; c2[u] = *(const uint2 *)(S.cc + 2 * p);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pb0 + u * WAVE + lane;
                const bool val = p < size;
                const bool keep = val && ax[u] == ~0u;
                const unsigned long long mk = __ballot(keep);
                if (keep) {
                    const int np = C + nKeep + __popcll(mk & lt);
                    gx2[np] = ogx[u]; gy2[np] = ogy[u]; cn2[np] = (uint16_t)ocn[u];
                    *(uint2 *)(S.cc + 2 * p) = make_uint2((uint32_t)np | (QT3_NONE << 16), QT3_NONE | (QT3_NONE << 16));
                } else if (val) {
                    const int ulx = ogx[u] & 0xFFFF, urx = ogx[u] >> 16, uly = ogy[u] & 0xFFFF, bry = ogy[u] >> 16;
                    const int mx = ulx + ((urx - ulx + 1) >> 1), my = uly + ((bry - uly + 1) >> 1);
                    int ci = (int)(ax[u] & 0xFFFF), vi = (int)(ax[u] >> 16);
                    const uint32_t cj[4] = {c2[u].x & 0xFFFF, c2[u].x >> 16, c2[u].y & 0xFFFF, c2[u].y >> 16};
                    uint32_t slot[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // n1 (UL), n2 (UR), n3 (BL), n4 (BR) in push order
                        slot[j] = QT3_NONE;
                        if (cj[j] > 0) {
                            const int np = C - 1 - ci;
                            const int cul = (j & 1) ? mx : ulx, cur_ = (j & 1) ? urx : mx;
                            const int cuy = (j & 2) ? my : uly, cby = (j & 2) ? bry : my;
                            gx2[np] = (uint32_t)cul | ((uint32_t)cur_ << 16);
                            gy2[np] = (uint32_t)cuy | ((uint32_t)cby << 16);
                            cn2[np] = (uint16_t)cj[j];
                            if (cj[j] > 1) S.V[vi++] = ((qt_u64)cj[j] << 32) | ((qt_u64)(unsigned)cul << 16) | (unsigned)np;
                            slot[j] = (uint32_t)np;
                            ++ci;
                        }
                    }
                    *(uint2 *)(S.cc + 2 * p) = make_uint2(slot[0] | (slot[1] << 16), slot[2] | (slot[3] << 16));
                }
                nKeep += __popcll(mk);
            }
        }
        const int size2 = C + nKeep;
        qt_sync<!NG>();
        // (f) prefix of the new counts; (g) list positions -> key offsets (write cursors)
        {
            int base = 0;
            for (int pb0 = 0; pb0 < size2; pb0 += 4 * WAVE) {
                int c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int p = pb0 + u * WAVE + lane; c[u] = p < size2 ? (int)cn2[p] : 0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb0 + u * WAVE + lane;
                    int total;
                    const int o = base + wave_excl_scan(c[u], total);
                    if (p < size2) bg2[p] = (uint16_t)o;
                    base += total;
                }
            }
        }
        qt_sync<!NG>();
        for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
            uint2 c2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c2[u] = *(const uint2 *)(S.cc + 2 * min(pb0 + u * WAVE + lane, size - 1));
            uint32_t o[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t sj[4] = {c2[u].x & 0xFFFF, c2[u].x >> 16, c2[u].y & 0xFFFF, c2[u].y >> 16};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[u][j] = sj[j] != QT3_NONE ? (uint32_t)bg2[sj[j]] : QT3_NONE;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pb0 + u * WAVE + lane;
                if (p < size) *(uint2 *)(S.cc + 2 * p) = make_uint2(o[u][0] | (o[u][1] << 16), o[u][2] | (o[u][3] << 16));
            }
        }
        qt_sync<!NG>();
        // (h) stable scatter of the keys to their new runs (in place when all keys are in registers)
        sweep(!inRegs, [&](const KeyInfo &k) {
            const bool div = k.act && S.aux[k.p] != ~0u;
            unsigned long long m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = __ballot(div && k.q == j);
            const unsigned long long mq = k.q == 0 ? m[0] : (k.q == 1 ? m[1] : (k.q == 2 ? m[2] : m[3]));
            const int sl = div ? k.q : 0;
            const uint32_t cursor = k.act ? (S.cc[2 * k.p + (sl >> 1)] >> (16 * (sl & 1))) & 0xFFFFu : 0u;
            const uint32_t dest = div ? cursor + (uint32_t)__popcll(mq & k.R & lt) : cursor + (uint32_t)k.e0;
            if (div && lane == k.lo) addSlots(k, m);  // (LDS operations of a wave complete in order: every lane has read its cursor)
            if (k.act) qt_key(B, dest) = k.key;
        });
        if (!inRegs) { KP t = A; A = B; B = t; }
        kkValid = false;
        cur ^= 1;
        size = size2;
        qt_sync<false>();
    };

    // ---- subdivision (:630-752): list walks until the next one could overshoot N, then sorted rounds (:701-751) until done.
    // (one call site: the pass is inlined and its state stays in registers)
    bool finish = false, sorted = false;
    while (!finish) {
        const int prevSize = size;
        int nSorted = -1;
        if (sorted) {
            nSorted = nV;
            if (lane == 0 && a.stop != 4) orbx_sort::sort(S.V, nSorted, QtSizedLess(), S.sortStack);
            qt_sync<!NG>();
        }
        pass(nSorted);
        if (size >= N || size == prevSize) finish = true;
        else if (!sorted && size + nV * 3 > N) {
            if (a.stop == 3) break;
            sorted = true;
        }
    }

    // ---- retain the best (first strictly greatest response) key of every node, in list order (:755-771) --
    {
        const uint16_t *cn = S.cnt[0] + cur * L;
        int base = 0;
        for (int pb = 0; pb < size; pb += WAVE) {
            const int p = pb + lane;
            const int n = p < size ? (int)cn[p] : 0;
            int total;
            const int b = base + wave_excl_scan(n, total);
            if (p < size) {
                uint32_t best = qt_key(A, b);
                for (int i = 1; i < n; ++i) {
                    const uint32_t key = qt_key(A, b + i);
                    if ((key >> 24) > (best >> 24)) best = key;
                }
                selOut[p] = best;
            }
            base += total;
        }
    }
    if (lane == 0) *selCnt = size;
}

static size_t qtNodeBytes(int P, int L)
{
    // gx, gy, beg, cnt (16 B) + fl, front, freeL (6 B) per pool entry; sized, prev (16 B) + order, tmpC, tmpK (6 B) per list entry
    return (((size_t)P * 22 + (size_t)L * 22 + 216 * 4) + 15) & ~(size_t)15;
}

#define QT_DEFERRED (-1)  // selCount marker: candidate set too large for the level-synchronous kernel, left to k_quadtree_serial

// common prologue: (frame, level) of this instance, its output slots and candidate count
__device__ __forceinline__ bool qt_instance(QtArgsK &a, int &f, int &level, int &M)
{
    if (!xcd_decode(a.nframes, a.levelsRun, f, level)) return false;
    level += a.levelBase;
    const int lane = threadIdx.x;
    const OrbxLevelGeom &g = a.geoms[level];
    const int *tc = a.tileCount + (size_t)f * a.tilesPerFrame + g.tileBase;
    M = 0;
    for (int tb = 0; tb < g.nTiles; tb += WAVE) {
        int v = tb + lane < g.nTiles ? tc[tb + lane] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        M += v;
    }
    // (the same value in every lane - say so: as a per-lane value it made `inRegs`, and with it every choice between the key buffers,
    // divergent in the compiler's eyes, and the key pointers per-lane 64-bit addresses instead of a scalar base + a 32-bit offset)
    M = __builtin_amdgcn_readfirstlane(M);
    return true;
}

template <int KCH>
__global__ __launch_bounds__(64) void k_quadtree(QtArgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qlds[];
    QtArgsK &a = *(QtArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    int f, level, M;
    if (a.pass2) {  // only what the first launch deferred (the common case: nothing)
        if (!xcd_decode(a.nframes, a.levelsRun, f, level)) return;
        if (a.selCount[(size_t)f * a.nlevels + level + a.levelBase] != QT_DEFERRED) return;
    }
    if (!qt_instance(a, f, level, M)) return;
    const OrbxLevelGeom &g = a.geoms[level];
    uint32_t *selOut = a.sel + (size_t)f * a.selFrameStride + g.selBase;
    int *selCnt = a.selCount + (size_t)f * a.nlevels + level;
    const int *tc = a.tileCount + (size_t)f * a.tilesPerFrame + g.tileBase;
    if (M == 0 || M > a.maxKeys) { if (threadIdx.x == 0) *selCnt = M == 0 ? 0 : QT_DEFERRED; return; }
    uint32_t *gA = a.scratchA + (size_t)f * a.candFrameStride + g.candBase;
    uint32_t *gB = a.scratchB + (size_t)f * a.candFrameStride + g.candBase;
    const bool inRegs = M <= KCH * WAVE;
    if (a.nodeScratch) {  // list capacity beyond LDS: node arrays (and keys) in global memory
        uint8_t *nodes = a.nodeScratch + ((size_t)f * a.levelsRun + (level - a.levelBase)) * (size_t)a.nodeBytes;
        qt3_body<true, KCH, QtGlobalKeys>(a, g, f, nodes, (QtGlobalKeys)gA, (QtGlobalKeys)(inRegs ? gA : gB), M, tc, selOut, selCnt);
        return;
    }
    // all keys in registers for a pass + one LDS buffer behind the node arrays; larger sets ping-pong in global scratch
    // The key buffers as GLOBAL pointers (the default: ORBX_QT_LDS_KEYS = 0).  As one generic pointer that is either the LDS buffer or the
    // global scratch, every key access was a flat load / store - which counts against BOTH wait counters, so that the next wait for an LDS
    // read of the node arrays also waited for the keys in flight, and back: in a kernel that is nothing but dependent latencies.
    // (ORBX_QT_LDS_KEYS - small key sets in an LDS buffer behind the node arrays, measured slower than the global scratch in round 3 - is
    // no longer an option of this kernel: its generic-pointer instance cost the kernel 46 more VGPRs)
    qt3_body<false, KCH, QtGlobalKeys>(a, g, f, qlds, (QtGlobalKeys)gA, (QtGlobalKeys)(inRegs ? gA : gB), M, tc, selOut, selCnt);
}

// =====================================================================================================
// The same level-synchronous formulation with ONE WORKGROUP OF QT4_NW WAVES per (frame, level) - for launches with few instances
// (a single frame, a 1080p stereo pair): there an instance is latency-bound and its two key sweeps per pass are what it spends its
// time in (a 1080p level holds 9 - 25 k candidates = up to 390 chunks per sweep on one wave).  The sweeps run on all waves, chunk c
// on wave c mod QT4_NW; the list steps (a, c - g) stay on wave 0.  What made the scatter sequential - a write cursor per (node,
// quadrant) that every chunk advances - is replaced by offsets known before the sweep: a key's rank inside its node's quadrant
// = (keys of that quadrant in the node's earlier chunks) + (rank inside the chunk).  The first term is only non-zero for the run
// that enters a chunk from the left; sweep (b) leaves the quadrant counts of every chunk's LAST run in Lc[], wave 1 turns them into
// inclusive prefixes, and with cs[c] = the last chunk before c that holds a run start
//      pre(c) = Lc[c - 1] - Lc[cs[c] - 1]
// (all chunks between cs[c] and c are one run of that node).  cb[c] = number of run starts before chunk c replaces the running count.
// =====================================================================================================
#define QT4_NW 8
#define QT4_NT (QT4_NW * WAVE)
#define QT4_KCH 8          // 64-key chunks per wave held in registers: sets of up to QT4_NW * QT4_KCH * 64 = 4096 keys are scattered in place

static size_t qt4NodeBytes(int L, int maxKeys)
{
    const size_t chunks = (size_t)maxKeys / 64 + 2;
    return ((qt3NodeBytes(L, maxKeys) + 15) & ~(size_t)15) + ((chunks * (8 + 2 + 2) + 16 + 15) & ~(size_t)15);
}

template <bool NG>
__device__ __forceinline__ void qt4_body(QtArgsK &a, const OrbxLevelGeom &g, int f, uint8_t *qlds, uint32_t *A, uint32_t *gB, int M,
                                         const int *tc, uint32_t *selOut, int *selCnt)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long lt = lanemask_lt();
    const int L = a.listCap;
    const int nChunks = (M + 63) >> 6;
    const bool inRegs = nChunks <= QT4_KCH * QT4_NW;
    uint32_t *B = inRegs ? A : gB;
    Qt3 S;
    S.V = (qt_u64 *)qlds;
    S.gx[0] = (uint32_t *)(S.V + L); S.gx[1] = S.gx[0] + L; S.gy[0] = S.gx[1] + L; S.gy[1] = S.gy[0] + L;
    S.cc = S.gy[1] + L; S.aux = S.cc + 2 * L;
    S.cnt[0] = (uint16_t *)(S.aux + L); S.cnt[1] = S.cnt[0] + L; S.beg[0] = S.cnt[1] + L; S.beg[1] = S.beg[0] + L + 2;
    S.proc = S.beg[1] + L + 2;
    S.sortStack = (int *)(S.proc + L);
    S.starts = (uint32_t *)(S.sortStack + 216);
    const int maxChunks = a.mwKeys / 64 + 2;
    qt_u64 *Lc = (qt_u64 *)(qlds + ((qt3NodeBytes(L, a.mwKeys) + 15) & ~(size_t)15));  // [maxChunks] last-run quadrant counts -> inclusive prefix
    uint16_t *cb = (uint16_t *)(Lc + maxChunks);                                       // [maxChunks] run starts before the chunk
    uint16_t *cs = cb + maxChunks;                                                      // [maxChunks] last chunk before it that holds a start
    int *bc = (int *)(cs + maxChunks);                                                  // [4] broadcast of wave 0's scalars
    auto sync = [&]() { __syncthreads(); };

    // ---- ordered gather of the FAST candidates (tile slots in cell-row-major order): a wave per tile, coalesced copies ------------
    const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
    const int N = g.quota;
    const int nIni = (int)roundf(__fdiv_rn((float)W, (float)H));
    const float hX = __fdiv_rn((float)W, (float)nIni);
    const bool groupRoots = nIni > 1;
    {
        int *tileOff = (int *)qlds;  // nTiles + 1 prefix offsets; the node arrays are not in use yet
        if (wv == 0) {
            int base = 0;
            for (int tb = 0; tb < g.nTiles; tb += WAVE) {
                const int t = tb + lane;
                const int myN = t < g.nTiles ? tc[t] : 0;
                int total;
                const int myOff = wave_excl_scan(myN, total);
                if (t < g.nTiles) tileOff[t] = base + myOff;
                base += total;
            }
            if (lane == 0) tileOff[g.nTiles] = base;
        }
        sync();
        const uint32_t *slots = a.cand + (size_t)f * a.candFrameStride + g.candBase;
        uint32_t *G = groupRoots ? gB : A;  // with several roots the keys are grouped by root on their way to A
        for (int t0 = wv; t0 < g.nTiles; t0 += 2 * QT4_NW) {  // two tiles per step: both loads in flight
            const int t1 = t0 + QT4_NW;
            const int o0 = tileOff[t0], n0 = tileOff[t0 + 1] - o0;
            const int o1 = t1 < g.nTiles ? tileOff[t1] : 0, n1 = t1 < g.nTiles ? tileOff[t1 + 1] - o1 : 0;
            const uint32_t *s0 = slots + (size_t)t0 * g.slotCap, *s1 = slots + (size_t)t1 * g.slotCap;
            for (int i = lane; i < max(n0, n1); i += WAVE) {
                const uint32_t v0 = i < n0 ? s0[i] : 0u, v1 = i < n1 ? s1[i] : 0u;
                if (i < n0) G[o0 + i] = v0;
                if (i < n1) G[o1 + i] = v1;
            }
        }
    }
    sync();
    if (a.stop == 1) { if (tid == 0) *selCnt = 0; return; }

    // ---- roots (:589-626): push_back in order, keys stably grouped by root ---------------------------------------------------------
    // One or two roots on camera images.  The grouping is a stable partition like the scatter of a pass: per chunk the counts per root
    // (Lc), their prefix over the chunks (wave 1), then dest = start of the root + keys of that root in earlier chunks + rank in the chunk.
    int cur = 0, size = 0;
    if (!groupRoots) {
        if (tid == 0) {
            S.gx[0][0] = 0u | ((uint32_t)(int)__fmul_rn(hX, 1.0f) << 16);
            S.gy[0][0] = 0u | ((uint32_t)H << 16);
            S.cnt[0][0] = (uint16_t)M;
        }
        size = 1;
    } else if (nIni <= 4) {
        auto rootOf = [&](uint32_t key) { return (int)__fdiv_rn((float)(key & 0xFFF), hX); };
        for (int c = wv; c < nChunks; c += QT4_NW) {
            const int e = c * WAVE + lane;
            const uint32_t key = e < M ? gB[e] : 0u;
            const int r = e < M ? rootOf(key) : -1;
            const qt_u64 v = (qt_u64)__popcll(__ballot(r == 0)) | ((qt_u64)__popcll(__ballot(r == 1)) << 16) |
                             ((qt_u64)__popcll(__ballot(r == 2)) << 32) | ((qt_u64)__popcll(__ballot(r == 3)) << 48);
            if (lane == 0) Lc[c] = v;
        }
        sync();
        if (wv == 1) {
            uint32_t blo = 0, bhi = 0;
            for (int c0 = 0; c0 < nChunks; c0 += WAVE) {
                const int c = c0 + lane;
                const qt_u64 v = c < nChunks ? Lc[c] : 0ull;
                int tlo, thi;
                const uint32_t elo = (uint32_t)wave_excl_scan((int)(uint32_t)v, tlo), ehi = (uint32_t)wave_excl_scan((int)(uint32_t)(v >> 32), thi);
                if (c < nChunks) Lc[c] = (qt_u64)(blo + elo + (uint32_t)v) | ((qt_u64)(bhi + ehi + (uint32_t)(v >> 32)) << 32);
                blo += (uint32_t)tlo; bhi += (uint32_t)thi;
            }
        }
        sync();
        const qt_u64 tot = Lc[nChunks - 1];
        int rootStart[4], off = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cnt = (int)((tot >> (16 * r)) & 0xFFFF);
            rootStart[r] = off;
            if (cnt > 0) {  // (roots without keys are not pushed, :621-626)
                if (tid == 0) {
                    S.gx[0][size] = (uint32_t)(int)__fmul_rn(hX, (float)r) | ((uint32_t)(int)__fmul_rn(hX, (float)(r + 1)) << 16);
                    S.gy[0][size] = 0u | ((uint32_t)H << 16);
                    S.cnt[0][size] = (uint16_t)cnt;
                }
                ++size;
            }
            off += cnt;
        }
        for (int c = wv; c < nChunks; c += QT4_NW) {
            const int e = c * WAVE + lane;
            const uint32_t key = e < M ? gB[e] : 0u;
            const int r = e < M ? rootOf(key) : -1;
            const unsigned long long m0 = __ballot(r == 0), m1 = __ballot(r == 1), m2 = __ballot(r == 2), m3 = __ballot(r == 3);
            const unsigned long long mr = r == 0 ? m0 : (r == 1 ? m1 : (r == 2 ? m2 : m3));
            const qt_u64 pre = c > 0 ? Lc[c - 1] : 0ull;
            if (e < M) {
                const int st = r == 0 ? rootStart[0] : (r == 1 ? rootStart[1] : (r == 2 ? rootStart[2] : rootStart[3]));
                A[st + (int)((pre >> (16 * r)) & 0xFFFF) + __popcll(mr & lt)] = key;
            }
        }
    } else {
        // more than four roots (aspect ratio above 4.5): wave 0 groups them one after the other
        int off = 0;
        for (int r = 0; r < nIni; ++r) {
            int cnt = 0;
            if (wv == 0)
                for (int base = 0; base < M; base += WAVE) {
                    const bool act = base + lane < M;
                    const uint32_t key = act ? gB[base + lane] : 0u;
                    const bool mine = act && (int)__fdiv_rn((float)(key & 0xFFF), hX) == r;
                    const unsigned long long m = __ballot(mine);
                    if (mine) A[off + cnt + __popcll(m & lt)] = key;
                    cnt += __popcll(m);
                }
            if (wv == 0 && cnt > 0 && lane == 0) {
                S.gx[0][size] = (uint32_t)(int)__fmul_rn(hX, (float)r) | ((uint32_t)(int)__fmul_rn(hX, (float)(r + 1)) << 16);
                S.gy[0][size] = 0u | ((uint32_t)H << 16);
                S.cnt[0][size] = (uint16_t)cnt;
            }
            if (cnt > 0) ++size;
            off += cnt;
        }
        if (tid == 0) bc[0] = size;  // (wave 0 counted: the other waves take the list size from it)
        sync();
        size = bc[0];
    }
    sync();
    if (a.stop == 2) { if (tid == 0) *selCnt = 0; return; }

    uint32_t kk[QT4_KCH];  // keys of chunks cb0 + u * QT4_NW + wv of the running batch
    bool kkValid = false;
    int nV = 0;  // entries of V after a pass
    auto pass = [&](int nSorted) {
        // (generation c of an array = base + c * length: indexing the pointer pairs with `cur` made them generic pointers and their
        // accesses flat loads / stores - see qt3_body)
        const int nxt = cur ^ 1;
        uint32_t *gx = S.gx[0] + cur * L, *gy = S.gy[0] + cur * L;
        uint16_t *cn = S.cnt[0] + cur * L, *bg = S.beg[0] + cur * (L + 2);
        uint32_t *gx2 = S.gx[0] + nxt * L, *gy2 = S.gy[0] + nxt * L;
        uint16_t *cn2 = S.cnt[0] + nxt * L, *bg2 = S.beg[0] + nxt * (L + 2);
        for (int i = tid; i < nChunks * 2 + 2; i += QT4_NT) S.starts[i] = 0;
        sync();
        if (wv == 0) {
            // (a) run starts: exclusive prefix of the counts, bitmap of first keys; child counters to zero
            int base = 0;
            for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
                int c[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int p = pb0 + u * WAVE + lane; c[u] = p < size ? (int)cn[p] : 0; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb0 + u * WAVE + lane;
                    int total;
                    const int o = base + wave_excl_scan(c[u], total);
                    if (p < size) {
                        bg[p] = (uint16_t)o;
                        atomicOr(&S.starts[o >> 5], 1u << (o & 31));
                        *(uint2 *)(S.cc + 2 * p) = make_uint2(0, 0);
                        S.aux[p] = ~0u;
                    }
                    base += total;
                }
            }
            if (lane == 0) bg[size] = (uint16_t)M;
            qt_sync<!NG>();
            // (a2) per chunk: run starts before it, last earlier chunk with a start
            int nb = 0, lastS = 0;
            for (int c0 = 0; c0 < nChunks; c0 += WAVE) {
                const int c = c0 + lane;
                const unsigned long long w = c < nChunks ? ((unsigned long long)S.starts[2 * c] | ((unsigned long long)S.starts[2 * c + 1] << 32)) : 0ull;
                const int n = __popcll(w);
                int total;
                const int ex = wave_excl_scan(n, total);
                const unsigned long long m = __ballot(n > 0), prev = m & lt;
                if (c < nChunks) {
                    cb[c] = (uint16_t)(nb + ex);
                    cs[c] = (uint16_t)(prev ? c0 + (63 - __clzll(prev)) : lastS);
                }
                if (m) lastS = c0 + (63 - __clzll(m));
                nb += total;
            }
        }
        sync();
        // key -> (node, quadrant, lane range of the node inside its 64-key chunk)
        struct KeyInfo { uint32_t key; int p, q, lo, e0, c; bool act, big; unsigned long long R, w; };
        auto classify = [&](int c, unsigned long long w, int p, uint32_t key, uint32_t ngx, uint32_t ngy, int nb, int nc) -> KeyInfo {
            KeyInfo k;
            const int base = c * WAVE, e = base + lane;
            k.act = e < M; k.p = p; k.key = key; k.c = c; k.w = w;
            const int ulx = ngx & 0xFFFF, urx = ngx >> 16, uly = ngy & 0xFFFF, bry = ngy >> 16;
            const int mx = ulx + ((urx - ulx + 1) >> 1), my = uly + ((bry - uly + 1) >> 1);  // UL + ceil(extent / 2)
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            k.q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
            k.big = k.act && nc > 1;
            k.e0 = e - nb;
            k.lo = max(nb - base, 0);
            const int hi = min(nb + nc - base, 64);
            k.R = (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << k.lo) - 1ull);
            return k;
        };
        // a sweep over all keys: wave wv takes chunks wv, wv + QT4_NW, ...; batches of QT4_KCH chunks per wave (keys -> registers),
        // inside a batch 4 chunks per step (bitmap words, then owners, then node fields), then `body(KeyInfo)` chunk by chunk
        auto sweep = [&](bool reload, auto &&body) {
            for (int cb0 = 0; cb0 < nChunks; cb0 += QT4_KCH * QT4_NW) {
                if (reload) {
#pragma unroll
                    for (int u = 0; u < QT4_KCH; ++u) {
                        const int c = cb0 + u * QT4_NW + wv;
                        const int e = c * WAVE + lane;
                        if (c < nChunks) kk[u] = e < M ? A[e] : 0u;
                    }
                }
#pragma unroll
                for (int u0 = 0; u0 < QT4_KCH; u0 += 4) {
                    if (cb0 + u0 * QT4_NW + wv < nChunks) {  // (no `break`: the loop must unroll completely, kk is indexed statically)
                    unsigned long long w[4];
                    int cbv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = cb0 + (u0 + u) * QT4_NW + wv;
                        const bool in = c < nChunks;
                        w[u] = in ? ((unsigned long long)S.starts[2 * c] | ((unsigned long long)S.starts[2 * c + 1] << 32)) : 0ull;
                        cbv[u] = in ? (int)cb[c] : 0;
                    }
                    int pp[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = cb0 + (u0 + u) * QT4_NW + wv;
                        pp[u] = c * WAVE + lane < M ? cbv[u] + __popcll(w[u] & (lt | (1ull << lane))) - 1 : 0;
                    }
                    uint32_t ngx[4], ngy[4]; int nb[4], nc[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { ngx[u] = gx[pp[u]]; ngy[u] = gy[pp[u]]; nb[u] = (int)bg[pp[u]]; nc[u] = (int)cn[pp[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int c = cb0 + (u0 + u) * QT4_NW + wv;
                        if (c < nChunks) body(classify(c, w[u], pp[u], kk[u0 + u], ngx[u], ngy[u], nb[u], nc[u]));
                    }
                    }
                }
            }
        };
        // (b) child counts (LDS atomics, one per run and pair of quadrants) and the quadrant counts of every chunk's last run
        sweep(!kkValid, [&](const KeyInfo &k) {
            unsigned long long m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = __ballot(k.big && k.q == j);
            if (k.big && lane == k.lo) {
                const uint32_t lo2 = (uint32_t)__popcll(m[0] & k.R) | ((uint32_t)__popcll(m[1] & k.R) << 16);
                const uint32_t hi2 = (uint32_t)__popcll(m[2] & k.R) | ((uint32_t)__popcll(m[3] & k.R) << 16);
                if (lo2) atomicAdd(&S.cc[2 * k.p], lo2);
                if (hi2) atomicAdd(&S.cc[2 * k.p + 1], hi2);
            }
            const int last = min(WAVE, M - k.c * WAVE) - 1;  // last key of the chunk (wave-uniform)
            const unsigned long long Rl = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k.R, last)) |
                                          ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k.R >> 32), last) << 32);
            if (lane == 0)
                Lc[k.c] = (qt_u64)__popcll(m[0] & Rl) | ((qt_u64)__popcll(m[1] & Rl) << 16) | ((qt_u64)__popcll(m[2] & Rl) << 32) |
                          ((qt_u64)__popcll(m[3] & Rl) << 48);
        });
        kkValid = inRegs;
        sync();
        int size2 = 0;
        if (wv == 1) {
            // inclusive prefix of Lc (four 16-bit fields: a quadrant's total is at most M < 65536, no carries between fields)
            uint32_t blo = 0, bhi = 0;
            for (int c0 = 0; c0 < nChunks; c0 += WAVE) {
                const int c = c0 + lane;
                const qt_u64 v = c < nChunks ? Lc[c] : 0ull;
                int tlo, thi;
                const uint32_t elo = (uint32_t)wave_excl_scan((int)(uint32_t)v, tlo), ehi = (uint32_t)wave_excl_scan((int)(uint32_t)(v >> 32), thi);
                if (c < nChunks) Lc[c] = (qt_u64)(blo + elo + (uint32_t)v) | ((qt_u64)(bhi + ehi + (uint32_t)(v >> 32)) << 32);
                blo += (uint32_t)tlo; bhi += (uint32_t)thi;
            }
        }
        if (wv == 0) {
            // (c) nodes to divide, in processing order
            int nProc = 0;
            if (nSorted < 0) {
                for (int pb = 0; pb < size; pb += WAVE) {
                    const int p = pb + lane;
                    const bool big = p < size && cn[p] > 1;
                    const unsigned long long m = __ballot(big);
                    if (big) S.proc[nProc + __popcll(m & lt)] = (uint16_t)p;
                    nProc += __popcll(m);
                }
            } else {
                for (int t = lane; t < nSorted; t += WAVE) S.proc[t] = (uint16_t)(S.V[nSorted - 1 - t] & 0xFFFF);
                nProc = nSorted;
            }
            qt_sync<!NG>();
            // (d) prefix sums over the processing order: list size before each divide (a sorted round stops dividing once the
            //     list holds N nodes), creation index of the first child, index of the first new V entry
            int C = 0;
            nV = 0;
            {
                int sz = size;
                bool stop = false;
                for (int tb = 0; tb < nProc && !stop; tb += WAVE) {
                    const int t = tb + lane;
                    const bool val = t < nProc;
                    const int p = val ? S.proc[t] : 0;
                    const uint2 c2 = *(const uint2 *)(S.cc + 2 * p);
                    const uint32_t c0 = c2.x & 0xFFFF, c1 = c2.x >> 16, c2_ = c2.y & 0xFFFF, c3 = c2.y >> 16;
                    const int ne = val ? (c0 > 0) + (c1 > 0) + (c2_ > 0) + (c3 > 0) : 0;
                    const int nx = val ? (c0 > 1) + (c1 > 1) + (c2_ > 1) + (c3 > 1) : 0;
                    int totNe, totNx, totG;
                    const int cbx = wave_excl_scan(ne, totNe);
                    const int vb = wave_excl_scan(nx, totNx);
                    const int gEx = wave_excl_scan(val ? ne - 1 : 0, totG);
                    bool take = val;
                    if (nSorted >= 0) {
                        // the reference divides node t and THEN breaks if size >= N: node t is divided iff the size before it is < N
                        take = val && sz + gEx < N;
                        const int nTake = __popcll(__ballot(take));  // a prefix of the chunk: sizes never decrease
                        if (nTake < min(WAVE, nProc - tb)) stop = true;
                        const int last = max(nTake - 1, 0);
                        const int iNe = __shfl(cbx + ne, last), iNx = __shfl(vb + nx, last), iG = __shfl(gEx + ne - 1, last);
                        totNe = nTake > 0 ? iNe : 0; totNx = nTake > 0 ? iNx : 0; totG = nTake > 0 ? iG : 0;
                    }
                    if (take) S.aux[p] = (uint32_t)(C + cbx) | ((uint32_t)(nV + vb) << 16);
                    C += totNe; nV += totNx; sz += totG;
                }
            }
            qt_sync<!NG>();
            // (e) new list: children (reverse creation order) then the untouched nodes; new V in creation order.  The slots of cc
            //     now hold the list position of the child (untouched node: its own new position in slot 0), QT3_NONE where there is none.
            int nKeep = 0;
            for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
                uint32_t ax[4], ogx[4], ogy[4], ocn[4]; uint2 c2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = min(pb0 + u * WAVE + lane, size - 1);
                    ax[u] = S.aux[p]; ogx[u] = gx[p]; ogy[u] = gy[p]; ocn[u] = cn[p]; c2[u] = *(const uint2 *)(S.cc + 2 * p);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb0 + u * WAVE + lane;
                    const bool val = p < size;
                    const bool keep = val && ax[u] == ~0u;
                    const unsigned long long mk = __ballot(keep);
                    if (keep) {
                        const int np = C + nKeep + __popcll(mk & lt);
                        gx2[np] = ogx[u]; gy2[np] = ogy[u]; cn2[np] = (uint16_t)ocn[u];
                        *(uint2 *)(S.cc + 2 * p) = make_uint2((uint32_t)np | (QT3_NONE << 16), QT3_NONE | (QT3_NONE << 16));
                    } else if (val) {
                        const int ulx = ogx[u] & 0xFFFF, urx = ogx[u] >> 16, uly = ogy[u] & 0xFFFF, bry = ogy[u] >> 16;
                        const int mx = ulx + ((urx - ulx + 1) >> 1), my = uly + ((bry - uly + 1) >> 1);
                        int ci = (int)(ax[u] & 0xFFFF), vi = (int)(ax[u] >> 16);
                        const uint32_t cj[4] = {c2[u].x & 0xFFFF, c2[u].x >> 16, c2[u].y & 0xFFFF, c2[u].y >> 16};
                        uint32_t slot[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {  // n1 (UL), n2 (UR), n3 (BL), n4 (BR) in push order
                            slot[j] = QT3_NONE;
                            if (cj[j] > 0) {
                                const int np = C - 1 - ci;
                                const int cul = (j & 1) ? mx : ulx, cur_ = (j & 1) ? urx : mx;
                                const int cuy = (j & 2) ? my : uly, cby = (j & 2) ? bry : my;
                                gx2[np] = (uint32_t)cul | ((uint32_t)cur_ << 16);
                                gy2[np] = (uint32_t)cuy | ((uint32_t)cby << 16);
                                cn2[np] = (uint16_t)cj[j];
                                if (cj[j] > 1) S.V[vi++] = ((qt_u64)cj[j] << 32) | ((qt_u64)(unsigned)cul << 16) | (unsigned)np;
                                slot[j] = (uint32_t)np;
                                ++ci;
                            }
                        }
                        *(uint2 *)(S.cc + 2 * p) = make_uint2(slot[0] | (slot[1] << 16), slot[2] | (slot[3] << 16));
                    }
                    nKeep += __popcll(mk);
                }
            }
            size2 = C + nKeep;
            qt_sync<!NG>();
            // (f) prefix of the new counts; (g) list positions -> key offsets of the children's runs
            {
                int base = 0;
                for (int pb0 = 0; pb0 < size2; pb0 += 4 * WAVE) {
                    int c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int p = pb0 + u * WAVE + lane; c[u] = p < size2 ? (int)cn2[p] : 0; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int p = pb0 + u * WAVE + lane;
                        int total;
                        const int o = base + wave_excl_scan(c[u], total);
                        if (p < size2) bg2[p] = (uint16_t)o;
                        base += total;
                    }
                }
            }
            qt_sync<!NG>();
            for (int pb0 = 0; pb0 < size; pb0 += 4 * WAVE) {
                uint2 c2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) c2[u] = *(const uint2 *)(S.cc + 2 * min(pb0 + u * WAVE + lane, size - 1));
                uint32_t o[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t sj[4] = {c2[u].x & 0xFFFF, c2[u].x >> 16, c2[u].y & 0xFFFF, c2[u].y >> 16};
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[u][j] = sj[j] != QT3_NONE ? (uint32_t)bg2[sj[j]] : QT3_NONE;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = pb0 + u * WAVE + lane;
                    if (p < size) *(uint2 *)(S.cc + 2 * p) = make_uint2(o[u][0] | (o[u][1] << 16), o[u][2] | (o[u][3] << 16));
                }
            }
            if (lane == 0) { bc[0] = size2; bc[1] = nV; }
        }
        sync();
        size2 = bc[0]; nV = bc[1];
        // (h) stable scatter of the keys to their new runs (in place when all keys are in registers): no cursor is advanced
        sweep(!inRegs, [&](const KeyInfo &k) {
            const bool div = k.act && S.aux[k.p] != ~0u;
            unsigned long long m[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) m[j] = __ballot(div && k.q == j);
            const unsigned long long mq = k.q == 0 ? m[0] : (k.q == 1 ? m[1] : (k.q == 2 ? m[2] : m[3]));
            const int sl = div ? k.q : 0;
            const uint32_t start = k.act ? (S.cc[2 * k.p + (sl >> 1)] >> (16 * (sl & 1))) & 0xFFFFu : 0u;
            // keys of this quadrant in the node's earlier chunks: only for the run that enters the chunk from the left
            qt_u64 pre = 0;
            if (k.c > 0 && !(k.w & 1ull)) {
                const int s = (int)cs[k.c];
                pre = Lc[k.c - 1] - (s > 0 ? Lc[s - 1] : 0ull);
            }
            const bool head = (k.w & ((2ull << lane) - 1ull)) == 0ull;  // no run start at or before this lane
            const uint32_t before = head ? (uint32_t)(pre >> (16 * k.q)) & 0xFFFFu : 0u;
            const uint32_t dest = div ? start + before + (uint32_t)__popcll(mq & k.R & lt) : start + (uint32_t)k.e0;
            if (k.act) B[dest] = k.key;
        });
        if (!inRegs) { uint32_t *t = A; A = B; B = t; }
        kkValid = false;
        cur ^= 1;
        size = size2;
        sync();
    };

    // std::sort of V[0, n) by compareNodes (:569-582, :708) - the permutation libstdc++'s introsort leaves, ties included - in the
    // data-parallel formulation of libstdcxx_sort.h (sort_pairing; pinned against std::sort by tests/cpp/test_sort_emulation.cpp):
    // wave 0 runs the introsort loop with every __unguarded_partition as one sweep (stoppers compacted with ballots, the k-th left one
    // swapped with the k-th right one), then all threads place every element at its stable rank (= the final insertion sort).
    // On one lane the sort of a 1080p level's ~500 expandable nodes took 0.3 ms per sorted round.
    auto sortV = [&](int n) {
        if (n <= 1) return;
        const QtSizedLess less;
        uint16_t *LP = S.proc, *RP = S.beg[0] + (cur ^ 1) * (L + 2);  // free between two passes
        if (wv == 0) {
            int *stFirst = S.sortStack, *stLast = stFirst + 72, *stDepth = stFirst + 144;
            int sp = 1;
            if (lane == 0) { stFirst[0] = 0; stLast[0] = n; stDepth[0] = 2 * (31 - __clz(n)); }
            while (sp > 0) {
                qt_sync<!NG>();
                --sp;
                const int first = __builtin_amdgcn_readfirstlane(stFirst[sp]);
                int last = __builtin_amdgcn_readfirstlane(stLast[sp]), depth = __builtin_amdgcn_readfirstlane(stDepth[sp]);
                while (last - first > 16) {
                    if (depth == 0) {
                        if (lane == 0) orbx_sort::heap_sort(S.V + first, last - first, less);
                        break;
                    }
                    --depth;
                    // __unguarded_partition_pivot: median of (first + 1, mid, last - 1) moved to first
                    const int ia = first + 1, ib = first + (last - first) / 2, ic = last - 1;
                    const qt_u64 vf = S.V[first], va = S.V[ia], vb = S.V[ib], vc = S.V[ic];
                    int im;
                    if (less(va, vb)) {
                        if (less(vb, vc)) im = ib;
                        else if (less(va, vc)) im = ic;
                        else im = ia;
                    } else if (less(va, vc)) im = ia;
                    else if (less(vb, vc)) im = ic;
                    else im = ib;
                    const qt_u64 pivot = im == ia ? va : (im == ib ? vb : vc);
                    qt_sync<!NG>();
                    if (lane == 0) { S.V[im] = vf; S.V[first] = pivot; }
                    qt_sync<!NG>();
                    // __unguarded_partition(first + 1, last, pivot): stoppers of both scans, in position order
                    int nL = 0, nR = 0;
                    for (int b = first + 1; b < last; b += WAVE) {
                        const int q = b + lane;
                        const bool act = q < last;
                        const qt_u64 v = act ? S.V[q] : 0ull;
                        const bool ls = act && !less(v, pivot), rs = act && !less(pivot, v);
                        const unsigned long long mL = __ballot(ls), mR = __ballot(rs);
                        if (ls) LP[nL + __popcll(mL & lt)] = (uint16_t)q;
                        if (rs) RP[nR + __popcll(mR & lt)] = (uint16_t)q;  // ascending: the k-th from the right is RP[nR - 1 - k]
                        nL += __popcll(mL); nR += __popcll(mR);
                    }
                    qt_sync<!NG>();
                    int K = 0;  // swaps: pairs with L_k < R_k, a prefix
                    const int kmax = min(nL, nR);
                    for (int k0 = 0; k0 < kmax; k0 += WAVE) {
                        const int k = k0 + lane;
                        const bool ok = k < kmax && LP[k] < RP[nR - 1 - k];
                        const int c = __popcll(__ballot(ok));
                        K += c;
                        if (c < WAVE) break;
                    }
                    for (int k = lane; k < K; k += WAVE) {
                        const int pl = LP[k], pr = RP[nR - 1 - k];
                        const qt_u64 vl = S.V[pl], vr = S.V[pr];
                        S.V[pl] = vr; S.V[pr] = vl;
                    }
                    const int rprev = K > 0 ? (int)RP[nR - K] : last;
                    const int lk = K < nL ? (int)LP[K] : 0x7FFFFFFF;
                    const int cut = __builtin_amdgcn_readfirstlane(lk < rprev ? lk : rprev);
                    qt_sync<!NG>();
                    // recurse on [cut, last) with the decremented depth, loop on [first, cut)
                    if (lane == 0) { stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; }
                    ++sp;
                    last = cut;
                }
            }
        }
        sync();
        // __final_insertion_sort = a stable sort of this arrangement: every element to its rank
        if (n <= 2 * QT4_NT) {
            const int i0 = tid, i1 = tid + QT4_NT;
            const qt_u64 e0 = i0 < n ? S.V[i0] : 0ull, e1 = i1 < n ? S.V[i1] : 0ull;
            const qt_u64 k0 = e0 >> 16, k1 = e1 >> 16;
            int r0 = 0, r1 = 0;
            for (int j = 0; j < n; ++j) {
                const qt_u64 kj = S.V[j] >> 16;
                r0 += (kj < k0 || (kj == k0 && j < i0)) ? 1 : 0;
                r1 += (kj < k1 || (kj == k1 && j < i1)) ? 1 : 0;
            }
            sync();
            if (i0 < n) S.V[r0] = e0;
            if (i1 < n) S.V[r1] = e1;
        } else if (tid == 0) {
            orbx_sort::PtrAccessor<qt_u64> acc{S.V};
            orbx_sort::insertion_sort<qt_u64>(acc, 0, 16, less);
            for (int i = 16; i != n; ++i) orbx_sort::unguarded_linear_insert<qt_u64>(acc, i, less);
        }
    };

    // ---- subdivision (:630-752): list walks until the next one could overshoot N, then sorted rounds (:701-751) until done
    bool finish = false, sorted = false;
    while (!finish) {
        const int prevSize = size;
        int nSorted = -1;
        if (sorted) {
            nSorted = nV;
            if (a.stop != 4) sortV(nSorted);
            sync();
        }
        pass(nSorted);
        if (size >= N || size == prevSize) finish = true;
        else if (!sorted && size + nV * 3 > N) {
            if (a.stop == 3) break;
            sorted = true;
        }
    }

    // ---- retain the best (first strictly greatest response) key of every node, in list order (:755-771) --
    {
        const uint16_t *cn = S.cnt[0] + cur * L, *bg = S.beg[0] + cur * (L + 2);
        for (int p = tid; p < size; p += QT4_NT) {
            const int n = (int)cn[p], b = (int)bg[p];
            uint32_t best = A[b];
            for (int i = 1; i < n; ++i) {
                const uint32_t key = A[b + i];
                if ((key >> 24) > (best >> 24)) best = key;
            }
            selOut[p] = best;
        }
    }
    if (tid == 0) *selCnt = size;
}

__global__ __launch_bounds__(QT4_NT) void k_quadtree_mw(QtArgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qlds[];
    QtArgsK &a = *(QtArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    // level-major: all instances of the largest level first (their candidate sets are the largest: longest job first)
    const int f = (int)(blockIdx.x % (unsigned)a.nframes);
    const int level = (int)(blockIdx.x / (unsigned)a.nframes) + a.levelBase;
    if (level >= a.levelBase + a.levelsRun) return;
    const OrbxLevelGeom &g = a.geoms[level];
    const int *tc = a.tileCount + (size_t)f * a.tilesPerFrame + g.tileBase;
    int M = 0;
    for (int tb = 0; tb < g.nTiles; tb += WAVE) {  // (every wave counts for itself)
        int v = tb + (int)(threadIdx.x & 63) < g.nTiles ? tc[tb + (threadIdx.x & 63)] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        M += v;
    }
    M = __builtin_amdgcn_readfirstlane(M);  // (wave-uniform: see qt_instance)
    uint32_t *selOut = a.sel + (size_t)f * a.selFrameStride + g.selBase;
    int *selCnt = a.selCount + (size_t)f * a.nlevels + level;
    if (M == 0 || M > a.maxKeys) { if (threadIdx.x == 0) *selCnt = M == 0 ? 0 : QT_DEFERRED; return; }
    uint32_t *gA = a.scratchA + (size_t)f * a.candFrameStride + g.candBase;
    uint32_t *gB = a.scratchB + (size_t)f * a.candFrameStride + g.candBase;
    if (a.nodeScratch) {  // list capacity beyond LDS: node arrays in global memory
        uint8_t *nodes = a.nodeScratch + ((size_t)f * a.levelsRun + (level - a.levelBase)) * (size_t)a.nodeBytes;
        qt4_body<true>(a, g, f, nodes, gA, gB, M, tc, selOut, selCnt);
        return;
    }
    // launches with very few instances keep candidate sets that are scattered in place (<= 4096 keys) in LDS behind the node arrays:
    // every pass re-reads the keys, and a single frame's level-0 instance is the longest item of the call
    uint32_t *lK = (uint32_t *)(qlds + a.nodeBytes);
    qt4_body<false>(a, g, f, qlds, M <= a.ldsKeys ? lK : gA, gB, M, tc, selOut, selCnt);
}

// serial formulation (one DivideNode at a time): every instance (a.variant == 1, A/B runs) or only those the kernel above
// deferred.  One key buffer in LDS when it fits; the second one is only used while a node still holds more than 64 keys (the
// first few divides), so it always lives in global scratch.
__global__ __launch_bounds__(64) void k_quadtree_serial(QtArgs)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qlds[];
    QtArgsK &a = *(QtArgsK *)__builtin_amdgcn_kernarg_segment_ptr();
    int f, level, M;
    if (!xcd_decode(a.nframes, a.levelsRun, f, level)) return;
    if (a.variant == 3 && a.selCount[(size_t)f * a.nlevels + level + a.levelBase] != QT_DEFERRED) return;  // the common case
    if (!qt_instance(a, f, level, M)) return;
    const OrbxLevelGeom &g = a.geoms[level];
    uint32_t *selOut = a.sel + (size_t)f * a.selFrameStride + g.selBase;
    int *selCnt = a.selCount + (size_t)f * a.nlevels + level;
    const int *tc = a.tileCount + (size_t)f * a.tilesPerFrame + g.tileBase;
    if (M == 0) { if (threadIdx.x == 0) *selCnt = 0; return; }
    uint32_t *gA = a.scratchA + (size_t)f * a.candFrameStride + g.candBase;
    uint32_t *gB = a.scratchB + (size_t)f * a.candFrameStride + g.candBase;
    if (a.nodeScratch) {
        uint8_t *nodes = a.nodeScratch + ((size_t)f * a.levelsRun + (level - a.levelBase)) * (size_t)a.nodeBytes;
        qt_body<false, true>(a, g, f, level, nodes, gA, gB, M, tc, selOut, selCnt);
    } else if (M <= a.ldsKeys) qt_body<true, false>(a, g, f, level, qlds, (uint32_t *)(qlds + a.nodeBytes), gB, M, tc, selOut, selCnt);
    else qt_body<false, false>(a, g, f, level, qlds, gA, gB, M, tc, selOut, selCnt);
}

// Global scratch the quadtree launch of levels [levelBase, levelBase + levelsRun) needs for its node arrays: 0 while they fit in
// LDS (list capacity up to ~3500 nodes), else the per-instance bytes of the larger of the two kernels x instances.
size_t orbxQuadtreeScratchBytes(int nframes, const OrbxLevelGeom *geomsHost, int levelBase, int levelsRun)
{
    int maxRoots = 1, maxTiles = 1, maxQuota = 0;
    for (int l = levelBase; l < levelBase + levelsRun; ++l) {
        const OrbxLevelGeom &g = geomsHost[l];
        maxQuota = std::max(maxQuota, g.quota);
        const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
        maxRoots = std::max(maxRoots, (int)roundf((float)W / (float)H));
        maxTiles = std::max(maxTiles, g.nTiles);
    }
    const int listCap = (std::max(maxQuota + 16, 4 * maxRoots + 8) + 3) & ~3;
    const int pool = (2 * maxQuota + 64 + 4 * maxRoots + 3) & ~3;
    const size_t tileOffBytes = (((size_t)maxTiles + 1) * 4 + 15) & ~(size_t)15;
    const size_t nb = std::max(std::max(std::max(qt3NodeBytes(listCap), qt4NodeBytes(listCap, QT3_MAX_KEYS2)), qtNodeBytes(pool, listCap)), tileOffBytes);
    if (nb + (size_t)QT_LDS_KEYS * sizeof(uint32_t) <= 160 * 1024) return 0;
    return nb * (size_t)nframes * (size_t)levelsRun;
}

hipError_t orbxLaunchQuadtree(hipStream_t st, int nframes, int nlevels, const OrbxLevelGeom *geomsHost,
                              const OrbxLevelGeom *geomsDev, const uint32_t *cand, long long candFrameStride,
                              const int *tileCount, int tilesPerFrame, uint32_t *scratchA, uint32_t *scratchB,
                              uint32_t *sel, int selFrameStride, int *selCount, int maxQuota, int levelBase, int levelsRun,
                              uint8_t *nodeScratch, size_t nodeScratchBytes, int which)
{
    QtArgs a;
    a.geoms = geomsDev;
    a.cand = cand; a.candFrameStride = candFrameStride;
    a.tileCount = tileCount; a.tilesPerFrame = tilesPerFrame;
    a.scratchA = scratchA; a.scratchB = scratchB;
    a.sel = sel; a.selFrameStride = selFrameStride; a.selCount = selCount;
    a.nlevels = nlevels; a.nframes = nframes;
    a.levelBase = levelBase; a.levelsRun = levelsRun;
    { static int stop = -1; if (stop < 0) { const char *e = getenv("ORBX_QT_STOP"); stop = e ? atoi(e) : 0; } a.stop = stop; }
    // LDS is sized for the levels of THIS launch (list capacity = largest quota + slack)
    int maxRoots = 1, maxTiles = 1;
    maxQuota = 0;
    for (int l = levelBase; l < levelBase + levelsRun; ++l) {
        const OrbxLevelGeom &g = geomsHost[l];
        maxQuota = std::max(maxQuota, g.quota);
        const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
        maxRoots = std::max(maxRoots, (int)roundf((float)W / (float)H));
        maxTiles = std::max(maxTiles, g.nTiles);
    }
    a.listCap = (std::max(maxQuota + 16, 4 * maxRoots + 8) + 3) & ~3;
    a.pool = (2 * maxQuota + 64 + 4 * maxRoots + 3) & ~3;
    {   // test hook: ORBX_QT_POOL_SLACK=k shrinks the pool to quota + k so that the tombstone compaction path runs
        static int slack = -2;
        if (slack == -2) { const char *e = getenv("ORBX_QT_POOL_SLACK"); slack = e ? atoi(e) : -1; }
        if (slack >= 8) a.pool = (maxQuota + slack + 3) & ~3;
    }
    if (a.pool > 32767 || a.listCap > 32767) return hipErrorInvalidValue;  // node ids / list positions are 15-bit
    static int lk = -1;
    if (lk < 0) { const char *e = getenv("ORBX_QT_LDS_KEYS"); lk = e ? atoi(e) : QT_LDS_KEYS; }
    a.ldsKeys = lk;  // measured: 0 (13 KB per instance, 12 per CU) beats 2048 (21 KB)
    { static int v = -1; if (v < 0) { const char *e = getenv("ORBX_QT_KERNEL"); v = e ? atoi(e) : 3; } a.variant = v; }
    static int mk = -1, mk2 = -1;
    if (mk < 0) { const char *e = getenv("ORBX_QT_MAX_KEYS"); mk = e ? std::min(std::max(atoi(e), 0), QT3_MAX_KEYS) : QT3_MAX_KEYS; }
    if (mk2 < 0) { const char *e = getenv("ORBX_QT_MAX_KEYS2"); mk2 = e ? std::min(std::max(atoi(e), 0), QT3_MAX_KEYS2) : QT3_MAX_KEYS2; }
    a.maxKeys = mk; a.pass2 = 0;
    static bool attrSetDev[64] = {};  // the attribute is per device
    int devId = 0;
    (void)hipGetDevice(&devId);
    bool &attrSet = attrSetDev[devId & 63];
    if (!attrSet) {
        (void)hipFuncSetAttribute((const void *)k_quadtree<QT3_KCH_LARGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_quadtree<QT3_KCH_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void *)k_quadtree_serial, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attrSet = true;
    }
    const size_t tileOffBytes = (((size_t)maxTiles + 1) * 4 + 15) & ~(size_t)15;  // both kernels stage the tile prefix in the node area first
    const dim3 grid(xcd_grid(nframes, levelsRun));
    const size_t ldsMax = 160 * 1024;
    a.mwKeys = 0;
    // few instances (a single frame, a batch of 1080p pairs): a workgroup of QT4_NW waves per instance (the sweeps of a pass on all
    // waves), one launch whatever the candidate count.  Many instances (the 1024-frame batches): one wave per instance - the chip is
    // filled by instances, and a wave's registers / LDS go further that way.  ORBX_QT_MW=0 / 1 forces the choice.
    static const int mwEnv = [] { const char *e = getenv("ORBX_QT_MW"); return e ? atoi(e) : -1; }();
    const bool multiWave = mwEnv >= 0 ? mwEnv != 0 : (long long)nframes * levelsRun <= 1024;
    if (a.variant == 3 && (which & ORBX_QT_LEVEL_SYNC) && multiWave) {
        long long maxArea = 0;
        for (int l = levelBase; l < levelBase + levelsRun; ++l) maxArea = std::max(maxArea, (long long)(geomsHost[l].w - 32) * (geomsHost[l].h - 32));
        const bool large = mk2 > mk && (maxArea / 40 > mk || mk != QT3_MAX_KEYS);
        a.maxKeys = large ? mk2 : mk;
        a.mwKeys = large ? QT3_MAX_KEYS2 : QT3_MAX_KEYS;
        a.nodeBytes = (int)std::max(qt4NodeBytes(a.listCap, a.mwKeys), tileOffBytes);
        static const int mwLdsKeys = [] { const char *e = getenv("ORBX_QT_MW_LDS_KEYS"); return e ? atoi(e) : QT4_NW * QT4_KCH * WAVE; }();
        a.ldsKeys = (long long)nframes * levelsRun <= 128 ? std::min(mwLdsKeys, QT4_NW * QT4_KCH * WAVE) : 0;
        if ((size_t)a.nodeBytes + (size_t)a.ldsKeys * sizeof(uint32_t) > ldsMax) a.ldsKeys = 0;
        size_t lds = (size_t)a.nodeBytes + (size_t)a.ldsKeys * sizeof(uint32_t);
        a.nodeScratch = nullptr;
        if (lds > ldsMax) {
            if (!nodeScratch || nodeScratchBytes < (size_t)a.nodeBytes * nframes * levelsRun) return hipErrorInvalidValue;
            a.nodeScratch = nodeScratch;
            lds = 0;
        }
        static bool mwAttr[64] = {};
        if (!mwAttr[devId & 63]) {
            (void)hipFuncSetAttribute((const void *)k_quadtree_mw, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            mwAttr[devId & 63] = true;
        }
        hipLaunchKernelGGL(k_quadtree_mw, dim3((unsigned)nframes * (unsigned)levelsRun), dim3(QT4_NT), lds, st, a);
    } else if (a.variant == 3 && (which & ORBX_QT_LEVEL_SYNC)) {
        // a level of the size of a VGA image never holds more than a few thousand candidates on camera images; the second launch
        // (6 KB bitmap) is only issued where a level is large enough for > 8192 (2.5 % of the FAST region) - or in the tests
        long long maxArea = 0;
        for (int l = levelBase; l < levelBase + levelsRun; ++l) maxArea = std::max(maxArea, (long long)(geomsHost[l].w - 32) * (geomsHost[l].h - 32));
        const int passes = (mk2 > mk && (maxArea / 40 > mk || mk != QT3_MAX_KEYS)) ? 2 : 1;
        for (int pass = 0; pass < passes; ++pass) {
            a.pass2 = pass;
            a.maxKeys = pass ? mk2 : mk;
            a.nodeBytes = (int)std::max(qt3NodeBytes(a.listCap, pass ? QT3_MAX_KEYS2 : QT3_MAX_KEYS), tileOffBytes);
            size_t lds = (size_t)a.nodeBytes + (size_t)a.ldsKeys * sizeof(uint32_t);
            a.nodeScratch = nullptr;
            if (lds > ldsMax) {  // quota too large for LDS: node arrays in the caller's scratch
                if (!nodeScratch || nodeScratchBytes < (size_t)a.nodeBytes * nframes * levelsRun) return hipErrorInvalidValue;
                a.nodeScratch = nodeScratch;
                lds = 0;
            }
            // fewer keys in registers = more instances per CU: for launches whose lists are short (ORBX_QT_KCH = 32 / 8 forces one)
            static const int kchEnv = [] { const char *e = getenv("ORBX_QT_KCH"); return e ? atoi(e) : 0; }();
            const bool smallKch = kchEnv ? kchEnv <= QT3_KCH_SMALL : maxQuota <= 300;
            if (getenv("ORBX_DEBUG")) {
                int nb = 0;
                if (smallKch) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_quadtree<QT3_KCH_SMALL>, 64, lds);
                else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_quadtree<QT3_KCH_LARGE>, 64, lds);
                fprintf(stderr, "[orbx] k_quadtree<%d> levels %d..%d pass %d: list %d, LDS %zu B, %d instances/CU\n", smallKch ? QT3_KCH_SMALL : QT3_KCH_LARGE,
                        levelBase, levelBase + levelsRun - 1, pass, a.listCap, lds, nb);
            }
            if (smallKch) hipLaunchKernelGGL(k_quadtree<QT3_KCH_SMALL>, grid, dim3(64), lds, st, a);
            else hipLaunchKernelGGL(k_quadtree<QT3_KCH_LARGE>, grid, dim3(64), lds, st, a);
        }
        a.pass2 = 0;
    }
    if ((which & ORBX_QT_SERIAL) || a.variant != 3) {   // candidate sets beyond QT3_MAX_KEYS2 (deferred above), or everything in the A/B variant
        // (its own LDS key buffer: the multi-wave launch above may have set 16 KB of keys, which orbxQuadtreeScratchBytes does not count -
        // with a quota of ~2700 per level that pushed this launch beyond 160 KB with no node scratch allocated: "invalid argument" on a
        // batch of 17 .. 128 instances, found by the batch sweep of tests/tools/stress_parity.py)
        a.ldsKeys = lk;
        a.nodeBytes = (int)std::max(qtNodeBytes(a.pool, a.listCap), tileOffBytes);
        size_t lds = (size_t)a.nodeBytes + (size_t)a.ldsKeys * sizeof(uint32_t);
        a.nodeScratch = nullptr;
        if (lds > ldsMax) {
            if (!nodeScratch || nodeScratchBytes < (size_t)a.nodeBytes * nframes * levelsRun) return hipErrorInvalidValue;
            a.nodeScratch = nodeScratch;
            lds = 0;
        }
        hipLaunchKernelGGL(k_quadtree_serial, grid, dim3(64), lds, st, a);
    }
    return hipGetLastError();
}

// =====================================================================================================
// Pack: border / octave / size (:1004-1017), pt *= scale and the mono / lapping split (:1283-1303)
// =====================================================================================================
struct OrbxKp { float x, y, size, angle, response; int octave, class_id; };

__global__ __launch_bounds__(256) void k_pack(OrbxTables tb, const uint32_t *sel, int selFrameStride, const int *selCount,
                                              int lap0, int lap1, OrbxKp *kps, int cap, uint32_t *workKey, int *workPos,
                                              int *nOut, int *monoOut)
{
    __shared__ int lvOff[ORBX_MAX_LEVELS + 1];
    __shared__ int waveTot[4];
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        int off = 0;
        for (int l = 0; l < tb.nlevels; ++l) { lvOff[l] = off; off += selCount[(size_t)f * tb.nlevels + l]; }
        lvOff[tb.nlevels] = off;
    }
    __syncthreads();
    const int n = min(lvOff[tb.nlevels], cap);
    int running = 0;
    const unsigned long long lt = lanemask_lt();
    for (int base = 0; base < n; base += 256) {
        const int gi = base + tid;
        const bool valid = gi < n;
        int l = 0;
        uint32_t key = 0;
        float fx = 0.f, fy = 0.f;
        bool lapping = false;
        int xl = 0, yl = 0;
        if (valid) {
            while (gi >= lvOff[l + 1]) ++l;
            key = sel[(size_t)f * selFrameStride + tb.selBase[l] + (gi - lvOff[l])];
            xl = (int)(key & 0xFFF) + ORBX_BORDER;
            yl = (int)((key >> 12) & 0xFFF) + ORBX_BORDER;
            fx = (float)xl; fy = (float)yl;
            if (l != 0) { fx = __fmul_rn(fx, tb.scale[l]); fy = __fmul_rn(fy, tb.scale[l]); }
            lapping = fx >= (float)lap0 && fx <= (float)lap1;
        }
        const bool isMono = valid && !lapping;
        const unsigned long long m = __ballot(isMono);
        if (lane == 0) waveTot[wave] = __popcll(m);
        __syncthreads();
        int before = running + __popcll(m & lt);
        for (int w2 = 0; w2 < wave; ++w2) before += waveTot[w2];
        const int blockTot = waveTot[0] + waveTot[1] + waveTot[2] + waveTot[3];
        if (valid) {
            const int pos = lapping ? (n - 1 - (gi - before)) : before;
            OrbxKp kp;
            kp.x = fx; kp.y = fy; kp.size = (float)tb.patchSize[l]; kp.angle = -1.f;
            kp.response = (float)(key >> 24); kp.octave = l; kp.class_id = -1;
            kps[(size_t)f * cap + pos] = kp;
            // work order of k_orient_desc = list order (a raster order inside each level was measured: no gain)
            workKey[(size_t)f * cap + gi] = (uint32_t)xl | ((uint32_t)yl << 12) | ((uint32_t)l << 24);
            workPos[(size_t)f * cap + gi] = pos;
        }
        running += blockTot;
        __syncthreads();
    }
    if (tid == 0) { nOut[f] = n; monoOut[f] = running; }
}

hipError_t orbxLaunchPack(hipStream_t st, int nframes, OrbxTables tabs, const uint32_t *sel, int selFrameStride,
                          const int *selCount, int lap0, int lap1, void *kps, int cap, uint32_t *workKey,
                          int *workPos, int *nOut, int *monoOut)
{
    hipLaunchKernelGGL(k_pack, dim3(nframes), dim3(256), 0, st, tabs, sel, selFrameStride, selCount, lap0, lap1,
                       (OrbxKp *)kps, cap, workKey, workPos, nOut, monoOut);
    return hipGetLastError();
}

// =====================================================================================================
// Orientation + descriptor: one wave per keypoint
// =====================================================================================================
__constant__ signed char c_pattern[1024];

// cv::fastAtan2 (degrees), SURVEY App. A.4 — every operation individually rounded.
__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = __fdiv_rn(ay, __fadd_rn(ax, eps));
        c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c);
    } else {
        c = __fdiv_rn(ax, __fadd_rn(ay, eps));
        c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c));
    }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}

#ifndef OD_KPW
#define OD_KPW 8   // keypoints per wave: small single-wave workgroups keep fewer frames in flight per XCD (L2 reuse of the
                   // patches); measured 0.76 ms (8 x 1 wave) vs 0.83 ms (32 x 4 waves) per batch
#endif
#ifndef OD_KPB
#define OD_KPB 8   // keypoints whose patch loads are in flight together in the moment pass
#endif
#ifndef OD_PF
#define OD_PF 3    // descriptor pass: keypoints whose patch loads are in flight ahead of the one being sampled
#endif
#ifndef OD_WPB
#define OD_WPB 1   // waves per workgroup (the waves are independent)
#endif
#define OD_PROWS 48    // blurred patch rows staged per keypoint: up to 6 tile rows of 8 (sample reach is +-18)
#define OD_PPITCH 64   // up to 4 tiles of 16 columns: covers [x - 18 - 15, x + 18] for any alignment

// Waves of one workgroup use disjoint LDS slices here, so ordering only has to hold inside a wave: wait for this
// wave's own LDS traffic and stop the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void __syncthreads_lds_wave()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum over the 64 lanes: 4 DPP steps give every lane its 16-lane row total, the 4 row totals are added as scalars
__device__ __forceinline__ int wave_sum_i32(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);  // row_half_mirror
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);  // row_mirror
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
}

// One wave = OD_KPW keypoints of one frame:
//   pass 1  per keypoint, wave-cooperative: IC_Angle moments over the radius-15 patch (16 byte loads in flight per
//           lane, DPP row reduction); lane k keeps the moments of keypoint k
//   pass 2  per lane: fastAtan2 + (float)cos/sin((double)angle) for all keypoints of the wave at once
//   pass 3  per keypoint, wave-cooperative: 256 steered-BRIEF tests, 64 per ballot
#ifdef OD_TIMING
// debug build (-DOD_TIMING): phase time stamps (s_memtime) of the first 8192 workgroups, read back with orbx_debug_od_timing
// (tools/od_timing.py)
__device__ unsigned long long g_odTiming[8192][8];
extern "C" int orbx_debug_od_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_odTiming), sizeof(g_odTiming)); }
#define OD_STAMP(i) do { if (blockIdx.x < 8192 && threadIdx.x == 0) g_odTiming[blockIdx.x][i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define OD_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(64 * OD_WPB) void k_orient_desc(OrbxTables tb, OrbxLevelPtrs lv, const uint32_t *workKey,
                                                     const int *workPos, const int *nOut, OrbxKp *kps, uint8_t *desc,
                                                     int cap, int nframes, int chunks, unsigned xcdM, int xcdL, unsigned xcdPer)
{
    int f, chunk;
    OD_STAMP(0);
    (void)chunks;
    if (!xcd_decode_magic(nframes, xcdM, xcdL, xcdPer, f, chunk)) return;   // (xcd_magic(chunks): no integer division in front of a wave's first load)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = nOut[f];
    const int base = (chunk * OD_WPB + wave) * OD_KPW;
    if (base >= n) return;
    const int cnt = min(OD_KPW, n - base);
    uint32_t myKey = 0;
    int myPos = 0;
    if (lane < cnt) {
        myKey = workKey[(size_t)f * cap + base + lane];
        myPos = workPos[(size_t)f * cap + base + lane];
    }
    // this lane's 4 test pairs (pairs lane, lane + 64, lane + 128, lane + 192) as floats
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int v = *(const int *)(c_pattern + (r * 64 + lane) * 4);
        px0[r] = (float)(signed char)(v & 0xFF);
        py0[r] = (float)(signed char)((v >> 8) & 0xFF);
        px1[r] = (float)(signed char)((v >> 16) & 0xFF);
        py1[r] = (float)(signed char)((v >> 24) & 0xFF);
    }

    // ---- pass 1: moments ------------------------------------------------------------------------------------
    // Lane k first resolves the level tables of keypoint k (raw / blurred patch pointers, pitches), so the
    // per-keypoint loops below only broadcast registers (v_readlane) and never wait on a dependent table load.
    unsigned long long myRaw = 0, myBlur = 0;
    int myRp = 0, myBp = 0;
    if (lane < cnt) {
        const int l = myKey >> 24, xl = myKey & 0xFFF, yl = (myKey >> 12) & 0xFFF;
        myRp = lv.rawPitch[l];
        myBp = lv.blurPitch[l];
        myRaw = (unsigned long long)(lv.raw[l] + (size_t)f * lv.rawFrameStride[l] + (size_t)yl * myRp + xl);
        // first 16 x 8 tile of the staged blurred patch (tiled layout, orbx_blur_tiled_offset; myBp = bytes per tile row)
        myBlur = (unsigned long long)(lv.blur[l] + (size_t)f * lv.blurFrameStride[l] + (size_t)((yl - 18) >> 3) * myBp + ((xl - 18) >> 4) * 128);
    }
    // (A tiled copy of the raw levels for this pass - 14 line fills per patch instead of 38 - was measured and dropped: the
    // per-dword circle masks it needs cost more VALU than the fills it saves, 0.67 vs 0.60 ms, and the copy slows the blur.)
    // The 31 x 31 patch is read as dwords: lane = (row r = lane >> 3, dword j = lane & 7) covers row v = 8 it + r - 15 (it = 0..3)
    // and the 4 columns u = 4j - 15 .. 4j - 12, loaded from the exact (byte-unaligned) address - four loads per lane and
    // keypoint instead of 16 byte loads (the texture addresser takes a wave's 64 addresses at the same rate for either).
    // Samples outside the circular patch (|u| > umax[|v|], or the 32nd row / column, which are still inside the image:
    // keypoints keep a 19-px margin) get weight 0, so the loads are unconditional and the loop is branch-free.
    //   m10 = sum u p = sum (u + 16) p - 16 sum p,   m01 = sum v p      with two v_dot4_u32_u8 per dword
    const int pj = lane & 7, pr = lane >> 3;
    uint32_t w1[4], w0[4];
    int vv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int v = 8 * it + pr - 15;
        const int av = v < 0 ? -v : v;
        const int um = (int)((tb.umaxPacked >> (4 * (av & 15))) & 15);
        w1[it] = 0; w0[it] = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int u = 4 * pj - 15 + b;
            const bool in = v <= 15 && (u < 0 ? -u : u) <= um;
            w1[it] |= in ? (uint32_t)(u + 16) << (8 * b) : 0u;
            w0[it] |= in ? 1u << (8 * b) : 0u;
        }
        vv[it] = v;
    }
    typedef const __attribute__((address_space(1))) uint8_t *gptr_t;
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    typedef const __attribute__((address_space(1))) u32_unaligned *gdw_t;
    int myM01 = 0, myM10 = 0;
    OD_STAMP(1);
    // OD_KPB keypoints per step: 4 * OD_KPB independent dword loads in flight per lane before the first reduction (the pass is
    // bound by the latency of these loads)
    for (int k0 = 0; k0 < cnt; k0 += OD_KPB) {
        uint32_t P[OD_KPB][4];
#pragma unroll
        for (int q = 0; q < OD_KPB; ++q) {
            const int k = min(k0 + q, cnt - 1);
            const int rp = __builtin_amdgcn_readlane(myRp, k);
            // the patch's first dword (row -15, column -15) as a wave-uniform base, the lane's dwords at NON-NEGATIVE 32-bit offsets from it:
            // a scalar base + a 32-bit vector offset is one addressing mode of the load (per lane 64-bit sums were five quarter-rate
            // v_lshl_add_u64 and a v_mul_lo_u32 per keypoint)
            const gptr_t c = (gptr_t)((((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(myRaw >> 32), k) << 32) |
                                       (unsigned)__builtin_amdgcn_readlane((int)myRaw, k)) - (unsigned long long)(15 * rp + 15));
            const uint32_t off0 = __umul24((uint32_t)pr, (uint32_t)rp) + 4u * (uint32_t)pj;
#pragma unroll
            for (int it = 0; it < 4; ++it) P[q][it] = *(gdw_t)(c + (off0 + (uint32_t)(8 * it) * (uint32_t)rp));
        }
#pragma unroll
        for (int q = 0; q < OD_KPB; ++q) {
            uint32_t s1 = 0, s0 = 0;
            int m01 = 0;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                s1 = __builtin_amdgcn_udot4(P[q][it], w1[it], s1, false);
                const uint32_t t = __builtin_amdgcn_udot4(P[q][it], w0[it], 0u, false);
                s0 += t;
                m01 += vv[it] * (int)t;
            }
            const int m10 = wave_sum_i32((int)s1 - 16 * (int)s0);
            m01 = wave_sum_i32(m01);
            if (k0 + q < cnt && lane == k0 + q) { myM10 = m10; myM01 = m01; }
        }
    }

    OD_STAMP(2);
    // ---- pass 2: orientation + rotation, one keypoint per lane ---------------------------------------------------
    const float angle = fast_atan2_deg((float)myM01, (float)myM10);
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = __fmul_rn(angle, factorPI);
    float myA, myB;
    if (tb.trigMode == 1) { myA = orbx_glibc_sincosf(ar, 1); myB = orbx_glibc_sincosf(ar, 0); }   // cosf / sinf (float overloads)
    else {  // ::cos(double) / ::sin(double), narrowed - one argument reduction for both (the same values as two calls, 100 instructions less)
        double sd, cd;
        sincos((double)ar, &sd, &cd);
        myA = (float)cd; myB = (float)sd;
    }
    if (lane < cnt) kps[(size_t)f * cap + myPos].angle = angle;

    OD_STAMP(3);
    // ---- pass 3: descriptors --------------------------------------------------------------------------------
    // The 512 sample points of a keypoint lie within +-18 px of it.  Instead of 8 wave-wide 64-way divergent byte
    // gathers from global memory (texture-address bound), the tiles that cover the 37 x 37 blurred patch (at most 4 x 6
    // tiles of 128 B, one 16-byte tile row per lane = 3 wave instructions) are staged row-major into this wave's LDS slice
    // and the samples are gathered from LDS.
    __shared__ __attribute__((aligned(16))) uint8_t patchLds[OD_WPB][OD_PROWS * OD_PPITCH];
    uint8_t *patch = patchLds[wave];
    // A wave spends most of this pass waiting for the staging loads (~5000 cycles under load against ~1000 cycles of work
    // per keypoint, tools/od_timing.py): the loads of the next OD_PF keypoints are in flight (a ring of 3-register slots,
    // indexed statically because the keypoint loop is fully unrolled) while the samples of keypoint k are consumed.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 nxt[OD_PF][3];
    auto tilesOf = [&](int kk, int &ntx, int &nty) {
        const uint32_t key = __builtin_amdgcn_readlane(myKey, kk);
        const int xl = key & 0xFFF, yl = (key >> 12) & 0xFFF;
        ntx = ((xl + 18) >> 4) - ((xl - 18) >> 4) + 1;
        nty = ((yl + 18) >> 3) - ((yl - 18) >> 3) + 1;
    };
    auto issue = [&](int kk, u32x4 (&dst)[3]) {
        typedef const __attribute__((address_space(1))) uint8_t *gptr_t;
        const gptr_t src = (gptr_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(myBlur >> 32), kk) << 32) |
                                    (unsigned)__builtin_amdgcn_readlane((int)myBlur, kk));
        const int bp = __builtin_amdgcn_readlane(myBp, kk);
        int ntx, nty;
        tilesOf(kk, ntx, nty);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = lane + 64 * i, r = idx >> 2, tc = idx & 3;  // patch row r = 8 * tile row + row in tile, tile column tc
            if (tc < ntx && (r >> 3) < nty)
                dst[i] = *(const __attribute__((address_space(1))) u32x4 *)(src + (__umul24((uint32_t)(r >> 3), (uint32_t)bp) + (uint32_t)(tc * 128 + (r & 7) * 16)));
        }
    };
#pragma unroll
    for (int k = 0; k < OD_PF; ++k)
        if (k < cnt) issue(k, nxt[k]);
#pragma unroll
    for (int k = 0; k < OD_KPW; ++k) {
        if (k >= cnt) break;  // wave-uniform
        const uint32_t key = __builtin_amdgcn_readlane(myKey, k);
        const int pos = __builtin_amdgcn_readlane(myPos, k);
        const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myA), k));
        const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myB), k));
        const int xl = key & 0xFFF, yl = (key >> 12) & 0xFFF;
        int ntx, nty;
        tilesOf(k, ntx, nty);
        __syncthreads_lds_wave();
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = lane + 64 * i, r = idx >> 2, tc = idx & 3;
            if (tc < ntx && (r >> 3) < nty) *(u32x4 *)(patch + r * OD_PPITCH + tc * 16) = nxt[k % OD_PF][i];
        }
        if (k + OD_PF < cnt) issue(k + OD_PF, nxt[k % OD_PF]);
        __syncthreads_lds_wave();
        const uint8_t *bc = patch + (yl - (((yl - 18) >> 3) << 3)) * OD_PPITCH + (xl - (((xl - 18) >> 4) << 4));
        // cvRound(x) = round to nearest even = what adding 1.5 * 2^23 does to a float below 2^22: the sum's bit pattern is 0x4B400000 + round(x)
        // (two's complement in the mantissa, either sign) - one full-rate v_add_f32 where v_rndne_f32 + v_cvt_i32_f32 are two slow-class
        // instructions, 16 times per keypoint.  The LDS address (iy << 6) + ix + bc then is (Y << 6) + X + bc - 65 * 0x4B400000 modulo 2^32.
        const float RM = 12582912.0f;
        const uint32_t bcm = (uint32_t)(size_t)(__attribute__((address_space(3))) const uint8_t *)bc - 65u * 0x4B400000u;
        int t0[4], t1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t Y0 = (uint32_t)__float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px0[r], b), __fmul_rn(py0[r], a)), RM));
            const uint32_t X0 = (uint32_t)__float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px0[r], a), __fmul_rn(py0[r], b)), RM));
            const uint32_t Y1 = (uint32_t)__float_as_int(__fadd_rn(__fadd_rn(__fmul_rn(px1[r], b), __fmul_rn(py1[r], a)), RM));
            const uint32_t X1 = (uint32_t)__float_as_int(__fadd_rn(__fsub_rn(__fmul_rn(px1[r], a), __fmul_rn(py1[r], b)), RM));
            static_assert(OD_PPITCH == 64, "sample address: row << 6");
            t0[r] = *(const __attribute__((address_space(3))) uint8_t *)(size_t)((Y0 << 6) + X0 + bcm);
            t1[r] = *(const __attribute__((address_space(3))) uint8_t *)(size_t)((Y1 << 6) + X1 + bcm);
        }
        unsigned long long word = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned long long bits = __ballot(t0[r] < t1[r]);
            if (lane == r) word = bits;
        }
        if (lane < 4) *(unsigned long long *)(desc + ((size_t)f * cap + pos) * 32 + lane * 8) = word;
    }
    OD_STAMP(4);
}

// One-time, per-device state that needs synchronous calls (the rBRIEF pattern in constant memory, the dynamic-LDS limits): done at
// orbx_create, never on the extraction path - a stream capture must not meet them.
hipError_t orbxInitDevice()
{
    static bool done[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || done[dev]) return hipSuccess;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), kOrbxPattern31, 1024);
    if (e != hipSuccess) return e;
    (void)hipFuncSetAttribute((const void *)k_quadtree<QT3_KCH_LARGE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_quadtree<QT3_KCH_SMALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void *)k_quadtree_serial, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    done[dev] = true;
    return hipSuccess;
}

hipError_t orbxLaunchOrientDesc(hipStream_t st, int nframes, OrbxTables tabs, OrbxLevelPtrs lv, const uint32_t *workKey,
                                const int *workPos, const int *nOut, void *kps, uint8_t *desc, int cap)
{
    {
        const hipError_t e = orbxInitDevice();
        if (e != hipSuccess) return e;
    }
    const int chunks = (cap + OD_WPB * OD_KPW - 1) / (OD_WPB * OD_KPW);
    const XcdMagic xm = xcd_magic(chunks);
    hipLaunchKernelGGL(k_orient_desc, dim3(xcd_grid(nframes, chunks)), dim3(64 * OD_WPB), 0, st, tabs, lv, workKey, workPos, nOut, (OrbxKp *)kps, desc, cap,
                       nframes, chunks, xm.m, xm.l, xm.per);
    return hipGetLastError();
}
