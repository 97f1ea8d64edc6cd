// orbx_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the ORB front-end.
//
// Pipeline per batch of frames (all levels, all frames per launch; everything stays in HBM):
//   k_pyr_fast   (x nlevels)  fused: level-l tile -> LDS; bilinear level l+1 (cv::resize fixed point);
//                             FAST-9/16 score at minThFAST; cell-local 3x3 NMS; per-cell iniTh/minTh
//                             fallback; ordered (cell-row-major, raster) compaction with wave ballots.
//                             Replaces ComputePyramid (reference src/ORBextractor.cc:1309-1329) +
//                             tileCalcKeypoints (:867-950) [and the fork's tileCalcKeypoints_kernel].
//   k_quadtree                one wave per (frame, level): DistributeOctTree (:584-774) with exact list
//                             order and libstdc++ std::sort tie order.
//   k_blur       (x nlevels)  cv::GaussianBlur 7x7 sigma 2, fixed point (:1270-1273).
//   k_pack                    border/octave/size (:1004-1017), pt *= scale, mono/stereo split (:1283-1303).
//   k_orient_desc             one wave per keypoint: IC_Angle moments + fastAtan2 (Angle.cl:24-60) and the
//                             256-bit steered BRIEF (computeOrbDescriptor, :105-149), 64 bits per ballot.
//
// Integer / bitwise work throughout: no MFMA.  Floating point only where the reference has it, evaluated
// with explicitly rounded, un-fused operations (__fmul_rn/__fadd_rn/__fdiv_rn; file is also built with
// -ffp-contract=off) so results are bit-identical to the x86 CPU path.
#include "orbx_internal.h"
#include "libstdcxx_sort.h"
#include "orbx_pattern.inc"

#define WAVE 64

__device__ __forceinline__ unsigned long long lanemask_lt()
{
    const unsigned lane = threadIdx.x & 63;
    return lane == 0 ? 0ull : (~0ull >> (64 - lane));
}

// =====================================================================================================
// FAST-9/16 (cv::FAST TYPE_9_16)
// =====================================================================================================

// true iff the 16-bit circular mask has >= 9 contiguous set bits
__device__ __forceinline__ bool ring_has9(unsigned m)
{
    m |= m << 16;
    unsigned r = m & (m >> 1);
    r &= r >> 2;
    r &= r >> 4;
    r &= m >> 8;
    return r != 0;
}

// Returns the FAST corner score (cornerScore<16>: largest threshold for which the pixel is still a corner,
// = max over the 16 arcs of 9 of min |diff| on one polarity, minus 1) if the pixel is a corner at
// threshold `th`, else 0.  `c` points at the centre pixel inside an LDS tile with row pitch `tp`.
__device__ __forceinline__ int fast_corner_score(const uint8_t *c, int tp, int th)
{
    const int v = c[0];
    int p[16];
    p[0] = c[3 * tp];       p[1] = c[3 * tp + 1];   p[2] = c[2 * tp + 2];   p[3] = c[tp + 3];
    p[4] = c[3];            p[5] = c[-tp + 3];      p[6] = c[-2 * tp + 2];  p[7] = c[-3 * tp + 1];
    p[8] = c[-3 * tp];      p[9] = c[-3 * tp - 1];  p[10] = c[-2 * tp - 2]; p[11] = c[-tp - 3];
    p[12] = c[-3];          p[13] = c[tp - 3];      p[14] = c[2 * tp - 2];  p[15] = c[3 * tp - 1];
    const int lo = v - th, hi = v + th;
    unsigned dark = 0, bright = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        dark |= (unsigned)(p[k] < lo) << k;
        bright |= (unsigned)(p[k] > hi) << k;
    }
    if (!(ring_has9(dark) || ring_has9(bright))) return 0;

    int d[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) d[k] = v - p[k];
    int mn3[16], mx3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        mn3[k] = min(d[k], min(d[(k + 1) & 15], d[(k + 2) & 15]));
        mx3[k] = max(d[k], max(d[(k + 1) & 15], d[(k + 2) & 15]));
    }
    int a = -256, b = 256;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int m9 = min(mn3[k], min(mn3[(k + 3) & 15], mn3[(k + 6) & 15]));
        int M9 = max(mx3[k], max(mx3[(k + 3) & 15], mx3[(k + 6) & 15]));
        a = max(a, m9);
        b = min(b, M9);
    }
    return max(a, -b) - 1;
}

struct PyrFastArgs {
    OrbxLevelGeom g;
    const uint8_t *src; int srcPitch; long long srcFrameStride; int srcAligned4;
    uint8_t *dst; int dw, dh, dstPitch; long long dstFrameStride;
    OrbxResizeTabs tabs;
    uint32_t *cand; long long candFrameStride;
    int *tileCount; int tilesPerFrame;
    int iniTh, minTh;
};

#define PF_MAX_GROUP_CELLS 32

__global__ __launch_bounds__(256) void k_pyr_fast(PyrFastArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const OrbxLevelGeom &g = a.g;
    const int grp = blockIdx.x, crow = blockIdx.y, f = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // interior (corner-capable) pixel ranges of this tile
    const int ty0 = ORBX_FAST_MARGIN + crow * g.hCell;
    const int ty1 = min(ty0 + g.hCell, g.h - ORBX_FAST_MARGIN);
    const int c0 = grp * g.groupCells, c1 = min(c0 + g.groupCells, g.ncx);
    const int tx0 = ORBX_FAST_MARGIN + c0 * g.wCell;
    const int tx1 = min(ORBX_FAST_MARGIN + c1 * g.wCell, g.w - ORBX_FAST_MARGIN);
    const int THi = ty1 - ty0, TWi = tx1 - tx0;
    // rows / columns staged in LDS: interior + 3-px ring halo; edge tiles extend to the image border so the
    // tile also owns the source pixels of the next pyramid level there
    const int ly0 = crow == 0 ? 0 : ty0 - 3, ly1 = crow == g.ncy - 1 ? g.h : ty1 + 3;
    const int lx0 = grp == 0 ? 0 : tx0 - 3, lx1 = grp == g.nGroups - 1 ? g.w : tx1 + 3;
    const int ax0 = lx0 & ~3;
    const int tp = g.tilePitch, sp = g.scorePitch;

    uint8_t *tile = lds;
    uint8_t *score = lds + (size_t)g.tileRows * tp;                      // (hCell + 2) x sp
    unsigned long long *maxMask = (unsigned long long *)(score + (size_t)(g.hCell + 2) * sp);  // hCell x maskW
    const int maskW = (g.groupCells * g.wCell + 63) >> 6;
    uint8_t *colInfo = (uint8_t *)(maxMask + (size_t)g.hCell * maskW);   // per interior column: cell | edge flags
    __shared__ int cntAll[PF_MAX_GROUP_CELLS], cntIni[PF_MAX_GROUP_CELLS], cellOff[PF_MAX_GROUP_CELLS + 1];

    // ---- stage the tile: coalesced dword loads of row-major image rows into LDS ------------------------
    const uint8_t *srcF = a.src + (size_t)f * a.srcFrameStride;
    const int nRowsL = ly1 - ly0;
    if (a.srcAligned4) {
        const int nDw = (lx1 - ax0 + 3) >> 2;
        for (int r = wave; r < nRowsL; r += 4) {
            const uint32_t *srow = (const uint32_t *)(srcF + (size_t)(ly0 + r) * a.srcPitch + ax0);
            uint32_t *trow = (uint32_t *)(tile + r * tp);
            for (int c = lane; c < nDw; c += 64) trow[c] = srow[c];
        }
    } else {
        const int nB = lx1 - ax0;
        for (int r = wave; r < nRowsL; r += 4) {
            const uint8_t *srow = srcF + (size_t)(ly0 + r) * a.srcPitch + ax0;
            uint8_t *trow = tile + r * tp;
            for (int c = lane; c < nB; c += 64) trow[c] = srow[c];
        }
    }
    if (tid < PF_MAX_GROUP_CELLS) { cntAll[tid] = 0; cntIni[tid] = 0; }
    // per-column cell index + cell-edge flags (bit7: first column of its cell, bit6: last column)
    for (int x = tid; x < TWi; x += 256) {
        int rel = x;  // relative to tx0, which is a cell boundary
        int cell = rel / g.wCell, cx = rel - cell * g.wCell;
        int cw = min(g.wCell, TWi - cell * g.wCell);
        colInfo[x] = (uint8_t)(cell | (cx == 0 ? 0x80 : 0) | (cx == cw - 1 ? 0x40 : 0));
    }
    // zero ring of the score tile
    for (int i = tid; i < TWi + 2; i += 256) { score[i] = 0; score[(THi + 1) * sp + i] = 0; }
    for (int i = tid; i < THi + 2; i += 256) { score[i * sp] = 0; score[i * sp + TWi + 1] = 0; }
    __syncthreads();

    // ---- next pyramid level: cv::resize INTER_LINEAR fixed point ---------------------
    if (a.dst) {
        const int dy0 = a.tabs.rowBeg[crow], dy1 = a.tabs.rowBeg[crow + 1];
        const int dx0 = a.tabs.colBeg[grp], dx1 = a.tabs.colBeg[grp + 1];
        uint8_t *dstF = a.dst + (size_t)f * a.dstFrameStride;
        for (int dy = dy0 + wave; dy < dy1; dy += 4) {
            const int sy = a.tabs.yofs[dy];
            const int sy0 = min(max(sy, 0), g.h - 1), sy1 = min(max(sy + 1, 0), g.h - 1);
            const int yab = a.tabs.yab[dy];
            const int b0 = (short)(yab & 0xFFFF), b1 = yab >> 16;
            const uint8_t *r0 = tile + (sy0 - ly0) * tp - ax0;
            const uint8_t *r1 = tile + (sy1 - ly0) * tp - ax0;
            uint8_t *drow = dstF + (size_t)dy * a.dstPitch;
            for (int dx = dx0 + lane; dx < dx1; dx += 64) {
                const int sx0 = a.tabs.xofs[dx];
                const int sx1 = min(sx0 + 1, g.w - 1);
                const int xab = a.tabs.xab[dx];
                const int a0 = (short)(xab & 0xFFFF), a1 = xab >> 16;
                const int h0 = r0[sx0] * a0 + r0[sx1] * a1;
                const int h1 = r1[sx0] * a0 + r1[sx1] * a1;
                drow[dx] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }

    // ---- FAST score map at minThFAST into LDS ---------------------------------------------------------
    const int minTh = min(max(a.minTh, 0), 255), iniTh = min(max(a.iniTh, 0), 255);
    const int minEff = max(minTh, 1), iniEff = max(iniTh, 1);
    for (int y = wave; y < THi; y += 4) {
        const uint8_t *trow = tile + (ty0 + y - ly0) * tp + (tx0 - ax0);
        uint8_t *srow = score + (y + 1) * sp + 1;
        for (int x = lane; x < TWi; x += 64) {
            int s = fast_corner_score(trow + x, tp, minTh);
            srow[x] = (uint8_t)(s >= minEff ? s : 0);
        }
    }
    __syncthreads();

    // ---- cell-local 3x3 non-max suppression; per-cell counts at both thresholds -------------------------
    for (int y = wave; y < THi; y += 4) {
        const uint8_t *s0 = score + y * sp + 1, *s1 = s0 + sp, *s2 = s1 + sp;
        for (int xb = 0; xb < TWi; xb += 64) {
            const int x = xb + lane;
            bool ismax = false;
            if (x < TWi) {
                const int s = s1[x];
                if (s > 0) {
                    const int ci = colInfo[x];
                    const bool le = ci & 0x80, re = ci & 0x40;
                    int m = max(s0[x], s2[x]);
                    if (!le) m = max(m, max(s0[x - 1], max(s1[x - 1], s2[x - 1])));
                    if (!re) m = max(m, max(s0[x + 1], max(s1[x + 1], s2[x + 1])));
                    if (s > m) {
                        ismax = true;
                        atomicAdd(&cntAll[ci & 0x3F], 1);
                        if (s >= iniEff) atomicAdd(&cntIni[ci & 0x3F], 1);
                    }
                }
            }
            unsigned long long mk = __ballot(ismax);
            if (lane == 0) maxMask[y * maskW + (xb >> 6)] = mk;
        }
    }
    __syncthreads();
    const int nCells = c1 - c0;
    if (tid == 0) {
        int off = 0;
        for (int c = 0; c < nCells; ++c) {
            cellOff[c] = off;
            off += cntIni[c] > 0 ? cntIni[c] : cntAll[c];
        }
        cellOff[nCells] = off;
        a.tileCount[(size_t)f * a.tilesPerFrame + g.tileBase + crow * g.nGroups + grp] = off;
    }
    __syncthreads();

    // ---- ordered compaction: cells in order, raster inside a cell (== tileCalcKeypoints emission order) --
    uint32_t *slot = a.cand + (size_t)f * a.candFrameStride + g.candBase +
                     (size_t)(crow * g.nGroups + grp) * g.slotCap;
    const unsigned long long lt = lanemask_lt();
    for (int c = wave; c < nCells; c += 4) {
        const int cx0 = c * g.wCell;
        const int cw = min(g.wCell, TWi - cx0);
        const int thr = cntIni[c] > 0 ? iniEff : minEff;
        int off = cellOff[c];
        const int total = cw * THi;
        for (int base = 0; base < total; base += 64) {
            const int p = base + lane;
            bool keep = false;
            int xt = 0, r = 0, s = 0;
            if (p < total) {
                r = p / cw;
                xt = cx0 + (p - r * cw);
                if ((maxMask[r * maskW + (xt >> 6)] >> (xt & 63)) & 1ull) {
                    s = score[(r + 1) * sp + xt + 1];
                    keep = s >= thr;
                }
            }
            const unsigned long long m = __ballot(keep);
            if (keep)
                slot[off + __popcll(m & lt)] =
                    ORBX_PACK(tx0 + xt - ORBX_BORDER, ty0 + r - ORBX_BORDER, s);
            off += __popcll(m);
        }
    }
}

hipError_t orbxLaunchPyrFast(hipStream_t st, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch,
                             long long srcFrameStride, uint8_t *dst, int dw, int dh, int dstPitch,
                             long long dstFrameStride, OrbxResizeTabs tabs, uint32_t *cand, long long candFrameStride,
                             int *tileCount, int tilesPerFrame, int iniTh, int minTh)
{
    PyrFastArgs a;
    a.g = g;
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    a.srcAligned4 = (((uintptr_t)src & 3) == 0 && (srcPitch & 3) == 0 && (srcFrameStride & 3) == 0) ? 1 : 0;
    a.dst = dst; a.dw = dw; a.dh = dh; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.tabs = tabs;
    a.cand = cand; a.candFrameStride = candFrameStride;
    a.tileCount = tileCount; a.tilesPerFrame = tilesPerFrame;
    a.iniTh = iniTh; a.minTh = minTh;
    if (g.groupCells > PF_MAX_GROUP_CELLS) return hipErrorInvalidValue;
    dim3 grid(g.nGroups, g.ncy, nframes);
    hipLaunchKernelGGL(k_pyr_fast, grid, dim3(256), g.ldsBytes, st, a);
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

struct BlurArgs {
    const uint8_t *src; int srcPitch; long long srcFrameStride;
    uint8_t *dst; int dstPitch; long long dstFrameStride;
    int w, h;
    int taps[7];
};

__global__ __launch_bounds__(256) void k_blur(BlurArgs a)
{
    __shared__ uint8_t in[BL_TH + 6][BL_TW + 8];
    __shared__ uint16_t hs[BL_TH + 6][BL_TW];
    const int x0 = blockIdx.x * BL_TW, y0 = blockIdx.y * BL_TH, f = blockIdx.z;
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
            dstF[(size_t)gy * a.dstPitch + gx] = (uint8_t)min(v, 255u);
        }
    }
}

hipError_t orbxLaunchBlur(hipStream_t st, int nframes, int w, int h, const uint8_t *src, int srcPitch,
                          long long srcFrameStride, uint8_t *dst, int dstPitch, long long dstFrameStride,
                          const int taps[7])
{
    BlurArgs a;
    a.src = src; a.srcPitch = srcPitch; a.srcFrameStride = srcFrameStride;
    a.dst = dst; a.dstPitch = dstPitch; a.dstFrameStride = dstFrameStride;
    a.w = w; a.h = h;
    for (int k = 0; k < 7; ++k) a.taps[k] = taps[k];
    dim3 grid((w + BL_TW - 1) / BL_TW, (h + BL_TH - 1) / BL_TH, nframes);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, st, a);
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
#define QT_LDS_KEYS 2048
#define QT_NIL (-1)

struct QtSized { int cnt, ulx, id; };
struct QtSizedLess {
    __device__ __forceinline__ bool operator()(const QtSized &a, const QtSized &b) const
    {
        return a.cnt < b.cnt || (a.cnt == b.cnt && a.ulx < b.ulx);
    }
};

struct QtArgs {
    const OrbxLevelGeom *geoms;  // device array [nlevels]
    const uint32_t *cand; long long candFrameStride;
    const int *tileCount; int tilesPerFrame;
    uint32_t *scratchA, *scratchB;  // [nframes][candFrameStride] spill space for levels with > QT_LDS_KEYS candidates
    uint32_t *sel; int selFrameStride;
    int *selCount;
    int nlevels;
    int pool;  // node pool size P
};

struct QtState {
    int *gx, *gy, *begin, *count;
    short *next, *prev, *noMore, *freeList;
    QtSized *sized;
    uint32_t *A, *B;
    int head, size, freeTop, bump, nSized, nToExpand;
};

__device__ __forceinline__ int qt_alloc(QtState &S)
{
    if (S.freeTop > 0) return S.freeList[--S.freeTop];
    return S.bump++;
}
__device__ __forceinline__ void qt_push_front(QtState &S, int id)
{
    S.prev[id] = QT_NIL;
    S.next[id] = (short)S.head;
    if (S.head != QT_NIL) S.prev[S.head] = (short)id;
    S.head = id;
    ++S.size;
}
__device__ __forceinline__ void qt_erase(QtState &S, int id)
{
    const int p = S.prev[id], n = S.next[id];
    if (p != QT_NIL) S.next[p] = (short)n; else S.head = n;
    if (n != QT_NIL) S.prev[n] = (short)p;
    S.freeList[S.freeTop++] = (short)id;
    --S.size;
}

// ExtractorNode::DivideNode + the "add childs if they contain points" blocks (:653-688 / :717-741) + erase.
__device__ void qt_divide(QtState &S, int node)
{
    const int lane = threadIdx.x;
    const unsigned long long lt = lanemask_lt();
    const int gx = S.gx[node], gy = S.gy[node];
    const int ulx = gx & 0xFFFF, urx = gx >> 16, uly = gy & 0xFFFF, bry = gy >> 16;
    const int begin = S.begin[node], cnt = S.count[node];
    const int mx = ulx + ((urx - ulx + 1) >> 1);  // UL.x + ceil((UR.x - UL.x) / 2)
    const int my = uly + ((bry - uly + 1) >> 1);
    int c[4] = {0, 0, 0, 0};
    if (cnt <= WAVE) {
        const bool act = lane < cnt;
        const uint32_t key = act ? S.A[begin + lane] : 0u;
        const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
        const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
        unsigned long long m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { m[k] = __ballot(act && q == k); c[k] = __popcll(m[k]); }
        const int o1 = c[0], o2 = o1 + c[1], o3 = o2 + c[2];
        const int offq = q == 0 ? 0 : (q == 1 ? o1 : (q == 2 ? o2 : o3));
        const unsigned long long mq = q == 0 ? m[0] : (q == 1 ? m[1] : (q == 2 ? m[2] : m[3]));
        __syncthreads();
        if (act) S.A[begin + offq + __popcll(mq & lt)] = key;
    } else {
        for (int base = 0; base < cnt; base += WAVE) {
            const bool act = base + lane < cnt;
            const uint32_t key = act ? S.A[begin + base + lane] : 0u;
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] += __popcll(__ballot(act && q == k));
        }
        int run[4] = {0, c[0], c[0] + c[1], c[0] + c[1] + c[2]};
        for (int base = 0; base < cnt; base += WAVE) {
            const bool act = base + lane < cnt;
            const uint32_t key = act ? S.A[begin + base + lane] : 0u;
            const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
            const int q = x < mx ? (y < my ? 0 : 2) : (y < my ? 1 : 3);
            unsigned long long m[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = __ballot(act && q == k);
            const int rq = q == 0 ? run[0] : (q == 1 ? run[1] : (q == 2 ? run[2] : run[3]));
            const unsigned long long mq = q == 0 ? m[0] : (q == 1 ? m[1] : (q == 2 ? m[2] : m[3]));
            if (act) S.B[begin + rq + __popcll(mq & lt)] = key;
#pragma unroll
            for (int k = 0; k < 4; ++k) run[k] += __popcll(m[k]);
        }
        __syncthreads();
        for (int i = lane; i < cnt; i += WAVE) S.A[begin + i] = S.B[begin + i];
    }
    __syncthreads();
    // children n1 (UL), n2 (UR), n3 (BL), n4 (BR) pushed to the list front in that order
    int off = begin;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c[k] > 0) {
            const int id = qt_alloc(S);
            const int cul = (k & 1) ? mx : ulx, cur = (k & 1) ? urx : mx;
            const int cuy = (k & 2) ? my : uly, cby = (k & 2) ? bry : my;
            S.gx[id] = cul | (cur << 16);
            S.gy[id] = cuy | (cby << 16);
            S.begin[id] = off;
            S.count[id] = c[k];
            S.noMore[id] = (short)(c[k] == 1);
            qt_push_front(S, id);
            if (c[k] > 1) {
                ++S.nToExpand;
                QtSized e; e.cnt = c[k]; e.ulx = cul; e.id = id;
                S.sized[S.nSized++] = e;
            }
            off += c[k];
        }
    }
    qt_erase(S, node);
    __syncthreads();
}

__global__ __launch_bounds__(64) void k_quadtree(QtArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qlds[];
    const int level = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const OrbxLevelGeom g = a.geoms[level];
    const int P = a.pool;
    QtState S;
    S.gx = (int *)qlds; S.gy = S.gx + P; S.begin = S.gy + P; S.count = S.begin + P;
    S.sized = (QtSized *)(S.count + P);
    QtSized *prevArr = S.sized + P;
    S.next = (short *)(prevArr + P); S.prev = S.next + P; S.noMore = S.prev + P; S.freeList = S.noMore + P;
    short *ord = S.freeList + P;
    uint32_t *ldsKeys = (uint32_t *)(qlds + (((size_t)P * (16 + 24 + 10)) + 15 & ~(size_t)15));

    uint32_t *selOut = a.sel + (size_t)f * a.selFrameStride + g.selBase;
    int *selCnt = a.selCount + (size_t)f * a.nlevels + level;

    // ---- ordered gather of the FAST candidates (tile slots in cell-row-major order) --------------------
    const int *tc = a.tileCount + (size_t)f * a.tilesPerFrame + g.tileBase;
    int M = 0;
    for (int t = 0; t < g.nTiles; ++t) M += tc[t];
    if (M == 0) { if (lane == 0) *selCnt = 0; return; }
    if (M <= QT_LDS_KEYS) { S.A = ldsKeys; S.B = ldsKeys + QT_LDS_KEYS; }
    else {
        S.A = a.scratchA + (size_t)f * a.candFrameStride + g.candBase;
        S.B = a.scratchB + (size_t)f * a.candFrameStride + g.candBase;
    }
    {
        const uint32_t *slots = a.cand + (size_t)f * a.candFrameStride + g.candBase;
        int off = 0;
        for (int t = 0; t < g.nTiles; ++t) {
            const int n = tc[t];
            const uint32_t *s = slots + (size_t)t * g.slotCap;
            for (int i = lane; i < n; i += WAVE) S.A[off + i] = s[i];
            off += n;
        }
    }
    __syncthreads();

    // ---- roots (:589-626) -----------------------------------------------------------------------------
    const int W = (g.w - ORBX_EDGE + 3) - ORBX_BORDER, H = (g.h - ORBX_EDGE + 3) - ORBX_BORDER;
    const int N = g.quota;
    const int nIni = (int)roundf(__fdiv_rn((float)W, (float)H));
    const float hX = __fdiv_rn((float)W, (float)nIni);
    S.head = QT_NIL; S.size = 0; S.freeTop = 0; S.bump = 0; S.nSized = 0; S.nToExpand = 0;
    {
        int tail = QT_NIL, off = 0;
        for (int r = 0; r < nIni; ++r) {
            int cnt;
            if (nIni == 1) cnt = M;
            else {
                // stable compaction of the keys whose root is r, appended to B
                cnt = 0;
                for (int base = 0; base < M; base += WAVE) {
                    const bool act = base + lane < M;
                    const uint32_t key = act ? S.A[base + lane] : 0u;
                    const bool mine = act && (int)__fdiv_rn((float)(key & 0xFFF), hX) == r;
                    const unsigned long long m = __ballot(mine);
                    if (mine) S.B[off + cnt + __popcll(m & lanemask_lt())] = key;
                    cnt += __popcll(m);
                }
            }
            if (cnt > 0) {
                const int id = S.bump++;
                S.gx[id] = (int)__fmul_rn(hX, (float)r) | ((int)__fmul_rn(hX, (float)(r + 1)) << 16);
                S.gy[id] = 0 | (H << 16);
                S.begin[id] = off; S.count[id] = cnt; S.noMore[id] = (short)(cnt == 1);
                // push_back
                S.prev[id] = (short)tail; S.next[id] = QT_NIL;
                if (tail != QT_NIL) S.next[tail] = (short)id; else S.head = id;
                tail = id;
                ++S.size;
            }
            off += cnt;
        }
        if (nIni > 1) {
            __syncthreads();
            for (int i = lane; i < M; i += WAVE) S.A[i] = S.B[i];
        }
    }
    __syncthreads();

    // ---- subdivision (:630-752) -----------------------------------------------------------------------
    bool finish = false;
    while (!finish) {
        const int prevSize = S.size;
        S.nToExpand = 0; S.nSized = 0;
        int cur = S.head;
        while (cur != QT_NIL) {
            const int nxt = S.next[cur];
            if (!S.noMore[cur]) qt_divide(S, cur);
            cur = nxt;
        }
        if (S.size >= N || S.size == prevSize) {
            finish = true;
        } else if (S.size + S.nToExpand * 3 > N) {
            while (!finish) {
                const int prevSize2 = S.size;
                const int nPrev = S.nSized;
                for (int i = lane; i < nPrev; i += WAVE) prevArr[i] = S.sized[i];
                S.nSized = 0;
                __syncthreads();
                if (lane == 0) orbx_sort::sort(prevArr, nPrev, QtSizedLess());
                __syncthreads();
                for (int j = nPrev - 1; j >= 0; --j) {
                    qt_divide(S, prevArr[j].id);
                    if (S.size >= N) break;
                }
                if (S.size >= N || S.size == prevSize2) finish = true;
            }
        }
    }

    // ---- retain the best (first strictly greatest response) key of every node, in list order (:755-771) --
    {
        int k = 0;
        for (int cur = S.head; cur != QT_NIL; cur = S.next[cur]) ord[k++] = (short)cur;
    }
    __syncthreads();
    for (int k = lane; k < S.size; k += WAVE) {
        const int node = ord[k];
        const int b = S.begin[node], n = S.count[node];
        uint32_t best = S.A[b];
        for (int i = 1; i < n; ++i) {
            const uint32_t key = S.A[b + i];
            if ((key >> 24) > (best >> 24)) best = key;
        }
        selOut[k] = best;
    }
    if (lane == 0) *selCnt = S.size;
}

static size_t qtLdsBytes(int P)
{
    size_t nodeBytes = ((size_t)P * (16 + 24 + 10) + 15) & ~(size_t)15;
    return nodeBytes + (size_t)QT_LDS_KEYS * 2 * sizeof(uint32_t);
}

hipError_t orbxLaunchQuadtree(hipStream_t st, int nframes, int nlevels, const OrbxLevelGeom *geomsHost,
                              const OrbxLevelGeom *geomsDev, const uint32_t *cand, long long candFrameStride,
                              const int *tileCount, int tilesPerFrame, uint32_t *scratchA, uint32_t *scratchB,
                              uint32_t *sel, int selFrameStride, int *selCount, int maxQuota)
{
    (void)geomsHost;
    QtArgs a;
    a.geoms = geomsDev;
    a.cand = cand; a.candFrameStride = candFrameStride;
    a.tileCount = tileCount; a.tilesPerFrame = tilesPerFrame;
    a.scratchA = scratchA; a.scratchB = scratchB;
    a.sel = sel; a.selFrameStride = selFrameStride; a.selCount = selCount;
    a.nlevels = nlevels;
    a.pool = maxQuota + 24;
    size_t lds = qtLdsBytes(a.pool);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attrSet = false;
    if (!attrSet && lds > 48 * 1024) {
        (void)hipFuncSetAttribute((const void *)k_quadtree, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attrSet = true;
    }
    dim3 grid(nlevels, nframes);
    hipLaunchKernelGGL(k_quadtree, grid, dim3(64), lds, st, a);
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

__global__ __launch_bounds__(256) void k_orient_desc(OrbxTables tb, OrbxLevelPtrs lv, const uint32_t *workKey,
                                                     const int *workPos, const int *nOut, OrbxKp *kps, uint8_t *desc,
                                                     int cap)
{
    const int f = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gi = blockIdx.x * 4 + wave;
    if (gi >= nOut[f]) return;
    const uint32_t key = workKey[(size_t)f * cap + gi];
    const int pos = workPos[(size_t)f * cap + gi];
    const int l = key >> 24, xl = key & 0xFFF, yl = (key >> 12) & 0xFFF;

    // ---- IC_Angle: m10 = sum u*I, m01 = sum v*I over the radius-15 circular patch of the un-blurred level
    const int rp = lv.rawPitch[l];
    const uint8_t *center = lv.raw[l] + (size_t)f * lv.rawFrameStride[l] + (size_t)yl * rp + xl;
    int m10 = 0, m01 = 0;
    const int u = (lane & 31) - 15;
    for (int it = 0; it < 16; ++it) {
        const int v = it * 2 + (lane >> 5) - 15;
        if (v <= 15) {
            const int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
            if (au <= tb.umax[av]) {
                const int I = center[v * rp + u];
                m10 += u * I;
                m01 += v * I;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    float angle;
    if (tb.angleMode == 1) {
        angle = atan2f((float)m01, (float)m10);
        angle = __fadd_rn(angle, (angle < 0) ? __fmul_rn(2.0f, 3.141592f) : 0.f);
        angle = __fmul_rn(angle, __fdiv_rn(180.0f, 3.141592f));
    } else {
        angle = fast_atan2_deg((float)m01, (float)m10);
    }
    if (lane == 0) kps[(size_t)f * cap + pos].angle = angle;

    // ---- steered BRIEF on the blurred level; 64 test pairs per ballot ---------------------------------
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    const float ar = __fmul_rn(angle, factorPI);
    const float a = (float)cos((double)ar), b = (float)sin((double)ar);
    const int bp = lv.blurPitch[l];
    const uint8_t *bc = lv.blur[l] + (size_t)f * lv.blurFrameStride[l] + (size_t)yl * bp + xl;
    uint8_t *dout = desc + ((size_t)f * cap + pos) * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const signed char *pt = c_pattern + (r * 64 + lane) * 4;
        const float x0 = (float)pt[0], y0 = (float)pt[1], x1 = (float)pt[2], y1 = (float)pt[3];
        const int iy0 = __float2int_rn(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)));
        const int ix0 = __float2int_rn(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)));
        const int iy1 = __float2int_rn(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)));
        const int ix1 = __float2int_rn(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)));
        const int t0 = bc[iy0 * bp + ix0], t1 = bc[iy1 * bp + ix1];
        const unsigned long long bits = __ballot(t0 < t1);
        if (lane == 0) *(unsigned long long *)(dout + r * 8) = bits;
    }
}

hipError_t orbxLaunchOrientDesc(hipStream_t st, int nframes, OrbxTables tabs, OrbxLevelPtrs lv, const uint32_t *workKey,
                                const int *workPos, const int *nOut, void *kps, uint8_t *desc, int cap)
{
    static bool patternLoaded[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !patternLoaded[dev]) {
        hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), kOrbxPattern31, 1024);
        if (e != hipSuccess) return e;
        patternLoaded[dev] = true;
    }
    dim3 grid((cap + 3) / 4, nframes);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, st, tabs, lv, workKey, workPos, nOut, (OrbxKp *)kps, desc, cap);
    return hipGetLastError();
}
