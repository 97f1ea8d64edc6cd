// orbx_match.hip — 256-bit Hamming matcher kernels for gfx950.
//
//   k_hamming_matrix  all-pairs ORBmatcher::DescriptorDistance (reference src/ORBmatcher3.cc:637-653)
//   k_knn2            cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) (reference src/Frame.cc:45,1174)
//   k_stereo_rows     Hamming stage of Frame::ComputeStereoMatches (reference src/Frame.cc:841-925)
//
// Integer xor + v_bcnt_u32_b32 popcount; train descriptors are staged through LDS in tiles so each one is
// read from HBM once per workgroup; a query lives in 8 VGPRs.  Tie rule of knnMatch (strict `<` insertion,
// lower train index first) is kept by scanning train rows in index order inside a lane and merging the
// per-split results with (distance, index) lexicographic order.
#include "orbx_internal.h"
#include <limits.h>

#define MT_TILE 256  // train descriptors per LDS tile (8 KB)

__device__ __forceinline__ int ham256(const uint32_t q[8], const uint32_t *t)
{
    int d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += __popc(q[i] ^ t[i]);
    return d;
}

__global__ __launch_bounds__(256) void k_hamming_matrix(const uint8_t *q, int nq, const uint8_t *t, int nt, int *dist)
{
    __shared__ uint32_t tt[MT_TILE * 8];
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * MT_TILE;
    const int ntile = min(MT_TILE, nt - t0);
    const uint4 *tsrc = (const uint4 *)(t + (size_t)t0 * 32);
    for (int i = threadIdx.x; i < ntile * 2; i += 256) ((uint4 *)tt)[i] = tsrc[i];
    __syncthreads();
    if (qi >= nq) return;
    uint32_t qq[8];
    const uint4 *qs = (const uint4 *)(q + (size_t)qi * 32);
    uint4 v0 = qs[0], v1 = qs[1];
    qq[0] = v0.x; qq[1] = v0.y; qq[2] = v0.z; qq[3] = v0.w; qq[4] = v1.x; qq[5] = v1.y; qq[6] = v1.z; qq[7] = v1.w;
    int *drow = dist + (size_t)qi * nt + t0;
    for (int j = 0; j < ntile; ++j) drow[j] = ham256(qq, tt + j * 8);
}

hipError_t orbxLaunchHammingMatrix(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int *dist)
{
    if (nq <= 0 || nt <= 0) return hipSuccess;
    dim3 grid((nq + 255) / 256, (nt + MT_TILE - 1) / MT_TILE);
    hipLaunchKernelGGL(k_hamming_matrix, grid, dim3(256), 0, st, q, nq, t, nt, dist);
    return hipGetLastError();
}

// One lane per query; lanes of a 64-thread workgroup share the LDS train tile.  64-thread groups (instead of
// 256) so that 4000 queries still spread over 63 CUs.
__global__ __launch_bounds__(64) void k_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, int *dist)
{
    __shared__ uint32_t tt[MT_TILE * 8];
    const int qi = blockIdx.x * 64 + threadIdx.x;
    uint32_t qq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) {
        const uint4 *qs = (const uint4 *)(q + (size_t)qi * 32);
        uint4 v0 = qs[0], v1 = qs[1];
        qq[0] = v0.x; qq[1] = v0.y; qq[2] = v0.z; qq[3] = v0.w; qq[4] = v1.x; qq[5] = v1.y; qq[6] = v1.z; qq[7] = v1.w;
    }
    int d0 = INT_MAX, d1 = INT_MAX, i0 = -1, i1 = -1;
    for (int t0 = 0; t0 < nt; t0 += MT_TILE) {
        const int ntile = min(MT_TILE, nt - t0);
        __syncthreads();
        const uint4 *tsrc = (const uint4 *)(t + (size_t)t0 * 32);
        for (int i = threadIdx.x; i < ntile * 2; i += 64) ((uint4 *)tt)[i] = tsrc[i];
        __syncthreads();
        for (int j = 0; j < ntile; ++j) {
            const int d = ham256(qq, tt + j * 8);
            // batchDistance insertion: d < dist[K-1]; shift while dist[k] > d (strict) -> earlier index wins ties
            if (d < d1) {
                if (d < d0) { d1 = d0; i1 = i0; d0 = d; i0 = t0 + j; }
                else { d1 = d; i1 = t0 + j; }
            }
        }
    }
    if (qi < nq) {
        idx[qi * 2] = i0; idx[qi * 2 + 1] = i1;
        dist[qi * 2] = d0; dist[qi * 2 + 1] = d1;
    }
}

hipError_t orbxLaunchKnn2(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, int *dist)
{
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_knn2, dim3((nq + 63) / 64), dim3(64), 0, st, q, nq, t, nt, idx, dist);
    return hipGetLastError();
}

struct MtKp { float x, y, size, angle, response; int octave, class_id; };

// Row-band stereo candidates: right keypoint iR is a candidate of left keypoint iL iff
// floor(yR - r) <= (int)yL <= ceil(yR + r), r = 2 * scale[octR]  (row buckets, :858-868), |octave diff| <= 1 and
// uR in [uL - maxD, uL - minD]; candidates are visited in increasing iR (bucket push order) with a strict `<`.
__global__ __launch_bounds__(64) void k_stereo_rows(const MtKp *kl, const uint8_t *dl, int nl, const MtKp *kr,
                                                    const uint8_t *dr, int nr, int nRows, const float *scale,
                                                    float minD, float maxD, int *bestIdx, int *bestDist)
{
    __shared__ uint32_t tt[MT_TILE * 8];
    __shared__ int tMinR[MT_TILE], tMaxR[MT_TILE], tOct[MT_TILE];
    __shared__ float tU[MT_TILE];
    const int li = blockIdx.x * 64 + threadIdx.x;
    uint32_t qq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int levelL = 0, row = -1;
    float minU = 0.f, maxU = -1.f;
    bool live = false;
    if (li < nl) {
        const uint4 *qs = (const uint4 *)(dl + (size_t)li * 32);
        uint4 v0 = qs[0], v1 = qs[1];
        qq[0] = v0.x; qq[1] = v0.y; qq[2] = v0.z; qq[3] = v0.w; qq[4] = v1.x; qq[5] = v1.y; qq[6] = v1.z; qq[7] = v1.w;
        const MtKp k = kl[li];
        levelL = k.octave;
        row = (int)k.y;
        minU = __fsub_rn(k.x, maxD);
        maxU = __fsub_rn(k.x, minD);
        live = row >= 0 && row < nRows && !(maxU < 0);
    }
    int best = 100, bi = -1;  // ORBmatcher::TH_HIGH (reference src/ORBmatcher1.cc:37)
    for (int t0 = 0; t0 < nr; t0 += MT_TILE) {
        const int ntile = min(MT_TILE, nr - t0);
        __syncthreads();
        const uint4 *tsrc = (const uint4 *)(dr + (size_t)t0 * 32);
        for (int i = threadIdx.x; i < ntile * 2; i += 64) ((uint4 *)tt)[i] = tsrc[i];
        for (int i = threadIdx.x; i < ntile; i += 64) {
            const MtKp k = kr[t0 + i];
            const float r = __fmul_rn(2.0f, scale[k.octave]);
            tMaxR[i] = (int)ceilf(__fadd_rn(k.y, r));
            tMinR[i] = (int)floorf(__fsub_rn(k.y, r));
            tOct[i] = k.octave;
            tU[i] = k.x;
        }
        __syncthreads();
        if (live) {
            for (int j = 0; j < ntile; ++j) {
                if (row < tMinR[j] || row > tMaxR[j]) continue;
                const int o = tOct[j];
                if (o < levelL - 1 || o > levelL + 1) continue;
                const float uR = tU[j];
                if (uR >= minU && uR <= maxU) {
                    const int d = ham256(qq, tt + j * 8);
                    if (d < best) { best = d; bi = t0 + j; }
                }
            }
        }
    }
    if (li < nl) { bestIdx[li] = bi; bestDist[li] = best; }
}

hipError_t orbxLaunchStereoRows(hipStream_t st, const void *kl, const uint8_t *dl, int nl, const void *kr,
                                const uint8_t *dr, int nr, int nRows, const float *scaleDev, float minD, float maxD,
                                int *bestIdx, int *bestDist)
{
    if (nl <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_stereo_rows, dim3((nl + 63) / 64), dim3(64), 0, st, (const MtKp *)kl, dl, nl, (const MtKp *)kr,
                       dr, nr, nRows, scaleDev, minD, maxD, bestIdx, bestDist);
    return hipGetLastError();
}

// =====================================================================================================
// Sub-pixel stereo refinement — Frame::ComputeStereoMatches, reference src/Frame.cc:926-1011
// (the step right after the row-band Hamming stage; it reads mvImagePyramid of both extractors,
// which stays on the device here).  One wave per left keypoint: 11x11 L1 (cv::norm NORM_L1) between the left patch and
// 11 shifted right patches, parabola fit through the best three, re-scale, disparity test.
// =====================================================================================================
struct StereoLevels {
    const uint8_t *left[ORBX_MAX_LEVELS], *right[ORBX_MAX_LEVELS];
    int pitchL[ORBX_MAX_LEVELS], pitchR[ORBX_MAX_LEVELS];
    int colsR[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS], invScale[ORBX_MAX_LEVELS];
};

__device__ __forceinline__ int mt_wave_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(x, 0) + __builtin_amdgcn_readlane(x, 16) + __builtin_amdgcn_readlane(x, 32) +
           __builtin_amdgcn_readlane(x, 48);
}

__global__ __launch_bounds__(256) void k_stereo_sad(StereoLevels lv, const MtKp *kl, int nl, const MtKp *kr, const int *bestIdx,
                                                    const int *bestDist, float mbf, float minD, float maxD, float *uRight, float *depth,
                                                    int *sadDist)
{
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= nl) return;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    const int bi = bestIdx[i], bd = bestDist[i];
    if (bi >= 0 && bd < (100 + 50) / 2) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2  (:846, :927)
        const MtKp kpL = kl[i];
        const int oct = kpL.octave;
        const float uL = kpL.x;
        const float uR0 = kr[bi].x;
        const float sf = lv.invScale[oct];
        const float scaleduL = roundf(__fmul_rn(kpL.x, sf));
        const float scaledvL = roundf(__fmul_rn(kpL.y, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)L), (float)w);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)L), (float)w), 1.0f);
        if (!(iniu < 0 || endu >= (float)lv.colsR[oct])) {
            const int y0 = (int)__fsub_rn(scaledvL, (float)w), xL0 = (int)__fsub_rn(scaleduL, (float)w);
            const int xRb = (int)scaleduR0 - w;  // first column of the unshifted right patch
            const uint8_t *pl = lv.left[oct] + (size_t)y0 * lv.pitchL[oct] + xL0;
            const uint8_t *pr = lv.right[oct] + (size_t)y0 * lv.pitchR[oct] + xRb;
            // lane handles patch pixels p0 = lane and p1 = lane + 64 (121 pixels)
            const int p0 = lane, p1 = lane + 64;
            const int y0p = p0 / 11, x0p = p0 - y0p * 11, y1p = p1 / 11, x1p = p1 - y1p * 11;
            const bool has1 = p1 < 121;
            const int a0 = pl[y0p * lv.pitchL[oct] + x0p];
            const int a1 = has1 ? pl[y1p * lv.pitchL[oct] + x1p] : 0;
            int sad[11];
#pragma unroll
            for (int inc = 0; inc < 11; ++inc) {
                const int b0 = pr[y0p * lv.pitchR[oct] + x0p + inc - L];
                const int b1 = has1 ? pr[y1p * lv.pitchR[oct] + x1p + inc - L] : 0;
                const int d0 = a0 - b0, d1 = a1 - b1;
                sad[inc] = (d0 < 0 ? -d0 : d0) + (has1 ? (d1 < 0 ? -d1 : d1) : 0);
            }
            int best = INT_MAX, bestInc = 0;
            int tot[11];
#pragma unroll
            for (int inc = 0; inc < 11; ++inc) {
                tot[inc] = mt_wave_sum(sad[inc]);
                if (tot[inc] < best) { best = tot[inc]; bestInc = inc - L; }
            }
            if (!(bestInc == -L || bestInc == L)) {
                float dist1 = 0.f, dist2 = 0.f, dist3 = 0.f;
#pragma unroll
                for (int inc = 1; inc < 10; ++inc)
                    if (inc - L == bestInc) { dist1 = (float)tot[inc - 1]; dist2 = (float)tot[inc]; dist3 = (float)tot[inc + 1]; }
                const float deltaR = __fdiv_rn(__fsub_rn(dist1, dist3),
                                               __fmul_rn(2.0f, __fsub_rn(__fadd_rn(dist1, dist3), __fmul_rn(2.0f, dist2))));
                if (!(deltaR < -1 || deltaR > 1)) {
                    float bestuR = __fmul_rn(lv.scale[oct], __fadd_rn(__fadd_rn(scaleduR0, (float)bestInc), deltaR));
                    float disparity = __fsub_rn(uL, bestuR);
                    if (disparity >= minD && disparity < maxD) {
                        if (disparity <= 0) {
                            disparity = (float)0.01;
                            bestuR = (float)((double)uL - 0.01);
                        }
                        outD = __fdiv_rn(mbf, disparity);
                        outU = bestuR;
                        outS = best;
                    }
                }
            }
        }
    }
    if (lane == 0) { uRight[i] = outU; depth[i] = outD; sadDist[i] = outS; }
}

// median-based outlier rejection (:996-1010): median = SAD at rank size/2 of the accepted matches (a k-th order
// statistic, no sort needed), thDist = 1.5f * 1.4f * median; every match with SAD >= thDist is dropped.
__global__ __launch_bounds__(256) void k_stereo_outliers(int nl, float *uRight, float *depth, const int *sadDist)
{
    __shared__ int red[4];
    __shared__ int total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    auto blockCount = [&](int thr) {  // number of accepted matches with SAD <= thr
        int c = 0;
        for (int i = tid; i < nl; i += 256) { const int s = sadDist[i]; c += (s >= 0 && s <= thr); }
        c = mt_wave_sum(c);
        __syncthreads();
        if (lane == 0) red[wave] = c;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    const int valid = blockCount(INT_MAX);
    if (valid == 0) return;
    const int k = valid / 2;  // rank of vDistIdx[size / 2]
    int lo = 0, hi = 121 * 255;
    while (lo < hi) {  // smallest v with count(SAD <= v) >= k + 1
        const int mid = (lo + hi) >> 1;
        if (blockCount(mid) >= k + 1) hi = mid; else lo = mid + 1;
    }
    const float thDist = __fmul_rn(1.5f * 1.4f, (float)lo);
    for (int i = tid; i < nl; i += 256) {
        const int s = sadDist[i];
        if (s >= 0 && !((float)s < thDist)) { uRight[i] = -1.0f; depth[i] = -1.0f; }
    }
    (void)total;
}

hipError_t orbxLaunchStereoRefine(hipStream_t st, const void *levels, const void *kl, int nl, const void *kr, const int *bestIdx,
                                  const int *bestDist, float mbf, float minD, float maxD, float *uRight, float *depth, int *sadDist)
{
    if (nl <= 0) return hipSuccess;
    const StereoLevels &lv = *(const StereoLevels *)levels;
    hipLaunchKernelGGL(k_stereo_sad, dim3((nl + 3) / 4), dim3(256), 0, st, lv, (const MtKp *)kl, nl, (const MtKp *)kr, bestIdx, bestDist,
                       mbf, minD, maxD, uRight, depth, sadDist);
    hipLaunchKernelGGL(k_stereo_outliers, dim3(1), dim3(256), 0, st, nl, uRight, depth, sadDist);
    return hipGetLastError();
}
