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
