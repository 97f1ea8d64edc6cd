// orbx_match.hip — 256-bit Hamming matcher kernels for gfx950.
//
//   k_hamming_matrix  all-pairs ORBmatcher::DescriptorDistance (reference src/ORBmatcher3.cc:637-653)
//   k_knn2_scan_mfma  cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2) (reference src/Frame.cc:45,1174) on the matrix cores
//   k_knn2_scan       the same scan on the vector ALU (ORBX_KNN_MFMA=0), k_knn2_merge: the per-split pairs merged
//   k_stereo_rows     Hamming stage of Frame::ComputeStereoMatches (reference src/Frame.cc:841-925), then the SAD refinement,
//                     the frame grid, SearchForInitialization, both SearchByProjection functions and their fisheye branches
//
// Distances are integer xor + v_bcnt_u32_b32 popcounts (a query lives in 8 VGPRs) - except in the kNN-2 scan, where descriptors
// become +-1 bytes and the distance a 256-deep int8 dot product.  The train rows are split over the grid; the tie rule of knnMatch
// (strict `<` insertion, lower train index first) is the lexicographic (distance, index) order, kept by packed keys inside a split
// and by k_knn2_merge across splits.  k_hamming_matrix stages train descriptors through LDS tiles.
#include "orbx_internal.h"
#include "../../include/orbx_libm.h"
#include <cstring>
#include <limits.h>
#include <algorithm>
#include <type_traits>

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

// ---- kNN-2 (cv::BFMatcher::knnMatch, k = 2) ------------------------------------------------------------------------
// Work decomposition: a wave = 64 queries (one per lane, the descriptor in 8 VGPRs) x one SPLIT of the train rows; the
// grid is (query waves) x (splits) x (pairs), so a single 4000 x 4000 pair already fills the chip (63 x 32 waves) and a
// batch of stereo pairs is one launch.  A train row is wave-uniform: it is read with scalar loads (s_load_dwordx8 from
// the scalar cache - no LDS staging, no barriers) and xor-ed into the lanes' queries; v_bcnt_u32_b32 accumulates.
// Order: batchDistance inserts with a strict `<`, so the result is the two smallest (distance, train index) pairs in
// lexicographic order.  Distance and index are packed into one key (distance << 23 | index; distance <= 256), the scan
// keeps the two smallest keys with min / med3, and the per-split pairs are merged by the same rule in k_knn2_merge -
// exact for any split boundaries, ties included.
#define KN_IDX_BITS 23                    // train rows per pair < 2^23
#define KN_NONE 0xFFFFFFFFu
struct Knn2Args {
    const uint8_t *q, *t;                 // descriptors of pair 0
    long long qPairStride, tPairStride;   // bytes between consecutive pairs
    const int *nqDev, *ntDev;             // per-pair row counts on the device (NULL: nq / nt for every pair)
    const int *qBeginDev, *tBeginDev;     // per-pair first row (Frame::monoLeft / monoRight, src/Frame.cc:1174), NULL: 0
    int nq, nt;                           // host-side counts = capacities of the grid
    int nSplit, splitRows;
    uint32_t *part;                       // [pair][split][nq][2] partial keys (nSplit > 1)
    int *idx, *dist;                      // [pair][nq][2]
    int sharedBound;                      // k_knn2_scan_mfma: the insert bound shared by the four lane groups of a query (ORBX_KNN_SHARED_BOUND)
    int pipeline;                         // k_knn2_scan_mfma: products of tile t + 1 issued before the keys of tile t are handled (ORBX_KNN_PIPELINE)
};

__device__ __forceinline__ void kn_insert(uint32_t &k0, uint32_t &k1, uint32_t key)
{
    uint32_t m;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m) : "v"(k0), "v"(key), "v"(k1));  // second smallest of {k0 <= k1, key}
    k0 = min(k0, key);
    k1 = m;
}

__device__ __forceinline__ void kn_store(int *idx, int *dist, size_t o, uint32_t k0, uint32_t k1)
{
    idx[o] = k0 == KN_NONE ? -1 : (int)(k0 & ((1u << KN_IDX_BITS) - 1));
    dist[o] = k0 == KN_NONE ? INT_MAX : (int)(k0 >> KN_IDX_BITS);
    idx[o + 1] = k1 == KN_NONE ? -1 : (int)(k1 & ((1u << KN_IDX_BITS) - 1));
    dist[o + 1] = k1 == KN_NONE ? INT_MAX : (int)(k1 >> KN_IDX_BITS);
}

__global__ __launch_bounds__(64) void k_knn2_scan(Knn2Args a)
{
    const int pair = blockIdx.z, split = blockIdx.y;
    const int qb = a.qBeginDev ? a.qBeginDev[pair] : 0, tb = a.tBeginDev ? a.tBeginDev[pair] : 0;
    const int nq = min(a.nqDev ? a.nqDev[pair] - qb : a.nq, a.nq);  // rows [qb, nqDev[pair]) are the queries
    const int nt = min(a.ntDev ? a.ntDev[pair] - tb : a.nt, a.nt);
    const int q0 = blockIdx.x * 64;
    if (q0 >= nq) return;  // wave-uniform
    const int qi = q0 + (int)threadIdx.x;
    uint32_t qq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (qi < nq) {
        const uint4 *qs = (const uint4 *)(a.q + (size_t)pair * a.qPairStride + (size_t)(qb + qi) * 32);
        const uint4 v0 = qs[0], v1 = qs[1];
        qq[0] = v0.x; qq[1] = v0.y; qq[2] = v0.z; qq[3] = v0.w; qq[4] = v1.x; qq[5] = v1.y; qq[6] = v1.z; qq[7] = v1.w;
    }
    const int j0 = split * a.splitRows, j1 = min(j0 + a.splitRows, nt);
    // wave-uniform train rows through the scalar cache
    typedef const __attribute__((address_space(4))) uint32_t CU32;
    CU32 *tr = (CU32 *)(a.t + (size_t)pair * a.tPairStride + (size_t)tb * 32);
    uint32_t k0 = KN_NONE, k1 = KN_NONE;
    auto one = [&](CU32 *row, int j) {
        uint32_t d = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) d += __popc(qq[w] ^ row[w]);
        kn_insert(k0, k1, (d << KN_IDX_BITS) | (uint32_t)j);
    };
    int j = j0;
    for (; j + 4 <= j1; j += 4) {  // four rows' scalar loads in flight
        CU32 *row = tr + (size_t)j * 8;
        one(row, j); one(row + 8, j + 1); one(row + 16, j + 2); one(row + 24, j + 3);
    }
    for (; j < j1; ++j) one(tr + (size_t)j * 8, j);
    if (qi >= nq) return;
    if (a.nSplit == 1) kn_store(a.idx, a.dist, ((size_t)pair * a.nq + qi) * 2, k0, k1);
    else {
        uint32_t *p = a.part + (((size_t)pair * a.nSplit + split) * a.nq + qi) * 2;
        p[0] = k0; p[1] = k1;
    }
}

__global__ __launch_bounds__(256) void k_knn2_merge(Knn2Args a)
{
    const int pair = blockIdx.y;
    const int qb = a.qBeginDev ? a.qBeginDev[pair] : 0;
    const int nq = min(a.nqDev ? a.nqDev[pair] - qb : a.nq, a.nq);
    const int qi = blockIdx.x * 256 + threadIdx.x;
    if (qi >= nq) return;
    uint32_t k0 = KN_NONE, k1 = KN_NONE;
    for (int s = 0; s < a.nSplit; ++s) {
        const uint32_t *p = a.part + (((size_t)pair * a.nSplit + s) * a.nq + qi) * 2;
        const uint32_t c0 = p[0], c1 = p[1];
        kn_insert(k0, k1, c0);
        kn_insert(k0, k1, c1);
    }
    kn_store(a.idx, a.dist, ((size_t)pair * a.nq + qi) * 2, k0, k1);
}

// ---- the same scan on the matrix cores ------------------------------------------------------------------------------------------
// With the bits of a descriptor written as +1 / -1 bytes, q . t = 256 - 2 * Hamming(q, t): a 256-deep int8 dot product, exact in int32.
// The scan above is bound by vector-instruction issue (16 xor / popcount instructions per train row and 64 queries, 0.66 of that
// peak); v_mfma_i32_16x16x64_i8 does 16 x 16 pairs x 64 bits in 16 cycles - four products per tile of 16 train rows x 16 queries.
//   * workgroup = 4 waves = 256 queries x one split of the train rows; a wave holds its 64 queries (4 tiles of 16) expanded in
//     registers as the B operand of all four k-steps (64 VGPRs) for the whole scan;
//   * the train rows pass through LDS 16 at a time: every thread expands 16 bits of the tile to 16 bytes - two 8-byte reads of a
//     256-entry table in LDS (byte -> eight -+1 bytes; 16 vector instructions when computed: a nibble -> four 0 / 1 bytes by a 24-bit
//     multiply and a mask, v_perm_b32 as a two-entry lookup) -, double-buffered in blocks of four tiles, one barrier per block; the
//     packed bits of the next block but one are already requested; rows are padded to 272 bytes so that the 16-byte operand reads of the 16 rows spread over the banks;
//   * the result tile has the query on the lane (column) and four train rows in the lane's registers, so a lane folds four keys of
//     ONE query (train rows are expanded with the opposite sign, so the product is 2 * Hamming - 256 and a key is one shift-add:
//     (256 + acc) << 22 | row = Hamming << 23 | row); two minimum instructions and a compare on the raw products decide whether any of the
//     four can enter the pair kept so far (after the first rows they almost never do) - only then keys are formed and inserted;
//   * at the end the four lane groups of a query (train rows 4g .. 4g + 3 of every tile) merge their pairs.
// Same keys, same order, same splits and merge kernel as the scan above: bit-identical results (tests/test_matcher_gpu.py runs both).
typedef int kn_v4i __attribute__((ext_vector_type(4)));
#define KNM_ROWB 272  // bytes of an expanded train row in LDS (256 + 16 of padding)
#define KNM_BLK 64    // train rows per LDS buffer = per barrier (four tiles of 16)

// bits 0..3 -> bytes 0..3, bit value b -> byte tab[b] (tab = 0x....[b = 1][b = 0]): the multiplication puts bit i at bit 0 of byte i,
// v_perm_b32 looks the four 0 / 1 bytes up in the two-entry table
__device__ __forceinline__ uint32_t knm_expand4(uint32_t bits4, uint32_t tab)
{
    const uint32_t b01 = __umul24(bits4, 0x00204081u) & 0x01010101u;
    return __builtin_amdgcn_perm(0u, tab, b01);
}
__device__ __forceinline__ kn_v4i knm_expand16(uint32_t w16, uint32_t tab)
{
    kn_v4i v;
    v[0] = (int)knm_expand4(w16 & 15u, tab); v[1] = (int)knm_expand4((w16 >> 4) & 15u, tab);
    v[2] = (int)knm_expand4((w16 >> 8) & 15u, tab); v[3] = (int)knm_expand4((w16 >> 12) & 15u, tab);
    return v;
}
#define KNM_TAB_Q 0x000001FFu  // queries: 1 -> +1, 0 -> -1
#define KNM_TAB_T 0x0000FF01u  // train rows: 1 -> -1, 0 -> +1, so that the product is 2 * Hamming - 256 and the key one shift-add

__global__ __launch_bounds__(256) void k_knn2_scan_mfma(Knn2Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tileL[2][KNM_BLK * KNM_ROWB];
    const int pair = blockIdx.z, split = blockIdx.y;
    const int qb = a.qBeginDev ? a.qBeginDev[pair] : 0, tb = a.tBeginDev ? a.tBeginDev[pair] : 0;
    const int nq = min(a.nqDev ? a.nqDev[pair] - qb : a.nq, a.nq);
    const int nt = min(a.ntDev ? a.ntDev[pair] - tb : a.nt, a.nt);
    if ((int)(blockIdx.x * 256) >= nq) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 256 + wv * 64;  // this wave's queries q0 .. q0 + 63: tile qt = queries q0 + 16 qt + n
    const uint8_t *qBase = a.q + (size_t)pair * a.qPairStride + (size_t)qb * 32;
    const uint8_t *tBase = a.t + (size_t)pair * a.tPairStride + (size_t)tb * 32;
    // B operands: k-step s of tile qt = bits 64 s + 16 g .. + 15 of query q0 + 16 qt + n = the 16-bit word 4 s + g of its descriptor
    kn_v4i B[4][4];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int qi = q0 + 16 * qt + n;
        uint32_t w[4] = {0, 0, 0, 0};
        if (qi < nq) {
#pragma unroll
            for (int sK = 0; sK < 4; ++sK) w[sK] = ((const uint16_t *)(qBase + (size_t)qi * 32))[4 * sK + g];
        }
#pragma unroll
        for (int sK = 0; sK < 4; ++sK) B[qt][sK] = knm_expand16(w[sK], KNM_TAB_Q);
    }
    const int j0 = split * a.splitRows, j1 = min(j0 + a.splitRows, nt);
    // expansion duty of this thread: row tid >> 4 of the tile, 16-bit word tid & 15
    const int er = tid >> 4, ew = tid & 15;
    auto loadBits = [&](int jb, uint32_t (&w)[4]) {  // packed bits of the block at row jb (rows beyond the split read row j1 - 1: masked below)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(jb + 16 * u + er, max(j1 - 1, 0));
            w[u] = ((const uint16_t *)(tBase + (size_t)j * 32))[ew];
        }
    };
    // byte -> eight -+1 bytes through a 2 KB table in LDS: two 8-byte reads instead of 16 vector instructions per thread and tile
    __shared__ uint2 lutT[256];
    lutT[tid] = make_uint2(knm_expand4((uint32_t)tid & 15u, KNM_TAB_T), knm_expand4((uint32_t)tid >> 4, KNM_TAB_T));
    auto expandTo = [&](int buf, const uint32_t (&w)[4]) {  // rows 16 u + er of the block, word ew = 4 s + g' -> bytes s * 64 + g' * 16
        uint2 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { lo[u] = lutT[w[u] & 255u]; hi[u] = lutT[w[u] >> 8]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kn_v4i v;
            v[0] = (int)lo[u].x; v[1] = (int)lo[u].y; v[2] = (int)hi[u].x; v[3] = (int)hi[u].y;
            *(kn_v4i *)(tileL[buf] + (16 * u + er) * KNM_ROWB + (ew >> 2) * 64 + (ew & 3) * 16) = v;
        }
    };
    __syncthreads();
    uint32_t k0[4], k1[4];
    int thr[4];  // k1 >> 22
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) { k0[qt] = KN_NONE; k1[qt] = KN_NONE; thr[qt] = (int)(KN_NONE >> 22); }
    if (j0 < j1) {
        // One barrier per block of KNM_BLK rows, not per tile: a barrier couples the four SIMDs a workgroup's waves run on, and every
        // 500 cycles was too often.  The packed bits of the next block but one are already requested.
        uint32_t wNext[4];
        loadBits(j0, wNext);
        expandTo(0, wNext);
        loadBits(j0 + KNM_BLK, wNext);
        __syncthreads();
        int buf = 0;
        // the 16 products of tile u of the current buffer into acc (train row n of the tile x the wave's four query tiles)
        auto issueTile = [&](int u, kn_v4i (&acc)[4]) {
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) acc[qt] = kn_v4i{0, 0, 0, 0};
            kn_v4i A[4];  // train row n of the tile, all four k-steps requested before the first product
#pragma unroll
            for (int sK = 0; sK < 4; ++sK) A[sK] = *(const kn_v4i *)(tileL[buf] + (16 * u + n) * KNM_ROWB + sK * 64 + g * 16);
#pragma unroll
            for (int sK = 0; sK < 4; ++sK) asm volatile("" : "+v"(A[sK]));
#pragma unroll
            for (int sK = 0; sK < 4; ++sK) {
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) acc[qt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[sK], B[qt][sK], acc[qt], 0, 0, 0);
            }
        };
        // the keys of one finished tile: guard, insert (rows jt .. jt + 15)
        auto processTile = [&](kn_v4i (&acc)[4], int jt, auto tailTag) {
            constexpr bool TAIL = decltype(tailTag)::value;
            // lane: query n of every tile qt, train rows jt + 4 g + r; acc = 2 * Hamming - 256
            const int jr = jt + 4 * g;
            uint32_t base[4];  // 256 << 22 | row: one add per row and tile, then a shift-add per key
#pragma unroll
            for (int r = 0; r < 4; ++r) { base[r] = (uint32_t)((1 << 30) + jr + r); asm volatile("" : "+v"(base[r])); }
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                // A key can only enter the pair if its distance field is SMALLER than the second key's, 256 + acc < k1 >> 22: the rows
                // of a scan come in increasing order, so at equal distance the new row is the larger one.  Tested on the smallest of the
                // four products; the keys are formed behind the test only.
                const int mAcc = min(min(acc[qt][0], acc[qt][1]), min(acc[qt][2], acc[qt][3]));
                if (__ballot(mAcc + 256 < thr[qt]) != 0ull) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        uint32_t key = ((uint32_t)acc[qt][r] << 22) + base[r];  // (256 + acc) << 22 | row = Hamming << 23 | row
                        if (TAIL && jr + r >= j1) key = KN_NONE;
                        kn_insert(k0[qt], k1[qt], key);
                    }
                    // (Round 4, measured and dropped: the bound shared by the four lane groups of a query - the smallest of their second
                    // keys, exact because each group's pair alone holds two keys below its own second key - refreshed at EVERY insert event.
                    // It cuts the events of an unsplit 4000-row scan from 85 % to ~40 % of the tile x query-tile steps, but the exchange per
                    // event cost more than the events saved: 2.01 against 1.87 ms per 512 pairs.  ORBX_KNN_SHARED_BOUND=1 keeps that form;
                    // round 5 refreshes the shared bound once per BLOCK of four tiles instead, below.)
                    if (a.sharedBound == 1) {   // the query-wide bound through v_permlane32/16_swap (no LDS round trip)
                        int t = (int)(k1[qt] >> 22);
                        auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
                        t = min((int)s32[0], (int)s32[1]);
                        auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
                        thr[qt] = min((int)s16[0], (int)s16[1]);
                    } else
                    thr[qt] = min(thr[qt], (int)(k1[qt] >> 22));   // (the shared bound of the last refresh is never above the lane's own)
                }
            }
        };
        auto tile = [&](int jt, int u, auto tailTag) {
            kn_v4i acc[4];
            issueTile(u, acc);
            processTile(acc, jt, tailTag);
        };
        // the same without the guard: every key is formed and inserted (3 vector instructions each, 52 per tile, no branch), so that the
        // whole tile is ONE scheduling region together with the next tile's 16 products and can be interleaved with them instruction by
        // instruction (ORBX_KNN_PIPELINE=2)
        auto processTileAll = [&](kn_v4i (&acc)[4], int jt) {
            const int jr = jt + 4 * g;
            uint32_t base[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) base[r] = (uint32_t)((1 << 30) + jr + r);
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) kn_insert(k0[qt], k1[qt], ((uint32_t)acc[qt][r] << 22) + base[r]);
            }
        };
        // in-order issue: a wave that has 16 products queued issues nothing else until the last one is accepted (16 x 22 cycles) - unless the
        // vector instructions stand BETWEEN the products in program order: one product, three vector instructions, 16 times
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        };
        auto refreshBound = [&]() {
            // Round 5: the insert bound SHARED by the four lane groups of a query (train rows 4 g .. 4 g + 3 of every tile), refreshed once per
            // block of four tiles: thr = the smallest of the groups' second keys.  Exact - the final pair lies below every group's second key,
            // and a bound only falls, so a stale one costs extra inserts, never a miss - and it cuts the insert events (taken for the whole
            // wave when ONE of its 64 lanes has a candidate) from 85 - 100 % of the steps to what a four times longer scan would see.
            if (a.sharedBound == 2) {
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) {
                    int t = thr[qt];
                    auto s32 = __builtin_amdgcn_permlane32_swap((unsigned)t, (unsigned)t, false, false);
                    t = min((int)s32[0], (int)s32[1]);
                    auto s16 = __builtin_amdgcn_permlane16_swap((unsigned)t, (unsigned)t, false, false);
                    thr[qt] = min((int)s16[0], (int)s16[1]);
                }
            }
        };
        if (a.pipeline) {
            // Round 5 (ORBX_KNN_PIPELINE=1; 2, the default, drops the guard and interleaves - above): software-pipelined tiles.  A wave that reads its accumulators right behind its 16
            // products waits for the matrix core (~350 cycles per tile) and only then runs its ~60 vector instructions; with the products
            // of tile t + 1 ISSUED before the keys of tile t are handled (a second accumulator set), the vector work of every wave runs in
            // the shadow of its own products - not only beside another wave's.  The pending tile crosses the block barrier; the last,
            // partial block of a split runs unpipelined (its rows beyond the split are masked).
            kn_v4i accA[4], accB[4];
            bool pend = false;   // accB holds tile 3 of the previous block (rows jbPrev + 48 ..), not yet handled
            int jtPend = 0;
            int jb = j0;
            if (a.pipeline == 2) {
                for (; jb + KNM_BLK <= j1; jb += KNM_BLK) {   // whole blocks, unguarded keys interleaved with the next tile's products
                    if (jb + KNM_BLK < j1) expandTo(buf ^ 1, wNext);
                    loadBits(jb + 2 * KNM_BLK, wNext);
                    if (pend) { issueTile(0, accA); processTileAll(accB, jtPend); interleave(); }
                    else issueTile(0, accA);
                    issueTile(1, accB); processTileAll(accA, jb); interleave();
                    issueTile(2, accA); processTileAll(accB, jb + 16); interleave();
                    issueTile(3, accB); processTileAll(accA, jb + 32); interleave();
                    pend = true; jtPend = jb + 48;
                    __syncthreads();
                    buf ^= 1;
                }
                if (pend) { processTileAll(accB, jtPend); pend = false; }
#pragma unroll
                for (int qt = 0; qt < 4; ++qt) thr[qt] = (int)(k1[qt] >> 22);
            }
            for (; jb + KNM_BLK <= j1; jb += KNM_BLK) {   // whole blocks
                if (jb + KNM_BLK < j1) expandTo(buf ^ 1, wNext);
                loadBits(jb + 2 * KNM_BLK, wNext);
                issueTile(0, accA);
                if (pend) processTile(accB, jtPend, std::false_type{});
                issueTile(1, accB);
                processTile(accA, jb, std::false_type{});
                issueTile(2, accA);
                processTile(accB, jb + 16, std::false_type{});
                issueTile(3, accB);
                processTile(accA, jb + 32, std::false_type{});
                pend = true; jtPend = jb + 48;
                refreshBound();
                __syncthreads();
                buf ^= 1;
            }
            if (pend) processTile(accB, jtPend, std::false_type{});
            if (jb < j1) {                                 // the last rows of the split
                for (int u = 0; u < 4 && jb + 16 * u < j1; ++u) {
                    if (jb + 16 * u + 16 <= j1) tile(jb + 16 * u, u, std::false_type{});
                    else tile(jb + 16 * u, u, std::true_type{});
                }
            }
        } else
        for (int jb = j0; jb < j1; jb += KNM_BLK) {
            if (jb + KNM_BLK < j1) expandTo(buf ^ 1, wNext);  // the next block into the other buffer
            loadBits(jb + 2 * KNM_BLK, wNext);
            if (jb + KNM_BLK <= j1) {  // four whole tiles
#pragma unroll
                for (int u = 0; u < 4; ++u) tile(jb + 16 * u, u, std::false_type{});
            } else {                   // the last rows of the split: keys of rows beyond it are masked
                for (int u = 0; u < 4 && jb + 16 * u < j1; ++u) {
                    if (jb + 16 * u + 16 <= j1) tile(jb + 16 * u, u, std::false_type{});
                    else tile(jb + 16 * u, u, std::true_type{});
                }
            }
            refreshBound();
            __syncthreads();
            buf ^= 1;
        }
    }
    // the four lane groups of a query hold the pairs of disjoint train rows: merge (same rule as the split merge)
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const uint32_t c0 = (uint32_t)__shfl_xor((int)k0[qt], o), c1 = (uint32_t)__shfl_xor((int)k1[qt], o);
            kn_insert(k0[qt], k1[qt], c0);
            kn_insert(k0[qt], k1[qt], c1);
        }
        const int qi = q0 + 16 * qt + n;
        if (g == 0 && qi < nq) {
            if (a.nSplit == 1) kn_store(a.idx, a.dist, ((size_t)pair * a.nq + qi) * 2, k0[qt], k1[qt]);
            else {
                uint32_t *p = a.part + (((size_t)pair * a.nSplit + split) * a.nq + qi) * 2;
                p[0] = k0[qt]; p[1] = k1[qt];
            }
        }
    }
}

// ---- the matrix-core scan with 32 x 32 tiles -----------------------------------------------------------------------------------------
// k_knn2_scan_mfma spends ~70 vector instructions per 16 products (16 train rows x 64 queries): a lane holds FOUR train rows of a query
// per tile, so every 16 x 16 tile costs it a minimum of four, a compare and a ballot - the vector ALU, not the matrix core, set its pace
// (0.40 of the matrix peak).  v_mfma_i32_32x32x32_i8 leaves SIXTEEN train rows of one query in a lane: the same test is one minimum tree
// over 16 values per 32 x 32 tile, i.e. a quarter of the vector work per descriptor pair (8 products of 32 x 32 x 32 per tile of 32 rows
// x 32 queries).  MEASURED SLOWER than the 16 x 16 form (2.06 against 1.80 ms per 512 pairs of 4000 x 4000; round 2 saw the same): 64
// registers of query operands + 64 of accumulators leave two waves per SIMD, and the block barrier then idles the SIMD.  Kept behind
// ORBX_KNN_MFMA=2 as the measured alternative; the default stays k_knn2_scan_mfma.  Same expanded LDS tile (rows of +-1 bytes in bit order, 272-byte pitch: the 16-byte
// operand reads of rows r and r + 16 fall into different lane groups of ds_read_b128), same keys, same splits and merge.
typedef int kn_v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_knn2_scan_mfma32(Knn2Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t tileL[2][KNM_BLK * KNM_ROWB];
    const int pair = blockIdx.z, split = blockIdx.y;
    const int qb = a.qBeginDev ? a.qBeginDev[pair] : 0, tb = a.tBeginDev ? a.tBeginDev[pair] : 0;
    const int nq = min(a.nqDev ? a.nqDev[pair] - qb : a.nq, a.nq);
    const int nt = min(a.ntDev ? a.ntDev[pair] - tb : a.nt, a.nt);
    if ((int)(blockIdx.x * 256) >= nq) return;  // workgroup-uniform
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.x * 256 + wv * 64;  // this wave's queries: tile qt = queries q0 + 32 qt + n
    const uint8_t *qBase = a.q + (size_t)pair * a.qPairStride + (size_t)qb * 32;
    const uint8_t *tBase = a.t + (size_t)pair * a.tPairStride + (size_t)tb * 32;
    // B operands: k-step s of tile qt = bits 32 s + 16 h .. + 15 of query q0 + 32 qt + n = the 16-bit word 2 s + h of its descriptor
    kn_v4i B[2][8];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = q0 + 32 * qt + n;
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qi < nq) {
#pragma unroll
            for (int sK = 0; sK < 8; ++sK) w[sK] = ((const uint16_t *)(qBase + (size_t)qi * 32))[2 * sK + h];
        }
#pragma unroll
        for (int sK = 0; sK < 8; ++sK) B[qt][sK] = knm_expand16(w[sK], KNM_TAB_Q);
    }
    const int j0 = split * a.splitRows, j1 = min(j0 + a.splitRows, nt);
    const int er = tid >> 4, ew = tid & 15;  // expansion duty of this thread: row tid >> 4 of a 16-row group, 16-bit word tid & 15
    auto loadBits = [&](int jb, uint32_t (&w)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(jb + 16 * u + er, max(j1 - 1, 0));
            w[u] = ((const uint16_t *)(tBase + (size_t)j * 32))[ew];
        }
    };
    __shared__ uint2 lutT[256];
    lutT[tid] = make_uint2(knm_expand4((uint32_t)tid & 15u, KNM_TAB_T), knm_expand4((uint32_t)tid >> 4, KNM_TAB_T));
    auto expandTo = [&](int buf, const uint32_t (&w)[4]) {  // rows 16 u + er of the block, word ew -> bytes 16 ew .. 16 ew + 15 (bit order)
        uint2 lo[4], hi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { lo[u] = lutT[w[u] & 255u]; hi[u] = lutT[w[u] >> 8]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            kn_v4i v;
            v[0] = (int)lo[u].x; v[1] = (int)lo[u].y; v[2] = (int)hi[u].x; v[3] = (int)hi[u].y;
            *(kn_v4i *)(tileL[buf] + (16 * u + er) * KNM_ROWB + ew * 16) = v;
        }
    };
    __syncthreads();
    uint32_t k0[2], k1[2];
    int thr[2];  // k1 >> 22
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) { k0[qt] = KN_NONE; k1[qt] = KN_NONE; thr[qt] = (int)(KN_NONE >> 22); }
    if (j0 < j1) {
        uint32_t wNext[4];
        loadBits(j0, wNext);
        expandTo(0, wNext);
        loadBits(j0 + KNM_BLK, wNext);
        __syncthreads();
        int buf = 0;
        auto tile = [&](int jt, int u, auto tailTag) {  // train rows jt .. jt + 31 = rows 32 u .. of the buffer
            constexpr bool TAIL = decltype(tailTag)::value;
            kn_v16i acc[2];
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[qt][i] = 0;
            const uint8_t *rowp = tileL[buf] + (32 * u + n) * KNM_ROWB + h * 16;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                kn_v4i A[4];  // four k-steps of train row n requested before the first product
#pragma unroll
                for (int sK = 0; sK < 4; ++sK) A[sK] = *(const kn_v4i *)(rowp + (4 * half + sK) * 32);
#pragma unroll
                for (int sK = 0; sK < 4; ++sK) asm volatile("" : "+v"(A[sK]));
#pragma unroll
                for (int sK = 0; sK < 4; ++sK) {
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt) acc[qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[sK], B[qt][4 * half + sK], acc[qt], 0, 0, 0);
                }
            }
            // lane: query n of tile qt, train rows jt + (i & 3) + 8 (i >> 2) + 4 h; acc = 2 * Hamming - 256
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                int m4[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) m4[c] = min(min(acc[qt][4 * c], acc[qt][4 * c + 1]), min(acc[qt][4 * c + 2], acc[qt][4 * c + 3]));
                const int mAcc = min(min(m4[0], m4[1]), min(m4[2], m4[3]));
                if (__ballot(mAcc + 256 < thr[qt]) != 0ull) {  // (rows come in increasing order: an equal distance cannot enter the pair)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int row = jt + (i & 3) + 8 * (i >> 2) + 4 * h;
                        uint32_t key = ((uint32_t)acc[qt][i] << 22) + (uint32_t)((1 << 30) + row);  // (256 + acc) << 22 | row = Hamming << 23 | row
                        if (TAIL && row >= j1) key = KN_NONE;
                        kn_insert(k0[qt], k1[qt], key);
                    }
                    thr[qt] = (int)(k1[qt] >> 22);
                }
            }
        };
        // a whole block of 64 rows: both 32-row tiles together, so that four independent accumulator chains (2 row tiles x 2 query tiles)
        // are in flight per wave - with two, the dependent products of a chain waited for each other
        auto test16 = [&](const kn_v16i &acc, int qt, int jt) {
            int m4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) m4[c] = min(min(acc[4 * c], acc[4 * c + 1]), min(acc[4 * c + 2], acc[4 * c + 3]));
            const int mAcc = min(min(m4[0], m4[1]), min(m4[2], m4[3]));
            if (__ballot(mAcc + 256 < thr[qt]) != 0ull) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = jt + (i & 3) + 8 * (i >> 2) + 4 * h;
                    kn_insert(k0[qt], k1[qt], ((uint32_t)acc[i] << 22) + (uint32_t)((1 << 30) + row));
                }
                thr[qt] = (int)(k1[qt] >> 22);
            }
        };
        auto block = [&](int jb) {
            kn_v16i acc[2][2];  // [row tile][query tile]
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[u][qt][i] = 0;
            const uint8_t *rowp = tileL[buf] + n * KNM_ROWB + h * 16;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {  // two k-steps of both row tiles per round of operand reads
                kn_v4i A[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int sK = 0; sK < 2; ++sK) A[u][sK] = *(const kn_v4i *)(rowp + 32 * u * KNM_ROWB + (2 * q4 + sK) * 32);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int sK = 0; sK < 2; ++sK) asm volatile("" : "+v"(A[u][sK]));
#pragma unroll
                for (int sK = 0; sK < 2; ++sK)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int qt = 0; qt < 2; ++qt)
                            acc[u][qt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[u][sK], B[qt][2 * q4 + sK], acc[u][qt], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) test16(acc[u][qt], qt, jb + 32 * u);
        };
        for (int jb = j0; jb < j1; jb += KNM_BLK) {
            if (jb + KNM_BLK < j1) expandTo(buf ^ 1, wNext);  // the next block into the other buffer
            loadBits(jb + 2 * KNM_BLK, wNext);
            if (jb + KNM_BLK <= j1) block(jb);
            else {  // the last rows of the split: keys of rows beyond it are masked
                for (int u = 0; u < 2 && jb + 32 * u < j1; ++u) {
                    if (jb + 32 * u + 32 <= j1) tile(jb + 32 * u, u, std::false_type{});
                    else tile(jb + 32 * u, u, std::true_type{});
                }
            }
            __syncthreads();
            buf ^= 1;
        }
    }
    // the two lane halves of a query hold the pairs of disjoint train rows: merge (same rule as the split merge)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const uint32_t c0 = (uint32_t)__shfl_xor((int)k0[qt], 32), c1 = (uint32_t)__shfl_xor((int)k1[qt], 32);
        kn_insert(k0[qt], k1[qt], c0);
        kn_insert(k0[qt], k1[qt], c1);
        const int qi = q0 + 32 * qt + n;
        if (h == 0 && qi < nq) {
            if (a.nSplit == 1) kn_store(a.idx, a.dist, ((size_t)pair * a.nq + qi) * 2, k0[qt], k1[qt]);
            else {
                uint32_t *p = a.part + (((size_t)pair * a.nSplit + split) * a.nq + qi) * 2;
                p[0] = k0[qt]; p[1] = k1[qt];
            }
        }
    }
}

// Number of train splits for `npairs` pairs of nq x nt rows: enough waves to fill 1024 SIMDs a few times over, splits of at
// least 64 rows.  `forceSplit` > 0 overrides it (tests of the merge).
int orbxKnn2Splits(int npairs, int nq, int nt, int forceSplit)
{
    if (nt <= 0) return 1;
    if (forceSplit > 0) return std::min(forceSplit, nt);
    const long long qWaves = (long long)npairs * ((nq + 63) / 64);
    long long s = (4096 + qWaves - 1) / std::max(qWaves, 1ll);
    s = std::min<long long>(s, (nt + 63) / 64);
    return (int)std::max<long long>(s, 1);
}

hipError_t orbxLaunchKnn2Batch(hipStream_t st, int npairs, const uint8_t *q, long long qPairStride, const int *nqDev, const int *qBeginDev,
                               int nq, const uint8_t *t, long long tPairStride, const int *ntDev, const int *tBeginDev, int nt,
                               int nSplit, uint32_t *part, int *idx, int *dist)
{
    if (npairs <= 0 || nq <= 0) return hipSuccess;
    if (nt >= (1 << KN_IDX_BITS)) return hipErrorInvalidValue;
    Knn2Args a;
    a.q = q; a.t = t; a.qPairStride = qPairStride; a.tPairStride = tPairStride;
    a.nqDev = nqDev; a.ntDev = ntDev; a.qBeginDev = qBeginDev; a.tBeginDev = tBeginDev;
    a.nq = nq; a.nt = nt;
    a.nSplit = std::max(nSplit, 1);
    a.splitRows = nt > 0 ? (nt + a.nSplit - 1) / a.nSplit : 1;
    a.part = part; a.idx = idx; a.dist = dist;
    // ORBX_KNN_MFMA=0: the scan on the vector ALU (A/B timing, tests); default: the matrix-core scan
    // ORBX_KNN_MFMA: 0 the scan on the vector ALU, 1 (default) matrix cores with 16 x 16 x 64 products, 2 with 32 x 32 x 32 products - a
    // quarter of the vector instructions per descriptor pair and a 23 % lower matrix floor (34 cycles per 32768 products against 22 per
    // 16384, tools/experiments/mfma_i8_rate.hip), but 202 VGPRs = two waves per SIMD: 2.06 against 1.80 ms per 512 pairs of 4000 x 4000
    static const int mfma = [] { const char *e = getenv("ORBX_KNN_MFMA"); return e ? atoi(e) : 1; }();
    static const int sharedBound = [] { const char *e = getenv("ORBX_KNN_SHARED_BOUND"); return e ? atoi(e) : 2; }();   // 0 off, 1 per event (round 4), 2 per block
    a.sharedBound = sharedBound;
    static const int pipeline = [] { const char *e = getenv("ORBX_KNN_PIPELINE"); return e ? atoi(e) : 2; }();   // 0 tile by tile, 1 pipelined + guard, 2 pipelined, unguarded, interleaved
    a.pipeline = pipeline;
    if (mfma == 2) hipLaunchKernelGGL(k_knn2_scan_mfma32, dim3((nq + 255) / 256, a.nSplit, npairs), dim3(256), 0, st, a);
    else if (mfma) hipLaunchKernelGGL(k_knn2_scan_mfma, dim3((nq + 255) / 256, a.nSplit, npairs), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_knn2_scan, dim3((nq + 63) / 64, a.nSplit, npairs), dim3(64), 0, st, a);
    if (a.nSplit > 1) hipLaunchKernelGGL(k_knn2_merge, dim3((nq + 255) / 256, npairs), dim3(256), 0, st, a);
    return hipGetLastError();
}

struct MtKp { float x, y, size, angle, response; int octave, class_id; };

// Row-band stereo candidates: right keypoint iR is a candidate of left keypoint iL iff
// floor(yR - r) <= (int)yL <= ceil(yR + r), r = 2 * scale[octR]  (row buckets, :858-868), |octave diff| <= 1 and
// uR in [uL - maxD, uL - minD]; candidates are visited in increasing iR (bucket push order) with a strict `<`.
// blockIdx.y = stereo pair: keypoints / descriptors / results of pair p start p * pairStride records further, and the counts come
// from the device arrays nlDev / nrDev when those are given (the extractors' per-frame keypoint counts) - a batch of pairs is one
// launch and nothing returns to the host in between.
struct StereoBatch {
    long long kpStrideL, kpStrideR;  // keypoint / descriptor records between consecutive pairs (left, right extractor)
    long long outStride;             // result records between consecutive pairs
    const int *nlDev, *nrDev;
};

// Row buckets of the right image (vRowIndices, reference src/Frame.cc:858-868) as CSR, built on the device: one workgroup per pair counts
// the band [floor(y - r), ceil(y + r)] of every right keypoint into LDS row counters, scans them, and fills the lists with LDS cursors.
// The lists are unordered - the search below does not depend on the visiting order (see there).
__device__ __forceinline__ int mt_excl_scan(int v, int &total);
__device__ __forceinline__ unsigned mt_wave_min_u32(unsigned v);

__global__ __launch_bounds__(256) void k_stereo_buckets(const MtKp *kr, int nr, int nRows, const float *scale, int *rowStart, int *rowItems,
                                                        int itemStride, StereoBatch sb)
{
    extern __shared__ int bk[];  // [nRows] counters, then cursors; + 256 partial sums
    __shared__ float sc[ORBX_MAX_LEVELS];
    __shared__ int waveBase[4];
    {
        const size_t p = blockIdx.x;
        kr += p * sb.kpStrideR; rowStart += p * (size_t)(nRows + 1); rowItems += p * (size_t)itemStride;
        if (sb.nrDev) nr = min(nr, sb.nrDev[p]);
    }
    int *cnt = bk;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int y = tid; y < nRows; y += 256) cnt[y] = 0;
    if (tid < ORBX_MAX_LEVELS) sc[tid] = scale[tid];
    __syncthreads();
    // A workgroup per pair is a latency chain: the keypoints are read ONCE, four loads in flight, and their bands stay in registers
    // for the fill (up to 8 x 256 keypoints; beyond that the band is computed again), the level scales come from LDS.
    constexpr int KEEP = 8;
    uint32_t bandOf[KEEP];  // lo | hi << 16, 0xFFFFFFFF: no keypoint
    auto band = [&](float y, int octave) -> uint32_t {
        const float r = __fmul_rn(2.0f, sc[min(max(octave, 0), ORBX_MAX_LEVELS - 1)]);
        const int hi = min((int)ceilf(__fadd_rn(y, r)), nRows - 1);
        const int lo = max((int)floorf(__fsub_rn(y, r)), 0);
        return lo <= hi ? (uint32_t)lo | ((uint32_t)hi << 16) : 0xFFFFFFFFu;  // (a band outside the image holds no row)
    };
#pragma unroll
    for (int k = 0; k < KEEP; ++k) bandOf[k] = 0xFFFFFFFFu;
    for (int i0 = 0; i0 < nr; i0 += 1024) {
        float ky[4];
        int ko[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = min(i0 + 256 * u + tid, nr - 1); ky[u] = kr[i].y; ko[u] = kr[i].octave; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u + tid;
            if (i < nr) {
                const uint32_t b = band(ky[u], ko[u]);
                if (i0 == 0) bandOf[u] = b; else if (i0 == 1024) bandOf[4 + u] = b;
                if (b != 0xFFFFFFFFu)
                    for (int y = (int)(b & 0xFFFF); y <= (int)(b >> 16); ++y) atomicAdd(&cnt[y], 1);
            }
        }
    }
    __syncthreads();
    // exclusive scan over the rows: each thread a contiguous chunk, a wave scan over the chunk sums, the four wave totals through LDS
    const int per = (nRows + 255) / 256, y0 = tid * per, y1 = min(y0 + per, nRows);
    int sum = 0;
    for (int y = y0; y < y1; ++y) sum += cnt[y];
    int waveTotal;
    int base = mt_excl_scan(sum, waveTotal);
    if (lane == 0) waveBase[wv] = waveTotal;
    __syncthreads();
    for (int w = 0; w < wv; ++w) base += waveBase[w];
    for (int y = y0; y < y1; ++y) { const int c = cnt[y]; rowStart[y] = base; cnt[y] = base; base += c; }
    if (tid == 255) rowStart[nRows] = base;
    __syncthreads();
    for (int i0 = 0; i0 < nr; i0 += 1024) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + 256 * u + tid;
            if (i < nr) {
                uint32_t b;
                if (i0 == 0) b = bandOf[u]; else if (i0 == 1024) b = bandOf[4 + u]; else b = band(kr[i].y, kr[i].octave);
                if (b != 0xFFFFFFFFu)
                    for (int y = (int)(b & 0xFFFF); y <= (int)(b >> 16); ++y) {
                        const int pos = atomicAdd(&cnt[y], 1);
                        if (pos < itemStride) rowItems[pos] = i;
                    }
            }
        }
    }
}

// minimum over the 16 lanes of a DPP row (every lane ends up with it): lane pairs, quad halves, mirrored halves of 8, mirrored row
__device__ __forceinline__ uint32_t mt_row_min_u32(uint32_t x)
{
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false));
    x = min(x, (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false));
    return x;
}

// Sixteen lanes per left keypoint (four keypoints per wave), one candidate per lane and step: the candidates are the right keypoints in
// the bucket of its row (:884-886), filtered by |octave difference| <= 1 and uR in [uL - maxD, uL - minD]; the reference visits them in
// increasing iR with a strict `<` starting from TH_HIGH, i.e. it keeps the candidate with the smallest (distance, iR) among those with
// distance < TH_HIGH - a minimum that does not depend on the order, taken here over the lanes.  (One lane per left keypoint walking its
// bucket paid the three dependent loads bucket entry -> keypoint -> descriptor once per candidate, and a wave ran until the longest
// of its 64 buckets was done: 0.42 ms per 512 pairs of 1200 keypoints against 0.0x now.)
__global__ __launch_bounds__(256) void k_stereo_rows(const MtKp *kl, const uint8_t *dl, int nl, const MtKp *kr,
                                                     const uint8_t *dr, int nr, int nRows, const int *rowStart, const int *rowItems,
                                                     int itemStride, float minD, float maxD, int *bestIdx, int *bestDist, StereoBatch sb)
{
    {
        const size_t p = blockIdx.y;
        kl += p * sb.kpStrideL; dl += p * sb.kpStrideL * 32; kr += p * sb.kpStrideR; dr += p * sb.kpStrideR * 32;
        bestIdx += p * sb.outStride; bestDist += p * sb.outStride;
        rowStart += p * (size_t)(nRows + 1); rowItems += p * (size_t)itemStride;
        if (sb.nlDev) nl = min(nl, sb.nlDev[p]);
        if ((int)(blockIdx.x * 16) >= nl) return;
    }
    const int sub = threadIdx.x & 15;
    const int li = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = li < nl;
    const int lic = min(li, nl - 1);
    uint32_t qq[8];
    {
        const uint4 *qs = (const uint4 *)(dl + (size_t)lic * 32);
        const uint4 v0 = qs[0], v1 = qs[1];
        qq[0] = v0.x; qq[1] = v0.y; qq[2] = v0.z; qq[3] = v0.w; qq[4] = v1.x; qq[5] = v1.y; qq[6] = v1.z; qq[7] = v1.w;
    }
    const float kx = kl[lic].x, ky = kl[lic].y;
    const int levelL = kl[lic].octave, row = (int)ky;
    const float minU = __fsub_rn(kx, maxD), maxU = __fsub_rn(kx, minD);
    uint32_t bestKey = (100u << 20);  // ORBmatcher::TH_HIGH (reference src/ORBmatcher1.cc:37): only smaller distances are taken
    int e = 0, e1 = 0;
    if (live && row >= 0 && row < nRows && !(maxU < 0)) { e = rowStart[row] + sub; e1 = min(rowStart[row + 1], itemStride); }
    while (__ballot(e < e1) != 0ull) {
        if (e < e1) {
            const int iR = rowItems[e];
            const int octR = kr[iR].octave;
            const float xR = kr[iR].x;
            if (octR >= levelL - 1 && octR <= levelL + 1 && xR >= minU && xR <= maxU) {
                const uint4 *ts = (const uint4 *)(dr + (size_t)iR * 32);
                const uint4 t0 = ts[0], t1 = ts[1];
                const uint32_t tw[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
                const uint32_t d = (uint32_t)ham256(qq, tw);
                bestKey = min(bestKey, (d << 20) | (uint32_t)iR);
            }
        }
        e += 16;
    }
    bestKey = mt_row_min_u32(bestKey);
    if (live && sub == 0) {
        const bool found = bestKey < (100u << 20);
        bestIdx[li] = found ? (int)(bestKey & 0xFFFFF) : -1;
        bestDist[li] = found ? (int)(bestKey >> 20) : 100;
    }
}

static StereoBatch stereoBatch(const OrbxStereoBatch *b)
{
    StereoBatch sb;
    sb.kpStrideL = b ? b->kpStrideL : 0; sb.kpStrideR = b ? b->kpStrideR : 0; sb.outStride = b ? b->outStride : 0;
    sb.nlDev = b ? b->nlDev : nullptr; sb.nrDev = b ? b->nrDev : nullptr;
    return sb;
}

hipError_t orbxLaunchStereoRows(hipStream_t st, const void *kl, const uint8_t *dl, int nl, const void *kr,
                                const uint8_t *dr, int nr, int nRows, const float *scaleDev, float minD, float maxD,
                                int *bestIdx, int *bestDist, const OrbxStereoBatch *batch, int *rowStart, int *rowItems, int itemStride)
{
    if (nl <= 0) return hipSuccess;
    if (nRows <= 0 || nRows > 8192 || nr >= (1 << 20) || !rowStart || !rowItems) return hipErrorInvalidValue;
    const int npairs = batch ? batch->npairs : 1;
    const StereoBatch sb = stereoBatch(batch);
    hipLaunchKernelGGL(k_stereo_buckets, dim3(npairs), dim3(256), (size_t)(nRows + 256) * sizeof(int), st, (const MtKp *)kr, nr, nRows, scaleDev, rowStart,
                       rowItems, itemStride, sb);
    hipLaunchKernelGGL(k_stereo_rows, dim3((nl + 15) / 16, npairs), dim3(256), 0, st, (const MtKp *)kl, dl, nl, (const MtKp *)kr,
                       dr, nr, nRows, rowStart, rowItems, itemStride, minD, maxD, bestIdx, bestDist, sb);
    return hipGetLastError();
}

// =====================================================================================================
// Sub-pixel stereo refinement — Frame::ComputeStereoMatches, reference src/Frame.cc:926-1011
// (the step right after the row-band Hamming stage; it reads mvImagePyramid of both extractors,
// which stays on the device here).  Sixteen lanes per left keypoint: 11x11 L1 (cv::norm NORM_L1) between the left patch and
// 11 shifted right patches, parabola fit through the best three, re-scale, disparity test.
// =====================================================================================================
struct StereoLevels {
    const uint8_t *left[ORBX_MAX_LEVELS], *right[ORBX_MAX_LEVELS];
    int pitchL[ORBX_MAX_LEVELS], pitchR[ORBX_MAX_LEVELS];
    int colsR[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS], invScale[ORBX_MAX_LEVELS];
    long long frameStrideL[ORBX_MAX_LEVELS], frameStrideR[ORBX_MAX_LEVELS];  // bytes between the level images of consecutive pairs
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

// sum over the 16 lanes of a DPP row (every lane ends up with it)
__device__ __forceinline__ int mt_row_sum(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true);
    return x;
}

// Sixteen lanes per left keypoint (four keypoints per wave); lane r < 11 owns row r of the 11 x 11 patch: 11 left bytes in three dwords,
// the 21 right bytes of all eleven shifts in six (loads from the exact, byte-unaligned addresses - nothing beyond the patch rows is
// read), a shifted window is three v_alignbyte and its row SAD three v_sad_u8; the rows are summed over the lanes of the DPP row.
// (A wave per keypoint with one or two pixels per lane issued 24 byte loads and 88 lane reductions per keypoint: 0.54 ms per 512 pairs
// of 1200 keypoints.)
__global__ __launch_bounds__(256) void k_stereo_sad(StereoLevels lv, const MtKp *kl, int nl, const MtKp *kr, const int *bestIdx,
                                                    const int *bestDist, float mbf, float minD, float maxD, float *uRight, float *depth,
                                                    int *sadDist, StereoBatch sb)
{
    const size_t pair = blockIdx.y;
    kl += pair * sb.kpStrideL; kr += pair * sb.kpStrideR;
    bestIdx += pair * sb.outStride; bestDist += pair * sb.outStride;
    uRight += pair * sb.outStride; depth += pair * sb.outStride; sadDist += pair * sb.outStride;
    if (sb.nlDev) nl = min(nl, sb.nlDev[pair]);
    if ((int)(blockIdx.x * 16) >= nl) return;
    const int sub = threadIdx.x & 15;
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i < nl;
    const int ic = min(i, nl - 1);
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    float outU = -1.0f, outD = -1.0f;
    int outS = -1;
    const int bi = bestIdx[ic], bd = bestDist[ic];
    if (live && bi >= 0 && bd < (100 + 50) / 2) {  // thOrbDist = (TH_HIGH + TH_LOW) / 2  (:846, :927)
        const float kx = kl[ic].x, ky = kl[ic].y;
        const int oct = kl[ic].octave;
        const float uL = kx;
        const float uR0 = kr[bi].x;
        const float sf = lv.invScale[oct];
        const float scaleduL = roundf(__fmul_rn(kx, sf));
        const float scaledvL = roundf(__fmul_rn(ky, sf));
        const float scaleduR0 = roundf(__fmul_rn(uR0, sf));
        const int w = 5, L = 5;
        const float iniu = __fsub_rn(__fadd_rn(scaleduR0, (float)L), (float)w);
        const float endu = __fadd_rn(__fadd_rn(__fadd_rn(scaleduR0, (float)L), (float)w), 1.0f);
        if (!(iniu < 0 || endu >= (float)lv.colsR[oct])) {
            const int y0 = (int)__fsub_rn(scaledvL, (float)w), xL0 = (int)__fsub_rn(scaleduL, (float)w);
            const int xRb = (int)scaleduR0 - w;  // first column of the unshifted right patch
            const int r = min(sub, 10);
            const uint8_t *pl = lv.left[oct] + pair * lv.frameStrideL[oct] + (size_t)(y0 + r) * lv.pitchL[oct] + xL0;
            const uint8_t *pr = lv.right[oct] + pair * lv.frameStrideR[oct] + (size_t)(y0 + r) * lv.pitchR[oct] + xRb - L;
            const bool rowLane = sub < 11;
            // left row: bytes 0..10; right row R[0..20] = columns xRb - 5 .. xRb + 15
            uint32_t a0 = *(const u32_unaligned *)pl, a1 = *(const u32_unaligned *)(pl + 4), a2 = *(const u32_unaligned *)(pl + 7) >> 8;
            uint32_t rw[7];
#pragma unroll
            for (int k = 0; k < 5; ++k) rw[k] = *(const u32_unaligned *)(pr + 4 * k);
            rw[5] = *(const u32_unaligned *)(pr + 17) >> 24;  // R[20] (R[21..23] are never part of a window: masked below)
            rw[6] = 0;
            if (!rowLane) { a0 = a1 = a2 = 0; rw[0] = rw[1] = rw[2] = rw[3] = rw[4] = rw[5] = 0; }
            int tot[11];
#pragma unroll
            for (int inc = 0; inc < 11; ++inc) {
                const int q = inc >> 2, sh = inc & 3;
                const uint32_t W0 = sh ? __builtin_amdgcn_alignbyte(rw[q + 1], rw[q], sh) : rw[q];
                const uint32_t W1 = sh ? __builtin_amdgcn_alignbyte(rw[q + 2], rw[q + 1], sh) : rw[q + 1];
                const uint32_t W2 = (sh ? __builtin_amdgcn_alignbyte(rw[q + 3], rw[q + 2], sh) : rw[q + 2]) & 0x00FFFFFFu;
                uint32_t sd = __builtin_amdgcn_sad_u8(a0, W0, 0u);
                sd = __builtin_amdgcn_sad_u8(a1, W1, sd);
                sd = __builtin_amdgcn_sad_u8(a2, W2, sd);
                tot[inc] = mt_row_sum((int)sd);
            }
            int best = INT_MAX, bestInc = 0;
#pragma unroll
            for (int inc = 0; inc < 11; ++inc)
                if (tot[inc] < best) { best = tot[inc]; bestInc = inc - L; }
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
    if (live && sub == 0) { uRight[i] = outU; depth[i] = outD; sadDist[i] = outS; }
}

// median-based outlier rejection (:996-1010): median = SAD at rank size/2 of the accepted matches (a k-th order
// statistic, no sort needed), thDist = 1.5f * 1.4f * median; every match with SAD >= thDist is dropped.
__global__ __launch_bounds__(256) void k_stereo_outliers(int nl, float *uRight, float *depth, const int *sadDist, StereoBatch sb)
{
    {
        const size_t pair = blockIdx.x;
        uRight += pair * sb.outStride; depth += pair * sb.outStride; sadDist += pair * sb.outStride;
        if (sb.nlDev) nl = min(nl, sb.nlDev[pair]);
    }
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
                                  const int *bestDist, float mbf, float minD, float maxD, float *uRight, float *depth, int *sadDist,
                                  const OrbxStereoBatch *batch)
{
    if (nl <= 0) return hipSuccess;
    static_assert(sizeof(StereoLevels) == sizeof(OrbxStereoLevels), "host / device level tables differ");
    const StereoLevels &lv = *(const StereoLevels *)levels;
    const int npairs = batch ? batch->npairs : 1;
    hipLaunchKernelGGL(k_stereo_sad, dim3((nl + 15) / 16, npairs), dim3(256), 0, st, lv, (const MtKp *)kl, nl, (const MtKp *)kr, bestIdx, bestDist,
                       mbf, minD, maxD, uRight, depth, sadDist, stereoBatch(batch));
    hipLaunchKernelGGL(k_stereo_outliers, dim3(npairs), dim3(256), 0, st, nl, uRight, depth, sadDist, stereoBatch(batch));
    return hipGetLastError();
}

// =====================================================================================================
// the frame grid (Frame::AssignFeaturesToGrid / GetFeaturesInArea, reference src/Frame.cc:387-418,
// 687-766) and ORBmatcher::SearchForInitialization (src/ORBmatcher1.cc:650-765) with ComputeThreeMaxima
// (src/ORBmatcher3.cc:592-633).
// The search is sequential in the reference - keypoint i1 sees the distances recorded by the matches of all earlier i1
// (vMatchedDistance) and steals their partners (vnMatches21) - so one wave walks i1 in order and spends its lanes on the
// candidates of the current i1; a batch of frame pairs would run one wave each.
// =====================================================================================================
#define GRID_COLS 64  // FRAME_GRID_COLS (reference include/Frame.h:45)
#define GRID_ROWS 48  // FRAME_GRID_ROWS (:44)
#define GRID_CELLS (GRID_COLS * GRID_ROWS)

struct GridBounds { float minX, minY, invW, invH; };

// exclusive prefix sum over the wave (all 64 lanes active): Kogge-Stone inside the rows of 16 lanes with DPP row shifts, then the row
// totals with the two row broadcasts - six VALU operations instead of six ds_bpermute round trips
__device__ __forceinline__ int mt_excl_scan(int v, int &total)
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

// One wave per frame: cell = ix * GRID_ROWS + iy of every keypoint (PosInGrid: std::round, half away from zero; -1 outside),
// counts, exclusive prefix (cellStart[GRID_CELLS + 1]) and the items of every cell in push_back order (increasing index).
// blockIdx.x = frame of a batch: keypoints / items / cellOf advance by kpStride records, cellStart by GRID_CELLS + 1, n = nDev[frame]
// Ordering point of a SINGLE-WAVE workgroup whose lanes talk through LDS: the wave's LDS operations execute in order, so it is enough
// to wait for them and to keep the compiler from moving LDS accesses across.  __syncthreads() also waits for the wave's global
// STORES (s_waitcnt vmcnt(0)) - a write acknowledgement (~1 us) per step of a loop that stores one result per step.
__device__ __forceinline__ void mt_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#define GB_LDS_N 8192
#define GB_NT 256  // four waves for the parallel parts (cells of the keypoints, prefix over the 3072 cells); the ordered placement is wave 0's
static_assert(GRID_CELLS % GB_NT == 0, "k_grid_build: cells per thread");
__global__ __launch_bounds__(GB_NT) void k_grid_build(const MtKp *k, int n, GridBounds b, int *cellStart, int *items, int *cellOf, long long kpStride,
                                                      const int *nDev)
{
    {
        const size_t p = blockIdx.x;
        k += p * kpStride; items += p * kpStride; cellOf += p * kpStride; cellStart += p * (GRID_CELLS + 1);
        if (nDev) n = min(n, nDev[p]);
    }
    __shared__ int cnt[GRID_CELLS];
    __shared__ int occ[GRID_CELLS];  // keypoints of the current 64 per cell
    // cells of the first GB_LDS_N keypoints (0xFFFF: outside the grid) for the placement loop
    __shared__ uint16_t cellL[GB_LDS_N];
    __shared__ int waveTot[GB_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < GRID_CELLS; c += GB_NT) { cnt[c] = 0; occ[c] = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4 * GB_NT) {  // four independent loads per thread and step
        float vx[4], vy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = min(i0 + GB_NT * u + tid, n - 1); vx[u] = k[i].x; vy[u] = k[i].y; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + GB_NT * u + tid;
            if (i < n) {
                const int px = (int)roundf(__fmul_rn(__fsub_rn(vx[u], b.minX), b.invW));
                const int py = (int)roundf(__fmul_rn(__fsub_rn(vy[u], b.minY), b.invH));
                const int c = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px * GRID_ROWS + py;
                cellOf[i] = c;
                if (i < GB_LDS_N) cellL[i] = (uint16_t)c;
                if (c >= 0) atomicAdd(&cnt[c], 1);
            }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the cells: a run of consecutive cells per thread, wave scan over the runs, wave totals through LDS
        constexpr int CPT = GRID_CELLS / GB_NT;
        int v[CPT], sum = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { v[j] = cnt[tid * CPT + j]; sum += v[j]; }
        int total;
        int o = mt_excl_scan(sum, total);
        if (lane == 0) waveTot[wave] = total;
        __syncthreads();
        int all = 0;
        for (int w = 0; w < GB_NT / 64; ++w) { const int t = waveTot[w]; if (w < wave) o += t; all += t; }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            cellStart[tid * CPT + j] = o;
            cnt[tid * CPT + j] = o;  // becomes the write cursor of the cell
            o += v[j];
        }
        if (tid == 0) cellStart[GRID_CELLS] = all;
    }
    __syncthreads();
    if (wave != 0) return;
    // stable placement: 64 keypoints per step in index order.  A keypoint that is alone in its cell among these 64 (nearly all of them)
    // goes straight to the cell's cursor; lanes that share a cell are ranked by lane, one cell per round.
    for (int ib = 0; ib < n; ib += 64) {
        const int i = ib + lane;
        int c = -1;
        if (i < n) {
            if (i < GB_LDS_N) { const int cl = cellL[i]; c = cl == 0xFFFF ? -1 : cl; }
            else c = cellOf[i];
        }
        if (c >= 0) atomicAdd(&occ[c], 1);
        mt_wave_sync();
        const bool alone = c >= 0 && occ[c] == 1;
        if (alone) { items[cnt[c]] = i; cnt[c] += 1; }
        mt_wave_sync();
        if (c >= 0) occ[c] = 0;
        unsigned long long todo = __ballot(c >= 0 && !alone);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int cl = __shfl(c, leader);
            const unsigned long long same = __ballot(c == cl);
            if (c == cl) items[cnt[cl] + __popcll(same & ((1ull << lane) - 1ull))] = i;
            mt_wave_sync();
            if (lane == leader) cnt[cl] += __popcll(same);
            mt_wave_sync();
            todo &= ~same;
        }
    }
}

// the octave-0 subset of F2 sorted by grid cell (k_si_sort): positions, key low bits, descriptor halves, cell starts - per pair
#define SI_CELLS_PAD (GRID_CELLS + 64)   // cell starts per pair (GRID_CELLS + 1 used)
struct SiSorted { float2 *xy; uint32_t *code; uint4 *dlo, *dhi; uint16_t *cellStart; };

struct SearchInitArgs {
    const MtKp *k1; const uint8_t *d1; int n1;
    const MtKp *k2; const uint8_t *d2; int n2;
    const int *cellStart, *items;  // grid of F2
    GridBounds b;
    float *prev;                   // vbPrevMatched [n1][2], in / out
    int *m12, *m21, *mdist;        // vnMatches12 [n1], vnMatches21 [n2], vMatchedDistance [n2]
    signed char *binOf;            // [n1] rotation-histogram bin of the match recorded for i1, -1: none
    int *nmatches;
    int window; float nnratio; int checkOri;
    // batch of (F1, F2) pairs (blockIdx.x of k_search_init_lds): per-pair strides in keypoint records, counts on the device
    long long stride1, stride2;
    const int *n1Dev, *n2Dev;
    int initPrev;                  // 1: vbPrevMatched starts as the positions of F1's keypoints (what the initializer passes first)
    // k_search_init_lds: the pre-pass results (nullptr: none).  top[t][SI_TOPK]: smallest candidate keys of the t-th octave-0 keypoint of
    // F1 (k_si_topk), topAng[t] its angle, list1G[t] its index, nL1G their number; origG / n0G: the octave-0 subset of F2 (k_si_subset);
    // all per pair (cap1 = stride of the t-indexed arrays).  topUse: entries of a list the walk looks at.
    const uint32_t *top; const float *topAng; const uint16_t *list1G, *origG; const int *nL1G, *n0G; int topUse, cap1;
    int ldsN1;                     // k_search_init_lds: octave-0 keypoints of F1 its LDS arrays are sized for (n1 when unknown)
    int blockWalk;                 // k_search_init_lds: 1 = eight keypoints per step of the walk where they do not depend on each other (ORBX_SI_BLOCK)
    SiSorted sorted;               // k_search_init_rounds: the sorted subset of F2 (pre-pass), for the scans behind undecided lists
    int roundsCap;                 // k_search_init_rounds: octave-0 keypoints of F1 it takes (SIR_CAP1; ORBX_SI_ROUNDS_CAP lowers it for the tests)
    int *walkPair;                 // batch: [npairs] written by k_search_init_rounds - 1 = this pair is beyond its capacity and is left to
                                   // k_search_init_lds, which then skips the others (nullptr: every pair takes the walk)
};

__global__ __launch_bounds__(64) void k_search_init(SearchInitArgs a)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;  // reference src/ORBmatcher1.cc:37-39
    __shared__ int hist[32];
    __shared__ int cand[512];  // candidate indices of the current i1, in GetFeaturesInArea order (larger sets are re-walked)
    const int lane = threadIdx.x;
    if (lane < 32) hist[lane] = 0;
    for (int i = lane; i < a.n1; i += 64) { a.m12[i] = -1; a.binOf[i] = -1; }
    for (int i = lane; i < a.n2; i += 64) { a.m21[i] = -1; a.mdist[i] = INT_MAX; }
    __threadfence_block();
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    const float r = (float)a.window;
    for (int i1 = 0; i1 < a.n1; ++i1) {
        if (a.k1[i1].octave > 0) continue;  // wave-uniform
        const int level1 = a.k1[i1].octave;
        const float x = a.prev[2 * i1], y = a.prev[2 * i1 + 1];
        // Frame::GetFeaturesInArea(x, y, r, level1, level1)
        const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        if (cx0 >= GRID_COLS) continue;
        const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        if (cx1 < 0) continue;
        const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        if (cy0 >= GRID_ROWS) continue;
        const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        if (cy1 < 0) continue;
        const int ny = cy1 - cy0 + 1, nCells = (cx1 - cx0 + 1) * ny;
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(a.d1 + (size_t)i1 * 32))[w];
        // lane-local best / second best in candidate order `ord`; the lanes are merged at the end
        int bestD = INT_MAX, best2 = INT_MAX, bestI = -1, bestOrd = INT_MAX;
        auto consider = [&](int i2, int ord) {
            const MtKp kp = a.k2[i2];
            if (kp.octave < level1 || kp.octave > level1) return;  // bCheckLevels with minLevel == maxLevel == level1
            if (!(fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r)) return;
            const uint32_t *t = (const uint32_t *)(a.d2 + (size_t)i2 * 32);
            int dist = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ t[w]);
            if (a.mdist[i2] <= dist) return;
            // the reference's sequential rule (`if dist < best ... else if dist < best2`) written so that it does not depend on
            // the order in which a lane meets its candidates: best = smallest (dist, ord), best2 = second smallest distance
            if (dist < bestD || (dist == bestD && ord < bestOrd)) { best2 = bestD; bestD = dist; bestI = i2; bestOrd = ord; }
            else if (dist < best2) best2 = dist;
        };
        int ordBase = 0, nCand = 0;
        for (int cb = 0; cb < nCells; cb += 64) {  // cells in (ix outer, iy inner) order
            const int ci = cb + lane;
            int s = 0, e = 0;
            if (ci < nCells) {
                const int ix = cx0 + ci / ny, iy = cy0 + ci % ny;
                s = a.cellStart[ix * GRID_ROWS + iy]; e = a.cellStart[ix * GRID_ROWS + iy + 1];
            }
            int total;
            const int off = ordBase + mt_excl_scan(e - s, total);
            for (int j = s; j < e; ++j) {  // (cells hold a few keypoints; every item gets its position in the reference's order)
                const int o = off + (j - s);
                if (o < 512) cand[o] = a.items[j];
                else consider(a.items[j], o);  // beyond the staging list: evaluated directly (same (dist, ord) rule)
            }
            ordBase += total;
        }
        nCand = min(ordBase, 512);
        __threadfence_block();
        __syncthreads();
        for (int o = lane; o < nCand; o += 64) consider(cand[o], o);  // independent loads for 64 candidates at a time
        __syncthreads();
        // merge: best = min (dist, ord); best2 = second smallest of the multiset
        unsigned key = bestI >= 0 ? ((unsigned)bestD << 20) | (unsigned)min(bestOrd, (1 << 20) - 1) : 0xFFFFFFFFu;
        unsigned kmin = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
        if (kmin == 0xFFFFFFFFu) continue;  // no candidate survived
        const unsigned long long win = __ballot(key == kmin);
        const int wl = __ffsll((long long)win) - 1;
        const int gBestD = __shfl(bestD, wl), gBestI = __shfl(bestI, wl);
        int second = lane == wl ? best2 : bestD;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) second = min(second, __shfl_xor(second, o));
        if (gBestD <= TH_LOW && (float)gBestD < __fmul_rn((float)second, a.nnratio)) {
            if (lane == 0) {
                const int old = a.m21[gBestI];
                if (old >= 0) a.m12[old] = -1;
                a.m12[i1] = gBestI; a.m21[gBestI] = i1; a.mdist[gBestI] = gBestD;
                if (a.checkOri) {
                    float rot = __fsub_rn(a.k1[i1].angle, a.k2[gBestI].angle);
                    if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                    int bin = (int)roundf(__fmul_rn(rot, factor));
                    if (bin == HISTO_LENGTH) bin = 0;
                    a.binOf[i1] = (signed char)bin;
                    hist[bin] += 1;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
    }
    if (a.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima (every lane, uniform)
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        for (int i = lane; i < a.n1; i += 64) {
            const int b = a.binOf[i];
            if (b >= 0 && b != ind1 && b != ind2 && b != ind3 && a.m12[i] >= 0) a.m12[i] = -1;
        }
        __threadfence_block();
        __syncthreads();
    }
    int cntM = 0;
    for (int i = lane; i < a.n1; i += 64) {
        const int m = a.m12[i];
        if (m >= 0) { a.prev[2 * i] = a.k2[m].x; a.prev[2 * i + 1] = a.k2[m].y; ++cntM; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cntM += __shfl_xor(cntM, o);
    // nmatches as the reference counts it (++ per accepted match, -- per stolen or rotation-filtered one) = entries still >= 0
    if (lane == 0) *a.nmatches = cntM;
}

#define SI_CAP2 2048   // octave-0 keypoints of F2 held in LDS (48 B each)
#define SI_CAND 1024   // staged candidates per F1 keypoint (more are evaluated straight from the cells)

// ---- SearchForInitialization, the part that does not depend on the walk ------------------------------------------------------------
// What keypoint i1 sees of the earlier ones is only `vMatchedDistance[i2] <= dist` (candidates an earlier keypoint matched at least as
// well are skipped) - the candidate set itself and every distance are fixed.  So they are computed for all i1 at once, a wave each, and
// the SI_TOPK candidates with the smallest (distance, visiting order) are handed to the sequential walk, which then, for nearly every
// keypoint, only looks at those: the first one not skipped is the reference's best, the next one its second best.
// key = dist << 23 | cell column << 17 | cell row << 11 | index in the octave-0 subset of F2: ascending keys = ascending distance, ties in
// Frame::GetFeaturesInArea's visiting order (cells column-major, push order = index order inside a cell).
#define SI_TOPK 8
#define SI_KEY_NONE 0xFFFFFFFFu

// octave-0 subsets of both frames in index order, one workgroup per pair and frame: orig[c2] = index in F2 (n0 of them), list1[t] = index
// in F1 (nL1).  Chunks of 64 keypoints, a wave each: the ballot of a chunk and its count go to LDS, a prefix over the chunk counts gives
// every chunk its place in the list.  (One wave per frame walked its 5000 keypoints in 20 dependent steps: 26 - 37 us.)
#define SUB_NT 1024
#define SUB_CHUNKS 1024   // 65536 keypoints
__global__ __launch_bounds__(SUB_NT) void k_si_subset(SearchInitArgs a, uint16_t *orig, int *n0Out, uint16_t *list1, int *nL1Out)
{
    __shared__ unsigned long long mask[SUB_CHUNKS];
    __shared__ int base[SUB_CHUNKS];
    __shared__ int waveSum[SUB_NT / 64];
    const size_t p = blockIdx.x;
    if (a.n1Dev) {
        a.n1 = min(a.n1, a.n1Dev[p]); a.n2 = min(a.n2, a.n2Dev[p]);
        a.k1 += p * a.stride1; a.k2 += p * a.stride2;
    }
    orig += p * (size_t)SI_CAP2; list1 += p * (size_t)a.cap1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool second = blockIdx.y == 0;   // blockIdx.y: 0 the subset of F2 (octave == 0), 1 the list of F1 (!(octave > 0))
    const MtKp *k = second ? a.k2 : a.k1;
    const int n = second ? a.n2 : a.n1, nChunks = (n + 63) >> 6;
    for (int c0 = wave; c0 < nChunks; c0 += 4 * (SUB_NT / 64)) {  // four independent loads per step
        int oct[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) oct[u] = k[min((c0 + u * (SUB_NT / 64)) * 64 + lane, n - 1)].octave;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + u * (SUB_NT / 64), i = c * 64 + lane;
            const unsigned long long m = __ballot(i < n && (second ? oct[u] == 0 : !(oct[u] > 0)));
            if (lane == 0 && c < nChunks) mask[c] = m;
        }
    }
    __syncthreads();
    {
        const int v = tid < nChunks ? __popcll(mask[tid]) : 0;
        int total;
        int off = mt_excl_scan(v, total);
        if (lane == 0) waveSum[wave] = total;
        __syncthreads();
        int all = 0;
        for (int w = 0; w < SUB_NT / 64; ++w) { if (w < wave) off += waveSum[w]; all += waveSum[w]; }
        base[tid] = off;
        if (tid == 0) { if (second) n0Out[p] = all; else nL1Out[p] = all; }
    }
    __syncthreads();
    for (int c = wave; c < nChunks; c += SUB_NT / 64) {
        const unsigned long long m = mask[c];
        if ((m >> lane) & 1ull) {
            const int pos = base[c] + __popcll(m & lt);
            if (second) { if (pos < SI_CAP2) orig[pos] = (uint16_t)(c * 64 + lane); }
            else list1[pos] = (uint16_t)(c * 64 + lane);
        }
    }
}

// ---- the candidate pre-pass (k_si_topk) --------------------------------------------------------------------------------------------------
// k_si_sort: the octave-0 subset of F2 of a pair sorted by Frame::PosInGrid cell (column-major, the order GetFeaturesInArea visits the cells
// in; any order inside a cell - the keys carry the subset index), one workgroup per pair: positions, key low bits (cell column << 17 | cell
// row << 11 | subset index), descriptors (two 16-byte halves) and the cell starts.  k_si_topk then only visits the keypoints in the window's
// cells (~11 % of F2 for a window of 100 px in 752 x 480) instead of testing every keypoint of F2 against every keypoint of F1, with
// coalesced loads that depend on nothing (through orig[] -> k2[] -> d2[], a wave per keypoint of F1, it walked a chain of three L2 latencies
// per 64 candidates: 28 us per keypoint, 0.97 of the 1.2 ms a batch of 256 pairs took).
#define SIS_NT 1024
__global__ __launch_bounds__(SIS_NT) void k_si_sort(SearchInitArgs a, SiSorted o)
{
    __shared__ int cnt[GRID_CELLS];
    __shared__ int waveSum[SIS_NT / 64];
    const size_t p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = a.n0G[p];
    if (n0 > SI_CAP2) return;   // (capacity error: the searches report it and never read the lists)
    if (a.n1Dev) { a.k2 += p * a.stride2; a.d2 += p * a.stride2 * 32; }
    const uint16_t *orig = a.origG + p * (size_t)SI_CAP2;
    o.xy += p * (size_t)SI_CAP2; o.code += p * (size_t)SI_CAP2; o.dlo += p * (size_t)SI_CAP2; o.dhi += p * (size_t)SI_CAP2;
    o.cellStart += p * (size_t)SI_CELLS_PAD;
    for (int c = tid; c < GRID_CELLS; c += SIS_NT) cnt[c] = 0;
    __syncthreads();
    constexpr int KPT = SI_CAP2 / SIS_NT;
    int cell[KPT], rank[KPT], i2s[KPT];
    float kxs[KPT], kys[KPT];
#pragma unroll
    for (int s = 0; s < KPT; ++s) {
        const int c2 = tid + s * SIS_NT;
        cell[s] = -1; rank[s] = 0; i2s[s] = 0; kxs[s] = kys[s] = 0.0f;
        if (c2 < n0) {
            const int i2 = orig[c2];
            const float kx = a.k2[i2].x, ky = a.k2[i2].y;
            const int px = (int)roundf(__fmul_rn(__fsub_rn(kx, a.b.minX), a.b.invW));  // Frame::PosInGrid (false outside the grid: never a candidate)
            const int py = (int)roundf(__fmul_rn(__fsub_rn(ky, a.b.minY), a.b.invH));
            i2s[s] = i2; kxs[s] = kx; kys[s] = ky;
            if (px >= 0 && px < GRID_COLS && py >= 0 && py < GRID_ROWS) { cell[s] = px * GRID_ROWS + py; rank[s] = atomicAdd(&cnt[cell[s]], 1); }
        }
    }
    __syncthreads();
    {   // exclusive prefix over the cells: three per thread, wave scan, wave totals through LDS
        constexpr int CPT = GRID_CELLS / SIS_NT;
        int v[CPT], sum = 0;
#pragma unroll
        for (int j = 0; j < CPT; ++j) { v[j] = cnt[tid * CPT + j]; sum += v[j]; }
        int total;
        int off = mt_excl_scan(sum, total);
        if (lane == 0) waveSum[wave] = total;
        __syncthreads();
        for (int w = 0; w < wave; ++w) off += waveSum[w];
#pragma unroll
        for (int j = 0; j < CPT; ++j) { cnt[tid * CPT + j] = off; o.cellStart[tid * CPT + j] = (uint16_t)off; off += v[j]; }
        if (tid == SIS_NT - 1) o.cellStart[GRID_CELLS] = (uint16_t)off;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < KPT; ++s) {
        if (cell[s] < 0) continue;
        const int c2 = tid + s * SIS_NT, idx = cnt[cell[s]] + rank[s];
        const int px = cell[s] / GRID_ROWS, py = cell[s] - px * GRID_ROWS;
        o.xy[idx] = make_float2(kxs[s], kys[s]);
        o.code[idx] = ((uint32_t)px << 17) | ((uint32_t)py << 11) | (uint32_t)c2;
        const uint4 *d = (const uint4 *)(a.d2 + (size_t)i2s[s] * 32);
        o.dlo[idx] = d[0]; o.dhi[idx] = d[1];
    }
}
static_assert(GRID_CELLS % SIS_NT == 0 && SI_CAP2 % SIS_NT == 0, "k_si_sort: cells / keypoints per thread");

// k_si_topk: the SI_TOPK smallest candidate keys of every octave-0 keypoint of F1.  A workgroup holds the sorted subset of F2 of its pair
// in LDS and takes SIT_NT / SIT_LPQ keypoints of F1 per turn: SIT_LPQ neighbouring lanes share a keypoint (lane j of them takes every
// SIT_LPQ-th cell column of the window), every lane keeps the eight smallest keys it saw in registers (a sorted insert: min / max
// pairs), and the lanes of a keypoint merge their lists with eight rounds of "minimum of the heads, its owner pops".
// blockIdx.x: the workgroup's share of the pair's keypoints (turns blockIdx.x, + gridDim.x, ...), blockIdx.y: the pair.
#define SIT_NT 1024
// minimum over the LPQ lanes that share a keypoint (DPP: inside a quad / a row of 16 lanes)
template <int LPQ>
__device__ __forceinline__ uint32_t sit_group_min(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2, 3, 0, 1]
    if constexpr (LPQ == 16) {
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));  // row_half_mirror
        v = min(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));  // row_mirror
    }
    return v;
}
template <int SIT_LPQ>   // 4: batches of pairs; 16: a few pairs (four times the workgroups, a quarter of the candidates per lane)
__global__ __launch_bounds__(SIT_NT) void k_si_topk(SearchInitArgs a, uint32_t *topAll, float *angAll, SiSorted o)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t sit_lds[];
    uint4 *dlo = (uint4 *)sit_lds;                          // [SI_CAP2]
    uint4 *dhi = dlo + SI_CAP2;                             // [SI_CAP2]
    float2 *xy = (float2 *)(dhi + SI_CAP2);                 // [SI_CAP2]
    uint32_t *code = (uint32_t *)(xy + SI_CAP2);            // [SI_CAP2]
    uint16_t *cellStart = (uint16_t *)(code + SI_CAP2);     // [SI_CELLS_PAD]
    static_assert(SIT_LPQ == 4 || SIT_LPQ == 16, "sit_group_min");
    constexpr int SIT_QPW = SIT_NT / SIT_LPQ;
    const int tid = threadIdx.x;
    const size_t p = blockIdx.y;
    const int nL1 = a.nL1G[p];
    if ((int)(blockIdx.x * SIT_QPW) >= nL1) return;   // (uniform)
    const int n0 = a.n0G[p];
    if (n0 > SI_CAP2) return;  // (the searches report the capacity error and never read the lists)
    if (a.n1Dev) {
        a.k1 += p * a.stride1; a.d1 += p * a.stride1 * 32;
        a.prev += p * a.stride1 * 2;
    }
    o.xy += p * (size_t)SI_CAP2; o.code += p * (size_t)SI_CAP2; o.dlo += p * (size_t)SI_CAP2; o.dhi += p * (size_t)SI_CAP2;
    o.cellStart += p * (size_t)SI_CELLS_PAD;
    for (int c = tid; c < GRID_CELLS + 1; c += SIT_NT) cellStart[c] = o.cellStart[c];
    for (int c = tid; c < n0; c += SIT_NT) { dlo[c] = o.dlo[c]; dhi[c] = o.dhi[c]; xy[c] = o.xy[c]; code[c] = o.code[c]; }
    __syncthreads();
    const float r = (float)a.window;
    const int sub = tid & (SIT_LPQ - 1);
    for (int t = blockIdx.x * SIT_QPW + tid / SIT_LPQ; t < nL1; t += gridDim.x * SIT_QPW) {
        const int i1 = a.list1G[p * (size_t)a.cap1 + t];
        const float x = a.initPrev ? a.k1[i1].x : a.prev[2 * i1], y = a.initPrev ? a.k1[i1].y : a.prev[2 * i1 + 1];
        if (sub == 0) angAll[p * (size_t)a.cap1 + t] = a.k1[i1].angle;
        // Frame::GetFeaturesInArea(x, y, r, 0, 0): the cell range, and "no candidates" where the reference returns early
        const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        uint32_t l[SI_TOPK];
#pragma unroll
        for (int k = 0; k < SI_TOPK; ++k) l[k] = SI_KEY_NONE;
        if (!(cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0)) {
            const uint4 q0 = ((const uint4 *)(a.d1 + (size_t)i1 * 32))[0], q1 = ((const uint4 *)(a.d1 + (size_t)i1 * 32))[1];
            int cx = cx0 + sub, idx = 0, end = 0;
            while (true) {
                while (idx >= end && cx <= cx1) {  // next cell column of this lane: the cells (cx, cy0 .. cy1) are neighbours in the sorted order
                    idx = cellStart[cx * GRID_ROWS + cy0]; end = cellStart[cx * GRID_ROWS + cy1 + 1];
                    cx += SIT_LPQ;
                }
                if (idx >= end) break;
                const float2 k2 = xy[idx];
                if (fabsf(__fsub_rn(k2.x, x)) < r && fabsf(__fsub_rn(k2.y, y)) < r) {
                    const uint4 d0 = dlo[idx], d1 = dhi[idx];
                    const int dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                                     __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
                    uint32_t key = ((uint32_t)dist << 23) | code[idx];
                    if (key < l[SI_TOPK - 1]) {
#pragma unroll
                        for (int k = 0; k < SI_TOPK; ++k) { const uint32_t lo = min(l[k], key); key = max(l[k], key); l[k] = lo; }
                    }
                }
                ++idx;
            }
        }
        // merge the lists of the SIT_LPQ lanes (keys are distinct: the subset index is part of them)
        uint32_t *top = topAll + (p * (size_t)a.cap1 + t) * SI_TOPK;
        uint32_t mine = 0, mine2 = 0;   // lane `sub` keeps entry sub (and, four lanes per keypoint, entry sub + 4) and stores it
#pragma unroll
        for (int k = 0; k < SI_TOPK; ++k) {
            const uint32_t mn = sit_group_min<SIT_LPQ>(l[0]);
            if (l[0] == mn && mn != SI_KEY_NONE) {
#pragma unroll
                for (int j = 0; j + 1 < SI_TOPK; ++j) l[j] = l[j + 1];
                l[SI_TOPK - 1] = SI_KEY_NONE;
            }
            if (k == sub) mine = mn;
            if (SIT_LPQ == 4 && k == sub + 4) mine2 = mn;
        }
        if (sub < SI_TOPK) top[sub] = mine;
        if (SIT_LPQ == 4) top[sub + 4] = mine2;
    }
}
static_assert(SI_TOPK == 8, "k_si_topk: list entries per lane");
static const size_t kSitLdsBytes = (size_t)SI_CAP2 * (16 + 16 + 8 + 4) + SI_CELLS_PAD * 2;

// The same search with everything the candidate loop touches in LDS.  Only octave-0 keypoints take part (F1: `level1 > 0` is
// skipped; F2: GetFeaturesInArea(.., level1, level1)), so the kernel compacts both sides to their octave-0 subsets, rebuilds
// the grid over the F2 subset (same PosInGrid cells, same push order) and keeps positions, angles, descriptors, vMatchedDistance
// and vnMatches21 of that subset in LDS: a global round trip per dependent step made the generic kernel above take ~21 us per
// keypoint of F1.  Used when F2 has at most SI_CAP2 octave-0 keypoints and both frames have fewer than 65536 keypoints.

#ifdef SI_TIMING
// debug build (-DSI_TIMING): s_memtime stamps of the last launch's workgroup 0 (start, subsets, grid counted, grid placed, descriptors in
// LDS, walk done, end) and the blocks the walk took eight keypoints at a time / step by step; read back with orbx_debug_si_timing
__device__ unsigned long long g_siTiming[16];
extern "C" int orbx_debug_si_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_siTiming), sizeof(g_siTiming)); }
#define SI_STAMP(i, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_siTiming[i] = (v); } while (0)
#else
#define SI_STAMP(i, v) do { } while (0)
#endif
__global__ __launch_bounds__(64) void k_search_init_lds(SearchInitArgs a)
{
    SI_STAMP(0, __builtin_amdgcn_s_memtime());
    const int TH_LOW = 50, HISTO_LENGTH = 30;  // reference src/ORBmatcher1.cc:37-39
    extern __shared__ __attribute__((aligned(16))) uint8_t si_lds[];
    uint32_t *desc = (uint32_t *)si_lds;                    // [SI_CAP2][8]
    float *kx = (float *)(desc + SI_CAP2 * 8);              // [SI_CAP2]
    float *ky = kx + SI_CAP2;                               // [SI_CAP2]
    float *kang = ky + SI_CAP2;                             // [SI_CAP2]
    int *m21 = (int *)(kang + SI_CAP2);                     // [SI_CAP2] vnMatches21 of the subset (F1 index)
    uint16_t *orig = (uint16_t *)(m21 + SI_CAP2);           // [SI_CAP2] index in F2
    uint16_t *mdist = orig + SI_CAP2;                       // [SI_CAP2] vMatchedDistance (0xFFFF = INT_MAX)
    uint16_t *items0 = mdist + SI_CAP2;                     // [SI_CAP2] subset indices by cell, push order
    uint16_t *cellOf = items0 + SI_CAP2;                    // [SI_CAP2] (setup only)
    uint16_t *cs0 = cellOf + SI_CAP2;                       // [GRID_CELLS + 2] cell starts of the subset grid
    uint16_t *cand = cs0 + GRID_CELLS + 2;                  // [SI_CAND]
    uint16_t *list1 = cand + SI_CAND;                       // [ldsN1 + 64] octave-0 keypoints of F1 in index order (position t)
    uint16_t *m12L = list1 + (a.ldsN1 + 64);                // [ldsN1 + 64] vnMatches12 by position t: subset index of the partner, 0xFFFF none
    signed char *binL = (signed char *)(m12L + (a.ldsN1 + 64));  // [ldsN1 + 64] rotation-histogram bin of the match recorded for t, -1 none
    int *cnt = (int *)desc;                                 // [GRID_CELLS] setup only: aliases the descriptors (loaded last)
    __shared__ int hist[32];
    const int lane = threadIdx.x;
    const unsigned long long lt = (1ull << lane) - 1ull;
    if (a.walkPair && !a.walkPair[blockIdx.x]) return;  // (uniform) k_search_init_rounds did this pair
    if (a.n1Dev) {  // pair p of a batch
        const size_t p = blockIdx.x;
        a.n1 = min(a.n1, a.n1Dev[p]); a.n2 = min(a.n2, a.n2Dev[p]);
        a.k1 += p * a.stride1; a.d1 += p * a.stride1 * 32; a.k2 += p * a.stride2; a.d2 += p * a.stride2 * 32;
        a.prev += p * a.stride1 * 2; a.m12 += p * a.stride1; a.binOf += p * a.stride1; a.nmatches += p;
        if (a.top) {
            a.top += p * (size_t)a.cap1 * SI_TOPK; a.topAng += p * (size_t)a.cap1; a.list1G += p * (size_t)a.cap1;
            a.origG += p * (size_t)SI_CAP2; a.nL1G += p; a.n0G += p;
        }
    }
    if (lane < 32) hist[lane] = 0;
    for (int i = lane; i < a.n1; i += 64) a.m12[i] = -1;
    // (one wave walks these arrays: eight independent loads per step, or every step costs a memory latency)
    if (a.initPrev)
        for (int i0 = 0; i0 < a.n1; i0 += 512) {
            float vx[8], vy[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = min(i0 + 64 * u + lane, a.n1 - 1); vx[u] = a.k1[i].x; vy[u] = a.k1[i].y; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + 64 * u + lane; if (i < a.n1) { a.prev[2 * i] = vx[u]; a.prev[2 * i + 1] = vy[u]; } }
        }
    // ---- octave-0 subsets ------------------------------------------------------------------------------------
    int n0 = 0, nL1 = 0;
    if (a.top) {  // the pre-pass compacted both frames already (k_si_subset): copy
        n0 = a.n0G[0]; nL1 = a.nL1G[0];
        for (int c = lane; c < min(n0, SI_CAP2); c += 64) orig[c] = a.origG[c];
        for (int t = lane; t < nL1; t += 64) list1[t] = a.list1G[t];
    } else {
        for (int ib = 0; ib < a.n2; ib += 64) {
            const int i = ib + lane;
            const bool z = i < a.n2 && a.k2[i].octave == 0;
            const unsigned long long m = __ballot(z);
            const int pos = n0 + __popcll(m & lt);
            if (z && pos < SI_CAP2) orig[pos] = (uint16_t)i;
            n0 += __popcll(m);
        }
    }
    if (n0 > SI_CAP2) {  // (a single call never gets here: the host counted; a pair of a batch reports the capacity error)
        if (lane == 0) *a.nmatches = -1;
        return;
    }
    __threadfence_block();
    if (!a.top) {
        for (int ib = 0; ib < a.n1; ib += 64) {
            const int i = ib + lane;
            const bool z = i < a.n1 && !(a.k1[i].octave > 0);
            const unsigned long long m = __ballot(z);
            if (z) list1[nL1 + __popcll(m & lt)] = (uint16_t)i;
            nL1 += __popcll(m);
        }
    }
    for (int t = lane; t < nL1; t += 64) { m12L[t] = 0xFFFF; binL[t] = -1; }
    SI_STAMP(1, __builtin_amdgcn_s_memtime());
    int *occ = cnt + GRID_CELLS;                            // [GRID_CELLS] setup only: keypoints of the current 64 per cell
    for (int c = lane; c < 2 * GRID_CELLS; c += 64) cnt[c] = 0;
    __syncthreads();
    for (int cb = 0; cb < n0; cb += 256) {  // Frame::PosInGrid of the subset
        float px_[4], py_[4], pa_[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const MtKp *kp = a.k2 + orig[min(cb + 64 * u + lane, n0 - 1)];
            px_[u] = kp->x; py_[u] = kp->y; pa_[u] = kp->angle;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = cb + 64 * u + lane;
            if (c < n0) {
                kx[c] = px_[u]; ky[c] = py_[u]; kang[c] = pa_[u]; m21[c] = -1; mdist[c] = 0xFFFF;
                const int px = (int)roundf(__fmul_rn(__fsub_rn(px_[u], a.b.minX), a.b.invW));
                const int py = (int)roundf(__fmul_rn(__fsub_rn(py_[u], a.b.minY), a.b.invH));
                const int cl = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? 0xFFFF : px * GRID_ROWS + py;
                cellOf[c] = (uint16_t)cl;
                if (cl != 0xFFFF) atomicAdd(&cnt[cl], 1);
            }
        }
    }
    __syncthreads();
    {
        int base = 0;
        for (int cb = 0; cb < GRID_CELLS; cb += 64) {
            int total;
            const int v = cnt[cb + lane];
            const int o = base + mt_excl_scan(v, total);
            cs0[cb + lane] = (uint16_t)o;
            cnt[cb + lane] = o;
            base += total;
        }
        if (lane == 0) cs0[GRID_CELLS] = (uint16_t)base;
    }
    __syncthreads();
    SI_STAMP(2, __builtin_amdgcn_s_memtime());
    for (int ib = 0; ib < n0; ib += 64) {  // stable placement, 64 keypoints per step in index order
        const int c = ib + lane;
        const int cl = c < n0 ? (int)cellOf[c] : 0xFFFF;
        // keypoints that are alone in their cell among these 64 (nearly all of them) go straight to the cell's cursor
        if (cl != 0xFFFF) atomicAdd(&occ[cl], 1);
        mt_wave_sync();
        const bool alone = cl != 0xFFFF && occ[cl] == 1;
        if (alone) { items0[cnt[cl]] = (uint16_t)c; cnt[cl] += 1; }
        mt_wave_sync();
        if (cl != 0xFFFF) occ[cl] = 0;
        unsigned long long todo = __ballot(cl != 0xFFFF && !alone);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int cll = __shfl(cl, leader);
            const unsigned long long same = __ballot(cl == cll);
            if (cl == cll) items0[cnt[cll] + __popcll(same & lt)] = (uint16_t)c;
            mt_wave_sync();
            if (lane == leader) cnt[cll] += __popcll(same);
            mt_wave_sync();
            todo &= ~same;
        }
    }
    mt_wave_sync();
    SI_STAMP(3, __builtin_amdgcn_s_memtime());
    for (int i0 = 0; i0 < n0 * 2; i0 += 512) {  // descriptors, 16 bytes per lane and load, eight loads in flight (over cnt[])
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 64 * u + lane, n0 * 2 - 1);
            v[u] = ((const uint4 *)(a.d2 + (size_t)orig[i >> 1] * 32))[i & 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)  // (keeps the compiler from sinking each load into the predicated block of its LDS write)
            asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 64 * u + lane;
            if (i < n0 * 2) ((uint4 *)desc)[i] = v[u];
        }
    }
    __syncthreads();

    SI_STAMP(4, __builtin_amdgcn_s_memtime());
#ifdef SI_TIMING
    unsigned long long nFast = 0, nSlow = 0;
#endif
    // ---- the sequential walk over the octave-0 keypoints of F1 ---------------------------------------------------
    const float factor = 1.0f / HISTO_LENGTH;
    const float r = (float)a.window;
    // an accepted match: vnMatches21's previous owner loses it (:725-729), the rotation histogram gets its vote (:736-746).
    // The walk only RECORDS it - three LDS stores, nothing read: who lost a keypoint (the holder is the last taker: m21) and the
    // histogram bins (the partner's angle) are worked out behind the walk, for all keypoints at once.  Read inside the walk they were an
    // LDS round trip in every accepted step of a chain that has nothing else to wait for.  (Everything the walk writes stays in LDS; a
    // global store per accepted match would have to be acknowledged before the next block's prefetch can be waited for.)
    auto accept = [&](int t, int bD, int bC, float) {
        if (lane == 0) { m12L[t] = (uint16_t)bC; m21[bC] = t; mdist[bC] = (uint16_t)bD; }
        // One wave: its LDS accesses execute in order, so the next step sees lane 0's writes without a barrier
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // every candidate of one keypoint, in Frame::GetFeaturesInArea(x, y, r, 0, 0)'s order
    auto fullScan = [&](int t, float x, float y, const uint32_t (&q)[8], float ang1) {
        const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        if (cx0 >= GRID_COLS) return;
        const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
        if (cx1 < 0) return;
        const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        if (cy0 >= GRID_ROWS) return;
        const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
        if (cy1 < 0) return;
        const int ny_ = cy1 - cy0 + 1, nCells = (cx1 - cx0 + 1) * ny_;
        int bestD = INT_MAX, best2 = INT_MAX, bestC = -1, bestOrd = INT_MAX;
        auto consider = [&](int c2, int ord) {
            if (!(fabsf(__fsub_rn(kx[c2], x)) < r && fabsf(__fsub_rn(ky[c2], y)) < r)) return;
            const uint32_t *tt = desc + c2 * 8;
            int dist = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ tt[w]);
            if ((int)mdist[c2] <= dist) return;
            if (dist < bestD || (dist == bestD && ord < bestOrd)) { best2 = bestD; bestD = dist; bestC = c2; bestOrd = ord; }
            else if (dist < best2) best2 = dist;
        };
        int ordBase = 0;
        for (int cb = 0; cb < nCells; cb += 64) {  // cells in (ix outer, iy inner) order
            const int ci = cb + lane;
            int sB = 0, eB = 0;
            if (ci < nCells) {
                const int cell = (cx0 + ci / ny_) * GRID_ROWS + cy0 + ci % ny_;
                sB = cs0[cell]; eB = cs0[cell + 1];
            }
            int total;
            const int off = ordBase + mt_excl_scan(eB - sB, total);
            for (int j = sB; j < eB; ++j) {
                const int o = off + (j - sB);
                if (o < SI_CAND) cand[o] = items0[j];
                else consider(items0[j], o);
            }
            ordBase += total;
        }
        const int nCand = min(ordBase, SI_CAND);
        __syncthreads();
        for (int o = lane; o < nCand; o += 64) consider(cand[o], o);
        __syncthreads();
        unsigned key = bestC >= 0 ? ((unsigned)bestD << 20) | (unsigned)min(bestOrd, (1 << 20) - 1) : 0xFFFFFFFFu;
        unsigned kmin = key;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kmin = min(kmin, (unsigned)__shfl_xor((int)kmin, o));
        if (kmin == 0xFFFFFFFFu) return;
        const int wl = __ffsll((long long)__ballot(key == kmin)) - 1;
        const int gBestD = __shfl(bestD, wl), gBestC = __shfl(bestC, wl);
        int second = lane == wl ? best2 : bestD;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) second = min(second, __shfl_xor(second, o));
        if (gBestD <= TH_LOW && (float)gBestD < __fmul_rn((float)second, a.nnratio)) accept(t, gBestD, gBestC, ang1);
    };
    const int topUse = a.top ? min(max(a.topUse, 0), SI_TOPK) : 0;
    if (topUse > 0) {
        // The short way: eight keypoints per block, lane 8 s + k holds the k-th smallest candidate key of the s-th of them (one coalesced
        // load per block, requested a block ahead - a load per step would put a memory latency into every step of the walk).  The
        // first candidate not skipped is the reference's best, the next one its second best; undecided (-> the full scan) only when
        // fewer than two of an incomplete list survive and the ratio test could still fail against a candidate beyond the list.
        const int sub = lane & 7, grp = lane >> 3;
        // four blocks are in flight: a block of eight steps is shorter than the latency of its load
        uint32_t rk[4];
        float ra[4];
        auto fetchBlock = [&](int tb, uint32_t &keys, float &angs) {
            const int t = tb + grp;
            keys = (t < nL1 && sub < topUse) ? a.top[(size_t)t * SI_TOPK + sub] : SI_KEY_NONE;
            angs = t < nL1 ? a.topAng[t] : 0.f;
        };
        // (lane indices below are wave-uniform: v_readlane, not a ds_bpermute round trip per value)
        auto rl = [&](int v, int idx) { return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(idx)); };
#pragma unroll
        for (int j = 0; j < 4; ++j) fetchBlock(8 * j, rk[j], ra[j]);
        for (int tb0 = 0; tb0 < nL1; tb0 += 32) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int tb = tb0 + 8 * j;
            if (tb >= nL1) break;
            const uint32_t bkeys = rk[j];
            const float bangs = ra[j];
            fetchBlock(tb + 32, rk[j], ra[j]);
            const int td = (int)(bkeys >> 23), tc = (int)(bkeys & (SI_CAP2 - 1));
            // vMatchedDistance of the block's 64 candidates in one LDS read; what the block's own matches change is patched in registers
            int md = (int)mdist[tc];
            if (a.blockWalk) {
                // All eight keypoints of the block at once, every group of eight lanes deciding for its keypoint.  What a keypoint sees of
                // the EARLIER ones of its block is only this: a match they accept lowers vMatchedDistance of its partner, which kills
                // the candidate entries of that partner at the same or a larger distance (`vMatchedDistance[i2] <= dist`, :700).
                // Keypoints adjacent in index are neighbours in the image and share candidates, so that happens in most blocks - the
                // block is therefore iterated to ITS fixed point: decide all eight, strike the entries the accepted matches of earlier
                // groups kill, decide again while anything changed (group g is final after g + 1 passes; two or three in practice,
                // against eight dependent steps).  Only a keypoint whose short list does not decide sends the block to the single steps.
                const bool live0 = bkeys != SI_KEY_NONE && !(md <= td);
                const int gl = lane & ~7;
                bool live = live0, acc = false, fail = false;
                int bD = 0, bC = 0;
                bool pAcc = false;
                int pD = -1, pC = -1;  // the decisions of the pass before
                for (int pass = 0; pass < 9; ++pass) {
                    const unsigned long long liveAll = __ballot(live);
                    const unsigned f = (unsigned)(liveAll >> gl) & 0xFFu, f2 = f & (f - 1u);
                    const uint32_t lastKey = (uint32_t)__shfl((int)bkeys, gl + topUse - 1);
                    const uint32_t key0 = (uint32_t)__shfl((int)bkeys, f ? gl + __ffs((int)f) - 1 : gl);
                    const uint32_t key1 = (uint32_t)__shfl((int)bkeys, f2 ? gl + __ffs((int)f2) - 1 : gl);
                    const bool complete = lastKey == SI_KEY_NONE;
                    const int lastD = (int)(lastKey >> 23);
                    bD = (int)(key0 >> 23); bC = (int)(key0 & (SI_CAP2 - 1));
                    int decided = 0, second = INT_MAX;  // as below: 1 no match, 2 test (best, second), 0 the full scan
                    if (f == 0u) decided = complete ? 1 : 0;
                    else if (bD > TH_LOW) decided = 1;
                    else if (f2) { second = (int)(key1 >> 23); decided = 2; }
                    else if (complete) decided = 2;
                    else if (a.nnratio > 0.0f && (float)bD < __fmul_rn((float)lastD, a.nnratio)) { second = lastD; decided = 2; }
                    acc = decided == 2 && (float)bD < __fmul_rn((float)second, a.nnratio);
                    if (__ballot(decided == 0) != 0ull) { fail = true; break; }
                    // (the same accepted matches as in the pass before strike the same entries: done)
                    if (pass > 0 && __ballot(acc != pAcc || (acc && (bC != pC || bD != pD))) == 0ull) break;
                    pAcc = acc; pC = bC; pD = bD;
                    bool nl = live0;  // the entries left alive by the matches the earlier groups accept in this pass
                    for (int sI = 0; sI < 7; ++sI) {
                        if (!__builtin_amdgcn_readlane((int)acc, 8 * sI)) continue;  // (wave-uniform)
                        const int cS = __builtin_amdgcn_readlane(bC, 8 * sI), dS = __builtin_amdgcn_readlane(bD, 8 * sI);
                        if (grp > sI && tc == cS && td >= dS) nl = false;
                    }
                    if (__ballot(nl != live) == 0ull) break;
                    live = nl;
                    if (pass == 8) fail = true;  // (cannot happen: group g is final after g + 1 passes)
                }
                if (!fail) {
                    // every accepted match is recorded; the partner's holder and distance are those of the LAST keypoint of the block that
                    // takes it (what the single steps would leave behind)
                    bool last = acc;
                    for (int sI = 1; sI < 8; ++sI) {
                        if (!__builtin_amdgcn_readlane((int)acc, 8 * sI)) continue;
                        const int cS = __builtin_amdgcn_readlane(bC, 8 * sI);
                        if (grp < sI && bC == cS) last = false;
                    }
                    if (sub == 0 && acc) {
                        const int t = tb + grp;
                        m12L[t] = (uint16_t)bC;
                        if (last) { m21[bC] = t; mdist[bC] = (uint16_t)bD; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
#ifdef SI_TIMING
                    ++nFast;
#endif
                    continue;
                }
#ifdef SI_TIMING
                ++nSlow;
#endif
            }
            for (int sI = 0; sI < 8 && tb + sI < nL1; ++sI) {
                const bool liveK = grp == sI && bkeys != SI_KEY_NONE && !(md <= td);
                const unsigned mlive = (unsigned)((__ballot(liveK) >> (8 * sI)) & 0xFFull);
                const int lastLane = 8 * sI + topUse - 1;
                const bool complete = (uint32_t)rl((int)bkeys, lastLane) == SI_KEY_NONE;
                const int lastD = rl(td, lastLane);
                int decided = 0, fBestD = 0, fBestC = 0, fSecond = INT_MAX;  // decided: 1 no match, 2 test (best, second)
                if (mlive == 0u) decided = complete ? 1 : 0;
                else {
                    const int l0 = 8 * sI + __ffs((int)mlive) - 1;
                    fBestD = rl(td, l0); fBestC = rl(tc, l0);
                    const unsigned m2 = mlive & (mlive - 1u);
                    if (fBestD > TH_LOW) decided = 1;
                    else if (m2) { fSecond = rl(td, 8 * sI + __ffs((int)m2) - 1); decided = 2; }
                    else if (complete) { fSecond = INT_MAX; decided = 2; }
                    else if (a.nnratio > 0.0f && (float)fBestD < __fmul_rn((float)lastD, a.nnratio)) { fSecond = lastD; decided = 2; }  // every later candidate is at least lastD away
                }
                const float ang1 = __int_as_float(rl(__float_as_int(bangs), 8 * sI));
                if (decided == 2) {
                    if ((float)fBestD < __fmul_rn((float)fSecond, a.nnratio)) {
                        accept(tb + sI, fBestD, fBestC, ang1);
                        if (tc == fBestC) md = fBestD;
                    }
                } else if (decided == 0) {
                    const int i1 = list1[tb + sI];
                    uint32_t q[8];
#pragma unroll
                    for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(a.d1 + (size_t)i1 * 32))[w];
                    fullScan(tb + sI, a.prev[2 * i1], a.prev[2 * i1 + 1], q, ang1);
                    md = (int)mdist[tc];
                }
            }
          }
        }
    } else {
        // registers of the NEXT keypoint are requested while the current one is matched
        float nx = 0.f, ny = 0.f, nang = 0.f;
        uint32_t nq[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        auto fetch = [&](int t) {
            if (t < nL1) {
                const int i1 = list1[t];
                nx = a.prev[2 * i1]; ny = a.prev[2 * i1 + 1]; nang = a.k1[i1].angle;
#pragma unroll
                for (int w = 0; w < 8; ++w) nq[w] = ((const uint32_t *)(a.d1 + (size_t)i1 * 32))[w];
            }
        };
        fetch(0);
        for (int t = 0; t < nL1; ++t) {
            const float x = nx, y = ny, ang1 = nang;
            uint32_t q[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) q[w] = nq[w];
            fetch(t + 1);
            fullScan(t, x, y, q, ang1);
        }
    }
    SI_STAMP(5, __builtin_amdgcn_s_memtime());
#ifdef SI_TIMING
    SI_STAMP(7, nFast); SI_STAMP(8, nSlow);
#endif
    __threadfence_block();
    __syncthreads();
    // what the walk left out: the bin of every recorded match (also of one that was stolen later), and the stolen ones
    for (int t = lane; t < nL1; t += 64) {
        const int c2 = m12L[t];
        if (c2 == 0xFFFF) continue;
        if (a.checkOri) {
            float rot = __fsub_rn(a.k1[list1[t]].angle, kang[c2]);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == HISTO_LENGTH) bin = 0;
            binL[t] = (signed char)bin;
        }
        if (m21[c2] != t) m12L[t] = 0xFFFF;  // a later keypoint of F1 took the partner
    }
    __syncthreads();
    if (a.checkOri) {
        // rotHist sizes (:742-746): every match that was recorded votes, also one that was stolen later
        for (int t = lane; t < nL1; t += 64) {
            const int b = binL[t];
            if (b >= 0) atomicAdd(&hist[b], 1);
        }
        __syncthreads();
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima (every lane, uniform)
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        for (int t = lane; t < nL1; t += 64) {
            const int b = binL[t];
            if (b >= 0 && b != ind1 && b != ind2 && b != ind3) m12L[t] = 0xFFFF;
        }
        __syncthreads();
    }
    // write-out: vnMatches12 (already -1 everywhere), the updated vbPrevMatched (:759-762) - stores only, the partner's position is in LDS
    int cntM = 0;
    for (int t = lane; t < nL1; t += 64) {
        const int c2 = m12L[t];
        if (c2 != 0xFFFF) {
            const int i1 = list1[t];
            a.m12[i1] = orig[c2]; a.prev[2 * i1] = kx[c2]; a.prev[2 * i1 + 1] = ky[c2]; ++cntM;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cntM += __shfl_xor(cntM, o);
    // nmatches as the reference counts it (++ per accepted match, -- per stolen or rotation-filtered one) = entries still >= 0
    if (lane == 0) *a.nmatches = cntM;
    SI_STAMP(6, __builtin_amdgcn_s_memtime());
}

// ---- SearchForInitialization of ONE pair as fixed-point rounds over ALL octave-0 keypoints of F1 (round 4) ------------------------------
// The walk of k_search_init_lds is a chain of ~136 dependent steps on one wave (0.2 of the 0.38 ms a single call takes).  What keypoint t
// sees of the earlier ones is only vMatchedDistance: the distance of the LAST earlier keypoint that matched a candidate - and since a
// keypoint can only take a partner at a distance BELOW its vMatchedDistance (`vMatchedDistance[i2] <= dist` skips, :700), that is the
// minimum over the earlier acceptors of that partner.  So every keypoint decides at once against the decisions of the round before -
//   md(c, t) = min { dist(t') : t' < t accepted partner c }   (per-partner lists of acceptors, rebuilt every round),
//   first / second entry of t's short list (k_si_topk: ascending distance, ties in visiting order) with md(c, t) > dist,
// and the rounds repeat until no decision changes: the decision of t depends on decisions of t' < t only, so after k rounds the first k
// keypoints are final - the fixed point IS the sequential result; in practice the chains are a handful of keypoints long.  A keypoint
// whose 8-entry list does not decide (fewer than two live entries of an incomplete list and the ratio test could still fail) scans all
// of F2's octave-0 keypoints for that round, a wave per such keypoint (same keys as k_si_topk, so the same order).  No grid, no
// descriptor copy: the setup of the single-wave kernel (0.15 ms) goes with the walk.  One workgroup of 1024 threads, up to 2048 keypoints.
#define SIR_NT 1024
#define SIR_KPT 2
#define SIR_CAP1 (SIR_NT * SIR_KPT)
__global__ __launch_bounds__(SIR_NT) void k_search_init_rounds(SearchInitArgs a)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;  // reference src/ORBmatcher1.cc:37-39
    __shared__ int head[SI_CAP2];          // per F2 subset keypoint: first acceptor of its list (keypoint position t), -1 none
    __shared__ int lastAcc[SI_CAP2];       // the last (largest t) acceptor: the keypoint that holds the partner in the end (vnMatches21)
    __shared__ int nxt[SIR_CAP1];
    __shared__ uint16_t decC[SIR_CAP1];    // decision of t: partner (subset index), 0xFFFF none
    __shared__ uint8_t decD[SIR_CAP1];     // its distance (<= TH_LOW)
    __shared__ uint16_t scanC[SIR_CAP1];   // result of t's scan of all of F2 in this round
    __shared__ uint8_t scanD[SIR_CAP1];
    __shared__ signed char binL[SIR_CAP1];
    __shared__ uint16_t undecided[SIR_CAP1];
    __shared__ int hist[32];
    __shared__ int nUndecided, changed, cntM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.n1Dev) {  // pair p of a batch (as k_search_init_lds)
        const size_t p = blockIdx.x;
        a.n1 = min(a.n1, a.n1Dev[p]); a.n2 = min(a.n2, a.n2Dev[p]);
        a.k1 += p * a.stride1; a.d1 += p * a.stride1 * 32; a.k2 += p * a.stride2; a.d2 += p * a.stride2 * 32;
        a.prev += p * a.stride1 * 2; a.m12 += p * a.stride1; a.nmatches += p;
        a.top += p * (size_t)a.cap1 * SI_TOPK; a.topAng += p * (size_t)a.cap1; a.list1G += p * (size_t)a.cap1;
        a.origG += p * (size_t)SI_CAP2; a.nL1G += p; a.n0G += p;
        a.sorted.xy += p * (size_t)SI_CAP2; a.sorted.code += p * (size_t)SI_CAP2; a.sorted.dlo += p * (size_t)SI_CAP2;
        a.sorted.dhi += p * (size_t)SI_CAP2; a.sorted.cellStart += p * (size_t)SI_CELLS_PAD;
    }
    const int nL1 = a.nL1G[0], n0 = a.n0G[0];
    if (n0 > SI_CAP2 || nL1 > a.roundsCap) {
        // a single call: the host checks before it takes this kernel; a batch (counts on the device): the pair is left to the walk
        if (tid == 0) { if (a.walkPair) a.walkPair[blockIdx.x] = 1; else *a.nmatches = -1; }
        return;
    }
    if (tid == 0 && a.walkPair) a.walkPair[blockIdx.x] = 0;
    for (int i = tid; i < a.n1; i += SIR_NT) a.m12[i] = -1;
    if (a.initPrev)
        for (int i = tid; i < a.n1; i += SIR_NT) { a.prev[2 * i] = a.k1[i].x; a.prev[2 * i + 1] = a.k1[i].y; }
    // this thread's keypoints t = tid + s * SIR_NT and their short lists
    uint32_t keys[SIR_KPT][SI_TOPK];
#pragma unroll
    for (int s = 0; s < SIR_KPT; ++s) {
        const int t = tid + s * SIR_NT;
#pragma unroll
        for (int e = 0; e < SI_TOPK; ++e) keys[s][e] = t < nL1 ? a.top[(size_t)t * SI_TOPK + e] : SI_KEY_NONE;
        decC[t] = 0xFFFF; decD[t] = 0; binL[t] = -1;
    }
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) { changed = 0; cntM = 0; }
    const float r = (float)a.window;
    const int topUse = min(max(a.topUse, 1), SI_TOPK);
    // vMatchedDistance of partner c as keypoint t sees it under the decisions of the last round
    auto seen = [&](int c, int t) {
        int md = 0x7FFFFFFF;
        for (int j = head[c]; j >= 0; j = nxt[j])
            if (j < t) md = min(md, (int)decD[j]);
        return md;
    };
    for (int round = 0; round <= nL1; ++round) {
        for (int c = tid; c < n0; c += SIR_NT) head[c] = -1;
        if (tid == 0) nUndecided = 0;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < SIR_KPT; ++s) {
            const int t = tid + s * SIR_NT;
            if (t < nL1 && decC[t] != 0xFFFF) nxt[t] = atomicExch(&head[decC[t]], t);
        }
        __syncthreads();
        int newC[SIR_KPT], newD[SIR_KPT];
#pragma unroll
        for (int s = 0; s < SIR_KPT; ++s) {
            const int t = tid + s * SIR_NT;
            newC[s] = 0xFFFF; newD[s] = 0;
            if (t >= nL1) continue;
            // first and second entry of the list that are not skipped (reference :690-712)
            int bD = 0, bC = -1, second = 0x7FFFFFFF, nLive = 0;
            uint32_t lastKey = SI_KEY_NONE;
#pragma unroll
            for (int e = 0; e < SI_TOPK; ++e) {
                const uint32_t key = keys[s][e];
                if (e == topUse - 1) lastKey = key;        // (ORBX_SI_TOPK < 8, tests: the list ends here - most keypoints then take the scan)
                if (e >= topUse || key == SI_KEY_NONE || nLive >= 2) continue;
                const int c = (int)(key & (SI_CAP2 - 1)), d = (int)(key >> 23);
                if (seen(c, t) <= d) continue;
                if (nLive == 0) { bD = d; bC = c; } else second = d;
                ++nLive;
            }
            const bool complete = lastKey == SI_KEY_NONE;
            const int lastD = (int)(lastKey >> 23);
            int decided = 0;  // 1 no match, 2 test (best, second), 0 the scan of all of F2
            if (nLive == 0) decided = complete ? 1 : 0;
            else if (bD > TH_LOW) decided = 1;
            else if (nLive >= 2) decided = 2;
            else if (complete) decided = 2;
            else if (a.nnratio > 0.0f && (float)bD < __fmul_rn((float)lastD, a.nnratio)) { second = lastD; decided = 2; }
            if (decided == 2 && (float)bD < __fmul_rn((float)second, a.nnratio)) { newC[s] = bC; newD[s] = bD; }
            if (decided == 0) { undecided[atomicAdd(&nUndecided, 1)] = (uint16_t)t; newC[s] = 0x10000; }
        }
        __syncthreads();
        // the undecided keypoints: every octave-0 keypoint of F2 in the window, a wave per keypoint (keys as in k_si_topk)
        const int nU = nUndecided;
        for (int u = wave; u < nU; u += SIR_NT / 64) {
            const int t = undecided[u];
            const int i1 = a.list1G[t];
            // (vbPrevMatched is only updated behind the rounds; initPrev: it is F1's keypoint positions, written by this workgroup above)
            const float x = a.initPrev ? a.k1[i1].x : a.prev[2 * i1], y = a.initPrev ? a.k1[i1].y : a.prev[2 * i1 + 1];
            const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
            const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
            const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
            const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
            uint32_t best = SI_KEY_NONE;
            int sec = 0x7FFFFFFF;
            if (!(cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0)) {
                // the window's cells in the sorted subset, a cell column per lane (as k_si_topk; the keys are the same, so is the order)
                const uint4 q0 = ((const uint4 *)(a.d1 + (size_t)i1 * 32))[0], q1 = ((const uint4 *)(a.d1 + (size_t)i1 * 32))[1];
                for (int cx = cx0 + lane; cx <= cx1; cx += 64) {
                    const int end = a.sorted.cellStart[cx * GRID_ROWS + cy1 + 1];
                    for (int idx = a.sorted.cellStart[cx * GRID_ROWS + cy0]; idx < end; ++idx) {
                        const float2 k2 = a.sorted.xy[idx];
                        if (!(fabsf(__fsub_rn(k2.x, x)) < r && fabsf(__fsub_rn(k2.y, y)) < r)) continue;
                        const uint4 d0 = a.sorted.dlo[idx], d1 = a.sorted.dhi[idx];
                        const int dist = __popc(q0.x ^ d0.x) + __popc(q0.y ^ d0.y) + __popc(q0.z ^ d0.z) + __popc(q0.w ^ d0.w) +
                                         __popc(q1.x ^ d1.x) + __popc(q1.y ^ d1.y) + __popc(q1.z ^ d1.z) + __popc(q1.w ^ d1.w);
                        const uint32_t code = a.sorted.code[idx];
                        if (seen((int)(code & (SI_CAP2 - 1)), t) <= dist) continue;
                        const uint32_t key = ((uint32_t)dist << 23) | code;
                        if (key < best) { if (best != SI_KEY_NONE) sec = min(sec, (int)(best >> 23)); best = key; }
                        else sec = min(sec, dist);
                    }
                }
            }
            const uint32_t gBest = mt_wave_min_u32(best);
            // second smallest distance over the wave: the lane that holds the best contributes its own second, the others their best
            int gSec = best == SI_KEY_NONE ? 0x7FFFFFFF : (best == gBest ? sec : (int)(best >> 23));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) gSec = min(gSec, __shfl_xor(gSec, o));
            if (lane == 0) {
                int c = 0xFFFF, d = 0;
                if (gBest != SI_KEY_NONE) {
                    const int bD = (int)(gBest >> 23);
                    if (bD <= TH_LOW && (float)bD < __fmul_rn((float)gSec, a.nnratio)) { c = (int)(gBest & (SI_CAP2 - 1)); d = bD; }
                }
                scanC[t] = (uint16_t)c; scanD[t] = (uint8_t)d;
            }
        }
        __syncthreads();   // every list walk of this round is done: the decisions may change
        bool ch = false;
#pragma unroll
        for (int s = 0; s < SIR_KPT; ++s) {
            const int t = tid + s * SIR_NT;
            if (t >= nL1) continue;
            if (newC[s] == 0x10000) { newC[s] = scanC[t]; newD[s] = scanD[t]; }
            if (decC[t] != (uint16_t)newC[s] || decD[t] != (uint8_t)newD[s]) ch = true;
            decC[t] = (uint16_t)newC[s]; decD[t] = (uint8_t)newD[s];
        }
        if (ch) changed = 1;
        __syncthreads();
        const int any = changed;
        __syncthreads();
        if (!any) break;
        if (tid == 0) changed = 0;   // (read again only behind the next round's barriers)
    }
    // ---- behind the rounds: who holds a partner in the end, rotation histogram, write-out (as k_search_init_lds) ------------------------
    for (int c = tid; c < n0; c += SIR_NT) lastAcc[c] = -1;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SIR_KPT; ++s) {
        const int t = tid + s * SIR_NT;
        if (t < nL1 && decC[t] != 0xFFFF) atomicMax(&lastAcc[decC[t]], t);
    }
    __syncthreads();
    const float factor = 1.0f / HISTO_LENGTH;
    bool kept[SIR_KPT];
#pragma unroll
    for (int s = 0; s < SIR_KPT; ++s) {
        const int t = tid + s * SIR_NT;
        kept[s] = false;
        if (t >= nL1 || decC[t] == 0xFFFF) continue;
        const int c2 = decC[t];
        if (a.checkOri) {   // every recorded match votes (:736-746), also one that a later keypoint took over
            float rot = __fsub_rn(a.topAng[t], a.k2[a.origG[c2]].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, factor));
            if (bin == HISTO_LENGTH) bin = 0;
            binL[t] = (signed char)bin;
            atomicAdd(&hist[bin], 1);
        }
        kept[s] = lastAcc[c2] == t;   // (:725-729: a later keypoint of F1 that takes the partner clears the earlier match)
    }
    __syncthreads();
    if (a.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima (every thread, uniform)
            const int sz = hist[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
#pragma unroll
        for (int s = 0; s < SIR_KPT; ++s) {
            const int t = tid + s * SIR_NT;
            if (t < nL1 && kept[s]) { const int b = binL[t]; if (b != ind1 && b != ind2 && b != ind3) kept[s] = false; }
        }
    }
    int mine = 0;
#pragma unroll
    for (int s = 0; s < SIR_KPT; ++s) {
        const int t = tid + s * SIR_NT;
        if (t < nL1 && kept[s]) {
            const int i1 = a.list1G[t], i2 = a.origG[decC[t]];
            a.m12[i1] = i2; a.prev[2 * i1] = a.k2[i2].x; a.prev[2 * i1 + 1] = a.k2[i2].y;   // (:759-762)
            ++mine;
        }
    }
    // nmatches as the reference counts it (++ per accepted match, -- per stolen or rotation-filtered one) = entries still >= 0
    if (mine) atomicAdd(&cntM, mine);
    __syncthreads();
    if (tid == 0) *a.nmatches = cntM;
}

// true when orbxLaunchSearchInit takes the LDS-resident kernel (which builds its own grid: no orbxLaunchGridBuild needed)
bool orbxSearchInitInLds(int n1, int n2, int level0of2);

static size_t searchInitLdsBytes(int n1)
{
    return (size_t)SI_CAP2 * (32 + 4 * 4 + 2 * 4) + (GRID_CELLS + 2) * 2 + SI_CAND * 2 + ((size_t)n1 + 64) * 5;
}

hipError_t orbxLaunchGridBuild(hipStream_t st, const void *k, int n, float minX, float maxX, float minY, float maxY, int *cellStart,
                               int *items, int *cellOf, int nframes, long long kpStride, const int *nDev)
{
    GridBounds b;
    b.minX = minX; b.minY = minY;
    b.invW = (float)GRID_COLS / (maxX - minX); b.invH = (float)GRID_ROWS / (maxY - minY);
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_grid_build, dim3(nframes), dim3(GB_NT), 0, st, (const MtKp *)k, n, b, cellStart, items, cellOf, kpStride, nDev);
    return hipGetLastError();
}

bool orbxSearchInitInLds(int n1, int n2, int level0of2)
{
    return level0of2 >= 0 && level0of2 <= SI_CAP2 && n1 < 65536 && n2 < 65536 && searchInitLdsBytes(n1) <= 160 * 1024 - 1024;
}

static hipError_t searchInitLdsAttr(size_t lds)
{
    static size_t attrBytes[64] = {};  // per device (the kernel also has 128 B of static LDS: the dynamic part may not be set to the full 160 KB)
    int devId = 0;
    (void)hipGetDevice(&devId);
    size_t &ab = attrBytes[devId & 63];
    if (lds > ab) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_search_init_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ab = lds;
    }
    return hipSuccess;
}

// SearchForInitialization for `npairs` (F1, F2) pairs, one wave (workgroup) each: F1 = pair p of k1 / d1 / prev / m12 (stride1 records),
// F2 = pair p of k2 / d2 (stride2 records), counts from the device arrays.  A pair whose F2 has more than SI_CAP2 octave-0 keypoints gets
// nmatches = -1.
// Work space of the candidate pre-pass (k_si_subset, k_si_topk) for `npairs` pairs of frames with up to cap1 keypoints in F1, in bytes
static size_t siAlign(size_t b) { return (b + 255) & ~(size_t)255; }
size_t orbxSearchInitWorkBytes(int npairs, int cap1)
{
    const size_t P = (size_t)std::max(npairs, 1), C = (size_t)std::max(cap1, 1);
    return siAlign(P * SI_CAP2 * 2) + siAlign(P * C * 2) + 3 * siAlign(P * 4) + siAlign(P * C * SI_TOPK * 4) + siAlign(P * C * 4) +
           siAlign(P * SI_CAP2 * 8) + siAlign(P * SI_CAP2 * 4) + 2 * siAlign(P * SI_CAP2 * 16) + siAlign(P * SI_CELLS_PAD * 2);
}

// ORBX_SI_TOPK: entries of the precomputed candidate lists the walk looks at (0: none - every keypoint takes the full scan; the tests
// run 1 and 2 so that the full scan behind an undecided short list is exercised)
static int searchInitTopUse()
{
    static const int v = [] { const char *e = getenv("ORBX_SI_TOPK"); return e ? std::min(std::max(atoi(e), 0), SI_TOPK) : SI_TOPK; }();
    return v;
}

// ORBX_SI_ROUNDS=0: the single-wave walk (k_search_init_lds) instead of the fixed-point rounds over all keypoints (A/B, tests)
static int searchInitRounds()
{
    static const int v = [] { const char *e = getenv("ORBX_SI_ROUNDS"); return e ? atoi(e) : 1; }();
    return v;
}
// ORBX_SI_ROUNDS_CAP: octave-0 keypoints of F1 up to which a pair takes the rounds (tests: a small value sends pairs of a batch to the walk)
static int searchInitRoundsCap()
{
    static const int v = [] { const char *e = getenv("ORBX_SI_ROUNDS_CAP"); return e ? std::min(std::max(atoi(e), 0), SIR_CAP1) : SIR_CAP1; }();
    return v;
}

// the pre-pass of `npairs` pairs described by `a` into `work`; fills the pre-pass fields of `a`
static void searchInitPrepass(hipStream_t st, SearchInitArgs &a, int npairs, int cap1, void *work)
{
    a.top = nullptr; a.topAng = nullptr; a.list1G = nullptr; a.origG = nullptr; a.nL1G = nullptr; a.n0G = nullptr; a.topUse = 0; a.cap1 = cap1;
    a.walkPair = nullptr; a.roundsCap = searchInitRoundsCap();
    static const int blockWalk = [] { const char *e = getenv("ORBX_SI_BLOCK"); return e ? atoi(e) != 0 : 1; }();  // tests: 0 = step by step
    a.blockWalk = blockWalk;
    const int use = searchInitTopUse();
    if (!work || use <= 0 || cap1 <= 0) return;
    const size_t P = (size_t)npairs, C = (size_t)cap1;
    uint8_t *w = (uint8_t *)work;
    uint16_t *orig = (uint16_t *)w; w += siAlign(P * SI_CAP2 * 2);
    uint16_t *list1 = (uint16_t *)w; w += siAlign(P * C * 2);
    int *n0 = (int *)w; w += siAlign(P * 4);
    int *nL1 = (int *)w; w += siAlign(P * 4);
    uint32_t *top = (uint32_t *)w; w += siAlign(P * C * SI_TOPK * 4);
    float *ang = (float *)w; w += siAlign(P * C * 4);
    a.walkPair = (int *)w; w += siAlign(P * 4);  // (the batch launcher clears the pointer again when it does not run the rounds kernel)
    SiSorted so;
    so.xy = (float2 *)w; w += siAlign(P * SI_CAP2 * 8);
    so.code = (uint32_t *)w; w += siAlign(P * SI_CAP2 * 4);
    so.dlo = (uint4 *)w; w += siAlign(P * SI_CAP2 * 16);
    so.dhi = (uint4 *)w; w += siAlign(P * SI_CAP2 * 16);
    so.cellStart = (uint16_t *)w;
    a.sorted = so;
    hipLaunchKernelGGL(k_si_subset, dim3(npairs, 2), dim3(SUB_NT), 0, st, a, orig, n0, list1, nL1);
    a.list1G = list1; a.origG = orig; a.nL1G = nL1; a.n0G = n0;
    hipLaunchKernelGGL(k_si_sort, dim3(npairs), dim3(SIS_NT), 0, st, a, so);
    // workgroups per pair: enough turns of SIT_NT / LPQ keypoints for cap1 of them, but no more workgroups than keep the device busy - every
    // one of them stages the pair's sorted subset (up to 96 KB) first
    const int lpq = npairs <= 8 ? 16 : 4;
    const int turns = (cap1 * lpq + SIT_NT - 1) / SIT_NT;
    const int perPair = std::max(1, std::min(turns, (1024 + npairs - 1) / npairs));
    {
        static bool attrSet[64] = {};  // per device: more than the default 64 KB of dynamic LDS
        int devId = 0;
        (void)hipGetDevice(&devId);
        if (!attrSet[devId & 63]) {
            if (hipFuncSetAttribute((const void *)k_si_topk<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSitLdsBytes) != hipSuccess ||
                hipFuncSetAttribute((const void *)k_si_topk<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSitLdsBytes) != hipSuccess) {
                a.top = nullptr;   // (no lists: the searches scan; cannot happen on gfx950's 160 KB)
                return;
            }
            attrSet[devId & 63] = true;
        }
    }
    if (lpq == 16) hipLaunchKernelGGL(k_si_topk<16>, dim3(perPair, npairs), dim3(SIT_NT), kSitLdsBytes, st, a, top, ang, so);
    else hipLaunchKernelGGL(k_si_topk<4>, dim3(perPair, npairs), dim3(SIT_NT), kSitLdsBytes, st, a, top, ang, so);
    a.top = top; a.topAng = ang; a.topUse = use;
}

hipError_t orbxLaunchSearchInitBatch(hipStream_t st, int npairs, const void *k1, const uint8_t *d1, long long stride1, const int *n1Dev,
                                     const void *k2, const uint8_t *d2, long long stride2, const int *n2Dev, int cap1, int cap2, float minX,
                                     float maxX, float minY, float maxY, float *prev, int initPrev, int *m12, signed char *binOf, int *nmatches,
                                     int window, float nnratio, int checkOri, void *work)
{
    if (npairs <= 0) return hipSuccess;
    SearchInitArgs a;
    a.k1 = (const MtKp *)k1; a.d1 = d1; a.n1 = cap1; a.k2 = (const MtKp *)k2; a.d2 = d2; a.n2 = cap2;
    a.cellStart = nullptr; a.items = nullptr;
    a.b.minX = minX; a.b.minY = minY;
    a.b.invW = (float)GRID_COLS / (maxX - minX); a.b.invH = (float)GRID_ROWS / (maxY - minY);
    a.prev = prev; a.m12 = m12; a.m21 = nullptr; a.mdist = nullptr; a.binOf = binOf; a.nmatches = nmatches;
    a.window = window; a.nnratio = nnratio; a.checkOri = checkOri;
    a.stride1 = stride1; a.stride2 = stride2; a.n1Dev = n1Dev; a.n2Dev = n2Dev; a.initPrev = initPrev;
    if (cap1 >= 65536 || cap2 >= 65536) return hipErrorInvalidValue;
    const size_t lds = searchInitLdsBytes(cap1);
    if (lds > 160 * 1024 - 1024) return hipErrorInvalidValue;
    const hipError_t e = searchInitLdsAttr(lds);
    if (e != hipSuccess) return e;
    a.ldsN1 = cap1;
    searchInitPrepass(st, a, npairs, cap1, work);
    // the fixed-point rounds, a workgroup per pair; a pair beyond their capacity (more than SIR_CAP1 octave-0 keypoints in F1: known on the
    // device only) marks itself in walkPair and the walk kernel behind takes it - its workgroups for the other pairs leave at once
    if (searchInitRounds() && a.top) hipLaunchKernelGGL(k_search_init_rounds, dim3(npairs), dim3(SIR_NT), 0, st, a);
    else a.walkPair = nullptr;
    hipLaunchKernelGGL(k_search_init_lds, dim3(npairs), dim3(64), lds, st, a);
    return hipGetLastError();
}

hipError_t orbxLaunchSearchInit(hipStream_t st, const void *k1, const uint8_t *d1, int n1, const void *k2, const uint8_t *d2, int n2,
                                const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY, float *prev,
                                int *m12, int *m21, int *mdist, signed char *binOf, int *nmatches, int window, float nnratio,
                                int checkOri, int level0of2, void *work, int level0of1)
{
    SearchInitArgs a;
    a.stride1 = 0; a.stride2 = 0; a.n1Dev = nullptr; a.n2Dev = nullptr; a.initPrev = 0;
    a.top = nullptr; a.topAng = nullptr; a.list1G = nullptr; a.origG = nullptr; a.nL1G = nullptr; a.n0G = nullptr; a.topUse = 0; a.cap1 = n1;
    a.blockWalk = 0; a.walkPair = nullptr; a.roundsCap = searchInitRoundsCap();
    a.k1 = (const MtKp *)k1; a.d1 = d1; a.n1 = n1; a.k2 = (const MtKp *)k2; a.d2 = d2; a.n2 = n2;
    a.cellStart = cellStart; a.items = items;
    a.b.minX = minX; a.b.minY = minY;
    a.b.invW = (float)GRID_COLS / (maxX - minX); a.b.invH = (float)GRID_ROWS / (maxY - minY);
    a.prev = prev; a.m12 = m12; a.m21 = m21; a.mdist = mdist; a.binOf = binOf; a.nmatches = nmatches;
    a.window = window; a.nnratio = nnratio; a.checkOri = checkOri;
    // level0of2 = number of octave-0 keypoints of F2 (counted by the caller on its host copy), < 0: unknown
    // level0of1 = number of octave-0 keypoints of F1 (< 0: unknown, then n1): what the LDS arrays by list position are sized for
    const int l1 = level0of1 >= 0 ? std::min(level0of1, n1) : n1;
    a.ldsN1 = l1;
    const size_t lds = searchInitLdsBytes(l1);
    const int rounds = searchInitRounds();
    if (rounds && work && searchInitTopUse() >= 1 && level0of2 >= 0 && level0of2 <= SI_CAP2 && level0of1 >= 0 && l1 <= searchInitRoundsCap() && n1 < 65536 &&
        n2 < 65536) {
        searchInitPrepass(st, a, 1, n1, work);
        a.walkPair = nullptr;
        hipLaunchKernelGGL(k_search_init_rounds, dim3(1), dim3(SIR_NT), 0, st, a);
    } else if (orbxSearchInitInLds(l1, n2, level0of2) && n1 < 65536) {
        const hipError_t e = searchInitLdsAttr(lds);
        if (e != hipSuccess) return e;
        searchInitPrepass(st, a, 1, n1, work);
        a.walkPair = nullptr;
        hipLaunchKernelGGL(k_search_init_lds, dim3(1), dim3(64), lds, st, a);
    } else {
        hipLaunchKernelGGL(k_search_init, dim3(1), dim3(64), 0, st, a);
    }
    return hipGetLastError();
}

// =====================================================================================================
// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) — reference src/ORBmatcher3.cc:256-467,
// CurrentFrame.Nleft == -1.  The reference walks the last frame's map points in order; point i may not take a current keypoint
// that an EARLIER accepted point with Observations() > 0 took (or that held such a map point before the call).  That is the only
// coupling between points, so the search runs in two parallel steps:
//   k_sbp_candidates  one wave per projected point: Frame::GetFeaturesInArea in the reference's order, the static filters
//                     (levels, mvuRight) and the Hamming distances -> an ordered (index, distance) list per point;
//   k_sbp_resolve     one workgroup iterates to the fixed point of "every point takes the first minimum of its list among the
//                     keypoints no earlier owner blocks; owner(i2) = the earliest accepted taker with observations".  The
//                     choice of point i depends only on points j < i, so after k rounds the first k points are final; the
//                     chains are short in practice (a conflict needs two points that want the same keypoint).
// =====================================================================================================
#define SBP_CAP 512  // candidates kept per projected point (more -> the call reports a capacity error)
static_assert(SBP_CAP == ORBX_SBP_CAP, "candidate buffer layout");

struct SbpProj { float u, v, invz, angle; int octave, flags; };

struct SbpArgs {
    const MtKp *kc; const uint8_t *dc; const float *uRight; const signed char *curState; int nc;
    const int *cellStart, *items;
    GridBounds b; float minX, maxX, minY, maxY;
    const float *scale; float mbf;
    const SbpProj *proj; const uint8_t *dproj; int np;
    float th; int forward, backward, checkOri;
    int mode;        // 0: SearchByProjection(Frame, Frame); 1: SearchByProjection(Frame, vector<MapPoint*>) - then SbpProj holds
                     //    u, v = mTrackProjX / Y, invz = mTrackProjXR, angle = mTrackViewCos, octave = mnTrackScaleLevel
    float nnratio;
    // mode 2: the keyframe-side projection searches (SearchByProjection(KeyFrame, Sim3, ..), SearchByProjection(Frame, KeyFrame, ..)): SbpProj
    // holds u, v, invz = the window radius, angle = the point's keypoint angle, octave = nPredictedLevel, flags; keypoint levels
    // nPredictedLevel - 1 .. nPredictedLevel + levelHi; every accepted point keeps its keypoint from the later ones
    int levelHi;
    float bound;     // a best distance is accepted iff (float)dist <= bound (TH_HIGH = 100 for the frame searches)
    uint32_t *cand;  // [np][SBP_CAP] index | dist << 16 | octave << 25, sorted by (dist, position in GetFeaturesInArea order)
    int *count;      // [np] list length, -1: point not searched
    int *choice;     // [np] work
    int *owner;      // [2][nc] work
    int *assign;     // [nc] out
    int *result;     // [2] out: nmatches, overflow flag
    // batch of frames (blockIdx.y of k_sbp_candidates / blockIdx.x of k_sbp_resolve): per-frame strides in records, counts on the device
    long long kcStride, projStride;
    const int *ncDev, *npDev;
    // CurrentFrame.Nleft != -1 (KB8 fisheye rig): kc / dc = mvKeys then mvKeysRight, nc = nLeft + nRight; a "step" 2 i / 2 i + 1 is the
    // left / right window search of point i (proj[] holds the steps interleaved, dproj one descriptor per point); np = number of steps
    int nLeft;                        // -1: not a fisheye frame
    const int *cellStartR, *itemsR;   // mGridRight over kc + nLeft
    const int *partner;               // [nc] mvLeftToRightMatch / mvRightToLeftMatch as combined indices, -1 none; NULL in mode 0
    int *head, *next;                 // [nc], [2 np] work of k_sbp_resolve_fe: writers of a keypoint as linked lists
    int ncBase;                       // batch of fisheye frames: keypoints of frame p = ncBase (the left capacity) + ncDev[p] right ones
};

// frame p of a batch: every per-frame array advances by its stride, the counts come from the device arrays
__device__ __forceinline__ void sbp_select_frame(SbpArgs &a, size_t p)
{
    if (a.ncDev) a.nc = min(a.nc, a.ncBase + a.ncDev[p]);
    if (a.npDev) a.np = min(a.np, a.nLeft >= 0 ? 2 * a.npDev[p] : a.npDev[p]);  // (fisheye: two steps per point)
    a.kc += p * a.kcStride; a.dc += p * a.kcStride * 32;
    if (a.uRight) a.uRight += p * a.kcStride;
    if (a.curState) a.curState += p * a.kcStride;
    a.cellStart += p * (GRID_CELLS + 1); a.items += p * a.kcStride;
    a.proj += p * a.projStride; a.dproj += p * (a.nLeft >= 0 ? a.projStride / 2 : a.projStride) * 32;  // (fisheye: two steps, one descriptor per point)
    a.cand += p * a.projStride * SBP_CAP; a.count += p * a.projStride; a.choice += p * a.projStride;
    if (a.owner) a.owner += p * 2 * a.kcStride;
    a.assign += p * a.kcStride; a.result += p * 2;
    if (a.cellStartR) { a.cellStartR += p * (GRID_CELLS + 1); a.itemsR += p * a.kcStride; }
    if (a.partner) a.partner += p * a.kcStride;
    if (a.head) { a.head += p * a.kcStride; a.next += p * 2 * a.projStride; }
}

__global__ __launch_bounds__(64) void k_sbp_candidates(SbpArgs a)
{
    sbp_select_frame(a, blockIdx.y);
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= a.np) return;
    const SbpProj P = a.proj[i];
    int *cnt = a.count + i;
    const bool fe = a.nLeft >= 0;
    const int side = fe ? (i & 1) : 0, base = side ? a.nLeft : 0;
    bool search = (P.flags & 1) != 0 && P.octave >= 0;  // (mnTrackScaleLevelR == -1: src/ORBmatcher1.cc:145)
    // the right step of a frame-to-frame search has no test of its own; k_sbp_resolve_fe drops it when the left one was skipped (:315)
    if (a.mode == 0 && side == 0) search = search && !(P.invz < 0) && !(P.u < a.minX || P.u > a.maxX) && !(P.v < a.minY || P.v > a.maxY);
    if (!search) { if (lane == 0) *cnt = -1; return; }
    if (side) { a.kc += a.nLeft; a.dc += (size_t)a.nLeft * 32; a.cellStart = a.cellStartR; a.items = a.itemsR; }
    const int lastOct = P.octave;
    float r;
    int minLevel, maxLevel;
    if (a.mode == 0) {
        r = __fmul_rn(a.th, a.scale[lastOct]);
        if (a.forward) { minLevel = lastOct; maxLevel = -1; }
        else if (a.backward) { minLevel = 0; maxLevel = lastOct; }
        else { minLevel = lastOct - 1; maxLevel = lastOct + 1; }
    } else if (a.mode == 2) {
        r = P.invz;
        minLevel = lastOct - 1; maxLevel = lastOct + a.levelHi;
    } else {
        float rv = (double)P.angle > 0.998 ? 2.5f : 4.0f;  // ORBmatcher::RadiusByViewingCos (src/ORBmatcher1.cc:217-223), a double compare
        if (a.th != 1.0f && side == 0) rv = __fmul_rn(rv, a.th);  // (the right search of a fisheye frame takes no th factor, :147)
        r = __fmul_rn(rv, a.scale[lastOct]);
        minLevel = lastOct - 1; maxLevel = lastOct;
    }
    const bool checkLevels = (minLevel > 0) || (maxLevel >= 0);
    const float x = P.u, y = P.v;
    const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
    const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
    const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
    const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
    if (cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0) { if (lane == 0) *cnt = 0; return; }
    uint32_t q[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(a.dproj + (size_t)(fe ? i >> 1 : i) * 32))[w];
    const float ur = a.mode == 0 ? __fsub_rn(x, __fmul_rn(a.mbf, P.invz)) : P.invz;  // mode 1: mTrackProjXR
    const int ny = cy1 - cy0 + 1, nCells = (cx1 - cx0 + 1) * ny;
    uint32_t *out = a.cand + (size_t)i * SBP_CAP;
    __shared__ int stage[1024];  // keypoints of up to 64 cells in the reference's order, then filtered with ballots
    __shared__ uint32_t kept[SBP_CAP];  // the list in the reference's order; it leaves sorted
    // A keypoint that held a map point with observations before the call is skipped by every point in every round (:338-340, src/
    // ORBmatcher1.cc:96-98): it never enters the list.  (Not so on a fisheye frame, where a later step can hand such a keypoint to a
    // point without observations: k_sbp_resolve_fe looks at the holder of the moment.)
    const bool dropHeld = !fe && a.curState != nullptr;
    int nOut = 0;
    bool overflow = false;
    for (int cb = 0; cb < nCells; cb += 64) {  // cells in (ix outer, iy inner) order
        const int ci = cb + lane;
        int s = 0, e = 0;
        if (ci < nCells) {
            const int cell = (cx0 + ci / ny) * GRID_ROWS + cy0 + ci % ny;
            s = a.cellStart[cell]; e = a.cellStart[cell + 1];
        }
        int total;
        const int off = mt_excl_scan(e - s, total);
        if (total > 1024) { overflow = true; break; }  // wave-uniform
        for (int j = s; j < e; ++j) stage[off + (j - s)] = a.items[j];
        __syncthreads();
        for (int ob = 0; ob < total; ob += 64) {
            const int o = ob + lane;
            bool keep = false;
            int i2 = 0, dist = 0, oct = 0;
            if (o < total) {
                i2 = stage[o];
                const MtKp kp = a.kc[i2];
                oct = kp.octave & 15;
                keep = true;
                if (checkLevels) {
                    if (kp.octave < minLevel) keep = false;
                    if (maxLevel >= 0 && kp.octave > maxLevel) keep = false;
                }
                if (keep) keep = fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r;
                if (keep && a.uRight) {
                    const float urc = a.uRight[i2];
                    if (urc > 0 && fabsf(__fsub_rn(ur, urc)) > r) keep = false;
                }
                if (keep && dropHeld && a.curState[i2] > 0) keep = false;
                if (keep) {
                    const uint32_t *t = (const uint32_t *)(a.dc + (size_t)i2 * 32);
#pragma unroll
                    for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ t[w]);
                }
            }
            const unsigned long long m = __ballot(keep);
            const int pos = nOut + __popcll(m & ((1ull << lane) - 1ull));
            if (keep) {
                if (pos < SBP_CAP) kept[pos] = (uint32_t)(i2 + base) | ((uint32_t)dist << 16) | ((uint32_t)oct << 25);
                else overflow = true;
            }
            nOut += __popcll(m);
        }
        __syncthreads();
    }
    if (__ballot(overflow) || nOut > SBP_CAP) { if (lane == 0) { *cnt = 0; atomicExch(&a.result[1], 1); } return; }
    if (lane == 0) *cnt = nOut;
    // The resolve rounds take, per point, the first minimum of the list among the keypoints nobody blocks (and the second smallest
    // distance): in the order (distance, list position) that is the first (and second) entry not blocked - the sequential rule
    // `dist < best .. else dist < best2` over any order of the candidates gives exactly those two.  So the list leaves SORTED by
    // (distance, position): a round looks at its head and walks on only while entries are blocked.  Rank of an entry = number of
    // smaller keys (distinct: the position is part of the key).
    __syncthreads();
    if (nOut <= 64) {
        // (the usual case) one entry per lane, the keys of the other lanes broadcast one by one (v_readlane with a uniform index)
        const uint32_t e = lane < nOut ? kept[lane] : 0u;
        const uint32_t key = lane < nOut ? (((e >> 16) & 0x1FFu) << 9) | (uint32_t)lane : 0xFFFFFFFFu;
        int rank = 0;
        for (int j = 0; j < nOut; ++j) rank += (uint32_t)__builtin_amdgcn_readlane((int)key, j) < key ? 1 : 0;
        if (lane < nOut) out[rank] = e;
    } else {
        for (int u = lane; u < nOut; u += 64) {
            const uint32_t e = kept[u];
            const uint32_t key = (((e >> 16) & 0x1FFu) << 9) | (uint32_t)u;
            int rank = 0;
            for (int j = 0; j < nOut; ++j) rank += ((((kept[j] >> 16) & 0x1FFu) << 9) | (uint32_t)j) < key ? 1 : 0;
            out[rank] = e;
        }
    }
    // (entries behind the list are never read: a round clamps its head to the list length)
}

// ---- the resolve rounds -------------------------------------------------------------------------------------------------------------
// One workgroup per frame.  A point's choice in a round is a function of the owners (of the round before) of the list entries it LOOKS
// AT: it walks its sorted list and stops at the first entry that settles the outcome - the first one nobody earlier holds (frame
// variant), the second such one (map-point variant: best and second best), or, whatever is blocked, the first entry that cannot
// change the outcome any more: before a best is found, one beyond TH_HIGH (every later best would be rejected, :343 / src/
// ORBmatcher1.cc:131); behind the best of the map-point variant, one so far away that the ratio test (:122-129) passes against it and
// all later ones.  Points without a real match (lists of random descriptors whose keypoints their real partners hold) end there
// instead of walking everything in every round.  What a round touches is in registers and LDS: the first 8 entries of the lists of a
// thread's first two points (frames of up to 2048 points) are loaded once, the next (up to 24) entries of every list and both owner
// arrays live in LDS; two barriers per round.  A single frame's resolve took 80 - 220 us with every point walking its list (in
// GetFeaturesInArea order: always to the end) from global memory in every round; it is 1024 threads of ONE CU, so a round costs its
// dependent instruction chain.  Measured alternatives (tools/experiments/sbp_timing.py): a compacted work list of the points one of whose
// looked-at keypoints changed its owner evaluates 12 x fewer points and is slower (five barriers and a scan of all points per round).
// Owner arrays hold point << 1 (SBP_FREE = nobody).
#define SBP_LDS_NC 12288
#define SBP_FREE 0x7FFFFFFE
#define SBP_HEAD 8   // entries at the head of a point's sorted list that the rounds keep in registers
#ifdef SBP_TIMING
// debug build (-DSBP_TIMING): s_memtime stamps of thread 0 of the last launch (start, setup done, rounds done, end) and the number of
// rounds; read back with orbx_debug_sbp_timing (tools/experiments/sbp_timing.py)
__device__ unsigned long long g_sbpTiming[8];
extern "C" int orbx_debug_sbp_timing(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sbpTiming), sizeof(g_sbpTiming)); }
#define SBP_STAMP(i, v) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sbpTiming[i] = (v); } while (0)
#else
#define SBP_STAMP(i, v) do { } while (0)
#endif

// One step of a point's walk: entry e (sorted position), the owner entry of its keypoint.  Returns false when the walk is over.
struct SbpWalk {
    int best = 256, bestI = -1, best2 = 256, bestL = -1, bestL2 = -1, found = 0;
    bool stop = false;
    // (mode 1: the map-point variant with its second best; 0 and 2: the first free entry decides)
    __device__ __forceinline__ bool done(int mode) const { return found >= 2 || stop || (found == 1 && mode != 1); }
    __device__ __forceinline__ void visit(uint32_t e, int owner, int i, int mode, float nnratio, float bound)
    {
        const int d = (e >> 16) & 0x1FF;
        if (found == 0 && !((float)d <= bound)) { stop = true; return; }
        if (found == 1 && (mode != 1 || (nnratio >= 0.f && !((float)best > __fmul_rn(nnratio, (float)d))))) { stop = true; return; }
        if ((owner >> 1) < i) return;  // taken by an earlier point with observations
        if (found == 0) { best = d; bestL = e >> 25; bestI = e & 0xFFFF; }
        else { best2 = d; bestL2 = e >> 25; }
        ++found;
    }
    __device__ __forceinline__ int choice(int mode, float nnratio, float bound) const
    {
        int ch = (float)best <= bound ? bestI : -1;
        // map-point variant: ratio to the second match, only if best and second are in the same scale level (:122-129)
        if (mode == 1 && ch >= 0 && bestL == bestL2 && (float)best > __fmul_rn(nnratio, (float)best2)) ch = -1;
        return ch;
    }
};

// votes of the accepted points (rotation histogram), CurrentFrame.mvpMapPoints[i2] = the LAST accepted taker, nmatches
template <class GetChoice, class SetChoice>
__device__ __forceinline__ void sbp_finish(const SbpArgs &a, int *hist, int *nAcc, GetChoice getChoice, SetChoice setChoice)
{
    const int HISTO_LENGTH = 30, tid = threadIdx.x;
    for (int i = tid; i < a.np; i += 1024) {
        const int ch = getChoice(i);
        int c = -1;
        if (ch >= 0) {
            atomicMax(&a.assign[ch], i);
            atomicAdd(nAcc, 1);
            c = ch;
            if (a.checkOri) {
                float rot = __fsub_rn(a.proj[i].angle, a.kc[ch].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
                if (bin == HISTO_LENGTH) bin = 0;
                atomicAdd(&hist[bin], 1);
                c = ch | (bin << 16);
            }
        }
        setChoice(i, c);
    }
    __syncthreads();
    if (a.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        __syncthreads();
        for (int i = tid; i < a.np; i += 1024) {
            const int c = getChoice(i);
            if (c < 0) continue;
            const int bin = c >> 16, ch = c & 0xFFFF;
            if (bin != ind1 && bin != ind2 && bin != ind3) { a.assign[ch] = -2; atomicSub(nAcc, 1); }
        }
    }
    __syncthreads();
    if (tid == 0) a.result[0] = *nAcc;
}

__global__ __launch_bounds__(1024) void k_sbp_resolve(SbpArgs a, int ldsNc, int extLen)
{
    extern __shared__ int sbpL[];
    sbp_select_frame(a, blockIdx.x);
    const int tid = threadIdx.x;
    SBP_STAMP(0, __builtin_amdgcn_s_memtime());
    __shared__ int changedF[2], hist[32], nAcc;
    int *const own = sbpL;                                  // [2][ldsNc] (offsets cur / nxt: swapped POINTERS would turn every read into a flat load)
    uint32_t *const ext = (uint32_t *)(sbpL + 2 * ldsNc);   // [np][extLen] entries SBP_HEAD .. SBP_HEAD + extLen - 1 of every list
    int cur = 0, nxt = ldsNc;
    for (int i = tid; i < a.nc; i += 1024) { own[cur + i] = SBP_FREE; own[nxt + i] = SBP_FREE; a.assign[i] = -1; }
    if (tid < 32) hist[tid] = 0;
    if (tid < 2) changedF[tid] = 0;
    if (tid == 0) nAcc = 0;
    // what a round needs of point i: list length, the "has observations" flag, the head of the list
    struct Head { int n, obs; uint32_t e[SBP_HEAD]; };
    auto loadHead = [&](int i, Head &h) {
        h.n = a.count[i]; h.obs = (a.proj[i].flags & 2) != 0;
#pragma unroll
        for (int k = 0; k < SBP_HEAD; ++k) h.e[k] = 0;
        if (h.n > 0) {
            const uint4 *t = (const uint4 *)(a.cand + (size_t)i * SBP_CAP);
            const uint4 t0 = t[0], t1 = t[1];  // (entries behind the list: whatever an earlier call left - masked by k < n below)
            h.e[0] = t0.x; h.e[1] = t0.y; h.e[2] = t0.z; h.e[3] = t0.w; h.e[4] = t1.x; h.e[5] = t1.y; h.e[6] = t1.z; h.e[7] = t1.w;
        }
    };
    if (extLen > 0)
        for (int i = tid; i < a.np; i += 1024) {
            const int n = a.count[i];
            const uint4 *src = (const uint4 *)(a.cand + (size_t)i * SBP_CAP + SBP_HEAD);
            uint4 *dst = (uint4 *)(ext + (size_t)i * extLen);
            for (int c = 0; c < extLen / 4 && SBP_HEAD + 4 * c < n; ++c) dst[c] = src[c];
        }
    const int ncm1 = max(a.nc - 1, 0);
    // the choice of point i against the owners of the round before
    auto choose = [&](int i, const Head &h) {
        if (h.n <= 0) return -1;
        // (opaque copies of the head: everything derived from an entry - index, distance, level - is loop-invariant, and hoisted out of
        // the rounds for two heads it took 80 registers: spilled to scratch at the 128 a 1024-thread workgroup leaves)
        uint32_t he[SBP_HEAD];
#pragma unroll
        for (int k = 0; k < SBP_HEAD; ++k) { he[k] = h.e[k]; asm volatile("" : "+v"(he[k])); }
        int ow[SBP_HEAD];
#pragma unroll
        for (int k = 0; k < SBP_HEAD; ++k) ow[k] = own[cur + min((int)(he[k] & 0xFFFF), ncm1)];  // eight independent reads
        SbpWalk wk;
#pragma unroll
        for (int k = 0; k < SBP_HEAD; ++k)
            if (k < h.n && !wk.done(a.mode)) wk.visit(he[k], ow[k], i, a.mode, a.nnratio, a.bound);
        // not settled by the head: on through the sorted list, four entries per step - in LDS as far as it is there
        if (!wk.done(a.mode) && h.n > SBP_HEAD) {
            const uint32_t *lst = a.cand + (size_t)i * SBP_CAP;
            const uint32_t *extI = ext + (size_t)i * extLen - SBP_HEAD;
            for (int c0 = SBP_HEAD; c0 < h.n && !wk.done(a.mode); c0 += 4) {
                const uint4 q = c0 + 4 <= SBP_HEAD + extLen ? *(const uint4 *)(extI + c0) : *(const uint4 *)(lst + c0);
                const uint32_t e4[4] = {q.x, q.y, q.z, q.w};
                int o4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) o4[k] = own[cur + min((int)(e4[k] & 0xFFFF), ncm1)];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (c0 + k < h.n && !wk.done(a.mode)) wk.visit(e4[k], o4[k], i, a.mode, a.nnratio, a.bound);
            }
        }
        return wk.choice(a.mode, a.nnratio, a.bound);
    };
    // the first two points of a thread live in registers: heads and choices; further ones (frames of more than 2048 points) go through memory
    Head h0, h1;
    h0.n = -1; h1.n = -1; h0.obs = 0; h1.obs = 0;
    if (tid < a.np) loadHead(tid, h0);
    if (tid + 1024 < a.np) loadHead(tid + 1024, h1);
    int ch0 = -1, ch1 = -1;
    __syncthreads();
    SBP_STAMP(1, __builtin_amdgcn_s_memtime());
    // Two barriers per round: choices + atomicMin into the (all free) next array | compare with the current array, which is cleared in
    // the same pass and becomes the next round's next array
    for (int round = 0; round <= a.np + 1; ++round) {
        SBP_STAMP(4, (unsigned long long)round + 1);
        if (tid < a.np) {
            ch0 = choose(tid, h0);
            if (ch0 >= 0 && h0.obs) atomicMin(&own[nxt + ch0], tid << 1);
        }
        if (tid + 1024 < a.np) {
            ch1 = choose(tid + 1024, h1);
            if (ch1 >= 0 && h1.obs) atomicMin(&own[nxt + ch1], (tid + 1024) << 1);
        }
        for (int i = tid + 2048; i < a.np; i += 1024) {
            Head h;
            loadHead(i, h);
            const int ch = choose(i, h);
            a.choice[i] = ch;
            if (ch >= 0 && h.obs) atomicMin(&own[nxt + ch], i << 1);
        }
        __syncthreads();
        bool diff = false;
        for (int i = tid; i < a.nc; i += 1024) { diff |= own[nxt + i] != own[cur + i]; own[cur + i] = SBP_FREE; }
        if (diff) changedF[round & 1] = 1;
        if (tid == 0) changedF[(round + 1) & 1] = 0;
        __syncthreads();
        const int chg = changedF[round & 1];
        const int t = cur; cur = nxt; nxt = t;
        if (!chg) break;
    }
    SBP_STAMP(2, __builtin_amdgcn_s_memtime());
    sbp_finish(a, hist, &nAcc,
               [&](int i) { return i < 1024 ? ch0 : i < 2048 ? ch1 : a.choice[i]; },
               [&](int i, int c) { if (i < 1024) ch0 = c; else if (i < 2048) ch1 = c; else a.choice[i] = c; });
    SBP_STAMP(3, __builtin_amdgcn_s_memtime());
}

// The same rounds without the LDS state, for frames of more than SBP_LDS_NC keypoints: owners in global memory, every point walks its
// sorted list in every round.
__global__ __launch_bounds__(1024) void k_sbp_resolve_walk(SbpArgs a)
{
    sbp_select_frame(a, blockIdx.x);
    const int tid = threadIdx.x;
    __shared__ int changedF[2], hist[32], nAcc;
    int *const own = a.owner;
    int cur = 0, nxt = a.nc;
    for (int i = tid; i < a.nc; i += 1024) { own[cur + i] = SBP_FREE; own[nxt + i] = SBP_FREE; a.assign[i] = -1; }
    if (tid < 32) hist[tid] = 0;
    if (tid < 2) changedF[tid] = 0;
    if (tid == 0) nAcc = 0;
    __syncthreads();
    for (int round = 0; round <= a.np + 1; ++round) {
        for (int i = tid; i < a.np; i += 1024) {
            const int n = a.count[i];
            SbpWalk wk;
            const uint32_t *glist = a.cand + (size_t)i * SBP_CAP;
            for (int c = 0; c < n && !wk.done(a.mode); ++c) {
                const uint32_t e = glist[c];
                wk.visit(e, own[cur + (e & 0xFFFF)], i, a.mode, a.nnratio, a.bound);
            }
            const int ch = n > 0 ? wk.choice(a.mode, a.nnratio, a.bound) : -1;
            a.choice[i] = ch;
            if (ch >= 0 && (a.proj[i].flags & 2)) atomicMin(&own[nxt + ch], i << 1);
        }
        __syncthreads();
        bool diff = false;
        for (int k = tid; k < a.nc; k += 1024) { diff |= own[nxt + k] != own[cur + k]; own[cur + k] = SBP_FREE; }
        if (diff) changedF[round & 1] = 1;
        if (tid == 0) changedF[(round + 1) & 1] = 0;
        __syncthreads();
        const int chg = changedF[round & 1];
        const int t = cur; cur = nxt; nxt = t;
        if (!chg) break;
    }
    sbp_finish(a, hist, &nAcc, [&](int i) { return a.choice[i]; }, [&](int i, int c) { a.choice[i] = c; });
}

// The Nleft != -1 branches of both SearchByProjection functions (src/ORBmatcher3.cc:300-436, src/ORBmatcher1.cc:45-215).  The reference
// walks the points in order and per point first the left, then the right search ("steps" 2 i, 2 i + 1); a step may not take a keypoint whose
// CURRENT map point has Observations() > 0.  In the map-point variant a match is also written to the keypoint's stereo partner without
// that test (:133-137, :198-202), so the holder of a keypoint can change to a point WITHOUT observations and free it again: what step s
// sees is the LAST writer before s.  Same fixed point as k_sbp_resolve - every step recomputes its choice from the writers of the previous
// round until nothing changes; step s depends only on steps < s, so the result is the sequential one - with the writers of a keypoint
// kept as a linked list (head[k] -> node 2 s + {0 direct, 1 partner}).
__global__ __launch_bounds__(1024) void k_sbp_resolve_fe(SbpArgs a)
{
    const int TH_HIGH = 100, HISTO_LENGTH = 30;
    sbp_select_frame(a, blockIdx.x);  // (a single call: strides 0, no device-side counts)
    const int tid = threadIdx.x, S = a.np, nc = a.nc;
    __shared__ int changed, hist[32], nAcc;
    for (int i = tid; i < nc; i += 1024) { a.head[i] = -1; a.assign[i] = -1; }
    for (int i = tid; i < S; i += 1024) a.choice[i] = -1;
    if (tid < 32) hist[tid] = 0;
    if (tid == 0) nAcc = 0;
    __syncthreads();
    for (int round = 0; round <= S + 1; ++round) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int pt = tid; 2 * pt < S; pt += 1024) {
            bool skipRight = false;
            for (int side = 0; side < 2; ++side) {
                const int s = 2 * pt + side;
                const int n = a.count[s];
                int ch = -1;
                if (side == 0 && a.mode == 0 && n <= 0) skipRight = true;  // the left part's `continue`s leave the right part out
                if (n > 0 && !(side && skipRight)) {
                    int best = 256, bestI = -1, best2 = 256, bestL = -1, bestL2 = -1;
                    const uint32_t *lst = a.cand + (size_t)s * SBP_CAP;
                    for (int c = 0; c < n; ++c) {
                        const uint32_t e = lst[c];
                        const int i2 = e & 0xFFFF, dist = (e >> 16) & 0x1FF, lv = e >> 25;
                        int last = -1;  // the last step before s that wrote mvpMapPoints[i2]
                        for (int nd = a.head[i2]; nd >= 0; nd = a.next[nd]) { const int w = nd >> 1; if (w < s && w > last) last = w; }
                        if (last >= 0 ? (a.proj[last].flags & 2) != 0 : (a.curState && a.curState[i2] > 0)) continue;
                        if (dist < best) { best2 = best; bestL2 = bestL; best = dist; bestL = lv; bestI = i2; }
                        else if (dist < best2) { best2 = dist; bestL2 = lv; }
                    }
                    ch = best <= TH_HIGH ? bestI : -1;
                    if (a.mode == 1 && ch >= 0 && bestL == bestL2 && (float)best > __fmul_rn(a.nnratio, (float)best2)) {
                        ch = -1;
                        if (side == 0) skipRight = true;  // src/ORBmatcher1.cc:126-127: `continue` to the next map point
                    }
                }
                if (a.choice[s] != ch) { a.choice[s] = ch; changed = 1; }
            }
        }
        __syncthreads();
        const int chg = changed;
        for (int i = tid; i < nc; i += 1024) a.head[i] = -1;
        __syncthreads();
        if (!chg) break;
        for (int s = tid; s < S; s += 1024) {
            const int ch = a.choice[s];
            if (ch < 0) continue;
            a.next[2 * s] = atomicExch(&a.head[ch], 2 * s);
            const int pk = a.partner ? a.partner[ch] : -1;
            if (pk >= 0) a.next[2 * s + 1] = atomicExch(&a.head[pk], 2 * s + 1);
        }
        __syncthreads();
    }
    // mvpMapPoints[k] = the map point of the last writer; nmatches counts every write
    for (int s = tid; s < S; s += 1024) {
        const int ch = a.choice[s];
        if (ch < 0) continue;
        atomicMax(&a.assign[ch], s >> 1);
        int cntW = 1;
        const int pk = a.partner ? a.partner[ch] : -1;
        if (pk >= 0) { atomicMax(&a.assign[pk], s >> 1); ++cntW; }
        atomicAdd(&nAcc, cntW);
        if (a.checkOri) {
            float rot = __fsub_rn(a.proj[s].angle, a.kc[ch].angle);
            if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
            int bin = (int)roundf(__fmul_rn(rot, 1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            atomicAdd(&hist[bin], 1);
            a.choice[s] = ch | (bin << 16);
        }
    }
    __syncthreads();
    if (a.checkOri) {
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima
            const int c = hist[i];
            if (c > max1) { max3 = max2; max2 = max1; max1 = c; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (c > max2) { max3 = max2; max2 = c; ind3 = ind2; ind2 = i; }
            else if (c > max3) { max3 = c; ind3 = i; }
        }
        if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
        __syncthreads();
        for (int s = tid; s < S; s += 1024) {
            const int c = a.choice[s];
            if (c < 0) continue;
            const int bin = c >> 16, ch = c & 0xFFFF;
            if (bin != ind1 && bin != ind2 && bin != ind3) { a.assign[ch] = -2; atomicSub(&nAcc, 1); }
        }
    }
    __syncthreads();
    if (tid == 0) a.result[0] = nAcc;
}

// ---- a batch of fisheye frames: small device-side preparations ----------------------------------------------------------------------
// steps of point i of frame p: 2 i = its left record, 2 i + 1 = its right record, with the merges searchByProjectionFisheyeImpl does on
// the host for a single call (mode 0: one keypoint of the last frame behind both records; mode 1: one map point)
__global__ void k_fe_steps(const SbpProj *pl, const SbpProj *pr, const int *npDev, int maxNp, int mode, SbpProj *steps)
{
    const size_t p = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= min(maxNp, npDev[p])) return;
    SbpProj L = pl[p * maxNp + i], R = pr[p * maxNp + i];
    if (mode == 0) { R.octave = L.octave; R.angle = L.angle; R.flags = L.flags; R.invz = L.invz; }
    else { if (R.octave == -1) R.flags &= ~1; R.flags = (R.flags & ~2) | (L.flags & 2); }
    steps[(p * maxNp + i) * 2] = L; steps[(p * maxNp + i) * 2 + 1] = R;
}
// stereo partners in combined indices (left i -> i, right j -> capL + j) from mvLeftToRightMatch / mvRightToLeftMatch
__global__ void k_fe_partner(const int *l2r, const int *r2l, const int *nlDev, const int *nrDev, int capL, int capR, int *partner)
{
    const size_t p = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x, capC = capL + capR;
    if (i >= capC) return;
    int v = -1;
    if (i < capL) { if (i < nlDev[p]) { const int r = l2r[p * capL + i]; if (r >= 0 && r < nrDev[p]) v = capL + r; } }
    else { const int j = i - capL; if (j < nrDev[p]) { const int l = r2l[p * capR + j]; if (l >= 0 && l < nlDev[p]) v = l; } }
    partner[p * capC + i] = v;
}
hipError_t orbxLaunchFisheyeSteps(hipStream_t st, const void *pl, const void *pr, const int *npDev, int maxNp, int nframes, int mode, void *steps)
{
    if (nframes <= 0 || maxNp <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fe_steps, dim3((maxNp + 255) / 256, nframes), dim3(256), 0, st, (const SbpProj *)pl, (const SbpProj *)pr, npDev, maxNp, mode, (SbpProj *)steps);
    return hipGetLastError();
}
hipError_t orbxLaunchFisheyePartner(hipStream_t st, const int *l2r, const int *r2l, const int *nlDev, const int *nrDev, int capL, int capR, int nframes,
                                    int *partner)
{
    if (nframes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fe_partner, dim3((capL + capR + 255) / 256, nframes), dim3(256), 0, st, l2r, r2l, nlDev, nrDev, capL, capR, partner);
    return hipGetLastError();
}

hipError_t orbxLaunchSearchByProjectionFisheye(hipStream_t st, const OrbxSbpFisheye &f)
{
    SbpArgs a;
    a.kcStride = 0; a.projStride = 0; a.ncDev = nullptr; a.npDev = nullptr; a.ncBase = 0;
    int nframes = 1;
    if (f.batch) {  // frames of a batch: keypoints [capL left slots | capR right slots] per frame, 2 maxNp steps per frame
        nframes = f.batch->nframes; a.kcStride = f.batch->kcStride; a.projStride = f.batch->projStride;
        a.ncDev = f.batch->ncDev; a.npDev = f.batch->npDev; a.ncBase = f.nLeft;
        if (nframes <= 0) return hipSuccess;
    }
    a.mode = f.mode; a.nnratio = f.nnratio;
    a.kc = (const MtKp *)f.kc; a.dc = f.dc; a.uRight = nullptr; a.curState = f.curState; a.nc = f.nLeft + f.nRight;
    a.cellStart = f.cellStartL; a.items = f.itemsL; a.cellStartR = f.cellStartR; a.itemsR = f.itemsR;
    a.b.minX = f.minX; a.b.minY = f.minY; a.b.invW = (float)GRID_COLS / (f.maxX - f.minX); a.b.invH = (float)GRID_ROWS / (f.maxY - f.minY);
    a.minX = f.minX; a.maxX = f.maxX; a.minY = f.minY; a.maxY = f.maxY;
    a.scale = f.scale; a.mbf = 0.f; a.proj = (const SbpProj *)f.steps; a.dproj = f.dproj; a.np = 2 * f.np;
    a.th = f.th; a.forward = f.forward; a.backward = f.backward; a.checkOri = f.checkOri;
    a.cand = f.cand; a.count = f.count; a.choice = f.choice; a.owner = nullptr; a.assign = f.assign; a.result = f.result;
    a.nLeft = f.nLeft; a.partner = f.partner; a.head = f.head; a.next = f.next; a.levelHi = 0; a.bound = 100.f;
    hipError_t e = hipMemsetAsync(f.result, 0, 2 * sizeof(int) * (size_t)nframes, st);
    if (e != hipSuccess) return e;
    if (f.np > 0) hipLaunchKernelGGL(k_sbp_candidates, dim3(2 * f.np, nframes), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_sbp_resolve_fe, dim3(nframes), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t orbxLaunchSearchByProjection(hipStream_t st, const void *kc, const uint8_t *dc, const float *uRight, const signed char *curState,
                                        int nc, const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY,
                                        const float *scale, float mbf, const void *proj, const uint8_t *dproj, int np, float th, int forward,
                                        int backward, int checkOri, int mode, float nnratio, uint32_t *cand, int *count, int *choice,
                                        int *owner, int *assign, int *result, const OrbxSbpBatch *batch, int levelHi, float bound)
{
    SbpArgs a;
    a.levelHi = levelHi; a.bound = bound;
    a.ncBase = 0; a.cellStartR = nullptr; a.itemsR = nullptr; a.partner = nullptr; a.head = nullptr; a.next = nullptr;
    const int nframes = batch ? batch->nframes : 1;
    a.kcStride = batch ? batch->kcStride : 0; a.projStride = batch ? batch->projStride : 0;
    a.ncDev = batch ? batch->ncDev : nullptr; a.npDev = batch ? batch->npDev : nullptr;
    if (nframes <= 0) return hipSuccess;
    a.mode = mode; a.nnratio = nnratio;
    a.kc = (const MtKp *)kc; a.dc = dc; a.uRight = uRight; a.curState = curState; a.nc = nc;
    a.cellStart = cellStart; a.items = items;
    a.b.minX = minX; a.b.minY = minY; a.b.invW = (float)GRID_COLS / (maxX - minX); a.b.invH = (float)GRID_ROWS / (maxY - minY);
    a.minX = minX; a.maxX = maxX; a.minY = minY; a.maxY = maxY;
    a.scale = scale; a.mbf = mbf; a.proj = (const SbpProj *)proj; a.dproj = dproj; a.np = np;
    a.th = th; a.forward = forward; a.backward = backward; a.checkOri = checkOri;
    a.cand = cand; a.count = count; a.choice = choice; a.owner = owner; a.assign = assign; a.result = result;
    a.nLeft = -1; a.cellStartR = nullptr; a.itemsR = nullptr; a.partner = nullptr; a.head = nullptr; a.next = nullptr;
    hipError_t e = hipMemsetAsync(result, 0, (size_t)nframes * 2 * sizeof(int), st);
    if (e != hipSuccess) return e;
    if (np > 0) hipLaunchKernelGGL(k_sbp_candidates, dim3(np, nframes), dim3(64), 0, st, a);
    static const bool forceGlobal = getenv("ORBX_SBP_RESOLVE_GLOBAL") != nullptr;  // tests: the variant for frames that do not fit the LDS
    // LDS: the two owner arrays, then as many entries of every point's list behind its head as fit (at most 24, in fours; more frames
    // than CUs: half the LDS each, so that two workgroups share a CU)
    const int ldsNc = (a.nc + 3) & ~3;
    const size_t budget = 160 * 1024 - 1024, owners = (size_t)ldsNc * 8, share = nframes > 256 ? budget / 2 - 512 : budget;
    if (a.nc <= SBP_LDS_NC && !forceGlobal) {
        int extLen = np > 0 && share > owners ? (int)std::min<size_t>(24, (share - owners) / ((size_t)np * 4)) & ~3 : 0;
        static const int extEnv = [] { const char *e = getenv("ORBX_SBP_LIST_LDS"); return e ? atoi(e) : -1; }();  // tests: fewer (0: no) list entries in LDS
        if (extEnv >= 0) extLen = std::min(extLen, extEnv & ~3);
        const size_t lds = owners + (size_t)np * extLen * 4 + 16;
        static size_t attrBytes[64] = {};  // per device
        int devId = 0;
        (void)hipGetDevice(&devId);
        if (lds > attrBytes[devId & 63]) {
            const hipError_t e2 = hipFuncSetAttribute((const void *)k_sbp_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)budget);
            if (e2 != hipSuccess) return e2;
            attrBytes[devId & 63] = budget;
        }
        hipLaunchKernelGGL(k_sbp_resolve, dim3(nframes), dim3(1024), lds, st, a, ldsNc, extLen);
    } else {
        hipLaunchKernelGGL(k_sbp_resolve_walk, dim3(nframes), dim3(1024), 0, st, a);
    }
    return hipGetLastError();
}


// the pieces of a packed host transfer: blockIdx.y = piece, 16-byte vectors where both ends are aligned
__global__ __launch_bounds__(256) void k_copy_pieces(OrbxCopyPieces p)
{
    const int pc = blockIdx.y;
    uint8_t *dst = (uint8_t *)p.dst[pc];
    const uint8_t *src = (const uint8_t *)p.src[pc];
    const size_t bytes = p.bytes[pc];
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)dst | (uintptr_t)src) & 15) == 0) {
        const size_t nv = bytes >> 4;
        for (size_t i = t; i < nv; i += nt) ((uint4 *)dst)[i] = ((const uint4 *)src)[i];
        for (size_t i = (nv << 4) + t; i < bytes; i += nt) dst[i] = src[i];
    } else {
        for (size_t i = t; i < bytes; i += nt) dst[i] = src[i];
    }
}
hipError_t orbxLaunchCopyPieces(hipStream_t st, const OrbxCopyPieces &p)
{
    if (p.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_copy_pieces, dim3(16, p.n), dim3(256), 0, st, p);
    return hipGetLastError();
}

// nmatches of a batch: result[2 p] (count) / result[2 p + 1] (candidate-list overflow) -> out[p] = count or -1
__global__ void k_sbp_gather(const int *result, int n, int *out)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n) out[p] = result[2 * p + 1] ? -1 : result[2 * p];
}
hipError_t orbxLaunchGatherResults(hipStream_t st, const int *result, int n, int *out)
{
    if (n > 0) hipLaunchKernelGGL(k_sbp_gather, dim3((n + 255) / 256), dim3(256), 0, st, result, n, out);
    return hipGetLastError();
}

// ---- Frame::UndistortKeyPoints / ComputeImageBounds (reference src/Frame.cc:777-845) ----------------------------------------------
// cv::undistortPoints with R = noArray() and P = mK: per point, in double (f64 is IEEE here and the library is built with
// -ffp-contract=off, so the five fixed-point iterations round exactly like the host's): normalise with the old camera matrix,
// invert the Brown-Conrady distortion, apply the new matrix.  One thread per point; the other keypoint fields are copied.
struct UndistortArgs {
    const MtKp *in;
    MtKp *out;
    int n;
    long long inStride, outStride;  // records between the frames of a batch (blockIdx.y)
    const int *nDev;                // per-frame counts on the device, or NULL
    double K[4], KN[4], k[12];  // fx fy cx cy (old / new), k1 k2 p1 p2 k3 k4 k5 k6 s1 s2 s3 s4
};

__global__ __launch_bounds__(256) void k_undistort(UndistortArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    {
        const size_t p = blockIdx.y;
        a.in += p * a.inStride; a.out += p * a.outStride;
        if (a.nDev) a.n = min(a.n, a.nDev[p]);
    }
    if (i >= a.n) return;
    MtKp kp = a.in[i];
    const double u = (double)kp.x, v = (double)kp.y;
    const double ifx = __ddiv_rn(1., a.K[0]), ify = __ddiv_rn(1., a.K[1]), cx = a.K[2], cy = a.K[3];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    const double *k = a.k;
#pragma nounroll
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = __ddiv_rn(1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2, 1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = a.KN[0] * x + 0. * y + a.KN[2], yy = 0. * x + a.KN[1] * y + a.KN[3], ww = __ddiv_rn(1., 0. * x + 0. * y + 1.);
    kp.x = (float)(xx * ww);
    kp.y = (float)(yy * ww);
    a.out[i] = kp;
}

hipError_t orbxLaunchUndistort(hipStream_t st, const void *in, void *out, int n, const float *k4, const float *dist, int ndist, const float *knew4,
                               int nframes, long long inStride, long long outStride, const int *nDev)
{
    if (n <= 0 || nframes <= 0) return hipSuccess;
    UndistortArgs a;
    a.in = (const MtKp *)in; a.out = (MtKp *)out; a.n = n;
    a.inStride = inStride; a.outStride = outStride; a.nDev = nDev;
    for (int i = 0; i < 4; ++i) { a.K[i] = (double)k4[i]; a.KN[i] = (double)knew4[i]; }
    for (int i = 0; i < 12; ++i) a.k[i] = i < ndist ? (double)dist[i] : 0.;
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256, nframes), dim3(256), 0, st, a);
    return hipGetLastError();
}

// =====================================================================================================================================
// Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:1156-1196) after the brute-force knnMatch: Lowe's ratio, then
// KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:306-375; unproject :116-143, project :67-93, Triangulate
// :393-406) for every good match - SURVEY 8(f) rank 3.  One thread per query keypoint (= left keypoint of the lapping area), pair =
// blockIdx.y.  Arithmetic: every float operation of the reference in source order and un-fused (this file is built with
// -ffp-contract=off), 3-term products summed left to right, transcendental calls as the double function narrowed to float, and the
// JacobiSVD null vector as a fixed one-sided Jacobi iteration in double (cyclic pairs, 12 sweeps) - the same restatement as
// oracle/orb_oracle.cpp (floating-point parity with Eigen / libm is unpinned, see there).
// mvRightToLeftMatch: the reference walks the queries in order and later matches overwrite earlier ones -> atomicMax of the left index.
// =====================================================================================================================================
struct FisheyeArgs {
    const MtKp *kl, *kr;               // keypoints of pair 0 (all of them; the queries start at monoLeft / monoRight)
    long long kpStrideL, kpStrideR;    // records between consecutive pairs
    const int *nlDev, *nrDev, *monoLDev, *monoRDev;  // per-pair counts and lapping offsets on the device, or NULL: nl, nr, monoL, monoR
    int nl, nr, monoL, monoR;
    const int *idx, *dist;             // kNN-2 results [pair][knnStride][2], relative to the lapping ranges
    long long knnStride;
    float cam1[8], cam2[8], precision;
    float R12[9], t12[3];
    float sigma2[ORBX_MAX_LEVELS];
    int *leftToRight, *rightToLeft;    // [pair][outStrideL], [pair][outStrideR]
    float *depth, *p3d;                // [pair][outStrideL], [pair][outStrideL][3]
    long long outStrideL, outStrideR;
    int *nMatches;                     // [pair]
};

// correctly rounded float square root: __fsqrt_rn is 1 ulp off on gfx950 (measured: 0x1.a88fb8p+0 -> 0x1.49adf4p+0 instead of
// 0x1.49adf6p+0); the double root narrowed once is exact (53 >= 2 * 24 + 2 bits)
__device__ __forceinline__ float fe_sqrtf(float x) { return (float)__dsqrt_rn((double)x); }
// Eigen's unrolled 3-term redux: x0 + (x1 + x2) (see the projection head below)
__device__ __forceinline__ float fe_dot3(const float *a, const float *b) { return __fadd_rn(__fmul_rn(a[0], b[0]), __fadd_rn(__fmul_rn(a[1], b[1]), __fmul_rn(a[2], b[2]))); }

__device__ void fe_unproject(const float *P, float precision, float px, float py, float r[3])
{
    const float pwx = __fdiv_rn(__fsub_rn(px, P[2]), P[0]), pwy = __fdiv_rn(__fsub_rn(py, P[3]), P[1]);
    float scale = 1.f;
    float theta_d = fe_sqrtf(__fadd_rn(__fmul_rn(pwx, pwx), __fmul_rn(pwy, pwy)));
    theta_d = fminf(fmaxf((float)(-3.14159265358979323846 / 2.f), theta_d), (float)(3.14159265358979323846 / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = __fmul_rn(theta, theta), theta4 = __fmul_rn(theta2, theta2), theta6 = __fmul_rn(theta4, theta2), theta8 = __fmul_rn(theta4, theta4);
            const float k0 = __fmul_rn(P[4], theta2), k1 = __fmul_rn(P[5], theta4), k2 = __fmul_rn(P[6], theta6), k3 = __fmul_rn(P[7], theta8);
            const float num = __fsub_rn(__fmul_rn(theta, __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(1.f, k0), k1), k2), k3)), theta_d);
            const float den = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(1.f, __fmul_rn(3.f, k0)), __fmul_rn(5.f, k1)), __fmul_rn(7.f, k2)), __fmul_rn(9.f, k3));
            const float theta_fix = __fdiv_rn(num, den);
            theta = __fsub_rn(theta, theta_fix);
            if (fabsf(theta_fix) < precision) break;
        }
        scale = __fdiv_rn(orbx_glibc_tanf(theta), theta_d);   // std::tan(float) = glibc's tanf, restated (include/orbx_libm.h)
    }
    r[0] = __fmul_rn(pwx, scale); r[1] = __fmul_rn(pwy, scale); r[2] = 1.f;
}

__device__ void fe_project(const float *P, const float v[3], float uv[2])
{
    const float x2y2 = __fadd_rn(__fmul_rn(v[0], v[0]), __fmul_rn(v[1], v[1]));
    const float theta = orbx_glibc_atan2f(fe_sqrtf(x2y2), v[2]);   // glibc's atan2f, restated (include/orbx_libm.h)
    const float psi = orbx_glibc_atan2f(v[1], v[0]);
    const float t2 = __fmul_rn(theta, theta), t3 = __fmul_rn(theta, t2), t5 = __fmul_rn(t3, t2), t7 = __fmul_rn(t5, t2), t9 = __fmul_rn(t7, t2);
    const float r = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(theta, __fmul_rn(P[4], t3)), __fmul_rn(P[5], t5)), __fmul_rn(P[6], t7)), __fmul_rn(P[7], t9));
    uv[0] = (float)__dadd_rn(__dmul_rn((double)__fmul_rn(P[0], r), cos((double)psi)), (double)P[2]);
    uv[1] = (float)__dadd_rn(__dmul_rn((double)__fmul_rn(P[1], r), sin((double)psi)), (double)P[3]);
}

// =====================================================================================================================================
// Projection head of the two tracking searches (round 4; include/orbx.h "The projection head"): orbx_projection records from world points
// and a pose, one thread per point.  Every operation in float, un-fused, in the order Eigen / Sophus evaluate it:
//   * a 3-term sum (matrix row times vector, dot, squaredNorm) is Eigen's unrolled redux x0 + (x1 + x2) (Eigen/src/Core/Redux.h,
//     redux_novec_unroller: func(run<Start, Length / 2>, run<Start + Length / 2, Length - Length / 2>));
//   * cross(a, b) = (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0) (Eigen/src/Geometry/OrthoMethods.h);
//   * Sophus SO3 * p = p + w * uv + q.vec x uv with uv = 2 (q.vec x p) (Thirdparty/Sophus/sophus/so3.hpp:358-367), SE3: + translation.
// =====================================================================================================================================
struct ProjPointIn { float pos[3], normal[3], minD, maxD, angle; int octave, flags; };   // orbx_world_point
static_assert(sizeof(ProjPointIn) == 44, "orbx_world_point layout");
__device__ __forceinline__ float pp_sum3(float x0, float x1, float x2) { return __fadd_rn(x0, __fadd_rn(x1, x2)); }
__device__ __forceinline__ void pp_cross(const float *a, const float *b, float *c)
{
    c[0] = __fsub_rn(__fmul_rn(a[1], b[2]), __fmul_rn(a[2], b[1]));
    c[1] = __fsub_rn(__fmul_rn(a[2], b[0]), __fmul_rn(a[0], b[2]));
    c[2] = __fsub_rn(__fmul_rn(a[0], b[1]), __fmul_rn(a[1], b[0]));
}
__device__ void fe_project(const float *P, const float v[3], float uv[2]);
__device__ __forceinline__ void pp_camera_project(int type, const float *P, const float v[3], float uv[2])
{
    if (type == 0) {  // Pinhole::project(Eigen::Vector3f), src/CameraModels/Pinhole.cpp:43-49
        uv[0] = __fadd_rn(__fdiv_rn(__fmul_rn(P[0], v[0]), v[2]), P[2]);
        uv[1] = __fadd_rn(__fdiv_rn(__fmul_rn(P[1], v[1]), v[2]), P[3]);
    } else fe_project(P, v, uv);  // KannalaBrandt8::project, src/CameraModels/KannalaBrandt8.cpp:67-84
}

// MapPoint::PredictScale (src/MapPoint.cc:514-546): ceil(log(ratio) / mfLogScaleFactor) in FLOAT - std::log(float) is glibc's logf
// (include/orbx_libm.h, pinned), a float division, std::ceil(float); the int conversion of a NaN / out-of-range value gives INT_MIN on
// x86-64 (cvttss2si), which the first clamp turns into level 0
__device__ __forceinline__ int pp_predict_scale(float maxD, float dist, float logScaleFactor, int nlevels)
{
    const float ratio = __fdiv_rn(maxD, dist);
    const float lv = ceilf(__fdiv_rn(orbx_glibc_logf(ratio), logScaleFactor));
    int nScale = (lv >= -2147483648.0f && lv < 2147483648.0f) ? (int)lv : INT_MIN;
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}

// one camera of Frame::isInFrustum / isInFrustumChecks (src/Frame.cc:549-599, 1198-1272): R (row-major), t = that camera's pose, twc its centre
struct PpSide { bool in; float u, v, invz, viewCos, PcDist; int level; };
__device__ PpSide pp_frustum_side(const float *R, const float *t, const float *twc, int camType, const float *cam, const OrbxProjectArgs &a,
                                  const ProjPointIn &pt)
{
    PpSide o;
    const float *p = pt.pos;
    float Pc[3];
    for (int k = 0; k < 3; ++k)
        Pc[k] = __fadd_rn(pp_sum3(__fmul_rn(R[3 * k], p[0]), __fmul_rn(R[3 * k + 1], p[1]), __fmul_rn(R[3 * k + 2], p[2])), t[k]);
    o.PcDist = fe_sqrtf(pp_sum3(__fmul_rn(Pc[0], Pc[0]), __fmul_rn(Pc[1], Pc[1]), __fmul_rn(Pc[2], Pc[2])));
    o.invz = __fdiv_rn(1.0f, Pc[2]);
    bool in = !(Pc[2] < 0.0f);
    float w[2];
    pp_camera_project(camType, cam, Pc, w);
    if (w[0] < a.minX || w[0] > a.maxX) in = false;
    if (w[1] < a.minY || w[1] > a.maxY) in = false;
    const float maxDistance = __fmul_rn(1.2f, pt.maxD), minDistance = __fmul_rn(0.8f, pt.minD);
    const float PO[3] = {__fsub_rn(p[0], twc[0]), __fsub_rn(p[1], twc[1]), __fsub_rn(p[2], twc[2])};
    const float dist = fe_sqrtf(pp_sum3(__fmul_rn(PO[0], PO[0]), __fmul_rn(PO[1], PO[1]), __fmul_rn(PO[2], PO[2])));
    if (dist < minDistance || dist > maxDistance) in = false;
    o.viewCos = __fdiv_rn(pp_sum3(__fmul_rn(PO[0], pt.normal[0]), __fmul_rn(PO[1], pt.normal[1]), __fmul_rn(PO[2], pt.normal[2])), dist);
    if (o.viewCos < a.viewCosLimit) in = false;
    o.level = pp_predict_scale(pt.maxD, dist, a.logScaleFactor, a.nlevels);
    o.u = w[0]; o.v = w[1]; o.in = in;
    return o;
}

// x3Dc = Tcw * x3Dw with Tcw a Sophus::SE3f: q = x y z w, t (Thirdparty/Sophus/sophus/so3.hpp:358-367, se3.hpp)
__device__ __forceinline__ void pp_se3_apply(const float *q, const float *t, const float *p, float *x)
{
    float uv[3], c2[3];
    pp_cross(q, p, uv);
    for (int k = 0; k < 3; ++k) uv[k] = __fadd_rn(uv[k], uv[k]);
    pp_cross(q, uv, c2);
    for (int k = 0; k < 3; ++k) x[k] = __fadd_rn(__fadd_rn(__fadd_rn(p[k], __fmul_rn(q[3], uv[k])), c2[k]), t[k]);
}

struct WindowPointOut { float u, v, ur, radius; int level, search; };   // orbx_window_point

__global__ __launch_bounds__(256) void k_project_points(OrbxProjectArgs a)
{
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const int np = a.npDev ? min(a.npDev[f], a.maxNp) : a.np;
    if (i >= np) return;
    const ProjPointIn pt = ((const ProjPointIn *)a.points)[(size_t)f * a.maxNp + i];
    const float *T = a.pose + (size_t)f * a.poseStride;
    SbpProj o;
    o.angle = pt.angle; o.octave = pt.octave; o.flags = pt.flags;
    const float *p = pt.pos;
    if (a.mode == 0) {
        // x3Dc = Tcw * x3Dw (src/ORBmatcher3.cc:283-284)
        float x[3];
        pp_se3_apply(T, T + 4, p, x);
        o.invz = (float)__ddiv_rn(1.0, (double)x[2]);   // const float invzc = 1.0 / x3Dc(2) (:288)
        float w[2];
        pp_camera_project(a.camType, a.cam, x, w);       // (:293)
        o.u = w[0]; o.v = w[1];
    } else if (a.mode == 1) {
        // Frame::isInFrustum, src/Frame.cc:549-599
        const PpSide sd = pp_frustum_side(T, T + 9, T + 12, a.camType, a.cam, a, pt);
        bool in = sd.in;
        if (a.farPoints && sd.PcDist > a.thFar) in = false;   // src/ORBmatcher1.cc:57-58 (mTrackDepth = Pc_dist)
        o.u = sd.u; o.v = sd.v;
        o.invz = __fsub_rn(sd.u, __fmul_rn(a.mbf, sd.invz));   // mTrackProjXR
        o.angle = sd.viewCos; o.octave = sd.level;
        o.flags = (pt.flags & ~1) | ((pt.flags & 1) && in ? 1 : 0);
    } else if (a.mode == 2) {
        // Frame::isInFrustum of a fisheye rig (src/Frame.cc:604-614): isInFrustumChecks for both cameras (:1198-1272).  Pose block: Rcw[9] tcw[3]
        // Ow[3] Rrl[9] trl[3] tlr[3] Rwc[9]; the right camera's pose as the reference forms it per call (:1204-1210), coefficient by coefficient
        const float *Rcw = T, *tcw = T + 9, *Ow = T + 12, *Rrl = T + 15, *trl = T + 24, *tlr = T + 27, *Rwc = T + 30;
        float mR[9], mt[3], twc[3];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c)
                mR[3 * r + c] = pp_sum3(__fmul_rn(Rrl[3 * r], Rcw[c]), __fmul_rn(Rrl[3 * r + 1], Rcw[3 + c]), __fmul_rn(Rrl[3 * r + 2], Rcw[6 + c]));
            mt[r] = __fadd_rn(pp_sum3(__fmul_rn(Rrl[3 * r], tcw[0]), __fmul_rn(Rrl[3 * r + 1], tcw[1]), __fmul_rn(Rrl[3 * r + 2], tcw[2])), trl[r]);
            twc[r] = __fadd_rn(pp_sum3(__fmul_rn(Rwc[3 * r], tlr[0]), __fmul_rn(Rwc[3 * r + 1], tlr[1]), __fmul_rn(Rwc[3 * r + 2], tlr[2])), Ow[r]);
        }
        const PpSide l = pp_frustum_side(Rcw, tcw, Ow, 1, a.cam, a, pt);
        const PpSide r = pp_frustum_side(mR, mt, twc, 1, a.cam2, a, pt);
        const bool live = (pt.flags & 1) && !(a.farPoints && l.PcDist > a.thFar);   // the far-point test reads the LEFT camera's depth for both sides
        o.u = l.u; o.v = l.v; o.invz = l.PcDist; o.angle = l.viewCos; o.octave = l.level;
        o.flags = (pt.flags & ~1) | (live && l.in ? 1 : 0);
        ((SbpProj *)a.out)[(size_t)f * a.maxNp + i] = o;
        o.u = r.u; o.v = r.v; o.invz = r.PcDist; o.angle = r.viewCos; o.octave = r.level;
        o.flags = (pt.flags & ~1) | (live && r.in ? 1 : 0);
        ((SbpProj *)a.out2)[(size_t)f * a.maxNp + i] = o;
        return;
    } else {
        // the head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame): src/ORBmatcher2.cc:469-520,
        // src/ORBmatcher1.cc:459-494, src/ORBmatcher3.cc:492-519.  Pose block: q[4] t[3] Ow[3]
        const float *Ow = T + 7;
        float x[3];
        pp_se3_apply(T, T + 4, p, x);
        bool ok = (pt.flags & 1) != 0;
        if ((a.tests & 1) && x[2] < 0.0f) ok = false;
        const float invz = __fdiv_rn(1.0f, x[2]);
        float w[2];
        pp_camera_project(a.camType, a.cam, x, w);
        if (a.tests & 4) { if (w[0] < a.minX || w[0] > a.maxX || w[1] < a.minY || w[1] > a.maxY) ok = false; }   // Frame: mnMinX .. bounds, inclusive
        else if (!(w[0] >= a.minX && w[0] < a.maxX && w[1] >= a.minY && w[1] < a.maxY)) ok = false;            // KeyFrame::IsInImage
        const float maxDistance = __fmul_rn(1.2f, pt.maxD), minDistance = __fmul_rn(0.8f, pt.minD);
        const float PO[3] = {__fsub_rn(p[0], Ow[0]), __fsub_rn(p[1], Ow[1]), __fsub_rn(p[2], Ow[2])};
        const float dist = fe_sqrtf(pp_sum3(__fmul_rn(PO[0], PO[0]), __fmul_rn(PO[1], PO[1]), __fmul_rn(PO[2], PO[2])));
        if (dist < minDistance || dist > maxDistance) ok = false;
        // `PO.dot(Pn) < 0.5 * dist3D`: the right side is a double product, the float dot product is promoted
        if ((a.tests & 2) &&
            (double)pp_sum3(__fmul_rn(PO[0], pt.normal[0]), __fmul_rn(PO[1], pt.normal[1]), __fmul_rn(PO[2], pt.normal[2])) < __dmul_rn(0.5, (double)dist))
            ok = false;
        const int level = pp_predict_scale(pt.maxD, dist, a.logScaleFactor, a.nlevels);
        WindowPointOut wo;
        wo.u = w[0]; wo.v = w[1]; wo.ur = __fsub_rn(w[0], __fmul_rn(a.mbf, invz)); wo.radius = __fmul_rn(a.th, a.scaleFactors[level]);
        wo.level = level; wo.search = ok ? 1 : 0;
        ((WindowPointOut *)a.out)[(size_t)f * a.maxNp + i] = wo;
        return;
    }
    ((SbpProj *)a.out)[(size_t)f * a.maxNp + i] = o;
}

hipError_t orbxLaunchProjectPoints(hipStream_t st, const OrbxProjectArgs *a)
{
    const int n = a->npDev ? a->maxNp : a->np;
    if (a->nframes <= 0 || n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_project_points, dim3((n + 255) / 256, a->nframes), dim3(256), 0, st, *a);
    return hipGetLastError();
}

__device__ void fe_null_vector4(const float A[16], float v[4])
{
    double U[4][4], V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { U[i][j] = (double)A[i * 4 + j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    alpha = __dadd_rn(alpha, __dmul_rn(U[i][p], U[i][p]));
                    beta = __dadd_rn(beta, __dmul_rn(U[i][q], U[i][q]));
                    gamma = __dadd_rn(gamma, __dmul_rn(U[i][p], U[i][q]));
                }
                if (fabs(gamma) <= 1e-300) continue;
                const double zeta = __ddiv_rn(__dsub_rn(beta, alpha), __dmul_rn(2.0, gamma));
                const double t = __ddiv_rn(zeta >= 0 ? 1.0 : -1.0, __dadd_rn(fabs(zeta), __dsqrt_rn(__dadd_rn(1.0, __dmul_rn(zeta, zeta)))));
                const double c = __ddiv_rn(1.0, __dsqrt_rn(__dadd_rn(1.0, __dmul_rn(t, t)))), s = __dmul_rn(c, t);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double up = U[i][p], uq = U[i][q];
                    U[i][p] = __dsub_rn(__dmul_rn(c, up), __dmul_rn(s, uq)); U[i][q] = __dadd_rn(__dmul_rn(s, up), __dmul_rn(c, uq));
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = __dsub_rn(__dmul_rn(c, vp), __dmul_rn(s, vq)); V[i][q] = __dadd_rn(__dmul_rn(s, vp), __dmul_rn(c, vq));
                }
            }
    }
    int best = 0;
    double bestN = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double n = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) n = __dadd_rn(n, __dmul_rn(U[i][j], U[i][j]));
        if (j == 0 || n < bestN) { bestN = n; best = j; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)(best == 0 ? V[i][0] : best == 1 ? V[i][1] : best == 2 ? V[i][2] : V[i][3]);
}

// (any argument block with cam1[8], cam2[8], precision, R12[9], t12[3]: FisheyeArgs, and the per-pair view of SearchForTriangulation)
template <class Rig>
__device__ float fe_triangulate(const Rig &a, float x1, float y1, float x2, float y2, float sigmaLevel, float unc, float p3D[3])
{
    float r1[3], r2[3];
    fe_unproject(a.cam1, a.precision, x1, y1, r1);
    fe_unproject(a.cam2, a.precision, x2, y2, r2);
    float r21[3];
    for (int i = 0; i < 3; ++i) r21[i] = fe_dot3(a.R12 + 3 * i, r2);
    const float cosPar = __fdiv_rn(fe_dot3(r1, r21), __fmul_rn(fe_sqrtf(fe_dot3(r1, r1)), fe_sqrtf(fe_dot3(r21, r21))));
    if ((double)cosPar > 0.9998) return -1;
    float R21[9], T2[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R21[i * 3 + j] = a.R12[j * 3 + i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T2[i * 4 + j] = R21[i * 3 + j];
        const float nr[3] = {-R21[i * 3], -R21[i * 3 + 1], -R21[i * 3 + 2]};
        T2[i * 4 + 3] = fe_dot3(nr, a.t12);
    }
    const float T1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    float A[16];
    for (int j = 0; j < 4; ++j) {
        A[0 * 4 + j] = __fsub_rn(__fmul_rn(r1[0], T1[2 * 4 + j]), T1[0 * 4 + j]);
        A[1 * 4 + j] = __fsub_rn(__fmul_rn(r1[1], T1[2 * 4 + j]), T1[1 * 4 + j]);
        A[2 * 4 + j] = __fsub_rn(__fmul_rn(r2[0], T2[2 * 4 + j]), T2[0 * 4 + j]);
        A[3 * 4 + j] = __fsub_rn(__fmul_rn(r2[1], T2[2 * 4 + j]), T2[1 * 4 + j]);
    }
    float xh[4];
    fe_null_vector4(A, xh);
    const float x3D[3] = {__fdiv_rn(xh[0], xh[3]), __fdiv_rn(xh[1], xh[3]), __fdiv_rn(xh[2], xh[3])};
    const float z1 = x3D[2];
    if (!(z1 > 0)) return -2;
    const float z2 = __fadd_rn(fe_dot3(R21 + 6, x3D), T2[2 * 4 + 3]);
    if (!(z2 > 0)) return -3;
    float uv1[2];
    fe_project(a.cam1, x3D, uv1);
    const float ex1 = __fsub_rn(uv1[0], x1), ey1 = __fsub_rn(uv1[1], y1);
    if ((double)__fadd_rn(__fmul_rn(ex1, ex1), __fmul_rn(ey1, ey1)) > __dmul_rn(5.991, (double)sigmaLevel)) return -4;
    float x3D2[3];
    for (int i = 0; i < 3; ++i) x3D2[i] = __fadd_rn(fe_dot3(R21 + 3 * i, x3D), T2[i * 4 + 3]);
    float uv2[2];
    fe_project(a.cam2, x3D2, uv2);
    const float ex2 = __fsub_rn(uv2[0], x2), ey2 = __fsub_rn(uv2[1], y2);
    if ((double)__fadd_rn(__fmul_rn(ex2, ex2), __fmul_rn(ey2, ey2)) > __dmul_rn(5.991, (double)unc)) return -5;
    p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
    return z1;
}

__global__ __launch_bounds__(256) void k_fisheye_init(FisheyeArgs a)
{
    const size_t p = blockIdx.y;
    const int nl = a.nlDev ? min(a.nlDev[p], a.nl) : a.nl, nr = a.nrDev ? min(a.nrDev[p], a.nr) : a.nr;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nl) { a.leftToRight[p * a.outStrideL + i] = -1; a.depth[p * a.outStrideL + i] = -1.0f; }
    if (i < nr) a.rightToLeft[p * a.outStrideR + i] = -1;
    if (i == 0) a.nMatches[p] = 0;
}

__global__ __launch_bounds__(64) void k_fisheye_triangulate(FisheyeArgs a)
{
    const size_t p = blockIdx.y;
    const int nl = a.nlDev ? min(a.nlDev[p], a.nl) : a.nl;
    const int monoL = a.monoLDev ? a.monoLDev[p] : a.monoL, monoR = a.monoRDev ? a.monoRDev[p] : a.monoR;
    const int i = blockIdx.x * 64 + threadIdx.x;   // query = left keypoint monoL + i
    if (i >= nl - monoL) return;
    const int *idx = a.idx + (p * a.knnStride + i) * 2, *dist = a.dist + (p * a.knnStride + i) * 2;
    if (idx[1] < 0) return;                                                     // fewer than two train rows
    if (!((double)(float)dist[0] < __dmul_rn((double)(float)dist[1], 0.7))) return;  // Lowe's ratio (:1181)
    const int iL = i + monoL, iR = idx[0] + monoR;
    const MtKp k1 = a.kl[p * a.kpStrideL + iL], k2 = a.kr[p * a.kpStrideR + iR];
    float p3[3];
    const float d = fe_triangulate(a, k1.x, k1.y, k2.x, k2.y, a.sigma2[k1.octave], a.sigma2[k2.octave], p3);
    if (d > 0.0001f) {
        a.leftToRight[p * a.outStrideL + iL] = iR;
        atomicMax(&a.rightToLeft[p * a.outStrideR + iR], iL);
        float *o = a.p3d + (p * a.outStrideL + iL) * 3;
        o[0] = p3[0]; o[1] = p3[1]; o[2] = p3[2];
        a.depth[p * a.outStrideL + iL] = d;
        atomicAdd(&a.nMatches[p], 1);
    }
}

hipError_t orbxLaunchFisheyeTriangulate(hipStream_t st, const OrbxFisheyeArgs *h, int npairs)
{
    if (npairs <= 0) return hipSuccess;
    static_assert(sizeof(FisheyeArgs) == sizeof(OrbxFisheyeArgs), "host / device argument blocks differ");
    const FisheyeArgs &a = *(const FisheyeArgs *)h;
    const int nmax = std::max(a.nl, a.nr);
    if (nmax <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_fisheye_init, dim3((nmax + 255) / 256, npairs), dim3(256), 0, st, a);
    if (a.nl > 0) hipLaunchKernelGGL(k_fisheye_triangulate, dim3((a.nl + 63) / 64, npairs), dim3(64), 0, st, a);
    return hipGetLastError();
}


// ---- ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) ---------------------------------------
// reference src/ORBmatcher1.cc:225-427, F.Nleft == -1: features are only compared inside the vocabulary node they share (the
// FeatureVectors of DBoW2's transform()).  A frame feature that received a map point is skipped by every later keyframe feature
// - but only features of the same node compete, so the nodes are independent: one wave per shared node walks the keyframe
// features of the node in order, its lanes take the frame features of the node (the first 64 keep their descriptor in
// registers), best / second best by two wave reductions of (distance << 16 | position) keys.  Then one workgroup applies the
// rotation-histogram check over all matches.
struct SbowArgs {
    const int *kfNode, *kfStart, *kfFeat; int kfNfv;
    const MtKp *kfKeys; const uint8_t *kfDesc; const signed char *kfValid;
    const int *fNode, *fStart, *fFeat; int fNfv;
    const MtKp *fKeys; const uint8_t *fDesc; int nF;
    float nnratio; int checkOri;
    int mode;                    // 0: SearchByBoW(KeyFrame, Frame); 1: SearchByBoW(KeyFrame, KeyFrame) - src/ORBmatcher2.cc:37-177
    const signed char *fValid;   // mode 1: the second keyframe's map point exists and !isBad()
    int nOut;                    // entries of matches / binOf: nF (mode 0, indexed by frame feature), kf_n (mode 1, by idx1)
    int *matches;  // [nOut] out (-1 preset)
    int *binOf;    // [nOut] work
    int *result;   // [0] nmatches, [1 .. 30] rotation histogram
};

// minimum over the wave (all 64 lanes active, result wave-uniform) without the LDS crossbar: four DPP steps leave each row's minimum
// in its lanes, the four rows are merged as scalars - a ds_bpermute round trip per step (__shfl_xor) made two minima cost ~1000 cycles
// of a dependent loop
__device__ __forceinline__ unsigned mt_wave_min_u32(unsigned v)
{
    int x = (int)v;
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false));  // row_half_mirror
    x = (int)min((unsigned)x, (unsigned)__builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false));  // row_mirror
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane(x, 0), r1 = (unsigned)__builtin_amdgcn_readlane(x, 16),
                   r2 = (unsigned)__builtin_amdgcn_readlane(x, 32), r3 = (unsigned)__builtin_amdgcn_readlane(x, 48);
    return min(min(r0, r1), min(r2, r3));
}

// The walk over a node's keyframe features is sequential (a frame feature taken by one is gone for the next), so nothing inside it may
// wait for memory: the node's frame features (index, angle, taken flag) are staged in LDS and the descriptors of its first 256 in
// registers; the keyframe features are loaded 64 at a time, a lane each (index, validity, angle, descriptor), and handed to the walk
// with v_readlane.  (One dependent global round trip per keyframe feature - its index, then its flag and descriptor, then on a match
// the frame feature's index and both angles - made a node of 200 features take 0.9 ms.)
#define SBOW_LDS_F 2048   // frame features of a node whose index / angle live in LDS (beyond: global reads)
#define SBOW_REG_F 4      // chunks of 64 frame features whose descriptors live in registers
__device__ __forceinline__ void sbow_match_body(const SbowArgs &a, const int j)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;  // reference src/ORBmatcher1.cc:37-39
    extern __shared__ uint8_t taken[];        // per frame feature of this node: already holds a map point
    __shared__ int fReal[SBOW_LDS_F];
    __shared__ float fAng[SBOW_LDS_F];
    const int lane = threadIdx.x;
    const int node = a.kfNode[j];
    int lo = 0, hi = a.fNfv;  // the merge of the two maps only ever pairs equal node ids
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.fNode[mid] < node) lo = mid + 1; else hi = mid; }
    if (lo >= a.fNfv || a.fNode[lo] != node) return;
    const int fs = a.fStart[lo], nFn = a.fStart[lo + 1] - fs;
    uint32_t dF[SBOW_REG_F][8];
#pragma unroll
    for (int c = 0; c < SBOW_REG_F; ++c) {
        const int iF = 64 * c + lane;
        int r = -1;
        if (iF < nFn) r = a.fFeat[fs + iF];
#pragma unroll
        for (int w = 0; w < 8; ++w) dF[c][w] = r >= 0 ? ((const uint32_t *)(a.fDesc + (size_t)r * 32))[w] : 0u;
        if (r >= 0) { fReal[iF] = r; fAng[iF] = a.fKeys[r].angle; taken[iF] = a.mode == 1 && !a.fValid[r] ? 1 : 0; }  // (mode 1: !pMP2 || isBad())
    }
    for (int iF = 64 * SBOW_REG_F + lane; iF < nFn; iF += 64) {
        const int r = a.fFeat[fs + iF];
        if (iF < SBOW_LDS_F) { fReal[iF] = r; fAng[iF] = a.fKeys[r].angle; }
        taken[iF] = a.mode == 1 && !a.fValid[r] ? 1 : 0;
    }
    mt_wave_sync();
    const unsigned NONE = (256u << 16) | 0xFFFFu;
    const float factor = 1.0f / HISTO_LENGTH;
    const int kEnd = a.kfStart[j + 1];
    for (int kb = a.kfStart[j]; kb < kEnd; kb += 64) {
        // this lane's keyframe feature of the block
        int myIdx = -1, myValid = 0;
        float myAng = 0.f;
        uint32_t myQ[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (kb + lane < kEnd) {
            myIdx = a.kfFeat[kb + lane];
            myValid = a.kfValid[myIdx] ? 1 : 0;  // pMP && !pMP->isBad()
            myAng = a.kfKeys[myIdx].angle;
#pragma unroll
            for (int w = 0; w < 8; ++w) myQ[w] = ((const uint32_t *)(a.kfDesc + (size_t)myIdx * 32))[w];
        }
        const int cnt = min(64, kEnd - kb);
        for (int t = 0; t < cnt; ++t) {
            if (!__builtin_amdgcn_readlane(myValid, t)) continue;  // wave-uniform
            uint32_t q[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) q[w] = (uint32_t)__builtin_amdgcn_readlane((int)myQ[w], t);
            unsigned k1 = NONE, k2 = NONE;  // this lane's smallest and second smallest key
#pragma unroll
            for (int c = 0; c < SBOW_REG_F; ++c) {
                const int iF = 64 * c + lane;
                if (64 * c < nFn && iF < nFn && !taken[iF]) {
                    int dist = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ dF[c][w]);
                    const unsigned key = ((unsigned)dist << 16) | (unsigned)iF;
                    if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                }
            }
            for (int c0 = 64 * SBOW_REG_F; c0 < nFn; c0 += 64) {
                const int iF = c0 + lane;
                if (iF < nFn && !taken[iF]) {
                    const int r = iF < SBOW_LDS_F ? fReal[iF] : a.fFeat[fs + iF];
                    const uint32_t *tt = (const uint32_t *)(a.fDesc + (size_t)r * 32);
                    int dist = 0;
#pragma unroll
                    for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ tt[w]);
                    const unsigned key = ((unsigned)dist << 16) | (unsigned)iF;
                    if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key;
                }
            }
            const unsigned best = mt_wave_min_u32(k1);
            const unsigned second = mt_wave_min_u32(k1 == best ? k2 : k1);  // (keys are unique: the position is part of them)
            const int bestDist1 = (int)(best >> 16), bestDist2 = (int)(second >> 16);
            const bool near = a.mode == 0 ? bestDist1 <= TH_LOW : bestDist1 < TH_LOW;  // (:352 `<=`, ORBmatcher2.cc:127 `<`)
            if (near && (float)bestDist1 < __fmul_rn(a.nnratio, (float)bestDist2)) {
                const int pos = (int)(best & 0xFFFFu);
                const int realIdxKF = __builtin_amdgcn_readlane(myIdx, t);
                const float ang1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myAng), t));
                if (lane == 0) {
                    const int bestIdxF = pos < SBOW_LDS_F ? fReal[pos] : a.fFeat[fs + pos];
                    taken[pos] = 1;
                    const int slot = a.mode == 0 ? bestIdxF : realIdxKF;  // vpMapPointMatches[bestIdxF] / vpMatches12[idx1]
                    a.matches[slot] = a.mode == 0 ? realIdxKF : bestIdxF;
                    if (a.checkOri) {
                        float rot = __fsub_rn(ang1, pos < SBOW_LDS_F ? fAng[pos] : a.fKeys[bestIdxF].angle);
                        if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                        int bin = (int)roundf(__fmul_rn(rot, factor));
                        if (bin == HISTO_LENGTH) bin = 0;
                        a.binOf[slot] = bin;
                        atomicAdd(&a.result[1 + bin], 1);
                    }
                    atomicAdd(&a.result[0], 1);
                }
                mt_wave_sync();  // (one wave) the next keyframe feature must see taken[pos]
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_sbow_match(SbowArgs a) { sbow_match_body(a, blockIdx.x); }

// A batch of (keyframe, frame) pairs whose keypoints, descriptors and FeatureVectors lie frame by frame in the extractor's / the vocabulary's
// device arrays (orbx_search_by_bow_frames): blockIdx.y = pair, its two frame indices and FeatureVector sizes from a device table.
struct SbowFramesArgs {
    SbowArgs base;          // arrays of frame 0; matches / binOf / result of pair 0
    const int4 *pairs;      // [npairs] keyframe-side frame, its FeatureVector nodes, frame-side frame, its nodes
    long long cap;          // records per frame (keypoints, FeatureVector entries; fv_start: cap + 1)
};
__device__ __forceinline__ void sbow_select_pair(SbowArgs &a, const SbowFramesArgs &fa, int k)
{
    const int4 e = fa.pairs[k];
    const size_t c = (size_t)fa.cap, f1 = (size_t)e.x, f2 = (size_t)e.z;
    a.kfNode += f1 * c; a.kfStart += f1 * (c + 1); a.kfFeat += f1 * c; a.kfNfv = e.y;
    a.kfKeys += f1 * c; a.kfDesc += f1 * c * 32; a.kfValid += f1 * c;
    a.fNode += f2 * c; a.fStart += f2 * (c + 1); a.fFeat += f2 * c; a.fNfv = e.w;
    a.fKeys += f2 * c; a.fDesc += f2 * c * 32;
    if (a.fValid) a.fValid += f2 * c;
    a.matches += (size_t)k * c; a.binOf += (size_t)k * c; a.result += 32 * (size_t)k;
}
__global__ __launch_bounds__(64) void k_sbow_match_frames(SbowFramesArgs fa)
{
    SbowArgs a = fa.base;
    sbow_select_pair(a, fa, blockIdx.y);
    if ((int)blockIdx.x >= a.kfNfv) return;
    sbow_match_body(a, blockIdx.x);
}

__device__ __forceinline__ void sbow_finish_body(const SbowArgs &a)
{
    const int HISTO_LENGTH = 30;
    if (!a.checkOri) return;
    __shared__ int removed;
    if (threadIdx.x == 0) removed = 0;
    int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima (every thread, uniform)
        const int s = a.result[1 + i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if ((float)max2 < __fmul_rn(0.1f, (float)max1)) { ind2 = -1; ind3 = -1; }
    else if ((float)max3 < __fmul_rn(0.1f, (float)max1)) { ind3 = -1; }
    __syncthreads();
    for (int i = threadIdx.x; i < a.nOut; i += 256) {
        if (a.matches[i] < 0) continue;
        const int bin = a.binOf[i];
        if (bin != ind1 && bin != ind2 && bin != ind3) { a.matches[i] = -1; atomicAdd(&removed, 1); }
    }
    __syncthreads();
    if (threadIdx.x == 0) a.result[0] -= removed;
}
__global__ __launch_bounds__(256) void k_sbow_finish(SbowArgs a) { sbow_finish_body(a); }
__global__ __launch_bounds__(256) void k_sbow_finish_frames(SbowFramesArgs fa)
{
    SbowArgs a = fa.base;
    sbow_select_pair(a, fa, blockIdx.x);
    sbow_finish_body(a);
}

hipError_t orbxLaunchSearchByBoW(hipStream_t st, const int *kfNode, const int *kfStart, const int *kfFeat, int kfNfv, const void *kfKeys,
                                 const uint8_t *kfDesc, const signed char *kfValid, const int *fNode, const int *fStart, const int *fFeat,
                                 int fNfv, const void *fKeys, const uint8_t *fDesc, int nF, int maxNodeF, float nnratio, int checkOri,
                                 int mode, const signed char *fValid, int nOut, int *matches, int *binOf, int *result)
{
    SbowArgs a;
    a.mode = mode; a.fValid = fValid; a.nOut = nOut;
    a.kfNode = kfNode; a.kfStart = kfStart; a.kfFeat = kfFeat; a.kfNfv = kfNfv; a.kfKeys = (const MtKp *)kfKeys; a.kfDesc = kfDesc;
    a.kfValid = kfValid; a.fNode = fNode; a.fStart = fStart; a.fFeat = fFeat; a.fNfv = fNfv; a.fKeys = (const MtKp *)fKeys; a.fDesc = fDesc;
    a.nF = nF; a.nnratio = nnratio; a.checkOri = checkOri; a.matches = matches; a.binOf = binOf; a.result = result;
    hipError_t e = hipMemsetAsync(result, 0, 32 * sizeof(int), st);
    if (e == hipSuccess && nOut > 0) e = hipMemsetAsync(matches, 0xFF, (size_t)nOut * sizeof(int), st);
    if (e != hipSuccess) return e;
    if (kfNfv > 0 && fNfv > 0) {
        const size_t lds = ((size_t)std::max(maxNodeF, 1) + 63) & ~(size_t)63;
        hipLaunchKernelGGL(k_sbow_match, dim3(kfNfv), dim3(64), lds, st, a);
        hipLaunchKernelGGL(k_sbow_finish, dim3(1), dim3(256), 0, st, a);
    }
    return hipGetLastError();
}

// `npairs` searches in two launches: frames of one extractor batch (cap records each) with the device FeatureVectors of the vocabulary's
// last batch transform; pairs[k] = {keyframe-side frame, its FeatureVector nodes, frame-side frame, its nodes} on the device;
// matches [npairs][cap], binOf [npairs][cap] work, result [npairs][32] ([0] = nmatches)
hipError_t orbxLaunchSearchByBoWFrames(hipStream_t st, const int *fvNode, const int *fvStart, const int *fvFeat, const void *keys,
                                       const uint8_t *desc, const signed char *valid, int cap, const void *pairs, int npairs, int maxNfv,
                                       float nnratio, int checkOri, int mode, int *matches, int *binOf, int *result)
{
    if (npairs <= 0) return hipSuccess;
    SbowFramesArgs fa;
    SbowArgs &a = fa.base;
    a.mode = mode; a.fValid = mode == 1 ? valid : nullptr; a.nOut = cap;
    a.kfNode = fvNode; a.kfStart = fvStart; a.kfFeat = fvFeat; a.kfNfv = 0; a.kfKeys = (const MtKp *)keys; a.kfDesc = desc; a.kfValid = valid;
    a.fNode = fvNode; a.fStart = fvStart; a.fFeat = fvFeat; a.fNfv = 0; a.fKeys = (const MtKp *)keys; a.fDesc = desc;
    a.nF = cap; a.nnratio = nnratio; a.checkOri = checkOri; a.matches = matches; a.binOf = binOf; a.result = result;
    fa.pairs = (const int4 *)pairs; fa.cap = cap;
    hipError_t e = hipMemsetAsync(result, 0, (size_t)npairs * 32 * sizeof(int), st);
    if (e == hipSuccess) e = hipMemsetAsync(matches, 0xFF, (size_t)npairs * cap * sizeof(int), st);
    if (e != hipSuccess) return e;
    if (maxNfv > 0) {
        const size_t lds = ((size_t)std::max(cap, 1) + 63) & ~(size_t)63;   // `taken`: a node holds at most every feature of the frame
        if (lds > 32 * 1024) {   // beyond the default dynamic limit next to the kernel's 16 KB of static LDS (cap admits up to 65535)
            static size_t raisedTo = 0;
            if (lds > raisedTo) {
                e = hipFuncSetAttribute((const void *)k_sbow_match_frames, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                if (e != hipSuccess) return e;
                raisedTo = lds;
            }
        }
        hipLaunchKernelGGL(k_sbow_match_frames, dim3(maxNfv, npairs), dim3(64), lds, st, fa);
        hipLaunchKernelGGL(k_sbow_finish_frames, dim3(npairs), dim3(256), 0, st, fa);
    }
    return hipGetLastError();
}


// ---- ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vMatchedPairs, bOnlyStereo, bCoarse) ------------------------
// reference src/ORBmatcher2.cc:179-418 (LocalMapping::CreateNewMapPoints): inside every vocabulary node the keyframes share, a feature
// of pKF1 without a map point takes the feature of pKF2 without a map point at the smallest descriptor distance <= TH_LOW that is far
// enough from the epipole (:305-313) and passes the epipolar constraint (:353) - the LAST one in node order on equal distances
// (`dist > bestDist`, :293, is strict).  Nothing is marked as taken (vbMatched2 is never set in this function), so the features of pKF1
// are independent: one wave per shared node, keyframe-1 features of the node one after the other, lanes over the node's keyframe-2
// features, one wave reduction of (distance << 16 | 0xFFFF - position) keys.  The rotation histogram is k_sbow_finish's.
// Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:107-129) on the host-built F12; KannalaBrandt8::epipolarConstrain
// (KannalaBrandt8.cpp:216-220) = TriangulateMatches(...) > 0.0001 with the device function Frame::ComputeStereoFishEyeMatches uses.
struct SftRig { float cam1[8], cam2[8], precision, R12[9], t12[3]; };
struct SftArgs {
    const int *node1, *start1, *feat1; int nfv1;
    const MtKp *keys1; const uint8_t *desc1; const signed char *hasMp1; const float *uRight1; int nLeft1;
    const int *node2, *start2, *feat2; int nfv2;
    const MtKp *keys2; const uint8_t *desc2; const signed char *hasMp2; const float *uRight2; int nLeft2;
    const float *scale2, *sigma1, *sigma2;   // mvScaleFactors of pKF2, mvLevelSigma2 of pKF1 / pKF2 (device arrays)
    float ep[2], F[9];
    int model, rig, onlyStereo, coarse, checkOri;
    float camL[8], camR[8], precision, R[4][9], t[4][3];
    int *matches, *binOf, *result;
};

__device__ __forceinline__ void sft_match_body(const SftArgs &a)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    const int j = blockIdx.x, lane = threadIdx.x;
    const int node = a.node1[j];
    int lo = 0, hi = a.nfv2;  // the merge of the two maps only ever pairs equal node ids
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a.node2[mid] < node) lo = mid + 1; else hi = mid; }
    if (lo >= a.nfv2 || a.node2[lo] != node) return;
    const int s2 = a.start2[lo], n2 = a.start2[lo + 1] - s2;
    const bool rig = a.rig != 0;
    const float factor = 1.0f / HISTO_LENGTH;
    const unsigned NONE = 0xFFFFFFFFu;
    // a wave per feature of the node (blockIdx.y): the features of pKF1 do not depend on each other (no vbMatched1), and one wave
    // walking a node's ~20 features paid six dependent memory round trips for each of them (115 us per call)
    for (int i1 = a.start1[j] + blockIdx.y; i1 < a.start1[j + 1]; i1 += gridDim.y) {
        const int idx1 = a.feat1[i1];
        if (a.hasMp1[idx1]) continue;
        const bool bStereo1 = !rig && a.uRight1 && a.uRight1[idx1] >= 0.f;
        if (a.onlyStereo && !bStereo1) continue;
        const MtKp kp1 = a.keys1[idx1];
        const bool bRight1 = !(a.nLeft1 == -1 || idx1 < a.nLeft1);
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(a.desc1 + (size_t)idx1 * 32))[w];
        // epipolar line of kp1 in the second image (Pinhole.cpp:115-117)
        const float la = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[0]), __fmul_rn(kp1.y, a.F[3])), a.F[6]);
        const float lb = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[1]), __fmul_rn(kp1.y, a.F[4])), a.F[7]);
        const float lc = __fadd_rn(__fadd_rn(__fmul_rn(kp1.x, a.F[2]), __fmul_rn(kp1.y, a.F[5])), a.F[8]);
        const float den = __fadd_rn(__fmul_rn(la, la), __fmul_rn(lb, lb));
        const float sigmaLevel = a.sigma1[kp1.octave];
        unsigned key = NONE;
        for (int c0 = 0; c0 < n2; c0 += 64) {
            const int p2 = c0 + lane;
            if (p2 >= n2) continue;
            const int idx2 = a.feat2[s2 + p2];
            if (a.hasMp2[idx2]) continue;
            const bool bStereo2 = !rig && a.uRight2 && a.uRight2[idx2] >= 0.f;
            if (a.onlyStereo && !bStereo2) continue;
            const uint32_t *t = (const uint32_t *)(a.desc2 + (size_t)idx2 * 32);
            int dist = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ t[w]);
            if (dist > TH_LOW) continue;
            const MtKp kp2 = a.keys2[idx2];
            if (!bStereo1 && !bStereo2 && !rig) {
                const float dex = __fsub_rn(a.ep[0], kp2.x), dey = __fsub_rn(a.ep[1], kp2.y);
                if (__fadd_rn(__fmul_rn(dex, dex), __fmul_rn(dey, dey)) < __fmul_rn(100.f, a.scale2[kp2.octave])) continue;
            }
            bool ok = a.coarse != 0;
            if (!ok) {
                const float unc = a.sigma2[kp2.octave];
                if (a.model == 0) {
                    const float num = __fadd_rn(__fadd_rn(__fmul_rn(la, kp2.x), __fmul_rn(lb, kp2.y)), lc);
                    if (den == 0.f) ok = false;
                    else ok = (double)__fdiv_rn(__fmul_rn(num, num), den) < __dmul_rn(3.84, (double)unc);
                } else {
                    const bool bRight2 = !(a.nLeft2 == -1 || idx2 < a.nLeft2);
                    const int k = rig ? 2 * (int)bRight1 + (int)bRight2 : 0;
                    SftRig r;
                    for (int i = 0; i < 8; ++i) { r.cam1[i] = rig && bRight1 ? a.camR[i] : a.camL[i]; r.cam2[i] = rig && bRight2 ? a.camR[i] : a.camL[i]; }
                    r.precision = a.precision;
                    for (int i = 0; i < 9; ++i) r.R12[i] = a.R[k][i];
                    for (int i = 0; i < 3; ++i) r.t12[i] = a.t[k][i];
                    float p3D[3];
                    ok = fe_triangulate(r, kp1.x, kp1.y, kp2.x, kp2.y, sigmaLevel, unc, p3D) > 0.0001f;
                }
            }
            if (ok) key = min(key, ((unsigned)dist << 16) | (unsigned)(0xFFFF - p2));
        }
        const unsigned best = mt_wave_min_u32(key);
        if (best != NONE && lane == 0) {
            const int bestIdx2 = a.feat2[s2 + (0xFFFF - (int)(best & 0xFFFFu))];
            a.matches[idx1] = bestIdx2;
            if (a.checkOri) {
                float rot = __fsub_rn(kp1.angle, a.keys2[bestIdx2].angle);
                if (rot < 0.0f) rot = __fadd_rn(rot, 360.0f);
                int bin = (int)roundf(__fmul_rn(rot, factor));
                if (bin == HISTO_LENGTH) bin = 0;
                a.binOf[idx1] = bin;
                atomicAdd(&a.result[1 + bin], 1);
            }
            atomicAdd(&a.result[0], 1);
        }
    }
}
__global__ __launch_bounds__(64) void k_sft_match(SftArgs a) { sft_match_body(a); }
// a batch of pairs (orbx_search_for_triangulation_frames): blockIdx.z = pair, its argument block from a device table
__global__ __launch_bounds__(64) void k_sft_match_batch(const SftArgs *tab)
{
    const SftArgs a = tab[blockIdx.z];
    if ((int)blockIdx.x >= a.nfv1 || a.nfv2 <= 0) return;
    sft_match_body(a);
}
__global__ __launch_bounds__(256) void k_sft_finish_batch(const SftArgs *tab, int nOut)
{
    const SftArgs &t = tab[blockIdx.x];
    SbowArgs f;
    f.checkOri = t.checkOri; f.nOut = nOut; f.matches = t.matches; f.binOf = t.binOf; f.result = t.result;
    sbow_finish_body(f);
}

hipError_t orbxLaunchSearchForTriangulation(hipStream_t st, const OrbxSftHost &h)
{
    SftArgs a;
    memset(&a, 0, sizeof a);
    a.node1 = h.node1; a.start1 = h.start1; a.feat1 = h.feat1; a.nfv1 = h.nfv1;
    a.keys1 = (const MtKp *)h.keys1; a.desc1 = h.desc1; a.hasMp1 = h.hasMp1; a.uRight1 = h.uRight1; a.nLeft1 = h.nLeft1;
    a.node2 = h.node2; a.start2 = h.start2; a.feat2 = h.feat2; a.nfv2 = h.nfv2;
    a.keys2 = (const MtKp *)h.keys2; a.desc2 = h.desc2; a.hasMp2 = h.hasMp2; a.uRight2 = h.uRight2; a.nLeft2 = h.nLeft2;
    a.scale2 = h.scale2; a.sigma1 = h.sigma1; a.sigma2 = h.sigma2;
    memcpy(a.ep, h.ep, sizeof a.ep); memcpy(a.F, h.F, sizeof a.F);
    a.model = h.model; a.rig = h.rig; a.onlyStereo = h.onlyStereo; a.coarse = h.coarse; a.checkOri = h.checkOri;
    memcpy(a.camL, h.camL, sizeof a.camL); memcpy(a.camR, h.camR, sizeof a.camR); a.precision = h.precision;
    memcpy(a.R, h.R, sizeof a.R); memcpy(a.t, h.t, sizeof a.t);
    a.matches = h.matches; a.binOf = h.binOf; a.result = h.result;
    hipError_t e = hipMemsetAsync(h.result, 0, 32 * sizeof(int), st);
    if (e == hipSuccess && h.n1 > 0) e = hipMemsetAsync(h.matches, 0xFF, (size_t)h.n1 * sizeof(int), st);
    if (e != hipSuccess) return e;
    if (h.nfv1 > 0 && h.nfv2 > 0) {
        hipLaunchKernelGGL(k_sft_match, dim3(h.nfv1, std::min(std::max(h.maxPerNode1, 1), 1024)), dim3(64), 0, st, a);
        SbowArgs f;
        memset(&f, 0, sizeof f);
        f.checkOri = h.checkOri; f.nOut = h.n1; f.matches = h.matches; f.binOf = h.binOf; f.result = h.result;
        hipLaunchKernelGGL(k_sbow_finish, dim3(1), dim3(256), 0, st, f);
    }
    return hipGetLastError();
}

static void fillSftArgs(SftArgs &a, const OrbxSftHost &h)
{
    memset(&a, 0, sizeof a);
    a.node1 = h.node1; a.start1 = h.start1; a.feat1 = h.feat1; a.nfv1 = h.nfv1;
    a.keys1 = (const MtKp *)h.keys1; a.desc1 = h.desc1; a.hasMp1 = h.hasMp1; a.uRight1 = h.uRight1; a.nLeft1 = h.nLeft1;
    a.node2 = h.node2; a.start2 = h.start2; a.feat2 = h.feat2; a.nfv2 = h.nfv2;
    a.keys2 = (const MtKp *)h.keys2; a.desc2 = h.desc2; a.hasMp2 = h.hasMp2; a.uRight2 = h.uRight2; a.nLeft2 = h.nLeft2;
    a.scale2 = h.scale2; a.sigma1 = h.sigma1; a.sigma2 = h.sigma2;
    memcpy(a.ep, h.ep, sizeof a.ep); memcpy(a.F, h.F, sizeof a.F);
    a.model = h.model; a.rig = h.rig; a.onlyStereo = h.onlyStereo; a.coarse = h.coarse; a.checkOri = h.checkOri;
    memcpy(a.camL, h.camL, sizeof a.camL); memcpy(a.camR, h.camR, sizeof a.camR); a.precision = h.precision;
    memcpy(a.R, h.R, sizeof a.R); memcpy(a.t, h.t, sizeof a.t);
    a.matches = h.matches; a.binOf = h.binOf; a.result = h.result;
}

size_t orbxSftArgsBytes() { return sizeof(SftArgs); }

// `n` pairs in two launches.  h[k].matches / result must be rows k of ONE [n][n1] / [n][32] array each (cleared here with one memset
// each); hostTable: n * orbxSftArgsBytes() bytes of host memory the caller keeps alive until the copy to dTable is done (it synchronises)
hipError_t orbxLaunchSearchForTriangulationBatch(hipStream_t st, const OrbxSftHost *h, int n, void *hostTable, void *dTable, int n1)
{
    if (n <= 0) return hipSuccess;
    SftArgs *tab = (SftArgs *)hostTable;
    int maxNfv1 = 0, maxPer = 1;
    for (int k = 0; k < n; ++k) {
        fillSftArgs(tab[k], h[k]);
        if (h[k].nfv2 > 0) maxNfv1 = std::max(maxNfv1, h[k].nfv1);
        maxPer = std::max(maxPer, h[k].maxPerNode1);
    }
    hipError_t e = hipMemcpyAsync(dTable, tab, (size_t)n * sizeof(SftArgs), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(h[0].result, 0, (size_t)n * 32 * sizeof(int), st);
    if (e == hipSuccess && n1 > 0) e = hipMemsetAsync(h[0].matches, 0xFF, (size_t)n * n1 * sizeof(int), st);
    if (e != hipSuccess) return e;
    if (maxNfv1 > 0) {
        hipLaunchKernelGGL(k_sft_match_batch, dim3(maxNfv1, std::min(maxPer, 1024), n), dim3(64), 0, st, (const SftArgs *)dTable);
        hipLaunchKernelGGL(k_sft_finish_batch, dim3(n), dim3(256), 0, st, (const SftArgs *)dTable, n1);
    }
    return hipGetLastError();
}


// ---- the window search of ORBmatcher::Fuse / SearchBySim3 ----------------------------------------------------------------------------
// reference src/ORBmatcher2.cc:522-579, 690-714 and src/ORBmatcher3.cc:120-145, 188-213 on KeyFrame::GetFeaturesInArea
// (src/KeyFrame.cc:704-748).  The projected map points do not interact (what Fuse does with a result is map bookkeeping of the caller),
// so one wave per point walks the window's cells in the reference's order (ix outer, iy inner, push order inside a cell) 64 keypoints at
// a time; "the first keypoint at the smallest distance" = the minimum of (distance << 32 | position in that order).
struct KfwPoint { float u, v, ur, radius; int level, search; };
struct KfwArgs {
    const MtKp *keys; const uint8_t *desc; const float *uRight; const float *invSigma2;
    const int *cellStart, *items; GridBounds b;
    const KfwPoint *pts; const uint8_t *ptDesc; int np, chi2;
    int *bestIdx, *bestDist;
    // SearchByProjection(KeyFrame, Sim3, ..): keypoints that hold a match on entry, and the earliest accepted point that wants a keypoint
    // in the current round of the fixed-point iteration (a point skips the keypoints an EARLIER point took) - NULL for Fuse / SearchBySim3
    const signed char *blocked; const int *owner;
    int levelHi;  // keypoint levels nPredictedLevel - 1 .. nPredictedLevel + levelHi (0; 1 for SearchByProjection(Frame, KeyFrame, ..))
    // a BATCH of keyframes (orbx_search_keyframe_windows_frames / orbx_fuse_frames): blockIdx.y = keyframe; record strides between
    // consecutive keyframes, the per-keyframe point counts on the device, and - Fuse - the acceptance bound and the per-keyframe counter
    long long keyStride, gridStride, itemStride, ptStride;   // keypoints / descriptors (x 32) / u_right; cellStart; items; points / descriptors (x 32) / outputs
    const int *npDev;
    int fuseTh;            // >= 0: bestIdx[i] = -1 unless bestDist <= fuseTh (TH_LOW), nFused[keyframe] counts the rest
    int *nFused;
};

__global__ __launch_bounds__(64) void k_kf_window_search(KfwArgs a)
{
    const int i = blockIdx.x, lane = threadIdx.x;
    if (a.npDev) {   // keyframe blockIdx.y of a batch
        const size_t f = blockIdx.y;
        a.np = min(a.np, a.npDev[f]);
        a.keys += f * a.keyStride; a.desc += f * a.keyStride * 32; if (a.uRight) a.uRight += f * a.keyStride;
        a.cellStart += f * a.gridStride; a.items += f * a.itemStride;
        a.pts += f * a.ptStride; a.ptDesc += f * a.ptStride * 32; a.bestIdx += f * a.ptStride; a.bestDist += f * a.ptStride;
        if (a.nFused) a.nFused += f;
    }
    if (i >= a.np) return;
    const KfwPoint P = a.pts[i];
    int outIdx = -1, outDist = INT_MAX;
    const float x = P.u, y = P.v, r = P.radius;
    const int cx0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
    const int cx1 = min(GRID_COLS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(x, a.b.minX), r), a.b.invW)));
    const int cy0 = max(0, (int)floorf(__fmul_rn(__fsub_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
    const int cy1 = min(GRID_ROWS - 1, (int)ceilf(__fmul_rn(__fadd_rn(__fsub_rn(y, a.b.minY), r), a.b.invH)));
    if (P.search && !(cx0 >= GRID_COLS || cx1 < 0 || cy0 >= GRID_ROWS || cy1 < 0)) {
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(a.ptDesc + (size_t)i * 32))[w];
        const int ny = cy1 - cy0 + 1, nCells = (cx1 - cx0 + 1) * ny;
        __shared__ int stage[64];
        unsigned long long best = ~0ull;
        unsigned seqBase = 0;
        for (int cb = 0; cb < nCells; cb += 64) {  // cells in (ix outer, iy inner) order, 64 at a time
            const int ci = cb + lane;
            int s = 0, e = 0;
            if (ci < nCells) {
                const int cell = (cx0 + ci / ny) * GRID_ROWS + cy0 + ci % ny;
                s = a.cellStart[cell]; e = a.cellStart[cell + 1];
            }
            int total;
            const int off = mt_excl_scan(e - s, total);
            // the keypoints of these cells in order, 64 per pass: lane -> the cell that holds position p (its lane range covers p)
            for (int pb = 0; pb < total; pb += 64) {
                __syncthreads();
                // every lane publishes the keypoints of its cell that fall into [pb, pb + 64)
                for (int j = max(s, s + (pb - off)); j < e && off + (j - s) < pb + 64; ++j) stage[off + (j - s) - pb] = a.items[j];
                __syncthreads();
                const int p = pb + lane;
                if (p < total) {
                    const int idx = stage[lane];
                    const MtKp kp = a.keys[idx];
                    bool keep = fabsf(__fsub_rn(kp.x, x)) < r && fabsf(__fsub_rn(kp.y, y)) < r;   // GetFeaturesInArea (:738-742)
                    const int kpLevel = kp.octave;
                    if (kpLevel < P.level - 1 || kpLevel > P.level + a.levelHi) keep = false;
                    if (keep && a.blocked && a.blocked[idx]) keep = false;
                    if (keep && a.owner) { const int o = a.owner[idx]; if (o >= 0 && o < i) keep = false; }
                    if (keep && a.chi2) {
                        const float ex = __fsub_rn(P.u, kp.x), ey = __fsub_rn(P.v, kp.y);
                        const float urc = a.uRight ? a.uRight[idx] : -1.f;
                        if (urc >= 0.f) {
                            const float er = __fsub_rn(P.ur, urc);
                            const float e2 = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(er, er));
                            if ((double)__fmul_rn(e2, a.invSigma2[kpLevel]) > 7.8) keep = false;
                        } else {
                            const float e2 = __fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey));
                            if ((double)__fmul_rn(e2, a.invSigma2[kpLevel]) > 5.99) keep = false;
                        }
                    }
                    if (keep) {
                        const uint32_t *t = (const uint32_t *)(a.desc + (size_t)idx * 32);
                        int dist = 0;
#pragma unroll
                        for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ t[w]);
                        const unsigned long long key = ((unsigned long long)(unsigned)dist << 32) | (unsigned long long)(seqBase + (unsigned)p);
                        best = min(best, key);
                    }
                }
            }
            seqBase += (unsigned)total;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best, o), hi = (unsigned)__shfl_xor((int)(unsigned)(best >> 32), o);
            best = min(best, ((unsigned long long)hi << 32) | lo);
        }
        if (best != ~0ull) {
            // position -> keypoint: walk the cells again on lane 0 (rare relative to the scan: once per point)
            outDist = (int)(best >> 32);
            unsigned pos = (unsigned)best;
            if (lane == 0) {
                for (int ci = 0; ci < nCells; ++ci) {
                    const int cell = (cx0 + ci / ny) * GRID_ROWS + cy0 + ci % ny;
                    const unsigned cnt = (unsigned)(a.cellStart[cell + 1] - a.cellStart[cell]);
                    if (pos < cnt) { outIdx = a.items[a.cellStart[cell] + (int)pos]; break; }
                    pos -= cnt;
                }
            }
        }
    }
    if (lane == 0) {
        if (a.fuseTh >= 0) {   // Fuse: bestDist <= TH_LOW (src/ORBmatcher2.cc:581, :716)
            if (outIdx >= 0 && outDist <= a.fuseTh) atomicAdd(a.nFused, 1);
            else outIdx = -1;
        }
        a.bestIdx[i] = outIdx; a.bestDist[i] = outDist;
    }
}

hipError_t orbxLaunchKeyframeWindowSearch(hipStream_t st, const void *keys, const uint8_t *desc, const float *uRight, const float *invSigma2,
                                          const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY, const void *pts,
                                          const uint8_t *ptDesc, int np, int chi2, int *bestIdx, int *bestDist, const signed char *blocked,
                                          const int *owner, int levelHi)
{
    if (np <= 0) return hipSuccess;
    KfwArgs a;
    a.blocked = blocked; a.owner = owner; a.levelHi = levelHi;
    a.keys = (const MtKp *)keys; a.desc = desc; a.uRight = uRight; a.invSigma2 = invSigma2; a.cellStart = cellStart; a.items = items;
    a.b.minX = minX; a.b.minY = minY; a.b.invW = (float)GRID_COLS / (maxX - minX); a.b.invH = (float)GRID_ROWS / (maxY - minY);
    a.pts = (const KfwPoint *)pts; a.ptDesc = ptDesc; a.np = np; a.chi2 = chi2; a.bestIdx = bestIdx; a.bestDist = bestDist;
    a.keyStride = a.gridStride = a.itemStride = a.ptStride = 0; a.npDev = nullptr; a.fuseTh = -1; a.nFused = nullptr;
    hipLaunchKernelGGL(k_kf_window_search, dim3(np), dim3(64), 0, st, a);
    return hipGetLastError();
}

// the same search for a batch of keyframes (blockIdx.y): see KfwArgs
hipError_t orbxLaunchKeyframeWindowSearchFrames(hipStream_t st, const OrbxKfwFrames *b)
{
    if (b->nframes <= 0 || b->maxNp <= 0) return hipSuccess;
    KfwArgs a;
    a.blocked = nullptr; a.owner = nullptr; a.levelHi = 0;
    a.keys = (const MtKp *)b->keys; a.desc = b->desc; a.uRight = b->uRight; a.invSigma2 = b->invSigma2; a.cellStart = b->cellStart; a.items = b->items;
    a.b.minX = b->minX; a.b.minY = b->minY; a.b.invW = (float)GRID_COLS / (b->maxX - b->minX); a.b.invH = (float)GRID_ROWS / (b->maxY - b->minY);
    a.pts = (const KfwPoint *)b->pts; a.ptDesc = b->ptDesc; a.np = b->maxNp; a.chi2 = b->chi2; a.bestIdx = b->bestIdx; a.bestDist = b->bestDist;
    a.keyStride = b->keyStride; a.gridStride = b->gridStride; a.itemStride = b->itemStride; a.ptStride = b->maxNp; a.npDev = b->npDev;
    a.fuseTh = b->fuseTh; a.nFused = b->nFused;
    if (b->nFused) { const hipError_t e = hipMemsetAsync(b->nFused, 0, (size_t)b->nframes * sizeof(int), st); if (e != hipSuccess) return e; }
    hipLaunchKernelGGL(k_kf_window_search, dim3(b->maxNp, b->nframes), dim3(64), 0, st, a);
    return hipGetLastError();
}

// ---- MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:329-403) -------------------------------------------------------
// One wave per map point, a lane per observed descriptor i (rows beyond 64 in further passes): the median of row i of the distance
// matrix is the order statistic k = (int)(0.5 (N - 1)) of its N distances (its own 0 included), found without storing or sorting the row:
// the smallest v in [0, 256] with #{j : dist(i, j) <= v} > k (a bisection over v, every step one sweep over the point's descriptors,
// which all lanes read at the same address).  BestIdx = the first row with the least median = min over (median << 16 | i).
__global__ __launch_bounds__(64) void k_distinctive(const uint8_t *desc, const int *start, int npoints, int *best)
{
    const int p = blockIdx.x, lane = threadIdx.x;
    if (p >= npoints) return;
    const int s = start[p], N = start[p + 1] - s;
    if (N <= 0) { if (lane == 0) best[p] = -1; return; }
    const uint8_t *D = desc + (size_t)s * 32;
    const int k = (int)(0.5 * (double)(N - 1));
    unsigned key = 0xFFFFFFFFu;
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        if (i >= N) continue;
        uint32_t q[8];
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = ((const uint32_t *)(D + (size_t)i * 32))[w];
        int lo = 0, hi = 256;  // invariant: #{dist <= hi} > k; answer in [lo, hi]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                const uint32_t *t = (const uint32_t *)(D + (size_t)j * 32);
                int dist = 0;
#pragma unroll
                for (int w = 0; w < 8; ++w) dist += __popc(q[w] ^ t[w]);
                cnt += dist <= mid ? 1 : 0;
            }
            if (cnt > k) hi = mid; else lo = mid + 1;
        }
        key = min(key, ((unsigned)lo << 16) | (unsigned)min(i, 0xFFFF));
    }
    key = mt_wave_min_u32(key);
    if (lane == 0) best[p] = (int)(key & 0xFFFFu);
}

hipError_t orbxLaunchDistinctive(hipStream_t st, const uint8_t *desc, const int *start, int npoints, int *best)
{
    if (npoints <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_distinctive, dim3(npoints), dim3(64), 0, st, desc, start, npoints, best);
    return hipGetLastError();
}
