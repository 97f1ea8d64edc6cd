// DBoW2 vocabulary-tree transform on the GPU: Frame::ComputeBoW (reference src/Frame.cc:768-775) ->
// TemplatedVocabulary<FORB::TDescriptor, FORB>::transform(features, BowVector, FeatureVector, levelsup)
// (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1196) with the per-feature descent of :1218-1261 and FORB::distance
// (FORB.cpp:79-98).  Two kernels:
//   k_bow_descend   16 lanes per feature walk the tree: every lane takes children sub, sub + 16, .. of the current node (their
//                   descriptors are stored contiguously in child-list order), the minimum of (distance, child position) over the
//                   16 lanes is the reference's "first child with the strictly smallest distance";
//   k_bow_assemble  one workgroup builds both std::map results: bitonic sort of (word, feature) and (node, feature) keys in LDS,
//                   run heads = map entries; a word's value is its weight added once per occurrence in sequence (addWeight), the
//                   norm is summed over the words in ascending order by one thread (floating-point sums are order-dependent and
//                   the map iterates in key order), then every value is divided by it.
// All arithmetic on the values is f64 in the reference's operation order (no contraction, IEEE division / sqrt): bit-identical.
#include "orbx_internal.h"

struct BowVocDev {
    const int *childStart;    // [nNodes + 1] CSR over the child lists (node ids ascending inside a list = loader push order)
    const int *childNode;     // [nNodes - 1] node id at each child-list position
    const uint4 *childDesc;   // [nNodes - 1][2] descriptors in child-list order
    const int *wordOfNode;    // [nNodes] word id of a leaf
    const double *weightOfNode;  // [nNodes]
    int L, scoring, weighting;
};

__device__ __forceinline__ int bow_popc256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) + __popc(a1.x ^ b1.x) +
           __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Batched form (nPerFrame != nullptr): blockIdx.y = frame, its features at feat + frame * cap * 2 and its outputs at frame * cap.
__global__ __launch_bounds__(256) void k_bow_descend(BowVocDev V, const uint4 *feat, int n, int levelsup, int *featWord, int *featNode,
                                                     double *featWeight, const int *nPerFrame, int cap)
{
    if (nPerFrame) {
        const size_t o = (size_t)blockIdx.y * cap;
        n = nPerFrame[blockIdx.y];
        feat += 2 * o; featWord += o; featNode += o; featWeight += o;
    }
    const int f = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    if (f >= n) return;  // whole 16-lane groups leave together: the shuffles below stay inside a group
    const uint4 q0 = feat[2 * f], q1 = feat[2 * f + 1];
    const int nidLevel = V.L - levelsup;
    int nid = nidLevel <= 0 ? 0 : -1;
    int node = 0, level = 0;
    int s = V.childStart[0], e = V.childStart[1];
    while (e > s && level < 64) {  // (parent < child for every node: the walk ends; 64 bounds it for a corrupted tree)
        unsigned best = 0xFFFFFFFFu;
        for (int c = s + sub; c < e; c += 16) {
            const uint4 d0 = V.childDesc[2 * (size_t)c], d1 = V.childDesc[2 * (size_t)c + 1];
            const unsigned key = ((unsigned)bow_popc256(q0, q1, d0, d1) << 20) | (unsigned)(c - s);
            best = min(best, key);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o, 16));
        node = V.childNode[s + (int)(best & 0xFFFFFu)];
        ++level;
        if (level == nidLevel) nid = node;
        s = V.childStart[node]; e = V.childStart[node + 1];
    }
    if (sub == 0) {
        const double w = V.weightOfNode[node];
        const bool on = w > 0;  // not stopped
        featWord[f] = on ? V.wordOfNode[node] : -1;
        featNode[f] = on ? (nid < 0 ? node : nid) : -1;
        featWeight[f] = w;
    }
}

#define BOW_NT 1024
typedef unsigned long long bow_u64;

__device__ __forceinline__ void bow_bitonic(bow_u64 *k, int np2)
{
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (np2 >> 1); t += BOW_NT) {
                const int lo = ((t & ~(stride - 1)) << 1) | (t & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const bow_u64 a = k[lo], b = k[hi];
                if ((a > b) == up) { k[lo] = b; k[hi] = a; }
            }
        }
    __syncthreads();
}

struct BowAsmArgs {
    const int *featWord, *featNode;
    const double *featWeight;
    int n, np2, scoring, weighting;
    int *bowWord; double *bowValue;          // [n]
    int *fvNode, *fvStart, *fvFeature;       // [n], [n + 1], [n]
    int *counts;                             // [2] words in the BowVector, nodes in the FeatureVector
    const int *nPerFrame; int cap;           // batched form: blockIdx.x = frame, n = nPerFrame[frame], arrays at frame * cap
};

__global__ __launch_bounds__(BOW_NT) void k_bow_assemble(BowAsmArgs a)
{
    if (a.nPerFrame) {
        const size_t o = (size_t)blockIdx.x * a.cap;
        a.n = a.nPerFrame[blockIdx.x];
        a.featWord += o; a.featNode += o; a.featWeight += o; a.bowWord += o; a.bowValue += o; a.fvNode += o; a.fvFeature += o;
        a.fvStart += o + blockIdx.x;  // (cap + 1 offsets per frame)
        a.counts += 2 * blockIdx.x;
    }
    extern __shared__ __attribute__((aligned(16))) bow_u64 keys[];  // [np2]
    __shared__ int headBase, nHeads, nValid;
    __shared__ double normS;
    const int tid = threadIdx.x;
    const bow_u64 PAD = ~0ull;
    for (int pass = 0; pass < 2; ++pass) {  // 0: BowVector (word keys), 1: FeatureVector (node keys)
        const int *src = pass == 0 ? a.featWord : a.featNode;
        for (int i = tid; i < a.np2; i += BOW_NT) {
            const int id = i < a.n ? src[i] : -1;
            keys[i] = id < 0 ? PAD : ((bow_u64)(unsigned)id << 32) | (unsigned)i;
        }
        bow_bitonic(keys, a.np2);
        if (tid == 0) { headBase = 0; nHeads = 0; nValid = 0; }
        __syncthreads();
        // run heads in order: chunks of BOW_NT sorted keys, ballot-free block scan through LDS atomics is not ordered, so each
        // chunk is ranked with a two-level count (per wave ballots + a serial pass over the 16 wave totals by thread 0)
        __shared__ int waveTot[BOW_NT / 64], waveValid[BOW_NT / 64];
        for (int c0 = 0; c0 < a.np2; c0 += BOW_NT) {
            const int i = c0 + tid;
            const bow_u64 k = i < a.np2 ? keys[i] : PAD;
            const bool valid = k != PAD;
            const bool head = valid && (i == 0 || (unsigned)(keys[i - 1] >> 32) != (unsigned)(k >> 32));
            const unsigned long long hb = __ballot(head), vb = __ballot(valid);
            const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
            if ((tid & 63) == 0) { waveTot[tid >> 6] = __popcll(hb); waveValid[tid >> 6] = __popcll(vb); }
            __syncthreads();
            int before = headBase;
            for (int w = 0; w < (tid >> 6); ++w) before += waveTot[w];
            const int slot = before + __popcll(hb & lt);
            if (head) {
                if (pass == 0) a.bowWord[slot] = (int)(k >> 32);
                else { a.fvNode[slot] = (int)(k >> 32); a.fvStart[slot] = i; }
            }
            if (pass == 0 && head) {
                // occurrences of this word = length of its run; value = w added once per occurrence, in sequence (addWeight);
                // IDF / BINARY keep the first w (addIfNotExist)
                int cnt = 1;
                while (i + cnt < a.np2 && (unsigned)(keys[i + cnt] >> 32) == (unsigned)(k >> 32)) ++cnt;
                const double w = a.featWeight[(unsigned)k];
                double acc = w;
                if (a.weighting == 0 || a.weighting == 1)
                    for (int r = 1; r < cnt; ++r) acc += w;
                a.bowValue[slot] = acc;
            }
            if (pass == 1 && valid) a.fvFeature[i] = (int)(unsigned)k;  // sorted by (node, feature): push_back order inside a node
            __syncthreads();
            if (tid == 0) {
                int h = 0, v = 0;
                for (int w = 0; w < BOW_NT / 64; ++w) { h += waveTot[w]; v += waveValid[w]; }
                headBase += h; nValid += v;
            }
            __syncthreads();
        }
        if (tid == 0) {
            nHeads = headBase;
            a.counts[pass] = headBase;
            if (pass == 1) a.fvStart[headBase] = nValid;
        }
        __syncthreads();
        if (pass == 0) {
            const int nw = nHeads;
            const bool must = a.scoring != 5;   // DotProductScoring: no normalisation (ScoringObject.h:74-89)
            const bool l2 = a.scoring == 1;     // L2Scoring; every other scoring normalises with L1
            __threadfence_block();
            if ((a.weighting == 0 || a.weighting == 1) && nw > 0 && !must) {
                const double nd = (double)nw;
                for (int i = tid; i < nw; i += BOW_NT) a.bowValue[i] = __ddiv_rn(a.bowValue[i], nd);
            }
            if (must) {
                // stage the values in LDS (the keys are done with), one thread sums them in ascending word order
                double *vals = (double *)keys;
                __syncthreads();
                for (int i = tid; i < nw; i += BOW_NT) vals[i] = a.bowValue[i];
                __syncthreads();
                if (tid == 0) {
                    double norm = 0.0;
                    if (!l2) { for (int i = 0; i < nw; ++i) norm += fabs(vals[i]); }
                    else { for (int i = 0; i < nw; ++i) norm += vals[i] * vals[i]; norm = __dsqrt_rn(norm); }
                    normS = norm;
                }
                __syncthreads();
                const double norm = normS;
                if (norm > 0.0)
                    for (int i = tid; i < nw; i += BOW_NT) a.bowValue[i] = __ddiv_rn(vals[i], norm);
            }
            __syncthreads();
        }
    }
}

hipError_t orbxLaunchBowTransform(hipStream_t st, const int *childStart, const int *childNode, const void *childDesc, const int *wordOfNode,
                                  const double *weightOfNode, int L, int scoring, int weighting, const void *feat, int n, int levelsup,
                                  int *featWord, int *featNode, double *featWeight, int *bowWord, double *bowValue, int *fvNode, int *fvStart,
                                  int *fvFeature, int *counts)
{
    BowVocDev V;
    V.childStart = childStart; V.childNode = childNode; V.childDesc = (const uint4 *)childDesc; V.wordOfNode = wordOfNode;
    V.weightOfNode = weightOfNode; V.L = L; V.scoring = scoring; V.weighting = weighting;
    if (n > 0) hipLaunchKernelGGL(k_bow_descend, dim3((n + 15) / 16), dim3(256), 0, st, V, (const uint4 *)feat, n, levelsup, featWord, featNode, featWeight,
                                  (const int *)nullptr, 0);
    BowAsmArgs a;
    a.featWord = featWord; a.featNode = featNode; a.featWeight = featWeight; a.n = n;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    a.np2 = np2; a.scoring = scoring; a.weighting = weighting;
    a.bowWord = bowWord; a.bowValue = bowValue; a.fvNode = fvNode; a.fvStart = fvStart; a.fvFeature = fvFeature; a.counts = counts;
    a.nPerFrame = nullptr; a.cap = 0;
    const size_t lds = (size_t)np2 * sizeof(bow_u64);
    static size_t attrBytes[64] = {};  // the attribute is per device: largest size set so far, by device ordinal
    int devId = 0;
    (void)hipGetDevice(&devId);
    size_t &ab = attrBytes[devId & 63];
    if (lds > 48 * 1024 && lds > ab) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_bow_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        ab = lds;
    }
    hipLaunchKernelGGL(k_bow_assemble, dim3(1), dim3(BOW_NT), lds, st, a);
    return hipGetLastError();
}

// Every frame of a batch in two launches: features of frame f at feat + f * cap * 32 bytes (the extractor's descriptor buffer),
// nPerFrame[f] of them; all per-feature and per-frame outputs at f * cap (fvStart at f * (cap + 1), counts at 2 f).
hipError_t orbxLaunchBowTransformBatch(hipStream_t st, const int *childStart, const int *childNode, const void *childDesc, const int *wordOfNode,
                                       const double *weightOfNode, int L, int scoring, int weighting, const void *feat, const int *nPerFrame,
                                       int nframes, int cap, int levelsup, int *featWord, int *featNode, double *featWeight, int *bowWord,
                                       double *bowValue, int *fvNode, int *fvStart, int *fvFeature, int *counts)
{
    if (nframes <= 0) return hipSuccess;
    BowVocDev V;
    V.childStart = childStart; V.childNode = childNode; V.childDesc = (const uint4 *)childDesc; V.wordOfNode = wordOfNode;
    V.weightOfNode = weightOfNode; V.L = L; V.scoring = scoring; V.weighting = weighting;
    hipLaunchKernelGGL(k_bow_descend, dim3((cap + 15) / 16, nframes), dim3(256), 0, st, V, (const uint4 *)feat, 0, levelsup, featWord, featNode,
                       featWeight, nPerFrame, cap);
    BowAsmArgs a;
    a.featWord = featWord; a.featNode = featNode; a.featWeight = featWeight; a.n = 0;
    int np2 = 2;
    while (np2 < cap) np2 <<= 1;
    a.np2 = np2; a.scoring = scoring; a.weighting = weighting;
    a.bowWord = bowWord; a.bowValue = bowValue; a.fvNode = fvNode; a.fvStart = fvStart; a.fvFeature = fvFeature; a.counts = counts;
    a.nPerFrame = nPerFrame; a.cap = cap;
    const size_t lds = (size_t)np2 * sizeof(bow_u64);
    if (lds > 48 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)k_bow_assemble, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_bow_assemble, dim3(nframes), dim3(BOW_NT), lds, st, a);
    return hipGetLastError();
}
