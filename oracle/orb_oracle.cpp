/*
 * orb_oracle.cpp — CPU restatement of the reference ORB front-end (parity oracle + timed CPU baseline).
 *
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the checker / baseline; the product never calls into it.
 *
 * PARITY STATUS: "parity unpinned" for image-dependent stages (the reference's tests pin no numbers on
 * this path and the reference needs OpenCV >= 4.4, absent here — see orb_oracle.h and DESIGN.md).
 *
 * Every function cites the reference file:line it follows (paths relative to the reference repo root).
 * OpenCV internals follow OpenCV 4.5.0 (the version recorded in the reference's ExecMean.txt:2), plain
 * C++ code paths (no IPP), as catalogued in SURVEY.md Appendix A.
 *
 * Build: g++ -O3 -std=c++17 -ffp-contract=off (no -march=native: the reference builds with plain -O3,
 * CMakeLists.txt:105-107, so x*b + y*a is an unfused multiply-add).
 */
#include "orb_oracle.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstring>
#include <list>
#include <thread>
#include <utility>
#include <map>
#include <vector>

#include "../include/orbx_synth.h"
#include "pattern_table.inc"

namespace {

// ---------------------------------------------------------------------------------------------------
// OpenCV scalar helpers: cvRound = round-half-to-even (SSE cvtss2si / lrint),
// cvFloor / cvCeil as in opencv2/core/fast_math.hpp.
// ---------------------------------------------------------------------------------------------------
inline int cvRoundF(float v) { return (int)lrintf(v); }
inline int cvRoundD(double v) { return (int)lrint(v); }
inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
inline int cvFloorD(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeilD(double v) { int i = (int)v; return i + (i < v); }
inline uint8_t sat8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
inline short satShort(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

const int PATCH_SIZE = 31;        // reference src/ORBextractor.cc:99
const int HALF_PATCH_SIZE = 15;   // :100
const int EDGE_THRESHOLD = 19;    // :101

struct Image {
    int w = 0, h = 0;
    std::vector<uint8_t> px;  // tight, row-major
    const uint8_t *row(int y) const { return px.data() + (size_t)y * w; }
    uint8_t *row(int y) { return px.data() + (size_t)y * w; }
};

// ---------------------------------------------------------------------------------------------------
// cv::resize(src, dst, sz, 0, 0, INTER_LINEAR) for CV_8UC1 — SURVEY App. A.1.
// Call site: reference src/ORBextractor.cc:1320.
// ---------------------------------------------------------------------------------------------------
void resizeLinear8u(const uint8_t *src, int sw, int sh, ptrdiff_t sstride, uint8_t *dst, int dw, int dh,
                    ptrdiff_t dstride)
{
    const int COEF_BITS = 11, COEF_SCALE = 1 << COEF_BITS;
    const double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    const double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;

    std::vector<int> xofs(dw), yofs(dh);
    std::vector<short> ialpha(dw * 2), ibeta(dh * 2);
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cvFloorF(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[dx * 2] = satShort(cvRoundF((1.f - fx) * COEF_SCALE));
        ialpha[dx * 2 + 1] = satShort(cvRoundF(fx * COEF_SCALE));
    }
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cvFloorF(fy);
        fy -= sy;
        yofs[dy] = sy;
        ibeta[dy * 2] = satShort(cvRoundF((1.f - fy) * COEF_SCALE));
        ibeta[dy * 2 + 1] = satShort(cvRoundF(fy * COEF_SCALE));
    }
    auto clipRow = [&](int y) { return y < 0 ? 0 : (y >= sh ? sh - 1 : y); };

    std::vector<int> h0(dw), h1(dw);
    auto hresize = [&](int sy, std::vector<int> &out) {
        const uint8_t *S = src + (ptrdiff_t)sy * sstride;
        for (int dx = 0; dx < dw; ++dx) {
            int sx = xofs[dx];
            int s1 = sx + 1 < sw ? sx + 1 : sw - 1;  // second tap replicated at the right edge (its weight is 0)
            out[dx] = S[sx] * ialpha[dx * 2] + S[s1] * ialpha[dx * 2 + 1];
        }
    };
    int cached0 = -1, cached1 = -1;
    for (int dy = 0; dy < dh; ++dy) {
        int sy0 = clipRow(yofs[dy]), sy1 = clipRow(yofs[dy] + 1);
        if (sy0 == cached1) { std::swap(h0, h1); std::swap(cached0, cached1); }
        if (sy0 != cached0) { hresize(sy0, h0); cached0 = sy0; }
        if (sy1 == cached0) { h1 = h0; cached1 = sy1; }
        else if (sy1 != cached1) { hresize(sy1, h1); cached1 = sy1; }
        const int b0 = ibeta[dy * 2], b1 = ibeta[dy * 2 + 1];
        uint8_t *D = dst + (ptrdiff_t)dy * dstride;
        for (int x = 0; x < dw; ++x)
            D[x] = (uint8_t)((((b0 * (h0[x] >> 4)) >> 16) + ((b1 * (h1[x] >> 4)) >> 16) + 2) >> 2);
    }
}

// ---------------------------------------------------------------------------------------------------
// cv::GaussianBlur(8U, Size(7,7), 2, 2, BORDER_REFLECT_101) — SURVEY App. A.2 (fixed-point 8.8 taps,
// exact 16-bit horizontal sums, (sum + 32768) >> 16 vertically, saturated).
// Call site: reference src/ORBextractor.cc:1270-1273 (on an isolated clone of the level).
// ---------------------------------------------------------------------------------------------------
const int kBlurTaps450[7] = {18, 34, 49, 55, 49, 34, 18};  // round(k*256), OpenCV 4.5.0 (sum 257)
const int kBlurTaps451[7] = {18, 34, 48, 56, 48, 34, 18};  // error-diffused, OpenCV >= 4.5.1 (sum 256)

inline int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) {
        if (p < 0) p = -p;
        else p = 2 * (len - 1) - p;
    }
    return p;
}

void gaussianBlur7(const uint8_t *src, int w, int h, ptrdiff_t sstride, uint8_t *dst, ptrdiff_t dstride,
                   const int taps[7])
{
    std::vector<uint16_t> tmp((size_t)w * h);
    std::vector<int> xi((size_t)w + 6);
    for (int x = -3; x < w + 3; ++x) xi[x + 3] = reflect101(x, w);
    for (int y = 0; y < h; ++y) {
        const uint8_t *S = src + (ptrdiff_t)y * sstride;
        uint16_t *T = tmp.data() + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            unsigned acc = 0;
            for (int k = 0; k < 7; ++k) acc += (unsigned)taps[k] * S[xi[x + k]];
            T[x] = (uint16_t)(acc > 65535u ? 65535u : acc);  // ufixedpoint16 saturating add (never hit: 257*255 = 65535)
        }
    }
    for (int y = 0; y < h; ++y) {
        const uint16_t *R[7];
        for (int k = 0; k < 7; ++k) R[k] = tmp.data() + (size_t)reflect101(y + k - 3, h) * w;
        uint8_t *D = dst + (ptrdiff_t)y * dstride;
        for (int x = 0; x < w; ++x) {
            uint32_t acc = 0;
            for (int k = 0; k < 7; ++k) acc += (uint32_t)taps[k] * R[k][x];
            D[x] = sat8((int)((acc + 32768u) >> 16));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// cv::FAST(roi, kps, threshold, nonmaxSuppression=true), TYPE_9_16 — SURVEY App. A.3.
// Call sites: reference src/ORBextractor.cc:908, 925.
// ---------------------------------------------------------------------------------------------------
const int kRing[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},   {3, 0},  {3, -1}, {2, -2}, {1, -3},
                          {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

// cornerScore<16>: largest threshold for which the pixel is still a FAST-9 corner.
int fastCornerScore(const uint8_t *p, const int pixel[25], int threshold)
{
    const int K = 8, N = K * 3 + 1;
    int v = p[0];
    int d[N];
    for (int k = 0; k < N; ++k) d[k] = v - p[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = std::min(d[k + 1], d[k + 2]);
        a = std::min(a, d[k + 3]);
        if (a <= a0) continue;
        a = std::min(a, d[k + 4]);
        a = std::min(a, d[k + 5]);
        a = std::min(a, d[k + 6]);
        a = std::min(a, d[k + 7]);
        a = std::min(a, d[k + 8]);
        a0 = std::max(a0, std::min(a, d[k]));
        a0 = std::max(a0, std::min(a, d[k + 9]));
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = std::max(d[k + 1], d[k + 2]);
        b = std::max(b, d[k + 3]);
        b = std::max(b, d[k + 4]);
        b = std::max(b, d[k + 5]);
        if (b >= b0) continue;
        b = std::max(b, d[k + 6]);
        b = std::max(b, d[k + 7]);
        b = std::max(b, d[k + 8]);
        b0 = std::min(b0, std::max(b, d[k]));
        b0 = std::min(b0, std::max(b, d[k + 9]));
    }
    return -b0 - 1;
}

struct FastPt { int x, y, score; };

void fast9_16(const uint8_t *img, int cols, int rows, ptrdiff_t step, int threshold, std::vector<FastPt> &out)
{
    out.clear();
    const int K = 8, N = 25;
    int pixel[25];
    for (int k = 0; k < 16; ++k) pixel[k] = kRing[k][0] + kRing[k][1] * (int)step;
    for (int k = 16; k < 25; ++k) pixel[k] = pixel[k - 16];
    threshold = std::min(std::max(threshold, 0), 255);
    if (cols < 7 || rows < 7) return;

    std::vector<uint8_t> bufmem((size_t)cols * 3, 0);
    std::vector<int> cpmem(((size_t)cols + 1) * 3, 0);
    uint8_t *buf[3] = {bufmem.data(), bufmem.data() + cols, bufmem.data() + 2 * cols};
    int *cpbuf[3] = {cpmem.data(), cpmem.data() + cols + 1, cpmem.data() + 2 * (cols + 1)};

    for (int i = 3; i < rows - 2; ++i) {
        const uint8_t *ptr = img + (ptrdiff_t)i * step + 3;
        uint8_t *curr = buf[(i - 3) % 3];
        int *cornerpos = cpbuf[(i - 3) % 3] + 1;
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; ++j, ++ptr) {
                int v = ptr[0];
                // contiguous-arc test, both polarities (a pixel cannot pass both: 9 + 9 > 16)
                bool corner = false;
                {
                    int vt = v - threshold, count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] < vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (!corner) {
                    int vt = v + threshold, count = 0;
                    for (int k = 0; k < N; ++k) {
                        if (ptr[pixel[k]] > vt) { if (++count > K) { corner = true; break; } }
                        else count = 0;
                    }
                }
                if (corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)fastCornerScore(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t *prev = buf[(i - 4 + 3) % 3];
        const uint8_t *pprev = buf[(i - 5 + 3) % 3];
        cornerpos = cpbuf[(i - 4 + 3) % 3] + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; ++k) {
            int j = cornerpos[k];
            int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] && score > pprev[j] &&
                score > pprev[j + 1] && score > curr[j - 1] && score > curr[j] && score > curr[j + 1])
                out.push_back({j, i - 1, score});
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// cv::fastAtan2 (degrees) — SURVEY App. A.4.  Imported by the reference at src/ORBextractor.cc:85.
// ---------------------------------------------------------------------------------------------------
float fastAtan2Deg(float y, float x)
{
    static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float eps = (float)2.2204460492503131e-16;  // (float)DBL_EPSILON
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------------------------------------------
// Quadtree (reference src/ORBextractor.cc:515-582 DivideNode / compareNodes, 584-774 DistributeOctTree;
// step specification in SURVEY App. D).  Keys are held as indices into the candidate arrays.
// ---------------------------------------------------------------------------------------------------
struct Cand { float x, y, response; };

struct QNode {
    std::vector<int> keys;
    int ulx = 0, uly = 0, urx = 0, bry = 0;  // UL.x, UL.y, UR.x, BR.y (BL/BR.x follow)
    std::list<QNode>::iterator self;
    bool noMore = false;
};

void divideNode(const QNode &n, const std::vector<Cand> &c, QNode out[4])
{
    const int halfX = (int)std::ceil(static_cast<float>(n.urx - n.ulx) / 2);
    const int halfY = (int)std::ceil(static_cast<float>(n.bry - n.uly) / 2);
    const int mx = n.ulx + halfX, my = n.uly + halfY;
    out[0].ulx = n.ulx; out[0].urx = mx;    out[0].uly = n.uly; out[0].bry = my;     // n1: upper-left
    out[1].ulx = mx;    out[1].urx = n.urx; out[1].uly = n.uly; out[1].bry = my;     // n2: upper-right
    out[2].ulx = n.ulx; out[2].urx = mx;    out[2].uly = my;    out[2].bry = n.bry;  // n3: bottom-left
    out[3].ulx = mx;    out[3].urx = n.urx; out[3].uly = my;    out[3].bry = n.bry;  // n4: bottom-right
    for (int q = 0; q < 4; ++q) out[q].keys.reserve(n.keys.size());
    for (int k : n.keys) {
        const Cand &kp = c[k];
        if (kp.x < (float)mx) {
            if (kp.y < (float)my) out[0].keys.push_back(k);
            else out[2].keys.push_back(k);
        } else if (kp.y < (float)my) out[1].keys.push_back(k);
        else out[3].keys.push_back(k);
    }
    for (int q = 0; q < 4; ++q)
        if (out[q].keys.size() == 1) out[q].noMore = true;
}

typedef std::pair<int, QNode *> SizedNode;
bool compareNodes(const SizedNode &e1, const SizedNode &e2)
{
    if (e1.first < e2.first) return true;
    if (e1.first > e2.first) return false;
    return e1.second->ulx < e2.second->ulx;
}

std::vector<int> distributeOctTree(const std::vector<Cand> &c, int minX, int maxX, int minY, int maxY, int N)
{
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::list<QNode> nodes;
    std::vector<QNode *> roots(nIni);
    for (int i = 0; i < nIni; ++i) {
        QNode ni;
        ni.ulx = (int)(hX * static_cast<float>(i));
        ni.urx = (int)(hX * static_cast<float>(i + 1));
        ni.uly = 0;
        ni.bry = maxY - minY;
        nodes.push_back(ni);
        roots[i] = &nodes.back();
    }
    for (size_t i = 0; i < c.size(); ++i) roots[(int)(c[i].x / hX)]->keys.push_back((int)i);

    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->keys.size() == 1) { it->noMore = true; ++it; }
        else if (it->keys.empty()) it = nodes.erase(it);
        else ++it;
    }

    bool finish = false;
    std::vector<SizedNode> sized;
    sized.reserve(nodes.size() * 4);
    auto pushChildren = [&](QNode ch[4], int *nToExpand) {
        for (int q = 0; q < 4; ++q) {
            if (ch[q].keys.empty()) continue;
            nodes.push_front(ch[q]);
            if (ch[q].keys.size() > 1) {
                if (nToExpand) ++*nToExpand;
                sized.push_back(std::make_pair((int)ch[q].keys.size(), &nodes.front()));
                nodes.front().self = nodes.begin();
            }
        }
    };
    while (!finish) {
        int prevSize = (int)nodes.size();
        int nToExpand = 0;
        sized.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->noMore) { ++it; continue; }
            QNode ch[4];
            divideNode(*it, c, ch);
            pushChildren(ch, &nToExpand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) {
            finish = true;
        } else if ((int)nodes.size() + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = (int)nodes.size();
                std::vector<SizedNode> prev = sized;
                sized.clear();
                std::sort(prev.begin(), prev.end(), compareNodes);  // libstdc++ introsort: tie order is observable
                for (int j = (int)prev.size() - 1; j >= 0; --j) {
                    QNode ch[4];
                    divideNode(*prev[j].second, c, ch);
                    pushChildren(ch, nullptr);
                    nodes.erase(prev[j].second->self);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || (int)nodes.size() == prevSize) finish = true;
            }
        }
    }
    std::vector<int> result;
    result.reserve(nodes.size());
    for (auto &n : nodes) {
        int best = n.keys[0];
        float maxResponse = c[best].response;
        for (size_t k = 1; k < n.keys.size(); ++k)
            if (c[n.keys[k]].response > maxResponse) { best = n.keys[k]; maxResponse = c[best].response; }
        result.push_back(best);
    }
    return result;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// The extractor object (reference include/ORBextractor.h:52-113, src/ORBextractor.cc:410-468).
// ---------------------------------------------------------------------------------------------------
struct oracle_handle {
    int nfeatures, nlevels, iniTh, minTh, blurVariant, trigMode;
    float scaleFactor;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> featuresPerLevel, umax;
    // state of the last extract
    std::vector<Image> pyr, blurred;
    std::vector<std::vector<FastPt>> cands;             // per level, relative to the 16-px border
    std::vector<std::vector<oracle_keypoint>> levelKps;  // per level, level coordinates, with angle

    oracle_handle(int nf, float sf, int nl, int ini, int mn, int bv, int am)
        : nfeatures(nf), nlevels(nl), iniTh(ini), minTh(mn), blurVariant(bv), trigMode(am), scaleFactor(sf)
    {
        scale.resize(nl); sigma2.resize(nl); invScale.resize(nl); invSigma2.resize(nl);
        scale[0] = 1.0f; sigma2[0] = 1.0f;
        for (int i = 1; i < nl; ++i) {
            scale[i] = scale[i - 1] * scaleFactor;
            sigma2[i] = scale[i] * scale[i];
        }
        for (int i = 0; i < nl; ++i) {
            invScale[i] = 1.0f / scale[i];
            invSigma2[i] = 1.0f / sigma2[i];
        }
        featuresPerLevel.resize(nl);
        float factor = 1.0f / scaleFactor;
        float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
        int sum = 0;
        for (int level = 0; level < nl - 1; ++level) {
            featuresPerLevel[level] = cvRoundF(nDesired);
            sum += featuresPerLevel[level];
            nDesired *= factor;
        }
        featuresPerLevel[nl - 1] = std::max(nfeatures - sum, 0);

        umax.resize(HALF_PATCH_SIZE + 1);
        int v, v0, vmax = cvFloorF(HALF_PATCH_SIZE * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeilD(HALF_PATCH_SIZE * std::sqrt(2.f) / 2);
        const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
        for (v = 0; v <= vmax; ++v) umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
        for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }

    const int *taps() const { return blurVariant ? kBlurTaps451 : kBlurTaps450; }

    // ORBextractor::ComputePyramid — reference src/ORBextractor.cc:1309-1329.  The 19-px REFLECT_101
    // padding of mvImagePyramid is never read by any extractor stage and is not built.
    void computePyramid(const uint8_t *img, int w, int h, ptrdiff_t stride)
    {
        pyr.assign(nlevels, Image());
        for (int level = 0; level < nlevels; ++level) {
            float s = invScale[level];
            int sw = cvRoundF((float)w * s), sh = cvRoundF((float)h * s);
            Image &L = pyr[level];
            L.w = sw; L.h = sh; L.px.resize((size_t)sw * sh);
            if (level == 0) {
                for (int y = 0; y < h; ++y) memcpy(L.row(y), img + (ptrdiff_t)y * stride, w);
            } else {
                const Image &P = pyr[level - 1];
                resizeLinear8u(P.px.data(), P.w, P.h, P.w, L.px.data(), sw, sh, sw);
            }
        }
    }

    // tileCalcKeypoints — reference src/ORBextractor.cc:867-950 (the CPU algorithm of record for FAST).
    void tileCalcKeypoints(const Image &img, std::vector<FastPt> &out) const
    {
        out.clear();
        const int border = EDGE_THRESHOLD - 3;
        const int minBorderX = border, minBorderY = border;
        const float W = 35;
        const int maxBorderX = img.w - EDGE_THRESHOLD + 3;
        const int maxBorderY = img.h - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX);
        const float height = (float)(maxBorderY - minBorderY);
        const int nCols = (int)(width / W);
        const int nRows = (int)(height / W);
        if (nCols < 1 || nRows < 1) return;  // image too small for the reference (division by zero there)
        const int wCell = (int)std::ceil(width / nCols);
        const int hCell = (int)std::ceil(height / nRows);
        std::vector<FastPt> cell;
        for (int i = 0; i < nRows; ++i) {
            const float iniY = (float)(minBorderY + i * hCell);
            float maxY = iniY + hCell + 6;
            if (iniY >= maxBorderY - 3) continue;
            if (maxY > maxBorderY) maxY = (float)maxBorderY;
            for (int j = 0; j < nCols; ++j) {
                const float iniX = (float)(minBorderX + j * wCell);
                float maxX = iniX + wCell + 6;
                if (iniX >= maxBorderX - 6) continue;
                if (maxX > maxBorderX) maxX = (float)maxBorderX;
                const int y0 = (int)iniY, y1 = (int)maxY, x0 = (int)iniX, x1 = (int)maxX;
                const uint8_t *roi = img.row(y0) + x0;
                fast9_16(roi, x1 - x0, y1 - y0, img.w, iniTh, cell);
                if (cell.empty()) fast9_16(roi, x1 - x0, y1 - y0, img.w, minTh, cell);
                for (const FastPt &p : cell) out.push_back({p.x + j * wCell, p.y + i * hCell, p.score});
            }
        }
    }

    // IC_Angle: upstream ORB-SLAM3 CPU function (removed from this fork; moment definition preserved in
    // reference src/OpenCL/Kernel/Angle.cl:24-60, umax at :3) + cv::fastAtan2.
    float icAngle(const Image &img, int cx, int cy) const
    {
        int m_01 = 0, m_10 = 0;
        const uint8_t *center = img.row(cy) + cx;
        const int step = img.w;
        for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
        for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
            int v_sum = 0;
            int d = umax[v];
            for (int u = -d; u <= d; ++u) {
                int val_plus = center[u + v * step], val_minus = center[u - v * step];
                v_sum += (val_plus - val_minus);
                m_10 += u * (val_plus + val_minus);
            }
            m_01 += v * v_sum;
        }
        return fastAtan2Deg((float)m_01, (float)m_10);
    }

    // computeOrbDescriptor — reference src/ORBextractor.cc:105-149.
    void orbDescriptor(const oracle_keypoint &kpt, const Image &img, uint8_t *desc) const
    {
        const float factorPI = (float)(M_PI / 180.f);
        float angle = (float)kpt.angle * factorPI;
        // `cos(angle)` is unqualified in the reference (:108, `using namespace std` commented out at :76-77): ::cos(double) with only
        // <cmath> in scope (trig_mode 0); the float overload = the platform libm's cosf / sinf if libstdc++'s <math.h> wrapper is
        // visible, or with upstream ORB-SLAM3's `using namespace std` (trig_mode 1: the REAL libm here, not a restatement)
        float a, b;
        if (trigMode == 1) { a = cosf(angle); b = sinf(angle); }
        else { a = (float)cos((double)angle); b = (float)sin((double)angle); }
        const uint8_t *center = img.row(cvRoundF(kpt.y)) + cvRoundF(kpt.x);
        const int step = img.w;
        const signed char *pat = kOraclePattern31;
        for (int i = 0; i < 32; ++i, pat += 32) {
            int val = 0;
            for (int k = 0; k < 8; ++k) {
                const signed char *p = pat + 4 * k;
                float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
                int t0 = center[cvRoundF(x0 * b + y0 * a) * step + cvRoundF(x0 * a - y0 * b)];
                int t1 = center[cvRoundF(x1 * b + y1 * a) * step + cvRoundF(x1 * a - y1 * b)];
                val |= (t0 < t1) << k;
            }
            desc[i] = (uint8_t)val;
        }
    }

    // ComputeKeyPointsOctTree — reference src/ORBextractor.cc:958-1027, with the CPU candidate path
    // (tileCalcKeypoints, :990-994) and the CPU border/octave/size loop preserved as a comment at :1004-1012.
    void computeKeyPointsOctTree()
    {
        const int border = EDGE_THRESHOLD - 3;
        cands.assign(nlevels, {});
        levelKps.assign(nlevels, {});
        for (int level = 0; level < nlevels; ++level) {
            const Image &img = pyr[level];
            const int maxBorderX = img.w - EDGE_THRESHOLD + 3;
            const int maxBorderY = img.h - EDGE_THRESHOLD + 3;
            tileCalcKeypoints(img, cands[level]);
            std::vector<Cand> c(cands[level].size());
            for (size_t i = 0; i < c.size(); ++i)
                c[i] = {(float)cands[level][i].x, (float)cands[level][i].y, (float)cands[level][i].score};
            std::vector<int> sel;
            if (!c.empty()) sel = distributeOctTree(c, border, maxBorderX, border, maxBorderY, featuresPerLevel[level]);
            const int scaledPatchSize = (int)(PATCH_SIZE * scale[level]);
            std::vector<oracle_keypoint> &out = levelKps[level];
            out.resize(sel.size());
            for (size_t i = 0; i < sel.size(); ++i) {
                oracle_keypoint &k = out[i];
                k.x = c[sel[i]].x + (float)border;
                k.y = c[sel[i]].y + (float)border;
                k.size = (float)scaledPatchSize;
                k.response = c[sel[i]].response;
                k.octave = level;
                k.class_id = -1;
                k.angle = icAngle(img, cvRoundF(k.x), cvRoundF(k.y));
            }
        }
    }

    // ORBextractor::operator() — reference src/ORBextractor.cc:1227-1307.
    int extract(const uint8_t *img, int w, int h, ptrdiff_t stride, int lap0, int lap1, oracle_keypoint *kps,
                uint8_t *desc, int *n_out)
    {
        if (!img || w <= 0 || h <= 0) { if (n_out) *n_out = 0; return -1; }
        computePyramid(img, w, h, stride);
        computeKeyPointsOctTree();
        int nkeypoints = 0;
        for (int level = 0; level < nlevels; ++level) nkeypoints += (int)levelKps[level].size();
        if (n_out) *n_out = nkeypoints;
        blurred.assign(nlevels, Image());
        int monoIndex = 0, stereoIndex = nkeypoints - 1;
        uint8_t d[32];
        for (int level = 0; level < nlevels; ++level) {
            std::vector<oracle_keypoint> &keypoints = levelKps[level];
            // the reference skips the blur for empty levels (:1266-1267); the accessor then returns zeros
            Image &B = blurred[level];
            B.w = pyr[level].w; B.h = pyr[level].h;
            if (keypoints.empty()) continue;
            B.px.resize(pyr[level].px.size());
            gaussianBlur7(pyr[level].px.data(), B.w, B.h, B.w, B.px.data(), B.w, taps());
            float s = scale[level];
            for (const oracle_keypoint &kp0 : keypoints) {
                orbDescriptor(kp0, B, d);
                oracle_keypoint kp = kp0;
                if (level != 0) { kp.x *= s; kp.y *= s; }
                int pos;
                if (kp.x >= (float)lap0 && kp.x <= (float)lap1) pos = stereoIndex--;
                else pos = monoIndex++;
                if (kps) kps[pos] = kp;
                if (desc) memcpy(desc + (size_t)pos * 32, d, 32);
            }
        }
        return monoIndex;
    }
};

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

oracle_handle *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th, int blur_variant,
                             int trig_mode)
{
    if (nfeatures <= 0 || nlevels <= 0 || nlevels > 32 || !(scale_factor > 1.0f)) return nullptr;
    return new oracle_handle(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant, trig_mode);
}
void oracle_destroy(oracle_handle *h) { delete h; }

int oracle_get_tables(const oracle_handle *h, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                      int32_t *fpl, int32_t *umax16)
{
    for (int i = 0; i < h->nlevels; ++i) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->invScale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->invSigma2[i];
        if (fpl) fpl[i] = h->featuresPerLevel[i];
    }
    if (umax16) for (int i = 0; i < 16; ++i) umax16[i] = h->umax[i];
    return h->nlevels;
}
int oracle_max_keypoints(const oracle_handle *h)
{
    int s = 0;
    for (int v : h->featuresPerLevel) s += v + 3;
    return s;
}
// DistributeOctTree returns up to max(N + 3, 4 * nIni) keypoints of a level: the first pass splits every one of the
// nIni = round(W / H) root nodes (src/ORBextractor.cc:589) into up to 4 before the quota is looked at (:697), so very wide images
// with small quotas exceed N + 3.
int oracle_max_keypoints_for(const oracle_handle *h, int w, int hh)
{
    int s = 0;
    for (size_t l = 0; l < h->featuresPerLevel.size(); ++l) {
        const int lw = l == 0 ? w : cvRoundF((float)w * h->invScale[l]), lh = l == 0 ? hh : cvRoundF((float)hh * h->invScale[l]);
        const int W = std::max(lw - 32, 1), H = std::max(lh - 32, 1);
        const int nIni = (int)std::round((float)W / (float)H);
        s += std::max(h->featuresPerLevel[l] + 3, 4 * nIni);
    }
    return s;
}
int oracle_extract(oracle_handle *h, const uint8_t *img, int w, int h_, ptrdiff_t stride, int lap0, int lap1,
                   oracle_keypoint *kps, uint8_t *desc, int *n_out)
{
    return h->extract(img, w, h_, stride, lap0, lap1, kps, desc, n_out);
}
int oracle_level_size(const oracle_handle *h, int level, int *w, int *hh)
{
    if (level < 0 || level >= (int)h->pyr.size()) return -1;
    *w = h->pyr[level].w; *hh = h->pyr[level].h;
    return 0;
}
int oracle_get_level(const oracle_handle *h, int level, int blurred, uint8_t *out)
{
    const std::vector<Image> &v = blurred ? h->blurred : h->pyr;
    if (level < 0 || level >= (int)v.size()) return -1;
    if (v[level].px.empty()) { memset(out, 0, (size_t)v[level].w * v[level].h); return 1; }
    memcpy(out, v[level].px.data(), v[level].px.size());
    return 0;
}
int oracle_get_candidates(const oracle_handle *h, int level, uint32_t *out, int cap)
{
    if (level < 0 || level >= (int)h->cands.size()) return -1;
    const auto &c = h->cands[level];
    for (int i = 0; i < (int)c.size() && i < cap; ++i)
        out[i] = (uint32_t)c[i].x | ((uint32_t)c[i].y << 12) | ((uint32_t)c[i].score << 24);
    return (int)c.size();
}
int oracle_get_level_keypoints(const oracle_handle *h, int level, oracle_keypoint *out, int cap)
{
    if (level < 0 || level >= (int)h->levelKps.size()) return -1;
    const auto &k = h->levelKps[level];
    for (int i = 0; i < (int)k.size() && i < cap; ++i) out[i] = k[i];
    return (int)k.size();
}

int oracle_extract_batch(oracle_handle *h, const uint8_t *imgs, int n, int w, int hh, ptrdiff_t stride, int lap0,
                         int lap1, int threads, int32_t *counts, int32_t *mono, oracle_keypoint *kps, uint8_t *desc,
                         int cap)
{
    if (threads < 1) threads = 1;
    std::atomic<int> next(0);
    auto worker = [&]() {
        oracle_handle local(h->nfeatures, h->scaleFactor, h->nlevels, h->iniTh, h->minTh, h->blurVariant, h->trigMode);
        std::vector<oracle_keypoint> k(oracle_max_keypoints(&local));
        std::vector<uint8_t> d((size_t)k.size() * 32);
        for (;;) {
            int f = next.fetch_add(1);
            if (f >= n) break;
            int cnt = 0;
            int m = local.extract(imgs + (size_t)f * hh * stride, w, hh, stride, lap0, lap1, k.data(), d.data(), &cnt);
            if (counts) counts[f] = cnt;
            if (mono) mono[f] = m;
            int c = std::min(cnt, cap);
            if (kps) memcpy(kps + (size_t)f * cap, k.data(), (size_t)c * sizeof(oracle_keypoint));
            if (desc) memcpy(desc + (size_t)f * cap * 32, d.data(), (size_t)c * 32);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return 0;
}

int oracle_resize_linear(const uint8_t *src, int sw, int sh, ptrdiff_t sstride, uint8_t *dst, int dw, int dh,
                         ptrdiff_t dstride)
{
    resizeLinear8u(src, sw, sh, sstride, dst, dw, dh, dstride);
    return 0;
}
int oracle_gaussian_blur7(const uint8_t *src, int w, int h, ptrdiff_t sstride, uint8_t *dst, ptrdiff_t dstride,
                          int blur_variant)
{
    gaussianBlur7(src, w, h, sstride, dst, dstride, blur_variant ? kBlurTaps451 : kBlurTaps450);
    return 0;
}
int oracle_fast_roi(const uint8_t *roi, int cols, int rows, ptrdiff_t stride, int threshold, uint32_t *out, int cap)
{
    std::vector<FastPt> v;
    fast9_16(roi, cols, rows, stride, threshold, v);
    for (int i = 0; i < (int)v.size() && i < cap; ++i)
        out[i] = (uint32_t)v[i].x | ((uint32_t)v[i].y << 12) | ((uint32_t)v[i].score << 24);
    return (int)v.size();
}
float oracle_fast_atan2(float y, float x) { return fastAtan2Deg(y, x); }

int oracle_distribute(const uint32_t *cand, int ncand, int W, int H, int N, int32_t *out_idx, int cap)
{
    std::vector<Cand> c(ncand);
    for (int i = 0; i < ncand; ++i)
        c[i] = {(float)(cand[i] & 0xFFF), (float)((cand[i] >> 12) & 0xFFF), (float)(cand[i] >> 24)};
    if (c.empty()) return 0;
    std::vector<int> r = distributeOctTree(c, 0, W, 0, H, N);
    for (int i = 0; i < (int)r.size() && i < cap; ++i) out_idx[i] = r[i];
    return (int)r.size();
}

int oracle_sort_pairs(const int32_t *count, const int32_t *ulx, int n, int32_t *order)
{
    std::vector<QNode> nodes(n);
    std::vector<SizedNode> v(n);
    for (int i = 0; i < n; ++i) { nodes[i].ulx = ulx[i]; v[i] = std::make_pair((int)count[i], &nodes[i]); }
    std::sort(v.begin(), v.end(), compareNodes);
    for (int i = 0; i < n; ++i) order[i] = (int32_t)(v[i].second - nodes.data());
    return 0;
}

// ORBmatcher::DescriptorDistance — reference src/ORBmatcher3.cc:637-653 (SWAR popcount of 8 xor-ed words).
int oracle_hamming256(const void *a, const void *b)
{
    uint32_t pa[8], pb[8];
    memcpy(pa, a, 32); memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; ++i) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) — SURVEY App. A.7; call site reference src/Frame.cc:1174.
int oracle_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    const int K = 2;
    for (int i = 0; i < nq; ++i) {
        int32_t *di = dist + (size_t)i * K, *ni = idx + (size_t)i * K;
        for (int k = 0; k < K; ++k) { di[k] = INT_MAX; ni[k] = -1; }
        for (int j = 0; j < nt; ++j) {
            int d = oracle_hamming256(q + (size_t)i * 32, t + (size_t)j * 32);
            if (d < di[K - 1]) {
                int k;
                for (k = K - 2; k >= 0 && di[k] > d; --k) { ni[k + 1] = ni[k]; di[k + 1] = di[k]; }
                ni[k + 1] = j;
                di[k + 1] = d;
            }
        }
    }
    return 0;
}

// Frame::ComputeStereoMatches, Hamming stage — reference src/Frame.cc:841-925 (TH_HIGH = 100,
// src/ORBmatcher1.cc:37).
int oracle_stereo_rows(const oracle_keypoint *kl, const uint8_t *dl, int nl, const oracle_keypoint *kr,
                       const uint8_t *dr, int nr, int n_rows, const float *sf, float min_d, float max_d,
                       int32_t *best_idx, int32_t *best_dist)
{
    const int TH_HIGH = 100;
    std::vector<std::vector<int>> rowIdx(n_rows);
    for (int iR = 0; iR < nr; ++iR) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * sf[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < n_rows) rowIdx[yi].push_back(iR);  // the reference would index out of range here
    }
    for (int iL = 0; iL < nl; ++iL) {
        best_idx[iL] = -1;
        best_dist[iL] = TH_HIGH;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        int row = (int)vL;
        if (row < 0 || row >= n_rows) continue;
        const std::vector<int> &cands = rowIdx[row];
        if (cands.empty()) continue;
        const float minU = uL - max_d, maxU = uL - min_d;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdx = -1;
        for (int iR : cands) {
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                int d = oracle_hamming256(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (d < bestDist) { bestDist = d; bestIdx = iR; }
            }
        }
        best_idx[iL] = bestIdx;
        best_dist[iL] = bestDist;
    }
    return 0;
}

// Frame::ComputeStereoMatches — reference src/Frame.cc:841-1011 (thOrbDist = (TH_HIGH + TH_LOW) / 2, :846).
int oracle_stereo_matches(const oracle_handle *L, const oracle_handle *R, const oracle_keypoint *kl, const uint8_t *dl, int nl,
                          const oracle_keypoint *kr, const uint8_t *dr, int nr, float mb, float mbf, float *uRightOut, float *depthOut)
{
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = L->pyr[0].h;
    for (int i = 0; i < nl; ++i) { uRightOut[i] = -1.0f; depthOut[i] = -1.0f; }
    std::vector<std::vector<int>> rowIdx(nRows);
    for (int iR = 0; iR < nr; ++iR) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * L->scale[kr[iR].octave];
        const int maxr = (int)std::ceil(kpY + r);
        const int minr = (int)std::floor(kpY - r);
        for (int yi = minr; yi <= maxr; ++yi)
            if (yi >= 0 && yi < nRows) rowIdx[yi].push_back(iR);
    }
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    std::vector<std::pair<int, int>> vDistIdx;
    for (int iL = 0; iL < nl; ++iL) {
        const oracle_keypoint &kpL = kl[iL];
        const int levelL = kpL.octave;
        const float vL = kpL.y, uL = kpL.x;
        const int row = (int)vL;
        if (row < 0 || row >= nRows) continue;
        const std::vector<int> &cands = rowIdx[row];
        if (cands.empty()) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH;
        size_t bestIdxR = 0;
        for (int iR : cands) {
            const oracle_keypoint &kpR = kr[iR];
            if (kpR.octave < levelL - 1 || kpR.octave > levelL + 1) continue;
            const float uR = kpR.x;
            if (uR >= minU && uR <= maxU) {
                const int dist = oracle_hamming256(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = L->invScale[kpL.octave];
            const float scaleduL = std::round(kpL.x * scaleFactor);
            const float scaledvL = std::round(kpL.y * scaleFactor);
            const float scaleduR0 = std::round(uR0 * scaleFactor);
            const int w = 5;
            const Image &imL = L->pyr[kpL.octave], &imR = R->pyr[kpL.octave];
            int bestDistS = INT_MAX, bestincR = 0;
            const int Ls = 5;
            float vDists[11];
            const float iniu = scaleduR0 + Ls - w;
            const float endu = scaleduR0 + Ls + w + 1;
            if (iniu < 0 || endu >= imR.w) continue;
            const int y0 = (int)(scaledvL - w), xL0 = (int)(scaleduL - w);
            for (int incR = -Ls; incR <= +Ls; ++incR) {
                const int xR0 = (int)(scaleduR0 + incR - w);
                int sad = 0;
                for (int yy = 0; yy < 2 * w + 1; ++yy)
                    for (int xx = 0; xx < 2 * w + 1; ++xx)
                        sad += std::abs((int)imL.row(y0 + yy)[xL0 + xx] - (int)imR.row(y0 + yy)[xR0 + xx]);
                const float dist = (float)sad;  // cv::norm(IL, IR, NORM_L1)
                if (dist < bestDistS) { bestDistS = (int)dist; bestincR = incR; }
                vDists[Ls + incR] = dist;
            }
            if (bestincR == -Ls || bestincR == Ls) continue;
            const float dist1 = vDists[Ls + bestincR - 1], dist2 = vDists[Ls + bestincR], dist3 = vDists[Ls + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = L->scale[kpL.octave] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = (uL - bestuR);
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01; bestuR = uL - 0.01; }  // double literals, as in the reference (:985-986)
                depthOut[iL] = mbf / disparity;
                uRightOut[iL] = bestuR;
                vDistIdx.push_back(std::make_pair(bestDistS, iL));
            }
        }
    }
    if (vDistIdx.empty()) return 0;  // the reference would index an empty vector here
    std::sort(vDistIdx.begin(), vDistIdx.end());
    const float median = (float)vDistIdx[vDistIdx.size() / 2].first;
    const float thDist = 1.5f * 1.4f * median;
    for (int i = (int)vDistIdx.size() - 1; i >= 0; --i) {
        if (vDistIdx[i].first < thDist) break;
        uRightOut[vDistIdx[i].second] = -1;
        depthOut[vDistIdx[i].second] = -1;
    }
    return (int)vDistIdx.size();
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------
// the frame grid and ORBmatcher::SearchForInitialization
// ---------------------------------------------------------------------------------------------------------
#define ORACLE_GRID_COLS 64  // FRAME_GRID_COLS, reference include/Frame.h:45
#define ORACLE_GRID_ROWS 48  // FRAME_GRID_ROWS, :44

struct FrameGrid {  // Frame::mGrid + the static image bounds (src/Frame.cc:41, 161-162)
    std::vector<std::vector<size_t>> cell;  // [ix * ROWS + iy], indices in push_back order
    float minX, minY, invW, invH;
};

// Frame::AssignFeaturesToGrid + Frame::PosInGrid (src/Frame.cc:387-418, 755-766), monocular branch (Nleft == -1)
static FrameGrid assignFeaturesToGrid(const oracle_keypoint *k, int n, float minX, float maxX, float minY, float maxY)
{
    FrameGrid G;
    G.cell.resize(ORACLE_GRID_COLS * ORACLE_GRID_ROWS);
    G.minX = minX; G.minY = minY;
    G.invW = static_cast<float>(ORACLE_GRID_COLS) / (maxX - minX);
    G.invH = static_cast<float>(ORACLE_GRID_ROWS) / (maxY - minY);
    for (int i = 0; i < n; ++i) {
        const int posX = (int)std::round((k[i].x - minX) * G.invW);  // std::round(float): half away from zero
        const int posY = (int)std::round((k[i].y - minY) * G.invH);
        if (posX < 0 || posX >= ORACLE_GRID_COLS || posY < 0 || posY >= ORACLE_GRID_ROWS) continue;
        G.cell[posX * ORACLE_GRID_ROWS + posY].push_back((size_t)i);
    }
    return G;
}

// Frame::GetFeaturesInArea (src/Frame.cc:687-753), bRight = false
static std::vector<size_t> getFeaturesInArea(const FrameGrid &G, const oracle_keypoint *k, float x, float y, float r, int minLevel,
                                             int maxLevel)
{
    std::vector<size_t> vIndices;
    const float factorX = r, factorY = r;
    const int nMinCellX = std::max(0, (int)std::floor((x - G.minX - factorX) * G.invW));
    if (nMinCellX >= ORACLE_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min((int)ORACLE_GRID_COLS - 1, (int)std::ceil((x - G.minX + factorX) * G.invW));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, (int)std::floor((y - G.minY - factorY) * G.invH));
    if (nMinCellY >= ORACLE_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min((int)ORACLE_GRID_ROWS - 1, (int)std::ceil((y - G.minY + factorY) * G.invH));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const std::vector<size_t> &vCell = G.cell[ix * ORACLE_GRID_ROWS + iy];
            for (size_t j = 0; j < vCell.size(); j++) {
                const oracle_keypoint &kpUn = k[vCell[j]];
                if (bCheckLevels) {
                    if (kpUn.octave < minLevel) continue;
                    if (maxLevel >= 0 && kpUn.octave > maxLevel) continue;
                }
                const float distx = kpUn.x - x, disty = kpUn.y - y;
                if (std::fabs(distx) < factorX && std::fabs(disty) < factorY) vIndices.push_back(vCell[j]);
            }
        }
    return vIndices;
}

// ORBmatcher::ComputeThreeMaxima (src/ORBmatcher3.cc:592-633) on the histogram sizes
static void computeThreeMaxima(const int *histoSize, int L, int &ind1, int &ind2, int &ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histoSize[i];
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

extern "C" {

int oracle_assign_grid(const oracle_keypoint *k, int n, float minX, float maxX, float minY, float maxY, int32_t *cell_start,
                       int32_t *items)
{
    const FrameGrid G = assignFeaturesToGrid(k, n, minX, maxX, minY, maxY);
    int o = 0;
    for (int c = 0; c < ORACLE_GRID_COLS * ORACLE_GRID_ROWS; ++c) {
        cell_start[c] = o;
        for (size_t v : G.cell[c]) items[o++] = (int32_t)v;
    }
    cell_start[ORACLE_GRID_COLS * ORACLE_GRID_ROWS] = o;
    return o;
}

int oracle_features_in_area(const oracle_keypoint *k, int n, float minX, float maxX, float minY, float maxY, float x, float y, float r,
                            int min_level, int max_level, int32_t *out, int cap)
{
    const FrameGrid G = assignFeaturesToGrid(k, n, minX, maxX, minY, maxY);
    const std::vector<size_t> v = getFeaturesInArea(G, k, x, y, r, min_level, max_level);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = (int32_t)v[i];
    return (int)v.size();
}

// ORBmatcher::SearchForInitialization (src/ORBmatcher1.cc:650-765).  F1 / F2 = undistorted keypoints + descriptors of the two
// frames; (minX .. maxY) = Frame::mnMinX .. mnMaxY; prev_matched = vbPrevMatched [n1][2], updated in place; matches12 [n1].
int oracle_search_for_initialization(const oracle_keypoint *k1, const uint8_t *d1, int n1, const oracle_keypoint *k2, const uint8_t *d2,
                                     int n2, float minX, float maxX, float minY, float maxY, float *prev_matched, int32_t *matches12,
                                     int window_size, float nnratio, int check_orientation)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;  // src/ORBmatcher1.cc:37-39
    const FrameGrid G2 = assignFeaturesToGrid(k2, n2, minX, maxX, minY, maxY);
    int nmatches = 0;
    std::vector<int> vnMatches12(n1, -1);
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    std::vector<int> vMatchedDistance(n2, INT_MAX);
    std::vector<int> vnMatches21(n2, -1);
    for (int i1 = 0; i1 < n1; i1++) {
        const oracle_keypoint kp1 = k1[i1];
        const int level1 = kp1.octave;
        if (level1 > 0) continue;
        const std::vector<size_t> vIndices2 =
            getFeaturesInArea(G2, k2, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, level1, level1);
        if (vIndices2.empty()) continue;
        const uint8_t *dd1 = d1 + (size_t)i1 * 32;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (size_t v = 0; v < vIndices2.size(); ++v) {
            const size_t i2 = vIndices2[v];
            const int dist = oracle_hamming256(dd1, d2 + i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = (int)i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW) {
            if (bestDist < (float)bestDist2 * nnratio) {
                if (vnMatches21[bestIdx2] >= 0) { vnMatches12[vnMatches21[bestIdx2]] = -1; nmatches--; }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = k1[i1].angle - k2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(i1);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[30];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0; j < rotHist[i].size(); j++) {
                const int idx1 = rotHist[i][j];
                if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
            }
        }
    }
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = vnMatches12[i1];
        if (vnMatches12[i1] >= 0) { prev_matched[2 * i1] = k2[vnMatches12[i1]].x; prev_matched[2 * i1 + 1] = k2[vnMatches12[i1]].y; }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) — reference
// src/ORBmatcher3.cc:256-467, branch CurrentFrame.Nleft == -1 (monocular / rectified stereo).  The pose and camera arithmetic
// that precedes the search in the reference loop (x3Dc = Tcw * x3Dw with Sophus, mpCamera->project) stays with the caller: a
// projection carries (u, v, invzc), the octave and angle of the last frame's keypoint, MapPoint::GetDescriptor() and two flags
// (bit 0: the keypoint has a map point and is not an outlier; bit 1: that map point has Observations() > 0).
// cur_state[nc] = CurrentFrame.mvpMapPoints before the call: -1 none, 0 a map point with no observations, 1 with observations.
// assign[nc] receives what the call did to CurrentFrame.mvpMapPoints[i2]: -1 untouched, j >= 0 = the map point of last-frame
// keypoint j, -2 set to NULL by the rotation check.  Returns nmatches.
int oracle_search_by_projection(const oracle_keypoint *kc, const uint8_t *dc, const float *u_right, const int8_t *cur_state, int nc,
                                float minX, float maxX, float minY, float maxY, const float *scale_factors, int nlevels, float mbf,
                                const oracle_projection *proj, const uint8_t *dproj, int np, float th, int forward, int backward,
                                int check_orientation, int32_t *assign)
{
    (void)nlevels;
    const int TH_HIGH = 100, HISTO_LENGTH = 30;  // src/ORBmatcher1.cc:37-39
    const FrameGrid G = assignFeaturesToGrid(kc, nc, minX, maxX, minY, maxY);
    std::vector<int> mp(nc), obs(nc);  // mvpMapPoints: -1 none, -2 NULLed, -3 the pre-existing one, j the last-frame index; its Observations() > 0
    for (int i = 0; i < nc; ++i) { mp[i] = cur_state[i] < 0 ? -1 : -3; obs[i] = cur_state[i] > 0; assign[i] = -1; }
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < np; i++) {
        const oracle_projection &P = proj[i];
        if (!(P.flags & 1)) continue;  // pMP && !mvbOutlier[i]
        const float invzc = P.invz;
        if (invzc < 0) continue;
        if (P.u < minX || P.u > maxX) continue;
        if (P.v < minY || P.v > maxY) continue;
        const int nLastOctave = P.octave;
        const float radius = th * scale_factors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (forward) vIndices2 = getFeaturesInArea(G, kc, P.u, P.v, radius, nLastOctave, -1);
        else if (backward) vIndices2 = getFeaturesInArea(G, kc, P.u, P.v, radius, 0, nLastOctave);
        else vIndices2 = getFeaturesInArea(G, kc, P.u, P.v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = dproj + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (size_t v = 0; v < vIndices2.size(); ++v) {
            const size_t i2 = vIndices2[v];
            if (mp[i2] != -1 && mp[i2] != -2)
                if (obs[i2]) continue;
            if (u_right && u_right[i2] > 0) {
                const float ur = P.u - mbf * invzc;
                const float er = std::fabs(ur - u_right[i2]);
                if (er > radius) continue;
            }
            const int dist = oracle_hamming256(dMP, dc + i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= TH_HIGH) {
            mp[bestIdx2] = i; obs[bestIdx2] = (P.flags >> 1) & 1;
            nmatches++;
            if (check_orientation) {
                float rot = P.angle - kc[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)std::round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[30];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { mp[rotHist[i][j]] = -2; nmatches--; }
    }
    for (int i = 0; i < nc; ++i) assign[i] = mp[i] == -3 ? -1 : mp[i];
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th, bFarPoints, thFarPoints) —
// reference src/ORBmatcher1.cc:45-215, branch F.Nleft == -1, map points with mbTrackInView (set by Frame::isInFrustum together
// with mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos; the caller also applies the isBad() and far-point tests and clears
// flag bit 0 for those).  A record = {u: mTrackProjX, v: mTrackProjY, invz: mTrackProjXR, angle: mTrackViewCos,
// octave: mnTrackScaleLevel, flags}.  cur_state / assign as in oracle_search_by_projection.  Returns nmatches.
int oracle_search_by_projection_points(const oracle_keypoint *kc, const uint8_t *dc, const float *u_right, const int8_t *cur_state, int nc,
                                       float minX, float maxX, float minY, float maxY, const float *scale_factors, int nlevels,
                                       const oracle_projection *mps, const uint8_t *dmp, int np, float th, float nnratio, int32_t *assign)
{
    (void)nlevels;
    const int TH_HIGH = 100;
    const FrameGrid G = assignFeaturesToGrid(kc, nc, minX, maxX, minY, maxY);
    std::vector<int> mp(nc), obs(nc);
    for (int i = 0; i < nc; ++i) { mp[i] = cur_state[i] < 0 ? -1 : -3; obs[i] = cur_state[i] > 0; }
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < np; iMP++) {
        const oracle_projection &P = mps[iMP];
        if (!(P.flags & 1)) continue;  // !mbTrackInView, far point or isBad()
        const int nPredictedLevel = P.octave;
        float r = P.angle > 0.998 ? 2.5 : 4.0;  // RadiusByViewingCos(mTrackViewCos), :217-223
        if (bFactor) r *= th;
        const std::vector<size_t> vIndices =
            getFeaturesInArea(G, kc, P.u, P.v, r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
        if (vIndices.empty()) continue;
        const uint8_t *MPdescriptor = dmp + (size_t)iMP * 32;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (size_t v = 0; v < vIndices.size(); ++v) {
            const size_t idx = vIndices[v];
            if (mp[idx] != -1)
                if (obs[idx]) continue;
            if (u_right && u_right[idx] > 0) {
                const float er = std::fabs(P.invz - u_right[idx]);
                if (er > r * scale_factors[nPredictedLevel]) continue;
            }
            const int dist = oracle_hamming256(MPdescriptor, dc + idx * 32);
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kc[idx].octave; bestIdx = (int)idx; }
            else if (dist < bestDist2) { bestLevel2 = kc[idx].octave; bestDist2 = dist; }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
            if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                mp[bestIdx] = iMP; obs[bestIdx] = (P.flags >> 1) & 1;
                nmatches++;
            }
        }
    }
    for (int i = 0; i < nc; ++i) assign[i] = mp[i] == -3 ? -1 : mp[i];
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) — reference src/ORBmatcher3.cc:256-467,
// branch CurrentFrame.Nleft != -1 (KB8 fisheye stereo rig): per last-frame map point the left window search (:300-372, on mGrid /
// mvKeys, no mvuRight test) and then the right one (:373-436, on mGridRight / mvKeysRight, keypoint index + Nleft).  kc / dc =
// mvKeys followed by mvKeysRight / mDescriptors (vconcat of both, Frame.cc:1138-1142), nc = n_left + n_right; cur_state / assign index
// that combined range.  proj_l[i] = the projection with mpCamera (u, v, invzc, angle, octave, flags), proj_r[i].u / .v =
// mpCamera->project(GetRelativePoseTrl() * x3Dc) (the reference projects the right side with mpCamera too, :375).  The `continue`s of
// the left part (invzc < 0, uv outside the bounds, an empty left window :315) skip the right part as the reference does.
int oracle_search_by_projection_fisheye(const oracle_keypoint *kc, const uint8_t *dc, int n_left, int n_right, const int8_t *cur_state,
                                        float minX, float maxX, float minY, float maxY, const float *scale_factors, int nlevels,
                                        const oracle_projection *proj_l, const oracle_projection *proj_r, const uint8_t *dproj, int np,
                                        float th, int forward, int backward, int check_orientation, int32_t *assign)
{
    (void)nlevels;
    const int TH_HIGH = 100, HISTO_LENGTH = 30, nc = n_left + n_right;
    const oracle_keypoint *kr = kc + n_left;
    const FrameGrid G = assignFeaturesToGrid(kc, n_left, minX, maxX, minY, maxY);
    const FrameGrid GR = assignFeaturesToGrid(kr, n_right, minX, maxX, minY, maxY);
    std::vector<int> mp(nc), obs(nc);
    for (int i = 0; i < nc; ++i) { mp[i] = cur_state[i] < 0 ? -1 : -3; obs[i] = cur_state[i] > 0; }
    int nmatches = 0;
    std::vector<int> rotHist[30];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < np; i++) {
        const oracle_projection &P = proj_l[i];
        if (!(P.flags & 1)) continue;
        if (P.invz < 0) continue;
        if (P.u < minX || P.u > maxX) continue;
        if (P.v < minY || P.v > maxY) continue;
        const int nLastOctave = P.octave;
        const float radius = th * scale_factors[nLastOctave];
        const uint8_t *dMP = dproj + (size_t)i * 32;
        for (int side = 0; side < 2; ++side) {
            const FrameGrid &g = side ? GR : G;
            const oracle_keypoint *k = side ? kr : kc;
            const int base = side ? n_left : 0;
            const float u = side ? proj_r[i].u : P.u, v = side ? proj_r[i].v : P.v;
            std::vector<size_t> vIndices2;
            if (forward) vIndices2 = getFeaturesInArea(g, k, u, v, radius, nLastOctave, -1);
            else if (backward) vIndices2 = getFeaturesInArea(g, k, u, v, radius, 0, nLastOctave);
            else vIndices2 = getFeaturesInArea(g, k, u, v, radius, nLastOctave - 1, nLastOctave + 1);
            if (side == 0 && vIndices2.empty()) break;  // :315 `continue` leaves the right part out as well
            int bestDist = 256, bestIdx2 = -1;
            for (size_t c = 0; c < vIndices2.size(); ++c) {
                const size_t i2 = vIndices2[c] + base;
                if (mp[i2] != -1 && mp[i2] != -2)
                    if (obs[i2]) continue;
                const int dist = oracle_hamming256(dMP, dc + i2 * 32);
                if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
            }
            if (bestDist <= TH_HIGH) {
                mp[bestIdx2] = i; obs[bestIdx2] = (P.flags >> 1) & 1;
                nmatches++;
                if (check_orientation) {
                    float rot = P.angle - kc[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)std::round(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rotHist[bin].push_back(bestIdx2);
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1, sizes[30];
        for (int i = 0; i < HISTO_LENGTH; ++i) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { mp[rotHist[i][j]] = -2; nmatches--; }
    }
    for (int i = 0; i < nc; ++i) assign[i] = mp[i] == -3 ? -1 : mp[i];
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, th, ..) — reference src/ORBmatcher1.cc:45-215, branch
// F.Nleft != -1.  mp_l[i] = {mTrackProjX, mTrackProjY, -, mTrackViewCos, mnTrackScaleLevel, flags} (flag bit 0 = mbTrackInView), mp_r[i] =
// {mTrackProjXR, mTrackProjYR, -, mTrackViewCosR, mnTrackScaleLevelR, flags} (bit 0 = mbTrackInViewR); the caller clears bit 0 on both
// for isBad() / far points.  left_to_right / right_to_left = F.mvLeftToRightMatch / mvRightToLeftMatch: a match on one side is copied to
// its stereo partner (:133-137, :198-202) without looking at what the partner slot held.  A left ratio-test failure `continue`s past the
// right search (:126-127); the right radius takes no th factor (:147).
int oracle_search_by_projection_points_fisheye(const oracle_keypoint *kc, const uint8_t *dc, int n_left, int n_right, const int8_t *cur_state,
                                               float minX, float maxX, float minY, float maxY, const float *scale_factors, int nlevels,
                                               const oracle_projection *mp_l, const oracle_projection *mp_r, const uint8_t *dmp, int np,
                                               const int32_t *left_to_right, const int32_t *right_to_left, float th, float nnratio,
                                               int32_t *assign)
{
    (void)nlevels;
    const int TH_HIGH = 100, nc = n_left + n_right;
    const oracle_keypoint *kr = kc + n_left;
    const FrameGrid G = assignFeaturesToGrid(kc, n_left, minX, maxX, minY, maxY);
    const FrameGrid GR = assignFeaturesToGrid(kr, n_right, minX, maxX, minY, maxY);
    std::vector<int> mp(nc), obs(nc);
    for (int i = 0; i < nc; ++i) { mp[i] = cur_state[i] < 0 ? -1 : -3; obs[i] = cur_state[i] > 0; }
    int nmatches = 0;
    const bool bFactor = th != 1.0;
    for (int iMP = 0; iMP < np; iMP++) {
        const oracle_projection &L = mp_l[iMP], &R = mp_r[iMP];
        if (!(L.flags & 1) && !(R.flags & 1)) continue;
        const uint8_t *MPdescriptor = dmp + (size_t)iMP * 32;
        const int o = (L.flags >> 1) & 1;
        if (L.flags & 1) {
            const int nPredictedLevel = L.octave;
            float r = L.angle > 0.998 ? 2.5 : 4.0;
            if (bFactor) r *= th;
            const std::vector<size_t> vIndices =
                getFeaturesInArea(G, kc, L.u, L.v, r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
            if (!vIndices.empty()) {
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (size_t c = 0; c < vIndices.size(); ++c) {
                    const size_t idx = vIndices[c];
                    if (mp[idx] != -1)
                        if (obs[idx]) continue;
                    const int dist = oracle_hamming256(MPdescriptor, dc + idx * 32);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kc[idx].octave; bestIdx = (int)idx; }
                    else if (dist < bestDist2) { bestLevel2 = kc[idx].octave; bestDist2 = dist; }
                }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (bestLevel != bestLevel2 || bestDist <= nnratio * bestDist2) {
                        mp[bestIdx] = iMP; obs[bestIdx] = o;
                        if (left_to_right[bestIdx] != -1) {
                            mp[left_to_right[bestIdx] + n_left] = iMP; obs[left_to_right[bestIdx] + n_left] = o;
                            nmatches++;
                        }
                        nmatches++;
                    }
                }
            }
        }
        if (R.flags & 1) {
            const int nPredictedLevel = R.octave;
            if (nPredictedLevel != -1) {
                const float r = R.angle > 0.998 ? 2.5 : 4.0;
                const std::vector<size_t> vIndices =
                    getFeaturesInArea(GR, kr, R.u, R.v, r * scale_factors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
                if (vIndices.empty()) continue;
                int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
                for (size_t c = 0; c < vIndices.size(); ++c) {
                    const size_t idx = vIndices[c];
                    if (mp[idx + n_left] != -1)
                        if (obs[idx + n_left]) continue;
                    const int dist = oracle_hamming256(MPdescriptor, dc + (idx + n_left) * 32);
                    if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestLevel2 = bestLevel; bestLevel = kr[idx].octave; bestIdx = (int)idx; }
                    else if (dist < bestDist2) { bestLevel2 = kr[idx].octave; bestDist2 = dist; }
                }
                if (bestDist <= TH_HIGH) {
                    if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
                    if (right_to_left[bestIdx] != -1) {
                        mp[right_to_left[bestIdx]] = iMP; obs[right_to_left[bestIdx]] = o;
                        nmatches++;
                    }
                    mp[bestIdx + n_left] = iMP; obs[bestIdx + n_left] = o;
                    nmatches++;
                }
            }
        }
    }
    for (int i = 0; i < nc; ++i) assign[i] = mp[i] == -3 ? -1 : mp[i];
    return nmatches;
}

// cv::undistortPoints(src, dst, cameraMatrix, distCoeffs, R = noArray(), P) as Frame::UndistortKeyPoints / ComputeImageBounds call it
// (reference src/Frame.cc:777-811, :813-845; OpenCV is an external dependency of the reference, not vendored: 4.x, calib3d
// cvUndistortPointsInternal restated here from its published source — parity unpinned): everything in double, five fixed-point
// iterations of the inverse Brown-Conrady model (TermCriteria(MAX_ITER, 5, 0.01)), coefficients k1 k2 p1 p2 [k3 [k4 k5 k6 [s1..s4]]],
// then the new camera matrix P (fx fy cx cy; zero skew).  k / kNew: fx, fy, cx, cy.
static void undistortPoint(double u, double v, const double K[4], const double k[12], const double KN[4], float &ox, float &oy)
{
    const double fx = K[0], fy = K[1], ifx = 1. / fx, ify = 1. / fy, cx = K[2], cy = K[3];
    double x = (u - cx) * ifx, y = (v - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
        if (icdist < 0) { x = (u - cx) * ifx; y = (v - cy) * ify; break; }
        const double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
        const double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    // RR = P * I: rows (fx' 0 cx'), (0 fy' cy'), (0 0 1)
    const double xx = KN[0] * x + 0. * y + KN[2], yy = 0. * x + KN[1] * y + KN[3], ww = 1. / (0. * x + 0. * y + 1.);
    ox = (float)(xx * ww);
    oy = (float)(yy * ww);
}

static void loadCamera(const float *k4, const float *dist, int ndist, const float *knew4, double K[4], double k[12], double KN[4])
{
    for (int i = 0; i < 4; ++i) { K[i] = k4[i]; KN[i] = knew4[i]; }
    for (int i = 0; i < 12; ++i) k[i] = i < ndist ? (double)dist[i] : 0.;
}

// Frame::UndistortKeyPoints — reference src/Frame.cc:777-811: copy when distCoef[0] == 0, else undistort the positions and keep
// the other fields.  Returns 0.
int oracle_undistort_keypoints(const oracle_keypoint *kps, int n, const float *k4, const float *dist, int ndist, const float *knew4,
                               oracle_keypoint *out)
{
    for (int i = 0; i < n; ++i) out[i] = kps[i];
    if (ndist <= 0 || dist[0] == 0.0) return 0;
    double K[4], k[12], KN[4];
    loadCamera(k4, dist, ndist, knew4, K, k, KN);
    for (int i = 0; i < n; ++i) undistortPoint(kps[i].x, kps[i].y, K, k, KN, out[i].x, out[i].y);
    return 0;
}

// Frame::ComputeImageBounds — reference src/Frame.cc:813-845: bounds[4] = mnMinX, mnMaxX, mnMinY, mnMaxY.
int oracle_compute_image_bounds(int cols, int rows, const float *k4, const float *dist, int ndist, const float *knew4, float *bounds)
{
    if (ndist <= 0 || dist[0] == 0.0) { bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows; return 0; }
    double K[4], k[12], KN[4];
    loadCamera(k4, dist, ndist, knew4, K, k, KN);
    const float cx[4] = {0.0f, (float)cols, 0.0f, (float)cols}, cy[4] = {0.0f, 0.0f, (float)rows, (float)rows};
    float ux[4], uy[4];
    for (int i = 0; i < 4; ++i) undistortPoint(cx[i], cy[i], K, k, KN, ux[i], uy[i]);
    bounds[0] = std::min(ux[0], ux[2]); bounds[1] = std::max(ux[1], ux[3]);
    bounds[2] = std::min(uy[0], uy[1]); bounds[3] = std::max(uy[2], uy[3]);
    return 0;
}

// ---- DBoW2 (vendored in the reference: Thirdparty/DBoW2/DBoW2) — SURVEY 8f rank 4 ---------------------------------------------------
// TemplatedVocabulary<FORB::TDescriptor, FORB> as Frame::ComputeBoW uses it (reference src/Frame.cc:768-775 ->
// TemplatedVocabulary.h:1127-1196 transform(features, BowVector, FeatureVector, levelsup) -> :1218-1261 transform(feature, ..) with
// FORB::distance, FORB.cpp:79-98).  The tree is given as the loader builds it (loadFromTextFile :1338-1423): node 0 is the root,
// node i > 0 has parent[i] < i, children in node-id order, word ids in order of the leaves.
struct OracleVocabulary {
    int k, L, scoring, weighting;
    std::vector<std::vector<int> > children;
    std::vector<int> parent, wordId;
    std::vector<uint8_t> desc;
    std::vector<double> weight;
    int nWords;
};

void *oracle_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                               const uint8_t *desc, const double *weight)
{
    OracleVocabulary *v = new OracleVocabulary;
    v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting;
    v->children.resize(n_nodes); v->parent.assign(parent, parent + n_nodes); v->wordId.assign(n_nodes, 0);
    v->desc.assign(desc, desc + (size_t)n_nodes * 32); v->weight.assign(weight, weight + n_nodes);
    v->nWords = 0;
    for (int nid = 1; nid < n_nodes; ++nid) {
        v->children[parent[nid]].push_back(nid);
        if (is_leaf[nid] > 0) v->wordId[nid] = v->nWords++;
    }
    return v;
}
void oracle_vocabulary_destroy(void *h) { delete (OracleVocabulary *)h; }
int oracle_vocabulary_words(void *h) { return ((OracleVocabulary *)h)->nWords; }

// transform(feature, word_id, weight, nid, levelsup) — TemplatedVocabulary.h:1218-1261.  (A leaf above the level asked for
// leaves *nid untouched in the reference, whose caller passes an uninitialised variable; here it then holds the leaf's id.)
static void vocTransformOne(const OracleVocabulary &V, const uint8_t *feature, int &word_id, double &weight, int &nid, int levelsup)
{
    const int nid_level = V.L - levelsup;
    bool nidSet = false;
    if (nid_level <= 0) { nid = 0; nidSet = true; }
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const std::vector<int> &nodes = V.children[final_id];
        final_id = nodes[0];
        double best_d = oracle_hamming256(feature, &V.desc[(size_t)final_id * 32]);
        for (size_t c = 1; c < nodes.size(); ++c) {
            const int id = nodes[c];
            const double d = oracle_hamming256(feature, &V.desc[(size_t)id * 32]);
            if (d < best_d) { best_d = d; final_id = id; }
        }
        if (current_level == nid_level) { nid = final_id; nidSet = true; }
    } while (!V.children[final_id].empty());
    if (!nidSet) nid = final_id;
    word_id = V.wordId[final_id];
    weight = V.weight[final_id];
}

// transform(features, v, fv, levelsup) — TemplatedVocabulary.h:1127-1196 with BowVector::addWeight / addIfNotExist / normalize
// (BowVector.cpp:34-84) and FeatureVector::addFeature (FeatureVector.cpp:31-45) on std::map, exactly as there.  Outputs: the
// BowVector as (word, value) in ascending word order, the FeatureVector as CSR over ascending node ids, plus the per-feature
// word / node (−1 where the word is stopped, weight <= 0).  Returns the number of words in the BowVector.
int oracle_vocabulary_transform(void *h, const uint8_t *desc, int n, int levelsup, int32_t *feat_word, int32_t *feat_node,
                                int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start,
                                int32_t *fv_feature, int32_t *n_fv)
{
    const OracleVocabulary &V = *(OracleVocabulary *)h;
    std::map<unsigned, double> v;
    std::map<unsigned, std::vector<unsigned> > fv;
    *n_bow = 0; *n_fv = 0; fv_start[0] = 0;
    if (V.children.empty() || V.children[0].empty()) return 0;  // empty()
    const bool must = V.scoring != 5;                 // DotProductScoring does not normalise (ScoringObject.h:74-89)
    const bool l2 = V.scoring == 1;                   // L2Scoring: L2, all others L1
    for (int i = 0; i < n; ++i) {
        int id, nid; double w;
        vocTransformOne(V, desc + (size_t)i * 32, id, w, nid, levelsup);
        feat_word[i] = w > 0 ? id : -1; feat_node[i] = w > 0 ? nid : -1;
        if (!(w > 0)) continue;  // stopped
        if (V.weighting == 0 || V.weighting == 1) {   // TF_IDF, TF: addWeight
            std::map<unsigned, double>::iterator vit = v.lower_bound((unsigned)id);
            if (vit != v.end() && !(v.key_comp()((unsigned)id, vit->first))) vit->second += w;
            else v.insert(vit, std::make_pair((unsigned)id, w));
        } else {                                      // IDF, BINARY: addIfNotExist
            std::map<unsigned, double>::iterator vit = v.lower_bound((unsigned)id);
            if (vit == v.end() || v.key_comp()((unsigned)id, vit->first)) v.insert(vit, std::make_pair((unsigned)id, w));
        }
        std::map<unsigned, std::vector<unsigned> >::iterator fit = fv.lower_bound((unsigned)nid);
        if (fit != fv.end() && fit->first == (unsigned)nid) fit->second.push_back((unsigned)i);
        else { fit = fv.insert(fit, std::make_pair((unsigned)nid, std::vector<unsigned>())); fit->second.push_back((unsigned)i); }
    }
    if ((V.weighting == 0 || V.weighting == 1) && !v.empty() && !must) {
        const double nd = v.size();
        for (std::map<unsigned, double>::iterator vit = v.begin(); vit != v.end(); vit++) vit->second /= nd;
    }
    if (must) {
        double norm = 0.0;
        if (!l2) { for (std::map<unsigned, double>::iterator it = v.begin(); it != v.end(); ++it) norm += fabs(it->second); }
        else { for (std::map<unsigned, double>::iterator it = v.begin(); it != v.end(); ++it) norm += it->second * it->second; norm = sqrt(norm); }
        if (norm > 0.0) for (std::map<unsigned, double>::iterator it = v.begin(); it != v.end(); ++it) it->second /= norm;
    }
    int nb = 0;
    for (std::map<unsigned, double>::iterator it = v.begin(); it != v.end(); ++it, ++nb) { bow_word[nb] = (int32_t)it->first; bow_value[nb] = it->second; }
    int nf = 0, pos = 0;
    for (std::map<unsigned, std::vector<unsigned> >::iterator it = fv.begin(); it != fv.end(); ++it, ++nf) {
        fv_node[nf] = (int32_t)it->first; fv_start[nf] = pos;
        for (size_t j = 0; j < it->second.size(); ++j) fv_feature[pos++] = (int32_t)it->second[j];
    }
    fv_start[nf] = pos;
    *n_bow = nb; *n_fv = nf;
    return nb;
}

// int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, vector<MapPoint*> &vpMapPointMatches) — reference src/ORBmatcher1.cc:225-427,
// branch F.Nleft == -1 (pKF->mpCamera2 == NULL).  Both FeatureVectors as CSR over ascending node ids (what transform() returns);
// kf_valid[i] = vpMapPointsKF[i] && !isBad(); kf_keys = pKF->mvKeysUn, f_keys = F.mvKeys (angles for the rotation histogram).
// matches[f] = index of the keyframe feature whose map point the frame feature f received, -1 = NULL.  Returns nmatches.
int oracle_search_by_bow(const int32_t *kf_node, const int32_t *kf_start, const int32_t *kf_feat, int kf_nfv, const oracle_keypoint *kf_keys,
                         const uint8_t *kf_desc, const uint8_t *kf_valid, const int32_t *f_node, const int32_t *f_start, const int32_t *f_feat,
                         int f_nfv, const oracle_keypoint *f_keys, const uint8_t *f_desc, int f_n, float nnratio, int check_orientation,
                         int32_t *matches)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < f_n; ++i) matches[i] = -1;
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int KFit = 0, Fit = 0;
    while (KFit != kf_nfv && Fit != f_nfv) {
        if (kf_node[KFit] == f_node[Fit]) {
            for (int iKF = kf_start[KFit]; iKF < kf_start[KFit + 1]; iKF++) {
                const int realIdxKF = kf_feat[iKF];
                if (!kf_valid[realIdxKF]) continue;  // !pMP || pMP->isBad()
                const uint8_t *dKF = kf_desc + (size_t)realIdxKF * 32;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int iF = f_start[Fit]; iF < f_start[Fit + 1]; iF++) {
                    const int realIdxF = f_feat[iF];
                    if (matches[realIdxF] >= 0) continue;  // vpMapPointMatches[realIdxF]
                    const int dist = oracle_hamming256(dKF, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches[bestIdxF] = realIdxKF;
                        if (check_orientation) {
                            float rot = kf_keys[realIdxKF].angle - f_keys[bestIdxF].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(bestIdxF);
                        }
                        nmatches++;
                    }
                }
            }
            KFit++; Fit++;
        } else if (kf_node[KFit] < f_node[Fit]) {
            KFit = (int)(std::lower_bound(kf_node, kf_node + kf_nfv, f_node[Fit]) - kf_node);
        } else {
            Fit = (int)(std::lower_bound(f_node, f_node + f_nfv, kf_node[KFit]) - f_node);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { matches[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

// int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) — reference src/ORBmatcher2.cc:37-177
// (NLeft == -1 on both sides).  valid1 / valid2[i] = the keyframe's map point i exists and !isBad(); matches12[idx1] = idx2 whose
// map point vpMatches12[idx1] becomes, -1 = NULL.  Differences from the frame variant: the second side is masked by valid2 and
// vbMatched2, the distance test is strict (bestDist1 < TH_LOW), the result and the rotation bins are indexed by idx1.
int oracle_search_by_bow_keyframes(const int32_t *node1, const int32_t *start1, const int32_t *feat1, int nfv1, const oracle_keypoint *keys1,
                                   const uint8_t *desc1, const uint8_t *valid1, int n1, const int32_t *node2, const int32_t *start2,
                                   const int32_t *feat2, int nfv2, const oracle_keypoint *keys2, const uint8_t *desc2, const uint8_t *valid2,
                                   int n2, float nnratio, int check_orientation, int32_t *matches12)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    std::vector<bool> vbMatched2((size_t)n2, false);
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    int f1it = 0, f2it = 0;
    while (f1it != nfv1 && f2it != nfv2) {
        if (node1[f1it] == node2[f2it]) {
            for (int i1 = start1[f1it]; i1 < start1[f1it + 1]; i1++) {
                const int idx1 = feat1[i1];
                if (!valid1[idx1]) continue;
                const uint8_t *d1 = desc1 + (size_t)idx1 * 32;
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = start2[f2it]; i2 < start2[f2it + 1]; i2++) {
                    const int idx2 = feat2[i2];
                    if (vbMatched2[idx2] || !valid2[idx2]) continue;
                    const int dist = oracle_hamming256(d1, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW) {
                    if (static_cast<float>(bestDist1) < nnratio * static_cast<float>(bestDist2)) {
                        matches12[idx1] = bestIdx2;
                        vbMatched2[bestIdx2] = true;
                        if (check_orientation) {
                            float rot = keys1[idx1].angle - keys2[bestIdx2].angle;
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)round(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rotHist[bin].push_back(idx1);
                        }
                        nmatches++;
                    }
                }
            }
            f1it++; f2it++;
        } else if (node1[f1it] < node2[f2it]) {
            f1it = (int)(std::lower_bound(node1, node1 + nfv1, node2[f2it]) - node1);
        } else {
            f2it = (int)(std::lower_bound(node2, node2 + nfv2, node1[f1it]) - node2);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    return nmatches;
}

}  // extern "C"

// =====================================================================================================================================
// KannalaBrandt8::TriangulateMatches + Frame::ComputeStereoFishEyeMatches — reference src/CameraModels/KannalaBrandt8.cpp:306-375 (with
// unproject :116-143, project :67-93, Triangulate :393-406) and src/Frame.cc:1156-1196 (second half).
// The reference evaluates this with Eigen (float, expression templates, JacobiSVD) and the platform libm (tanf / atan2f / cos / sin); Eigen
// is not in this image and the reference holds no vectors for it.  What is restated here, operation by operation, and what the GPU kernel
// reproduces bit for bit:
//   * every +, -, *, / of the reference in float, in source order, un-fused; 3-term dot products, squared norms and the rows of fixed-size
//     3 x 3 products as Eigen's unrolled redux x0 + (x1 + x2) (derivation at the projection head below; round 5 - rounds 2-4 summed left
//     to right);
//   * sqrtf correctly rounded; `atan2f` (:69-70) and `std::tan(float)` = tanf (:139) are the REAL libm calls here (glibc 2.35) - the GPU
//     evaluates include/orbx_libm.h, which is pinned to this libm exhaustively (tests/tools/libm_pin.cpp, profiles/r05_libm_pin.txt);
//     `cos(psi)` / `sin(psi)` in project() are unqualified and no using-directive reaches KannalaBrandt8.cpp (its include chain has none;
//     g2o's sits in block_solver.hpp, which it does not include): taken as ::cos(double), the product kept in double and narrowed once
//     when it is stored into the Eigen::Vector2f (:81-82);
//   * Eigen::JacobiSVD<Matrix4f>(A, ComputeFullV).matrixV().col(3) (:402-404) = the right singular vector of the smallest singular value.
//     Restated as a one-sided (Hestenes) Jacobi iteration on the columns of A in DOUBLE: cyclic pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3),
//     exactly 12 sweeps, rotation skipped when |a_p . a_q| <= 1e-300; the column with the smallest norm (first one on ties) gives the
//     vector, narrowed to float.  Its sign cancels in x3D = x3Dh.head(3) / x3Dh(3).  TOLERANCE-LEVEL against the reference: Eigen's float
//     two-sided Jacobi sweep cannot be reproduced bit for bit without Eigen; depth / 3-D point agree with a float SVD to a few ulp of the
//     conditioning, and every decision taken from them (z > 0, the reprojection gates) is the same away from its threshold.
// =====================================================================================================================================
static void kb8Unproject(const float *P, float precision, float px, float py, float r[3])
{
    const float pwx = (px - P[2]) / P[0], pwy = (py - P[3]) / P[1];
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf((float)(-M_PI / 2.f), theta_d), (float)(M_PI / 2.f));
    if ((double)theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = P[4] * theta2, k1_theta4 = P[5] * theta4;
            const float k2_theta6 = P[6] * theta6, k3_theta8 = P[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < precision) break;
        }
        scale = tanf(theta) / theta_d;                                             // std::tan(float), :139
    }
    r[0] = pwx * scale; r[1] = pwy * scale; r[2] = 1.f;
}

static void kb8Project(const float *P, const float v[3], float uv[2])
{
    const float x2_plus_y2 = v[0] * v[0] + v[1] * v[1];
    const float theta = atan2f(sqrtf(x2_plus_y2), v[2]);                           // :69
    const float psi = atan2f(v[1], v[0]);                                          // :70
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2, theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + P[4] * theta3 + P[5] * theta5 + P[6] * theta7 + P[7] * theta9;
    uv[0] = (float)((double)(P[0] * r) * cos((double)psi) + (double)P[2]);
    uv[1] = (float)((double)(P[1] * r) * sin((double)psi) + (double)P[3]);
}

// Eigen's 3-term redux (dot, squaredNorm, a row of a fixed-size 3 x 3 product): x0 + (x1 + x2), see the projection head below
static inline float dot3f(const float *a, const float *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }

// =====================================================================================================================================
// Projection head of the two tracking searches: reference src/ORBmatcher3.cc:281-296 (motion model: Sophus::SE3f * point, invzc, project)
// and src/Frame.cc:542-600 (Frame::isInFrustum, Nleft == -1) with MapPoint::PredictScale (src/MapPoint.cc:531-546).
// PARITY UNPINNED for the evaluation order: Eigen / Sophus are not in this image.  Restated from their sources:
//   * Eigen evaluates a fixed-size 3 x 3 times 3 x 1 product coefficient by coefficient (product_type_selector<Small, 1, Small> =
//     CoeffBasedProductMode), each coefficient as (row.transpose().cwiseProduct(col)).sum(), and sum() / dot() / squaredNorm() of three
//     terms through the unrolled redux redux_novec_unroller<Func, Eval, 0, 3> = func(x0, func(x1, x2)): x0 + (x1 + x2);
//   * cross() of two 3-vectors: (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0);
//   * Sophus SO3 * p: uv = q.vec().cross(p); uv += uv; return p + q.w() * uv + q.vec().cross(uv); SE3 * p adds the translation;
//   * all of it in float, un-fused (the reference is built with -O3 and no -march: no FMA contraction); `1.0 / x3Dc(2)` in double,
//     narrowed by the assignment to `const float invzc`; norm() = std::sqrt(float) correctly rounded.
// (KannalaBrandt8::TriangulateMatches above follows the same derivation since round 5.)
// =====================================================================================================================================
static inline float sum3e(float x0, float x1, float x2) { return x0 + (x1 + x2); }
static inline void cross3e(const float *a, const float *b, float *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
static void cameraProject(int type, const float *P, const float v[3], float uv[2])
{
    if (type == 0) {  // Pinhole::project(const Eigen::Vector3f &), src/CameraModels/Pinhole.cpp:43-49
        uv[0] = P[0] * v[0] / v[2] + P[2];
        uv[1] = P[1] * v[1] / v[2] + P[3];
    } else kb8Project(P, v, uv);
}

// the float -> int conversion of an out-of-range or NaN value is undefined in C++; x86-64's cvttss2si returns INT_MIN for it, which the
// first clamp turns into level 0 - restated so (dist == 0 gives ratio = inf; such a point fails the distance test anyway)
extern "C" int oracle_predict_scale(float ratio, float log_scale_factor, int nlevels)
{
    const float lv = std::ceil(std::log(ratio) / log_scale_factor);
    int nScale = (lv >= -2147483648.0f && lv < 2147483648.0f) ? (int)lv : INT32_MIN;
    if (nScale < 0) nScale = 0;
    else if (nScale >= nlevels) nScale = nlevels - 1;
    return nScale;
}

extern "C" void oracle_project_last_frame_points(const float *q, const float *t, int cam_type, const float *cam, const oracle_world_point *pts, int np,
                                                 oracle_projection *out)
{
    for (int i = 0; i < np; ++i) {
        const float *p = pts[i].pos;
        float uv[3], c2[3], x[3];
        cross3e(q, p, uv);                                  // so3.hpp:364
        for (int k = 0; k < 3; ++k) uv[k] = uv[k] + uv[k];  // :365
        cross3e(q, uv, c2);
        for (int k = 0; k < 3; ++k) x[k] = ((p[k] + q[3] * uv[k]) + c2[k]) + t[k];  // :366, se3.hpp (so3() * p + translation())
        oracle_projection o;
        o.invz = (float)(1.0 / (double)x[2]);               // src/ORBmatcher3.cc:288
        float w[2];
        cameraProject(cam_type, cam, x, w);                 // :293
        o.u = w[0]; o.v = w[1];
        o.angle = pts[i].angle; o.octave = pts[i].octave; o.flags = pts[i].flags;
        out[i] = o;
    }
}

// one camera of Frame::isInFrustum / Frame::isInFrustumChecks (src/Frame.cc:549-599, 1198-1272): R, t = the world -> camera pose of THAT
// camera, twc its centre.  The checks return early in the reference; computed to the end here, the values that are stored when all pass are
// the same.  Returns mbTrackInView; uv, invz, viewCos, level, PcDist as the reference forms them.
struct FrustumSide { bool in; float uv[2], invz, viewCos, PcDist; int level; };
static FrustumSide frustumSide(const float *R, const float *t, const float *twc, int cam_type, const float *cam, const float *bounds,
                               float viewing_cos_limit, float log_scale_factor, int nlevels, const oracle_world_point &pt)
{
    FrustumSide o;
    const float *P = pt.pos;
    float Pc[3];
    for (int k = 0; k < 3; ++k) Pc[k] = sum3e(R[3 * k] * P[0], R[3 * k + 1] * P[1], R[3 * k + 2] * P[2]) + t[k];  // src/Frame.cc:553
    o.PcDist = sqrtf(sum3e(Pc[0] * Pc[0], Pc[1] * Pc[1], Pc[2] * Pc[2]));                                         // :554
    const float PcZ = Pc[2];
    o.invz = 1.0f / PcZ;                                                                                          // :558
    bool in = !(PcZ < 0.0f);                                                                                      // :559-560
    cameraProject(cam_type, cam, Pc, o.uv);                                                                       // :562
    if (o.uv[0] < bounds[0] || o.uv[0] > bounds[1]) in = false;                                                   // :564-567
    if (o.uv[1] < bounds[2] || o.uv[1] > bounds[3]) in = false;
    const float maxDistance = 1.2f * pt.max_distance, minDistance = 0.8f * pt.min_distance;                      // src/MapPoint.cc
    const float PO[3] = {P[0] - twc[0], P[1] - twc[1], P[2] - twc[2]};                                            // :575
    const float dist = sqrtf(sum3e(PO[0] * PO[0], PO[1] * PO[1], PO[2] * PO[2]));                                 // :576
    if (dist < minDistance || dist > maxDistance) in = false;                                                     // :578-579
    const float *Pn = pt.normal;
    o.viewCos = sum3e(PO[0] * Pn[0], PO[1] * Pn[1], PO[2] * Pn[2]) / dist;                                        // :584
    if (o.viewCos < viewing_cos_limit) in = false;                                                                // :586-587
    // MapPoint::PredictScale(dist, pF), src/MapPoint.cc:531-546: `int nScale = ceil(log(ratio) / pF->mfLogScaleFactor)` with float ratio and
    // float mfLogScaleFactor under `using namespace std` (TemplatedVocabulary.h:36 through MapPoint.h:23): std::log(float) = logf, a float
    // division, std::ceil(float), then the conversion to int and the two clamps.
    const float ratio = pt.max_distance / dist;
    o.level = oracle_predict_scale(ratio, log_scale_factor, nlevels);
    o.in = in;
    return o;
}

extern "C" void oracle_project_in_frustum(const float *R, const float *t, const float *Ow, int cam_type, const float *cam, const float *bounds,
                                          float mbf, float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points,
                                          float th_far_points, const oracle_world_point *pts, int np, oracle_projection *out)
{
    for (int i = 0; i < np; ++i) {
        const FrustumSide sd = frustumSide(R, t, Ow, cam_type, cam, bounds, viewing_cos_limit, log_scale_factor, nlevels, pts[i]);
        bool in = sd.in;
        if (far_points && sd.PcDist > th_far_points) in = false;                                                  // src/ORBmatcher1.cc:57-58
        oracle_projection o;
        o.u = sd.uv[0]; o.v = sd.uv[1];
        o.invz = sd.uv[0] - mbf * sd.invz;                                                                        // mTrackProjXR, :594
        o.angle = sd.viewCos; o.octave = sd.level;
        o.flags = (pts[i].flags & ~1) | (((pts[i].flags & 1) && in) ? 1 : 0);
        out[i] = o;
    }
}

// Frame::isInFrustum for a fisheye rig (the Nleft != -1 branch, src/Frame.cc:604-614): isInFrustumChecks(pMP, limit) for the left camera and
// isInFrustumChecks(pMP, limit, true) for the right one (:1198-1272).  The right camera's pose is formed per call in the reference (:1204-1210):
//   mR = Rrl * mRcw;  mt = Rrl * mtcw + trl;  twc = mRwc * mTlr.translation() + mOw      (Rrl / trl = mTrl.rotationMatrix() / .translation())
// fixed-size Eigen products, coefficient by coefficient with the 3-term redux x0 + (x1 + x2) (see the derivation above).
// Records: {mTrackProjX, mTrackProjY, mTrackDepth, mTrackViewCos, mnTrackScaleLevel, flags} and the same with the R members; flags bit 0 =
// the input bit && mbTrackInView(R) && !(far_points && mTrackDepth > th_far_points) - the far-point test of ORBmatcher::SearchByProjection
// (src/ORBmatcher1.cc:57-58) reads the LEFT camera's mTrackDepth for both sides.  DEVIATION, stated: the reference stores mTrackDepth only when
// the left check passes to its end, so for a point the left camera rejects it compares whatever an EARLIER frame left in the MapPoint; here
// that test always uses the left camera's Pc_dist of this frame (which isInFrustumChecks computes before its first check).
extern "C" void oracle_project_in_frustum_fisheye(const float *Rcw, const float *tcw, const float *Ow, const float *Rrl, const float *trl,
                                                  const float *tlr, const float *Rwc, const float *cam1, const float *cam2, const float *bounds,
                                                  float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points, float th_far_points,
                                                  const oracle_world_point *pts, int np, oracle_projection *out_left, oracle_projection *out_right)
{
    float mR[9], mt[3], twc[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) mR[3 * i + j] = sum3e(Rrl[3 * i] * Rcw[j], Rrl[3 * i + 1] * Rcw[3 + j], Rrl[3 * i + 2] * Rcw[6 + j]);
        mt[i] = sum3e(Rrl[3 * i] * tcw[0], Rrl[3 * i + 1] * tcw[1], Rrl[3 * i + 2] * tcw[2]) + trl[i];
        twc[i] = sum3e(Rwc[3 * i] * tlr[0], Rwc[3 * i + 1] * tlr[1], Rwc[3 * i + 2] * tlr[2]) + Ow[i];
    }
    for (int i = 0; i < np; ++i) {
        const FrustumSide l = frustumSide(Rcw, tcw, Ow, 1, cam1, bounds, viewing_cos_limit, log_scale_factor, nlevels, pts[i]);
        const FrustumSide r = frustumSide(mR, mt, twc, 1, cam2, bounds, viewing_cos_limit, log_scale_factor, nlevels, pts[i]);
        const bool live = (pts[i].flags & 1) && !(far_points && l.PcDist > th_far_points);
        oracle_projection o;
        o.u = l.uv[0]; o.v = l.uv[1]; o.invz = l.PcDist; o.angle = l.viewCos; o.octave = l.level;
        o.flags = (pts[i].flags & ~1) | ((live && l.in) ? 1 : 0);
        out_left[i] = o;
        o.u = r.uv[0]; o.v = r.uv[1]; o.invz = r.PcDist; o.angle = r.viewCos; o.octave = r.level;
        o.flags = (pts[i].flags & ~1) | ((live && r.in) ? 1 : 0);
        out_right[i] = o;
    }
}

// The projection head of the keyframe-side searches - ORBmatcher::Fuse (src/ORBmatcher2.cc:446-520 and :640-688), SearchByProjection(KeyFrame*,
// Sim3, ..) (src/ORBmatcher1.cc:451-494, :557-600) and SearchByProjection(Frame&, KeyFrame*, ..) (src/ORBmatcher3.cc:486-519): per map point
//   p3Dc = Tcw * p3Dw (Sophus::SE3f: quaternion form, as oracle_project_last_frame_points); [p3Dc(2) < 0 -> skip]; invz = 1 / p3Dc(2);
//   uv = pCamera->project(p3Dc); IsInImage(uv) (KeyFrame: x >= mnMinX && x < mnMaxX .., include/KeyFrame.h; Frame: uv < min || uv > max rejects);
//   ur = uv(0) - bf * invz; PO = p3Dw - Ow; dist = PO.norm(); [dist < 0.8 min || dist > 1.2 max -> skip]; [PO.dot(Pn) < 0.5 * dist -> skip:
//   `0.5 * dist3D` is a double product, the float dot product is promoted for the comparison]; nPredictedLevel = PredictScale(dist, pKF);
//   radius = th * mvScaleFactors[nPredictedLevel].
// tests: bit 0 the positive-depth test (Fuse, Sim3), bit 1 the viewing-angle test (Fuse, Sim3), bit 2 Frame-style bounds (SearchByProjection(Frame,
// KeyFrame)).  out[i] = {u, v, ur, radius, level, search}; search = the point's flags bit 0 (the caller's isBad / IsInKeyFrame / already-found
// tests) && every test passed.
extern "C" void oracle_project_window_points(const float *q, const float *t, const float *Ow, int cam_type, const float *cam, const float *bounds,
                                             float bf, float th, const float *scale_factors, float log_scale_factor, int nlevels, int tests,
                                             const oracle_world_point *pts, int np, void *out_)
{
    oracle_window_point *out = (oracle_window_point *)out_;
    for (int i = 0; i < np; ++i) {
        const float *p = pts[i].pos;
        float uvx[3], c2[3], x[3];
        cross3e(q, p, uvx);
        for (int k = 0; k < 3; ++k) uvx[k] = uvx[k] + uvx[k];
        cross3e(q, uvx, c2);
        for (int k = 0; k < 3; ++k) x[k] = ((p[k] + q[3] * uvx[k]) + c2[k]) + t[k];
        bool ok = (pts[i].flags & 1) != 0;
        if ((tests & 1) && x[2] < 0.0f) ok = false;
        const float invz = 1 / x[2];                      // `const float invz = 1/p3Dc(2)`: int 1 converted to float, a float division
        float uv[2];
        cameraProject(cam_type, cam, x, uv);
        if (tests & 4) { if (uv[0] < bounds[0] || uv[0] > bounds[1] || uv[1] < bounds[2] || uv[1] > bounds[3]) ok = false; }
        else if (!(uv[0] >= bounds[0] && uv[0] < bounds[1] && uv[1] >= bounds[2] && uv[1] < bounds[3])) ok = false;
        const float ur = uv[0] - bf * invz;
        const float maxDistance = 1.2f * pts[i].max_distance, minDistance = 0.8f * pts[i].min_distance;
        const float PO[3] = {p[0] - Ow[0], p[1] - Ow[1], p[2] - Ow[2]};
        const float dist = sqrtf(sum3e(PO[0] * PO[0], PO[1] * PO[1], PO[2] * PO[2]));
        if (dist < minDistance || dist > maxDistance) ok = false;
        const float *Pn = pts[i].normal;
        if ((tests & 2) && (double)sum3e(PO[0] * Pn[0], PO[1] * Pn[1], PO[2] * Pn[2]) < 0.5 * (double)dist) ok = false;
        const int level = oracle_predict_scale(pts[i].max_distance / dist, log_scale_factor, nlevels);
        oracle_window_point o;
        o.u = uv[0]; o.v = uv[1]; o.ur = ur; o.radius = th * scale_factors[level]; o.level = level; o.search = ok ? 1 : 0;
        out[i] = o;
    }
}

// right singular vector of the smallest singular value of the 4 x 4 matrix A (row-major), see the header comment
static void smallestRightSingularVector4(const float A[16], float v[4])
{
    double U[4][4], V[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { U[i][j] = (double)A[i * 4 + j]; V[i][j] = i == j ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 12; ++sweep)
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < 4; ++i) { alpha += U[i][p] * U[i][p]; beta += U[i][q] * U[i][q]; gamma += U[i][p] * U[i][q]; }
                if (fabs(gamma) <= 1e-300) continue;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < 4; ++i) {
                    const double up = U[i][p], uq = U[i][q];
                    U[i][p] = c * up - s * uq; U[i][q] = s * up + c * uq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
                }
            }
    int best = 0;
    double bestN = 0;
    for (int j = 0; j < 4; ++j) {
        double n = 0;
        for (int i = 0; i < 4; ++i) n += U[i][j] * U[i][j];
        if (j == 0 || n < bestN) { bestN = n; best = j; }
    }
    for (int i = 0; i < 4; ++i) v[i] = (float)V[i][best];
}

extern "C" float oracle_kb8_triangulate_matches(const float *cam1, const float *cam2, float precision, float x1, float y1, float x2, float y2,
                                     const float *R12, const float *t12, float sigmaLevel, float unc, float *p3D)
{
    float r1[3], r2[3];
    kb8Unproject(cam1, precision, x1, y1, r1);
    kb8Unproject(cam2, precision, x2, y2, r2);
    float r21[3];
    for (int i = 0; i < 3; ++i) r21[i] = dot3f(R12 + 3 * i, r2);
    const float cosParallaxRays = dot3f(r1, r21) / (sqrtf(dot3f(r1, r1)) * sqrtf(dot3f(r21, r21)));
    if ((double)cosParallaxRays > 0.9998) return -1;
    float R21[9], Tcw2[12];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R21[i * 3 + j] = R12[j * 3 + i];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) Tcw2[i * 4 + j] = R21[i * 3 + j];
        const float nr[3] = {-R21[i * 3], -R21[i * 3 + 1], -R21[i * 3 + 2]};
        Tcw2[i * 4 + 3] = dot3f(nr, t12);
    }
    static const float Tcw1[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    float A[16];
    for (int j = 0; j < 4; ++j) {
        A[0 * 4 + j] = r1[0] * Tcw1[2 * 4 + j] - Tcw1[0 * 4 + j];
        A[1 * 4 + j] = r1[1] * Tcw1[2 * 4 + j] - Tcw1[1 * 4 + j];
        A[2 * 4 + j] = r2[0] * Tcw2[2 * 4 + j] - Tcw2[0 * 4 + j];
        A[3 * 4 + j] = r2[1] * Tcw2[2 * 4 + j] - Tcw2[1 * 4 + j];
    }
    float x3Dh[4];
    smallestRightSingularVector4(A, x3Dh);
    const float x3D[3] = {x3Dh[0] / x3Dh[3], x3Dh[1] / x3Dh[3], x3Dh[2] / x3Dh[3]};
    const float z1 = x3D[2];
    if (!(z1 > 0)) return -2;   // `if (z1 <= 0)`: a NaN depth falls through there and fails `depth > 0.0001f` at the call site all the same
    const float z2 = dot3f(R21 + 6, x3D) + Tcw2[2 * 4 + 3];
    if (!(z2 > 0)) return -3;
    float uv1[2];
    kb8Project(cam1, x3D, uv1);
    const float errX1 = uv1[0] - x1, errY1 = uv1[1] - y1;
    if ((double)(errX1 * errX1 + errY1 * errY1) > 5.991 * (double)sigmaLevel) return -4;
    float x3D2[3];
    for (int i = 0; i < 3; ++i) x3D2[i] = dot3f(R21 + 3 * i, x3D) + Tcw2[i * 4 + 3];
    float uv2[2];
    kb8Project(cam2, x3D2, uv2);
    const float errX2 = uv2[0] - x2, errY2 = uv2[1] - y2;
    if ((double)(errX2 * errX2 + errY2 * errY2) > 5.991 * (double)unc) return -5;
    p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2];
    return z1;
}

// Frame::ComputeStereoFishEyeMatches — reference src/Frame.cc:1156-1196.  kl / dl / nl / mono_left: mvKeys, mDescriptors, Nleft, monoLeft
// (likewise right); level_sigma2 = mvLevelSigma2.  Outputs sized nl / nr: mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints
// (3 floats per left keypoint; untouched entries are left as they were).  Returns nMatches.
extern "C" int oracle_fisheye_stereo_matches(const oracle_keypoint *kl, const uint8_t *dl, int nl, int mono_left, const oracle_keypoint *kr,
                                  const uint8_t *dr, int nr, int mono_right, const float *cam1, const float *cam2, float precision,
                                  const float *R12, const float *t12, const float *level_sigma2, int32_t *left_to_right,
                                  int32_t *right_to_left, float *depth, float *p3d)
{
    for (int i = 0; i < nl; ++i) { left_to_right[i] = -1; depth[i] = -1.0f; }
    for (int i = 0; i < nr; ++i) right_to_left[i] = -1;
    const int nq = nl - mono_left, nt = nr - mono_right;
    if (nq <= 0) return 0;
    std::vector<int32_t> idx((size_t)nq * 2), dist((size_t)nq * 2);
    oracle_knn2(dl + (size_t)mono_left * 32, nq, dr + (size_t)mono_right * 32, std::max(nt, 0), idx.data(), dist.data());
    int nMatches = 0;
    for (int i = 0; i < nq; ++i) {
        if (idx[2 * i + 1] < 0) continue;                                                  // (*it).size() >= 2
        if (!((double)(float)dist[2 * i] < (double)(float)dist[2 * i + 1] * 0.7)) continue;  // Lowe's ratio, DMatch::distance is float
        const int iL = i + mono_left, iR = idx[2 * i] + mono_right;  // queryIdx / trainIdx are relative to the rowRange
        const float sigma1 = level_sigma2[kl[iL].octave], sigma2 = level_sigma2[kr[iR].octave];
        float p[3];
        const float d = oracle_kb8_triangulate_matches(cam1, cam2, precision, kl[iL].x, kl[iL].y, kr[iR].x, kr[iR].y, R12, t12, sigma1, sigma2, p);
        if (d > 0.0001f) {
            left_to_right[iL] = iR;
            right_to_left[iR] = iL;
            p3d[3 * iL] = p[0]; p3d[3 * iL + 1] = p[1]; p3d[3 * iL + 2] = p[2];
            depth[iL] = d;
            nMatches++;
        }
    }
    return nMatches;
}

extern "C" {

void oracle_synth_frame(uint64_t seed, int w, int h, uint8_t *out, ptrdiff_t stride)
{
    orbx_synth_frame_impl(seed, w, h, out, stride);
}

}  // extern "C"

// =====================================================================================================================================
// ORBmatcher::SearchForTriangulation — reference src/ORBmatcher2.cc:179-418, with Pinhole::epipolarConstrain
// (src/CameraModels/Pinhole.cpp:107-129) and KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220).  The pose
// algebra in front of the loop (:185-214: epipole, T12 / Tll .. Trr, F12 = K1^-T [t12]x R12 K2^-1) is Sophus / Eigen code of the caller
// and arrives as `g`; everything per feature pair is restated here in the reference's float operations, in source order, un-fused.
// =====================================================================================================================================
extern "C" int oracle_search_for_triangulation(const oracle_keyframe_view *kf1, const oracle_keyframe_view *kf2,
                                               const oracle_triangulation_geometry *g, int only_stereo, int coarse, int check_orientation,
                                               int32_t *matches12, int32_t *pairs)
{
    const int TH_LOW = 50, HISTO_LENGTH = 30;
    const bool rig = g->rig != 0;  // pKF1->mpCamera2 && pKF2->mpCamera2
    int nmatches = 0;
    for (int i = 0; i < kf1->n; ++i) matches12[i] = -1;     // vector<int> vMatches12(pKF1->N, -1)   (:223)
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    int f1it = 0, f2it = 0;
    const int f1end = kf1->nfv, f2end = kf2->nfv;
    while (f1it != f1end && f2it != f2end) {
        if (kf1->fv_node[f1it] == kf2->fv_node[f2it]) {
            for (int i1 = kf1->fv_start[f1it]; i1 < kf1->fv_start[f1it + 1]; i1++) {
                const int idx1 = kf1->fv_feature[i1];
                if (kf1->has_map_point[idx1]) continue;                                   // :246-252
                const bool bStereo1 = !rig && kf1->u_right && kf1->u_right[idx1] >= 0;     // :254 (!pKF1->mpCamera2 && mvuRight[idx1] >= 0)
                if (only_stereo && !bStereo1) continue;
                const oracle_keypoint &kp1 = kf1->keys[idx1];                              // :259-261
                const bool bRight1 = !(kf1->n_left == -1 || idx1 < kf1->n_left);
                const uint8_t *d1 = kf1->desc + (size_t)idx1 * 32;
                int bestDist = TH_LOW, bestIdx2 = -1;
                for (int i2 = kf2->fv_start[f2it]; i2 < kf2->fv_start[f2it + 1]; i2++) {
                    const int idx2 = kf2->fv_feature[i2];
                    if (kf2->has_map_point[idx2]) continue;                               // :275-279 (vbMatched2 is never set)
                    const bool bStereo2 = !rig && kf2->u_right && kf2->u_right[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = oracle_hamming256(d1, kf2->desc + (size_t)idx2 * 32);
                    if (dist > TH_LOW || dist > bestDist) continue;                       // :293
                    const oracle_keypoint &kp2 = kf2->keys[idx2];
                    const bool bRight2 = !(kf2->n_left == -1 || idx2 < kf2->n_left);
                    if (!bStereo1 && !bStereo2 && !rig) {                                 // :302-311
                        const float distex = g->ep[0] - kp2.x;
                        const float distey = g->ep[1] - kp2.y;
                        if (distex * distex + distey * distey < 100 * kf2->scale_factors[kp2.octave]) continue;
                    }
                    bool ok = coarse != 0;
                    if (!ok) {
                        const float sigmaLevel = kf1->level_sigma2[kp1.octave], unc = kf2->level_sigma2[kp2.octave];
                        if (g->model == 0) {  // Pinhole::epipolarConstrain
                            const float *F = g->F12;  // F12(i, j) = F[3 * i + j]
                            const float a = kp1.x * F[0] + kp1.y * F[3] + F[6];
                            const float b = kp1.x * F[1] + kp1.y * F[4] + F[7];
                            const float c = kp1.x * F[2] + kp1.y * F[5] + F[8];
                            const float num = a * kp2.x + b * kp2.y + c;
                            const float den = a * a + b * b;
                            if (den == 0) ok = false;
                            else {
                                const float dsqr = num * num / den;
                                ok = dsqr < 3.84 * unc;
                            }
                        } else {  // KannalaBrandt8::epipolarConstrain: TriangulateMatches(...) > 0.0001f
                            const int k = rig ? 2 * (int)bRight1 + (int)bRight2 : 0;
                            const float *camA = rig && bRight1 ? g->cam_right : g->cam_left, *camB = rig && bRight2 ? g->cam_right : g->cam_left;
                            float p3D[3];
                            ok = oracle_kb8_triangulate_matches(camA, camB, g->precision, kp1.x, kp1.y, kp2.x, kp2.y, g->R[k], g->t[k], sigmaLevel, unc,
                                                                p3D) > 0.0001f;
                        }
                    }
                    if (ok) { bestIdx2 = idx2; bestDist = dist; }
                }
                if (bestIdx2 >= 0) {
                    const oracle_keypoint &kp2 = kf2->keys[bestIdx2];
                    matches12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1.angle - kp2.angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)round(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rotHist[bin].push_back(idx1);
                    }
                }
            }
            f1it++; f2it++;
        } else if (kf1->fv_node[f1it] < kf2->fv_node[f2it]) {
            f1it = (int)(std::lower_bound(kf1->fv_node, kf1->fv_node + f1end, kf2->fv_node[f2it]) - kf1->fv_node);
        } else {
            f2it = (int)(std::lower_bound(kf2->fv_node, kf2->fv_node + f2end, kf1->fv_node[f1it]) - kf2->fv_node);
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++) {
            if (i == ind1 || i == ind2 || i == ind3) continue;
            for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { matches12[rotHist[i][j]] = -1; nmatches--; }
        }
    }
    if (pairs) {
        int k = 0;
        for (int i = 0; i < kf1->n; i++) {
            if (matches12[i] < 0) continue;
            pairs[2 * k] = i; pairs[2 * k + 1] = matches12[i]; ++k;
        }
    }
    return nmatches;
}

// =====================================================================================================================================
// The window search of ORBmatcher::Fuse (src/ORBmatcher2.cc:522-579 with the reprojection test, :690-714 without) and of both halves of
// ORBmatcher::SearchBySim3 (src/ORBmatcher3.cc:120-145, 188-213), on KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:704-748 = the frame's
// grid walk without a level filter).  The keyframe's grid is the frame's (KeyFrame copies F.mGrid, src/KeyFrame.cc:67-75).
// =====================================================================================================================================
extern "C" int oracle_search_keyframe_windows(const oracle_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x,
                                              float max_x, float min_y, float max_y, const float *inv_level_sigma2,
                                              const oracle_window_point *points, const uint8_t *point_desc, int np, int chi2_test,
                                              int32_t *best_idx, int32_t *best_dist)
{
    const FrameGrid G = assignFeaturesToGrid(keys, n, min_x, max_x, min_y, max_y);
    for (int i = 0; i < np; ++i) {
        best_idx[i] = -1; best_dist[i] = INT_MAX;
        const oracle_window_point &P = points[i];
        if (!P.search) continue;
        const int nPredictedLevel = P.level;
        const std::vector<size_t> vIndices = getFeaturesInArea(G, keys, P.u, P.v, P.radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = point_desc + (size_t)i * 32;
        int bestDist = INT_MAX, bestIdx = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
            const size_t idx = *vit;
            const oracle_keypoint &kp = keys[idx];
            const int kpLevel = kp.octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            if (chi2_test) {  // Fuse(pKF, vpMapPoints, th, bRight) only (:545-569)
                if (u_right && u_right[idx] >= 0) {
                    const float ex = P.u - kp.x, ey = P.v - kp.y, er = P.ur - u_right[idx];
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * inv_level_sigma2[kpLevel] > 7.8) continue;
                } else {
                    const float ex = P.u - kp.x, ey = P.v - kp.y;
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * inv_level_sigma2[kpLevel] > 5.99) continue;
                }
            }
            const int dist = oracle_hamming256(dMP, desc + idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        best_idx[i] = bestIdx; best_dist[i] = bestDist;
    }
    return 0;
}

// ORBmatcher::SearchBySim3 — reference src/ORBmatcher3.cc:37-254: both directed searches with TH_HIGH, then the agreement check (:236-250).
extern "C" int oracle_search_by_sim3(const oracle_keypoint *keys1, const uint8_t *kf_desc1, int n1, const float *bounds1,
                                     const oracle_keypoint *keys2, const uint8_t *kf_desc2, int n2, const float *bounds2,
                                     const oracle_window_point *points_1in2, const uint8_t *mp_desc1, const oracle_window_point *points_2in1,
                                     const uint8_t *mp_desc2, int32_t *matches12)
{
    const int TH_HIGH = 100;
    std::vector<int32_t> vnMatch1((size_t)n1, -1), vnMatch2((size_t)n2, -1), bi((size_t)std::max(n1, n2)), bd((size_t)std::max(n1, n2));
    oracle_search_keyframe_windows(keys2, kf_desc2, nullptr, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3], nullptr, points_1in2, mp_desc1, n1, 0,
                                   bi.data(), bd.data());
    for (int i1 = 0; i1 < n1; i1++)
        if (bd[i1] <= TH_HIGH) vnMatch1[i1] = bi[i1];
    oracle_search_keyframe_windows(keys1, kf_desc1, nullptr, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3], nullptr, points_2in1, mp_desc2, n2, 0,
                                   bi.data(), bd.data());
    for (int i2 = 0; i2 < n2; i2++)
        if (bd[i2] <= TH_HIGH) vnMatch2[i2] = bi[i2];
    int nFound = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        matches12[i1] = -1;
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) { matches12[i1] = idx2; nFound++; }
        }
    }
    return nFound;
}

// MapPoint::ComputeDistinctiveDescriptors — reference src/MapPoint.cc:364-398 (the part after the observations have been collected)
extern "C" int oracle_distinctive_descriptors(const uint8_t *desc, const int32_t *start, int npoints, int32_t *best)
{
    for (int p = 0; p < npoints; ++p) {
        const size_t N = (size_t)(start[p + 1] - start[p]);
        best[p] = -1;
        if (N == 0) continue;                                        // if(vDescriptors.empty()) return;
        const uint8_t *D = desc + (size_t)start[p] * 32;
        std::vector<float> Distances(N * N);                          // float Distances[N][N];
        for (size_t i = 0; i < N; i++) {
            Distances[i * N + i] = 0;
            for (size_t j = i + 1; j < N; j++) {
                const int distij = oracle_hamming256(D + i * 32, D + j * 32);
                Distances[i * N + j] = distij;
                Distances[j * N + i] = distij;
            }
        }
        int BestMedian = INT_MAX, BestIdx = 0;
        for (size_t i = 0; i < N; i++) {
            std::vector<int> vDists(Distances.begin() + i * N, Distances.begin() + (i + 1) * N);
            std::sort(vDists.begin(), vDists.end());
            const int median = vDists[(size_t)(0.5 * (N - 1))];
            if (median < BestMedian) { BestMedian = median; BestIdx = (int)i; }
        }
        best[p] = BestIdx;
    }
    return 0;
}

// ORBmatcher::SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, vpMatched, th, ratioHamming) — reference src/ORBmatcher1.cc:429-533
// (and :535-648, which only adds vpMatchedKF): the loop from GetFeaturesInArea on; vpMatched is updated while the points are walked.
extern "C" int oracle_search_by_projection_sim3(const oracle_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x, float min_y,
                                                float max_y, const oracle_window_point *points, const uint8_t *point_desc, int np,
                                                const uint8_t *matched, float ratio_hamming, int32_t *new_match)
{
    const int TH_LOW = 50;
    const FrameGrid G = assignFeaturesToGrid(keys, n, min_x, max_x, min_y, max_y);
    std::vector<char> vpMatched((size_t)n);
    for (int i = 0; i < n; ++i) { vpMatched[i] = matched && matched[i] ? 1 : 0; new_match[i] = -1; }
    int nmatches = 0;
    for (int iMP = 0; iMP < np; iMP++) {
        const oracle_window_point &P = points[iMP];
        if (!P.search) continue;
        const int nPredictedLevel = P.level;
        const std::vector<size_t> vIndices = getFeaturesInArea(G, keys, P.u, P.v, P.radius, -1, -1);
        if (vIndices.empty()) continue;
        const uint8_t *dMP = point_desc + (size_t)iMP * 32;
        int bestDist = 256, bestIdx = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices.begin(), vend = vIndices.end(); vit != vend; vit++) {
            const size_t idx = *vit;
            if (vpMatched[idx]) continue;
            const int kpLevel = keys[idx].octave;
            if (kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel) continue;
            const int dist = oracle_hamming256(dMP, desc + idx * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx = (int)idx; }
        }
        if (bestDist <= TH_LOW * ratio_hamming) {
            vpMatched[bestIdx] = 1;
            new_match[bestIdx] = iMP;
            nmatches++;
        }
    }
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) — reference src/ORBmatcher3.cc:469-590, from
// `GetFeaturesInArea(uv(0), uv(1), radius, nPredictedLevel-1, nPredictedLevel+1)` (:513) on.
extern "C" int oracle_search_by_projection_keyframe(const oracle_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x, float min_y,
                                                    float max_y, const oracle_window_point *points, const uint8_t *point_desc,
                                                    const float *point_angle, int np, const uint8_t *has_map_point, int orb_dist,
                                                    int check_orientation, int32_t *new_match)
{
    const int HISTO_LENGTH = 30;
    const FrameGrid G = assignFeaturesToGrid(keys, n, min_x, max_x, min_y, max_y);
    std::vector<char> mvpMapPoints((size_t)n);
    for (int i = 0; i < n; ++i) { mvpMapPoints[i] = has_map_point && has_map_point[i] ? 1 : 0; new_match[i] = -1; }
    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    for (int i = 0; i < np; i++) {
        const oracle_window_point &P = points[i];
        if (!P.search) continue;
        const int nPredictedLevel = P.level;
        const std::vector<size_t> vIndices2 = getFeaturesInArea(G, keys, P.u, P.v, P.radius, nPredictedLevel - 1, nPredictedLevel + 1);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = point_desc + (size_t)i * 32;
        int bestDist = 256, bestIdx2 = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices2.begin(); vit != vIndices2.end(); vit++) {
            const size_t i2 = *vit;
            if (mvpMapPoints[i2]) continue;
            const int dist = oracle_hamming256(dMP, desc + i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= orb_dist) {
            mvpMapPoints[bestIdx2] = 1;
            new_match[bestIdx2] = i;
            nmatches++;
            if (check_orientation) {
                float rot = point_angle[i] - keys[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        int sizes[HISTO_LENGTH];
        for (int i = 0; i < HISTO_LENGTH; i++) sizes[i] = (int)rotHist[i].size();
        computeThreeMaxima(sizes, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) { new_match[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}
