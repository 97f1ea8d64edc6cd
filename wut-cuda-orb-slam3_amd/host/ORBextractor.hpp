// ORBextractor.hpp — C++ host-side mirror of the reference's operator interface over the C ABI (include/orbx.h).
//
// Same names, argument meaning and error behaviour as the reference for this path:
//   ORB_SLAM3::ORBextractor            reference include/ORBextractor.h:52-113, src/ORBextractor.cc:410-468, 1227-1307
//   ORB_SLAM3::ORBmatcher::DescriptorDistance + TH_LOW/TH_HIGH/HISTO_LENGTH
//                                      reference include/ORBmatcher.h:43, src/ORBmatcher3.cc:637-653, src/ORBmatcher1.cc:37-39
//   cv::BFMatcher(NORM_HAMMING).knnMatch  reference src/Frame.cc:45, 1174
// Header-only; links against liborbx.so.  OpenCV is NOT required: images are passed as orbx::Image views and
// keypoints as orbx::KeyPoint, which is layout-identical to cv::KeyPoint (28 bytes).  When OpenCV headers are
// present (the reference's build), the cv:: overloads at the bottom make this class a drop-in for the
// reference's call site `(*mpORBextractorLeft)(im, cv::Mat(), mvKeys, mDescriptors, vLapping)` (src/Frame.cc:420-428).
//
// Errors: like the fork (throw std::runtime_error on a failed launch, src/ORBextractor.cc:499,862,974,1213)
// library errors are rethrown as std::runtime_error; an empty image returns -1 (src/ORBextractor.cc:1231-1232).
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <algorithm>
#include <map>
#include <vector>

#include "../../include/orbx.h"

namespace orbx {

struct Point2f { float x, y; };

// == cv::KeyPoint (pt, size, angle, response, octave, class_id)
struct KeyPoint {
    Point2f pt;
    float size;
    float angle;
    float response;
    int octave;
    int class_id;
};
static_assert(sizeof(KeyPoint) == 28 && sizeof(KeyPoint) == sizeof(orbx_keypoint), "KeyPoint must match cv::KeyPoint");

// 8-bit single-channel image view (== cv::Mat of type CV_8UC1: data / cols / rows / step)
struct Image {
    const uint8_t *data = nullptr;
    int cols = 0, rows = 0;
    ptrdiff_t step = 0;
    Image() = default;
    Image(const uint8_t *d, int c, int r, ptrdiff_t s = 0) : data(d), cols(c), rows(r), step(s ? s : c) {}
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};

// n x 32 CV_8U descriptor matrix
struct Descriptors {
    int rows = 0;
    std::vector<uint8_t> data;
    void create(int n) { rows = n; data.assign((size_t)n * 32, 0); }
    void release() { rows = 0; data.clear(); }
    const uint8_t *row(int i) const { return data.data() + (size_t)i * 32; }
    uint8_t *row(int i) { return data.data() + (size_t)i * 32; }
    bool empty() const { return rows == 0; }
};

struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
};

inline void check(int rc, const char *what)
{
    if (rc != ORBX_OK) throw std::runtime_error(std::string(what) + ": " + orbx_last_error());
}

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int device = 0, int maxWidth = 1920,
                 int maxHeight = 1080, int blurVariant = ORBX_BLUR_OPENCV_450, int trigMode = ORBX_TRIG_DOUBLE, int maxBatch = 1)
    {
        orbx_params p;
        orbx_default_params(&p);
        p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels;
        p.ini_th_fast = iniThFAST; p.min_th_fast = minThFAST;
        p.device = device; p.max_width = maxWidth; p.max_height = maxHeight; p.max_batch = maxBatch;   // (> 1: SubmitBatch / WaitBatch below)
        p.blur_variant = blurVariant; p.trig_mode = trigMode;
        check(orbx_create(&p, &h_), "ORBextractor");
    }
    ~ORBextractor() { orbx_destroy(h_); }
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image; mask is ignored (as in the reference).
    int operator()(const Image &image, const Image & /*mask*/, std::vector<KeyPoint> &keypoints, Descriptors &descriptors,
                   std::vector<int> &vLappingArea)
    {
        if (image.empty()) return -1;
        const int cap = orbx_max_keypoints(h_);
        keypoints.resize(cap);
        descriptors.create(cap);
        int n = 0;
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        int mono = orbx_extract(h_, image.data, image.cols, image.rows, image.step, lap0, lap1,
                                reinterpret_cast<orbx_keypoint *>(keypoints.data()), descriptors.data.data(), cap, &n);
        if (mono <= -101) throw std::runtime_error(std::string("ORBextractor::operator(): ") + orbx_last_error());
        keypoints.resize(n);
        if (n == 0) descriptors.release();
        else { descriptors.rows = n; descriptors.data.resize((size_t)n * 32); }
        return mono;
    }

    int GetLevels() { return orbx_get_levels(h_); }
    float GetScaleFactor() { return orbx_get_scale_factor(h_); }
    std::vector<float> GetScaleFactors() { return table(orbx_get_scale_factors); }
    std::vector<float> GetInverseScaleFactors() { return table(orbx_get_inverse_scale_factors); }
    std::vector<float> GetScaleSigmaSquares() { return table(orbx_get_scale_sigma_squares); }
    std::vector<float> GetInverseScaleSigmaSquares() { return table(orbx_get_inverse_scale_sigma_squares); }

    // mvImagePyramid[level] of the last call (interior, tight rows) — reference include/ORBextractor.h:92
    std::vector<uint8_t> ImagePyramidLevel(int level, int &cols, int &rows)
    {
        check(orbx_get_level_size(h_, level, &cols, &rows), "ImagePyramidLevel");
        std::vector<uint8_t> out((size_t)cols * rows);
        check(orbx_get_pyramid_level(h_, 0, level, 0, out.data()), "ImagePyramidLevel");
        return out;
    }
    orbx_extractor *handle() { return h_; }

    // A batch of frames with the link kept busy across calls (round 5): SubmitBatch queues the batch and returns, WaitBatch returns when the
    // OLDEST submitted batch's results are in the arrays it was given; two may be pending.  Frames and result arrays page-locked
    // (orbx_host_alloc / orbx_host_register) and untouched until their WaitBatch.  The reference keeps its two extractor objects busy side by
    // side for the same reason (src/Frame.cc:124-127).
    void SubmitBatch(const uint8_t *frames, int nframes, int cols, int rows, std::ptrdiff_t step, std::ptrdiff_t frameStep, int lap0, int lap1,
                     orbx_keypoint *kps, uint8_t *desc, int capacity, int32_t *n, int32_t *monoIndex)
    {
        check(orbx_extract_batch_submit(h_, frames, nframes, cols, rows, step, frameStep, lap0, lap1, kps, desc, capacity, n, monoIndex), "SubmitBatch");
    }
    void WaitBatch() { check(orbx_extract_batch_wait(h_), "WaitBatch"); }

private:
    std::vector<float> table(int (*fn)(const orbx_extractor *, float *))
    {
        std::vector<float> v(orbx_get_levels(h_));
        fn(h_, v.data());
        return v;
    }
    orbx_extractor *h_ = nullptr;
};

// A stream of frames dealt over the GPUs of one node from ONE host process (frames are independent, so they shard with no
// data-path collective): one extractor handle per device, block b of `block` consecutive frames on handle b mod G, every handle
// driven by a std::thread of its own inside orbx_extract_batch_sharded - the way the reference runs its left and right extractor
// side by side (src/Frame.cc:124-127) - and every frame's keypoints and descriptors at its frame index.
class ShardedExtractor {
public:
    // devices: HIP ordinals, one handle each (the same ordinal may appear twice: two handles share that GPU)
    ShardedExtractor(const std::vector<int> &devices, int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST, int maxWidth,
                     int maxHeight, int maxBatchPerDevice, int block = 1, int blurVariant = ORBX_BLUR_OPENCV_450, int trigMode = ORBX_TRIG_DOUBLE)
        : block_(block)
    {
        for (size_t i = 0; i < devices.size(); ++i) {
            orbx_params p;
            orbx_default_params(&p);
            p.nfeatures = nfeatures; p.scale_factor = scaleFactor; p.nlevels = nlevels;
            p.ini_th_fast = iniThFAST; p.min_th_fast = minThFAST;
            p.device = devices[i]; p.max_width = maxWidth; p.max_height = maxHeight; p.max_batch = maxBatchPerDevice;
            p.blur_variant = blurVariant; p.trig_mode = trigMode;
            orbx_extractor *h = nullptr;
            if (orbx_create(&p, &h) != ORBX_OK) {
                const std::string why = orbx_last_error();
                for (size_t k = 0; k < h_.size(); ++k) orbx_destroy(h_[k]);
                throw std::runtime_error("ShardedExtractor: " + why);
            }
            h_.push_back(h);
        }
        if (h_.empty()) throw std::runtime_error("ShardedExtractor: no device given");
    }
    ~ShardedExtractor() { for (size_t k = 0; k < h_.size(); ++k) orbx_destroy(h_[k]); }
    ShardedExtractor(const ShardedExtractor &) = delete;
    ShardedExtractor &operator=(const ShardedExtractor &) = delete;

    int capacity() const { return orbx_max_keypoints(h_[0]); }  // rows per frame in the output arrays
    size_t devices() const { return h_.size(); }

    // nframes images of cols x rows, `step` bytes per row, `frameStep` bytes from one image to the next -> per frame f: n[f], mono[f]
    // (= operator()'s return value), keypoints [f * capacity() ..) and descriptors [f * capacity() * 32 ..)
    void extract(const uint8_t *images, int nframes, int cols, int rows, ptrdiff_t step, ptrdiff_t frameStep, const std::vector<int> &vLappingArea,
                 std::vector<KeyPoint> &keypoints, std::vector<uint8_t> &descriptors, std::vector<int32_t> &n, std::vector<int32_t> &mono)
    {
        const int cap = capacity();
        keypoints.resize((size_t)nframes * cap);
        descriptors.resize((size_t)nframes * cap * 32);
        n.assign(nframes, 0);
        mono.assign(nframes, 0);
        const int lap0 = vLappingArea.size() > 0 ? vLappingArea[0] : 0, lap1 = vLappingArea.size() > 1 ? vLappingArea[1] : 0;
        check(orbx_extract_batch_sharded(h_.data(), (int)h_.size(), block_, images, nframes, cols, rows, step, frameStep, lap0, lap1,
                                         reinterpret_cast<orbx_keypoint *>(keypoints.data()), descriptors.data(), cap, n.data(), mono.data()),
              "ShardedExtractor::extract");
    }

private:
    std::vector<orbx_extractor *> h_;
    int block_;
};

// DBoW2::BowVector / DBoW2::FeatureVector — reference Thirdparty/DBoW2/DBoW2/BowVector.h, FeatureVector.h
typedef std::map<unsigned int, double> BowVector;
typedef std::map<unsigned int, std::vector<unsigned int> > FeatureVector;

// The part of `ORB_SLAM3::Frame` the windowed searches read (reference include/Frame.h): undistorted keypoints, descriptors and
// the static image bounds mnMinX .. mnMaxY (src/Frame.cc:41, 808-839) that define the 64 x 48 feature grid.
struct FrameView {
    const std::vector<KeyPoint> *mvKeysUn = nullptr;
    const Descriptors *mDescriptors = nullptr;
    float mnMinX = 0.f, mnMaxX = 0.f, mnMinY = 0.f, mnMaxY = 0.f;
};
class ORBmatcher {
public:
    static const int TH_LOW = 50;      // reference src/ORBmatcher1.cc:38
    static const int TH_HIGH = 100;    // :37
    static const int HISTO_LENGTH = 30;  // :39
    // ORBmatcher(float nnratio = 0.6, bool checkOri = true) — reference include/ORBmatcher.h:40.  (The reference's matcher is a
    // stateless value object; this one additionally owns the device buffers of the searches, sized at construction.)
    explicit ORBmatcher(float nnratio = 0.6f, bool checkOri = true, int maxKeypoints = 8192, int device = 0)
        : mfNNratio(nnratio), mbCheckOrientation(checkOri)
    {
        check(orbx_matcher_create(device, maxKeypoints, maxKeypoints, &m_), "ORBmatcher");
    }
    ~ORBmatcher() { orbx_matcher_destroy(m_); }
    ORBmatcher(const ORBmatcher &) = delete;
    ORBmatcher &operator=(const ORBmatcher &) = delete;

    // Computes the Hamming distance between two ORB descriptors (rows of 32 bytes)
    static int DescriptorDistance(const uint8_t *a, const uint8_t *b) { return orbx_hamming256(a, b); }

    // Matching for the Map Initialization (only used in the monocular case) — reference include/ORBmatcher.h:72,
    // src/ORBmatcher1.cc:650-765
    int SearchForInitialization(const FrameView &F1, const FrameView &F2, std::vector<Point2f> &vbPrevMatched, std::vector<int> &vnMatches12,
                                int windowSize = 10)
    {
        const int n1 = (int)F1.mvKeysUn->size(), n2 = (int)F2.mvKeysUn->size();
        if ((int)vbPrevMatched.size() != n1) throw std::runtime_error("SearchForInitialization: vbPrevMatched.size() != F1.mvKeysUn.size()");
        vnMatches12.assign((size_t)n1, -1);
        static_assert(sizeof(Point2f) == 2 * sizeof(float), "Point2f layout");
        const int n = orbx_search_for_initialization(m_, (const orbx_keypoint *)F1.mvKeysUn->data(), F1.mDescriptors->data.data(), n1,
                                                     (const orbx_keypoint *)F2.mvKeysUn->data(),
                                                     F2.mDescriptors->data.data(), n2, F2.mnMinX, F2.mnMaxX, F2.mnMinY, F2.mnMaxY,
                                                     n1 ? &vbPrevMatched[0].x : nullptr, vnMatches12.data(), windowSize, mfNNratio,
                                                     mbCheckOrientation ? 1 : 0);
        if (n < 0) check(n, "SearchForInitialization");
        return n;
    }

    // Project MapPoints tracked in last frame into the current frame and search matches (Used to track from previous frame) —
    // reference include/ORBmatcher.h:54, src/ORBmatcher3.cc:256-467 (CurrentFrame.Nleft == -1).  The caller projects
    // (x3Dc = Tcw * x3Dw, uv = mpCamera->project(x3Dc)) and passes the result per last-frame keypoint; mvpMapPointsState holds
    // -1 / 0 / 1 = CurrentFrame.mvpMapPoints[i] NULL / without / with observations; vnAssigned receives -1 untouched, j the map
    // point of last-frame keypoint j, -2 NULLed by the rotation check.
    int SearchByProjection(const FrameView &CurrentFrame, const std::vector<float> &mvuRight, const std::vector<int8_t> &mvpMapPointsState,
                           const std::vector<float> &mvScaleFactors, float mbf, const std::vector<orbx_projection> &vProjections,
                           const Descriptors &vMapPointDescriptors, std::vector<int> &vnAssigned, const float th, const bool bForward,
                           const bool bBackward)
    {
        const int nc = (int)CurrentFrame.mvKeysUn->size(), np = (int)vProjections.size();
        if ((int)mvpMapPointsState.size() != nc || (!mvuRight.empty() && (int)mvuRight.size() != nc) || vMapPointDescriptors.rows != np)
            throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnAssigned.assign((size_t)nc, -1);
        const int n = orbx_search_by_projection(m_, (const orbx_keypoint *)CurrentFrame.mvKeysUn->data(), CurrentFrame.mDescriptors->data.data(),
                                                mvuRight.empty() ? nullptr : mvuRight.data(), mvpMapPointsState.data(), nc, CurrentFrame.mnMinX,
                                                CurrentFrame.mnMaxX, CurrentFrame.mnMinY, CurrentFrame.mnMaxY, mvScaleFactors.data(),
                                                (int)mvScaleFactors.size(), mbf, vProjections.data(), vMapPointDescriptors.data.data(), np, th,
                                                bForward, bBackward, mbCheckOrientation ? 1 : 0, vnAssigned.data());
        if (n < 0) check(n, "SearchByProjection");
        return n;
    }

    // Search matches between Frame keypoints and projected MapPoints. Returns number of matches (Used to track the local map) —
    // reference include/ORBmatcher.h:48, src/ORBmatcher1.cc:45-215 (F.Nleft == -1).  One record per map point with what
    // Frame::isInFrustum left on it: u, v = mTrackProjX / Y, invz = mTrackProjXR, angle = mTrackViewCos, octave = mnTrackScaleLevel,
    // flags bit 0 = mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints), bit 1 = Observations() > 0.
    int SearchByProjection(const FrameView &F, const std::vector<float> &mvuRight, const std::vector<int8_t> &mvpMapPointsState,
                           const std::vector<float> &mvScaleFactors, const std::vector<orbx_projection> &vpMapPoints,
                           const Descriptors &vMapPointDescriptors, std::vector<int> &vnAssigned, const float th = 3)
    {
        const int nc = (int)F.mvKeysUn->size(), np = (int)vpMapPoints.size();
        if ((int)mvpMapPointsState.size() != nc || (!mvuRight.empty() && (int)mvuRight.size() != nc) || vMapPointDescriptors.rows != np)
            throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnAssigned.assign((size_t)nc, -1);
        const int n = orbx_search_by_projection_points(m_, (const orbx_keypoint *)F.mvKeysUn->data(), F.mDescriptors->data.data(),
                                                       mvuRight.empty() ? nullptr : mvuRight.data(), mvpMapPointsState.data(), nc, F.mnMinX, F.mnMaxX,
                                                       F.mnMinY, F.mnMaxY, mvScaleFactors.data(), (int)mvScaleFactors.size(), vpMapPoints.data(),
                                                       vMapPointDescriptors.data.data(), np, th, mfNNratio, vnAssigned.data());
        if (n < 0) check(n, "SearchByProjection");
        return n;
    }

    // The CurrentFrame.Nleft != -1 branch of SearchByProjection(CurrentFrame, LastFrame, th, bMono) — reference src/ORBmatcher3.cc:300-436
    // (two KB8 fisheye cameras).  CurrentFrame.mvKeysUn carries mvKeys followed by mvKeysRight, mDescriptors the vconcat of both;
    // vProjectionsRight[i].u / .v = mpCamera->project(GetRelativePoseTrl() * x3Dc); state / vnAssigned run over N = Nleft + Nright.
    int SearchByProjection(const FrameView &CurrentFrame, int Nleft, const std::vector<int8_t> &mvpMapPointsState,
                           const std::vector<float> &mvScaleFactors, const std::vector<orbx_projection> &vProjections,
                           const std::vector<orbx_projection> &vProjectionsRight, const Descriptors &vMapPointDescriptors,
                           std::vector<int> &vnAssigned, const float th, const bool bForward, const bool bBackward)
    {
        const int nc = (int)CurrentFrame.mvKeysUn->size(), np = (int)vProjections.size();
        if ((int)mvpMapPointsState.size() != nc || (int)vProjectionsRight.size() != np || vMapPointDescriptors.rows != np || Nleft < 0 || Nleft > nc)
            throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnAssigned.assign((size_t)nc, -1);
        const int n = orbx_search_by_projection_fisheye(m_, (const orbx_keypoint *)CurrentFrame.mvKeysUn->data(), CurrentFrame.mDescriptors->data.data(),
                                                        Nleft, nc - Nleft, mvpMapPointsState.data(), CurrentFrame.mnMinX, CurrentFrame.mnMaxX,
                                                        CurrentFrame.mnMinY, CurrentFrame.mnMaxY, mvScaleFactors.data(), (int)mvScaleFactors.size(),
                                                        vProjections.data(), vProjectionsRight.data(), vMapPointDescriptors.data.data(), np, th,
                                                        bForward, bBackward, mbCheckOrientation ? 1 : 0, vnAssigned.data());
        if (n < 0) check(n, "SearchByProjection");
        return n;
    }

    // The F.Nleft != -1 branch of SearchByProjection(F, vpMapPoints, th, ..) — reference src/ORBmatcher1.cc:45-215.  vpMapPoints /
    // vpMapPointsRight: the mTrack*/mTrack*R fields of every map point (flags bit 0 = mbTrackInView / mbTrackInViewR);
    // mvLeftToRightMatch / mvRightToLeftMatch as Frame::ComputeStereoFishEyeMatches left them.
    int SearchByProjection(const FrameView &F, int Nleft, const std::vector<int8_t> &mvpMapPointsState, const std::vector<float> &mvScaleFactors,
                           const std::vector<orbx_projection> &vpMapPoints, const std::vector<orbx_projection> &vpMapPointsRight,
                           const Descriptors &vMapPointDescriptors, const std::vector<int> &mvLeftToRightMatch,
                           const std::vector<int> &mvRightToLeftMatch, std::vector<int> &vnAssigned, const float th = 3)
    {
        const int nc = (int)F.mvKeysUn->size(), np = (int)vpMapPoints.size();
        if ((int)mvpMapPointsState.size() != nc || (int)vpMapPointsRight.size() != np || vMapPointDescriptors.rows != np || Nleft < 0 || Nleft > nc ||
            (int)mvLeftToRightMatch.size() != Nleft || (int)mvRightToLeftMatch.size() != nc - Nleft)
            throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnAssigned.assign((size_t)nc, -1);
        const int n = orbx_search_by_projection_points_fisheye(m_, (const orbx_keypoint *)F.mvKeysUn->data(), F.mDescriptors->data.data(), Nleft,
                                                               nc - Nleft, mvpMapPointsState.data(), F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY,
                                                               mvScaleFactors.data(), (int)mvScaleFactors.size(), vpMapPoints.data(),
                                                               vpMapPointsRight.data(), vMapPointDescriptors.data.data(), np,
                                                               mvLeftToRightMatch.data(), mvRightToLeftMatch.data(), th, mfNNratio,
                                                               vnAssigned.data());
        if (n < 0) check(n, "SearchByProjection");
        return n;
    }

    // int SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches) — reference include/ORBmatcher.h:62,
    // src/ORBmatcher1.cc:225-427 (F.Nleft == -1).  KF: mvKeysUn + mDescriptors of the keyframe, vbValidKF[i] = its map point i exists
    // and !isBad(); F: mvKeys + mDescriptors of the frame (FrameView::mvKeysUn carries them).  vnMatches[f] = keyframe feature
    // whose map point F.mvpMapPoints[f] becomes, -1 = NULL.
    int SearchByBoW(const FeatureVector &vFeatVecKF, const FrameView &KF, const std::vector<uint8_t> &vbValidKF, const FeatureVector &vFeatVecF,
                    const FrameView &F, std::vector<int> &vnMatches)
    {
        const int nk = (int)KF.mvKeysUn->size(), nf = (int)F.mvKeysUn->size();
        if ((int)vbValidKF.size() != nk) throw std::runtime_error("SearchByBoW: inconsistent array sizes");
        std::vector<int32_t> node[2], start[2], feat[2];
        const FeatureVector *fv[2] = {&vFeatVecKF, &vFeatVecF};
        for (int s = 0; s < 2; ++s) {
            start[s].push_back(0);
            for (FeatureVector::const_iterator it = fv[s]->begin(); it != fv[s]->end(); ++it) {
                node[s].push_back((int32_t)it->first);
                feat[s].insert(feat[s].end(), it->second.begin(), it->second.end());
                start[s].push_back((int32_t)feat[s].size());
            }
        }
        vnMatches.assign((size_t)nf, -1);
        const int n = orbx_search_by_bow(m_, node[0].data(), start[0].data(), feat[0].data(), (int)node[0].size(), (const orbx_keypoint *)KF.mvKeysUn->data(),
                                         KF.mDescriptors->data.data(), vbValidKF.data(), nk, node[1].data(), start[1].data(), feat[1].data(),
                                         (int)node[1].size(), (const orbx_keypoint *)F.mvKeysUn->data(), F.mDescriptors->data.data(), nf, mfNNratio,
                                         mbCheckOrientation ? 1 : 0, vnMatches.data());
        if (n < 0) check(n, "SearchByBoW");
        return n;
    }

    // int SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12) — reference include/ORBmatcher.h:63,
    // src/ORBmatcher2.cc:37-177 (NLeft == -1).  vnMatches12[idx1] = feature of pKF2 whose map point vpMatches12[idx1] becomes, -1 = NULL.
    int SearchByBoW(const FeatureVector &vFeatVec1, const FrameView &KF1, const std::vector<uint8_t> &vbValid1, const FeatureVector &vFeatVec2,
                    const FrameView &KF2, const std::vector<uint8_t> &vbValid2, std::vector<int> &vnMatches12)
    {
        const int n1 = (int)KF1.mvKeysUn->size(), n2 = (int)KF2.mvKeysUn->size();
        if ((int)vbValid1.size() != n1 || (int)vbValid2.size() != n2) throw std::runtime_error("SearchByBoW: inconsistent array sizes");
        std::vector<int32_t> node[2], start[2], feat[2];
        const FeatureVector *fv[2] = {&vFeatVec1, &vFeatVec2};
        for (int s = 0; s < 2; ++s) {
            start[s].push_back(0);
            for (FeatureVector::const_iterator it = fv[s]->begin(); it != fv[s]->end(); ++it) {
                node[s].push_back((int32_t)it->first);
                feat[s].insert(feat[s].end(), it->second.begin(), it->second.end());
                start[s].push_back((int32_t)feat[s].size());
            }
        }
        vnMatches12.assign((size_t)n1, -1);
        const int n = orbx_search_by_bow_keyframes(m_, node[0].data(), start[0].data(), feat[0].data(), (int)node[0].size(),
                                                   (const orbx_keypoint *)KF1.mvKeysUn->data(), KF1.mDescriptors->data.data(), vbValid1.data(), n1,
                                                   node[1].data(), start[1].data(), feat[1].data(), (int)node[1].size(),
                                                   (const orbx_keypoint *)KF2.mvKeysUn->data(), KF2.mDescriptors->data.data(), vbValid2.data(), n2,
                                                   mfNNratio, mbCheckOrientation ? 1 : 0, vnMatches12.data());
        if (n < 0) check(n, "SearchByBoW");
        return n;
    }

    // The part of `ORB_SLAM3::KeyFrame` SearchForTriangulation reads (reference include/KeyFrame.h): keys = mvKeysUn (NLeft == -1) or
    // mvKeys followed by mvKeysRight, mDescriptors, mvuRight (empty: none), vbHasMapPoint[i] = (GetMapPoint(i) != NULL), mFeatVec,
    // mvScaleFactors, mvLevelSigma2.
    struct KeyFrameView {
        const std::vector<KeyPoint> *keys = nullptr;
        const Descriptors *mDescriptors = nullptr;
        int NLeft = -1;
        const std::vector<float> *mvuRight = nullptr;
        const std::vector<uint8_t> *vbHasMapPoint = nullptr;
        const FeatureVector *mFeatVec = nullptr;
        const std::vector<float> *mvScaleFactors = nullptr, *mvLevelSigma2 = nullptr;
    };

    // int SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<std::pair<size_t, size_t>> &vMatchedPairs, const bool
    // bOnlyStereo, const bool bCoarse = false) — reference include/ORBmatcher.h:79-80, src/ORBmatcher2.cc:179-418.  `geometry`: what the
    // reference derives from the two poses in front of its loop (:185-214), see orbx_triangulation_geometry.
    int SearchForTriangulation(const KeyFrameView &KF1, const KeyFrameView &KF2, const orbx_triangulation_geometry &geometry,
                               std::vector<std::pair<size_t, size_t> > &vMatchedPairs, const bool bOnlyStereo, const bool bCoarse = false)
    {
        std::vector<int32_t> node[2], start[2], feat[2];
        orbx_keyframe_view v[2];
        const KeyFrameView *kf[2] = {&KF1, &KF2};
        for (int s = 0; s < 2; ++s) {
            const KeyFrameView &k = *kf[s];
            if (!k.keys || !k.mDescriptors || !k.vbHasMapPoint || !k.mFeatVec || !k.mvScaleFactors || !k.mvLevelSigma2 ||
                k.vbHasMapPoint->size() != k.keys->size() || (k.mvuRight && !k.mvuRight->empty() && k.mvuRight->size() != k.keys->size()))
                throw std::runtime_error("SearchForTriangulation: incomplete keyframe view");
            start[s].push_back(0);
            for (FeatureVector::const_iterator it = k.mFeatVec->begin(); it != k.mFeatVec->end(); ++it) {
                node[s].push_back((int32_t)it->first);
                feat[s].insert(feat[s].end(), it->second.begin(), it->second.end());
                start[s].push_back((int32_t)feat[s].size());
            }
            v[s].keys = (const orbx_keypoint *)k.keys->data(); v[s].desc = k.mDescriptors->data.data();
            v[s].n = (int32_t)k.keys->size(); v[s].n_left = k.NLeft;
            v[s].u_right = k.mvuRight && !k.mvuRight->empty() ? k.mvuRight->data() : nullptr;
            v[s].has_map_point = k.vbHasMapPoint->data();
            v[s].fv_node = node[s].data(); v[s].fv_start = start[s].data(); v[s].fv_feature = feat[s].data(); v[s].nfv = (int32_t)node[s].size();
            v[s].scale_factors = k.mvScaleFactors->data(); v[s].level_sigma2 = k.mvLevelSigma2->data();
            v[s].nlevels = (int32_t)std::min(k.mvScaleFactors->size(), k.mvLevelSigma2->size());
        }
        std::vector<int32_t> matches12((size_t)v[0].n + 1, -1), pairs(2 * (size_t)v[0].n + 2);
        const int n = orbx_search_for_triangulation(m_, &v[0], &v[1], &geometry, bOnlyStereo ? 1 : 0, bCoarse ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                    matches12.data(), pairs.data());
        if (n < 0) check(n, "SearchForTriangulation");
        vMatchedPairs.clear();
        vMatchedPairs.reserve((size_t)n);
        for (int i = 0; i < n; ++i) vMatchedPairs.push_back(std::make_pair((size_t)pairs[2 * i], (size_t)pairs[2 * i + 1]));
        return n;
    }

    // The search part of `int Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th, const bool bRight)` (reference
    // include/ORBmatcher.h:82, src/ORBmatcher2.cc:420-610; bChi2 = true) and of `Fuse(pKF, Scw, vpPoints, th, vpReplacePoint)` (:612-770;
    // bChi2 = false).  KF: the keyframe's keypoints / descriptors of the side being searched and its image bounds; vuRight = mvuRight (may
    // be empty), vInvLevelSigma2 = mvInvLevelSigma2; vPoints[i] = map point i projected by the caller (orbx_window_point: search = 0 for a
    // point that failed one of the tests in front of GetFeaturesInArea); vPointDescriptors = their MapPoint::GetDescriptor() rows.
    // vnFusedIdx[i] = keypoint with bestDist <= TH_LOW or -1; Replace / AddObservation / AddMapPoint (:581-600) remain the caller's.
    int Fuse(const FrameView &KF, const std::vector<float> &vuRight, const std::vector<float> &vInvLevelSigma2,
             const std::vector<orbx_window_point> &vPoints, const Descriptors &vPointDescriptors, bool bChi2, std::vector<int> &vnFusedIdx)
    {
        const int n = (int)KF.mvKeysUn->size(), np = (int)vPoints.size();
        if (vPointDescriptors.rows != np || (!vuRight.empty() && (int)vuRight.size() != n)) throw std::runtime_error("Fuse: inconsistent array sizes");
        vnFusedIdx.assign((size_t)np, -1);
        const int r = orbx_fuse(m_, (const orbx_keypoint *)KF.mvKeysUn->data(), KF.mDescriptors->data.data(), vuRight.empty() ? nullptr : vuRight.data(), n,
                                KF.mnMinX, KF.mnMaxX, KF.mnMinY, KF.mnMaxY, vInvLevelSigma2.data(), (int)vInvLevelSigma2.size(), vPoints.data(),
                                vPointDescriptors.data.data(), np, bChi2 ? 1 : 0, vnFusedIdx.data());
        if (r < 0) check(r, "Fuse");
        return r;
    }

    // int SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12, const Sim3f &S12, const float th) — reference
    // include/ORBmatcher.h:76, src/ORBmatcher3.cc:37-254.  vPoints1in2[i1] / vPoints2in1[i2]: the map point of feature i1 of pKF1 (i2 of pKF2)
    // transformed with S21 (S12) and projected, search = 0 where the reference `continue`s (:70-118, :158-186); vMapPointDesc1 / 2 their
    // descriptors.  vnMatches12[i1] = i2 whose map point vpMatches12[i1] becomes, -1 = unchanged.
    int SearchBySim3(const FrameView &KF1, const FrameView &KF2, const std::vector<orbx_window_point> &vPoints1in2, const Descriptors &vMapPointDesc1,
                     const std::vector<orbx_window_point> &vPoints2in1, const Descriptors &vMapPointDesc2, int nLevels, std::vector<int> &vnMatches12)
    {
        const int n1 = (int)KF1.mvKeysUn->size(), n2 = (int)KF2.mvKeysUn->size();
        if ((int)vPoints1in2.size() != n1 || (int)vPoints2in1.size() != n2 || vMapPointDesc1.rows != n1 || vMapPointDesc2.rows != n2)
            throw std::runtime_error("SearchBySim3: one projected point and one descriptor per feature of each keyframe");
        const float b1[4] = {KF1.mnMinX, KF1.mnMaxX, KF1.mnMinY, KF1.mnMaxY}, b2[4] = {KF2.mnMinX, KF2.mnMaxX, KF2.mnMinY, KF2.mnMaxY};
        vnMatches12.assign((size_t)n1, -1);
        const int r = orbx_search_by_sim3(m_, (const orbx_keypoint *)KF1.mvKeysUn->data(), KF1.mDescriptors->data.data(), n1, b1,
                                          (const orbx_keypoint *)KF2.mvKeysUn->data(), KF2.mDescriptors->data.data(), n2, b2, vPoints1in2.data(),
                                          vMapPointDesc1.data.data(), vPoints2in1.data(), vMapPointDesc2.data.data(), nLevels, vnMatches12.data());
        if (r < 0) check(r, "SearchBySim3");
        return r;
    }

    // int SearchByProjection(KeyFrame *pKF, Sim3f &Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th, float
    // ratioHamming) — reference include/ORBmatcher.h:59-60, src/ORBmatcher1.cc:429-533 (and :535-648).  vbMatched[idx] = (vpMatched[idx] !=
    // NULL) on entry; vnNewMatch[idx] = the point whose map point vpMatched[idx] becomes, -1 = unchanged.
    int SearchByProjection(const FrameView &KF, const std::vector<orbx_window_point> &vPoints, const Descriptors &vPointDescriptors,
                           const std::vector<uint8_t> &vbMatched, int nLevels, float ratioHamming, std::vector<int> &vnNewMatch)
    {
        const int n = (int)KF.mvKeysUn->size(), np = (int)vPoints.size();
        if (vPointDescriptors.rows != np || (int)vbMatched.size() != n) throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnNewMatch.assign((size_t)n, -1);
        const int r = orbx_search_by_projection_sim3(m_, (const orbx_keypoint *)KF.mvKeysUn->data(), KF.mDescriptors->data.data(), n, KF.mnMinX, KF.mnMaxX,
                                                     KF.mnMinY, KF.mnMaxY, nLevels, vPoints.data(), vPointDescriptors.data.data(), np, vbMatched.data(),
                                                     ratioHamming, vnNewMatch.data());
        if (r < 0) check(r, "SearchByProjection");
        return r;
    }

    // int SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int ORBdist) —
    // reference include/ORBmatcher.h:57, src/ORBmatcher3.cc:469-590 (Relocalization).  vPoints[i] / vPointAngles[i]: the keyframe's map point i
    // projected with the frame's pose, pKF->mvKeysUn[i].angle; vbHasMapPoint[i2] = (CurrentFrame.mvpMapPoints[i2] != NULL) on entry.
    int SearchByProjection(const FrameView &CurrentFrame, const std::vector<orbx_window_point> &vPoints, const Descriptors &vPointDescriptors,
                           const std::vector<float> &vPointAngles, const std::vector<uint8_t> &vbHasMapPoint, int nLevels, int ORBdist,
                           std::vector<int> &vnNewMatch)
    {
        const int n = (int)CurrentFrame.mvKeysUn->size(), np = (int)vPoints.size();
        if (vPointDescriptors.rows != np || (int)vPointAngles.size() != np || (int)vbHasMapPoint.size() != n)
            throw std::runtime_error("SearchByProjection: inconsistent array sizes");
        vnNewMatch.assign((size_t)n, -1);
        const int r = orbx_search_by_projection_keyframe(m_, (const orbx_keypoint *)CurrentFrame.mvKeysUn->data(), CurrentFrame.mDescriptors->data.data(), n,
                                                         CurrentFrame.mnMinX, CurrentFrame.mnMaxX, CurrentFrame.mnMinY, CurrentFrame.mnMaxY, nLevels,
                                                         vPoints.data(), vPointDescriptors.data.data(), vPointAngles.data(), np, vbHasMapPoint.data(),
                                                         ORBdist, mbCheckOrientation ? 1 : 0, vnNewMatch.data());
        if (r < 0) check(r, "SearchByProjection");
        return r;
    }

    // void MapPoint::ComputeDistinctiveDescriptors() — reference include/MapPoint.h, src/MapPoint.cc:329-403, for many map points at once:
    // vObserved = the rows the reference collects into vDescriptors, point p owning rows [vStart[p], vStart[p + 1]); vnBest[p] = the row
    // (relative) that becomes mDescriptor, -1 without observations.
    void ComputeDistinctiveDescriptors(const Descriptors &vObserved, const std::vector<int32_t> &vStart, std::vector<int> &vnBest)
    {
        if (vStart.empty()) { vnBest.clear(); return; }
        if (vStart.back() != vObserved.rows) throw std::runtime_error("ComputeDistinctiveDescriptors: vStart does not cover the descriptor rows");
        vnBest.assign(vStart.size() - 1, -1);
        check(orbx_distinctive_descriptors(m_, vObserved.data.data(), vStart.data(), (int)vStart.size() - 1, vnBest.data()), "ComputeDistinctiveDescriptors");
    }

    // Frame::UndistortKeyPoints — reference src/Frame.cc:777-811: mvKeysUn from mvKeys.  K = toK() and mK as {fx, fy, cx, cy},
    // mDistCoef as the reference's 4 / 5 / 8 / 12 floats (copies when mDistCoef[0] == 0).
    void UndistortKeyPoints(const std::vector<KeyPoint> &mvKeys, const float (&K)[4], const std::vector<float> &mDistCoef, const float (&mK)[4],
                            std::vector<KeyPoint> &mvKeysUn)
    {
        mvKeysUn.resize(mvKeys.size());
        check(orbx_undistort_keypoints(m_, (const orbx_keypoint *)mvKeys.data(), (int)mvKeys.size(), K, mDistCoef.data(), (int)mDistCoef.size(), mK,
                                       (orbx_keypoint *)mvKeysUn.data()), "UndistortKeyPoints");
    }

    // Frame::ComputeImageBounds — reference src/Frame.cc:813-845: fills F.mnMinX / mnMaxX / mnMinY / mnMaxY.
    void ComputeImageBounds(int cols, int rows, const float (&K)[4], const std::vector<float> &mDistCoef, const float (&mK)[4], FrameView &F)
    {
        float b[4];
        check(orbx_compute_image_bounds(m_, cols, rows, K, mDistCoef.data(), (int)mDistCoef.size(), mK, b), "ComputeImageBounds");
        F.mnMinX = b[0]; F.mnMaxX = b[1]; F.mnMinY = b[2]; F.mnMaxY = b[3];
    }

    // The loop head of SearchByProjection(CurrentFrame, LastFrame, ..) — reference src/ORBmatcher3.cc:281-296 — for all map points of the last
    // frame at once: x3Dc = Tcw * x3Dw (Tcw = CurrentFrame.GetPose() as unit quaternion x, y, z, w and translation), invzc, mpCamera->project.
    // vProj[i] is the record SearchByProjection(CurrentFrame, .., vProj, ..) above takes.
    void ProjectLastFramePoints(const float (&qxyzw)[4], const float (&t)[3], const orbx_camera &mpCamera, const std::vector<orbx_world_point> &vPoints,
                                std::vector<orbx_projection> &vProj)
    {
        vProj.resize(vPoints.size());
        check(orbx_project_last_frame_points(m_, qxyzw, t, &mpCamera, vPoints.data(), (int)vPoints.size(), vProj.data()), "ProjectLastFramePoints");
    }

    // Frame::isInFrustum(pMP, 0.5) for the whole local map — reference src/Frame.cc:542-600 (Tracking::SearchLocalPoints, src/Tracking4.cc:700-737)
    // - plus the far-point test of src/ORBmatcher1.cc:57-58: vMapPoints[i] = {mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos,
    // mnTrackScaleLevel, flags}, what SearchByProjection(F, .., vMapPoints, ..) above takes.  F supplies the image bounds.
    void IsInFrustum(const FrameView &F, const float (&mRcw)[9], const float (&mtcw)[3], const float (&mOw)[3], const orbx_camera &mpCamera, float mbf,
                     float viewingCosLimit, float mfLogScaleFactor, int mnScaleLevels, const std::vector<orbx_world_point> &vPoints,
                     std::vector<orbx_projection> &vMapPoints, bool bFarPoints = false, float thFarPoints = 0.f)
    {
        vMapPoints.resize(vPoints.size());
        const float bounds[4] = {F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY};
        check(orbx_project_in_frustum(m_, mRcw, mtcw, mOw, &mpCamera, bounds, mbf, viewingCosLimit, mfLogScaleFactor, mnScaleLevels, bFarPoints ? 1 : 0,
                                      thFarPoints, vPoints.data(), (int)vPoints.size(), vMapPoints.data()), "IsInFrustum");
    }

    // Frame::AssignFeaturesToGrid — reference src/Frame.cc:387-418: mGrid[ix][iy] = items[cellStart[ix * 48 + iy] .. cellStart[.. + 1])
    void AssignFeaturesToGrid(const FrameView &F, std::vector<int32_t> &cellStart, std::vector<int32_t> &items)
    {
        const int n = (int)F.mvKeysUn->size();
        cellStart.assign(ORBX_FRAME_GRID_COLS * ORBX_FRAME_GRID_ROWS + 1, 0);
        items.assign((size_t)std::max(n, 1), 0);
        const int placed = orbx_assign_features_to_grid(m_, (const orbx_keypoint *)F.mvKeysUn->data(), n, F.mnMinX, F.mnMaxX, F.mnMinY, F.mnMaxY, cellStart.data(),
                                                        items.data());
        if (placed < 0) check(placed, "AssignFeaturesToGrid");
        items.resize((size_t)placed);
    }

    // Frame::ComputeStereoFishEyeMatches — reference src/Frame.cc:1156-1196 with KannalaBrandt8::TriangulateMatches
    // (src/CameraModels/KannalaBrandt8.cpp:306-375).  camera1 / camera2 = mvParameters of mpCamera / mpCamera2 (8 floats),
    // Rlr / tlr = mRlr (row-major) / mtlr.  Fills mvLeftToRightMatch, mvRightToLeftMatch, mvDepth, mvStereo3Dpoints; returns nMatches.
    int ComputeStereoFishEyeMatches(const std::vector<KeyPoint> &mvKeys, const Descriptors &mDescriptors, int monoLeft,
                                    const std::vector<KeyPoint> &mvKeysRight, const Descriptors &mDescriptorsRight, int monoRight,
                                    const float (&camera1)[8], const float (&camera2)[8], const float (&Rlr)[9], const float (&tlr)[3],
                                    const std::vector<float> &mvLevelSigma2, std::vector<int> &mvLeftToRightMatch, std::vector<int> &mvRightToLeftMatch,
                                    std::vector<float> &mvDepth, std::vector<float> &mvStereo3Dpoints, float precision = 1e-6f)
    {
        const int nl = (int)mvKeys.size(), nr = (int)mvKeysRight.size();
        mvLeftToRightMatch.assign((size_t)std::max(nl, 1), -1); mvRightToLeftMatch.assign((size_t)std::max(nr, 1), -1);
        mvDepth.assign((size_t)std::max(nl, 1), -1.0f); mvStereo3Dpoints.assign((size_t)std::max(nl, 1) * 3, 0.f);
        const int n = orbx_compute_stereo_fisheye_matches(m_, (const orbx_keypoint *)mvKeys.data(), mDescriptors.data.data(), nl, monoLeft,
                                                          (const orbx_keypoint *)mvKeysRight.data(), mDescriptorsRight.data.data(), nr, monoRight, camera1,
                                                          camera2, precision, Rlr, tlr, mvLevelSigma2.data(), (int)mvLevelSigma2.size(),
                                                          mvLeftToRightMatch.data(), mvRightToLeftMatch.data(), mvDepth.data(), mvStereo3Dpoints.data());
        if (n < 0) check(n, "ComputeStereoFishEyeMatches");
        mvLeftToRightMatch.resize((size_t)nl); mvRightToLeftMatch.resize((size_t)nr); mvDepth.resize((size_t)nl); mvStereo3Dpoints.resize((size_t)nl * 3);
        return n;
    }

    orbx_matcher *handle() { return m_; }

private:
    float mfNNratio;
    bool mbCheckOrientation;
    orbx_matcher *m_ = nullptr;
};

// cv::BFMatcher(cv::NORM_HAMMING) for k = 2 (the only use in the reference, src/Frame.cc:1174)
class BFMatcher {
public:
    explicit BFMatcher(int maxQuery = 8192, int maxTrain = 8192, int device = 0) { check(orbx_matcher_create(device, maxQuery, maxTrain, &m_), "BFMatcher"); }
    ~BFMatcher() { orbx_matcher_destroy(m_); }
    BFMatcher(const BFMatcher &) = delete;
    BFMatcher &operator=(const BFMatcher &) = delete;

    void knnMatch(const Descriptors &query, const Descriptors &train, std::vector<std::vector<DMatch>> &matches, int k = 2)
    {
        if (k != 2) throw std::runtime_error("BFMatcher::knnMatch: only k == 2 is provided (reference src/Frame.cc:1174)");
        std::vector<int32_t> idx((size_t)query.rows * 2), dist((size_t)query.rows * 2);
        check(orbx_match_knn2(m_, query.data.data(), query.rows, train.data.data(), train.rows, idx.data(), dist.data()), "knnMatch");
        matches.assign(query.rows, {});
        for (int i = 0; i < query.rows; ++i)
            for (int j = 0; j < 2 && idx[i * 2 + j] >= 0; ++j)
                matches[i].push_back(DMatch{i, idx[i * 2 + j], 0, (float)dist[i * 2 + j]});
    }
    orbx_matcher *handle() { return m_; }

private:
    orbx_matcher *m_ = nullptr;
};

// ORB_SLAM3::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (reference include/ORBVocabulary.h) for the
// calls of System::System (loadFromTextFile) and Frame::ComputeBoW / KeyFrame::ComputeBoW (transform, src/Frame.cc:768-775).
class ORBVocabulary {
public:
    explicit ORBVocabulary(int device = 0) : device_(device) {}
    ~ORBVocabulary() { orbx_vocabulary_destroy(v_); }
    ORBVocabulary(const ORBVocabulary &) = delete;
    ORBVocabulary &operator=(const ORBVocabulary &) = delete;

    // Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423
    bool loadFromTextFile(const std::string &filename)
    {
        orbx_vocabulary_destroy(v_);
        v_ = nullptr;
        return orbx_vocabulary_load_text(device_, filename.c_str(), &v_) == ORBX_OK;
    }
    // the same tree from arrays (node 0 = root, parent[i] < i)
    void create(int k, int L, int weighting, int scoring, const std::vector<int32_t> &parent, const std::vector<uint8_t> &isLeaf,
                const std::vector<uint8_t> &descriptors, const std::vector<double> &weights)
    {
        orbx_vocabulary_destroy(v_);
        v_ = nullptr;
        if (isLeaf.size() != parent.size() || weights.size() != parent.size() || descriptors.size() != parent.size() * 32)
            throw std::runtime_error("ORBVocabulary::create: one is_leaf / descriptor / weight per node");
        check(orbx_vocabulary_create(device_, k, L, scoring, weighting, (int)parent.size(), parent.data(), isLeaf.data(), descriptors.data(),
                                     weights.data(), &v_), "ORBVocabulary::create");
    }
    bool empty() const { return size() == 0; }
    unsigned int size() const { return (unsigned int)orbx_vocabulary_size(v_); }

    // void transform(const std::vector<TDescriptor> &features, BowVector &v, FeatureVector &fv, int levelsup) const
    // — TemplatedVocabulary.h:1127-1196.  features = the rows of mDescriptors (Converter::toDescriptorVector).
    void transform(const Descriptors &features, BowVector &v, FeatureVector &fv, int levelsup) const
    {
        v.clear();
        fv.clear();
        if (!v_ || features.rows == 0) return;
        const int n = features.rows;
        std::vector<int32_t> bw((size_t)n), fn((size_t)n), fs((size_t)n + 1), ff((size_t)n);
        std::vector<double> bv((size_t)n);
        int32_t nb = 0, nf = 0;
        const int rc = orbx_vocabulary_transform(v_, features.data.data(), n, levelsup, nullptr, nullptr, bw.data(), bv.data(), &nb, fn.data(),
                                                 fs.data(), ff.data(), &nf);
        if (rc < 0) check(rc, "ORBVocabulary::transform");
        for (int i = 0; i < nb; ++i) v.insert(v.end(), BowVector::value_type((unsigned)bw[i], bv[i]));
        for (int j = 0; j < nf; ++j)
            fv.insert(fv.end(), FeatureVector::value_type((unsigned)fn[j], std::vector<unsigned int>(ff.begin() + fs[j], ff.begin() + fs[j + 1])));
    }

private:
    int device_;
    orbx_vocabulary *v_ = nullptr;
};

}  // namespace orbx

// ---- OpenCV adaptor (compiled only inside the reference's build, where OpenCV >= 4.4 exists) ----------------
namespace orbx {
// The body of the drop-in for `int ORB_SLAM3::ORBextractor::operator()(cv::InputArray, cv::InputArray, std::vector<cv::KeyPoint>&,
// cv::OutputArray, std::vector<int>&)` (reference include/ORBextractor.h:66-68, src/ORBextractor.cc:1227-1307), written against the few
// members of cv::_InputArray / cv::Mat / cv::_OutputArray it uses - empty(), getMat(), Mat::type() / data / cols / rows / step, release(),
// create(rows, cols, type) - so that it is compiled and run in this repository's tests with small stand-in types
// (tests/cpp/test_host_mirror.cpp::runCvShapedAdaptor; the image has no OpenCV), and instantiated with the real ones below.
// KeyPointT must have cv::KeyPoint's 28-byte layout; type8UC1 / type8U = CV_8UC1 / CV_8U.
template <class InputArrayT, class OutputArrayT, class KeyPointT>
inline int extractInto(ORBextractor &ex, const InputArrayT &_image, std::vector<KeyPointT> &_keypoints, OutputArrayT &_descriptors,
                       std::vector<int> &vLappingArea, int type8UC1, int type8U)
{
    static_assert(sizeof(KeyPointT) == sizeof(KeyPoint), "cv::KeyPoint layout (28 bytes: pt, size, angle, response, octave, class_id)");
    if (_image.empty()) return -1;                                                         // :1231-1232
    auto image = _image.getMat();
    if (image.type() != type8UC1) throw std::runtime_error("ORBextractor: image must be CV_8UC1");  // assert(image.type() == CV_8UC1), :1235
    std::vector<KeyPoint> kps;
    Descriptors d;
    const int mono = ex(Image(image.data, image.cols, image.rows, (ptrdiff_t)image.step), Image(), kps, d, vLappingArea);
    _keypoints.resize(kps.size());
    if (!kps.empty()) std::memcpy((void *)_keypoints.data(), kps.data(), kps.size() * sizeof(KeyPoint));
    if (kps.empty()) _descriptors.release();                                               // :1248-1249
    else {
        _descriptors.create((int)kps.size(), 32, type8U);                                  // :1251
        std::memcpy(_descriptors.getMat().data, d.data.data(), d.data.size());
    }
    return mono;
}
}  // namespace orbx

#if defined(ORBX_WITH_OPENCV) && __has_include(<opencv2/core.hpp>)
#include <opencv2/core.hpp>
namespace orbx {
inline int extract(ORBextractor &ex, cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                   cv::OutputArray _descriptors, std::vector<int> &vLappingArea)
{
    return extractInto(ex, _image, _keypoints, _descriptors, vLappingArea, CV_8UC1, CV_8U);
}
}  // namespace orbx
#endif
