// C++ parity test of the host-side mirror (wut-cuda-orb-slam3_amd/host/ORBextractor.hpp) against the CPU oracle.
// Reads like the reference's own tests (test/src/OpenCLTest.cpp: build an extractor, run one stage, check the
// result) — but asserts numeric equality instead of "the kernel started".  Needs a GPU; built and run by
// tests/test_host_mirror_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "../../include/orbx_synth.h"
#include "../../oracle/orb_oracle.h"
#include "../../wut-cuda-orb-slam3_amd/host/ORBextractor.hpp"

#define ASSERT_TRUE(c)                                                          \
    do {                                                                        \
        if (!(c)) { printf("ASSERT_TRUE failed: %s (line %d)\n", #c, __LINE__); return 1; } \
    } while (0)
#define ASSERT_EQ(a, b) ASSERT_TRUE((a) == (b))

static int runExtractAndCompare(int w, int h, int nfeatures, std::vector<int> lap)
{
    std::vector<uint8_t> img((size_t)w * h);
    orbx_synth_frame_impl(42, w, h, img.data(), w);

    orbx::ORBextractor extractor(nfeatures, 1.2f, 8, 20, 7, 0, w, h);
    std::vector<orbx::KeyPoint> keypoints;
    orbx::Descriptors descriptors;
    int mono = extractor(orbx::Image(img.data(), w, h), orbx::Image(), keypoints, descriptors, lap);

    oracle_handle *ref = oracle_create(nfeatures, 1.2f, 8, 20, 7, 0, 0);
    std::vector<oracle_keypoint> rk(oracle_max_keypoints(ref) + 64);
    std::vector<uint8_t> rd(rk.size() * 32);
    int rn = 0;
    int rmono = oracle_extract(ref, img.data(), w, h, w, lap[0], lap[1], rk.data(), rd.data(), &rn);

    ASSERT_EQ(mono, rmono);
    ASSERT_EQ((int)keypoints.size(), rn);
    ASSERT_EQ(descriptors.rows, rn);
    ASSERT_TRUE(rn >= nfeatures);
    ASSERT_TRUE(memcmp(keypoints.data(), rk.data(), (size_t)rn * 28) == 0);
    ASSERT_TRUE(memcmp(descriptors.data.data(), rd.data(), (size_t)rn * 32) == 0);

    // getters (reference include/ORBextractor.h:70-90)
    float sc[8], isc[8], s2[8], is2[8];
    int32_t fpl[8], um[16];
    oracle_get_tables(ref, sc, isc, s2, is2, fpl, um);
    ASSERT_EQ(extractor.GetLevels(), 8);
    ASSERT_TRUE(memcmp(extractor.GetScaleFactors().data(), sc, sizeof sc) == 0);
    ASSERT_TRUE(memcmp(extractor.GetInverseScaleFactors().data(), isc, sizeof isc) == 0);
    ASSERT_TRUE(memcmp(extractor.GetScaleSigmaSquares().data(), s2, sizeof s2) == 0);
    ASSERT_TRUE(memcmp(extractor.GetInverseScaleSigmaSquares().data(), is2, sizeof is2) == 0);

    // mvImagePyramid
    for (int level = 0; level < 8; ++level) {
        int lw, lh, ow, oh;
        std::vector<uint8_t> lv = extractor.ImagePyramidLevel(level, lw, lh);
        oracle_level_size(ref, level, &ow, &oh);
        ASSERT_EQ(lw, ow);
        ASSERT_EQ(lh, oh);
        std::vector<uint8_t> rl((size_t)ow * oh);
        oracle_get_level(ref, level, 0, rl.data());
        ASSERT_TRUE(lv == rl);
    }

    // DescriptorDistance + knnMatch on the descriptors just produced
    ASSERT_EQ(orbx::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(0)), 0);
    ASSERT_EQ(orbx::ORBmatcher::DescriptorDistance(descriptors.row(0), descriptors.row(1)),
              oracle_hamming256(descriptors.row(0), descriptors.row(1)));
    orbx::BFMatcher BFmatcher(8192, 8192);
    std::vector<std::vector<orbx::DMatch>> matches;
    orbx::Descriptors train = descriptors;
    BFmatcher.knnMatch(descriptors, train, matches, 2);
    std::vector<int32_t> ridx((size_t)rn * 2), rdist((size_t)rn * 2);
    oracle_knn2(descriptors.data.data(), rn, train.data.data(), rn, ridx.data(), rdist.data());
    ASSERT_EQ((int)matches.size(), rn);
    for (int i = 0; i < rn; ++i) {
        ASSERT_EQ((int)matches[i].size(), 2);
        for (int j = 0; j < 2; ++j) {
            ASSERT_EQ(matches[i][j].trainIdx, ridx[i * 2 + j]);
            ASSERT_EQ(matches[i][j].distance, (float)rdist[i * 2 + j]);
        }
        ASSERT_EQ(matches[i][0].distance, 0.0f);  // self-match
    }
    oracle_destroy(ref);
    return 0;
}

// Monocular initialization matching as src/Tracking3.cc:700-741 drives it: 5 x nFeatures extractor on two frames,
// mvbPrevMatched = the first frame's keypoint positions, ORBmatcher(0.9, true).SearchForInitialization(.., 100).
static int runSearchForInitialization()
{
    const int w = 640, h = 480;
    std::vector<uint8_t> img1((size_t)w * h), img2((size_t)w * h);
    orbx_synth_frame_impl(11, w, h, img1.data(), w);
    for (int y = 0; y < h; ++y)  // the same scene 5 px to the right, 3 px down
        for (int x = 0; x < w; ++x) img2[(size_t)y * w + x] = img1[(size_t)((y + h - 3) % h) * w + (x + w - 5) % w];
    orbx::ORBextractor mpIniORBextractor(5000, 1.2f, 8, 20, 7, 0, w, h);
    std::vector<int> lap = {0, 0};
    std::vector<orbx::KeyPoint> keys1, keys2;
    orbx::Descriptors desc1, desc2;
    mpIniORBextractor(orbx::Image(img1.data(), w, h), orbx::Image(), keys1, desc1, lap);
    mpIniORBextractor(orbx::Image(img2.data(), w, h), orbx::Image(), keys2, desc2, lap);
    orbx::FrameView mInitialFrame, mCurrentFrame;
    mInitialFrame.mvKeysUn = &keys1; mInitialFrame.mDescriptors = &desc1;
    mCurrentFrame.mvKeysUn = &keys2; mCurrentFrame.mDescriptors = &desc2;
    mInitialFrame.mnMinX = mCurrentFrame.mnMinX = 0.f; mInitialFrame.mnMaxX = mCurrentFrame.mnMaxX = (float)w;
    mInitialFrame.mnMinY = mCurrentFrame.mnMinY = 0.f; mInitialFrame.mnMaxY = mCurrentFrame.mnMaxY = (float)h;
    std::vector<orbx::Point2f> mvbPrevMatched(keys1.size());
    for (size_t i = 0; i < keys1.size(); i++) mvbPrevMatched[i] = keys1[i].pt;
    std::vector<float> refPrev(keys1.size() * 2);
    for (size_t i = 0; i < keys1.size(); i++) { refPrev[2 * i] = keys1[i].pt.x; refPrev[2 * i + 1] = keys1[i].pt.y; }

    orbx::ORBmatcher matcher(0.9f, true);
    std::vector<int> mvIniMatches;
    const int nmatches = matcher.SearchForInitialization(mInitialFrame, mCurrentFrame, mvbPrevMatched, mvIniMatches, 100);

    std::vector<int32_t> refMatches(keys1.size());
    const int refN = oracle_search_for_initialization((const oracle_keypoint *)keys1.data(), desc1.data.data(), (int)keys1.size(),
                                                      (const oracle_keypoint *)keys2.data(), desc2.data.data(), (int)keys2.size(), 0.f,
                                                      (float)w, 0.f, (float)h, refPrev.data(), refMatches.data(), 100, 0.9f, 1);
    ASSERT_EQ(nmatches, refN);
    ASSERT_TRUE(nmatches >= 100);  // enough for the initializer (src/Tracking3.cc:744)
    ASSERT_EQ(mvIniMatches.size(), refMatches.size());
    for (size_t i = 0; i < refMatches.size(); ++i) ASSERT_EQ(mvIniMatches[i], refMatches[i]);
    ASSERT_TRUE(memcmp(mvbPrevMatched.data(), refPrev.data(), refPrev.size() * sizeof(float)) == 0);

    std::vector<int32_t> cellStart, items, rcs(64 * 48 + 1), ritems(keys2.size());
    matcher.AssignFeaturesToGrid(mCurrentFrame, cellStart, items);
    const int placed = oracle_assign_grid((const oracle_keypoint *)keys2.data(), (int)keys2.size(), 0.f, (float)w, 0.f, (float)h, rcs.data(),
                                          ritems.data());
    ritems.resize((size_t)placed);
    ASSERT_TRUE(cellStart == rcs);
    ASSERT_TRUE(items == ritems);
    return 0;
}

// Tracking against the last frame and against the local map: both SearchByProjection overloads of include/ORBmatcher.h:48,54 on
// two views of one scene (src/Tracking4.cc:190-218 TrackWithMotionModel, :737 SearchLocalPoints).
static int runSearchByProjection()
{
    const int w = 640, h = 480;
    std::vector<uint8_t> img1((size_t)w * h), img2((size_t)w * h);
    orbx_synth_frame_impl(19, w, h, img1.data(), w);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) img2[(size_t)y * w + x] = img1[(size_t)((y + h - 2) % h) * w + (x + w - 4) % w];
    orbx::ORBextractor ex(1200, 1.2f, 8, 20, 7, 0, w, h);
    std::vector<int> lap = {0, 0};
    std::vector<orbx::KeyPoint> keysLast, keysCur;
    orbx::Descriptors descLast, descCur;
    ex(orbx::Image(img1.data(), w, h), orbx::Image(), keysLast, descLast, lap);
    ex(orbx::Image(img2.data(), w, h), orbx::Image(), keysCur, descCur, lap);
    orbx::FrameView CurrentFrame;
    CurrentFrame.mvKeysUn = &keysCur; CurrentFrame.mDescriptors = &descCur;
    CurrentFrame.mnMinX = 0.f; CurrentFrame.mnMaxX = (float)w; CurrentFrame.mnMinY = 0.f; CurrentFrame.mnMaxY = (float)h;
    const std::vector<float> mvScaleFactors = ex.GetScaleFactors();
    const int nc = (int)keysCur.size(), np = (int)keysLast.size();
    std::vector<orbx_projection> proj((size_t)np);
    for (int i = 0; i < np; ++i) {
        proj[i].u = keysLast[i].pt.x + 4.f + (float)(i % 5 - 2) * 0.5f;
        proj[i].v = keysLast[i].pt.y + 2.f + (float)(i % 3 - 1) * 0.5f;
        proj[i].invz = 0.25f;
        proj[i].angle = keysLast[i].angle;
        proj[i].octave = keysLast[i].octave;
        proj[i].flags = (i % 7 ? 1 : 0) | (i % 4 ? 2 : 0);
    }
    std::vector<int8_t> state((size_t)nc);
    for (int i = 0; i < nc; ++i) state[i] = i % 9 == 0 ? 1 : (i % 5 == 0 ? 0 : -1);
    std::vector<float> noRight;
    std::vector<int> assigned;
    std::vector<int32_t> ref((size_t)nc);

    orbx::ORBmatcher matcher(0.9f, true);
    const int n1 = matcher.SearchByProjection(CurrentFrame, noRight, state, mvScaleFactors, 0.f, proj, descLast, assigned, 15, false, false);
    const int r1 = oracle_search_by_projection((const oracle_keypoint *)keysCur.data(), descCur.data.data(), nullptr, state.data(), nc, 0.f, (float)w,
                                               0.f, (float)h, mvScaleFactors.data(), (int)mvScaleFactors.size(), 0.f,
                                               (const oracle_projection *)proj.data(), descLast.data.data(), np, 15, 0, 0, 1, ref.data());
    ASSERT_EQ(n1, r1);
    ASSERT_TRUE(n1 > 50);
    for (int i = 0; i < nc; ++i) ASSERT_EQ(assigned[i], ref[i]);

    // Frame::UndistortKeyPoints / ComputeImageBounds with EuRoC's camera (src/Frame.cc:777-845)
    {
        const float K[4] = {458.654f, 457.296f, 367.215f, 248.375f};
        const std::vector<float> mDistCoef = {-0.28340811f, 0.07395907f, 0.00019359f, 1.76187114e-05f};
        std::vector<orbx::KeyPoint> mvKeysUn;
        matcher.UndistortKeyPoints(keysCur, K, mDistCoef, K, mvKeysUn);
        std::vector<oracle_keypoint> refUn(keysCur.size());
        oracle_undistort_keypoints((const oracle_keypoint *)keysCur.data(), nc, K, mDistCoef.data(), 4, K, refUn.data());
        ASSERT_TRUE(memcmp(mvKeysUn.data(), refUn.data(), refUn.size() * sizeof(oracle_keypoint)) == 0);
        orbx::FrameView B;
        matcher.ComputeImageBounds(w, h, K, mDistCoef, K, B);
        float rb[4];
        oracle_compute_image_bounds(w, h, K, mDistCoef.data(), 4, K, rb);
        ASSERT_TRUE(B.mnMinX == rb[0] && B.mnMaxX == rb[1] && B.mnMinY == rb[2] && B.mnMaxY == rb[3]);
        ASSERT_TRUE(B.mnMinX < 0.f && B.mnMaxX > (float)w);
    }

    // round 4: the loop head on the device - Tracking::TrackWithMotionModel's x3Dc = Tcw * x3Dw; invzc; project (src/ORBmatcher3.cc:281-296) and
    // Frame::isInFrustum over the local map (src/Frame.cc:542-600): records bit-equal to the oracle's, then straight into the searches
    {
        const orbx_camera cam = {0, {458.654f, 457.296f, 367.215f, 248.375f, 0, 0, 0, 0}};
        std::vector<orbx_world_point> wp((size_t)np);
        for (int i = 0; i < np; ++i) {
            const float z = 2.f + (float)(i % 11);
            wp[i].pos[0] = (keysLast[i].pt.x - cam.p[2]) / cam.p[0] * z; wp[i].pos[1] = (keysLast[i].pt.y - cam.p[3]) / cam.p[1] * z; wp[i].pos[2] = i % 37 ? z : -z;
            wp[i].normal[0] = 0.f; wp[i].normal[1] = (float)(i % 3) * 0.3f; wp[i].normal[2] = 1.f - (float)(i % 3) * 0.1f;
            wp[i].min_distance = 0.5f; wp[i].max_distance = z * 1.2f * 1.2f * (i % 4 ? 1.f : 0.5f);
            wp[i].angle = keysLast[i].angle; wp[i].octave = keysLast[i].octave; wp[i].flags = (i % 7 ? 1 : 0) | (i % 4 ? 2 : 0);
        }
        const float q[4] = {0.001f, -0.002f, 0.0015f, 0.99999f}, t[3] = {0.02f, 0.01f, -0.015f};
        std::vector<orbx_projection> vProj, refProj((size_t)np);
        matcher.ProjectLastFramePoints(q, t, cam, wp, vProj);
        oracle_project_last_frame_points(q, t, cam.type, cam.p, (const oracle_world_point *)wp.data(), np, (oracle_projection *)refProj.data());
        ASSERT_TRUE(memcmp(vProj.data(), refProj.data(), (size_t)np * sizeof(orbx_projection)) == 0);
        const int n3 = matcher.SearchByProjection(CurrentFrame, noRight, state, mvScaleFactors, 0.f, vProj, descLast, assigned, 15, false, false);
        const int r3 = oracle_search_by_projection((const oracle_keypoint *)keysCur.data(), descCur.data.data(), nullptr, state.data(), nc, 0.f, (float)w,
                                                   0.f, (float)h, mvScaleFactors.data(), (int)mvScaleFactors.size(), 0.f,
                                                   (const oracle_projection *)refProj.data(), descLast.data.data(), np, 15, 0, 0, 1, ref.data());
        ASSERT_EQ(n3, r3);
        for (int i = 0; i < nc; ++i) ASSERT_EQ(assigned[i], ref[i]);
        const float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Ow[3] = {-t[0], -t[1], -t[2]};
        const float logsf = logf(1.2f);
        std::vector<orbx_projection> vMP;
        matcher.IsInFrustum(CurrentFrame, R, t, Ow, cam, 40.f, 0.5f, logsf, 8, wp, vMP, true, 9.5f);
        const float bounds[4] = {0.f, (float)w, 0.f, (float)h};
        oracle_project_in_frustum(R, t, Ow, cam.type, cam.p, bounds, 40.f, 0.5f, logsf, 8, 1, 9.5f, (const oracle_world_point *)wp.data(), np,
                                  (oracle_projection *)refProj.data());
        ASSERT_TRUE(memcmp(vMP.data(), refProj.data(), (size_t)np * sizeof(orbx_projection)) == 0);
        int inView = 0;
        for (int i = 0; i < np; ++i) inView += vMP[i].flags & 1;
        ASSERT_TRUE(inView > 50 && inView < np);
    }

    // local map: the same points as Frame::isInFrustum would leave them (mTrackViewCos, mnTrackScaleLevel, mTrackProjXR)
    for (int i = 0; i < np; ++i) { proj[i].angle = i % 2 ? 0.9995f : 0.97f; proj[i].invz = proj[i].u - 10.f; }
    orbx::ORBmatcher local(0.8f);
    const int n2 = local.SearchByProjection(CurrentFrame, noRight, state, mvScaleFactors, proj, descLast, assigned, 3);
    const int r2 = oracle_search_by_projection_points((const oracle_keypoint *)keysCur.data(), descCur.data.data(), nullptr, state.data(), nc, 0.f,
                                                      (float)w, 0.f, (float)h, mvScaleFactors.data(), (int)mvScaleFactors.size(),
                                                      (const oracle_projection *)proj.data(), descLast.data.data(), np, 3, 0.8f, ref.data());
    ASSERT_EQ(n2, r2);
    ASSERT_TRUE(n2 > 50);
    for (int i = 0; i < nc; ++i) ASSERT_EQ(assigned[i], ref[i]);
    return 0;
}

// Frame::ComputeBoW (src/Frame.cc:768-775): mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4) on a synthetic
// 6-ary, 4-level vocabulary (the reference's ORBvoc.txt is not in its tree) with the extractor's own descriptors.
static int runComputeBoW()
{
    const int k = 6, L = 4, w = 640, h = 480;
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> isLeaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    uint64_t rs = 88172645463325252ull;
    auto rnd = [&]() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return rs; };
    // breadth first: level by level, every node of a level gets k children (ids ascend with the parent's)
    size_t levelBegin = 0, levelEnd = 1;
    for (int depth = 1; depth <= L; ++depth) {
        for (size_t p = levelBegin; p < levelEnd; ++p)
            for (int c = 0; c < k; ++c) {
                parent.push_back((int32_t)p);
                isLeaf.push_back(depth == L);
                for (int b = 0; b < 32; ++b) desc.push_back((uint8_t)(rnd() >> 32));
                weight.push_back(depth == L ? (rnd() % 16 == 0 ? 0.0 : 0.25 + (double)(rnd() % 1000) / 128.0) : 0.0);
            }
        levelBegin = levelEnd;
        levelEnd = parent.size();
    }
    orbx::ORBVocabulary mpORBvocabulary;
    mpORBvocabulary.create(k, L, 0 /*TF_IDF*/, 0 /*L1_NORM*/, parent, isLeaf, desc, weight);
    ASSERT_EQ(mpORBvocabulary.size(), (unsigned)(k * k * k * k));

    std::vector<uint8_t> img((size_t)w * h);
    orbx_synth_frame_impl(23, w, h, img.data(), w);
    orbx::ORBextractor ex(1500, 1.2f, 8, 20, 7, 0, w, h);
    std::vector<int> lap = {0, 0};
    std::vector<orbx::KeyPoint> keys;
    orbx::Descriptors mDescriptors;
    ex(orbx::Image(img.data(), w, h), orbx::Image(), keys, mDescriptors, lap);
    orbx::BowVector mBowVec;
    orbx::FeatureVector mFeatVec;
    mpORBvocabulary.transform(mDescriptors, mBowVec, mFeatVec, 2);  // two levels above the words, as 4 is for L = 6

    // TrackReferenceKeyFrame: the same scene 3 px to the right as the "keyframe", ORBmatcher(0.7, true).SearchByBoW (src/ORBmatcher1.cc:225-427)
    {
        std::vector<uint8_t> img2((size_t)w * h);
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) img2[(size_t)y * w + x] = img[(size_t)y * w + (x + w - 3) % w];
        std::vector<orbx::KeyPoint> keysKF;
        orbx::Descriptors descKF;
        ex(orbx::Image(img2.data(), w, h), orbx::Image(), keysKF, descKF, lap);
        orbx::BowVector bowKF;
        orbx::FeatureVector featKF;
        mpORBvocabulary.transform(descKF, bowKF, featKF, 3);
        orbx::FeatureVector featF;
        orbx::BowVector bowF;
        mpORBvocabulary.transform(mDescriptors, bowF, featF, 3);
        std::vector<uint8_t> valid(keysKF.size());
        for (size_t i = 0; i < valid.size(); ++i) valid[i] = i % 5 != 0;
        orbx::FrameView KF, F;
        KF.mvKeysUn = &keysKF; KF.mDescriptors = &descKF; F.mvKeysUn = &keys; F.mDescriptors = &mDescriptors;
        orbx::ORBmatcher matcher(0.7f, true);
        std::vector<int> vnMatches;
        const int nm = matcher.SearchByBoW(featKF, KF, valid, featF, F, vnMatches);
        // the same through the oracle, FeatureVectors flattened from the oracle's own transform
        void *rv = oracle_vocabulary_create(k, L, 0, 0, (int)parent.size(), parent.data(), isLeaf.data(), desc.data(), weight.data());
        const int nk = descKF.rows, nf = mDescriptors.rows, nn = std::max(nk, nf);
        std::vector<int32_t> a0(nn), a1(nn), bwx(nn), kn(nn), ks(nn + 1), kfe(nn), fnn(nn), fss(nn + 1), ffe(nn), refM(nf);
        std::vector<double> bvx(nn);
        int32_t nbx = 0, nkv = 0, nfv = 0;
        oracle_vocabulary_transform(rv, descKF.data.data(), nk, 3, a0.data(), a1.data(), bwx.data(), bvx.data(), &nbx, kn.data(), ks.data(), kfe.data(), &nkv);
        oracle_vocabulary_transform(rv, mDescriptors.data.data(), nf, 3, a0.data(), a1.data(), bwx.data(), bvx.data(), &nbx, fnn.data(), fss.data(), ffe.data(), &nfv);
        oracle_vocabulary_destroy(rv);
        const int rm = oracle_search_by_bow(kn.data(), ks.data(), kfe.data(), nkv, (const oracle_keypoint *)keysKF.data(), descKF.data.data(), valid.data(),
                                            fnn.data(), fss.data(), ffe.data(), nfv, (const oracle_keypoint *)keys.data(), mDescriptors.data.data(), nf,
                                            0.7f, 1, refM.data());
        ASSERT_EQ(nm, rm);
        ASSERT_TRUE(nm > 30);
        for (int i = 0; i < nf; ++i) ASSERT_EQ(vnMatches[i], refM[i]);
    }

    void *ref = oracle_vocabulary_create(k, L, 0, 0, (int)parent.size(), parent.data(), isLeaf.data(), desc.data(), weight.data());
    const int n = mDescriptors.rows;
    std::vector<int32_t> fw(n), fnode(n), bw(n), vn(n), vs(n + 1), vf(n);
    std::vector<double> bv(n);
    int32_t nb = 0, nv = 0;
    oracle_vocabulary_transform(ref, mDescriptors.data.data(), n, 2, fw.data(), fnode.data(), bw.data(), bv.data(), &nb, vn.data(), vs.data(),
                                vf.data(), &nv);
    oracle_vocabulary_destroy(ref);
    ASSERT_EQ((int)mBowVec.size(), nb);
    ASSERT_TRUE(nb > 100);
    int i = 0;
    for (orbx::BowVector::const_iterator it = mBowVec.begin(); it != mBowVec.end(); ++it, ++i) {
        ASSERT_EQ(it->first, (unsigned)bw[i]);
        ASSERT_TRUE(memcmp(&it->second, &bv[i], sizeof(double)) == 0);
    }
    ASSERT_EQ((int)mFeatVec.size(), nv);
    ASSERT_TRUE(nv > 1 && nv <= k * k);
    int j = 0;
    for (orbx::FeatureVector::const_iterator it = mFeatVec.begin(); it != mFeatVec.end(); ++it, ++j) {
        ASSERT_EQ(it->first, (unsigned)vn[j]);
        ASSERT_EQ((int)it->second.size(), vs[j + 1] - vs[j]);
        for (size_t q = 0; q < it->second.size(); ++q) ASSERT_EQ(it->second[q], (unsigned)vf[vs[j] + q]);
    }
    return 0;
}

// Fuse (search part) and MapPoint::ComputeDistinctiveDescriptors through the C++ mirror against the oracle (round 3 widening)
static int runKeyframeSearches()
{
    const int n = 1500, np = 900, nlev = 8;
    uint64_t st = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    std::vector<orbx::KeyPoint> keys(n);
    orbx::Descriptors desc; desc.create(n);
    for (int i = 0; i < n; ++i) {
        keys[i].pt.x = 5.f + (float)(rnd() % 74000) / 100.f; keys[i].pt.y = 5.f + (float)(rnd() % 47000) / 100.f;
        keys[i].octave = (int)(rnd() % nlev); keys[i].size = 31.f; keys[i].angle = 0.f; keys[i].response = 1.f; keys[i].class_id = -1;
        for (int b = 0; b < 32; ++b) desc.data[(size_t)i * 32 + b] = (uint8_t)rnd();
    }
    std::vector<float> invSigma2(nlev), uRight(n);
    float sc = 1.f;
    for (int l = 0; l < nlev; ++l) { invSigma2[l] = 1.f / (sc * sc); sc *= 1.2f; }
    for (int i = 0; i < n; ++i) uRight[i] = (rnd() & 1) ? keys[i].pt.x - 7.f : -1.f;
    std::vector<orbx_window_point> pts(np);
    orbx::Descriptors pd; pd.create(np);
    for (int i = 0; i < np; ++i) {
        const int src = (int)(rnd() % n);
        pts[i].u = keys[src].pt.x + (float)((int)(rnd() % 400) - 200) / 100.f; pts[i].v = keys[src].pt.y + (float)((int)(rnd() % 400) - 200) / 100.f;
        pts[i].ur = pts[i].u - 7.f; pts[i].level = keys[src].octave; pts[i].radius = 6.f * (1.f + 0.2f * (float)pts[i].level); pts[i].search = (rnd() % 10) != 0;
        for (int b = 0; b < 32; ++b) pd.data[(size_t)i * 32 + b] = desc.data[(size_t)src * 32 + b] ^ (uint8_t)((rnd() % 5 == 0) ? (1u << (rnd() % 8)) : 0);
    }
    orbx::FrameView KF;
    KF.mvKeysUn = &keys; KF.mDescriptors = &desc; KF.mnMinX = 0.f; KF.mnMaxX = 752.f; KF.mnMinY = 0.f; KF.mnMaxY = 480.f;
    orbx::ORBmatcher matcher(0.6f, true, 4096, 0);
    std::vector<int> fused;
    const int nFused = matcher.Fuse(KF, uRight, invSigma2, pts, pd, true, fused);
    std::vector<int32_t> bi(np), bd(np);
    oracle_search_keyframe_windows((const oracle_keypoint *)keys.data(), desc.data.data(), uRight.data(), n, 0.f, 752.f, 0.f, 480.f, invSigma2.data(),
                                   (const oracle_window_point *)pts.data(), pd.data.data(), np, 1, bi.data(), bd.data());
    int expFused = 0;
    for (int i = 0; i < np; ++i) {
        const int e = bd[i] <= 50 ? bi[i] : -1;
        ASSERT_EQ(fused[i], e);
        expFused += e >= 0;
    }
    ASSERT_EQ(nFused, expFused);
    ASSERT_TRUE(nFused > np / 4);
    // distinctive descriptors: 200 map points with 1 .. 9 observations each
    std::vector<int32_t> start(1, 0);
    for (int p = 0; p < 200; ++p) start.push_back(start.back() + 1 + (int)(rnd() % 9));
    orbx::Descriptors obs; obs.create(start.back());
    for (size_t b = 0; b < obs.data.size(); ++b) obs.data[b] = (uint8_t)(rnd() % 4 ? 0x5A : rnd());
    std::vector<int> best;
    matcher.ComputeDistinctiveDescriptors(obs, start, best);
    std::vector<int32_t> rb(200);
    oracle_distinctive_descriptors(obs.data.data(), start.data(), 200, rb.data());
    for (int p = 0; p < 200; ++p) ASSERT_EQ(best[p], rb[p]);
    return 0;
}

// The operator() adaptor (orbx::extractInto) instantiated with stand-ins for the OpenCV types it touches: what a build with
// -DORBX_WITH_OPENCV instantiates with cv::InputArray / cv::Mat / cv::OutputArray / cv::KeyPoint.  The stand-ins carry exactly the members
// the adaptor uses, with cv::Mat's semantics (a header that shares its buffer, step in bytes, create() re-allocates).
namespace cvshape {
struct Mat {
    uint8_t *data = nullptr;
    int cols = 0, rows = 0, flags = 0;
    size_t step = 0;
    std::shared_ptr<std::vector<uint8_t>> buf;
    int type() const { return flags; }
};
struct InputArray {
    const Mat *m;
    bool empty() const { return !m || !m->data || m->rows == 0 || m->cols == 0; }
    Mat getMat() const { return *m; }
};
struct OutputArray {
    Mat *m;
    void release() { *m = Mat(); }
    void create(int rows, int cols, int type)
    {
        m->buf = std::make_shared<std::vector<uint8_t>>((size_t)rows * cols);
        m->data = m->buf->data(); m->rows = rows; m->cols = cols; m->step = (size_t)cols; m->flags = type;
    }
    Mat getMat() const { return *m; }
};
struct KeyPoint { float x, y, size, angle, response; int octave, class_id; };  // cv::KeyPoint: Point2f pt; float size, angle, response; int octave, class_id
}  // namespace cvshape

static int runCvShapedAdaptor()
{
    const int w = 640, h = 480, pitch = 704, T8UC1 = 0 /* CV_8UC1 */, T8U = 0 /* CV_8U */;
    std::vector<uint8_t> img((size_t)pitch * h, 0x5A);  // a view with row padding, as a cv::Mat ROI has
    orbx_synth_frame_impl(77, w, h, img.data(), pitch);
    cvshape::Mat image;
    image.data = img.data(); image.cols = w; image.rows = h; image.step = pitch; image.flags = T8UC1;
    orbx::ORBextractor extractor(1000, 1.2f, 8, 20, 7, 0, w, h);
    std::vector<cvshape::KeyPoint> keypoints(3);
    cvshape::Mat descriptors;
    cvshape::OutputArray out{&descriptors};
    std::vector<int> lap = {0, 1000};
    const int mono = orbx::extractInto(extractor, cvshape::InputArray{&image}, keypoints, out, lap, T8UC1, T8U);
    oracle_handle *ref = oracle_create(1000, 1.2f, 8, 20, 7, 0, 0);
    std::vector<oracle_keypoint> rk(oracle_max_keypoints(ref) + 64);
    std::vector<uint8_t> rd(rk.size() * 32);
    int rn = 0;
    const int rmono = oracle_extract(ref, img.data(), w, h, pitch, lap[0], lap[1], rk.data(), rd.data(), &rn);
    oracle_destroy(ref);
    ASSERT_EQ(mono, rmono);
    ASSERT_EQ((int)keypoints.size(), rn);
    ASSERT_EQ(descriptors.rows, rn);
    ASSERT_EQ(descriptors.cols, 32);
    ASSERT_TRUE(memcmp(keypoints.data(), rk.data(), (size_t)rn * 28) == 0);
    ASSERT_TRUE(memcmp(descriptors.data, rd.data(), (size_t)rn * 32) == 0);
    // empty image -> -1, keypoints / descriptors untouched; wrong type -> exception (the reference asserts)
    cvshape::Mat none;
    ASSERT_EQ(orbx::extractInto(extractor, cvshape::InputArray{&none}, keypoints, out, lap, T8UC1, T8U), -1);
    ASSERT_EQ((int)keypoints.size(), rn);
    image.flags = 16;
    bool thrown = false;
    try { orbx::extractInto(extractor, cvshape::InputArray{&image}, keypoints, out, lap, T8UC1, T8U); } catch (const std::runtime_error &) { thrown = true; }
    ASSERT_TRUE(thrown);
    return 0;
}

// frames dealt over several handles from one process; one handle per GPU where the box has several, two on GPU 0 otherwise
static int runShardedBatch()
{
    const int w = 640, h = 480, B = 21;
    std::vector<uint8_t> imgs((size_t)B * w * h);
    for (int f = 0; f < B; ++f) orbx_synth_frame_impl(9000 + f, w, h, imgs.data() + (size_t)f * w * h, w);
    const int ndev = orbx_device_count();
    std::vector<int> devices;
    for (int d = 0; d < (ndev >= 2 ? std::min(ndev, 4) : 2); ++d) devices.push_back(ndev >= 2 ? d : 0);
    printf("sharded batch: %d frames over %zu handles on %d device(s)\n", B, devices.size(), ndev);
    orbx::ShardedExtractor sharded(devices, 1000, 1.2f, 8, 20, 7, w, h, /*maxBatchPerDevice=*/8, /*block=*/2);
    std::vector<orbx::KeyPoint> kps;
    std::vector<uint8_t> desc;
    std::vector<int32_t> n, mono;
    std::vector<int> lap = {100, 400};
    sharded.extract(imgs.data(), B, w, h, w, (ptrdiff_t)w * h, lap, kps, desc, n, mono);
    const int cap = sharded.capacity();
    oracle_handle *ref = oracle_create(1000, 1.2f, 8, 20, 7, 0, 0);
    std::vector<oracle_keypoint> rk(oracle_max_keypoints(ref) + 64);
    std::vector<uint8_t> rd(rk.size() * 32);
    for (int f = 0; f < B; f += 4) {  // every 4th frame against the oracle: blocks of both / all handles, first and later rounds
        int rn = 0;
        const int rmono = oracle_extract(ref, imgs.data() + (size_t)f * w * h, w, h, w, lap[0], lap[1], rk.data(), rd.data(), &rn);
        ASSERT_EQ(mono[f], rmono);
        ASSERT_EQ(n[f], rn);
        ASSERT_TRUE(memcmp(kps.data() + (size_t)f * cap, rk.data(), (size_t)rn * 28) == 0);
        ASSERT_TRUE(memcmp(desc.data() + (size_t)f * cap * 32, rd.data(), (size_t)rn * 32) == 0);
    }
    oracle_destroy(ref);
    return 0;
}

// two batches in flight on one handle through the C++ mirror (round 5: orbx_extract_batch_submit / _wait): page-locked frames and result
// arrays, every frame against the oracle
static int runSubmitWait()
{
    const int w = 640, h = 480, B = 40, NB = 3;
    orbx::ORBextractor extractor(1000, 1.2f, 8, 20, 7, 0, w, h, ORBX_BLUR_OPENCV_450, ORBX_TRIG_DOUBLE, B);
    const int cap = orbx_max_keypoints(extractor.handle());
    uint8_t *frames[NB]; orbx_keypoint *kps[2]; uint8_t *desc[2]; int32_t *n[2], *mono[2];
    for (int b = 0; b < NB; ++b) {
        frames[b] = (uint8_t *)orbx_host_alloc((size_t)B * w * h);
        ASSERT_TRUE(frames[b] != nullptr);
        for (int f = 0; f < B; ++f) orbx_synth_frame_impl(7000 + 100 * b + f, w, h, frames[b] + (size_t)f * w * h, w);
    }
    for (int s = 0; s < 2; ++s) {
        kps[s] = (orbx_keypoint *)orbx_host_alloc((size_t)B * cap * sizeof(orbx_keypoint)); desc[s] = (uint8_t *)orbx_host_alloc((size_t)B * cap * 32);
        n[s] = (int32_t *)orbx_host_alloc(B * sizeof(int32_t)); mono[s] = (int32_t *)orbx_host_alloc(B * sizeof(int32_t));
    }
    oracle_handle *ref = oracle_create(1000, 1.2f, 8, 20, 7, 0, 0);
    std::vector<oracle_keypoint> rk(oracle_max_keypoints(ref) + 64);
    std::vector<uint8_t> rd(rk.size() * 32);
    auto check = [&](int b, int s) {
        for (int f = 0; f < B; f += 7) {
            int rn = 0;
            const int rmono = oracle_extract(ref, frames[b] + (size_t)f * w * h, w, h, w, 0, 1000, rk.data(), rd.data(), &rn);
            if (mono[s][f] != rmono || n[s][f] != rn || memcmp(kps[s] + (size_t)f * cap, rk.data(), (size_t)rn * 28) != 0 ||
                memcmp(desc[s] + (size_t)f * cap * 32, rd.data(), (size_t)rn * 32) != 0) return 1;
        }
        return 0;
    };
    extractor.SubmitBatch(frames[0], B, w, h, w, (std::ptrdiff_t)w * h, 0, 1000, kps[0], desc[0], cap, n[0], mono[0]);
    for (int k = 1; k < 6; ++k) {
        extractor.SubmitBatch(frames[k % NB], B, w, h, w, (std::ptrdiff_t)w * h, 0, 1000, kps[k & 1], desc[k & 1], cap, n[k & 1], mono[k & 1]);
        extractor.WaitBatch();                       // batch k - 1
        ASSERT_EQ(check((k - 1) % NB, (k - 1) & 1), 0);
    }
    extractor.WaitBatch();
    ASSERT_EQ(check(5 % NB, 5 & 1), 0);
    bool thrown = false;
    try { extractor.WaitBatch(); } catch (const std::runtime_error &) { thrown = true; }   // nothing pending
    ASSERT_TRUE(thrown);
    oracle_destroy(ref);
    for (int b = 0; b < NB; ++b) orbx_host_free(frames[b]);
    for (int s = 0; s < 2; ++s) { orbx_host_free(kps[s]); orbx_host_free(desc[s]); orbx_host_free(n[s]); orbx_host_free(mono[s]); }
    printf("submit / wait: 6 batches of %d frames, two in flight\n", B);
    return 0;
}

int main()
{
    // empty image -> -1 (reference src/ORBextractor.cc:1231-1232)
    {
        orbx::ORBextractor extractor(1000, 1.2f, 8, 20, 7, 0, 640, 480);
        std::vector<orbx::KeyPoint> kps;
        orbx::Descriptors d;
        std::vector<int> lap = {0, 0};
        ASSERT_EQ(extractor(orbx::Image(), orbx::Image(), kps, d, lap), -1);
        // too small for the FAST cell grid -> std::runtime_error, like the fork's failed launches
        std::vector<uint8_t> tiny(64 * 64, 0);
        bool thrown = false;
        try { extractor(orbx::Image(tiny.data(), 64, 64), orbx::Image(), kps, d, lap); } catch (const std::runtime_error &) { thrown = true; }
        ASSERT_TRUE(thrown);
    }
    if (runExtractAndCompare(640, 480, 1000, {0, 1000})) return 1;   // mono call of Frame.cc:313
    if (runExtractAndCompare(752, 480, 1200, {0, 0})) return 1;      // EuRoC geometry
    if (runExtractAndCompare(512, 512, 1500, {100, 300})) return 1;  // TUM-VI geometry with a lapping band
    if (runSearchForInitialization()) return 1;
    if (runSearchByProjection()) return 1;                           // SURVEY 8(f) rank 2, tracking searches
    if (runComputeBoW()) return 1;
    if (runShardedBatch()) return 1;                                 // SURVEY 8(e), one process
    if (runKeyframeSearches()) return 1;                             // Fuse + ComputeDistinctiveDescriptors through the C++ mirror
    if (runCvShapedAdaptor()) return 1;                              // the operator() adaptor of the OpenCV build, on stand-in types
    if (runSubmitWait()) return 1;                                   // two host batches in flight on one handle
    printf("OK host mirror\n");
    return 0;
}
