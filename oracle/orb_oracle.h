/*
 * orb_oracle.h — C ABI of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product (liborbx.so) never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for every image-dependent stage — the reference's own tests hold no
 * numeric golden vectors for this path and the reference cannot be built here
 * (OpenCV >= 4.4 absent).  What IS pinned against the reference text: umax (Angle.cl:3), the
 * 256-pair sampling pattern (SHA-256), per-level quotas / level sizes / scale tables (App. C),
 * DescriptorDistance identities.  OpenCV internals (resize, FAST, GaussianBlur, fastAtan2,
 * BFMatcher) are restated from OpenCV 4.5.0 semantics as documented in SURVEY.md App. A.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_handle oracle_handle;

/* 28-byte record, identical field order to cv::KeyPoint / key_point_t
 * (reference include/OpenCL/Kernel/key_point.hpp). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} oracle_keypoint;

/* blur_variant: 0 = OpenCV 4.5.0 taps {18,34,49,55,49,34,18}; 1 = OpenCV >= 4.5.1 taps
 * {18,34,48,56,48,34,18}.  trig_mode: 0 = (float)cos((double)angle) in computeOrbDescriptor; 1 = libm cosf / sinf (the float overloads). */
oracle_handle *oracle_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th,
                             int blur_variant, int trig_mode);
void oracle_destroy(oracle_handle *);

/* constructor tables (reference src/ORBextractor.cc:410-468) */
int oracle_get_tables(const oracle_handle *, float *scale, float *inv_scale, float *sigma2, float *inv_sigma2,
                      int32_t *features_per_level, int32_t *umax16);
int oracle_max_keypoints(const oracle_handle *); /* sum(N_l + 3) */
/* sum over levels of max(N_l + 3, 4 * nIni_l) for a w x h image: the bound that also holds for very wide images (nIni = round(W / H)
 * root nodes are split into up to 4 before the quota is looked at); kps / desc of oracle_extract must hold this many entries. */
int oracle_max_keypoints_for(const oracle_handle *, int w, int h);

/* ORBextractor::operator() (reference src/ORBextractor.cc:1227-1307).  Returns monoIndex, or -1 on an
 * empty image.  kps/desc must hold oracle_max_keypoints() entries. */
int oracle_extract(oracle_handle *, const uint8_t *img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                   oracle_keypoint *kps, uint8_t *desc, int *n_out);

/* stage accessors for the last oracle_extract() on this handle */
int oracle_level_size(const oracle_handle *, int level, int *w, int *h);
int oracle_get_level(const oracle_handle *, int level, int blurred, uint8_t *out /* w*h, tight */);
/* FAST candidates (A5) of a level: packed x | y<<12 | score<<24, coordinates relative to the 16-px border */
int oracle_get_candidates(const oracle_handle *, int level, uint32_t *out, int cap);
/* per-level keypoints after DistributeOctTree + border + orientation (level coordinates) */
int oracle_get_level_keypoints(const oracle_handle *, int level, oracle_keypoint *out, int cap);

/* Batch runner used for the CPU baseline: frames are contiguous [n][h][stride]; `threads` workers each
 * with a private extractor; fills counts[n] (and kps/desc when non-NULL, cap entries per frame). */
int oracle_extract_batch(oracle_handle *, const uint8_t *imgs, int n, int w, int h, ptrdiff_t stride,
                         int lap0, int lap1, int threads, int32_t *counts, int32_t *mono,
                         oracle_keypoint *kps, uint8_t *desc, int cap);

/* stand-alone stages, for stage-level parity tests */
int oracle_resize_linear(const uint8_t *src, int sw, int sh, ptrdiff_t sstride, uint8_t *dst, int dw, int dh,
                         ptrdiff_t dstride);
int oracle_gaussian_blur7(const uint8_t *src, int w, int h, ptrdiff_t sstride, uint8_t *dst, ptrdiff_t dstride,
                          int blur_variant);
/* cv::FAST(roi, th, nms=true) on a ROI; out = packed x | y<<12 | score<<24 in emission order */
int oracle_fast_roi(const uint8_t *roi, int cols, int rows, ptrdiff_t stride, int threshold, uint32_t *out,
                    int cap);
float oracle_fast_atan2(float y, float x);
/* DistributeOctTree on packed candidates (x | y<<12 | score<<24), region W x H (= maxX-minX, maxY-minY),
 * quota N; out_idx = indices into cand[] in result order.  Returns count. */
int oracle_distribute(const uint32_t *cand, int ncand, int W, int H, int N, int32_t *out_idx, int cap);
/* std::sort(pairs, compareNodes) permutation probe: keys (count, ulx); writes the sorted order of
 * original indices.  Used to pin the device introsort emulation. */
int oracle_sort_pairs(const int32_t *count, const int32_t *ulx, int n, int32_t *order);

/* matcher leafs */
int oracle_hamming256(const void *a, const void *b); /* ORBmatcher::DescriptorDistance, ORBmatcher3.cc:637-653 */
/* cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) (reference src/Frame.cc:1174); idx = -1 / dist = INT_MAX when absent */
int oracle_knn2(const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx /*nq*2*/, int32_t *dist /*nq*2*/);
/* Hamming stage of Frame::ComputeStereoMatches (reference src/Frame.cc:841-925).  best_idx[i] = right index
 * or -1, best_dist[i] = distance (TH_HIGH when none). */
int oracle_stereo_rows(const oracle_keypoint *kl, const uint8_t *dl, int nl, const oracle_keypoint *kr,
                       const uint8_t *dr, int nr, int n_rows, const float *scale_factors, float min_d, float max_d,
                       int32_t *best_idx, int32_t *best_dist);

/* Frame::ComputeStereoMatches, complete (reference src/Frame.cc:841-1011): row-band Hamming stage, 11x11 SAD
 * sliding-window refinement on the pyramids of the two extractors' last oracle_extract(), parabola fit, disparity
 * check and the median-based outlier rejection.  u_right[nl] / depth[nl] = -1 where no match (mvuRight / mvDepth). */
int oracle_stereo_matches(const oracle_handle *left, const oracle_handle *right, const oracle_keypoint *kl, const uint8_t *dl,
                          int nl, const oracle_keypoint *kr, const uint8_t *dr, int nr, float mb, float mbf, float *u_right,
                          float *depth);

/* SURVEY §8(f) rank 2.  Frame::AssignFeaturesToGrid + PosInGrid (reference src/Frame.cc:387-418, 755-766): the 64 x 48 grid over
 * [min_x, max_x) x [min_y, max_y) (Frame::mnMinX ..) as CSR: cell_start[64 * 48 + 1] (cell = ix * 48 + iy), items in push order. */
int oracle_assign_grid(const oracle_keypoint *k, int n, float min_x, float max_x, float min_y, float max_y, int32_t *cell_start,
                       int32_t *items);
/* Frame::GetFeaturesInArea (src/Frame.cc:687-753) on that grid; returns the count, indices in the reference's order. */
int oracle_features_in_area(const oracle_keypoint *k, int n, float min_x, float max_x, float min_y, float max_y, float x, float y, float r,
                            int min_level, int max_level, int32_t *out, int cap);
/* ORBmatcher::SearchForInitialization (src/ORBmatcher1.cc:650-765) incl. ComputeThreeMaxima (src/ORBmatcher3.cc:592-633).
 * prev_matched = vbPrevMatched [n1][2] (in / out), matches12 [n1] = vnMatches12; returns nmatches. */
int oracle_search_for_initialization(const oracle_keypoint *k1, const uint8_t *d1, int n1, const oracle_keypoint *k2, const uint8_t *d2,
                                     int n2, float min_x, float max_x, float min_y, float max_y, float *prev_matched, int32_t *matches12,
                                     int window_size, float nnratio, int check_orientation);

/* One projected map point of the last frame, as ORBmatcher::SearchByProjection(Frame&, const Frame&, ..) consumes it after
 * `x3Dc = Tcw * x3Dw; uv = mpCamera->project(x3Dc)` (reference src/ORBmatcher3.cc:283-296): same layout as orbx_projection. */
typedef struct {
    float u, v, invz;  /* projection and 1 / depth in the current camera */
    float angle;       /* angle of the last frame's keypoint (kpLF.angle) */
    int32_t octave;    /* nLastOctave */
    int32_t flags;     /* bit 0: map point present and not an outlier; bit 1: its Observations() > 0 */
} oracle_projection;
/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th, bMono) (src/ORBmatcher3.cc:256-467), the
 * CurrentFrame.Nleft == -1 branch; see orb_oracle.cpp for the argument contract.  Returns nmatches. */
int oracle_search_by_projection(const oracle_keypoint *kc, const uint8_t *dc, const float *u_right, const int8_t *cur_state, int nc,
                                float min_x, float max_x, float min_y, float max_y, const float *scale_factors, int nlevels, float mbf,
                                const oracle_projection *proj, const uint8_t *dproj, int np, float th, int forward, int backward,
                                int check_orientation, int32_t *assign);

/* ---- projection head of the two tracking searches (the PROJECT half of "project-then-match") ---------------------------------------
 * What the loops read of one MapPoint: same layout as orbx_world_point. */
typedef struct {
    float pos[3], normal[3], min_distance, max_distance, angle;
    int32_t octave, flags;
} oracle_world_point;
/* x3Dc = Tcw * x3Dw; invzc = 1.0 / x3Dc(2); uv = mpCamera->project(x3Dc) - reference src/ORBmatcher3.cc:283-293 with Tcw a Sophus::SE3f
 * (Thirdparty/Sophus/sophus/so3.hpp:358-367, se3.hpp) and the camera of src/CameraModels/{Pinhole.cpp:43-49, KannalaBrandt8.cpp:67-84}
 * (cam_type 0 / 1, cam = mvParameters).  out[i] = {uv(0), uv(1), invzc, angle, octave, flags}. */
void oracle_project_last_frame_points(const float *q_xyzw, const float *t, int cam_type, const float *cam, const oracle_world_point *pts, int np,
                                      oracle_projection *out);
/* MapPoint::PredictScale (src/MapPoint.cc:514-546; the Frame and the KeyFrame overload are the same arithmetic): ratio = mfMaxDistance /
 * currentDist (float), nScale = ceil(log(ratio) / mfLogScaleFactor) in FLOAT - std::log(float) = the real logf, float division, std::ceil(float)
 * - then the int conversion and the clamps to [0, nlevels - 1]. */
int oracle_predict_scale(float ratio, float log_scale_factor, int nlevels);
/* Frame::isInFrustum (src/Frame.cc:542-600, Nleft == -1) + MapPoint::PredictScale (src/MapPoint.cc:531-546) + the far-point test of
 * src/ORBmatcher1.cc:57-58; out[i] = {mTrackProjX, mTrackProjY, mTrackProjXR, mTrackViewCos, mnTrackScaleLevel, flags}. */
void oracle_project_in_frustum(const float *Rcw, const float *tcw, const float *Ow, int cam_type, const float *cam, const float *bounds, float mbf,
                               float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points, float th_far_points,
                               const oracle_world_point *pts, int np, oracle_projection *out);

/* Frame::isInFrustum for a fisheye rig (src/Frame.cc:604-614 + isInFrustumChecks :1198-1272): both cameras' records from the left pose and the
 * rig's relative pose (Rrl / trl = mTrl.rotationMatrix() / .translation(), tlr = mTlr.translation(), Rwc = mRwc); see orb_oracle.cpp. */
void oracle_project_in_frustum_fisheye(const float *Rcw, const float *tcw, const float *Ow, const float *Rrl, const float *trl, const float *tlr,
                                       const float *Rwc, const float *cam1, const float *cam2, const float *bounds, float viewing_cos_limit,
                                       float log_scale_factor, int nlevels, int far_points, float th_far_points, const oracle_world_point *pts,
                                       int np, oracle_projection *out_left, oracle_projection *out_right);
/* The projection head of ORBmatcher::Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame) (src/ORBmatcher2.cc:446-520,
 * src/ORBmatcher1.cc:451-494, src/ORBmatcher3.cc:486-519) -> the oracle_window_point records (declared below) of the window searches; tests: bit 0
 * positive depth, bit 1 viewing angle, bit 2 Frame-style bounds. */
void oracle_project_window_points(const float *q_xyzw, const float *t, const float *Ow, int cam_type, const float *cam, const float *bounds, float bf,
                                  float th, const float *scale_factors, float log_scale_factor, int nlevels, int tests,
                                  const oracle_world_point *pts, int np, void *out /* oracle_window_point[np] */);

/* ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*>&, th, ..) (src/ORBmatcher1.cc:45-215), F.Nleft == -1, points
 * with mbTrackInView: a record holds u = mTrackProjX, v = mTrackProjY, invz = mTrackProjXR, angle = mTrackViewCos,
 * octave = mnTrackScaleLevel, flags as above.  Returns nmatches. */
int oracle_search_by_projection_points(const oracle_keypoint *kc, const uint8_t *dc, const float *u_right, const int8_t *cur_state, int nc,
                                       float min_x, float max_x, float min_y, float max_y, const float *scale_factors, int nlevels,
                                       const oracle_projection *mps, const uint8_t *dmp, int np, float th, float nnratio, int32_t *assign);

/* The CurrentFrame.Nleft != -1 (KB8 fisheye rig) branches of the two functions above (src/ORBmatcher3.cc:300-436,
 * src/ORBmatcher1.cc:45-215): kc / dc = mvKeys then mvKeysRight / mDescriptors, cur_state / assign over n_left + n_right. */
int oracle_search_by_projection_fisheye(const oracle_keypoint *kc, const uint8_t *dc, int n_left, int n_right, const int8_t *cur_state,
                                        float min_x, float max_x, float min_y, float max_y, const float *scale_factors, int nlevels,
                                        const oracle_projection *proj_l, const oracle_projection *proj_r, const uint8_t *dproj, int np,
                                        float th, int forward, int backward, int check_orientation, int32_t *assign);
int oracle_search_by_projection_points_fisheye(const oracle_keypoint *kc, const uint8_t *dc, int n_left, int n_right, const int8_t *cur_state,
                                               float min_x, float max_x, float min_y, float max_y, const float *scale_factors, int nlevels,
                                               const oracle_projection *mp_l, const oracle_projection *mp_r, const uint8_t *dmp, int np,
                                               const int32_t *left_to_right, const int32_t *right_to_left, float th, float nnratio,
                                               int32_t *assign);

/* Frame::UndistortKeyPoints / Frame::ComputeImageBounds (src/Frame.cc:777-845) over cv::undistortPoints (OpenCV 4.x, restated;
 * parity unpinned).  k4 / knew4 = fx, fy, cx, cy of toK() / mK; dist = mDistCoef (4, 5, 8 or 12 floats). */
int oracle_undistort_keypoints(const oracle_keypoint *kps, int n, const float *k4, const float *dist, int ndist, const float *knew4,
                               oracle_keypoint *out);
int oracle_compute_image_bounds(int cols, int rows, const float *k4, const float *dist, int ndist, const float *knew4, float *bounds);

/* KannalaBrandt8::TriangulateMatches (reference src/CameraModels/KannalaBrandt8.cpp:306-375) and Frame::ComputeStereoFishEyeMatches
 * (src/Frame.cc:1156-1196).  cam1 / cam2 = mvParameters (fx fy cx cy k0 k1 k2 k3), precision = KannalaBrandt8::precision (1e-6),
 * R12 (row-major 3 x 3) / t12 = Frame::mRlr / mtlr.  Floating point PARITY UNPINNED (Eigen + libm): see the comment in orb_oracle.cpp. */
float oracle_kb8_triangulate_matches(const float *cam1, const float *cam2, float precision, float x1, float y1, float x2, float y2,
                                     const float *R12, const float *t12, float sigmaLevel, float unc, float *p3D);
int oracle_fisheye_stereo_matches(const oracle_keypoint *kl, const uint8_t *dl, int nl, int mono_left, const oracle_keypoint *kr,
                                  const uint8_t *dr, int nr, int mono_right, const float *cam1, const float *cam2, float precision,
                                  const float *R12, const float *t12, const float *level_sigma2, int32_t *left_to_right,
                                  int32_t *right_to_left, float *depth, float *p3d);

/* DBoW2 vocabulary tree as Frame::ComputeBoW uses it (src/Frame.cc:768-775 -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1261).
 * Nodes as loadFromTextFile builds them: node 0 = root, parent[i] < i, word ids in leaf order; scoring / weighting = the enums of
 * BowVector.h:39-56 (L1_NORM 0 .. DOT_PRODUCT 5; TF_IDF 0, TF 1, IDF 2, BINARY 3). */
void *oracle_vocabulary_create(int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                               const uint8_t *desc, const double *weight);
void oracle_vocabulary_destroy(void *h);
int oracle_vocabulary_words(void *h);
int oracle_vocabulary_transform(void *h, const uint8_t *desc, int n, int levelsup, int32_t *feat_word, int32_t *feat_node,
                                int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start,
                                int32_t *fv_feature, int32_t *n_fv);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) (src/ORBmatcher1.cc:225-427), F.Nleft == -1; FeatureVectors as CSR. */
int oracle_search_by_bow(const int32_t *kf_node, const int32_t *kf_start, const int32_t *kf_feat, int kf_nfv, const oracle_keypoint *kf_keys,
                         const uint8_t *kf_desc, const uint8_t *kf_valid, const int32_t *f_node, const int32_t *f_start, const int32_t *f_feat,
                         int f_nfv, const oracle_keypoint *f_keys, const uint8_t *f_desc, int f_n, float nnratio, int check_orientation,
                         int32_t *matches);

/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) (src/ORBmatcher2.cc:37-177), NLeft == -1. */
int oracle_search_by_bow_keyframes(const int32_t *node1, const int32_t *start1, const int32_t *feat1, int nfv1, const oracle_keypoint *keys1,
                                   const uint8_t *desc1, const uint8_t *valid1, int n1, const int32_t *node2, const int32_t *start2,
                                   const int32_t *feat2, int nfv2, const oracle_keypoint *keys2, const uint8_t *desc2, const uint8_t *valid2,
                                   int n2, float nnratio, int check_orientation, int32_t *matches12);

/* ORBmatcher::SearchForTriangulation (src/ORBmatcher2.cc:179-418).  kf1 / kf2 / geometry: structs with the layout of orbx_keyframe_view /
 * orbx_triangulation_geometry (include/orbx.h), restated here so that the oracle stays self-contained. */
typedef struct oracle_keyframe_view {
    const oracle_keypoint *keys; const uint8_t *desc; int32_t n, n_left; const float *u_right; const uint8_t *has_map_point;
    const int32_t *fv_node, *fv_start, *fv_feature; int32_t nfv; const float *scale_factors, *level_sigma2; int32_t nlevels;
} oracle_keyframe_view;
typedef struct oracle_triangulation_geometry {
    float ep[2]; int32_t model, rig; float F12[9]; float cam_left[8], cam_right[8], precision; float R[4][9], t[4][3];
} oracle_triangulation_geometry;
int oracle_search_for_triangulation(const oracle_keyframe_view *kf1, const oracle_keyframe_view *kf2, const oracle_triangulation_geometry *g,
                                    int only_stereo, int coarse, int check_orientation, int32_t *matches12, int32_t *pairs);

/* The per-map-point window search of ORBmatcher::Fuse (src/ORBmatcher2.cc:522-579, 690-714) and SearchBySim3 (src/ORBmatcher3.cc:120-145,
 * 188-213) over KeyFrame::GetFeaturesInArea (src/KeyFrame.cc:704-748).  points: {u, v, ur, radius, level, search} as orbx_window_point. */
typedef struct { float u, v, ur, radius; int32_t level, search; } oracle_window_point;
int oracle_search_keyframe_windows(const oracle_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x, float max_x,
                                   float min_y, float max_y, const float *inv_level_sigma2, const oracle_window_point *points,
                                   const uint8_t *point_desc, int np, int chi2_test, int32_t *best_idx, int32_t *best_dist);
int oracle_search_by_sim3(const oracle_keypoint *keys1, const uint8_t *kf_desc1, int n1, const float *bounds1, const oracle_keypoint *keys2,
                          const uint8_t *kf_desc2, int n2, const float *bounds2, const oracle_window_point *points_1in2, const uint8_t *mp_desc1,
                          const oracle_window_point *points_2in1, const uint8_t *mp_desc2, int32_t *matches12);

/* ORBmatcher::SearchByProjection(KeyFrame*, Sim3f&, vpPoints, vpMatched, th, ratioHamming) (src/ORBmatcher1.cc:429-533; :535-648). */
int oracle_search_by_projection_sim3(const oracle_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x, float min_y, float max_y,
                                     const oracle_window_point *points, const uint8_t *point_desc, int np, const uint8_t *matched,
                                     float ratio_hamming, int32_t *new_match);

/* ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, sAlreadyFound, th, ORBdist) (src/ORBmatcher3.cc:469-590). */
int oracle_search_by_projection_keyframe(const oracle_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x, float min_y,
                                         float max_y, const oracle_window_point *points, const uint8_t *point_desc, const float *point_angle,
                                         int np, const uint8_t *has_map_point, int orb_dist, int check_orientation, int32_t *new_match);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:329-403) for a batch of map points (CSR rows of `desc`). */
int oracle_distinctive_descriptors(const uint8_t *desc, const int32_t *start, int npoints, int32_t *best);

/* Deterministic synthetic frame (tools: same integer generator as the product's orbx_synth_frame; kept
 * here so the oracle library is self-contained for CPU-only tests). */
void oracle_synth_frame(uint64_t seed, int w, int h, uint8_t *out, ptrdiff_t stride);

#ifdef __cplusplus
}
#endif
#endif
