/*
 * orbx.h — C ABI of the MI355X-native ORB front-end (liborbx.so).
 *
 * Drop-in boundary for the ORB hot path of kpmrozowski/wut-cuda-orb-slam3.  Plain
 * pointers and sizes only; no C++/torch types cross this boundary.  Every entry point names the
 * reference interface it replaces (file:line relative to the reference repository root).
 *
 * Error model (reference: `return -1` on an empty image, src/ORBextractor.cc:1231-1232; exceptions /
 * exit(1) elsewhere, src/OpenCL/Manager.cpp:13-20): functions return an orbx_status (0 = OK, negative =
 * error) unless documented otherwise; orbx_last_error() returns a thread-local message.  The library
 * never exits the process and never throws across this boundary.  There is NO CPU fallback: without a
 * HIP device every compute entry point fails with ORBX_ERR_NO_DEVICE.
 *
 * Threading (reference: operator() of two different extractor objects runs in two std::threads,
 * src/Frame.cc:124-127): handles are independent and may be used concurrently from different threads;
 * one handle must not be entered concurrently.  Each handle owns one HIP stream.
 */
#ifndef ORBX_H
#define ORBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORBX_VERSION 100

typedef enum {
    ORBX_OK = 0,
    ORBX_ERR_INVALID_ARG = -1,
    ORBX_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime failure at init */
    ORBX_ERR_HIP = -3,         /* a HIP call or kernel launch failed (reference: throw std::runtime_error, ORBextractor.cc:499,862,974,1213) */
    ORBX_ERR_CAPACITY = -4,    /* image, batch or output capacity exceeds what the handle was created for */
    ORBX_ERR_TOO_SMALL = -5,   /* a pyramid level is too small for the 35-px FAST cell grid (the reference divides by zero there) */
    ORBX_ERR_OUT_OF_MEMORY = -6
} orbx_status;

/* 28-byte keypoint record == cv::KeyPoint == key_point_t (reference include/OpenCL/Manager.hpp:11-24,
 * include/OpenCL/Kernel/key_point.hpp): a C++ adaptor can memcpy into std::vector<cv::KeyPoint>. */
typedef struct {
    float x, y;       /* pt, level-0 image coordinates */
    float size;       /* (int)(31 * scaleFactor[octave]) */
    float angle;      /* degrees in [0, 360) */
    float response;   /* FAST score */
    int32_t octave;   /* pyramid level */
    int32_t class_id; /* -1 */
} orbx_keypoint;

/* GaussianBlur fixed-point tap table: the 8-bit taps depend on the OpenCV version. */
typedef enum {
    ORBX_BLUR_OPENCV_450 = 0, /* {18,34,49,55,49,34,18} — OpenCV <= 4.5.0 (the version in the reference's ExecMean.txt:2) */
    ORBX_BLUR_OPENCV_451 = 1  /* {18,34,48,56,48,34,18} — OpenCV >= 4.5.1 / 3.4.13 (sum forced to 256) */
} orbx_blur_variant;

/* Which overload `cos(angle)` / `sin(angle)` of computeOrbDescriptor resolves to (reference src/ORBextractor.cc:108: the call is
 * unqualified and `using namespace std` is commented out at :76-77).  With only <cmath> in scope it is ::cos(double) and the result
 * is narrowed to float; if any transitive header pulls in libstdc++'s <math.h> wrapper - or with upstream ORB-SLAM3's
 * `using namespace std` - it is the float overload, i.e. glibc's cosf / sinf.  The two differ in the last bit for 0.3 % / 0.65 % of
 * the angles.  Mode 1 evaluates glibc 2.35's cosf / sinf algorithm (double-precision polynomial after a one-multiply range
 * reduction), checked bit for bit against the real libm on every float in [0, 6.5] (tests/test_oracle_kat.py). */
typedef enum {
    ORBX_TRIG_DOUBLE = 0, /* (float)cos((double)angle) - the fork as written */
    ORBX_TRIG_FLOAT = 1   /* cosf(angle) / sinf(angle), glibc >= 2.28 */
} orbx_trig_mode;

/* Constructor arguments of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:58-59) plus the
 * device-side sizing the reference re-derives on every call (src/ORBextractor.cc:1315). */
typedef struct {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t device;      /* HIP device ordinal */
    int32_t max_width;   /* largest image the handle will see */
    int32_t max_height;
    int32_t max_batch;   /* frames per orbx_extract_batch call */
    int32_t blur_variant; /* orbx_blur_variant */
    int32_t trig_mode;    /* orbx_trig_mode */
} orbx_params;

typedef struct orbx_extractor orbx_extractor;

/* Fills *p with the reference defaults (nfeatures 1000, 1.2, 8 levels, 20/7, 640x480, batch 1). */
void orbx_default_params(orbx_params *p);

/* ORBextractor::ORBextractor — reference src/ORBextractor.cc:410-468.  Allocates all device memory and
 * pinned staging once; no allocation happens on the extraction path afterwards. */
int orbx_create(const orbx_params *p, orbx_extractor **out);
void orbx_destroy(orbx_extractor *ex);

/* Thread-local description of the last error on the calling thread. */
const char *orbx_last_error(void);

/* Getters — reference include/ORBextractor.h:70-90 (GetLevels, GetScaleFactor, GetScaleFactors,
 * GetInverseScaleFactors, GetScaleSigmaSquares, GetInverseScaleSigmaSquares).  Arrays hold nlevels entries. */
int orbx_get_levels(const orbx_extractor *ex);
float orbx_get_scale_factor(const orbx_extractor *ex);
int orbx_get_scale_factors(const orbx_extractor *ex, float *out);
int orbx_get_inverse_scale_factors(const orbx_extractor *ex, float *out);
int orbx_get_scale_sigma_squares(const orbx_extractor *ex, float *out);
int orbx_get_inverse_scale_sigma_squares(const orbx_extractor *ex, float *out);
/* mnFeaturesPerLevel (src/ORBextractor.cc:436-447) and umax[16] (:455-467) */
int orbx_get_features_per_level(const orbx_extractor *ex, int32_t *out);
int orbx_get_umax(const orbx_extractor *ex, int32_t *out16);
/* Output capacity a caller must provide per frame: the sum over levels of max(quota + 3, 16, 4 * nIni) - DistributeOctTree can
 * overshoot its quota by up to 3 (src/ORBextractor.cc:697), and its first pass splits each of the nIni = round(W / H) root nodes
 * (:589) into four before any quota is looked at (very wide images with small quotas). */
int orbx_max_keypoints(const orbx_extractor *ex);

/* ORBextractor::operator()(image, mask (ignored), keypoints, descriptors, vLappingArea) —
 * reference include/ORBextractor.h:66-68, src/ORBextractor.cc:1227-1307, invoked through
 * Frame::ExtractORB (src/Frame.cc:430-455).
 * img: host pointer, 8-bit single channel, `stride` bytes per row.  Keypoints with pt.x in [lap0, lap1] are
 * written from the back of the output in reverse order, the rest from the front (:1293-1301).
 * Returns monoIndex (>= 0), or -1 for an empty image (img == NULL or w/h <= 0) exactly like the reference;
 * library errors are kept apart from that -1 by returning (orbx_status - 100), i.e. values <= -101. */
int orbx_extract(orbx_extractor *ex, const uint8_t *img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                 orbx_keypoint *kps, uint8_t *desc /* capacity x 32, row-major */, int capacity, int *n_out);

/* Batch form for the multi-frame configurations: nframes images, frame f at imgs + f*frame_stride.
 * Per frame f: n_out[f] keypoints, mono_out[f] = monoIndex; kps/desc rows of frame f start at f*capacity.
 * Host pointers in, host pointers out (H2D and D2H included).  Returns ORBX_OK or an orbx_status. */
int orbx_extract_batch(orbx_extractor *ex, const uint8_t *imgs, int nframes, int w, int h, ptrdiff_t stride,
                       ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc, int capacity,
                       int32_t *n_out, int32_t *mono_out);

/* orbx_extract_batch in two halves, for callers that keep the link busy across calls.  submit queues every copy and launch of the batch
 * and returns without waiting; wait blocks until the results of the OLDEST submitted batch are in the caller's arrays (n_out / mono_out
 * included) and returns that call's status.  TWO batches may be pending on a handle: a host-pointer call ends in a tail nothing of the
 * same call can cover (the last sub-batch's kernels and its results, ~0.9 of 6.7 ms for 1024 VGA frames), and with
 *     submit(b0); submit(b1); for (;;) { wait(); submit(next); }
 * the frames of the following batch cross the link - into a second staging buffer - under the kernels and the copy-out of the one before:
 * the link never waits for the GPU (the reference keeps two extractor objects busy side by side for the same reason, src/Frame.cc:124-127,
 * 1089-1092).  Two HANDLES in flight from one process do not reach this: their streams share the runtime's four hardware queues and
 * serialise behind each other's barrier packets (profiles/r05_submit_wait.txt).  Requirements: the frames in page-locked memory
 * (orbx_host_alloc / orbx_host_register; pageable frames need the calling thread: ORBX_ERR_INVALID_ARG); frames and result arrays stay
 * valid and untouched until the batch's wait returns; while a batch is pending every other extraction call on the handle is refused
 * (a second submit of a batch of more than 16 frames excepted), and the handle's device outputs (orbx_device_outputs, the *_frames
 * matcher calls) belong to the newest pending batch.  A batch of <= 16 frames (the single-frame schedule) can be submitted only alone.
 * An error of a submit or a wait drops every pending batch (nothing of them still runs when the call returns).  orbx_destroy drains
 * batches nobody waited for.  One thread per handle, as for every other call. */
int orbx_extract_batch_submit(orbx_extractor *ex, const uint8_t *imgs, int nframes, int w, int h, ptrdiff_t stride,
                              ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc, int capacity,
                              int32_t *n_out, int32_t *mono_out);
int orbx_extract_batch_wait(orbx_extractor *ex);

/* The same batch dealt over several handles - normally one per GPU of the node (orbx_params.device) - from ONE host process: block b of
 * `block` consecutive frames goes to handles[b mod nhandles] (block = 1: frame f -> GPU f mod G, the partition of SURVEY 8(e); larger
 * blocks keep each handle's host copies long), every handle runs on a host thread of its own - the reference drives its left and right
 * extractor from two std::threads in the same way (src/Frame.cc:124-127, 1089-1092) - in batches of at most its max_batch frames, and
 * every result lands at its frame index in the caller's arrays, laid out as for orbx_extract_batch.  No collective: frames are
 * independent (multi-process use over RCCL: wut-cuda-orb-slam3_amd/sharding.py).  All handles must have been created with the same
 * five extractor parameters; a handle may appear only once. */
int orbx_extract_batch_sharded(orbx_extractor *const *handles, int nhandles, int block, const uint8_t *imgs, int nframes, int w, int h,
                               ptrdiff_t stride, ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc,
                               int capacity, int32_t *n_out, int32_t *mono_out);

/* The ORBX_* environment switches the library reads (A / B measurements and tests; none is needed - DESIGN.md appendix).  orbx_create /
 * orbx_matcher_create refuse an ORBX_* variable that is not one of them (ORBX_ERR_INVALID_ARG): a mistyped switch must not silently run the
 * default.  Writes up to `capacity` names, returns their number. */
int orbx_known_switches(const char **names, int capacity);

/* Page-locked host memory for frames and result arrays.  orbx_extract_batch works from any host memory; frames in page-locked memory
 * cross the link asynchronously without a helper thread, and result arrays in page-locked memory (with capacity == orbx_max_keypoints)
 * are written by the device directly.  In that direct mode the device copies FULL-CAPACITY rows: the rows of frame f beyond n_out[f]
 * are overwritten with unspecified bytes (with pageable or differently laid-out result arrays exactly n_out[f] rows per frame are
 * written and the rest is left alone).  The whole range a call reads / writes must be page-locked for the asynchronous routes to be
 * taken: a range registered only in part falls back to the staged routes.  (The reference has no counterpart: its cv::Mat /
 * std::vector buffers are pageable and re-allocated per call, src/ORBextractor.cc:1251,1257,1315.) */
void *orbx_host_alloc(size_t bytes);
void orbx_host_free(void *p);
/* The same for buffers the caller already owns (the cv::Mat data of a camera ring, a std::vector of frames): page-locks the range
 * [p, p + bytes) in place so that orbx_extract_batch takes its asynchronous route from it; orbx_host_unregister before the memory is
 * freed.  Registering costs milliseconds per 100 MB: once per buffer, not per call.  Returns ORBX_OK or a negative orbx_status. */
int orbx_host_register(void *p, size_t bytes);
int orbx_host_unregister(void *p);

/* Device-resident batch: d_imgs is a device pointer (frames already in HBM); results stay in HBM in the
 * handle's own output buffers, retrievable with orbx_device_outputs().  Launches asynchronously on the
 * handle's stream; orbx_sync() waits.  This is the path bench.py times. */
int orbx_extract_batch_device(orbx_extractor *ex, const void *d_imgs, int nframes, int w, int h, ptrdiff_t stride,
                              ptrdiff_t frame_stride, int lap0, int lap1);
int orbx_sync(orbx_extractor *ex);
/* Device pointers of the last batch's outputs: kps [max_batch][capacity], desc [max_batch][capacity][32],
 * n [max_batch], mono [max_batch]; capacity = orbx_max_keypoints(). */
int orbx_device_outputs(orbx_extractor *ex, void **d_kps, void **d_desc, void **d_n, void **d_mono);
/* Copies the last batch's outputs to host arrays laid out as in orbx_extract_batch. */
int orbx_download_outputs(orbx_extractor *ex, int nframes, orbx_keypoint *kps, uint8_t *desc, int capacity,
                          int32_t *n_out, int32_t *mono_out);
/* The HIP stream (hipStream_t) the handle launches on — for event timing by the caller. */
void *orbx_stream(orbx_extractor *ex);

/* Stage timing with HIP events recorded on the handle's own stream (the reference's counterpart is the
 * orb::benchmark MEASURE_* wrappers around the same stages, include/orb/Benchmark.h:83-92; reference
 * src/ORBextractor.cc:969,998,1238,1241,1273,1278).  When enabled, every batch records events around each
 * stage; orbx_get_stage_times() synchronises the stream and returns the mean milliseconds per batch of the
 * batches recorded since the last call (at most the 32 most recent), then resets.  A batch is software-pipelined
 * in sub-batches over three streams, so stage times are sums over the sub-batches, measured on the stream the
 * stage runs on, and they overlap in time (they do not add up to the batch time).  Stage order:
 * 0 = fused pyramid+FAST (all per-level launches), 1 = quadtree, 2 = GaussianBlur (all levels), 3 = pack,
 * 4 = orientation+descriptor.  Returns the number of batches averaged (0 if none). */
#define ORBX_NUM_STAGES 5
int orbx_set_profiling(orbx_extractor *ex, int enable);
int orbx_get_stage_times(orbx_extractor *ex, float *ms /* ORBX_NUM_STAGES */);

/* mvImagePyramid[level] — public member read by Frame::ComputeStereoMatches (reference
 * include/ORBextractor.h:92, src/Frame.cc:848,938-953).  Copies the interior (un-padded) level image of
 * frame `frame` of the last batch to `out` (w*h bytes, tight).  blurred != 0 returns the GaussianBlur-ed
 * working image (src/ORBextractor.cc:1270-1273) instead. */
int orbx_get_level_size(const orbx_extractor *ex, int level, int *w, int *h);
int orbx_get_pyramid_level(orbx_extractor *ex, int frame, int level, int blurred, uint8_t *out);

/* Stage taps for parity tests (no reference counterpart): FAST candidates of a level in
 * tileCalcKeypoints order (src/ORBextractor.cc:867-950), packed x | y<<12 | score<<24 relative to the 16-px
 * border; and the per-level DistributeOctTree output in list order (same packing). Return the count. */
int orbx_debug_get_candidates(orbx_extractor *ex, int frame, int level, uint32_t *out, int cap);
int orbx_debug_get_selected(orbx_extractor *ex, int frame, int level, uint32_t *out, int cap);

/* ---- matcher leafs ------------------------------------------------------------------------------ */

/* ORBmatcher::DescriptorDistance(a, b) — reference include/ORBmatcher.h:43, src/ORBmatcher3.cc:637-653
 * (duplicate: Thirdparty/DBoW2/DBoW2/FORB.cpp:81-101).  Host inline; a, b point to 32 bytes. */
static inline int orbx_hamming256(const void *a, const void *b)
{
    const uint64_t *pa = (const uint64_t *)a, *pb = (const uint64_t *)b;
    int d = 0;
    for (int i = 0; i < 4; ++i) {
        uint64_t v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x5555555555555555ull);
        v = (v & 0x3333333333333333ull) + ((v >> 2) & 0x3333333333333333ull);
        d += (int)((((v + (v >> 4)) & 0x0F0F0F0F0F0F0F0Full) * 0x0101010101010101ull) >> 56);
    }
    return d;
}

typedef struct orbx_matcher orbx_matcher;
/* Matcher context: device ordinal + capacities (max query / train rows). */
int orbx_matcher_create(int device, int max_query, int max_train, orbx_matcher **out);
void orbx_matcher_destroy(orbx_matcher *m);

/* All-pairs distance matrix dist[nq][nt] (int32) — the loop body every ORBmatcher search routine runs
 * (18 call sites of DescriptorDistance).  Host pointers. */
int orbx_hamming_matrix(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *dist);

/* cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, matches, 2) — reference src/Frame.cc:45,1174.  Per query the
 * two smallest distances, ties ranked by lower train index; idx = -1 and dist = INT32_MAX when fewer than
 * k train rows exist.  Host pointers; idx/dist are [nq][2]. */
int orbx_match_knn2(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx,
                    int32_t *dist);
/* Same with device pointers (descriptors straight from orbx_device_outputs), asynchronous on the
 * matcher's stream; orbx_matcher_sync() waits. */
int orbx_match_knn2_device(orbx_matcher *m, const void *d_q, int nq, const void *d_t, int nt, void *d_idx,
                           void *d_dist);
/* The same for `npairs` independent (query, train) sets in ONE launch, everything device-resident and asynchronous - the
 * form in which Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:1156-1196) runs over a batch of stereo pairs:
 *   matcher->knnMatch(mDescriptors.rowRange(monoLeft, rows), mDescriptorsRight.rowRange(monoRight, rows), matches, 2)  (:1174)
 * Pair p reads its query rows at d_q + p * q_pair_stride + 32 * (q_begin .. nq_p - 1) with nq_p = d_nq[p] (device array; NULL:
 * max_nq rows for every pair) and q_begin = d_q_begin[p] (device array; NULL: 0), likewise the train rows - so the descriptors,
 * counts and monoIndex values of two extractors' orbx_device_outputs() plug in directly (pair stride = capacity * 32).
 * idx / dist are [npairs][max_nq][2]; row i of pair p belongs to query row q_begin + i and train indices are relative to
 * t_begin, as cv::DMatch::queryIdx / trainIdx are relative to the rowRange.  Rows i >= nq_p - q_begin are not written. */
int orbx_match_knn2_batch_device(orbx_matcher *m, int npairs, const void *d_q, int64_t q_pair_stride, const int32_t *d_nq,
                                 const int32_t *d_q_begin, int max_nq, const void *d_t, int64_t t_pair_stride,
                                 const int32_t *d_nt, const int32_t *d_t_begin, int max_nt, void *d_idx, void *d_dist);
/* The batch form bound to two extractors: pair p = frame p of `query`'s and `train`'s last batch (orbx_extract_batch[_device]);
 * descriptors, counts and - with lapping_only != 0 - the monoIndex row offsets are read from the extractors' device outputs.
 * The call orders the matcher's stream behind both extractors' work and the extractors' next batch behind the matcher, so it can
 * be issued right after orbx_extract_batch_device without a host synchronisation.  d_idx / d_dist: device buffers
 * [npairs][orbx_max_keypoints(query)][2] (int32). */
int orbx_match_knn2_frames(orbx_matcher *m, orbx_extractor *query, orbx_extractor *train, int npairs, int lapping_only,
                           void *d_idx, void *d_dist);
/* Copies `bytes` from a device buffer to the host on the matcher's stream and waits for it (results of the asynchronous,
 * device-resident entry points). */
int orbx_matcher_download(orbx_matcher *m, const void *d_src, void *h_dst, size_t bytes);
int orbx_matcher_sync(orbx_matcher *m);

/* Hamming stage of Frame::ComputeStereoMatches — reference src/Frame.cc:841-925: for each left keypoint
 * the best right keypoint whose row band [floor(y-r), ceil(y+r)], r = 2*scale[octave], contains the left
 * row, with |octave difference| <= 1 and uR in [uL-max_d, uL-min_d]; best_dist starts at TH_HIGH = 100
 * (src/ORBmatcher1.cc:37).  best_idx = -1 when none. Host pointers. */
int orbx_match_stereo_rows(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int nl,
                           const orbx_keypoint *kr, const uint8_t *dr, int nr, int n_rows, const float *scale_factors,
                           int nlevels, float min_d, float max_d, int32_t *best_idx, int32_t *best_dist);

/* ---- the frame grid and the windowed initialization search -------------------------------------- */
#define ORBX_FRAME_GRID_COLS 64 /* FRAME_GRID_COLS, reference include/Frame.h:45 */
#define ORBX_FRAME_GRID_ROWS 48 /* FRAME_GRID_ROWS, reference include/Frame.h:44 */

/* Frame::AssignFeaturesToGrid + Frame::PosInGrid — reference src/Frame.cc:387-418, 755-766 (monocular branch: mvKeysUn).
 * Buckets the n keypoints into the 64 x 48 grid over [min_x, max_x) x [min_y, max_y) (Frame::mnMinX .. mnMaxY, src/Frame.cc:41,
 * 808-839) and returns it as CSR: cell_start[64 * 48 + 1] with cell = ix * 48 + iy (mGrid[ix][iy]), items[] = the indices of
 * every cell in push_back order.  Returns the number of keypoints placed (those inside the grid) or a negative status. */
int orbx_assign_features_to_grid(orbx_matcher *m, const orbx_keypoint *kps, int n, float min_x, float max_x, float min_y, float max_y,
                                 int32_t *cell_start, int32_t *items);

/* int ORBmatcher::SearchForInitialization(Frame &F1, Frame &F2, vector<cv::Point2f> &vbPrevMatched, vector<int> &vnMatches12,
 * int windowSize) — reference src/ORBmatcher1.cc:650-765 (decl include/ORBmatcher.h:72), with Frame::GetFeaturesInArea
 * (src/Frame.cc:687-753), DescriptorDistance, TH_LOW = 50, the NN ratio test and the 30-bin rotation histogram +
 * ORBmatcher::ComputeThreeMaxima (src/ORBmatcher3.cc:592-633) when check_orientation != 0 (mbCheckOrientation).
 * k1/d1, k2/d2 = F1 / F2 .mvKeysUn and .mDescriptors; prev_matched = vbPrevMatched as n1 (x, y) pairs, updated in place like
 * the reference does; matches12[n1] receives vnMatches12 (-1 = no match).  nnratio = mfNNratio (the reference's monocular
 * initializer constructs ORBmatcher(0.9, true) and passes windowSize 100, src/Tracking3.cc:740-741).  Returns nmatches (>= 0) or a negative status. */
int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *k1, const uint8_t *d1, int n1, const orbx_keypoint *k2,
                                   const uint8_t *d2, int n2, float min_x, float max_x, float min_y, float max_y, float *prev_matched,
                                   int32_t *matches12, int window_size, float nnratio, int check_orientation);

/* The same search for a BATCH of frame pairs, device-resident (one wave per pair; the reference's loop is sequential inside a pair):
 * pair p = (frame p of `ex1`'s last batch, frame p + frame2_offset of `ex2`'s - the same handle with offset 1 pairs consecutive frames of
 * one batch).  Both frames' keypoints go through Frame::UndistortKeyPoints (src/Frame.cc:777-811; k4 / dist / knew4 as in
 * orbx_undistort_keypoints) first.  d_prev_matched [npairs][cap1][2] floats = vbPrevMatched, in / out; with init_prev_matched != 0 it is
 * first set to F1's undistorted keypoint positions (what MonocularInitialization passes on its first attempt, src/Tracking3.cc:717-720).
 * d_matches12 [npairs][cap1] int32 = vnMatches12, d_nmatches [npairs] (-1: F2 holds more than 2048 octave-0 keypoints, the capacity of the
 * LDS-resident search).  cap1 = orbx_max_keypoints(ex1); ORBX_ERR_CAPACITY when cap1 exceeds what the search keeps in LDS per F1 keypoint
 * (5 bytes each beside the 2048-keypoint subset of F2: about 7000 keypoints).  Asynchronous on the matcher's stream. */
int orbx_search_for_initialization_frames(orbx_matcher *m, orbx_extractor *ex1, orbx_extractor *ex2, int frame2_offset, int npairs,
                                          const float *k4, const float *dist, int ndist, const float *knew4, float min_x, float max_x,
                                          float min_y, float max_y, float *d_prev_matched, int init_prev_matched, int window_size,
                                          float nnratio, int check_orientation, void *d_matches12, void *d_nmatches);

/* One projected map point of the last frame, as ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, ..)
 * has it after `x3Dc = Tcw * x3Dw; invzc = 1.0 / x3Dc(2); uv = mpCamera->project(x3Dc)` (reference src/ORBmatcher3.cc:283-296). */
typedef struct {
    float u, v, invz; /* uv(0), uv(1), invzc */
    float angle;      /* kpLF.angle: LastFrame.mvKeysUn[i].angle (:345) */
    int32_t octave;   /* nLastOctave: LastFrame.mvKeys[i].octave (:303) */
    int32_t flags;    /* bit 0: LastFrame.mvpMapPoints[i] != NULL && !LastFrame.mvbOutlier[i] (:277-280);
                         bit 1: that map point has Observations() > 0 (decides whether it blocks later points, :325-327) */
} orbx_projection;

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono) — reference
 * src/ORBmatcher3.cc:256-467 (decl include/ORBmatcher.h:54), the CurrentFrame.Nleft == -1 branch (monocular / rectified stereo),
 * called every frame by Tracking::TrackWithMotionModel (src/Tracking4.cc:210,218).  The pose / camera arithmetic of the loop head
 * (Sophus `Tcw * x3Dw`, GeometricCamera::project) stays with the caller, who passes the projections; the library does the bounds
 * tests (invzc < 0, uv outside mnMinX .. mnMaxY), the window search radius = th * scale_factors[octave] with
 * Frame::GetFeaturesInArea (forward: levels >= octave, backward: <= octave, else octave - 1 .. octave + 1), the mvuRight
 * consistency test, DescriptorDistance against proj_desc[i] (MapPoint::GetDescriptor), TH_HIGH = 100, and the rotation histogram.
 *   kc / dc / nc        CurrentFrame.mvKeysUn / mDescriptors / N;  u_right = CurrentFrame.mvuRight or NULL (monocular)
 *   cur_state[nc]       CurrentFrame.mvpMapPoints before the call: -1 NULL, 0 a point with Observations() == 0, 1 with > 0
 *   forward / backward  bForward / bBackward (:271-272): tlc(2) > mb && !bMono, -tlc(2) > mb && !bMono
 *   assign[nc]          what the call did to CurrentFrame.mvpMapPoints[i2]: -1 untouched, j >= 0 now LastFrame.mvpMapPoints[j],
 *                       -2 set to NULL by the rotation-consistency check (:447-463)
 * Returns nmatches (>= 0) or a negative status.  At most 512 window candidates per projected point. */
int orbx_search_by_projection(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, const float *u_right, const int8_t *cur_state,
                              int nc, float min_x, float max_x, float min_y, float max_y, const float *scale_factors, int nlevels,
                              float mbf, const orbx_projection *proj, const uint8_t *proj_desc, int np, float th, int forward,
                              int backward, int check_orientation, int32_t *assign);

/* The CurrentFrame.Nleft != -1 branch of the same function (two KB8 fisheye cameras, reference src/ORBmatcher3.cc:300-436): per map point
 * of the last frame the left window search on mGrid / mvKeys (no mvuRight test) and then the right one on mGridRight / mvKeysRight
 * (:373-436), which writes CurrentFrame.mvpMapPoints[bestIdx2 + Nleft].  kc / dc = mvKeys followed by mvKeysRight / mDescriptors (the
 * vconcat of both sides, src/Frame.cc:1138-1142); cur_state / assign run over n_left + n_right.  proj_left[i] = the projection with
 * mpCamera as above, flags / octave / angle from the last frame's keypoint i (mvKeys or mvKeysRight, :300-301, :358-360); of
 * proj_right[i] only u, v are read = mpCamera->project(GetRelativePoseTrl() * x3Dc) (:374-375).  The `continue`s of the left part
 * (invzc < 0, uv outside the bounds, an empty left window :315) skip the right part too, as in the reference. */
int orbx_search_by_projection_fisheye(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, int n_left, int n_right,
                                      const int8_t *cur_state, float min_x, float max_x, float min_y, float max_y, const float *scale_factors,
                                      int nlevels, const orbx_projection *proj_left, const orbx_projection *proj_right,
                                      const uint8_t *proj_desc, int np, float th, int forward, int backward, int check_orientation,
                                      int32_t *assign);

/* The tracker's every-frame matcher over a BATCH of frames, device-resident end to end (nothing crosses PCIe between the extractor and
 * the match result): frame p = frame p of `ex`'s last batch; its keypoints go through Frame::UndistortKeyPoints (src/Frame.cc:777-811;
 * k4 / dist / knew4 as in orbx_undistort_keypoints, knew4 NULL = k4), Frame::AssignFeaturesToGrid (:387-418) over [min_x, max_x) x
 * [min_y, max_y) (orbx_compute_image_bounds, once per camera) and ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
 * (src/ORBmatcher3.cc:256-467) against the projections d_proj[p][0 .. d_np[p]) / d_proj_desc[p][..][32] of "its" last frame - device
 * arrays with `max_np` records per frame.  d_u_right [nframes][cap] = mvuRight or NULL (monocular); d_cur_state [nframes][cap] =
 * mvpMapPoints state before the call (-1 / 0 / 1) or NULL (all empty).  Outputs, device: d_kps_un [nframes][cap] = mvKeysUn (may be
 * NULL), d_assign [nframes][cap] as `assign` above, d_nmatches [nframes] (-1: a search window held more than 512 candidates).
 * cap = orbx_max_keypoints(ex).  Six launches for the whole batch, asynchronous on the matcher's stream. */
int orbx_search_by_projection_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                     const float *knew4, float min_x, float max_x, float min_y, float max_y, float mbf,
                                     const float *d_u_right, const int8_t *d_cur_state, const orbx_projection *d_proj,
                                     const uint8_t *d_proj_desc, const int32_t *d_np, int max_np, float th, int forward, int backward,
                                     int check_orientation, void *d_kps_un, void *d_assign, void *d_nmatches);

/* The same batch pipeline for Tracking::SearchLocalPoints' matcher, ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &, th,
 * bFarPoints, thFarPoints) (reference src/ORBmatcher1.cc:45-215): d_map_points[p][..] are the records of orbx_search_by_projection_points. */
int orbx_search_by_projection_points_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                            const float *knew4, float min_x, float max_x, float min_y, float max_y,
                                            const float *d_u_right, const int8_t *d_cur_state, const orbx_projection *d_map_points,
                                            const uint8_t *d_map_point_desc, const int32_t *d_np, int max_np, float th, float nnratio,
                                            void *d_kps_un, void *d_assign, void *d_nmatches);

/* The CurrentFrame.Nleft != -1 branches of both functions (two KB8 fisheye cameras: reference src/ORBmatcher3.cc:300-436 and
 * src/ORBmatcher1.cc:45-215, see orbx_search_by_projection_fisheye / orbx_search_by_projection_points_fisheye) for a BATCH of frames,
 * device-resident: frame p = frame p of `left`'s last batch (mvKeys) + frame p of `right`'s (mvKeysRight); no undistortion (KB8 frames
 * search on mvKeys).  Combined keypoint indices - rows of d_cur_state and d_assign, [nframes][capL + capR] with cap = orbx_max_keypoints -
 * are i for left keypoint i and capL + j for right keypoint j (rows beyond a frame's counts are not written).  d_proj_left / d_proj_right
 * [nframes][max_np] and d_proj_desc [nframes][max_np][32] as proj_left / proj_right / proj_desc of the single-frame calls, d_np [nframes].
 * The map-point variant takes mvLeftToRightMatch / mvRightToLeftMatch as d_left_to_right [nframes][capL] / d_right_to_left
 * [nframes][capR] (what orbx_compute_stereo_fisheye_matches_frames writes).  d_nmatches [nframes] (-1: a window held more than 512
 * candidates).  Asynchronous on the matcher's stream. */
int orbx_search_by_projection_fisheye_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int nframes, float min_x, float max_x,
                                             float min_y, float max_y, const int8_t *d_cur_state, const orbx_projection *d_proj_left,
                                             const orbx_projection *d_proj_right, const uint8_t *d_proj_desc, const int32_t *d_np, int max_np,
                                             float th, int forward, int backward, int check_orientation, void *d_assign, void *d_nmatches);
int orbx_search_by_projection_points_fisheye_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int nframes, float min_x,
                                                    float max_x, float min_y, float max_y, const int8_t *d_cur_state,
                                                    const orbx_projection *d_map_points_left, const orbx_projection *d_map_points_right,
                                                    const uint8_t *d_map_point_desc, const int32_t *d_np, int max_np,
                                                    const int32_t *d_left_to_right, const int32_t *d_right_to_left, float th, float nnratio,
                                                    void *d_assign, void *d_nmatches);

/* ---- The projection head of the two tracking searches (round 4): "project-then-match" with the PROJECT half on the device ----------
 * The record-taking entry points above stay; these compute the orbx_projection records from world points and a pose, in the reference's
 * own float arithmetic - every +, -, *, / in float and un-fused, in the order Eigen / Sophus evaluate it (3-term sums as x0 + (x1 + x2):
 * Eigen's unrolled redux of a 3-vector; the reference builds with -O3 and no -march, so nothing is contracted), sqrt correctly rounded,
 * 1.0 / z in double narrowed once.  libm calls of the reference - atan2f (KannalaBrandt8::project), logf (MapPoint::PredictScale) - are
 * evaluated with include/orbx_libm.h, glibc 2.35's algorithms restated and PINNED exhaustively against the real libm
 * (tests/tools/libm_pin.cpp, profiles/r05_libm_pin.txt).  PARITY: bit-equal to the CPU oracle (oracle_project_*), which calls the real libm;
 * the Eigen evaluation order is restated from Eigen's sources, not checked against a build of the reference (Eigen / Sophus are not in
 * this image): "unpinned" in that respect only. */

/* GeometricCamera::project(const Eigen::Vector3f &): type 0 = Pinhole (src/CameraModels/Pinhole.cpp:43-49: fx * x / z + cx),
 * type 1 = KannalaBrandt8 (src/CameraModels/KannalaBrandt8.cpp:67-84); p = mvParameters: fx fy cx cy [k0 k1 k2 k3]. */
typedef struct {
    int32_t type;
    float p[8];
} orbx_camera;

/* What the two loops read of one MapPoint (and, for the motion-model search, of the last frame's keypoint that observes it). */
typedef struct {
    float pos[3];      /* pMP->GetWorldPos() */
    float normal[3];   /* pMP->GetNormal()                                                   (in-frustum mode only) */
    float min_distance, max_distance; /* mfMinDistance, mfMaxDistance: GetMinDistanceInvariance() = 0.8f * min, GetMax.. = 1.2f * max
                                         (src/MapPoint.cc; in-frustum mode only) */
    float angle;       /* LastFrame.mvKeysUn[i].angle                                        (motion-model mode only) */
    int32_t octave;    /* LastFrame.mvKeys[i].octave                                         (motion-model mode only) */
    int32_t flags;     /* orbx_projection.flags before the geometric tests: bit 0 the point takes part (exists, !isBad(), not an outlier /
                          not already matched in this frame), bit 1 Observations() > 0 */
} orbx_world_point;

/* Mode 0, the head of ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, ..) - reference
 * src/ORBmatcher3.cc:281-296: x3Dc = Tcw * x3Dw with Tcw a Sophus::SE3f (unit quaternion q = x, y, z, w and translation t;
 * Thirdparty/Sophus/sophus/so3.hpp:358-367: uv = q.vec x p, uv += uv, p + w * uv + q.vec x uv; se3.hpp: + t), invzc = 1.0 / x3Dc(2),
 * uv = mpCamera->project(x3Dc).  proj_out[i] = {uv(0), uv(1), invzc, angle, octave, flags} - the tests on invzc and the image bounds
 * stay in the search (orbx_search_by_projection), as in the reference's loop. */
int orbx_project_last_frame_points(orbx_matcher *m, const float q_xyzw[4], const float t[3], const orbx_camera *cam,
                                   const orbx_world_point *points, int np, orbx_projection *proj_out);

/* Mode 1, Frame::isInFrustum(MapPoint *pMP, float viewingCosLimit) - reference src/Frame.cc:542-600 (Nleft == -1), as
 * Tracking::SearchLocalPoints runs it over the local map (src/Tracking4.cc:700-737) - plus the far-point test at the head of
 * ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &, ..) (src/ORBmatcher1.cc:57-58).  Rcw (row-major 3 x 3), tcw, Ow =
 * Frame::mRcw, mtcw, mOw; bounds = mnMinX, mnMaxX, mnMinY, mnMaxY; log_scale_factor / nlevels = mfLogScaleFactor / mnScaleLevels
 * (MapPoint::PredictScale, src/MapPoint.cc:531-546: ceil(log(mfMaxDistance / dist) / mfLogScaleFactor) in FLOAT - `using namespace std`
 * reaches MapPoint.cc (MapPoint.h:23 -> .. -> Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:36), so the call is std::log(float) = logf, the
 * division float / float, std::ceil(float) - then the int conversion and the clamps; pass log_scale_factor = logf(mfScaleFactor)).
 * proj_out[i] = {mTrackProjX, mTrackProjY, mTrackProjXR = uv(0) - mbf * invz, mTrackViewCos, mnTrackScaleLevel, flags} with flags bit 0 =
 * the input bit && mbTrackInView && !(far_points && mTrackDepth > th_far_points) - exactly the record orbx_search_by_projection_points
 * takes.  (logf is glibc's, restated in include/orbx_libm.h and equal to it on every positive float: the level is the reference's.) */
int orbx_project_in_frustum(orbx_matcher *m, const float Rcw[9], const float tcw[3], const float Ow[3], const orbx_camera *cam,
                            const float bounds[4], float mbf, float viewing_cos_limit, float log_scale_factor, int nlevels,
                            int far_points, float th_far_points, const orbx_world_point *points, int np, orbx_projection *proj_out);

/* Both modes for a BATCH of frames, device-resident: d_pose [nframes][16] floats per frame - mode 0: q x y z w, t x y z; mode 1: Rcw[9],
 * tcw[3], Ow[3] -, d_points [nframes][max_np], d_np [nframes]; d_proj_out [nframes][max_np] is what
 * orbx_search_by_projection_frames / orbx_search_by_projection_points_frames take as d_proj / d_map_points (same stream: no
 * synchronisation in between).  The scalar parameters after `cam` are read in mode 1 only.  One launch, asynchronous on the matcher's
 * stream. */
int orbx_project_points_frames(orbx_matcher *m, int mode, int nframes, const float *d_pose, const orbx_camera *cam, const float bounds[4],
                               float mbf, float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points,
                               float th_far_points, const orbx_world_point *d_points, const int32_t *d_np, int max_np,
                               orbx_projection *d_proj_out);

/* Mode 2 (round 5), Frame::isInFrustum for a FISHEYE RIG - the Nleft != -1 branch, reference src/Frame.cc:604-614: isInFrustumChecks(pMP,
 * viewingCosLimit) for the left camera and isInFrustumChecks(pMP, viewingCosLimit, true) for the right one (:1198-1272).  Rcw / tcw / Ow =
 * Frame::mRcw, mtcw, mOw; Rrl (row-major) / trl = mTrl.rotationMatrix() / mTrl.translation(), tlr = mTlr.translation(), Rwc = mRwc: the
 * right camera's pose is formed from them per call as the reference does (:1204-1210: mR = Rrl * mRcw, mt = Rrl * mtcw + trl, twc = mRwc *
 * tlr + mOw, fixed-size Eigen products coefficient by coefficient).  cam1 / cam2 = mvParameters of mpCamera / mpCamera2 (KannalaBrandt8,
 * 8 floats).  proj_left_out[i] / proj_right_out[i] = {mTrackProjX(R), mTrackProjY(R), mTrackDepth(R), mTrackViewCos(R), mnTrackScaleLevel(R),
 * flags} - what orbx_search_by_projection_points_fisheye(_frames) takes as map_points_left / map_points_right; flags bit 0 = the input bit &&
 * mbTrackInView(R) && !(far_points && mTrackDepth > th_far_points): the far-point test of ORBmatcher::SearchByProjection
 * (src/ORBmatcher1.cc:57-58) reads the LEFT camera's depth for both sides.  Stated deviation: the reference stores mTrackDepth only when the
 * left check passes to its end, so for a point the left camera rejects it compares whatever an earlier frame left in the MapPoint; here the
 * test uses the left camera's Pc_dist of this frame.  Arithmetic as in mode 1 (atan2f / logf: include/orbx_libm.h). */
int orbx_project_in_frustum_fisheye(orbx_matcher *m, const float Rcw[9], const float tcw[3], const float Ow[3], const float Rrl[9],
                                    const float trl[3], const float tlr[3], const float Rwc[9], const float cam1[8], const float cam2[8],
                                    const float bounds[4], float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points,
                                    float th_far_points, const orbx_world_point *points, int np, orbx_projection *proj_left_out,
                                    orbx_projection *proj_right_out);
/* The same for a batch of frames, device-resident: d_pose [nframes][40] floats per frame = Rcw[9] tcw[3] Ow[3] Rrl[9] trl[3] tlr[3] Rwc[9]
 * and one float of padding; outputs [nframes][max_np] each.  One launch, asynchronous on the matcher's stream. */
int orbx_project_in_frustum_fisheye_frames(orbx_matcher *m, int nframes, const float *d_pose, const float cam1[8], const float cam2[8],
                                           const float bounds[4], float viewing_cos_limit, float log_scale_factor, int nlevels,
                                           int far_points, float th_far_points, const orbx_world_point *d_points, const int32_t *d_np,
                                           int max_np, orbx_projection *d_proj_left_out, orbx_projection *d_proj_right_out);

/* Mode 3 (round 5), the projection head of the KEYFRAME-SIDE searches - what orbx_window_point (below) says the caller computes: ORBmatcher::
 * Fuse (reference src/ORBmatcher2.cc:446-520; the Sim3 form :640-688), SearchByProjection(KeyFrame *, Sim3, ..) (src/ORBmatcher1.cc:451-494,
 * :557-600) and SearchByProjection(Frame &, KeyFrame *, ..) (src/ORBmatcher3.cc:486-519).  Per map point: p3Dc = Tcw * p3Dw (q_xyzw / t: the
 * Sophus::SE3f the function holds - for the Sim3 forms SE3f(Scw.rotationMatrix(), Scw.translation() / Scw.scale())), [p3Dc(2) < 0: skip],
 * invz = 1 / p3Dc(2), uv = pCamera->project(p3Dc), KeyFrame::IsInImage (x >= mnMinX && x < mnMaxX ..; src/KeyFrame.cc:750-753) or the
 * Frame form (uv < mnMin || uv > mnMax rejects), ur = uv(0) - bf * invz, PO = p3Dw - Ow, dist = PO.norm(), [dist outside 0.8 mfMinDistance ..
 * 1.2 mfMaxDistance: skip], [PO.dot(Pn) < 0.5 * dist: skip], nPredictedLevel = MapPoint::PredictScale(dist, pKF) (src/MapPoint.cc:514-529,
 * float arithmetic as above), radius = th * mvScaleFactors[nPredictedLevel].  tests: bit 0 the positive-depth test, bit 1 the viewing-angle
 * test (Fuse and the Sim3 search: 3), bit 2 Frame-style bounds (SearchByProjection(Frame, KeyFrame): 4).  points[i].flags bit 0 = the
 * caller's own skips (NULL / isBad() / IsInKeyFrame / already found) passed; points_out[i].search = that && every test passed.  The
 * records feed orbx_search_keyframe_windows / orbx_fuse / orbx_search_by_projection_sim3 / orbx_search_by_projection_keyframe and their
 * _frames forms without leaving the device. */
struct orbx_window_point_s;
int orbx_project_window_points(orbx_matcher *m, const float q_xyzw[4], const float t[3], const float Ow[3], const orbx_camera *cam,
                               const float bounds[4], float bf, float th, const float *scale_factors, float log_scale_factor, int nlevels,
                               int tests, const orbx_world_point *points, int np, struct orbx_window_point_s *points_out);
/* Batch form: d_pose [nframes][16] floats per frame = q x y z w, t x y z, Ow x y z (+ 6 of padding); d_points_out [nframes][max_np]. */
int orbx_project_window_points_frames(orbx_matcher *m, int nframes, const float *d_pose, const orbx_camera *cam, const float bounds[4],
                                      float bf, float th, const float *scale_factors, float log_scale_factor, int nlevels, int tests,
                                      const orbx_world_point *d_points, const int32_t *d_np, int max_np,
                                      struct orbx_window_point_s *d_points_out);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF, Frame &F, std::vector<MapPoint*> &vpMapPointMatches) — reference
 * src/ORBmatcher1.cc:225-427 (decl include/ORBmatcher.h:62), F.Nleft == -1 / pKF->mpCamera2 == NULL; Tracking::TrackReferenceKeyFrame
 * and Relocalization.  Both FeatureVectors (pKF->mFeatVec, F.mFeatVec) as orbx_vocabulary_transform returns them: ascending node
 * ids, fv_start offsets, feature indices.  kf_keys = pKF->mvKeysUn, kf_desc = pKF->mDescriptors, kf_valid[i] = the keyframe's
 * map point i exists and !isBad(); f_keys = F.mvKeys, f_desc = F.mDescriptors.  Inside every shared node each keyframe feature
 * (in order) takes the closest frame feature that has no map point yet (TH_LOW = 50, ratio nnratio = mfNNratio to the second
 * closest), then the rotation-histogram check.  matches[f] = keyframe feature index whose map point F.mvpMapPoints[f] becomes,
 * -1 = NULL.  Returns nmatches (>= 0) or a negative status. */
int orbx_search_by_bow(orbx_matcher *m, const int32_t *kf_fv_node, const int32_t *kf_fv_start, const int32_t *kf_fv_feature, int kf_nfv,
                       const orbx_keypoint *kf_keys, const uint8_t *kf_desc, const uint8_t *kf_valid, int kf_n, const int32_t *f_fv_node,
                       const int32_t *f_fv_start, const int32_t *f_fv_feature, int f_nfv, const orbx_keypoint *f_keys, const uint8_t *f_desc,
                       int f_n, float nnratio, int check_orientation, int32_t *matches);

/* int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, std::vector<MapPoint*> &vpMatches12) — reference
 * src/ORBmatcher2.cc:37-177 (decl include/ORBmatcher.h:63; loop closing / merging), NLeft == -1 on both sides.  As
 * orbx_search_by_bow, with valid2 masking the second keyframe's features (its map point exists and !isBad()), the strict distance
 * test of that overload (bestDist1 < TH_LOW) and the result indexed by the first keyframe's feature: matches12[idx1] = idx2 whose
 * map point vpMatches12[idx1] becomes, -1 = NULL.  Returns nmatches (>= 0) or a negative status. */
int orbx_search_by_bow_keyframes(orbx_matcher *m, const int32_t *fv1_node, const int32_t *fv1_start, const int32_t *fv1_feature, int nfv1,
                                 const orbx_keypoint *keys1, const uint8_t *desc1, const uint8_t *valid1, int n1, const int32_t *fv2_node,
                                 const int32_t *fv2_start, const int32_t *fv2_feature, int nfv2, const orbx_keypoint *keys2,
                                 const uint8_t *desc2, const uint8_t *valid2, int n2, float nnratio, int check_orientation,
                                 int32_t *matches12);

/* What the BoW-guided keyframe searches read of one KeyFrame (reference include/KeyFrame.h): keys = mvKeysUn when NLeft == -1, else
 * mvKeys (the n_left keypoints of the left image) followed by mvKeysRight - the keypoint the reference picks with
 * `(idx < NLeft) ? mvKeys[idx] : mvKeysRight[idx - NLeft]` (src/ORBmatcher2.cc:259-261) is then keys[idx]; desc = mDescriptors;
 * u_right = mvuRight (NULL: no stereo keypoints); has_map_point[i] = (GetMapPoint(i) != NULL); the FeatureVector mFeatVec as
 * orbx_vocabulary_transform returns it; scale_factors / level_sigma2 = mvScaleFactors / mvLevelSigma2. */
typedef struct orbx_keyframe_view {
    const orbx_keypoint *keys;
    const uint8_t *desc;
    int32_t n, n_left;              /* N, NLeft (-1: one camera) */
    const float *u_right;
    const uint8_t *has_map_point;
    const int32_t *fv_node, *fv_start, *fv_feature;
    int32_t nfv;
    const float *scale_factors, *level_sigma2;
    int32_t nlevels;
} orbx_keyframe_view;

/* The two-view geometry ORBmatcher::SearchForTriangulation derives from the keyframes' poses before its loop (src/ORBmatcher2.cc:185-214):
 * the host computes it exactly as the reference does (Sophus / Eigen) and hands it over.
 *   ep      = pKF2->mpCamera->project(T2w * Cw), the epipole in the second image (:189-191);
 *   model 0 = Pinhole::epipolarConstrain (src/CameraModels/Pinhole.cpp:107-129): F12 = K1^-T [t12]x R12 K2^-1 (:109-112), row-major;
 *   model 1 = KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220): TriangulateMatches(...) > 0.0001 with
 *             cam_left / cam_right = mvParameters of mpCamera / mpCamera2 (fx fy cx cy k0..k3), precision, and R / t:
 *             rig == 0: R[0], t[0] = T12 = T1w * Tw2 (:200-202), both keypoints seen by cam_left;
 *             rig != 0 (both keyframes have mpCamera2): R[2 * bRight1 + bRight2], t[...] = Tll, Tlr, Trl, Trr (:205-214), camera of a
 *             keypoint = cam_right iff its index is >= NLeft (:319-351). */
typedef struct orbx_triangulation_geometry {
    float ep[2];
    int32_t model, rig;
    float F12[9];
    float cam_left[8], cam_right[8], precision;
    float R[4][9], t[4][3];
} orbx_triangulation_geometry;

/* int ORBmatcher::SearchForTriangulation(KeyFrame *pKF1, KeyFrame *pKF2, vector<pair<size_t, size_t>> &vMatchedPairs, bool bOnlyStereo,
 * bool bCoarse) — reference src/ORBmatcher2.cc:179-418 (decl include/ORBmatcher.h:79-80; LocalMapping::CreateNewMapPoints).  Inside
 * every vocabulary node the two keyframes share, each feature of pKF1 without a map point takes the feature of pKF2 without a map
 * point at the smallest descriptor distance <= TH_LOW (the LAST one on ties: `dist > bestDist` at :293 is strict) among those that lie
 * further than 10 sqrt(scale) px from the epipole (monocular pairs, :305-313) and satisfy the epipolar constraint (bCoarse skips it),
 * then the rotation-histogram check.  (vbMatched2 is never set in this function: the features of pKF1 do not compete.)
 * matches12[idx1] = idx2 or -1 (kf1->n entries); pairs (optional, 2 * kf1->n ints): vMatchedPairs as (idx1, idx2) in increasing idx1.
 * Returns nmatches (>= 0) or a negative status. */
int orbx_search_for_triangulation(orbx_matcher *m, const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2,
                                  const orbx_triangulation_geometry *geometry, int only_stereo, int coarse, int check_orientation,
                                  int32_t *matches12, int32_t *pairs);

/* One map point projected into a keyframe, as ORBmatcher::Fuse / SearchBySim3 have it when they reach `GetFeaturesInArea`: the pose
 * algebra and the visibility tests in front (Tcw * p3Dw, project, IsInImage, the distance and viewing-angle tests, PredictScale -
 * src/ORBmatcher2.cc:469-520, 645-688, src/ORBmatcher3.cc:82-118) are the caller's Sophus / Eigen code; a point that failed one of them
 * is passed with search = 0.  Since round 5 orbx_project_window_points(_frames) computes these records on the device. */
typedef struct orbx_window_point_s {
    float u, v;       /* uv(0), uv(1) */
    float ur;         /* uv(0) - bf * invz (Fuse with stereo keypoints, :489); unused otherwise */
    float radius;     /* th * mvScaleFactors[nPredictedLevel] */
    int32_t level;    /* nPredictedLevel */
    int32_t search;   /* 0: skipped by the caller's tests */
} orbx_window_point;

/* The search every one of these functions runs per projected map point (src/ORBmatcher2.cc:522-579 and 690-714, src/ORBmatcher3.cc:
 * 120-145 and 188-213): KeyFrame::GetFeaturesInArea(u, v, radius) (src/KeyFrame.cc:704-748: cells ix outer / iy inner, push order inside a
 * cell, |dx| < r && |dy| < r) over the keyframe's 64 x 48 grid on [min_x, max_x) x [min_y, max_y), keypoints of level nPredictedLevel - 1
 * or nPredictedLevel only, with chi2_test != 0 the reprojection test of Fuse (:545-569: e2 * mvInvLevelSigma2[level] > 7.8 with
 * mvuRight >= 0, > 5.99 without), then the FIRST keypoint at the smallest DescriptorDistance to the point's descriptor.
 * keys / desc / u_right: the keyframe's mvKeysUn / mDescriptors / mvuRight - for a rig the side being searched (mvKeys or mvKeysRight and
 * its descriptor rows; add NLeft to the result for the right side as :571 does).  best_idx[i] = that keypoint or -1, best_dist[i] its
 * distance (INT32_MAX when none).  Returns 0 or a negative status.  The points do not interact: one wave each. */
int orbx_search_keyframe_windows(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x,
                                 float max_x, float min_y, float max_y, const float *inv_level_sigma2, int nlevels,
                                 const orbx_window_point *points, const uint8_t *point_desc, int np, int chi2_test, int32_t *best_idx,
                                 int32_t *best_dist);

/* int ORBmatcher::Fuse(KeyFrame *pKF, const vector<MapPoint*> &vpMapPoints, const float th, const bool bRight) — reference
 * src/ORBmatcher2.cc:420-610 (chi2_test = 1) and Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) :612-770 (chi2_test = 0): the search part.
 * fused_idx[i] = keypoint bestIdx with bestDist <= TH_LOW (50) that map point i fuses into, -1 otherwise; what happens then
 * (Replace / AddObservation / AddMapPoint, :581-600) is map bookkeeping and stays with the caller, who walks fused_idx in order.
 * Returns the number of points with fused_idx >= 0 (nFused when no pMPinKF is bad) or a negative status. */
int orbx_fuse(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x, float max_x,
              float min_y, float max_y, const float *inv_level_sigma2, int nlevels, const orbx_window_point *points,
              const uint8_t *point_desc, int np, int chi2_test, int32_t *fused_idx);

/* The same two searches over a BATCH of keyframes, device-resident (round 4): keyframe p = frame p of `ex`'s last batch - its keypoints
 * through Frame::UndistortKeyPoints (k4 / dist / knew4 as in orbx_undistort_keypoints; mvKeysUn is what KeyFrame::GetFeaturesInArea reads),
 * its descriptors where the extractor left them, its 64 x 48 grid built on the device - the shape of LocalMapping::SearchInNeighbors
 * (src/LocalMapping.cc:772-803: the new keyframe's map points fused into its 10 - 30 neighbours, and theirs into it) and of the loop
 * closer's Fuse calls.  d_points [nframes][max_np] / d_point_desc [nframes][max_np][32] / d_np [nframes]: the map points projected into
 * keyframe p (orbx_window_point, as for the single-keyframe calls); d_u_right [nframes][cap] = mvuRight or NULL; inv_level_sigma2 =
 * the extractor's own table (mvInvLevelSigma2).  Outputs, device: d_best_idx / d_best_dist [nframes][max_np] (orbx_search_keyframe_windows)
 * resp. d_fused_idx [nframes][max_np] + d_nfused [nframes] (orbx_fuse).  cap = orbx_max_keypoints(ex).  Three launches for the whole batch,
 * asynchronous on the matcher's stream. */
int orbx_search_keyframe_windows_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                        const float *knew4, float min_x, float max_x, float min_y, float max_y, const float *d_u_right,
                                        const orbx_window_point *d_points, const uint8_t *d_point_desc, const int32_t *d_np, int max_np,
                                        int chi2_test, void *d_best_idx, void *d_best_dist);
int orbx_fuse_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist, const float *knew4,
                     float min_x, float max_x, float min_y, float max_y, const float *d_u_right, const orbx_window_point *d_points,
                     const uint8_t *d_point_desc, const int32_t *d_np, int max_np, int chi2_test, void *d_fused_idx, void *d_nfused);

/* int ORBmatcher::SearchBySim3(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12, const Sim3f &S12, const float th) —
 * reference src/ORBmatcher3.cc:37-254 (loop closing): the map points of pKF1 projected into pKF2 (points_1in2, one per feature of pKF1,
 * search = 0 for a feature without a usable map point, an already matched one (:70-82) or a failed visibility test) search pKF2's
 * windows, those of pKF2 search pKF1's (points_2in1), both with TH_HIGH (100), and a pair is kept when the two searches agree
 * (:236-250).  desc1[i] / desc2[i] = MapPoint::GetDescriptor() of feature i's map point.  matches12[i1] = i2 whose map point
 * vpMatches12[i1] becomes, -1 = left as it was.  Returns nFound or a negative status. */
int orbx_search_by_sim3(orbx_matcher *m, const orbx_keypoint *keys1, const uint8_t *kf_desc1, int n1, const float *bounds1,
                        const orbx_keypoint *keys2, const uint8_t *kf_desc2, int n2, const float *bounds2,
                        const orbx_window_point *points_1in2, const uint8_t *mp_desc1, const orbx_window_point *points_2in1,
                        const uint8_t *mp_desc2, int nlevels, int32_t *matches12);

/* void MapPoint::ComputeDistinctiveDescriptors() — reference src/MapPoint.cc:329-403 (decl include/MapPoint.h), for a batch of map points:
 * the observed descriptors of point p are rows [start[p], start[p + 1]) of `desc` (vDescriptors in the order the reference collects
 * them, :346-362); best[p] = the row (relative to start[p]) with the least median DescriptorDistance to all of them - the median being
 * element (int)(0.5 * (N - 1)) of the sorted distances including the 0 to itself (:386-389), the FIRST such row on ties (:391) - or -1 for
 * a point without observations.  mDescriptor = that row.  LocalMapping / Tracking call it after every change of a point's observations
 * (12 call sites); one wave per map point here.  Returns 0 or a negative status. */
int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *desc, const int32_t *start, int npoints, int32_t *best);

/* int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sim3f &Scw, const vector<MapPoint*> &vpPoints, vector<MapPoint*> &vpMatched, int th,
 * float ratioHamming) — reference src/ORBmatcher1.cc:429-533 (decl include/ORBmatcher.h:59-60), and the overload that also records the
 * points' keyframes (:535-648): the same window search as orbx_search_keyframe_windows (no reprojection test), but a keypoint that
 * already holds a match - matched[idx] != 0 on entry (vpMatched[idx] != NULL), or taken by an EARLIER accepted point of this call - is
 * skipped (:500-501), and a point is accepted when bestDist <= TH_LOW * ratioHamming (:520).  points[i].search = 0 for a point that is
 * bad, already found (spAlreadyFound, :444-453) or fails a visibility test.  new_match[idx] = the point i whose map point
 * vpMatched[idx] (and vpMatchedKF[idx] = vpPointsKFs[i]) becomes, -1 = unchanged.  Returns nmatches or a negative status. */
int orbx_search_by_projection_sim3(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x, float min_y,
                                   float max_y, int nlevels, const orbx_window_point *points, const uint8_t *point_desc, int np,
                                   const uint8_t *matched, float ratio_hamming, int32_t *new_match);

/* int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const set<MapPoint*> &sAlreadyFound, const float th, const int
 * ORBdist) — reference src/ORBmatcher3.cc:469-590 (decl include/ORBmatcher.h:57; Tracking::Relocalization), from GetFeaturesInArea on:
 * keys / desc / has_map_point = CurrentFrame.mvKeysUn / mDescriptors / (mvpMapPoints[i] != NULL); points[i] = the keyframe's map point i
 * projected with the frame's pose (search = 0: NULL, bad, in sAlreadyFound, outside the bounds or the distance range, :485-508),
 * level = nPredictedLevel - the window takes levels nPredictedLevel - 1 .. + 1 (:513); point_angle[i] = pKF->mvKeysUn[i].angle.  A keypoint
 * that holds a map point - on entry or from an earlier accepted point - is skipped (:525-526); accepted: bestDist <= ORBdist (:537); then the
 * rotation histogram when check_orientation != 0 (:544-584).  new_match[i2] = i, the keyframe feature whose map point
 * CurrentFrame.mvpMapPoints[i2] becomes, -1 = unchanged.  Returns nmatches or a negative status. */
int orbx_search_by_projection_keyframe(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x,
                                       float min_y, float max_y, int nlevels, const orbx_window_point *points, const uint8_t *point_desc,
                                       const float *point_angle, int np, const uint8_t *has_map_point, int orb_dist, int check_orientation,
                                       int32_t *new_match);


/* void Frame::UndistortKeyPoints() — reference src/Frame.cc:777-811 (decl include/Frame.h): mvKeysUn from mvKeys.
 * k4 = fx, fy, cx, cy of static_cast<Pinhole*>(mpCamera)->toK(); dist = mDistCoef (k1 k2 p1 p2 [k3 [k4 k5 k6 [s1 s2 s3 s4]]], up to
 * 12 floats); knew4 = fx, fy, cx, cy of mK.  dist[0] == 0 (or ndist == 0) copies, as the reference does; otherwise
 * cv::undistortPoints(mat, mat, K, mDistCoef, cv::Mat(), mK): double arithmetic, five iterations.  out may alias kps. */
int orbx_undistort_keypoints(orbx_matcher *m, const orbx_keypoint *kps, int n, const float *k4, const float *dist, int ndist,
                             const float *knew4, orbx_keypoint *out);

/* void Frame::ComputeImageBounds(const cv::Mat &imLeft) — reference src/Frame.cc:813-845: bounds = mnMinX, mnMaxX, mnMinY, mnMaxY
 * (the undistorted image corners; 0, cols, 0, rows without distortion) — the bounds the grid / search entry points take. */
int orbx_compute_image_bounds(orbx_matcher *m, int cols, int rows, const float *k4, const float *dist, int ndist, const float *knew4,
                              float *bounds);

/* int ORBmatcher::SearchByProjection(Frame &F, const std::vector<MapPoint*> &vpMapPoints, const float th, const bool bFarPoints,
 * const float thFarPoints) — reference src/ORBmatcher1.cc:45-215 (decl include/ORBmatcher.h:48), the F.Nleft == -1 branch, called
 * every frame by Tracking::SearchLocalPoints (src/Tracking4.cc:737).  One orbx_projection per map point with the fields
 * Frame::isInFrustum (src/Frame.cc:527-616) left on it: u, v = mTrackProjX, mTrackProjY; invz = mTrackProjXR; angle = mTrackViewCos;
 * octave = mnTrackScaleLevel; flags bit 0 = mbTrackInView && !isBad() && !(bFarPoints && mTrackDepth > thFarPoints), bit 1 =
 * Observations() > 0.  Window radius = RadiusByViewingCos(viewCos) [* th if th != 1] * scale_factors[level], levels level - 1 ..
 * level, best and second best with the ratio test only when both lie in the same level (nnratio = mfNNratio), TH_HIGH = 100.
 * cur_state / assign as in orbx_search_by_projection.  Returns nmatches (>= 0) or a negative status. */
int orbx_search_by_projection_points(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, const float *u_right,
                                     const int8_t *cur_state, int nc, float min_x, float max_x, float min_y, float max_y,
                                     const float *scale_factors, int nlevels, const orbx_projection *map_points,
                                     const uint8_t *map_point_desc, int np, float th, float nnratio, int32_t *assign);

/* The F.Nleft != -1 branch of the same function (two KB8 fisheye cameras, reference src/ORBmatcher1.cc:45-215): per map point the left
 * search (mbTrackInView; mTrackProjX / Y, mTrackViewCos, mnTrackScaleLevel) and then the right one (mbTrackInViewR; mTrackProjXR / YR,
 * mTrackViewCosR, mnTrackScaleLevelR != -1; radius without the th factor, :147) on mGridRight / mvKeysRight.  A match on one side is
 * also written to the keypoint's stereo partner (left_to_right = F.mvLeftToRightMatch [n_left], right_to_left = F.mvRightToLeftMatch
 * [n_right], -1 = none; :133-137, :198-202) and counted; a left ratio-test failure skips the point's right search (:126-127).
 * kc / dc / cur_state / assign as in orbx_search_by_projection_fisheye.  map_points_left[i] / map_points_right[i]: u, v, angle (viewing
 * cosine), octave (predicted level) of each side; flags bit 0 = mbTrackInView / mbTrackInViewR with the isBad() and far-point tests
 * folded in, bit 1 (read from the left record) = Observations() > 0. */
int orbx_search_by_projection_points_fisheye(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, int n_left, int n_right,
                                             const int8_t *cur_state, float min_x, float max_x, float min_y, float max_y,
                                             const float *scale_factors, int nlevels, const orbx_projection *map_points_left,
                                             const orbx_projection *map_points_right, const uint8_t *map_point_desc, int np,
                                             const int32_t *left_to_right, const int32_t *right_to_left, float th, float nnratio,
                                             int32_t *assign);

/* Frame::ComputeStereoMatches, complete — reference src/Frame.cc:841-1011: the row-band Hamming
 * stage above, then for every accepted left keypoint the 11x11 L1 sliding-window refinement (+-5 px) on
 * mvImagePyramid[octave] of the left / right extractor (the device-resident pyramids of frame `frame_left` /
 * `frame_right` of each handle's last batch: no pyramid leaves the GPU), parabola fit, re-scaling, the disparity test
 * with minD = 0, maxD = mbf / mb, and the median-based outlier rejection.  u_right[nl] / depth[nl] receive mvuRight /
 * mvDepth (-1 where no match).  Keypoints / descriptors are the host arrays the extractors returned. */
int orbx_compute_stereo_matches(orbx_matcher *m, orbx_extractor *left, int frame_left, orbx_extractor *right, int frame_right,
                                const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr, const uint8_t *dr, int nr,
                                float mb, float mbf, float *u_right, float *depth);

/* The same for `npairs` stereo pairs at once, entirely device-resident: pair p = frame p of the left / right extractor's last
 * batch; keypoints, descriptors, counts and pyramids are read where the extractors left them, the three kernels run once over
 * all pairs (blockIdx.y = pair) and nothing crosses PCIe - this is how the stereo Frame constructor (reference src/Frame.cc:143)
 * runs over a batch.  d_u_right / d_depth: device buffers [npairs][orbx_max_keypoints(left)] (float); entries beyond a frame's
 * keypoint count are not written.  Asynchronous on the matcher's stream, ordered against the extractors like
 * orbx_match_knn2_frames. */
int orbx_compute_stereo_matches_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int npairs, float mb, float mbf,
                                       void *d_u_right, void *d_depth);

/* void Frame::ComputeStereoFishEyeMatches() — reference src/Frame.cc:1156-1196 (fisheye stereo Frame constructor, :1078-1153), with
 * KannalaBrandt8::TriangulateMatches (src/CameraModels/KannalaBrandt8.cpp:306-375).  The brute-force knnMatch runs on the lapping-area
 * rows [mono_left, nl) x [mono_right, nr) (kNN-2 above), Lowe's ratio 0.7 selects the good matches, and each is triangulated between
 * the two Kannala-Brandt cameras: parallax limit 0.9998, linear triangulation (null vector of the 4 x 4 system), positive depth in
 * both cameras, reprojection error below 5.991 * sigma2[octave] in both images.
 *   cam1 / cam2   mvParameters of mpCamera / mpCamera2: fx fy cx cy k0 k1 k2 k3;  precision = KannalaBrandt8::precision (1e-6)
 *   R12 / t12     Frame::mRlr (row-major 3 x 3) / mtlr: x_left = R12 x_right + t12
 *   level_sigma2  mvLevelSigma2 (orbx_get_scale_sigma_squares)
 * Outputs: left_to_right[nl] = mvLeftToRightMatch, right_to_left[nr] = mvRightToLeftMatch, depth[nl] = mvDepth (-1: none),
 * p3d[nl][3] = mvStereo3Dpoints (written where depth > 0).  Returns nMatches (>= 0) or a negative status.
 * Floating point: the reference evaluates this with Eigen and libm.  What is the reference's own arithmetic here: every float operation in
 * source order with Eigen's 3-term redux x0 + (x1 + x2), atan2f / tanf as glibc computes them (include/orbx_libm.h, pinned exhaustively).
 * What is TOLERANCE-LEVEL, not bit-equal, against the reference: the null vector of the 4 x 4 system - Eigen::JacobiSVD<Matrix4f> (a float
 * two-sided Jacobi sweep) is restated as a double-precision one-sided Jacobi iteration (oracle/orb_oracle.cpp), so depth[] and p3d[] agree
 * with the reference's to a few float ulp times the conditioning of the triangulation, and the match decisions taken from them (positive
 * depth, reprojection error against 5.991 sigma2) agree away from their thresholds.  Against the CPU oracle all outputs are bit-equal. */
int orbx_compute_stereo_fisheye_matches(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int nl, int mono_left,
                                        const orbx_keypoint *kr, const uint8_t *dr, int nr, int mono_right, const float *cam1,
                                        const float *cam2, float precision, const float *R12, const float *t12,
                                        const float *level_sigma2, int nlevels, int32_t *left_to_right, int32_t *right_to_left,
                                        float *depth, float *p3d);
/* The same for `npairs` stereo pairs in three launches on the two extractors' device-resident outputs (keypoints, descriptors, counts
 * and monoIndex values of frame p of each handle's last batch).  Device outputs: d_left_to_right / d_depth [npairs][cap_left],
 * d_right_to_left [npairs][cap_right], d_p3d [npairs][cap_left][3], d_nmatches [npairs]; cap = orbx_max_keypoints.  Asynchronous. */
int orbx_compute_stereo_fisheye_matches_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int npairs, const float *cam1,
                                               const float *cam2, float precision, const float *R12, const float *t12,
                                               void *d_left_to_right, void *d_right_to_left, void *d_depth, void *d_p3d, void *d_nmatches);

/* ---- utilities ---------------------------------------------------------------------------------- */

/* Number of HIP devices (reference: Eden_resources::get_gpus_count, src/HipEdenResources/Ngpus_Ncpus.cpp). */
int orbx_device_count(void);
/* Deterministic synthetic frame (include/orbx_synth.h): host buffer. */
void orbx_synth_frame(uint64_t frame_index, int w, int h, uint8_t *out, ptrdiff_t stride);
/* Per-level geometry for a w x h image: sizes out[level] (reference src/ORBextractor.cc:1313). */
int orbx_level_sizes(const orbx_extractor *ex, int w, int h, int32_t *lw, int32_t *lh);

/* ---- DBoW2 vocabulary ---------------------------------------------------------------------------------------
 * ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (reference include/ORBVocabulary.h, vendored
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h): the k-ary tree lives in HBM (child descriptors contiguous per node) and
 * Frame::ComputeBoW / KeyFrame::ComputeBoW (src/Frame.cc:768-775) become one call. */
typedef struct orbx_vocabulary orbx_vocabulary;

/* The tree as TemplatedVocabulary::loadFromTextFile builds it (TemplatedVocabulary.h:1338-1423): node 0 = root, parent[i] < i, one
 * 32-byte descriptor and one weight per node, leaves numbered as words in node order; scoring / weighting = the enums of
 * BowVector.h:39-56 (L1_NORM 0, L2_NORM 1, CHI_SQUARE 2, KL 3, BHATTACHARYYA 4, DOT_PRODUCT 5; TF_IDF 0, TF 1, IDF 2, BINARY 3). */
int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent, const uint8_t *is_leaf,
                           const uint8_t *desc, const double *weight, orbx_vocabulary **out);
/* bool loadFromTextFile(const std::string &filename) — the ORBvoc.txt format ("k L scoring weighting", then "parent is_leaf d0..d31
 * weight" per node). */
int orbx_vocabulary_load_text(int device, const char *path, orbx_vocabulary **out);
void orbx_vocabulary_destroy(orbx_vocabulary *v);
int orbx_vocabulary_size(const orbx_vocabulary *v); /* unsigned int size() const: number of words */

/* void transform(const std::vector<TDescriptor> &features, BowVector &v, FeatureVector &fv, int levelsup) const —
 * TemplatedVocabulary.h:1127-1196 with the descent of :1218-1261 (first child with the strictly smallest FORB::distance).
 * desc: n x 32 bytes (n <= 16384).  BowVector = (bow_word[i], bow_value[i]), i < *n_bow, ascending word id, weighted and
 * normalised as the vocabulary's weighting / scoring prescribe (doubles, bit-identical to the std::map arithmetic).
 * FeatureVector = node fv_node[j], j < *n_fv (ascending), holding features fv_feature[fv_start[j] .. fv_start[j + 1]) in
 * increasing order.  feat_word / feat_node (optional): the word and the node `levelsup` levels up of every feature, -1 for a
 * stopped word (weight <= 0).  Capacities: n entries each, fv_start n + 1.  Returns *n_bow or a negative status. */
int orbx_vocabulary_transform(orbx_vocabulary *v, const uint8_t *desc, int n, int levelsup, int32_t *feat_word, int32_t *feat_node,
                              int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start, int32_t *fv_feature,
                              int32_t *n_fv);

/* Frame::ComputeBoW on the extractor's own outputs: the descriptors of frame `frame` of the extractor's last batch are read where
 * they lie in HBM (no host round trip; the extractor's pending work is awaited).  Capacities: orbx_max_keypoints(ex) entries
 * (fv_start one more).  Outputs and return value as orbx_vocabulary_transform. */
int orbx_vocabulary_transform_frame(orbx_vocabulary *v, orbx_extractor *ex, int frame, int levelsup, int32_t *feat_word, int32_t *feat_node,
                                    int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start,
                                    int32_t *fv_feature, int32_t *n_fv);

/* The same for the first `nframes` frames of the extractor's last batch in two launches (throughput use: a batch of frames goes from
 * pixels to BowVectors without leaving HBM in between).  cap = orbx_max_keypoints(ex); frame f: BowVector (bow_word, bow_value)
 * [f * cap .. + n_bow[f]), FeatureVector nodes fv_node[f * cap .. + n_fv[f]), offsets fv_start[f * (cap + 1) ..], features
 * fv_feature[f * cap + offset]. */
int orbx_vocabulary_transform_batch(orbx_vocabulary *v, orbx_extractor *ex, int nframes, int levelsup, int32_t *bow_word, double *bow_value,
                                    int32_t *n_bow, int32_t *fv_node, int32_t *fv_start, int32_t *fv_feature, int32_t *n_fv);

/* ORBmatcher::SearchByBoW for a batch of pairs, device-resident (round 4): Tracking::Relocalization's loop over the candidate keyframes
 * (reference src/Tracking5.cc:83, SearchByBoW(pKF, mCurrentFrame, vvpMapPointMatches[i]): keyframe_pair = 0, semantics of orbx_search_by_bow)
 * and LoopClosing's loop over the covisible keyframes of a candidate (src/LoopClosing1.cc:662, SearchByBoW(mpCurrentKF, vpCovKFi[j], ..):
 * keyframe_pair = 1, semantics of orbx_search_by_bow_keyframes).  Both sides of pair k are frames of `ex`'s last batch - kf_frames[k] the
 * keyframe (first keyframe), f_frames[k] the frame (second keyframe) - with the keypoints and descriptors where the extractor left them and
 * the FeatureVectors of the last orbx_vocabulary_transform_batch(voc, ex, ..) call, which must have covered them.  d_valid [nframes][cap]
 * bytes on the device: the map point of keypoint i of frame f exists and is not bad (read for the keyframe side; for both sides when
 * keyframe_pair = 1); cap = orbx_max_keypoints(ex) <= 65535.  Outputs, device: d_matches [npairs][cap] int32 - keyframe_pair = 0: by
 * frame feature, the keyframe feature whose map point it takes; 1: by feature of the first keyframe, the feature of the second - entries
 * beyond the frame's keypoint count stay -1; d_nmatches [npairs].  Two launches for all pairs, asynchronous on the matcher's stream. */
int orbx_search_by_bow_frames(orbx_matcher *m, orbx_vocabulary *voc, orbx_extractor *ex, const int32_t *kf_frames, const int32_t *f_frames,
                              int npairs, int keyframe_pair, const uint8_t *d_valid, float nnratio, int check_orientation, void *d_matches,
                              void *d_nmatches);

/* The same search for ONE keyframe against a list of neighbours, device-resident (round 4): LocalMapping::CreateNewMapPoints' matcher call
 * (reference src/LocalMapping.cc:466, inside the loop over vpNeighKFs).  Keyframes = frames of `ex`'s last batch (single camera, NLeft == -1);
 * their mvKeysUn come from Frame::UndistortKeyPoints on the device (k4 / dist / knew4 as in orbx_undistort_keypoints), their descriptors lie
 * where the extractor left them, and their FeatureVectors are the DEVICE results of the last orbx_vocabulary_transform_batch(voc, ex, ..)
 * call - which must have covered these frames.  ref_frame = pKF1, neighbours[k] = pKF2 of pair k, geometry[k] (host) its two-view geometry
 * as for the single call; d_has_map_point [nframes][cap] bytes (GetMapPoint(i) != NULL), d_u_right [nframes][cap] or NULL, cap =
 * orbx_max_keypoints(ex).  Outputs, device: d_matches12 [nneigh][cap] (entries >= the keyframe's keypoint count stay -1), d_nmatches
 * [nneigh].  Two launches for all neighbours on the matcher's stream; nothing of the keyframes crosses the link. */
int orbx_search_for_triangulation_frames(orbx_matcher *m, orbx_vocabulary *voc, orbx_extractor *ex, int ref_frame, const int32_t *neighbours,
                                         int nneigh, const float *k4, const float *dist, int ndist, const float *knew4,
                                         const orbx_triangulation_geometry *geometry, const uint8_t *d_has_map_point, const float *d_u_right,
                                         int only_stereo, int coarse, int check_orientation, void *d_matches12, void *d_nmatches);


#ifdef __cplusplus
}
#endif
#endif /* ORBX_H */
