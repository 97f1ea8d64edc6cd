// orbx_api.cpp — C ABI (include/orbx.h) and host orchestration of the gfx950 ORB front-end.
//
// Host side of ORB_SLAM3::ORBextractor (reference include/ORBextractor.h:52-113, src/ORBextractor.cc:410-468,
// 1227-1329): constructor tables, per-image-size geometry (level sizes, FAST cell grid, cv::resize coefficient
// tables), device buffer ownership and the launch sequence.  All pixel work is in orbx_kernels.hip; there is
// no CPU fallback — every compute entry point fails if HIP is unavailable.
#include "../../include/orbx.h"
#include "../../include/orbx_synth.h"
#include "orbx_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

// ---------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------
static thread_local std::string g_lastError;
// Setup work - allocations, synchronous uploads, handle creation / destruction, the capture + instantiation of a launch graph - is
// serialised over all handles of the process: a synchronising call must not fall into another thread's open stream capture (it
// invalidates the capture, and the runtime does not survive the two racing), and it is rare (first call, new image size, new source
// buffer).  Steady-state calls - graph replay or kernel launches on the handle's own streams, asynchronous downloads - take no lock.
static std::recursive_mutex g_setupMutex;
#define ORBX_SETUP_LOCK std::lock_guard<std::recursive_mutex> orbxSetupLock__(g_setupMutex)

static int setError(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_lastError = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return setError(ORBX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *orbx_last_error(void) { return g_lastError.c_str(); }

// Every entry point runs with the calling thread's stream-capture mode set to "relaxed".  A handle for single frames captures its
// launch sequence as a hipGraph; while ANY capture is open in the process, a thread in the default (global) mode may not make a
// synchronising call - an allocation, a synchronous copy - even on unrelated streams, so a second handle used from another thread (the
// reference extracts the left and the right image in two threads, src/Frame.cc:124-127) failed with "operation would make the legacy
// stream depend on a capturing blocking stream" and invalidated the first thread's capture.  All streams of this library are
// non-blocking and never depend on the legacy stream, which is what makes the relaxed mode safe here.
struct RelaxedCaptureMode {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    bool ok;
    RelaxedCaptureMode() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; }
    ~RelaxedCaptureMode() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&mode); }
    RelaxedCaptureMode(const RelaxedCaptureMode &) = delete;
    RelaxedCaptureMode &operator=(const RelaxedCaptureMode &) = delete;
};
#define ORBX_API_ENTRY RelaxedCaptureMode orbxCaptureModeGuard__

// OpenCV scalar helpers
static inline int cvRoundF(float v) { return (int)lrintf(v); }
static inline int cvRoundD(double v) { return (int)lrint(v); }
static inline int cvFloorF(float v) { int i = (int)v; return i - (i > v); }
static inline int cvCeilD(double v) { int i = (int)v; return i + (i < v); }
static inline short satShort(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }
static inline size_t roundUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
static std::atomic<unsigned long long> gBatchSerial{0};   // see orbx_extractor::batchSerial

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int ensure(size_t count)
    {
        if (count <= n) return ORBX_OK;
        ORBX_SETUP_LOCK;
        if (p) { (void)hipFree(p); p = nullptr; n = 0; }
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T) + 256);
        if (e != hipSuccess) return setError(ORBX_ERR_OUT_OF_MEMORY, "hipMalloc(%zu bytes): %s", count * sizeof(T), hipGetErrorString(e));
        n = count;
        return ORBX_OK;
    }
    void release() { if (p) { ORBX_SETUP_LOCK; (void)hipFree(p); } p = nullptr; n = 0; }
};

#define ORBX_GRAPH_MAX_FRAMES 16  // "small batch": schedule, tile width and download path for a front end that feeds single frames

struct orbx_extractor {
    orbx_params prm;
    // constructor tables (reference src/ORBextractor.cc:418-467)
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> quota, umax, levelCap;
    int maxKeypoints = 0, maxQuota = 0;
    int taps[7];
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // blur stream
    hipStream_t stream3 = nullptr;          // back stream: quadtree -> pack -> orientation + descriptors
    hipEvent_t evJoin = nullptr;
    hipEvent_t evFront[8] = {}, evBlur[8] = {};  // per sub-batch hand-offs between the streams
    hipStream_t streamCopy = nullptr;       // host-pointer API: H2D of sub-batch i + 1 under the kernels of sub-batch i
    hipStream_t streamOut = nullptr;        // host-pointer API: D2H of sub-batch i's results under the kernels / H2D of the following ones
    hipEvent_t evIn = nullptr;
    // What ONE host batch in flight owns (two slots: orbx_extract_batch_submit keeps two batches in flight on one handle): the events of
    // its sub-batches, the page-locked staging of its results, its device staging buffer (slot 1's: allocated by the first second submit)
    struct PipeSlot {
        hipEvent_t evCopy[8] = {}, evBack[8] = {}, evOut[8] = {};
        hipEvent_t evCopy0[8] = {};             // ORBX_HOST_TIMELINE: start of sub-batch i's input copy
        uint8_t *h_stage = nullptr; size_t h_stageBytes = 0;  // page-locked staging of a whole batch's results (counts, keypoints, descriptors)
        DevBuf<uint8_t> d_in1;                  // slot 1's staging for the input frames (slot 0 uses d_in)
    } slots[2];
    int curSlot = 0;                        // the slot of the host batch runBatch is queueing
    hipEvent_t evLevel[ORBX_MAX_LEVELS] = {};    // level l of the pyramid is complete (recorded after pyr_fast(l-1))

    // geometry of the current image size
    int curW = 0, curH = 0;
    std::vector<OrbxLevelGeom> geom;
    long long candFrameStride = 0;
    int tilesPerFrame = 0, selFrameStride = 0;
    std::vector<int> tabOffsets;  // per level transition: offsets into d_tabs for xofs,xab,yofs,yab,rowBeg,colBeg
    std::vector<int> tileOffsets; // per level: offset (ints) of the OrbxPfTile records of k_pyr_fast in d_tabs
    int chainOffset = 0, chainTiles = 0, chainBufBytes = 0, chainTabBytes = 0;  // k_pyr_chain (small batches): tile rectangles in d_tabs, tiles per frame, LDS per buffer / for the table slices

    // device memory
    DevBuf<uint8_t> d_in;                       // staging for host-pointer input (level 0)
    DevBuf<uint8_t> d_inPacked;                 // host frames with packed rows land here first (see stageHostFrames)
    int inPitch = 0; long long inFrameStride = 0;
    DevBuf<uint8_t> d_qtNodes;                   // quadtree node arrays when a level quota does not fit in LDS (ensureBuffers)
    std::vector<DevBuf<uint8_t>> d_raw, d_blur; // per level [max_batch][h][pitch]
    DevBuf<int> d_tabs;
    DevBuf<OrbxLevelGeom> d_geom;
    DevBuf<uint32_t> d_cand, d_scratchA, d_scratchB, d_sel, d_workKey;
    DevBuf<int> d_tileCount, d_selCount, d_workPos, d_n, d_mono;
    DevBuf<orbx_keypoint> d_kps;
    DevBuf<uint8_t> d_desc;

    // stage profiling: ring of event sets recorded on `stream`
    bool profiling = false;
    std::vector<hipEvent_t> events;  // [PROF_RING][sub-batch][stage][begin, end]
    std::vector<int> profSubs;       // sub-batches recorded in each ring slot
    int profCount = 0;

    // small batches (a SLAM front end calls operator() one frame at a time): the launch sequence as a captured hipGraph
    struct GraphKey { const void *src; int nframes, w, h; long long stride, frameStride; int lap0, lap1, generation; };
    // a few captured sequences, least recently used replaced: a caller that alternates two or three source buffers (a camera ring) or
    // lapping areas must not re-capture on every call (capture + instantiate run under the process-wide setup lock)
    static const int kGraphSlots = 4;        // graph slot k owns the per-level argument block k (pfArgsKey[k]): a captured sequence has
                                             // the block's address baked in, so the block may only change with the graph of its slot
    hipGraphExec_t graphExec[kGraphSlots] = {};
    GraphKey graphKey[kGraphSlots] = {};
    unsigned long long graphUse[kGraphSlots] = {}, graphClock = 0;
    int generation = 0;                      // bumped whenever the geometry or a device buffer changes
    hipStream_t qtStream[ORBX_MAX_LEVELS] = {};  // one branch per level for the quadtree instances of a small batch
    hipEvent_t evQt[ORBX_MAX_LEVELS] = {};
    hipEvent_t evPart[4] = {}, evStag[4] = {};   // front parts of a large batch (ORBX_FRONT_PARTS): end of a part's chain / of its level 0
    uint8_t *h_out = nullptr; size_t h_outBytes = 0;  // pinned staging for the single-frame download
    DevBuf<uint8_t> d_outPack;                        // ... and the device side of it: counts, keypoints, descriptors packed by one launch
    // per-level argument blocks of k_pyr_fast<.., ML> (small batches: FAST of all levels in one launch), a few source buffers cached
    struct LevelArgsKey { const void *src; int nframes; long long stride, frameStride; int generation, iniTh, minTh; };
    DevBuf<uint8_t> d_pfArgs;
    LevelArgsKey pfArgsKey[2 * kGraphSlots] = {};  // [0, kGraphSlots): the graphs' blocks; the rest: calls without a graph, round robin
    int pfArgsNext = 0;
    const void *curLevelArgs = nullptr;      // the slot the next issueSmallBatch uses

    // level-0 source of the last batch (for orbx_get_pyramid_level)
    const uint8_t *lastSrc = nullptr; int lastSrcPitch = 0; long long lastSrcFrameStride = 0; int lastFrames = 0;
    struct PendingBatch *pending[2] = {nullptr, nullptr};   // batches submitted with orbx_extract_batch_submit and not yet waited for, oldest first
    int nPending = 0;
    unsigned long long batchSerial = 0;   // unique over all handles of the process: names the batch whose outputs the device buffers hold (stale-consumer checks)
};

// ---------------------------------------------------------------------------------------------------------
// construction
// ---------------------------------------------------------------------------------------------------------
extern "C" void orbx_default_params(orbx_params *p)
{
    ORBX_API_ENTRY;
    if (!p) return;
    p->nfeatures = 1000; p->scale_factor = 1.2f; p->nlevels = 8; p->ini_th_fast = 20; p->min_th_fast = 7;
    p->device = 0; p->max_width = 640; p->max_height = 480; p->max_batch = 1;
    p->blur_variant = ORBX_BLUR_OPENCV_450; p->trig_mode = ORBX_TRIG_DOUBLE;
}

extern "C" int orbx_device_count(void)
{
    ORBX_API_ENTRY;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// blurred levels are stored as 16 x 8 tiles of 128 B (orbx_kernels.hip: orbx_blur_tiled_offset)
static inline int blurTilesX(const OrbxLevelGeom &g) { return (g.w + 15) / 16; }
static inline long long blurFrameBytes(const OrbxLevelGeom &g) { return (long long)blurTilesX(g) * ((g.h + 7) / 8) * 128; }

static int computeGeometry(orbx_extractor *ex, int w, int h);
static int ensureBuffers(orbx_extractor *ex);

static bool hostTimeline()
{
    static const bool v = [] { const char *e = getenv("ORBX_HOST_TIMELINE"); return e && atoi(e) != 0; }();
    return v;
}

// The ORBX_* environment switches exist for A / B measurements and so that the tests can force every kernel variant; none
// is needed to run the library.  A name that no source reads is reported at the create calls: a mistyped switch that silently runs the
// default would make an A / B measurement - or a test of a variant - meaningless.  In production an ORBX_* variable of some other tool
// (ORBX_HOME ..) must not take the path down, so the default is ONE warning on stderr; with ORBX_STRICT_ENV=1 - set by the tests and the
// A / B scripts - the create call fails instead.  (tests/test_abi.py keeps this list equal to the getenv() calls of the sources; ORBX_LIB /
// ORBX_BENCH_BACKEND / ORBX_COMMIT belong to the Python side and the tooling.)
static const char *const kKnownSwitches[] = {
    "ORBX_BLUR_KERNEL", "ORBX_BLUR_SEG", "ORBX_BLUR_WAVES", "ORBX_CHAIN_TILE", "ORBX_DEBUG", "ORBX_EARLY_BLUR", "ORBX_FRONT_PARTS", "ORBX_FRONT_STAGGER",
    "ORBX_GRAPH", "ORBX_HOST_FEEDER", "ORBX_HOST_SUBBATCHES", "ORBX_HOST_TIMELINE", "ORBX_KNN_MFMA", "ORBX_KNN_PIPELINE", "ORBX_KNN_SHARED_BOUND", "ORBX_KNN_SPLIT", "ORBX_NO_BLUR",
    "ORBX_PF_CLCAP", "ORBX_PF_GQCAP", "ORBX_PF_KERNEL", "ORBX_PF_LANE_ALLOC", "ORBX_PF_LDS_FIT", "ORBX_PF_LDS_PAD", "ORBX_PF_PHASES", "ORBX_PROJECTION_HOST_ROUNDS", "ORBX_QT_KCH",
    "ORBX_QT_KERNEL", "ORBX_QT_LDS_KEYS", "ORBX_QT_MAX_KEYS", "ORBX_QT_MAX_KEYS2", "ORBX_QT_MW", "ORBX_QT_MW_LDS_KEYS", "ORBX_QT_PER_LEVEL", "ORBX_QT_POOL_SLACK",
    "ORBX_QT_STOP", "ORBX_QT_STREAMS", "ORBX_SBP_LIST_LDS", "ORBX_SBP_RESOLVE_GLOBAL", "ORBX_SEARCH_INIT_GENERIC", "ORBX_SI_BLOCK", "ORBX_SI_ROUNDS", "ORBX_SI_ROUNDS_CAP", "ORBX_SI_TOPK",
    "ORBX_SMALL_BLUR_LEVELS", "ORBX_SMALL_PYR_CHAIN", "ORBX_SMALL_QT_BRANCHES", "ORBX_SMALL_SPLIT", "ORBX_STREAM_PRIO", "ORBX_SUBBATCHES", "ORBX_TILE_W",
    "ORBX_STRICT_ENV", "ORBX_LIB", "ORBX_BENCH_BACKEND", "ORBX_COMMIT",
};
extern char **environ;
static int checkEnvironmentSwitches(const char *fn)
{
    for (char **e = environ; e && *e; ++e) {
        if (strncmp(*e, "ORBX_", 5) != 0) continue;
        const char *eq = strchr(*e, '=');
        const size_t len = eq ? (size_t)(eq - *e) : strlen(*e);
        bool known = false;
        for (const char *k : kKnownSwitches)
            if (strlen(k) == len && strncmp(k, *e, len) == 0) { known = true; break; }
        if (known) continue;
        const char *strict = getenv("ORBX_STRICT_ENV");
        if (strict && atoi(strict) != 0)
            return setError(ORBX_ERR_INVALID_ARG, "%s: unknown environment switch %.*s (see DESIGN.md, appendix \"environment switches\")", fn, (int)len, *e);
        static std::atomic<bool> warned{false};
        if (!warned.exchange(true))
            fprintf(stderr, "liborbx: warning: unknown environment switch %.*s is ignored (ORBX_STRICT_ENV=1 makes this an error; DESIGN.md, appendix \"environment switches\")\n", (int)len, *e);
    }
    return ORBX_OK;
}
extern "C" int orbx_known_switches(const char **names, int capacity)
{
    const int n = (int)(sizeof kKnownSwitches / sizeof kKnownSwitches[0]);
    for (int i = 0; i < n && i < capacity; ++i) names[i] = kKnownSwitches[i];
    return n;
}

extern "C" int orbx_create(const orbx_params *p, orbx_extractor **out)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!p || !out) return setError(ORBX_ERR_INVALID_ARG, "orbx_create: null argument");
    *out = nullptr;
    { const int rcEnv = checkEnvironmentSwitches("orbx_create"); if (rcEnv != ORBX_OK) return rcEnv; }
    if (p->nfeatures <= 0 || p->nlevels <= 0 || p->nlevels > ORBX_MAX_LEVELS || !(p->scale_factor > 1.0f))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_create: nfeatures>0, 1<=nlevels<=%d, scale_factor>1 required", ORBX_MAX_LEVELS);
    if (p->trig_mode != ORBX_TRIG_DOUBLE && p->trig_mode != ORBX_TRIG_FLOAT) return setError(ORBX_ERR_INVALID_ARG, "orbx_create: trig_mode must be 0 or 1");
    if (p->blur_variant != ORBX_BLUR_OPENCV_450 && p->blur_variant != ORBX_BLUR_OPENCV_451) return setError(ORBX_ERR_INVALID_ARG, "orbx_create: blur_variant must be 0 or 1");
    if (p->max_width <= 0 || p->max_height <= 0 || p->max_batch <= 0 || p->max_width > 4096 || p->max_height > 4096)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_create: image size must be within 1..4096, max_batch >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return setError(ORBX_ERR_NO_DEVICE, "orbx_create: no HIP device available (this library has no CPU fallback)");
    if (p->device < 0 || p->device >= ndev) return setError(ORBX_ERR_INVALID_ARG, "orbx_create: device %d out of range (%d devices)", p->device, ndev);
    HIP_TRY(hipSetDevice(p->device));

    orbx_extractor *ex = new (std::nothrow) orbx_extractor();
    if (!ex) return setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_create: host allocation failed");
    ex->prm = *p;
    const int nl = p->nlevels;
    // scale tables — reference src/ORBextractor.cc:418-432 (float arithmetic)
    ex->scale.resize(nl); ex->sigma2.resize(nl); ex->invScale.resize(nl); ex->invSigma2.resize(nl);
    ex->scale[0] = 1.0f; ex->sigma2[0] = 1.0f;
    for (int i = 1; i < nl; ++i) {
        ex->scale[i] = ex->scale[i - 1] * p->scale_factor;
        ex->sigma2[i] = ex->scale[i] * ex->scale[i];
    }
    for (int i = 0; i < nl; ++i) {
        ex->invScale[i] = 1.0f / ex->scale[i];
        ex->invSigma2[i] = 1.0f / ex->sigma2[i];
    }
    // per-level quotas — :436-447
    ex->quota.resize(nl);
    float factor = 1.0f / p->scale_factor;
    float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
    int sum = 0;
    for (int level = 0; level < nl - 1; ++level) {
        ex->quota[level] = cvRoundF(nDesired);
        sum += ex->quota[level];
        nDesired *= factor;
    }
    ex->quota[nl - 1] = std::max(p->nfeatures - sum, 0);
    // umax — :455-467
    ex->umax.assign(16, 0);
    {
        int v, v0, vmax = cvFloorF(ORBX_HALF_PATCH * std::sqrt(2.f) / 2 + 1);
        int vmin = cvCeilD(ORBX_HALF_PATCH * std::sqrt(2.f) / 2);
        const double hp2 = ORBX_HALF_PATCH * ORBX_HALF_PATCH;
        for (v = 0; v <= vmax; ++v) ex->umax[v] = cvRoundD(std::sqrt(hp2 - v * v));
        for (v = ORBX_HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (ex->umax[v0] == ex->umax[v0 + 1]) ++v0;
            ex->umax[v] = v0;
            ++v0;
        }
    }
    // DistributeOctTree may overshoot a quota by up to 3 (:697), and its first pass splits each of the nIni = round(W / H) root nodes
    // (:589) into up to 4 before any quota is looked at: a level returns up to max(N + 3, 4 * nIni) keypoints.  nIni is largest for
    // the widest, flattest image the handle accepts (width max_width, FAST region one 35-px cell high).
    ex->levelCap.resize(nl);
    ex->maxKeypoints = 0; ex->maxQuota = 0;
    for (int l = 0; l < nl; ++l) {
        const int wl = l == 0 ? p->max_width : cvRoundF((float)p->max_width * ex->invScale[l]);
        const int nIniMax = (int)std::round((float)std::max(wl - 2 * ORBX_BORDER, 1) / 35.0f);
        ex->levelCap[l] = std::max(std::max(ex->quota[l] + 3, 16), 4 * nIniMax);
        ex->maxKeypoints += ex->levelCap[l];
        ex->maxQuota = std::max(ex->maxQuota, ex->quota[l]);
    }
    static const int t450[7] = {18, 34, 49, 55, 49, 34, 18}, t451[7] = {18, 34, 48, 56, 48, 34, 18};
    memcpy(ex->taps, p->blur_variant == ORBX_BLUR_OPENCV_451 ? t451 : t450, sizeof ex->taps);
    if (getenv("ORBX_DEBUG"))  // the two OpenCV / libm dependent choices an integrator pins with tests/tools/validate_against_opencv.py
        fprintf(stderr, "[orbx] orbx_create: blur_variant %d (%s Gaussian taps %d %d %d %d), trig_mode %d (%s)\n", p->blur_variant,
                p->blur_variant == ORBX_BLUR_OPENCV_451 ? "OpenCV >= 4.5.1" : "OpenCV <= 4.5.0", ex->taps[0], ex->taps[1], ex->taps[2], ex->taps[3],
                p->trig_mode, p->trig_mode == ORBX_TRIG_FLOAT ? "cosf / sinf" : "cos / sin (double) narrowed to float");

    hipError_t e = hipStreamCreateWithFlags(&ex->stream, hipStreamNonBlocking);
    {
        // blur stream above / back (quadtree) stream below the front stream: the quadtree's single-wave, LDS-heavy
        // workgroups are long-lived, so the throughput kernels should get free CU resources first
        int prLo = 0, prHi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prLo, &prHi);  // lo = numerically largest = lowest priority
        static const int mode = [] { const char *q = getenv("ORBX_STREAM_PRIO"); return q ? atoi(q) : 0; }();  // measured: no gain (6.03 vs 5.98 ms)
        if (e == hipSuccess) e = mode ? hipStreamCreateWithPriority(&ex->stream2, hipStreamNonBlocking, prHi) : hipStreamCreateWithFlags(&ex->stream2, hipStreamNonBlocking);
        if (e == hipSuccess) e = mode ? hipStreamCreateWithPriority(&ex->stream3, hipStreamNonBlocking, prLo) : hipStreamCreateWithFlags(&ex->stream3, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->evJoin, hipEventDisableTiming);
    // ORBX_HOST_TIMELINE=1 (debug): the events of the host-pointer pipeline keep time stamps and every orbx_extract_batch call prints
    // when each sub-batch's copy in, kernels and copy out started / ended (stderr; tools/time_host_batch.py --timeline)
    const unsigned pipeFlags = hostTimeline() ? hipEventDefault : hipEventDisableTiming;
    for (int i = 0; i < 8 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&ex->evFront[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->evBlur[i], hipEventDisableTiming);
        for (auto &sl : ex->slots) {
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.evCopy[i], pipeFlags);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.evCopy0[i], pipeFlags);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.evBack[i], pipeFlags);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.evOut[i], pipeFlags);
        }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ex->evIn, pipeFlags);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ex->streamCopy, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ex->streamOut, hipStreamNonBlocking);
    for (int i = 0; i < ORBX_MAX_LEVELS && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ex->evLevel[i], hipEventDisableTiming);
    if (e != hipSuccess) { orbx_destroy(ex); return setError(ORBX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    ex->d_raw.resize(nl); ex->d_blur.resize(nl);
    if ((e = orbxInitDevice()) != hipSuccess) { orbx_destroy(ex); return setError(ORBX_ERR_HIP, "device initialisation: %s", hipGetErrorString(e)); }
    int rc = computeGeometry(ex, p->max_width, p->max_height);
    if (rc == ORBX_OK) rc = ensureBuffers(ex);
    if (rc != ORBX_OK) { orbx_destroy(ex); return rc; }
    *out = ex;
    return ORBX_OK;
}

static void dropPending(orbx_extractor *ex);
extern "C" void orbx_destroy(orbx_extractor *ex)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!ex) return;
    (void)hipSetDevice(ex->prm.device);
    dropPending(ex);   // a submitted batch nobody waited for: nothing of it may still copy into the caller's arrays
    if (ex->streamCopy) { (void)hipStreamSynchronize(ex->streamCopy); (void)hipStreamDestroy(ex->streamCopy); }
    if (ex->streamOut) { (void)hipStreamSynchronize(ex->streamOut); (void)hipStreamDestroy(ex->streamOut); }
    if (ex->stream3) { (void)hipStreamSynchronize(ex->stream3); (void)hipStreamDestroy(ex->stream3); }
    if (ex->stream2) { (void)hipStreamSynchronize(ex->stream2); (void)hipStreamDestroy(ex->stream2); }
    if (ex->stream) { (void)hipStreamSynchronize(ex->stream); (void)hipStreamDestroy(ex->stream); }
    if (ex->evJoin) (void)hipEventDestroy(ex->evJoin);
    for (int i = 0; i < 8; ++i) {
        if (ex->evFront[i]) (void)hipEventDestroy(ex->evFront[i]);
        if (ex->evBlur[i]) (void)hipEventDestroy(ex->evBlur[i]);
        for (auto &sl : ex->slots) {
            if (sl.evCopy[i]) (void)hipEventDestroy(sl.evCopy[i]);
            if (sl.evCopy0[i]) (void)hipEventDestroy(sl.evCopy0[i]);
            if (sl.evBack[i]) (void)hipEventDestroy(sl.evBack[i]);
            if (sl.evOut[i]) (void)hipEventDestroy(sl.evOut[i]);
        }
    }
    if (ex->evIn) (void)hipEventDestroy(ex->evIn);
    for (auto &sl : ex->slots) { if (sl.h_stage) (void)hipHostFree(sl.h_stage); sl.d_in1.release(); }
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i)
        if (ex->evLevel[i]) (void)hipEventDestroy(ex->evLevel[i]);
    for (auto &e : ex->events) (void)hipEventDestroy(e);
    for (int k = 0; k < orbx_extractor::kGraphSlots; ++k)
        if (ex->graphExec[k]) (void)hipGraphExecDestroy(ex->graphExec[k]);
    for (int i = 0; i < ORBX_MAX_LEVELS; ++i) {
        if (ex->qtStream[i]) { (void)hipStreamSynchronize(ex->qtStream[i]); (void)hipStreamDestroy(ex->qtStream[i]); }
        if (ex->evQt[i]) (void)hipEventDestroy(ex->evQt[i]);
        if (i < 4 && ex->evPart[i]) (void)hipEventDestroy(ex->evPart[i]);
        if (i < 4 && ex->evStag[i]) (void)hipEventDestroy(ex->evStag[i]);
    }
    if (ex->h_out) (void)hipHostFree(ex->h_out);
    ex->d_in.release(); ex->d_inPacked.release(); ex->d_outPack.release();
    for (auto &b : ex->d_raw) b.release();
    for (auto &b : ex->d_blur) b.release();
    ex->d_qtNodes.release(); ex->d_pfArgs.release();
    ex->d_tabs.release(); ex->d_geom.release(); ex->d_cand.release(); ex->d_scratchA.release(); ex->d_scratchB.release();
    ex->d_sel.release(); ex->d_workKey.release(); ex->d_tileCount.release(); ex->d_selCount.release();
    ex->d_workPos.release(); ex->d_n.release(); ex->d_mono.release(); ex->d_kps.release(); ex->d_desc.release();
    delete ex;
}

// ---------------------------------------------------------------------------------------------------------
// getters (reference include/ORBextractor.h:70-90)
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_get_levels(const orbx_extractor *ex) { return ex ? ex->prm.nlevels : 0; }
extern "C" float orbx_get_scale_factor(const orbx_extractor *ex) { return ex ? ex->prm.scale_factor : 0.f; }
#define ORBX_COPY_TABLE(fn, type, member)                                          \
    extern "C" int fn(const orbx_extractor *ex, type *out)                          \
    {                                                                               \
        if (!ex || !out) return setError(ORBX_ERR_INVALID_ARG, #fn ": null argument"); \
        for (size_t i = 0; i < ex->member.size(); ++i) out[i] = ex->member[i];      \
        return (int)ex->member.size();                                              \
    }
ORBX_COPY_TABLE(orbx_get_scale_factors, float, scale)
ORBX_COPY_TABLE(orbx_get_inverse_scale_factors, float, invScale)
ORBX_COPY_TABLE(orbx_get_scale_sigma_squares, float, sigma2)
ORBX_COPY_TABLE(orbx_get_inverse_scale_sigma_squares, float, invSigma2)
ORBX_COPY_TABLE(orbx_get_features_per_level, int32_t, quota)
ORBX_COPY_TABLE(orbx_get_umax, int32_t, umax)
extern "C" int orbx_max_keypoints(const orbx_extractor *ex) { return ex ? ex->maxKeypoints : 0; }
extern "C" void *orbx_stream(orbx_extractor *ex) { return ex ? (void *)ex->stream : nullptr; }

extern "C" int orbx_level_sizes(const orbx_extractor *ex, int w, int h, int32_t *lw, int32_t *lh)
{
    ORBX_API_ENTRY;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_level_sizes: null handle");
    for (int l = 0; l < ex->prm.nlevels; ++l) {
        if (lw) lw[l] = cvRoundF((float)w * ex->invScale[l]);
        if (lh) lh[l] = cvRoundF((float)h * ex->invScale[l]);
    }
    return ex->prm.nlevels;
}

// ---------------------------------------------------------------------------------------------------------
// geometry: level sizes (:1313), FAST cell grid (:870-886), tiles, cv::resize tables
// ---------------------------------------------------------------------------------------------------------
static int computeGeometry(orbx_extractor *ex, int w, int h)
{
    if (w == ex->curW && h == ex->curH && !ex->geom.empty()) return ORBX_OK;
    ORBX_SETUP_LOCK;
    ++ex->generation;  // captured launch sequences of the old geometry are stale
    const int nl = ex->prm.nlevels;
    std::vector<OrbxLevelGeom> geom(nl);
    long long candOff = 0;
    int tileOff = 0, selOff = 0;
    for (int l = 0; l < nl; ++l) {
        OrbxLevelGeom &g = geom[l];
        memset(&g, 0, sizeof g);
        g.w = cvRoundF((float)w * ex->invScale[l]);
        g.h = cvRoundF((float)h * ex->invScale[l]);
        g.pitch = (int)roundUp(g.w, 64);
        g.frameStride = (long long)roundUp((size_t)g.pitch * g.h, 256);
        // columns a tile stages beyond its last interior column when another column group follows: the 3 of the FAST ring, or what the
        // resize phase reads there - it produces up to 3 destination columns beyond the group's own (boundaries at multiples of 4, see
        // the column ownership below): ceil(3 * scale) source columns further, + 1 for the right tap
        int rightHalo = 3;
        if (l + 1 < nl) {
            const int w1 = cvRoundF((float)w * ex->invScale[l + 1]);
            rightHalo = std::max(3, (int)std::ceil(3.0 * g.w / std::max(w1, 1)) + 1);
        }
        g.rightHalo = rightHalo;
        const int minB = ORBX_BORDER, maxBX = g.w - ORBX_EDGE + 3, maxBY = g.h - ORBX_EDGE + 3;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const int nCols = (int)(width / 35.f), nRows = (int)(height / 35.f);
        if (nCols < 1 || nRows < 1 || g.w < 2 * ORBX_FAST_MARGIN + 1 || g.h < 2 * ORBX_FAST_MARGIN + 1)
            return setError(ORBX_ERR_TOO_SMALL, "pyramid level %d (%dx%d) is smaller than one 35-px FAST cell", l, g.w, g.h);
        // DistributeOctTree starts from nIni = round(W / H) root nodes (src/ORBextractor.cc:589): below 0.5 that is 0, hX = W / 0 and the
        // reference indexes an empty vector (:611-614) - undefined behaviour there, an error here
        if ((int)std::round(width / height) < 1)
            return setError(ORBX_ERR_INVALID_ARG, "pyramid level %d (%dx%d): aspect ratio below 0.5 leaves DistributeOctTree without a root node "
                                                  "(nIni = round(W / H) = 0, reference src/ORBextractor.cc:589)", l, g.w, g.h);
        g.wCell = (int)std::ceil(width / nCols);
        g.hCell = (int)std::ceil(height / nRows);
        const int iw = g.w - 2 * ORBX_FAST_MARGIN, ih = g.h - 2 * ORBX_FAST_MARGIN;  // corner-capable interior
        g.ncx = std::min(nCols, (iw + g.wCell - 1) / g.wCell);
        g.ncy = std::min(nRows, (ih + g.hCell - 1) / g.hCell);
        // target interior width of a tile: wide tiles for throughput (fewer halo re-reads, less fixed cost per pixel); handles for small
        // batches take narrow ones - a single frame is latency-bound, its launches should be many short workgroups
        static const int tileWEnv = [] { const char *e = getenv("ORBX_TILE_W"); return e ? atoi(e) : 0; }();
        // (round 4: 360 instead of 288 - with the kernel's loops shorter, what a WAVE does once weighs more (~300 scalar instructions of
        // geometry per wave, 10 % of the kernel), and wider tiles mean fewer waves per pixel: c2 1.59 -> 1.55 ms, c4 1.38 -> 1.33, c3 - 1 %,
        // c5 unchanged; three workgroups of <= 53 KB per CU instead of four of <= 40 KB - tools/experiments/pf_tile_cfg_r4.sh)
        const int tileW = tileWEnv > 0 ? tileWEnv : (ex->prm.max_batch <= ORBX_GRAPH_MAX_FRAMES ? 108 : 360);
        // cells are 35 .. 69 px high (height / nRows): beyond the usual 35 .. 40 rows keep the tile area - and with it the LDS
        // footprint - about constant
        const int tileWl = std::max(g.wCell, tileW * 40 / std::max(g.hCell, 40));
        int gmax = std::max(1, tileWl / g.wCell);
        g.nGroups = (g.ncx + gmax - 1) / gmax;
        g.groupCells = (g.ncx + g.nGroups - 1) / g.nGroups;
        // three workgroups per CU need <= 53 KB of LDS each (160 KB; rounds 2 - 3: four of <= 40 KB): give up to two cells of tile width
        // for it - an estimate of the layout below with the resize tables at their nominal size; narrower than that the fixed cost per
        // tile weighs more than the next workgroup
        // ORBX_PF_LDS_FIT: 0 no fit; 1 (default) three workgroups per CU; > 1024: that many bytes per workgroup (40448 = four per CU, the
        // fit of rounds 2 - 3)
        static const int ldsFit = [] { const char *e = getenv("ORBX_PF_LDS_FIT"); return e ? atoi(e) : 1; }();
        if (ldsFit && tileWEnv <= 0 && ex->prm.max_batch > ORBX_GRAPH_MAX_FRAMES) {
            auto estimate = [&](int cells) {  // the LDS layout below for `cells` cells per group (resize tables at their nominal size)
                const int tw = cells * g.wCell, ng = (g.ncx + cells - 1) / cells;
                const size_t maxCols = ng == 1 ? (size_t)g.w : (size_t)tw + rightHalo + ORBX_FAST_MARGIN;   // edge groups hold the image margin
                const size_t rows = g.ncy == 1 ? (size_t)g.h : (size_t)g.hCell + 3 + ORBX_FAST_MARGIN;
                const size_t tp = roundUp(maxCols + 32, 16), sp = roundUp((size_t)tw, 16);
                const size_t dx = roundUp((size_t)((tw + ORBX_FAST_MARGIN) / ex->prm.scale_factor) + 4, 2);
                const size_t dy = roundUp((size_t)((g.hCell + ORBX_FAST_MARGIN) / ex->prm.scale_factor) + 4, 2);
                const size_t gq = roundUp((size_t)2 * ((tw + 7) / 8 + 1) * g.hCell, 8);
                return rows * tp + (g.hCell + 2) * sp + roundUp((size_t)cells * g.hCell * ((g.wCell + 31) / 32) * 4, 16) +
                       roundUp(std::max((dx + dy) * 8, (size_t)2048), 16) + gq * 2 + roundUp((size_t)tw, 16);
            };
            const size_t limit = ldsFit > 1024 ? (size_t)ldsFit : 160 * 1024 / 3 - 1024;  // (static LDS of the kernel, allocation granularity)
            const int wide = g.groupCells;
            for (int drop = 0; drop <= 2 && wide - drop >= 4; ++drop) {
                const int ng = (g.ncx + wide - drop - 1) / (wide - drop), balanced = (g.ncx + ng - 1) / ng;
                if (balanced < wide - 2) break;  // (a narrow level would fall apart into much smaller tiles)
                if (estimate(balanced) <= limit) { g.groupCells = balanced; break; }
            }
        }
        g.nGroups = (g.ncx + g.groupCells - 1) / g.groupCells;
        g.groupCells = (g.ncx + g.nGroups - 1) / g.nGroups;  // (balanced over the groups)
        g.nGroups = (g.ncx + g.groupCells - 1) / g.groupCells;
        if (g.groupCells > 32) return setError(ORBX_ERR_CAPACITY, "level %d: %d cells per tile group exceeds 32", l, g.groupCells);
        g.nTiles = g.nGroups * g.ncy;
        g.slotCap = g.groupCells * ((g.wCell + 1) / 2) * ((g.hCell + 1) / 2);
        g.candBase = candOff; candOff += (long long)g.nTiles * g.slotCap;
        g.tileBase = tileOff; tileOff += g.nTiles;
        g.quota = ex->quota[l];
        g.selBase = selOff; selOff += ex->levelCap[l];
        // LDS layout of k_pyr_fast
        int maxCols = 0;
        for (int gi = 0; gi < g.nGroups; ++gi) {
            int c0 = gi * g.groupCells, c1 = std::min(c0 + g.groupCells, g.ncx);
            int tx0 = ORBX_FAST_MARGIN + c0 * g.wCell, tx1 = std::min(ORBX_FAST_MARGIN + c1 * g.wCell, g.w - ORBX_FAST_MARGIN);
            int lx0 = gi == 0 ? 0 : tx0 - 3, lx1 = gi == g.nGroups - 1 ? g.w : tx1 + rightHalo;
            maxCols = std::max(maxCols, lx1 - (lx0 & ~15));
        }
        // LDS column 16 holds image column ax0; strip windows / 16-byte loads may overshoot by up to 16 bytes
        g.tilePitch = (int)roundUp(maxCols + 32, 16);  // XO (16) + widest overshoot of strip windows (24) / 16-byte loads (31 - 16)
        g.scorePitch = (int)roundUp((size_t)g.groupCells * g.wCell, 16);  // scores exist for interior columns only (cell-local NMS)
        g.wpc = (g.wCell + 31) / 32;
        // Rows of the pixel tile.  The first and the last cell-row also own the image rows beyond their 3-px halo (up to the border: sources
        // of the next level only, 16 rows at the top).  Held in the pixel tile they cost EVERY workgroup of the launch those 16 LDS rows -
        // and, with 360-px tiles, the fourth resident workgroup per CU; so an edge tile stages them as "margin bands" in the bytes of the
        // score tile + bitmaps (OrbxPfTile::bandTop / bandBot) whenever they fit there.  The last level has no next level: no extension.
        // (k_pyr_fast_cw keeps the classic layout: its resize tables are not aliased and its score tile is zeroed in the prologue.)
        {
            static const bool cwKernel = [] { const char *e = getenv("ORBX_PF_KERNEL"); return e && atoi(e) == 2; }();
            const int lastTy0 = ORBX_FAST_MARGIN + (g.ncy - 1) * g.hCell, lastTy1 = std::min(lastTy0 + g.hCell, g.h - ORBX_FAST_MARGIN);
            const int topBand = ORBX_FAST_MARGIN - 3, botBand = g.h - std::min(lastTy1 + 3, g.h);
            const size_t scoreBytes = (size_t)(g.hCell + 2) * g.scorePitch + roundUp((size_t)g.groupCells * g.hCell * g.wpc * 4, 16);
            g.bands = (l + 1 < nl && !cwKernel && (size_t)(topBand + botBand) * g.tilePitch <= scoreBytes) ? 1 : 0;
            int maxRows = 0;
            for (int c = 0; c < g.ncy; ++c) {
                int ty0 = ORBX_FAST_MARGIN + c * g.hCell, ty1 = std::min(ty0 + g.hCell, g.h - ORBX_FAST_MARGIN);
                const bool extend = l + 1 < nl && !g.bands;
                int ly0 = c == 0 && extend ? 0 : ty0 - 3, ly1 = c == g.ncy - 1 && extend ? g.h : std::min(ty1 + 3, g.h);
                maxRows = std::max(maxRows, ly1 - ly0);
            }
            g.tileRows = maxRows;
        }
        g.rsMaxDx = g.rsMaxDy = 0;  // filled in with the resize tables below
    }
    // resize coefficient tables + tile ownership of destination rows / columns, per level transition
    std::vector<int> tabs;
    std::vector<int> offs((size_t)nl * 6, 0);
    for (int l = 0; l + 1 < nl; ++l) {
        const OrbxLevelGeom &s = geom[l], &d = geom[l + 1];
        const int COEF = 2048;
        const double scale_x = 1. / ((double)d.w / s.w), scale_y = 1. / ((double)d.h / s.h);
        std::vector<int> xofs(d.w), xab(d.w), yofs(d.h), yab(d.h), rowBeg(s.ncy + 1), colBeg(s.nGroups + 1);
        for (int dx = 0; dx < d.w; ++dx) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = cvFloorF(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= s.w - 1) { fx = 0; sx = s.w - 1; }
            xofs[dx] = sx;
            int a0 = satShort(cvRoundF((1.f - fx) * COEF)), a1 = satShort(cvRoundF(fx * COEF));
            xab[dx] = (a0 & 0xFFFF) | (a1 << 16);
        }
        for (int dy = 0; dy < d.h; ++dy) {
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            int sy = cvFloorF(fy);
            fy -= sy;
            yofs[dy] = sy;
            int b0 = satShort(cvRoundF((1.f - fy) * COEF)), b1 = satShort(cvRoundF(fy * COEF));
            yab[dy] = (b0 & 0xFFFF) | (b1 << 16);
        }
        // destination row dy is produced by the source cell-row that owns source row clamp(yofs[dy])
        auto rowOwner = [&](int sy) {
            sy = std::min(std::max(sy, 0), s.h - 1);
            int c = (sy - ORBX_FAST_MARGIN) / s.hCell;
            if (sy < ORBX_FAST_MARGIN) c = 0;
            return std::min(c, s.ncy - 1);
        };
        auto colOwner = [&](int sx) {
            int c = sx < ORBX_FAST_MARGIN ? 0 : (sx - ORBX_FAST_MARGIN) / s.wCell;
            c = std::min(c, s.ncx - 1);
            return c / s.groupCells;
        };
        {
            int dy = 0;
            for (int c = 0; c < s.ncy; ++c) {
                rowBeg[c] = dy;
                while (dy < d.h && rowOwner(yofs[dy]) == c) ++dy;
            }
            rowBeg[s.ncy] = dy;
            if (dy != d.h) return setError(ORBX_ERR_INVALID_ARG, "internal: resize row ownership is not monotone (level %d)", l);
            int dx = 0;
            for (int gi = 0; gi < s.nGroups; ++gi) {
                colBeg[gi] = dx;
                while (dx < d.w && colOwner(xofs[dx]) == gi) ++dx;
            }
            colBeg[s.nGroups] = dx;
            if (dx != d.w) return setError(ORBX_ERR_INVALID_ARG, "internal: resize column ownership is not monotone (level %d)", l);
            // The resize phase of k_pyr_fast writes 4 destination pixels per lane with one 32-bit store: the boundaries between the column
            // groups move up to the next multiple of 4, so that no 4-pixel group straddles two tiles (a lane with a partial group stored
            // byte by byte under four exec masks - ~17 scalar instructions per turn of EVERY lane's loop).  The left tile then produces up
            // to 3 more destination columns, whose right taps lie up to 5 source columns beyond its last interior column: the tiles of
            // such groups stage OrbxLevelGeom::rightHalo (5 at scale 1.2) instead of 3 columns there.  (The last group ends at the image edge: its last
            // store may run into the row's padding, pitch = roundUp(w, 64).)
            for (int gi = 1; gi < s.nGroups; ++gi) {
                colBeg[gi] = std::min((colBeg[gi] + 3) & ~3, d.w);
                const int c1 = std::min(gi * s.groupCells, s.ncx);
                const int tx1 = std::min(ORBX_FAST_MARGIN + c1 * s.wCell, s.w - ORBX_FAST_MARGIN);
                if (colBeg[gi] > colBeg[gi - 1] && std::min(xofs[colBeg[gi] - 1] + 1, s.w - 1) >= tx1 + s.rightHalo)
                    return setError(ORBX_ERR_INVALID_ARG, "internal: resize column group %d of level %d reads beyond its staged columns", gi - 1, l);
            }
            for (int gi = 0; gi < s.nGroups; ++gi)
                if (colBeg[gi + 1] < colBeg[gi]) return setError(ORBX_ERR_INVALID_ARG, "internal: resize column ownership is not monotone (level %d)", l);
        }
        for (int c = 0; c < s.ncy; ++c) geom[l].rsMaxDy = std::max(geom[l].rsMaxDy, rowBeg[c + 1] - rowBeg[c]);
        for (int gi = 0; gi < s.nGroups; ++gi) geom[l].rsMaxDx = std::max(geom[l].rsMaxDx, colBeg[gi + 1] - colBeg[gi]);
        std::vector<int> *parts[6] = {&xofs, &xab, &yofs, &yab, &rowBeg, &colBeg};
        for (int k = 0; k < 6; ++k) {
            offs[(size_t)l * 6 + k] = (int)tabs.size();
            tabs.insert(tabs.end(), parts[k]->begin(), parts[k]->end());
        }
    }
    for (int l = 0; l < nl; ++l) {
        OrbxLevelGeom &g = geom[l];
        g.rsMaxDx = (int)roundUp(g.rsMaxDx + 1, 2);
        g.rsMaxDy = (int)roundUp(g.rsMaxDy + 1, 2);
        // strips cover [8s + 4, 8s + 12): at most (interior width + 7) / 8 + 1 strips per row, 2 groups each
        const int gqFull = (int)roundUp((size_t)2 * ((g.groupCells * g.wCell + 7) / 8 + 1) * g.hCell, 8);
        g.gqCap = gqFull;
        // LDS sub-buffers: pixel tile | score tile | NMS bitmaps (zeroed together with the score tile: 16-byte multiple) | resize
        // tables | group queue | corner queue (1024 entries) | group masks | per-column cell info
        g.oScore = g.tileRows * g.tilePitch;
        g.oBits = g.oScore + (g.hCell + 2) * g.scorePitch;
        g.oXt = g.oBits + (int)roundUp((size_t)g.groupCells * g.hCell * g.wpc * 4, 16);
        g.oYt = g.oXt + g.rsMaxDx * 8;
        // the corner queue lives in the bytes of the resize tables: the tables are last read in the resize phase, the queue is first
        // written in the ring-test phase, behind the barrier that ends the pre-test - every wave has left the resize phase by then
        g.oCq = g.oXt;
        g.oGq = g.oXt + (int)roundUp(std::max((size_t)(g.rsMaxDx + g.rsMaxDy) * 8, (size_t)1024 * 2), 16);
        g.oGqm = 0;
        {
            // A queue for EVERY group of the tile is 6 KB at VGA and fills to 11 - 22 % at iniThFAST: where the full queue is what pushes the
            // workgroup beyond a quarter of the CU's LDS (160 KB: 40960 B per workgroup, static LDS and the 1280-byte allocation granule
            // included), it holds what is left - at least half of the groups and 1024 entries, otherwise the level stays at three workgroups
            // per CU.  A round that queues more runs the dense path (k_pyr_fast (B)).  ORBX_PF_GQCAP: a fixed capacity (tests of that path).
            static const int gqCapEnv = [] { const char *e = getenv("ORBX_PF_GQCAP"); return e ? atoi(e) : 0; }();
            const int colBytes = (int)roundUp(g.groupCells * g.wCell, 16);
            const int rest = g.oGq + 16 + colBytes;   // everything but the queue's entries (16: the dump slot, rounded)
            const int limit4 = 160 * 1024 / 4 - 160;
            if (rest + 2 * gqFull > limit4) {
                const int fit = ((limit4 - rest) / 2) & ~7;
                if (fit >= std::max(1024, gqFull / 2)) g.gqCap = fit;
            }
            if (gqCapEnv > 0) g.gqCap = std::min(gqFull, std::max(32, gqCapEnv & ~7));   // (>= 64 B: the emission phase keeps its wave totals there)
        }
        g.oColInfo = g.oGq + (int)roundUp((size_t)(g.gqCap + 1) * 2, 16);   // (+ 1: the dump slot behind the queue)
        g.ldsBytes = (int)roundUp((size_t)g.oColInfo + roundUp(g.groupCells * g.wCell, 16), 16);
        if (getenv("ORBX_DEBUG"))
            fprintf(stderr, "[orbx] level %d LDS layout: tile %d x %d = %d | score %d x %d | bits %d | tables %d + %d / corner queue | group queue %d x 2 | "
                            "columns %d -> %d B%s\n", l, g.tileRows, g.tilePitch, g.oScore, g.hCell + 2, g.scorePitch, g.oXt - g.oBits, g.rsMaxDx * 8,
                    g.rsMaxDy * 8, g.gqCap, g.groupCells * g.wCell, g.ldsBytes, g.bands ? " (margin bands in the score tile)" : "");
        if (g.ldsBytes > 64 * 1024)
            return setError(ORBX_ERR_CAPACITY, "level %d: fused pyramid+FAST tile needs %d B of LDS (> 64 KiB)", l, g.ldsBytes);
        // k_pyr_fast_cw: 8 rings of 128 u16 (one per wave), a list of cwClCap u16 per cell (ring-test survivors in raster order, compacted in
        // place to the keypoints of the cell; a cell with more corners than that takes the scanning path - ORBX_PF_CLCAP lowers it for tests)
        static const int clCapEnv = [] { const char *e = getenv("ORBX_PF_CLCAP"); return e ? std::min(std::max(atoi(e), 8), 2048) : 192; }();
        g.cwClCap = (clCapEnv + 7) & ~7;
        g.cwXt = g.oScore + (g.hCell + 2) * g.scorePitch;
        g.cwYt = g.cwXt + g.rsMaxDx * 8;
        g.cwGq = g.cwYt + (int)roundUp((size_t)g.rsMaxDy * 8, 16);
        g.cwCl = g.cwGq + 8 * 128 * 2;
        g.cwLdsBytes = (int)roundUp((size_t)g.cwCl + (size_t)g.groupCells * g.cwClCap * 2, 16);
        if (g.cwLdsBytes > 64 * 1024)
            return setError(ORBX_ERR_CAPACITY, "level %d: fused pyramid+FAST tile needs %d B of LDS (> 64 KiB)", l, g.cwLdsBytes);
    }
    // tile records of k_pyr_fast (OrbxPfTile): everything that depends only on (cell-row, cell group) of a level
    std::vector<int> tileOffs(nl, 0);
    for (int l = 0; l < nl; ++l) {
        const OrbxLevelGeom &g = geom[l];
        while (tabs.size() % 32) tabs.push_back(0);  // 128-byte records, 128-byte aligned inside the table buffer
        tileOffs[l] = (int)tabs.size();
        const int *rowBeg = l + 1 < nl ? &tabs[offs[(size_t)l * 6 + 4]] : nullptr, *colBeg = l + 1 < nl ? &tabs[offs[(size_t)l * 6 + 5]] : nullptr;
        std::vector<OrbxPfTile> recs((size_t)g.nTiles);
        for (int crow = 0; crow < g.ncy; ++crow)
            for (int grp = 0; grp < g.nGroups; ++grp) {
                OrbxPfTile &t = recs[(size_t)crow * g.nGroups + grp];
                memset(&t, 0, sizeof t);
                const int ty0 = ORBX_FAST_MARGIN + crow * g.hCell, ty1 = std::min(ty0 + g.hCell, g.h - ORBX_FAST_MARGIN);
                const int c0 = grp * g.groupCells, c1 = std::min(c0 + g.groupCells, g.ncx);
                const int tx0 = ORBX_FAST_MARGIN + c0 * g.wCell, tx1 = std::min(ORBX_FAST_MARGIN + c1 * g.wCell, g.w - ORBX_FAST_MARGIN);
                // rows / columns staged in LDS: interior + 3-px ring halo; edge tiles extend to the image border so that the tile also
                // owns the source pixels of the next pyramid level there - in x inside the pixel tile, in y as margin bands in the score
                // tile's bytes (g.bands) or, where those do not fit, inside the pixel tile as well
                const bool extend = l + 1 < nl && !g.bands;
                const int ly0 = crow == 0 && extend ? 0 : ty0 - 3, ly1 = crow == g.ncy - 1 && extend ? g.h : std::min(ty1 + 3, g.h);
                const int lx0 = grp == 0 ? 0 : tx0 - 3, lx1 = grp == g.nGroups - 1 ? g.w : tx1 + g.rightHalo;
                t.ty0 = ty0; t.THi = ty1 - ty0; t.ly0 = ly0; t.nRowsL = ly1 - ly0; t.rowOff = ty0 - ly0;
                t.bandTop = g.bands && crow == 0 ? ly0 : 0; t.bandBot = g.bands && crow == g.ncy - 1 ? g.h - ly1 : 0;
                t.ax0 = lx0 & ~15; t.lx1 = lx1; t.nV = (lx1 - t.ax0 + 15) >> 4; t.nVec = (t.bandTop + t.nRowsL + t.bandBot) * t.nV;
                if (t.nRowsL > g.tileRows || (size_t)(t.bandTop + t.bandBot) * g.tilePitch > (size_t)(g.oXt - g.oScore))
                    return setError(ORBX_ERR_INVALID_ARG, "internal: level %d: tile rows %d (+ %d + %d) exceed the LDS layout", l, t.nRowsL, t.bandTop, t.bandBot);
                auto magic20 = [](int n, int range) {  // ceil(2^20 / n) if it divides every i <= range exactly, else 0
                    if (n <= 0) return 0;
                    const unsigned m = ((1u << 20) + (unsigned)n - 1) / (unsigned)n;
                    for (int i = 0; i <= range; ++i)
                        if ((int)(((unsigned)i * m) >> 20) != i / n) return 0;
                    return (int)m;
                };

                t.colOf0 = 16 - t.ax0;  // PF_XO: LDS column 16 holds image column ax0
                t.cInt0 = tx0 + t.colOf0; t.cInt1 = tx1 + t.colOf0;
                t.nCells = c1 - c0; t.TWi = tx1 - tx0;
                t.s0 = (t.cInt0 - 4) >> 3; t.NS = ((t.cInt1 - 1 - 4) >> 3) - t.s0 + 1;
                if (t.NS >= orbxPyrFastThreads())  // (the pre-test deals its lanes over (row class, strip): at least one row class)
                    return setError(ORBX_ERR_CAPACITY, "level %d: %d 8-pixel strips per tile row exceed the workgroup's %d lanes", l, t.NS, orbxPyrFastThreads());
                t.nVMagic = magic20(t.nV, 1023);
                if (!t.nVMagic) return setError(ORBX_ERR_INVALID_ARG, "internal: level %d: no 20-bit reciprocal for %d vectors per tile row", l, t.nV);
                t.RC = 1;
                if (rowBeg) {
                    t.dy0 = rowBeg[crow]; t.dy1 = rowBeg[crow + 1]; t.dx0 = colBeg[grp]; t.dx1 = colBeg[grp + 1];
                    t.q0 = t.dx0 >> 2; t.nq = ((t.dx1 + 3) >> 2) - t.q0;
                    t.RC = t.nq > 0 ? std::max(1, orbxPyrFastThreads() / t.nq) : 1;  // resize row classes: 512 lanes over (row class, 4-pixel group)
                    t.nqMagic = t.nq > 0 && t.nq < orbxPyrFastThreads() ? magic20(t.nq, 511) : 0;
                    if (t.nq > 0 && t.nq < orbxPyrFastThreads() && !t.nqMagic)
                        return setError(ORBX_ERR_INVALID_ARG, "internal: level %d: no 20-bit reciprocal for %d pixel groups per row", l, t.nq);
                }
            }
        const int *ri = (const int *)recs.data();
        tabs.insert(tabs.end(), ri, ri + recs.size() * (sizeof(OrbxPfTile) / sizeof(int)));
    }
    // k_pyr_chain (the whole pyramid of a small batch in one launch): the pixels of every level are dealt over one grid of tiles (32 x 32
    // at level 1, ORBX_CHAIN_TILE); per tile and level the rectangle it owns and the rectangle it needs = owned + the sources of what it needs one level up
    int chainOffset = 0, chainTiles = 0, chainBufBytes = 0, chainTabBytes = 0;
    if (nl >= 2 && ex->prm.max_batch <= ORBX_GRAPH_MAX_FRAMES) {
        static const int chainTile = [] { const char *e = getenv("ORBX_CHAIN_TILE"); return e ? std::min(std::max(atoi(e), 16), 128) : 32; }();
        const int nTx = (geom[1].w + chainTile - 1) / chainTile, nTy = (geom[1].h + chainTile - 1) / chainTile;
        auto cut = [](int extent, int i, int n, bool align4) {
            if (i <= 0) return 0;
            if (i >= n) return extent;
            const int c = (int)((long long)extent * i / n);
            return align4 ? c & ~3 : c;
        };
        while (tabs.size() % 4) tabs.push_back(0);
        chainOffset = (int)tabs.size();
        chainTiles = nTx * nTy;
        std::vector<int> rects((size_t)chainTiles * nl * 8, 0);
        for (int j = 0; j < nTy; ++j)
            for (int i = 0; i < nTx; ++i) {
                int *rt = &rects[((size_t)j * nTx + i) * nl * 8];
                for (int l = 1; l < nl; ++l) {
                    int *q = rt + l * 8;
                    q[4] = cut(geom[l].w, i, nTx, true); q[5] = cut(geom[l].w, i + 1, nTx, true);
                    q[6] = cut(geom[l].h, j, nTy, false); q[7] = cut(geom[l].h, j + 1, nTy, false);
                }
                int nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;  // needed one level up (empty)
                int tabEntries = 0;
                for (int l = nl - 1; l >= 0; --l) {
                    int *q = rt + l * 8;
                    int x0 = 0, x1 = 0, y0 = 0, y1 = 0;
                    if (l + 1 < nl && nx1 > nx0 && ny1 > ny0) {  // sources of needed[l + 1] in level l (transition tables l -> l + 1)
                        const int *xo = &tabs[offs[(size_t)l * 6 + 0]], *yo = &tabs[offs[(size_t)l * 6 + 2]];
                        const int wm1 = geom[l].w - 1, hm1 = geom[l].h - 1;
                        x0 = std::min(std::max(xo[nx0], 0), wm1); x1 = std::min(std::max(xo[nx1 - 1], 0) + 1, wm1) + 1;
                        y0 = std::min(std::max(yo[ny0], 0), hm1); y1 = std::min(std::max(yo[ny1 - 1] + 1, 0), hm1) + 1;
                    }
                    if (l >= 1 && q[5] > q[4] && q[7] > q[6]) {  // + what the tile owns at this level
                        if (x1 > x0 && y1 > y0) { x0 = std::min(x0, q[4]); x1 = std::max(x1, q[5]); y0 = std::min(y0, q[6]); y1 = std::max(y1, q[7]); }
                        else { x0 = q[4]; x1 = q[5]; y0 = q[6]; y1 = q[7]; }
                    }
                    q[0] = x0; q[1] = x1; q[2] = y0; q[3] = y1;
                    const int pitch = l == 0 ? ((x1 - x0) + 3) & ~3 : (((x1 + 3) >> 2) - (x0 >> 2)) * 4;
                    chainBufBytes = std::max(chainBufBytes, pitch * (y1 - y0));
                    if (l >= 1) tabEntries += pitch + (y1 - y0);
                    nx0 = x0; nx1 = x1; ny0 = y0; ny1 = y1;
                }
                chainTabBytes = std::max(chainTabBytes, tabEntries * 8);
            }
        chainBufBytes = (int)roundUp((size_t)chainBufBytes + 16, 16);
        chainTabBytes = (int)roundUp((size_t)chainTabBytes + 16, 16);
        tabs.insert(tabs.end(), rects.begin(), rects.end());
        if (2 * (size_t)chainBufBytes + chainTabBytes > 64 * 1024) chainTiles = 0;  // (does not happen with tiles of this size)
    }
    if (tabs.empty()) tabs.push_back(0);
    (void)hipSetDevice(ex->prm.device);
    int rc = ex->d_tabs.ensure(tabs.size());
    if (rc != ORBX_OK) return rc;
    rc = ex->d_geom.ensure(nl);
    if (rc != ORBX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(ex->stream));
    HIP_TRY(hipMemcpy(ex->d_tabs.p, tabs.data(), tabs.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ex->d_geom.p, geom.data(), nl * sizeof(OrbxLevelGeom), hipMemcpyHostToDevice));
    ex->geom.swap(geom);
    ex->tabOffsets.swap(offs);
    ex->tileOffsets.swap(tileOffs);
    ex->chainOffset = chainOffset; ex->chainTiles = chainTiles; ex->chainBufBytes = chainBufBytes; ex->chainTabBytes = chainTabBytes;
    ex->candFrameStride = candOff;
    ex->tilesPerFrame = tileOff;
    ex->selFrameStride = selOff;
    ex->curW = w; ex->curH = h;
    return ORBX_OK;
}

static int ensureBuffers(orbx_extractor *ex)
{
    const int nl = ex->prm.nlevels;
    const size_t B = (size_t)ex->prm.max_batch;
    int rc;
    for (int l = 0; l < nl; ++l) {
        const OrbxLevelGeom &g = ex->geom[l];
        if (l > 0 && (rc = ex->d_raw[l].ensure(B * g.frameStride + 64)) != ORBX_OK) return rc;
        if ((rc = ex->d_blur[l].ensure(B * (size_t)blurFrameBytes(g) + 64)) != ORBX_OK) return rc;
    }
    {   // quadtree node arrays in global memory for level quotas beyond LDS (list capacity > ~3500 nodes): sized for the worst launch
        size_t need = 0;
        for (int l0 = 0; l0 < nl; ++l0)
            for (int run = 1; l0 + run <= nl; ++run) need = std::max(need, orbxQuadtreeScratchBytes((int)B, ex->geom.data(), l0, run));
        if (need && (rc = ex->d_qtNodes.ensure(need)) != ORBX_OK) return rc;
    }
    ex->inPitch = ex->geom[0].pitch;
    ex->inFrameStride = (long long)ex->inPitch * ex->geom[0].h;  // rows of consecutive frames are contiguous: one bulk H2D copy per batch
    if ((rc = ex->d_in.ensure(B * ex->inFrameStride + 64)) != ORBX_OK) return rc;
    if ((rc = ex->d_cand.ensure(B * ex->candFrameStride)) != ORBX_OK) return rc;
    if ((rc = ex->d_scratchA.ensure(B * ex->candFrameStride)) != ORBX_OK) return rc;
    if ((rc = ex->d_scratchB.ensure(B * ex->candFrameStride)) != ORBX_OK) return rc;
    if ((rc = ex->d_tileCount.ensure(B * ex->tilesPerFrame)) != ORBX_OK) return rc;
    if ((rc = ex->d_sel.ensure(B * ex->selFrameStride)) != ORBX_OK) return rc;
    if ((rc = ex->d_selCount.ensure(B * nl)) != ORBX_OK) return rc;
    const size_t cap = (size_t)ex->maxKeypoints;
    if ((rc = ex->d_workKey.ensure(B * cap)) != ORBX_OK) return rc;
    if ((rc = ex->d_workPos.ensure(B * cap)) != ORBX_OK) return rc;
    if ((rc = ex->d_kps.ensure(B * cap)) != ORBX_OK) return rc;
    if ((rc = ex->d_desc.ensure(B * cap * 32)) != ORBX_OK) return rc;
    if ((rc = ex->d_n.ensure(B)) != ORBX_OK) return rc;
    if ((rc = ex->d_mono.ensure(B)) != ORBX_OK) return rc;
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// stage profiling (HIP events on the handle's stream)
// ---------------------------------------------------------------------------------------------------------
static const int PROF_RING = 32;
#define ORBX_MAX_SUB 8
static const int PROF_EVENTS = ORBX_MAX_SUB * ORBX_NUM_STAGES * 2;  // [sub-batch][stage][begin, end]

extern "C" int orbx_set_profiling(orbx_extractor *ex, int enable)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_set_profiling: null handle");
    HIP_TRY(hipSetDevice(ex->prm.device));
    if (enable && ex->events.empty()) {
        ex->events.resize((size_t)PROF_RING * PROF_EVENTS);
        ex->profSubs.assign(PROF_RING, 0);
        for (auto &e : ex->events) HIP_TRY(hipEventCreate(&e));
    }
    ex->profiling = enable != 0;
    ex->profCount = 0;
    return ORBX_OK;
}

extern "C" int orbx_get_stage_times(orbx_extractor *ex, float *ms)
{
    ORBX_API_ENTRY;
    if (!ex || !ms) return setError(ORBX_ERR_INVALID_ARG, "orbx_get_stage_times: null argument");
    for (int s = 0; s < ORBX_NUM_STAGES; ++s) ms[s] = 0.f;
    HIP_TRY(hipSetDevice(ex->prm.device));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    const int nb = std::min(ex->profCount, PROF_RING);
    for (int b = 0; b < nb; ++b) {
        const int slot = (ex->profCount - 1 - b) % PROF_RING;
        hipEvent_t *ev = &ex->events[(size_t)slot * PROF_EVENTS];
        for (int sb = 0; sb < ex->profSubs[slot]; ++sb)
            for (int s = 0; s < ORBX_NUM_STAGES; ++s) {
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, ev[(sb * ORBX_NUM_STAGES + s) * 2], ev[(sb * ORBX_NUM_STAGES + s) * 2 + 1]));
                ms[s] += t;
            }
    }
    for (int s = 0; s < ORBX_NUM_STAGES && nb > 0; ++s) ms[s] /= (float)nb;
    ex->profCount = 0;
    return nb;
}

// ---------------------------------------------------------------------------------------------------------
// extraction
// ---------------------------------------------------------------------------------------------------------
static OrbxTables makeTables(const orbx_extractor *ex)
{
    OrbxTables t;
    memset(&t, 0, sizeof t);
    t.nlevels = ex->prm.nlevels;
    t.trigMode = ex->prm.trig_mode;
    for (int l = 0; l < t.nlevels; ++l) {
        t.scale[l] = ex->scale[l];
        t.quota[l] = ex->quota[l];
        t.selBase[l] = ex->geom[l].selBase;
        t.patchSize[l] = (int)(31 * ex->scale[l]);  // const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level], :1017
    }
    for (int i = 0; i < 16; ++i) { t.umax[i] = ex->umax[i]; t.umaxPacked |= (unsigned long long)(ex->umax[i] & 15) << (4 * i); }
    for (int i = 0; i < 7; ++i) t.taps[i] = ex->taps[i];
    return t;
}

// Launch sequence of one batch.  Frames are independent, so the batch can be cut into up to ORBX_MAX_SUB sub-batches
// (env ORBX_SUBBATCHES; default 1 — measured on MI355X: 1 -> 6.73 ms, 2 -> 6.69 ms, 4 -> 6.94 ms, 8 -> 7.60 ms per
// 1024-frame batch: the stages are throughput-bound in aggregate, so deeper pipelining only adds contention) and
// software-pipelined over three streams: the VALU-bound fused pyramid+FAST launches of sub-batch s+1 (front stream)
// overlap the GaussianBlur (blur stream) and the latency-bound quadtree -> pack -> orientation+descriptor chain (back
// stream) of sub-batch s.  No buffer is shared between sub-batches (everything is indexed by frame), so no double
// buffering is needed.  The front stream finally waits for the back stream: orbx_sync() and the next batch order
// themselves after all work of this batch.
// Host-pointer API (orbx_extract_batch and the sharded form): where the frames of this handle's batch lie in host memory, and where its
// results go.  Local frame i of the handle is host frame hostIndex(i): runs of `run` consecutive frames, `runStep` frames from one run's
// start to the next (a plain batch: one run; the sharded call deals blocks of frames round-robin over the handles, so handle g of G sees
// run = block, runStep = G * block and pointers that start at its first block).  Results use the same mapping into the caller's arrays.
struct HostPipe;
struct HostFrames {
    const uint8_t *imgs; ptrdiff_t stride, frameStride;
    int run, runStep;
    HostPipe *pipe;
    size_t hostIndex(int i) const { return (size_t)(i / run) * (size_t)runStep + (size_t)(i % run); }
};

static bool isPageLocked(const void *p)
{
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }  // plain malloc'ed memory: "invalid value"
    return a.type == hipMemoryTypeHost;
}
// the whole range [p, p + bytes): orbx_host_register accepts any sub-range of a buffer, so the first byte alone does not say that the
// tail is page-locked too (a pageable tail would turn the asynchronous copies into blocking ones)
static bool isPageLockedRange(const void *p, size_t bytes)
{
    if (!isPageLocked(p)) return false;
    return bytes <= 1 || isPageLocked((const uint8_t *)p + bytes - 1);
}
// after an error in the middle of a batch: nothing of the failed call may still run (or copy out of / into the caller's arrays) when the
// next call on this handle starts
static void drainHandle(orbx_extractor *ex)
{
    for (hipStream_t st : {ex->stream, ex->stream2, ex->stream3, ex->streamCopy, ex->streamOut})
        if (st) (void)hipStreamSynchronize(st);
    for (hipStream_t st : ex->qtStream)
        if (st) (void)hipStreamSynchronize(st);
}

// H2D of frames [f0, f0 + nf) of a host batch into the pitched device buffer.  A host-to-device copy whose two pitches differ is a 2-D
// copy, which the runtime executes row by row: 3.0 ms for ONE 1242 x 375 frame (KITTI) against 0.03 ms for a 1024-wide one whose pitch
// happens to equal its width.  Packed host frames therefore cross the link as linear copies into a packed device buffer, and the
// pitch is applied by a small kernel behind them on the same stream.  One copy per run of consecutive host frames.
static int stageHostFrames(orbx_extractor *ex, uint8_t *dst, ptrdiff_t dstPitch, ptrdiff_t dstFrameStride, const HostFrames *host, int f0, int nf,
                           int nframesTotal, int w, int h, hipStream_t st)
{
    const bool rowsAlike = host->stride == dstPitch && host->frameStride == host->stride * h && dstFrameStride == dstPitch * h;
    const bool packed = host->stride == w && host->frameStride == (ptrdiff_t)w * h && h <= 65535 && nf <= 65535;
    const size_t frameBytes = (size_t)w * h;
    if (!rowsAlike && packed) {
        int rc = ex->d_inPacked.ensure((size_t)std::max(nframesTotal, ex->prm.max_batch) * frameBytes + 64);
        if (rc != ORBX_OK) return rc;
    }
    for (int i = f0; i < f0 + nf;) {
        const int len = std::min(f0 + nf - i, host->run - i % host->run);  // frames up to the end of this run
        const uint8_t *srcH = host->imgs + host->hostIndex(i) * (size_t)host->frameStride;
        uint8_t *dstF = dst + (size_t)i * dstFrameStride;
        if (rowsAlike) {  // identical layouts
            HIP_TRY(hipMemcpyAsync(dstF, srcH, (size_t)len * h * dstPitch - (dstPitch - w), hipMemcpyHostToDevice, st));
        } else if (packed) {
            HIP_TRY(hipMemcpyAsync(ex->d_inPacked.p + (size_t)i * frameBytes, srcH, (size_t)len * frameBytes, hipMemcpyHostToDevice, st));
        } else if (host->frameStride == host->stride * h && dstFrameStride == dstPitch * h) {
            HIP_TRY(hipMemcpy2DAsync(dstF, dstPitch, srcH, host->stride, w, (size_t)h * len, hipMemcpyHostToDevice, st));
        } else {
            for (int f = 0; f < len; ++f)
                HIP_TRY(hipMemcpy2DAsync(dstF + (size_t)f * dstFrameStride, dstPitch, srcH + (size_t)f * host->frameStride, host->stride, w, h,
                                         hipMemcpyHostToDevice, st));
        }
        i += len;
    }
    if (!rowsAlike && packed)
        HIP_TRY(orbxLaunchUnpackFrames(st, ex->d_inPacked.p + (size_t)f0 * frameBytes, dst + (size_t)f0 * dstFrameStride, w, h, dstPitch, dstFrameStride, nf));
    return ORBX_OK;
}

// The host side of one host-pointer batch (H2D of the frames and D2H of the results are part of the metric).  The link
// moves 55 - 57 GB/s in each direction on the GPU box, from pageable memory too - but a copy from or to PAGEABLE memory blocks the calling
// thread until it is done (the runtime stages or pins behind the call), and with the copies and the launches on one thread the copy
// engine idled while the thread enqueued a sub-batch's kernels and the kernels idled during the next copy (timeline of round 2's call,
// tools/experiments/host_batch_trace.sh: 11.5 ms per 1024 VGA frames for 5.7 ms of copy time).  So:
//  * frames in pageable memory are fed by a helper thread that does nothing but the blocking copies, sub-batch after sub-batch, and
//    signals each one; the calling thread enqueues a sub-batch's kernels as soon as its copy has been issued.  Page-locked frames
//    (orbx_host_alloc, hipHostMalloc, hipHostRegister) need no thread: their copies are asynchronous;
//  * the results of sub-batch s leave on a third stream as soon as its last kernel is done - into the caller's arrays if those are
//    page-locked and laid out like the handle's (same capacity, consecutive frames), else into page-locked staging of the handle, from
//    where the calling thread copies exactly n[f] rows per frame while later sub-batches are still in flight.
struct HostPipe {
    orbx_keypoint *kps; uint8_t *desc; int capacity; int32_t *nOut, *monoOut;
    bool feeder = false, direct = false;
    int slot = 0;   // orbx_extractor::slots[slot]: this batch's events and staging
    int nsub = 0, f0[ORBX_MAX_SUB], nf[ORBX_MAX_SUB], drained = 0;
    std::mutex mu; std::condition_variable cv; int staged = 0, feederRc = ORBX_OK; std::string feederErr;
    std::thread thread;
    ~HostPipe() { if (thread.joinable()) thread.join(); }
};

// orbx_extract_batch_submit .. orbx_extract_batch_wait: what the wait needs of the call - the pipe and the frame mapping live here, not on
// the submitting call's stack
struct PendingBatch {
    HostPipe pipe;
    HostFrames hf;
    bool small = false;   // a batch of the small-batch schedule: the results are fetched by the wait (one round trip)
    int nframes = 0, capacity = 0;
    orbx_keypoint *kps = nullptr; uint8_t *desc = nullptr; int32_t *nOut = nullptr, *monoOut = nullptr;
};
// (the pipe of the NEWEST pending batch: the one a running submit is queueing)
static HostPipe *pendingPipeOf(orbx_extractor *ex) { return ex->nPending > 0 ? &ex->pending[ex->nPending - 1]->pipe : nullptr; }
static void dropPending(orbx_extractor *ex)
{
    if (ex->nPending == 0) return;
    drainHandle(ex);
    for (int k = 0; k < ex->nPending; ++k) { delete ex->pending[k]; ex->pending[k] = nullptr; }
    ex->nPending = 0;
}

static int ensureStage(orbx_extractor *ex, int slot)
{
    orbx_extractor::PipeSlot &sl = ex->slots[slot];
    const size_t B = (size_t)ex->prm.max_batch, cap = (size_t)ex->maxKeypoints;
    const size_t need = ((2 * B * sizeof(int32_t) + 63) & ~(size_t)63) + B * cap * (sizeof(orbx_keypoint) + 32) + 64;
    if (sl.h_stageBytes >= need) return ORBX_OK;
    ORBX_SETUP_LOCK;
    if (sl.h_stage) { (void)hipHostFree(sl.h_stage); sl.h_stage = nullptr; sl.h_stageBytes = 0; }
    HIP_TRY(hipHostMalloc((void **)&sl.h_stage, need, hipHostMallocDefault));
    sl.h_stageBytes = need;
    return ORBX_OK;
}
static inline int32_t *stageN(orbx_extractor *ex, int slot) { return (int32_t *)ex->slots[slot].h_stage; }
static inline int32_t *stageMono(orbx_extractor *ex, int slot) { return (int32_t *)ex->slots[slot].h_stage + ex->prm.max_batch; }
static inline orbx_keypoint *stageKps(orbx_extractor *ex, int slot) { return (orbx_keypoint *)(ex->slots[slot].h_stage + ((2 * (size_t)ex->prm.max_batch * sizeof(int32_t) + 63) & ~(size_t)63)); }
static inline uint8_t *stageDesc(orbx_extractor *ex, int slot) { return (uint8_t *)(stageKps(ex, slot) + (size_t)ex->prm.max_batch * ex->maxKeypoints); }

// results of sub-batch sb: D2H on the output stream behind the back stream's last kernel of that sub-batch
static int pipeSendResults(orbx_extractor *ex, HostPipe *pp, int sb)
{
    const int f0 = pp->f0[sb], nf = pp->nf[sb];
    const size_t cap = (size_t)ex->maxKeypoints;
    HIP_TRY(hipEventRecord(ex->slots[pp->slot].evBack[sb], ex->stream3));
    HIP_TRY(hipStreamWaitEvent(ex->streamOut, ex->slots[pp->slot].evBack[sb], 0));
    HIP_TRY(hipMemcpyAsync(stageN(ex, pp->slot) + f0, ex->d_n.p + f0, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, ex->streamOut));
    HIP_TRY(hipMemcpyAsync(stageMono(ex, pp->slot) + f0, ex->d_mono.p + f0, (size_t)nf * sizeof(int32_t), hipMemcpyDeviceToHost, ex->streamOut));
    orbx_keypoint *hk = pp->direct ? pp->kps : stageKps(ex, pp->slot);
    uint8_t *hd = pp->direct ? pp->desc : stageDesc(ex, pp->slot);
    if (pp->kps) HIP_TRY(hipMemcpyAsync(hk + (size_t)f0 * cap, ex->d_kps.p + (size_t)f0 * cap, (size_t)nf * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, ex->streamOut));
    if (pp->desc) HIP_TRY(hipMemcpyAsync(hd + (size_t)f0 * cap * 32, ex->d_desc.p + (size_t)f0 * cap * 32, (size_t)nf * cap * 32, hipMemcpyDeviceToHost, ex->streamOut));
    HIP_TRY(hipEventRecord(ex->slots[pp->slot].evOut[sb], ex->streamOut));
    return ORBX_OK;
}

// hands the finished sub-batches to the caller's arrays, in order; wait = false: only those whose copies have landed already
static int pipeDrain(orbx_extractor *ex, const HostFrames *host, bool wait, int upTo)
{
    HostPipe *pp = host->pipe;
    const size_t cap = (size_t)ex->maxKeypoints;
    while (pp->drained < upTo) {
        const int sb = pp->drained;
        if (wait) HIP_TRY(hipEventSynchronize(ex->slots[pp->slot].evOut[sb]));
        else {
            const hipError_t q = hipEventQuery(ex->slots[pp->slot].evOut[sb]);
            if (q == hipErrorNotReady) { (void)hipGetLastError(); return ORBX_OK; }
            if (q != hipSuccess) return setError(ORBX_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        }
        const int32_t *hn = stageN(ex, pp->slot), *hm = stageMono(ex, pp->slot);
        for (int f = pp->f0[sb]; f < pp->f0[sb] + pp->nf[sb]; ++f) {
            if (hn[f] > pp->capacity) return setError(ORBX_ERR_CAPACITY, "frame %d has %d keypoints, output capacity is %d", f, hn[f], pp->capacity);
            const size_t o = host->hostIndex(f);
            if (!pp->direct && hn[f] > 0) {
                if (pp->kps) memcpy(pp->kps + o * pp->capacity, stageKps(ex, pp->slot) + (size_t)f * cap, (size_t)hn[f] * sizeof(orbx_keypoint));
                if (pp->desc) memcpy(pp->desc + o * pp->capacity * 32, stageDesc(ex, pp->slot) + (size_t)f * cap * 32, (size_t)hn[f] * 32);
            }
            if (pp->nOut) pp->nOut[o] = hn[f];
            if (pp->monoOut) pp->monoOut[o] = hm[f];
        }
        ++pp->drained;
    }
    return ORBX_OK;
}

// ---- small batches ---------------------------------------------------------------------------------------------------
// A SLAM front end calls operator() one frame at a time (Frame::ExtractORB, reference src/Frame.cc:430-455).  For a single frame the
// critical path is the chain of dependent k_pyr_fast launches, then the longest quadtree instance (level 0, one wave), then pack +
// orientation.  Small batches therefore use a schedule of their own: the quadtree of the first levels starts on the back stream as
// soon as their candidates exist, each group on a branch of its own (ORBX_SMALL_QT_BRANCHES), and the handle's tiles are
// narrower (computeGeometry) so that a level's launch is many short workgroups instead of a few long ones.
// The sequence is captured once per (image size, source buffer, lapping area) as a hipGraph and replayed with one hipGraphLaunch per
// call: launched directly, the ~30 launches and ~15 event operations of a call keep the host behind the GPU (the k_pyr_fast chain
// stalls waiting for its next launch).  Measured (MI355X, ROCm 7.2, 640x480, device-resident call + sync): batch schedule 0.34 ms,
// this schedule launched directly 0.33 (1 - 2 branches) / 0.41 (4 branches), replayed as a graph 0.29 / 0.275 / 0.30 ms with
// 1 / 2 / 4 quadtree branches - with one-wave quadtree instances.  With the 8-wave instances (k_quadtree_mw, 0.06 instead of 0.15 ms
// for the level-0 instance) nothing is left to overlap: 0.189 / 0.200 / 0.202 / 0.242 ms with 1 / 2 / 3 / 4 branches; default 1.
// ORBX_GRAPH=0: batch schedule; 2: this schedule without capture (A/B runs, tools/experiments/single_frame_sweep.sh).

static int smallBatchPrepare(orbx_extractor *ex)
{
    // the quadtree branches run concurrently: not when a quota is so large that the node arrays live in the one global scratch area
    if (orbxQuadtreeScratchBytes(ex->prm.max_batch, ex->geom.data(), 0, ex->prm.nlevels) != 0) return 0;
    if (ex->qtStream[3] && ex->evQt[3]) return 1;
    ORBX_SETUP_LOCK;
    for (int k = 0; k < 4; ++k) {
        if (!ex->qtStream[k] && hipStreamCreateWithFlags(&ex->qtStream[k], hipStreamNonBlocking) != hipSuccess) return 0;
        if (!ex->evQt[k] && hipEventCreateWithFlags(&ex->evQt[k], hipEventDisableTiming) != hipSuccess) return 0;
    }
    return 1;
}

// Small batches run FAST of all levels in ONE launch behind a chain of resize-only launches (ORBX_SMALL_SPLIT=0: the fused per-level
// chain): the device array of per-level argument blocks for this source buffer, uploaded when it is not cached.  Synchronous - must run
// outside a stream capture.
static const bool kSmallSplit = [] { const char *e = getenv("ORBX_SMALL_SPLIT"); return e ? atoi(e) != 0 : true; }();

// graphSlot >= 0: the block of that graph slot (rewritten when it holds another call signature - the caller replaces the graph with it);
// -1: a call without a graph looks its signature up among the other blocks, round robin on a miss
static int prepareLevelArgs(orbx_extractor *ex, const uint8_t *src0, int nf, ptrdiff_t stride, ptrdiff_t frame_stride, int graphSlot)
{
    ex->curLevelArgs = nullptr;
    if (!kSmallSplit) return ORBX_OK;
    const int nl = ex->prm.nlevels, G = orbx_extractor::kGraphSlots;
    const size_t blk = orbxPyrFastArgsBytes(), slotBytes = blk * ORBX_MAX_LEVELS;
    int rc = ex->d_pfArgs.ensure(2 * G * slotBytes);
    if (rc != ORBX_OK) return rc;
    orbx_extractor::LevelArgsKey key;
    memset(&key, 0, sizeof key);
    key.src = src0; key.nframes = nf; key.stride = stride; key.frameStride = frame_stride; key.generation = ex->generation;
    key.iniTh = ex->prm.ini_th_fast; key.minTh = ex->prm.min_th_fast;
    for (int k = graphSlot >= 0 ? graphSlot : G; k < (graphSlot >= 0 ? graphSlot + 1 : 2 * G); ++k)
        if (memcmp(&key, &ex->pfArgsKey[k], sizeof key) == 0) { ex->curLevelArgs = ex->d_pfArgs.p + k * slotBytes; return ORBX_OK; }
    ORBX_SETUP_LOCK;
    int slot = graphSlot;
    if (slot < 0) { slot = G + ex->pfArgsNext; ex->pfArgsNext = (ex->pfArgsNext + 1) % G; }
    std::vector<uint8_t> host(slotBytes, 0);
    for (int l = 0; l < nl; ++l) {
        const OrbxLevelGeom &g = ex->geom[l];
        orbxFillPyrFastLevelArgs(host.data() + l * blk, nf, g, l == 0 ? src0 : ex->d_raw[l].p, l == 0 ? (int)stride : g.pitch,
                                 l == 0 ? (long long)frame_stride : g.frameStride, (const OrbxPfTile *)(ex->d_tabs.p + ex->tileOffsets[l]),
                                 ex->d_cand.p, ex->candFrameStride, ex->d_tileCount.p, ex->tilesPerFrame, ex->prm.ini_th_fast, ex->prm.min_th_fast);
    }
    HIP_TRY(hipStreamSynchronize(ex->stream));  // (work in flight may still read the slot that is replaced)
    HIP_TRY(hipMemcpy(ex->d_pfArgs.p + slot * slotBytes, host.data(), slotBytes, hipMemcpyHostToDevice));
    ex->pfArgsKey[slot] = key;
    ex->curLevelArgs = ex->d_pfArgs.p + slot * slotBytes;
    return ORBX_OK;
}

static hipError_t issueSmallBatch(orbx_extractor *ex, const uint8_t *src0, int nf, ptrdiff_t stride, ptrdiff_t frame_stride, int lap0, int lap1)
{
    const int nl = ex->prm.nlevels;
    hipStream_t sFront = ex->stream, sBlur = ex->stream2, sBack = ex->stream3;
    const OrbxTables tb = makeTables(ex);
    const int cap = ex->maxKeypoints;
    hipError_t e;
#define SB_TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
    static const bool noBlur = getenv("ORBX_NO_BLUR") != nullptr;
    // quadtree launches.  Default (1 branch): one launch for all levels behind the last k_pyr_fast.  2 branches: levels 0-1 start
    // behind k_pyr_fast(1) on a branch of their own, the rest behind the last level on a second one (4 branches: {0}, {1}, {2, 3},
    // {4..}) - what paid while the level-0 instance was a single 150-us wave.  The branches launch the level-synchronous kernel only;
    // the serial kernel for deferred instances (none on camera images) runs once, over all levels, after the join.
    static const int branchEnv = [] { const char *q = getenv("ORBX_SMALL_QT_BRANCHES"); return q ? atoi(q) : 1; }();
    int gEnd[4];  // last level of each group
    int ng = 0;
    if (branchEnv >= 4 && nl >= 5) { gEnd[0] = 0; gEnd[1] = 1; gEnd[2] = 3; gEnd[3] = nl - 1; ng = 4; }
    else if (branchEnv >= 2 && nl >= 3) { gEnd[0] = 1; gEnd[1] = nl - 1; ng = 2; }
    else { gEnd[0] = nl - 1; ng = 1; }
    auto quadtree = [&](hipStream_t st, int l0, int run, int which) {
        return orbxLaunchQuadtree(st, nf, nl, ex->geom.data(), ex->d_geom.p, ex->d_cand.p, ex->candFrameStride, ex->d_tileCount.p, ex->tilesPerFrame,
                                  ex->d_scratchA.p, ex->d_scratchB.p, ex->d_sel.p, ex->selFrameStride, ex->d_selCount.p, ex->maxQuota, l0, run,
                                  ex->d_qtNodes.p, ex->d_qtNodes.n, which);
    };
    int gi = 0, gBegin = 0;
    const bool split = kSmallSplit && ex->curLevelArgs != nullptr;
    if (split) {
        // the pyramid by a chain of resize-only launches (a few microseconds each), then FAST of all levels in one launch: the chain of
        // eight dependent fused launches was 106 of the ~195 us of a single-frame call; the blur starts right behind the resize chain
        static const bool pyrOneLaunch = [] { const char *q = getenv("ORBX_SMALL_PYR_CHAIN"); return q ? atoi(q) != 0 : true; }();
        bool pyramidDone = false;
        if (pyrOneLaunch && ex->chainTiles > 0) {  // the whole pyramid in one launch (k_pyr_chain)
            uint8_t *dp[ORBX_MAX_LEVELS] = {}; int dpitch[ORBX_MAX_LEVELS] = {}, lw[ORBX_MAX_LEVELS] = {}, lh[ORBX_MAX_LEVELS] = {};
            long long dfs[ORBX_MAX_LEVELS] = {};
            const int *xo[ORBX_MAX_LEVELS] = {}, *xa[ORBX_MAX_LEVELS] = {}, *yo[ORBX_MAX_LEVELS] = {}, *ya[ORBX_MAX_LEVELS] = {};
            for (int l = 0; l < nl; ++l) {
                const OrbxLevelGeom &g = ex->geom[l];
                lw[l] = g.w; lh[l] = g.h;
                if (l > 0) {
                    dp[l] = ex->d_raw[l].p; dpitch[l] = g.pitch; dfs[l] = g.frameStride;
                    const int *o = &ex->tabOffsets[(size_t)(l - 1) * 6];
                    xo[l] = ex->d_tabs.p + o[0]; xa[l] = ex->d_tabs.p + o[1]; yo[l] = ex->d_tabs.p + o[2]; ya[l] = ex->d_tabs.p + o[3];
                }
            }
            e = orbxLaunchPyrChain(sFront, nf, nl, src0, (int)stride, (long long)frame_stride, dp, dpitch, dfs, lw, lh, xo, xa, yo, ya,
                                   ex->d_tabs.p + ex->chainOffset, ex->chainTiles, ex->chainBufBytes, ex->chainTabBytes);
            if (e == hipSuccess) pyramidDone = true;
            else if (e != hipErrorInvalidValue) return e;
        }
        for (int l = 0; l + 1 < nl && !pyramidDone; ++l) {
            const OrbxLevelGeom &g = ex->geom[l], &d = ex->geom[l + 1];
            OrbxResizeTabs tabs;
            memset(&tabs, 0, sizeof tabs);
            const int *o = &ex->tabOffsets[(size_t)l * 6];
            tabs.xofs = ex->d_tabs.p + o[0]; tabs.xab = ex->d_tabs.p + o[1]; tabs.yofs = ex->d_tabs.p + o[2]; tabs.yab = ex->d_tabs.p + o[3];
            SB_TRY(orbxLaunchResizeLevel(sFront, nf, l == 0 ? src0 : ex->d_raw[l].p, l == 0 ? (int)stride : g.pitch,
                                         l == 0 ? (long long)frame_stride : g.frameStride, g.h, ex->d_raw[l + 1].p, d.w, d.h, d.pitch, d.frameStride, tabs));
        }
        SB_TRY(hipEventRecord(ex->evFront[0], sFront));  // the pyramid exists
        SB_TRY(orbxLaunchPyrFastLevels(sFront, nf, ex->geom.data(), nl, ex->curLevelArgs));
        SB_TRY(hipEventRecord(ex->evLevel[nl - 1], sFront));
        SB_TRY(hipStreamWaitEvent(ex->qtStream[0], ex->evLevel[nl - 1], 0));
        SB_TRY(quadtree(ex->qtStream[0], 0, nl, ORBX_QT_LEVEL_SYNC));
        SB_TRY(hipEventRecord(ex->evQt[0], ex->qtStream[0]));
        ng = 1;
    }
    for (int l = 0; l < nl && !split; ++l) {
        const OrbxLevelGeom &g = ex->geom[l];
        const uint8_t *src = l == 0 ? src0 : ex->d_raw[l].p;
        const int srcPitch = l == 0 ? (int)stride : g.pitch;
        const long long srcFS = l == 0 ? (long long)frame_stride : g.frameStride;
        uint8_t *dst = nullptr; int dw = 0, dh = 0, dPitch = 0; long long dFS = 0;
        OrbxResizeTabs tabs;
        memset(&tabs, 0, sizeof tabs);
        tabs.pfTiles = (const OrbxPfTile *)(ex->d_tabs.p + ex->tileOffsets[l]);
        if (l + 1 < nl) {
            const OrbxLevelGeom &d = ex->geom[l + 1];
            dst = ex->d_raw[l + 1].p; dw = d.w; dh = d.h; dPitch = d.pitch; dFS = d.frameStride;
            const int *o = &ex->tabOffsets[(size_t)l * 6];
            tabs.xofs = ex->d_tabs.p + o[0]; tabs.xab = ex->d_tabs.p + o[1]; tabs.yofs = ex->d_tabs.p + o[2];
            tabs.yab = ex->d_tabs.p + o[3]; tabs.rowBeg = ex->d_tabs.p + o[4]; tabs.colBeg = ex->d_tabs.p + o[5];
        }
        SB_TRY(orbxLaunchPyrFast(sFront, nf, g, src, srcPitch, srcFS, dst, dw, dh, dPitch, dFS, tabs, ex->d_cand.p, ex->candFrameStride,
                                 ex->d_tileCount.p, ex->tilesPerFrame, ex->prm.ini_th_fast, ex->prm.min_th_fast));
        if (gi < ng && l == gEnd[gi]) {
            SB_TRY(hipEventRecord(ex->evLevel[l], sFront));
            SB_TRY(hipStreamWaitEvent(ex->qtStream[gi], ex->evLevel[l], 0));
            SB_TRY(quadtree(ex->qtStream[gi], gBegin, l - gBegin + 1, ORBX_QT_LEVEL_SYNC));
            SB_TRY(hipEventRecord(ex->evQt[gi], ex->qtStream[gi]));
            gBegin = l + 1;
            ++gi;
        }
    }
    if (!split) SB_TRY(hipEventRecord(ex->evFront[0], sFront));
    // blur stream: every level, once the pyramid exists (short kernels; they run beside the quadtree)
    SB_TRY(hipStreamWaitEvent(sBlur, ex->evFront[0], 0));
    bool blurDone = noBlur;
    static const bool blurOneLaunch = [] { const char *q = getenv("ORBX_SMALL_BLUR_LEVELS"); return q ? atoi(q) != 0 : true; }();
    if (!blurDone && blurOneLaunch) {  // all levels in one launch (eight short launches in a row were the longest branch of the call)
        int lw[ORBX_MAX_LEVELS], lh[ORBX_MAX_LEVELS], sp[ORBX_MAX_LEVELS], dp[ORBX_MAX_LEVELS];
        long long sfs[ORBX_MAX_LEVELS], dfs[ORBX_MAX_LEVELS];
        const uint8_t *sp0[ORBX_MAX_LEVELS]; uint8_t *dp0[ORBX_MAX_LEVELS];
        for (int l = 0; l < nl; ++l) {
            const OrbxLevelGeom &g = ex->geom[l];
            lw[l] = g.w; lh[l] = g.h; sp0[l] = l == 0 ? src0 : ex->d_raw[l].p; sp[l] = l == 0 ? (int)stride : g.pitch;
            sfs[l] = l == 0 ? (long long)frame_stride : g.frameStride; dp0[l] = ex->d_blur[l].p; dp[l] = blurTilesX(g); dfs[l] = blurFrameBytes(g);
        }
        e = orbxLaunchBlurLevels(sBlur, nf, nl, lw, lh, sp0, sp, sfs, dp0, dp, dfs, ex->taps);
        if (e == hipSuccess) blurDone = true;
        else if (e != hipErrorNotSupported) return e;
    }
    for (int l = 0; l < nl && !blurDone; ++l) {
        const OrbxLevelGeom &g = ex->geom[l];
        SB_TRY(orbxLaunchBlur(sBlur, nf, g.w, g.h, l == 0 ? src0 : ex->d_raw[l].p, l == 0 ? (int)stride : g.pitch,
                              l == 0 ? (long long)frame_stride : g.frameStride, ex->d_blur[l].p, blurTilesX(g), blurFrameBytes(g), ex->taps));
    }
    SB_TRY(hipEventRecord(ex->evBlur[0], sBlur));
    for (int k = 0; k < ng; ++k) SB_TRY(hipStreamWaitEvent(sBack, ex->evQt[k], 0));
    SB_TRY(quadtree(sBack, 0, nl, ORBX_QT_SERIAL));
    SB_TRY(orbxLaunchPack(sBack, nf, tb, ex->d_sel.p, ex->selFrameStride, ex->d_selCount.p, lap0, lap1, ex->d_kps.p, cap, ex->d_workKey.p,
                          ex->d_workPos.p, ex->d_n.p, ex->d_mono.p));
    SB_TRY(hipStreamWaitEvent(sBack, ex->evBlur[0], 0));
    OrbxLevelPtrs lv;
    memset(&lv, 0, sizeof lv);
    for (int l = 0; l < nl; ++l) {
        const OrbxLevelGeom &g = ex->geom[l];
        lv.raw[l] = l == 0 ? src0 : ex->d_raw[l].p;
        lv.rawPitch[l] = l == 0 ? (int)stride : g.pitch;
        lv.rawFrameStride[l] = l == 0 ? (long long)frame_stride : g.frameStride;
        lv.blur[l] = ex->d_blur[l].p; lv.blurPitch[l] = blurTilesX(g) * 128; lv.blurFrameStride[l] = blurFrameBytes(g);
        lv.w[l] = g.w; lv.h[l] = g.h;
    }
    SB_TRY(orbxLaunchOrientDesc(sBack, nf, tb, lv, ex->d_workKey.p, ex->d_workPos.p, ex->d_n.p, ex->d_kps.p, ex->d_desc.p, cap));
    SB_TRY(hipEventRecord(ex->evJoin, sBack));
    SB_TRY(hipStreamWaitEvent(sFront, ex->evJoin, 0));
#undef SB_TRY
    return hipSuccess;
}

static int runSmallBatch(orbx_extractor *ex, const uint8_t *src0, int nf, int w, int h, ptrdiff_t stride, ptrdiff_t frame_stride, int lap0, int lap1,
                         bool useGraph)
{
    if (!useGraph) {
        const int rcl = prepareLevelArgs(ex, src0, nf, stride, frame_stride, -1);
        if (rcl != ORBX_OK) return rcl;
        HIP_TRY(issueSmallBatch(ex, src0, nf, stride, frame_stride, lap0, lap1));
        return ORBX_OK;
    }
    orbx_extractor::GraphKey key;
    memset(&key, 0, sizeof key);  // (padding bytes: the keys are compared with memcmp)
    key.src = src0; key.nframes = nf; key.w = w; key.h = h; key.stride = (long long)stride; key.frameStride = (long long)frame_stride;
    key.lap0 = lap0; key.lap1 = lap1; key.generation = ex->generation;
    int slot = -1;
    for (int k = 0; k < orbx_extractor::kGraphSlots; ++k)
        if (ex->graphExec[k] && memcmp(&key, &ex->graphKey[k], sizeof key) == 0) { slot = k; break; }
    const bool hit = slot >= 0;
    if (!hit) {
        ORBX_SETUP_LOCK;
        slot = 0;  // an empty slot, else the least recently used one
        for (int k = 0; k < orbx_extractor::kGraphSlots; ++k) {
            if (!ex->graphExec[k]) { slot = k; break; }
            if (ex->graphUse[k] < ex->graphUse[slot]) slot = k;
        }
        if (ex->graphExec[slot]) {
            // (a replaced sequence may still be running: its launch was the last use of the exec object)
            HIP_TRY(hipStreamSynchronize(ex->stream));
            (void)hipGraphExecDestroy(ex->graphExec[slot]);
            ex->graphExec[slot] = nullptr;
        }
    }
    // the per-level argument block of THIS slot (a hit finds it unchanged; before any capture: a rewrite copies synchronously).  The
    // blocks used to be a cache of their own with a replacement order of its own: a graph that stayed cached while its block was
    // handed to another source then read that source's frame (six buffers visited in turn: wrong keypoints for the third one).
    {
        const int rcl = prepareLevelArgs(ex, src0, nf, stride, frame_stride, slot);
        if (rcl != ORBX_OK) return rcl;
    }
    if (!hit) {
        ORBX_SETUP_LOCK;
        HIP_TRY(hipStreamBeginCapture(ex->stream, hipStreamCaptureModeRelaxed));
        const hipError_t ie = issueSmallBatch(ex, src0, nf, stride, frame_stride, lap0, lap1);
        hipGraph_t graph = nullptr;
        const hipError_t ce = hipStreamEndCapture(ex->stream, &graph);  // (always leave capture mode)
        if (ie != hipSuccess || ce != hipSuccess || !graph) {
            if (graph) (void)hipGraphDestroy(graph);
            return setError(ORBX_ERR_HIP, "capturing the launch sequence failed: %s", hipGetErrorString(ie != hipSuccess ? ie : ce));
        }
        const hipError_t ge = hipGraphInstantiate(&ex->graphExec[slot], graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ge != hipSuccess) { ex->graphExec[slot] = nullptr; return setError(ORBX_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(ge)); }
        memcpy(&ex->graphKey[slot], &key, sizeof key);
    }
    ex->graphUse[slot] = ++ex->graphClock;
    HIP_TRY(hipGraphLaunch(ex->graphExec[slot], ex->stream));
    return ORBX_OK;
}

static HostPipe *pendingPipeOf(orbx_extractor *ex);
static int runBatch(orbx_extractor *ex, const void *d_imgs, int nframes, int w, int h, ptrdiff_t stride, ptrdiff_t frame_stride, int lap0,
                    int lap1, const HostFrames *host);

extern "C" int orbx_extract_batch_device(orbx_extractor *ex, const void *d_imgs, int nframes, int w, int h, ptrdiff_t stride,
                                         ptrdiff_t frame_stride, int lap0, int lap1)
{
    ORBX_API_ENTRY;
    return runBatch(ex, d_imgs, nframes, w, h, stride, frame_stride, lap0, lap1, nullptr);
}

static int runBatch(orbx_extractor *ex, const void *d_imgs, int nframes, int w, int h, ptrdiff_t stride, ptrdiff_t frame_stride, int lap0,
                    int lap1, const HostFrames *host)
{
    if (!ex || !d_imgs) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_device: null argument");
    if (ex->nPending > 0 && !(host && host->pipe && host->pipe == pendingPipeOf(ex)))
        return setError(ORBX_ERR_INVALID_ARG, "a batch submitted with orbx_extract_batch_submit is still pending on this handle: call orbx_extract_batch_wait first");
    if (nframes <= 0 || w <= 0 || h <= 0 || stride < w) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_device: bad shape");
    if (nframes > ex->prm.max_batch) return setError(ORBX_ERR_CAPACITY, "batch %d exceeds max_batch %d", nframes, ex->prm.max_batch);
    // max_width / max_height are the largest image the handle will see: the per-level capacities of the selection and output buffers
    // were fixed from them at orbx_create
    if (w > ex->prm.max_width || h > ex->prm.max_height)
        return setError(ORBX_ERR_CAPACITY, "image %dx%d exceeds the handle's max_width x max_height = %dx%d", w, h, ex->prm.max_width, ex->prm.max_height);
    HIP_TRY(hipSetDevice(ex->prm.device));
    int rc = computeGeometry(ex, w, h);
    if (rc != ORBX_OK) return rc;
    if ((rc = ensureBuffers(ex)) != ORBX_OK) return rc;
    const int nl = ex->prm.nlevels;
    const uint8_t *src0 = (const uint8_t *)d_imgs;
    ex->lastSrc = src0; ex->lastSrcPitch = (int)stride; ex->lastSrcFrameStride = (long long)frame_stride; ex->lastFrames = nframes;
    ex->batchSerial = ++gBatchSerial;
    hipStream_t sFront = ex->stream, sBlur = ex->stream2, sBack = ex->stream3;
    const OrbxTables tb = makeTables(ex);
    const int cap = ex->maxKeypoints;

    // (an explicit batch-schedule switch - the A/B experiments of DESIGN 6 - keeps the batch schedule for small batches too)
    static const int graphMode = [] {
        if (getenv("ORBX_QT_PER_LEVEL") || getenv("ORBX_EARLY_BLUR") || getenv("ORBX_SUBBATCHES")) return 0;
        const char *e = getenv("ORBX_GRAPH");
        return e ? (atoi(e) == 2 ? 2 : (atoi(e) != 0)) : 1;  // 1 (default): captured graph; 2: the same schedule launched directly; 0: batch schedule
    }();
    if (graphMode && !ex->profiling && nframes <= ORBX_GRAPH_MAX_FRAMES && smallBatchPrepare(ex)) {
        if (host) {  // stage the frames on the front stream, right in front of the graph
            const int rcs = stageHostFrames(ex, (uint8_t *)const_cast<void *>(d_imgs), stride, frame_stride, host, 0, nframes, nframes, w, h, sFront);
            if (rcs != ORBX_OK) return rcs;
        }
        return runSmallBatch(ex, src0, nframes, w, h, stride, frame_stride, lap0, lap1, graphMode == 1);
    }

    // sub-batch size: a multiple of 8 frames (one frame per XCD in flight), at least 64 frames
    static const int maxSub = [] { const char *e = getenv("ORBX_SUBBATCHES"); int v = e ? atoi(e) : 1; return std::min(std::max(v, 1), ORBX_MAX_SUB); }();
    int nsub = std::min(maxSub, std::max(1, nframes / 64));
    // frames arriving from the host: sub-batches of >= 128 frames, so that the copy of the next one runs under this one's kernels
    // (up to 8 sub-batches of at least 16 MB of pixels and 8 frames: what is not hidden is the first sub-batch's copy and the last one's
    // kernels and results)
    // Round 5 (profiles/r05_submit_wait.txt, 1024 VGA frames): one call at a time 5 sub-batches (6.84 ms; 8: 7.4 - a 128-frame sub-batch keeps the
    // GPU busy for 0.65 ms, eight of them 5.2 ms against 3.0 ms for the batch in one piece); a SUBMITTED batch 4 (its head and tail run under
    // the neighbouring batches: 2 - 5 sub-batches 174 - 176 k frames/s, 6: 167 k, 8: 137 k).
    static const int hostSubEnv = [] { const char *e = getenv("ORBX_HOST_SUBBATCHES"); return e ? std::min(std::max(atoi(e), 1), ORBX_MAX_SUB) : 0; }();
    const int hostSub = hostSubEnv ? hostSubEnv : (host && host->pipe && host->pipe == pendingPipeOf(ex) ? 4 : 5);
    if (host) nsub = std::max(nsub, std::min(hostSub, std::max(1, (int)std::min((long long)nframes / 8, (long long)nframes * w * h / (16 << 20)))));
    const int subSize = ((nframes + nsub - 1) / nsub + 7) / 8 * 8;
    nsub = (nframes + subSize - 1) / subSize;
    HostPipe *pipe = host ? host->pipe : nullptr;
    if (pipe) {
        pipe->nsub = nsub;
        for (int sb = 0; sb < nsub; ++sb) { pipe->f0[sb] = sb * subSize; pipe->nf[sb] = std::min(subSize, nframes - sb * subSize); }
    }

    // 0: one quadtree launch for all levels; 1: one launch per level, next to k_pyr_fast of the following level; k >= 2: groups
    // of k - 1 levels per launch after k_pyr_fast (LDS per instance sized by the group's largest quota)
    // Default: pairs of levels for large batches (1024 frames: 0 4.34 ms, 2 4.40, 3 4.17, 4 4.20, 5 4.29); one launch for small ones,
    // where the launches of a stream run one after the other and each costs an instance's latency (1 frame: 0.37 ms per call with
    // one launch, 0.67 ms with pairs; 512 frames: 2.05 vs 2.10 ms; tools/experiments/qt_mode_sweep.sh)
    static const int qtPerLevelEnv = [] { const char *e = getenv("ORBX_QT_PER_LEVEL"); return e ? atoi(e) : -1; }();
    // Many features per pixel (512 x 512 / 1500: 5.7 per 1000 pixels; VGA / 1000: 3.3): the quadtree, not the blur beside it, is the longer
    // branch - groups of three levels, dealt over two streams so that the thin tails of the launches overlap (3.60 -> 3.36 ms per
    // 1024 frames; the same schedule costs the VGA workload 1 %, because there it takes LDS from the blur, which is its longer branch;
    // tools/experiments/qt_streams_sweep.sh)
    const bool qtHeavy = (double)ex->prm.nfeatures > 0.0045 * (double)w * (double)h;
    const int qtPerLevel = qtPerLevelEnv >= 0 ? qtPerLevelEnv : (subSize >= 640 ? (qtHeavy ? 4 : 3) : 0);
    static const bool earlyBlur = [] { const char *e = getenv("ORBX_EARLY_BLUR"); return e ? atoi(e) != 0 : false; }();  // measured: 6.37 ms with early blur vs 6.02 ms with blur || quadtree
    hipEvent_t *prof = ex->profiling ? &ex->events[(size_t)(ex->profCount % PROF_RING) * PROF_EVENTS] : nullptr;
    if (earlyBlur) {
        // the blur stream must not start on this batch's level 0 before the previous batch is done with the blur buffers
        HIP_TRY(hipEventRecord(ex->evJoin, sFront));
        HIP_TRY(hipStreamWaitEvent(sBlur, ex->evJoin, 0));
    }
    for (int sb = 0; sb < nsub; ++sb) {
        const int f0 = sb * subSize, nf = std::min(subSize, nframes - f0);
        if (host) {
            // stage this sub-batch's frames
            uint8_t *dIn = (uint8_t *)const_cast<void *>(d_imgs);
            if (sb == 0) {  // the staging buffer is free once everything ordered before this call on the front stream is done
                HIP_TRY(hipEventRecord(ex->evIn, sFront));
                // (a second batch in flight - orbx_extract_batch_submit with one pending - stages into the OTHER slot's buffer, whose last
                // user has been waited for: its copies start at once, under the pending batch's kernels)
                if (!(ex->nPending == 2 && pipe && pipe == pendingPipeOf(ex))) HIP_TRY(hipStreamWaitEvent(ex->streamCopy, ex->evIn, 0));
                if (pipe && pipe->feeder) {
                    // pageable frames: the blocking copies run on a helper thread, one sub-batch after the other (joined by ~HostPipe)
                    pipe->thread = std::thread([=]() {
                        ORBX_API_ENTRY;
                        int rcf = hipSetDevice(ex->prm.device) == hipSuccess ? ORBX_OK : setError(ORBX_ERR_HIP, "hipSetDevice failed in the copy thread");
                        for (int k = 0; k < nsub; ++k) {
                            if (rcf == ORBX_OK && hostTimeline()) (void)hipEventRecord(ex->slots[ex->curSlot].evCopy0[k], ex->streamCopy);
                            if (rcf == ORBX_OK) rcf = stageHostFrames(ex, dIn, stride, frame_stride, host, pipe->f0[k], pipe->nf[k], nframes, w, h, ex->streamCopy);
                            if (rcf == ORBX_OK && hipEventRecord(ex->slots[ex->curSlot].evCopy[k], ex->streamCopy) != hipSuccess) rcf = setError(ORBX_ERR_HIP, "hipEventRecord failed in the copy thread");
                            std::lock_guard<std::mutex> lk(pipe->mu);
                            if (rcf != ORBX_OK && pipe->feederRc == ORBX_OK) { pipe->feederRc = rcf; pipe->feederErr = g_lastError; }
                            pipe->staged = k + 1;
                            pipe->cv.notify_all();
                        }
                    });
                }
            }
            if (pipe && pipe->feeder) {
                std::unique_lock<std::mutex> lk(pipe->mu);
                pipe->cv.wait(lk, [&] { return pipe->staged > sb; });
                if (pipe->feederRc != ORBX_OK) { g_lastError = pipe->feederErr; return pipe->feederRc; }
            } else {
                if (hostTimeline()) (void)hipEventRecord(ex->slots[ex->curSlot].evCopy0[sb], ex->streamCopy);
                const int rcs = stageHostFrames(ex, dIn, stride, frame_stride, host, f0, nf, nframes, w, h, ex->streamCopy);
                if (rcs != ORBX_OK) return rcs;
                HIP_TRY(hipEventRecord(ex->slots[ex->curSlot].evCopy[sb], ex->streamCopy));
            }
            HIP_TRY(hipStreamWaitEvent(sFront, ex->slots[ex->curSlot].evCopy[sb], 0));
            HIP_TRY(hipStreamWaitEvent(sBlur, ex->slots[ex->curSlot].evCopy[sb], 0));
        }
        // ---- front: fused pyramid + FAST, one launch per level -----------------------------------------------
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 0) * 2], sFront));
        // The eight launches of the chain depend on each other per FRAME only, so the chain can be cut into `nParts` groups of frames, each
        // a chain on a stream of its own (optionally staggered by one level) whose kernels fill the draining tail of the other chain's
        // launches.  Measured (1024 frames, tools/experiments/front_parts_sweep.sh): 1 part 1.883 ms, 2 parts 1.836 (staggered 1.881),
        // 3 parts 1.956, 4 parts 1.935 - the tails are not where the time goes; default 1 (one chain: per-launch times stay readable).
        static const int frontPartsEnv = [] { const char *e = getenv("ORBX_FRONT_PARTS"); return e ? atoi(e) : 1; }();
        static const bool frontStagger = [] { const char *e = getenv("ORBX_FRONT_STAGGER"); return e ? atoi(e) != 0 : false; }();
        int nParts = qtPerLevel == 1 || earlyBlur || nf < 256 ? 1 : std::min(std::max(frontPartsEnv, 1), 4);
        for (int k = 1; k < nParts; ++k) {
            if (!ex->qtStream[k] && hipStreamCreateWithFlags(&ex->qtStream[k], hipStreamNonBlocking) != hipSuccess) { nParts = 1; break; }
            if (!ex->evPart[k] && hipEventCreateWithFlags(&ex->evPart[k], hipEventDisableTiming) != hipSuccess) { nParts = 1; break; }
        }
        for (int k = 0; k < nParts; ++k)
            if (!ex->evStag[k] && hipEventCreateWithFlags(&ex->evStag[k], hipEventDisableTiming) != hipSuccess) { nParts = 1; break; }
        const int partSize = ((nf + nParts - 1) / nParts + 7) / 8 * 8;
        if (nParts > 1) HIP_TRY(hipEventRecord(ex->evStag[0], sFront));  // what the front stream is ordered after: every part waits for it
        for (int part = 0; part < nParts; ++part) {
        const int pf0 = part * partSize, pnf = std::min(partSize, nf - pf0);
        if (pnf <= 0) break;
        hipStream_t sPart = part == 0 ? sFront : ex->qtStream[part];
        if (part > 0) HIP_TRY(hipStreamWaitEvent(sPart, ex->evStag[frontStagger ? part : 0], 0));
        const int f0p = f0 + pf0;
        for (int l = 0; l < nl; ++l) {
            const OrbxLevelGeom &g = ex->geom[l];
            const uint8_t *src = l == 0 ? src0 + (size_t)f0p * frame_stride : ex->d_raw[l].p + (size_t)f0p * g.frameStride;
            const int srcPitch = l == 0 ? (int)stride : g.pitch;
            const long long srcFS = l == 0 ? (long long)frame_stride : g.frameStride;
            uint8_t *dst = nullptr; int dw = 0, dh = 0, dPitch = 0; long long dFS = 0;
            OrbxResizeTabs tabs;
            memset(&tabs, 0, sizeof tabs);
            tabs.pfTiles = (const OrbxPfTile *)(ex->d_tabs.p + ex->tileOffsets[l]);
            if (l + 1 < nl) {
                const OrbxLevelGeom &d = ex->geom[l + 1];
                dst = ex->d_raw[l + 1].p + (size_t)f0p * d.frameStride; dw = d.w; dh = d.h; dPitch = d.pitch; dFS = d.frameStride;
                const int *o = &ex->tabOffsets[(size_t)l * 6];
                tabs.xofs = ex->d_tabs.p + o[0]; tabs.xab = ex->d_tabs.p + o[1]; tabs.yofs = ex->d_tabs.p + o[2];
                tabs.yab = ex->d_tabs.p + o[3]; tabs.rowBeg = ex->d_tabs.p + o[4]; tabs.colBeg = ex->d_tabs.p + o[5];
            }
            HIP_TRY(orbxLaunchPyrFast(sPart, pnf, g, src, srcPitch, srcFS, dst, dw, dh, dPitch, dFS, tabs,
                                      ex->d_cand.p + (size_t)f0p * ex->candFrameStride, ex->candFrameStride,
                                      ex->d_tileCount.p + (size_t)f0p * ex->tilesPerFrame, ex->tilesPerFrame, ex->prm.ini_th_fast,
                                      ex->prm.min_th_fast));
            if (l == 0 && frontStagger && part + 1 < nParts) HIP_TRY(hipEventRecord(ex->evStag[part + 1], sPart));
            if (qtPerLevel == 1) {
                // the quadtree of this level only needs this level's candidates: it runs on the back stream next to the
                // pyramid + FAST launches of the smaller levels (latency-bound single waves under a throughput-bound kernel)
                HIP_TRY(hipEventRecord(ex->evLevel[l], sFront));
                HIP_TRY(hipStreamWaitEvent(sBack, ex->evLevel[l], 0));
                HIP_TRY(orbxLaunchQuadtree(sBack, nf, nl, ex->geom.data(), ex->d_geom.p, ex->d_cand.p + (size_t)f0 * ex->candFrameStride,
                                           ex->candFrameStride, ex->d_tileCount.p + (size_t)f0 * ex->tilesPerFrame, ex->tilesPerFrame,
                                           ex->d_scratchA.p + (size_t)f0 * ex->candFrameStride, ex->d_scratchB.p + (size_t)f0 * ex->candFrameStride,
                                           ex->d_sel.p + (size_t)f0 * ex->selFrameStride, ex->selFrameStride,
                                           ex->d_selCount.p + (size_t)f0 * nl, ex->maxQuota, l, 1, ex->d_qtNodes.p, ex->d_qtNodes.n));
            }
            if (l + 1 < nl && earlyBlur) {
                // level l + 1 now exists: its GaussianBlur may start on the blur stream while the front stream goes on
                // with pyr_fast(l + 1) (the fused kernel is barrier-bound, the blur fills its idle issue slots)
                HIP_TRY(hipEventRecord(ex->evLevel[l + 1], sFront));
            }
        }
        if (part > 0) HIP_TRY(hipEventRecord(ex->evPart[part], sPart));
        }
        for (int part = 1; part < nParts; ++part) HIP_TRY(hipStreamWaitEvent(sFront, ex->evPart[part], 0));
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 0) * 2 + 1], sFront));
        HIP_TRY(hipEventRecord(ex->evFront[sb], sFront));

        // ---- blur stream: GaussianBlur of every level (needs only the pyramid) ---------------------------------
        if (!earlyBlur) HIP_TRY(hipStreamWaitEvent(sBlur, ex->evFront[sb], 0));
        static const bool noBlur = getenv("ORBX_NO_BLUR") != nullptr;  // timing experiments only
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 2) * 2], sBlur));
        for (int l = 0; l < nl && !noBlur; ++l) {
            const OrbxLevelGeom &g = ex->geom[l];
            const uint8_t *src = l == 0 ? src0 + (size_t)f0 * frame_stride : ex->d_raw[l].p + (size_t)f0 * g.frameStride;
            const int srcPitch = l == 0 ? (int)stride : g.pitch;
            const long long srcFS = l == 0 ? (long long)frame_stride : g.frameStride;
            if (earlyBlur && l > 0) HIP_TRY(hipStreamWaitEvent(sBlur, ex->evLevel[l], 0));
            HIP_TRY(orbxLaunchBlur(sBlur, nf, g.w, g.h, src, srcPitch, srcFS, ex->d_blur[l].p + (size_t)f0 * blurFrameBytes(g), blurTilesX(g),
                                   blurFrameBytes(g), ex->taps));
        }
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 2) * 2 + 1], sBlur));
        HIP_TRY(hipEventRecord(ex->evBlur[sb], sBlur));

        // ---- back stream: quadtree -> pack -> (join blur) -> orientation + descriptors -----------------------------
        HIP_TRY(hipStreamWaitEvent(sBack, ex->evFront[sb], 0));
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 1) * 2], sBack));
        if (qtPerLevel != 1) {
            // one launch per group of `qtPerLevel` levels (0: all levels in one launch): LDS per instance follows the quota.  The groups
            // are independent (a level's instances read that level's candidates only), so each goes on a stream of its own: a launch of
            // latency-bound single-wave instances ends in a long thin tail, and behind each other on one stream the tails add up.
            const int step = qtPerLevel >= 2 ? qtPerLevel - 1 : nl;
            static const int qtStreamsEnv = [] { const char *e = getenv("ORBX_QT_STREAMS"); return e ? std::min(std::max(atoi(e), 0), 3) : -1; }();
            const int qtWant = qtStreamsEnv >= 0 ? qtStreamsEnv : (qtHeavy ? 1 : 0);
            const int qtExtra = qtWant > 0 && step < nl && smallBatchPrepare(ex) != 0 ? qtWant : 0;  // streams beside the back stream
            int gi = 0;
            bool used[4] = {false, false, false, false};
            for (int l0 = 0; l0 < nl; l0 += step, ++gi) {
                const int si = gi % (qtExtra + 1);  // groups dealt round-robin: 0 = the back stream itself
                hipStream_t sq = si ? ex->qtStream[si] : sBack;
                if (si && !used[si]) { HIP_TRY(hipStreamWaitEvent(sq, ex->evFront[sb], 0)); used[si] = true; }
                HIP_TRY(orbxLaunchQuadtree(sq, nf, nl, ex->geom.data(), ex->d_geom.p, ex->d_cand.p + (size_t)f0 * ex->candFrameStride,
                                           ex->candFrameStride, ex->d_tileCount.p + (size_t)f0 * ex->tilesPerFrame, ex->tilesPerFrame,
                                           ex->d_scratchA.p + (size_t)f0 * ex->candFrameStride, ex->d_scratchB.p + (size_t)f0 * ex->candFrameStride,
                                           ex->d_sel.p + (size_t)f0 * ex->selFrameStride, ex->selFrameStride,
                                           ex->d_selCount.p + (size_t)f0 * nl, ex->maxQuota, l0, std::min(step, nl - l0), ex->d_qtNodes.p, ex->d_qtNodes.n));
            }
            for (int k = 1; k < 4; ++k)
                if (used[k]) { HIP_TRY(hipEventRecord(ex->evQt[k], ex->qtStream[k])); HIP_TRY(hipStreamWaitEvent(sBack, ex->evQt[k], 0)); }
        }
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 1) * 2 + 1], sBack));
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 3) * 2], sBack));
        HIP_TRY(orbxLaunchPack(sBack, nf, tb, ex->d_sel.p + (size_t)f0 * ex->selFrameStride, ex->selFrameStride,
                               ex->d_selCount.p + (size_t)f0 * nl, lap0, lap1, ex->d_kps.p + (size_t)f0 * cap, cap,
                               ex->d_workKey.p + (size_t)f0 * cap, ex->d_workPos.p + (size_t)f0 * cap, ex->d_n.p + f0, ex->d_mono.p + f0));
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 3) * 2 + 1], sBack));
        HIP_TRY(hipStreamWaitEvent(sBack, ex->evBlur[sb], 0));
        OrbxLevelPtrs lv;
        memset(&lv, 0, sizeof lv);
        for (int l = 0; l < nl; ++l) {
            const OrbxLevelGeom &g = ex->geom[l];
            lv.raw[l] = l == 0 ? src0 + (size_t)f0 * frame_stride : ex->d_raw[l].p + (size_t)f0 * g.frameStride;
            lv.rawPitch[l] = l == 0 ? (int)stride : g.pitch;
            lv.rawFrameStride[l] = l == 0 ? (long long)frame_stride : g.frameStride;
            lv.blur[l] = ex->d_blur[l].p + (size_t)f0 * blurFrameBytes(g); lv.blurPitch[l] = blurTilesX(g) * 128; lv.blurFrameStride[l] = blurFrameBytes(g);
            lv.w[l] = g.w; lv.h[l] = g.h;
        }
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 4) * 2], sBack));
        HIP_TRY(orbxLaunchOrientDesc(sBack, nf, tb, lv, ex->d_workKey.p + (size_t)f0 * cap, ex->d_workPos.p + (size_t)f0 * cap,
                                     ex->d_n.p + f0, ex->d_kps.p + (size_t)f0 * cap, ex->d_desc.p + (size_t)f0 * cap * 32, cap));
        if (prof) HIP_TRY(hipEventRecord(prof[(sb * ORBX_NUM_STAGES + 4) * 2 + 1], sBack));
        if (pipe) {  // this sub-batch's results start for the host; those of earlier ones that have landed go to the caller's arrays
            int rcp = pipeSendResults(ex, pipe, sb);
            if (rcp == ORBX_OK) rcp = pipeDrain(ex, host, false, sb);
            if (rcp != ORBX_OK) return rcp;
        }
    }
    // join: everything of this batch is ordered before whatever follows on the front stream
    HIP_TRY(hipEventRecord(ex->evJoin, sBack));
    HIP_TRY(hipStreamWaitEvent(sFront, ex->evJoin, 0));
    if (prof) { ex->profSubs[ex->profCount % PROF_RING] = nsub; ++ex->profCount; }
    return ORBX_OK;
}

extern "C" int orbx_sync(orbx_extractor *ex)
{
    ORBX_API_ENTRY;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_sync: null handle");
    HIP_TRY(hipSetDevice(ex->prm.device));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    return ORBX_OK;
}

extern "C" int orbx_device_outputs(orbx_extractor *ex, void **d_kps, void **d_desc, void **d_n, void **d_mono)
{
    ORBX_API_ENTRY;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_device_outputs: null handle");
    if (d_kps) *d_kps = ex->d_kps.p;
    if (d_desc) *d_desc = ex->d_desc.p;
    if (d_n) *d_n = ex->d_n.p;
    if (d_mono) *d_mono = ex->d_mono.p;
    return ORBX_OK;
}

extern "C" int orbx_download_outputs(orbx_extractor *ex, int nframes, orbx_keypoint *kps, uint8_t *desc, int capacity,
                                     int32_t *n_out, int32_t *mono_out)
{
    ORBX_API_ENTRY;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_download_outputs: null handle");
    if (nframes <= 0 || nframes > ex->prm.max_batch) return setError(ORBX_ERR_INVALID_ARG, "orbx_download_outputs: bad nframes");
    HIP_TRY(hipSetDevice(ex->prm.device));
    std::vector<int32_t> n(nframes), mono(nframes);
    HIP_TRY(hipMemcpyAsync(n.data(), ex->d_n.p, nframes * sizeof(int32_t), hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipMemcpyAsync(mono.data(), ex->d_mono.p, nframes * sizeof(int32_t), hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    const int cap = ex->maxKeypoints;
    for (int f = 0; f < nframes; ++f)
        if (n[f] > capacity) return setError(ORBX_ERR_CAPACITY, "frame %d has %d keypoints, output capacity is %d", f, n[f], capacity);
    if (capacity == cap && nframes > 1) {
        // same per-frame capacity on both sides: two bulk copies (rows beyond n[f] are unspecified)
        if (kps) HIP_TRY(hipMemcpyAsync(kps, ex->d_kps.p, (size_t)nframes * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToHost, ex->stream));
        if (desc) HIP_TRY(hipMemcpyAsync(desc, ex->d_desc.p, (size_t)nframes * cap * 32, hipMemcpyDeviceToHost, ex->stream));
    } else {
        for (int f = 0; f < nframes; ++f) {
            if (kps && n[f] > 0)
                HIP_TRY(hipMemcpyAsync(kps + (size_t)f * capacity, ex->d_kps.p + (size_t)f * cap, (size_t)n[f] * sizeof(orbx_keypoint),
                                       hipMemcpyDeviceToHost, ex->stream));
            if (desc && n[f] > 0)
                HIP_TRY(hipMemcpyAsync(desc + (size_t)f * capacity * 32, ex->d_desc.p + (size_t)f * cap * 32, (size_t)n[f] * 32,
                                       hipMemcpyDeviceToHost, ex->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(ex->stream));
    for (int f = 0; f < nframes; ++f) {
        if (n_out) n_out[f] = n[f];
        if (mono_out) mono_out[f] = mono[f];
    }
    return ORBX_OK;
}

// Small batches: everything (counts, keypoints, descriptors at full capacity) in four asynchronous copies into page-locked staging
// and ONE synchronisation, instead of counts first and exact-size copies afterwards (two round trips).
static int downloadSmall(orbx_extractor *ex, const HostFrames *host, int nframes, orbx_keypoint *kps, uint8_t *desc, int capacity, int32_t *n_out,
                         int32_t *mono_out)
{
    const int cap = ex->maxKeypoints;
    const size_t bN = (size_t)nframes * sizeof(int32_t), bK = (size_t)nframes * cap * sizeof(orbx_keypoint), bD = (size_t)nframes * cap * 32;
    const size_t need = 2 * bN + bK + bD + 64;
    if (ex->h_outBytes < need) {
        ORBX_SETUP_LOCK;
        if (ex->h_out) { (void)hipHostFree(ex->h_out); ex->h_out = nullptr; ex->h_outBytes = 0; }
        HIP_TRY(hipHostMalloc((void **)&ex->h_out, need, hipHostMallocDefault));
        ex->h_outBytes = need;
    }
    int32_t *hn = (int32_t *)ex->h_out, *hm = hn + nframes;
    const size_t oK = (2 * bN + 15) & ~(size_t)15;
    orbx_keypoint *hk = (orbx_keypoint *)(ex->h_out + oK);
    uint8_t *hd = (uint8_t *)(hk + (size_t)nframes * cap);
    // the four result arrays packed by one launch and fetched by ONE copy (a copy costs ~6 us on the device and as much again on the
    // host whatever its size: four of them were a tenth of a single-frame call)
    {
        int rcp = ex->d_outPack.ensure(need);
        if (rcp != ORBX_OK) return rcp;
        OrbxCopyPieces pc;
        pc.n = 0;
        auto piece = [&](size_t off, const void *src, size_t bytes) { pc.dst[pc.n] = ex->d_outPack.p + off; pc.src[pc.n] = src; pc.bytes[pc.n] = (unsigned)bytes; ++pc.n; };
        piece(0, ex->d_n.p, bN);
        piece(bN, ex->d_mono.p, bN);
        if (kps) piece(oK, ex->d_kps.p, bK);
        if (desc) piece(oK + bK, ex->d_desc.p, bD);
        HIP_TRY(orbxLaunchCopyPieces(ex->stream, pc));
        const size_t total = desc ? oK + bK + bD : (kps ? oK + bK : 2 * bN);
        HIP_TRY(hipMemcpyAsync(ex->h_out, ex->d_outPack.p, total, hipMemcpyDeviceToHost, ex->stream));
    }
    HIP_TRY(hipStreamSynchronize(ex->stream));
    for (int f = 0; f < nframes; ++f)
        if (hn[f] > capacity) return setError(ORBX_ERR_CAPACITY, "frame %d has %d keypoints, output capacity is %d", f, hn[f], capacity);
    for (int f = 0; f < nframes; ++f) {
        const size_t o = host->hostIndex(f);
        if (kps && hn[f] > 0) memcpy(kps + o * capacity, hk + (size_t)f * cap, (size_t)hn[f] * sizeof(orbx_keypoint));
        if (desc && hn[f] > 0) memcpy(desc + o * capacity * 32, hd + (size_t)f * cap * 32, (size_t)hn[f] * 32);
        if (n_out) n_out[o] = hn[f];
        if (mono_out) mono_out[o] = hm[f];
    }
    return ORBX_OK;
}

// One handle's share of a host batch: frames and results at the run mapping of `hf` (see HostFrames).
static int extractBatchHost(orbx_extractor *ex, HostFrames hf, int nframes, int w, int h, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc,
                            int capacity, int32_t *n_out, int32_t *mono_out, bool submit = false)
{
    if (!ex || !hf.imgs) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch: null argument");
    const bool small = nframes <= ORBX_GRAPH_MAX_FRAMES;
    // one submitted batch may be pending when another one of the pipelined (large-batch) kind is SUBMITTED; every other call waits first
    if (ex->nPending > 0 && !(submit && !small && ex->nPending == 1 && !ex->pending[0]->small))
        return setError(ORBX_ERR_INVALID_ARG, ex->nPending == 2 && submit ? "two submitted batches are pending on this handle: call orbx_extract_batch_wait first"
                        : "a batch submitted with orbx_extract_batch_submit is still pending on this handle: call orbx_extract_batch_wait first");
    if (nframes <= 0 || w <= 0 || h <= 0 || hf.stride < w) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch: bad shape");
    if (nframes > ex->prm.max_batch) return setError(ORBX_ERR_CAPACITY, "batch %d exceeds max_batch %d", nframes, ex->prm.max_batch);
    if (w > ex->prm.max_width || h > ex->prm.max_height)
        return setError(ORBX_ERR_CAPACITY, "image %dx%d exceeds the handle's max_width x max_height = %dx%d", w, h, ex->prm.max_width, ex->prm.max_height);
    HIP_TRY(hipSetDevice(ex->prm.device));
    int rc = computeGeometry(ex, w, h);
    if (rc != ORBX_OK) return rc;
    if ((rc = ensureBuffers(ex)) != ORBX_OK) return rc;
    if (small) {  // small batches: one replayed graph, one round trip for the results
        hf.pipe = nullptr;
        rc = runBatch(ex, ex->d_in.p, nframes, w, h, ex->inPitch, ex->inFrameStride, lap0, lap1, &hf);
        if (rc != ORBX_OK) return rc;
        if (submit) {  // the launches are queued; the wait fetches the results
            PendingBatch *pb = new (std::nothrow) PendingBatch();
            if (!pb) { drainHandle(ex); return setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_extract_batch_submit: host allocation failed"); }
            pb->small = true; pb->hf = hf; pb->nframes = nframes; pb->capacity = capacity; pb->kps = kps; pb->desc = desc; pb->nOut = n_out; pb->monoOut = mono_out;
            ex->pending[0] = pb; ex->nPending = 1;
            return ORBX_OK;
        }
        return downloadSmall(ex, &hf, nframes, kps, desc, capacity, n_out, mono_out);
    }
    // the slot of this batch: the one the pending batch (if any) does not hold
    const int slot = ex->nPending == 1 ? 1 - ex->pending[0]->pipe.slot : 0;
    if ((rc = ensureStage(ex, slot)) != ORBX_OK) return rc;
    uint8_t *dIn = ex->d_in.p;
    if (slot == 1) {
        if ((rc = ex->slots[1].d_in1.ensure((size_t)ex->prm.max_batch * ex->inFrameStride + 64)) != ORBX_OK) return rc;
        dIn = ex->slots[1].d_in1.p;
    }
    // H2D staging and the D2H of the results happen inside the launch sequence, sub-batch by sub-batch
    PendingBatch *pb = nullptr;
    if (submit) {
        pb = new (std::nothrow) PendingBatch();
        if (!pb) return setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_extract_batch_submit: host allocation failed");
    }
    HostPipe localPipe;
    HostPipe &pipe = pb ? pb->pipe : localPipe;
    pipe.kps = kps; pipe.desc = desc; pipe.capacity = capacity; pipe.nOut = n_out; pipe.monoOut = mono_out; pipe.slot = slot;
    static const int feederEnv = [] { const char *e = getenv("ORBX_HOST_FEEDER"); return e ? atoi(e) : -1; }();  // A/B: 0 no helper thread, 1 always
    // (first to last byte this handle reads: its last frame sits at host index hostIndex(nframes - 1))
    const size_t imgBytes = hf.hostIndex(nframes - 1) * (size_t)hf.frameStride + (size_t)(h - 1) * (size_t)hf.stride + (size_t)w;
    pipe.feeder = feederEnv >= 0 ? feederEnv != 0 : !(hf.frameStride > 0 && isPageLockedRange(hf.imgs, imgBytes));
    pipe.direct = capacity == ex->maxKeypoints && hf.run >= nframes && (!kps || isPageLockedRange(kps, (size_t)nframes * capacity * sizeof(orbx_keypoint))) &&
                  (!desc || isPageLockedRange(desc, (size_t)nframes * capacity * 32));
    hf.pipe = &pipe;
    if (pb) {
        // submitted: every copy and launch of the batch is queued by this call and nothing of it is waited for - which only page-locked
        // frames allow (pageable ones cross the link in blocking copies on a helper thread that the call has to stay with)
        if (pipe.feeder) { delete pb; return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_submit: the frames must lie in page-locked memory (orbx_host_alloc / orbx_host_register), consecutive frames at a positive frame_stride"); }
        pb->hf = hf; pb->hf.pipe = &pb->pipe; pb->nframes = nframes; pb->capacity = capacity; pb->kps = kps; pb->desc = desc; pb->nOut = n_out; pb->monoOut = mono_out;
        ex->pending[ex->nPending++] = pb;
        ex->curSlot = slot;
        // the results of the batch in flight leave the device buffers this batch's back stream writes: behind its last copy-out
        rc = ORBX_OK;
        if (ex->nPending == 2) {
            const PendingBatch *prev = ex->pending[0];
            const hipError_t he = hipStreamWaitEvent(ex->stream3, ex->slots[prev->pipe.slot].evOut[prev->pipe.nsub - 1], 0);
            if (he != hipSuccess) rc = setError(ORBX_ERR_HIP, "orbx_extract_batch_submit: hipStreamWaitEvent: %s", hipGetErrorString(he));
        }
        if (rc == ORBX_OK) rc = runBatch(ex, dIn, nframes, w, h, ex->inPitch, ex->inFrameStride, lap0, lap1, &pb->hf);
        if (rc != ORBX_OK) { dropPending(ex); }   // (nothing of either batch may still run: both are dropped, the call reports the error)
        return rc;
    }
    ex->curSlot = 0;
    rc = runBatch(ex, ex->d_in.p, nframes, w, h, ex->inPitch, ex->inFrameStride, lap0, lap1, &hf);
    if (rc == ORBX_OK) rc = pipeDrain(ex, &hf, true, pipe.nsub);
    if (rc != ORBX_OK) drainHandle(ex);  // nothing of this call may still be running, or copying out of / into the caller's arrays
    if (rc == ORBX_OK && hostTimeline()) {
        auto at = [&](hipEvent_t e) { float ms = -1.f; (void)hipEventElapsedTime(&ms, ex->evIn, e); return ms; };
        fprintf(stderr, "[orbx] host batch of %d frames in %d sub-batches (ms since the call's first event):\n", nframes, pipe.nsub);
        for (int k = 0; k < pipe.nsub; ++k)
            fprintf(stderr, "[orbx]   sub-batch %d (%4d frames): copy in %6.3f - %6.3f | kernels done %6.3f | results out %6.3f\n", k, pipe.nf[k],
                    at(ex->slots[0].evCopy0[k]), at(ex->slots[0].evCopy[k]), at(ex->slots[0].evBack[k]), at(ex->slots[0].evOut[k]));
    }
    return rc;  // (~HostPipe joins the copy thread)
}

extern "C" int orbx_extract_batch(orbx_extractor *ex, const uint8_t *imgs, int nframes, int w, int h, ptrdiff_t stride,
                                  ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc, int capacity,
                                  int32_t *n_out, int32_t *mono_out)
{
    ORBX_API_ENTRY;
    const HostFrames hf = {imgs, stride, frame_stride, std::max(nframes, 1), std::max(nframes, 1), nullptr};
    return extractBatchHost(ex, hf, nframes, w, h, lap0, lap1, kps, desc, capacity, n_out, mono_out);
}

// Submit / wait: the same batch call in two halves.  A host-pointer call ends in a tail that no other work of the SAME call can cover - the
// kernels of its last sub-batch and the copy-out of their results, ~0.9 of the 6.7 ms of a 1024-frame VGA call whose input copies take
// 5.5 ms of link time (profiles/r03_host_batch_timeline.txt).  With the call split, two batches are in flight on the handle: the second
// one's frames cross the link into the second staging buffer (PipeSlot::d_in1) under the first one's kernels and copy-out; its kernels
// queue behind the first one's on the same streams (extractBatchHost, `submit`).  Two HANDLES fed in turn do not get there - their
// streams serialise in the runtime's hardware queues (profiles/r05_submit_wait.txt).
extern "C" int orbx_extract_batch_submit(orbx_extractor *ex, const uint8_t *imgs, int nframes, int w, int h, ptrdiff_t stride,
                                         ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc, int capacity,
                                         int32_t *n_out, int32_t *mono_out)
{
    ORBX_API_ENTRY;
    const HostFrames hf = {imgs, stride, frame_stride, std::max(nframes, 1), std::max(nframes, 1), nullptr};
    return extractBatchHost(ex, hf, nframes, w, h, lap0, lap1, kps, desc, capacity, n_out, mono_out, true);
}

extern "C" int orbx_extract_batch_wait(orbx_extractor *ex)
{
    ORBX_API_ENTRY;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_wait: null handle");
    if (ex->nPending == 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_wait: no submitted batch on this handle");
    PendingBatch *pb = ex->pending[0];   // the oldest
    HIP_TRY(hipSetDevice(ex->prm.device));
    ex->pending[0] = ex->pending[1]; ex->pending[1] = nullptr; --ex->nPending;
    int rc;
    if (pb->small) rc = downloadSmall(ex, &pb->hf, pb->nframes, pb->kps, pb->desc, pb->capacity, pb->nOut, pb->monoOut);
    else rc = pipeDrain(ex, &pb->hf, true, pb->pipe.nsub);
    if (rc != ORBX_OK) dropPending(ex);   // (with everything drained; a second pending batch is dropped with it)
    delete pb;
    return rc;
}

// The C++-side sharding entry (frames shard over the GPUs, no data-path collective): block b of `block` consecutive
// frames goes to handle b mod nhandles (block = 1: frame f -> GPU f mod G), every handle works through its blocks on a thread of its own
// - the reference drives its two extractors from two std::threads the same way (src/Frame.cc:124-127) - in batches of at most its
// max_batch frames, and every result lands in the caller's arrays at its frame index.
extern "C" int orbx_extract_batch_sharded(orbx_extractor *const *handles, int nhandles, int block, const uint8_t *imgs, int nframes, int w, int h,
                                          ptrdiff_t stride, ptrdiff_t frame_stride, int lap0, int lap1, orbx_keypoint *kps, uint8_t *desc,
                                          int capacity, int32_t *n_out, int32_t *mono_out)
{
    ORBX_API_ENTRY;
    if (!handles || nhandles <= 0 || !imgs) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_sharded: null argument");
    if (block <= 0 || nframes <= 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_sharded: block and nframes must be positive");
    for (int g = 0; g < nhandles; ++g) {
        if (!handles[g]) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_sharded: handle %d is null", g);
        for (int k = 0; k < g; ++k)
            if (handles[k] == handles[g]) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract_batch_sharded: handle %d is listed twice (one handle is never entered concurrently)", g);
        if (handles[g]->prm.max_batch < block) return setError(ORBX_ERR_CAPACITY, "orbx_extract_batch_sharded: handle %d has max_batch %d < block %d", g, handles[g]->prm.max_batch, block);
    }
    std::vector<int> rcs(nhandles, ORBX_OK);
    std::vector<std::string> errs(nhandles);
    auto work = [&](int g) {
        ORBX_API_ENTRY;
        orbx_extractor *ex = handles[g];
        // this handle's frames: blocks g, g + G, g + 2 G ... ; local frame i = frame (i / block) * G * block + g * block + i % block
        const int nblocksAll = (nframes + block - 1) / block;
        int mine = 0;
        for (int b = g; b < nblocksAll; b += nhandles) mine += std::min(block, nframes - b * block);
        const int perCall = ex->prm.max_batch / block * block;  // whole blocks per call
        for (int done = 0; done < mine && rcs[g] == ORBX_OK; done += perCall) {
            const int cnt = std::min(perCall, mine - done);
            const size_t first = (size_t)(done / block) * nhandles * block + (size_t)g * block;  // frame index of local frame `done`
            const HostFrames hf = {imgs + first * (size_t)frame_stride, stride, frame_stride, block, nhandles * block, nullptr};
            rcs[g] = extractBatchHost(ex, hf, cnt, w, h, lap0, lap1, kps ? kps + first * capacity : nullptr,
                                      desc ? desc + first * (size_t)capacity * 32 : nullptr, capacity, n_out ? n_out + first : nullptr,
                                      mono_out ? mono_out + first : nullptr);
        }
        if (rcs[g] != ORBX_OK) errs[g] = g_lastError;
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < nhandles; ++g) threads.emplace_back(work, g);
    work(0);
    for (auto &t : threads) t.join();
    for (int g = 0; g < nhandles; ++g)
        if (rcs[g] != ORBX_OK) return setError(rcs[g], "handle %d (device %d): %s", g, handles[g]->prm.device, errs[g].c_str());
    return ORBX_OK;
}

// Page-locked host memory for frames and results: copies from / to it are asynchronous and need no staging (see HostPipe).
extern "C" void *orbx_host_alloc(size_t bytes)
{
    ORBX_API_ENTRY;
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_host_alloc(%zu bytes) failed", bytes); return nullptr; }
    return p;
}
extern "C" void orbx_host_free(void *p)
{
    ORBX_API_ENTRY;
    if (p) (void)hipHostFree(p);
}
extern "C" int orbx_host_register(void *p, size_t bytes)
{
    ORBX_API_ENTRY;
    if (!p || bytes == 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_host_register: null pointer or empty range");
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e != hipSuccess) { (void)hipGetLastError(); return setError(ORBX_ERR_HIP, "orbx_host_register(%zu bytes): %s", bytes, hipGetErrorString(e)); }
    return ORBX_OK;
}
extern "C" int orbx_host_unregister(void *p)
{
    ORBX_API_ENTRY;
    if (!p) return ORBX_OK;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return setError(ORBX_ERR_HIP, "orbx_host_unregister: %s", hipGetErrorString(e)); }
    return ORBX_OK;
}

extern "C" int orbx_extract(orbx_extractor *ex, const uint8_t *img, int w, int h, ptrdiff_t stride, int lap0, int lap1,
                            orbx_keypoint *kps, uint8_t *desc, int capacity, int *n_out)
{
    ORBX_API_ENTRY;
    if (n_out) *n_out = 0;
    if (!ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_extract: null handle") - 100;
    if (!img || w <= 0 || h <= 0) return -1;  // `if (_image.empty()) return -1;` (reference src/ORBextractor.cc:1231-1232)
    int32_t n = 0, mono = 0;
    int rc = orbx_extract_batch(ex, img, 1, w, h, stride, (ptrdiff_t)stride * h, lap0, lap1, kps, desc, capacity, &n, &mono);
    if (rc != ORBX_OK) return rc - 100;
    if (n_out) *n_out = n;
    return mono;
}

// ---------------------------------------------------------------------------------------------------------
// pyramid access + stage taps
// ---------------------------------------------------------------------------------------------------------
extern "C" int orbx_get_level_size(const orbx_extractor *ex, int level, int *w, int *h)
{
    ORBX_API_ENTRY;
    if (!ex || level < 0 || level >= ex->prm.nlevels || ex->geom.empty()) return setError(ORBX_ERR_INVALID_ARG, "orbx_get_level_size: bad argument");
    if (w) *w = ex->geom[level].w;
    if (h) *h = ex->geom[level].h;
    return ORBX_OK;
}

extern "C" int orbx_get_pyramid_level(orbx_extractor *ex, int frame, int level, int blurred, uint8_t *out)
{
    ORBX_API_ENTRY;
    if (!ex || !out || level < 0 || level >= ex->prm.nlevels || frame < 0 || frame >= ex->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_get_pyramid_level: bad argument");
    HIP_TRY(hipSetDevice(ex->prm.device));
    const OrbxLevelGeom &g = ex->geom[level];
    HIP_TRY(hipStreamSynchronize(ex->stream));
    if (blurred) {  // tiled on the device: fetch the frame's tiles and lay them out row-major
        const int tx = blurTilesX(g);
        std::vector<uint8_t> tiles((size_t)blurFrameBytes(g));
        HIP_TRY(hipMemcpy(tiles.data(), ex->d_blur[level].p + (size_t)frame * blurFrameBytes(g), tiles.size(), hipMemcpyDeviceToHost));
        for (int y = 0; y < g.h; ++y)
            for (int x = 0; x < g.w; ++x)
                out[(size_t)y * g.w + x] = tiles[((size_t)(y >> 3) * tx + (x >> 4)) * 128 + (size_t)((y & 7) * 16 + (x & 15))];
        return ORBX_OK;
    }
    const uint8_t *src; int pitch; long long fs;
    if (level == 0) { src = ex->lastSrc; pitch = ex->lastSrcPitch; fs = ex->lastSrcFrameStride; }
    else { src = ex->d_raw[level].p; pitch = g.pitch; fs = g.frameStride; }
    HIP_TRY(hipMemcpy2D(out, g.w, src + (size_t)frame * fs, pitch, g.w, g.h, hipMemcpyDeviceToHost));
    return ORBX_OK;
}

extern "C" int orbx_debug_get_candidates(orbx_extractor *ex, int frame, int level, uint32_t *out, int cap)
{
    ORBX_API_ENTRY;
    if (!ex || level < 0 || level >= ex->prm.nlevels || frame < 0 || frame >= ex->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_debug_get_candidates: bad argument");
    HIP_TRY(hipSetDevice(ex->prm.device));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    const OrbxLevelGeom &g = ex->geom[level];
    std::vector<int> tc(g.nTiles);
    HIP_TRY(hipMemcpy(tc.data(), ex->d_tileCount.p + (size_t)frame * ex->tilesPerFrame + g.tileBase, g.nTiles * sizeof(int), hipMemcpyDeviceToHost));
    int total = 0;
    for (int t = 0; t < g.nTiles; ++t) {
        int n = tc[t];
        if (n < 0 || n > g.slotCap) return setError(ORBX_ERR_HIP, "tile %d count %d out of range", t, n);
        int ncopy = std::min(n, std::max(0, cap - total));
        if (ncopy > 0 && out)
            HIP_TRY(hipMemcpy(out + total, ex->d_cand.p + (size_t)frame * ex->candFrameStride + g.candBase + (size_t)t * g.slotCap,
                              ncopy * sizeof(uint32_t), hipMemcpyDeviceToHost));
        total += n;
    }
    return total;
}

extern "C" int orbx_debug_get_selected(orbx_extractor *ex, int frame, int level, uint32_t *out, int cap)
{
    ORBX_API_ENTRY;
    if (!ex || level < 0 || level >= ex->prm.nlevels || frame < 0 || frame >= ex->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_debug_get_selected: bad argument");
    HIP_TRY(hipSetDevice(ex->prm.device));
    HIP_TRY(hipStreamSynchronize(ex->stream));
    int n = 0;
    HIP_TRY(hipMemcpy(&n, ex->d_selCount.p + (size_t)frame * ex->prm.nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    int ncopy = std::min(n, cap);
    if (ncopy > 0 && out)
        HIP_TRY(hipMemcpy(out, ex->d_sel.p + (size_t)frame * ex->selFrameStride + ex->geom[level].selBase, ncopy * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return n;
}

extern "C" void orbx_synth_frame(uint64_t frame_index, int w, int h, uint8_t *out, ptrdiff_t stride)
{
    ORBX_API_ENTRY;
    orbx_synth_frame_impl(frame_index, w, h, out, stride);
}

// ---------------------------------------------------------------------------------------------------------
// matcher
// ---------------------------------------------------------------------------------------------------------
// staging of the packed host transfers (HostPack / HostFetch below) of a handle: page-locked memory both ways, a device arena both ways
struct HostXfer {
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_arena, d_arenaOut;
    uint8_t *h_arena = nullptr, *h_out = nullptr;
    size_t h_arenaBytes = 0, h_outBytes = 0;
    void release()
    {
        d_arena.release(); d_arenaOut.release();
        if (h_arena) (void)hipHostFree(h_arena);
        if (h_out) (void)hipHostFree(h_out);
        h_arena = h_out = nullptr; h_arenaBytes = h_outBytes = 0;
    }
};

struct orbx_matcher {
    int device = 0, maxQ = 0, maxT = 0;
    hipStream_t stream = nullptr;
    DevBuf<uint8_t> d_q, d_t;
    DevBuf<int> d_idx, d_dist, d_mat;
    DevBuf<uint32_t> d_part;      // kNN-2: per-split partial results
    DevBuf<int> d_bIdx, d_bDist, d_bSad;  // batched stereo matching: per-pair row-band results
    DevBuf<orbx_keypoint> d_bKps;         // batched tracking: undistorted keypoints
    DevBuf<int> d_rowStart, d_rowItems;   // stereo row buckets (CSR) of the right image(s)
    hipEvent_t evWait = nullptr;  // ordering against the extractors' streams
    DevBuf<orbx_keypoint> d_kl, d_kr;
    DevBuf<float> d_scale, d_uRight, d_depth;
    DevBuf<int> d_sad;
    DevBuf<int> d_grid, d_work;   // frame grid (cellStart + items + cellOf), SearchForInitialization state
    DevBuf<float> d_prev;
    DevBuf<signed char> d_bin;
    DevBuf<uint32_t> d_sbpCand;   // SearchByProjection: candidate lists [maxQ][ORBX_SBP_CAP]
    DevBuf<uint8_t> d_siWork;     // SearchForInitialization: octave-0 subset of F2 and the shortest candidates of every F1 keypoint
    DevBuf<orbx_projection> d_proj;
    // single calls with host pointers: all inputs of a call cross the link in ONE transfer (HostPack), the results in one more
    HostXfer xf;
};

// The inputs of a single call with host pointers (keypoints, descriptors, flags, projections ...: two to ten arrays of a few KB each)
// are packed into the matcher's page-locked staging and cross the link as ONE copy into an arena of the same layout; a piece either
// stays there (dev()) or one small launch moves the pieces on to the buffers the kernels read (add(.., ddst)).  The results go the
// same way back (HostFetch).  A hipMemcpyAsync from or to pageable memory costs 20 - 30 us of host time and ~7 us on the device
// whatever its size (tools/experiments/single_call_trace.sh): seven of them were a third of a single-frame orbx_search_by_projection.
// Only for calls that synchronise the stream before they return: the staging is reused by the next call.
static int growPinned(uint8_t *&p, size_t &have, size_t need, size_t keep)
{
    if (have >= need) return ORBX_OK;
    ORBX_SETUP_LOCK;
    uint8_t *q = nullptr;
    const size_t want = std::max(need + need / 2, (size_t)1 << 20);
    HIP_TRY(hipHostMalloc((void **)&q, want, hipHostMallocDefault));
    if (p) { if (keep) memcpy(q, p, keep); (void)hipHostFree(p); }
    p = q; have = want;
    return ORBX_OK;
}
static const size_t kDirectBytes = (size_t)256 << 10;
struct HostPack {
    orbx_matcher *m;         // (NULL for another handle's transfers)
    HostXfer *x;
    OrbxCopyPieces sc;       // pieces that go on to a buffer of their own (src = offset into the arena until send())
    size_t bytes = 0;
    int err = ORBX_OK, npieces = 0;
    explicit HostPack(orbx_matcher *mm) : m(mm), x(&mm->xf) { sc.n = 0; }
    explicit HostPack(HostXfer *xx) : m(nullptr), x(xx) { sc.n = 0; }
    // packs `nbytes` from `src`; returns the offset of the piece in the arena.  ddst != NULL: the piece is copied on to ddst by send()
    size_t add(const void *src, size_t nbytes, void *ddst = nullptr)
    {
        const size_t off = (bytes + 255) & ~(size_t)255;
        if (err != ORBX_OK || nbytes == 0) return off;
        if (ddst && nbytes > kDirectBytes) {  // a large piece gains nothing from the detour: straight to its buffer, in stream order
            if (hipMemcpyAsync(ddst, src, nbytes, hipMemcpyHostToDevice, x->stream) != hipSuccess) err = setError(ORBX_ERR_HIP, "hipMemcpyAsync (host to device) failed");
            return off;
        }
        if ((err = growPinned(x->h_arena, x->h_arenaBytes, off + nbytes, bytes)) != ORBX_OK) return off;
        memcpy(x->h_arena + off, src, nbytes);
        bytes = off + nbytes;
        ++npieces;
        if (ddst) {
            if (sc.n >= ORBX_COPY_PIECES || nbytes > 0xFFFFFFFFull) { err = setError(ORBX_ERR_INVALID_ARG, "internal: too many / too large pieces in a packed transfer"); return off; }
            sc.dst[sc.n] = ddst; sc.src[sc.n] = (const void *)off; sc.bytes[sc.n] = (unsigned)nbytes; ++sc.n;
        }
        return off;
    }
    int send()
    {
        if (err != ORBX_OK) return err;
        if (bytes == 0) return ORBX_OK;
        if (sc.n == 1 && npieces == 1) {  // one piece with a buffer of its own: straight from the staging
            HIP_TRY(hipMemcpyAsync(sc.dst[0], x->h_arena + (size_t)sc.src[0], sc.bytes[0], hipMemcpyHostToDevice, x->stream));
            return ORBX_OK;
        }
        int rc = x->d_arena.ensure(bytes);
        if (rc != ORBX_OK) return rc;
        HIP_TRY(hipMemcpyAsync(x->d_arena.p, x->h_arena, bytes, hipMemcpyHostToDevice, x->stream));
        for (int i = 0; i < sc.n; ++i) sc.src[i] = x->d_arena.p + (size_t)sc.src[i];
        HIP_TRY(orbxLaunchCopyPieces(x->stream, sc));
        return ORBX_OK;
    }
    template <class T> T *dev(size_t off) const { return (T *)(x->d_arena.p + off); }
};
// The results of such a call: the pieces are gathered into one device buffer by one launch, cross the link as one copy into page-locked
// staging and are handed to the caller's arrays after the stream has been synchronised (finish()).
struct HostFetch {
    HostXfer *x;
    OrbxCopyPieces gt;       // dst = offset into the gather buffer until finish()
    void *host[ORBX_COPY_PIECES];
    size_t bytes = 0;
    int err = ORBX_OK;
    explicit HostFetch(orbx_matcher *mm) : x(&mm->xf) { gt.n = 0; }
    explicit HostFetch(HostXfer *xx) : x(xx) { gt.n = 0; }
    void add(void *hdst, const void *dsrc, size_t nbytes)
    {
        if (err != ORBX_OK || nbytes == 0) return;
        if (nbytes > kDirectBytes) {  // (see HostPack::add)
            if (hipMemcpyAsync(hdst, dsrc, nbytes, hipMemcpyDeviceToHost, x->stream) != hipSuccess) err = setError(ORBX_ERR_HIP, "hipMemcpyAsync (device to host) failed");
            return;
        }
        if (gt.n >= ORBX_COPY_PIECES || nbytes > 0xFFFFFFFFull) { err = setError(ORBX_ERR_INVALID_ARG, "internal: too many / too large pieces in a packed transfer"); return; }
        const size_t off = (bytes + 255) & ~(size_t)255;
        gt.dst[gt.n] = (void *)off; gt.src[gt.n] = dsrc; gt.bytes[gt.n] = (unsigned)nbytes; host[gt.n] = hdst; ++gt.n;
        bytes = off + nbytes;
    }
    // gathers, copies, synchronises the stream, delivers
    int finish()
    {
        if (err != ORBX_OK) return err;
        if (bytes == 0) { HIP_TRY(hipStreamSynchronize(x->stream)); return ORBX_OK; }
        int rc = x->d_arenaOut.ensure(bytes);
        if (rc != ORBX_OK) return rc;
        if ((rc = growPinned(x->h_out, x->h_outBytes, bytes, 0)) != ORBX_OK) return rc;
        size_t offs[ORBX_COPY_PIECES];
        for (int i = 0; i < gt.n; ++i) { offs[i] = (size_t)gt.dst[i]; gt.dst[i] = x->d_arenaOut.p + offs[i]; }
        if (gt.n == 1) {
            HIP_TRY(hipMemcpyAsync(x->h_out + offs[0], gt.src[0], gt.bytes[0], hipMemcpyDeviceToHost, x->stream));
        } else {
            HIP_TRY(orbxLaunchCopyPieces(x->stream, gt));
            HIP_TRY(hipMemcpyAsync(x->h_out, x->d_arenaOut.p, bytes, hipMemcpyDeviceToHost, x->stream));
        }
        HIP_TRY(hipStreamSynchronize(x->stream));
        for (int i = 0; i < gt.n; ++i) memcpy(host[i], x->h_out + offs[i], gt.bytes[i]);
        return ORBX_OK;
    }
};

extern "C" int orbx_matcher_create(int device, int max_query, int max_train, orbx_matcher **out)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!out || max_query <= 0 || max_train <= 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_matcher_create: bad argument");
    *out = nullptr;
    { const int rcEnv = checkEnvironmentSwitches("orbx_matcher_create"); if (rcEnv != ORBX_OK) return rcEnv; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return setError(ORBX_ERR_NO_DEVICE, "orbx_matcher_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return setError(ORBX_ERR_INVALID_ARG, "orbx_matcher_create: device out of range");
    HIP_TRY(hipSetDevice(device));
    orbx_matcher *m = new (std::nothrow) orbx_matcher();
    if (!m) return setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_matcher_create: host allocation failed");
    m->device = device; m->maxQ = max_query; m->maxT = max_train;
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete m; return setError(ORBX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    m->xf.stream = m->stream;
    int rc;
    if ((rc = m->d_q.ensure((size_t)max_query * 32)) != ORBX_OK || (rc = m->d_t.ensure((size_t)max_train * 32)) != ORBX_OK ||
        (rc = m->d_idx.ensure((size_t)max_query * 2)) != ORBX_OK || (rc = m->d_dist.ensure((size_t)max_query * 2)) != ORBX_OK ||
        (rc = m->d_kl.ensure(max_query)) != ORBX_OK || (rc = m->d_kr.ensure(max_train)) != ORBX_OK ||
        (rc = m->d_scale.ensure(2 * ORBX_MAX_LEVELS)) != ORBX_OK) {
        orbx_matcher_destroy(m);
        return rc;
    }
    *out = m;
    return ORBX_OK;
}

extern "C" void orbx_matcher_destroy(orbx_matcher *m)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) { (void)hipStreamSynchronize(m->stream); (void)hipStreamDestroy(m->stream); }
    m->d_q.release(); m->d_t.release(); m->d_idx.release(); m->d_dist.release(); m->d_mat.release(); m->d_part.release(); m->d_bIdx.release(); m->d_bDist.release(); m->d_bSad.release(); m->d_bKps.release(); m->d_rowStart.release(); m->d_rowItems.release();
    if (m->evWait) (void)hipEventDestroy(m->evWait);
    m->d_kl.release(); m->d_kr.release(); m->d_scale.release(); m->d_uRight.release(); m->d_depth.release(); m->d_sad.release();
    m->d_grid.release(); m->d_work.release(); m->d_prev.release(); m->d_bin.release(); m->d_sbpCand.release(); m->d_proj.release();
    m->xf.release();
    delete m;
}

extern "C" int orbx_matcher_sync(orbx_matcher *m)
{
    ORBX_API_ENTRY;
    if (!m) return setError(ORBX_ERR_INVALID_ARG, "orbx_matcher_sync: null handle");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

// query / train descriptors of a call into d_q / d_t (pieces of the call's packed transfer: in.send() moves them)
static int uploadQT(HostPack &in, const uint8_t *q, int nq, const uint8_t *t, int nt)
{
    orbx_matcher *m = in.m;
    if (nq > m->maxQ || nt > m->maxT) return setError(ORBX_ERR_CAPACITY, "matcher capacity (%d x %d) exceeded by %d x %d", m->maxQ, m->maxT, nq, nt);
    if (nq > 0) in.add(q, (size_t)nq * 32, m->d_q.p);
    if (nt > 0) in.add(t, (size_t)nt * 32, m->d_t.p);
    return in.err;
}

extern "C" int orbx_hamming_matrix(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *dist)
{
    ORBX_API_ENTRY;
    if (!m || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || !dist) return setError(ORBX_ERR_INVALID_ARG, "orbx_hamming_matrix: bad argument");
    if (nq == 0 || nt == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    int rc = uploadQT(in, q, nq, t, nt);
    if (rc != ORBX_OK || (rc = in.send()) != ORBX_OK) return rc;
    if ((rc = m->d_mat.ensure((size_t)nq * nt)) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchHammingMatrix(m->stream, m->d_q.p, nq, m->d_t.p, nt, m->d_mat.p));
    HostFetch out(m);
    out.add(dist, m->d_mat.p, (size_t)nq * nt * sizeof(int));
    return out.finish();
}

// kNN-2 of `npairs` independent (query, train) sets in one launch: split count, scratch for the per-split partial results
static int knn2Launch(orbx_matcher *m, int npairs, const uint8_t *q, long long qStride, const int *nqDev, const int *qBeginDev, int nq,
                      const uint8_t *t, long long tStride, const int *ntDev, const int *tBeginDev, int nt, int *idx, int *dist)
{
    if (nt >= (1 << 23)) return setError(ORBX_ERR_CAPACITY, "kNN-2: at most 2^23 - 1 train rows per pair (got %d)", nt);
    const char *fe = getenv("ORBX_KNN_SPLIT");  // tests of the split merge force a split count
    const int force = fe ? atoi(fe) : 0;
    const int nSplit = orbxKnn2Splits(npairs, nq, nt, force);
    if (nSplit > 1) {
        int rc = m->d_part.ensure((size_t)npairs * nSplit * nq * 2);
        if (rc != ORBX_OK) return rc;
    }
    HIP_TRY(orbxLaunchKnn2Batch(m->stream, npairs, q, qStride, nqDev, qBeginDev, nq, t, tStride, ntDev, tBeginDev, nt, nSplit, m->d_part.p, idx, dist));
    return ORBX_OK;
}

extern "C" int orbx_match_knn2_device(orbx_matcher *m, const void *d_q, int nq, const void *d_t, int nt, void *d_idx, void *d_dist)
{
    ORBX_API_ENTRY;
    if (!m || nq < 0 || nt < 0 || !d_idx || !d_dist || (nq > 0 && !d_q) || (nt > 0 && !d_t))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_match_knn2_device: bad argument");
    if (nq == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    return knn2Launch(m, 1, (const uint8_t *)d_q, 0, nullptr, nullptr, nq, (const uint8_t *)d_t, 0, nullptr, nullptr, nt, (int *)d_idx, (int *)d_dist);
}

extern "C" int orbx_match_knn2_batch_device(orbx_matcher *m, int npairs, const void *d_q, int64_t q_pair_stride, const int32_t *d_nq,
                                            const int32_t *d_q_begin, int max_nq, const void *d_t, int64_t t_pair_stride,
                                            const int32_t *d_nt, const int32_t *d_t_begin, int max_nt, void *d_idx, void *d_dist)
{
    ORBX_API_ENTRY;
    if (!m || npairs < 0 || max_nq < 0 || max_nt < 0 || !d_idx || !d_dist || (max_nq > 0 && !d_q) || (max_nt > 0 && !d_t) || q_pair_stride < 0 ||
        t_pair_stride < 0)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_match_knn2_batch_device: bad argument");
    if (npairs == 0 || max_nq == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    return knn2Launch(m, npairs, (const uint8_t *)d_q, q_pair_stride, d_nq, d_q_begin, max_nq, (const uint8_t *)d_t, t_pair_stride, d_nt, d_t_begin,
                      max_nt, (int *)d_idx, (int *)d_dist);
}

extern "C" int orbx_match_knn2(orbx_matcher *m, const uint8_t *q, int nq, const uint8_t *t, int nt, int32_t *idx, int32_t *dist)
{
    ORBX_API_ENTRY;
    if (!m || nq < 0 || nt < 0 || (nq > 0 && !q) || (nt > 0 && !t) || !idx || !dist) return setError(ORBX_ERR_INVALID_ARG, "orbx_match_knn2: bad argument");
    if (nq == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    int rc = uploadQT(in, q, nq, t, nt);
    if (rc != ORBX_OK || (rc = in.send()) != ORBX_OK) return rc;
    if ((rc = knn2Launch(m, 1, m->d_q.p, 0, nullptr, nullptr, nq, m->d_t.p, 0, nullptr, nullptr, nt, m->d_idx.p, m->d_dist.p)) != ORBX_OK) return rc;
    HostFetch out(m);
    out.add(idx, m->d_idx.p, (size_t)nq * 2 * sizeof(int));
    out.add(dist, m->d_dist.p, (size_t)nq * 2 * sizeof(int));
    return out.finish();
}

extern "C" int orbx_match_stereo_rows(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr,
                                      const uint8_t *dr, int nr, int n_rows, const float *scale_factors, int nlevels, float min_d,
                                      float max_d, int32_t *best_idx, int32_t *best_dist)
{
    ORBX_API_ENTRY;
    if (!m || nl < 0 || nr < 0 || (nl > 0 && (!kl || !dl)) || (nr > 0 && (!kr || !dr)) || !scale_factors || nlevels <= 0 ||
        nlevels > ORBX_MAX_LEVELS || !best_idx || !best_dist)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_match_stereo_rows: bad argument");
    if (nl == 0) return ORBX_OK;
    for (int i = 0; i < nr; ++i)
        if (kr[i].octave < 0 || kr[i].octave >= nlevels) return setError(ORBX_ERR_INVALID_ARG, "right keypoint %d has octave %d outside [0,%d)", i, kr[i].octave, nlevels);
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    int rc = uploadQT(in, dl, nl, dr, nr);
    if (rc != ORBX_OK) return rc;
    in.add(kl, (size_t)nl * sizeof(orbx_keypoint), m->d_kl.p);
    in.add(kr, (size_t)nr * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(scale_factors, (size_t)nlevels * sizeof(float), m->d_scale.p);
    float maxScale = 1.f;
    for (int i = 0; i < nlevels; ++i) maxScale = std::max(maxScale, scale_factors[i]);
    const int itemStride = orbxStereoBandItems(nr, maxScale);
    if ((rc = m->d_rowStart.ensure((size_t)n_rows + 1)) != ORBX_OK || (rc = m->d_rowItems.ensure((size_t)std::max(itemStride, 1))) != ORBX_OK) return rc;
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchStereoRows(m->stream, m->d_kl.p, m->d_q.p, nl, m->d_kr.p, m->d_t.p, nr, n_rows, m->d_scale.p, min_d, max_d,
                                 m->d_idx.p, m->d_dist.p, nullptr, m->d_rowStart.p, m->d_rowItems.p, itemStride));
    HostFetch out(m);
    out.add(best_idx, m->d_idx.p, (size_t)nl * sizeof(int));
    out.add(best_dist, m->d_dist.p, (size_t)nl * sizeof(int));
    return out.finish();
}

#define ORBX_GRID_CELLS (64 * 48)

static int checkGridBounds(float minX, float maxX, float minY, float maxY, const char *who)
{
    if (!(maxX > minX) || !(maxY > minY)) return setError(ORBX_ERR_INVALID_ARG, "%s: empty image bounds", who);
    return ORBX_OK;
}

// ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) — reference src/ORBmatcher1.cc:225-427 (F.Nleft == -1).
static int checkFeatureVector(const int32_t *node, const int32_t *start, const int32_t *feat, int nfv, int n, const char *which)
{
    if (nfv < 0 || (nfv > 0 && (!node || !start || !feat))) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: bad %s FeatureVector", which);
    if (nfv == 0) return ORBX_OK;
    if (start[0] != 0 || start[nfv] > n) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: %s FeatureVector offsets outside [0, %d]", which, n);
    for (int j = 0; j < nfv; ++j) {
        // a FeatureVector node holds at least one feature (std::map entries are created by push_back): strictly ascending offsets,
        // which also bounds nfv by n - the device buffers of the node table are sized by the feature capacity
        if (start[j + 1] <= start[j]) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: %s FeatureVector offsets not strictly ascending", which);
        if (j > 0 && node[j] <= node[j - 1]) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: %s FeatureVector node ids not ascending", which);
    }
    for (int i = 0; i < start[nfv]; ++i)
        if (feat[i] < 0 || feat[i] >= n) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: %s FeatureVector holds feature %d of %d", which, feat[i], n);
    return ORBX_OK;
}

// mode 0: matches[f_n] indexed by frame feature; mode 1 (keyframe - keyframe, src/ORBmatcher2.cc:37-177): f_valid masks the second
// side, matches[kf_n] indexed by the first keyframe's feature
static int searchByBowImpl(orbx_matcher *m, const int32_t *kf_fv_node, const int32_t *kf_fv_start, const int32_t *kf_fv_feature,
                           int kf_nfv, const orbx_keypoint *kf_keys, const uint8_t *kf_desc, const uint8_t *kf_valid, int kf_n,
                           const int32_t *f_fv_node, const int32_t *f_fv_start, const int32_t *f_fv_feature, int f_nfv,
                           const orbx_keypoint *f_keys, const uint8_t *f_desc, const uint8_t *f_valid, int f_n, float nnratio,
                           int check_orientation, int mode, int32_t *matches)
{
    const int nOut = mode == 0 ? f_n : kf_n;
    if (!m || kf_n < 0 || f_n < 0 || (kf_n > 0 && (!kf_keys || !kf_desc || !kf_valid)) || (f_n > 0 && (!f_keys || !f_desc)) ||
        (nOut > 0 && !matches) || (mode == 1 && f_n > 0 && !f_valid))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_bow: bad argument");
    if (kf_n > m->maxQ || f_n > m->maxT || f_n > 65535)
        return setError(ORBX_ERR_CAPACITY, "%d x %d features exceed the matcher capacity %d x %d", kf_n, f_n, m->maxQ, std::min(m->maxT, 65535));
    int rc;
    if ((rc = checkFeatureVector(kf_fv_node, kf_fv_start, kf_fv_feature, kf_nfv, kf_n, "keyframe")) != ORBX_OK) return rc;
    if ((rc = checkFeatureVector(f_fv_node, f_fv_start, f_fv_feature, f_nfv, f_n, "frame")) != ORBX_OK) return rc;
    for (int i = 0; i < nOut; ++i) matches[i] = -1;
    if (kf_nfv == 0 || f_nfv == 0 || kf_n == 0 || f_n == 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, kf_desc, kf_n, f_desc, f_n)) != ORBX_OK) return rc;
    const size_t Q = (size_t)m->maxQ, T = (size_t)m->maxT, X = std::max(Q, T);
    if ((rc = m->d_work.ensure(3 * Q + 3 * T + 2 * X + 2 + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_bin.ensure(Q + T)) != ORBX_OK) return rc;
    int *kfNode = m->d_work.p, *kfStart = kfNode + Q, *kfFeat = kfStart + Q + 1, *fNode = kfFeat + Q, *fStart = fNode + T, *fFeat = fStart + T + 1,
        *dMatches = fFeat + T, *binOf = dMatches + X, *result = binOf + X;
    int maxNodeF = 1;
    for (int j = 0; j < f_nfv; ++j) maxNodeF = std::max(maxNodeF, f_fv_start[j + 1] - f_fv_start[j]);
    hipStream_t st = m->stream;
    in.add(kf_fv_node, (size_t)kf_nfv * sizeof(int), kfNode);
    in.add(kf_fv_start, ((size_t)kf_nfv + 1) * sizeof(int), kfStart);
    if (kf_fv_start[kf_nfv] > 0) in.add(kf_fv_feature, (size_t)kf_fv_start[kf_nfv] * sizeof(int), kfFeat);
    in.add(f_fv_node, (size_t)f_nfv * sizeof(int), fNode);
    in.add(f_fv_start, ((size_t)f_nfv + 1) * sizeof(int), fStart);
    if (f_fv_start[f_nfv] > 0) in.add(f_fv_feature, (size_t)f_fv_start[f_nfv] * sizeof(int), fFeat);
    in.add(kf_keys, (size_t)kf_n * sizeof(orbx_keypoint), m->d_kl.p);
    in.add(f_keys, (size_t)f_n * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(kf_valid, (size_t)kf_n, m->d_bin.p);
    if (mode == 1) in.add(f_valid, (size_t)f_n, m->d_bin.p + Q);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchSearchByBoW(st, kfNode, kfStart, kfFeat, kf_nfv, m->d_kl.p, m->d_q.p, m->d_bin.p, fNode, fStart, fFeat, f_nfv, m->d_kr.p,
                                  m->d_t.p, f_n, maxNodeF, nnratio, check_orientation != 0, mode, m->d_bin.p + Q, nOut, dMatches, binOf, result));
    int nmatches = 0;
    HostFetch out(m);
    out.add(matches, dMatches, (size_t)nOut * sizeof(int));
    out.add(&nmatches, result, sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return nmatches;
}

extern "C" int orbx_search_by_bow(orbx_matcher *m, const int32_t *kf_fv_node, const int32_t *kf_fv_start, const int32_t *kf_fv_feature,
                                  int kf_nfv, const orbx_keypoint *kf_keys, const uint8_t *kf_desc, const uint8_t *kf_valid, int kf_n,
                                  const int32_t *f_fv_node, const int32_t *f_fv_start, const int32_t *f_fv_feature, int f_nfv,
                                  const orbx_keypoint *f_keys, const uint8_t *f_desc, int f_n, float nnratio, int check_orientation,
                                  int32_t *matches)
{
    ORBX_API_ENTRY;
    return searchByBowImpl(m, kf_fv_node, kf_fv_start, kf_fv_feature, kf_nfv, kf_keys, kf_desc, kf_valid, kf_n, f_fv_node, f_fv_start,
                           f_fv_feature, f_nfv, f_keys, f_desc, nullptr, f_n, nnratio, check_orientation, 0, matches);
}

// int ORBmatcher::SearchByBoW(KeyFrame *pKF1, KeyFrame *pKF2, vector<MapPoint*> &vpMatches12) — reference src/ORBmatcher2.cc:37-177.
extern "C" int orbx_search_by_bow_keyframes(orbx_matcher *m, const int32_t *fv1_node, const int32_t *fv1_start, const int32_t *fv1_feature,
                                            int nfv1, const orbx_keypoint *keys1, const uint8_t *desc1, const uint8_t *valid1, int n1,
                                            const int32_t *fv2_node, const int32_t *fv2_start, const int32_t *fv2_feature, int nfv2,
                                            const orbx_keypoint *keys2, const uint8_t *desc2, const uint8_t *valid2, int n2, float nnratio,
                                            int check_orientation, int32_t *matches12)
{
    ORBX_API_ENTRY;
    return searchByBowImpl(m, fv1_node, fv1_start, fv1_feature, nfv1, keys1, desc1, valid1, n1, fv2_node, fv2_start, fv2_feature, nfv2, keys2,
                           desc2, valid2, n2, nnratio, check_orientation, 1, matches12);
}

// int ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) — reference src/ORBmatcher2.cc:179-418.
static int checkKeyFrameView(const orbx_keyframe_view *kf, const char *which, int maxN)
{
    if (!kf || kf->n < 0 || (kf->n > 0 && (!kf->keys || !kf->desc || !kf->has_map_point)) || kf->nlevels <= 0 || kf->nlevels > ORBX_MAX_LEVELS ||
        !kf->scale_factors || !kf->level_sigma2 || kf->n_left < -1 || kf->n_left > kf->n)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_triangulation: bad %s keyframe view", which);
    if (kf->n > maxN || kf->n > 65535) return setError(ORBX_ERR_CAPACITY, "%s keyframe: %d features exceed the matcher capacity %d", which, kf->n, std::min(maxN, 65535));
    for (int i = 0; i < kf->n; ++i)
        if (kf->keys[i].octave < 0 || kf->keys[i].octave >= kf->nlevels)
            return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_triangulation: %s keypoint %d has octave %d outside [0, %d)", which, i, kf->keys[i].octave, kf->nlevels);
    return checkFeatureVector(kf->fv_node, kf->fv_start, kf->fv_feature, kf->nfv, kf->n, which);
}

extern "C" int orbx_search_for_triangulation(orbx_matcher *m, const orbx_keyframe_view *kf1, const orbx_keyframe_view *kf2,
                                             const orbx_triangulation_geometry *g, int only_stereo, int coarse, int check_orientation,
                                             int32_t *matches12, int32_t *pairs)
{
    ORBX_API_ENTRY;
    if (!m || !g || !kf1 || !kf2 || (kf1->n > 0 && !matches12)) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_triangulation: null argument");
    if (g->model != 0 && g->model != 1) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_triangulation: model must be 0 (pinhole) or 1 (KannalaBrandt8)");
    if (g->rig && g->model != 1) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_triangulation: a two-camera rig needs the KannalaBrandt8 model");
    int rc;
    if ((rc = checkKeyFrameView(kf1, "first", m->maxQ)) != ORBX_OK) return rc;
    if ((rc = checkKeyFrameView(kf2, "second", m->maxT)) != ORBX_OK) return rc;
    const int n1 = kf1->n, n2 = kf2->n;
    for (int i = 0; i < n1; ++i) matches12[i] = -1;
    if (kf1->nfv == 0 || kf2->nfv == 0 || n1 == 0 || n2 == 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, kf1->desc, n1, kf2->desc, n2)) != ORBX_OK) return rc;
    const size_t Q = (size_t)m->maxQ, T = (size_t)m->maxT;
    if ((rc = m->d_work.ensure(3 * Q + 3 * T + 2 * Q + 2 + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_bin.ensure(Q + T)) != ORBX_OK) return rc;
    if ((rc = m->d_prev.ensure(Q + T + 3 * ORBX_MAX_LEVELS)) != ORBX_OK) return rc;
    int *node1 = m->d_work.p, *start1 = node1 + Q, *feat1 = start1 + Q + 1, *node2 = feat1 + Q, *start2 = node2 + T, *feat2 = start2 + T + 1,
        *dMatches = feat2 + T, *binOf = dMatches + Q, *result = binOf + Q;
    float *uR1 = m->d_prev.p, *uR2 = uR1 + Q, *scale2 = uR2 + T, *sigma1 = scale2 + ORBX_MAX_LEVELS, *sigma2 = sigma1 + ORBX_MAX_LEVELS;
    hipStream_t st = m->stream;
    auto up = [&](void *dst, const void *src, size_t bytes) { in.add(src, bytes, dst); return hipSuccess; };  // (pieces of one packed transfer)
    HIP_TRY(up(node1, kf1->fv_node, (size_t)kf1->nfv * 4)); HIP_TRY(up(start1, kf1->fv_start, ((size_t)kf1->nfv + 1) * 4));
    HIP_TRY(up(feat1, kf1->fv_feature, (size_t)kf1->fv_start[kf1->nfv] * 4));
    HIP_TRY(up(node2, kf2->fv_node, (size_t)kf2->nfv * 4)); HIP_TRY(up(start2, kf2->fv_start, ((size_t)kf2->nfv + 1) * 4));
    HIP_TRY(up(feat2, kf2->fv_feature, (size_t)kf2->fv_start[kf2->nfv] * 4));
    HIP_TRY(up(m->d_kl.p, kf1->keys, (size_t)n1 * sizeof(orbx_keypoint))); HIP_TRY(up(m->d_kr.p, kf2->keys, (size_t)n2 * sizeof(orbx_keypoint)));
    HIP_TRY(up(m->d_bin.p, kf1->has_map_point, (size_t)n1)); HIP_TRY(up(m->d_bin.p + Q, kf2->has_map_point, (size_t)n2));
    if (kf1->u_right) HIP_TRY(up(uR1, kf1->u_right, (size_t)n1 * 4));
    if (kf2->u_right) HIP_TRY(up(uR2, kf2->u_right, (size_t)n2 * 4));
    HIP_TRY(up(scale2, kf2->scale_factors, (size_t)kf2->nlevels * 4));
    HIP_TRY(up(sigma1, kf1->level_sigma2, (size_t)kf1->nlevels * 4)); HIP_TRY(up(sigma2, kf2->level_sigma2, (size_t)kf2->nlevels * 4));
    OrbxSftHost h;
    memset(&h, 0, sizeof h);
    h.node1 = node1; h.start1 = start1; h.feat1 = feat1; h.nfv1 = kf1->nfv; h.keys1 = m->d_kl.p; h.desc1 = m->d_q.p; h.hasMp1 = m->d_bin.p;
    h.uRight1 = kf1->u_right ? uR1 : nullptr; h.nLeft1 = kf1->n_left; h.n1 = n1;
    h.node2 = node2; h.start2 = start2; h.feat2 = feat2; h.nfv2 = kf2->nfv; h.keys2 = m->d_kr.p; h.desc2 = m->d_t.p; h.hasMp2 = m->d_bin.p + Q;
    h.uRight2 = kf2->u_right ? uR2 : nullptr; h.nLeft2 = kf2->n_left;
    h.scale2 = scale2; h.sigma1 = sigma1; h.sigma2 = sigma2;
    memcpy(h.ep, g->ep, sizeof h.ep); memcpy(h.F, g->F12, sizeof h.F);
    h.model = g->model; h.rig = g->rig != 0; h.onlyStereo = only_stereo != 0; h.coarse = coarse != 0; h.checkOri = check_orientation != 0;
    memcpy(h.camL, g->cam_left, sizeof h.camL); memcpy(h.camR, g->cam_right, sizeof h.camR); h.precision = g->precision;
    memcpy(h.R, g->R, sizeof h.R); memcpy(h.t, g->t, sizeof h.t);
    h.matches = dMatches; h.binOf = binOf; h.result = result;
    for (int j = 0; j < kf1->nfv; ++j) h.maxPerNode1 = std::max(h.maxPerNode1, kf1->fv_start[j + 1] - kf1->fv_start[j]);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchSearchForTriangulation(st, h));
    int nmatches = 0;
    HostFetch out(m);
    out.add(matches12, dMatches, (size_t)n1 * sizeof(int));
    out.add(&nmatches, result, sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    if (pairs) {  // vMatchedPairs (:405-413): (idx1, idx2) in increasing idx1
        int k = 0;
        for (int i = 0; i < n1; ++i)
            if (matches12[i] >= 0) { pairs[2 * k] = i; pairs[2 * k + 1] = matches12[i]; ++k; }
    }
    return nmatches;
}

// The window search of ORBmatcher::Fuse (src/ORBmatcher2.cc:420-610, 612-770) and ORBmatcher::SearchBySim3 (src/ORBmatcher3.cc:37-254).
static int keyframeWindowsImpl(const char *fn, orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, const float *u_right, int n,
                               float min_x, float max_x, float min_y, float max_y, const float *inv_level_sigma2, int nlevels,
                               const orbx_window_point *points, const uint8_t *point_desc, int np, int chi2_test, int32_t *best_idx,
                               int32_t *best_dist)
{
    static_assert(sizeof(orbx_window_point) == 24, "orbx_window_point layout");
    if (!m || n < 0 || np < 0 || (n > 0 && (!keys || !desc)) || (np > 0 && (!points || !point_desc || !best_idx || !best_dist)) || nlevels <= 0 ||
        nlevels > ORBX_MAX_LEVELS || (chi2_test && !inv_level_sigma2))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    if (np > m->maxQ || n > m->maxT) return setError(ORBX_ERR_CAPACITY, "%s: %d points x %d keypoints exceed the matcher capacity %d x %d", fn, np, n, m->maxQ, m->maxT);
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, fn);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < n; ++i)
        if (keys[i].octave < 0 || keys[i].octave >= nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: keypoint %d has octave %d outside [0, %d)", fn, i, keys[i].octave, nlevels);
    for (int i = 0; i < np; ++i) { best_idx[i] = -1; best_dist[i] = INT32_MAX; }
    if (n == 0 || np == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, point_desc, np, desc, n)) != ORBX_OK) return rc;
    if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
    if ((rc = m->d_work.ensure(2 * (size_t)m->maxQ + 6 * (size_t)m->maxQ + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_prev.ensure((size_t)m->maxT + ORBX_MAX_LEVELS)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
    int *dIdx = m->d_work.p, *dDist = dIdx + m->maxQ;
    orbx_window_point *dPts = (orbx_window_point *)(dDist + m->maxQ);
    float *dUR = u_right ? m->d_prev.p : nullptr, *dSig = m->d_prev.p + m->maxT;
    hipStream_t st = m->stream;
    in.add(keys, (size_t)n * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(points, (size_t)np * sizeof(orbx_window_point), dPts);
    if (u_right) in.add(u_right, (size_t)n * sizeof(float), dUR);
    if (inv_level_sigma2) in.add(inv_level_sigma2, (size_t)nlevels * sizeof(float), dSig);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchGridBuild(st, m->d_kr.p, n, min_x, max_x, min_y, max_y, cs, it, co));
    HIP_TRY(orbxLaunchKeyframeWindowSearch(st, m->d_kr.p, m->d_t.p, dUR, dSig, cs, it, min_x, max_x, min_y, max_y, dPts, m->d_q.p, np, chi2_test != 0,
                                           dIdx, dDist));
    HostFetch out(m);
    out.add(best_idx, dIdx, (size_t)np * sizeof(int));
    out.add(best_dist, dDist, (size_t)np * sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return ORBX_OK;
}

extern "C" int orbx_search_keyframe_windows(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x,
                                            float max_x, float min_y, float max_y, const float *inv_level_sigma2, int nlevels,
                                            const orbx_window_point *points, const uint8_t *point_desc, int np, int chi2_test, int32_t *best_idx,
                                            int32_t *best_dist)
{
    ORBX_API_ENTRY;
    return keyframeWindowsImpl("orbx_search_keyframe_windows", m, keys, desc, u_right, n, min_x, max_x, min_y, max_y, inv_level_sigma2, nlevels, points,
                               point_desc, np, chi2_test, best_idx, best_dist);
}

extern "C" int orbx_fuse(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, const float *u_right, int n, float min_x, float max_x,
                         float min_y, float max_y, const float *inv_level_sigma2, int nlevels, const orbx_window_point *points,
                         const uint8_t *point_desc, int np, int chi2_test, int32_t *fused_idx)
{
    ORBX_API_ENTRY;
    if (np > 0 && !fused_idx) return setError(ORBX_ERR_INVALID_ARG, "orbx_fuse: null output");
    std::vector<int32_t> dist((size_t)std::max(np, 1));
    const int rc = keyframeWindowsImpl("orbx_fuse", m, keys, desc, u_right, n, min_x, max_x, min_y, max_y, inv_level_sigma2, nlevels, points, point_desc, np,
                                       chi2_test, fused_idx, dist.data());
    if (rc != ORBX_OK) return rc;
    int nFused = 0;
    for (int i = 0; i < np; ++i) {
        if (dist[i] <= 50) ++nFused;  // bestDist <= TH_LOW (:581, :716)
        else fused_idx[i] = -1;
    }
    return nFused;
}

extern "C" int orbx_search_by_sim3(orbx_matcher *m, const orbx_keypoint *keys1, const uint8_t *kf_desc1, int n1, const float *bounds1,
                                   const orbx_keypoint *keys2, const uint8_t *kf_desc2, int n2, const float *bounds2,
                                   const orbx_window_point *points_1in2, const uint8_t *mp_desc1, const orbx_window_point *points_2in1,
                                   const uint8_t *mp_desc2, int nlevels, int32_t *matches12)
{
    ORBX_API_ENTRY;
    if (!bounds1 || !bounds2 || (n1 > 0 && !matches12)) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_sim3: null argument");
    const int TH_HIGH = 100;
    std::vector<int32_t> m1((size_t)std::max(n1, 1)), d1((size_t)std::max(n1, 1)), m2((size_t)std::max(n2, 1)), d2((size_t)std::max(n2, 1));
    int rc = keyframeWindowsImpl("orbx_search_by_sim3", m, keys2, kf_desc2, nullptr, n2, bounds2[0], bounds2[1], bounds2[2], bounds2[3], nullptr, nlevels,
                                 points_1in2, mp_desc1, n1, 0, m1.data(), d1.data());
    if (rc != ORBX_OK) return rc;
    rc = keyframeWindowsImpl("orbx_search_by_sim3", m, keys1, kf_desc1, nullptr, n1, bounds1[0], bounds1[1], bounds1[2], bounds1[3], nullptr, nlevels,
                             points_2in1, mp_desc2, n2, 0, m2.data(), d2.data());
    if (rc != ORBX_OK) return rc;
    int nFound = 0;
    for (int i1 = 0; i1 < n1; ++i1) {  // check agreement (:236-250)
        matches12[i1] = -1;
        const int idx2 = d1[i1] <= TH_HIGH ? m1[i1] : -1;
        if (idx2 >= 0 && d2[idx2] <= TH_HIGH && m2[idx2] == i1) { matches12[i1] = idx2; ++nFound; }
    }
    return nFound;
}

// int ORBmatcher::SearchByProjection(KeyFrame *pKF, Sim3f &Scw, vpPoints, vpMatched, th, ratioHamming) — reference src/ORBmatcher1.cc:429-533
// (:535-648 with vpMatchedKF).  Point i may not take a keypoint that an earlier accepted point took, and nothing else couples the points: the
// choice of point i depends on the points j < i only.  So the window search runs for ALL points at once against "the earliest accepted
// taker of every keypoint" of the previous round and is repeated until no choice changes - after k rounds the first k points are final;
// in practice two or three rounds (a conflict needs two points that want the same keypoint).
static int projectionRoundsImpl(const char *fn, orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x,
                                float min_y, float max_y, int nlevels, const orbx_window_point *points, const uint8_t *point_desc, int np,
                                const uint8_t *matched, float bound, int level_hi, const float *point_angle, int check_orientation,
                                int32_t *new_match)
{
    if (!m || n < 0 || np < 0 || (n > 0 && (!keys || !desc || !new_match)) || (np > 0 && (!points || !point_desc)) || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS)
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    if (np > m->maxQ || n > m->maxT) return setError(ORBX_ERR_CAPACITY, "%s: %d points x %d keypoints exceed the matcher capacity %d x %d", fn, np, n, m->maxQ, m->maxT);
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, fn);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < n; ++i) {
        if (keys[i].octave < 0 || keys[i].octave >= nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: keypoint %d has octave %d outside [0, %d)", fn, i, keys[i].octave, nlevels);
        new_match[i] = -1;
    }
    if (n == 0 || np == 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    // The rounds on the device (k_sbp_candidates in its keyframe mode + k_sbp_resolve: sorted candidate lists, the fixed point of "an
    // earlier accepted point took it" inside one launch, rotation histogram) - unless a window holds more candidates than a list takes:
    // then the rounds of the window search below (a launch and a round trip per round, no limit).
    static const bool hostRounds = getenv("ORBX_PROJECTION_HOST_ROUNDS") != nullptr;  // tests: the fallback
    if (n < 65536 && !hostRounds) {
        if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
        if ((rc = m->d_work.ensure(2 * (size_t)m->maxQ + 3 * (size_t)m->maxT + 2 + 4)) != ORBX_OK) return rc;
        if ((rc = m->d_sbpCand.ensure((size_t)m->maxQ * ORBX_SBP_CAP)) != ORBX_OK) return rc;
        int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
        int *count = m->d_work.p, *choice = count + m->maxQ, *owner = choice + m->maxQ, *result = owner + 2 * m->maxT, *dAssign = result + 2;
        std::vector<orbx_projection> pr((size_t)np);
        for (int i = 0; i < np; ++i) {
            pr[i].u = points[i].u; pr[i].v = points[i].v; pr[i].invz = points[i].radius; pr[i].angle = point_angle ? point_angle[i] : 0.f;
            pr[i].octave = points[i].level; pr[i].flags = points[i].search ? 3 : 0;
        }
        std::vector<int8_t> blocked((size_t)n, 0);
        if (matched) for (int i = 0; i < n; ++i) blocked[i] = matched[i] ? 1 : 0;
        HostPack in(m);
        const size_t oK = in.add(keys, (size_t)n * sizeof(orbx_keypoint)), oD = in.add(desc, (size_t)n * 32), oB = in.add(blocked.data(), (size_t)n);
        const size_t oP = in.add(pr.data(), (size_t)np * sizeof(orbx_projection)), oPd = in.add(point_desc, (size_t)np * 32);
        if ((rc = in.send()) != ORBX_OK) return rc;
        const orbx_keypoint *dK = in.dev<orbx_keypoint>(oK);
        HIP_TRY(orbxLaunchGridBuild(m->stream, dK, n, min_x, max_x, min_y, max_y, cs, it, co));
        HIP_TRY(orbxLaunchSearchByProjection(m->stream, dK, in.dev<uint8_t>(oD), nullptr, in.dev<signed char>(oB), n, cs, it, min_x, max_x, min_y, max_y,
                                             nullptr, 0.f, in.dev<orbx_projection>(oP), in.dev<uint8_t>(oPd), np, 0.f, 0, 0,
                                             check_orientation && point_angle ? 1 : 0, 2, 0.f, m->d_sbpCand.p, count, choice, owner, dAssign, result,
                                             nullptr, level_hi, bound));
        int res[2] = {0, 0};
        HostFetch out(m);
        out.add(res, result, sizeof res);
        out.add(new_match, dAssign, (size_t)n * sizeof(int));
        if ((rc = out.finish()) != ORBX_OK) return rc;
        if (!res[1]) {
            for (int i = 0; i < n; ++i) if (new_match[i] < 0) new_match[i] = -1;  // (-2: removed by the rotation check)
            return res[0];
        }
        for (int i = 0; i < n; ++i) new_match[i] = -1;  // a list overflowed: the rounds below
    }
    HostPack in(m);
    if ((rc = uploadQT(in, point_desc, np, desc, n)) != ORBX_OK) return rc;
    if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
    if ((rc = m->d_work.ensure(8 * (size_t)m->maxQ + (size_t)m->maxT + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_bin.ensure((size_t)m->maxQ + (size_t)m->maxT)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
    int *dIdx = m->d_work.p, *dDist = dIdx + m->maxQ;
    orbx_window_point *dPts = (orbx_window_point *)(dDist + m->maxQ);
    int *dOwner = (int *)(dPts + m->maxQ);
    signed char *dBlocked = m->d_bin.p;
    hipStream_t st = m->stream;
    std::vector<signed char> blocked((size_t)n, 0);
    if (matched) for (int i = 0; i < n; ++i) blocked[i] = matched[i] ? 1 : 0;
    in.add(keys, (size_t)n * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(points, (size_t)np * sizeof(orbx_window_point), dPts);
    in.add(blocked.data(), (size_t)n, dBlocked);
    std::vector<int32_t> idx((size_t)np), dist((size_t)np), choice((size_t)np, -2), owner((size_t)n, -1);
    // round 0's owner array rides with the inputs: the staging still belongs to `in`'s asynchronous copy until the first
    // synchronisation (HostFetch::finish at the end of round 0) - a second HostPack before that would overwrite it under the copy
    in.add(owner.data(), (size_t)n * sizeof(int), dOwner);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchGridBuild(st, m->d_kr.p, n, min_x, max_x, min_y, max_y, cs, it, co));
    for (int round = 0; round <= np; ++round) {
        if (round > 0) {
            HostPack own(m);  // (the staging is free again: every round ends with a synchronised stream)
            own.add(owner.data(), (size_t)n * sizeof(int), dOwner);
            if ((rc = own.send()) != ORBX_OK) return rc;
        }
        HIP_TRY(orbxLaunchKeyframeWindowSearch(st, m->d_kr.p, m->d_t.p, nullptr, nullptr, cs, it, min_x, max_x, min_y, max_y, dPts, m->d_q.p, np, 0, dIdx,
                                               dDist, dBlocked, dOwner, level_hi));
        HostFetch out(m);
        out.add(idx.data(), dIdx, (size_t)np * sizeof(int));
        out.add(dist.data(), dDist, (size_t)np * sizeof(int));
        if ((rc = out.finish()) != ORBX_OK) return rc;
        bool changed = false;
        std::fill(owner.begin(), owner.end(), -1);
        for (int i = 0; i < np; ++i) {  // accepted choices of this round; the earliest taker owns a keypoint
            const int c = idx[i] >= 0 && (float)dist[i] <= bound ? idx[i] : -1;
            if (c != choice[i]) { changed = true; choice[i] = c; }
            if (c >= 0 && owner[c] < 0) owner[c] = i;
        }
        if (!changed) break;
    }
    int nmatches = 0;
    for (int i = 0; i < np; ++i)
        if (choice[i] >= 0 && owner[choice[i]] == i) { new_match[choice[i]] = i; ++nmatches; }
    if (check_orientation && point_angle) {  // rotation consistency (src/ORBmatcher3.cc:544-584)
        const int HISTO_LENGTH = 30;
        const float factor = 1.0f / HISTO_LENGTH;
        std::vector<int> rotHist[HISTO_LENGTH];
        for (int i = 0; i < np; ++i) {  // in the order of the reference's walk over the map points
            if (choice[i] < 0 || owner[choice[i]] != i) continue;
            float rot = point_angle[i] - keys[choice[i]].angle;
            if (rot < 0.0) rot += 360.0f;
            int bin = (int)std::round(rot * factor);
            if (bin == HISTO_LENGTH) bin = 0;
            rotHist[bin].push_back(choice[i]);
        }
        int ind1 = -1, ind2 = -1, ind3 = -1, max1 = 0, max2 = 0, max3 = 0;
        for (int i = 0; i < HISTO_LENGTH; i++) {  // ORBmatcher::ComputeThreeMaxima (src/ORBmatcher3.cc:592-633)
            const int sz = (int)rotHist[i].size();
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0; j < rotHist[i].size(); j++) { new_match[rotHist[i][j]] = -1; nmatches--; }
    }
    return nmatches;
}

extern "C" int orbx_search_by_projection_sim3(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x,
                                              float min_y, float max_y, int nlevels, const orbx_window_point *points, const uint8_t *point_desc,
                                              int np, const uint8_t *matched, float ratio_hamming, int32_t *new_match)
{
    ORBX_API_ENTRY;
    // bestDist <= TH_LOW * ratioHamming (:520), int * float in float
    return projectionRoundsImpl("orbx_search_by_projection_sim3", m, keys, desc, n, min_x, max_x, min_y, max_y, nlevels, points, point_desc, np, matched,
                                50 * ratio_hamming, 0, nullptr, 0, new_match);
}

extern "C" int orbx_search_by_projection_keyframe(orbx_matcher *m, const orbx_keypoint *keys, const uint8_t *desc, int n, float min_x, float max_x,
                                                  float min_y, float max_y, int nlevels, const orbx_window_point *points, const uint8_t *point_desc,
                                                  const float *point_angle, int np, const uint8_t *has_map_point, int orb_dist, int check_orientation,
                                                  int32_t *new_match)
{
    ORBX_API_ENTRY;
    if (check_orientation && np > 0 && !point_angle) return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_projection_keyframe: check_orientation needs point_angle");
    return projectionRoundsImpl("orbx_search_by_projection_keyframe", m, keys, desc, n, min_x, max_x, min_y, max_y, nlevels, points, point_desc, np,
                                has_map_point, (float)orb_dist, 1, point_angle, check_orientation, new_match);
}

// void MapPoint::ComputeDistinctiveDescriptors() — reference src/MapPoint.cc:329-403, batched over map points.
extern "C" int orbx_distinctive_descriptors(orbx_matcher *m, const uint8_t *desc, const int32_t *start, int npoints, int32_t *best)
{
    ORBX_API_ENTRY;
    if (!m || npoints < 0 || (npoints > 0 && (!start || !best))) return setError(ORBX_ERR_INVALID_ARG, "orbx_distinctive_descriptors: bad argument");
    if (npoints == 0) return ORBX_OK;
    if (start[0] != 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_distinctive_descriptors: start[0] must be 0");
    for (int p = 0; p < npoints; ++p) {
        if (start[p + 1] < start[p]) return setError(ORBX_ERR_INVALID_ARG, "orbx_distinctive_descriptors: start is not ascending at point %d", p);
        if (start[p + 1] - start[p] > 65535) return setError(ORBX_ERR_CAPACITY, "orbx_distinctive_descriptors: point %d has more than 65535 observations", p);
    }
    const size_t total = (size_t)start[npoints];
    if (total > 0 && !desc) return setError(ORBX_ERR_INVALID_ARG, "orbx_distinctive_descriptors: null descriptors");
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    int rc;
    if ((rc = m->d_siWork.ensure(total * 32 + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_mat.ensure(2 * (size_t)npoints + 2)) != ORBX_OK) return rc;
    int *dStart = m->d_mat.p, *dBest = dStart + npoints + 1;
    hipStream_t st = m->stream;
    if (total) in.add(desc, total * 32, m->d_siWork.p);
    in.add(start, ((size_t)npoints + 1) * sizeof(int), dStart);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchDistinctive(st, m->d_siWork.p, dStart, npoints, dBest));
    HostFetch out(m);
    out.add(best, dBest, (size_t)npoints * sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return ORBX_OK;
}

// Frame::UndistortKeyPoints — reference src/Frame.cc:777-811.
static int checkCamera(const float *k4, const float *dist, int ndist, const float *knew4, const char *fn)
{
    if (!k4 || !knew4 || ndist < 0 || ndist > 12 || (ndist > 0 && !dist)) return setError(ORBX_ERR_INVALID_ARG, "%s: bad camera arguments", fn);
    if (!(k4[0] != 0.f) || !(k4[1] != 0.f)) return setError(ORBX_ERR_INVALID_ARG, "%s: zero focal length", fn);
    return ORBX_OK;
}

extern "C" int orbx_undistort_keypoints(orbx_matcher *m, const orbx_keypoint *kps, int n, const float *k4, const float *dist, int ndist,
                                        const float *knew4, orbx_keypoint *out)
{
    ORBX_API_ENTRY;
    if (!m || n < 0 || (n > 0 && (!kps || !out))) return setError(ORBX_ERR_INVALID_ARG, "orbx_undistort_keypoints: bad argument");
    int rc = checkCamera(k4, dist, ndist, knew4, "orbx_undistort_keypoints");
    if (rc != ORBX_OK) return rc;
    if (n == 0) return ORBX_OK;
    if (ndist == 0 || dist[0] == 0.0) {  // mvKeysUn = mvKeys (:779-783)
        if (out != kps) memcpy(out, kps, (size_t)n * sizeof(orbx_keypoint));
        return ORBX_OK;
    }
    if (n > m->maxT) return setError(ORBX_ERR_CAPACITY, "%d keypoints exceed the matcher capacity %d", n, m->maxT);
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    in.add(kps, (size_t)n * sizeof(orbx_keypoint), m->d_kr.p);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchUndistort(m->stream, m->d_kr.p, m->d_kl.p, n, k4, dist, ndist, knew4));
    HostFetch res(m);
    res.add(out, m->d_kl.p, (size_t)n * sizeof(orbx_keypoint));
    if ((rc = res.finish()) != ORBX_OK) return rc;
    return ORBX_OK;
}

// ---- projection head of the tracking searches (include/orbx.h "The projection head"; k_project_points) ------------------------------
static int projectArgs(OrbxProjectArgs &a, const char *fn, int mode, const orbx_camera *cam, const float *bounds, float mbf, float viewCosLimit,
                       float logScaleFactor, int nlevels, int farPoints, float thFar)
{
    if (mode != 0 && mode != 1) return setError(ORBX_ERR_INVALID_ARG, "%s: mode must be 0 (motion model) or 1 (in frustum)", fn);
    if (!cam || (cam->type != 0 && cam->type != 1)) return setError(ORBX_ERR_INVALID_ARG, "%s: camera type must be 0 (Pinhole) or 1 (KannalaBrandt8)", fn);
    memset(&a, 0, sizeof a);
    a.mode = mode; a.camType = cam->type; a.poseStride = 16;
    memcpy(a.cam, cam->p, sizeof a.cam);
    if (mode == 1) {
        if (!bounds || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(logScaleFactor > 0.f))
            return setError(ORBX_ERR_INVALID_ARG, "%s: in-frustum mode needs bounds, 0 < nlevels <= %d and log_scale_factor > 0", fn, ORBX_MAX_LEVELS);
        a.minX = bounds[0]; a.maxX = bounds[1]; a.minY = bounds[2]; a.maxY = bounds[3];
        a.mbf = mbf; a.viewCosLimit = viewCosLimit; a.logScaleFactor = logScaleFactor; a.nlevels = nlevels; a.farPoints = farPoints; a.thFar = thFar;
    }
    return ORBX_OK;
}

// out2: the right camera's records of mode 2 (the device buffer holds both halves, maxQ records apart)
static int projectHost(const char *fn, orbx_matcher *m, OrbxProjectArgs &a, const float *pose, const orbx_world_point *points, int np,
                       void *out, void *out2 = nullptr)
{
    static_assert(sizeof(orbx_projection) == sizeof(orbx_window_point), "the projection heads share one device buffer");
    if (!m || np < 0 || (np > 0 && (!points || !out || (a.mode == 2 && !out2)))) return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    if (np == 0) return ORBX_OK;
    if (np > m->maxQ) return setError(ORBX_ERR_CAPACITY, "%s: %d points exceed the matcher capacity %d", fn, np, m->maxQ);
    HIP_TRY(hipSetDevice(m->device));
    int rc = m->d_proj.ensure(2 * (size_t)m->maxQ);
    if (rc != ORBX_OK) return rc;
    HostPack in(m);
    const size_t oP = in.add(pose, (size_t)a.poseStride * sizeof(float)), oW = in.add(points, (size_t)np * sizeof(orbx_world_point));
    if ((rc = in.send()) != ORBX_OK) return rc;
    a.nframes = 1; a.maxNp = np; a.np = np; a.npDev = nullptr;
    a.pose = in.dev<float>(oP); a.points = in.dev<uint8_t>(oW); a.out = m->d_proj.p; a.out2 = m->d_proj.p + m->maxQ;
    HIP_TRY(orbxLaunchProjectPoints(m->stream, &a));
    HostFetch res(m);
    res.add(out, m->d_proj.p, (size_t)np * sizeof(orbx_projection));
    if (a.mode == 2) res.add(out2, m->d_proj.p + m->maxQ, (size_t)np * sizeof(orbx_projection));
    return res.finish();
}

extern "C" int orbx_project_last_frame_points(orbx_matcher *m, const float *q_xyzw, const float *t, const orbx_camera *cam,
                                              const orbx_world_point *points, int np, orbx_projection *proj_out)
{
    ORBX_API_ENTRY;
    if (!q_xyzw || !t) return setError(ORBX_ERR_INVALID_ARG, "orbx_project_last_frame_points: null pose");
    OrbxProjectArgs a;
    int rc = projectArgs(a, "orbx_project_last_frame_points", 0, cam, nullptr, 0.f, 0.f, 0.f, 0, 0, 0.f);
    if (rc != ORBX_OK) return rc;
    float pose[16] = {q_xyzw[0], q_xyzw[1], q_xyzw[2], q_xyzw[3], t[0], t[1], t[2]};
    return projectHost("orbx_project_last_frame_points", m, a, pose, points, np, proj_out);
}

extern "C" int orbx_project_in_frustum(orbx_matcher *m, const float *Rcw, const float *tcw, const float *Ow, const orbx_camera *cam,
                                       const float *bounds, float mbf, float viewing_cos_limit, float log_scale_factor, int nlevels,
                                       int far_points, float th_far_points, const orbx_world_point *points, int np, orbx_projection *proj_out)
{
    ORBX_API_ENTRY;
    if (!Rcw || !tcw || !Ow) return setError(ORBX_ERR_INVALID_ARG, "orbx_project_in_frustum: null pose");
    OrbxProjectArgs a;
    int rc = projectArgs(a, "orbx_project_in_frustum", 1, cam, bounds, mbf, viewing_cos_limit, log_scale_factor, nlevels, far_points, th_far_points);
    if (rc != ORBX_OK) return rc;
    float pose[16] = {0};
    memcpy(pose, Rcw, 9 * sizeof(float)); memcpy(pose + 9, tcw, 3 * sizeof(float)); memcpy(pose + 12, Ow, 3 * sizeof(float));
    return projectHost("orbx_project_in_frustum", m, a, pose, points, np, proj_out);
}

extern "C" int orbx_project_points_frames(orbx_matcher *m, int mode, int nframes, const float *d_pose, const orbx_camera *cam, const float *bounds,
                                          float mbf, float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points,
                                          float th_far_points, const orbx_world_point *d_points, const int32_t *d_np, int max_np,
                                          orbx_projection *d_proj_out)
{
    ORBX_API_ENTRY;
    if (!m || nframes < 0 || max_np < 0 || (nframes > 0 && max_np > 0 && (!d_pose || !d_points || !d_np || !d_proj_out)))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_project_points_frames: bad argument");
    OrbxProjectArgs a;
    int rc = projectArgs(a, "orbx_project_points_frames", mode, cam, bounds, mbf, viewing_cos_limit, log_scale_factor, nlevels, far_points, th_far_points);
    if (rc != ORBX_OK) return rc;
    if (nframes == 0 || max_np == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    a.nframes = nframes; a.maxNp = max_np; a.np = max_np; a.npDev = d_np;
    a.pose = d_pose; a.points = d_points; a.out = d_proj_out;
    HIP_TRY(orbxLaunchProjectPoints(m->stream, &a));
    return ORBX_OK;
}

// Frame::isInFrustum of a fisheye rig: both cameras' records (mode 2 of k_project_points).
static int fisheyeFrustumArgs(OrbxProjectArgs &a, const char *fn, const float *cam1, const float *cam2, const float *bounds, float viewCosLimit,
                              float logScaleFactor, int nlevels, int farPoints, float thFar)
{
    if (!cam1 || !cam2 || !bounds || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(logScaleFactor > 0.f))
        return setError(ORBX_ERR_INVALID_ARG, "%s: needs both cameras, bounds, 0 < nlevels <= %d and log_scale_factor > 0", fn, ORBX_MAX_LEVELS);
    memset(&a, 0, sizeof a);
    a.mode = 2; a.camType = 1; a.poseStride = 40;
    memcpy(a.cam, cam1, sizeof a.cam); memcpy(a.cam2, cam2, sizeof a.cam2);
    a.minX = bounds[0]; a.maxX = bounds[1]; a.minY = bounds[2]; a.maxY = bounds[3];
    a.viewCosLimit = viewCosLimit; a.logScaleFactor = logScaleFactor; a.nlevels = nlevels; a.farPoints = farPoints; a.thFar = thFar;
    return ORBX_OK;
}

extern "C" int orbx_project_in_frustum_fisheye(orbx_matcher *m, const float *Rcw, const float *tcw, const float *Ow, const float *Rrl,
                                               const float *trl, const float *tlr, const float *Rwc, const float *cam1, const float *cam2,
                                               const float *bounds, float viewing_cos_limit, float log_scale_factor, int nlevels, int far_points,
                                               float th_far_points, const orbx_world_point *points, int np, orbx_projection *proj_left_out,
                                               orbx_projection *proj_right_out)
{
    ORBX_API_ENTRY;
    if (!Rcw || !tcw || !Ow || !Rrl || !trl || !tlr || !Rwc) return setError(ORBX_ERR_INVALID_ARG, "orbx_project_in_frustum_fisheye: null pose");
    OrbxProjectArgs a;
    int rc = fisheyeFrustumArgs(a, "orbx_project_in_frustum_fisheye", cam1, cam2, bounds, viewing_cos_limit, log_scale_factor, nlevels, far_points, th_far_points);
    if (rc != ORBX_OK) return rc;
    float pose[40] = {0};
    memcpy(pose, Rcw, 36); memcpy(pose + 9, tcw, 12); memcpy(pose + 12, Ow, 12); memcpy(pose + 15, Rrl, 36); memcpy(pose + 24, trl, 12);
    memcpy(pose + 27, tlr, 12); memcpy(pose + 30, Rwc, 36);
    return projectHost("orbx_project_in_frustum_fisheye", m, a, pose, points, np, proj_left_out, proj_right_out);
}

extern "C" int orbx_project_in_frustum_fisheye_frames(orbx_matcher *m, int nframes, const float *d_pose, const float *cam1, const float *cam2,
                                                      const float *bounds, float viewing_cos_limit, float log_scale_factor, int nlevels,
                                                      int far_points, float th_far_points, const orbx_world_point *d_points, const int32_t *d_np,
                                                      int max_np, orbx_projection *d_proj_left_out, orbx_projection *d_proj_right_out)
{
    ORBX_API_ENTRY;
    if (!m || nframes < 0 || max_np < 0 || (nframes > 0 && max_np > 0 && (!d_pose || !d_points || !d_np || !d_proj_left_out || !d_proj_right_out)))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_project_in_frustum_fisheye_frames: bad argument");
    OrbxProjectArgs a;
    int rc = fisheyeFrustumArgs(a, "orbx_project_in_frustum_fisheye_frames", cam1, cam2, bounds, viewing_cos_limit, log_scale_factor, nlevels, far_points, th_far_points);
    if (rc != ORBX_OK) return rc;
    if (nframes == 0 || max_np == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    a.nframes = nframes; a.maxNp = max_np; a.np = max_np; a.npDev = d_np;
    a.pose = d_pose; a.points = d_points; a.out = d_proj_left_out; a.out2 = d_proj_right_out;
    HIP_TRY(orbxLaunchProjectPoints(m->stream, &a));
    return ORBX_OK;
}

// The head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame): orbx_window_point records (mode 3).
static int windowPointArgs(OrbxProjectArgs &a, const char *fn, const orbx_camera *cam, const float *bounds, float bf, float th,
                           const float *scaleFactors, float logScaleFactor, int nlevels, int tests)
{
    if (!cam || (cam->type != 0 && cam->type != 1)) return setError(ORBX_ERR_INVALID_ARG, "%s: camera type must be 0 (Pinhole) or 1 (KannalaBrandt8)", fn);
    if (!bounds || !scaleFactors || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(logScaleFactor > 0.f) || (tests & ~7))
        return setError(ORBX_ERR_INVALID_ARG, "%s: needs bounds, scale factors, 0 < nlevels <= %d, log_scale_factor > 0 and tests in 0..7", fn, ORBX_MAX_LEVELS);
    memset(&a, 0, sizeof a);
    a.mode = 3; a.camType = cam->type; a.poseStride = 16;
    memcpy(a.cam, cam->p, sizeof a.cam);
    a.minX = bounds[0]; a.maxX = bounds[1]; a.minY = bounds[2]; a.maxY = bounds[3];
    a.mbf = bf; a.th = th; a.logScaleFactor = logScaleFactor; a.nlevels = nlevels; a.tests = tests;
    for (int l = 0; l < nlevels; ++l) a.scaleFactors[l] = scaleFactors[l];
    return ORBX_OK;
}

extern "C" int orbx_project_window_points(orbx_matcher *m, const float *q_xyzw, const float *t, const float *Ow, const orbx_camera *cam,
                                          const float *bounds, float bf, float th, const float *scale_factors, float log_scale_factor, int nlevels,
                                          int tests, const orbx_world_point *points, int np, struct orbx_window_point_s *points_out)
{
    ORBX_API_ENTRY;
    if (!q_xyzw || !t || !Ow) return setError(ORBX_ERR_INVALID_ARG, "orbx_project_window_points: null pose");
    OrbxProjectArgs a;
    int rc = windowPointArgs(a, "orbx_project_window_points", cam, bounds, bf, th, scale_factors, log_scale_factor, nlevels, tests);
    if (rc != ORBX_OK) return rc;
    float pose[16] = {0};
    memcpy(pose, q_xyzw, 16); memcpy(pose + 4, t, 12); memcpy(pose + 7, Ow, 12);
    return projectHost("orbx_project_window_points", m, a, pose, points, np, points_out);
}

extern "C" int orbx_project_window_points_frames(orbx_matcher *m, int nframes, const float *d_pose, const orbx_camera *cam, const float *bounds,
                                                 float bf, float th, const float *scale_factors, float log_scale_factor, int nlevels, int tests,
                                                 const orbx_world_point *d_points, const int32_t *d_np, int max_np, struct orbx_window_point_s *d_points_out)
{
    ORBX_API_ENTRY;
    if (!m || nframes < 0 || max_np < 0 || (nframes > 0 && max_np > 0 && (!d_pose || !d_points || !d_np || !d_points_out)))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_project_window_points_frames: bad argument");
    OrbxProjectArgs a;
    int rc = windowPointArgs(a, "orbx_project_window_points_frames", cam, bounds, bf, th, scale_factors, log_scale_factor, nlevels, tests);
    if (rc != ORBX_OK) return rc;
    if (nframes == 0 || max_np == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    a.nframes = nframes; a.maxNp = max_np; a.np = max_np; a.npDev = d_np;
    a.pose = d_pose; a.points = d_points; a.out = d_points_out;
    HIP_TRY(orbxLaunchProjectPoints(m->stream, &a));
    return ORBX_OK;
}

// Frame::ComputeImageBounds — reference src/Frame.cc:813-845: the four image corners through the same kernel.
extern "C" int orbx_compute_image_bounds(orbx_matcher *m, int cols, int rows, const float *k4, const float *dist, int ndist,
                                         const float *knew4, float *bounds)
{
    ORBX_API_ENTRY;
    if (!m || !bounds || cols <= 0 || rows <= 0) return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_image_bounds: bad argument");
    int rc = checkCamera(k4, dist, ndist, knew4, "orbx_compute_image_bounds");
    if (rc != ORBX_OK) return rc;
    if (ndist == 0 || dist[0] == 0.0) { bounds[0] = 0.0f; bounds[1] = (float)cols; bounds[2] = 0.0f; bounds[3] = (float)rows; return ORBX_OK; }
    orbx_keypoint c[4], u[4];
    memset(c, 0, sizeof c);
    c[1].x = (float)cols; c[2].y = (float)rows; c[3].x = (float)cols; c[3].y = (float)rows;
    if ((rc = orbx_undistort_keypoints(m, c, 4, k4, dist, ndist, knew4, u)) != ORBX_OK) return rc;
    bounds[0] = std::min(u[0].x, u[2].x); bounds[1] = std::max(u[1].x, u[3].x);
    bounds[2] = std::min(u[0].y, u[1].y); bounds[3] = std::max(u[2].y, u[3].y);
    return ORBX_OK;
}

// Frame::AssignFeaturesToGrid — reference src/Frame.cc:387-418 (+ PosInGrid :755-766).
extern "C" int orbx_assign_features_to_grid(orbx_matcher *m, const orbx_keypoint *kps, int n, float min_x, float max_x, float min_y,
                                            float max_y, int32_t *cell_start, int32_t *items)
{
    ORBX_API_ENTRY;
    if (!m || n < 0 || (n > 0 && !kps) || !cell_start || (n > 0 && !items))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_assign_features_to_grid: bad argument");
    if (n > m->maxT) return setError(ORBX_ERR_CAPACITY, "%d keypoints exceed the matcher capacity %d", n, m->maxT);
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, "orbx_assign_features_to_grid");
    if (rc != ORBX_OK) return rc;
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
    HostPack in(m);
    in.add(kps, (size_t)n * sizeof(orbx_keypoint), m->d_kr.p);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchGridBuild(m->stream, m->d_kr.p, n, min_x, max_x, min_y, max_y, cs, it, co));
    // every keypoint is in at most one cell: the first n entries of `items` hold all of them (total <= n)
    std::vector<int32_t> all((size_t)n);
    HostFetch out(m);
    out.add(cell_start, cs, (ORBX_GRID_CELLS + 1) * sizeof(int));
    out.add(all.data(), it, (size_t)n * sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    const int total = cell_start[ORBX_GRID_CELLS];
    if (total > n) return setError(ORBX_ERR_HIP, "orbx_assign_features_to_grid: internal: %d items for %d keypoints", total, n);
    if (total > 0) memcpy(items, all.data(), (size_t)total * sizeof(int));
    return total;
}

// ORBmatcher::SearchForInitialization — reference src/ORBmatcher1.cc:650-765.
extern "C" int orbx_search_for_initialization(orbx_matcher *m, const orbx_keypoint *k1, const uint8_t *d1, int n1, const orbx_keypoint *k2,
                                              const uint8_t *d2, int n2, float min_x, float max_x, float min_y, float max_y,
                                              float *prev_matched, int32_t *matches12, int window_size, float nnratio,
                                              int check_orientation)
{
    ORBX_API_ENTRY;
    if (!m || n1 < 0 || n2 < 0 || (n1 > 0 && (!k1 || !d1 || !prev_matched || !matches12)) || (n2 > 0 && (!k2 || !d2)) || window_size < 0)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_search_for_initialization: bad argument");
    if (n1 > m->maxQ || n2 > m->maxT) return setError(ORBX_ERR_CAPACITY, "%d x %d keypoints exceed the matcher capacity %d x %d", n1, n2, m->maxQ, m->maxT);
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, "orbx_search_for_initialization");
    if (rc != ORBX_OK) return rc;
    if (n1 == 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, d1, n1, d2, n2)) != ORBX_OK) return rc;
    if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
    if ((rc = m->d_work.ensure((size_t)m->maxQ + 2 * (size_t)m->maxT + 4)) != ORBX_OK) return rc;
    if ((rc = m->d_prev.ensure(2 * (size_t)m->maxQ)) != ORBX_OK) return rc;
    if ((rc = m->d_bin.ensure((size_t)m->maxQ)) != ORBX_OK) return rc;
    if ((rc = m->d_siWork.ensure(orbxSearchInitWorkBytes(1, m->maxQ))) != ORBX_OK) return rc;
    int level0of2 = 0;  // the LDS-resident kernel takes F2 subsets of limited size; ORBX_SEARCH_INIT_GENERIC forces the generic one
    for (int i = 0; i < n2; ++i) level0of2 += k2[i].octave == 0;
    int level0of1 = 0;
    for (int i = 0; i < n1; ++i) level0of1 += !(k1[i].octave > 0);
    static const bool forceGeneric = getenv("ORBX_SEARCH_INIT_GENERIC") != nullptr;
    if (forceGeneric) level0of2 = -1;
    int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
    int *m12 = m->d_work.p, *m21 = m12 + m->maxQ, *md = m21 + m->maxT, *nm = md + m->maxT;
    in.add(k1, (size_t)n1 * sizeof(orbx_keypoint), m->d_kl.p);
    if (n2 > 0) in.add(k2, (size_t)n2 * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(prev_matched, 2 * (size_t)n1 * sizeof(float), m->d_prev.p);
    if ((rc = in.send()) != ORBX_OK) return rc;
    if (!(orbxSearchInitInLds(level0of1, n2, level0of2) && n1 < 65536)) HIP_TRY(orbxLaunchGridBuild(m->stream, m->d_kr.p, n2, min_x, max_x, min_y, max_y, cs, it, co));
    HIP_TRY(orbxLaunchSearchInit(m->stream, m->d_kl.p, m->d_q.p, n1, m->d_kr.p, m->d_t.p, n2, cs, it, min_x, max_x, min_y, max_y,
                                 m->d_prev.p, m12, m21, md, m->d_bin.p, nm, window_size, nnratio, check_orientation, level0of2, m->d_siWork.p, level0of1));
    int nmatches = 0;
    HostFetch out(m);
    out.add(matches12, m12, (size_t)n1 * sizeof(int));
    out.add(prev_matched, m->d_prev.p, 2 * (size_t)n1 * sizeof(float));
    out.add(&nmatches, nm, sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return nmatches;
}

// ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) — reference src/ORBmatcher3.cc:256-467 (Nleft == -1).
static int searchByProjectionImpl(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, const float *u_right,
                                  const int8_t *cur_state, int nc, float min_x, float max_x, float min_y, float max_y,
                                  const float *scale_factors, int nlevels, float mbf, const orbx_projection *proj,
                                  const uint8_t *proj_desc, int np, float th, int forward, int backward, int check_orientation,
                                  int mode, float nnratio, int32_t *assign)
{
    if (!m || nc < 0 || np < 0 || (nc > 0 && (!kc || !dc || !cur_state || !assign)) || (np > 0 && (!proj || !proj_desc)) || !scale_factors ||
        nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(th >= 0.f))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_search_by_projection: bad argument");
    if (np > m->maxQ || nc > m->maxT) return setError(ORBX_ERR_CAPACITY, "%d x %d exceeds the matcher capacity %d x %d", np, nc, m->maxQ, m->maxT);
    if (nc >= 65536) return setError(ORBX_ERR_CAPACITY, "orbx_search_by_projection: at most 65535 keypoints in the current frame");
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, "orbx_search_by_projection");
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < np; ++i)
        if ((proj[i].flags & 1) && (proj[i].octave < 0 || proj[i].octave >= nlevels))
            return setError(ORBX_ERR_INVALID_ARG, "projection %d has octave %d outside [0,%d)", i, proj[i].octave, nlevels);
    if (nc == 0) return 0;
    static_assert(sizeof(orbx_projection) == 24, "orbx_projection layout");
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = m->d_grid.ensure((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT)) != ORBX_OK) return rc;
    // work: count[maxQ], choice[maxQ], owner[2 * maxT], result[2] + assign[maxT] (one piece: they leave together)
    const size_t workInts = 2 * (size_t)m->maxQ + 3 * (size_t)m->maxT + 2 + 4;
    if ((rc = m->d_work.ensure(workInts)) != ORBX_OK) return rc;
    if ((rc = m->d_sbpCand.ensure((size_t)m->maxQ * ORBX_SBP_CAP)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + ORBX_GRID_CELLS + 1, *co = it + m->maxT;
    int *count = m->d_work.p, *choice = count + m->maxQ, *owner = choice + m->maxQ, *result = owner + 2 * m->maxT, *dAssign = result + 2;
    HostPack in(m);
    const size_t oKc = in.add(kc, (size_t)nc * sizeof(orbx_keypoint)), oDc = in.add(dc, (size_t)nc * 32), oState = in.add(cur_state, (size_t)nc);
    const size_t oUr = u_right ? in.add(u_right, (size_t)nc * sizeof(float)) : 0;
    const size_t oProj = in.add(proj, (size_t)np * sizeof(orbx_projection)), oPd = in.add(proj_desc, (size_t)np * 32);
    const size_t oScale = in.add(scale_factors, (size_t)nlevels * sizeof(float));
    if ((rc = in.send()) != ORBX_OK) return rc;
    const orbx_keypoint *dKc = in.dev<orbx_keypoint>(oKc);
    HIP_TRY(orbxLaunchGridBuild(m->stream, dKc, nc, min_x, max_x, min_y, max_y, cs, it, co));
    HIP_TRY(orbxLaunchSearchByProjection(m->stream, dKc, in.dev<uint8_t>(oDc), u_right ? in.dev<float>(oUr) : nullptr, in.dev<signed char>(oState), nc,
                                         cs, it, min_x, max_x, min_y, max_y, in.dev<float>(oScale), mbf, in.dev<orbx_projection>(oProj),
                                         in.dev<uint8_t>(oPd), np, th, forward != 0, backward != 0, check_orientation != 0, mode, nnratio,
                                         m->d_sbpCand.p, count, choice, owner, dAssign, result));
    int res[2] = {0, 0};
    HostFetch out(m);
    out.add(res, result, sizeof res);
    out.add(assign, dAssign, (size_t)nc * sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    if (res[1]) return setError(ORBX_ERR_CAPACITY, "orbx_search_by_projection: a search window holds more than %d candidates", ORBX_SBP_CAP);
    return res[0];
}

extern "C" int orbx_search_by_projection(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, const float *u_right,
                                         const int8_t *cur_state, int nc, float min_x, float max_x, float min_y, float max_y,
                                         const float *scale_factors, int nlevels, float mbf, const orbx_projection *proj,
                                         const uint8_t *proj_desc, int np, float th, int forward, int backward, int check_orientation,
                                         int32_t *assign)
{
    ORBX_API_ENTRY;
    return searchByProjectionImpl(m, kc, dc, u_right, cur_state, nc, min_x, max_x, min_y, max_y, scale_factors, nlevels, mbf, proj, proj_desc,
                                  np, th, forward, backward, check_orientation, 0, 0.f, assign);
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th, ..) — reference src/ORBmatcher1.cc:45-215 (Nleft == -1).
extern "C" int orbx_search_by_projection_points(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, const float *u_right,
                                                const int8_t *cur_state, int nc, float min_x, float max_x, float min_y, float max_y,
                                                const float *scale_factors, int nlevels, const orbx_projection *map_points,
                                                const uint8_t *map_point_desc, int np, float th, float nnratio, int32_t *assign)
{
    ORBX_API_ENTRY;
    return searchByProjectionImpl(m, kc, dc, u_right, cur_state, nc, min_x, max_x, min_y, max_y, scale_factors, nlevels, 0.f, map_points,
                                  map_point_desc, np, th, 0, 0, 0, 1, nnratio, assign);
}

// The CurrentFrame.Nleft != -1 branches (KB8 fisheye rig) of both SearchByProjection functions — reference src/ORBmatcher3.cc:300-436 and
// src/ORBmatcher1.cc:45-215.
static int searchByProjectionFisheyeImpl(const char *fn, orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, int n_left, int n_right,
                                         const int8_t *cur_state, float min_x, float max_x, float min_y, float max_y,
                                         const float *scale_factors, int nlevels, const orbx_projection *pl, const orbx_projection *pr,
                                         const uint8_t *proj_desc, int np, const int32_t *left_to_right, const int32_t *right_to_left,
                                         float th, int forward, int backward, int check_orientation, int mode, float nnratio, int32_t *assign)
{
    const long long ncl = (long long)n_left + n_right;
    if (!m || n_left < 0 || n_right < 0 || np < 0 || (ncl > 0 && (!kc || !dc || !cur_state || !assign)) || (np > 0 && (!pl || !pr || !proj_desc)) ||
        !scale_factors || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(th >= 0.f) || (mode == 1 && ((n_left > 0 && !left_to_right) || (n_right > 0 && !right_to_left))))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    if (2LL * np > m->maxQ || ncl > m->maxT)
        return setError(ORBX_ERR_CAPACITY, "%s: %d steps x %lld keypoints exceed the matcher capacity %d x %d", fn, 2 * np, ncl, m->maxQ, m->maxT);
    if (ncl >= 65536) return setError(ORBX_ERR_CAPACITY, "%s: at most 65535 keypoints in the current frame", fn);
    const int nc = (int)ncl;
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, fn);
    if (rc != ORBX_OK) return rc;
    // the steps of point i: 2 i = its left record, 2 i + 1 = its right record
    std::vector<orbx_projection> steps(2 * (size_t)np);
    for (int i = 0; i < np; ++i) {
        orbx_projection L = pl[i], R = pr[i];
        if (mode == 0) { R.octave = L.octave; R.angle = L.angle; R.flags = L.flags; R.invz = L.invz; }  // one keypoint of the last frame behind both
        else { if (R.octave == -1) R.flags &= ~1; R.flags = (R.flags & ~2) | (L.flags & 2); }          // one map point behind both
        if (((L.flags & 1) && (L.octave < 0 || L.octave >= nlevels)) || ((R.flags & 1) && (R.octave < 0 || R.octave >= nlevels)))
            return setError(ORBX_ERR_INVALID_ARG, "%s: point %d has a level outside [0,%d)", fn, i, nlevels);
        steps[2 * (size_t)i] = L; steps[2 * (size_t)i + 1] = R;
    }
    std::vector<int> partner;
    if (mode == 1) {
        partner.assign((size_t)nc, -1);
        for (int i = 0; i < n_left; ++i) {
            if (left_to_right[i] < -1 || left_to_right[i] >= n_right) return setError(ORBX_ERR_INVALID_ARG, "%s: left_to_right[%d] = %d", fn, i, left_to_right[i]);
            if (left_to_right[i] >= 0) partner[i] = n_left + left_to_right[i];
        }
        for (int i = 0; i < n_right; ++i) {
            if (right_to_left[i] < -1 || right_to_left[i] >= n_left) return setError(ORBX_ERR_INVALID_ARG, "%s: right_to_left[%d] = %d", fn, i, right_to_left[i]);
            if (right_to_left[i] >= 0) partner[(size_t)n_left + i] = right_to_left[i];
        }
    }
    if (nc == 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, proj_desc, np, dc, nc)) != ORBX_OK) return rc;
    const size_t gridInts = (size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)m->maxT;  // cellStart, items, cellOf of one grid
    if ((rc = m->d_grid.ensure(2 * gridInts)) != ORBX_OK) return rc;
    // work: count[maxQ], choice[maxQ], next[2 maxQ], head[maxT], partner[maxT], assign[maxT], result[2], curState (maxT bytes)
    const size_t workInts = 4 * (size_t)m->maxQ + 3 * (size_t)m->maxT + 2 + ((size_t)m->maxT + 3) / 4 + 4;
    if ((rc = m->d_work.ensure(workInts)) != ORBX_OK) return rc;
    if ((rc = m->d_sbpCand.ensure((size_t)m->maxQ * ORBX_SBP_CAP)) != ORBX_OK) return rc;
    if ((rc = m->d_proj.ensure((size_t)m->maxQ)) != ORBX_OK) return rc;
    int *csL = m->d_grid.p, *itL = csL + ORBX_GRID_CELLS + 1, *coL = itL + m->maxT;
    int *csR = m->d_grid.p + gridInts, *itR = csR + ORBX_GRID_CELLS + 1, *coR = itR + m->maxT;
    int *count = m->d_work.p, *choice = count + m->maxQ, *next = choice + m->maxQ, *head = next + 2 * (size_t)m->maxQ, *dPartner = head + m->maxT,
        *dAssign = dPartner + m->maxT, *result = dAssign + m->maxT;
    signed char *dState = (signed char *)(result + 2);
    in.add(kc, (size_t)nc * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(cur_state, (size_t)nc, dState);
    if (np > 0) in.add(steps.data(), steps.size() * sizeof(orbx_projection), m->d_proj.p);
    if (mode == 1) in.add(partner.data(), (size_t)nc * sizeof(int), dPartner);
    in.add(scale_factors, (size_t)nlevels * sizeof(float), m->d_scale.p);
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchGridBuild(m->stream, m->d_kr.p, n_left, min_x, max_x, min_y, max_y, csL, itL, coL));
    HIP_TRY(orbxLaunchGridBuild(m->stream, m->d_kr.p + n_left, n_right, min_x, max_x, min_y, max_y, csR, itR, coR));
    OrbxSbpFisheye f;
    f.kc = m->d_kr.p; f.dc = m->d_t.p; f.curState = dState; f.nLeft = n_left; f.nRight = n_right;
    f.cellStartL = csL; f.itemsL = itL; f.cellStartR = csR; f.itemsR = itR;
    f.minX = min_x; f.maxX = max_x; f.minY = min_y; f.maxY = max_y; f.scale = m->d_scale.p;
    f.steps = m->d_proj.p; f.dproj = m->d_q.p; f.np = np;
    f.th = th; f.forward = forward != 0; f.backward = backward != 0; f.checkOri = check_orientation != 0; f.mode = mode; f.nnratio = nnratio;
    f.partner = mode == 1 ? dPartner : nullptr;
    f.cand = m->d_sbpCand.p; f.count = count; f.choice = choice; f.head = head; f.next = next; f.assign = dAssign; f.result = result;
    HIP_TRY(orbxLaunchSearchByProjectionFisheye(m->stream, f));
    int res[2] = {0, 0};
    HostFetch out(m);
    out.add(assign, dAssign, (size_t)nc * sizeof(int));
    out.add(res, result, sizeof res);
    if ((rc = out.finish()) != ORBX_OK) return rc;  // (also keeps `steps` / `partner` alive until the uploads have been consumed)
    if (res[1]) return setError(ORBX_ERR_CAPACITY, "%s: a search window holds more than %d candidates", fn, ORBX_SBP_CAP);
    return res[0];
}

extern "C" int orbx_search_by_projection_fisheye(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, int n_left, int n_right,
                                                 const int8_t *cur_state, float min_x, float max_x, float min_y, float max_y,
                                                 const float *scale_factors, int nlevels, const orbx_projection *proj_left,
                                                 const orbx_projection *proj_right, const uint8_t *proj_desc, int np, float th, int forward,
                                                 int backward, int check_orientation, int32_t *assign)
{
    ORBX_API_ENTRY;
    return searchByProjectionFisheyeImpl("orbx_search_by_projection_fisheye", m, kc, dc, n_left, n_right, cur_state, min_x, max_x, min_y, max_y,
                                         scale_factors, nlevels, proj_left, proj_right, proj_desc, np, nullptr, nullptr, th, forward, backward,
                                         check_orientation, 0, 0.f, assign);
}

extern "C" int orbx_search_by_projection_points_fisheye(orbx_matcher *m, const orbx_keypoint *kc, const uint8_t *dc, int n_left, int n_right,
                                                        const int8_t *cur_state, float min_x, float max_x, float min_y, float max_y,
                                                        const float *scale_factors, int nlevels, const orbx_projection *map_points_left,
                                                        const orbx_projection *map_points_right, const uint8_t *map_point_desc, int np,
                                                        const int32_t *left_to_right, const int32_t *right_to_left, float th, float nnratio,
                                                        int32_t *assign)
{
    ORBX_API_ENTRY;
    return searchByProjectionFisheyeImpl("orbx_search_by_projection_points_fisheye", m, kc, dc, n_left, n_right, cur_state, min_x, max_x, min_y,
                                         max_y, scale_factors, nlevels, map_points_left, map_points_right, map_point_desc, np, left_to_right,
                                         right_to_left, th, 0, 0, 0, 1, nnratio, assign);
}

// Frame::ComputeStereoMatches, complete — reference src/Frame.cc:841-1011.
extern "C" int orbx_compute_stereo_matches(orbx_matcher *m, orbx_extractor *left, int frame_left, orbx_extractor *right, int frame_right,
                                           const orbx_keypoint *kl, const uint8_t *dl, int nl, const orbx_keypoint *kr, const uint8_t *dr,
                                           int nr, float mb, float mbf, float *u_right, float *depth)
{
    ORBX_API_ENTRY;
    if (!m || !left || !right || nl < 0 || nr < 0 || (nl > 0 && (!kl || !dl)) || (nr > 0 && (!kr || !dr)) || !u_right || !depth)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches: bad argument");
    if (frame_left < 0 || frame_left >= left->lastFrames || frame_right < 0 || frame_right >= right->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches: frame index outside the extractors' last batch");
    if (left->prm.device != m->device || right->prm.device != m->device)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches: matcher and extractors must live on the same device");
    if (left->prm.nlevels != right->prm.nlevels || left->curW != right->curW || left->curH != right->curH)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches: left / right extractors differ in geometry");
    if (nl == 0) return ORBX_OK;
    const int nlev = left->prm.nlevels;
    for (int i = 0; i < nl; ++i)
        if (kl[i].octave < 0 || kl[i].octave >= nlev) return setError(ORBX_ERR_INVALID_ARG, "left keypoint %d has octave %d", i, kl[i].octave);
    for (int i = 0; i < nr; ++i)
        if (kr[i].octave < 0 || kr[i].octave >= nlev) return setError(ORBX_ERR_INVALID_ARG, "right keypoint %d has octave %d", i, kr[i].octave);
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    // the pyramids must be complete before the matcher stream reads them
    HIP_TRY(hipStreamSynchronize(left->stream));
    HIP_TRY(hipStreamSynchronize(right->stream));
    int rc = uploadQT(in, dl, nl, dr, nr);
    if (rc != ORBX_OK) return rc;
    if ((rc = m->d_uRight.ensure(m->maxQ)) != ORBX_OK || (rc = m->d_depth.ensure(m->maxQ)) != ORBX_OK || (rc = m->d_sad.ensure(m->maxQ)) != ORBX_OK)
        return rc;
    in.add(kl, (size_t)nl * sizeof(orbx_keypoint), m->d_kl.p);
    if (nr > 0) in.add(kr, (size_t)nr * sizeof(orbx_keypoint), m->d_kr.p);
    in.add(left->scale.data(), (size_t)nlev * sizeof(float), m->d_scale.p);
    const float minZ = mb, minD = 0.f, maxD = mbf / minZ;
    const int nRows = left->geom[0].h;  // mvImagePyramid[0].rows (:848)
    const int itemStride = orbxStereoBandItems(nr, left->scale.back());
    if ((rc = m->d_rowStart.ensure((size_t)nRows + 1)) != ORBX_OK || (rc = m->d_rowItems.ensure((size_t)std::max(itemStride, 1))) != ORBX_OK) return rc;
    if ((rc = in.send()) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchStereoRows(m->stream, m->d_kl.p, m->d_q.p, nl, m->d_kr.p, m->d_t.p, nr, nRows, m->d_scale.p, minD, maxD,
                                 m->d_idx.p, m->d_dist.p, nullptr, m->d_rowStart.p, m->d_rowItems.p, itemStride));
    OrbxStereoLevels lv;
    memset(&lv, 0, sizeof lv);
    for (int l = 0; l < nlev; ++l) {
        const OrbxLevelGeom &g = left->geom[l];
        lv.left[l] = l == 0 ? left->lastSrc + (size_t)frame_left * left->lastSrcFrameStride : left->d_raw[l].p + (size_t)frame_left * g.frameStride;
        lv.right[l] = l == 0 ? right->lastSrc + (size_t)frame_right * right->lastSrcFrameStride : right->d_raw[l].p + (size_t)frame_right * g.frameStride;
        lv.pitchL[l] = l == 0 ? left->lastSrcPitch : g.pitch;
        lv.pitchR[l] = l == 0 ? right->lastSrcPitch : g.pitch;
        lv.colsR[l] = g.w;
        lv.scale[l] = left->scale[l];
        lv.invScale[l] = left->invScale[l];
    }
    HIP_TRY(orbxLaunchStereoRefine(m->stream, &lv, m->d_kl.p, nl, m->d_kr.p, m->d_idx.p, m->d_dist.p, mbf, minD, maxD, m->d_uRight.p,
                                   m->d_depth.p, m->d_sad.p, nullptr));
    HostFetch out(m);
    out.add(u_right, m->d_uRight.p, (size_t)nl * sizeof(float));
    out.add(depth, m->d_depth.p, (size_t)nl * sizeof(float));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return ORBX_OK;
}

// ---------------------------------------------------------------------------------------------------------
// Device-resident, batched forms: pair p = frame p of the two extractors' last batches.  Nothing crosses PCIe; the
// matcher's stream is ordered behind the extractors' work by events, and the extractors' next batch behind the matcher.
// ---------------------------------------------------------------------------------------------------------
static int orderAfterExtractors(orbx_matcher *m, orbx_extractor *a, orbx_extractor *b)
{
    if (!m->evWait) HIP_TRY(hipEventCreateWithFlags(&m->evWait, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(m->evWait, a->stream));
    HIP_TRY(hipStreamWaitEvent(m->stream, m->evWait, 0));
    if (b != a) {
        HIP_TRY(hipEventRecord(m->evWait, b->stream));
        HIP_TRY(hipStreamWaitEvent(m->stream, m->evWait, 0));
    }
    return ORBX_OK;
}
static int orderExtractorsAfter(orbx_matcher *m, orbx_extractor *a, orbx_extractor *b)
{
    HIP_TRY(hipEventRecord(m->evWait, m->stream));
    HIP_TRY(hipStreamWaitEvent(a->stream, m->evWait, 0));
    if (b != a) HIP_TRY(hipStreamWaitEvent(b->stream, m->evWait, 0));
    return ORBX_OK;
}
static int checkFramePairs(const char *fn, orbx_matcher *m, orbx_extractor *a, orbx_extractor *b, int npairs)
{
    if (!m || !a || !b) return setError(ORBX_ERR_INVALID_ARG, "%s: null handle", fn);
    if (a->prm.device != m->device || b->prm.device != m->device)
        return setError(ORBX_ERR_INVALID_ARG, "%s: matcher and extractors must live on the same device", fn);
    if (npairs < 0 || npairs > a->lastFrames || npairs > b->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "%s: %d pairs, but the extractors' last batches hold %d / %d frames", fn, npairs, a->lastFrames, b->lastFrames);
    return ORBX_OK;
}

extern "C" int orbx_match_knn2_frames(orbx_matcher *m, orbx_extractor *query, orbx_extractor *train, int npairs, int lapping_only,
                                      void *d_idx, void *d_dist)
{
    ORBX_API_ENTRY;
    int rc = checkFramePairs("orbx_match_knn2_frames", m, query, train, npairs);
    if (rc != ORBX_OK) return rc;
    if (!d_idx || !d_dist) return setError(ORBX_ERR_INVALID_ARG, "orbx_match_knn2_frames: null output");
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    if ((rc = orderAfterExtractors(m, query, train)) != ORBX_OK) return rc;
    const int capQ = query->maxKeypoints, capT = train->maxKeypoints;
    rc = knn2Launch(m, npairs, query->d_desc.p, (long long)capQ * 32, query->d_n.p, lapping_only ? query->d_mono.p : nullptr, capQ,
                    train->d_desc.p, (long long)capT * 32, train->d_n.p, lapping_only ? train->d_mono.p : nullptr, capT, (int *)d_idx, (int *)d_dist);
    if (rc != ORBX_OK) return rc;
    return orderExtractorsAfter(m, query, train);
}

extern "C" int orbx_compute_stereo_matches_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int npairs, float mb, float mbf,
                                                  void *d_u_right, void *d_depth)
{
    ORBX_API_ENTRY;
    int rc = checkFramePairs("orbx_compute_stereo_matches_frames", m, left, right, npairs);
    if (rc != ORBX_OK) return rc;
    if (!d_u_right || !d_depth) return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches_frames: null output");
    if (left->prm.nlevels != right->prm.nlevels || left->curW != right->curW || left->curH != right->curH)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_compute_stereo_matches_frames: left / right extractors differ in geometry");
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const int nlev = left->prm.nlevels, capL = left->maxKeypoints, capR = right->maxKeypoints;
    const size_t nOut = (size_t)npairs * capL;
    if ((rc = m->d_bIdx.ensure(nOut)) != ORBX_OK || (rc = m->d_bDist.ensure(nOut)) != ORBX_OK || (rc = m->d_bSad.ensure(nOut)) != ORBX_OK) return rc;
    if ((rc = orderAfterExtractors(m, left, right)) != ORBX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_scale.p, left->scale.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, m->stream));
    const float minZ = mb, minD = 0.f, maxD = mbf / minZ;
    OrbxStereoBatch sb;
    sb.npairs = npairs; sb.kpStrideL = capL; sb.kpStrideR = capR; sb.outStride = capL; sb.nlDev = left->d_n.p; sb.nrDev = right->d_n.p;
    const int nRowsB = left->geom[0].h, itemStride = orbxStereoBandItems(capR, left->scale.back());
    if ((rc = m->d_rowStart.ensure((size_t)npairs * (nRowsB + 1))) != ORBX_OK || (rc = m->d_rowItems.ensure((size_t)npairs * itemStride)) != ORBX_OK) return rc;
    HIP_TRY(orbxLaunchStereoRows(m->stream, left->d_kps.p, left->d_desc.p, capL, right->d_kps.p, right->d_desc.p, capR, nRowsB, m->d_scale.p,
                                 minD, maxD, m->d_bIdx.p, m->d_bDist.p, &sb, m->d_rowStart.p, m->d_rowItems.p, itemStride));
    OrbxStereoLevels lv;
    memset(&lv, 0, sizeof lv);
    for (int l = 0; l < nlev; ++l) {
        const OrbxLevelGeom &g = left->geom[l];
        lv.left[l] = l == 0 ? left->lastSrc : left->d_raw[l].p;
        lv.right[l] = l == 0 ? right->lastSrc : right->d_raw[l].p;
        lv.pitchL[l] = l == 0 ? left->lastSrcPitch : g.pitch;
        lv.pitchR[l] = l == 0 ? right->lastSrcPitch : g.pitch;
        lv.frameStrideL[l] = l == 0 ? left->lastSrcFrameStride : g.frameStride;
        lv.frameStrideR[l] = l == 0 ? right->lastSrcFrameStride : g.frameStride;
        lv.colsR[l] = g.w;
        lv.scale[l] = left->scale[l];
        lv.invScale[l] = left->invScale[l];
    }
    HIP_TRY(orbxLaunchStereoRefine(m->stream, &lv, left->d_kps.p, capL, right->d_kps.p, m->d_bIdx.p, m->d_bDist.p, mbf, minD, maxD, (float *)d_u_right,
                                   (float *)d_depth, m->d_bSad.p, &sb));
    return orderExtractorsAfter(m, left, right);
}

// Tracking::TrackWithMotionModel's matcher over a batch of frames, device-resident end to end: the extractor's keypoints ->
// Frame::UndistortKeyPoints -> Frame::AssignFeaturesToGrid -> ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono)
// (reference src/Frame.cc:777-811, 387-418; src/ORBmatcher3.cc:256-467, Nleft == -1).  Frame p of the batch = frame p of the
// extractor's last batch; the projected map points of "its" last frame come as device arrays.  Six launches for the whole batch.
static int searchByProjectionFramesImpl(const char *fn, int mode, float nnratio, orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4,
                                        const float *dist, int ndist, const float *knew4, float min_x, float max_x, float min_y, float max_y, float mbf,
                                        const float *d_u_right, const int8_t *d_cur_state, const orbx_projection *d_proj,
                                        const uint8_t *d_proj_desc, const int32_t *d_np, int max_np, float th, int forward, int backward,
                                        int check_orientation, void *d_kps_un, void *d_assign, void *d_nmatches)
{
    if (!m || !ex) return setError(ORBX_ERR_INVALID_ARG, "%s: null handle", fn);
    if (ex->prm.device != m->device) return setError(ORBX_ERR_INVALID_ARG, "%s: matcher and extractor must live on the same device", fn);
    if (nframes < 0 || nframes > ex->lastFrames) return setError(ORBX_ERR_INVALID_ARG, "%s: %d frames, the extractor's last batch holds %d", fn, nframes, ex->lastFrames);
    if (max_np < 0 || (max_np > 0 && (!d_proj || !d_proj_desc)) || !d_np || !d_assign || !d_nmatches || !(th >= 0.f))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    int rc = checkCamera(k4, dist, ndist, knew4 ? knew4 : k4, fn);
    if (rc != ORBX_OK) return rc;
    if ((rc = checkGridBounds(min_x, max_x, min_y, max_y, fn)) != ORBX_OK) return rc;
    const int cap = ex->maxKeypoints, nlev = ex->prm.nlevels;
    if (cap >= 65536) return setError(ORBX_ERR_CAPACITY, "%s: at most 65535 keypoints per frame", fn);
    if (nframes == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t F = (size_t)nframes;
    // scratch: undistorted keypoints (unless the caller wants them), grid CSR + cellOf, candidate lists, resolve state
    if (!d_kps_un) {
        if ((rc = m->d_bKps.ensure(F * cap)) != ORBX_OK) return rc;
        d_kps_un = m->d_bKps.p;
    }
    if ((rc = m->d_grid.ensure(F * ((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)cap))) != ORBX_OK) return rc;
    const size_t workInts = F * (2 * (size_t)std::max(max_np, 1) + 2 * (size_t)cap + 2);
    if ((rc = m->d_work.ensure(workInts)) != ORBX_OK) return rc;
    if ((rc = m->d_sbpCand.ensure(F * (size_t)std::max(max_np, 1) * ORBX_SBP_CAP)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + F * (ORBX_GRID_CELLS + 1), *co = it + F * cap;
    int *count = m->d_work.p, *choice = count + F * std::max(max_np, 1), *owner = choice + F * std::max(max_np, 1), *result = owner + F * 2 * cap;
    if ((rc = orderAfterExtractors(m, ex, ex)) != ORBX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(m->d_scale.p, ex->scale.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, m->stream));
    if (ndist == 0 || dist[0] == 0.0)  // mvKeysUn = mvKeys (src/Frame.cc:779-783)
        HIP_TRY(hipMemcpyAsync(d_kps_un, ex->d_kps.p, F * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToDevice, m->stream));
    else
        HIP_TRY(orbxLaunchUndistort(m->stream, ex->d_kps.p, d_kps_un, cap, k4, dist, ndist, knew4 ? knew4 : k4, nframes, cap, cap, ex->d_n.p));
    HIP_TRY(orbxLaunchGridBuild(m->stream, d_kps_un, cap, min_x, max_x, min_y, max_y, cs, it, co, nframes, cap, ex->d_n.p));
    OrbxSbpBatch b;
    b.nframes = nframes; b.kcStride = cap; b.projStride = max_np; b.ncDev = ex->d_n.p; b.npDev = d_np;
    HIP_TRY(orbxLaunchSearchByProjection(m->stream, d_kps_un, ex->d_desc.p, d_u_right, d_cur_state, cap, cs, it, min_x, max_x, min_y, max_y,
                                         m->d_scale.p, mbf, d_proj, d_proj_desc, max_np, th, forward != 0, backward != 0, check_orientation != 0, mode, nnratio,
                                         m->d_sbpCand.p, count, choice, owner, (int *)d_assign, result, &b));
    // nmatches[p] = result[2 p]; a window with more than SBP_CAP candidates is reported as -1 for that frame
    HIP_TRY(orbxLaunchGatherResults(m->stream, result, nframes, (int *)d_nmatches));
    return orderExtractorsAfter(m, ex, ex);
}

// KeyFrame::GetFeaturesInArea window searches (Fuse, SearchBySim3's halves) for a BATCH of keyframes = the frames of an extractor's last
// batch: undistort -> grid -> one launch of the window search over (point, keyframe).  fuse: bestDist <= TH_LOW decides on the device.
static int keyframeWindowsFramesImpl(const char *fn, bool fuse, orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist,
                                     int ndist, const float *knew4, float min_x, float max_x, float min_y, float max_y, const float *d_u_right,
                                     const orbx_window_point *d_points, const uint8_t *d_point_desc, const int32_t *d_np, int max_np, int chi2_test,
                                     void *d_out_idx, void *d_out2)
{
    if (!m || !ex) return setError(ORBX_ERR_INVALID_ARG, "%s: null handle", fn);
    if (ex->prm.device != m->device) return setError(ORBX_ERR_INVALID_ARG, "%s: matcher and extractor must live on the same device", fn);
    if (nframes < 0 || nframes > ex->lastFrames) return setError(ORBX_ERR_INVALID_ARG, "%s: %d keyframes, the extractor's last batch holds %d", fn, nframes, ex->lastFrames);
    if (max_np < 0 || (max_np > 0 && (!d_points || !d_point_desc)) || !d_np || !d_out_idx || !d_out2)
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    int rc = checkCamera(k4, dist, ndist, knew4 ? knew4 : k4, fn);
    if (rc != ORBX_OK) return rc;
    if ((rc = checkGridBounds(min_x, max_x, min_y, max_y, fn)) != ORBX_OK) return rc;
    const int cap = ex->maxKeypoints, nlev = ex->prm.nlevels;
    if (nframes == 0 || max_np == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t F = (size_t)nframes;
    if ((rc = m->d_bKps.ensure(F * cap)) != ORBX_OK) return rc;
    if ((rc = m->d_grid.ensure(F * ((size_t)ORBX_GRID_CELLS + 1 + 2 * (size_t)cap))) != ORBX_OK) return rc;
    if (fuse && (rc = m->d_work.ensure(F * (size_t)max_np + 64)) != ORBX_OK) return rc;
    int *cs = m->d_grid.p, *it = cs + F * (ORBX_GRID_CELLS + 1), *co = it + F * cap;
    orbx_keypoint *kun = m->d_bKps.p;
    if ((rc = orderAfterExtractors(m, ex, ex)) != ORBX_OK) return rc;
    // mvInvLevelSigma2 behind the scale table (d_scale holds 2 * ORBX_MAX_LEVELS floats)
    HIP_TRY(hipMemcpyAsync(m->d_scale.p + ORBX_MAX_LEVELS, ex->invSigma2.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, m->stream));
    if (ndist == 0 || dist[0] == 0.0)
        HIP_TRY(hipMemcpyAsync(kun, ex->d_kps.p, F * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToDevice, m->stream));
    else
        HIP_TRY(orbxLaunchUndistort(m->stream, ex->d_kps.p, kun, cap, k4, dist, ndist, knew4 ? knew4 : k4, nframes, cap, cap, ex->d_n.p));
    HIP_TRY(orbxLaunchGridBuild(m->stream, kun, cap, min_x, max_x, min_y, max_y, cs, it, co, nframes, cap, ex->d_n.p));
    OrbxKfwFrames b;
    b.nframes = nframes; b.maxNp = max_np; b.chi2 = chi2_test != 0; b.fuseTh = fuse ? 50 : -1;   // TH_LOW (src/ORBmatcher1.cc:38)
    b.keys = kun; b.desc = ex->d_desc.p; b.uRight = d_u_right; b.invSigma2 = m->d_scale.p + ORBX_MAX_LEVELS;
    b.cellStart = cs; b.items = it; b.keyStride = cap; b.gridStride = ORBX_GRID_CELLS + 1; b.itemStride = cap;
    b.minX = min_x; b.maxX = max_x; b.minY = min_y; b.maxY = max_y;
    b.pts = d_points; b.ptDesc = d_point_desc; b.npDev = d_np;
    b.bestIdx = (int *)d_out_idx;
    b.bestDist = fuse ? m->d_work.p : (int *)d_out2;
    b.nFused = fuse ? (int *)d_out2 : nullptr;
    HIP_TRY(orbxLaunchKeyframeWindowSearchFrames(m->stream, &b));
    return orderExtractorsAfter(m, ex, ex);
}

extern "C" int orbx_search_keyframe_windows_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                                   const float *knew4, float min_x, float max_x, float min_y, float max_y, const float *d_u_right,
                                                   const orbx_window_point *d_points, const uint8_t *d_point_desc, const int32_t *d_np,
                                                   int max_np, int chi2_test, void *d_best_idx, void *d_best_dist)
{
    ORBX_API_ENTRY;
    return keyframeWindowsFramesImpl("orbx_search_keyframe_windows_frames", false, m, ex, nframes, k4, dist, ndist, knew4, min_x, max_x, min_y, max_y,
                                     d_u_right, d_points, d_point_desc, d_np, max_np, chi2_test, d_best_idx, d_best_dist);
}

extern "C" int orbx_fuse_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist, const float *knew4,
                                float min_x, float max_x, float min_y, float max_y, const float *d_u_right, const orbx_window_point *d_points,
                                const uint8_t *d_point_desc, const int32_t *d_np, int max_np, int chi2_test, void *d_fused_idx, void *d_nfused)
{
    ORBX_API_ENTRY;
    return keyframeWindowsFramesImpl("orbx_fuse_frames", true, m, ex, nframes, k4, dist, ndist, knew4, min_x, max_x, min_y, max_y, d_u_right, d_points,
                                     d_point_desc, d_np, max_np, chi2_test, d_fused_idx, d_nfused);
}

// The CurrentFrame.Nleft != -1 branches (two KB8 fisheye cameras) of both SearchByProjection functions for a BATCH of frames, device-resident:
// frame p = (frame p of `left`'s last batch, frame p of `right`'s).  Keypoints and descriptors of a frame are gathered into one buffer
// [capL left slots | capR right slots] (the kernels address the right keypoints at a fixed offset: "NLeft" = capL, unused slots never
// enter a grid); combined indices - in d_cur_state, d_assign - are i for left keypoint i and capL + j for right keypoint j.
static int searchByProjectionFisheyeFramesImpl(const char *fn, int mode, float nnratio, orbx_matcher *m, orbx_extractor *left, orbx_extractor *right,
                                               int nframes, float min_x, float max_x, float min_y, float max_y, const int8_t *d_cur_state,
                                               const orbx_projection *d_proj_left, const orbx_projection *d_proj_right, const uint8_t *d_proj_desc,
                                               const int32_t *d_np, int max_np, const int32_t *d_left_to_right, const int32_t *d_right_to_left,
                                               float th, int forward, int backward, int check_orientation, void *d_assign, void *d_nmatches)
{
    if (!m || !left || !right) return setError(ORBX_ERR_INVALID_ARG, "%s: null handle", fn);
    if (left == right) return setError(ORBX_ERR_INVALID_ARG, "%s: left and right must be two extractors", fn);
    if (left->prm.device != m->device || right->prm.device != m->device) return setError(ORBX_ERR_INVALID_ARG, "%s: matcher and extractors must live on the same device", fn);
    if (left->prm.nlevels != right->prm.nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: the extractors differ in nlevels", fn);
    if (nframes < 0 || nframes > left->lastFrames || nframes > right->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "%s: %d frames, the extractors' last batches hold %d / %d", fn, nframes, left->lastFrames, right->lastFrames);
    if (max_np < 0 || (max_np > 0 && (!d_proj_left || !d_proj_right || !d_proj_desc)) || !d_np || !d_cur_state || !d_assign || !d_nmatches || !(th >= 0.f) ||
        (mode == 1 && (!d_left_to_right || !d_right_to_left)))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    int rc = checkGridBounds(min_x, max_x, min_y, max_y, fn);
    if (rc != ORBX_OK) return rc;
    const int capL = left->maxKeypoints, capR = right->maxKeypoints, capC = capL + capR, nlev = left->prm.nlevels;
    if (capC >= 65536) return setError(ORBX_ERR_CAPACITY, "%s: at most 65535 keypoints per frame (left + right capacity)", fn);
    if (nframes == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t F = (size_t)nframes, NP = (size_t)std::max(max_np, 1);
    // scratch: combined keypoints / descriptors, two grids per frame, steps, candidate lists, resolve state
    if ((rc = m->d_bKps.ensure(F * capC)) != ORBX_OK) return rc;
    if ((rc = m->d_siWork.ensure(F * capC * 32 + 64)) != ORBX_OK) return rc;
    const size_t gridInts = (size_t)ORBX_GRID_CELLS + 1;
    if ((rc = m->d_grid.ensure(2 * F * gridInts + 4 * F * capC)) != ORBX_OK) return rc;
    // work: count[2 NP], choice[2 NP], next[4 NP] per frame; head[capC], partner[capC] per frame; result[2] per frame
    if ((rc = m->d_work.ensure(F * (8 * NP + 2 * (size_t)capC + 2) + 64)) != ORBX_OK) return rc;
    if ((rc = m->d_sbpCand.ensure(F * 2 * NP * ORBX_SBP_CAP)) != ORBX_OK) return rc;
    if ((rc = m->d_proj.ensure(F * 2 * NP)) != ORBX_OK) return rc;
    int *csL = m->d_grid.p, *csR = csL + F * gridInts, *itL = csR + F * gridInts, *itR = itL + F * capC, *coL = itR + F * capC, *coR = coL + F * capC;
    int *count = m->d_work.p, *choice = count + F * 2 * NP, *next = choice + F * 2 * NP, *head = next + F * 4 * NP, *dPartner = head + F * capC,
        *result = dPartner + F * capC;
    hipStream_t st = m->stream;
    if ((rc = orderAfterExtractors(m, left, right)) != ORBX_OK) return rc;
    // gather: rows = frames, the left block then the right block of every frame
    orbx_keypoint *kc = m->d_bKps.p;
    uint8_t *dc = m->d_siWork.p;
    HIP_TRY(hipMemcpy2DAsync(kc, (size_t)capC * sizeof(orbx_keypoint), left->d_kps.p, (size_t)capL * sizeof(orbx_keypoint), (size_t)capL * sizeof(orbx_keypoint), F, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(kc + capL, (size_t)capC * sizeof(orbx_keypoint), right->d_kps.p, (size_t)capR * sizeof(orbx_keypoint), (size_t)capR * sizeof(orbx_keypoint), F, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(dc, (size_t)capC * 32, left->d_desc.p, (size_t)capL * 32, (size_t)capL * 32, F, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync(dc + (size_t)capL * 32, (size_t)capC * 32, right->d_desc.p, (size_t)capR * 32, (size_t)capR * 32, F, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(m->d_scale.p, left->scale.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(orbxLaunchGridBuild(st, kc, capL, min_x, max_x, min_y, max_y, csL, itL, coL, nframes, capC, left->d_n.p));
    HIP_TRY(orbxLaunchGridBuild(st, kc + capL, capR, min_x, max_x, min_y, max_y, csR, itR, coR, nframes, capC, right->d_n.p));
    HIP_TRY(orbxLaunchFisheyeSteps(st, d_proj_left, d_proj_right, d_np, max_np, nframes, mode, m->d_proj.p));
    if (mode == 1) HIP_TRY(orbxLaunchFisheyePartner(st, d_left_to_right, d_right_to_left, left->d_n.p, right->d_n.p, capL, capR, nframes, dPartner));
    OrbxSbpBatch b;
    b.nframes = nframes; b.kcStride = capC; b.projStride = 2 * (long long)max_np; b.ncDev = right->d_n.p; b.npDev = d_np;
    OrbxSbpFisheye f;
    f.kc = kc; f.dc = dc; f.curState = (const signed char *)d_cur_state; f.nLeft = capL; f.nRight = capR;
    f.cellStartL = csL; f.itemsL = itL; f.cellStartR = csR; f.itemsR = itR;
    f.minX = min_x; f.maxX = max_x; f.minY = min_y; f.maxY = max_y; f.scale = m->d_scale.p;
    f.steps = m->d_proj.p; f.dproj = d_proj_desc; f.np = max_np;
    f.th = th; f.forward = forward != 0; f.backward = backward != 0; f.checkOri = check_orientation != 0; f.mode = mode; f.nnratio = nnratio;
    f.partner = mode == 1 ? dPartner : nullptr;
    f.cand = m->d_sbpCand.p; f.count = count; f.choice = choice; f.head = head; f.next = next; f.assign = (int *)d_assign; f.result = result;
    f.batch = &b;
    HIP_TRY(orbxLaunchSearchByProjectionFisheye(st, f));
    HIP_TRY(orbxLaunchGatherResults(st, result, nframes, (int *)d_nmatches));
    return orderExtractorsAfter(m, left, right);
}

extern "C" int orbx_search_by_projection_fisheye_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int nframes, float min_x,
                                                        float max_x, float min_y, float max_y, const int8_t *d_cur_state,
                                                        const orbx_projection *d_proj_left, const orbx_projection *d_proj_right,
                                                        const uint8_t *d_proj_desc, const int32_t *d_np, int max_np, float th, int forward,
                                                        int backward, int check_orientation, void *d_assign, void *d_nmatches)
{
    ORBX_API_ENTRY;
    return searchByProjectionFisheyeFramesImpl("orbx_search_by_projection_fisheye_frames", 0, 0.f, m, left, right, nframes, min_x, max_x, min_y, max_y,
                                               d_cur_state, d_proj_left, d_proj_right, d_proj_desc, d_np, max_np, nullptr, nullptr, th, forward, backward,
                                               check_orientation, d_assign, d_nmatches);
}

extern "C" int orbx_search_by_projection_points_fisheye_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int nframes, float min_x,
                                                               float max_x, float min_y, float max_y, const int8_t *d_cur_state,
                                                               const orbx_projection *d_map_points_left, const orbx_projection *d_map_points_right,
                                                               const uint8_t *d_map_point_desc, const int32_t *d_np, int max_np,
                                                               const int32_t *d_left_to_right, const int32_t *d_right_to_left, float th,
                                                               float nnratio, void *d_assign, void *d_nmatches)
{
    ORBX_API_ENTRY;
    return searchByProjectionFisheyeFramesImpl("orbx_search_by_projection_points_fisheye_frames", 1, nnratio, m, left, right, nframes, min_x, max_x, min_y,
                                               max_y, d_cur_state, d_map_points_left, d_map_points_right, d_map_point_desc, d_np, max_np, d_left_to_right,
                                               d_right_to_left, th, 0, 0, 0, d_assign, d_nmatches);
}

extern "C" int orbx_search_by_projection_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                                const float *knew4, float min_x, float max_x, float min_y, float max_y, float mbf,
                                                const float *d_u_right, const int8_t *d_cur_state, const orbx_projection *d_proj,
                                                const uint8_t *d_proj_desc, const int32_t *d_np, int max_np, float th, int forward, int backward,
                                                int check_orientation, void *d_kps_un, void *d_assign, void *d_nmatches)
{
    ORBX_API_ENTRY;
    return searchByProjectionFramesImpl("orbx_search_by_projection_frames", 0, 0.f, m, ex, nframes, k4, dist, ndist, knew4, min_x, max_x, min_y, max_y, mbf,
                                        d_u_right, d_cur_state, d_proj, d_proj_desc, d_np, max_np, th, forward, backward, check_orientation, d_kps_un, d_assign,
                                        d_nmatches);
}

// ORBmatcher::SearchForInitialization for a batch of frame pairs, device-resident - reference src/ORBmatcher1.cc:650-765 on
// Frame::UndistortKeyPoints' output (src/Frame.cc:777-811) of both frames.
extern "C" int orbx_search_for_initialization_frames(orbx_matcher *m, orbx_extractor *ex1, orbx_extractor *ex2, int frame2_offset, int npairs,
                                                     const float *k4, const float *dist, int ndist, const float *knew4, float min_x, float max_x,
                                                     float min_y, float max_y, float *d_prev_matched, int init_prev_matched, int window_size,
                                                     float nnratio, int check_orientation, void *d_matches12, void *d_nmatches)
{
    ORBX_API_ENTRY;
    const char *fn = "orbx_search_for_initialization_frames";
    if (!m || !ex1 || !ex2) return setError(ORBX_ERR_INVALID_ARG, "%s: null handle", fn);
    if (ex1->prm.device != m->device || ex2->prm.device != m->device)
        return setError(ORBX_ERR_INVALID_ARG, "%s: matcher and extractors must live on the same device", fn);
    if (npairs < 0 || frame2_offset < 0 || npairs > ex1->lastFrames || npairs + frame2_offset > ex2->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "%s: %d pairs (second frame + %d), the extractors' last batches hold %d / %d frames", fn, npairs,
                        frame2_offset, ex1->lastFrames, ex2->lastFrames);
    if (!d_prev_matched || !d_matches12 || !d_nmatches || window_size < 0)
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    int rc = checkCamera(k4, dist, ndist, knew4 ? knew4 : k4, fn);
    if (rc != ORBX_OK) return rc;
    if ((rc = checkGridBounds(min_x, max_x, min_y, max_y, fn)) != ORBX_OK) return rc;
    const int cap1 = ex1->maxKeypoints, cap2 = ex2->maxKeypoints;
    if (cap1 >= 65536 || cap2 >= 65536) return setError(ORBX_ERR_CAPACITY, "%s: at most 65535 keypoints per frame", fn);
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t P = (size_t)npairs;
    const orbx_keypoint *un1 = ex1->d_kps.p, *un2 = ex2->d_kps.p + (size_t)frame2_offset * cap2;
    if ((rc = m->d_bin.ensure(P * cap1)) != ORBX_OK) return rc;
    if ((rc = m->d_siWork.ensure(orbxSearchInitWorkBytes(npairs, cap1))) != ORBX_OK) return rc;
    if ((rc = orderAfterExtractors(m, ex1, ex2)) != ORBX_OK) return rc;
    if (!(ndist == 0 || dist[0] == 0.0)) {  // mvKeysUn of both frames (without distortion mvKeysUn = mvKeys, src/Frame.cc:779-783)
        if ((rc = m->d_bKps.ensure(P * ((size_t)cap1 + cap2))) != ORBX_OK) return rc;
        orbx_keypoint *u1 = m->d_bKps.p, *u2 = u1 + P * cap1;
        HIP_TRY(orbxLaunchUndistort(m->stream, un1, u1, cap1, k4, dist, ndist, knew4 ? knew4 : k4, npairs, cap1, cap1, ex1->d_n.p));
        HIP_TRY(orbxLaunchUndistort(m->stream, un2, u2, cap2, k4, dist, ndist, knew4 ? knew4 : k4, npairs, cap2, cap2, ex2->d_n.p + frame2_offset));
        un1 = u1; un2 = u2;
    }
    const hipError_t e = orbxLaunchSearchInitBatch(m->stream, npairs, un1, ex1->d_desc.p, cap1, ex1->d_n.p, un2,
                                                   ex2->d_desc.p + (size_t)frame2_offset * cap2 * 32, cap2, ex2->d_n.p + frame2_offset, cap1, cap2, min_x,
                                                   max_x, min_y, max_y, d_prev_matched, init_prev_matched != 0, (int *)d_matches12, m->d_bin.p,
                                                   (int *)d_nmatches, window_size, nnratio, check_orientation != 0, m->d_siWork.p);
    if (e == hipErrorInvalidValue) return setError(ORBX_ERR_CAPACITY, "%s: %d keypoints per frame exceed what the search keeps in LDS", fn, cap1);
    HIP_TRY(e);
    return orderExtractorsAfter(m, ex1, ex2);
}

// Tracking::SearchLocalPoints' matcher (ORBmatcher::SearchByProjection(Frame &, const vector<MapPoint*> &, th, ..), reference
// src/ORBmatcher1.cc:45-215, Nleft == -1) over a batch of frames: the same pipeline with the map-point records of orbx_search_by_projection_points
extern "C" int orbx_search_by_projection_points_frames(orbx_matcher *m, orbx_extractor *ex, int nframes, const float *k4, const float *dist, int ndist,
                                                       const float *knew4, float min_x, float max_x, float min_y, float max_y,
                                                       const float *d_u_right, const int8_t *d_cur_state, const orbx_projection *d_map_points,
                                                       const uint8_t *d_map_point_desc, const int32_t *d_np, int max_np, float th, float nnratio,
                                                       void *d_kps_un, void *d_assign, void *d_nmatches)
{
    ORBX_API_ENTRY;
    return searchByProjectionFramesImpl("orbx_search_by_projection_points_frames", 1, nnratio, m, ex, nframes, k4, dist, ndist, knew4, min_x, max_x, min_y, max_y,
                                        0.f, d_u_right, d_cur_state, d_map_points, d_map_point_desc, d_np, max_np, th, 0, 0, 0, d_kps_un, d_assign, d_nmatches);
}

// Frame::ComputeStereoFishEyeMatches — reference src/Frame.cc:1156-1196: knnMatch on the lapping-area rows, Lowe's ratio, TriangulateMatches
static int fillFisheyeArgs(OrbxFisheyeArgs &a, const float *cam1, const float *cam2, float precision, const float *R12, const float *t12,
                           const float *level_sigma2, int nlevels, const char *fn)
{
    if (!cam1 || !cam2 || !R12 || !t12 || !level_sigma2 || nlevels <= 0 || nlevels > ORBX_MAX_LEVELS || !(precision > 0))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad camera / level arguments", fn);
    memset(&a, 0, sizeof a);
    for (int i = 0; i < 8; ++i) { a.cam1[i] = cam1[i]; a.cam2[i] = cam2[i]; }
    for (int i = 0; i < 9; ++i) a.R12[i] = R12[i];
    for (int i = 0; i < 3; ++i) a.t12[i] = t12[i];
    for (int i = 0; i < nlevels; ++i) a.sigma2[i] = level_sigma2[i];
    a.precision = precision;
    if (cam1[0] == 0.f || cam1[1] == 0.f || cam2[0] == 0.f || cam2[1] == 0.f) return setError(ORBX_ERR_INVALID_ARG, "%s: zero focal length", fn);
    return ORBX_OK;
}

extern "C" int orbx_compute_stereo_fisheye_matches(orbx_matcher *m, const orbx_keypoint *kl, const uint8_t *dl, int nl, int mono_left,
                                                   const orbx_keypoint *kr, const uint8_t *dr, int nr, int mono_right, const float *cam1,
                                                   const float *cam2, float precision, const float *R12, const float *t12,
                                                   const float *level_sigma2, int nlevels, int32_t *left_to_right, int32_t *right_to_left,
                                                   float *depth, float *p3d)
{
    ORBX_API_ENTRY;
    const char *fn = "orbx_compute_stereo_fisheye_matches";
    if (!m || nl < 0 || nr < 0 || mono_left < 0 || mono_right < 0 || mono_left > nl || mono_right > nr || (nl > 0 && (!kl || !dl || !left_to_right || !depth || !p3d)) ||
        (nr > 0 && (!kr || !dr || !right_to_left)))
        return setError(ORBX_ERR_INVALID_ARG, "%s: bad argument", fn);
    OrbxFisheyeArgs a;
    int rc = fillFisheyeArgs(a, cam1, cam2, precision, R12, t12, level_sigma2, nlevels, fn);
    if (rc != ORBX_OK) return rc;
    for (int i = 0; i < nl; ++i)
        if (kl[i].octave < 0 || kl[i].octave >= nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: left keypoint %d has octave %d", fn, i, kl[i].octave);
    for (int i = 0; i < nr; ++i)
        if (kr[i].octave < 0 || kr[i].octave >= nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: right keypoint %d has octave %d", fn, i, kr[i].octave);
    for (int i = 0; i < nl; ++i) { left_to_right[i] = -1; depth[i] = -1.0f; }
    for (int i = 0; i < nr; ++i) right_to_left[i] = -1;
    if (nl - mono_left <= 0) return 0;
    HIP_TRY(hipSetDevice(m->device));
    HostPack in(m);
    if ((rc = uploadQT(in, dl, nl, dr, nr)) != ORBX_OK) return rc;
    if ((rc = m->d_uRight.ensure(m->maxQ)) != ORBX_OK || (rc = m->d_depth.ensure((size_t)m->maxQ * 3)) != ORBX_OK || (rc = m->d_sad.ensure(m->maxQ)) != ORBX_OK ||
        (rc = m->d_bIdx.ensure((size_t)std::max(m->maxT, 16) + 16)) != ORBX_OK)
        return rc;
    in.add(kl, (size_t)nl * sizeof(orbx_keypoint), m->d_kl.p);
    if (nr > 0) in.add(kr, (size_t)nr * sizeof(orbx_keypoint), m->d_kr.p);
    const int nq = nl - mono_left, nt = nr - mono_right;
    if ((rc = in.send()) != ORBX_OK) return rc;
    if ((rc = knn2Launch(m, 1, m->d_q.p + (size_t)mono_left * 32, 0, nullptr, nullptr, nq, m->d_t.p + (size_t)mono_right * 32, 0, nullptr, nullptr, nt,
                         m->d_idx.p, m->d_dist.p)) != ORBX_OK)
        return rc;
    a.kl = m->d_kl.p; a.kr = m->d_kr.p; a.nl = nl; a.nr = nr; a.monoL = mono_left; a.monoR = mono_right;
    a.idx = m->d_idx.p; a.dist = m->d_dist.p; a.knnStride = nq;
    a.leftToRight = m->d_sad.p; a.rightToLeft = m->d_bIdx.p; a.depth = m->d_uRight.p; a.p3d = m->d_depth.p;
    a.outStrideL = nl; a.outStrideR = nr;
    a.nMatches = m->d_bIdx.p + std::max(m->maxT, 16);
    HIP_TRY(orbxLaunchFisheyeTriangulate(m->stream, &a, 1));
    int nMatches = 0;
    HostFetch out(m);
    out.add(left_to_right, a.leftToRight, (size_t)nl * sizeof(int));
    if (nr > 0) out.add(right_to_left, a.rightToLeft, (size_t)nr * sizeof(int));
    out.add(depth, a.depth, (size_t)nl * sizeof(float));
    out.add(p3d, a.p3d, (size_t)nl * 3 * sizeof(float));
    out.add(&nMatches, a.nMatches, sizeof(int));
    if ((rc = out.finish()) != ORBX_OK) return rc;
    return nMatches;
}

extern "C" int orbx_compute_stereo_fisheye_matches_frames(orbx_matcher *m, orbx_extractor *left, orbx_extractor *right, int npairs, const float *cam1,
                                                          const float *cam2, float precision, const float *R12, const float *t12,
                                                          void *d_left_to_right, void *d_right_to_left, void *d_depth, void *d_p3d, void *d_nmatches)
{
    ORBX_API_ENTRY;
    const char *fn = "orbx_compute_stereo_fisheye_matches_frames";
    int rc = checkFramePairs(fn, m, left, right, npairs);
    if (rc != ORBX_OK) return rc;
    if (!d_left_to_right || !d_right_to_left || !d_depth || !d_p3d || !d_nmatches) return setError(ORBX_ERR_INVALID_ARG, "%s: null output", fn);
    if (left->prm.nlevels != right->prm.nlevels) return setError(ORBX_ERR_INVALID_ARG, "%s: left / right extractors differ in levels", fn);
    OrbxFisheyeArgs a;
    if ((rc = fillFisheyeArgs(a, cam1, cam2, precision, R12, t12, left->sigma2.data(), left->prm.nlevels, fn)) != ORBX_OK) return rc;
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const int capL = left->maxKeypoints, capR = right->maxKeypoints;
    if ((rc = m->d_bIdx.ensure((size_t)npairs * capL * 2)) != ORBX_OK || (rc = m->d_bDist.ensure((size_t)npairs * capL * 2)) != ORBX_OK) return rc;
    if ((rc = orderAfterExtractors(m, left, right)) != ORBX_OK) return rc;
    // knnMatch(mDescriptors.rowRange(monoLeft, rows), mDescriptorsRight.rowRange(monoRight, rows), matches, 2) for every pair
    if ((rc = knn2Launch(m, npairs, left->d_desc.p, (long long)capL * 32, left->d_n.p, left->d_mono.p, capL, right->d_desc.p, (long long)capR * 32,
                         right->d_n.p, right->d_mono.p, capR, m->d_bIdx.p, m->d_bDist.p)) != ORBX_OK)
        return rc;
    a.kl = left->d_kps.p; a.kr = right->d_kps.p; a.kpStrideL = capL; a.kpStrideR = capR;
    a.nlDev = left->d_n.p; a.nrDev = right->d_n.p; a.monoLDev = left->d_mono.p; a.monoRDev = right->d_mono.p;
    a.nl = capL; a.nr = capR;
    a.idx = m->d_bIdx.p; a.dist = m->d_bDist.p; a.knnStride = capL;
    a.leftToRight = (int *)d_left_to_right; a.rightToLeft = (int *)d_right_to_left; a.depth = (float *)d_depth; a.p3d = (float *)d_p3d;
    a.outStrideL = capL; a.outStrideR = capR;
    a.nMatches = (int *)d_nmatches;
    HIP_TRY(orbxLaunchFisheyeTriangulate(m->stream, &a, npairs));
    return orderExtractorsAfter(m, left, right);
}

extern "C" int orbx_matcher_download(orbx_matcher *m, const void *d_src, void *h_dst, size_t bytes)
{
    ORBX_API_ENTRY;
    if (!m || (bytes > 0 && (!d_src || !h_dst))) return setError(ORBX_ERR_INVALID_ARG, "orbx_matcher_download: bad argument");
    HIP_TRY(hipSetDevice(m->device));
    if (bytes > 0) HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

// =====================================================================================================
// DBoW2 vocabulary: ORBVocabulary = TemplatedVocabulary<FORB::TDescriptor, FORB> resident in HBM.
// =====================================================================================================
#define ORBX_BOW_MAX_FEATURES 16384  // (word, feature) keys of one transform are sorted in one workgroup's LDS (128 KB)

struct orbx_vocabulary {
    int device = 0, k = 0, L = 0, scoring = 0, weighting = 0, nNodes = 0, nWords = 0;
    hipStream_t stream = nullptr;
    DevBuf<int> d_childStart, d_childNode, d_wordOfNode;
    DevBuf<uint8_t> d_childDesc, d_feat;
    DevBuf<double> d_weight, d_featWeight, d_bowValue;
    DevBuf<int> d_work;  // featWord, featNode, bowWord, fvNode, fvStart, fvFeature, counts
    DevBuf<int> d_batchInt;       // the same per frame of a batch (orbx_vocabulary_transform_batch)
    DevBuf<double> d_batchDbl;
    // the last batch transform's FeatureVectors stay on the device for the keyframe-side batch searches (orbx_search_for_triangulation_frames)
    const orbx_extractor *batchEx = nullptr; int batchFrames = 0, batchCap = 0;
    unsigned long long batchSerial = 0;   // orbx_extractor::batchSerial of the batch the cached FeatureVectors were computed from
    const int *batchFvNode = nullptr, *batchFvStart = nullptr, *batchFvFeature = nullptr;
    std::vector<int> batchNFv;
    HostXfer xf;                  // packed transfers of the single-frame transform
};

extern "C" void orbx_vocabulary_destroy(orbx_vocabulary *v)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!v) return;
    (void)hipSetDevice(v->device);
    if (v->stream) { (void)hipStreamSynchronize(v->stream); (void)hipStreamDestroy(v->stream); }
    v->d_childStart.release(); v->d_childNode.release(); v->d_wordOfNode.release(); v->d_childDesc.release(); v->d_feat.release();
    v->d_weight.release(); v->d_featWeight.release(); v->d_bowValue.release(); v->d_work.release();
    v->d_batchInt.release(); v->d_batchDbl.release();
    v->xf.release();
    delete v;
}

// Nodes as TemplatedVocabulary::loadFromTextFile builds them (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423): node 0 is
// the root, node i > 0 hangs under parent[i] < i, children keep node-id order, leaves get word ids in node order.
extern "C" int orbx_vocabulary_create(int device, int k, int L, int scoring, int weighting, int n_nodes, const int32_t *parent,
                                      const uint8_t *is_leaf, const uint8_t *desc, const double *weight, orbx_vocabulary **out)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!out) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_create: null output");
    *out = nullptr;
    if (k < 0 || k > 20 || L < 1 || L > 10 || scoring < 0 || scoring > 5 || weighting < 0 || weighting > 3)  // the loader's own check (:1359)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_create: k / L / scoring / weighting out of range");
    if (n_nodes < 1 || !parent || !is_leaf || !desc || !weight) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_create: bad argument");
    for (int i = 1; i < n_nodes; ++i)
        if (parent[i] < 0 || parent[i] >= i) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_create: node %d has parent %d (must be an earlier node)", i, parent[i]);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return setError(ORBX_ERR_NO_DEVICE, "orbx_vocabulary_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_create: device out of range");
    HIP_TRY(hipSetDevice(device));
    // child lists (counting sort by parent keeps node-id order) and the descriptors in child-list order
    std::vector<int> childStart((size_t)n_nodes + 1, 0), childNode((size_t)std::max(n_nodes - 1, 1)), wordOfNode((size_t)n_nodes, 0);
    for (int i = 1; i < n_nodes; ++i) childStart[(size_t)parent[i] + 1]++;
    for (int i = 0; i < n_nodes; ++i) childStart[(size_t)i + 1] += childStart[i];
    std::vector<int> fill(childStart.begin(), childStart.end() - 1);
    std::vector<uint8_t> childDesc((size_t)std::max(n_nodes - 1, 1) * 32);
    int nWords = 0;
    for (int i = 1; i < n_nodes; ++i) {
        const int p = fill[parent[i]]++;
        childNode[p] = i;
        memcpy(&childDesc[(size_t)p * 32], desc + (size_t)i * 32, 32);
        if (is_leaf[i] > 0) wordOfNode[i] = nWords++;
    }
    orbx_vocabulary *v = new (std::nothrow) orbx_vocabulary();
    if (!v) return setError(ORBX_ERR_OUT_OF_MEMORY, "orbx_vocabulary_create: host allocation failed");
    v->device = device; v->k = k; v->L = L; v->scoring = scoring; v->weighting = weighting; v->nNodes = n_nodes; v->nWords = nWords;
    hipError_t e = hipStreamCreateWithFlags(&v->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete v; return setError(ORBX_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    const size_t F = ORBX_BOW_MAX_FEATURES;
    int rc;
    if ((rc = v->d_childStart.ensure(childStart.size())) != ORBX_OK || (rc = v->d_childNode.ensure(childNode.size())) != ORBX_OK ||
        (rc = v->d_wordOfNode.ensure(wordOfNode.size())) != ORBX_OK || (rc = v->d_childDesc.ensure(childDesc.size())) != ORBX_OK ||
        (rc = v->d_weight.ensure((size_t)n_nodes)) != ORBX_OK || (rc = v->d_feat.ensure(F * 32)) != ORBX_OK ||
        (rc = v->d_featWeight.ensure(F)) != ORBX_OK || (rc = v->d_bowValue.ensure(F)) != ORBX_OK || (rc = v->d_work.ensure(6 * F + 8)) != ORBX_OK) {
        orbx_vocabulary_destroy(v);
        return rc;
    }
    e = hipMemcpy(v->d_childStart.p, childStart.data(), childStart.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_childNode.p, childNode.data(), childNode.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_wordOfNode.p, wordOfNode.data(), wordOfNode.size() * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_childDesc.p, childDesc.data(), childDesc.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(v->d_weight.p, weight, (size_t)n_nodes * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { orbx_vocabulary_destroy(v); return setError(ORBX_ERR_HIP, "orbx_vocabulary_create: upload: %s", hipGetErrorString(e)); }
    *out = v;
    return ORBX_OK;
}

// bool TemplatedVocabulary::loadFromTextFile(const std::string &filename) — Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1423
// (ORBvoc.txt; called by System::System, src/System.cc).  First line "k L scoring weighting", then one node per line:
// "parent is_leaf d0 .. d31 weight".  (An empty last line, which the reference's eof loop turns into a root child with an
// unspecified descriptor, is skipped.)
extern "C" int orbx_vocabulary_load_text(int device, const char *path, orbx_vocabulary **out)
{
    ORBX_API_ENTRY;
    ORBX_SETUP_LOCK;
    if (!out) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_load_text: null output");
    *out = nullptr;
    if (!path) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_load_text: null path");
    FILE *f = fopen(path, "r");
    if (!f) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_load_text: cannot open %s", path);
    int k = 0, L = 0, n1 = 0, n2 = 0;
    if (fscanf(f, "%d %d %d %d", &k, &L, &n1, &n2) != 4 || k < 0 || k > 20 || L < 1 || L > 10 || n1 < 0 || n1 > 5 || n2 < 0 || n2 > 3) {
        fclose(f);
        return setError(ORBX_ERR_INVALID_ARG, "Vocabulary loading failure: This is not a correct text file!");
    }
    std::vector<int32_t> parent(1, 0);
    std::vector<uint8_t> leaf(1, 0), desc(32, 0);
    std::vector<double> weight(1, 0.0);
    for (;;) {
        int pid, isLeaf;
        if (fscanf(f, "%d %d", &pid, &isLeaf) != 2) break;
        uint8_t d[32];
        bool ok = true;
        for (int i = 0; i < 32 && ok; ++i) { int b; ok = fscanf(f, "%d", &b) == 1; d[i] = (uint8_t)b; }
        double w = 0;
        ok = ok && fscanf(f, "%lf", &w) == 1;
        if (!ok || pid < 0 || pid >= (int)parent.size()) { fclose(f); return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_load_text: malformed node %zu", parent.size()); }
        parent.push_back(pid); leaf.push_back(isLeaf > 0); desc.insert(desc.end(), d, d + 32); weight.push_back(w);
    }
    fclose(f);
    return orbx_vocabulary_create(device, k, L, n1, n2, (int)parent.size(), parent.data(), leaf.data(), desc.data(), weight.data(), out);
}

extern "C" int orbx_vocabulary_size(const orbx_vocabulary *v) { return v ? v->nWords : 0; }  // size(): number of words

// void TemplatedVocabulary::transform(const std::vector<TDescriptor> &features, BowVector &v, FeatureVector &fv, int levelsup)
// — TemplatedVocabulary.h:1127-1196, as Frame::ComputeBoW / KeyFrame::ComputeBoW call it with levelsup = 4.
// desc: host descriptors, or (deviceDesc) descriptors already in HBM on the vocabulary's device
static int vocabularyTransformImpl(orbx_vocabulary *v, const uint8_t *desc, bool deviceDesc, int n, int levelsup, int32_t *feat_word,
                                   int32_t *feat_node, int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start,
                                   int32_t *fv_feature, int32_t *n_fv)
{
    if (!v || n < 0 || (n > 0 && !desc) || !n_bow || !n_fv || !fv_start || (n > 0 && (!bow_word || !bow_value || !fv_node || !fv_feature)))
        return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform: bad argument");
    if (n > ORBX_BOW_MAX_FEATURES) return setError(ORBX_ERR_CAPACITY, "%d features exceed the transform capacity %d", n, ORBX_BOW_MAX_FEATURES);
    *n_bow = 0; *n_fv = 0; fv_start[0] = 0;
    if (n == 0 || v->nNodes <= 1) return 0;  // empty()
    HIP_TRY(hipSetDevice(v->device));
    const size_t F = ORBX_BOW_MAX_FEATURES;
    int *featWord = v->d_work.p, *featNode = featWord + F, *bowWord = featNode + F, *fvNode = bowWord + F, *fvStart = fvNode + F,
        *fvFeature = fvStart + F + 1, *counts = fvFeature + F;
    v->xf.stream = v->stream;
    if (!deviceDesc) {
        HostPack in(&v->xf);
        in.add(desc, (size_t)n * 32, v->d_feat.p);
        const int rcs = in.send();
        if (rcs != ORBX_OK) return rcs;
    }
    HIP_TRY(orbxLaunchBowTransform(v->stream, v->d_childStart.p, v->d_childNode.p, v->d_childDesc.p, v->d_wordOfNode.p, v->d_weight.p, v->L,
                                   v->scoring, v->weighting, deviceDesc ? (const void *)desc : (const void *)v->d_feat.p, n, levelsup, featWord,
                                   featNode, v->d_featWeight.p, bowWord, v->d_bowValue.p, fvNode, fvStart, fvFeature, counts));
    // One transfer back: the counts and every array at its largest possible size (n entries; 36 n bytes) - the counts are only known on
    // the host, and asking for them first meant three round trips per frame.
    int cnt[2] = {0, 0};
    std::vector<int32_t> tBowWord((size_t)n), tFvNode((size_t)n), tFvStart((size_t)n + 1), tFvFeature((size_t)n);
    std::vector<double> tBowValue((size_t)n);
    HostFetch out(&v->xf);
    out.add(cnt, counts, sizeof cnt);
    if (feat_word) out.add(feat_word, featWord, (size_t)n * sizeof(int));
    if (feat_node) out.add(feat_node, featNode, (size_t)n * sizeof(int));
    out.add(tBowWord.data(), bowWord, (size_t)n * sizeof(int));
    out.add(tBowValue.data(), v->d_bowValue.p, (size_t)n * sizeof(double));
    out.add(tFvNode.data(), fvNode, (size_t)n * sizeof(int));
    out.add(tFvStart.data(), fvStart, ((size_t)n + 1) * sizeof(int));
    out.add(tFvFeature.data(), fvFeature, (size_t)n * sizeof(int));
    {
        const int rcf = out.finish();
        if (rcf != ORBX_OK) return rcf;
    }
    if (cnt[0] < 0 || cnt[0] > n || cnt[1] < 0 || cnt[1] > n) return setError(ORBX_ERR_HIP, "orbx_vocabulary_transform: internal: counts %d / %d for %d features", cnt[0], cnt[1], n);
    if (cnt[0] > 0) { memcpy(bow_word, tBowWord.data(), (size_t)cnt[0] * sizeof(int)); memcpy(bow_value, tBowValue.data(), (size_t)cnt[0] * sizeof(double)); }
    if (cnt[1] > 0) memcpy(fv_node, tFvNode.data(), (size_t)cnt[1] * sizeof(int));
    memcpy(fv_start, tFvStart.data(), ((size_t)cnt[1] + 1) * sizeof(int));
    if (fv_start[cnt[1]] > 0) memcpy(fv_feature, tFvFeature.data(), (size_t)std::min(fv_start[cnt[1]], n) * sizeof(int));
    *n_bow = cnt[0]; *n_fv = cnt[1];
    return cnt[0];
}

extern "C" int orbx_vocabulary_transform(orbx_vocabulary *v, const uint8_t *desc, int n, int levelsup, int32_t *feat_word, int32_t *feat_node,
                                         int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start,
                                         int32_t *fv_feature, int32_t *n_fv)
{
    ORBX_API_ENTRY;
    return vocabularyTransformImpl(v, desc, false, n, levelsup, feat_word, feat_node, bow_word, bow_value, n_bow, fv_node, fv_start, fv_feature, n_fv);
}

// Frame::ComputeBoW straight from the extractor's results in HBM: the descriptors of frame `frame` of the last batch never visit
// the host (the extractor's work is awaited first; both handles must live on the same device).
extern "C" int orbx_vocabulary_transform_frame(orbx_vocabulary *v, orbx_extractor *ex, int frame, int levelsup, int32_t *feat_word,
                                               int32_t *feat_node, int32_t *bow_word, double *bow_value, int32_t *n_bow, int32_t *fv_node,
                                               int32_t *fv_start, int32_t *fv_feature, int32_t *n_fv)
{
    ORBX_API_ENTRY;
    if (!v || !ex) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_frame: null handle");
    if (ex->prm.device != v->device) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_frame: extractor and vocabulary on different devices");
    if (frame < 0 || frame >= ex->lastFrames) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_frame: frame %d of %d", frame, ex->lastFrames);
    HIP_TRY(hipSetDevice(v->device));
    int32_t n = 0;
    HIP_TRY(hipMemcpyAsync(&n, ex->d_n.p + frame, sizeof n, hipMemcpyDeviceToHost, ex->stream));
    HIP_TRY(hipStreamSynchronize(ex->stream));  // everything of the batch is ordered before this on the front stream
    const uint8_t *d = ex->d_desc.p + (size_t)frame * ex->maxKeypoints * 32;
    return vocabularyTransformImpl(v, d, true, n, levelsup, feat_word, feat_node, bow_word, bow_value, n_bow, fv_node, fv_start, fv_feature, n_fv);
}

// Frame::ComputeBoW for every frame of the extractor's last batch, descriptors read in HBM: two launches for the whole batch.
// Per-frame outputs at stride cap = orbx_max_keypoints(ex) (fv_start: cap + 1).
extern "C" int orbx_vocabulary_transform_batch(orbx_vocabulary *v, orbx_extractor *ex, int nframes, int levelsup, int32_t *bow_word,
                                               double *bow_value, int32_t *n_bow, int32_t *fv_node, int32_t *fv_start, int32_t *fv_feature,
                                               int32_t *n_fv)
{
    ORBX_API_ENTRY;
    if (!v || !ex || !bow_word || !bow_value || !n_bow || !fv_node || !fv_start || !fv_feature || !n_fv)
        return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_batch: null argument");
    if (ex->prm.device != v->device) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_batch: extractor and vocabulary on different devices");
    if (nframes <= 0 || nframes > ex->lastFrames) return setError(ORBX_ERR_INVALID_ARG, "orbx_vocabulary_transform_batch: %d frames of %d", nframes, ex->lastFrames);
    const int cap = ex->maxKeypoints;
    if (cap > ORBX_BOW_MAX_FEATURES) return setError(ORBX_ERR_CAPACITY, "%d keypoints per frame exceed the transform capacity %d", cap, ORBX_BOW_MAX_FEATURES);
    HIP_TRY(hipSetDevice(v->device));
    const size_t F = (size_t)nframes * cap;
    int rc;
    if ((rc = v->d_batchInt.ensure(6 * F + 3 * (size_t)nframes + 16)) != ORBX_OK || (rc = v->d_batchDbl.ensure(2 * F)) != ORBX_OK) return rc;
    int *featWord = v->d_batchInt.p, *featNode = featWord + F, *bowWord = featNode + F, *fvNode = bowWord + F, *fvFeature = fvNode + F,
        *fvStart = fvFeature + F, *counts = fvStart + F + nframes;
    double *featWeight = v->d_batchDbl.p, *bowValue = featWeight + F;
    HIP_TRY(hipStreamSynchronize(ex->stream));  // the batch's results are complete
    if (v->nNodes <= 1) {  // empty()
        for (int f = 0; f < nframes; ++f) { n_bow[f] = 0; n_fv[f] = 0; fv_start[(size_t)f * (cap + 1)] = 0; }
        return ORBX_OK;
    }
    HIP_TRY(hipMemsetAsync(counts, 0, 2 * (size_t)nframes * sizeof(int), v->stream));
    HIP_TRY(orbxLaunchBowTransformBatch(v->stream, v->d_childStart.p, v->d_childNode.p, v->d_childDesc.p, v->d_wordOfNode.p, v->d_weight.p, v->L,
                                        v->scoring, v->weighting, ex->d_desc.p, ex->d_n.p, nframes, cap, levelsup, featWord, featNode, featWeight,
                                        bowWord, bowValue, fvNode, fvStart, fvFeature, counts));
    std::vector<int> cnt(2 * (size_t)nframes);
    HIP_TRY(hipMemcpyAsync(cnt.data(), counts, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(bow_word, bowWord, F * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(bow_value, bowValue, F * sizeof(double), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(fv_node, fvNode, F * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(fv_feature, fvFeature, F * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipMemcpyAsync(fv_start, fvStart, (F + nframes) * sizeof(int), hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    for (int f = 0; f < nframes; ++f) { n_bow[f] = cnt[2 * (size_t)f]; n_fv[f] = cnt[2 * (size_t)f + 1]; }
    v->batchEx = ex; v->batchFrames = nframes; v->batchCap = cap; v->batchSerial = ex->batchSerial;
    v->batchFvNode = fvNode; v->batchFvStart = fvStart; v->batchFvFeature = fvFeature;
    v->batchNFv.assign(n_fv, n_fv + nframes);
    return ORBX_OK;
}

// ORBmatcher::SearchByBoW for a batch of pairs of frames of `ex`'s last batch, everything where the extractor and the vocabulary left it on
// the device: Tracking::Relocalization's loop over the candidate keyframes (reference src/Tracking5.cc:83: SearchByBoW(pKF, mCurrentFrame, ..),
// keyframe_pair = 0) and LoopClosing's loop over the covisible keyframes of a candidate (src/LoopClosing1.cc:662: SearchByBoW(mpCurrentKF,
// vpCovKFi[j], ..), keyframe_pair = 1).  Two launches for all pairs.
extern "C" int orbx_search_by_bow_frames(orbx_matcher *m, orbx_vocabulary *voc, orbx_extractor *ex, const int32_t *kf_frames,
                                         const int32_t *f_frames, int npairs, int keyframe_pair, const uint8_t *d_valid, float nnratio,
                                         int check_orientation, void *d_matches, void *d_nmatches)
{
    ORBX_API_ENTRY;
    const char *fn = "orbx_search_by_bow_frames";
    if (!m || !voc || !ex || npairs < 0 || (npairs > 0 && (!kf_frames || !f_frames || !d_valid || !d_matches || !d_nmatches)))
        return setError(ORBX_ERR_INVALID_ARG, "%s: null argument", fn);
    if (keyframe_pair != 0 && keyframe_pair != 1) return setError(ORBX_ERR_INVALID_ARG, "%s: keyframe_pair must be 0 or 1", fn);
    if (ex->prm.device != m->device || voc->device != m->device) return setError(ORBX_ERR_INVALID_ARG, "%s: matcher, vocabulary and extractor must live on the same device", fn);
    if (voc->batchEx != ex || voc->batchFrames <= 0 || voc->batchCap != ex->maxKeypoints)
        return setError(ORBX_ERR_INVALID_ARG, "%s: call orbx_vocabulary_transform_batch(voc, ex, ..) on this batch first (its FeatureVectors are read on the device)", fn);
    const int F = voc->batchFrames, cap = ex->maxKeypoints;
    if (voc->batchSerial != ex->batchSerial || F > ex->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "%s: the extractor ran another batch since orbx_vocabulary_transform_batch - its keypoints no longer belong to the cached FeatureVectors; transform again", fn);
    if (cap > 65535) return setError(ORBX_ERR_CAPACITY, "%s: %d keypoints per frame exceed 65535", fn, cap);
    std::vector<int> table((size_t)npairs * 4);
    int maxNfv = 0;
    for (int k = 0; k < npairs; ++k) {
        if (kf_frames[k] < 0 || kf_frames[k] >= F || f_frames[k] < 0 || f_frames[k] >= F)
            return setError(ORBX_ERR_INVALID_ARG, "%s: pair %d is (%d, %d) of %d frames", fn, k, kf_frames[k], f_frames[k], F);
        table[4 * k] = kf_frames[k]; table[4 * k + 1] = voc->batchNFv[kf_frames[k]];
        table[4 * k + 2] = f_frames[k]; table[4 * k + 3] = voc->batchNFv[f_frames[k]];
        if (table[4 * k + 3] > 0) maxNfv = std::max(maxNfv, table[4 * k + 1]);
    }
    if (npairs == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t K = (size_t)npairs;
    int rc;
    if ((rc = m->d_work.ensure(K * (size_t)cap + 32 * K + 4 * K + 64)) != ORBX_OK) return rc;
    int *binOf = m->d_work.p, *result = binOf + K * cap, *pairs = result + 32 * K;
    hipStream_t st = m->stream;
    if ((rc = orderAfterExtractors(m, ex, ex)) != ORBX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(voc->stream));   // (the transform returned synchronised; a later one on the same vocabulary would have changed batchEx)
    HIP_TRY(hipMemcpyAsync(pairs, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));   // (the table is a local vector)
    HIP_TRY(orbxLaunchSearchByBoWFrames(st, voc->batchFvNode, voc->batchFvStart, voc->batchFvFeature, ex->d_kps.p, ex->d_desc.p,
                                        (const signed char *)d_valid, cap, pairs, npairs, maxNfv, nnratio, check_orientation != 0, keyframe_pair,
                                        (int *)d_matches, binOf, result));
    HIP_TRY(hipMemcpy2DAsync(d_nmatches, sizeof(int), result, 32 * sizeof(int), sizeof(int), K, hipMemcpyDeviceToDevice, st));
    return orderExtractorsAfter(m, ex, ex);
}

// int ORBmatcher::SearchForTriangulation(pKF1, pKF2, ..) for ONE keyframe against a list of neighbours, everything where the extractor and the
// vocabulary left it on the device (LocalMapping::CreateNewMapPoints, reference src/LocalMapping.cc:466 inside the loop over vpNeighKFs):
// keyframes = frames of `ex`'s last batch, FeatureVectors = the device results of the last orbx_vocabulary_transform_batch(voc, ex, ..).
// Two launches for all neighbours (k_sft_match_batch: grid z = neighbour, its argument block - pointers + two-view geometry - from a table).
extern "C" int orbx_search_for_triangulation_frames(orbx_matcher *m, orbx_vocabulary *voc, orbx_extractor *ex, int ref_frame, const int32_t *neighbours,
                                                    int nneigh, const float *k4, const float *dist, int ndist, const float *knew4,
                                                    const orbx_triangulation_geometry *geometry, const uint8_t *d_has_map_point,
                                                    const float *d_u_right, int only_stereo, int coarse, int check_orientation,
                                                    void *d_matches12, void *d_nmatches)
{
    ORBX_API_ENTRY;
    const char *fn = "orbx_search_for_triangulation_frames";
    if (!m || !voc || !ex || nneigh < 0 || (nneigh > 0 && (!neighbours || !geometry || !d_has_map_point || !d_matches12 || !d_nmatches)))
        return setError(ORBX_ERR_INVALID_ARG, "%s: null argument", fn);
    if (ex->prm.device != m->device || voc->device != m->device) return setError(ORBX_ERR_INVALID_ARG, "%s: matcher, vocabulary and extractor must live on the same device", fn);
    if (voc->batchEx != ex || voc->batchFrames <= 0 || voc->batchCap != ex->maxKeypoints)
        return setError(ORBX_ERR_INVALID_ARG, "%s: call orbx_vocabulary_transform_batch(voc, ex, ..) on this batch first (its FeatureVectors are read on the device)", fn);
    const int F = voc->batchFrames, cap = ex->maxKeypoints, nlev = ex->prm.nlevels;
    if (voc->batchSerial != ex->batchSerial || F > ex->lastFrames)
        return setError(ORBX_ERR_INVALID_ARG, "%s: the extractor ran another batch since orbx_vocabulary_transform_batch - its keypoints no longer belong to the cached FeatureVectors; transform again", fn);
    if (ref_frame < 0 || ref_frame >= F) return setError(ORBX_ERR_INVALID_ARG, "%s: reference keyframe %d of %d", fn, ref_frame, F);
    for (int k = 0; k < nneigh; ++k) {
        if (neighbours[k] < 0 || neighbours[k] >= F) return setError(ORBX_ERR_INVALID_ARG, "%s: neighbour %d is keyframe %d of %d", fn, k, neighbours[k], F);
        if (geometry[k].model != 0 && geometry[k].model != 1) return setError(ORBX_ERR_INVALID_ARG, "%s: model must be 0 (pinhole) or 1 (KannalaBrandt8)", fn);
        if (geometry[k].rig) return setError(ORBX_ERR_INVALID_ARG, "%s: single-camera keyframes only (NLeft == -1)", fn);
    }
    int rc = checkCamera(k4, dist, ndist, knew4 ? knew4 : k4, fn);
    if (rc != ORBX_OK) return rc;
    if (nneigh == 0) return ORBX_OK;
    HIP_TRY(hipSetDevice(m->device));
    const size_t FF = (size_t)F, K = (size_t)nneigh;
    if ((rc = m->d_bKps.ensure(FF * cap)) != ORBX_OK) return rc;
    if ((rc = m->d_work.ensure(K * (size_t)cap + 32 * K + K * ((orbxSftArgsBytes() + 3) / 4) + 64 + 8)) != ORBX_OK) return rc;
    if ((rc = m->d_prev.ensure(3 * ORBX_MAX_LEVELS)) != ORBX_OK) return rc;
    orbx_keypoint *kun = m->d_bKps.p;
    int *binOf = m->d_work.p, *result = binOf + K * cap;
    float *scale = m->d_prev.p, *sigma = scale + ORBX_MAX_LEVELS;
    hipStream_t st = m->stream;
    if ((rc = orderAfterExtractors(m, ex, ex)) != ORBX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(voc->stream));   // (the transform returned synchronised; a later one on the same vocabulary would have changed batchEx)
    HIP_TRY(hipMemcpyAsync(scale, ex->scale.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(sigma, ex->sigma2.data(), (size_t)nlev * sizeof(float), hipMemcpyHostToDevice, st));
    if (ndist == 0 || dist[0] == 0.0)
        HIP_TRY(hipMemcpyAsync(kun, ex->d_kps.p, FF * cap * sizeof(orbx_keypoint), hipMemcpyDeviceToDevice, st));
    else
        HIP_TRY(orbxLaunchUndistort(st, ex->d_kps.p, kun, cap, k4, dist, ndist, knew4 ? knew4 : k4, F, cap, cap, ex->d_n.p));
    std::vector<OrbxSftHost> hs((size_t)nneigh);
    for (int k = 0; k < nneigh; ++k) {
        const int f1 = ref_frame, f2 = neighbours[k];
        const orbx_triangulation_geometry *g = geometry + k;
        OrbxSftHost &h = hs[(size_t)k];
        memset(&h, 0, sizeof h);
        h.node1 = voc->batchFvNode + (size_t)f1 * cap; h.start1 = voc->batchFvStart + (size_t)f1 * (cap + 1); h.feat1 = voc->batchFvFeature + (size_t)f1 * cap;
        h.nfv1 = voc->batchNFv[f1];
        h.node2 = voc->batchFvNode + (size_t)f2 * cap; h.start2 = voc->batchFvStart + (size_t)f2 * (cap + 1); h.feat2 = voc->batchFvFeature + (size_t)f2 * cap;
        h.nfv2 = voc->batchNFv[f2];
        h.keys1 = kun + (size_t)f1 * cap; h.desc1 = ex->d_desc.p + (size_t)f1 * cap * 32; h.hasMp1 = (const signed char *)d_has_map_point + (size_t)f1 * cap;
        h.keys2 = kun + (size_t)f2 * cap; h.desc2 = ex->d_desc.p + (size_t)f2 * cap * 32; h.hasMp2 = (const signed char *)d_has_map_point + (size_t)f2 * cap;
        h.uRight1 = d_u_right ? d_u_right + (size_t)f1 * cap : nullptr; h.uRight2 = d_u_right ? d_u_right + (size_t)f2 * cap : nullptr;
        h.nLeft1 = -1; h.nLeft2 = -1; h.n1 = cap;
        h.scale2 = scale; h.sigma1 = sigma; h.sigma2 = sigma;
        memcpy(h.ep, g->ep, sizeof h.ep); memcpy(h.F, g->F12, sizeof h.F);
        h.model = g->model; h.rig = 0; h.onlyStereo = only_stereo != 0; h.coarse = coarse != 0; h.checkOri = check_orientation != 0;
        memcpy(h.camL, g->cam_left, sizeof h.camL); memcpy(h.camR, g->cam_right, sizeof h.camR); h.precision = g->precision;
        memcpy(h.R, g->R, sizeof h.R); memcpy(h.t, g->t, sizeof h.t);
        h.matches = (int *)d_matches12 + (size_t)k * cap; h.binOf = binOf + (size_t)k * cap; h.result = result + 32 * (size_t)k;
        h.maxPerNode1 = 8;   // (the per-node feature counts are on the device: a fixed second grid dimension, the kernel strides over a node's features)
    }
    {   // all neighbours in two launches: the argument blocks (pointers into the device arrays + the pair's geometry) go up as one table
        const size_t tb = orbxSftArgsBytes();
        std::vector<uint8_t> table(K * tb);
        uint8_t *dTable = (uint8_t *)(((uintptr_t)(result + 32 * K) + 15) & ~(uintptr_t)15);
        HIP_TRY(orbxLaunchSearchForTriangulationBatch(st, hs.data(), nneigh, table.data(), dTable, cap));
        HIP_TRY(hipStreamSynchronize(st));   // (the table is a local vector: its copy must be done before it goes)
    }
    HIP_TRY(hipMemcpy2DAsync(d_nmatches, sizeof(int), result, 32 * sizeof(int), sizeof(int), K, hipMemcpyDeviceToDevice, st));
    return orderExtractorsAfter(m, ex, ex);
}
