// orbx_internal.h — structures shared between the host orchestration (orbx_api.cpp) and the gfx950
// kernels (orbx_kernels.hip).  Not part of the public ABI (include/orbx.h).
#ifndef ORBX_INTERNAL_H
#define ORBX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define ORBX_MAX_LEVELS 16
#define ORBX_EDGE 19          // EDGE_THRESHOLD, reference src/ORBextractor.cc:101
#define ORBX_BORDER 16        // EDGE_THRESHOLD - 3 (FAST ROI origin), :960
#define ORBX_FAST_MARGIN 19   // first pixel that can be a corner: border + 3
#define ORBX_HALF_PATCH 15    // :100

// Candidate / selected-keypoint packing: x | y << 12 | score << 24, coordinates relative to the 16-px border.
#define ORBX_PACK(x, y, s) ((uint32_t)(x) | ((uint32_t)(y) << 12) | ((uint32_t)(s) << 24))

// Geometry of one pyramid level for the current image size (host computes; kernels take it by value).
struct OrbxLevelGeom {
    int w, h;              // level size (reference src/ORBextractor.cc:1313)
    int pitch;             // bytes per row in the handle's level buffers (multiple of 64)
    long long frameStride; // bytes between consecutive frames in the level buffers
    // FAST cell grid (reference tileCalcKeypoints, src/ORBextractor.cc:870-886)
    int wCell, hCell;
    int ncx, ncy;          // cells with a non-empty interior
    // tiles of the fused pyramid+FAST kernel: one cell-row x one group of cells
    int groupCells, nGroups, nTiles;
    int slotCap;           // u32 entries per tile candidate slot
    long long candBase;    // offset (entries) of this level's slots inside a frame's candidate area
    int tileBase;          // offset of this level's tiles inside a frame's tile-count array
    int quota;             // mnFeaturesPerLevel[level]
    int selBase;           // offset of this level's selected keypoints inside a frame's selection area
    // LDS layout of the fused kernel
    int tilePitch;         // bytes per LDS tile row of the pixel tile (multiple of 16)
    int scorePitch;        // bytes per row of the score tile: interior columns only (multiple of 16)
    int tileRows;          // max rows held in LDS
    int bands;             // 1: edge tiles stage the rows beyond their 3-px halo as margin bands in the score tile's bytes (OrbxPfTile::bandTop)
    int wpc;               // 32-bit words per cell row in the NMS bitmaps
    int rsMaxDx, rsMaxDy;  // max destination columns / rows of the next level produced by one tile
    int rightHalo;         // columns staged beyond a tile's last interior column when another column group follows (3 = FAST ring; resize: more)
    int gqCap;             // capacity of the pre-test group queue: 2 groups per strip x all strips of the tile, or fewer where that keeps
                           // a fourth workgroup resident per CU (a round that queues more takes the dense path over every group of the tile)
    int ldsBytes;
    // byte offsets of the LDS sub-buffers (the pixel tile starts at 0): score tile, NMS bitmaps (directly behind the score tile: one
    // zeroing loop), resize x / y tables and - in the same bytes, first written after the barrier behind the pre-test - the corner
    // queue, group queue, per-column cell info (oGqm: unused)
    int oScore, oBits, oXt, oYt, oGq, oCq, oGqm, oColInfo;
    // LDS layout of k_pyr_fast_cw (a wave per FAST cell): pixel tile | score tile | resize tables (never aliased: no barrier separates the
    // resize from the cell phase) | one ring of pre-test survivors per wave | one corner / keypoint list per cell
    int cwXt, cwYt, cwGq, cwCl, cwClCap, cwLdsBytes;
};

// One tile of the fused pyramid+FAST kernel (cell-row x group of cells), built by the host per level and image size: everything
// that depends only on the tile, so that the kernel reads it with scalar loads where it needs it instead of deriving ~40 values in
// its prologue and keeping them live (128 B per tile).
struct OrbxPfTile {
    int ty0, THi;          // first interior (corner-capable) row, interior rows
    int ly0, nRowsL;       // first staged row, staged rows (interior + 3-px halo; edge tiles reach the image border)
    int rowOff;            // ty0 - ly0
    int dy0, dy1;          // destination rows of the next level owned by this tile's cell-row
    int nVec;              // 16-byte vectors of the staged tile (nRowsL * nV)
    int cInt0, cInt1;      // LDS columns of the interior [tx0, tx1) + colOf0
    int colOf0;            // LDS column of image column x is x + colOf0
    int ax0, lx1;          // staged image columns [ax0, lx1), ax0 16-aligned
    int nV;                // 16-byte vectors per staged row
    int nCells, TWi;       // cells of the group, interior width
    int s0, NS;            // first 8-pixel strip, number of strips (strip s covers LDS columns [8 s + 4, 8 s + 12))
    int dx0, dx1;          // destination columns of the next level owned by this group
    int q0, nq, RC;        // resize: first 4-pixel group of a destination row, groups, row classes (512 / nq)
    // 20-bit reciprocals, ceil(2^20 / n): i / n = (i * magic) >> 20, checked by the host for the range the kernel uses them on
    int nVMagic;           // nV, i < 1024 (the first two staging passes)
    int nqMagic;           // nq, i < 512 (a thread's row class in the resize phase); 0 when nq >= 512
    // Margin bands (round 5): the image rows above the first / below the last cell-row's 3-px halo are sources of the next level only.
    // Edge tiles stage them into the bytes of the score tile + NMS bitmaps (free until the FAST rounds; the resize tables hold LDS row
    // ADDRESSES, so a source row may live anywhere) instead of the pixel tile: the pixel tile of EVERY workgroup of the launch is
    // hCell + 6 rows instead of up to hCell + 22, which is what keeps a fourth workgroup resident per CU with 360-px tiles.
    // Staged image rows: [ly0 - bandTop, ly0 + nRowsL + bandBot); nVec counts all of them.
    int bandTop, bandBot;
    int pad[5];
};
static_assert(sizeof(OrbxPfTile) == 128, "OrbxPfTile is read with aligned scalar loads");

// Bilinear-resize coefficient tables for producing level l+1 from level l (OpenCV fixed point,
// SURVEY App. A.1).  Device arrays; one set per level transition.
struct OrbxResizeTabs {
    const int *xofs;   // [dw] source column of the left tap (clamped)
    const int *xab;    // [dw] a0 | a1 << 16 (int16 each, 11-bit weights)
    const int *yofs;   // [dh] unclipped source row of the top tap
    const int *yab;    // [dh] b0 | b1 << 16
    const int *rowBeg; // [ncy + 1] first dst row owned by each cell-row of the source level
    const int *colBeg; // [nGroups + 1] first dst column owned by each column group
    const OrbxPfTile *pfTiles;  // [nTiles] tile records of the SOURCE level (also set for the last level, which has no resize tables)
};

struct OrbxLevelPtrs {
    const uint8_t *raw[ORBX_MAX_LEVELS];
    const uint8_t *blur[ORBX_MAX_LEVELS];
    int rawPitch[ORBX_MAX_LEVELS];
    long long rawFrameStride[ORBX_MAX_LEVELS];
    int blurPitch[ORBX_MAX_LEVELS];
    long long blurFrameStride[ORBX_MAX_LEVELS];
    int w[ORBX_MAX_LEVELS], h[ORBX_MAX_LEVELS];
};

struct OrbxTables {
    float scale[ORBX_MAX_LEVELS];
    int quota[ORBX_MAX_LEVELS];
    int selBase[ORBX_MAX_LEVELS];
    int patchSize[ORBX_MAX_LEVELS];  // (int)(31 * scale)
    int umax[16];
    unsigned long long umaxPacked;  // umax[v] in nibble v
    int taps[7];
    int nlevels;
    int trigMode;
};

// ---- kernel launchers (orbx_kernels.hip) -------------------------------------------------------------
hipError_t orbxInitDevice();
hipError_t orbxLaunchPyrFast(hipStream_t st, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch,
                             long long srcFrameStride, uint8_t *dst, int dw, int dh, int dstPitch,
                             long long dstFrameStride, OrbxResizeTabs tabs, uint32_t *cand, long long candFrameStride,
                             int *tileCount, int tilesPerFrame, int iniTh, int minTh);
hipError_t orbxLaunchBlur(hipStream_t st, int nframes, int w, int h, const uint8_t *src, int srcPitch,
                          long long srcFrameStride, uint8_t *dst, int dstPitch, long long dstFrameStride,
                          const int taps[7]);
hipError_t orbxLaunchQuadtree(hipStream_t st, int nframes, int nlevels, const OrbxLevelGeom *geomsHost,
                              const OrbxLevelGeom *geomsDev, const uint32_t *cand, long long candFrameStride,
                              const int *tileCount, int tilesPerFrame, uint32_t *scratchA, uint32_t *scratchB,
                              uint32_t *sel, int selFrameStride, int *selCount, int maxQuota, int levelBase, int levelsRun,
                              uint8_t *nodeScratch, size_t nodeScratchBytes, int which = 3);
#define ORBX_QT_LEVEL_SYNC 1  // `which`: the level-synchronous kernel (first and, for large levels, second launch)
#define ORBX_QT_SERIAL 2      //          the serial kernel for what the level-synchronous one deferred
size_t orbxQuadtreeScratchBytes(int nframes, const OrbxLevelGeom *geomsHost, int levelBase, int levelsRun);
hipError_t orbxLaunchPack(hipStream_t st, int nframes, OrbxTables tabs, const uint32_t *sel, int selFrameStride,
                          const int *selCount, int lap0, int lap1, void *kps, int cap, uint32_t *workKey,
                          int *workPos, int *nOut, int *monoOut);
hipError_t orbxLaunchOrientDesc(hipStream_t st, int nframes, OrbxTables tabs, OrbxLevelPtrs lv, const uint32_t *workKey,
                                const int *workPos, const int *nOut, void *kps, uint8_t *desc, int cap);

// matcher (orbx_match.hip)
hipError_t orbxLaunchHammingMatrix(hipStream_t st, const uint8_t *q, int nq, const uint8_t *t, int nt, int *dist);
int orbxKnn2Splits(int npairs, int nq, int nt, int forceSplit);
hipError_t orbxLaunchKnn2Batch(hipStream_t st, int npairs, const uint8_t *q, long long qPairStride, const int *nqDev, const int *qBeginDev,
                               int nq, const uint8_t *t, long long tPairStride, const int *ntDev, const int *tBeginDev, int nt,
                               int nSplit, uint32_t *part, int *idx, int *dist);
// a batch of stereo pairs in one launch (NULL: one pair, host-side counts)
struct OrbxStereoBatch {
    int npairs;
    long long kpStrideL, kpStrideR;  // keypoint / descriptor records between consecutive pairs
    long long outStride;             // result records between consecutive pairs
    const int *nlDev, *nrDev;        // per-pair keypoint counts on the device
};
hipError_t orbxLaunchStereoRows(hipStream_t st, const void *kl, const uint8_t *dl, int nl, const void *kr,
                                const uint8_t *dr, int nr, int nRows, const float *scaleDev, float minD, float maxD,
                                int *bestIdx, int *bestDist, const OrbxStereoBatch *batch, int *rowStart, int *rowItems, int itemStride);
// scratch of the row buckets: (nRows + 1) ints of row starts and itemStride = orbxStereoBandItems(nr, max scale) ints of items per pair
static inline int orbxStereoBandItems(int nr, float maxScale) { return nr * ((int)(4.0f * maxScale) + 4); }

// stereo refinement (orbx_match.hip); `levels` points to a host StereoLevelsHost with the layout below
struct OrbxStereoLevels {
    const uint8_t *left[ORBX_MAX_LEVELS], *right[ORBX_MAX_LEVELS];
    int pitchL[ORBX_MAX_LEVELS], pitchR[ORBX_MAX_LEVELS];
    int colsR[ORBX_MAX_LEVELS];
    float scale[ORBX_MAX_LEVELS], invScale[ORBX_MAX_LEVELS];
    long long frameStrideL[ORBX_MAX_LEVELS], frameStrideR[ORBX_MAX_LEVELS];
};
hipError_t orbxLaunchStereoRefine(hipStream_t st, const void *levels, const void *kl, int nl, const void *kr, const int *bestIdx,
                                  const int *bestDist, float mbf, float minD, float maxD, float *uRight, float *depth, int *sadDist,
                                  const OrbxStereoBatch *batch);

hipError_t orbxLaunchBowTransform(hipStream_t st, const int *childStart, const int *childNode, const void *childDesc, const int *wordOfNode,
                                  const double *weightOfNode, int L, int scoring, int weighting, const void *feat, int n, int levelsup,
                                  int *featWord, int *featNode, double *featWeight, int *bowWord, double *bowValue, int *fvNode, int *fvStart,
                                  int *fvFeature, int *counts);
hipError_t orbxLaunchSearchByBoW(hipStream_t st, const int *kfNode, const int *kfStart, const int *kfFeat, int kfNfv, const void *kfKeys,
                                 const uint8_t *kfDesc, const signed char *kfValid, const int *fNode, const int *fStart, const int *fFeat,
                                 int fNfv, const void *fKeys, const uint8_t *fDesc, int nF, int maxNodeF, float nnratio, int checkOri,
                                 int mode, const signed char *fValid, int nOut, int *matches, int *binOf, int *result);
hipError_t orbxLaunchSearchByBoWFrames(hipStream_t st, const int *fvNode, const int *fvStart, const int *fvFeat, const void *keys,
                                       const uint8_t *desc, const signed char *valid, int cap, const void *pairs, int npairs, int maxNfv,
                                       float nnratio, int checkOri, int mode, int *matches, int *binOf, int *result);
// ORBmatcher::SearchForTriangulation (orbx_match.hip): all pointers are device pointers
struct OrbxSftHost {
    const int *node1, *start1, *feat1; int nfv1; const void *keys1; const uint8_t *desc1; const signed char *hasMp1; const float *uRight1; int nLeft1, n1;
    const int *node2, *start2, *feat2; int nfv2; const void *keys2; const uint8_t *desc2; const signed char *hasMp2; const float *uRight2; int nLeft2;
    const float *scale2, *sigma1, *sigma2;
    float ep[2], F[9]; int model, rig, onlyStereo, coarse, checkOri;
    float camL[8], camR[8], precision, R[4][9], t[4][3];
    int *matches, *binOf, *result;
    int maxPerNode1;  // most features of pKF1 under one node (the launch's second dimension)
};
hipError_t orbxLaunchSearchForTriangulation(hipStream_t st, const OrbxSftHost &h);
size_t orbxSftArgsBytes();
hipError_t orbxLaunchSearchForTriangulationBatch(hipStream_t st, const OrbxSftHost *h, int n, void *hostTable, void *dTable, int n1);
hipError_t orbxLaunchKeyframeWindowSearch(hipStream_t st, const void *keys, const uint8_t *desc, const float *uRight, const float *invSigma2,
                                          const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY, const void *pts,
                                          const uint8_t *ptDesc, int np, int chi2, int *bestIdx, int *bestDist, const signed char *blocked = nullptr,
                                          const int *owner = nullptr, int levelHi = 0);
hipError_t orbxLaunchDistinctive(hipStream_t st, const uint8_t *desc, const int *start, int npoints, int *best);
hipError_t orbxLaunchBowTransformBatch(hipStream_t st, const int *childStart, const int *childNode, const void *childDesc, const int *wordOfNode,
                                       const double *weightOfNode, int L, int scoring, int weighting, const void *feat, const int *nPerFrame,
                                       int nframes, int cap, int levelsup, int *featWord, int *featNode, double *featWeight, int *bowWord,
                                       double *bowValue, int *fvNode, int *fvStart, int *fvFeature, int *counts);
// Frame::ComputeStereoFishEyeMatches after knnMatch (orbx_match.hip); layout == FisheyeArgs there
struct OrbxFisheyeArgs {
    const void *kl, *kr;
    long long kpStrideL, kpStrideR;
    const int *nlDev, *nrDev, *monoLDev, *monoRDev;
    int nl, nr, monoL, monoR;
    const int *idx, *dist;
    long long knnStride;
    float cam1[8], cam2[8], precision;
    float R12[9], t12[3];
    float sigma2[ORBX_MAX_LEVELS];
    int *leftToRight, *rightToLeft;
    float *depth, *p3d;
    long long outStrideL, outStrideR;
    int *nMatches;
};
hipError_t orbxLaunchFisheyeTriangulate(hipStream_t st, const OrbxFisheyeArgs *args, int npairs);
hipError_t orbxLaunchUndistort(hipStream_t st, const void *in, void *out, int n, const float *k4, const float *dist, int ndist, const float *knew4,
                               int nframes = 1, long long inStride = 0, long long outStride = 0, const int *nDev = nullptr);
hipError_t orbxLaunchGridBuild(hipStream_t st, const void *k, int n, float minX, float maxX, float minY, float maxY, int *cellStart,
                               int *items, int *cellOf, int nframes = 1, long long kpStride = 0, const int *nDev = nullptr);
bool orbxSearchInitInLds(int n1, int n2, int level0of2);
hipError_t orbxLaunchSearchInit(hipStream_t st, const void *k1, const uint8_t *d1, int n1, const void *k2, const uint8_t *d2, int n2,
                                const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY, float *prev,
                                int *m12, int *m21, int *mdist, signed char *binOf, int *nmatches, int window, float nnratio,
                                int checkOri, int level0of2, void *work, int level0of1 = -1);

hipError_t orbxLaunchBlurLevels(hipStream_t st, int nframes, int nlevels, const int *w, const int *h, const uint8_t *const *src, const int *srcPitch,
                                const long long *srcFrameStride, uint8_t *const *dst, const int *dstPitch, const long long *dstFrameStride,
                                const int taps[7]);
hipError_t orbxLaunchPyrChain(hipStream_t st, int nframes, int nlevels, const uint8_t *src, int srcPitch, long long srcFrameStride,
                              uint8_t *const *dst, const int *dstPitch, const long long *dstFrameStride, const int *w, const int *h,
                              const int *const *xofs, const int *const *xab, const int *const *yofs, const int *const *yab, const int *rects,
                              int nTiles, int bufBytes, int tabBytes);
hipError_t orbxLaunchResizeLevel(hipStream_t st, int nframes, const uint8_t *src, int srcPitch, long long srcFrameStride, int sh, uint8_t *dst,
                                 int dw, int dh, int dstPitch, long long dstFrameStride, OrbxResizeTabs tabs);
size_t orbxPyrFastArgsBytes();
int orbxPyrFastThreads();  // threads per workgroup of k_pyr_fast (the resize phase deals that many lanes over row classes x pixel groups)
void orbxFillPyrFastLevelArgs(void *out, int nframes, const OrbxLevelGeom &g, const uint8_t *src, int srcPitch, long long srcFrameStride,
                              const OrbxPfTile *tiles, uint32_t *cand, long long candFrameStride, int *tileCount, int tilesPerFrame, int iniTh,
                              int minTh);
hipError_t orbxLaunchPyrFastLevels(hipStream_t st, int nframes, const OrbxLevelGeom *geomsHost, int nlevels, const void *dLevelArgs);
hipError_t orbxLaunchUnpackFrames(hipStream_t st, const uint8_t *packed, uint8_t *dst, int w, int h, long long pitch, long long frameStride, int nf);
// `work`: orbxSearchInitWorkBytes(npairs, cap1) bytes for the candidate pre-pass of SearchForInitialization (nullptr: no pre-pass)
size_t orbxSearchInitWorkBytes(int npairs, int cap1);
hipError_t orbxLaunchSearchInitBatch(hipStream_t st, int npairs, const void *k1, const uint8_t *d1, long long stride1, const int *n1Dev,
                                     const void *k2, const uint8_t *d2, long long stride2, const int *n2Dev, int cap1, int cap2, float minX,
                                     float maxX, float minY, float maxY, float *prev, int initPrev, int *m12, signed char *binOf, int *nmatches,
                                     int window, float nnratio, int checkOri, void *work);

hipError_t orbxLaunchSearchByProjection(hipStream_t st, const void *kc, const uint8_t *dc, const float *uRight, const signed char *curState,
                                        int nc, const int *cellStart, const int *items, float minX, float maxX, float minY, float maxY,
                                        const float *scale, float mbf, const void *proj, const uint8_t *dproj, int np, float th, int forward,
                                        int backward, int checkOri, int mode, float nnratio, uint32_t *cand, int *count, int *choice,
                                        int *owner, int *assign, int *result, const struct OrbxSbpBatch *batch = nullptr, int levelHi = 0,
                                        float bound = 100.f);  // mode 2 (keyframe-side searches): proj.invz = window radius, levels level - 1 .. level + levelHi
hipError_t orbxLaunchGatherResults(hipStream_t st, const int *result, int n, int *out);
// up to ORBX_COPY_PIECES device-to-device copies in one launch (the pieces of a packed host transfer to / from their buffers)
#define ORBX_COPY_PIECES 24
struct OrbxCopyPieces { void *dst[ORBX_COPY_PIECES]; const void *src[ORBX_COPY_PIECES]; unsigned bytes[ORBX_COPY_PIECES]; int n; };
hipError_t orbxLaunchCopyPieces(hipStream_t st, const OrbxCopyPieces &p);
// a batch of frames through the SearchByProjection kernels: per-frame strides (records) and device-side counts
struct OrbxSbpBatch { int nframes; long long kcStride, projStride; const int *ncDev, *npDev; };
#define ORBX_SBP_CAP 512  // == SBP_CAP in orbx_match.hip
// the CurrentFrame.Nleft != -1 (fisheye rig) branches: kc / dc = left then right keypoints, one grid per side, steps[2 np] = the left and
// right record of every point interleaved, dproj[np]; partner[nLeft + nRight] (mode 1) or NULL
struct OrbxSbpFisheye {
    const void *kc; const uint8_t *dc; const signed char *curState; int nLeft, nRight;
    const int *cellStartL, *itemsL, *cellStartR, *itemsR;
    float minX, maxX, minY, maxY; const float *scale;
    const void *steps; const uint8_t *dproj; int np;
    float th; int forward, backward, checkOri, mode; float nnratio;
    const int *partner;
    uint32_t *cand; int *count, *choice, *head, *next, *assign, *result;
    const OrbxSbpBatch *batch = nullptr;  // a batch of frames: kc / dc / curState / partner / head / assign advance by kcStride = capL + capR records per frame
                                          // (nLeft = capL: right keypoints start at the left capacity), steps / dproj by projStride = 2 maxNp / maxNp
};
// KeyFrame window search / Fuse for a batch of keyframes (orbx_search_keyframe_windows_frames, orbx_fuse_frames)
struct OrbxKfwFrames {
    int nframes, maxNp, chi2, fuseTh;
    const void *keys; const uint8_t *desc; const float *uRight; const float *invSigma2;
    const int *cellStart, *items; long long keyStride, gridStride, itemStride;
    float minX, maxX, minY, maxY;
    const void *pts; const uint8_t *ptDesc; const int *npDev;
    int *bestIdx, *bestDist, *nFused;
};
hipError_t orbxLaunchKeyframeWindowSearchFrames(hipStream_t st, const OrbxKfwFrames *b);
// projection head of the tracking searches (orbx_project_points_frames): scalar parameters of the in-frustum mode
struct OrbxProjectArgs {
    int mode, nframes, maxNp;          // mode 0 motion model (Sophus SE3f * point), 1 Frame::isInFrustum, 2 the same for a fisheye rig (both
                                       // cameras), 3 the head of the keyframe-side window searches (orbx_window_point records)
    const float *pose;                 // [nframes][poseStride] device (or the one pose of a single call, also device)
    int poseStride;                    // floats per frame: 16 (modes 0, 1, 3), 40 (mode 2)
    int camType; float cam[8], cam2[8];  // cam2: the right camera of mode 2
    float th, scaleFactors[ORBX_MAX_LEVELS]; int tests;   // mode 3: radius = th * mvScaleFactors[level]; tests bit 0 depth, 1 viewing angle, 2 Frame bounds
    void *out2;                        // mode 2: the right camera's records
    float minX, maxX, minY, maxY, mbf, viewCosLimit, logScaleFactor, thFar;
    int nlevels, farPoints;
    const void *points; const int *npDev; int np;  // npDev NULL: np points in every frame
    void *out;
};
hipError_t orbxLaunchProjectPoints(hipStream_t st, const OrbxProjectArgs *a);
hipError_t orbxLaunchFisheyeSteps(hipStream_t st, const void *pl, const void *pr, const int *npDev, int maxNp, int nframes, int mode, void *steps);
hipError_t orbxLaunchFisheyePartner(hipStream_t st, const int *l2r, const int *r2l, const int *nlDev, const int *nrDev, int capL, int capR, int nframes,
                                    int *partner);
hipError_t orbxLaunchSearchByProjectionFisheye(hipStream_t st, const OrbxSbpFisheye &f);

#endif
