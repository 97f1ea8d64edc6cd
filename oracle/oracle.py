"""ctypes wrapper of the CPU parity oracle (oracle/liborb_oracle.so).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
checker / timed CPU baseline.  The product never imports it.  Parity status: see orb_oracle.h ("parity unpinned"
for image-dependent stages; constructor tables, umax, sampling pattern and Hamming identities are pinned
against the reference text).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_float, c_int, c_ssize_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORB_ORACLE_LIB: another build of the same source (bench.py's cpu_baseline leg times a -march=native build next to the -O3 one)
LIB_PATH = os.environ.get("ORB_ORACLE_LIB") or os.path.join(_HERE, "liborb_oracle.so")

KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)

_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing — run `make -C oracle`")
        L = ctypes.CDLL(LIB_PATH)
        L.oracle_create.restype = c_void_p
        L.oracle_create.argtypes = [c_int, c_float, c_int, c_int, c_int, c_int, c_int]
        L.oracle_destroy.argtypes = [c_void_p]
        L.oracle_destroy.restype = None
        L.oracle_get_tables.argtypes = [c_void_p] + [c_void_p] * 6
        L.oracle_max_keypoints.argtypes = [c_void_p]
        L.oracle_max_keypoints_for.argtypes = [c_void_p, c_int, c_int]
        L.oracle_extract.argtypes = [c_void_p, c_void_p, c_int, c_int, c_ssize_t, c_int, c_int, c_void_p, c_void_p, POINTER(c_int)]
        L.oracle_level_size.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]
        L.oracle_get_level.argtypes = [c_void_p, c_int, c_int, c_void_p]
        L.oracle_get_candidates.argtypes = [c_void_p, c_int, c_void_p, c_int]
        L.oracle_get_level_keypoints.argtypes = [c_void_p, c_int, c_void_p, c_int]
        L.oracle_extract_batch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_ssize_t, c_int, c_int, c_int, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_int]
        L.oracle_resize_linear.argtypes = [c_void_p, c_int, c_int, c_ssize_t, c_void_p, c_int, c_int, c_ssize_t]
        L.oracle_gaussian_blur7.argtypes = [c_void_p, c_int, c_int, c_ssize_t, c_void_p, c_ssize_t, c_int]
        L.oracle_fast_roi.argtypes = [c_void_p, c_int, c_int, c_ssize_t, c_int, c_void_p, c_int]
        L.oracle_fast_atan2.argtypes = [c_float, c_float]
        L.oracle_fast_atan2.restype = c_float
        L.oracle_distribute.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int]
        L.oracle_sort_pairs.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
        L.oracle_hamming256.argtypes = [c_void_p, c_void_p]
        L.oracle_knn2.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]
        L.oracle_stereo_rows.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float,
                                         c_void_p, c_void_p]
        L.oracle_stereo_matches.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float,
                                            c_void_p, c_void_p]
        L.oracle_kb8_triangulate_matches.argtypes = [c_void_p, c_void_p, c_float, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_float,
                                                     c_float, c_void_p]
        L.oracle_kb8_triangulate_matches.restype = c_float
        L.oracle_fisheye_stereo_matches.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_float,
                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.oracle_assign_grid.argtypes = [c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p]
        L.oracle_features_in_area.argtypes = [c_void_p, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_int,
                                              c_void_p, c_int]
        L.oracle_search_for_initialization.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float,
                                                       c_float, c_void_p, c_void_p, c_int, c_float, c_int]
        L.oracle_project_last_frame_points.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]
        L.oracle_project_last_frame_points.restype = None
        L.oracle_project_in_frustum.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_float, c_int, c_int,
                                                c_float, c_void_p, c_int, c_void_p]
        L.oracle_project_in_frustum.restype = None
        L.oracle_project_in_frustum_fisheye.argtypes = [c_void_p] * 10 + [c_float, c_float, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]
        L.oracle_project_in_frustum_fisheye.restype = None
        L.oracle_project_window_points.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float, c_void_p, c_float, c_int,
                                                   c_int, c_void_p, c_int, c_void_p]
        L.oracle_project_window_points.restype = None
        L.oracle_predict_scale.argtypes = [c_float, c_float, c_int]
        L.oracle_predict_scale.restype = c_int
        L.oracle_search_by_projection.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                                  c_int, c_float, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p]
        L.oracle_search_by_projection_points.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float,
                                                         c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p]
        L.oracle_search_by_projection_fisheye.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                                          c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p]
        L.oracle_search_by_projection_points_fisheye.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float,
                                                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                                 c_float, c_float, c_void_p]
        L.oracle_undistort_keypoints.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        L.oracle_compute_image_bounds.argtypes = [c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        L.oracle_vocabulary_create.argtypes = [c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
        L.oracle_vocabulary_create.restype = c_void_p
        L.oracle_vocabulary_destroy.argtypes = [c_void_p]
        L.oracle_vocabulary_destroy.restype = None
        L.oracle_vocabulary_words.argtypes = [c_void_p]
        L.oracle_vocabulary_transform.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 9
        L.oracle_search_by_bow.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]
        L.oracle_search_by_bow_keyframes.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                                     c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]
        L.oracle_search_keyframe_windows.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                                     c_int, c_int, c_void_p, c_void_p]
        L.oracle_search_by_sim3.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                            c_void_p]
        L.oracle_search_by_projection_sim3.argtypes = [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_int, c_void_p,
                                                       c_float, c_void_p]
        L.oracle_search_by_projection_keyframe.argtypes = [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                                           c_void_p, c_int, c_int, c_void_p]
        L.oracle_distinctive_descriptors.argtypes = [c_void_p, c_void_p, c_int, c_void_p]
        L.oracle_search_for_triangulation.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
        L.oracle_synth_frame.argtypes = [c_uint64, c_int, c_int, c_void_p, c_ssize_t]
        L.oracle_synth_frame.restype = None
        _lib = L
    return _lib


class Oracle:
    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, blur_variant=0, trig_mode=0):
        self.L = lib()
        self.h = self.L.oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th, blur_variant, trig_mode)
        if not self.h:
            raise ValueError("oracle_create rejected the parameters")
        self.nlevels = nlevels
        self.capacity = self.L.oracle_max_keypoints(self.h)

    def __del__(self):  # pragma: no cover
        try:
            if self.h:
                self.L.oracle_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        fpl = np.zeros(n, np.int32)
        um = np.zeros(16, np.int32)
        self.L.oracle_get_tables(self.h, sc.ctypes.data, isc.ctypes.data, s2.ctypes.data, is2.ctypes.data, fpl.ctypes.data, um.ctypes.data)
        return dict(scale=sc, inv_scale=isc, sigma2=s2, inv_sigma2=is2, features_per_level=fpl, umax=um)

    def extract(self, img: np.ndarray, lap=(0, 0)):
        if img is None or img.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        cap = max(self.L.oracle_max_keypoints_for(self.h, w, h), self.capacity, 64) + 64
        self._last_cap = cap
        kps = np.zeros(cap, KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = c_int()
        mono = self.L.oracle_extract(self.h, img.ctypes.data, w, h, img.strides[0], int(lap[0]), int(lap[1]), kps.ctypes.data,
                                     desc.ctypes.data, byref(n))
        return mono, kps[: n.value].copy(), desc[: n.value].copy()

    def level_size(self, level):
        w, h = c_int(), c_int()
        self.L.oracle_level_size(self.h, level, byref(w), byref(h))
        return w.value, h.value

    def level(self, level, blurred=False):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        self.L.oracle_get_level(self.h, level, int(blurred), out.ctypes.data)
        return out

    def candidates(self, level, cap=1 << 20):
        out = np.zeros(cap, np.uint32)
        n = self.L.oracle_get_candidates(self.h, level, out.ctypes.data, cap)
        return out[:n].copy()

    def level_keypoints(self, level):
        out = np.zeros(max(self.capacity, getattr(self, "_last_cap", 0)) + 64, KEYPOINT_DTYPE)
        n = self.L.oracle_get_level_keypoints(self.h, level, out.ctypes.data, len(out))
        return out[:n].copy()

    def extract_batch(self, imgs: np.ndarray, lap=(0, 0), threads=1, want_outputs=False):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        b, h, w = imgs.shape
        counts = np.zeros(b, np.int32)
        mono = np.zeros(b, np.int32)
        cap = max(self.capacity, self.L.oracle_max_keypoints_for(self.h, w, h))
        kps = np.zeros((b, cap), KEYPOINT_DTYPE) if want_outputs else None
        desc = np.zeros((b, cap, 32), np.uint8) if want_outputs else None
        self.L.oracle_extract_batch(self.h, imgs.ctypes.data, b, w, h, imgs.strides[1], int(lap[0]), int(lap[1]), threads,
                                    counts.ctypes.data, mono.ctypes.data, kps.ctypes.data if want_outputs else None,
                                    desc.ctypes.data if want_outputs else None, cap)
        return counts, mono, kps, desc


def synth_frame(frame_index, w=640, h=480):
    img = np.empty((h, w), np.uint8)
    lib().oracle_synth_frame(frame_index, w, h, img.ctypes.data, w)
    return img


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oracle_resize_linear(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], dst.ctypes.data, dw, dh, dw)
    return dst


def gaussian_blur7(src: np.ndarray, blur_variant=0) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().oracle_gaussian_blur7(src.ctypes.data, src.shape[1], src.shape[0], src.strides[0], dst.ctypes.data, dst.strides[0], blur_variant)
    return dst


def fast_roi(roi: np.ndarray, threshold: int) -> np.ndarray:
    roi = np.ascontiguousarray(roi, np.uint8)
    out = np.zeros(roi.size, np.uint32)
    n = lib().oracle_fast_roi(roi.ctypes.data, roi.shape[1], roi.shape[0], roi.strides[0], threshold, out.ctypes.data, len(out))
    return out[:n].copy()


def fast_atan2(y: float, x: float) -> float:
    return float(lib().oracle_fast_atan2(y, x))


def distribute(cand: np.ndarray, W: int, H: int, N: int) -> np.ndarray:
    cand = np.ascontiguousarray(cand, np.uint32)
    out = np.zeros(N + 64, np.int32)
    n = lib().oracle_distribute(cand.ctypes.data, len(cand), W, H, N, out.ctypes.data, len(out))
    return out[:n].copy()


def sort_pairs(count: np.ndarray, ulx: np.ndarray) -> np.ndarray:
    count = np.ascontiguousarray(count, np.int32)
    ulx = np.ascontiguousarray(ulx, np.int32)
    order = np.zeros(len(count), np.int32)
    lib().oracle_sort_pairs(count.ctypes.data, ulx.ctypes.data, len(count), order.ctypes.data)
    return order


def hamming256(a: np.ndarray, b: np.ndarray) -> int:
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return int(lib().oracle_hamming256(a.ctypes.data, b.ctypes.data))


def knn2(q: np.ndarray, t: np.ndarray):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx = np.zeros((len(q), 2), np.int32)
    dist = np.zeros((len(q), 2), np.int32)
    lib().oracle_knn2(q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data, dist.ctypes.data)
    return idx, dist


def stereo_rows(kl, dl, kr, dr, n_rows, scale_factors, min_d, max_d):
    kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE)
    kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    bi = np.zeros(len(kl), np.int32)
    bd = np.zeros(len(kl), np.int32)
    lib().oracle_stereo_rows(kl.ctypes.data, dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data, len(kr), n_rows,
                             sf.ctypes.data, min_d, max_d, bi.ctypes.data, bd.ctypes.data)
    return bi, bd


def stereo_matches(left: Oracle, right: Oracle, kl, dl, kr, dr, mb: float, mbf: float):
    """Frame::ComputeStereoMatches on the pyramids of the two oracles' last extract(); returns (uRight, depth, n_valid_before_outlier_cut)."""
    kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE)
    kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
    dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
    ur = np.zeros(len(kl), np.float32)
    dp = np.zeros(len(kl), np.float32)
    n = lib().oracle_stereo_matches(left.h, right.h, kl.ctypes.data, dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data, len(kr),
                                    mb, mbf, ur.ctypes.data, dp.ctypes.data)
    return ur, dp, n


def assign_grid(kps, bounds):
    """Frame::AssignFeaturesToGrid (reference src/Frame.cc:387-418) -> (cell_start[64 * 48 + 1], items)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    cs = np.zeros(64 * 48 + 1, np.int32)
    items = np.zeros(max(len(kps), 1), np.int32)
    n = lib().oracle_assign_grid(kps.ctypes.data, len(kps), *[float(b) for b in bounds], cs.ctypes.data, items.ctypes.data)
    return cs, items[:n]


def features_in_area(kps, bounds, x, y, r, min_level=-1, max_level=-1):
    """Frame::GetFeaturesInArea (reference src/Frame.cc:687-753)."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    out = np.zeros(max(len(kps), 1), np.int32)
    n = lib().oracle_features_in_area(kps.ctypes.data, len(kps), *[float(b) for b in bounds], float(x), float(y), float(r), int(min_level),
                                      int(max_level), out.ctypes.data, len(out))
    return out[:n]


def search_for_initialization(k1, d1, k2, d2, prev_matched, bounds, window_size=10, nnratio=0.6, check_orientation=True):
    """ORBmatcher::SearchForInitialization (reference src/ORBmatcher1.cc:650-765) -> (nmatches, vnMatches12, vbPrevMatched)."""
    k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE)
    k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
    prev = np.array(prev_matched, np.float32).reshape(-1, 2).copy()
    m12 = np.full(len(k1), -1, np.int32)
    n = lib().oracle_search_for_initialization(k1.ctypes.data, d1.ctypes.data, len(k1), k2.ctypes.data, d2.ctypes.data, len(k2),
                                               *[float(b) for b in bounds], prev.ctypes.data, m12.ctypes.data, int(window_size),
                                               float(nnratio), int(bool(check_orientation)))
    return n, m12, prev


PROJECTION_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("invz", "<f4"), ("angle", "<f4"), ("octave", "<i4"), ("flags", "<i4")])


WORLD_POINT_DTYPE = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("angle", "<f4"),
                              ("octave", "<i4"), ("flags", "<i4")])  # oracle_world_point == orbx_world_point (44 bytes)


def project_last_frame_points(q_xyzw, t, cam_type, cam, points):
    """x3Dc = Tcw * x3Dw; invzc; uv = mpCamera->project(x3Dc) (reference src/ORBmatcher3.cc:283-293) -> PROJECTION_DTYPE records."""
    q = np.ascontiguousarray(q_xyzw, np.float32); tt = np.ascontiguousarray(t, np.float32)
    c = np.zeros(8, np.float32); c[:len(cam)] = cam
    pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
    out = np.zeros(len(pts), PROJECTION_DTYPE)
    lib().oracle_project_last_frame_points(q.ctypes.data, tt.ctypes.data, int(cam_type), c.ctypes.data, pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def project_in_frustum(Rcw, tcw, Ow, cam_type, cam, bounds, mbf, viewing_cos_limit, log_scale_factor, nlevels, points, far_points=False,
                       th_far_points=0.0):
    """Frame::isInFrustum over `points` (reference src/Frame.cc:542-600) -> PROJECTION_DTYPE records as SearchByProjection(F, vpMapPoints) reads them."""
    R = np.ascontiguousarray(Rcw, np.float32).reshape(9); tt = np.ascontiguousarray(tcw, np.float32); ow = np.ascontiguousarray(Ow, np.float32)
    c = np.zeros(8, np.float32); c[:len(cam)] = cam
    b = np.ascontiguousarray(bounds, np.float32)
    pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
    out = np.zeros(len(pts), PROJECTION_DTYPE)
    lib().oracle_project_in_frustum(R.ctypes.data, tt.ctypes.data, ow.ctypes.data, int(cam_type), c.ctypes.data, b.ctypes.data, float(mbf),
                                    float(viewing_cos_limit), float(log_scale_factor), int(nlevels), int(bool(far_points)), float(th_far_points),
                                    pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def project_in_frustum_fisheye(Rcw, tcw, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, viewing_cos_limit, log_scale_factor, nlevels, points,
                               far_points=False, th_far_points=0.0):
    """Frame::isInFrustum of a fisheye rig (reference src/Frame.cc:604-614, 1198-1272) -> (left records, right records), PROJECTION_DTYPE."""
    f32 = lambda a, n: np.ascontiguousarray(a, np.float32).reshape(n)
    arrs = [f32(Rcw, 9), f32(tcw, 3), f32(Ow, 3), f32(Rrl, 9), f32(trl, 3), f32(tlr, 3), f32(Rwc, 9)]
    c1 = np.zeros(8, np.float32); c1[:len(cam1)] = cam1
    c2 = np.zeros(8, np.float32); c2[:len(cam2)] = cam2
    b = np.ascontiguousarray(bounds, np.float32)
    pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
    ol = np.zeros(len(pts), PROJECTION_DTYPE); orr = np.zeros(len(pts), PROJECTION_DTYPE)
    lib().oracle_project_in_frustum_fisheye(*[a.ctypes.data for a in arrs], c1.ctypes.data, c2.ctypes.data, b.ctypes.data, float(viewing_cos_limit),
                                            float(log_scale_factor), int(nlevels), int(bool(far_points)), float(th_far_points), pts.ctypes.data,
                                            len(pts), ol.ctypes.data, orr.ctypes.data)
    return ol, orr


def project_window_points(q_xyzw, t, Ow, cam_type, cam, bounds, bf, th, scale_factors, log_scale_factor, tests, points):
    """The projection head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame) -> WINDOW_POINT_DTYPE records."""
    q = np.ascontiguousarray(q_xyzw, np.float32); tt = np.ascontiguousarray(t, np.float32); ow = np.ascontiguousarray(Ow, np.float32)
    c = np.zeros(8, np.float32); c[:len(cam)] = cam
    b = np.ascontiguousarray(bounds, np.float32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
    out = np.zeros(len(pts), WINDOW_POINT_DTYPE)
    lib().oracle_project_window_points(q.ctypes.data, tt.ctypes.data, ow.ctypes.data, int(cam_type), c.ctypes.data, b.ctypes.data, float(bf), float(th),
                                       sf.ctypes.data, float(log_scale_factor), len(sf), int(tests), pts.ctypes.data, len(pts), out.ctypes.data)
    return out


def predict_scale(ratio, log_scale_factor, nlevels):
    """MapPoint::PredictScale's arithmetic on an already formed float ratio (reference src/MapPoint.cc:538-544)."""
    return int(lib().oracle_predict_scale(float(np.float32(ratio)), float(np.float32(log_scale_factor)), int(nlevels)))


def search_by_projection(kc, dc, cur_state, bounds, scale_factors, proj, proj_desc, th, u_right=None, mbf=0.0, forward=False,
                         backward=False, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference src/ORBmatcher3.cc:256-467, Nleft == -1)
    -> (nmatches, assign)."""
    kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
    dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
    st = np.ascontiguousarray(cur_state, np.int8)
    pr = np.ascontiguousarray(proj, PROJECTION_DTYPE)
    pd = np.ascontiguousarray(proj_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    assign = np.full(len(kc), -1, np.int32)
    n = lib().oracle_search_by_projection(kc.ctypes.data, dc.ctypes.data, None if ur is None else ur.ctypes.data, st.ctypes.data, len(kc),
                                          *[float(b) for b in bounds], sf.ctypes.data, len(sf), float(mbf), pr.ctypes.data, pd.ctypes.data,
                                          len(pr), float(th), int(forward), int(backward), int(bool(check_orientation)), assign.ctypes.data)
    return n, assign


def search_by_projection_points(kc, dc, cur_state, bounds, scale_factors, mps, mp_desc, th=1.0, nnratio=0.6, u_right=None):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ..) (reference src/ORBmatcher1.cc:45-215, Nleft == -1) -> (nmatches, assign).
    mps: PROJECTION_DTYPE with u, v = mTrackProjX / Y, invz = mTrackProjXR, angle = mTrackViewCos, octave = mnTrackScaleLevel."""
    kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
    dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
    st = np.ascontiguousarray(cur_state, np.int8)
    pr = np.ascontiguousarray(mps, PROJECTION_DTYPE)
    pd = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    assign = np.full(len(kc), -1, np.int32)
    n = lib().oracle_search_by_projection_points(kc.ctypes.data, dc.ctypes.data, None if ur is None else ur.ctypes.data, st.ctypes.data,
                                                 len(kc), *[float(b) for b in bounds], sf.ctypes.data, len(sf), pr.ctypes.data,
                                                 pd.ctypes.data, len(pr), float(th), float(nnratio), assign.ctypes.data)
    return n, assign


def search_by_projection_fisheye(kc, dc, n_left, cur_state, bounds, scale_factors, proj_l, proj_r, proj_desc, th, forward=False,
                                 backward=False, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), CurrentFrame.Nleft != -1 (reference src/ORBmatcher3.cc:256-467).
    kc / dc: mvKeys then mvKeysRight / mDescriptors; -> (nmatches, assign[n_left + n_right])."""
    kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
    dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
    st = np.ascontiguousarray(cur_state, np.int8)
    pl = np.ascontiguousarray(proj_l, PROJECTION_DTYPE)
    pr = np.ascontiguousarray(proj_r, PROJECTION_DTYPE)
    pd = np.ascontiguousarray(proj_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    assert len(pl) == len(pr) == len(pd) and len(st) == len(kc) == len(dc)
    assign = np.full(len(kc), -1, np.int32)
    n = lib().oracle_search_by_projection_fisheye(kc.ctypes.data, dc.ctypes.data, int(n_left), len(kc) - int(n_left), st.ctypes.data,
                                                  *[float(b) for b in bounds], sf.ctypes.data, len(sf), pl.ctypes.data, pr.ctypes.data,
                                                  pd.ctypes.data, len(pl), float(th), int(forward), int(backward),
                                                  int(bool(check_orientation)), assign.ctypes.data)
    return n, assign


def search_by_projection_points_fisheye(kc, dc, n_left, cur_state, bounds, scale_factors, mp_l, mp_r, mp_desc, left_to_right, right_to_left,
                                        th=1.0, nnratio=0.6):
    """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ..), F.Nleft != -1 (reference src/ORBmatcher1.cc:45-215) -> (nmatches, assign)."""
    kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
    dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
    st = np.ascontiguousarray(cur_state, np.int8)
    pl = np.ascontiguousarray(mp_l, PROJECTION_DTYPE)
    pr = np.ascontiguousarray(mp_r, PROJECTION_DTYPE)
    pd = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
    sf = np.ascontiguousarray(scale_factors, np.float32)
    l2r = np.ascontiguousarray(left_to_right, np.int32)
    r2l = np.ascontiguousarray(right_to_left, np.int32)
    assert len(pl) == len(pr) == len(pd) and len(st) == len(kc) == len(dc) and len(l2r) == n_left and len(r2l) == len(kc) - n_left
    assign = np.full(len(kc), -1, np.int32)
    n = lib().oracle_search_by_projection_points_fisheye(kc.ctypes.data, dc.ctypes.data, int(n_left), len(kc) - int(n_left), st.ctypes.data,
                                                         *[float(b) for b in bounds], sf.ctypes.data, len(sf), pl.ctypes.data,
                                                         pr.ctypes.data, pd.ctypes.data, len(pl), l2r.ctypes.data, r2l.ctypes.data,
                                                         float(th), float(nnratio), assign.ctypes.data)
    return n, assign


def _camera(k4, dist, knew4):
    k4 = np.ascontiguousarray(k4, np.float32); d = np.ascontiguousarray(dist, np.float32).ravel()
    kn = k4 if knew4 is None else np.ascontiguousarray(knew4, np.float32)
    assert k4.shape == (4,) and kn.shape == (4,) and len(d) <= 12
    return k4, d, kn


def undistort_keypoints(kps, k4, dist, knew4=None):
    """Frame::UndistortKeyPoints (reference src/Frame.cc:777-811) -> mvKeysUn.  k4 = (fx, fy, cx, cy), dist = mDistCoef."""
    kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
    k4, d, kn = _camera(k4, dist, knew4)
    out = np.empty_like(kps)
    lib().oracle_undistort_keypoints(kps.ctypes.data, len(kps), k4.ctypes.data, d.ctypes.data, len(d), kn.ctypes.data, out.ctypes.data)
    return out


def compute_image_bounds(cols, rows, k4, dist, knew4=None):
    """Frame::ComputeImageBounds (reference src/Frame.cc:813-845) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
    k4, d, kn = _camera(k4, dist, knew4)
    b = np.zeros(4, np.float32)
    lib().oracle_compute_image_bounds(int(cols), int(rows), k4.ctypes.data, d.ctypes.data, len(d), kn.ctypes.data, b.ctypes.data)
    return tuple(float(v) for v in b)


def kb8_triangulate_matches(cam1, cam2, pt1, pt2, R12, t12, sigma_level, unc, precision=1e-6):
    """KannalaBrandt8::TriangulateMatches (reference src/CameraModels/KannalaBrandt8.cpp:306-375) -> (depth or negative code, p3D[3])."""
    c1 = np.ascontiguousarray(cam1, np.float32); c2 = np.ascontiguousarray(cam2, np.float32)
    R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32)
    p = np.zeros(3, np.float32)
    d = lib().oracle_kb8_triangulate_matches(c1.ctypes.data, c2.ctypes.data, precision, float(pt1[0]), float(pt1[1]), float(pt2[0]), float(pt2[1]),
                                             R.ctypes.data, t.ctypes.data, float(sigma_level), float(unc), p.ctypes.data)
    return float(d), p


def fisheye_stereo_matches(kl, dl, mono_left, kr, dr, mono_right, cam1, cam2, R12, t12, level_sigma2, precision=1e-6):
    """Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:1156-1196) ->
    (nMatches, mvLeftToRightMatch[nl], mvRightToLeftMatch[nr], mvDepth[nl], mvStereo3Dpoints[nl, 3])."""
    kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
    dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32); dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
    c1 = np.ascontiguousarray(cam1, np.float32); c2 = np.ascontiguousarray(cam2, np.float32)
    R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32)
    ls = np.ascontiguousarray(level_sigma2, np.float32)
    l2r = np.full(len(kl), -1, np.int32); r2l = np.full(len(kr), -1, np.int32)
    depth = np.full(len(kl), -1, np.float32); p3d = np.zeros((len(kl), 3), np.float32)
    n = lib().oracle_fisheye_stereo_matches(kl.ctypes.data, dl.ctypes.data, len(kl), int(mono_left), kr.ctypes.data, dr.ctypes.data, len(kr),
                                            int(mono_right), c1.ctypes.data, c2.ctypes.data, precision, R.ctypes.data, t.ctypes.data,
                                            ls.ctypes.data, l2r.ctypes.data, r2l.ctypes.data, depth.ctypes.data, p3d.ctypes.data)
    return int(n), l2r, r2l, depth, p3d


class Vocabulary:
    """DBoW2 TemplatedVocabulary<FORB> as Frame::ComputeBoW uses it (reference Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1127-1261).
    parent / is_leaf / desc / weight: one entry per node in loadFromTextFile's order, node 0 = root."""

    def __init__(self, k, L, scoring, weighting, parent, is_leaf, desc, weight):
        self.parent = np.ascontiguousarray(parent, np.int32)
        self.is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.weight = np.ascontiguousarray(weight, np.float64)
        n = len(self.parent)
        assert len(self.is_leaf) == n and len(self.desc) == n and len(self.weight) == n
        self._h = lib().oracle_vocabulary_create(int(k), int(L), int(scoring), int(weighting), n, self.parent.ctypes.data,
                                                 self.is_leaf.ctypes.data, self.desc.ctypes.data, self.weight.ctypes.data)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().oracle_vocabulary_destroy(self._h)
        except Exception:  # interpreter shutdown
            pass
        self._h = None

    def size(self):
        return lib().oracle_vocabulary_words(self._h)

    def transform(self, desc, levelsup=4):
        """-> dict(feat_word, feat_node, bow_word, bow_value, fv_node, fv_start, fv_feature): BowVector in ascending word order,
        FeatureVector as CSR over ascending node ids."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        fw = np.zeros(n, np.int32); fn = np.zeros(n, np.int32)
        bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64); nb = np.zeros(1, np.int32)
        vn = np.zeros(max(n, 1), np.int32); vs = np.zeros(n + 1, np.int32); vf = np.zeros(max(n, 1), np.int32); nv = np.zeros(1, np.int32)
        lib().oracle_vocabulary_transform(self._h, d.ctypes.data, n, int(levelsup), fw.ctypes.data, fn.ctypes.data, bw.ctypes.data,
                                          bv.ctypes.data, nb.ctypes.data, vn.ctypes.data, vs.ctypes.data, vf.ctypes.data, nv.ctypes.data)
        nb, nv = int(nb[0]), int(nv[0])
        return dict(feat_word=fw, feat_node=fn, bow_word=bw[:nb].copy(), bow_value=bv[:nb].copy(), fv_node=vn[:nv].copy(),
                    fv_start=vs[:nv + 1].copy(), fv_feature=vf[:int(vs[nv])].copy())


def search_by_bow(kf_fv, kf_keys, kf_desc, kf_valid, f_fv, f_keys, f_desc, nnratio=0.7, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (reference src/ORBmatcher1.cc:225-427, Nleft == -1) -> (nmatches, matches).
    kf_fv / f_fv: dicts with fv_node, fv_start, fv_feature (Vocabulary.transform); kf_valid[i] = map point i exists and is not bad."""
    kk = np.ascontiguousarray(kf_keys, KEYPOINT_DTYPE); fk = np.ascontiguousarray(f_keys, KEYPOINT_DTYPE)
    kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32); fd = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
    kv = np.ascontiguousarray(kf_valid, np.uint8)
    a = [np.ascontiguousarray(kf_fv[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
    b = [np.ascontiguousarray(f_fv[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
    matches = np.full(len(fk), -1, np.int32)
    n = lib().oracle_search_by_bow(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]), kk.ctypes.data, kd.ctypes.data, kv.ctypes.data,
                                   b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, len(b[0]), fk.ctypes.data, fd.ctypes.data, len(fk),
                                   float(nnratio), int(bool(check_orientation)), matches.ctypes.data)
    return n, matches


def search_by_bow_keyframes(fv1, keys1, desc1, valid1, fv2, keys2, desc2, valid2, nnratio=0.8, check_orientation=True):
    """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher2.cc:37-177) -> (nmatches, matches12[n1])."""
    k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
    v1 = np.ascontiguousarray(valid1, np.uint8); v2 = np.ascontiguousarray(valid2, np.uint8)
    a = [np.ascontiguousarray(fv1[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
    b = [np.ascontiguousarray(fv2[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
    matches = np.full(len(k1), -1, np.int32)
    n = lib().oracle_search_by_bow_keyframes(a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]), k1.ctypes.data, d1.ctypes.data,
                                             v1.ctypes.data, len(k1), b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, len(b[0]),
                                             k2.ctypes.data, d2.ctypes.data, v2.ctypes.data, len(k2), float(nnratio),
                                             int(bool(check_orientation)), matches.ctypes.data)
    return n, matches


def search_for_triangulation(kf1, kf2, geometry, only_stereo=False, coarse=False, check_orientation=True):
    """ORBmatcher::SearchForTriangulation (reference src/ORBmatcher2.cc:179-418) -> (nmatches, pairs[nmatches, 2], matches12[n1]).
    kf1 / kf2 / geometry: the ctypes structures of the package's keyframe_view() / triangulation_geometry() - oracle_keyframe_view and
    oracle_triangulation_geometry have the same layout."""
    import ctypes
    v1, v2 = kf1[0], kf2[0]
    matches = np.full(v1.n, -1, np.int32)
    pairs = np.zeros((max(v1.n, 1), 2), np.int32)
    n = lib().oracle_search_for_triangulation(ctypes.addressof(v1), ctypes.addressof(v2), ctypes.addressof(geometry), int(bool(only_stereo)),
                                              int(bool(coarse)), int(bool(check_orientation)), matches.ctypes.data, pairs.ctypes.data)
    return n, pairs[:n].copy(), matches


WINDOW_POINT_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("ur", np.float32), ("radius", np.float32), ("level", np.int32),
                               ("search", np.int32)])


def search_keyframe_windows(keys, desc, bounds, points, point_desc, u_right=None, inv_level_sigma2=None, chi2_test=False):
    """The window search of ORBmatcher::Fuse / SearchBySim3 (reference src/ORBmatcher2.cc:522-579, 690-714; src/ORBmatcher3.cc:120-145) ->
    (best_idx[np], best_dist[np])."""
    k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    bi = np.full(len(p), -1, np.int32); bd = np.full(len(p), np.iinfo(np.int32).max, np.int32)
    lib().oracle_search_keyframe_windows(k.ctypes.data, d.ctypes.data, None if ur is None else ur.ctypes.data, len(k), bounds[0], bounds[1], bounds[2],
                                         bounds[3], None if sg is None else sg.ctypes.data, p.ctypes.data, pd.ctypes.data, len(p), int(bool(chi2_test)),
                                         bi.ctypes.data, bd.ctypes.data)
    return bi, bd


def search_by_sim3(keys1, kf_desc1, bounds1, keys2, kf_desc2, bounds2, points_1in2, mp_desc1, points_2in1, mp_desc2):
    """ORBmatcher::SearchBySim3 (reference src/ORBmatcher3.cc:37-254) -> (nFound, matches12[n1])."""
    k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
    d1 = np.ascontiguousarray(kf_desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(kf_desc2, np.uint8).reshape(-1, 32)
    p12 = np.ascontiguousarray(points_1in2, WINDOW_POINT_DTYPE); p21 = np.ascontiguousarray(points_2in1, WINDOW_POINT_DTYPE)
    m1 = np.ascontiguousarray(mp_desc1, np.uint8).reshape(-1, 32); m2 = np.ascontiguousarray(mp_desc2, np.uint8).reshape(-1, 32)
    b1 = np.ascontiguousarray(bounds1, np.float32); b2 = np.ascontiguousarray(bounds2, np.float32)
    out = np.full(len(k1), -1, np.int32)
    n = lib().oracle_search_by_sim3(k1.ctypes.data, d1.ctypes.data, len(k1), b1.ctypes.data, k2.ctypes.data, d2.ctypes.data, len(k2), b2.ctypes.data,
                                    p12.ctypes.data, m1.ctypes.data, p21.ctypes.data, m2.ctypes.data, out.ctypes.data)
    return n, out


def distinctive_descriptors(desc, start):
    """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:329-403), batched -> best[npoints]."""
    d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); st = np.ascontiguousarray(start, np.int32)
    best = np.full(len(st) - 1, -1, np.int32)
    lib().oracle_distinctive_descriptors(d.ctypes.data, st.ctypes.data, len(st) - 1, best.ctypes.data)
    return best


def search_by_projection_sim3(keys, desc, bounds, points, point_desc, matched, ratio_hamming=1.0):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) (reference src/ORBmatcher1.cc:429-533) -> (nmatches,
    new_match[n])."""
    k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
    mt = np.ascontiguousarray(matched, np.uint8)
    out = np.full(len(k), -1, np.int32)
    n = lib().oracle_search_by_projection_sim3(k.ctypes.data, d.ctypes.data, len(k), bounds[0], bounds[1], bounds[2], bounds[3], p.ctypes.data,
                                               pd.ctypes.data, len(p), mt.ctypes.data, float(ratio_hamming), out.ctypes.data)
    return n, out


def search_by_projection_keyframe(keys, desc, bounds, points, point_desc, point_angle, has_map_point, orb_dist=100, check_orientation=True):
    """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (reference src/ORBmatcher3.cc:469-590) -> (nmatches, new_match[n])."""
    k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
    pa = np.ascontiguousarray(point_angle, np.float32); hm = np.ascontiguousarray(has_map_point, np.uint8)
    out = np.full(len(k), -1, np.int32)
    n = lib().oracle_search_by_projection_keyframe(k.ctypes.data, d.ctypes.data, len(k), bounds[0], bounds[1], bounds[2], bounds[3], p.ctypes.data,
                                                   pd.ctypes.data, pa.ctypes.data, len(p), hm.ctypes.data, int(orb_dist), int(bool(check_orientation)),
                                                   out.ctypes.data)
    return n, out
