"""wut-cuda-orb-slam3_amd — MI355X-native ORB front-end (ctypes binding over the C ABI in include/orbx.h).

The product is ``liborbx.so`` (hand-written gfx950 HIP kernels + C ABI, sources in ``csrc/``).  This module is a
thin host-side mirror of the reference's operator interface for tests and benchmarks:

* :class:`ORBextractor` mirrors ``ORB_SLAM3::ORBextractor`` (reference include/ORBextractor.h:52-113):
  same constructor arguments, ``__call__(image, mask, lapping_area)`` returning
  ``(monoIndex, keypoints, descriptors)``, the ``Get*`` getters.
* :class:`ORBmatcher` mirrors the Hamming leafs of ``ORBmatcher`` / ``cv::BFMatcher``
  (reference src/ORBmatcher3.cc:637-653, src/Frame.cc:1174, 841-925).

There is no CPU fallback: importing works anywhere, but every compute call raises if ``liborbx.so`` is not
built or no HIP device exists.  Because the directory name contains hyphens, load this package with
``__graft_entry__.load_package()`` (importlib) — it registers itself as ``wut_cuda_orb_slam3_amd``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_ssize_t, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ORBX_LIB") or os.path.join(_HERE, "liborbx.so")  # ORBX_LIB: an experimental build of the same library

PROJECTION_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("invz", "<f4"), ("angle", "<f4"), ("octave", "<i4"), ("flags", "<i4")])  # orbx_projection
KEYPOINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")]
)
# orbx_window_point: a map point projected into a keyframe (ORBmatcher::Fuse / SearchBySim3)
WINDOW_POINT_DTYPE = np.dtype([("u", np.float32), ("v", np.float32), ("ur", np.float32), ("radius", np.float32), ("level", np.int32),
                               ("search", np.int32)])
assert KEYPOINT_DTYPE.itemsize == 28

TH_HIGH = 100  # reference src/ORBmatcher1.cc:37
TH_LOW = 50  # reference src/ORBmatcher1.cc:38
HISTO_LENGTH = 30  # reference src/ORBmatcher1.cc:39

ORBX_BLUR_OPENCV_450 = 0
ORBX_BLUR_OPENCV_451 = 1


class OrbxError(RuntimeError):
    pass


class _Params(ctypes.Structure):
    _fields_ = [
        ("nfeatures", c_int32), ("scale_factor", c_float), ("nlevels", c_int32), ("ini_th_fast", c_int32),
        ("min_th_fast", c_int32), ("device", c_int32), ("max_width", c_int32), ("max_height", c_int32),
        ("max_batch", c_int32), ("blur_variant", c_int32), ("trig_mode", c_int32),
    ]


# every symbol include/orbx.h declares (checked by tests/test_abi.py)
WORLD_POINT_DTYPE = np.dtype([("pos", "<f4", 3), ("normal", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"), ("angle", "<f4"),
                              ("octave", "<i4"), ("flags", "<i4")])  # orbx_world_point (44 bytes)


class _Camera(ctypes.Structure):  # orbx_camera
    _fields_ = [("type", c_int), ("p", c_float * 8)]


def _camera(cam_type: int, params) -> "_Camera":
    c = _Camera()
    c.type = int(cam_type)
    for i, v in enumerate(list(params)[:8]):
        c.p[i] = float(v)
    return c


ABI_SYMBOLS = [
    "orbx_default_params", "orbx_create", "orbx_destroy", "orbx_last_error", "orbx_get_levels", "orbx_get_scale_factor",
    "orbx_get_scale_factors", "orbx_get_inverse_scale_factors", "orbx_get_scale_sigma_squares",
    "orbx_get_inverse_scale_sigma_squares", "orbx_get_features_per_level", "orbx_get_umax", "orbx_max_keypoints",
    "orbx_extract", "orbx_extract_batch", "orbx_extract_batch_submit", "orbx_extract_batch_wait", "orbx_extract_batch_sharded", "orbx_host_alloc", "orbx_host_free", "orbx_host_register", "orbx_host_unregister", "orbx_extract_batch_device", "orbx_sync", "orbx_device_outputs",
    "orbx_download_outputs", "orbx_stream", "orbx_get_level_size", "orbx_get_pyramid_level", "orbx_debug_get_candidates",
    "orbx_debug_get_selected", "orbx_matcher_create", "orbx_matcher_destroy", "orbx_hamming_matrix", "orbx_match_knn2",
    "orbx_match_knn2_device", "orbx_match_knn2_batch_device", "orbx_match_knn2_frames", "orbx_compute_stereo_matches_frames",
    "orbx_matcher_download", "orbx_compute_stereo_fisheye_matches", "orbx_compute_stereo_fisheye_matches_frames", "orbx_search_by_projection_frames", "orbx_search_by_projection_points_frames", "orbx_search_for_initialization_frames", "orbx_matcher_sync", "orbx_match_stereo_rows", "orbx_device_count", "orbx_synth_frame",
    "orbx_level_sizes", "orbx_set_profiling", "orbx_get_stage_times", "orbx_compute_stereo_matches",
    "orbx_assign_features_to_grid", "orbx_search_for_initialization", "orbx_search_by_projection",
    "orbx_search_by_projection_points", "orbx_search_by_projection_fisheye", "orbx_search_by_projection_points_fisheye",
    "orbx_search_by_bow",
    "orbx_search_by_bow_keyframes",
    "orbx_search_for_triangulation",
    "orbx_search_keyframe_windows",
    "orbx_fuse",
    "orbx_search_by_sim3",
    "orbx_distinctive_descriptors",
    "orbx_search_by_projection_sim3",
    "orbx_search_by_projection_keyframe",
    "orbx_search_by_projection_fisheye_frames",
    "orbx_search_by_projection_points_fisheye_frames",
    "orbx_undistort_keypoints",
    "orbx_compute_image_bounds",
    "orbx_project_last_frame_points",
    "orbx_project_in_frustum",
    "orbx_project_points_frames",
    "orbx_project_in_frustum_fisheye",
    "orbx_project_in_frustum_fisheye_frames",
    "orbx_project_window_points",
    "orbx_project_window_points_frames",
    "orbx_search_keyframe_windows_frames",
    "orbx_fuse_frames",
    "orbx_search_for_triangulation_frames",
    "orbx_search_by_bow_frames",
    "orbx_known_switches",
    "orbx_vocabulary_create",
    "orbx_vocabulary_load_text",
    "orbx_vocabulary_destroy",
    "orbx_vocabulary_size",
    "orbx_vocabulary_transform",
    "orbx_vocabulary_transform_frame",
    "orbx_vocabulary_transform_batch",
]
STAGE_NAMES = ["pyramid+FAST", "quadtree", "blur", "pack", "orient+desc"]

_lib = None


def load_library() -> ctypes.CDLL:
    """dlopen liborbx.so; raises OrbxError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbxError(f"{LIB_PATH} is missing — run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64 (SONAME libamdhip64.so.7, the
    # same as /opt/rocm's).  If liborbx.so pulled in the system copy first, a later `import torch` would bring up a
    # second runtime that cannot see the GPU.  Importing torch first makes liborbx.so bind to the copy torch uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.orbx_last_error.restype = c_char_p
    lib.orbx_create.argtypes = [POINTER(_Params), POINTER(c_void_p)]
    lib.orbx_destroy.argtypes = [c_void_p]
    lib.orbx_destroy.restype = None
    lib.orbx_default_params.argtypes = [POINTER(_Params)]
    lib.orbx_default_params.restype = None
    lib.orbx_get_levels.argtypes = [c_void_p]
    lib.orbx_get_scale_factor.argtypes = [c_void_p]
    lib.orbx_get_scale_factor.restype = c_float
    for name in ("orbx_get_scale_factors", "orbx_get_inverse_scale_factors", "orbx_get_scale_sigma_squares",
                 "orbx_get_inverse_scale_sigma_squares", "orbx_get_features_per_level", "orbx_get_umax"):
        getattr(lib, name).argtypes = [c_void_p, c_void_p]
    lib.orbx_max_keypoints.argtypes = [c_void_p]
    lib.orbx_extract.argtypes = [c_void_p, c_void_p, c_int, c_int, c_ssize_t, c_int, c_int, c_void_p, c_void_p, c_int, POINTER(c_int)]
    lib.orbx_extract_batch.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_ssize_t, c_ssize_t, c_int, c_int, c_void_p,
                                       c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_extract_batch_submit.argtypes = lib.orbx_extract_batch.argtypes
    lib.orbx_extract_batch_wait.argtypes = [c_void_p]
    lib.orbx_extract_batch_sharded.argtypes = [POINTER(c_void_p), c_int, c_int, c_void_p, c_int, c_int, c_int, c_ssize_t, c_ssize_t, c_int, c_int,
                                               c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_host_alloc.argtypes = [c_size_t]
    lib.orbx_host_alloc.restype = c_void_p
    lib.orbx_host_free.argtypes = [c_void_p]
    lib.orbx_host_free.restype = None
    lib.orbx_host_register.argtypes = [c_void_p, c_size_t]
    lib.orbx_host_register.restype = c_int
    lib.orbx_host_unregister.argtypes = [c_void_p]
    lib.orbx_host_unregister.restype = c_int
    lib.orbx_extract_batch_device.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_ssize_t, c_ssize_t, c_int, c_int]
    lib.orbx_sync.argtypes = [c_void_p]
    lib.orbx_device_outputs.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]
    lib.orbx_download_outputs.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_stream.argtypes = [c_void_p]
    lib.orbx_stream.restype = c_void_p
    lib.orbx_get_level_size.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]
    lib.orbx_get_pyramid_level.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p]
    lib.orbx_debug_get_candidates.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int]
    lib.orbx_debug_get_selected.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int]
    lib.orbx_matcher_create.argtypes = [c_int, c_int, c_int, POINTER(c_void_p)]
    lib.orbx_matcher_destroy.argtypes = [c_void_p]
    lib.orbx_matcher_destroy.restype = None
    lib.orbx_matcher_sync.argtypes = [c_void_p]
    lib.orbx_hamming_matrix.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]
    lib.orbx_match_knn2.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_match_knn2_device.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_match_knn2_batch_device.argtypes = [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p,
                                                 c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_match_knn2_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_compute_stereo_matches_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p]
    lib.orbx_matcher_download.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    lib.orbx_search_by_projection_points_frames.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_float,
                                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p]
    lib.orbx_search_for_initialization_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float,
                                                          c_float, c_float, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p]
    lib.orbx_search_by_projection_frames.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_float,
                                                     c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int,
                                                     c_void_p, c_void_p, c_void_p]
    lib.orbx_compute_stereo_fisheye_matches.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                                        c_float, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.orbx_compute_stereo_fisheye_matches_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.orbx_match_stereo_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                           c_int, c_float, c_float, c_void_p, c_void_p]
    lib.orbx_synth_frame.argtypes = [c_uint64, c_int, c_int, c_void_p, c_ssize_t]
    lib.orbx_synth_frame.restype = None
    lib.orbx_level_sizes.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_compute_stereo_matches.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                                c_float, c_float, c_void_p, c_void_p]
    lib.orbx_assign_features_to_grid.argtypes = [c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p]
    lib.orbx_search_for_initialization.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_float,
                                                   c_float, c_void_p, c_void_p, c_int, c_float, c_int]
    lib.orbx_search_by_projection_fisheye.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                                      c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p]
    lib.orbx_search_by_projection_points_fisheye.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_float, c_float, c_float,
                                                             c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
                                                             c_float, c_void_p]
    lib.orbx_search_by_projection.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float,
                                              c_void_p, c_int, c_float, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p]
    lib.orbx_search_by_projection_points.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float,
                                                     c_float, c_void_p, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p]
    lib.orbx_search_by_bow.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]
    lib.orbx_search_by_bow_keyframes.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                 c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_void_p]
    lib.orbx_search_for_triangulation.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_search_keyframe_windows.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_int,
                                                 c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_fuse.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_int, c_void_p, c_void_p,
                              c_int, c_int, c_void_p]
    lib.orbx_search_by_sim3.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p]
    lib.orbx_distinctive_descriptors.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p]
    lib.orbx_search_by_projection_sim3.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                                                   c_int, c_void_p, c_float, c_void_p]
    lib.orbx_search_by_projection_keyframe.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p,
                                                       c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]
    lib.orbx_search_by_projection_fisheye_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p, c_void_p,
                                                             c_void_p, c_void_p, c_void_p, c_int, c_float, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_search_by_projection_points_fisheye_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_float, c_float, c_float, c_float, c_void_p,
                                                                    c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float, c_float,
                                                                    c_void_p, c_void_p]
    lib.orbx_undistort_keypoints.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_compute_image_bounds.argtypes = [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    for _name in ("orbx_search_keyframe_windows_frames", "orbx_fuse_frames"):
        getattr(lib, _name).argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_float, c_float, c_float, c_float, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_search_by_bow_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_void_p, c_void_p]
    lib.orbx_search_by_bow_frames.restype = c_int
    lib.orbx_search_for_triangulation_frames.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                                         c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.orbx_project_last_frame_points.argtypes = [c_void_p, c_void_p, c_void_p, POINTER(_Camera), c_void_p, c_int, c_void_p]
    lib.orbx_project_in_frustum.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(_Camera), c_void_p, c_float, c_float, c_float, c_int,
                                            c_int, c_float, c_void_p, c_int, c_void_p]
    lib.orbx_project_points_frames.argtypes = [c_void_p, c_int, c_int, c_void_p, POINTER(_Camera), c_void_p, c_float, c_float, c_float, c_int, c_int,
                                               c_float, c_void_p, c_void_p, c_int, c_void_p]
    lib.orbx_project_in_frustum_fisheye.argtypes = [c_void_p] * 11 + [c_float, c_float, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_project_in_frustum_fisheye_frames.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int, c_int, c_float,
                                                           c_void_p, c_void_p, c_int, c_void_p, c_void_p]
    lib.orbx_project_window_points.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(_Camera), c_void_p, c_float, c_float, c_void_p, c_float,
                                               c_int, c_int, c_void_p, c_int, c_void_p]
    lib.orbx_project_window_points_frames.argtypes = [c_void_p, c_int, c_void_p, POINTER(_Camera), c_void_p, c_float, c_float, c_void_p, c_float, c_int,
                                                      c_int, c_void_p, c_void_p, c_int, c_void_p]
    lib.orbx_vocabulary_create.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p)]
    lib.orbx_vocabulary_load_text.argtypes = [c_int, ctypes.c_char_p, ctypes.POINTER(c_void_p)]
    lib.orbx_vocabulary_destroy.argtypes = [c_void_p]
    lib.orbx_vocabulary_destroy.restype = None
    lib.orbx_vocabulary_size.argtypes = [c_void_p]
    lib.orbx_vocabulary_transform.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 9
    lib.orbx_vocabulary_transform_frame.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 9
    lib.orbx_vocabulary_transform_batch.argtypes = [c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 7
    lib.orbx_set_profiling.argtypes = [c_void_p, c_int]
    lib.orbx_get_stage_times.argtypes = [c_void_p, c_void_p]
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load_library().orbx_last_error()
        raise OrbxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def synth_frame(frame_index: int, w: int = 640, h: int = 480) -> np.ndarray:
    """Deterministic synthetic mono frame (include/orbx_synth.h)."""
    img = np.empty((h, w), np.uint8)
    load_library().orbx_synth_frame(frame_index, w, h, img.ctypes.data, w)
    return img


def device_count() -> int:
    return int(load_library().orbx_device_count())


class KeyFrameViewC(ctypes.Structure):  # orbx_keyframe_view
    _fields_ = [("keys", c_void_p), ("desc", c_void_p), ("n", c_int), ("n_left", c_int), ("u_right", c_void_p), ("has_map_point", c_void_p),
                ("fv_node", c_void_p), ("fv_start", c_void_p), ("fv_feature", c_void_p), ("nfv", c_int), ("scale_factors", c_void_p),
                ("level_sigma2", c_void_p), ("nlevels", c_int)]


class TriangulationGeometryC(ctypes.Structure):  # orbx_triangulation_geometry
    _fields_ = [("ep", c_float * 2), ("model", c_int), ("rig", c_int), ("F12", c_float * 9), ("cam_left", c_float * 8), ("cam_right", c_float * 8),
                ("precision", c_float), ("R", (c_float * 9) * 4), ("t", (c_float * 3) * 4)]


def keyframe_view(keys, desc, has_map_point, fv, scale_factors, level_sigma2, u_right=None, n_left=-1):
    """(orbx_keyframe_view, keep-alive list): keys = mvKeysUn (n_left == -1) or mvKeys ++ mvKeysRight; fv = the FeatureVector dict of
    ORBVocabulary.transform; the arrays must stay alive while the view is used (the second value holds them)."""
    k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    h = np.ascontiguousarray(has_map_point, np.uint8)
    assert len(k) == len(d) == len(h)
    f = [np.ascontiguousarray(fv[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
    sf = np.ascontiguousarray(scale_factors, np.float32); ls = np.ascontiguousarray(level_sigma2, np.float32)
    assert len(sf) == len(ls)
    ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    v = KeyFrameViewC(k.ctypes.data, d.ctypes.data, len(k), int(n_left), None if ur is None else ur.ctypes.data, h.ctypes.data, f[0].ctypes.data,
                      f[1].ctypes.data, f[2].ctypes.data, len(f[0]), sf.ctypes.data, ls.ctypes.data, len(sf))
    return v, [k, d, h, f, sf, ls, ur]


def triangulation_geometry(ep, F12=None, model=0, rig=False, cam_left=None, cam_right=None, precision=1e-6, R=None, t=None):
    """orbx_triangulation_geometry: ep (2,), F12 (3, 3) for the pinhole model; KannalaBrandt8: camera parameter vectors (8,), R [4][3][3]
    or [3][3], t [4][3] or [3]."""
    g = TriangulationGeometryC()
    g.ep[0], g.ep[1] = float(ep[0]), float(ep[1])
    g.model, g.rig, g.precision = int(model), int(bool(rig)), float(precision)
    if F12 is not None:
        for i, v in enumerate(np.asarray(F12, np.float32).reshape(9)):
            g.F12[i] = float(v)
    for name, arr in (("cam_left", cam_left), ("cam_right", cam_right if cam_right is not None else cam_left)):
        if arr is not None:
            for i, v in enumerate(np.asarray(arr, np.float32).reshape(8)):
                getattr(g, name)[i] = float(v)
    if R is not None:
        Rs = np.asarray(R, np.float32).reshape(-1, 9); ts = np.asarray(t, np.float32).reshape(-1, 3)
        for k in range(4):
            for i in range(9):
                g.R[k][i] = float(Rs[min(k, len(Rs) - 1)][i])
            for i in range(3):
                g.t[k][i] = float(ts[min(k, len(ts) - 1)][i])
    return g


class _HostBlock:
    """Owner of one orbx_host_alloc block; numpy arrays made by host_array() keep it alive through their .base chain."""

    def __init__(self, nbytes: int):
        self._lib = load_library()
        self.ptr = self._lib.orbx_host_alloc(max(int(nbytes), 1))
        if not self.ptr:
            raise OrbxError("orbx_host_alloc: " + self._lib.orbx_last_error().decode())
        self.buf = (ctypes.c_uint8 * max(int(nbytes), 1)).from_address(self.ptr)
        self.buf._orbx_owner = self  # the ctypes array (numpy's base) holds the owner

    def __del__(self):
        try:
            self._lib.orbx_host_free(self.ptr)
        except Exception:
            pass


def host_array(shape, dtype) -> np.ndarray:
    """A numpy array in page-locked host memory (orbx_host_alloc): frames read from it cross the link asynchronously, results are
    written into it by the device."""
    dt = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
    blk = _HostBlock(nbytes)
    return np.frombuffer(blk.buf, dtype=np.uint8, count=nbytes).view(dt).reshape(shape)


class host_registered:
    """Context manager around orbx_host_register / orbx_host_unregister: page-locks an existing (C-contiguous) numpy array in place for
    the duration of the block, so that orbx_extract_batch reads frames from it asynchronously - what a caller does once for the buffers
    of a camera ring it already owns."""

    def __init__(self, array: np.ndarray):
        if not array.flags["C_CONTIGUOUS"]:
            raise ValueError("host_registered needs a C-contiguous array")
        self.array = array
        self._lib = load_library()

    def __enter__(self):
        _check(self._lib.orbx_host_register(self.array.ctypes.data, self.array.nbytes), "orbx_host_register")
        return self.array

    def __exit__(self, *exc):
        _check(self._lib.orbx_host_unregister(self.array.ctypes.data), "orbx_host_unregister")
        return False


def extract_batch_sharded(extractors, images: np.ndarray, vLappingArea=(0, 0), block: int = 1, out=None):
    """orbx_extract_batch_sharded: images [B, H, W] dealt over the extractors (normally one per GPU) in blocks of `block` frames, from
    one process, one host thread per handle -> (n[B], mono[B], kps[B, cap], desc[B, cap, 32]) in frame order."""
    images = np.ascontiguousarray(images, np.uint8)
    b, h, w = images.shape
    ex0 = extractors[0]
    n, mono, kps, desc = out if out is not None else ex0.output_arrays(b)
    hs = (c_void_p * len(extractors))(*[e._h for e in extractors])
    _check(ex0._lib.orbx_extract_batch_sharded(hs, len(extractors), int(block), images.ctypes.data, b, w, h, images.strides[1], images.strides[0],
                                               int(vLappingArea[0]), int(vLappingArea[1]), kps.ctypes.data, desc.ctypes.data, ex0.capacity,
                                               n.ctypes.data, mono.ctypes.data), "orbx_extract_batch_sharded")
    return n, mono, kps, desc


class ORBextractor:
    """Mirror of ``ORB_SLAM3::ORBextractor`` (reference include/ORBextractor.h:58-92)."""

    def __init__(self, nfeatures: int, scaleFactor: float, nlevels: int, iniThFAST: int, minThFAST: int, *, device: int = 0,
                 max_width: int = 640, max_height: int = 480, max_batch: int = 1, blur_variant: int = ORBX_BLUR_OPENCV_450,
                 trig_mode: int = 0):
        self._lib = load_library()
        p = _Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device, max_width, max_height, max_batch,
                    blur_variant, trig_mode)
        h = c_void_p()
        _check(self._lib.orbx_create(byref(p), byref(h)), "orbx_create")
        self._h = h
        self.device = device
        self.nlevels = nlevels
        self.max_batch = max_batch
        self.capacity = int(self._lib.orbx_max_keypoints(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orbx_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- getters (reference include/ORBextractor.h:70-90) --------------------------------------------------
    def GetLevels(self) -> int:
        return int(self._lib.orbx_get_levels(self._h))

    def GetScaleFactor(self) -> float:
        return float(self._lib.orbx_get_scale_factor(self._h))

    def _table(self, fn: str, dtype, n=None) -> np.ndarray:
        out = np.zeros(n or self.nlevels, dtype)
        getattr(self._lib, fn)(self._h, out.ctypes.data)
        return out

    def GetScaleFactors(self):
        return self._table("orbx_get_scale_factors", np.float32)

    def GetInverseScaleFactors(self):
        return self._table("orbx_get_inverse_scale_factors", np.float32)

    def GetScaleSigmaSquares(self):
        return self._table("orbx_get_scale_sigma_squares", np.float32)

    def GetInverseScaleSigmaSquares(self):
        return self._table("orbx_get_inverse_scale_sigma_squares", np.float32)

    def features_per_level(self):
        return self._table("orbx_get_features_per_level", np.int32)

    def umax(self):
        return self._table("orbx_get_umax", np.int32, 16)

    # -- operator() (reference src/ORBextractor.cc:1227-1307) --------------------------------------------
    def __call__(self, image, mask=None, vLappingArea=(0, 0)):
        """Returns (monoIndex, keypoints[n] (KEYPOINT_DTYPE), descriptors[n, 32]); monoIndex == -1 on an empty image."""
        if image is None or image.size == 0:
            return -1, np.zeros(0, KEYPOINT_DTYPE), np.zeros((0, 32), np.uint8)
        image = np.ascontiguousarray(image, np.uint8)
        assert image.ndim == 2
        h, w = image.shape
        kps = np.zeros(self.capacity, KEYPOINT_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = c_int()
        mono = self._lib.orbx_extract(self._h, image.ctypes.data, w, h, image.strides[0], int(vLappingArea[0]),
                                      int(vLappingArea[1]), kps.ctypes.data, desc.ctypes.data, self.capacity, byref(n))
        if mono <= -101:
            _check(mono + 100, "orbx_extract")
        return int(mono), kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, images: np.ndarray, vLappingArea=(0, 0), out=None):
        """images: [B, H, W] uint8 host array -> (n[B], mono[B], kps[B, cap], desc[B, cap, 32]).  out = (n, mono, kps, desc) of an earlier
        call (or output_arrays()) is reused instead of allocating 60 KB per frame anew - the reference's caller owns and reuses its
        output containers the same way; rows beyond n[f] are then whatever the arrays held."""
        images = np.ascontiguousarray(images, np.uint8)
        b, h, w = images.shape
        n, mono, kps, desc = out if out is not None else self.output_arrays(b)
        assert kps.shape == (b, self.capacity) and desc.shape == (b, self.capacity, 32) and len(n) == b and len(mono) == b
        _check(self._lib.orbx_extract_batch(self._h, images.ctypes.data, b, w, h, images.strides[1], images.strides[0],
                                            int(vLappingArea[0]), int(vLappingArea[1]), kps.ctypes.data, desc.ctypes.data,
                                            self.capacity, n.ctypes.data, mono.ctypes.data), "orbx_extract_batch")
        return n, mono, kps, desc

    def extract_batch_submit(self, images: np.ndarray, vLappingArea=(0, 0), out=None):
        """orbx_extract_batch_submit: queues the batch and returns at once; `images` (page-locked: host_array) and `out` must stay alive and
        untouched until the extract_batch_wait() that returns the same (n, mono, kps, desc).  Two batches may be pending on a handle."""
        assert images.dtype == np.uint8 and images.flags.c_contiguous and images.ndim == 3
        b, h, w = images.shape
        n, mono, kps, desc = out if out is not None else self.output_arrays(b)
        assert kps.shape == (b, self.capacity) and desc.shape == (b, self.capacity, 32) and len(n) == b and len(mono) == b
        _check(self._lib.orbx_extract_batch_submit(self._h, images.ctypes.data, b, w, h, images.strides[1], images.strides[0],
                                                   int(vLappingArea[0]), int(vLappingArea[1]), kps.ctypes.data, desc.ctypes.data,
                                                   self.capacity, n.ctypes.data, mono.ctypes.data), "orbx_extract_batch_submit")
        if not hasattr(self, "_submitted"):
            self._submitted = []
        self._submitted.append((images, (n, mono, kps, desc)))   # (keeps the arrays alive; up to two batches may be pending)

    def extract_batch_wait(self):
        """orbx_extract_batch_wait: blocks until the OLDEST submitted batch's results are in the arrays given to its extract_batch_submit."""
        _check(self._lib.orbx_extract_batch_wait(self._h), "orbx_extract_batch_wait")
        _, out = self._submitted.pop(0)
        return out

    def output_arrays(self, b: int, page_locked: bool = False):
        """(n[b], mono[b], kps[b, cap], desc[b, cap, 32]) zero-filled host arrays; page_locked: in orbx_host_alloc memory, which the
        device writes directly (no staging copy)."""
        mk = host_array if page_locked else (lambda shape, dtype: np.zeros(shape, dtype))
        out = mk((b,), np.int32), mk((b,), np.int32), mk((b, self.capacity), KEYPOINT_DTYPE), mk((b, self.capacity, 32), np.uint8)
        for a in out:
            a[...] = 0
        return out

    def extract_batch_device(self, d_ptr: int, nframes: int, w: int, h: int, stride: int, frame_stride: int, vLappingArea=(0, 0)):
        """Asynchronous device-resident batch (the path bench.py times); d_ptr is a raw device address."""
        _check(self._lib.orbx_extract_batch_device(self._h, c_void_p(d_ptr), nframes, w, h, stride, frame_stride,
                                                   int(vLappingArea[0]), int(vLappingArea[1])), "orbx_extract_batch_device")

    def sync(self) -> None:
        _check(self._lib.orbx_sync(self._h), "orbx_sync")

    def stream(self) -> int:
        return int(self._lib.orbx_stream(self._h) or 0)

    def set_profiling(self, enable: bool) -> None:
        _check(self._lib.orbx_set_profiling(self._h, int(enable)), "orbx_set_profiling")

    def stage_times(self):
        """(mean ms per batch for each of STAGE_NAMES, batches averaged) since the previous call."""
        ms = np.zeros(len(STAGE_NAMES), np.float32)
        nb = self._lib.orbx_get_stage_times(self._h, ms.ctypes.data)
        if nb < 0:
            _check(nb, "orbx_get_stage_times")
        return ms, nb

    def device_outputs(self):
        k, d, n, m = c_void_p(), c_void_p(), c_void_p(), c_void_p()
        _check(self._lib.orbx_device_outputs(self._h, byref(k), byref(d), byref(n), byref(m)), "orbx_device_outputs")
        return k.value, d.value, n.value, m.value

    def download_outputs(self, nframes: int):
        kps = np.zeros((nframes, self.capacity), KEYPOINT_DTYPE)
        desc = np.zeros((nframes, self.capacity, 32), np.uint8)
        n = np.zeros(nframes, np.int32)
        mono = np.zeros(nframes, np.int32)
        _check(self._lib.orbx_download_outputs(self._h, nframes, kps.ctypes.data, desc.ctypes.data, self.capacity,
                                               n.ctypes.data, mono.ctypes.data), "orbx_download_outputs")
        return n, mono, kps, desc

    # -- mvImagePyramid + stage taps -------------------------------------------------------------------
    def level_size(self, level: int):
        w, h = c_int(), c_int()
        _check(self._lib.orbx_get_level_size(self._h, level, byref(w), byref(h)), "orbx_get_level_size")
        return w.value, h.value

    def pyramid_level(self, frame: int, level: int, blurred: bool = False) -> np.ndarray:
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self._lib.orbx_get_pyramid_level(self._h, frame, level, int(blurred), out.ctypes.data), "orbx_get_pyramid_level")
        return out

    def debug_candidates(self, frame: int, level: int, cap: int = 1 << 20) -> np.ndarray:
        out = np.zeros(cap, np.uint32)
        n = self._lib.orbx_debug_get_candidates(self._h, frame, level, out.ctypes.data, cap)
        if n < 0:
            _check(n, "orbx_debug_get_candidates")
        return out[:n].copy()

    def debug_selected(self, frame: int, level: int, cap: int = 1 << 16) -> np.ndarray:
        out = np.zeros(cap, np.uint32)
        n = self._lib.orbx_debug_get_selected(self._h, frame, level, out.ctypes.data, cap)
        if n < 0:
            _check(n, "orbx_debug_get_selected")
        return out[:n].copy()


class ORBmatcher:
    """Hamming leafs of the reference matcher on the GPU (include/orbx.h matcher section)."""

    TH_HIGH = TH_HIGH
    TH_LOW = TH_LOW
    HISTO_LENGTH = HISTO_LENGTH

    def __init__(self, max_query: int = 8192, max_train: int = 8192, device: int = 0):
        self._lib = load_library()
        h = c_void_p()
        _check(self._lib.orbx_matcher_create(device, max_query, max_train, byref(h)), "orbx_matcher_create")
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orbx_matcher_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def DescriptorDistance(a: np.ndarray, b: np.ndarray) -> int:
        """ORBmatcher::DescriptorDistance (reference src/ORBmatcher3.cc:637-653) — host leaf."""
        return int(np.unpackbits(np.bitwise_xor(np.asarray(a, np.uint8), np.asarray(b, np.uint8))).sum())

    def hamming_matrix(self, q: np.ndarray, t: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(t)), np.int32)
        _check(self._lib.orbx_hamming_matrix(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), out.ctypes.data), "orbx_hamming_matrix")
        return out

    def knnMatch(self, q: np.ndarray, t: np.ndarray):
        """cv::BFMatcher(NORM_HAMMING).knnMatch(q, t, k=2) -> (idx[nq, 2], dist[nq, 2])."""
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx = np.full((len(q), 2), -1, np.int32)
        dist = np.full((len(q), 2), np.iinfo(np.int32).max, np.int32)
        _check(self._lib.orbx_match_knn2(self._h, q.ctypes.data, len(q), t.ctypes.data, len(t), idx.ctypes.data, dist.ctypes.data), "orbx_match_knn2")
        return idx, dist

    def sync(self) -> None:
        _check(self._lib.orbx_matcher_sync(self._h), "orbx_matcher_sync")
        self._inflight = []  # the matcher's stream has drained: nothing reads the stashed inputs any more

    # ---- device-resident, batched forms: pair p = frame p of the two extractors' last batches -----------------------
    # Stream contract of the *Frames methods.  The kernels run on the MATCHER's stream, the tensors are allocated by torch:
    #  * outputs are filled with their "no match" value on torch's stream and the device is synchronised BEFORE the call, so rows beyond a
    #    frame's keypoint count (which the kernels do not write) are defined, and nothing torch queued races the matcher's kernels;
    #  * with download=False the call returns while the kernels still run.  The input tensors (projections, descriptors, counts ...) are
    #    kept alive in self._inflight until the next sync() - released to torch's caching allocator earlier, their blocks could be handed
    #    out and overwritten mid-search - and the returned tensors must not be touched by any torch operation before m.sync().
    def _dev_out(self, shape, dtype, device: int, fill):
        import torch  # device memory only
        return torch.full(shape, fill, dtype=dtype, device=f"cuda:{device}")

    def _launch_fence(self, *inputs) -> None:
        import torch
        # every device the tensors live on (torch.cuda.synchronize() alone waits for torch's CURRENT device only: with the matcher on
        # another GPU the fills / uploads above could land after its kernels)
        devs = {t.device for t in inputs if t is not None and getattr(t, "is_cuda", False)}
        for d in devs or {None}:
            torch.cuda.synchronize(d)
        if not hasattr(self, "_inflight"):
            self._inflight = []
        self._inflight.extend(t for t in inputs if t is not None)

    def _download(self, tensor) -> np.ndarray:
        out = np.empty(tuple(tensor.shape), {4: np.int32}.get(tensor.element_size(), np.int32) if not tensor.is_floating_point() else np.float32)
        _check(self._lib.orbx_matcher_download(self._h, c_void_p(tensor.data_ptr()), out.ctypes.data, out.nbytes), "orbx_matcher_download")
        self._inflight = []  # the download waited for the matcher's stream
        return out

    def knnMatchFrames(self, query: "ORBextractor", train: "ORBextractor", npairs: int, lapping_only: bool = False, download: bool = True):
        """knnMatch(k=2) of `npairs` frame pairs in one launch on the extractors' device-resident descriptors
        (Frame::ComputeStereoFishEyeMatches' matcher call, reference src/Frame.cc:1174; lapping_only = the rowRange(monoIndex, rows)
        form).  Returns (idx, dist) as [npairs, query.capacity, 2] - numpy arrays, or torch device tensors with download=False."""
        import torch
        dev = query.device
        idx = self._dev_out((npairs, query.capacity, 2), torch.int32, dev, -1)
        dist = self._dev_out((npairs, query.capacity, 2), torch.int32, dev, np.iinfo(np.int32).max)
        self._launch_fence(idx, dist)
        _check(self._lib.orbx_match_knn2_frames(self._h, query._h, train._h, npairs, int(lapping_only), c_void_p(idx.data_ptr()),
                                                c_void_p(dist.data_ptr())), "orbx_match_knn2_frames")
        if not download:
            return idx, dist
        return self._download(idx), self._download(dist)

    def ComputeStereoMatchesFrames(self, left: "ORBextractor", right: "ORBextractor", npairs: int, mb: float, mbf: float, download: bool = True):
        """Frame::ComputeStereoMatches (reference src/Frame.cc:841-1011) for `npairs` stereo pairs in three launches, reading the
        extractors' keypoints, descriptors and pyramids where they lie in HBM -> (mvuRight, mvDepth) as [npairs, left.capacity]."""
        import torch
        dev = left.device
        ur = self._dev_out((npairs, left.capacity), torch.float32, dev, -1.0)
        dp = self._dev_out((npairs, left.capacity), torch.float32, dev, -1.0)
        self._launch_fence(ur, dp)
        _check(self._lib.orbx_compute_stereo_matches_frames(self._h, left._h, right._h, npairs, mb, mbf, c_void_p(ur.data_ptr()),
                                                            c_void_p(dp.data_ptr())), "orbx_compute_stereo_matches_frames")
        if not download:
            return ur, dp
        return self._download(ur), self._download(dp)

    def ComputeStereoFishEyeMatches(self, kl, dl, mono_left, kr, dr, mono_right, cam1, cam2, R12, t12, level_sigma2, precision: float = 1e-6):
        """Frame::ComputeStereoFishEyeMatches (reference src/Frame.cc:1156-1196) ->
        (nMatches, mvLeftToRightMatch[nl], mvRightToLeftMatch[nr], mvDepth[nl], mvStereo3Dpoints[nl, 3])."""
        kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE); kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32); dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        c1 = np.ascontiguousarray(cam1, np.float32); c2 = np.ascontiguousarray(cam2, np.float32)
        R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32)
        ls = np.ascontiguousarray(level_sigma2, np.float32)
        l2r = np.full(len(kl), -1, np.int32); r2l = np.full(len(kr), -1, np.int32)
        depth = np.full(len(kl), -1, np.float32); p3d = np.zeros((len(kl), 3), np.float32)
        n = self._lib.orbx_compute_stereo_fisheye_matches(self._h, kl.ctypes.data, dl.ctypes.data, len(kl), int(mono_left), kr.ctypes.data, dr.ctypes.data,
                                                          len(kr), int(mono_right), c1.ctypes.data, c2.ctypes.data, precision, R.ctypes.data, t.ctypes.data,
                                                          ls.ctypes.data, len(ls), l2r.ctypes.data, r2l.ctypes.data, depth.ctypes.data, p3d.ctypes.data)
        if n < 0:
            _check(n, "orbx_compute_stereo_fisheye_matches")
        return int(n), l2r, r2l, depth, p3d

    def ComputeStereoFishEyeMatchesFrames(self, left: "ORBextractor", right: "ORBextractor", npairs: int, cam1, cam2, R12, t12, precision: float = 1e-6):
        """The batch form on the extractors' device-resident outputs -> (nMatches[P], left_to_right[P, capL], right_to_left[P, capR],
        depth[P, capL], p3d[P, capL, 3]) as numpy arrays."""
        import torch
        dev = left.device
        c1 = np.ascontiguousarray(cam1, np.float32); c2 = np.ascontiguousarray(cam2, np.float32)
        R = np.ascontiguousarray(R12, np.float32).reshape(9); t = np.ascontiguousarray(t12, np.float32)
        l2r = self._dev_out((npairs, left.capacity), torch.int32, dev, -1); r2l = self._dev_out((npairs, right.capacity), torch.int32, dev, -1)
        depth = self._dev_out((npairs, left.capacity), torch.float32, dev, -1.0); p3d = self._dev_out((npairs, left.capacity, 3), torch.float32, dev, 0.0)
        nm = self._dev_out((npairs,), torch.int32, dev, 0)
        self._launch_fence(l2r, r2l, depth, p3d, nm)
        _check(self._lib.orbx_compute_stereo_fisheye_matches_frames(self._h, left._h, right._h, npairs, c1.ctypes.data, c2.ctypes.data, precision,
                                                                    R.ctypes.data, t.ctypes.data, c_void_p(l2r.data_ptr()), c_void_p(r2l.data_ptr()),
                                                                    c_void_p(depth.data_ptr()), c_void_p(p3d.data_ptr()), c_void_p(nm.data_ptr())),
               "orbx_compute_stereo_fisheye_matches_frames")
        return self._download(nm), self._download(l2r), self._download(r2l), self._download(depth), self._download(p3d)

    def FuseFrames(self, ex: "ORBextractor", nframes: int, k4, dist, bounds, points, point_desc, chi2_test: bool = True, u_right=None,
                   fuse: bool = True):
        """ORBmatcher::Fuse's search (fuse=True: fused_idx + nFused) or the plain KeyFrame window search (fuse=False: best_idx + best_dist) for the
        first `nframes` frames of the extractor's last batch taken as keyframes, device-resident (orbx_fuse_frames /
        orbx_search_keyframe_windows_frames).  points[p] / point_desc[p]: WINDOW_POINT_DTYPE records / [n, 32] uint8 of keyframe p.
        Returns (fused_idx [P, max_np], nfused [P]) resp. (best_idx, best_dist).  points may also be the (d_points, d_np, max_np) that
        ProjectWindowPointsFrames returned: the records then never leave the device."""
        import torch
        dev = ex.device
        cap = ex.capacity
        dv = f"cuda:{dev}"
        on_device = isinstance(points, tuple) and len(points) == 3 and hasattr(points[0], "data_ptr")
        max_np = points[2] if on_device else max(1, max(len(p) for p in points))
        hd = np.zeros((nframes, max_np, 32), np.uint8)
        for p in range(nframes):
            pd = np.asarray(point_desc[p], np.uint8).reshape(-1, 32); hd[p, : len(pd)] = pd
        d_d = torch.from_numpy(hd).to(dv)
        if on_device:
            d_p, d_n = points[0], points[1]
        else:
            hp = np.zeros((nframes, max_np), WINDOW_POINT_DTYPE); hn = np.zeros(nframes, np.int32)
            for p in range(nframes):
                hn[p] = len(points[p]); hp[p, : hn[p]] = points[p]
            d_p = torch.from_numpy(hp.view(np.uint8).reshape(nframes, -1)).to(dv); d_n = torch.from_numpy(hn).to(dv)
        d_ur = torch.from_numpy(np.ascontiguousarray(u_right, np.float32)).to(dv) if u_right is not None else None
        o1 = self._dev_out((nframes, max_np), torch.int32, dev, -1)
        o2 = self._dev_out((nframes,) if fuse else (nframes, max_np), torch.int32, dev, 0 if fuse else np.iinfo(np.int32).max)
        k4 = np.ascontiguousarray(k4, np.float32); dist = np.ascontiguousarray(dist, np.float32).ravel()
        self._launch_fence(d_p, d_d, d_n, d_ur, o1, o2)
        fn = self._lib.orbx_fuse_frames if fuse else self._lib.orbx_search_keyframe_windows_frames
        _check(fn(self._h, ex._h, nframes, k4.ctypes.data, dist.ctypes.data if len(dist) else None, len(dist), None, bounds[0], bounds[1], bounds[2],
                  bounds[3], c_void_p(d_ur.data_ptr()) if d_ur is not None else None, c_void_p(d_p.data_ptr()), c_void_p(d_d.data_ptr()),
                  c_void_p(d_n.data_ptr()), max_np, int(bool(chi2_test)), c_void_p(o1.data_ptr()), c_void_p(o2.data_ptr())),
               "orbx_fuse_frames" if fuse else "orbx_search_keyframe_windows_frames")
        return self._download(o1), self._download(o2)

    # ---- projection head of the tracking searches (include/orbx.h "The projection head") ------------------------------------------
    def ProjectLastFramePoints(self, q_xyzw, t, cam_type: int, cam, points) -> np.ndarray:
        """x3Dc = Tcw * x3Dw, invzc, mpCamera->project(x3Dc) for the last frame's map points (reference src/ORBmatcher3.cc:283-293):
        Tcw as the Sophus::SE3f's unit quaternion (x, y, z, w) and translation; points: WORLD_POINT_DTYPE.  Returns PROJECTION_DTYPE records."""
        q = np.ascontiguousarray(q_xyzw, np.float32); tt = np.ascontiguousarray(t, np.float32)
        pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
        out = np.zeros(len(pts), PROJECTION_DTYPE)
        c = _camera(cam_type, cam)
        _check(self._lib.orbx_project_last_frame_points(self._h, q.ctypes.data, tt.ctypes.data, ctypes.byref(c), pts.ctypes.data, len(pts),
                                                        out.ctypes.data), "orbx_project_last_frame_points")
        return out

    def ProjectInFrustum(self, Rcw, tcw, Ow, cam_type: int, cam, bounds, mbf: float, viewing_cos_limit: float, log_scale_factor: float, nlevels: int,
                         points, far_points: bool = False, th_far_points: float = 0.0) -> np.ndarray:
        """Frame::isInFrustum over the local map points (reference src/Frame.cc:542-600) -> the records SearchByProjectionPoints takes."""
        R = np.ascontiguousarray(Rcw, np.float32).reshape(9); tt = np.ascontiguousarray(tcw, np.float32); ow = np.ascontiguousarray(Ow, np.float32)
        b = np.ascontiguousarray(bounds, np.float32)
        pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
        out = np.zeros(len(pts), PROJECTION_DTYPE)
        c = _camera(cam_type, cam)
        _check(self._lib.orbx_project_in_frustum(self._h, R.ctypes.data, tt.ctypes.data, ow.ctypes.data, ctypes.byref(c), b.ctypes.data, float(mbf),
                                                 float(viewing_cos_limit), float(log_scale_factor), int(nlevels), int(bool(far_points)),
                                                 float(th_far_points), pts.ctypes.data, len(pts), out.ctypes.data), "orbx_project_in_frustum")
        return out

    def ProjectPointsFrames(self, mode: int, poses, cam_type: int, cam, points, device: int, bounds=None, mbf: float = 0.0,
                            viewing_cos_limit: float = 0.5, log_scale_factor: float = 0.0, nlevels: int = 0, far_points: bool = False,
                            th_far_points: float = 0.0):
        """The projection head for a batch of frames, device-resident: poses [P, 16] (mode 0: q xyzw, t; mode 1: Rcw[9], tcw, Ow), points = list
        of P arrays of WORLD_POINT_DTYPE.  Returns (d_proj [P, max_np, 24] uint8 torch tensor on the device, d_np [P] int32 tensor, max_np):
        pass d_proj / d_np as `proj=` / `proj_counts=` of SearchByProjectionFrames - nothing of the projections crosses the link."""
        import torch
        P = len(points)
        max_np = max(1, max(len(p) for p in points))
        hw = np.zeros((P, max_np), WORLD_POINT_DTYPE); hn = np.zeros(P, np.int32)
        for f in range(P):
            hn[f] = len(points[f]); hw[f, : hn[f]] = points[f]
        hpose = np.zeros((P, 16), np.float32); hpose[:, : np.asarray(poses).shape[1]] = np.asarray(poses, np.float32)
        dv = f"cuda:{device}"
        d_w = torch.from_numpy(hw.view(np.uint8).reshape(P, -1)).to(dv); d_n = torch.from_numpy(hn).to(dv); d_pose = torch.from_numpy(hpose).to(dv)
        d_out = self._dev_out((P, max_np, PROJECTION_DTYPE.itemsize), torch.uint8, device, 0)
        b = np.ascontiguousarray(bounds if bounds is not None else (0, 0, 0, 0), np.float32)
        c = _camera(cam_type, cam)
        self._launch_fence(d_w, d_n, d_pose, d_out)
        _check(self._lib.orbx_project_points_frames(self._h, int(mode), P, c_void_p(d_pose.data_ptr()), ctypes.byref(c), b.ctypes.data, float(mbf),
                                                    float(viewing_cos_limit), float(log_scale_factor), int(nlevels), int(bool(far_points)),
                                                    float(th_far_points), c_void_p(d_w.data_ptr()), c_void_p(d_n.data_ptr()), max_np,
                                                    c_void_p(d_out.data_ptr())), "orbx_project_points_frames")
        return d_out, d_n, max_np

    def ProjectInFrustumFisheye(self, Rcw, tcw, Ow, Rrl, trl, tlr, Rwc, cam1, cam2, bounds, viewing_cos_limit: float, log_scale_factor: float,
                                nlevels: int, points, far_points: bool = False, th_far_points: float = 0.0):
        """orbx_project_in_frustum_fisheye: Frame::isInFrustum of a fisheye rig (reference src/Frame.cc:604-614, 1198-1272) -> (left, right)
        PROJECTION_DTYPE records."""
        f32 = lambda a, n: np.ascontiguousarray(a, np.float32).reshape(n)
        arrs = [f32(Rcw, 9), f32(tcw, 3), f32(Ow, 3), f32(Rrl, 9), f32(trl, 3), f32(tlr, 3), f32(Rwc, 9)]
        c1 = np.zeros(8, np.float32); c1[:len(cam1)] = cam1
        c2 = np.zeros(8, np.float32); c2[:len(cam2)] = cam2
        b = np.ascontiguousarray(bounds, np.float32)
        pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
        ol = np.zeros(len(pts), PROJECTION_DTYPE); orr = np.zeros(len(pts), PROJECTION_DTYPE)
        _check(self._lib.orbx_project_in_frustum_fisheye(self._h, *[a.ctypes.data for a in arrs], c1.ctypes.data, c2.ctypes.data, b.ctypes.data,
                                                         float(viewing_cos_limit), float(log_scale_factor), int(nlevels), int(bool(far_points)),
                                                         float(th_far_points), pts.ctypes.data, len(pts), ol.ctypes.data, orr.ctypes.data),
               "orbx_project_in_frustum_fisheye")
        return ol, orr

    def ProjectInFrustumFisheyeFrames(self, poses40, cam1, cam2, bounds, viewing_cos_limit: float, log_scale_factor: float, nlevels: int, points,
                                      device: int, far_points: bool = False, th_far_points: float = 0.0):
        """Batch form, device-resident: poses40 [P, 40] (Rcw[9] tcw[3] Ow[3] Rrl[9] trl[3] tlr[3] Rwc[9] + pad), points = list of P arrays.
        Returns (d_left, d_right [P, max_np, 24] uint8 device tensors, d_np, max_np)."""
        import torch
        P = len(points)
        max_np = max(1, max(len(p) for p in points))
        hw = np.zeros((P, max_np), WORLD_POINT_DTYPE); hn = np.zeros(P, np.int32)
        for f in range(P):
            hn[f] = len(points[f]); hw[f, : hn[f]] = points[f]
        hpose = np.zeros((P, 40), np.float32); hpose[:, : np.asarray(poses40).shape[1]] = np.asarray(poses40, np.float32)
        dv = f"cuda:{device}"
        d_w = torch.from_numpy(hw.view(np.uint8).reshape(P, -1)).to(dv); d_n = torch.from_numpy(hn).to(dv); d_pose = torch.from_numpy(hpose).to(dv)
        d_l = torch.zeros((P, max_np, PROJECTION_DTYPE.itemsize), dtype=torch.uint8, device=dv)
        d_r = torch.zeros((P, max_np, PROJECTION_DTYPE.itemsize), dtype=torch.uint8, device=dv)
        c1 = np.zeros(8, np.float32); c1[:len(cam1)] = cam1
        c2 = np.zeros(8, np.float32); c2[:len(cam2)] = cam2
        b = np.ascontiguousarray(bounds, np.float32)
        self._launch_fence(d_w, d_n, d_pose, d_l, d_r)
        _check(self._lib.orbx_project_in_frustum_fisheye_frames(self._h, P, c_void_p(d_pose.data_ptr()), c1.ctypes.data, c2.ctypes.data, b.ctypes.data,
                                                                float(viewing_cos_limit), float(log_scale_factor), int(nlevels), int(bool(far_points)),
                                                                float(th_far_points), c_void_p(d_w.data_ptr()), c_void_p(d_n.data_ptr()), max_np,
                                                                c_void_p(d_l.data_ptr()), c_void_p(d_r.data_ptr())), "orbx_project_in_frustum_fisheye_frames")
        return d_l, d_r, d_n, max_np

    def ProjectWindowPoints(self, q_xyzw, t, Ow, cam_type: int, cam, bounds, bf: float, th: float, scale_factors, log_scale_factor: float, tests: int,
                            points):
        """orbx_project_window_points: the head of Fuse / SearchByProjection(KeyFrame, Sim3) / SearchByProjection(Frame, KeyFrame) (reference
        src/ORBmatcher2.cc:446-520, src/ORBmatcher1.cc:451-494, src/ORBmatcher3.cc:486-519) -> WINDOW_POINT_DTYPE records."""
        q = np.ascontiguousarray(q_xyzw, np.float32); tt = np.ascontiguousarray(t, np.float32); ow = np.ascontiguousarray(Ow, np.float32)
        b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
        pts = np.ascontiguousarray(points, WORLD_POINT_DTYPE)
        out = np.zeros(len(pts), WINDOW_POINT_DTYPE)
        c = _camera(cam_type, cam)
        _check(self._lib.orbx_project_window_points(self._h, q.ctypes.data, tt.ctypes.data, ow.ctypes.data, ctypes.byref(c), b.ctypes.data, float(bf),
                                                    float(th), sf.ctypes.data, float(log_scale_factor), len(sf), int(tests), pts.ctypes.data, len(pts),
                                                    out.ctypes.data), "orbx_project_window_points")
        return out

    def ProjectWindowPointsFrames(self, poses, cam_type: int, cam, bounds, bf: float, th: float, scale_factors, log_scale_factor: float, tests: int,
                                  points, device: int):
        """Batch form, device-resident: poses [P, 10] (q xyzw, t, Ow), points = list of P arrays.  Returns (d_points [P, max_np, 24] uint8 device
        tensor of orbx_window_point records - what FuseFrames / SearchKeyframeWindowsFrames take -, d_np, max_np)."""
        import torch
        P = len(points)
        max_np = max(1, max(len(p) for p in points))
        hw = np.zeros((P, max_np), WORLD_POINT_DTYPE); hn = np.zeros(P, np.int32)
        for f in range(P):
            hn[f] = len(points[f]); hw[f, : hn[f]] = points[f]
        hpose = np.zeros((P, 16), np.float32); hpose[:, : np.asarray(poses).shape[1]] = np.asarray(poses, np.float32)
        dv = f"cuda:{device}"
        d_w = torch.from_numpy(hw.view(np.uint8).reshape(P, -1)).to(dv); d_n = torch.from_numpy(hn).to(dv); d_pose = torch.from_numpy(hpose).to(dv)
        d_out = torch.zeros((P, max_np, WINDOW_POINT_DTYPE.itemsize), dtype=torch.uint8, device=dv)
        b = np.ascontiguousarray(bounds, np.float32); sf = np.ascontiguousarray(scale_factors, np.float32)
        c = _camera(cam_type, cam)
        self._launch_fence(d_w, d_n, d_pose, d_out)
        _check(self._lib.orbx_project_window_points_frames(self._h, P, c_void_p(d_pose.data_ptr()), ctypes.byref(c), b.ctypes.data, float(bf), float(th),
                                                           sf.ctypes.data, float(log_scale_factor), len(sf), int(tests), c_void_p(d_w.data_ptr()),
                                                           c_void_p(d_n.data_ptr()), max_np, c_void_p(d_out.data_ptr())),
               "orbx_project_window_points_frames")
        return d_out, d_n, max_np

    def SearchByProjectionFrames(self, ex: "ORBextractor", nframes: int, k4, dist, bounds, mbf: float, proj, proj_desc, th: float,
                                 forward: bool = False, backward: bool = False, checkOrientation: bool = True, u_right=None, cur_state=None,
                                 download: bool = True, map_points_nnratio=None, proj_counts=None):
        """Batched, device-resident tracking matcher: UndistortKeyPoints -> AssignFeaturesToGrid -> SearchByProjection(Frame, Frame) for the
        first `nframes` frames of the extractor's last batch.  proj[p] / proj_desc[p]: the projections of frame p (lists of arrays of
        PROJECTION_DTYPE / [n, 32] uint8) - or, with proj_counts, the device tensors ProjectPointsFrames returned (d_proj, d_np): the
        projection head's records are consumed where they are.  Returns (nmatches[P], assign[P, cap], kps_un[P, cap])."""
        import torch
        dev = ex.device
        cap = ex.capacity
        if proj_counts is not None:
            d_p, d_n = proj, proj_counts
            max_np = int(d_p.shape[1])
            hd = np.zeros((nframes, max_np, 32), np.uint8)
            for p in range(nframes):
                dd = np.asarray(proj_desc[p], np.uint8).reshape(-1, 32); hd[p, : len(dd)] = dd
        else:
            max_np = max(1, max(len(p) for p in proj))
            hp = np.zeros((nframes, max_np), PROJECTION_DTYPE); hd = np.zeros((nframes, max_np, 32), np.uint8); hn = np.zeros(nframes, np.int32)
            for p in range(nframes):
                hn[p] = len(proj[p]); hp[p, : hn[p]] = proj[p]; hd[p, : hn[p]] = np.asarray(proj_desc[p], np.uint8).reshape(-1, 32)
            d_p = torch.from_numpy(hp.view(np.uint8)).to(f"cuda:{dev}"); d_n = torch.from_numpy(hn).to(f"cuda:{dev}")
        d_d = torch.from_numpy(hd).to(f"cuda:{dev}")
        d_ur = torch.from_numpy(np.ascontiguousarray(u_right, np.float32)).to(f"cuda:{dev}") if u_right is not None else None
        d_cs = torch.from_numpy(np.ascontiguousarray(cur_state, np.int8)).to(f"cuda:{dev}") if cur_state is not None else None
        kun = self._dev_out((nframes, cap * 28), torch.uint8, dev, 0)
        assign = self._dev_out((nframes, cap), torch.int32, dev, -1); nm = self._dev_out((nframes,), torch.int32, dev, 0)
        k4 = np.ascontiguousarray(k4, np.float32); dist = np.ascontiguousarray(dist, np.float32).ravel()
        self._launch_fence(d_p, d_d, d_n, d_ur, d_cs, kun, assign, nm)
        ur_p = c_void_p(d_ur.data_ptr()) if d_ur is not None else None
        cs_p = c_void_p(d_cs.data_ptr()) if d_cs is not None else None
        if map_points_nnratio is None:
            _check(self._lib.orbx_search_by_projection_frames(self._h, ex._h, nframes, k4.ctypes.data, dist.ctypes.data if len(dist) else None, len(dist), None,
                                                              bounds[0], bounds[1], bounds[2], bounds[3], mbf, ur_p, cs_p, c_void_p(d_p.data_ptr()),
                                                              c_void_p(d_d.data_ptr()), c_void_p(d_n.data_ptr()), max_np, th, int(forward), int(backward),
                                                              int(checkOrientation), c_void_p(kun.data_ptr()), c_void_p(assign.data_ptr()),
                                                              c_void_p(nm.data_ptr())), "orbx_search_by_projection_frames")
        else:  # SearchByProjection(Frame, vector<MapPoint*>): `proj` holds the map-point records
            _check(self._lib.orbx_search_by_projection_points_frames(self._h, ex._h, nframes, k4.ctypes.data, dist.ctypes.data if len(dist) else None, len(dist),
                                                                     None, bounds[0], bounds[1], bounds[2], bounds[3], ur_p, cs_p, c_void_p(d_p.data_ptr()),
                                                                     c_void_p(d_d.data_ptr()), c_void_p(d_n.data_ptr()), max_np, th, float(map_points_nnratio),
                                                                     c_void_p(kun.data_ptr()), c_void_p(assign.data_ptr()), c_void_p(nm.data_ptr())),
                   "orbx_search_by_projection_points_frames")
        if not download:
            return nm, assign, kun
        hk = np.empty((nframes, cap), KEYPOINT_DTYPE)
        _check(self._lib.orbx_matcher_download(self._h, c_void_p(kun.data_ptr()), hk.ctypes.data, hk.nbytes), "orbx_matcher_download")
        return self._download(nm), self._download(assign), hk

    def SearchByProjectionFisheyeFrames(self, left: "ORBextractor", right: "ORBextractor", nframes: int, bounds, cur_state, proj_left, proj_right,
                                        proj_desc, th: float, forward: bool = False, backward: bool = False, checkOrientation: bool = True,
                                        left_to_right=None, right_to_left=None, map_points_nnratio=None):
        """The Nleft != -1 (KB8 fisheye rig) branches of ORBmatcher::SearchByProjection for a batch of frames, device-resident: frame p =
        frame p of both extractors' last batches.  cur_state [P, capL + capR] (combined indices: left i, right capL + j); proj_left[p] /
        proj_right[p] / proj_desc[p]: the records of frame p.  map_points_nnratio: the map-point variant (then left_to_right [P, capL] /
        right_to_left [P, capR] = mvLeftToRightMatch / mvRightToLeftMatch).  Returns (nmatches[P], assign[P, capL + capR])."""
        import torch
        dev = left.device
        capC = left.capacity + right.capacity
        max_np = max(1, max(len(p) for p in proj_left))
        hl = np.zeros((nframes, max_np), PROJECTION_DTYPE); hr = np.zeros((nframes, max_np), PROJECTION_DTYPE)
        hd = np.zeros((nframes, max_np, 32), np.uint8); hn = np.zeros(nframes, np.int32)
        for p in range(nframes):
            hn[p] = len(proj_left[p]); hl[p, : hn[p]] = proj_left[p]; hr[p, : hn[p]] = proj_right[p]
            hd[p, : hn[p]] = np.asarray(proj_desc[p], np.uint8).reshape(-1, 32)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(f"cuda:{dev}")
        d_l, d_r, d_d, d_n = up(hl.view(np.uint8)), up(hr.view(np.uint8)), up(hd), up(hn)
        d_cs = up(np.ascontiguousarray(cur_state, np.int8).reshape(nframes, capC))
        assign = self._dev_out((nframes, capC), torch.int32, dev, -1); nm = self._dev_out((nframes,), torch.int32, dev, 0)
        keep = [d_l, d_r, d_d, d_n, d_cs, assign, nm]
        if map_points_nnratio is None:
            self._launch_fence(*keep)
            _check(self._lib.orbx_search_by_projection_fisheye_frames(self._h, left._h, right._h, nframes, bounds[0], bounds[1], bounds[2], bounds[3],
                                                                      c_void_p(d_cs.data_ptr()), c_void_p(d_l.data_ptr()), c_void_p(d_r.data_ptr()),
                                                                      c_void_p(d_d.data_ptr()), c_void_p(d_n.data_ptr()), max_np, th, int(forward),
                                                                      int(backward), int(checkOrientation), c_void_p(assign.data_ptr()),
                                                                      c_void_p(nm.data_ptr())), "orbx_search_by_projection_fisheye_frames")
        else:
            d_lr = up(np.ascontiguousarray(left_to_right, np.int32).reshape(nframes, left.capacity))
            d_rl = up(np.ascontiguousarray(right_to_left, np.int32).reshape(nframes, right.capacity))
            self._launch_fence(*keep, d_lr, d_rl)
            _check(self._lib.orbx_search_by_projection_points_fisheye_frames(self._h, left._h, right._h, nframes, bounds[0], bounds[1], bounds[2], bounds[3],
                                                                             c_void_p(d_cs.data_ptr()), c_void_p(d_l.data_ptr()), c_void_p(d_r.data_ptr()),
                                                                             c_void_p(d_d.data_ptr()), c_void_p(d_n.data_ptr()), max_np,
                                                                             c_void_p(d_lr.data_ptr()), c_void_p(d_rl.data_ptr()), th,
                                                                             float(map_points_nnratio), c_void_p(assign.data_ptr()),
                                                                             c_void_p(nm.data_ptr())), "orbx_search_by_projection_points_fisheye_frames")
        return self._download(nm), self._download(assign)

    def SearchForInitializationFrames(self, ex1: "ORBextractor", ex2: "ORBextractor", npairs: int, k4, dist, bounds, windowSize: int = 100,
                                      nnratio: float = 0.9, checkOrientation: bool = True, frame2_offset: int = 0, prev_matched=None,
                                      download: bool = True):
        """ORBmatcher::SearchForInitialization (reference src/ORBmatcher1.cc:650-765) for `npairs` frame pairs in one launch, on the
        extractors' device-resident keypoints / descriptors: pair p = (frame p of ex1, frame p + frame2_offset of ex2).  prev_matched
        [npairs, cap, 2] float32 (None: F1's undistorted keypoint positions).  Returns (nmatches[P], matches12[P, cap], prev[P, cap, 2])."""
        import torch
        dev = ex1.device
        cap = ex1.capacity
        if prev_matched is None:
            prev = self._dev_out((npairs, cap, 2), torch.float32, dev, 0.0)
        else:
            prev = torch.from_numpy(np.ascontiguousarray(prev_matched, np.float32).reshape(npairs, cap, 2)).to(f"cuda:{dev}")
        m12 = self._dev_out((npairs, cap), torch.int32, dev, -1); nm = self._dev_out((npairs,), torch.int32, dev, 0)
        k4 = np.ascontiguousarray(k4, np.float32); dist = np.ascontiguousarray(dist, np.float32).ravel()
        self._launch_fence(prev, m12, nm)
        _check(self._lib.orbx_search_for_initialization_frames(self._h, ex1._h, ex2._h, int(frame2_offset), int(npairs), k4.ctypes.data,
                                                               dist.ctypes.data if len(dist) else None, len(dist), None, bounds[0], bounds[1], bounds[2],
                                                               bounds[3], c_void_p(prev.data_ptr()), int(prev_matched is None), int(windowSize),
                                                               float(nnratio), int(checkOrientation), c_void_p(m12.data_ptr()), c_void_p(nm.data_ptr())),
               "orbx_search_for_initialization_frames")
        if not download:
            return nm, m12, prev
        hp = np.empty((npairs, cap, 2), np.float32)
        _check(self._lib.orbx_matcher_download(self._h, c_void_p(prev.data_ptr()), hp.ctypes.data, hp.nbytes), "orbx_matcher_download")
        return self._download(nm), self._download(m12), hp

    def stereo_rows(self, kl, dl, kr, dr, n_rows: int, scale_factors, min_d: float, max_d: float):
        kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE)
        kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        bi = np.full(len(kl), -1, np.int32)
        bd = np.full(len(kl), TH_HIGH, np.int32)
        _check(self._lib.orbx_match_stereo_rows(self._h, kl.ctypes.data, dl.ctypes.data, len(kl), kr.ctypes.data, dr.ctypes.data,
                                                len(kr), n_rows, sf.ctypes.data, len(sf), min_d, max_d, bi.ctypes.data, bd.ctypes.data),
               "orbx_match_stereo_rows")
        return bi, bd

    def ComputeStereoMatches(self, left: "ORBextractor", right: "ORBextractor", kl, dl, kr, dr, mb: float, mbf: float, frame_left: int = 0,
                             frame_right: int = 0):
        """Frame::ComputeStereoMatches (reference src/Frame.cc:841-1011) -> (mvuRight, mvDepth)."""
        kl = np.ascontiguousarray(kl, KEYPOINT_DTYPE)
        kr = np.ascontiguousarray(kr, KEYPOINT_DTYPE)
        dl = np.ascontiguousarray(dl, np.uint8).reshape(-1, 32)
        dr = np.ascontiguousarray(dr, np.uint8).reshape(-1, 32)
        ur = np.full(len(kl), -1, np.float32)
        dp = np.full(len(kl), -1, np.float32)
        _check(self._lib.orbx_compute_stereo_matches(self._h, left._h, frame_left, right._h, frame_right, kl.ctypes.data, dl.ctypes.data,
                                                     len(kl), kr.ctypes.data, dr.ctypes.data, len(kr), mb, mbf, ur.ctypes.data,
                                                     dp.ctypes.data), "orbx_compute_stereo_matches")
        return ur, dp

    # ---- SURVEY §8(f) rank 2 ----------------------------------------------------------------------------------------
    FRAME_GRID_COLS = 64
    FRAME_GRID_ROWS = 48

    def SearchByBoW(self, kf_fv, kf_keys, kf_desc, kf_valid, f_fv, f_keys, f_desc, nnratio: float = 0.7, checkOrientation: bool = True):
        """ORBmatcher::SearchByBoW(pKF, F, vpMapPointMatches) (reference src/ORBmatcher1.cc:225-427, Nleft == -1).  kf_fv / f_fv: the
        FeatureVectors as ORBVocabulary.transform returns them (fv_node, fv_start, fv_feature); kf_valid[i] = map point i of the
        keyframe exists and is not bad.  Returns (nmatches, matches[f_n]) with matches[f] = keyframe feature index or -1."""
        kk = np.ascontiguousarray(kf_keys, KEYPOINT_DTYPE); fk = np.ascontiguousarray(f_keys, KEYPOINT_DTYPE)
        kd = np.ascontiguousarray(kf_desc, np.uint8).reshape(-1, 32); fd = np.ascontiguousarray(f_desc, np.uint8).reshape(-1, 32)
        kv = np.ascontiguousarray(kf_valid, np.uint8)
        assert len(kd) == len(kk) == len(kv) and len(fd) == len(fk)
        a = [np.ascontiguousarray(kf_fv[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
        b = [np.ascontiguousarray(f_fv[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
        matches = np.full(len(fk), -1, np.int32)
        n = self._lib.orbx_search_by_bow(self._h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]), kk.ctypes.data, kd.ctypes.data,
                                         kv.ctypes.data, len(kk), b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, len(b[0]), fk.ctypes.data,
                                         fd.ctypes.data, len(fk), float(nnratio), int(bool(checkOrientation)), matches.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_bow")
        return n, matches

    def SearchByBoWKeyFrames(self, fv1, keys1, desc1, valid1, fv2, keys2, desc2, valid2, nnratio: float = 0.8, checkOrientation: bool = True):
        """ORBmatcher::SearchByBoW(pKF1, pKF2, vpMatches12) (reference src/ORBmatcher2.cc:37-177).  Returns (nmatches, matches12[n1]) with
        matches12[idx1] = feature of the second keyframe whose map point vpMatches12[idx1] becomes, or -1."""
        k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(desc2, np.uint8).reshape(-1, 32)
        v1 = np.ascontiguousarray(valid1, np.uint8); v2 = np.ascontiguousarray(valid2, np.uint8)
        assert len(d1) == len(k1) == len(v1) and len(d2) == len(k2) == len(v2)
        a = [np.ascontiguousarray(fv1[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
        b = [np.ascontiguousarray(fv2[key], np.int32) for key in ("fv_node", "fv_start", "fv_feature")]
        matches = np.full(len(k1), -1, np.int32)
        n = self._lib.orbx_search_by_bow_keyframes(self._h, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]), k1.ctypes.data,
                                                   d1.ctypes.data, v1.ctypes.data, len(k1), b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data,
                                                   len(b[0]), k2.ctypes.data, d2.ctypes.data, v2.ctypes.data, len(k2), float(nnratio),
                                                   int(bool(checkOrientation)), matches.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_bow_keyframes")
        return n, matches

    def SearchByProjectionSim3(self, keys, desc, bounds, points, point_desc, matched, ratioHamming: float = 1.0, nlevels: int = 8):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th, ratioHamming) (reference src/ORBmatcher1.cc:429-533, 535-648), from
        GetFeaturesInArea on -> (nmatches, new_match[n]): the point that now matches keypoint idx, -1 = unchanged."""
        k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
        mt = np.ascontiguousarray(matched, np.uint8)
        assert len(k) == len(d) == len(mt) and len(p) == len(pd)
        out = np.full(len(k), -1, np.int32)
        n = self._lib.orbx_search_by_projection_sim3(self._h, k.ctypes.data, d.ctypes.data, len(k), bounds[0], bounds[1], bounds[2], bounds[3], int(nlevels),
                                                     p.ctypes.data, pd.ctypes.data, len(p), mt.ctypes.data, float(ratioHamming), out.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection_sim3")
        return n, out

    def SearchByProjectionKeyFrame(self, keys, desc, bounds, points, point_desc, point_angle, has_map_point, ORBdist: int = 100,
                                   checkOrientation: bool = True, nlevels: int = 8):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (reference src/ORBmatcher3.cc:469-590), from
        GetFeaturesInArea on -> (nmatches, new_match[n]): keyframe feature whose map point the frame's keypoint i2 now holds, -1 = unchanged."""
        k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
        pa = np.ascontiguousarray(point_angle, np.float32); hm = np.ascontiguousarray(has_map_point, np.uint8)
        assert len(k) == len(d) == len(hm) and len(p) == len(pd) == len(pa)
        out = np.full(len(k), -1, np.int32)
        n = self._lib.orbx_search_by_projection_keyframe(self._h, k.ctypes.data, d.ctypes.data, len(k), bounds[0], bounds[1], bounds[2], bounds[3],
                                                         int(nlevels), p.ctypes.data, pd.ctypes.data, pa.ctypes.data, len(p), hm.ctypes.data, int(ORBdist),
                                                         int(bool(checkOrientation)), out.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection_keyframe")
        return n, out

    def ComputeDistinctiveDescriptors(self, desc, start):
        """MapPoint::ComputeDistinctiveDescriptors (reference src/MapPoint.cc:329-403) for a batch of map points: the observed descriptors of
        point p are rows start[p] .. start[p + 1] of desc -> best[p] = row (relative) that becomes mDescriptor, -1 without observations."""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32); st = np.ascontiguousarray(start, np.int32)
        best = np.full(len(st) - 1, -1, np.int32)
        _check(self._lib.orbx_distinctive_descriptors(self._h, d.ctypes.data, st.ctypes.data, len(st) - 1, best.ctypes.data), "orbx_distinctive_descriptors")
        return best

    def SearchKeyFrameWindows(self, keys, desc, bounds, points, point_desc, nlevels: int = 8, u_right=None, inv_level_sigma2=None, chi2_test: bool = False):
        """The per-map-point search of ORBmatcher::Fuse / SearchBySim3 (reference src/ORBmatcher2.cc:522-579, 690-714; src/ORBmatcher3.cc:
        120-145, 188-213): points = WINDOW_POINT_DTYPE records -> (best_idx[np], best_dist[np])."""
        k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
        assert len(k) == len(d) and len(p) == len(pd)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        bi = np.full(len(p), -1, np.int32); bd = np.full(len(p), np.iinfo(np.int32).max, np.int32)
        _check(self._lib.orbx_search_keyframe_windows(self._h, k.ctypes.data, d.ctypes.data, None if ur is None else ur.ctypes.data, len(k), bounds[0],
                                                      bounds[1], bounds[2], bounds[3], None if sg is None else sg.ctypes.data, int(nlevels),
                                                      p.ctypes.data, pd.ctypes.data, len(p), int(bool(chi2_test)), bi.ctypes.data, bd.ctypes.data),
               "orbx_search_keyframe_windows")
        return bi, bd

    def Fuse(self, keys, desc, bounds, points, point_desc, nlevels: int = 8, u_right=None, inv_level_sigma2=None, chi2_test: bool = True):
        """ORBmatcher::Fuse, the search part (reference src/ORBmatcher2.cc:420-610; chi2_test=False: the Sim3 overload :612-770) ->
        (nFused, fused_idx[np]): keypoint each map point fuses into (bestDist <= TH_LOW) or -1."""
        k = np.ascontiguousarray(keys, KEYPOINT_DTYPE); d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        p = np.ascontiguousarray(points, WINDOW_POINT_DTYPE); pd = np.ascontiguousarray(point_desc, np.uint8).reshape(-1, 32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        sg = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        fi = np.full(len(p), -1, np.int32)
        n = self._lib.orbx_fuse(self._h, k.ctypes.data, d.ctypes.data, None if ur is None else ur.ctypes.data, len(k), bounds[0], bounds[1], bounds[2],
                                bounds[3], None if sg is None else sg.ctypes.data, int(nlevels), p.ctypes.data, pd.ctypes.data, len(p),
                                int(bool(chi2_test)), fi.ctypes.data)
        if n < 0:
            _check(n, "orbx_fuse")
        return n, fi

    def SearchBySim3(self, keys1, kf_desc1, bounds1, keys2, kf_desc2, bounds2, points_1in2, mp_desc1, points_2in1, mp_desc2, nlevels: int = 8):
        """ORBmatcher::SearchBySim3 (reference src/ORBmatcher3.cc:37-254) -> (nFound, matches12[n1])."""
        k1 = np.ascontiguousarray(keys1, KEYPOINT_DTYPE); k2 = np.ascontiguousarray(keys2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(kf_desc1, np.uint8).reshape(-1, 32); d2 = np.ascontiguousarray(kf_desc2, np.uint8).reshape(-1, 32)
        p12 = np.ascontiguousarray(points_1in2, WINDOW_POINT_DTYPE); p21 = np.ascontiguousarray(points_2in1, WINDOW_POINT_DTYPE)
        m1 = np.ascontiguousarray(mp_desc1, np.uint8).reshape(-1, 32); m2 = np.ascontiguousarray(mp_desc2, np.uint8).reshape(-1, 32)
        assert len(p12) == len(k1) == len(m1) and len(p21) == len(k2) == len(m2)
        b1 = np.ascontiguousarray(bounds1, np.float32); b2 = np.ascontiguousarray(bounds2, np.float32)
        out = np.full(len(k1), -1, np.int32)
        n = self._lib.orbx_search_by_sim3(self._h, k1.ctypes.data, d1.ctypes.data, len(k1), b1.ctypes.data, k2.ctypes.data, d2.ctypes.data, len(k2),
                                          b2.ctypes.data, p12.ctypes.data, m1.ctypes.data, p21.ctypes.data, m2.ctypes.data, int(nlevels), out.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_sim3")
        return n, out

    def SearchForTriangulation(self, kf1, kf2, geometry, bOnlyStereo: bool = False, bCoarse: bool = False, checkOrientation: bool = True):
        """ORBmatcher::SearchForTriangulation(pKF1, pKF2, vMatchedPairs, bOnlyStereo, bCoarse) (reference src/ORBmatcher2.cc:179-418).
        kf1 / kf2: keyframe_view(...) results (view, keep-alive), geometry: triangulation_geometry(...).
        Returns (nmatches, vMatchedPairs[nmatches, 2], matches12[n1])."""
        v1, v2 = kf1[0], kf2[0]
        matches = np.full(v1.n, -1, np.int32)
        pairs = np.zeros((max(v1.n, 1), 2), np.int32)
        n = self._lib.orbx_search_for_triangulation(self._h, byref(v1), byref(v2), byref(geometry), int(bool(bOnlyStereo)), int(bool(bCoarse)),
                                                    int(bool(checkOrientation)), matches.ctypes.data, pairs.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_for_triangulation")
        return n, pairs[:n].copy(), matches

    def SearchForTriangulationFrames(self, voc: "ORBVocabulary", ex: "ORBextractor", ref_frame: int, neighbours, k4, dist, geometries,
                                     has_map_point, u_right=None, bOnlyStereo: bool = False, bCoarse: bool = False,
                                     checkOrientation: bool = True):
        """ORBmatcher::SearchForTriangulation of keyframe `ref_frame` against each of `neighbours` (frames of the extractor's last batch),
        device-resident: the FeatureVectors are the device results of the last voc.transform_batch(ex, ..) (orbx_search_for_triangulation_frames).
        geometries: one triangulation_geometry(..) per neighbour; has_map_point [P, cap] uint8; u_right [P, cap] or None.
        Returns (nmatches [K], matches12 [K, cap])."""
        import torch
        dev = ex.device
        cap = ex.capacity
        nb = np.ascontiguousarray(neighbours, np.int32)
        K = len(nb)
        garr = (TriangulationGeometryC * max(K, 1))(*geometries)
        dv = f"cuda:{dev}"
        d_h = torch.from_numpy(np.ascontiguousarray(has_map_point, np.uint8)).to(dv)
        d_ur = torch.from_numpy(np.ascontiguousarray(u_right, np.float32)).to(dv) if u_right is not None else None
        m12 = self._dev_out((max(K, 1), cap), torch.int32, dev, -1); nm = self._dev_out((max(K, 1),), torch.int32, dev, 0)
        k4 = np.ascontiguousarray(k4, np.float32); dist = np.ascontiguousarray(dist, np.float32).ravel()
        self._launch_fence(d_h, d_ur, m12, nm)
        _check(self._lib.orbx_search_for_triangulation_frames(self._h, voc._h, ex._h, int(ref_frame), nb.ctypes.data, K, k4.ctypes.data,
                                                              dist.ctypes.data if len(dist) else None, len(dist), None, ctypes.byref(garr),
                                                              c_void_p(d_h.data_ptr()), c_void_p(d_ur.data_ptr()) if d_ur is not None else None,
                                                              int(bool(bOnlyStereo)), int(bool(bCoarse)), int(bool(checkOrientation)),
                                                              c_void_p(m12.data_ptr()), c_void_p(nm.data_ptr())), "orbx_search_for_triangulation_frames")
        return self._download(nm)[:K], self._download(m12)[:K]

    def SearchByBoWFrames(self, voc: "ORBVocabulary", ex: "ORBextractor", kf_frames, f_frames, valid, nnratio: float = 0.75,
                          checkOrientation: bool = True, keyframe_pair: bool = False):
        """ORBmatcher::SearchByBoW for pairs of frames of the extractor's last batch, device-resident (orbx_search_by_bow_frames): pair k =
        (keyframe kf_frames[k], frame f_frames[k]) as in Tracking::Relocalization, or (first, second keyframe) with keyframe_pair=True as in
        LoopClosing; valid [P, cap] uint8 (the keypoint's map point exists and is not bad).  Returns (nmatches [K], matches [K, cap])."""
        import torch
        dev = ex.device
        cap = ex.capacity
        a = np.ascontiguousarray(kf_frames, np.int32); b = np.ascontiguousarray(f_frames, np.int32)
        assert len(a) == len(b)
        K = len(a)
        d_v = torch.from_numpy(np.ascontiguousarray(valid, np.uint8)).to(f"cuda:{dev}")
        mt = self._dev_out((max(K, 1), cap), torch.int32, dev, -1); nm = self._dev_out((max(K, 1),), torch.int32, dev, 0)
        self._launch_fence(d_v, mt, nm)
        _check(self._lib.orbx_search_by_bow_frames(self._h, voc._h, ex._h, a.ctypes.data, b.ctypes.data, K, int(bool(keyframe_pair)),
                                                   c_void_p(d_v.data_ptr()), float(nnratio), int(bool(checkOrientation)), c_void_p(mt.data_ptr()),
                                                   c_void_p(nm.data_ptr())), "orbx_search_by_bow_frames")
        return self._download(nm)[:K], self._download(mt)[:K]

    @staticmethod
    def _camera(k4, dist, knew4):
        k4 = np.ascontiguousarray(k4, np.float32)
        d = np.ascontiguousarray(dist, np.float32).ravel()
        kn = k4 if knew4 is None else np.ascontiguousarray(knew4, np.float32)
        if k4.shape != (4,) or kn.shape != (4,) or len(d) > 12:
            raise ValueError("camera: k4 / knew4 = (fx, fy, cx, cy), at most 12 distortion coefficients")
        return k4, d, kn

    def UndistortKeyPoints(self, kps, k4, dist, knew4=None):
        """Frame::UndistortKeyPoints (reference src/Frame.cc:777-811) -> mvKeysUn.  k4 = toK() as (fx, fy, cx, cy), dist = mDistCoef,
        knew4 = mK (default: k4)."""
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        k4, d, kn = self._camera(k4, dist, knew4)
        out = np.empty_like(kps)
        _check(self._lib.orbx_undistort_keypoints(self._h, kps.ctypes.data, len(kps), k4.ctypes.data, d.ctypes.data if len(d) else None, len(d),
                                                  kn.ctypes.data, out.ctypes.data), "orbx_undistort_keypoints")
        return out

    def ComputeImageBounds(self, cols: int, rows: int, k4, dist, knew4=None):
        """Frame::ComputeImageBounds (reference src/Frame.cc:813-845) -> (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        k4, d, kn = self._camera(k4, dist, knew4)
        b = np.zeros(4, np.float32)
        _check(self._lib.orbx_compute_image_bounds(self._h, int(cols), int(rows), k4.ctypes.data, d.ctypes.data if len(d) else None, len(d),
                                                   kn.ctypes.data, b.ctypes.data), "orbx_compute_image_bounds")
        return tuple(float(v) for v in b)

    def AssignFeaturesToGrid(self, kps, bounds):
        """Frame::AssignFeaturesToGrid (reference src/Frame.cc:387-418) for the undistorted keypoints `kps` and the image bounds
        (mnMinX, mnMaxX, mnMinY, mnMaxY).  Returns (cell_start[64 * 48 + 1], items): mGrid[ix][iy] = items[cell_start[c] :
        cell_start[c + 1]] with c = ix * 48 + iy."""
        kps = np.ascontiguousarray(kps, KEYPOINT_DTYPE)
        cs = np.zeros(self.FRAME_GRID_COLS * self.FRAME_GRID_ROWS + 1, np.int32)
        items = np.zeros(max(len(kps), 1), np.int32)
        n = self._lib.orbx_assign_features_to_grid(self._h, kps.ctypes.data, len(kps), *[float(b) for b in bounds], cs.ctypes.data,
                                                   items.ctypes.data)
        if n < 0:
            _check(n, "orbx_assign_features_to_grid")
        return cs, items[:n]

    def SearchForInitialization(self, k1, d1, k2, d2, vbPrevMatched, bounds, windowSize: int = 10, nnratio: float = 0.6,
                                checkOrientation: bool = True):
        """ORBmatcher::SearchForInitialization (reference src/ORBmatcher1.cc:650-765; ORBmatcher(nnratio, checkOri) as in
        include/ORBmatcher.h:40).  Returns (nmatches, vnMatches12, vbPrevMatched updated)."""
        k1 = np.ascontiguousarray(k1, KEYPOINT_DTYPE)
        k2 = np.ascontiguousarray(k2, KEYPOINT_DTYPE)
        d1 = np.ascontiguousarray(d1, np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(d2, np.uint8).reshape(-1, 32)
        prev = np.array(vbPrevMatched, np.float32).reshape(-1, 2).copy()
        assert len(prev) == len(k1)
        m12 = np.full(len(k1), -1, np.int32)
        n = self._lib.orbx_search_for_initialization(self._h, k1.ctypes.data, d1.ctypes.data, len(k1), k2.ctypes.data, d2.ctypes.data,
                                                     len(k2), *[float(b) for b in bounds], prev.ctypes.data, m12.ctypes.data,
                                                     int(windowSize), float(nnratio), int(bool(checkOrientation)))
        if n < 0:
            _check(n, "orbx_search_for_initialization")
        return n, m12, prev

    def SearchByProjection(self, kc, dc, cur_state, bounds, scale_factors, proj, proj_desc, th: float, *, u_right=None, mbf: float = 0.0,
                           forward: bool = False, backward: bool = False, checkOrientation: bool = True):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (reference src/ORBmatcher3.cc:256-467, Nleft == -1) on
        projections computed by the caller (PROJECTION_DTYPE).  Returns (nmatches, assign[nc]) with assign = -1 untouched, j >= 0 the
        map point of last-frame keypoint j, -2 set to NULL by the rotation check."""
        kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
        dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
        st = np.ascontiguousarray(cur_state, np.int8)
        pr = np.ascontiguousarray(proj, PROJECTION_DTYPE)
        pd = np.ascontiguousarray(proj_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        assert len(st) == len(kc) and len(pd) == len(pr) and (ur is None or len(ur) == len(kc))
        assign = np.full(len(kc), -1, np.int32)
        n = self._lib.orbx_search_by_projection(self._h, kc.ctypes.data, dc.ctypes.data, None if ur is None else ur.ctypes.data, st.ctypes.data,
                                                len(kc), *[float(b) for b in bounds], sf.ctypes.data, len(sf), float(mbf), pr.ctypes.data,
                                                pd.ctypes.data, len(pr), float(th), int(forward), int(backward), int(checkOrientation),
                                                assign.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection")
        return n, assign

    def SearchByProjectionPoints(self, kc, dc, cur_state, bounds, scale_factors, map_points, mp_desc, th: float = 1.0, nnratio: float = 0.6,
                                 *, u_right=None):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ..) (reference src/ORBmatcher1.cc:45-215, Nleft == -1).  map_points:
        PROJECTION_DTYPE with u, v = mTrackProjX / Y, invz = mTrackProjXR, angle = mTrackViewCos, octave = mnTrackScaleLevel.
        Returns (nmatches, assign[nc])."""
        kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
        dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
        st = np.ascontiguousarray(cur_state, np.int8)
        pr = np.ascontiguousarray(map_points, PROJECTION_DTYPE)
        pd = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        ur = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        assert len(st) == len(kc) and len(pd) == len(pr) and (ur is None or len(ur) == len(kc))
        assign = np.full(len(kc), -1, np.int32)
        n = self._lib.orbx_search_by_projection_points(self._h, kc.ctypes.data, dc.ctypes.data, None if ur is None else ur.ctypes.data,
                                                       st.ctypes.data, len(kc), *[float(b) for b in bounds], sf.ctypes.data, len(sf),
                                                       pr.ctypes.data, pd.ctypes.data, len(pr), float(th), float(nnratio), assign.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection_points")
        return n, assign

    def SearchByProjectionFisheye(self, kc, dc, n_left: int, cur_state, bounds, scale_factors, proj_left, proj_right, proj_desc, th: float, *,
                                  forward: bool = False, backward: bool = False, checkOrientation: bool = True):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono), CurrentFrame.Nleft != -1 (reference src/ORBmatcher3.cc:300-436):
        kc / dc = mvKeys then mvKeysRight / mDescriptors; proj_right[i].u / .v = the projection into the right camera.  Returns
        (nmatches, assign[n_left + n_right])."""
        kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
        dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
        st = np.ascontiguousarray(cur_state, np.int8)
        pl = np.ascontiguousarray(proj_left, PROJECTION_DTYPE)
        pr = np.ascontiguousarray(proj_right, PROJECTION_DTYPE)
        pd = np.ascontiguousarray(proj_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        assert len(st) == len(kc) == len(dc) and len(pd) == len(pl) == len(pr) and 0 <= n_left <= len(kc)
        assign = np.full(len(kc), -1, np.int32)
        n = self._lib.orbx_search_by_projection_fisheye(self._h, kc.ctypes.data, dc.ctypes.data, int(n_left), len(kc) - int(n_left), st.ctypes.data,
                                                        *[float(b) for b in bounds], sf.ctypes.data, len(sf), pl.ctypes.data, pr.ctypes.data,
                                                        pd.ctypes.data, len(pl), float(th), int(forward), int(backward), int(checkOrientation),
                                                        assign.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection_fisheye")
        return n, assign

    def SearchByProjectionPointsFisheye(self, kc, dc, n_left: int, cur_state, bounds, scale_factors, map_points_left, map_points_right, mp_desc,
                                        left_to_right, right_to_left, th: float = 1.0, nnratio: float = 0.6):
        """ORBmatcher::SearchByProjection(F, vpMapPoints, th, ..), F.Nleft != -1 (reference src/ORBmatcher1.cc:45-215).  Returns
        (nmatches, assign[n_left + n_right])."""
        kc = np.ascontiguousarray(kc, KEYPOINT_DTYPE)
        dc = np.ascontiguousarray(dc, np.uint8).reshape(-1, 32)
        st = np.ascontiguousarray(cur_state, np.int8)
        pl = np.ascontiguousarray(map_points_left, PROJECTION_DTYPE)
        pr = np.ascontiguousarray(map_points_right, PROJECTION_DTYPE)
        pd = np.ascontiguousarray(mp_desc, np.uint8).reshape(-1, 32)
        sf = np.ascontiguousarray(scale_factors, np.float32)
        l2r = np.ascontiguousarray(left_to_right, np.int32)
        r2l = np.ascontiguousarray(right_to_left, np.int32)
        assert len(st) == len(kc) == len(dc) and len(pd) == len(pl) == len(pr) and len(l2r) == n_left and len(r2l) == len(kc) - n_left
        assign = np.full(len(kc), -1, np.int32)
        n = self._lib.orbx_search_by_projection_points_fisheye(self._h, kc.ctypes.data, dc.ctypes.data, int(n_left), len(kc) - int(n_left),
                                                               st.ctypes.data, *[float(b) for b in bounds], sf.ctypes.data, len(sf),
                                                               pl.ctypes.data, pr.ctypes.data, pd.ctypes.data, len(pl), l2r.ctypes.data,
                                                               r2l.ctypes.data, float(th), float(nnratio), assign.ctypes.data)
        if n < 0:
            _check(n, "orbx_search_by_projection_points_fisheye")
        return n, assign


class ORBVocabulary:
    """Mirror of ORB_SLAM3::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB> (reference include/ORBVocabulary.h,
    Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h) for the part Frame::ComputeBoW uses: the tree resident in HBM, transform() on the
    GPU.  Build it from node arrays (parent / is_leaf / desc / weight in loadFromTextFile's node order) or from an ORBvoc.txt."""

    def __init__(self, k=10, L=5, weighting=0, scoring=0, *, parent=None, is_leaf=None, desc=None, weight=None, device: int = 0):
        self._lib = load_library()
        self._h = c_void_p()
        self._device = device
        if parent is not None:
            parent = np.ascontiguousarray(parent, np.int32)
            is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
            desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
            weight = np.ascontiguousarray(weight, np.float64)
            if not (len(parent) == len(is_leaf) == len(desc) == len(weight)):
                raise ValueError("ORBVocabulary: one parent / is_leaf / descriptor / weight per node")
            _check(self._lib.orbx_vocabulary_create(device, int(k), int(L), int(scoring), int(weighting), len(parent), parent.ctypes.data,
                                                    is_leaf.ctypes.data, desc.ctypes.data, weight.ctypes.data, ctypes.byref(self._h)),
                   "orbx_vocabulary_create")

    def loadFromTextFile(self, filename: str) -> bool:
        """bool loadFromTextFile(const std::string &filename) — TemplatedVocabulary.h:1338-1423 (ORBvoc.txt)."""
        self.close()
        rc = self._lib.orbx_vocabulary_load_text(self._device, str(filename).encode(), ctypes.byref(self._h))
        if rc != 0:
            self._h = c_void_p()
            return False
        return True

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orbx_vocabulary_destroy(self._h)
        self._h = c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def empty(self) -> bool:
        return not self._h or self.size() == 0

    def size(self) -> int:
        return self._lib.orbx_vocabulary_size(self._h) if self._h else 0

    def transform(self, features, levelsup: int = 4):
        """transform(features, BowVector &v, FeatureVector &fv, levelsup) — TemplatedVocabulary.h:1127-1196.  Returns a dict:
        bow_word / bow_value (the BowVector, ascending word ids), fv_node / fv_start / fv_feature (the FeatureVector as CSR over
        ascending node ids), feat_word / feat_node (per feature, -1 = stopped word)."""
        if not self._h:
            raise OrbxError("ORBVocabulary: no vocabulary loaded")
        d = np.ascontiguousarray(features, np.uint8).reshape(-1, 32)
        n = len(d)
        fw = np.zeros(n, np.int32); fn = np.zeros(n, np.int32)
        bw = np.zeros(max(n, 1), np.int32); bv = np.zeros(max(n, 1), np.float64); nb = np.zeros(1, np.int32)
        vn = np.zeros(max(n, 1), np.int32); vs = np.zeros(n + 1, np.int32); vf = np.zeros(max(n, 1), np.int32); nv = np.zeros(1, np.int32)
        rc = self._lib.orbx_vocabulary_transform(self._h, d.ctypes.data, n, int(levelsup), fw.ctypes.data, fn.ctypes.data, bw.ctypes.data,
                                                 bv.ctypes.data, nb.ctypes.data, vn.ctypes.data, vs.ctypes.data, vf.ctypes.data, nv.ctypes.data)
        if rc < 0:
            _check(rc, "orbx_vocabulary_transform")
        nb, nv = int(nb[0]), int(nv[0])
        return dict(feat_word=fw, feat_node=fn, bow_word=bw[:nb].copy(), bow_value=bv[:nb].copy(), fv_node=vn[:nv].copy(),
                    fv_start=vs[:nv + 1].copy(), fv_feature=vf[:int(vs[nv])].copy())

    def transform_frame(self, extractor: "ORBextractor", frame: int = 0, levelsup: int = 4):
        """Frame::ComputeBoW on the extractor's device-resident descriptors of `frame` of its last batch (no host round trip)."""
        if not self._h:
            raise OrbxError("ORBVocabulary: no vocabulary loaded")
        n = extractor.capacity
        fw = np.zeros(n, np.int32); fn = np.zeros(n, np.int32)
        bw = np.zeros(n, np.int32); bv = np.zeros(n, np.float64); nb = np.zeros(1, np.int32)
        vn = np.zeros(n, np.int32); vs = np.zeros(n + 1, np.int32); vf = np.zeros(n, np.int32); nv = np.zeros(1, np.int32)
        rc = self._lib.orbx_vocabulary_transform_frame(self._h, extractor._h, int(frame), int(levelsup), fw.ctypes.data, fn.ctypes.data,
                                                       bw.ctypes.data, bv.ctypes.data, nb.ctypes.data, vn.ctypes.data, vs.ctypes.data,
                                                       vf.ctypes.data, nv.ctypes.data)
        if rc < 0:
            _check(rc, "orbx_vocabulary_transform_frame")
        nb, nv = int(nb[0]), int(nv[0])
        nf = int(vs[nv])
        return dict(bow_word=bw[:nb].copy(), bow_value=bv[:nb].copy(), fv_node=vn[:nv].copy(), fv_start=vs[:nv + 1].copy(), fv_feature=vf[:nf].copy())

    def transform_batch(self, extractor: "ORBextractor", nframes: int, levelsup: int = 4):
        """Frame::ComputeBoW for the first `nframes` frames of the extractor's last batch, two launches, descriptors read in HBM.
        Returns a list of per-frame dicts like transform_frame."""
        if not self._h:
            raise OrbxError("ORBVocabulary: no vocabulary loaded")
        cap = extractor.capacity
        bw = np.zeros((nframes, cap), np.int32); bv = np.zeros((nframes, cap), np.float64); nb = np.zeros(nframes, np.int32)
        vn = np.zeros((nframes, cap), np.int32); vs = np.zeros((nframes, cap + 1), np.int32); vf = np.zeros((nframes, cap), np.int32)
        nv = np.zeros(nframes, np.int32)
        _check(self._lib.orbx_vocabulary_transform_batch(self._h, extractor._h, int(nframes), int(levelsup), bw.ctypes.data, bv.ctypes.data,
                                                         nb.ctypes.data, vn.ctypes.data, vs.ctypes.data, vf.ctypes.data, nv.ctypes.data),
               "orbx_vocabulary_transform_batch")
        return [dict(bow_word=bw[f, :nb[f]].copy(), bow_value=bv[f, :nb[f]].copy(), fv_node=vn[f, :nv[f]].copy(),
                     fv_start=vs[f, :nv[f] + 1].copy(), fv_feature=vf[f, :vs[f, nv[f]]].copy()) for f in range(nframes)]
