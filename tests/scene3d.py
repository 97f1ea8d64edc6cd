"""Synthetic 3-D tracking scenes for the projection head (tests/test_projection_head_gpu.py, tests/tools/stress_widening.py): the last frame's
keypoints are un-projected to random depths with the last pose = identity, and seen again from a current pose a small motion away."""
import numpy as np

import ctypes

_libm = ctypes.CDLL("libm.so.6")
_libm.logf.argtypes = [ctypes.c_float]; _libm.logf.restype = ctypes.c_float


def libm_logf(x):
    """The platform's logf - what `log(mfScaleFactor)` (src/Frame.cc:114) and MapPoint::PredictScale's `log(ratio)` resolve to under
    `using namespace std`; numpy's float32 log is its own SIMD routine and may differ in the last bit."""
    return np.float32(_libm.logf(float(np.float32(x))))


def predict_scale_float(ratio, log_scale_factor, nlevels):
    """MapPoint::PredictScale (src/MapPoint.cc:538-544) from its definition: logf, a float division, ceil, int, the two clamps."""
    lv = np.ceil(np.float32(libm_logf(ratio) / np.float32(log_scale_factor)))
    return int(min(max(int(lv), 0), nlevels - 1)) if np.isfinite(lv) else 0


def predict_scale_double(ratio, log_scale_factor, nlevels):
    """The wrong restatement of rounds 1-4 (log and the division in double) - kept to show that a test's inputs tell the two apart."""
    lv = np.ceil(np.log(np.float64(np.float32(ratio))) / np.float64(np.float32(log_scale_factor)))
    return int(min(max(int(lv), 0), nlevels - 1)) if np.isfinite(lv) else 0


PINHOLE = (0, (458.654, 457.296, 367.215, 248.375))                                               # EuRoC cam0 (fx fy cx cy)
KB8 = (1, (190.978477, 190.973307, 254.931706, 256.897442, 0.0034823894, 0.0007150348, -0.0020532361, 0.0002029367))  # TUM-VI cam0


def quat_from_axis_angle(axis, angle):
    a = np.asarray(axis, np.float64); a = a / np.linalg.norm(a)
    s = np.sin(angle / 2)
    return np.float32([a[0] * s, a[1] * s, a[2] * s, np.cos(angle / 2)])   # x y z w


def rotation_matrix(q):
    """Eigen::Quaternionf::toRotationMatrix in double, narrowed: only an INPUT of the in-frustum mode (mRcw is given to the reference too)."""
    x, y, z, w = [float(v) for v in q]
    return np.float32([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                       [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                       [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def unproject(cam, u, v, z):
    ctype, p = cam
    x = (u - p[2]) / p[0]; y = (v - p[3]) / p[1]
    if ctype == 1:  # KB8: theta_d = r (no distortion inversion needed for a test scene: any ray is as good as another)
        r = np.hypot(x, y); th = np.minimum(r, 1.4)
        s = np.where(r > 1e-9, np.tan(th) / np.maximum(r, 1e-9), 1.0)
        x, y = x * s, y * s
    return np.stack([x * z, y * z, z], -1)


def make_scene(rng, keys, cam, world_dtype, motion=0.05, rot=0.02, behind=0.02):
    """World points of the keypoints `keys` (last pose = identity) and a current pose (q xyzw, t, Rcw, Ow).  A few points end up behind the
    current camera / outside the image."""
    n = len(keys)
    z = rng.uniform(0.8, 15.0, n)
    P = unproject(cam, keys["x"].astype(np.float64), keys["y"].astype(np.float64), z)
    P[rng.random(n) < behind, 2] *= -1
    q = quat_from_axis_angle(rng.normal(size=3), float(rng.normal(0, rot)))
    t = rng.normal(0, motion, 3).astype(np.float32)
    R = rotation_matrix(q)
    Ow = (-(R.astype(np.float64).T @ t.astype(np.float64))).astype(np.float32)
    pts = np.zeros(n, world_dtype)
    pts["pos"] = P.astype(np.float32)
    view = P - Ow.astype(np.float64)
    nrm = view / np.maximum(np.linalg.norm(view, axis=1, keepdims=True), 1e-9)
    nrm += rng.normal(0, 0.35, (n, 3))                      # mean viewing directions of the other observations: some beyond 60 degrees
    pts["normal"] = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    d = np.linalg.norm(view, axis=1)
    pts["max_distance"] = (d * rng.uniform(0.6, 3.0, n)).astype(np.float32)     # mfMaxDistance: some points outside 1.2 x
    pts["min_distance"] = (pts["max_distance"] / np.float32(1.2) ** 7 * rng.uniform(0.5, 1.5, n)).astype(np.float32)
    pts["angle"] = keys["angle"]
    pts["octave"] = keys["octave"]
    pts["flags"] = (rng.random(n) < 0.85).astype(np.int32) | ((rng.random(n) < 0.7).astype(np.int32) << 1)
    return pts, q, t, R, Ow
