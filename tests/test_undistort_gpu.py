"""GPU parity of Frame::UndistortKeyPoints / Frame::ComputeImageBounds (reference src/Frame.cc:777-845) against
the CPU oracle.  The arithmetic is double on both sides in the same operation order, so the float results must be bit-identical;
camera parameters are the ones of the reference's example settings (EuRoC, TUM-VI-like, KITTI without distortion)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAMERAS = {
    "euroc": ((458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), (752, 480)),
    "tum_rgbd": ((517.306408, 516.469215, 318.643040, 255.313989), (0.262383, -0.953104, -0.005358, 0.002628, 1.163314), (640, 480)),
    "strong_k3k6": ((420.0, 415.0, 320.0, 240.0), (-0.35, 0.15, 0.001, -0.002, -0.03, 0.01, -0.005, 0.002), (640, 480)),
    "thin_prism": ((600.0, 600.0, 640.0, 360.0), (0.1, -0.2, 0.0005, 0.0003, 0.05, 0.0, 0.0, 0.0, 0.001, -0.0005, 0.0008, 0.0002), (1280, 720)),
}


def _keypoints(pkg, rng, n, w, h):
    k = np.zeros(n, pkg.KEYPOINT_DTYPE)
    k["x"] = rng.uniform(0, w, n).astype(np.float32)
    k["y"] = rng.uniform(0, h, n).astype(np.float32)
    k["size"] = 31.0
    k["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k["response"] = rng.uniform(7, 200, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["class_id"] = -1
    return k


@pytest.mark.parametrize("cam", sorted(CAMERAS))
def test_undistort_keypoints_bit_exact(pkg, orc, cam):
    k4, dist, (w, h) = CAMERAS[cam]
    rng = np.random.default_rng(5)
    kps = _keypoints(pkg, rng, 3000, w, h)
    kps[:4]["x"] = [0, w, 0, w]; kps[:4]["y"] = [0, 0, h, h]  # the corners ComputeImageBounds uses
    m = pkg.ORBmatcher(8192, 8192)
    out = m.UndistortKeyPoints(kps, k4, dist)
    ref = orc.undistort_keypoints(kps, k4, dist)
    assert out.tobytes() == ref.tobytes()
    assert np.abs(out["x"] - kps["x"]).max() > 0.5  # the distortion does move points
    for f in ("size", "angle", "response", "octave", "class_id"):
        assert np.array_equal(out[f], kps[f])
    # a rectified output camera (mK != toK()), as for stereo-rectified settings
    knew = (k4[0] * 0.9, k4[1] * 0.9, w / 2.0, h / 2.0)
    assert m.UndistortKeyPoints(kps, k4, dist, knew).tobytes() == orc.undistort_keypoints(kps, k4, dist, knew).tobytes()
    assert m.ComputeImageBounds(w, h, k4, dist) == orc.compute_image_bounds(w, h, k4, dist)
    b = m.ComputeImageBounds(w, h, k4, dist)
    assert b[0] < b[1] and b[2] < b[3]
    m.close()


def test_undistort_without_distortion_and_edges(pkg, orc):
    rng = np.random.default_rng(6)
    kps = _keypoints(pkg, rng, 500, 1241, 376)
    k4 = (718.856, 718.856, 607.1928, 185.2157)  # KITTI: rectified, mDistCoef = 0
    m = pkg.ORBmatcher(8192, 8192)
    for dist in ((0.0, 0.0, 0.0, 0.0), (0.0, 0.3, 0.0, 0.0), ()):  # only distCoef[0] is looked at (src/Frame.cc:779)
        out = m.UndistortKeyPoints(kps, k4, dist)
        assert out.tobytes() == kps.tobytes() == orc.undistort_keypoints(kps, k4, dist).tobytes()
        assert m.ComputeImageBounds(1241, 376, k4, dist) == (0.0, 1241.0, 0.0, 376.0) == orc.compute_image_bounds(1241, 376, k4, dist)
    assert len(m.UndistortKeyPoints(kps[:0], k4, (0.1, 0, 0, 0))) == 0
    # icdist < 0 (a wildly wrong model): OpenCV falls back to the normalised distorted point; same on both sides
    bad = (-40.0, 0.0, 0.0, 0.0)
    assert m.UndistortKeyPoints(kps, k4, bad).tobytes() == orc.undistort_keypoints(kps, k4, bad).tobytes()
    with pytest.raises(pkg.OrbxError, match="focal"):
        m.UndistortKeyPoints(kps, (0.0, 1.0, 0.0, 0.0), (0.1, 0, 0, 0))
    with pytest.raises(pkg.OrbxError, match="capacity"):
        m.UndistortKeyPoints(_keypoints(pkg, rng, 9000, 640, 480), k4, (0.1, 0, 0, 0))
    m.close()
