"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the same seeded
inputs.  Bar: bit-exact keypoints (all 7 fields incl. float angle), octaves, order and 32-byte descriptors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIGS = [  # w, h, nfeatures — BASELINE.json configs 1-5 geometries
    (640, 480, 1000),
    (752, 480, 1200),
    (512, 512, 1500),
    (1920, 1080, 4000),
]


def _pack_level_kps(k):
    return ((k["x"].astype(np.int64) - 16) | ((k["y"].astype(np.int64) - 16) << 12) | (k["response"].astype(np.int64) << 24)).astype(np.uint32)


def _assert_same(pkg, kps, desc, rk, rd, what):
    assert len(kps) == len(rk), f"{what}: count {len(kps)} vs {len(rk)}"
    for name in kps.dtype.names:
        a, b = kps[name].view(np.uint32), rk[name].view(np.uint32)
        if not np.array_equal(a, b):
            i = int(np.nonzero(a != b)[0][0])
            raise AssertionError(f"{what}: field {name} differs at {i}: {kps[name][i]!r} vs {rk[name][i]!r}")
    assert np.array_equal(desc, rd), f"{what}: descriptors differ at rows {np.nonzero((desc != rd).any(axis=1))[0][:5]}"


@pytest.mark.parametrize("w,h,nf", CONFIGS)
def test_stagewise_and_end_to_end(pkg, orc, w, h, nf):
    nframes = 2 if w < 1000 else 1
    imgs = np.stack([pkg.synth_frame(100 + i, w, h) for i in range(nframes)])
    ex = pkg.ORBextractor(nf, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=nframes)
    lap = (0, 1000)
    n, mono, kps, desc = ex.extract_batch(imgs, lap)
    for f in range(nframes):
        ref = orc.Oracle(nf, 1.2, 8, 20, 7)
        rmono, rk, rd = ref.extract(imgs[f], lap)
        for l in range(8):
            assert ex.level_size(l) == ref.level_size(l)
            assert np.array_equal(ex.pyramid_level(f, l), ref.level(l)), f"pyramid level {l}"
            assert np.array_equal(ex.debug_candidates(f, l), ref.candidates(l)), f"FAST candidates level {l}"
            assert np.array_equal(ex.debug_selected(f, l), _pack_level_kps(ref.level_keypoints(l))), f"quadtree level {l}"
            assert np.array_equal(ex.pyramid_level(f, l, True), ref.level(l, True)), f"blur level {l}"
        assert mono[f] == rmono
        _assert_same(pkg, kps[f][: n[f]], desc[f][: n[f]], rk, rd, f"{w}x{h} frame {f}")
        assert n[f] >= nf  # quotas saturate on the synthetic frames
    ex.close()


@pytest.mark.parametrize("lap", [(0, 1000), (0, 0), (250, 400), (639, 5000), (-5, 18)])
def test_lapping_split(pkg, orc, lap):
    img = pkg.synth_frame(21, 640, 480)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    mono, kps, desc = ex(img, None, lap)
    rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract(img, lap)
    assert mono == rmono
    _assert_same(pkg, kps, desc, rk, rd, f"lap {lap}")
    ex.close()


@pytest.mark.parametrize("nf,scale,nlevels,ini,mn", [(500, 1.2, 8, 20, 7), (2000, 1.2, 8, 12, 5), (1000, 1.5, 4, 20, 7),
                                                     (300, 1.1, 6, 30, 30), (1000, 1.2, 1, 20, 7), (7500, 1.2, 8, 20, 7),
                                                     (40, 1.2, 8, 20, 7), (1000, 1.2, 8, 20, 0), (1000, 1.2, 8, 300, 7)])
def test_parameter_space(pkg, orc, nf, scale, nlevels, ini, mn):
    img = pkg.synth_frame(33, 640, 480)
    ex = pkg.ORBextractor(nf, scale, nlevels, ini, mn)
    ref = orc.Oracle(nf, scale, nlevels, ini, mn)
    assert np.array_equal(ex.features_per_level(), ref.tables()["features_per_level"])
    assert np.array_equal(ex.GetScaleFactors(), ref.tables()["scale"])
    assert np.array_equal(ex.GetInverseScaleFactors(), ref.tables()["inv_scale"])
    assert np.array_equal(ex.GetScaleSigmaSquares(), ref.tables()["sigma2"])
    assert np.array_equal(ex.GetInverseScaleSigmaSquares(), ref.tables()["inv_sigma2"])
    assert np.array_equal(ex.umax(), ref.tables()["umax"])
    assert ex.GetLevels() == nlevels and abs(ex.GetScaleFactor() - scale) < 1e-6
    mono, kps, desc = ex(img, None, (0, 1000))
    rmono, rk, rd = ref.extract(img, (0, 1000))
    assert mono == rmono
    _assert_same(pkg, kps, desc, rk, rd, f"params {(nf, scale, nlevels, ini, mn)}")
    ex.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_blur_variants(pkg, orc, variant):
    img = pkg.synth_frame(2, 640, 480)
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, blur_variant=variant)
    mono, kps, desc = ex(img, None, (0, 1000))
    rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7, blur_variant=variant).extract(img, (0, 1000))
    _assert_same(pkg, kps, desc, rk, rd, f"blur variant {variant}")
    ex.close()


def test_image_content_edge_cases(pkg, orc):
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    ref = orc.Oracle(1000, 1.2, 8, 20, 7)
    rng = np.random.default_rng(4)
    cases = {
        "flat": np.full((480, 640), 128, np.uint8),                       # no corners at all -> n == 0
        "noise": rng.integers(0, 256, (480, 640), dtype=np.uint8),        # corners everywhere, heavy quadtree
        "weak": (128 + rng.integers(-6, 7, (480, 640))).astype(np.uint8),  # only minThFAST corners
        "saturated": np.where(rng.random((480, 640)) < 0.5, 0, 255).astype(np.uint8),
        "one_corner": np.pad(np.full((100, 100), 255, np.uint8), ((190, 190), (270, 270))),
        "checker": (((np.indices((480, 640)).sum(axis=0) // 17) % 2) * 200 + 20).astype(np.uint8),
    }
    for name, img in cases.items():
        mono, kps, desc = ex(img, None, (0, 1000))
        rmono, rk, rd = ref.extract(img, (0, 1000))
        assert mono == rmono, name
        _assert_same(pkg, kps, desc, rk, rd, name)
    assert len(ex(cases["flat"], None, (0, 0))[1]) == 0
    ex.close()


def test_empty_and_error_paths(pkg):
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=2)
    mono, kps, desc = ex(np.zeros((0, 0), np.uint8))
    assert mono == -1 and len(kps) == 0  # reference: `return -1` on an empty image (ORBextractor.cc:1231-1232)
    lib = pkg.load_library()
    n = __import__("ctypes").c_int(7)
    assert lib.orbx_extract(ex._h, None, 640, 480, 640, 0, 0, None, None, 0, __import__("ctypes").byref(n)) == -1 and n.value == 0
    with pytest.raises(pkg.OrbxError, match="smaller than one 35-px FAST cell"):
        ex(np.zeros((100, 100), np.uint8))
    with pytest.raises(pkg.OrbxError, match="max_batch"):
        ex.extract_batch(np.zeros((3, 480, 640), np.uint8))
    # portrait images with W / H < 0.5: DistributeOctTree would start from round(W / H) = 0 root nodes (the reference divides by
    # zero and indexes an empty vector there) -> a clean error
    with pytest.raises(pkg.OrbxError, match="aspect ratio"):
        pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=236, max_height=483)(np.zeros((483, 236), np.uint8))
    # an image larger than the handle was created for is refused (the per-level capacities were fixed from max_width / max_height)
    with pytest.raises(pkg.OrbxError, match="max_width x max_height"):
        ex(np.zeros((480, 700), np.uint8))
    with pytest.raises(pkg.OrbxError, match="trig_mode"):
        pkg.ORBextractor(1000, 1.2, 8, 20, 7, trig_mode=2)
    img = pkg.synth_frame(1, 640, 480)
    # an extraction still works after the errors, and a smaller image re-derives the geometry
    assert len(ex(img, None, (0, 0))[1]) >= 1000
    assert len(ex(img[:300, :400], None, (0, 0))[1]) > 0
    ex.close()


def test_wide_image_small_quota(pkg, orc):
    """DistributeOctTree starts from nIni = round(W / H) root nodes and splits each of them into up to 4 before it looks at the
    quota: a very wide image with few features returns up to 4 * nIni keypoints per level, far more than N + 3."""
    img = pkg.synth_frame(77, 943, 137)
    ex = pkg.ORBextractor(96, 1.25, 4, 8, 2, max_width=943, max_height=137)
    ref = orc.Oracle(96, 1.25, 4, 8, 2)
    mono, kps, desc = ex(img, None, (764, 1777))
    rmono, rk, rd = ref.extract(img, (764, 1777))
    assert mono == rmono
    _assert_same(pkg, kps, desc, rk, rd, "943x137, 96 features")
    assert len(kps) > 96 + 4 * 3  # more than sum(N_l + 3)
    assert ex.capacity >= len(kps)
    ex.close()


def test_level_quota_beyond_lds(pkg, orc):
    """A level quota whose list does not fit in LDS (~3500 nodes): the quadtree kernels keep their node arrays in global scratch."""
    img = np.random.default_rng(3).integers(0, 256, (600, 800), dtype=np.uint8)
    for nf, nlevels in ((5544, 1), (9000, 2)):
        ex = pkg.ORBextractor(nf, 1.25, nlevels, 29, 3, max_width=800, max_height=600)
        mono, kps, desc = ex(img, None, (85, 936))
        rmono, rk, rd = orc.Oracle(nf, 1.25, nlevels, 29, 3).extract(img, (85, 936))
        assert mono == rmono
        _assert_same(pkg, kps, desc, rk, rd, f"nfeatures {nf} on {nlevels} level(s)")
        assert len(kps) > 4000
        ex.close()


def test_batch_handle_with_large_level_quota(pkg, orc):
    """A batch of 17 .. 128 quadtree instances takes the multi-wave kernel, whose 16 KB LDS key buffer once leaked into the serial
    kernel's launch behind it: with ~2700 features per level (few levels, many features) that launch went beyond 160 KB of LDS with no
    node scratch allocated and the call failed with "invalid argument" (found by STRESS_BATCH=20 tests/tools/stress_parity.py)."""
    rng = np.random.default_rng(68)
    for (w, h, nf, nlev, scale, ini, mn) in [(761, 541, 5457, 3, 1.25, 19, 10), (625, 727, 4543, 4, 2.0, 30, 23)]:
        img = np.kron(rng.integers(0, 256, (h // 16 + 1, w // 16 + 1), dtype=np.uint8), np.ones((16, 16), np.uint8))[:h, :w].copy()
        imgs = np.stack([np.roll(img, 17 * i, axis=1) for i in range(20)])
        ex = pkg.ORBextractor(nf, scale, nlev, ini, mn, max_width=w, max_height=h, max_batch=20)
        n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
        ref = orc.Oracle(nf, scale, nlev, ini, mn)
        for i in (0, 19):
            rmono, rk, rd = ref.extract(imgs[i], (0, 0))
            assert mono[i] == rmono
            _assert_same(pkg, kps[i][: n[i]], desc[i][: n[i]], rk, rd, f"{w}x{h} nf={nf} frame {i}")
        ex.close()


@pytest.mark.parametrize("pitch_pad,offset", [(4, 0), (7, 0), (0, 3), (20, 4)])
def test_device_batch_with_unaligned_rows(pkg, orc, pitch_pad, offset):
    """orbx_extract_batch_device on frames whose rows are not 16-byte aligned (a caller's padded or offset device images): k_pyr_fast
    stages level 0 with 4-byte or single-byte loads - on a batch handle also the margin bands of the edge tiles.  Bytes equal the oracle."""
    import torch
    B, w, h = 20, 640, 480
    pitch = w + pitch_pad
    frames = np.stack([pkg.synth_frame(300 + i, w, h) for i in range(B)])
    host = np.zeros((B * h * pitch + 64,), np.uint8)
    view = host[offset: offset + B * h * pitch].reshape(B, h, pitch)
    view[:, :, :w] = frames
    view[:, :, w:] = 255  # (the padding must not leak into the result)
    dev = torch.from_numpy(host).cuda()
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    ex.extract_batch_device(dev.data_ptr() + offset, B, w, h, pitch, pitch * h, (0, 1000))
    n, mono, kps, desc = ex.download_outputs(B)
    ref = orc.Oracle(1000, 1.2, 8, 20, 7)
    for f in (0, 7, B - 1):
        rmono, rk, rd = ref.extract(frames[f], (0, 1000))
        assert mono[f] == rmono
        _assert_same(pkg, kps[f][: n[f]], desc[f][: n[f]], rk, rd, f"pitch {pitch} offset {offset} frame {f}")
    ex.close()


def test_strided_and_unaligned_input(pkg, orc):
    base = np.zeros((480, 643 + 5), np.uint8)
    img = pkg.synth_frame(8, 640, 480)
    base[:, 3:643] = img
    view = base[:, 3:643]  # row stride 648, start address offset by 3 -> byte-granular staging path
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7)
    lib = pkg.load_library()
    import ctypes
    kps = np.zeros(ex.capacity, pkg.KEYPOINT_DTYPE)
    desc = np.zeros((ex.capacity, 32), np.uint8)
    n = ctypes.c_int()
    mono = lib.orbx_extract(ex._h, view.ctypes.data, 640, 480, view.strides[0], 0, 1000, kps.ctypes.data, desc.ctypes.data,
                            ex.capacity, ctypes.byref(n))
    rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract(img, (0, 1000))
    assert mono == rmono
    _assert_same(pkg, kps[: n.value], desc[: n.value], rk, rd, "strided input")
    ex.close()


def test_odd_sizes(pkg, orc):
    for (w, h) in [(641, 479), (333, 257), (1241, 376), (1024, 768)]:
        img = pkg.synth_frame(50, w, h)
        ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h)
        mono, kps, desc = ex(img, None, (0, 1000))
        rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract(img, (0, 1000))
        assert mono == rmono
        _assert_same(pkg, kps, desc, rk, rd, f"{w}x{h}")
        ex.close()


def test_batch_parity_and_composition_independence(pkg, orc):
    B = 48
    imgs = np.stack([pkg.synth_frame(1000 + i, 640, 480) for i in range(B)])
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 1000))
    rc, rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract_batch(imgs, (0, 1000), threads=8, want_outputs=True)
    assert np.array_equal(n, rc) and np.array_equal(mono, rmono)
    for f in range(B):
        assert kps[f][: n[f]].tobytes() == rk[f][: n[f]].tobytes(), f"frame {f}"
        assert desc[f][: n[f]].tobytes() == rd[f][: n[f]].tobytes(), f"frame {f}"
    # a frame's result does not depend on its batch slot, nor on the previous batch's contents
    perm = np.random.default_rng(0).permutation(B)
    n2, mono2, kps2, desc2 = ex.extract_batch(imgs[perm], (0, 1000))
    for i, f in enumerate(perm):
        assert n2[i] == n[f] and kps2[i][: n2[i]].tobytes() == kps[f][: n[f]].tobytes()
        assert desc2[i][: n2[i]].tobytes() == desc[f][: n[f]].tobytes()
    ex.close()


@pytest.mark.parametrize("scale,nlevels,size", [(1.1, 6, (640, 480)), (1.33, 5, (752, 480)), (1.7, 4, (640, 480)), (2.0, 4, (800, 600)),
                                                (2.4, 3, (800, 600)), (3.0, 3, (1024, 768)), (3.4, 2, (640, 480)), (1.2, 8, (1001, 701))])
def test_batch_resize_over_scale_factors(pkg, orc, scale, nlevels, size):
    """The FUSED pyramid step of the large-batch schedule (k_pyr_fast; single frames build their pyramid with another kernel) over scale
    factors: since round 5 the horizontal pass of cv::resize reads 8-byte windows (two pixels' taps) and runs on v_perm_b32 /
    v_dot2_u32_u16 for scale factors up to 3, and falls back to the byte reads beyond.  Every pyramid level byte for byte, then the
    keypoints and descriptors of every frame."""
    B = 20
    w, h = size
    imgs = np.stack([pkg.synth_frame(4000 + i, w, h) for i in range(B)])
    ex = pkg.ORBextractor(800, scale, nlevels, 20, 7, max_width=w, max_height=h, max_batch=B)
    ref = orc.Oracle(800, scale, nlevels, 20, 7)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
    for f in (0, B - 1):
        ref.extract(imgs[f], (0, 0))
        for lvl in range(nlevels):
            assert ex.level_size(lvl) == ref.level_size(lvl)
            assert np.array_equal(ex.pyramid_level(f, lvl), ref.level(lvl)), f"scale {scale}: frame {f} level {lvl}"
    rc, rmono, rk, rd = ref.extract_batch(imgs, (0, 0), threads=8, want_outputs=True)
    assert np.array_equal(n, rc) and np.array_equal(mono, rmono)
    for f in range(B):
        assert kps[f][: n[f]].tobytes() == rk[f][: n[f]].tobytes(), f"scale {scale}: frame {f}"
        assert desc[f][: n[f]].tobytes() == rd[f][: n[f]].tobytes(), f"scale {scale}: frame {f}"
    ex.close()


def test_host_batch_in_sub_batches(pkg, orc):
    """Host-pointer batches of >= 256 frames are staged and processed in sub-batches (copy of the next one under the kernels of
    this one): 300 frames (sub-batches of 152 + 148), all 300 against the oracle's counts, every 7th frame byte for byte, and
    the same frames again through the device-resident call (no sub-batches)."""
    B = 300
    imgs = np.stack([pkg.synth_frame(5000 + i, 640, 480) for i in range(B)])
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    n, mono, kps, desc = ex.extract_batch(imgs, (100, 500))
    sel = np.arange(0, B, 7)
    rc, rmono, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract_batch(imgs[sel], (100, 500), threads=8, want_outputs=True)
    assert np.array_equal(n[sel], rc) and np.array_equal(mono[sel], rmono)
    for j, f in enumerate(sel):
        assert kps[f][: n[f]].tobytes() == rk[j][: rc[j]].tobytes() and desc[f][: n[f]].tobytes() == rd[j][: rc[j]].tobytes(), f"frame {f}"
    import torch
    d = torch.from_numpy(imgs).cuda()
    ex.extract_batch_device(d.data_ptr(), B, 640, 480, 640, 640 * 480, (100, 500))
    n2, mono2, kps2, desc2 = ex.download_outputs(B)
    assert np.array_equal(n, n2) and np.array_equal(mono, mono2)
    for f in range(B):
        assert kps[f][: n[f]].tobytes() == kps2[f][: n[f]].tobytes() and desc[f][: n[f]].tobytes() == desc2[f][: n[f]].tobytes(), f"frame {f}"
    ex.close()


def test_small_batch_graph_cache_alternating_sources(pkg, orc):
    """A single-frame handle replays a captured launch graph per (source buffer, image size, lapping area); a caller that alternates camera
    buffers or lapping areas must hit the handle's small cache of graphs (4 entries) and get every frame's own result - also after more
    distinct keys than the cache holds."""
    import torch
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=1)
    imgs = [pkg.synth_frame(6000 + i, 640, 480) for i in range(6)]
    dev = [torch.from_numpy(im).cuda() for im in imgs]
    ref = orc.Oracle(1000, 1.2, 8, 20, 7)
    want = {}
    for i, im in enumerate(imgs):
        for lap in ((0, 1000), (100, 300)):
            want[(i, lap)] = ref.extract(im, lap)
    order = [(0, (0, 1000)), (1, (0, 1000)), (0, (0, 1000)), (1, (100, 300)), (0, (100, 300)), (1, (0, 1000)),   # 4 keys, alternating
             (2, (0, 1000)), (3, (0, 1000)), (4, (0, 1000)), (5, (0, 1000)), (0, (0, 1000)), (2, (100, 300)), (1, (0, 1000))]  # more than 4
    for i, lap in order * 2:
        ex.extract_batch_device(dev[i].data_ptr(), 1, 640, 480, 640, 640 * 480, lap)
        n, mono, kps, desc = ex.download_outputs(1)
        rmono, rk, rd = want[(i, lap)]
        assert mono[0] == rmono and n[0] == len(rk) and kps[0][: n[0]].tobytes() == rk.tobytes() and desc[0][: n[0]].tobytes() == rd.tobytes(), (i, lap)
    ex.close()


def _same_valid_rows(a, b):
    (n, mono, kps, desc), (n2, mono2, kps2, desc2) = a, b
    assert np.array_equal(n, n2) and np.array_equal(mono, mono2)
    for f in range(len(n)):
        assert kps[f][: n[f]].tobytes() == kps2[f][: n[f]].tobytes() and desc[f][: n[f]].tobytes() == desc2[f][: n[f]].tobytes(), f"frame {f}"


def test_host_batch_page_locked_and_reused_arrays(pkg):
    """The host pipeline's four routes give the same bytes: pageable frames (copy thread) / page-locked frames (asynchronous copies) x
    results through the handle's staging / written by the device straight into page-locked result arrays; result arrays are reused
    between calls (rows below n[f] are overwritten, the call does not depend on what they held)."""
    B, w, h = 200, 752, 480  # (pitch 768 != width: packed rows go through the unpack kernel)
    imgs = np.stack([pkg.synth_frame(7000 + i, w, h) for i in range(B)])
    ex = pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    ref = ex.extract_batch(imgs, (0, 0))
    pimgs = pkg.host_array(imgs.shape, np.uint8)
    pimgs[...] = imgs
    out = ex.output_arrays(B)
    for a in out:
        a.view(np.uint8)[...] = 0xA5
    _same_valid_rows(ref, ex.extract_batch(pimgs, (0, 0), out=out))
    pout = ex.output_arrays(B, page_locked=True)
    _same_valid_rows(ref, ex.extract_batch(pimgs, (0, 0), out=pout))
    _same_valid_rows(ref, ex.extract_batch(imgs, (0, 0), out=pout))
    ex.extract_batch(imgs[::-1].copy(), (0, 0), out=pout)  # (a different batch in between)
    _same_valid_rows(ref, ex.extract_batch(pimgs, (0, 0), out=pout))
    ex.close()


def test_submit_wait_two_batches_in_flight(pkg):
    """orbx_extract_batch_submit / _wait (round 5): two batches in flight on ONE handle - submit b0, submit b1, then wait / submit in turn -
    so that a batch's frames cross the link (into the second staging buffer) under the kernels and the copy-out of the one before.  Every
    frame's result is its own, byte for byte what the synchronous call gives: through staging (pageable result arrays) and written by
    the device (page-locked ones), different batches in flight, different batch sizes, a small (graph-schedule) batch, two handles.  A
    third submit, any other extraction call on a busy handle, pageable frames and a wait with nothing pending are refused."""
    B, w, h = 160, 640, 480
    batches = [np.stack([pkg.synth_frame(9000 + 1000 * k + i, w, h) for i in range(B)]) for k in range(3)]
    a = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    b = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    refs = [[x.copy() for x in a.extract_batch(img, (0, 0))] for img in batches]
    pims = []
    for img in batches:
        p_ = pkg.host_array(img.shape, np.uint8); p_[...] = img; pims.append(p_)
    outs = [a.output_arrays(B, page_locked=True), a.output_arrays(B), a.output_arrays(B, page_locked=True)]
    a.extract_batch_submit(pims[0], (0, 0), out=outs[0])
    a.extract_batch_submit(pims[1], (0, 0), out=outs[1])
    with pytest.raises(pkg.OrbxError, match="two submitted batches"):
        a.extract_batch_submit(pims[2], (0, 0), out=outs[2])
    with pytest.raises(pkg.OrbxError, match="pending"):
        a.extract_batch(batches[2], (0, 0))
    with pytest.raises(pkg.OrbxError, match="pending"):
        a(batches[2][0], None, (0, 0))
    _same_valid_rows(refs[0], a.extract_batch_wait())
    a.extract_batch_submit(pims[2], (0, 0), out=outs[2])
    _same_valid_rows(refs[1], a.extract_batch_wait())
    _same_valid_rows(refs[2], a.extract_batch_wait())
    # steady state: wait / submit in turn, the arrays reused; every other batch is shorter (other sub-batch sizes) and starts inside its arrays
    queue = []
    for k in range(9):
        if len(queue) == 2:
            _same_valid_rows(queue.pop(0), a.extract_batch_wait())
        lo = 8 if k & 1 else 0
        a.extract_batch_submit(pims[k % 3][lo:], (0, 0), out=tuple(x[lo:] for x in outs[k % 3]))
        queue.append([x[lo:] for x in refs[k % 3]])
    while queue:
        _same_valid_rows(queue.pop(0), a.extract_batch_wait())
    with pytest.raises(pkg.OrbxError, match="no submitted batch"):
        a.extract_batch_wait()
    with pytest.raises(pkg.OrbxError, match="page-locked"):
        a.extract_batch_submit(batches[0], (0, 0), out=outs[0])          # pageable frames need the calling thread
    _same_valid_rows(refs[0], a.extract_batch(batches[0], (0, 0)))        # the handle is usable after the refusals
    # two handles, each with batches in flight, from one thread
    outb = [b.output_arrays(B, page_locked=True), b.output_arrays(B)]
    a.extract_batch_submit(pims[0], (0, 0), out=outs[0]); b.extract_batch_submit(pims[1], (0, 0), out=outb[0])
    a.extract_batch_submit(pims[2], (0, 0), out=outs[2]); b.extract_batch_submit(pims[0], (0, 0), out=outb[1])
    _same_valid_rows(refs[1], b.extract_batch_wait()); _same_valid_rows(refs[0], a.extract_batch_wait())
    _same_valid_rows(refs[2], a.extract_batch_wait()); _same_valid_rows(refs[0], b.extract_batch_wait())
    # a small batch: alone only
    small = pims[0][:5]
    sref = [x.copy() for x in a.extract_batch(np.ascontiguousarray(small), (0, 0))]
    a.extract_batch_submit(small, (0, 0), out=a.output_arrays(5))
    with pytest.raises(pkg.OrbxError, match="pending"):
        a.extract_batch_submit(pims[1], (0, 0), out=outs[1])
    _same_valid_rows(sref, a.extract_batch_wait())
    a.extract_batch_submit(pims[1], (0, 0), out=outs[1])
    with pytest.raises(pkg.OrbxError, match="pending"):
        a.extract_batch_submit(small, (0, 0), out=a.output_arrays(5))
    _same_valid_rows(refs[1], a.extract_batch_wait())
    b.extract_batch_submit(pims[1], (0, 0), out=outb[0])                  # batches nobody waits for: close() drains them
    b.extract_batch_submit(pims[2], (0, 0), out=outb[1])
    b.close(); a.close()


def test_host_batch_from_registered_caller_memory(pkg):
    """orbx_host_register page-locks a buffer the caller already owns (a numpy array here, cv::Mat data in the reference): the batch call
    then takes its asynchronous route from it - same bytes out as from the pageable array, before and after unregistering; a null
    range is an error, not a crash."""
    B = 160
    imgs = np.stack([pkg.synth_frame(500 + i, 640, 480) for i in range(B)])
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    ref = [a.copy() for a in ex.extract_batch(imgs, (0, 0))]
    lib = pkg.load_library()
    with pkg.host_registered(imgs) as locked:
        got = ex.extract_batch(locked, (0, 0))
        for a, b in zip(got, ref):
            assert a.tobytes() == b.tobytes()
    got = ex.extract_batch(imgs, (0, 0))
    for a, b in zip(got, ref):
        assert a.tobytes() == b.tobytes()
    assert lib.orbx_host_register(None, 16) != 0 and lib.orbx_host_register(imgs.ctypes.data, 0) != 0 and lib.orbx_host_unregister(None) == 0
    ex.close()


def test_sharded_host_batch_two_handles(pkg):
    """orbx_extract_batch_sharded: frames dealt over two handles (here on one device; one per GPU on a node) in blocks of 1 and of 8
    frames, more frames than the handles hold at once, a ragged tail - every frame's result at its index, equal to one handle's."""
    w, h, B = 640, 480, 77
    imgs = np.stack([pkg.synth_frame(8000 + i, w, h) for i in range(B)])
    one = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=B)
    ref = one.extract_batch(imgs, (50, 300))
    one.close()
    for max_batch, block in ((64, 1), (24, 8), (16, 3)):  # 16: each handle's calls take the small-batch (graph) route
        exs = [pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=max_batch) for _ in range(2)]
        got = pkg.extract_batch_sharded(exs, imgs, (50, 300), block=block)
        _same_valid_rows(ref, got)
        for e in exs:
            e.close()
    # error behaviour: a handle listed twice, a block larger than max_batch
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=w, max_height=h, max_batch=8)
    with pytest.raises(pkg.OrbxError):
        pkg.extract_batch_sharded([ex, ex], imgs[:8], (0, 0))
    with pytest.raises(pkg.OrbxError):
        pkg.extract_batch_sharded([ex], imgs[:32], (0, 0), block=16)
    ex.close()


def test_full_size_batch_properties(pkg, orc):
    """BASELINE configs[1] at full size (1024 frames, device-resident): size-independent properties."""
    import torch
    B = 1024
    base = np.stack([pkg.synth_frame(i, 640, 480) for i in range(64)])
    host = np.concatenate([base] * (B // 64))  # frame f == base[f % 64]
    dev = torch.from_numpy(host).cuda()
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
    ex.extract_batch_device(dev.data_ptr(), B, 640, 480, 640, 640 * 480, (0, 1000))
    n, mono, kps, desc = ex.download_outputs(B)
    ex.extract_batch_device(dev.data_ptr(), B, 640, 480, 640, 640 * 480, (0, 1000))
    n2, mono2, kps2, desc2 = ex.download_outputs(B)
    assert np.array_equal(n, n2) and kps.tobytes() == kps2.tobytes() and desc.tobytes() == desc2.tobytes()  # idempotent
    assert np.all(n >= 1000) and np.all(n <= ex.capacity) and np.all(mono == 0)
    for f in range(64, B):  # periodic input -> periodic output (no cross-frame leakage at full occupancy)
        assert n[f] == n[f % 64]
    assert kps[64:128].tobytes() == kps[:64].tobytes() and desc[960:].tobytes() == desc[:64].tobytes()
    rc, _, rk, rd = orc.Oracle(1000, 1.2, 8, 20, 7).extract_batch(base[:16], (0, 1000), threads=8, want_outputs=True)
    for f in range(16):
        assert n[f + 512] == rc[f]
        assert kps[f + 512][: rc[f]].tobytes() == rk[f][: rc[f]].tobytes()
        assert desc[f + 512][: rc[f]].tobytes() == rd[f][: rc[f]].tobytes()
    # keypoint invariants: inside the image margin, octave/size consistent, angle range, level-major reversed order
    k = kps[0][: n[0]]
    sf = ex.GetScaleFactors()
    assert np.all((k["octave"] >= 0) & (k["octave"] < 8)) and np.all(k["class_id"] == -1)
    assert np.all((k["angle"] >= 0) & (k["angle"] <= 360))
    assert np.array_equal(k["size"], np.floor(31 * sf[k["octave"]]).astype(np.float32))
    assert np.all(k["x"] >= 19) and np.all(k["x"] <= 640) and np.all(k["y"] >= 19) and np.all(k["y"] <= 480)
    assert np.all(np.diff(k["octave"]) <= 0)
    ex.close()


def test_two_handles_concurrently(pkg, orc):
    """Reference runs left/right extractors in two threads (src/Frame.cc:124-127): handles are independent."""
    import threading
    imgs = [pkg.synth_frame(60, 752, 480), pkg.synth_frame(61, 752, 480)]
    exs = [pkg.ORBextractor(1200, 1.2, 8, 20, 7, max_width=752, max_height=480) for _ in range(2)]
    out = [None, None]

    def work(i):
        for _ in range(5):
            out[i] = exs[i](imgs[i], None, (0, 0))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        rmono, rk, rd = orc.Oracle(1200, 1.2, 8, 20, 7).extract(imgs[i], (0, 0))
        assert out[i][0] == rmono
        _assert_same(pkg, out[i][1], out[i][2], rk, rd, f"handle {i}")
    [e.close() for e in exs]


def test_two_threads_recapturing_all_the_time(pkg, orc):
    """Two single-frame handles in two threads while both keep re-capturing their launch graphs (the image size changes every call) and
    a third thread creates / destroys handles and matchers (allocations, synchronous copies): a thread in the default stream-capture mode
    may not make a synchronising call while ANY capture is open, so every entry point runs in relaxed mode.  Results equal the oracle."""
    import threading
    sizes = [(640, 480), (752, 480), (400, 300)]
    imgs = {(i, s): pkg.synth_frame(200 + 10 * i + k, *s) for i in range(2) for k, s in enumerate(sizes)}
    refs = {key: orc.Oracle(800, 1.2, 8, 20, 7).extract(im, (0, 0)) for key, im in imgs.items()}
    exs = [pkg.ORBextractor(800, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=1) for _ in range(2)]
    errors = []

    def work(i):
        try:
            for it in range(24):
                s = sizes[it % 3]
                mono, kps, desc = exs[i](imgs[(i, s)], None, (0, 0))
                rmono, rk, rd = refs[(i, s)]
                if mono != rmono or kps.tobytes() != rk.tobytes() or desc.tobytes() != rd.tobytes():
                    errors.append(f"thread {i} iteration {it}: result differs")
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {i}: {e}")

    def churn():
        try:
            for _ in range(12):
                ex = pkg.ORBextractor(500, 1.2, 8, 20, 7, max_width=320, max_height=240, max_batch=1)
                ex(pkg.synth_frame(5, 320, 240), None, (0, 0))
                m = pkg.ORBmatcher(2048, 2048)
                m.close(); ex.close()
        except Exception as e:  # noqa: BLE001
            errors.append(f"churn: {e}")

    th = [threading.Thread(target=work, args=(0,)), threading.Thread(target=work, args=(1,)), threading.Thread(target=churn)]
    [t.start() for t in th]
    [t.join() for t in th]
    [e.close() for e in exs]
    assert not errors, errors[:3]


def _gpu_check_subprocess(extra_env, arg_sets):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    for args in arg_sets:
        out = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_check.py")] + args, env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
        assert out.returncode == 0 and "ALL STAGES BIT-IDENTICAL" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("clcap", ["192", "8"])
def test_pyr_fast_wave_per_cell_kernel(pkg, clcap):
    """k_pyr_fast_cw (ORBX_PF_KERNEL=2: a wave per FAST cell, corners as an order-preserving list per cell) must give the bytes of the
    phase-organised kernel.  clcap 8: every cell with more than 8 corners takes the scanning path for NMS and emission."""
    _gpu_check_subprocess({"ORBX_PF_KERNEL": "2", "ORBX_PF_CLCAP": clcap},
                          (["640", "480", "1000", "2"], ["752", "480", "1200", "1"], ["512", "512", "1500", "20"]))


@pytest.mark.parametrize("env", [{"ORBX_PF_GQCAP": "64"}, {"ORBX_PF_GQCAP": "64", "ORBX_PF_LANE_ALLOC": "0"}, {}])
def test_pyr_fast_group_queue_overflow(pkg, env):
    """The pre-test's group queue of k_pyr_fast may hold fewer entries than the tile has groups (what keeps a fourth workgroup per
    CU resident at VGA); a round that queues more runs the ring test over every group of the tile.  ORBX_PF_GQCAP=64 sends nearly
    every tile down that path (also with the ballot-prefix queue slots of the profiling instance); the noise frames overflow the
    corner queue as well (NMS over every scored pixel) - with the default capacity too, on a batch handle.  Bytes equal the oracle."""
    _gpu_check_subprocess(env, (["640", "480", "1000", "40"], ["752", "480", "1200", "34", "noise"], ["370", "278", "800", "33", "saturated"]))


def test_pyr_fast_classic_tile_layout(pkg):
    """Edge tiles stage the image rows beyond their 3-px halo as margin bands in the score tile's bytes; a level whose bands do not
    fit keeps them in the pixel tile.  Tall cells over narrow tiles (few cells per group: small score tile) take that layout."""
    _gpu_check_subprocess({"ORBX_TILE_W": "40"}, (["640", "480", "1000", "40"], ["400", "500", "600", "33"]))


def test_quadtree_pool_compaction_path(pkg):
    """The serial quadtree formulation (k_quadtree_serial) recycles node ids by compacting its tombstoned list when the node
    pool runs low.  With the default pool (2 * quota + 64) that path is rare; force the serial kernel with a tiny pool in a
    subprocess (the switches are read once per process) and check full parity."""
    _gpu_check_subprocess({"ORBX_QT_KERNEL": "1", "ORBX_QT_POOL_SLACK": "12"},
                          (["640", "480", "1000", "2"], ["752", "480", "1200", "1"]))


@pytest.mark.parametrize("mw", ["0", "1"])
def test_quadtree_deferred_instances(pkg, mw):
    """Candidate sets above the capacity of the level-synchronous kernels are marked and taken by the serial kernel in a
    second launch.  At VGA no set reaches the real limit (8192 keys): lower it so that the large levels are deferred while
    the small ones stay on the parallel kernel, and check full parity of the mixed result.  ORBX_QT_MW: one wave per instance
    (k_quadtree, what large batches run) / a workgroup of 8 waves per instance (k_quadtree_mw, what small batches run)."""
    _gpu_check_subprocess({"ORBX_QT_MW": mw, "ORBX_QT_MAX_KEYS": "600", "ORBX_QT_MAX_KEYS2": "0"},
                          (["640", "480", "1000", "2"], ["752", "480", "1200", "1"]))


@pytest.mark.parametrize("mw", ["0", "1"])
def test_quadtree_second_pass_instances(pkg, mw):
    """Sets between 8192 and 49152 keys (the large levels of 1080p frames): k_quadtree takes them in a second launch with a larger
    run-start bitmap, k_quadtree_mw in its one launch; only what lies beyond goes to the serial kernel.  Lowered limits make all of
    them contribute to one result at VGA; the 1080p case runs the real limits."""
    _gpu_check_subprocess({"ORBX_QT_MW": mw, "ORBX_QT_MAX_KEYS": "500", "ORBX_QT_MAX_KEYS2": "1500"},
                          (["640", "480", "1000", "2"], ["752", "480", "1200", "1"]))
    _gpu_check_subprocess({"ORBX_QT_MW": mw}, (["1920", "1080", "4000", "2"],))


@pytest.mark.parametrize("mw", ["0", "1"])
def test_quadtree_kernels_on_other_shapes(pkg, mw):
    """Both level-synchronous kernels forced on shapes their default choice would not give them: a batch of 40 frames on the 8-wave
    kernel / single frames on the one-wave kernel, images with 1, 2, 3 and 5 root nodes (nIni = round(W / H): the 8-wave kernel groups
    up to four roots chunk-parallel, more on one wave), and the 8-wave kernel's in-place scatter limit (4096 keys) crossed at 1080p."""
    _gpu_check_subprocess({"ORBX_QT_MW": mw}, (["640", "480", "1000", "40"], ["752", "480", "1200", "1"], ["1000", "350", "1500", "2"],
                                               ["1500", "300", "1200", "2"], ["333", "257", "600", "3"]))
    if mw == "0":  # the one-wave kernel exists with 32 and with 8 key chunks in registers (chosen by the launch's largest quota)
        for kch in ("32", "8"):
            _gpu_check_subprocess({"ORBX_QT_MW": "0", "ORBX_QT_KCH": kch}, (["640", "480", "1000", "24"], ["752", "480", "2500", "2"]))


@pytest.mark.parametrize("env", [{"ORBX_BLUR_KERNEL": "2"}, {"ORBX_BLUR_KERNEL": "1"}, {"ORBX_BLUR_WAVES": "8"}, {"ORBX_BLUR_SEG": "8"},
                                 {"ORBX_BLUR_SEG": "8", "ORBX_BLUR_WAVES": "8"}])
def test_blur_fallback_kernels(pkg, env):
    """The GaussianBlur normally runs on the matrix cores (k_blur3, 4 waves x 4 row blocks per workgroup); taps that do not fit int8
    or unaligned buffers take the byte-SWAR kernel (k_blur2) or the generic one (k_blur), and k_blur3 has an 8-wave and an 8-block
    shape.  Force each in a subprocess (the switch is read once per process) and check every stage against the oracle - on a frame
    whose height is not a multiple of the 16-row blocks, too."""
    _gpu_check_subprocess(env, (["640", "480", "1000", "2"], ["333", "257", "600", "1"], ["752", "473", "900", "17"]))


@pytest.mark.parametrize("env,args", [({"ORBX_QT_PER_LEVEL": "1"}, ["640", "480", "1000", "3"]), ({"ORBX_QT_PER_LEVEL": "3"}, ["640", "480", "1000", "3"]),
                                      ({"ORBX_QT_PER_LEVEL": "4"}, ["752", "480", "1200", "2"]), ({"ORBX_EARLY_BLUR": "1"}, ["640", "480", "1000", "3"]),
                                      ({"ORBX_SUBBATCHES": "3"}, ["640", "480", "1000", "200"]), ({"ORBX_PF_LDS_FIT": "0"}, ["752", "480", "1200", "20"]),
                                      ({"ORBX_FRONT_PARTS": "2"}, ["640", "480", "1000", "300"]),
                                      ({"ORBX_QT_STREAMS": "2", "ORBX_QT_PER_LEVEL": "3"}, ["640", "480", "1000", "24"]),
                                      ({"ORBX_QT_STREAMS": "3", "ORBX_QT_PER_LEVEL": "2"}, ["752", "480", "1200", "5"]),
                                      ({}, ["512", "512", "1500", "648"]),
                                      ({"ORBX_PF_LANE_ALLOC": "0"}, ["640", "480", "1000", "24"]),
                                      ({"ORBX_FRONT_PARTS": "3", "ORBX_FRONT_STAGGER": "1"}, ["640", "480", "1000", "300"])])
def test_launch_schedules(pkg, env, args):
    """The launch sequence has measured alternatives behind environment switches (quadtree per level / per group of levels, blur
    of a level as soon as it exists, device batches cut into sub-batches, the k_pyr_fast chain cut into frame groups on their own
    streams, tile widths that ignore the 40 KB LDS limit of four workgroups per CU, the quadtree launch groups dealt over several
    streams - which is also what a large batch with many features per pixel does by itself, the last case): every one of them must
    give the same bytes."""
    _gpu_check_subprocess(env, (args,))


@pytest.mark.gpu
@pytest.mark.parametrize("trig_mode", [0, 1])
def test_trig_mode_parity(pkg, orc, trig_mode):
    """computeOrbDescriptor's cos / sin (reference src/ORBextractor.cc:108) as ::cos(double) narrowed (0) or as glibc's float
    overloads (1): the GPU evaluates include/orbx_trig.h, the oracle calls the real libm - descriptors bit-equal over 20 k keypoints."""
    imgs = np.stack([pkg.synth_frame(900 + i, 640, 480) for i in range(12)])
    ex = pkg.ORBextractor(1500, 1.2, 8, 20, 7, max_width=640, max_height=480, max_batch=12, trig_mode=trig_mode)
    n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
    ref = orc.Oracle(1500, 1.2, 8, 20, 7, trig_mode=trig_mode)
    rn, rmono, rk, rd = ref.extract_batch(imgs, (0, 0), threads=8, want_outputs=True)
    assert np.array_equal(n, rn) and n.sum() > 15000
    for f in range(len(imgs)):
        assert kps[f, : n[f]].tobytes() == rk[f, : n[f]].tobytes(), f
        assert desc[f, : n[f]].tobytes() == rd[f, : n[f]].tobytes(), f
    ex.close()


@pytest.mark.parametrize("env", [{"ORBX_GRAPH": "0"}, {"ORBX_GRAPH": "2"}, {"ORBX_GRAPH": "1"}, {"ORBX_SMALL_QT_BRANCHES": "2"},
                                 {"ORBX_SMALL_QT_BRANCHES": "4", "ORBX_GRAPH": "2"}, {"ORBX_SMALL_QT_BRANCHES": "4"},
                                 {"ORBX_SMALL_SPLIT": "0"}, {"ORBX_SMALL_SPLIT": "0", "ORBX_SMALL_QT_BRANCHES": "2"},
                                 {"ORBX_SMALL_BLUR_LEVELS": "0"}, {"ORBX_SMALL_SPLIT": "1", "ORBX_GRAPH": "2"}, {"ORBX_SMALL_PYR_CHAIN": "0"},
                                 {"ORBX_CHAIN_TILE": "64"}, {"ORBX_CHAIN_TILE": "16"}])
def test_small_batch_schedules(pkg, env):
    """Batches of up to 16 frames run a schedule of their own (narrow tiles, one quadtree launch behind the pyramid chain or - with
    ORBX_SMALL_QT_BRANCHES = 2 / 4 - groups of levels on branches of their own as soon as their candidates exist), captured once as a
    hipGraph and replayed per call (ORBX_GRAPH=1, default); 2 = that schedule without capture, 0 = the batch schedule.  By default the
    pyramid comes from a chain of resize-only launches and FAST of all levels / the blur of all levels are one launch each
    (ORBX_SMALL_SPLIT=0: the fused per-level chain, ORBX_SMALL_BLUR_LEVELS=0: a blur launch per level, ORBX_SMALL_PYR_CHAIN=0: a resize
    launch per level instead of the one-launch pyramid, ORBX_CHAIN_TILE: its tile size).  Every stage of all of them must be bit-identical
    to the oracle."""
    _gpu_check_subprocess(env, (["640", "480", "1000", "1"], ["752", "480", "1200", "3"], ["333", "257", "600", "2"]))


@pytest.mark.gpu
def test_graph_replay_and_recapture(pkg, orc):
    """One handle, many calls (how Frame::ExtractORB drives operator()): the captured graph is replayed while the call signature
    stays the same and re-captured when the image size, the lapping area or the batch size changes; results always equal the oracle."""
    ex = pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=752, max_height=480, max_batch=4)
    ref = orc.Oracle(1000, 1.2, 8, 20, 7)
    plan = [(640, 480, (0, 0)), (640, 480, (0, 0)), (640, 480, (0, 0)), (752, 480, (0, 0)), (752, 480, (100, 400)), (752, 480, (100, 400)),
            (640, 480, (0, 0)), (400, 300, (0, 1000))]
    for i, (w, h, lap) in enumerate(plan):
        img = pkg.synth_frame(40 + i, w, h)
        mono, kps, desc = ex(img, None, lap)
        rmono, rk, rd = ref.extract(img, lap)
        assert mono == rmono and kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes(), (i, w, h, lap)
    imgs = np.stack([pkg.synth_frame(70 + i, 640, 480) for i in range(3)])
    for _ in range(2):
        n, mono, kps, desc = ex.extract_batch(imgs, (0, 0))
        for f in range(3):
            rmono, rk, rd = ref.extract(imgs[f], (0, 0))
            assert n[f] == len(rk) and kps[f, : n[f]].tobytes() == rk.tobytes() and desc[f, : n[f]].tobytes() == rd.tobytes()
    # device-resident frames in six different buffers, visited round-robin: the per-level argument blocks of the one-launch FAST are
    # cached for four source buffers, so slots are replaced and graphs re-captured all the time
    import torch
    frames = [pkg.synth_frame(90 + i, 640, 480) for i in range(6)]
    dev = [torch.from_numpy(f[None]).to(f"cuda:{ex.device}") for f in frames]
    refs = [ref.extract(f, (0, 0)) for f in frames]
    for rnd in range(3):
        for i in (0, 1, 2, 3, 4, 5, 2, 0):
            ex.extract_batch_device(dev[i].data_ptr(), 1, 640, 480, 640, 640 * 480, (0, 0))
            n, mono, kps, desc = ex.download_outputs(1)
            rmono, rk, rd = refs[i]
            assert n[0] == len(rk) and mono[0] == rmono and kps[0, : n[0]].tobytes() == rk.tobytes() and desc[0, : n[0]].tobytes() == rd.tobytes(), (rnd, i)
    ex.close()
