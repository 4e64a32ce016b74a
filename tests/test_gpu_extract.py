"""GPU parity: HIP extractor (through the C-ABI) vs the CPU oracle, stage by stage.

Bar: bit-exact pyramids, blurred levels, FAST candidates, octree selection,
angles (float bits), descriptors and output keypoints.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd import ORBextractor
from orb_slam2_e_amd.synth import synth_frame

PARAMS = (2000, 1.2, 8, 20, 7)


def _kp_equal(a, b):
    assert len(a) == len(b)
    for f in a.dtype.names:
        assert np.array_equal(a[f].view(np.uint32) if a[f].dtype.kind == "f" else a[f],
                              b[f].view(np.uint32) if b[f].dtype.kind == "f" else b[f]), f"field {f} differs"


@pytest.fixture(scope="module")
def frame0():
    img = synth_frame(0)
    o = oracle.OrbOracle(*PARAMS)
    okps, odesc = o.extract(img)
    ex = ORBextractor(*PARAMS)
    kps, desc = ex(img)
    return img, o, okps, odesc, ex, kps, desc


def test_getters_match_oracle(frame0):
    _, o, _, _, ex, _, _ = frame0
    assert list(ex.features_per_level()) == o.features_per_level()
    assert np.array_equal(ex.GetScaleFactors(), np.array(o.scale_factors(), np.float32))
    assert ex.GetLevels() == 8


def test_pyramid_bit_exact(frame0):
    _, o, _, _, ex, _, _ = frame0
    for l in range(8):
        w, h, _ = o.level_dims(l)
        assert ex.level_size(l) == (w, h)
        ref = o.level_padded(l)[:, :w + 38]
        got = ex.pyramid_level(0, l, padded=True)
        assert np.array_equal(got, ref), f"padded pyramid level {l} differs"
        assert np.array_equal(ex.pyramid_level(0, l), o.level_image(l)[:, :w])


def test_blur_bit_exact(frame0):
    _, o, _, _, ex, _, _ = frame0
    for l in range(8):
        assert np.array_equal(ex.blurred_level(0, l), o.level_blurred(l)), f"blurred level {l} differs"


def test_fast_candidates_bit_exact(frame0):
    _, o, _, _, ex, _, _ = frame0
    for l in range(8):
        ref = o.level_cands(l)
        got = ex.level_candidates(0, l)
        assert len(got) == len(ref), f"level {l}: {len(got)} vs {len(ref)} candidates"
        assert np.array_equal(got[:, 0], ref["x"]) and np.array_equal(got[:, 1], ref["y"])
        assert np.array_equal(got[:, 2], ref["response"])


def test_level_keypoints_bit_exact(frame0):
    _, o, _, _, ex, _, _ = frame0
    for l in range(8):
        ref = o.level_kps(l)
        got = ex.level_keypoints(0, l)
        _kp_equal(got, ref)


def test_full_output_bit_exact(frame0):
    _, _, okps, odesc, _, kps, desc = frame0
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


def test_empty_image_leaves_outputs_untouched():
    ex = ORBextractor(*PARAMS)
    kps, desc = ex(np.zeros((0, 0), np.uint8))
    assert len(kps) == 0 and desc.shape == (0, 32)


def test_flat_image_yields_no_keypoints():
    ex = ORBextractor(*PARAMS)
    kps, desc = ex(np.full((480, 640), 77, np.uint8))
    assert len(kps) == 0


def test_too_small_image_is_an_error():
    from orb_slam2_e_amd import OrbxError
    ex = ORBextractor(*PARAMS)
    with pytest.raises(OrbxError):
        ex(np.zeros((100, 100), np.uint8))


@pytest.mark.parametrize("shape,params", [
    ((480, 640), (1000, 1.2, 8, 20, 7)),      # TUM settings (TUM1.yaml)
    ((375, 1242), (2000, 1.2, 8, 20, 7)),     # KITTI-shaped: 4 octree roots at level 0
    ((300, 400), (500, 1.5, 4, 30, 10)),
    ((480, 640), (4000, 1.2, 8, 20, 7)),      # mpIniORBextractor uses 2*nFeatures
    ((480, 752), (1000, 1.2, 8, 20, 7)),      # EuRoC-shaped
    ((487, 645), (1500, 1.2, 8, 20, 7)),      # odd sizes: every cell / row start at another byte alignment
    ((251, 333), (700, 1.3, 5, 15, 5)),
    ((376, 1241), (2000, 1.2, 8, 12, 7)),     # KITTI 00-02 size (Examples/Stereo/KITTI00-02.yaml), odd width
    ((480, 640), (5000, 1.2, 8, 20, 7)),      # level-0 quota 1085: node state beyond 1024 leaves
    ((600, 800), (3000, 1.5, 3, 15, 5)),      # level-0 quota 1421
    # THIS FORK's own settings (20 of its launch files: roslaunch/sHamlyn*.yaml, sHCULB_*.yaml, sEndoscope*.yaml, Examples/Stereo/EuRoC.yaml):
    # 1200 features, scale 1.1, 6 levels, FAST 24 / 7 -- at the sizes their principal points suggest
    ((360, 640), (1200, 1.1, 6, 24, 7)),      # Hamlyn (cx 327.9, cy 165.5)
    ((480, 752), (1200, 1.1, 6, 24, 7)),      # EuRoC stereo as the fork sets it
    ((1080, 1440), (1200, 1.1, 6, 24, 7)),    # HCULB (cx 770.2, cy 530.7)
    ((576, 720), (1200, 1.1, 6, 24, 7)),      # endoscope, PAL frame
    ((480, 640), (1200, 1.1, 6, 12, 7)),      # the two launch files with lower initial thresholds
    ((480, 640), (1200, 1.1, 6, 9, 4)),
])
def test_other_shapes_and_params(shape, params):
    img = synth_frame(3, w=shape[1], h=shape[0])
    o = oracle.OrbOracle(*params)
    okps, odesc = o.extract(img)
    ex = ORBextractor(*params)
    kps, desc = ex(img)
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


@pytest.mark.parametrize("shape,params", [
    ((760, 700), (400, 2.5, 3, 20, 7)),     # scale 2.5: a dword group's source columns span > 8 bytes -> no resize fast path
    ((410, 333), (600, 3.0, 2, 15, 5)),
    ((97 * 4 + 1, 251), (500, 1.2, 6, 20, 7)),   # w = 4k + 1: a partial interior group next to the border at every level
    ((96 * 3, 26 * 7), (500, 1.2, 4, 20, 7)),    # level 0 exactly 3 x 7 blur strips
    ((96 * 3 + 1, 26 * 7 + 1), (500, 1.2, 4, 20, 7)),   # ... and one pixel more in both directions
    ((463, 431), (300, 1.44, 4, 12, 4)),
])
def test_pyramid_and_blur_paths(shape, params):
    """Every level of the padded pyramid and of the blurred pyramid, bit for bit, on shapes that hit the resize
    kernel's interior / border split (and its fallback) and the blur's strip and tile edges."""
    w, h = shape
    img = synth_frame(5, w, h)
    o = oracle.OrbOracle(*params)
    o.extract(img)
    ex = ORBextractor(*params)
    ex(img)
    for l in range(params[2]):
        lw, lh, _ = o.level_dims(l)
        assert np.array_equal(ex.pyramid_level(0, l, padded=True), o.level_padded(l)[:, :lw + 38]), f"padded level {l}"
        assert np.array_equal(ex.blurred_level(0, l), o.level_blurred(l)), f"blurred level {l}"


@pytest.mark.parametrize("shape,params", [
    ((120, 86), (300, 1.2, 1, 20, 7)),      # one column of 57-px-wide cells: the 80-byte tile (32 lanes per tile row)
    ((88, 130), (300, 1.2, 1, 20, 7)),      # one row of 59-px-tall cells: a second chunk of tile rows
    ((88, 87), (200, 1.2, 1, 12, 5)),       # a single 58 x 59 cell
    ((200, 150), (400, 1.2, 3, 20, 7)),     # 45-px cells: the 64-byte tile
    ((140, 164), (400, 1.1, 2, 20, 7)),     # 44-px cells: the widest the 52-byte tile takes
])
def test_every_tile_class_of_the_fast_kernel(shape, params):
    """k_fast_cells has three instantiations (tile row 52 / 64 / 80 bytes) chosen by the widest cell, and its tile loader
    works in chunks of 44 (22) rows: small images with one or two cells per row reach the wide and tall cases."""
    for seed, kind in ((5, 0), (6, 1)):
        img = synth_frame(seed, w=shape[1], h=shape[0]) if kind == 0 else \
            np.random.default_rng(seed).integers(0, 256, shape, dtype=np.uint8)
        o = oracle.OrbOracle(*params)
        okps, odesc = o.extract(img)
        ex = ORBextractor(*params)
        kps, desc = ex(img)
        assert len(okps) > 0
        _kp_equal(kps, okps)
        assert np.array_equal(desc, odesc)
        for l in range(params[2]):
            ref, got = o.level_cands(l), ex.level_candidates(0, l)
            assert len(got) == len(ref), f"level {l}: {len(got)} vs {len(ref)} candidates"
            assert np.array_equal(got[:, 0], ref["x"]) and np.array_equal(got[:, 1], ref["y"])
            assert np.array_equal(got[:, 2], ref["response"])


def test_sparse_image_threshold_fallback_and_short_levels():
    # few weak corners: exercises the minThFAST fallback and levels below quota
    rng = np.random.default_rng(5)
    img = np.full((480, 640), 100, np.uint8)
    for _ in range(40):
        x, y = rng.integers(30, 600), rng.integers(30, 440)
        img[y:y + 12, x:x + 12] = 100 + rng.integers(8, 40)
    o = oracle.OrbOracle(*PARAMS)
    okps, odesc = o.extract(img)
    ex = ORBextractor(*PARAMS)
    kps, desc = ex(img)
    assert 0 < len(okps) < 2000
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


def test_dense_texture_every_cell_full():
    # white noise: almost every pixel passes the pre-test, cells emit their maximum number of candidates
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (480, 640), dtype=np.uint8)
    img[100:200, 100:300] = rng.integers(90, 110, (100, 200), dtype=np.uint8)   # and one low-contrast region
    o = oracle.OrbOracle(*PARAMS)
    okps, odesc = o.extract(img)
    ex = ORBextractor(*PARAMS)
    kps, desc = ex(img)
    for l in range(8):
        ref = o.level_cands(l)
        got = ex.level_candidates(0, l)
        assert len(got) == len(ref) and len(ref) > 100, f"level {l}: {len(got)} vs {len(ref)} candidates"
        assert np.array_equal(got[:, 0], ref["x"]) and np.array_equal(got[:, 1], ref["y"]) and np.array_equal(got[:, 2], ref["response"])
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


def test_batch_matches_single_and_oracle():
    B = 6
    imgs = np.stack([synth_frame(10 + k) for k in range(B)])
    ex = ORBextractor(*PARAMS)
    ex.extract_batch(imgs)
    o = oracle.OrbOracle(*PARAMS)
    for f in range(B):
        kps, desc = ex.download(f)
        okps, odesc = o.extract(imgs[f])
        _kp_equal(kps, okps)
        assert np.array_equal(desc, odesc)


def test_blur_variant_1():
    img = synth_frame(1)
    o = oracle.OrbOracle(*PARAMS)
    o.set_blur_taps([18, 34, 49, 55, 49, 34, 18])
    okps, odesc = o.extract(img)
    ex = ORBextractor(*PARAMS, blur_variant=1)
    kps, desc = ex(img)
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


def test_trig_variant_1():
    """cos / sin of the keypoint angle as the rounded double-precision values (the default is the C library's cosf / sinf, which
    every other test here exercises): still bit-exact against the oracle evaluated the same way."""
    descs = []
    for v in (0, 1):
        n_diff = 0
        for k in (1, 2, 3):
            img = synth_frame(k)
            o = oracle.OrbOracle(*PARAMS)
            o.set_trig_variant(v)
            okps, odesc = o.extract(img)
            ex = ORBextractor(*PARAMS, trig_variant=v)
            kps, desc = ex(img)
            _kp_equal(kps, okps)
            assert np.array_equal(desc, odesc)
            descs.append(desc)
    # the two variants give the same keypoints and (nearly) the same descriptors: one ulp in a or b rarely moves a sampling point
    same = sum(int((a != b).any(axis=1).sum()) for a, b in zip(descs[:3], descs[3:]))
    assert same <= 6, same


def test_download_batch_equals_per_frame_download():
    imgs = np.stack([synth_frame(20 + k) for k in range(3)])
    ex = ORBextractor(*PARAMS)
    ex.extract_batch(imgs)
    kps, desc, cnt = ex.download_batch()
    for f in range(3):
        k1, d1 = ex.download(f)
        assert cnt[f] == len(k1)
        _kp_equal(kps[f, :cnt[f]], k1)
        assert np.array_equal(desc[f, :cnt[f]], d1)


def test_random_sizes_parameters_and_contents():
    """Differential sweep (tests/fuzz_extract.py runs the same loop for hundreds of cases): random image sizes,
    pyramid depths / scale factors / quotas / thresholds and four kinds of content."""
    rng = np.random.default_rng(2026)
    done = 0
    while done < 40:
        w = int(rng.integers(220, 900)); h = int(rng.integers(180, 700))
        nlev = int(rng.integers(3, 9)); sf = float(rng.choice([1.1, 1.2, 1.25, 1.3, 1.5, 2.0]))
        nfeat = int(rng.integers(100, 2500)); ini = int(rng.integers(8, 40)); mn = int(rng.integers(3, ini + 1))
        kind = rng.integers(0, 4)
        if kind == 0: img = synth_frame(int(rng.integers(0, 10000)), w, h)
        elif kind == 1: img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        elif kind == 2: img = (synth_frame(int(rng.integers(0, 10000)), w, h) // 4 + 100).astype(np.uint8)
        else:
            img = np.full((h, w), 90, np.uint8)
            for _ in range(int(rng.integers(1, 60))):
                x, y = rng.integers(0, w - 20), rng.integers(0, h - 20)
                img[y:y + rng.integers(3, 20), x:x + rng.integers(3, 20)] = rng.integers(0, 256)
        params = (nfeat, sf, nlev, ini, mn)
        try:
            ex = ORBextractor(*params)
            kps, desc = ex(img)
        except Exception as e:                      # sizes the reference cannot run either, or the documented capacity limits
            assert "error -5" in str(e) or "error -1" in str(e), str(e)
            continue
        okps, odesc = oracle.OrbOracle(*params).extract(img)
        _kp_equal(kps, okps)
        assert np.array_equal(desc, odesc), (params, w, h, kind)
        done += 1


def test_reserve_sizes_the_workspace_up_front():
    """orbx_reserve: the workspace for (width, height, batch) is built before the first frame; bad sizes are an error and a
    later extraction at the reserved size gives the usual result."""
    from orb_slam2_e_amd._lib import lib
    ex = ORBextractor(*PARAMS)
    L = lib()
    assert L.orbx_reserve(ex._h, 640, 480, 2) == 0
    assert L.orbx_reserve(ex._h, 640, 480, 1) == 0          # a smaller batch fits the same workspace
    assert L.orbx_reserve(ex._h, 0, 480, 1) != 0 and L.orbx_reserve(ex._h, 640, 480, 0) != 0
    assert L.orbx_reserve(None, 640, 480, 1) != 0
    img = synth_frame(1)
    o = oracle.OrbOracle(*PARAMS)
    okps, odesc = o.extract(img)
    kps, desc = ex(img)
    _kp_equal(kps, okps)
    assert np.array_equal(desc, odesc)


@pytest.mark.parametrize("w,h,nfeat", [(1920, 1080, 2000), (1920, 1080, 5000), (3999, 501, 3000), (3840, 2160, 4000)])
def test_full_hd_and_4k_frames_bit_exact(w, h, nfeat):
    """Sizes beyond the bench's.  k_octree ranks nodes by a packed (size, creation seq) key while a level has fewer than 2^21
    candidate slots (a good quarter of its FAST zone); a 3840 x 2160 level 0 has 2.1 M and takes the unpacked comparison.
    4000 x 4000 exceeds the octree's LDS node pool and must be refused, not mangled."""
    import oracle
    from orb_slam2_e_amd import ORBextractor, OrbxError
    from orb_slam2_e_amd.synth import synth_frame
    prm = (nfeat, 1.2, 8, 20, 7)
    img = synth_frame(7, w, h)
    k, d = ORBextractor(*prm)(img)
    ok_, od = oracle.OrbOracle(*prm).extract(img)
    assert len(k) == len(ok_) >= nfeat - 20 and np.array_equal(d, od) and np.array_equal(k.view(np.uint8), ok_.view(np.uint8))
    if w == 1920 and nfeat == 2000:
        with pytest.raises(OrbxError) as e:
            ORBextractor(8000, 1.2, 8, 20, 7)(np.zeros((4000, 4000), np.uint8))
        assert e.value.code == -5


@pytest.mark.parametrize("nfeat", [1, 3, 5, 10])
def test_tiny_feature_counts(nfeat):
    """Per-level quotas of 0 and 1: DistributeOctTree still splits its initial node once, so a level returns up to four
    keypoints (ORBextractor.cc:575-669) and the frame more than nfeatures + 3 nlevels."""
    import oracle
    from orb_slam2_e_amd import ORBextractor
    from orb_slam2_e_amd.synth import synth_frame
    prm = (nfeat, 1.2, 8, 20, 7)
    img = synth_frame(3, 640, 480)
    k, d = ORBextractor(*prm)(img)
    ok_, od = oracle.OrbOracle(*prm).extract(img)
    assert len(k) == len(ok_) > nfeat and np.array_equal(d, od) and np.array_equal(k.view(np.uint8), ok_.view(np.uint8))


@pytest.mark.parametrize("prm,size", [((108, 1.25, 8, 38, 25), (819, 320)), ((40, 1.2, 8, 20, 7), (1242, 375)),
                                      ((12, 1.2, 4, 20, 7), (1242, 375)), ((2000, 1.2, 8, 20, 7), (1242, 375))])
def test_small_quotas_on_wide_frames(prm, size):
    """DistributeOctTree starts from round(width / height) nodes per level and splits every one of them before it first asks
    whether there are enough (ORBextractor.cc:543, :575-669): a level returns up to 4 nIni keypoints whatever its quota, 16 per
    level on a KITTI-shaped frame.  With small quotas that is more than nfeatures + 3 nlevels in total -- the capacity the
    library reports follows the frame size (found by the differential fuzzer: 137 keypoints against a capacity of 132)."""
    import oracle
    from orb_slam2_e_amd import ORBextractor
    from orb_slam2_e_amd.synth import synth_frame
    img = synth_frame(1, *size)
    ex = ORBextractor(*prm)
    k, d = ex(img)
    ok_, od = oracle.OrbOracle(*prm).extract(img)
    assert len(k) == len(ok_) and np.array_equal(d, od) and np.array_equal(k.view(np.uint8), ok_.view(np.uint8))
    assert ex.capacity >= len(k)
    if prm[0] < 200:
        assert len(k) > prm[0] + 3 * prm[2]
    # the batch entry and a second frame size on the same handle
    ex.extract_batch(np.stack([img, img[::-1].copy()]))
    kb, db, cb = ex.download_batch()
    assert cb[0] == len(k) and np.array_equal(db[0, :cb[0]], od)
    img2 = synth_frame(2, 640, 480)
    k2, d2 = ex(img2)
    ok2, od2 = oracle.OrbOracle(*prm).extract(img2)
    assert len(k2) == len(ok2) and np.array_equal(d2, od2)


@pytest.mark.parametrize("shape,params", [
    ((480, 640), (2000, 1.2, 8, 20, 7)),
    ((375, 1242), (2000, 1.2, 8, 20, 7)),     # KITTI-shaped
    ((300, 400), (500, 1.5, 4, 30, 10)),      # 4 levels, scale 1.5
    ((243, 317), (300, 1.2, 8, 20, 7)),       # odd sizes: widths that are no multiple of 4 at most levels
    ((2160, 3840), (2000, 1.2, 8, 20, 7)),    # 4K: many tiles
    ((200, 260), (200, 1.2, 5, 20, 7)),       # small: few tiles, shallow rectangles
])
def test_pyramid_in_one_launch_equals_the_per_level_launches(shape, params, monkeypatch):
    """Small batches build levels 1.. in ONE launch (k_pyr_chain: a workgroup carries a tile through all levels in LDS, neighbours
    recompute the overlap) and run k_octree with 512 threads per (level, frame); large ones keep a launch per level and 256 threads.  Same
    padded levels byte for byte, same keypoints and descriptors -- one frame and a batch of three, each under both forms (the two
    environment switches are read per call)."""
    from orb_slam2_e_amd.synth import synth_frame
    h, w = shape
    frames = [np.ascontiguousarray(np.tile(synth_frame(k), ((h + 479) // 480, (w + 639) // 640))[:h, :w]) for k in range(3)]
    rng = np.random.default_rng(5)
    frames = [np.clip(f.astype(np.int16) + rng.integers(-3, 4, f.shape), 0, 255).astype(np.uint8) for f in frames]
    out = {}
    for form, limit in (("chain", "64"), ("levels", "0")):
        monkeypatch.setenv("ORBX_PYR_CHAIN_MAX_BATCH", limit)
        monkeypatch.setenv("ORBX_OCT_WIDE_MAX_BATCH", limit)    # ... and the 512-thread k_octree of small batches against the 256-thread one
        ex = ORBextractor(*params)
        kps, desc = ex(frames[0])
        one = [ex.pyramid_level(0, l, padded=True).copy() for l in range(params[2])]
        ex.extract_batch(frames)
        res = ex.download_batch()
        three = [ex.pyramid_level(f, l, padded=True).copy() for f in range(3) for l in range(params[2])]
        out[form] = (kps, desc, one, res, three)
    a, b = out["chain"], out["levels"]
    for x, y in zip(a[2] + a[4], b[2] + b[4]):
        assert np.array_equal(x, y)
    assert a[0].tobytes() == b[0].tobytes() and np.array_equal(a[1], b[1])
    (ka, da, ca), (kb, db, cb) = a[3], b[3]       # the batch: what lies behind a frame's count is not defined
    assert np.array_equal(ca, cb)
    for f in range(3):
        assert ka[f, :ca[f]].tobytes() == kb[f, :cb[f]].tobytes() and np.array_equal(da[f, :ca[f]], db[f, :cb[f]])


@pytest.mark.parametrize("shape", [(480, 640), (375, 1242), (243, 317)])
@pytest.mark.parametrize("pad", [4, 13, 64])
def test_rows_with_a_pitch_larger_than_the_width(shape, pad):
    """ORBextractor::operator() takes any cv::Mat (ORBextractor.cc:1056: _image.getMat()): rows `stride` bytes apart with stride > width
    -- a region of interest, an aligned allocation.  Same keypoints and descriptors as the continuous copy, for pitches that keep the
    rows dword-aligned and for one that does not; the bytes between the rows are never read as pixels (they are poisoned here), and
    nothing is read behind the last row's last pixel (the buffer ends there)."""
    import ctypes as C
    from orb_slam2_e_amd._lib import check
    from orb_slam2_e_amd.extractor import KP_DTYPE
    h, w = shape
    img = np.ascontiguousarray(np.tile(synth_frame(3), ((h + 479) // 480, (w + 639) // 640))[:h, :w])
    ex = ORBextractor(*PARAMS)
    k0, d0 = ex(img)
    stride = w + pad
    buf = np.full((h - 1) * stride + w, 0xA5, np.uint8)          # ends with the last pixel of the last row
    for y in range(h):
        buf[y * stride: y * stride + w] = img[y]
    ex2 = ORBextractor(*PARAMS)
    ex2._reserve(w, h, 1)
    kps = np.zeros(ex2.capacity, KP_DTYPE); desc = np.zeros((ex2.capacity, 32), np.uint8); n = C.c_int(0)
    for rep in range(2):       # (the second call of a size takes the cached path)
        check(ex2._L.orbx_extract(ex2._h, buf.ctypes.data_as(C.c_void_p), w, h, stride, kps.ctypes.data_as(C.c_void_p),
                                  desc.ctypes.data_as(C.c_void_p), ex2.capacity, C.byref(n)))
        assert n.value == len(k0) and kps[:n.value].tobytes() == k0.tobytes() and np.array_equal(desc[:n.value], d0)


def test_one_handle_sizes_and_batches_in_turn():
    """One ORBextractor serving frames of changing size and batches of changing length (orbx_reserve re-plans; the pair path's captured
    graph, the small-batch kernel forms and the cached plan must all follow): every result equals a fresh handle's."""
    rng = np.random.default_rng(9)
    shapes = [(480, 640), (240, 320), (480, 640), (375, 1242), (480, 640)]
    ex = ORBextractor(*PARAMS)
    for step, (h, w) in enumerate(shapes):
        img = np.ascontiguousarray(np.tile(synth_frame(step), ((h + 479) // 480, (w + 639) // 640))[:h, :w])
        fresh = ORBextractor(*PARAMS)
        k0, d0 = fresh(img)
        k1, d1 = ex(img)
        assert k1.tobytes() == k0.tobytes() and np.array_equal(d1, d0), (step, h, w)
        B = int(rng.integers(2, 10))
        frames = np.stack([np.roll(img, 7 * b, axis=1) for b in range(B)])
        ex.extract_batch(frames); kb, db, cb = ex.download_batch()
        fresh.extract_batch(frames); kf, df, cf = fresh.download_batch()
        assert np.array_equal(cb, cf)
        for b in range(B):
            assert kb[b, :cb[b]].tobytes() == kf[b, :cf[b]].tobytes() and np.array_equal(db[b, :cb[b]], df[b, :cf[b]]), (step, B, b)
        k2, d2 = ex(img)            # ... and back to one frame after the batch
        assert k2.tobytes() == k0.tobytes() and np.array_equal(d2, d0)


def test_image_in_pinned_memory_is_read_in_place():
    """An image that already lies in pinned host memory (a capture buffer allocated with hipHostMalloc) is not staged: the level-0 kernel
    reads it where it is.  Same results as from pageable memory -- also as a region of interest of a larger pinned array."""
    import torch
    img = synth_frame(4)
    ex = ORBextractor(*PARAMS)
    k0, d0 = ex(img)
    pin = torch.from_numpy(img).pin_memory().numpy()
    assert pin.ctypes.data != img.ctypes.data
    k1, d1 = ex(pin)
    assert k1.tobytes() == k0.tobytes() and np.array_equal(d1, d0)
    big = torch.full((500, 700), 0x33, dtype=torch.uint8).pin_memory().numpy()
    big[7:487, 12:652] = img
    k2, d2 = ex(big[7:487, 12:652])
    assert k2.tobytes() == k0.tobytes() and np.array_equal(d2, d0)
    k3, d3 = ex(img)           # ... and pageable again
    assert k3.tobytes() == k0.tobytes()
