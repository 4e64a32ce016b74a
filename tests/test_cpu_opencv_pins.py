"""Consumes tests/golden/opencv_pins.npz -- outputs of REAL OpenCV for the primitives the oracle restates, written by
tools/pin_with_opencv.py on a machine that has cv2 -- and compares the oracle with them bit for bit.  The file cannot be produced in
this repository's containers (no OpenCV anywhere: SURVEY 8c), so until someone commits it this module is SKIPPED and parity stays
"unpinned" (DESIGN.md 5).  When it exists, it is the only reference-held evidence the oracle has."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle

PINS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_pins.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PINS), reason="tests/golden/opencv_pins.npz not committed: run tools/pin_with_opencv.py where cv2 exists")


def _z():
    return dict(np.load(PINS))


def _frames(z):
    from orb_slam2_e_amd.synth import synth_frame, synth_stereo_pair
    return [synth_frame(1000 + k) for k in range(int(z["nframes"]))] + [synth_stereo_pair(0)[0]]


@pytest.mark.parametrize("tag", ["plain", "optimized"])
def test_pyramid_border_and_blur_equal_opencv(tag):
    z = _z()
    params = tuple(z["params"])
    variants = []
    for f, img in enumerate(_frames(z)):
        o = oracle.OrbOracle(int(params[0]), float(params[1]), int(params[2]), int(params[3]), int(params[4]))
        o.extract(img)
        for l in range(int(params[2])):
            assert np.array_equal(o.level_image(l), z[f"{tag}/f{f}/l{l}/level"]), f"resize: frame {f} level {l}"
            assert np.array_equal(o.level_padded(l), z[f"{tag}/f{f}/l{l}/padded"]), f"copyMakeBorder: frame {f} level {l}"
        # GaussianBlur: the taps depend on the OpenCV version (blur_variant 0: >= 3.4.9 / 4.x, 1: 3.2 - 3.4.8); one of the two must
        # match every level of every frame, and which one is recorded
        for taps in ((18, 34, 48, 56, 48, 34, 18), (18, 34, 49, 55, 49, 34, 18)):
            o2 = oracle.OrbOracle(int(params[0]), float(params[1]), int(params[2]), int(params[3]), int(params[4]))
            o2.set_blur_taps(taps)
            o2.extract(img)
            if all(np.array_equal(o2.level_blurred(l), z[f"{tag}/f{f}/l{l}/blur"]) for l in range(int(params[2]))):
                variants.append(taps[2])
                break
        else:
            pytest.fail(f"GaussianBlur: frame {f}: neither tap set reproduces OpenCV {z['cv2_version']}")
    assert len(set(variants)) == 1
    print("OpenCV", z["cv2_version"], "GaussianBlur taps variant:", 0 if variants[0] == 48 else 1)


@pytest.mark.parametrize("tag", ["plain", "optimized"])
def test_fast_equals_opencv(tag):
    z = _z()
    L = oracle.lib()
    cand = np.dtype([("x", "<f4"), ("y", "<f4"), ("score", "<f4")])           # oracle_cand
    for key in [k for k in z if k.startswith(tag + "/") and (k.endswith("/fast20") or k.endswith("/fast7") or k.endswith("_fast20"))]:
        base = key.rsplit("/", 1)[0]
        img = z[base + "/level"]
        if "roi" in key:
            x0, y0, w, h = z[key[:-len("_fast20")]]
            img = img[y0:y0 + h, x0:x0 + w]
        th = 7 if key.endswith("fast7") else 20
        img = np.ascontiguousarray(img)
        out = np.zeros(img.size, cand)
        L.oracle_fast_detect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        n = L.oracle_fast_detect(img.ctypes.data, img.shape[1], img.shape[0], img.shape[1], th, out.ctypes.data, len(out))
        ref = z[key]
        assert n == len(ref), key
        got = np.stack([out["x"][:n], out["y"][:n], out["score"][:n]], 1)
        assert np.array_equal(got, ref), key             # positions, responses AND emission order


@pytest.mark.parametrize("tag", ["plain", "optimized"])
def test_fast_atan2_equals_opencv(tag):
    z = _z()
    L = oracle.lib()
    yx, ref = z[f"{tag}/atan2_yx"], z[f"{tag}/atan2"]
    got = np.array([L.oracle_fastAtan2(float(y), float(x)) for y, x in yx], np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
