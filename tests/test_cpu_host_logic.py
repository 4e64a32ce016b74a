"""Host-side logic of the library that needs no device: the frame-grid counting sort of the whole-loop searches
(orbm_sorted_frame) against the oracle's AssignFeaturesToGrid, argument checks of the host-array entry points, the
no-device error path of the workspace pool.  Also what tools/asan_host.sh runs against the host-only ASan / UBSan build."""
import ctypes as C

import numpy as np
import pytest

import oracle
from orb_slam2_e_amd import KP_DTYPE, OrbxError
from orb_slam2_e_amd._lib import check, lib


def _sorted_frame(kps, skip, uright, bounds):
    L = lib()
    n = len(kps)
    perm = np.zeros(max(n, 1), np.int32); off = np.zeros(64 * 48 + 1, np.int32); ns = C.c_int(0)
    L.orbm_sorted_frame.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p] * 3
    p = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
    check(L.orbm_sorted_frame(p(kps), n, p(skip), p(uright), *bounds, p(perm), p(off), C.byref(ns)))
    return perm[:ns.value].copy(), off


@pytest.mark.parametrize("seed,n,bounds", [(0, 2000, (0.0, 0.0, 640.0, 480.0)), (1, 5000, (-12.5, -7.25, 1254.0, 380.5)),
                                           (2, 1, (0.0, 0.0, 640.0, 480.0)), (3, 0, (0.0, 0.0, 640.0, 480.0)), (4, 65535, (0.0, 0.0, 1241.0, 376.0))])
def test_frame_grid_counting_sort_equals_assign_features_to_grid(seed, n, bounds):
    rng = np.random.default_rng(seed)
    kps = np.zeros(n, KP_DTYPE)
    w, h = bounds[2] - bounds[0], bounds[3] - bounds[1]
    kps["x"] = rng.uniform(bounds[0] - 0.05 * w, bounds[2] + 0.05 * w, n)          # some outside the grid
    kps["y"] = rng.uniform(bounds[1] - 0.05 * h, bounds[3] + 0.05 * h, n)
    if n > 10:                                                                      # exact cell borders and the far corner
        kps["x"][:5] = [bounds[0], bounds[2], bounds[0] + w / 64 * 7.5, bounds[2] - 1e-3, bounds[0] + w / 128]
        kps["y"][:5] = [bounds[1], bounds[3], bounds[1] + h / 48 * 3.5, bounds[3] - 1e-3, bounds[1] + h / 96]
    kps["octave"] = rng.integers(0, 8, n)
    perm, off = _sorted_frame(kps, None, None, bounds)
    g = oracle.Grid(np.stack([kps["x"], kps["y"]], 1).astype(np.float32) if n else np.zeros((0, 2), np.float32), kps["octave"], *bounds)
    ooff, oitems = g.tables()
    assert np.array_equal(off, ooff) and np.array_equal(perm, oitems)
    if n > 100:
        assert 0 < len(perm) < n
        skip = (rng.random(n) < 0.3).astype(np.uint8)
        perm2, off2 = _sorted_frame(kps, skip, None, bounds)
        keep = skip[oitems] == 0
        assert np.array_equal(perm2, oitems[keep])                                  # same order, skipped keypoints left out
        cells = np.repeat(np.arange(64 * 48), np.diff(ooff))
        assert np.array_equal(np.diff(off2), np.bincount(cells[keep], minlength=64 * 48))


def test_sorted_frame_rejects_bad_arguments():
    kps = np.zeros(4, KP_DTYPE)
    with pytest.raises(OrbxError):
        _sorted_frame(kps, None, None, (0.0, 0.0, 0.0, 480.0))                      # empty grid
    with pytest.raises(OrbxError):
        _sorted_frame(np.zeros(65536, KP_DTYPE), None, None, (0.0, 0.0, 640.0, 480.0))   # index does not fit the 16-bit sort key


def test_host_array_calls_fail_loudly_without_a_device():
    """Every host-array entry point stages through the workspace pool; without a device the call must come back with
    ORBX_ERR_NO_DEVICE (no fallback, no crash) -- and keep doing so on repeated calls (pool state stays consistent)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from orb_slam2_e_amd import ORBmatcher
    from orb_slam2_e_amd.fem import FEA2, FEM_TET4
    A = np.zeros((10, 32), np.uint8)
    for _ in range(3):
        with pytest.raises(OrbxError) as e:
            ORBmatcher().match_bruteforce(A, A)
        assert e.value.code == -2 and "no usable HIP device" in str(e.value)
        with pytest.raises(OrbxError):
            FEA2(np.zeros((4, 3), np.float32), np.array([[0, 1, 2, 3]], np.int32), FEM_TET4)
