"""Host-side mirror of ORB_SLAM2::ORBextractor over the C-ABI (ctypes).

Same constructor arguments, getters and call semantics as the reference class
(include/ORBextractor.h:46-112, src/ORBextractor.cc:1051-1113); images and results
are numpy arrays instead of cv::Mat / std::vector<cv::KeyPoint>.
"""
import ctypes as C

import numpy as np

from ._lib import bind, check, lib, ptr as _p

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


class _Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("blur_variant", C.c_int32), ("trig_variant", C.c_int32)]




class ORBextractor:
    def __init__(self, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, blur_variant=0, trig_variant=0):
        self._L = lib()
        self._h = C.c_void_p()
        prm = _Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, blur_variant, trig_variant)
        check(self._L.orbx_create(C.byref(prm), C.byref(self._h)))
        self.nlevels = nlevels
        self.capacity = self._L.orbx_keypoint_capacity(self._h)
        self._shape = None
        self._reserved = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbx_destroy(h)
            self._h = None

    # ---- getters (ORBextractor.h:61-84)
    def GetLevels(self):
        return self._L.orbx_get_levels(self._h)

    def _vec(self, fn, dtype):
        out = np.zeros(self.nlevels, dtype=dtype)
        check(fn(self._h, _p(out)))
        return out

    def GetScaleFactor(self):
        return float(self.GetScaleFactors()[1]) if self.nlevels > 1 else 1.0

    def GetScaleFactors(self):
        return self._vec(self._L.orbx_get_scale_factors, np.float32)

    def GetInverseScaleFactors(self):
        return self._vec(self._L.orbx_get_inv_scale_factors, np.float32)

    def GetScaleSigmaSquares(self):
        return self._vec(self._L.orbx_get_level_sigma2, np.float32)

    def GetInverseScaleSigmaSquares(self):
        return self._vec(self._L.orbx_get_inv_level_sigma2, np.float32)

    def features_per_level(self):
        return self._vec(self._L.orbx_get_features_per_level, np.int32)

    # ---- operator()
    def __call__(self, image, mask=None):
        """Returns (keypoints[KP_DTYPE], descriptors[n,32] uint8).  The mask is
        ignored, as in the reference (ORBextractor.cc:1051: _mask unused)."""
        if image is None or image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1058)"
        if not (image.strides[1] == 1 and image.strides[0] >= image.shape[1]):   # rows with a pitch (a view of a wider array: a cv::Mat region of interest) go as they are
            image = np.ascontiguousarray(image)
        h, w = image.shape
        self._reserve(w, h, 1)
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int(0)
        check(self._L.orbx_extract(self._h, _p(image), w, h, image.strides[0], _p(kps), _p(desc),
                                   self.capacity, C.byref(n)))
        self._shape = (h, w)
        return kps[:n.value].copy(), desc[:n.value].copy()

    def _reserve(self, w, h, B):
        """The workspace for this frame size; the keypoint capacity follows the size (orbx_reserve: 4 initial-node children per
        level can exceed quota + 3 on wide frames with tiny quotas)."""
        if self._reserved != (w, h, B):
            check(self._L.orbx_reserve(self._h, w, h, B))
            self._reserved = (w, h, B)
            self.capacity = self._L.orbx_keypoint_capacity(self._h)

    # ---- batch API
    def extract_batch(self, images):
        """images: [B,H,W] uint8 numpy array (host)."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        B, h, w = images.shape
        self._reserve(w, h, B)
        check(self._L.orbx_extract_batch(self._h, _p(images), 0, w, h, w, C.c_size_t(w * h), B, None))
        self._shape = (h, w)
        self._last_B = B

    def extract_batch_device(self, dev_ptr, B, h, w, stream=None):
        """dev_ptr: device address of [B,H,W] uint8 (e.g. torch tensor .data_ptr())."""
        self._reserve(w, h, B)
        check(self._L.orbx_extract_batch(self._h, C.c_void_p(dev_ptr), 1, w, h, w, C.c_size_t(w * h), B,
                                         C.c_void_p(stream) if stream else None))
        self._shape = (h, w)
        self._last_B = B

    def download(self, frame):
        kps = np.zeros(self.capacity, KP_DTYPE)
        desc = np.zeros((self.capacity, 32), np.uint8)
        n = C.c_int(0)
        check(self._L.orbx_download(self._h, frame, _p(kps), _p(desc), self.capacity, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def download_batch(self):
        """All frames of the last batch: (kps[B,cap], desc[B,cap,32], counts[B])."""
        B = self._last_B
        kps = np.zeros((B, self.capacity), KP_DTYPE); desc = np.zeros((B, self.capacity, 32), np.uint8)
        cnt = np.zeros(B, np.int32)
        check(self._L.orbx_download_batch(self._h, _p(kps), _p(desc), _p(cnt)))
        return kps, desc, cnt

    def result_dev(self):
        kps, desc, cnt, cap = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int()
        check(self._L.orbx_result_dev(self._h, C.byref(kps), C.byref(desc), C.byref(cnt), C.byref(cap)))
        return kps.value, desc.value, cnt.value, cap.value

    # ---- mvImagePyramid and staged outputs
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        check(self._L.orbx_level_size(self._h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def pyramid_level(self, frame, level, padded=False):
        w, h = self.level_size(level)
        if padded:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        fn = self._L.orbx_pyramid_level_padded if padded else self._L.orbx_pyramid_level
        check(fn(self._h, frame, level, _p(out), w))
        return out

    def blurred_level(self, frame, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        check(self._L.orbx_debug_blurred_level(self._h, frame, level, _p(out), w))
        return out

    def level_candidates(self, frame, level):
        w, h = self.level_size(level)
        cap = w * h // 4 + 16
        out = np.zeros((cap, 3), np.float32)
        n = C.c_int(0)
        check(self._L.orbx_debug_level_candidates(self._h, frame, level, _p(out), cap, C.byref(n)))
        return out[:n.value].copy()

    def level_keypoints(self, frame, level):
        out = np.zeros(self.capacity, KP_DTYPE)
        n = C.c_int(0)
        check(self._L.orbx_debug_level_keypoints(self._h, frame, level, _p(out), self.capacity, C.byref(n)))
        return out[:n.value].copy()


class _ExtractPlan(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nlevels", "keypoint_capacity", "octree_nodes", "cells_per_frame", "sel_per_frame", "fast_tile_stride",
                                         "fast_lds", "octree_lds", "octree_kshift", "reserved")] + \
               [("frame_bytes", C.c_int64), ("cands_per_frame", C.c_int64)] + \
               [(n, C.c_int32 * 16) for n in ("level_w", "level_h", "level_quota", "level_nini", "level_slots", "level_cells")]


def plan(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, width, height):
    """orbx_plan: what orbx_reserve decides about a frame size, on the host alone (no GPU needed).  Returns a dict; the per-level
    entries are lists of nlevels values."""
    L = lib()
    prm = _Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, 0)
    info = _ExtractPlan()
    check(L.orbx_plan(C.byref(prm), int(width), int(height), C.byref(info)))
    out = {}
    for n, t in _ExtractPlan._fields_:
        v = getattr(info, n)
        out[n] = list(v)[:info.nlevels] if n.startswith("level_") else v
    return out


def extract_pair(left, right, image_left, image_right):
    """The two images of a stereo frame through their two extractors in ONE call from one host thread (orbx_extract_pair: both
    kernel chains enqueued before the host waits; the reference runs them on two threads, Frame.cc:78-81).
    Returns ((kpsL, descL), (kpsR, descR)) as two ORBextractor.__call__ would."""
    il = np.ascontiguousarray(image_left); ir = np.ascontiguousarray(image_right)
    assert il.dtype == np.uint8 and il.ndim == 2 and il.shape == ir.shape and ir.dtype == np.uint8
    h, w = il.shape
    left._reserve(w, h, 1); right._reserve(w, h, 1)
    out = []
    for e in (left, right):
        out.append((np.zeros(e.capacity, KP_DTYPE), np.zeros((e.capacity, 32), np.uint8), C.c_int(0)))
    (kl, dl, nl), (kr, dr, nr) = out
    check(left._L.orbx_extract_pair(left._h, _p(il), right._h, _p(ir), w, h, il.strides[0], _p(kl), _p(dl), left.capacity, C.byref(nl),
                                    _p(kr), _p(dr), right.capacity, C.byref(nr)))
    left._shape = right._shape = (h, w)
    return (kl[:nl.value].copy(), dl[:nl.value].copy()), (kr[:nr.value].copy(), dr[:nr.value].copy())


def stereo_match_batch(left, right, mb, mbf, stream=None):
    """orbx_stereo_match on every frame of the last batch extracted on both handles (asynchronous on `stream`)."""
    L = left._L
    bind(L.orbx_stereo_match, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p])
    check(L.orbx_stereo_match(left._h, right._h, mb, mbf, C.c_void_p(stream) if stream else None))


def stereo_download_batch(left):
    """(mvuRight[B, cap], mvDepth[B, cap], counts[B]) of the last stereo_match_batch; rows past a frame's count unspecified."""
    B = left._last_B
    u = np.zeros((B, left.capacity), np.float32); d = np.zeros((B, left.capacity), np.float32); c = np.zeros(B, np.int32)
    check(left._L.orbx_stereo_download_batch(left._h, _p(u), _p(d), _p(c)))
    return u, d, c


def ComputeStereoMatches(left, right, mb, mbf, frame=0):
    """Frame::ComputeStereoMatches (src/Frame.cc:527-701) on the last results of two
    ORBextractor objects (left / right images extracted with the same parameters).
    Returns (mvuRight, mvDepth) for `frame`; -1 where unmatched."""
    L = left._L
    bind(L.orbx_stereo_match, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p])
    check(L.orbx_stereo_match(left._h, right._h, mb, mbf, None))
    u = np.zeros(left.capacity, np.float32); d = np.zeros(left.capacity, np.float32)
    n = C.c_int(0)
    check(L.orbx_stereo_download(left._h, frame, _p(u), _p(d), left.capacity, C.byref(n)))
    return u[:n.value].copy(), d[:n.value].copy()
