"""Host-side mirror of the ORBmatcher data plane over the C-ABI (ctypes).

Constants and semantics follow include/ORBmatcher.h:37-111 and
src/ORBmatcher.cc:37-40 (TH_HIGH=95, TH_LOW=45, TH_RELOC=60, HISTO_LENGTH=30).
"""
import ctypes as C

import numpy as np

from ._lib import bind, check, lib, ptr as _p




class ORBmatcher:
    # src/ORBmatcher.cc:37-40
    TH_HIGH = 95
    TH_LOW = 45
    TH_RELOC = 60
    HISTO_LENGTH = 30

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = np.float32(nnratio)
        self.mbCheckOrientation = checkOri
        self._L = lib()

    @staticmethod
    def DescriptorDistance(a, b):
        """ORBmatcher::DescriptorDistance for one pair (runs the device kernel)."""
        return int(ORBmatcher.hamming_matrix(np.reshape(a, (1, 32)), np.reshape(b, (1, 32)))[0, 0])

    @staticmethod
    def hamming_matrix(A, B):
        L = lib()
        A = np.ascontiguousarray(A, np.uint8); B = np.ascontiguousarray(B, np.uint8)
        out = np.zeros((len(A), len(B)), np.uint16)
        check(L.orbm_hamming_matrix(_p(A), len(A), _p(B), len(B), _p(out)))
        return out

    ALLPAIRS_AUTO, ALLPAIRS_POPCOUNT, ALLPAIRS_MFMA = 0, 1, 2

    @staticmethod
    def set_allpairs_kernel(kind):
        """orbm_set_allpairs_kernel: which kernel the all-pairs matchers launch (process-wide); returns the previous one."""
        prev = lib().orbm_set_allpairs_kernel(int(kind))
        if prev < 0:
            raise ValueError("unknown all-pairs kernel kind")
        return prev

    def match_bruteforce(self, A, B):
        A = np.ascontiguousarray(A, np.uint8); B = np.ascontiguousarray(B, np.uint8)
        nA = len(A)
        best = np.zeros(nA, np.int32); second = np.zeros(nA, np.int32); idx = np.zeros(nA, np.int32)
        check(self._L.orbm_match_bruteforce(_p(A), nA, _p(B), len(B), _p(best), _p(second), _p(idx)))
        return best, second, idx

    def match_candidates(self, A, B, cand_off, cand_idx):
        A = np.ascontiguousarray(A, np.uint8); B = np.ascontiguousarray(B, np.uint8)
        off = np.ascontiguousarray(cand_off, np.int32); ci = np.ascontiguousarray(cand_idx, np.int32)
        nA = len(A)
        best = np.zeros(nA, np.int32); second = np.zeros(nA, np.int32); idx = np.zeros(nA, np.int32)
        check(self._L.orbm_match_candidates(_p(A), nA, _p(B), len(B), _p(off), _p(ci), _p(best), _p(second), _p(idx)))
        return best, second, idx

    def filter(self, best, second, idx, th=None):
        th = self.TH_LOW if th is None else th
        m = np.zeros(len(best), np.int32)
        n = C.c_int(0)
        check(self._L.orbm_match_filter(len(best), _p(np.ascontiguousarray(best, np.int32)),
                                        _p(np.ascontiguousarray(second, np.int32)),
                                        _p(np.ascontiguousarray(idx, np.int32)), th, C.c_float(self.mfNNratio),
                                        _p(m), C.byref(n)))
        return m, n.value

    def match_batch_device(self, desc_dev, counts_dev, cap, pair_a_dev, pair_b_dev, npairs,
                           best_dev, second_dev, idx_dev, match12_dev, nmatch_dev, th=None, stream=None):
        th = self.TH_LOW if th is None else th
        vp = lambda v: C.c_void_p(v) if v else None
        check(self._L.orbm_match_batch_dev(vp(desc_dev), vp(counts_dev), cap, vp(pair_a_dev), vp(pair_b_dev), npairs,
                                           th, C.c_float(self.mfNNratio), vp(best_dev), vp(second_dev), vp(idx_dev),
                                           vp(match12_dev), vp(nmatch_dev), vp(stream)))

    def match_triangulation(self, kps1, desc1, kps2, desc2, cand_off, cand_idx, has_mp1, has_mp2, stereo1, stereo2,
                            F12, ex, ey, scale_factors2, level_sigma2, bOnlyStereo=False):
        """Inner loop of ORBmatcher::SearchForTriangulation (ORBmatcher.cc:892-990).
        Returns (vMatches12, bestDist); rotation histogram / pair list stay on the host."""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        off = np.ascontiguousarray(cand_off, np.int32); ci = np.ascontiguousarray(cand_idx, np.int32)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        m1, m2, s1, s2 = u8(has_mp1), u8(has_mp2), u8(stereo1), u8(stereo2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sc = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2, np.float32)
        n1 = len(kps1)
        m12 = np.zeros(n1, np.int32); bd = np.zeros(n1, np.int32)
        bind(self._L.orbm_match_triangulation, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p,
                                                     C.c_int, C.c_void_p, C.c_void_p])
        check(self._L.orbm_match_triangulation(_p(kps1), _p(d1), n1, _p(kps2), _p(d2), len(kps2), _p(off), _p(ci), _p(m1),
                                               _p(m2), _p(s1), _p(s2), 1 if bOnlyStereo else 0, _p(F), ex, ey, _p(sc),
                                               _p(sg), len(sc), _p(m12), _p(bd)))
        return m12, bd

    def SearchForTriangulation(self, kps1, desc1, fv1, has_mp1, stereo1, kps2, desc2, fv2, has_mp2, stereo2, F12, ex, ey,
                               scale_factors2, level_sigma2, bOnlyStereo=False):
        """ORBmatcher::SearchForTriangulation (ORBmatcher.cc:858-1024): FeatureVector co-iteration into per-keypoint
        candidate lists (host), the gated loop on the device, rotation histogram + ComputeThreeMaxima + pair list (host).
        fv = (nodes ascending, off, items).  Returns (vMatchedPairs as an (m, 2) array, nmatches, vMatches12)."""
        n1, n2 = len(kps1), len(kps2)
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        u8 = lambda a: np.ascontiguousarray(a, np.uint8)
        k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        (nd1, of1, it1), (nd2, of2, it2) = (tuple(i32(a) for a in fv) for fv in (fv1, fv2))
        h1, h2, s1, s2 = u8(has_mp1), u8(has_mp2), u8(stereo1), u8(stereo2)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2, np.float32)
        m12 = np.full(n1, -1, np.int32); nm = C.c_int(0)
        bind(self._L.orbm_search_for_triangulation, ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] * 2 +
                                                          [C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]))
        check(self._L.orbm_search_for_triangulation(_p(k1), _p(d1), n1, _p(nd1), _p(of1), _p(it1), len(nd1), _p(h1), _p(s1),
                                                    _p(k2), _p(d2), n2, _p(nd2), _p(of2), _p(it2), len(nd2), _p(h2), _p(s2),
                                                    int(bool(bOnlyStereo)), _p(F), float(ex), float(ey), _p(sf), _p(sg), len(sf),
                                                    int(self.mbCheckOrientation), _p(m12), C.byref(nm)))
        i1 = np.nonzero(m12 >= 0)[0]
        return np.stack([i1, m12[i1]], 1), nm.value, m12

    @staticmethod
    def feature_vector_candidates(n1, fv1, fv2):
        """The co-iteration of two FeatureVectors (ORBmatcher.cc:881-891, 1004-1012) as per-keypoint candidate lists: a
        keypoint of frame 1 whose node also exists in frame 2 gets that node's members of frame 2, in their order.
        fv = (nodes ascending, off, items).  Returns (cand_off[n1 + 1], cand_idx)."""
        nodes1, off1, items1 = (np.asarray(a, np.int64) for a in fv1); nodes2, off2, items2 = (np.asarray(a, np.int64) for a in fv2)
        _, ia, ib = np.intersect1d(nodes1, nodes2, assume_unique=True, return_indices=True)
        len1 = off1[ia + 1] - off1[ia]; len2 = off2[ib + 1] - off2[ib]
        tot1 = int(len1.sum())
        seg = np.repeat(off1[ia], len1) + (np.arange(tot1) - np.repeat(np.cumsum(len1) - len1, len1))
        members1 = items1[seg]                                      # keypoints of frame 1 that have candidates
        cnt = np.zeros(n1, np.int64); start2 = np.zeros(n1, np.int64)
        cnt[members1] = np.repeat(len2, len1); start2[members1] = np.repeat(off2[ib], len1)
        cand_off = np.zeros(n1 + 1, np.int32); cand_off[1:] = np.cumsum(cnt)
        tot = int(cand_off[-1])
        pos = np.repeat(start2, cnt) + (np.arange(tot) - np.repeat(cand_off[:-1].astype(np.int64), cnt))
        return cand_off, items2[pos].astype(np.int32)

    @staticmethod
    def ComputeThreeMaxima(hist):
        """ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1802-1843) on the bin sizes -> the bins that stay."""
        max1 = max2 = max3 = 0; ind1 = ind2 = ind3 = -1
        for i, s in enumerate(int(x) for x in hist):
            if s > max1: max3, max2, max1, ind3, ind2, ind1 = max2, max1, s, ind2, ind1, i
            elif s > max2: max3, max2, ind3, ind2 = max2, s, ind2, i
            elif s > max3: max3, ind3 = s, i
        if max2 < np.float32(0.1) * np.float32(max1): ind2 = ind3 = -1
        elif max3 < np.float32(0.1) * np.float32(max1): ind3 = -1
        return [i for i in (ind1, ind2, ind3) if i >= 0]

    WQ_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("r", "<f4"), ("xr", "<f4"), ("min_level", "<i4"), ("max_level", "<i4")])

    def search_window(self, queries, qdesc, kps, desc, bounds, skip=None, uright=None, init_dist=256):
        """GetFeaturesInArea + best/second-with-levels (ORBmatcher.cc:69-118).
        queries: array of WQ_DTYPE; bounds = (mnMinX, mnMinY, mnMaxX, mnMaxY).
        Returns best, best_level, second, second_level, idx."""
        q = np.ascontiguousarray(queries, self.WQ_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
        kps = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
        nq = len(q)
        outs = [np.zeros(nq, np.int32) for _ in range(5)]
        sk = np.ascontiguousarray(skip, np.uint8) if skip is not None else None
        ur = np.ascontiguousarray(uright, np.float32) if uright is not None else None
        bind(self._L.orbm_search_window, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 5)
        check(self._L.orbm_search_window(_p(q), _p(qd), nq, _p(kps), _p(d), len(kps), _p(sk) if sk is not None else None,
                                         _p(ur) if ur is not None else None, *[float(b) for b in bounds], init_dist,
                                         *[_p(o) for o in outs]))
        return tuple(outs)

    def search_projection(self, queries, qdesc, qangle, qtakes, kps, desc, bounds, occupied=None, uright=None, th_accept=None,
                          ratio_same_level=False):
        """The SearchByProjection family as a whole loop (ORBmatcher.cc:46-132, :491-604, :1529-1800): in-loop
        assignment, acceptance, rotation check (self.mbCheckOrientation).  Returns (match_kp, match_q, nmatches):
        match_kp[j] = query assigned to keypoint j (-1 untouched, -2 cleared by the rotation check)."""
        q = np.ascontiguousarray(queries, self.WQ_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
        qa = np.ascontiguousarray(qangle, np.float32) if qangle is not None else None
        qt = np.ascontiguousarray(qtakes, np.uint8) if qtakes is not None else None
        kps = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
        oc = np.ascontiguousarray(occupied, np.uint8) if occupied is not None else None
        ur = np.ascontiguousarray(uright, np.float32) if uright is not None else None
        nq, n = len(q), len(kps)
        mk = np.zeros(n, np.int32); mq = np.zeros(nq, np.int32); nm = C.c_int(0)
        opt = lambda a: _p(a) if a is not None else None
        bind(self._L.orbm_search_projection, [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int, C.c_int,
                                                   C.c_void_p, C.c_void_p, C.c_void_p])
        check(self._L.orbm_search_projection(_p(q), _p(qd), opt(qa), opt(qt), nq, _p(kps), _p(d), n, opt(oc), opt(ur),
                                             *[float(b) for b in bounds], self.TH_HIGH if th_accept is None else int(th_accept),
                                             C.c_float(self.mfNNratio), int(ratio_same_level), int(self.mbCheckOrientation),
                                             _p(mk), _p(mq), C.byref(nm)))
        return mk, mq, nm.value

    def search_fuse(self, queries, qdesc, kps, desc, bounds, uright=None, inv_level_sigma2=None):
        """Candidate loop of ORBmatcher::Fuse (ORBmatcher.cc:1092-1146) -> (bestDist, bestIdx) per projected point."""
        q = np.ascontiguousarray(queries, self.WQ_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
        kps = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
        ur = np.ascontiguousarray(uright, np.float32) if uright is not None else None
        sg = np.ascontiguousarray(inv_level_sigma2, np.float32) if inv_level_sigma2 is not None else None
        best = np.zeros(len(q), np.int32); idx = np.zeros(len(q), np.int32)
        bind(self._L.orbm_search_fuse, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p])
        check(self._L.orbm_search_fuse(_p(q), _p(qd), len(q), _p(kps), _p(d), len(kps), _p(ur) if ur is not None else None,
                                       _p(sg) if sg is not None else None, len(sg) if sg is not None else 0,
                                       *[float(b) for b in bounds], _p(best), _p(idx)))
        return best, idx

    def SearchBySim3(self, q12, qdesc1, kps2, desc2, q21, qdesc2, kps1, desc1, bounds):
        """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1303-1527) over projected points (r < 0 = skipped point): two
        independent window searches (levels [l-1, l], <= TH_HIGH) and the agreement check (:1509-1524).
        Returns (match12, nFound)."""
        b1, _, _, _, i1 = self.search_window(q12, qdesc1, kps2, desc2, bounds, None, None, 2**31 - 1)
        b2, _, _, _, i2 = self.search_window(q21, qdesc2, kps1, desc1, bounds, None, None, 2**31 - 1)
        m1 = np.where((i1 >= 0) & (b1 <= self.TH_HIGH), i1, -1); m2 = np.where((i2 >= 0) & (b2 <= self.TH_HIGH), i2, -1)
        ok = m1 >= 0
        agree = np.zeros(len(m1), bool)
        agree[ok] = m2[m1[ok]] == np.nonzero(ok)[0]
        out = np.where(agree, m1, -1).astype(np.int32)
        return out, int(agree.sum())

    def SearchForInitialization(self, kps1, desc1, kps2, desc2, vbPrevMatched, bounds, windowSize=10):
        """ORBmatcher::SearchForInitialization (ORBmatcher.cc:606-721).  kps = mvKeysUn of F1 / F2, bounds = F2's
        grid bounds.  Returns (vnMatches12, updated vbPrevMatched, nmatches)."""
        k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        pv = np.array(vbPrevMatched, np.float32, copy=True).reshape(-1, 2)
        m12 = np.zeros(len(k1), np.int32); nm = C.c_int(0)
        bind(self._L.orbm_search_for_initialization, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                                           C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float, C.c_int,
                                                           C.c_void_p, C.c_void_p])
        check(self._L.orbm_search_for_initialization(_p(k1), _p(d1), len(k1), _p(k2), _p(d2), len(k2), _p(pv),
                                                     *[float(b) for b in bounds], int(windowSize), C.c_float(self.mfNNratio),
                                                     int(self.mbCheckOrientation), _p(m12), C.byref(nm)))
        return m12, pv, nm.value

    def SearchByBoW(self, fv1, valid1, desc1, angle1, fv2, valid2, desc2, angle2, kf_kf=False):
        """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (ORBmatcher.cc:360-489; kf_kf=False) or
        SearchByBoW(KeyFrame*, KeyFrame*, ...) (:723-856; kf_kf=True).  fv = feature_vector_arrays(...) of each
        side, valid = "owns a good MapPoint" masks (valid2 only for the KeyFrame form), angle = mvKeysUn angles.
        Returns (match12, match21, nmatches)."""
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        d1 = np.ascontiguousarray(desc1, np.uint8); d2 = np.ascontiguousarray(desc2, np.uint8)
        a1 = np.ascontiguousarray(angle1, np.float32); a2 = np.ascontiguousarray(angle2, np.float32)
        nodes1, off1, it1 = [i32(a) for a in fv1]; nodes2, off2, it2 = [i32(a) for a in fv2]
        v1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = np.ascontiguousarray(valid2, np.uint8) if kf_kf else None
        n1, n2 = len(d1), len(d2)
        m12 = np.zeros(n1, np.int32); m21 = np.zeros(n2, np.int32); nm = C.c_int(0)
        bind(self._L.orbm_search_by_bow, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
        check(self._L.orbm_search_by_bow(_p(nodes1), _p(off1), _p(it1), len(nodes1), _p(v1), _p(d1), _p(a1), n1,
                                         _p(nodes2), _p(off2), _p(it2), len(nodes2), _p(v2) if v2 is not None else None, _p(d2),
                                         _p(a2), n2, self.TH_LOW, int(kf_kf), C.c_float(self.mfNNratio),
                                         int(self.mbCheckOrientation), _p(m12), _p(m21), C.byref(nm)))
        return m12, m21, nm.value

    CAM_DTYPE = np.dtype([("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("min_x", "<i4"), ("max_x", "<i4"),
                          ("min_y", "<i4"), ("max_y", "<i4"), ("gminx", "<f4"), ("gminy", "<f4"), ("gmaxx", "<f4"), ("gmaxy", "<f4")])

    def SearchByProjectionMap(self, kps, desc, has_mp, mp_pos, mp_normal, mp_min_dist, mp_max_dist, mp_desc, Rcw, tcw, cam,
                              scale_factors, th=1.0):
        """The fork's SearchByProjection(Frame&, Map*, Rcw, tcw, ...) (ORBmatcher.cc:134-222).
        Returns (vMatchedMPs as indices or -1, nmatches, proj[m,4])."""
        kps = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
        hm = np.ascontiguousarray(has_mp, np.uint8)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, nrm, mn, mx = f32(mp_pos), f32(mp_normal), f32(mp_min_dist), f32(mp_max_dist)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        R = np.ascontiguousarray(Rcw, np.float64).reshape(9); t = np.ascontiguousarray(tcw, np.float64).reshape(3)
        cam = np.ascontiguousarray(cam, self.CAM_DTYPE).reshape(1)
        sc = f32(scale_factors)
        n, m = len(kps), len(pos)
        matched = np.zeros(n, np.int32); proj = np.zeros((m, 4), np.float32); nm = C.c_int(0)
        bind(self._L.orbm_search_by_projection_map, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int,
                                                          C.c_void_p, C.c_void_p, C.c_void_p])
        check(self._L.orbm_search_by_projection_map(_p(kps), _p(d), n, _p(hm), _p(pos), _p(nrm), _p(mn), _p(mx), _p(md), m,
                                                    _p(R), _p(t), _p(cam), _p(sc), len(sc), th, C.c_float(self.mfNNratio),
                                                    self.TH_RELOC, _p(matched), C.byref(nm), _p(proj)))
        return matched, nm.value, proj

    PROJ_DTYPE = np.dtype([("u", "<f4"), ("v", "<f4"), ("ur", "<f4"), ("view_cos", "<f4"), ("dist", "<f4"), ("level", "<i4"),
                           ("visible", "<i4")])
    PROJECT_FRUSTUM, PROJECT_FUSE, PROJECT_FUSE_SIM3 = 0, 1, 2

    def project_points(self, mode, mp_pos, mp_normal, mp_min_distance, mp_max_distance, Rcw, tcw, Ow, cam, mbf, log_scale_factor,
                       scale_factors, th=1.0, viewing_cos_limit=0.5):
        """orbm_project_points: Frame::isInFrustum + MapPoint::PredictScale (Frame.cc:284-340, MapPoint.cc:464-480;
        mode PROJECT_FRUSTUM) or the projection block of ORBmatcher::Fuse (ORBmatcher.cc:1053-1094 / :1212-1250) for all
        map points at once.  Returns (projected[PROJ_DTYPE], window queries[WQ_DTYPE])."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, nrm, mn, mx = f32(mp_pos), f32(mp_normal), f32(mp_min_distance), f32(mp_max_distance)
        R, t, O = f32(Rcw).reshape(9), f32(tcw).reshape(3), f32(Ow).reshape(3)
        cam = np.ascontiguousarray(cam, self.CAM_DTYPE).reshape(1)
        sc = f32(scale_factors)
        m = len(pos)
        out = np.zeros(m, self.PROJ_DTYPE); q = np.zeros(m, self.WQ_DTYPE)
        bind(self._L.orbm_project_points, [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4 + [C.c_float] * 3 + \
                                               [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p])
        check(self._L.orbm_project_points(int(mode), _p(pos), _p(nrm), _p(mn), _p(mx), m, _p(R), _p(t), _p(O), _p(cam), float(mbf),
                                          float(viewing_cos_limit), float(log_scale_factor), _p(sc), len(sc), float(th), _p(out), _p(q)))
        return out, q

    def Fuse(self, kps, desc, uright, bounds, inv_level_sigma2, vpMapPoints, mp_pos, mp_normal, mp_min_distance, mp_max_distance,
             mp_desc, Rcw, tcw, Ow, cam, mbf, log_scale_factor, scale_factors, th, map_ops, sim3=False):
        """ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) as a whole (ORBmatcher.cc:1026-1176; sim3=True: the
        projection and candidate loop of the Sim3 form, :1178-1301, whose map update differs -- see fuse_replay):
        projection of every listed map point and the gated candidate loop on the device, then the reference's loop tail
        on the host in list order.  vpMapPoints[i] = map point handle or -1 (NULL entry); mp_* arrays are indexed by i.
        map_ops: is_bad(h), is_in_keyframe(h), slot_owner(idx) -> handle or -1, observations(h), replace(dead, heir),
        add(h, idx); the Sim3 form needs already_found(h) (spAlreadyFound, a snapshot) and record_replace(i, h) instead of
        is_in_keyframe / observations / replace.  Returns nFused."""
        mode = self.PROJECT_FUSE_SIM3 if sim3 else self.PROJECT_FUSE
        proj, q = self.project_points(mode, mp_pos, mp_normal, mp_min_distance, mp_max_distance, Rcw, tcw, Ow, cam, mbf,
                                      log_scale_factor, scale_factors, th)
        best, idx = self.search_fuse(q, mp_desc, kps, desc, bounds, uright, None if sim3 else inv_level_sigma2)
        if sim3:
            return self.fuse_replay_sim3(vpMapPoints, proj["visible"], best, idx, map_ops)
        return self.fuse_replay(vpMapPoints, proj["visible"], best, idx, map_ops)

    def fuse_replay_sim3(self, vpPoints, visible, best, idx, map_ops):
        """The tail of the Sim3 form (ORBmatcher.cc:1194-1205 skips, :1279-1296 update): the "already found" test reads the
        snapshot of the key frame's map points taken before the loop -- map_ops.already_found(h) must answer from that
        snapshot (KeyFrame::GetMapPoints at call time), not from the state the loop is changing; a taken slot is only
        recorded, map_ops.record_replace(i, h_in_kf) (vpReplacePoint[i] = pMPinKF); a free slot gets the point."""
        nFused = 0
        for i, h in enumerate(vpPoints):
            if h < 0 or map_ops.is_bad(h) or map_ops.already_found(h):
                continue
            if not visible[i] or idx[i] < 0:
                continue
            if best[i] <= self.TH_LOW:
                other = map_ops.slot_owner(int(idx[i]))
                if other >= 0:
                    if not map_ops.is_bad(other):
                        map_ops.record_replace(i, other)
                else:
                    map_ops.add(h, int(idx[i]))
                nFused += 1
        return nFused

    def fuse_replay(self, vpMapPoints, visible, best, idx, map_ops):
        """The tail of the Fuse loop (ORBmatcher.cc:1046-1053 skips, :1149-1170 update) in the list's order: the candidate
        search of a point does not depend on the map state, the skips and the update do."""
        nFused = 0
        for i, h in enumerate(vpMapPoints):
            if h < 0 or map_ops.is_bad(h) or map_ops.is_in_keyframe(h):
                continue
            if not visible[i] or idx[i] < 0:
                continue
            if best[i] <= self.TH_LOW:
                other = map_ops.slot_owner(int(idx[i]))
                if other >= 0:
                    if not map_ops.is_bad(other):
                        if map_ops.observations(other) > map_ops.observations(h):
                            map_ops.replace(h, other)       # pMP->Replace(pMPinKF)
                        else:
                            map_ops.replace(other, h)       # pMPinKF->Replace(pMP)
                else:
                    map_ops.add(h, int(idx[i]))             # AddObservation + AddMapPoint
                nFused += 1
        return nFused

    @staticmethod
    def distinctive_descriptors(desc, off):
        """MapPoint::ComputeDistinctiveDescriptors for a batch (MapPoint.cc:305-370)."""
        L = lib()
        d = np.ascontiguousarray(desc, np.uint8); o = np.ascontiguousarray(off, np.int32)
        best = np.zeros(len(o) - 1, np.int32)
        check(L.orbm_distinctive_descriptors(_p(d), _p(o), len(o) - 1, _p(best)))
        return best

    # ---------------------------------------------------------------- searches on a resident frame (orbm_frame)
    def frame_search_window(self, frame, queries, qdesc, skip=None, init_dist=256):
        q = np.ascontiguousarray(queries, self.WQ_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
        nq = len(q)
        outs = [np.zeros(nq, np.int32) for _ in range(5)]
        sk = np.ascontiguousarray(skip, np.uint8) if skip is not None else None
        bind(self._L.orbm_frame_search_window, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5)
        check(self._L.orbm_frame_search_window(frame._h, _p(q), _p(qd), nq, _p(sk) if sk is not None else None, int(init_dist),
                                               *[_p(o) for o in outs]))
        return tuple(outs)

    def frame_search_fuse(self, frame, queries, qdesc, inv_level_sigma2=None):
        q = np.ascontiguousarray(queries, self.WQ_DTYPE); qd = np.ascontiguousarray(qdesc, np.uint8)
        sg = np.ascontiguousarray(inv_level_sigma2, np.float32) if inv_level_sigma2 is not None else None
        best = np.zeros(len(q), np.int32); idx = np.zeros(len(q), np.int32)
        bind(self._L.orbm_frame_search_fuse, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
        check(self._L.orbm_frame_search_fuse(frame._h, _p(q), _p(qd), len(q), _p(sg) if sg is not None else None,
                                             len(sg) if sg is not None else 0, _p(best), _p(idx)))
        return best, idx

    def frame_search_projection(self, frame, queries, qdesc, qangle, qtakes, occupied=None, th_accept=None, ratio_same_level=False):
        """orbm_search_projection on a resident frame -> (match_kp, match_q, nmatches)."""
        asc = np.ascontiguousarray
        q = asc(queries, self.WQ_DTYPE); qd = asc(qdesc, np.uint8)
        qa = asc(qangle, np.float32) if qangle is not None else None     # (locals: a converted copy must outlive the call)
        qt = asc(qtakes, np.uint8) if qtakes is not None else None
        oc = asc(occupied, np.uint8) if occupied is not None else None
        nq = len(q)
        mk = np.zeros(max(frame.n, 1), np.int32); mq = np.zeros(nq, np.int32); nm = C.c_int(0)
        fn = bind(self._L.orbm_frame_search_projection, [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int] +
                  [C.c_void_p] * 3)
        check(fn(frame._h, _p(q), _p(qd), _p(qa) if qa is not None else None, _p(qt) if qt is not None else None, nq,
                 _p(oc) if oc is not None else None,
                 self.TH_HIGH if th_accept is None else int(th_accept), self.mfNNratio, int(ratio_same_level),
                 int(self.mbCheckOrientation), _p(mk), _p(mq), C.byref(nm)))
        return mk[:frame.n], mq, nm.value

    def frame_search_for_initialization(self, frame2, kps2, kps1, desc1, vbPrevMatched, windowSize=10):
        k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2); d1 = np.ascontiguousarray(desc1, np.uint8)
        pv = np.array(vbPrevMatched, np.float32, copy=True).reshape(-1, 2)
        m12 = np.zeros(len(k1), np.int32); nm = C.c_int(0)
        bind(self._L.orbm_frame_search_for_initialization, [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                                                                    C.c_void_p, C.c_void_p])
        check(self._L.orbm_frame_search_for_initialization(frame2._h, _p(k2), _p(k1), _p(d1), len(k1), _p(pv), int(windowSize),
                                                           C.c_float(self.mfNNratio), int(self.mbCheckOrientation), _p(m12), C.byref(nm)))
        return m12, pv, nm.value

    def frame_search_by_projection_map(self, frame, has_mp, mp_pos, mp_normal, mp_min_dist, mp_max_dist, mp_desc, Rcw, tcw, cam,
                                       scale_factors, th=1.0):
        hm = np.ascontiguousarray(has_mp, np.uint8)
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        pos, nrm, mn, mx = f32(mp_pos), f32(mp_normal), f32(mp_min_dist), f32(mp_max_dist)
        md = np.ascontiguousarray(mp_desc, np.uint8)
        R = np.ascontiguousarray(Rcw, np.float64).reshape(9); t = np.ascontiguousarray(tcw, np.float64).reshape(3)
        cam = np.ascontiguousarray(cam, self.CAM_DTYPE).reshape(1)
        sc = f32(scale_factors)
        m = len(pos)
        matched = np.zeros(max(frame.n, 1), np.int32); proj = np.zeros((m, 4), np.float32); nm = C.c_int(0)
        bind(self._L.orbm_frame_search_by_projection_map, [C.c_void_p] * 7 + [C.c_int] + [C.c_void_p] * 4 + \
                                                               [C.c_int, C.c_float, C.c_float, C.c_int] + [C.c_void_p] * 3)
        check(self._L.orbm_frame_search_by_projection_map(frame._h, _p(hm), _p(pos), _p(nrm), _p(mn), _p(mx), _p(md), m, _p(R), _p(t),
                                                          _p(cam), _p(sc), len(sc), th, C.c_float(self.mfNNratio), self.TH_RELOC,
                                                          _p(matched), C.byref(nm), _p(proj)))
        return matched[:frame.n], nm.value, proj

    def frame_search_by_bow(self, frame1, fv1, valid1, frame2, fv2, valid2=None, kf_kf=False):
        """SearchByBoW on two resident frames (orbm_frame_search_by_bow): fv = feature_vector_arrays(...) of each side.
        Returns (match12, match21, nmatches)."""
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        nodes1, off1, it1 = [i32(a) for a in fv1]; nodes2, off2, it2 = [i32(a) for a in fv2]
        v1 = np.ascontiguousarray(valid1, np.uint8)
        v2 = np.ascontiguousarray(valid2, np.uint8) if kf_kf else None
        m12 = np.zeros(max(frame1.n, 1), np.int32); m21 = np.zeros(max(frame2.n, 1), np.int32); nm = C.c_int(0)
        fn = bind(self._L.orbm_frame_search_by_bow, [C.c_void_p] * 4 + [C.c_int, C.c_void_p] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p] +
                  [C.c_int, C.c_int, C.c_float, C.c_int] + [C.c_void_p] * 3)
        check(fn(frame1._h, _p(nodes1), _p(off1), _p(it1), len(nodes1), _p(v1), frame2._h, _p(nodes2), _p(off2), _p(it2), len(nodes2),
                 _p(v2) if v2 is not None else None, self.TH_LOW, int(kf_kf), self.mfNNratio, int(self.mbCheckOrientation),
                 _p(m12), _p(m21), C.byref(nm)))
        return m12[:frame1.n], m21[:frame2.n], nm.value

    def frame_search_for_triangulation(self, kf1, fv1, has_mp1, kf2, fv2, has_mp2, F12, ex, ey, scale_factors2, level_sigma2,
                                       bOnlyStereo=False):
        """SearchForTriangulation on two resident keyframes (orbm_frame_search_for_triangulation).
        Returns (vMatchedPairs as an (m, 2) array, nmatches, vMatches12)."""
        i32 = lambda a: np.ascontiguousarray(a, np.int32)
        (nd1, of1, it1), (nd2, of2, it2) = (tuple(i32(a) for a in fv) for fv in (fv1, fv2))
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        F = np.ascontiguousarray(F12, np.float32).reshape(9)
        sf = np.ascontiguousarray(scale_factors2, np.float32); sg = np.ascontiguousarray(level_sigma2, np.float32)
        m12 = np.full(max(kf1.n, 1), -1, np.int32); nm = C.c_int(0)
        fn = bind(self._L.orbm_frame_search_for_triangulation, [C.c_void_p] * 4 + [C.c_int, C.c_void_p] + [C.c_void_p] * 4 +
                  [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p])
        check(fn(kf1._h, _p(nd1), _p(of1), _p(it1), len(nd1), _p(h1), kf2._h, _p(nd2), _p(of2), _p(it2), len(nd2), _p(h2),
                 int(bool(bOnlyStereo)), _p(F), float(ex), float(ey), _p(sf), _p(sg), len(sf), int(self.mbCheckOrientation), _p(m12),
                 C.byref(nm)))
        m12 = m12[:kf1.n]
        i1 = np.nonzero(m12 >= 0)[0]
        return np.stack([i1, m12[i1]], 1), nm.value, m12

    # ---------------------------------------------------------------- the projection searches as whole functions
    def SearchByProjectionLast(self, cur, view, Tcw, Tlw, last, occupied, th, bMono, want_queries=False):
        """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono) (ORBmatcher.cc:1529-1671) in one call: cur = resident
        current frame, view = View(...), last = Points(valid, pos, desc, takes, octave, angle) per keypoint of the last frame,
        occupied[j] = mvpMapPoints[j] holds an observed point.  Returns (match_kp, match_q, nmatches[, queries])."""
        return self._whole(self._L.orbm_search_by_projection_last, cur, view, [Tcw, Tlw], last, occupied,
                           [C.c_float(th), int(bool(bMono)), self.TH_HIGH, int(self.mbCheckOrientation)], want_queries)

    def SearchByProjectionPoints(self, cur, view, Tcw, points, occupied, th=1.0, viewing_cos_limit=0.5):
        """Tracking::SearchLocalPoints' data plane: Frame::isInFrustum for every listed point chained into
        SearchByProjection(Frame&, vector<MapPoint*>&, th) (ORBmatcher.cc:46-132) on a resident frame.
        Returns (match_kp, match_q, nmatches, projected[PROJ_DTYPE], queries)."""
        T = np.ascontiguousarray(Tcw, np.float32).reshape(16)
        oc = np.ascontiguousarray(occupied, np.uint8) if occupied is not None else None
        mk = np.zeros(max(cur.n, 1), np.int32); mq = np.zeros(max(points.n, 1), np.int32); nm = C.c_int(0)
        proj = np.zeros(max(points.n, 1), self.PROJ_DTYPE); q = np.zeros(max(points.n, 1), self.WQ_DTYPE)
        fn = self._L.orbm_search_by_projection_points
        fn.argtypes = None
        check(fn(cur._h, C.byref(view.c), _p(T), C.byref(points.c), _p(oc) if oc is not None else None, C.c_float(th),
                 C.c_float(viewing_cos_limit), self.TH_HIGH, C.c_float(self.mfNNratio), _p(mk), _p(mq), C.byref(nm), _p(proj), _p(q)))
        return mk[:cur.n], mq[:points.n], nm.value, proj[:points.n], q[:points.n]

    def SearchByProjectionKeyFrame(self, cur, view, Tcw, kf, occupied, th, ORBdist, want_queries=False):
        """ORBmatcher::SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (ORBmatcher.cc:1673-1800)."""
        return self._whole(self._L.orbm_search_by_projection_keyframe, cur, view, [Tcw], kf, occupied,
                           [C.c_float(th), int(ORBdist), int(self.mbCheckOrientation)], want_queries)

    def SearchByProjectionSim3(self, kf, view, Scw, points, occupied, th, want_queries=False):
        """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (ORBmatcher.cc:491-604)."""
        return self._whole(self._L.orbm_search_by_projection_sim3, kf, view, [Scw], points, occupied, [int(th), self.TH_LOW], want_queries)

    def _whole(self, fn, frame, view, poses, pts, occupied, scalars, want_queries):
        mats = [np.ascontiguousarray(T, np.float32).reshape(16) for T in poses]
        oc = np.ascontiguousarray(occupied, np.uint8) if occupied is not None else None
        mk = np.zeros(max(frame.n, 1), np.int32); mq = np.zeros(max(pts.n, 1), np.int32); nm = C.c_int(0)
        q = np.zeros(max(pts.n, 1), self.WQ_DTYPE) if want_queries else None
        fn.argtypes = None
        check(fn(frame._h, C.byref(view.c), *[_p(m) for m in mats], C.byref(pts.c), _p(oc) if oc is not None else None, *scalars,
                 _p(mk), _p(mq), C.byref(nm), _p(q) if q is not None else None))
        out = (mk[:frame.n], mq[:pts.n], nm.value)
        return out + (q[:pts.n],) if want_queries else out

    def SearchBySim3Whole(self, kf1, kf2, view, T1w, T2w, s12, R12, t12, points1, points2, th, want_queries=False):
        """ORBmatcher::SearchBySim3 (ORBmatcher.cc:1303-1527) in one call.  Returns (match12, nFound, vnMatch1, vnMatch2[, q12, q21])."""
        f32 = lambda a, k: np.ascontiguousarray(a, np.float32).reshape(k)
        T1, T2, R, t = f32(T1w, 16), f32(T2w, 16), f32(R12, 9), f32(t12, 3)
        n1, n2 = kf1.n, kf2.n
        v1 = np.zeros(max(n1, 1), np.int32); v2 = np.zeros(max(n2, 1), np.int32); m12 = np.zeros(max(n1, 1), np.int32); nf = C.c_int(0)
        q12 = np.zeros(max(n1, 1), self.WQ_DTYPE) if want_queries else None; q21 = np.zeros(max(n2, 1), self.WQ_DTYPE) if want_queries else None
        fn = self._L.orbm_search_by_sim3
        fn.argtypes = None
        check(fn(kf1._h, kf2._h, C.byref(view.c), _p(T1), _p(T2), C.c_float(s12), _p(R), _p(t), C.byref(points1.c), C.byref(points2.c),
                 C.c_float(th), self.TH_HIGH, _p(v1), _p(v2), _p(m12), C.byref(nf), _p(q12) if want_queries else None,
                 _p(q21) if want_queries else None))
        out = (m12[:n1], nf.value, v1[:n1], v2[:n2])
        return out + (q12[:n1], q21[:n2]) if want_queries else out


class _CPoints(C.Structure):
    _fields_ = [("n", C.c_int32), ("valid", C.c_void_p), ("pos", C.c_void_p), ("normal", C.c_void_p), ("min_distance", C.c_void_p),
                ("max_distance", C.c_void_p), ("desc", C.c_void_p), ("takes", C.c_void_p), ("octave", C.c_void_p), ("angle", C.c_void_p)]


class _CView(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("mb", C.c_float), ("mbf", C.c_float),
                ("log_scale_factor", C.c_float), ("nlevels", C.c_int32), ("scale_factors", C.c_void_p)]


class Points:
    """orbm_points: the flat form of a vector<MapPoint*> (see include/orbslam_hip.h)."""

    def __init__(self, valid, pos, desc, normal=None, min_distance=None, max_distance=None, takes=None, octave=None, angle=None):
        f32 = lambda a: np.ascontiguousarray(a, np.float32) if a is not None else None
        u8 = lambda a: np.ascontiguousarray(a, np.uint8) if a is not None else None
        self._keep = dict(valid=u8(valid), pos=f32(pos), normal=f32(normal), min_distance=f32(min_distance),
                          max_distance=f32(max_distance), desc=u8(desc), takes=u8(takes),
                          octave=np.ascontiguousarray(octave, np.int32) if octave is not None else None, angle=f32(angle))
        self.n = len(self._keep["valid"])
        self.c = _CPoints(self.n, *[(_p(self._keep[k]) if self._keep[k] is not None else None)
                                    for k in ("valid", "pos", "normal", "min_distance", "max_distance", "desc", "takes", "octave", "angle")])


class View:
    """orbm_view: calibration and scale pyramid of the searched frame."""

    def __init__(self, fx, fy, cx, cy, mb, mbf, log_scale_factor, scale_factors):
        self._sc = np.ascontiguousarray(scale_factors, np.float32)
        self.c = _CView(fx, fy, cx, cy, mb, mbf, log_scale_factor, len(self._sc), _p(self._sc))


class Frame:
    """orbm_frame: a Frame / KeyFrame resident in HBM (grid built once, on the device)."""

    def __init__(self, kps=None, desc=None, bounds=(0.0, 0.0, 640.0, 480.0), uright=None, _handle=None):
        self._L = lib()
        self._h = C.c_void_p()
        self.bounds = tuple(float(b) for b in bounds)
        if _handle is not None:
            self._h = _handle
        else:
            k = np.ascontiguousarray(kps); d = np.ascontiguousarray(desc, np.uint8)
            ur = np.ascontiguousarray(uright, np.float32) if uright is not None else None
            bind(self._L.orbm_frame_create, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float,
                                                  C.c_void_p])
            check(self._L.orbm_frame_create(_p(k), _p(d), len(k), _p(ur) if ur is not None else None, *self.bounds, C.byref(self._h)))
        n = C.c_int(0); ns = C.c_int(0)
        check(self._L.orbm_frame_size(self._h, C.byref(n), C.byref(ns)))
        self.n, self.ns = n.value, ns.value

    @classmethod
    def from_extractor(cls, ex, frame=0, bounds=(0.0, 0.0, 640.0, 480.0), xy_undistorted=None, uright=None, uright_from_stereo=False):
        L = lib()
        h = C.c_void_p()
        xy = np.ascontiguousarray(xy_undistorted, np.float32) if xy_undistorted is not None else None
        ur = np.ascontiguousarray(uright, np.float32) if uright is not None else None
        bind(L.orbm_frame_from_extractor, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float,
                                                C.c_float, C.c_void_p])
        check(L.orbm_frame_from_extractor(ex._h, int(frame), _p(xy) if xy is not None else None, _p(ur) if ur is not None else None,
                                          int(bool(uright_from_stereo)), *[float(b) for b in bounds], C.byref(h)))
        return cls(bounds=bounds, _handle=h)

    def alias(self, bounds):
        """orbm_frame_alias: the same device data searched with other bounds (a KeyFrame's int-valued mnMinX .. mnMaxY)."""
        h = C.c_void_p()
        bind(self._L.orbm_frame_alias, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p])
        check(self._L.orbm_frame_alias(self._h, *[float(b) for b in bounds], C.byref(h)))
        return Frame(bounds=bounds, _handle=h)

    def layout(self):
        perm = np.zeros(max(self.ns, 1), np.int32); cell_off = np.zeros(64 * 48 + 1, np.int32)
        check(self._L.orbm_frame_layout(self._h, _p(perm), _p(cell_off)))
        return perm[:self.ns], cell_off

    def close(self):
        if getattr(self, "_h", None):
            self._L.orbm_frame_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()
