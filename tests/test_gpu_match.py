"""GPU parity: Hamming match kernels (through the C-ABI) vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.synth import synth_bow_case, synth_projection_case
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.synth import synth_descriptors


def test_hamming_matrix():
    A, B, _ = synth_descriptors(300, seed=3)
    assert np.array_equal(ORBmatcher.hamming_matrix(A, B[:211]), oracle.hamming_matrix(A, B[:211]))


def test_known_answers():
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, z) == 0
    assert ORBmatcher.DescriptorDistance(z, f) == 256
    one = z.copy(); one[17] = 0x10
    assert ORBmatcher.DescriptorDistance(z, one) == 1


@pytest.fixture(params=["matrix cores", "matrix cores, tiles split over waves", "popcount"])
def allpairs_kernel(request):
    """All three all-pairs kernels behind orbm_match_bruteforce / orbm_match_batch_dev must give the same integers."""
    prev = ORBmatcher.set_allpairs_kernel({"popcount": ORBmatcher.ALLPAIRS_POPCOUNT, "matrix cores": ORBmatcher.ALLPAIRS_AUTO}.get(request.param, ORBmatcher.ALLPAIRS_MFMA))
    yield request.param
    ORBmatcher.set_allpairs_kernel(prev)


@pytest.mark.parametrize("nA,nB", [(2000, 2000), (1, 1), (257, 3), (5, 1000), (2024, 2024), (100, 0)])
def test_bruteforce_config2(nA, nB, allpairs_kernel):
    A, B, _ = synth_descriptors(max(nA, nB, 1), seed=7)
    A, B = A[:nA], B[:nB]
    m = ORBmatcher(0.6)
    got = m.match_bruteforce(A, B)
    ref = oracle.match_bruteforce(A, B)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    gm, gn = m.filter(*got)
    rm, rn = oracle.match_filter(*ref, ORBmatcher.TH_LOW, 0.6)
    assert gn == rn and np.array_equal(gm, rm)


def test_ties_first_index_wins():
    A = np.zeros((4, 32), np.uint8)
    B = np.zeros((600, 32), np.uint8)
    B[:, 0] = 1          # all at distance 1
    B[300:, 1] = 1       # second half at distance 2
    got = ORBmatcher().match_bruteforce(A, B)
    assert list(got[0]) == [1] * 4 and list(got[1]) == [1] * 4 and list(got[2]) == [0] * 4


@pytest.mark.parametrize("nA,nB", [(3, 31), (33, 32), (129, 33), (64, 95), (130, 127), (7, 129), (40, 32768), (40, 32769)])
def test_bruteforce_tile_edges_and_extreme_distances(nA, nB, allpairs_kernel):
    """The matrix-core kernel's corners: ragged / exactly full 32-row train tiles, a ragged query tile, the largest
    train set whose index fits the key (32768; one more row takes the popcount kernel), distances 0 and 256
    (complemented rows), duplicated rows (first index wins) and far-apart duplicates of the best row."""
    rng = np.random.default_rng(nA * 100003 + nB)
    A = rng.integers(0, 256, (nA, 32), dtype=np.uint8)
    B = rng.integers(0, 256, (nB, 32), dtype=np.uint8)
    B[nB - 1] = A[0]                 # distance 0 in the last (ragged) row
    B[nB // 2] = A[0]                # ... and earlier: the earlier one must win, second = 0
    if nA > 1:
        B[nB // 3] = ~A[1]           # distance 256 somewhere
    if nA > 2:
        A[2] = 0
        B[:] = np.where(rng.random((nB, 1)) < 0.5, 0xFF, B)     # half the train rows all ones: 256 from the zero query
    got = ORBmatcher().match_bruteforce(A, B)
    ref = oracle.match_bruteforce(A, B)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    Z = np.zeros((5, 32), np.uint8); F = np.full((nB, 32), 0xFF, np.uint8)          # every distance 256
    got = ORBmatcher().match_bruteforce(Z, F)
    assert list(got[0]) == [256] * 5 and list(got[2]) == [0] * 5 and list(got[1]) == [256 if nB > 1 else 2**31 - 1] * 5


def test_match_batch_dev_ragged_sets_and_pairs(allpairs_kernel):
    """orbm_match_batch_dev on descriptor sets resident in HBM: ragged counts (0, 1, a partial tile, cap), arbitrary
    (query set, train set) pairs incl. a set against itself, the fused acceptance test and the per-pair match count."""
    import torch
    rng = np.random.default_rng(5)
    cap, counts = 300, [300, 0, 1, 37, 256, 129]
    nsets = len(counts)
    desc = rng.integers(0, 256, (nsets, cap, 32), dtype=np.uint8)
    desc[4, :100] = desc[0, 100:200]                       # exact matches between sets 0 and 4
    desc[5, :60] = desc[0, :60] ^ np.packbits(rng.random((60, 256)) < 0.05, axis=1, bitorder="little")   # near matches
    pa = np.array([0, 4, 5, 3, 2, 1, 0, 0, 5], np.int32)   # queries
    pb = np.array([4, 0, 0, 5, 0, 0, 1, 0, 2], np.int32)   # train
    d = torch.from_numpy(desc).cuda(); c = torch.tensor(counts, dtype=torch.int32).cuda()
    ta, tb = torch.from_numpy(pa).cuda(), torch.from_numpy(pb).cuda()
    out = [torch.full((len(pa), cap), -7, dtype=torch.int32).cuda() for _ in range(4)]
    nm = torch.full((len(pa),), -7, dtype=torch.int32).cuda()
    m = ORBmatcher(0.6)
    m.match_batch_device(d.data_ptr(), c.data_ptr(), cap, ta.data_ptr(), tb.data_ptr(), len(pa), out[0].data_ptr(), out[1].data_ptr(),
                         out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), th=50)
    torch.cuda.synchronize()
    best, second, idx, m12 = (o.cpu().numpy() for o in out)
    for p in range(len(pa)):
        nA, nB = counts[pa[p]], counts[pb[p]]
        rb, rs, ri = oracle.match_bruteforce(desc[pa[p], :nA], desc[pb[p], :nB])
        rm, rn = oracle.match_filter(rb, rs, ri, 50, 0.6)
        assert np.array_equal(best[p, :nA], rb) and np.array_equal(second[p, :nA], rs) and np.array_equal(idx[p, :nA], ri), f"pair {p}"
        assert np.array_equal(m12[p, :nA], rm) and int(nm[p]) == rn, f"pair {p}: acceptance"
        assert (best[p, nA:] == -7).all() and (m12[p, nA:] == -7).all(), f"pair {p}: rows past the query count untouched"


def test_candidate_lists():
    rng = np.random.default_rng(11)
    A, B, _ = synth_descriptors(500, seed=9)
    off = [0]; idx = []
    for i in range(len(A)):
        k = int(rng.integers(0, 40))
        idx.extend(rng.integers(0, len(B), size=k).tolist())
        off.append(len(idx))
    got = ORBmatcher().match_candidates(A, B, off, idx)
    ref = oracle.match_candidates(A, B, off, idx)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)


def _triangulation_case(seed, only_stereo):
    """Two keyframes seeing the same 3-D points from a translated camera; BoW-node
    candidate lists are simulated as random groups that contain the true match."""
    from orb_slam2_e_amd import KP_DTYPE
    rng = np.random.default_rng(seed)
    n = 600
    fx = fy = 500.0; cx, cy = 320.0, 240.0
    X = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1.5, 1.5, n), rng.uniform(4, 10, n)], 1)
    t = np.array([0.3, 0.02, 0.05])                                     # camera 2 = camera 1 shifted
    def proj(P):
        return np.stack([fx * P[:, 0] / P[:, 2] + cx, fy * P[:, 1] / P[:, 2] + cy], 1)
    p1, p2 = proj(X), proj(X - t)
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    # x1^T F12 x2 = 0 with X2 = X1 - t  (R = I):  F12 = K^-T [t]x K^-1
    F12 = (np.linalg.inv(K).T @ tx @ np.linalg.inv(K)).astype(np.float32)
    C2 = -t; ex = np.float32(fx * C2[0] / C2[2] + cx); ey = np.float32(fy * C2[1] / C2[2] + cy)
    k1 = np.zeros(n, KP_DTYPE); k2 = np.zeros(n, KP_DTYPE)
    k1["x"], k1["y"] = p1[:, 0] + rng.normal(0, 0.3, n), p1[:, 1] + rng.normal(0, 0.3, n)
    k2["x"], k2["y"] = p2[:, 0] + rng.normal(0, 0.3, n), p2[:, 1] + rng.normal(0, 0.3, n)
    k1["octave"] = rng.integers(0, 8, n); k2["octave"] = rng.integers(0, 8, n)
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d2 = d1 ^ np.packbits(rng.random((n, 256)) < 0.06, axis=1, bitorder="little")
    dup = rng.choice(n, 40, replace=False); d2[dup] = d2[(dup + 1) % n]   # equal-distance ties: later candidate wins
    off = [0]; idx = []
    for i in range(n):
        c = set(rng.integers(0, n, rng.integers(0, 12)).tolist()) | ({i} if rng.random() < 0.8 else set())
        c = list(c); rng.shuffle(c)
        idx += c; off.append(len(idx))
    mp1 = rng.random(n) < 0.2; mp2 = rng.random(n) < 0.2
    s1 = rng.random(n) < 0.3; s2 = rng.random(n) < 0.3
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32); sg = (sf * sf).astype(np.float32)
    return (k1, d1, k2, d2, off, idx, mp1, mp2, s1, s2, F12, ex, ey, sf.astype(np.float32), sg), only_stereo


@pytest.mark.parametrize("seed,only_stereo", [(0, False), (1, True), (2, False)])
def test_search_for_triangulation_inner_loop(seed, only_stereo):
    args, os_ = _triangulation_case(seed, only_stereo)
    got = ORBmatcher(0.6, False).match_triangulation(*args, bOnlyStereo=os_)
    ref = oracle.match_triangulation(*args, only_stereo=os_)
    assert (ref[0] >= 0).sum() > (20 if os_ else 100)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed,only_stereo,ori", [(0, False, True), (1, True, True), (2, False, False), (3, False, True)])
def test_search_for_triangulation_whole_function(seed, only_stereo, ori):
    """The whole SearchForTriangulation: FeatureVector co-iteration, gated loop, rotation histogram, pair list."""
    (k1, d1, k2, d2, _, _, mp1, mp2, s1, s2, F12, ex, ey, sf, sg), _ = _triangulation_case(seed, only_stereo)
    rng = np.random.default_rng(100 + seed)
    n = len(k1)
    node1 = rng.integers(0, 45, n) * 3 + 7
    node2 = np.where(rng.random(n) < 0.9, node1, rng.integers(0, 50, n) * 3 + 7)       # most true matches share a node
    node2[node2 == node1[0]] += 1 if seed == 3 else 0                                    # a node present on one side only
    k1["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k2["angle"] = np.where(rng.random(n) < 0.8, (k1["angle"] + rng.normal(0, 4, n)) % 360, rng.uniform(0, 360, n)).astype(np.float32)
    keep1 = rng.random(n) < 0.95; keep2 = rng.random(n) < 0.95                          # stopped words are in no node
    fv1, fv2 = oracle.feature_vector(node1, keep1), oracle.feature_vector(node2, keep2)
    m = ORBmatcher(0.6, ori)
    pairs, nm, m12 = m.SearchForTriangulation(k1, d1, fv1, mp1, s1, k2, d2, fv2, mp2, s2, F12, ex, ey, sf, sg, only_stereo)
    r12, rn = oracle.search_for_triangulation(k1, d1, fv1, mp1, s1, k2, d2, fv2, mp2, s2, F12, ex, ey, sf, sg, only_stereo, ori)
    assert rn > (10 if only_stereo else 60)
    assert nm == rn and np.array_equal(m12, r12)
    assert np.array_equal(pairs[:, 0], np.nonzero(r12 >= 0)[0]) and np.array_equal(pairs[:, 1], r12[r12 >= 0])
    if ori:
        assert rn < (oracle.search_for_triangulation(k1, d1, fv1, mp1, s1, k2, d2, fv2, mp2, s2, F12, ex, ey, sf, sg, only_stereo, False)[1])


@pytest.mark.parametrize("seed,only_stereo,ori", [(0, False, True), (1, True, True), (2, False, False), (3, False, True)])
def test_search_for_triangulation_on_resident_frames(seed, only_stereo, ori):
    """orbm_frame_search_for_triangulation: both keyframes resident (sorted by cell, stereo = right coordinate >= 0, some keypoints
    outside the grid), member lists shared per node, angle differences formed on the device -- equal to the host-array call and to
    the oracle's literal function."""
    from orb_slam2_e_amd import Frame
    (k1, d1, k2, d2, _, _, mp1, mp2, s1, s2, F12, ex, ey, sf, sg), _ = _triangulation_case(seed, only_stereo)
    rng = np.random.default_rng(100 + seed)
    n = len(k1)
    node1 = rng.integers(0, 45, n) * 3 + 7
    node2 = np.where(rng.random(n) < 0.9, node1, rng.integers(0, 50, n) * 3 + 7)
    node2[node2 == node1[0]] += 1 if seed == 3 else 0
    k1["angle"] = rng.uniform(0, 360, n).astype(np.float32)
    k2["angle"] = np.where(rng.random(n) < 0.8, (k1["angle"] + rng.normal(0, 4, n)) % 360, rng.uniform(0, 360, n)).astype(np.float32)
    keep1 = rng.random(n) < 0.95; keep2 = rng.random(n) < 0.95
    fv1, fv2 = oracle.feature_vector(node1, keep1), oracle.feature_vector(node2, keep2)
    bounds = (float(np.percentile(k1["x"], 5)), float(np.percentile(k1["y"], 5)), float(np.percentile(k1["x"], 95)),
              float(np.percentile(k1["y"], 95)))                       # tighter than the keypoints' extent: the rim is outside the grid
    ur1 = np.where(s1, 5.0, -1.0).astype(np.float32); ur2 = np.where(s2, 5.0, -1.0).astype(np.float32)
    f1, f2 = Frame(k1, d1, bounds, ur1), Frame(k2, d2, bounds, ur2)
    assert f1.layout()[0].size < n and f2.layout()[0].size < n
    m = ORBmatcher(0.6, ori)
    pairs, nm, m12 = m.frame_search_for_triangulation(f1, fv1, mp1, f2, fv2, mp2, F12, ex, ey, sf, sg, only_stereo)
    r12, rn = oracle.search_for_triangulation(k1, d1, fv1, mp1, s1, k2, d2, fv2, mp2, s2, F12, ex, ey, sf, sg, only_stereo, ori)
    assert rn > (10 if only_stereo else 60)
    assert nm == rn and np.array_equal(m12, r12)
    assert np.array_equal(pairs[:, 0], np.nonzero(r12 >= 0)[0]) and np.array_equal(pairs[:, 1], r12[r12 >= 0])
    # mono frames (no right coordinates): everything is "not stereo"
    g1, g2 = Frame(k1, d1, bounds), Frame(k2, d2, bounds)
    z = np.zeros(n, bool)
    _, nm0, m0 = m.frame_search_for_triangulation(g1, fv1, mp1, g2, fv2, mp2, F12, ex, ey, sf, sg, False)
    r0, rn0 = oracle.search_for_triangulation(k1, d1, fv1, mp1, z, k2, d2, fv2, mp2, z, F12, ex, ey, sf, sg, False, ori)
    assert nm0 == rn0 and np.array_equal(m0, r0)
    # a level table shorter than the frame's octaves is refused, not read past
    with pytest.raises(Exception):
        m.frame_search_for_triangulation(f1, fv1, mp1, f2, fv2, mp2, F12, ex, ey, sf[:3], sg[:3], only_stereo)
    for f in (f1, f2, g1, g2): f.close()


@pytest.mark.parametrize("seed,stereo,init", [(0, False, 256), (1, True, 256), (2, False, 2**31 - 1)])
def test_search_window_equals_grid_then_selection(seed, stereo, init):
    """orbm_search_window vs GetFeaturesInArea (grid order) + the SearchByProjection loop."""
    from orb_slam2_e_amd import KP_DTYPE
    rng = np.random.default_rng(seed)
    n, nq = 2000, 1500
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rng.uniform(-5, 645, n); kps["y"] = rng.uniform(-5, 485, n)   # some fall outside the grid
    kps["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc[rng.choice(n, 300, replace=False)] = desc[0]                        # many equal distances: order matters
    src = rng.integers(0, n, nq)
    q = np.zeros(nq, ORBmatcher.WQ_DTYPE)
    q["u"] = kps["x"][src] + rng.normal(0, 3, nq); q["v"] = kps["y"][src] + rng.normal(0, 3, nq)
    q["r"] = rng.choice([3.0, 4.5, 15.0, 40.0], nq) * (1.2 ** kps["octave"][src])
    lvl = kps["octave"][src]
    q["min_level"] = np.where(rng.random(nq) < 0.2, -1, lvl - 1); q["max_level"] = np.where(rng.random(nq) < 0.2, -1, lvl)
    q["xr"] = q["u"] - rng.uniform(0, 30, nq)
    qd = desc[src] ^ np.packbits(rng.random((nq, 256)) < 0.05, axis=1, bitorder="little")
    skip = (rng.random(n) < 0.1).astype(np.uint8)
    ur = np.where(rng.random(n) < 0.5, kps["x"] - rng.uniform(0, 30, n), -1).astype(np.float32) if stereo else None
    bounds = (0.0, 0.0, 640.0, 480.0)
    got = ORBmatcher().search_window(q, qd, kps, desc, bounds, skip, ur, init)
    ref = oracle.search_window(q, qd, kps, desc, bounds, skip, ur, init)
    assert (ref[4] >= 0).sum() > nq // 2
    for g, r, name in zip(got, ref, ("best", "best_level", "second", "second_level", "idx")):
        assert np.array_equal(g, r), name


@pytest.mark.parametrize("seed,th", [(0, 1.0), (1, 3.0)])
def test_whole_map_search_by_projection(seed, th):
    """The fork's relocalisation search over EVERY map point (ORBmatcher.cc:134-222):
    frustum test in mixed float/double, window search, TH_RELOC + ratio, last map point wins."""
    from orb_slam2_e_amd import KP_DTYPE
    rng = np.random.default_rng(seed)
    m, n = 5000, 2000
    fx = fy = 520.0; cx, cy = 320.0, 240.0
    ang = 0.1 * (seed + 1)
    Rcw = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    tcw = np.array([0.2, -0.1, 0.3])
    pos = np.stack([rng.uniform(-6, 6, m), rng.uniform(-4, 4, m), rng.uniform(-2, 12, m)], 1).astype(np.float32)
    Ow = -Rcw.T @ tcw
    view = pos - Ow
    dist = np.linalg.norm(view, axis=1)
    nrm = (view / dist[:, None] + rng.normal(0, 0.4, (m, 3))).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    maxd = (dist * rng.uniform(0.7, 3.0, m) / 1.2).astype(np.float32)   # GetMaxDistanceInvariance() = 1.2f*mfMaxDistance
    mind = (dist / rng.uniform(0.9, 4.5, m) / 0.8).astype(np.float32)
    mind_inv = (np.float32(0.8) * mind).astype(np.float32); maxd_inv = (np.float32(1.2) * maxd).astype(np.float32)
    mp_desc = rng.integers(0, 256, (m, 32), dtype=np.uint8)
    # frame keypoints: projections of a subset of the map points + clutter
    Pc = (Rcw @ pos.T).T + tcw
    uv = np.stack([fx * Pc[:, 0] / Pc[:, 2] + cx, fy * Pc[:, 1] / Pc[:, 2] + cy], 1)
    vis = np.where((Pc[:, 2] > 0.5) & (uv[:, 0] > 5) & (uv[:, 0] < 635) & (uv[:, 1] > 5) & (uv[:, 1] < 475))[0]
    pick = rng.choice(vis, min(n - 400, len(vis)), replace=False)
    kps = np.zeros(n, KP_DTYPE); desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    k = len(pick)
    kps["x"][:k] = uv[pick, 0] + rng.normal(0, 1.0, k); kps["y"][:k] = uv[pick, 1] + rng.normal(0, 1.0, k)
    kps["x"][k:] = rng.uniform(0, 640, n - k); kps["y"][k:] = rng.uniform(0, 480, n - k)
    kps["octave"] = rng.integers(0, 8, n)
    desc[:k] = mp_desc[pick] ^ np.packbits(rng.random((k, 256)) < 0.07, axis=1, bitorder="little")
    desc[k:k + 100] = desc[:100]                         # duplicated descriptors: several map points claim one keypoint / ties
    has_mp = rng.random(n) < 0.15
    cam = np.zeros(1, ORBmatcher.CAM_DTYPE)
    cam["fx"], cam["fy"], cam["cx"], cam["cy"] = fx, fy, cx, cy
    cam["min_x"], cam["max_x"], cam["min_y"], cam["max_y"] = 0, 640, 0, 480
    cam["gminx"], cam["gminy"], cam["gmaxx"], cam["gmaxy"] = 0, 0, 640, 480
    sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    got = ORBmatcher(0.6).SearchByProjectionMap(kps, desc, has_mp, pos, nrm, mind_inv, maxd_inv, mp_desc, Rcw, tcw, cam, sf, th)
    ref = oracle.search_by_projection_map(kps, desc, has_mp, pos, nrm, mind_inv, maxd_inv, mp_desc, Rcw, tcw, cam, sf, th)
    assert (ref[2][:, 3] >= 0).sum() > 300 and ref[1] > 50          # many map points in the frustum, many matches
    assert np.array_equal(got[2].view(np.uint32), ref[2].view(np.uint32))   # u, v, viewCos, level: float bits
    assert got[1] == ref[1] and np.array_equal(got[0], ref[0])


def test_distinctive_descriptors_batch():
    rng = np.random.default_rng(5)
    sizes = np.concatenate([[0, 1, 2, 3, 128, 129, 130, 333, 1000], rng.integers(1, 60, 400)])   # above 128: the row-by-row histogram path
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
    desc = np.concatenate([base[i] ^ np.packbits(rng.random((n, 256)) < 0.1, axis=1, bitorder="little") for i, n in enumerate(sizes) if n > 0])
    desc[off[10]:off[10] + 2] = desc[off[10] + 2]               # identical rows: first index wins the tie
    got = ORBmatcher.distinctive_descriptors(desc, off)
    ref = oracle.distinctive_descriptors(desc, off)
    assert got[0] == -1 and np.array_equal(got, ref)


@pytest.mark.parametrize("seed,th,ratio_lvl,ori,stereo", [(0, 95, False, True, False), (1, 95, True, False, True),
                                                          (2, 60, False, True, True), (3, 45, False, False, False)])
def test_search_projection_whole_loop(seed, th, ratio_lvl, ori, stereo, resolver):
    """orbm_search_projection vs the literal sequential loops (in-loop assignment + rotation check)."""
    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(seed, stereo=stereo)
    m = ORBmatcher(0.8 if ratio_lvl else 0.6, ori)
    got = m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, th, ratio_lvl)
    ref = oracle.search_projection_seq(q, qd, qa, takes, kps, desc, bounds, occ, ur, th, m.mfNNratio, ratio_lvl, ori)
    # the coupling is exercised: an independent-query evaluation gives a different answer
    free = oracle.search_projection_seq(q, qd, qa, np.zeros_like(takes), kps, desc, bounds, occ, ur, th, m.mfNNratio, ratio_lvl, ori)
    assert ref[2] > 200 and not np.array_equal(free[1], ref[1])
    if ori:
        assert (ref[0] == -2).sum() > 0
    assert got[2] == ref[2]
    assert np.array_equal(got[1], ref[1]) and np.array_equal(got[0], ref[0])


def test_search_projection_edge_cases(resolver):
    from orb_slam2_e_amd import KP_DTYPE
    m = ORBmatcher(0.6, True)
    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(5, n=300, nq=100, hot=20)
    # no queries / no keypoints
    mk, mq, nm = m.search_projection(q[:0], qd[:0], qa[:0], takes[:0], kps, desc, bounds)
    assert nm == 0 and (mk == -1).all() and len(mq) == 0
    mk, mq, nm = m.search_projection(q, qd, qa, takes, kps[:0], desc[:0], bounds)
    assert nm == 0 and (mq == -1).all()
    # qtakes = NULL means every accepted query blocks its keypoint
    got = m.search_projection(q, qd, qa, None, kps, desc, bounds)
    ref = oracle.search_projection_seq(q, qd, qa, np.ones(len(q), np.uint8), kps, desc, bounds)
    assert got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # one oversized candidate list (> the LDS staging buffer): every keypoint in one window, many queries on it
    n = 8000
    rng = np.random.default_rng(9)
    kb = np.zeros(n, KP_DTYPE); kb["x"] = rng.uniform(100, 140, n); kb["y"] = rng.uniform(100, 140, n)
    kb["angle"] = rng.uniform(0, 360, n)
    db = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    nq = 40
    qb = np.zeros(nq, ORBmatcher.WQ_DTYPE); qb["u"] = 120; qb["v"] = 120; qb["r"] = 30; qb["min_level"] = -1; qb["max_level"] = -1
    src = rng.integers(0, 50, nq)
    qdb = db[src] ^ np.packbits(rng.random((nq, 256)) < 0.03, axis=1, bitorder="little")
    qab = rng.uniform(0, 360, nq).astype(np.float32)
    got = m.search_projection(qb, qdb, qab, None, kb, db, bounds)
    ref = oracle.search_projection_seq(qb, qdb, qab, np.ones(nq, np.uint8), kb, db, bounds)
    assert ref[2] > 5 and got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed,ratio,ori,window,n1,alike", [(0, 0.9, True, 100, 2000, 3), (1, 0.9, False, 30, 2000, 3), (2, 0.7, True, 100, 2000, 3),
                                                            (3, 0.95, True, 100, 2600, 40), (4, 1.0, False, 640, 1500, 1)])
def test_search_for_initialization(seed, ratio, ori, window, n1, alike, resolver):
    """ORBmatcher::SearchForInitialization whole (steal rule, vMatchedDistance gate, rotation check, prev update), on the parallel
    fixed point (k_resolve_init_par) and on the one-wave sequential resolver.  n1 = 2600: more queries than the fixed point keeps in
    registers; `alike` = 1: 600 look-alike queries after ONE keypoint each of frame 2 -- more acceptors than a slot's list holds,
    the call repeats itself on the sequential resolver; window 640: every list outgrows its region (the exact path)."""
    from orb_slam2_e_amd import KP_DTYPE
    rng = np.random.default_rng(seed)
    n2 = 2200
    k1 = np.zeros(n1, KP_DTYPE)
    k1["x"] = rng.uniform(0, 640, n1); k1["y"] = rng.uniform(0, 480, n1)
    k1["octave"] = rng.choice(8, n1, p=[0.5, 0.15, 0.1, 0.08, 0.07, 0.05, 0.03, 0.02]); k1["angle"] = rng.uniform(0, 360, n1)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d1[rng.choice(n1, 600, replace=False)] = d1[:alike][rng.integers(0, alike, 600)]  # look-alikes compete for the same F2 keypoints
    src = rng.integers(0, n1, n2)
    k2 = k1[src].copy()
    k2["x"] += rng.normal(0, 6, n2); k2["y"] += rng.normal(0, 6, n2)
    k2["angle"] = (k2["angle"] + rng.choice([20.0, 140.0, 250.0], n2, p=[0.7, 0.2, 0.1]) + rng.normal(0, 3, n2)) % 360
    d2 = d1[src] ^ np.packbits(rng.random((n2, 256)) < 0.03, axis=1, bitorder="little")
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    bounds = (0.0, 0.0, 640.0, 480.0)
    m = ORBmatcher(ratio, ori)
    got = m.SearchForInitialization(k1, d1, k2, d2, prev, bounds, window)
    ref = oracle.search_for_initialization(k1, d1, k2, d2, prev, bounds, window, ratio, ori)
    assert ref[2] > 100
    assert got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


@pytest.mark.parametrize("seed,kf_kf,ori,ratio", [(0, False, True, 0.7), (1, True, True, 0.75), (2, False, False, 0.6), (3, True, False, 0.9)])
def test_search_by_bow_whole_loop(seed, kf_kf, ori, ratio, resolver):
    """orbm_search_by_bow (+ the host co-iteration) vs the literal SearchByBoW loops, on the parallel fixed point (all nodes in one
    workgroup: queries of different nodes never meet) and on the sequential resolver (one workgroup per node)."""
    from orb_slam2_e_amd.vocabulary import feature_vector_arrays
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(seed)
    fv1 = feature_vector_arrays(node1, keep1); fv2 = feature_vector_arrays(node2, keep2)
    got = ORBmatcher(ratio, ori).SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, kf_kf)
    ref = oracle.search_by_bow(oracle.feature_vector(node1, keep1), valid1, d1, a1, oracle.feature_vector(node2, keep2),
                               valid2 if kf_kf else None, d2, a2, kf_kf, ratio, ori)
    assert ref[2] > 300
    assert got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    # the skip of already matched features is exercised: some query's unconstrained best was taken before it
    best = oracle.match_bruteforce(d1, d2)[2]
    taken = sum(1 for i in np.nonzero(ref[0] >= 0)[0] if ref[0][i] != best[i])
    assert taken > 0


@pytest.mark.parametrize("seed,kf_kf,ori,ratio", [(0, False, True, 0.7), (1, True, True, 0.75), (3, True, False, 0.9)])
def test_search_by_bow_on_resident_frames(seed, kf_kf, ori, ratio, resolver):
    """orbm_frame_search_by_bow: both frames resident, queries gathered on the device, features addressed by index -- including the
    ones that lie outside the frame's grid (Frame::PosInGrid false), which no window search returns but SearchByBoW must see."""
    from orb_slam2_e_amd import KP_DTYPE, Frame
    from orb_slam2_e_amd.vocabulary import feature_vector_arrays
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(seed)
    rng = np.random.default_rng(100 + seed)
    bounds = (-3.5, -2.25, 645.5, 483.0)

    def keypoints(angles):
        n = len(angles)
        k = np.zeros(n, KP_DTYPE)
        k["x"] = rng.uniform(0, 640, n); k["y"] = rng.uniform(0, 480, n); k["octave"] = rng.integers(0, 8, n); k["angle"] = angles
        out = rng.choice(n, 40, replace=False)                       # undistorted past the corner-based bounds
        k["x"][out[:20]] = rng.choice([-9.0, 652.0], 20); k["y"][out[20:]] = rng.choice([-7.5, 490.0], 20)
        return k
    k1, k2 = keypoints(a1), keypoints(a2)
    f1, f2 = Frame(k1, d1, bounds), Frame(k2, d2, bounds)
    assert f1.layout()[0].size < len(k1) and f2.layout()[0].size < len(k2)      # some keypoints are outside the grid
    fv1 = feature_vector_arrays(node1, keep1); fv2 = feature_vector_arrays(node2, keep2)
    m = ORBmatcher(ratio, ori)
    got = m.frame_search_by_bow(f1, fv1, valid1, f2, fv2, valid2, kf_kf)
    host = m.SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, kf_kf)
    ref = oracle.search_by_bow(oracle.feature_vector(node1, keep1), valid1, d1, a1, oracle.feature_vector(node2, keep2),
                               valid2 if kf_kf else None, d2, a2, kf_kf, ratio, ori)
    assert ref[2] > 300
    for r in (host, got):
        assert r[2] == ref[2] and np.array_equal(r[0], ref[0]) and np.array_equal(r[1], ref[1])
    # the frames still serve window searches: a query over everything never returns an outside keypoint
    q = np.zeros(1, ORBmatcher.WQ_DTYPE); q["u"] = 320; q["v"] = 240; q["r"] = 2000; q["min_level"] = -1; q["max_level"] = -1
    outside = np.setdiff1d(np.arange(len(k2)), f2.layout()[0])
    qo = np.repeat(q, len(outside))
    mk, mq, nm = ORBmatcher(1.0, False).frame_search_projection(f2, qo, d2[outside], a2[outside], None, None, 256)
    assert len(outside) >= 30 and (mq >= 0).any() and not np.isin(mq[mq >= 0], outside).any()   # their own descriptors do not find them
    # empty feature vectors / an empty first frame
    e = (np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    g0 = m.frame_search_by_bow(f1, e, valid1, f2, fv2, valid2, kf_kf)
    assert g0[2] == 0 and (g0[0] == -1).all() and (g0[1] == -1).all()
    f0 = Frame(k1[:0], d1[:0], bounds)
    g1 = m.frame_search_by_bow(f0, e, valid1[:0], f2, fv2, valid2, kf_kf)
    assert g1[2] == 0 and len(g1[0]) == 0
    for f in (f0, f1, f2): f.close()


def test_matcher_calls_are_reentrant_across_threads():
    """Tracking, LocalMapping and LoopClosing call ORBmatcher concurrently: the host-array entry
    points lease separate workspaces, so concurrent calls must return what sequential calls return."""
    import threading
    cases = [synth_projection_case(s, n=1500, nq=1500, hot=300) for s in range(4)]
    m = ORBmatcher(0.6, True)
    want = [m.search_projection(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], 95) for c in cases]
    want_w = [m.search_window(c[0], c[1], c[4], c[5], c[6], c[7], c[8]) for c in cases]
    errors = []

    def worker(k):
        try:
            c = cases[k]
            for _ in range(10):
                got = ORBmatcher(0.6, True).search_projection(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], 95)
                assert got[2] == want[k][2] and np.array_equal(got[0], want[k][0]) and np.array_equal(got[1], want[k][1])
                gw = ORBmatcher().search_window(c[0], c[1], c[4], c[5], c[6], c[7], c[8])
                assert all(np.array_equal(a, b) for a, b in zip(gw, want_w[k]))
                best, second, idx = ORBmatcher().match_bruteforce(c[1][:500], c[5])
                assert np.array_equal(idx, oracle.match_bruteforce(c[1][:500], c[5])[2])
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert not errors, errors


@pytest.mark.parametrize("seed,stereo,gate", [(0, False, True), (1, True, True), (2, True, False)])
def test_fuse_candidate_loop(seed, stereo, gate):
    """orbm_search_fuse vs the literal candidate loop of ORBmatcher::Fuse (reprojection gate 5.99 / 7.8)."""
    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(seed, n=2500, nq=2500, hot=2500, stereo=stereo)
    q["r"] = 3.0 * 1.2 ** (q["max_level"].clip(0, 7))             # th * scale[level]: a few px, as Fuse uses it
    q["min_level"] = q["max_level"] - 1
    if ur is not None:
        ur = np.where(ur < 0, -1.0, ur).astype(np.float32)
        q["xr"] = np.where(ur[np.arange(len(q)) % len(kps)] >= 0, q["u"] - (kps["x"] - ur)[np.arange(len(q)) % len(kps)], q["xr"])
    sg = (1.0 / (1.2 ** np.arange(8)) ** 2).astype(np.float32) if gate else None
    got = ORBmatcher().search_fuse(q, qd, kps, desc, bounds, ur, sg)
    ref = oracle.search_fuse(q, qd, kps, desc, bounds, ur, sg)
    assert (ref[1] >= 0).sum() > 300
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    if gate:  # the gate removes candidates the plain window search would take
        free = oracle.search_fuse(q, qd, kps, desc, bounds, ur, None)
        assert not np.array_equal(free[1], ref[1])


def test_search_by_sim3_agreement(resolver):
    """SearchBySim3 = two window searches + the mutual-agreement check, against the oracle composition."""
    from orb_slam2_e_amd import KP_DTYPE
    rng = np.random.default_rng(4)
    n = 1800
    k1 = np.zeros(n, KP_DTYPE); k1["x"] = rng.uniform(10, 630, n); k1["y"] = rng.uniform(10, 470, n); k1["octave"] = rng.integers(0, 8, n)
    perm = rng.permutation(n)
    k2 = k1[perm].copy(); k2["x"] += rng.normal(0, 2, n); k2["y"] += rng.normal(0, 2, n)
    d1 = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    d1[rng.choice(n, 300, replace=False)] = d1[0]                   # look-alikes break the agreement for some
    d2 = d1[perm] ^ np.packbits(rng.random((n, 256)) < 0.04, axis=1, bitorder="little")
    inv = np.argsort(perm)
    def queries(src_k, dst_k, dst_of_src):
        q = np.zeros(n, ORBmatcher.WQ_DTYPE)
        q["u"] = dst_k["x"][dst_of_src] + rng.normal(0, 1.5, n); q["v"] = dst_k["y"][dst_of_src] + rng.normal(0, 1.5, n)
        q["r"] = np.where(rng.random(n) < 0.1, -1.0, 7.5 * 1.2 ** src_k["octave"])   # 10 % skipped points
        q["max_level"] = src_k["octave"] + rng.integers(0, 2, n); q["min_level"] = q["max_level"] - 1
        return q
    q12 = queries(k1, k2, inv); q21 = queries(k2, k1, perm)
    bounds = (0.0, 0.0, 640.0, 480.0)
    got = ORBmatcher().SearchBySim3(q12, d1, k2, d2, q21, d2, k1, d1, bounds)
    ref = oracle.search_by_sim3(q12, d1, k2, d2, q21, d2, k1, d1, bounds)
    assert ref[1] > 500 and got[1] == ref[1] and np.array_equal(got[0], ref[0])
