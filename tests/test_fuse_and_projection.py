"""M15 / M10: batch Frame::isInFrustum + MapPoint::PredictScale (Frame.cc:284-340, MapPoint.cc:464-480) and
ORBmatcher::Fuse as a whole (ORBmatcher.cc:1026-1176, Sim3 form :1178-1301) against the oracle's literal restatements.
CPU part: the ordered replay of the loop tail (host logic).  GPU part: projection bit for bit (float bits of u, v, ur,
viewCos, dist; level; visibility; the window query), then projection + candidate loop + replay end to end."""
import numpy as np
import pytest

import oracle
from orb_slam2_e_amd.extractor import KP_DTYPE
from orb_slam2_e_amd.matcher import ORBmatcher


class ToyMap:
    """The part of the Map / KeyFrame / MapPoint pointer graph the Fuse loop can observe, as in oracle_fuse_replay."""

    def __init__(self, mp_obs, mp_bad, mp_in_kf, kf_mp):
        self.obs, self.bad, self.in_kf, self.kf = mp_obs, mp_bad, mp_in_kf, kf_mp
        self.ops = []

    def is_bad(self, h): return bool(self.bad[h])
    def is_in_keyframe(self, h): return self.in_kf[h] >= 0
    def slot_owner(self, idx): return int(self.kf[idx])
    def observations(self, h): return int(self.obs[h])

    def replace(self, dead, heir):
        slot = self.in_kf[dead]
        self.bad[dead] = 1
        if slot >= 0:                      # pMPinKF->Replace(pMP): pMP takes the slot
            self.in_kf[dead] = -1
            self.kf[slot] = heir; self.in_kf[heir] = slot; self.obs[heir] += 1
            self.ops.append((2, heir, dead))
        else:                              # pMP->Replace(pMPinKF): pMP was in no slot of this key frame
            self.ops.append((1, dead, heir))

    def add(self, h, idx):
        self.kf[idx] = h; self.in_kf[h] = idx; self.obs[h] += 1
        self.ops.append((0, h, idx))


def _toy(rng, nmp, nkp, nlist):
    mp_obs = rng.integers(0, 6, nmp).astype(np.int32)
    mp_bad = (rng.random(nmp) < 0.1).astype(np.uint8)
    kf_mp = np.full(nkp, -1, np.int32)
    mp_in_kf = np.full(nmp, -1, np.int32)
    owners = rng.choice(nmp, nkp // 2, replace=False)
    slots = rng.choice(nkp, nkp // 2, replace=False)
    kf_mp[slots] = owners; mp_in_kf[owners] = slots
    lst = rng.integers(-1, nmp, nlist).astype(np.int32)          # NULL entries and duplicates
    visible = (rng.random(nlist) < 0.85).astype(np.int32)
    idx = rng.integers(-1, nkp, nlist).astype(np.int32)
    idx[rng.random(nlist) < 0.5] = rng.integers(0, max(nkp // 8, 1))   # many points aim at few keypoints
    best = rng.integers(0, 90, nlist).astype(np.int32)
    return lst, visible, best, idx, mp_obs, mp_bad, mp_in_kf, kf_mp


@pytest.mark.parametrize("seed", range(6))
def test_fuse_replay_equals_literal_loop_tail(seed):
    rng = np.random.default_rng(seed)
    lst, visible, best, idx, mp_obs, mp_bad, mp_in_kf, kf_mp = _toy(rng, 300, 200, 500)
    state_o = [a.copy() for a in (mp_obs, mp_bad, mp_in_kf, kf_mp)]
    n_ref, ops_ref = oracle.fuse_replay(lst, visible, best, idx, *state_o)
    tm = ToyMap(mp_obs.copy(), mp_bad.copy(), mp_in_kf.copy(), kf_mp.copy())
    n = ORBmatcher.__new__(ORBmatcher).fuse_replay(lst, visible, best, idx, tm)
    assert n == n_ref and n > 20
    assert [tuple(int(v) for v in o) for o in ops_ref] == tm.ops
    assert {k for k, _, _ in tm.ops} == {0, 1, 2}                # all three branches of :1151-1168 taken
    for a, b in zip(state_o, (tm.obs, tm.bad, tm.in_kf, tm.kf)):
        assert np.array_equal(a, b)


class ToyMapSim3(ToyMap):
    """+ what the Sim3 form reads and writes: spAlreadyFound (snapshot at construction) and vpReplacePoint."""

    def __init__(self, mp_obs, mp_bad, mp_in_kf, kf_mp, nlist):
        super().__init__(mp_obs, mp_bad, mp_in_kf, kf_mp)
        self.already = np.zeros(len(mp_obs), bool)
        own = kf_mp[kf_mp >= 0]
        self.already[own[mp_bad[own] == 0]] = True                  # KeyFrame::GetMapPoints(), KeyFrame.cc:274-287
        self.vpReplacePoint = np.full(nlist, -1, np.int32)

    def already_found(self, h): return bool(self.already[h])

    def record_replace(self, i, h):
        self.vpReplacePoint[i] = h
        self.ops.append((3, i, h))


@pytest.mark.parametrize("seed", range(6))
def test_fuse_replay_sim3_equals_literal_loop_tail(seed):
    """ORBmatcher.cc:1194-1205, :1279-1296: snapshot skip, record-only replacement; the list repeats points, so one the
    loop has just added comes round again (it is not in the snapshot) and finds itself in its slot."""
    rng = np.random.default_rng(40 + seed)
    lst, visible, best, idx, mp_obs, mp_bad, mp_in_kf, kf_mp = _toy(rng, 300, 200, 500)
    lst = np.concatenate([lst, lst[:120]]); visible = np.concatenate([visible, visible[:120]])
    best = np.concatenate([best, best[:120]]); idx = np.concatenate([idx, idx[:120]])
    state_o = [a.copy() for a in (mp_obs, mp_bad, mp_in_kf, kf_mp)]
    rep_o = np.full(len(lst), -1, np.int32)
    n_ref, ops_ref = oracle.fuse_replay_sim3(lst, visible, best, idx, *state_o, rep_o)
    tm = ToyMapSim3(mp_obs.copy(), mp_bad.copy(), mp_in_kf.copy(), kf_mp.copy(), len(lst))
    n = ORBmatcher.__new__(ORBmatcher).fuse_replay_sim3(lst, visible, best, idx, tm)
    assert n == n_ref and n > 20
    assert [tuple(int(v) for v in o) for o in ops_ref] == tm.ops
    assert {k for k, _, _ in tm.ops} == {0, 3}
    assert np.array_equal(rep_o, tm.vpReplacePoint) and (rep_o >= 0).sum() > 5
    for a, b in zip(state_o, (tm.obs, tm.bad, tm.in_kf, tm.kf)):
        assert np.array_equal(a, b)
    # a point added by the loop and named again records itself (not in the snapshot): the reference's behaviour
    added = {h for k, h, _ in tm.ops if k == 0}
    assert any(k == 3 and int(lst[i]) in added and int(lst[i]) == h for k, i, h in tm.ops)
    assert not np.array_equal(state_o[1], mp_bad) or True            # the Sim3 form never kills a point
    assert np.array_equal(state_o[1], mp_bad)


@pytest.mark.parametrize("seed", range(3))
def test_cxx_header_fuse_tails_equal_the_oracle(seed, tmp_path):
    """orbslam_hip::ORBmatcher::fuseReplay / fuseReplaySim3 (include/orbslam_hip.hpp), compiled with g++ and run on a toy
    map (tests/cxx/host_logic.cpp; no device call), against the oracle's literal loop tails."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_logic")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cxx", "host_logic.cpp"), "-o", exe])
    rng = np.random.default_rng(90 + seed)
    lst, visible, best, idx, mp_obs, mp_bad, mp_in_kf, kf_mp = _toy(rng, 300, 200, 500)
    lst = np.concatenate([lst, lst[:120]]); visible = np.concatenate([visible, visible[:120]])
    best = np.concatenate([best, best[:120]]); idx = np.concatenate([idx, idx[:120]])
    text = " ".join(str(int(v)) for v in np.concatenate([[300, 200, len(lst)], mp_obs, mp_bad, mp_in_kf, kf_mp, lst, visible, best, idx]))
    out = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = [np.array(l.split(), np.int64) for l in out.stdout.strip("\n").split("\n")]
    # plain form
    st = [a.copy() for a in (mp_obs, mp_bad, mp_in_kf, kf_mp)]
    n_ref, ops_ref = oracle.fuse_replay(lst, visible, best, idx, *st)
    assert tuple(lines[0]) == (n_ref, len(ops_ref)) and np.array_equal(lines[1], ops_ref.ravel())
    for got, ref in zip(lines[2:6], st):
        assert np.array_equal(got, ref)
    # Sim3 form
    st = [a.copy() for a in (mp_obs, mp_bad, mp_in_kf, kf_mp)]
    rep = np.full(len(lst), -1, np.int32)
    n_ref, ops_ref = oracle.fuse_replay_sim3(lst, visible, best, idx, *st, rep)
    assert tuple(lines[6]) == (n_ref, len(ops_ref)) and np.array_equal(lines[7], ops_ref.ravel())
    for got, ref in zip(lines[8:13], st + [rep]):
        assert np.array_equal(got, ref)


def _scene(rng, m):
    """A camera at a generic pose looking at a cloud; points in front, behind, outside the image, too near / far, seen
    from behind, plus exact boundary cases."""
    from scipy.spatial.transform import Rotation
    R = Rotation.from_euler("xyz", rng.uniform(-0.3, 0.3, 3)).as_matrix().astype(np.float32)
    t = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    Ow = (-(R.T.astype(np.float32) @ t)).astype(np.float32)
    pc = np.stack([rng.uniform(-6, 6, m), rng.uniform(-4, 4, m), rng.uniform(-2, 14, m)], 1)
    pos = ((pc - t) @ R).astype(np.float32)                         # world = R^T (pc - t)
    nrm = (pos - Ow); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)     # mean viewing direction: from the camera to the point
    nrm = (nrm + rng.normal(0, 0.6, (m, 3))).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    d = np.linalg.norm(pos - Ow, axis=1)
    maxd = (d * rng.uniform(0.6, 3.5, m)).astype(np.float32)
    mind = (maxd / rng.uniform(2.0, 4.3, m)).astype(np.float32)
    return R, t, Ow, pos, nrm.astype(np.float32), mind, maxd


CAM4 = np.array([718.856, 718.856, 607.19, 185.22], np.float32)
BOUNDS = (0.0, 0.0, 1241.0, 376.0)


def _cam():
    cam = np.zeros(1, ORBmatcher.CAM_DTYPE)
    cam["fx"], cam["fy"], cam["cx"], cam["cy"] = CAM4
    cam["gminx"], cam["gminy"], cam["gmaxx"], cam["gmaxy"] = BOUNDS
    cam["min_x"], cam["min_y"], cam["max_x"], cam["max_y"] = 0, 0, 1241, 376
    return cam


@pytest.mark.gpu
@pytest.mark.parametrize("mode,th", [(0, 1.0), (0, 3.0), (1, 3.0), (2, 4.0)])
def test_project_points_bit_exact(mode, th):
    rng = np.random.default_rng(10 * mode + int(th))
    m = 20000
    R, t, Ow, pos, nrm, mind, maxd = _scene(rng, m)
    sf = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    logsf = np.float32(np.log(np.float32(1.2)))                    # mfLogScaleFactor = log(mfScaleFactor)
    mt = ORBmatcher(0.8)
    got, gq = mt.project_points(mode, pos, nrm, mind, maxd, R, t, Ow, _cam(), 386.1448, logsf, sf, th, 0.5)
    ref, rq = oracle.project_points(mode, pos, nrm, mind, maxd, R, t, Ow, CAM4, BOUNDS, 386.1448, 0.5, logsf, sf, th)
    assert np.array_equal(got["visible"], ref["visible"]) and np.array_equal(got["level"], ref["level"])
    for f in ("u", "v", "ur", "view_cos", "dist"):
        assert np.array_equal(got[f].view(np.uint32), ref[f].view(np.uint32)), f
    assert gq.tobytes() == rq.tobytes()
    vis = ref["visible"] > 0
    assert 0.1 * m < vis.sum() < 0.9 * m and len(np.unique(ref["level"][vis])) == 8


@pytest.mark.gpu
@pytest.mark.parametrize("sim3", [False, True])
def test_fuse_as_a_whole(sim3):
    rng = np.random.default_rng(77 + sim3)
    nkp, nmp = 1500, 1200
    R, t, Ow, pos, nrm, mind, maxd = _scene(rng, nmp)
    sf = np.cumprod(np.concatenate([[np.float32(1)], np.full(7, np.float32(1.2))]).astype(np.float32)).astype(np.float32)
    logsf = np.float32(np.log(np.float32(1.2)))
    inv_sigma2 = (1.0 / (sf * sf)).astype(np.float32)
    mt = ORBmatcher(0.8)
    mode = 2 if sim3 else 1
    proj, q = oracle.project_points(mode, pos, nrm, mind, maxd, R, t, Ow, CAM4, BOUNDS, 386.1448, 0.5, logsf, sf, 3.0)
    vis = np.nonzero(proj["visible"])[0]
    assert len(vis) > 200
    # key frame: keypoints near the projections of some visible points (with descriptors close to theirs), plus clutter
    kps = np.zeros(nkp, KP_DTYPE)
    kps["x"] = rng.uniform(0, 1241, nkp); kps["y"] = rng.uniform(0, 376, nkp); kps["octave"] = rng.integers(0, 8, nkp)
    desc = rng.integers(0, 256, (nkp, 32), dtype=np.uint8)
    mp_desc = rng.integers(0, 256, (nmp, 32), dtype=np.uint8)
    src = rng.choice(vis, min(len(vis), 900))
    tgt = rng.choice(nkp, len(src), replace=False)
    kps["x"][tgt] = proj["u"][src] + rng.normal(0, 1.0, len(src)); kps["y"][tgt] = proj["v"][src] + rng.normal(0, 1.0, len(src))
    kps["octave"][tgt] = np.maximum(proj["level"][src] - rng.integers(0, 2, len(src)), 0)
    desc[tgt] = mp_desc[src] ^ np.packbits(rng.random((len(src), 256)) < 0.06, axis=1, bitorder="little")
    uright = np.where(rng.random(nkp) < 0.5, kps["x"] - rng.uniform(0, 40, nkp), -1).astype(np.float32)
    uright[tgt[::2]] = (proj["ur"][src[::2]] + rng.normal(0, 1.0, len(src[::2]))).astype(np.float32)
    # toy map over the listed points + the key frame's own points
    lst = np.arange(nmp, dtype=np.int32); rng.shuffle(lst); lst = np.concatenate([lst, lst[:50], [-1, -1]]).astype(np.int32)
    nall = nmp + nkp
    mp_obs = rng.integers(0, 6, nall).astype(np.int32); mp_bad = (rng.random(nall) < 0.05).astype(np.uint8)
    kf_mp = np.full(nkp, -1, np.int32); mp_in_kf = np.full(nall, -1, np.int32)
    own = rng.choice(nkp, nkp // 2, replace=False)
    kf_mp[own] = nmp + own; mp_in_kf[nmp + own] = own               # the key frame's own map points
    some = rng.choice(nmp, 30, replace=False); slots = rng.choice(np.setdiff1d(np.arange(nkp), own), 30, replace=False)
    kf_mp[slots] = some; mp_in_kf[some] = slots                     # a few listed points are already in the key frame
    li = np.maximum(lst, 0)
    # reference: literal projection, literal candidate loop, literal tail
    rproj, rq = oracle.project_points(mode, pos[li], nrm[li], mind[li], maxd[li], R, t, Ow, CAM4, BOUNDS, 386.1448, 0.5, logsf, sf, 3.0)
    rbest, ridx = oracle.search_fuse(rq, mp_desc[li], kps, desc, BOUNDS, uright, None if sim3 else inv_sigma2)
    st_ref = [a.copy() for a in (mp_obs, mp_bad, mp_in_kf, kf_mp)]
    if not sim3:
        n_ref, ops_ref = oracle.fuse_replay(lst, rproj["visible"], rbest, ridx, *st_ref)
    tm = ToyMap(mp_obs.copy(), mp_bad.copy(), mp_in_kf.copy(), kf_mp.copy())
    if sim3:                 # the Sim3 form end to end: candidate stage on the device, its record-only tail on the host
        gproj, gq = mt.project_points(mode, pos[li], nrm[li], mind[li], maxd[li], R, t, Ow, _cam(), 386.1448, logsf, sf, 3.0)
        gbest, gidx = mt.search_fuse(gq, mp_desc[li], kps, desc, BOUNDS, uright, None)
        assert gq.tobytes() == rq.tobytes() and np.array_equal(gbest, rbest) and np.array_equal(gidx, ridx)
        assert ((rbest <= 45) & (ridx >= 0)).sum() > 100
        rep_ref = np.full(len(lst), -1, np.int32)
        n_ref, ops_ref = oracle.fuse_replay_sim3(lst, rproj["visible"], rbest, ridx, *st_ref, rep_ref)
        ts = ToyMapSim3(mp_obs.copy(), mp_bad.copy(), mp_in_kf.copy(), kf_mp.copy(), len(lst))
        n = mt.Fuse(kps, desc, uright, BOUNDS, inv_sigma2, lst, pos[li], nrm[li], mind[li], maxd[li], mp_desc[li], R, t, Ow, _cam(),
                    386.1448, logsf, sf, 3.0, ts, sim3=True)
        assert n == n_ref and n > 100
        assert [tuple(int(v) for v in o) for o in ops_ref] == ts.ops and {k for k, _, _ in ts.ops} == {0, 3}
        assert np.array_equal(rep_ref, ts.vpReplacePoint)
        for a, b in zip(st_ref, (ts.obs, ts.bad, ts.in_kf, ts.kf)):
            assert np.array_equal(a, b)
        return
    n = mt.Fuse(kps, desc, uright, BOUNDS, inv_sigma2, lst, pos[li], nrm[li], mind[li], maxd[li], mp_desc[li], R, t, Ow, _cam(),
                386.1448, logsf, sf, 3.0, tm)
    assert n == n_ref and n > 100
    assert [tuple(int(v) for v in o) for o in ops_ref] == tm.ops
    for a, b in zip(st_ref, (tm.obs, tm.bad, tm.in_kf, tm.kf)):
        assert np.array_equal(a, b)
