"""Differential fuzz of the searches that address a RESIDENT frame's features by index (orbm_frame_search_by_bow,
orbm_frame_search_for_triangulation) vs the literal oracle loops: random sizes (down to one keypoint), random grid bounds (so that a
random share of the keypoints lies outside the grid), random stereo masks, both BoW forms, both resolvers.
usage (GPU box): PYTHONPATH=. python tests/fuzz_frames.py [cases] [seed]"""
import sys
import numpy as np
import oracle
from orb_slam2_e_amd import KP_DTYPE, Frame, ORBmatcher
from orb_slam2_e_amd._lib import lib
from orb_slam2_e_amd.synth import synth_bow_case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0; tot_bow = 0; tot_tri = 0
sf = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32); sg = (sf * sf).astype(np.float32)


def keypoints(count, angles):
    k = np.zeros(count, KP_DTYPE)
    k["x"] = rng.uniform(0, 640, count); k["y"] = rng.uniform(0, 480, count); k["octave"] = rng.integers(0, 8, count); k["angle"] = angles
    return k


for case in range(n):
    n1 = int(rng.integers(1, 2500)); n2 = int(rng.integers(1, 2500)); nn = int(rng.integers(1, 200))
    d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(int(rng.integers(0, 1 << 30)), n1=n1, n2=n2, nnodes=nn)
    k1, k2 = keypoints(n1, a1), keypoints(n2, a2)
    lo = float(rng.uniform(-20, 200)); hi = float(rng.uniform(lo + 50, 700))
    bounds = (lo, lo * 0.7, hi, hi * 0.75 + 40)
    s1 = rng.random(n1) < rng.uniform(0, 1); s2 = rng.random(n2) < rng.uniform(0, 1)
    f1 = Frame(k1, d1, bounds, np.where(s1, 3.0, -1.0).astype(np.float32)); f2 = Frame(k2, d2, bounds, np.where(s2, 3.0, -1.0).astype(np.float32))
    kf = bool(rng.integers(0, 2)); ori = bool(rng.integers(0, 2)); ratio = float(rng.choice([0.6, 0.75, 0.9]))
    lib().orbm_debug_force_sequential_resolver(int(rng.integers(0, 2)))
    m = ORBmatcher(ratio, ori)
    fv1, fv2 = oracle.feature_vector(node1, keep1), oracle.feature_vector(node2, keep2)
    got = m.frame_search_by_bow(f1, fv1, valid1, f2, fv2, valid2, kf)
    ref = oracle.search_by_bow(fv1, valid1, d1, a1, fv2, valid2 if kf else None, d2, a2, kf, ratio, ori)
    tot_bow += int(ref[2])
    if not (got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
        bad += 1; print("MISMATCH frame bow", case, n1, n2, nn, kf, ori, ratio, bounds, flush=True)
    # triangulation: random F12 scaled so that a fair share of the candidates passes the epipolar gate
    F12 = (rng.normal(0, 1, (3, 3)) * np.array([[1e-6, 1e-6, 1e-3], [1e-6, 1e-6, 1e-3], [1e-3, 1e-3, 0.3]])).astype(np.float32)
    ex, ey = np.float32(rng.uniform(-100, 740)), np.float32(rng.uniform(-100, 580))
    mp1 = rng.random(n1) < 0.2; mp2 = rng.random(n2) < 0.2
    only = bool(rng.integers(0, 2))
    _, nm, m12 = m.frame_search_for_triangulation(f1, fv1, mp1, f2, fv2, mp2, F12, ex, ey, sf, sg, only)
    r12, rn = oracle.search_for_triangulation(k1, d1, fv1, mp1, s1, k2, d2, fv2, mp2, s2, F12, ex, ey, sf, sg, only, ori)
    tot_tri += int(rn)
    if not (nm == rn and np.array_equal(m12, r12)):
        bad += 1; print("MISMATCH frame triangulation", case, n1, n2, nn, only, ori, flush=True)
    f1.close(); f2.close()
lib().orbm_debug_force_sequential_resolver(0)
print("cases", n, "mismatches", bad, "matches seen: bow", tot_bow, "triangulation", tot_tri)
sys.exit(1 if bad else 0)
