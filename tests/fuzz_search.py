"""Differential fuzz: the whole-loop projection search (parallel fixed-point resolver, with its sequential fallback) vs the
oracle's literal loop, over random densities, window sizes, crowding, `takes` fractions, acceptance modes and thresholds.
usage: fuzz_search.py [ncases] [seed]"""
import sys
import numpy as np
import oracle
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd._lib import lib
from orb_slam2_e_amd.extractor import KP_DTYPE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = lib()
bad = 0
fallbacks = 0
iters = []
for case in range(n):
    nk = int(rng.integers(1, 3000)); nq = int(rng.integers(1, 4000))
    W, H = float(rng.choice([320, 640, 1241])), float(rng.choice([240, 480, 376]))
    kps = np.zeros(nk, KP_DTYPE)
    crowd = rng.random() < 0.3
    if crowd:
        kps["x"] = rng.uniform(W / 2 - 30, W / 2 + 30, nk); kps["y"] = rng.uniform(H / 2 - 30, H / 2 + 30, nk)
    else:
        kps["x"] = rng.uniform(-5, W + 5, nk); kps["y"] = rng.uniform(-5, H + 5, nk)
    kps["octave"] = rng.integers(0, 8, nk); kps["angle"] = rng.uniform(0, 360, nk)
    nproto = int(rng.integers(1, max(2, nk // 3)))
    proto = rng.integers(0, 256, (nproto, 32), dtype=np.uint8)
    desc = proto[rng.integers(0, nproto, nk)] ^ np.packbits(rng.random((nk, 256)) < rng.choice([0.0, 0.02, 0.1]), axis=1, bitorder="little")
    src = rng.integers(0, max(1, int(nk * rng.choice([0.02, 0.3, 1.0]))), nq) % nk
    q = np.zeros(nq, ORBmatcher.WQ_DTYPE)
    q["u"] = kps["x"][src] + rng.normal(0, 2, nq); q["v"] = kps["y"][src] + rng.normal(0, 2, nq)
    q["r"] = rng.choice([3.0, 7.0, 15.0, 40.0, 120.0]) * rng.uniform(0.5, 1.5, nq)
    lv = kps["octave"][src]
    q["min_level"] = np.where(rng.random(nq) < 0.3, -1, lv - 1); q["max_level"] = np.where(rng.random(nq) < 0.3, -1, lv + rng.integers(0, 2, nq))
    q["xr"] = q["u"] - rng.uniform(0, 30, nq)
    qd = desc[src] ^ np.packbits(rng.random((nq, 256)) < rng.choice([0.0, 0.03, 0.15]), axis=1, bitorder="little")
    qa = ((kps["angle"][src] + rng.normal(0, 20, nq)) % 360).astype(np.float32)
    takes = (rng.random(nq) < rng.choice([0.0, 0.5, 0.9, 1.0])).astype(np.uint8)
    occ = (rng.random(nk) < rng.choice([0.0, 0.1])).astype(np.uint8)
    ur = np.where(rng.random(nk) < 0.5, kps["x"] - rng.uniform(0, 30, nk), -1).astype(np.float32) if rng.random() < 0.4 else None
    th = int(rng.choice([30, 45, 60, 95, 100, 255])); lvl = bool(rng.random() < 0.4); ori = bool(rng.random() < 0.7)
    ratio = float(rng.choice([0.6, 0.8, 0.9]))
    m = ORBmatcher(ratio, ori)
    bounds = (0.0, 0.0, W, H)
    got = m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, th, lvl)
    it = L.orbm_debug_last_resolver_iterations()
    fallbacks += it < 0
    if it > 0: iters.append(it)
    ref = oracle.search_projection_seq(q, qd, qa, takes, kps, desc, bounds, occ, ur, th, ratio, lvl, ori)
    if not (got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
        bad += 1
        print("MISMATCH case", case, "nk", nk, "nq", nq, "crowd", crowd, "th", th, "lvl", lvl, "ori", ori, "iterations", it, flush=True)
    if case % 20 == 19:
        print("...", case + 1, "cases,", bad, "mismatches", flush=True)
# ORBmatcher::SearchForInitialization (k_resolve_init_par / k_resolve<1>): look-alike queries competing for the same keypoints
# (steals, the matched-distance gate), random window sizes incl. ones whose lists outgrow their regions
ifall, iiters = 0, []
for case in range(max(1, n // 4)):
    n1 = int(rng.integers(1, 3000)); n2 = int(rng.integers(1, 3000))
    k1 = np.zeros(n1, KP_DTYPE)
    k1["x"] = rng.uniform(0, 640, n1); k1["y"] = rng.uniform(0, 480, n1)
    k1["octave"] = rng.choice(8, n1, p=[0.5, 0.15, 0.1, 0.08, 0.07, 0.05, 0.03, 0.02]); k1["angle"] = rng.uniform(0, 360, n1)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    nalike = int(rng.choice([0, n1 // 10, n1 // 3, n1])); protos = int(rng.choice([1, 3, 30]))
    if nalike:
        who = rng.choice(n1, nalike, replace=False)
        d1[who] = d1[:protos][rng.integers(0, min(protos, n1), nalike)] ^ np.packbits(rng.random((nalike, 256)) < rng.choice([0.0, 0.01]), axis=1, bitorder="little")
        if rng.random() < 0.5:     # ... and close together, so that they see the same windows
            k1["x"][who] = rng.uniform(300, 340, nalike); k1["y"][who] = rng.uniform(200, 240, nalike); k1["octave"][who] = 0
    src = rng.integers(0, n1, n2)
    k2 = k1[src].copy()
    k2["x"] += rng.normal(0, 6, n2); k2["y"] += rng.normal(0, 6, n2)
    k2["angle"] = (k2["angle"] + rng.choice([20.0, 140.0, 250.0], n2, p=[0.7, 0.2, 0.1]) + rng.normal(0, 3, n2)) % 360
    d2 = d1[src] ^ np.packbits(rng.random((n2, 256)) < rng.choice([0.0, 0.03, 0.08]), axis=1, bitorder="little")
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    window = int(rng.choice([10, 30, 100, 300, 700])); ratio = float(rng.choice([0.6, 0.9, 1.0])); ori = bool(rng.random() < 0.7)
    bounds = (0.0, 0.0, 640.0, 480.0)
    got = ORBmatcher(ratio, ori).SearchForInitialization(k1, d1, k2, d2, prev.copy(), bounds, window)
    it = L.orbm_debug_last_resolver_iterations()
    ifall += it < 0
    if it > 0: iiters.append(it)
    ref = oracle.search_for_initialization(k1, d1, k2, d2, prev.copy(), bounds, window, ratio, ori)
    if not (got[2] == ref[2] and np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])):
        bad += 1
        print("MISMATCH initialization case", case, "n1", n1, "n2", n2, "alike", nalike, protos, "window", window, "ratio", ratio, "ori", ori, "iterations", it, flush=True)
print("initialization cases", max(1, n // 4), "| fallbacks", ifall, "| fixed-point iterations: median", int(np.median(iiters)) if iiters else 0, "max", max(iiters) if iiters else 0)
print("cases", n, "mismatches", bad, "| fallbacks to the sequential resolver", fallbacks, "| fixed-point iterations: median", int(np.median(iters)) if iters else 0,
      "max", max(iters) if iters else 0)
sys.exit(1 if bad else 0)
