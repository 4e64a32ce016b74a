"""The projection-family resolver A/B on one box: k_resolve_par (parallel fixed point, rotation check fused) against k_resolve + k_rotation.
usage (GPU box): python3 tools/resolver_ab.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd._lib import lib
from orb_slam2_e_amd.synth import synth_projection_case

L = lib()
m = ORBmatcher(0.6, True)
for name, kw in (("2000 x 2000, one query per keypoint", dict(n=2000, nq=2000, hot=2000)), ("3000 queries on 400 hot keypoints", dict(n=2000, nq=3000, hot=400)),
                 ("1000 x 1000", dict(n=1000, nq=1000, hot=1000))):
    q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(0, **kw)
    f = lambda: m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95)
    res = {}
    for rep in range(3):
        for label, on in (("parallel", 0), ("sequential", 1)):
            L.orbm_debug_force_sequential_resolver(on)
            for _ in range(20): f()
            t0 = time.perf_counter()
            for _ in range(100): f()
            res.setdefault(label, []).append((time.perf_counter() - t0) / 100 * 1e3)
    L.orbm_debug_force_sequential_resolver(0)
    f()
    print("  iterations of the fixed point:", L.orbm_debug_last_resolver_iterations())
    print(name, {k: [round(x, 4) for x in v] for k, v in res.items()}, flush=True)
