"""Kernel trace target: 300 whole-loop projection searches (2000 x 2000).  usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/search_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.synth import synth_projection_case
q, qd, qa, takes, kps, desc, bounds, occ, ur = synth_projection_case(0, n=2000, nq=2000, hot=2000)
m = ORBmatcher(0.6, True)
for _ in range(20): m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95)
t0 = time.perf_counter()
for _ in range(300): m.search_projection(q, qd, qa, takes, kps, desc, bounds, occ, ur, 95)
print("ms per call", (time.perf_counter() - t0) / 300 * 1e3)
