"""Kernel-trace target: ORBmatcher::SearchForInitialization at 2000 x 2200 (the bench's case), 60 calls.
usage (GPU box): rocprofv3 --kernel-trace --stats -- python3 tools/init_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.synth import synth_initialization_case
k1, d1, k2, d2, prev, bounds = synth_initialization_case(0)
m = ORBmatcher(0.9, True)
for _ in range(10): m.SearchForInitialization(k1, d1, k2, d2, prev, bounds, 100)
t0 = time.perf_counter()
for _ in range(50): r = m.SearchForInitialization(k1, d1, k2, d2, prev, bounds, 100)
print("%.3f ms per call, %d matches" % ((time.perf_counter() - t0) / 50 * 1e3, r[2]))
