"""Kernel-trace target: ORBmatcher::SearchByBoW on two RESIDENT frames (orbm_frame_search_by_bow) at 2000 x 2100 (the bench's case), 200 calls.
usage (GPU box): rocprofv3 --kernel-trace --stats -- python3 tools/bow_resident_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_e_amd import ORBmatcher, Frame, KP_DTYPE
from orb_slam2_e_amd.synth import synth_bow_case
from orb_slam2_e_amd.vocabulary import feature_vector_arrays
d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(0)
fv1 = feature_vector_arrays(node1, keep1); fv2 = feature_vector_arrays(node2, keep2)
rng = np.random.default_rng(5)
def fr(d, a):
    k = np.zeros(len(d), KP_DTYPE); k["x"] = rng.uniform(0, 640, len(d)); k["y"] = rng.uniform(0, 480, len(d)); k["angle"] = a
    return Frame(k, d, (0.0, 0.0, 640.0, 480.0))
f1, f2 = fr(d1, a1), fr(d2, a2)
m = ORBmatcher(0.6, True)
for _ in range(20): m.frame_search_by_bow(f1, fv1, valid1, f2, fv2, None, False)
t = []
for _ in range(20):
    t0 = time.perf_counter()
    for _ in range(10): r = m.frame_search_by_bow(f1, fv1, valid1, f2, fv2, None, False)
    t.append((time.perf_counter() - t0) / 10)
print("median %.4f ms best %.4f ms per call, matches %d" % (np.median(t) * 1e3, min(t) * 1e3, r[2]))
