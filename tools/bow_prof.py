"""Kernel-trace target: ORBmatcher::SearchByBoW at 2000 x 2100 (the bench's case), 60 calls.
usage (GPU box): rocprofv3 --kernel-trace --stats -- python3 tools/bow_prof.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.synth import synth_bow_case
from orb_slam2_e_amd.vocabulary import feature_vector_arrays
d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2 = synth_bow_case(0)
fv1 = feature_vector_arrays(node1, keep1); fv2 = feature_vector_arrays(node2, keep2)
m = ORBmatcher(0.6, True)
for _ in range(10): m.SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, False)
t0 = time.perf_counter()
for _ in range(50): r = m.SearchByBoW(fv1, valid1, d1, a1, fv2, valid2, d2, a2, False)
print("%.3f ms per call, %d matches" % ((time.perf_counter() - t0) / 50 * 1e3, r[2]))
