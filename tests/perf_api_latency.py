"""Wall time of the per-frame host-array entry points at the sizes one frame produces (B = 1)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.vocabulary import ORBVocabulary
from orb_slam2_e_amd.synth import synth_descriptors
from test_gpu_bow import _synthetic_vocabulary
from test_gpu_match import _triangulation_case


def t(f, reps=30):
    for _ in range(10): f()          # the card drops to a low power state while the host times the oracle
    t0 = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t0) / reps * 1e3


off, ids, desc, word, weight, L = _synthetic_vocabulary(k=10, L=5)
voc = ORBVocabulary(off, ids, desc, word, weight, L)
rng = np.random.default_rng(1)
leaves = np.where(word >= 0)[0]
feats = desc[rng.choice(leaves, 2000)] ^ np.packbits(rng.random((2000, 256)) < 0.04, axis=1, bitorder="little")
print("bow descent (orbm_bow_transform) 2000 features, k=10 L=5: %.3f ms" % t(lambda: voc.descend(feats, 4)))
args, os_ = _triangulation_case(0, False)
m = ORBmatcher(0.6, False)
print("match_triangulation: %.3f ms" % t(lambda: m.match_triangulation(*args, bOnlyStereo=os_)))
A, B = synth_descriptors(2000, 7)[:2]
offc = np.arange(0, 2001 * 30, 30, dtype=np.int32); cand = rng.integers(0, 2000, 2000 * 30).astype(np.int32)
print("match_candidates 2000 x 30: %.3f ms" % t(lambda: m.match_candidates(A, B, offc, cand)))
d = rng.integers(0, 256, (3000 * 12, 32), dtype=np.uint8); o = np.arange(0, 3001 * 12, 12, dtype=np.int32)
print("distinctive_descriptors 3000 points x 12 obs: %.3f ms" % t(lambda: m.distinctive_descriptors(d, o)))
