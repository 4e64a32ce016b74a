"""Seed sweep over the whole-loop parity tests of tests/test_gpu_match.py and tests/test_fuse_and_projection.py: the tests' own
generators and comparisons, on seeds and parameter combinations the suite does not run (every case = one call of a test function;
its sanity asserts about case richness count as failures too and are reported with the case).  usage: fuzz_loops.py [ncases] [seed]"""
import os
import sys
import traceback

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import test_fuse_and_projection as tf
import test_gpu_bow as tbow
import test_gpu_match as tm
from orb_slam2_e_amd._lib import lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
L = lib()
B = lambda: bool(rng.random() < 0.5)
S = lambda: int(rng.integers(10, 1 << 30))


def with_resolver(fn, *a):
    seq = B()
    prev = L.orbm_debug_force_sequential_resolver(1 if seq else 0)
    try:
        fn(*a, "sequential" if seq else "parallel fixed point")
    finally:
        L.orbm_debug_force_sequential_resolver(prev)


def bow_case(irregular):
    """DBoW2 descent on random vocabulary trees (regular k^L and k-means-like irregular ones): word, weight and the node at
    levelsup of every feature against the oracle's literal descent, and the assembled BowVector / FeatureVector."""
    import oracle
    from orb_slam2_e_amd.vocabulary import ORBVocabulary, assemble_bow
    if irregular:
        off, ids, desc, word, weight, Lv = tbow._irregular_vocabulary(S(), int(rng.integers(1, 7)))
    else:
        k, Lv = [(4, 3), (5, 4), (7, 3), (10, 2), (10, 3), (12, 2), (4, 5)][int(rng.integers(0, 7))]   # (the generator wants k >= 4 and >= 50 words)
        off, ids, desc, word, weight, Lv = tbow._synthetic_vocabulary(k, Lv, S())
    leaves = np.where(word >= 0)[0]
    nf = int(rng.integers(1, 3000))
    feats = desc[rng.choice(leaves, nf)] ^ np.packbits(rng.random((nf, 256)) < rng.choice([0.0, 0.04, 0.3]), axis=1, bitorder="little")
    levelsup = int(rng.integers(0, Lv + 3))
    voc = ORBVocabulary(off, ids, desc, word, weight, Lv)
    got = voc.descend(feats, levelsup)
    ref = oracle.bow_descend(off, ids, desc, word, weight, Lv, feats, levelsup)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    gb, gf = voc.transform(feats, levelsup)
    rb, rf = assemble_bow(*ref)
    assert gb == rb and gf == rf


CASES = [
    ("bow descent, regular tree", lambda: bow_case(False)),
    ("bow descent, irregular tree", lambda: bow_case(True)),
    ("triangulation inner loop", lambda: tm.test_search_for_triangulation_inner_loop(S(), B())),
    ("triangulation whole", lambda: tm.test_search_for_triangulation_whole_function(S(), B(), B())),
    ("window == grid + selection", lambda: tm.test_search_window_equals_grid_then_selection(S(), B(), int(rng.choice([256, 2**31 - 1])))),
    ("whole-map projection", lambda: tm.test_whole_map_search_by_projection(int(rng.integers(0, 12)), float(rng.choice([1.0, 2.0, 3.0, 5.0])))),
    ("projection loop", lambda: with_resolver(tm.test_search_projection_whole_loop, S(), int(rng.choice([60, 95, 100])), B(), B(), B())),
    ("initialization", lambda: with_resolver(tm.test_search_for_initialization, S(), float(rng.choice([0.7, 0.9, 1.0])), B(),
                                             int(rng.choice([30, 100, 300])), int(rng.choice([1200, 2000, 2600])), int(rng.choice([3, 10, 40])))),
    ("bow loop", lambda: with_resolver(tm.test_search_by_bow_whole_loop, S(), B(), B(), float(rng.choice([0.6, 0.75, 0.9])))),
    ("fuse candidate loop", lambda: tm.test_fuse_candidate_loop(S(), B(), B())),
    ("fuse replay tail", lambda: tf.test_fuse_replay_equals_literal_loop_tail(S())),
    ("fuse replay tail (Sim3)", lambda: tf.test_fuse_replay_sim3_equals_literal_loop_tail(S())),
]
bad = thin = 0
THIN = ("assert (ref", "assert ref[", "assert rn >", "assert rn <", "assert taken", "assert not np.array_equal(free", "assert {k for", "assert any(")
count = {}
for case in range(n):
    name, fn = CASES[case % len(CASES)]
    state = rng.bit_generator.state
    try:
        fn()
    except AssertionError:
        tb = traceback.extract_tb(sys.exc_info()[2])[-1]
        # the tests also assert that a case is RICH enough (many matches, every branch taken, the coupling exercised): on a
        # random seed such a line failing says the case is thin, not that the library is wrong
        if tb.line.startswith(THIN):
            thin += 1
        else:
            bad += 1
            print("FAILED", name, "case", case, "at", f"{os.path.basename(tb.filename)}:{tb.lineno}", tb.line, flush=True)
    count[name] = count.get(name, 0) + 1
print("cases", n, count, "| thin cases (a richness assert of the test, not a comparison)", thin, "| failures", bad)
sys.exit(1 if bad else 0)
