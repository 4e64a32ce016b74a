"""GPU parity: DBoW2 vocabulary-tree descent (Frame::ComputeBoW's Hamming half)
vs the CPU oracle on a synthetic k=10, L=4 vocabulary (the real ORBvoc is not in
the mount: SURVEY 8f rank 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.vocabulary import ORBVocabulary, assemble_bow


def _synthetic_vocabulary(k=10, L=4, seed=0):
    rng = np.random.default_rng(seed)
    child_off, child_ids, desc, word, weight = [0], [], [rng.integers(0, 256, 32, dtype=np.uint8)], [-1], [0.0]
    level = [0]
    nwords = 0
    for depth in range(1, L + 1):
        nxt = []
        for parent in level:
            pass
        # breadth-first: children of every node of the previous level get consecutive ids
        new_level = []
        for parent in level:
            first = len(desc)
            for c in range(k):
                d = desc[parent] ^ np.packbits(rng.random(256) < 0.5 ** depth, bitorder="little")
                desc.append(d)
                leaf = depth == L
                word.append(nwords if leaf else -1)
                weight.append(float(rng.uniform(0.5, 8.0)) if leaf else 0.0)
                nwords += leaf
                new_level.append(first + c)
        level = new_level
    n = len(desc)
    # child lists in node order
    kids = {i: [] for i in range(n)}
    idx = 1
    q = [0]
    for depth in range(1, L + 1):
        nq = []
        for parent in q:
            kids[parent] = list(range(idx, idx + k)); nq += kids[parent]; idx += k
        q = nq
    off = [0]
    ids = []
    for i in range(n):
        ids += kids[i]; off.append(len(ids))
    desc = np.stack(desc)
    desc[kids[0][3]] = desc[kids[0][1]]            # equal children: the first one must win ties
    w = np.array(weight); w[rng.choice(np.where(np.array(word) >= 0)[0], 50, replace=False)] = 0.0   # stop words
    return np.array(off, np.int32), np.array(ids, np.int32), desc, np.array(word, np.int32), w, L


@pytest.mark.parametrize("levelsup", [4, 2, 0, 6])
def test_bow_descent_and_vectors(levelsup):
    off, ids, desc, word, weight, L = _synthetic_vocabulary()
    rng = np.random.default_rng(1)
    leaves = np.where(word >= 0)[0]
    feats = desc[rng.choice(leaves, 2000)] ^ np.packbits(rng.random((2000, 256)) < 0.04, axis=1, bitorder="little")
    voc = ORBVocabulary(off, ids, desc, word, weight, L)
    got = voc.descend(feats, levelsup)
    ref = oracle.bow_descend(off, ids, desc, word, weight, L, feats, levelsup)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    gb, gf = voc.transform(feats, levelsup)
    rb, rf = assemble_bow(*ref)
    assert gb == rb and gf == rf and abs(sum(gb.values()) - 1.0) < 1e-12 and len(gb) > 500
