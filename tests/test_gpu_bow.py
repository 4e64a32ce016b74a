"""GPU parity: DBoW2 vocabulary-tree descent (Frame::ComputeBoW's Hamming half)
vs the CPU oracle on a synthetic k=10, L=4 vocabulary (the real ORBvoc is not in
the mount: SURVEY 8f rank 2)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.vocabulary import ORBVocabulary, assemble_bow, load_vocabulary_text, save_vocabulary_text


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


def _irregular_vocabulary(seed=5, L=5):
    """A tree as k-means can leave it: between 1 and 7 children per node, leaves at every depth 1..L, node ids in
    creation (breadth-first) order, as parent-array so the text writer can emit it."""
    rng = np.random.default_rng(seed)
    parent, depth, desc = [0], [0], [np.zeros(32, np.uint8)]
    frontier = [0]
    for d in range(1, L + 1):
        nxt = []
        for p in frontier:
            for _ in range(int(rng.integers(1, 8))):
                parent.append(p); depth.append(d)
                desc.append(desc[p] ^ np.packbits(rng.random(256) < 0.5 ** d, bitorder="little"))
                nxt.append(len(parent) - 1)
        frontier = [i for i in nxt if d < L and rng.random() < 0.8]      # the others stay leaves at depth d
        if not frontier and d < L:
            frontier = nxt[:1]
    n = len(parent)
    parent = np.array(parent)
    inner = np.zeros(n, bool); inner[parent[1:]] = True
    word = np.full(n, -1, np.int32); word[~inner] = np.arange(int((~inner).sum()))
    weight = np.where(inner, 0.0, rng.uniform(0.1, 9.0, n)); weight[np.where(~inner)[0][::17]] = 0.0
    order = np.argsort(parent[1:], kind="stable") + 1
    off = np.zeros(n + 1, np.int32); np.cumsum(np.bincount(parent[1:], minlength=n), out=off[1:])
    return off, order.astype(np.int32), np.stack(desc), word, weight, L


@pytest.mark.parametrize("levelsup", [4, 1])
@pytest.mark.parametrize("scoring,weighting", [(0, 0), (1, 0), (5, 1), (0, 2), (5, 3)])
def test_vocabulary_loaded_from_text_file(tmp_path, levelsup, scoring, weighting):
    """ORBvoc.txt route: write an irregular tree in the saveToTextFile layout, load it (System.cc:69), and the
    descent + BowVector / FeatureVector must equal the oracle's on the arrays the file was written from."""
    off, ids, desc, word, weight, L = _irregular_vocabulary()
    path = str(tmp_path / "voc.txt")
    save_vocabulary_text(path, off, ids, desc, word, weight, 7, L, scoring, weighting)
    voc = ORBVocabulary.loadFromTextFile(path)
    assert voc.m_L == L and voc.m_k == 7 and (voc.m_scoring, voc.m_weighting) == (scoring, weighting)
    assert np.array_equal(voc.child_off, off) and np.array_equal(voc.child_ids, ids) and np.array_equal(voc.node_word, word)
    assert np.array_equal(voc.node_desc[1:], desc[1:]) and np.array_equal(voc.node_weight, weight)
    rng = np.random.default_rng(2)
    leaves = np.where(word >= 0)[0]
    feats = desc[rng.choice(leaves, 1500)] ^ np.packbits(rng.random((1500, 256)) < 0.05, axis=1, bitorder="little")
    got = voc.descend(feats, levelsup)
    ref = oracle.bow_descend(off, ids, desc, word, weight, L, feats, levelsup)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    assert voc.transform(feats, levelsup) == assemble_bow(*ref, scoring, weighting)


@pytest.mark.parametrize("levelsup", [4, 0])
def test_bow_batch_on_resident_descriptor_sets(levelsup):
    """orbm_bow_transform_batch_dev: the extractor's result layout ([sets][cap][32], counts) in HBM, ids left there."""
    import torch
    off, ids, desc, word, weight, L = _synthetic_vocabulary()
    voc = ORBVocabulary(off, ids, desc, word, weight, L)
    rng = np.random.default_rng(3)
    cap, counts = 500, [500, 0, 1, 17, 333]
    leaves = np.where(word >= 0)[0]
    sets = desc[rng.choice(leaves, (len(counts), cap))] ^ np.packbits(rng.random((len(counts), cap, 256)) < 0.04, axis=2, bitorder="little")
    d = torch.from_numpy(np.ascontiguousarray(sets)).cuda(); c = torch.tensor(counts, dtype=torch.int32).cuda()
    w = torch.full((len(counts), cap), -9, dtype=torch.int32).cuda(); n = torch.full_like(w, -9)
    voc.descend_batch_device(d.data_ptr(), c.data_ptr(), cap, len(counts), levelsup, w.data_ptr(), n.data_ptr())
    torch.cuda.synchronize()
    w, n = w.cpu().numpy(), n.cpu().numpy()
    for s_, cnt in enumerate(counts):
        rw, rn, _ = oracle.bow_descend(off, ids, desc, word, weight, L, sets[s_, :cnt], levelsup)
        assert np.array_equal(w[s_, :cnt], rw) and np.array_equal(n[s_, :cnt], rn), f"set {s_}"
        assert (w[s_, cnt:] == -9).all() and (n[s_, cnt:] == -9).all()
