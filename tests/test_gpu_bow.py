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


def test_bow_descent_orbvoc_shape(tmp_path):
    """The reference's real vocabulary shape -- k = 10, L = 6: 1,111,111 nodes, 35.6 MB of node descriptors (Vocabulary/ORBvoc.txt
    is missing from the reference mount) -- written in DBoW2's text layout, read back by the loader (TemplatedVocabulary.h:1338-1420),
    and 64 x 2000 descriptors descended with levelsup = 4 (Frame.cc:415): host-array call and the resident batch form, word and
    node ids bit-equal to the oracle; ties between equal siblings go to the first (:1246-1254)."""
    import torch
    from orb_slam2_e_amd.synth import synth_vocabulary, synth_vocabulary_features
    voc = synth_vocabulary(10, 6, seed=0)
    path = str(tmp_path / "ORBvoc_synth.txt")
    save_vocabulary_text(path, *voc[:5], 10, 6)
    v = ORBVocabulary.loadFromTextFile(path)
    assert v.m_k == 10 and v.m_L == 6 and len(v.node_word) == 1111111 and int((v.node_word >= 0).sum()) == 10 ** 6
    for a, b in zip((v.child_off, v.child_ids, v.node_word, v.node_weight), (voc[0], voc[1], voc[3], voc[4])):
        assert np.array_equal(a, b)
    assert np.array_equal(v.node_desc[1:], voc[2][1:])            # (the root has no line and no descriptor)
    B, N = 64, 2000
    feats = synth_vocabulary_features(voc, B * N, seed=1).reshape(B, N, 32)
    # a few features that sit exactly between two equal siblings: copies of tied nodes
    tied = np.nonzero((voc[2][2:-2] == voc[2][4:]).all(1))[0] + 2
    assert len(tied) >= 32
    feats[0, :32] = voc[2][tied[:32]]
    ref = oracle.bow_descend(*voc, feats.reshape(-1, 32), 4)
    assert len(np.unique(ref[1])) > 90 and len(np.unique(ref[0])) > 50000 and (ref[2] == 0).sum() > 0
    got = v.descend(feats[3], 4)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r[3 * N:4 * N])
    d_f = torch.from_numpy(feats).cuda()
    cnt = np.full(B, N, np.int32); cnt[5] = 1234; cnt[9] = 0
    d_c = torch.from_numpy(cnt).cuda()
    d_w = torch.full((B, N), -7, dtype=torch.int32, device="cuda"); d_n = torch.full_like(d_w, -7)
    s = torch.cuda.Stream()
    v.descend_batch_device(d_f.data_ptr(), d_c.data_ptr(), N, B, 4, d_w.data_ptr(), d_n.data_ptr(), s.cuda_stream)
    s.synchronize()
    w, nid = d_w.cpu().numpy(), d_n.cpu().numpy()
    rw, rn = ref[0].reshape(B, N), ref[1].reshape(B, N)
    for b in range(B):
        assert np.array_equal(w[b, :cnt[b]], rw[b, :cnt[b]]) and np.array_equal(nid[b, :cnt[b]], rn[b, :cnt[b]])
        assert (w[b, cnt[b]:] == -7).all() and (nid[b, cnt[b]:] == -7).all()      # rows past a set's count are not written
    for lu in (0, 2, 6, 7):                                                        # other levels of the FeatureVector key
        g = v.descend(feats[1, :500], lu)
        r = oracle.bow_descend(*voc, feats[1, :500], lu)
        assert all(np.array_equal(x, y) for x, y in zip(g, r))
