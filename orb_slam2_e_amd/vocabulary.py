"""Host-side mirror of the DBoW2 vocabulary transform (Frame::ComputeBoW,
src/Frame.cc:410-417): the tree descent runs on the device through the C-ABI, the
BowVector / FeatureVector maps are assembled here exactly as
TemplatedVocabulary::transform does (TemplatedVocabulary.h:1127-1160: addWeight in
feature order, then L1 normalisation for the TF_IDF / L1_NORM setting ORB-SLAM uses)."""
import ctypes as C

import numpy as np

from ._lib import bind, check, lib, ptr as _p




class ORBVocabulary:
    def __init__(self, child_off, child_ids, node_desc, node_word, node_weight, L, scoring=0, weighting=0):
        self._L = lib()
        self.m_scoring, self.m_weighting = scoring, weighting
        self.child_off = np.ascontiguousarray(child_off, np.int32); self.child_ids = np.ascontiguousarray(child_ids, np.int32)
        self.node_desc = np.ascontiguousarray(node_desc, np.uint8); self.node_word = np.ascontiguousarray(node_word, np.int32)
        self.node_weight = np.ascontiguousarray(node_weight, np.float64)
        self.m_L = L
        self._h = C.c_void_p()
        check(self._L.orbm_vocab_create(_p(self.child_off), _p(self.child_ids), _p(self.node_desc), _p(self.node_word),
                                        _p(self.node_weight), len(self.node_word), L, C.byref(self._h)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.orbm_vocab_destroy(h)
            self._h = None

    def descend(self, features, levelsup):
        f = np.ascontiguousarray(features, np.uint8)
        n = len(f)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); w = np.zeros(n, np.float64)
        bind(self._L.orbm_bow_transform, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
        check(self._L.orbm_bow_transform(self._h, _p(f), n, levelsup, _p(word), _p(node), _p(w)))
        return word, node, w

    def descend_batch_device(self, desc_dev, counts_dev, cap, nsets, levelsup, word_id_dev, node_id_dev, stream=None):
        """orbm_bow_transform_batch_dev: descriptor sets resident in HBM ([nsets][cap][32] bytes, counts[nsets]) ->
        word / node ids [nsets][cap] int32 on the device (rows past a set's count are not written).  Asynchronous."""
        vp = lambda v: C.c_void_p(v) if v else None
        check(self._L.orbm_bow_transform_batch_dev(self._h, vp(desc_dev), vp(counts_dev), cap, nsets, levelsup, vp(word_id_dev),
                                                   vp(node_id_dev), vp(stream)))

    def transform(self, features, levelsup=4):
        """Returns (BowVector: {word id: value}, FeatureVector: {node id: [feature indices]})."""
        word, node, w = self.descend(features, levelsup)
        return assemble_bow(word, node, w, self.m_scoring, self.m_weighting)

    @classmethod
    def loadFromTextFile(cls, path):
        """ORBvoc.txt -> device tree (System.cc:69 mpVocabulary->loadFromTextFile)."""
        arrays, (k, scoring, weighting) = load_vocabulary_text(path)
        v = cls(*arrays, scoring=scoring, weighting=weighting)
        v.m_k = k
        return v


def load_vocabulary_text(path):
    """ORBvoc.txt reader = TemplatedVocabulary::loadFromTextFile (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1338-1420):
    first line `k L scoring weighting`, then one node per line `parent isLeaf d0 .. d31 weight`; node ids are line
    numbers (the root is node 0 and has no line), children keep their order of appearance, word ids number the leaves
    in file order.  Returns the flat arrays ORBVocabulary / orbm_vocab_create take:
    (child_off, child_ids, node_desc, node_word, node_weight, L) plus (k, scoring, weighting)."""
    with open(path) as f:
        head = f.readline().split()
        k, L, scoring, weighting = (int(x) for x in head[:4])
        if k < 0 or k > 20 or L < 1 or L > 10 or scoring < 0 or scoring > 5 or weighting < 0 or weighting > 3:
            raise ValueError("not a DBoW2 text vocabulary")           # the reference's check (:1359)
        rows = np.loadtxt(f, dtype=np.float64, ndmin=2)
    n = len(rows) + 1
    parent = rows[:, 0].astype(np.int64); leaf = rows[:, 1] > 0
    if rows.shape[1] != 35 or (parent < 0).any() or (parent >= np.arange(1, n)).any():
        raise ValueError("malformed vocabulary line")
    desc = np.zeros((n, 32), np.uint8); desc[1:] = rows[:, 2:34].astype(np.uint8)
    weight = np.zeros(n, np.float64); weight[1:] = rows[:, 34]
    word = np.full(n, -1, np.int32); word[1:][leaf] = np.arange(int(leaf.sum()), dtype=np.int32)
    order = np.argsort(parent, kind="stable")                        # children grouped by parent, in order of appearance
    child_ids = (order + 1).astype(np.int32)
    child_off = np.zeros(n + 1, np.int32)
    np.cumsum(np.bincount(parent, minlength=n), out=child_off[1:])
    return (child_off, child_ids, desc, word, weight, L), (k, scoring, weighting)


def save_vocabulary_text(path, child_off, child_ids, node_desc, node_word, node_weight, k, L, scoring=0, weighting=0):
    """Writer in the layout of TemplatedVocabulary::saveToTextFile (:1425-1450), for tests: nodes must be numbered so
    that every parent precedes its children (breadth-first ids as DBoW2 creates them)."""
    n = len(node_word)
    child_off = np.asarray(child_off); child_ids = np.asarray(child_ids)
    parent = np.zeros(n, np.int64)
    parent[child_ids] = np.repeat(np.arange(n), np.diff(child_off))
    rows = np.empty((n - 1, 35), np.float64)
    rows[:, 0] = parent[1:]; rows[:, 1] = np.asarray(node_word)[1:] >= 0
    rows[:, 2:34] = np.asarray(node_desc)[1:]; rows[:, 34] = np.asarray(node_weight)[1:]
    with open(path, "w") as f:
        f.write(f"{k} {L}  {scoring} {weighting}\n")
        np.savetxt(f, rows, fmt=["%d"] * 34 + ["%.17g"])      # (%.17g round-trips a double; 13 s for ORBvoc's 1.1 M lines)


def assemble_bow(word, node, w, scoring=0, weighting=0):
    """scoring / weighting: DBoW2 ScoringType / WeightingType numbers (BowVector.h:36-53); ORBvoc is L1_NORM, TF_IDF = 0, 0."""
    bow, fv = {}, {}
    tf = weighting in (0, 1)
    for i in range(len(word)):
        if w[i] > 0:                                   # not stopped
            if tf:
                bow[int(word[i])] = bow.get(int(word[i]), 0.0) + float(w[i])  # BowVector::addWeight
            else:
                bow.setdefault(int(word[i]), float(w[i]))                     # BowVector::addIfNotExist
            fv.setdefault(int(node[i]), []).append(i)                         # FeatureVector::addFeature
    must = scoring != 5                                # all but DOT_PRODUCT normalise (ScoringObject.h:74-89)
    if tf and bow and not must:
        nd = float(len(bow))
        for k in bow:
            bow[k] /= nd
    if must:
        norm = 0.0
        for k in sorted(bow):                          # BowVector::normalize, std::map order
            norm += bow[k] * bow[k] if scoring == 1 else abs(bow[k])
        if scoring == 1:
            norm = float(np.sqrt(norm))
        if norm > 0.0:
            for k in bow:
                bow[k] /= norm
    return bow, fv


def feature_vector_arrays(node_id, keep=None):
    """DBoW2::FeatureVector of one frame as arrays in std::map order: (nodes ascending, off, items in feature order).
    keep: optional mask of the features that enter it (weight > 0, Frame.cc:410-417 / assemble_bow)."""
    node_id = np.asarray(node_id)
    idx = np.arange(len(node_id)) if keep is None else np.nonzero(keep)[0]
    order = idx[np.argsort(node_id[idx], kind="stable")]
    nodes, start = np.unique(node_id[order], return_index=True)
    return nodes.astype(np.int32), np.append(start, len(order)).astype(np.int32), order.astype(np.int32)
