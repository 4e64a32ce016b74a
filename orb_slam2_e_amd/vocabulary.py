"""Host-side mirror of the DBoW2 vocabulary transform (Frame::ComputeBoW,
src/Frame.cc:410-417): the tree descent runs on the device through the C-ABI, the
BowVector / FeatureVector maps are assembled here exactly as
TemplatedVocabulary::transform does (TemplatedVocabulary.h:1127-1160: addWeight in
feature order, then L1 normalisation for the TF_IDF / L1_NORM setting ORB-SLAM uses)."""
import ctypes as C

import numpy as np

from ._lib import check, lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBVocabulary:
    def __init__(self, child_off, child_ids, node_desc, node_word, node_weight, L):
        self._L = lib()
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
        self._L.orbm_bow_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        check(self._L.orbm_bow_transform(self._h, _p(f), n, levelsup, _p(word), _p(node), _p(w)))
        return word, node, w

    def transform(self, features, levelsup=4):
        """Returns (BowVector: {word id: value}, FeatureVector: {node id: [feature indices]})."""
        word, node, w = self.descend(features, levelsup)
        return assemble_bow(word, node, w)


def assemble_bow(word, node, w):
    bow, fv = {}, {}
    for i in range(len(word)):
        if w[i] > 0:                                   # not stopped
            bow[int(word[i])] = bow.get(int(word[i]), 0.0) + float(w[i])      # BowVector::addWeight
            fv.setdefault(int(node[i]), []).append(i)                         # FeatureVector::addFeature
    norm = 0.0
    for k in sorted(bow):                              # BowVector::normalize(L1), std::map order
        norm += abs(bow[k])
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
