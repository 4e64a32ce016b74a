"""GPU parity: Hamming match kernels (through the C-ABI) vs the CPU oracle. Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd import ORBmatcher
from orb_slam2_e_amd.synth import synth_descriptors


def test_hamming_matrix():
    A, B, _ = synth_descriptors(300, seed=3)
    assert np.array_equal(ORBmatcher.hamming_matrix(A, B[:211]), oracle.hamming_matrix(A, B[:211]))


def test_known_answers():
    z = np.zeros(32, np.uint8); f = np.full(32, 255, np.uint8)
    assert ORBmatcher.DescriptorDistance(z, z) == 0
    assert ORBmatcher.DescriptorDistance(z, f) == 256
    one = z.copy(); one[17] = 0x10
    assert ORBmatcher.DescriptorDistance(z, one) == 1


@pytest.mark.parametrize("nA,nB", [(2000, 2000), (1, 1), (257, 3), (5, 1000), (2024, 2024), (100, 0)])
def test_bruteforce_config2(nA, nB):
    A, B, _ = synth_descriptors(max(nA, nB, 1), seed=7)
    A, B = A[:nA], B[:nB]
    m = ORBmatcher(0.6)
    got = m.match_bruteforce(A, B)
    ref = oracle.match_bruteforce(A, B)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    gm, gn = m.filter(*got)
    rm, rn = oracle.match_filter(*ref, ORBmatcher.TH_LOW, 0.6)
    assert gn == rn and np.array_equal(gm, rm)


def test_ties_first_index_wins():
    A = np.zeros((4, 32), np.uint8)
    B = np.zeros((600, 32), np.uint8)
    B[:, 0] = 1          # all at distance 1
    B[300:, 1] = 1       # second half at distance 2
    got = ORBmatcher().match_bruteforce(A, B)
    assert list(got[0]) == [1] * 4 and list(got[1]) == [1] * 4 and list(got[2]) == [0] * 4


def test_candidate_lists():
    rng = np.random.default_rng(11)
    A, B, _ = synth_descriptors(500, seed=9)
    off = [0]; idx = []
    for i in range(len(A)):
        k = int(rng.integers(0, 40))
        idx.extend(rng.integers(0, len(B), size=k).tolist())
        off.append(len(idx))
    got = ORBmatcher().match_candidates(A, B, off, idx)
    ref = oracle.match_candidates(A, B, off, idx)
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
