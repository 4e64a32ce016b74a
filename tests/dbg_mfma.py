import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from orb_slam2_e_amd.matcher import ORBmatcher
rng = np.random.default_rng(0)
for nA, nB in [(32, 32), (64, 64), (128, 2016), (100, 2000), (2000, 2000)]:
    A = rng.integers(0, 256, (nA, 32), dtype=np.uint8); B = rng.integers(0, 256, (nB, 32), dtype=np.uint8)
    got = ORBmatcher().match_bruteforce(A, B); ref = oracle.match_bruteforce(A, B)
    bad = [int((g != r).sum()) for g, r in zip(got, ref)]
    print(nA, nB, "mismatches best/second/idx", bad)
    if sum(bad):
        w = np.nonzero((got[0] != ref[0]) | (got[2] != ref[2]) | (got[1] != ref[1]))[0][:8]
        for i in w:
            print("  q", i, "got", got[0][i], got[1][i], got[2][i], "ref", ref[0][i], ref[1][i], ref[2][i])
