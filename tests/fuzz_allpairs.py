"""Differential fuzz of the all-pairs matcher (both kernels: FP4 matrix cores and popcount) and its neighbours vs the oracle:
orbm_match_bruteforce on random set sizes with planted duplicates / complements / all-equal rows, orbm_match_batch_dev on resident
sets with ragged counts and arbitrary (query, train) pairs, orbm_match_candidates, orbm_distinctive_descriptors.
usage: fuzz_allpairs.py [ncases] [seed]"""
import sys
import numpy as np
import torch
import oracle
from orb_slam2_e_amd import ORBmatcher

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0


def rand_sets(nA, nB):
    style = int(rng.integers(0, 4))
    if style == 0:            # unrelated rows
        A = rng.integers(0, 256, (nA, 32), dtype=np.uint8); B = rng.integers(0, 256, (nB, 32), dtype=np.uint8)
    elif style == 1:          # B = noisy A (cyclic), planted exact duplicates and complements
        A = rng.integers(0, 256, (nA, 32), dtype=np.uint8)
        B = A[np.arange(nB) % nA] ^ np.packbits(rng.random((nB, 256)) < rng.choice([0.0, 0.03, 0.2]), axis=1, bitorder="little")
        B[rng.integers(0, nB, max(1, nB // 10))] = A[rng.integers(0, nA)]
        B[rng.integers(0, nB)] = ~A[rng.integers(0, nA)]
    elif style == 2:          # few distinct rows: ties everywhere (first index wins)
        P = rng.integers(0, 256, (int(rng.integers(1, 5)), 32), dtype=np.uint8)
        A = P[rng.integers(0, len(P), nA)]; B = P[rng.integers(0, len(P), nB)]
    else:                     # low-weight rows: small distances, many equal
        A = np.packbits(rng.random((nA, 256)) < 0.02, axis=1, bitorder="little"); B = np.packbits(rng.random((nB, 256)) < 0.02, axis=1, bitorder="little")
    return np.ascontiguousarray(A), np.ascontiguousarray(B)


for case in range(n):
    kern = [ORBmatcher.ALLPAIRS_POPCOUNT, ORBmatcher.ALLPAIRS_MFMA, ORBmatcher.ALLPAIRS_AUTO][int(rng.choice(3, p=[0.3, 0.3, 0.4]))]
    prev = ORBmatcher.set_allpairs_kernel(kern)
    try:
        big = rng.random() < 0.1
        nA = int(rng.integers(1, 6000 if big else 700)); nB = int(rng.integers(1, 6000 if big else 700))
        A, B = rand_sets(nA, nB)
        ratio = float(rng.choice([0.6, 0.8, 1.0])); th = int(rng.choice([30, 50, 100, 256]))
        m = ORBmatcher(ratio)
        got = m.match_bruteforce(A, B); ref = oracle.match_bruteforce(A, B)
        if not all(np.array_equal(g, r) for g, r in zip(got, ref)):
            bad += 1; print("MISMATCH bruteforce", case, nA, nB, "kernel", kern, flush=True)
        # resident sets
        cap = int(rng.integers(1, 400)); nsets = int(rng.integers(1, 7))
        counts = np.array([int(rng.choice([0, 1, cap, int(rng.integers(0, cap + 1))])) for _ in range(nsets)], np.int32)
        desc = rng.integers(0, 256, (nsets, cap, 32), dtype=np.uint8)
        for s_ in range(1, nsets):
            k = int(rng.integers(0, cap + 1))
            desc[s_, :k] = desc[0, :k] ^ np.packbits(rng.random((k, 256)) < rng.choice([0.0, 0.05]), axis=1, bitorder="little")
        npair = int(rng.integers(1, 10))
        pa = rng.integers(0, nsets, npair).astype(np.int32); pb = rng.integers(0, nsets, npair).astype(np.int32)
        d = torch.from_numpy(desc).cuda(); c = torch.from_numpy(counts).cuda()
        ta, tb = torch.from_numpy(pa).cuda(), torch.from_numpy(pb).cuda()
        out = [torch.full((npair, cap), -7, dtype=torch.int32).cuda() for _ in range(4)]
        nm = torch.full((npair,), -7, dtype=torch.int32).cuda()
        m.match_batch_device(d.data_ptr(), c.data_ptr(), cap, ta.data_ptr(), tb.data_ptr(), npair, out[0].data_ptr(), out[1].data_ptr(),
                             out[2].data_ptr(), out[3].data_ptr(), nm.data_ptr(), th=th)
        torch.cuda.synchronize()
        best, second, idx, m12 = (o.cpu().numpy() for o in out)
        nmh = nm.cpu().numpy()
        for p in range(npair):
            qa, qb = int(counts[pa[p]]), int(counts[pb[p]])
            if qa == 0:
                ok = (best[p] == -7).all() and int(nmh[p]) in (0, -7)
            elif qb == 0:
                continue      # an empty train set: the reference never matches against one; outputs are documented as "no match"
            else:
                rb, rs, ri = oracle.match_bruteforce(desc[pa[p], :qa], desc[pb[p], :qb])
                rm, rn = oracle.match_filter(rb, rs, ri, th, ratio)
                ok = (np.array_equal(best[p, :qa], rb) and np.array_equal(second[p, :qa], rs) and np.array_equal(idx[p, :qa], ri) and
                      np.array_equal(m12[p, :qa], rm) and int(nmh[p]) == rn and (best[p, qa:] == -7).all())
            if not ok:
                bad += 1; print("MISMATCH batch", case, "pair", p, qa, qb, "cap", cap, "kernel", kern, flush=True); break
    finally:
        ORBmatcher.set_allpairs_kernel(prev)
    # candidate lists
    nA = int(rng.integers(1, 800)); nB = int(rng.integers(1, 800))
    A, B = rand_sets(nA, nB)
    off = [0]; idx = []
    for i in range(nA):
        k = int(rng.integers(0, 50)) if rng.random() < 0.9 else int(rng.integers(0, 600))
        idx.extend(rng.integers(0, nB, size=k).tolist()); off.append(len(idx))
    got = ORBmatcher().match_candidates(A, B, off, idx); ref = oracle.match_candidates(A, B, off, idx)
    if not all(np.array_equal(g, r) for g, r in zip(got, ref)):
        bad += 1; print("MISMATCH candidates", case, nA, nB, flush=True)
    # ComputeDistinctiveDescriptors over many map points
    sizes = rng.integers(0, int(rng.choice([5, 60, 300])), int(rng.integers(1, 300)))
    offd = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    if offd[-1] > 0:
        base = rng.integers(0, 256, (len(sizes), 32), dtype=np.uint8)
        dd = np.concatenate([base[i] ^ np.packbits(rng.random((k, 256)) < rng.choice([0.0, 0.1]), axis=1, bitorder="little") for i, k in enumerate(sizes) if k > 0])
        got = ORBmatcher.distinctive_descriptors(dd, offd); ref = oracle.distinctive_descriptors(dd, offd)
        if not np.array_equal(got, ref):
            bad += 1; print("MISMATCH distinctive", case, len(sizes), flush=True)
print("cases", n, "mismatches", bad)
sys.exit(1 if bad else 0)
