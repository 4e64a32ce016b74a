"""Round-based (scan-friendly) formulation of DistributeOctTree.

This is the algorithm the HIP kernel k_octree implements: every pass of the
reference's list surgery (ORBextractor.cc:594-739) becomes one "round" whose new
list positions are closed-form prefix sums.  Kept in Python so the formulation
itself can be checked against the literal list-based oracle on the CPU.
"""
import math
import numpy as np


def _f32(v):
    return np.float32(v)


def octree_rounds(x, y, resp, min_x, max_x, min_y, max_y, N):
    n = len(x)
    W, H = max_x - min_x, max_y - min_y
    q = _f32(W) / _f32(H)
    nIni = int(math.floor(float(q) + 0.5)) if q >= 0 else -int(math.floor(-float(q) + 0.5))  # roundf
    assert nIni >= 1
    hX = _f32(W) / _f32(nIni)
    # roots
    nx0 = [int(_f32(hX * _f32(i))) for i in range(nIni)]
    nx1 = [int(_f32(hX * _f32(i + 1))) for i in range(nIni)]
    ny0 = [0] * nIni
    ny1 = [H] * nIni
    node = np.array([int(_f32(x[k]) / hX) for k in range(n)], dtype=np.int64)
    cnt = [int((node == i).sum()) for i in range(nIni)]
    # drop empty roots (order kept)
    keep = [i for i in range(nIni) if cnt[i] > 0]
    remap = {s: i for i, s in enumerate(keep)}
    node = np.array([remap[s] for s in node], dtype=np.int64)
    nx0 = [nx0[i] for i in keep]; nx1 = [nx1[i] for i in keep]
    ny0 = [ny0[i] for i in keep]; ny1 = [ny1[i] for i in keep]
    cnt = [cnt[i] for i in keep]
    seq = [0] * len(cnt)
    S = len(cnt)
    mode = 1
    while True:
        cand = [s for s in range(S) if cnt[s] > 1]
        if not cand:
            break
        # child boxes + counts for every candidate
        cc = {}
        box = {}
        kq = np.zeros(n, dtype=np.int64)
        for s in cand:
            halfX = int(math.ceil(float(_f32(nx1[s] - nx0[s]) / _f32(2))))
            halfY = int(math.ceil(float(_f32(ny1[s] - ny0[s]) / _f32(2))))
            mx, my = nx0[s] + halfX, ny0[s] + halfY
            box[s] = [(nx0[s], mx, ny0[s], my), (mx, nx1[s], ny0[s], my),
                      (nx0[s], mx, my, ny1[s]), (mx, nx1[s], my, ny1[s])]
            cc[s] = [0, 0, 0, 0]
        for k in range(n):
            s = node[k]
            if cnt[s] > 1:
                mx, my = box[s][0][1], box[s][0][3]
                qd = (0 if x[k] < mx else 1) + (0 if y[k] < my else 2)
                kq[k] = qd
                cc[s][qd] += 1
        nne = {s: sum(1 for c in cc[s] if c > 0) for s in cand}
        eexp = {s: sum(1 for c in cc[s] if c > 1) for s in cand}
        if mode == 1:
            order = cand[:]                       # processing order = list order
            nproc = len(order)
        else:
            order = sorted(cand, key=lambda s: (cnt[s], seq[s]), reverse=True)
            size = S
            nproc = len(order)
            for r, s in enumerate(order):
                size += nne[s] - 1
                if size >= N:
                    nproc = r + 1
                    break
        proc = order[:nproc]
        split = set(proc)
        F = sum(nne[s] for s in proc)
        # positions
        P = 0
        block_start = {}
        ebase = {}
        E = 0
        for s in proc:
            P += nne[s]
            block_start[s] = F - P
            ebase[s] = E
            E += eexp[s]
        new = {}
        childpos = {}
        for s in proc:
            childpos[s] = [None] * 4
            e = ebase[s]
            for qd in range(4):
                if cc[s][qd] > 0:
                    after = sum(1 for q2 in range(qd + 1, 4) if cc[s][q2] > 0)
                    pos = block_start[s] + after
                    b = box[s][qd]
                    sq = 0
                    if cc[s][qd] > 1:
                        sq = e
                        e += 1
                    new[pos] = (b[0], b[1], b[2], b[3], cc[s][qd], sq)
                    childpos[s][qd] = pos
        qn = 0
        movepos = {}
        for s in range(S):
            if s not in split:
                pos = F + qn
                qn += 1
                new[pos] = (nx0[s], nx1[s], ny0[s], ny1[s], cnt[s], seq[s])
                movepos[s] = pos
        S2 = F + qn
        for k in range(n):
            s = node[k]
            node[k] = childpos[s][kq[k]] if s in split else movepos[s]
        prevS = S
        S = S2
        nx0 = [new[p][0] for p in range(S)]; nx1 = [new[p][1] for p in range(S)]
        ny0 = [new[p][2] for p in range(S)]; ny1 = [new[p][3] for p in range(S)]
        cnt = [new[p][4] for p in range(S)]; seq = [new[p][5] for p in range(S)]
        if S >= N or S == prevS:
            break
        if mode == 1 and S + 3 * E > N:
            mode = 2
    # best key per node: max response, first index wins
    best = [-1] * S
    for k in range(n):
        s = node[k]
        if best[s] < 0 or resp[k] > resp[best[s]]:
            best[s] = k
    return np.array(best, dtype=np.int64)
