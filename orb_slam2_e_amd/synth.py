"""Seeded synthetic inputs for tests and bench.

numpy-only; identical on the build container and the GPU box (same numpy).
"""
import numpy as np


def _scene(rng, w, h, nrect, ndisc):
    """Mid-grey 128 + nrect filled rectangles (side 8-80, intensity U[0,255]) + ndisc filled discs (radius 4-30)."""
    img = np.full((h, w), 128, dtype=np.int16)
    for _ in range(nrect):
        sw, sh = rng.integers(8, 81, size=2)
        x0 = rng.integers(-sw // 2, w - sw // 2)
        y0 = rng.integers(-sh // 2, h - sh // 2)
        img[max(y0, 0):max(y0 + sh, 0), max(x0, 0):max(x0 + sw, 0)] = rng.integers(0, 256)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(ndisc):
        r = int(rng.integers(4, 31))
        cx = int(rng.integers(0, w)); cy = int(rng.integers(0, h))
        v = int(rng.integers(0, 256))
        y0, y1 = max(cy - r, 0), min(cy + r + 1, h)
        x0, x1 = max(cx - r, 0), min(cx + r + 1, w)
        m = (yy[y0:y1, x0:x1] - cy) ** 2 + (xx[y0:y1, x0:x1] - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = v
    return img


def synth_frame(k, w=640, h=480):
    """Frame k (seed 1000+k): mid-grey 128 + 300 filled rectangles (side 8-80,
    intensity U[0,255]) + 200 filled discs (radius 4-30) + noise U[-6,6], clamped."""
    rng = np.random.Generator(np.random.PCG64(1000 + k))
    img = _scene(rng, w, h, 300, 200)
    img += rng.integers(-6, 7, size=(h, w), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


SEQ_LEN = 64          # frames of one camera pan = one GPU's shard of the batch
SEQ_STEP = (2, 1)     # pan per frame, px


def synth_sequence(n, w=640, h=480, start=0):
    """Frames start .. start+n-1 of the bench workload: consecutive frames see the SAME scene under a camera pan, so
    that matching frame i against frame i+1 finds hundreds of true correspondences (independent scenes give none).
    Frames [64 q, 64 q + 64) are one pan over scene q (seed 1000 + 64 q, the content statistics of synth_frame on a
    canvas large enough for the pan): frame j of the pan is the w x h window at (2 j, j) plus its own noise
    U[-6,6] (its own generator, seed 500000 + frame number), clamped."""
    out = np.empty((n, h, w), np.uint8)
    canvas, cq = None, -1
    for i in range(n):
        k = start + i
        q, j = divmod(k, SEQ_LEN)
        if q != cq:
            cw, chh = w + SEQ_STEP[0] * (SEQ_LEN - 1), h + SEQ_STEP[1] * (SEQ_LEN - 1)
            area = (cw * chh) / float(w * h)
            canvas = _scene(np.random.Generator(np.random.PCG64(1000 + SEQ_LEN * q)), cw, chh,
                            int(round(300 * area)), int(round(200 * area)))
            cq = q
        rng = np.random.Generator(np.random.PCG64(500000 + k))
        x0, y0 = SEQ_STEP[0] * j, SEQ_STEP[1] * j
        img = canvas[y0:y0 + h, x0:x0 + w] + rng.integers(-6, 7, size=(h, w), dtype=np.int16)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def synth_frames(n, w=640, h=480, start=0):
    return np.stack([synth_frame(start + k, w, h) for k in range(n)])


def synth_descriptors(n=2000, seed=7, flip_p=0.08, n_replace=200):
    """Match input of config 2: A random; B = A with bits flipped w.p. flip_p,
    rows shuffled, n_replace rows replaced by fresh random bytes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    flips = rng.random((n, 256)) < flip_p
    B = A ^ np.packbits(flips, axis=1, bitorder="little")
    perm = rng.permutation(n)
    B = B[perm]
    rep = rng.choice(n, size=min(n_replace, n), replace=False)
    B[rep] = rng.integers(0, 256, size=(len(rep), 32), dtype=np.uint8)
    return A, np.ascontiguousarray(B), perm


def _tet_grid(ncell):
    """ncell: one number (cube) or (nx, ny, nz) cells."""
    nx, ny, nz = (ncell, ncell, ncell) if np.isscalar(ncell) else ncell
    X, Y, Z = np.meshgrid(np.arange(nx + 1, dtype=np.float64), np.arange(ny + 1, dtype=np.float64),
                          np.arange(nz + 1, dtype=np.float64), indexing="ij")
    nodes = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    # Kuhn split: 6 tets per cube along the main diagonal (0,0,0)-(1,1,1)
    perms = [(0, 1, 2), (0, 2, 1), (1, 0, 2), (1, 2, 0), (2, 0, 1), (2, 1, 0)]
    I, J, K = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    base = np.stack([I.ravel(), J.ravel(), K.ravel()], 1)                    # [cells, 3]
    tets = np.zeros((len(base), 6, 4), np.int64)
    nid = lambda c: (c[:, 0] * (ny + 1) + c[:, 1]) * (nz + 1) + c[:, 2]
    for t, pm in enumerate(perms):
        cur = base.copy()
        tets[:, t, 0] = nid(cur)
        for v, ax in enumerate(pm):
            cur = cur.copy(); cur[:, ax] += 1
            tets[:, t, v + 1] = nid(cur)
    tets = tets.reshape(-1, 4)
    p = nodes[tets]
    det = np.einsum("ij,ij->i", p[:, 1] - p[:, 0], np.cross(p[:, 2] - p[:, 0], p[:, 3] - p[:, 0]))
    neg = det < 0
    tets[neg, 2], tets[neg, 3] = tets[neg, 3].copy(), tets[neg, 2].copy()
    return nodes, tets


def synth_tet_mesh(ncell=12, seed=11, jitter=0.1):
    """Config-3 mesh: ncell^3 unit cubes, each split into 6 tets (Kuhn),
    node jitter U[-jitter,jitter]*spacing on interior nodes; 12 -> 10,368 tets,
    2,197 nodes, 6,591 dofs.  Returns nodes[nn,3] f32, tets[ne,4] i32 (positively
    oriented), fixed dofs (z=0 face), load vector (unit traction on z=max face)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nodes, tets = _tet_grid(ncell)
    dims = np.array((ncell, ncell, ncell) if np.isscalar(ncell) else ncell, np.float64)
    interior = ((nodes > 0) & (nodes < dims)).all(1)
    ij = nodes[:, :2].copy()
    ztop = nodes[:, 2] == dims[2]
    fixed_nodes = np.where(nodes[:, 2] == 0)[0]
    nodes[interior] += rng.uniform(-jitter, jitter, size=(int(interior.sum()), 3))
    fixed = (3 * fixed_nodes[:, None] + np.arange(3)[None]).ravel()
    load = np.zeros(3 * len(nodes))
    top = np.where(ztop)[0]
    wgt = np.where((ij[top, 0] == 0) | (ij[top, 0] == dims[0]), 0.5, 1.0) * \
        np.where((ij[top, 1] == 0) | (ij[top, 1] == dims[1]), 0.5, 1.0)
    load[3 * top + 2] = wgt  # unit traction x tributary area
    return nodes.astype(np.float32), tets.astype(np.int32), fixed.astype(np.int32), load


def synth_tet_chain(nn, seed=5):
    """A mesh with exactly `nn` nodes (any nn >= 4): nodes on a helix, tets (i, i+1, i+2, i+3) -- a banded matrix of 7 node
    blocks per row; for probing size limits that grids cannot hit.  Returns nodes, tets (positively oriented), fixed dofs
    (the first three nodes), load (unit pull on the last node)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    t = np.arange(nn) * 0.9
    nodes = np.stack([np.cos(t), np.sin(t), 0.35 * t], 1) + rng.uniform(-0.02, 0.02, (nn, 3))
    tets = np.stack([np.arange(nn - 3) + k for k in range(4)], 1)
    p = nodes[tets]
    vol = np.einsum("ij,ij->i", np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0]), p[:, 3] - p[:, 0])
    flip = vol < 0
    tets[flip] = tets[flip][:, [1, 0, 2, 3]]
    fixed = np.arange(9, dtype=np.int32)
    load = np.zeros(3 * nn); load[-1] = 1.0
    return nodes.astype(np.float32), tets.astype(np.int32), fixed, load


def synth_tet_batch(nmesh, ncell=12, seed=11):
    """nmesh meshes sharing the Kuhn topology, each with its own jitter (distinct matrices)."""
    base, tets, fixed, load = synth_tet_mesh(ncell, seed)
    nodes = np.stack([synth_tet_mesh(ncell, seed + 1000 * m)[0] for m in range(nmesh)]) if nmesh > 1 else base[None]
    return nodes, tets, fixed, load


def synth_tet_batch_distinct(nmesh, seed=11, base=12, dims_seed=88):
    """nmesh meshes with their OWN topologies: grids of (nx, ny, nz) cells, nx, ny, nz in base-2 .. base+2 drawn per mesh
    (about the config-3 size on average; the sizes depend on dims_seed only, so every rank of a multi-GPU run holds the
    same sizes), each with its own jitter (seed).  Returns lists (nodes, tets, fixed dofs, loads)."""
    rng = np.random.Generator(np.random.PCG64(dims_seed))
    out = ([], [], [], [])
    for m in range(nmesh):
        dims = tuple(int(v) for v in rng.integers(base - 2, base + 3, 3))
        for lst, v in zip(out, synth_tet_mesh(dims, seed + 1000 * m)):
            lst.append(v)
    return out


def synth_stereo_pair(k=0, w=1242, h=375, dmin=2.0, dmax=60.0):
    """Config-5 input: KITTI-shaped pair; right = left warped by a smooth
    synthetic disparity field in [dmin, dmax] px (right(x) = left(x + d(x, y)), bilinear)."""
    left = synth_frame(500 + k, w, h)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    d = dmin + (dmax - dmin) * (0.5 + 0.25 * np.sin(xx / 97.0 + 0.3 * k) + 0.25 * np.cos(yy / 61.0))
    xs = np.clip(xx + d, 0, w - 1.001)
    x0 = np.floor(xs).astype(np.int64); fx = xs - x0
    rows = np.arange(h)[:, None]
    L = left.astype(np.float64)
    right = (1 - fx) * L[rows, x0] + fx * L[rows, x0 + 1]
    return left, np.clip(np.rint(right), 0, 255).astype(np.uint8)


def synth_keyframe_pair_case(kps1, desc1, kps2, desc2, seed=0, nwords=90, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157,
                             baseline=0.5372, stereo1=None):
    """Config 5's SearchForTriangulation input from a stereo pair's own keypoints: the left and the right image as two
    keyframes one baseline apart (KITTI00-02.yaml:8-11,25: fx = fy = 718.856, bf = 386.1448 -> b = 0.5372 m).  FeatureVectors
    from a flat stand-in vocabulary (node = the nearest of `nwords` descriptors drawn from keyframe 1: matching features
    mostly share a node, as with DBoW2 at levelsup = 4); 30 % of the keypoints already hold a map point, stereo flags from
    `stereo1` (e.g. mvuRight >= 0) / at random.  F12 = K^-T [t]x K^-1 for camera 2 = camera 1 moved by (b, 0, 1e-3 b): the
    epipolar lines are the image rows to within a hundredth of a pixel and the epipole stays finite.
    Returns (fv1, fv2, has_mp1, has_mp2, stereo1, stereo2, F12, ex, ey)."""
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    d1 = np.asarray(desc1, np.uint8); d2 = np.asarray(desc2, np.uint8)
    words = d1[rng.choice(len(d1), nwords, replace=False)]
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)

    def nodes_of(d):
        dist = lut[d[:, None, :] ^ words[None, :, :]].sum(2)
        return (np.argmin(dist, 1) * 7 + 11).astype(np.int32), dist.min(1)

    n1, m1 = nodes_of(d1); n2, m2 = nodes_of(d2)
    from .vocabulary import feature_vector_arrays
    fv1 = feature_vector_arrays(n1, rng.random(len(d1)) < 0.97)      # stopped words are in no node
    fv2 = feature_vector_arrays(n2, rng.random(len(d2)) < 0.97)
    has1 = rng.random(len(d1)) < 0.3; has2 = rng.random(len(d2)) < 0.3
    s1 = np.asarray(stereo1, bool) if stereo1 is not None else rng.random(len(d1)) < 0.5
    s2 = rng.random(len(d2)) < 0.5
    t = np.array([baseline, 0.0, 1e-3 * baseline])                    # X2 = X1 - t
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F12 = (np.linalg.inv(K).T @ tx @ np.linalg.inv(K)).astype(np.float32)
    C2 = -t
    ex = np.float32(fx * C2[0] / C2[2] + cx); ey = np.float32(fy * C2[1] / C2[2] + cy)
    return fv1, fv2, has1, has2, s1, s2, F12, ex, ey


def synth_initialization_case(seed=0, n1=2000, n2=2200):
    """Two frames for ORBmatcher::SearchForInitialization: frame 2's keypoints are frame 1's, moved by a few pixels, with
    slightly perturbed descriptors.  Returns (kps1, desc1, kps2, desc2, vbPrevMatched, bounds)."""
    from .extractor import KP_DTYPE
    rng = np.random.Generator(np.random.PCG64(9000 + seed))
    k1 = np.zeros(n1, KP_DTYPE)
    k1["x"] = rng.uniform(0, 640, n1); k1["y"] = rng.uniform(0, 480, n1)
    k1["octave"] = rng.choice(8, n1, p=[0.5, 0.15, 0.1, 0.08, 0.07, 0.05, 0.03, 0.02]); k1["angle"] = rng.uniform(0, 360, n1)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    src = rng.integers(0, n1, n2)
    k2 = k1[src].copy()
    k2["x"] += rng.normal(0, 6, n2); k2["y"] += rng.normal(0, 6, n2)
    d2 = d1[src] ^ np.packbits(rng.random((n2, 256)) < 0.03, axis=1, bitorder="little")
    prev = np.stack([k1["x"], k1["y"]], 1).astype(np.float32)
    return k1, d1, k2, d2, prev, (0.0, 0.0, 640.0, 480.0)


def synth_projection_case(seed, n=2000, nq=3000, hot=400, stereo=False):
    """Many queries aim at few keypoints, so the in-loop assignment matters."""
    from .extractor import KP_DTYPE
    from .matcher import ORBmatcher
    rng = np.random.default_rng(seed)
    kps = np.zeros(n, KP_DTYPE)
    kps["x"] = rng.uniform(-5, 645, n); kps["y"] = rng.uniform(-5, 485, n)
    kps["octave"] = rng.integers(0, 8, n); kps["angle"] = rng.uniform(0, 360, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    desc[rng.choice(n, min(200, n // 2), replace=False)] = desc[min(1, n - 1)]
    src = rng.choice(rng.choice(n, hot, replace=False), nq)
    q = np.zeros(nq, ORBmatcher.WQ_DTYPE)
    q["u"] = kps["x"][src] + rng.normal(0, 2, nq); q["v"] = kps["y"][src] + rng.normal(0, 2, nq)
    q["r"] = rng.choice([7.0, 15.0, 30.0], nq) * (1.2 ** kps["octave"][src])
    lvl = kps["octave"][src]
    q["min_level"] = lvl - 1; q["max_level"] = lvl + rng.integers(0, 2, nq)
    q["xr"] = q["u"] - rng.uniform(0, 30, nq)
    qd = desc[src] ^ np.packbits(rng.random((nq, 256)) < 0.04, axis=1, bitorder="little")
    # rotation: most matches agree on one of three rotations, the rest are scattered
    qa = (kps["angle"][src] + rng.choice([10.0, 95.0, 200.0, 300.0, 333.0], nq, p=[0.5, 0.25, 0.15, 0.05, 0.05])
          + rng.normal(0, 2, nq)) % 360
    takes = (rng.random(nq) < 0.8).astype(np.uint8)
    occ = (rng.random(n) < 0.05).astype(np.uint8)
    ur = np.where(rng.random(n) < 0.5, kps["x"] - rng.uniform(0, 30, n), -1).astype(np.float32) if stereo else None
    return q, qd, qa.astype(np.float32), takes, kps, desc, (0.0, 0.0, 640.0, 480.0), occ, ur


def synth_bow_case(seed, n1=2000, n2=2100, nnodes=90):
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    nl = min(500, n1 // 4)
    d1[rng.choice(n1, nl, replace=False)] = d1[:min(5, n1)][rng.integers(0, 5, nl) % min(5, n1)]      # look-alikes: the "already matched" skip matters
    node1 = rng.integers(0, nnodes, n1) * 7 + 3
    node1[d1[:, 0] % 5 == 0] = 3                                                   # one crowded node
    src = rng.integers(0, n1, n2)
    d2 = d1[src] ^ np.packbits(rng.random((n2, 256)) < 0.03, axis=1, bitorder="little")
    node2 = node1[src].copy()
    move = rng.random(n2) < 0.1
    node2[move] = rng.integers(0, nnodes + 20, move.sum()) * 7 + 3                # some land in other (or unseen) nodes
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = ((a1[src] - rng.choice([15.0, 100.0, 260.0], n2, p=[0.75, 0.2, 0.05]) + rng.normal(0, 2, n2)) % 360).astype(np.float32)
    valid1 = (rng.random(n1) < 0.8).astype(np.uint8); valid2 = (rng.random(n2) < 0.85).astype(np.uint8)
    keep1 = rng.random(n1) < 0.97; keep2 = rng.random(n2) < 0.97                   # stop words never enter the FeatureVector
    return d1, a1, node1, keep1, valid1, d2, a2, node2, keep2, valid2


def synth_vocabulary(k=10, L=6, seed=0, nstop=50, nties=64):
    """A complete k-ary DBoW2 vocabulary tree of L levels in the shape of ORBvoc (k = 10, L = 6: 1,111,111 nodes, 10^6 words,
    35 MB of node descriptors; the real file is missing from the reference mount, Vocabulary/ORBvoc.txt.tar.gz), nodes numbered
    breadth-first as DBoW2 creates them (TemplatedVocabulary.h:560-640, HKmeansStep): the children of node g are g k + 1 .. g k + k.
    A child is its parent with every bit flipped with probability 2^-min(depth, 5), so that the descent of a noisy leaf descriptor
    is decided by a few bits at the deep levels; `nties` nodes get a sibling's descriptor (the first minimum must win,
    TemplatedVocabulary.h:1240-1250), `nstop` words weight 0 (stop words).  Returns the flat arrays of orbm_vocab_create /
    ORBVocabulary: (child_off, child_ids, node_desc, node_word, node_weight, L)."""
    rng = np.random.default_rng(seed)
    level_n = [k ** d for d in range(L + 1)]
    n = sum(level_n); nint = n - level_n[L]
    desc = np.empty((n, 32), np.uint8)
    desc[0] = rng.integers(0, 256, 32, dtype=np.uint8)
    first = 0
    for d in range(1, L + 1):
        pf, cf, cn = first, first + level_n[d - 1], level_n[d]
        flips = rng.integers(0, 256, (cn, 32), dtype=np.uint8)
        for _ in range(min(d, 5) - 1):
            flips &= rng.integers(0, 256, (cn, 32), dtype=np.uint8)
        desc[cf:cf + cn] = np.repeat(desc[pf:cf], k, axis=0) ^ flips
        first = cf
    if nties and k > 3:
        g = rng.choice(np.arange(nint), min(nties, nint), replace=False)      # parents whose child 3 repeats child 1
        desc[g * k + 4] = desc[g * k + 2]
    child_off = np.minimum(np.arange(n + 1, dtype=np.int64), nint) * k
    child_ids = np.arange(1, n, dtype=np.int32)
    word = np.full(n, -1, np.int32); word[nint:] = np.arange(level_n[L], dtype=np.int32)
    weight = np.zeros(n, np.float64); weight[nint:] = rng.uniform(0.5, 8.0, level_n[L])
    if nstop:
        weight[nint + rng.choice(level_n[L], min(nstop, level_n[L]), replace=False)] = 0.0
    return child_off.astype(np.int32), child_ids, desc, word, weight, L


def synth_vocabulary_features(voc, n, seed=1, flip_p=0.04):
    """n descriptors near random words of a synth_vocabulary tree (leaf descriptor with bits flipped with probability flip_p)."""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc[3] >= 0)[0]
    return voc[2][rng.choice(leaves, n)] ^ np.packbits(rng.random((n, 256)) < flip_p, axis=1, bitorder="little")


# ---- scenes for the projection searches as whole functions (ORBmatcher.cc:491-604, :1303-1527, :1529-1800) ----------

SCENE_CAM = dict(fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, mbf=40.0)   # TUM1-like pinhole, 640 x 480


def _rot(rng, deg):
    """Small random rotation (Rodrigues), float64."""
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(deg) * rng.uniform(0.3, 1.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * K @ K


def _pose(R, t):
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return T.astype(np.float32)


def synth_view_frame(rng, T, pts, mp_desc, n, nlevels=8, scale=1.2, stereo=False, noise=1.5, flip=0.04, mp_angle=None, turn=(0.0,)):
    """A frame that sees the map: n keypoints, most of them noisy projections of map points (descriptor = the point's with
    a few bits flipped), the rest clutter.  Returns (kps, desc, uright or None, src = map point behind each keypoint or -1)."""
    from .extractor import KP_DTYPE
    c = SCENE_CAM
    Pc = pts @ T[:3, :3].T.astype(np.float64) + T[:3, 3].astype(np.float64)
    z = Pc[:, 2]
    u = c["fx"] * Pc[:, 0] / z + c["cx"]; v = c["fy"] * Pc[:, 1] / z + c["cy"]
    vis = np.nonzero((z > 0.3) & (u > -20) & (u < 660) & (v > -20) & (v < 500))[0]
    nsee = min(len(vis), int(0.8 * n))
    src = np.full(n, -1, np.int64)
    src[:nsee] = rng.choice(vis, nsee, replace=len(vis) < nsee)
    rng.shuffle(src)
    kps = np.zeros(n, KP_DTYPE)
    seen = src >= 0
    kps["x"] = np.where(seen, u[src] + rng.normal(0, noise, n), rng.uniform(-5, 645, n))
    kps["y"] = np.where(seen, v[src] + rng.normal(0, noise, n), rng.uniform(-5, 485, n))
    lvl = np.clip(np.round(np.log(np.clip(8.0 / np.maximum(z[src], 0.3), 1e-3, None)) / np.log(scale) + rng.normal(0, 0.8, n)), 0, nlevels - 1)
    kps["octave"] = np.where(seen, lvl, rng.integers(0, nlevels, n)).astype(np.int32)
    kps["angle"] = rng.uniform(0, 360, n)
    if mp_angle is not None:     # a seen point's keypoint: the point's own angle turned by one of a few common amounts
        kps["angle"] = np.where(seen, (mp_angle[src] + rng.choice(turn, n) + rng.normal(0, 2, n)) % 360, kps["angle"])
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    fl = np.packbits(rng.random((n, 256)) < flip, axis=1, bitorder="little")
    desc[seen] = mp_desc[src[seen]] ^ fl[seen]
    ur = None
    if stereo:
        ur = np.where(seen & (rng.random(n) < 0.7), kps["x"] - c["mbf"] / np.maximum(z[src], 0.3) + rng.normal(0, 0.7, n), -1.0).astype(np.float32)
    return kps, desc, ur, src


def synth_tracking_scene(seed, n=2000, nmp=2500, stereo=False, motion="none", nlevels=8, scale=1.2):
    """One scene for the four whole-function searches.  Returns a dict:
    cam (fx, fy, cx, cy), mb, mbf, bounds, scale_factors, log_scale_factor;
    map: pos, normal, mind, maxd, desc (nmp points, a fifth of them near-duplicates of others so that list entries compete
    for the same keypoint; some behind the camera, out of range or seen from behind);
    cur: Tcw, kps, desc, uright, src; last: Tlw + per-keypoint arrays (valid, mp index, takes, octave, angle);
    kf2: a second view (T2w, kps, desc, src) for the key-frame forms; Scw / (s12, R12, t12) for the Sim3 forms."""
    rng = np.random.default_rng(seed)
    c = SCENE_CAM
    sf = (scale ** np.arange(nlevels)).astype(np.float32)
    for i in range(1, nlevels):
        sf[i] = np.float32(sf[i - 1] * np.float32(scale))
    mb = np.float32(c["mbf"]) / np.float32(c["fx"])
    Rcw = _rot(rng, 8); tcw = rng.normal(0, 0.3, 3)
    Tcw = _pose(Rcw, tcw)
    # map points: a slab in front of the current camera, plus outliers
    Pc = np.stack([rng.uniform(-0.68, 0.68, nmp), rng.uniform(-0.52, 0.52, nmp), np.ones(nmp)], 1) * rng.uniform(0.8, 12, (nmp, 1))
    far = rng.random(nmp) < 0.08
    Pc[far] *= [2.5, 2.5, 1.0]                             # outside the image
    Pc[rng.random(nmp) < 0.05, 2] *= -1                    # behind the camera
    Pc[rng.integers(0, nmp)] = [0.3, -0.2, 0.0]            # z = 0: the projection divides by zero
    dup = rng.choice(nmp, nmp // 5, replace=False)
    Pc[dup] = Pc[rng.integers(0, nmp, len(dup))] + rng.normal(0, 0.01, (len(dup), 3))
    pos = ((Pc - tcw) @ Rcw).astype(np.float32)            # world = Rcw^T (Pc - tcw)
    Ow = -Rcw.T @ tcw
    dvec = pos.astype(np.float64) - Ow
    dist = np.linalg.norm(dvec, axis=1)
    lvl0 = rng.integers(0, nlevels, nmp)
    maxd = (dist * rng.uniform(0.6, 1.6, nmp) * sf[lvl0]).astype(np.float32)
    mind = (maxd / sf[nlevels - 1]).astype(np.float32)
    nrm = dvec / np.maximum(dist[:, None], 1e-6) + rng.normal(0, 0.5, (nmp, 3))     # mean viewing direction, camera -> point
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    mp_desc = rng.integers(0, 256, (nmp, 32), dtype=np.uint8)
    mp_desc[dup] = mp_desc[rng.integers(0, nmp, len(dup))] ^ np.packbits(rng.random((len(dup), 256)) < 0.02, axis=1, bitorder="little")
    mp_angle = rng.uniform(0, 360, nmp)
    kps, desc, ur, src = synth_view_frame(rng, Tcw, pos.astype(np.float64), mp_desc, n, nlevels, scale, stereo, mp_angle=mp_angle)
    # last frame: the camera a few baselines behind / ahead along z (forward / backward motion), or almost in place
    dz = {"forward": 3.0, "backward": -3.0, "none": 0.2}[motion] * float(mb)
    Rlc = _rot(rng, 2)
    Tlw = _pose(Rlc @ Rcw, Rlc @ tcw + np.array([0.0, 0.0, dz]))
    nl = n
    lmp = rng.integers(0, nmp, nl)
    take = rng.random(nl) < 0.75
    pick = src[rng.integers(0, n, int(take.sum()))]
    lmp[take] = np.where(pick >= 0, pick, lmp[take])
    lval = (rng.random(nl) < 0.8).astype(np.uint8)
    ltakes = (rng.random(nl) < 0.8).astype(np.uint8)
    loct = np.clip(lvl0[lmp] + rng.integers(-1, 2, nl), 0, nlevels - 1).astype(np.int32)
    lang = ((mp_angle[lmp] + rng.choice([12.0, 100.0, 215.0, 320.0], nl, p=[0.6, 0.25, 0.1, 0.05]) + rng.normal(0, 2, nl)) % 360).astype(np.float32)
    occ = (rng.random(n) < 0.05).astype(np.uint8)
    # a second key frame looking at the same map from the side, and the similarity between the two
    R2 = _rot(rng, 10) @ Rcw; t2 = _rot(rng, 5) @ tcw + rng.normal(0, 0.25, 3)
    T2w = _pose(R2, t2)
    k2, d2, _, src2 = synth_view_frame(rng, T2w, pos.astype(np.float64), mp_desc, n + 100, nlevels, scale, False, mp_angle=mp_angle,
                                       turn=(15.0, 15.0, 15.0, 110.0, 110.0, 250.0, 333.0))
    T12 = Tcw.astype(np.float64) @ np.linalg.inv(T2w.astype(np.float64))
    s12 = np.float32(rng.uniform(0.9, 1.1))
    R12 = (T12[:3, :3] @ _rot(rng, 0.5)).astype(np.float32); t12 = (T12[:3, 3] + rng.normal(0, 0.02, 3)).astype(np.float32)
    sw = np.float32(rng.uniform(0.8, 1.25))
    Scw = Tcw.copy(); Scw[:3, :] *= sw
    return dict(cam=(np.float32(c["fx"]), np.float32(c["fy"]), np.float32(c["cx"]), np.float32(c["cy"])), mb=mb, mbf=np.float32(c["mbf"]),
                bounds=(0.0, 0.0, 640.0, 480.0), scale_factors=sf, log_scale_factor=np.float32(np.log(np.float32(scale))), nlevels=nlevels,
                pos=pos, normal=nrm, mind=mind, maxd=maxd, mp_desc=mp_desc,
                Tcw=Tcw, kps=kps, desc=desc, uright=ur, src=src, occupied=occ,
                Tlw=Tlw, last_mp=lmp, last_valid=lval, last_takes=ltakes, last_octave=loct, last_angle=lang,
                T2w=T2w, kps2=k2, desc2=d2, src2=src2, s12=s12, R12=R12, t12=t12, Scw=Scw)
