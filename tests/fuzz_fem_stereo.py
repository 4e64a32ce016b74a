"""Differential fuzz of the FEM assembly / K*a path (random surface meshes, C3D6 / C3D8 / tet4, random material)
and of ComputeStereoMatches (random pair sizes) against the oracle."""
import sys
import numpy as np
import oracle
from orb_slam2_e_amd import ComputeStereoMatches, ORBextractor
from orb_slam2_e_amd.fem import FEA2, FEM_C3D6, FEM_C3D8, FEM_TET4, second_layer, extrude_elems
from orb_slam2_e_amd.synth import synth_stereo_pair

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n):
    kind = int(rng.integers(0, 4))
    if kind == 3:
        # tetrahedra (config 3's element): a small Kuhn grid with random jitter, a few nodes collapsed onto a neighbour (zero-volume
        # elements: Inf / NaN gradients) and, now and then, coordinates scaled until entries overflow
        from orb_slam2_e_amd.synth import synth_tet_mesh
        dims = tuple(int(v) for v in rng.integers(1, 5, 3))
        nodes, elems, _, _ = synth_tet_mesh(dims, int(rng.integers(0, 1 << 30)), jitter=float(rng.uniform(0, 0.4)))
        for k in rng.choice(len(nodes), int(rng.integers(0, 3)), replace=False):
            nodes[k] = nodes[(k + 1) % len(nodes)]
        nodes = (nodes.astype(np.float64) * float(rng.choice([1.0, 1.0, 1e-16, 1e-21, 1e-30, 1e15]))).astype(np.float32)
        elems = elems[rng.permutation(len(elems))]
        E = int(rng.integers(100, 100000)); nu = float(rng.uniform(0.05, 0.499))
        fea = FEA2(nodes, elems, FEM_TET4, E=E, nu=nu)
        fea.MatrixAssembly()
        K, ref = fea.K_dense(), oracle.fem_assemble_dense(4, nodes, elems, E, nu)
        nan = np.isnan(ref)
        if not (np.array_equal(np.isnan(K), nan) and np.array_equal(K.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])):
            bad += 1; print("MISMATCH fem tet", case, dims, E, nu, flush=True)
        continue
    g = int(rng.integers(3, 12))
    X, Y = np.meshgrid(np.arange(g, dtype=np.float32), np.arange(g, dtype=np.float32), indexing="ij")
    top = np.stack([X.ravel() + rng.normal(0, 0.15, g * g), Y.ravel() + rng.normal(0, 0.15, g * g), rng.normal(0, 0.3, g * g)], 1).astype(np.float32)
    nid = lambda i, j: i * g + j
    E = int(rng.integers(100, 100000)); nu = float(rng.uniform(0.05, 0.499))
    if kind == 0:
        faces = np.array([[nid(i, j), nid(i + 1, j), nid(i + 1, j + 1)] for i in range(g - 1) for j in range(g - 1)] +
                         [[nid(i, j), nid(i + 1, j + 1), nid(i, j + 1)] for i in range(g - 1) for j in range(g - 1)], np.int32)
        et, code = FEM_C3D6, 2
    else:
        faces = np.array([[nid(i, j), nid(i + 1, j), nid(i + 1, j + 1), nid(i, j + 1)] for i in range(g - 1) for j in range(g - 1)], np.int32)
        if kind == 2:                                  # a few degenerate quads (repeated node), as tri2quad leaves them
            k = rng.choice(len(faces), max(1, len(faces) // 6), replace=False); faces[k, 3] = faces[k, 0]
        et, code = FEM_C3D8, 1
    faces = faces[rng.permutation(len(faces))]
    nodes = second_layer(top, float(rng.uniform(0.2, 1.0)))
    elems = extrude_elems(faces, len(top))
    fea = FEA2(nodes, elems, et, E=E, nu=nu)
    fea.MatrixAssembly()
    K = oracle.fem_assemble_dense(code, nodes, elems, E, nu) if "E" in oracle.fem_assemble_dense.__code__.co_varnames else None
    if K is None:
        print("oracle.fem_assemble_dense has no material arguments"); sys.exit(2)
    ok = np.array_equal(fea.K_dense(), K, equal_nan=True)
    a = rng.normal(0, 1e-2, 3 * len(nodes)).astype(np.float32)
    if ok and not np.isnan(K).any():
        ok = np.array_equal(fea.ComputeForces(a)[0], oracle.fem_matvec_dense(K, a))
    if not ok:
        bad += 1; print("MISMATCH fem", case, kind, g, E, nu, flush=True)
    elif not np.isnan(K).any() and rng.random() < 0.5:
        # the LM hook on this mesh (levenberg.cpp:159-175): penalties, resident set-up, trials -- a bit-exact, energies within 1e-5;
        # the derived (mid-edge / barycentre) nodes of the C3D8 form on random earlier nodes
        ntop = len(top)
        ids = np.arange(ntop, 2 * ntop, dtype=np.int32)
        fea.ImposeDirichletEncastre_K(ids)
        oracle.fem_dirichlet_K(K, ids)
        nder = int(rng.integers(0, min(30, ntop - 2))) if rng.random() < 0.5 else 0
        npts = ntop - nder
        der = None
        if nder:
            der = []
            for d_ in range(nder):
                hi = npts + d_ if d_ % 3 == 2 else npts
                der.append([2, *rng.integers(0, hi, 2), 0] if rng.random() < 0.5 else [3, *rng.integers(0, hi, 3)])
            der = np.array(der, np.int32)
        u0 = nodes.ravel()
        fea.trial_setup(u0, ids, npts, der)
        for trial in range(2):
            pts = top[:npts].astype(np.float64) + rng.normal(0, 0.01, (npts, 3))
            a_, sE, nsE = fea.trial_energy(pts)
            oa = oracle.fem_trial_displacement(pts, der, u0, ids)
            of = oracle.fem_matvec_dense(K, oa)
            osE, onsE = oracle.fem_strain_energy(oa, of)
            # a^T f is a float sum with cancellation, and the reference forms it in Eigen (FEA2.cc:1877-1886: MultiplyMatricesEigen,
            # summation order unspecified) where the oracle adds left to right: two orders may differ by the rounding of the sum
            # itself, ~sqrt(n) eps sum|a_i f_i|, which on a random mesh can exceed 1e-5 of a strongly cancelled result
            slack = 2e-6 * float(np.abs(oa.astype(np.float64) * of.astype(np.float64)).sum())
            if not (np.array_equal(a_[0], oa) and abs(sE[0] - osE) <= max(1e-5 * abs(osE), slack) and
                    abs(nsE[0] - onsE) <= max(1e-5 * abs(onsE), slack / (len(oa) // 3))):
                bad += 1; print("MISMATCH LM trial", case, kind, g, nder, sE[0], osE, flush=True); break
for case in range(max(4, n // 8)):
    w = int(rng.integers(400, 1300)); h = int(rng.integers(200, 500))
    # the reference's stereo settings most of the time, otherwise any pyramid (the row band of a keypoint is +-2 scale[octave] rows)
    P = (1500, 1.2, 8, 20, 7) if rng.random() < 0.5 else (int(rng.integers(50, 2500)), float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0])), int(rng.integers(1, 9)),
                                                          int(rng.integers(8, 30)), int(rng.integers(3, 9)))
    left, right = synth_stereo_pair(int(rng.integers(0, 1000)), w=w, h=h, dmin=float(rng.uniform(0, 5)), dmax=float(rng.uniform(20, 90)))
    oL, oR = oracle.OrbOracle(*P), oracle.OrbOracle(*P)
    try:
        kL, dL = oL.extract(left); kR, dR = oR.extract(right)
    except RuntimeError:                               # a size the reference cannot run either (a level below the 30-px cell grid)
        try:
            ORBextractor(*P)(left)
            bad += 1; print("MISMATCH stereo: the oracle rejects", w, h, "the library does not", flush=True)
        except Exception:
            pass
        continue
    fx, bf = float(rng.uniform(300, 900)), float(rng.uniform(100, 500))
    mb = np.float32(bf) / np.float32(fx)
    ou, od, nd = oracle.stereo_matches(oL, oR, kL, dL, kR, dR, mb, np.float32(bf))
    try:
        eL, eR = ORBextractor(*P), ORBextractor(*P)
        eL(left); eR(right)
    except Exception as e:
        if "error -5" in str(e): continue         # a documented capacity limit (one level with a quota above 2047)
        raise
    gu, gd = ComputeStereoMatches(eL, eR, mb, np.float32(bf))
    if not (np.array_equal(gu.view(np.uint32), ou.view(np.uint32)) and np.array_equal(gd.view(np.uint32), od.view(np.uint32))):
        bad += 1; print("MISMATCH stereo", case, w, h, fx, bf, flush=True)
print("cases", n, "mismatches", bad)
sys.exit(1 if bad else 0)
