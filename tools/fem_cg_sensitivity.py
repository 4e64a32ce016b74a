"""One case of tests/fuzz_fem_cg.py replayed (uniform batches only): the batch result against the single-mesh paths and the oracle, the coarse\nmatrices and kept modes of both sides, and the sensitivity of the two-level PCG to last-bit perturbations of the coarse inverse.\nusage (GPU box): PYTHONPATH=. python3 tools/fem_cg_sensitivity.py <seed> <case>"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import oracle
from orb_slam2_e_amd.fem import FEA2, FEA2Batch, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_mesh
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    nm = int(rng.choice([1, 3, 15, 16, 40, 63, 64, 65, 100, 130]))
    big = rng.random() < 0.3
    lo, hi = (11, 15) if big else (2, 9)
    iters = int(rng.integers(1, 60))
    seg = rng.random() < 0.5
    two = case % 2 == 1
    warp = two and rng.random() < 0.4
    def shape(nodes):
        if not warp: return nodes
        nodes = np.array(nodes, np.float32, copy=True); t = nodes[..., 0] - nodes[..., 0].min()
        nodes[..., 0] = (t.max() * (t / t.max()) ** 3).astype(np.float32)
        return nodes
    if seg:
        dims = [tuple(int(v) for v in rng.integers(lo, hi + 1, 3)) for _ in range(nm)]
        if case == target: raise SystemExit("target is a segmented case")
        rng.integers(0, nm)
        continue
    d = tuple(int(v) for v in rng.integers(lo, hi + 1, 3))
    base = synth_tet_mesh(d, 1000 * case)
    nodes = shape(np.stack([base[0] + rng.normal(0, 0.01, base[0].shape).astype(np.float32) for _ in range(nm)]))
    b = np.tile(base[3], (nm, 1)) * rng.uniform(0.5, 2.0, (nm, 1)); b[:, base[2]] = 0
    pick = sorted(set([0, nm - 1, int(rng.integers(0, nm))]))
    if case != target: continue
    print("case", case, "nm", nm, "dims", d, "iters", iters, "two", two, "warp", warp, "pick", pick)
    fea = FEA2(nodes, base[1], FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(base[2])
    if two: fea.cg_preconditioner("two_level")
    x, done, rel = fea.solve_cg(b, iters=iters, tol=0.0)
    for k in pick:
        rp, col, val = fea.csr(k)
        mk = np.zeros(len(b[k]), np.uint8); mk[base[2]] = 1
        res = {}
        for it in (5, 10, 20, 30, iters):
            ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[k], it, nodes[k], mk) if two else oracle.fem_cg(rp, col, val, b[k], it, 0.0)
            res[it] = orel
        one = FEA2(nodes[k:k + 1], base[1], FEM_TET4); one.MatrixAssembly(); one.eliminate_dofs(base[2])
        if two: one.cg_preconditioner("two_level")
        out = {}
        for env in ("1", "0"):
            os.environ["FEM_CG_XCD"] = env
            x1, _, rel1 = one.solve_cg(b[k:k + 1], iters=iters, tol=0.0)
            out[env] = (x1[0].copy(), rel1[0])
        A = None
        from scipy.sparse import csr_matrix
        A = csr_matrix((np.asarray(val, np.float64), col, rp), shape=(len(b[k]), len(b[k])))
        ev = np.linalg.eigvalsh(A.toarray())
        print(" mesh", k, "oracle relres by iterations", {i: float("%.4g" % v) for i, v in res.items()}, "| batch", rel[k], "| single xcd", out["1"][1], "| single per-phase", out["0"][1],
              "| x batch vs xcd", np.abs(x[k] - out["1"][0]).max() / np.abs(ox).max(), "xcd vs per-phase", np.abs(out["1"][0] - out["0"][0]).max(), "| batch vs oracle", np.abs(x[k] - ox).max() / np.abs(ox).max(),
              "| eig min %.3g max %.3g" % (ev.min(), ev.max()))
    # the coarse matrices and which coarse dofs each side keeps (mesh 64)
    k = pick[-1]
    rp, col, val = fea.csr(k)
    mk = np.zeros(len(b[k]), np.uint8); mk[base[2]] = 1
    oAc = oracle.fem_coarse_matrix(rp, col, val, nodes[k], mk)
    dAc = fea.cg_coarse_matrix(k)
    print("Ac max rel diff", np.abs(oAc - dAc).max() / np.abs(oAc).max())
    def pivots(A):
        A = 0.5 * (A + A.T); n = 48; L = np.zeros((n, n)); keep = np.diag(A) > 1e-12 * np.diag(A).max(); ratio = np.zeros(n)
        for j in range(n):
            if not keep[j]: continue
            s = A[j, j] - (L[j, :j] ** 2).sum(); ratio[j] = s / A[j, j]
            if not (s > 1e-4 * A[j, j]): keep[j] = False; L[j, :j] = 0; continue
            L[j, j] = np.sqrt(s)
            for i in range(j + 1, n):
                if keep[i]: L[i, j] = (A[i, j] - (L[i, :j] * L[j, :j]).sum()) / L[j, j]
        return keep, ratio
    ko, ro = pivots(oAc); kd, rd = pivots(dAc)
    print("kept oracle", int(ko.sum()), "device", int(kd.sum()), "differ at", np.nonzero(ko != kd)[0].tolist())
    small = np.nonzero((ro < 1e-6) | (rd < 1e-6))[0]
    print("pivot / diagonal of the near-dependent modes:", [(int(i), float("%.3g" % ro[i]), float("%.3g" % rd[i])) for i in small])
    agg, q = oracle.fem_coarse_space(nodes[k]); print("aggregate sizes", np.bincount(agg, minlength=8).tolist())
    # sensitivity: the same two-level PCG in numpy (double), with the oracle's inverse and with inverses perturbed in the last bits
    from scipy.sparse import csr_matrix
    A = csr_matrix((np.asarray(val, np.float64), col, rp), shape=(len(b[k]), len(b[k])))
    n = len(b[k]); nn = n // 3
    Z = np.zeros((n, 48))
    for i in range(nn):
        a = agg[i]; qq = q[i].astype(np.float64) if q.ndim == 2 else q[3 * i:3 * i + 3].astype(np.float64)
        for c in range(3):
            if mk[3 * i + c]: continue
            Z[3 * i + c, 6 * a + c] = 1
        # rotations: c = v + omega x q
        if not mk[3 * i]: Z[3 * i, 6 * a + 4] = qq[2]; Z[3 * i, 6 * a + 5] = -qq[1]
        if not mk[3 * i + 1]: Z[3 * i + 1, 6 * a + 5] = qq[0]; Z[3 * i + 1, 6 * a + 3] = -qq[2]
        if not mk[3 * i + 2]: Z[3 * i + 2, 6 * a + 3] = qq[1]; Z[3 * i + 2, 6 * a + 4] = -qq[0]
    Aci = oracle.fem_coarse_inverse(oAc.copy())
    dinv = 1.0 / A.diagonal()
    def pcg(Aci, its):
        x = np.zeros(n); r = b[k].astype(np.float64).copy(); z = dinv * r + Z @ (Aci @ (Z.T @ r)); p = z.copy(); rz = r @ z
        for _ in range(its):
            Ap = A @ p; al = rz / (p @ Ap); x += al * p; r -= al * Ap
            z = dinv * r + Z @ (Aci @ (Z.T @ r)); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
        return np.linalg.norm(r) / np.linalg.norm(b[k])
    base_rel = pcg(Aci, iters)
    rs = np.random.default_rng(1)
    pert = [pcg(Aci * (1 + 1e-15 * rs.standard_normal(Aci.shape)), iters) for _ in range(5)]
    print("numpy two-level PCG relres after", iters, ":", base_rel, "| with the inverse perturbed by 1e-15 relative:", [float("%.4g" % v) for v in pert], "| max |Aci|", np.abs(Aci).max())
