"""The reference's own prism meshes (tests/golden) under the CG, both preconditioners, device and oracle: K as FEA2 assembles it is indefinite
(eigenvalues printed), the CG breaks down (NaN) on every one of them on both sides -- why the CG legs of the bench and the tests use the
tetrahedral path.  usage (GPU box): PYTHONPATH=. python tools/fem_c3d6_two_level.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, oracle
from orb_slam2_e_amd.fem import FEA2, FEM_C3D6, extrude_elems, second_layer
GOLD = os.path.join(os.getcwd(), "tests", "golden")
for name in ("min", "median", "p90", "large"):
    m = np.load(os.path.join(GOLD, f"fem_mesh_{name}.npz")); top, tris = m["points"], m["triangles"]
    p = top[tris]; tris = tris[~((p[:, 0] == p[:, 1]).all(1) | (p[:, 0] == p[:, 2]).all(1) | (p[:, 1] == p[:, 2]).all(1))]
    nodes = second_layer(top, 0.5); elems = extrude_elems(tris, len(top))
    ids = np.arange(len(top) + 1, 2 * len(top) + 1, dtype=np.int32)      # 1-based: the bottom layer
    fea = FEA2(nodes, elems, FEM_C3D6); fea.MatrixAssembly(); fea.ImposeDirichletEncastre_K(ids)
    n = 3 * len(nodes)
    rng = np.random.default_rng(1); b = np.zeros(n); b[: 3 * len(top)] = rng.normal(size=3 * len(top))
    rp, col, val = fea.csr(0)
    A = np.zeros((n, n)); A[np.repeat(np.arange(n), np.diff(rp)), col] = val
    sym = np.abs(A - A.T).max() / np.abs(A).max()
    ev = np.linalg.eigvalsh((A + A.T) / 2)
    mask = np.zeros(n, np.uint8)
    for k in range(3): mask[3 * (ids - 1) + k] = 1
    out = [name, "n", n, "asym %.1e" % sym, "eig min %.2e max %.2e" % (ev[0], ev[-1])]
    for kind in ("jacobi", "two_level"):
        fea.cg_preconditioner(kind)
        x, it, rel = fea.solve_cg(b, iters=3000, tol=1e-8)
        if kind == "jacobi": ox, oit, orel = oracle.fem_cg(rp, col, val, b, 3000, 1e-8)
        else: ox, oit, orel = oracle.fem_cg_two_level(rp, col, val, b, 3000, nodes, mask, 1e-8)
        out += [kind, it, "%.1e" % rel[0], "oracle", oit, "xdiff %.1e" % (np.abs(x[0] - ox).max() / max(np.abs(ox).max(), 1e-300))]
    print(*out, flush=True)
