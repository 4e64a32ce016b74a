"""Per-frame FEM latency at the reference's mesh sizes (its own surface meshes, tests/golden): what
Optimizer::PoseOptimizationNR pays per call -- build + assemble (FEA2::Compute(1)) -- and per LM trial
(levenberg.cpp:159-175: Set_uf, ComputeDisplacement, ComputeForces, ComputeStrainEnergy)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle
from orb_slam2_e_amd.fem import FEA2, FEM_C3D6, second_layer, extrude_elems

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
for name in ("median", "p90", "large"):
    m = np.load(os.path.join(GOLD, f"fem_mesh_{name}.npz"))
    top, tris = m["points"], m["triangles"]
    p = top[tris]
    tris = tris[~((p[:, 0] == p[:, 1]).all(1) | (p[:, 0] == p[:, 2]).all(1) | (p[:, 1] == p[:, 2]).all(1))]
    nodes = second_layer(top, 0.5); elems = extrude_elems(tris, len(top))
    ids = np.arange(len(top), 2 * len(top), dtype=np.int32)

    def build():
        fea = FEA2(nodes, elems, FEM_C3D6); fea.MatrixAssembly(); fea.ImposeDirichletEncastre_K(ids)
        fea.trial_setup(nodes.ravel(), ids, len(top), None)
        return fea
    fea = build()
    t0 = time.perf_counter()
    for _ in range(20): fea = build()
    t_build = (time.perf_counter() - t0) / 20
    pts = top.astype(np.float64) + 0.003
    for _ in range(10): fea.trial_energy(pts, want_a=False)
    t0 = time.perf_counter()
    for _ in range(100): fea.trial_energy(pts, want_a=False)
    t_trial = (time.perf_counter() - t0) / 100
    # the reference's own cost per trial: dense Ksize^2 multiply (+ the copies into Eigen)
    K = fea.K_dense(); a = fea.trial_energy(pts)[0][0]
    t0 = time.perf_counter()
    for _ in range(5): oracle.fem_matvec_dense(K, a)
    t_ref = (time.perf_counter() - t0) / 5
    print("%-6s nTop %4d  Ksize %5d : build+assemble %.3f ms   LM trial %.3f ms   (oracle dense K*a alone %.3f ms)" % (
        name, len(top), 6 * len(top), t_build * 1e3, t_trial * 1e3, t_ref * 1e3))
