"""How differently the meshes of a batch converge (iterations to relres 1e-8 per mesh, in slices of 25, both preconditioners): what a
per-mesh stop inside the resident CG kernel could save.  usage (GPU box): PYTHONPATH=. python tools/fem_convergence_spread.py"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from orb_slam2_e_amd.fem import FEA2, FEA2Batch, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch, synth_tet_batch_distinct
def spread(fea, b, n, label):
    for kind in ("jacobi", "two_level"):
        fea.cg_preconditioner(kind); fea.cg_setup(b)
        done = np.zeros(n, int); it = 0
        while (done == 0).any() and it < 4000:
            fea.cg_iterate(25); it += 25
            rel = fea.cg_relres()
            done[(done == 0) & (rel <= 1e-8)] = it
        print(label, kind, "max", done.max(), "mean", round(done.mean(), 1), "min", done.min(), "mean/max", round(done.mean() / done.max(), 3), flush=True)
nodes_l, tets_l, fixed_l, load_l = synth_tet_batch_distinct(256, seed=11)
fb = FEA2Batch(nodes_l, tets_l, FEM_TET4); fb.MatrixAssembly()
fixed = np.concatenate([fb.dof0[k] + fx for k, fx in enumerate(fixed_l)]).astype(np.int32)
fb.eliminate_dofs(fixed); b = np.concatenate(load_l)[None].copy(); b[:, fixed] = 0
spread(fb, b, 256, "distinct")
nodes, tets, fixed, load = synth_tet_batch(256, 12, seed=11)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (256, 1)); b[:, fixed] = 0
spread(fea, b, 256, "uniform")
