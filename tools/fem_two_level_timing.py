"""Config 3 (256 meshes of 12^3 cells, 6,591 dofs each): per-iteration time and time to ||r|| <= 1e-8 ||b|| of the resident CG under
point-Jacobi and under the two-level preconditioner.  usage: python tools/fem_two_level_timing.py [nmesh]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch
nmesh = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=12)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
for kind in ("jacobi", "two_level", "jacobi", "two_level"):
    fea.cg_preconditioner(kind)
    t = time.perf_counter(); fea.cg_setup(b); torch.cuda.synchronize(); ts = time.perf_counter() - t
    fea.cg_iterate(50); torch.cuda.synchronize()
    t = time.perf_counter(); fea.cg_iterate(200); torch.cuda.synchronize(); per = (time.perf_counter() - t) / 200
    t = time.perf_counter(); x, it, rel = fea.solve_cg(b, iters=20000, tol=1e-8); tt = time.perf_counter() - t
    print("%-9s setup %.2f ms, %.4f ms per iteration, to 1e-8: %d iterations, %.1f ms (setup and copies included), relres max %.2e" %
          (kind, ts * 1e3, per * 1e3, it, tt * 1e3, rel.max()), flush=True)
