"""One config-3 mesh (6,591 dofs): CG iterations per second of the one-launch kernel (k_fem_cg_xcd) against the launch-per-phase path
(FEM_CG_XCD=0), 200 and 1000 iterations per call, median of 9 calls.  usage (GPU box): python3 tools/fem_xcd_time.py [ncell]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_mesh

ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 12
nodes, tets, fixed, load = synth_tet_mesh(ncell)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = load.copy()[None]; b[:, fixed] = 0
for env in ("1", "0"):
    os.environ["FEM_CG_XCD"] = env
    for iters in (200, 1000):
        t = []
        for rep in range(9):
            fea.cg_setup(b)
            t0 = time.perf_counter(); fea.cg_iterate(iters); x, rel = fea.cg_result(); t.append(time.perf_counter() - t0)
        med = float(np.median(t))
        print(f"FEM_CG_XCD={env} ndof {fea.Ksize} iters {iters}: {med / iters * 1e6:.2f} us per iteration (median call incl. result copy), "
              f"{iters / med / 1e3:.1f} k iterations/s, best {iters / min(t) / 1e3:.1f} k, relres {rel[0]:.3e}", flush=True)
# ... and under the two-level preconditioner: us per iteration, and the solve to a relative residual of 1e-8 (set-up included, slices of
# 25 iterations each followed by the residual's trip to the host, as bench.py's `to_tolerance`)
fea.cg_preconditioner("two_level")
for env in ("1", "0"):
    os.environ["FEM_CG_XCD"] = env
    t = []
    for rep in range(7):
        fea.cg_setup(b)
        t0 = time.perf_counter(); fea.cg_iterate(400); x, rel = fea.cg_result(); t.append(time.perf_counter() - t0)
    tt = []
    for rep in range(5):
        t0 = time.perf_counter(); fea.cg_setup(b); it = 0; r = 1.0
        while it < 6000 and r > 1e-8:
            fea.cg_iterate(25); it += 25; r = float(fea.cg_relres().max())
        tt.append(time.perf_counter() - t0)
    print(f"two-level FEM_CG_XCD={env} ndof {fea.Ksize}: {np.median(t) / 400 * 1e6:.2f} us per iteration; to 1e-8: {it} iterations, {np.median(tt) * 1e3:.2f} ms (median of 5, set-up included)", flush=True)
fea.cg_preconditioner("jacobi")
os.environ["FEM_CG_XCD"] = "1"
tt = []
for rep in range(5):
    t0 = time.perf_counter(); fea.cg_setup(b); it = 0; r = 1.0
    while it < 6000 and r > 1e-8:
        fea.cg_iterate(25); it += 25; r = float(fea.cg_relres().max())
    tt.append(time.perf_counter() - t0)
print(f"Jacobi FEM_CG_XCD=1: to 1e-8: {it} iterations, {np.median(tt) * 1e3:.2f} ms")
