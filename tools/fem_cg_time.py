"""CG timing of a uniform batch (BASELINE config 3's mesh): ms per iteration and the per-kernel split.
usage (GPU box): python3 tools/fem_cg_time.py [nmesh] [iters] [ncell]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch

nm = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ncell = int(sys.argv[3]) if len(sys.argv) > 3 else 12
nodes, tets, fixed, load = synth_tet_batch(nm, ncell, seed=11)
fea = FEA2(nodes, tets, FEM_TET4)
fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (nm, 1)); b[:, fixed] = 0
fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()   # (warm-up of the same length: every launch in a trace of this tool is alike)
for rep in range(3):
    fea.cg_setup(b)
    t0 = time.perf_counter(); fea.cg_iterate(iters); x, rel = fea.cg_result(); dt = time.perf_counter() - t0
    print(f"nmesh {nm} ndof {fea.Ksize} iters {iters}: {dt / iters * 1e3:.4f} ms/iter (incl. result copy), "
          f"{nm * iters / dt / 1e6:.3f} M mesh-iter/s, relres max {rel.max():.3e}", flush=True)
fea.profile(True); fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()
for k, v in fea.profile_read().items():
    if v[1]:
        print(f"  {k}: {v[0] / v[1]:.4f} ms x {v[1]}")
# k_fem_spmv by itself on the same resident matrix and vectors, as bench.py's `spmv_alone` (50 launches): the kernel north_star names
fea.profile(4); fea.spmv_repeat(50); fea.cg_result()
v = fea.profile_read()["k_fem_spmv"]
print(f"  k_fem_spmv alone: {v[0] / max(v[1], 1):.4f} ms x {v[1]}")
