"""CG timing of a batch of meshes with their own topologies (the bench's distinct batch).
usage (GPU box): python3 tools/fem_cg_time_distinct.py [nmesh] [iters]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam2_e_amd.fem import FEA2Batch, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch_distinct

nm = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nodes_l, tets_l, fixed_l, load_l = synth_tet_batch_distinct(nm, seed=11)
fea = FEA2Batch(nodes_l, tets_l, FEM_TET4)
fixed = np.concatenate([fea.dof0[k] + fx for k, fx in enumerate(fixed_l)]).astype(np.int32)
b = np.concatenate(load_l)[None].copy(); b[:, fixed] = 0
fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()   # (warm-up of the same length: every launch in a trace of this tool is alike)
for rep in range(3):
    fea.cg_setup(b)
    t0 = time.perf_counter(); fea.cg_iterate(iters); x, rel = fea.cg_result(); dt = time.perf_counter() - t0
    print(f"nmesh {nm} dofs {fea.Ksize} iters {iters}: {dt / iters * 1e3:.4f} ms/iter (incl. result copy), "
          f"{nm * iters / dt / 1e6:.3f} M mesh-iter/s, relres max {rel.max():.3e}", flush=True)
fea.profile(True); fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()
for k, v in fea.profile_read().items():
    if v[1]:
        print(f"  {k}: {v[0] / v[1]:.4f} ms x {v[1]}")
# k_fem_spmv by itself on the same resident matrix and vectors, as bench.py's `spmv_alone` (50 launches): the kernel north_star names
fea.profile(4); fea.spmv_repeat(50); fea.cg_result()
v = fea.profile_read()["k_fem_spmv"]
print(f"  k_fem_spmv alone: {v[0] / max(v[1], 1):.4f} ms x {v[1]}")
