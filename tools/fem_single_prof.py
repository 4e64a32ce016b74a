"""One 6,591-dof mesh, 1000 CG iterations (hipGraph path): target for
rocprofv3 --kernel-trace --stats to read true per-kernel durations."""
import sys, time
import numpy as np
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch

nm = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nodes, tets, fixed, load = synth_tet_batch(nm, 12, seed=11)
fea = FEA2(nodes, tets, FEM_TET4)
fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (nm, 1)); b[:, fixed] = 0
fea.cg_setup(b); fea.cg_iterate(100); fea.cg_result()
fea.cg_setup(b)
t0 = time.perf_counter(); fea.cg_iterate(1000); fea.cg_result(); dt = time.perf_counter() - t0
print("meshes", nm, "us/iter", dt / 1000 * 1e6)
