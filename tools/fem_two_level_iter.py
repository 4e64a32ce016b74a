"""Per-iteration time of the resident CG on config 3's batch under both preconditioners (timing only; ORBX_LIB picks the library)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch
nmesh = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ncell = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=ncell)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
out = []
for kind in ("jacobi", "two_level"):
    fea.cg_preconditioner(kind); fea.cg_setup(b); fea.cg_iterate(100); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); fea.cg_iterate(200); torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t) / 200)
    out.append("%s %.4f ms" % (kind, best * 1e3))
print(os.environ.get("ORBX_LIB", "default")[-9:], nmesh, "meshes", ncell, "cells", " | ".join(out), flush=True)
