"""Free device memory before and after 300 create / use / destroy cycles of an extractor handle and of a FEM model (both preconditioners).\nusage (GPU box): python3 tools/leak_check.py   -- round 5: +0.0 MiB for both."""
import os, sys, gc, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from orb_slam2_e_amd import ORBextractor, ORBmatcher
from orb_slam2_e_amd.synth import synth_frame, synth_tet_mesh
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
def free_mb():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
img = synth_frame(1)
def cycle_extract():
    ex = ORBextractor(2000, 1.2, 8, 20, 7); ex(img); ex.extract_batch(np.stack([img] * 3)); del ex
nodes, tets, fixed, load = synth_tet_mesh(6)
def cycle_fem():
    fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
    b = load.copy()[None]; b[:, fixed] = 0
    fea.cg_setup(b); fea.cg_iterate(10); fea.cg_result(); fea.cg_preconditioner("two_level"); fea.cg_setup(b); fea.cg_iterate(10); fea.cg_result(); del fea
ex0 = ORBextractor(2000, 1.2, 8, 20, 7); k, d = ex0(img)
m = ORBmatcher(0.6, True)
def cycle_frame():
    f = m.frame_create(k, d, 0, 0, 640, 480) if hasattr(m, "frame_create") else None
    del f
for name, fn, n in (("extractor", cycle_extract, 300), ("fem model", cycle_fem, 300)):
    for _ in range(20): fn()
    gc.collect(); a = free_mb()
    for _ in range(n): fn()
    gc.collect(); b = free_mb()
    print("%-10s %d create/use/destroy cycles: free device memory %.1f -> %.1f MiB (%+.1f)" % (name, n, a, b, b - a), flush=True)
