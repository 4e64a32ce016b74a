"""Per-phase clock counts of k_fem_cg_xcd (a -DXG_TIMING build of the library selected with ORBX_LIB): averages per iteration of workgroups
0..3.  usage (GPU box): ORBX_LIB=orb_slam2_e_amd/lib_xgtiming.so python3 tools/fem_xcd_phases.py [ncell] [iters] [two_level]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_mesh
from orb_slam2_e_amd._lib import lib
ncell = int(sys.argv[1]) if len(sys.argv) > 1 else 12
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 500
nodes, tets, fixed, load = synth_tet_mesh(ncell)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = load.copy()[None]; b[:, fixed] = 0
if len(sys.argv) > 3 and sys.argv[3] == "two_level": fea.cg_preconditioner("two_level")
fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()
fea.cg_setup(b); fea.cg_iterate(iters); fea.cg_result()
out = np.zeros(32 + 8 * 64, np.uint32)
L = lib(); L.fem_debug_xcd_timing.argtypes = [C.c_void_p, C.c_void_p]
assert L.fem_debug_xcd_timing(fea._h, out.ctypes.data) == 0
names = ["spmv phase 1", "spmv phase 2 + puts", "hop A (poll pAp, Ap row)", "update + puts", "hop B + new p (two-level: the new p only)",
         "(two-level) hop B poll", "(two-level) restriction + puts", "(two-level) hop C + coarse solve"]
for r in range(3):
    t = out[2 + 8 * r: 10 + 8 * r]
    print("rank", r, "clocks per iteration:", ", ".join(f"{n} {int(v)}" for n, v in zip(names, t)), "| sum", int(t.sum()))
if os.environ.get("XG_ALL"):
    for r in range(64):
        t = out[32 + 8 * r: 40 + 8 * r]
        h = int(t[7])
        if t.sum(): print("rank %2d" % r, " ".join("%5d" % int(v) for v in t[:7]), "| sum", int(t[:7].sum()), "| se %d sh %d cu %2d simd %d wave %2d" % ((h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3, h & 15))
