import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_mesh
from orb_slam2_e_amd._lib import lib
ncell = int(sys.argv[1]); iters = int(sys.argv[2])
nodes, tets, fixed, load = synth_tet_mesh(ncell)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = load.copy()[None]; b[:, fixed] = 0
fea.cg_setup(b)
L = lib()
info = np.zeros(8, np.int32); plan = np.zeros((64, 4), np.int32)
L.fem_debug_xcd.argtypes = [C.c_void_p] * 4 + [C.c_int]
L.fem_debug_xcd(fea._h, info.ctypes.data, plan.ctypes.data, None, 0)
print("info cg_xcd,P,ldr,ldq,lds,total,nchunk,nchunk_s:", info.tolist()); print("plan", plan[:info[1]].tolist())
fea.cg_iterate(iters)
try:
    x, rel = fea.cg_result(); print("ok rel", rel)
except Exception as e:
    print("FAILED", e)
g = np.zeros((info[5], 4), np.uint32)
L.fem_debug_xcd(fea._h, info.ctypes.data, plan.ctypes.data, g.ctypes.data, int(info[5]))
a = int(info[4]) & 0xffffffff
print("abort word: %#x -> where %d rank %d wave %d it %d" % (a, a & 0xff, (a >> 8) & 0xff, (a >> 16) & 0xf, (a >> 20) & 0x7ff))
ndof, nchunk, ns = fea.Ksize, int(info[6]), int(info[7])
names = [("ap", 0, ndof), ("r", ndof, ndof), ("pap", 2 * ndof, 2 * ns), ("rz", 2 * ndof + 2 * ns, 2 * nchunk), ("rr", 2 * ndof + 2 * ns + 2 * nchunk, 2 * nchunk)]
for nm, o, n in names:
    tags = g[o:o + n, 2]
    ok = g[o:o + n, 3] == (g[o:o + n, 0] ^ g[o:o + n, 1] ^ g[o:o + n, 2] ^ np.uint32(0x5bd1e995))
    print(nm, "tags:", dict(zip(*np.unique(tags, return_counts=True))), "check ok", int(ok.sum()), "of", n)
