import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch
nodes, tets, fixed, load = synth_tet_batch(256, 12)
fea = FEA2(nodes, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
b = np.tile(load, (256, 1)); b[:, fixed] = 0
fea.cg_setup(b)
fea.profile(4)
fea.spmv_repeat(50); fea.cg_result()
pr = fea.profile_read()["k_fem_spmv"]
print("spmv avg ms", pr[0]/pr[1], "GB/s", 256*(fea.nnz*8+(fea.Ksize+1)*4+2*fea.Ksize*8)/(pr[0]/pr[1]*1e-3)/1e9)
