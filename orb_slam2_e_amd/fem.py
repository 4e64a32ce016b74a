"""Host-side mirror of the FEM core of FEA2 over the C-ABI (ctypes).

Names follow Thirdparty/g2o/g2o/FEA/include/FEA2.h: the constructor takes the
material constants of FEA2::FEA2 (E unsigned, nu, h, fg, nElType) and the methods
are the reference's (SetSecondLayer, MatrixAssembly, ImposeDirichletEncastre_K,
ComputeDisplacement, ComputeForces, ComputeStrainEnergy, NormalizeStrainEnergy),
plus the CG solve that fills the slot of the dead InvertMatrixEigen path.
"""
import ctypes as C

import numpy as np

from ._lib import bind, check, lib, ptr as _p

FEM_C3D8, FEM_C3D6, FEM_TET4 = 1, 2, 4
_NPE = {FEM_C3D8: 8, FEM_C3D6: 6, FEM_TET4: 4}
_BOUND = False




def _bind(L):
    global _BOUND
    if _BOUND:
        return
    bind(L.fem_create, [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint, C.c_float,
                             C.c_float, C.c_void_p])
    bind(L.fem_second_layer, [C.c_void_p, C.c_int, C.c_float, C.c_void_p])
    bind(L.fem_dirichlet_penalty, [C.c_void_p, C.c_void_p, C.c_int, C.c_float])
    bind(L.fem_dirichlet_eliminate, [C.c_void_p, C.c_void_p, C.c_int])
    bind(L.fem_displacement, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p])
    bind(L.fem_cg, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p])
    bind(L.fem_cg_iterate, [C.c_void_p, C.c_int, C.c_void_p])
    bind(L.fem_spmv_repeat, [C.c_void_p, C.c_int, C.c_void_p])
    for n in ("fem_destroy", "fem_assemble"):
        getattr(L, n).argtypes = [C.c_void_p]
    _BOUND = True


def second_layer(top, h):
    """FEA2::SetSecondLayer: nodes = top || top - (h,h,h)."""
    L = lib(); _bind(L)
    top = np.ascontiguousarray(top, np.float32)
    out = np.zeros((2 * len(top), 3), np.float32)
    check(L.fem_second_layer(_p(top), len(top), h, _p(out)))
    return out


def extrude_elems(faces, ntop):
    """Element node ids as MatrixAssemblyC3D8/6 forms them (FEA2.cc:1392-1399,
    :1518-1523): top ids || top ids + nTop."""
    faces = np.asarray(faces, np.int32)
    return np.ascontiguousarray(np.concatenate([faces, faces + ntop], axis=1), np.int32)


class FEA2:
    def __init__(self, nodes, elems, nElType, E=3500, nu=0.495, fg=0.577350269):
        """nodes: [nn,3] or [nmesh,nn,3] float32; elems: [ne,npe] int32."""
        self._L = lib(); _bind(self._L)
        nodes = np.ascontiguousarray(nodes, np.float32)
        if nodes.ndim == 2:
            nodes = nodes[None]
        self.nmesh, self.nn = nodes.shape[0], nodes.shape[1]
        elems = np.ascontiguousarray(elems, np.int32).reshape(-1, _NPE[nElType])
        self.ne, self.npe, self.nElType = len(elems), _NPE[nElType], nElType
        self._h = C.c_void_p()
        check(self._L.fem_create(nElType, _p(nodes), self.nmesh, self.nn, _p(elems), self.ne, int(E), nu, fg,
                                 C.byref(self._h)))
        nm, nd, nnz = C.c_int(), C.c_int(), C.c_int64()
        check(self._L.fem_sizes(self._h, C.byref(nm), C.byref(nd), C.byref(nnz)))
        self.Ksize, self.nnz = nd.value, nnz.value

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.fem_destroy(h)
            self._h = None

    def _vec(self, a, dtype):
        return np.ascontiguousarray(a, dtype).reshape(self.nmesh, self.Ksize)

    def material(self):
        lam, G = C.c_float(), C.c_float()
        D = np.zeros(36, np.float32)
        check(self._L.fem_material(self._h, C.byref(lam), C.byref(G), _p(D)))
        return lam.value, G.value, D.reshape(6, 6)

    def MatrixAssembly(self):
        check(self._L.fem_assemble(self._h))

    def ImposeDirichletEncastre_K(self, ids, Klarge=100000000.0):
        ids = np.ascontiguousarray(ids, np.int32)
        check(self._L.fem_dirichlet_penalty(self._h, _p(ids), len(ids), Klarge))

    def eliminate_dofs(self, dofs):
        dofs = np.ascontiguousarray(dofs, np.int32)
        check(self._L.fem_dirichlet_eliminate(self._h, _p(dofs), len(dofs)))

    def Kei(self, elem, mesh=0):
        nd = 3 * self.npe
        out = np.zeros((nd, nd), np.float32)
        check(self._L.fem_get_ke(self._h, mesh, elem, _p(out)))
        return out

    def csr(self, mesh=0):
        rp = np.zeros(self.Ksize + 1, np.int32); col = np.zeros(self.nnz, np.int32); val = np.zeros(self.nnz, np.float32)
        check(self._L.fem_get_csr(self._h, mesh, _p(rp), _p(col), _p(val)))
        return rp, col, val

    def K_dense(self, mesh=0):
        rp, col, val = self.csr(mesh)
        K = np.zeros((self.Ksize, self.Ksize), np.float32)
        rows = np.repeat(np.arange(self.Ksize), np.diff(rp))
        K[rows, col] = val
        return K

    def ComputeDisplacement(self, uf, u0, ids, Klarge=100000000.0):
        uf = self._vec(uf, np.float32); u0 = self._vec(u0, np.float32)
        ids = np.ascontiguousarray(ids, np.int32)
        a = np.zeros_like(uf)
        check(self._L.fem_displacement(self._h, _p(uf), _p(u0), _p(ids), len(ids), Klarge, _p(a)))
        return a

    def ComputeForces(self, a):
        a = self._vec(a, np.float32)
        f = np.zeros_like(a)
        check(self._L.fem_matvec(self._h, _p(a), _p(f)))
        return f

    def ComputeStrainEnergy(self, a):
        """Returns (sE, nsE) per mesh: |a^T K a| and sE / int(Ksize/3)."""
        a = self._vec(a, np.float32)
        sE = np.zeros(self.nmesh, np.float32); nsE = np.zeros(self.nmesh, np.float32)
        check(self._L.fem_strain_energy(self._h, _p(a), _p(sE), _p(nsE)))
        return sE, nsE

    def trial_setup(self, u0, ids, npoints, derived=None, Klarge=100000000.0):
        """Resident LM hook state (levenberg.cpp:159-175): u0, Dirichlet ids, vNewPointsBase."""
        u0 = np.ascontiguousarray(u0, np.float32).reshape(self.Ksize)
        ids = np.ascontiguousarray(ids, np.int32)
        der = np.ascontiguousarray(derived if derived is not None else np.zeros((0, 4)), np.int32).reshape(-1, 4)
        self._npoints = npoints
        bind(self._L.fem_trial_setup, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int])
        check(self._L.fem_trial_setup(self._h, _p(u0), _p(ids), len(ids), Klarge, npoints, _p(der), len(der)))

    def trial_energy(self, points, want_a=True):
        """Set_uf + ComputeDisplacement + ComputeForces + ComputeStrainEnergy + NormalizeStrainEnergy
        for the optimiser's current vertex estimates (double).  Returns (a, sE, nsE); want_a=False: what the hook itself
        reads back -- the two energies only (a is None)."""
        pts = np.ascontiguousarray(points, np.float64).reshape(self.nmesh, self._npoints, 3)
        a = np.zeros((self.nmesh, self.Ksize), np.float32) if want_a else None
        sE = np.zeros(self.nmesh, np.float32); nsE = np.zeros(self.nmesh, np.float32)
        check(self._L.fem_trial_energy(self._h, _p(pts), _p(a) if want_a else None, _p(sE), _p(nsE)))
        return a, sE, nsE

    def solve_cg(self, b, iters=200, tol=0.0):
        b = self._vec(b, np.float64)
        x = np.zeros_like(b); rel = np.zeros(self.nmesh, np.float64); done = C.c_int(0)
        check(self._L.fem_cg(self._h, _p(b), _p(x), iters, tol, C.byref(done), _p(rel)))
        return x, done.value, rel

    # resident / timing API
    def cg_preconditioner(self, kind):
        """fem_cg_preconditioner: "jacobi" (default) or "two_level" (Jacobi + rigid-body modes of 2 x 2 x 2 aggregates)."""
        k = {"jacobi": 0, "two_level": 1}.get(kind, kind)
        check(bind(self._L.fem_cg_preconditioner, [C.c_void_p, C.c_int])(self._h, int(k)))

    def one_launch_stats(self):
        """(launches of the one-launch CG kernel, how many of them gave up and were made good on the launch-per-phase path)."""
        a, b = C.c_int64(0), C.c_int64(0)
        check(bind(self._L.fem_cg_one_launch_stats, [C.c_void_p, C.c_void_p, C.c_void_p])(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def cg_coarse_matrix(self, mesh=0):
        Ac = np.zeros((48, 48), np.float64)
        check(bind(self._L.fem_cg_coarse_matrix, [C.c_void_p, C.c_int, C.c_void_p])(self._h, mesh, _p(Ac)))
        return Ac

    def cg_setup(self, b):
        b = self._vec(b, np.float64)
        check(self._L.fem_cg_setup(self._h, _p(b)))

    def cg_iterate(self, n, stream=None):
        check(self._L.fem_cg_iterate(self._h, n, C.c_void_p(stream) if stream else None))

    def spmv_repeat(self, n, stream=None):
        check(self._L.fem_spmv_repeat(self._h, n, C.c_void_p(stream) if stream else None))

    def cg_result(self):
        x = np.zeros((self.nmesh, self.Ksize), np.float64); rel = np.zeros(self.nmesh, np.float64)
        check(self._L.fem_cg_result(self._h, _p(x), _p(rel)))
        return x, rel

    def cg_relres(self):
        """||r|| / ||b|| per mesh after the iterations run so far (synchronises; the iterate stays on the device)."""
        rel = np.zeros(getattr(self, "nseg", self.nmesh), np.float64)
        check(self._L.fem_cg_result(self._h, None, _p(rel)))
        return rel

    def profile(self, on):
        """on: False/0 = off, True = every kernel kind, int = bit mask of kinds
        (k_fem_ke 1, k_fem_assemble 2, k_fem_spmv 4, k_fem_cg_update 8, k_fem_cg_dir 16)."""
        check(self._L.fem_profile_enable(self._h, -1 if on is True else int(on)))

    def profile_read(self):
        names = (C.c_char_p * 16)(); ms = (C.c_double * 16)(); ln = (C.c_int64 * 16)(); nk = C.c_int(0)
        check(self._L.fem_profile_read(self._h, 16, names, ms, ln, C.byref(nk)))
        return {names[i].decode(): (ms[i], ln[i]) for i in range(nk.value)}


class FEA2Batch(FEA2):
    """A batch of meshes with their OWN topologies (fem_create_batch) -- the reference builds a new mesh on every
    PoseOptimizationNR call (src/Optimizer.cc:480, FEA2.cc:80-121).  The batch is one block-diagonal system in global
    numbering: vectors are one array of all dofs (mesh k owns dofs [dof0[k], dof0[k+1])), Dirichlet ids / dofs are global;
    strain energies and CG residuals come back one per mesh.  Kei / csr / K_dense address a mesh's own elements, rows and
    columns.  Methods are those of FEA2 (the LM hook excepted)."""

    def __init__(self, nodes_list, elems_list, nElType, E=3500, nu=0.495, fg=0.577350269):
        self._L = lib(); _bind(self._L)
        self.npe, self.nElType = _NPE[nElType], nElType
        nodes_list = [np.ascontiguousarray(n, np.float32).reshape(-1, 3) for n in nodes_list]
        elems_list = [np.ascontiguousarray(e, np.int32).reshape(-1, self.npe) for e in elems_list]
        self.nseg = len(nodes_list)
        nn = np.array([len(n) for n in nodes_list], np.int32); ne = np.array([len(e) for e in elems_list], np.int32)
        nodes = np.ascontiguousarray(np.concatenate(nodes_list), np.float32)
        elems = np.ascontiguousarray(np.concatenate(elems_list), np.int32)
        self._h = C.c_void_p()
        bind(self._L.fem_create_batch, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_float,
                                             C.c_float, C.c_void_p])
        check(self._L.fem_create_batch(nElType, self.nseg, _p(nn), _p(ne), _p(nodes), _p(elems), int(E), nu, fg, C.byref(self._h)))
        nm, nd, nnz = C.c_int(), C.c_int(), C.c_int64()
        check(self._L.fem_sizes(self._h, C.byref(nm), C.byref(nd), C.byref(nnz)))
        assert nm.value == self.nseg
        self.Ksize, self.nnz = nd.value, nnz.value          # totals over the batch
        self.nmesh = 1                                      # vectors: one array of all dofs
        self.node0 = np.zeros(self.nseg + 1, np.int32); self.elem0 = np.zeros(self.nseg + 1, np.int32); self.nnz0 = np.zeros(self.nseg + 1, np.int32)
        check(self._L.fem_batch_offsets(self._h, _p(self.node0), _p(self.elem0), _p(self.nnz0)))
        self.dof0 = 3 * self.node0

    def csr(self, mesh=0):
        n = int(self.dof0[mesh + 1] - self.dof0[mesh]); nz = int(self.nnz0[mesh + 1] - self.nnz0[mesh])
        rp = np.zeros(n + 1, np.int32); col = np.zeros(nz, np.int32); val = np.zeros(nz, np.float32)
        check(self._L.fem_get_csr(self._h, mesh, _p(rp), _p(col), _p(val)))
        return rp, col, val

    def K_dense(self, mesh=0):
        rp, col, val = self.csr(mesh)
        n = len(rp) - 1
        K = np.zeros((n, n), np.float32)
        K[np.repeat(np.arange(n), np.diff(rp)), col] = val
        return K

    def ComputeStrainEnergy(self, a):
        a = self._vec(a, np.float32)
        sE = np.zeros(self.nseg, np.float32); nsE = np.zeros(self.nseg, np.float32)
        check(self._L.fem_strain_energy(self._h, _p(a), _p(sE), _p(nsE)))
        return sE, nsE

    def solve_cg(self, b, iters=200, tol=0.0):
        b = self._vec(b, np.float64)
        x = np.zeros_like(b); rel = np.zeros(self.nseg, np.float64); done = C.c_int(0)
        check(self._L.fem_cg(self._h, _p(b), _p(x), iters, tol, C.byref(done), _p(rel)))
        return x, done.value, rel

    def cg_result(self):
        x = np.zeros((1, self.Ksize), np.float64); rel = np.zeros(self.nseg, np.float64)
        check(self._L.fem_cg_result(self._h, _p(x), _p(rel)))
        return x, rel

    def trial_setup(self, *a, **k):
        raise NotImplementedError("the LM hook works on one mesh per model")


class _PlanInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("ndof", "nblk", "spb", "spmv_lds", "fused_lds", "nchunk_tot", "nchunk_s_tot", "resident",
                                         "resident_big", "resident_lds", "nrcd", "maxel")] + [("nnz", C.c_int64), ("ncontrib", C.c_int64),
                                                                                                     ("rows_lds", C.c_int32), ("reserved", C.c_int32)]


def plan_single_cg(elems, nn, nElType):
    """fem_plan_single_cg: how the CG of ONE mesh of this topology will run (host only).  Returns a dict: eligible (the one-launch
    kernel k_fem_cg_xcd), workgroups, chunks_per_workgroup, lds, nchunk, nchunk_s, plan[workgroups, 4], vector_chunk[workgroups]."""
    L = lib()
    e = np.ascontiguousarray(elems, np.int32).reshape(-1, _NPE[nElType])
    info = np.zeros(6, np.int32); pl = np.zeros((64, 4), np.int32)
    bind(L.fem_plan_single_cg, [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
    check(L.fem_plan_single_cg(nElType, int(nn), _p(e), len(e), _p(info), _p(pl)))
    pl = pl[:info[1]].copy()
    vec = (pl[:, 1] >> 16) - 1                      # the vector chunk (256 rows) a workgroup owns, or -1
    pl[:, 1] &= 0xffff
    return {"eligible": bool(info[0]), "workgroups": int(info[1]), "chunks_per_workgroup": int(info[2]), "lds": int(info[3]),
            "nchunk": int(info[4]), "nchunk_s": int(info[5]), "plan": pl, "vector_chunk": vec}


def plan(elems_list, nn_list, nElType, uniform_copies=0):
    """fem_plan: the host-side planning of fem_create (uniform_copies > 0, one mesh) / fem_create_batch (0) without any
    device call.  Returns a dict: the info fields + rowptr, lcol, diag, bp, bcol3, rcd[nrcd, 4], rcfirst, chunk_mesh."""
    L = lib()
    npe = _NPE[nElType]
    elems_list = [np.ascontiguousarray(e, np.int32).reshape(-1, npe) for e in elems_list]
    nn = np.ascontiguousarray(nn_list, np.int32); ne = np.array([len(e) for e in elems_list], np.int32)
    elems = np.ascontiguousarray(np.concatenate(elems_list), np.int32) if len(elems_list) else np.zeros((0, npe), np.int32)
    bind(L.fem_plan, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_void_p] * 8)
    info = _PlanInfo()
    args = (nElType, len(nn), _p(nn), _p(ne), _p(elems), int(uniform_copies))
    check(L.fem_plan(*args, C.byref(info), *([None] * 8)))
    out = {n: getattr(info, n) for n, _ in _PlanInfo._fields_}
    nmesh = uniform_copies if uniform_copies else len(nn)
    arr = {"rowptr": np.zeros(info.ndof + 1, np.int32), "lcol": np.zeros(info.nnz, np.int32), "diag": np.zeros(info.ndof, np.int32),
           "bp": np.zeros(info.ndof // 3 + 1, np.int32), "bcol3": np.zeros(info.nnz // 9, np.int32),
           "rcd": np.zeros((max(info.nrcd, 1), 4), np.int32), "rcfirst": np.zeros((1 if uniform_copies else len(nn)) + 1, np.int32),
           "chunk_mesh": np.zeros(max(info.nchunk_tot, 1), np.int32)}
    check(L.fem_plan(*args, C.byref(info), *[_p(arr[k]) for k in ("rowptr", "lcol", "diag", "bp", "bcol3", "rcd", "rcfirst", "chunk_mesh")]))
    arr["rcd"] = arr["rcd"][:info.nrcd]
    out.update(arr)
    out["nmesh"] = nmesh
    return out
