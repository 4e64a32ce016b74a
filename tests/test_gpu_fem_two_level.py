"""GPU parity of the CG's two-level preconditioner (fem_cg_preconditioner(FEM_PRECOND_TWO_LEVEL): Jacobi + the rigid-body modes of
2 x 2 x 2 aggregates, include/fem_hip.h) against oracle_fem_cg_two_level on the exported CSR of the same mesh.  Neither has a
reference counterpart (the reference's only solve is a dead dense inverse, FEA2.cc:1661-1691): the oracle is the definition.
Displacements within 1e-5 relative (north_star), the coarse matrix Z^T K Z within 1e-10 of its largest entry, relative residuals
after a fixed number of iterations within 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.fem import FEA2, FEA2Batch, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_batch, synth_tet_batch_distinct, synth_tet_mesh

RTOL = 1e-5


def _mask(n, fixed):
    m = np.zeros(n, np.uint8); m[fixed] = 1
    return m


def _check(x, rel, ox, orel, what):
    assert np.abs(x - ox).max() <= RTOL * np.abs(ox).max(), what
    assert abs(rel - orel) <= 1e-6 * orel + 1e-12, what


@pytest.mark.parametrize("ncell", [3, 6, 12])
def test_single_mesh_two_level(ncell):
    """One mesh (launch per phase): the coarse matrix, 120 iterations, and the solve to 1e-10 --
    same iteration count as the oracle to within the 25-iteration test cadence, and far fewer than Jacobi needs."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=ncell)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    b = load.copy(); b[fixed] = 0
    xj, itj, relj = fea.solve_cg(b, iters=20000, tol=1e-10)
    fea.cg_preconditioner("two_level")
    rp, col, val = fea.csr(0)
    mask = _mask(len(b), fixed)
    x, done, rel = fea.solve_cg(b, iters=120, tol=0.0)
    Ac = fea.cg_coarse_matrix(0)
    oAc = oracle.fem_coarse_matrix(rp, col, val, nodes, mask)
    assert np.abs(Ac - oAc).max() <= 1e-10 * np.abs(oAc).max()
    ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b, 120, nodes, mask)
    assert done == 120
    _check(x[0], rel[0], ox, orel, "120 iterations")
    x, it, rel = fea.solve_cg(b, iters=20000, tol=1e-10)
    ox, oit, orel = oracle.fem_cg_two_level(rp, col, val, b, 20000, nodes, mask, tol=1e-10)
    assert rel[0] <= 1e-10 and oit <= it < oit + 25 and it < 0.7 * itj, (it, oit, itj)
    assert np.abs(x[0] - ox).max() <= RTOL * np.abs(ox).max()
    assert np.abs(x[0] - xj[0]).max() <= RTOL * np.abs(xj[0]).max()
    # back to Jacobi: bit-identical to the run before the preconditioner was ever switched on
    fea.cg_preconditioner("jacobi")
    xj2, itj2, relj2 = fea.solve_cg(b, iters=20000, tol=1e-10)
    assert itj2 == itj and np.array_equal(xj2, xj) and np.array_equal(relj2, relj)


@pytest.mark.parametrize("nmesh,ncell,iters", [(5, 4, 60), (20, 3, 40), (64, 5, 61), (70, 12, 30), (64, 14, 25), (16, 15, 20)])
def test_uniform_batch_two_level(nmesh, ncell, iters):
    """Batches of one topology with their own coordinates (so their own aggregates' centroids and coarse matrices): below and above
    the fused-step threshold (k_fem_cg_step<true>: ncell = 14 in its registers-only form, ncell = 15 -- 12,288 dofs -- in the general
    one), and the sizes that run resident on the compute units (k_fem_cg_resident<false, true>)."""
    nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=ncell)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
    x, done, rel = fea.solve_cg(b, iters=iters, tol=0.0)
    assert done == iters
    mask = _mask(b.shape[1], fixed)
    for m in sorted({0, nmesh // 2, nmesh - 1}):
        rp, col, val = fea.csr(m)
        oAc = oracle.fem_coarse_matrix(rp, col, val, nodes[m], mask)
        assert np.abs(fea.cg_coarse_matrix(m) - oAc).max() <= 1e-10 * np.abs(oAc).max()
        ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[m], iters, nodes[m], mask)
        _check(x[m], rel[m], ox, orel, m)
    assert not np.array_equal(x[0], x[1])
    # the same iterations in three calls: bit-identical
    fea.cg_setup(b)
    fea.cg_iterate(1); fea.cg_iterate(iters - 8); fea.cg_iterate(7)
    x2, rel2 = fea.cg_result()
    assert np.array_equal(x2, x) and np.array_equal(rel2, rel)


@pytest.mark.parametrize("n,base", [(7, 4), (72, 5), (64, 12)])
def test_segmented_batch_two_level(n, base):
    """Meshes of their own sizes and topologies concatenated (fem_create_batch): aggregates, coarse matrices and the coarse dot
    products are per mesh."""
    nodes_l, tets_l, fixed_l, load_l = synth_tet_batch_distinct(n, base=base)
    fb = FEA2Batch(nodes_l, tets_l, FEM_TET4)
    fb.MatrixAssembly()
    fixed = np.concatenate([fb.dof0[k] + fx for k, fx in enumerate(fixed_l)]).astype(np.int32)
    fb.eliminate_dofs(fixed)
    fb.cg_preconditioner("two_level")
    b = np.concatenate(load_l); b[fixed] = 0
    x, done, rel = fb.solve_cg(b, iters=50, tol=0.0)
    assert done == 50 and len(rel) == n
    for k in range(0, n, 1 if n < 10 else 9):
        rp, col, val = fb.csr(k)
        d0, d1 = fb.dof0[k], fb.dof0[k + 1]
        mask = _mask(d1 - d0, fixed_l[k])
        oAc = oracle.fem_coarse_matrix(rp, col, val, nodes_l[k], mask)
        assert np.abs(fb.cg_coarse_matrix(k) - oAc).max() <= 1e-10 * np.abs(oAc).max()
        ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[d0:d1], 50, nodes_l[k], mask)
        _check(x[0, d0:d1], rel[k], ox, orel, k)


def test_two_level_with_penalty_dirichlet():
    """The reference's own boundary condition (ImposeDirichletEncastre_K, FEA2.cc:1628-1643: Klarge on the diagonal of node id - 1):
    the dofs it names leave the coarse space exactly as eliminated dofs do.  The penalised matrix is badly scaled (1e8 beside
    ~1e3): what is compared is the coarse matrix and a fixed number of iterations."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=5)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    ids = (np.unique(np.asarray(fixed) // 3) + 1).astype(np.int32)      # 1-based ids, as the reference passes them
    fea.ImposeDirichletEncastre_K(ids)
    fea.cg_preconditioner("two_level")
    b = load.copy()
    x, done, rel = fea.solve_cg(b, iters=80, tol=0.0)
    rp, col, val = fea.csr(0)
    mask = np.zeros(len(b), np.uint8)
    for k in range(3):
        mask[3 * (ids - 1) + k] = 1
    oAc = oracle.fem_coarse_matrix(rp, col, val, nodes, mask)
    assert np.abs(fea.cg_coarse_matrix(0) - oAc).max() <= 1e-10 * np.abs(oAc).max()
    ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b, 80, nodes, mask)
    _check(x[0], rel[0], ox, orel, "penalty")
    # a new assembly forgets the constrained dofs: the coarse space is rebuilt without them
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed[: len(fixed) // 2])
    x, done, rel = fea.solve_cg(b, iters=10, tol=0.0)
    rp, col, val = fea.csr(0)
    oAc = oracle.fem_coarse_matrix(rp, col, val, nodes, _mask(len(b), fixed[: len(fixed) // 2]))
    assert np.abs(fea.cg_coarse_matrix(0) - oAc).max() <= 1e-10 * np.abs(oAc).max()


def test_two_level_lopsided_aggregates_leave_the_resident_kernel():
    """A batch whose meshes would run resident (64 meshes, 6,591 dofs) but whose nodes crowd into one half of the bounding box: an
    aggregate of more than 320 nodes does not fit the resident kernel's registers, the solve goes phase by phase -- same numbers."""
    nmesh = 64
    nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=12)
    nodes = np.array(nodes, np.float32, copy=True)
    nodes[..., 0] = np.float32(10.0) * (nodes[..., 0] / nodes[..., 0].max()) ** 4     # x crowded towards 0: five of six nodes below the midpoint
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
    fea.profile(True)
    x, done, rel = fea.solve_cg(b, iters=30, tol=0.0)
    prof = fea.profile_read()
    assert prof["k_fem_spmv"][1] >= 30 and not prof.get("k_fem_cg_resident", (0, 0))[1]
    mask = _mask(b.shape[1], fixed)
    agg, _ = oracle.fem_coarse_space(nodes[0])
    assert np.bincount(agg, minlength=8).max() > 320
    for m in (0, nmesh - 1):
        rp, col, val = fea.csr(m)
        ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[m], 30, nodes[m], mask)
        _check(x[m], rel[m], ox, orel, m)


def test_two_level_resident_kernel_is_the_one_that_runs():
    """Config 3's batch under the two-level preconditioner: one launch of k_fem_cg_resident per call, no product kernel after the
    48 of the set-up."""
    nmesh = 64
    nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=12)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
    fea.cg_setup(b)
    fea.profile(True)
    fea.cg_iterate(40)
    x, rel = fea.cg_result()
    prof = fea.profile_read()
    assert prof["k_fem_cg_resident"][1] == 1 and not prof["k_fem_spmv"][1]
    rp, col, val = fea.csr(5)
    ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[5], 40, nodes[5], _mask(b.shape[1], fixed))
    _check(x[5], rel[5], ox, orel, 5)


def test_two_level_one_large_mesh_takes_the_three_kernel_coarse_path():
    """Meshes of more than 65,536 nodes restrict, solve and prolong in three launches (k_fem_cz_restrict with a workgroup per
    aggregate, k_fem_cz_solve, k_fem_cz_prolong) instead of one workgroup per mesh: 74,088 nodes, 12 iterations."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=41)
    assert len(nodes) > 65536
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = load.copy(); b[fixed] = 0
    x, done, rel = fea.solve_cg(b, iters=12, tol=0.0)
    rp, col, val = fea.csr(0)
    mask = _mask(len(b), fixed)
    oAc = oracle.fem_coarse_matrix(rp, col, val, nodes, mask)
    assert np.abs(fea.cg_coarse_matrix(0) - oAc).max() <= 1e-10 * np.abs(oAc).max()
    ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b, 12, nodes, mask)
    _check(x[0], rel[0], ox, orel, "large mesh")


@pytest.mark.parametrize("nmesh", [1, 20, 64])
def test_two_level_freezes_a_mesh_without_load_and_stays_finite_past_convergence(nmesh):
    """As test_gpu_fem.py's guards, under the two-level preconditioner: a mesh with a zero right-hand side (w = Z^T r = 0, r.z = 0)
    is frozen -- x == 0, no NaN -- in all three forms (launch per phase, fused step, resident), and 400 iterations on an 81-dof mesh
    (whose aggregates hold 1-4 nodes each: most rotations are dropped as dependent) stay finite and at the solution."""
    nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=3)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
    dead = nmesh // 3
    b[dead] = 0
    x, done, rel = fea.solve_cg(b, iters=30, tol=0.0)
    assert np.isfinite(x).all() and np.isfinite(rel).all()
    assert not x[dead].any() and rel[dead] == 0
    mask = _mask(b.shape[1], fixed)
    for m in sorted({0, nmesh - 1} - {dead}):
        rp, col, val = fea.csr(m)
        ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[m], 30, nodes[m], mask)
        _check(x[m], rel[m], ox, orel, m)
    nodes, tets, fixed, load = synth_tet_batch(nmesh, ncell=2)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    fea.cg_preconditioner("two_level")
    b = np.tile(load, (nmesh, 1)); b[:, fixed] = 0
    x, done, rel = fea.solve_cg(b, iters=400, tol=0.0)
    assert np.isfinite(x).all() and np.isfinite(rel).all() and rel.max() < 1e-9
    rp, col, val = fea.csr(0)
    r = b[0] - oracle.fem_csr_matvec(rp, col, val, x[0])
    assert np.linalg.norm(r) <= 1e-9 * np.linalg.norm(b[0])


def test_two_level_api_errors_are_loud():
    """Argument errors of the two entry points come back as OrbxError, not as silence: an unknown preconditioner, the coarse matrix
    before any set-up, under Jacobi, and for a mesh that does not exist."""
    from orb_slam2_e_amd._lib import OrbxError
    nodes, tets, fixed, load = synth_tet_mesh(ncell=3)
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly()
    fea.eliminate_dofs(fixed)
    with pytest.raises(OrbxError):
        fea.cg_preconditioner(7)
    with pytest.raises(OrbxError):
        fea.cg_coarse_matrix(0)                       # Jacobi: there is none
    fea.cg_preconditioner("two_level")
    with pytest.raises(OrbxError):
        fea.cg_coarse_matrix(0)                       # no fem_cg_setup yet
    b = load.copy(); b[fixed] = 0
    fea.solve_cg(b, iters=5)
    assert fea.cg_coarse_matrix(0).shape == (48, 48)
    with pytest.raises(OrbxError):
        fea.cg_coarse_matrix(1)


@pytest.mark.parametrize("nm", [1, 3, 70])
def test_nodes_that_belong_to_no_element_stay_at_zero(nm):
    """800 of the reference's 853 surface meshes have points that no triangle uses: K has a zero row and column for each of their dofs.
    Such a dof has no diagonal to precondition with (1 / 0: every vector turned NaN, on the device and in the oracle alike) -- it stays
    at 0 and counts as constrained in the coarse space; the rest of the system is solved as if those nodes were not there.  One mesh
    (the one-launch kernel and the launch-per-phase path) and batches, both preconditioners, against the oracle and against the same
    meshes without the extra nodes."""
    import os
    from orb_slam2_e_amd.fem import FEA2, FEM_TET4
    from orb_slam2_e_amd.synth import synth_tet_mesh
    nodes, tets, fixed, load = synth_tet_mesh(5)
    rng = np.random.default_rng(4)
    extra = np.array([[9.0, 9.0, 9.0], [-3.0, 1.0, 2.0], [0.5, 0.5, 7.0]], np.float32)
    many = np.stack([np.vstack([nodes + rng.normal(0, 0.01, nodes.shape).astype(np.float32), extra]) for _ in range(nm)])
    b = np.tile(np.concatenate([load, np.zeros(9)]), (nm, 1)); b[:, fixed] = 0
    for pre in ("jacobi", "two_level"):
        for xcd in (("1", "0") if nm == 1 else ("1",)):
            os.environ["FEM_CG_XCD"] = xcd
            try:
                fea = FEA2(many, tets, FEM_TET4); fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
                fea.cg_preconditioner(pre)
                x, done, rel = fea.solve_cg(b, iters=60, tol=0.0)
            finally:
                os.environ.pop("FEM_CG_XCD", None)
            assert np.isfinite(x).all() and np.isfinite(rel).all() and np.all(x[:, len(load):] == 0)
            ref = FEA2(many[:, :len(nodes)], tets, FEM_TET4); ref.MatrixAssembly(); ref.eliminate_dofs(fixed)
            ref.cg_preconditioner(pre)
            x0, _, rel0 = ref.solve_cg(b[:, :len(load)], iters=60, tol=0.0)
            if pre == "jacobi":      # (the two-level form's aggregates are cut by the bounding box, which the extra nodes move)
                assert np.abs(x[:, :len(load)] - x0).max() <= 1e-12 * np.abs(x0).max()
            for k in sorted({0, nm - 1}):
                rp, col, val = fea.csr(k)
                mk = np.zeros(b.shape[1], np.uint8); mk[fixed] = 1
                ox, _, orel = (oracle.fem_cg(rp, col, val, b[k], 60, 0.0) if pre == "jacobi" else oracle.fem_cg_two_level(rp, col, val, b[k], 60, many[k], mk))
                assert np.abs(x[k] - ox).max() <= 1e-5 * np.abs(ox).max() and abs(rel[k] - orel) <= 1e-5 * orel
