"""GPU parity: the one-mesh CG as ONE launch over the compute units of one XCD (k_fem_cg_xcd: <= 32 resident workgroups, two
hand-rolled barriers per iteration, the matrix in registers, p replicated in LDS) against the launch-per-phase path it replaces
(k_fem_spmv + k_fem_cg_update + k_fem_cg_dir, selected with FEM_CG_XCD=0) -- BIT FOR BIT: the kernel keeps that path's chunk
decomposition and summation orders -- and against the oracle's CG at 1e-5 (north_star)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.fem import FEA2, FEM_TET4
from orb_slam2_e_amd.synth import synth_tet_chain, synth_tet_mesh

RTOL = 1e-5


def _model(nodes, tets, fixed, load):
    fea = FEA2(nodes, tets, FEM_TET4)
    fea.MatrixAssembly(); fea.eliminate_dofs(fixed)
    b = load.copy(); b[fixed] = 0
    return fea, b[None]


def _run(fea, b, slices, xcd):
    os.environ["FEM_CG_XCD"] = {True: "1", False: "0"}.get(xcd, xcd)
    try:
        fea.profile(True)
        fea.cg_setup(b)
        for n in slices:
            fea.cg_iterate(n)
        x, rel = fea.cg_result()
        return x, rel, fea.profile_read()
    finally:
        os.environ.pop("FEM_CG_XCD", None)


@pytest.mark.parametrize("ncell,iters", [(12, 200), (2, 40), (5, 120), (9, 150), (11, 60)])
def test_one_launch_equals_the_launch_per_phase_path_bit_for_bit(ncell, iters):
    nodes, tets, fixed, load = synth_tet_mesh(ncell=ncell)
    fea, b = _model(nodes, tets, fixed, load)
    xa, ra, pa = _run(fea, b, [iters], xcd=True)
    xb, rb, pb = _run(fea, b, [iters], xcd=False)
    assert (pa["k_fem_cg_xcd"][1] == 1 and not pa["k_fem_spmv"][1]) or fea.one_launch_stats()[1]   # ONE launch did all the iterations (on an idle chip)
    assert pb["k_fem_spmv"][1] == iters and not pb.get("k_fem_cg_xcd", (0, 0))[1]
    assert np.isfinite(xa).all() and xa.tobytes() == xb.tobytes() and ra.tobytes() == rb.tobytes()
    rp, col, val = fea.csr(0)
    ox, _, orel = oracle.fem_cg(rp, col, val, b[0], iters, 0.0)
    assert np.abs(xa[0] - ox).max() <= RTOL * np.abs(ox).max()
    assert abs(ra[0] - orel) <= 1e-6 * orel + 1e-12


def test_split_launches_and_mixed_paths_continue_the_same_iteration():
    """fem_cg_iterate in slices -- odd lengths, one path handing over to the other in the middle -- is the same sequence of
    iterations: x after 7 + 1 + 30 + 12 equals x after 50, bit for bit, whichever path ran which slice."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=12)
    fea, b = _model(nodes, tets, fixed, load)
    x50, r50, _ = _run(fea, b, [50], xcd=False)
    xs, rs, _ = _run(fea, b, [7, 1, 30, 12], xcd=True)
    assert xs.tobytes() == x50.tobytes() and rs.tobytes() == r50.tobytes()
    fea.cg_setup(b)
    for n, xcd in ((7, True), (1, False), (30, True), (3, False), (9, True)):
        os.environ["FEM_CG_XCD"] = "1" if xcd else "0"
        fea.cg_iterate(n)
    os.environ.pop("FEM_CG_XCD", None)
    xm, rm = fea.cg_result()
    assert xm.tobytes() == x50.tobytes() and rm.tobytes() == r50.tobytes()


@pytest.mark.parametrize("nn", [27, 2197, 2596, 2730, 2731, 3000])
def test_sizes_around_the_kernels_limits(nn):
    """Chains of tetrahedra of nn nodes: few chunks (fewer participants than 32), 7,788 dofs (the reference's largest mesh), the last
    size the kernel takes (8,190 dofs: 32 vector chunks) and the first it does not (8,193: the launch-per-phase path runs by itself)."""
    nodes, tets, fixed, load = synth_tet_chain(nn)
    fea, b = _model(nodes, tets, fixed, load)
    xa, ra, pa = _run(fea, b, [30], xcd=True)
    xb, rb, pb = _run(fea, b, [30], xcd=False)
    took = bool(pa.get("k_fem_cg_xcd", (0, 0))[1])
    assert took == (3 * nn <= 8192)
    assert xa.tobytes() == xb.tobytes() and ra.tobytes() == rb.tobytes()
    rp, col, val = fea.csr(0)
    ox, _, _ = oracle.fem_cg(rp, col, val, b[0], 30, 0.0)
    assert np.abs(xa[0] - ox).max() <= RTOL * np.abs(ox).max()


def test_solve_to_tolerance_under_both_preconditioners():
    nodes, tets, fixed, load = synth_tet_mesh(ncell=12)
    fea, b = _model(nodes, tets, fixed, load)
    fea.profile(True)
    x, done, rel = fea.solve_cg(b, iters=3000, tol=1e-8)          # slices of 25 between convergence tests: one launch each
    prof = fea.profile_read()
    quiet = fea.one_launch_stats()[1] == 0        # (beside another process's load a launch may give up and be made good: the counts below are an idle chip's)
    assert rel[0] <= 1e-8 and done % 25 == 0 and (prof["k_fem_cg_xcd"][1] == done // 25 or not quiet)
    os.environ["FEM_CG_XCD"] = "0"
    try:
        x0, done0, rel0 = fea.solve_cg(b, iters=3000, tol=1e-8)
    finally:
        os.environ.pop("FEM_CG_XCD", None)
    assert done0 == done and x0.tobytes() == x.tobytes()
    fea.cg_preconditioner("two_level")                             # the coarse correction runs inside the one-launch kernel too
    fea.profile(True)
    x2, done2, rel2 = fea.solve_cg(b, iters=3000, tol=1e-8)
    prof = fea.profile_read()
    quiet = fea.one_launch_stats()[1] == 0
    assert rel2[0] <= 1e-8 and done2 < done // 2 and ((prof["k_fem_cg_xcd"][1] == done2 // 25 and not prof["k_fem_spmv"][1]) or not quiet)
    assert np.abs(x2 - x).max() <= 1e-6 * np.abs(x).max()
    fea.cg_preconditioner("jacobi")


def test_many_models_solved_at_once_from_many_threads():
    """Twelve threads, a model each, every one solving its own mesh over and over: more one-launch kernels than the chip can hold
    complete sets of waiting workgroups for.  The library admits six at a time and sends the rest down the launch-per-phase path
    (same bits), so nothing waits for a workgroup that cannot be scheduled: no barrier times out, every result is the
    single-threaded one."""
    import threading
    models = []
    for k in range(12):
        nodes, tets, fixed, load = synth_tet_mesh(ncell=6 + k % 4, seed=20 + k)
        fea, b = _model(nodes, tets, fixed, load)
        models.append((fea, b))
    os.environ["FEM_CG_XCD"] = "0"
    want = []
    for fea, b in models:
        fea.cg_setup(b); fea.cg_iterate(80)
        want.append(fea.cg_result()[0].tobytes())
    os.environ.pop("FEM_CG_XCD", None)
    errors = []

    def work(k):
        fea, b = models[k]
        try:
            for rep in range(15):
                fea.cg_setup(b); fea.cg_iterate(30); fea.cg_iterate(50)
                if fea.cg_result()[0].tobytes() != want[k]:
                    errors.append((k, rep, "differs")); return
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(12)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors[:3]


def test_system_scope_granules_give_the_same_bits():
    """FEM_CG_XCD=safe: the kernel's fallback for participants that do NOT share an XCD (system-scope granules on both sides, served
    by memory) -- never taken by itself so far, so forced here: same bits, more time per iteration."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=12)
    fea, b = _model(nodes, tets, fixed, load)
    xs, rs, ps = _run(fea, b, [40, 25], xcd="safe")
    xb, rb, pb = _run(fea, b, [65], xcd=False)
    assert ps["k_fem_cg_xcd"][1] == 2 and xs.tobytes() == xb.tobytes() and rs.tobytes() == rb.tobytes()


@pytest.mark.parametrize("ncell,iters", [(12, 120), (5, 60), (3, 40), (9, 77)])
def test_two_level_inside_the_launch_equals_the_launch_per_phase_path(ncell, iters):
    """fem_cg_preconditioner(two_level) on one mesh: the coarse correction inside k_fem_cg_xcd (every workgroup stages all of r and
    walks cz_apply_block's sixteen waves with its four) against the launch-per-phase path with k_fem_cz_apply between the update and
    the direction kernel -- bit for bit, slices and mixed paths included -- and against the oracle's two-level CG at 1e-5."""
    nodes, tets, fixed, load = synth_tet_mesh(ncell=ncell)
    fea, b = _model(nodes, tets, fixed, load)
    fea.cg_preconditioner("two_level")
    xa, ra, pa = _run(fea, b, [iters], xcd=True)
    xb, rb, pb = _run(fea, b, [iters], xcd=False)
    assert ((pa["k_fem_cg_xcd"][1] == 1 and not pa["k_fem_spmv"][1]) or fea.one_launch_stats()[1]) and pb["k_fem_spmv"][1] == iters
    assert np.isfinite(xa).all() and xa.tobytes() == xb.tobytes() and ra.tobytes() == rb.tobytes()
    xs, rs, _ = _run(fea, b, [7, iters - 20, 13], xcd=True)
    assert xs.tobytes() == xb.tobytes() and rs.tobytes() == rb.tobytes()
    rp, col, val = fea.csr(0)
    mk = np.zeros(fea.Ksize, np.uint8); mk[fixed] = 1
    ox, _, orel = oracle.fem_cg_two_level(rp, col, val, b[0], iters, nodes, mk)
    assert np.abs(xa[0] - ox).max() <= RTOL * np.abs(ox).max()
    fea.cg_preconditioner("jacobi")


def test_one_launch_cg_beside_a_saturating_extraction_load():
    """The kernel's workgroups wait for each other, so all of them must get a seat: here three threads keep the chip full with 64-frame
    extraction batches while a fourth solves the 6,591-dof mesh over and over under both preconditioners.  A participant that is not
    scheduled within 2 ms makes its launch give up (seen: the small workgroups of the other streams keep taking the seats); the host
    then runs the owed iterations on the launch-per-phase path and stays there for a while.  Either way no call fails and every result
    is the launch-per-phase path's, bit for bit -- also when one-launch and launch-per-phase slices mix inside one solve."""
    import threading
    from orb_slam2_e_amd import ORBextractor
    from orb_slam2_e_amd.synth import synth_sequence
    nodes, tets, fixed, load = synth_tet_mesh(ncell=12)
    fea, b = _model(nodes, tets, fixed, load)
    want = {}
    for pre in ("jacobi", "two_level"):
        fea.cg_preconditioner(pre)
        os.environ["FEM_CG_XCD"] = "0"
        fea.cg_setup(b); fea.cg_iterate(120)
        want[pre] = fea.cg_result()[0].tobytes()
        os.environ.pop("FEM_CG_XCD", None)
    frames = synth_sequence(64)
    stop = threading.Event()
    errors, batches = [], [0, 0, 0]

    def load_thread(k):
        try:
            ex = ORBextractor(2000, 1.2, 8, 20, 7)
            while not stop.is_set():
                ex.extract_batch(frames); ex.download_batch(); batches[k] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("load", k, repr(e)))

    ts = [threading.Thread(target=load_thread, args=(k,)) for k in range(3)]
    for t in ts: t.start()
    solves = 0
    try:
        import time
        t0 = time.time()
        while time.time() - t0 < 6.0 and not errors:
            for pre in ("jacobi", "two_level"):
                fea.cg_preconditioner(pre)
                fea.cg_setup(b); fea.cg_iterate(49); fea.cg_iterate(71)       # (an odd slice: the next one starts from the odd rz slot)
                got = fea.cg_result()[0].tobytes()
                if got != want[pre]: errors.append(("cg", pre, "differs"))
                solves += 1
    finally:
        stop.set()
        for t in ts: t.join()
        fea.cg_preconditioner("jacobi")
    launches, recovered = fea.one_launch_stats()
    print("solves", solves, "one-launch kernel launches", launches, "of which gave up and were made good", recovered, "extraction batches", batches)
    assert not errors, errors[:3]
    assert solves >= 6 and min(batches) >= 5 and launches >= 1, (solves, batches, launches)
