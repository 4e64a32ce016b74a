"""The host half of fem_create / fem_create_batch (symbolic CSR of 3 x 3 node blocks, chunk tables, the resident CG's
chunk table) through fem_plan -- no device call, so it runs without a GPU (and, rebuilt host-only under ASan / UBSan by
tools/asan_host.sh, is what the sanitizers see).  The pattern is checked against a numpy construction from the element
list (MatrixAssemblyC3D8/6's scatter, FEA2.cc:1379-1624: entry (3 I + r, 3 J + c) exists iff nodes I, J share an element)."""
import os

import numpy as np
import pytest

from orb_slam2_e_amd.fem import FEM_C3D6, FEM_C3D8, FEM_TET4, extrude_elems, plan
from orb_slam2_e_amd.synth import synth_tet_batch_distinct, synth_tet_mesh

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pattern(elems, nn):
    """rowptr, lcol of the block pattern: node I's neighbours = itself + all nodes sharing an element, ascending."""
    nbr = [set([i]) for i in range(nn)]
    for e in elems:
        for a in e:
            nbr[a].update(int(b) for b in e)
    rowptr, lcol = [0], []
    for i in range(nn):
        cols = [3 * j + c for j in sorted(nbr[i]) for c in range(3)]
        for r in range(3):
            lcol += cols
            rowptr.append(len(lcol))
    return np.array(rowptr, np.int32), np.array(lcol, np.int32)


def _check_tables(p, rowptr, lcol, mesh_rows):
    assert p["ndof"] == len(rowptr) - 1 and p["nnz"] == len(lcol) == 9 * p["nblk"]
    assert np.array_equal(p["rowptr"], rowptr) and np.array_equal(p["lcol"], lcol)
    rows = np.arange(p["ndof"])
    assert np.array_equal(p["lcol"][p["diag"]], rows)                       # every row's diagonal position
    assert np.all((p["diag"] >= rowptr[:-1]) & (p["diag"] < rowptr[1:]))
    # node-block tables: bp = blocks before a block row, bcol3 = a block's first column
    nb = np.diff(rowptr)[0::3] // 3
    assert np.array_equal(p["bp"], np.concatenate([[0], np.cumsum(nb)]))
    first = np.concatenate([lcol[rowptr[3 * i]:rowptr[3 * i + 1]:3] for i in range(p["ndof"] // 3)])
    assert np.array_equal(p["bcol3"], first)
    # chunks of the resident CG: every block row of every mesh exactly once, in order, at most 256 blocks per chunk
    if p["resident"]:
        rcd, rcf = p["rcd"], p["rcfirst"]
        assert rcf[0] == 0 and rcf[-1] == len(rcd) and len(rcf) == len(mesh_rows) + 1
        for k, (row0, nrows) in enumerate(mesh_rows):
            c = rcd[rcf[k]:rcf[k + 1]]
            assert c[0, 0] == row0 // 3 and c[-1, 0] + c[-1, 1] == (row0 + nrows) // 3
            assert np.array_equal(c[1:, 0], c[:-1, 0] + c[:-1, 1])
            assert np.array_equal(c[:, 2], p["bp"][c[:, 0]]) and np.array_equal(c[:, 2] + c[:, 3], p["bp"][c[:, 0] + c[:, 1]])
            assert c[:, 3].max() <= 256 and c[:, 1].min() >= 1 and c[:, 1].max() <= (21 if p["resident_big"] else 63)
    else:
        assert p["nrcd"] == 0


@pytest.mark.parametrize("name,eltype", [("min", FEM_C3D6), ("median", FEM_C3D6), ("p90", FEM_C3D6), ("median", FEM_C3D8)])
def test_plan_of_the_references_own_meshes(name, eltype):
    m = np.load(os.path.join(GOLD, f"fem_mesh_{name}.npz"))
    top, tris = m["points"], m["triangles"]
    if eltype == FEM_C3D8:
        faces = np.array([[a, b, c, a] for a, b, c in tris], np.int32)      # degenerate quads: repeated node ids in an element
    else:
        faces = tris
    elems = extrude_elems(faces, len(top))
    nn = 2 * len(top)
    p = plan([elems], [nn], eltype, uniform_copies=1)
    rowptr, lcol = _pattern(elems, nn)
    _check_tables(p, rowptr, lcol, [(0, 3 * nn)])
    assert not p["resident"] and p["nchunk_tot"] == (3 * nn + 255) // 256
    assert p["ncontrib"] == len(elems) * elems.shape[1] ** 2 and p["fused_lds"] > 0
    # the shared-row assembly: the entry-by-entry kernel's LDS + a flag per Gauss point + nine floats per contribution of the fullest row
    assert p["fused_lds"] < p["rows_lds"] <= 64 * 1024
    p64 = plan([elems], [nn], eltype, uniform_copies=64)                       # 64 copies: one compute unit each
    assert p64["resident"] and not p64["resident_big"] and p64["nchunk_tot"] == 64 * p["nchunk_tot"]
    _check_tables(p64, rowptr, lcol, [(0, 3 * nn)])


def test_plan_of_config3_and_the_lds_limits():
    nodes, tets, fixed, load = synth_tet_mesh(ncell=12)
    p = plan([tets], [len(nodes)], FEM_TET4, uniform_copies=256)
    assert (p["ndof"], p["nnz"]) == (6591, 261477) and p["resident"] and not p["resident_big"]
    rowptr, lcol = _pattern(tets, len(nodes))
    _check_tables(p, rowptr, lcol, [(0, 6591)])
    # (2 n + 3 * 8 * 256 + 48) * 8 <= 160 KB with p and Ap in LDS, n + ... with p alone: the documented limits
    for ncell, resident, big in ((13, True, True), (15, True, True), (16, False, True)):
        nd, tt, _, _ = synth_tet_mesh(ncell=ncell)
        q = plan([tt], [len(nd)], FEM_TET4, uniform_copies=64)
        assert (bool(q["resident"]), bool(q["resident_big"])) == (resident, big), (ncell, q["ndof"])
        assert not resident or q["resident_lds"] <= 160 * 1024
    assert not plan([tets], [len(nodes)], FEM_TET4, uniform_copies=63)["resident"]      # fewer than 64 meshes: launch per phase


@pytest.mark.parametrize("nn,resident", [(4762, True), (4763, False)])
def test_resident_limit_is_the_lds_formula(nn, resident):
    """include/fem_hip.h: at most 14,288 dofs per mesh on a compute unit, (n + 6,192) x 8 B <= 160 KB -- 14,286 dofs
    (4,762 nodes) run resident with p alone in LDS, 14,289 dofs take the launch-per-phase path."""
    from orb_slam2_e_amd.synth import synth_tet_chain
    nodes, tets, fixed, load = synth_tet_chain(nn)
    q = plan([tets], [nn], FEM_TET4, uniform_copies=64)
    assert bool(q["resident"]) == resident and q["ndof"] == 3 * nn
    if resident:
        assert q["resident_big"] and q["resident_lds"] == (3 * nn + 3 * 8 * 256 + 48) * 8 <= 160 * 1024


def test_plan_of_a_batch_with_its_own_topologies():
    nodes_l, tets_l, _, _ = synth_tet_batch_distinct(70, base=4)
    nn = [len(n) for n in nodes_l]
    p = plan(tets_l, nn, FEM_TET4)
    rps, lcs, rows, off = [np.zeros(1, np.int32)], [], [], 0
    for t, n in zip(tets_l, nn):
        rp, lc = _pattern(t, n)
        rps.append(rp[1:] + rps[-1][-1]); lcs.append(lc + 3 * off); rows.append((3 * off, 3 * n)); off += n
    _check_tables(p, np.concatenate(rps), np.concatenate(lcs), rows)
    assert p["resident"]
    # a chunk of the vector kernels never crosses a mesh
    cm = p["chunk_mesh"]
    assert len(cm) == p["nchunk_tot"] == sum((3 * n + 255) // 256 for n in nn)
    assert np.array_equal(cm, np.repeat(np.arange(70), [(3 * n + 255) // 256 for n in nn]))


def test_plan_rejects_what_create_rejects():
    from orb_slam2_e_amd import OrbxError
    with pytest.raises(OrbxError):
        plan([np.array([[0, 1, 2, 9]], np.int32)], [4], FEM_TET4, uniform_copies=1)        # node id out of range
    with pytest.raises(OrbxError):
        plan([np.zeros((0, 4), np.int32)], [1], FEM_TET4, uniform_copies=1)                 # Ksize <= 3
    with pytest.raises(OrbxError):
        plan([np.array([[0, 1, 2, 3]], np.int32)] * 2, [4, 4], FEM_TET4, uniform_copies=2)  # copies only of ONE mesh
    p = plan([np.zeros((0, 4), np.int32)], [5], FEM_TET4, uniform_copies=1)                 # no elements: the identity pattern
    assert p["nnz"] == 45 and np.array_equal(p["lcol"].reshape(5, 3, 3)[:, 0, :], np.arange(15).reshape(5, 3))


def test_symbolic_phase_two_formulations_agree():
    """fem_plan_selfcheck: the linear-pass symbolic phase fem_create uses == the first, list-based formulation, every table
    (block pattern, contribution lists in the reference's scatter order, per-node element lists) -- on the reference's
    meshes as prisms and degenerate hexahedra (repeated node ids), grids of tets, random element soups, an empty mesh."""
    import ctypes as C
    from orb_slam2_e_amd._lib import check, lib
    L = lib()
    L.fem_plan_selfcheck.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
    def run(eltype, nn, elems):
        elems = np.ascontiguousarray(elems, np.int32)
        check(L.fem_plan_selfcheck(eltype, nn, elems.ctypes.data_as(C.c_void_p), len(elems)))
    for name in ("min", "median", "p90", "large"):
        m = np.load(os.path.join(GOLD, f"fem_mesh_{name}.npz"))
        top, tris = m["points"], m["triangles"]
        run(FEM_C3D6, 2 * len(top), extrude_elems(tris, len(top)))
        run(FEM_C3D8, 2 * len(top), extrude_elems(np.array([[a, b, c, a] for a, b, c in tris], np.int32), len(top)))
    nodes, tets, _, _ = synth_tet_mesh(ncell=12)
    run(FEM_TET4, len(nodes), tets)
    rng = np.random.default_rng(0)
    for npe, et in ((4, FEM_TET4), (6, FEM_C3D6), (8, FEM_C3D8)):
        run(et, 50, rng.integers(0, 50, (300, npe)))          # soup: repeated ids inside elements, isolated nodes
    run(FEM_TET4, 7, np.zeros((0, 4), np.int32))


def test_single_mesh_cg_plan_covers_every_chunk_and_every_column():
    """fem_plan_single_cg (host only): the plan of the one-launch CG kernel.  Every SpMV chunk belongs to exactly one workgroup, in
    contiguous runs; every workgroup's column range holds its own rows, its vector chunk's rows and every column its blocks read
    (checked against the CSR pattern of fem_plan), is even-aligned, and fits the LDS the plan asks for; meshes beyond 8,192 dofs
    are not eligible; the kernel variant (1 / 3 / 6 chunks per workgroup) is the smallest that fits."""
    from orb_slam2_e_amd.fem import FEM_TET4, plan, plan_single_cg
    from orb_slam2_e_amd.synth import synth_tet_chain, synth_tet_mesh
    for ncell, nn_chain in ((12, None), (2, None), (7, None), (None, 2596), (None, 2730), (None, 2731), (None, 27)):
        nodes, tets, fixed, load = synth_tet_mesh(ncell) if ncell else synth_tet_chain(nn_chain)
        nn = len(nodes)
        sp = plan_single_cg(tets, nn, FEM_TET4)
        assert sp["eligible"] == (3 * nn <= 8192), (nn, sp)
        if not sp["eligible"]:
            continue
        full = plan([tets], [nn], FEM_TET4, uniform_copies=1)
        spb, ndof = full["spb"], 3 * nn
        assert sp["nchunk"] == (ndof + 255) // 256 and sp["nchunk_s"] == (ndof + spb - 1) // spb
        P, pl = sp["workgroups"], sp["plan"]
        assert 1 <= P <= 64 and P >= sp["nchunk"] and (P <= 32 or sp["chunks_per_workgroup"] <= 3)
        kch = max(int(c1 - c0) for c0, c1, _, _ in pl)
        assert sp["chunks_per_workgroup"] == (1 if kch <= 1 else 3 if kch <= 3 else 6)
        assert sorted(int(v) for v in sp["vector_chunk"] if v >= 0) == list(range(sp["nchunk"]))   # every vector chunk once
        owned = np.concatenate([np.arange(c0, c1) for c0, c1, _, _ in pl])
        assert np.array_equal(np.sort(owned), np.arange(sp["nchunk_s"]))              # every chunk once (runs need not follow the ranks)
        if P > 32:      # two per compute unit: the second-comers (rank >= 32) hold the smaller runs and, with an older neighbour free, no vector chunk
            n = np.array([int(c1 - c0) for c0, c1, _, _ in pl])
            assert n[:32].min() == n[:32].max() == kch and n[32:].max() <= max(2, kch - (kch == 3 and sp["nchunk_s"] <= 160))
            assert (np.asarray(sp["vector_chunk"])[32:] >= 0).sum() <= max(0, sp["nchunk"] - 26)
        rowptr, lcol = full["rowptr"], full["lcol"]
        for w, (c0, c1, lo, hi) in enumerate(pl):
            assert lo % 2 == 0 and hi % 2 == 0 and 0 <= lo < hi <= ndof + 1
            need = []
            if c1 > c0:
                r0, r1 = c0 * spb, min(c1 * spb, ndof)
                need += [r0, r1 - 1, int(lcol[rowptr[r0]:rowptr[r1]].min()), int(lcol[rowptr[r0]:rowptr[r1]].max())]
            v = int(sp["vector_chunk"][w])
            if v >= 0:
                need += [256 * v, min(256 * v + 256, ndof) - 1]
            assert lo <= min(need) and max(need) < hi, (w, lo, hi, need)
        assert sp["lds"] <= 150 * 1024 and sp["lds"] >= 16 * int((pl[:, 3] - pl[:, 2]).max())
