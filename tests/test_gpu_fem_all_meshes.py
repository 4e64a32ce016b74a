"""GPU parity on ALL of the reference's own FEM inputs: the 853 surface meshes under output/PointClouds/pcr_t_f*.vtk (the only
real data the reference tree holds for this path; tests/golden/fem_meshes_all.npz, made by tools/make_fem_fixtures.py), each as
the prism (C3D6) two-layer model the shipped launch files select (h = 0.5, E = 3500, nu = 0.495: Optimizer.cc:480,
FEA2.cc:1184-1219, :1312-1376, :1505-1658, :1811-1902).

* the RAW dumps (coincident points, repeated indices: zero Jacobians, NaN K_e in the reference) as ONE fem_create_batch of 853
  distinct topologies: the NaN pattern and every other bit of every mesh's assembled K;
* the same meshes without their degenerate triangles, as one batch AND one model per mesh: K, the Dirichlet penalty (id - 1
  quirk), a = uf - u0, f = K a bit for bit; strain energy within 1e-5."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle
from orb_slam2_e_amd.fem import FEA2, FEA2Batch, FEM_C3D6, extrude_elems, second_layer

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fem_meshes_all.npz")
RTOL = 1e-5


def _meshes(clean):
    z = dict(np.load(GOLD))          # (an NpzFile decompresses an array on every access)
    out = []
    for k in range(len(z["frame"])):
        top = z["points"][z["pt_off"][k]:z["pt_off"][k + 1]]
        tris = z["triangles"][z["tri_off"][k]:z["tri_off"][k + 1]]
        if clean:
            p = top[tris]
            tris = tris[~((p[:, 0] == p[:, 1]).all(1) | (p[:, 0] == p[:, 2]).all(1) | (p[:, 1] == p[:, 2]).all(1))]
        if len(tris):
            out.append((int(z["frame"][k]), top, tris))
    return out


def _bits_equal(a, b):
    """Equal NaN pattern, equal bits elsewhere."""
    na, nb = np.isnan(a), np.isnan(b)
    return np.array_equal(na, nb) and np.array_equal(a.view(np.uint32)[~na], b.view(np.uint32)[~nb])


def test_all_853_raw_meshes_one_batch_nan_pattern_and_bits():
    ms = _meshes(clean=False)
    assert len(ms) == 853
    nodes_l = [second_layer(top, 0.5) for _, top, _ in ms]
    elems_l = [extrude_elems(tris, len(top)) for _, top, tris in ms]
    fea = FEA2Batch(nodes_l, elems_l, FEM_C3D6)
    assert fea.Ksize == 6 * sum(len(top) for _, top, _ in ms)
    fea.MatrixAssembly()
    with_nan = 0
    for k, (frame, top, tris) in enumerate(ms):
        K = oracle.fem_assemble_dense(2, nodes_l[k], elems_l[k])
        with_nan += bool(np.isnan(K).any())
        assert _bits_equal(fea.K_dense(k), K), f"mesh {k} (pcr_t_f{frame}.vtk)"
    assert with_nan > 100          # the degenerate dumps are the rule, not the exception


def test_all_clean_meshes_batch_and_single_against_oracle():
    ms = _meshes(clean=True)
    assert len(ms) == 844          # nine dumps hold degenerate triangles only
    nodes_l = [second_layer(top, 0.5) for _, top, _ in ms]
    elems_l = [extrude_elems(tris, len(top)) for _, top, tris in ms]
    batch = FEA2Batch(nodes_l, elems_l, FEM_C3D6)
    batch.MatrixAssembly()
    # global Dirichlet ids of the batch: mesh k's bottom layer, vvDir_t = nTop + i (FEA2.cc:1198) in its own numbering
    ids_l = [np.arange(len(top), 2 * len(top), dtype=np.int32) for _, top, _ in ms]
    gids = np.concatenate([batch.node0[k] + ids for k, ids in enumerate(ids_l)]).astype(np.int32)
    batch.ImposeDirichletEncastre_K(gids)
    rng = np.random.default_rng(7)
    u0 = np.concatenate([n.ravel() for n in nodes_l])
    uf = (u0 + rng.normal(0, 0.01, u0.shape)).astype(np.float32)
    a_b = batch.ComputeDisplacement(uf, u0, gids)[0]
    f_b = batch.ComputeForces(a_b)[0]
    sE_b, nsE_b = batch.ComputeStrainEnergy(a_b)
    finite, worst = 0, 0.0
    for k, (frame, top, tris) in enumerate(ms):
        d0, d1 = int(batch.dof0[k]), int(batch.dof0[k + 1])
        K = oracle.fem_assemble_dense(2, nodes_l[k], elems_l[k])
        oracle.fem_dirichlet_K(K, ids_l[k])
        a = oracle.fem_displacement(uf[d0:d1], u0[d0:d1], ids_l[k])
        f = oracle.fem_matvec_dense(K, a)
        sE, nsE = oracle.fem_strain_energy(a, f)
        tag = f"mesh {k} (pcr_t_f{frame}.vtk)"
        assert _bits_equal(batch.K_dense(k), K), tag
        assert np.array_equal(a_b[d0:d1], a), tag
        assert _bits_equal(f_b[d0:d1], f), tag
        # a^T f: the reference sums it in float (Eigen's packet order, FEA2.cc:1886), the oracle in float left to right, the device
        # in double -- three roundings of one sum whose terms partly cancel.  The device must be within a float ulp or two of the
        # exact sum, and within 1e-5 of the oracle's figure on the scale the float sum's own rounding lives on, sum |a_i f_i|
        # (>= |sE|; on these meshes the oracle's left-to-right sum is itself up to 5.9e-5 of |sE| away from the exact one)
        exact = abs(float(np.dot(a.astype(np.float64), f.astype(np.float64))))
        scale = float(np.abs(a.astype(np.float64) * f.astype(np.float64)).sum())
        nel = len(a) // 3
        if np.isfinite(sE):
            finite += 1
            assert abs(sE_b[k] - exact) <= 2.5e-7 * exact and abs(nsE_b[k] - exact / nel) <= 4e-7 * exact / nel, tag
            assert abs(sE_b[k] - sE) <= RTOL * scale and abs(nsE_b[k] - nsE) <= RTOL * scale / nel, tag
            worst = max(worst, abs(sE_b[k] - sE) / abs(sE))
        else:
            assert not np.isfinite(sE_b[k]), tag
        # ... and as a model of its own, the way PoseOptimizationNR builds one per call
        one = FEA2(nodes_l[k], elems_l[k], FEM_C3D6)
        one.MatrixAssembly(); one.ImposeDirichletEncastre_K(ids_l[k])
        assert _bits_equal(one.K_dense(), K), tag
        f1 = one.ComputeForces(a)[0]
        assert _bits_equal(f1, f), tag
        s1, n1 = one.ComputeStrainEnergy(a)
        assert (s1[0] == sE_b[k] and n1[0] == nsE_b[k]) if np.isfinite(sE) else not np.isfinite(s1[0]), tag      # one model or a batch: the same bits
    assert finite >= 800 and worst < 1e-4          # measured: 5.9e-5 on the worst of the 844 (the float sum's own noise)
