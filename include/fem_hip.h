/*
 * fem_hip.h -- C ABI of the FEM core that the ORB_SLAM2_E fork injects into
 * Optimizer::PoseOptimizationNR (src/Optimizer.cc:478-834) through class FEA2
 * (Thirdparty/g2o/g2o/FEA/{include/FEA2.h,src/FEA2.cc}).
 *
 * Scope: element stiffness K_e (C3D8 / C3D6 exactly as the reference computes
 * them, plus a linear tetrahedron), global assembly (CSR instead of the
 * reference's dense vector<vector<float>>), penalty Dirichlet, f = K*a, strain
 * energy, and a Jacobi-preconditioned CG solve that fills the slot of the
 * reference's dead dense inverse (FEA2.cc:1661-1691).  The PCL meshing front half
 * of FEA2 (FEA2.cc:124-1181) is out of scope: meshes come in as node / element
 * arrays.  Same conventions as orbslam_hip.h (int status, host pointers unless
 * named *_dev, no CPU fallback).
 *
 * A fem_model holds a batch of independent meshes as one block-diagonal CSR matrix
 * resident in HBM, in one of two layouts:
 *   fem_create        `nmesh` meshes that share one topology (element connectivity) but have
 *                     their own node coordinates, hence their own matrices; ONE column-index /
 *                     row-pointer array serves all of them.  Arrays are [nmesh][ndof].
 *   fem_create_batch  meshes of different sizes and topologies -- the reference builds a new
 *                     mesh on every call (src/Optimizer.cc:480, FEA2.cc:80-121) -- concatenated:
 *                     node, element, dof and non-zero numbers are global over the batch (mesh k
 *                     starts at the offsets fem_batch_offsets returns), vectors are one array of
 *                     all dofs, Dirichlet ids / dofs are global numbers; per-mesh results (strain
 *                     energy, CG residuals) come back one per mesh.
 */
#ifndef FEM_HIP_H
#define FEM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    FEM_C3D8 = 1, /* nElType 1: 8-node hexahedron, FEA2::ComputeKeiC3D8 (FEA2.cc:1244-1309) */
    FEM_C3D6 = 2, /* nElType 2: 6-node prism,      FEA2::ComputeKeiC3D6 (FEA2.cc:1312-1376) */
    FEM_TET4 = 4  /* 4-node linear tetrahedron (no reference counterpart)        */
};

typedef struct fem_model fem_model;

/* FEA2::SetSecondLayer (FEA2.cc:1184-1219): nodes_out[2*ntop][3] = top || top-(h,h,h).
 * Pure host helper (3*ntop subtractions). */
int fem_second_layer(const float *top, int ntop, float h, float *nodes_out);

/* FEA2::FEA2(frameId, E, nu, h, fg, nElType, debug) (FEA2.cc:48-73) + the mesh.
 * nodes[nmesh][nn][3] float, elems[ne][npe] int32 node ids (npe = 8 / 6 / 4).
 * E is unsigned as in FEA2.h:236.  Builds the CSR pattern on the host once. */
int fem_create(int eltype, const float *nodes, int nmesh, int nn, const int32_t *elems, int ne,
               unsigned int E, float nu, float fg, fem_model **out);
/* A batch of meshes with their own topologies: mesh k has mesh_nn[k] nodes and mesh_ne[k] elements;
 * nodes[sum nn][3] and elems[sum ne][npe] are the meshes one after the other, element node ids local
 * to their mesh.  fem_sizes then returns the number of meshes and the TOTAL dofs / non-zeros;
 * fem_get_ke / fem_get_csr address a mesh's own elements, rows and columns.  The LM hook
 * (fem_trial_*) is per single mesh and not available on such a model. */
int fem_create_batch(int eltype, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const float *nodes,
                     const int32_t *elems, unsigned int E, float nu, float fg, fem_model **out);
/* First node / element / non-zero of every mesh in the batch numbering, [nmesh + 1] each (any may be NULL);
 * dof offset = 3 * node offset. */
int fem_batch_offsets(const fem_model *m, int32_t *node0, int32_t *elem0, int32_t *nnz0);
int fem_destroy(fem_model *m);

/* Sizes: meshes, dofs per mesh (Ksize = 3*nn), scalar non-zeros per mesh. */
int fem_sizes(const fem_model *m, int *nmesh, int *ndof, int64_t *nnz);
/* lambda, G, D[36] as the constructor computes them in float (FEA2.cc:53-62). */
int fem_material(const fem_model *m, float *lambda, float *G, float *D36);

/* MatrixAssemblyC3D8 / MatrixAssemblyC3D6 (FEA2.cc:1379-1624): K_e for every
 * element of every mesh, then the global K; each entry sums its element
 * contributions in element order, so it equals the reference's dense scatter-add
 * bit for bit (given the same float K_e). */
int fem_assemble(fem_model *m);

/* ImposeDirichletEncastre_K (FEA2.cc:1628-1645): K[d][d] = klarge for
 * d = 3*(ids[i]-1)+{0,1,2} -- the reference's off-by-one is kept. */
int fem_dirichlet_penalty(fem_model *m, const int32_t *ids, int nids, float klarge);
/* Exact elimination (rows/cols of the listed dofs -> identity); used for the SPD
 * tetrahedral benchmark, not by the reference. */
int fem_dirichlet_eliminate(fem_model *m, const int32_t *dofs, int ndofs);

/* Parity accessors. */
int fem_get_ke(fem_model *m, int mesh, int elem, float *ke /* (3*npe)^2 */);
int fem_get_csr(fem_model *m, int mesh, int32_t *rowptr /*ndof+1*/, int32_t *col /*nnz*/, float *val /*nnz*/);

/* What fem_create / fem_create_batch work out on the host before their first device call -- the symbolic phase (CSR pattern
 * of 3 x 3 node blocks in the reference's scatter order), the chunking of rows over workgroups, the node-block tables of the
 * SpMV and the chunk table of the compute-unit-resident CG -- WITHOUT touching the device: no GPU is needed.  Introspection /
 * test aid (the CPU tests and the host sanitizer build, tools/asan_host.sh, drive the host code through it).
 * uniform_copies > 0: fem_create's layout (nmesh must be 1: that many meshes sharing the one topology); 0: fem_create_batch's
 * (nmesh meshes of their own sizes, concatenated in global numbering).  Every output array may be NULL; sizes: rowptr[ndof + 1],
 * lcol[nnz], diag[ndof] (position of each row's diagonal entry), bp[ndof / 3 + 1] (blocks before a block row), bcol3[nnz / 9]
 * (first column of a block), rcd[4 nrcd] = {first block row, block rows, first block, blocks} per chunk of the resident CG and
 * rcfirst[meshes + 1] (both only if info->resident), chunk_mesh[nchunk_tot] (batch layout only). */
typedef struct {
    int32_t ndof, nblk, spb, spmv_lds, fused_lds, nchunk_tot, nchunk_s_tot, resident, resident_big, resident_lds, nrcd, maxel;
    int64_t nnz, ncontrib;
    int32_t rows_lds, reserved;   /* LDS bytes of the shared-row assembly (0: the entry-by-entry kernel assembles) */
} fem_plan_info;
int fem_plan(int eltype, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const int32_t *elems, int uniform_copies,
             fem_plan_info *info, int32_t *rowptr, int32_t *lcol, int32_t *diag, int32_t *bp, int32_t *bcol3, int32_t *rcd,
             int32_t *rcfirst, int32_t *chunk_mesh);

/* Diagnostic: how many k_fem_cg_xcd launches this model has made, and how many times one of them gave up and was made good.  The kernel's
 * workgroups wait for each other, so every one of them needs a compute unit at the same time; when the chip is full of other streams' work
 * a participant may not be scheduled for long (observed: seconds under three threads of 64-frame extraction batches).  A participant is
 * waited for 2 ms; then the launch leaves x, r, p and the scalars untouched, the next fem_cg_result / convergence check runs the owed
 * iterations on the launch-per-phase path (same bits), and the model's next 32 fem_cg_iterate calls (doubling while such events follow each other) stay on that path. */
int fem_cg_one_launch_stats(fem_model *m, int64_t *launches, int64_t *recovered);

/* How fem_cg_iterate / fem_cg will run ONE mesh of this topology (host only, no GPU needed): info6 = {one-launch kernel
 * k_fem_cg_xcd eligible (a single mesh of at most 8,192 dofs whose chunk tables fit), participating workgroups P (<= 64), chunks per
 * workgroup of the kernel variant (1, 3 or 6), its LDS bytes, vector chunks (256 rows), SpMV chunks (48 or 96 rows)}; plan[P][4] =
 * {first SpMV chunk, end | (own vector chunk + 1) << 16, first dof, end of the column range whose p the workgroup keeps in LDS} per
 * workgroup (may be NULL; room for 64 entries).  The
 * kernel gives the launch-per-phase path's results bit for bit under either preconditioner (FEM_CG_XCD=0 selects that path). */
int fem_plan_single_cg(int eltype, int nn, const int32_t *elems, int ne, int32_t *info6, int32_t *plan);

/* Cross-check of the symbolic phase (host only): the linear-pass formulation fem_create uses against the first, list-based
 * one, table by table (block pattern, contribution lists in scatter order, element lists).  ORBX_OK when identical. */
int fem_plan_selfcheck(int eltype, int nn, const int32_t *elems, int ne);

/* ComputeDisplacement + ImposeDirichletEncastre_a (FEA2.cc:1799-1808,1648-1658):
 * a = uf - u0, then a[3*(id-1)+k] = 1/klarge.  Arrays [nmesh][ndof]. */
int fem_displacement(fem_model *m, const float *uf, const float *u0, const int32_t *ids, int nids,
                     float klarge, float *a);
/* ComputeForces (FEA2.cc:1811-1816 -> MultiplyMatricesEigen :1694-1729): f = K*a
 * in float, each row summed left to right.  a, f: [nmesh][ndof]. */
int fem_matvec(fem_model *m, const float *a, float *f);
/* ComputeStrainEnergy + NormalizeStrainEnergy (FEA2.cc:1877-1902):
 * sE[mesh] = |a^T K a|, nsE = sE / int(Ksize/3).  Either output may be NULL. */
int fem_strain_energy(fem_model *m, const float *a, float *sE, float *nsE);

/* The LM hook (Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:159-175)
 * with K, u0, the Dirichlet list and the derived-node table resident on the device
 * (no per-trial copy of K).  setup: u0[Ksize] (FEA2::Set_u0), the
 * Dirichlet ids of ImposeDirichletEncastre_a, npoints = number of optimiser vertices
 * (pFEA2->vVertices), derived[nder][4] = {count 2|3, i0, i1, i2} = vNewPointsBase
 * (mid-edge / barycentre nodes recomputed by FEA2::Set_uf, FEA2.cc:1746-1775);
 * npoints + nder = top-layer nodes.  energy: points[nmesh][npoints][3] are the
 * vertex estimates in double (GetPointCoordinates casts them to float); runs Set_uf,
 * ComputeDisplacement, ComputeForces, ComputeStrainEnergy, NormalizeStrainEnergy and
 * returns sE / nsE per mesh (a_out, optional: the displacement vector). */
int fem_trial_setup(fem_model *m, const float *u0, const int32_t *ids, int nids, float klarge, int npoints,
                    const int32_t *derived, int nder);
int fem_trial_energy(fem_model *m, const double *points, float *a_out, float *sE, float *nsE);

/* Jacobi-preconditioned conjugate gradients, double vectors on the float matrix:
 * K x = b per mesh, x0 = 0.  Runs until `iters` iterations, or earlier when every
 * mesh has ||r|| <= tol*||b|| (checked every 25 iterations; tol <= 0 disables).
 * b, x: [nmesh][ndof] double.  iters_done / relres[nmesh] may be NULL.
 * The solver works on a block-major copy of the matrix values (3 x 3 node blocks kept together), made from the values
 * as they stand when fem_cg / fem_cg_setup is called: another 4 bytes per non-zero of device memory while a model has
 * been solved with, and changes of K (assembly, penalties) after that call are seen by the next call, not by
 * fem_cg_iterate.  Batches of 64 or more meshes of at most 14,288 dofs each (what 160 KB of LDS hold: (n + 6,192) x 8
 * bytes with the search direction alone in LDS; up to 7,168 dofs also K p stays there) run one mesh per compute unit
 * with the iteration vectors in LDS and registers: fem_cg_iterate(n) is ONE launch of n iterations, fem_cg launches
 * slices of 25 iterations between its convergence tests.  ONE mesh of at most 8,192 dofs (either preconditioner) also runs
 * fem_cg_iterate(n) as one launch (k_fem_cg_xcd: <= 32 workgroups of one XCD stay resident and exchange tagged 16-byte
 * granules; fem_plan_single_cg tells); its results equal the launch-per-phase path's bit for bit, at most six such launches
 * are in flight per process (further calls take the other path), and should one of its bounded waits ever time out,
 * fem_cg_result / fem_cg return ORBX_ERR_HIP (FEM_CG_XCD=0 keeps the kernel off).  Other models launch phase by phase.
 * All paths sum in fixed orders: a given model, right-hand side and iteration count give the same bits every run. */
int fem_cg(fem_model *m, const double *b, double *x, int iters, double tol, int *iters_done, double *relres);

/* Preconditioner of fem_cg / fem_cg_setup (default FEM_PRECOND_JACOBI: z = r / diag).  FEM_PRECOND_TWO_LEVEL adds a coarse correction,
 * z = r / diag + Z Ac^-1 Z^T r: Z = the six rigid-body modes (translations, rotations about the centroid) of 2 x 2 x 2 geometric
 * aggregates of each mesh's nodes (split at the midpoint of the bounding box per axis) -- 48 coarse dofs per mesh --, Ac = Z^T K Z,
 * rows of Z at dofs named in fem_dirichlet_penalty / fem_dirichlet_eliminate since the last fem_assemble are zero.  Point-Jacobi
 * leaves the smooth error of a near-incompressible solid (nu -> 0.5) to thousands of iterations; the coarse term removes it:
 * 1,274 -> 470 iterations to ||r|| <= 1e-8 ||b|| on BASELINE config 3's mesh.  Costs eight passes over the matrix and a 48 x 48
 * inverse per mesh in fem_cg_setup (0.24 ms for one 6,591-dof mesh, 1.7 ms for 256) and 6-13 per cent per iteration (batches that run on the compute units; k_fem_cg_resident).  Coarse dofs without a free fine dof, and modes that the modes before them
 * already span or nearly span (Cholesky pivot <= 1e-4 of the diagonal: the rotations of an aggregate whose free nodes lie on or near a line), are dropped.  No reference counterpart (neither has the CG): the oracle's
 * oracle_fem_cg_two_level is the definition.  Takes effect at the next fem_cg / fem_cg_setup.  A dof whose diagonal entry is 0 -- a node that belongs to no element: K has a zero row and column there,
 * as for the points no triangle uses in 800 of the reference's 853 surface meshes -- stays at 0 under either preconditioner and counts as
 * constrained in the coarse space (until round 5: 1 / 0, every vector NaN). */
enum { FEM_PRECOND_JACOBI = 0, FEM_PRECOND_TWO_LEVEL = 1 };
int fem_cg_preconditioner(fem_model *m, int kind);
/* Z^T K Z of one mesh as the last fem_cg_setup formed it (48 x 48, row-major); introspection for the tests. */
int fem_cg_coarse_matrix(fem_model *m, int mesh, double *Ac);

/* Resident variants for timing: upload the right-hand side and reset the solver
 * state; run n iterations asynchronously on `stream` (no host sync, no
 * convergence test); read the iterate back. */
int fem_cg_setup(fem_model *m, const double *b);
int fem_cg_iterate(fem_model *m, int n, void *stream);
int fem_cg_result(fem_model *m, double *x, double *relres);
/* n launches of the CG SpMV kernel alone (Ap = K*p on the resident vectors). */
int fem_spmv_repeat(fem_model *m, int n, void *stream);

/* HIP-event timing per kernel kind (k_fem_ke, k_fem_assemble, k_fem_spmv,
 * k_fem_cg_update, k_fem_cg_dir), as orbx_profile_*. */
int fem_profile_enable(fem_model *m, int on);
int fem_profile_read(fem_model *m, int max_kinds, const char **names, double *total_ms, int64_t *launches, int *nkinds);

#ifdef __cplusplus
}
#endif
#endif /* FEM_HIP_H */
