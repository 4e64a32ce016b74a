/*
 * oracle/fem_oracle.c -- CPU restatement of the FEM core of FEA2 (ORB_SLAM2_E fork).
 *
 * TEST INFRASTRUCTURE ONLY (see orb_oracle.c header).  PARITY UNPINNED: the
 * reference has no tests/golden vectors for this path and FEA2.cc cannot be built
 * here (needs PCL + Eigen).  Pinned by first-principles known-answer tests only
 * (symmetry, rigid-translation null space, patch tests).
 *
 * Follows Thirdparty/g2o/g2o/FEA/src/FEA2.cc: ctor :48-73 (Lame matrix, Gauss
 * points), SetSecondLayer :1184-1219, ComputeKeiC3D8 :1244-1309, ComputeKeiC3D6
 * :1312-1376 (literal, including the signed Jacobian and the three sign
 * deviations in the inverse Jacobian), MatrixAssemblyC3D8/6
 * :1379-1624 (dense, element order), ImposeDirichletEncastre_K/_a :1628-1658
 * (the id-1 quirk, App. C5), ComputeDisplacement :1799-1808, ComputeForces
 * :1811-1816, ComputeStrainEnergy :1877-1894, NormalizeStrainEnergy :1897-1902.
 * The linear tetrahedron and the CG solve have NO reference counterpart (the
 * reference's only solve is a dead dense inverse, :1661-1691): for them this
 * file is the definition.
 *
 * All arithmetic in float exactly as written in the reference (-ffp-contract=off),
 * CG in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* FEA2::FEA2, FEA2.cc:53-62 */
void oracle_fem_material(unsigned int E, float nu, float *lambda_out, float *G_out, float *D /*36*/)
{
    float lambda = (nu * E) / ((1 + nu) * (1 - 2 * nu));
    float G = E / (2 * (1 + nu));
    int i;
    for (i = 0; i < 36; i++) D[i] = 0.0f;
    D[0] = D[7] = D[14] = lambda + 2 * G;
    D[1] = D[2] = D[6] = D[8] = D[12] = D[13] = lambda;
    D[21] = D[28] = D[35] = G;
    if (lambda_out) *lambda_out = lambda;
    if (G_out) *G_out = G;
}

/* Gauss points, FEA2.cc:64-72 */
void oracle_fem_gauss(float fg, float *gs /*24*/)
{
    const float s[8][3] = {{-1, -1, -1}, {+1, -1, -1}, {+1, +1, -1}, {-1, +1, -1},
                           {-1, -1, +1}, {+1, -1, +1}, {+1, +1, +1}, {-1, +1, +1}};
    int i, j;
    for (i = 0; i < 8; i++)
        for (j = 0; j < 3; j++) gs[3 * i + j] = s[i][j] < 0 ? -fg : +fg;
}

/* Shared tail of ComputeKeiC3D8/C3D6: Jacobian, "inverse", B, BtD, BtDB*Jac.
 * dN[a][n] = dN_n/d(xi|eta|zeta), nn nodes. */
static void ke_accumulate(int nn, const float dNdxi[], const float dNdeta[], const float dNdzeta[],
                          const float *P /*nn x 3*/, const float *D, float *Ke)
{
    const int nd = 3 * nn;
    float J_00 = 0, J_01 = 0, J_02 = 0, J_10 = 0, J_11 = 0, J_12 = 0, J_20 = 0, J_21 = 0, J_22 = 0, Jac;
    float J1_00, J1_01, J1_02, J1_10, J1_11, J1_12, J1_20, J1_21, J1_22;
    float dNdx[8], dNdy[8], dNdz[8];
    float B[6][24], BtD[24][6];
    int n, i, j;
    /* left-to-right sums as written (:1263-1271); first term assigned, rest added */
    for (n = 0; n < nn; n++) {
        if (n == 0) {
            J_00 = dNdxi[0] * P[0];   J_01 = dNdxi[0] * P[1];   J_02 = dNdxi[0] * P[2];
            J_10 = dNdeta[0] * P[0];  J_11 = dNdeta[0] * P[1];  J_12 = dNdeta[0] * P[2];
            J_20 = dNdzeta[0] * P[0]; J_21 = dNdzeta[0] * P[1]; J_22 = dNdzeta[0] * P[2];
        } else {
            J_00 = J_00 + dNdxi[n] * P[3 * n];   J_01 = J_01 + dNdxi[n] * P[3 * n + 1];   J_02 = J_02 + dNdxi[n] * P[3 * n + 2];
            J_10 = J_10 + dNdeta[n] * P[3 * n];  J_11 = J_11 + dNdeta[n] * P[3 * n + 1];  J_12 = J_12 + dNdeta[n] * P[3 * n + 2];
            J_20 = J_20 + dNdzeta[n] * P[3 * n]; J_21 = J_21 + dNdzeta[n] * P[3 * n + 1]; J_22 = J_22 + dNdzeta[n] * P[3 * n + 2];
        }
    }
    Jac = J_00 * J_11 * J_22 + J_01 * J_12 * J_20 + J_10 * J_21 * J_02 - J_20 * J_11 * J_02 - J_10 * J_01 * J_22 - J_21 * J_12 * J_00;
    J1_00 = (+1) * ((J_11 * J_22) - (J_21 * J_12)) / Jac; J1_01 = (-1) * ((J_01 * J_22) - (J_21 * J_02)) / Jac; J1_02 = (-1) * ((J_01 * J_12) - (J_11 * J_02)) / Jac;
    J1_10 = (-1) * ((J_10 * J_22) - (J_20 * J_12)) / Jac; J1_11 = (-1) * ((J_00 * J_22) - (J_20 * J_02)) / Jac; J1_12 = (-1) * ((J_00 * J_12) - (J_10 * J_02)) / Jac;
    J1_20 = (+1) * ((J_10 * J_21) - (J_20 * J_11)) / Jac; J1_21 = (-1) * ((J_00 * J_21) - (J_20 * J_01)) / Jac; J1_22 = (-1) * ((J_00 * J_11) - (J_10 * J_01)) / Jac;
    for (n = 0; n < nn; n++) {
        dNdx[n] = J1_00 * dNdxi[n] + J1_01 * dNdeta[n] + J1_02 * dNdzeta[n];
        dNdy[n] = J1_10 * dNdxi[n] + J1_11 * dNdeta[n] + J1_12 * dNdzeta[n];
        dNdz[n] = J1_20 * dNdxi[n] + J1_21 * dNdeta[n] + J1_22 * dNdzeta[n];
    }
    for (n = 0; n < nn; n++) { /* B as laid out at :1288-1293 */
        const int c = 3 * n;
        B[0][c] = dNdx[n]; B[0][c + 1] = 0.0f;    B[0][c + 2] = 0.0f;
        B[1][c] = 0.0f;    B[1][c + 1] = dNdy[n]; B[1][c + 2] = 0.0f;
        B[2][c] = 0.0f;    B[2][c + 1] = 0.0f;    B[2][c + 2] = dNdz[n];
        B[3][c] = dNdy[n]; B[3][c + 1] = dNdx[n]; B[3][c + 2] = 0.0f;
        B[4][c] = dNdz[n]; B[4][c + 1] = 0.0f;    B[4][c + 2] = dNdx[n];
        B[5][c] = 0.0f;    B[5][c + 1] = dNdz[n]; B[5][c + 2] = dNdy[n];
    }
    for (i = 0; i < nd; i++)
        for (j = 0; j < 6; j++)
            BtD[i][j] = B[0][i] * D[0 * 6 + j] + B[1][i] * D[1 * 6 + j] + B[2][i] * D[2 * 6 + j] + B[3][i] * D[3 * 6 + j] + B[4][i] * D[4 * 6 + j] + B[5][i] * D[5 * 6 + j];
    for (i = 0; i < nd; i++)
        for (j = 0; j < nd; j++) {
            float aux = BtD[i][0] * B[0][j] + BtD[i][1] * B[1][j] + BtD[i][2] * B[2][j] + BtD[i][3] * B[3][j] + BtD[i][4] * B[4][j] + BtD[i][5] * B[5][j];
            Ke[i * nd + j] += aux * Jac;
        }
}

/* FEA2::ComputeKeiC3D8, FEA2.cc:1244-1309 */
void oracle_fem_ke_c3d8(const float *P /*8x3*/, const float *D, const float *gs, float *Ke /*24x24*/)
{
    int ops;
    memset(Ke, 0, sizeof(float) * 24 * 24);
    for (ops = 0; ops < 8; ops++) {
        float xi = gs[3 * ops], eta = gs[3 * ops + 1], zeta = gs[3 * ops + 2];
        float a[8], b[8], c[8];
        a[0] = -0.125 * ((1 - eta) * (1 - zeta)); b[0] = -0.125 * ((1 - xi) * (1 - zeta)); c[0] = -0.125 * ((1 - xi) * (1 - eta));
        a[1] = +0.125 * ((1 - eta) * (1 - zeta)); b[1] = -0.125 * ((1 + xi) * (1 - zeta)); c[1] = -0.125 * ((1 + xi) * (1 - eta));
        a[2] = +0.125 * ((1 + eta) * (1 - zeta)); b[2] = +0.125 * ((1 + xi) * (1 - zeta)); c[2] = -0.125 * ((1 + xi) * (1 + eta));
        a[3] = -0.125 * ((1 + eta) * (1 - zeta)); b[3] = +0.125 * ((1 - xi) * (1 - zeta)); c[3] = -0.125 * ((1 - xi) * (1 + eta));
        a[4] = -0.125 * ((1 - eta) * (1 + zeta)); b[4] = -0.125 * ((1 - xi) * (1 + zeta)); c[4] = +0.125 * ((1 - xi) * (1 - eta));
        a[5] = +0.125 * ((1 - eta) * (1 + zeta)); b[5] = -0.125 * ((1 + xi) * (1 + zeta)); c[5] = +0.125 * ((1 + xi) * (1 - eta));
        a[6] = +0.125 * ((1 + eta) * (1 + zeta)); b[6] = +0.125 * ((1 + xi) * (1 + zeta)); c[6] = +0.125 * ((1 + xi) * (1 + eta));
        a[7] = -0.125 * ((1 + eta) * (1 + zeta)); b[7] = +0.125 * ((1 - xi) * (1 + zeta)); c[7] = +0.125 * ((1 - xi) * (1 + eta));
        ke_accumulate(8, a, b, c, P, D, Ke);
    }
}

/* FEA2::ComputeKeiC3D6, FEA2.cc:1312-1376 (prism, same 8 hex Gauss points: App. C4) */
void oracle_fem_ke_c3d6(const float *P /*6x3*/, const float *D, const float *gs, float *Ke /*18x18*/)
{
    int ops;
    memset(Ke, 0, sizeof(float) * 18 * 18);
    for (ops = 0; ops < 8; ops++) {
        float xi = gs[3 * ops], eta = gs[3 * ops + 1], zeta = gs[3 * ops + 2];
        float a[6], b[6], c[6];
        a[0] = -(1 + zeta) / 2; b[0] = -(1 + zeta) / 2; c[0] = (1 - xi - eta) / 2;
        a[1] = (1 + zeta) / 2;  b[1] = 0.0;             c[1] = xi / 2;
        a[2] = 0.0;             b[2] = (1 + zeta) / 2;  c[2] = eta / 2;
        a[3] = -(1 - zeta) / 2; b[3] = -(1 - zeta) / 2; c[3] = -(1 - xi - eta) / 2;
        a[4] = (1 - zeta) / 2;  b[4] = 0.0;             c[4] = -xi / 2;
        a[5] = 0.0;             b[5] = (1 - zeta) / 2;  c[5] = -eta / 2;
        ke_accumulate(6, a, b, c, P, D, Ke);
    }
}

/* Linear tetrahedron (no reference counterpart): constant strain,
 * K_e = (B^T D B) * V with V = |det[p1-p0,p2-p0,p3-p0]| / 6, float arithmetic in
 * the same BtD -> BtDB order as the reference elements. */
void oracle_fem_ke_tet4(const float *P /*4x3*/, const float *D, float *Ke /*12x12*/)
{
    float e1x = P[3] - P[0], e1y = P[4] - P[1], e1z = P[5] - P[2];
    float e2x = P[6] - P[0], e2y = P[7] - P[1], e2z = P[8] - P[2];
    float e3x = P[9] - P[0], e3y = P[10] - P[1], e3z = P[11] - P[2];
    /* cofactors: gradients of N1..N3 are rows of inv([e1;e2;e3]) transposed */
    float c1x = e2y * e3z - e2z * e3y, c1y = e2z * e3x - e2x * e3z, c1z = e2x * e3y - e2y * e3x;
    float c2x = e3y * e1z - e3z * e1y, c2y = e3z * e1x - e3x * e1z, c2z = e3x * e1y - e3y * e1x;
    float c3x = e1y * e2z - e1z * e2y, c3y = e1z * e2x - e1x * e2z, c3z = e1x * e2y - e1y * e2x;
    float det = e1x * c1x + e1y * c1y + e1z * c1z;
    float V = fabsf(det) / 6;
    float gx[4], gy[4], gz[4], B[6][12], BtD[12][6];
    int n, i, j;
    gx[1] = c1x / det; gy[1] = c1y / det; gz[1] = c1z / det;
    gx[2] = c2x / det; gy[2] = c2y / det; gz[2] = c2z / det;
    gx[3] = c3x / det; gy[3] = c3y / det; gz[3] = c3z / det;
    gx[0] = -(gx[1] + gx[2] + gx[3]); gy[0] = -(gy[1] + gy[2] + gy[3]); gz[0] = -(gz[1] + gz[2] + gz[3]);
    for (n = 0; n < 4; n++) {
        const int c = 3 * n;
        B[0][c] = gx[n]; B[0][c + 1] = 0.0f;  B[0][c + 2] = 0.0f;
        B[1][c] = 0.0f;  B[1][c + 1] = gy[n]; B[1][c + 2] = 0.0f;
        B[2][c] = 0.0f;  B[2][c + 1] = 0.0f;  B[2][c + 2] = gz[n];
        B[3][c] = gy[n]; B[3][c + 1] = gx[n]; B[3][c + 2] = 0.0f;
        B[4][c] = gz[n]; B[4][c + 1] = 0.0f;  B[4][c + 2] = gx[n];
        B[5][c] = 0.0f;  B[5][c + 1] = gz[n]; B[5][c + 2] = gy[n];
    }
    for (i = 0; i < 12; i++)
        for (j = 0; j < 6; j++)
            BtD[i][j] = B[0][i] * D[j] + B[1][i] * D[6 + j] + B[2][i] * D[12 + j] + B[3][i] * D[18 + j] + B[4][i] * D[24 + j] + B[5][i] * D[30 + j];
    for (i = 0; i < 12; i++)
        for (j = 0; j < 12; j++) {
            float aux = BtD[i][0] * B[0][j] + BtD[i][1] * B[1][j] + BtD[i][2] * B[2][j] + BtD[i][3] * B[3][j] + BtD[i][4] * B[4][j] + BtD[i][5] * B[5][j];
            Ke[i * 12 + j] = aux * V;
        }
}

/* SetSecondLayer, FEA2.cc:1184-1219: bottom = top - (h,h,h); nodes = top || bottom. */
void oracle_fem_second_layer(const float *top, int nTop, float h, float *nodes /*2*nTop*3*/)
{
    int i;
    memcpy(nodes, top, sizeof(float) * 3 * nTop);
    for (i = 0; i < 3 * nTop; i++) nodes[3 * nTop + i] = top[i] - h;
}

static int nodes_per_elem(int eltype) { return eltype == 1 ? 8 : (eltype == 2 ? 6 : 4); }

void oracle_fem_ke(int eltype, const float *P, const float *D, const float *gs, float *Ke)
{
    if (eltype == 1) oracle_fem_ke_c3d8(P, D, gs, Ke);
    else if (eltype == 2) oracle_fem_ke_c3d6(P, D, gs, Ke);
    else oracle_fem_ke_tet4(P, D, Ke);
}

/* MatrixAssemblyC3D8/C3D6, FEA2.cc:1379-1624: dense K (n = 3*nn), scatter-add in
 * element order.  eltype: 1 = C3D8 (nElType 1), 2 = C3D6 (nElType 2), 4 = tet4.
 * elems holds all element node ids (the reference forms them as top ids ||
 * top ids + nTop, :1392-1399/:1518-1523; the caller does that). */
void oracle_fem_assemble_dense(int eltype, const float *nodes, int nn, const int *elems, int ne,
                               const float *D, const float *gs, float *K)
{
    const int npe = nodes_per_elem(eltype), nd = 3 * npe, n = 3 * nn;
    float P[24], Ke[24 * 24];
    int e, a, ni, nj, m, q;
    memset(K, 0, sizeof(float) * (size_t)n * n);
    for (e = 0; e < ne; e++) {
        const int *en = elems + (size_t)e * npe;
        for (a = 0; a < npe; a++) { P[3 * a] = nodes[3 * en[a]]; P[3 * a + 1] = nodes[3 * en[a] + 1]; P[3 * a + 2] = nodes[3 * en[a] + 2]; }
        oracle_fem_ke(eltype, P, D, gs, Ke);
        for (ni = 0; ni < npe; ni++)
            for (nj = 0; nj < npe; nj++)
                for (m = 0; m < 3; m++)
                    for (q = 0; q < 3; q++) {
                        const int r = en[ni] * 3 + m, c = en[nj] * 3 + q;
                        if (r >= n || c >= n) continue;
                        K[(size_t)r * n + c] += Ke[(3 * ni + m) * nd + 3 * nj + q];
                    }
    }
}

/* ImposeDirichletEncastre_K, FEA2.cc:1628-1645: K[d][d]=Klarge for d = 3*(id-1)+{0,1,2}. */
void oracle_fem_dirichlet_K(float *K, int n, const int *ids, int nids, float Klarge)
{
    int i, k;
    for (i = 0; i < nids; i++)
        for (k = 0; k < 3; k++) {
            const int d = 3 * (ids[i] - 1) + k;
            K[(size_t)d * n + d] = Klarge;
        }
}

/* ComputeDisplacement + ImposeDirichletEncastre_a, FEA2.cc:1799-1808,1648-1658 */
void oracle_fem_displacement(const float *uf, const float *u0, int n, const int *ids, int nids, float Klarge, float *a)
{
    int i, k;
    for (i = 0; i < n; i++) a[i] = uf[i] - u0[i];
    for (i = 0; i < nids; i++)
        for (k = 0; k < 3; k++) a[3 * (ids[i] - 1) + k] = 1 / Klarge;
}

/* ComputeForces f = K*a (dense, float; row sums taken left to right -- Eigen's
 * own summation order is not specified, hence a tolerance on this one). */
void oracle_fem_matvec_dense(const float *K, int n, const float *a, float *f)
{
    int i, j;
    for (i = 0; i < n; i++) {
        float s = 0.0f;
        for (j = 0; j < n; j++) s += K[(size_t)i * n + j] * a[j];
        f[i] = s;
    }
}

/* ComputeStrainEnergy sE = |a^T f| (:1877-1894) and NormalizeStrainEnergy
 * nsE = sE / int(Ksize/3) (:1897-1902). */
float oracle_fem_strain_energy(const float *a, const float *f, int n, float *nsE)
{
    float sE = 0.0f;
    int i;
    for (i = 0; i < n; i++) sE += a[i] * f[i];
    if (sE < 0.0) sE = -sE;
    if (nsE) *nsE = sE / (n / 3);
    return sE;
}

/* Dense -> CSR (pattern = non-zero entries plus the diagonal). Returns nnz. */
int oracle_fem_dense_to_csr(const float *K, int n, int *rowptr, int *col, float *val, int cap)
{
    int i, j, nnz = 0;
    for (i = 0; i < n; i++) {
        rowptr[i] = nnz;
        for (j = 0; j < n; j++)
            if (K[(size_t)i * n + j] != 0.0f || i == j) {
                if (nnz < cap) { col[nnz] = j; val[nnz] = K[(size_t)i * n + j]; }
                nnz++;
            }
    }
    rowptr[n] = nnz;
    return nnz;
}

/* Exact Dirichlet elimination on CSR (used by the SPD tet benchmark, SURVEY
 * App. C tail): rows/cols of constrained dofs -> identity. */
void oracle_fem_csr_eliminate(int n, const int *rowptr, const int *col, float *val, const uint8_t *fixed)
{
    int i, k;
    for (i = 0; i < n; i++)
        for (k = rowptr[i]; k < rowptr[i + 1]; k++)
            if (fixed[i] || fixed[col[k]]) val[k] = (col[k] == i) ? 1.0f : 0.0f;
}

/* Jacobi-preconditioned CG in double on a float CSR matrix (no reference
 * counterpart: this is the definition).  x0 = 0.  Stops after `iters`
 * iterations or when ||r|| <= tol*||b||.  Returns iterations done. */
int oracle_fem_cg(int n, const int *rowptr, const int *col, const float *val, const double *b, double *x,
                  int iters, double tol, double *relres)
{
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n);
    double *p = (double *)malloc(sizeof(double) * n), *Ap = (double *)malloc(sizeof(double) * n);
    double *dinv = (double *)malloc(sizeof(double) * n);
    double rz = 0, bb = 0, rr;
    int i, k, it = 0;
    for (i = 0; i < n; i++) {
        double d = 1.0;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) if (col[k] == i) d = (double)val[k];
        dinv[i] = d != 0.0 ? 1.0 / d : 0.0;      /* a dof without a diagonal -- a node in no element: K has a zero row and column there -- stays at 0 */
        x[i] = 0; r[i] = b[i]; z[i] = r[i] * dinv[i]; p[i] = z[i];
        rz += r[i] * z[i]; bb += b[i] * b[i];
    }
    rr = bb;
    while (it < iters && !(sqrt(rr) <= tol * sqrt(bb))) {
        double pAp = 0, alpha, beta, rz2 = 0;
        for (i = 0; i < n; i++) {
            double s = 0;
            for (k = rowptr[i]; k < rowptr[i + 1]; k++) s += (double)val[k] * p[col[k]];
            Ap[i] = s; pAp += p[i] * s;
        }
        alpha = pAp > 0.0 ? rz / pAp : 0.0;   /* a finished (or unloaded) system is frozen, not divided 0 / 0 */
        rr = 0;
        for (i = 0; i < n; i++) {
            x[i] += alpha * p[i]; r[i] -= alpha * Ap[i];
            z[i] = r[i] * dinv[i]; rz2 += r[i] * z[i]; rr += r[i] * r[i];
        }
        beta = rz > 0.0 ? rz2 / rz : 0.0; rz = rz2;
        for (i = 0; i < n; i++) p[i] = z[i] + beta * p[i];
        it++;
    }
    if (relres) *relres = bb > 0 ? sqrt(rr / bb) : 0;
    free(r); free(z); free(p); free(Ap); free(dinv);
    return it;
}

/* ---- Two-level preconditioner of the CG (no reference counterpart: this is the definition, as for the CG itself).
 * M^-1 = D^-1 + Z Ac^-1 Z^T, Ac = Z^T K Z, Z = the six rigid-body modes (three translations, three rotations about the
 * aggregate's centroid) of 2 x 2 x 2 geometric aggregates of the nodes: 48 coarse dofs.  A node's aggregate: per axis, bit =
 * (double)P > 0.5 * ((double)lo + (double)hi) with lo / hi the float extremes of that coordinate; aggregate = 4 bx + 2 by + bz.
 * q = (float)((double)P - centroid), centroid = double sum in node order / count.  Rows of constrained dofs (cmask != 0) are zero
 * in Z.  Ac is symmetrised, coarse dofs whose diagonal is <= 1e-12 of the largest, or whose Cholesky pivot is <= 1e-4 of their
 * diagonal (modes the earlier ones span or nearly span: kept at 6e-7 one made the iteration a function of the last bits), are dropped (row and column zero in the inverse), the inverse comes from the Cholesky
 * factorisation in double and is symmetrised again. */
#define CZ_NA 8
#define CZ_NC 48
void oracle_fem_coarse_space(int nn, const float *nodes, int *agg, float *q)
{
    double mid[3], cen[CZ_NA][3];
    int cnt[CZ_NA], i, k, a;
    for (k = 0; k < 3; k++) {
        float lo = nodes[k], hi = nodes[k];
        for (i = 1; i < nn; i++) { float v = nodes[3 * i + k]; if (v < lo) lo = v; if (v > hi) hi = v; }
        mid[k] = 0.5 * ((double)lo + (double)hi);
    }
    for (a = 0; a < CZ_NA; a++) { cnt[a] = 0; cen[a][0] = cen[a][1] = cen[a][2] = 0; }
    for (i = 0; i < nn; i++) {
        a = 0;
        for (k = 0; k < 3; k++) a = 2 * a + ((double)nodes[3 * i + k] > mid[k] ? 1 : 0);
        agg[i] = a; cnt[a]++;
        for (k = 0; k < 3; k++) cen[a][k] += (double)nodes[3 * i + k];
    }
    for (a = 0; a < CZ_NA; a++) for (k = 0; k < 3; k++) if (cnt[a]) cen[a][k] /= (double)cnt[a];
    for (i = 0; i < nn; i++) for (k = 0; k < 3; k++) q[3 * i + k] = (float)((double)nodes[3 * i + k] - cen[agg[i]][k]);
}
/* w = Z^T r */
static void cz_restrict(int nn, const int *agg, const float *q, const uint8_t *cmask, const double *r, double *w)
{
    int i, k;
    for (k = 0; k < CZ_NC; k++) w[k] = 0;
    for (i = 0; i < nn; i++) {
        double *wa = w + 6 * agg[i];
        const double r0 = cmask[3 * i] ? 0.0 : r[3 * i], r1 = cmask[3 * i + 1] ? 0.0 : r[3 * i + 1], r2 = cmask[3 * i + 2] ? 0.0 : r[3 * i + 2];
        const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
        wa[0] += r0; wa[1] += r1; wa[2] += r2;
        wa[3] += q1 * r2 - q2 * r1; wa[4] += q2 * r0 - q0 * r2; wa[5] += q0 * r1 - q1 * r0;   /* q x r */
    }
}
/* c = Z v */
static void cz_prolong(int nn, const int *agg, const float *q, const uint8_t *cmask, const double *v, double *c)
{
    int i;
    for (i = 0; i < nn; i++) {
        const double *va = v + 6 * agg[i];
        const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
        c[3 * i] = cmask[3 * i] ? 0.0 : va[0] + (va[4] * q2 - va[5] * q1);            /* v + omega x q */
        c[3 * i + 1] = cmask[3 * i + 1] ? 0.0 : va[1] + (va[5] * q0 - va[3] * q2);
        c[3 * i + 2] = cmask[3 * i + 2] ? 0.0 : va[2] + (va[3] * q1 - va[4] * q0);
    }
}
/* inverse of the symmetrised coarse matrix with the dropped dofs zeroed; a[] is overwritten */
void oracle_fem_coarse_inverse(double *a /*48 x 48, in: Ac, out: inverse*/)
{
    double L[CZ_NC][CZ_NC], inv[CZ_NC][CZ_NC], y[CZ_NC], dmax = 0;
    int keep[CZ_NC], i, j, k;
    for (i = 0; i < CZ_NC; i++) for (j = 0; j < i; j++) { double s = 0.5 * (a[i * CZ_NC + j] + a[j * CZ_NC + i]); a[i * CZ_NC + j] = a[j * CZ_NC + i] = s; }
    for (i = 0; i < CZ_NC; i++) if (a[i * CZ_NC + i] > dmax) dmax = a[i * CZ_NC + i];
    for (i = 0; i < CZ_NC; i++) keep[i] = a[i * CZ_NC + i] > 1e-12 * dmax;
    for (i = 0; i < CZ_NC; i++) for (j = 0; j < CZ_NC; j++) L[i][j] = 0;
    for (j = 0; j < CZ_NC; j++) {
        double s;
        if (!keep[j]) continue;
        s = a[j * CZ_NC + j];
        for (k = 0; k < j; k++) s -= L[j][k] * L[j][k];
        if (!(s > 1e-4 * a[j * CZ_NC + j])) { keep[j] = 0; for (k = 0; k < j; k++) L[j][k] = 0; continue; }   /* spanned by the modes before it: dropped too */
        L[j][j] = sqrt(s);
        for (i = j + 1; i < CZ_NC; i++) {
            if (!keep[i]) continue;
            s = a[i * CZ_NC + j];
            for (k = 0; k < j; k++) s -= L[i][k] * L[j][k];
            L[i][j] = s / L[j][j];
        }
    }
    for (k = 0; k < CZ_NC; k++) {                 /* column k of the inverse: L y = e_k, L^T x = y */
        for (i = 0; i < CZ_NC; i++) inv[i][k] = 0;
        if (!keep[k]) continue;
        for (i = 0; i < CZ_NC; i++) {
            double s = (i == k) ? 1.0 : 0.0;
            if (!keep[i]) { y[i] = 0; continue; }
            for (j = 0; j < i; j++) s -= L[i][j] * y[j];
            y[i] = s / L[i][i];
        }
        for (i = CZ_NC - 1; i >= 0; i--) {
            double s = y[i];
            if (!keep[i]) continue;
            for (j = i + 1; j < CZ_NC; j++) s -= L[j][i] * inv[j][k];
            inv[i][k] = s / L[i][i];
        }
    }
    for (i = 0; i < CZ_NC; i++) for (j = 0; j <= i; j++) a[i * CZ_NC + j] = a[j * CZ_NC + i] = 0.5 * (inv[i][j] + inv[j][i]);
}
/* The coarse matrix Z^T K Z of a float CSR matrix (column k = Z^T (K (Z e_k))); exported for the tests. */
void oracle_fem_coarse_matrix(int n, const int *rowptr, const int *col, const float *val, const float *nodes, const uint8_t *cmask,
                              double *Ac /*48 x 48*/)
{
    const int nn = n / 3;
    int *agg = (int *)malloc(sizeof(int) * nn), i, k, kk;
    float *q = (float *)malloc(sizeof(float) * 3 * nn);
    double *z = (double *)malloc(sizeof(double) * n), *y = (double *)malloc(sizeof(double) * n), e[CZ_NC], w[CZ_NC];
    oracle_fem_coarse_space(nn, nodes, agg, q);
    for (k = 0; k < CZ_NC; k++) {
        for (kk = 0; kk < CZ_NC; kk++) e[kk] = kk == k;
        cz_prolong(nn, agg, q, cmask, e, z);
        for (i = 0; i < n; i++) { double s = 0; for (kk = rowptr[i]; kk < rowptr[i + 1]; kk++) s += (double)val[kk] * z[col[kk]]; y[i] = s; }
        cz_restrict(nn, agg, q, cmask, y, w);
        for (kk = 0; kk < CZ_NC; kk++) Ac[kk * CZ_NC + k] = w[kk];
    }
    free(agg); free(q); free(z); free(y);
}
/* CG as oracle_fem_cg with z = r/diag + Z Ac^-1 Z^T r; r.z is formed as r.(r/diag) + (Z^T r).(Ac^-1 Z^T r). */
int oracle_fem_cg_two_level(int n, const int *rowptr, const int *col, const float *val, const double *b, double *x,
                            int iters, double tol, double *relres, const float *nodes, const uint8_t *cmask)
{
    const int nn = n / 3;
    double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n), *c = (double *)malloc(sizeof(double) * n);
    double *p = (double *)malloc(sizeof(double) * n), *Ap = (double *)malloc(sizeof(double) * n);
    double *dinv = (double *)malloc(sizeof(double) * n), Aci[CZ_NC * CZ_NC], w[CZ_NC], v[CZ_NC];
    int *agg = (int *)malloc(sizeof(int) * nn);
    float *q = (float *)malloc(sizeof(float) * 3 * nn);
    double rz = 0, bb = 0, rr, wv;
    int i, k, j, it = 0;
    /* dofs without a diagonal (a node in no element) count as constrained: zero rows in Z, and they stay at 0 */
    uint8_t *cm2 = (uint8_t *)malloc(n);
    for (i = 0; i < n; i++) {
        double d = 1.0;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) if (col[k] == i) d = (double)val[k];
        cm2[i] = (uint8_t)((cmask && cmask[i]) || d == 0.0);
    }
    cmask = cm2;
    oracle_fem_coarse_space(nn, nodes, agg, q);
    oracle_fem_coarse_matrix(n, rowptr, col, val, nodes, cmask, Aci);
    oracle_fem_coarse_inverse(Aci);
    for (i = 0; i < n; i++) {
        double d = 1.0;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) if (col[k] == i) d = (double)val[k];
        dinv[i] = d != 0.0 ? 1.0 / d : 0.0;
        x[i] = 0; r[i] = b[i]; z[i] = r[i] * dinv[i];
        rz += r[i] * z[i]; bb += b[i] * b[i];
    }
    cz_restrict(nn, agg, q, cmask, r, w);
    for (k = 0, wv = 0; k < CZ_NC; k++) { double s = 0; for (j = 0; j < CZ_NC; j++) s += Aci[k * CZ_NC + j] * w[j]; v[k] = s; wv += w[k] * s; }
    cz_prolong(nn, agg, q, cmask, v, c);
    rz += wv;
    for (i = 0; i < n; i++) p[i] = z[i] + c[i];
    rr = bb;
    while (it < iters && !(sqrt(rr) <= tol * sqrt(bb))) {
        double pAp = 0, alpha, beta, rz2 = 0;
        for (i = 0; i < n; i++) {
            double s = 0;
            for (k = rowptr[i]; k < rowptr[i + 1]; k++) s += (double)val[k] * p[col[k]];
            Ap[i] = s; pAp += p[i] * s;
        }
        alpha = pAp > 0.0 ? rz / pAp : 0.0;
        rr = 0;
        for (i = 0; i < n; i++) {
            x[i] += alpha * p[i]; r[i] -= alpha * Ap[i];
            z[i] = r[i] * dinv[i]; rz2 += r[i] * z[i]; rr += r[i] * r[i];
        }
        cz_restrict(nn, agg, q, cmask, r, w);
        for (k = 0, wv = 0; k < CZ_NC; k++) { double s = 0; for (j = 0; j < CZ_NC; j++) s += Aci[k * CZ_NC + j] * w[j]; v[k] = s; wv += w[k] * s; }
        cz_prolong(nn, agg, q, cmask, v, c);
        rz2 += wv;
        beta = rz > 0.0 ? rz2 / rz : 0.0; rz = rz2;
        for (i = 0; i < n; i++) p[i] = (z[i] + c[i]) + beta * p[i];
        it++;
    }
    if (relres) *relres = bb > 0 ? sqrt(rr / bb) : 0;
    free(r); free(z); free(c); free(p); free(Ap); free(dinv); free(agg); free(q); free(cm2);
    return it;
}

/* CSR y = A*x in double (float values), for residual checks. */
void oracle_fem_csr_matvec(int n, const int *rowptr, const int *col, const float *val, const double *x, double *y)
{
    int i, k;
    for (i = 0; i < n; i++) {
        double s = 0;
        for (k = rowptr[i]; k < rowptr[i + 1]; k++) s += (double)val[k] * x[col[k]];
        y[i] = s;
    }
}

/* The LM hook's per-trial sequence, optimization_algorithm_levenberg.cpp:159-175:
 * GetPointCoordinates (:293-311, double -> float), FEA2::Set_uf (FEA2.cc:1732-1796,
 * incl. the recomputed mid-edge / barycentre nodes of vNewPointsBase), ComputeDisplacement
 * (:1799-1808), then the caller runs ComputeForces / ComputeStrainEnergy on `a`.
 * derived[d] = {count (2|3), i0, i1, i2}; nTop = npoints + nder; u0 = [top || bottom]. */
void oracle_fem_trial_displacement(const double *points, int npoints, const int *derived, int nder, const float *u0,
                                   const int *ids, int nids, float Klarge, float *a)
{
    const int nTop = npoints + nder, n = 6 * nTop;
    float *top = (float *)malloc(sizeof(float) * 3 * (nTop > 0 ? nTop : 1));
    int i, k, d;
    for (i = 0; i < npoints; i++)
        for (k = 0; k < 3; k++) top[3 * i + k] = (float)points[3 * i + k];
    for (d = 0; d < nder; d++) {
        const int *e = derived + 4 * d;
        float *mi = top + 3 * (npoints + d);
        if (e[0] == 2) for (k = 0; k < 3; k++) mi[k] = (top[3 * e[1] + k] + top[3 * e[2] + k]) / 2;
        else for (k = 0; k < 3; k++) mi[k] = (top[3 * e[1] + k] + top[3 * e[2] + k] + top[3 * e[3] + k]) / 3;
    }
    for (i = 0; i < 3 * nTop; i++) a[i] = top[i] - u0[i];          /* uf - u0, top layer */
    for (i = 3 * nTop; i < n; i++) a[i] = u0[i] - u0[i];           /* bottom layer: uf keeps vMPsXYZN_t2 */
    for (i = 0; i < nids; i++)
        for (k = 0; k < 3; k++) a[3 * (ids[i] - 1) + k] = 1 / Klarge;
    free(top);
}
