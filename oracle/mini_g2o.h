/* TEST INFRASTRUCTURE -- never part of the product, never linked into liborbslam_hip.so.
 *
 * A stand-in for the g2o graph of Optimizer::PoseOptimizationNR (src/Optimizer.cc:484-707): one free camera pose
 * (VertexSE3Expmap, id 0), fixed keyframe poses, free marginalised points (VertexSBAPointXYZ), reprojection edges
 * (EdgeSE3ProjectXYZ) with Huber kernels, and the BlockSolver_6_3 linear step as a Schur complement on the one pose.
 * g2o itself is OUT of the hot path's scope and absent from this image; what the FEM hook needs from it is
 * a source of real Levenberg trials -- estimates, chi2, scale, solver failures -- in a closed loop, and this is it:
 * the oracle's literal LM loop (oracle/pose_nr_oracle.c) and the product's PoseOptimizationNR_fem (include/orbslam_hip.hpp,
 * driven by tests/cxx/pose_nr_lm.cpp) both run on THIS graph, the first with the oracle's CPU strain energy, the second with
 * fem_trial_energy on the device.  Plain C99 (static functions), includable from C++.
 *
 * What follows the reference's g2o line by line (Thirdparty/g2o/g2o):
 *   error / Jacobians            types/types_six_dof_expmap.h:95-100, types_six_dof_expmap.cpp:103-147
 *   pose update                  types/types_six_dof_expmap.h (oplusImpl: exp(update) * estimate), types/se3quat.h:223-257
 *   Huber kernel, weighting      core/robust_kernel_impl.cpp:78-91, core/base_edge.h:96-102, core/base_binary_edge.hpp:55-114
 *   lambda init / scale          core/optimization_algorithm_levenberg.cpp:242-267 (tau = 1e-5)
 *   inlier / outlier pass        src/Optimizer.cc:752-790
 * What does not: poses are rotation matrices (g2o: unit quaternions, renormalised after every product), the 3 x 3 and
 * 6 x 6 solves are a cofactor inverse and a plain Cholesky (g2o: Eigen), sums run in edge order.  Double throughout. */
#ifndef ORACLE_MINI_G2O_H
#define ORACLE_MINI_G2O_H

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int npts, nkf, nedges;
    double R[9], t[3];        /* vertex 0 */
    double *kfR, *kft;        /* [nkf][9], [nkf][3], fixed */
    double *X;                /* [npts][3] */
    int *e_pt, *e_cam;        /* e_cam: -1 = vertex 0, else keyframe index */
    double *e_obs, *e_info, *e_K;   /* [nedges][2], [nedges], [nedges][4] = fx fy cx cy */
    int *e_level;
    double *e_err;            /* [nedges][2]: _error as the last computeError left it */
    unsigned char *outlier, *reloc_check;   /* pFrame->mvbOutlier[idx], pMP->bRelocCheck (per map point) */
    int nBad;
    double delta;             /* sqrt(5.991), Optimizer.cc:623 */
    /* linear system of the active part */
    double Hpp[36], bp[6];
    double *Hll, *bl, *Hpl;   /* [npts][9], [npts][3], [npts][18] (6 x 3, row major) */
    unsigned char *pt_active;
    double *x;                /* [6 + 3 npts]: pose first, then the points in index order (inactive: 0) */
    double sR[9], st[3], *sX; /* push / pop */
} mg_problem;

static mg_problem *mg_create(int npts, int nkf, int nedges)
{
    mg_problem *p = (mg_problem *)calloc(1, sizeof(mg_problem));
    p->npts = npts; p->nkf = nkf; p->nedges = nedges;
    p->kfR = (double *)calloc((size_t)9 * (nkf + 1), sizeof(double)); p->kft = (double *)calloc((size_t)3 * (nkf + 1), sizeof(double));
    p->X = (double *)calloc((size_t)3 * npts, sizeof(double)); p->sX = (double *)calloc((size_t)3 * npts, sizeof(double));
    p->e_pt = (int *)calloc(nedges, sizeof(int)); p->e_cam = (int *)calloc(nedges, sizeof(int)); p->e_level = (int *)calloc(nedges, sizeof(int));
    p->e_obs = (double *)calloc((size_t)2 * nedges, sizeof(double)); p->e_info = (double *)calloc(nedges, sizeof(double));
    p->e_K = (double *)calloc((size_t)4 * nedges, sizeof(double)); p->e_err = (double *)calloc((size_t)2 * nedges, sizeof(double));
    p->outlier = (unsigned char *)calloc(npts, 1); p->reloc_check = (unsigned char *)malloc(npts); memset(p->reloc_check, 1, npts);
    p->Hll = (double *)calloc((size_t)9 * npts, sizeof(double)); p->bl = (double *)calloc((size_t)3 * npts, sizeof(double));
    p->Hpl = (double *)calloc((size_t)18 * npts, sizeof(double)); p->pt_active = (unsigned char *)calloc(npts, 1);
    p->x = (double *)calloc((size_t)6 + 3 * npts, sizeof(double));
    p->delta = sqrt(5.991);
    return p;
}

static void mg_free(mg_problem *p)
{
    if (!p) return;
    free(p->kfR); free(p->kft); free(p->X); free(p->sX); free(p->e_pt); free(p->e_cam); free(p->e_level); free(p->e_obs); free(p->e_info);
    free(p->e_K); free(p->e_err); free(p->outlier); free(p->reloc_check); free(p->Hll); free(p->bl); free(p->Hpl); free(p->pt_active); free(p->x);
    free(p);
}

static void mg_cam(const mg_problem *p, int e, const double **R, const double **t)
{
    if (p->e_cam[e] < 0) { *R = p->R; *t = p->t; }
    else { *R = p->kfR + 9 * p->e_cam[e]; *t = p->kft + 3 * p->e_cam[e]; }
}

/* EdgeSE3ProjectXYZ::computeError, types_six_dof_expmap.h:95-100 */
static void mg_compute_error(mg_problem *p, int e)
{
    const double *R, *t, *X = p->X + 3 * p->e_pt[e], *K = p->e_K + 4 * e;
    double c[3];
    int i;
    mg_cam(p, e, &R, &t);
    for (i = 0; i < 3; i++) c[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + t[i];
    p->e_err[2 * e] = p->e_obs[2 * e] - (c[0] / c[2] * K[0] + K[2]);
    p->e_err[2 * e + 1] = p->e_obs[2 * e + 1] - (c[1] / c[2] * K[1] + K[3]);
}

static double mg_edge_chi2(const mg_problem *p, int e)      /* _error . (information * _error), information = invSigma2 I */
{
    return p->e_err[2 * e] * (p->e_info[e] * p->e_err[2 * e]) + p->e_err[2 * e + 1] * (p->e_info[e] * p->e_err[2 * e + 1]);
}

static void mg_huber(const mg_problem *p, double e2, double rho[3])   /* robust_kernel_impl.cpp:78-91 */
{
    const double dsqr = p->delta * p->delta;
    if (e2 <= dsqr) { rho[0] = e2; rho[1] = 1.; rho[2] = 0.; }
    else { const double sqrte = sqrt(e2); rho[0] = 2 * sqrte * p->delta - dsqr; rho[1] = p->delta / sqrte; rho[2] = -0.5 * rho[1] / e2; }
}

/* optimizer.initializeOptimization(0): the level-0 edges and the non-fixed vertices they touch */
static void mg_initialize_optimization(mg_problem *p)
{
    int e;
    memset(p->pt_active, 0, p->npts);
    for (e = 0; e < p->nedges; e++)
        if (p->e_level[e] == 0) p->pt_active[p->e_pt[e]] = 1;
}

/* computeActiveErrors() + activeRobustChi2() */
static double mg_active_robust_chi2(mg_problem *p)
{
    double chi = 0, rho[3];
    int e;
    for (e = 0; e < p->nedges; e++) {
        if (p->e_level[e] != 0) continue;
        mg_compute_error(p, e);
        mg_huber(p, mg_edge_chi2(p, e), rho);
        chi += rho[0];
    }
    return chi;
}

/* _solver->buildSystem(): linearizeOplus + constructQuadraticForm of every active edge at the current estimates, with
 * the errors the last computeActiveErrors stored */
static void mg_build_system(mg_problem *p)
{
    int e, i, j, k;
    memset(p->Hpp, 0, sizeof(p->Hpp)); memset(p->bp, 0, sizeof(p->bp));
    memset(p->Hll, 0, sizeof(double) * 9 * p->npts); memset(p->bl, 0, sizeof(double) * 3 * p->npts); memset(p->Hpl, 0, sizeof(double) * 18 * p->npts);
    for (e = 0; e < p->nedges; e++) {
        const double *R, *t, *X = p->X + 3 * p->e_pt[e], *K = p->e_K + 4 * e;
        double c[3], tmp[6], A[6], B[12], rho[3], w, wr[2];
        const int pt = p->e_pt[e];
        if (p->e_level[e] != 0) continue;
        mg_cam(p, e, &R, &t);
        for (i = 0; i < 3; i++) c[i] = R[3 * i] * X[0] + R[3 * i + 1] * X[1] + R[3 * i + 2] * X[2] + t[i];
        {
            const double x = c[0], y = c[1], z = c[2], z_2 = z * z, fx = K[0], fy = K[1];
            tmp[0] = fx; tmp[1] = 0; tmp[2] = -x / z * fx;
            tmp[3] = 0; tmp[4] = fy; tmp[5] = -y / z * fy;
            for (i = 0; i < 2; i++)                                         /* _jacobianOplusXi = -1./z * tmp * R */
                for (j = 0; j < 3; j++) A[3 * i + j] = -1. / z * (tmp[3 * i] * R[j] + tmp[3 * i + 1] * R[3 + j] + tmp[3 * i + 2] * R[6 + j]);
            B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx; B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
            B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy; B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
        }
        mg_huber(p, mg_edge_chi2(p, e), rho);
        w = rho[1] * p->e_info[e];                                          /* robustInformation */
        wr[0] = -(p->e_info[e] * p->e_err[2 * e]) * rho[1]; wr[1] = -(p->e_info[e] * p->e_err[2 * e + 1]) * rho[1];   /* omega_r *= rho[1] */
        for (i = 0; i < 3; i++) {
            p->bl[3 * pt + i] += A[i] * wr[0] + A[3 + i] * wr[1];
            for (j = 0; j < 3; j++) p->Hll[9 * pt + 3 * i + j] += A[i] * w * A[j] + A[3 + i] * w * A[3 + j];
        }
        if (p->e_cam[e] < 0) {
            for (i = 0; i < 6; i++) {
                p->bp[i] += B[i] * wr[0] + B[6 + i] * wr[1];
                for (j = 0; j < 6; j++) p->Hpp[6 * i + j] += B[i] * w * B[j] + B[6 + i] * w * B[6 + j];
                for (k = 0; k < 3; k++) p->Hpl[18 * pt + 3 * i + k] += B[i] * w * A[k] + B[6 + i] * w * A[3 + k];
            }
        }
    }
}

/* computeLambdaInit(), levenberg.cpp:242-256: tau * the largest diagonal entry of the active vertices' Hessians */
static double mg_lambda_init(const mg_problem *p)
{
    double m = 0;
    int i, j;
    for (j = 0; j < 6; j++) m = fmax(fabs(p->Hpp[7 * j]), m);
    for (i = 0; i < p->npts; i++)
        if (p->pt_active[i]) for (j = 0; j < 3; j++) m = fmax(fabs(p->Hll[9 * i + 4 * j]), m);
    return 1e-5 * m;
}

static void mg_push(mg_problem *p) { memcpy(p->sR, p->R, sizeof(p->R)); memcpy(p->st, p->t, sizeof(p->t)); memcpy(p->sX, p->X, sizeof(double) * 3 * p->npts); }
static void mg_pop(mg_problem *p) { memcpy(p->R, p->sR, sizeof(p->R)); memcpy(p->t, p->st, sizeof(p->t)); memcpy(p->X, p->sX, sizeof(double) * 3 * p->npts); }

static int mg_inv3(const double *a, double *o)
{
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (!(fabs(det) > 0) || !isfinite(det)) return 0;
    o[0] = c0 / det; o[1] = (a[2] * a[7] - a[1] * a[8]) / det; o[2] = (a[1] * a[5] - a[2] * a[4]) / det;
    o[3] = c1 / det; o[4] = (a[0] * a[8] - a[2] * a[6]) / det; o[5] = (a[2] * a[3] - a[0] * a[5]) / det;
    o[6] = c2 / det; o[7] = (a[1] * a[6] - a[0] * a[7]) / det; o[8] = (a[0] * a[4] - a[1] * a[3]) / det;
    return 1;
}

/* SE3Quat::exp(update) * estimate, se3quat.h:223-257 (omega = update[0..2], upsilon = update[3..5]) */
static void mg_oplus_pose(mg_problem *p, const double *u)
{
    const double w[3] = {u[0], u[1], u[2]}, theta = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double O[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
    double O2[9], Re[9], V[9], te[3], Rn[9], tn[3];
    int i, j, k;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += O[3 * i + k] * O[3 * k + j]; O2[3 * i + j] = s; }
    if (theta < 0.00001) {
        for (i = 0; i < 9; i++) { Re[i] = (i % 4 == 0 ? 1. : 0.) + O[i] + O2[i]; V[i] = Re[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3);
        for (i = 0; i < 9; i++) { Re[i] = (i % 4 == 0 ? 1. : 0.) + a * O[i] + b * O2[i]; V[i] = (i % 4 == 0 ? 1. : 0.) + b * O[i] + c * O2[i]; }
    }
    for (i = 0; i < 3; i++) te[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    for (i = 0; i < 3; i++) {
        for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += Re[3 * i + k] * p->R[3 * k + j]; Rn[3 * i + j] = s; }
        tn[i] = Re[3 * i] * p->t[0] + Re[3 * i + 1] * p->t[1] + Re[3 * i + 2] * p->t[2] + te[i];
    }
    memcpy(p->R, Rn, sizeof(Rn)); memcpy(p->t, tn, sizeof(tn));
}

/* setLambda(lambda, true); solve(); update(x); restoreDiagonal(): the Schur complement of the points on the one pose.
 * Returns ok2 (0: a point block or the reduced system is not positive definite; x = 0 then, nothing moves). */
static int mg_solve_and_update(mg_problem *p, double lambda)
{
    double S[36], bs[6], Lc[36], y[6], xp[6];
    double *inv = (double *)malloc(sizeof(double) * 9 * (p->npts > 0 ? p->npts : 1));
    int i, j, k, n, ok = 1;
    memcpy(S, p->Hpp, sizeof(S)); memcpy(bs, p->bp, sizeof(bs));
    for (j = 0; j < 6; j++) S[7 * j] += lambda;
    memset(p->x, 0, sizeof(double) * (6 + 3 * (size_t)p->npts));
    for (n = 0; n < p->npts && ok; n++) {
        double D[9], W[18];   /* W = Hpl Hll^-1 (6 x 3) */
        const double *H = p->Hpl + 18 * n;
        if (!p->pt_active[n]) continue;
        memcpy(D, p->Hll + 9 * n, sizeof(D));
        for (j = 0; j < 3; j++) D[4 * j] += lambda;
        if (!mg_inv3(D, inv + 9 * n)) { ok = 0; break; }
        for (i = 0; i < 6; i++)
            for (j = 0; j < 3; j++) W[3 * i + j] = H[3 * i] * inv[9 * n + j] + H[3 * i + 1] * inv[9 * n + 3 + j] + H[3 * i + 2] * inv[9 * n + 6 + j];
        for (i = 0; i < 6; i++) {
            bs[i] -= W[3 * i] * p->bl[3 * n] + W[3 * i + 1] * p->bl[3 * n + 1] + W[3 * i + 2] * p->bl[3 * n + 2];
            for (j = 0; j < 6; j++) S[6 * i + j] -= W[3 * i] * H[3 * j] + W[3 * i + 1] * H[3 * j + 1] + W[3 * i + 2] * H[3 * j + 2];
        }
    }
    if (ok) {                                          /* Cholesky S = Lc Lc^T */
        memset(Lc, 0, sizeof(Lc));
        for (i = 0; i < 6 && ok; i++)
            for (j = 0; j <= i; j++) {
                double s = S[6 * i + j];
                for (k = 0; k < j; k++) s -= Lc[6 * i + k] * Lc[6 * j + k];
                if (i == j) { if (!(s > 0) || !isfinite(s)) { ok = 0; break; } Lc[6 * i + i] = sqrt(s); }
                else Lc[6 * i + j] = s / Lc[6 * j + j];
            }
    }
    if (ok) {
        for (i = 0; i < 6; i++) { double s = bs[i]; for (k = 0; k < i; k++) s -= Lc[6 * i + k] * y[k]; y[i] = s / Lc[6 * i + i]; }
        for (i = 5; i >= 0; i--) { double s = y[i]; for (k = i + 1; k < 6; k++) s -= Lc[6 * k + i] * xp[k]; xp[i] = s / Lc[6 * i + i]; }
        memcpy(p->x, xp, sizeof(xp));
        for (n = 0; n < p->npts; n++) {
            double r[3];
            const double *H = p->Hpl + 18 * n;
            if (!p->pt_active[n]) continue;
            for (j = 0; j < 3; j++) { double s = p->bl[3 * n + j]; for (i = 0; i < 6; i++) s -= H[3 * i + j] * xp[i]; r[j] = s; }
            for (j = 0; j < 3; j++) p->x[6 + 3 * n + j] = inv[9 * n + 3 * j] * r[0] + inv[9 * n + 3 * j + 1] * r[1] + inv[9 * n + 3 * j + 2] * r[2];
        }
        mg_oplus_pose(p, p->x);                        /* _optimizer->update(x): oplus on every active vertex */
        for (n = 0; n < p->npts; n++)
            if (p->pt_active[n]) for (j = 0; j < 3; j++) p->X[3 * n + j] += p->x[6 + 3 * n + j];
    }
    free(inv);
    return ok;
}

/* computeScale(), levenberg.cpp:258-267: sum x_j (lambda x_j + b_j) over the whole vector */
static double mg_compute_scale(const mg_problem *p, double lambda)
{
    double s = 0;
    int j, n;
    for (j = 0; j < 6; j++) s += p->x[j] * (lambda * p->x[j] + p->bp[j]);
    for (n = 0; n < p->npts; n++)
        if (p->pt_active[n]) for (j = 0; j < 3; j++) s += p->x[6 + 3 * n + j] * (lambda * p->x[6 + 3 * n + j] + p->bl[3 * n + j]);
    return s;
}

/* The pass after every optimize(10), src/Optimizer.cc:747-790 (chi2[it] = 5.991 in all four rounds) */
static void mg_classify_outliers(mg_problem *p)
{
    int e;
    p->nBad = 0;                                                   /* :747 */
    for (e = 0; e < p->nedges; e++) {
        const int idx = p->e_pt[e];
        if (p->outlier[idx]) mg_compute_error(p, e);               /* :759-760 */
        if (mg_edge_chi2(p, e) > 5.991) {                          /* :764-773 */
            p->outlier[idx] = 1;
            p->e_level[e] = 1;
            if (p->reloc_check[idx]) { p->nBad++; p->reloc_check[idx] = 0; }
        } else {                                                   /* :774-783 (chi2 <= 5.991; a NaN falls through both, as there) */
            if (mg_edge_chi2(p, e) <= 5.991) {
                p->outlier[idx] = 0;
                p->e_level[e] = 0;
                if (!p->reloc_check[idx]) { p->reloc_check[idx] = 1; p->nBad--; }
            }
        }
    }
}

#endif /* ORACLE_MINI_G2O_H */
