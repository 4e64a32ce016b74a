/* CPU oracle: Optimizer::PoseOptimizationNR's non-linear optimisation in a closed loop -- src/Optimizer.cc:726-809 (four
 * rounds of optimizer.optimize(10) with the inlier / outlier pass between them, the pose and point write-back) around
 * SparseOptimizer::optimize (Thirdparty/g2o/g2o/core/sparse_optimizer.cpp:425-504: stop at the first result that is
 * not OK) around OptimizationAlgorithmLevenberg::solve with this fork's FEM hook
 * (core/optimization_algorithm_levenberg.cpp:63-241; the hook :159-199, GetPointCoordinates :293-311).
 *
 * TEST INFRASTRUCTURE: imported only by tests/ (and never timed).  The graph -- g2o's side, out of the hot path's scope --
 * is the stand-in of oracle/mini_g2o.h; the FEM side is this oracle's literal restatement of FEA2 (fem_oracle.c:
 * Set_uf + ComputeDisplacement = oracle_fem_trial_displacement, ComputeForces = oracle_fem_matvec_dense on the dense K,
 * ComputeStrainEnergy / NormalizeStrainEnergy = oracle_fem_strain_energy).  "parity unpinned", like the rest of the oracle:
 * the reference holds no vectors for this path. */
#include <float.h>
#include <stdint.h>

#include "mini_g2o.h"

void oracle_fem_trial_displacement(const double *points, int npoints, const int *derived, int nder, const float *u0,
                                   const int *ids, int nids, float Klarge, float *a);
void oracle_fem_matvec_dense(const float *K, int n, const float *a, float *f);
float oracle_fem_strain_energy(const float *a, const float *f, int n, float *nsE);

/* One record per Levenberg trial, as PoseOptimizationNR_fem::Trial (include/orbslam_hip.hpp) */
typedef struct { float sE, nsE; double tempChi, currentChi, rho, lambda; int qmax, accepted; } oracle_lm_trial;

/* scene: the graph as flat arrays (see tests/pose_nr_scene.py); K: dense Ksize x Ksize after ImposeDirichletEncastre_K;
 * nVertices point vertices + nder derived nodes = the top layer.  Out: trials[<= max_trials], results per iteration
 * (1 OK, 2 Terminate), the final pose (R row major, t), the final points, the inlier count of :833.  Returns the number
 * of trials, or -1 if the logs overflow. */
int oracle_pose_optimization_nr(int npts, int nkf, int nedges, const double *R0, const double *t0, const double *kfR, const double *kft,
                                const double *X0, const int *e_pt, const int *e_cam, const double *e_obs, const double *e_info,
                                const double *e_K, const float *K, int Ksize, const float *u0, const int *ids, int nids,
                                const int *derived, int nder, float Klarge, oracle_lm_trial *trials, int max_trials,
                                int *results, int max_results, int *nresults, double *R_out, double *t_out, double *X_out,
                                int *inliers, unsigned char *outlier_out)
{
    mg_problem *g = mg_create(npts, nkf, nedges);
    float *a = (float *)malloc(sizeof(float) * Ksize), *f = (float *)malloc(sizeof(float) * Ksize);
    int nt = 0, nr = 0, it, i, overflow = 0;
    /* OptimizationAlgorithmLevenberg's members, levenberg.cpp:46-57 */
    double _currentLambda = -1., _ni = 2.;
    const double _goodStepLowerScale = 1. / 3., _goodStepUpperScale = 2. / 3.;
    const int _maxTrialsAfterFailure = 10;
    int _nBad = 0;
    const int its[4] = {10, 10, 10, 10};                                        /* Optimizer.cc:730 */

    memcpy(g->R, R0, sizeof(g->R)); memcpy(g->t, t0, sizeof(g->t));
    if (nkf) { memcpy(g->kfR, kfR, sizeof(double) * 9 * nkf); memcpy(g->kft, kft, sizeof(double) * 3 * nkf); }
    memcpy(g->X, X0, sizeof(double) * 3 * npts);
    memcpy(g->e_pt, e_pt, sizeof(int) * nedges); memcpy(g->e_cam, e_cam, sizeof(int) * nedges);
    memcpy(g->e_obs, e_obs, sizeof(double) * 2 * nedges); memcpy(g->e_info, e_info, sizeof(double) * nedges);
    memcpy(g->e_K, e_K, sizeof(double) * 4 * nedges);

    for (it = 0; it < 4 && !overflow; it++) {                                   /* Optimizer.cc:733 */
        int iteration, ok = 1;
        mg_initialize_optimization(g);                                          /* :735 */
        for (iteration = 0; iteration < its[it] && ok; iteration++) {           /* sparse_optimizer.cpp:453 */
            /* ---- OptimizationAlgorithmLevenberg::solve(iteration), bInFEA set ---- */
            double currentChi = mg_active_robust_chi2(g);                       /* :80-97 */
            double tempChi = currentChi;
            const double iniChi = currentChi;
            double rho = 0;
            int qmax = 0, result;
            mg_build_system(g);                                                 /* :102 */
            if (iteration == 0) {                                               /* :109-114 */
                _currentLambda = mg_lambda_init(g);
                _ni = 2;
                _nBad = 0;
            }
            do {
                int ok2, good;
                float sE = 0.0, nsE = 0.0;
                float w_rE = 1.0, w_sE = 5.0;                                   /* :184-185 */
                double scale;
                mg_push(g);                                                     /* :123 */
                ok2 = mg_solve_and_update(g, _currentLambda);                   /* :130-146 */
                tempChi = mg_active_robust_chi2(g);                             /* :148-157 */
                if (!ok2) tempChi = DBL_MAX;                                    /* :159-160 */
                /* the hook, :162-199 */
                oracle_fem_trial_displacement(g->X, npts, derived, nder, u0, ids, nids, Klarge, a);   /* GetPointCoordinates, Set_uf, ComputeDisplacement */
                oracle_fem_matvec_dense(K, Ksize, a, f);                        /* ComputeForces */
                sE = oracle_fem_strain_energy(a, f, Ksize, &nsE);               /* ComputeStrainEnergy, NormalizeStrainEnergy */
                if (qmax == 0) {                                                /* :186-193 */
                    w_rE = 1.0;
                    w_sE = 2.0;
                    currentChi += nsE;
                }
                tempChi = w_rE * tempChi + w_sE * nsE;                          /* :198 (float * double + float * float) */
                rho = (currentChi - tempChi);                                   /* :201 */
                scale = mg_compute_scale(g, _currentLambda);
                scale += 1e-3;
                rho /= scale;
                good = rho > 0 && isfinite(tempChi);
                if (good) {                                                     /* :207-217 */
                    double alpha = 1. - pow((2 * rho - 1), 3);
                    double scaleFactor;
                    alpha = fmin(alpha, _goodStepUpperScale);
                    scaleFactor = fmax(_goodStepLowerScale, alpha);
                    _currentLambda *= scaleFactor;
                    _ni = 2;
                    currentChi = tempChi;
                    /* discardTop(): the pushed state is dropped */
                } else {                                                        /* :218-223 */
                    _currentLambda *= _ni;
                    _ni *= 2;
                    mg_pop(g);
                }
                if (nt >= max_trials) { overflow = 1; break; }
                trials[nt].sE = sE; trials[nt].nsE = nsE; trials[nt].tempChi = tempChi; trials[nt].currentChi = currentChi;
                trials[nt].rho = rho; trials[nt].lambda = _currentLambda; trials[nt].qmax = qmax; trials[nt].accepted = good;
                nt++;
                qmax++;
            } while (rho < 0 && qmax < _maxTrialsAfterFailure);                 /* :226 (terminate() is false) */
            if (overflow) break;
            if (qmax == _maxTrialsAfterFailure || rho == 0) result = 2;         /* :228-229 Terminate */
            else {
                if ((iniChi - currentChi) * 1e3 < iniChi) _nBad++;              /* :232-235 */
                else _nBad = 0;
                result = _nBad >= 3 ? 2 : 1;                                    /* :237-240 */
            }
            if (nr >= max_results) { overflow = 1; break; }
            results[nr++] = result;
            ok = result == 1;                                                   /* sparse_optimizer.cpp:470 */
        }
        mg_classify_outliers(g);                                                /* Optimizer.cc:747-790 */
    }
    *nresults = nr;
    memcpy(R_out, g->R, sizeof(g->R)); memcpy(t_out, g->t, sizeof(g->t));      /* :798-801 */
    memcpy(X_out, g->X, sizeof(double) * 3 * npts);                             /* :804-809 */
    *inliers = npts - g->nBad;                                                  /* :833: nInitialCorrespondences - nBad */
    for (i = 0; i < npts; i++) outlier_out[i] = g->outlier[i];
    free(a); free(f);
    mg_free(g);
    return overflow ? -1 : nt;
}
