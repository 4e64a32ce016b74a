// orbslam_hip::PoseOptimizationNR_fem (include/orbslam_hip.hpp) in a CLOSED loop: the Levenberg trials come from a real (small)
// bundle problem -- oracle/mini_g2o.h, the stand-in for the g2o graph of Optimizer.cc:484-707 -- and every trial's strain energy
// from fem_trial_energy on the device.  Built and run by tests/test_gpu_fem.py::test_pose_optimization_nr_closed_loop, which
// compares the trial log, the results per iteration, the final pose / points and the inlier count with the oracle's literal
// loop on the same graph (oracle/pose_nr_oracle.c).
// usage: pose_nr_lm <scene.bin> <out.bin>
//   scene: int32 {nElType, nTop, nFaces, nVertices, nDerived, nKF, nEdges}, f32 top[3 nTop], i32 faces[nv nFaces], i32 derived[4 nDerived],
//          f64 R0[9], t0[3], kfR[9 nKF], kft[3 nKF], X0[3 nVertices], i32 e_pt[nEdges], e_cam[nEdges], f64 e_obs[2 nEdges], e_info[nEdges], e_K[4 nEdges]
//   out:   int32 nTrials, nIterations; per trial {f32 sE, nsE; f64 tempChi, currentChi, rho, lambda; i32 qmax, accepted}; int32 result per
//          iteration; f64 R[9], t[3], X[3 nVertices]; int32 inliers; u8 outlier[nVertices]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "orbslam_hip.hpp"
#include "../../oracle/mini_g2o.h"

using namespace orbslam_hip;

// the `Problem` PoseOptimizationNR_fem is written against (orbslam_hip.hpp: what g2o supplies in the real build)
struct MiniG2O {
    mg_problem *g;
    void initializeOptimization(int) { mg_initialize_optimization(g); }
    double activeRobustChi2() { return mg_active_robust_chi2(g); }
    void buildSystem() { mg_build_system(g); }
    double computeLambdaInit() { return mg_lambda_init(g); }
    void push() { mg_push(g); }
    void pop() { mg_pop(g); }
    void discardTop() {}
    bool solveAndUpdate(double lambda) { return mg_solve_and_update(g, lambda) != 0; }
    double computeScale(double lambda) { return mg_compute_scale(g, lambda); }
    void pointEstimates(std::vector<double> &xyz) { xyz.assign(g->X, g->X + 3 * (size_t)g->npts); }
    bool terminate() { return false; }
    void classifyOutliers(int) { mg_classify_outliers(g); }
};

template <class T> static void rd(FILE *f, T *v, size_t n)
{
    if (n && fread(v, sizeof(T), n, f) != n) { fprintf(stderr, "short scene\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hd[7]; rd(f, hd, 7);
    const int nElType = hd[0], nTop = hd[1], nFaces = hd[2], nVertices = hd[3], nDerived = hd[4], nKF = hd[5], nEdges = hd[6];
    std::vector<float> top((size_t)3 * nTop); rd(f, top.data(), top.size());
    std::vector<int32_t> faces((size_t)(nElType == 1 ? 4 : 3) * nFaces); rd(f, faces.data(), faces.size());
    std::vector<int32_t> derived((size_t)4 * nDerived); rd(f, derived.data(), derived.size());
    mg_problem *g = mg_create(nVertices, nKF, nEdges);
    rd(f, g->R, 9); rd(f, g->t, 3); rd(f, g->kfR, (size_t)9 * nKF); rd(f, g->kft, (size_t)3 * nKF); rd(f, g->X, (size_t)3 * nVertices);
    rd(f, g->e_pt, nEdges); rd(f, g->e_cam, nEdges); rd(f, g->e_obs, (size_t)2 * nEdges); rd(f, g->e_info, nEdges); rd(f, g->e_K, (size_t)4 * nEdges);
    fclose(f);

    PoseOptimizationNR_fem nr(nElType);
    if (!nr.Compute(top, faces, nVertices, derived)) { fprintf(stderr, "Compute(1) failed: %d %s\n", nr.status(), orbx_last_error()); return 1; }
    MiniG2O g2o{g};
    std::vector<PoseOptimizationNR_fem::Trial> log;
    std::vector<int> results;
    const int n = nr.Optimize(g2o, &log, &results);
    if (nr.status() != ORBX_OK) { fprintf(stderr, "status %d %s\n", nr.status(), orbx_last_error()); return 1; }
    FILE *o = fopen(argv[2], "wb");
    if (!o) return 2;
    const int32_t cnt[2] = {(int32_t)log.size(), n};
    fwrite(cnt, 4, 2, o);
    for (const auto &t : log) {
        fwrite(&t.sE, 4, 1, o); fwrite(&t.nsE, 4, 1, o);
        fwrite(&t.tempChi, 8, 1, o); fwrite(&t.currentChi, 8, 1, o); fwrite(&t.rho, 8, 1, o); fwrite(&t.lambda, 8, 1, o);
        const int32_t q[2] = {t.qmax, t.accepted};
        fwrite(q, 4, 2, o);
    }
    for (int r : results) { const int32_t v = r; fwrite(&v, 4, 1, o); }
    fwrite(g->R, 8, 9, o); fwrite(g->t, 8, 3, o); fwrite(g->X, 8, (size_t)3 * nVertices, o);
    const int32_t inl = nVertices - g->nBad;          // Optimizer.cc:833: nInitialCorrespondences - nBad
    fwrite(&inl, 4, 1, o);
    fwrite(g->outlier, 1, nVertices, o);
    fclose(o);
    printf("OK %zu trials, %d iterations, %d inliers\n", log.size(), n, (int)inl);
    mg_free(g);
    return 0;
}
