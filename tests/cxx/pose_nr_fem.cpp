// orbslam_hip::PoseOptimizationNR_fem (include/orbslam_hip.hpp) driven by a scripted stand-in for g2o: built and run by
// tests/test_gpu_fem.py::test_pose_optimization_nr_fem_sequence, which compares the trial log with the oracle's literal
// sequence (Optimizer.cc:723-790, optimization_algorithm_levenberg.cpp:63-232).
// usage: pose_nr_fem <script.bin> <out.bin>
//   script: int32 {nElType, nTop, nFaces, nVertices, nDerived, T, I}, f64 lambdaInit, f32 top[3 nTop], i32 faces[nv nFaces],
//           i32 derived[4 nDerived], f64 pts[T][3 nVertices], f64 chi[T], f64 scale[T], i32 ok2[T], f64 iterChi[I]
//   out:    int32 nTrials, nIterations; per trial {f32 sE, nsE; f64 tempChi, currentChi, rho, lambda; i32 qmax, accepted};
//           int32 result per iteration
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <vector>

#include "orbslam_hip.hpp"

using namespace orbslam_hip;

struct Script {
    int nv = 0, T = 0, I = 0, t = -1, it = 0;
    bool afterSolve = false;
    double lambdaInit = 0;
    std::vector<double> pts, chi, scale, iterChi;
    std::vector<int32_t> ok2;
    bool exhausted = false;
    void initializeOptimization(int) {}
    double activeRobustChi2()
    {
        if (afterSolve) { afterSolve = false; return chi[t]; }
        if (it >= I) { exhausted = true; return 0; }
        return iterChi[it++];
    }
    void buildSystem() {}
    double computeLambdaInit() { return lambdaInit; }
    void push() {}
    void pop() {}
    void discardTop() {}
    bool solveAndUpdate(double)
    {
        if (t + 1 >= T) { exhausted = true; } else ++t;
        afterSolve = true;
        return ok2[t] != 0;
    }
    double computeScale(double) { return scale[t]; }
    void pointEstimates(std::vector<double> &xyz) { xyz.assign(pts.begin() + (size_t)t * 3 * nv, pts.begin() + (size_t)(t + 1) * 3 * nv); }
    bool terminate() { return exhausted; }
    void classifyOutliers(int) {}
};

template <class T> static void rd(FILE *f, std::vector<T> &v, size_t n)
{
    v.resize(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short script\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<int32_t> hd; rd(f, hd, 7);
    const int nElType = hd[0], nTop = hd[1], nFaces = hd[2], nVertices = hd[3], nDerived = hd[4], T = hd[5], I = hd[6];
    std::vector<double> li; rd(f, li, 1);
    std::vector<float> top; rd(f, top, (size_t)3 * nTop);
    std::vector<int32_t> faces; rd(f, faces, (size_t)(nElType == 1 ? 4 : 3) * nFaces);
    std::vector<int32_t> derived; rd(f, derived, (size_t)4 * nDerived);
    Script g2o;
    g2o.nv = nVertices; g2o.T = T; g2o.I = I; g2o.lambdaInit = li[0];
    rd(f, g2o.pts, (size_t)T * 3 * nVertices); rd(f, g2o.chi, T); rd(f, g2o.scale, T); rd(f, g2o.ok2, T); rd(f, g2o.iterChi, I);
    fclose(f);

    PoseOptimizationNR_fem nr(nElType);
    if (!nr.Compute(top, faces, nVertices, derived)) { fprintf(stderr, "Compute(1) failed: %d %s\n", nr.status(), orbx_last_error()); return 1; }
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
    fclose(o);
    printf("OK %zu trials, %d iterations, script exhausted: %d\n", log.size(), n, (int)g2o.exhausted);
    return 0;
}
