// C++ host-side smoke test of include/orbslam_hip.hpp (built and run by
// tests/test_cxx_host.py).  argv[1] = "nodevice": expect ORBX_ERR_NO_DEVICE.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "orbslam_hip.hpp"

using namespace orbslam_hip;

static int popcount_row(const uint8_t *a, const uint8_t *b)
{
    int d = 0;
    for (int i = 0; i < 32; ++i) d += __builtin_popcount((unsigned)(a[i] ^ b[i]));
    return d;
}

int main(int argc, char **argv)
{
    const bool nodevice = argc > 1 && !strcmp(argv[1], "nodevice");
    const int W = 640, H = 480;
    std::vector<uint8_t> img((size_t)W * H);
    unsigned s = 12345;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            s = s * 1664525u + 1013904223u;
            const int checker = (((x / 24) + (y / 24)) & 1) ? 200 : 60;
            img[(size_t)y * W + x] = (uint8_t)(checker + (int)((s >> 24) % 9) - 4);
        }
    ORBextractor ex(1000, 1.2f, 8, 20, 7);
    if (ex.status() != ORBX_OK || ex.GetLevels() != 8 || ex.GetScaleFactors().size() != 8) { printf("FAIL ctor\n"); return 1; }
    std::vector<KeyPoint> kps, kps2;
    std::vector<uint8_t> desc, desc2;
    ex(ImageView{img.data(), W, H, W}, ImageView{}, kps, desc);
    if (nodevice) {
        if (ex.status() != ORBX_ERR_NO_DEVICE || !kps.empty()) { printf("FAIL expected ORBX_ERR_NO_DEVICE, got %d\n", ex.status()); return 1; }
        printf("OK nodevice: %s\n", orbx_last_error());
        return 0;
    }
    if (ex.status() != ORBX_OK || kps.size() < 500 || desc.size() != kps.size() * 32) { printf("FAIL extract %d n=%zu\n", ex.status(), kps.size()); return 1; }
    ex(ImageView{img.data(), W, H, W}, ImageView{}, kps2, desc2);
    if (kps2.size() != kps.size() || memcmp(desc.data(), desc2.data(), desc.size())) { printf("FAIL determinism\n"); return 1; }
    ex(ImageView{}, ImageView{}, kps2, desc2); // empty image: outputs untouched
    if (kps2.size() != kps.size()) { printf("FAIL empty-image semantics\n"); return 1; }
    ex.SyncImagePyramid();
    if (ex.mvImagePyramid[0].cols != W || ex.mvImagePyramid[1].cols != 533 ||
        memcmp(ex.mvImagePyramid[0].pixels.data(), img.data(), img.size())) { printf("FAIL pyramid\n"); return 1; }

    ORBmatcher m(0.6f, true);
    for (int i = 0; i < 10; ++i)
        if (ORBmatcher::DescriptorDistance(&desc[32 * i], &desc[32 * (i + 7)]) != popcount_row(&desc[32 * i], &desc[32 * (i + 7)])) { printf("FAIL distance\n"); return 1; }
    std::vector<int32_t> m12;
    const int n = (int)kps.size();
    const int nm = m.MatchBruteForce(desc.data(), n, desc.data(), n, ORBmatcher::TH_LOW, m12);
    int self = 0;
    for (int i = 0; i < n; ++i) self += m12[i] == i;
    if (nm <= 0 || self < nm * 9 / 10) { printf("FAIL self-match nm=%d self=%d\n", nm, self); return 1; }

    // FEA2: 3x3 patch of triangles, prism elements
    std::vector<float> top;
    std::vector<int32_t> tris;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { top.push_back((float)i); top.push_back((float)j); top.push_back(0.1f * i * j); }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int a = i * 4 + j, b = a + 4, c = a + 5, d = a + 1;
            tris.insert(tris.end(), {a, b, c, a, c, d});
        }
    FEA2 fea(3500, 0.495f, 0.5f, 0.577350269f, 2);
    if (!fea.Compute(top, tris) || fea.Ksize != 96) { printf("FAIL fea compute %d\n", fea.status()); return 1; }
    std::vector<float> moved = top;
    for (size_t i = 2; i < moved.size(); i += 3) moved[i] += 0.01f;
    fea.ComputeDisplacement(moved);
    fea.ComputeForces();
    const float sE = fea.ComputeStrainEnergy();
    if (!(sE > 0.f) || !(fea.NormalizeStrainEnergy() == sE / 32)) { printf("FAIL energy %g %g\n", sE, fea.NormalizeStrainEnergy()); return 1; }
    std::vector<double> vest(moved.begin(), moved.end());
    const float nsE2 = fea.TrialEnergy(vest);
    if (fea.status() != ORBX_OK || !(nsE2 == fea.NormalizeStrainEnergy()) || !(nsE2 > 0.f)) { printf("FAIL trial energy %g\n", nsE2); return 1; }
    {
        const float rel = (nsE2 - sE / 32) / (sE / 32);
        if (rel > 1e-5f || rel < -1e-5f) { printf("FAIL trial energy differs from the stepwise path %g %g\n", nsE2, sE / 32); return 1; }
    }

    // stereo: right = left shifted by 7 px with its own noise (identical images would give an all-zero
    // correlation distance, a zero median and hence no survivors of the median cut, Frame.cc:687-700)
    std::vector<uint8_t> imgR((size_t)W * H);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            s = s * 1664525u + 1013904223u;
            const int xs = x + 7 < W ? x + 7 : W - 1;
            const int checker = (((xs / 24) + (y / 24)) & 1) ? 200 : 60;
            imgR[(size_t)y * W + x] = (uint8_t)(checker + (int)((s >> 24) % 9) - 4);
        }
    ORBextractor exR(1000, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> kr; std::vector<uint8_t> dr;
    exR(ImageView{imgR.data(), W, H, W}, ImageView{}, kr, dr);
    std::vector<float> uR, depth;
    if (ComputeStereoMatches(ex, exR, 0.5f, 400.f, uR, depth) != ORBX_OK || uR.size() != kps.size()) { printf("FAIL stereo\n"); return 1; }
    int nst = 0;
    for (size_t i = 0; i < uR.size(); ++i) nst += uR[i] >= 0;
    if (nst < 20) { printf("FAIL stereo matches %d\n", nst); return 1; }
    for (size_t i = 0; i < uR.size(); ++i)
        if (uR[i] >= 0 && !(depth[i] > 0.f)) { printf("FAIL stereo depth\n"); return 1; }
    {   // the same stereo frame through ExtractPair, three times (the second call captures each image's launch chain as a graph,
        // the third replays it): same keypoints, descriptors and stereo matches as the two separate calls above
        ORBextractor pl(1000, 1.2f, 8, 20, 7), pr(1000, 1.2f, 8, 20, 7);
        for (int rep = 0; rep < 3; ++rep) {
            std::vector<KeyPoint> kl2, kr2; std::vector<uint8_t> dl2, dr2;
            if (ExtractPair(pl, ImageView{img.data(), W, H, W}, kl2, dl2, pr, ImageView{imgR.data(), W, H, W}, kr2, dr2) != ORBX_OK ||
                kl2.size() != kps.size() || kr2.size() != kr.size() || memcmp(dl2.data(), desc.data(), desc.size()) || memcmp(dr2.data(), dr.data(), dr.size()) ||
                memcmp(kl2.data(), kps.data(), sizeof(KeyPoint) * kps.size())) { printf("FAIL ExtractPair rep %d\n", rep); return 1; }
            std::vector<float> u2, d2;
            if (ComputeStereoMatches(pl, pr, 0.5f, 400.f, u2, d2) != ORBX_OK || u2.size() != uR.size() || memcmp(u2.data(), uR.data(), sizeof(float) * uR.size()) ||
                memcmp(d2.data(), depth.data(), sizeof(float) * depth.size())) { printf("FAIL stereo after ExtractPair rep %d\n", rep); return 1; }
        }
    }
    // whole-loop searches on the frame against itself: every keypoint must find itself
    ORBmatcher::FrameView F;
    F.mvKeysUn = kps.data(); F.mDescriptors = desc.data(); F.N = n; F.mnMinX = 0.f; F.mnMinY = 0.f; F.mnMaxX = (float)W; F.mnMaxY = (float)H;
    {
        std::vector<float> prev((size_t)2 * n);
        int level0 = 0;
        for (int i = 0; i < n; ++i) { prev[2 * i] = kps[i].x; prev[2 * i + 1] = kps[i].y; level0 += kps[i].octave == 0; }
        std::vector<int32_t> m12i;
        ORBmatcher mi(0.9f, true);
        const int ni = mi.SearchForInitialization(F, F, prev, m12i, 100);
        int selfi = 0;
        for (int i = 0; i < n; ++i) selfi += m12i[i] == i;
        if (mi.status() != ORBX_OK || ni != selfi || selfi < level0 / 2) { printf("FAIL SearchForInitialization %d %d %d\n", ni, selfi, level0); return 1; }
        for (int i = 0; i < n; ++i)
            if (kps[i].octave > 0 && m12i[i] >= 0) { printf("FAIL SearchForInitialization level\n"); return 1; }
    }
    {
        std::vector<ORBmatcher::ProjectedPoint> pts(n);
        for (int i = 0; i < n; ++i) {
            pts[i].window = {kps[i].x, kps[i].y, 7.0f * 1.2f, 0.f, kps[i].octave - 1, kps[i].octave + 1};
            pts[i].descriptor = &desc[(size_t)32 * i]; pts[i].angle = kps[i].angle; pts[i].blocksSlot = true;
        }
        std::vector<int32_t> owner;
        const int np = m.SearchByProjection(F, pts, std::vector<uint8_t>(), ORBmatcher::TH_HIGH, false, owner);
        int selfp = 0;
        for (int i = 0; i < n; ++i) selfp += owner[i] == i;
        if (m.status() != ORBX_OK || np != n || selfp != n) { printf("FAIL SearchByProjection %d %d of %d\n", np, selfp, n); return 1; }
    }
    {   // Tracking::TrackWithMotionModel's shape on a frame that stays in HBM: the frame handle is made from the extractor's device
        // results (`ex` last extracted `img`), every keypoint is given a map point on its own ray, identity pose:
        // SearchByProjection(Cur, Last, th, mono) must hand every keypoint its own point, for th and 2 th, and equal the
        // handle made from the host arrays
        ORBmatcher::ResidentFrame cur(ex, nullptr, nullptr, false, 0.f, 0.f, (float)W, (float)H), cur2(F);
        if (cur.status() != ORBX_OK || cur2.status() != ORBX_OK || cur.N != n || cur2.N != n) { printf("FAIL ResidentFrame %d %d\n", cur.status(), cur.N); return 1; }
        ORBmatcher::Calibration K;
        K.fx = 500.f; K.fy = 500.f; K.cx = 320.f; K.cy = 240.f; K.mbf = 40.f; K.mb = 0.08f; K.mfLogScaleFactor = logf(1.2f);
        K.mvScaleFactors.resize(8);
        for (int l = 0; l < 8; ++l) K.mvScaleFactors[l] = l ? K.mvScaleFactors[l - 1] * 1.2f : 1.0f;
        ORBmatcher::PointList last;
        last.resize(n, false, false, true);
        for (int i = 0; i < n; ++i) {
            const float z = 2.0f + (float)(i % 7);
            last.valid[i] = 1; last.octave[i] = kps[i].octave; last.angle[i] = kps[i].angle;
            last.pos[3 * i] = (kps[i].x - K.cx) / K.fx * z; last.pos[3 * i + 1] = (kps[i].y - K.cy) / K.fy * z; last.pos[3 * i + 2] = z;
            std::copy(&desc[(size_t)32 * i], &desc[(size_t)32 * i] + 32, &last.desc[(size_t)32 * i]);
        }
        const float I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<int32_t> ownerA, ownerB;
            const int na = m.SearchByProjection(cur, K, I4, I4, last, std::vector<uint8_t>(), rep ? 14.0f : 7.0f, true, ownerA);
            const int nb2 = m.SearchByProjection(cur2, K, I4, I4, last, std::vector<uint8_t>(), rep ? 14.0f : 7.0f, true, ownerB);
            int selfa = 0;
            for (int i = 0; i < n; ++i) selfa += ownerA[i] == i;
            if (m.status() != ORBX_OK || na != nb2 || ownerA != ownerB || selfa < n * 95 / 100 || na < selfa) { printf("FAIL whole SearchByProjection(Cur, Last) %d %d self %d of %d\n", na, nb2, selfa, n); return 1; }
        }
        // SearchLocalPoints: the same points with ranges and normals that pass isInFrustum
        ORBmatcher::PointList loc = last;
        loc.minDistance.resize(n); loc.maxDistance.resize(n); loc.normal.resize((size_t)3 * n); loc.octave.clear();
        for (int i = 0; i < n; ++i) {
            const float d = sqrtf(loc.pos[3 * i] * loc.pos[3 * i] + loc.pos[3 * i + 1] * loc.pos[3 * i + 1] + loc.pos[3 * i + 2] * loc.pos[3 * i + 2]);
            loc.maxDistance[i] = d * K.mvScaleFactors[kps[i].octave]; loc.minDistance[i] = loc.maxDistance[i] / K.mvScaleFactors[7];
            for (int k = 0; k < 3; ++k) loc.normal[3 * i + k] = loc.pos[3 * i + k] / d;
        }
        std::vector<int32_t> ownerL;
        std::vector<orbm_projected_point> proj;
        ORBmatcher ml(0.8f, true);
        const int nlp = ml.SearchLocalPoints(cur, K, I4, loc, std::vector<uint8_t>(), 1.0f, ownerL, &proj);
        int selfl = 0, vis = 0;
        for (int i = 0; i < n; ++i) { selfl += ownerL[i] == i; vis += proj[i].visible; }
        if (ml.status() != ORBX_OK || vis < n * 9 / 10 || selfl < vis * 8 / 10 || nlp < selfl) { printf("FAIL SearchLocalPoints %d self %d visible %d of %d\n", nlp, selfl, vis, n); return 1; }
    }
    {
        std::vector<int32_t> node(n);
        for (int i = 0; i < n; ++i) node[i] = (desc[(size_t)32 * i] & 31) * 3 + 1;   // a stand-in vocabulary node per feature
        const ORBmatcher::FeatureVector fv = ORBmatcher::FeatureVector::FromNodeIds(node);
        std::vector<uint8_t> valid(n, 1);
        std::vector<int32_t> m12b;
        ORBmatcher mb(0.9f, true);
        const int nb = mb.SearchByBoW(fv, valid, F, fv, nullptr, F, m12b);
        int selfb = 0;
        for (int i = 0; i < n; ++i) selfb += m12b[i] == i;
        if (mb.status() != ORBX_OK || nb != selfb || selfb < n * 8 / 10) { printf("FAIL SearchByBoW %d %d of %d\n", nb, selfb, n); return 1; }
        // the same with both frames resident
        ORBmatcher::ResidentFrame rf(F);
        std::vector<int32_t> m12r;
        const int nbr = mb.SearchByBoW(fv, valid, rf, fv, nullptr, rf, m12r);
        if (rf.status() != ORBX_OK || mb.status() != ORBX_OK || nbr != nb || m12r != m12b) { printf("FAIL SearchByBoW on resident frames %d vs %d\n", nbr, nb); return 1; }
    }
    {
        // SearchForTriangulation of the frame against itself with a skew-symmetric F12 = [t]x (x^T [t]x x = 0: every keypoint
        // lies on its own epipolar line; the epipole t is far outside the image), mono keypoints: candidates = members of the
        // same node, so every keypoint must pair with a keypoint of its own node at distance 0 -- itself, or an identical
        // descriptor on the same line after it (ties: the later candidate wins, :950).
        std::vector<int32_t> node(n);
        for (int i = 0; i < n; ++i) node[i] = (desc[(size_t)32 * i] & 15) + 1;
        const ORBmatcher::FeatureVector fv = ORBmatcher::FeatureVector::FromNodeIds(node);
        std::vector<uint8_t> none(n, 0);
        const float t3[3] = {1.0f, 0.5f, -1.0e-4f};
        const float F12[9] = {0, -t3[2], t3[1], t3[2], 0, -t3[0], -t3[1], t3[0], 0};
        std::vector<float> sf(8), sg(8);
        for (int l = 0; l < 8; ++l) { sf[l] = l ? sf[l - 1] * 1.2f : 1.0f; sg[l] = sf[l] * sf[l]; }
        std::vector<std::pair<size_t, size_t>> pairs;
        ORBmatcher mt(0.6f, false);
        const int nt = mt.SearchForTriangulation(F, fv, none, none, F, fv, none, none, F12, t3[0] / t3[2], t3[1] / t3[2], false, sf, sg, pairs);
        int good = 0;
        for (const auto &pr : pairs) good += node[pr.first] == node[pr.second] && popcount_row(&desc[32 * pr.first], &desc[32 * pr.second]) == 0;
        if (mt.status() != ORBX_OK || nt != n || (int)pairs.size() != n || good != n) { printf("FAIL SearchForTriangulation %d %zu %d of %d\n", nt, pairs.size(), good, n); return 1; }
        ORBmatcher::ResidentFrame rf(F);
        std::vector<std::pair<size_t, size_t>> pairs_r;
        const int ntr = mt.SearchForTriangulation(rf, fv, none, rf, fv, none, F12, t3[0] / t3[2], t3[1] / t3[2], false, sf, sg, pairs_r);
        if (rf.status() != ORBX_OK || mt.status() != ORBX_OK || ntr != nt || pairs_r != pairs) { printf("FAIL SearchForTriangulation on resident frames %d vs %d\n", ntr, nt); return 1; }
        // ComputeDistinctiveDescriptors: three copies of a descriptor and one outlier -> one of the copies (the first)
        std::vector<uint8_t> obs(4 * 32);
        for (int r = 0; r < 4; ++r) memcpy(&obs[32 * r], &desc[(size_t)32 * (r == 1 ? 7 : 3)], 32);
        std::vector<int32_t> offs = {0, 4}, best;
        if (ORBmatcher::ComputeDistinctiveDescriptors(obs.data(), offs, best) != ORBX_OK || best.size() != 1 || best[0] != 0) {
            printf("FAIL ComputeDistinctiveDescriptors %d\n", best.empty() ? -99 : best[0]); return 1;
        }
    }
    {
        // ORBVocabulary: write a k=4, L=2 tree in the ORBvoc.txt layout whose leaves are the first 16 descriptors,
        // load it, transform the frame: every one of those 16 features must land on its own word.
        const char *path = argc > 2 ? argv[2] : "/tmp/dropin_voc.txt";
        FILE *f = fopen(path, "w");
        if (!f || n < 16) { printf("FAIL vocabulary file\n"); return 1; }
        fprintf(f, "4 2  0 0\n");
        for (int c = 0; c < 4; ++c) {                                        // nodes 1..4: inner nodes, descriptor = first leaf below
            fprintf(f, "0 0 ");
            for (int b = 0; b < 32; ++b) fprintf(f, "%d ", desc[(size_t)32 * (4 * c) + b]);
            fprintf(f, "0\n");
        }
        for (int l = 0; l < 16; ++l) {                                       // nodes 5..20: leaves = words 0..15
            fprintf(f, "%d 1 ", 1 + l / 4);
            for (int b = 0; b < 32; ++b) fprintf(f, "%d ", desc[(size_t)32 * l + b]);
            fprintf(f, "%.17g\n", 1.0 + 0.25 * l);
        }
        fclose(f);
        ORBVocabulary voc;
        if (!voc.loadFromTextFile(path) || voc.size() != 16 || voc.getBranchingFactor() != 4 || voc.getDepthLevels() != 2) {
            printf("FAIL loadFromTextFile %d\n", voc.status()); return 1;
        }
        DBoW2::BowVector bv;
        DBoW2::FeatureVector fvm;
        voc.transform(desc.data(), n, bv, fvm, 1);
        double sum = 0; size_t nfeat = 0;
        for (const auto &e : bv) sum += e.second;
        for (const auto &e : fvm) { nfeat += e.second.size(); if (e.first < 1 || e.first > 4) { printf("FAIL node id %u\n", e.first); return 1; } }
        const ORBmatcher::FeatureVector flat = ORBmatcher::FeatureVector::FromMap(fvm);
        if (voc.status() != ORBX_OK || bv.empty() || sum < 1 - 1e-9 || sum > 1 + 1e-9 || (int)nfeat != n || (int)flat.items.size() != n) {
            printf("FAIL transform %d sum %g nfeat %zu\n", voc.status(), sum, nfeat); return 1;
        }
        // the inner nodes carry the descriptor of their first leaf, so feature 4c (distance 0 to node 1+c) reaches word 4c
        std::vector<int32_t> word(n), node(n); std::vector<double> w(n);
        orbm_bow_transform(voc.handle(), desc.data(), n, 1, word.data(), node.data(), w.data());
        for (int c = 0; c < 4; ++c) {
            bool dup = false;                                                // an identical descriptor earlier in the tree wins the tie
            for (int l = 0; l < 4 * c; ++l) dup = dup || popcount_row(&desc[(size_t)32 * l], &desc[(size_t)32 * 4 * c]) == 0;
            if (!dup && (word[4 * c] != 4 * c || node[4 * c] != 1 + c || w[4 * c] != 1.0 + 0.25 * 4 * c)) {
                printf("FAIL descent of feature %d: word %d node %d\n", 4 * c, word[4 * c], node[4 * c]); return 1;
            }
        }
    }
    {
        // ProjectInFrustum + Fuse as a whole through the C++ composition: an identity camera pose looking down +z, one map
        // point per keypoint placed on that keypoint's ray (so point i projects onto keypoint i and carries its descriptor);
        // a toy pointer graph: even slots already hold an older point with more observations (pMP->Replace(pMPinKF)),
        // every fourth of the odd ones a weaker one (pMPinKF->Replace(pMP)), the rest are free (AddObservation).
        ORBmatcher::FrameView F;
        F.mvKeysUn = kps.data(); F.mDescriptors = desc.data(); F.N = n;
        F.mnMinX = 0.f; F.mnMinY = 0.f; F.mnMaxX = (float)W; F.mnMaxY = (float)H;
        std::vector<float> sf(8), isg(8);
        for (int l = 0; l < 8; ++l) { sf[l] = l ? sf[l - 1] * 1.2f : 1.0f; isg[l] = 1.0f / (sf[l] * sf[l]); }
        ORBmatcher::PoseView P;
        const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(P.Rcw, I9, sizeof(I9));
        for (int k = 0; k < 3; ++k) { P.tcw[k] = 0.f; P.Ow[k] = 0.f; }
        P.cam.fx = P.cam.fy = 500.f; P.cam.cx = 320.f; P.cam.cy = 240.f;
        P.cam.min_x = 0; P.cam.min_y = 0; P.cam.max_x = W; P.cam.max_y = H;
        P.cam.grid_min_x = 0.f; P.cam.grid_min_y = 0.f; P.cam.grid_max_x = (float)W; P.cam.grid_max_y = (float)H;
        P.mbf = 40.f; P.mfLogScaleFactor = logf(1.2f); P.mvScaleFactors = sf.data(); P.mvInvLevelSigma2 = isg.data(); P.mnScaleLevels = 8;
        std::vector<float> pos(3 * n), nrm(3 * n), mind(n), maxd(n);
        for (int i = 0; i < n; ++i) {
            const float z = 4.0f, x = (kps[i].x - 320.f) / 500.f * z, y = (kps[i].y - 240.f) / 500.f * z;
            pos[3 * i] = x; pos[3 * i + 1] = y; pos[3 * i + 2] = z;
            const float d = sqrtf(x * x + y * y + z * z);
            nrm[3 * i] = x / d; nrm[3 * i + 1] = y / d; nrm[3 * i + 2] = z / d;    // PO.Pn = dist >= 0.5 dist
            maxd[i] = d * powf(1.2f, (float)kps[i].octave) * 0.9999f;              // PredictScale = ceil(log(ratio) / log 1.2) -> the keypoint's octave
            mind[i] = maxd[i] / 5.0f;
        }
        ORBmatcher::MapPointArrays mps;
        mps.pos = pos.data(); mps.normal = nrm.data(); mps.mfMinDistance = mind.data(); mps.mfMaxDistance = maxd.data();
        mps.descriptors = desc.data(); mps.n = n;
        ORBmatcher mf(0.6f, true);
        std::vector<orbm_projected_point> proj;
        std::vector<orbm_window_query> win;
        if (mf.ProjectInFrustum(P, mps, 0.5f, 1.0f, proj, win) != ORBX_OK) { printf("FAIL ProjectInFrustum %d\n", mf.status()); return 1; }
        int vis = 0, lev = 0;
        for (int i = 0; i < n; ++i) {
            vis += proj[i].visible;
            lev += proj[i].visible && proj[i].level == kps[i].octave && fabsf(proj[i].u - kps[i].x) < 1e-2f && fabsf(proj[i].v - kps[i].y) < 1e-2f;
        }
        if (vis < n * 9 / 10 || lev < vis * 9 / 10) { printf("FAIL ProjectInFrustum visible %d level/pixel ok %d of %d\n", vis, lev, n); return 1; }
        struct Toy {
            int n;
            std::vector<int> slot, obs, inKF; std::vector<char> bad;     // handles: 0..n-1 listed points, n..2n-1 the key frame's own
            int adds = 0, dies = 0, takes = 0;
            typedef int H;
            H at(int i) const { return i; }
            bool null(H h) const { return h < 0; }
            bool isBad(H h) const { return bad[h] != 0; }
            bool isInKeyFrame(H h) const { return inKF[h] >= 0; }
            H slotOwner(int idx) const { return slot[idx]; }
            int observations(H h) const { return obs[h]; }
            void replace(H dead, H heir) { bad[dead] = 1; if (inKF[dead] >= 0) { slot[inKF[dead]] = heir; inKF[heir] = inKF[dead]; inKF[dead] = -1; ++takes; } else ++dies; }
            void add(H h, int idx) { slot[idx] = h; inKF[h] = idx; ++adds; }
        } toy;
        toy.n = n; toy.slot.assign(n, -1); toy.obs.assign(2 * n, 2); toy.inKF.assign(2 * n, -1); toy.bad.assign(2 * n, 0);
        for (int j = 0; j < n; ++j) {
            if (j % 2 == 0) { toy.slot[j] = n + j; toy.inKF[n + j] = j; toy.obs[n + j] = 5; }
            else if (j % 4 == 1) { toy.slot[j] = n + j; toy.inKF[n + j] = j; toy.obs[n + j] = 1; }
        }
        const int nFused = mf.Fuse(F, P, mps, 3.0f, toy);
        if (mf.status() != ORBX_OK || nFused < vis * 8 / 10 || nFused != toy.adds + toy.dies + toy.takes || toy.adds == 0 || toy.dies == 0 || toy.takes == 0) {
            printf("FAIL Fuse %d: nFused %d adds %d dies %d takes %d\n", mf.status(), nFused, toy.adds, toy.dies, toy.takes); return 1;
        }
    }
    printf("OK %d keypoints, %d self matches, %d stereo matches, sE=%g\n", n, nm, nst, sE);
    return 0;
}
