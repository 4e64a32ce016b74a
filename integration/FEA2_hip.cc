// FEA2_hip.cc -- the numeric half of FEA2 (Thirdparty/g2o/g2o/FEA/src/FEA2.cc) over liborbslam_hip.so.  Compiled into libg2o.so and
// libORB_SLAM2_E.so next to FEA2.cc, whose own definitions of these methods are compiled out by integration/reference.patch
// (FEA2_NUMERIC_ON_HIP).  The PCL half (MLS, greedy projection, tri2quad, SetSecondLayer, Set_u0; FEA2.cc:124-1241) is untouched: it
// produces the mesh and u0, this file turns them into K on the device and answers the per-trial strain-energy query of the LM hook
// (optimization_algorithm_levenberg.cpp:159-175) without K ever visiting the host.
//   FEA2.h gains one member: `struct fem_model *mFem = nullptr;` (and `std::vector<double> mTrialPts;`).
#include "FEA2.h"

#include "fem_hip.h"

// MatrixAssemblyC3D8 / C3D6 (FEA2.cc:1379-1502, :1505-1624), nMode 1: element nodes = top ids || top ids + nTop (:1392-1399, :1518-1523);
// K = sum of K_e scattered by 3 x 3 node blocks, in the reference's summation order, kept as CSR in HBM.
static bool assemble_on_device(FEA2 *fea, int eltype, const std::vector<std::vector<int> > &faces, int npf)
{
    const int nTop = (int)fea->vMPsXYZN_t.size(), nn = nTop + (int)fea->vMPsXYZN_t2.size();
    fea->Ksize = 3 * nn;
    if (fea->Ksize <= 3) return false;                                           // :1385-1386
    std::vector<float> nodes(3 * (size_t)nn);
    for (int i = 0; i < nTop; ++i) for (int k = 0; k < 3; ++k) nodes[3 * i + k] = fea->vMPsXYZN_t[i][k];
    for (int i = nTop; i < nn; ++i) for (int k = 0; k < 3; ++k) nodes[3 * i + k] = fea->vMPsXYZN_t2[i - nTop][k];
    std::vector<int32_t> elems;
    elems.reserve(2 * (size_t)npf * faces.size());
    for (size_t e = 0; e < faces.size(); ++e) {
        for (int k = 0; k < npf; ++k) elems.push_back(faces[e][k]);
        for (int k = 0; k < npf; ++k) elems.push_back(faces[e][k] + nTop);
    }
    if (fea->mFem) { fem_destroy(fea->mFem); fea->mFem = nullptr; }
    if (fem_create(eltype, nodes.data(), 1, nn, elems.data(), (int)faces.size(), fea->E, fea->nu, fea->fg, &fea->mFem) != 0) return false;
    return fem_assemble(fea->mFem) == 0;
}

bool FEA2::MatrixAssemblyC3D8(int nMode)
{
    if (nMode != 1) return false;              // nMode 2 (the untracked layer) feeds the dead inverse path only (FEA2.cc:110-113)
    return assemble_on_device(this, FEM_C3D8, quads_t, 4);
}

bool FEA2::MatrixAssemblyC3D6(int nMode)
{
    if (nMode != 1) return false;
    return assemble_on_device(this, FEM_C3D6, triangles_t, 3);
}

// ImposeDirichletEncastre_K (FEA2.cc:1628-1645): K[d][d] = Klarge for d in 3 (id - 1) + {0, 1, 2} -- the reference's off-by-one kept --
// and, since this is the last step of Compute(1) (:104-108), everything the per-trial hook needs is parked on the device here.
void FEA2::ImposeDirichletEncastre_K(int nMode, vector<vector<int> > vD, float Klarge)
{
    if (nMode != 1 || !mFem) return;
    std::vector<int32_t> ids(vD.size());
    for (size_t i = 0; i < vD.size(); ++i) ids[i] = vD[i][0];
    fem_dirichlet_penalty(mFem, ids.data(), (int)ids.size(), Klarge);
    std::vector<int32_t> derived(4 * vNewPointsBase.size(), 0);                   // mid-edge / barycentre nodes Set_uf recomputes (:1746-1775)
    for (size_t i = 0; i < vNewPointsBase.size(); ++i) {
        derived[4 * i] = (int32_t)vNewPointsBase[i].size();
        for (size_t k = 0; k < vNewPointsBase[i].size() && k < 3; ++k) derived[4 * i + 1 + k] = vNewPointsBase[i][k];
    }
    const int npoints = (int)vMPsXYZN_t.size() - (int)vNewPointsBase.size();     // the optimiser's point vertices
    fem_trial_setup(mFem, u0.data(), ids.data(), (int)ids.size(), Klarge, npoints, derived.data(), (int)vNewPointsBase.size());
}

void FEA2::ImposeDirichletEncastre_a(vector<vector<int> >, float) {}            // folded into the device-side displacement (:1648-1658)

// The hook's five calls (levenberg.cpp:164-171).  Set_uf keeps the trial's point estimates; the energy query does Set_uf's node
// rebuild, ComputeDisplacement, ComputeForces, ComputeStrainEnergy and NormalizeStrainEnergy in two launches on the resident K.
void FEA2::Set_uf(vector<vector<float> > vPoints)
{
    mTrialPts.resize(3 * vPoints.size());
    for (size_t i = 0; i < vPoints.size(); ++i) for (int k = 0; k < 3; ++k) mTrialPts[3 * i + k] = vPoints[i][k];   // float values, exactly
    mTrialDone = false;
}

static void trial(FEA2 *fea)
{
    if (fea->mTrialDone || !fea->mFem) return;
    fea->sE = fea->nsE = 0.0f;
    std::vector<float> a;
    if (fea->bDebugMode) a.resize(fea->Ksize);
    fem_trial_energy(fea->mFem, fea->mTrialPts.data(), fea->bDebugMode ? a.data() : nullptr, &fea->sE, &fea->nsE);
    if (fea->bDebugMode) { fea->vva.assign(fea->Ksize, vector<float>(1)); for (unsigned i = 0; i < fea->Ksize; ++i) fea->vva[i][0] = a[i]; }
    fea->CurrentSE = fea->sE;
    fea->mTrialDone = true;
}

void FEA2::ComputeDisplacement() { trial(this); }                               // FEA2.cc:1799-1808
void FEA2::ComputeForces() { trial(this); }                                     // :1811-1816 (vvf is no longer materialised)
float FEA2::ComputeStrainEnergy() { trial(this); return sE; }                   // :1877-1894
float FEA2::NormalizeStrainEnergy() { trial(this); return nsE; }                // :1897-1902
