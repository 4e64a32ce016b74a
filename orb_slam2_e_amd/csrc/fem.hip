// fem.hip -- FEM stiffness assembly, K*a, strain energy and CG for gfx950.
//
// Replaces the numeric core of FEA2 (Thirdparty/g2o/g2o/FEA/src/FEA2.cc):
//   k_fem_ke        ComputeKeiC3D8 / ComputeKeiC3D6 (:1244-1376) + a linear tet
//   k_fem_assemble_rows  MatrixAssemblyC3D8/6 (:1379-1624) into CSR with K_e formed on the chip, gather form:
//                   every matrix entry sums its element contributions in element order,
//                   so no atomics and the result equals the dense scatter-add
//                   (k_fem_assemble_fused: entry by entry, for a D that is not isotropic;
//                   k_fem_assemble: from K_e in HBM, when a node's elements do not fit LDS)
//   k_fem_matvec    ComputeForces f = K*a (:1811-1816), float, row order
//   k_fem_energy    ComputeStrainEnergy |a^T f| (:1877-1894)
//   k_fem_spmv / k_fem_cg_update / k_fem_cg_dir: Jacobi-PCG on the resident
//                   block-diagonal batch (the slot of the dead dense inverse, :1661-1691)
//
// HBM layout for a batch of M meshes with n dofs and nnz non-zeros each:
// vals[M][nnz] f32 per mesh; ONE column-index array lcol[nnz] i32 and ONE rowptr[n+1] for the
// whole batch (shared topology: they stay in L2); CG vectors x, r, p, Ap [M][n] f64.  SpMV streams
// the values from HBM once per iteration; reductions are fixed-order (chunk partials
// summed in index order), so results are run-to-run reproducible.
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/fem_hip.h"
#include "common.h"

namespace {

constexpr int RPB = 256; // rows per block in the CG vector kernels
constexpr int LPR = 8;   // lanes per row in the SpMV
constexpr int CGT = 256; // threads per block in the CG kernels

struct FemConst {
    float D[36];
    float gs[24];
};

__device__ __forceinline__ float bmat(int m, int comp, float gx, float gy, float gz)
{
    // B rows as laid out at FEA2.cc:1288-1293 for one node's 3 columns
    switch (m) {
    case 0: return comp == 0 ? gx : 0.0f;
    case 1: return comp == 1 ? gy : 0.0f;
    case 2: return comp == 2 ? gz : 0.0f;
    case 3: return comp == 0 ? gy : (comp == 1 ? gx : 0.0f);
    case 4: return comp == 0 ? gz : (comp == 2 ? gx : 0.0f);
    default: return comp == 1 ? gz : (comp == 2 ? gy : 0.0f);
    }
}

// One Gauss point of one element (one tetrahedron: its single point): Jacobian, the reference's "inverse", physical
// shape gradients G[3][NPE] and the weight JAC, in the literal float operation order of ComputeKeiC3D8 / ComputeKeiC3D6
// (FEA2.cc:1254-1277, :1322-1345).  P = the element's node coordinates [NPE][3].  GS = row stride of G.
template <int NPE, int ELT, int GS>
__device__ __forceinline__ void fem_gauss_point(const float *P, int gp, const FemConst &fc, float *G, float &jac)
{
    if constexpr (ELT == FEM_TET4) {
        const float e1x = P[3] - P[0], e1y = P[4] - P[1], e1z = P[5] - P[2];
        const float e2x = P[6] - P[0], e2y = P[7] - P[1], e2z = P[8] - P[2];
        const float e3x = P[9] - P[0], e3y = P[10] - P[1], e3z = P[11] - P[2];
        const float c1x = e2y * e3z - e2z * e3y, c1y = e2z * e3x - e2x * e3z, c1z = e2x * e3y - e2y * e3x;
        const float c2x = e3y * e1z - e3z * e1y, c2y = e3z * e1x - e3x * e1z, c2z = e3x * e1y - e3y * e1x;
        const float c3x = e1y * e2z - e1z * e2y, c3y = e1z * e2x - e1x * e2z, c3z = e1x * e2y - e1y * e2x;
        const float det = e1x * c1x + e1y * c1y + e1z * c1z;
        float gx[4], gy[4], gz[4];
        gx[1] = c1x / det; gy[1] = c1y / det; gz[1] = c1z / det;
        gx[2] = c2x / det; gy[2] = c2y / det; gz[2] = c2z / det;
        gx[3] = c3x / det; gy[3] = c3y / det; gz[3] = c3z / det;
        gx[0] = -(gx[1] + gx[2] + gx[3]); gy[0] = -(gy[1] + gy[2] + gy[3]); gz[0] = -(gz[1] + gz[2] + gz[3]);
#pragma unroll
        for (int n = 0; n < 4; ++n) { G[n % NPE] = gx[n]; G[GS + n % NPE] = gy[n]; G[2 * GS + n % NPE] = gz[n]; }
        jac = fabsf(det) / 6;
    } else {
        const float xi = fc.gs[3 * gp], eta = fc.gs[3 * gp + 1], zeta = fc.gs[3 * gp + 2];
        float a[NPE], b[NPE], c[NPE];
        if constexpr (ELT == FEM_C3D8) { // FEA2.cc:1254-1261
            a[0 % NPE] = -0.125f * ((1 - eta) * (1 - zeta)); b[0 % NPE] = -0.125f * ((1 - xi) * (1 - zeta)); c[0 % NPE] = -0.125f * ((1 - xi) * (1 - eta));
            a[1 % NPE] = +0.125f * ((1 - eta) * (1 - zeta)); b[1 % NPE] = -0.125f * ((1 + xi) * (1 - zeta)); c[1 % NPE] = -0.125f * ((1 + xi) * (1 - eta));
            a[2 % NPE] = +0.125f * ((1 + eta) * (1 - zeta)); b[2 % NPE] = +0.125f * ((1 + xi) * (1 - zeta)); c[2 % NPE] = -0.125f * ((1 + xi) * (1 + eta));
            a[3 % NPE] = -0.125f * ((1 + eta) * (1 - zeta)); b[3 % NPE] = +0.125f * ((1 - xi) * (1 - zeta)); c[3 % NPE] = -0.125f * ((1 - xi) * (1 + eta));
            a[4 % NPE] = -0.125f * ((1 - eta) * (1 + zeta)); b[4 % NPE] = -0.125f * ((1 - xi) * (1 + zeta)); c[4 % NPE] = +0.125f * ((1 - xi) * (1 - eta));
            a[5 % NPE] = +0.125f * ((1 - eta) * (1 + zeta)); b[5 % NPE] = -0.125f * ((1 + xi) * (1 + zeta)); c[5 % NPE] = +0.125f * ((1 + xi) * (1 - eta));
            a[6 % NPE] = +0.125f * ((1 + eta) * (1 + zeta)); b[6 % NPE] = +0.125f * ((1 + xi) * (1 + zeta)); c[6 % NPE] = +0.125f * ((1 + xi) * (1 + eta));
            a[7 % NPE] = -0.125f * ((1 + eta) * (1 + zeta)); b[7 % NPE] = +0.125f * ((1 - xi) * (1 + zeta)); c[7 % NPE] = +0.125f * ((1 - xi) * (1 + eta));
        } else { // C3D6, FEA2.cc:1322-1327
            a[0] = -(1 + zeta) / 2; b[0] = -(1 + zeta) / 2; c[0] = (1 - xi - eta) / 2;
            a[1] = (1 + zeta) / 2;  b[1] = 0.0f;            c[1] = xi / 2;
            a[2] = 0.0f;            b[2] = (1 + zeta) / 2;  c[2] = eta / 2;
            a[3] = -(1 - zeta) / 2; b[3] = -(1 - zeta) / 2; c[3] = -(1 - xi - eta) / 2;
            a[4 % NPE] = (1 - zeta) / 2;  b[4 % NPE] = 0.0f;            c[4 % NPE] = -xi / 2;
            a[5 % NPE] = 0.0f;            b[5 % NPE] = (1 - zeta) / 2;  c[5 % NPE] = -eta / 2;
        }
        float J_00 = a[0] * P[0], J_01 = a[0] * P[1], J_02 = a[0] * P[2];
        float J_10 = b[0] * P[0], J_11 = b[0] * P[1], J_12 = b[0] * P[2];
        float J_20 = c[0] * P[0], J_21 = c[0] * P[1], J_22 = c[0] * P[2];
#pragma unroll
        for (int n = 1; n < NPE; ++n) {
            J_00 = J_00 + a[n] * P[3 * n]; J_01 = J_01 + a[n] * P[3 * n + 1]; J_02 = J_02 + a[n] * P[3 * n + 2];
            J_10 = J_10 + b[n] * P[3 * n]; J_11 = J_11 + b[n] * P[3 * n + 1]; J_12 = J_12 + b[n] * P[3 * n + 2];
            J_20 = J_20 + c[n] * P[3 * n]; J_21 = J_21 + c[n] * P[3 * n + 1]; J_22 = J_22 + c[n] * P[3 * n + 2];
        }
        // signed determinant and the reference's inverse with its three sign deviations
        const float Jac = J_00 * J_11 * J_22 + J_01 * J_12 * J_20 + J_10 * J_21 * J_02 - J_20 * J_11 * J_02 - J_10 * J_01 * J_22 - J_21 * J_12 * J_00;
        const float J1_00 = (+1) * ((J_11 * J_22) - (J_21 * J_12)) / Jac, J1_01 = (-1) * ((J_01 * J_22) - (J_21 * J_02)) / Jac, J1_02 = (-1) * ((J_01 * J_12) - (J_11 * J_02)) / Jac;
        const float J1_10 = (-1) * ((J_10 * J_22) - (J_20 * J_12)) / Jac, J1_11 = (-1) * ((J_00 * J_22) - (J_20 * J_02)) / Jac, J1_12 = (-1) * ((J_00 * J_12) - (J_10 * J_02)) / Jac;
        const float J1_20 = (+1) * ((J_10 * J_21) - (J_20 * J_11)) / Jac, J1_21 = (-1) * ((J_00 * J_21) - (J_20 * J_01)) / Jac, J1_22 = (-1) * ((J_00 * J_11) - (J_10 * J_01)) / Jac;
#pragma unroll
        for (int n = 0; n < NPE; ++n) {
            G[n] = J1_00 * a[n] + J1_01 * b[n] + J1_02 * c[n];
            G[GS + n] = J1_10 * a[n] + J1_11 * b[n] + J1_12 * c[n];
            G[2 * GS + n] = J1_20 * a[n] + J1_21 * b[n] + J1_22 * c[n];
        }
        jac = Jac;
    }
}

// (B^T D B)[i][j] at ONE Gauss point, the literal six-term chains of FEA2.cc:1295-1303.  g: that point's gradients [3][NPE].
template <int NPE>
__device__ __forceinline__ float fem_ke_gp(const float *g, const FemConst &fc, int ni, int ci, int nj, int cj)
{
    const float gxi = g[ni], gyi = g[NPE + ni], gzi = g[2 * NPE + ni];
    const float gxj = g[nj], gyj = g[NPE + nj], gzj = g[2 * NPE + nj];
    float Bi[6], Bj[6], BtD[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) { Bi[m] = bmat(m, ci, gxi, gyi, gzi); Bj[m] = bmat(m, cj, gxj, gyj, gzj); }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        BtD[k] = Bi[0] * fc.D[k] + Bi[1] * fc.D[6 + k] + Bi[2] * fc.D[12 + k] + Bi[3] * fc.D[18 + k] + Bi[4] * fc.D[24 + k] + Bi[5] * fc.D[30 + k];
    return BtD[0] * Bj[0] + BtD[1] * Bj[1] + BtD[2] * Bj[2] + BtD[3] * Bj[3] + BtD[4] * Bj[4] + BtD[5] * Bj[5];
}

// K_e[i][j] = sum over the Gauss points of (B^T D B)[i][j] * Jac, accumulated in Gauss-point order (FEA2.cc:1295-1306).
// G: [NGP][3][NPE] (one element), JAC: [NGP].
template <int NPE, int NGP>
__device__ __forceinline__ float fem_ke_entry(const float *G, const float *JAC, const FemConst &fc, int i, int j)
{
    const int ni = i / 3, ci = i - 3 * ni, nj = j / 3, cj = j - 3 * nj;
    float acc = 0.0f;
    for (int gp = 0; gp < NGP; ++gp) acc += fem_ke_gp<NPE>(G + gp * 3 * NPE, fc, ni, ci, nj, cj) * JAC[gp];
    return acc;
}

// The nine (B_i^T D B_j)[m][n] of a node pair at one Gauss point with only the terms that are not structurally zero: B has
// three non-zeros per column and D (isotropic: FEA2.cc:56-62) a dense 3 x 3 corner and a diagonal, so of the literal 72
// products per entry 2 to 3 pairs survive.  The sums keep the literal order (k ascending), and what is left out is exact:
// every dropped product has a structural +0 factor and -- g, h and g D finite, which the caller guarantees -- is +-0, and
// x + (+-0) = x for every x != 0; where a whole chain is zero only the SIGN of that zero could differ, and the accumulations
// this feeds start from +0.0f, (+0) + (+-0) = +0.  (g = node i's gradient, h = node j's.)
__device__ __forceinline__ void fem_block_gp(float gx, float gy, float gz, float hx, float hy, float hz, const FemConst &fc, float aux[9])
{
    const float a0 = gx * fc.D[0], a1 = gx * fc.D[1], a2 = gx * fc.D[2], a3 = gy * fc.D[21], a4 = gz * fc.D[28];      // B_i column 0: [gx 0 0 gy gz 0]
    aux[0] = a0 * hx + a3 * hy + a4 * hz;
    aux[1] = a1 * hy + a3 * hx;
    aux[2] = a2 * hz + a4 * hx;
    const float b0 = gy * fc.D[6], b1 = gy * fc.D[7], b2 = gy * fc.D[8], b3 = gx * fc.D[21], b5 = gz * fc.D[35];      // column 1: [0 gy 0 gx 0 gz]
    aux[3] = b0 * hx + b3 * hy;
    aux[4] = b1 * hy + b3 * hx + b5 * hz;
    aux[5] = b2 * hz + b5 * hy;
    const float c0 = gz * fc.D[12], c1 = gz * fc.D[13], c2 = gz * fc.D[14], c4 = gx * fc.D[28], c5 = gy * fc.D[35];   // column 2: [0 0 gz 0 gx gy]
    aux[6] = c0 * hx + c4 * hz;
    aux[7] = c1 * hy + c5 * hz;
    aux[8] = c2 * hz + c4 * hx + c5 * hy;
}

// One 64-lane workgroup per element (the K_e accessor and the two-kernel assembly): lanes 0..NGP-1 evaluate one Gauss
// point each, then all lanes form the entries of K_e.
template <int NPE, int ELT>
__global__ __launch_bounds__(64) void k_fem_ke(const float *__restrict__ nodes, int nn, const int *__restrict__ elems,
                                               int ne, FemConst fc, float *__restrict__ ke_all, int e0, int mesh0)
{
    constexpr int ND = 3 * NPE;
    constexpr int NGP = ELT == FEM_TET4 ? 1 : 8;
    __shared__ float P[NPE * 3];
    __shared__ float G[NGP * 3 * NPE];
    __shared__ float JAC[NGP];
    const int e = e0 + blockIdx.x, mesh = mesh0 + blockIdx.y, lane = threadIdx.x; // K_e lands at ke_all[blockIdx.y][blockIdx.x]
    if (lane < NPE * 3) {
        const int node = elems[e * NPE + lane / 3];
        P[lane] = nodes[((size_t)mesh * nn + node) * 3 + lane % 3];
    }
    __syncthreads();
    if (lane < NGP) {
        float jac;
        fem_gauss_point<NPE, ELT, NPE>(P, lane, fc, G + lane * 3 * NPE, jac);
        JAC[lane] = jac;
    }
    __syncthreads();
    float *ke = ke_all + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * ND * ND;
    for (int idx = lane; idx < ND * ND; idx += 64) {
        const int i = idx / ND, j = idx - i * ND;
        ke[idx] = fem_ke_entry<NPE, NGP>(G, JAC, fc, i, j);
    }
}

// MatrixAssemblyC3D8 / C3D6 (FEA2.cc:1379-1624) with K_e never leaving the chip: one workgroup per node I = block row.
// Phase 1: Gauss-point data (shape gradients, Jac) of every element that holds node I, one (element, Gauss point) per
// thread, into LDS.  Phase 2: one thread per scalar of the row's 3x3 blocks walks its contribution list
// (ascending (element, li, lj) = the reference's scatter order), forms each K_e entry from the cached gradients
// exactly as k_fem_ke does and adds it up.  HBM sees the node coordinates, the lists and ONE write of the values
// (instead of a write and a read of every K_e: 2.2x the algorithmic traffic).
template <int NPE, int ELT>
__global__ __launch_bounds__(256) void k_fem_assemble_fused(const float *__restrict__ nodes, int nn, const int *__restrict__ elems,
                                                            FemConst fc, const int *__restrict__ bptr, const int *__restrict__ cptr,
                                                            const int *__restrict__ contrib_loc, const int *__restrict__ nel_ptr,
                                                            const int *__restrict__ nel, const int *__restrict__ rowptr,
                                                            float *__restrict__ vals, size_t nnz)
{
    constexpr int NGP = ELT == FEM_TET4 ? 1 : 8;
    constexpr int GSZ = NGP * 3 * NPE;                 // floats of gradients per element
    extern __shared__ float s_fem[];                   // [nelI][GSZ] gradients, then [nelI][NGP] weights
    const int I = blockIdx.x, mesh = blockIdx.y, tid = threadIdx.x;
    const int e0 = nel_ptr[I], nelI = nel_ptr[I + 1] - e0;
    float *G = s_fem, *JAC = s_fem + (size_t)nelI * GSZ;
    for (int t = tid; t < nelI * NGP; t += 256) {
        const int el = t / NGP, gp = t - el * NGP, e = nel[e0 + el];
        float P[NPE * 3];
#pragma unroll
        for (int n = 0; n < NPE; ++n) {
            const float *q = nodes + ((size_t)mesh * nn + elems[e * NPE + n]) * 3;
            P[3 * n] = q[0]; P[3 * n + 1] = q[1]; P[3 * n + 2] = q[2];
        }
        float jac;
        fem_gauss_point<NPE, ELT, NPE>(P, gp, fc, G + (size_t)el * GSZ + gp * 3 * NPE, jac);
        JAC[el * NGP + gp] = jac;
    }
    __syncthreads();
    const int b0 = bptr[I], nb = bptr[I + 1] - b0;
    for (int t = tid; t < nb * 9; t += 256) {
        const int bl = t / 9, mn = t - 9 * bl, m = mn / 3, n = mn - 3 * m, b = b0 + bl;
        float v = 0.0f;
        for (int c = cptr[b]; c < cptr[b + 1]; ++c) {
            const int pk = contrib_loc[c];
            const int el = pk >> 6, li = (pk >> 3) & 7, lj = pk & 7;
            v += fem_ke_entry<NPE, NGP>(G + (size_t)el * GSZ, JAC + el * NGP, fc, 3 * li + m, 3 * lj + n);
        }
        vals[(size_t)mesh * nnz + rowptr[3 * I + m] + 3 * bl + n] = v;
    }
}

// The same assembly with the work of a block row shared (the default; k_fem_assemble_fused is kept for a D that is not
// isotropic).  k_fem_assemble_fused forms every K_e entry by itself -- two B columns picked with selects, the 6 x 6 product
// with D, ~360 instructions per entry, 2.2 G wave-instructions per 256 config-3 meshes.  Here a thread takes one
// CONTRIBUTION (element, local i, local j) of the row and forms its nine entries together from the six gradient components
// (fem_block_gp: 66 flops per Gauss point for all nine), parks them in LDS, and then a thread per scalar of the row's blocks
// adds its contributions up in the reference's scatter order, exactly as before.  An element whose gradients are not finite
// or large enough for g D to overflow (the reference's degenerate prisms: NaN / Inf patterns must come out the same) takes
// the literal chains (fem_ke_gp) instead, flagged per (element, Gauss point) in phase 1.
#ifndef FEM_ROWS_T
#define FEM_ROWS_T 64
#endif
constexpr int ROWS_T = FEM_ROWS_T;
template <int NPE, int ELT>
__global__ __launch_bounds__(ROWS_T) void k_fem_assemble_rows(const float *__restrict__ nodes, int nn, const int *__restrict__ elems,
                                                           FemConst fc, float glimit, const int *__restrict__ bptr,
                                                           const int *__restrict__ cptr, const int *__restrict__ contrib_loc,
                                                           const int *__restrict__ nel_ptr, const int *__restrict__ nel,
                                                           const int *__restrict__ rowptr, float *__restrict__ vals, size_t nnz)
{
    constexpr int NGP = ELT == FEM_TET4 ? 1 : 8;
    constexpr int GSZ = NGP * 3 * NPE;
    extern __shared__ float s_fem[];                   // [nelI][GSZ] gradients, [nelI][NGP] weights, [nelI][NGP] flags, [ncI][9] entries
    const int I = blockIdx.x, mesh = blockIdx.y, tid = threadIdx.x;
    const int e0 = nel_ptr[I], nelI = nel_ptr[I + 1] - e0;
    float *G = s_fem, *JAC = s_fem + (size_t)nelI * GSZ, *SAFE = JAC + nelI * NGP, *CV = SAFE + nelI * NGP;
    for (int t = tid; t < nelI * NGP; t += ROWS_T) {
        const int el = t / NGP, gp = t - el * NGP, e = nel[e0 + el];
        float P[NPE * 3];
#pragma unroll
        for (int n = 0; n < NPE; ++n) {
            const float *q = nodes + ((size_t)mesh * nn + elems[e * NPE + n]) * 3;
            P[3 * n] = q[0]; P[3 * n + 1] = q[1]; P[3 * n + 2] = q[2];
        }
        float jac, g[3 * NPE];
        fem_gauss_point<NPE, ELT, NPE>(P, gp, fc, g, jac);
        bool safe = true;
#pragma unroll
        for (int k = 0; k < 3 * NPE; ++k) { G[(size_t)el * GSZ + gp * 3 * NPE + k] = g[k]; safe = safe && fabsf(g[k]) < glimit; }
        JAC[t] = jac;
        SAFE[t] = safe ? 1.0f : 0.0f;
    }
    __syncthreads();
    const int b0 = bptr[I], nb = bptr[I + 1] - b0, cbase = cptr[b0], ncI = cptr[b0 + nb] - cbase;
    for (int t = tid; t < ncI; t += ROWS_T) {
        const int pk = contrib_loc[cbase + t];
        const int el = pk >> 6, li = (pk >> 3) & 7, lj = pk & 7;
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0f;
        for (int gp = 0; gp < NGP; ++gp) {
            const float *g = G + (size_t)el * GSZ + gp * 3 * NPE;
            const float w = JAC[el * NGP + gp];
            float aux[9];
            if (SAFE[el * NGP + gp] != 0.0f) {
                fem_block_gp(g[li], g[NPE + li], g[2 * NPE + li], g[lj], g[NPE + lj], g[2 * NPE + lj], fc, aux);
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) aux[k] = fem_ke_gp<NPE>(g, fc, li, k / 3, lj, k % 3);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[k] += aux[k] * w;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) CV[t * 9 + k] = acc[k];
    }
    __syncthreads();
    for (int t = tid; t < nb * 9; t += ROWS_T) {
        const int bl = t / 9, mn = t - 9 * bl, m = mn / 3, n = mn - 3 * m, b = b0 + bl;
        float v = 0.0f;
        for (int c = cptr[b] - cbase; c < cptr[b + 1] - cbase; ++c) v += CV[c * 9 + mn];
        vals[(size_t)mesh * nnz + rowptr[3 * I + m] + 3 * bl + n] = v;
    }
}

// One thread per scalar entry of a 3x3 block: sums the block's element
// contributions in (element, local i, local j) order = the reference's loop order.
__global__ __launch_bounds__(256) void k_fem_assemble(const float *__restrict__ ke_all, int ne, int nd, int nblk,
                                                      const int *__restrict__ blk_row, const int *__restrict__ bptr,
                                                      const int *__restrict__ cptr, const int *__restrict__ contrib,
                                                      const int *__restrict__ rowptr, float *__restrict__ vals, size_t nnz)
{
    const int t = blockIdx.x * 256 + threadIdx.x, mesh = blockIdx.y;
    if (t >= nblk * 9) return;
    const int b = t / 9, mn = t - 9 * b, m = mn / 3, n = mn - 3 * m;
    const int I = blk_row[b];
    const float *ke = ke_all + (size_t)mesh * ne * nd * nd;
    float v = 0.0f;
    for (int c = cptr[b]; c < cptr[b + 1]; ++c) {
        const int pk = contrib[c];
        const int e = pk >> 6, li = (pk >> 3) & 7, lj = pk & 7;
        v += ke[((size_t)e * nd + 3 * li + m) * nd + 3 * lj + n];
    }
    const int k = rowptr[3 * I + m] + 3 * (b - bptr[I]) + n;
    vals[(size_t)mesh * nnz + k] = v;
}

__global__ void k_fem_penalty(float *__restrict__ vals, size_t nnz, const int *__restrict__ diag_idx,
                              const int *__restrict__ ids, int nids, float klarge)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, mesh = blockIdx.y;
    if (t >= nids * 3) return;
    const int d = 3 * (ids[t / 3] - 1) + t % 3; // FEA2.cc:1630: mp0 = 3*(vD[i][0] - 1)
    vals[(size_t)mesh * nnz + diag_idx[d]] = klarge;
}

// One thread per 3 x 3 block q (block row I = blk_row[q], first column bcol3[q]): a block none of whose three rows and
// three columns is fixed -- nearly all of them -- costs two table reads and six flag bytes and leaves the values alone.
// (One thread per ROW walking its ~40 entries read a column index per non-zero at a 160-byte stride: 0.49 ms per 256 config-3
// meshes, now 0.05.)
__global__ __launch_bounds__(256) void k_fem_eliminate(float *__restrict__ vals, const int *__restrict__ blk_row,
                                                       const int *__restrict__ bcol3, const int *__restrict__ bp,
                                                       const int *__restrict__ rowptr, int nblk, size_t nnz,
                                                       const uint8_t *__restrict__ fixed)
{
    const int q = blockIdx.x * 256 + threadIdx.x, mesh = blockIdx.y;
    if (q >= nblk) return;
    const int I = blk_row[q], c0 = bcol3[q];
    const unsigned fr = fixed[3 * I] | (fixed[3 * I + 1] << 1) | (fixed[3 * I + 2] << 2);
    const unsigned fc = fixed[c0] | (fixed[c0 + 1] << 1) | (fixed[c0 + 2] << 2);
    if (!(fr | fc)) return;
    const int j = q - bp[I];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float *row = vals + (size_t)mesh * nnz + rowptr[3 * I + i] + 3 * j;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (((fr >> i) | (fc >> c)) & 1u) row[c] = c0 + c == 3 * I + i ? 1.0f : 0.0f;
    }
}

// f = K*a in float, ascending column order per row (= the dense row sum of MultiplyMatricesEigen with exact zeros skipped, and the
// oracle's left-to-right sum, bit for bit).  16 lanes per row: they fetch 16 entries of the row side by side -- value, column,
// a[column] -- and lane 0 of the group adds the 16 products IN ORDER, the products handed down the group one lane per step
// (DPP row_shl:1).  The order of the additions is the reference's; only the memory round trips run side by side.  (One thread
// walking its row alone waited for memory at every entry: 34 us for 3,756 rows of ~117 entries, most of an LM trial.)
// Entries past the row's end contribute +0.0f, which leaves a float sum that started at +0.0f unchanged (it can never be -0.0f).
__device__ __forceinline__ float fem_row_sum16(const float *__restrict__ v, const int *__restrict__ lcol, const float *__restrict__ am,
                                               int k0, int k1, int sub)
{
    float s = 0.0f;
    for (int kb = k0; kb < k1; kb += 16) {
        const int k = kb + sub;
        float p = 0.0f;
        if (k < k1) p = v[k] * am[lcol[k]];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            s += p;                                                                                   // lane 0: + product j of the chunk
            p = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, p), 0x101, 0xf, 0xf, true));   // row_shl:1
        }
    }
    return s;   // valid in lane 0 of the 16-lane group
}
__global__ __launch_bounds__(256) void k_fem_matvec(const float *__restrict__ vals, const int *__restrict__ lcol,
                                                    const int *__restrict__ rowptr, size_t nnz, int ndof, const float *__restrict__ a,
                                                    float *__restrict__ f)
{
    const int r = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15, mesh = blockIdx.y;
    if (r >= ndof) return;
    const float s = fem_row_sum16(vals + (size_t)mesh * nnz, lcol, a + (size_t)mesh * ndof, rowptr[r], rowptr[r + 1], sub);
    if (sub == 0) f[(size_t)mesh * ndof + r] = s;
}

// alpha = rz / p.Ap and beta = rz' / rz of a mesh that has nothing left to do (zero load, or a residual that reached exactly 0
// while the other meshes of its batch go on iterating) would be 0 / 0: such a mesh is frozen instead -- the step and the
// new direction are zero, x keeps its value.  Whenever the divisor is positive this is the plain quotient, bit for bit
// (K is positive definite after the Dirichlet elimination, so p.Ap > 0 for p != 0).  The oracle's CG has the same guard.
__device__ __forceinline__ double cg_ratio(double num, double den) { return den > 0.0 ? num / den : 0.0; }

// Sum over the wave, the same value in every lane, in a fixed order: four DPP row_shr steps leave each row of 16 lanes' total in its
// last lane, the four row totals are read into scalars and added row 0 .. 3.  No LDS round trips: the xor butterfly through
// ds_bpermute (12 of them per f64 sum, each step waiting for the last) was 2 us of the 5 the coarse correction added per iteration.
template <int N> __device__ __forceinline__ double dpp_shr_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x110 + N, 0xf, 0xf, true);         // row_shr:N, 0 from beyond the row
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x110 + N, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_shr_f64<1>(v);
    v += dpp_shr_f64<2>(v);
    v += dpp_shr_f64<4>(v);
    v += dpp_shr_f64<8>(v);
    return ((readlane_f64(v, 15) + readlane_f64(v, 31)) + readlane_f64(v, 47)) + readlane_f64(v, 63);
}
// The same sum, bit for bit, of a v that is +0.0 in every lane but 0 and 8 of each row (k_fem_spmv's p.Ap terms: one lane in eight): the
// four steps above reduce to  row total = (v[8] + v[0]) + 0.0  -- every other term of the scan adds a +0.0, which changes nothing but
// the sign of a -0.0, and the one `+ 0.0` left does the same (x + (-x) rounds to +0.0, so no case tells the two apart).
__device__ __forceinline__ double wave_sum_f64_lanes08(double v)
{
    v += dpp_shr_f64<8>(v);
    v += 0.0;
    return ((readlane_f64(v, 8) + readlane_f64(v, 24)) + readlane_f64(v, 40)) + readlane_f64(v, 56);
}
__device__ __forceinline__ double block_sum(double v, double *sh)
{
    v = wave_sum_f64(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    double t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
    return t;
}

// Sum of n per-workgroup partials in a fixed order, the same value in every lane: lane l
// adds part[l], part[l+64], ... and wave_sum_f64 joins the 64 lane sums.  (A serial
// loop over the partials was 5 of the 8 us of k_fem_cg_update on one 6.6k-dof mesh.)
__device__ __forceinline__ double chunk_sum(const double *__restrict__ part, int n)
{
    double v = 0;
    for (int c = threadIdx.x & 63; c < n; c += 64) v += part[c];
    return wave_sum_f64(v);
}

// sE = |a^T f|, nsE = sE / int(Ksize/3): one block per mesh.
__global__ __launch_bounds__(256) void k_fem_energy(const float *__restrict__ a, const float *__restrict__ f, int ndof,
                                                    float *__restrict__ sE, float *__restrict__ nsE, const int4 *__restrict__ minfo)
{
    __shared__ double sh[4];
    const int mesh = blockIdx.x;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const int nrows = minfo ? minfo[mesh].y : ndof;
    double s = 0;
    for (int i = threadIdx.x; i < nrows; i += 256) s += (double)a[row0 + i] * (double)f[row0 + i];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        float e = (float)s;
        if (e < 0.0f) e = -e;
        if (sE) sE[mesh] = e;
        if (nsE) nsE[mesh] = e / (float)(nrows / 3);
    }
}

__global__ void k_fem_displacement(const float *__restrict__ uf, const float *__restrict__ u0, float *__restrict__ a, size_t total)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) a[i] = uf[i] - u0[i];
}
__global__ void k_fem_displacement_dir(float *__restrict__ a, int ndof, const int *__restrict__ ids, int nids, float klarge)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x, mesh = blockIdx.y;
    if (t < nids * 3) a[(size_t)mesh * ndof + 3 * (ids[t / 3] - 1) + t % 3] = 1 / klarge;
}

// LM-hook trial (optimization_algorithm_levenberg.cpp:159-175): GetPointCoordinates
// (double -> float), Set_uf with the recomputed mid-edge / barycentre nodes
// (FEA2.cc:1732-1796), ComputeDisplacement (:1799-1808).  K, u0, the Dirichlet list and
// the derived-node table stay resident; only the optimiser's points come in.
__global__ __launch_bounds__(256) void k_fem_trial_top(const double *__restrict__ points, int npoints,
                                                       const int *__restrict__ derived, int nder, int sequential,
                                                       float *__restrict__ top)
{
    const int mesh = blockIdx.y, nTop = npoints + nder;
    float *t = top + (size_t)mesh * nTop * 3;
    const double *p = points + (size_t)mesh * npoints * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < 3 * npoints; i += gridDim.x * 256) t[i] = (float)p[i];
    if (nder == 0 || blockIdx.x != 0) return;
    __syncthreads(); // block 0 wrote... only its own share: derived nodes need all points -> grid has ONE block when nder > 0
    if (sequential) {
        if (threadIdx.x == 0)
            for (int d = 0; d < nder; ++d) {
                const int *e = derived + 4 * d;
                for (int k = 0; k < 3; ++k)
                    t[3 * (npoints + d) + k] = e[0] == 2 ? (t[3 * e[1] + k] + t[3 * e[2] + k]) / 2
                                                         : (t[3 * e[1] + k] + t[3 * e[2] + k] + t[3 * e[3] + k]) / 3;
            }
    } else {
        for (int i = threadIdx.x; i < 3 * nder; i += 256) {
            const int d = i / 3, k = i - 3 * d;
            const int *e = derived + 4 * d;
            t[3 * (npoints + d) + k] = e[0] == 2 ? (t[3 * e[1] + k] + t[3 * e[2] + k]) / 2
                                                 : (t[3 * e[1] + k] + t[3 * e[2] + k] + t[3 * e[3] + k]) / 3;
        }
    }
}
__global__ void k_fem_trial_a(const float *__restrict__ top, const float *__restrict__ u0, int nTop, float *__restrict__ a)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, mesh = blockIdx.y, ndof = 6 * nTop;
    if (i >= ndof) return;
    const float u = u0[i];
    a[(size_t)mesh * ndof + i] = (i < 3 * nTop ? top[(size_t)mesh * 3 * nTop + i] : u) - u;
}

// The hook's five steps as TWO launches (they were five, 4-5 us each, for 40 trials per frame).
// (1) k_fem_trial_a_fused, ONE workgroup per mesh: the vertex estimates -> float top layer (read straight from the caller's
// pinned block), the derived mid-edge / barycentre nodes (FEA2.cc:1746-1775; in the reference's order when one builds on
// another), a = uf - u0, the Dirichlet entries of ImposeDirichletEncastre_a -- the steps wait for each other at workgroup
// barriers instead of at kernel boundaries.
constexpr int TRIAL_T = 1024;
__global__ __launch_bounds__(TRIAL_T) void k_fem_trial_a_fused(const double *__restrict__ points, int npoints, const int *__restrict__ derived,
                                                               int nder, int sequential, float *__restrict__ top, const float *__restrict__ u0,
                                                               float *__restrict__ a, const int *__restrict__ ids, int nids, float klarge)
{
    const int mesh = blockIdx.x, nTop = npoints + nder, ndof = 6 * nTop, tid = threadIdx.x;
    float *t = top + (size_t)mesh * nTop * 3;
    const double *p = points + (size_t)mesh * npoints * 3;
    for (int i = tid; i < 3 * npoints; i += TRIAL_T) t[i] = (float)p[i];
    __syncthreads();
    if (nder) {
        if (sequential) {
            if (tid == 0)
                for (int d = 0; d < nder; ++d) {
                    const int *e = derived + 4 * d;
                    for (int k = 0; k < 3; ++k)
                        t[3 * (npoints + d) + k] = e[0] == 2 ? (t[3 * e[1] + k] + t[3 * e[2] + k]) / 2
                                                             : (t[3 * e[1] + k] + t[3 * e[2] + k] + t[3 * e[3] + k]) / 3;
                }
        } else {
            for (int i = tid; i < 3 * nder; i += TRIAL_T) {
                const int d = i / 3, k = i - 3 * d;
                const int *e = derived + 4 * d;
                t[3 * (npoints + d) + k] = e[0] == 2 ? (t[3 * e[1] + k] + t[3 * e[2] + k]) / 2
                                                     : (t[3 * e[1] + k] + t[3 * e[2] + k] + t[3 * e[3] + k]) / 3;
            }
        }
        __syncthreads();
    }
    float *am = a + (size_t)mesh * ndof;
    for (int i = tid; i < ndof; i += TRIAL_T) {
        const float u = u0[i];
        am[i] = (i < 3 * nTop ? t[i] : u) - u;
    }
    __syncthreads();
    for (int q = tid; q < nids * 3; q += TRIAL_T) am[3 * (ids[q / 3] - 1) + q % 3] = 1 / klarge;
}
// (2) k_fem_matvec_energy: f = K a as k_fem_matvec (16 lanes per row, the additions in ascending column order), and the workgroup that finishes LAST
// (a counter per mesh, which it resets) runs k_fem_energy's reduction -- the same 256 threads, strides and summation order, so
// sE / nsE have the bits the two separate kernels give -- and writes them where the caller wants them (the pinned block).
__global__ __launch_bounds__(256) void k_fem_matvec_energy(const float *__restrict__ vals, const int *__restrict__ lcol,
                                                           const int *__restrict__ rowptr, size_t nnz, int ndof, const float *__restrict__ a,
                                                           float *__restrict__ f, unsigned *__restrict__ done, float *__restrict__ sE,
                                                           float *__restrict__ nsE)
{
    __shared__ double sh[4];
    __shared__ int s_last;
    const int r = (blockIdx.x * 256 + threadIdx.x) >> 4, sub = threadIdx.x & 15, mesh = blockIdx.y;
    const float *am = a + (size_t)mesh * ndof;
    float *fm = f + (size_t)mesh * ndof;
    if (r < ndof) {
        const float s = fem_row_sum16(vals + (size_t)mesh * nnz, lcol, am, rowptr[r], rowptr[r + 1], sub);
        if (sub == 0) fm[r] = s;
    }
    __threadfence();                      // this workgroup's rows of f are visible device-wide before it is counted
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&done[mesh], 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double s = 0;
    for (int i = threadIdx.x; i < ndof; i += 256) s += (double)am[i] * (double)__builtin_nontemporal_load(fm + i);
    s = block_sum(s, sh);
    if (threadIdx.x == 0) {
        float e = (float)s;
        if (e < 0.0f) e = -e;
        if (sE) sE[mesh] = e;
        if (nsE) nsE[mesh] = e / (float)(ndof / 3);
        done[mesh] = 0;                   // ready for the next trial
    }
}

// ------------------------------------------------------------------------ CG
struct CgScal { double rz[2]; double bb; double rr; };

// Two batch layouts.  Uniform (fem_create): nmesh meshes of one topology, mesh = blockIdx.y, chunk = blockIdx.x.
// Segmented (fem_create_batch): meshes of different sizes and topologies concatenated into ONE block-diagonal
// system (global row / column / non-zero numbering); a chunk never crosses a mesh, chunk -> mesh comes from
// cmesh[], a mesh's row range and chunk range from minfo[mesh] = {row0, nrows, chunk0, nchunks}.  cmesh == nullptr
// selects the uniform layout.
struct Seg {
    int mesh, chunk; // mesh (= index of its CgScal), chunk inside the mesh
    int row0, nrows; // the mesh's rows in the batch vectors
    int tab0;        // index of the mesh's first row in rowptr / diag_idx (0 in the uniform layout: shared tables)
    int part0, nparts; // the mesh's slots in a per-chunk partial array
    size_t voff;     // offset of the mesh's values (0 when segmented: global non-zero numbering)
};
__device__ __forceinline__ Seg seg_of(const int *__restrict__ cmesh, const int4 *__restrict__ minfo, int ndof, int nchunk, size_t nnz)
{
    Seg s;
    if (cmesh) {
        s.mesh = cmesh[blockIdx.x];
        const int4 mi = minfo[s.mesh];
        s.row0 = mi.x; s.nrows = mi.y; s.chunk = (int)blockIdx.x - mi.z; s.part0 = mi.z; s.nparts = mi.w; s.tab0 = mi.x; s.voff = 0;
    } else {
        s.mesh = blockIdx.y; s.chunk = blockIdx.x;
        s.row0 = s.mesh * ndof; s.nrows = ndof; s.tab0 = 0; s.part0 = s.mesh * nchunk; s.nparts = nchunk; s.voff = (size_t)s.mesh * nnz;
    }
    return s;
}

// x = 0, r = b, p = z = r/diag; partial r.z and b.b per chunk.
__global__ __launch_bounds__(CGT) void k_fem_cg_init(const float *__restrict__ vals, const int *__restrict__ diag_idx,
                                                     size_t nnz, int ndof, int nchunk, const double *__restrict__ b,
                                                     double *__restrict__ x, double *__restrict__ r, double *__restrict__ p,
                                                     double *__restrict__ dinv, double *__restrict__ part_a,
                                                     double *__restrict__ part_b, const int *__restrict__ cmesh,
                                                     const int4 *__restrict__ minfo)
{
    __shared__ double sh[CGT / 64];
    const Seg sg = seg_of(cmesh, minfo, ndof, nchunk, nnz);
    double s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < RPB; i += CGT) {
        const int row = sg.chunk * RPB + i;
        if (row < sg.nrows) {
            const size_t g = (size_t)sg.row0 + row;
            const double d = (double)vals[sg.voff + diag_idx[sg.tab0 + row]];
            const double di = d != 0.0 ? 1.0 / d : 0.0, bi = b[g];   // (no diagonal -- a node in no element, K has a zero row and column there: 800 of the reference's 853 meshes have such points --: the dof stays at 0)
            dinv[g] = di; x[g] = 0; r[g] = bi; p[g] = bi * di;
            s1 += bi * (bi * di); s2 += bi * bi;
        }
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { part_a[sg.part0 + sg.chunk] = s1; part_b[sg.part0 + sg.chunk] = s2; }
}
__global__ void k_fem_cg_init2(int nchunk, const double *__restrict__ part_a, const double *__restrict__ part_b, CgScal *__restrict__ sc,
                               const int4 *__restrict__ minfo, const double *__restrict__ wv)
{
    const int mesh = blockIdx.x;
    const int p0 = minfo ? minfo[mesh].z : mesh * nchunk, np = minfo ? minfo[mesh].w : nchunk;
    double a = 0, b = 0;
    for (int c = 0; c < np; ++c) { a += part_a[p0 + c]; b += part_b[p0 + c]; }
    if (wv) a += wv[mesh];   // two-level preconditioner: r.z = r.(r/diag) + (Z^T r).(Ac^-1 Z^T r)
    sc[mesh].rz[0] = a; sc[mesh].rz[1] = a; sc[mesh].bb = b; sc[mesh].rr = b;
}

// Ap = K p, CSR-stream form: a workgroup owns SPB consecutive rows = one contiguous
// run of non-zeros.  Phase 1 streams vals/cols of that run with fully coalesced,
// 3-deep independent 16-byte loads (1 KiB per wave instruction), gathers
// p[col] (L1/L2-resident: 52 KB per mesh) and parks the f64 products in LDS.  Phase 2
// sums each row from LDS with 8 lanes + fixed-order shuffles, writes Ap and the
// partial p.Ap.  HBM sees every matrix byte exactly once.
// SPB = rows per workgroup: 64 for batches (fewest row-pointer reads per byte streamed),
// 32 when the whole launch would otherwise be under ~2 workgroups per CU (one small mesh).
constexpr int SPUB = 2;  // independent 3 x 3 blocks in flight per lane
template <int N> __device__ __forceinline__ double dpp_shl_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x100 + N, 0xf, 0xf, true);         // row_shl:N, 0 beyond the row
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x100 + N, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// The CG works on a block-major copy of the values (fem_cg_setup): the matrix is made of 3 x 3 node blocks
// (build_symbolic: the rows 3I, 3I+1, 3I+2 hold the same columns, in triples 3c, 3c+1, 3c+2), block q of block row I
// (q = bp[I] + j) keeps its nine values together, row-major, at 9 q.  One thread per row.
__global__ __launch_bounds__(256) void k_fem_to_blocks(const float *__restrict__ vals, float *__restrict__ vals_b,
                                                       const int *__restrict__ rowptr, const int *__restrict__ bp,
                                                       const int *__restrict__ blk_row, int nblk, size_t nnz)
{
    // one thread per block: three 12-byte pieces in (consecutive blocks of a block row read consecutive memory in each of its
    // three rows), 36 contiguous bytes out (one thread per ROW copying its ~14 triples at a 160-byte stride in and a 36-byte
    // stride out took 0.29 ms per 256 config-3 meshes; 268 MB in and out)
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nblk) return;
    const int I = blk_row[q], j = q - bp[I];
    float v[9];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float *src = vals + (size_t)blockIdx.y * nnz + rowptr[3 * I + i] + 3 * j;
        v[3 * i] = src[0]; v[3 * i + 1] = src[1]; v[3 * i + 2] = src[2];
    }
    float *dst = vals_b + (size_t)blockIdx.y * nnz + 9 * (size_t)q;
#pragma unroll
    for (int k = 0; k < 9; ++k) dst[k] = v[k];
}

// Ap = K p on the block-major values, workgroup = SPB consecutive rows (a multiple of 3: whole block rows) of one mesh =
// one contiguous run of blocks.  Phase 1, a lane per block: its first column (4 bytes), its nine values (36 contiguous
// bytes, consecutive lanes read consecutive memory) and ONE gather of three consecutive entries of p for all nine
// products; the block's three row sums are parked in LDS.  The kernel is bound by the texture addresser (76 % busy in
// the row-major form, one access per lane and load whatever its width): per non-zero this is 2/9 of a gather, 3/9 of a
// value load and 1/9 of an index load, where the row-major forms needed 1 + 1/4 + 1/4 (quads) or 2/3 + 1/3 + 1/3
// (triples).  No predicate: lanes past the run repeat its last block (same values to the same slots).  Phase 2: 8 lanes
// per row sum its blocks' partials from LDS, DPP `row_shl` reduce in a fixed order, fused p.Ap partial.
// PAP: also emit the workgroup's partial of p.Ap (the chunked vector kernels of a single mesh want it; k_fem_cg_step, which
// reads p and Ap anyway, forms p.Ap itself and spares this kernel a predicated gather of p and a workgroup reduction).
template <int SPB, bool PAP>
__global__ __launch_bounds__(CGT) void k_fem_spmv(const float *__restrict__ vals_b, const int *__restrict__ bcol3,
                                                  const int *__restrict__ bp, size_t nnz, int ndof, int nchunk,
                                                  const double *__restrict__ p, double *__restrict__ Ap,
                                                  double *__restrict__ part_pAp, const int *__restrict__ cmesh,
                                                  const int4 *__restrict__ minfo)
{
    extern __shared__ __align__(16) double part[];   // [blocks of the run][3]
    __shared__ double sh[CGT / 64];
    __shared__ int s_bp[SPB / 3 + 1];                // the run's slice of bp: phase 2 reads it 16 times per row otherwise
    const Seg sg = seg_of(cmesh, minfo, ndof, nchunk, nnz);
    const int tid = threadIdx.x;
    const int r0 = sg.tab0 + sg.chunk * SPB, r1 = min(r0 + SPB, sg.tab0 + sg.nrows); // rows as the tables number them
    const int q0 = bp[r0 / 3], nq = bp[r1 / 3] - q0;
    if (tid <= (r1 - r0) / 3) s_bp[tid] = bp[r0 / 3 + tid] - q0;
    // uniform layout: one index array for all meshes (shared topology): it stays in L2, HBM streams the values only;
    // segmented layout: every mesh has its own (global) column indices, streamed from HBM beside the values
    const float *v = vals_b + sg.voff + 9 * (size_t)q0;
    const int *bc = bcol3 + q0;
    const size_t vbase = (size_t)sg.row0 - sg.tab0;   // batch-vector index of the tables' row 0: mesh * ndof, or 0
    const double *pm = p + vbase;
    for (int q = tid; q < nq; q += SPUB * CGT) {
        float va[SPUB][9]; int ca[SPUB], qq[SPUB];
#pragma unroll
        for (int u = 0; u < SPUB; ++u) { // clamped index: unconditional loads, all in flight together
            qq[u] = min(q + u * CGT, nq - 1);
            __builtin_memcpy(va[u], v + 9 * qq[u], 36);
            ca[u] = bc[qq[u]];
        }
#pragma unroll
        for (int u = 0; u < SPUB; ++u) {
            const double *pp = pm + ca[u];
            const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
            double *dst = part + 3 * qq[u];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                dst[i] = ((double)va[u][3 * i] * p0 + (double)va[u][3 * i + 1] * p1) + (double)va[u][3 * i + 2] * p2;
        }
    }
    __syncthreads();
    const int sub = tid / LPR, sl = tid % LPR;
    double acc = 0;
#pragma unroll
    for (int pass = 0; pass < (SPB + CGT / LPR - 1) / (CGT / LPR); ++pass) {
        const int row = r0 + pass * (CGT / LPR) + sub;
        double s = 0;
        if (row < r1) {
            const int I = (row - r0) / 3, i = row - r0 - 3 * I, b0 = s_bp[I], nb = s_bp[I + 1] - b0;
            for (int j = sl; j < nb; j += LPR) s += part[3 * (b0 + j) + i];
        }
        // sum of the 8 lanes of a row into its lane 0: DPP row_shl (lane i reads lane i+n of its 16-lane row), fixed order
        s += dpp_shl_f64<4>(s);
        s += dpp_shl_f64<2>(s);
        s += dpp_shl_f64<1>(s);
        if (row < r1 && sl == 0) {
            const size_t g = vbase + row;
            Ap[g] = s;
            if (PAP) acc += p[g] * s;
        }
    }
    if (PAP) {
        acc = block_sum(acc, sh);
        if (tid == 0) part_pAp[sg.part0 + sg.chunk] = acc;
    }
}

// alpha = rz/pAp; x += alpha p; r -= alpha Ap; partial r.(r/diag) and r.r.
__global__ __launch_bounds__(CGT) void k_fem_cg_update(int ndof, int nchunk, int nchunk_s, int cur, const CgScal *__restrict__ sc,
                                                       const double *__restrict__ part_pAp, const double *__restrict__ p,
                                                       const double *__restrict__ Ap, const double *__restrict__ dinv,
                                                       double *__restrict__ x, double *__restrict__ r,
                                                       double *__restrict__ part_rz, double *__restrict__ part_rr,
                                                       const int *__restrict__ cmesh, const int4 *__restrict__ minfo,
                                                       const int4 *__restrict__ minfo_s)
{
    __shared__ double sh[CGT / 64];
    const Seg sg = seg_of(cmesh, minfo, ndof, nchunk, 0);
    const int sp0 = minfo_s ? minfo_s[sg.mesh].z : sg.mesh * nchunk_s, snp = minfo_s ? minfo_s[sg.mesh].w : nchunk_s;
    const double pAp = chunk_sum(part_pAp + sp0, snp);
    const double alpha = cg_ratio(sc[sg.mesh].rz[cur], pAp);
    double s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < RPB; i += CGT) {
        const int row = sg.chunk * RPB + i;
        if (row < sg.nrows) {
            const size_t g = (size_t)sg.row0 + row;
            x[g] += alpha * p[g];
            const double ri = r[g] - alpha * Ap[g];
            r[g] = ri;
            s1 += ri * (ri * dinv[g]);
            s2 += ri * ri;
        }
    }
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    if (threadIdx.x == 0) { part_rz[sg.part0 + sg.chunk] = s1; part_rr[sg.part0 + sg.chunk] = s2; }
}

// beta = rz_new/rz; p = r/diag + beta p; chunk 0 publishes rz_new for the next iteration.
__global__ __launch_bounds__(CGT) void k_fem_cg_dir(int ndof, int nchunk, int cur, CgScal *__restrict__ sc,
                                                    const double *__restrict__ part_rz, const double *__restrict__ part_rr,
                                                    const double *__restrict__ r, const double *__restrict__ dinv,
                                                    double *__restrict__ p, const int *__restrict__ cmesh,
                                                    const int4 *__restrict__ minfo, const double *__restrict__ cz,
                                                    const double *__restrict__ wv)
{
    const Seg sg = seg_of(cmesh, minfo, ndof, nchunk, 0);
    double rz2 = chunk_sum(part_rz + sg.part0, sg.nparts);
    const double rr = chunk_sum(part_rr + sg.part0, sg.nparts);
    if (cz) rz2 += wv[sg.mesh];   // two-level preconditioner: z = r/diag + cz, r.z = r.(r/diag) + w.v (k_fem_cz_solve)
    const double beta = cg_ratio(rz2, sc[sg.mesh].rz[cur]);
    for (int i = threadIdx.x; i < RPB; i += CGT) {
        const int row = sg.chunk * RPB + i;
        if (row < sg.nrows) {
            const size_t g = (size_t)sg.row0 + row;
            p[g] = cz ? (r[g] * dinv[g] + cz[g]) + beta * p[g] : r[g] * dinv[g] + beta * p[g];
        }
    }
    if (sg.chunk == 0 && threadIdx.x == 0) { sc[sg.mesh].rz[cur ^ 1] = rz2; sc[sg.mesh].rr = rr; }
}

// ---- Two-level preconditioner (fem_cg_preconditioner(FEM_PRECOND_TWO_LEVEL)): z = r/diag + Z Ac^-1 Z^T r with Z = the six rigid-body
// modes (three translations, three rotations about the centroid) of 2 x 2 x 2 geometric aggregates of a mesh's nodes -- 48 coarse
// dofs -- and Ac = Z^T K Z.  Point-Jacobi leaves the smooth, near-rigid error of a near-incompressible solid to thousands of
// iterations; the coarse term removes it (config 3: 1,274 -> 470 iterations to 1e-8).  r.z = r.(r/diag) + w.v with w = Z^T r and v =
// Ac^-1 w, so the vector kernels keep their sums and the coarse part adds a 48-term dot product.
// Data, per mesh: cz[] = one float4 per node in AGGREGATE order {q = node - centroid of its aggregate, bits: node | constrained dofs
// << 28} (a constrained dof has a zero row in Z), czptr[9] = where each aggregate's nodes start, aci[48][48] = the symmetric inverse.
constexpr int CZ_NA = 8, CZ_NC = 6 * CZ_NA, CZ_T = 256, CZR_U = 5;   // CZR_U x 64: the largest aggregate k_fem_cg_resident takes
struct CzNode { double q0, q1, q2; int node; bool m0, m1, m2; };
__device__ __forceinline__ CzNode cz_node(const float4 e)
{
    const unsigned id = __float_as_uint(e.w);
    return CzNode{(double)e.x, (double)e.y, (double)e.z, (int)(id & 0x0fffffffu), (id >> 28 & 1) != 0, (id >> 29 & 1) != 0, (id >> 30 & 1) != 0};
}
// six sums of an aggregate's node: the three components and q x r
#define CZ_RESTRICT_ADD(s, n, r0, r1, r2)                                                                                         \
    do {                                                                                                                          \
        s[0] += r0; s[1] += r1; s[2] += r2;                                                                                       \
        s[3] += n.q1 * r2 - n.q2 * r1; s[4] += n.q2 * r0 - n.q0 * r2; s[5] += n.q0 * r1 - n.q1 * r0;                              \
    } while (0)
// The three steps of the coarse correction as launches of their own, for meshes of more than 65,536 nodes (smaller ones: k_fem_cz_apply).
// w[mesh][6 a + m] = (Z^T src)[6 a + m]: a workgroup per (aggregate, mesh), thread-strided partials and the block sum in its fixed order
__global__ __launch_bounds__(CZ_T) void k_fem_cz_restrict(const float4 *__restrict__ cz, const int *__restrict__ czptr,
                                                          const double *__restrict__ src, double *__restrict__ out, int ndof,
                                                          const int4 *__restrict__ minfo)
{
    __shared__ double sh[CZ_T / 64];
    const int a = blockIdx.x, mesh = blockIdx.y;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const float4 *lz = cz + row0 / 3;
    const double *r = src + row0;
    const int p0 = czptr[9 * mesh + a], p1 = czptr[9 * mesh + a + 1];
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int p = p0 + (int)threadIdx.x; p < p1; p += CZ_T) {
        const CzNode n = cz_node(lz[p]);
        const double r0 = n.m0 ? 0.0 : r[3 * n.node], r1 = n.m1 ? 0.0 : r[3 * n.node + 1], r2 = n.m2 ? 0.0 : r[3 * n.node + 2];
        CZ_RESTRICT_ADD(s, n, r0, r1, r2);
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) s[m] = block_sum(s[m], sh);
    if (threadIdx.x == 0)
#pragma unroll
        for (int m = 0; m < 6; ++m) out[(size_t)mesh * CZ_NC + 6 * a + m] = s[m];
}
// v = Ac^-1 w and w.v, one wave per mesh (the inverse is symmetric: lane k reads column k, consecutive lanes consecutive memory)
__global__ __launch_bounds__(64) void k_fem_cz_solve(const double *__restrict__ aci, const double *__restrict__ w, double *__restrict__ v,
                                                     double *__restrict__ wv)
{
    const int mesh = blockIdx.x, k = min((int)threadIdx.x, CZ_NC - 1);
    const double *A = aci + (size_t)mesh * (CZ_NC * CZ_NC), *wm = w + (size_t)mesh * CZ_NC;
    double s = 0;
    for (int j = 0; j < CZ_NC; ++j) s += A[j * CZ_NC + k] * wm[j];
    double t = (int)threadIdx.x < CZ_NC ? wm[k] * s : 0.0;
    t = wave_sum_f64(t);
    if ((int)threadIdx.x < CZ_NC) v[(size_t)mesh * CZ_NC + k] = s;
    if (threadIdx.x == 0) wv[mesh] = t;
}
// out (=, or += when `accumulate`) Z v: v + omega x q per node, 0 at constrained dofs.  Every node is in exactly one aggregate, so
// `=` writes the whole vector.
__global__ __launch_bounds__(CZ_T) void k_fem_cz_prolong(const float4 *__restrict__ cz, const int *__restrict__ czptr,
                                                         const double *__restrict__ v, double *__restrict__ out,
                                                         int accumulate, int ndof, const int4 *__restrict__ minfo)
{
    const int a = blockIdx.x, mesh = blockIdx.y;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const float4 *lz = cz + row0 / 3;
    double *o = out + row0;
    const int p0 = czptr[9 * mesh + a], p1 = czptr[9 * mesh + a + 1];
    double va[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) va[m] = v[(size_t)mesh * CZ_NC + 6 * a + m];
    for (int p = p0 + (int)threadIdx.x; p < p1; p += CZ_T) {
        const CzNode n = cz_node(lz[p]);
        const double c0 = n.m0 ? 0.0 : va[0] + (va[4] * n.q2 - va[5] * n.q1);
        const double c1 = n.m1 ? 0.0 : va[1] + (va[5] * n.q0 - va[3] * n.q2);
        const double c2 = n.m2 ? 0.0 : va[2] + (va[3] * n.q1 - va[4] * n.q0);
        double *d = o + 3 * n.node;
        if (accumulate) { d[0] += c0; d[1] += c1; d[2] += c2; }
        else { d[0] = c0; d[1] = c1; d[2] = c2; }
    }
}

// The coarse space of one mesh per workgroup (fem_cg_setup, once per set of constrained dofs): bounding box (block min / max),
// aggregate = 4 bx + 2 by + bz with bit = coordinate > midpoint (in double), the nodes of each aggregate in node order (counting
// pass, then placement by wave ballots -- stable), centroids as double sums over those lists IN LIST ORDER (one thread per aggregate
// and axis: the oracle's sums, bit for bit), q = (float)(node - centroid).  cmask: constrained dofs (numbered like the rows: per
// mesh when the layout is uniform, globally when segmented), or nullptr.
// ... and such dofs count as constrained in the coarse space (zero rows in Z): the mask the caller's constraints gave, OR "no diagonal"
__global__ void k_fem_mask_no_diagonal(const float *__restrict__ vals, const int *__restrict__ diag_idx, int ndof, uint8_t *__restrict__ cmask)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < ndof && vals[diag_idx[row]] == 0.0f) cmask[row] = 1;
}
__global__ __launch_bounds__(CZ_T) void k_fem_cz_build(const float *__restrict__ nodes, const uint8_t *__restrict__ cmask, int mask_global,
                                                       float4 *__restrict__ cz, float4 *__restrict__ cznode,
                                                       int *__restrict__ czptr, int *__restrict__ maxagg, int ndof, const int4 *__restrict__ minfo)
{
    __shared__ float s_lo[3][CZ_T / 64], s_hi[3][CZ_T / 64];
    __shared__ int s_cnt[CZ_T / 64][CZ_NA], s_base[CZ_NA], s_ptr[CZ_NA + 1];
    __shared__ double s_cen[CZ_NA][3];
    const int mesh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const int cnt = (minfo ? minfo[mesh].y : ndof) / 3;
    const float *P = nodes + row0;                       // 3 floats per node, as the rows
    float4 *lz = cz + row0 / 3;
    const uint8_t *mk = cmask ? cmask + (mask_global ? row0 : 0) : nullptr;
    float lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = P[k]; hi[k] = P[k]; }
    for (int i = tid; i < cnt; i += CZ_T)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float v = P[3 * i + k]; lo[k] = fminf(lo[k], v); hi[k] = fmaxf(hi[k], v); }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off)); }
        if (lane == 0) { s_lo[k][wave] = lo[k]; s_hi[k][wave] = hi[k]; }
    }
    __syncthreads();
    double mid[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float l = s_lo[k][0], h = s_hi[k][0];
        for (int w = 1; w < CZ_T / 64; ++w) { l = fminf(l, s_lo[k][w]); h = fmaxf(h, s_hi[k][w]); }
        mid[k] = 0.5 * ((double)l + (double)h);
    }
    auto agg_of = [&](int i) { return 4 * ((double)P[3 * i] > mid[0]) + 2 * ((double)P[3 * i + 1] > mid[1]) + ((double)P[3 * i + 2] > mid[2]); };
    // counting pass
    int c8[CZ_NA] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < cnt; i += CZ_T) {
        const int a = agg_of(i);
#pragma unroll
        for (int b = 0; b < CZ_NA; ++b) c8[b] += a == b;
    }
#pragma unroll
    for (int b = 0; b < CZ_NA; ++b) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) c8[b] += __shfl_xor(c8[b], off);
        if (lane == 0) s_cnt[wave][b] = c8[b];
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0, mx = 0;
        for (int b = 0; b < CZ_NA; ++b) {
            int t = 0;
            for (int w = 0; w < CZ_T / 64; ++w) t += s_cnt[w][b];
            s_ptr[b] = run; s_base[b] = run; run += t; mx = max(mx, t);
            czptr[9 * mesh + b] = s_ptr[b];
        }
        s_ptr[CZ_NA] = run; czptr[9 * mesh + CZ_NA] = run;
        atomicMax(maxagg, mx);
    }
    __syncthreads();
    // placement, a tile of CZ_T consecutive nodes at a time: rank inside the wave by ballot, waves in order, tiles in order
    for (int base = 0; base < cnt; base += CZ_T) {
        const int i = base + tid, a = i < cnt ? agg_of(i) : -1;
        int rank = 0;
#pragma unroll
        for (int b = 0; b < CZ_NA; ++b) {
            const unsigned long long bal = __ballot(a == b);
            if (a == b) rank = __popcll(bal & ((1ull << lane) - 1));
            if (lane == 0) s_cnt[wave][b] = __popcll(bal);
        }
        __syncthreads();
        if (a >= 0) {
            int pos = s_base[a] + rank;
            for (int w = 0; w < wave; ++w) pos += s_cnt[w][a];
            unsigned bits = (unsigned)i;
            if (mk) bits |= (unsigned)(mk[3 * i] != 0) << 28 | (unsigned)(mk[3 * i + 1] != 0) << 29 | (unsigned)(mk[3 * i + 2] != 0) << 30;
            lz[pos].w = __uint_as_float(bits);
        }
        __syncthreads();
        if (tid < CZ_NA) { int t = 0; for (int w = 0; w < CZ_T / 64; ++w) t += s_cnt[w][tid]; s_base[tid] += t; }
        __syncthreads();
    }
    __threadfence_block();
    // centroids: thread (a, k) adds its aggregate's coordinate k in list order
    if (tid < 3 * CZ_NA) {
        const int a = tid / 3, k = tid - 3 * a, p0 = s_ptr[a], p1 = s_ptr[a + 1];
        double sum = 0;
        for (int p = p0; p < p1; ++p) sum += (double)P[3 * (int)(__float_as_uint(lz[p].w) & 0x0fffffffu) + k];
        s_cen[a][k] = p1 > p0 ? sum / (double)(p1 - p0) : 0.0;
    }
    __syncthreads();
    for (int p = tid; p < cnt; p += CZ_T) {
        int a = 0;
#pragma unroll
        for (int b = 1; b < CZ_NA; ++b) a += p >= s_ptr[b];
        const int i = (int)(__float_as_uint(lz[p].w) & 0x0fffffffu);
        const float q0 = (float)((double)P[3 * i] - s_cen[a][0]), q1 = (float)((double)P[3 * i + 1] - s_cen[a][1]), q2 = (float)((double)P[3 * i + 2] - s_cen[a][2]);
        lz[p].x = q0; lz[p].y = q1; lz[p].z = q2;
        // the same by node: {q, aggregate | constrained dofs << 28} (k_fem_cz_kz looks a block's column node up)
        cznode[row0 / 3 + i] = float4{q0, q1, q2, __uint_as_float((unsigned)a | (__float_as_uint(lz[p].w) & 0x70000000u))};
    }
}

// Z^T K Z six columns at a time.  Pass b (one launch each of the two kernels below per aggregate b) takes the six modes of aggregate b:
// k_fem_cz_kz streams K once (a workgroup = the blocks of KZ_SPB consecutive rows, a lane a block, as the product kernel), forms
// K_IJ G_J (3 x 6; G_J = [I | -[q_J]x], rows of constrained dofs zero) for the blocks whose column node J is in aggregate b -- G_J
// from the node table, no vectors --, sums them per row in block order and writes Y = K Z_b (6 doubles per row);
// k_fem_cz_restrict6 forms Z_a^T Y for every aggregate a: the 48 x 6 columns 6b .. 6b+5 of Ac.  Eight streams of K instead of the
// 48 products of single columns (prolong, product, restrict: 4.3 -> 1.7 ms of fem_cg_setup on 256 config-3 meshes).
constexpr int KZ_SPB = 24;   // rows per workgroup: 8 block rows, ~110-220 blocks, 16-32 KB of LDS (18 sums per block) -- the product
                             // kernel's 96 rows would be 70 KB: two workgroups per compute unit, 158 us per pass instead of 98
__global__ __launch_bounds__(CGT) void k_fem_cz_kz(const float *__restrict__ vals_b, const int *__restrict__ bcol3, const int *__restrict__ bp,
                                                   size_t nnzs, int ndof, const float4 *__restrict__ cznode, int b, double *__restrict__ Y)
{
    // uniform layout: mesh = blockIdx.y, tables numbered from 0; segmented: launched as ONE system in global numbering (block rows
    // are independent of each other: a workgroup's rows may straddle two meshes)
    extern __shared__ __align__(16) double part[];   // [blocks of the run][18]: entry 6 t + m = row t of the block, mode m
    __shared__ int s_bp[KZ_SPB / 3 + 1];
    const int tid = threadIdx.x, mesh = blockIdx.y;
    const int r0 = blockIdx.x * KZ_SPB, r1 = min(r0 + KZ_SPB, ndof);
    const int q0 = bp[r0 / 3], nq = bp[r1 / 3] - q0;
    if (tid <= (r1 - r0) / 3) s_bp[tid] = bp[r0 / 3 + tid] - q0;
    const float *v = vals_b + (size_t)mesh * nnzs + 9 * (size_t)q0;
    const int *bc = bcol3 + q0;
    const size_t vbase = (size_t)mesh * ndof;
    const float4 *nqm = cznode + vbase / 3;
    for (int q = tid; q < nq; q += CGT) {
        const float4 e = nqm[bc[q] / 3];
        const unsigned bits = __float_as_uint(e.w);
        double *dst = part + 18 * q;
        if ((int)(bits & 0xfu) == b) {       // one block in eight: the other seven do not even read their values
            float va[9];
            __builtin_memcpy(va, v + 9 * q, 36);
            const bool m0 = (bits >> 28 & 1) != 0, m1 = (bits >> 29 & 1) != 0, m2 = (bits >> 30 & 1) != 0;
            const double x = e.x, y = e.y, z = e.z;
            // G_J, row by dof of J: translations e_m; rotations about e_0, e_1, e_2: (0, -z, y), (z, 0, -x), (-y, x, 0) in the columns
            const double g0[6] = {m0 ? 0.0 : 1.0, 0, 0, 0, m0 ? 0.0 : z, m0 ? 0.0 : -y};
            const double g1[6] = {0, m1 ? 0.0 : 1.0, 0, m1 ? 0.0 : -z, 0, m1 ? 0.0 : x};
            const double g2[6] = {0, 0, m2 ? 0.0 : 1.0, m2 ? 0.0 : y, m2 ? 0.0 : -x, 0};
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 6; ++m)
                    dst[6 * t + m] = ((double)va[3 * t] * g0[m] + (double)va[3 * t + 1] * g1[m]) + (double)va[3 * t + 2] * g2[m];
        } else {
#pragma unroll
            for (int k = 0; k < 18; ++k) dst[k] = 0.0;
        }
    }
    __syncthreads();
    const int nbr = (r1 - r0) / 3;
    for (int idx = tid; idx < 18 * nbr; idx += CGT) {
        const int I = idx / 18, tm = idx - 18 * I, b0 = s_bp[I], nb = s_bp[I + 1] - b0;
        double sum = 0;
        for (int j = 0; j < nb; ++j) sum += part[18 * (b0 + j) + tm];
        Y[(vbase + r0 + 3 * I) * 6 + tm] = sum;      // rows 3I .. 3I+2 of the run, six modes each: entry 6 t + m again
    }
}
// ac[mesh][(6 a + m') * 48 + 6 b + m] = (Z_a^T Y)[m'][m]: a workgroup per (aggregate a, mesh), thread-strided partials, DPP wave sums,
// the four waves added in order
__global__ __launch_bounds__(CZ_T) void k_fem_cz_restrict6(const float4 *__restrict__ cz, const int *__restrict__ czptr,
                                                           const double *__restrict__ Y, int b, double *__restrict__ ac, int ndof,
                                                           const int4 *__restrict__ minfo)
{
    __shared__ double sh[CZ_T / 64][36];
    const int a = blockIdx.x, mesh = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const float4 *lz = cz + row0 / 3;
    const double *Ym = Y + row0 * 6;
    const int p0 = czptr[9 * mesh + a], p1 = czptr[9 * mesh + a + 1];
    double W[6][6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int m = 0; m < 6; ++m) W[i][m] = 0;
    for (int p = p0 + (int)threadIdx.x; p < p1; p += CZ_T) {
        const CzNode n = cz_node(lz[p]);
        const double *yn = Ym + 18 * (size_t)n.node;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const double r0 = n.m0 ? 0.0 : yn[m], r1 = n.m1 ? 0.0 : yn[6 + m], r2 = n.m2 ? 0.0 : yn[12 + m];
            W[0][m] += r0; W[1][m] += r1; W[2][m] += r2;
            W[3][m] += n.q1 * r2 - n.q2 * r1; W[4][m] += n.q2 * r0 - n.q0 * r2; W[5][m] += n.q0 * r1 - n.q1 * r0;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            const double t = wave_sum_f64(W[i][m]);
            if (lane == 0) sh[wave][6 * i + m] = t;
        }
    __syncthreads();
    if (threadIdx.x < 36) {
        double t = 0;
        for (int w = 0; w < CZ_T / 64; ++w) t += sh[w][threadIdx.x];
        const int i = threadIdx.x / 6, m = threadIdx.x - 6 * i;
        ac[(size_t)mesh * (CZ_NC * CZ_NC) + (size_t)(6 * a + i) * CZ_NC + 6 * b + m] = t;
    }
}

// aci = inverse of the symmetrised coarse matrix ac (48 x 48, one wave per mesh, everything in LDS): Cholesky column by column
// (lane i owns row i), then lane c solves L y = e_c and L^T x = y for column c, and the result is symmetrised again.  A coarse dof
// whose diagonal is <= 1e-12 of the largest (an aggregate without a free dof) or whose pivot is <= CZ_PIVOT_MIN = 1e-4 of its diagonal
// (a mode that the ones before it span, or all but span: the rotations of an aggregate whose free nodes lie on or near a line) is
// dropped: zero row and column in the inverse.  A pivot test against 0 would keep or drop such a mode by the last bit of Ac, and a
// kept one makes the preconditioner all but singular: until round 5 the limit was 1e-8, and a warped 54-node mesh whose fifth
// aggregate kept a rotation at 6e-7 (largest entry of the inverse 1.7e10) made the iteration a function of the last bits --
// relres after 45 iterations 0.15 here, 0.29 in the oracle, 0.39 .. 5.5 with the inverse perturbed by 1e-15 (profiles/r05_notes.md).
// The operation order of the oracle's oracle_fem_coarse_inverse.
constexpr double CZ_PIVOT_MIN = 1e-4;
__global__ __launch_bounds__(64) void k_fem_cz_invert(const double *__restrict__ ac, double *__restrict__ aci)
{
    constexpr int N = CZ_NC;
    __shared__ double A[N * N], L[N * N], X[N * N];   // A doubles as y of the triangular solves
    __shared__ int keep[N];
    const int mesh = blockIdx.x, l = threadIdx.x, i = min(l, N - 1);
    const double *G = ac + (size_t)mesh * N * N;
    for (int j = 0; j < N; ++j) { A[i * N + j] = 0.5 * (G[i * N + j] + G[j * N + i]); L[i * N + j] = 0; X[i * N + j] = 0; }
    double dmax = l < N ? G[i * N + i] : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dmax = fmax(dmax, __shfl_xor(dmax, off));
    if (l < N) keep[i] = G[i * N + i] > 1e-12 * dmax;
    __syncthreads();
    for (int j = 0; j < N; ++j) {
        if (keep[j]) {                                   // uniform
            double d = A[j * N + j];
            for (int k = 0; k < j; ++k) d -= L[j * N + k] * L[j * N + k];
            if (!(d > CZ_PIVOT_MIN * A[j * N + j])) {
                __syncthreads();
                if (l < j) L[j * N + l] = 0;
                if (l == j) keep[j] = 0;
            } else {
                const double ljj = sqrt(d);
                if (l < N && l > j && keep[l]) {
                    double t = A[l * N + j];
                    for (int k = 0; k < j; ++k) t -= L[l * N + k] * L[j * N + k];
                    L[l * N + j] = t / ljj;
                }
                if (l == j) L[j * N + j] = ljj;
            }
        }
        __syncthreads();
    }
    if (l < N && keep[l]) {
        const int c = l;
        for (int r = 0; r < N; ++r) {
            if (!keep[r]) { A[r * N + c] = 0; continue; }
            double t = r == c ? 1.0 : 0.0;
            for (int j = 0; j < r; ++j) t -= L[r * N + j] * A[j * N + c];
            A[r * N + c] = t / L[r * N + r];
        }
        for (int r = N - 1; r >= 0; --r) {
            if (!keep[r]) continue;
            double t = A[r * N + c];
            for (int j = r + 1; j < N; ++j) t -= L[j * N + r] * X[j * N + c];
            X[r * N + c] = t / L[r * N + r];
        }
    }
    __syncthreads();
    if (l < N)
        for (int j = 0; j < N; ++j) aci[(size_t)mesh * N * N + j * N + l] = 0.5 * (X[j * N + l] + X[l * N + j]);
}

// Batches: the vector half of an iteration in one launch, one 1024-thread workgroup per mesh.  alpha and beta are per
// mesh, so a mesh's workgroup needs nobody else: pAp = p.Ap; alpha = rz / pAp; x += alpha p; r -= alpha Ap; rz' = r.(r/diag) and
// rr = r.r summed by the workgroup (thread-strided partials, wave butterflies, the waves in order: fixed order, no
// atomics); beta = rz' / rz; p = r/diag + beta p.  Rows go through in blocks of CGS_U x 1024 with all loads of a block in
// flight together (one memory round trip per block and phase).  A single mesh (the reference's own use, one mesh per
// PoseOptimizationNR call) keeps the two launches above: there one workgroup is one CU's bandwidth, ~100 are the chip's.
constexpr int CGS_T = 1024, CGS_U = 10, CGS_MIN_MESHES = 16;
// The coarse correction of one mesh by one 1024-thread workgroup (all threads call it; rm = the mesh's slice of the vector to
// restrict, complete before the call or at its first barrier; cm = where Z v goes): waves a and a + 8 take alternate groups of 64
// nodes of aggregate a's list, partials in wave order; wave 0 forms v = Ac^-1 w and w.v; the waves write (or add) Z v for their
// nodes.  Returns w.v.  Ends with a barrier: Z v is complete.
__device__ __forceinline__ double cz_apply_block(const float4 *__restrict__ lz, const int *__restrict__ ptr9, const double *__restrict__ A,
                                                 const double *rm, double *cm, bool accumulate, double (*s_w)[CZ_NC], double *s_v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, a = wave & 7, half = wave >> 3;
    const int zp0 = ptr9[a], zp1 = ptr9[a + 1];
    __syncthreads();                                   // the vector is complete
    double w6[6] = {0, 0, 0, 0, 0, 0};
    // four groups of 64 nodes at a time: their entries requested together, then their r -- two round trips per four groups (config
    // 3's 275 nodes per aggregate: two in all) where a plain loop makes two per group
    constexpr int U = 4;
    for (int q0 = zp0 + 64 * half + lane; q0 - lane < zp1; q0 += 128 * U) {
        float4 e[U]; double g[U][3];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = lz[min(q0 + 128 * u, zp1 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *rs = rm + 3 * (int)(__float_as_uint(e[u].w) & 0x0fffffffu);
            g[u][0] = rs[0]; g[u][1] = rs[1]; g[u][2] = rs[2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const CzNode n = cz_node(e[u]);
            const bool in = q0 + 128 * u < zp1;
            const double r0 = n.m0 || !in ? 0.0 : g[u][0], r1 = n.m1 || !in ? 0.0 : g[u][1], r2 = n.m2 || !in ? 0.0 : g[u][2];
            CZ_RESTRICT_ADD(w6, n, r0, r1, r2);
        }
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) { w6[m] = wave_sum_f64(w6[m]); if (lane == m) s_w[half][6 * a + m] = w6[m]; }
    __syncthreads();
    if (wave == 0) {
        const int k = min(lane, CZ_NC - 1);
        double v = 0;
        for (int j = 0; j < CZ_NC; ++j) v += A[j * CZ_NC + k] * (s_w[0][j] + s_w[1][j]);
        const double t = wave_sum_f64(lane < CZ_NC ? (s_w[0][k] + s_w[1][k]) * v : 0.0);
        if (lane < CZ_NC) s_v[k] = v;
        if (lane == 0) s_v[CZ_NC] = t;
    }
    __syncthreads();
    double va[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) va[m] = s_v[6 * a + m];
    for (int q0 = zp0 + 64 * half + lane; q0 - lane < zp1; q0 += 128 * U) {
        float4 e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) e[u] = lz[min(q0 + 128 * u, zp1 - 1)];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (q0 + 128 * u >= zp1) continue;
            const CzNode n = cz_node(e[u]);
            double *cs = cm + 3 * n.node;
            const double c0 = n.m0 ? 0.0 : va[0] + (va[4] * n.q2 - va[5] * n.q1);
            const double c1 = n.m1 ? 0.0 : va[1] + (va[5] * n.q0 - va[3] * n.q2);
            const double c2 = n.m2 ? 0.0 : va[2] + (va[3] * n.q1 - va[4] * n.q0);
            if (accumulate) { cs[0] += c0; cs[1] += c1; cs[2] += c2; }
            else { cs[0] = c0; cs[1] = c1; cs[2] = c2; }
        }
    }
    const double wv = s_v[CZ_NC];
    __syncthreads();                                   // Z v complete
    return wv;
}
// The same as a launch of its own: restriction, coarse solve and prolongation of few meshes in ONE launch instead of three (a single
// mesh is launch-bound: six launches per iteration became four, 12.8 -> 10.4 ms to 1e-8 on config 3's mesh)
__global__ __launch_bounds__(1024) void k_fem_cz_apply(const float4 *__restrict__ cz, const int *__restrict__ czptr, const double *__restrict__ aci,
                                                       const double *src, double *out, int accumulate, double *__restrict__ wv, int ndof,
                                                       const int4 *__restrict__ minfo)
{
    __shared__ double s_w[2][CZ_NC], s_v[CZ_NC + 1];
    const int mesh = blockIdx.x;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const double t = cz_apply_block(cz + row0 / 3, czptr + 9 * mesh, aci + (size_t)mesh * (CZ_NC * CZ_NC), src + row0, out + row0, accumulate != 0, s_w, s_v);
    if (threadIdx.x == 0) wv[mesh] = t;
}

// ---- ONE mesh, all iterations of a call in ONE launch, spread over the compute units of one XCD (k_fem_cg_xcd).
// A single mesh cannot use k_fem_cg_resident's trick at speed (one compute unit's bandwidth: measured slower) and pays 3 launches of
// ~3 us each per iteration on the launch-per-phase path.  Here P <= 32 workgroups of a 256-workgroup launch stay resident for the
// whole call -- blockIdx % 8 == 0: workgroups are dealt round-robin over the XCDs, so these share one (for speed only: nothing below
// relies on it).  The arithmetic is the launch-per-phase path's, chunk for chunk: workgroup `rank` owns the SpMV chunks [c0, c1)
// (SPB rows each: same lanes, same LDS partials, same DPP sums as k_fem_spmv, same p.Ap partial per chunk) and the vector chunk
// `rank` (RPB rows: k_fem_cg_update's and k_fem_cg_dir's formulas, one row per thread, same block sums); chunk partials are joined
// in chunk_sum's order -- x, r, p and the scalars come out bit for bit.
//   * the matrix never moves: a thread keeps the nine values and the column of its blocks (<= XG_MAXCH x XG_MAXQ) in registers
//   * p lives in LDS, replicated: every workgroup keeps p (and 1/diag) over the column range [lo, hi) its blocks and rows touch and
//     forms p = r/diag + beta p there itself, from the r its peers published -- p is never handed over
//   * handed over per iteration: K p rows, r rows and three arrays of chunk partials, every value as a TAGGED 16-byte granule
//     {value, tag, check}: one sc1 store by the producer, sc1 loads polled by the consumer until tag (= launch base + 2 it + phase) and
//     check word match -- no counter, no store drain, no separate barrier: an iteration is TWO store -> load hops across the fabric.
//     (A first version with two counter barriers per iteration -- stores, s_waitcnt, agent-scope add, poll, loads -- made ~8
//     dependent fabric round trips of an iteration: 10.9 us on the 6,591-dof mesh, 7.1 on a 648-dof one; tools/ubench/xcd_barrier.hip
//     prices the pieces: 1.2 us per barrier at P = 32, 2.3 us to stage 53 KB.)  MI355X_MICROARCH.md observes 16-byte sc1 granules
//     untorn on gfx950 without promising it: the check word (value ^ tag ^ constant) turns a torn granule into "not there yet".
//   * no write-after-read hazard without barriers: a slot is rewritten only by a workgroup that has since consumed data whose
//     existence implies the slot's readers are done (K p rows and r rows: their consumer produced what the writer waited for; the
//     chunk partials are double-buffered by iteration parity, and a workgroup two iterations ahead is impossible -- each phase needs
//     every workgroup's partial of the phase before).
// Every spin is bounded: a timeout raises the abort word, every workgroup leaves, and fem_cg_result / fem_cg report the failure.
constexpr int XG_MAXCH = 6, XG_MAXQ = 2, XG_SU = 8, XG_STRIDE = 8, XG_MAXP = 64;   // XG_MAXP: up to two workgroups per compute unit of the XCD (variants MC <= 3)
constexpr unsigned XG_KEY = 0x5bd1e995u;
constexpr unsigned long long XG_TIMEOUT = 200000ull;      // 2 ms of the 100-MHz wall clock (s_memrealtime): see xg_get
#ifdef XG_TIMING
struct XgCtl { unsigned abort_flag, pad[31 + 8 * 64]; };   // development build: per-phase clocks of every workgroup behind the control words
#else
struct XgCtl { unsigned abort_flag, pad[31]; };
#endif
typedef unsigned xg_u32x4 __attribute__((ext_vector_type(4)));
// Two cache policies for the granules, chosen per launch by the participants themselves:
//   SAME_XCD  every participant reads its XCC id at start and publishes it; if all are equal, the workgroups share ONE L2, which is
//             then the point of coherence: plain stores (write-through L1 -> L2, the line stays in L2) and sc1 loads (past L1, served
//             by L2) -- a hop is an L2 round trip (tools/ubench/xcd_barrier.hip, `plain-stores 1 stride 8`: 0 stale values in 20,000
//             rounds; the same run across XCDs reads stale data at once, which is why this mode is only taken on the ids' evidence)
//   otherwise sc0 sc1 on both sides (system scope: served by memory, never by an L2 line) -- correct for any placement, and three
//             times slower per hop (16.4 us per iteration on the 6,591-dof mesh when it was the only mode)
__device__ __forceinline__ void xg_put(char *gb, int slot, double v, unsigned tag, bool same_xcd)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    xg_u32x4 g;
    g.x = (unsigned)b; g.y = (unsigned)(b >> 32); g.z = tag; g.w = g.x ^ g.y ^ tag ^ XG_KEY;
    const unsigned a = 16u * (unsigned)slot;              // (base in scalar registers + a 32-bit byte offset: the buffer is far below 4 GB)
    if (same_xcd) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(a), "v"(g), "s"(gb) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(a), "v"(g), "s"(gb) : "memory");
}
__device__ __forceinline__ bool xg_ok(const xg_u32x4 g, unsigned tag) { return g.z == tag && g.w == (g.x ^ g.y ^ tag ^ XG_KEY); }
__device__ __forceinline__ double xg_val(const xg_u32x4 g) { return __builtin_bit_cast(double, ((unsigned long long)g.y << 32) | g.x); }
// The U loads of a poll round and their wait as ONE asm statement per U.  Hand-written: to the compiler a buffer-load builtin is a pure
// read of an unchanging address, and it hoisted all U of them out of the poll loop (neither the builtin's volatile bit nor a memory
// clobber stopped it) -- the bug of this kernel's first day: every wave but the one that had stored the granule itself spun on a
// register.  One statement: with a statement per load the compiler is free to copy a register a load has been issued into before the
// wait (it did, once the loads became conditional: the copies held the registers' old contents).
template <int U> struct XgLoad;
// (addresses: the granule buffer's base in scalar registers + a 32-bit byte offset per lane)
template <> struct XgLoad<1> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[1], const char *gb, const unsigned (&a)[1], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %1, %2 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]) : "v"(a[0]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]) : "v"(a[0]), "s"(gb) : "memory");
} };
template <> struct XgLoad<2> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[2], const char *gb, const unsigned (&a)[2], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %2, %4 sc1\n\tglobal_load_dwordx4 %1, %3, %4 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]) : "v"(a[0]), "v"(a[1]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %2, %4 sc0 sc1\n\tglobal_load_dwordx4 %1, %3, %4 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]) : "v"(a[0]), "v"(a[1]), "s"(gb) : "memory");
} };
template <> struct XgLoad<3> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[3], const char *gb, const unsigned (&a)[3], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %3, %6 sc1\n\tglobal_load_dwordx4 %1, %4, %6 sc1\n\tglobal_load_dwordx4 %2, %5, %6 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %3, %6 sc0 sc1\n\tglobal_load_dwordx4 %1, %4, %6 sc0 sc1\n\tglobal_load_dwordx4 %2, %5, %6 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "s"(gb) : "memory");
} };
template <> struct XgLoad<4> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[4], const char *gb, const unsigned (&a)[4], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %4, %8 sc1\n\tglobal_load_dwordx4 %1, %5, %8 sc1\n\tglobal_load_dwordx4 %2, %6, %8 sc1\n\tglobal_load_dwordx4 %3, %7, %8 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %4, %8 sc0 sc1\n\tglobal_load_dwordx4 %1, %5, %8 sc0 sc1\n\tglobal_load_dwordx4 %2, %6, %8 sc0 sc1\n\tglobal_load_dwordx4 %3, %7, %8 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "s"(gb) : "memory");
} };
template <> struct XgLoad<6> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[6], const char *gb, const unsigned (&a)[6], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %6, %12 sc1\n\tglobal_load_dwordx4 %1, %7, %12 sc1\n\tglobal_load_dwordx4 %2, %8, %12 sc1\n\tglobal_load_dwordx4 %3, %9, %12 sc1\n\tglobal_load_dwordx4 %4, %10, %12 sc1\n\tglobal_load_dwordx4 %5, %11, %12 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %6, %12 sc0 sc1\n\tglobal_load_dwordx4 %1, %7, %12 sc0 sc1\n\tglobal_load_dwordx4 %2, %8, %12 sc0 sc1\n\tglobal_load_dwordx4 %3, %9, %12 sc0 sc1\n\tglobal_load_dwordx4 %4, %10, %12 sc0 sc1\n\tglobal_load_dwordx4 %5, %11, %12 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "s"(gb) : "memory");
} };
template <> struct XgLoad<8> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[8], const char *gb, const unsigned (&a)[8], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %8, %16 sc1\n\tglobal_load_dwordx4 %1, %9, %16 sc1\n\tglobal_load_dwordx4 %2, %10, %16 sc1\n\tglobal_load_dwordx4 %3, %11, %16 sc1\n\tglobal_load_dwordx4 %4, %12, %16 sc1\n\tglobal_load_dwordx4 %5, %13, %16 sc1\n\tglobal_load_dwordx4 %6, %14, %16 sc1\n\tglobal_load_dwordx4 %7, %15, %16 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %8, %16 sc0 sc1\n\tglobal_load_dwordx4 %1, %9, %16 sc0 sc1\n\tglobal_load_dwordx4 %2, %10, %16 sc0 sc1\n\tglobal_load_dwordx4 %3, %11, %16 sc0 sc1\n\tglobal_load_dwordx4 %4, %12, %16 sc0 sc1\n\tglobal_load_dwordx4 %5, %13, %16 sc0 sc1\n\tglobal_load_dwordx4 %6, %14, %16 sc0 sc1\n\tglobal_load_dwordx4 %7, %15, %16 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "s"(gb) : "memory");
} };
template <> struct XgLoad<10> { static __device__ __forceinline__ void run(xg_u32x4 (&g)[10], const char *gb, const unsigned (&a)[10], bool same_xcd) {
    if (same_xcd) asm volatile("global_load_dwordx4 %0, %10, %20 sc1\n\tglobal_load_dwordx4 %1, %11, %20 sc1\n\tglobal_load_dwordx4 %2, %12, %20 sc1\n\tglobal_load_dwordx4 %3, %13, %20 sc1\n\tglobal_load_dwordx4 %4, %14, %20 sc1\n\tglobal_load_dwordx4 %5, %15, %20 sc1\n\tglobal_load_dwordx4 %6, %16, %20 sc1\n\tglobal_load_dwordx4 %7, %17, %20 sc1\n\tglobal_load_dwordx4 %8, %18, %20 sc1\n\tglobal_load_dwordx4 %9, %19, %20 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]), "=&v"(g[8]), "=&v"(g[9]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "s"(gb) : "memory");
    else asm volatile("global_load_dwordx4 %0, %10, %20 sc0 sc1\n\tglobal_load_dwordx4 %1, %11, %20 sc0 sc1\n\tglobal_load_dwordx4 %2, %12, %20 sc0 sc1\n\tglobal_load_dwordx4 %3, %13, %20 sc0 sc1\n\tglobal_load_dwordx4 %4, %14, %20 sc0 sc1\n\tglobal_load_dwordx4 %5, %15, %20 sc0 sc1\n\tglobal_load_dwordx4 %6, %16, %20 sc0 sc1\n\tglobal_load_dwordx4 %7, %17, %20 sc0 sc1\n\tglobal_load_dwordx4 %8, %18, %20 sc0 sc1\n\tglobal_load_dwordx4 %9, %19, %20 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]), "=&v"(g[8]), "=&v"(g[9]) : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]), "s"(gb) : "memory");
} };
// Wave-wide poll of UL + UR granules per lane (slot < 0: nothing wanted) until every wanted one carries `tag`; false: timed out / aborted.
// The first UL slots are the LATE ones (chunk partials: the last thing their producers publish), the other UR the EARLY ones (rows).
// A round is straight-line: the loads (a lane that wants nothing in a slot reads granule 0), one wait, the checks, one vote.  The early
// group is asked for until the whole wave has all of it -- kept then, and the rounds after that load the late group only; nothing else is
// remembered.  (A granule that carried the tag keeps it until this workgroup has moved on -- the hazard note above --, so a second read is
// harmless.)  Two earlier forms: a "still wanted" flag per slot and lane, asking only for those, compiled to ~50 scalar mask operations
// per slot and round (500 instructions per round of the ten-slot poll, against the ~1,000 clocks of the round trip itself); asking for
// everything every round made the rounds as long as their bytes take through the compute unit's one load path (8 waves x 10 KB per
// round at 64 B per clock): 6.7 us per iteration against 5.4.
template <int UL, int UR>
__device__ __forceinline__ bool xg_get(const char *gb, const int (&slot)[UL + UR], unsigned tag, double (&out)[UL + UR], XgCtl *ctl, unsigned where, bool same_xcd)
{
    constexpr int U = UL + UR;
    unsigned addr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { addr[u] = 16u * (unsigned)max(slot[u], 0); out[u] = 0.0; }
    bool early = UR > 0;                                  // wave-uniform: the early group is still being asked for
    unsigned long long t0 = 0;
    for (unsigned spins = 0;; ++spins) {
        bool all = true;
        if (early) {
            xg_u32x4 g[U];
            XgLoad<U>::run(g, gb, addr, same_xcd);
            bool allr = true;
#pragma unroll
            for (int u = UL; u < U; ++u) allr = allr && (slot[u] < 0 || xg_ok(g[u], tag));
            if (__all(allr)) {
                early = false;
#pragma unroll
                for (int u = UL; u < U; ++u) out[u] = slot[u] >= 0 ? xg_val(g[u]) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) { all = all && (slot[u] < 0 || xg_ok(g[u], tag)); out[u] = slot[u] >= 0 ? xg_val(g[u]) : 0.0; }
        } else if constexpr (UL > 0) {
            xg_u32x4 g[UL];
            unsigned al[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) al[u] = addr[u];
            XgLoad<UL>::run(g, gb, al, same_xcd);
#pragma unroll
            for (int u = 0; u < UL; ++u) { all = all && (slot[u] < 0 || xg_ok(g[u], tag)); out[u] = slot[u] >= 0 ? xg_val(g[u]) : 0.0; }
        }
        if (!early && __all(all)) return true;
        // Bounded by wall time: a participant that has not been given a compute unit yet -- the chip full of other streams' work whose small
        // workgroups keep taking the seats this kernel's large ones need: seen for seconds under three threads of 64-frame extraction
        // batches -- is waited for 2 ms; then the launch gives up (nothing of it has reached x, r or p) and the host makes the iterations
        // good on the launch-per-phase path (xg_recover).  (Until round 5 the bound was a spin count worth seconds, and the call failed.)
        // (20 ms at first: 1,620 solves in 6 s beside that load, 10 % of the launches giving up; 2 ms: 2,160 solves, 12 %.)
        bool late = false;
        if ((spins & 63u) == 63u) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            late = now - t0 > XG_TIMEOUT;
        }
        if (late || ((spins & 255u) == 255u && __hip_atomic_load(&ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
            if (late) {                 // the first to time out says where (development aid: fem_debug_xcd reads the word)
                unsigned expect = 0;
                __hip_atomic_compare_exchange_strong(&ctl->abort_flag, &expect, where | 0x80000000u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) out[u] = 0.0;
            return false;
        }
        // (no s_sleep between rounds: a round is one L2 round trip, ~1,000 clocks, and the 64 clocks of an s_sleep 1 were 1.7 % of an iteration)
    }
}
// Workgroup barrier for data that lives in LDS only: __syncthreads() is a workgroup-scope fence as well, i.e. `s_waitcnt vmcnt(0)` in front
// of the s_barrier -- every barrier of an iteration then waited for the granule stores in flight to be acknowledged (the row puts behind
// the product: 5,000 of an iteration's 21,000 clocks).  The waves of k_fem_cg_xcd exchange LDS contents only.
__device__ __forceinline__ void xg_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Granule buffer of a model: K p rows | r rows | p.Ap partials x 2 parities | r.z partials x 2 | r.r partials x 2 | XCC ids | the restriction's 2 x 48 sums x 2 parities
struct XgLayout { int ap, r, pap, rz, rr, id, cw, total; };
__host__ __device__ inline XgLayout xg_layout(int ndof, int nchunk, int nchunk_s)
{
    XgLayout l;
    l.ap = 0; l.r = ndof; l.pap = 2 * ndof; l.rz = l.pap + 2 * nchunk_s; l.rr = l.rz + 2 * nchunk; l.id = l.rr + 2 * nchunk; l.cw = l.id + XG_MAXP; l.total = l.cw + 4 * 48;
    return l;
}
// MC: the most SpMV chunks a workgroup owns (1, 3 or 6: small meshes do not pay for six chunks' worth of unrolled code).
// COARSE: the two-level preconditioner inside the launch, as a THIRD hop per iteration.  Aggregate a belongs to workgroup a % P: with
// the r.z partials it also polls the r rows of its aggregate's nodes (into LDS, in the aggregate list's order) and restricts them as
// cz_apply_block does -- the two waves (a, a + 8) of that 1024-thread workgroup are two waves here, same lanes, same order: the same
// twelve sums bit for bit --, and publishes them; every workgroup then polls the 96 sums, forms v = Ac^-1 w and w.v (cz_apply_block's
// wave 0, word for word) and Z v row by row over its column range from the by-node table (a closed form per node: the same bits as
// the list walk).  r.z = the partials + w.v, p = (r/diag + Z v) + beta p: k_fem_cg_dir's expressions.  Bit-identical to the
// launch-per-phase two-level path of a single mesh (k_fem_cz_apply between k_fem_cg_update and k_fem_cg_dir).
// (A first version had every workgroup stage ALL of r and run the whole correction itself: no third hop, 28 x 105 KB of granules per
// iteration through one L2 and sixteen waves' work on four -- 22.9 us per iteration against the launch-per-phase path's 17.9.)
template <int SPB, int MC, bool COARSE>
__global__ __launch_bounds__(CGT, MC <= 3 ? 2 : 1) void k_fem_cg_xcd(const float *__restrict__ vals_b, const int *__restrict__ bcol3, const int *__restrict__ bp,
                                                    int ndof, int nchunk, int nchunk_s, int niter, int cur, CgScal *__restrict__ sc,
                                                    double *__restrict__ p, const double *__restrict__ dinv, double *__restrict__ x,
                                                    double *__restrict__ r, void *__restrict__ gran, const int4 *__restrict__ plan, int P,
                                                    int ldr, int ldq, XgCtl *__restrict__ ctl, unsigned base, const float4 *__restrict__ cz,
                                                    const int *__restrict__ czptr, const float4 *__restrict__ cznode, const double *__restrict__ aci, int lda)
{
    static_assert(CGT == RPB, "one row of the vector chunk per thread");
    if (blockIdx.x % XG_STRIDE != 0 || (int)(blockIdx.x / XG_STRIDE) >= P) return;
    extern __shared__ __align__(16) double lds[];
    __shared__ double sh[MC][CGT / 64], shv[2][CGT / 64];
    __shared__ int s_bp[MC][SPB / 3 + 1];
    __shared__ int s_fail;
    const int rank = blockIdx.x / XG_STRIDE, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int4 pl = plan[rank];
    const int c0 = pl.x, nch = (pl.y & 0xffff) - pl.x, vch = (pl.y >> 16) - 1, lo = pl.z, rng = pl.w - pl.z;   // vch: the own vector chunk, or -1
    const XgLayout L = xg_layout(ndof, nchunk, nchunk_s);
    char *gb = static_cast<char *>(gran);
    double *p_s = lds, *d_s = lds + ldr, *part = lds + 2 * ldr;        // part: MC regions of 3 ldq doubles (+ the zero slot, + 1 of padding)
    // COARSE: the own aggregate's r, the coarse vectors, and per row of the column range what the prolongation needs (first coarse dof of the row's
    // aggregate | component << 8 | constrained << 10; the two components of q the row's rotation term multiplies)
    double *r_a = part + 3 * MC * ldq + 2, *s_w = r_a + lda, *s_v = s_w + 2 * CZ_NC;      // r_a: the own aggregate's r rows (3 per node, list order); s_w[2][48], s_v[48 + 1]
    float2 *rq = reinterpret_cast<float2 *>(s_v + CZ_NC + 2);
    int *rinfo = reinterpret_cast<int *>(rq + ldr);
    double *a_s = reinterpret_cast<double *>(rinfo + ((ldr + 1) & ~1));   // Ac^-1 (48 x 48): read from memory by wave 0's solve it was 48 round trips, 14,000 clocks per iteration
    // the blocks of this thread, for the whole launch
    float va[MC][XG_MAXQ][9]; int ca[MC][XG_MAXQ]; int nqa[MC], r0a[MC], r1a[MC];
    bool wide = false;                                    // some chunk of this workgroup has more than CGT blocks
#pragma unroll
    for (int ch = 0; ch < MC; ++ch) {
        nqa[ch] = 0; r0a[ch] = r1a[ch] = 0;
#pragma unroll
        for (int u = 0; u < XG_MAXQ; ++u) ca[ch][u] = 0;
        if (ch < nch) {
            const int r0 = (c0 + ch) * SPB, r1 = min(r0 + SPB, ndof);
            const int q0 = bp[r0 / 3], nq = bp[r1 / 3] - q0;
            if (tid <= (r1 - r0) / 3) s_bp[ch][tid] = bp[r0 / 3 + tid] - q0;
            nqa[ch] = nq; r0a[ch] = r0; r1a[ch] = r1;
            wide = wide || nq > CGT;
#pragma unroll
            for (int u = 0; u < XG_MAXQ; ++u) {
                const int qq = min(tid + u * CGT, nq - 1);
                __builtin_memcpy(va[ch][u], vals_b + 9 * (size_t)(q0 + qq), 36);
                ca[ch][u] = bcol3[q0 + qq] - lo;
            }
        }
    }
    for (int i = tid; i < rng; i += CGT)
        if (lo + i < ndof) { p_s[i] = p[lo + i]; d_s[i] = dinv[lo + i]; }
    if (tid == 0) s_fail = __hip_atomic_load(&ctl->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // an earlier launch gave up: this one must not touch the state either
    __syncthreads();
    if (s_fail) return;
    if constexpr (COARSE) {
        for (int i = tid; i < CZ_NC * CZ_NC; i += CGT) a_s[i] = aci[i];
        for (int i = tid; i < rng; i += CGT)
            if (lo + i < ndof) {
                const int g = lo + i, n = g / 3, kk = g - 3 * n;
                const float4 e = cznode[n];
                const unsigned id = __float_as_uint(e.w);
                const float q[3] = {e.x, e.y, e.z};
                rinfo[i] = (int)(6u * (id & 0x0fffffffu)) | kk << 8 | (int)((id >> (28 + kk)) & 1u) << 10;
                rq[i] = float2{q[(kk + 2) % 3], q[(kk + 1) % 3]};
            }
    }
    const int row = max(vch, 0) * RPB + tid;
    const bool vec = vch >= 0, has = vec && row < ndof;
    double xv = has ? x[row] : 0, rv = has ? r[row] : 0;
    const double dv = has ? dinv[row] : 0;
    double rz = sc[0].rz[cur & 1], rr = sc[0].rr;
    // where the participants run: every one publishes its XCC id (system scope), every one reads all of them
    bool fast;
    {
        if (tid == 0) xg_put(gb, L.id + rank, (double)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf), base, false);   // HW_REG_XCC_ID, bits 0-3
        int slot[1]; double got[1];
        slot[0] = lane < P ? L.id + lane : -1;
        if (!xg_get<1, 0>(gb, slot, base, got, ctl, 4u | (unsigned)rank << 8 | (unsigned)w << 16, false)) s_fail = 1;
        const double mine = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(__builtin_bit_cast(unsigned long long, got[0]) >> 32)) << 32) |
                                                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__builtin_bit_cast(unsigned long long, got[0])));   // rank 0's id
        fast = !__any(lane < P && got[0] != mine) && !(cur & 2);
    }
    __syncthreads();
    if (s_fail) return;
    const int sub = tid / LPR, sl = tid % LPR;
    // the row groups of this thread (chunk ch, pass): LDS offset of the row's first partial, its blocks, its row -- fixed for the launch.
    // Groups that do not exist (ch >= nch, rows past the chunk) read slot 0 of the partials and are never used.
    constexpr int NPASS = (SPB + CGT / LPR - 1) / (CGT / LPR);
    // goff: LDS index of the partials sl, sl + 8, sl + 16, sl + 24 of the row (past the row's end: `zero`, a double that stays 0.0, so the
    // sum needs neither a mask nor a branch); gmore: the row has more than 32 blocks (wave-wide: anymore).
    int goff[MC][NPASS][4], gbase[MC][NPASS], gnb[MC][NPASS], grow[MC][NPASS];
    const int zero = 3 * MC * ldq;            // part[zero]: one spare double behind the chunks' regions
    if (tid == 0) part[zero] = 0.0;
    bool more = false, more2 = false;
#pragma unroll
    for (int ch = 0; ch < MC; ++ch)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
            gbase[ch][pass] = zero; gnb[ch][pass] = 0; grow[ch][pass] = -1;
            const int rw = r0a[ch] + pass * (CGT / LPR) + sub;
            if (ch < nch && rw < r1a[ch]) {
                const int I = (rw - r0a[ch]) / 3, i = rw - r0a[ch] - 3 * I, b0 = s_bp[ch][I];
                gbase[ch][pass] = 3 * (ch * ldq + b0) + i; gnb[ch][pass] = s_bp[ch][I + 1] - b0; grow[ch][pass] = rw;
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) goff[ch][pass][jj] = sl + LPR * jj < gnb[ch][pass] ? gbase[ch][pass] + 3 * (sl + LPR * jj) : zero;
            more = more || gnb[ch][pass] > 4 * LPR;
            more2 = more2 || gnb[ch][pass] > 2 * LPR;
        }
    const bool anymore = __any(more) != 0;
    const bool deep = __any(more2) != 0;              // some row of this wave has more than 16 blocks
    xg_sync();
#ifdef XG_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();
#define XG_T(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define XG_T(k) do { } while (0)
#endif
    for (int it = 0; it < niter; ++it) {
        const unsigned tagA = base + 3u * (unsigned)it + 1u, tagB = tagA + 1u, tagC = tagA + 2u;   // (tagC: the two-level form's third hop)
        const int par = it & 1;
        // ---- K p on the own chunks (k_fem_spmv, phase 1: a lane per block, three row sums parked in LDS)
        // No lane masks: the lanes past a chunk's run hold its LAST block (values and column were loaded with a clamped index) and store that
        // block's three sums once more to the same slots, as k_fem_spmv's do; the tests left are workgroup-uniform (scalar branches), so
        // the LDS reads of all the blocks are in flight together.  The values stay FLOATS in registers: the empty asm statement makes them
        // "new" every iteration -- without it the compiler hoisted the 54 conversions out of the loop, kept 108 registers of doubles, spilled
        // 18 of them and read those back from scratch one `s_waitcnt vmcnt(0)` at a time (which also waited for the previous phase's stores).
        double pin[MC][XG_MAXQ][3];
#pragma unroll
        for (int ch = 0; ch < MC; ++ch)                   // (all the reads first -- chunks that do not exist read p[0..2] --, so that no block waits for its own)
#pragma unroll
            for (int u = 0; u < XG_MAXQ; ++u) {
                if (u > 0 && !wide) continue;
                const double *pp = p_s + ca[ch][u];
                pin[ch][u][0] = pp[0]; pin[ch][u][1] = pp[1]; pin[ch][u][2] = pp[2];
            }
#pragma unroll
        for (int ch = 0; ch < MC; ++ch) {
            if (ch < nch) {
#pragma unroll
                for (int u = 0; u < XG_MAXQ; ++u) {
                    if (u > 0 && u * CGT >= nqa[ch]) continue;
                    const int qq = min(tid + u * CGT, nqa[ch] - 1);
                    const double p0 = pin[ch][u][0], p1 = pin[ch][u][1], p2 = pin[ch][u][2];
                    double *dst = part + 3 * (ch * ldq + qq);
#pragma unroll
                    for (int i = 0; i < 9; ++i) asm volatile("" : "+v"(va[ch][u][i]));
#pragma unroll
                    for (int i = 0; i < 3; ++i)
                        dst[i] = ((double)va[ch][u][3 * i] * p0 + (double)va[ch][u][3 * i + 1] * p1) + (double)va[ch][u][3 * i + 2] * p2;
                }
            }
        }
        xg_sync();
        XG_T(0);   // SpMV phase 1
        // (phase 2: 8 lanes per row, DPP row_shl sums in a fixed order; the chunk's partial of p.Ap)
        // No branch, no mask and no loop in here: the <= 48 partials a thread adds are asked for together (a first version -- k_fem_spmv's
        // loop over a row's blocks, under `if (row < r1)` -- compiled to 72 LDS reads each waited for behind its own branch: 11,000 of
        // an iteration's 21,000 clocks; a second one with selects kept 48 loop-invariant lane masks in spilled scalar registers: 9,000).
        // A lane adds the partials sl, sl + 8, .. of its row in that order; the slots past the row's end read a 0.0, and `+ 0.0` changes
        // nothing (a sum that starts at +0.0 never is -0.0).  Rows of more than 32 blocks take the loop for the rest (wave-uniform test).
        double acc[MC], srow[MC][NPASS];
        auto row_sums = [&](auto nj_) {                   // NJ = 2: no row of this wave has more than 16 blocks, the partials 16.. are not asked for at all
            constexpr int NJ = decltype(nj_)::value;
            double pv[MC][NPASS][NJ], pr[MC][NPASS];
#pragma unroll
            for (int ch = 0; ch < MC; ++ch)
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) pv[ch][pass][jj] = part[goff[ch][pass][jj]];
                    pr[ch][pass] = p_s[max(grow[ch][pass], lo) - lo];
                }
#pragma unroll
            for (int ch = 0; ch < MC; ++ch) {
                acc[ch] = 0;
#pragma unroll
                for (int pass = 0; pass < NPASS; ++pass) {
                    double sm = 0;
#pragma unroll
                    for (int jj = 0; jj < NJ; ++jj) sm += pv[ch][pass][jj];
                    // (NJ == 2: the two slots left out held +0.0, and sm -- a sum that began at +0.0 -- is never -0.0: adding them changed no bit)
                    if (NJ == 4 && anymore)
                        for (int j = sl + 4 * LPR; j < gnb[ch][pass]; j += LPR) sm += part[gbase[ch][pass] + 3 * j];
                    sm += dpp_shl_f64<4>(sm);
                    sm += dpp_shl_f64<2>(sm);
                    sm += dpp_shl_f64<1>(sm);
                    srow[ch][pass] = sm;
                    acc[ch] += grow[ch][pass] >= 0 && sl == 0 ? pr[ch][pass] * sm : 0.0;
                }
                acc[ch] = wave_sum_f64_lanes08(acc[ch]);      // block_sum, all chunks behind one barrier
            }
        };
        if (deep) row_sums(std::integral_constant<int, 4>{}); else row_sums(std::integral_constant<int, 2>{});
        if constexpr (!COARSE) XG_T(5);
        // (the rows go out behind the sums, not between them: a store is a hand-written asm statement with a memory clobber, and ten of
        // them inside the loop above made ten chains of LDS reads run one after the other -- 10,600 of an iteration's 20,700 clocks)
#pragma unroll
        for (int ch = 0; ch < MC; ++ch)
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass)
                if (grow[ch][pass] >= 0 && sl == 0) xg_put(gb, L.ap + grow[ch][pass], srow[ch][pass], tagA, fast);
        if (lane == 0) {
#pragma unroll
            for (int ch = 0; ch < MC; ++ch) if (ch < nch) sh[ch][w] = acc[ch];
        }
        if constexpr (!COARSE) XG_T(6);
        xg_sync();
        if constexpr (!COARSE) XG_T(7);
        if (tid < nch) {                                  // a lane per chunk (one thread for all of them: three chains of reads and adds in a row)
            double t = 0;
#pragma unroll
            for (int i = 0; i < CGT / 64; ++i) t += sh[tid][i];
            xg_put(gb, L.pap + par * nchunk_s + c0 + tid, t, tagA, fast);
        }
        XG_T(1);   // SpMV phase 2 + partials out
        // ---- alpha = rz / p.Ap; x += alpha p; r -= alpha K p; partials of r.(r/diag) and r.r (k_fem_cg_update)
        double pAp, api;
        {
            int slot[4]; double got[4];
#pragma unroll
            for (int u = 0; u < 3; ++u) slot[u] = lane + 64 * u < nchunk_s ? L.pap + par * nchunk_s + lane + 64 * u : -1;
            slot[3] = has ? L.ap + row : -1;
            if (!xg_get<3, 1>(gb, slot, tagA, got, ctl, 1u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
            double v = 0;
#pragma unroll
            for (int u = 0; u < 3; ++u) if (lane + 64 * u < nchunk_s) v += got[u];      // chunk_sum's order
            pAp = wave_sum_f64(v);
            api = got[3];
        }
        XG_T(2);   // hop A: the p.Ap partials and the own K p row
        const double alpha = cg_ratio(rz, pAp);
        if (vec) {
            double s1 = 0, s2 = 0;
            if (has) {
                xv += alpha * p_s[row - lo];
                const double ri = rv - alpha * api;
                rv = ri;
                s1 += ri * (ri * dv);
                s2 += ri * ri;
                xg_put(gb, L.r + row, ri, tagB, fast);
            }
            s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);       // two block_sums behind one barrier
            if (lane == 0) { shv[0][w] = s1; shv[1][w] = s2; }
        }
        xg_sync();
        if (s_fail) return;
        if (vec && tid < 2) {                             // lane 0: the r.(r/diag) partial, lane 1: r.r
            double t = 0;
#pragma unroll
            for (int i = 0; i < CGT / 64; ++i) t += shv[tid][i];
            xg_put(gb, (tid ? L.rr : L.rz) + par * nchunk + vch, t, tagB, fast);
        }
        // ---- beta = rz_new / rz; p = r/diag + beta p over the own column range, from the r everybody published (k_fem_cg_dir)
        XG_T(3);   // update + partials out
        // one poll for the chunk partials and the first XG_SU x 256 rows of the range (ranges beyond that -- irregular numberings --
        // take further rounds of rows only)
        double rz2, beta;
        if constexpr (!COARSE) {
            int slot[2 + XG_SU]; double got[2 + XG_SU];
            slot[0] = lane < nchunk ? L.rz + par * nchunk + lane : -1;
            slot[1] = lane < nchunk ? L.rr + par * nchunk + lane : -1;
#pragma unroll
            for (int u = 0; u < XG_SU; ++u) { const int i = u * CGT + tid; slot[2 + u] = i < rng && lo + i < ndof ? L.r + lo + i : -1; }
            if (!xg_get<2, XG_SU>(gb, slot, tagB, got, ctl, 2u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
            rz2 = wave_sum_f64(lane < nchunk ? got[0] : 0.0);       // chunk_sum with <= 32 partials: a lane per partial
            rr = wave_sum_f64(lane < nchunk ? got[1] : 0.0);
            beta = cg_ratio(rz2, rz);
#pragma unroll
            for (int u = 0; u < XG_SU; ++u) { const int i = u * CGT + tid; if (slot[2 + u] >= 0) p_s[i] = got[2 + u] * d_s[i] + beta * p_s[i]; }
            for (int b0 = XG_SU * CGT; b0 < rng; b0 += XG_SU * CGT) {
                int slot2[XG_SU]; double got2[XG_SU];
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) { const int i = b0 + u * CGT + tid; slot2[u] = i < rng && lo + i < ndof ? L.r + lo + i : -1; }
                if (!xg_get<0, XG_SU>(gb, slot2, tagB, got2, ctl, 3u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) { const int i = b0 + u * CGT + tid; if (slot2[u] >= 0) p_s[i] = got2[u] * d_s[i] + beta * p_s[i]; }
            }
        } else {
            constexpr int XG_SA = 4;                          // aggregate rows per thread and poll (<= 1,024 rows = 341 nodes; larger: more rounds)
            // hop B: the chunk partials, the column range's r rows and -- aggregate owners -- the first rows of the own aggregate
            const bool owner = rank < CZ_NA;                  // aggregate a = rank, rank + P, .. (P < 8: several per workgroup)
            // An aggregate's owner asks for the partials and its aggregate's rows first and restricts -- every workgroup waits for those sums --
            // and for the rows of its own column range afterwards, when they have long arrived; the others ask for partials and range at once.
            double rgot[XG_SU];
            if (owner) {
                int slot[2 + XG_SA]; double got[2 + XG_SA];
                slot[0] = lane < nchunk ? L.rz + par * nchunk + lane : -1;
                slot[1] = lane < nchunk ? L.rr + par * nchunk + lane : -1;
                const int zp0 = czptr[rank], na3 = 3 * (czptr[rank + 1] - zp0);
#pragma unroll
                for (int u = 0; u < XG_SA; ++u) {
                    const int e = u * CGT + tid;
                    slot[2 + u] = e < na3 ? L.r + 3 * (int)(__float_as_uint(cz[zp0 + e / 3].w) & 0x0fffffffu) + e % 3 : -1;
                }
                if (!xg_get<2, XG_SA>(gb, slot, tagB, got, ctl, 2u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
                rz2 = wave_sum_f64(lane < nchunk ? got[0] : 0.0);
                rr = wave_sum_f64(lane < nchunk ? got[1] : 0.0);
#pragma unroll
                for (int u = 0; u < XG_SA; ++u) { const int e = u * CGT + tid; if (e < na3) r_a[e] = got[2 + u]; }
            } else {
                int slot[2 + XG_SU]; double got[2 + XG_SU];
                slot[0] = lane < nchunk ? L.rz + par * nchunk + lane : -1;
                slot[1] = lane < nchunk ? L.rr + par * nchunk + lane : -1;
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) { const int i = u * CGT + tid; slot[2 + u] = i < rng && lo + i < ndof ? L.r + lo + i : -1; }
                if (!xg_get<2, XG_SU>(gb, slot, tagB, got, ctl, 2u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
                rz2 = wave_sum_f64(lane < nchunk ? got[0] : 0.0);
                rr = wave_sum_f64(lane < nchunk ? got[1] : 0.0);
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) rgot[u] = got[2 + u];
            }
            XG_T(5);   // (two-level) hop B's poll
            // restriction of the owned aggregates: cz_apply_block's waves (a, half = 0) and (a, half = 1) are waves 0 and 1 here
            for (int ag = rank; ag < CZ_NA; ag += P) {
                const int zp0 = czptr[ag], zp1 = czptr[ag + 1], na3 = 3 * (zp1 - zp0);
                for (int e0 = (ag == rank ? XG_SA : 0) * CGT; e0 < na3; e0 += XG_SA * CGT) {       // what the first poll did not cover
                    if (ag != rank && e0 == 0) xg_sync();                                          // (the previous aggregate's rows have been used)
                    int slot[XG_SA]; double got[XG_SA];
#pragma unroll
                    for (int u = 0; u < XG_SA; ++u) {
                        const int e = e0 + u * CGT + tid;
                        slot[u] = e < na3 ? L.r + 3 * (int)(__float_as_uint(cz[zp0 + e / 3].w) & 0x0fffffffu) + e % 3 : -1;
                    }
                    if (!xg_get<0, XG_SA>(gb, slot, tagB, got, ctl, 5u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
#pragma unroll
                    for (int u = 0; u < XG_SA; ++u) { const int e = e0 + u * CGT + tid; if (e < na3) r_a[e] = got[u]; }
                }
                xg_sync();                                    // the aggregate's rows are in LDS
                if (w < 2) {
                    const int half = w;
                    double w6[6] = {0, 0, 0, 0, 0, 0};
                    constexpr int U = 4;
                    for (int q0 = zp0 + 64 * half + lane; q0 - lane < zp1; q0 += 128 * U) {
                        float4 e[U]; double g[U][3];
#pragma unroll
                        for (int u = 0; u < U; ++u) e[u] = cz[min(q0 + 128 * u, zp1 - 1)];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const double *rs_ = r_a + 3 * (min(q0 + 128 * u, zp1 - 1) - zp0);
                            g[u][0] = rs_[0]; g[u][1] = rs_[1]; g[u][2] = rs_[2];
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const CzNode n = cz_node(e[u]);
                            const bool in = q0 + 128 * u < zp1;
                            const double r0 = n.m0 || !in ? 0.0 : g[u][0], r1 = n.m1 || !in ? 0.0 : g[u][1], r2 = n.m2 || !in ? 0.0 : g[u][2];
                            CZ_RESTRICT_ADD(w6, n, r0, r1, r2);
                        }
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) { w6[m] = wave_sum_f64(w6[m]); if (lane == m) xg_put(gb, L.cw + par * 2 * CZ_NC + half * CZ_NC + 6 * ag + m, w6[m], tagC, fast); }
                }
            }
            if (owner) {                                      // (the column range's rows, behind the restriction)
                int slot[XG_SU];
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) { const int i = u * CGT + tid; slot[u] = i < rng && lo + i < ndof ? L.r + lo + i : -1; }
                if (!xg_get<0, XG_SU>(gb, slot, tagB, rgot, ctl, 7u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
            }
            XG_T(6);   // (two-level) restriction of the own aggregate + puts
            // hop C: the 96 sums; v = Ac^-1 w and w.v as cz_apply_block's wave 0
            if (w == 0) {
                int slot[2]; double got[2];
                slot[0] = lane < CZ_NC ? L.cw + par * 2 * CZ_NC + lane : -1;
                slot[1] = lane < CZ_NC ? L.cw + par * 2 * CZ_NC + CZ_NC + lane : -1;
                if (!xg_get<2, 0>(gb, slot, tagC, got, ctl, 6u | (unsigned)rank << 8 | (unsigned)it << 20, fast)) s_fail = 1;
                if (lane < CZ_NC) { s_w[lane] = got[0]; s_w[CZ_NC + lane] = got[1]; }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's own LDS writes, read back below
                const int kq = min(lane, CZ_NC - 1);
                double v = 0;
#pragma unroll 8
                for (int j = 0; j < CZ_NC; ++j) v += a_s[j * CZ_NC + kq] * (s_w[j] + s_w[CZ_NC + j]);
                const double t = wave_sum_f64(lane < CZ_NC ? (s_w[kq] + s_w[CZ_NC + kq]) * v : 0.0);
                if (lane < CZ_NC) s_v[kq] = v;
                if (lane == 0) s_v[CZ_NC] = t;
            }
            xg_sync();
            XG_T(7);   // (two-level) hop C + coarse solve
            rz2 += s_v[CZ_NC];                                // r.z = r.(r/diag) + w.v (k_fem_cg_dir)
            beta = cg_ratio(rz2, rz);
#pragma unroll
            for (int u = 0; u < XG_SU; ++u) {
                const int i = u * CGT + tid;
                if (i < rng && lo + i < ndof) {
                    const int info = rinfo[i], c6 = info & 0xff, kk = (info >> 8) & 3;
                    const double *va = s_v + c6;                    // the six coarse dofs of the row's aggregate
                    const float2 q = rq[i];
                    const double czr = (info >> 10) & 1 ? 0.0 : va[kk] + (va[3 + (kk + 1) % 3] * (double)q.x - va[3 + (kk + 2) % 3] * (double)q.y);
                    p_s[i] = (rgot[u] * d_s[i] + czr) + beta * p_s[i];
                }
            }
            for (int b0 = XG_SU * CGT; b0 < rng; b0 += XG_SU * CGT) {      // ranges beyond 2,048 rows (irregular numberings)
                int slot2[XG_SU]; double got2[XG_SU];
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) { const int i = b0 + u * CGT + tid; slot2[u] = i < rng && lo + i < ndof ? L.r + lo + i : -1; }
                if (!xg_get<0, XG_SU>(gb, slot2, tagB, got2, ctl, 3u | (unsigned)rank << 8 | (unsigned)w << 16 | (unsigned)it << 20, fast)) s_fail = 1;
#pragma unroll
                for (int u = 0; u < XG_SU; ++u) {
                    const int i = b0 + u * CGT + tid;
                    if (slot2[u] >= 0) {
                        const int info = rinfo[i], c6 = info & 0xff, kk = (info >> 8) & 3;
                        const double *va = s_v + c6;
                        const float2 q = rq[i];
                        const double czr = (info >> 10) & 1 ? 0.0 : va[kk] + (va[3 + (kk + 1) % 3] * (double)q.x - va[3 + (kk + 2) % 3] * (double)q.y);
                        p_s[i] = (got2[u] * d_s[i] + czr) + beta * p_s[i];
                    }
                }
            }
        }
        rz = rz2;
        xg_sync();
        XG_T(4);   // hop B: the r.z partials and the range's r rows; the new p
        if (s_fail) return;
    }
#ifdef XG_TIMING
    if (tid == 0 && rank < 3) for (int k = 0; k < 8; ++k) ctl->pad[1 + 8 * rank + k] = (unsigned)(tacc[k] / (unsigned long long)max(niter, 1));
    if (tid == 0) for (int k = 0; k < 8; ++k) ctl->pad[31 + 8 * rank + k] = (unsigned)(tacc[k] / (unsigned long long)max(niter, 1));
    if (tid == 0) ctl->pad[31 + 8 * rank + 7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID instead of the barrier's clocks
#endif
    if (has) { x[row] = xv; r[row] = rv; p[row] = p_s[row - lo]; }
    if (rank == 0 && tid == 0) { sc[0].rz[0] = rz; sc[0].rz[1] = rz; sc[0].rr = rr; ctl->pad[0] = base + 3u * (unsigned)niter; }   // (pad[0]: the tag this launch ended at = it ran to its end, xg_recover)
}

// COARSE: the two-level preconditioner inside the same launch.  After the update (r is in the batch vector) the sixteen waves sum
// the aggregates' modes from r -- waves a and a + 8 take alternate groups of 64 nodes of aggregate a's list, partials in wave order --,
// wave 0 forms v = Ac^-1 w and w.v, the waves write Z v for their nodes into Ap's slots (K p has been used), and the direction
// update adds it: three more workgroup barriers, 16 bytes per dof more through the compute unit's caches, no more launches.
template <bool COARSE>
__global__ __launch_bounds__(CGS_T) void k_fem_cg_step(int ndof, int cur, CgScal *__restrict__ sc, double *__restrict__ p,
                                                       double *__restrict__ Ap, const double *__restrict__ dinv,
                                                       double *__restrict__ x, double *__restrict__ r,
                                                       const int4 *__restrict__ minfo, const float4 *__restrict__ cz,
                                                       const int *__restrict__ czptr, const double *__restrict__ aci)
{
    __shared__ double sh[CGS_T / 64];
    __shared__ double s_w[2][CZ_NC], s_v[CZ_NC + 1];
    const int mesh = blockIdx.x;
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const int nrows = minfo ? minfo[mesh].y : ndof;
    const double rz = sc[mesh].rz[cur];
    double s1 = 0, s2 = 0;
    // z = r/diag + Z Ac^-1 Z^T r: leaves Z v in Ap[row0 ...] and returns w.v.  Called by all threads after r has been stored.
    auto coarse = [&]() -> double {
        return cz_apply_block(cz + row0 / 3, czptr + 9 * mesh, aci + (size_t)mesh * (CZ_NC * CZ_NC), r + row0, Ap + row0, false, s_w, s_v);
    };
    if (nrows <= CGS_U * CGS_T) {
        // the whole mesh in one block: p and r/diag stay in registers across the reduction -- 5 vector reads and 3 writes
        double pv[CGS_U], zv[CGS_U];
        {
            double av[CGS_U], rv[CGS_U], xv[CGS_U], dv[CGS_U];
#pragma unroll
            for (int u = 0; u < CGS_U; ++u) {   // clamped index: unconditional loads, all in flight together
                const size_t g = row0 + min(u * CGS_T + (int)threadIdx.x, nrows - 1);
                pv[u] = p[g]; av[u] = Ap[g]; rv[u] = r[g]; xv[u] = x[g]; dv[u] = dinv[g];
            }
            double s0 = 0;
#pragma unroll
            for (int u = 0; u < CGS_U; ++u)
                if (u * CGS_T + (int)threadIdx.x < nrows) s0 += pv[u] * av[u];
            const double alpha = cg_ratio(rz, block_sum(s0, sh));   // p.Ap: the SpMV leaves it to us
#pragma unroll
            for (int u = 0; u < CGS_U; ++u) {
                const int i = u * CGS_T + (int)threadIdx.x;
                const double ri = rv[u] - alpha * av[u];
                zv[u] = ri * dv[u];
                if (i < nrows) {
                    const size_t g = row0 + i;
                    x[g] = xv[u] + alpha * pv[u];
                    r[g] = ri;
                    s1 += ri * zv[u];
                    s2 += ri * ri;
                }
            }
        }
        double rz2 = block_sum(s1, sh);
        const double rr = block_sum(s2, sh);
        if constexpr (COARSE) rz2 += coarse();
        const double beta = cg_ratio(rz2, rz);
#pragma unroll
        for (int u = 0; u < CGS_U; ++u) {
            const int i = u * CGS_T + (int)threadIdx.x;
            if (i < nrows) p[row0 + i] = COARSE ? (zv[u] + Ap[row0 + i]) + beta * pv[u] : zv[u] + beta * pv[u];
        }
        if (threadIdx.x == 0) { sc[mesh].rz[cur ^ 1] = rz2; sc[mesh].rr = rr; }
        return;
    }
    double s0 = 0;
    for (int i = threadIdx.x; i < nrows; i += CGS_T) s0 += p[row0 + i] * Ap[row0 + i];
    const double alpha = cg_ratio(rz, block_sum(s0, sh));
    for (int base = 0; base < nrows; base += CGS_U * CGS_T) {
        double pv[CGS_U], av[CGS_U], rv[CGS_U], xv[CGS_U], dv[CGS_U];
#pragma unroll
        for (int u = 0; u < CGS_U; ++u) {
            const size_t g = row0 + min(base + u * CGS_T + (int)threadIdx.x, nrows - 1);
            pv[u] = p[g]; av[u] = Ap[g]; rv[u] = r[g]; xv[u] = x[g]; dv[u] = dinv[g];
        }
#pragma unroll
        for (int u = 0; u < CGS_U; ++u) {
            const int i = base + u * CGS_T + (int)threadIdx.x;
            if (i < nrows) {
                const size_t g = row0 + i;
                x[g] = xv[u] + alpha * pv[u];
                const double ri = rv[u] - alpha * av[u];
                r[g] = ri;
                s1 += ri * (ri * dv[u]);
                s2 += ri * ri;
            }
        }
    }
    double rz2 = block_sum(s1, sh);
    const double rr = block_sum(s2, sh);
    if constexpr (COARSE) rz2 += coarse();
    const double beta = cg_ratio(rz2, rz);
    for (int base = 0; base < nrows; base += CGS_U * CGS_T) {   // r[g]: this thread's own stores of the first phase
        double pv[CGS_U], rv[CGS_U], dv[CGS_U], cv[COARSE ? CGS_U : 1];
#pragma unroll
        for (int u = 0; u < CGS_U; ++u) {
            const size_t g = row0 + min(base + u * CGS_T + (int)threadIdx.x, nrows - 1);
            pv[u] = p[g]; rv[u] = r[g]; dv[u] = dinv[g];
            if constexpr (COARSE) cv[u] = Ap[g];
        }
#pragma unroll
        for (int u = 0; u < CGS_U; ++u) {
            const int i = base + u * CGS_T + (int)threadIdx.x;
            if (i < nrows) p[row0 + i] = COARSE ? (rv[u] * dv[u] + cv[u]) + beta * pv[u] : rv[u] * dv[u] + beta * pv[u];
        }
    }
    if (threadIdx.x == 0) { sc[mesh].rz[cur ^ 1] = rz2; sc[mesh].rr = rr; }
}

// Batches of meshes that fit a compute unit: ALL iterations of a mesh's CG in one launch, one 512-thread (CGR_T) workgroup per mesh
// (one per CU: the batch of 256 fills the chip).  What an iteration needs besides the matrix stays on the CU: p and Ap in LDS
// (16 bytes per dof), x, r and 1/diag in the registers of the thread that owns the row, the scalars in the workgroup.  HBM
// then streams the block-major values and ONE column index per 3 x 3 block, once per iteration, and nothing else: no vector
// traffic (a third of an iteration's bytes in the launch-per-phase form), no gathers through the texture addresser (p comes
// from LDS), no launches, and the loads of the next iteration's first blocks are in flight while the workgroup reduces.
// A wave takes chunks of <= CGR_CB blocks = whole block rows (host table: first block row, block rows, first block, blocks),
// a lane a block; the three row sums of a block are parked in the wave's own LDS slice and summed per row by one lane, in
// block order; no workgroup barrier inside the product.  Three barriers per iteration.
// BIG: meshes of up to 14,288 dofs -- what (n + 3 CGR_CB CGR_W + 6 CGR_W) x 8 B <= 160 KB of LDS leaves for p (plan_model; the register
// tiling CGR_MAXROWS_BIG would take 14,336) --: Ap and x live in the mesh's slice of the batch vectors instead
// (written and read by the same compute unit: with 1/diag + 48 bytes per dof and iteration beside the matrix's ~175), r in registers.
constexpr int CGR_T = 512, CGR_W = CGR_T / 64, CGR_NB = 4, CGR_CB = 64 * CGR_NB, CGR_MAXROWS = 7168, CGR_MAXROWS_BIG = 14336, CGR_MIN_MESHES = 64;
// COARSE (not with BIG): the two-level preconditioner inside the launch.  Wave a owns aggregate a (CGR_W = CZ_NA): after the update
// r is parked in Ap's LDS slots (K p is no longer needed), wave a sums its aggregate's six modes from there (lane-strided, xor
// butterflies) and forms ITS six columns' share of v = Ac^-1 w -- lane k holds Ac^-1[k][6a .. 6a+5] in registers for the whole launch --,
// every wave then adds the eight shares in wave order, has w.v for beta, and writes Z v for its aggregate's nodes over r in LDS,
// which the direction update adds.  Five barriers per iteration instead of three; no vector leaves the compute unit.
template <bool BIG, bool COARSE>
__global__ __launch_bounds__(CGR_T) void k_fem_cg_resident(const float *__restrict__ vals_b, const int *__restrict__ bcol3,
                                                           const int *__restrict__ bp, const int4 *__restrict__ rcd,
                                                           const int *__restrict__ rcfirst, size_t nnzs, int ndof, int ldn,
                                                           int niter, CgScal *__restrict__ sc, double *__restrict__ p,
                                                           const double *__restrict__ dinv, double *__restrict__ x,
                                                           double *__restrict__ r, double *__restrict__ Apg,
                                                           const int4 *__restrict__ minfo, const float4 *__restrict__ cz,
                                                           const int *__restrict__ czptr, const double *__restrict__ aci)
{
    static_assert(CGS_MIN_MESHES == 16 && CGR_W == CZ_NA && CZR_U == 5, "the coarse correction takes a wave per aggregate");
    extern __shared__ __align__(16) double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mesh = blockIdx.x;
    // uniform layout: shared tables numbered from 0, values at mesh * nnzs; segmented: global numbering throughout
    const size_t row0 = minfo ? (size_t)minfo[mesh].x : (size_t)mesh * ndof;
    const int nrows = minfo ? minfo[mesh].y : ndof;
    const int tabrow0 = minfo ? (int)row0 : 0;
    const int c0 = rcfirst[minfo ? mesh : 0], c1 = rcfirst[minfo ? mesh + 1 : 1];
    const float *vb = vals_b + (minfo ? (size_t)0 : (size_t)mesh * nnzs);
    constexpr int CGR_U = (BIG ? CGR_MAXROWS_BIG : CGR_MAXROWS) / CGR_T, NV = BIG ? 1 : 2;   // NV: vectors in LDS
    double *p_s = lds, *Ap_s = lds + ldn, *part = lds + NV * ldn + wave * (3 * CGR_CB), *sh = lds + NV * ldn + CGR_W * (3 * CGR_CB);
    double *xg = x + row0, *ag = Apg + row0;
    const double *dg = dinv + row0;
    double xv[BIG ? 1 : CGR_U], rv[CGR_U], dv[BIG ? 1 : CGR_U];
#pragma unroll
    for (int u = 0; u < CGR_U; ++u) {
        const int i = min(u * CGR_T + tid, nrows - 1);
        if (!BIG) { xv[u] = x[row0 + i]; dv[u] = dinv[row0 + i]; }
        rv[u] = r[row0 + i];
        if (u * CGR_T + tid < nrows) p_s[i] = p[row0 + i];
    }
    double rz = sc[mesh].rz[0], rr = sc[mesh].rr;   // both rz slots hold the current value between launches of this kernel
    // COARSE: this wave's aggregate (its nodes in cz[zp0 .. zp1)), lane k's six entries of the inverse, the scratch (in wave 0's
    // slice of the product's partials, idle between the barriers that use it): eight shares of v, then w
    const float4 *lz = COARSE ? cz + row0 / 3 : nullptr;
    const int zp0 = COARSE ? czptr[9 * mesh + wave] : 0, zp1 = COARSE ? czptr[9 * mesh + wave + 1] : 0;
    double *cz_share = lds + NV * ldn, *cz_w = cz_share + CZ_NA * CZ_NC;
    // (BIG: no registers to spare -- the six entries are read again in every iteration, from L2)
    const double *acl = COARSE ? aci + (size_t)mesh * (CZ_NC * CZ_NC) + (6 * wave) * CZ_NC + min(lane, CZ_NC - 1) : nullptr;
    double ac6[COARSE && !BIG ? 6 : 1];
    if constexpr (COARSE && !BIG) {
#pragma unroll
        for (int m = 0; m < 6; ++m) ac6[m] = acl[m * CZ_NC];
    }
    // a chunk's loads: its descriptor (wave-uniform), per lane two blocks (clamped: lanes past the chunk repeat its last
    // block) and the block-row pointer of block row `lane`
    float nva[CGR_NB][9]; int nca[CGR_NB], nbpl; int4 nd;
    auto issue = [&](int c) {
        nd = rcd[c];
#pragma unroll
        for (int u = 0; u < CGR_NB; ++u) {
            const int q = nd.z + min(lane + 64 * u, nd.w - 1);
            __builtin_memcpy(nva[u], vb + 9 * (size_t)q, 36);
            nca[u] = bcol3[q] - tabrow0;
        }
        nbpl = bp[nd.x + min(lane, nd.y)] - nd.z;
    };
    const bool any = c0 + wave < c1;
    if (any) issue(c0 + wave);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        double pap = 0;
        for (int c = c0 + wave; c < c1; c += CGR_W) {
            float va[CGR_NB][9]; int ca[CGR_NB];
            const int4 d = nd;
            const int bpl = nbpl;
#pragma unroll
            for (int u = 0; u < CGR_NB; ++u) {
                ca[u] = nca[u];
#pragma unroll
                for (int k = 0; k < 9; ++k) va[u][k] = nva[u][k];
            }
            {   // the next chunk of this wave -- across the iteration boundary too: the matrix does not wait for p
                const int cn = c + CGR_W < c1 ? c + CGR_W : c0 + wave;
                if (c + CGR_W < c1 || it + 1 < niter) issue(cn);
            }
#pragma unroll
            for (int u = 0; u < CGR_NB; ++u) {
                const double *pp = p_s + ca[u];
                const double p0 = pp[0], p1 = pp[1], p2 = pp[2];
                double *dst = part + 3 * min(lane + 64 * u, d.w - 1);
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    dst[i] = __builtin_fma((double)va[u][3 * i + 2], p2, __builtin_fma((double)va[u][3 * i + 1], p1, (double)va[u][3 * i] * p0));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // a lane per row of the chunk: its blocks' partials in block order (8 lanes per row with a shuffle tree, as k_fem_spmv
            // does it, cost this kernel twice the instructions of the products themselves)
            const int nr = 3 * d.y;
            if constexpr (BIG) {
                // Ap goes to the batch vector: ONE unconditional store per chunk (a BIG chunk has at most 21 block rows = one
                // pass; lanes past its rows repeat the last row: same value to the same address).  A store behind a branch or in
                // a loop would leave the compiler unable to count the memory operations issued after the next chunk's loads, and
                // it would then wait for everything -- this store's completion included -- at the top of every chunk
                const int row = min(lane, nr - 1), I = (row * 171) >> 9, i = row - 3 * I;
                const int b0 = __shfl(bpl, I), nb = __shfl(bpl, I + 1) - b0;
                const double *q = part + 3 * b0 + i;
                double s = 0;
#pragma unroll 4
                for (int j = 0; j < nb; ++j) s += q[3 * j];
                const int g = 3 * d.x - tabrow0 + row;
                ag[g] = s;
                pap += lane < nr ? p_s[g] * s : 0.0;
            } else
            for (int rb = 0; rb < nr; rb += 64) {
                const int row = rb + lane, I = (row * 171) >> 9, i = row - 3 * I;   // row / 3 for row < 512
                const int b0 = __shfl(bpl, I), nb = __shfl(bpl, I + 1) - b0;
                if (row < nr) {
                    const double *q = part + 3 * b0 + i;
                    double s = 0;
#pragma unroll 4
                    for (int j = 0; j < nb; ++j) s += q[3 * j];
                    const int g = 3 * d.x - tabrow0 + row;   // the mesh's own row number
                    Ap_s[g] = s;
                    pap += p_s[g] * s;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        pap = wave_sum_f64(pap);
        double *shi = sh + (it & 1) * (3 * CGR_W);     // two sets of slots: a set is rewritten two barriers after its last read
        if (lane == 0) shi[wave] = pap;
        __syncthreads();                                // Ap complete, partials of p.Ap visible
        double t = 0;
#pragma unroll
        for (int w = 0; w < CGR_W; ++w) t += shi[w];
        const double alpha = cg_ratio(rz, t);
        double s1 = 0, s2 = 0;
        double dd[BIG ? CGR_U : 1];
        int tv = tid;
        asm volatile("" : "+v"(tv));   // BIG: keeps 28 row indices, predicates and addresses from being hoisted out of the iteration loop into registers that do not exist
        if constexpr (BIG) {
            // fourteen rows at a time: their Ap, x and 1/diag requested together, the next fourteen only after these are used
            // (the fence keeps the compiler from hoisting all 84 loads into registers it does not have); 1/diag stays in
            // registers until the direction update below
#pragma unroll
            for (int sb = 0; sb < CGR_U; sb += 14) {
                double av[14], xx[14];
#pragma unroll
                for (int v = 0; v < 14; ++v) {
                    const int ic = min((sb + v) * CGR_T + tv, nrows - 1);
                    av[v] = ag[ic]; xx[v] = xg[ic]; dd[sb + v] = dg[ic];
                }
#pragma unroll
                for (int v = 0; v < 14; ++v) {
                    const int u = sb + v, i = u * CGR_T + tv;
                    const double ri = rv[u] - alpha * av[v];
                    if (i < nrows) {
                        xg[i] = xx[v] + alpha * p_s[i];
                        rv[u] = ri;
                        s1 += ri * (ri * dd[u]);
                        s2 += ri * ri;
                        if constexpr (COARSE) ag[i] = ri;   // K p has been used: its slot (in the batch vector) carries r to the restriction
                    }
                }
                asm volatile("" ::: "memory");
            }
        } else {
#pragma unroll
            for (int u = 0; u < CGR_U; ++u) {
                const int i = u * CGR_T + tid, ic = min(i, nrows - 1);
                const double ri = rv[u] - alpha * Ap_s[ic];
                if (i < nrows) {
                    xv[u] += alpha * p_s[ic];
                    rv[u] = ri;
                    s1 += ri * (ri * dv[u]);
                    s2 += ri * ri;
                    if constexpr (COARSE) Ap_s[i] = ri;   // K p has been used: its slot carries r to the restriction
                }
            }
        }
        s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
        if (lane == 0) { shi[CGR_W + wave] = s1; shi[2 * CGR_W + wave] = s2; }
        // COARSE: the wave's aggregate -- at most CZR_U x 64 nodes (plan: larger aggregates go phase by phase; config 3 has 275) -- is
        // requested before the barrier, all entries at once (a loop of unknown length around a load would also make the compiler wait
        // for ALL outstanding loads, the next chunk's matrix blocks included, wherever it loses count)
        float4 ce[COARSE && !BIG ? CZR_U : 1];
        if constexpr (COARSE && !BIG) {
            int zl = lane;
            asm volatile("" : "+v"(zl));   // per iteration: hoisted out of the loop these twenty registers would be spilled
#pragma unroll
            for (int u = 0; u < CZR_U; ++u) ce[u] = lz[max(min(zp0 + zl + 64 * u, zp1 - 1), 0)];
        }
        __syncthreads();
        double rz2 = 0; rr = 0;
#pragma unroll
        for (int w = 0; w < CGR_W; ++w) { rz2 += shi[CGR_W + w]; rr += shi[2 * CGR_W + w]; }
        if constexpr (COARSE) {
            double w6[6] = {0, 0, 0, 0, 0, 0};
            if constexpr (BIG) {
                // aggregates of up to 600 nodes: a loop, r from the batch vector (this compute unit wrote it: its caches hold it)
                for (int q = zp0 + lane; q < zp1; q += 64) {
                    const CzNode n = cz_node(lz[q]);
                    const double *rs = ag + 3 * n.node;
                    const double g0 = rs[0], g1 = rs[1], g2 = rs[2];
                    const double r0 = n.m0 ? 0.0 : g0, r1 = n.m1 ? 0.0 : g1, r2 = n.m2 ? 0.0 : g2;
                    CZ_RESTRICT_ADD(w6, n, r0, r1, r2);
                }
            } else {
#pragma unroll
                for (int u = 0; u < CZR_U; ++u) {
                    const CzNode n = cz_node(ce[u]);
                    const double *rs = Ap_s + 3 * n.node;
                    const bool in = zp0 + lane + 64 * u < zp1;
                    const double r0 = n.m0 || !in ? 0.0 : rs[0], r1 = n.m1 || !in ? 0.0 : rs[1], r2 = n.m2 || !in ? 0.0 : rs[2];
                    CZ_RESTRICT_ADD(w6, n, r0, r1, r2);
                }
            }
            double share = 0;
#pragma unroll
            for (int m = 0; m < 6; ++m) { w6[m] = wave_sum_f64(w6[m]); share += (BIG ? acl[m * CZ_NC] : ac6[m]) * w6[m]; }
            if (lane < CZ_NC) cz_share[wave * CZ_NC + lane] = share;
#pragma unroll
            for (int m = 0; m < 6; ++m) if (lane == m) cz_w[6 * wave + m] = w6[m];
            __syncthreads();                            // every aggregate's w and share of v
            double v = 0;
#pragma unroll
            for (int a = 0; a < CZ_NA; ++a) v += cz_share[a * CZ_NC + min(lane, CZ_NC - 1)];
            rz2 += wave_sum_f64(lane < CZ_NC ? cz_w[lane] * v : 0.0);
            double va[6];
#pragma unroll
            for (int m = 0; m < 6; ++m) va[m] = readlane_f64(v, 6 * wave + m);
            auto prolong = [&](const CzNode n) {
                double *cs = Ap_s + 3 * n.node;
                cs[0] = n.m0 ? 0.0 : va[0] + (va[4] * n.q2 - va[5] * n.q1);
                cs[1] = n.m1 ? 0.0 : va[1] + (va[5] * n.q0 - va[3] * n.q2);
                cs[2] = n.m2 ? 0.0 : va[2] + (va[3] * n.q1 - va[4] * n.q0);
            };
            if constexpr (BIG) {
                // no second vector in LDS to pass Z v through, and 28 more loads per thread from the batch vector cost 20 us (the
                // allocator spills around them): the owners form r/diag + beta p first, then each wave adds Z v for its aggregate's
                // nodes into p in LDS -- (r/diag + beta p) + Z v, the other forms' (r/diag + Z v) + beta p to within rounding
                const double beta = cg_ratio(rz2, rz);
#pragma unroll
                for (int u = 0; u < CGR_U; ++u) {
                    const int i = u * CGR_T + tv;
                    if (i < nrows) p_s[i] = rv[u] * dd[u] + beta * p_s[i];
                }
                __syncthreads();
                for (int q = zp0 + lane; q < zp1; q += 64) {
                    const CzNode n = cz_node(lz[q]);
                    double *ps = p_s + 3 * n.node;
                    if (!n.m0) ps[0] += va[0] + (va[4] * n.q2 - va[5] * n.q1);
                    if (!n.m1) ps[1] += va[1] + (va[5] * n.q0 - va[3] * n.q2);
                    if (!n.m2) ps[2] += va[2] + (va[3] * n.q1 - va[4] * n.q0);
                }
            } else {
                {   // the same entries again (cache-resident now; kept in registers across the barrier they would be spilled)
                    int zl = lane;
                    asm volatile("" : "+v"(zl));
#pragma unroll
                    for (int u = 0; u < CZR_U; ++u) ce[u] = lz[max(min(zp0 + zl + 64 * u, zp1 - 1), 0)];
                }
#pragma unroll
                for (int u = 0; u < CZR_U; ++u)
                    if (zp0 + lane + 64 * u < zp1) prolong(cz_node(ce[u]));
            }
            if constexpr (!BIG) __syncthreads();        // Z v complete
        }
        const double beta = cg_ratio(rz2, rz);
        rz = rz2;
        if constexpr (BIG && COARSE) {
            // (done inside the coarse section)
        } else if constexpr (BIG) {
#pragma unroll
            for (int u = 0; u < CGR_U; ++u) {
                const int i = u * CGR_T + tv;
                if (i < nrows) p_s[i] = rv[u] * dd[u] + beta * p_s[i];
            }
        } else {
#pragma unroll
            for (int u = 0; u < CGR_U; ++u) {
                const int i = u * CGR_T + tid;
                if (i < nrows) p_s[i] = COARSE ? (rv[u] * dv[u] + Ap_s[i]) + beta * p_s[i] : rv[u] * dv[u] + beta * p_s[i];   // r/diag: the same product as in the sum above
            }
        }
        __syncthreads();                                // the new p is complete before anyone gathers from it
    }
#pragma unroll
    for (int u = 0; u < CGR_U; ++u) {
        const int i = u * CGR_T + tid;
        if (i < nrows) { if (!BIG) x[row0 + i] = xv[u]; r[row0 + i] = rv[u]; p[row0 + i] = p_s[i]; }
    }
    if (tid == 0) { sc[mesh].rz[0] = rz; sc[mesh].rz[1] = rz; sc[mesh].rr = rr; }
}

// FEA2 is a stack object per PoseOptimizationNR call (Optimizer.cc:480): a model is created and destroyed every
// frame with nearly the same sizes.  Device blocks, pinned blocks and streams are therefore recycled through
// small process-wide caches (size classes = powers of two, blocks above 64 MiB are not kept), so a steady-state
// fem_create / fem_destroy pair performs no hipMalloc / hipFree / hipStreamCreate.
struct BlockCache {
    std::mutex mu;
    std::unordered_map<size_t, std::vector<void *>> free_;
    std::unordered_map<void *, size_t> cls_of;
    bool pinned;
    explicit BlockCache(bool pin) : pinned(pin) {}
    static size_t cls(size_t bytes) { size_t c = 256; while (c < bytes) c <<= 1; return c; }
    void *get(size_t bytes)
    {
        const size_t c = cls(bytes ? bytes : 1);
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = free_.find(c);
            if (it != free_.end() && !it->second.empty()) { void *p = it->second.back(); it->second.pop_back(); return p; }
        }
        void *p = nullptr;
        const hipError_t e = pinned ? hipHostMalloc(&p, c, hipHostMallocDefault) : hipMalloc(&p, c);
        if (e != hipSuccess) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        cls_of[p] = c;
        return p;
    }
    void put(void *p)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        const size_t c = cls_of[p];
        if (c > ((size_t)64 << 20)) { cls_of.erase(p); if (pinned) (void)hipHostFree(p); else (void)hipFree(p); return; }
        free_[c].push_back(p);
    }
};
BlockCache g_dev_cache(false), g_pin_cache(true);
std::mutex g_stream_mu;
std::vector<hipStream_t> g_stream_free;

hipStream_t stream_get()
{
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        if (!g_stream_free.empty()) { hipStream_t s = g_stream_free.back(); g_stream_free.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    // (a high-priority queue does not get k_fem_cg_xcd's workgroups seated sooner beside a saturating load on other streams: tried, 79 against 78
    // launches of ~730 gave up either way, profiles/r05_notes.md)
    return hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess ? s : nullptr;
}
void stream_put(hipStream_t s)
{
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_free.push_back(s);
}

template <typename T> int dalloc(T **p, size_t n) { *p = static_cast<T *>(g_dev_cache.get((n ? n : 1) * sizeof(T))); return *p ? 0 : -1; }
inline void dfree(void *p) { g_dev_cache.put(p); }

} // namespace

struct fem_model {
    int eltype, npe, nd, nmesh, nn, ne, ndof, nblk, nchunk, nchunk_s, spmv_lds, spb;
    size_t nnz, nnzs; // non-zeros per mesh; per-mesh stride of vals/cols (multiple of 4: 16-B aligned streams)
    unsigned int E;
    float nu, fg, lambda, G;
    FemConst fc;
    std::vector<int> h_rowptr, h_lcol, h_diag, h_bp;
    // segmented layout (fem_create_batch): nseg meshes of their own sizes concatenated; nmesh == 1 then and nn / ne / ndof /
    // nnz are the totals.  Uniform layout: nseg == nmesh, no tables.
    int nseg = 0, nchunk_tot = 0, nchunk_s_tot = 0;
    std::vector<int> seg_node0, seg_elem0, seg_nnz0; // [nseg + 1]
    int *d_nel_ptr = nullptr, *d_nel = nullptr, *d_contrib_loc = nullptr; // fused assembly: elements per node, contributions by local element
    int fused_lds = 0;                                                      // LDS bytes of k_fem_assemble_fused (0: two-kernel assembly)
    int rows_lds = 0;                                                       // LDS bytes of k_fem_assemble_rows (0: D not isotropic / does not fit)
    float glimit = 0.0f;                                                    // gradients below this keep g D finite (k_fem_assemble_rows)
    float *d_ke1 = nullptr;                                                 // one K_e for the accessor
    int *d_cmesh = nullptr, *d_cmesh_s = nullptr;
    int4 *d_minfo = nullptr, *d_minfo_s = nullptr;
    bool segmented() const { return d_cmesh != nullptr; }
    bool assembled = false, cg_ready = false, trial_ready = false;
    int tr_npoints = 0, tr_nder = 0, tr_nids = 0, tr_seq = 0;
    float tr_klarge = 0.f;
    double *d_tr_points = nullptr;
    char *h_tr_pin = nullptr; // pinned staging of the LM hook: points in, a / sE / nsE out (pageable copies above a few KB pin on the fly)
    size_t h_tr_pin_bytes = 0;
    float *d_tr_top = nullptr, *d_tr_u0 = nullptr;
    int *d_tr_derived = nullptr, *d_tr_ids = nullptr;
    unsigned *d_tr_done = nullptr;   // per mesh: workgroups of k_fem_matvec_energy that have finished
    int cg_it = 0;
    // device
    char *d_tables = nullptr;   // ONE block: node coordinates and every index table below (interior pointers; create_model)
    float *d_nodes = nullptr, *d_ke = nullptr, *d_vals = nullptr, *d_a = nullptr, *d_f = nullptr, *d_u = nullptr, *d_e = nullptr;
    int *d_elems = nullptr, *d_blk_row = nullptr, *d_bptr = nullptr, *d_cptr = nullptr, *d_contrib = nullptr;
    int *d_rowptr = nullptr, *d_lcol = nullptr, *d_diag = nullptr, *d_bcol3 = nullptr, *d_bp = nullptr;
    float *d_vals_b = nullptr;   // block-major copy of d_vals for the CG (fem_cg_setup)
    // k_fem_cg_resident: chunk table {first block row, block rows, first block, blocks} and each mesh's chunk range
    int4 *d_rcd = nullptr; int *d_rcfirst = nullptr;
    bool cg_resident = false, cgr_big = false; int cgr_lds = 0, cgr_ldn = 0;
    // k_fem_cg_xcd (one mesh): participants, LDS doubles per vector / blocks per chunk region, LDS bytes, the per-workgroup plan, the
    // barrier block and how far its counter has been driven
    bool cg_xcd = false; int xg_P = 0, xg_ldr = 0, xg_ldq = 0, xg_lds = 0, xg_mc = 0; int4 *d_xg_plan = nullptr; XgCtl *d_xg_ctl = nullptr; unsigned xg_bar = 0;
    char *d_xg_gran = nullptr;   // its tagged 16-byte granules (xg_layout)
    hipEvent_t xg_done = nullptr; bool xg_inflight = false;   // admission: see xg_admit
    std::vector<std::pair<unsigned, int>> xg_pending;        // k_fem_cg_xcd launches since the last check of the abort word: {first tag, iterations}
    int xg_cooldown = 0, xg_backoff = 0;                      // calls that stay on the launch-per-phase path after a recovery (doubling while recoveries follow each other)
    long long xg_launches = 0, xg_recovered = 0;             // (fem_cg_one_launch_stats)
    double *d_b = nullptr, *d_x = nullptr, *d_r = nullptr, *d_p = nullptr, *d_Ap = nullptr, *d_dinv = nullptr;
    double *d_part[4] = {nullptr, nullptr, nullptr, nullptr};
    CgScal *d_sc = nullptr;
    // two-level preconditioner (fem_cg_preconditioner): constrained dofs as the Dirichlet calls recorded them, the coarse space
    // (k_fem_cz_*), the inverse coarse matrices and the coarse vectors w, v (48 per mesh) and w.v
    int precond = 0;
    std::vector<uint8_t> h_cmask;
    float4 *d_cz = nullptr, *d_cznode = nullptr; int *d_czptr = nullptr, *d_czmax = nullptr; uint8_t *d_cmask = nullptr;
    double *d_cy = nullptr;   // K Z_b, six doubles per row (k_fem_cz_kz)
    int kz_lds = 0;           // its LDS bytes: 18 doubles per block of the fullest KZ_SPB rows
    double *d_ac = nullptr, *d_aci = nullptr, *d_cw = nullptr, *d_cv = nullptr, *d_cwv = nullptr;
    bool cz_space_valid = false;   // the coarse space depends on the nodes (fixed) and on the constrained dofs
    bool coarse() const { return precond == FEM_PRECOND_TWO_LEVEL; }
    void name_kernel_kinds()
    {
        static const char *names[5] = {"k_fem_ke", "k_fem_assemble", "k_fem_spmv", "k_fem_cg_update", "k_fem_cg_dir"};
        for (int i = 0; i < 5; ++i) prof.names[i] = names[i];
        if (fused_step()) { prof.names[3] = "k_fem_cg_step"; prof.names[4] = nullptr; }   // one launch does both
        prof.names[5] = cg_resident ? "k_fem_cg_resident" : (cg_xcd ? "k_fem_cg_xcd" : nullptr);
        prof.names[6] = coarse() ? "k_fem_cz_*" : nullptr;
    }
    // the vector half of an iteration as ONE per-mesh workgroup (k_fem_cg_step): batches of 16 meshes and more (a single mesh under
    // the two-level preconditioner was tried there too, to save launches: one compute unit's bandwidth, 12.8 -> 14.1 ms to 1e-8)
    bool fused_step() const { return nseg >= 16; }
    // the whole solve in one launch per call; the two-level form needs r in LDS, which the one-vector (BIG) layout has no room for:
    // those meshes go phase by phase
    // (nor for an aggregate of more than CZR_U x 64 nodes: cz_max_agg, known after fem_cg_setup)
    int cz_max_agg = 0;
    bool resident_now() const { return cg_resident && !(coarse() && !cgr_big && cz_max_agg > 64 * 5); }
    // the one-XCD kernel: one mesh under point Jacobi (FEM_CG_XCD=0 keeps the launch-per-phase path, which it equals bit for bit)
    bool xcd_now() const   // (the switch is read per call: the tests flip it)
    {
        const char *e = getenv("FEM_CG_XCD");
        if (!cg_xcd || (e && e[0] == '0')) return false;
        // the two-level form also keeps an aggregate's r rows, the coarse vectors and a per-row table in LDS: it must fit beside the rest
        return !coarse() || (size_t)xg_lds + (size_t)(3 * cz_max_agg + 200 + 48 * 48) * 8 + (size_t)xg_ldr * 12 <= (size_t)(xg_P > 32 ? 78 : 150) * 1024;
    }
    hipStream_t stream = nullptr;
    hipStream_t cg_stream = nullptr; // the stream the last fem_cg_iterate ran on
    orbx::KernelProfiler prof;
};

namespace {

// Admission of k_fem_cg_xcd launches.  A launch parks up to 32 workgroups that wait for EACH OTHER, and such a workgroup fills its compute
// unit (512 registers per lane): the chip holds 256 of them.  More than XG_MAX_INFLIGHT launches at once (models solved from many host
// threads, or queued on many streams) could leave every launch with some of its workgroups resident and none complete -- they would spin
// until their timeouts.  So the library counts the launches it has in flight (an event behind each; settled when the model's next call
// finds it complete, at fem_cg_result, or at destruction) and sends a call beyond the limit down the launch-per-phase path instead, which
// gives the same bits.
constexpr int XG_MAX_INFLIGHT = 6;
std::atomic<int> g_xg_inflight{0};
void xg_settle(fem_model *m, bool wait)
{
    if (!m->xg_inflight) return;
    if (wait) (void)hipEventSynchronize(m->xg_done);
    else if (hipEventQuery(m->xg_done) != hipSuccess) return;
    m->xg_inflight = false;
    g_xg_inflight.fetch_sub(1);
}
bool xg_admit(fem_model *m)
{
    xg_settle(m, false);
    if (m->xg_inflight) return true;                    // this model's own earlier launch, same stream order: no new seat needed
    if (g_xg_inflight.fetch_add(1) >= XG_MAX_INFLIGHT) { g_xg_inflight.fetch_sub(1); return false; }
    if (!m->xg_done && hipEventCreateWithFlags(&m->xg_done, hipEventDisableTiming) != hipSuccess) { m->xg_done = nullptr; g_xg_inflight.fetch_sub(1); return false; }
    m->xg_inflight = true;
    return true;
}

void fem_free(fem_model *m)
{
    void *ptrs[] = {m->d_tables, m->d_ke, m->d_vals, m->d_a, m->d_f, m->d_u, m->d_e, m->d_b, m->d_x, m->d_r,
                    m->d_p, m->d_Ap, m->d_dinv, m->d_part[0], m->d_part[1], m->d_part[2], m->d_part[3], m->d_sc, m->d_tr_points, m->d_tr_top, m->d_tr_u0,
                    m->d_tr_derived, m->d_tr_ids, m->d_tr_done, m->d_ke1, m->d_vals_b, m->d_cz, m->d_cznode, m->d_cy, m->d_czptr, m->d_ac, m->d_aci, m->d_cw, m->d_cv, m->d_cwv, m->d_czmax, m->d_cmask, m->d_xg_ctl, m->d_xg_gran};
    xg_settle(m, true);
    if (m->xg_done) (void)hipEventDestroy(m->xg_done);
    if (m->stream) (void)hipStreamSynchronize(m->stream); // blocks go back to the cache: nothing may still use them
    for (void *q : ptrs)
        if (q) dfree(q);
    g_pin_cache.put(m->h_tr_pin);
    stream_put(m->stream);
}

int ensure_vecs(fem_model *m)
{
    const size_t N = (size_t)m->nmesh * m->ndof;
    if (!m->d_a && (dalloc(&m->d_a, N) || dalloc(&m->d_f, N) || dalloc(&m->d_u, N) || dalloc(&m->d_e, 2 * (size_t)m->nseg)))
        return -1;
    return 0;
}

int ensure_cg(fem_model *m)
{
    const size_t N = (size_t)m->nmesh * m->ndof, C = (size_t)std::max(m->nchunk_tot, m->nchunk_s_tot);
    if (m->d_b) return 0;
    if (dalloc(&m->d_b, N) || dalloc(&m->d_x, N) || dalloc(&m->d_r, N) || dalloc(&m->d_p, N) || dalloc(&m->d_Ap, N) ||
        dalloc(&m->d_dinv, N) || dalloc(&m->d_part[0], C) || dalloc(&m->d_part[1], C) || dalloc(&m->d_part[2], C) ||
        dalloc(&m->d_part[3], C) || dalloc(&m->d_sc, (size_t)m->nseg) || dalloc(&m->d_vals_b, (size_t)m->nmesh * m->nnzs) ||
        (m->cg_xcd && (dalloc(&m->d_xg_ctl, (size_t)1) || dalloc(&m->d_xg_gran, (size_t)16 * xg_layout(m->ndof, m->nchunk, m->nchunk_s).total))))
        return -1;
    return 0;
}

// grids: uniform (chunks of one mesh, meshes) / segmented (all chunks, 1)
inline dim3 grid_cg(const fem_model *m) { return m->segmented() ? dim3(m->nchunk_tot) : dim3(m->nchunk, m->nmesh); }
inline dim3 grid_spmv(const fem_model *m) { return m->segmented() ? dim3(m->nchunk_s_tot) : dim3(m->nchunk_s, m->nmesh); }

void launch_spmv(fem_model *m, hipStream_t st)
{
    const bool pap = !m->fused_step();   // the per-mesh k_fem_cg_step forms p.Ap itself
    hipLaunchKernelGGL(m->spb == 48 ? (pap ? k_fem_spmv<48, true> : k_fem_spmv<48, false>) : (pap ? k_fem_spmv<96, true> : k_fem_spmv<96, false>),
                       grid_spmv(m), dim3(CGT), m->spmv_lds, st,
                       m->d_vals_b, m->d_bcol3, m->d_bp, m->nnzs, m->ndof, m->nchunk_s, m->d_p, m->d_Ap, m->d_part[0],
                       (const int *)m->d_cmesh_s, (const int4 *)m->d_minfo_s);
}

// Coarse space (k_fem_cz_build, again only after the constrained dofs changed), Ac = Z^T K Z column by column (Z e_k through the
// product kernel of the CG, restricted again: 48 products, once per fem_cg_setup) and its inverse (k_fem_cz_invert): all on the
// model's stream, nothing comes back to the host but the size of the largest aggregate.  Needs the block-major values
// (k_fem_to_blocks) in place.
int setup_coarse(fem_model *m)
{
    const size_t NN = (size_t)m->nmesh * m->nn, NC2 = (size_t)CZ_NC * CZ_NC;
    if (!m->d_cz && (dalloc(&m->d_cz, NN) || dalloc(&m->d_cznode, NN) || dalloc(&m->d_cy, 6 * (size_t)m->nmesh * m->ndof) ||
                     dalloc(&m->d_czptr, 9 * (size_t)m->nseg) || dalloc(&m->d_ac, NC2 * m->nseg) ||
                     dalloc(&m->d_aci, NC2 * m->nseg) || dalloc(&m->d_cw, (size_t)CZ_NC * m->nseg) || dalloc(&m->d_cv, (size_t)CZ_NC * m->nseg) ||
                     dalloc(&m->d_cwv, (size_t)m->nseg) || dalloc(&m->d_czmax, 1) || dalloc(&m->d_cmask, (size_t)m->ndof)))
        return -1;
    if (!m->cz_space_valid) {
        if (m->h_cmask.empty()) { if (hipMemsetAsync(m->d_cmask, 0, (size_t)m->ndof, m->stream) != hipSuccess) return -1; }
        else if (hipMemcpyAsync(m->d_cmask, m->h_cmask.data(), m->h_cmask.size(), hipMemcpyHostToDevice, m->stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_fem_mask_no_diagonal, dim3((m->ndof + 255) / 256), dim3(256), 0, m->stream, (const float *)m->d_vals, (const int *)m->d_diag, m->ndof, m->d_cmask);
        if (hipMemsetAsync(m->d_czmax, 0, sizeof(int), m->stream) != hipSuccess) return -1;
        hipLaunchKernelGGL(k_fem_cz_build, dim3(m->nseg), dim3(CZ_T), 0, m->stream, (const float *)m->d_nodes,
                           (const uint8_t *)m->d_cmask, m->segmented() ? 1 : 0, m->d_cz, m->d_cznode,
                           m->d_czptr, m->d_czmax, m->ndof, (const int4 *)m->d_minfo);
    }
    const dim3 g(CZ_NA, m->nseg);
    for (int b = 0; b < CZ_NA; ++b) {
        hipLaunchKernelGGL(k_fem_cz_kz, dim3((m->ndof + KZ_SPB - 1) / KZ_SPB, m->nmesh), dim3(CGT), m->kz_lds, m->stream, (const float *)m->d_vals_b,
                           (const int *)m->d_bcol3, (const int *)m->d_bp, m->nnzs, m->ndof, (const float4 *)m->d_cznode, b, m->d_cy);
        hipLaunchKernelGGL(k_fem_cz_restrict6, g, dim3(CZ_T), 0, m->stream, (const float4 *)m->d_cz, (const int *)m->d_czptr, (const double *)m->d_cy, b,
                           m->d_ac, m->ndof, (const int4 *)m->d_minfo);
    }
    hipLaunchKernelGGL(k_fem_cz_invert, dim3(m->nseg), dim3(64), 0, m->stream, (const double *)m->d_ac, m->d_aci);
    if (!m->cz_space_valid) {
        if (hipMemcpyAsync(&m->cz_max_agg, m->d_czmax, sizeof(int), hipMemcpyDeviceToHost, m->stream) != hipSuccess ||
            hipStreamSynchronize(m->stream) != hipSuccess)
            return -1;
        m->cz_space_valid = true;
    }
    return 0;
}

// w = Z^T src, v = Ac^-1 w, w.v, and out (=, +=) Z v: the coarse half of the two-level preconditioner
void coarse_correction(fem_model *m, hipStream_t st, const double *src, double *out, int accumulate)
{
    const dim3 g(CZ_NA, m->nseg);
    m->prof.start(6, st);
    if ((size_t)m->nmesh * m->nn <= (size_t)m->nseg * 65536) {   // one workgroup per mesh does all three steps
        hipLaunchKernelGGL(k_fem_cz_apply, dim3(m->nseg), dim3(1024), 0, st, (const float4 *)m->d_cz, (const int *)m->d_czptr, (const double *)m->d_aci,
                           src, out, accumulate, m->d_cwv, m->ndof, (const int4 *)m->d_minfo);
        m->prof.stop(6, st);
        return;
    }
    hipLaunchKernelGGL(k_fem_cz_restrict, g, dim3(CZ_T), 0, st, (const float4 *)m->d_cz, (const int *)m->d_czptr, src, m->d_cw, m->ndof,
                       (const int4 *)m->d_minfo);
    hipLaunchKernelGGL(k_fem_cz_solve, dim3(m->nseg), dim3(64), 0, st, (const double *)m->d_aci, (const double *)m->d_cw, m->d_cv, m->d_cwv);
    hipLaunchKernelGGL(k_fem_cz_prolong, g, dim3(CZ_T), 0, st, (const float4 *)m->d_cz, (const int *)m->d_czptr, (const double *)m->d_cv, out,
                       accumulate, m->ndof, (const int4 *)m->d_minfo);
    m->prof.stop(6, st);
}

// the variant of k_fem_cg_xcd a model runs: rows per SpMV chunk x chunks per workgroup
inline const void *xg_kernel(int spb, int mc, bool coarse)
{
#define XG_K(S, M) (coarse ? reinterpret_cast<const void *>(k_fem_cg_xcd<S, M, true>) : reinterpret_cast<const void *>(k_fem_cg_xcd<S, M, false>))
    (void)spb;   // 48 for every mesh the kernel takes (plan_model: a single mesh of <= 8,192 dofs is far below the 96-row threshold)
    return mc == 1 ? XG_K(48, 1) : mc == 3 ? XG_K(48, 3) : XG_K(48, 6);
#undef XG_K
}
// LDS bytes of the variant: the Jacobi layout, and behind it all of r, the coarse vectors and the prolongation's per-row table
inline int xg_lda(const fem_model *m) { return (3 * std::max(m->cz_max_agg, 1) + 1) & ~1; }   // LDS doubles of an aggregate's r rows
inline size_t xg_lds_bytes(const fem_model *m, bool coarse)
{
    size_t b = (size_t)m->xg_lds;
    if (coarse) b += (size_t)(xg_lda(m) + 3 * CZ_NC + 2 + CZ_NC * CZ_NC) * sizeof(double) + (size_t)m->xg_ldr * (sizeof(float2) + sizeof(int)) + 32;
    return b;
}

void launch_iter(fem_model *m, hipStream_t st)
{
    const dim3 g = grid_cg(m);
    const int cur = m->cg_it & 1;
    m->prof.start(2, st);
    launch_spmv(m, st);
    m->prof.stop(2, st);
    if (m->fused_step()) {
        m->prof.start(3, st);
        hipLaunchKernelGGL(m->coarse() ? k_fem_cg_step<true> : k_fem_cg_step<false>, dim3(m->nseg), dim3(CGS_T), 0, st, m->ndof, cur, m->d_sc,
                           m->d_p, m->d_Ap, m->d_dinv, m->d_x, m->d_r, (const int4 *)m->d_minfo, (const float4 *)m->d_cz, (const int *)m->d_czptr,
                           (const double *)m->d_aci);
        m->prof.stop(3, st);
        m->cg_it++;
        return;
    }
    m->prof.start(3, st);
    hipLaunchKernelGGL(k_fem_cg_update, g, dim3(CGT), 0, st, m->ndof, m->nchunk, m->nchunk_s, cur, m->d_sc, m->d_part[0], m->d_p,
                       m->d_Ap, m->d_dinv, m->d_x, m->d_r, m->d_part[1], m->d_part[2], (const int *)m->d_cmesh,
                       (const int4 *)m->d_minfo, (const int4 *)m->d_minfo_s);
    m->prof.stop(3, st);
    if (m->coarse()) coarse_correction(m, st, m->d_r, m->d_Ap, 0);   // Ap is free between the update and the next product
    m->prof.start(4, st);
    hipLaunchKernelGGL(k_fem_cg_dir, g, dim3(CGT), 0, st, m->ndof, m->nchunk, cur, m->d_sc, m->d_part[1], m->d_part[2],
                       m->d_r, m->d_dinv, m->d_p, (const int *)m->d_cmesh, (const int4 *)m->d_minfo,
                       m->coarse() ? (const double *)m->d_Ap : nullptr, (const double *)m->d_cwv);
    m->prof.stop(4, st);
    m->cg_it++;
}

// n iterations of the batch: one launch where a mesh fits a compute unit (k_fem_cg_resident), else launch by launch
void run_iters(fem_model *m, int n, hipStream_t st)
{
    if (m->resident_now()) {
        if (n <= 0) return;
        m->prof.start(5, st);
        const auto kern = m->cgr_big ? (m->coarse() ? k_fem_cg_resident<true, true> : k_fem_cg_resident<true, false>)
                                     : (m->coarse() ? k_fem_cg_resident<false, true> : k_fem_cg_resident<false, false>);
        hipLaunchKernelGGL(kern, dim3(m->nseg), dim3(CGR_T), m->cgr_lds, st,
                           m->d_vals_b, m->d_bcol3, m->d_bp, (const int4 *)m->d_rcd, (const int *)m->d_rcfirst, m->nnzs, m->ndof,
                           m->cgr_ldn, n, m->d_sc, m->d_p, m->d_dinv, m->d_x, m->d_r, m->d_Ap, (const int4 *)m->d_minfo,
                           (const float4 *)m->d_cz, (const int *)m->d_czptr, (const double *)m->d_aci);
        m->prof.stop(5, st);
        m->cg_it += 2 * ((n + 1) / 2);   // both rz slots are current after the launch: keep the parity of the other path even
        return;
    }
    if (m->xg_cooldown > 0 && n > 0) m->xg_cooldown--;
    else if (m->xcd_now() && n > 0 && xg_admit(m)) {
        // (the kernel starts from the even rz slot: should it give up, the state and the host's parity still agree -- see xg_recover)
        if (m->cg_it & 1) { launch_iter(m, st); if (--n == 0) return; }
        m->xg_pending.emplace_back(m->xg_bar, n); m->xg_launches++;
        m->prof.start(5, st);
        {
            // bit 0: which rz slot is current; bit 1: FEM_CG_XCD=safe -- system-scope granules whatever the placement (the mode the tests
            // cannot otherwise reach: the participants have shared an XCD in every run so far)
            const char *xe = getenv("FEM_CG_XCD");
            int cur = (m->cg_it & 1) | (xe && xe[0] == 's' ? 2 : 0);
            void *gran = m->d_xg_gran;
            const int4 *plan = m->d_xg_plan;
            const bool co = m->coarse();
            const float4 *czl = co ? m->d_cz : nullptr, *czn = co ? m->d_cznode : nullptr;
            const int *czp = co ? m->d_czptr : nullptr;
            const double *aci = co ? m->d_aci : nullptr;
            const size_t lds = xg_lds_bytes(m, co);
            int lda = co ? xg_lda(m) : 0;
            void *args[] = {&m->d_vals_b, &m->d_bcol3, &m->d_bp, &m->ndof, &m->nchunk, &m->nchunk_s, &n, &cur, &m->d_sc, &m->d_p, &m->d_dinv, &m->d_x,
                            &m->d_r, &gran, &plan, &m->xg_P, &m->xg_ldr, &m->xg_ldq, &m->d_xg_ctl, &m->xg_bar, &czl, &czp, &czn, &aci, &lda};
            const void *fn = xg_kernel(m->spb, m->xg_mc, co);
            {   // the limit belongs to the FUNCTION, not to the model: raised once per variant to what the hardware has (a per-model value
                // would be lowered by the next smaller model, and the larger one's next launch would fail)
                static std::atomic<unsigned> raised{0};
                const unsigned bit = 1u << ((co ? 3 : 0) + (m->xg_mc == 1 ? 0 : m->xg_mc == 3 ? 1 : 2));
                if (!(raised.load() & bit)) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); raised.fetch_or(bit); }   // (the kernel's static arrays take a few KB of the 160)
            }
            (void)hipLaunchKernel(fn, dim3(XG_STRIDE * XG_MAXP), dim3(CGT), args, lds, st);
        }
        m->prof.stop(5, st);
        (void)hipEventRecord(m->xg_done, st);
        m->xg_bar += 3u * (unsigned)n;                        // the granules' tags: three per iteration, never reused (fem_cg_setup clears the buffer)
        m->cg_it = 2 * ((m->cg_it + n + 1) / 2);              // both rz slots are current after the launch
        return;
    }
    for (int i = 0; i < n; ++i) launch_iter(m, st);
}

// Behind a synchronisation of `st` that brought the two control words along: a one-launch call that gave up (a participant was not
// scheduled in time) has changed nothing -- x, r, p and the scalars are written after the last iteration only, every later launch left
// at its first instruction, and launch-per-phase calls in between ran from the state as it was, with the right rz slot (the kernel
// always starts from the even one) -- so the iterations of the launches that did NOT run to their end (they say so: pad[0]) are simply
// run now, launch by launch: CG does not care how its iterations are cut into calls (same bits).  The model then stays on the
// launch-per-phase path for a while.
int xg_recover(fem_model *m, hipStream_t st, const unsigned (&ctl2)[2])   // {abort word, the end tag of the last launch that ran to its end}
{
    int redo = 0;
    for (const auto &l : m->xg_pending)
        if ((int)(l.first + 3u * (unsigned)l.second - ctl2[1]) > 0) redo += l.second;   // (tags only grow between two fem_cg_setup calls)
    const bool had = !m->xg_pending.empty();
    m->xg_pending.clear();
    if (!ctl2[0]) { if (had) m->xg_backoff = 0; return ORBX_OK; }
    ORBX_HIP(hipMemsetAsync(&m->d_xg_ctl->abort_flag, 0, sizeof(unsigned), st));
    m->xg_backoff = std::min(m->xg_backoff + 1, 7); m->xg_cooldown = 16 << m->xg_backoff; m->xg_recovered++;   // 32, 64 .. 2,048 calls on the launch-per-phase path
    for (int i = 0; i < redo; ++i) launch_iter(m, st);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipStreamSynchronize(st));
    return ORBX_OK;
}

// Symbolic phase of one mesh (host, once per topology): block pattern, contribution lists, CSR pattern.
struct Symbolic {
    std::vector<int> bptr, bcol, blk_row, cptr, contrib, rowptr, lcol, diag;
    std::vector<int> nel_ptr, nel, contrib_loc; // elements holding each node (ascending); contrib with the element as its index in the row node's list
    int nblk = 0, maxel = 0;
};
// The first formulation (adjacency lists, binary searches): kept as the cross-check of build_symbolic (fem_plan_selfcheck).
void build_symbolic_simple(int npe, int nn, int ne, const int32_t *elems, Symbolic &y)
{
    std::vector<std::vector<int>> adj(nn);
    for (int i = 0; i < nn; ++i) adj[i].push_back(i);
    for (int e = 0; e < ne; ++e)
        for (int a = 0; a < npe; ++a)
            for (int b = 0; b < npe; ++b) adj[elems[e * npe + a]].push_back(elems[e * npe + b]);
    y.bptr.assign(nn + 1, 0); y.bcol.clear(); y.blk_row.clear();
    for (int i = 0; i < nn; ++i) {
        std::sort(adj[i].begin(), adj[i].end());
        adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
        y.bptr[i + 1] = y.bptr[i] + (int)adj[i].size();
        for (int j : adj[i]) { y.bcol.push_back(j); y.blk_row.push_back(i); }
    }
    const int nblk = y.nblk = (int)y.bcol.size();
    y.cptr.assign(nblk + 1, 0);
    auto blk_of = [&](int I, int J) {
        return y.bptr[I] + (int)(std::lower_bound(adj[I].begin(), adj[I].end(), J) - adj[I].begin());
    };
    for (int e = 0; e < ne; ++e)
        for (int a = 0; a < npe; ++a)
            for (int b = 0; b < npe; ++b) y.cptr[blk_of(elems[e * npe + a], elems[e * npe + b]) + 1]++;
    for (int b = 0; b < nblk; ++b) y.cptr[b + 1] += y.cptr[b];
    y.contrib.assign(y.cptr[nblk], 0);
    std::vector<int> fill(y.cptr.begin(), y.cptr.end() - 1);
    for (int e = 0; e < ne; ++e) // ascending (e, li, lj) inside every block = the reference's scatter order
        for (int a = 0; a < npe; ++a)
            for (int b = 0; b < npe; ++b) y.contrib[fill[blk_of(elems[e * npe + a], elems[e * npe + b])]++] = (e << 6) | (a << 3) | b;
    // elements per node (ascending, once each) and the contributions re-indexed by them: what the fused assembly caches
    y.nel_ptr.assign(nn + 1, 0); y.nel.clear(); y.maxel = 0;
    {
        std::vector<std::vector<int>> el(nn);
        for (int e = 0; e < ne; ++e)
            for (int a = 0; a < npe; ++a) {
                std::vector<int> &v = el[elems[e * npe + a]];
                if (v.empty() || v.back() != e) v.push_back(e);
            }
        for (int i = 0; i < nn; ++i) {
            y.nel_ptr[i + 1] = y.nel_ptr[i] + (int)el[i].size();
            y.nel.insert(y.nel.end(), el[i].begin(), el[i].end());
            y.maxel = std::max(y.maxel, (int)el[i].size());
        }
        y.contrib_loc.resize(y.contrib.size());
        for (int b = 0; b < nblk; ++b) {
            const std::vector<int> &v = el[y.blk_row[b]];
            for (int c = y.cptr[b]; c < y.cptr[b + 1]; ++c) {
                const int e = y.contrib[c] >> 6;
                const int loc = (int)(std::lower_bound(v.begin(), v.end(), e) - v.begin());
                y.contrib_loc[c] = (loc << 6) | (y.contrib[c] & 63);
            }
        }
    }
    const int ndof = 3 * nn;
    y.rowptr.assign(ndof + 1, 0);
    y.lcol.assign((size_t)9 * nblk, 0);
    y.diag.assign(ndof, 0);
    for (int I = 0; I < nn; ++I) {
        const int nb = y.bptr[I + 1] - y.bptr[I];
        for (int r = 0; r < 3; ++r) {
            const int row = 3 * I + r, start = 9 * y.bptr[I] + r * 3 * nb;
            y.rowptr[row] = start;
            for (int jb = 0; jb < nb; ++jb)
                for (int c = 0; c < 3; ++c) {
                    const int col = 3 * y.bcol[y.bptr[I] + jb] + c;
                    y.lcol[start + 3 * jb + c] = col;
                    if (col == row) y.diag[row] = start + 3 * jb + c;
                }
        }
    }
    y.rowptr[ndof] = 9 * nblk;
}

// Symbolic phase in linear passes over flat arrays (it runs once per fem_create, i.e. once per PoseOptimizationNR call,
// and was 90 % of Compute(1) at the reference's mesh sizes in its first formulation): (1) node -> elements in CSR form by
// counting; (2) per node I: its neighbours gathered from its elements through a stamp array, sorted (a handful of entries),
// their positions parked in a scratch array; the contributions of block row I -- every (e, li, lj) with elems[e][li] == I
// -- are produced right there in ascending (e, li, lj) order = the reference's scatter order, counted per block and laid
// out behind the row's prefix.  No per-node containers, no searches.
void build_symbolic(int npe, int nn, int ne, const int32_t *elems, Symbolic &y)
{
    // (1) elements holding each node, ascending, once each (an element may repeat a node id: degenerate faces)
    y.nel_ptr.assign((size_t)nn + 1, 0);
    for (int e = 0; e < ne; ++e)
        for (int a = 0; a < npe; ++a) {
            const int I = elems[e * npe + a];
            bool seen = false;
            for (int a2 = 0; a2 < a; ++a2) seen |= elems[e * npe + a2] == I;
            if (!seen) y.nel_ptr[I + 1]++;
        }
    y.maxel = 0;
    for (int i = 0; i < nn; ++i) { y.maxel = std::max(y.maxel, y.nel_ptr[i + 1]); y.nel_ptr[i + 1] += y.nel_ptr[i]; }
    y.nel.assign((size_t)y.nel_ptr[nn], 0);
    {
        std::vector<int> fill(y.nel_ptr.begin(), y.nel_ptr.end() - 1);
        for (int e = 0; e < ne; ++e)
            for (int a = 0; a < npe; ++a) {
                const int I = elems[e * npe + a];
                bool seen = false;
                for (int a2 = 0; a2 < a; ++a2) seen |= elems[e * npe + a2] == I;
                if (!seen) y.nel[fill[I]++] = e;
            }
    }
    // (2) block rows
    y.bptr.assign((size_t)nn + 1, 0); y.bcol.clear(); y.blk_row.clear(); y.cptr.assign(1, 0); y.contrib.clear(); y.contrib_loc.clear();
    y.bcol.reserve((size_t)nn * 16); y.blk_row.reserve((size_t)nn * 16);
    y.contrib.reserve((size_t)ne * npe * npe); y.contrib_loc.reserve((size_t)ne * npe * npe);
    std::vector<int> stamp((size_t)nn, -1), pos((size_t)nn, 0), nb, cnt, start;
    for (int I = 0; I < nn; ++I) {
        nb.clear();
        nb.push_back(I); stamp[I] = I;
        for (int k = y.nel_ptr[I]; k < y.nel_ptr[I + 1]; ++k) {
            const int32_t *el = elems + (size_t)y.nel[k] * npe;
            for (int b = 0; b < npe; ++b)
                if (stamp[el[b]] != I) { stamp[el[b]] = I; nb.push_back(el[b]); }
        }
        std::sort(nb.begin(), nb.end());
        const int b0 = (int)y.bcol.size(), nbI = (int)nb.size();
        for (int j = 0; j < nbI; ++j) { pos[nb[j]] = j; y.bcol.push_back(nb[j]); y.blk_row.push_back(I); }
        y.bptr[I + 1] = b0 + nbI;
        // contributions of the row: count per block, then fill in (e, li, lj) order
        cnt.assign((size_t)nbI, 0);
        for (int k = y.nel_ptr[I]; k < y.nel_ptr[I + 1]; ++k) {
            const int32_t *el = elems + (size_t)y.nel[k] * npe;
            for (int a = 0; a < npe; ++a)
                if (el[a] == I)
                    for (int b = 0; b < npe; ++b) cnt[pos[el[b]]]++;
        }
        const int c0 = y.cptr.back();
        start.assign((size_t)nbI, 0);
        int run = c0;
        for (int j = 0; j < nbI; ++j) { start[j] = run; run += cnt[j]; y.cptr.push_back(run); }
        y.contrib.resize((size_t)run); y.contrib_loc.resize((size_t)run);
        for (int k = y.nel_ptr[I]; k < y.nel_ptr[I + 1]; ++k) {
            const int e = y.nel[k], loc = k - y.nel_ptr[I];
            const int32_t *el = elems + (size_t)e * npe;
            for (int a = 0; a < npe; ++a)
                if (el[a] == I)
                    for (int b = 0; b < npe; ++b) {
                        const int at = start[pos[el[b]]]++;
                        y.contrib[at] = (e << 6) | (a << 3) | b;
                        y.contrib_loc[at] = (loc << 6) | (a << 3) | b;
                    }
        }
    }
    const int nblk = y.nblk = (int)y.bcol.size();
    const int ndof = 3 * nn;
    y.rowptr.assign((size_t)ndof + 1, 0);
    y.lcol.assign((size_t)9 * nblk, 0);
    y.diag.assign((size_t)ndof, 0);
    for (int I = 0; I < nn; ++I) {
        const int nbI = y.bptr[I + 1] - y.bptr[I];
        for (int r = 0; r < 3; ++r) {
            const int row = 3 * I + r, st = 9 * y.bptr[I] + r * 3 * nbI;
            y.rowptr[row] = st;
            for (int jb = 0; jb < nbI; ++jb)
                for (int c = 0; c < 3; ++c) {
                    const int col = 3 * y.bcol[y.bptr[I] + jb] + c;
                    y.lcol[st + 3 * jb + c] = col;
                    if (col == row) y.diag[row] = st + 3 * jb + c;
                }
        }
    }
    y.rowptr[ndof] = 9 * nblk;
}

// Everything model construction computes on the host before its first device call: chunking, segment tables, the
// node-block tables of the SpMV and the chunk table of the resident CG.  plan_model touches no device state (it also
// serves fem_plan, which the CPU tests and the host sanitizer build drive without a GPU).
struct HostPlan {
    std::vector<int> cmesh, cmesh_s, bp, bcol3, rcfirst;
    std::vector<int4> minfo, minfo_s, rcd, xg;   // xg: k_fem_cg_xcd's plan, {first SpMV chunk, end, first dof, end} per workgroup
    bool resident = false, big = false, xcd = false;
    int xg_P = 0, xg_ldr = 0, xg_ldq = 0, xg_mc = 0; size_t xg_lds = 0;
    size_t resident_lds = 0;
    int maxrows = 0;
};

// seg_nn == nullptr: uniform layout (nmesh meshes of nn nodes sharing `y`); else the segmented layout over nseg meshes
// whose symbolic data `y` already holds concatenated in global numbering (nmesh == 1, nn / ne = totals).  Fills the
// host fields of `m` (sizes, material, chunk counts, h_* tables; y's row tables are moved into m) and `P`.
int plan_model(fem_model *m, int eltype, int npe, int nmesh, int nn, int ne, unsigned int E, float nu, float fg, Symbolic &y, int nseg,
               const int *seg_nn, const int *seg_ne, HostPlan &P)
{
    m->eltype = eltype; m->npe = npe; m->nd = 3 * npe; m->nmesh = nmesh; m->nn = nn; m->ne = ne; m->ndof = 3 * nn;
    m->nseg = seg_nn ? nseg : nmesh;
    m->E = E; m->nu = nu; m->fg = fg;
    // FEA2::FEA2, FEA2.cc:53-72 (float arithmetic, E unsigned)
    m->lambda = (nu * E) / ((1 + nu) * (1 - 2 * nu));
    m->G = E / (2 * (1 + nu));
    for (int i = 0; i < 36; ++i) m->fc.D[i] = 0.0f;
    m->fc.D[0] = m->fc.D[7] = m->fc.D[14] = m->lambda + 2 * m->G;
    m->fc.D[1] = m->fc.D[2] = m->fc.D[6] = m->fc.D[8] = m->fc.D[12] = m->fc.D[13] = m->lambda;
    m->fc.D[21] = m->fc.D[28] = m->fc.D[35] = m->G;
    static const int sg[8][3] = {{-1, -1, -1}, {+1, -1, -1}, {+1, +1, -1}, {-1, +1, -1}, {-1, -1, +1}, {+1, -1, +1}, {+1, +1, +1}, {-1, +1, +1}};
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 3; ++j) m->fc.gs[3 * i + j] = sg[i][j] < 0 ? -fg : +fg;

    const int nblk = m->nblk = y.nblk;
    m->nnz = (size_t)9 * nblk;
    m->nnzs = (m->nnz + 3) & ~(size_t)3;
    m->h_rowptr.swap(y.rowptr); m->h_lcol.swap(y.lcol); m->h_diag.swap(y.diag);
    // chunking: CG vector kernels RPB rows per workgroup, SpMV SPB rows; a chunk never crosses a mesh
    size_t rows_of_blocks = 0;
    if (seg_nn) for (int k = 0; k < nseg; ++k) rows_of_blocks += (size_t)(3 * seg_nn[k] + 95) / 96;
    else rows_of_blocks = (size_t)nmesh * ((m->ndof + 95) / 96);
    m->spb = rows_of_blocks < 512 ? 48 : 96;   // multiples of 3: a workgroup's rows are whole node-block rows (k_fem_spmv)
    const int SPB = m->spb;
    m->nchunk = (m->ndof + RPB - 1) / RPB;
    m->nchunk_s = (m->ndof + SPB - 1) / SPB;
    m->nchunk_tot = m->nchunk * nmesh; m->nchunk_s_tot = m->nchunk_s * nmesh;
    int maxrun = 0;
    auto scan_runs = [&](int row0, int nrows) {
        for (int r0 = row0; r0 < row0 + nrows; r0 += SPB) {
            const int r1 = r0 + SPB < row0 + nrows ? r0 + SPB : row0 + nrows;
            maxrun = std::max(maxrun, m->h_rowptr[r1] - m->h_rowptr[r0]);
        }
    };
    if (seg_nn) {
        m->seg_node0.assign(nseg + 1, 0); m->seg_elem0.assign(nseg + 1, 0); m->seg_nnz0.assign(nseg + 1, 0);
        int c0 = 0, s0 = 0;
        for (int k = 0; k < nseg; ++k) {
            m->seg_node0[k + 1] = m->seg_node0[k] + seg_nn[k];
            m->seg_elem0[k + 1] = m->seg_elem0[k] + seg_ne[k];
            const int row0 = 3 * m->seg_node0[k], nrows = 3 * seg_nn[k];
            m->seg_nnz0[k + 1] = m->h_rowptr[row0 + nrows];
            const int nc = (nrows + RPB - 1) / RPB, ns = (nrows + SPB - 1) / SPB;
            P.minfo.push_back(make_int4(row0, nrows, c0, nc)); P.minfo_s.push_back(make_int4(row0, nrows, s0, ns));
            P.cmesh.insert(P.cmesh.end(), nc, k); P.cmesh_s.insert(P.cmesh_s.end(), ns, k);
            c0 += nc; s0 += ns;
            scan_runs(row0, nrows);
        }
        m->nchunk_tot = c0; m->nchunk_s_tot = s0;
    } else {
        scan_runs(0, m->ndof);
    }
    // k_fem_spmv parks three row sums per 3 x 3 block of a workgroup's run: maxrun / 9 blocks
    if (maxrun / 3 * (int)sizeof(double) > 150 * 1024) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "rows too long for the SpMV staging buffer");
    m->spmv_lds = (maxrun / 3 + 8) * (int)sizeof(double);
    {   // K_e stays on the chip when the Gauss-point data of a node's elements fits LDS (it always does for real meshes)
        const int ngp = eltype == FEM_TET4 ? 1 : 8;
        const size_t lds = (size_t)y.maxel * (ngp * 3 * npe + ngp) * sizeof(float);
        m->fused_lds = lds <= 64 * 1024 ? (int)std::max(lds, (size_t)16) : 0;
        // k_fem_assemble_rows: one more flag per Gauss point and nine floats per contribution of the fullest block row; it
        // leaves out the products with D's structural zeros, so D must have them
        size_t maxcon = 0;
        for (int I = 0; I < m->nn; ++I) maxcon = std::max(maxcon, (size_t)(y.cptr[y.bptr[I + 1]] - y.cptr[y.bptr[I]]));
        const size_t lds_rows = (size_t)y.maxel * (ngp * 3 * npe + 2 * ngp) * sizeof(float) + maxcon * 9 * sizeof(float);
        bool iso = true;
        float dmax = 0.0f;
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                const float d = m->fc.D[6 * r + c];
                if (!((r < 3 && c < 3) || r == c) && d != 0.0f) iso = false;
                dmax = std::max(dmax, fabsf(d));
            }
        m->glimit = !(dmax <= 3.0e38f) ? 0.0f : (dmax > 1.0f ? 8.0e37f / dmax : 8.0e37f);   // NaN / Inf in D: nothing is "safe"
        m->rows_lds = (m->fused_lds && iso && lds_rows <= 64 * 1024) ? (int)std::max(lds_rows, (size_t)16) : 0;
    }
    {   // node-block tables of k_fem_spmv: bp[I] = blocks before block row I, bcol3[q] = first column of block q
        const int nbr = m->ndof / 3;
        P.bp.assign((size_t)nbr + 1, 0); P.bcol3.assign((size_t)m->nnz / 9, 0);
        size_t q = 0;
        for (int I = 0; I < nbr; ++I) {
            const int k0 = m->h_rowptr[3 * I], nb = (m->h_rowptr[3 * I + 1] - k0) / 3;
            P.bp[I] = (int)q;
            for (int j = 0; j < nb && q < P.bcol3.size(); ++j) P.bcol3[q++] = m->h_lcol[k0 + 3 * j];
        }
        P.bp[nbr] = (int)q;
        if (q != P.bcol3.size()) ORBX_FAIL(ORBX_ERR_ARG, "matrix pattern is not made of 3 x 3 node blocks");
        m->h_bp = P.bp;
        int mb = 0;   // k_fem_cz_kz: 18 doubles per block of the fullest KZ_SPB rows (a quarter of the product kernel's bound times six: fits)
        for (int I = 0; I < nbr; I += KZ_SPB / 3) mb = std::max(mb, P.bp[std::min(I + KZ_SPB / 3, nbr)] - P.bp[I]);
        m->kz_lds = (mb + 1) * 18 * (int)sizeof(double);
    }
    {   // k_fem_cg_resident: whole batches of meshes small enough for one compute unit each
        P.rcfirst.assign(1, 0);
        bool ok = m->nseg >= CGR_MIN_MESHES;
        int maxrows = 0;
        if (seg_nn) for (int k = 0; k < nseg; ++k) maxrows = std::max(maxrows, 3 * seg_nn[k]);
        else maxrows = m->ndof;
        size_t lds = ((size_t)2 * maxrows + CGR_W * 3 * CGR_CB + 6 * CGR_W) * sizeof(double);
        const bool big = maxrows > CGR_MAXROWS || lds > 160 * 1024;   // p alone in LDS
        if (big) lds -= (size_t)maxrows * sizeof(double);
        const int maxbr = big ? 21 : 63;   // block rows per chunk: the row pass of the big form is a single one
        auto chunks_of = [&](int I0, int nbr) {
            for (int I = I0; I < I0 + nbr && ok;) {
                const int q0 = m->h_bp[I];
                int J = I;
                while (J < I0 + nbr && m->h_bp[J + 1] - q0 <= CGR_CB && J - I < maxbr) ++J;
                if (J == I) { ok = false; break; }       // a block row longer than a chunk
                P.rcd.push_back(make_int4(I, J - I, q0, m->h_bp[J] - q0));
                I = J;
            }
            P.rcfirst.push_back((int)P.rcd.size());
        };
        if (seg_nn) for (int k = 0; k < nseg && ok; ++k) chunks_of(m->seg_node0[k], seg_nn[k]);
        else if (ok) chunks_of(0, m->ndof / 3);
        ok = ok && maxrows > 0 && maxrows <= (big ? CGR_MAXROWS_BIG : CGR_MAXROWS) && lds <= 160 * 1024;
        P.resident = ok; P.big = big; P.resident_lds = lds; P.maxrows = maxrows;
    }
    if (!seg_nn && nmesh == 1 && m->nchunk <= 32) {   // k_fem_cg_xcd: ONE mesh, its chunks dealt in contiguous runs to the workgroups of one XCD
        // 32 workgroups (one per compute unit), or up to 64 (two per compute unit: the variants of <= 3 chunks per workgroup are
        // compiled for two workgroups per compute unit -- two waves per SIMD hide each other's latencies) when that takes the chunks per
        // workgroup from 4-6 to 2-3.  More than two per compute unit would leave workgroups waiting for a seat that the spinning ones hold.
        // Two on a compute unit do not run alike: the SIMDs issue oldest-first, the workgroup that came second (in dispatch order: rank >=
        // 32) gets the slots the first leaves -- measured 1.25-1.3x slower in every phase, its partner not at all -- and an iteration lasts
        // as long as its slowest workgroup.  So the second-comers get the smaller share (2 chunks against 3, and no vector chunk where an
        // older neighbour can take it), and the runs of the two kinds alternate along the rows so that every column range stays one
        // neighbourhood of the mesh.  (Placement is the dispatcher's habit, not a promise: a different one costs speed, never results.)
        const int N = m->nchunk_s;
        struct Run { int c0, c1, rank; bool young; };
        std::vector<Run> runs;
        int kch = (N + 31) / 32;
        if (kch > 3 && (N + 63) / 64 <= 3) {
            const int so = N > 128 ? 3 : 2, sy = N > 160 ? 3 : 2;                  // chunks of a first-comer / of a second-comer
            const int Y = std::min(32, (N - 32 * so + sy - 1) / sy), R = 32 + Y;
            int c = 0, o = 0, y = 0;
            for (int j = 0; j < R; ++j) {
                const bool young = (j + 1) * Y / R > j * Y / R;                     // spread evenly; the last run is a second-comer's
                const int n = std::min(young ? sy : so, N - c);
                runs.push_back(Run{c, c + n, young ? 32 + y++ : o++, young});
                c += n;
            }
            kch = so;
            if (c != N) { runs.clear(); kch = XG_MAXCH + 1; }                       // (cannot happen: 32 so + Y sy >= N)
        } else {
            for (int j = 0, c = 0; c < N; ++j, c += kch) runs.push_back(Run{c, std::min(c + kch, N), j, false});
        }
        for (int w = (int)runs.size(); w < m->nchunk; ++w) runs.push_back(Run{N, N, w, false});   // (more vector chunks than runs: workgroups without rows)
        const int Pn = (int)runs.size();
        bool ok = kch <= XG_MAXCH && Pn > 0;
        int maxq = 0, maxr = 0;
        // vector chunk v (rows 256 v ..) goes to the workgroup whose SpMV rows hold its first row -- to the first-comer beside it if that one
        // is a second-comer --, so that a workgroup's column range stays one neighbourhood of the mesh; where two chunks would meet in
        // one workgroup (runs longer than 256 rows): chunk v to workgroup v
        std::vector<int> vec_of(Pn, -1);                                            // by rank
        if (ok) {
            std::vector<int> run_of(std::max(N, 1), 0);
            for (int j = 0; j < Pn; ++j) for (int c = runs[j].c0; c < runs[j].c1; ++c) run_of[c] = j;
            bool clash = false;
            for (int v = 0; v < m->nchunk; ++v) {
                const int j = run_of[std::min((v * RPB) / SPB, N - 1)];
                const int cand[4] = {runs[j].young ? -1 : j, j > 0 && !runs[j - 1].young ? j - 1 : -1, j + 1 < Pn && !runs[j + 1].young ? j + 1 : -1, j};
                int take = -1;
                for (int k = 0; k < 4 && take < 0; ++k) if (cand[k] >= 0 && vec_of[runs[cand[k]].rank] < 0) take = cand[k];
                clash = clash || take < 0;
                if (take >= 0) vec_of[runs[take].rank] = v;
            }
            if (clash) for (int w = 0; w < Pn; ++w) vec_of[w] = w < m->nchunk ? w : -1;
        }
        P.xg.assign(ok ? Pn : 0, make_int4(0, 0, 0, 0));
        for (int j = 0; j < Pn && ok; ++j) {
            const int w = runs[j].rank, c0 = runs[j].c0, c1 = runs[j].c1;
            int lo = m->ndof, hi = 0;
            if (vec_of[w] >= 0) { lo = std::min(lo, vec_of[w] * RPB); hi = std::max(hi, std::min(vec_of[w] * RPB + RPB, m->ndof)); }   // the own vector chunk's rows
            for (int c = c0; c < c1; ++c) {
                const int r0 = c * SPB, r1 = std::min(r0 + SPB, m->ndof), q0 = P.bp[r0 / 3], q1 = P.bp[r1 / 3];
                maxq = std::max(maxq, q1 - q0);
                lo = std::min(lo, r0); hi = std::max(hi, r1);                                                              // the own rows (p.Ap)
                for (int q = q0; q < q1; ++q) { lo = std::min(lo, P.bcol3[q]); hi = std::max(hi, P.bcol3[q] + 3); }       // the columns
            }
            if (hi <= lo) { lo = 0; hi = 2; }
            lo &= ~1; hi = (hi + 1) & ~1;                       // 16-byte pieces of r
            maxr = std::max(maxr, hi - lo);
            P.xg[w] = make_int4(c0, c1 | (vec_of[w] + 1) << 16, lo, hi);   // (end of the chunk run | (vector chunk + 1) << 16)
        }
        const int mc = kch <= 1 ? 1 : kch <= 3 ? 3 : XG_MAXCH;   // the kernel's template variants
        const size_t lds = ((size_t)2 * maxr + (size_t)3 * mc * std::max(maxq, 1) + 2) * sizeof(double);   // (+ the zero slot)
        ok = ok && SPB == 48 && maxq <= XG_MAXQ * CGT && lds <= (Pn > 32 ? 60 : 150) * 1024;   // (two per compute unit: 160 KB for both, the two-level form's extras included)
        if (ok) { P.xcd = true; P.xg_P = Pn; P.xg_ldr = maxr; P.xg_ldq = std::max(maxq, 1); P.xg_lds = lds; P.xg_mc = mc; }
        else P.xg.clear();
    }
    return ORBX_OK;
}

// Model construction: plan on the host, then ONE device block for all tables and the node coordinates, filled through ONE
// pinned staging block by ONE copy on the model's stream (the first version issued about twenty blocking pageable copies;
// the reference builds a new mesh on every PoseOptimizationNR call, Optimizer.cc:480, so this is per-frame work).
int create_model(int eltype, int npe, const float *nodes, int nmesh, int nn, const int32_t *elems, int ne, unsigned int E, float nu,
                 float fg, Symbolic &y, int nseg, const int *seg_nn, const int *seg_ne, fem_model **out)
{
    fem_model *m = new fem_model();
    HostPlan P;
    int rc = plan_model(m, eltype, npe, nmesh, nn, ne, E, nu, fg, y, nseg, seg_nn, seg_ne, P);
    if (rc != ORBX_OK) { delete m; return rc; }
    const int nblk = m->nblk;
    const size_t M = (size_t)nmesh;
    m->cgr_big = P.big;

    // layout of the table block: 256-byte aligned pieces
    struct Piece { void **dst; const void *src; size_t bytes, room, off; };
    std::vector<Piece> pieces;
    size_t total = 0;
    auto piece = [&](auto **dst, const void *src, size_t bytes, size_t room = 0) {
        room = std::max(room, bytes);
        pieces.push_back(Piece{reinterpret_cast<void **>(dst), src, bytes, room, total});
        total += (std::max(room, (size_t)1) + 255) & ~(size_t)255;
    };
    piece(&m->d_nodes, nodes, sizeof(float) * M * nn * 3);
    piece(&m->d_elems, elems, sizeof(int) * (size_t)ne * npe);
    piece(&m->d_blk_row, y.blk_row.data(), sizeof(int) * (size_t)nblk);
    piece(&m->d_bptr, y.bptr.data(), sizeof(int) * ((size_t)nn + 1));
    piece(&m->d_cptr, y.cptr.data(), sizeof(int) * ((size_t)nblk + 1));
    piece(&m->d_contrib, y.contrib.data(), sizeof(int) * y.contrib.size());
    if (m->fused_lds) {
        piece(&m->d_nel_ptr, y.nel_ptr.data(), sizeof(int) * ((size_t)nn + 1));
        piece(&m->d_nel, y.nel.data(), sizeof(int) * y.nel.size());
        piece(&m->d_contrib_loc, y.contrib_loc.data(), sizeof(int) * y.contrib_loc.size());
    }
    piece(&m->d_rowptr, m->h_rowptr.data(), sizeof(int) * ((size_t)m->ndof + 1));
    piece(&m->d_lcol, m->h_lcol.data(), sizeof(int) * m->nnz, sizeof(int) * m->nnzs);   // the padding tail is read by the SpMV's last quad: zeros = valid columns
    piece(&m->d_diag, m->h_diag.data(), sizeof(int) * (size_t)m->ndof);
    piece(&m->d_bcol3, P.bcol3.data(), sizeof(int) * P.bcol3.size(), sizeof(int) * (P.bcol3.size() + 1));
    piece(&m->d_bp, P.bp.data(), sizeof(int) * P.bp.size(), sizeof(int) * (P.bp.size() + 1));
    if (seg_nn) {
        piece(&m->d_cmesh, P.cmesh.data(), sizeof(int) * P.cmesh.size());
        piece(&m->d_cmesh_s, P.cmesh_s.data(), sizeof(int) * P.cmesh_s.size());
        piece(&m->d_minfo, P.minfo.data(), sizeof(int4) * (size_t)nseg);
        piece(&m->d_minfo_s, P.minfo_s.data(), sizeof(int4) * (size_t)nseg);
    }
    if (P.resident) {
        piece(&m->d_rcd, P.rcd.data(), sizeof(int4) * P.rcd.size());
        piece(&m->d_rcfirst, P.rcfirst.data(), sizeof(int) * P.rcfirst.size());
    }
    if (P.xcd) piece(&m->d_xg_plan, P.xg.data(), sizeof(int4) * P.xg.size());

    int bad = 0;
    bad |= dalloc(&m->d_tables, total) | dalloc(&m->d_ke1, (size_t)m->nd * m->nd);
    if (!m->fused_lds) bad |= dalloc(&m->d_ke, M * ne * m->nd * m->nd);
    bad |= dalloc(&m->d_vals, M * m->nnzs);
    char *stage = bad ? nullptr : static_cast<char *>(g_pin_cache.get(total));
    if (bad || !stage || !(m->stream = stream_get())) {
        g_pin_cache.put(stage);
        fem_free(m); delete m;
        ORBX_FAIL(ORBX_ERR_HIP, "device allocation failed");
    }
    for (const Piece &pc : pieces) {
        *pc.dst = m->d_tables + pc.off;
        if (pc.bytes) memcpy(stage + pc.off, pc.src, pc.bytes);
        if (pc.room > pc.bytes) memset(stage + pc.off + pc.bytes, 0, pc.room - pc.bytes);
    }
    hipError_t e = hipMemcpyAsync(m->d_tables, stage, total, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);   // the staging block goes back to the cache
    g_pin_cache.put(stage);
    if (e == hipSuccess && P.resident) {
        if (P.big)
            for (const void *fn : {reinterpret_cast<const void *>(k_fem_cg_resident<true, false>), reinterpret_cast<const void *>(k_fem_cg_resident<true, true>)})
                { if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.resident_lds); }
        else
            for (const void *fn : {reinterpret_cast<const void *>(k_fem_cg_resident<false, false>), reinterpret_cast<const void *>(k_fem_cg_resident<false, true>)})
                if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.resident_lds);
    }
    if (e == hipSuccess && m->kz_lds > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_fem_cz_kz), hipFuncAttributeMaxDynamicSharedMemorySize, m->kz_lds);
    if (e == hipSuccess && m->spmv_lds > 48 * 1024)
        for (const void *fn : {reinterpret_cast<const void *>(k_fem_spmv<48, true>), reinterpret_cast<const void *>(k_fem_spmv<48, false>),
                               reinterpret_cast<const void *>(k_fem_spmv<96, true>), reinterpret_cast<const void *>(k_fem_spmv<96, false>)})
            if (e == hipSuccess) e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, m->spmv_lds);
    if (e != hipSuccess) {
        fem_free(m); delete m;
        return orbx::set_error(ORBX_ERR_HIP, hipGetErrorString(e), __FILE__, __LINE__);
    }
    if (P.resident) { m->cg_resident = true; m->cgr_lds = (int)P.resident_lds; m->cgr_ldn = P.maxrows; }
    if (P.xcd) { m->cg_xcd = true; m->xg_P = P.xg_P; m->xg_ldr = P.xg_ldr; m->xg_ldq = P.xg_ldq; m->xg_lds = (int)P.xg_lds; m->xg_mc = P.xg_mc; }
    m->name_kernel_kinds();
    *out = m;
    return ORBX_OK;
}

// The symbolic phase of a batch of meshes with their own topologies: per mesh (independent: a few host threads), then
// concatenated in global node / element / dof / non-zero numbering.  gelems receives the elements in global node ids.
int build_symbolic_batch(int npe, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const int32_t *elems,
                         const std::vector<long long> &node0, const std::vector<long long> &elem0, Symbolic &y, std::vector<int32_t> &gelems)
{
    std::vector<Symbolic> ys(nmesh);
    {
        const int nthr = std::max(1, std::min(nmesh, std::min(16, (int)std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back([&, t]() {
                for (int k = t; k < nmesh; k += nthr) build_symbolic(npe, mesh_nn[k], mesh_ne[k], elems + elem0[k] * npe, ys[k]);
            });
        for (std::thread &th : pool) th.join();
    }
    // offsets of every mesh's slice in every concatenated table, then the slices filled side by side
    std::vector<long long> blk0(nmesh + 1, 0), con0(nmesh + 1, 0), nel0(nmesh + 1, 0);
    for (int k = 0; k < nmesh; ++k) {
        blk0[k + 1] = blk0[k] + ys[k].nblk;
        con0[k + 1] = con0[k] + (long long)ys[k].contrib.size();
        nel0[k + 1] = nel0[k] + (long long)ys[k].nel.size();
        y.maxel = std::max(y.maxel, ys[k].maxel);
    }
    if (9 * blk0[nmesh] >= (1ll << 31) || con0[nmesh] >= (1ll << 31)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "batch exceeds 2^31 non-zeros");
    const long long nnt = node0[nmesh], nbt = blk0[nmesh];
    gelems.assign((size_t)elem0[nmesh] * npe, 0);
    y.blk_row.assign((size_t)nbt, 0); y.bptr.assign((size_t)nnt + 1, 0); y.cptr.assign((size_t)nbt + 1, 0);
    y.contrib.assign((size_t)con0[nmesh], 0); y.contrib_loc.assign((size_t)con0[nmesh], 0);
    y.nel_ptr.assign((size_t)nnt + 1, 0); y.nel.assign((size_t)nel0[nmesh], 0);
    y.rowptr.assign((size_t)3 * nnt + 1, 0); y.lcol.assign((size_t)9 * nbt, 0); y.diag.assign((size_t)3 * nnt, 0);
    auto place = [&](int k) {
        const Symbolic &z = ys[k];
        const int nd0 = (int)node0[k], el0 = (int)elem0[k], b0 = (int)blk0[k], nz0 = 9 * b0, c0 = (int)con0[k], n0 = (int)nel0[k];
        for (long long i = elem0[k] * npe; i < elem0[k + 1] * npe; ++i) gelems[i] = elems[i] + nd0;
        for (size_t i = 0; i < z.blk_row.size(); ++i) y.blk_row[b0 + i] = z.blk_row[i] + nd0;
        for (size_t i = 1; i < z.bptr.size(); ++i) y.bptr[nd0 + i] = z.bptr[i] + b0;
        for (size_t i = 1; i < z.cptr.size(); ++i) y.cptr[b0 + i] = z.cptr[i] + c0;
        for (size_t i = 0; i < z.contrib.size(); ++i) y.contrib[c0 + i] = z.contrib[i] + (el0 << 6);
        std::copy(z.contrib_loc.begin(), z.contrib_loc.end(), y.contrib_loc.begin() + c0);   // local element numbers: unchanged
        for (size_t i = 1; i < z.nel_ptr.size(); ++i) y.nel_ptr[nd0 + i] = z.nel_ptr[i] + n0;
        for (size_t i = 0; i < z.nel.size(); ++i) y.nel[n0 + i] = z.nel[i] + el0;
        for (size_t i = 0; i + 1 < z.rowptr.size(); ++i) y.rowptr[3 * (size_t)nd0 + i] = z.rowptr[i] + nz0;
        for (size_t i = 0; i < z.lcol.size(); ++i) y.lcol[(size_t)nz0 + i] = z.lcol[i] + 3 * nd0;
        for (size_t i = 0; i < z.diag.size(); ++i) y.diag[3 * (size_t)nd0 + i] = z.diag[i] + nz0;
    };
    {
        const int nthr = std::max(1, std::min(nmesh, std::min(16, (int)std::thread::hardware_concurrency())));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthr; ++t)
            pool.emplace_back([&, t]() { for (int k = t; k < nmesh; k += nthr) place(k); });
        for (std::thread &th : pool) th.join();
    }
    const long long blk0_total = nbt;
    y.rowptr[(size_t)3 * nnt] = (int)(9 * blk0_total);
    y.nblk = (int)blk0_total;
    return ORBX_OK;
}

// argument checks shared by fem_create_batch and fem_plan
int check_batch(int eltype, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const int32_t *elems, int &npe,
                std::vector<long long> &node0, std::vector<long long> &elem0)
{
    npe = eltype == FEM_C3D8 ? 8 : eltype == FEM_C3D6 ? 6 : eltype == FEM_TET4 ? 4 : 0;
    if (!npe) ORBX_FAIL(ORBX_ERR_ARG, "unknown element type");
    node0.assign(nmesh + 1, 0); elem0.assign(nmesh + 1, 0);
    for (int k = 0; k < nmesh; ++k) {
        if (mesh_nn[k] < 2 || mesh_ne[k] < 0) ORBX_FAIL(ORBX_ERR_ARG, "mesh with fewer than two nodes (FEA2.cc:1386: assembly refuses Ksize <= 3)");
        node0[k + 1] = node0[k] + mesh_nn[k]; elem0[k + 1] = elem0[k] + mesh_ne[k];
    }
    if (elem0[nmesh] >= (1ll << 25)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "too many elements");
    if (3 * node0[nmesh] >= (1ll << 31)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "batch exceeds 2^31 dofs");
    for (int k = 0; k < nmesh; ++k)
        for (long long i = elem0[k] * npe; i < elem0[k + 1] * npe; ++i)
            if (elems[i] < 0 || elems[i] >= mesh_nn[k]) ORBX_FAIL(ORBX_ERR_ARG, "element node id out of range");
    return ORBX_OK;
}

} // namespace

extern "C" {

int fem_second_layer(const float *top, int ntop, float h, float *nodes_out)
{
    if (!top || !nodes_out || ntop < 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    for (int i = 0; i < 3 * ntop; ++i) {
        nodes_out[i] = top[i];
        nodes_out[3 * ntop + i] = top[i] - h; // FEA2.cc:1189-1191: world xyz, not along normals
    }
    return ORBX_OK;
}

int fem_create(int eltype, const float *nodes, int nmesh, int nn, const int32_t *elems, int ne, unsigned int E, float nu,
               float fg, fem_model **out)
{
    if (!nodes || !elems || !out || nmesh < 1 || nn < 1 || ne < 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    const int npe = eltype == FEM_C3D8 ? 8 : eltype == FEM_C3D6 ? 6 : eltype == FEM_TET4 ? 4 : 0;
    if (!npe) ORBX_FAIL(ORBX_ERR_ARG, "unknown element type");
    if (3 * nn <= 3) ORBX_FAIL(ORBX_ERR_ARG, "Ksize<=3 (FEA2.cc:1386: assembly refuses)");
    for (int i = 0; i < ne * npe; ++i)
        if (elems[i] < 0 || elems[i] >= nn) ORBX_FAIL(ORBX_ERR_ARG, "element node id out of range");
    if ((long long)ne >= (1ll << 25)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "too many elements");
    if ((long long)nmesh * 3 * nn >= (1ll << 31)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "batch exceeds 2^31 dofs");
    ORBX_NEED_DEVICE();
    Symbolic y;
    build_symbolic(npe, nn, ne, elems, y);
    return create_model(eltype, npe, nodes, nmesh, nn, elems, ne, E, nu, fg, y, 0, nullptr, nullptr, out);
}

int fem_create_batch(int eltype, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const float *nodes, const int32_t *elems,
                     unsigned int E, float nu, float fg, fem_model **out)
{
    if (!nodes || !elems || !out || !mesh_nn || !mesh_ne || nmesh < 1) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    int npe = 0;
    std::vector<long long> node0, elem0;
    int rc = check_batch(eltype, nmesh, mesh_nn, mesh_ne, elems, npe, node0, elem0);
    if (rc != ORBX_OK) return rc;
    ORBX_NEED_DEVICE();
    Symbolic y;
    std::vector<int32_t> gelems;
    rc = build_symbolic_batch(npe, nmesh, mesh_nn, mesh_ne, elems, node0, elem0, y, gelems);
    if (rc != ORBX_OK) return rc;
    std::vector<int> snn(mesh_nn, mesh_nn + nmesh), sne(mesh_ne, mesh_ne + nmesh);
    return create_model(eltype, npe, nodes, 1, (int)node0[nmesh], gelems.data(), (int)elem0[nmesh], E, nu, fg, y, nmesh, snn.data(), sne.data(), out);
}

int fem_plan_selfcheck(int eltype, int nn, const int32_t *elems, int ne)
{
    const int npe = eltype == FEM_C3D8 ? 8 : eltype == FEM_C3D6 ? 6 : eltype == FEM_TET4 ? 4 : 0;
    if (!npe || !elems || nn < 1 || ne < 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    for (int i = 0; i < ne * npe; ++i)
        if (elems[i] < 0 || elems[i] >= nn) ORBX_FAIL(ORBX_ERR_ARG, "element node id out of range");
    Symbolic a, b;
    build_symbolic(npe, nn, ne, elems, a);
    build_symbolic_simple(npe, nn, ne, elems, b);
    const bool same = a.nblk == b.nblk && a.maxel == b.maxel && a.bptr == b.bptr && a.bcol == b.bcol && a.blk_row == b.blk_row &&
                      a.cptr == b.cptr && a.contrib == b.contrib && a.contrib_loc == b.contrib_loc && a.nel_ptr == b.nel_ptr &&
                      a.nel == b.nel && a.rowptr == b.rowptr && a.lcol == b.lcol && a.diag == b.diag;
    if (!same) ORBX_FAIL(ORBX_ERR_HIP, "the two formulations of the symbolic phase disagree");
    return ORBX_OK;
}

int fem_plan(int eltype, int nmesh, const int32_t *mesh_nn, const int32_t *mesh_ne, const int32_t *elems, int uniform_copies,
             fem_plan_info *info, int32_t *rowptr, int32_t *lcol, int32_t *diag, int32_t *bp, int32_t *bcol3, int32_t *rcd, int32_t *rcfirst,
             int32_t *chunk_mesh)
{
    if (!elems || !mesh_nn || !mesh_ne || !info || nmesh < 1 || uniform_copies < 0 || (uniform_copies && nmesh != 1))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    int npe = 0;
    std::vector<long long> node0, elem0;
    int rc = check_batch(eltype, nmesh, mesh_nn, mesh_ne, elems, npe, node0, elem0);
    if (rc != ORBX_OK) return rc;
    fem_model m;
    HostPlan P;
    Symbolic y;
    if (uniform_copies) {     // fem_create's layout: `uniform_copies` meshes sharing the one topology
        if ((long long)uniform_copies * 3 * mesh_nn[0] >= (1ll << 31)) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "batch exceeds 2^31 dofs");
        build_symbolic(npe, mesh_nn[0], mesh_ne[0], elems, y);
        rc = plan_model(&m, eltype, npe, uniform_copies, mesh_nn[0], mesh_ne[0], 3500, 0.495f, 0.577350269f, y, 0, nullptr, nullptr, P);
    } else {                  // fem_create_batch's
        std::vector<int32_t> gelems;
        rc = build_symbolic_batch(npe, nmesh, mesh_nn, mesh_ne, elems, node0, elem0, y, gelems);
        if (rc != ORBX_OK) return rc;
        std::vector<int> snn(mesh_nn, mesh_nn + nmesh), sne(mesh_ne, mesh_ne + nmesh);
        rc = plan_model(&m, eltype, npe, 1, (int)node0[nmesh], (int)elem0[nmesh], 3500, 0.495f, 0.577350269f, y, nmesh, snn.data(), sne.data(), P);
    }
    if (rc != ORBX_OK) return rc;
    memset(info, 0, sizeof(*info));
    info->ndof = m.ndof; info->nblk = m.nblk; info->nnz = (int64_t)m.nnz; info->spb = m.spb; info->spmv_lds = m.spmv_lds;
    info->fused_lds = m.fused_lds; info->rows_lds = m.rows_lds; info->nchunk_tot = m.nchunk_tot; info->nchunk_s_tot = m.nchunk_s_tot;
    info->resident = P.resident ? 1 : 0; info->resident_big = P.big ? 1 : 0; info->resident_lds = (int32_t)P.resident_lds;
    info->nrcd = P.resident ? (int32_t)P.rcd.size() : 0; info->ncontrib = (int64_t)y.contrib.size(); info->maxel = y.maxel;
    if (rowptr) memcpy(rowptr, m.h_rowptr.data(), sizeof(int) * m.h_rowptr.size());
    if (lcol) memcpy(lcol, m.h_lcol.data(), sizeof(int) * m.h_lcol.size());
    if (diag) memcpy(diag, m.h_diag.data(), sizeof(int) * m.h_diag.size());
    if (bp) memcpy(bp, P.bp.data(), sizeof(int) * P.bp.size());
    if (bcol3) memcpy(bcol3, P.bcol3.data(), sizeof(int) * P.bcol3.size());
    if (rcd && P.resident) memcpy(rcd, P.rcd.data(), sizeof(int4) * P.rcd.size());
    if (rcfirst && P.resident) memcpy(rcfirst, P.rcfirst.data(), sizeof(int) * P.rcfirst.size());
    if (chunk_mesh && !uniform_copies) memcpy(chunk_mesh, P.cmesh.data(), sizeof(int) * P.cmesh.size());
    return ORBX_OK;
}

int fem_plan_single_cg(int eltype, int nn, const int32_t *elems, int ne, int32_t *info6, int32_t *plan)
{
    if (!elems || !info6 || nn < 1 || ne < 1) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    int npe = 0;
    std::vector<long long> node0, elem0;
    const int32_t mnn[1] = {nn}, mne[1] = {ne};
    int rc = check_batch(eltype, 1, mnn, mne, elems, npe, node0, elem0);
    if (rc != ORBX_OK) return rc;
    fem_model m;
    HostPlan P;
    Symbolic y;
    build_symbolic(npe, nn, ne, elems, y);
    rc = plan_model(&m, eltype, npe, 1, nn, ne, 3500, 0.495f, 0.577350269f, y, 0, nullptr, nullptr, P);
    if (rc != ORBX_OK) return rc;
    const int v[6] = {P.xcd ? 1 : 0, P.xg_P, P.xg_mc, (int)P.xg_lds, m.nchunk, m.nchunk_s};
    memcpy(info6, v, sizeof(v));
    if (plan && P.xcd) memcpy(plan, P.xg.data(), sizeof(int4) * P.xg.size());
    return ORBX_OK;
}

int fem_batch_offsets(const fem_model *m, int32_t *node0, int32_t *elem0, int32_t *nnz0)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    for (int k = 0; k <= m->nseg; ++k) {
        if (node0) node0[k] = m->segmented() ? m->seg_node0[k] : k * m->nn;
        if (elem0) elem0[k] = m->segmented() ? m->seg_elem0[k] : k * m->ne;
        if (nnz0) nnz0[k] = m->segmented() ? m->seg_nnz0[k] : (int32_t)((size_t)k * m->nnz);
    }
    return ORBX_OK;
}

int fem_destroy(fem_model *m)
{
    if (!m) return ORBX_OK;
    fem_free(m);
    delete m;
    return ORBX_OK;
}

int fem_sizes(const fem_model *m, int *nmesh, int *ndof, int64_t *nnz)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    if (nmesh) *nmesh = m->nseg;
    if (ndof) *ndof = m->ndof;
    if (nnz) *nnz = (int64_t)m->nnz;
    return ORBX_OK;
}

int fem_material(const fem_model *m, float *lambda, float *G, float *D36)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    if (lambda) *lambda = m->lambda;
    if (G) *G = m->G;
    if (D36) memcpy(D36, m->fc.D, sizeof(float) * 36);
    return ORBX_OK;
}

int fem_assemble(fem_model *m)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    hipStream_t st = m->stream;
    if (m->fused_lds) {
        // K_e never leaves the chip: one workgroup per node forms and sums the contributions of its block row
        m->prof.start(1, st);
        const dim3 g(m->nn, m->nmesh);
#define ORBX_FUSED(NPE, ELT)                                                                                                    \
        do {                                                                                                                    \
            if (m->rows_lds) {                                                                                                  \
                if (m->rows_lds > 48 * 1024)                                                                                    \
                    ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fem_assemble_rows<NPE, ELT>),                 \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, m->rows_lds));                     \
                hipLaunchKernelGGL((k_fem_assemble_rows<NPE, ELT>), g, dim3(ROWS_T), m->rows_lds, st, m->d_nodes, m->nn,           \
                                   m->d_elems, m->fc, m->glimit, m->d_bptr, m->d_cptr, m->d_contrib_loc, m->d_nel_ptr,          \
                                   m->d_nel, m->d_rowptr, m->d_vals, m->nnzs);                                                  \
                break;                                                                                                          \
            }                                                                                                                   \
            if (m->fused_lds > 48 * 1024)                                                                                       \
                ORBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(k_fem_assemble_fused<NPE, ELT>),                    \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, m->fused_lds));                        \
            hipLaunchKernelGGL((k_fem_assemble_fused<NPE, ELT>), g, dim3(256), m->fused_lds, st, m->d_nodes, m->nn, m->d_elems,  \
                               m->fc, m->d_bptr, m->d_cptr, m->d_contrib_loc, m->d_nel_ptr, m->d_nel, m->d_rowptr, m->d_vals,  \
                               m->nnzs);                                                                                        \
        } while (0)
        if (m->eltype == FEM_C3D8) ORBX_FUSED(8, FEM_C3D8);
        else if (m->eltype == FEM_C3D6) ORBX_FUSED(6, FEM_C3D6);
        else ORBX_FUSED(4, FEM_TET4);
#undef ORBX_FUSED
        m->prof.stop(1, st);
    } else {
        // a node with so many elements that their Gauss-point data does not fit LDS: K_e through HBM
        m->prof.start(0, st);
        if (m->ne > 0) {
            const dim3 g(m->ne, m->nmesh);
            if (m->eltype == FEM_C3D8)
                hipLaunchKernelGGL((k_fem_ke<8, FEM_C3D8>), g, dim3(64), 0, st, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke, 0, 0);
            else if (m->eltype == FEM_C3D6)
                hipLaunchKernelGGL((k_fem_ke<6, FEM_C3D6>), g, dim3(64), 0, st, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke, 0, 0);
            else
                hipLaunchKernelGGL((k_fem_ke<4, FEM_TET4>), g, dim3(64), 0, st, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke, 0, 0);
        }
        m->prof.stop(0, st);
        m->prof.start(1, st);
        hipLaunchKernelGGL(k_fem_assemble, dim3((m->nblk * 9 + 255) / 256, m->nmesh), dim3(256), 0, st, m->d_ke, m->ne, m->nd,
                           m->nblk, m->d_blk_row, m->d_bptr, m->d_cptr, m->d_contrib, m->d_rowptr, m->d_vals, m->nnzs);
        m->prof.stop(1, st);
    }
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipStreamSynchronize(st));
    m->assembled = true;
    m->cg_ready = false;
    m->h_cmask.clear();   // a fresh K has no constrained dofs
    m->cz_space_valid = false;
    return ORBX_OK;
}

int fem_dirichlet_penalty(fem_model *m, const int32_t *ids, int nids, float klarge)
{
    if (!m || !m->assembled || nids < 0 || (nids && !ids)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    for (int i = 0; i < nids; ++i)
        if (ids[i] - 1 < 0 || ids[i] - 1 >= m->nn) ORBX_FAIL(ORBX_ERR_ARG, "Dirichlet id out of range (reference would write out of bounds)");
    if (nids == 0) return ORBX_OK;
    // the id list through a pinned block that the kernel reads itself (it is mapped into the device's address space): no
    // device allocation, no blocking pageable copy in front of a 3-us kernel
    const size_t bytes = sizeof(int) * (size_t)nids;
    int *h_ids = static_cast<int *>(g_pin_cache.get(bytes));
    if (!h_ids) ORBX_FAIL(ORBX_ERR_HIP, "pinned allocation failed");
    memcpy(h_ids, ids, bytes);
    hipLaunchKernelGGL(k_fem_penalty, dim3((nids * 3 + 255) / 256, m->nmesh), dim3(256), 0, m->stream, m->d_vals, m->nnzs,
                       m->d_diag, (const int *)h_ids, nids, klarge);
    const hipError_t e = hipStreamSynchronize(m->stream);
    g_pin_cache.put(h_ids);
    ORBX_HIP(e);
    m->h_cmask.resize(m->ndof, 0);
    for (int i = 0; i < nids; ++i)
        for (int k = 0; k < 3; ++k) m->h_cmask[3 * (ids[i] - 1) + k] = 1;
    m->cz_space_valid = false;
    m->cg_ready = false;
    return ORBX_OK;
}

int fem_dirichlet_eliminate(fem_model *m, const int32_t *dofs, int ndofs)
{
    if (!m || !m->assembled || ndofs < 0 || (ndofs && !dofs)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    std::vector<uint8_t> fixed(m->ndof, 0);
    for (int i = 0; i < ndofs; ++i) {
        if (dofs[i] < 0 || dofs[i] >= m->ndof) ORBX_FAIL(ORBX_ERR_ARG, "dof out of range");
        fixed[dofs[i]] = 1;
    }
    uint8_t *d_fixed = nullptr;
    if (dalloc(&d_fixed, (size_t)m->ndof)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    ORBX_HIP(hipMemcpyAsync(d_fixed, fixed.data(), m->ndof, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_fem_eliminate, dim3((m->nblk + 255) / 256, m->nmesh), dim3(256), 0, m->stream, m->d_vals,
                       m->d_blk_row, m->d_bcol3, m->d_bp, m->d_rowptr, m->nblk, m->nnzs, d_fixed);
    ORBX_HIP(hipStreamSynchronize(m->stream));
    dfree(d_fixed);
    m->h_cmask.resize(m->ndof, 0);
    for (int i = 0; i < m->ndof; ++i) m->h_cmask[i] |= fixed[i];
    m->cz_space_valid = false;
    m->cg_ready = false;
    return ORBX_OK;
}

int fem_get_ke(fem_model *m, int mesh, int elem, float *ke)
{
    if (!m || !m->assembled || !ke || mesh < 0 || mesh >= m->nseg || elem < 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    size_t e;
    if (m->segmented()) {
        if (elem >= m->seg_elem0[mesh + 1] - m->seg_elem0[mesh]) ORBX_FAIL(ORBX_ERR_ARG, "element out of range");
        e = (size_t)m->seg_elem0[mesh] + elem;
    } else {
        if (elem >= m->ne) ORBX_FAIL(ORBX_ERR_ARG, "element out of range");
        e = (size_t)mesh * m->ne + elem;
    }
    // one element's K_e on demand (the assembly keeps K_e on the chip)
    const int mesh0 = m->segmented() ? 0 : mesh, el = (int)(m->segmented() ? e : e - (size_t)mesh * m->ne);
    if (m->eltype == FEM_C3D8)
        hipLaunchKernelGGL((k_fem_ke<8, FEM_C3D8>), dim3(1), dim3(64), 0, m->stream, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke1, el, mesh0);
    else if (m->eltype == FEM_C3D6)
        hipLaunchKernelGGL((k_fem_ke<6, FEM_C3D6>), dim3(1), dim3(64), 0, m->stream, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke1, el, mesh0);
    else
        hipLaunchKernelGGL((k_fem_ke<4, FEM_TET4>), dim3(1), dim3(64), 0, m->stream, m->d_nodes, m->nn, m->d_elems, m->ne, m->fc, m->d_ke1, el, mesh0);
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipMemcpyAsync(ke, m->d_ke1, sizeof(float) * m->nd * m->nd, hipMemcpyDeviceToHost, m->stream));
    ORBX_HIP(hipStreamSynchronize(m->stream));
    return ORBX_OK;
}

int fem_get_csr(fem_model *m, int mesh, int32_t *rowptr, int32_t *col, float *val)
{
    if (!m || !m->assembled || mesh < 0 || mesh >= m->nseg) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    if (m->segmented()) { // the mesh's own CSR: local rows and columns
        const int row0 = 3 * m->seg_node0[mesh], nrows = 3 * (m->seg_node0[mesh + 1] - m->seg_node0[mesh]);
        const int k0 = m->seg_nnz0[mesh], k1 = m->seg_nnz0[mesh + 1];
        if (rowptr) for (int r = 0; r <= nrows; ++r) rowptr[r] = m->h_rowptr[row0 + r] - k0;
        if (col) for (int k = k0; k < k1; ++k) col[k - k0] = m->h_lcol[k] - row0;
        if (val) { ORBX_HIP(hipMemcpyAsync(val, m->d_vals + k0, sizeof(float) * (size_t)(k1 - k0), hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
        return ORBX_OK;
    }
    if (rowptr) memcpy(rowptr, m->h_rowptr.data(), sizeof(int) * (m->ndof + 1));
    if (col) memcpy(col, m->h_lcol.data(), sizeof(int) * m->nnz);
    if (val) { ORBX_HIP(hipMemcpyAsync(val, m->d_vals + (size_t)mesh * m->nnzs, sizeof(float) * m->nnz, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    return ORBX_OK;
}

int fem_displacement(fem_model *m, const float *uf, const float *u0, const int32_t *ids, int nids, float klarge, float *a)
{
    if (!m || !uf || !u0 || !a || nids < 0 || (nids && !ids)) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    for (int i = 0; i < nids; ++i)
        if (ids[i] - 1 < 0 || ids[i] - 1 >= m->nn) ORBX_FAIL(ORBX_ERR_ARG, "Dirichlet id out of range");
    if (ensure_vecs(m)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    const size_t N = (size_t)m->nmesh * m->ndof;
    ORBX_HIP(hipMemcpyAsync(m->d_a, uf, sizeof(float) * N, hipMemcpyHostToDevice, m->stream));
    ORBX_HIP(hipMemcpyAsync(m->d_u, u0, sizeof(float) * N, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_fem_displacement, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, m->stream, m->d_a, m->d_u, m->d_f, N);
    if (nids) {
        int *d_ids = nullptr;
        if (dalloc(&d_ids, (size_t)nids)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
        ORBX_HIP(hipMemcpyAsync(d_ids, ids, sizeof(int) * nids, hipMemcpyHostToDevice, m->stream));
        hipLaunchKernelGGL(k_fem_displacement_dir, dim3((nids * 3 + 255) / 256, m->nmesh), dim3(256), 0, m->stream, m->d_f,
                           m->ndof, d_ids, nids, klarge);
        ORBX_HIP(hipStreamSynchronize(m->stream));
        dfree(d_ids);
    }
    ORBX_HIP(hipStreamSynchronize(m->stream));
    { ORBX_HIP(hipMemcpyAsync(a, m->d_f, sizeof(float) * N, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    return ORBX_OK;
}

int fem_matvec(fem_model *m, const float *a, float *f)
{
    if (!m || !m->assembled || !a || !f) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    if (ensure_vecs(m)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    const size_t N = (size_t)m->nmesh * m->ndof;
    ORBX_HIP(hipMemcpyAsync(m->d_a, a, sizeof(float) * N, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_fem_matvec, dim3((m->ndof + 15) / 16, m->nmesh), dim3(256), 0, m->stream, m->d_vals, m->d_lcol,
                       m->d_rowptr, m->nnzs, m->ndof, m->d_a, m->d_f);
    ORBX_HIP(hipStreamSynchronize(m->stream));
    { ORBX_HIP(hipMemcpyAsync(f, m->d_f, sizeof(float) * N, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    return ORBX_OK;
}

int fem_strain_energy(fem_model *m, const float *a, float *sE, float *nsE)
{
    if (!m || !m->assembled || !a) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    if (ensure_vecs(m)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    const size_t N = (size_t)m->nmesh * m->ndof;
    ORBX_HIP(hipMemcpyAsync(m->d_a, a, sizeof(float) * N, hipMemcpyHostToDevice, m->stream));
    hipLaunchKernelGGL(k_fem_matvec, dim3((m->ndof + 15) / 16, m->nmesh), dim3(256), 0, m->stream, m->d_vals, m->d_lcol,
                       m->d_rowptr, m->nnzs, m->ndof, m->d_a, m->d_f);
    hipLaunchKernelGGL(k_fem_energy, dim3(m->nseg), dim3(256), 0, m->stream, m->d_a, m->d_f, m->ndof, m->d_e, m->d_e + m->nseg,
                       (const int4 *)m->d_minfo);
    ORBX_HIP(hipStreamSynchronize(m->stream));
    if (sE) { ORBX_HIP(hipMemcpyAsync(sE, m->d_e, sizeof(float) * m->nseg, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    if (nsE) { ORBX_HIP(hipMemcpyAsync(nsE, m->d_e + m->nseg, sizeof(float) * m->nseg, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    return ORBX_OK;
}

int fem_trial_setup(fem_model *m, const float *u0, const int32_t *ids, int nids, float klarge, int npoints,
                    const int32_t *derived, int nder)
{
    if (!m || !m->assembled || !u0 || nids < 0 || (nids && !ids) || npoints < 0 || nder < 0 || (nder && !derived))
        ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    if (m->segmented()) ORBX_FAIL(ORBX_ERR_UNSUPPORTED, "the LM hook works on one mesh per model (fem_create)");
    const int nTop = npoints + nder;
    if (6 * nTop != m->ndof) ORBX_FAIL(ORBX_ERR_ARG, "npoints + nder must equal the number of top-layer nodes (Ksize / 6)");
    for (int i = 0; i < nids; ++i)
        if (ids[i] - 1 < 0 || ids[i] - 1 >= m->nn) ORBX_FAIL(ORBX_ERR_ARG, "Dirichlet id out of range");
    int seq = 0;
    for (int d = 0; d < nder; ++d) {
        const int c = derived[4 * d];
        if (c != 2 && c != 3) ORBX_FAIL(ORBX_ERR_ARG, "derived node needs 2 or 3 base points (FEA2.cc:1749,1761)");
        for (int k = 1; k <= c; ++k) {
            if (derived[4 * d + k] < 0 || derived[4 * d + k] >= npoints + d) ORBX_FAIL(ORBX_ERR_ARG, "derived base index out of range");
            if (derived[4 * d + k] >= npoints) seq = 1; // built on an earlier derived node: keep the reference's order
        }
    }
    void *old[] = {m->d_tr_points, m->d_tr_top, m->d_tr_u0, m->d_tr_derived, m->d_tr_ids, m->d_tr_done};
    for (void *q : old)
        if (q) dfree(q);
    m->d_tr_points = nullptr; m->d_tr_top = m->d_tr_u0 = nullptr; m->d_tr_derived = m->d_tr_ids = nullptr; m->d_tr_done = nullptr;
    if (ensure_vecs(m) || dalloc(&m->d_tr_points, (size_t)m->nmesh * npoints * 3) || dalloc(&m->d_tr_top, (size_t)m->nmesh * nTop * 3) ||
        dalloc(&m->d_tr_u0, (size_t)m->ndof + 4) || dalloc(&m->d_tr_derived, (size_t)4 * nder + 4) || dalloc(&m->d_tr_ids, (size_t)nids + 4) ||
        dalloc(&m->d_tr_done, (size_t)m->nmesh))
        ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    {   // u0, the derived-node table and the Dirichlet list through ONE pinned block, copied by the compute queue (common.h)
        const size_t b0 = (sizeof(float) * (size_t)m->ndof + 15) & ~(size_t)15, b1 = (sizeof(int) * 4 * (size_t)nder + 15) & ~(size_t)15,
                     b2 = (sizeof(int) * (size_t)nids + 15) & ~(size_t)15;
        char *h = static_cast<char *>(g_pin_cache.get(b0 + b1 + b2));
        if (!h) ORBX_FAIL(ORBX_ERR_HIP, "pinned allocation failed");
        memcpy(h, u0, sizeof(float) * m->ndof);
        if (nder) memcpy(h + b0, derived, sizeof(int) * 4 * nder);
        if (nids) memcpy(h + b0 + b1, ids, sizeof(int) * nids);
        hipError_t e = orbx::stage_in(m->d_tr_u0, h, b0, m->stream);
        if (e == hipSuccess && nder) e = orbx::stage_in(m->d_tr_derived, h + b0, b1, m->stream);
        if (e == hipSuccess && nids) e = orbx::stage_in(m->d_tr_ids, h + b0 + b1, b2, m->stream);
        if (e == hipSuccess) e = hipMemsetAsync(m->d_tr_done, 0, sizeof(unsigned) * m->nmesh, m->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
        g_pin_cache.put(h);
        ORBX_HIP(e);
    }
    m->tr_npoints = npoints; m->tr_nder = nder; m->tr_nids = nids; m->tr_seq = seq; m->tr_klarge = klarge;
    const size_t pin_bytes = sizeof(double) * (size_t)m->nmesh * npoints * 3 + sizeof(float) * (size_t)m->nmesh * (m->ndof + 2);
    if (pin_bytes > m->h_tr_pin_bytes) {
        g_pin_cache.put(m->h_tr_pin);
        m->h_tr_pin_bytes = 0;
        m->h_tr_pin = static_cast<char *>(g_pin_cache.get(pin_bytes));
        if (!m->h_tr_pin) ORBX_FAIL(ORBX_ERR_HIP, "pinned allocation failed");
        m->h_tr_pin_bytes = pin_bytes;
    }
    m->trial_ready = true;
    return ORBX_OK;
}

int fem_trial_energy(fem_model *m, const double *points, float *a_out, float *sE, float *nsE)
{
    if (!m || !m->trial_ready || !points) ORBX_FAIL(ORBX_ERR_ARG, "call fem_trial_setup first");
    hipStream_t st = m->stream;
    const size_t pbytes = sizeof(double) * (size_t)m->nmesh * m->tr_npoints * 3, abytes = sizeof(float) * (size_t)m->nmesh * m->ndof;
    double *h_points = reinterpret_cast<double *>(m->h_tr_pin);
    float *h_a = reinterpret_cast<float *>(m->h_tr_pin + pbytes), *h_e = h_a + (size_t)m->nmesh * m->ndof;
    memcpy(h_points, points, pbytes);
    // the kernels read the estimates from, and write the two energies into, the pinned block themselves (it is mapped into the
    // device's address space): a copy-engine transfer on either side of two short kernels costs more than they do (common.h)
    hipLaunchKernelGGL(k_fem_trial_a_fused, dim3(m->nmesh), dim3(TRIAL_T), 0, st, (const double *)h_points, m->tr_npoints, m->d_tr_derived,
                       m->tr_nder, m->tr_seq, m->d_tr_top, m->d_tr_u0, m->d_a, m->d_tr_ids, m->tr_nids, m->tr_klarge);
    hipLaunchKernelGGL(k_fem_matvec_energy, dim3((m->ndof + 15) / 16, m->nmesh), dim3(256), 0, st, m->d_vals, m->d_lcol, m->d_rowptr,
                       m->nnzs, m->ndof, m->d_a, m->d_f, m->d_tr_done, h_e, h_e + m->nmesh);
    ORBX_HIP(hipGetLastError());
    if (a_out) ORBX_HIP(hipMemcpyAsync(h_a, m->d_a, abytes, hipMemcpyDeviceToHost, st));
    ORBX_HIP(hipStreamSynchronize(st));
    if (a_out) memcpy(a_out, h_a, abytes);
    if (sE) memcpy(sE, h_e, sizeof(float) * m->nmesh);
    if (nsE) memcpy(nsE, h_e + m->nmesh, sizeof(float) * m->nmesh);
    return ORBX_OK;
}

int fem_cg_setup(fem_model *m, const double *b)
{
    if (!m || !m->assembled || !b) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments / not assembled");
    if (ensure_cg(m)) ORBX_FAIL(ORBX_ERR_HIP, "hipMalloc failed");
    const size_t N = (size_t)m->nmesh * m->ndof;
    ORBX_HIP(hipMemcpyAsync(m->d_b, b, sizeof(double) * N, hipMemcpyHostToDevice, m->stream));   // on the model's stream: a copy on the legacy stream waits for every other thread's work
    // the values as they stand now (assembled, penalties applied), block-major, for k_fem_spmv
    hipLaunchKernelGGL(k_fem_to_blocks, dim3((m->nblk + 255) / 256, m->nmesh), dim3(256), 0, m->stream, m->d_vals, m->d_vals_b, m->d_rowptr,
                       m->d_bp, m->d_blk_row, m->nblk, m->nnzs);
    if (m->coarse() && setup_coarse(m)) ORBX_FAIL(ORBX_ERR_HIP, "coarse space of the two-level preconditioner: allocation or copy failed");
    hipLaunchKernelGGL(k_fem_cg_init, grid_cg(m), dim3(CGT), 0, m->stream, m->d_vals, m->d_diag, m->nnzs,
                       m->ndof, m->nchunk, m->d_b, m->d_x, m->d_r, m->d_p, m->d_dinv, m->d_part[0], m->d_part[1],
                       (const int *)m->d_cmesh, (const int4 *)m->d_minfo);
    if (m->coarse()) coarse_correction(m, m->stream, m->d_r, m->d_p, 1);   // p = z = r/diag + Z Ac^-1 Z^T r
    hipLaunchKernelGGL(k_fem_cg_init2, dim3(m->nseg), dim3(1), 0, m->stream, m->nchunk, m->d_part[0], m->d_part[1], m->d_sc,
                       (const int4 *)m->d_minfo, m->coarse() ? (const double *)m->d_cwv : nullptr);
    if (m->d_xg_ctl) {
        ORBX_HIP(hipMemsetAsync(m->d_xg_ctl, 0, sizeof(XgCtl), m->stream));
        ORBX_HIP(hipMemsetAsync(m->d_xg_gran, 0, (size_t)16 * xg_layout(m->ndof, m->nchunk, m->nchunk_s).total, m->stream));
        m->xg_bar = 0;
    }
    ORBX_HIP(hipGetLastError());
    ORBX_HIP(hipStreamSynchronize(m->stream));
    m->cg_it = 0;
    m->xg_pending.clear();
    m->cg_ready = true;
    return ORBX_OK;
}

int fem_cg_one_launch_stats(fem_model *m, int64_t *launches, int64_t *recovered)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    if (launches) *launches = m->xg_launches;
    if (recovered) *recovered = m->xg_recovered;
    return ORBX_OK;
}

int fem_cg_preconditioner(fem_model *m, int kind)
{
    if (!m || (kind != FEM_PRECOND_JACOBI && kind != FEM_PRECOND_TWO_LEVEL)) ORBX_FAIL(ORBX_ERR_ARG, "unknown preconditioner");
    if (kind != m->precond) {
        m->precond = kind;
        m->cg_ready = false;
        m->name_kernel_kinds();
    }
    return ORBX_OK;
}

int fem_cg_coarse_matrix(fem_model *m, int mesh, double *Ac)
{
    if (!m || !Ac || mesh < 0 || mesh >= m->nseg) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    if (!m->coarse() || !m->cg_ready) ORBX_FAIL(ORBX_ERR_ARG, "no two-level preconditioner set up (fem_cg_preconditioner, fem_cg_setup)");
    { ORBX_HIP(hipMemcpyAsync(Ac, m->d_ac + (size_t)CZ_NC * CZ_NC * mesh, sizeof(double) * CZ_NC * CZ_NC, hipMemcpyDeviceToHost, m->stream)); ORBX_HIP(hipStreamSynchronize(m->stream)); }
    return ORBX_OK;
}

// (Until round 4 small batches replayed a captured hipGraph of 50 iterations here.  While ANY stream captures, another thread's copy
// on the legacy stream fails -- "would make the legacy stream depend on a capturing blocking stream", also with the capture on a
// non-blocking stream of its own -- and invalidates the capture: tests/stress_threads.py.  The replay was worth 2 % on a single
// mesh (97 k against 99-102 k iterations/s): the kernels run back to back either way.  Taken out.)
int fem_cg_iterate(fem_model *m, int n, void *stream)
{
    if (!m || !m->cg_ready || n < 0) ORBX_FAIL(ORBX_ERR_ARG, "call fem_cg_setup first");
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    m->cg_stream = st;
    run_iters(m, n, st);
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int fem_spmv_repeat(fem_model *m, int n, void *stream)
{
    if (!m || !m->cg_ready || n < 0) ORBX_FAIL(ORBX_ERR_ARG, "call fem_cg_setup first");
    hipStream_t st = stream ? (hipStream_t)stream : m->stream;
    for (int i = 0; i < n; ++i) {
        m->prof.start(2, st);
        launch_spmv(m, st);
        m->prof.stop(2, st);
    }
    ORBX_HIP(hipGetLastError());
    return ORBX_OK;
}

int fem_cg_result(fem_model *m, double *x, double *relres)
{
    if (!m || !m->cg_ready) ORBX_FAIL(ORBX_ERR_ARG, "call fem_cg_setup first");
    // the copies wait for the stream the iterations ran on, not for the whole device
    hipStream_t st = m->cg_stream ? m->cg_stream : m->stream;
    std::vector<CgScal> sc(relres ? m->nseg : 0);
    // k_fem_cg_xcd launches since the last check: the two control words come along with the results; had a launch given up, the iterations
    // are made good (xg_recover) and the results fetched again
    const bool chk = m->d_xg_ctl && !m->xg_pending.empty();
    unsigned ctl2[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        if (chk && pass == 0) ORBX_HIP(hipMemcpyAsync(ctl2, m->d_xg_ctl, sizeof(ctl2), hipMemcpyDeviceToHost, st));
        if (x) ORBX_HIP(hipMemcpyAsync(x, m->d_x, sizeof(double) * (size_t)m->nmesh * m->ndof, hipMemcpyDeviceToHost, st));
        if (relres) ORBX_HIP(hipMemcpyAsync(sc.data(), m->d_sc, sizeof(CgScal) * m->nseg, hipMemcpyDeviceToHost, st));
        xg_settle(m, true);
        ORBX_HIP(hipStreamSynchronize(st));
        if (!chk || pass == 1) break;
        const int rc = xg_recover(m, st, ctl2);
        if (rc != ORBX_OK) return rc;
        if (!ctl2[0]) break;
    }
    if (relres) {
        for (int i = 0; i < m->nseg; ++i) relres[i] = sc[i].bb > 0 ? sqrt(sc[i].rr / sc[i].bb) : 0.0;
    }
    return ORBX_OK;
}

int fem_cg(fem_model *m, const double *b, double *x, int iters, double tol, int *iters_done, double *relres)
{
    if (!x || iters < 0) ORBX_FAIL(ORBX_ERR_ARG, "bad arguments");
    int rc = fem_cg_setup(m, b);
    if (rc != ORBX_OK) return rc;
    std::vector<CgScal> sc(m->nseg);
    int done = 0;
    while (done < iters) {
        if (tol > 0) { // convergence test on the device-side scalars, every 25 iterations
            const bool chk = m->d_xg_ctl && !m->xg_pending.empty();   // (a one-launch slice that gave up is made good before its residual is looked at)
            unsigned ctl2[2] = {0, 0};
            if (chk) ORBX_HIP(hipMemcpyAsync(ctl2, m->d_xg_ctl, sizeof(ctl2), hipMemcpyDeviceToHost, m->stream));
            ORBX_HIP(hipMemcpyAsync(sc.data(), m->d_sc, sizeof(CgScal) * m->nseg, hipMemcpyDeviceToHost, m->stream));
            ORBX_HIP(hipStreamSynchronize(m->stream));
            if (chk) {
                rc = xg_recover(m, m->stream, ctl2);
                if (rc != ORBX_OK) return rc;
                if (ctl2[0]) {
                    ORBX_HIP(hipMemcpyAsync(sc.data(), m->d_sc, sizeof(CgScal) * m->nseg, hipMemcpyDeviceToHost, m->stream));
                    ORBX_HIP(hipStreamSynchronize(m->stream));
                }
            }
            bool all = true;
            for (int i = 0; i < m->nseg; ++i) all = all && (sqrt(sc[i].rr) <= tol * sqrt(sc[i].bb) || !(sc[i].rr == sc[i].rr));   // (a mesh whose residual is NaN -- non-finite K: degenerate elements -- will not converge: not waited for)
            if (all) break;
        }
        const int n = iters - done < 25 ? iters - done : 25;
        run_iters(m, n, m->stream);
        done += n;
    }
    ORBX_HIP(hipGetLastError());
    if (iters_done) *iters_done = done;
    return fem_cg_result(m, x, relres);
}

// Development aid (not in include/fem_hip.h): the one-launch CG's plan and its granule buffer as it stands.
int fem_debug_xcd(fem_model *m, int32_t *info8, int32_t *plan4, uint32_t *gran, int max_granules)
{
    if (!m || !info8) return -1;
    const XgLayout L = xg_layout(m->ndof, m->nchunk, m->nchunk_s);
    int v[8] = {m->cg_xcd, m->xg_P, m->xg_ldr, m->xg_ldq, m->xg_lds, L.total, m->nchunk, m->nchunk_s};
    hipStream_t st = m->cg_stream ? m->cg_stream : m->stream;
    if (m->cg_xcd && m->d_xg_ctl) { unsigned a = 0; if (hipMemcpyAsync(&a, &m->d_xg_ctl->abort_flag, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -1; v[4] = (int)a; }   // (slot 4: the abort word)
    memcpy(info8, v, sizeof(v));
    if (!m->cg_xcd) return 0;
    if (plan4 && hipMemcpyAsync(plan4, m->d_xg_plan, sizeof(int4) * m->xg_P, hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    if (gran && m->d_xg_gran && hipMemcpyAsync(gran, m->d_xg_gran, (size_t)16 * std::min(max_granules, L.total), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

int fem_debug_xcd_timing(fem_model *m, uint32_t *out32)
{
    if (!m || !m->d_xg_ctl || !out32) return -1;
    hipStream_t st = m->cg_stream ? m->cg_stream : m->stream;
    if (hipMemcpyAsync(out32, m->d_xg_ctl, sizeof(XgCtl), hipMemcpyDeviceToHost, st) != hipSuccess) return -1;
    return hipStreamSynchronize(st) == hipSuccess ? 0 : -1;
}

int fem_profile_enable(fem_model *m, int on)
{
    if (!m) ORBX_FAIL(ORBX_ERR_ARG, "null model");
    m->prof.reset();
    m->prof.mask = on < 0 ? 0xffffu : (unsigned)on;
    return ORBX_OK;
}

int fem_profile_read(fem_model *m, int max_kinds, const char **names, double *total_ms, int64_t *launches, int *nkinds)
{
    if (!m || !nkinds) ORBX_FAIL(ORBX_ERR_ARG, "null argument");
    m->prof.flush();
    int n = 0;
    for (int i = 0; i < orbx::KernelProfiler::MAXK && n < max_kinds; ++i) {
        if (!m->prof.names[i]) continue;
        if (names) names[n] = m->prof.names[i];
        if (total_ms) total_ms[n] = m->prof.ms[i];
        if (launches) launches[n] = m->prof.launches[i];
        ++n;
    }
    *nkinds = n;
    return ORBX_OK;
}

} // extern "C"
