// gfx950 kernels of the tetrahedral (gdim == 3) instance of the stabilized_schur step: P1/P1 on affine tetrahedra,
// 12 + 4 element dofs, 4x4 vertex blocks (A00 [9], A01 [3], A10 [3], A11 per graph entry).
//
//  * moments3_kernel : M_ab = int_K tau l_a l_b (10 values) and L = int_K tau_LSIC on the 343-point rule
//                      (stabilized_schur.py:100-118; u_prev only -> once per time step)
//  * asm3_kernel     : fused element residual + Jacobian + Dirichlet rows/cols + lifting (stabilized_schur.py:67-123,
//                      144-175,185-189).  One lane per (row vertex, cell) incidence computes the 4x16 row block of its
//                      cell.  Unlike triangles around a vertex, tetrahedra around a vertex form no fan, so the blocks of
//                      a workgroup's rows are accumulated in LDS with ds_add_f64 and written out coalesced -- no global
//                      atomics; the summation order inside LDS is not fixed, so 3-D assembly is reproducible to
//                      round-off, not bitwise.
//  * spmv3 kernels   : 8 lanes per vertex row over the 4x4 block CSR, DPP reductions
// Algebra: oracle/np_twin_nd.py (SURVEY.md Appendix A with d = 3).  vector layout: [u 3*nv | p nv].
#include <hip/hip_runtime.h>

#include <cmath>

#include "cfdh_internal.hpp"
#include "quad_tet.h"

#define TPB 256

__constant__ double d3_qw[CFDH3_NQ];
__constant__ double d3_ql[CFDH3_NQ][4];

int k3_upload_quadrature(cfdh_ctx *c) {
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_qw), CFDH3_QW, sizeof(CFDH3_QW)));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_ql), CFDH3_QL, sizeof(CFDH3_QL)));
  return 0;
}

template <int CTRL>
__device__ __forceinline__ double dpp3(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double g8sum(double v) {
  v += dpp3<0xB1>(v);
  v += dpp3<0x4E>(v);
  v += dpp3<0x141>(v);
  return v;
}
__device__ __forceinline__ double wsum3(double v) {
  v = g8sum(v);
  v += dpp3<0x140>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double bsum3(double v, double *sh) {
  v = wsum3(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// gradients of the barycentrics, volume, greatest vertex distance of a positively oriented tetrahedron
__device__ __forceinline__ void tet_geom(const double X[4][3], double g[4][3], double &vol, double &h) {
  double d[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < 3; i++) d[a][i] = X[a + 1][i] - X[0][i];
  double cr[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int p = (a + 1) % 3, q = (a + 2) % 3;
    cr[a][0] = d[p][1] * d[q][2] - d[p][2] * d[q][1];
    cr[a][1] = d[p][2] * d[q][0] - d[p][0] * d[q][2];
    cr[a][2] = d[p][0] * d[q][1] - d[p][1] * d[q][0];
  }
  const double det = d[0][0] * cr[0][0] + d[0][1] * cr[0][1] + d[0][2] * cr[0][2];
  const double idet = 1.0 / det;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < 3; i++) g[a + 1][i] = cr[a][i] * idet;
#pragma unroll
  for (int i = 0; i < 3; i++) g[0][i] = -(g[1][i] + g[2][i] + g[3][i]);
  vol = fabs(det) * (1.0 / 6.0);
  double h2 = 0.0;
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = a + 1; b < 4; b++) {
      const double e0 = X[a][0] - X[b][0], e1 = X[a][1] - X[b][1], e2 = X[a][2] - X[b][2];
      h2 = fmax(h2, e0 * e0 + e1 * e1 + e2 * e2);
    }
  h = sqrt(h2);
}

// ---------------------------------------------------------------- tau moments
// record per cell (12 doubles): M00 M01 M02 M03 M11 M12 M13 M22 M23 M33 L pad
__global__ __launch_bounds__(TPB) void moments3_kernel(int nc, int nv, const int *__restrict__ cells, const double *__restrict__ coords,
                                                       const double *__restrict__ un, double *__restrict__ mom, double dt, double nu) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= nc) return;
  int vs[4];
  double X[4][3], U[4][3];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    vs[a] = cells[4 * (size_t)e + a];
#pragma unroll
    for (int i = 0; i < 3; i++) { X[a][i] = coords[3 * (size_t)vs[a] + i]; U[a][i] = un[3 * (size_t)vs[a] + i]; }
  }
  double g[4][3], vol, h;
  tet_geom(X, g, vol, h);
  const double ih2 = 1.0 / (h * h);
  const double t2 = 4.0 / (dt * dt), t3 = 16.0 * nu * nu * ih2 * ih2, hr = h / (2.0 * nu);
  double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, L = 0.0;
  for (int q = 0; q < CFDH3_NQ; q++) {
    const double l0 = d3_ql[q][0], l1 = d3_ql[q][1], l2 = d3_ql[q][2], l3 = d3_ql[q][3], wq = d3_qw[q];
    const double ux = l0 * U[0][0] + l1 * U[1][0] + l2 * U[2][0] + l3 * U[3][0];
    const double uy = l0 * U[0][1] + l1 * U[1][1] + l2 * U[2][1] + l3 * U[3][1];
    const double uz = l0 * U[0][2] + l1 * U[1][2] + l2 * U[2][2] + l3 * U[3][2];
    const double s = ux * ux + uy * uy + uz * uz;
    const double t1 = fmax(4.0 * s, 1e-30) * ih2;
    const double tau = 1.0 / sqrt(t1 + t2 + t3);
    const double vn = sqrt(s), Re = vn * hr;
    const double z = (Re <= 3.0) ? Re * (1.0 / 3.0) : 1.0;
    L += wq * vn * h * z * 0.5;
    const double w = wq * tau;
    m[0] += w * l0 * l0; m[1] += w * l0 * l1; m[2] += w * l0 * l2; m[3] += w * l0 * l3;
    m[4] += w * l1 * l1; m[5] += w * l1 * l2; m[6] += w * l1 * l3;
    m[7] += w * l2 * l2; m[8] += w * l2 * l3; m[9] += w * l3 * l3;
  }
  double *o = mom + 12 * (size_t)e;
#pragma unroll
  for (int k = 0; k < 10; k++) o[k] = vol * m[k];
  o[10] = vol * L;
  o[11] = 0.0;
}

int k3_moments(cfdh_ctx *c) {
  prof_begin(c, 2);
  hipLaunchKernelGGL(moments3_kernel, dim3((c->nc + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nc, c->nv, c->cells.p, c->coords.p,
                     c->xprev.p, c->mom.p, c->dt, c->mu / c->rho);
  prof_end(c, 2);
  HIPCHK(c, hipGetLastError());
  c->mom_valid = true;
  return 0;
}

// ---------------------------------------------------------------- fused assembly
struct Asm3Args {
  const double *coords, *mom, *x, *un, *un2, *bcval, *bcmult;
  const int *cells, *vptr, *vdiag, *blk_row, *blk_iptr, *inc_cell, *inc_row;
  const unsigned long long *inc_slots;
  const unsigned char *cflag, *bcflag;
  double *A00, *A01, *A10, *A11, *F;
  int nv;
  double dt, rho, mu, muf, f[3];
  double theta, a0, a1, a2;
  int ds_terms, hist2;
};

// MODE 0: residual only; 1: residual + Jacobian; 2: residual with lifting (Jacobian in registers only)
template <int MODE>
__global__ __launch_bounds__(TPB, 1) void asm3_kernel(Asm3Args p) {
  constexpr bool JAC = (MODE != 0);
  constexpr bool WJ = (MODE == 1);
  extern __shared__ double lds[];  // [nslots][16] value accumulators (MODE 1), then [nrows][4] residual accumulators
  const int blk = blockIdx.x, t = threadIdx.x;
  const int row0 = p.blk_row[blk], row1 = p.blk_row[blk + 1], nrows = row1 - row0;
  const int slot0 = p.vptr[row0], nslots = p.vptr[row1] - slot0;
  double *accJ = lds;
  double *accF = lds + (WJ ? 16 * nslots : 0);
  for (int i = t; i < (WJ ? 16 * nslots : 0) + 4 * nrows; i += TPB) lds[i] = 0.0;
  __syncthreads();
  const int nv = p.nv;
  const double rho = p.rho, mu = p.mu, idt = 1.0 / p.dt, th = p.theta, a0idt = p.a0 * idt;
  for (int inc = p.blk_iptr[blk] + t; inc < p.blk_iptr[blk + 1]; inc += TPB) {
    const int ca = p.inc_cell[inc], e = ca >> 2, a = ca & 3;
    const int rloc = p.inc_row[inc];
    const unsigned long long sl = p.inc_slots[inc];
    int vs[4];
    double X[4][3], ub[4][3], w[4][3], ue[4][3], pe[4];
    unsigned fl[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
      vs[b] = p.cells[4 * (size_t)e + b];
      fl[b] = p.bcflag[vs[b]];
      pe[b] = p.x[3 * (size_t)nv + vs[b]];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        X[b][i] = p.coords[3 * (size_t)vs[b] + i];
        const double u = p.x[3 * (size_t)vs[b] + i], un = p.un[3 * (size_t)vs[b] + i];
        ue[b][i] = u;
        ub[b][i] = th * u + (1.0 - th) * un;
        double wt = p.a0 * u + p.a1 * un;
        if (p.hist2) wt += p.a2 * p.un2[3 * (size_t)vs[b] + i];
        w[b][i] = wt * idt;
      }
    }
    double g[4][3], vol, hh;
    tet_geom(X, g, vol, hh);
    // moments
    const double *mo = p.mom + 12 * (size_t)e;
    double M[4][4];
    M[0][0] = mo[0]; M[0][1] = M[1][0] = mo[1]; M[0][2] = M[2][0] = mo[2]; M[0][3] = M[3][0] = mo[3];
    M[1][1] = mo[4]; M[1][2] = M[2][1] = mo[5]; M[1][3] = M[3][1] = mo[6];
    M[2][2] = mo[7]; M[2][3] = M[3][2] = mo[8]; M[3][3] = mo[9];
    const double Lm = mo[10];
    double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gp[3] = {0, 0, 0};
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        gp[i] += pe[b] * g[b][i];
#pragma unroll
        for (int j = 0; j < 3; j++) G[i][j] += g[b][i] * ub[b][j];
      }
    const double divu = G[0][0] + G[1][1] + G[2][2];
    double Rr[4][3], wc[4][3], beta[4][4], mt[4], Q[4][3];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const double cn = ub[b][0] * G[0][j] + ub[b][1] * G[1][j] + ub[b][2] * G[2][j];
        wc[b][j] = w[b][j] + cn;
        Rr[b][j] = rho * wc[b][j] + gp[j] - rho * p.f[j];
      }
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
      for (int b = 0; b < 4; b++) beta[d][b] = ub[d][0] * g[b][0] + ub[d][1] * g[b][1] + ub[d][2] * g[b][2];
    double T = 0.0;
#pragma unroll
    for (int b = 0; b < 4; b++) { mt[b] = M[b][0] + M[b][1] + M[b][2] + M[b][3]; T += mt[b]; }
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
      for (int i = 0; i < 3; i++) Q[d][i] = M[0][d] * Rr[0][i] + M[1][d] * Rr[1][i] + M[2][d] * Rr[2][i] + M[3][d] * Rr[3][i];
    const double pbar = 0.25 * (pe[0] + pe[1] + pe[2] + pe[3]);
    const double m1 = vol * (1.0 / 20.0);
    // quantities of the row vertex a (runtime index: selected once)
    double ga[3], betaA[4], QA_unused = 0.0;
    (void)QA_unused;
#pragma unroll
    for (int i = 0; i < 3; i++) ga[i] = a == 0 ? g[0][i] : (a == 1 ? g[1][i] : (a == 2 ? g[2][i] : g[3][i]));
#pragma unroll
    for (int d = 0; d < 4; d++) betaA[d] = ub[d][0] * ga[0] + ub[d][1] * ga[1] + ub[d][2] * ga[2];  // beta[d][a]
    const double mtA = a == 0 ? mt[0] : (a == 1 ? mt[1] : (a == 2 ? mt[2] : mt[3]));
    const unsigned flA = a == 0 ? fl[0] : (a == 1 ? fl[1] : (a == 2 ? fl[2] : fl[3]));
    // ---- residual rows of vertex a
    double Fr[4];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double v = 0.0;
#pragma unroll
      for (int b = 0; b < 4; b++) v += rho * m1 * ((b == a) ? 2.0 : 1.0) * wc[b][i];
      v -= rho * p.f[i] * vol * 0.25;
      double Eg = 0.0;
#pragma unroll
      for (int k = 0; k < 3; k++) Eg += 0.5 * (G[i][k] + G[k][i]) * ga[k];
      v += vol * (2.0 * mu * Eg - pbar * ga[i]);
#pragma unroll
      for (int d = 0; d < 4; d++) v += betaA[d] * Q[d][i];
      v += rho * Lm * divu * ga[i];
      Fr[i] = v;
    }
    {
      double v = vol * 0.25 * divu;
#pragma unroll
      for (int b = 0; b < 4; b++) v += mt[b] * (Rr[b][0] * ga[0] + Rr[b][1] * ga[1] + Rr[b][2] * ga[2]) / rho;
      Fr[3] = v;
    }
    // exterior facets containing vertex a (every facet f != a)
    const unsigned cf = p.ds_terms ? (unsigned)p.cflag[e] : 0u;
    double mtBA = 0.0;  // mtB[a] = sum_d mt[d] beta[d][a]
#pragma unroll
    for (int d = 0; d < 4; d++) mtBA += mt[d] * betaA[d];
    double MBa[4];  // MB[c][a] = sum_d M[c][d] beta[d][a]
#pragma unroll
    for (int cI = 0; cI < 4; cI++) MBa[cI] = M[cI][0] * betaA[0] + M[cI][1] * betaA[1] + M[cI][2] * betaA[2] + M[cI][3] * betaA[3];
    // ---- one column block at a time: 16 values, Dirichlet handling, LDS accumulation
#pragma unroll
    for (int b = 0; b < 4; b++) {
      double J00[3][3], J01[3], J10[3], J11 = 0.0;
      if (JAC) {
        const double mab = m1 * ((b == a) ? 2.0 : 1.0);
        double mBab = 0.0, BMB = 0.0, mtBb = 0.0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          mBab += m1 * ((d == a) ? 2.0 : 1.0) * beta[d][b];
          BMB += beta[d][b] * MBa[d];
          mtBb += mt[d] * beta[d][b];
        }
        const double gg = ga[0] * g[b][0] + ga[1] * g[b][1] + ga[2] * g[b][2];
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const double dij = (i == j) ? 1.0 : 0.0;
            double v = rho * mab * dij * a0idt;
            v += rho * th * (mab * G[j][i] + dij * mBab);
            v += vol * mu * th * (g[b][i] * ga[j] + gg * dij);
            v += rho * ((dij * a0idt + th * G[j][i]) * MBa[b] + th * dij * BMB);
            v += th * ga[j] * Q[b][i];
            v += rho * Lm * th * g[b][j] * ga[i];
            J00[i][j] = v;
          }
          J01[i] = -vol * 0.25 * ga[i] + g[b][i] * mtBA;
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double Gg = G[j][0] * ga[0] + G[j][1] * ga[1] + G[j][2] * ga[2];
          J10[j] = vol * 0.25 * th * g[b][j] + mt[b] * (ga[j] * a0idt + th * Gg) + th * ga[j] * mtBb;
        }
        J11 = T * gg / rho;
      }
      // facet terms
      if (cf) {
#pragma unroll
        for (int f = 0; f < 4; f++) {
          if (f == a || !((cf >> f) & 1u)) continue;
          const double gl = sqrt(g[f][0] * g[f][0] + g[f][1] * g[f][1] + g[f][2] * g[f][2]);
          const double n[3] = {-g[f][0] / gl, -g[f][1] / gl, -g[f][2] / gl};
          const double fm = 3.0 * vol * gl;
          if (b == 0) {  // residual part once per incidence
            double pint = 0.0;
#pragma unroll
            for (int q = 0; q < 4; q++) if (q != f) pint += pe[q] * ((q == a) ? 2.0 : 1.0);
            pint *= (1.0 / 12.0);
#pragma unroll
            for (int i = 0; i < 3; i++) {
              const double Gn = G[i][0] * n[0] + G[i][1] * n[1] + G[i][2] * n[2];
              Fr[i] += n[i] * fm * pint - p.muf * Gn * fm * (1.0 / 3.0);
            }
          }
          if (JAC) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
              if (b != f) J01[i] += n[i] * fm * ((b == a) ? 2.0 : 1.0) * (1.0 / 12.0);
#pragma unroll
              for (int j = 0; j < 3; j++) J00[i][j] -= p.muf * th * g[b][i] * n[j] * fm * (1.0 / 3.0);
            }
          }
        }
      }
      // Dirichlet columns of vertex b: lifting F += J[:, bc] (g - x), then zero the column
      if (JAC && fl[b]) {
#pragma unroll
        for (int j = 0; j < 3; j++)
          if ((fl[b] >> j) & 1u) {
            const double gx = p.bcval[4 * (size_t)vs[b] + j] - ue[b][j];
            Fr[0] += J00[0][j] * gx; Fr[1] += J00[1][j] * gx; Fr[2] += J00[2][j] * gx; Fr[3] += J10[j] * gx;
            J00[0][j] = 0.0; J00[1][j] = 0.0; J00[2][j] = 0.0; J10[j] = 0.0;
          }
        if (fl[b] & 8u) {
          const double gx = p.bcval[4 * (size_t)vs[b] + 3] - pe[b];
          Fr[0] += J01[0] * gx; Fr[1] += J01[1] * gx; Fr[2] += J01[2] * gx; Fr[3] += J11 * gx;
          J01[0] = 0.0; J01[1] = 0.0; J01[2] = 0.0; J11 = 0.0;
        }
      }
      if (WJ) {
        // Dirichlet rows of vertex a: zero (the diagonal is set at write-out)
#pragma unroll
        for (int i = 0; i < 3; i++)
          if ((flA >> i) & 1u) { J00[i][0] = 0.0; J00[i][1] = 0.0; J00[i][2] = 0.0; J01[i] = 0.0; }
        if (flA & 8u) { J10[0] = 0.0; J10[1] = 0.0; J10[2] = 0.0; J11 = 0.0; }
        double *dst = accJ + 16 * (size_t)((sl >> (16 * b)) & 0xffffull);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
          for (int j = 0; j < 3; j++) atomicAdd(dst + 3 * i + j, J00[i][j]);
#pragma unroll
        for (int i = 0; i < 3; i++) { atomicAdd(dst + 9 + i, J01[i]); atomicAdd(dst + 12 + i, J10[i]); }
        atomicAdd(dst + 15, J11);
      }
    }
    // Dirichlet rows: residual replaced at write-out
#pragma unroll
    for (int i = 0; i < 3; i++) if ((flA >> i) & 1u) Fr[i] = 0.0;
    if (flA & 8u) Fr[3] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++) atomicAdd(accF + 4 * rloc + i, Fr[i]);
  }
  __syncthreads();
  // ---- Dirichlet rows: diagonal = number of bc objects, F = x - g
  for (int r = t; r < nrows; r += TPB) {
    const int row = row0 + r;
    const unsigned flr = p.bcflag[row];
    if (!flr) continue;
    double *dg = accJ + 16 * (size_t)(p.vdiag[row] - slot0);
#pragma unroll
    for (int i = 0; i < 3; i++)
      if ((flr >> i) & 1u) {
        if (WJ) dg[4 * i] = p.bcmult[4 * (size_t)row + i];
        accF[4 * r + i] = p.x[3 * (size_t)row + i] - p.bcval[4 * (size_t)row + i];
      }
    if (flr & 8u) {
      if (WJ) dg[15] = p.bcmult[4 * (size_t)row + 3];
      accF[4 * r + 3] = p.x[3 * (size_t)nv + row] - p.bcval[4 * (size_t)row + 3];
    }
  }
  __syncthreads();
  // ---- coalesced write-out of the workgroup's slot range
  if (WJ) {
    for (int i = t; i < 9 * nslots; i += TPB) p.A00[9 * (size_t)slot0 + i] = accJ[16 * (i / 9) + (i % 9)];
    for (int i = t; i < 3 * nslots; i += TPB) {
      p.A01[3 * (size_t)slot0 + i] = accJ[16 * (i / 3) + 9 + (i % 3)];
      p.A10[3 * (size_t)slot0 + i] = accJ[16 * (i / 3) + 12 + (i % 3)];
    }
    for (int i = t; i < nslots; i += TPB) p.A11[(size_t)slot0 + i] = accJ[16 * i + 15];
  }
  for (int i = t; i < 3 * nrows; i += TPB) p.F[3 * (size_t)row0 + i] = accF[4 * (i / 3) + (i % 3)];
  for (int i = t; i < nrows; i += TPB) p.F[3 * (size_t)nv + row0 + i] = accF[4 * i + 3];
}

int k3_assemble(cfdh_ctx *c, const double *xstate, int mode) {
  Asm3Args a;
  a.coords = c->coords.p; a.mom = c->mom.p; a.x = xstate; a.un = c->xprev.p; a.un2 = c->xprev2.p; a.bcval = c->bcval.p; a.bcmult = c->bcmult.p;
  a.cells = c->cells.p; a.vptr = c->vptr.p; a.vdiag = c->vdiag.p; a.blk_row = c->a3_blk_row.p; a.blk_iptr = c->a3_blk_iptr.p;
  a.inc_cell = c->a3_inc_cell.p; a.inc_row = c->a3_inc_row.p; a.inc_slots = c->a3_inc_slots.p;
  a.cflag = c->cflag.p; a.bcflag = c->bcflag.p;
  a.A00 = c->A00.p; a.A01 = c->A01.p; a.A10 = c->A10.p; a.A11 = c->A11.p; a.F = c->F.p;
  a.nv = c->nv; a.dt = c->dt; a.rho = c->rho; a.mu = c->mu; a.muf = c->muf; a.f[0] = c->f[0]; a.f[1] = c->f[1]; a.f[2] = c->f[2];
  a.theta = c->ts_theta; a.a0 = c->ts_a[0]; a.a1 = c->ts_a[1]; a.a2 = c->ts_a[2];
  a.ds_terms = c->ds_terms ? 1 : 0; a.hist2 = c->ts_a[2] != 0.0 ? 1 : 0;
  const size_t lds = sizeof(double) * (16 * (size_t)CFDH3_MAX_SLOTS + 4 * 64);
  prof_begin(c, 0);
  if (mode == 1) hipLaunchKernelGGL((asm3_kernel<1>), dim3(c->a3_nblk), dim3(TPB), lds, c->stream, a);
  else if (mode == 2) hipLaunchKernelGGL((asm3_kernel<2>), dim3(c->a3_nblk), dim3(TPB), sizeof(double) * 4 * 64, c->stream, a);
  else hipLaunchKernelGGL((asm3_kernel<0>), dim3(c->a3_nblk), dim3(TPB), sizeof(double) * 4 * 64, c->stream, a);
  prof_end(c, 0);
  HIPCHK(c, hipGetLastError());
  if (mode == 1) c->jac_valid = true;
  return 0;
}

// ---------------------------------------------------------------- block SpMV
// y = J x over the monolithic vector [u 3 nv | p nv]: 8 lanes per vertex row
__global__ __launch_bounds__(TPB) void spmv3_full_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                         const double *__restrict__ A00, const double *__restrict__ A01,
                                                         const double *__restrict__ A10, const double *__restrict__ A11,
                                                         const double *__restrict__ x, double *__restrict__ y) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (row < nv) {
    for (int k = vptr[row] + l, ke = vptr[row + 1]; k < ke; k += 8) {
      const int w = vcol[k];
      const double x0 = x[3 * (size_t)w], x1 = x[3 * (size_t)w + 1], x2 = x[3 * (size_t)w + 2], xp = x[3 * (size_t)nv + w];
      const double *b = A00 + 9 * (size_t)k, *c01 = A01 + 3 * (size_t)k, *c10 = A10 + 3 * (size_t)k;
      a0 += b[0] * x0 + b[1] * x1 + b[2] * x2 + c01[0] * xp;
      a1 += b[3] * x0 + b[4] * x1 + b[5] * x2 + c01[1] * xp;
      a2 += b[6] * x0 + b[7] * x1 + b[8] * x2 + c01[2] * xp;
      a3 += c10[0] * x0 + c10[1] * x1 + c10[2] * x2 + A11[k] * xp;
    }
  }
  a0 = g8sum(a0); a1 = g8sum(a1); a2 = g8sum(a2); a3 = g8sum(a3);
  if (row < nv && l == 0) {
    y[3 * (size_t)row] = a0; y[3 * (size_t)row + 1] = a1; y[3 * (size_t)row + 2] = a2;
    y[3 * (size_t)nv + row] = a3;
  }
}
int k3_spmv_full(cfdh_ctx *c, const double *x, double *y) {
  prof_begin(c, 1);
  hipLaunchKernelGGL(spmv3_full_kernel, dim3((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, c->nvo, c->vptr.p, c->vcol.p,
                     c->A00.p, c->A01.p, c->A10.p, c->A11.p, x, y);
  prof_end(c, 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// BLK 2: y_u = b_u - A01 x_p ; BLK 3: y_p = b_p - A10 x_u  (b null: y = A x)
template <int BLK>
__global__ __launch_bounds__(TPB) void spmv3_blk_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                        const double *__restrict__ A, const double *__restrict__ x,
                                                        double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0, a2 = 0;
  if (row < nv) {
    for (int k = vptr[row] + l, ke = vptr[row + 1]; k < ke; k += 8) {
      const int w = vcol[k];
      const double *cc = A + 3 * (size_t)k;
      if (BLK == 2) {
        const double xp = x[w];
        a0 += cc[0] * xp; a1 += cc[1] * xp; a2 += cc[2] * xp;
      } else {
        a0 += cc[0] * x[3 * (size_t)w] + cc[1] * x[3 * (size_t)w + 1] + cc[2] * x[3 * (size_t)w + 2];
      }
    }
  }
  a0 = g8sum(a0);
  if (BLK == 2) { a1 = g8sum(a1); a2 = g8sum(a2); }
  if (row < nv && l == 0) {
    if (BLK == 2) {
      const size_t o = 3 * (size_t)row;
      y[o] = bvec ? bvec[o] - a0 : a0; y[o + 1] = bvec ? bvec[o + 1] - a1 : a1; y[o + 2] = bvec ? bvec[o + 2] - a2 : a2;
    } else {
      y[row] = bvec ? bvec[row] - a0 : a0;
    }
  }
}
int k3_spmv_block(cfdh_ctx *c, int blk, const double *x, double *y, const double *b) {
  dim3 grid((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), block(TPB);
  if (blk == 2) hipLaunchKernelGGL((spmv3_blk_kernel<2>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A01.p, x, y, b);
  else if (blk == 3) hipLaunchKernelGGL((spmv3_blk_kernel<3>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A10.p, x, y, b);
  else return cfdh_fail(c, CFDH_E_ARG, "k3_spmv_block: block %d not available for tetrahedra", blk);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ||J n|| for the normalised constant-pressure vector (MatNullSpaceTest, stabilized_schur.py:314)
__global__ __launch_bounds__(TPB) void nulltest3_kernel(int nv, const int *__restrict__ vptr, const double *__restrict__ A01,
                                                        const double *__restrict__ A11, double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0;
  for (int row = blockIdx.x * TPB + threadIdx.x; row < nv; row += gridDim.x * TPB) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int k = vptr[row]; k < vptr[row + 1]; k++) { s0 += A01[3 * (size_t)k]; s1 += A01[3 * (size_t)k + 1]; s2 += A01[3 * (size_t)k + 2]; s3 += A11[k]; }
    a += s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3;
  }
  a = bsum3(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
__global__ __launch_bounds__(TPB) void final3_kernel(int nb, int stride, const double *__restrict__ partial, double *__restrict__ out) {
  __shared__ double sh[4];
  double a = 0;
  for (int i = threadIdx.x; i < nb; i += TPB) a += partial[(size_t)blockIdx.x * stride + i];
  a = bsum3(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}
static int read2(cfdh_ctx *c, double *v, int n) {
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->red_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) v[i] = c->h_pinned[i];
  return 0;
}
int k3_nullspace_test(cfdh_ctx *c, double *nrm) {
  const int nb = 256;
  c->mirror_src = nullptr;
  hipLaunchKernelGGL(nulltest3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nvo, c->vptr.p, c->A01.p, c->A11.p, c->red_partial.p);
  hipLaunchKernelGGL(final3_kernel, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
  HIPCHK(c, hipGetLastError());
  double s;
  CHK(read2(c, &s, 1));
  *nrm = sqrt(s);
  return 0;
}

// ---------------------------------------------------------------- functionals
// kind 2/3: ||u||_L2, ||p||_L2 (scenario.py:315-324); kind 7: outward volume flux through the facets of `marker`
__global__ __launch_bounds__(TPB) void l2_3_kernel(int nc, int nv, const int *__restrict__ cells, const double *__restrict__ coords,
                                                   const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double au = 0, ap = 0;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nc; e += gridDim.x * TPB) {
    int vs[4];
    double X[4][3];
    for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
    double g[4][3], vol, h;
    tet_geom(X, g, vol, h);
    double su = 0, sp = 0;
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) {
        const double m = (a == b ? 2.0 : 1.0);
        su += m * (x[3 * (size_t)vs[a]] * x[3 * (size_t)vs[b]] + x[3 * (size_t)vs[a] + 1] * x[3 * (size_t)vs[b] + 1] +
                   x[3 * (size_t)vs[a] + 2] * x[3 * (size_t)vs[b] + 2]);
        sp += m * x[3 * (size_t)nv + vs[a]] * x[3 * (size_t)nv + vs[b]];
      }
    au += vol * su * (1.0 / 20.0);
    ap += vol * sp * (1.0 / 20.0);
  }
  au = bsum3(au, sh);
  ap = bsum3(ap, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = au; partial[gridDim.x + blockIdx.x] = ap; }
}
__global__ __launch_bounds__(TPB) void flux3_kernel(int nfac, int marker, int nv, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                    const int *__restrict__ fmarker, const int *__restrict__ cells,
                                                    const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double q = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker) continue;
    const int e = fcell[k], fl = flocal[k];
    int vs[4];
    double X[4][3];
    for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
    double g[4][3], vol, h;
    tet_geom(X, g, vol, h);
    // |f| n = -3 vol grad l_f ; flux = |f| n . mean of the three facet vertices' velocities
    double um[3] = {0, 0, 0};
    for (int a = 0; a < 4; a++) if (a != fl) for (int i = 0; i < 3; i++) um[i] += x[3 * (size_t)vs[a] + i] * (1.0 / 3.0);
    q += -3.0 * vol * (g[fl][0] * um[0] + g[fl][1] * um[1] + g[fl][2] * um[2]);
  }
  q = bsum3(q, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = q;
}
int k3_functional(cfdh_ctx *c, int kind, int marker, double *out) {
  const int nb = 256;
  c->mirror_src = nullptr;
  if (kind == 2 || kind == 3) {
    hipLaunchKernelGGL(l2_3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nc, c->nv, c->cells.p, c->coords.p, c->x.p, c->red_partial.p);
    hipLaunchKernelGGL(final3_kernel, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
    HIPCHK(c, hipGetLastError());
    double v[2];
    CHK(read2(c, v, 2));
    *out = sqrt(kind == 2 ? v[0] : v[1]);
    return 0;
  }
  if (kind == 7) {
    hipLaunchKernelGGL(flux3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nfac, marker, c->nv, c->d_fac_cell.p, c->d_fac_local.p,
                       c->d_fac_marker.p, c->cells.p, c->coords.p, c->x.p, c->red_partial.p);
    hipLaunchKernelGGL(final3_kernel, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
    HIPCHK(c, hipGetLastError());
    return read2(c, out, 1);
  }
  if (kind >= 4 && kind <= 6) {
    const int nu = 3 * c->nvo;
    const double *a = kind == 5 ? c->xprev.p : c->x.p;
    const double *b = kind == 6 ? c->xprev.p : nullptr;
    return v_norminf_diff(c, nu, a, b, out);
  }
  return cfdh_fail(c, CFDH_E_ARG, "functional kind %d is not available for tetrahedra (2, 3: L2 norms; 4-6: inf-norms; 7: flux)", kind);
}

// wall shear stress (solverBase.py:163-195) on triangles: (1/|f|) oint l_a Tt ds = Tt / 3 for the three facet vertices
__global__ __launch_bounds__(TPB) void wss3_kernel(int nfac, int nv, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                   const int *__restrict__ cells, const double *__restrict__ coords,
                                                   const double *__restrict__ x, double mu, double *__restrict__ out) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nfac) return;
  const int e = fcell[k], fl = flocal[k];
  int vs[4];
  double X[4][3];
  for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
  double g[4][3], vol, h;
  tet_geom(X, g, vol, h);
  const double gl = sqrt(g[fl][0] * g[fl][0] + g[fl][1] * g[fl][1] + g[fl][2] * g[fl][2]);
  const double n[3] = {-g[fl][0] / gl, -g[fl][1] / gl, -g[fl][2] / gl};
  double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int a = 0; a < 4; a++)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) G[i][j] += g[a][i] * x[3 * (size_t)vs[a] + j];
  double T[3];
  for (int i = 0; i < 3; i++) {
    T[i] = 0.0;
    for (int j = 0; j < 3; j++) T[i] -= mu * (G[i][j] + G[j][i]) * n[j];
  }
  const double Tn = T[0] * n[0] + T[1] * n[1] + T[2] * n[2];
  for (int a = 0; a < 4; a++) {
    if (a == fl) continue;
    for (int i = 0; i < 3; i++) atomicAdd(out + 3 * (size_t)vs[a] + i, (T[i] - Tn * n[i]) * (1.0 / 3.0));
  }
}
int k3_wss(cfdh_ctx *c, double *out) {
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(double) * 3 * (size_t)c->nv, c->stream));
  if (c->nfac > 0)
    hipLaunchKernelGGL(wss3_kernel, dim3((c->nfac + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nfac, c->nv, c->d_fac_cell.p, c->d_fac_local.p,
                       c->cells.p, c->coords.p, c->x.p, c->mu, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
