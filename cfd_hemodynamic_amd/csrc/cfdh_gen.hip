// Nodal equal-order elements beyond P1 simplices: P2/P2 triangles
// (`p_grade = 2`, /root/reference/src/solvers/stabilized_schur_backflow.py:84-87) and Q1/Q1 parallelograms
// (/root/reference/src/scenarios/unit_square_pipe.py:101-105, `create_rectangle(..., CellType.quadrilateral)`).
//
// Every node carries (u_x, u_y, p), so the whole linear-algebra side of the library -- node graph with 3x3 blocks in SoA
// arrays, block SpMV, FGMRES, Cahouet-Chabard + AMG preconditioner -- is the P1 code on the NODE graph.  What differs is
// the element integration: basis gradients vary inside the cell (and the strong residual keeps its viscous part
// mu (lap u + grad div u): second derivatives are cell constants under the affine map), so all terms go through the
// quadrature loop -- the same 49-point rule as the tau-moments of the P1 path on triangles, 7 x 7 Gauss on quadrilaterals.
//
// Kernel: one lane per (cell, local test node a).  The nodal data of a cell (coordinates, iterate, u_prev, u_prev2, p,
// Dirichlet flags and lifting values) are staged once in LDS and shared by its nloc lanes; a lane keeps the 3 x 3 nloc row
// block of its test node in registers over the quadrature loop.  Round 4: NO atomics.  The lane stores its blocks (plain
// stores) into a staging array ordered by DESTINATION -- the contributions to one entry of the block-CSR arrays (the cells
// that contain both nodes: 1, 2, or a handful on the diagonal) sit next to each other, in cell order -- and a second kernel,
// one lane per CSR entry (one lane per node for the residual), sums them in that fixed order and writes every entry exactly
// once: bitwise reproducible, no clears of the matrix arrays, and the 25 M fp64 atomics of a Q1 pass (45 % of it) are gone.
// Dirichlet rows / columns and the lifting F += J (g - x) are applied on the element level as DOLFINx does
// (stabilized_schur.py:144-175); a small node kernel then writes the rows x - g and the diagonal multiplicities.
//
// HBM-bound in the limit (324 block entries of 8 B per P2 cell out, ~0.5 KB in), ALU-bound in practice like the tau-moment
// kernels (49 points x ~nloc^2 fused multiply-adds per lane); no MFMA (no dense contraction with a shared operand).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <numeric>

#include "cfdh_internal.hpp"
#include "cfdh_quad_gl.h"
#include "cfdh_quad_tri.h"

#define TPB 256
#define GEN_MAXL 6
#define GEN_NQ 49

namespace {

// reference tables of one element type: weights (times the reference measure), basis values, reference gradients
struct GenTab {
  double w[GEN_NQ];
  double phi[GEN_NQ][GEN_MAXL];
  double dphi[GEN_NQ][GEN_MAXL][2];
};
__constant__ GenTab d_tab[3];  // index: 0 P1 triangle, 1 P2 triangle, 2 Q1 quadrilateral
__constant__ double d_gl2[2][2], d_gl4[2][4];  // facet rules: [0] points, [1] weights

__host__ __device__ inline int gen_nloc(int et) { return et == 0 ? 3 : (et == 1 ? 6 : 4); }

// basis values and reference gradients at a reference point (local order of DOLFINx, see oracle/np_twin_gen.py)
template <int ET>
__host__ __device__ inline void tabulate(double x, double y, double *phi, double (*d)[2]) {
  if (ET == 2) {
    phi[0] = (1 - x) * (1 - y); phi[1] = x * (1 - y); phi[2] = (1 - x) * y; phi[3] = x * y;
    d[0][0] = -(1 - y); d[0][1] = -(1 - x);
    d[1][0] = (1 - y);  d[1][1] = -x;
    d[2][0] = -y;       d[2][1] = (1 - x);
    d[3][0] = y;        d[3][1] = x;
    return;
  }
  const double l[3] = {1.0 - x - y, x, y};
  const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  if (ET == 0) {
    for (int a = 0; a < 3; a++) { phi[a] = l[a]; d[a][0] = dl[a][0]; d[a][1] = dl[a][1]; }
    return;
  }
  const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  for (int a = 0; a < 3; a++) {
    phi[a] = l[a] * (2.0 * l[a] - 1.0);
    for (int k = 0; k < 2; k++) d[a][k] = (4.0 * l[a] - 1.0) * dl[a][k];
  }
  for (int e = 0; e < 3; e++) {
    const int i = ed[e][0], j = ed[e][1];
    phi[3 + e] = 4.0 * l[i] * l[j];
    for (int k = 0; k < 2; k++) d[3 + e][k] = 4.0 * (l[i] * dl[j][k] + l[j] * dl[i][k]);
  }
}
// constant reference Hessian of basis function a: (H00, H01 = H10, H11)
template <int ET>
__host__ __device__ inline void ref_hessian(int a, double H[3]) {
  H[0] = H[1] = H[2] = 0.0;
  if (ET == 1) {
    const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
    const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
    if (a < 3) { H[0] = 4.0 * dl[a][0] * dl[a][0]; H[1] = 4.0 * dl[a][0] * dl[a][1]; H[2] = 4.0 * dl[a][1] * dl[a][1]; }
    else {
      const int i = ed[a - 3][0], j = ed[a - 3][1];
      H[0] = 8.0 * dl[i][0] * dl[j][0];
      H[1] = 4.0 * (dl[i][0] * dl[j][1] + dl[j][0] * dl[i][1]);
      H[2] = 8.0 * dl[i][1] * dl[j][1];
    }
  } else if (ET == 2) {
    H[1] = (a == 0 || a == 3) ? 1.0 : -1.0;
  }
}
template <int ET> __host__ __device__ inline int facet_node(int f, int k) {  // the two end nodes of local facet f
  if (ET == 2) { const int q[4][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 3}}; return q[f][k]; }
  const int t[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  return t[f][k];
}
template <int ET> __host__ __device__ inline void ref_vertex(int v, double r[2]) {
  if (ET == 2) { r[0] = (v & 1) ? 1.0 : 0.0; r[1] = (v & 2) ? 1.0 : 0.0; }
  else { r[0] = v == 1 ? 1.0 : 0.0; r[1] = v == 2 ? 1.0 : 0.0; }
}

template <int ET>
void fill_tab(GenTab &T) {
  if (ET == 2) {
    for (int i = 0; i < 7; i++)
      for (int j = 0; j < 7; j++) {
        const int q = 7 * i + j;
        T.w[q] = CFDH_GL7_W[i] * CFDH_GL7_W[j];
        tabulate<ET>(CFDH_GL7_X[i], CFDH_GL7_X[j], T.phi[q], T.dphi[q]);
      }
  } else {
    for (int q = 0; q < CFDH_NQ; q++) {
      T.w[q] = 0.5 * CFDH_QW[q];
      tabulate<ET>(CFDH_QL[q][1], CFDH_QL[q][2], T.phi[q], T.dphi[q]);
    }
  }
}

struct GenArgs {
  int nc, nvo, mode;  // mode 1: F + J, 2: F only (lifting included)
  const int *cells;
  const double *coords;
  const int *slot;               // [nc][nloc * nloc]: position of the (a, b) block of the cell in the staging array E
  const int *fdst;               // [nc][nloc]: position of the residual rows of local node a in the staging array EF
  double *E, *EF;                // [nc * nloc * nloc][9] (A00 xx xy yx yy | A01 x y | A10 x y | A11), [nc * nloc][3]
  const unsigned short *flag;    // [nc] bit f: exterior facet f, bit 8 + f: backflow facet f
  const double *x, *xprev, *xprev2;
  const unsigned char *bcflag;   // per node: bit0 ux, bit1 uy, bit2 p
  const double *bcval;           // [nv][3]
  double dt, rho, mu, muf, f0, f1, theta, a0, a1, a2, beta;
  int ds_terms;
  double *F, *A00, *A01, *A10, *A11;
};

__device__ __forceinline__ void tau_pair(double s, double h, double dt, double nu, double &tau, double &tauL) {
  double t1 = 4.0 * s;
  t1 = t1 < 1e-30 ? 1e-30 : t1;
  t1 /= h * h;
  tau = cfdh_rsqrt(t1 + 4.0 / (dt * dt) + 16.0 * nu * nu / (h * h * h * h));
  const double vn = sqrt(s), Re = vn * h / (2.0 * nu), z = Re <= 3.0 ? Re * (1.0 / 3.0) : 1.0;
  tauL = vn * h * z * 0.5;
}

// offsets of a node's velocity / pressure in the state layout [u owned 2 nvo | p owned nvo | (u_x, u_y, p) per ghost] (round 4:
// P2 / Q1 contexts take part in partitioned runs; with nvo == nv these are 2 v and 2 nvo + v)
__device__ __forceinline__ size_t guo(int v, int nvo) { return v < nvo ? 2 * (size_t)v : 3 * (size_t)nvo + 3 * (size_t)(v - nvo); }
__device__ __forceinline__ size_t gpo(int v, int nvo) { return v < nvo ? 2 * (size_t)nvo + v : 3 * (size_t)nvo + 3 * (size_t)(v - nvo) + 2; }

// per-cell nodal data in LDS
template <int NL>
struct CellData {
  double X[NL][2], ub[NL][2], wn[NL][2], un[NL][2], p[NL], lift[NL][3];
  unsigned char bc[NL];
  int node[NL];
};

// JAC = false: residual-only pass (trial point of the line search).  The element Jacobian is then formed only in cells whose
// Dirichlet nodes still need lifting (F += J (g - x)) -- none once the first Newton update has put the boundary values in place.
template <int ET, bool JAC>
__global__ __launch_bounds__(TPB) void gen_asm_kernel(GenArgs P) {
  constexpr int NL = ET == 0 ? 3 : (ET == 1 ? 6 : 4);
  constexpr int CPB = TPB / NL;  // cells per workgroup
  constexpr int NV = ET == 2 ? 4 : 3, NF = ET == 2 ? 4 : 3;
  __shared__ CellData<NL> sh[CPB];
  __shared__ double sh_tau[CPB][GEN_NQ][2];  // tau, tau_L at the quadrature points of each cell: computed once, by its lanes in turn
  const int lc = threadIdx.x / NL, a = threadIdx.x % NL;
  const int cell = blockIdx.x * CPB + lc;
  const bool live = lc < CPB && cell < P.nc;
  const int nvo = P.nvo;
  if (live) {
    // lane a stages node a of its cell
    CellData<NL> &D = sh[lc];
    const int v = P.cells[(size_t)cell * NL + a];
    D.node[a] = v;
    const unsigned char bf = P.bcflag[v];
    D.bc[a] = bf;
    for (int i = 0; i < 2; i++) {
      D.X[a][i] = P.coords[2 * (size_t)v + i];
      const double u = P.x[guo(v, nvo) + i], un = P.xprev[guo(v, nvo) + i];
      D.un[a][i] = un;
      D.ub[a][i] = P.theta * u + (1.0 - P.theta) * un;
      D.wn[a][i] = (P.a0 * u + P.a1 * un + (P.a2 != 0.0 ? P.a2 * P.xprev2[guo(v, nvo) + i] : 0.0)) / P.dt;
      D.lift[a][i] = (bf >> i) & 1 ? P.bcval[3 * (size_t)v + i] - u : 0.0;
    }
    const double pv = P.x[gpo(v, nvo)];
    D.p[a] = pv;
    D.lift[a][2] = (bf >> 2) & 1 ? P.bcval[3 * (size_t)v + 2] - pv : 0.0;
  }
  __syncthreads();
  // idle lanes (tail of the last workgroup, 256 mod nloc) follow cell 0 of the workgroup up to the barrier below and write nothing
  const CellData<NL> &D = sh[live ? lc : 0];
  // affine map from the first three vertices
  const double J00 = D.X[1][0] - D.X[0][0], J01 = D.X[2][0] - D.X[0][0], J10 = D.X[1][1] - D.X[0][1], J11 = D.X[2][1] - D.X[0][1];
  const double det = J00 * J11 - J01 * J10, adet = fabs(det), idet = 1.0 / det;
  const double Ji[2][2] = {{J11 * idet, -J01 * idet}, {-J10 * idet, J00 * idet}};  // Ji[k][i] = d xi_k / d x_i
  double h = 0.0;
  for (int q = 0; q < NV; q++)
    for (int r = q + 1; r < NV; r++) h = fmax(h, hypot(D.X[q][0] - D.X[r][0], D.X[q][1] - D.X[r][1]));
  // physical Hessians of all basis functions (cell constants): (xx, xy, yy); viscous part of the strong residual
  double Hs[NL][3], visc[2] = {0.0, 0.0};
#pragma unroll
  for (int b = 0; b < NL; b++) {
    double Hr[3];
    ref_hessian<ET>(b, Hr);
    Hs[b][0] = Hr[0] * Ji[0][0] * Ji[0][0] + 2.0 * Hr[1] * Ji[0][0] * Ji[1][0] + Hr[2] * Ji[1][0] * Ji[1][0];
    Hs[b][1] = Hr[0] * Ji[0][0] * Ji[0][1] + Hr[1] * (Ji[0][0] * Ji[1][1] + Ji[1][0] * Ji[0][1]) + Hr[2] * Ji[1][0] * Ji[1][1];
    Hs[b][2] = Hr[0] * Ji[0][1] * Ji[0][1] + 2.0 * Hr[1] * Ji[0][1] * Ji[1][1] + Hr[2] * Ji[1][1] * Ji[1][1];
    const double lapb = Hs[b][0] + Hs[b][2];
    visc[0] += P.mu * (lapb * D.ub[b][0] + Hs[b][0] * D.ub[b][0] + Hs[b][1] * D.ub[b][1]);
    visc[1] += P.mu * (lapb * D.ub[b][1] + Hs[b][1] * D.ub[b][0] + Hs[b][2] * D.ub[b][1]);
  }
  const double rho = P.rho, mu = P.mu, th = P.theta, a0dt = P.a0 / P.dt, nu = mu / rho;
  // stabilisation parameters (functions of u_prev and h only): lane a of the cell takes the points a, a + NL, ...
  for (int q = a; q < GEN_NQ; q += NL) {
    double u0 = 0.0, u1 = 0.0;
#pragma unroll
    for (int b = 0; b < NL; b++) { u0 += d_tab[ET].phi[q][b] * D.un[b][0]; u1 += d_tab[ET].phi[q][b] * D.un[b][1]; }
    double tq, tlq;
    tau_pair(u0 * u0 + u1 * u1, h, P.dt, nu, tq, tlq);
    if (live) { sh_tau[lc][q][0] = tq; sh_tau[lc][q][1] = tlq; }
  }
  __syncthreads();
  if (!live) return;
  // row block of test node a: residual (u_x, u_y, p) and the 3 x 3 blocks against every node b
  double Fa[3] = {0.0, 0.0, 0.0};
  double Juu[NL][2][2], Jup[NL][2], Jpu[NL][2], Jpp[NL];
#pragma unroll
  for (int b = 0; b < NL; b++) { Juu[b][0][0] = Juu[b][0][1] = Juu[b][1][0] = Juu[b][1][1] = 0.0; Jup[b][0] = Jup[b][1] = Jpu[b][0] = Jpu[b][1] = Jpp[b] = 0.0; }
  bool needj = JAC;
  if (!JAC) {
#pragma unroll
    for (int b = 0; b < NL; b++) needj = needj || D.lift[b][0] != 0.0 || D.lift[b][1] != 0.0 || D.lift[b][2] != 0.0;
  }
  const GenTab &T = d_tab[ET];
  for (int q = 0; q < GEN_NQ; q++) {
    const double dv = adet * T.w[q];
    double g[NL][2], ph[NL];
#pragma unroll
    for (int b = 0; b < NL; b++) {
      ph[b] = T.phi[q][b];
      g[b][0] = T.dphi[q][b][0] * Ji[0][0] + T.dphi[q][b][1] * Ji[1][0];
      g[b][1] = T.dphi[q][b][0] * Ji[0][1] + T.dphi[q][b][1] * Ji[1][1];
    }
    double uq[2] = {0, 0}, wv[2] = {0, 0}, G[2][2] = {{0, 0}, {0, 0}}, gp[2] = {0, 0}, pq = 0.0;
#pragma unroll
    for (int b = 0; b < NL; b++) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        uq[i] += ph[b] * D.ub[b][i]; wv[i] += ph[b] * D.wn[b][i];
        gp[i] += g[b][i] * D.p[b];
        G[i][0] += g[b][i] * D.ub[b][0]; G[i][1] += g[b][i] * D.ub[b][1];
      }
      pq += ph[b] * D.p[b];
    }
    const double divu = G[0][0] + G[1][1];
    const double C[2] = {uq[0] * G[0][0] + uq[1] * G[1][0], uq[0] * G[0][1] + uq[1] * G[1][1]};
    const double R[2] = {rho * (wv[0] + C[0]) - visc[0] + gp[0] - rho * P.f0, rho * (wv[1] + C[1]) - visc[1] + gp[1] - rho * P.f1};
    const double tau = sh_tau[lc][q][0], tauL = sh_tau[lc][q][1];
    const double bga = uq[0] * g[a][0] + uq[1] * g[a][1];  // ubar . grad phi_a
    const double fvec[2] = {P.f0, P.f1};
#pragma unroll
    for (int i = 0; i < 2; i++) {
      double v = rho * ph[a] * (wv[i] + C[i] - fvec[i]) + mu * (g[a][0] * (G[i][0] + G[0][i]) + g[a][1] * (G[i][1] + G[1][i]));
      v += -pq * g[a][i] + tau * R[i] * bga + tauL * rho * divu * g[a][i];
      Fa[i] += dv * v;
    }
    Fa[2] += dv * (ph[a] * divu + tau / rho * (R[0] * g[a][0] + R[1] * g[a][1]));
    if (!needj) continue;
#pragma unroll
    for (int b = 0; b < NL; b++) {
      const double bgb = uq[0] * g[b][0] + uq[1] * g[b][1];
      const double gg = g[a][0] * g[b][0] + g[a][1] * g[b][1];
      const double lapb = Hs[b][0] + Hs[b][2];
#pragma unroll
      for (int j = 0; j < 2; j++) {
        double dR[2], dWC[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const double dij = i == j ? 1.0 : 0.0;
          dWC[i] = rho * (a0dt * ph[b] * dij + th * (ph[b] * G[j][i] + dij * bgb));
          dR[i] = dWC[i] - mu * th * (lapb * dij + Hs[b][i + j]);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const double dij = i == j ? 1.0 : 0.0;
          Juu[b][i][j] += dv * (ph[a] * dWC[i] + mu * th * (g[a][j] * g[b][i] + dij * gg) + tau * dR[i] * bga + th * tau * R[i] * ph[b] * g[a][j] +
                                rho * th * tauL * g[b][j] * g[a][i]);
        }
        Jpu[b][j] += dv * (th * ph[a] * g[b][j] + tau / rho * (dR[0] * g[a][0] + dR[1] * g[a][1]));
      }
      Jup[b][0] += dv * (-ph[b] * g[a][0] + tau * g[b][0] * bga);
      Jup[b][1] += dv * (-ph[b] * g[a][1] + tau * g[b][1] * bga);
      Jpp[b] += dv * tau / rho * gg;
    }
  }
  // exterior-facet terms of this cell that involve test node a
  const unsigned fl = P.flag[cell];
  // (the facet terms enter the velocity rows of node a only: nothing to do when both are Dirichlet rows, i.e. on no-slip walls)
  if (fl && (D.bc[a] & 3u) != 3u) {
    double cen[2] = {0, 0};
    for (int q = 0; q < NV; q++) { cen[0] += D.X[q][0] * (1.0 / NV); cen[1] += D.X[q][1] * (1.0 / NV); }
    constexpr int NQF = ET == 1 ? 4 : 2;
    for (int f = 0; f < NF; f++) {
      const bool ext = P.ds_terms && ((fl >> f) & 1u), bfl = P.beta != 0.0 && ((fl >> (8 + f)) & 1u);
      if (!ext && !bfl) continue;
      const int va = facet_node<ET>(f, 0), vb = facet_node<ET>(f, 1);
      const double tx = D.X[vb][0] - D.X[va][0], ty = D.X[vb][1] - D.X[va][1], elen = hypot(tx, ty);
      double n[2] = {ty / elen, -tx / elen};
      if ((0.5 * (D.X[va][0] + D.X[vb][0]) - cen[0]) * n[0] + (0.5 * (D.X[va][1] + D.X[vb][1]) - cen[1]) * n[1] < 0) { n[0] = -n[0]; n[1] = -n[1]; }
      double ra[2], rb[2];
      ref_vertex<ET>(va, ra); ref_vertex<ET>(vb, rb);
      for (int q = 0; q < NQF; q++) {
        const double t = ET == 1 ? d_gl4[0][q] : d_gl2[0][q], m = elen * (ET == 1 ? d_gl4[1][q] : d_gl2[1][q]);
        double ph[GEN_MAXL], dr[GEN_MAXL][2], g[NL][2];
        tabulate<ET>((1 - t) * ra[0] + t * rb[0], (1 - t) * ra[1] + t * rb[1], ph, dr);
        if (ph[a] == 0.0) continue;  // test function vanishes on this facet
#pragma unroll
        for (int b = 0; b < NL; b++) { g[b][0] = dr[b][0] * Ji[0][0] + dr[b][1] * Ji[1][0]; g[b][1] = dr[b][0] * Ji[0][1] + dr[b][1] * Ji[1][1]; }
        double uq[2] = {0, 0}, G[2][2] = {{0, 0}, {0, 0}}, pq = 0.0, sn = 0.0;
#pragma unroll
        for (int b = 0; b < NL; b++) {
#pragma unroll
          for (int i = 0; i < 2; i++) {
            uq[i] += ph[b] * D.ub[b][i];
            sn += ph[b] * D.un[b][i] * n[i];
            G[i][0] += g[b][i] * D.ub[b][0]; G[i][1] += g[b][i] * D.ub[b][1];
          }
          pq += ph[b] * D.p[b];
        }
        if (ext) {
#pragma unroll
          for (int i = 0; i < 2; i++) {
            Fa[i] += m * ph[a] * (pq * n[i] - P.muf * (G[i][0] * n[0] + G[i][1] * n[1]));
#pragma unroll
            for (int b = 0; b < NL; b++) {
              Jup[b][i] += m * ph[a] * ph[b] * n[i];
              Juu[b][i][0] -= P.muf * th * m * ph[a] * g[b][i] * n[0];
              Juu[b][i][1] -= P.muf * th * m * ph[a] * g[b][i] * n[1];
            }
          }
        }
        if (bfl) {
          const double cq = P.beta * rho * 0.5 * (sn - fabs(sn)) * m;
#pragma unroll
          for (int i = 0; i < 2; i++) {
            Fa[i] -= cq * ph[a] * uq[i];
#pragma unroll
            for (int b = 0; b < NL; b++) Juu[b][i][i] -= th * cq * ph[a] * ph[b];
          }
        }
      }
    }
  }
  // Dirichlet handling on the element level (assemble_vector_block(..., x0 = x, alpha = -1), stabilized_schur.py:172-174):
  // lifting with the FULL element row, then constrained rows and columns dropped
  const unsigned bca = D.bc[a];
#pragma unroll
  for (int b = 0; b < NL; b++) {
    const double l0 = D.lift[b][0], l1 = D.lift[b][1], l2 = D.lift[b][2];
    Fa[0] += Juu[b][0][0] * l0 + Juu[b][0][1] * l1 + Jup[b][0] * l2;
    Fa[1] += Juu[b][1][0] * l0 + Juu[b][1][1] * l1 + Jup[b][1] * l2;
    Fa[2] += Jpu[b][0] * l0 + Jpu[b][1] * l1 + Jpp[b] * l2;
  }
  // constrained rows / columns contribute zeros (their rows are written by gen_bc_rows_kernel)
  const int fd = P.fdst[(size_t)cell * NL + a];
  if (fd < 0) return;  // row of a ghost node: assembled by its owner (one-cell overlap of the partition)
  {
    double *ef = P.EF + 3 * (size_t)fd;
    ef[0] = (bca & 1u) ? 0.0 : Fa[0];
    ef[1] = (bca & 2u) ? 0.0 : Fa[1];
    ef[2] = (bca & 4u) ? 0.0 : Fa[2];
  }
  if (P.mode != 1) return;
  const int *sl = P.slot + ((size_t)cell * NL + a) * NL;
#pragma unroll
  for (int b = 0; b < NL; b++) {
    const unsigned bcb = D.bc[b];
    double *eb = P.E + 9 * (size_t)sl[b];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bool ri = (bca >> i) & 1u;
#pragma unroll
      for (int j = 0; j < 2; j++) eb[2 * i + j] = (ri || ((bcb >> j) & 1u)) ? 0.0 : Juu[b][i][j];
      eb[4 + i] = (ri || (bcb & 4u)) ? 0.0 : Jup[b][i];
    }
    const bool rp = bca & 4u;
#pragma unroll
    for (int j = 0; j < 2; j++) eb[6 + j] = (rp || ((bcb >> j) & 1u)) ? 0.0 : Jpu[b][j];
    eb[8] = (rp || (bcb & 4u)) ? 0.0 : Jpp[b];
  }
}

// second phase: fixed-order sums of the staged contributions, every output written once
__global__ __launch_bounds__(TPB) void gen_gather_F_kernel(int nvo, const int *__restrict__ fptr, const double *__restrict__ EF, double *__restrict__ F) {
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= nvo) return;
  double f0 = 0.0, f1 = 0.0, f2 = 0.0;
  for (int k = fptr[v], ke = fptr[v + 1]; k < ke; k++) { f0 += EF[3 * (size_t)k]; f1 += EF[3 * (size_t)k + 1]; f2 += EF[3 * (size_t)k + 2]; }
  F[2 * (size_t)v] = f0; F[2 * (size_t)v + 1] = f1; F[2 * (size_t)nvo + v] = f2;
}
__global__ __launch_bounds__(TPB) void gen_gather_J_kernel(int nnz, const int *__restrict__ eptr, const double *__restrict__ E, double *__restrict__ A00,
                                                           double *__restrict__ A01, double *__restrict__ A10, double *__restrict__ A11) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nnz) return;
  double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = eptr[k], qe = eptr[k + 1]; q < qe; q++) {
    const double *e = E + 9 * (size_t)q;
#pragma unroll
    for (int t = 0; t < 9; t++) a[t] += e[t];
  }
  *(double2 *)(A00 + 4 * (size_t)k) = make_double2(a[0], a[1]);
  *(double2 *)(A00 + 4 * (size_t)k + 2) = make_double2(a[2], a[3]);
  *(double2 *)(A01 + 2 * (size_t)k) = make_double2(a[4], a[5]);
  *(double2 *)(A10 + 2 * (size_t)k) = make_double2(a[6], a[7]);
  A11[k] = a[8];
}

// rows of constrained dofs: F = x - g; diagonal = number of Dirichlet objects holding the dof (stabilized_schur.py:144-175)
__global__ __launch_bounds__(TPB) void gen_bc_rows_kernel(int nvo, int mode, const unsigned char *__restrict__ bcflag, const double *__restrict__ bcval,
                                                          const double *__restrict__ bcmult, const int *__restrict__ vdiag, const double *__restrict__ x,
                                                          double *__restrict__ F, double *__restrict__ A00, double *__restrict__ A11) {
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= nvo) return;
  const unsigned bf = bcflag[v];
  if (!bf) return;
  const size_t k = (size_t)vdiag[v];
  for (int i = 0; i < 2; i++)
    if ((bf >> i) & 1u) {
      F[2 * (size_t)v + i] = x[2 * (size_t)v + i] - bcval[3 * (size_t)v + i];
      if (mode == 1) A00[4 * k + 3 * i] += bcmult[3 * (size_t)v + i];
    }
  if (bf & 4u) {
    F[2 * (size_t)nvo + v] = x[2 * (size_t)nvo + v] - bcval[3 * (size_t)v + 2];
    if (mode == 1) A11[k] += bcmult[3 * (size_t)v + 2];
  }
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ double block_sum_d(double v, double *sh) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// int u.u and int p^2 with the element's own mass matrix (scenario.py:315-324)
template <int ET>
__global__ __launch_bounds__(TPB) void gen_l2_kernel(int nc, int nvo, const int *__restrict__ cells, const unsigned char *__restrict__ cell_owned,
                                                     const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  constexpr int NL = ET == 0 ? 3 : (ET == 1 ? 6 : 4);
  __shared__ double sh[4];
  double au = 0, ap = 0;
  const GenTab &T = d_tab[ET];
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nc; e += gridDim.x * TPB) {
    if (!cell_owned[e]) continue;  // every cell is integrated by exactly one rank
    double X[3][2], u[NL][2], p[NL];
    for (int a = 0; a < NL; a++) {
      const int v = cells[(size_t)e * NL + a];
      if (a < 3) { X[a][0] = coords[2 * (size_t)v]; X[a][1] = coords[2 * (size_t)v + 1]; }
      u[a][0] = x[guo(v, nvo)]; u[a][1] = x[guo(v, nvo) + 1]; p[a] = x[gpo(v, nvo)];
    }
    const double adet = fabs((X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]));
    for (int q = 0; q < GEN_NQ; q++) {
      double u0 = 0, u1 = 0, pq = 0;
      for (int a = 0; a < NL; a++) { u0 += T.phi[q][a] * u[a][0]; u1 += T.phi[q][a] * u[a][1]; pq += T.phi[q][a] * p[a]; }
      au += adet * T.w[q] * (u0 * u0 + u1 * u1);
      ap += adet * T.w[q] * pq * pq;
    }
  }
  au = block_sum_d(au, sh);
  ap = block_sum_d(ap, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = au; partial[gridDim.x + blockIdx.x] = ap; }
}

// kind 7: flux int u.n over the facets with the given marker (outward normal); kinds 0 / 1: drag / lift of dfg_1.py:183-202
template <int ET>
__global__ __launch_bounds__(TPB) void gen_facet_functional_kernel(int nfac, int marker, int kind, int nvo, const int *__restrict__ fcell,
                                                                   const int *__restrict__ flocal, const int *__restrict__ fmarker,
                                                                   const int *__restrict__ cells, const unsigned char *__restrict__ cell_owned,
                                                                   const double *__restrict__ coords, const double *__restrict__ x, double mu,
                                                                   double *__restrict__ partial) {
  constexpr int NL = ET == 0 ? 3 : (ET == 1 ? 6 : 4);
  constexpr int NV = ET == 2 ? 4 : 3, NQF = ET == 1 ? 4 : 2;
  __shared__ double sh[4];
  double a0 = 0, a1 = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker || !cell_owned[fcell[k]]) continue;
    const int e = fcell[k], f = flocal[k];
    double X[NL][2], u[NL][2], p[NL];
    for (int a = 0; a < NL; a++) {
      const int v = cells[(size_t)e * NL + a];
      X[a][0] = coords[2 * (size_t)v]; X[a][1] = coords[2 * (size_t)v + 1];
      u[a][0] = x[guo(v, nvo)]; u[a][1] = x[guo(v, nvo) + 1]; p[a] = x[gpo(v, nvo)];
    }
    const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0], J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
    const double idet = 1.0 / (J00 * J11 - J01 * J10);
    const double Ji[2][2] = {{J11 * idet, -J01 * idet}, {-J10 * idet, J00 * idet}};
    double cen[2] = {0, 0};
    for (int q = 0; q < NV; q++) { cen[0] += X[q][0] * (1.0 / NV); cen[1] += X[q][1] * (1.0 / NV); }
    const int va = facet_node<ET>(f, 0), vb = facet_node<ET>(f, 1);
    const double tx = X[vb][0] - X[va][0], ty = X[vb][1] - X[va][1], elen = hypot(tx, ty);
    double n[2] = {ty / elen, -tx / elen};
    if ((0.5 * (X[va][0] + X[vb][0]) - cen[0]) * n[0] + (0.5 * (X[va][1] + X[vb][1]) - cen[1]) * n[1] < 0) { n[0] = -n[0]; n[1] = -n[1]; }
    double ra[2], rb[2];
    ref_vertex<ET>(va, ra); ref_vertex<ET>(vb, rb);
    for (int q = 0; q < NQF; q++) {
      const double t = ET == 1 ? d_gl4[0][q] : d_gl2[0][q], m = elen * (ET == 1 ? d_gl4[1][q] : d_gl2[1][q]);
      double ph[GEN_MAXL], dr[GEN_MAXL][2];
      tabulate<ET>((1 - t) * ra[0] + t * rb[0], (1 - t) * ra[1] + t * rb[1], ph, dr);
      double uq[2] = {0, 0}, pq = 0, gut[2] = {0, 0};
      // drag / lift are written with n = -FacetNormal, t = (n_y, -n_x), u_t = t . u
      const double nn[2] = {-n[0], -n[1]}, tt[2] = {nn[1], -nn[0]};
      for (int a = 0; a < NL; a++) {
        uq[0] += ph[a] * u[a][0]; uq[1] += ph[a] * u[a][1]; pq += ph[a] * p[a];
        const double ut = u[a][0] * tt[0] + u[a][1] * tt[1];
        gut[0] += ut * (dr[a][0] * Ji[0][0] + dr[a][1] * Ji[1][0]);
        gut[1] += ut * (dr[a][0] * Ji[0][1] + dr[a][1] * Ji[1][1]);
      }
      if (kind == 7) a0 += m * (uq[0] * n[0] + uq[1] * n[1]);
      else {
        const double dn = gut[0] * nn[0] + gut[1] * nn[1];
        a0 += m * (mu * dn * nn[1] - pq * nn[0]);
        a1 -= m * (mu * dn * nn[0] + pq * nn[1]);
      }
    }
  }
  a0 = block_sum_d(a0, sh);
  a1 = block_sum_d(a1, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = a0; partial[gridDim.x + blockIdx.x] = a1; }
}

// wall shear stress (solverBase.py:163-195): (1/|e|) oint w . (T - (T.n) n), T = -sigma(u,p) n, per facet node
template <int ET>
__global__ __launch_bounds__(TPB) void gen_wss_kernel(int nfac, int nvo, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                      const int *__restrict__ cells, const double *__restrict__ coords, const double *__restrict__ x,
                                                      double mu, double *__restrict__ out) {
  constexpr int NL = ET == 0 ? 3 : (ET == 1 ? 6 : 4);
  constexpr int NV = ET == 2 ? 4 : 3, NQF = ET == 1 ? 4 : 2;
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nfac) return;
  const int e = fcell[k], f = flocal[k];
  int vs[NL];
  double X[NL][2], u[NL][2];
  for (int a = 0; a < NL; a++) {
    vs[a] = cells[(size_t)e * NL + a];
    X[a][0] = coords[2 * (size_t)vs[a]]; X[a][1] = coords[2 * (size_t)vs[a] + 1];
    u[a][0] = x[guo(vs[a], nvo)]; u[a][1] = x[guo(vs[a], nvo) + 1];
  }
  const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0], J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
  const double idet = 1.0 / (J00 * J11 - J01 * J10);
  const double Ji[2][2] = {{J11 * idet, -J01 * idet}, {-J10 * idet, J00 * idet}};
  double cen[2] = {0, 0};
  for (int q = 0; q < NV; q++) { cen[0] += X[q][0] * (1.0 / NV); cen[1] += X[q][1] * (1.0 / NV); }
  const int va = facet_node<ET>(f, 0), vb = facet_node<ET>(f, 1);
  const double tx = X[vb][0] - X[va][0], ty = X[vb][1] - X[va][1], elen = hypot(tx, ty);
  double n[2] = {ty / elen, -tx / elen};
  if ((0.5 * (X[va][0] + X[vb][0]) - cen[0]) * n[0] + (0.5 * (X[va][1] + X[vb][1]) - cen[1]) * n[1] < 0) { n[0] = -n[0]; n[1] = -n[1]; }
  double ra[2], rb[2];
  ref_vertex<ET>(va, ra); ref_vertex<ET>(vb, rb);
  double acc[GEN_MAXL][2];
  for (int a = 0; a < NL; a++) acc[a][0] = acc[a][1] = 0.0;
  for (int q = 0; q < NQF; q++) {
    const double t = ET == 1 ? d_gl4[0][q] : d_gl2[0][q], wq = ET == 1 ? d_gl4[1][q] : d_gl2[1][q];  // (1/|e|) |e| w_q
    double ph[GEN_MAXL], dr[GEN_MAXL][2];
    tabulate<ET>((1 - t) * ra[0] + t * rb[0], (1 - t) * ra[1] + t * rb[1], ph, dr);
    double G[2][2] = {{0, 0}, {0, 0}};
    for (int a = 0; a < NL; a++)
      for (int i = 0; i < 2; i++) {
        const double gi = dr[a][0] * Ji[0][i] + dr[a][1] * Ji[1][i];
        G[i][0] += gi * u[a][0]; G[i][1] += gi * u[a][1];
      }
    const double E01 = 0.5 * (G[0][1] + G[1][0]);
    const double Tv[2] = {-2.0 * mu * (G[0][0] * n[0] + E01 * n[1]), -2.0 * mu * (E01 * n[0] + G[1][1] * n[1])};
    const double Tn = Tv[0] * n[0] + Tv[1] * n[1];
    for (int a = 0; a < NL; a++) { acc[a][0] += wq * ph[a] * (Tv[0] - Tn * n[0]); acc[a][1] += wq * ph[a] * (Tv[1] - Tn * n[1]); }
  }
  for (int a = 0; a < NL; a++)
    if (acc[a][0] != 0.0 || acc[a][1] != 0.0) { atomicAdd(out + 2 * (size_t)vs[a], acc[a][0]); atomicAdd(out + 2 * (size_t)vs[a] + 1, acc[a][1]); }
}

inline uint32_t part1by1(uint32_t x) {
  x &= 0x0000ffff;
  x = (x ^ (x << 8)) & 0x00ff00ff;
  x = (x ^ (x << 4)) & 0x0f0f0f0f;
  x = (x ^ (x << 2)) & 0x33333333;
  x = (x ^ (x << 1)) & 0x55555555;
  return x;
}

}  // namespace

int kg_upload_tables(cfdh_ctx *c) {
  static GenTab tab[3];
  fill_tab<0>(tab[0]); fill_tab<1>(tab[1]); fill_tab<2>(tab[2]);
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d_tab), tab, sizeof tab));
  double g2[2][2], g4[2][4];
  for (int q = 0; q < 2; q++) { g2[0][q] = CFDH_GL2_X[q]; g2[1][q] = CFDH_GL2_W[q]; }
  for (int q = 0; q < 4; q++) { g4[0][q] = CFDH_GL4_X[q]; g4[1][q] = CFDH_GL4_W[q]; }
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d_gl2), g2, sizeof g2));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d_gl4), g4, sizeof g4));
  return 0;
}

// nodes of local facet f of a cell (host): the end nodes, for P2 also the edge node
int cfdh_facet_nodes(const cfdh_ctx *c, int f, int out[3]) {
  if (c->etype == 2) { const int q[4][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 3}}; out[0] = q[f][0]; out[1] = q[f][1]; return 2; }
  const int t[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  out[0] = t[f][0]; out[1] = t[f][1];
  if (c->etype == 1) { out[2] = 3 + f; return 3; }
  return 2;
}

// Mesh upload for the generic element path: Morton numbering of the nodes, node graph, value slots of every local node pair,
// stiffness / diagonal mass of the element on the graph (Cahouet-Chabard preconditioner), state and work vectors.
int cfdh_build_mesh_gen(cfdh_ctx *c, int etype, int64_t nv64, int64_t nvo64, int64_t nc64, const int32_t *cells, const double *coords, int64_t nfac64,
                        const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker) {
  const int nv = (int)nv64, nvo = (int)nvo64, nc = (int)nc64, nfac = (int)nfac64;
  if (nvo <= 0 || nvo > nv) return cfdh_fail(c, CFDH_E_ARG, "bad owned node count");
  const int et = etype == 3 ? 0 : etype;  // 3: P1 triangles through the generic kernels (cross-check of the closed-form path)
  const int NL = gen_nloc(et), NF = et == 2 ? 4 : 3;
  if (nv <= 0 || nc <= 0) return cfdh_fail(c, CFDH_E_ARG, "bad mesh sizes");
  if (nv64 > (1ll << 29) || nc64 > (1ll << 27)) return cfdh_fail(c, CFDH_E_ARG, "mesh too large for int32 indexing");
  for (int64_t k = 0; k < (int64_t)NL * nc; k++)
    if (cells[k] < 0 || cells[k] >= nv) return cfdh_fail(c, CFDH_E_ARG, "cell node index out of range");
  for (int k = 0; k < nfac; k++)
    if (fcell[k] < 0 || fcell[k] >= nc || flocal[k] < 0 || flocal[k] >= NF) return cfdh_fail(c, CFDH_E_ARG, "facet (cell, local) out of range");
  c->etype = et; c->nloc = NL; c->gen = true;
  // partitioned runs (round 4): nodes [0, nvo) are owned, the rest are the ghost nodes of the one-cell overlap in the order of the
  // halo plan; rows are assembled for owned nodes only; vectors carry the ghost tail [(u_x, u_y, p) per ghost]
  c->nv = nv; c->nvo = nvo; c->ng = nv - nvo;
  c->NO = 3 * nvo; c->NL = 3 * nvo + 3 * c->ng;
  // ---- node numbering: owned nodes along a Morton curve, ghosts unchanged
  c->perm.resize(nv); c->iperm.resize(nv);
  {
    std::vector<int> order(nvo);
    std::iota(order.begin(), order.end(), 0);
    double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
    for (int v = 0; v < nv; v++)
      for (int i = 0; i < 2; i++) { lo[i] = std::min(lo[i], coords[2 * v + i]); hi[i] = std::max(hi[i], coords[2 * v + i]); }
    const double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
    if (!(ext > 0)) return cfdh_fail(c, CFDH_E_ARG, "degenerate coordinates");
    std::vector<uint32_t> key(nvo);
    for (int v = 0; v < nvo; v++) {
      const uint32_t qx = (uint32_t)std::min(65535.0, (coords[2 * v] - lo[0]) / ext * 65535.0), qy = (uint32_t)std::min(65535.0, (coords[2 * v + 1] - lo[1]) / ext * 65535.0);
      key[v] = part1by1(qx) | (part1by1(qy) << 1);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int k = 0; k < nvo; k++) { c->iperm[k] = order[k]; c->perm[order[k]] = k; }
    for (int v = nvo; v < nv; v++) { c->iperm[v] = v; c->perm[v] = v; }
  }
  c->h_coords.resize(2 * (size_t)nv);
  for (int k = 0; k < nv; k++) { c->h_coords[2 * k] = coords[2 * c->iperm[k]]; c->h_coords[2 * k + 1] = coords[2 * c->iperm[k] + 1]; }
  // ---- cells (user order kept: no fan structure to build), facets
  c->nc = nc;
  c->h_cells.resize((size_t)NL * nc);
  c->cell_user.resize(nc);
  for (int e = 0; e < nc; e++) {
    c->cell_user[e] = e;
    for (int a = 0; a < NL; a++) c->h_cells[(size_t)NL * e + a] = c->perm[cells[(size_t)NL * e + a]];
    const double *X = c->h_coords.data();
    const int *v = &c->h_cells[(size_t)NL * e];
    const double det = (X[2 * v[1]] - X[2 * v[0]]) * (X[2 * v[2] + 1] - X[2 * v[0] + 1]) - (X[2 * v[1] + 1] - X[2 * v[0] + 1]) * (X[2 * v[2]] - X[2 * v[0]]);
    if (!(std::fabs(det) > 0)) return cfdh_fail(c, CFDH_E_ARG, "zero-area cell %d", e);
    if (et == 2) {
      // Q1 cells must be parallelograms (affine map): x3 = x1 + x2 - x0
      const double ex = X[2 * v[3]] - (X[2 * v[1]] + X[2 * v[2]] - X[2 * v[0]]), ey = X[2 * v[3] + 1] - (X[2 * v[1] + 1] + X[2 * v[2] + 1] - X[2 * v[0] + 1]);
      if (std::hypot(ex, ey) > 1e-9 * std::sqrt(std::fabs(det))) return cfdh_fail(c, CFDH_E_ARG, "quadrilateral %d is not a parallelogram: only affine Q1 cells are supported", e);
    }
    if (et == 1) {
      // P2 on a straight-sided triangulation: edge nodes at the edge midpoints
      const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
      for (int q = 0; q < 3; q++) {
        const double mx = 0.5 * (X[2 * v[ed[q][0]]] + X[2 * v[ed[q][1]]]) - X[2 * v[3 + q]], my = 0.5 * (X[2 * v[ed[q][0]] + 1] + X[2 * v[ed[q][1]] + 1]) - X[2 * v[3 + q] + 1];
        if (std::hypot(mx, my) > 1e-9 * std::sqrt(std::fabs(det))) return cfdh_fail(c, CFDH_E_ARG, "P2 cell %d: edge node %d is not the edge midpoint (curved cells are not supported)", e, q);
      }
    }
  }
  c->fac_cell.assign(fcell, fcell + nfac); c->fac_local.assign(flocal, flocal + nfac);
  c->fac_marker.resize(nfac); c->fac_user.resize(nfac);
  for (int k = 0; k < nfac; k++) { c->fac_marker[k] = fmarker ? fmarker[k] : 0; c->fac_user[k] = k; }
  c->nfac = c->nfac_user = nfac;
  // ---- node graph
  std::vector<int> ncptr(nv + 1, 0);
  for (size_t k = 0; k < c->h_cells.size(); k++) ncptr[c->h_cells[k] + 1]++;
  for (int v = 0; v < nv; v++) ncptr[v + 1] += ncptr[v];
  std::vector<int> ncell(ncptr[nv]);
  {
    std::vector<int> fill(nv, 0);
    for (int e = 0; e < nc; e++)
      for (int a = 0; a < NL; a++) { const int v = c->h_cells[(size_t)NL * e + a]; ncell[ncptr[v] + fill[v]++] = e; }
  }
  c->h_vptr.assign(nvo + 1, 0);
  c->h_vcol.clear(); c->h_vcol.reserve((size_t)14 * nvo);
  c->h_vdiag.resize(nvo);
  {
    std::vector<int> tmp;
    for (int v = 0; v < nvo; v++) {
      if (ncptr[v + 1] == ncptr[v]) return cfdh_fail(c, CFDH_E_ARG, "node %d belongs to no cell", c->iperm[v]);
      tmp.clear();
      for (int k = ncptr[v]; k < ncptr[v + 1]; k++)
        for (int a = 0; a < NL; a++) tmp.push_back(c->h_cells[(size_t)NL * ncell[k] + a]);
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      c->h_vdiag[v] = (int)c->h_vcol.size() + (int)(std::lower_bound(tmp.begin(), tmp.end(), v) - tmp.begin());
      c->h_vcol.insert(c->h_vcol.end(), tmp.begin(), tmp.end());
      c->h_vptr[v + 1] = (int)c->h_vcol.size();
    }
  }
  c->nnzv = (int)c->h_vcol.size();
  c->ninc = (int)c->h_cells.size();
  // ---- slots, stiffness and diagonal mass on the graph
  std::vector<int> slot((size_t)nc * NL * NL);
  c->h_Lval.assign(c->nnzv, 0.0);
  c->h_Ml.assign(nv, 0.0);
  GenTab T;
  if (et == 0) fill_tab<0>(T); else if (et == 1) fill_tab<1>(T); else fill_tab<2>(T);
  double msum = 0.0, dsum = 0.0;
  std::vector<double> mdiag(nv, 0.0);
  for (int e = 0; e < nc; e++) {
    const int *v = &c->h_cells[(size_t)NL * e];
    const double *X = c->h_coords.data();
    const double J00 = X[2 * v[1]] - X[2 * v[0]], J01 = X[2 * v[2]] - X[2 * v[0]], J10 = X[2 * v[1] + 1] - X[2 * v[0] + 1], J11 = X[2 * v[2] + 1] - X[2 * v[0] + 1];
    const double det = J00 * J11 - J01 * J10, adet = std::fabs(det);
    const double Ji[2][2] = {{J11 / det, -J01 / det}, {-J10 / det, J00 / det}};
    double K[GEN_MAXL][GEN_MAXL] = {{0}}, Md[GEN_MAXL] = {0};
    for (int q = 0; q < GEN_NQ; q++) {
      double g[GEN_MAXL][2];
      for (int a = 0; a < NL; a++)
        for (int i = 0; i < 2; i++) g[a][i] = T.dphi[q][a][0] * Ji[0][i] + T.dphi[q][a][1] * Ji[1][i];
      for (int a = 0; a < NL; a++) {
        Md[a] += adet * T.w[q] * T.phi[q][a] * T.phi[q][a];
        for (int b = 0; b < NL; b++) K[a][b] += adet * T.w[q] * (g[a][0] * g[b][0] + g[a][1] * g[b][1]);
      }
    }
    msum += adet * (et == 2 ? 1.0 : 0.5);
    for (int a = 0; a < NL; a++) {
      mdiag[v[a]] += Md[a];
      dsum += Md[a];
      if (v[a] >= nvo) {  // row of a ghost node: assembled by its owner
        for (int b = 0; b < NL; b++) slot[((size_t)e * NL + a) * NL + b] = -1;
        continue;
      }
      const int *nb = &c->h_vcol[c->h_vptr[v[a]]];
      const int deg = c->h_vptr[v[a] + 1] - c->h_vptr[v[a]];
      for (int b = 0; b < NL; b++) {
        const int k = c->h_vptr[v[a]] + (int)(std::lower_bound(nb, nb + deg, v[b]) - nb);
        slot[((size_t)e * NL + a) * NL + b] = k;
        c->h_Lval[k] += K[a][b];
      }
    }
  }
  if (et == 1) {
    // P1 subspace of the P2 space: vertex nodes (local positions 0..2) numbered in order of first appearance along the node
    // numbering; an edge node interpolates its two end vertices
    std::vector<int> vid(nv, -1), ea(nv, -1), eb(nv, -1);
    for (int e = 0; e < nc; e++) {
      const int *v = &c->h_cells[(size_t)NL * e];
      const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
      for (int q = 0; q < 3; q++) { vid[v[q]] = 0; ea[v[3 + q]] = v[ed[q][0]]; eb[v[3 + q]] = v[ed[q][1]]; }
    }
    int nvert = 0;
    for (int v = 0; v < nv; v++) if (vid[v] == 0) vid[v] = nvert++;
    CsrHost &P = c->gen_P1;
    P.n = nv; P.m = nvert;
    P.rowptr.assign(nv + 1, 0); P.col.clear(); P.val.clear();
    for (int v = 0; v < nv; v++) {
      if (vid[v] >= 0) { P.col.push_back(vid[v]); P.val.push_back(1.0); }
      else {
        int a = vid[ea[v]], b = vid[eb[v]];
        if (a > b) std::swap(a, b);
        P.col.push_back(a); P.val.push_back(0.5); P.col.push_back(b); P.val.push_back(0.5);
      }
      P.rowptr[v + 1] = (int)P.col.size();
    }
  }
  // staging order of the assembly (see the header): contributions to one block entry / one node adjacent, in cell order
  std::vector<int> eptr((size_t)c->nnzv + 1, 0), fptr((size_t)nvo + 1, 0), fdst((size_t)nc * NL, -1);
  for (size_t t = 0; t < slot.size(); t++) if (slot[t] >= 0) eptr[slot[t] + 1]++;
  for (int k = 0; k < c->nnzv; k++) eptr[k + 1] += eptr[k];
  {
    std::vector<int> fill(eptr.begin(), eptr.end() - 1);
    for (size_t t = 0; t < slot.size(); t++) if (slot[t] >= 0) slot[t] = fill[slot[t]]++;  // (cells ascending: t runs over e first)
  }
  for (size_t t = 0; t < (size_t)nc * NL; t++) if (c->h_cells[t] < nvo) fptr[c->h_cells[t] + 1]++;
  for (int v = 0; v < nvo; v++) fptr[v + 1] += fptr[v];
  {
    std::vector<int> fill(fptr.begin(), fptr.end() - 1);
    for (size_t t = 0; t < (size_t)nc * NL; t++) if (c->h_cells[t] < nvo) fdst[t] = fill[c->h_cells[t]]++;
  }
  // diagonal mass scaled to the total measure (HRZ lumping: row sums vanish at P2 vertices); preconditioner only
  for (int v = 0; v < nv; v++) c->h_Ml[v] = mdiag[v] * (msum / dsum);
  // ---- uploads and allocations
  hipStream_t s = c->stream;
  std::vector<unsigned short> gflag(nc, 0);
  for (int k = 0; k < nfac; k++) gflag[fcell[k]] |= (unsigned short)(1u << flocal[k]);
  HIPCHK(c, c->coords.upload(c->h_coords, s));
  HIPCHK(c, c->cells.upload(c->h_cells, s));
  HIPCHK(c, c->gflag.upload(gflag, s));
  HIPCHK(c, c->gslot.upload(slot, s));
  HIPCHK(c, c->g_eptr.upload(eptr, s)); HIPCHK(c, c->g_fptr.upload(fptr, s)); HIPCHK(c, c->g_fdst.upload(fdst, s));
  HIPCHK(c, c->gE.alloc(9 * (size_t)nc * NL * NL)); HIPCHK(c, c->gEF.alloc(3 * (size_t)nc * NL));
  HIPCHK(c, c->vptr.upload(c->h_vptr, s));
  HIPCHK(c, c->vcol.upload(c->h_vcol, s));
  HIPCHK(c, c->vdiag.upload(c->h_vdiag, s));
  HIPCHK(c, c->A00.alloc(4 * (size_t)c->nnzv));
  HIPCHK(c, c->A01.alloc(2 * (size_t)c->nnzv));
  HIPCHK(c, c->A10.alloc(2 * (size_t)c->nnzv));
  HIPCHK(c, c->A11.alloc((size_t)c->nnzv));
  std::vector<unsigned char> cown(nc, 1);
  for (int e = 0; e < nc; e++) cown[e] = cells[(size_t)NL * e] < nvo ? 1 : 0;  // the rank that owns a cell's first node integrates it in global functionals
  HIPCHK(c, c->cell_owned.upload(cown, s));
  std::vector<double> rnd(2 * (size_t)nv);
  {
    uint64_t st = 0x2545F4914F6CDD1Dull;
    for (auto &v : rnd) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = ((st >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
    HIPCHK(c, c->prand.upload(rnd, s));
  }
  if (nfac) {
    HIPCHK(c, c->d_fac_cell.upload(c->fac_cell, s));
    HIPCHK(c, c->d_fac_local.upload(c->fac_local, s));
    HIPCHK(c, c->d_fac_marker.upload(c->fac_marker, s));
  }
  c->h_bcflag.assign(nv, 0);
  c->h_bcval.assign(3 * (size_t)nv, 0.0);
  c->h_bcmult.assign(3 * (size_t)nv, 0.0);
  HIPCHK(c, c->bcflag.alloc(nv));
  HIPCHK(c, c->bcval.alloc(3 * (size_t)nv));
  HIPCHK(c, c->bcmult.alloc(3 * (size_t)nv));
  c->bc_dirty = true;
  const size_t NLv = c->NL;
  HIPCHK(c, c->x.alloc(NLv)); HIPCHK(c, c->xt.alloc(NLv)); HIPCHK(c, c->xprev.alloc(NLv)); HIPCHK(c, c->xprev2.alloc(NLv));
  HIPCHK(c, c->F.alloc(NLv)); HIPCHK(c, c->dvec.alloc(NLv));
  HIPCHK(c, c->x.zero(s)); HIPCHK(c, c->xt.zero(s)); HIPCHK(c, c->xprev.zero(s)); HIPCHK(c, c->xprev2.zero(s)); HIPCHK(c, c->F.zero(s)); HIPCHK(c, c->dvec.zero(s));
  c->red_blocks = 1024;
  HIPCHK(c, c->red_partial.alloc((size_t)c->red_blocks * 260));
  HIPCHK(c, c->red_out.alloc(1024));
  HIPCHK(c, hipHostMalloc((void **)&c->h_pinned, 1024 * sizeof(double)));
  HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_pinned_dev, c->h_pinned, 0));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_h, hipEventDisableTiming));
  HIPCHK(c, c->dinvA.alloc(2 * (size_t)nv));
  HIPCHK(c, c->pu0.alloc(2 * (size_t)nv)); HIPCHK(c, c->pu1.alloc(2 * (size_t)nv)); HIPCHK(c, c->pu2.alloc(2 * (size_t)nv));
  HIPCHK(c, c->pr.alloc(2 * (size_t)nv));
  HIPCHK(c, c->pp0.alloc(nv)); HIPCHK(c, c->pp1.alloc(nv));
  c->mom_valid = true;  // no tau-moment pass: tau is evaluated inside the quadrature loop
  HIPCHK(c, hipStreamSynchronize(s));
  return 0;
}

// stiffness K [nloc][nloc] of one cell of the context's element type (nodes v in the caller's numbering, coordinates [..][2]):
// the global pressure Laplacian of a partitioned run (cfdh_set_global_pressure_space)
int cfdh_gen_element_stiffness(const cfdh_ctx *c, const int32_t *v, const double *X, double *K) {
  static GenTab T[3];
  static bool init = false;
  if (!init) { fill_tab<0>(T[0]); fill_tab<1>(T[1]); fill_tab<2>(T[2]); init = true; }
  const int et = c->etype, NL = c->nloc;
  const double J00 = X[2 * v[1]] - X[2 * v[0]], J01 = X[2 * v[2]] - X[2 * v[0]], J10 = X[2 * v[1] + 1] - X[2 * v[0] + 1], J11 = X[2 * v[2] + 1] - X[2 * v[0] + 1];
  const double det = J00 * J11 - J01 * J10, adet = std::fabs(det);
  if (!(adet > 0)) return CFDH_E_ARG;
  const double Ji[2][2] = {{J11 / det, -J01 / det}, {-J10 / det, J00 / det}};
  for (int k = 0; k < NL * NL; k++) K[k] = 0.0;
  for (int q = 0; q < GEN_NQ; q++) {
    double g[GEN_MAXL][2];
    for (int a = 0; a < NL; a++)
      for (int i = 0; i < 2; i++) g[a][i] = T[et].dphi[q][a][0] * Ji[0][i] + T[et].dphi[q][a][1] * Ji[1][i];
    for (int a = 0; a < NL; a++)
      for (int b = 0; b < NL; b++) K[a * NL + b] += adet * T[et].w[q] * (g[a][0] * g[b][0] + g[a][1] * g[b][1]);
  }
  return 0;
}

int kg_assemble(cfdh_ctx *c, const double *xstate, int mode) {
  if (mode == 0) mode = 2;
  const int NL = c->nloc, cpb = TPB / NL;
  GenArgs P;
  P.nc = c->nc; P.nvo = c->nvo; P.mode = mode;
  P.cells = c->cells.p; P.coords = c->coords.p; P.slot = c->gslot.p; P.flag = c->gflag.p;
  P.x = xstate; P.xprev = c->xprev.p; P.xprev2 = c->xprev2.p;
  P.bcflag = c->bcflag.p; P.bcval = c->bcval.p;
  P.dt = c->dt; P.rho = c->rho; P.mu = c->mu; P.muf = c->muf; P.f0 = c->f[0]; P.f1 = c->f[1];
  P.theta = c->ts_theta; P.a0 = c->ts_a[0]; P.a1 = c->ts_a[1]; P.a2 = c->ts_a[2];
  P.beta = c->bf_marker >= 0 ? c->bf_beta : 0.0;
  P.ds_terms = c->ds_terms ? 1 : 0;
  P.F = c->F.p; P.A00 = c->A00.p; P.A01 = c->A01.p; P.A10 = c->A10.p; P.A11 = c->A11.p;
  P.fdst = c->g_fdst.p; P.E = c->gE.p; P.EF = c->gEF.p;
  const dim3 grid((c->nc + cpb - 1) / cpb), block(TPB);
  prof_begin(c, 0);
#define CFDH_GEN_LAUNCH(ET) do { if (mode == 1) hipLaunchKernelGGL((gen_asm_kernel<ET, true>), grid, block, 0, c->stream, P); \
                                 else hipLaunchKernelGGL((gen_asm_kernel<ET, false>), grid, block, 0, c->stream, P); } while (0)
  if (c->etype == 1) CFDH_GEN_LAUNCH(1);
  else if (c->etype == 2) CFDH_GEN_LAUNCH(2);
  else CFDH_GEN_LAUNCH(0);
#undef CFDH_GEN_LAUNCH
  hipLaunchKernelGGL(gen_gather_F_kernel, dim3((c->nvo + TPB - 1) / TPB), block, 0, c->stream, c->nvo, c->g_fptr.p, c->gEF.p, c->F.p);
  if (mode == 1)
    hipLaunchKernelGGL(gen_gather_J_kernel, dim3((c->nnzv + TPB - 1) / TPB), block, 0, c->stream, c->nnzv, c->g_eptr.p, c->gE.p, c->A00.p, c->A01.p,
                       c->A10.p, c->A11.p);
  hipLaunchKernelGGL(gen_bc_rows_kernel, dim3((c->nvo + TPB - 1) / TPB), block, 0, c->stream, c->nvo, mode, c->bcflag.p, c->bcval.p, c->bcmult.p,
                     c->vdiag.p, xstate, c->F.p, c->A00.p, c->A11.p);
  prof_end(c, 0);
  HIPCHK(c, hipGetLastError());
  if (mode == 1) c->jac_valid = true;
  return 0;
}

int kg_functional_partials(cfdh_ctx *c, int kind, int marker, int nb) {
  const dim3 grid(nb), block(TPB);
#define ET_DISPATCH(KERNEL, ...)                                                                        \
  do {                                                                                                  \
    if (c->etype == 1) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, c->stream, __VA_ARGS__);         \
    else if (c->etype == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, c->stream, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<0>), grid, block, 0, c->stream, __VA_ARGS__);                       \
  } while (0)
  if (kind == 2 || kind == 3) ET_DISPATCH(gen_l2_kernel, c->nc, c->nvo, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
  else ET_DISPATCH(gen_facet_functional_kernel, c->nfac, marker, kind, c->nvo, c->d_fac_cell.p, c->d_fac_local.p, c->d_fac_marker.p, c->cells.p, c->cell_owned.p,
                   c->coords.p, c->x.p, c->mu, c->red_partial.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int kg_wss(cfdh_ctx *c, double *out) {
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)c->nv, c->stream));
  if (c->nfac > 0) {
    const dim3 grid((c->nfac + TPB - 1) / TPB), block(TPB);
    ET_DISPATCH(gen_wss_kernel, c->nfac, c->nvo, c->d_fac_cell.p, c->d_fac_local.p, c->cells.p, c->coords.p, c->x.p, c->mu, out);
  }
#undef ET_DISPATCH
  HIPCHK(c, hipGetLastError());
  return 0;
}
