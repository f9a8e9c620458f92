// Nodal equal-order elements in 3-D beyond P1 tetrahedra (SURVEY.md section 8f-4, 3-D half): Q1/Q1 hexahedra
// (/root/reference/src/scenarios/unit_cube_pipe.py:103-109, `create_box(..., cell_type=CellType.hexahedron)`; parallelepipeds:
// the geometry map must be affine) and P2/P2 tetrahedra (`p_grade = 2`, /root/reference/src/solvers/stabilized_schur_backflow.py:84-87
// on the 3-D meshes of scenario_factory.py:47-49).  Every node carries (u_x, u_y, u_z, p): the linear-algebra side -- node graph
// with 4 x 4 blocks in SoA arrays, block SpMV, FGMRES, Cahouet-Chabard + AMG -- is the tetrahedral code on the NODE graph.
//
// Element integration by quadrature: 7 x 7 x 7 Gauss points on hexahedra, the 171-point degree-13 rule of the P1 tau-moments
// on tetrahedra, i.e. 343 | 171 x (4 nloc)^2 Jacobian contributions per cell -- compute-bound by two orders of magnitude more than
// the closed-form P1 path, whatever the mapping.  The mapping that keeps the redundancy out:
//   * ONE CELL PER WORKGROUP, one lane per block (a, b) of the element matrix (64 lanes for a hexahedron, 100 of 128 for a P2
//     tetrahedron); a lane keeps its 4 x 4 block in registers over the whole quadrature loop;
//   * the points are processed in chunks of 64: lane t < 64 evaluates everything that belongs to the POINT -- physical
//     basis values / gradients / second derivatives of all nodes, the fields u_mid, grad u_mid, the strong residual, tau, tau_L --
//     once, into LDS; then all lanes sweep the chunk reading those values (broadcast reads) and only do the work of their block;
//   * no atomics: blocks and residual rows are staged by DESTINATION and summed in a fixed order by a second kernel, exactly as
//     the 2-D generic path does (cfdh_gen.hip): bitwise reproducible.
// Dirichlet rows / columns and the lifting F += J (g - x) are applied on the element level as DOLFINx does
// (stabilized_schur.py:144-175).  No MFMA: fp64, and the operands of the per-point products differ from lane to lane.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <numeric>

#include "cfdh_internal.hpp"
#include "cfdh_quad_gl.h"
#include "cfdh_quad_tet.h"
#include "cfdh_quad_tri.h"

#define TPB 256
#define G3_NQ 343  // table size: 7 x 7 x 7 Gauss points on hexahedra; tetrahedra use the first CFDH3_NQ (171) entries
__host__ __device__ constexpr int g3_nq(int et) { return et == 2 ? 343 : CFDH3_NQ; }
static_assert(CFDH3_NQ <= G3_NQ, "point table");

namespace {

// ET: 0 P1 tetrahedron (through the generic kernels: cross-check of the closed-form path), 1 P2 tetrahedron, 2 Q1 hexahedron
__host__ __device__ constexpr int g3_nloc(int et) { return et == 0 ? 4 : (et == 1 ? 10 : 8); }
// The assembly workgroup: 256 lanes = NG point groups of GS lanes; lane (g, a, b) sums block (a, b) of the element matrix over the
// points s = g, g + NG, ... of every 64-point chunk, and the groups are added in a fixed order at the end.
__host__ __device__ constexpr int g3_wgs(int) { return 256; }
__host__ __device__ constexpr int g3_gs(int et) { return et == 0 ? 16 : (et == 1 ? 128 : 64); }  // >= nloc^2

__constant__ double d3_pts[2][G3_NQ][4];  // [0] tetrahedron, [1] hexahedron: reference point (x, y, z), weight times reference measure
__constant__ double d3_tri[CFDH_NQ][4];   // triangle rule: barycentric point, weight (sum 1)
__constant__ double d3_gl2[2][2];         // 2-point Gauss on [0, 1]: points, weights

__device__ __host__ inline void tet_edge(int e, int &i, int &j) {  // Basix edge order of the tetrahedron
  const int E[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
  i = E[e][0]; j = E[e][1];
}
__device__ __host__ inline int facet_nvert(int et) { return et == 2 ? 4 : 3; }
__device__ __host__ inline int facet_vertex(int et, int f, int k) {
  if (et == 2) { const int H[6][4] = {{0, 1, 2, 3}, {0, 1, 4, 5}, {0, 2, 4, 6}, {1, 3, 5, 7}, {2, 3, 6, 7}, {4, 5, 6, 7}}; return H[f][k]; }
  const int T[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
  return T[f][k];
}
__device__ __host__ inline void ref_vertex3(int et, int v, double r[3]) {
  if (et == 2) { r[0] = v & 1; r[1] = (v >> 1) & 1; r[2] = (v >> 2) & 1; }
  else { r[0] = v == 1; r[1] = v == 2; r[2] = v == 3; }
}

// value and reference gradient of basis function a at the reference point pt
template <int ET>
__device__ __host__ inline void basis3(int a, const double pt[3], double &phi, double d[3]) {
  const double x = pt[0], y = pt[1], z = pt[2];
  if (ET == 2) {
    const int i = a & 1, j = (a >> 1) & 1, k = (a >> 2) & 1;
    const double fx = i ? x : 1 - x, fy = j ? y : 1 - y, fz = k ? z : 1 - z;
    const double dx = i ? 1.0 : -1.0, dy = j ? 1.0 : -1.0, dz = k ? 1.0 : -1.0;
    phi = fx * fy * fz;
    d[0] = dx * fy * fz; d[1] = fx * dy * fz; d[2] = fx * fy * dz;
    return;
  }
  const double l[4] = {1.0 - x - y - z, x, y, z};
  const double dl[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  if (ET == 0) { phi = l[a]; d[0] = dl[a][0]; d[1] = dl[a][1]; d[2] = dl[a][2]; return; }
  if (a < 4) {
    phi = l[a] * (2.0 * l[a] - 1.0);
    const double s = 4.0 * l[a] - 1.0;
    d[0] = s * dl[a][0]; d[1] = s * dl[a][1]; d[2] = s * dl[a][2];
  } else {
    int i, j;
    tet_edge(a - 4, i, j);
    phi = 4.0 * l[i] * l[j];
    for (int k = 0; k < 3; k++) d[k] = 4.0 * (l[i] * dl[j][k] + l[j] * dl[i][k]);
  }
}
// reference Hessian of basis function a at pt: (xx, xy, xz, yy, yz, zz)
template <int ET>
__device__ __host__ inline void hess3(int a, const double pt[3], double H[6]) {
  for (int k = 0; k < 6; k++) H[k] = 0.0;
  if (ET == 0) return;
  if (ET == 2) {
    const int i = a & 1, j = (a >> 1) & 1, k = (a >> 2) & 1;
    const double fx = i ? pt[0] : 1 - pt[0], fy = j ? pt[1] : 1 - pt[1], fz = k ? pt[2] : 1 - pt[2];
    const double dx = i ? 1.0 : -1.0, dy = j ? 1.0 : -1.0, dz = k ? 1.0 : -1.0;
    H[1] = dx * dy * fz; H[2] = dx * fy * dz; H[4] = fx * dy * dz;
    return;
  }
  const double dl[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  const int id[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  if (a < 4) { for (int k = 0; k < 6; k++) H[k] = 4.0 * dl[a][id[k][0]] * dl[a][id[k][1]]; }
  else {
    int i, j;
    tet_edge(a - 4, i, j);
    for (int k = 0; k < 6; k++) H[k] = 4.0 * (dl[i][id[k][0]] * dl[j][id[k][1]] + dl[j][id[k][0]] * dl[i][id[k][1]]);
  }
}
// physical second derivatives (xx, xy, xz, yy, yz, zz) from the reference ones: Hs_ij = sum_kl Href_kl Ji[k][i] Ji[l][j]
__device__ __host__ inline void hess_phys(const double Hr[6], const double Ji[3][3], double Hs[6]) {
  const double R[3][3] = {{Hr[0], Hr[1], Hr[2]}, {Hr[1], Hr[3], Hr[4]}, {Hr[2], Hr[4], Hr[5]}};
  double T[3][3];  // T[k][j] = sum_l R[k][l] Ji[l][j]
  for (int k = 0; k < 3; k++)
    for (int j = 0; j < 3; j++) T[k][j] = R[k][0] * Ji[0][j] + R[k][1] * Ji[1][j] + R[k][2] * Ji[2][j];
  const int id[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
  for (int q = 0; q < 6; q++) { const int i = id[q][0], j = id[q][1]; Hs[q] = Ji[0][i] * T[0][j] + Ji[1][i] * T[1][j] + Ji[2][i] * T[2][j]; }
}
__device__ __host__ inline double sym6(const double H[6], int i, int j) {
  const int m[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  return H[m[i][j]];
}

__device__ __forceinline__ void tau_pair3(double s, double h, double dt, double nu, double &tau, double &tauL) {
  double t1 = 4.0 * s;
  t1 = t1 < 1e-30 ? 1e-30 : t1;
  t1 /= h * h;
  tau = cfdh_rsqrt(t1 + 4.0 / (dt * dt) + 16.0 * nu * nu / (h * h * h * h));
  const double vn = sqrt(s), Re = vn * h / (2.0 * nu), z = Re <= 3.0 ? Re * (1.0 / 3.0) : 1.0;
  tauL = vn * h * z * 0.5;
}

// affine geometry of a cell from its vertex coordinates: Ji[k][i] = d xi_k / d x_i, |det|, h
template <int ET, typename XT>
__device__ __host__ inline void geom3(const XT &X, double Ji[3][3], double &adet, double &h) {
  constexpr int c3 = ET == 2 ? 4 : 3, NV = ET == 2 ? 8 : 4;
  double J[3][3];
  for (int i = 0; i < 3; i++) { J[i][0] = X[1][i] - X[0][i]; J[i][1] = X[2][i] - X[0][i]; J[i][2] = X[c3][i] - X[0][i]; }
  const double det = J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0]) + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
  const double id = 1.0 / det;
  Ji[0][0] = (J[1][1] * J[2][2] - J[1][2] * J[2][1]) * id; Ji[0][1] = (J[0][2] * J[2][1] - J[0][1] * J[2][2]) * id; Ji[0][2] = (J[0][1] * J[1][2] - J[0][2] * J[1][1]) * id;
  Ji[1][0] = (J[1][2] * J[2][0] - J[1][0] * J[2][2]) * id; Ji[1][1] = (J[0][0] * J[2][2] - J[0][2] * J[2][0]) * id; Ji[1][2] = (J[0][2] * J[1][0] - J[0][0] * J[1][2]) * id;
  Ji[2][0] = (J[1][0] * J[2][1] - J[1][1] * J[2][0]) * id; Ji[2][1] = (J[0][1] * J[2][0] - J[0][0] * J[2][1]) * id; Ji[2][2] = (J[0][0] * J[1][1] - J[0][1] * J[1][0]) * id;
  adet = fabs(det);
  h = 0.0;
  for (int a = 0; a < NV; a++)
    for (int b = a + 1; b < NV; b++) {
      const double d0 = X[a][0] - X[b][0], d1 = X[a][1] - X[b][1], d2 = X[a][2] - X[b][2];
      h = fmax(h, sqrt(d0 * d0 + d1 * d1 + d2 * d2));
    }
}
// outward unit normal and measure of local facet f
template <int ET, typename XT>
__device__ __host__ inline void facet_geom3(const XT &X, int f, double n[3], double &area) {
  constexpr int NV = ET == 2 ? 8 : 4, et = ET;
  const int v0 = facet_vertex(et, f, 0), v1 = facet_vertex(et, f, 1), v2 = facet_vertex(et, f, 2);
  double e1[3], e2[3], cen[3] = {0, 0, 0}, fc[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) { e1[i] = X[v1][i] - X[v0][i]; e2[i] = X[v2][i] - X[v0][i]; }
  n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
  const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  area = ET == 2 ? nn : 0.5 * nn;
  for (int a = 0; a < NV; a++) for (int i = 0; i < 3; i++) cen[i] += X[a][i] * (1.0 / NV);
  const int nfv = facet_nvert(et);
  for (int k = 0; k < nfv; k++) for (int i = 0; i < 3; i++) fc[i] += X[facet_vertex(et, f, k)][i] / nfv;
  const double sg = ((fc[0] - cen[0]) * n[0] + (fc[1] - cen[1]) * n[1] + (fc[2] - cen[2]) * n[2]) < 0 ? -1.0 : 1.0;
  for (int i = 0; i < 3; i++) n[i] *= sg / nn;
}
// facet quadrature: number of points; point q of local facet f in cell reference coordinates, weight (sum 1)
template <int ET> __device__ __host__ constexpr int facet_nq() { return ET == 2 ? 4 : (ET == 1 ? CFDH_NQ : 6); }
template <int ET>
__device__ inline void facet_point(int f, int q, double pt[3], double &w) {
  constexpr int et = ET;
  double rv[4][3];
  for (int k = 0; k < facet_nvert(et); k++) ref_vertex3(et, facet_vertex(et, f, k), rv[k]);
  if (ET == 2) {
    const double s = d3_gl2[0][q >> 1], t = d3_gl2[0][q & 1];
    for (int i = 0; i < 3; i++) pt[i] = (1 - s) * (1 - t) * rv[0][i] + s * (1 - t) * rv[1][i] + (1 - s) * t * rv[2][i] + s * t * rv[3][i];
    w = d3_gl2[1][q >> 1] * d3_gl2[1][q & 1];
  } else if (ET == 1) {
    for (int i = 0; i < 3; i++) pt[i] = d3_tri[q][0] * rv[0][i] + d3_tri[q][1] * rv[1][i] + d3_tri[q][2] * rv[2][i];
    w = d3_tri[q][3];
  } else {
    const double A = 0.659027622374092, B = 0.231933368553031, C = 0.109039009072877;
    const double P[6][3] = {{A, B, C}, {A, C, B}, {B, A, C}, {B, C, A}, {C, A, B}, {C, B, A}};
    for (int i = 0; i < 3; i++) pt[i] = P[q][0] * rv[0][i] + P[q][1] * rv[1][i] + P[q][2] * rv[2][i];
    w = 1.0 / 6.0;
  }
}

// offsets of a node's velocity / pressure in the state layout [u owned 3 nvo | p owned nvo | (u_x, u_y, u_z, p) per ghost]
__device__ __forceinline__ size_t g3uo(int v, int nvo) { return v < nvo ? 3 * (size_t)v : 4 * (size_t)nvo + 4 * (size_t)(v - nvo); }
__device__ __forceinline__ size_t g3po(int v, int nvo) { return v < nvo ? 3 * (size_t)nvo + v : 4 * (size_t)nvo + 4 * (size_t)(v - nvo) + 3; }

struct Gen3Args {
  int nc, nvo, mode;  // mode 1: F + J, 2: F only (lifting included)
  const int *cells;
  const double *coords;
  const int *slot;               // [nc][nloc * nloc]: position of block (a, b) of the cell in the staging array E
  const int *fdst;               // [nc][nloc * nloc]: position of the residual contribution of lane (a, b) in EF
  const unsigned short *flag;    // [nc] bit f: exterior facet f, bit 8 + f: backflow facet f
  const double *x, *xprev, *xprev2;
  const unsigned char *bcflag;   // per node: bits 0-2 velocity components, bit 3 pressure
  const double *bcval;           // [nv][4]
  double dt, rho, mu, muf, f[3], theta, a0, a1, a2, beta;
  int ds_terms;
  double *E, *EF;                // [nc nloc^2][16] (A00 row-major 9 | A01 3 | A10 3 | A11), [nc nloc^2][4]
};

template <int NL>
struct Cell3 {
  double X[NL][3], ub[NL][3], wn[NL][3], un[NL][3], p[NL], lift[NL][4];
  unsigned char bc[NL];
};

#define FLD 23  // per-point record: uq 0-2 | G 3-11 (G[i][j] = d_i ubar_j) | R 12-14 | rho (w + C - f) 15-17 | pq 18 | tau 19 | tauL 20 | dv 21

template <int ET, bool JAC>
__global__ __launch_bounds__(g3_wgs(ET), 2) void gen3_asm_kernel(Gen3Args P) {
  constexpr int NL = g3_nloc(ET), WGS = g3_wgs(ET), GS = g3_gs(ET), NG = WGS / GS;
  constexpr int CH = ET == 2 ? 32 : 64;              // points per chunk (32 on hexahedra: their per-point Hessians would cost a third workgroup per CU)
  constexpr int NQ = g3_nq(ET), NCH = (NQ + CH - 1) / CH;
  __shared__ Cell3<NL> D;
  __shared__ double fld[CH][FLD];
  __shared__ double bas[CH][NL][4];                  // physical (phi, grad phi) of every node at the chunk's points
  __shared__ double hes[ET == 2 ? CH : 1][NL][6];    // physical second derivatives: per point on hexahedra, cell constants on P2 tetrahedra
  __shared__ double geo[12];                         // Ji (9), |det|, h, needj
  const int cell = blockIdx.x, t = threadIdx.x;
  const int grp = t / GS, blk = t % GS;
  const int a = blk / NL, b = blk % NL;
  const bool live = blk < NL * NL;
  const int nvo = P.nvo;
  if (t < NL) {
    const int v = P.cells[(size_t)cell * NL + t];
    const unsigned char bf = P.bcflag[v];
    D.bc[t] = bf;
    for (int i = 0; i < 3; i++) {
      D.X[t][i] = P.coords[3 * (size_t)v + i];
      const double u = P.x[g3uo(v, nvo) + i], un = P.xprev[g3uo(v, nvo) + i];
      D.un[t][i] = un;
      D.ub[t][i] = P.theta * u + (1.0 - P.theta) * un;
      D.wn[t][i] = (P.a0 * u + P.a1 * un + (P.a2 != 0.0 ? P.a2 * P.xprev2[g3uo(v, nvo) + i] : 0.0)) / P.dt;
      D.lift[t][i] = (bf >> i) & 1 ? P.bcval[4 * (size_t)v + i] - u : 0.0;
    }
    const double pv = P.x[g3po(v, nvo)];
    D.p[t] = pv;
    D.lift[t][3] = (bf >> 3) & 1 ? P.bcval[4 * (size_t)v + 3] - pv : 0.0;
  }
  __syncthreads();
  if (t == 0) {
    double Ji0[3][3], ad, hh;
    geom3<ET>(D.X, Ji0, ad, hh);
    for (int k = 0; k < 9; k++) geo[k] = Ji0[k / 3][k % 3];
    geo[9] = ad; geo[10] = hh;
    bool nj = JAC;
    if (!JAC)
      for (int c = 0; c < NL; c++) nj = nj || D.lift[c][0] != 0.0 || D.lift[c][1] != 0.0 || D.lift[c][2] != 0.0 || D.lift[c][3] != 0.0;
    geo[11] = nj ? 1.0 : 0.0;
  }
  if (ET != 2 && t < NL) {  // constant second derivatives (P2 tetrahedra; zero for P1): needs Ji, formed redundantly here
    double Ji0[3][3], ad, hh, Hr[6], Hs[6];
    const double p0[3] = {0.25, 0.25, 0.25};
    geom3<ET>(D.X, Ji0, ad, hh);
    hess3<ET>(t, p0, Hr);
    hess_phys(Hr, Ji0, Hs);
    for (int k = 0; k < 6; k++) hes[0][t][k] = Hs[k];
  }
  __syncthreads();
  double Ji[3][3];
  for (int k = 0; k < 9; k++) Ji[k / 3][k % 3] = geo[k];
  const double adet = geo[9], h = geo[10];
  const bool needj = geo[11] != 0.0;
  const double rho = P.rho, mu = P.mu, th = P.theta, a0dt = P.a0 / P.dt, nu = mu / rho;
  // this lane's block of the element matrix and (b == 0) the residual rows of node a
  double Juu[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Jup[3] = {0, 0, 0}, Jpu[3] = {0, 0, 0}, Jpp = 0.0, Fa[4] = {0, 0, 0, 0};
  for (int ch = 0; ch < NCH; ch++) {
    // ---- basis functions of the chunk's points: one (point, node) pair per lane and pass; the node is uniform in a wavefront
    for (int item = t; item < CH * NL; item += WGS) {
      const int sp = item % CH, c = item / CH, q = ch * CH + sp;
      if (q >= NQ) continue;
      const double *pt = d3_pts[ET == 2 ? 1 : 0][q];
      double ph, dr[3];
      basis3<ET>(c, pt, ph, dr);
      bas[sp][c][0] = ph;
      for (int i = 0; i < 3; i++) bas[sp][c][1 + i] = dr[0] * Ji[0][i] + dr[1] * Ji[1][i] + dr[2] * Ji[2][i];
      if (ET == 2) {
        double Hr[6], Hs[6];
        hess3<ET>(c, pt, Hr);
        hess_phys(Hr, Ji, Hs);
        for (int k = 0; k < 6; k++) hes[sp][c][k] = Hs[k];
      }
    }
    __syncthreads();
    const int q = ch * CH + t;
    if (t < CH && q < NQ) {
      // ---- everything that belongs to the point, once
      const double *pt = d3_pts[ET == 2 ? 1 : 0][q];
      double uq[3] = {0, 0, 0}, wv[3] = {0, 0, 0}, unq[3] = {0, 0, 0}, G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gp[3] = {0, 0, 0}, pq = 0.0;
      double visc[3] = {0, 0, 0};
#pragma unroll
      for (int c = 0; c < NL; c++) {
        const double ph = bas[t][c][0], g[3] = {bas[t][c][1], bas[t][c][2], bas[t][c][3]};
        double Hs[6];
        for (int k = 0; k < 6; k++) Hs[k] = hes[ET == 2 ? t : 0][c][k];
        const double lapc = Hs[0] + Hs[3] + Hs[5];
        for (int i = 0; i < 3; i++) {
          uq[i] += ph * D.ub[c][i]; wv[i] += ph * D.wn[c][i]; unq[i] += ph * D.un[c][i];
          gp[i] += g[i] * D.p[c];
          for (int j = 0; j < 3; j++) G[i][j] += g[i] * D.ub[c][j];
          if (ET != 0) visc[i] += mu * (lapc * D.ub[c][i] + sym6(Hs, i, 0) * D.ub[c][0] + sym6(Hs, i, 1) * D.ub[c][1] + sym6(Hs, i, 2) * D.ub[c][2]);
        }
        pq += ph * D.p[c];
      }
      double tau, tauL;
      tau_pair3(unq[0] * unq[0] + unq[1] * unq[1] + unq[2] * unq[2], h, P.dt, nu, tau, tauL);
      double *fr = fld[t];
      for (int i = 0; i < 3; i++) {
        const double Ci = uq[0] * G[0][i] + uq[1] * G[1][i] + uq[2] * G[2][i];
        fr[i] = uq[i];
        fr[12 + i] = rho * (wv[i] + Ci) - visc[i] + gp[i] - rho * P.f[i];
        fr[15 + i] = rho * (wv[i] + Ci - P.f[i]);
        for (int j = 0; j < 3; j++) fr[3 + 3 * i + j] = G[i][j];
      }
      fr[18] = pq; fr[19] = tau; fr[20] = tauL; fr[21] = adet * pt[3];
    }
    __syncthreads();
    const int npt = min(CH, NQ - ch * CH);
    // residual-only pass of a cell without lifting: no Jacobian block is needed, so the NL lanes of a row share the row's points
    // (lane (a, b) takes the points grp + NG b, grp + NG (b + NL), ...) and the row is summed over b at the end
    const bool fsplit = !needj;
    if (live) {
      for (int s = fsplit ? grp + NG * b : grp; s < npt; s += fsplit ? NG * NL : NG) {
        const double *fr = fld[s];
        const double pha = bas[s][a][0], ga[3] = {bas[s][a][1], bas[s][a][2], bas[s][a][3]};
        const double uq[3] = {fr[0], fr[1], fr[2]}, R[3] = {fr[12], fr[13], fr[14]};
        const double tau = fr[19], tauL = fr[20], dv = fr[21];
        const double bga = uq[0] * ga[0] + uq[1] * ga[1] + uq[2] * ga[2];
        if (b == 0 || fsplit) {
          const double pq = fr[18], divu = fr[3] + fr[7] + fr[11];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            double v = pha * fr[15 + i];
#pragma unroll
            for (int j = 0; j < 3; j++) v += mu * ga[j] * (fr[3 + 3 * i + j] + fr[3 + 3 * j + i]);
            v += -pq * ga[i] + tau * R[i] * bga + tauL * rho * divu * ga[i];
            Fa[i] += dv * v;
          }
          Fa[3] += dv * (pha * divu + tau / rho * (R[0] * ga[0] + R[1] * ga[1] + R[2] * ga[2]));
        }
        if (!needj) continue;
        // ---- block (a, b) of the Jacobian at this point, with everything that does not depend on (i, j) formed once:
        //   dWC_i(j) = k1 G[j][i] + delta_ij s_d,   dR_i(j) = dWC_i(j) - mu th (delta_ij lap_b + H_b[i][j])
        //   Juu[i][j] += A1 dWC_i(j) - A2 (delta_ij lap_b + H_b[i][j]) + ga[j] w_i + z_j ga[i] + delta_ij dv mu th (ga . gb)
        const double phb = bas[s][b][0], gb[3] = {bas[s][b][1], bas[s][b][2], bas[s][b][3]};
        const double *Hb = hes[ET == 2 ? s : 0][b];
        const double lapb = Hb[0] + Hb[3] + Hb[5];
        const double bgb = uq[0] * gb[0] + uq[1] * gb[1] + uq[2] * gb[2];
        const double gg = ga[0] * gb[0] + ga[1] * gb[1] + ga[2] * gb[2];
        const double ct = dv * tau, ctb = ct * bga, mt = mu * th;
        const double A1 = dv * pha + ctb, A2 = ctb * mt;
        const double k1 = rho * th * phb, sd = rho * (a0dt * phb + th * bgb);
        const double B1 = A1 * k1, Dg = A1 * sd - A2 * lapb + dv * mt * gg;
        const double c3 = ct * th * phb, c4 = dv * rho * th * tauL, dmt = dv * mt;
        const double w[3] = {dmt * gb[0] + c3 * R[0], dmt * gb[1] + c3 * R[1], dmt * gb[2] + c3 * R[2]};
        const double z[3] = {c4 * gb[0], c4 * gb[1], c4 * gb[2]};
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) Juu[i][j] += B1 * fr[3 + 3 * j + i] - A2 * sym6(Hb, i, j) + ga[j] * w[i] + z[j] * ga[i];
          Juu[i][i] += Dg;
        }
        const double cr = ct / rho, e1 = dv * th * pha, e2 = sd - mt * lapb;
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double Gga = fr[3 + 3 * j] * ga[0] + fr[4 + 3 * j] * ga[1] + fr[5 + 3 * j] * ga[2];
          const double Hga = sym6(Hb, 0, j) * ga[0] + sym6(Hb, 1, j) * ga[1] + sym6(Hb, 2, j) * ga[2];
          Jpu[j] += e1 * gb[j] + cr * (k1 * Gga + e2 * ga[j] - mt * Hga);
        }
        const double f1 = -dv * phb;
#pragma unroll
        for (int i = 0; i < 3; i++) Jup[i] += f1 * ga[i] + ctb * gb[i];
        Jpp += cr * gg;
      }
    }
    __syncthreads();
  }
  // ---- the point groups are added in the order 0, 1, ..., NG - 1 (fixed: bitwise reproducible); VB values per lane and pass fit the
  //      basis table, which is free now
  {
    constexpr int VB = CH * NL * 4 / WGS;
    static_assert(VB >= 1 && VB * WGS <= CH * NL * 4, "the reduction buffer is the basis table");
    double *red = &bas[0][0][0];
    double acc[20];
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) acc[3 * i + j] = Juu[i][j]; acc[9 + i] = Jup[i]; acc[12 + i] = Jpu[i]; }
    acc[15] = Jpp;
    for (int i = 0; i < 4; i++) acc[16 + i] = Fa[i];
#pragma unroll
    for (int k0 = 0; k0 < 20; k0 += VB) {
      if (grp > 0) {
#pragma unroll
        for (int k = 0; k < VB; k++) if (k0 + k < 20) red[(size_t)k * WGS + t] = acc[k0 + k];
      }
      __syncthreads();
      if (grp == 0) {
#pragma unroll
        for (int k = 0; k < VB; k++)
          if (k0 + k < 20)
            for (int g = 1; g < NG; g++) acc[k0 + k] += red[(size_t)k * WGS + g * GS + blk];
      }
      __syncthreads();
    }
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) Juu[i][j] = acc[3 * i + j]; Jup[i] = acc[9 + i]; Jpu[i] = acc[12 + i]; }
    Jpp = acc[15];
    for (int i = 0; i < 4; i++) Fa[i] = acc[16 + i];
    if (!needj) {  // rows shared by their NL lanes: sum over b in the order 0, 1, ..., NL - 1 (uniform in the workgroup)
      if (live && grp == 0) for (int i = 0; i < 4; i++) red[4 * blk + i] = Fa[i];
      __syncthreads();
      if (live && grp == 0 && b == 0)
        for (int i = 0; i < 4; i++) {
          double v = red[4 * blk + i];
          for (int bb = 1; bb < NL; bb++) v += red[4 * (blk + bb) + i];
          Fa[i] = v;
        }
    }
  }
  if (!live || grp != 0) return;
  // ---- Dirichlet handling on the element level: lifting with the FULL block, then constrained rows and columns dropped
  const unsigned bca = D.bc[a], bcb = D.bc[b];
  const double l0 = D.lift[b][0], l1 = D.lift[b][1], l2 = D.lift[b][2], l3 = D.lift[b][3];
  double Fl[4];
  for (int i = 0; i < 3; i++) Fl[i] = (b == 0 ? Fa[i] : 0.0) + Juu[i][0] * l0 + Juu[i][1] * l1 + Juu[i][2] * l2 + Jup[i] * l3;
  Fl[3] = (b == 0 ? Fa[3] : 0.0) + Jpu[0] * l0 + Jpu[1] * l1 + Jpu[2] * l2 + Jpp * l3;
  const int fd = P.fdst[(size_t)cell * NL * NL + blk];
  if (fd < 0) return;  // row of a ghost node: assembled by its owner (one-cell overlap of the partition)
  double *ef = P.EF + 4 * (size_t)fd;
  for (int i = 0; i < 4; i++) ef[i] = ((bca >> i) & 1u) ? 0.0 : Fl[i];
  if (P.mode != 1) return;
  double *eb = P.E + 16 * (size_t)P.slot[(size_t)cell * NL * NL + blk];
  for (int i = 0; i < 3; i++) {
    const bool ri = (bca >> i) & 1u;
    for (int j = 0; j < 3; j++) eb[3 * i + j] = (ri || ((bcb >> j) & 1u)) ? 0.0 : Juu[i][j];
    eb[9 + i] = (ri || (bcb & 8u)) ? 0.0 : Jup[i];
  }
  const bool rp = bca & 8u;
  for (int j = 0; j < 3; j++) eb[12 + j] = (rp || ((bcb >> j) & 1u)) ? 0.0 : Jpu[j];
  eb[15] = (rp || (bcb & 8u)) ? 0.0 : Jpp;
}

// Exterior-facet terms (the ds pair of the do-nothing outlet, the backflow term) of the flagged cells, ADDED to what the volume kernel
// staged: every (cell, a, b) entry of E / EF belongs to one lane, so the update needs no atomics and keeps the fixed summation order.
// The Dirichlet handling is linear in the block, so lifting and masks apply to the facet part on its own.  A kernel of its own
// because inlined into the volume kernel it cost that kernel a third of its registers (314 instead of 218) on every cell.
template <int ET>
__global__ __launch_bounds__(128) void gen3_facet_kernel(Gen3Args P, const int *__restrict__ fcells) {
  constexpr int NL = g3_nloc(ET), NF = ET == 2 ? 6 : 4;
  __shared__ Cell3<NL> D;
  __shared__ double geo[9];
  const int cell = fcells[blockIdx.x], t = threadIdx.x;
  const int blk = t, a = blk / NL, b = blk % NL;
  const bool live = blk < NL * NL;
  const int nvo = P.nvo;
  if (t < NL) {
    const int v = P.cells[(size_t)cell * NL + t];
    const unsigned char bf = P.bcflag[v];
    D.bc[t] = bf;
    for (int i = 0; i < 3; i++) {
      D.X[t][i] = P.coords[3 * (size_t)v + i];
      const double u = P.x[g3uo(v, nvo) + i], un = P.xprev[g3uo(v, nvo) + i];
      D.un[t][i] = un;
      D.ub[t][i] = P.theta * u + (1.0 - P.theta) * un;
      D.lift[t][i] = (bf >> i) & 1 ? P.bcval[4 * (size_t)v + i] - u : 0.0;
    }
    const double pv = P.x[g3po(v, nvo)];
    D.p[t] = pv;
    D.lift[t][3] = (bf >> 3) & 1 ? P.bcval[4 * (size_t)v + 3] - pv : 0.0;
  }
  __syncthreads();
  if (t == 0) {
    double Ji0[3][3], ad, hh;
    geom3<ET>(D.X, Ji0, ad, hh);
    for (int k = 0; k < 9; k++) geo[k] = Ji0[k / 3][k % 3];
  }
  __syncthreads();
  if (!live) return;
  double Ji[3][3];
  for (int k = 0; k < 9; k++) Ji[k / 3][k % 3] = geo[k];
  const double rho = P.rho, th = P.theta;
  double Juu[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, Jup[3] = {0, 0, 0}, Fa[4] = {0, 0, 0, 0};
  const unsigned fl = P.flag[cell];
  // the facet terms enter the velocity rows of node a only: when all three are Dirichlet rows (a node on a no-slip wall) they would
  // be dropped below -- and the lanes of the nodes off the facet see a vanishing test function.  Most exterior facets are walls.
  if ((D.bc[a] & 7u) == 7u) return;
  if (fl) {
    for (int f = 0; f < NF; f++) {
      const bool ext = P.ds_terms && ((fl >> f) & 1u), bfl = P.beta != 0.0 && ((fl >> (8 + f)) & 1u);
      if (!ext && !bfl) continue;
      {  // a facet all of whose nodes are no-slip nodes contributes to dropped rows only (uniform in the workgroup)
        constexpr int et = ET;
        bool wall = true;
        for (int k = 0; k < facet_nvert(et); k++) wall = wall && (D.bc[facet_vertex(et, f, k)] & 7u) == 7u;
        if (ET == 1)
          for (int e = 0; e < 6; e++) {
            int i, j;
            tet_edge(e, i, j);
            if (i != f && j != f) wall = wall && (D.bc[4 + e] & 7u) == 7u;
          }
        if (wall) continue;
      }
      double n[3], area;
      facet_geom3<ET>(D.X, f, n, area);
      for (int q = 0; q < facet_nq<ET>(); q++) {
        double pt[3], w;
        facet_point<ET>(f, q, pt, w);
        const double m = area * w;
        double pha, dra[3], phb, drb[3];
        basis3<ET>(a, pt, pha, dra);
        if (pha == 0.0) continue;  // test function vanishes on this facet
        basis3<ET>(b, pt, phb, drb);
        double gb[3];
        for (int i = 0; i < 3; i++) gb[i] = drb[0] * Ji[0][i] + drb[1] * Ji[1][i] + drb[2] * Ji[2][i];
        double uq[3] = {0, 0, 0}, G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, pq = 0.0, sn = 0.0;
        for (int c = 0; c < NL; c++) {
          double ph, dr[3];
          basis3<ET>(c, pt, ph, dr);
          for (int i = 0; i < 3; i++) {
            const double gi = dr[0] * Ji[0][i] + dr[1] * Ji[1][i] + dr[2] * Ji[2][i];
            uq[i] += ph * D.ub[c][i];
            sn += ph * D.un[c][i] * n[i];
            for (int j = 0; j < 3; j++) G[i][j] += gi * D.ub[c][j];
          }
          pq += ph * D.p[c];
        }
        if (ext) {
          for (int i = 0; i < 3; i++) {
            if (b == 0) Fa[i] += m * pha * (pq * n[i] - P.muf * (G[i][0] * n[0] + G[i][1] * n[1] + G[i][2] * n[2]));
            Jup[i] += m * pha * phb * n[i];
            for (int j = 0; j < 3; j++) Juu[i][j] -= P.muf * th * m * pha * gb[i] * n[j];
          }
        }
        if (bfl) {
          const double cq = P.beta * rho * 0.5 * (sn - fabs(sn)) * m;
          for (int i = 0; i < 3; i++) {
            if (b == 0) Fa[i] -= cq * pha * uq[i];
            Juu[i][i] -= th * cq * pha * phb;
          }
        }
      }
    }
  }
  const unsigned bca = D.bc[a], bcb = D.bc[b];
  const double l0 = D.lift[b][0], l1 = D.lift[b][1], l2 = D.lift[b][2], l3 = D.lift[b][3];
  const int fd = P.fdst[(size_t)cell * NL * NL + blk];
  if (fd < 0) return;
  double *ef = P.EF + 4 * (size_t)fd;
  for (int i = 0; i < 3; i++) {
    const double Fl = (b == 0 ? Fa[i] : 0.0) + Juu[i][0] * l0 + Juu[i][1] * l1 + Juu[i][2] * l2 + Jup[i] * l3;
    if (!((bca >> i) & 1u)) ef[i] += Fl;
  }
  if (P.mode != 1) return;
  double *eb = P.E + 16 * (size_t)P.slot[(size_t)cell * NL * NL + blk];
  for (int i = 0; i < 3; i++) {
    const bool ri = (bca >> i) & 1u;
    for (int j = 0; j < 3; j++) if (!(ri || ((bcb >> j) & 1u))) eb[3 * i + j] += Juu[i][j];
    if (!(ri || (bcb & 8u))) eb[9 + i] += Jup[i];
  }
}

// second phase: fixed-order sums of the staged contributions, every output written once
__global__ __launch_bounds__(TPB) void gen3_gather_F_kernel(int nvo, const int *__restrict__ fptr, const double *__restrict__ EF, double *__restrict__ F) {
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= nvo) return;
  double f[4] = {0, 0, 0, 0};
  for (int k = fptr[v], ke = fptr[v + 1]; k < ke; k++)
    for (int i = 0; i < 4; i++) f[i] += EF[4 * (size_t)k + i];
  F[3 * (size_t)v] = f[0]; F[3 * (size_t)v + 1] = f[1]; F[3 * (size_t)v + 2] = f[2]; F[3 * (size_t)nvo + v] = f[3];
}
__global__ __launch_bounds__(TPB) void gen3_gather_J_kernel(int nnz, const int *__restrict__ eptr, const double *__restrict__ E, double *__restrict__ A00,
                                                            double *__restrict__ A01, double *__restrict__ A10, double *__restrict__ A11) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nnz) return;
  double a[16];
  for (int t = 0; t < 16; t++) a[t] = 0.0;
  for (int q = eptr[k], qe = eptr[k + 1]; q < qe; q++) {
    const double *e = E + 16 * (size_t)q;
#pragma unroll
    for (int t = 0; t < 16; t++) a[t] += e[t];
  }
  for (int t = 0; t < 9; t++) A00[9 * (size_t)k + t] = a[t];
  for (int t = 0; t < 3; t++) { A01[3 * (size_t)k + t] = a[9 + t]; A10[3 * (size_t)k + t] = a[12 + t]; }
  A11[k] = a[15];
}
// rows of constrained dofs: F = x - g; diagonal = number of Dirichlet objects holding the dof (stabilized_schur.py:144-175)
__global__ __launch_bounds__(TPB) void gen3_bc_rows_kernel(int nvo, int mode, const unsigned char *__restrict__ bcflag, const double *__restrict__ bcval,
                                                           const double *__restrict__ bcmult, const int *__restrict__ vdiag, const double *__restrict__ x,
                                                           double *__restrict__ F, double *__restrict__ A00, double *__restrict__ A11) {
  const int v = blockIdx.x * TPB + threadIdx.x;
  if (v >= nvo) return;
  const unsigned bf = bcflag[v];
  if (!bf) return;
  const size_t k = (size_t)vdiag[v];
  for (int i = 0; i < 3; i++)
    if ((bf >> i) & 1u) {
      F[3 * (size_t)v + i] = x[3 * (size_t)v + i] - bcval[4 * (size_t)v + i];
      if (mode == 1) A00[9 * k + 4 * i] += bcmult[4 * (size_t)v + i];
    }
  if (bf & 8u) {
    F[3 * (size_t)nvo + v] = x[3 * (size_t)nvo + v] - bcval[4 * (size_t)v + 3];
    if (mode == 1) A11[k] += bcmult[4 * (size_t)v + 3];
  }
}

__device__ __forceinline__ double wave_sum3(double v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return v;
}
__device__ __forceinline__ double block_sum3(double v, double *sh) {
  v = wave_sum3(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// int u.u and int p^2 with the element's own mass matrix (scenario.py:315-324)
template <int ET>
__global__ __launch_bounds__(TPB) void gen3_l2_kernel(int nc, int nvo, const int *__restrict__ cells, const unsigned char *__restrict__ cell_owned,
                                                      const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  constexpr int NL = g3_nloc(ET), NV = ET == 2 ? 8 : 4;
  __shared__ double sh[4];
  double au = 0, ap = 0;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nc; e += gridDim.x * TPB) {
    if (!cell_owned[e]) continue;  // every cell is integrated by exactly one rank
    double X[NV][3], u[NL][3], p[NL];
    for (int a = 0; a < NL; a++) {
      const int v = cells[(size_t)e * NL + a];
      if (a < NV) for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)v + i];
      for (int i = 0; i < 3; i++) u[a][i] = x[g3uo(v, nvo) + i];
      p[a] = x[g3po(v, nvo)];
    }
    double Ji[3][3], adet, h;
    geom3<ET>(X, Ji, adet, h);
    for (int q = 0; q < g3_nq(ET); q++) {
      const double *pt = d3_pts[ET == 2 ? 1 : 0][q];
      double uq[3] = {0, 0, 0}, pq = 0;
      for (int a = 0; a < NL; a++) {
        double ph, dr[3];
        basis3<ET>(a, pt, ph, dr);
        uq[0] += ph * u[a][0]; uq[1] += ph * u[a][1]; uq[2] += ph * u[a][2]; pq += ph * p[a];
      }
      au += adet * pt[3] * (uq[0] * uq[0] + uq[1] * uq[1] + uq[2] * uq[2]);
      ap += adet * pt[3] * pq * pq;
    }
  }
  au = block_sum3(au, sh);
  ap = block_sum3(ap, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = au; partial[gridDim.x + blockIdx.x] = ap; }
}
// kind 7: flux int u.n over the facets with the given marker (outward normal)
template <int ET>
__global__ __launch_bounds__(TPB) void gen3_flux_kernel(int nfac, int marker, int nvo, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                        const int *__restrict__ fmarker, const int *__restrict__ cells, const unsigned char *__restrict__ cell_owned,
                                                        const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  constexpr int NL = g3_nloc(ET), NV = ET == 2 ? 8 : 4;
  __shared__ double sh[4];
  double a0 = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker || !cell_owned[fcell[k]]) continue;
    const int e = fcell[k], f = flocal[k];
    double X[NV][3], u[NL][3];
    for (int a = 0; a < NL; a++) {
      const int v = cells[(size_t)e * NL + a];
      if (a < NV) for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)v + i];
      for (int i = 0; i < 3; i++) u[a][i] = x[g3uo(v, nvo) + i];
    }
    double n[3], area;
    facet_geom3<ET>(X, f, n, area);
    for (int q = 0; q < facet_nq<ET>(); q++) {
      double pt[3], w;
      facet_point<ET>(f, q, pt, w);
      double un = 0;
      for (int a = 0; a < NL; a++) {
        double ph, dr[3];
        basis3<ET>(a, pt, ph, dr);
        un += ph * (u[a][0] * n[0] + u[a][1] * n[1] + u[a][2] * n[2]);
      }
      a0 += area * w * un;
    }
  }
  a0 = block_sum3(a0, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = a0; partial[gridDim.x + blockIdx.x] = 0.0; }
}
// wall shear stress (solverBase.py:163-195): (1/|f|) oint w . (T - (T.n) n), T = -2 mu eps(u) n, per facet node
template <int ET>
__global__ __launch_bounds__(TPB) void gen3_wss_kernel(int nfac, int nvo, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                       const int *__restrict__ cells, const double *__restrict__ coords, const double *__restrict__ x,
                                                       double mu, double *__restrict__ out) {
  constexpr int NL = g3_nloc(ET), NV = ET == 2 ? 8 : 4;
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nfac) return;
  const int e = fcell[k], f = flocal[k];
  int vs[NL];
  double X[NV][3], u[NL][3];
  for (int a = 0; a < NL; a++) {
    vs[a] = cells[(size_t)e * NL + a];
    if (a < NV) for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i];
    for (int i = 0; i < 3; i++) u[a][i] = x[g3uo(vs[a], nvo) + i];
  }
  double Ji[3][3], adet, h, n[3], area;
  geom3<ET>(X, Ji, adet, h);
  facet_geom3<ET>(X, f, n, area);
  double acc[NL][3];
  for (int a = 0; a < NL; a++) acc[a][0] = acc[a][1] = acc[a][2] = 0.0;
  for (int q = 0; q < facet_nq<ET>(); q++) {
    double pt[3], w, phv[NL];
    facet_point<ET>(f, q, pt, w);
    double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int a = 0; a < NL; a++) {
      double dr[3];
      basis3<ET>(a, pt, phv[a], dr);
      for (int i = 0; i < 3; i++) {
        const double gi = dr[0] * Ji[0][i] + dr[1] * Ji[1][i] + dr[2] * Ji[2][i];
        for (int j = 0; j < 3; j++) G[i][j] += gi * u[a][j];
      }
    }
    double T[3], Tn = 0;
    for (int i = 0; i < 3; i++) {
      T[i] = 0.0;
      for (int j = 0; j < 3; j++) T[i] -= mu * (G[i][j] + G[j][i]) * n[j];
      Tn += T[i] * n[i];
    }
    for (int a = 0; a < NL; a++)
      for (int i = 0; i < 3; i++) acc[a][i] += w * phv[a] * (T[i] - Tn * n[i]);
  }
  for (int a = 0; a < NL; a++)
    if (acc[a][0] != 0.0 || acc[a][1] != 0.0 || acc[a][2] != 0.0)
      for (int i = 0; i < 3; i++) atomicAdd(out + 3 * (size_t)vs[a] + i, acc[a][i]);
}

inline uint32_t part1by2(uint32_t x) {
  x &= 0x000003ff;
  x = (x ^ (x << 16)) & 0xff0000ff;
  x = (x ^ (x << 8)) & 0x0300f00f;
  x = (x ^ (x << 4)) & 0x030c30c3;
  x = (x ^ (x << 2)) & 0x09249249;
  return x;
}

}  // namespace

int kg3_upload_tables(cfdh_ctx *c) {
  static double pts[2][G3_NQ][4], tri[CFDH_NQ][4], g2[2][2];
  for (int q = 0; q < G3_NQ; q++) {
    if (q < CFDH3_NQ) { pts[0][q][0] = CFDH3_QL[q][1]; pts[0][q][1] = CFDH3_QL[q][2]; pts[0][q][2] = CFDH3_QL[q][3]; pts[0][q][3] = CFDH3_QW[q] / 6.0; }
    const int i = q / 49, j = (q / 7) % 7, k = q % 7;
    pts[1][q][0] = CFDH_GL7_X[i]; pts[1][q][1] = CFDH_GL7_X[j]; pts[1][q][2] = CFDH_GL7_X[k]; pts[1][q][3] = CFDH_GL7_W[i] * CFDH_GL7_W[j] * CFDH_GL7_W[k];
  }
  for (int q = 0; q < CFDH_NQ; q++) { tri[q][0] = CFDH_QL[q][0]; tri[q][1] = CFDH_QL[q][1]; tri[q][2] = CFDH_QL[q][2]; tri[q][3] = CFDH_QW[q]; }
  for (int q = 0; q < 2; q++) { g2[0][q] = CFDH_GL2_X[q]; g2[1][q] = CFDH_GL2_W[q]; }
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_pts), pts, sizeof pts));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_tri), tri, sizeof tri));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_gl2), g2, sizeof g2));
  return 0;
}

// nodes of local facet f of a 3-D cell (host): hexahedron 4, P2 tetrahedron 3 vertices + 3 edge nodes, P1 tetrahedron 3
int cfdh_facet_nodes3(const cfdh_ctx *c, int f, int out[8]) {
  const int et = c->etype;
  const int nfv = facet_nvert(et);
  for (int k = 0; k < nfv; k++) out[k] = facet_vertex(et, f, k);
  if (et != 1) return nfv;
  int n = 3;
  for (int e = 0; e < 6; e++) {
    int i, j;
    tet_edge(e, i, j);
    if (i != f && j != f) out[n++] = 4 + e;
  }
  return n;
}

// Mesh upload for the 3-D generic element path (the 3-D counterpart of cfdh_build_mesh_gen): Morton numbering of the nodes,
// node graph, staging order of the element blocks, stiffness / diagonal mass of the element on the graph (preconditioner),
// the P1 subspace of a P2 space (p-multigrid step), state and work vectors.
int cfdh_build_mesh_gen3(cfdh_ctx *c, int etype, int64_t nv64, int64_t nvo64, int64_t nc64, const int32_t *cells, const double *coords, int64_t nfac64,
                         const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker) {
  const int nv = (int)nv64, nvo = (int)nvo64, nc = (int)nc64, nfac = (int)nfac64;
  if (nvo <= 0 || nvo > nv) return cfdh_fail(c, CFDH_E_ARG, "bad owned node count");
  const int et = etype == 3 ? 0 : etype;
  const int NL = g3_nloc(et), NF = et == 2 ? 6 : 4, NV = et == 2 ? 8 : 4;
  if (nv <= 0 || nc <= 0) return cfdh_fail(c, CFDH_E_ARG, "bad mesh sizes");
  if (nv64 > (1ll << 28) || nc64 > (1ll << 24)) return cfdh_fail(c, CFDH_E_ARG, "mesh too large for int32 indexing of the staged element blocks");
  for (int64_t k = 0; k < (int64_t)NL * nc; k++)
    if (cells[k] < 0 || cells[k] >= nv) return cfdh_fail(c, CFDH_E_ARG, "cell node index out of range");
  for (int k = 0; k < nfac; k++)
    if (fcell[k] < 0 || fcell[k] >= nc || flocal[k] < 0 || flocal[k] >= NF) return cfdh_fail(c, CFDH_E_ARG, "facet (cell, local) out of range");
  c->dim = 3;
  c->etype = et; c->nloc = NL; c->gen = true;
  // partitioned runs: nodes [0, nvo) owned, ghosts after (halo-plan order); rows for owned nodes only; ghost tail of 4 doubles per node
  c->nv = nv; c->nvo = nvo; c->ng = nv - nvo;
  c->NO = 4 * nvo; c->NL = 4 * nvo + 4 * c->ng;
  c->perm.resize(nv); c->iperm.resize(nv);
  {
    std::vector<int> order(nvo);
    std::iota(order.begin(), order.end(), 0);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < nv; v++)
      for (int i = 0; i < 3; i++) { lo[i] = std::min(lo[i], coords[3 * v + i]); hi[i] = std::max(hi[i], coords[3 * v + i]); }
    const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0)) return cfdh_fail(c, CFDH_E_ARG, "degenerate coordinates");
    std::vector<uint32_t> key(nvo);
    for (int v = 0; v < nvo; v++) {
      uint32_t qd[3];
      for (int i = 0; i < 3; i++) qd[i] = (uint32_t)std::min(1023.0, (coords[3 * v + i] - lo[i]) / ext * 1023.0);
      key[v] = part1by2(qd[0]) | (part1by2(qd[1]) << 1) | (part1by2(qd[2]) << 2);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int k = 0; k < nvo; k++) { c->iperm[k] = order[k]; c->perm[order[k]] = k; }
    for (int v = nvo; v < nv; v++) { c->iperm[v] = v; c->perm[v] = v; }
  }
  c->h_coords.resize(3 * (size_t)nv);
  for (int k = 0; k < nv; k++) for (int i = 0; i < 3; i++) c->h_coords[3 * (size_t)k + i] = coords[3 * (size_t)c->iperm[k] + i];
  c->nc = nc;
  c->h_cells.resize((size_t)NL * nc);
  c->cell_user.resize(nc);
  const double *X = c->h_coords.data();
  auto P3 = [&](int v, int i) { return X[3 * (size_t)v + i]; };
  for (int e = 0; e < nc; e++) {
    c->cell_user[e] = e;
    for (int a = 0; a < NL; a++) c->h_cells[(size_t)NL * e + a] = c->perm[cells[(size_t)NL * e + a]];
    const int *v = &c->h_cells[(size_t)NL * e];
    double Xe[8][3];
    for (int a = 0; a < NV; a++) for (int i = 0; i < 3; i++) Xe[a][i] = P3(v[a], i);
    double Ji[3][3], adet, h;
    if (et == 2) geom3<2>(Xe, Ji, adet, h); else geom3<0>(Xe, Ji, adet, h);
    if (!(adet > 0) || !std::isfinite(adet)) return cfdh_fail(c, CFDH_E_ARG, "zero-volume cell %d", e);
    const double tol = 1e-9 * std::cbrt(adet);
    if (et == 2)  // parallelepipeds only (affine map): x_v = x_0 + i (x_1 - x_0) + j (x_2 - x_0) + k (x_4 - x_0)
      for (int a = 0; a < 8; a++)
        for (int i = 0; i < 3; i++) {
          const double ex = Xe[0][i] + (a & 1) * (Xe[1][i] - Xe[0][i]) + ((a >> 1) & 1) * (Xe[2][i] - Xe[0][i]) + ((a >> 2) & 1) * (Xe[4][i] - Xe[0][i]);
          if (std::fabs(Xe[a][i] - ex) > tol) return cfdh_fail(c, CFDH_E_ARG, "hexahedron %d is not a parallelepiped: only affine Q1 cells are supported", e);
        }
    if (et == 1)  // straight-sided P2: edge nodes at the edge midpoints
      for (int q = 0; q < 6; q++) {
        int i, j;
        tet_edge(q, i, j);
        for (int d = 0; d < 3; d++)
          if (std::fabs(0.5 * (P3(v[i], d) + P3(v[j], d)) - P3(v[4 + q], d)) > tol)
            return cfdh_fail(c, CFDH_E_ARG, "P2 cell %d: edge node %d is not the edge midpoint (curved cells are not supported)", e, q);
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
  c->h_vcol.clear(); c->h_vcol.reserve((size_t)30 * nvo);
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
  // ---- staging order, stiffness and diagonal mass on the graph
  std::vector<int> slot((size_t)nc * NL * NL);
  c->h_Lval.assign(c->nnzv, 0.0);
  c->h_Ml.assign(nv, 0.0);
  double msum = 0.0, dsum = 0.0;
  std::vector<double> mdiag(nv, 0.0);
  static double pts[2][G3_NQ][4];
  for (int q = 0; q < G3_NQ; q++) {
    if (q < CFDH3_NQ) { pts[0][q][0] = CFDH3_QL[q][1]; pts[0][q][1] = CFDH3_QL[q][2]; pts[0][q][2] = CFDH3_QL[q][3]; pts[0][q][3] = CFDH3_QW[q] / 6.0; }
    const int i = q / 49, j = (q / 7) % 7, k = q % 7;
    pts[1][q][0] = CFDH_GL7_X[i]; pts[1][q][1] = CFDH_GL7_X[j]; pts[1][q][2] = CFDH_GL7_X[k]; pts[1][q][3] = CFDH_GL7_W[i] * CFDH_GL7_W[j] * CFDH_GL7_W[k];
  }
  // reference stiffness-like integrals are cell dependent only through Ji: K_ab = |det| sum_q w_q (Ji^T dphi_a) . (Ji^T dphi_b)
  const int nq = g3_nq(et);
  std::vector<double> rphi((size_t)G3_NQ * NL), rd((size_t)G3_NQ * NL * 3);
  for (int q = 0; q < nq; q++)
    for (int a = 0; a < NL; a++) {
      double ph, dr[3];
      const double *pt = pts[et == 2 ? 1 : 0][q];
      if (et == 2) basis3<2>(a, pt, ph, dr); else if (et == 1) basis3<1>(a, pt, ph, dr); else basis3<0>(a, pt, ph, dr);
      rphi[(size_t)q * NL + a] = ph;
      for (int i = 0; i < 3; i++) rd[((size_t)q * NL + a) * 3 + i] = dr[i];
    }
  // reference matrices: Mref_a = sum_q w phi_a^2 ; Dref[a][b][k][l] = sum_q w dphi_a[k] dphi_b[l]  ->  K_ab = |det| sum_kl Dref (Ji Ji^T)[k][l]
  std::vector<double> Mref(NL, 0.0), Dref((size_t)NL * NL * 9, 0.0);
  for (int q = 0; q < nq; q++) {
    const double w = pts[et == 2 ? 1 : 0][q][3];
    for (int a = 0; a < NL; a++) {
      Mref[a] += w * rphi[(size_t)q * NL + a] * rphi[(size_t)q * NL + a];
      for (int b = 0; b < NL; b++)
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) Dref[(((size_t)a * NL + b) * 3 + k) * 3 + l] += w * rd[((size_t)q * NL + a) * 3 + k] * rd[((size_t)q * NL + b) * 3 + l];
    }
  }
  for (int e = 0; e < nc; e++) {
    const int *v = &c->h_cells[(size_t)NL * e];
    double Xe[8][3];
    for (int a = 0; a < NV; a++) for (int i = 0; i < 3; i++) Xe[a][i] = P3(v[a], i);
    double Ji[3][3], adet, h, M[3][3];
    if (et == 2) geom3<2>(Xe, Ji, adet, h); else geom3<0>(Xe, Ji, adet, h);
    for (int k = 0; k < 3; k++)
      for (int l = 0; l < 3; l++) M[k][l] = Ji[k][0] * Ji[l][0] + Ji[k][1] * Ji[l][1] + Ji[k][2] * Ji[l][2];
    msum += adet * (et == 2 ? 1.0 : 1.0 / 6.0);
    for (int a = 0; a < NL; a++) {
      mdiag[v[a]] += adet * Mref[a];
      dsum += adet * Mref[a];
      if (v[a] >= nvo) {  // row of a ghost node: assembled by its owner
        for (int b = 0; b < NL; b++) slot[((size_t)e * NL + a) * NL + b] = -1;
        continue;
      }
      const int *nb = &c->h_vcol[c->h_vptr[v[a]]];
      const int deg = c->h_vptr[v[a] + 1] - c->h_vptr[v[a]];
      for (int b = 0; b < NL; b++) {
        const int k = c->h_vptr[v[a]] + (int)(std::lower_bound(nb, nb + deg, v[b]) - nb);
        slot[((size_t)e * NL + a) * NL + b] = k;
        double Kab = 0.0;
        for (int kk = 0; kk < 3; kk++)
          for (int l = 0; l < 3; l++) Kab += Dref[(((size_t)a * NL + b) * 3 + kk) * 3 + l] * M[kk][l];
        c->h_Lval[k] += adet * Kab;
      }
    }
  }
  if (et == 1) {
    // P1 subspace of the P2 space (first coarse level of both hierarchies): vertex nodes in order of first appearance, an edge
    // node interpolates its two end vertices
    std::vector<int> vid(nv, -1), ea(nv, -1), eb(nv, -1);
    for (int e = 0; e < nc; e++) {
      const int *v = &c->h_cells[(size_t)NL * e];
      for (int q = 0; q < 4; q++) vid[v[q]] = 0;
      for (int q = 0; q < 6; q++) { int i, j; tet_edge(q, i, j); ea[v[4 + q]] = v[i]; eb[v[4 + q]] = v[j]; }
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
  // staging order of the assembly: contributions to one block entry / one node adjacent, in (cell, lane) order
  std::vector<int> eptr((size_t)c->nnzv + 1, 0), fptr((size_t)nvo + 1, 0), fdst((size_t)nc * NL * NL, -1);
  for (size_t t = 0; t < slot.size(); t++) if (slot[t] >= 0) eptr[slot[t] + 1]++;
  for (int k = 0; k < c->nnzv; k++) eptr[k + 1] += eptr[k];
  {
    std::vector<int> fill(eptr.begin(), eptr.end() - 1);
    for (size_t t = 0; t < slot.size(); t++) if (slot[t] >= 0) slot[t] = fill[slot[t]]++;
  }
  for (int e = 0; e < nc; e++)
    for (int a = 0; a < NL; a++) if (c->h_cells[(size_t)NL * e + a] < nvo) fptr[c->h_cells[(size_t)NL * e + a] + 1] += NL;
  for (int v = 0; v < nvo; v++) fptr[v + 1] += fptr[v];
  {
    std::vector<int> fill(fptr.begin(), fptr.end() - 1);
    for (int e = 0; e < nc; e++)
      for (int a = 0; a < NL; a++)
        if (c->h_cells[(size_t)NL * e + a] < nvo)
          for (int b = 0; b < NL; b++) fdst[((size_t)e * NL + a) * NL + b] = fill[c->h_cells[(size_t)NL * e + a]]++;
  }
  for (int v = 0; v < nv; v++) c->h_Ml[v] = mdiag[v] * (msum / dsum);
  // ---- uploads and allocations
  hipStream_t s = c->stream;
  std::vector<unsigned short> gflag(nc, 0);
  for (int k = 0; k < nfac; k++) gflag[fcell[k]] |= (unsigned short)(1u << flocal[k]);
  HIPCHK(c, c->coords.upload(c->h_coords, s));
  HIPCHK(c, c->cells.upload(c->h_cells, s));
  HIPCHK(c, c->gflag.upload(gflag, s));
  {
    std::vector<int> fcl;
    for (int e = 0; e < nc; e++) if (gflag[e]) fcl.push_back(e);
    c->g3_nfcells = (int)fcl.size();
    if (c->g3_nfcells) HIPCHK(c, c->g3_fcells.upload(fcl, s));
  }
  HIPCHK(c, c->gslot.upload(slot, s));
  HIPCHK(c, c->g_eptr.upload(eptr, s)); HIPCHK(c, c->g_fptr.upload(fptr, s)); HIPCHK(c, c->g_fdst.upload(fdst, s));
  HIPCHK(c, c->gE.alloc(16 * (size_t)nc * NL * NL)); HIPCHK(c, c->gEF.alloc(4 * (size_t)nc * NL * NL));
  HIPCHK(c, c->vptr.upload(c->h_vptr, s));
  HIPCHK(c, c->vcol.upload(c->h_vcol, s));
  HIPCHK(c, c->vdiag.upload(c->h_vdiag, s));
  HIPCHK(c, c->A00.alloc(9 * (size_t)c->nnzv));
  HIPCHK(c, c->A01.alloc(3 * (size_t)c->nnzv));
  HIPCHK(c, c->A10.alloc(3 * (size_t)c->nnzv));
  HIPCHK(c, c->A11.alloc((size_t)c->nnzv));
  std::vector<unsigned char> cown(nc, 1);
  for (int e = 0; e < nc; e++) cown[e] = cells[(size_t)NL * e] < nvo ? 1 : 0;
  HIPCHK(c, c->cell_owned.upload(cown, s));
  std::vector<double> rnd(3 * (size_t)nv);
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
  c->h_bcval.assign(4 * (size_t)nv, 0.0);
  c->h_bcmult.assign(4 * (size_t)nv, 0.0);
  HIPCHK(c, c->bcflag.alloc(nv));
  HIPCHK(c, c->bcval.alloc(4 * (size_t)nv));
  HIPCHK(c, c->bcmult.alloc(4 * (size_t)nv));
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
  HIPCHK(c, c->dinvA.alloc(3 * (size_t)nv));
  HIPCHK(c, c->pu0.alloc(3 * (size_t)nv)); HIPCHK(c, c->pu1.alloc(3 * (size_t)nv)); HIPCHK(c, c->pu2.alloc(3 * (size_t)nv));
  HIPCHK(c, c->pr.alloc(3 * (size_t)nv));
  HIPCHK(c, c->pp0.alloc(nv)); HIPCHK(c, c->pp1.alloc(nv));
  c->mom_valid = true;  // no tau-moment pass: tau is evaluated inside the quadrature loop
  HIPCHK(c, hipStreamSynchronize(s));
  return 0;
}

// stiffness K [nloc][nloc] of one 3-D cell of the context's element type (global pressure Laplacian of a partitioned run)
int cfdh_gen3_element_stiffness(const cfdh_ctx *c, const int32_t *v, const double *X, double *K) {
  const int et = c->etype, NL = c->nloc, NV = et == 2 ? 8 : 4;
  double Xe[8][3], Ji[3][3], adet, h;
  for (int a = 0; a < NV; a++) for (int i = 0; i < 3; i++) Xe[a][i] = X[3 * (size_t)v[a] + i];
  if (et == 2) geom3<2>(Xe, Ji, adet, h); else geom3<0>(Xe, Ji, adet, h);
  if (!(adet > 0)) return CFDH_E_ARG;
  for (int k = 0; k < NL * NL; k++) K[k] = 0.0;
  for (int q = 0; q < g3_nq(et); q++) {
    double pt[3], w, g[10][3];
    if (et == 2) {
      const int i = q / 49, j = (q / 7) % 7, k = q % 7;
      pt[0] = CFDH_GL7_X[i]; pt[1] = CFDH_GL7_X[j]; pt[2] = CFDH_GL7_X[k]; w = CFDH_GL7_W[i] * CFDH_GL7_W[j] * CFDH_GL7_W[k];
    } else { pt[0] = CFDH3_QL[q][1]; pt[1] = CFDH3_QL[q][2]; pt[2] = CFDH3_QL[q][3]; w = CFDH3_QW[q] / 6.0; }
    for (int a = 0; a < NL; a++) {
      double ph, dr[3];
      if (et == 2) basis3<2>(a, pt, ph, dr); else if (et == 1) basis3<1>(a, pt, ph, dr); else basis3<0>(a, pt, ph, dr);
      for (int i = 0; i < 3; i++) g[a][i] = dr[0] * Ji[0][i] + dr[1] * Ji[1][i] + dr[2] * Ji[2][i];
    }
    for (int a = 0; a < NL; a++)
      for (int b = 0; b < NL; b++) K[a * NL + b] += adet * w * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
  }
  return 0;
}

int kg3_assemble(cfdh_ctx *c, const double *xstate, int mode) {
  if (mode == 0) mode = 2;
  Gen3Args P;
  P.nc = c->nc; P.nvo = c->nvo; P.mode = mode;
  P.cells = c->cells.p; P.coords = c->coords.p; P.slot = c->gslot.p; P.fdst = c->g_fdst.p; P.flag = c->gflag.p;
  P.x = xstate; P.xprev = c->xprev.p; P.xprev2 = c->xprev2.p;
  P.bcflag = c->bcflag.p; P.bcval = c->bcval.p;
  P.dt = c->dt; P.rho = c->rho; P.mu = c->mu; P.muf = c->muf; P.f[0] = c->f[0]; P.f[1] = c->f[1]; P.f[2] = c->f[2];
  P.theta = c->ts_theta; P.a0 = c->ts_a[0]; P.a1 = c->ts_a[1]; P.a2 = c->ts_a[2];
  P.beta = c->bf_marker >= 0 ? c->bf_beta : 0.0;
  P.ds_terms = c->ds_terms ? 1 : 0;
  P.E = c->gE.p; P.EF = c->gEF.p;
  const dim3 grid(c->nc);
  prof_begin(c, 0);
#define G3_LAUNCH(ET) do { if (mode == 1) hipLaunchKernelGGL((gen3_asm_kernel<ET, true>), grid, dim3(g3_wgs(ET)), 0, c->stream, P); \
                           else hipLaunchKernelGGL((gen3_asm_kernel<ET, false>), grid, dim3(g3_wgs(ET)), 0, c->stream, P); } while (0)
  if (c->etype == 1) G3_LAUNCH(1);
  else if (c->etype == 2) G3_LAUNCH(2);
  else G3_LAUNCH(0);
#undef G3_LAUNCH
  if (c->g3_nfcells > 0 && (P.ds_terms || P.beta != 0.0)) {
    const dim3 fg(c->g3_nfcells), fb(128);
    if (c->etype == 1) hipLaunchKernelGGL((gen3_facet_kernel<1>), fg, fb, 0, c->stream, P, c->g3_fcells.p);
    else if (c->etype == 2) hipLaunchKernelGGL((gen3_facet_kernel<2>), fg, fb, 0, c->stream, P, c->g3_fcells.p);
    else hipLaunchKernelGGL((gen3_facet_kernel<0>), fg, fb, 0, c->stream, P, c->g3_fcells.p);
  }
  const dim3 block(TPB);
  hipLaunchKernelGGL(gen3_gather_F_kernel, dim3((c->nvo + TPB - 1) / TPB), block, 0, c->stream, c->nvo, c->g_fptr.p, c->gEF.p, c->F.p);
  if (mode == 1)
    hipLaunchKernelGGL(gen3_gather_J_kernel, dim3((c->nnzv + TPB - 1) / TPB), block, 0, c->stream, c->nnzv, c->g_eptr.p, c->gE.p, c->A00.p, c->A01.p,
                       c->A10.p, c->A11.p);
  hipLaunchKernelGGL(gen3_bc_rows_kernel, dim3((c->nvo + TPB - 1) / TPB), block, 0, c->stream, c->nvo, mode, c->bcflag.p, c->bcval.p, c->bcmult.p,
                     c->vdiag.p, xstate, c->F.p, c->A00.p, c->A11.p);
  prof_end(c, 0);
  HIPCHK(c, hipGetLastError());
  if (mode == 1) c->jac_valid = true;
  return 0;
}

// per-block partial sums of a functional into red_partial ([nb] first value, [nb] second value): kinds 2 / 3 (L2 norms), 7 (flux)
int kg3_functional_partials(cfdh_ctx *c, int kind, int marker, int nb) {
  const dim3 grid(nb), block(TPB);
#define ET3_DISPATCH(KERNEL, ...)                                                                       \
  do {                                                                                                  \
    if (c->etype == 1) hipLaunchKernelGGL((KERNEL<1>), grid, block, 0, c->stream, __VA_ARGS__);         \
    else if (c->etype == 2) hipLaunchKernelGGL((KERNEL<2>), grid, block, 0, c->stream, __VA_ARGS__);    \
    else hipLaunchKernelGGL((KERNEL<0>), grid, block, 0, c->stream, __VA_ARGS__);                       \
  } while (0)
  if (kind == 2 || kind == 3) ET3_DISPATCH(gen3_l2_kernel, c->nc, c->nvo, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
  else if (kind == 7) ET3_DISPATCH(gen3_flux_kernel, c->nfac, marker, c->nvo, c->d_fac_cell.p, c->d_fac_local.p, c->d_fac_marker.p, c->cells.p,
                                   c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
  else return cfdh_fail(c, CFDH_E_ARG, "functional kind %d is not available for 3-D P2 / Q1 contexts", kind);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int kg3_wss(cfdh_ctx *c, double *out) {
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(double) * 3 * (size_t)c->nv, c->stream));
  if (c->nfac > 0) {
    const dim3 grid((c->nfac + TPB - 1) / TPB), block(TPB);
    ET3_DISPATCH(gen3_wss_kernel, c->nfac, c->nvo, c->d_fac_cell.p, c->d_fac_local.p, c->cells.p, c->coords.p, c->x.p, c->mu, out);
  }
#undef ET3_DISPATCH
  HIPCHK(c, hipGetLastError());
  return 0;
}
