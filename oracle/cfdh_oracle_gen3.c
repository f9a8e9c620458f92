/* CPU oracle for nodal equal-order elements in 3-D beyond P1 tetrahedra: Q1/Q1 hexahedra (parallelepipeds) and P2/P2
 * tetrahedra (SURVEY.md section 8f-4, 3-D half)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see cfdh_oracle.c).
 *
 * Plain-C restatement, with scalar loops over cells / quadrature points / local nodes, of the element residual and its exact
 * derivative as oracle/np_twin_gen3.py states them (which on P1 tetrahedra equals the closed-form twin np_twin_nd.py to round-off):
 *   residual          /root/reference/src/solvers/stabilized_schur.py:67-123
 *   backflow variant  /root/reference/src/solvers/stabilized_schur_backflow.py:84-87 (p_grade), :107 (no ds pair), :158-176
 *   hexahedra         /root/reference/src/scenarios/unit_cube_pipe.py:103-109
 * Local orders, quadrature and conventions: header of np_twin_gen3.py.  Element types: 3 P1 tetrahedron (cross-check),
 * 4 P2 tetrahedron, 5 Q1 hexahedron.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "cfdh_quad_gl.h"
#include "cfdh_quad_tet.h"
#include "cfdh_quad_tri.h"

#define MAXL 10
#define MAXQ 343
#define MAXQF 49
#define EPS_VNORM 1e-15

typedef struct {
  double dt, rho, mu, muf, f[3], theta, a0, a1, a2, beta;
  int32_t ds_terms, pad;
} orcg3_params;

static const int TET_EDGES[6][2] = {{2, 3}, {1, 3}, {1, 2}, {0, 3}, {0, 2}, {0, 1}};
static const int HEX_FACETS[6][4] = {{0, 1, 2, 3}, {0, 1, 4, 5}, {0, 2, 4, 6}, {1, 3, 5, 7}, {2, 3, 6, 7}, {4, 5, 6, 7}};
static const int TET_FACETS[4][3] = {{1, 2, 3}, {0, 2, 3}, {0, 1, 3}, {0, 1, 2}};
static const double TET_DL[4][3] = {{-1, -1, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};

static int nloc_of(int et) { return et == 3 ? 4 : (et == 4 ? 10 : 8); }

/* basis values, reference gradients and reference Hessians at a reference point */
static void tabulate(int et, const double *pt, double *phi, double (*d)[3], double (*H)[3][3]) {
  const double x = pt[0], y = pt[1], z = pt[2];
  memset(H, 0, sizeof(double) * MAXL * 9);
  if (et == 5) {
    const double f[3][2] = {{1 - x, x}, {1 - y, y}, {1 - z, z}};
    static const double df[2] = {-1.0, 1.0};
    for (int v = 0; v < 8; v++) {
      const int i = v & 1, j = (v >> 1) & 1, k = (v >> 2) & 1;
      phi[v] = f[0][i] * f[1][j] * f[2][k];
      d[v][0] = df[i] * f[1][j] * f[2][k];
      d[v][1] = f[0][i] * df[j] * f[2][k];
      d[v][2] = f[0][i] * f[1][j] * df[k];
      H[v][0][1] = H[v][1][0] = df[i] * df[j] * f[2][k];
      H[v][0][2] = H[v][2][0] = df[i] * f[1][j] * df[k];
      H[v][1][2] = H[v][2][1] = f[0][i] * df[j] * df[k];
    }
    return;
  }
  const double l[4] = {1.0 - x - y - z, x, y, z};
  if (et == 3) {
    for (int a = 0; a < 4; a++) { phi[a] = l[a]; for (int k = 0; k < 3; k++) d[a][k] = TET_DL[a][k]; }
    return;
  }
  for (int a = 0; a < 4; a++) {
    phi[a] = l[a] * (2.0 * l[a] - 1.0);
    for (int k = 0; k < 3; k++) {
      d[a][k] = (4.0 * l[a] - 1.0) * TET_DL[a][k];
      for (int m = 0; m < 3; m++) H[a][k][m] = 4.0 * TET_DL[a][k] * TET_DL[a][m];
    }
  }
  for (int e = 0; e < 6; e++) {
    const int i = TET_EDGES[e][0], j = TET_EDGES[e][1];
    phi[4 + e] = 4.0 * l[i] * l[j];
    for (int k = 0; k < 3; k++) {
      d[4 + e][k] = 4.0 * (l[i] * TET_DL[j][k] + l[j] * TET_DL[i][k]);
      for (int m = 0; m < 3; m++) H[4 + e][k][m] = 4.0 * (TET_DL[i][k] * TET_DL[j][m] + TET_DL[j][k] * TET_DL[i][m]);
    }
  }
}

static void ref_vertex(int et, int v, double *r) {
  if (et == 5) { r[0] = v & 1; r[1] = (v >> 1) & 1; r[2] = (v >> 2) & 1; }
  else { r[0] = v == 1; r[1] = v == 2; r[2] = v == 3; }
}

static void tau_pair(double s, double h, const orcg3_params *P, double *tau, double *tauL) {
  const double nu = P->mu / P->rho;
  double t1 = 4.0 * s;
  if (t1 < EPS_VNORM * EPS_VNORM) t1 = EPS_VNORM * EPS_VNORM;
  t1 /= h * h;
  *tau = 1.0 / sqrt(t1 + 4.0 / (P->dt * P->dt) + 16.0 * nu * nu / (h * h * h * h));
  const double vn = sqrt(s), Re = vn * h / (2.0 * nu), z = Re <= 3.0 ? Re / 3.0 : 1.0;
  *tauL = vn * h * z / 2.0;
}

/* facet rule: reference points of local facet f (in cell reference coordinates) and weights summing to 1 */
static int facet_rule(int et, int f, double (*pts)[3], double *w) {
  if (et == 5) {
    double rv[4][3];
    for (int k = 0; k < 4; k++) ref_vertex(et, HEX_FACETS[f][k], rv[k]);
    int n = 0;
    for (int a = 0; a < 2; a++)
      for (int b = 0; b < 2; b++, n++) {
        const double s = CFDH_GL2_X[a], t = CFDH_GL2_X[b];
        for (int i = 0; i < 3; i++) pts[n][i] = (1 - s) * (1 - t) * rv[0][i] + s * (1 - t) * rv[1][i] + (1 - s) * t * rv[2][i] + s * t * rv[3][i];
        w[n] = CFDH_GL2_W[a] * CFDH_GL2_W[b];
      }
    return 4;
  }
  double rv[3][3];
  for (int k = 0; k < 3; k++) ref_vertex(et, TET_FACETS[f][k], rv[k]);
  if (et == 3) { /* 6-point degree-3 rule of Strang and Fix (np_twin_nd.facet_rule(3)) */
    static const double A = 0.659027622374092, B = 0.231933368553031, C = 0.109039009072877;
    const double P6[6][3] = {{A, B, C}, {A, C, B}, {B, A, C}, {B, C, A}, {C, A, B}, {C, B, A}};
    for (int n = 0; n < 6; n++) {
      for (int i = 0; i < 3; i++) pts[n][i] = P6[n][0] * rv[0][i] + P6[n][1] * rv[1][i] + P6[n][2] * rv[2][i];
      w[n] = 1.0 / 6.0;
    }
    return 6;
  }
  for (int n = 0; n < CFDH_NQ; n++) {
    for (int i = 0; i < 3; i++) pts[n][i] = CFDH_QL[n][0] * rv[0][i] + CFDH_QL[n][1] * rv[1][i] + CFDH_QL[n][2] * rv[2][i];
    w[n] = CFDH_QW[n];
  }
  return CFDH_NQ;
}

/* Fe [nc][4 nloc], Je [nc][4 nloc][4 nloc] (row-major); flags: bit f exterior facet f, bit 8+f backflow facet f.
 * un2 may be NULL when a2 == 0; Je may be NULL when want_jac == 0. */
void orcg3_element_tensors(int et, int64_t nc, const int64_t *cells, const double *x, const double *u, const double *un, const double *un2,
                           const double *p, const orcg3_params *P, const uint16_t *flags, int want_jac, double *Fe, double *Je) {
  const int nl = nloc_of(et), nd = 4 * nl, nvert = et == 5 ? 8 : 4, nfac = et == 5 ? 6 : 4, PO = 3 * nl;
  const int nq = et == 5 ? 343 : CFDH3_NQ;
  /* tables */
  double (*phiq)[MAXL] = malloc(sizeof(double) * MAXQ * MAXL);
  double (*dphiq)[MAXL][3] = malloc(sizeof(double) * MAXQ * MAXL * 3);
  double (*hq)[MAXL][3][3] = malloc(sizeof(double) * MAXQ * MAXL * 9);
  double *wq = malloc(sizeof(double) * MAXQ);
  for (int q = 0; q < nq; q++) {
    double pt[3];
    if (et == 5) {
      const int i = q / 49, j = (q / 7) % 7, k = q % 7;
      pt[0] = CFDH_GL7_X[i]; pt[1] = CFDH_GL7_X[j]; pt[2] = CFDH_GL7_X[k];
      wq[q] = CFDH_GL7_W[i] * CFDH_GL7_W[j] * CFDH_GL7_W[k];
    } else {
      pt[0] = CFDH3_QL[q][1]; pt[1] = CFDH3_QL[q][2]; pt[2] = CFDH3_QL[q][3];
      wq[q] = CFDH3_QW[q] / 6.0;
    }
    tabulate(et, pt, phiq[q], dphiq[q], hq[q]);
  }
  const double rho = P->rho, mu = P->mu, th = P->theta, a0dt = P->a0 / P->dt;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < nc; c++) {
    const int64_t *cl = cells + c * nl;
    double X[MAXL][3], ub[MAXL][3], wn[MAXL][3], une[MAXL][3], pe[MAXL];
    for (int a = 0; a < nl; a++) {
      const int64_t v = cl[a];
      for (int i = 0; i < 3; i++) {
        X[a][i] = x[3 * v + i];
        une[a][i] = un[3 * v + i];
        ub[a][i] = th * u[3 * v + i] + (1.0 - th) * une[a][i];
        wn[a][i] = (P->a0 * u[3 * v + i] + P->a1 * une[a][i] + (P->a2 != 0.0 ? P->a2 * un2[3 * v + i] : 0.0)) / P->dt;
      }
      pe[a] = p[v];
    }
    /* affine map: columns x_1 - x_0, x_2 - x_0, x_3 - x_0 (tetrahedron) / x_4 - x_0 (hexahedron) */
    const int c3 = et == 5 ? 4 : 3;
    double Jm3[3][3], Ji[3][3];
    for (int i = 0; i < 3; i++) { Jm3[i][0] = X[1][i] - X[0][i]; Jm3[i][1] = X[2][i] - X[0][i]; Jm3[i][2] = X[c3][i] - X[0][i]; }
    const double det = Jm3[0][0] * (Jm3[1][1] * Jm3[2][2] - Jm3[1][2] * Jm3[2][1]) - Jm3[0][1] * (Jm3[1][0] * Jm3[2][2] - Jm3[1][2] * Jm3[2][0]) +
                       Jm3[0][2] * (Jm3[1][0] * Jm3[2][1] - Jm3[1][1] * Jm3[2][0]);
    const double adet = fabs(det);
    /* Ji[k][i] = d xi_k / d x_i = (J^-1)[k][i] */
    Ji[0][0] = (Jm3[1][1] * Jm3[2][2] - Jm3[1][2] * Jm3[2][1]) / det; Ji[0][1] = (Jm3[0][2] * Jm3[2][1] - Jm3[0][1] * Jm3[2][2]) / det; Ji[0][2] = (Jm3[0][1] * Jm3[1][2] - Jm3[0][2] * Jm3[1][1]) / det;
    Ji[1][0] = (Jm3[1][2] * Jm3[2][0] - Jm3[1][0] * Jm3[2][2]) / det; Ji[1][1] = (Jm3[0][0] * Jm3[2][2] - Jm3[0][2] * Jm3[2][0]) / det; Ji[1][2] = (Jm3[0][2] * Jm3[1][0] - Jm3[0][0] * Jm3[1][2]) / det;
    Ji[2][0] = (Jm3[1][0] * Jm3[2][1] - Jm3[1][1] * Jm3[2][0]) / det; Ji[2][1] = (Jm3[0][1] * Jm3[2][0] - Jm3[0][0] * Jm3[2][1]) / det; Ji[2][2] = (Jm3[0][0] * Jm3[1][1] - Jm3[0][1] * Jm3[1][0]) / det;
    double h = 0.0;
    for (int a = 0; a < nvert; a++)
      for (int b = a + 1; b < nvert; b++) {
        const double d = sqrt((X[a][0] - X[b][0]) * (X[a][0] - X[b][0]) + (X[a][1] - X[b][1]) * (X[a][1] - X[b][1]) + (X[a][2] - X[b][2]) * (X[a][2] - X[b][2]));
        if (d > h) h = d;
      }
    double *F = Fe + c * nd, *Jm = want_jac ? Je + c * (int64_t)nd * nd : NULL;
    for (int r = 0; r < nd; r++) F[r] = 0.0;
    if (Jm) for (int r = 0; r < nd * nd; r++) Jm[r] = 0.0;
    for (int q = 0; q < nq; q++) {
      const double dv = adet * wq[q];
      double g[MAXL][3], Hs[MAXL][3][3], lap[MAXL];
      const double *ph = phiq[q];
      for (int a = 0; a < nl; a++) {
        for (int i = 0; i < 3; i++) g[a][i] = dphiq[q][a][0] * Ji[0][i] + dphiq[q][a][1] * Ji[1][i] + dphiq[q][a][2] * Ji[2][i];
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) {
            double s = 0.0;
            for (int k = 0; k < 3; k++)
              for (int l = 0; l < 3; l++) s += hq[q][a][k][l] * Ji[k][i] * Ji[l][j];
            Hs[a][i][j] = s;
          }
        lap[a] = Hs[a][0][0] + Hs[a][1][1] + Hs[a][2][2];
      }
      double uq[3] = {0, 0, 0}, wv[3] = {0, 0, 0}, unq[3] = {0, 0, 0}, G[3][3] = {{0}}, gp[3] = {0, 0, 0}, visc[3] = {0, 0, 0}, pq = 0.0;
      for (int a = 0; a < nl; a++) {
        for (int i = 0; i < 3; i++) {
          uq[i] += ph[a] * ub[a][i]; wv[i] += ph[a] * wn[a][i]; unq[i] += ph[a] * une[a][i];
          gp[i] += g[a][i] * pe[a];
          for (int j = 0; j < 3; j++) G[i][j] += g[a][i] * ub[a][j];
          visc[i] += mu * (lap[a] * ub[a][i] + Hs[a][i][0] * ub[a][0] + Hs[a][i][1] * ub[a][1] + Hs[a][i][2] * ub[a][2]);
        }
        pq += ph[a] * pe[a];
      }
      const double divu = G[0][0] + G[1][1] + G[2][2];
      double C[3], R[3], bgr[MAXL], tau, tauL;
      for (int j = 0; j < 3; j++) C[j] = uq[0] * G[0][j] + uq[1] * G[1][j] + uq[2] * G[2][j];
      for (int i = 0; i < 3; i++) R[i] = rho * (wv[i] + C[i]) - visc[i] + gp[i] - rho * P->f[i];
      tau_pair(unq[0] * unq[0] + unq[1] * unq[1] + unq[2] * unq[2], h, P, &tau, &tauL);
      for (int a = 0; a < nl; a++) bgr[a] = uq[0] * g[a][0] + uq[1] * g[a][1] + uq[2] * g[a][2];
      for (int a = 0; a < nl; a++) {
        for (int i = 0; i < 3; i++) {
          double v = rho * ph[a] * (wv[i] + C[i] - P->f[i]);
          for (int j = 0; j < 3; j++) v += g[a][j] * mu * (G[i][j] + G[j][i]);
          v += -pq * g[a][i] + tau * R[i] * bgr[a] + tauL * rho * divu * g[a][i];
          F[3 * a + i] += dv * v;
        }
        F[PO + a] += dv * (ph[a] * divu + tau / rho * (R[0] * g[a][0] + R[1] * g[a][1] + R[2] * g[a][2]));
      }
      if (!Jm) continue;
      for (int b = 0; b < nl; b++)
        for (int j = 0; j < 3; j++) {
          double dR[3], dWC[3];
          for (int i = 0; i < 3; i++) {
            const double dij = i == j ? 1.0 : 0.0;
            dWC[i] = rho * (a0dt * ph[b] * dij + th * (ph[b] * G[j][i] + dij * bgr[b]));
            dR[i] = dWC[i] - mu * th * (lap[b] * dij + Hs[b][i][j]);
          }
          for (int a = 0; a < nl; a++) {
            const double gg = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
            for (int i = 0; i < 3; i++) {
              const double dij = i == j ? 1.0 : 0.0;
              const double v = ph[a] * dWC[i] + mu * th * (g[a][j] * g[b][i] + dij * gg) + tau * dR[i] * bgr[a] + th * tau * R[i] * ph[b] * g[a][j] +
                               rho * th * tauL * g[b][j] * g[a][i];
              Jm[(3 * a + i) * nd + 3 * b + j] += dv * v;
            }
            Jm[(PO + a) * nd + 3 * b + j] += dv * (th * ph[a] * g[b][j] + tau / rho * (dR[0] * g[a][0] + dR[1] * g[a][1] + dR[2] * g[a][2]));
          }
        }
      for (int b = 0; b < nl; b++)
        for (int a = 0; a < nl; a++) {
          for (int i = 0; i < 3; i++) Jm[(3 * a + i) * nd + PO + b] += dv * (-ph[b] * g[a][i] + tau * g[b][i] * bgr[a]);
          Jm[(PO + a) * nd + PO + b] += dv * tau / rho * (g[b][0] * g[a][0] + g[b][1] * g[a][1] + g[b][2] * g[a][2]);
        }
    }
    /* exterior-facet terms */
    const unsigned fl = flags ? flags[c] : 0u;
    if (!fl) continue;
    double cen[3] = {0, 0, 0};
    for (int a = 0; a < nvert; a++) for (int i = 0; i < 3; i++) cen[i] += X[a][i] / nvert;
    for (int f = 0; f < nfac; f++) {
      const int ext = P->ds_terms && ((fl >> f) & 1u), bfl = P->beta != 0.0 && ((fl >> (8 + f)) & 1u);
      if (!ext && !bfl) continue;
      const int *fv = et == 5 ? HEX_FACETS[f] : TET_FACETS[f];
      const int nfv = et == 5 ? 4 : 3;
      double e1[3], e2[3], n[3], fc[3] = {0, 0, 0};
      for (int i = 0; i < 3; i++) { e1[i] = X[fv[1]][i] - X[fv[0]][i]; e2[i] = X[fv[2]][i] - X[fv[0]][i]; }
      n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
      const double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      const double area = et == 5 ? nn : 0.5 * nn;
      for (int k = 0; k < nfv; k++) for (int i = 0; i < 3; i++) fc[i] += X[fv[k]][i] / nfv;
      double sgn = ((fc[0] - cen[0]) * n[0] + (fc[1] - cen[1]) * n[1] + (fc[2] - cen[2]) * n[2]) < 0 ? -1.0 : 1.0;
      for (int i = 0; i < 3; i++) n[i] *= sgn / nn;
      double fpts[MAXQF][3], fwt[MAXQF];
      const int nqf = facet_rule(et, f, fpts, fwt);
      for (int q = 0; q < nqf; q++) {
        const double m = area * fwt[q];
        double ph[MAXL], dr[MAXL][3], Hd[MAXL][3][3], g[MAXL][3];
        tabulate(et, fpts[q], ph, dr, Hd);
        for (int a = 0; a < nl; a++)
          for (int i = 0; i < 3; i++) g[a][i] = dr[a][0] * Ji[0][i] + dr[a][1] * Ji[1][i] + dr[a][2] * Ji[2][i];
        double uq[3] = {0, 0, 0}, G[3][3] = {{0}}, pq = 0.0, sn = 0.0;
        for (int a = 0; a < nl; a++) {
          for (int i = 0; i < 3; i++) {
            uq[i] += ph[a] * ub[a][i];
            sn += ph[a] * une[a][i] * n[i];
            for (int j = 0; j < 3; j++) G[i][j] += g[a][i] * ub[a][j];
          }
          pq += ph[a] * pe[a];
        }
        if (ext) {
          for (int a = 0; a < nl; a++)
            for (int i = 0; i < 3; i++) {
              F[3 * a + i] += m * ph[a] * (pq * n[i] - P->muf * (G[i][0] * n[0] + G[i][1] * n[1] + G[i][2] * n[2]));
              if (Jm)
                for (int b = 0; b < nl; b++) {
                  Jm[(3 * a + i) * nd + PO + b] += m * ph[a] * ph[b] * n[i];
                  for (int j = 0; j < 3; j++) Jm[(3 * a + i) * nd + 3 * b + j] -= P->muf * th * m * ph[a] * g[b][i] * n[j];
                }
            }
        }
        if (bfl) {
          const double cq = P->beta * rho * 0.5 * (sn - fabs(sn)) * m;
          for (int a = 0; a < nl; a++)
            for (int i = 0; i < 3; i++) {
              F[3 * a + i] -= cq * ph[a] * uq[i];
              if (Jm)
                for (int b = 0; b < nl; b++) Jm[(3 * a + i) * nd + 3 * b + i] -= th * cq * ph[a] * ph[b];
            }
        }
      }
    }
  }
  free(phiq); free(dphiq); free(hq); free(wq);
}

/* per-cell element stiffness K [nc][nl*nl], diagonal of the consistent mass Md [nc][nl] and cell measure (host side of the
 * Cahouet-Chabard preconditioner in the C driver of cfdh_oracle.c; same quantities as csrc/cfdh_gen3.hip::cfdh_build_mesh_gen3) */
void orcg3_stiff_mass(int et, int64_t nc, const int64_t *cells, const double *x, double *K, double *Md, double *meas) {
  const int nl = nloc_of(et);
  double (*phiq)[MAXL] = malloc(sizeof(double) * MAXQ * MAXL);
  double (*dphiq)[MAXL][3] = malloc(sizeof(double) * MAXQ * MAXL * 3);
  double (*hq)[MAXL][3][3] = malloc(sizeof(double) * MAXL * 9);
  double *wq = malloc(sizeof(double) * MAXQ);
  const int nq = et == 5 ? 343 : CFDH3_NQ;
  for (int q = 0; q < nq; q++) {
    double pt[3];
    if (et == 5) {
      const int i = q / 49, j = (q / 7) % 7, k = q % 7;
      pt[0] = CFDH_GL7_X[i]; pt[1] = CFDH_GL7_X[j]; pt[2] = CFDH_GL7_X[k];
      wq[q] = CFDH_GL7_W[i] * CFDH_GL7_W[j] * CFDH_GL7_W[k];
    } else {
      pt[0] = CFDH3_QL[q][1]; pt[1] = CFDH3_QL[q][2]; pt[2] = CFDH3_QL[q][3];
      wq[q] = CFDH3_QW[q] / 6.0;
    }
    tabulate(et, pt, phiq[q], dphiq[q], hq[0]);
  }
  const int c3 = et == 5 ? 4 : 3;
  for (int64_t c = 0; c < nc; c++) {
    const int64_t *cl = cells + c * nl;
    double Jm3[3][3], Ji[3][3];
    for (int i = 0; i < 3; i++) { Jm3[i][0] = x[3 * cl[1] + i] - x[3 * cl[0] + i]; Jm3[i][1] = x[3 * cl[2] + i] - x[3 * cl[0] + i]; Jm3[i][2] = x[3 * cl[c3] + i] - x[3 * cl[0] + i]; }
    const double det = Jm3[0][0] * (Jm3[1][1] * Jm3[2][2] - Jm3[1][2] * Jm3[2][1]) - Jm3[0][1] * (Jm3[1][0] * Jm3[2][2] - Jm3[1][2] * Jm3[2][0]) +
                       Jm3[0][2] * (Jm3[1][0] * Jm3[2][1] - Jm3[1][1] * Jm3[2][0]);
    const double adet = fabs(det);
    Ji[0][0] = (Jm3[1][1] * Jm3[2][2] - Jm3[1][2] * Jm3[2][1]) / det; Ji[0][1] = (Jm3[0][2] * Jm3[2][1] - Jm3[0][1] * Jm3[2][2]) / det; Ji[0][2] = (Jm3[0][1] * Jm3[1][2] - Jm3[0][2] * Jm3[1][1]) / det;
    Ji[1][0] = (Jm3[1][2] * Jm3[2][0] - Jm3[1][0] * Jm3[2][2]) / det; Ji[1][1] = (Jm3[0][0] * Jm3[2][2] - Jm3[0][2] * Jm3[2][0]) / det; Ji[1][2] = (Jm3[0][2] * Jm3[1][0] - Jm3[0][0] * Jm3[1][2]) / det;
    Ji[2][0] = (Jm3[1][0] * Jm3[2][1] - Jm3[1][1] * Jm3[2][0]) / det; Ji[2][1] = (Jm3[0][1] * Jm3[2][0] - Jm3[0][0] * Jm3[2][1]) / det; Ji[2][2] = (Jm3[0][0] * Jm3[1][1] - Jm3[0][1] * Jm3[1][0]) / det;
    double *Kc = K + c * nl * nl, *Mc = Md + c * nl;
    for (int r = 0; r < nl * nl; r++) Kc[r] = 0.0;
    for (int a = 0; a < nl; a++) Mc[a] = 0.0;
    for (int q = 0; q < nq; q++) {
      double g[MAXL][3];
      for (int a = 0; a < nl; a++)
        for (int i = 0; i < 3; i++) g[a][i] = dphiq[q][a][0] * Ji[0][i] + dphiq[q][a][1] * Ji[1][i] + dphiq[q][a][2] * Ji[2][i];
      for (int a = 0; a < nl; a++) {
        Mc[a] += adet * wq[q] * phiq[q][a] * phiq[q][a];
        for (int b = 0; b < nl; b++) Kc[a * nl + b] += adet * wq[q] * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
      }
    }
    meas[c] = adet * (et == 5 ? 1.0 : 1.0 / 6.0);
  }
  free(phiq); free(dphiq); free(hq); free(wq);
}
int orcg3_facet_nodes(int et, int f, int *out) {
  if (et == 5) { for (int k = 0; k < 4; k++) out[k] = HEX_FACETS[f][k]; return 4; }
  for (int k = 0; k < 3; k++) out[k] = TET_FACETS[f][k];
  if (et != 4) return 3;
  int n = 3;
  for (int e = 0; e < 6; e++) if (TET_EDGES[e][0] != f && TET_EDGES[e][1] != f) out[n++] = 4 + e;
  return n;
}
