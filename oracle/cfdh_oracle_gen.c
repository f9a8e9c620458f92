/* CPU oracle for nodal equal-order elements beyond P1 simplices: P2/P2 triangles and Q1/Q1 parallelograms
 *   --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.  PARITY UNPINNED (see cfdh_oracle.c).
 *
 * Plain-C restatement, with scalar loops over cells / quadrature points / local nodes, of the element residual and its
 * exact derivative as oracle/np_twin_gen.py states them (which on P1 triangles equals the closed-form twin to round-off):
 *   residual          /root/reference/src/solvers/stabilized_schur.py:67-123
 *   backflow variant  /root/reference/src/solvers/stabilized_schur_backflow.py:84-87 (p_grade), :107 (no ds pair), :158-176
 *   quadrilaterals    /root/reference/src/scenarios/unit_square_pipe.py:101-105
 * It plugs into the twin's assembly (oracle/orcg.py replaces np_twin_gen.element_tensors), as cfdh_oracle3.c does for
 * tetrahedra.  Local orders, quadrature and conventions: header of np_twin_gen.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "cfdh_quad_gl.h"
#include "cfdh_quad_tri.h"

#define MAXL 6   /* local nodes */
#define MAXQ 49  /* cell quadrature points (49 on triangles, 7 x 7 on quadrilaterals) */
#define EPS_VNORM 1e-15

typedef struct {
  double dt, rho, mu, muf, f[2], theta, a0, a1, a2, beta;
  int32_t ds_terms, pad;
} orcg_params;

static int nloc_of(int et) { return et == 0 ? 3 : (et == 1 ? 6 : 4); }

/* basis values and reference gradients at a reference point */
static void tabulate(int et, double x, double y, double *phi, double (*d)[2]) {
  if (et == 2) {
    phi[0] = (1 - x) * (1 - y); phi[1] = x * (1 - y); phi[2] = (1 - x) * y; phi[3] = x * y;
    d[0][0] = -(1 - y); d[0][1] = -(1 - x);
    d[1][0] = (1 - y);  d[1][1] = -x;
    d[2][0] = -y;       d[2][1] = (1 - x);
    d[3][0] = y;        d[3][1] = x;
    return;
  }
  const double l[3] = {1.0 - x - y, x, y};
  static const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  if (et == 0) {
    for (int a = 0; a < 3; a++) { phi[a] = l[a]; d[a][0] = dl[a][0]; d[a][1] = dl[a][1]; }
    return;
  }
  static const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
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

/* constant reference Hessians */
static void ref_hessians(int et, double (*H)[2][2]) {
  memset(H, 0, sizeof(double) * MAXL * 4);
  static const double dl[3][2] = {{-1.0, -1.0}, {1.0, 0.0}, {0.0, 1.0}};
  static const int ed[3][2] = {{1, 2}, {0, 2}, {0, 1}};
  if (et == 1) {
    for (int a = 0; a < 3; a++)
      for (int k = 0; k < 2; k++)
        for (int l = 0; l < 2; l++) H[a][k][l] = 4.0 * dl[a][k] * dl[a][l];
    for (int e = 0; e < 3; e++)
      for (int k = 0; k < 2; k++)
        for (int l = 0; l < 2; l++) H[3 + e][k][l] = 4.0 * (dl[ed[e][0]][k] * dl[ed[e][1]][l] + dl[ed[e][1]][k] * dl[ed[e][0]][l]);
  } else if (et == 2) {
    static const double s[4] = {1.0, -1.0, -1.0, 1.0};
    for (int a = 0; a < 4; a++) H[a][0][1] = H[a][1][0] = s[a];
  }
}

static int cell_rule(int et, double (*xi)[2], double *w) {
  if (et == 2) {
    for (int i = 0; i < 7; i++)
      for (int j = 0; j < 7; j++) { xi[7 * i + j][0] = CFDH_GL7_X[i]; xi[7 * i + j][1] = CFDH_GL7_X[j]; w[7 * i + j] = CFDH_GL7_W[i] * CFDH_GL7_W[j]; }
    return 49;
  }
  for (int q = 0; q < CFDH_NQ; q++) { xi[q][0] = CFDH_QL[q][1]; xi[q][1] = CFDH_QL[q][2]; w[q] = 0.5 * CFDH_QW[q]; }
  return CFDH_NQ;
}

static const int TRI_FACETS[3][2] = {{1, 2}, {0, 2}, {0, 1}};
static const int QUAD_FACETS[4][2] = {{0, 1}, {0, 2}, {1, 3}, {2, 3}};
static const double TRI_REF[3][2] = {{0, 0}, {1, 0}, {0, 1}};
static const double QUAD_REF[4][2] = {{0, 0}, {1, 0}, {0, 1}, {1, 1}};

static void tau_pair(double s, double h, const orcg_params *P, double *tau, double *tauL) {
  const double nu = P->mu / P->rho;
  double t1 = 4.0 * s;
  if (t1 < EPS_VNORM * EPS_VNORM) t1 = EPS_VNORM * EPS_VNORM;
  t1 /= h * h;
  *tau = 1.0 / sqrt(t1 + 4.0 / (P->dt * P->dt) + 16.0 * nu * nu / (h * h * h * h));
  const double vn = sqrt(s), Re = vn * h / (2.0 * nu), z = Re <= 3.0 ? Re / 3.0 : 1.0;
  *tauL = vn * h * z / 2.0;
}

/* Fe [nc][3 nloc], Je [nc][3 nloc][3 nloc] (row-major); flags: bit f exterior facet f, bit 8+f backflow facet f.
 * un2 may be NULL when a2 == 0; Je may be NULL when want_jac == 0. */
void orcg_element_tensors(int et, int64_t nc, const int64_t *cells, const double *x, const double *u, const double *un, const double *un2,
                          const double *p, const orcg_params *P, const uint16_t *flags, int want_jac, double *Fe, double *Je) {
  const int nl = nloc_of(et), nd = 3 * nl, nvert = et == 2 ? 4 : 3, nfac = et == 2 ? 4 : 3;
  double xi[MAXQ][2], wq[MAXQ];
  const int nq = cell_rule(et, xi, wq);
  double phiq[MAXQ][MAXL], dphiq[MAXQ][MAXL][2];
  double Href[MAXL][2][2];
  for (int q = 0; q < nq; q++) tabulate(et, xi[q][0], xi[q][1], phiq[q], dphiq[q]);
  ref_hessians(et, Href);
  const int nqf = et == 1 ? 4 : 2;
  const double *ft = et == 1 ? CFDH_GL4_X : CFDH_GL2_X, *fw = et == 1 ? CFDH_GL4_W : CFDH_GL2_W;
  const double rho = P->rho, mu = P->mu, th = P->theta, a0dt = P->a0 / P->dt;
#pragma omp parallel for schedule(static)
  for (int64_t c = 0; c < nc; c++) {
    const int64_t *cl = cells + c * nl;
    double X[MAXL][2], ub[MAXL][2], wn[MAXL][2], une[MAXL][2], pe[MAXL];
    for (int a = 0; a < nl; a++) {
      const int64_t v = cl[a];
      for (int i = 0; i < 2; i++) {
        X[a][i] = x[2 * v + i];
        une[a][i] = un[2 * v + i];
        ub[a][i] = th * u[2 * v + i] + (1.0 - th) * une[a][i];
        wn[a][i] = (P->a0 * u[2 * v + i] + P->a1 * une[a][i] + (P->a2 != 0.0 ? P->a2 * un2[2 * v + i] : 0.0)) / P->dt;
      }
      pe[a] = p[v];
    }
    /* affine map from the first three vertices */
    const double J00 = X[1][0] - X[0][0], J01 = X[2][0] - X[0][0], J10 = X[1][1] - X[0][1], J11 = X[2][1] - X[0][1];
    const double det = J00 * J11 - J01 * J10, adet = fabs(det);
    const double Ji[2][2] = {{J11 / det, -J01 / det}, {-J10 / det, J00 / det}};  /* Ji[k][i] = d xi_k / d x_i */
    double h = 0.0;
    for (int a = 0; a < nvert; a++)
      for (int b = a + 1; b < nvert; b++) { const double d = hypot(X[a][0] - X[b][0], X[a][1] - X[b][1]); if (d > h) h = d; }
    /* physical Hessians (cell constants), Laplacians, viscous part of the strong residual */
    double Hs[MAXL][2][2], lap[MAXL], visc[2] = {0.0, 0.0};
    for (int a = 0; a < nl; a++) {
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) {
          double s = 0.0;
          for (int k = 0; k < 2; k++)
            for (int l = 0; l < 2; l++) s += Href[a][k][l] * Ji[k][i] * Ji[l][j];
          Hs[a][i][j] = s;
        }
      lap[a] = Hs[a][0][0] + Hs[a][1][1];
      for (int i = 0; i < 2; i++) visc[i] += mu * (lap[a] * ub[a][i] + Hs[a][i][0] * ub[a][0] + Hs[a][i][1] * ub[a][1]);
    }
    double *F = Fe + c * nd, *Jm = want_jac ? Je + c * nd * nd : NULL;
    for (int r = 0; r < nd; r++) F[r] = 0.0;
    if (Jm) for (int r = 0; r < nd * nd; r++) Jm[r] = 0.0;
    for (int q = 0; q < nq; q++) {
      const double dv = adet * wq[q];
      double g[MAXL][2];
      for (int a = 0; a < nl; a++)
        for (int i = 0; i < 2; i++) g[a][i] = dphiq[q][a][0] * Ji[0][i] + dphiq[q][a][1] * Ji[1][i];
      const double *ph = phiq[q];
      double uq[2] = {0, 0}, wv[2] = {0, 0}, unq[2] = {0, 0}, G[2][2] = {{0, 0}, {0, 0}}, gp[2] = {0, 0}, pq = 0.0;
      for (int a = 0; a < nl; a++) {
        for (int i = 0; i < 2; i++) {
          uq[i] += ph[a] * ub[a][i]; wv[i] += ph[a] * wn[a][i]; unq[i] += ph[a] * une[a][i];
          gp[i] += g[a][i] * pe[a];
          for (int j = 0; j < 2; j++) G[i][j] += g[a][i] * ub[a][j];
        }
        pq += ph[a] * pe[a];
      }
      const double divu = G[0][0] + G[1][1];
      double C[2], R[2], bgr[MAXL], tau, tauL;
      for (int j = 0; j < 2; j++) C[j] = uq[0] * G[0][j] + uq[1] * G[1][j];
      for (int i = 0; i < 2; i++) R[i] = rho * (wv[i] + C[i]) - visc[i] + gp[i] - rho * P->f[i];
      tau_pair(unq[0] * unq[0] + unq[1] * unq[1], h, P, &tau, &tauL);
      for (int a = 0; a < nl; a++) bgr[a] = uq[0] * g[a][0] + uq[1] * g[a][1];
      for (int a = 0; a < nl; a++) {
        for (int i = 0; i < 2; i++) {
          double v = rho * ph[a] * (wv[i] + C[i] - P->f[i]);
          for (int j = 0; j < 2; j++) v += g[a][j] * mu * (G[i][j] + G[j][i]);
          v += -pq * g[a][i] + tau * R[i] * bgr[a] + tauL * rho * divu * g[a][i];
          F[2 * a + i] += dv * v;
        }
        F[2 * nl + a] += dv * (ph[a] * divu + tau / rho * (R[0] * g[a][0] + R[1] * g[a][1]));
      }
      if (!Jm) continue;
      for (int b = 0; b < nl; b++)
        for (int j = 0; j < 2; j++) {
          /* derivative of the strong residual and of rho (w + C) with respect to u_(b,j) */
          double dR[2], dWC[2];
          for (int i = 0; i < 2; i++) {
            const double dij = i == j ? 1.0 : 0.0;
            dWC[i] = rho * (a0dt * ph[b] * dij + th * (ph[b] * G[j][i] + dij * bgr[b]));
            dR[i] = dWC[i] - mu * th * (lap[b] * dij + Hs[b][i][j]);
          }
          for (int a = 0; a < nl; a++) {
            const double gg = g[a][0] * g[b][0] + g[a][1] * g[b][1];
            for (int i = 0; i < 2; i++) {
              const double dij = i == j ? 1.0 : 0.0;
              double v = ph[a] * dWC[i] + mu * th * (g[a][j] * g[b][i] + dij * gg) + tau * dR[i] * bgr[a] + th * tau * R[i] * ph[b] * g[a][j] +
                         rho * th * tauL * g[b][j] * g[a][i];
              Jm[(2 * a + i) * nd + 2 * b + j] += dv * v;
            }
            Jm[(2 * nl + a) * nd + 2 * b + j] += dv * (th * ph[a] * g[b][j] + tau / rho * (dR[0] * g[a][0] + dR[1] * g[a][1]));
          }
        }
      for (int b = 0; b < nl; b++)
        for (int a = 0; a < nl; a++) {
          for (int i = 0; i < 2; i++) Jm[(2 * a + i) * nd + 2 * nl + b] += dv * (-ph[b] * g[a][i] + tau * g[b][i] * bgr[a]);
          Jm[(2 * nl + a) * nd + 2 * nl + b] += dv * tau / rho * (g[b][0] * g[a][0] + g[b][1] * g[a][1]);
        }
    }
    /* exterior-facet terms */
    const unsigned fl = flags ? flags[c] : 0u;
    if (!fl) continue;
    double cen[2] = {0, 0};
    for (int a = 0; a < nvert; a++) { cen[0] += X[a][0] / nvert; cen[1] += X[a][1] / nvert; }
    for (int f = 0; f < nfac; f++) {
      const int ext = P->ds_terms && ((fl >> f) & 1u), bfl = P->beta != 0.0 && ((fl >> (8 + f)) & 1u);
      if (!ext && !bfl) continue;
      const int va = et == 2 ? QUAD_FACETS[f][0] : TRI_FACETS[f][0], vb = et == 2 ? QUAD_FACETS[f][1] : TRI_FACETS[f][1];
      const double tx = X[vb][0] - X[va][0], ty = X[vb][1] - X[va][1], elen = hypot(tx, ty);
      double n[2] = {ty / elen, -tx / elen};
      if ((0.5 * (X[va][0] + X[vb][0]) - cen[0]) * n[0] + (0.5 * (X[va][1] + X[vb][1]) - cen[1]) * n[1] < 0) { n[0] = -n[0]; n[1] = -n[1]; }
      const double(*ref)[2] = et == 2 ? QUAD_REF : TRI_REF;
      for (int q = 0; q < nqf; q++) {
        const double t = ft[q], m = elen * fw[q];
        double ph[MAXL], dr[MAXL][2], g[MAXL][2];
        tabulate(et, (1 - t) * ref[va][0] + t * ref[vb][0], (1 - t) * ref[va][1] + t * ref[vb][1], ph, dr);
        for (int a = 0; a < nl; a++)
          for (int i = 0; i < 2; i++) g[a][i] = dr[a][0] * Ji[0][i] + dr[a][1] * Ji[1][i];
        double uq[2] = {0, 0}, G[2][2] = {{0, 0}, {0, 0}}, pq = 0.0, sn = 0.0;
        for (int a = 0; a < nl; a++) {
          for (int i = 0; i < 2; i++) {
            uq[i] += ph[a] * ub[a][i];
            sn += ph[a] * une[a][i] * n[i];
            for (int j = 0; j < 2; j++) G[i][j] += g[a][i] * ub[a][j];
          }
          pq += ph[a] * pe[a];
        }
        if (ext) {
          for (int a = 0; a < nl; a++)
            for (int i = 0; i < 2; i++) {
              F[2 * a + i] += m * ph[a] * (pq * n[i] - P->muf * (G[i][0] * n[0] + G[i][1] * n[1]));
              if (Jm)
                for (int b = 0; b < nl; b++) {
                  Jm[(2 * a + i) * nd + 2 * nl + b] += m * ph[a] * ph[b] * n[i];
                  for (int j = 0; j < 2; j++) Jm[(2 * a + i) * nd + 2 * b + j] -= P->muf * th * m * ph[a] * g[b][i] * n[j];
                }
            }
        }
        if (bfl) {
          const double cq = P->beta * rho * 0.5 * (sn - fabs(sn)) * m;
          for (int a = 0; a < nl; a++)
            for (int i = 0; i < 2; i++) {
              F[2 * a + i] -= cq * ph[a] * uq[i];
              if (Jm)
                for (int b = 0; b < nl; b++) Jm[(2 * a + i) * nd + 2 * b + i] -= th * cq * ph[a] * ph[b];
            }
        }
      }
    }
  }
}

/* per-cell element stiffness K [nc][nl*nl], diagonal of the consistent mass Md [nc][nl] and cell measure (host side of the
 * Cahouet-Chabard preconditioner in the C driver of cfdh_oracle.c; same quantities as csrc/cfdh_gen.hip::cfdh_build_mesh_gen) */
void orcg_stiff_mass(int et, int64_t nc, const int64_t *cells, const double *x, double *K, double *Md, double *meas) {
  const int nl = nloc_of(et);
  double xi[MAXQ][2], wq[MAXQ];
  const int nq = cell_rule(et, xi, wq);
  double phiq[MAXQ][MAXL], dphiq[MAXQ][MAXL][2];
  for (int q = 0; q < nq; q++) tabulate(et, xi[q][0], xi[q][1], phiq[q], dphiq[q]);
  for (int64_t c = 0; c < nc; c++) {
    const int64_t *cl = cells + c * nl;
    const double J00 = x[2 * cl[1]] - x[2 * cl[0]], J01 = x[2 * cl[2]] - x[2 * cl[0]], J10 = x[2 * cl[1] + 1] - x[2 * cl[0] + 1], J11 = x[2 * cl[2] + 1] - x[2 * cl[0] + 1];
    const double det = J00 * J11 - J01 * J10, adet = fabs(det);
    const double Ji[2][2] = {{J11 / det, -J01 / det}, {-J10 / det, J00 / det}};
    double *Kc = K + c * nl * nl, *Mc = Md + c * nl;
    for (int r = 0; r < nl * nl; r++) Kc[r] = 0.0;
    for (int a = 0; a < nl; a++) Mc[a] = 0.0;
    for (int q = 0; q < nq; q++) {
      double g[MAXL][2];
      for (int a = 0; a < nl; a++)
        for (int i = 0; i < 2; i++) g[a][i] = dphiq[q][a][0] * Ji[0][i] + dphiq[q][a][1] * Ji[1][i];
      for (int a = 0; a < nl; a++) {
        Mc[a] += adet * wq[q] * phiq[q][a] * phiq[q][a];
        for (int b = 0; b < nl; b++) Kc[a * nl + b] += adet * wq[q] * (g[a][0] * g[b][0] + g[a][1] * g[b][1]);
      }
    }
    meas[c] = adet * (et == 2 ? 1.0 : 0.5);
  }
}
int orcg_facet_nodes(int et, int f, int *out) {
  if (et == 2) { out[0] = QUAD_FACETS[f][0]; out[1] = QUAD_FACETS[f][1]; return 2; }
  out[0] = TRI_FACETS[f][0]; out[1] = TRI_FACETS[f][1];
  if (et == 1) { out[2] = 3 + f; return 3; }
  return 2;
}
