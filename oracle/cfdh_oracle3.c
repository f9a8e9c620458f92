/*
 * cfdh_oracle3.c -- CPU restatement (plain C) of the element residual and Jacobian of the reference's
 * `stabilized_schur` weak form on affine P1/P1 TETRAHEDRA (gdim = 3): the 3-D companion of cfdh_oracle.c.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product imports, links or executes this file; it is the independent
 * checker of the tetrahedral HIP kernels (tests/) next to the NumPy twin oracle/np_twin_nd.py, written from the
 * same algebra (SURVEY.md Appendix A with d = 3) but with scalar loops instead of einsum expressions.
 * Parity is unpinned as for the 2-D oracle: the reference holds no fixture for this path.
 *
 * What it restates, per cell (/root/reference/src/solvers/stabilized_schur.py):
 *   :71        u_mid = theta u + (1 - theta) u_prev (theta = 1/2; stabilized_schur_bdf2.py:79-110: theta = 1 and
 *              the (a0 u + a1 u_prev + a2 u_prev2) / dt time term)
 *   :72-80     Galerkin terms, the ds pair on exterior facets (:79) with the raw mu
 *   :82-88     h = greatest vertex distance
 *   :100-108   tau_SUPG = tau_PSPG = ((2/dt)^2 + (2 |u_prev| / h)^2 + (4 nu / h^2)^2)^(-1/2), max(2|u_prev|, eps)
 *   :116-118   tau_LSIC = |u_prev| h z / 2, z = Re_h / 3 for Re_h <= 3 else 1
 *   :119-123   SUPG / PSPG / LSIC terms;  :185-189 exact Jacobian (tau depends on u_prev only)
 *   stabilized_schur_backflow.py:165-176  backflow stabilisation on flagged facets
 * Local dof order: velocity (vertex a, component i) -> 3 a + i, pressure of vertex a -> 12 + a.
 * facet_flags[cell]: bit f (0..3) = facet opposite local vertex f is exterior, bit 4 + f = ... is a backflow facet.
 */
#include <math.h>
#include <stddef.h>
#include <string.h>

#include "cfdh_quad_tet.h"

#define EPS_VNORM2 1e-30 /* (1e-15)^2: max(2|u|, eps)^2 with eps = 1e-15 as 4 s vs eps^2 */

static const double FR_A = 0.659027622374092, FR_B = 0.231933368553031, FR_C = 0.109039009072877;

/* Fe [nc][16], Je [nc][16][16] (row-major, may be NULL).  Returns 0. */
int orc3_element_tensors(int nc, const double *x, const int *cells, const double *u, const double *un, const double *un2,
                         const double *p, const unsigned char *facet_flags, double dt, double rho, double mu, double muf,
                         const double *f, double theta, double a0, double a1, double a2, int ds_terms, double beta_bf,
                         double *Fe, double *Je) {
  const double FRP[6][3] = {{FR_A, FR_B, FR_C}, {FR_A, FR_C, FR_B}, {FR_B, FR_A, FR_C},
                            {FR_B, FR_C, FR_A}, {FR_C, FR_A, FR_B}, {FR_C, FR_B, FR_A}};
#pragma omp parallel for schedule(static)
  for (int c = 0; c < nc; c++) {
    int vs[4];
    double X[4][3], ue[4][3], une[4][3], u2e[4][3], pe[4];
    for (int a = 0; a < 4; a++) {
      vs[a] = cells[4 * (size_t)c + a];
      for (int i = 0; i < 3; i++) {
        X[a][i] = x[3 * (size_t)vs[a] + i];
        ue[a][i] = u[3 * (size_t)vs[a] + i];
        une[a][i] = un[3 * (size_t)vs[a] + i];
        u2e[a][i] = un2 ? un2[3 * (size_t)vs[a] + i] : 0.0;
      }
      pe[a] = p[vs[a]];
    }
    /* geometry: J = [x1-x0 | x2-x0 | x3-x0] (columns), rows of J^-1 are grad lambda_1..3 */
    double Jm[3][3];
    for (int i = 0; i < 3; i++)
      for (int a = 0; a < 3; a++) Jm[i][a] = X[a + 1][i] - X[0][i];
    const double det = Jm[0][0] * (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) - Jm[0][1] * (Jm[1][0] * Jm[2][2] - Jm[1][2] * Jm[2][0]) +
                       Jm[0][2] * (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]);
    double g[4][3];
    {
      double inv[3][3];
      inv[0][0] = (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) / det;
      inv[0][1] = (Jm[0][2] * Jm[2][1] - Jm[0][1] * Jm[2][2]) / det;
      inv[0][2] = (Jm[0][1] * Jm[1][2] - Jm[0][2] * Jm[1][1]) / det;
      inv[1][0] = (Jm[1][2] * Jm[2][0] - Jm[1][0] * Jm[2][2]) / det;
      inv[1][1] = (Jm[0][0] * Jm[2][2] - Jm[0][2] * Jm[2][0]) / det;
      inv[1][2] = (Jm[0][2] * Jm[1][0] - Jm[0][0] * Jm[1][2]) / det;
      inv[2][0] = (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]) / det;
      inv[2][1] = (Jm[0][1] * Jm[2][0] - Jm[0][0] * Jm[2][1]) / det;
      inv[2][2] = (Jm[0][0] * Jm[1][1] - Jm[0][1] * Jm[1][0]) / det;
      for (int a = 0; a < 3; a++)
        for (int i = 0; i < 3; i++) g[a + 1][i] = inv[a][i];
      for (int i = 0; i < 3; i++) g[0][i] = -(inv[0][i] + inv[1][i] + inv[2][i]);
    }
    const double vol = fabs(det) / 6.0;
    double h = 0.0;
    for (int a = 0; a < 4; a++)
      for (int b = a + 1; b < 4; b++) {
        double s2 = 0.0;
        for (int i = 0; i < 3; i++) s2 += (X[a][i] - X[b][i]) * (X[a][i] - X[b][i]);
        if (sqrt(s2) > h) h = sqrt(s2);
      }
    /* fields */
    double ub[4][3], w[4][3];
    for (int a = 0; a < 4; a++)
      for (int i = 0; i < 3; i++) {
        ub[a][i] = theta * ue[a][i] + (1.0 - theta) * une[a][i];
        w[a][i] = (a0 * ue[a][i] + a1 * une[a][i]) / dt;
        if (a2 != 0.0) w[a][i] += a2 * u2e[a][i] / dt;
      }
    double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, gp[3] = {0, 0, 0};
    for (int a = 0; a < 4; a++)
      for (int i = 0; i < 3; i++) {
        gp[i] += pe[a] * g[a][i];
        for (int j = 0; j < 3; j++) G[i][j] += g[a][i] * ub[a][j];
      }
    const double divu = G[0][0] + G[1][1] + G[2][2];
    double wc[4][3], R[4][3], beta[4][4];
    for (int a = 0; a < 4; a++)
      for (int j = 0; j < 3; j++) {
        double cn = 0.0;
        for (int i = 0; i < 3; i++) cn += ub[a][i] * G[i][j];
        wc[a][j] = w[a][j] + cn;
        R[a][j] = rho * wc[a][j] + gp[j] - rho * f[j];
      }
    for (int b = 0; b < 4; b++)
      for (int a = 0; a < 4; a++) beta[b][a] = ub[b][0] * g[a][0] + ub[b][1] * g[a][1] + ub[b][2] * g[a][2];
    double mab[4][4];
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) mab[a][b] = vol * (a == b ? 2.0 : 1.0) / 20.0;
    /* tau moments on the degree-13 rule of cfdh_quad_tet.h (171 points) */
    double M[4][4], Lm = 0.0;
    memset(M, 0, sizeof M);
    {
      const double nu = mu / rho, t2 = 4.0 / (dt * dt), t3 = 16.0 * nu * nu / (h * h * h * h);
      for (int q = 0; q < CFDH3_NQ; q++) {
        double uq[3] = {0, 0, 0};
        for (int a = 0; a < 4; a++)
          for (int i = 0; i < 3; i++) uq[i] += CFDH3_QL[q][a] * une[a][i];
        const double s = uq[0] * uq[0] + uq[1] * uq[1] + uq[2] * uq[2];
        const double t1 = (4.0 * s > EPS_VNORM2 ? 4.0 * s : EPS_VNORM2) / (h * h);
        const double tau = 1.0 / sqrt(t1 + t2 + t3);
        const double vn = sqrt(s), Re = vn * h / (2.0 * nu);
        const double z = Re <= 3.0 ? Re / 3.0 : 1.0;
        Lm += CFDH3_QW[q] * vn * h * z / 2.0;
        for (int a = 0; a < 4; a++)
          for (int b = 0; b < 4; b++) M[a][b] += CFDH3_QW[q] * tau * CFDH3_QL[q][a] * CFDH3_QL[q][b];
      }
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) M[a][b] *= vol;
      Lm *= vol;
    }
    double mt[4], T = 0.0, Q[4][3];
    for (int b = 0; b < 4; b++) { mt[b] = M[b][0] + M[b][1] + M[b][2] + M[b][3]; T += mt[b]; }
    for (int d = 0; d < 4; d++)
      for (int i = 0; i < 3; i++) Q[d][i] = M[0][d] * R[0][i] + M[1][d] * R[1][i] + M[2][d] * R[2][i] + M[3][d] * R[3][i];
    const double pbar = 0.25 * (pe[0] + pe[1] + pe[2] + pe[3]);
    double Fu[4][3], Fp[4];
    for (int a = 0; a < 4; a++) {
      for (int i = 0; i < 3; i++) {
        double v = 0.0;
        for (int b = 0; b < 4; b++) v += rho * mab[a][b] * wc[b][i];
        v -= rho * f[i] * vol / 4.0;
        double Eg = 0.0;
        for (int k = 0; k < 3; k++) Eg += 0.5 * (G[i][k] + G[k][i]) * g[a][k];
        v += vol * (2.0 * mu * Eg - pbar * g[a][i]);
        for (int d = 0; d < 4; d++) v += beta[d][a] * Q[d][i];
        v += rho * Lm * divu * g[a][i];
        Fu[a][i] = v;
      }
      double v = vol / 4.0 * divu;
      for (int b = 0; b < 4; b++) v += mt[b] * (R[b][0] * g[a][0] + R[b][1] * g[a][1] + R[b][2] * g[a][2]) / rho;
      Fp[a] = v;
    }
    double *J = Je ? Je + 256 * (size_t)c : NULL;
    if (J) {
      memset(J, 0, 256 * sizeof(double));
      double MB[4][4], BMB[4][4], mB[4][4], mtB[4], gg[4][4];
      for (int b = 0; b < 4; b++)
        for (int a = 0; a < 4; a++) {
          double s1 = 0.0;
          for (int d = 0; d < 4; d++) s1 += M[b][d] * beta[d][a];
          MB[b][a] = s1;
        }
      for (int b = 0; b < 4; b++)
        for (int a = 0; a < 4; a++) {
          double s1 = 0.0, s2 = 0.0;
          for (int d = 0; d < 4; d++) { s1 += beta[d][b] * MB[d][a]; s2 += mab[a][d] * beta[d][b]; }
          BMB[b][a] = s1;
          mB[a][b] = s2;
          gg[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2];
        }
      for (int a = 0; a < 4; a++) {
        double s1 = 0.0;
        for (int d = 0; d < 4; d++) s1 += mt[d] * beta[d][a];
        mtB[a] = s1;
      }
      for (int a = 0; a < 4; a++)
        for (int b = 0; b < 4; b++) {
          for (int i = 0; i < 3; i++) {
            for (int j = 0; j < 3; j++) {
              const double dij = i == j ? 1.0 : 0.0;
              double v = rho * mab[a][b] * dij * a0 / dt;
              v += rho * theta * (mab[a][b] * G[j][i] + dij * mB[a][b]);
              v += vol * mu * theta * (g[b][i] * g[a][j] + gg[a][b] * dij);
              v += rho * ((dij * a0 / dt + theta * G[j][i]) * MB[b][a] + theta * dij * BMB[b][a]);
              v += theta * g[a][j] * Q[b][i];
              v += rho * Lm * theta * g[b][j] * g[a][i];
              J[16 * (3 * a + i) + 3 * b + j] = v;
            }
            J[16 * (3 * a + i) + 12 + b] = -vol / 4.0 * g[a][i] + g[b][i] * mtB[a];
          }
          for (int j = 0; j < 3; j++) {
            const double Gg = G[j][0] * g[a][0] + G[j][1] * g[a][1] + G[j][2] * g[a][2];
            J[16 * (12 + a) + 3 * b + j] = vol / 4.0 * theta * g[b][j] + mt[b] * (g[a][j] * a0 / dt + theta * Gg) + theta * g[a][j] * mtB[b];
          }
          J[16 * (12 + a) + 12 + b] = T * gg[a][b] / rho;
        }
    }
    const unsigned flags = facet_flags ? facet_flags[c] : 0u;
    /* backflow stabilisation: F -= beta rho int_f (u_prev . n)_- (ubar . v) ds, 6-point degree-3 rule */
    if (beta_bf != 0.0 && (flags >> 4))
      for (int fc = 0; fc < 4; fc++) {
        if (!((flags >> (4 + fc)) & 1u)) continue;
        const double gl = sqrt(g[fc][0] * g[fc][0] + g[fc][1] * g[fc][1] + g[fc][2] * g[fc][2]);
        const double n[3] = {-g[fc][0] / gl, -g[fc][1] / gl, -g[fc][2] / gl};
        const double fm = 3.0 * vol * gl;
        int ev[3], ne = 0;
        for (int a = 0; a < 4; a++) if (a != fc) ev[ne++] = a;
        double sv[3];
        for (int k = 0; k < 3; k++) sv[k] = une[ev[k]][0] * n[0] + une[ev[k]][1] * n[1] + une[ev[k]][2] * n[2];
        for (int q = 0; q < 6; q++) {
          const double sq = FRP[q][0] * sv[0] + FRP[q][1] * sv[1] + FRP[q][2] * sv[2];
          const double cq = beta_bf * rho * 0.5 * (sq - fabs(sq)) * (1.0 / 6.0) * fm;
          double uq[3] = {0, 0, 0};
          for (int k = 0; k < 3; k++)
            for (int i = 0; i < 3; i++) uq[i] += FRP[q][k] * ub[ev[k]][i];
          for (int ka = 0; ka < 3; ka++) {
            for (int i = 0; i < 3; i++) Fu[ev[ka]][i] -= cq * FRP[q][ka] * uq[i];
            if (J)
              for (int kb = 0; kb < 3; kb++)
                for (int i = 0; i < 3; i++) J[16 * (3 * ev[ka] + i) + 3 * ev[kb] + i] -= theta * cq * FRP[q][ka] * FRP[q][kb];
          }
        }
      }
    /* the ds pair of :79 on exterior facets */
    if (ds_terms && (flags & 15u))
      for (int fc = 0; fc < 4; fc++) {
        if (!((flags >> fc) & 1u)) continue;
        const double gl = sqrt(g[fc][0] * g[fc][0] + g[fc][1] * g[fc][1] + g[fc][2] * g[fc][2]);
        const double n[3] = {-g[fc][0] / gl, -g[fc][1] / gl, -g[fc][2] / gl};
        const double fm = 3.0 * vol * gl;
        double Gn[3];
        for (int i = 0; i < 3; i++) Gn[i] = G[i][0] * n[0] + G[i][1] * n[1] + G[i][2] * n[2];
        for (int a = 0; a < 4; a++) {
          if (a == fc) continue;
          double pint = 0.0;
          for (int b = 0; b < 4; b++) if (b != fc) pint += pe[b] * (a == b ? 2.0 : 1.0);
          pint /= 12.0;
          for (int i = 0; i < 3; i++) {
            Fu[a][i] += n[i] * fm * pint - muf * Gn[i] * fm / 3.0;
            if (J) {
              for (int b = 0; b < 4; b++) if (b != fc) J[16 * (3 * a + i) + 12 + b] += n[i] * fm * (a == b ? 2.0 : 1.0) / 12.0;
              for (int b = 0; b < 4; b++)
                for (int j = 0; j < 3; j++) J[16 * (3 * a + i) + 3 * b + j] -= muf * theta * g[b][i] * n[j] * fm / 3.0;
            }
          }
        }
      }
    double *F = Fe + 16 * (size_t)c;
    for (int a = 0; a < 4; a++) {
      for (int i = 0; i < 3; i++) F[3 * a + i] = Fu[a][i];
      F[12 + a] = Fp[a];
    }
  }
  return 0;
}
