/*
 * cfdh_oracle.c -- CPU restatement of the reference's per-time-step path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may build, load or call this file.  The
 * product (libcfdh.so, HIP) never links or calls it.
 *
 * PARITY UNPINNED.  The reference's arithmetic lives in un-vendored third-party
 * code (fenics-dolfinx 0.9.0 / UFL 2024.2 / Basix 0.9 / FFCx 0.9 and the
 * PETSc+MPICH of the image dolfinx/dolfinx:v0.9.0, /root/reference/singularity.def:2)
 * which can neither be built nor imported in this container, and the reference
 * holds no test, fixture or recorded output for this path.
 * This file restates the published algorithm of those call sites and is
 * pinned only by the substitutes listed in DESIGN.md ("Oracle"): the NumPy
 * twin with a direct solver (oracle/np_twin.py), Jacobian = d(residual) finite
 * differences, a brute-force quadrature of the weak form, patch tests,
 * Poiseuille flow and the Ghia / DFG literature values.
 *
 * What is restated (2-D affine P1/P1 triangles, IEEE double, int32 indices):
 *   element residual   /root/reference/src/solvers/stabilized_schur.py:67-123
 *   element Jacobian   stabilized_schur.py:185-189 (exact Gateaux derivative)
 *   block assembly, Dirichlet rows/cols, lifting (x0=x, alpha=-1)
 *                      stabilized_schur.py:144-175
 *   Newton (SNES newtonls, bt line search) + FGMRES(200) + PCFIELDSPLIT Schur
 *   FULL / SELFP with GMRES(30)+ILU(0) on A00 and preonly+ILU(0) on
 *   Sp = A11 - A10 diag(A00)^-1 A01       stabilized_schur.py:201-275
 *   (single rank: PCASM with one subdomain per rank == ILU(0) of the block)
 *   constant-pressure null space           stabilized_schur.py:282-293,314-319
 *   functionals                            /root/reference/src/scenarios/dfg_1.py:183-202,
 *                                          /root/reference/src/scenario.py:315-324
 * Algebra of the closed-form element integrals: SURVEY.md Appendix A.
 *
 * Monolithic ordering (stabilized_schur.py:194-196): all velocity dofs
 * (vertex-major, component-minor), then all pressure dofs.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "cfdh_quad_tri.h"

#define EPS_VNORM 1e-15 /* np.finfo(float64).resolution, stabilized_schur.py:100 */

typedef struct {
  int n;
  int *rowptr, *col;
  double *val;
  int *diag;
} csr_t;

typedef struct orc_ctx {
  int nv, nc, nf, ndof;
  /* D = geometric dimension: 2 (triangles, this file's element routine) or 3 (tetrahedra: cfdh_oracle3.c supplies the element
   * tensors, everything from the assembly on is written for D + 1 unknowns per vertex).  NL = D + 1 vertices per cell. */
  int D, NL;
  /* NC = nodes per cell: D + 1 for the P1 simplices, 6 / 4 / 10 / 8 for the generic elements (etg != 0: P2 triangles 1, Q1
   * parallelograms 2, P2 tetrahedra 4, Q1 hexahedra 5 -- the element codes of np_twin_gen.py / np_twin_gen3.py), whose element
   * tensors come from cfdh_oracle_gen.c / cfdh_oracle_gen3.c; NL = D + 1 stays the number of unknowns per node */
  int NC, etg;
  int64_t *cells64;   /* [nc][NC] for the generic element routines */
  uint16_t *gflag;    /* [nc] generic elements: bit f exterior facet f, bit 8 + f backflow facet f */
  int *cells;   /* [nc][NC] */
  double *x;    /* [nv][D] */
  uint8_t *fflag; /* [nc] bit f: facet opposite local vertex f is exterior; bit 3+f: it is a backflow (outlet) facet */
  int *fcell, *flocal;
  double dt, rho, mu, muf, f[3];
  /* time scheme: spatial terms at theta*u + (1-theta)*u_n, time term (a0 u + a1 u_n + a2 u_nm1)/dt.
   * stabilized_schur.py:72-80 -> (1/2; 1,-1,0); stabilized_schur_bdf2.py:79-110 -> (1; 1,-1,0) then (1; 1.5,-2,.5) */
  double theta, a0, a1, a2;
  /* boundary terms: ds_terms = the ds pair of stabilized_schur.py:79 on all exterior facets (dropped by
   * stabilized_schur_backflow.py:107); beta_bf = backflow coefficient on the facets flagged in bits 3..5
   * of fflag (stabilized_schur_backflow.py:158-176) */
  int ds_terms;
  double beta_bf;
  uint8_t *isbc;
  double *bcval, *bcmult;
  int any_pbc;
  /* vertex graph */
  int *vptr, *vadj;       /* neighbours incl. self, sorted */
  int *vcptr, *vcell;     /* vertex -> incident cells (cell*3+local) */
  /* monolithic CSR */
  int *rowptr, *col;
  double *val;
  int nnz;
  int *cellpos;           /* [nc][9]: index of neighbour b in row of vertex a */
  /* per-step data */
  double *un;             /* [nv][2] */
  double *un2;            /* [nv][2] u_prev2 of stabilized_schur_bdf2.py:72 */
  double *Mom;            /* [nc][7]: M00 M01 M02 M11 M12 M22 L */
  double *Fe, *Je;        /* [nc][9], [nc][81] scratch */
  /* solver workspace (lazily allocated) */
  csr_t A00f, Sp;         /* ILU factors */
  int *sp_rowptr, *sp_col;
  /* pc_kind 1 / 2 */
  struct amg_hier *hS, *hL, *hA;
  struct amg_level *Hlev;
  int amg_valid, amg_its_ref, amg_force;
  double *dinvA, lmaxA;
  int singular;
  double *Lval, *Ml;      /* P1 stiffness on the vertex graph, lumped mass (geometry) */
  double *ccMl, cc_alpha, cc_beta;
  uint8_t *ccPbc;
  int hL_singular;
  int nthreads;
  char err[256];
  /* projected initial guess (orc_opts.ksp_guess): per Newton index the kept corrections [gm vectors] and this step's iterate */
  double *gU[4], *gX[4];
  int gcnt[4], ghead[4], gm, gstored[4];
} orc_ctx;

typedef struct {
  double snes_rtol, snes_atol, snes_stol;
  int snes_max_it;
  double ksp_rtol, ksp_atol;
  int ksp_max_it, ksp_restart;
  double sub_rtol; /* inner GMRES(30) on A00 */
  int sub_max_it, sub_restart;
  int remove_p_mean; /* nullsp.remove(x_n), stabilized_schur.py:319 */
  int verbose;
  /* pc_kind 0: the reference's sub-solvers (GMRES(30)+ILU(0) on A00, ILU(0) on Sp);
   * pc_kind 1: CPU port of the sub-solvers the GPU path uses inside the same Schur
   * factorisation (Jacobi-Chebyshev on A00, one smoothed-aggregation V-cycle on the
   * lagged Sp) -- used where ILU(0) stops converging (>~50k vertices) and as the
   * like-for-like CPU baseline of bench.py */
  int pc_kind;
  int cheb_degree;
  double cheb_ratio;
  int amg_smooth_degree;
  double amg_smooth_ratio, amg_theta;
  int amg_max_coarse;
  int cc_smooth_degree; /* pc_kind 2: Chebyshev steps on H */
  int schur_upper; /* pc_kind 2: block upper-triangular factor (z_p = S^-1 r_p; z_u = A^-1 (r_u - A01 z_p)), the GPU default */
  int ksp_guess;   /* > 0: projected initial guess of the linear solves from the corrections of the last ksp_guess steps (the CPU
                    * port of cfdh_options.ksp_guess, csrc/cfdh_solver.cpp::guess_project); 0 (default): zero guess, as the reference's KSP */
} orc_opts;

typedef struct {
  int newton_its, krylov_its, reason, sub_its;
  double fnorm0, fnorm;
  double ms_assemble, ms_solve;
} orc_stats;

static double now_ms(void) {
#ifdef _OPENMP
  return omp_get_wtime() * 1e3;
#else
  return 0.0;
#endif
}

/* ------------------------------------------------------------------ element */

static void geom(const double xe[3][2], double g[3][2], double *area, double *h) {
  double x0 = xe[0][0], y0 = xe[0][1], x1 = xe[1][0], y1 = xe[1][1], x2 = xe[2][0], y2 = xe[2][1];
  double det = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0);
  g[0][0] = (y1 - y2) / det; g[0][1] = (x2 - x1) / det;
  g[1][0] = (y2 - y0) / det; g[1][1] = (x0 - x2) / det;
  g[2][0] = (y0 - y1) / det; g[2][1] = (x1 - x0) / det;
  *area = 0.5 * fabs(det);
  double d01 = hypot(x0 - x1, y0 - y1), d12 = hypot(x1 - x2, y1 - y2), d20 = hypot(x2 - x0, y2 - y0);
  *h = fmax(d01, fmax(d12, d20));
}

/* M_ab = int tau l_a l_b (6 packed), L = int tau_L; stabilized_schur.py:100-118 */
static void moments(const double une[3][2], double area, double h, double dt, double nu, double mom[7]) {
  double t2 = 4.0 / (dt * dt), t3 = 16.0 * nu * nu / (h * h * h * h), ih2 = 1.0 / (h * h);
  double m[6] = {0, 0, 0, 0, 0, 0}, L = 0;
  for (int q = 0; q < CFDH_NQ; q++) {
    const double *l = CFDH_QL[q];
    double ux = l[0] * une[0][0] + l[1] * une[1][0] + l[2] * une[2][0];
    double uy = l[0] * une[0][1] + l[1] * une[1][1] + l[2] * une[2][1];
    double s = ux * ux + uy * uy;
    double t1 = fmax(4.0 * s, EPS_VNORM * EPS_VNORM) * ih2;
    double tau = 1.0 / sqrt(t1 + t2 + t3);
    double vn = sqrt(s);
    double Re = vn * h / (2.0 * nu);
    double z = (Re <= 3.0) ? Re / 3.0 : 1.0;
    double tl = vn * h * z * 0.5;
    double w = CFDH_QW[q] * tau;
    m[0] += w * l[0] * l[0]; m[1] += w * l[0] * l[1]; m[2] += w * l[0] * l[2];
    m[3] += w * l[1] * l[1]; m[4] += w * l[1] * l[2]; m[5] += w * l[2] * l[2];
    L += CFDH_QW[q] * tl;
  }
  for (int k = 0; k < 6; k++) mom[k] = area * m[k];
  mom[6] = area * L;
}

/* Element residual Fe[9] / Jacobian Je[9][9]; local order (a,i)->2a+i, p_a->6+a. */
static void element(const orc_ctx *c, const double xe[3][2], const double ue[3][2], const double une[3][2],
                    const double un2e[3][2], const double pe[3], const double mom[7], int fflag, double Fe[9], double *Je /*81 or NULL*/) {
  const double rho = c->rho, mu = c->mu, dt = c->dt, muf = c->muf;
  const double th = c->theta, a0 = c->a0, a1 = c->a1, a2 = c->a2;
  double g[3][2], area, h;
  geom(xe, g, &area, &h);
  double M[3][3] = {{mom[0], mom[1], mom[2]}, {mom[1], mom[3], mom[4]}, {mom[2], mom[4], mom[5]}};
  double Lm = mom[6];
  double ub[3][2], w[3][2], G[2][2] = {{0, 0}, {0, 0}}, gp[2] = {0, 0};
  for (int a = 0; a < 3; a++)
    for (int i = 0; i < 2; i++) {
      ub[a][i] = th * ue[a][i] + (1.0 - th) * une[a][i];
      w[a][i] = (a0 * ue[a][i] + a1 * une[a][i] + a2 * un2e[a][i]) / dt;
    }
  for (int a = 0; a < 3; a++)
    for (int i = 0; i < 2; i++) {
      gp[i] += pe[a] * g[a][i];
      for (int j = 0; j < 2; j++) G[i][j] += g[a][i] * ub[a][j];
    }
  double divu = G[0][0] + G[1][1];
  double Cn[3][2], R[3][2], beta[3][3], mab[3][3], mt[3], T = 0, Q[3][2];
  for (int a = 0; a < 3; a++)
    for (int j = 0; j < 2; j++) {
      Cn[a][j] = ub[a][0] * G[0][j] + ub[a][1] * G[1][j];
      R[a][j] = rho * (w[a][j] + Cn[a][j]) + gp[j] - rho * c->f[j];
    }
  for (int d = 0; d < 3; d++)
    for (int a = 0; a < 3; a++) {
      beta[d][a] = ub[d][0] * g[a][0] + ub[d][1] * g[a][1];
      mab[d][a] = area * (d == a ? 2.0 : 1.0) / 12.0;
    }
  for (int b = 0; b < 3; b++) { mt[b] = M[b][0] + M[b][1] + M[b][2]; T += mt[b]; }
  for (int d = 0; d < 3; d++)
    for (int i = 0; i < 2; i++) Q[d][i] = M[0][d] * R[0][i] + M[1][d] * R[1][i] + M[2][d] * R[2][i];
  double E[2][2] = {{G[0][0], 0.5 * (G[0][1] + G[1][0])}, {0.5 * (G[0][1] + G[1][0]), G[1][1]}};
  double pbar = (pe[0] + pe[1] + pe[2]) / 3.0;
  for (int a = 0; a < 3; a++) {
    for (int i = 0; i < 2; i++) {
      double v = 0;
      for (int b = 0; b < 3; b++) v += rho * mab[a][b] * (w[b][i] + Cn[b][i]);
      v -= rho * c->f[i] * area / 3.0;
      v += area * (2.0 * mu * (E[i][0] * g[a][0] + E[i][1] * g[a][1]) - pbar * g[a][i]);
      for (int d = 0; d < 3; d++) v += beta[d][a] * Q[d][i];
      v += rho * Lm * divu * g[a][i];
      Fe[2 * a + i] = v;
    }
    double v = area / 3.0 * divu;
    for (int b = 0; b < 3; b++) v += mt[b] * (R[b][0] * g[a][0] + R[b][1] * g[a][1]) / rho;
    Fe[6 + a] = v;
  }
  if (Je) {
    double MB[3][3], BMB[3][3], mB[3][3], mtB[3], gg[3][3];
    for (int b = 0; b < 3; b++)
      for (int a = 0; a < 3; a++) {
        MB[b][a] = M[b][0] * beta[0][a] + M[b][1] * beta[1][a] + M[b][2] * beta[2][a];
        mB[a][b] = mab[a][0] * beta[0][b] + mab[a][1] * beta[1][b] + mab[a][2] * beta[2][b];
        gg[a][b] = g[a][0] * g[b][0] + g[a][1] * g[b][1];
      }
    for (int b = 0; b < 3; b++)
      for (int a = 0; a < 3; a++) BMB[b][a] = beta[0][b] * MB[0][a] + beta[1][b] * MB[1][a] + beta[2][b] * MB[2][a];
    for (int a = 0; a < 3; a++) mtB[a] = mt[0] * beta[0][a] + mt[1] * beta[1][a] + mt[2] * beta[2][a];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        for (int i = 0; i < 2; i++) {
          for (int j = 0; j < 2; j++) {
            double dij = (i == j) ? 1.0 : 0.0;
            double v = rho * mab[a][b] * dij * a0 / dt;
            v += rho * th * (mab[a][b] * G[j][i] + dij * mB[a][b]);
            v += area * mu * th * (g[b][i] * g[a][j] + gg[a][b] * dij);
            v += rho * ((dij * a0 / dt + th * G[j][i]) * MB[b][a] + th * dij * BMB[b][a]);
            v += th * g[a][j] * Q[b][i];
            v += rho * Lm * th * g[b][j] * g[a][i];
            Je[(2 * a + i) * 9 + 2 * b + j] = v;
          }
          Je[(2 * a + i) * 9 + 6 + b] = -area / 3.0 * g[a][i] + g[b][i] * mtB[a];
        }
        for (int j = 0; j < 2; j++) {
          double Gg = G[j][0] * g[a][0] + G[j][1] * g[a][1];
          double v = area / 3.0 * th * g[b][j];
          v += mt[b] * (g[a][j] * a0 / dt + th * Gg);
          v += th * g[a][j] * mtB[b];
          Je[(6 + a) * 9 + 2 * b + j] = v;
        }
        Je[(6 + a) * 9 + 6 + b] = T * gg[a][b] / rho;
      }
  }
  /* backflow stabilisation on outlet facets: - beta rho oint (u_prev.n)_- (ubar.v), (s)_- = (s-|s|)/2
   * (stabilized_schur_backflow.py:165-176).  Estimated degree 1+1+1 = 3 (UFL: degree(abs(x)) = degree(x))
   * -> 2-point Gauss-Legendre on the facet. */
  if (c->beta_bf != 0.0 && (fflag >> 3)) {
    const double gq = 0.5 / sqrt(3.0);
    for (int f = 0; f < 3; f++) {
      if (!((fflag >> (3 + f)) & 1)) continue;
      double gl = hypot(g[f][0], g[f][1]);
      double n[2] = {-g[f][0] / gl, -g[f][1] / gl};
      double elen = 2.0 * area * gl;
      int ev[2] = {(f + 1) % 3, (f + 2) % 3};
      double s1 = une[ev[0]][0] * n[0] + une[ev[0]][1] * n[1], s2 = une[ev[1]][0] * n[0] + une[ev[1]][1] * n[1];
      for (int q = 0; q < 2; q++) {
        double t = q == 0 ? 0.5 - gq : 0.5 + gq;
        double lam[2] = {1.0 - t, t};
        double sq = lam[0] * s1 + lam[1] * s2;
        double cq = c->beta_bf * rho * 0.5 * (sq - fabs(sq)) * 0.5 * elen;
        double uq[2] = {lam[0] * ub[ev[0]][0] + lam[1] * ub[ev[1]][0], lam[0] * ub[ev[0]][1] + lam[1] * ub[ev[1]][1]};
        for (int ka = 0; ka < 2; ka++)
          for (int i = 0; i < 2; i++) {
            Fe[2 * ev[ka] + i] -= cq * lam[ka] * uq[i];
            if (Je)
              for (int kb = 0; kb < 2; kb++) Je[(2 * ev[ka] + i) * 9 + 2 * ev[kb] + i] -= th * cq * lam[ka] * lam[kb];
          }
      }
    }
  }
  /* exterior facets: + oint p n.v - oint mu_f (nabla_grad(ubar) n).v   (stabilized_schur.py:79) */
  for (int f = 0; f < 3 && c->ds_terms; f++) {
    if (!((fflag >> f) & 1)) continue;
    double gl = hypot(g[f][0], g[f][1]);
    double n[2] = {-g[f][0] / gl, -g[f][1] / gl};
    double elen = 2.0 * area * gl;
    double Gn[2] = {G[0][0] * n[0] + G[0][1] * n[1], G[1][0] * n[0] + G[1][1] * n[1]};
    int ev[2] = {(f + 1) % 3, (f + 2) % 3};
    for (int ka = 0; ka < 2; ka++) {
      int a = ev[ka];
      double pint = 0;
      for (int kb = 0; kb < 2; kb++) pint += pe[ev[kb]] * (ev[kb] == a ? 2.0 : 1.0) / 6.0;
      for (int i = 0; i < 2; i++) {
        Fe[2 * a + i] += n[i] * elen * pint - muf * Gn[i] * elen * 0.5;
        if (Je) {
          for (int kb = 0; kb < 2; kb++) {
            int b = ev[kb];
            Je[(2 * a + i) * 9 + 6 + b] += n[i] * elen * (a == b ? 2.0 : 1.0) / 6.0;
          }
          for (int b = 0; b < 3; b++)
            for (int j = 0; j < 2; j++) Je[(2 * a + i) * 9 + 2 * b + j] -= muf * th * g[b][i] * n[j] * elen * 0.5;
        }
      }
    }
  }
}

/* exported for unit tests: one element, moments computed here */
void orc_element(double dt, double rho, double mu, double muf, const double *f, const double *xe, const double *ue,
                 const double *une, const double *pe, int fflag, double *Fe, double *Je) {
  orc_ctx c;
  memset(&c, 0, sizeof c);
  c.dt = dt; c.rho = rho; c.mu = mu; c.muf = muf; c.f[0] = f[0]; c.f[1] = f[1];
  c.theta = 0.5; c.a0 = 1.0; c.a1 = -1.0; c.a2 = 0.0; c.ds_terms = 1;
  double g[3][2], area, h, mom[7];
  geom((const double(*)[2])xe, g, &area, &h);
  moments((const double(*)[2])une, area, h, dt, mu / rho, mom);
  element(&c, (const double(*)[2])xe, (const double(*)[2])ue, (const double(*)[2])une, (const double(*)[2])une, pe,
          mom, fflag, Fe, Je);
}

/* ------------------------------------------------------------------ setup */

static int cmp_int(const void *a, const void *b) { return (*(const int *)a > *(const int *)b) - (*(const int *)a < *(const int *)b); }

typedef struct { double dt, rho, mu, muf, f[2], theta, a0, a1, a2, beta; int32_t ds_terms, pad; } orcg_params_t;
typedef struct { double dt, rho, mu, muf, f[3], theta, a0, a1, a2, beta; int32_t ds_terms, pad; } orcg3_params_t;
void orcg_element_tensors(int et, int64_t nc, const int64_t *cells, const double *x, const double *u, const double *un, const double *un2,
                          const double *p, const orcg_params_t *P, const uint16_t *flags, int want_jac, double *Fe, double *Je);
void orcg3_element_tensors(int et, int64_t nc, const int64_t *cells, const double *x, const double *u, const double *un, const double *un2,
                           const double *p, const orcg3_params_t *P, const uint16_t *flags, int want_jac, double *Fe, double *Je);
/* per-cell stiffness [nc][NC*NC] and diagonal of the consistent mass [nc][NC] of a generic element, cell measures (Cahouet-Chabard set-up) */
void orcg_stiff_mass(int et, int64_t nc, const int64_t *cells, const double *x, double *K, double *Md, double *meas);
void orcg3_stiff_mass(int et, int64_t nc, const int64_t *cells, const double *x, double *K, double *Md, double *meas);
int orcg_facet_nodes(int et, int f, int *out);   /* local nodes of local facet f (2-D elements); returns their number */
int orcg3_facet_nodes(int et, int f, int *out);  /* ... 3-D elements */
int orc3_element_tensors(int nc, const double *x, const int *cells, const double *u, const double *un, const double *un2,
                         const double *p, const unsigned char *facet_flags, double dt, double rho, double mu, double muf,
                         const double *f, double theta, double a0, double a1, double a2, int ds_terms, double beta_bf,
                         double *Fe, double *Je);

static orc_ctx *create_common(int D, int etg, int nv, int nc, const int *cells, const double *x, int nf, const int *fcell, const int *flocal) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof *c);
  const int NU = D + 1;  /* unknowns per node */
  const int NL = etg == 0 ? D + 1 : (etg == 1 ? 6 : (etg == 2 ? 4 : (etg == 4 ? 10 : 8)));  /* nodes per cell, from here to the end of this function */
  const int ND = NU * NL;  /* element dofs */
  c->D = D; c->NL = NU; c->NC = NL; c->etg = etg;
  c->nv = nv; c->nc = nc; c->nf = nf; c->ndof = NU * nv;
  c->cells = (int *)malloc(sizeof(int) * NL * nc); memcpy(c->cells, cells, sizeof(int) * NL * nc);
  if (etg) {
    c->cells64 = (int64_t *)malloc(sizeof(int64_t) * NL * (size_t)nc);
    for (size_t k = 0; k < (size_t)NL * nc; k++) c->cells64[k] = cells[k];
    c->gflag = (uint16_t *)calloc(nc, sizeof(uint16_t));
    for (int k = 0; k < nf; k++) c->gflag[fcell[k]] |= (uint16_t)(1u << flocal[k]);
  }
  c->x = (double *)malloc(sizeof(double) * D * nv); memcpy(c->x, x, sizeof(double) * D * nv);
  c->fflag = (uint8_t *)calloc(nc, 1);
  c->fcell = (int *)malloc(sizeof(int) * (nf + 1)); c->flocal = (int *)malloc(sizeof(int) * (nf + 1));
  for (int k = 0; k < nf; k++) { c->fcell[k] = fcell[k]; c->flocal[k] = flocal[k]; c->fflag[fcell[k]] |= (uint8_t)(1u << flocal[k]); }
  c->isbc = (uint8_t *)calloc(c->ndof, 1);
  c->bcval = (double *)calloc(c->ndof, sizeof(double));
  c->bcmult = (double *)calloc(c->ndof, sizeof(double));
  c->un = (double *)calloc(D * nv, sizeof(double));
  c->un2 = (double *)calloc(D * nv, sizeof(double));
  c->Mom = (double *)calloc((size_t)7 * nc, sizeof(double));
  c->Fe = (double *)malloc(sizeof(double) * ND * (size_t)nc);
  c->Je = (double *)malloc(sizeof(double) * ND * ND * (size_t)nc);
  c->rho = 1; c->mu = 1; c->muf = 1; c->dt = 1;
  c->theta = 0.5; c->a0 = 1.0; c->a1 = -1.0; c->a2 = 0.0; c->ds_terms = 1; c->beta_bf = 0.0;
#ifdef _OPENMP
  /* never oversubscribe a cgroup-limited box: default to <= 8 threads unless told otherwise */
  c->nthreads = omp_get_max_threads();
  {
    const char *e = getenv("CFDH_ORACLE_THREADS");
    int cap = e ? atoi(e) : 8;
    if (cap < 1) cap = 1;
    if (c->nthreads > cap) c->nthreads = cap;
    omp_set_num_threads(c->nthreads);
  }
#else
  c->nthreads = 1;
#endif
  /* vertex -> cells */
  c->vcptr = (int *)calloc(nv + 1, sizeof(int));
  for (int k = 0; k < NL * nc; k++) c->vcptr[c->cells[k] + 1]++;
  for (int v = 0; v < nv; v++) c->vcptr[v + 1] += c->vcptr[v];
  c->vcell = (int *)malloc(sizeof(int) * NL * nc);
  int *fill = (int *)calloc(nv, sizeof(int));
  for (int k = 0; k < NL * nc; k++) { int v = c->cells[k]; c->vcell[c->vcptr[v] + fill[v]++] = k; }
  free(fill);
  /* vertex graph (neighbours incl. self, sorted) */
  c->vptr = (int *)calloc(nv + 1, sizeof(int));
  int *tmp = (int *)malloc(sizeof(int) * (4 * 64 + 8));
  int cap = 0;
  for (int pass = 0; pass < 2; pass++) {
    for (int v = 0; v < nv; v++) {
      int n = 0;
      int deg = c->vcptr[v + 1] - c->vcptr[v];
      if (NL * deg + 1 > cap) { cap = NL * deg + 64; tmp = (int *)realloc(tmp, sizeof(int) * cap); }
      tmp[n++] = v;
      for (int k = c->vcptr[v]; k < c->vcptr[v + 1]; k++) {
        int cell = c->vcell[k] / NL;
        for (int a = 0; a < NL; a++) tmp[n++] = c->cells[NL * cell + a];
      }
      qsort(tmp, n, sizeof(int), cmp_int);
      int m = 0;
      for (int k = 0; k < n; k++) if (k == 0 || tmp[k] != tmp[k - 1]) tmp[m++] = tmp[k];
      if (pass == 0) c->vptr[v + 1] = c->vptr[v] + m;
      else memcpy(c->vadj + c->vptr[v], tmp, sizeof(int) * m);
    }
    if (pass == 0) c->vadj = (int *)malloc(sizeof(int) * c->vptr[nv]);
  }
  free(tmp);
  /* monolithic CSR: row of dof has [D*w+j for w in N(v)] then [D nv+w] */
  int nu = D * nv;
  c->rowptr = (int *)malloc(sizeof(int) * (c->ndof + 1));
  c->rowptr[0] = 0;
  for (int r = 0; r < c->ndof; r++) {
    int v = r < nu ? r / D : r - nu;
    c->rowptr[r + 1] = c->rowptr[r] + NU * (c->vptr[v + 1] - c->vptr[v]);
  }
  c->nnz = c->rowptr[c->ndof];
  c->col = (int *)malloc(sizeof(int) * c->nnz);
  c->val = (double *)calloc(c->nnz, sizeof(double));
  for (int r = 0; r < c->ndof; r++) {
    int v = r < nu ? r / D : r - nu;
    int deg = c->vptr[v + 1] - c->vptr[v];
    int *cc = c->col + c->rowptr[r];
    for (int k = 0; k < deg; k++) {
      int w = c->vadj[c->vptr[v] + k];
      for (int j = 0; j < D; j++) cc[D * k + j] = D * w + j;
      cc[D * deg + k] = nu + w;
    }
  }
  c->cellpos = (int *)malloc(sizeof(int) * NL * NL * (size_t)nc);
  for (int e = 0; e < nc; e++)
    for (int a = 0; a < NL; a++) {
      int va = c->cells[NL * e + a];
      for (int b = 0; b < NL; b++) {
        int vb = c->cells[NL * e + b];
        int *base = c->vadj + c->vptr[va];
        int deg = c->vptr[va + 1] - c->vptr[va];
        int *p = (int *)bsearch(&vb, base, deg, sizeof(int), cmp_int);
        c->cellpos[NL * NL * e + NL * a + b] = (int)(p - base);
      }
    }
  return c;
}

orc_ctx *orc_create_d(int D, int nv, int nc, const int *cells, const double *x, int nf, const int *fcell, const int *flocal) {
  return create_common(D, 0, nv, nc, cells, x, nf, fcell, flocal);
}
/* nodal equal-order elements beyond P1: etg 1 P2 triangles, 2 Q1 parallelograms, 4 P2 tetrahedra, 5 Q1 hexahedra;
 * cells [nc][nodes per cell] in the DOLFINx local order, x = node coordinates.  pc_kind 2 only. */
orc_ctx *orc_create_gen(int D, int etg, int nv, int nc, const int *cells, const double *x, int nf, const int *fcell, const int *flocal) {
  if (!((D == 2 && (etg == 1 || etg == 2)) || (D == 3 && (etg == 4 || etg == 5)))) return NULL;
  return create_common(D, etg, nv, nc, cells, x, nf, fcell, flocal);
}
orc_ctx *orc_create(int nv, int nc, const int *cells, const double *x, int nf, const int *fcell, const int *flocal) {
  return orc_create_d(2, nv, nc, cells, x, nf, fcell, flocal);
}

static void amg_free(struct orc_ctx *c);
static void free_csr(csr_t *m) { free(m->rowptr); free(m->col); free(m->val); free(m->diag); memset(m, 0, sizeof *m); }

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  free(c->cells64); free(c->gflag);
  free(c->cells); free(c->x); free(c->fflag); free(c->fcell); free(c->flocal); free(c->isbc); free(c->bcval);
  free(c->bcmult); free(c->un); free(c->un2); free(c->Mom); free(c->Fe); free(c->Je); free(c->vcptr); free(c->vcell);
  free(c->vptr); free(c->vadj); free(c->rowptr); free(c->col); free(c->val); free(c->cellpos);
  free_csr(&c->A00f); free_csr(&c->Sp); free(c->sp_rowptr); free(c->sp_col);
  amg_free(c); free(c->dinvA);
  for (int k = 0; k < 4; k++) { free(c->gU[k]); free(c->gX[k]); }
  free(c);
}

void orc_set_params(orc_ctx *c, double dt, double rho, double mu, double muf, const double *f) {
  c->dt = dt; c->rho = rho; c->mu = mu; c->muf = muf; c->f[0] = f[0]; c->f[1] = f[1]; c->f[2] = c->D == 3 ? f[2] : 0.0;
}
/* time scheme (see orc_ctx) */
void orc_set_scheme(orc_ctx *c, double theta, double a0, double a1, double a2) {
  c->theta = theta; c->a0 = a0; c->a1 = a1; c->a2 = a2;
}
/* boundary terms (see orc_ctx): nbf facets (indices into the exterior-facet arrays) carry the backflow term */
void orc_set_boundary_terms(orc_ctx *c, int ds_terms, double beta, int nbf, const int *bf_facets) {
  c->ds_terms = ds_terms; c->beta_bf = beta;
  /* bits 0..D: exterior facet, bits D+1..2D+1: backflow facet (cfdh_oracle3.c: bit 4 + f) */
  const unsigned ext = c->D == 3 ? 15u : 7u, bf0 = c->D == 3 ? 16u : 8u;
  for (int e = 0; e < c->nc; e++) c->fflag[e] &= ext;
  for (int k = 0; k < nbf; k++) c->fflag[c->fcell[bf_facets[k]]] |= (uint8_t)(bf0 << c->flocal[bf_facets[k]]);
  if (c->gflag) {
    for (int e = 0; e < c->nc; e++) c->gflag[e] &= 0xffu;
    for (int k = 0; k < nbf; k++) c->gflag[c->fcell[bf_facets[k]]] |= (uint16_t)(256u << c->flocal[bf_facets[k]]);
  }
}
/* u_prev2 of stabilized_schur_bdf2.py:72,324 */
void orc_set_un2(orc_ctx *c, const double *un2) { memcpy(c->un2, un2, sizeof(double) * c->D * c->nv); }

void orc_set_threads(orc_ctx *c, int n) {
#ifdef _OPENMP
  if (n > 0) { omp_set_num_threads(n); c->nthreads = n; }
#else
  (void)c; (void)n;
#endif
}
int orc_get_threads(orc_ctx *c) { return c->nthreads; }

void orc_clear_bcs(orc_ctx *c) {
  memset(c->isbc, 0, c->ndof); memset(c->bcval, 0, sizeof(double) * c->ndof); memset(c->bcmult, 0, sizeof(double) * c->ndof);
  c->any_pbc = 0;
}
/* one DirichletBC object; field 0: velocity (values [n][2]), 1: pressure (values [n]).
 * Later objects overwrite the value; the diagonal counts the objects. */
void orc_add_bc(orc_ctx *c, int field, int n, const int *nodes, const double *vals) {
  for (int k = 0; k < n; k++) {
    if (field == 0) {
      for (int i = 0; i < c->D; i++) { int d = c->D * nodes[k] + i; c->isbc[d] = 1; c->bcval[d] = vals[c->D * k + i]; c->bcmult[d] += 1.0; }
    } else {
      int d = c->D * c->nv + nodes[k]; c->isbc[d] = 1; c->bcval[d] = vals[k]; c->bcmult[d] += 1.0; c->any_pbc = 1;
    }
  }
}

/* u_prev for the step; refreshes the tau moments (they depend on u_prev only) */
void orc_set_un(orc_ctx *c, const double *un) {
  memcpy(c->un, un, sizeof(double) * c->D * c->nv);
  if (c->D == 3 || c->etg) return;  /* the tetrahedral and the generic element routines integrate tau themselves */
  double nu = c->mu / c->rho;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < c->nc; e++) {
    double xe[3][2], une[3][2], g[3][2], area, h;
    for (int a = 0; a < 3; a++) {
      int v = c->cells[3 * e + a];
      xe[a][0] = c->x[2 * v]; xe[a][1] = c->x[2 * v + 1];
      une[a][0] = c->un[2 * v]; une[a][1] = c->un[2 * v + 1];
    }
    geom(xe, g, &area, &h);
    moments(une, area, h, c->dt, nu, c->Mom + 7 * (size_t)e);
  }
}

/* ------------------------------------------------------------------ assembly */

/* F (always) and the CSR values (want_jac) at monolithic state xv.
 * Dirichlet semantics of assemble_vector_block(F, F_form, J_form, bcs, x0=x, alpha=-1)
 * and assemble_matrix_block(J, J_form, bcs): stabilized_schur.py:144-175. */
void orc_assemble(orc_ctx *c, const double *xv, int want_jac, double *F) {
  const int D = c->D, NL = c->NL, NC = c->NC, ND = NL * NC, PO = D * NC;  /* PO: offset of the pressure dofs in the element vector */
  const int nv = c->nv, nu = D * nv, nc = c->nc;
  int any_lift = 0;
  for (int d = 0; d < c->ndof && !any_lift; d++)
    if (c->isbc[d] && c->bcval[d] != xv[d]) any_lift = 1;
  const int need_j = want_jac || any_lift;
  if (c->etg) { /* generic elements: element tensors of all cells by cfdh_oracle_gen.c / cfdh_oracle_gen3.c */
    if (D == 2) {
      orcg_params_t P = {c->dt, c->rho, c->mu, c->muf, {c->f[0], c->f[1]}, c->theta, c->a0, c->a1, c->a2, c->beta_bf, c->ds_terms, 0};
      orcg_element_tensors(c->etg, nc, c->cells64, c->x, xv, c->un, c->a2 != 0.0 ? c->un2 : NULL, xv + nu, &P, c->gflag, need_j, c->Fe, need_j ? c->Je : NULL);
    } else {
      orcg3_params_t P = {c->dt, c->rho, c->mu, c->muf, {c->f[0], c->f[1], c->f[2]}, c->theta, c->a0, c->a1, c->a2, c->beta_bf, c->ds_terms, 0};
      orcg3_element_tensors(c->etg, nc, c->cells64, c->x, xv, c->un, c->a2 != 0.0 ? c->un2 : NULL, xv + nu, &P, c->gflag, need_j, c->Fe, need_j ? c->Je : NULL);
    }
  } else if (D == 3) /* element tensors of all cells by the tetrahedral restatement (cfdh_oracle3.c) */
    orc3_element_tensors(nc, c->x, c->cells, xv, c->un, c->a2 != 0.0 ? c->un2 : NULL, xv + nu, c->fflag, c->dt, c->rho, c->mu, c->muf, c->f,
                         c->theta, c->a0, c->a1, c->a2, c->ds_terms, c->beta_bf, c->Fe, need_j ? c->Je : NULL);
#pragma omp parallel for schedule(static)
  for (int e = 0; e < nc; e++) {
    int ld[40];
    for (int a = 0; a < NC; a++) {
      int v = c->cells[NC * e + a];
      for (int i = 0; i < D; i++) ld[D * a + i] = D * v + i;
      ld[PO + a] = nu + v;
    }
    double *Fe = c->Fe + ND * (size_t)e, *Je = c->Je + (size_t)ND * ND * e;
    if (D == 2 && !c->etg) {
      double xe[3][2], ue[3][2], une[3][2], un2e[3][2], pe[3];
      for (int a = 0; a < 3; a++) {
        int v = c->cells[3 * e + a];
        xe[a][0] = c->x[2 * v]; xe[a][1] = c->x[2 * v + 1];
        ue[a][0] = xv[2 * v]; ue[a][1] = xv[2 * v + 1];
        une[a][0] = c->un[2 * v]; une[a][1] = c->un[2 * v + 1];
        un2e[a][0] = c->un2[2 * v]; un2e[a][1] = c->un2[2 * v + 1];
        pe[a] = xv[nu + v];
      }
      element(c, xe, ue, une, un2e, pe, c->Mom + 7 * (size_t)e, c->fflag[e], Fe, need_j ? Je : NULL);
    }
    int anybc = 0;
    for (int k = 0; k < ND; k++) anybc |= c->isbc[ld[k]];
    if (anybc) {
      if (any_lift)
        for (int k = 0; k < ND; k++)
          if (c->isbc[ld[k]]) {
            double gx = c->bcval[ld[k]] - xv[ld[k]];
            if (gx != 0.0)
              for (int r = 0; r < ND; r++) Fe[r] += Je[r * ND + k] * gx;
          }
      for (int k = 0; k < ND; k++)
        if (c->isbc[ld[k]]) {
          Fe[k] = 0.0;
          if (need_j)
            for (int r = 0; r < ND; r++) { Je[k * ND + r] = 0.0; Je[r * ND + k] = 0.0; }
        }
    }
  }
  /* row gather: deterministic, race free */
#pragma omp parallel for schedule(static)
  for (int v = 0; v < nv; v++) {
    int deg = c->vptr[v + 1] - c->vptr[v];
    double fv[4] = {0, 0, 0, 0};
    int rows[4];
    double *rr[4];
    for (int t = 0; t < D; t++) rows[t] = D * v + t;
    rows[D] = nu + v;
    for (int t = 0; t < NL; t++) {
      rr[t] = c->val + c->rowptr[rows[t]];
      if (want_jac) memset(rr[t], 0, sizeof(double) * NL * deg);
    }
    for (int k = c->vcptr[v]; k < c->vcptr[v + 1]; k++) {
      int e = c->vcell[k] / NC, a = c->vcell[k] % NC;
      const double *Fe = c->Fe + ND * (size_t)e, *Je = c->Je + (size_t)ND * ND * e;
      int er[4];  /* element rows of vertex a: velocity components, pressure */
      for (int t = 0; t < D; t++) er[t] = D * a + t;
      er[D] = PO + a;
      for (int t = 0; t < NL; t++) fv[t] += Fe[er[t]];
      if (want_jac)
        for (int b = 0; b < NC; b++) {
          int kb = c->cellpos[NC * NC * e + NC * a + b];
          for (int t = 0; t < NL; t++) {
            for (int j = 0; j < D; j++) rr[t][D * kb + j] += Je[er[t] * ND + D * b + j];
            rr[t][D * deg + kb] += Je[er[t] * ND + PO + b];
          }
        }
    }
    /* position of the diagonal inside the row */
    int kd = 0;
    while (c->vadj[c->vptr[v] + kd] != v) kd++;
    for (int t = 0; t < NL; t++) {
      int d = rows[t];
      if (c->isbc[d]) {
        fv[t] = xv[d] - c->bcval[d];
        if (want_jac) rr[t][t < D ? D * kd + t : D * deg + kd] = c->bcmult[d];
      }
      F[d] = fv[t];
    }
  }
}

void orc_get_csr(orc_ctx *c, int *nnz, int **rowptr, int **col, double **val) {
  *nnz = c->nnz; *rowptr = c->rowptr; *col = c->col; *val = c->val;
}

/* ------------------------------------------------------------------ linear algebra */

static double vdot(int n, const double *a, const double *b) {
  double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
  for (int i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}
static double vnorm(int n, const double *a) { return sqrt(vdot(n, a, a)); }
static void vaxpy(int n, double al, const double *x, double *y) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) y[i] += al * x[i];
}
static void vcopy(int n, const double *x, double *y) { memcpy(y, x, sizeof(double) * n); }
static void vscale(int n, double al, double *x) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) x[i] *= al;
}

/* y = A[rows r0..r1, cols c0..c1) x : sub-block of the monolithic CSR.
 * columns of a row are sorted with all velocity columns first, so a block is a
 * contiguous slice of the row. */
typedef struct { const orc_ctx *c; int blk; } blk_t; /* blk: 0=full,1=A00,2=A01,3=A10,4=A11 */

static void blk_range(const orc_ctx *c, int blk, int *r0, int *r1) {
  int nu = c->D * c->nv;
  if (blk == 0) { *r0 = 0; *r1 = c->ndof; }
  else if (blk == 1 || blk == 2) { *r0 = 0; *r1 = nu; }
  else { *r0 = nu; *r1 = c->ndof; }
}
static void blk_mult(const orc_ctx *c, int blk, const double *x, double *y) {
  const int D = c->D;
  int nu = D * c->nv, r0, r1;
  blk_range(c, blk, &r0, &r1);
#pragma omp parallel for schedule(static)
  for (int r = r0; r < r1; r++) {
    int v = r < nu ? r / D : r - nu;
    int deg = c->vptr[v + 1] - c->vptr[v];
    int s = c->rowptr[r], k0 = s, k1 = s + (D + 1) * deg, off = 0;
    if (blk == 1 || blk == 3) k1 = s + D * deg;
    if (blk == 2 || blk == 4) { k0 = s + D * deg; off = nu; }
    double acc = 0;
    for (int k = k0; k < k1; k++) acc += c->val[k] * x[c->col[k] - off];
    y[r - r0] = acc;
  }
}

/* ILU(0) of a CSR matrix with sorted columns */
static int ilu0(csr_t *m) {
  int n = m->n;
  if (!m->diag) m->diag = (int *)malloc(sizeof(int) * n);
  int *pos = (int *)malloc(sizeof(int) * n);
  for (int i = 0; i < n; i++) pos[i] = -1;
  for (int i = 0; i < n; i++) {
    int s = m->rowptr[i], e = m->rowptr[i + 1];
    for (int k = s; k < e; k++) pos[m->col[k]] = k;
    m->diag[i] = -1;
    for (int k = s; k < e; k++) {
      int j = m->col[k];
      if (j >= i) { if (j == i) m->diag[i] = k; break; }
      double piv = m->val[m->diag[j]];
      double l = m->val[k] / piv;
      m->val[k] = l;
      for (int kk = m->diag[j] + 1; kk < m->rowptr[j + 1]; kk++) {
        int p = pos[m->col[kk]];
        if (p >= 0) m->val[p] -= l * m->val[kk];
      }
    }
    if (m->diag[i] < 0) { for (int k = s; k < e; k++) if (m->col[k] == i) m->diag[i] = k; }
    if (m->diag[i] < 0 || m->val[m->diag[i]] == 0.0) { free(pos); return -1; }
    for (int k = s; k < e; k++) pos[m->col[k]] = -1;
  }
  free(pos);
  return 0;
}
static void ilu_solve(const csr_t *m, const double *b, double *x) {
  int n = m->n;
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = m->rowptr[i]; k < m->diag[i]; k++) s -= m->val[k] * x[m->col[k]];
    x[i] = s;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = m->diag[i] + 1; k < m->rowptr[i + 1]; k++) s -= m->val[k] * x[m->col[k]];
    x[i] = s / m->val[m->diag[i]];
  }
}


/* ------------------------------------------------------------------ pc_kind 1: Chebyshev + SA-AMG (CPU port) */

typedef struct amg_level {
  csr_t A, P, R;
  double *dinv, *wdinv, *x, *b, *r, *d0, *d1;
  double lmax, lmin;
  int n;
} amg_level;

static void csr_alloc(csr_t *m, int n, int nnz) {
  m->n = n; m->rowptr = (int *)calloc(n + 1, sizeof(int)); m->col = (int *)malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
  m->val = (double *)malloc(sizeof(double) * (nnz > 0 ? nnz : 1)); m->diag = NULL;
}
static void csr_mult(const csr_t *A, const double *x, double *y, int mode, const double *b) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < A->n; i++) {
    double s = 0;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) s += A->val[k] * x[A->col[k]];
    if (mode == 0) y[i] = s; else if (mode == 1) y[i] = b[i] - s; else y[i] += s;
  }
}
/* C = A * B, B has mB columns; sorted columns */
static void csr_spgemm(const csr_t *A, const csr_t *B, int mB, csr_t *C) {
  int n = A->n, cap = A->rowptr[n] * 4 + 16, nn = 0;
  C->n = n; C->rowptr = (int *)calloc(n + 1, sizeof(int)); C->col = (int *)malloc(sizeof(int) * cap);
  C->val = (double *)malloc(sizeof(double) * cap); C->diag = NULL;
  int *mark = (int *)malloc(sizeof(int) * mB);
  double *acc = (double *)malloc(sizeof(double) * mB);
  for (int i = 0; i < mB; i++) mark[i] = -1;
  for (int i = 0; i < n; i++) {
    int start = nn;
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      int j = A->col[k];
      double a = A->val[k];
      for (int k2 = B->rowptr[j]; k2 < B->rowptr[j + 1]; k2++) {
        int cc = B->col[k2];
        if (mark[cc] != i) {
          mark[cc] = i; acc[cc] = 0.0;
          if (nn == cap) { cap *= 2; C->col = (int *)realloc(C->col, sizeof(int) * cap); C->val = (double *)realloc(C->val, sizeof(double) * cap); }
          C->col[nn++] = cc;
        }
        acc[cc] += a * B->val[k2];
      }
    }
    qsort(C->col + start, nn - start, sizeof(int), cmp_int);
    for (int k = start; k < nn; k++) C->val[k] = acc[C->col[k]];
    C->rowptr[i + 1] = nn;
  }
  free(mark); free(acc);
}
static void csr_transpose(const csr_t *A, int m, csr_t *T) {
  int nnz = A->rowptr[A->n];
  csr_alloc(T, m, nnz);
  for (int k = 0; k < nnz; k++) T->rowptr[A->col[k] + 1]++;
  for (int i = 0; i < m; i++) T->rowptr[i + 1] += T->rowptr[i];
  int *fill = (int *)malloc(sizeof(int) * (m + 1));
  memcpy(fill, T->rowptr, sizeof(int) * (m + 1));
  for (int i = 0; i < A->n; i++)
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) { int p = fill[A->col[k]]++; T->col[p] = i; T->val[p] = A->val[k]; }
  free(fill);
}
static double power_lmax(const csr_t *A, const double *dinv, int its) {
  int n = A->n;
  double *v = (double *)malloc(sizeof(double) * n), *w = (double *)malloc(sizeof(double) * n);
  unsigned long long st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; i++) { st = st * 6364136223846793005ull + 1442695040888963407ull; v[i] = ((double)(st >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
  double l = 1;
  for (int it = 0; it < its; it++) {
    csr_mult(A, v, w, 0, NULL);
    double nn = 0;
    for (int i = 0; i < n; i++) { w[i] *= dinv[i]; nn += w[i] * w[i]; }
    l = sqrt(nn);
    if (!(l > 0)) { l = 1; break; }
    for (int i = 0; i < n; i++) v[i] = w[i] / l;
  }
  free(v); free(w);
  return l;
}
static int amg_aggregate(const csr_t *A, double theta, int *agg) {
  int n = A->n;
  double *d = (double *)calloc(n, sizeof(double));
  for (int i = 0; i < n; i++)
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) if (A->col[k] == i) d[i] = fabs(A->val[k]);
  int *sptr = (int *)calloc(n + 1, sizeof(int)), *scol = (int *)malloc(sizeof(int) * (A->rowptr[n] + 1));
  int ns = 0;
  for (int i = 0; i < n; i++) {
    for (int k = A->rowptr[i]; k < A->rowptr[i + 1]; k++) {
      int j = A->col[k];
      if (j != i && fabs(A->val[k]) >= theta * sqrt(d[i] * d[j])) scol[ns++] = j;
    }
    sptr[i + 1] = ns;
  }
  for (int i = 0; i < n; i++) agg[i] = -1;
  int na = 0;
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    int ok = sptr[i + 1] > sptr[i];
    for (int k = sptr[i]; k < sptr[i + 1] && ok; k++) if (agg[scol[k]] >= 0) ok = 0;
    if (!ok) continue;
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) agg[scol[k]] = na;
    na++;
  }
  int *agg2 = (int *)malloc(sizeof(int) * n);
  memcpy(agg2, agg, sizeof(int) * n);
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) if (agg[scol[k]] >= 0) { agg2[i] = agg[scol[k]]; break; }
  }
  memcpy(agg, agg2, sizeof(int) * n);
  free(agg2);
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    if (sptr[i + 1] == sptr[i]) continue; /* isolated unknown (Dirichlet row): no coarse correction */
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) if (agg[scol[k]] < 0) agg[scol[k]] = na;
    na++;
  }
  free(d); free(sptr); free(scol);
  return na;
}
typedef struct amg_hier {
  amg_level lev[16];
  int nlev;
  double *cinv;
  int cn;
} amg_hier;
static void level_free(amg_level *L) {
  free_csr(&L->A); free_csr(&L->P); free_csr(&L->R);
  free(L->dinv); free(L->wdinv); free(L->x); free(L->b); free(L->r); free(L->d0); free(L->d1);
  memset(L, 0, sizeof *L);
}
static void hier_free(amg_hier *H) {
  if (!H) return;
  for (int l = 0; l < H->nlev; l++) level_free(&H->lev[l]);
  free(H->cinv);
  memset(H, 0, sizeof *H);
}
static void amg_free(orc_ctx *c) {
  hier_free(c->hS); hier_free(c->hL); hier_free(c->hA);
  free(c->hS); free(c->hL); free(c->hA); c->hS = c->hL = c->hA = NULL;
  if (c->Hlev) { level_free(c->Hlev); free(c->Hlev); c->Hlev = NULL; }
  free(c->Lval); free(c->Ml); free(c->ccMl); free(c->ccPbc);
  c->Lval = c->Ml = c->ccMl = NULL; c->ccPbc = NULL;
}
static int dense_inv(double *a, int n) {
  double *inv = (double *)calloc((size_t)n * n, sizeof(double));
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int col = 0; col < n; col++) {
    int piv = col;
    double best = fabs(a[(size_t)col * n + col]);
    for (int r = col + 1; r < n; r++) if (fabs(a[(size_t)r * n + col]) > best) { best = fabs(a[(size_t)r * n + col]); piv = r; }
    if (!(best > 0)) { free(inv); return -1; }
    if (piv != col)
      for (int k = 0; k < n; k++) {
        double t = a[(size_t)piv * n + k]; a[(size_t)piv * n + k] = a[(size_t)col * n + k]; a[(size_t)col * n + k] = t;
        t = inv[(size_t)piv * n + k]; inv[(size_t)piv * n + k] = inv[(size_t)col * n + k]; inv[(size_t)col * n + k] = t;
      }
    double d = 1.0 / a[(size_t)col * n + col];
    for (int k = 0; k < n; k++) { a[(size_t)col * n + k] *= d; inv[(size_t)col * n + k] *= d; }
    for (int r = 0; r < n; r++) {
      if (r == col) continue;
      double f = a[(size_t)r * n + col];
      if (f == 0.0) continue;
      for (int k = 0; k < n; k++) { a[(size_t)r * n + k] -= f * a[(size_t)col * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)col * n + k]; }
    }
  }
  memcpy(a, inv, sizeof(double) * (size_t)n * n);
  free(inv);
  return 0;
}
/* operator copy + Jacobi diagonal + spectral bound + work vectors */
static void level_setup(amg_level *L, csr_t A, double ratio) {
  int n = A.n;
  L->n = n; L->A = A;
  L->dinv = (double *)malloc(sizeof(double) * n); L->wdinv = (double *)malloc(sizeof(double) * n);
  for (int i = 0; i < n; i++) {
    L->dinv[i] = 1.0;
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i && A.val[k] != 0.0) L->dinv[i] = 1.0 / A.val[k];
  }
  double lm = power_lmax(&A, L->dinv, 15);
  L->lmax = 1.1 * lm; L->lmin = L->lmax / ratio;
  double itheta = 2.0 / (L->lmax + L->lmin);
  for (int i = 0; i < n; i++) {
    int offd = 0;
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] != i && A.val[k] != 0.0) offd++;
    L->wdinv[i] = L->dinv[i] * (offd == 0 ? 1.0 : itheta); /* diagonal-only rows are solved exactly */
  }
  L->x = (double *)calloc(n, sizeof(double)); L->b = (double *)calloc(n, sizeof(double)); L->r = (double *)calloc(n, sizeof(double));
  L->d0 = (double *)calloc(n, sizeof(double)); L->d1 = (double *)calloc(n, sizeof(double));
}
static csr_t csr_copy(const csr_t *S) {
  csr_t A;
  int nnz = S->rowptr[S->n];
  csr_alloc(&A, S->n, nnz);
  memcpy(A.rowptr, S->rowptr, sizeof(int) * (S->n + 1)); memcpy(A.col, S->col, sizeof(int) * nnz); memcpy(A.val, S->val, sizeof(double) * nnz);
  return A;
}
/* smoothed aggregation hierarchy of S (copied) */
static int amg_setup(orc_ctx *c, amg_hier *H, const csr_t *S, const orc_opts *o, int singular) {
  hier_free(H);
  csr_t A = csr_copy(S);
  for (;;) {
    amg_level *L = &H->lev[H->nlev++];
    int n = A.n;
    level_setup(L, A, o->amg_smooth_ratio);
    double lm = L->lmax / 1.1;
    if (n <= o->amg_max_coarse || H->nlev >= 16) break;
    int *agg = (int *)malloc(sizeof(int) * n);
    int na = amg_aggregate(&A, o->amg_theta, agg);
    if (na >= n || na < 1) { free(agg); break; }
    csr_t P0, AP0, P, R, AP, Ac;
    csr_alloc(&P0, n, n);
    int np0 = 0;
    for (int i = 0; i < n; i++) { P0.rowptr[i] = np0; if (agg[i] >= 0) { P0.col[np0] = agg[i]; P0.val[np0] = 1.0; np0++; } }
    P0.rowptr[n] = np0;
    csr_spgemm(&A, &P0, na, &AP0);
    double omega = 4.0 / 3.0 / lm;
    /* P = P0 - omega D^-1 A P0 on aggregated rows (A P0 contains column agg[i] there); empty rows for isolated unknowns */
    csr_alloc(&P, n, AP0.rowptr[n]);
    int np = 0;
    for (int i = 0; i < n; i++) {
      P.rowptr[i] = np;
      if (agg[i] < 0) continue;
      for (int k = AP0.rowptr[i]; k < AP0.rowptr[i + 1]; k++) {
        P.col[np] = AP0.col[k];
        P.val[np] = -omega * L->dinv[i] * AP0.val[k] + (AP0.col[k] == agg[i] ? 1.0 : 0.0);
        np++;
      }
    }
    P.rowptr[n] = np;
    csr_transpose(&P, na, &R);
    csr_spgemm(&A, &P, na, &AP);
    csr_spgemm(&R, &AP, na, &Ac);
    L->P = P; L->R = R;
    free_csr(&P0); free_csr(&AP0); free_csr(&AP); free(agg);
    A = Ac;
  }
  {
    int n = A.n;
    double *D = (double *)calloc((size_t)n * n, sizeof(double)), tr = 0;
    for (int i = 0; i < n; i++)
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) { D[(size_t)i * n + A.col[k]] = A.val[k]; if (A.col[k] == i) tr += fabs(A.val[k]); }
    if (singular) { double al = tr / n / n; for (size_t k = 0; k < (size_t)n * n; k++) D[k] += al; }
    if (dense_inv(D, n)) { free(D); snprintf(c->err, sizeof c->err, "singular coarsest AMG operator"); return -1; }
    H->cinv = D; H->cn = n;
  }
  return 0;
}
static void cheb_smooth(const csr_t *A, const double *dinv, double lmin, double lmax, int deg, const double *b, double *x,
                        int zero_guess, double *r, double *d0, double *d1) {
  const int n = A->n;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin), sigma = theta / delta;
  double rho = 1.0 / sigma;
  const double *rin = b;
  if (!zero_guess) { csr_mult(A, x, r, 1, b); rin = r; }
  double *dold = d0, *dnew = d1;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) { double v = dinv[i] * rin[i] / theta; dold[i] = v; x[i] = zero_guess ? v : x[i] + v; }
  for (int k = 1; k < deg; k++) {
    double rho_new = 1.0 / (2.0 * sigma - rho), c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) {
      double s = 0;
      for (int kk = A->rowptr[i]; kk < A->rowptr[i + 1]; kk++) s += A->val[kk] * dold[A->col[kk]];
      double rr = rin[i] - s, dn = c1 * dold[i] + c2 * dinv[i] * rr;
      r[i] = rr; dnew[i] = dn; x[i] += dn;
    }
    rin = r;
    double *t = dold; dold = dnew; dnew = t;
    rho = rho_new;
  }
}
static void amg_vcycle(amg_hier *H, const orc_opts *o, int lev, const double *b, double *x) {
  amg_level *L = &H->lev[lev];
  if (lev + 1 == H->nlev) {
    int n = H->cn;
    for (int i = 0; i < n; i++) { double s = 0; for (int k = 0; k < n; k++) s += H->cinv[(size_t)i * n + k] * b[k]; x[i] = s; }
    return;
  }
  amg_level *N = &H->lev[lev + 1];
  if (o->amg_smooth_degree == 1) { /* damped Jacobi, as the GPU's jacobi_pre/post kernels */
    const int n = L->n;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) L->d0[i] = L->wdinv[i] * b[i];
    csr_mult(&L->A, L->d0, L->r, 1, b);
    csr_mult(&L->R, L->r, N->b, 0, NULL);
    amg_vcycle(H, o, lev + 1, N->b, N->x);
    csr_mult(&L->P, N->x, L->d0, 2, NULL);
    csr_mult(&L->A, L->d0, L->r, 1, b);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) x[i] = L->d0[i] + L->wdinv[i] * L->r[i];
    return;
  }
  cheb_smooth(&L->A, L->dinv, L->lmin, L->lmax, o->amg_smooth_degree, b, x, 1, L->r, L->d0, L->d1);
  csr_mult(&L->A, x, L->r, 1, b);
  csr_mult(&L->R, L->r, N->b, 0, NULL);
  amg_vcycle(H, o, lev + 1, N->b, N->x);
  csr_mult(&L->P, N->x, x, 2, NULL);
  cheb_smooth(&L->A, L->dinv, L->lmin, L->lmax, o->amg_smooth_degree, b, x, 0, L->r, L->d0, L->d1);
}

/* pc_kind 2: host side of the Cahouet-Chabard-type preconditioner (same construction as
 * cfd_hemodynamic_amd/csrc/cfdh_solver.cpp::build_cc_host, written independently) */
static int cc_setup(orc_ctx *c, const orc_opts *o) {
  const int D = c->D, NL = c->NL;
  const int nv = c->nv, nu = D * nv;
  if (!c->Lval && c->etg) {
    /* generic elements: stiffness of the element by its own quadrature on the node graph; lumped mass = diagonal of the consistent
     * mass scaled to the total measure (row sums vanish at P2 vertices) -- as csrc/cfdh_gen.hip / cfdh_gen3.hip build them */
    const int NC = c->NC;
    c->Lval = (double *)calloc(c->vptr[nv], sizeof(double));
    c->Ml = (double *)calloc(nv, sizeof(double));
    double *K = (double *)malloc(sizeof(double) * NC * NC * (size_t)c->nc), *Md = (double *)malloc(sizeof(double) * NC * (size_t)c->nc);
    double *meas = (double *)malloc(sizeof(double) * c->nc);
    if (D == 2) orcg_stiff_mass(c->etg, c->nc, c->cells64, c->x, K, Md, meas);
    else orcg3_stiff_mass(c->etg, c->nc, c->cells64, c->x, K, Md, meas);
    double msum = 0.0, dsum = 0.0;
    for (int e = 0; e < c->nc; e++) {
      msum += meas[e];
      for (int a = 0; a < NC; a++) {
        const int va = c->cells[NC * e + a];
        c->Ml[va] += Md[(size_t)NC * e + a];
        dsum += Md[(size_t)NC * e + a];
        for (int b = 0; b < NC; b++) c->Lval[c->vptr[va] + c->cellpos[NC * NC * e + NC * a + b]] += K[((size_t)NC * e + a) * NC + b];
      }
    }
    for (int i = 0; i < nv; i++) c->Ml[i] *= msum / dsum;
    free(K); free(Md); free(meas);
  }
  if (!c->Lval && D == 3) { /* tetrahedra: grad lambda from the inverse of [x1-x0 | x2-x0 | x3-x0], volume |det| / 6 */
    c->Lval = (double *)calloc(c->vptr[nv], sizeof(double));
    c->Ml = (double *)calloc(nv, sizeof(double));
    for (int e = 0; e < c->nc; e++) {
      int vs[4];
      double X[4][3], Jm[3][3], g[4][3];
      for (int a = 0; a < 4; a++) { vs[a] = c->cells[4 * e + a]; for (int i = 0; i < 3; i++) X[a][i] = c->x[3 * vs[a] + i]; }
      for (int i = 0; i < 3; i++) for (int a = 0; a < 3; a++) Jm[i][a] = X[a + 1][i] - X[0][i];
      const double det = Jm[0][0] * (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) - Jm[0][1] * (Jm[1][0] * Jm[2][2] - Jm[1][2] * Jm[2][0]) +
                         Jm[0][2] * (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]);
      /* rows of J^-1 (cofactors / det) are grad lambda_1..3 */
      g[1][0] = (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) / det; g[1][1] = (Jm[0][2] * Jm[2][1] - Jm[0][1] * Jm[2][2]) / det; g[1][2] = (Jm[0][1] * Jm[1][2] - Jm[0][2] * Jm[1][1]) / det;
      g[2][0] = (Jm[1][2] * Jm[2][0] - Jm[1][0] * Jm[2][2]) / det; g[2][1] = (Jm[0][0] * Jm[2][2] - Jm[0][2] * Jm[2][0]) / det; g[2][2] = (Jm[0][2] * Jm[1][0] - Jm[0][0] * Jm[1][2]) / det;
      g[3][0] = (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]) / det; g[3][1] = (Jm[0][1] * Jm[2][0] - Jm[0][0] * Jm[2][1]) / det; g[3][2] = (Jm[0][0] * Jm[1][1] - Jm[0][1] * Jm[1][0]) / det;
      for (int i = 0; i < 3; i++) g[0][i] = -(g[1][i] + g[2][i] + g[3][i]);
      const double vol = fabs(det) / 6.0;
      for (int a = 0; a < 4; a++) {
        c->Ml[vs[a]] += vol / 4.0;
        for (int b = 0; b < 4; b++)
          c->Lval[c->vptr[vs[a]] + c->cellpos[16 * e + 4 * a + b]] += vol * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
      }
    }
  }
  if (!c->Lval) { /* geometry: P1 stiffness on the vertex graph + lumped mass */
    c->Lval = (double *)calloc(c->vptr[nv], sizeof(double));
    c->Ml = (double *)calloc(nv, sizeof(double));
    for (int e = 0; e < c->nc; e++) {
      double xe[3][2], g[3][2], area, h;
      int vs[3];
      for (int a = 0; a < 3; a++) { vs[a] = c->cells[3 * e + a]; xe[a][0] = c->x[2 * vs[a]]; xe[a][1] = c->x[2 * vs[a] + 1]; }
      geom(xe, g, &area, &h);
      for (int a = 0; a < 3; a++) {
        c->Ml[vs[a]] += area / 3.0;
        for (int b = 0; b < 3; b++) c->Lval[c->vptr[vs[a]] + c->cellpos[9 * e + 3 * a + b]] += area * (g[a][0] * g[b][0] + g[a][1] * g[b][1]);
      }
    }
  }
  if (!c->hA) c->hA = (amg_hier *)calloc(1, sizeof(amg_hier));
  if (!c->hL) c->hL = (amg_hier *)calloc(1, sizeof(amg_hier));
  /* scalar proxy (A00_xx + A00_yy)/2 on the vertex graph, zeros dropped */
  {
    csr_t Ah;
    csr_alloc(&Ah, nv, c->vptr[nv]);
    int n = 0;
    for (int i = 0; i < nv; i++) {
      Ah.rowptr[i] = n;
      int deg = c->vptr[i + 1] - c->vptr[i];
      for (int k = 0; k < deg; k++) {
        int w = c->vadj[c->vptr[i] + k];
        double v = 0.0;  /* mean of the diagonal of the D x D block */
        for (int q = 0; q < D; q++) v += c->val[c->rowptr[D * i + q] + D * k + q];
        v /= D;
        if (v == 0.0 && w != i) continue;
        Ah.col[n] = w; Ah.val[n] = v; n++;
      }
    }
    Ah.rowptr[nv] = n;
    int rc = amg_setup(c, c->hA, &Ah, o, 0);
    free_csr(&Ah);
    if (rc) return rc;
  }
  /* Dirichlet set of the pressure Laplacian: the pressure-Dirichlet dofs, plus -- with a do-nothing
   * boundary (ds_terms off) -- the vertices of every exterior facet that is not a no-slip/inflow facet
   * (outflow boundary of the pressure Poisson problem; Elman-Silvester-Wathen 2014, sec. 9.2).
   * pbc: bit0 = pressure dof is Dirichlet (identity row in H, z_p = r_p), bit1 = Dirichlet in L only. */
  uint8_t *pbc = (uint8_t *)malloc(nv);
  int changed = (c->ccPbc == NULL) || c->hL_singular != c->singular;
  for (int i = 0; i < nv; i++) pbc[i] = c->isbc[nu + i] ? 1 : 0;
  if (!c->ds_terms)
    for (int k = 0; k < c->nf; k++) {
      int e = c->fcell[k], fl = c->flocal[k];
      int fixed = 1, loc[8], nn = 0;
      if (c->etg) nn = D == 2 ? orcg_facet_nodes(c->etg, fl, loc) : orcg3_facet_nodes(c->etg, fl, loc);
      else for (int q = 0; q < NL; q++) if (q != fl) loc[nn++] = q;
      for (int q = 0; q < nn; q++) {
        int v = c->cells[c->NC * e + loc[q]];
        for (int i = 0; i < D; i++) fixed = fixed && c->isbc[D * v + i];
      }
      if (fixed) continue;
      for (int q = 0; q < nn; q++) pbc[c->cells[c->NC * e + loc[q]]] |= 2;
    }
  for (int i = 0; i < nv; i++) if (c->ccPbc && c->ccPbc[i] != pbc[i]) changed = 1;
  if (changed) {
    csr_t Lh;
    csr_alloc(&Lh, nv, c->vptr[nv]);
    int n = 0;
    for (int i = 0; i < nv; i++) {
      Lh.rowptr[i] = n;
      if (pbc[i]) { Lh.col[n] = i; Lh.val[n] = 1.0; n++; continue; }
      for (int k = c->vptr[i]; k < c->vptr[i + 1]; k++) { int w = c->vadj[k]; if (pbc[w]) continue; Lh.col[n] = w; Lh.val[n] = c->Lval[k]; n++; }
    }
    Lh.rowptr[nv] = n;
    int any_l = 0;
    for (int i = 0; i < nv; i++) any_l |= pbc[i];
    int rc = amg_setup(c, c->hL, &Lh, o, c->singular || !any_l);
    free_csr(&Lh);
    if (rc) { free(pbc); return rc; }
    free(c->ccPbc); c->ccPbc = pbc; c->hL_singular = c->singular;
    free(c->ccMl); c->ccMl = (double *)malloc(sizeof(double) * nv);
    for (int i = 0; i < nv; i++) c->ccMl[i] = pbc[i] ? 0.0 : c->Ml[i];
  } else free(pbc);
  c->cc_alpha = c->rho * c->a0 / (c->theta * c->dt); c->cc_beta = c->mu;
  {
    csr_t Hh;
    csr_alloc(&Hh, nv, c->vptr[nv]);
    int n = 0;
    for (int i = 0; i < nv; i++) {
      Hh.rowptr[i] = n;
      if (c->ccPbc[i] & 1) { Hh.col[n] = i; Hh.val[n] = 1.0; n++; continue; }
      int deg = c->vptr[i + 1] - c->vptr[i];
      const double *rp = c->val + c->rowptr[nu + i] + D * deg; /* A11 row */
      int kd = 0;
      while (c->vadj[c->vptr[i] + kd] != i) kd++;
      double Ld = c->Lval[c->vptr[i] + kd];
      double T = Ld > 0 ? rp[kd] / Ld : 0.0;
      for (int k = 0; k < deg; k++) {
        int w = c->vadj[c->vptr[i] + k];
        if (c->ccPbc[w] & 1) continue;
        double v = c->cc_beta * rp[k];
        if (w == i) v += (1.0 + c->cc_alpha * T) * c->Ml[i];
        Hh.col[n] = w; Hh.val[n] = v; n++;
      }
    }
    Hh.rowptr[nv] = n;
    if (c->Hlev) level_free(c->Hlev); else c->Hlev = (amg_level *)calloc(1, sizeof(amg_level));
    level_setup(c->Hlev, Hh, 8.0);
  }
  return 0;
}

/* PC setup for the current Jacobian: ILU(0) of A00 and of Sp = A11 - A10 D^-1 A01 */
static int pc_setup(orc_ctx *c, const orc_opts *o) {
  const int nv = c->nv, nu = 2 * nv;
  if (c->D != 2 && o->pc_kind != 2) { snprintf(c->err, sizeof c->err, "tetrahedra: pc_kind 2 only"); return -1; }
  if (o->pc_kind == 2) {
    if (!c->amg_valid || c->amg_force) {
      if (cc_setup(c, o)) return -1;
      c->amg_valid = 1; c->amg_force = 0; c->amg_its_ref = 0;
    }
    return 0;
  }
  /* A00 copy */
  csr_t *A = &c->A00f;
  if (!A->rowptr) {
    A->n = nu;
    A->rowptr = (int *)malloc(sizeof(int) * (nu + 1));
    A->rowptr[0] = 0;
    for (int r = 0; r < nu; r++) { int v = r / 2; A->rowptr[r + 1] = A->rowptr[r] + 2 * (c->vptr[v + 1] - c->vptr[v]); }
    A->col = (int *)malloc(sizeof(int) * A->rowptr[nu]);
    A->val = (double *)malloc(sizeof(double) * A->rowptr[nu]);
    for (int r = 0; r < nu; r++) memcpy(A->col + A->rowptr[r], c->col + c->rowptr[r], sizeof(int) * (A->rowptr[r + 1] - A->rowptr[r]));
  }
  for (int r = 0; r < nu; r++) memcpy(A->val + A->rowptr[r], c->val + c->rowptr[r], sizeof(double) * (A->rowptr[r + 1] - A->rowptr[r]));
  double *Dinv = (double *)malloc(sizeof(double) * nu);
  for (int r = 0; r < nu; r++) {
    double d = 0;
    for (int k = A->rowptr[r]; k < A->rowptr[r + 1]; k++) if (A->col[k] == r) d = A->val[k];
    Dinv[r] = 1.0 / d;
  }
  /* Sp pattern (once): union over u-columns k of row i of the p-columns of row k */
  csr_t *S = &c->Sp;
  if (!S->rowptr) {
    S->n = nv;
    S->rowptr = (int *)calloc(nv + 1, sizeof(int));
    int *mark = (int *)malloc(sizeof(int) * nv);
    for (int i = 0; i < nv; i++) mark[i] = -1;
    int cap = 32 * nv, n = 0;
    S->col = (int *)malloc(sizeof(int) * cap);
    for (int i = 0; i < nv; i++) {
      int start = n;
      for (int kk = c->vptr[i]; kk < c->vptr[i + 1]; kk++) {
        int w = c->vadj[kk];
        for (int k2 = c->vptr[w]; k2 < c->vptr[w + 1]; k2++) {
          int j = c->vadj[k2];
          if (mark[j] != i) {
            mark[j] = i;
            if (n == cap) { cap *= 2; S->col = (int *)realloc(S->col, sizeof(int) * cap); }
            S->col[n++] = j;
          }
        }
      }
      qsort(S->col + start, n - start, sizeof(int), cmp_int);
      S->rowptr[i + 1] = n;
    }
    free(mark);
    S->val = (double *)malloc(sizeof(double) * n);
  }
  {
    int *pos = (int *)malloc(sizeof(int) * nv);
    for (int i = 0; i < nv; i++) pos[i] = -1;
    for (int i = 0; i < nv; i++) {
      int s = S->rowptr[i], e = S->rowptr[i + 1];
      for (int k = s; k < e; k++) { pos[S->col[k]] = k; S->val[k] = 0.0; }
      int r = nu + i, deg = c->vptr[i + 1] - c->vptr[i];
      int rs = c->rowptr[r];
      for (int k = 0; k < deg; k++) S->val[pos[c->col[rs + 2 * deg + k] - nu]] += c->val[rs + 2 * deg + k];
      for (int k = 0; k < 2 * deg; k++) {
        int ucol = c->col[rs + k];
        double a = c->val[rs + k] * Dinv[ucol];
        if (a == 0.0) continue;
        int w = ucol / 2, dw = c->vptr[w + 1] - c->vptr[w], us = c->rowptr[ucol];
        for (int k2 = 0; k2 < dw; k2++) S->val[pos[c->col[us + 2 * dw + k2] - nu]] -= a * c->val[us + 2 * dw + k2];
      }
      for (int k = s; k < e; k++) pos[S->col[k]] = -1;
    }
    free(pos);
  }
  if (o->pc_kind == 1) {
    /* Jacobi diagonal + spectral bound every Jacobian, hierarchy lagged (as on the GPU) */
    free(c->dinvA); c->dinvA = Dinv;
    csr_t A00v; /* view of the unfactored copy */
    A00v = *A;
    c->lmaxA = 1.15 * power_lmax(&A00v, Dinv, 8);
    if (!c->amg_valid || c->amg_force) {
      if (!c->hS) c->hS = (amg_hier *)calloc(1, sizeof(amg_hier));
      if (amg_setup(c, c->hS, S, o, c->singular)) return -1;
      c->amg_valid = 1; c->amg_force = 0; c->amg_its_ref = 0;
    }
    return 0;
  }
  free(Dinv);
  if (ilu0(A)) { snprintf(c->err, sizeof c->err, "zero pivot in ILU(A00)"); return -1; }
  if (ilu0(S)) { snprintf(c->err, sizeof c->err, "zero pivot in ILU(Sp)"); return -1; }
  return 0;
}

/* inner KSP on A00: GMRES(restart) left-preconditioned with ILU(0), rtol on the
 * preconditioned residual (PETSc defaults for ksp_u, stabilized_schur.py:261) */
typedef struct {
  int n, m;
  double *V, *H, *w, *t, *cs, *sn, *g, *y;
} gm_ws;
static void gm_alloc(gm_ws *ws, int n, int m) {
  ws->n = n; ws->m = m;
  ws->V = (double *)malloc(sizeof(double) * (size_t)n * (m + 1));
  ws->H = (double *)malloc(sizeof(double) * (m + 1) * m);
  ws->w = (double *)malloc(sizeof(double) * n); ws->t = (double *)malloc(sizeof(double) * n);
  ws->cs = (double *)malloc(sizeof(double) * m); ws->sn = (double *)malloc(sizeof(double) * m);
  ws->g = (double *)malloc(sizeof(double) * (m + 1)); ws->y = (double *)malloc(sizeof(double) * m);
}
static void gm_free(gm_ws *ws) { free(ws->V); free(ws->H); free(ws->w); free(ws->t); free(ws->cs); free(ws->sn); free(ws->g); free(ws->y); }

static int a00_solve(orc_ctx *c, gm_ws *ws, const double *b, double *x, double rtol, int max_it, int *its_out) {
  const int n = ws->n, m = ws->m;
  memset(x, 0, sizeof(double) * n);
  int its = 0;
  double r0n = -1;
  for (;;) {
    /* r = M^-1 (b - A x) */
    blk_mult(c, 1, x, ws->t);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) ws->t[i] = b[i] - ws->t[i];
    ilu_solve(&c->A00f, ws->t, ws->w);
    double beta = vnorm(n, ws->w);
    if (r0n < 0) r0n = beta;
    if (beta <= rtol * r0n || beta == 0.0 || its >= max_it) break;
    double *V = ws->V;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) V[i] = ws->w[i] / beta;
    memset(ws->g, 0, sizeof(double) * (m + 1));
    ws->g[0] = beta;
    int j;
    for (j = 0; j < m && its < max_it; j++) {
      blk_mult(c, 1, V + (size_t)j * n, ws->t);
      ilu_solve(&c->A00f, ws->t, ws->w);
      double *Hj = ws->H + (size_t)j * (m + 1);
      for (int i = 0; i <= j; i++) { Hj[i] = vdot(n, ws->w, V + (size_t)i * n); vaxpy(n, -Hj[i], V + (size_t)i * n, ws->w); }
      Hj[j + 1] = vnorm(n, ws->w);
      double *vn = V + (size_t)(j + 1) * n;
      double inv = Hj[j + 1] != 0.0 ? 1.0 / Hj[j + 1] : 0.0;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++) vn[i] = ws->w[i] * inv;
      for (int i = 0; i < j; i++) {
        double t = ws->cs[i] * Hj[i] + ws->sn[i] * Hj[i + 1];
        Hj[i + 1] = -ws->sn[i] * Hj[i] + ws->cs[i] * Hj[i + 1];
        Hj[i] = t;
      }
      double d = hypot(Hj[j], Hj[j + 1]);
      ws->cs[j] = Hj[j] / d; ws->sn[j] = Hj[j + 1] / d;
      Hj[j] = d; Hj[j + 1] = 0;
      ws->g[j + 1] = -ws->sn[j] * ws->g[j]; ws->g[j] = ws->cs[j] * ws->g[j];
      its++;
      if (fabs(ws->g[j + 1]) <= rtol * r0n) { j++; break; }
    }
    for (int i = j - 1; i >= 0; i--) {
      double s = ws->g[i];
      for (int k = i + 1; k < j; k++) s -= ws->H[(size_t)k * (m + 1) + i] * ws->y[k];
      ws->y[i] = s / ws->H[(size_t)i * (m + 1) + i];
    }
    for (int k = 0; k < j; k++) vaxpy(n, ws->y[k], V + (size_t)k * n, x);
  }
  *its_out += its;
  return 0;
}

/* z = P^-1 r: PCFIELDSPLIT Schur FULL (stabilized_schur.py:231-235,256-264) */
typedef struct {
  gm_ws sub;
  double *yu, *yp, *tu, *tp;
  double sub_rtol;
  int sub_max_it, sub_its;
  const orc_opts *o;
  double *cr, *cd0, *cd1;
} pc_ws;

static void pc_apply(orc_ctx *c, pc_ws *p, const double *r, double *z) {
  const int D = c->D;
  const int nv = c->nv, nu = D * nv;
  if (p->o->pc_kind == 2) {
    const orc_opts *o = p->o;
    /* y_u = V(A~) r_u, component by component */
    for (int cpt = 0; cpt < D && !o->schur_upper; cpt++) {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) p->cd0[i] = r[D * i + cpt];
      amg_vcycle(c->hA, o, 0, p->cd0, p->cd1);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) p->yu[D * i + cpt] = p->cd1[i];
    }
    if (o->schur_upper) {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) p->tp[i] = r[nu + i];
    } else {
      blk_mult(c, 3, p->yu, p->tp);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) p->tp[i] = r[nu + i] - p->tp[i];
    }
    /* z_p = (a' L^-1 + b' M_l^-1) H^-1 t_p */
    amg_level *Hl = c->Hlev;
    cheb_smooth(&Hl->A, Hl->dinv, Hl->lmin, Hl->lmax, o->cc_smooth_degree, p->tp, p->yp, 1, Hl->r, Hl->d0, Hl->d1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nv; i++) p->cd0[i] = c->ccMl[i] * p->yp[i];
    amg_vcycle(c->hL, o, 0, p->cd0, p->cd1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nv; i++) z[nu + i] = (c->ccPbc[i] & 1) ? p->tp[i] : c->cc_alpha * p->cd1[i] + c->cc_beta * p->yp[i];
    blk_mult(c, 2, z + nu, p->tu);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nu; i++) p->tu[i] = r[i] - p->tu[i];
    for (int cpt = 0; cpt < D; cpt++) {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) p->cd0[i] = p->tu[D * i + cpt];
      amg_vcycle(c->hA, o, 0, p->cd0, p->cd1);
#pragma omp parallel for schedule(static)
      for (int i = 0; i < nv; i++) z[D * i + cpt] = p->cd1[i];
    }
    return;
  }
  if (p->o->pc_kind == 1) {
    const orc_opts *o = p->o;
    const csr_t *A = &c->A00f;
    const double lmin = c->lmaxA / o->cheb_ratio;
    cheb_smooth(A, c->dinvA, lmin, c->lmaxA, o->cheb_degree, r, p->yu, 1, p->cr, p->cd0, p->cd1);
    blk_mult(c, 3, p->yu, p->tp);
    for (int i = 0; i < nv; i++) p->tp[i] = r[nu + i] - p->tp[i];
    amg_vcycle(c->hS, o, 0, p->tp, z + nu);
    blk_mult(c, 2, z + nu, p->tu);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nu; i++) p->tu[i] = r[i] - p->tu[i];
    cheb_smooth(A, c->dinvA, lmin, c->lmaxA, o->cheb_degree, p->tu, z, 1, p->cr, p->cd0, p->cd1);
    return;
  }
  a00_solve(c, &p->sub, r, p->yu, p->sub_rtol, p->sub_max_it, &p->sub_its); /* y_u = A00^-1 r_u */
  blk_mult(c, 3, p->yu, p->tp);                                            /* A10 y_u */
  for (int i = 0; i < nv; i++) p->tp[i] = r[nu + i] - p->tp[i];
  ilu_solve(&c->Sp, p->tp, z + nu);                                          /* y_p = ILU(Sp)^-1 (...) */
  blk_mult(c, 2, z + nu, p->tu);                                             /* A01 y_p */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < nu; i++) p->tu[i] = r[i] - p->tu[i];
  a00_solve(c, &p->sub, p->tu, z, p->sub_rtol, p->sub_max_it, &p->sub_its);  /* y_u = A00^-1 (r_u - A01 y_p) */
}

static void remove_pmean(const orc_ctx *c, double *v) {
  double s = 0;
  const int nv = c->nv, nu = c->D * nv;
  for (int i = 0; i < nv; i++) s += v[nu + i];
  s /= nv;
  for (int i = 0; i < nv; i++) v[nu + i] -= s;
}

/* outer FGMRES (right preconditioned, true-residual norm test against |b|, x0 = 0) */
static void guess_ensure(orc_ctx *c, const orc_opts *o) {
  const int n = c->ndof, m = o->ksp_guess > 8 ? 8 : o->ksp_guess;
  if (m <= 0 || c->gm == m) return;
  for (int k = 0; k < 4; k++) {
    free(c->gU[k]); free(c->gX[k]);
    c->gU[k] = (double *)calloc((size_t)n * m, sizeof(double)); c->gX[k] = (double *)calloc(n, sizeof(double));
    c->gcnt[k] = c->ghead[k] = c->gstored[k] = 0;
  }
  c->gm = m;
}

/* x0 = U y, y = argmin |b - J U y| over the kept corrections of Newton index `slot` (normal equations, pivoted Cholesky) */
static void guess_project(orc_ctx *c, const orc_opts *o, int slot, const double *b, double *x, int singular) {
  const int n = c->ndof;
  if (o->ksp_guess <= 0 || slot < 0 || slot >= 4 || c->gm <= 0) return;
  const int k = c->gcnt[slot];
  if (k == 0) return;
  double *W = (double *)malloc(sizeof(double) * (size_t)n * k);
  double G[8][8], g[8], L[8][8], dg[8], t[8], y[8];
  int piv[8], taken[8], r = 0;
  for (int i = 0; i < k; i++) blk_mult(c, 0, c->gU[slot] + (size_t)i * n, W + (size_t)i * n);
  for (int i = 0; i < k; i++) {
    g[i] = vdot(n, W + (size_t)i * n, b);
    for (int q = 0; q <= i; q++) G[i][q] = G[q][i] = vdot(n, W + (size_t)i * n, W + (size_t)q * n);
  }
  double dmax = 0.0;
  for (int i = 0; i < k; i++) { dg[i] = G[i][i]; taken[i] = 0; y[i] = 0.0; if (dg[i] > dmax) dmax = dg[i]; for (int q = 0; q < k; q++) L[i][q] = 0.0; }
  if (dmax > 0.0)
    for (int rr = 0; rr < k; rr++) {
      int p = -1;
      for (int i = 0; i < k; i++) if (!taken[i] && (p < 0 || dg[i] > dg[p])) p = i;
      if (p < 0 || !(dg[p] > 1e-10 * G[p][p]) || !(dg[p] > 1e-14 * dmax)) break;
      taken[p] = 1; piv[r] = p;
      const double lpp = sqrt(dg[p]);
      L[p][r] = lpp;
      for (int i = 0; i < k; i++) {
        if (taken[i]) continue;
        double sacc = G[i][p];
        for (int q = 0; q < r; q++) sacc -= L[i][q] * L[p][q];
        L[i][r] = sacc / lpp;
        dg[i] -= L[i][r] * L[i][r];
      }
      r++;
    }
  for (int a = 0; a < r; a++) { double sacc = g[piv[a]]; for (int q = 0; q < a; q++) sacc -= L[piv[a]][q] * t[q]; t[a] = sacc / L[piv[a]][a]; }
  for (int a = r - 1; a >= 0; a--) { double sacc = t[a]; for (int q = a + 1; q < r; q++) sacc -= L[piv[q]][a] * y[piv[q]]; y[piv[a]] = sacc / L[piv[a]][a]; }
  for (int i = 0; i < k; i++) if (isfinite(y[i]) && y[i] != 0.0) vaxpy(n, y[i], c->gU[slot] + (size_t)i * n, x);
  if (singular) remove_pmean(c, x);
  free(W);
}

static int fgmres(orc_ctx *c, const orc_opts *o, pc_ws *pc, const double *b, double *x, int singular, int *its_out, int slot) {
  const int n = c->ndof, m = o->ksp_restart;
  double *V = (double *)malloc(sizeof(double) * (size_t)n * (m + 1));
  double *Z = (double *)malloc(sizeof(double) * (size_t)n * m);
  double *H = (double *)malloc(sizeof(double) * (m + 1) * m);
  double *w = (double *)malloc(sizeof(double) * n);
  double *cs = (double *)malloc(sizeof(double) * m), *sn = (double *)malloc(sizeof(double) * m);
  double *g = (double *)malloc(sizeof(double) * (m + 1)), *y = (double *)malloc(sizeof(double) * m);
  memset(x, 0, sizeof(double) * n);
  double bn = vnorm(n, b);
  int its = 0, reason = 0;
  if (bn == 0.0) { reason = 1; goto done; }
  guess_project(c, o, slot, b, x, singular);  /* no-op unless orc_opts.ksp_guess > 0 */
  for (;;) {
    blk_mult(c, 0, x, w);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) w[i] = b[i] - w[i];
    double beta = vnorm(n, w);
    if (beta <= fmax(o->ksp_rtol * bn, o->ksp_atol)) { reason = 2; break; }
    if (its >= o->ksp_max_it) { reason = -3; break; }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; i++) V[i] = w[i] / beta;
    memset(g, 0, sizeof(double) * (m + 1));
    g[0] = beta;
    int j;
    for (j = 0; j < m && its < o->ksp_max_it; j++) {
      double *zj = Z + (size_t)j * n;
      pc_apply(c, pc, V + (size_t)j * n, zj);
      if (singular) remove_pmean(c, zj);
      blk_mult(c, 0, zj, w);
      double *Hj = H + (size_t)j * (m + 1);
      for (int i = 0; i <= j; i++) { Hj[i] = vdot(n, w, V + (size_t)i * n); vaxpy(n, -Hj[i], V + (size_t)i * n, w); }
      Hj[j + 1] = vnorm(n, w);
      double inv = Hj[j + 1] != 0.0 ? 1.0 / Hj[j + 1] : 0.0;
      double *vn = V + (size_t)(j + 1) * n;
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++) vn[i] = w[i] * inv;
      for (int i = 0; i < j; i++) {
        double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1];
        Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
        Hj[i] = t;
      }
      double d = hypot(Hj[j], Hj[j + 1]);
      cs[j] = Hj[j] / d; sn[j] = Hj[j + 1] / d;
      Hj[j] = d; Hj[j + 1] = 0;
      g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
      its++;
      if (o->verbose > 1) printf("      oracle fgmres %3d  |r|/|b| = %.3e\n", its, fabs(g[j + 1]) / bn);
      if (fabs(g[j + 1]) <= fmax(o->ksp_rtol * bn, o->ksp_atol)) { j++; break; }
    }
    for (int i = j - 1; i >= 0; i--) {
      double s = g[i];
      for (int k = i + 1; k < j; k++) s -= H[(size_t)k * (m + 1) + i] * y[k];
      y[i] = s / H[(size_t)i * (m + 1) + i];
    }
    for (int k = 0; k < j; k++) vaxpy(n, y[k], Z + (size_t)k * n, x);
  }
done:
  free(V); free(Z); free(H); free(w); free(cs); free(sn); free(g); free(y);
  *its_out += its;
  if (reason > 0 && o->ksp_guess > 0 && slot >= 0 && slot < 4 && c->gm > 0) {
    vcopy(n, x, c->gU[slot] + (size_t)c->ghead[slot] * n);
    c->gstored[slot] = 1;
    c->ghead[slot] = (c->ghead[slot] + 1) % c->gm;
    if (c->gcnt[slot] < c->gm) c->gcnt[slot]++;
  }
  return reason;
}

void orc_default_opts(orc_opts *o) {
  /* PETSc defaults + the caps of stabilized_schur.py:269-274 */
  o->snes_rtol = 1e-8; o->snes_atol = 1e-50; o->snes_stol = 1e-8; o->snes_max_it = 100;
  o->ksp_rtol = 1e-5; o->ksp_atol = 1e-50; o->ksp_max_it = 1000; o->ksp_restart = 200;
  o->sub_rtol = 1e-5; o->sub_max_it = 10000; o->sub_restart = 30;
  o->remove_p_mean = 1; o->verbose = 0;
  o->pc_kind = 0; o->cheb_degree = 3; o->cheb_ratio = 10.0; o->amg_smooth_degree = 1; o->amg_smooth_ratio = 8.0;
  o->amg_theta = 0.08; o->amg_max_coarse = 1000; o->cc_smooth_degree = 2; o->schur_upper = 0;
  o->ksp_guess = 0;
}

/* One time step: Newton on the monolithic vector xv (in: initial guess = previous
 * converged vector, out: solution).  stabilized_schur.py:313-334. */
int orc_solve_step(orc_ctx *c, double *xv, const orc_opts *o, orc_stats *st) {
  const int n = c->ndof, nv = c->nv, nu = c->D * nv;
  memset(st, 0, sizeof *st);
  double *F = (double *)malloc(sizeof(double) * n), *d = (double *)malloc(sizeof(double) * n);
  double *xt = (double *)malloc(sizeof(double) * n), *Ft = (double *)malloc(sizeof(double) * n);
  pc_ws pc;
  memset(&pc, 0, sizeof pc);
  gm_alloc(&pc.sub, nu, o->sub_restart);
  pc.yu = (double *)malloc(sizeof(double) * nu); pc.tu = (double *)malloc(sizeof(double) * nu);
  pc.yp = (double *)malloc(sizeof(double) * nv); pc.tp = (double *)malloc(sizeof(double) * nv);
  pc.sub_rtol = o->sub_rtol; pc.sub_max_it = o->sub_max_it;
  pc.o = o;
  pc.cr = (double *)malloc(sizeof(double) * nu); pc.cd0 = (double *)malloc(sizeof(double) * nu); pc.cd1 = (double *)malloc(sizeof(double) * nu);
  int singular = 0; /* decided below by MatNullSpaceTest on the assembled Jacobian (stabilized_schur.py:314) */
  if (o->remove_p_mean) remove_pmean(c, xv);
  double t0 = now_ms();
  orc_assemble(c, xv, 1, F);
  st->ms_assemble += now_ms() - t0;
  double fn = vnorm(n, F);
  st->fnorm0 = fn;
  {
    /* nullsp.test(A): |J n| < 1e-7 for the normalised constant-pressure vector n.  An open outlet
     * (free velocity + the ds terms of :79) makes the system regular even without a pressure condition. */
    /* One criterion in the product (csrc/cfdh_solver.cpp), here and in np_twin_nd.py: PETSc's absolute bound AND
     * |J n| <= 1e-6 | |J| n | -- the absolute bound alone is passed by ANY matrix on a mesh in metres with millimetre
     * cells (a deviation from MatNullSpaceTest where the two disagree). */
    double *nvec = (double *)calloc(n, sizeof(double)), *yv = (double *)malloc(sizeof(double) * n);
    for (int i = 0; i < nv; i++) nvec[nu + i] = 1.0 / sqrt((double)nv);
    blk_mult(c, 0, nvec, yv);
    double an2 = 0.0;
    for (int r = 0; r < n; r++) {
      double a = 0.0;
      for (int k = c->rowptr[r]; k < c->rowptr[r + 1]; k++) a += fabs(c->val[k]) * nvec[c->col[k]];
      an2 += a * a;
    }
    const double jn = vnorm(n, yv);
    singular = jn < 1e-7 && jn <= 1e-6 * sqrt(an2);
    free(nvec); free(yv);
    if (singular != c->singular) { c->singular = singular; c->amg_valid = 0; }
  }
  int reason = 0;
  guess_ensure(c, o);
  for (int it = 0;; it++) {
    if (o->verbose) printf("  oracle newton %d |F| = %.6e\n", it, fn);
    if (fn < o->snes_atol) { reason = 2; break; }
    if (it > 0 && fn <= o->snes_rtol * st->fnorm0) { reason = 3; break; }
    if (it >= o->snes_max_it) { reason = -5; break; }
    t0 = now_ms();
    if (pc_setup(c, o)) { reason = -3; break; }
    int its_before = st->krylov_its;
    if (o->ksp_guess > 0 && it < 4) { c->gstored[it] = 0; vcopy(n, xv, c->gX[it]); }
    int kr = fgmres(c, o, &pc, F, d, singular, &st->krylov_its, it < 4 ? it : -1);
    if (o->pc_kind >= 1) { /* adaptive lagging of the hierarchy, as in libcfdh */
      int kits = st->krylov_its - its_before;
      if (c->amg_its_ref == 0) c->amg_its_ref = kits > 0 ? kits : 1;
      else if (kits > (3 * c->amg_its_ref) / 2 + 5) c->amg_force = 1;
    }
    st->ms_solve += now_ms() - t0;
    if (kr < 0) { reason = -3; snprintf(c->err, sizeof c->err, "linear solve failed (%d)", kr); break; }
    /* bt line search on 1/2|F|^2 (Dennis-Schnabel backtracking, alpha = 1e-4) */
    double lam = 1.0, fnew = 0;
    t0 = now_ms();
    int ok = 0;
    for (int ls = 0; ls < 40; ls++) {
#pragma omp parallel for schedule(static)
      for (int i = 0; i < n; i++) xt[i] = xv[i] - lam * d[i];
      orc_assemble(c, xt, 0, Ft);
      fnew = vnorm(n, Ft);
      if (fnew * fnew <= fn * fn * (1.0 - 2.0 * 1e-4 * lam) || fnew < o->snes_atol) { ok = 1; break; }
      /* quadratic model through phi(0)=fn^2/2, phi'(0)=-fn^2, phi(lam) */
      double l2 = fn * fn * lam * lam / (2.0 * (0.5 * fnew * fnew - 0.5 * fn * fn + fn * fn * lam));
      if (!(l2 > 0.1 * lam)) l2 = 0.1 * lam;
      if (l2 > 0.5 * lam) l2 = 0.5 * lam;
      lam = l2;
    }
    if (!ok) { reason = -6; st->ms_assemble += now_ms() - t0; break; }
    double dxn = lam * vnorm(n, d), xn = vnorm(n, xt);
    vcopy(n, xt, xv);
    st->newton_its = it + 1;
    /* Jacobian + residual at the new iterate */
    orc_assemble(c, xv, 1, F);
    st->ms_assemble += now_ms() - t0;
    fn = vnorm(n, F);
    if (dxn < o->snes_stol * xn && fn > o->snes_rtol * st->fnorm0 && fn >= o->snes_atol) {
      if (o->verbose) printf("  oracle newton %d |F| = %.6e (stol)\n", it + 1, fn);
      reason = 4; break;
    }
  }
  if (reason > 0 && o->ksp_guess > 0 && c->gm > 0)
    for (int k = 0; k < st->newton_its && k < 4; k++) {  /* kept corrections become x_k - x_final (csrc/cfdh_solver.cpp::guess_refine) */
      if (!c->gstored[k]) continue;
      double *u = c->gU[k] + (size_t)((c->ghead[k] + c->gm - 1) % c->gm) * n;
      for (int i = 0; i < n; i++) u[i] = c->gX[k][i] - xv[i];
    }
  st->fnorm = fn;
  st->reason = reason;
  st->sub_its = pc.sub_its;
  gm_free(&pc.sub);
  free(pc.yu); free(pc.tu); free(pc.yp); free(pc.tp); free(pc.cr); free(pc.cd0); free(pc.cd1);
  free(F); free(d); free(xt); free(Ft);
  return reason;
}

const char *orc_last_error(orc_ctx *c) { return c->err; }

/* y = J x with the currently assembled values (tests) */
void orc_spmv(orc_ctx *c, const double *x, double *y) { blk_mult(c, 0, x, y); }

/* kind 0: F_D, 1: F_L over facets of `marker` list (dfg_1.py:183-202; caller scales by 500)
 *      2: ||u||_L2, 3: ||p||_L2 (scenario.py:315-324) */
double orc_functional(orc_ctx *c, const double *xv, int kind, int nfac, const int *facets, double mu) {
  const int nu = 2 * c->nv;
  if (kind >= 2) {
    double s = 0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int e = 0; e < c->nc; e++) {
      double xe[3][2], g[3][2], area, h;
      int vs[3];
      for (int a = 0; a < 3; a++) { vs[a] = c->cells[3 * e + a]; xe[a][0] = c->x[2 * vs[a]]; xe[a][1] = c->x[2 * vs[a] + 1]; }
      geom(xe, g, &area, &h);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
          double m = area * (a == b ? 2.0 : 1.0) / 12.0;
          if (kind == 2) s += m * (xv[2 * vs[a]] * xv[2 * vs[b]] + xv[2 * vs[a] + 1] * xv[2 * vs[b] + 1]);
          else s += m * xv[nu + vs[a]] * xv[nu + vs[b]];
        }
    }
    return sqrt(s);
  }
  double FD = 0, FL = 0;
  for (int k = 0; k < nfac; k++) {
    int e = c->fcell[facets[k]], fl = c->flocal[facets[k]];
    double xe[3][2], g[3][2], area, h;
    int vs[3];
    for (int a = 0; a < 3; a++) { vs[a] = c->cells[3 * e + a]; xe[a][0] = c->x[2 * vs[a]]; xe[a][1] = c->x[2 * vs[a] + 1]; }
    geom(xe, g, &area, &h);
    double gl = hypot(g[fl][0], g[fl][1]);
    double n[2] = {g[fl][0] / gl, g[fl][1] / gl}; /* n = -FacetNormal */
    double elen = 2.0 * area * gl;
    double t[2] = {n[1], -n[0]};
    double gut[2] = {0, 0};
    for (int a = 0; a < 3; a++) {
      double ut = xv[2 * vs[a]] * t[0] + xv[2 * vs[a] + 1] * t[1];
      gut[0] += ut * g[a][0]; gut[1] += ut * g[a][1];
    }
    double dn = gut[0] * n[0] + gut[1] * n[1];
    double pm = 0.5 * (xv[nu + vs[(fl + 1) % 3]] + xv[nu + vs[(fl + 2) % 3]]);
    FD += elen * (mu * dn * n[1] - pm * n[0]);
    FL -= elen * (mu * dn * n[0] + pm * n[1]);
  }
  return kind == 0 ? FD : FL;
}

/* Wall shear stress, the per-step assemble_wss() of /root/reference/src/solverBase.py:163-195:
 *   Lt = (1/FacetArea) inner(w, Tt) ds,  T = -sigma(u,p) n,  sigma = 2 mu eps(u) - p I (solverBase.py:176-182),
 *   Tt = T - (T.n) n  (the pressure part is purely normal and drops out).
 * P1 test functions on a straight facet e: (1/|e|) oint lambda_a ds = 1/2, so every exterior facet adds Tt/2 to
 * both of its vertices.  out[2*nv], zero away from the boundary. */
void orc_wss(orc_ctx *c, const double *xv, double mu, double *out) {
  memset(out, 0, sizeof(double) * 2 * c->nv);
  for (int k = 0; k < c->nf; k++) {
    const int e = c->fcell[k], fl = c->flocal[k];
    double xe[3][2], g[3][2], area, h;
    int vs[3];
    for (int a = 0; a < 3; a++) { vs[a] = c->cells[3 * e + a]; xe[a][0] = c->x[2 * vs[a]]; xe[a][1] = c->x[2 * vs[a] + 1]; }
    geom(xe, g, &area, &h);
    const double gl = hypot(g[fl][0], g[fl][1]);
    const double n[2] = {-g[fl][0] / gl, -g[fl][1] / gl}; /* outward normal: grad(lambda_fl) points inwards */
    double G[2][2] = {{0, 0}, {0, 0}};                    /* G_ij = d_i u_j */
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) G[i][j] += g[a][i] * xv[2 * vs[a] + j];
    const double E01 = 0.5 * (G[0][1] + G[1][0]);
    const double T[2] = {-2.0 * mu * (G[0][0] * n[0] + E01 * n[1]), -2.0 * mu * (E01 * n[0] + G[1][1] * n[1])};
    const double Tn = T[0] * n[0] + T[1] * n[1];
    for (int q = 1; q <= 2; q++) {
      const int v = vs[(fl + q) % 3];
      out[2 * v] += 0.5 * (T[0] - Tn * n[0]);
      out[2 * v + 1] += 0.5 * (T[1] - Tn * n[1]);
    }
  }
}
