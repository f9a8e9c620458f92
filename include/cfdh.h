/*
 * cfdh.h -- C ABI of libcfdh.so: the MI355X (gfx950) implementation of the
 * per-time-step hot path of the reference's `stabilized_schur` solver.
 *
 * The reference has no C-level interface for this path: its boundary is the
 * duck-typed Python plugin surface
 *     Solver(mesh, dt, rho, mu, f, initial_velocity=None, **kw)
 *     Solver.setup(bcu, bcp, facet_tags=None, tags=None)
 *     Solver.solveStep()
 * (/root/reference/src/solvers/stabilized_schur.py:40-52,177-183,313 and
 * /root/reference/src/solverBase.py:25-40,96-102), behind which DOLFINx/PETSc
 * do the arithmetic.  Each entry point below cites the reference lines whose
 * work it performs; INTEGRATION.md shows the ctypes stub that binds them from
 * the reference's own `Solver` class.
 *
 * Conventions: every function returns 0 on success or a negative CFDH_E_*
 * code and never throws; cfdh_last_error() gives the message.  All pointer
 * arguments are caller-owned, C-contiguous host buffers that are copied at the
 * call; outputs are written into caller-allocated buffers.  One context per
 * GPU / per process rank; a context is not thread-safe.  Doubles are IEEE
 * binary64, indices int32 (sizes int64), as in the reference
 * (PETSc.ScalarType = float64, /root/reference/src/solverBase.py:37).
 *
 * Local numbering (multi-GPU): a context holds the vertices [0,nv) of its
 * part, the first nv_owned of them owned, the rest ghosts (one-cell overlap;
 * SURVEY.md 8e).  Velocity arrays are vertex-major/component-minor
 * (u[gdim*v+i]), as DOLFINx lays out the blocked P1 space
 * (stabilized_schur.py:55-57).
 *
 * Dimension: gdim = 2 (triangles) or 3 (tetrahedra,
 * /root/reference/src/scenarios/simple_bifurcation.py, scenario_factory.py:47-49)
 * is fixed at cfdh_create; "d" below stands for it.  Tetrahedral contexts are
 * single-GPU in this version (cfdh_set_halo returns CFDH_E_ARG) and use pc_type 1.
 */
#ifndef CFDH_H
#define CFDH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CFDH_ABI_VERSION 1

enum {
  CFDH_OK = 0,
  CFDH_E_ARG = -1,      /* bad argument (ValueError on the Python side) */
  CFDH_E_HIP = -2,      /* HIP runtime / device error */
  CFDH_E_STATE = -3,    /* call out of order (e.g. solve before set_params) */
  CFDH_E_DIVERGED = -4, /* Newton / Krylov did not converge (RuntimeError) */
  CFDH_E_COMM = -5,     /* RCCL / halo exchange failure */
  CFDH_E_NOMEM = -6
};

/* converged reasons, numbered after PETSc's SNESConvergedReason
 * (stabilized_schur.py:332-334 raises on reason < 0) */
enum {
  CFDH_CONVERGED_FNORM_ABS = 2,
  CFDH_CONVERGED_FNORM_RELATIVE = 3,
  CFDH_CONVERGED_SNORM_RELATIVE = 4,
  CFDH_DIVERGED_LINEAR_SOLVE = -3,
  CFDH_DIVERGED_MAX_IT = -5,
  CFDH_DIVERGED_LINE_SEARCH = -6,
  CFDH_DIVERGED_FNORM_NAN = -4
};
/* reasons of the linear solve (numbered after PETSc's KSPConvergedReason: 2 = CONVERGED_RTOL/ATOL on the true residual,
 * -3 = DIVERGED_ITS, -9 = DIVERGED_NANORINF).  One code has no PETSc counterpart: the solve was stopped ABOVE its tolerance
 * because two cycles in a row converged by the recurrence without moving the true residual (attainable accuracy of the
 * system; only within 10x the tolerance or below 1e-6 |b|).  Such stops are counted: cfdh_info(ctx, 72). */
enum {
  CFDH_KSP_CONVERGED_RTOL = 2,
  CFDH_KSP_CONVERGED_ATTAINABLE = 12,
  CFDH_KSP_DIVERGED_ITS = -3,
  CFDH_KSP_DIVERGED_NANORINF = -9
};

typedef struct cfdh_ctx cfdh_ctx;

typedef struct cfdh_options {
  /* Newton: PETSc SNES defaults, cap of stabilized_schur.py:270 */
  double snes_rtol, snes_atol, snes_stol;
  int32_t snes_max_it;
  /* outer FGMRES: PETSc KSP defaults, caps of stabilized_schur.py:272-273 */
  double ksp_rtol, ksp_atol;
  int32_t ksp_max_it, ksp_restart;
  /* GPU preconditioner (replaces PCFIELDSPLIT Schur FULL/SELFP + ILU(0),
   * stabilized_schur.py:231-264) */
  int32_t cheb_degree;      /* Jacobi-Chebyshev sweeps per A00 solve */
  double cheb_ratio;        /* lambda_max / lambda_min targeted */
  int32_t schur_full;       /* 2 (default): block UPPER-triangular factor (one A00 solve), 1: FULL
                             * factorisation as PC_FIELDSPLIT_SCHUR_FACT_FULL, stabilized_schur.py:225 (two A00
                             * solves), 0: LOWER.  All are right preconditioners of the same FGMRES. */
  int32_t amg_smooth_degree;
  double amg_smooth_ratio;
  double amg_theta;         /* strength threshold of the aggregation; < 0 (default): 0.07 for gdim 2, 0.02 for gdim 3 */
  int32_t amg_max_coarse;
  int32_t pc_refresh;       /* 0: adaptive lagging of the Sp hierarchy, n>0: every n steps, -1: every Jacobian */
  int32_t remove_p_mean;    /* nullsp.remove(x_n), stabilized_schur.py:319 */
  int32_t verbose;
  int32_t pc_type;          /* 0: SELFP Schur matrix + Chebyshev(A00) (the reference's SELFP, :235);
                             * 1: Cahouet-Chabard Schur approximation + AMG(A00) (mesh-independent) */
  int32_t cc_smooth_degree; /* pc_type 1: Chebyshev steps on the mass-like operator H (default 2) */
  int32_t ksp_guess;        /* initial guess of the linear solves (PETSc: KSPGuess, -ksp_guess_type fischer): for each Newton index the
                             * corrections of the last ksp_guess time steps are kept (at the end of a step: iterate minus converged
                             * solution) and the new solve starts from their best combination, x0 = U y with
                             * y = argmin |b - J U y| on the CURRENT Jacobian (so |r0| <= |b|); convergence is still tested against
                             * rtol |b|.  0: zero initial guess (what the reference's KSP does); default 4.  Converged results do
                             * not depend on it, iteration counts do. */
} cfdh_options;

typedef struct cfdh_stats {
  int32_t newton_its, krylov_its, reason, pc_refreshes;
  double fnorm0, fnorm;
  double ms_assemble, ms_solve, ms_pc_setup, ms_total;
} cfdh_stats;

/* ---- life cycle --------------------------------------------------------- */

/* Upload a (part of a) P1 triangle or tetrahedron mesh and build the fixed CSR pattern.
 * Replaces: functionspace/Function/create_matrix_block/create_vector_block
 * (solverBase.py:104-142, stabilized_schur.py:55-57,191-193) and the DG0 cell
 * size h = mesh.h (stabilized_schur.py:82-88).
 * gdim 2 or 3.  cells [nc][gdim+1] (any orientation); coords [nv][gdim];
 * exterior facets: owning cell, local facet index (= local index of the
 * opposite vertex), marker (0 = untagged; facet_marker may be NULL, see
 * cfdh_set_facet_markers). */
int cfdh_create(cfdh_ctx **out, int device, int gdim, int64_t nv, int64_t nv_owned, int64_t nc,
                const int32_t *cells, const double *coords, int64_t nfacets, const int32_t *facet_cells,
                const int32_t *facet_local, const int32_t *facet_marker);
/* The same for nodal equal-order elements beyond P1 simplices: `p_grade = 2` of
 * stabilized_schur_backflow.py:84-87 (P2/P2 triangles on a straight-sided triangulation) and quadrilateral cells as
 * unit_square_pipe.py:101-105 builds them (Q1/Q1; parallelograms only: the geometry map must be affine).
 * cells [nc][nloc] list NODE ids in the DOLFINx/Basix local order -- P2 triangle: vertices 0,1,2 then the edge nodes opposite
 * to them (nloc 6); Q1 quadrilateral: (0,0),(1,0),(0,1),(1,1) (nloc 4); node_coords [nn][2].  Local facets: triangle f =
 * opposite vertex f; quadrilateral 0:(0,1) 1:(0,2) 2:(1,3) 3:(2,3).  Every node carries (u_x, u_y, p): all other calls take
 * node ids / per-node arrays where they say vertex.  gdim 3 (round 4): CFDH_ELEM_Q1 = hexahedral cells as unit_cube_pipe.py:103-109
 * builds them (parallelepipeds; nloc 8, vertex v = i + 2 j + 4 k, local facets 0:(0,1,2,3) 1:(0,1,4,5) 2:(0,2,4,6) 3:(1,3,5,7)
 * 4:(2,3,6,7) 5:(4,5,6,7)), CFDH_ELEM_P2 = P2/P2 tetrahedra (nloc 10: vertices, then the edge nodes in Basix edge order (2,3) (1,3) (1,2)
 * (0,3) (0,2) (0,1); facet f opposite vertex f); node_coords [nn][3].  This entry point creates a whole mesh on one GPU (parts of a
 * partitioned run: cfdh_create_elem_part).  CFDH_ELEM_P1_GENERIC runs P1 triangles / tetrahedra through the
 * quadrature kernels of the P2/Q1 path (a cross-check of the closed-form P1 kernels). */
enum { CFDH_ELEM_P1 = 0, CFDH_ELEM_P2_TRIANGLE = 1, CFDH_ELEM_Q1_QUADRILATERAL = 2, CFDH_ELEM_P1_GENERIC = 3 };
int cfdh_create_elem(cfdh_ctx **out, int device, int gdim, int elem, int64_t nn, int64_t nc, const int32_t *cells,
                     const double *node_coords, int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local,
                     const int32_t *facet_marker);
/* The same for one part of a partitioned run (SURVEY.md 8e for the 8f-4 elements: the reference runs every solver under
 * mpirun, /root/reference/src/simulation_hpc.sh:14-19): nodes [0, nn_owned) are owned, the rest are the ghost nodes of the
 * one-cell overlap, numbered contiguously per neighbour as cfdh_set_halo expects; cells = all cells touching an owned node. */
int cfdh_create_elem_part(cfdh_ctx **out, int device, int gdim, int elem, int64_t nn, int64_t nn_owned, int64_t nc, const int32_t *cells,
                          const double *node_coords, int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local,
                          const int32_t *facet_marker);
/* (Re)assign the markers of the exterior facets after cfdh_create: the reference hands `facet_tags` / `tags` to
 * Solver.setup(), not to the constructor (/root/reference/src/scenario.py:137-149;
 * stabilized_schur_backflow.py:158-163 builds ds_out from them there).  markers[nfacets] in the facet order of
 * cfdh_create (nfacets must equal its nfacets; a rank keeps the entries of the facets whose cell it holds).
 * Drag/lift by marker (cfdh_functional) and the backflow marker of cfdh_set_boundary_terms follow the new values;
 * an active backflow term invalidates Jacobian and preconditioner. */
int cfdh_set_facet_markers(cfdh_ctx *ctx, int64_t nfacets, const int32_t *markers);
void cfdh_destroy(cfdh_ctx *ctx);
const char *cfdh_last_error(const cfdh_ctx *ctx); /* ctx may be NULL after a failed create */
int cfdh_abi_version(void);

/* dt, rho, mu Constants and body force (solverBase.py:36-40; f[2] is read for gdim 3 only); mu_facet is the
 * raw python float used in the ds term (stabilized_schur.py:79). */
int cfdh_set_params(cfdh_ctx *ctx, double dt, double rho, double mu, double mu_facet, const double f[3]);
int cfdh_default_options(cfdh_options *opt);
int cfdh_set_options(cfdh_ctx *ctx, const cfdh_options *opt);

/* ---- Dirichlet data ------------------------------------------------------ */

/* Drop all DirichletBC objects (stabilized_schur.py:198-199 rebuilds them in setup). */
int cfdh_clear_dirichlet(cfdh_ctx *ctx);
/* Append one DirichletBC object (boundaryCondition.py:41-52): field 0 = velocity
 * (values [n][gdim]), 1 = pressure (values [n]).  A later object overrides the
 * value of a shared dof; the matrix diagonal counts the objects holding it.
 * Re-callable every step (bc.update(), stabilized_schur.py:170). */
int cfdh_add_dirichlet(cfdh_ctx *ctx, int field, int64_t n, const int32_t *nodes, const double *values);
/* New VALUES for dofs that are already constrained (`bc.update()` of a time-dependent condition, stabilized_schur.py:170, when
 * the dof sets of all objects are unchanged): no object is added, the matrix diagonal keeps its counts, only the listed
 * vertices are re-sent to the device.  The caller passes the dofs whose value this object determines (a later object that
 * holds the same dof keeps its value).  CFDH_E_ARG if a listed dof is not constrained. */
int cfdh_update_dirichlet(cfdh_ctx *ctx, int field, int64_t n, const int32_t *nodes, const double *values);

/* ---- state ---------------------------------------------------------------- */

/* u_prev/p_prev (solverBase.py:121,139) and the Newton iterate x_n = (u,p)
 * (stabilized_schur.py:216-223).  NULL keeps the device copy.  Arrays cover
 * all nv local vertices (ghost entries are overwritten by the halo exchange). */
int cfdh_set_state(cfdh_ctx *ctx, const double *u_prev, const double *p_prev, const double *u, const double *p);
/* u_sol/p_sol after the step = the Newton iterate (updateSolution, :125-142) */
int cfdh_get_solution(cfdh_ctx *ctx, double *u, double *p);
/* u_prev/p_prev as held on the device (after cfdh_advance they equal the last solution) */
int cfdh_get_previous(cfdh_ctx *ctx, double *u_prev, double *p_prev);
/* u_residual/p_residual (_updateResidual, :295-311) */
int cfdh_get_residual(cfdh_ctx *ctx, double *ru, double *rp);
/* device-side u_prev <- u_sol, p_prev <- p_sol (scenario.py:306-307) */
int cfdh_advance(cfdh_ctx *ctx);
/* one line of that copy: field 0: u_prev <- u_sol (scenario.py:306), field 1: p_prev <- p_sol (:307); lets a binding
 * map the reference's literal `u_prev.x.array[:] = u_sol.x.array[:]` onto the device without a host round trip */
int cfdh_advance_field(cfdh_ctx *ctx, int field);

/* ---- time scheme (the `stabilized_schur_bdf2` variant) ------------------------ */

/* Spatial terms are evaluated at theta*u + (1-theta)*u_prev and the time term is
 * (a0*u + a1*u_prev + a2*u_prev2)/dt.  Default (1/2; 1,-1,0) is the midpoint form of
 * stabilized_schur.py:72-80.  stabilized_schur_bdf2.py:79-80 (u_mid = u_sol) and :95-110,
 * :298-305 (a0,a1,a2 Constants switched by step_count) map to (1; 1,-1,0) on the first
 * step and (1; 1.5,-2,0.5) afterwards.  theta in (0,1], a0 > 0. */
int cfdh_set_time_scheme(cfdh_ctx *ctx, double theta, double a0, double a1, double a2);
/* ---- boundary terms (the `stabilized_schur_backflow` variant) --------------------- */

/* ds_terms != 0 (default): the pair `dot(p n, v) ds - dot(mu grad(u_mid) n, v) ds` of
 * stabilized_schur.py:79 on ALL exterior facets.  stabilized_schur_backflow.py:107 drops it
 * (do-nothing outlet): ds_terms = 0.  beta > 0 adds the backflow stabilisation
 * -beta rho (u_prev.n)_- (u_mid . v) ds, (s)_- = (s-|s|)/2, on the exterior facets whose marker
 * (facet_marker of cfdh_create) equals backflow_marker (`ds_out`, tags["outlet"],
 * stabilized_schur_backflow.py:158-176), integrated with a rule of FFCx's estimated degree 3:
 * 2-point Gauss on edges, the 6-point Strang-Fix rule on the triangles of a tetrahedral mesh. */
int cfdh_set_boundary_terms(cfdh_ctx *ctx, int ds_terms, int backflow_marker, double beta);

/* u_prev2 (stabilized_schur_bdf2.py:72): upload / download; nv local vertices x gdim */
int cfdh_set_previous2(cfdh_ctx *ctx, const double *u_prev2);
int cfdh_get_previous2(cfdh_ctx *ctx, double *u_prev2);
/* device-side u_prev2 <- u_prev (stabilized_schur_bdf2.py:324, end of solveStep) */
int cfdh_shift_history(cfdh_ctx *ctx);

/* ---- assembly (exposed for parity tests) ---------------------------------- */

/* assembleResidual (+ assembleJacobian when want_jacobian) at the current
 * iterate: stabilized_schur.py:144-175. */
int cfdh_assemble(cfdh_ctx *ctx, int want_jacobian);
/* Monolithic scalar CSR of the owned rows in the reference's block ordering
 * ([all u dofs | all p dofs], stabilized_schur.py:194-196,237-252), local
 * column numbering.  Call with rowptr=col=vals=NULL to query nnz. */
int cfdh_get_csr(cfdh_ctx *ctx, int64_t *nnz, int32_t *rowptr, int32_t *col, double *vals);
/* y = J x on the device with the assembled Jacobian; x: [gdim*nv | nv] monolithic
 * local vector, y: owned rows [gdim*nv_owned | nv_owned]. */
int cfdh_spmv(cfdh_ctx *ctx, const double *x, double *y);

/* ---- the step ------------------------------------------------------------- */

/* solveStep (stabilized_schur.py:313-334): null-space handling, Newton with
 * line search, FGMRES + block-Schur preconditioner; returns CFDH_E_DIVERGED
 * with stats->reason < 0 when the reference would raise RuntimeError. */
int cfdh_solve_step(cfdh_ctx *ctx, cfdh_stats *stats);

/* kind 0: F_D, 1: F_L over the exterior facets of `marker`
 * (/root/reference/src/scenarios/dfg_1.py:183-202; the scenario prints 500*F),
 * 2: ||u||_L2, 3: ||p||_L2 (/root/reference/src/scenario.py:315-324),
 * 4: ||u_sol||_inf, 5: ||u_prev||_inf, 6: ||u_sol-u_prev||_inf (scenario.py:268-280),
 * 7: volume flux  int u_sol . n ds  over the exterior facets of `marker` (outward normal; the outlet flow rates the
 * tree and bifurcation scenarios report).  Kinds 0/1 exist for gdim 2 only.
 * Sums/maxima over the owned part; the caller (or the communicator) reduces. */
int cfdh_functional(cfdh_ctx *ctx, int kind, int marker, double *out);

/* Wall shear stress, the per-step `assemble_wss()` of solverBase.py:163-195,
 * (1/FacetArea) * inner(w, Tt) * ds with T = -sigma(u_sol, p_sol) n, Tt = T - (T.n) n, assembled on
 * the device from the current solution into a P1 vector field (zero away from the boundary).
 * shear: nv x gdim host array, or NULL to compute without downloading. */
int cfdh_wall_shear_stress(cfdh_ctx *ctx, double *shear);

/* ---- multi-GPU ---------------------------------------------- */

/* Halo plan in local vertex numbers: for neighbour k, send_idx[send_ptr[k]..send_ptr[k+1])
 * are owned vertices whose values go to rank nbr_rank[k]; recv_idx likewise the
 * ghosts filled from it (ghosts must be numbered contiguously per neighbour, in
 * neighbour order, starting at nv_owned). */
int cfdh_set_halo(cfdh_ctx *ctx, int nnbr, const int32_t *nbr_rank, const int64_t *send_ptr,
                  const int32_t *send_idx, const int64_t *recv_ptr, const int32_t *recv_idx);
/* Global pressure space of a partitioned run.  The pressure part of the preconditioner is a Poisson-type
 * solve whose low modes couple all parts; a rank-local (block-Jacobi) version multiplies the FGMRES
 * iterations by ~10.  Every rank therefore receives the whole (replicated, geometry-only) mesh and the
 * global pressure-Dirichlet set, builds the same global Laplacian hierarchy and applies it redundantly to
 * the all-reduced right-hand side.  owned_global[nv_owned]: global vertex id of each owned local vertex
 * (local numbering of cfdh_create).  Call after the Dirichlet data are known; no-op need for one rank. */
int cfdh_set_global_pressure_space(cfdh_ctx *ctx, int64_t nv_global, int64_t nc_global, const int32_t *cells_global,
                                   const double *coords_global, const int32_t *owned_global, int64_t n_pbc,
                                   const int32_t *pbc_nodes_global);
/* RCCL over xGMI: rank 0 creates the 128-byte unique id, the launcher
 * broadcasts it, every rank calls cfdh_comm_init. */
int cfdh_comm_unique_id(void *id128);
int cfdh_comm_init(cfdh_ctx *ctx, const void *id128, int rank, int nranks);
/* Host-staged alternative (CPU/gloo tests, debugging): the library calls back
 * with host buffers.  allreduce: in-place sum (op 0) / max (op 1) of n doubles;
 * exchange: send/recv byte buffers per neighbour as laid out by cfdh_set_halo
 * (3 doubles per vertex: ux, uy, p). */
typedef int (*cfdh_allreduce_fn)(void *user, double *buf, int n, int op);
typedef int (*cfdh_exchange_fn)(void *user, const double *sendbuf, double *recvbuf);
int cfdh_comm_set_callbacks(cfdh_ctx *ctx, cfdh_allreduce_fn ar, cfdh_exchange_fn ex, void *user, int rank, int nranks);

/* ---- measurement ------------------------------------------------------------ */

/* HIP-event timing of the hot kernels on the library's stream.
 * kind 0: fused residual+Jacobian assembly, 1: monolithic SpMV, 2: tau moments,
 * 3: A00 SpMV (Chebyshev sweep, pc_type 0), 4: level-0 up-sweep of the pressure hierarchy (x = Sb b + Sc x_c; a
 * Jacobi sweep of the unfused cycle), 5: the same for the velocity hierarchy (two right-hand sides),
 * 8 / 9: level-0 down-sweep (b_c = G b) of the pressure / velocity hierarchy,
 * 7: EMPTY event pairs recorded when profiling is switched on (the per-launch overhead of the method). */
int cfdh_profile_enable(cfdh_ctx *ctx, int on);
int cfdh_profile_get(cfdh_ctx *ctx, int kind, double *total_ms, int64_t *launches);
int cfdh_profile_reset(cfdh_ctx *ctx);
/* sizes for roofline accounting: 0 nv_owned, 1 nv, 2 nc, 3 vertex-graph nnz,
 * 4 Sp nnz, 5 incidences, 6 AMG levels, 7 assembly workgroups, 8 velocity-proxy nnz;
 * communicator state: 9 padded part size of the pressure all-gather (0: all-reduce path), 10: RCCL attached,
 * 11: size of the replicated coarse level below the distributed finest pressure level (0: fully replicated cycle),
 * 12: overlapping (restricted additive Schwarz) velocity cycle in use;
 * counters since cfdh_create / cfdh_profile_reset: 13 all-reduce calls, 14 halo exchanges, 15 host synchronisations
 * (stream/event waits for scalars), 16 FGMRES iterations, 17 all-gathers; 18: communicator size;
 * fused AMG cycle: 19 / 20 entries of Sb + Sc on level 0 (pressure / velocity hierarchy), 21 / 22 entries of G on
 * level 0, 23 / 24 size of level 1, 25: fused cycle in use; 26: gdim;
 * 27: microseconds the last preconditioner build took on the device (0: it was built on the host); 28: element type (CFDH_ELEM_*),
 * 29: nodes per cell;
 * 30 + l / 40 + l: rows / entries of level l of the velocity hierarchy, 50 + l / 60 + l: of the pressure hierarchy (l < 10, 0 past the end);
 * 70: linear solves that started from a projected initial guess (cfdh_options.ksp_guess), 71: their mean |r0| / |b| in units of 1e-6 */
int64_t cfdh_info(const cfdh_ctx *ctx, int what);

#ifdef __cplusplus
}
#endif
#endif /* CFDH_H */
