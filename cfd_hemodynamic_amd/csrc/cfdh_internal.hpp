// Internal declarations of libcfdh.so (gfx950).  See include/cfdh.h for the ABI
// and DESIGN.md for the data layout.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "cfdh.h"

// geometry of a 2-D assembly workgroup (overridable for tuning experiments, tools/build_variant.sh)
#ifndef CFDH_MAX_INC
#define CFDH_MAX_INC 256   // incidences (= threads) per assembly workgroup
#endif
#ifndef CFDH_MAX_BV
#define CFDH_MAX_BV 192    // distinct vertices staged in LDS per assembly workgroup (8-bit local index)
#endif
#ifndef CFDH_MAX_BC
#define CFDH_MAX_BC 192    // distinct cells staged in LDS per assembly workgroup
#endif
#ifndef CFDH_MAX_ROWS
#define CFDH_MAX_ROWS 128  // rows per assembly workgroup
#endif

#if defined(__HIPCC__)
// 1/sqrt(x) for normal positive x: hardware estimate (v_rsq_f64) + two Newton steps (error -> ~1 ulp); the compiler's
// IEEE sequence for 1.0 / sqrt(x) (sqrt with scaling + full division) is ~3x as many instructions, and the tau-moment
// kernels evaluate it at 49 / 171 points per cell
__device__ __forceinline__ double cfdh_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
  y = __builtin_fma(y, __builtin_fma(-h * y, y, 0.5), y);
  return y;
}
#endif

template <class T>
struct dbuf {
  T *p = nullptr;
  size_t n = 0;
  dbuf() = default;
  dbuf(const dbuf &) = delete;
  dbuf &operator=(const dbuf &) = delete;
  ~dbuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  hipError_t alloc(size_t count) {
    if (count == n && p) return hipSuccess;
    release();
    if (count == 0) return hipSuccess;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e == hipSuccess) n = count;
    return e;
  }
  hipError_t upload(const std::vector<T> &h, hipStream_t s) {
    hipError_t e = alloc(h.size());
    if (e != hipSuccess || h.empty()) return e;
    return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s);
  }
  hipError_t zero(hipStream_t s) { return n ? hipMemsetAsync(p, 0, n * sizeof(T), s) : hipSuccess; }
  void adopt(dbuf &o) {  // take over o's allocation
    if (&o == this) return;
    release();
    p = o.p; n = o.n;
    o.p = nullptr; o.n = 0;
  }
};

struct CsrHost {
  int n = 0, m = 0;
  std::vector<int> rowptr, col;
  std::vector<double> val;
  int nnz() const { return (int)col.size(); }
};

struct CsrDev {
  int n = 0, m = 0, nnz = 0;
  dbuf<int> rowptr, col;
  dbuf<double> val;
  // SELL-64 copy (sliced ELLPACK, slices of 64 consecutive rows, column-major inside a slice):
  // one lane per row streams fully coalesced, no cross-lane reduction.  sptr[s] = first padded
  // entry of slice s (in units of entries); width of slice s = (sptr[s+1]-sptr[s]) / 64.
  int nslice = 0, sell_maxw = 0;  // widest slice (entries per row)
  dbuf<int> sptr, scol;
  // SELL values are kept in fp32: they feed preconditioner sweeps only (FGMRES is flexible and measures the
  // true fp64 residual), and the matrix stream is 60 % of a fine-level sweep's traffic
  dbuf<float> sval, svalw;  // svalw: values scaled by a column weight (Jacobi pre-sweep), optional
  dbuf<float> valf;         // CSR-order fp32 copy of val (composite operators of the fine levels), optional
  // Measured on the level-0 up-sweep of the fused cycle (14.8 us) and dropped: column and fp32 value packed into one
  // 8-byte word (17.5 us); four lanes per row on 16-row slices (18.5 us).
};
enum { CFDH_UP_CSR = 1, CFDH_UP_SELL = 2, CFDH_UP_CSRF = 4};  // parts of a CsrDev to upload

struct AmgLevel {
  int n = 0;
  CsrDev A, P, R;
  dbuf<double> dinv, wdinv, x, b, r, d0, d1;  // wdinv: Jacobi weight (1/theta, or 1 on diagonal-only rows) * dinv; work vectors hold ncol values per row
  double lmax = 0, lmin = 0;
  // Composite operators of the fused V(1,1) Jacobi cycle (AmgHier::fused).  With W = diag(wdinv) the sweeps of a level
  //   pre : xa = W b ; r = b - A xa ; b_c = R r              ==>  b_c = G b ,              G  = R (I - A W)
  //   post: x1 = xa + P x_c ; x = x1 + W (b - A x1)          ==>  x = Sb b + Sc x_c ,      Sb = 2W - W A W , Sc = (I - W A) P
  // are the SAME linear maps, applied with one kernel per level and direction instead of two.  On the level above
  // a dense coarsest solve the correction is folded in as well: x = Sb b + D b_c , D = Sc A_c^-1 (dense, fp32).
  CsrDev G, Sb, Sc;
  dbuf<float> D;
  int Dn = 0;        // columns of D (= size of the coarsest level); 0: not folded
  bool fine = false; // short regular rows: fp32 values in G (down-sweep)
  bool sell = false; // ... and SELL-64 / fp32 for Sb, Sc (up-sweep); not for three right-hand sides on tetrahedra (measured slower)
};

// One smoothed-aggregation hierarchy.  ncol = 2 applies the same scalar operators to two
// right-hand sides at once (the interleaved velocity components), halving matrix traffic.
struct AmgHier {
  std::vector<AmgLevel *> lev;
  dbuf<double> coarse_inv;  // dense inverse of the coarsest operator
  int coarse_n = 0, ncol = 1;
  long long fine_nnz = 0;
  bool valid = false;
  bool fused = false;  // composite operators present on every level (built for damped-Jacobi smoothing)
  long long nnz_G0 = 0, nnz_S0 = 0;  // entries of G and of Sb + Sc on the finest level (roofline accounting)
  // host copies of the finest level (operator, prolongator, Jacobi weights): kept on request so that a
  // partitioned run can cut its rows of the replicated pressure hierarchy out of them
  bool keep_host0 = false;
  CsrHost h_A0, h_P0;
  std::vector<double> h_wdinv0;
  void clear() {
    for (AmgLevel *l : lev) delete l;
    lev.clear();
    valid = false;
  }
  ~AmgHier() { clear(); }
};

struct ProfSlot {
  double total_ms = 0;
  long long launches = 0;
};

struct cfdh_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;

  // geometric dimension: 2 (triangles, 3x3 vertex blocks) or 3 (tetrahedra, 4x4 vertex blocks; single GPU)
  int dim = 2;
  // element: 0 P1 simplices (closed-form kernels), 1 P2 triangles / tetrahedra, 2 Q1 parallelograms / parallelepipeds -- `gen`: the quadrature kernels of
  // cfdh_gen.hip assemble (also for etype 0 when created as CFDH_ELEM_P1_GENERIC); "vertex" then means node everywhere below
  int etype = 0, nloc = 3;
  bool gen = false;
  dbuf<int> gslot;             // [nc][nloc * nloc] position of the block of the local node pair (a, b) in the staging array gE
  // two-phase assembly of the generic elements (cfdh_gen.hip): staged element blocks / residual rows ordered by destination and
  // the extents of every block entry / node in them
  dbuf<int> g_eptr, g_fptr, g_fdst;   // [nnzv + 1], [nv + 1], [nc][nloc]
  dbuf<double> gE, gEF;               // [nc nloc^2][9], [nc nloc][3]
  dbuf<unsigned short> gflag;  // [nc] bit f: exterior facet f, bit 8 + f: backflow facet f
  dbuf<int> g3_fcells;         // 3-D generic elements: the cells with an exterior facet (gen3_facet_kernel)
  int g3_nfcells = 0;
  // P2: the P1 subspace as the first coarse level of both hierarchies (p-multigrid step): prolongator [nodes x vertex nodes],
  // 1 at a vertex node, 1/2 + 1/2 at an edge node (host copy; internal numbering)
  CsrHost gen_P1;
  // sizes (local part): nv = nvo owned + ng ghosts
  int nv = 0, nvo = 0, ng = 0, nc = 0, nfac = 0;
  int NL = 0;   // vector length incl. ghost tail = 3*nvo + 3*ng
  int NO = 0;   // owned compact length 3*nvo

  // user <-> internal vertex numbering
  std::vector<int> perm;   // user -> internal
  std::vector<int> iperm;  // internal -> user
  std::vector<int> h_cells;       // internal ids, [nc][3]
  std::vector<int> cell_user;     // internal cell -> user cell
  std::vector<double> h_coords;   // internal order [nv][2]
  std::vector<int> fac_cell, fac_local, fac_marker;  // internal cell ids
  std::vector<int> fac_user;  // index of the kept facet in the caller's facet list (cfdh_create order)
  int nfac_user = 0;

  bool params_set = false;
  double dt = 0, rho = 0, mu = 0, muf = 0, f[3] = {0, 0, 0};
  cfdh_options opt;

  // device mesh
  dbuf<double> coords;          // [nv][2]
  dbuf<int> cells;              // [nc][3]
  dbuf<unsigned char> cflag;    // [nc] exterior-facet bits
  dbuf<double> mom;             // [nc][8]: M00 M01 M02 M11 M12 M22 L pad
  dbuf<unsigned char> cell_owned;  // [nc] 1 when this rank integrates the cell in global functionals
  bool mom_valid = false;

  // vertex graph of owned rows (columns may be ghosts), block values
  std::vector<int> h_vptr, h_vcol, h_vdiag;
  int nnzv = 0;
  dbuf<int> vptr, vcol, vdiag;
  dbuf<double> A00, A01, A10, A11;  // [nnzv][4], [nnzv][2], [nnzv][2], [nnzv]
  bool jac_valid = false;

  // incidences (row vertex, cell), grouped in workgroup blocks of whole rows
  int ninc = 0, nblk = 0;
  bool gs_refine_long = false;  // FGMRES: a long cycle lost orthogonality on this context once (cfdh_solver.cpp)
  dbuf<unsigned> inc_slot;   // per lane: slot0 | slot1<<8 | slot2<<16 | a<<24 | emit_v2<<26 | has_prev<<27
  dbuf<unsigned> inc_rank;   // per lane: position in row | row length<<8 | (lane offset of the fan predecessor + 64)<<16
  dbuf<int> blk_row;         // [nblk+1]
  dbuf<int> blk_vptr, blk_vlist;  // per-block list of the vertices its cells touch
  dbuf<int> blk_cptr, blk_clist;  // per-block list of distinct cells
  dbuf<unsigned> inc_loc;    // per lane: lcell | lv0<<8 | lv1<<16 | lv2<<24 (block-local, rotated); ~0u = idle lane
  dbuf<int> wave_maxlen;     // per wavefront: longest row (bound of the segmented reduction)

  // 3-D assembly (cfdh3_*): workgroups of consecutive rows whose value slots are accumulated in LDS
  int a3_nblk = 0;
  dbuf<int> a3_blk_row, a3_blk_iptr;   // [nblk+1] first row / first incidence of a workgroup
  dbuf<int> a3_inc_cell;               // per incidence: cell * 4 + local index of the row vertex
  dbuf<int> a3_inc_row;                // per incidence: row relative to the workgroup's first row
  dbuf<unsigned long long> a3_inc_slots;  // per incidence: 4 x 16-bit slot offsets (relative to the workgroup's first slot) of the cell's columns

  // Dirichlet data (host master copies in internal numbering)
  std::vector<unsigned char> h_bcflag;  // bit0 ux, bit1 uy, bit2 p
  std::vector<double> h_bcval;          // [nv][3]
  std::vector<double> h_bcmult;         // [nv][3]
  bool bc_dirty = true;
  // vertices written by cfdh_add_dirichlet since the last clear / since the last upload: clearing and uploading touch these
  // only (a pulsatile inlet re-sends every object every step; the full arrays are 49 B per vertex)
  std::vector<int> bc_touched, bc_pending;
  size_t bc_touched_sent = 0;          // bc_touched[0 .. sent) are on the device already (upload_bc takes the rest as pending)
  std::vector<int> bc_mark;            // epoch marker per vertex (de-duplication)
  int bc_epoch = 0;
  bool bc_full_upload = true;          // device arrays not initialised yet
  dbuf<int> bc_uidx;                   // staging of the sparse update
  dbuf<unsigned char> bc_uflag;
  dbuf<double> bc_uval, bc_umult;
  int n_pbc = 0;
  dbuf<unsigned char> bcflag;
  dbuf<double> bcval, bcmult;

  // exterior facets on device (functionals)
  dbuf<int> d_fac_cell, d_fac_local, d_fac_marker;

  // state: layout [u owned 2*nvo | p owned nvo | ghosts 3*ng (ux,uy,p)]
  dbuf<double> x, xt, xprev, xprev2, F, dvec;
  dbuf<double> wss;  // [nv][2] wall shear stress of the last cfdh_wall_shear_stress call (allocated on first use)
  double ts_theta = 0.5, ts_a[3] = {1.0, -1.0, 0.0};  // cfdh_set_time_scheme
  bool ds_terms = true;      // cfdh_set_boundary_terms
  double bf_beta = 0.0;
  int bf_marker = -1;
  bool state_set = false;

  // reductions
  dbuf<double> red_partial, red_out;
  double *h_pinned = nullptr;  // pinned host scratch for scalar read-back
  double *h_stage = nullptr;   // pinned staging of whole fields (set_state / get_* of the host-copy loop), NL doubles
  double *h_pinned_dev = nullptr;  // the same buffer as the device sees it (kernels write read-back scalars into it)
  const double *mirror_src = nullptr;  // device scalars whose host-mapped copy is current (see read_scalars)
  int mirror_cnt = 0;
  hipEvent_t ev_h = nullptr;   // marks 'Gram-Schmidt coefficients are in h_pinned'
  // FGMRES read-back ring: the host processes the Gram-Schmidt coefficients of an iteration up to KRING - 2 iterations after it
  // was launched (cfdh_fgmres), so every iteration in flight owns a slot of host-mapped memory and an event
  static constexpr int KRING = 12;
  double *h_ring = nullptr, *h_ring_dev = nullptr;
  size_t h_ring_stride = 0;
  hipEvent_t ev_ring[KRING] = {};
  long long n_krylov_discarded = 0;  // iterations launched ahead of the convergence test and thrown away (cfdh_info 73)
  int red_blocks = 0;

  // Krylov workspace
  int kry_m = 0;
  // projected initial guess (cfdh_options.ksp_guess): ring of earlier solutions per Newton index, [GUESS_NEWTON][ksp_guess] vectors
  // of leading dimension ld; guess_slot = Newton index of the solve in progress (-1: zero guess)
  static constexpr int GUESS_NEWTON = 4;
  dbuf<double> guessU, guessX;  // guessX: the Newton iterates x_k of the step in progress (k < GUESS_NEWTON)
  bool guess_stored[GUESS_NEWTON] = {false, false, false, false};
  int guess_m = 0, guess_slot = -1;
  int guess_cnt[GUESS_NEWTON] = {0, 0, 0, 0}, guess_head[GUESS_NEWTON] = {0, 0, 0, 0};
  long long n_guess_solves = 0;
  double guess_reduction_sum = 0.0;  // sum of |r0| / |b| over the solves that started from a projected guess
  dbuf<double> kV, kZ, kw, kh;  // V[(m+1)*NL], Z[m*NL], w[NL], h[2*(m+1)+2]
  dbuf<float> kV32;             // fp32 copy of V for the Gram-Schmidt passes of long cycles (allocated on first use)
  bool krylov_fp32_ok = true;   // cleared when the orthogonality watchdog trips on a cycle that used the copy
  int guess_last_its[4] = {0, 0, 0, 0};  // iterations of the last solve per Newton index (expected length of the next)
  dbuf<double> ky;

  // preconditioner
  dbuf<double> dinvA;          // 1/diag(A00), [2*nvo]
  double lmaxA = 0;
  dbuf<double> pu0, pu1, pu2, pr, pp0, pp1;  // PC work vectors
  dbuf<double> prand;                        // fixed start vector of the power iteration
  dbuf<double> cheb_coef;                    // [1/theta, (c1,c2) per step] of the A00 Chebyshev solve
  struct PcGraph { const double *r; double *z; hipGraphExec_t exec[6]; };
  std::vector<PcGraph> pc_graphs;            // one captured preconditioner application per Krylov slot
  bool pc_graph_valid = false, capturing = false, use_graph = true;
  AmgHier hS;               // SELFP Schur matrix Sp (pc_type 0)
  AmgHier hL;               // pressure Laplacian (pc_type 1), built once per Dirichlet set
  AmgHier hA;               // scalar proxy of A00 applied to both velocity components (pc_type 1)
  AmgLevel Hlev;            // H = (I + a'T) M_l + b' A11 of the Cahouet-Chabard Schur approximation
  dbuf<double> ccMl;        // lumped pressure mass (0 on pressure-Dirichlet rows)
  dbuf<unsigned char> ccPbc;
  double cc_alpha = 0, cc_beta = 0;
  // Cahouet-Chabard combination z_p = alpha t + beta zH (r on Dirichlet rows) applied in the epilogue of the last
  // kernel of the pressure cycle instead of a kernel of its own
  int up0_rows = 0;  // > 0: the fused cycle's finest up-sweep computes the first up0_rows rows only (owned rows of the overlapping velocity cycle)
  struct Epilogue { bool on = false, done = false; double alpha = 0, beta = 0; const double *zH = nullptr, *r = nullptr; const unsigned char *pbc = nullptr; double *out = nullptr; } epi;
  std::vector<double> h_Lval, h_Ml;  // P1 stiffness on the vertex graph, lumped mass (geometry only)
  dbuf<double> d_Lval, d_Ml;         // device copies (device-side preconditioner set-up)
  dbuf<double> amg_rand;             // start vector of the spectral-bound power iterations (device-side set-up)
  double ms_pc_build_dev = 0;        // time spent in the last device-side build (verbose / tests)
  long long bc_version = 0;
  // replicated global pressure space (multi-rank pc_type 1)
  int gp_n = 0;                       // global vertex count (0: not set, pressure solve is rank-local)
  CsrHost gp_L;                       // global Laplacian with Dirichlet rows (host, for the hierarchy)
  bool gp_singular = false, gp_dirty = false;
  AmgHier hLg;                        // its hierarchy (identical on every rank)
  dbuf<int> gp_l2g;                   // [nvo] global id of owned vertex (internal numbering)
  dbuf<double> gp_rhs, gp_sol;        // [gp_n]
  // RCCL runs: the owned slices travel by all-gather (half the bytes of the all-reduce of a zero-padded vector):
  // every rank sends its owned values ordered by global id, padded to the largest part
  // restricted additive Schwarz with one layer of overlap for the velocity block of a partitioned run: the local
  // hierarchy covers owned + ghost vertices (ghost rows come from their owners), the residual is halo-exchanged
  // before the cycle and only the owned part of the result is kept
  bool ras = false;
  std::vector<int> h_gid;              // [nv] global id of every local vertex (internal numbering)
  std::vector<int> h_g2l;              // [gp_n] local internal index of a global vertex, -1 if not local
  dbuf<double> ras_b, ras_x;           // [2 nv] extended right-hand side / solution of the velocity cycle
  // ghost rows of the velocity proxy fetched from their owners on the device (cfdh_proxy_ras_dev): the pattern is exchanged once
  // per halo plan, the values at every rebuild in ceil(maxlen / (dim + 1)) halo exchanges
  struct RasPlan {
    bool ready = false;
    int maxlen = 0;              // longest row of the vertex graph over all ranks
    int nent = 0;
    dbuf<int> gptr, gcol, gsrc;  // ghost rows: [ng + 1]; per entry the local column (ascending) and its position in the staging row
    dbuf<double> gval;           // [ng][maxlen] entry k of the owner's vertex-graph row, as received
  } rasp;
  // distributed finest level of the replicated pressure hierarchy: every rank smooths its own rows (owned rows,
  // owned + ghost columns), the coarse right-hand side is all-reduced and levels >= 1 stay replicated
  struct DistL0 {
    bool on = false;
    bool ghost_rhs = true;  // exchange the ghost layer of the right-hand side before the pre-smoothing (false: pre-smoothed iterate zero on the ghosts)
    int n1 = 0;
    CsrDev A;      // owned rows x local (owned + ghost) columns
    CsrDev P;      // local rows (owned + ghost) x coarse columns
    CsrDev PT;     // coarse rows x owned columns (restriction of the owned residual)
    dbuf<double> wdinv, b, xa, r, x1;
  } dl0;
  bool gp_allgather = false;
  int gp_maxcnt = 0;
  dbuf<int> gp_send_idx;              // [nvo] local (internal) index of the k-th owned vertex in global-id order
  dbuf<int> gp_src_idx;               // [gp_n] position of global vertex g in the gathered buffer
  dbuf<double> gp_sendbuf, gp_recvbuf;  // [gp_maxcnt], [nranks * gp_maxcnt]
  dbuf<double> pcw;                   // [NL] scratch vector with ghost tail for the coupling products
  double *h_big = nullptr;            // pinned staging of the host-callback all-reduce
  size_t h_big_n = 0;
  std::vector<unsigned char> hL_pbc;  // Dirichlet set hL was built for
  int hL_singular = -1;
  bool pc_valid = false;
  int pc_its_ref = 0;       // FGMRES iterations right after the last refresh
  int steps_since_refresh = 0;
  int singular = 0;         // constant pressure in the null space (tested per step)

  // halo / comm
  int nnbr = 0;
  std::vector<int> nbr_rank;
  std::vector<long long> send_ptr, recv_ptr;
  dbuf<int> send_idx;        // internal owned vertex ids, concatenated per neighbour
  dbuf<double> send_buf;     // 3 doubles per send vertex
  std::vector<double> h_send, h_recv;
  int rank = 0, nranks = 1;
  double nvo_global = 0;     // number of pressure dofs over all ranks
  void *nccl_comm = nullptr;
  cfdh_allreduce_fn cb_ar = nullptr;
  cfdh_exchange_fn cb_ex = nullptr;
  void *cb_user = nullptr;

  // profiling
  bool prof_on = false;
  ProfSlot prof[12];
  std::vector<hipEvent_t> ev_pool;
  struct EvRec { int kind; hipEvent_t a, b; };
  std::vector<EvRec> ev_pending;
  size_t ev_next = 0;

  cfdh_stats last_stats;
  // communication / synchronisation counters (cfdh_info 13..17): all-reduces, halo exchanges, host synchronisations
  // of the solve, FGMRES iterations, all-gathers -- cumulative, reset by cfdh_profile_reset
  long long n_allreduce = 0, n_halo = 0, n_host_sync = 0, n_krylov = 0, n_allgather = 0;
  long long n_attainable_stops = 0;  // FGMRES solves ended by the attainable-accuracy rule (cfdh_info 72), cumulative over the context's life
};

// ---- error helpers -----------------------------------------------------------
int cfdh_fail(cfdh_ctx *c, int code, const char *fmt, ...);
#define HIPCHK(c, call)                                                                           \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess) return cfdh_fail((c), CFDH_E_HIP, "%s: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                           __FILE__, __LINE__);                                   \
  } while (0)
#define CHK(call)            \
  do {                       \
    int r_ = (call);         \
    if (r_ != 0) return r_;  \
  } while (0)

// ---- setup (cfdh_setup.cpp) ------------------------------------------------------
int cfdh_build_mesh(cfdh_ctx *c, int64_t nv, int64_t nvo, int64_t nc, const int32_t *cells, const double *coords,
                    int64_t nfac, const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker);
int cfdh_amg_setup(cfdh_ctx *c, AmgHier &H, const CsrHost &A, bool singular, int ncol);
int cfdh_level_setup(cfdh_ctx *c, AmgLevel &L, const CsrHost &A, double ratio, int ncol, std::vector<double> *w_out = nullptr);
int cfdh_upload_csr(cfdh_ctx *c, const CsrHost &H, CsrDev &D, const std::vector<double> *colw = nullptr, int parts = CFDH_UP_CSR | CFDH_UP_SELL);
int cfdh_aggregate_host_csr(const CsrHost &A, double theta, std::vector<int> &agg);

// ---- device-side hierarchy set-up (cfdh_amg_dev.hip) ---------------------------------
bool cfdh_amg_dev_enabled(const cfdh_ctx *c);  // fused Jacobi cycle requested and CFDH_AMG_HOST unset
int cfdh_amg_setup_dev(cfdh_ctx *c, AmgHier &H, CsrDev &A0, bool singular, int ncol);   // A0 (device CSR) is consumed
int cfdh_level_setup_dev(cfdh_ctx *c, AmgLevel &L, CsrDev &A, double ratio, int ncol);  // A is consumed
int cfdh_proxy_dev(cfdh_ctx *c, CsrDev &out);                       // scalar proxy of A00 on the owned vertices
int cfdh_proxy_ras_dev(cfdh_ctx *c, CsrDev &out);                   // the same on owned + ghost vertices (collective: halo exchanges)
int cfdh_cc_h_dev(cfdh_ctx *c, double alpha, double beta, CsrDev &out);  // H = (I + a'T) M_l + b' A11 (rows of ccPbc & 1: identity)

// ---- tetrahedra (cfdh3_setup.cpp, cfdh3_kernels.hip) -------------------------------
#define CFDH3_MAX_SLOTS 320   // value slots (16 doubles each) accumulated in LDS per assembly workgroup
int cfdh_build_mesh3(cfdh_ctx *c, int64_t nv, int64_t nvo, int64_t nc, const int32_t *cells, const double *coords,
                     int64_t nfac, const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker);
int k3_upload_quadrature(cfdh_ctx *c);
int k3_moments(cfdh_ctx *c);
int k3_assemble(cfdh_ctx *c, const double *xstate, int mode);
int k3_spmv_full(cfdh_ctx *c, const double *x, double *y);
int k3_spmv_full_multi(cfdh_ctx *c, const double *X, double *Y, int ld, int nvec);
int k3_spmv_block(cfdh_ctx *c, int blk, const double *x, double *y, const double *b);
int k3_spmv_block_ghost(cfdh_ctx *c, int blk, const double *xv, double *y, const double *b);  // xv: full vector with refreshed ghost tail  // 2: b - A01 x_p ; 3: b - A10 x_u (b may be null)
int k3_nullspace_test(cfdh_ctx *c, double *nrm, double *absnrm);
int k3_functional(cfdh_ctx *c, int kind, int marker, double *out);
int k3_wss(cfdh_ctx *c, double *out);

// ---- nodal elements beyond P1 (cfdh_gen.hip) ------------------------------------------
int kg_upload_tables(cfdh_ctx *c);
int cfdh_build_mesh_gen(cfdh_ctx *c, int etype, int64_t nv, int64_t nv_owned, int64_t nc, const int32_t *cells, const double *coords, int64_t nfac,
                        const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker);
int cfdh_gen_element_stiffness(const cfdh_ctx *c, const int32_t *v, const double *X, double *K);
int cfdh_facet_nodes(const cfdh_ctx *c, int f, int out[3]);  // local nodes of local facet f; returns their number
int kg_assemble(cfdh_ctx *c, const double *xstate, int mode);
int kg_functional_partials(cfdh_ctx *c, int kind, int marker, int nb);  // per-block partial sums into red_partial
int kg_wss(cfdh_ctx *c, double *out);

// ---- nodal elements beyond P1 in 3-D: Q1 hexahedra, P2 tetrahedra (cfdh_gen3.hip) -----------------
int kg3_upload_tables(cfdh_ctx *c);
int cfdh_build_mesh_gen3(cfdh_ctx *c, int etype, int64_t nv, int64_t nv_owned, int64_t nc, const int32_t *cells, const double *coords, int64_t nfac,
                         const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker);
int cfdh_gen3_element_stiffness(const cfdh_ctx *c, const int32_t *v, const double *X, double *K);
int cfdh_facet_nodes3(const cfdh_ctx *c, int f, int out[8]);  // local nodes of local facet f of a 3-D cell; returns their number
int kg3_assemble(cfdh_ctx *c, const double *xstate, int mode);
int kg3_functional_partials(cfdh_ctx *c, int kind, int marker, int nb);
int kg3_wss(cfdh_ctx *c, double *out);

// ---- kernels (cfdh_kernels.hip) ----------------------------------------------------
void prof_begin(cfdh_ctx *c, int kind);
void prof_end(cfdh_ctx *c, int kind);
void prof_flush(cfdh_ctx *c);

int k_upload_quadrature(cfdh_ctx *c);
int k_halo_pack(cfdh_ctx *c, const double *vec);
int v_pointwise_mult(cfdh_ctx *c, int n, const double *a, const double *b, double *out);
int k_moments(cfdh_ctx *c);
int k_assemble(cfdh_ctx *c, const double *xstate, int mode);  // mode 0: F only, 1: F+J, 2: F with lifting (no J write)
int k_spmv_full(cfdh_ctx *c, const double *x, double *y);
int k_spmv_full_multi(cfdh_ctx *c, const double *X, double *Y, int ld, int nvec);  // Y_v = J X_v, vectors ld apart
int k_spmv_block(cfdh_ctx *c, int blk, const double *x, double *y, const double *b, double alpha);  // y = b*? see .hip
int k_spmv_block_ghost(cfdh_ctx *c, int blk, const double *xv, double *y, const double *b);
int k_extract_diag(cfdh_ctx *c);
int k_cheb_a00(cfdh_ctx *c, const double *b, double *x);
int k_cheb_a00_coeffs(cfdh_ctx *c);  // x = Cheb_k(A00) b, zero initial guess
int k_csr_spmv(cfdh_ctx *c, const CsrDev &A, const double *x, double *y, int mode, const double *b);  // mode 0: y=Ax, 1: y=b-Ax, 2: y+=Ax
int k_csr_spmv_ncol(cfdh_ctx *c, const CsrDev &A, const double *x, double *y, int mode, const double *b, int ncol);
int k_amg_vcycle(cfdh_ctx *c, AmgHier &H, const double *b, double *x);
int k_level_smooth(cfdh_ctx *c, AmgLevel *L, const double *b, double *x, int degree);
bool k_cc_cheb2_scale(cfdh_ctx *c, AmgLevel *L, const double *b, double *x, const double *ml, double *y);
int k_cc_scale(cfdh_ctx *c, int n, const double *ml, const double *z, double *y);
int k_dl0_down(cfdh_ctx *c, const double *halo_vec);  // distributed level 0 of the pressure cycle: down sweep + restriction
int k_dl0_up(cfdh_ctx *c, double *out);               // replicated coarse cycle, prolongation, post-smoothing
int k_ext_pack(cfdh_ctx *c, const double *vec, double *out);  // [u | p | ghost triplets] -> nv contiguous (ux,uy) pairs
int k_scatter_global(cfdh_ctx *c, int n, const int *l2g, const double *loc, double *glob);
int k_gather_global(cfdh_ctx *c, int n, const int *l2g, const double *glob, double *loc);
int k_cc_combine(cfdh_ctx *c, int n, double alpha, double beta, const double *t, const double *z, const double *r, const unsigned char *pbc, double *out, double *out2 = nullptr);
int k_nullspace_test(cfdh_ctx *c, double *nrm, double *absnrm);
int k_bc_scatter(cfdh_ctx *c, int n, int ncomp, const int *idx, const unsigned char *flag, const double *val, const double *mult);

// vector ops on [0,n)
int v_copy(cfdh_ctx *c, int n, const double *x, double *y);
int v_zero(cfdh_ctx *c, int n, double *y);
int v_axpy(cfdh_ctx *c, int n, double a, const double *x, double *y);
int v_waxpy(cfdh_ctx *c, int n, double a, const double *x, const double *y, double *w);  // w = y + a x
int v_scale(cfdh_ctx *c, int n, double a, double *x);
int v_dot(cfdh_ctx *c, int n, const double *x, const double *y, double *out_host);   // with comm reduction
int v_norm2(cfdh_ctx *c, int n, const double *x, double *out_host);
int v_norminf_diff(cfdh_ctx *c, int n, const double *x, const double *y, double *out_host);  // y may be null
int v_sub_mean(cfdh_ctx *c, int n, double *p);  // remove the (global) mean of p[0..n)
int v_multidot(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *w, double *h_dev, bool with_ww, double *mirror = nullptr,
               bool reduce_ranks = true);  // mirror: device view of host-mapped memory that receives the reduced values as well
int v_scale_to(cfdh_ctx *c, int n, double a, const double *x, double *y);  // y = a x
int v_multidot32(cfdh_ctx *c, int n, const float *V, int ld, int nvec, const double *w, double *h_dev, double *mirror);  // fp32 copy of the basis
int v_gs_update32(cfdh_ctx *c, int n, const float *V, int ld, int nvec, const double *h_dev, const double *w, double *vn, float *v32n,
                  double *s_dev, double *mirror);
int v_store32(cfdh_ctx *c, int n, const double *v, float *v32);
int v_gram(cfdh_ctx *c, int n, const double *W, int ld, int k, const double *b, double *out_dev);  // out[8 i + q] = W_q . (W_i | b), rank-local
int v_multiaxpy(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *h_dev, double *w);  // w -= sum h_i V_i
// Gram-Schmidt update fused with the normalisation: vn = (w - sum h_i V_i) / s, s = sqrt(h[nvec] - sum h_i^2) (h[nvec] = w.w);
// s (or sqrt(w.w) when the difference cancels) is stored in *s_dev
int v_gs_update_normalize(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *h_dev, const double *w, double *vn, double *s_dev);
int v_norm_to_dev(cfdh_ctx *c, int n, const double *w, double *out_dev);  // ||w|| (global) into device scalar
int v_norm_to_dev_local(cfdh_ctx *c, int n, const double *w, double *out_dev);  // no reduction over the ranks
int v_scale_inv_dev(cfdh_ctx *c, int n, const double *w, const double *nrm_dev, double *v);  // v = w / *nrm
int v_lincomb(cfdh_ctx *c, int n, const double *Z, int ld, int nvec, const double *y_dev, double *x);  // x += sum y_k Z_k
int v_pack_state(cfdh_ctx *c, const double *u_user, const double *p_user, double *dst);  // host staging helpers
int k_functional(cfdh_ctx *c, int kind, int marker, double *out);
int k_wss(cfdh_ctx *c, double *out);

// ---- comm (cfdh_comm.cpp) ----------------------------------------------------------
int comm_allreduce_dev(cfdh_ctx *c, double *dev, int n, int op);
int comm_allgather_dev(cfdh_ctx *c, const double *send, double *recv, int count);  // RCCL communicators only  // in-stream
int comm_halo(cfdh_ctx *c, double *vec);                          // fill the ghost tail of vec
int comm_finalize(cfdh_ctx *c);

// ---- solver (cfdh_solver.cpp) ------------------------------------------------------
int cfdh_pc_update(cfdh_ctx *c, bool force_refresh);
int cfdh_pc_apply(cfdh_ctx *c, const double *r, double *z);
int cfdh_host_threads();  // cfdh_setup.cpp: thread count of the host loops (CFDH_HOST_THREADS, default 8)
int cfdh_fgmres(cfdh_ctx *c, const double *b, double *x, int *its, int *reason, double bnorm = -1.0);
int v_norm2_pair(cfdh_ctx *c, int n, const double *x, const double *y, double *nx, double *ny);
int cfdh_newton_step(cfdh_ctx *c, cfdh_stats *st);
int cfdh_download_blocks(cfdh_ctx *c, std::vector<double> &a00, std::vector<double> &a01, std::vector<double> &a10,
                         std::vector<double> &a11);
