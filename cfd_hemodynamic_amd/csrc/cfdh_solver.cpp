// Host driver of the per-step solve (stabilized_schur.py:313-334): Newton with a
// backtracking line search (SNES newtonls/bt), right-preconditioned FGMRES
// (KSPFGMRES, restart/caps of :272-273) and a GPU block-Schur preconditioner in
// the factorisation the reference configures (PCFIELDSPLIT Schur, :231-235) with
// GPU-friendly sub-solvers instead of GMRES+ILU(0)/ILU(0):
//   pc_type 1 (default): velocity block = one AMG V-cycle of the scalar proxy of A00
//     (both components at once); Schur complement of Cahouet-Chabard type,
//     S^-1 ~ (a' L^-1 + b' M_l^-1) H^-1 -- Chebyshev(2) on the mass-like H, one AMG
//     V-cycle on the P1 Laplacian L; block upper-triangular (default) / FULL / lower;
//   pc_type 0: the reference's SELFP matrix Sp = A11 - A10 diag(A00)^-1 A01 with an
//     AMG V-cycle, Jacobi-Chebyshev on A00.
// Hierarchies are built on the host and lagged.  All vectors stay in HBM; the host
// sees a few scalars per iteration.
#include <algorithm>
#include <chrono>
#include <cmath>

#include "cfdh_internal.hpp"

int cfdh_host_threads();

static double wall_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int cfdh_download_blocks(cfdh_ctx *c, std::vector<double> &a00, std::vector<double> &a01, std::vector<double> &a10,
                         std::vector<double> &a11) {
  const size_t nz = (size_t)c->nnzv, d = (size_t)c->dim;
  a00.resize(d * d * nz); a01.resize(d * nz); a10.resize(d * nz); a11.resize(nz);
  HIPCHK(c, hipMemcpyAsync(a00.data(), c->A00.p, sizeof(double) * d * d * nz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(a01.data(), c->A01.p, sizeof(double) * d * nz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(a10.data(), c->A10.p, sizeof(double) * d * nz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(a11.data(), c->A11.p, sizeof(double) * nz, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// Sp = A11 - A10 diag(A00)^-1 A01 on the owned vertices (rank-local, like the
// reference's per-rank ASM blocks)
static int build_schur_host(cfdh_ctx *c, CsrHost &S) {
  std::vector<double> a00, a01, a10, a11;
  CHK(cfdh_download_blocks(c, a00, a01, a10, a11));
  const int nvo = c->nvo;
  const std::vector<int> &vp = c->h_vptr, &vc = c->h_vcol;
  std::vector<double> dinv(2 * (size_t)nvo);
  for (int v = 0; v < nvo; v++) {
    const size_t k = (size_t)c->h_vdiag[v];
    dinv[2 * v] = 1.0 / a00[4 * k]; dinv[2 * v + 1] = 1.0 / a00[4 * k + 3];
  }
  S.n = S.m = nvo;
  std::vector<std::vector<int>> cols(nvo);
  std::vector<std::vector<double>> vals(nvo);
#pragma omp parallel num_threads(cfdh_host_threads())
  {
    std::vector<int> mark(nvo, -1), list;
    std::vector<double> acc(nvo, 0.0);
#pragma omp for schedule(dynamic, 512)
    for (int i = 0; i < nvo; i++) {
      list.clear();
      for (int k = vp[i]; k < vp[i + 1]; k++) {
        const int w = vc[k];
        if (w >= nvo) continue;
        if (mark[w] != i) { mark[w] = i; acc[w] = 0.0; list.push_back(w); }
        acc[w] += a11[k];
        const double s0 = a10[2 * (size_t)k] * dinv[2 * w], s1 = a10[2 * (size_t)k + 1] * dinv[2 * w + 1];
        if (s0 == 0.0 && s1 == 0.0) continue;
        for (int k2 = vp[w]; k2 < vp[w + 1]; k2++) {
          const int j = vc[k2];
          if (j >= nvo) continue;
          if (mark[j] != i) { mark[j] = i; acc[j] = 0.0; list.push_back(j); }
          acc[j] -= s0 * a01[2 * (size_t)k2] + s1 * a01[2 * (size_t)k2 + 1];
        }
      }
      std::sort(list.begin(), list.end());
      cols[i] = list;
      vals[i].resize(list.size());
      for (size_t t = 0; t < list.size(); t++) vals[i][t] = acc[list[t]];
    }
  }
  S.rowptr.assign(nvo + 1, 0);
  for (int i = 0; i < nvo; i++) S.rowptr[i + 1] = S.rowptr[i] + (int)cols[i].size();
  S.col.resize(S.rowptr[nvo]); S.val.resize(S.rowptr[nvo]);
  for (int i = 0; i < nvo; i++) {
    std::copy(cols[i].begin(), cols[i].end(), S.col.begin() + S.rowptr[i]);
    std::copy(vals[i].begin(), vals[i].end(), S.val.begin() + S.rowptr[i]);
  }
  return 0;
}

// Do-nothing boundary (ds_terms off): the nodes of every exterior facet that is not a no-slip / inflow facet (all velocity
// components of all its nodes constrained) form the outflow boundary of the preconditioner's pressure Poisson problem: bit 1.
static void mark_outflow_nodes(const cfdh_ctx *c, std::vector<unsigned char> &pbc) {
  const unsigned umask = (1u << c->dim) - 1u;
  const int nvo = c->nvo;
  for (int k = 0; k < c->nfac; k++) {
    const int e = c->fac_cell[k], fl = c->fac_local[k];
    int loc[8], nn = 0;
    if (c->gen) nn = c->dim == 3 ? cfdh_facet_nodes3(c, fl, loc) : cfdh_facet_nodes(c, fl, loc);
    else for (int q = 0; q <= c->dim; q++) if (q != fl) loc[nn++] = q;
    bool fixed = true;
    for (int q = 0; q < nn; q++) fixed = fixed && (c->h_bcflag[c->h_cells[(size_t)c->nloc * e + loc[q]]] & umask) == umask;
    if (fixed) continue;
    for (int q = 0; q < nn; q++) {
      const int v = c->h_cells[(size_t)c->nloc * e + loc[q]];
      if (v < nvo) pbc[v] |= 2;
    }
  }
}

// pc_type 1 -- host side of the Cahouet-Chabard-type preconditioner (all rank-local, owned x owned):
//   * hA : SA hierarchy of the scalar proxy (A00_xx + A00_yy)/2 of the velocity block, applied to both
//          components at once (the xy coupling of the symmetric-gradient term is dropped: Korn-equivalent);
//   * hL : SA hierarchy of the P1 pressure Laplacian (Dirichlet rows where p is prescribed) -- geometry
//          and Dirichlet set only, so it is built once;
//   * H  = (I + a'T) M_l + b' A11 with a' = 2 rho/dt, b' = mu, T = diag(A11)/diag(L) (nodal tau/rho):
//          S^-1 ~ (a' L^-1 + b' M_l^-1) [ (I + a'T) + b' A11 M_l^-1 ]^-1 for
//          S = A11 + (1/2) B A00^-1 B^T ~ A11 + (a' L^-1 + b' M_l^-1)^-1   (A11 = (tau/rho)-weighted Laplacian).
static int upload_csr_plain(cfdh_ctx *c, const CsrHost &H, CsrDev &D);

// Hierarchy of a level-0 operator assembled on the host (partitioned runs: ghost rows, replicated pressure space): the sparse
// products, aggregation and formats still run on the device (cfdh_amg_dev.hip) -- the operator is uploaded once.  Falls back to
// the host build when the device build is switched off or gives up.
static int amg_setup_from_host(cfdh_ctx *c, AmgHier &H, const CsrHost &A, bool singular, int ncol) {
  static const bool host_multi = getenv("CFDH_AMG_HOST_MULTI") && getenv("CFDH_AMG_HOST_MULTI")[0] == '1';
  if (cfdh_amg_dev_enabled(c) && !host_multi) {
    CsrDev Ad;
    if (upload_csr_plain(c, A, Ad) == 0 && cfdh_amg_setup_dev(c, H, Ad, singular, ncol) == 0) return 0;
    if (c->opt.verbose) fprintf(stderr, "[cfdh] device-side AMG set-up gave up (%s): building this hierarchy on the host\n", c->err.c_str());
    c->err.clear();
  }
  return cfdh_amg_setup(c, H, A, singular, ncol);
}

// Replicated pressure space of a partitioned run (cfdh_set_global_pressure_space): geometry only, built once per Dirichlet set.
static int build_global_pressure(cfdh_ctx *c) {
  const int nvo = c->nvo;
  if (c->gp_n > 0 && (c->gp_dirty || !c->hLg.valid)) {
    const bool dist = c->nranks > 1 && (int)c->h_gid.size() == c->nv && c->ng > 0;
    c->hLg.keep_host0 = dist;
    CHK(amg_setup_from_host(c, c->hLg, c->gp_L, c->gp_singular, 1));
    c->gp_dirty = false;
    // Distributed finest level: this rank's rows of the global level-0 operator / prolongator (the hierarchy is
    // geometry-only and identical on all ranks, so no exchange is needed to build them).  The cycle is then the
    // SAME arithmetic as the replicated one -- Jacobi sweeps are row-local once the ghost values are there.
    cfdh_ctx::DistL0 &d = c->dl0;
    d.on = false;
    if (dist && c->hLg.lev.size() >= 2) {
      const CsrHost &A0 = c->hLg.h_A0, &P0 = c->hLg.h_P0;
      const std::vector<double> &w0 = c->hLg.h_wdinv0;
      const int nv = c->nv, n1 = c->hLg.lev[1]->n;
      bool ok = (int)w0.size() == c->gp_n && A0.n == c->gp_n && P0.n == c->gp_n && P0.m == n1;
      CsrHost Al, Pl, Pt;
      Al.n = nvo; Al.m = nv; Al.rowptr.assign(nvo + 1, 0);
      for (int i = 0; i < nvo && ok; i++) {
        const int g = c->h_gid[i];
        for (int k = A0.rowptr[g]; k < A0.rowptr[g + 1]; k++) {
          const int loc = c->h_g2l[A0.col[k]];
          if (loc < 0) { ok = false; break; }  // a neighbour of an owned vertex is always local (one-cell overlap)
          Al.col.push_back(loc); Al.val.push_back(A0.val[k]);
        }
        Al.rowptr[i + 1] = (int)Al.col.size();
      }
      Pl.n = nv; Pl.m = n1; Pl.rowptr.assign(nv + 1, 0);
      std::vector<int> cnt(n1 + 1, 0);
      for (int i = 0; i < nv && ok; i++) {
        const int g = c->h_gid[i];
        for (int k = P0.rowptr[g]; k < P0.rowptr[g + 1]; k++) {
          Pl.col.push_back(P0.col[k]); Pl.val.push_back(P0.val[k]);
          if (i < nvo) cnt[P0.col[k] + 1]++;
        }
        Pl.rowptr[i + 1] = (int)Pl.col.size();
      }
      if (ok) {
        Pt.n = n1; Pt.m = nvo; Pt.rowptr.assign(n1 + 1, 0);
        for (int I = 0; I < n1; I++) Pt.rowptr[I + 1] = Pt.rowptr[I] + cnt[I + 1];
        Pt.col.resize(Pt.rowptr[n1]); Pt.val.resize(Pt.rowptr[n1]);
        std::vector<int> fill(Pt.rowptr.begin(), Pt.rowptr.end() - 1);
        for (int i = 0; i < nvo; i++)  // ascending owned index within every coarse row: fixed summation order
          for (int k = Pl.rowptr[i]; k < Pl.rowptr[i + 1]; k++) { const int q = fill[Pl.col[k]]++; Pt.col[q] = i; Pt.val[q] = Pl.val[k]; }
        std::vector<double> wl(nv);
        for (int i = 0; i < nv; i++) wl[i] = w0[c->h_gid[i]];
        CHK(cfdh_upload_csr(c, Al, d.A, &wl)); CHK(cfdh_upload_csr(c, Pl, d.P)); CHK(cfdh_upload_csr(c, Pt, d.PT));
        HIPCHK(c, d.wdinv.upload(wl, c->stream));
        HIPCHK(c, d.b.alloc(nv)); HIPCHK(c, d.xa.alloc(nv)); HIPCHK(c, d.r.alloc(nvo)); HIPCHK(c, d.x1.alloc(nv));
        HIPCHK(c, d.b.zero(c->stream)); HIPCHK(c, d.xa.zero(c->stream));  // their ghost parts stay zero when the rhs's ghost layer is not exchanged
        HIPCHK(c, hipStreamSynchronize(c->stream));
        d.n1 = n1;
        d.on = true;
        // Round 4: the ghost layer of the cycle's right-hand side is NOT exchanged any more -- the pre-smoothed iterate W b is taken
        // as zero on the ghost vertices.  That perturbs the pre-smoothing on the interface rows only; measured with 4 ranks
        // (gpurun_out/r4_d.log, r4_e.log): dfg_1 22.6 / 22.6 iterations per step with / without the exchange, stenosis 87.5 / 87.5,
        // tetrahedra 70.2 / 70.5, cavity 26.7 / 26.7 -- one halo exchange per FGMRES iteration less.  CFDH_DL0_GHOST_RHS=1 restores it.
        { const char *e = getenv("CFDH_DL0_GHOST_RHS"); d.ghost_rhs = e && e[0] == '1'; }
      }
      double bad = d.on ? 0.0 : 1.0;  // all ranks or none
      HIPCHK(c, hipMemcpyAsync(c->red_out.p + 21, &bad, sizeof(double), hipMemcpyHostToDevice, c->stream));
      CHK(comm_allreduce_dev(c, c->red_out.p + 21, 1, 1));
      HIPCHK(c, hipMemcpyAsync(&bad, c->red_out.p + 21, sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      d.on = bad == 0.0;
      c->hLg.h_A0 = CsrHost(); c->hLg.h_P0 = CsrHost(); c->hLg.h_wdinv0.clear();
    }
  }
  return 0;
}

static int build_cc_host(cfdh_ctx *c) {
  std::vector<double> a00, a01, a10, a11;
  CHK(cfdh_download_blocks(c, a00, a01, a10, a11));
  const int nvo = c->nvo;
  const std::vector<int> &vp = c->h_vptr, &vc = c->h_vcol;
  // --- scalar proxy of A00
  // Partitioned run with the upper-triangular factor: restricted additive Schwarz with one layer of overlap.  The
  // local operator is the principal submatrix on owned + ghost vertices; the ghost rows are fetched from their
  // owners (entry k of every owned row travels in round k of the ordinary halo exchange, tagged with the global
  // column id).  With exact subdomain solves this brings the iteration count of 4-8 strips back to the
  // single-domain one; without overlap (block Jacobi) it grows by half.
  const bool ras = c->nranks > 1 && c->gp_n > 0 && c->opt.schur_full == 2 && (int)c->h_gid.size() == c->nv && c->ng > 0;
  {
    const int nloc = ras ? c->nv : nvo;
    CsrHost Ah;
    Ah.n = Ah.m = nloc;
    Ah.rowptr.assign(nloc + 1, 0);
    int maxlen = 0;
    for (int i = 0; i < nvo; i++) {
      for (int k = vp[i]; k < vp[i + 1]; k++) {
        const int w = vc[k];
        if (w >= nloc) continue;
        // mean of the diagonal entries of the dim x dim block: (A00_xx + A00_yy [+ A00_zz]) / dim
        double v = 0.0;
        for (int q = 0; q < c->dim; q++) v += a00[(size_t)c->dim * c->dim * k + (size_t)q * (c->dim + 1)];
        v /= c->dim;
        if (v == 0.0 && w != i) continue;
        Ah.col.push_back(w); Ah.val.push_back(v);
      }
      Ah.rowptr[i + 1] = (int)Ah.col.size();
      maxlen = std::max(maxlen, Ah.rowptr[i + 1] - Ah.rowptr[i]);
    }
    if (ras) {
      double ml = (double)maxlen;  // rounds = longest owned row over all ranks
      HIPCHK(c, hipMemcpyAsync(c->red_out.p + 20, &ml, sizeof(double), hipMemcpyHostToDevice, c->stream));
      CHK(comm_allreduce_dev(c, c->red_out.p + 20, 1, 1));
      HIPCHK(c, hipMemcpyAsync(&ml, c->red_out.p + 20, sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const int rounds = (int)ml;
      if (!c->pcw.p) { HIPCHK(c, c->pcw.alloc(c->NL)); }
      std::vector<std::vector<std::pair<int, double>>> grow(c->ng);
      std::vector<double> hv((size_t)c->NL);
      for (int k = 0; k < rounds; k++) {
        std::fill(hv.begin(), hv.end(), -1.0);
        for (int i = 0; i < nvo; i++) {
          const int p = Ah.rowptr[i] + k;
          // the pair (global column id, value) rides in the first two velocity slots of the vertex record
          if (p < Ah.rowptr[i + 1]) { hv[(size_t)c->dim * i] = (double)c->h_gid[Ah.col[p]]; hv[(size_t)c->dim * i + 1] = Ah.val[p]; }
        }
        HIPCHK(c, c->pcw.upload(hv, c->stream));
        CHK(comm_halo(c, c->pcw.p));
        const size_t W = (size_t)c->dim + 1;  // doubles per vertex record
        HIPCHK(c, hipMemcpyAsync(hv.data() + W * (size_t)nvo, c->pcw.p + W * (size_t)nvo, sizeof(double) * W * (size_t)c->ng,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int g = 0; g < c->ng; g++) {
          const double gd = hv[W * (size_t)nvo + W * (size_t)g];
          if (!(gd >= 0)) continue;
          const int loc = c->h_g2l[(int)gd];
          if (loc >= 0) grow[g].push_back({loc, hv[W * (size_t)nvo + W * (size_t)g + 1]});
        }
      }
      HIPCHK(c, c->pcw.zero(c->stream));
      for (int g = 0; g < c->ng; g++) {
        auto &r = grow[g];
        std::sort(r.begin(), r.end());
        bool diag = false;
        for (auto &e : r) { Ah.col.push_back(e.first); Ah.val.push_back(e.second); diag |= e.first == nvo + g; }
        if (!diag) return cfdh_fail(c, CFDH_E_COMM, "ghost row %d arrived without its diagonal", g);
        Ah.rowptr[nvo + g + 1] = (int)Ah.col.size();
      }
      if (!c->ras_b.p) { HIPCHK(c, c->ras_b.alloc((size_t)c->dim * c->nv)); HIPCHK(c, c->ras_x.alloc((size_t)c->dim * c->nv)); }
    }
    c->ras = ras;
    CHK(amg_setup_from_host(c, c->hA, Ah, false, c->dim));
  }
  // --- pressure Laplacian hierarchy (once per Dirichlet set)
  // pbc bit0: the pressure dof is Dirichlet (identity row in H, z_p = r_p); bit1: Dirichlet in L only --
  // with a do-nothing boundary (ds_terms off) the vertices of every exterior facet that is not a
  // no-slip/inflow facet form the outflow boundary of the pressure Poisson problem.
  std::vector<unsigned char> pbc(nvo);
  const unsigned pbit = 1u << c->dim;  // bits 0..dim-1: velocity components, bit dim: pressure
  for (int i = 0; i < nvo; i++) pbc[i] = (c->h_bcflag[i] & pbit) ? 1 : 0;
  if (!c->ds_terms) mark_outflow_nodes(c, pbc);
  if (!c->hL.valid || c->hL_pbc != pbc || c->hL_singular != c->singular) {
    CsrHost Lh;
    Lh.n = Lh.m = nvo;
    Lh.rowptr.assign(nvo + 1, 0);
    for (int i = 0; i < nvo; i++) {
      if (pbc[i]) { Lh.col.push_back(i); Lh.val.push_back(1.0); }
      else
        for (int k = vp[i]; k < vp[i + 1]; k++) {
          const int w = vc[k];
          if (w >= nvo || pbc[w]) continue;
          Lh.col.push_back(w); Lh.val.push_back(c->h_Lval[k]);
        }
      Lh.rowptr[i + 1] = (int)Lh.col.size();
    }
    bool any_pbc = false;
    for (int i = 0; i < nvo; i++) any_pbc |= pbc[i] != 0;
    // a part without pressure-Dirichlet rows has a pure-Neumann (singular) local Laplacian
    CHK(amg_setup_from_host(c, c->hL, Lh, c->singular != 0 || !any_pbc, 1));
    c->hL_pbc = pbc;
    c->hL_singular = c->singular;
    std::vector<double> ml(nvo);
    for (int i = 0; i < nvo; i++) ml[i] = pbc[i] ? 0.0 : c->h_Ml[i];
    HIPCHK(c, c->ccMl.upload(ml, c->stream));
    HIPCHK(c, c->ccPbc.upload(pbc, c->stream));
  }
  CHK(build_global_pressure(c));
  // --- H
  c->cc_alpha = c->rho * c->ts_a[0] / (c->ts_theta * c->dt);  // = 2 rho/dt for the midpoint scheme
  c->cc_beta = c->mu;
  {
    CsrHost Hh;
    Hh.n = Hh.m = nvo;
    Hh.rowptr.assign(nvo + 1, 0);
    for (int i = 0; i < nvo; i++) {
      if (pbc[i] & 1) { Hh.col.push_back(i); Hh.val.push_back(1.0); }
      else {
        const size_t kd = (size_t)c->h_vdiag[i];
        const double T = c->h_Lval[kd] > 0 ? a11[kd] / c->h_Lval[kd] : 0.0;
        for (int k = vp[i]; k < vp[i + 1]; k++) {
          const int w = vc[k];
          if (w >= nvo || (pbc[w] & 1)) continue;
          double v = c->cc_beta * a11[k];
          if (w == i) v += (1.0 + c->cc_alpha * T) * c->h_Ml[i];
          Hh.col.push_back(w); Hh.val.push_back(v);
        }
      }
      Hh.rowptr[i + 1] = (int)Hh.col.size();
    }
    CsrDev Hd;
    if (cfdh_amg_dev_enabled(c) && upload_csr_plain(c, Hh, Hd) == 0) CHK(cfdh_level_setup_dev(c, c->Hlev, Hd, 8.0, 1));
    else CHK(cfdh_level_setup(c, c->Hlev, Hh, 8.0, 1));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// The same preconditioner data built where the Jacobian lives (cfdh_amg_dev.hip): no download of the blocks, sparse
// products / aggregation / formats by kernels.
static int upload_csr_plain(cfdh_ctx *c, const CsrHost &H, CsrDev &D) {
  D.n = H.n; D.m = H.m; D.nnz = H.nnz();
  HIPCHK(c, D.rowptr.upload(H.rowptr, c->stream));
  HIPCHK(c, D.col.upload(H.col, c->stream));
  HIPCHK(c, D.val.upload(H.val, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
static int download_csr_plain(cfdh_ctx *c, const CsrDev &D, CsrHost &H) {
  H.n = D.n; H.m = D.m;
  H.rowptr.resize((size_t)D.n + 1); H.col.resize(D.nnz); H.val.resize(D.nnz);
  HIPCHK(c, hipMemcpyAsync(H.rowptr.data(), D.rowptr.p, sizeof(int) * ((size_t)D.n + 1), hipMemcpyDeviceToHost, c->stream));
  if (D.nnz) {
    HIPCHK(c, hipMemcpyAsync(H.col.data(), D.col.p, sizeof(int) * (size_t)D.nnz, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(H.val.data(), D.val.p, sizeof(double) * (size_t)D.nnz, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
// Round 4: partitioned runs build here too.  The ghost rows of the overlapping velocity proxy are fetched from their owners by
// halo exchanges of device buffers (cfdh_proxy_ras_dev), H drops the ghost columns in its kernel, and nothing of the Jacobian is
// downloaded.  What is collective (the ghost-row exchange) succeeds or fails on all ranks alike; the hierarchy set-up behind it is
// rank-local, so a rank whose device set-up gives up finishes on the host from a copy of its operator without the others noticing.
static int build_cc_dev(cfdh_ctx *c) {
  const int nvo = c->nvo;
  const std::vector<int> &vp = c->h_vptr, &vc = c->h_vcol;
  const double t0 = wall_ms();
  const bool multi = c->nranks > 1;
  const bool ras = multi && c->gp_n > 0 && c->opt.schur_full == 2 && (int)c->h_gid.size() == c->nv && c->ng > 0;
  c->ras = false;
  {
    CsrDev Ah, keep;
    if (ras) {
      CHK(cfdh_proxy_ras_dev(c, Ah));
      if (!c->ras_b.p) { HIPCHK(c, c->ras_b.alloc((size_t)c->dim * c->nv)); HIPCHK(c, c->ras_x.alloc((size_t)c->dim * c->nv)); }
    } else CHK(cfdh_proxy_dev(c, Ah));
    c->ras = ras;
    if (multi) {  // the set-up consumes its operator
      keep.n = Ah.n; keep.m = Ah.m; keep.nnz = Ah.nnz;
      HIPCHK(c, keep.rowptr.alloc((size_t)Ah.n + 1)); HIPCHK(c, keep.col.alloc((size_t)std::max(Ah.nnz, 1))); HIPCHK(c, keep.val.alloc((size_t)std::max(Ah.nnz, 1)));
      HIPCHK(c, hipMemcpyAsync(keep.rowptr.p, Ah.rowptr.p, sizeof(int) * ((size_t)Ah.n + 1), hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(keep.col.p, Ah.col.p, sizeof(int) * (size_t)Ah.nnz, hipMemcpyDeviceToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(keep.val.p, Ah.val.p, sizeof(double) * (size_t)Ah.nnz, hipMemcpyDeviceToDevice, c->stream));
    }
    const int rc = cfdh_amg_setup_dev(c, c->hA, Ah, false, c->dim);
    if (rc != 0) {
      if (!multi) return rc;
      if (c->opt.verbose) fprintf(stderr, "[cfdh] device-side AMG set-up gave up (%s): building the velocity hierarchy on the host\n", c->err.c_str());
      c->err.clear(); c->hA.clear();
      CsrHost Ahh;
      CHK(download_csr_plain(c, keep, Ahh));
      CHK(cfdh_amg_setup(c, c->hA, Ahh, false, c->dim));
    }
  }
  const double t1 = wall_ms();
  // pressure Laplacian hierarchy: geometry and Dirichlet set only (see build_cc_host for the pbc bits)
  std::vector<unsigned char> pbc(nvo);
  const unsigned pbit = 1u << c->dim;
  for (int i = 0; i < nvo; i++) pbc[i] = (c->h_bcflag[i] & pbit) ? 1 : 0;
  if (!c->ds_terms) mark_outflow_nodes(c, pbc);
  if (!c->hL.valid || c->hL_pbc != pbc || c->hL_singular != c->singular) {
    CsrHost Lh;
    Lh.n = Lh.m = nvo;
    Lh.rowptr.assign(nvo + 1, 0);
    Lh.col.reserve(c->nnzv); Lh.val.reserve(c->nnzv);
    bool any_pbc = false;
    for (int i = 0; i < nvo; i++) {
      any_pbc |= pbc[i] != 0;
      if (pbc[i]) { Lh.col.push_back(i); Lh.val.push_back(1.0); }
      else
        for (int k = vp[i]; k < vp[i + 1]; k++) {
          const int w = vc[k];
          if (w >= nvo || pbc[w]) continue;
          Lh.col.push_back(w); Lh.val.push_back(c->h_Lval[k]);
        }
      Lh.rowptr[i + 1] = (int)Lh.col.size();
    }
    if (multi) CHK(amg_setup_from_host(c, c->hL, Lh, c->singular != 0 || !any_pbc, 1));
    else {
      CsrDev Ld;
      CHK(upload_csr_plain(c, Lh, Ld));
      CHK(cfdh_amg_setup_dev(c, c->hL, Ld, c->singular != 0 || !any_pbc, 1));
    }
    c->hL_pbc = pbc;
    c->hL_singular = c->singular;
    std::vector<double> ml(nvo);
    for (int i = 0; i < nvo; i++) ml[i] = pbc[i] ? 0.0 : c->h_Ml[i];
    HIPCHK(c, c->ccMl.upload(ml, c->stream));
    HIPCHK(c, c->ccPbc.upload(pbc, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (multi) CHK(build_global_pressure(c));
  const double t2 = wall_ms();
  c->cc_alpha = c->rho * c->ts_a[0] / (c->ts_theta * c->dt);
  c->cc_beta = c->mu;
  {
    CsrDev Hd;
    CHK(cfdh_cc_h_dev(c, c->cc_alpha, c->cc_beta, Hd));
    CHK(cfdh_level_setup_dev(c, c->Hlev, Hd, 8.0, 1));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->ms_pc_build_dev = wall_ms() - t0;
  if (c->opt.verbose)
    fprintf(stderr, "[cfdh] preconditioner built on the device: velocity hierarchy %.1f ms, pressure Laplacian %.1f ms, H %.1f ms\n", t1 - t0, t2 - t1,
            wall_ms() - t2);
  return 0;
}
static int build_cc(cfdh_ctx *c) {
  // CFDH_PC_HOST_ASSEMBLY=1: level-0 operators of a partitioned run assembled on the host as in rounds 2-3 (comparison, tests)
  static const bool host_multi = (getenv("CFDH_AMG_HOST_MULTI") && getenv("CFDH_AMG_HOST_MULTI")[0] == '1') ||
                                 (getenv("CFDH_PC_HOST_ASSEMBLY") && getenv("CFDH_PC_HOST_ASSEMBLY")[0] == '1');
  if (c->nranks > 1 && cfdh_amg_dev_enabled(c) && !host_multi) return build_cc_dev(c);  // no per-rank fall-back across a collective
  if (c->nranks == 1 && cfdh_amg_dev_enabled(c)) {
    const int rc = build_cc_dev(c);
    if (rc == 0) return 0;
    // the device build gave up (a product row beyond its tables, a zero pivot): keep the message, build on the host
    if (c->opt.verbose) fprintf(stderr, "[cfdh] device-side hierarchy set-up failed (%s): host build\n", c->err.c_str());
    c->hL.clear(); c->hA.clear();
  }
  return build_cc_host(c);
}

// refresh the parts of the preconditioner that follow the current Jacobian:
// always the Jacobi diagonal and the spectral bound of D^-1 A00; the Sp
// hierarchy only when asked (lagged preconditioner)
int cfdh_pc_update(cfdh_ctx *c, bool refresh_amg) {
  const int nu = c->dim * c->nvo;
  if (c->dim == 3 && c->opt.pc_type != 1) return cfdh_fail(c, CFDH_E_ARG, "tetrahedral contexts support pc_type 1 only");
  if (c->opt.pc_type == 1) {
    if (refresh_amg || !c->pc_valid) {
      c->pc_graph_valid = false;
      CHK(build_cc(c));
      c->pc_valid = true;
      c->pc_its_ref = 0;
      c->steps_since_refresh = 0;
      c->last_stats.pc_refreshes++;
    }
    return 0;
  }
  CHK(k_extract_diag(c));
  // lambda_max(D^-1 A00) by power iteration from a fixed start vector
  double *v = c->pu0.p, *w = c->pu1.p;
  CHK(v_copy(c, nu, c->prand.p, v));
  double lam = 1.0;
  for (int it = 0; it < 8; it++) {
    CHK(k_spmv_block(c, 1, v, w, nullptr, 0));
    CHK(v_pointwise_mult(c, nu, w, c->dinvA.p, w));
    CHK(v_norm_to_dev(c, nu, w, c->red_out.p + 4));
    CHK(v_scale_inv_dev(c, nu, w, c->red_out.p + 4, v));
  }
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->red_out.p + 4, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  lam = c->h_pinned[0];
  if (!(lam > 0) || !std::isfinite(lam)) return cfdh_fail(c, CFDH_E_DIVERGED, "non-finite spectral estimate of D^-1 A00 (NaN in the Jacobian?)");
  c->lmaxA = 1.15 * lam;
  CHK(k_cheb_a00_coeffs(c));
  if (refresh_amg || !c->pc_valid) {
    c->pc_graph_valid = false;  // the hierarchies' buffers and coefficients are baked into the graphs
    CsrHost S;
    CHK(build_schur_host(c, S));
    bool any_pbc = false;
    for (int i = 0; i < c->nvo; i++) any_pbc |= (c->h_bcflag[i] & (1u << c->dim)) != 0;
    CHK(cfdh_amg_setup(c, c->hS, S, c->singular != 0 || !any_pbc, 1));
    c->pc_valid = true;
    c->pc_its_ref = 0;
    c->steps_since_refresh = 0;
    c->last_stats.pc_refreshes++;
  }
  return 0;
}

static bool ras_ghost_rhs() {
  static const bool on = !(getenv("CFDH_RAS_GHOST_RHS") && getenv("CFDH_RAS_GHOST_RHS")[0] == '0');
  return on;
}

// The application is a sequence of stages separated by the exchanges of a partitioned run:
//   stage 0: y_u = V(A00~) r_u                                   | halo(y_u)
//   stage 1: t_p = r_p - A10 y_u ; zH = Cheb3(H) t_p ; y = M_l zH | all-reduce of the global pressure rhs
//   stage 2: t = V(L) y ; z_p = a' t + b' zH                      | halo(z_p)
//   stage 3: z_u = V(A00~)(r_u - A01 z_p)
// With one rank (no exchanges) all stages run back to back; each stage is graph-capturable.
static int pc_stage(cfdh_ctx *c, const double *r, double *z, int stage) {
  const int nvo = c->nvo, nu = c->dim * nvo;
  const double *ru = r, *rp = r + nu;
  double *zu = z, *zp = z + nu;
  const bool multi = c->nranks > 1;
  const bool global_p = c->gp_n > 0 && multi;
  if (c->opt.pc_type == 1) {
    const bool upper = c->opt.schur_full == 2;  // block upper-triangular: z_p = S^-1 r_p, z_u = A^-1 (r_u - A01 z_p)
    // logical stages: 0 first velocity cycle, 1 H solve, 2 pressure cycle, 3 coupling product, 4 overlapping velocity
    // cycle; a partitioned run with the distributed pressure level inserts the down sweep (10) after the H solve
    const bool dist = global_p && c->dl0.on;
    int ls = stage;
    if (dist) ls = stage <= 1 ? stage : (stage == 2 ? 10 : stage - 1);
    switch (ls) {
      case 0:
        if (upper) return 0;
        // in a partitioned run y_u lands in the halo scratch vector so that its ghosts can be refreshed
        CHK(k_amg_vcycle(c, c->hA, ru, multi ? c->pcw.p : c->pu0.p));
        return 0;
      case 1:
        if (upper) { /* t_p = r_p: used in place */ }
        else if (multi) CHK(k_spmv_block_ghost(c, 3, c->pcw.p, c->pp0.p, rp));   // t_p = r_p - A10 y_u (with ghosts)
        else CHK(k_spmv_block(c, 3, c->pu0.p, c->pp0.p, rp, 0));
        {
          // right-hand side of the pressure cycle.  Distributed finest level: straight into the level's own vector (ghost part zero);
          // with the ghost-layer exchange switched on (CFDH_DL0_GHOST_RHS=1) into the pressure slot of the halo scratch vector
          double *y = !dist ? c->pu1.p : (c->dl0.ghost_rhs ? c->pcw.p + nu : c->dl0.b.p);
          if (!(c->opt.cc_smooth_degree == 2 && k_cc_cheb2_scale(c, &c->Hlev, upper ? rp : c->pp0.p, c->pp1.p, c->ccMl.p, y))) {
            CHK(k_level_smooth(c, &c->Hlev, upper ? rp : c->pp0.p, c->pp1.p, c->opt.cc_smooth_degree));
            CHK(k_cc_scale(c, nvo, c->ccMl.p, c->pp1.p, y));
          }
        }
        if (dist) {
        } else if (global_p && c->gp_allgather) {
          CHK(k_gather_global(c, nvo, c->gp_send_idx.p, c->pu1.p, c->gp_sendbuf.p));  // owned values in global-id order
        } else if (global_p) {
          CHK(v_zero(c, c->gp_n, c->gp_rhs.p));
          CHK(k_scatter_global(c, nvo, c->gp_l2g.p, c->pu1.p, c->gp_rhs.p));
        }
        return 0;
      case 10:
        return k_dl0_down(c, c->pcw.p);  // pre-smoothing of the owned rows, owned part of the coarse right-hand side
      case 2:
        if (dist) {
          CHK(k_dl0_up(c, c->pu2.p));    // replicated coarse cycle, prolongation, post-smoothing of the owned rows
        } else if (global_p) {
          if (c->gp_allgather) CHK(k_gather_global(c, c->gp_n, c->gp_src_idx.p, c->gp_recvbuf.p, c->gp_rhs.p));
          CHK(k_amg_vcycle(c, c->hLg, c->gp_rhs.p, c->gp_sol.p));  // the same global V-cycle on every rank
          CHK(k_gather_global(c, nvo, c->gp_l2g.p, c->gp_sol.p, c->pu2.p));
        } else {
          // rank-local cycle: the combination below runs in the epilogue of its last kernel when the fused cycle is used
          static const int lcycles = getenv("CFDH_L_CYCLES") ? atoi(getenv("CFDH_L_CYCLES")) : 1;
          if (lcycles > 1 && c->hL.lev.size() >= 1 && c->hL.lev[0]->A.val.p) {
            // experiment: k V-cycles on the pressure Laplacian (stationary iteration x += V(b - L x))
            AmgLevel *L0 = c->hL.lev[0];
            CHK(k_amg_vcycle(c, c->hL, c->pu1.p, c->pu2.p));
            for (int cyc = 1; cyc < lcycles; cyc++) {
              CHK(k_csr_spmv(c, L0->A, c->pu2.p, L0->r.p, 1, c->pu1.p));   // r = y - L t
              CHK(k_amg_vcycle(c, c->hL, L0->r.p, L0->d0.p));
              CHK(v_axpy(c, nvo, 1.0, L0->d0.p, c->pu2.p));
            }
            CHK(k_cc_combine(c, nvo, c->cc_alpha, c->cc_beta, c->pu2.p, c->pp1.p, upper ? rp : c->pp0.p, c->ccPbc.p, zp));
            if (multi && c->opt.schur_full) CHK(v_copy(c, nvo, zp, c->pcw.p + nu));
            return 0;
          }
          c->epi.on = true; c->epi.done = false;
          c->epi.alpha = c->cc_alpha; c->epi.beta = c->cc_beta; c->epi.zH = c->pp1.p; c->epi.r = upper ? rp : c->pp0.p;
          c->epi.pbc = c->ccPbc.p; c->epi.out = zp;
          const int rc = k_amg_vcycle(c, c->hL, c->pu1.p, c->pu2.p);
          c->epi.on = false;
          CHK(rc);
        }
        if (c->epi.done) {
          c->epi.done = false;
          if (multi && c->opt.schur_full) CHK(v_copy(c, nvo, zp, c->pcw.p + nu));  // z_p into the halo scratch vector
        } else {
          CHK(k_cc_combine(c, nvo, c->cc_alpha, c->cc_beta, c->pu2.p, c->pp1.p, upper ? rp : c->pp0.p, c->ccPbc.p, zp,
                           (multi && c->opt.schur_full) ? c->pcw.p + nu : nullptr));
        }
        return 0;
      case 3:
        if (c->opt.schur_full) {
          // t_u = r_u - A01 z_p (with ghosts); for the overlapping cycle straight into the velocity slots of the halo scratch vector
          // (the kernel reads the pressure slots and ghost records of that vector and writes its owned velocity slots: disjoint)
          if (multi) CHK(k_spmv_block_ghost(c, 2, c->pcw.p, c->ras ? c->pcw.p : c->pu0.p, ru));
          else CHK(k_spmv_block(c, 2, zp, c->pu0.p, ru, 0));
          if (multi && c->ras) {
            // experiment (CFDH_RAS_GHOST_RHS=0): no exchange of the overlap residual -- the right-hand side of the overlapping
            // cycle is zero on the ghost layer
            if (!ras_ghost_rhs()) CHK(v_zero(c, c->NL - c->NO, c->pcw.p + c->NO));
          }
          else {
            CHK(k_amg_vcycle(c, c->hA, c->pu0.p, zu));
            static const int acycles = getenv("CFDH_A_CYCLES") ? atoi(getenv("CFDH_A_CYCLES")) : 1;
            if (acycles > 1 && !multi && c->hA.lev[0]->A.val.p) {
              // experiment: further V-cycles on the velocity proxy (stationary iteration)
              AmgLevel *L0 = c->hA.lev[0];
              for (int cyc = 1; cyc < acycles; cyc++) {
                CHK(k_csr_spmv_ncol(c, L0->A, zu, L0->r.p, 1, c->pu0.p, c->dim));
                CHK(k_amg_vcycle(c, c->hA, L0->r.p, L0->d0.p));
                CHK(v_axpy(c, nu, 1.0, L0->d0.p, zu));
              }
            }
          }
        } else {
          CHK(v_copy(c, nu, multi ? c->pcw.p : c->pu0.p, zu));              // block lower-triangular variant
        }
        return 0;
      case 4:
        if (multi && c->ras && c->opt.schur_full) {
          // restricted additive Schwarz: cycle on owned + ghost vertices, keep the owned part
          CHK(k_ext_pack(c, c->pcw.p, c->ras_b.p));
          if (c->hA.fused && c->opt.amg_smooth_degree == 1 && c->hA.lev.size() >= 2) {
            c->up0_rows = nvo;  // the finest up-sweep writes the owned rows only, straight into z_u
            const int rc = k_amg_vcycle(c, c->hA, c->ras_b.p, zu);
            c->up0_rows = 0;
            CHK(rc);
          } else {
            CHK(k_amg_vcycle(c, c->hA, c->ras_b.p, c->ras_x.p));
            CHK(v_copy(c, nu, c->ras_x.p, zu));
          }
        }
        return 0;
      default:
        return 0;  // stage slot unused by this configuration
    }
  }
  if (stage != 0) return 0;  // pc_type 0 is rank-local: one stage
  CHK(k_cheb_a00(c, ru, c->pu0.p));                       // y_u = C(A00) r_u
  CHK(k_spmv_block(c, 3, c->pu0.p, c->pp0.p, rp, 0));     // t_p = r_p - A10 y_u
  CHK(k_amg_vcycle(c, c->hS, c->pp0.p, zp));              // z_p = V(Sp) t_p
  if (c->opt.schur_full) {
    CHK(k_spmv_block(c, 2, zp, c->pu0.p, ru, 0));         // t_u = r_u - A01 z_p
    CHK(k_cheb_a00(c, c->pu0.p, zu));                     // z_u = C(A00) t_u
  } else {
    CHK(v_copy(c, nu, c->pu0.p, zu));
  }
  return 0;
}

// exchange that follows stage `stage` in a partitioned run
static int pc_exchange(cfdh_ctx *c, int stage) {
  if (c->nranks <= 1 || c->opt.pc_type != 1) return 0;
  const bool dist = c->gp_n > 0 && c->dl0.on;
  int ls = stage;
  if (dist) ls = stage <= 1 ? stage : (stage == 2 ? 10 : stage - 1);
  switch (ls) {
    case 0: return c->opt.schur_full == 2 ? 0 : comm_halo(c, c->pcw.p);
    case 1:
      if (c->gp_n <= 0) return 0;
      if (dist) return c->dl0.ghost_rhs ? comm_halo(c, c->pcw.p) : 0;  // right-hand side of the pressure cycle on the ghost layer
      if (c->gp_allgather) return comm_allgather_dev(c, c->gp_sendbuf.p, c->gp_recvbuf.p, c->gp_maxcnt);
      return comm_allreduce_dev(c, c->gp_rhs.p, c->gp_n, 0);
    case 10: return comm_allreduce_dev(c, c->hLg.lev[1]->b.p, c->dl0.n1, 0);  // coarse right-hand side of the replicated levels
    case 2: return c->opt.schur_full ? comm_halo(c, c->pcw.p) : 0;
    case 3: return (c->ras && c->opt.schur_full && ras_ghost_rhs()) ? comm_halo(c, c->pcw.p) : 0;  // residual of the overlap layer
    default: return 0;
  }
}

// z = P^-1 r.  The kernels of one application have fixed shapes, so every stage is captured into a
// hipGraph and replayed (the Krylov loop is launch-bound otherwise: MI355X guide, "graph-replay-floor").
// Operands differ per Krylov slot (r = V_j, z = Z_j), so graphs are kept per (r, z) pair -- no staging
// copies; all are dropped when a hierarchy is rebuilt.
int cfdh_pc_apply(cfdh_ctx *c, const double *r, double *z) {
  const int nvo = c->nvo;
  const bool graph = c->use_graph && !c->prof_on;
  const bool multi = c->nranks > 1 && c->opt.pc_type == 1;
  if (multi && !c->pcw.p) { HIPCHK(c, c->pcw.alloc(c->NL)); HIPCHK(c, c->pcw.zero(c->stream)); }
  if (!graph) {
    for (int st = 0; st < 6; st++) { CHK(pc_stage(c, r, z, st)); CHK(pc_exchange(c, st)); }
  } else {
    if (!c->pc_graph_valid) {
      for (auto &e : c->pc_graphs) for (auto &x : e.exec) if (x) (void)hipGraphExecDestroy(x);
      c->pc_graphs.clear();
      c->pc_graph_valid = true;
    }
    cfdh_ctx::PcGraph *pg = nullptr;
    for (auto &e : c->pc_graphs) if (e.r == r && e.z == z) { pg = &e; break; }
    if (!pg) {
      cfdh_ctx::PcGraph ng{r, z, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
      // one rank: all four stages in one graph; partitioned: one graph per stage
      for (int gidx = 0; gidx < (multi ? 6 : 1); gidx++) {
        hipGraph_t g = nullptr;
        HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        c->capturing = true;
        int rc = 0;
        if (multi) rc = pc_stage(c, r, z, gidx);
        else for (int st = 0; st < 6 && !rc; st++) rc = pc_stage(c, r, z, st);
        c->capturing = false;
        hipError_t e = hipStreamEndCapture(c->stream, &g);
        if (rc) return rc;
        if (e != hipSuccess || !g) return cfdh_fail(c, CFDH_E_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
        e = hipGraphInstantiate(&ng.exec[gidx], g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) return cfdh_fail(c, CFDH_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
      }
      c->pc_graphs.push_back(ng);
      pg = &c->pc_graphs.back();
    }
    if (!multi) {
      HIPCHK(c, hipGraphLaunch(pg->exec[0], c->stream));
    } else {
      for (int st = 0; st < 6; st++) { HIPCHK(c, hipGraphLaunch(pg->exec[st], c->stream)); CHK(pc_exchange(c, st)); }
    }
  }
  if (c->singular) CHK(v_sub_mean(c, nvo, z + (size_t)c->dim * nvo));
  return 0;
}

static int ensure_krylov(cfdh_ctx *c) {
  const int m = c->opt.ksp_restart;
  if (c->kry_m == m && c->kV.p) return 0;
  const size_t NL = ((size_t)c->NL + 1) & ~(size_t)1;  // even leading dimension: every V_j / Z_j stays 16-B aligned
  c->pc_graph_valid = false;  // captured graphs hold pointers into V / Z
  HIPCHK(c, c->kV.alloc(NL * (m + 1)));
  HIPCHK(c, c->kZ.alloc(NL * m));
  HIPCHK(c, c->kw.alloc(NL));
  // kh also serves guess_project as the scratch of its Gram system: 8 (k + 1) doubles with k <= 8 kept vectors (the cap of
  // cfdh_set_options), whatever the restart length
  HIPCHK(c, c->kh.alloc(std::max(2 * (size_t)(m + 2) + 8, (size_t)8 * (8 + 1))));
  HIPCHK(c, c->ky.alloc(m + 8));
  // read-back ring of the iterations in flight: host-mapped slots of [h_0 .. h_j, w.w, measured norm] and one event each
  if (c->h_ring) { (void)hipHostFree(c->h_ring); c->h_ring = nullptr; }
  c->h_ring_stride = ((size_t)m + 5) & ~(size_t)1;
  HIPCHK(c, hipHostMalloc((void **)&c->h_ring, sizeof(double) * c->h_ring_stride * (cfdh_ctx::KRING + 1)));  // + one slot: staging of y
  HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_ring_dev, c->h_ring, 0));
  for (auto &e : c->ev_ring) if (!e) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  HIPCHK(c, c->kV.zero(c->stream)); HIPCHK(c, c->kZ.zero(c->stream)); HIPCHK(c, c->kw.zero(c->stream));
  if ((size_t)(m + 2) * 1024 > c->red_partial.n) HIPCHK(c, c->red_partial.alloc((size_t)(m + 2) * 1024 + 1024));
  c->kry_m = m;
  return 0;
}

// ---- projected initial guess (cfdh_options.ksp_guess; PETSc's KSPGuess of Fischer type, with the CURRENT matrix) ----------
// The sequence of linear systems of a time-stepping run is smooth in time: the k-th Newton update of step n+1 is close to
// the k-th update of step n.  U = the last few solutions of solves with the same Newton index; W = J U (one SpMV each);
// x0 = U y with y = argmin |b - W y| from the normal equations (at most 8 x 8, solved on the host with a pivoted Cholesky that
// drops directions which have become numerically dependent).  |b - J x0| <= |b| by construction, so a bad history can only
// cost the SpMVs.  The Krylov workspace is free before the first cycle: V_1.. hold the halo-extended copies, Z_0.. hold W.
static int guess_ensure(cfdh_ctx *c) {
  const int m = c->opt.ksp_guess;
  const size_t ld = ((size_t)c->NL + 1) & ~(size_t)1;
  if (m == c->guess_m && (m == 0 || c->guessU.p)) return 0;
  c->guess_m = m;
  for (int k = 0; k < cfdh_ctx::GUESS_NEWTON; k++) c->guess_cnt[k] = c->guess_head[k] = 0;
  if (m > 0) {
    HIPCHK(c, c->guessU.alloc(ld * (size_t)m * cfdh_ctx::GUESS_NEWTON)); HIPCHK(c, c->guessU.zero(c->stream));
    HIPCHK(c, c->guessX.alloc(ld * (size_t)cfdh_ctx::GUESS_NEWTON));
  }
  return 0;
}

static int guess_project(cfdh_ctx *c, const double *b, double *x, bool *used) {
  *used = false;
  CHK(guess_ensure(c));
  const int slot = c->guess_slot, m = c->guess_m;
  if (m <= 0 || slot < 0 || slot >= cfdh_ctx::GUESS_NEWTON || c->guess_cnt[slot] == 0 || c->kry_m < m + 1) return 0;
  const int k = c->guess_cnt[slot], n = c->NO;
  const size_t ld = ((size_t)c->NL + 1) & ~(size_t)1;
  double *U = c->guessU.p + ld * (size_t)m * slot;  // the ring of this Newton index (order is irrelevant to the projection)
  double *V = c->kV.p, *Z = c->kZ.p, *hd = c->kh.p;
  if (c->nranks > 1) {  // halo-extended copies in V_1.. (the kept vectors hold owned entries only)
    for (int i = 0; i < k; i++) {
      double *t = V + (size_t)(i + 1) * ld;
      CHK(v_copy(c, n, U + (size_t)i * ld, t));
      CHK(comm_halo(c, t));
    }
    CHK(k_spmv_full_multi(c, V + ld, Z, (int)ld, k));
  } else {
    CHK(k_spmv_full_multi(c, U, Z, (int)ld, k));  // one pass over the Jacobian for all kept vectors
  }
  // Gram matrix G = W^T W (column by column) and g = W^T b
  std::vector<double> G((size_t)k * k), g(k), y(k, 0.0);
  HIPCHK(c, hipMemsetAsync(hd, 0, sizeof(double) * 8 * (size_t)(k + 1), c->stream));
  CHK(v_gram(c, n, Z, (int)ld, k, b, hd));  // one pass over W and b, ONE read-back
  CHK(comm_allreduce_dev(c, hd, 8 * (k + 1), 0));  // ONE reduction over the ranks for the whole Gram system
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, hd, sizeof(double) * 8 * (size_t)(k + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->n_host_sync++;
  for (int i = 0; i <= k; i++)
    for (int q = 0; q < k; q++) (i < k ? G[(size_t)q * k + i] : g[q]) = c->h_pinned[(size_t)i * 8 + q];
  if (c->opt.verbose > 1) {
    double bb = 0.0;
    CHK(v_norm2(c, n, b, &bb));
    fprintf(stderr, "[cfdh]     guess diag (newton %d): cos(W_i, b) =", slot);
    for (int i = 0; i < k; i++) fprintf(stderr, " %.6f", g[i] / (std::sqrt(G[(size_t)i * k + i]) * bb));
    fprintf(stderr, "; |W_i|/|b| =");
    for (int i = 0; i < k; i++) fprintf(stderr, " %.4f", std::sqrt(G[(size_t)i * k + i]) / bb);
    fprintf(stderr, "\n");
  }
  // pivoted Cholesky of G with a relative drop tolerance, then the two triangular solves
  std::vector<int> piv;
  std::vector<double> Lc((size_t)k * k, 0.0), dg(k);
  std::vector<char> taken(k, 0);
  for (int i = 0; i < k; i++) dg[i] = G[(size_t)i * k + i];
  const double dmax0 = *std::max_element(dg.begin(), dg.end());
  if (!(dmax0 > 0.0) || !std::isfinite(dmax0)) return 0;
  for (int r = 0; r < k; r++) {
    int p = -1;
    for (int i = 0; i < k; i++) if (!taken[i] && (p < 0 || dg[i] > dg[p])) p = i;
    if (p < 0 || !(dg[p] > 1e-10 * G[(size_t)p * k + p]) || !(dg[p] > 1e-14 * dmax0)) break;
    taken[p] = 1;
    const int rr = (int)piv.size();
    piv.push_back(p);
    const double lpp = std::sqrt(dg[p]);
    Lc[(size_t)p * k + rr] = lpp;
    for (int i = 0; i < k; i++) {
      if (taken[i]) continue;
      double sacc = G[(size_t)i * k + p];
      for (int q = 0; q < rr; q++) sacc -= Lc[(size_t)i * k + q] * Lc[(size_t)p * k + q];
      Lc[(size_t)i * k + rr] = sacc / lpp;
      dg[i] -= Lc[(size_t)i * k + rr] * Lc[(size_t)i * k + rr];
    }
  }
  const int r = (int)piv.size();
  if (r == 0) return 0;
  std::vector<double> t(r);
  for (int a = 0; a < r; a++) {
    double sacc = g[piv[a]];
    for (int q = 0; q < a; q++) sacc -= Lc[(size_t)piv[a] * k + q] * t[q];
    t[a] = sacc / Lc[(size_t)piv[a] * k + a];
  }
  for (int a = r - 1; a >= 0; a--) {
    double sacc = t[a];
    for (int q = a + 1; q < r; q++) sacc -= Lc[(size_t)piv[q] * k + a] * y[piv[q]];
    y[piv[a]] = sacc / Lc[(size_t)piv[a] * k + a];
  }
  for (int i = 0; i < k; i++) if (!std::isfinite(y[i])) return 0;
  HIPCHK(c, hipMemcpyAsync(c->ky.p, y.data(), sizeof(double) * k, hipMemcpyHostToDevice, c->stream));
  CHK(v_lincomb(c, n, U, (int)ld, k, c->ky.p, x));  // x (zeroed by the caller) += U y
  // singular system: the guess, like every preconditioned vector, carries no constant-pressure component (the Krylov vectors
  // cannot remove one, and through the kept corrections it would feed back from step to step)
  if (c->singular) CHK(v_sub_mean(c, c->nvo, x + (size_t)c->dim * c->nvo));
  // r0 = b - J x0 = b - W y without another product (the true residual is formed after every cycle anyway)
  CHK(v_copy(c, n, b, V));
  CHK(v_multiaxpy(c, n, Z, (int)ld, k, c->ky.p, V));
  HIPCHK(c, hipStreamSynchronize(c->stream));        // y is a host temporary
  *used = true;
  return 0;
}

// keep the solution of a converged solve for the guesses of later time steps
static int guess_store(cfdh_ctx *c, const double *x) {
  const int slot = c->guess_slot, m = c->guess_m;
  if (m <= 0 || slot < 0 || slot >= cfdh_ctx::GUESS_NEWTON) return 0;
  const size_t ld = ((size_t)c->NL + 1) & ~(size_t)1;
  double *U = c->guessU.p + ld * (size_t)m * slot;
  CHK(v_copy(c, c->NO, x, U + (size_t)c->guess_head[slot] * ld));
  c->guess_stored[slot] = true;
  c->guess_head[slot] = (c->guess_head[slot] + 1) % m;
  if (c->guess_cnt[slot] < m) c->guess_cnt[slot]++;
  return 0;
}

// At the end of a converged step the kept solutions are replaced by what the solves were approximating: the remaining Newton
// correction x_k - x_final (J_k d = F(x_k), x_final = x_k - d up to the quadratic remainder).  The linear solves stop at
// rtol 1e-5; x_final is converged to snes_rtol, so the kept vectors lose the solver noise that otherwise bounds the quality of
// the next steps' projections at ~1e-4.
static int guess_refine(cfdh_ctx *c, int newton_its, const double *x_final) {
  const int m = c->guess_m;
  if (m <= 0 || !c->guessX.p) return 0;
  const size_t ld = ((size_t)c->NL + 1) & ~(size_t)1;
  for (int k = 0; k < newton_its && k < cfdh_ctx::GUESS_NEWTON; k++) {
    if (!c->guess_stored[k]) continue;
    double *U = c->guessU.p + ld * (size_t)m * k;
    const int latest = (c->guess_head[k] + m - 1) % m;
    CHK(v_waxpy(c, c->NO, -1.0, x_final, c->guessX.p + (size_t)k * ld, U + (size_t)latest * ld));
  }
  return 0;
}

// Solve J x = b, x0 = 0 or the projected guess.  Right preconditioning, convergence on the true
// residual norm relative to |b| (KSP defaults: rtol, atol; KSP_NORM_UNPRECONDITIONED
// for FGMRES).  Classical Gram-Schmidt with one re-orthogonalisation pass; the
// 2j+3 scalars of an iteration come back in a single read.
int cfdh_fgmres(cfdh_ctx *c, const double *b, double *x, int *its_out, int *reason_out, double bnorm) {
  CHK(ensure_krylov(c));
  const int n = c->NO, m = c->kry_m;
  const size_t ld = ((size_t)c->NL + 1) & ~(size_t)1;
  const cfdh_options &o = c->opt;
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), hh(2 * (size_t)(m + 1) + 8);
  double bn = bnorm;  // the caller may know |b| already (Newton: |F| of the accepted iterate)
  CHK(v_zero(c, c->NL, x));
  if (!(bn >= 0.0)) CHK(v_norm2(c, n, b, &bn));
  int its = 0, reason = 0;
  if (!std::isfinite(bn)) { *its_out = 0; *reason_out = -9; return 0; }
  if (bn == 0.0) { *its_out = 0; *reason_out = 2; return 0; }
  const double tol = std::max(o.ksp_rtol * bn, o.ksp_atol);
  double *V = c->kV.p, *Z = c->kZ.p, *w = c->kw.p, *hd = c->kh.p;
  bool first = true;
  bool guessed = false;
  CHK(guess_project(c, b, x, &guessed));
  if (guessed) first = false;  // the cycle starts from the true residual of x0, as after a restart
  // Long cycles at the default tolerance orthogonalise against an fp32 COPY of the basis (half the traffic of the two passes over
  // V, which are ~40 % of an iteration at depth 25).  Chosen per solve from the length of the last solve with the same Newton
  // index; never for tolerances below 1e-6 (parity runs), never again on a context whose watchdog tripped with the copy in use.
  // CFDH_KRYLOV_FP32 = 0: never, 2: always (where the tolerance allows).
  const int fp32_env = getenv("CFDH_KRYLOV_FP32") ? atoi(getenv("CFDH_KRYLOV_FP32")) : 1;
  const int gslot = c->guess_slot;
  const bool expect_long = gslot >= 0 && gslot < cfdh_ctx::GUESS_NEWTON && c->guess_last_its[gslot] >= 20;
  // In a partitioned run the copy costs one more all-reduce per iteration (the measured norm) and saves 1/nranks of the traffic it
  // saves on one GPU: taken only where a rank still holds >= 2 M unknowns (decided from the GLOBAL count: the same on every rank).
  const bool worth32 = c->nranks <= 1 || (c->nvo_global / c->nranks) * (c->dim + 1) >= 2.0e6;
  bool use32 = fp32_env > 0 && o.ksp_rtol >= 1e-6 && c->krylov_fp32_ok && ((expect_long && worth32) || fp32_env >= 2);
  float *V32 = nullptr;
  const size_t ld32 = (ld + 3) & ~(size_t)3;  // columns of the copy start on 16-B boundaries (float4 loads)
  if (use32) {
    if (c->kV32.n < ld32 * (size_t)(m + 1)) HIPCHK(c, c->kV32.alloc(ld32 * (size_t)(m + 1)));
    V32 = c->kV32.p;
  }
  double last_true = -1.0;  // true residual at the last disagreement between recurrence and true residual
  double est_prev = 0.0, beta_start = bn;  // residual estimate at the end / true residual at the start of the last cycle
  int j_prev = 0;
  for (;;) {
    double beta;
    if (first) {
      beta = bn;  // r0 = b: v_0 = b / |b| is formed below straight from b
    } else {
      if (!(guessed && its == 0 && j_prev == 0)) {  // (after a projected guess V_0 already holds r0 = b - W y)
        CHK(comm_halo(c, x));
        CHK(k_spmv_full(c, x, w));
        CHK(v_waxpy(c, n, -1.0, w, b, V));
      } else if (getenv("CFDH_GUESS_CHECK")) {
        // test hook: the residual assembled from the multi-vector product must be the true residual of x0
        double diff = 0.0;
        CHK(comm_halo(c, x));
        CHK(k_spmv_full(c, x, w));
        CHK(v_waxpy(c, n, -1.0, w, b, w));   // w = b - J x0
        CHK(v_waxpy(c, n, -1.0, V, w, w));   // w -= V_0
        CHK(v_norm2(c, n, w, &diff));
        if (!(diff <= 1e-10 * bn)) return cfdh_fail(c, CFDH_E_STATE, "projected guess: |(b - J x0) - (b - W y)| = %.3e |b|", diff / bn);
      }
      CHK(v_norm2(c, n, V, &beta));
      if (guessed && its == 0 && j_prev == 0) {
        c->n_guess_solves++; c->guess_reduction_sum += beta / bn;
        if (o.verbose) fprintf(stderr, "[cfdh]     projected guess (newton %d, %d vectors): |r0| / |b| = %.3e\n", c->guess_slot, c->guess_cnt[c->guess_slot], beta / bn);
      }
      // Orthogonality watchdog.  Unrefined classical Gram-Schmidt (PETSc's default) can lose the basis in a long cycle:
      // the recurrence then reports convergence while the true residual, formed here after every cycle anyway, does not
      // follow.  Once that is seen on a context, long cycles are re-orthogonalised (DGKS) from then on, and a cycle that
      // made the residual worse is taken back.
      if (use32 && j_prev > 0 && beta > 10.0 * std::max(est_prev, tol)) {
        // the fp32 copy is the first suspect: this solve and all later ones of the context go back to the fp64 basis
        use32 = false; c->krylov_fp32_ok = false;
        if (o.verbose) fprintf(stderr, "[cfdh]     fgmres: true residual %.3e vs recurrence %.3e after a %d-vector cycle with the fp32 basis copy: switched off\n", beta, est_prev, j_prev);
      } else if (!c->gs_refine_long && j_prev > 0 && beta > 10.0 * std::max(est_prev, tol)) {
        c->gs_refine_long = true;
        if (o.verbose) fprintf(stderr, "[cfdh]     fgmres: true residual %.3e vs recurrence %.3e after a %d-vector cycle: re-orthogonalising long cycles from now on\n", beta, est_prev, j_prev);
        if (beta > beta_start && j_prev > 0) {
          for (int i = 0; i < j_prev; i++) y[i] = -y[i];
          HIPCHK(c, hipMemcpyAsync(c->ky.p, y.data(), sizeof(double) * j_prev, hipMemcpyHostToDevice, c->stream));
          CHK(v_lincomb(c, n, Z, (int)ld, j_prev, c->ky.p, x));
          HIPCHK(c, hipStreamSynchronize(c->stream));
          est_prev = beta_start;  // do not trip again on the restored iterate
          continue;
        }
      }
      // Attainable accuracy.  The convergence test of a cycle is the recurrence norm (as in PETSc, which never looks further);
      // here the true residual is formed after every cycle and another cycle follows if it disagrees.  When a second cycle
      // in a row ends "converged" by the recurrence without halving the true residual, that residual is the floor of this
      // system (rounding level of J x, or the component of b outside the range of a singular Jacobian: lid cavity at
      // ksp_rtol 1e-10) and the iteration stops instead of spending ksp_max_it on it.
      // The stop is bounded and reported as what it is: it applies only within 10x the tolerance or once the residual is six
      // orders below |b| (beyond what the reference's rtol = 1e-5 ever asks for); anything else runs on to ksp_max_it and fails
      // like PETSc's DIVERGED_ITS.  It returns its own reason code (CFDH_KSP_CONVERGED_ATTAINABLE) and is counted
      // (cfdh_info 72).  CFDH_NO_ATTAINABLE_STOP=1 disables it.
      static const bool no_attainable = getenv("CFDH_NO_ATTAINABLE_STOP") && getenv("CFDH_NO_ATTAINABLE_STOP")[0] == '1';
      if (j_prev > 0 && est_prev <= tol && beta > tol && !no_attainable) {
        if (last_true > 0.0 && beta > 0.5 * last_true && (beta <= 10.0 * tol || beta <= 1e-6 * bn)) {
          if (o.verbose) fprintf(stderr, "[cfdh]     fgmres: true residual %.3e stays above the tolerance %.3e although the recurrence converged twice: attainable accuracy, stopping\n", beta, tol);
          reason = CFDH_KSP_CONVERGED_ATTAINABLE;
          c->n_attainable_stops++;
          break;
        }
        last_true = beta;
      } else {
        last_true = -1.0;  // a cycle that ended without recurrence convergence (restart): the floor has to be seen twice IN A ROW
      }
      beta_start = beta;
    }
    if (beta <= tol) { reason = 2; break; }
    if (its >= o.ksp_max_it) { reason = -3; break; }
    if (!std::isfinite(beta)) { reason = -9; break; }
    if (first) CHK(v_scale_to(c, n, 1.0 / beta, b, V));
    else CHK(v_scale(c, n, 1.0 / beta, V));
    if (use32) CHK(v_store32(c, n, V, V32));
    first = false;
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    // ---- one cycle.  The device needs nothing from the host to go from one iteration to the next (the Gram-Schmidt update forms
    // its own scale from the reduced coefficients), so iterations are LAUNCHED ahead of the host's bookkeeping: every iteration in
    // flight publishes [h_0..h_j, w.w, (measured norm)] into its own slot of a host-mapped ring behind an event, and the host
    // catches up -- Hessenberg column, Givens rotation, convergence test -- in batches, always leaving one iteration running so
    // that the GPU never waits for the host.  How far to run ahead follows from the convergence rate seen so far (iterations the
    // tolerance is still away at the current contraction factor, at most KRING - 3): an iteration launched beyond the one that
    // converges is discarded (x uses the columns up to convergence only) and counted (cfdh_info 73).  Solves that decide a second
    // Gram-Schmidt pass per vector (ksp_rtol < 1e-7, long cycles after the watchdog tripped) stay synchronous; the rare
    // cancellation case (|w'|^2 < 1e-2 |w|^2) refines the vector when the host sees it and re-launches what ran ahead of it.
    static const int refine_env = getenv("CFDH_GS_REFINE_FROM") ? atoi(getenv("CFDH_GS_REFINE_FROM")) : -1;
    static const int lag_env = getenv("CFDH_KSP_LAG") ? atoi(getenv("CFDH_KSP_LAG")) : cfdh_ctx::KRING - 3;
    const int refine_from = refine_env >= 0 ? refine_env : (c->gs_refine_long ? 24 : (1 << 30));
    const int lagmax = (o.ksp_rtol < 1e-7 || c->prof_on) ? 0 : std::max(0, std::min(lag_env, cfdh_ctx::KRING - 3));
    const int its_base = its;
    const int maxl = std::min(m, o.ksp_max_it - its_base);  // iterations this cycle may run
    int j = 0;    // iterations processed by the host (complete Hessenberg columns)
    int jl = 0;   // iterations launched
    // expected length of this solve: the last solve with the same Newton index took guess_last_its iterations (0: unknown)
    const int e_its = (gslot >= 0 && gslot < cfdh_ctx::GUESS_NEWTON) ? c->guess_last_its[gslot] : 0;
    // iterations still needed, counted from iteration j: from the history alone before any residual of this cycle is known,
    // afterwards the smaller of the rate-based prediction and what the history leaves
    int need = std::max(1, std::min(e_its - its - 2, 3));
    double res_hist[4] = {beta, 0, 0, 0};
    int nhist = 1;
    bool done = false;
    double *s_dev = hd + 2 * (m + 2) + 1;
    const size_t rs = c->h_ring_stride;
    auto launch = [&](int jj) -> int {
      double *vj = V + (size_t)jj * ld, *zj = Z + (size_t)jj * ld, *vn = V + (size_t)(jj + 1) * ld;
      double *slot = c->h_ring_dev + (size_t)(jj % cfdh_ctx::KRING) * rs;
      CHK(cfdh_pc_apply(c, vj, zj));
      CHK(comm_halo(c, zj));
      CHK(k_spmv_full(c, zj, w));
      // classical Gram-Schmidt (PETSc's default for (F)GMRES): h = [V^T w ; w.w] comes from ONE fused multi-dot
      if (use32) {
        // the same Gram-Schmidt step against the fp32 copy: h = V32^T w, v_{j+1} = (w - V32 h) / |w - V32 h| with the norm
        // MEASURED (the identity below needs columns that are orthonormal to round-off); both the fp64 vector (input of the
        // next preconditioner application) and its fp32 copy are written
        CHK(v_multidot32(c, n, V32, (int)ld32, jj + 1, w, hd, slot));
        CHK(v_gs_update32(c, n, V32, (int)ld32, jj + 1, hd, w, vn, V32 + (size_t)(jj + 1) * ld32, s_dev, slot + (jj + 2)));
        HIPCHK(c, hipEventRecord(c->ev_ring[jj % cfdh_ctx::KRING], c->stream));
      } else {
        CHK(v_multidot(c, n, V, (int)ld, jj + 1, w, hd, true, slot));
        // h (all-reduced in a partitioned run) sits in the host-mapped slot: the event marks THAT; the update of w below
        // overlaps with the host's Hessenberg bookkeeping and the next launches
        HIPCHK(c, hipEventRecord(c->ev_ring[jj % cfdh_ctx::KRING], c->stream));
        // v_{j+1} = (w - V h) / s with s = sqrt(w.w - |h|^2) formed on the device from the reduced coefficients: update and
        // normalisation in one pass (the host forms the same norm for the Hessenberg matrix from its copy of h)
        CHK(v_gs_update_normalize(c, n, V, (int)ld, jj + 1, hd, w, vn, s_dev));
      }
      return 0;
    };
    while (!done && reason == 0) {
      // launch: one iteration beyond the batch the host will process next stays in flight, unless the batch is predicted to
      // contain the converging iteration
      const bool sync_now = lagmax == 0 || j >= refine_from;
      // (one short of the prediction when it is long: the last predicted iteration is confirmed before anything follows it)
      const int ahead = sync_now ? 1 : std::max(1, std::min(need - (need >= 4 ? 1 : 0), lagmax + 1));   // iterations wanted in flight
      while (jl < maxl && jl - j < ahead) { CHK(launch(jl)); jl++; }
      if (jl == j) break;  // cycle full (restart) or iteration cap
      // process: everything launched if that may finish the solve / the cycle, otherwise all but the newest
      int upto = jl;
      if (!sync_now && jl - j > 1 && jl < maxl && need > jl - j) upto = jl - 1;
      c->n_host_sync++;
      HIPCHK(c, hipEventSynchronize(c->ev_ring[(upto - 1) % cfdh_ctx::KRING]));
      for (; j < upto && !done; ) {
        const double *hs = c->h_ring + (size_t)(j % cfdh_ctx::KRING) * rs;
        double *vn = V + (size_t)(j + 1) * ld;
        double ww = hs[j + 1], hh2 = 0.0;
        for (int i = 0; i <= j; i++) { hh[i] = hs[i]; hh2 += hh[i] * hh[i]; }
        double nrm2 = use32 ? hs[j + 2] * hs[j + 2] : ww - hh2;
        // PETSc's default never refines.  Here: tolerances down to ~1e-7 refine only when two digits
        // cancel; tighter solves (parity runs at 1e-10) use the DGKS criterion (|w'| < |w|/sqrt(2)),
        // because classical Gram-Schmidt then loses the orthogonality the deep convergence needs
        // ... and so can a long Krylov cycle: beyond ~two dozen vectors the unrefined basis may drift far enough from
        // orthogonality that the least-squares solution picks up huge spurious components (Newton corrections ten times
        // the size of the iterate on the tree domain, config 5) although the residual norm looks converged.  Paying the
        // second pass on every long cycle costs 14 % of a 3-D step whose 47-vector cycles never need it, so it is switched
        // on by the watchdog above (CFDH_GS_REFINE_FROM=<j> forces it from vector j on)
        // (at the reference's tolerance: only when three digits cancel -- one classical pass then leaves an orthogonality
        // error of ~1e3 eps, far below rtol 1e-5.  With a good preconditioner w = J M^-1 v_j is close to v_j, so a threshold of
        // 1e-2 -- rounds 2 and 3 -- sent a third of all iterations of the headline run through the second pass for nothing.)
        static const double eta2_env = getenv("CFDH_GS_ETA2") ? atof(getenv("CFDH_GS_ETA2")) : 1e-6;
        const double eta2 = (o.ksp_rtol < 1e-7 || j >= refine_from) ? 0.5 : eta2_env;
        const bool refine = !(nrm2 > eta2 * ww);
        double hnorm;
        if (refine) {
          // second Gram-Schmidt pass on the (already scaled) vector: vn = w'/s  ->  h2 = V^T vn, vn -= V h2, vn /= |vn|;
          // in terms of w: h += s h2, |w''| = s |vn|.  The coefficients and the squared norm of the corrected vector travel
          // in ONE reduction over the ranks: [h2 ; vn.vn] from one fused multi-dot, |vn - V h2|^2 = vn.vn - |h2|^2 (the basis is
          // orthonormal to round-off here -- the fp32 copy measures its norm instead)
          if (jl > j + 1) { c->n_krylov_discarded += jl - (j + 1); jl = j + 1; }  // what ran ahead used the unrefined vector
          CHK(v_multidot(c, n, V, (int)ld, j + 1, vn, hd + (m + 2), true));
          CHK(v_multiaxpy(c, n, V, (int)ld, j + 1, hd + (m + 2), vn));
          if (use32) CHK(v_norm_to_dev(c, n, vn, hd + 2 * (m + 2)));
          HIPCHK(c, hipMemcpyAsync(c->h_pinned, hd + (m + 2), sizeof(double) * (m + 4), hipMemcpyDeviceToHost, c->stream));
          c->n_host_sync++;
          HIPCHK(c, hipStreamSynchronize(c->stream));
          // scale of the first pass, formed exactly as gs_update_normalize_kernel forms it (the device word s_dev may belong to
          // an iteration that ran ahead by now); the fp32 path measured it: slot word j + 2
          const double s = use32 ? hs[j + 2] : ((nrm2 > 0.0 && nrm2 <= ww) ? std::sqrt(nrm2) : std::sqrt(ww));
          double h22 = 0.0;
          for (int i = 0; i <= j; i++) { hh[i] += s * c->h_pinned[i]; h22 += c->h_pinned[i] * c->h_pinned[i]; }
          double vnorm;
          if (use32) vnorm = c->h_pinned[m + 2];
          else {
            const double d2 = c->h_pinned[j + 1] - h22;
            vnorm = d2 > 0.0 ? std::sqrt(d2) : 0.0;
          }
          hnorm = s * vnorm;
          if (use32) { CHK(v_scale_inv_dev(c, n, vn, hd + 2 * (m + 2), vn)); CHK(v_store32(c, n, vn, V32 + (size_t)(j + 1) * ld32)); }
          else CHK(v_scale(c, n, vnorm > 0.0 ? 1.0 / vnorm : 0.0, vn));
        } else {
          hnorm = std::sqrt(nrm2);
        }
        double *Hj = &H[(size_t)j * (m + 1)];
        for (int i = 0; i <= j; i++) Hj[i] = hh[i];
        Hj[j + 1] = hnorm;
        for (int i = 0; i < j; i++) {
          const double t = cs[i] * Hj[i] + sn[i] * Hj[i + 1];
          Hj[i + 1] = -sn[i] * Hj[i] + cs[i] * Hj[i + 1];
          Hj[i] = t;
        }
        const double d = std::hypot(Hj[j], Hj[j + 1]);
        if (!(d > 0) || !std::isfinite(d)) { reason = -9; j++; break; }
        cs[j] = Hj[j] / d; sn[j] = Hj[j + 1] / d;
        Hj[j] = d; Hj[j + 1] = 0.0;
        g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
        its++;
        j++;
        const double res = std::fabs(g[j]);
        if (o.verbose > 1) fprintf(stderr, "[cfdh]     fgmres %3d  |r|/|b| = %.3e\n", its, res / bn);
        if (res <= tol) { done = true; break; }
        // iterations the tolerance is away at the contraction factor of the last (up to three) iterations
        if (nhist < 4) res_hist[nhist++] = res;
        else { res_hist[0] = res_hist[1]; res_hist[1] = res_hist[2]; res_hist[2] = res_hist[3]; res_hist[3] = res; }
        const double rho = std::pow(res / res_hist[0], 1.0 / (nhist - 1));
        int n_rem = (rho > 0.0 && rho < 0.97) ? (int)std::ceil(std::log(tol / res) / std::log(rho)) : (1 << 20);
        if (nhist == 2) n_rem = std::min(n_rem, 3);  // one sample of the rate: a short look ahead only
        need = std::max(1, e_its > 0 ? std::min(n_rem, std::max(e_its - its, 1) + 2) : n_rem);
      }
      if (done || reason != 0) break;
    }
    if (jl > j) c->n_krylov_discarded += jl - j;  // launched ahead of the converging iteration: not part of the solution
    if (reason == -9) break;
    // y = H^-1 g ; x += Z y
    for (int i = j - 1; i >= 0; i--) {
      double s = g[i];
      for (int k = i + 1; k < j; k++) s -= H[(size_t)k * (m + 1) + i] * y[k];
      y[i] = s / H[(size_t)i * (m + 1) + i];
    }
    // y travels through a pinned slot behind the ring (no stream synchronisation: the slot is rewritten at the end of the next
    // cycle at the earliest, after events recorded behind this copy have been waited for)
    double *ystage = c->h_ring + c->h_ring_stride * cfdh_ctx::KRING;
    for (int i = 0; i < j; i++) ystage[i] = y[i];
    HIPCHK(c, hipMemcpyAsync(c->ky.p, ystage, sizeof(double) * j, hipMemcpyHostToDevice, c->stream));
    CHK(v_lincomb(c, n, Z, (int)ld, j, c->ky.p, x));
    est_prev = std::fabs(g[j]);
    j_prev = j;
    (void)done;
  }
  c->n_krylov += its;
  *its_out = its;
  *reason_out = reason;
  if (gslot >= 0 && gslot < cfdh_ctx::GUESS_NEWTON) c->guess_last_its[gslot] = its;
  if (reason > 0) CHK(guess_store(c, x));
  return 0;
}

static int upload_bc(cfdh_ctx *c) {
  if (!c->bc_dirty) return 0;
  const int st = c->dim + 1;
  // vertices whose device entries may be stale: cleared since the last upload, or (re)written since
  // (entries of bc_touched below the watermark went up with an earlier upload; cfdh_update_dirichlet queues its nodes itself)
  c->bc_pending.insert(c->bc_pending.end(), c->bc_touched.begin() + (std::ptrdiff_t)std::min(c->bc_touched_sent, c->bc_touched.size()), c->bc_touched.end());
  if ((int)c->bc_mark.size() != c->nv) { c->bc_mark.assign(c->nv, -1); c->bc_full_upload = true; }
  c->bc_epoch++;
  size_t K = 0;
  for (int v : c->bc_pending)
    if (c->bc_mark[v] != c->bc_epoch) { c->bc_mark[v] = c->bc_epoch; c->bc_pending[K++] = v; }
  c->bc_pending.resize(K);
  // bc_touched keeps growing with duplicates when objects are re-added without a clear in between: compact it as well
  if (c->bc_touched.size() > 4 * (size_t)c->nv) { std::sort(c->bc_touched.begin(), c->bc_touched.end()); c->bc_touched.erase(std::unique(c->bc_touched.begin(), c->bc_touched.end()), c->bc_touched.end()); }
  c->bc_touched_sent = c->bc_touched.size();
  // staging of the sparse update: declared here so that they outlive the stream synchronisation below
  std::vector<unsigned char> fl;
  std::vector<double> va, mu;
  if (c->bc_full_upload || K > (size_t)c->nv / 8) {
    HIPCHK(c, hipMemcpyAsync(c->bcflag.p, c->h_bcflag.data(), c->h_bcflag.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->bcval.p, c->h_bcval.data(), sizeof(double) * c->h_bcval.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->bcmult.p, c->h_bcmult.data(), sizeof(double) * c->h_bcmult.size(), hipMemcpyHostToDevice, c->stream));
    c->bc_full_upload = false;
  } else if (K > 0) {
    // sparse update: (vertex, flag, values, multiplicities) of the K vertices, scattered on the device
    fl.resize(K); va.resize(K * st); mu.resize(K * st);
    for (size_t k = 0; k < K; k++) {
      const int v = c->bc_pending[k];
      fl[k] = c->h_bcflag[v];
      for (int i = 0; i < st; i++) { va[k * st + i] = c->h_bcval[(size_t)st * v + i]; mu[k * st + i] = c->h_bcmult[(size_t)st * v + i]; }
    }
    HIPCHK(c, c->bc_uidx.upload(c->bc_pending, c->stream));
    HIPCHK(c, c->bc_uflag.upload(fl, c->stream));
    HIPCHK(c, c->bc_uval.upload(va, c->stream));
    HIPCHK(c, c->bc_umult.upload(mu, c->stream));
    CHK(k_bc_scatter(c, (int)K, st, c->bc_uidx.p, c->bc_uflag.p, c->bc_uval.p, c->bc_umult.p));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->bc_pending.clear();
  c->bc_dirty = false;
  return 0;
}

int cfdh_prepare_assembly(cfdh_ctx *c) {
  CHK(upload_bc(c));
  if (!c->mom_valid) {
    CHK(comm_halo(c, c->xprev.p));
    CHK(k_moments(c));
  }
  return 0;
}

int cfdh_newton_step(cfdh_ctx *c, cfdh_stats *st) {
  const int n = c->NO, nvo = c->nvo;
  const cfdh_options &o = c->opt;
  const double t_begin = wall_ms();
  memset(st, 0, sizeof *st);
  c->last_stats = *st;
  if (!c->params_set) return cfdh_fail(c, CFDH_E_STATE, "cfdh_set_params was not called");
  if (!c->state_set) return cfdh_fail(c, CFDH_E_STATE, "cfdh_set_state was not called");
  CHK(cfdh_prepare_assembly(c));
  if (o.remove_p_mean) CHK(v_sub_mean(c, nvo, c->x.p + (size_t)c->dim * nvo));
  CHK(comm_halo(c, c->x.p));
  double t0 = wall_ms();
  CHK(k_assemble(c, c->x.p, 1));
  double fn;
  CHK(v_norm2(c, n, c->F.p, &fn));
  st->ms_assemble += wall_ms() - t0;
  st->fnorm0 = fn;
  // constant-pressure null space (MatNullSpaceTest: |J n| < 1e-7 for the unit vector n).  PETSc's bound is absolute; on a
  // mesh in metres with millimetre cells every entry of J lies below it and ANY matrix would pass (3-D bifurcation with
  // do-nothing outlets: the pressure level is fixed, yet the test says singular and the projected solve stagnates at 2e-2).
  // The vector is therefore accepted only if it is also small relative to |J| n.
  {
    double nrm, absnrm;
    CHK(k_nullspace_test(c, &nrm, &absnrm));
    const double np = c->nranks > 1 ? c->nvo_global : (double)nvo;
    const int sing = ((nrm / std::sqrt(np)) < 1e-7 && nrm <= 1e-6 * absnrm) ? 1 : 0;
    if (sing != c->singular) { c->singular = sing; c->pc_valid = false; }
  }
  bool force_refresh = (o.pc_refresh > 0 && c->steps_since_refresh >= o.pc_refresh);
  int reason = 0;
  double *x = c->x.p, *xt = c->xt.p, *d = c->dvec.p;
  // The trial point of the line search is assembled with its Jacobian, because an accepted point that has not converged
  // needs it next.  When the previous reduction predicts that the trial WILL meet the tolerance, only the residual is
  // assembled there (mode 2: same residual bit for bit, no matrix written); a wrong prediction costs one more pass.
  bool jac_current = true;
  double fn_before = 0.0;
  CHK(guess_ensure(c));
  for (int it = 0;; it++) {
    if (o.verbose) fprintf(stderr, "[cfdh]   newton %d |F| = %.6e\n", it, fn);
    if (!std::isfinite(fn)) { reason = CFDH_DIVERGED_FNORM_NAN; break; }
    if (fn < o.snes_atol) { reason = CFDH_CONVERGED_FNORM_ABS; break; }
    if (it > 0 && fn <= o.snes_rtol * st->fnorm0) { reason = CFDH_CONVERGED_FNORM_RELATIVE; break; }
    if (it >= o.snes_max_it) { reason = CFDH_DIVERGED_MAX_IT; break; }
    t0 = wall_ms();
    if (!jac_current) {
      CHK(k_assemble(c, x, 1));
      jac_current = true;
      st->ms_assemble += wall_ms() - t0;
      t0 = wall_ms();
    }
    const bool expect_converged = it > 0 && fn_before > 0.0 && 10.0 * (fn / fn_before) * fn <= o.snes_rtol * st->fnorm0;
    fn_before = fn;
    CHK(cfdh_pc_update(c, force_refresh || o.pc_refresh < 0));
    force_refresh = false;
    st->ms_pc_setup += wall_ms() - t0;
    t0 = wall_ms();
    int kits = 0, kreason = 0;
    c->guess_slot = it;
    if (it < cfdh_ctx::GUESS_NEWTON) {
      c->guess_stored[it] = false;
      if (c->guessX.p) CHK(v_copy(c, n, x, c->guessX.p + (size_t)it * (((size_t)c->NL + 1) & ~(size_t)1)));
    }
    CHK(cfdh_fgmres(c, c->F.p, d, &kits, &kreason, fn));
    if (kreason < 0 && c->pc_its_ref > 0) {
      // a lagged hierarchy that stopped working: rebuild once and retry
      CHK(cfdh_pc_update(c, true));
      st->krylov_its += kits;
      CHK(cfdh_fgmres(c, c->F.p, d, &kits, &kreason, fn));
    }
    c->guess_slot = -1;
    st->krylov_its += kits;
    st->ms_solve += wall_ms() - t0;
    if (kreason < 0) { reason = CFDH_DIVERGED_LINEAR_SOLVE; cfdh_fail(c, CFDH_E_DIVERGED, "FGMRES failed (reason %d) after %d iterations", kreason, kits); break; }
    // adaptive lagging: rebuild when a solve needs 1.5x (+5) the iterations the fresh hierarchy needed.  The reference count
    // is the LARGEST solve of the step the hierarchy was built in and of the step after it: the solves of one step differ
    // (the first Newton iteration often only repairs boundary rows in a handful of iterations), and a reference taken from
    // such a solve made every later one trip the rule -- 16 rebuilds in a row during the start-up ramp of config 5.
    if (c->pc_its_ref == 0) c->pc_its_ref = std::max(kits, 1);
    else if (c->steps_since_refresh <= 1) c->pc_its_ref = std::max(c->pc_its_ref, kits);
    else if (o.pc_refresh == 0 && kits > (3 * c->pc_its_ref) / 2 + 5) force_refresh = true;
    // backtracking line search on 1/2 |F|^2 (Dennis-Schnabel, alpha = 1e-4)
    t0 = wall_ms();
    double lam = 1.0, fnew = 0.0;
    bool ok = false;
    for (int ls = 0; ls < 40; ls++) {
      CHK(v_waxpy(c, n, -lam, d, x, xt));
      CHK(comm_halo(c, xt));
      CHK(k_assemble(c, xt, expect_converged ? 2 : 1));  // residual (and Jacobian) at the trial point in one pass
      CHK(v_norm2(c, n, c->F.p, &fnew));
      if (std::isfinite(fnew) && (fnew * fnew <= fn * fn * (1.0 - 2.0e-4 * lam) || fnew < o.snes_atol)) { ok = true; break; }
      double l2 = std::isfinite(fnew) ? fn * fn * lam * lam / (2.0 * (0.5 * fnew * fnew - 0.5 * fn * fn + fn * fn * lam)) : 0.0;
      if (!(l2 > 0.1 * lam)) l2 = 0.1 * lam;
      if (l2 > 0.5 * lam) l2 = 0.5 * lam;
      lam = l2;
    }
    st->ms_assemble += wall_ms() - t0;
    if (!ok) { reason = CFDH_DIVERGED_LINE_SEARCH; break; }
    double dn, xn;
    CHK(v_norm2_pair(c, n, d, xt, &dn, &xn));
    if (o.verbose) fprintf(stderr, "[cfdh]     step length %.3e, |dx| = %.3e, |x| = %.3e, %d FGMRES iterations\n", lam, dn, xn, kits);
    std::swap(c->x.p, c->xt.p);
    x = c->x.p; xt = c->xt.p;
    jac_current = !expect_converged;
    st->newton_its = it + 1;
    fn = fnew;
    if (lam * dn < o.snes_stol * xn && fn > o.snes_rtol * st->fnorm0 && fn >= o.snes_atol) {
      if (o.verbose) fprintf(stderr, "[cfdh]   newton %d |F| = %.6e (stol)\n", it + 1, fn);
      reason = CFDH_CONVERGED_SNORM_RELATIVE;
      break;
    }
  }
  if (reason > 0) CHK(guess_refine(c, st->newton_its, c->x.p));
  c->steps_since_refresh++;
  st->fnorm = fn;
  st->reason = reason;
  st->pc_refreshes = c->last_stats.pc_refreshes;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  st->ms_total = wall_ms() - t_begin;
  c->last_stats = *st;
  if (reason < 0) {
    if (c->err.empty() || reason != CFDH_DIVERGED_LINEAR_SOLVE) cfdh_fail(c, CFDH_E_DIVERGED, "Did not converge, reason: %d.", reason);
    return CFDH_E_DIVERGED;
  }
  return 0;
}
