// extern "C" entry points of libcfdh.so (see include/cfdh.h).
#include <algorithm>
#include <cmath>
#include <numeric>

#include <dlfcn.h>

#include "cfdh_internal.hpp"

std::string g_cfdh_last_error;

int cfdh_prepare_assembly(cfdh_ctx *c);

// every entry point makes the context's device current for the calling thread
#define ENTER(c)                         \
  do {                                   \
    if (!(c)) return CFDH_E_ARG;         \
    (void)hipSetDevice((c)->device);     \
  } while (0)

extern "C" {

int cfdh_abi_version(void) { return CFDH_ABI_VERSION; }

const char *cfdh_last_error(const cfdh_ctx *c) { return c ? c->err.c_str() : g_cfdh_last_error.c_str(); }

int cfdh_default_options(cfdh_options *o) {
  if (!o) return CFDH_E_ARG;
  // PETSc defaults + the caps the reference sets (stabilized_schur.py:269-274)
  o->snes_rtol = 1e-8; o->snes_atol = 1e-50; o->snes_stol = 1e-8; o->snes_max_it = 100;
  o->ksp_rtol = 1e-5; o->ksp_atol = 1e-50; o->ksp_max_it = 1000; o->ksp_restart = 200;
  o->cheb_degree = 3; o->cheb_ratio = 10.0; o->schur_full = 2;
  o->amg_smooth_degree = 1; o->amg_smooth_ratio = 8.0; o->amg_theta = -1.0; o->amg_max_coarse = 1000;
  o->pc_refresh = 0; o->remove_p_mean = 1; o->verbose = 0; o->pc_type = 1; o->cc_smooth_degree = 2;
  o->ksp_guess = getenv("CFDH_KSP_GUESS") ? atoi(getenv("CFDH_KSP_GUESS")) : 4;
  return 0;
}

static int create_ctx(cfdh_ctx **out, int device, int gdim, int etype, int64_t nv, int64_t nv_owned, int64_t nc, const int32_t *cells,
                      const double *coords, int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local,
                      const int32_t *facet_marker) {
  if (!out) return cfdh_fail(nullptr, CFDH_E_ARG, "null output pointer");
  *out = nullptr;
  if (gdim != 2 && gdim != 3) return cfdh_fail(nullptr, CFDH_E_ARG, "gdim must be 2 (P1 triangles) or 3 (P1 tetrahedra)");
  if (etype < 0 || etype > 3) return cfdh_fail(nullptr, CFDH_E_ARG, "unknown element type %d", etype);
  if (!cells || !coords || (nfacets > 0 && (!facet_cells || !facet_local)))
    return cfdh_fail(nullptr, CFDH_E_ARG, "null mesh array");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return cfdh_fail(nullptr, CFDH_E_HIP, "no HIP device available: libcfdh has no CPU fallback");
  if (device < 0 || device >= ndev) return cfdh_fail(nullptr, CFDH_E_ARG, "device %d out of range (%d devices)", device, ndev);
  cfdh_ctx *c = new (std::nothrow) cfdh_ctx();
  if (!c) return cfdh_fail(nullptr, CFDH_E_NOMEM, "out of host memory");
  c->device = device;
  cfdh_default_options(&c->opt);
  { const char *e = getenv("CFDH_NO_GRAPH"); c->use_graph = !(e && e[0] == '1'); }
  // No hipGraph replay under a rocprofiler-sdk tool on a HIP runtime >= 7.2.  That runtime submits the kernel packets of a graph
  // launch with ONE doorbell; ROCr's intercepted queue hands such a batch to the profiler's queue interceptor as (pointer into
  // the ring, packet count) without splitting it at the ring's wrap-around, and the interceptor walks `count` packets linearly:
  // the first graph launch whose packets straddle the end of the ring makes it read past the ring's mapping -- the SIGSEGV at
  // a page-aligned address below cfdh_pc_apply -> hipGraphLaunch of VERDICT round 3 (frames resolved in DESIGN.md section 6).
  // bench.py never met it because torch loads its own HIP 7.0 runtime first, which rings the doorbell per packet.  The kernels
  // are the same with direct launches, so a kernel trace loses nothing.  CFDH_GRAPH_UNDER_PROFILER=1 overrides.
  if (c->use_graph && (getenv("ROCP_TOOL_LIBRARIES") || dlsym(RTLD_DEFAULT, "rocprofiler_configure"))) {
    int rt = 0;
    const char *ov = getenv("CFDH_GRAPH_UNDER_PROFILER");
    if (!(ov && ov[0] == '1') && hipRuntimeGetVersion(&rt) == hipSuccess && rt >= 70200000) c->use_graph = false;
  }
  memset(&c->last_stats, 0, sizeof c->last_stats);
  int rc = 0;
  do {
    if (hipSetDevice(device) != hipSuccess) { rc = cfdh_fail(nullptr, CFDH_E_HIP, "hipSetDevice(%d) failed", device); break; }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = cfdh_fail(nullptr, CFDH_E_HIP, "hipStreamCreate failed"); break; }
    rc = gdim == 3 ? k3_upload_quadrature(c) : k_upload_quadrature(c);
    if (rc) break;
    if (etype != 0) {
      rc = gdim == 3 ? kg3_upload_tables(c) : kg_upload_tables(c);
      if (rc) break;
      rc = gdim == 3 ? cfdh_build_mesh_gen3(c, etype, nv, nv_owned, nc, cells, coords, nfacets, facet_cells, facet_local, facet_marker)
                     : cfdh_build_mesh_gen(c, etype, nv, nv_owned, nc, cells, coords, nfacets, facet_cells, facet_local, facet_marker);
      break;
    }
    rc = gdim == 3 ? cfdh_build_mesh3(c, nv, nv_owned, nc, cells, coords, nfacets, facet_cells, facet_local, facet_marker)
                   : cfdh_build_mesh(c, nv, nv_owned, nc, cells, coords, nfacets, facet_cells, facet_local, facet_marker);
  } while (0);
  if (rc) {
    g_cfdh_last_error = c->err.empty() ? g_cfdh_last_error : c->err;
    cfdh_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

int cfdh_create(cfdh_ctx **out, int device, int gdim, int64_t nv, int64_t nv_owned, int64_t nc, const int32_t *cells,
                const double *coords, int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local,
                const int32_t *facet_marker) {
  return create_ctx(out, device, gdim, CFDH_ELEM_P1, nv, nv_owned, nc, cells, coords, nfacets, facet_cells, facet_local, facet_marker);
}

int cfdh_create_elem(cfdh_ctx **out, int device, int gdim, int elem, int64_t nn, int64_t nc, const int32_t *cells, const double *node_coords,
                     int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local, const int32_t *facet_marker) {
  return create_ctx(out, device, gdim, elem, nn, nn, nc, cells, node_coords, nfacets, facet_cells, facet_local, facet_marker);
}

int cfdh_create_elem_part(cfdh_ctx **out, int device, int gdim, int elem, int64_t nn, int64_t nn_owned, int64_t nc, const int32_t *cells,
                          const double *node_coords, int64_t nfacets, const int32_t *facet_cells, const int32_t *facet_local, const int32_t *facet_marker) {
  return create_ctx(out, device, gdim, elem, nn, nn_owned, nc, cells, node_coords, nfacets, facet_cells, facet_local, facet_marker);
}

void cfdh_destroy(cfdh_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  comm_finalize(c);
  c->hS.clear(); c->hL.clear(); c->hA.clear(); c->hLg.clear();
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (auto &e : c->pc_graphs) for (auto &x : e.exec) if (x) (void)hipGraphExecDestroy(x);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->ev_h) (void)hipEventDestroy(c->ev_h);
  if (c->h_big) (void)hipHostFree(c->h_big);
  if (c->h_ring) (void)hipHostFree(c->h_ring);
  for (hipEvent_t e : c->ev_ring) if (e) (void)hipEventDestroy(e);
  hipStream_t s = c->stream;
  delete c;
  if (s) (void)hipStreamDestroy(s);
}

int cfdh_set_params(cfdh_ctx *c, double dt, double rho, double mu, double mu_facet, const double f[3]) {
  ENTER(c);
  if (!(dt > 0) || !(rho > 0) || !(mu > 0)) return cfdh_fail(c, CFDH_E_ARG, "dt, rho, mu must be positive");
  const bool changed = !c->params_set || dt != c->dt || rho != c->rho || mu != c->mu;
  c->dt = dt; c->rho = rho; c->mu = mu; c->muf = mu_facet;
  c->f[0] = f ? f[0] : 0.0; c->f[1] = f ? f[1] : 0.0; c->f[2] = (f && c->dim == 3) ? f[2] : 0.0;
  c->params_set = true;
  if (changed) { c->mom_valid = false; c->pc_valid = false; }
  return 0;
}

int cfdh_set_options(cfdh_ctx *c, const cfdh_options *o) {
  if (!c || !o) return CFDH_E_ARG;
  if (o->ksp_restart < 1 || o->ksp_restart > 1000 || o->cheb_degree < 1 || o->cc_smooth_degree < 1 || !(o->cheb_ratio > 1) || o->amg_smooth_degree < 1 ||
      !(o->amg_smooth_ratio > 1) || o->amg_max_coarse < 8 || o->amg_max_coarse > 4000 || o->ksp_guess < 0 || o->ksp_guess > 8)
    return cfdh_fail(c, CFDH_E_ARG, "option out of range");
  const bool pc_changed = o->amg_theta != c->opt.amg_theta || o->amg_max_coarse != c->opt.amg_max_coarse ||
                          o->amg_smooth_ratio != c->opt.amg_smooth_ratio || o->pc_type != c->opt.pc_type ||
                          o->amg_smooth_degree != c->opt.amg_smooth_degree ||  // composite operators exist for degree 1 only
                          o->schur_full != c->opt.schur_full;  // the velocity hierarchy covers owned + ghost vertices only for schur_full == 2 (ras)
  c->opt = *o;
  if (pc_changed) c->pc_valid = false;
  c->pc_graph_valid = false;  // degrees / schur_full are baked into the captured graph
  return 0;
}

int cfdh_clear_dirichlet(cfdh_ctx *c) {
  ENTER(c);
  // only the entries some object wrote (the arrays start out zero): O(#boundary vertices) per step, not O(nv)
  const int st = c->dim + 1;
  for (int v : c->bc_touched) {
    c->h_bcflag[v] = 0;
    for (int i = 0; i < st; i++) { c->h_bcval[(size_t)st * v + i] = 0.0; c->h_bcmult[(size_t)st * v + i] = 0.0; }
  }
  c->bc_pending.insert(c->bc_pending.end(), c->bc_touched.begin(), c->bc_touched.end());  // must be reset on the device too
  c->bc_touched.clear();
  c->bc_touched_sent = 0;
  c->n_pbc = 0;
  c->bc_dirty = true;
  c->bc_version++;
  return 0;
}

int cfdh_add_dirichlet(cfdh_ctx *c, int field, int64_t n, const int32_t *nodes, const double *values) {
  if (!c || (field != 0 && field != 1) || n < 0 || (n > 0 && (!nodes || !values))) return cfdh_fail(c, CFDH_E_ARG, "bad Dirichlet arguments");
  for (int64_t k = 0; k < n; k++)
    if (nodes[k] < 0 || nodes[k] >= c->nv) return cfdh_fail(c, CFDH_E_ARG, "Dirichlet node %d out of range", (int)nodes[k]);
  const int d = c->dim, st = d + 1;  // per vertex: d velocity components, then the pressure
  for (int64_t k = 0; k < n; k++) {
    const int v = c->perm[nodes[k]];
    c->bc_touched.push_back(v);
    if (field == 0) {
      for (int i = 0; i < d; i++) {
        c->h_bcflag[v] |= (unsigned char)(1u << i);
        c->h_bcval[(size_t)st * v + i] = values[(size_t)d * k + i];
        c->h_bcmult[(size_t)st * v + i] += 1.0;
      }
    } else {
      c->h_bcflag[v] |= (unsigned char)(1u << d);
      c->h_bcval[(size_t)st * v + d] = values[k];
      c->h_bcmult[(size_t)st * v + d] += 1.0;
      c->n_pbc++;
    }
  }
  c->bc_dirty = true;
  return 0;
}

int cfdh_update_dirichlet(cfdh_ctx *c, int field, int64_t n, const int32_t *nodes, const double *values) {
  if (!c || (field != 0 && field != 1) || n < 0 || (n > 0 && (!nodes || !values))) return cfdh_fail(c, CFDH_E_ARG, "bad Dirichlet arguments");
  const int d = c->dim, st = d + 1;
  const unsigned need = field == 0 ? ((1u << d) - 1u) : (1u << d);
  for (int64_t k = 0; k < n; k++) {
    if (nodes[k] < 0 || nodes[k] >= c->nv) return cfdh_fail(c, CFDH_E_ARG, "Dirichlet node %d out of range", (int)nodes[k]);
    if ((c->h_bcflag[c->perm[nodes[k]]] & need) != need) return cfdh_fail(c, CFDH_E_ARG, "cfdh_update_dirichlet: node %d is not constrained", (int)nodes[k]);
  }
  for (int64_t k = 0; k < n; k++) {
    const int v = c->perm[nodes[k]];
    c->bc_pending.push_back(v);  // the node is in bc_touched already (an add constrained it): only its device entry is stale
    if (field == 0) for (int i = 0; i < d; i++) c->h_bcval[(size_t)st * v + i] = values[(size_t)d * k + i];
    else c->h_bcval[(size_t)st * v + d] = values[k];
  }
  if (n > 0) c->bc_dirty = true;
  return 0;
}

// user arrays (u [nv][2], p [nv], user numbering) -> internal vector layout
// Host <-> device field transfers of the literal reference loop (scenario.py:306-307 copies the state through the
// host every step): staged through one pinned buffer (pageable copies run at a fraction of the PCIe rate) and
// permuted between the caller's numbering and the internal layout with all host threads.
typedef std::vector<double> hvec;
static int stage_buffer(cfdh_ctx *c, double **h) {
  if (!c->h_stage) HIPCHK(c, hipHostMalloc((void **)&c->h_stage, sizeof(double) * (size_t)c->NL));
  *h = c->h_stage;
  return 0;
}
static void pack_vec(const cfdh_ctx *c, const double *u, const double *p, std::vector<double> &out, const std::vector<double> *keep) {
  const int nvo = c->nvo, nv = c->nv, d = c->dim;
  out.resize((size_t)c->NL);
  if (keep) out = *keep;
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (nv > 20000)
  for (int k = 0; k < nv; k++) {
    const int v = c->iperm[k];
    const size_t uo = k < nvo ? (size_t)d * k : (size_t)(d + 1) * k, po = k < nvo ? (size_t)d * nvo + k : (size_t)(d + 1) * k + d;
    if (u) for (int i = 0; i < d; i++) out[uo + i] = u[(size_t)d * v + i];
    if (p) out[po] = p[v];
  }
}
static void unpack_vec(const cfdh_ctx *c, const std::vector<double> &in, double *u, double *p) {
  const int nvo = c->nvo, nv = c->nv, d = c->dim;
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (nv > 20000)
  for (int k = 0; k < nv; k++) {
    const int v = c->iperm[k];
    const size_t uo = k < nvo ? (size_t)d * k : (size_t)(d + 1) * k, po = k < nvo ? (size_t)d * nvo + k : (size_t)(d + 1) * k + d;
    if (u) for (int i = 0; i < d; i++) u[(size_t)d * v + i] = in[uo + i];
    if (p) p[v] = in[po];
  }
}
static int download_vec(cfdh_ctx *c, const double *dev, std::vector<double> &h) {
  double *st;
  CHK(stage_buffer(c, &st));
  h.resize((size_t)c->NL);
  HIPCHK(c, hipMemcpyAsync(st, dev, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  memcpy(h.data(), st, sizeof(double) * h.size());
  return 0;
}
static int upload_vec(cfdh_ctx *c, const std::vector<double> &h, double *dev) {
  double *st;
  CHK(stage_buffer(c, &st));
  memcpy(st, h.data(), sizeof(double) * h.size());
  HIPCHK(c, hipMemcpyAsync(dev, st, sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int cfdh_set_state(cfdh_ctx *c, const double *u_prev, const double *p_prev, const double *u, const double *p) {
  ENTER(c);
  std::vector<double> h, cur;
  if (u_prev || p_prev) {
    const std::vector<double> *keep = nullptr;
    if (!(u_prev && p_prev)) { CHK(download_vec(c, c->xprev.p, cur)); keep = &cur; }
    pack_vec(c, u_prev, p_prev, h, keep);
    CHK(upload_vec(c, h, c->xprev.p));
    if (u_prev) c->mom_valid = false;
  }
  if (u || p) {
    const std::vector<double> *keep = nullptr;
    if (!(u && p)) { CHK(download_vec(c, c->x.p, cur)); keep = &cur; }
    pack_vec(c, u, p, h, keep);
    CHK(upload_vec(c, h, c->x.p));
  }
  c->state_set = true;
  return 0;
}

int cfdh_get_solution(cfdh_ctx *c, double *u, double *p) {
  ENTER(c);
  std::vector<double> h;
  CHK(download_vec(c, c->x.p, h));
  unpack_vec(c, h, u, p);
  return 0;
}

int cfdh_get_previous(cfdh_ctx *c, double *u, double *p) {
  ENTER(c);
  std::vector<double> h;
  CHK(download_vec(c, c->xprev.p, h));
  unpack_vec(c, h, u, p);
  return 0;
}

int cfdh_get_residual(cfdh_ctx *c, double *ru, double *rp) {
  ENTER(c);
  std::vector<double> h;
  CHK(download_vec(c, c->F.p, h));
  // ghost entries of F are not defined: report zeros there
  for (size_t k = (size_t)(c->dim + 1) * c->nvo; k < h.size(); k++) h[k] = 0.0;
  unpack_vec(c, h, ru, rp);
  return 0;
}

int cfdh_advance(cfdh_ctx *c) {
  ENTER(c);
  CHK(v_copy(c, c->NL, c->x.p, c->xprev.p));
  c->mom_valid = false;
  return 0;
}

int cfdh_advance_field(cfdh_ctx *c, int field) {
  ENTER(c);
  const size_t nu = (size_t)c->dim * c->nvo;
  if (field == 0) {
    CHK(v_copy(c, (int)nu, c->x.p, c->xprev.p));  // ghost entries follow with the halo exchange before the next assembly
    c->mom_valid = false;
  } else if (field == 1) {
    CHK(v_copy(c, c->nvo, c->x.p + nu, c->xprev.p + nu));
  } else {
    return cfdh_fail(c, CFDH_E_ARG, "cfdh_advance_field: field must be 0 (velocity) or 1 (pressure)");
  }
  return 0;
}

int cfdh_set_time_scheme(cfdh_ctx *c, double theta, double a0, double a1, double a2) {
  ENTER(c);
  if (!(theta > 0) || theta > 1 || !(a0 > 0)) return cfdh_fail(c, CFDH_E_ARG, "time scheme needs 0 < theta <= 1 and a0 > 0");
  if (theta != c->ts_theta || a0 != c->ts_a[0]) c->pc_valid = false;  // a0/(theta dt) scales the preconditioner
  if (theta != c->ts_theta || a0 != c->ts_a[0] || a1 != c->ts_a[1] || a2 != c->ts_a[2]) c->jac_valid = false;
  c->ts_theta = theta; c->ts_a[0] = a0; c->ts_a[1] = a1; c->ts_a[2] = a2;
  return 0;
}

// per-cell facet bits of the assembly kernel: bits 0..d exterior facet, bits d+1..2d+1 backflow facet (marker == bf_marker)
static int upload_cell_facet_flags(cfdh_ctx *c) {
  if (c->gen) {
    std::vector<unsigned short> gf((size_t)c->nc, 0);
    for (int k = 0; k < c->nfac; k++) {
      gf[c->fac_cell[k]] |= (unsigned short)(1u << c->fac_local[k]);
      if (c->bf_marker >= 0 && c->fac_marker[k] == c->bf_marker) gf[c->fac_cell[k]] |= (unsigned short)(256u << c->fac_local[k]);
    }
    HIPCHK(c, c->gflag.upload(gf, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->jac_valid = false;
    c->pc_valid = false;
    return 0;
  }
  std::vector<unsigned char> cflag((size_t)c->nc, 0);
  for (int k = 0; k < c->nfac; k++) {
    cflag[c->fac_cell[k]] |= (unsigned char)(1u << c->fac_local[k]);
    if (c->bf_marker >= 0 && c->fac_marker[k] == c->bf_marker)
      cflag[c->fac_cell[k]] |= (unsigned char)((c->dim == 3 ? 16u : 8u) << c->fac_local[k]);
  }
  HIPCHK(c, c->cflag.upload(cflag, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->jac_valid = false;
  c->pc_valid = false;
  return 0;
}

int cfdh_set_boundary_terms(cfdh_ctx *c, int ds_terms, int backflow_marker, double beta) {
  ENTER(c);
  if (beta < 0) return cfdh_fail(c, CFDH_E_ARG, "backflow beta must be >= 0");
  if (!(beta > 0)) backflow_marker = -1;
  const bool changed = (ds_terms != 0) != c->ds_terms || beta != c->bf_beta || backflow_marker != c->bf_marker;
  c->ds_terms = ds_terms != 0; c->bf_beta = beta; c->bf_marker = backflow_marker;
  if (!changed) return 0;
  return upload_cell_facet_flags(c);
}

int cfdh_set_facet_markers(cfdh_ctx *c, int64_t nfacets, const int32_t *markers) {
  ENTER(c);
  if (nfacets != c->nfac_user) return cfdh_fail(c, CFDH_E_ARG, "cfdh_set_facet_markers: %lld markers for the %d exterior facets of cfdh_create", (long long)nfacets, c->nfac_user);
  if (nfacets > 0 && !markers) return cfdh_fail(c, CFDH_E_ARG, "null marker array");
  for (int k = 0; k < c->nfac; k++) c->fac_marker[k] = markers[c->fac_user[k]];
  if (c->nfac) {
    HIPCHK(c, c->d_fac_marker.upload(c->fac_marker, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  // the backflow term follows the outlet marker: refresh the per-cell flags (invalidates Jacobian and preconditioner)
  if (c->bf_marker >= 0) return upload_cell_facet_flags(c);
  return 0;
}

int cfdh_set_previous2(cfdh_ctx *c, const double *u_prev2) {
  ENTER(c);
  if (!u_prev2) return cfdh_fail(c, CFDH_E_ARG, "u_prev2 is NULL");
  std::vector<double> h;
  pack_vec(c, u_prev2, nullptr, h, nullptr);
  CHK(upload_vec(c, h, c->xprev2.p));
  return 0;
}

int cfdh_get_previous2(cfdh_ctx *c, double *u_prev2) {
  ENTER(c);
  std::vector<double> h;
  CHK(download_vec(c, c->xprev2.p, h));
  unpack_vec(c, h, u_prev2, nullptr);
  return 0;
}

int cfdh_shift_history(cfdh_ctx *c) {
  ENTER(c);
  CHK(v_copy(c, c->NL, c->xprev.p, c->xprev2.p));
  return 0;
}

int cfdh_assemble(cfdh_ctx *c, int want_jacobian) {
  ENTER(c);
  if (!c->params_set) return cfdh_fail(c, CFDH_E_STATE, "cfdh_set_params was not called");
  CHK(cfdh_prepare_assembly(c));
  CHK(comm_halo(c, c->x.p));
  CHK(k_assemble(c, c->x.p, want_jacobian ? 1 : 2));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

int cfdh_get_csr(cfdh_ctx *c, int64_t *nnz, int32_t *rowptr, int32_t *col, double *vals) {
  if (!c || !nnz) return CFDH_E_ARG;
  *nnz = (long long)(c->dim + 1) * (c->dim + 1) * c->nnzv;
  if (!rowptr && !col && !vals) return 0;
  if (!rowptr || !col || !vals) return cfdh_fail(c, CFDH_E_ARG, "pass all of rowptr/col/vals or none");
  if (!c->jac_valid) return cfdh_fail(c, CFDH_E_STATE, "no Jacobian assembled yet");
  std::vector<double> a00, a01, a10, a11;
  CHK(cfdh_download_blocks(c, a00, a01, a10, a11));
  const int nvo = c->nvo, nv = c->nv, d = c->dim;
  // rows in user numbering: u rows d*v+i (v < nvo), then p rows d*nvo+v
  int64_t pos = 0;
  std::vector<std::pair<int, int>> ord;  // (user column vertex, slot k)
  auto emit_rows = [&](int kind) {
    for (int vu = 0; vu < nvo; vu++) {
      const int r = c->perm[vu];
      ord.clear();
      for (int k = c->h_vptr[r]; k < c->h_vptr[r + 1]; k++) ord.push_back({c->iperm[c->h_vcol[k]], k});
      std::sort(ord.begin(), ord.end());
      const int nrow = kind == 0 ? d : 1;
      for (int i = 0; i < nrow; i++) {
        const int row = kind == 0 ? d * vu + i : d * nvo + vu;
        rowptr[row] = (int32_t)pos;
        for (auto &e : ord)
          for (int j = 0; j < d; j++) {
            col[pos] = d * e.first + j;
            vals[pos] = kind == 0 ? a00[(size_t)d * d * e.second + (size_t)d * i + j] : a10[(size_t)d * e.second + j];
            pos++;
          }
        for (auto &e : ord) {
          col[pos] = d * nv + e.first;
          vals[pos] = kind == 0 ? a01[(size_t)d * e.second + i] : a11[e.second];
          pos++;
        }
      }
    }
  };
  emit_rows(0);
  emit_rows(1);
  rowptr[(d + 1) * nvo] = (int32_t)pos;
  return 0;
}

int cfdh_spmv(cfdh_ctx *c, const double *x, double *y) {
  if (!c || !x || !y) return CFDH_E_ARG;
  if (!c->jac_valid) return cfdh_fail(c, CFDH_E_STATE, "no Jacobian assembled yet");
  std::vector<double> h;
  pack_vec(c, x, x + (size_t)c->dim * c->nv, h, nullptr);
  CHK(upload_vec(c, h, c->xt.p));
  CHK(k_spmv_full(c, c->xt.p, c->dvec.p));
  std::vector<double> o;
  CHK(download_vec(c, c->dvec.p, o));
  for (int k = 0; k < c->nvo; k++) {
    const int v = c->iperm[k], d = c->dim;
    for (int i = 0; i < d; i++) y[(size_t)d * v + i] = o[(size_t)d * k + i];
    y[(size_t)d * c->nvo + v] = o[(size_t)d * c->nvo + k];
  }
  return 0;
}

int cfdh_set_global_pressure_space(cfdh_ctx *c, int64_t nvg, int64_t ncg, const int32_t *cells, const double *coords,
                                   const int32_t *owned_global, int64_t n_pbc, const int32_t *pbc_nodes) {
  ENTER(c);
  if (nvg <= 0 || ncg <= 0 || !cells || !coords || !owned_global || (n_pbc > 0 && !pbc_nodes))
    return cfdh_fail(c, CFDH_E_ARG, "bad global pressure space arguments");
  const int n = (int)nvg;
  const int D = c->dim, NLc = D + 1;  // vertex records of the halo vectors: D + 1 doubles
  const int NCc = c->gen ? c->nloc : D + 1;  // global cells [ncg][nodes per cell], coords [nvg][D]
  for (int64_t k = 0; k < (int64_t)NCc * ncg; k++) if (cells[k] < 0 || cells[k] >= n) return cfdh_fail(c, CFDH_E_ARG, "global cell vertex out of range");
  std::vector<unsigned char> pbc(n, 0);
  for (int64_t k = 0; k < n_pbc; k++) {
    if (pbc_nodes[k] < 0 || pbc_nodes[k] >= n) return cfdh_fail(c, CFDH_E_ARG, "global Dirichlet node out of range");
    pbc[pbc_nodes[k]] = 1;
  }
  // global P1 stiffness as (row, col, value) triplets -> CSR with Dirichlet rows/cols removed
  std::vector<std::vector<std::pair<int, double>>> rows(n);
  if (c->gen) {  // P2 / Q1: the element's own stiffness by quadrature (csrc/cfdh_gen.hip), as h_Lval holds it for the local part
    std::vector<double> K((size_t)NCc * NCc);
    for (int64_t e = 0; e < ncg; e++) {
      const int32_t *v = cells + (size_t)NCc * e;
      CHK(D == 3 ? cfdh_gen3_element_stiffness(c, v, coords, K.data()) : cfdh_gen_element_stiffness(c, v, coords, K.data()));
      for (int a = 0; a < NCc; a++) {
        if (pbc[v[a]]) continue;
        for (int b = 0; b < NCc; b++) if (!pbc[v[b]]) rows[v[a]].push_back({v[b], K[(size_t)a * NCc + b]});
      }
    }
  }
  for (int64_t e = 0; e < ncg && D == 3 && !c->gen; e++) {
    // tetrahedra: rows of the inverse of [x1-x0 | x2-x0 | x3-x0] are grad lambda_1..3, volume |det| / 6
    const int32_t *v = cells + 4 * e;
    double d[3][3], cr[3][3], g[4][3];
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 3; i++) d[a][i] = coords[3 * (size_t)v[a + 1] + i] - coords[3 * (size_t)v[0] + i];
    for (int a = 0; a < 3; a++) {
      const double *p = d[(a + 1) % 3], *q = d[(a + 2) % 3];
      cr[a][0] = p[1] * q[2] - p[2] * q[1]; cr[a][1] = p[2] * q[0] - p[0] * q[2]; cr[a][2] = p[0] * q[1] - p[1] * q[0];
    }
    const double det = d[0][0] * cr[0][0] + d[0][1] * cr[0][1] + d[0][2] * cr[0][2], vol = std::fabs(det) / 6.0;
    if (!(vol > 0)) return cfdh_fail(c, CFDH_E_ARG, "zero-volume global cell");
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 3; i++) g[a + 1][i] = cr[a][i] / det;
    for (int i = 0; i < 3; i++) g[0][i] = -(g[1][i] + g[2][i] + g[3][i]);
    for (int a = 0; a < 4; a++) {
      if (pbc[v[a]]) continue;
      for (int b = 0; b < 4; b++) {
        if (pbc[v[b]]) continue;
        rows[v[a]].push_back({v[b], vol * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2])});
      }
    }
  }
  for (int64_t e = 0; e < ncg && D == 2 && !c->gen; e++) {
    const int32_t *v = cells + 3 * e;
    const double x0 = coords[2 * v[0]], y0 = coords[2 * v[0] + 1], x1 = coords[2 * v[1]], y1 = coords[2 * v[1] + 1],
                 x2 = coords[2 * v[2]], y2 = coords[2 * v[2] + 1];
    const double det = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0), area = 0.5 * std::fabs(det);
    if (!(area > 0)) return cfdh_fail(c, CFDH_E_ARG, "zero-area global cell");
    const double g[3][2] = {{(y1 - y2) / det, (x2 - x1) / det}, {(y2 - y0) / det, (x0 - x2) / det}, {(y0 - y1) / det, (x1 - x0) / det}};
    for (int a = 0; a < 3; a++) {
      if (pbc[v[a]]) continue;
      for (int b = 0; b < 3; b++) {
        if (pbc[v[b]]) continue;
        rows[v[a]].push_back({v[b], area * (g[a][0] * g[b][0] + g[a][1] * g[b][1])});
      }
    }
  }
  CsrHost &L = c->gp_L;
  L.n = L.m = n; L.rowptr.assign(n + 1, 0); L.col.clear(); L.val.clear();
  for (int i = 0; i < n; i++) {
    if (pbc[i]) { L.col.push_back(i); L.val.push_back(1.0); }
    else {
      auto &r = rows[i];
      std::sort(r.begin(), r.end());
      for (size_t k = 0; k < r.size(); k++) {
        if (k > 0 && r[k].first == r[k - 1].first) L.val.back() += r[k].second;
        else { L.col.push_back(r[k].first); L.val.push_back(r[k].second); }
      }
    }
    L.rowptr[i + 1] = (int)L.col.size();
  }
  c->gp_singular = (n_pbc == 0);
  std::vector<int> l2g(c->nvo);
  for (int k = 0; k < c->nvo; k++) {
    const int g = owned_global[c->iperm[k]];
    if (g < 0 || g >= n) return cfdh_fail(c, CFDH_E_ARG, "owned_global out of range");
    l2g[k] = g;
  }
  HIPCHK(c, c->gp_l2g.upload(l2g, c->stream));
  HIPCHK(c, c->gp_rhs.alloc(n)); HIPCHK(c, c->gp_sol.alloc(n));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // global ids of all local vertices (ghost ids through one halo exchange) and the inverse map: the overlapping
  // velocity preconditioner matches the matrix rows it receives from the owners of its ghosts by global id
  c->h_gid.clear(); c->h_g2l.clear();
  c->rasp.ready = false;
  if (c->nranks > 1 && c->ng > 0 && c->nnbr > 0) {
    std::vector<double> hv((size_t)c->NL, -1.0);
    for (int k = 0; k < c->nvo; k++) hv[(size_t)D * k] = (double)l2g[k];  // first velocity slot of the vertex record
    if (!c->pcw.p) HIPCHK(c, c->pcw.alloc(c->NL));
    HIPCHK(c, c->pcw.upload(hv, c->stream));
    CHK(comm_halo(c, c->pcw.p));
    HIPCHK(c, hipMemcpyAsync(hv.data(), c->pcw.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, c->pcw.zero(c->stream));
    c->h_gid.assign(c->nv, -1);
    c->h_g2l.assign(n, -1);
    bool ok = true;
    for (int k = 0; k < c->nvo; k++) c->h_gid[k] = l2g[k];
    for (int i = 0; i < c->ng; i++) {
      const double gd = hv[(size_t)NLc * c->nvo + (size_t)NLc * i];
      if (!(gd >= 0 && gd < n)) { ok = false; break; }
      c->h_gid[c->nvo + i] = (int)gd;
    }
    if (ok) for (int k = 0; k < c->nv; k++) c->h_g2l[c->h_gid[k]] = k;
    else { c->h_gid.clear(); c->h_g2l.clear(); }
  }
  c->gp_allgather = false;
  if (c->nranks > 1 && c->nccl_comm) {
    // gather plan.  One all-reduce of an owner map (rank+1 at the owned global ids) tells every rank who owns
    // what; parts send their owned values in ascending global-id order, so the position of global vertex g in
    // the gathered buffer follows from the map alone.
    std::vector<double> own(n, 0.0);
    for (int k = 0; k < c->nvo; k++) own[l2g[k]] = (double)(c->rank + 1);
    HIPCHK(c, c->gp_rhs.upload(own, c->stream));
    CHK(comm_allreduce_dev(c, c->gp_rhs.p, n, 0));
    HIPCHK(c, hipMemcpyAsync(own.data(), c->gp_rhs.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<int> cnt(c->nranks, 0), src(n);
    for (int g = 0; g < n; g++) {
      const int r = (int)own[g] - 1;
      if (own[g] != (double)(r + 1) || r < 0 || r >= c->nranks)
        return cfdh_fail(c, CFDH_E_ARG, "global vertex %d is owned by %g parts (expected exactly one)", g, own[g]);
      cnt[r]++;
    }
    int maxcnt = 0;
    for (int r = 0; r < c->nranks; r++) maxcnt = std::max(maxcnt, cnt[r]);
    if (cnt[c->rank] != c->nvo) return cfdh_fail(c, CFDH_E_STATE, "owner map disagrees with nv_owned");
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int g = 0; g < n; g++) { const int r = (int)own[g] - 1; src[g] = r * maxcnt + cnt[r]++; }
    std::vector<std::pair<int, int>> byg(c->nvo);
    for (int k = 0; k < c->nvo; k++) byg[k] = {l2g[k], k};
    std::sort(byg.begin(), byg.end());
    std::vector<int> sidx(c->nvo);
    for (int k = 0; k < c->nvo; k++) sidx[k] = byg[k].second;
    HIPCHK(c, c->gp_send_idx.upload(sidx, c->stream));
    HIPCHK(c, c->gp_src_idx.upload(src, c->stream));
    HIPCHK(c, c->gp_sendbuf.alloc((size_t)maxcnt)); HIPCHK(c, c->gp_recvbuf.alloc((size_t)maxcnt * c->nranks));
    HIPCHK(c, c->gp_sendbuf.zero(c->stream));  // the padding beyond nv_owned stays zero
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->gp_maxcnt = maxcnt;
    // known-answer pass through the gather: every rank sends the global ids of its owned vertices and must find
    // g at position g afterwards; otherwise the all-reduce of the padded vector stays in charge
    {
      std::vector<double> ids((size_t)maxcnt, 0.0), back(n);
      for (int k = 0; k < c->nvo; k++) ids[k] = (double)byg[k].first;
      HIPCHK(c, c->gp_sendbuf.upload(ids, c->stream));
      CHK(comm_allgather_dev(c, c->gp_sendbuf.p, c->gp_recvbuf.p, maxcnt));
      CHK(k_gather_global(c, n, c->gp_src_idx.p, c->gp_recvbuf.p, c->gp_rhs.p));
      HIPCHK(c, hipMemcpyAsync(back.data(), c->gp_rhs.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, c->gp_sendbuf.zero(c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      bool ok = true;
      for (int g = 0; g < n && ok; g++) ok = back[g] == (double)g;
      double bad = ok ? 0.0 : 1.0;  // every rank takes the same decision
      HIPCHK(c, hipMemcpyAsync(c->red_out.p + 16, &bad, sizeof(double), hipMemcpyHostToDevice, c->stream));
      CHK(comm_allreduce_dev(c, c->red_out.p + 16, 1, 1));
      HIPCHK(c, hipMemcpyAsync(&bad, c->red_out.p + 16, sizeof(double), hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));
      c->gp_allgather = bad == 0.0;
      if (!c->gp_allgather && c->rank == 0)
        fprintf(stderr, "[cfdh] WARNING: all-gather self-check failed; the pressure right-hand side is all-reduced instead\n");
    }
  }
  c->gp_n = n;
  c->gp_dirty = true;
  c->pc_valid = false;
  return 0;
}

int cfdh_solve_step(cfdh_ctx *c, cfdh_stats *stats) {
  ENTER(c);
  cfdh_stats st;
  c->err.clear();
  int rc = cfdh_newton_step(c, &st);
  if (stats) *stats = st;
  return rc;
}

int cfdh_functional(cfdh_ctx *c, int kind, int marker, double *out) {
  if (!out) return CFDH_E_ARG;
  ENTER(c);
  return k_functional(c, kind, marker, out);
}

int cfdh_wall_shear_stress(cfdh_ctx *c, double *shear) {
  ENTER(c);
  const size_t d = (size_t)c->dim;
  if (!c->wss.p) HIPCHK(c, c->wss.alloc(d * (size_t)c->nv));
  CHK(comm_halo(c, c->x.p));
  CHK(k_wss(c, c->wss.p));
  if (!shear) return 0;
  std::vector<double> h(d * (size_t)c->nv);
  HIPCHK(c, hipMemcpyAsync(h.data(), c->wss.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < c->nv; k++) {
    const int v = c->iperm[k];
    for (size_t i = 0; i < d; i++) shear[d * (size_t)v + i] = h[d * (size_t)k + i];
  }
  return 0;
}

int cfdh_profile_enable(cfdh_ctx *c, int on) {
  ENTER(c);
  prof_flush(c);
  c->prof_on = on != 0;
  if (c->prof_on) {
    // calibration: the elapsed time of an EMPTY event pair on this stream (kind 7).  Every measured launch
    // carries this much in excess of the kernel's own duration; callers may subtract the average.
    for (int i = 0; i < 256; i++) { prof_begin(c, 7); prof_end(c, 7); }
    prof_flush(c);
  }
  return 0;
}
int cfdh_profile_get(cfdh_ctx *c, int kind, double *total_ms, int64_t *launches) {
  if (!c || kind < 0 || kind >= 12) return CFDH_E_ARG;
  prof_flush(c);
  if (total_ms) *total_ms = c->prof[kind].total_ms;
  if (launches) *launches = c->prof[kind].launches;
  return 0;
}
int cfdh_profile_reset(cfdh_ctx *c) {
  ENTER(c);
  prof_flush(c);
  for (auto &p : c->prof) { p.total_ms = 0; p.launches = 0; }
  c->n_allreduce = c->n_halo = c->n_host_sync = c->n_krylov = c->n_allgather = 0;
  c->n_krylov_discarded = 0;
  return 0;
}

int64_t cfdh_info(const cfdh_ctx *c, int what) {
  if (!c) return -1;
  switch (what) {
    case 0: return c->nvo;
    case 1: return c->nv;
    case 2: return c->nc;
    case 3: return c->nnzv;
    case 4: return c->opt.pc_type == 1 ? c->hL.fine_nnz : c->hS.fine_nnz;
    case 5: return c->ninc;
    case 6: return (int64_t)(c->opt.pc_type == 1 ? c->hL.lev.size() : c->hS.lev.size());
    case 7: return c->nblk;
    case 8: return c->hA.fine_nnz;
    case 9: return c->gp_allgather ? c->gp_maxcnt : 0;
    case 10: return c->nccl_comm ? 1 : 0;
    case 11: return c->dl0.on ? c->dl0.n1 : 0;
    case 12: return c->ras ? 1 : 0;
    case 13: return c->n_allreduce;
    case 14: return c->n_halo;
    case 15: return c->n_host_sync;
    case 16: return c->n_krylov;
    case 17: return c->n_allgather;
    case 18: return c->nranks;
    case 26: return c->dim;
    case 30: case 31: case 32: case 33: case 34: case 35: case 36: case 37: case 38: case 39:
    case 40: case 41: case 42: case 43: case 44: case 45: case 46: case 47: case 48: case 49:
    case 50: case 51: case 52: case 53: case 54: case 55: case 56: case 57: case 58: case 59:
    case 60: case 61: case 62: case 63: case 64: case 65: case 66: case 67: case 68: case 69: {
      const AmgHier &h = what < 50 ? c->hA : (c->opt.pc_type == 1 ? c->hL : c->hS);
      const size_t l = (size_t)(what % 10);
      if (l >= h.lev.size()) return 0;
      return ((what / 10) & 1) ? (int64_t)h.lev[l]->n : (int64_t)h.lev[l]->A.nnz;
    }
    case 28: return c->etype;
    case 70: return c->n_guess_solves;  // linear solves started from a projected guess (cfdh_options.ksp_guess)
    case 71: return c->n_guess_solves ? (int64_t)(1e6 * c->guess_reduction_sum / (double)c->n_guess_solves) : 0;  // mean |r0| / |b| of those, in 1e-6
    case 29: return c->nloc;
    case 73: return c->n_krylov_discarded;  // FGMRES iterations launched ahead of the host's convergence test and discarded (not in krylov_its)
    case 72: return c->n_attainable_stops;  // solves stopped at the attainable accuracy (reason CFDH_KSP_CONVERGED_ATTAINABLE), above their tolerance
    case 27: return (int64_t)(1000.0 * c->ms_pc_build_dev);  // microseconds of the last device-side preconditioner build (0: host build)
    case 19: return c->opt.pc_type == 1 ? c->hL.nnz_S0 : c->hS.nnz_S0;
    case 20: return c->hA.nnz_S0;
    case 21: return c->opt.pc_type == 1 ? c->hL.nnz_G0 : c->hS.nnz_G0;
    case 22: return c->hA.nnz_G0;
    case 23: { const AmgHier &h = c->opt.pc_type == 1 ? c->hL : c->hS; return h.lev.size() > 1 ? h.lev[1]->n : 0; }
    case 24: return c->hA.lev.size() > 1 ? c->hA.lev[1]->n : 0;
    case 25: { const AmgHier &h = c->opt.pc_type == 1 ? c->hL : c->hS; return h.fused ? 1 : 0; }
    default: return -1;
  }
}

}  // extern "C"
