// Host-side setup of a tetrahedral context (gdim == 3): internal vertex numbering along a 3-D Morton curve, the
// vertex-graph pattern of the 4x4 block matrix (create_matrix_block, stabilized_schur.py:191, on the P1/P1 spaces of
// :55-57 with gdim 3), the incidence lists of the LDS-accumulating assembly kernel, P1 stiffness / lumped mass for the
// Cahouet-Chabard preconditioner.  Single GPU: nv_owned must equal nv.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "cfdh_internal.hpp"

static inline uint64_t spread3(uint64_t x) {  // 21 bits -> every third bit
  x &= 0x1fffff;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}

int cfdh_build_mesh3(cfdh_ctx *c, int64_t nv64, int64_t nvo64, int64_t nc64, const int32_t *cells, const double *coords,
                     int64_t nfac64, const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker) {
  const int nv = (int)nv64, nvo = (int)nvo64, ncu = (int)nc64, nfac = (int)nfac64;
  if (nv <= 0 || nvo <= 0 || nvo > nv || ncu <= 0) return cfdh_fail(c, CFDH_E_ARG, "bad mesh sizes");
  if (nv64 > (1ll << 28) || nc64 > (1ll << 29)) return cfdh_fail(c, CFDH_E_ARG, "mesh too large for int32 indexing");
  for (int64_t k = 0; k < 4 * nc64; k++)
    if (cells[k] < 0 || cells[k] >= nv) return cfdh_fail(c, CFDH_E_ARG, "cell vertex index out of range");
  for (int k = 0; k < nfac; k++)
    if (fcell[k] < 0 || fcell[k] >= ncu || flocal[k] < 0 || flocal[k] > 3) return cfdh_fail(c, CFDH_E_ARG, "facet (cell, local) out of range");
  c->dim = 3; c->nloc = 4;
  // a part of a partitioned mesh: owned vertices first, ghosts after (as in 2-D); vectors [u owned 3 nvo | p owned nvo |
  // ghosts (ux, uy, uz, p) ng]
  c->nv = nv; c->nvo = nvo; c->ng = nv - nvo;
  c->NO = 4 * nvo; c->NL = 4 * nvo + 4 * c->ng;
  // ---- Morton numbering of the owned vertices (ghosts keep their order: grouped by owner)
  c->perm.resize(nv); c->iperm.resize(nv);
  {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int v = 0; v < nv; v++)
      for (int i = 0; i < 3; i++) { lo[i] = std::min(lo[i], coords[3 * v + i]); hi[i] = std::max(hi[i], coords[3 * v + i]); }
    const double ext = std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2]));
    if (!(ext > 0)) return cfdh_fail(c, CFDH_E_ARG, "degenerate coordinates");
    std::vector<uint64_t> key(nvo);
    for (int v = 0; v < nvo; v++) {
      uint64_t q[3];
      for (int i = 0; i < 3; i++) q[i] = (uint64_t)std::min(2097151.0, (coords[3 * v + i] - lo[i]) / ext * 2097151.0);
      key[v] = spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
    }
    std::vector<int> order(nvo);
    std::iota(order.begin(), order.end(), 0);
    const char *nr = getenv("CFDH_NO_RENUMBER");
    if (!(nr && nr[0] == '1')) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    for (int k = 0; k < nvo; k++) { c->iperm[k] = order[k]; c->perm[order[k]] = k; }
    for (int v = nvo; v < nv; v++) { c->iperm[v] = v; c->perm[v] = v; }
  }
  c->h_coords.resize(3 * (size_t)nv);
  for (int k = 0; k < nv; k++)
    for (int i = 0; i < 3; i++) c->h_coords[3 * (size_t)k + i] = coords[3 * (size_t)c->iperm[k] + i];
  // ---- cells in internal ids, sorted by smallest vertex, positively oriented
  {
    std::vector<std::pair<int, int>> keyed;
    keyed.reserve(ncu);
    for (int e = 0; e < ncu; e++) {
      int mn = nv;
      for (int a = 0; a < 4; a++) mn = std::min(mn, c->perm[cells[4 * e + a]]);
      if (mn < nvo) keyed.push_back({mn, e});  // cells touching an owned vertex
    }
    std::stable_sort(keyed.begin(), keyed.end());
    const int nck = (int)keyed.size();
    c->nc = nck;
    c->h_cells.resize(4 * (size_t)nck);
    c->cell_user.resize(nck);
    std::vector<int> cmap(ncu, -1);
    std::vector<unsigned char> flipped(nck, 0);
    const double *X = c->h_coords.data();
    for (int k = 0; k < nck; k++) {
      const int e = keyed[k].second;
      cmap[e] = k; c->cell_user[k] = e;
      int *v = &c->h_cells[4 * (size_t)k];
      for (int a = 0; a < 4; a++) v[a] = c->perm[cells[4 * e + a]];
      for (int a = 0; a < 4; a++)
        for (int b = a + 1; b < 4; b++)
          if (v[a] == v[b]) return cfdh_fail(c, CFDH_E_ARG, "degenerate cell %d", e);
      double d[3][3];
      for (int a = 0; a < 3; a++)
        for (int i = 0; i < 3; i++) d[a][i] = X[3 * (size_t)v[a + 1] + i] - X[3 * (size_t)v[0] + i];
      const double det = d[0][0] * (d[1][1] * d[2][2] - d[1][2] * d[2][1]) - d[0][1] * (d[1][0] * d[2][2] - d[1][2] * d[2][0]) +
                         d[0][2] * (d[1][0] * d[2][1] - d[1][1] * d[2][0]);
      if (!(std::fabs(det) > 0)) return cfdh_fail(c, CFDH_E_ARG, "zero-volume cell %d", e);
      if (det < 0) { std::swap(v[2], v[3]); flipped[k] = 1; }
    }
    c->fac_cell.clear(); c->fac_local.clear(); c->fac_marker.clear(); c->fac_user.clear();
    c->nfac_user = nfac;
    for (int k = 0; k < nfac; k++) {
      const int e = cmap[fcell[k]];
      if (e < 0) continue;
      int fl = flocal[k];
      if (flipped[e] && fl >= 2) fl = 5 - fl;  // local vertices 2 and 3 were swapped
      c->fac_cell.push_back(e); c->fac_local.push_back(fl);
      c->fac_marker.push_back(fmarker ? fmarker[k] : 0);
      c->fac_user.push_back(k);
    }
    c->nfac = (int)c->fac_cell.size();
  }
  const int nc = c->nc;
  // ---- vertex -> incident (cell, local); vertex graph
  std::vector<int> vcptr(nvo + 1, 0);
  for (size_t k = 0; k < 4 * (size_t)nc; k++) if (c->h_cells[k] < nvo) vcptr[c->h_cells[k] + 1]++;
  for (int v = 0; v < nvo; v++) vcptr[v + 1] += vcptr[v];
  const int ninc = vcptr[nvo];
  std::vector<int> vcell(ninc);
  {
    std::vector<int> fill(nvo, 0);
    for (int e = 0; e < nc; e++)
      for (int a = 0; a < 4; a++) { const int v = c->h_cells[4 * (size_t)e + a]; if (v < nvo) vcell[vcptr[v] + fill[v]++] = 4 * e + a; }
  }
  c->h_vptr.assign(nvo + 1, 0);
  c->h_vcol.clear(); c->h_vcol.reserve((size_t)16 * nvo);
  c->h_vdiag.resize(nvo);
  {
    std::vector<int> tmp;
    for (int v = 0; v < nvo; v++) {
      if (vcptr[v + 1] == vcptr[v]) return cfdh_fail(c, CFDH_E_ARG, "vertex %d has no cell", c->iperm[v]);
      tmp.clear();
      for (int k = vcptr[v]; k < vcptr[v + 1]; k++) { const int e = vcell[k] >> 2; for (int a = 0; a < 4; a++) tmp.push_back(c->h_cells[4 * (size_t)e + a]); }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      if ((int)tmp.size() > CFDH3_MAX_SLOTS) return cfdh_fail(c, CFDH_E_ARG, "vertex valence %d exceeds %d", (int)tmp.size(), CFDH3_MAX_SLOTS);
      c->h_vdiag[v] = (int)c->h_vcol.size() + (int)(std::lower_bound(tmp.begin(), tmp.end(), v) - tmp.begin());
      c->h_vcol.insert(c->h_vcol.end(), tmp.begin(), tmp.end());
      c->h_vptr[v + 1] = (int)c->h_vcol.size();
    }
  }
  c->nnzv = (int)c->h_vcol.size();
  c->ninc = ninc;
  // ---- P1 stiffness and lumped mass (geometry only)
  c->h_Lval.assign(c->nnzv, 0.0);
  c->h_Ml.assign(nvo, 0.0);
  for (int e = 0; e < nc; e++) {
    const int *v = &c->h_cells[4 * (size_t)e];
    const double *X = c->h_coords.data();
    double d[3][3];
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 3; i++) d[a][i] = X[3 * (size_t)v[a + 1] + i] - X[3 * (size_t)v[0] + i];
    // rows of the inverse of [d0 d1 d2]^T-columns matrix = gradients of l_1..l_3: g_a = (d_b x d_c) / det
    double cr[3][3];
    for (int a = 0; a < 3; a++) {
      const double *p = d[(a + 1) % 3], *q = d[(a + 2) % 3];
      cr[a][0] = p[1] * q[2] - p[2] * q[1]; cr[a][1] = p[2] * q[0] - p[0] * q[2]; cr[a][2] = p[0] * q[1] - p[1] * q[0];
    }
    const double det = d[0][0] * cr[0][0] + d[0][1] * cr[0][1] + d[0][2] * cr[0][2];
    double g[4][3];
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 3; i++) g[a + 1][i] = cr[a][i] / det;
    for (int i = 0; i < 3; i++) g[0][i] = -(g[1][i] + g[2][i] + g[3][i]);
    const double vol = std::fabs(det) / 6.0;
    for (int a = 0; a < 4; a++) {
      if (v[a] >= nvo) continue;
      c->h_Ml[v[a]] += vol / 4.0;
      const int *nb = &c->h_vcol[c->h_vptr[v[a]]];
      const int deg = c->h_vptr[v[a] + 1] - c->h_vptr[v[a]];
      for (int b = 0; b < 4; b++) {
        const int sidx = (int)(std::lower_bound(nb, nb + deg, v[b]) - nb);
        c->h_Lval[c->h_vptr[v[a]] + sidx] += vol * (g[a][0] * g[b][0] + g[a][1] * g[b][1] + g[a][2] * g[b][2]);
      }
    }
  }
  // ---- assembly workgroups: consecutive rows, at most CFDH3_MAX_SLOTS value slots
  std::vector<int> blk_row(1, 0), blk_iptr(1, 0), inc_cell, inc_row;
  std::vector<unsigned long long> inc_slots;
  inc_cell.reserve(ninc); inc_row.reserve(ninc); inc_slots.reserve(ninc);
  {
    int r0 = 0;
    while (r0 < nvo) {
      int r1 = r0;
      const int s0 = c->h_vptr[r0];
      while (r1 < nvo && c->h_vptr[r1 + 1] - s0 <= CFDH3_MAX_SLOTS && r1 - r0 < 64) r1++;
      if (r1 == r0) return cfdh_fail(c, CFDH_E_ARG, "row too long for one assembly workgroup");
      // The rows of a workgroup are dealt to its four wavefronts (longest first, to the least loaded wavefront: deterministic),
      // and ALL incidences of a row are processed by the wavefront that got it.  Every LDS accumulator (row, column) then
      // receives its contributions from one wavefront only, in program order (and in lane order inside one ds_add_f64
      // instruction): a FIXED order -- the assembly is bitwise reproducible although it still sums in LDS.
      std::vector<int> order(r1 - r0);
      for (int r = r0; r < r1; r++) order[r - r0] = r;
      std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return vcptr[x + 1] - vcptr[x] > vcptr[y + 1] - vcptr[y]; });
      std::vector<int> wrows[4];
      int load[4] = {0, 0, 0, 0};
      for (int r : order) {
        int w = 0;
        for (int q = 1; q < 4; q++) if (load[q] < load[w]) w = q;
        wrows[w].push_back(r);
        load[w] += vcptr[r + 1] - vcptr[r];
      }
      for (int w = 0; w < 4; w++) {
        std::sort(wrows[w].begin(), wrows[w].end());
        for (int r : wrows[w]) {
          const int *nb = &c->h_vcol[c->h_vptr[r]];
          const int deg = c->h_vptr[r + 1] - c->h_vptr[r];
          for (int k = vcptr[r]; k < vcptr[r + 1]; k++) {
            const int e = vcell[k] >> 2;
            unsigned long long sl = 0;
            for (int b = 0; b < 4; b++) {
              const int wv = c->h_cells[4 * (size_t)e + b];
              const unsigned long long sidx = (unsigned long long)(c->h_vptr[r] - s0 + (int)(std::lower_bound(nb, nb + deg, wv) - nb));
              sl |= sidx << (16 * b);
            }
            inc_cell.push_back(vcell[k]); inc_row.push_back(r - r0); inc_slots.push_back(sl);
          }
        }
        if (w < 3) blk_iptr.push_back((int)inc_cell.size());  // [4 blk + w + 1]: end of wavefront w's list
      }
      blk_row.push_back(r1); blk_iptr.push_back((int)inc_cell.size());
      r0 = r1;
    }
  }
  c->a3_nblk = (int)blk_row.size() - 1;
  c->nblk = c->a3_nblk;
  // ---- uploads / allocations
  hipStream_t s = c->stream;
  std::vector<unsigned char> cflag(nc, 0);
  for (int k = 0; k < c->nfac; k++) cflag[c->fac_cell[k]] |= (unsigned char)(1u << c->fac_local[k]);
  HIPCHK(c, c->coords.upload(c->h_coords, s));
  HIPCHK(c, c->cells.upload(c->h_cells, s));
  HIPCHK(c, c->cflag.upload(cflag, s));
  HIPCHK(c, c->mom.alloc(12 * (size_t)nc));
  HIPCHK(c, c->vptr.upload(c->h_vptr, s));
  HIPCHK(c, c->vcol.upload(c->h_vcol, s));
  HIPCHK(c, c->vdiag.upload(c->h_vdiag, s));
  HIPCHK(c, c->A00.alloc(9 * (size_t)c->nnzv));
  HIPCHK(c, c->A01.alloc(3 * (size_t)c->nnzv));
  HIPCHK(c, c->A10.alloc(3 * (size_t)c->nnzv));
  HIPCHK(c, c->A11.alloc((size_t)c->nnzv));
  HIPCHK(c, c->a3_blk_row.upload(blk_row, s));
  HIPCHK(c, c->a3_blk_iptr.upload(blk_iptr, s));
  HIPCHK(c, c->a3_inc_cell.upload(inc_cell, s));
  HIPCHK(c, c->a3_inc_row.upload(inc_row, s));
  HIPCHK(c, c->a3_inc_slots.upload(inc_slots, s));
  // with overlapping parts a cell is integrated (global functionals) only by the rank that owns its first vertex
  std::vector<unsigned char> cown(nc, 1);  // lives until the stream synchronisation at the end of this function
  for (int k = 0; k < nc; k++) cown[k] = cells[4 * (size_t)c->cell_user[k]] < nvo ? 1 : 0;
  HIPCHK(c, c->cell_owned.upload(cown, s));
  if (c->nfac) {
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
  const size_t NL = c->NL;
  HIPCHK(c, c->x.alloc(NL)); HIPCHK(c, c->xt.alloc(NL)); HIPCHK(c, c->xprev.alloc(NL)); HIPCHK(c, c->xprev2.alloc(NL));
  HIPCHK(c, c->F.alloc(NL)); HIPCHK(c, c->dvec.alloc(NL));
  HIPCHK(c, c->x.zero(s)); HIPCHK(c, c->xt.zero(s)); HIPCHK(c, c->xprev.zero(s)); HIPCHK(c, c->xprev2.zero(s)); HIPCHK(c, c->F.zero(s)); HIPCHK(c, c->dvec.zero(s));
  c->red_blocks = 1024;
  HIPCHK(c, c->red_partial.alloc((size_t)c->red_blocks * 260));
  HIPCHK(c, c->red_out.alloc(1024));
  HIPCHK(c, hipHostMalloc((void **)&c->h_pinned, 1024 * sizeof(double)));
  HIPCHK(c, hipHostGetDevicePointer((void **)&c->h_pinned_dev, c->h_pinned, 0));
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_h, hipEventDisableTiming));
  HIPCHK(c, c->pu0.alloc(3 * (size_t)nvo)); HIPCHK(c, c->pu1.alloc(3 * (size_t)nvo)); HIPCHK(c, c->pu2.alloc(3 * (size_t)nvo));
  HIPCHK(c, c->pr.alloc(3 * (size_t)nvo));
  HIPCHK(c, c->pp0.alloc(nvo)); HIPCHK(c, c->pp1.alloc(nvo));
  HIPCHK(c, hipStreamSynchronize(s));
  return 0;
}
