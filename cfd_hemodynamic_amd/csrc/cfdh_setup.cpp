// Host-side, one-off setup of libcfdh.so: internal vertex numbering, the fixed
// vertex-graph CSR pattern (create_matrix_block, stabilized_schur.py:191), the
// (row vertex, cell) incidence lists that drive the atomic-free assembly, and
// the smoothed-aggregation hierarchy for the SELFP Schur matrix
// Sp = A11 - A10 diag(A00)^-1 A01 (stabilized_schur.py:235).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <numeric>

#include "cfdh_internal.hpp"

int cfdh_fail(cfdh_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  extern std::string g_cfdh_last_error;
  g_cfdh_last_error = buf;
  if (c) c->err = buf;
  return code;
}

// host threads for the one-off setup loops: never oversubscribe a cgroup-limited box
int cfdh_host_threads() {
  const char *e = getenv("CFDH_HOST_THREADS");
  int n = e ? atoi(e) : 8;
  return n < 1 ? 1 : (n > 64 ? 64 : n);
}

static inline uint32_t part1by1(uint32_t x) {
  x &= 0x0000ffff;
  x = (x ^ (x << 8)) & 0x00ff00ff;
  x = (x ^ (x << 4)) & 0x0f0f0f0f;
  x = (x ^ (x << 2)) & 0x33333333;
  x = (x ^ (x << 1)) & 0x55555555;
  return x;
}

int cfdh_build_mesh(cfdh_ctx *c, int64_t nv64, int64_t nvo64, int64_t nc64, const int32_t *cells, const double *coords,
                    int64_t nfac64, const int32_t *fcell, const int32_t *flocal, const int32_t *fmarker) {
  const int nv = (int)nv64, nvo = (int)nvo64, ncu = (int)nc64, nfac = (int)nfac64;
  if (nv <= 0 || nvo <= 0 || nvo > nv || ncu <= 0) return cfdh_fail(c, CFDH_E_ARG, "bad mesh sizes");
  if (nv64 > (1ll << 29) || nc64 > (1ll << 29)) return cfdh_fail(c, CFDH_E_ARG, "mesh too large for int32 indexing");
  for (int64_t k = 0; k < 3 * nc64; k++)
    if (cells[k] < 0 || cells[k] >= nv) return cfdh_fail(c, CFDH_E_ARG, "cell vertex index out of range");
  for (int k = 0; k < nfac; k++)
    if (fcell[k] < 0 || fcell[k] >= ncu || flocal[k] < 0 || flocal[k] > 2)
      return cfdh_fail(c, CFDH_E_ARG, "facet (cell, local) out of range");
  c->nv = nv; c->nvo = nvo; c->ng = nv - nvo;
  c->NO = 3 * nvo; c->NL = 3 * nvo + 3 * c->ng;

  // ---- internal numbering: owned vertices along a Morton curve, ghosts unchanged
  c->perm.resize(nv); c->iperm.resize(nv);
  {
    std::vector<int> order(nvo);
    std::iota(order.begin(), order.end(), 0);
    const char *nr = getenv("CFDH_NO_RENUMBER");
    if (!(nr && nr[0] == '1')) {
      double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
      for (int v = 0; v < nv; v++)
        for (int i = 0; i < 2; i++) { lo[i] = std::min(lo[i], coords[2 * v + i]); hi[i] = std::max(hi[i], coords[2 * v + i]); }
      double ext = std::max(hi[0] - lo[0], hi[1] - lo[1]);
      if (!(ext > 0)) return cfdh_fail(c, CFDH_E_ARG, "degenerate coordinates");
      std::vector<uint32_t> key(nvo);
      for (int v = 0; v < nvo; v++) {
        uint32_t qx = (uint32_t)std::min(65535.0, (coords[2 * v] - lo[0]) / ext * 65535.0);
        uint32_t qy = (uint32_t)std::min(65535.0, (coords[2 * v + 1] - lo[1]) / ext * 65535.0);
        key[v] = part1by1(qx) | (part1by1(qy) << 1);
      }
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
    for (int k = 0; k < nvo; k++) { c->iperm[k] = order[k]; c->perm[order[k]] = k; }
    for (int v = nvo; v < nv; v++) { c->iperm[v] = v; c->perm[v] = v; }
  }
  c->h_coords.resize(2 * (size_t)nv);
  for (int k = 0; k < nv; k++) { c->h_coords[2 * k] = coords[2 * c->iperm[k]]; c->h_coords[2 * k + 1] = coords[2 * c->iperm[k] + 1]; }

  // ---- cells touching an owned vertex, internal ids, sorted by smallest vertex
  {
    std::vector<std::pair<int, int>> keyed;
    keyed.reserve(ncu);
    for (int e = 0; e < ncu; e++) {
      int a = c->perm[cells[3 * e]], b = c->perm[cells[3 * e + 1]], d = c->perm[cells[3 * e + 2]];
      if (a == b || b == d || a == d) return cfdh_fail(c, CFDH_E_ARG, "degenerate cell %d", e);
      int mn = std::min(a, std::min(b, d));
      if (mn < nvo) keyed.push_back({mn, e});
    }
    std::stable_sort(keyed.begin(), keyed.end());
    c->nc = (int)keyed.size();
    c->h_cells.resize(3 * (size_t)c->nc);
    c->cell_user.resize(c->nc);
    std::vector<int> cmap(ncu, -1);
    for (int k = 0; k < c->nc; k++) {
      int e = keyed[k].second;
      cmap[e] = k; c->cell_user[k] = e;
      for (int a = 0; a < 3; a++) c->h_cells[3 * k + a] = c->perm[cells[3 * e + a]];
    }
    c->fac_cell.clear(); c->fac_local.clear(); c->fac_marker.clear(); c->fac_user.clear();
    c->nfac_user = nfac;
    for (int k = 0; k < nfac; k++)
      if (cmap[fcell[k]] >= 0) {
        c->fac_cell.push_back(cmap[fcell[k]]); c->fac_local.push_back(flocal[k]);
        c->fac_marker.push_back(fmarker ? fmarker[k] : 0);
        c->fac_user.push_back(k);
      }
    c->nfac = (int)c->fac_cell.size();
  }
  const int nc = c->nc;
  {
    // positive orientation of the internal cells (the assembly kernel walks the cells of a vertex
    // counter-clockwise); a flipped cell swaps its local vertices 1 and 2, and so its local facets
    std::vector<unsigned char> flipped(nc, 0);
    for (int e = 0; e < nc; e++) {
      int *v = &c->h_cells[3 * e];
      const double *X = c->h_coords.data();
      double det = (X[2 * v[1]] - X[2 * v[0]]) * (X[2 * v[2] + 1] - X[2 * v[0] + 1]) - (X[2 * v[1] + 1] - X[2 * v[0] + 1]) * (X[2 * v[2]] - X[2 * v[0]]);
      if (!(std::fabs(det) > 0)) return cfdh_fail(c, CFDH_E_ARG, "zero-area cell %d", c->cell_user[e]);
      if (det < 0) { std::swap(v[1], v[2]); flipped[e] = 1; }
    }
    for (int k = 0; k < c->nfac; k++)
      if (flipped[c->fac_cell[k]] && c->fac_local[k] != 0) c->fac_local[k] = 3 - c->fac_local[k];
  }

  // ---- vertex -> incident (cell, local) for owned rows; vertex graph
  std::vector<int> vcptr(nvo + 1, 0);
  for (int e = 0; e < nc; e++)
    for (int a = 0; a < 3; a++) { int v = c->h_cells[3 * e + a]; if (v < nvo) vcptr[v + 1]++; }
  for (int v = 0; v < nvo; v++) vcptr[v + 1] += vcptr[v];
  const int ninc = vcptr[nvo];
  std::vector<int> vcell(ninc);
  {
    std::vector<int> fill(nvo, 0);
    for (int e = 0; e < nc; e++)
      for (int a = 0; a < 3; a++) { int v = c->h_cells[3 * e + a]; if (v < nvo) vcell[vcptr[v] + fill[v]++] = 4 * e + a; }
  }
  c->h_vptr.assign(nvo + 1, 0);
  c->h_vcol.clear(); c->h_vcol.reserve((size_t)7 * nvo);
  c->h_vdiag.resize(nvo);
  {
    std::vector<int> tmp;
    for (int v = 0; v < nvo; v++) {
      tmp.clear(); tmp.push_back(v);
      for (int k = vcptr[v]; k < vcptr[v + 1]; k++) { int e = vcell[k] >> 2; for (int a = 0; a < 3; a++) tmp.push_back(c->h_cells[3 * e + a]); }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      if ((int)tmp.size() > 255) return cfdh_fail(c, CFDH_E_ARG, "vertex valence > 254 is not supported");
      if (vcptr[v + 1] == vcptr[v]) return cfdh_fail(c, CFDH_E_ARG, "owned vertex %d has no cell", c->iperm[v]);
      c->h_vdiag[v] = (int)c->h_vcol.size() + (int)(std::lower_bound(tmp.begin(), tmp.end(), v) - tmp.begin());
      c->h_vcol.insert(c->h_vcol.end(), tmp.begin(), tmp.end());
      c->h_vptr[v + 1] = (int)c->h_vcol.size();
    }
  }
  c->nnzv = (int)c->h_vcol.size();
  c->ninc = ninc;
  // P1 stiffness (grad l_a . grad l_b) on the vertex graph and lumped mass of the owned rows: geometry only,
  // used by the Cahouet-Chabard Schur approximation (pc_type 1)
  c->h_Lval.assign(c->nnzv, 0.0);
  c->h_Ml.assign(nvo, 0.0);
  for (int e = 0; e < nc; e++) {
    const int *v = &c->h_cells[3 * e];
    const double *X = c->h_coords.data();
    const double x0 = X[2 * v[0]], y0 = X[2 * v[0] + 1], x1 = X[2 * v[1]], y1 = X[2 * v[1] + 1], x2 = X[2 * v[2]], y2 = X[2 * v[2] + 1];
    const double det = (x1 - x0) * (y2 - y0) - (y1 - y0) * (x2 - x0), area = 0.5 * std::fabs(det);
    const double g[3][2] = {{(y1 - y2) / det, (x2 - x1) / det}, {(y2 - y0) / det, (x0 - x2) / det}, {(y0 - y1) / det, (x1 - x0) / det}};
    for (int a = 0; a < 3; a++) {
      if (v[a] >= nvo) continue;
      c->h_Ml[v[a]] += area / 3.0;
      const int *nb = &c->h_vcol[c->h_vptr[v[a]]];
      const int deg = c->h_vptr[v[a] + 1] - c->h_vptr[v[a]];
      for (int b = 0; b < 3; b++) {
        const int sidx = (int)(std::lower_bound(nb, nb + deg, v[b]) - nb);
        c->h_Lval[c->h_vptr[v[a]] + sidx] += area * (g[a][0] * g[b][0] + g[a][1] * g[b][1]);
      }
    }
  }

  // ---- incidences of a row ordered as counter-clockwise fans around the vertex: consecutive
  // lanes then share the edge between their cells, so the two contributions of an off-diagonal
  // block are summed with ONE lane shuffle and the diagonal block with a segmented wave reduction
  // (no LDS accumulation, no atomics, fixed order => bitwise reproducible).
  struct LaneRec { int cellslot; unsigned slots; unsigned seg; unsigned char a, has_prev, emit_v2; };
  std::vector<LaneRec> rowlanes(ninc);  // in row order, position = vcptr[v] + fan position
  {
    std::vector<int> ord, v1s, v2s, used;
    for (int v = 0; v < nvo; v++) {
      const int k0 = vcptr[v], n = vcptr[v + 1] - k0;
      if (n > 64) return cfdh_fail(c, CFDH_E_ARG, "vertex %d has %d cells: more than one wavefront", c->iperm[v], n);
      v1s.assign(n, 0); v2s.assign(n, 0); used.assign(n, 0); ord.clear();
      for (int q = 0; q < n; q++) {
        const int e = vcell[k0 + q] >> 2, a = vcell[k0 + q] & 3;
        v1s[q] = c->h_cells[3 * e + (a + 1) % 3]; v2s[q] = c->h_cells[3 * e + (a + 2) % 3];
      }
      auto find_by_v1 = [&](int w) { for (int q = 0; q < n; q++) if (!used[q] && v1s[q] == w) return q; return -1; };
      auto has_v2 = [&](int w, int self) { for (int q = 0; q < n; q++) if (q != self && v2s[q] == w) return true; return false; };
      std::vector<int> fan_first_pos, fan_len, fan_closed;
      // open fans first (start = cell whose (i,v1) edge has no other cell), then closed ones
      for (int pass = 0; pass < 2; pass++)
        for (int q0 = 0; q0 < n; q0++) {
          if (used[q0]) continue;
          if (pass == 0 && has_v2(v1s[q0], q0)) continue;
          const int first = (int)ord.size();
          int q = q0;
          while (q >= 0) { used[q] = 1; ord.push_back(q); q = find_by_v1(v2s[q]); }
          const int len = (int)ord.size() - first;
          const bool closed = (pass == 1) && v2s[ord.back()] == v1s[ord[first]];
          fan_first_pos.push_back(first); fan_len.push_back(len); fan_closed.push_back(closed ? 1 : 0);
        }
      const int *nb = &c->h_vcol[c->h_vptr[v]];
      const int deg = c->h_vptr[v + 1] - c->h_vptr[v];
      for (size_t f = 0; f < fan_len.size(); f++)
        for (int t = 0; t < fan_len[f]; t++) {
          const int pos = fan_first_pos[f] + t, q = ord[pos];
          const int e = vcell[k0 + q] >> 2, a = vcell[k0 + q] & 3;
          LaneRec L;
          L.cellslot = vcell[k0 + q]; L.a = (unsigned char)a;
          unsigned slots = 0;
          for (int bq = 0; bq < 3; bq++) {
            const int w = c->h_cells[3 * e + (a + bq) % 3];
            slots |= (unsigned)(std::lower_bound(nb, nb + deg, w) - nb) << (8 * bq);
          }
          L.slots = slots;
          int prev_off = 0;
          if (t > 0) prev_off = -1;
          else if (fan_closed[f] && fan_len[f] > 1) prev_off = fan_len[f] - 1;
          L.has_prev = prev_off != 0;
          L.emit_v2 = (t == fan_len[f] - 1) && !fan_closed[f];
          if (fan_closed[f] && fan_len[f] == 1) { L.has_prev = 0; L.emit_v2 = 1; }
          L.seg = (unsigned)pos | ((unsigned)n << 8) | ((unsigned)(prev_off + 64) << 16);
          rowlanes[k0 + pos] = L;
        }
    }
  }
  // ---- pack whole rows into wavefronts (64 lanes) and 4 wavefronts into a workgroup; every
  // workgroup carries compact lists of the vertices and cells it touches (staged in LDS)
  std::vector<int> blk_row(1, 0), blk_vptr(1, 0), blk_cptr(1, 0), blk_vlist, blk_clist;
  std::vector<unsigned> lane_loc, lane_meta, lane_seg;   // per lane of every workgroup (0xFFFFFFFF = idle lane)
  std::vector<int> wave_maxlen;
  {
    std::vector<int> vmark(nv, -1), vloc(nv, 0), cmark(nc, -1), cloc(nc, 0);
    int r0 = 0, bid = 0;
    while (r0 < nvo) {
      int r1 = r0, nvl = 0, ncl = 0, wave = 0, lane = 0, wmax = 0;
      std::vector<unsigned> bloc(CFDH_MAX_INC, 0xFFFFFFFFu), bmeta(CFDH_MAX_INC, 0), bseg(CFDH_MAX_INC, 0);
      int wmaxes[CFDH_MAX_INC / 64] = {0};
      while (r1 < nvo) {
        const int di = vcptr[r1 + 1] - vcptr[r1];
        if (r1 - r0 >= CFDH_MAX_ROWS) break;
        int w2 = wave, l2 = lane;
        if (l2 + di > 64) { w2++; l2 = 0; }
        if (w2 >= CFDH_MAX_INC / 64) break;
        int addv = 0, addc = 0;
        for (int k = vcptr[r1]; k < vcptr[r1 + 1]; k++) {
          const int e = vcell[k] >> 2;
          if (cmark[e] != bid) { cmark[e] = bid; cloc[e] = -1; addc++; }
          for (int q = 0; q < 3; q++) { const int w = c->h_cells[3 * e + q]; if (vmark[w] != bid) { vmark[w] = bid; vloc[w] = -1; addv++; } }
        }
        if (nvl + addv > CFDH_MAX_BV || ncl + addc > CFDH_MAX_BC) {
          if (r1 == r0) return cfdh_fail(c, CFDH_E_ARG, "vertex patch too large for one workgroup");
          for (int k = vcptr[r1]; k < vcptr[r1 + 1]; k++) {
            const int e = vcell[k] >> 2;
            if (cmark[e] == bid && cloc[e] == -1) cmark[e] = -1;
            for (int q = 0; q < 3; q++) { const int w = c->h_cells[3 * e + q]; if (vmark[w] == bid && vloc[w] == -1) vmark[w] = -1; }
          }
          break;
        }
        wave = w2; lane = l2;
        for (int k = vcptr[r1]; k < vcptr[r1 + 1]; k++) {
          const LaneRec &L = rowlanes[k];
          const int e = L.cellslot >> 2, a = L.a;
          if (cloc[e] == -1) { cloc[e] = ncl++; blk_clist.push_back(e); }
          for (int q = 0; q < 3; q++) { const int w = c->h_cells[3 * e + q]; if (vloc[w] == -1) { vloc[w] = nvl++; blk_vlist.push_back(w); } }
          unsigned loc = (unsigned)cloc[e];
          for (int q = 0; q < 3; q++) loc |= (unsigned)vloc[c->h_cells[3 * e + (a + q) % 3]] << (8 * (q + 1));
          const int t = wave * 64 + lane;
          bloc[t] = loc;
          bmeta[t] = L.slots | ((unsigned)a << 24) | ((unsigned)L.emit_v2 << 26) | ((unsigned)L.has_prev << 27);
          bseg[t] = L.seg;
          lane++;
        }
        wmaxes[wave] = std::max(wmaxes[wave], di);
        wmax = std::max(wmax, di);
        r1++;
      }
      lane_loc.insert(lane_loc.end(), bloc.begin(), bloc.end());
      lane_meta.insert(lane_meta.end(), bmeta.begin(), bmeta.end());
      lane_seg.insert(lane_seg.end(), bseg.begin(), bseg.end());
      for (int w = 0; w < CFDH_MAX_INC / 64; w++) wave_maxlen.push_back(wmaxes[w]);
      blk_row.push_back(r1);
      blk_vptr.push_back((int)blk_vlist.size()); blk_cptr.push_back((int)blk_clist.size());
      r0 = r1; bid++;
    }
  }
  std::vector<unsigned> inc_slot(lane_meta), inc_rank(lane_seg), inc_loc(lane_loc);
  c->nblk = (int)blk_row.size() - 1;

  // ---- uploads
  hipStream_t s = c->stream;
  std::vector<unsigned char> cflag(nc, 0);
  for (int k = 0; k < c->nfac; k++) cflag[c->fac_cell[k]] |= (unsigned char)(1u << c->fac_local[k]);
  HIPCHK(c, c->coords.upload(c->h_coords, s));
  HIPCHK(c, c->cells.upload(c->h_cells, s));
  HIPCHK(c, c->cflag.upload(cflag, s));
  HIPCHK(c, c->mom.alloc(8 * (size_t)nc));
  HIPCHK(c, c->vptr.upload(c->h_vptr, s));
  HIPCHK(c, c->vcol.upload(c->h_vcol, s));
  HIPCHK(c, c->vdiag.upload(c->h_vdiag, s));
  HIPCHK(c, c->A00.alloc(4 * (size_t)c->nnzv));
  HIPCHK(c, c->A01.alloc(2 * (size_t)c->nnzv));
  HIPCHK(c, c->A10.alloc(2 * (size_t)c->nnzv));
  HIPCHK(c, c->A11.alloc((size_t)c->nnzv));
  HIPCHK(c, c->inc_slot.upload(inc_slot, s));
  HIPCHK(c, c->inc_rank.upload(inc_rank, s));
  HIPCHK(c, c->blk_row.upload(blk_row, s));
  HIPCHK(c, c->blk_vptr.upload(blk_vptr, s)); HIPCHK(c, c->blk_cptr.upload(blk_cptr, s));
  HIPCHK(c, c->blk_vlist.upload(blk_vlist, s)); HIPCHK(c, c->blk_clist.upload(blk_clist, s));
  HIPCHK(c, c->inc_loc.upload(inc_loc, s));
  HIPCHK(c, c->wave_maxlen.upload(wave_maxlen, s));
  {
    std::vector<unsigned char> cown(nc);
    for (int k = 0; k < nc; k++) cown[k] = cells[3 * c->cell_user[k]] < nvo ? 1 : 0;
    HIPCHK(c, c->cell_owned.upload(cown, s));
  }
  {
    std::vector<double> rnd(2 * (size_t)nvo);
    uint64_t st = 0x2545F4914F6CDD1Dull;
    for (auto &v : rnd) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = ((st >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
    HIPCHK(c, c->prand.upload(rnd, s));
  }
  if (c->nfac) {
    HIPCHK(c, c->d_fac_cell.upload(c->fac_cell, s));
    HIPCHK(c, c->d_fac_local.upload(c->fac_local, s));
    HIPCHK(c, c->d_fac_marker.upload(c->fac_marker, s));
  }
  c->h_bcflag.assign(nv, 0);
  c->h_bcval.assign(3 * (size_t)nv, 0.0);
  c->h_bcmult.assign(3 * (size_t)nv, 0.0);
  HIPCHK(c, c->bcflag.alloc(nv));
  HIPCHK(c, c->bcval.alloc(3 * (size_t)nv));
  HIPCHK(c, c->bcmult.alloc(3 * (size_t)nv));
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
  HIPCHK(c, c->dinvA.alloc(2 * (size_t)nvo));
  HIPCHK(c, c->pu0.alloc(2 * (size_t)nvo)); HIPCHK(c, c->pu1.alloc(2 * (size_t)nvo)); HIPCHK(c, c->pu2.alloc(2 * (size_t)nvo));
  HIPCHK(c, c->pr.alloc(2 * (size_t)nvo));
  HIPCHK(c, c->pp0.alloc(nvo)); HIPCHK(c, c->pp1.alloc(nvo));
  HIPCHK(c, hipStreamSynchronize(s));
  return 0;
}

// =============================================================================
// Smoothed-aggregation AMG for the (lagged) SELFP Schur matrix -- host setup.
// =============================================================================

static void csr_spmv_host(const CsrHost &A, const double *x, double *y) {
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads())
  for (int i = 0; i < A.n; i++) {
    double s = 0;
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) s += A.val[k] * x[A.col[k]];
    y[i] = s;
  }
}

static double lam_max_host(const CsrHost &A, const std::vector<double> &dinv, int its) {
  int n = A.n;
  std::vector<double> v(n), w(n);
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; i++) { st = st * 6364136223846793005ull + 1442695040888963407ull; v[i] = ((st >> 11) * (1.0 / 9007199254740992.0)) - 0.5; }
  double l = 1;
  for (int it = 0; it < its; it++) {
    csr_spmv_host(A, v.data(), w.data());
    double nn = 0;
    for (int i = 0; i < n; i++) { w[i] *= dinv[i]; nn += w[i] * w[i]; }
    l = std::sqrt(nn);
    if (!(l > 0)) return 1.0;
    for (int i = 0; i < n; i++) v[i] = w[i] / l;
  }
  return l;
}

// standard three-phase greedy aggregation on the strength graph
static int aggregate_host(const CsrHost &A, double theta, std::vector<int> &agg) {
  int n = A.n;
  std::vector<double> d(n, 0.0);
  for (int i = 0; i < n; i++)
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] == i) d[i] = std::fabs(A.val[k]);
  std::vector<int> sptr(n + 1, 0), scol;
  scol.reserve(A.col.size());
  for (int i = 0; i < n; i++) {
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
      int j = A.col[k];
      if (j != i && std::fabs(A.val[k]) >= theta * std::sqrt(d[i] * d[j])) scol.push_back(j);
    }
    sptr[i + 1] = (int)scol.size();
  }
  agg.assign(n, -1);
  int na = 0;
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    bool freeNb = sptr[i + 1] > sptr[i];
    for (int k = sptr[i]; k < sptr[i + 1] && freeNb; k++) if (agg[scol[k]] >= 0) freeNb = false;
    if (!freeNb) continue;
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) agg[scol[k]] = na;
    na++;
  }
  std::vector<int> agg2 = agg;
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) if (agg[scol[k]] >= 0) { agg2[i] = agg[scol[k]]; break; }
  }
  agg.swap(agg2);
  for (int i = 0; i < n; i++) {
    if (agg[i] >= 0) continue;
    if (sptr[i + 1] == sptr[i]) continue;  // isolated unknown (e.g. Dirichlet row): no coarse correction
    agg[i] = na;
    for (int k = sptr[i]; k < sptr[i + 1]; k++) if (agg[scol[k]] < 0) agg[scol[k]] = na;
    na++;
  }
  return na;
}

int cfdh_aggregate_host_csr(const CsrHost &A, double theta, std::vector<int> &agg) { return aggregate_host(A, theta, agg); }

// C = A * B (Gustavson, sorted output columns)
static void spgemm_host(const CsrHost &A, const CsrHost &B, CsrHost &C) {
  C.n = A.n; C.m = B.m;
  C.rowptr.assign(A.n + 1, 0);
  std::vector<std::vector<int>> cols(A.n);
  std::vector<std::vector<double>> vals(A.n);
#pragma omp parallel num_threads(cfdh_host_threads())
  {
    std::vector<int> mark(B.m, -1), list;
    std::vector<double> acc(B.m, 0.0);
#pragma omp for schedule(dynamic, 256)
    for (int i = 0; i < A.n; i++) {
      list.clear();
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) {
        int j = A.col[k];
        double a = A.val[k];
        for (int k2 = B.rowptr[j]; k2 < B.rowptr[j + 1]; k2++) {
          int cc = B.col[k2];
          if (mark[cc] != i) { mark[cc] = i; acc[cc] = 0.0; list.push_back(cc); }
          acc[cc] += a * B.val[k2];
        }
      }
      std::sort(list.begin(), list.end());
      cols[i] = list;
      vals[i].resize(list.size());
      for (size_t t = 0; t < list.size(); t++) vals[i][t] = acc[list[t]];
    }
  }
  for (int i = 0; i < A.n; i++) C.rowptr[i + 1] = C.rowptr[i] + (int)cols[i].size();
  C.col.resize(C.rowptr[A.n]); C.val.resize(C.rowptr[A.n]);
  for (int i = 0; i < A.n; i++) {
    std::copy(cols[i].begin(), cols[i].end(), C.col.begin() + C.rowptr[i]);
    std::copy(vals[i].begin(), vals[i].end(), C.val.begin() + C.rowptr[i]);
  }
}

static void transpose_host(const CsrHost &A, CsrHost &T) {
  T.n = A.m; T.m = A.n;
  T.rowptr.assign(A.m + 1, 0);
  for (int k = 0; k < A.nnz(); k++) T.rowptr[A.col[k] + 1]++;
  for (int i = 0; i < A.m; i++) T.rowptr[i + 1] += T.rowptr[i];
  T.col.resize(A.nnz()); T.val.resize(A.nnz());
  std::vector<int> fill(T.rowptr.begin(), T.rowptr.end() - 1);
  for (int i = 0; i < A.n; i++)
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) { int p = fill[A.col[k]]++; T.col[p] = i; T.val[p] = A.val[k]; }
}

// colw (optional): per-column weights for the scaled copy svalw[k] = val[k] * colw[col[k]]
static int upload_csr(cfdh_ctx *c, const CsrHost &H, CsrDev &D, const std::vector<double> *colw = nullptr,
                      int parts = CFDH_UP_CSR | CFDH_UP_SELL) {
  D.n = H.n; D.m = H.m; D.nnz = H.nnz();
  if (parts & (CFDH_UP_CSR | CFDH_UP_CSRF)) {
    HIPCHK(c, D.rowptr.upload(H.rowptr, c->stream));
    HIPCHK(c, D.col.upload(H.col, c->stream));
  }
  if (parts & CFDH_UP_CSR) HIPCHK(c, D.val.upload(H.val, c->stream));
  if (parts & CFDH_UP_CSRF) {
    std::vector<float> vf(H.val.begin(), H.val.end());
    HIPCHK(c, D.valf.upload(vf, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  if (!(parts & CFDH_UP_SELL)) { HIPCHK(c, hipStreamSynchronize(c->stream)); return 0; }
  // SELL-64
  const int ns = (H.n + 63) / 64;
  std::vector<int> sptr(ns + 1, 0);
  D.sell_maxw = 0;
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (ns > 256)
  for (int sl = 0; sl < ns; sl++) {
    int w = 0;
    for (int r = sl * 64; r < std::min(H.n, sl * 64 + 64); r++) w = std::max(w, H.rowptr[r + 1] - H.rowptr[r]);
    sptr[sl + 1] = 64 * w;
  }
  for (int sl = 0; sl < ns; sl++) { D.sell_maxw = std::max(D.sell_maxw, sptr[sl + 1] / 64); sptr[sl + 1] += sptr[sl]; }
  std::vector<int> scol((size_t)sptr[ns], 0);
  std::vector<float> sval((size_t)sptr[ns], 0.0f), svalw;
  if (colw) svalw.assign((size_t)sptr[ns], 0.0f);
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (H.n > 4096)
  for (int r = 0; r < H.n; r++) {
    const int sl = r >> 6, lane = r & 63;
    for (int k = H.rowptr[r], j = 0; k < H.rowptr[r + 1]; k++, j++) {
      const size_t p = (size_t)sptr[sl] + (size_t)j * 64 + lane;
      scol[p] = H.col[k]; sval[p] = (float)H.val[k];
      if (colw) svalw[p] = (float)(H.val[k] * (*colw)[H.col[k]]);
    }
    // padding keeps column 0 with value 0: a harmless in-range gather
  }
  D.nslice = ns;
  HIPCHK(c, D.sptr.upload(sptr, c->stream));
  HIPCHK(c, D.scol.upload(scol, c->stream));
  HIPCHK(c, D.sval.upload(sval, c->stream));
  if (colw) HIPCHK(c, D.svalw.upload(svalw, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // host temporaries go out of scope
  return 0;
}

// dense inverse by Gauss-Jordan with partial pivoting (coarsest level, n <= ~500)
static bool dense_inverse(std::vector<double> &a, int n) {
  std::vector<double> inv((size_t)n * n, 0.0);
  for (int i = 0; i < n; i++) inv[(size_t)i * n + i] = 1.0;
  for (int col = 0; col < n; col++) {
    int piv = col;
    double best = std::fabs(a[(size_t)col * n + col]);
    for (int r = col + 1; r < n; r++) if (std::fabs(a[(size_t)r * n + col]) > best) { best = std::fabs(a[(size_t)r * n + col]); piv = r; }
    if (!(best > 0)) return false;
    if (piv != col)
      for (int k = 0; k < n; k++) { std::swap(a[(size_t)piv * n + k], a[(size_t)col * n + k]); std::swap(inv[(size_t)piv * n + k], inv[(size_t)col * n + k]); }
    double d = 1.0 / a[(size_t)col * n + col];
    for (int k = 0; k < n; k++) { a[(size_t)col * n + k] *= d; inv[(size_t)col * n + k] *= d; }
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (n > 256)
    for (int r = 0; r < n; r++) {
      if (r == col) continue;
      double f = a[(size_t)r * n + col];
      if (f == 0.0) continue;
      for (int k = 0; k < n; k++) { a[(size_t)r * n + k] -= f * a[(size_t)col * n + k]; inv[(size_t)r * n + k] -= f * inv[(size_t)col * n + k]; }
    }
  }
  a.swap(inv);
  return true;
}

// device copy of one operator with its Jacobi diagonal, spectral bound and work vectors
int cfdh_upload_csr(cfdh_ctx *c, const CsrHost &H, CsrDev &D, const std::vector<double> *colw, int parts) { return upload_csr(c, H, D, colw, parts); }

// C = alpha * A + B restricted to the union pattern (both with sorted columns); scale_col / scale_row (optional)
// multiply the entries of A by s_col[j] / s_row[i] first.
static void csr_axpby_scaled(const CsrHost &A, double alpha, const double *s_row, const double *s_col, const CsrHost &B, CsrHost &C) {
  C.n = A.n; C.m = A.m;
  C.rowptr.assign(A.n + 1, 0);
  const int nt = cfdh_host_threads();
  // pass 1: size of the union pattern per row; pass 2: fill (rows are independent)
#pragma omp parallel for schedule(static) num_threads(nt) if (A.n > 4096)
  for (int i = 0; i < A.n; i++) {
    int ka = A.rowptr[i], kb = B.rowptr[i], cnt = 0;
    const int ea = A.rowptr[i + 1], eb = B.rowptr[i + 1];
    while (ka < ea || kb < eb) {
      const int ja = ka < ea ? A.col[ka] : 0x7fffffff, jb = kb < eb ? B.col[kb] : 0x7fffffff;
      const int j = ja < jb ? ja : jb;
      if (ja == j) ka++;
      if (jb == j) kb++;
      cnt++;
    }
    C.rowptr[i + 1] = cnt;
  }
  for (int i = 0; i < A.n; i++) C.rowptr[i + 1] += C.rowptr[i];
  C.col.resize(C.rowptr[A.n]); C.val.resize(C.rowptr[A.n]);
#pragma omp parallel for schedule(static) num_threads(nt) if (A.n > 4096)
  for (int i = 0; i < A.n; i++) {
    int ka = A.rowptr[i], kb = B.rowptr[i], p = C.rowptr[i];
    const int ea = A.rowptr[i + 1], eb = B.rowptr[i + 1];
    const double sr = s_row ? s_row[i] : 1.0;
    while (ka < ea || kb < eb) {
      const int ja = ka < ea ? A.col[ka] : 0x7fffffff, jb = kb < eb ? B.col[kb] : 0x7fffffff;
      const int j = ja < jb ? ja : jb;
      double v = 0.0;
      if (ja == j) { v += alpha * sr * A.val[ka] * (s_col ? s_col[j] : 1.0); ka++; }
      if (jb == j) { v += B.val[kb]; kb++; }
      C.col[p] = j; C.val[p] = v; p++;
    }
  }
}

int cfdh_level_setup(cfdh_ctx *c, AmgLevel &L, const CsrHost &A, double ratio, int ncol, std::vector<double> *w_out) {
  L.n = A.n;
  std::vector<double> dinv(A.n, 1.0);
  for (int i = 0; i < A.n; i++)
    for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++)
      if (A.col[k] == i && A.val[k] != 0.0) dinv[i] = 1.0 / A.val[k];
  const double lm = lam_max_host(A, dinv, 15);
  L.lmax = 1.1 * lm;
  L.lmin = L.lmax / ratio;
  HIPCHK(c, L.dinv.upload(dinv, c->stream));
  {
    // damped-Jacobi weights: 1/theta, theta = (lmax+lmin)/2; rows that are diagonal-only are solved exactly
    std::vector<double> w(A.n);
    const double itheta = 2.0 / (L.lmax + L.lmin);
    for (int i = 0; i < A.n; i++) {
      int offd = 0;
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) if (A.col[k] != i && A.val[k] != 0.0) offd++;
      w[i] = dinv[i] * (offd == 0 ? 1.0 : itheta);
    }
    HIPCHK(c, L.wdinv.upload(w, c->stream));
    CHK(upload_csr(c, A, L.A, &w));
    if (w_out) *w_out = w;
  }
  const size_t nn = (size_t)A.n * ncol;
  HIPCHK(c, L.x.alloc(nn)); HIPCHK(c, L.b.alloc(nn)); HIPCHK(c, L.r.alloc(nn));
  HIPCHK(c, L.d0.alloc(nn)); HIPCHK(c, L.d1.alloc(nn));
  return 0;
}

static double setup_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

int cfdh_amg_setup(cfdh_ctx *c, AmgHier &H, const CsrHost &A0, bool singular, int ncol) {
  double tm[6] = {0, 0, 0, 0, 0, 0};  // level set-up (+uploads), aggregation, prolongator, Galerkin/composite products, uploads, dense
  double tq = setup_ms();
#define TICK(i) do { const double now_ = setup_ms(); tm[i] += now_ - tq; tq = now_; } while (0)
  H.clear();
  H.ncol = ncol;
  const cfdh_options &o = c->opt;
  CsrHost A = A0;
  const int maxlev = 16;
  // composite operators of the fused cycle: only for the damped-Jacobi V(1,1) cycle (the default smoother)
  const char *nf = getenv("CFDH_NO_FUSED_AMG");
  const bool want_fused = o.amg_smooth_degree == 1 && !(nf && nf[0] == '1');
  H.fused = false; H.nnz_G0 = H.nnz_S0 = 0;
  std::vector<double> prevW, curW;
  CsrHost lastSc;  // Sc of the level above the current one (for the folded dense correction)
  for (;;) {
    AmgLevel *L = new AmgLevel();
    H.lev.push_back(L);
    const bool keep0 = H.keep_host0 && H.lev.size() == 1;
    std::vector<double> w;  // Jacobi weights of this level
    tq = setup_ms();
    CHK(cfdh_level_setup(c, *L, A, o.amg_smooth_ratio, ncol, &w));
    TICK(0);
    if (keep0) { H.h_wdinv0 = w; H.h_A0 = A; }
    // short regular rows (finest level of a P1 operator: ~7 entries on triangles, ~15 on tetrahedra): SELL-64 / fp32 kernels
    L->fine = A.nnz() <= (c->dim == 3 ? 20ll : 12ll) * A.n && A.n >= 16384;
    L->sell = L->fine && (A.nnz() <= 12ll * A.n || ncol == 1);
    prevW.swap(curW); curW = w;
    const double lm = L->lmax / 1.1;
    if (A.n <= o.amg_max_coarse || (int)H.lev.size() >= maxlev) break;
    std::vector<int> agg;
    // automatic threshold: the six-tetrahedra (Kuhn) and other anisotropic 3-D stencils hold many weak edges; with 0.08 the
    // aggregates fall apart (3-D bifurcation, 1.0 M DOF: 216 instead of 92 FGMRES iterations per step)
    // triangles: flat between 0.03 and 0.08 on all four 2-D configurations, degrading from 0.09 on (stenosis: 1132 -> 1759
    // iterations over 12 steps; 0.10 on the bench mesh: 467 -> 1459), tools/theta_scan.py -- 0.07 has the lowest counts on three of the four and keeps clear of that edge
    static const double theta_env = getenv("CFDH_AMG_THETA") ? atof(getenv("CFDH_AMG_THETA")) : -1.0;  // tuning hook for the default
    const double theta = o.amg_theta >= 0 ? o.amg_theta : (theta_env >= 0 ? theta_env : (c->dim == 3 ? 0.02 : 0.07));
    int na = aggregate_host(A, theta, agg);
    TICK(1);
    if (na >= A.n || na < 1) break;  // no coarsening possible
    std::vector<double> dinv(A.n, 1.0);
    for (int i = 0; i < A.n; i++)
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++)
        if (A.col[k] == i && A.val[k] != 0.0) dinv[i] = 1.0 / A.val[k];
    // P = (I - omega D^-1 A) P0,  P0 = piecewise constant; rows of isolated (diagonal-only) unknowns
    // carry no coarse correction (agg = -1): the smoother solves them exactly
    CsrHost P0;
    P0.n = A.n; P0.m = na;
    P0.rowptr.assign(A.n + 1, 0);
    for (int i = 0; i < A.n; i++) P0.rowptr[i + 1] = P0.rowptr[i] + (agg[i] >= 0 ? 1 : 0);
    P0.col.resize(P0.rowptr[A.n]); P0.val.assign(P0.rowptr[A.n], 1.0);
    for (int i = 0; i < A.n; i++) if (agg[i] >= 0) P0.col[P0.rowptr[i]] = agg[i];
    CsrHost AP0, P;
    spgemm_host(A, P0, AP0);
    const double omega = 4.0 / 3.0 / lm;
    P.n = A.n; P.m = na;
    P.rowptr.assign(A.n + 1, 0);
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (A.n > 4096)
    for (int i = 0; i < A.n; i++) {
      if (agg[i] < 0) { P.rowptr[i + 1] = 0; continue; }
      bool has = false;
      for (int k = AP0.rowptr[i]; k < AP0.rowptr[i + 1]; k++) if (AP0.col[k] == agg[i]) has = true;
      P.rowptr[i + 1] = (AP0.rowptr[i + 1] - AP0.rowptr[i]) + (has ? 0 : 1);
    }
    for (int i = 0; i < A.n; i++) P.rowptr[i + 1] += P.rowptr[i];
    P.col.resize(P.rowptr[A.n]); P.val.resize(P.rowptr[A.n]);
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads()) if (A.n > 4096)
    for (int i = 0; i < A.n; i++) {
      if (agg[i] < 0) continue;
      int p = P.rowptr[i];
      bool placed = false;
      for (int k = AP0.rowptr[i]; k < AP0.rowptr[i + 1]; k++) {
        int j = AP0.col[k];
        double v = -omega * dinv[i] * AP0.val[k];
        if (!placed && j > agg[i]) { P.col[p] = agg[i]; P.val[p] = 1.0; p++; placed = true; }
        if (j == agg[i]) { v += 1.0; placed = true; }
        P.col[p] = j; P.val[p] = v; p++;
      }
      if (!placed) { P.col[p] = agg[i]; P.val[p] = 1.0; p++; }
    }
    TICK(2);
    CsrHost R, AP, Ac;
    transpose_host(P, R);
    spgemm_host(A, P, AP);
    TICK(3);
    CHK(upload_csr(c, P, L->P));
    if (H.keep_host0 && H.lev.size() == 1) H.h_P0 = P;
    CHK(upload_csr(c, R, L->R));
    TICK(4);
    if (want_fused) {
      // G = R - (R A) W ; Sb = 2W - W A W ; Sc = P - W (A P) ; A_c = (R A) P
      CsrHost RA, G, Sb, Sc, Dg;
      spgemm_host(R, A, RA);
      csr_axpby_scaled(RA, -1.0, nullptr, curW.data(), R, G);
      Dg.n = Dg.m = A.n; Dg.rowptr.resize(A.n + 1); Dg.col.resize(A.n); Dg.val.resize(A.n);
      for (int i = 0; i < A.n; i++) { Dg.rowptr[i] = i; Dg.col[i] = i; Dg.val[i] = 2.0 * curW[i]; }
      Dg.rowptr[A.n] = A.n;
      csr_axpby_scaled(A, -1.0, curW.data(), curW.data(), Dg, Sb);
      csr_axpby_scaled(AP, -1.0, curW.data(), nullptr, P, Sc);
      spgemm_host(RA, P, Ac);
      TICK(3);
      const int fmt = L->fine ? CFDH_UP_CSRF : CFDH_UP_CSR;
      CHK(upload_csr(c, G, L->G, nullptr, fmt));
      CHK(upload_csr(c, Sb, L->Sb, nullptr, L->sell ? CFDH_UP_SELL : CFDH_UP_CSR));
      CHK(upload_csr(c, Sc, L->Sc, nullptr, L->sell ? CFDH_UP_SELL : CFDH_UP_CSR));
      if (H.lev.size() == 1) { H.nnz_G0 = G.nnz(); H.nnz_S0 = (long long)Sb.nnz() + Sc.nnz(); }
      TICK(4);
      lastSc.n = Sc.n; lastSc.m = Sc.m; lastSc.rowptr.swap(Sc.rowptr); lastSc.col.swap(Sc.col); lastSc.val.swap(Sc.val);
    } else {
      spgemm_host(R, AP, Ac);
      TICK(3);
    }
    A.n = Ac.n; A.m = Ac.m;
    A.rowptr.swap(Ac.rowptr); A.col.swap(Ac.col); A.val.swap(Ac.val);
  }
  // coarsest level.  Coarsening can stall on a level that is still large: where the mass term dominates
  // (rho/dt M against mu K on the aggregates) the positive mass and negative stiffness couplings cancel, no
  // connection is strong any more and the operator is close to diagonal.  Such a level needs no coarser
  // correction -- it is closed with two damped-Jacobi sweeps instead of a dense inverse (which would cost
  // O(n^3) on the host for n in the ten thousands).
  H.coarse_n = 0;
  if (A.n > 2500) {
    if (want_fused) {
      // two damped-Jacobi sweeps from a zero guess: x = (2W - W A W) b
      AmgLevel *L = H.lev.back();
      CsrHost Sb, Dg;
      Dg.n = Dg.m = A.n; Dg.rowptr.resize(A.n + 1); Dg.col.resize(A.n); Dg.val.resize(A.n);
      for (int i = 0; i < A.n; i++) { Dg.rowptr[i] = i; Dg.col[i] = i; Dg.val[i] = 2.0 * curW[i]; }
      Dg.rowptr[A.n] = A.n;
      csr_axpby_scaled(A, -1.0, curW.data(), curW.data(), Dg, Sb);
      CHK(upload_csr(c, Sb, L->Sb, nullptr, L->sell ? CFDH_UP_SELL : CFDH_UP_CSR));
      H.fused = H.lev.size() >= 1;
    }
    H.valid = true;
    H.fine_nnz = A0.nnz();
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->opt.verbose) {
      fprintf(stderr, "[cfdh] AMG hierarchy (ncol %d, smoothed coarsest level%s):", ncol, H.fused ? ", fused" : "");
      for (AmgLevel *l : H.lev) fprintf(stderr, " (%d, nnz %d; G %d Sb %d Sc %d)", l->n, l->A.nnz, l->G.nnz, l->Sb.nnz, l->Sc.nnz);
      fprintf(stderr, "\n");
    }
    return 0;
  }
  // otherwise: dense inverse; a singular (constant null vector) operator is
  // regularised with alpha * 1 1^T so that the inverse acts as a pseudo-inverse
  {
    int n = A.n;
    std::vector<double> D((size_t)n * n, 0.0);
    double tr = 0;
    for (int i = 0; i < n; i++)
      for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; k++) { D[(size_t)i * n + A.col[k]] = A.val[k]; if (A.col[k] == i) tr += std::fabs(A.val[k]); }
    if (singular) {
      double alpha = tr / n / n;
      for (size_t k = 0; k < D.size(); k++) D[k] += alpha;
    }
    if (!dense_inverse(D, n)) return cfdh_fail(c, CFDH_E_STATE, "singular coarsest AMG operator (n=%d)", n);
    HIPCHK(c, H.coarse_inv.upload(D, c->stream));
    H.coarse_n = n;
    if (want_fused && H.lev.size() >= 2) {
      H.fused = true;
      // fold the dense coarsest solve into the up-sweep of the level above: x = Sb b + (Sc A_c^-1) b_c
      AmgLevel *U = H.lev[H.lev.size() - 2];
      const long long ent = (long long)lastSc.n * n;
      U->Dn = 0;
      if (lastSc.m == n && ent <= 8000000ll && !U->sell) {
        std::vector<float> Df((size_t)ent);
#pragma omp parallel for schedule(static) num_threads(cfdh_host_threads())
        for (int i = 0; i < lastSc.n; i++) {
          std::vector<double> acc(n, 0.0);
          for (int k = lastSc.rowptr[i]; k < lastSc.rowptr[i + 1]; k++) {
            const double s = lastSc.val[k];
            const double *row = &D[(size_t)lastSc.col[k] * n];
            for (int j = 0; j < n; j++) acc[j] += s * row[j];
          }
          for (int j = 0; j < n; j++) Df[(size_t)i * n + j] = (float)acc[j];
        }
        HIPCHK(c, U->D.upload(Df, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        U->Dn = n;
      }
    }
  }
  H.fine_nnz = A0.nnz();
  H.valid = true;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (c->opt.verbose) {
    fprintf(stderr, "[cfdh] AMG hierarchy (ncol %d%s):", ncol, H.fused ? ", fused" : "");
    for (AmgLevel *l : H.lev) fprintf(stderr, " (%d, nnz %d; G %d Sb %d Sc %d D %d)", l->n, l->A.nnz, l->G.nnz, l->Sb.nnz, l->Sc.nnz, l->Dn);
    TICK(5);
    fprintf(stderr, "\n[cfdh]   host set-up ms: levels %.0f, aggregation %.0f, prolongator %.0f, products %.0f, uploads %.0f, dense %.0f\n", tm[0], tm[1],
            tm[2], tm[3], tm[4], tm[5]);
  }
#undef TICK
  return 0;
}
