// gfx950 kernels of the tetrahedral (gdim == 3) instance of the stabilized_schur step: P1/P1 on affine tetrahedra,
// 12 + 4 element dofs, 4x4 vertex blocks (A00 [9], A01 [3], A10 [3], A11 per graph entry).
//
//  * moments3_kernel : M_ab = int_K tau l_a l_b (10 values) and L = int_K tau_LSIC on the 171-point degree-13 rule (include/cfdh_quad_tet.h)
//                      (stabilized_schur.py:100-118; u_prev only -> once per time step)
//  * asm3q_kernel    : fused element residual + Jacobian + Dirichlet rows/cols + lifting (stabilized_schur.py:67-123,
//                      144-175,185-189).  Four lanes per (row vertex, cell) incidence, one 4x4 column block each.
//                      Unlike triangles around a vertex, tetrahedra around a vertex form no fan, so the blocks of
//                      a workgroup's rows are accumulated in LDS with ds_add_f64 and written out coalesced -- no global
//                      atomics; the summation order inside LDS is not fixed, so 3-D assembly is reproducible to
//                      round-off, not bitwise.
//  * spmv3 kernels   : 8 lanes per vertex row over the 4x4 block CSR, DPP reductions
// Algebra: oracle/np_twin_nd.py (SURVEY.md Appendix A with d = 3).  vector layout: [u 3*nv | p nv].
#include <hip/hip_runtime.h>

#include <cmath>

#include "cfdh_internal.hpp"
#include "cfdh_quad_tet.h"

#define TPB 256

__constant__ double d3_qw[CFDH3_NQ];
__constant__ double d3_ql[CFDH3_NQ][4];

int k3_upload_quadrature(cfdh_ctx *c) {
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_qw), CFDH3_QW, sizeof(CFDH3_QW)));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d3_ql), CFDH3_QL, sizeof(CFDH3_QL)));
  return 0;
}

template <int CTRL>
__device__ __forceinline__ double dpp3(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// quad-local permutation (every source lane lies in the reader's own quad, so no lane ever reads an undefined value and
// the destination needs no initialisation: one v_mov_b32_dpp per half instead of v_mov + v_mov_b32_dpp)
template <int CTRL>
__device__ __forceinline__ double dppq(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double g8sum(double v) {
  v += dpp3<0xB1>(v);
  v += dpp3<0x4E>(v);
  v += dpp3<0x141>(v);
  return v;
}
__device__ __forceinline__ double wsum3(double v) {
  v = g8sum(v);
  v += dpp3<0x140>(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ double bsum3(double v, double *sh) {
  v = wsum3(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}

// gradients of the barycentrics, volume, greatest vertex distance of a positively oriented tetrahedron
__device__ __forceinline__ void tet_geom(const double X[4][3], double g[4][3], double &vol, double &h) {
  double d[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < 3; i++) d[a][i] = X[a + 1][i] - X[0][i];
  double cr[3][3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int p = (a + 1) % 3, q = (a + 2) % 3;
    cr[a][0] = d[p][1] * d[q][2] - d[p][2] * d[q][1];
    cr[a][1] = d[p][2] * d[q][0] - d[p][0] * d[q][2];
    cr[a][2] = d[p][0] * d[q][1] - d[p][1] * d[q][0];
  }
  const double det = d[0][0] * cr[0][0] + d[0][1] * cr[0][1] + d[0][2] * cr[0][2];
  const double idet = 1.0 / det;
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int i = 0; i < 3; i++) g[a + 1][i] = cr[a][i] * idet;
#pragma unroll
  for (int i = 0; i < 3; i++) g[0][i] = -(g[1][i] + g[2][i] + g[3][i]);
  vol = fabs(det) * (1.0 / 6.0);
  double h2 = 0.0;
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = a + 1; b < 4; b++) {
      const double e0 = X[a][0] - X[b][0], e1 = X[a][1] - X[b][1], e2 = X[a][2] - X[b][2];
      h2 = fmax(h2, e0 * e0 + e1 * e1 + e2 * e2);
    }
  h = sqrt(h2);
}

// offsets into a vector in the partitioned layout [u owned 3 nvo | p owned nvo | ghosts (ux, uy, uz, p) ng]; a vertex id
// w >= nvo is a ghost (4 w = 4 nvo + 4 (w - nvo)).  On one rank (ng = 0) these are 3 w and 3 nv + w.
__device__ __forceinline__ size_t uo3(int w, int nvo) { return w < nvo ? 3 * (size_t)w : 4 * (size_t)w; }
__device__ __forceinline__ size_t po3(int w, int nvo) { return w < nvo ? 3 * (size_t)nvo + w : 4 * (size_t)w + 3; }

// ---------------------------------------------------------------- tau moments
// record per cell (12 doubles): M00 M01 M02 M03 M11 M12 M13 M22 M23 M33 L pad
__global__ __launch_bounds__(TPB) void moments3_kernel(int nc, int nv, const int *__restrict__ cells, const double *__restrict__ coords,
                                                       const double *__restrict__ un, double *__restrict__ mom, double dt, double nu) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= nc) return;
  int vs[4];
  double X[4][3], U[4][3];
#pragma unroll
  for (int a = 0; a < 4; a++) {
    vs[a] = cells[4 * (size_t)e + a];
#pragma unroll
    for (int i = 0; i < 3; i++) { X[a][i] = coords[3 * (size_t)vs[a] + i]; U[a][i] = un[uo3(vs[a], nv) + i]; }
  }
  double g[4][3], vol, h;
  tet_geom(X, g, vol, h);
  const double ih2 = 1.0 / (h * h);
  const double t2 = 4.0 / (dt * dt), t3 = 16.0 * nu * nu * ih2 * ih2, hr = h / (2.0 * nu);
  double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, L = 0.0;
  for (int q = 0; q < CFDH3_NQ; q++) {
    const double l0 = d3_ql[q][0], l1 = d3_ql[q][1], l2 = d3_ql[q][2], l3 = d3_ql[q][3], wq = d3_qw[q];
    const double ux = l0 * U[0][0] + l1 * U[1][0] + l2 * U[2][0] + l3 * U[3][0];
    const double uy = l0 * U[0][1] + l1 * U[1][1] + l2 * U[2][1] + l3 * U[3][1];
    const double uz = l0 * U[0][2] + l1 * U[1][2] + l2 * U[2][2] + l3 * U[3][2];
    const double s = ux * ux + uy * uy + uz * uz;
    const double t1 = fmax(4.0 * s, 1e-30) * ih2;
    const double tau = cfdh_rsqrt(t1 + t2 + t3);
    const double vn = s > 1e-280 ? s * cfdh_rsqrt(s) : 0.0, Re = vn * hr;
    const double z = (Re <= 3.0) ? Re * (1.0 / 3.0) : 1.0;
    L += wq * vn * h * z * 0.5;
    const double w = wq * tau;
    m[0] += w * l0 * l0; m[1] += w * l0 * l1; m[2] += w * l0 * l2; m[3] += w * l0 * l3;
    m[4] += w * l1 * l1; m[5] += w * l1 * l2; m[6] += w * l1 * l3;
    m[7] += w * l2 * l2; m[8] += w * l2 * l3; m[9] += w * l3 * l3;
  }
  double *o = mom + 12 * (size_t)e;
#pragma unroll
  for (int k = 0; k < 10; k++) o[k] = vol * m[k];
  o[10] = vol * L;
  o[11] = 0.0;
}

int k3_moments(cfdh_ctx *c) {
  prof_begin(c, 2);
  hipLaunchKernelGGL(moments3_kernel, dim3((c->nc + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nc, c->nvo, c->cells.p, c->coords.p,
                     c->xprev.p, c->mom.p, c->dt, c->mu / c->rho);
  prof_end(c, 2);
  HIPCHK(c, hipGetLastError());
  c->mom_valid = true;
  return 0;
}

// ---------------------------------------------------------------- fused assembly
// LDS image of a workgroup's CSR rows: 16 values per slot at a stride of 17 doubles.  With stride 16 (128 B) the k-th value
// of every slot lies in the same pair of LDS banks, and each ds_add_f64 of a wave (64 different slots, same k) is a
// 32-way bank conflict.
#define A3S 17
struct Asm3Args {
  const double *coords, *mom, *x, *un, *un2, *bcval, *bcmult;
  const int *cells, *vptr, *vdiag, *blk_row, *blk_iptr, *inc_cell, *inc_row;
  const unsigned long long *inc_slots;
  const unsigned char *cflag, *bcflag;
  double *A00, *A01, *A10, *A11, *F;
  int nv;
  double dt, rho, mu, muf, f[3];
  double theta, a0, a1, a2;
  int ds_terms, hist2;
  double beta_bf;  // backflow coefficient beta*rho on the facets flagged in cflag bits 4..7 (cfdh_set_boundary_terms)
};

// MODE 0: residual only; 1: residual + Jacobian; 2: residual with lifting (Jacobian values of Dirichlet columns only)
//
// Four lanes per incidence.  Lanes 4k .. 4k+3 of a workgroup share incidence k (row vertex a of cell e); lane q owns local vertex q of the cell: it loads
// that vertex's data, computes grad(lambda_q) and the q-th column block (16 values) of the row.  Everything that couples the
// vertices (G, grad p, the tau-weighted sums, the residual) is a sum over q and is formed with quad-local DPP butterflies,
// which leave the same value in all four lanes.  Per lane this is a quarter of the gathers and ~60 % of the registers of the
// first version (one lane per incidence computing all four column blocks: 338 registers, one wave per SIMD, 2.1 ms at
// 1.0 M DOF; this kernel: 206 registers, two waves per SIMD, 0.99 ms -- of which 0.35 ms came from the LDS stride alone).
__device__ __forceinline__ double quad_allsum(double v) {
  v += dppq<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dppq<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double quad_bcast(double v, int k) {  // value of lane k of the quad (k uniform in the quad)
  const double b0 = dppq<0x00>(v), b1 = dppq<0x55>(v), b2 = dppq<0xAA>(v), b3 = dppq<0xFF>(v);
  return k == 0 ? b0 : (k == 1 ? b1 : (k == 2 ? b2 : b3));
}
__device__ __forceinline__ int tet_mom_index(int r, int c) {  // packed upper triangle: 00 01 02 03 11 12 13 22 23 33
  const int lo = r < c ? r : c, hi = r < c ? c : r;
  return (lo == 0 ? 0 : (lo == 1 ? 4 : (lo == 2 ? 7 : 9))) + hi - lo;
}

#ifndef CFDH3_ASMQ_OCC
#define CFDH3_ASMQ_OCC 2  // 3 waves per SIMD (168 registers) spills 36 of them: 1.67 ms instead of 0.99
#endif
template <int MODE>
__global__ __launch_bounds__(TPB, CFDH3_ASMQ_OCC) void asm3q_kernel(Asm3Args p) {
  constexpr bool JAC = (MODE != 0);
  constexpr bool WJ = (MODE == 1);
  extern __shared__ double lds[];  // [nslots][16] value accumulators (MODE 1), then [nrows][4] residual accumulators
  const int blk = blockIdx.x, t = threadIdx.x, q = t & 3, lane = t & 63;
  const int row0 = p.blk_row[blk], row1 = p.blk_row[blk + 1], nrows = row1 - row0;
  const int slot0 = p.vptr[row0], nslots = p.vptr[row1] - slot0;
  double *accJ = lds;
  double *accF = lds + (WJ ? A3S * nslots : 0);
  for (int i = t; i < (WJ ? A3S * nslots : 0) + 4 * nrows; i += TPB) lds[i] = 0.0;
  __syncthreads();
  const int nv = p.nv;
  const double rho = p.rho, mu = p.mu, idt = 1.0 / p.dt, th = p.theta, a0idt = p.a0 * idt;
  // each wavefront walks its own list of incidences (whole rows: cfdh3_setup.cpp), 16 quads at a time; no barrier inside the loop
  const int i0 = p.blk_iptr[4 * blk + (t >> 6)], i1 = p.blk_iptr[4 * blk + (t >> 6) + 1];
  for (int base = i0; base < i1; base += 16) {
    const int inc_raw = base + (lane >> 2);
    const bool valid = inc_raw < i1;  // a whole quad is valid or not; invalid quads recompute the last incidence and add nothing
    const int inc = valid ? inc_raw : i1 - 1;
    const int ca = p.inc_cell[inc], e = ca >> 2, a = ca & 3;
    const int rloc = p.inc_row[inc];
    const unsigned slot = (unsigned)((p.inc_slots[inc] >> (16 * q)) & 0xffffull);
    const int vq = p.cells[4 * (size_t)e + q];
    const unsigned flq = p.bcflag[vq];
    const double pq = p.x[po3(vq, nv)];
    const size_t uq0 = uo3(vq, nv);
    double Xq[3], ubq[3], wq[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Xq[i] = p.coords[3 * (size_t)vq + i];
      const double u = p.x[uq0 + i], un = p.un[uq0 + i];
      ubq[i] = th * u + (1.0 - th) * un;
      double wt = p.a0 * u + p.a1 * un;
      if (p.hist2) wt += p.a2 * p.un2[uq0 + i];
      wq[i] = wt * idt;
    }
    const double *mo = p.mom + 12 * (size_t)e;
    double Mq[4];  // row q of the symmetric moment matrix
#pragma unroll
    for (int d = 0; d < 4; d++) Mq[d] = mo[tet_mom_index(q, d)];
    const double Lm = mo[10];
    // ---- grad(lambda_q) = n / (n . (A - B)), n = (C - B) x (D - B), with (A, B, C, D) = vertices (q, q+1, q+2, q+3) mod 4
    double gq[3], vol;
    {
      double cb[3], db[3], ab[3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const double B = dppq<0x39>(Xq[i]), C = dppq<0x4E>(Xq[i]), D = dppq<0x93>(Xq[i]);
        cb[i] = C - B; db[i] = D - B; ab[i] = Xq[i] - B;
      }
      const double n0 = cb[1] * db[2] - cb[2] * db[1], n1 = cb[2] * db[0] - cb[0] * db[2], n2 = cb[0] * db[1] - cb[1] * db[0];
      const double den = n0 * ab[0] + n1 * ab[1] + n2 * ab[2];
      const double iden = 1.0 / den;
      gq[0] = n0 * iden; gq[1] = n1 * iden; gq[2] = n2 * iden;
      vol = dppq<0x00>(fabs(den)) * (1.0 / 6.0);  // lane 0 of the quad speaks for all
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- quantities of the row vertex a (lane a of the quad)
    const int srcA = (lane & ~3) | a;
    double ga[3], uA[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { ga[i] = __shfl(gq[i], srcA); uA[i] = __shfl(ubq[i], srcA); }
    const unsigned flA = (unsigned)__shfl((int)flq, srcA);
    const double dqa = (q == a) ? 2.0 : 1.0;
    __builtin_amdgcn_sched_barrier(0);
    // ---- sums over the cell's vertices
    double G[3][3], gp[3], usum[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      gp[i] = quad_allsum(pq * gq[i]);
      usum[i] = quad_allsum(ubq[i]);
#pragma unroll
      for (int j = 0; j < 3; j++) G[i][j] = quad_allsum(gq[i] * ubq[j]);
    }
    const double psum = quad_allsum(pq);
    const double divu = G[0][0] + G[1][1] + G[2][2];
    double wcq[3], Rq[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const double cn = ubq[0] * G[0][j] + ubq[1] * G[1][j] + ubq[2] * G[2][j];
      wcq[j] = wq[j] + cn;
      Rq[j] = rho * wcq[j] + gp[j] - rho * p.f[j];
    }
    __builtin_amdgcn_sched_barrier(0);
    const double mtq = Mq[0] + Mq[1] + Mq[2] + Mq[3];
    const double T = quad_allsum(mtq);
    const double betaAq = ubq[0] * ga[0] + ubq[1] * ga[1] + ubq[2] * ga[2];  // beta[q][a]
    // MBa[q] = sum_d M[q][d] beta[d][a],  Q[q][i] = sum_c M[q][c] R[c][i]
    double MBaq = 0.0, Qq[3] = {0, 0, 0};
    {
      const double b0 = dppq<0x00>(betaAq), b1 = dppq<0x55>(betaAq), b2 = dppq<0xAA>(betaAq), b3 = dppq<0xFF>(betaAq);
      MBaq = Mq[0] * b0 + Mq[1] * b1 + Mq[2] * b2 + Mq[3] * b3;
#pragma unroll
      for (int i = 0; i < 3; i++)
        Qq[i] = Mq[0] * dppq<0x00>(Rq[i]) + Mq[1] * dppq<0x55>(Rq[i]) + Mq[2] * dppq<0xAA>(Rq[i]) + Mq[3] * dppq<0xFF>(Rq[i]);
    }
    const double m1 = vol * (1.0 / 20.0);
    double s1[3], s2[3], s3[3];  // m1 (sum_d ubar_d + ubar_a), sum_d MBa[d] ubar_d, sum_d mt[d] ubar_d
#pragma unroll
    for (int i = 0; i < 3; i++) {
      s1[i] = m1 * (usum[i] + uA[i]);
      s2[i] = quad_allsum(MBaq * ubq[i]);
      s3[i] = quad_allsum(mtq * ubq[i]);
    }
    const double mtBA = s3[0] * ga[0] + s3[1] * ga[1] + s3[2] * ga[2];
    double Gg[3];
#pragma unroll
    for (int j = 0; j < 3; j++) Gg[j] = G[j][0] * ga[0] + G[j][1] * ga[1] + G[j][2] * ga[2];
    const double pbar = 0.25 * psum;
    __builtin_amdgcn_sched_barrier(0);
    // ---- exterior facets containing vertex a: n_f |f| = -3 vol grad(lambda_f); SN = sum over the flagged facets f != a
    const unsigned cfraw = p.cflag[e];
    const unsigned cf = p.ds_terms ? (cfraw & 15u) : 0u;
    const unsigned cb = p.beta_bf != 0.0 ? (cfraw >> 4) : 0u;  // backflow facets
    double SN[3] = {0, 0, 0}, NFq[3] = {0, 0, 0};
    const bool facet_q = ((cf >> q) & 1u) && q != a;
    if (cf) {
#pragma unroll
      for (int i = 0; i < 3; i++) {
        NFq[i] = facet_q ? -3.0 * vol * gq[i] : 0.0;
        SN[i] = quad_allsum(NFq[i]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- residual of row a: this lane's share (terms of vertex q; the terms without a vertex sum go to lane 0)
    double Fr[4];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      double v = rho * m1 * dqa * wcq[i] + betaAq * Qq[i];
      if (q == 0) {
        double Eg = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) Eg += 0.5 * (G[i][k] + G[k][i]) * ga[k];
        v += vol * (2.0 * mu * Eg - pbar * ga[i]) - rho * p.f[i] * vol * 0.25 + rho * Lm * divu * ga[i];
      }
      Fr[i] = v;
    }
    Fr[3] = mtq * (Rq[0] * ga[0] + Rq[1] * ga[1] + Rq[2] * ga[2]) / rho + (q == 0 ? vol * 0.25 * divu : 0.0);
    if (cf) {
      const double pAv = __shfl(pq, srcA);
      if (facet_q) {
        // facet f = q (f != a): pint = (sum_k p_k (1 + delta_ka) - p_f) / 12
        const double pint = (psum + pAv - pq) * (1.0 / 12.0);
#pragma unroll
        for (int i = 0; i < 3; i++) {
          const double GN = G[i][0] * NFq[0] + G[i][1] * NFq[1] + G[i][2] * NFq[2];
          Fr[i] += NFq[i] * pint - p.muf * GN * (1.0 / 3.0);
        }
      }
    }
    // ---- backflow stabilisation on the flagged facets f that contain row vertex a and this lane's vertex q
    //      (stabilized_schur_backflow.py:165-176): F_a -= beta rho int_f (u_prev.n)_- lambda_a ubar ds, (s)_- = (s - |s|)/2, with the
    //      6-point degree-3 rule; n_f |f| = -3 vol grad(lambda_f), so sigma = u_prev(x_k) . (n_f |f|) carries the facet measure.
    //      Wbf = sum_f sum_k c_k lambda_a(x_k) lambda_q(x_k): -theta Wbf on the diagonal of J00, -Wbf ubar_q in the residual.
    double Wbf = 0.0;
    if (cb) {
      double unq[3];
#pragma unroll
      for (int i = 0; i < 3; i++) unq[i] = p.un[uq0 + i];
#pragma unroll
      for (int f = 0; f < 4; f++) {
        if (!((cb >> f) & 1u) || f == a) continue;  // uniform in the quad
        const double sc3 = -3.0 * vol;
        const double G0 = sc3 * quad_bcast(gq[0], f), G1 = sc3 * quad_bcast(gq[1], f), G2 = sc3 * quad_bcast(gq[2], f);
        const double tq = unq[0] * G0 + unq[1] * G1 + unq[2] * G2;
        const double t0 = dppq<0x00>(tq), t1 = dppq<0x55>(tq), t2 = dppq<0xAA>(tq), t3 = dppq<0xFF>(tq);
        // facet vertices in increasing local index (m0 < m1 < m2) and the positions of a and q among them
        const double s0 = f == 0 ? t1 : t0, s1 = f <= 1 ? t2 : t1, s2 = f <= 2 ? t3 : t2;
        const int ja = a - (a > f ? 1 : 0), jq = q - (q > f ? 1 : 0);
        if (q != f) {
          const double A = 0.659027622374092, B = 0.231933368553031, C = 0.109039009072877;
          const double P[6][3] = {{A, B, C}, {A, C, B}, {B, A, C}, {B, C, A}, {C, A, B}, {C, B, A}};
          double acc = 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) {
            const double sg = P[k][0] * s0 + P[k][1] * s1 + P[k][2] * s2;
            const double la = ja == 0 ? P[k][0] : (ja == 1 ? P[k][1] : P[k][2]);
            const double lq = jq == 0 ? P[k][0] : (jq == 1 ? P[k][1] : P[k][2]);
            acc += (sg - fabs(sg)) * la * lq;
          }
          Wbf += p.beta_bf * (0.5 / 6.0) * acc;
        }
      }
#pragma unroll
      for (int i = 0; i < 3; i++) Fr[i] -= Wbf * ubq[i];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- column block q of row a: the 16 values are produced one at a time and go straight to their LDS accumulator
    //      (never all live).  Dirichlet handling on the way: a Dirichlet column j of vertex q lifts F += J[:, j] (g - x) and
    //      is zeroed, a Dirichlet row of vertex a is zeroed (its diagonal is set at write-out).  MODE 2 needs the values of
    //      Dirichlet columns only.
    if (JAC && (WJ || flq)) {
      const double mab = m1 * dqa;
      const double gg = ga[0] * gq[0] + ga[1] * gq[1] + ga[2] * gq[2];
      const double mBab = s1[0] * gq[0] + s1[1] * gq[1] + s1[2] * gq[2];
      const double BMB = s2[0] * gq[0] + s2[1] * gq[1] + s2[2] * gq[2];
      const double mtBb = s3[0] * gq[0] + s3[1] * gq[1] + s3[2] * gq[2];
      const double cG = rho * th * (mab + MBaq);
      const double cD = rho * a0idt * (mab + MBaq) + rho * th * (mBab + BMB) + vol * mu * th * gg - th * Wbf;
      const double vmt = vol * mu * th, rlt = rho * Lm * th, cf3 = p.muf * th * (1.0 / 3.0);
      double *dst = accJ + A3S * (size_t)slot;
      const bool store = WJ && valid;
      if (!(flq | flA)) {
        // interior quad lanes: no masks
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            double v = cG * G[j][i] + vmt * gq[i] * ga[j] + th * ga[j] * Qq[i] + rlt * gq[j] * ga[i];
            if (i == j) v += cD;
            if (cf) v -= cf3 * gq[i] * SN[j];
            if (store) atomicAdd(dst + 3 * i + j, v);
          }
          double v = -vol * 0.25 * ga[i] + gq[i] * mtBA;
          if (cf) v += (SN[i] - NFq[i]) * dqa * (1.0 / 12.0);  // facets f != a, f != q
          if (store) atomicAdd(dst + 9 + i, v);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
          const double v = vol * 0.25 * th * gq[j] + mtq * (ga[j] * a0idt + th * Gg[j]) + th * ga[j] * mtBb;
          if (store) atomicAdd(dst + 12 + j, v);
        }
        if (store) atomicAdd(dst + 15, T * gg / rho);
      } else {
        double gx[4] = {0, 0, 0, 0};  // g - x on the Dirichlet columns of vertex q, zero elsewhere
#pragma unroll
        for (int jc = 0; jc < 3; jc++)
          if ((flq >> jc) & 1u) gx[jc] = p.bcval[4 * (size_t)vq + jc] - p.x[uq0 + jc];
        if (flq & 8u) gx[3] = p.bcval[4 * (size_t)vq + 3] - pq;
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
          for (int j = 0; j < 3; j++) {
            double v = cG * G[j][i] + vmt * gq[i] * ga[j] + th * ga[j] * Qq[i] + rlt * gq[j] * ga[i];
            if (i == j) v += cD;
            if (cf) v -= cf3 * gq[i] * SN[j];
            Fr[i] += v * gx[j];
            if (((flq >> j) | (flA >> i)) & 1u) v = 0.0;
            if (store) atomicAdd(dst + 3 * i + j, v);
          }
          double v = -vol * 0.25 * ga[i] + gq[i] * mtBA;
          if (cf) v += (SN[i] - NFq[i]) * dqa * (1.0 / 12.0);
          Fr[i] += v * gx[3];
          if (((flq >> 3) | (flA >> i)) & 1u) v = 0.0;
          if (store) atomicAdd(dst + 9 + i, v);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
          double v = vol * 0.25 * th * gq[j] + mtq * (ga[j] * a0idt + th * Gg[j]) + th * ga[j] * mtBb;
          Fr[3] += v * gx[j];
          if (((flq >> j) | (flA >> 3)) & 1u) v = 0.0;
          if (store) atomicAdd(dst + 12 + j, v);
        }
        {
          double v = T * gg / rho;
          Fr[3] += v * gx[3];
          if (((flq | flA) >> 3) & 1u) v = 0.0;
          if (store) atomicAdd(dst + 15, v);
        }
      }
    }
    // ---- residual: sum of the four shares; lane q adds component q (Dirichlet rows are replaced at write-out)
#pragma unroll
    for (int i = 0; i < 4; i++) Fr[i] = quad_allsum(Fr[i]);
    const double Fq = q == 0 ? Fr[0] : (q == 1 ? Fr[1] : (q == 2 ? Fr[2] : Fr[3]));
    const bool rowbc = (flA >> q) & 1u;  // bits 0-2: velocity components, bit 3: pressure
    if (valid && !rowbc) atomicAdd(accF + 4 * rloc + q, Fq);
  }
  __syncthreads();
  // ---- Dirichlet rows: diagonal = number of bc objects, F = x - g
  for (int r = t; r < nrows; r += TPB) {
    const int row = row0 + r;
    const unsigned flr = p.bcflag[row];
    if (!flr) continue;
    double *dg = accJ + A3S * (size_t)(p.vdiag[row] - slot0);
#pragma unroll
    for (int i = 0; i < 3; i++)
      if ((flr >> i) & 1u) {
        if (WJ) dg[4 * i] = p.bcmult[4 * (size_t)row + i];
        accF[4 * r + i] = p.x[3 * (size_t)row + i] - p.bcval[4 * (size_t)row + i];
      }
    if (flr & 8u) {
      if (WJ) dg[15] = p.bcmult[4 * (size_t)row + 3];
      accF[4 * r + 3] = p.x[3 * (size_t)nv + row] - p.bcval[4 * (size_t)row + 3];
    }
  }
  __syncthreads();
  // ---- coalesced write-out of the workgroup's slot range
  if (WJ) {
    for (int i = t; i < 9 * nslots; i += TPB) p.A00[9 * (size_t)slot0 + i] = accJ[A3S * (i / 9) + (i % 9)];
    for (int i = t; i < 3 * nslots; i += TPB) {
      p.A01[3 * (size_t)slot0 + i] = accJ[A3S * (i / 3) + 9 + (i % 3)];
      p.A10[3 * (size_t)slot0 + i] = accJ[A3S * (i / 3) + 12 + (i % 3)];
    }
    for (int i = t; i < nslots; i += TPB) p.A11[(size_t)slot0 + i] = accJ[A3S * i + 15];
  }
  for (int i = t; i < 3 * nrows; i += TPB) p.F[3 * (size_t)row0 + i] = accF[4 * (i / 3) + (i % 3)];
  for (int i = t; i < nrows; i += TPB) p.F[3 * (size_t)nv + row0 + i] = accF[4 * i + 3];
}

int k3_assemble(cfdh_ctx *c, const double *xstate, int mode) {
  Asm3Args a;
  a.coords = c->coords.p; a.mom = c->mom.p; a.x = xstate; a.un = c->xprev.p; a.un2 = c->xprev2.p; a.bcval = c->bcval.p; a.bcmult = c->bcmult.p;
  a.cells = c->cells.p; a.vptr = c->vptr.p; a.vdiag = c->vdiag.p; a.blk_row = c->a3_blk_row.p; a.blk_iptr = c->a3_blk_iptr.p;
  a.inc_cell = c->a3_inc_cell.p; a.inc_row = c->a3_inc_row.p; a.inc_slots = c->a3_inc_slots.p;
  a.cflag = c->cflag.p; a.bcflag = c->bcflag.p;
  a.A00 = c->A00.p; a.A01 = c->A01.p; a.A10 = c->A10.p; a.A11 = c->A11.p; a.F = c->F.p;
  a.nv = c->nvo;  // owned vertices: rows, and the split point of the partitioned vector layout
  a.dt = c->dt; a.rho = c->rho; a.mu = c->mu; a.muf = c->muf; a.f[0] = c->f[0]; a.f[1] = c->f[1]; a.f[2] = c->f[2];
  a.theta = c->ts_theta; a.a0 = c->ts_a[0]; a.a1 = c->ts_a[1]; a.a2 = c->ts_a[2];
  a.ds_terms = c->ds_terms ? 1 : 0; a.hist2 = c->ts_a[2] != 0.0 ? 1 : 0;
  a.beta_bf = (c->bf_beta > 0.0 && c->bf_marker >= 0) ? c->bf_beta * c->rho : 0.0;
  const size_t lds = sizeof(double) * (A3S * (size_t)CFDH3_MAX_SLOTS + 4 * 64);
  prof_begin(c, 0);
  if (mode == 1) hipLaunchKernelGGL((asm3q_kernel<1>), dim3(c->a3_nblk), dim3(TPB), lds, c->stream, a);
  else if (mode == 2) hipLaunchKernelGGL((asm3q_kernel<2>), dim3(c->a3_nblk), dim3(TPB), sizeof(double) * 4 * 64, c->stream, a);
  else hipLaunchKernelGGL((asm3q_kernel<0>), dim3(c->a3_nblk), dim3(TPB), sizeof(double) * 4 * 64, c->stream, a);
  prof_end(c, 0);
  HIPCHK(c, hipGetLastError());
  if (mode == 1) c->jac_valid = true;
  return 0;
}

// ---------------------------------------------------------------- block SpMV
// y = J x over the monolithic vector [u 3 nv | p nv]: 8 lanes per vertex row
__global__ __launch_bounds__(TPB) void spmv3_full_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                         const double *__restrict__ A00, const double *__restrict__ A01,
                                                         const double *__restrict__ A10, const double *__restrict__ A11,
                                                         const double *__restrict__ x, double *__restrict__ y) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (row < nv) {
    // (requesting the two blocks of a lane together -- 60 doubles in flight -- was measured: 159 us instead of 107)
    for (int k = vptr[row] + l, ke = vptr[row + 1]; k < ke; k += 8) {
      const int w = vcol[k];
      const size_t uo = uo3(w, nv);
      const double x0 = x[uo], x1 = x[uo + 1], x2 = x[uo + 2], xp = x[po3(w, nv)];
      const double *b = A00 + 9 * (size_t)k, *c01 = A01 + 3 * (size_t)k, *c10 = A10 + 3 * (size_t)k;
      a0 += b[0] * x0 + b[1] * x1 + b[2] * x2 + c01[0] * xp;
      a1 += b[3] * x0 + b[4] * x1 + b[5] * x2 + c01[1] * xp;
      a2 += b[6] * x0 + b[7] * x1 + b[8] * x2 + c01[2] * xp;
      a3 += c10[0] * x0 + c10[1] * x1 + c10[2] * x2 + A11[k] * xp;
    }
  }
  a0 = g8sum(a0); a1 = g8sum(a1); a2 = g8sum(a2); a3 = g8sum(a3);
  if (row < nv && l == 0) {
    y[3 * (size_t)row] = a0; y[3 * (size_t)row + 1] = a1; y[3 * (size_t)row + 2] = a2;
    y[3 * (size_t)nv + row] = a3;
  }
}
// NV vectors at once (leading dimension ld): one pass over the 4 x 4 blocks (projected initial guess, cfdh_solver.cpp)
template <int NV>
__global__ __launch_bounds__(TPB) void spmv3_full_multi_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                               const double *__restrict__ A00, const double *__restrict__ A01,
                                                               const double *__restrict__ A10, const double *__restrict__ A11,
                                                               const double *__restrict__ X, double *__restrict__ Y, size_t ld) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a[NV][4];
#pragma unroll
  for (int v = 0; v < NV; v++) a[v][0] = a[v][1] = a[v][2] = a[v][3] = 0.0;
  if (row < nv) {
    for (int k = vptr[row] + l, ke = vptr[row + 1]; k < ke; k += 8) {
      const int w = vcol[k];
      const size_t uo = uo3(w, nv), po = po3(w, nv);
      const double *b = A00 + 9 * (size_t)k, *c01 = A01 + 3 * (size_t)k, *c10 = A10 + 3 * (size_t)k;
      double m[16];
#pragma unroll
      for (int t = 0; t < 9; t++) m[t] = b[t];
      m[9] = c01[0]; m[10] = c01[1]; m[11] = c01[2]; m[12] = c10[0]; m[13] = c10[1]; m[14] = c10[2]; m[15] = A11[k];
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double *x = X + (size_t)v * ld;
        const double x0 = x[uo], x1 = x[uo + 1], x2 = x[uo + 2], xp = x[po];
        a[v][0] += m[0] * x0 + m[1] * x1 + m[2] * x2 + m[9] * xp;
        a[v][1] += m[3] * x0 + m[4] * x1 + m[5] * x2 + m[10] * xp;
        a[v][2] += m[6] * x0 + m[7] * x1 + m[8] * x2 + m[11] * xp;
        a[v][3] += m[12] * x0 + m[13] * x1 + m[14] * x2 + m[15] * xp;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const double s0 = g8sum(a[v][0]), s1 = g8sum(a[v][1]), s2 = g8sum(a[v][2]), s3 = g8sum(a[v][3]);
    if (row < nv && l == 0) {
      double *y = Y + (size_t)v * ld;
      y[3 * (size_t)row] = s0; y[3 * (size_t)row + 1] = s1; y[3 * (size_t)row + 2] = s2;
      y[3 * (size_t)nv + row] = s3;
    }
  }
}
int k3_spmv_full_multi(cfdh_ctx *c, const double *X, double *Y, int ld, int nvec) {
  if (nvec < 2 || nvec > 4) {
    for (int v = 0; v < nvec; v++) CHK(k3_spmv_full(c, X + (size_t)v * ld, Y + (size_t)v * ld));
    return 0;
  }
  const dim3 gr((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), bl(TPB);
#define CFDH_SPMM3(NV) hipLaunchKernelGGL((spmv3_full_multi_kernel<NV>), gr, bl, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A00.p, \
                                          c->A01.p, c->A10.p, c->A11.p, X, Y, (size_t)ld)
  if (nvec == 2) CFDH_SPMM3(2); else if (nvec == 3) CFDH_SPMM3(3); else CFDH_SPMM3(4);
#undef CFDH_SPMM3
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k3_spmv_full(cfdh_ctx *c, const double *x, double *y) {
  prof_begin(c, 1);
  hipLaunchKernelGGL(spmv3_full_kernel, dim3((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, c->nvo, c->vptr.p, c->vcol.p,
                     c->A00.p, c->A01.p, c->A10.p, c->A11.p, x, y);
  prof_end(c, 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// BLK 2: y_u = b_u - A01 x_p ; BLK 3: y_p = b_p - A10 x_u  (b null: y = A x)
// GHOST false: x is a compact block vector of the owned vertices (ghost columns are skipped: rank-local product);
// GHOST true: x is a full vector in the partitioned layout whose ghost tail was refreshed by comm_halo.
template <int BLK, bool GHOST = false>
__global__ __launch_bounds__(TPB) void spmv3_blk_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                        const double *__restrict__ A, const double *__restrict__ x,
                                                        double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0, a2 = 0;
  if (row < nv) {
    for (int k = vptr[row] + l, ke = vptr[row + 1]; k < ke; k += 8) {
      const int w = vcol[k];
      if (!GHOST && w >= nv) continue;
      const double *cc = A + 3 * (size_t)k;
      if (BLK == 2) {
        const double xp = GHOST ? x[po3(w, nv)] : x[w];
        a0 += cc[0] * xp; a1 += cc[1] * xp; a2 += cc[2] * xp;
      } else {
        const size_t uo = GHOST ? uo3(w, nv) : 3 * (size_t)w;
        a0 += cc[0] * x[uo] + cc[1] * x[uo + 1] + cc[2] * x[uo + 2];
      }
    }
  }
  a0 = g8sum(a0);
  if (BLK == 2) { a1 = g8sum(a1); a2 = g8sum(a2); }
  if (row < nv && l == 0) {
    if (BLK == 2) {
      const size_t o = 3 * (size_t)row;
      y[o] = bvec ? bvec[o] - a0 : a0; y[o + 1] = bvec ? bvec[o + 1] - a1 : a1; y[o + 2] = bvec ? bvec[o + 2] - a2 : a2;
    } else {
      y[row] = bvec ? bvec[row] - a0 : a0;
    }
  }
}
// y_u = b_u - A01 x_p, once per FGMRES iteration: four lanes per row, the first four entries of every lane requested
// together (the 8-lane loop above moves 24 B of matrix per load round: 27.9 us at 1 M DOF; this form: see DESIGN.md)
__global__ __launch_bounds__(TPB) void spmv3_a01_resid_kernel(int nv, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                              const double *__restrict__ A, const double *__restrict__ x,
                                                              double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 2, l = gid & 3;
  double a0 = 0, a1 = 0, a2 = 0;
  if (row < nv) {
    const int ks = vptr[row], ke = vptr[row + 1];
    int kq[4], wq[4];
    double cq[4][3], xq[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int k = ks + l + 4 * q; kq[q] = k < ke ? k : ks; wq[q] = vcol[kq[q]]; }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double *cc = A + 3 * (size_t)kq[q];
      cq[q][0] = cc[0]; cq[q][1] = cc[1]; cq[q][2] = cc[2];
      xq[q] = wq[q] < nv ? x[wq[q]] : 0.0;  // compact operand: ghost columns are skipped
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const double xp = (ks + l + 4 * q < ke) ? xq[q] : 0.0;
      a0 += cq[q][0] * xp; a1 += cq[q][1] * xp; a2 += cq[q][2] * xp;
    }
    for (int k = ks + l + 16; k < ke; k += 4) {
      if (vcol[k] >= nv) continue;
      const double xp = x[vcol[k]];
      const double *cc = A + 3 * (size_t)k;
      a0 += cc[0] * xp; a1 += cc[1] * xp; a2 += cc[2] * xp;
    }
  }
  a0 += dpp3<0xB1>(a0); a0 += dpp3<0x4E>(a0);
  a1 += dpp3<0xB1>(a1); a1 += dpp3<0x4E>(a1);
  a2 += dpp3<0xB1>(a2); a2 += dpp3<0x4E>(a2);
  if (row < nv && l == 0) {
    const size_t o = 3 * (size_t)row;
    y[o] = bvec[o] - a0; y[o + 1] = bvec[o + 1] - a1; y[o + 2] = bvec[o + 2] - a2;
  }
}
int k3_spmv_block_ghost(cfdh_ctx *c, int blk, const double *xv, double *y, const double *b) {
  dim3 grid((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), block(TPB);
  if (blk == 2) hipLaunchKernelGGL((spmv3_blk_kernel<2, true>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A01.p, xv, y, b);
  else hipLaunchKernelGGL((spmv3_blk_kernel<3, true>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A10.p, xv, y, b);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k3_spmv_block(cfdh_ctx *c, int blk, const double *x, double *y, const double *b) {
  dim3 grid((unsigned)((8ll * c->nvo + TPB - 1) / TPB)), block(TPB);
  if (blk == 2 && b)
    hipLaunchKernelGGL(spmv3_a01_resid_kernel, dim3((unsigned)((4ll * c->nvo + TPB - 1) / TPB)), block, 0, c->stream, c->nvo, c->vptr.p,
                       c->vcol.p, c->A01.p, x, y, b);
  else if (blk == 2) hipLaunchKernelGGL((spmv3_blk_kernel<2>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A01.p, x, y, b);
  else if (blk == 3) hipLaunchKernelGGL((spmv3_blk_kernel<3>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A10.p, x, y, b);
  else return cfdh_fail(c, CFDH_E_ARG, "k3_spmv_block: block %d not available for tetrahedra", blk);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ||J n|| and || |J| n || for the constant-pressure vector (MatNullSpaceTest, stabilized_schur.py:314)
__global__ __launch_bounds__(TPB) void nulltest3_kernel(int nv, const int *__restrict__ vptr, const double *__restrict__ A01,
                                                        const double *__restrict__ A11, double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0, b = 0;
  for (int row = blockIdx.x * TPB + threadIdx.x; row < nv; row += gridDim.x * TPB) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int k = vptr[row]; k < vptr[row + 1]; k++) {
      const double c0 = A01[3 * (size_t)k], c1 = A01[3 * (size_t)k + 1], c2 = A01[3 * (size_t)k + 2], c3 = A11[k];
      s0 += c0; s1 += c1; s2 += c2; s3 += c3;
      t0 += fabs(c0); t1 += fabs(c1); t2 += fabs(c2); t3 += fabs(c3);
    }
    a += s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3;
    b += t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3;
  }
  a = bsum3(a, sh);
  b = bsum3(b, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = a; partial[gridDim.x + blockIdx.x] = b; }
}
__global__ __launch_bounds__(TPB) void final3_kernel(int nb, int stride, const double *__restrict__ partial, double *__restrict__ out) {
  __shared__ double sh[4];
  double a = 0;
  for (int i = threadIdx.x; i < nb; i += TPB) a += partial[(size_t)blockIdx.x * stride + i];
  a = bsum3(a, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = a;
}
static int read2(cfdh_ctx *c, double *v, int n) {
  if (c->nranks > 1) CHK(comm_allreduce_dev(c, c->red_out.p, n, 0));  // sums over the parts of a partitioned mesh
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, c->red_out.p, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) v[i] = c->h_pinned[i];
  return 0;
}
int k3_nullspace_test(cfdh_ctx *c, double *nrm, double *absnrm) {
  const int nb = 256;
  c->mirror_src = nullptr;
  hipLaunchKernelGGL(nulltest3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nvo, c->vptr.p, c->A01.p, c->A11.p, c->red_partial.p);
  hipLaunchKernelGGL(final3_kernel, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
  HIPCHK(c, hipGetLastError());
  double s[2];
  CHK(read2(c, s, 2));
  *nrm = sqrt(s[0]);
  *absnrm = sqrt(s[1]);
  return 0;
}

// ---------------------------------------------------------------- functionals
// kind 2/3: ||u||_L2, ||p||_L2 (scenario.py:315-324); kind 7: outward volume flux through the facets of `marker`
__global__ __launch_bounds__(TPB) void l2_3_kernel(int nc, int nv, const int *__restrict__ cells, const unsigned char *__restrict__ cown,
                                                   const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double au = 0, ap = 0;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nc; e += gridDim.x * TPB) {
    if (!cown[e]) continue;  // overlapping parts: a cell is integrated by the rank that owns its first vertex
    int vs[4];
    double X[4][3];
    for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
    double g[4][3], vol, h;
    tet_geom(X, g, vol, h);
    double su = 0, sp = 0;
    for (int a = 0; a < 4; a++)
      for (int b = 0; b < 4; b++) {
        const double m = (a == b ? 2.0 : 1.0);
        const size_t ua = uo3(vs[a], nv), ub = uo3(vs[b], nv);
        su += m * (x[ua] * x[ub] + x[ua + 1] * x[ub + 1] + x[ua + 2] * x[ub + 2]);
        sp += m * x[po3(vs[a], nv)] * x[po3(vs[b], nv)];
      }
    au += vol * su * (1.0 / 20.0);
    ap += vol * sp * (1.0 / 20.0);
  }
  au = bsum3(au, sh);
  ap = bsum3(ap, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = au; partial[gridDim.x + blockIdx.x] = ap; }
}
__global__ __launch_bounds__(TPB) void flux3_kernel(int nfac, int marker, int nv, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                    const int *__restrict__ fmarker, const int *__restrict__ cells, const unsigned char *__restrict__ cown,
                                                    const double *__restrict__ coords, const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double q = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker) continue;
    const int e = fcell[k], fl = flocal[k];
    if (!cown[e]) continue;
    int vs[4];
    double X[4][3];
    for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
    double g[4][3], vol, h;
    tet_geom(X, g, vol, h);
    // |f| n = -3 vol grad l_f ; flux = |f| n . mean of the three facet vertices' velocities
    double um[3] = {0, 0, 0};
    for (int a = 0; a < 4; a++) if (a != fl) for (int i = 0; i < 3; i++) um[i] += x[uo3(vs[a], nv) + i] * (1.0 / 3.0);
    q += -3.0 * vol * (g[fl][0] * um[0] + g[fl][1] * um[1] + g[fl][2] * um[2]);
  }
  q = bsum3(q, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = q;
}
int k3_functional(cfdh_ctx *c, int kind, int marker, double *out) {
  const int nb = 256;
  c->mirror_src = nullptr;
  if (c->gen && (kind == 2 || kind == 3 || kind == 7)) {  // hexahedra / P2 tetrahedra: the element's own quadrature (cfdh_gen3.hip)
    CHK(kg3_functional_partials(c, kind, marker, nb));
    hipLaunchKernelGGL(final3_kernel, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
    HIPCHK(c, hipGetLastError());
    double v[2];
    CHK(read2(c, v, 2));
    *out = kind == 7 ? v[0] : sqrt(kind == 2 ? v[0] : v[1]);
    return 0;
  }
  if (kind == 2 || kind == 3) {
    hipLaunchKernelGGL(l2_3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nc, c->nvo, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
    hipLaunchKernelGGL(final3_kernel, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
    HIPCHK(c, hipGetLastError());
    double v[2];
    CHK(read2(c, v, 2));
    *out = sqrt(kind == 2 ? v[0] : v[1]);
    return 0;
  }
  if (kind == 7) {
    hipLaunchKernelGGL(flux3_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nfac, marker, c->nvo, c->d_fac_cell.p, c->d_fac_local.p,
                       c->d_fac_marker.p, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
    hipLaunchKernelGGL(final3_kernel, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p);
    HIPCHK(c, hipGetLastError());
    return read2(c, out, 1);
  }
  if (kind >= 4 && kind <= 6) {
    const int nu = 3 * c->nvo;
    const double *a = kind == 5 ? c->xprev.p : c->x.p;
    const double *b = kind == 6 ? c->xprev.p : nullptr;
    return v_norminf_diff(c, nu, a, b, out);
  }
  return cfdh_fail(c, CFDH_E_ARG, "functional kind %d is not available for tetrahedra (2, 3: L2 norms; 4-6: inf-norms; 7: flux)", kind);
}

// wall shear stress (solverBase.py:163-195) on triangles: (1/|f|) oint l_a Tt ds = Tt / 3 for the three facet vertices
__global__ __launch_bounds__(TPB) void wss3_kernel(int nfac, int nv, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                   const int *__restrict__ cells, const double *__restrict__ coords,
                                                   const double *__restrict__ x, double mu, double *__restrict__ out) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nfac) return;
  const int e = fcell[k], fl = flocal[k];
  int vs[4];
  double X[4][3];
  for (int a = 0; a < 4; a++) { vs[a] = cells[4 * (size_t)e + a]; for (int i = 0; i < 3; i++) X[a][i] = coords[3 * (size_t)vs[a] + i]; }
  double g[4][3], vol, h;
  tet_geom(X, g, vol, h);
  const double gl = sqrt(g[fl][0] * g[fl][0] + g[fl][1] * g[fl][1] + g[fl][2] * g[fl][2]);
  const double n[3] = {-g[fl][0] / gl, -g[fl][1] / gl, -g[fl][2] / gl};
  double G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int a = 0; a < 4; a++)
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) G[i][j] += g[a][i] * x[uo3(vs[a], nv) + j];
  double T[3];
  for (int i = 0; i < 3; i++) {
    T[i] = 0.0;
    for (int j = 0; j < 3; j++) T[i] -= mu * (G[i][j] + G[j][i]) * n[j];
  }
  const double Tn = T[0] * n[0] + T[1] * n[1] + T[2] * n[2];
  for (int a = 0; a < 4; a++) {
    if (a == fl) continue;
    for (int i = 0; i < 3; i++) atomicAdd(out + 3 * (size_t)vs[a] + i, (T[i] - Tn * n[i]) * (1.0 / 3.0));
  }
}
int k3_wss(cfdh_ctx *c, double *out) {
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(double) * 3 * (size_t)c->nv, c->stream));
  if (c->nfac > 0)
    hipLaunchKernelGGL(wss3_kernel, dim3((c->nfac + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nfac, c->nvo, c->d_fac_cell.p, c->d_fac_local.p,
                       c->cells.p, c->coords.p, c->x.p, c->mu, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
