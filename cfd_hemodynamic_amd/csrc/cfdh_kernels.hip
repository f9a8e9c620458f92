// Hand-written gfx950 (CDNA4, wave64) kernels of the stabilized_schur hot path.
//
//  * k_moments    : tau / tau_LSIC moments per cell (stabilized_schur.py:100-118)
//  * asm_kernel   : fused element residual + Jacobian, one lane per (row vertex, cell)
//                   incidence; patch data staged in LDS; the cells of a vertex form a
//                   counter-clockwise fan, so off-diagonal blocks are completed by one
//                   lane shuffle and the diagonal block / residual by a segmented
//                   wavefront reduction -- no LDS accumulation, no atomics, fixed order
//                   (bitwise reproducible); plain stores of complete 3x3 blocks
//                   (stabilized_schur.py:67-123,144-175,185-189)
//  * spmv kernels : 8 lanes per vertex row over the block CSR, DPP reductions
//  * AMG          : fused V(1,1) Jacobi cycle on composite operators (one kernel per level
//                   and direction), SELL-64 / fp32 on the fine levels, dense coarse solve
//                   folded into the level above; sweep-by-sweep cycle for the other smoothers
//                   and the distributed finest level of a partitioned run
//  * vector kernels of FGMRES (fused multi-dot, Gram-Schmidt update + normalisation)
//
// Everything is HBM-bound fp64 stream/gather work; MFMA is not used (nothing
// here is a dense contraction).  Algebra: SURVEY.md Appendix A / DESIGN.md.
#include <hip/hip_runtime.h>

#include <cmath>

#include "cfdh_internal.hpp"
#include "cfdh_quad_tri.h"

#define TPB 256
// streamed-once operands (matrix values / columns of the AMG sweeps): non-temporal loads keep them from evicting the
// gathered vector entries out of the 32 KB L1
#define NTLOAD(p) __builtin_nontemporal_load(p)

__constant__ double d_qw[CFDH_NQ];
__constant__ double d_ql[CFDH_NQ][3];

int k_upload_quadrature(cfdh_ctx *c) {
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d_qw), CFDH_QW, sizeof(CFDH_QW)));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(d_ql), CFDH_QL, sizeof(CFDH_QL)));
  return 0;
}

// ---------------------------------------------------------------- profiling
void prof_begin(cfdh_ctx *c, int kind) {
  if (!c->prof_on || c->capturing) return;
  if (c->ev_next + 2 > c->ev_pool.size()) {
    for (int i = 0; i < 64; i++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; c->ev_pool.push_back(e); }
  }
  cfdh_ctx::EvRec r;
  r.kind = kind; r.a = c->ev_pool[c->ev_next++]; r.b = c->ev_pool[c->ev_next++];
  (void)hipEventRecord(r.a, c->stream);
  c->ev_pending.push_back(r);
}
void prof_end(cfdh_ctx *c, int kind) {
  if (!c->prof_on || c->capturing || c->ev_pending.empty()) return;
  (void)kind;
  (void)hipEventRecord(c->ev_pending.back().b, c->stream);
  if (c->ev_pending.size() >= 4096) prof_flush(c);
}
void prof_flush(cfdh_ctx *c) {
  if (c->ev_pending.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  for (auto &r : c->ev_pending) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { c->prof[r.kind].total_ms += ms; c->prof[r.kind].launches++; }
  }
  c->ev_pending.clear();
  c->ev_next = 0;
}

// ---------------------------------------------------------------- wave-level helpers
template <int CTRL>
__device__ __forceinline__ double dpp_shuffle(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// sum over aligned groups of 8 lanes (result in every lane of the group)
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_shuffle<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_shuffle<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_shuffle<0x141>(v);  // row_half_mirror
  return v;
}
// sum over aligned groups of 4 lanes
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_shuffle<0xB1>(v);  // quad_perm [1,0,3,2]
  v += dpp_shuffle<0x4E>(v);  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// sum over the 64 lanes of the wave (result in every lane)
__device__ __forceinline__ double wave_sum(double v) {
  v = group8_sum(v);
  v += dpp_shuffle<0x140>(v);  // row_mirror -> sums of 16
  return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_shuffle<0xB1>(v));
  v = fmax(v, dpp_shuffle<0x4E>(v));
  v = fmax(v, dpp_shuffle<0x141>(v));
  v = fmax(v, dpp_shuffle<0x140>(v));
  return fmax(fmax(readlane_d(v, 0), readlane_d(v, 16)), fmax(readlane_d(v, 32), readlane_d(v, 48)));
}
// block (256 threads) sum; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double *sh /*[4]*/) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
  __syncthreads();
  return r;
}
__device__ __forceinline__ double block_max(double v, double *sh) {
  v = wave_max(v);
  int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  double r = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
  __syncthreads();
  return r;
}

// vector layout: [u owned 2*nvo | p owned nvo | ghosts (ux,uy,p) x ng]
__device__ __forceinline__ int uoff(int w, int nvo) { return w < nvo ? 2 * w : 3 * w; }
__device__ __forceinline__ int poff(int w, int nvo) { return w < nvo ? 2 * nvo + w : 3 * w + 2; }

// ---------------------------------------------------------------- tau moments
// M_ab = int_K tau l_a l_b, L = int_K tau_L on the 49-point rule; tau, tau_L
// depend on u_prev only (stabilized_schur.py:91-93,100-108,116-118) so this
// runs once per time step.  One lane per cell, coalesced 64-B records out.
__global__ __launch_bounds__(TPB) void moments_kernel(int nc, int nvo, const int *__restrict__ cells,
                                                      const double *__restrict__ coords,
                                                      const double *__restrict__ un, double *__restrict__ mom,
                                                      double dt, double nu) {
  int e = blockIdx.x * TPB + threadIdx.x;
  if (e >= nc) return;
  int v0 = cells[3 * e], v1 = cells[3 * e + 1], v2 = cells[3 * e + 2];
  double2 X0 = *(const double2 *)(coords + 2 * v0), X1 = *(const double2 *)(coords + 2 * v1), X2 = *(const double2 *)(coords + 2 * v2);
  double u0x = un[uoff(v0, nvo)], u0y = un[uoff(v0, nvo) + 1];
  double u1x = un[uoff(v1, nvo)], u1y = un[uoff(v1, nvo) + 1];
  double u2x = un[uoff(v2, nvo)], u2y = un[uoff(v2, nvo) + 1];
  double det = (X1.x - X0.x) * (X2.y - X0.y) - (X1.y - X0.y) * (X2.x - X0.x);
  double area = 0.5 * fabs(det);
  double d01 = hypot(X0.x - X1.x, X0.y - X1.y), d12 = hypot(X1.x - X2.x, X1.y - X2.y), d20 = hypot(X2.x - X0.x, X2.y - X0.y);
  double h = fmax(d01, fmax(d12, d20));
  double ih2 = 1.0 / (h * h);
  double t2 = 4.0 / (dt * dt), t3 = 16.0 * nu * nu * ih2 * ih2;
  double hr = h / (2.0 * nu);
  double m00 = 0, m01 = 0, m02 = 0, m11 = 0, m12 = 0, m22 = 0, L = 0;
#pragma unroll 7
  for (int q = 0; q < CFDH_NQ; q++) {
    double l0 = d_ql[q][0], l1 = d_ql[q][1], l2 = d_ql[q][2], wq = d_qw[q];
    double ux = l0 * u0x + l1 * u1x + l2 * u2x;
    double uy = l0 * u0y + l1 * u1y + l2 * u2y;
    double s = ux * ux + uy * uy;
    double t1 = fmax(4.0 * s, 1e-30) * ih2;  // (max(2|u|, eps))^2 / h^2, eps = 1e-15
    double tau = cfdh_rsqrt(t1 + t2 + t3);
    double vn = s > 1e-280 ? s * cfdh_rsqrt(s) : 0.0;
    double Re = vn * hr;
    double z = (Re <= 3.0) ? Re * (1.0 / 3.0) : 1.0;
    double tl = vn * h * z * 0.5;
    double w = wq * tau;
    m00 += w * l0 * l0; m01 += w * l0 * l1; m02 += w * l0 * l2;
    m11 += w * l1 * l1; m12 += w * l1 * l2; m22 += w * l2 * l2;
    L += wq * tl;
  }
  double4 *o = (double4 *)(mom + 8 * (size_t)e);
  o[0] = make_double4(area * m00, area * m01, area * m02, area * m11);
  o[1] = make_double4(area * m12, area * m22, area * L, 0.0);
}

int k_moments(cfdh_ctx *c) {
  if (c->gen) { c->mom_valid = true; return 0; }  // tau is evaluated inside the quadrature loop of the generic kernels
  if (c->dim == 3) return k3_moments(c);
  prof_begin(c, 2);
  hipLaunchKernelGGL(moments_kernel, dim3((c->nc + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nc, c->nvo, c->cells.p,
                     c->coords.p, c->xprev.p, c->mom.p, c->dt, c->mu / c->rho);
  prof_end(c, 2);
  HIPCHK(c, hipGetLastError());
  c->mom_valid = true;
  return 0;
}

// ---------------------------------------------------------------- fused assembly
struct AsmArgs {
  const double *coords, *mom, *x, *un, *un2, *bcval, *bcmult;
  const int *vptr, *blk_row, *blk_vptr, *blk_vlist, *blk_cptr, *blk_clist, *wave_maxlen;
  const unsigned *inc_slot, *inc_rank, *inc_loc;
  const unsigned char *cflag, *bcflag;
  double *A00, *A01, *A10, *A11, *F;
  int nvo;
  double dt, rho, mu, muf, fx, fy;
  double theta, a0, a1, a2;  // time scheme (cfdh_set_time_scheme)
  double beta_bf;            // backflow coefficient beta*rho on facets flagged in cflag bits 3..5 (cfdh_set_boundary_terms)
  int ds_terms;              // the ds pair of stabilized_schur.py:79 on all exterior facets
#ifdef CFDH_ASM_TIMING
  long long *dbg;            // [nblk][8] phase time stamps of wave 0 (diagnostic build only)
#endif
};

// MODE 0: residual only; 1: residual + Jacobian; 2: residual with lifting (Jacobian in registers only)
//
// Workgroup = a run of whole matrix rows.  Prologue: the vertices and cells the block touches
// (compact per-block lists built at setup) are loaded ONCE, coalesced, into LDS -- nodal
// coordinates, iterate, u_prev, Dirichlet flags, and the 64-B tau-moment records; every lane then
// gathers its element from LDS only (a per-lane gather from global memory costs one L1 line
// per lane per load and made the first version of this kernel L1/TA-bound).
// HIST2: the time term carries a2*u_prev2 (BDF2 steps of stabilized_schur_bdf2.py:95-110).
#ifndef CFDH_BF_OCC
#define CFDH_BF_OCC 2  // the backflow variant needs ~10 more VGPRs than the 168 of 3 waves/SIMD: run it at 2
#endif
// BF: backflow facets present (compiled out otherwise so that the base solver keeps its register budget).
#ifndef CFDH_ASM_OCC
#define CFDH_ASM_OCC 3
#endif
#ifndef MOMS
#define MOMS 5
#endif
template <int MODE, bool HIST2 = false, bool BF = false, int OCC = (BF ? CFDH_BF_OCC : CFDH_ASM_OCC)>
__global__ __launch_bounds__(CFDH_MAX_INC, OCC) void asm_kernel(AsmArgs p) {
  constexpr bool JAC = (MODE != 0);
  constexpr bool WJ = (MODE == 1);
  __shared__ double2 sX[CFDH_MAX_BV], sU[CFDH_MAX_BV], sUn[CFDH_MAX_BV];
  __shared__ double2 sUn2[HIST2 ? CFDH_MAX_BV : 1];
  __shared__ double sP[CFDH_MAX_BV];
  __shared__ int sVid[CFDH_MAX_BV];
  __shared__ unsigned char sFl[CFDH_MAX_BV];
  __shared__ double2 sMom[CFDH_MAX_BC * MOMS];  // 64-B records at a stride of 80 B: with stride 64 the k-th quarter of every record
                                                // falls into the same 4 of the 64 LDS banks (16-way conflict on every read)
  __shared__ unsigned char sCf[CFDH_MAX_BC];
  __shared__ int sRow[CFDH_MAX_ROWS + 1];
  const int t = threadIdx.x, blk = blockIdx.x;
#ifdef CFDH_ASM_TIMING
  long long ts[6];
  ts[0] = __builtin_readcyclecounter();
#endif
  const int row0 = p.blk_row[blk], row1 = p.blk_row[blk + 1];
  const int nrows = row1 - row0;
  const int nvo = p.nvo;
  // the lane's own incidence record does not depend on the staged data: requested first, so that its latency
  // overlaps the staging loads instead of following the barrier
  const size_t lk = (size_t)blk * CFDH_MAX_INC + t;
  const unsigned loc = p.inc_loc[lk], meta = p.inc_slot[lk], seg = p.inc_rank[lk];
  const int wmax = p.wave_maxlen[blk * (CFDH_MAX_INC / 64) + (t >> 6)];
  {
    const int v0 = p.blk_vptr[blk], nvl = p.blk_vptr[blk + 1] - v0;
    for (int i = t; i < nvl; i += CFDH_MAX_INC) {
      const int vid = p.blk_vlist[v0 + i];
      const int uo = uoff(vid, nvo), po = poff(vid, nvo);
      sVid[i] = vid;
      sX[i] = *(const double2 *)(p.coords + 2 * (size_t)vid);
      sU[i] = make_double2(p.x[uo], p.x[uo + 1]);
      sP[i] = p.x[po];
      sUn[i] = make_double2(p.un[uo], p.un[uo + 1]);
      if (HIST2) sUn2[i] = make_double2(p.un2[uo], p.un2[uo + 1]);
      sFl[i] = p.bcflag[vid];
    }
    const int c0 = p.blk_cptr[blk], ncl = p.blk_cptr[blk + 1] - c0;
    for (int i = t; i < 4 * ncl; i += CFDH_MAX_INC) {  // 4 x 16 B per cell record, consecutive lanes
      const int cid = p.blk_clist[c0 + (i >> 2)];
      sMom[MOMS * (i >> 2) + (i & 3)] = *(const double2 *)(p.mom + 8 * (size_t)cid + 2 * (i & 3));
    }
    for (int i = t; i < ncl; i += CFDH_MAX_INC) sCf[i] = p.cflag[p.blk_clist[c0 + i]];
    for (int i = t; i <= nrows; i += CFDH_MAX_INC) sRow[i] = p.vptr[row0 + i];
  }
#ifdef CFDH_ASM_TIMING
  ts[1] = __builtin_readcyclecounter();
#endif
  __syncthreads();
#ifdef CFDH_ASM_TIMING
  ts[2] = __builtin_readcyclecounter();
#endif

  const bool active = loc != 0xFFFFFFFFu;
  double Fr[3] = {0, 0, 0};
  double J00[3][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}}, J01[3][2] = {{0, 0}, {0, 0}, {0, 0}},
         J10[3][2] = {{0, 0}, {0, 0}, {0, 0}}, J11[3] = {0, 0, 0};
  int row = 0;
  unsigned fl0 = 0;
  double xrow[3] = {0, 0, 0};
  if (active) {
    const int a = (meta >> 24) & 3;
    const int lc = loc & 255;
    const int lv[3] = {(int)((loc >> 8) & 255), (int)((loc >> 16) & 255), (int)((loc >> 24) & 255)};
    const int vv[3] = {sVid[lv[0]], sVid[lv[1]], sVid[lv[2]]};
    const int v0 = vv[0], v1 = vv[1], v2 = vv[2];
    row = v0;
    const unsigned cfraw = sCf[lc];
    unsigned cf = cfraw & 7u, cb = cfraw >> 3;
    cf = ((cf >> a) | (cf << (3 - a))) & 7u;
    cb = ((cb >> a) | (cb << (3 - a))) & 7u;
    if (!p.ds_terms) cf = 0;
    // moments, rotated
    const double2 m0 = sMom[MOMS * lc], m1 = sMom[MOMS * lc + 1], m2 = sMom[MOMS * lc + 2], m3 = sMom[MOMS * lc + 3];
    const double o00 = m0.x, o01 = m0.y, o02 = m1.x, o11 = m1.y, o12 = m2.x, o22 = m2.y, Lm = m3.x;
    double M[3][3];
    M[0][0] = a == 0 ? o00 : (a == 1 ? o11 : o22);
    M[1][1] = a == 0 ? o11 : (a == 1 ? o22 : o00);
    M[2][2] = a == 0 ? o22 : (a == 1 ? o00 : o11);
    M[0][1] = M[1][0] = a == 0 ? o01 : (a == 1 ? o12 : o02);
    M[0][2] = M[2][0] = a == 0 ? o02 : (a == 1 ? o01 : o12);
    M[1][2] = M[2][1] = a == 0 ? o12 : (a == 1 ? o02 : o01);
    // geometry and nodal values from LDS
    double X[3][2], ue[3][2], une[3][2], pe[3];
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const double2 xx = sX[lv[b]], uu = sU[lv[b]], un2 = sUn[lv[b]];
      X[b][0] = xx.x; X[b][1] = xx.y;
      ue[b][0] = uu.x; ue[b][1] = uu.y; pe[b] = sP[lv[b]];
      une[b][0] = un2.x; une[b][1] = un2.y;
    }
    const double det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
    const double idet = 1.0 / det;
    double g[3][2];
    g[0][0] = (X[1][1] - X[2][1]) * idet; g[0][1] = (X[2][0] - X[1][0]) * idet;
    g[1][0] = (X[2][1] - X[0][1]) * idet; g[1][1] = (X[0][0] - X[2][0]) * idet;
    g[2][0] = (X[0][1] - X[1][1]) * idet; g[2][1] = (X[1][0] - X[0][0]) * idet;
    const double area = 0.5 * fabs(det);
    const double rho = p.rho, mu = p.mu, idt = 1.0 / p.dt;
    const double th = p.theta, a0idt = p.a0 * idt;
    double ub[3][2], w[3][2], G[2][2] = {{0, 0}, {0, 0}}, gp[2] = {0, 0};
#pragma unroll
    for (int b = 0; b < 3; b++) {
      double h2[2] = {0, 0};
      if (HIST2) {
        const double2 q = sUn2[lv[b]];
        h2[0] = p.a2 * q.x; h2[1] = p.a2 * q.y;
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        ub[b][i] = th * ue[b][i] + (1.0 - th) * une[b][i];
        w[b][i] = (p.a0 * ue[b][i] + p.a1 * une[b][i] + h2[i]) * idt;
      }
    }
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
      for (int i = 0; i < 2; i++) {
        gp[i] += pe[b] * g[b][i];
#pragma unroll
        for (int j = 0; j < 2; j++) G[i][j] += g[b][i] * ub[b][j];
      }
    const double divu = G[0][0] + G[1][1];
    double Cn[3][2], R[3][2], beta[3][3], mt[3], Q[3][2];
    const double ff[2] = {p.fx, p.fy};
#pragma unroll
    for (int b = 0; b < 3; b++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        Cn[b][j] = ub[b][0] * G[0][j] + ub[b][1] * G[1][j];
        R[b][j] = rho * (w[b][j] + Cn[b][j]) + gp[j] - rho * ff[j];
      }
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int b = 0; b < 3; b++) beta[d][b] = ub[d][0] * g[b][0] + ub[d][1] * g[b][1];
    double T = 0;
#pragma unroll
    for (int b = 0; b < 3; b++) { mt[b] = M[b][0] + M[b][1] + M[b][2]; T += mt[b]; }
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
      for (int i = 0; i < 2; i++) Q[d][i] = M[0][d] * R[0][i] + M[1][d] * R[1][i] + M[2][d] * R[2][i];
    const double E01 = 0.5 * (G[0][1] + G[1][0]);
    const double E[2][2] = {{G[0][0], E01}, {E01, G[1][1]}};
    const double pbar = (pe[0] + pe[1] + pe[2]) * (1.0 / 3.0);
    const double mab0[3] = {area * (2.0 / 12.0), area * (1.0 / 12.0), area * (1.0 / 12.0)};
    // ---- residual rows of local vertex 0
#pragma unroll
    for (int i = 0; i < 2; i++) {
      double v = 0;
#pragma unroll
      for (int b = 0; b < 3; b++) v += rho * mab0[b] * (w[b][i] + Cn[b][i]);
      v -= rho * ff[i] * area * (1.0 / 3.0);
      v += area * (2.0 * mu * (E[i][0] * g[0][0] + E[i][1] * g[0][1]) - pbar * g[0][i]);
#pragma unroll
      for (int d = 0; d < 3; d++) v += beta[d][0] * Q[d][i];
      v += rho * Lm * divu * g[0][i];
      Fr[i] = v;
    }
    {
      double v = area * (1.0 / 3.0) * divu;
#pragma unroll
      for (int b = 0; b < 3; b++) v += mt[b] * (R[b][0] * g[0][0] + R[b][1] * g[0][1]) / rho;
      Fr[2] = v;
    }
    // ---- Jacobian row block of local vertex 0
    if (JAC) {
      double MBa[3], mtB[3];
#pragma unroll
      for (int cidx = 0; cidx < 3; cidx++) MBa[cidx] = M[cidx][0] * beta[0][0] + M[cidx][1] * beta[1][0] + M[cidx][2] * beta[2][0];
#pragma unroll
      for (int b = 0; b < 3; b++) mtB[b] = mt[0] * beta[0][b] + mt[1] * beta[1][b] + mt[2] * beta[2][b];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const double mBa = mab0[0] * beta[0][b] + mab0[1] * beta[1][b] + mab0[2] * beta[2][b];
        const double BMBa = beta[0][b] * MBa[0] + beta[1][b] * MBa[1] + beta[2][b] * MBa[2];
        const double gg0b = g[0][0] * g[b][0] + g[0][1] * g[b][1];
#pragma unroll
        for (int i = 0; i < 2; i++) {
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const double dij = (i == j) ? 1.0 : 0.0;
            double v = rho * mab0[b] * dij * a0idt;
            v += rho * th * (mab0[b] * G[j][i] + dij * mBa);
            v += area * mu * th * (g[b][i] * g[0][j] + gg0b * dij);
            v += rho * ((dij * a0idt + th * G[j][i]) * MBa[b] + th * dij * BMBa);
            v += th * g[0][j] * Q[b][i];
            v += rho * Lm * th * g[b][j] * g[0][i];
            J00[b][i][j] = v;
          }
          J01[b][i] = -area * (1.0 / 3.0) * g[0][i] + g[b][i] * mtB[0];
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const double Gg = G[j][0] * g[0][0] + G[j][1] * g[0][1];
          J10[b][j] = area * (1.0 / 3.0) * th * g[b][j] + mt[b] * (g[0][j] * a0idt + th * Gg) + th * g[0][j] * mtB[b];
        }
        J11[b] = T * gg0b / rho;
      }
    }
    // ---- backflow stabilisation on outlet facets that contain local vertex 0:
    //      F -= beta rho oint (u_prev.n)_- (ubar.v), 2-point Gauss (stabilized_schur_backflow.py:165-176)
    if (BF && (cb & 6u)) {
#pragma unroll
      for (int f = 1; f < 3; f++) {
        if (!((cb >> f) & 1u)) continue;
        const int other = (f == 1) ? 2 : 1;
        const double gl = hypot(g[f][0], g[f][1]);
        const double n[2] = {-g[f][0] / gl, -g[f][1] / gl};
        const double elen = 2.0 * area * gl;
        // nodal values re-read from LDS (keeps the register live ranges of the element algebra short)
        const double2 un0 = sUn[lv[0]], uno = sUn[lv[other]];
        const double s0 = un0.x * n[0] + un0.y * n[1], so = uno.x * n[0] + uno.y * n[1];
        const double gq = 0.28867513459481288;  // 1/(2 sqrt 3)
        double w00 = 0.0, w0o = 0.0;  // sum_q c_q l0 l0, sum_q c_q l0 lo
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const double t = q == 0 ? 0.5 - gq : 0.5 + gq;
          const double l0 = 1.0 - t, lo = t;
          const double sq = l0 * s0 + lo * so;
          const double cq = p.beta_bf * 0.25 * (sq - fabs(sq)) * elen;
          w00 += cq * l0 * l0;
          w0o += cq * l0 * lo;
        }
        const double2 u0 = sU[lv[0]], uo = sU[lv[other]];
        Fr[0] -= w00 * (th * u0.x + (1.0 - th) * un0.x) + w0o * (th * uo.x + (1.0 - th) * uno.x);
        Fr[1] -= w00 * (th * u0.y + (1.0 - th) * un0.y) + w0o * (th * uo.y + (1.0 - th) * uno.y);
        if (JAC) {
          J00[0][0][0] -= th * w00; J00[0][1][1] -= th * w00;
          J00[other][0][0] -= th * w0o; J00[other][1][1] -= th * w0o;
        }
      }
    }
    // ---- exterior facets that contain local vertex 0 (facets 1 and 2)
    if (cf & 6u) {
#pragma unroll
      for (int f = 1; f < 3; f++) {
        if (!((cf >> f) & 1u)) continue;
        const int other = (f == 1) ? 2 : 1;
        const double gl = hypot(g[f][0], g[f][1]);
        const double n[2] = {-g[f][0] / gl, -g[f][1] / gl};
        const double elen = 2.0 * area * gl;
        const double pint = (2.0 * pe[0] + pe[other]) * (1.0 / 6.0);
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const double Gn = G[i][0] * n[0] + G[i][1] * n[1];
          Fr[i] += n[i] * elen * pint - p.muf * Gn * elen * 0.5;
          if (JAC) {
            J01[0][i] += n[i] * elen * (2.0 / 6.0);
            J01[other][i] += n[i] * elen * (1.0 / 6.0);
#pragma unroll
            for (int b = 0; b < 3; b++)
#pragma unroll
              for (int j = 0; j < 2; j++) J00[b][i][j] -= p.muf * (0.5 * th) * g[b][i] * n[j] * elen;
          }
        }
      }
    }
    // ---- Dirichlet: lifting F += J[:,bc](g - x), zero bc columns and rows
    fl0 = sFl[lv[0]];
    const unsigned fl1 = sFl[lv[1]], fl2 = sFl[lv[2]];
    (void)v1; (void)v2;
    xrow[0] = ue[0][0]; xrow[1] = ue[0][1]; xrow[2] = pe[0];
    if (fl0 | fl1 | fl2) {
      const unsigned flb[3] = {fl0, fl1, fl2};
#pragma unroll
      for (int b = 0; b < 3; b++) {
        if (!flb[b]) continue;
#pragma unroll
        for (int j = 0; j < 2; j++)
          if ((flb[b] >> j) & 1u) {
            if (JAC) {
              const double gx = p.bcval[3 * vv[b] + j] - ue[b][j];
              if (MODE == 2 || gx != 0.0) { Fr[0] += J00[b][0][j] * gx; Fr[1] += J00[b][1][j] * gx; Fr[2] += J10[b][j] * gx; }
              J00[b][0][j] = 0.0; J00[b][1][j] = 0.0; J10[b][j] = 0.0;
            }
          }
        if (flb[b] & 4u) {
          if (JAC) {
            const double gx = p.bcval[3 * vv[b] + 2] - pe[b];
            if (MODE == 2 || gx != 0.0) { Fr[0] += J01[b][0] * gx; Fr[1] += J01[b][1] * gx; Fr[2] += J11[b] * gx; }
            J01[b][0] = 0.0; J01[b][1] = 0.0; J11[b] = 0.0;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
        if ((fl0 >> i) & 1u) {
          Fr[i] = 0.0;
          if (JAC) {
#pragma unroll
            for (int b = 0; b < 3; b++) { J00[b][i][0] = 0.0; J00[b][i][1] = 0.0; J01[b][i] = 0.0; }
          }
        }
      if (fl0 & 4u) {
        Fr[2] = 0.0;
        if (JAC) {
#pragma unroll
          for (int b = 0; b < 3; b++) { J10[b][0] = 0.0; J10[b][1] = 0.0; J11[b] = 0.0; }
        }
      }
    }
  }
#ifdef CFDH_ASM_TIMING
  ts[3] = __builtin_readcyclecounter();
#endif
  // ---- wavefront-level segmented reduction (all 64 lanes take part; idle lanes carry zeros)
  const int lane = t & 63;
  const int seg_pos = seg & 255, seg_len = (seg >> 8) & 255, prev_off = (int)((seg >> 16) & 255) - 64;
  const bool has_prev = (meta >> 27) & 1u, emit_v2 = (meta >> 26) & 1u;
  if (WJ) {
    // off-diagonal block of the edge (i, v1): this cell's B1 plus B2 of the previous cell of the fan
    const int src = lane + (has_prev ? prev_off : 0);
#pragma unroll
    for (int i = 0; i < 2; i++) {
#pragma unroll
      for (int j = 0; j < 2; j++) { const double v = __shfl(J00[2][i][j], src); if (has_prev) J00[1][i][j] += v; }
      { const double v = __shfl(J01[2][i], src); if (has_prev) J01[1][i] += v; }
      { const double v = __shfl(J10[2][i], src); if (has_prev) J10[1][i] += v; }
    }
    { const double v = __shfl(J11[2], src); if (has_prev) J11[1] += v; }
  }
  // diagonal block and residual: sum over the lanes of the row, result in its first lane
  for (int off = 1; off < wmax; off <<= 1) {
    const bool take = active && (seg_pos + off < seg_len);
#pragma unroll
    for (int i = 0; i < 3; i++) { const double v = __shfl_down(Fr[i], off); if (take) Fr[i] += v; }
    if (WJ) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int j = 0; j < 2; j++) { const double v = __shfl_down(J00[0][i][j], off); if (take) J00[0][i][j] += v; }
        { const double v = __shfl_down(J01[0][i], off); if (take) J01[0][i] += v; }
        { const double v = __shfl_down(J10[0][i], off); if (take) J10[0][i] += v; }
      }
      { const double v = __shfl_down(J11[0], off); if (take) J11[0] += v; }
    }
  }
#ifdef CFDH_ASM_TIMING
  ts[4] = __builtin_readcyclecounter();
#endif
  // ---- every block of the row now sits complete in exactly one lane: plain stores
  if (active) {
    const size_t sb = (size_t)sRow[row - row0];
    if (WJ) {
      {
        const size_t s1 = sb + ((meta >> 8) & 255);
        *(double2 *)(p.A00 + 4 * s1) = make_double2(J00[1][0][0], J00[1][0][1]);
        *(double2 *)(p.A00 + 4 * s1 + 2) = make_double2(J00[1][1][0], J00[1][1][1]);
        *(double2 *)(p.A01 + 2 * s1) = make_double2(J01[1][0], J01[1][1]);
        *(double2 *)(p.A10 + 2 * s1) = make_double2(J10[1][0], J10[1][1]);
        p.A11[s1] = J11[1];
      }
      if (emit_v2) {
        const size_t s2 = sb + ((meta >> 16) & 255);
        *(double2 *)(p.A00 + 4 * s2) = make_double2(J00[2][0][0], J00[2][0][1]);
        *(double2 *)(p.A00 + 4 * s2 + 2) = make_double2(J00[2][1][0], J00[2][1][1]);
        *(double2 *)(p.A01 + 2 * s2) = make_double2(J01[2][0], J01[2][1]);
        *(double2 *)(p.A10 + 2 * s2) = make_double2(J10[2][0], J10[2][1]);
        p.A11[s2] = J11[2];
      }
    }
    if (seg_pos == 0) {
      // Dirichlet rows: diagonal = number of bc objects, F = x - g (everything else in the row is zero)
      if (fl0) {
        if (fl0 & 1u) { J00[0][0][0] = p.bcmult[3 * row]; Fr[0] = xrow[0] - p.bcval[3 * row]; }
        if (fl0 & 2u) { J00[0][1][1] = p.bcmult[3 * row + 1]; Fr[1] = xrow[1] - p.bcval[3 * row + 1]; }
        if (fl0 & 4u) { J11[0] = p.bcmult[3 * row + 2]; Fr[2] = xrow[2] - p.bcval[3 * row + 2]; }
      }
      if (WJ) {
        const size_t s0 = sb + (meta & 255);
        *(double2 *)(p.A00 + 4 * s0) = make_double2(J00[0][0][0], J00[0][0][1]);
        *(double2 *)(p.A00 + 4 * s0 + 2) = make_double2(J00[0][1][0], J00[0][1][1]);
        *(double2 *)(p.A01 + 2 * s0) = make_double2(J01[0][0], J01[0][1]);
        *(double2 *)(p.A10 + 2 * s0) = make_double2(J10[0][0], J10[0][1]);
        p.A11[s0] = J11[0];
      }
      *(double2 *)(p.F + 2 * (size_t)row) = make_double2(Fr[0], Fr[1]);
      p.F[2 * (size_t)nvo + row] = Fr[2];
    }
  }
#ifdef CFDH_ASM_TIMING
  __builtin_amdgcn_s_waitcnt(0);
  ts[5] = __builtin_readcyclecounter();
  if (t == 0 && p.dbg)
    for (int k = 0; k < 6; k++) p.dbg[8 * (size_t)blk + k] = ts[k];
#endif
}

int k_assemble(cfdh_ctx *c, const double *xstate, int mode) {
  if (c->gen) return c->dim == 3 ? kg3_assemble(c, xstate, mode) : kg_assemble(c, xstate, mode);
  if (c->dim == 3) return k3_assemble(c, xstate, mode);
  AsmArgs a;
  a.coords = c->coords.p; a.mom = c->mom.p; a.x = xstate; a.un = c->xprev.p; a.un2 = c->xprev2.p; a.bcval = c->bcval.p; a.bcmult = c->bcmult.p;
  a.vptr = c->vptr.p;
  a.blk_vptr = c->blk_vptr.p; a.blk_vlist = c->blk_vlist.p; a.blk_cptr = c->blk_cptr.p; a.blk_clist = c->blk_clist.p; a.inc_loc = c->inc_loc.p; a.wave_maxlen = c->wave_maxlen.p;
  a.blk_row = c->blk_row.p;
  a.inc_slot = c->inc_slot.p; a.inc_rank = c->inc_rank.p; a.cflag = c->cflag.p; a.bcflag = c->bcflag.p;
  a.A00 = c->A00.p; a.A01 = c->A01.p; a.A10 = c->A10.p; a.A11 = c->A11.p; a.F = c->F.p;
  a.nvo = c->nvo; a.dt = c->dt; a.rho = c->rho; a.mu = c->mu; a.muf = c->muf; a.fx = c->f[0]; a.fy = c->f[1];
  a.theta = c->ts_theta; a.a0 = c->ts_a[0]; a.a1 = c->ts_a[1]; a.a2 = c->ts_a[2];
  a.beta_bf = c->bf_beta * c->rho; a.ds_terms = c->ds_terms ? 1 : 0;
  const bool hist2 = c->ts_a[2] != 0.0;
#ifdef CFDH_ASM_TIMING
  static long long *dbg = nullptr;
  static int dbg_calls = 0;
  if (!dbg) hipMalloc(&dbg, sizeof(long long) * 8 * (size_t)c->nblk);
  a.dbg = dbg;
#endif
  prof_begin(c, 0);
  const dim3 gr(c->nblk), bl(CFDH_MAX_INC);
  const bool bf = c->bf_beta > 0.0 && c->bf_marker >= 0;
#define CFDH_ASM_LAUNCH(H2, BFV)                                                                   \
  do {                                                                                             \
    if (mode == 1) hipLaunchKernelGGL((asm_kernel<1, H2, BFV>), gr, bl, 0, c->stream, a);         \
    else if (mode == 2) hipLaunchKernelGGL((asm_kernel<2, H2, BFV>), gr, bl, 0, c->stream, a);    \
    else hipLaunchKernelGGL((asm_kernel<0, H2, BFV>), gr, bl, 0, c->stream, a);                   \
  } while (0)
  if (!hist2 && !bf) CFDH_ASM_LAUNCH(false, false);
  else if (hist2 && !bf) CFDH_ASM_LAUNCH(true, false);
  else if (!hist2 && bf) CFDH_ASM_LAUNCH(false, true);
  else CFDH_ASM_LAUNCH(true, true);
#undef CFDH_ASM_LAUNCH
  prof_end(c, 0);
  HIPCHK(c, hipGetLastError());
#ifdef CFDH_ASM_TIMING
  if (mode == 1 && (++dbg_calls % 20) == 0) {
    std::vector<long long> h(8 * (size_t)c->nblk);
    hipStreamSynchronize(c->stream);
    hipMemcpy(h.data(), dbg, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double acc[5] = {0, 0, 0, 0, 0};
    long long tmin = h[0], tmax = h[5];
    for (int b = 0; b < c->nblk; b++) {
      for (int k = 0; k < 5; k++) acc[k] += (double)(h[8 * (size_t)b + k + 1] - h[8 * (size_t)b + k]);
      tmin = std::min(tmin, h[8 * (size_t)b]); tmax = std::max(tmax, h[8 * (size_t)b + 5]);
    }
    fprintf(stderr, "[asm timing] ticks per block: stage-issue %.0f, barrier %.0f, element %.0f, reduce %.0f, stores+drain %.0f; kernel span %lld ticks, %d blocks\n",
            acc[0] / c->nblk, acc[1] / c->nblk, acc[2] / c->nblk, acc[3] / c->nblk, acc[4] / c->nblk, tmax - tmin, c->nblk);
  }
#endif
  if (mode == 1) c->jac_valid = true;
  return 0;
}

// ---------------------------------------------------------------- block SpMV
// 8 lanes per vertex row: lane l of a group takes block k = rowstart + l (+8,...),
// so the value arrays are streamed fully coalesced across the wave; partial
// sums are combined with DPP shuffles.  y = J x over the monolithic vector.
__global__ __launch_bounds__(TPB) void spmv_full_kernel(int nvo, const int *__restrict__ vptr,
                                                        const int *__restrict__ vcol, const double *__restrict__ A00,
                                                        const double *__restrict__ A01, const double *__restrict__ A10,
                                                        const double *__restrict__ A11, const double *__restrict__ x,
                                                        double *__restrict__ y) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0, a2 = 0;
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int w = vcol[k];
      const int uo = uoff(w, nvo), po = poff(w, nvo);
      const double xu0 = x[uo], xu1 = x[uo + 1], xp = x[po];
      const double2 b0 = *(const double2 *)(A00 + 4 * (size_t)k), b1 = *(const double2 *)(A00 + 4 * (size_t)k + 2);
      const double2 c01 = *(const double2 *)(A01 + 2 * (size_t)k), c10 = *(const double2 *)(A10 + 2 * (size_t)k);
      const double c11 = A11[k];
      a0 += b0.x * xu0 + b0.y * xu1 + c01.x * xp;
      a1 += b1.x * xu0 + b1.y * xu1 + c01.y * xp;
      a2 += c10.x * xu0 + c10.y * xu1 + c11 * xp;
    }
  }
  a0 = group8_sum(a0); a1 = group8_sum(a1); a2 = group8_sum(a2);
  if (row < nvo && l == 0) {
    *(double2 *)(y + 2 * (size_t)row) = make_double2(a0, a1);
    y[2 * (size_t)nvo + row] = a2;
  }
}


// The same product for NV vectors at once (leading dimension ld): the Jacobian is read once -- the projected initial guess of
// the linear solves multiplies its 2-4 kept corrections with the current matrix (cfdh_solver.cpp::guess_project).
template <int NV>
__global__ __launch_bounds__(TPB) void spmv_full_multi_kernel(int nvo, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                              const double *__restrict__ A00, const double *__restrict__ A01,
                                                              const double *__restrict__ A10, const double *__restrict__ A11,
                                                              const double *__restrict__ X, double *__restrict__ Y, size_t ld) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0[NV], a1[NV], a2[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) a0[v] = a1[v] = a2[v] = 0.0;
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int w = vcol[k];
      const int uo = uoff(w, nvo), po = poff(w, nvo);
      const double2 b0 = *(const double2 *)(A00 + 4 * (size_t)k), b1 = *(const double2 *)(A00 + 4 * (size_t)k + 2);
      const double2 c01 = *(const double2 *)(A01 + 2 * (size_t)k), c10 = *(const double2 *)(A10 + 2 * (size_t)k);
      const double c11 = A11[k];
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double *x = X + (size_t)v * ld;
        const double xu0 = x[uo], xu1 = x[uo + 1], xp = x[po];
        a0[v] += b0.x * xu0 + b0.y * xu1 + c01.x * xp;
        a1[v] += b1.x * xu0 + b1.y * xu1 + c01.y * xp;
        a2[v] += c10.x * xu0 + c10.y * xu1 + c11 * xp;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const double s0 = group8_sum(a0[v]), s1 = group8_sum(a1[v]), s2 = group8_sum(a2[v]);
    if (row < nvo && l == 0) {
      double *y = Y + (size_t)v * ld;
      *(double2 *)(y + 2 * (size_t)row) = make_double2(s0, s1);
      y[2 * (size_t)nvo + row] = s2;
    }
  }
}

// Y_v = J X_v for v < nvec (vectors ld apart; ghost tails of X filled by the caller)
int k_spmv_full_multi(cfdh_ctx *c, const double *X, double *Y, int ld, int nvec) {
  if (c->dim == 3) return k3_spmv_full_multi(c, X, Y, ld, nvec);
  if (nvec < 2 || nvec > 4) {
    for (int v = 0; v < nvec; v++) CHK(k_spmv_full(c, X + (size_t)v * ld, Y + (size_t)v * ld));
    return 0;
  }
  const long long nthreads = 8ll * c->nvo;
  const dim3 gr((unsigned)((nthreads + TPB - 1) / TPB)), bl(TPB);
#define CFDH_SPMM(NV) hipLaunchKernelGGL((spmv_full_multi_kernel<NV>), gr, bl, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A00.p, \
                                         c->A01.p, c->A10.p, c->A11.p, X, Y, (size_t)ld)
  if (nvec == 2) CFDH_SPMM(2); else if (nvec == 3) CFDH_SPMM(3); else CFDH_SPMM(4);
#undef CFDH_SPMM
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_spmv_full(cfdh_ctx *c, const double *x, double *y) {
  if (c->dim == 3) return k3_spmv_full(c, x, y);
  const long long nthreads = 8ll * c->nvo;
  prof_begin(c, 1);
  hipLaunchKernelGGL(spmv_full_kernel, dim3((unsigned)((nthreads + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, c->nvo,
                     c->vptr.p, c->vcol.p, c->A00.p, c->A01.p, c->A10.p, c->A11.p, x, y);
  prof_end(c, 1);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// sub-block products on owned columns only (the preconditioner is rank-local):
// BLK 1: y_u = A00 x_u, 2: y_u = A01 x_p, 3: y_p = A10 x_u, 4: y_p = A11 x_p;  MODE 1: y = b - A x
template <int BLK, int MODE>
__global__ __launch_bounds__(TPB) void spmv_blk_kernel(int nvo, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                       const double *__restrict__ A, const double *__restrict__ x,
                                                       double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0;
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int w = vcol[k];
      if (w >= nvo) continue;
      if (BLK == 1) {
        const double2 xx = *(const double2 *)(x + 2 * (size_t)w);
        const double2 b0 = *(const double2 *)(A + 4 * (size_t)k), b1 = *(const double2 *)(A + 4 * (size_t)k + 2);
        a0 += b0.x * xx.x + b0.y * xx.y; a1 += b1.x * xx.x + b1.y * xx.y;
      } else if (BLK == 2) {
        const double xp = x[w];
        const double2 cc = *(const double2 *)(A + 2 * (size_t)k);
        a0 += cc.x * xp; a1 += cc.y * xp;
      } else if (BLK == 3) {
        const double2 xx = *(const double2 *)(x + 2 * (size_t)w);
        const double2 cc = *(const double2 *)(A + 2 * (size_t)k);
        a0 += cc.x * xx.x + cc.y * xx.y;
      } else {
        a0 += A[k] * x[w];
      }
    }
  }
  a0 = group8_sum(a0);
  if (BLK <= 2) a1 = group8_sum(a1);
  if (row < nvo && l == 0) {
    if (BLK <= 2) {
      double2 o = make_double2(a0, a1);
      if (MODE == 1) { const double2 bb = *(const double2 *)(bvec + 2 * (size_t)row); o.x = bb.x - o.x; o.y = bb.y - o.y; }
      *(double2 *)(y + 2 * (size_t)row) = o;
    } else {
      y[row] = (MODE == 1) ? bvec[row] - a0 : a0;
    }
  }
}

// y_u = b_u - A01 x_p (the coupling product of the block-triangular preconditioner, once per FGMRES iteration) with four
// lanes per row and the first two entries of every lane requested together: 16 B of matrix per entry is too little per
// load for the 8-lane scheme, which left this kernel at 3.6 TB/s (16.1 us for 58 MB; this form: 12.2 us.  The same
// change does nothing for the full product, whose lanes already carry 72 B of matrix per entry: 37.3 vs 36.6 us)
__global__ __launch_bounds__(TPB) void spmv_a01_resid_kernel(int nvo, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                             const double *__restrict__ A, const double *__restrict__ x,
                                                             double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 2, l = gid & 3;
  double a0 = 0, a1 = 0;
  double2 bb = make_double2(0.0, 0.0);
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    if (l == 0) bb = *(const double2 *)(bvec + 2 * (size_t)row);
    const int k0 = ks + l, k1 = k0 + 4;
    const bool h0 = k0 < ke, h1 = k1 < ke;
    const int q0 = h0 ? k0 : ks, q1 = h1 ? k1 : ks;  // ks is always a valid entry (the diagonal block exists)
    const int w0 = vcol[q0], w1 = vcol[q1];
    const double2 c0 = *(const double2 *)(A + 2 * (size_t)q0), c1 = *(const double2 *)(A + 2 * (size_t)q1);
    const double x0 = (h0 && w0 < nvo) ? x[w0] : 0.0, x1 = (h1 && w1 < nvo) ? x[w1] : 0.0;
    a0 = c0.x * x0 + c1.x * x1;
    a1 = c0.y * x0 + c1.y * x1;
    for (int k = k1 + 4; k < ke; k += 4) {
      const int w = vcol[k];
      if (w >= nvo) continue;
      const double xp = x[w];
      const double2 cc = *(const double2 *)(A + 2 * (size_t)k);
      a0 += cc.x * xp; a1 += cc.y * xp;
    }
  }
  a0 = quad_sum(a0); a1 = quad_sum(a1);
  if (row < nvo && l == 0) *(double2 *)(y + 2 * (size_t)row) = make_double2(bb.x - a0, bb.y - a1);
}

// coupling blocks INCLUDING ghost columns: xv is a full vector in the [u | p | ghost triplets] layout whose
// ghost tail was refreshed by comm_halo.  GB 2: y_u = b_u - A01 x_p ; GB 3: y_p = b_p - A10 x_u.
template <int GB>
__global__ __launch_bounds__(TPB) void spmv_blk_ghost_kernel(int nvo, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                             const double *__restrict__ A, const double *__restrict__ xv,
                                                             double *__restrict__ y, const double *__restrict__ bvec) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0;
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int w = vcol[k];
      const double2 cc = *(const double2 *)(A + 2 * (size_t)k);
      if (GB == 2) {
        const double xp = xv[poff(w, nvo)];
        a0 += cc.x * xp; a1 += cc.y * xp;
      } else {
        const int uo = uoff(w, nvo);
        a0 += cc.x * xv[uo] + cc.y * xv[uo + 1];
      }
    }
  }
  a0 = group8_sum(a0);
  if (GB == 2) a1 = group8_sum(a1);
  if (row < nvo && l == 0) {
    if (GB == 2) {
      const double2 bb = *(const double2 *)(bvec + 2 * (size_t)row);
      *(double2 *)(y + 2 * (size_t)row) = make_double2(bb.x - a0, bb.y - a1);
    } else {
      y[row] = bvec[row] - a0;
    }
  }
}
int k_spmv_block_ghost(cfdh_ctx *c, int blk, const double *xv, double *y, const double *b) {
  if (c->dim == 3) return k3_spmv_block_ghost(c, blk, xv, y, b);
  const long long nthreads = 8ll * c->nvo;
  dim3 grid((unsigned)((nthreads + TPB - 1) / TPB)), block(TPB);
  if (blk == 2) hipLaunchKernelGGL((spmv_blk_ghost_kernel<2>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A01.p, xv, y, b);
  else hipLaunchKernelGGL((spmv_blk_ghost_kernel<3>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A10.p, xv, y, b);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_spmv_block(cfdh_ctx *c, int blk, const double *x, double *y, const double *b, double /*alpha*/) {
  if (c->dim == 3) return k3_spmv_block(c, blk, x, y, b);
  const long long nthreads = 8ll * c->nvo;
  dim3 grid((unsigned)((nthreads + TPB - 1) / TPB)), block(TPB);
  const int mode = b ? 1 : 0;
#define LAUNCH_BLK(B, M, AP) hipLaunchKernelGGL((spmv_blk_kernel<B, M>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, AP, x, y, b)
  if (blk == 1) { if (mode) LAUNCH_BLK(1, 1, c->A00.p); else LAUNCH_BLK(1, 0, c->A00.p); }
  else if (blk == 2) {
    if (mode) {
      const long long n4 = 4ll * c->nvo;
      hipLaunchKernelGGL(spmv_a01_resid_kernel, dim3((unsigned)((n4 + TPB - 1) / TPB)), block, 0, c->stream, c->nvo, c->vptr.p,
                         c->vcol.p, c->A01.p, x, y, b);
    } else {
      LAUNCH_BLK(2, 0, c->A01.p);
    }
  }
  else if (blk == 3) { if (mode) LAUNCH_BLK(3, 1, c->A10.p); else LAUNCH_BLK(3, 0, c->A10.p); }
  else { if (mode) LAUNCH_BLK(4, 1, c->A11.p); else LAUNCH_BLK(4, 0, c->A11.p); }
#undef LAUNCH_BLK
  HIPCHK(c, hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(TPB) void extract_diag_kernel(int nvo, const int *__restrict__ vdiag,
                                                           const double *__restrict__ A00, double *__restrict__ dinv) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= nvo) return;
  const size_t k = (size_t)vdiag[row];
  const double d0 = A00[4 * k], d1 = A00[4 * k + 3];
  *(double2 *)(dinv + 2 * (size_t)row) = make_double2(1.0 / d0, 1.0 / d1);
}

int k_extract_diag(cfdh_ctx *c) {
  hipLaunchKernelGGL(extract_diag_kernel, dim3((c->nvo + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nvo, c->vdiag.p,
                     c->A00.p, c->dinvA.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- Chebyshev on D^-1 A00
// one step: r_out = r_in - A00 d_old ; d_new = c1 d_old + c2 D^-1 r_out ; x += d_new
// FIRST: fused with the zero-guess initialisation (d_old = D^-1 r_in / theta formed while gathering,
// x = d_old + d_new); LAST: r_out / d_new not written.  Coefficients come from device memory so that
// a captured graph of the preconditioner survives a refresh of the spectral bound of D^-1 A00.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(TPB) void cheb_a00_step_kernel(int nvo, const int *__restrict__ vptr,
                                                            const int *__restrict__ vcol, const double *__restrict__ A00,
                                                            const double *__restrict__ dinv, const double *__restrict__ rin,
                                                            double *__restrict__ rout, const double *__restrict__ dold,
                                                            double *__restrict__ dnew, double *__restrict__ x,
                                                            const double *__restrict__ coef, int kstep) {
  const double c1 = coef[2 * kstep], c2 = coef[2 * kstep + 1], itheta = coef[0];
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a0 = 0, a1 = 0;
  if (row < nvo) {
    const int ks = vptr[row], ke = vptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int w = vcol[k];
      if (w >= nvo) continue;
      double2 xx;
      if (FIRST) {
        const double2 dj = *(const double2 *)(dinv + 2 * (size_t)w), rj = *(const double2 *)(rin + 2 * (size_t)w);
        xx = make_double2(dj.x * rj.x * itheta, dj.y * rj.y * itheta);
      } else {
        xx = *(const double2 *)(dold + 2 * (size_t)w);
      }
      const double2 b0 = *(const double2 *)(A00 + 4 * (size_t)k), b1 = *(const double2 *)(A00 + 4 * (size_t)k + 2);
      a0 += b0.x * xx.x + b0.y * xx.y; a1 += b1.x * xx.x + b1.y * xx.y;
    }
  }
  a0 = group8_sum(a0); a1 = group8_sum(a1);
  if (row < nvo && l == 0) {
    const size_t o = 2 * (size_t)row;
    const double2 ri = *(const double2 *)(rin + o), di = *(const double2 *)(dinv + o);
    double2 dd;
    if (FIRST) dd = make_double2(di.x * ri.x * itheta, di.y * ri.y * itheta);
    else dd = *(const double2 *)(dold + o);
    const double r0 = ri.x - a0, r1 = ri.y - a1;
    const double n0 = c1 * dd.x + c2 * di.x * r0, n1 = c1 * dd.y + c2 * di.y * r1;
    if (!LAST) {
      *(double2 *)(rout + o) = make_double2(r0, r1);
      *(double2 *)(dnew + o) = make_double2(n0, n1);
    }
    if (FIRST) {
      *(double2 *)(x + o) = make_double2(dd.x + n0, dd.y + n1);
    } else {
      double2 xo = *(double2 *)(x + o);
      xo.x += n0; xo.y += n1;
      *(double2 *)(x + o) = xo;
    }
  }
}

// d0 = D^-1 b / theta ; x = d0
__global__ __launch_bounds__(TPB) void cheb_init_kernel(int n, const double *__restrict__ dinv, const double *__restrict__ b,
                                                        double *__restrict__ d0, double *__restrict__ x, double itheta,
                                                        int accumulate, const double *__restrict__ coef) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  if (coef) itheta = coef[0];
  const double v = dinv[i] * b[i] * itheta;
  d0[i] = v;
  x[i] = accumulate ? x[i] + v : v;
}

// Chebyshev coefficients of the A00 solve -> device (called whenever lmaxA changes)
int k_cheb_a00_coeffs(cfdh_ctx *c) {
  const double lmax = c->lmaxA, lmin = lmax / c->opt.cheb_ratio;
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  const double sigma = theta / delta;
  double rho = 1.0 / sigma;
  std::vector<double> h(2 * 16 + 2, 0.0);
  h[0] = 1.0 / theta;
  for (int k = 1; k < c->opt.cheb_degree && k < 16; k++) {
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    h[2 * k] = rho_new * rho; h[2 * k + 1] = 2.0 * rho_new / delta;
    rho = rho_new;
  }
  HIPCHK(c, c->cheb_coef.alloc(h.size()));
  HIPCHK(c, hipMemcpyAsync(c->cheb_coef.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // h is a host temporary
  return 0;
}

// x = Cheb_k(D^-1 A00) b with zero initial guess; lambda in [lmax/ratio, lmax]
int k_cheb_a00(cfdh_ctx *c, const double *b, double *x) {
  const int nu = 2 * c->nvo, deg = c->opt.cheb_degree;
  double *dold = c->pu1.p, *dnew = c->pu2.p, *r = c->pr.p;
  if (deg == 1) {
    hipLaunchKernelGGL(cheb_init_kernel, dim3((nu + TPB - 1) / TPB), dim3(TPB), 0, c->stream, nu, c->dinvA.p, b, dold, x,
                       0.0, 0, c->cheb_coef.p);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  const long long nthreads = 8ll * c->nvo;
  dim3 grid((unsigned)((nthreads + TPB - 1) / TPB)), block(TPB);
  for (int k = 1; k < deg; k++) {
    const bool first = (k == 1), last = (k == deg - 1);
    prof_begin(c, 3);
#define LAUNCH_A00(F, LST) hipLaunchKernelGGL((cheb_a00_step_kernel<F, LST>), grid, block, 0, c->stream, c->nvo, c->vptr.p, c->vcol.p, c->A00.p, c->dinvA.p, first ? b : r, r, dold, dnew, x, c->cheb_coef.p, k)
    if (first) { if (last) LAUNCH_A00(true, true); else LAUNCH_A00(true, false); }
    else { if (last) LAUNCH_A00(false, true); else LAUNCH_A00(false, false); }
#undef LAUNCH_A00
    prof_end(c, 3);
    std::swap(dold, dnew);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- scalar CSR operators (AMG levels)
// All level kernels are templated on the vector element T: double (one right-hand side) or
// double2 (two right-hand sides sharing one scalar operator: the two velocity components).
// three right-hand sides sharing one scalar operator: the velocity components of a tetrahedral mesh
struct d3 { double x, y, z; };
__device__ __forceinline__ d3 vzero(const d3 *) { return d3{0.0, 0.0, 0.0}; }
__device__ __forceinline__ d3 vfma(double a, d3 x, d3 acc) { return d3{acc.x + a * x.x, acc.y + a * x.y, acc.z + a * x.z}; }
__device__ __forceinline__ d3 vsub(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 vadd(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 vscale(double a, d3 x) { return d3{a * x.x, a * x.y, a * x.z}; }
__device__ __forceinline__ d3 g8(d3 v) { return d3{group8_sum(v.x), group8_sum(v.y), group8_sum(v.z)}; }
__device__ __forceinline__ d3 wsum(d3 v) { return d3{wave_sum(v.x), wave_sum(v.y), wave_sum(v.z)}; }
__device__ __forceinline__ double vzero(const double *) { return 0.0; }
__device__ __forceinline__ double2 vzero(const double2 *) { return make_double2(0.0, 0.0); }
__device__ __forceinline__ double vfma(double a, double x, double acc) { return acc + a * x; }
__device__ __forceinline__ double2 vfma(double a, double2 x, double2 acc) { return make_double2(acc.x + a * x.x, acc.y + a * x.y); }
__device__ __forceinline__ double vsub(double a, double b) { return a - b; }
__device__ __forceinline__ double2 vsub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double vadd(double a, double b) { return a + b; }
__device__ __forceinline__ double2 vadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double vscale(double a, double x) { return a * x; }
__device__ __forceinline__ double2 vscale(double a, double2 x) { return make_double2(a * x.x, a * x.y); }
__device__ __forceinline__ double g8(double v) { return group8_sum(v); }
__device__ __forceinline__ double2 g8(double2 v) { return make_double2(group8_sum(v.x), group8_sum(v.y)); }
__device__ __forceinline__ double wsum(double v) { return wave_sum(v); }
__device__ __forceinline__ double2 wsum(double2 v) { return make_double2(wave_sum(v.x), wave_sum(v.y)); }

// One row of a SELL-64 matrix times x, latency-oriented: at ~1 M DOF a sweep is one wave of work per SIMD lane group and
// 85 % of a wave's life is spent waiting on memory (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r02_pmc_sq_tcc.json), in a
// chain  columns -> gathers -> next columns ...  With all column/value loads of the row issued first, then all gathers,
// the chain is three round trips whatever the row length.  The slice width is wave-uniform: the guards are scalar branches.
template <int MAXW, typename T>
__device__ __forceinline__ T sell_row_dot(const int *__restrict__ sp, const int *__restrict__ sc, const float *__restrict__ sv,
                                          const T *__restrict__ x, int sl, int lane, T a) {
  sl = __builtin_amdgcn_readfirstlane(sl);  // wave-uniform: slice pointers and width live in scalar registers
  const int p0 = sp[sl], w = (sp[sl + 1] - p0) >> 6;
  int cidx[MAXW];
  float cval[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; k++)
    if (k < w) { const int p = p0 + k * 64 + lane; cidx[k] = NTLOAD(sc + p); cval[k] = NTLOAD(sv + p); }
  T g[MAXW];
#pragma unroll
  for (int k = 0; k < MAXW; k++)
    if (k < w) g[k] = x[cidx[k]];
#pragma unroll
  for (int k = 0; k < MAXW; k++)
    if (k < w) a = vfma((double)cval[k], g[k], a);
  for (int k = MAXW; k < w; k++) { const int p = p0 + k * 64 + lane; a = vfma((double)sv[p], x[sc[p]], a); }
  return a;
}

// MODE 0: y = A x; 1: y = b - A x; 2: y += A x; 3: y = b + A x
template <int MODE, typename T>
__global__ __launch_bounds__(TPB) void csr_spmv_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                       const double *__restrict__ val, const T *__restrict__ x,
                                                       T *__restrict__ y, const T *__restrict__ b) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) a = vfma(val[k], x[col[k]], a);
  }
  a = g8(a);
  if (row < n && l == 0) {
    if (MODE == 0) y[row] = a;
    else if (MODE == 1) y[row] = vsub(b[row], a);
    else if (MODE == 2) y[row] = vadd(y[row], a);
    else y[row] = vadd(b[row], a);
  }
}

template <typename T>
static int csr_spmv_t(cfdh_ctx *c, const CsrDev &A, const T *x, T *y, int mode, const T *b) {
  const long long nthreads = 8ll * A.n;
  dim3 grid((unsigned)((nthreads + TPB - 1) / TPB)), block(TPB);
  if (mode == 0) hipLaunchKernelGGL((csr_spmv_kernel<0, T>), grid, block, 0, c->stream, A.n, A.rowptr.p, A.col.p, A.val.p, x, y, b);
  else if (mode == 1) hipLaunchKernelGGL((csr_spmv_kernel<1, T>), grid, block, 0, c->stream, A.n, A.rowptr.p, A.col.p, A.val.p, x, y, b);
  else if (mode == 2) hipLaunchKernelGGL((csr_spmv_kernel<2, T>), grid, block, 0, c->stream, A.n, A.rowptr.p, A.col.p, A.val.p, x, y, b);
  else hipLaunchKernelGGL((csr_spmv_kernel<3, T>), grid, block, 0, c->stream, A.n, A.rowptr.p, A.col.p, A.val.p, x, y, b);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_csr_spmv(cfdh_ctx *c, const CsrDev &A, const double *x, double *y, int mode, const double *b) {
  return csr_spmv_t<double>(c, A, x, y, mode, b);
}
// the same scalar matrix applied to ncol interleaved right-hand sides (velocity components through the scalar proxy)
int k_csr_spmv_ncol(cfdh_ctx *c, const CsrDev &A, const double *x, double *y, int mode, const double *b, int ncol) {
  if (ncol == 2) return csr_spmv_t<double2>(c, A, (const double2 *)x, (double2 *)y, mode, (const double2 *)b);
  if (ncol == 3) return csr_spmv_t<d3>(c, A, (const d3 *)x, (d3 *)y, mode, (const d3 *)b);
  return csr_spmv_t<double>(c, A, x, y, mode, b);
}

// One Chebyshev step on a scalar CSR level (single right-hand side):
//   r_out = r_in - A d_old ; d_new = c1 d_old + c2 D^-1 r_out ; x (+)= ...
// MODE 0: x += d_new.
// MODE 1: first step of a zero-guess smoothing fused with its initialisation: d_old = D^-1 r_in / theta
//         is formed on the fly while gathering (never stored), x = d_old + d_new.
// MODE 2: first step after csr_resid_init_kernel (which left d_old = D^-1 r / theta unapplied): x += d_old + d_new.
// LAST: r_out / d_new are not needed any more and are not written.
template <int MODE, bool LAST>
__global__ __launch_bounds__(TPB) void cheb_csr_step_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                            const double *__restrict__ val, const double *__restrict__ dinv,
                                                            const double *__restrict__ rin, double *__restrict__ rout,
                                                            const double *__restrict__ dold, double *__restrict__ dnew,
                                                            double *__restrict__ x, double c1, double c2, double itheta) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a = 0;
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int j = col[k];
      a += val[k] * (MODE == 1 ? dinv[j] * rin[j] * itheta : dold[j]);
    }
  }
  a = group8_sum(a);
  if (row < n && l == 0) {
    const double di = dinv[row], ri = rin[row];
    const double dd = (MODE == 1) ? di * ri * itheta : dold[row];
    const double r = ri - a;
    const double dn = c1 * dd + c2 * di * r;
    if (!LAST) { rout[row] = r; dnew[row] = dn; }
    if (MODE == 0) x[row] += dn;
    else if (MODE == 1) x[row] = dd + dn;
    else x[row] += dd + dn;
  }
}

// r = b - A x ; d0 = D^-1 r / theta   (residual of a non-zero guess fused with the Chebyshev initialisation)
__global__ __launch_bounds__(TPB) void csr_resid_init_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                             const double *__restrict__ val, const double *__restrict__ dinv,
                                                             const double *__restrict__ b, const double *__restrict__ x,
                                                             double *__restrict__ r, double *__restrict__ d0, double itheta) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  double a = 0;
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) a += val[k] * x[col[k]];
  }
  a = group8_sum(a);
  if (row < n && l == 0) {
    const double rr = b[row] - a;
    r[row] = rr;
    d0[row] = dinv[row] * rr * itheta;
  }
}

// Chebyshev smoothing with the level operator (ncol = 1): zero_guess ? x = S b : x <- x + S (b - A x)
static int level_cheb(cfdh_ctx *c, AmgLevel *L, const double *b, double *x, bool zero_guess, int deg, bool prof) {
  const int n = L->n;
  const double theta = 0.5 * (L->lmax + L->lmin), delta = 0.5 * (L->lmax - L->lmin), sigma = theta / delta;
  double rho = 1.0 / sigma;
  double *dold = L->d0.p, *dnew = L->d1.p, *r = L->r.p;
  const double *rin = b;
  const double itheta = 1.0 / theta;
  const long long nthreads = 8ll * n;
  dim3 grid((unsigned)((nthreads + TPB - 1) / TPB)), block(TPB);
  const int *rp = L->A.rowptr.p, *cl = L->A.col.p;
  const double *vl = L->A.val.p, *di = L->dinv.p;
  if (deg == 1) {  // plain damped Jacobi
    if (!zero_guess) { CHK(k_csr_spmv(c, L->A, x, r, 1, b)); rin = r; }
    hipLaunchKernelGGL(cheb_init_kernel, dim3((n + TPB - 1) / TPB), block, 0, c->stream, n, di, rin, dold, x, itheta,
                       zero_guess ? 0 : 1, (const double *)nullptr);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  if (!zero_guess) {
    hipLaunchKernelGGL(csr_resid_init_kernel, grid, block, 0, c->stream, n, rp, cl, vl, di, b, x, r, dold, itheta);
    rin = r;
  }
  for (int k = 1; k < deg; k++) {
    const double rho_new = 1.0 / (2.0 * sigma - rho);
    const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
    const bool last = (k == deg - 1);
    const int mode = (k == 1) ? (zero_guess ? 1 : 2) : 0;
    if (prof) prof_begin(c, 4);
#define LAUNCH_STEP(M, LST) hipLaunchKernelGGL((cheb_csr_step_kernel<M, LST>), grid, block, 0, c->stream, n, rp, cl, vl, di, rin, r, dold, dnew, x, c1, c2, itheta)
    if (mode == 1) { if (last) LAUNCH_STEP(1, true); else LAUNCH_STEP(1, false); }
    else if (mode == 2) { if (last) LAUNCH_STEP(2, true); else LAUNCH_STEP(2, false); }
    else { if (last) LAUNCH_STEP(0, true); else LAUNCH_STEP(0, false); }
#undef LAUNCH_STEP
    if (prof) prof_end(c, 4);
    rin = r;
    std::swap(dold, dnew);
    rho = rho_new;
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_level_smooth(cfdh_ctx *c, AmgLevel *L, const double *b, double *x, int degree) { return level_cheb(c, L, b, x, true, degree, false); }

// Two Chebyshev steps from a zero guess on a SELL-64 level in ONE pass over the matrix, with the
// Cahouet-Chabard scaling fused:  d = w D^-1 b ; r = b - A d ; x = d + (c1 d + c2 D^-1 r) ; y = ml .* x.
// (svalw carries the column weights w D^-1, as in the Jacobi pre-sweep.)
__global__ __launch_bounds__(TPB) void sell_cheb2_scale_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                               const float *__restrict__ svalw, const double *__restrict__ wdinv,
                                                               const double *__restrict__ dinv, const double *__restrict__ b,
                                                               double *__restrict__ x, const double *__restrict__ ml,
                                                               double *__restrict__ y, double c1, double c2) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n) return;
  const int sl = row >> 6, lane = row & 63;
  const int p0 = sptr[sl], w = (sptr[sl + 1] - p0) >> 6;
  (void)p0; (void)w;
  const double a = sell_row_dot<10, double>(sptr, scol, svalw, b, sl, lane, 0.0);
  const double bi = b[row];
  const double dd = wdinv[row] * bi;
  const double xv = dd + (c1 * dd + c2 * dinv[row] * (bi - a));
  x[row] = xv;
  y[row] = ml[row] * xv;
}
// x = Cheb2(H) b and y = ml .* x; false when the level does not qualify (the caller takes the generic path)
bool k_cc_cheb2_scale(cfdh_ctx *c, AmgLevel *L, const double *b, double *x, const double *ml, double *y) {
  const int n = L->n;
  if (!(L->A.nnz <= 20ll * n && n >= 16384)) return false;  // rows of up to ~15 entries (tetrahedra): 10 preloaded + tail
  const double theta = 0.5 * (L->lmax + L->lmin), delta = 0.5 * (L->lmax - L->lmin), sigma = theta / delta;
  const double rho = 1.0 / sigma, rho_new = 1.0 / (2.0 * sigma - rho);
  const double c1 = rho_new * rho, c2 = 2.0 * rho_new / delta;
  hipLaunchKernelGGL(sell_cheb2_scale_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, L->A.sptr.p, L->A.scol.p,
                     L->A.svalw.p, L->wdinv.p, L->dinv.p, b, x, ml, y, c1, c2);
  return hipGetLastError() == hipSuccess;
}

// y = Minv b for the dense coarsest inverse: one wave per row
template <typename T>
__global__ __launch_bounds__(64) void dense_mv_kernel(int n, const double *__restrict__ Minv, const T *__restrict__ b,
                                                      T *__restrict__ y) {
  const int row = blockIdx.x, l = threadIdx.x;
  T a = vzero((const T *)nullptr);
  for (int k = l; k < n; k += 64) a = vfma(Minv[(size_t)row * n + k], b[k], a);
  a = wsum(a);
  if (l == 0) y[row] = a;
}

// damped-Jacobi V-cycle building blocks: each touches the level matrix once.  A row whose only entry is
// its diagonal (Dirichlet row, isolated unknown) is solved exactly (weight 1 instead of 1/theta).
//   pre : xa = w D^-1 b (formed while gathering) ; r = b - A xa
//   post: x_out = x_in + w D^-1 (b - A x_in)
template <typename T>
__global__ __launch_bounds__(TPB) void jacobi_pre_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                         const double *__restrict__ val, const double *__restrict__ wdinv,
                                                         const T *__restrict__ b, T *__restrict__ xa, T *__restrict__ r) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) {
      const int j = col[k];
      a = vfma(val[k] * wdinv[j], b[j], a);
    }
  }
  a = g8(a);
  if (row < n && l == 0) {
    const T bi = b[row];
    xa[row] = vscale(wdinv[row], bi);
    r[row] = vsub(bi, a);
  }
}
template <typename T>
__global__ __launch_bounds__(TPB) void jacobi_post_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                          const double *__restrict__ val, const double *__restrict__ wdinv,
                                                          const T *__restrict__ b, const T *__restrict__ xin,
                                                          T *__restrict__ xout) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    for (int k = ks + l; k < ke; k += 8) a = vfma(val[k], xin[col[k]], a);
  }
  a = g8(a);
  if (row < n && l == 0) {
    xout[row] = vadd(xin[row], vscale(wdinv[row], vsub(b[row], a)));
  }
}

// ---- SELL-64 variants: one lane per row, fully coalesced matrix stream, no cross-lane reduction
template <int MODE, typename T>
__global__ __launch_bounds__(TPB) void sell_spmv_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                        const float *__restrict__ sval, const T *__restrict__ x,
                                                        T *__restrict__ y, const T *__restrict__ b) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n) return;
  const int sl = row >> 6, lane = row & 63;
  const int p0 = sptr[sl], w = (sptr[sl + 1] - p0) >> 6;
  T a = vzero((const T *)nullptr);
#pragma unroll 4
  for (int k = 0; k < w; k++) {
    const int p = p0 + k * 64 + lane;
    a = vfma((double)sval[p], x[scol[p]], a);
  }
  if (MODE == 0) y[row] = a;
  else if (MODE == 1) y[row] = vsub(b[row], a);
  else if (MODE == 2) y[row] = vadd(y[row], a);
  else y[row] = vadd(b[row], a);
}
template <typename T>
__global__ __launch_bounds__(TPB) void sell_jacobi_pre_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                              const float *__restrict__ svalw, const double *__restrict__ wdinv,
                                                              const T *__restrict__ b, T *__restrict__ xa, T *__restrict__ r) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n) return;
  const int sl = row >> 6, lane = row & 63;
  const int p0 = sptr[sl], w = (sptr[sl + 1] - p0) >> 6;
  T a = vzero((const T *)nullptr);
#pragma unroll 4
  for (int k = 0; k < w; k++) {
    const int p = p0 + k * 64 + lane;
    a = vfma((double)svalw[p], b[scol[p]], a);   // A (w D^-1 b): the column weight is folded into svalw
  }
  const T bi = b[row];
  xa[row] = vscale(wdinv[row], bi);
  r[row] = vsub(bi, a);
}
template <typename T>
__global__ __launch_bounds__(TPB) void sell_jacobi_post_kernel(int n, const int *__restrict__ sptr, const int *__restrict__ scol,
                                                               const float *__restrict__ sval, const double *__restrict__ wdinv,
                                                               const T *__restrict__ b, const T *__restrict__ xin,
                                                               T *__restrict__ xout) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n) return;
  const int sl = row >> 6, lane = row & 63;
  const int p0 = sptr[sl], w = (sptr[sl + 1] - p0) >> 6;
  T a = vzero((const T *)nullptr);
#pragma unroll 4
  for (int k = 0; k < w; k++) {
    const int p = p0 + k * 64 + lane;
    a = vfma((double)sval[p], xin[scol[p]], a);
  }
  xout[row] = vadd(xin[row], vscale(wdinv[row], vsub(b[row], a)));
}

template <typename T>
static int amg_cycle_jacobi(cfdh_ctx *c, AmgHier &H, size_t lev, const T *b, T *x, int prof) {
  AmgLevel *L = H.lev[lev];
  if (lev + 1 == H.lev.size()) {
    if (H.coarse_n > 0) {
      hipLaunchKernelGGL((dense_mv_kernel<T>), dim3(H.coarse_n), dim3(64), 0, c->stream, H.coarse_n, H.coarse_inv.p, b, x);
    } else {
      // near-diagonal coarsest level (coarsening stalled, cfdh_amg_setup): two damped-Jacobi sweeps
      const int n = L->n;
      dim3 block(TPB), gridC((unsigned)((8ll * n + TPB - 1) / TPB));
      T *xa = (T *)L->d0.p, *r = (T *)L->r.p;
      hipLaunchKernelGGL((jacobi_pre_kernel<T>), gridC, block, 0, c->stream, n, L->A.rowptr.p, L->A.col.p, L->A.val.p,
                         L->wdinv.p, b, xa, r);
      hipLaunchKernelGGL((jacobi_post_kernel<T>), gridC, block, 0, c->stream, n, L->A.rowptr.p, L->A.col.p, L->A.val.p,
                         L->wdinv.p, b, (const T *)xa, x);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  AmgLevel *N = H.lev[lev + 1];
  const int n = L->n;
  // short, regular rows (the finest levels of FE operators): SELL-64, one lane per row;
  // long coarse-level rows: 8 lanes per row over CSR
  const bool sell = L->A.nnz <= 12ll * n && n >= 16384;
  dim3 block(TPB), gridS((unsigned)((n + TPB - 1) / TPB)), gridC((unsigned)((8ll * n + TPB - 1) / TPB));
  T *xa = (T *)L->d0.p, *x1 = (T *)L->d1.p, *r = (T *)L->r.p;
  if (prof && lev == 0) prof_begin(c, prof);
  if (sell)
    hipLaunchKernelGGL((sell_jacobi_pre_kernel<T>), gridS, block, 0, c->stream, n, L->A.sptr.p, L->A.scol.p, L->A.svalw.p,
                       L->wdinv.p, b, xa, r);
  else
    hipLaunchKernelGGL((jacobi_pre_kernel<T>), gridC, block, 0, c->stream, n, L->A.rowptr.p, L->A.col.p, L->A.val.p,
                       L->wdinv.p, b, xa, r);
  if (prof && lev == 0) prof_end(c, prof);
  CHK(csr_spmv_t<T>(c, L->R, r, (T *)N->b.p, 0, (const T *)nullptr));   // b_c = R r
  CHK(amg_cycle_jacobi<T>(c, H, lev + 1, (const T *)N->b.p, (T *)N->x.p, prof));
  if (sell)
    hipLaunchKernelGGL((sell_spmv_kernel<3, T>), gridS, block, 0, c->stream, n, L->P.sptr.p, L->P.scol.p, L->P.sval.p,
                       (const T *)N->x.p, x1, (const T *)xa);           // x1 = xa + P x_c
  else
    CHK(csr_spmv_t<T>(c, L->P, (const T *)N->x.p, x1, 3, xa));
  if (prof && lev == 0) prof_begin(c, prof);
  if (sell)
    hipLaunchKernelGGL((sell_jacobi_post_kernel<T>), gridS, block, 0, c->stream, n, L->A.sptr.p, L->A.scol.p, L->A.sval.p,
                       L->wdinv.p, b, x1, x);
  else
    hipLaunchKernelGGL((jacobi_post_kernel<T>), gridC, block, 0, c->stream, n, L->A.rowptr.p, L->A.col.p, L->A.val.p,
                       L->wdinv.p, b, x1, x);
  if (prof && lev == 0) prof_end(c, prof);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- fused V(1,1) Jacobi cycle: one kernel per level and direction on precomputed composite operators
// (AmgLevel::G / Sb / Sc / D, built in cfdh_amg_setup).  Same linear map as amg_cycle_jacobi.
template <int LPR>
__device__ __forceinline__ double lpr_sum(double v) {
  v = group8_sum(v);
  if (LPR >= 16) v += dpp_shuffle<0x140>(v);  // row_mirror: sums of 16
  if (LPR >= 32) v += __shfl_xor(v, 16);
  if (LPR >= 64) v += __shfl_xor(v, 32);
  return v;
}
template <int LPR> __device__ __forceinline__ double lsum(double v) { return lpr_sum<LPR>(v); }
template <int LPR> __device__ __forceinline__ double2 lsum(double2 v) { return make_double2(lpr_sum<LPR>(v.x), lpr_sum<LPR>(v.y)); }
template <int LPR> __device__ __forceinline__ d3 lsum(d3 v) { return d3{lpr_sum<LPR>(v.x), lpr_sum<LPR>(v.y), lpr_sum<LPR>(v.z)}; }

__device__ __forceinline__ double epi_apply(double acc, int row, double alpha, double beta, const double *zH, const double *r, const unsigned char *pbc) {
  return (pbc[row] & 1) ? r[row] : alpha * acc + beta * zH[row];
}
__device__ __forceinline__ double2 epi_apply(double2 acc, int, double, double, const double *, const double *, const unsigned char *) { return acc; }
__device__ __forceinline__ double epi_value(double acc, unsigned flag, double alpha, double beta, double zh, double r) { return (flag & 1u) ? r : alpha * acc + beta * zh; }
__device__ __forceinline__ double2 epi_value(double2 acc, unsigned, double, double, double, double) { return acc; }
__device__ __forceinline__ d3 epi_apply(d3 acc, int, double, double, const double *, const double *, const unsigned char *) { return acc; }
__device__ __forceinline__ d3 epi_value(d3 acc, unsigned, double, double, double, double) { return acc; }

// y = G x, LPR lanes per row (rows of the coarse level: tens to hundreds of entries)
template <int LPR, typename VT, typename T>
__global__ __launch_bounds__(TPB) void fused_down_kernel(int n, const int *__restrict__ rowptr, const int *__restrict__ col,
                                                         const VT *__restrict__ val, const T *__restrict__ x, T *__restrict__ y) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid / LPR, l = gid % LPR;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    const int ks = rowptr[row], ke = rowptr[row + 1];
    // up to four entries per lane with all loads in flight together, then the gathers (see sell_row_dot)
    int cidx[4];
    VT cval[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int k = ks + l + q * LPR; if (k < ke) { cidx[q] = col[k]; cval[q] = val[k]; } }
    T g[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { const int k = ks + l + q * LPR; if (k < ke) g[q] = x[cidx[q]]; }
#pragma unroll
    for (int q = 0; q < 4; q++) { const int k = ks + l + q * LPR; if (k < ke) a = vfma((double)cval[q], g[q], a); }
    for (int k = ks + l + 4 * LPR; k < ke; k += LPR) a = vfma((double)val[k], x[col[k]], a);
  }
  a = lsum<LPR>(a);
  if (row < n && l == 0) y[row] = a;
}
// x = Sb b + Sc xc  (Sc may be absent: smoothing-only coarsest level), 8 lanes per row over CSR
template <typename T>
__global__ __launch_bounds__(TPB) void fused_up_csr_kernel(int n, const int *__restrict__ rpB, const int *__restrict__ clB,
                                                           const double *__restrict__ vlB, const T *__restrict__ b,
                                                           const int *__restrict__ rpC, const int *__restrict__ clC,
                                                           const double *__restrict__ vlC, const T *__restrict__ xc,
                                                           T *__restrict__ x, double ea, double eb, const double *__restrict__ ezH,
                                                           const double *__restrict__ er, const unsigned char *__restrict__ epbc) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 3, l = gid & 7;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    // two entries of each matrix per lane requested together, then their gathers (rows of 10-30 entries on the coarse levels:
    // the plain loops make two to four dependent round trips per matrix)
    const int kb = rpB[row] + l, keB = rpB[row + 1];
    int kc = 0, keC = 0;
    if (rpC) { kc = rpC[row] + l; keC = rpC[row + 1]; }
    int cb[2], cc[2];
    double vb[2], vc[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (kb + 8 * q < keB) { cb[q] = clB[kb + 8 * q]; vb[q] = vlB[kb + 8 * q]; }
      if (kc + 8 * q < keC) { cc[q] = clC[kc + 8 * q]; vc[q] = vlC[kc + 8 * q]; }
    }
    T gb[2], gc[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (kb + 8 * q < keB) gb[q] = b[cb[q]];
      if (kc + 8 * q < keC) gc[q] = xc[cc[q]];
    }
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (kb + 8 * q < keB) a = vfma(vb[q], gb[q], a);
      if (kc + 8 * q < keC) a = vfma(vc[q], gc[q], a);
    }
    for (int k = kb + 16; k < keB; k += 8) a = vfma(vlB[k], b[clB[k]], a);
    for (int k = kc + 16; k < keC; k += 8) a = vfma(vlC[k], xc[clC[k]], a);
  }
  a = g8(a);
  if (row < n && l == 0) x[row] = epbc ? epi_apply(a, row, ea, eb, ezH, er, epbc) : a;
}
// the same on SELL-64 (fp32 values), one lane per row.  WIDE: rows longer than one chunk of preloaded entries (tetrahedra)
template <typename T, bool WIDE>
__global__ __launch_bounds__(TPB) void fused_up_sell_kernel(int n, const int *__restrict__ spB, const int *__restrict__ scB,
                                                            const float *__restrict__ svB, const T *__restrict__ b,
                                                            const int *__restrict__ spC, const int *__restrict__ scC,
                                                            const float *__restrict__ svC, const T *__restrict__ xc,
                                                            T *__restrict__ x, double ea, double eb, const double *__restrict__ ezH,
                                                            const double *__restrict__ er, const unsigned char *__restrict__ epbc) {
  const int row = blockIdx.x * TPB + threadIdx.x;
  if (row >= n) return;
  const int sl = __builtin_amdgcn_readfirstlane(row >> 6), lane = row & 63;
  T a = vzero((const T *)nullptr);
  // operands of the epilogue: requested now, so that they arrive together with the matrix data
  double e_zh = 0.0, e_r = 0.0;
  unsigned e_f = 0;
  if (epbc) { e_f = epbc[row]; e_zh = ezH[row]; e_r = er[row]; }
  // both matrices together: slice pointers, then ALL column/value loads of a chunk of the row, then ALL its gathers (three
  // dependent round trips; the straightforward loops make about ten).  Triangle meshes need one chunk (rows of <= 10
  // entries), the 15- to 25-entry rows of tetrahedral meshes two or three.
  constexpr int MW = sizeof(T) > 8 ? 8 : 10;
  const int pB = spB[sl], wB = (spB[sl + 1] - pB) >> 6;
  int pC = 0, wC = 0;
  if (spC) { pC = spC[sl]; wC = (spC[sl + 1] - pC) >> 6; }
  const int wmax = wB > wC ? wB : wC;
  for (int k0 = 0; k0 < (WIDE ? wmax : 1); k0 += MW) {
    int cB[MW], cC[MW];
    float vB[MW], vC[MW];
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wB) { const int p = pB + (k0 + k) * 64 + lane; cB[k] = NTLOAD(scB + p); vB[k] = NTLOAD(svB + p); }
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wC) { const int p = pC + (k0 + k) * 64 + lane; cC[k] = NTLOAD(scC + p); vC[k] = NTLOAD(svC + p); }
    T gB[MW], gC[MW];
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wB) gB[k] = b[cB[k]];
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wC) gC[k] = xc[cC[k]];
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wB) a = vfma((double)vB[k], gB[k], a);
#pragma unroll
    for (int k = 0; k < MW; k++) if (k0 + k < wC) a = vfma((double)vC[k], gC[k], a);
  }
  if (!WIDE) {  // rows beyond the chunk (none on triangle meshes)
    for (int k = MW; k < wB; k++) { const int p = pB + k * 64 + lane; a = vfma((double)svB[p], b[scB[p]], a); }
    for (int k = MW; k < wC; k++) { const int p = pC + k * 64 + lane; a = vfma((double)svC[p], xc[scC[p]], a); }
  }
  x[row] = epbc ? epi_value(a, e_f, ea, eb, e_zh, e_r) : a;
}
// x = Sb b + D bc with the dense folded coarse correction D [n][nc] (fp32), one wave per row
template <typename T>
__global__ __launch_bounds__(TPB) void fused_up_dense_kernel(int n, const int *__restrict__ rpB, const int *__restrict__ clB,
                                                             const double *__restrict__ vlB, const T *__restrict__ b,
                                                             const float *__restrict__ D, int nc, const T *__restrict__ bc,
                                                             T *__restrict__ x) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int row = gid >> 6, l = gid & 63;
  T a = vzero((const T *)nullptr);
  if (row < n) {
    for (int k = rpB[row] + l, ke = rpB[row + 1]; k < ke; k += 64) a = vfma(vlB[k], b[clB[k]], a);
    const float *Dr = D + (size_t)row * nc;
    // the whole row in flight at once when it fits 12 steps (coarsest levels of <= 768 unknowns), remainder in a loop
    float dv[12];
    T bv[12];
#pragma unroll
    for (int q = 0; q < 12; q++) { const int j = l + 64 * q; if (j < nc) { dv[q] = Dr[j]; bv[q] = bc[j]; } }
#pragma unroll
    for (int q = 0; q < 12; q++) { const int j = l + 64 * q; if (j < nc) a = vfma((double)dv[q], bv[q], a); }
    for (int j = l + 768; j < nc; j += 64) a = vfma((double)Dr[j], bc[j], a);
  }
  a = wsum(a);
  if (row < n && l == 0) x[row] = a;
}

template <typename VT, typename T>
static void launch_down(cfdh_ctx *c, const CsrDev &G, const VT *val, const T *x, T *y) {
  const long long avg = G.n > 0 ? (G.nnz + G.n - 1) / G.n : 1;
  const int lpr = avg > 96 ? 64 : (avg > 48 ? 32 : (avg > 20 ? 16 : 8));  // 2-4 entries per lane, loaded together
  dim3 block(TPB), grid((unsigned)(((long long)G.n * lpr + TPB - 1) / TPB));
  if (lpr == 64) hipLaunchKernelGGL((fused_down_kernel<64, VT, T>), grid, block, 0, c->stream, G.n, G.rowptr.p, G.col.p, val, x, y);
  else if (lpr == 32) hipLaunchKernelGGL((fused_down_kernel<32, VT, T>), grid, block, 0, c->stream, G.n, G.rowptr.p, G.col.p, val, x, y);
  else if (lpr == 16) hipLaunchKernelGGL((fused_down_kernel<16, VT, T>), grid, block, 0, c->stream, G.n, G.rowptr.p, G.col.p, val, x, y);
  else hipLaunchKernelGGL((fused_down_kernel<8, VT, T>), grid, block, 0, c->stream, G.n, G.rowptr.p, G.col.p, val, x, y);
}

template <typename T>
// l0 > 0: the cycle of the levels l0 .. (the replicated levels of a partitioned run below its distributed finest pressure level);
// b / x are then level l0's vectors
static int amg_cycle_fused(cfdh_ctx *c, AmgHier &H, const T *b, T *x, int prof, int l0 = 0) {
  const int nl = (int)H.lev.size();
  // down: right-hand sides of all coarse levels.  (Merging the coarse levels' down-sweeps into one launch through the
  // products G_2 G_1, ... was measured and dropped: the products fill in -- 1.1 M entries for a 713-row level -- and the one
  // launch costs more than the two it replaces.)
  for (int l = l0; l + 1 < nl; l++) {
    AmgLevel *L = H.lev[l], *N = H.lev[l + 1];
    const T *src = l == l0 ? b : (const T *)L->b.p;
    if (prof && l == 0) prof_begin(c, prof + 4);
    if (L->fine) launch_down<float, T>(c, L->G, L->G.valf.p, src, (T *)N->b.p);
    else launch_down<double, T>(c, L->G, L->G.val.p, src, (T *)N->b.p);
    if (prof && l == 0) prof_end(c, prof + 4);
  }
  // coarsest level (or the level above it when the dense solve is folded into its up-sweep)
  int l = nl - 1;
  {
    AmgLevel *L = H.lev[l];
    const T *bl = l == l0 ? b : (const T *)L->b.p;
    T *xl = l == l0 ? x : (T *)L->x.p;
    if (nl - 2 >= l0 && H.lev[nl - 2]->Dn > 0) {
      AmgLevel *U = H.lev[nl - 2];
      const T *bu = nl - 2 == l0 ? b : (const T *)U->b.p;
      T *xu = nl - 2 == l0 ? x : (T *)U->x.p;
      // (a two-level hierarchy: this IS the finest up-sweep -- owned rows only where the caller keeps just those)
      const int un = (nl - 2 == 0 && c->up0_rows > 0 && c->up0_rows < U->n) ? c->up0_rows : U->n;
      hipLaunchKernelGGL((fused_up_dense_kernel<T>), dim3((unsigned)((64ll * un + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, un,
                         U->Sb.rowptr.p, U->Sb.col.p, U->Sb.val.p, bu, U->D.p, U->Dn, bl, xu);
      l = nl - 3;
    } else {
      if (H.coarse_n > 0)
        hipLaunchKernelGGL((dense_mv_kernel<T>), dim3(H.coarse_n), dim3(64), 0, c->stream, H.coarse_n, H.coarse_inv.p, bl, xl);
      else if (L->sell)
        hipLaunchKernelGGL((fused_up_sell_kernel<T, true>), dim3((unsigned)((L->n + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, L->n,
                           L->Sb.sptr.p, L->Sb.scol.p, L->Sb.sval.p, bl, (const int *)nullptr, (const int *)nullptr,
                           (const float *)nullptr, (const T *)nullptr, xl, 0.0, 0.0, (const double *)nullptr, (const double *)nullptr,
                           (const unsigned char *)nullptr);
      else
        hipLaunchKernelGGL((fused_up_csr_kernel<T>), dim3((unsigned)((8ll * L->n + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, L->n,
                           L->Sb.rowptr.p, L->Sb.col.p, L->Sb.val.p, bl, (const int *)nullptr, (const int *)nullptr,
                           (const double *)nullptr, (const T *)nullptr, xl, 0.0, 0.0, (const double *)nullptr, (const double *)nullptr,
                           (const unsigned char *)nullptr);
      l = nl - 2;
    }
  }
  // up
  for (; l >= l0; l--) {
    AmgLevel *L = H.lev[l], *N = H.lev[l + 1];
    const T *bl = l == l0 ? b : (const T *)L->b.p;
    T *xl = l == l0 ? x : (T *)L->x.p;
    // Cahouet-Chabard combination in the epilogue of the last kernel of the (single right-hand side) pressure cycle
    const bool epi = l == 0 && c->epi.on && sizeof(T) == sizeof(double);
    const double ea = epi ? c->epi.alpha : 0.0, eb = epi ? c->epi.beta : 0.0;
    const double *ezH = epi ? c->epi.zH : nullptr, *er = epi ? c->epi.r : nullptr;
    const unsigned char *epbc = epi ? c->epi.pbc : nullptr;
    if (epi) { xl = (T *)c->epi.out; c->epi.done = true; }
    // overlapping velocity cycle of a partitioned run: only the owned rows (the first ones) of the finest level's result are kept
    const int nrow = (l == 0 && c->up0_rows > 0 && c->up0_rows < L->n) ? c->up0_rows : L->n;
    if (prof && l == 0) prof_begin(c, prof);
    // a few longer rows (irregular vertices of a triangle mesh) go through the tail loop of the one-chunk kernel; the chunked
    // kernel is for meshes whose typical row exceeds a chunk (tetrahedra)
    if (L->sell && L->Sb.sell_maxw <= 14 && L->Sc.sell_maxw <= 14)
      hipLaunchKernelGGL((fused_up_sell_kernel<T, false>), dim3((unsigned)((nrow + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, nrow,
                         L->Sb.sptr.p, L->Sb.scol.p, L->Sb.sval.p, bl, L->Sc.sptr.p, L->Sc.scol.p, L->Sc.sval.p, (const T *)N->x.p, xl,
                         ea, eb, ezH, er, epbc);
    else if (L->sell)
      hipLaunchKernelGGL((fused_up_sell_kernel<T, true>), dim3((unsigned)((nrow + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, nrow,
                         L->Sb.sptr.p, L->Sb.scol.p, L->Sb.sval.p, bl, L->Sc.sptr.p, L->Sc.scol.p, L->Sc.sval.p, (const T *)N->x.p, xl,
                         ea, eb, ezH, er, epbc);
    else
      hipLaunchKernelGGL((fused_up_csr_kernel<T>), dim3((unsigned)((8ll * nrow + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, nrow,
                         L->Sb.rowptr.p, L->Sb.col.p, L->Sb.val.p, bl, L->Sc.rowptr.p, L->Sc.col.p, L->Sc.val.p, (const T *)N->x.p, xl,
                         ea, eb, ezH, er, epbc);
    if (prof && l == 0) prof_end(c, prof);
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}

// V-cycle with Chebyshev smoothing of degree >= 2 (single right-hand side)
static int amg_cycle_cheb(cfdh_ctx *c, AmgHier &H, size_t lev, const double *b, double *x, bool prof) {
  AmgLevel *L = H.lev[lev];
  if (lev + 1 == H.lev.size()) {
    if (H.coarse_n > 0) {
      hipLaunchKernelGGL((dense_mv_kernel<double>), dim3(H.coarse_n), dim3(64), 0, c->stream, H.coarse_n, H.coarse_inv.p, b, x);
      HIPCHK(c, hipGetLastError());
      return 0;
    }
    return level_cheb(c, L, b, x, true, c->opt.amg_smooth_degree > 2 ? c->opt.amg_smooth_degree : 2, false);  // near-diagonal coarsest level: smoothing only
  }
  AmgLevel *N = H.lev[lev + 1];
  const int deg = c->opt.amg_smooth_degree;
  CHK(level_cheb(c, L, b, x, true, deg, prof && lev == 0));
  CHK(k_csr_spmv(c, L->A, x, L->r.p, 1, b));     // r = b - A x
  CHK(k_csr_spmv(c, L->R, L->r.p, N->b.p, 0, nullptr));  // b_c = R r
  CHK(amg_cycle_cheb(c, H, lev + 1, N->b.p, N->x.p, prof));
  CHK(k_csr_spmv(c, L->P, N->x.p, x, 2, nullptr));  // x += P x_c
  CHK(level_cheb(c, L, b, x, false, deg, prof && lev == 0));
  return 0;
}

// ---- distributed finest level of the replicated pressure hierarchy (cfdh_ctx::DistL0)
// b_loc = pressure slot of a halo-layout vector on owned + ghost vertices; xa = w D^-1 b on all of them
__global__ __launch_bounds__(TPB) void dl0_pack_kernel(int nvo, int nv, int dim, const double *__restrict__ vec, const double *__restrict__ wdinv,
                                                       double *__restrict__ b, double *__restrict__ xa, int ghosts) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= nv) return;
  // pressure slot: owned at dim nvo + i, ghost record (u..., p) of dim + 1 doubles behind the owned part
  // ghosts == 0: the ghost layer of the right-hand side was not exchanged -- the pre-smoothed iterate is taken as zero there
  const double v = i < nvo ? vec[(size_t)dim * nvo + i] : (ghosts ? vec[((size_t)dim + 1) * nvo + ((size_t)dim + 1) * (size_t)(i - nvo) + dim] : 0.0);
  b[i] = v;
  xa[i] = wdinv[i] * v;
}
// pre-smoothing on the owned rows and the owned part of the coarse right-hand side: lev[1].b = P_owned^T (b - A xa)
int k_dl0_down(cfdh_ctx *c, const double *halo_vec) {
  cfdh_ctx::DistL0 &d = c->dl0;
  AmgLevel *N = c->hLg.lev[1];
  const int nvo = c->nvo, nv = c->nv;
  // without the exchange of the right-hand side's ghost layer the producer (the H solve) has written the owned part of d.b itself;
  // the ghost parts of d.b and d.xa stay zero
  if (d.ghost_rhs) hipLaunchKernelGGL(dl0_pack_kernel, dim3((nv + TPB - 1) / TPB), dim3(TPB), 0, c->stream, nvo, nv, c->dim, halo_vec, d.wdinv.p, d.b.p, d.xa.p, 1);
  dim3 block(TPB), grid((unsigned)((8ll * nvo + TPB - 1) / TPB)), gridS((unsigned)((nvo + TPB - 1) / TPB));
  if (d.A.nnz <= 12ll * nvo && nvo >= 16384)  // short regular rows: SELL-64 (as the replicated level 0 would use)
    hipLaunchKernelGGL((sell_jacobi_pre_kernel<double>), gridS, block, 0, c->stream, nvo, d.A.sptr.p, d.A.scol.p, d.A.svalw.p,
                       d.wdinv.p, (const double *)d.b.p, d.xa.p, d.r.p);
  else
    hipLaunchKernelGGL((jacobi_pre_kernel<double>), grid, block, 0, c->stream, nvo, d.A.rowptr.p, d.A.col.p, d.A.val.p, d.wdinv.p,
                       (const double *)d.b.p, d.xa.p, d.r.p);
  HIPCHK(c, hipGetLastError());
  return csr_spmv_t<double>(c, d.PT, d.r.p, N->b.p, 0, (const double *)nullptr);
}
// replicated coarse cycle from level 1, prolongation to owned + ghost rows, post-smoothing of the owned rows -> out
int k_dl0_up(cfdh_ctx *c, double *out) {
  cfdh_ctx::DistL0 &d = c->dl0;
  AmgLevel *N = c->hLg.lev[1];
  const int nvo = c->nvo;
  // the replicated levels: composite-operator cycle from level 1 (6 launches for four coarse levels instead of 13 sweeps)
  static const bool sweeps = getenv("CFDH_DL0_COARSE_SWEEPS") && getenv("CFDH_DL0_COARSE_SWEEPS")[0] == '1';
  if (c->hLg.fused && c->opt.amg_smooth_degree == 1 && !sweeps) CHK(amg_cycle_fused<double>(c, c->hLg, (const double *)N->b.p, N->x.p, 0, 1));
  else CHK(amg_cycle_jacobi<double>(c, c->hLg, 1, (const double *)N->b.p, N->x.p, 0));
  if (d.P.nnz <= 12ll * c->nv && c->nv >= 16384)
    hipLaunchKernelGGL((sell_spmv_kernel<3, double>), dim3((unsigned)((c->nv + TPB - 1) / TPB)), dim3(TPB), 0, c->stream, c->nv,
                       d.P.sptr.p, d.P.scol.p, d.P.sval.p, (const double *)N->x.p, d.x1.p, (const double *)d.xa.p);
  else
    CHK(csr_spmv_t<double>(c, d.P, (const double *)N->x.p, d.x1.p, 3, (const double *)d.xa.p));  // x1 = xa + P x_c
  dim3 block(TPB), grid((unsigned)((8ll * nvo + TPB - 1) / TPB)), gridS((unsigned)((nvo + TPB - 1) / TPB));
  if (d.A.nnz <= 12ll * nvo && nvo >= 16384)
    hipLaunchKernelGGL((sell_jacobi_post_kernel<double>), gridS, block, 0, c->stream, nvo, d.A.sptr.p, d.A.scol.p, d.A.sval.p,
                       d.wdinv.p, (const double *)d.b.p, (const double *)d.x1.p, out);
  else
    hipLaunchKernelGGL((jacobi_post_kernel<double>), grid, block, 0, c->stream, nvo, d.A.rowptr.p, d.A.col.p, d.A.val.p, d.wdinv.p,
                       (const double *)d.b.p, (const double *)d.x1.p, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// x = V(H) b; for ncol == 2 b and x hold interleaved pairs
int k_amg_vcycle(cfdh_ctx *c, AmgHier &H, const double *b, double *x) {
  if (!H.valid || H.lev.empty()) return cfdh_fail(c, CFDH_E_STATE, "AMG hierarchy not built");
  const bool prof = (&H == &c->hS) || (&H == &c->hL) || (&H == &c->hLg);
  if (H.ncol == 3) {
    if (H.fused && c->opt.amg_smooth_degree == 1 && H.lev.size() >= 2) return amg_cycle_fused<d3>(c, H, (const d3 *)b, (d3 *)x, 5);
    return amg_cycle_jacobi<d3>(c, H, 0, (const d3 *)b, (d3 *)x, 5);
  }
  if (H.fused && c->opt.amg_smooth_degree == 1 && H.lev.size() >= 2) {
    if (H.ncol == 2) return amg_cycle_fused<double2>(c, H, (const double2 *)b, (double2 *)x, 5);
    return amg_cycle_fused<double>(c, H, b, x, prof ? 4 : 0);
  }
  if (H.ncol == 2) return amg_cycle_jacobi<double2>(c, H, 0, (const double2 *)b, (double2 *)x, 5);
  if (c->opt.amg_smooth_degree == 1) return amg_cycle_jacobi<double>(c, H, 0, b, x, prof ? 4 : 0);
  return amg_cycle_cheb(c, H, 0, b, x, prof);
}

// Cahouet-Chabard combination: y = M_l z (0 on Dirichlet rows) ; out = alpha t + beta z, out = r on Dirichlet rows
__global__ __launch_bounds__(TPB) void cc_scale_kernel(int n, const double *__restrict__ ml, const double *__restrict__ z, double *__restrict__ y) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) y[i] = ml[i] * z[i];
}
__global__ __launch_bounds__(TPB) void cc_combine_kernel(int n, double alpha, double beta, const double *__restrict__ t,
                                                         const double *__restrict__ z, const double *__restrict__ r,
                                                         const unsigned char *__restrict__ pbc, double *__restrict__ out, double *__restrict__ out2) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) {
    const double v = (pbc[i] & 1) ? r[i] : alpha * t[i] + beta * z[i];
    out[i] = v;
    if (out2) out2[i] = v;  // partitioned run: z_p also into the pressure slot of the halo scratch vector
  }
}
__global__ __launch_bounds__(TPB) void scatter_global_kernel(int n, const int *__restrict__ l2g, const double *__restrict__ loc, double *__restrict__ glob) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) glob[l2g[i]] = loc[i];
}
__global__ __launch_bounds__(TPB) void gather_global_kernel(int n, const int *__restrict__ l2g, const double *__restrict__ glob, double *__restrict__ loc) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) loc[i] = glob[l2g[i]];
}
// velocity part of a halo-layout vector ([u owned | p owned | (ux,uy,p) per ghost]) as nv contiguous pairs
__global__ __launch_bounds__(TPB) void ext_pack_kernel(int nvo, int nv, const double *__restrict__ vec, double2 *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= nv) return;
  if (i < nvo) out[i] = make_double2(vec[2 * (size_t)i], vec[2 * (size_t)i + 1]);
  else {
    const double *t = vec + 3 * (size_t)nvo + 3 * (size_t)(i - nvo);
    out[i] = make_double2(t[0], t[1]);
  }
}
__global__ __launch_bounds__(TPB) void ext_pack3_kernel(int nvo, int nv, const double *__restrict__ vec, double *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= nv) return;
  const double *t = i < nvo ? vec + 3 * (size_t)i : vec + 4 * (size_t)i;  // ghost record (ux, uy, uz, p) at 4 nvo + 4 (i - nvo)
  out[3 * (size_t)i] = t[0]; out[3 * (size_t)i + 1] = t[1]; out[3 * (size_t)i + 2] = t[2];
}
int k_ext_pack(cfdh_ctx *c, const double *vec, double *out) {
  if (c->dim == 3) {
    hipLaunchKernelGGL(ext_pack3_kernel, dim3((c->nv + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nvo, c->nv, vec, out);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(ext_pack_kernel, dim3((c->nv + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nvo, c->nv, vec, (double2 *)out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_scatter_global(cfdh_ctx *c, int n, const int *l2g, const double *loc, double *glob) {
  hipLaunchKernelGGL(scatter_global_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, l2g, loc, glob);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_gather_global(cfdh_ctx *c, int n, const int *l2g, const double *glob, double *loc) {
  hipLaunchKernelGGL(gather_global_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, l2g, glob, loc);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_cc_scale(cfdh_ctx *c, int n, const double *ml, const double *z, double *y) {
  hipLaunchKernelGGL(cc_scale_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, ml, z, y);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int k_cc_combine(cfdh_ctx *c, int n, double alpha, double beta, const double *t, const double *z, const double *r,
                 const unsigned char *pbc, double *out, double *out2) {
  hipLaunchKernelGGL(cc_combine_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, alpha, beta, t, z, r, pbc, out, out2);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- vector kernels
__global__ __launch_bounds__(TPB) void axpy_kernel(int n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) y[i] += a * x[i];
}
__global__ __launch_bounds__(TPB) void waxpy_kernel(int n, double a, const double *__restrict__ x, const double *__restrict__ y,
                                                    double *__restrict__ w) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) w[i] = y[i] + a * x[i];
}
__global__ __launch_bounds__(TPB) void scale_kernel(int n, double a, double *__restrict__ x) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) x[i] *= a;
}
static inline int vgrid(int n) { int g = (n + TPB * 4 - 1) / (TPB * 4); return g < 1 ? 1 : (g > 2048 ? 2048 : g); }

__global__ __launch_bounds__(TPB) void pmult_kernel(int n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ o) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) o[i] = a[i] * b[i];
}
int v_pointwise_mult(cfdh_ctx *c, int n, const double *a, const double *b, double *out) {
  hipLaunchKernelGGL(pmult_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, a, b, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int v_copy(cfdh_ctx *c, int n, const double *x, double *y) {
  HIPCHK(c, hipMemcpyAsync(y, x, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}
int v_zero(cfdh_ctx *c, int n, double *y) {
  HIPCHK(c, hipMemsetAsync(y, 0, sizeof(double) * (size_t)n, c->stream));
  return 0;
}
int v_axpy(cfdh_ctx *c, int n, double a, const double *x, double *y) {
  hipLaunchKernelGGL(axpy_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, a, x, y);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int v_waxpy(cfdh_ctx *c, int n, double a, const double *x, const double *y, double *w) {
  hipLaunchKernelGGL(waxpy_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, a, x, y, w);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int v_scale(cfdh_ctx *c, int n, double a, double *x) {
  hipLaunchKernelGGL(scale_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, a, x);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- reductions: per-block partials (fixed order) -> one final block; deterministic
// OP 0: sum x*y, 1: max |x - y| (y may be null)
template <int OP>
__global__ __launch_bounds__(TPB) void reduce_partial_kernel(int n, const double *__restrict__ x, const double *__restrict__ y,
                                                             double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    if (OP == 0) a += x[i] * y[i];
    else a = fmax(a, fabs(y ? x[i] - y[i] : x[i]));
  }
  a = (OP == 0) ? block_sum(a, sh) : block_max(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// out[v] = reduce(partial[v*stride .. +nblk)), one block per v; OP 2: sqrt of the sum
template <int OP>
__global__ __launch_bounds__(TPB) void reduce_final_kernel(int nblk, int stride, const double *__restrict__ partial,
                                                           double *__restrict__ out, double *__restrict__ mirror = nullptr) {
  __shared__ double sh[4];
  const double *pp = partial + (size_t)blockIdx.x * stride;
  double a = 0;
  for (int i = threadIdx.x; i < nblk; i += TPB) a = (OP == 1) ? fmax(a, pp[i]) : a + pp[i];
  a = (OP == 1) ? block_max(a, sh) : block_sum(a, sh);
  if (threadIdx.x == 0) {
    const double v = (OP == 2) ? sqrt(a) : a;
    out[blockIdx.x] = v;
    if (mirror) mirror[blockIdx.x] = v;  // host-mapped copy: the host reads it after an event, no copy kernel
  }
}

// Single rank: the final reduction kernel also stores its results into host-mapped memory (h_pinned + 300),
// so reading them back costs a stream synchronisation instead of a copy kernel (~11 us each).
#define CFDH_MIRROR_OFF 300
static double *scalar_mirror(cfdh_ctx *c, const double *out_dev, int cnt) {
  if (c->nranks > 1 || cnt > 64) { c->mirror_src = nullptr; return nullptr; }
  c->mirror_src = out_dev; c->mirror_cnt = cnt;
  return c->h_pinned_dev + CFDH_MIRROR_OFF;
}
__global__ __launch_bounds__(TPB) void mirror_copy_kernel(int n, const double *__restrict__ src, double *__restrict__ dst) {
  for (int i = threadIdx.x; i < n; i += TPB) dst[i] = src[i];
}
// all-reduce of n freshly reduced scalars; in a partitioned run the REDUCED values are then published to the
// host-mapped scratch by a one-block kernel behind the collective (one rank: the reduction kernel did it already)
static int finish_scalars(cfdh_ctx *c, double *out_dev, int n, int op) {
  CHK(comm_allreduce_dev(c, out_dev, n, op));
  if (c->nranks > 1 && n <= 64) {
    hipLaunchKernelGGL(mirror_copy_kernel, dim3(1), dim3(TPB), 0, c->stream, n, (const double *)out_dev, c->h_pinned_dev + CFDH_MIRROR_OFF);
    HIPCHK(c, hipGetLastError());
    c->mirror_src = out_dev; c->mirror_cnt = n;
  }
  return 0;
}
static int read_scalars(cfdh_ctx *c, const double *dev, int n, double *host) {
  if (c->mirror_src == dev && n <= c->mirror_cnt) {
    c->mirror_src = nullptr;  // one shot
    c->n_host_sync++;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; i++) host[i] = c->h_pinned[CFDH_MIRROR_OFF + i];
    return 0;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_pinned, dev, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  c->n_host_sync++;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < n; i++) host[i] = c->h_pinned[i];
  return 0;
}

static int reduce_dev(cfdh_ctx *c, int op, int n, const double *x, const double *y, double *out_dev) {
  const int nb = vgrid(n) > c->red_blocks ? c->red_blocks : vgrid(n);
  if (op == 0) {
    hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(nb), dim3(TPB), 0, c->stream, n, x, y, c->red_partial.p);
    hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, out_dev,
                       scalar_mirror(c, out_dev, 1));
  } else {
    hipLaunchKernelGGL(reduce_partial_kernel<1>, dim3(nb), dim3(TPB), 0, c->stream, n, x, y, c->red_partial.p);
    hipLaunchKernelGGL(reduce_final_kernel<1>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, out_dev,
                       scalar_mirror(c, out_dev, 1));
  }
  HIPCHK(c, hipGetLastError());
  return finish_scalars(c, out_dev, 1, op);
}

int v_dot(cfdh_ctx *c, int n, const double *x, const double *y, double *out_host) {
  CHK(reduce_dev(c, 0, n, x, y, c->red_out.p));
  return read_scalars(c, c->red_out.p, 1, out_host);
}
int v_norm2(cfdh_ctx *c, int n, const double *x, double *out_host) {
  CHK(v_dot(c, n, x, x, out_host));
  *out_host = sqrt(*out_host);
  return 0;
}
// |x| and |y| with one read-back (one host synchronisation instead of two)
int v_norm2_pair(cfdh_ctx *c, int n, const double *x, const double *y, double *nx, double *ny) {
  const int nb = vgrid(n) > c->red_blocks ? c->red_blocks : vgrid(n);
  if ((size_t)2 * nb > c->red_partial.n) return cfdh_fail(c, CFDH_E_STATE, "reduction workspace too small");
  hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(nb), dim3(TPB), 0, c->stream, n, x, x, c->red_partial.p);
  hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(nb), dim3(TPB), 0, c->stream, n, y, y, c->red_partial.p + nb);
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p,
                     scalar_mirror(c, c->red_out.p, 2));
  HIPCHK(c, hipGetLastError());
  CHK(finish_scalars(c, c->red_out.p, 2, 0));
  double v[2];
  CHK(read_scalars(c, c->red_out.p, 2, v));
  *nx = sqrt(v[0]); *ny = sqrt(v[1]);
  return 0;
}
int v_norminf_diff(cfdh_ctx *c, int n, const double *x, const double *y, double *out_host) {
  CHK(reduce_dev(c, 1, n, x, y, c->red_out.p));
  return read_scalars(c, c->red_out.p, 1, out_host);
}

__global__ __launch_bounds__(TPB) void sub_scalar_kernel(int n, double *__restrict__ p, const double *__restrict__ s, double scale) {
  const double m = s[0] * scale;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) p[i] -= m;
}
__global__ __launch_bounds__(TPB) void sum_partial_kernel(int n, const double *__restrict__ x, double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) a += x[i];
  a = block_sum(a, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = a;
}
// p -= mean(p) over all ranks (constant-pressure null vector, stabilized_schur.py:282-293,319)
int v_sub_mean(cfdh_ctx *c, int n, double *p) {
  const int nb = vgrid(n) > c->red_blocks ? c->red_blocks : vgrid(n);
  double *acc = c->red_out.p + 8;  // [sum, count]
  hipLaunchKernelGGL(sum_partial_kernel, dim3(nb), dim3(TPB), 0, c->stream, n, p, c->red_partial.p);
  c->mirror_src = nullptr;
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, acc);
  HIPCHK(c, hipGetLastError());
  double scale = 1.0 / n;
  if (c->nranks > 1) {
    CHK(comm_allreduce_dev(c, acc, 1, 0));
    // global number of pressure dofs: constant, reduced once when the communicator is attached (global_counts)
    if (n != c->nvo || !(c->nvo_global > 0)) return cfdh_fail(c, CFDH_E_STATE, "v_sub_mean: global count unknown");
    scale = 1.0 / c->nvo_global;
  }
  hipLaunchKernelGGL(sub_scalar_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, p, acc, scale);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---- Gram-Schmidt building blocks: h_i = V_i . w for i < nvec (V column-major, leading dim ld)
#define MD_G 8     // vectors reduced together: one pass over the w chunk feeds 8 dot products
#define MD_NB 1024 // blocks: 4 per CU, each owning a contiguous chunk (w stays L1/L2 resident across groups)
__global__ __launch_bounds__(TPB) void multidot_kernel(int n, const double *__restrict__ V, size_t ld, int nvec,
                                                       const double *__restrict__ w, double *__restrict__ partial, int nblk,
                                                       int with_ww) {
  __shared__ double sh[4][MD_G];
  const int nout = nvec + with_ww;
  const int per = (((n + nblk - 1) / nblk) + 1) & ~1;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int g0 = 0; g0 < nout; g0 += MD_G) {
    const double *ptr[MD_G];
    double acc[MD_G];
#pragma unroll
    for (int q = 0; q < MD_G; q++) {
      const int v = g0 + q;
      ptr[q] = (v < nvec) ? V + (size_t)v * ld : w;  // v == nvec: w.w ; v > nvec: dummy (discarded)
      acc[q] = 0.0;
    }
    // two consecutive entries per lane (16-B loads; lo and the leading dimension are even), odd tail by the last lane
    const int hi2 = hi > lo ? lo + ((hi - lo) & ~1) : hi;  // blocks past the end of a short vector own nothing
    for (int i = lo + 2 * threadIdx.x; i < hi2; i += 2 * TPB) {
      const double2 wi = *(const double2 *)(w + i);
#pragma unroll
      for (int q = 0; q < MD_G; q++) {
        const double2 vi = *(const double2 *)(ptr[q] + i);
        acc[q] += vi.x * wi.x + vi.y * wi.y;
      }
    }
    if (hi2 < hi && threadIdx.x == 0) {
      const double wi = w[hi2];
#pragma unroll
      for (int q = 0; q < MD_G; q++) acc[q] += ptr[q][hi2] * wi;
    }
#pragma unroll
    for (int q = 0; q < MD_G; q++) {
      const double r = wave_sum(acc[q]);
      if (lane == 0) sh[wv][q] = r;
    }
    __syncthreads();
    if (threadIdx.x < MD_G && g0 + (int)threadIdx.x < nout)
      partial[(size_t)(g0 + threadIdx.x) * nblk + blockIdx.x] =
          (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    __syncthreads();
  }
}
// Gram system of the projected initial guess in ONE pass over the K + 1 vectors: out slot (i, q) = W_q . W_i for i < K and
// W_q . b for i = K, laid out as out[8 i + q] (the unused slots of the 8-wide rows are written as zeros).
template <int K>
__global__ __launch_bounds__(TPB) void gram_kernel(int n, const double *__restrict__ W, size_t ld, const double *__restrict__ b,
                                                   double *__restrict__ partial, int nblk) {
  constexpr int NP = K * (K + 1) / 2 + K;
  __shared__ double sh[4][NP];
  const int per = (((n + nblk - 1) / nblk) + 1) & ~1;
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double acc[NP];
#pragma unroll
  for (int t = 0; t < NP; t++) acc[t] = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += TPB) {
    double x[K];
#pragma unroll
    for (int q = 0; q < K; q++) x[q] = W[(size_t)q * ld + i];
    const double bi = b[i];
    int t = 0;
#pragma unroll
    for (int q = 0; q < K; q++) {
#pragma unroll
      for (int r = q; r < K; r++) acc[t++] += x[q] * x[r];
    }
#pragma unroll
    for (int q = 0; q < K; q++) acc[t++] += x[q] * bi;
  }
#pragma unroll
  for (int t = 0; t < NP; t++) {
    const double r = wave_sum(acc[t]);
    if (lane == 0) sh[wv][t] = r;
  }
  __syncthreads();
  // scatter into the 8-wide slot layout (symmetric entries twice)
  if (threadIdx.x < 8 * (K + 1)) {
    const int i = threadIdx.x >> 3, q = threadIdx.x & 7;
    double v = 0.0;
    if (q < K) {
      int t;
      if (i < K) { const int lo_ = min(i, q), hi_ = max(i, q); t = lo_ * K - lo_ * (lo_ - 1) / 2 + (hi_ - lo_); }
      else t = K * (K + 1) / 2 + q;
      v = (sh[0][t] + sh[1][t]) + (sh[2][t] + sh[3][t]);
    }
    partial[(size_t)threadIdx.x * nblk + blockIdx.x] = v;
  }
}
// out_dev[8 i + q] as above, NOT reduced over the ranks (the caller does that once)
int v_gram(cfdh_ctx *c, int n, const double *W, int ld, int k, const double *b, double *out_dev) {
  const int nb = MD_NB, nout = 8 * (k + 1);
  if (k < 2 || k > 4 || (size_t)nout * nb > c->red_partial.n) {
    for (int i = 0; i <= k; i++) CHK(v_multidot(c, n, W, ld, k, i < k ? W + (size_t)i * ld : b, out_dev + (size_t)i * 8, false, nullptr, false));
    return 0;
  }
  if (k == 2) hipLaunchKernelGGL((gram_kernel<2>), dim3(nb), dim3(TPB), 0, c->stream, n, W, (size_t)ld, b, c->red_partial.p, nb);
  else if (k == 3) hipLaunchKernelGGL((gram_kernel<3>), dim3(nb), dim3(TPB), 0, c->stream, n, W, (size_t)ld, b, c->red_partial.p, nb);
  else hipLaunchKernelGGL((gram_kernel<4>), dim3(nb), dim3(TPB), 0, c->stream, n, W, (size_t)ld, b, c->red_partial.p, nb);
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(nout), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, out_dev, (double *)nullptr);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// (A "last block does the final reduction" variant was measured and dropped: the device-scope release fence every
// block needs before taking its ticket writes the XCD's L2 back -- 133 us per launch against 17 + 4 us for two kernels.)
// h_dev[0..nvec) = V^T w (and h_dev[nvec] = w.w when with_ww), reduced over all ranks
int v_multidot(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *w, double *h_dev, bool with_ww, double *mirror, bool reduce_ranks) {
  const int nb = MD_NB, nout = nvec + (with_ww ? 1 : 0);
  if ((size_t)nout * nb > c->red_partial.n) return cfdh_fail(c, CFDH_E_STATE, "multidot workspace too small");
  hipLaunchKernelGGL(multidot_kernel, dim3(nb), dim3(TPB), 0, c->stream, n, V, (size_t)ld, nvec, w, c->red_partial.p, nb,
                     with_ww ? 1 : 0);
  // single rank: the h values also land in host-mapped memory (`mirror`: device view of a slot of the FGMRES read-back ring)
  // straight from the kernel
  double *mir = (mirror && c->nranks <= 1) ? mirror : nullptr;
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(nout), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, h_dev, mir);
  HIPCHK(c, hipGetLastError());
  if (!reduce_ranks) return 0;  // the caller reduces several results over the ranks at once
  CHK(comm_allreduce_dev(c, h_dev, nout, 0));
  if (mirror && c->nranks > 1) {  // partitioned: publish the REDUCED coefficients the same way (behind the all-reduce)
    hipLaunchKernelGGL(mirror_copy_kernel, dim3(1), dim3(TPB), 0, c->stream, nout, (const double *)h_dev, mirror);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}
__global__ __launch_bounds__(TPB) void scale_to_kernel(int n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) y[i] = a * x[i];
}
int v_scale_to(cfdh_ctx *c, int n, double a, const double *x, double *y) {
  hipLaunchKernelGGL(scale_to_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, a, x, y);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// w -= sum_i h_i V_i
__global__ __launch_bounds__(TPB) void multiaxpy_kernel(int n, const double *__restrict__ V, size_t ld, int nvec,
                                                        const double *__restrict__ h, double *__restrict__ w, double sign) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) {
    double a0 = w[i], a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int v = 0;
    for (; v + 4 <= nvec; v += 4) {  // four independent streams in flight
      const double x0 = V[(size_t)v * ld + i], x1 = V[(size_t)(v + 1) * ld + i], x2 = V[(size_t)(v + 2) * ld + i],
                   x3 = V[(size_t)(v + 3) * ld + i];
      a0 += sign * h[v] * x0; a1 += sign * h[v + 1] * x1; a2 += sign * h[v + 2] * x2; a3 += sign * h[v + 3] * x3;
    }
    for (; v < nvec; v++) a0 += sign * h[v] * V[(size_t)v * ld + i];
    w[i] = (a0 + a1) + (a2 + a3);
  }
}
int v_multiaxpy(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *h_dev, double *w) {
  hipLaunchKernelGGL(multiaxpy_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, V, (size_t)ld, nvec, h_dev, w, -1.0);
  HIPCHK(c, hipGetLastError());
  return 0;
}
__global__ __launch_bounds__(TPB) void gs_update_normalize_kernel(int n, const double *__restrict__ V, size_t ld, int nvec,
                                                                 const double *__restrict__ h, const double *__restrict__ w,
                                                                 double *__restrict__ vn, double *__restrict__ s_out) {
  const double ww = h[nvec];
  double hh2 = 0.0;
  for (int v = 0; v < nvec; v++) hh2 += h[v] * h[v];
  const double nrm2 = ww - hh2;
  const double s = (nrm2 > 0.0 && nrm2 <= ww) ? sqrt(nrm2) : sqrt(ww);  // cancellation: any positive scale, the caller re-orthogonalises
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *s_out = s;
  // two consecutive entries per lane (16-B loads: ld is even and all vectors are 16-B aligned); same summation order per entry
  const int n2 = n & ~1;
  for (int i = 2 * (blockIdx.x * TPB + threadIdx.x); i < n2; i += 2 * gridDim.x * TPB) {
    const double2 wi = *(const double2 *)(w + i);
    double a0 = wi.x, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = wi.y, b1 = 0.0, b2 = 0.0, b3 = 0.0;
    int v = 0;
    for (; v + 4 <= nvec; v += 4) {
      const double2 x0 = *(const double2 *)(V + (size_t)v * ld + i), x1 = *(const double2 *)(V + (size_t)(v + 1) * ld + i),
                    x2 = *(const double2 *)(V + (size_t)(v + 2) * ld + i), x3 = *(const double2 *)(V + (size_t)(v + 3) * ld + i);
      a0 -= h[v] * x0.x; a1 -= h[v + 1] * x1.x; a2 -= h[v + 2] * x2.x; a3 -= h[v + 3] * x3.x;
      b0 -= h[v] * x0.y; b1 -= h[v + 1] * x1.y; b2 -= h[v + 2] * x2.y; b3 -= h[v + 3] * x3.y;
    }
    for (; v < nvec; v++) { const double2 xv = *(const double2 *)(V + (size_t)v * ld + i); a0 -= h[v] * xv.x; b0 -= h[v] * xv.y; }
    *(double2 *)(vn + i) = make_double2(((a0 + a1) + (a2 + a3)) * inv, ((b0 + b1) + (b2 + b3)) * inv);
  }
  if (n2 < n && blockIdx.x == 0 && threadIdx.x == 0) {
    double a0 = w[n2];
    for (int v = 0; v < nvec; v++) a0 -= h[v] * V[(size_t)v * ld + n2];
    vn[n2] = a0 * inv;
  }
}
int v_gs_update_normalize(cfdh_ctx *c, int n, const double *V, int ld, int nvec, const double *h_dev, const double *w, double *vn, double *s_dev) {
  hipLaunchKernelGGL(gs_update_normalize_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, V, (size_t)ld, nvec, h_dev, w, vn, s_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}
__global__ void sqrt_kernel(double *s);
// ---- Gram-Schmidt against an fp32 COPY of the basis (long Krylov cycles: the two passes over V are 40 % of an iteration at
// depth 25; the fp64 vectors stay where the preconditioner reads them).  The norm of the new vector is measured, not inferred
// from w.w - |h|^2 (that identity needs an orthonormal basis to round-off, which rounded columns are not).
__global__ __launch_bounds__(TPB) void multidot32_kernel(int n, const float *__restrict__ V, size_t ld, int nvec,
                                                         const double *__restrict__ w, double *__restrict__ partial, int nblk) {
  __shared__ double sh[4][MD_G + 1];
  const int per = (((n + nblk - 1) / nblk) + 3) & ~3;  // chunks of whole float4 / 2 x double2 groups
  const int lo = blockIdx.x * per, hi = min(n, lo + per);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int hi4 = hi > lo ? lo + ((hi - lo) & ~3) : hi;
  for (int g0 = 0; g0 < nvec; g0 += MD_G) {
    const float *ptr[MD_G];
    double acc[MD_G], aww = 0.0;
#pragma unroll
    for (int q = 0; q < MD_G; q++) { ptr[q] = V + (size_t)min(g0 + q, nvec - 1) * ld; acc[q] = 0.0; }  // past the end: a dummy, discarded
    // four consecutive entries per lane: one 16-B load per fp32 column, two for w
    for (int i = lo + 4 * threadIdx.x; i < hi4; i += 4 * TPB) {
      const double2 w0 = *(const double2 *)(w + i), w1 = *(const double2 *)(w + i + 2);
      if (g0 == 0) aww += (w0.x * w0.x + w0.y * w0.y) + (w1.x * w1.x + w1.y * w1.y);
#pragma unroll
      for (int q = 0; q < MD_G; q++) {
        const float4 vi = *(const float4 *)(ptr[q] + i);
        acc[q] += ((double)vi.x * w0.x + (double)vi.y * w0.y) + ((double)vi.z * w1.x + (double)vi.w * w1.y);
      }
    }
    if (threadIdx.x == 0)
      for (int i = hi4; i < hi; i++) {
        const double wi = w[i];
        if (g0 == 0) aww += wi * wi;
#pragma unroll
        for (int q = 0; q < MD_G; q++) acc[q] += (double)ptr[q][i] * wi;
      }
#pragma unroll
    for (int q = 0; q < MD_G; q++) {
      const double r = wave_sum(acc[q]);
      if (lane == 0) sh[wv][q] = r;
    }
    if (g0 == 0) { const double r = wave_sum(aww); if (lane == 0) sh[wv][MD_G] = r; }
    __syncthreads();
    if (threadIdx.x < MD_G && g0 + (int)threadIdx.x < nvec)
      partial[(size_t)(g0 + threadIdx.x) * nblk + blockIdx.x] =
          (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    if (g0 == 0 && threadIdx.x == MD_G)
      partial[(size_t)nvec * nblk + blockIdx.x] = (sh[0][MD_G] + sh[1][MD_G]) + (sh[2][MD_G] + sh[3][MD_G]);
    __syncthreads();
  }
}
// h_dev[0..nvec) = V32^T w, h_dev[nvec] = w.w (reduced over the ranks, mirrored like v_multidot)
int v_multidot32(cfdh_ctx *c, int n, const float *V, int ld, int nvec, const double *w, double *h_dev, double *mirror) {
  const int nb = MD_NB, nout = nvec + 1;
  if ((size_t)nout * nb > c->red_partial.n) return cfdh_fail(c, CFDH_E_STATE, "multidot workspace too small");
  hipLaunchKernelGGL(multidot32_kernel, dim3(nb), dim3(TPB), 0, c->stream, n, V, (size_t)ld, nvec, w, c->red_partial.p, nb);
  double *mir = c->nranks <= 1 ? mirror : nullptr;
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(nout), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, h_dev, mir);
  HIPCHK(c, hipGetLastError());
  CHK(comm_allreduce_dev(c, h_dev, nout, 0));
  if (c->nranks > 1) {
    hipLaunchKernelGGL(mirror_copy_kernel, dim3(1), dim3(TPB), 0, c->stream, nout, (const double *)h_dev, mirror);
    HIPCHK(c, hipGetLastError());
  }
  return 0;
}
// vn = w - V32 h (not normalised) and the block partials of |vn|^2
__global__ __launch_bounds__(TPB) void gs_update32_kernel(int n, const float *__restrict__ V, size_t ld, int nvec, const double *__restrict__ h,
                                                          const double *__restrict__ w, double *__restrict__ vn, double *__restrict__ partial) {
  __shared__ double sh[4];
  double ss = 0.0;
  const int n2 = n & ~1;
  for (int i = 2 * (blockIdx.x * TPB + threadIdx.x); i < n2; i += 2 * gridDim.x * TPB) {
    const double2 wi = *(const double2 *)(w + i);
    double a0 = wi.x, a1 = 0.0, b0 = wi.y, b1 = 0.0;
    int v = 0;
    for (; v + 2 <= nvec; v += 2) {
      const float2 x0 = *(const float2 *)(V + (size_t)v * ld + i), x1 = *(const float2 *)(V + (size_t)(v + 1) * ld + i);
      a0 -= h[v] * (double)x0.x; a1 -= h[v + 1] * (double)x1.x;
      b0 -= h[v] * (double)x0.y; b1 -= h[v + 1] * (double)x1.y;
    }
    for (; v < nvec; v++) { const float2 xv = *(const float2 *)(V + (size_t)v * ld + i); a0 -= h[v] * (double)xv.x; b0 -= h[v] * (double)xv.y; }
    const double r0 = a0 + a1, r1 = b0 + b1;
    *(double2 *)(vn + i) = make_double2(r0, r1);
    ss += r0 * r0 + r1 * r1;
  }
  if (n2 < n && blockIdx.x == 0 && threadIdx.x == 0) {
    double a0 = w[n2];
    for (int v = 0; v < nvec; v++) a0 -= h[v] * (double)V[(size_t)v * ld + n2];
    vn[n2] = a0;
    ss += a0 * a0;
  }
  ss = block_sum(ss, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = ss;
}
// vn /= s (s on the device) and its fp32 copy
__global__ __launch_bounds__(TPB) void scale_store32_kernel(int n, double *__restrict__ vn, const double *__restrict__ s, float *__restrict__ v32) {
  const double inv = s[0] > 0.0 ? 1.0 / s[0] : 0.0;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) { const double x = vn[i] * inv; vn[i] = x; v32[i] = (float)x; }
}
__global__ __launch_bounds__(TPB) void store32_kernel(int n, const double *__restrict__ v, float *__restrict__ v32) {
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) v32[i] = (float)v[i];
}
int v_store32(cfdh_ctx *c, int n, const double *v, float *v32) {
  hipLaunchKernelGGL(store32_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, v, v32);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// vn = (w - V32 h) / |w - V32 h| with its fp32 copy in v32n; s_dev[0] = that norm (reduced over the ranks), mirrored to
// the host-mapped word `mirror` for the host
int v_gs_update32(cfdh_ctx *c, int n, const float *V, int ld, int nvec, const double *h_dev, const double *w, double *vn, float *v32n,
                  double *s_dev, double *mirror) {
  const int nb = vgrid(n) > c->red_blocks ? c->red_blocks : vgrid(n);
  double *part = c->red_partial.p + (size_t)(MD_NB) * 8;  // behind the first multi-dot groups (the stream serialises the users)
  hipLaunchKernelGGL(gs_update32_kernel, dim3(nb), dim3(TPB), 0, c->stream, n, V, (size_t)ld, nvec, h_dev, w, vn, part);
  if (c->nranks <= 1) {  // square root and host-mapped copy in the reduction kernel itself
    hipLaunchKernelGGL(reduce_final_kernel<2>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, part, s_dev, mirror);
  } else {
    hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, part, s_dev, (double *)nullptr);
    HIPCHK(c, hipGetLastError());
    CHK(comm_allreduce_dev(c, s_dev, 1, 0));
    hipLaunchKernelGGL(sqrt_kernel, dim3(1), dim3(1), 0, c->stream, s_dev);
    hipLaunchKernelGGL(mirror_copy_kernel, dim3(1), dim3(TPB), 0, c->stream, 1, (const double *)s_dev, mirror);
  }
  hipLaunchKernelGGL(scale_store32_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, vn, (const double *)s_dev, v32n);
  HIPCHK(c, hipGetLastError());
  return 0;
}
int v_lincomb(cfdh_ctx *c, int n, const double *Z, int ld, int nvec, const double *y_dev, double *x) {
  hipLaunchKernelGGL(multiaxpy_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, Z, (size_t)ld, nvec, y_dev, x, 1.0);
  HIPCHK(c, hipGetLastError());
  return 0;
}
__global__ void sqrt_kernel(double *s) { s[0] = sqrt(s[0]); }
int v_norm_to_dev(cfdh_ctx *c, int n, const double *w, double *out_dev) {
  CHK(reduce_dev(c, 0, n, w, w, out_dev));
  hipLaunchKernelGGL(sqrt_kernel, dim3(1), dim3(1), 0, c->stream, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}
// the same without the reduction over the ranks: norms of rank-local operators (hierarchy set-up of a partitioned run)
int v_norm_to_dev_local(cfdh_ctx *c, int n, const double *w, double *out_dev) {
  const int nb = vgrid(n) > c->red_blocks ? c->red_blocks : vgrid(n);
  hipLaunchKernelGGL(reduce_partial_kernel<0>, dim3(nb), dim3(TPB), 0, c->stream, n, w, w, c->red_partial.p);
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(1), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, out_dev, (double *)nullptr);
  hipLaunchKernelGGL(sqrt_kernel, dim3(1), dim3(1), 0, c->stream, out_dev);
  HIPCHK(c, hipGetLastError());
  return 0;
}
__global__ __launch_bounds__(TPB) void scale_inv_dev_kernel(int n, const double *__restrict__ w, const double *__restrict__ nrm,
                                                            double *__restrict__ v) {
  const double s = nrm[0] != 0.0 ? 1.0 / nrm[0] : 0.0;
  for (int i = blockIdx.x * TPB + threadIdx.x; i < n; i += gridDim.x * TPB) v[i] = w[i] * s;
}
int v_scale_inv_dev(cfdh_ctx *c, int n, const double *w, const double *nrm_dev, double *v) {
  hipLaunchKernelGGL(scale_inv_dev_kernel, dim3(vgrid(n)), dim3(TPB), 0, c->stream, n, w, nrm_dev, v);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ||J n|| and || |J| n || for the constant-pressure vector n (MatNullSpaceTest, stabilized_schur.py:314; the second norm makes
// the decision scale-free, see cfdh_newton_step)
__global__ __launch_bounds__(TPB) void nulltest_kernel(int nvo, const int *__restrict__ vptr, const double *__restrict__ A01,
                                                       const double *__restrict__ A11, double *__restrict__ partial) {
  __shared__ double sh[4];
  double a = 0, b = 0;
  for (int row = blockIdx.x * TPB + threadIdx.x; row < nvo; row += gridDim.x * TPB) {
    double s0 = 0, s1 = 0, s2 = 0, t0 = 0, t1 = 0, t2 = 0;
    for (int k = vptr[row]; k < vptr[row + 1]; k++) {
      const double c0 = A01[2 * (size_t)k], c1 = A01[2 * (size_t)k + 1], c2 = A11[k];
      s0 += c0; s1 += c1; s2 += c2;
      t0 += fabs(c0); t1 += fabs(c1); t2 += fabs(c2);
    }
    a += s0 * s0 + s1 * s1 + s2 * s2;
    b += t0 * t0 + t1 * t1 + t2 * t2;
  }
  a = block_sum(a, sh);
  b = block_sum(b, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = a; partial[gridDim.x + blockIdx.x] = b; }
}
int k_nullspace_test(cfdh_ctx *c, double *nrm, double *absnrm) {
  if (c->dim == 3) return k3_nullspace_test(c, nrm, absnrm);
  const int nb = vgrid(c->nvo) > c->red_blocks ? c->red_blocks : vgrid(c->nvo);
  hipLaunchKernelGGL(nulltest_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nvo, c->vptr.p, c->A01.p, c->A11.p, c->red_partial.p);
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p,
                     scalar_mirror(c, c->red_out.p, 2));
  HIPCHK(c, hipGetLastError());
  CHK(finish_scalars(c, c->red_out.p, 2, 0));
  double s[2];
  CHK(read_scalars(c, c->red_out.p, 2, s));
  *nrm = sqrt(s[0]);
  *absnrm = sqrt(s[1]);
  return 0;
}

// sparse update of the Dirichlet arrays (cfdh_solver.cpp::upload_bc): entry k describes vertex idx[k] completely
__global__ __launch_bounds__(TPB) void bc_scatter_kernel(int n, int ncomp, const int *__restrict__ idx, const unsigned char *__restrict__ flag,
                                                         const double *__restrict__ val, const double *__restrict__ mult,
                                                         unsigned char *__restrict__ bcflag, double *__restrict__ bcval,
                                                         double *__restrict__ bcmult) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= n) return;
  const int v = idx[k];
  bcflag[v] = flag[k];
  for (int i = 0; i < ncomp; i++) { bcval[(size_t)ncomp * v + i] = val[(size_t)ncomp * k + i]; bcmult[(size_t)ncomp * v + i] = mult[(size_t)ncomp * k + i]; }
}
int k_bc_scatter(cfdh_ctx *c, int n, int ncomp, const int *idx, const unsigned char *flag, const double *val, const double *mult) {
  hipLaunchKernelGGL(bc_scatter_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, ncomp, idx, flag, val, mult, c->bcflag.p,
                     c->bcval.p, c->bcmult.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- functionals
// kind 0/1: drag / lift over exterior facets with the given marker (dfg_1.py:183-202)
__global__ __launch_bounds__(TPB) void draglift_kernel(int nfac, int marker, int nvo, const int *__restrict__ fcell,
                                                       const int *__restrict__ flocal, const int *__restrict__ fmarker,
                                                       const int *__restrict__ cells, const unsigned char *__restrict__ cown,
                                                       const double *__restrict__ coords,
                                                       const double *__restrict__ x, double mu, double *__restrict__ partial) {
  __shared__ double sh[4];
  double aD = 0, aL = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker) continue;
    const int e = fcell[k], fl = flocal[k];
    if (!cown[e]) continue;
    int vs[3];
    double X[3][2];
    for (int a = 0; a < 3; a++) { vs[a] = cells[3 * e + a]; X[a][0] = coords[2 * vs[a]]; X[a][1] = coords[2 * vs[a] + 1]; }
    const double det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
    double g[3][2];
    g[0][0] = (X[1][1] - X[2][1]) / det; g[0][1] = (X[2][0] - X[1][0]) / det;
    g[1][0] = (X[2][1] - X[0][1]) / det; g[1][1] = (X[0][0] - X[2][0]) / det;
    g[2][0] = (X[0][1] - X[1][1]) / det; g[2][1] = (X[1][0] - X[0][0]) / det;
    const double area = 0.5 * fabs(det);
    const double gfx = fl == 0 ? g[0][0] : (fl == 1 ? g[1][0] : g[2][0]);
    const double gfy = fl == 0 ? g[0][1] : (fl == 1 ? g[1][1] : g[2][1]);
    const double gl = hypot(gfx, gfy);
    const double n0 = gfx / gl, n1 = gfy / gl;  // n = -FacetNormal
    const double elen = 2.0 * area * gl;
    const double t0 = n1, t1 = -n0;
    double gu0 = 0, gu1 = 0;
    for (int a = 0; a < 3; a++) {
      const int uo = uoff(vs[a], nvo);
      const double ut = x[uo] * t0 + x[uo + 1] * t1;
      gu0 += ut * g[a][0]; gu1 += ut * g[a][1];
    }
    const double dn = gu0 * n0 + gu1 * n1;
    const int a1 = vs[(fl + 1) % 3], a2 = vs[(fl + 2) % 3];
    const double pm = 0.5 * (x[poff(a1, nvo)] + x[poff(a2, nvo)]);
    aD += elen * (mu * dn * n1 - pm * n0);
    aL -= elen * (mu * dn * n0 + pm * n1);
  }
  aD = block_sum(aD, sh);
  aL = block_sum(aL, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = aD; partial[gridDim.x + blockIdx.x] = aL; }
}
// kind 7: volume flux  sum_facets |e| n . (u_a + u_b) / 2  with the outward normal n = -grad(lambda_fl) / |grad(lambda_fl)|
// and |e| |grad(lambda_fl)| = |det|
__global__ __launch_bounds__(TPB) void flux_kernel(int nfac, int marker, int nvo, const int *__restrict__ fcell,
                                                   const int *__restrict__ flocal, const int *__restrict__ fmarker,
                                                   const int *__restrict__ cells, const unsigned char *__restrict__ cown,
                                                   const double *__restrict__ coords, const double *__restrict__ x,
                                                   double *__restrict__ partial) {
  __shared__ double sh[4];
  double q = 0;
  for (int k = blockIdx.x * TPB + threadIdx.x; k < nfac; k += gridDim.x * TPB) {
    if (fmarker[k] != marker) continue;
    const int e = fcell[k], fl = flocal[k];
    if (!cown[e]) continue;
    int vs[3];
    double X[3][2];
    for (int a = 0; a < 3; a++) { vs[a] = cells[3 * e + a]; X[a][0] = coords[2 * vs[a]]; X[a][1] = coords[2 * vs[a] + 1]; }
    const double det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
    const int a1 = (fl + 1) % 3, a2 = (fl + 2) % 3;
    // det * grad(lambda_fl) = rot(X[a1] - X[a2])
    const double gx = X[a1][1] - X[a2][1], gy = X[a2][0] - X[a1][0];
    const int u1 = uoff(vs[a1], nvo), u2 = uoff(vs[a2], nvo);
    const double sgn = det > 0 ? -0.5 : 0.5;
    q += sgn * (gx * (x[u1] + x[u2]) + gy * (x[u1 + 1] + x[u2 + 1]));
  }
  q = block_sum(q, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = q; partial[gridDim.x + blockIdx.x] = 0.0; }
}
// kind 2/3: int u.u, int p^2; with overlapping parts a cell is integrated only by
// the rank that owns its first vertex (cell_owned), so global sums count it once
__global__ __launch_bounds__(TPB) void l2_kernel(int nc, int nvo, const int *__restrict__ cells, const unsigned char *__restrict__ cown,
                                                 const double *__restrict__ coords, const double *__restrict__ x,
                                                 double *__restrict__ partial) {
  __shared__ double sh[4];
  double au = 0, ap = 0;
  for (int e = blockIdx.x * TPB + threadIdx.x; e < nc; e += gridDim.x * TPB) {
    if (cown && !cown[e]) continue;
    int vs[3];
    double X[3][2];
    for (int a = 0; a < 3; a++) { vs[a] = cells[3 * e + a]; X[a][0] = coords[2 * vs[a]]; X[a][1] = coords[2 * vs[a] + 1]; }
    const double det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
    const double area = 0.5 * fabs(det);
    double ux[3], uy[3], pp[3];
    for (int a = 0; a < 3; a++) { const int uo = uoff(vs[a], nvo); ux[a] = x[uo]; uy[a] = x[uo + 1]; pp[a] = x[poff(vs[a], nvo)]; }
    double su = 0, sp = 0;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        const double m = (a == b ? 2.0 : 1.0);
        su += m * (ux[a] * ux[b] + uy[a] * uy[b]);
        sp += m * pp[a] * pp[b];
      }
    au += area * su * (1.0 / 12.0);
    ap += area * sp * (1.0 / 12.0);
  }
  au = block_sum(au, sh);
  ap = block_sum(ap, sh);
  if (threadIdx.x == 0) { partial[blockIdx.x] = au; partial[gridDim.x + blockIdx.x] = ap; }
}

// wall shear stress (solverBase.py:163-195): shear[v] += (1/|e|) oint lambda_v Tt ds = Tt/2 for both vertices of each
// exterior facet, Tt = T - (T.n) n, T = -sigma(u,p) n (the pressure part is purely normal).  A boundary vertex of a
// 2-D mesh receives two contributions, so the atomic sum is order-independent.
__global__ __launch_bounds__(TPB) void wss_kernel(int nfac, int nvo, const int *__restrict__ fcell, const int *__restrict__ flocal,
                                                  const int *__restrict__ cells, const double *__restrict__ coords,
                                                  const double *__restrict__ x, double mu, double *__restrict__ out) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k >= nfac) return;
  const int e = fcell[k], fl = flocal[k];
  int vs[3];
  double X[3][2], u[3][2];
  for (int a = 0; a < 3; a++) {
    vs[a] = cells[3 * e + a];
    X[a][0] = coords[2 * vs[a]]; X[a][1] = coords[2 * vs[a] + 1];
    const int uo = uoff(vs[a], nvo);
    u[a][0] = x[uo]; u[a][1] = x[uo + 1];
  }
  const double det = (X[1][0] - X[0][0]) * (X[2][1] - X[0][1]) - (X[1][1] - X[0][1]) * (X[2][0] - X[0][0]);
  double g[3][2];
  g[0][0] = (X[1][1] - X[2][1]) / det; g[0][1] = (X[2][0] - X[1][0]) / det;
  g[1][0] = (X[2][1] - X[0][1]) / det; g[1][1] = (X[0][0] - X[2][0]) / det;
  g[2][0] = (X[0][1] - X[1][1]) / det; g[2][1] = (X[1][0] - X[0][0]) / det;
  const double gfx = fl == 0 ? g[0][0] : (fl == 1 ? g[1][0] : g[2][0]);
  const double gfy = fl == 0 ? g[0][1] : (fl == 1 ? g[1][1] : g[2][1]);
  const double gl = hypot(gfx, gfy);
  const double n[2] = {-gfx / gl, -gfy / gl};
  double G[2][2] = {{0, 0}, {0, 0}};  // G_ij = d_i u_j
  for (int a = 0; a < 3; a++)
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 2; j++) G[i][j] += g[a][i] * u[a][j];
  const double E01 = 0.5 * (G[0][1] + G[1][0]);
  const double T[2] = {-2.0 * mu * (G[0][0] * n[0] + E01 * n[1]), -2.0 * mu * (E01 * n[0] + G[1][1] * n[1])};
  const double Tn = T[0] * n[0] + T[1] * n[1];
  const double Tt[2] = {0.5 * (T[0] - Tn * n[0]), 0.5 * (T[1] - Tn * n[1])};
  const int v1 = vs[(fl + 1) % 3], v2 = vs[(fl + 2) % 3];
  atomicAdd(out + 2 * (size_t)v1, Tt[0]); atomicAdd(out + 2 * (size_t)v1 + 1, Tt[1]);
  atomicAdd(out + 2 * (size_t)v2, Tt[0]); atomicAdd(out + 2 * (size_t)v2 + 1, Tt[1]);
}
int k_wss(cfdh_ctx *c, double *out) {
  if (c->gen) return c->dim == 3 ? kg3_wss(c, out) : kg_wss(c, out);
  if (c->dim == 3) return k3_wss(c, out);
  HIPCHK(c, hipMemsetAsync(out, 0, sizeof(double) * 2 * (size_t)c->nv, c->stream));
  if (c->nfac > 0)
    hipLaunchKernelGGL(wss_kernel, dim3((c->nfac + TPB - 1) / TPB), dim3(TPB), 0, c->stream, c->nfac, c->nvo, c->d_fac_cell.p,
                       c->d_fac_local.p, c->cells.p, c->coords.p, c->x.p, c->mu, out);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int k_functional(cfdh_ctx *c, int kind, int marker, double *out) {
  if (c->dim == 3) return k3_functional(c, kind, marker, out);
  const int nb = 256;
  if (c->gen && (kind <= 3 || kind == 7)) {
    CHK(kg_functional_partials(c, kind, marker, nb));
  } else if (kind == 0 || kind == 1) {
    // a part without exterior facets (nfac == 0) still launches: the kernel then only writes zero partials, and the
    // rank takes part in the reduction below like every other one (skipping it would desynchronise the collectives)
    hipLaunchKernelGGL(draglift_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nfac, marker, c->nvo, c->d_fac_cell.p,
                       c->d_fac_local.p, c->d_fac_marker.p, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->mu,
                       c->red_partial.p);
  } else if (kind == 7) {
    hipLaunchKernelGGL(flux_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nfac, marker, c->nvo, c->d_fac_cell.p, c->d_fac_local.p,
                       c->d_fac_marker.p, c->cells.p, c->cell_owned.p, c->coords.p, c->x.p, c->red_partial.p);
  } else if (kind == 2 || kind == 3) {
    hipLaunchKernelGGL(l2_kernel, dim3(nb), dim3(TPB), 0, c->stream, c->nc, c->nvo, c->cells.p, c->cell_owned.p, c->coords.p,
                       c->x.p, c->red_partial.p);
  } else if (kind >= 4 && kind <= 6) {
    const int nu = 2 * c->nvo;
    const double *a = kind == 5 ? c->xprev.p : c->x.p;
    const double *b = kind == 6 ? c->xprev.p : nullptr;
    double v;
    CHK(reduce_dev(c, 1, nu, a, b, c->red_out.p));
    CHK(read_scalars(c, c->red_out.p, 1, &v));
    *out = v;
    return 0;
  } else {
    return cfdh_fail(c, CFDH_E_ARG, "unknown functional kind %d", kind);
  }
  hipLaunchKernelGGL(reduce_final_kernel<0>, dim3(2), dim3(TPB), 0, c->stream, nb, nb, c->red_partial.p, c->red_out.p,
                     scalar_mirror(c, c->red_out.p, 2));
  HIPCHK(c, hipGetLastError());
  CHK(finish_scalars(c, c->red_out.p, 2, 0));
  double v[2];
  CHK(read_scalars(c, c->red_out.p, 2, v));
  if (kind == 0 || kind == 7) *out = v[0];
  else if (kind == 1) *out = v[1];
  else if (kind == 2) *out = sqrt(v[0]);
  else *out = sqrt(v[1]);
  return 0;
}

// ---------------------------------------------------------------- halo pack
__global__ __launch_bounds__(TPB) void halo_pack_kernel(int n, int nvo, const int *__restrict__ idx, const double *__restrict__ vec,
                                                        double *__restrict__ buf) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const int v = idx[i];
  buf[3 * (size_t)i] = vec[2 * (size_t)v];
  buf[3 * (size_t)i + 1] = vec[2 * (size_t)v + 1];
  buf[3 * (size_t)i + 2] = vec[2 * (size_t)nvo + v];
}
__global__ __launch_bounds__(TPB) void halo_pack3_kernel(int n, int nvo, const int *__restrict__ idx, const double *__restrict__ vec,
                                                         double *__restrict__ buf) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const int v = idx[i];
  buf[4 * (size_t)i] = vec[3 * (size_t)v];
  buf[4 * (size_t)i + 1] = vec[3 * (size_t)v + 1];
  buf[4 * (size_t)i + 2] = vec[3 * (size_t)v + 2];
  buf[4 * (size_t)i + 3] = vec[3 * (size_t)nvo + v];
}
int k_halo_pack(cfdh_ctx *c, const double *vec) {
  const int n = (int)c->send_idx.n;
  if (n == 0) return 0;
  if (c->dim == 3) {
    hipLaunchKernelGGL(halo_pack3_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, c->nvo, c->send_idx.p, vec, c->send_buf.p);
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(halo_pack_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, c->stream, n, c->nvo, c->send_idx.p, vec,
                     c->send_buf.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}
