// Device-side set-up of the smoothed-aggregation hierarchies (gfx950, wave64).
//
// The reference rebuilds its preconditioner inside every Newton iteration (PCSetUp under SNES.solve,
// stabilized_schur.py:225-267); here the hierarchies are lagged, but every (re)build used to run on the host
// (OpenMP SpGEMM after a 195 MB download of the Jacobian blocks).  This file builds them where the matrix lives:
//
//   * aggregation        : distance-2 maximal independent set of the strength graph (hashed priorities, fixed-point
//                          iteration of three kernels), then two joining passes -- deterministic, no atomics on values
//   * sparse products    : one WAVEFRONT PER ROW, hash table in LDS (ds_cmpst / ds_add_f64), two passes (count, fill),
//                          rows binned by size so that the table fits; the finished row is sorted in LDS by a bitonic
//                          network and written with sorted columns.  Long rows of coarse levels use a dense accumulator
//                          per wavefront in global memory.
//   * operators          : P = (I - w D^-1 A) P0, R = P^T, A P, G = R (I - A W), Sb = 2W - W A W, Sc = P - W (A P),
//                          A_c = R (A P); the I - ... factors are applied on the fly inside the product kernels
//   * coarsest level     : dense Gauss-Jordan inverse (one launch per pivot, ping-pong), folded into the level above
//   * formats            : fp32 CSR copy / SELL-64 of the composite operators, written by kernels
//
// Everything is index/hash work on short rows: HBM/latency-bound integer code, no MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>

#include "cfdh_internal.hpp"

#define TPB 256

namespace {

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------------ scans
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int t = __shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
// inclusive scan of a[0..n) in place, 1024 elements per block; block totals to sums[]
__global__ __launch_bounds__(TPB) void scan_block_kernel(int n, int *__restrict__ a, int *__restrict__ sums) {
  __shared__ int wsum[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int base = blockIdx.x * 1024 + t * 4;
  int v[4], s = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) { v[q] = base + q < n ? a[base + q] : 0; s += v[q]; v[q] = s; }
  int inc = wave_incl_scan(s, lane);
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  int off = inc - s;
  for (int w = 0; w < wv; w++) off += wsum[w];
#pragma unroll
  for (int q = 0; q < 4; q++) if (base + q < n) a[base + q] = v[q] + off;
  if (t == TPB - 1) sums[blockIdx.x] = off + s;
}
__global__ __launch_bounds__(TPB) void scan_sums_kernel(int nb, int *__restrict__ sums) {
  __shared__ int wsum[4];
  __shared__ int carry_s;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += TPB) {
    const int v = b0 + t < nb ? sums[b0 + t] : 0;
    int inc = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int off = carry_s;
    for (int w = 0; w < wv; w++) off += wsum[w];
    if (b0 + t < nb) sums[b0 + t] = inc + off;
    __syncthreads();
    if (t == TPB - 1) carry_s = inc + off;
    __syncthreads();
  }
}
__global__ __launch_bounds__(TPB) void scan_add_kernel(int n, int *__restrict__ a, const int *__restrict__ sums) {
  const int b = blockIdx.x + 1;  // block 0 needs no offset
  const int off = sums[b - 1];
  const int base = b * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int q = 0; q < 4; q++) if (base + q < n) a[base + q] += off;
}

struct Dev {
  cfdh_ctx *c;
  hipStream_t s;
  dbuf<int> scan_sums;
  int rc = 0;
  explicit Dev(cfdh_ctx *c_) : c(c_), s(c_->stream) {}
  // inclusive scan in place
  int scan(int *a, int n) {
    if (n <= 0) return 0;
    const int nb = (n + 1023) / 1024;
    if ((size_t)nb > scan_sums.n) HIPCHK(c, scan_sums.alloc((size_t)nb + 1024));
    hipLaunchKernelGGL(scan_block_kernel, dim3(nb), dim3(TPB), 0, s, n, a, scan_sums.p);
    if (nb > 1) {
      hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(TPB), 0, s, nb, scan_sums.p);
      hipLaunchKernelGGL(scan_add_kernel, dim3(nb - 1), dim3(TPB), 0, s, n, a, scan_sums.p);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }
  int read_int(const int *p, int *out) {
    HIPCHK(c, hipMemcpyAsync(out, p, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return 0;
  }
};

// ------------------------------------------------------------------------------------------------ product operands
struct MatV { const int *rp; const int *col; const double *val; };
// left factor
struct AOpPlain {
  MatV A;
  __device__ int begin(int i) const { return A.rp[i]; }
  __device__ int end(int i) const { return A.rp[i + 1]; }
  __device__ int col(int k) const { return A.col[k]; }
  __device__ double val(int, int k, int) const { return A.val[k]; }
};
// I - omega D^-1 A  (prolongator smoothing)
struct AOpImwDA {
  MatV A;
  const double *dinv;
  double omega;
  const int *agg;  // rows of unknowns outside every aggregate stay empty (no coarse correction), as in the host build
  __device__ int begin(int i) const { return A.rp[i]; }
  __device__ int end(int i) const { return agg[i] >= 0 ? A.rp[i + 1] : A.rp[i]; }
  __device__ int col(int k) const { return A.col[k]; }
  __device__ double val(int i, int k, int j) const { return (i == j ? 1.0 : 0.0) - omega * dinv[i] * A.val[k]; }
};
// right factor
struct BOpPlain {
  MatV B;
  __device__ int begin(int k) const { return B.rp[k]; }
  __device__ int end(int k) const { return B.rp[k + 1]; }
  __device__ int col(int q) const { return B.col[q]; }
  __device__ double val(int, int q, int) const { return B.val[q]; }
};
// piecewise-constant tentative prolongator given by the aggregate ids (row k: one unit entry, none for agg < 0)
struct BOpAgg {
  const int *agg;
  __device__ int begin(int k) const { return k; }
  __device__ int end(int k) const { return k + (agg[k] >= 0 ? 1 : 0); }
  __device__ int col(int q) const { return agg[q]; }
  __device__ double val(int, int, int) const { return 1.0; }
};
// I - A W  (pre-smoothing + residual folded into the restriction)
struct BOpImAW {
  MatV A;
  const double *w;
  __device__ int begin(int k) const { return A.rp[k]; }
  __device__ int end(int k) const { return A.rp[k + 1]; }
  __device__ int col(int q) const { return A.col[q]; }
  __device__ double val(int k, int q, int j) const { return (k == j ? 1.0 : 0.0) - A.val[q] * w[j]; }
};

// upper bound of the entries of row i of A*B: sum of the B-row lengths
template <class AOp, class BOp>
__global__ __launch_bounds__(TPB) void spgemm_ub_kernel(int n, AOp A, BOp B, int *__restrict__ ub) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int s = 0;
  for (int k = A.begin(i), e = A.end(i); k < e; k++) { const int r = A.col(k); s += B.end(r) - B.begin(r); }
  ub[i] = s;
}
// rows sorted into size classes: lists[b * n + pos]
// size classes of a row (entries, or their upper bound): LDS tables of 32 / 128 / 512 / 2048 / 8192 slots at a load factor
// <= 0.75, class 5: dense accumulator.  merge45: no 8192-slot class (tables with values: 12 B per slot)
constexpr int NBIN = 6;
__global__ __launch_bounds__(TPB) void bin_rows_kernel(int n, const int *__restrict__ size, int merge45, int *__restrict__ lists,
                                                       int *__restrict__ cnt) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  const int s = i < n ? size[i] : 0;
  int b = s <= 0 ? -1 : (s <= 24 ? 0 : (s <= 96 ? 1 : (s <= 384 ? 2 : (s <= 1536 ? 3 : (s <= 6144 ? 4 : 5)))));  // -1: empty row, nothing to compute
  if (merge45 && b == 4) b = 5;
  // one atomic per wavefront and class (a counter per class is a single hot address: per-row atomics serialise at the L2,
  // 4 ms for the 336 k short rows of a finest-level product)
  const int lane = threadIdx.x & 63;
  for (int q = 0; q < NBIN; q++) {
    const unsigned long long m = __ballot(b == q);
    if (m == 0) continue;
    int base = 0;
    if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&cnt[q], __popcll(m));
    base = __shfl(base, __ffsll((long long)m) - 1);
    if (b == q) lists[(size_t)q * n + base + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
}
__global__ __launch_bounds__(TPB) void rowlen_kernel(int n, const int *__restrict__ rp, int *__restrict__ len) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) len[i] = rp[i + 1] - rp[i];
}

// bitonic sort of (keys[, vals]) in LDS by one wavefront; TS a power of two
template <int TS, bool NUM>
__device__ __forceinline__ void bitonic_sort(int *keys, double *vals, int lane) {
  for (int k = 2; k <= TS; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < TS / 2; t += 64) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // index with bit j cleared
        const int hi = lo | j;
        const bool up = (lo & k) == 0;
        const int a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b; keys[hi] = a;
          if (NUM) { const double x = vals[lo]; vals[lo] = vals[hi]; vals[hi] = x; }
        }
      }
      __syncthreads();
    }
}

// One wavefront per row: C(i,:) = sum_k A(i,k) B(k,:) accumulated in an LDS hash table.
// NUM = false: count the distinct columns -> Crp[i + 1];  NUM = true: values, sorted by column -> Ccol / Cval at Crp[i].
template <int TS, bool NUM, class AOp, class BOp>
__global__ __launch_bounds__(64) void spgemm_hash_kernel(const int *__restrict__ rows, const int *__restrict__ nrows, AOp A, BOp B,
                                                         int *__restrict__ Crp, int *__restrict__ Ccol, double *__restrict__ Cval,
                                                         int *__restrict__ fail) {
  __shared__ int keys[TS];
  __shared__ double vals[NUM ? TS : 1];
  if ((int)blockIdx.x >= *nrows) return;
  const int lane = threadIdx.x, i = rows[blockIdx.x];
  for (int t = lane; t < TS; t += 64) { keys[t] = -1; if (NUM) vals[t] = 0.0; }
  __syncthreads();
  const int as = A.begin(i), ae = A.end(i);
  const int sub = lane >> 3, l8 = lane & 7;
  for (int ka = as + sub; ka < ae; ka += 8) {
    const int k = A.col(ka);
    const double av = NUM ? A.val(i, ka, k) : 0.0;
    for (int q = B.begin(k) + l8, qe = B.end(k); q < qe; q += 8) {
      const int j = B.col(q);
      unsigned h = ((unsigned)j * 2654435761u) >> 7;
      int probes = 0;
      for (;;) {
        h &= (unsigned)(TS - 1);
        const int old = atomicCAS(&keys[h], -1, j);
        if (old == -1 || old == j) break;
        h++;
        if (++probes > TS) { *fail = 1; break; }
      }
      if (NUM && probes <= TS) atomicAdd(&vals[h & (TS - 1)], av * B.val(k, q, j));
    }
  }
  __syncthreads();
  if (!NUM) {
    int cnt = 0;
    for (int t = lane; t < TS; t += 64) cnt += keys[t] >= 0 ? 1 : 0;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
    if (lane == 0) Crp[i + 1] = cnt;
    return;
  }
  for (int t = lane; t < TS; t += 64) if (keys[t] < 0) keys[t] = 0x7fffffff;
  __syncthreads();
  bitonic_sort<TS, true>(keys, vals, lane);
  const int base = Crp[i], cnt = Crp[i + 1] - base;
  for (int t = lane; t < cnt; t += 64) { Ccol[base + t] = keys[t]; Cval[base + t] = vals[t]; }
}

// Short rows (at most 24 products: the prolongator and A P on the finest level, hundreds of thousands of rows): eight lanes
// per row, 32 rows per workgroup, 32-slot tables -- a wavefront per row would be all launch overhead there.
template <bool NUM, class AOp, class BOp>
__global__ __launch_bounds__(256) void spgemm_small_kernel(const int *__restrict__ rows, const int *__restrict__ nrows, AOp A, BOp B,
                                                           int *__restrict__ Crp, int *__restrict__ Ccol, double *__restrict__ Cval) {
  constexpr int TS = 32;
  __shared__ int keys[32][TS];
  __shared__ double vals[NUM ? 32 : 1][NUM ? TS : 1];
  const int g = threadIdx.x >> 3, l8 = threadIdx.x & 7;
  const int idx = blockIdx.x * 32 + g;
  const bool live = idx < *nrows;
  const int i = live ? rows[idx] : 0;
  int *kk = keys[g];
  double *vv = NUM ? vals[g] : nullptr;
  for (int t = l8; t < TS; t += 8) { kk[t] = -1; if (NUM) vv[t] = 0.0; }
  __syncthreads();
  if (live)
    for (int ka = A.begin(i), ae = A.end(i); ka < ae; ka++) {
      const int k = A.col(ka);
      const double av = NUM ? A.val(i, ka, k) : 0.0;
      for (int q = B.begin(k) + l8, qe = B.end(k); q < qe; q += 8) {
        const int j = B.col(q);
        unsigned h = ((unsigned)j * 2654435761u) >> 7;
        for (;;) {  // at most 24 distinct keys in 32 slots: a free slot always exists
          h &= (unsigned)(TS - 1);
          const int old = atomicCAS(&kk[h], -1, j);
          if (old == -1 || old == j) break;
          h++;
        }
        if (NUM) atomicAdd(&vv[h], av * B.val(k, q, j));
      }
    }
  __syncthreads();
  if (!NUM) {
    int cnt = 0;
    for (int t = l8; t < TS; t += 8) cnt += kk[t] >= 0 ? 1 : 0;
    cnt += __shfl_xor(cnt, 1); cnt += __shfl_xor(cnt, 2); cnt += __shfl_xor(cnt, 4);
    if (live && l8 == 0) Crp[i + 1] = cnt;
    return;
  }
  for (int t = l8; t < TS; t += 8) if (kk[t] < 0) kk[t] = 0x7fffffff;
  __syncthreads();
  for (int k = 2; k <= TS; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = l8; t < TS / 2; t += 8) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
        const bool up = (lo & k) == 0;
        const int a = kk[lo], b = kk[hi];
        if ((a > b) == up) { kk[lo] = b; kk[hi] = a; const double x = vv[lo]; vv[lo] = vv[hi]; vv[hi] = x; }
      }
      __syncthreads();
    }
  if (live) {
    const int base = Crp[i], cnt = Crp[i + 1] - base;
    for (int t = l8; t < cnt; t += 8) { Ccol[base + t] = kk[t]; Cval[base + t] = vv[t]; }
  }
}

// Long rows (coarse levels: few columns in total): dense accumulator per wavefront in global memory.  marker holds the
// stamp of the last row that touched a column; acc is kept zero between rows.
template <bool NUM, class AOp, class BOp>
__global__ __launch_bounds__(64) void spgemm_dense_kernel(const int *__restrict__ rows, const int *__restrict__ nrows, AOp A, BOp B, int m,
                                                          int *__restrict__ marker, double *__restrict__ acc, int *__restrict__ Crp,
                                                          int *__restrict__ Ccol, double *__restrict__ Cval) {
  __shared__ int cnt_s;
  const int lane = threadIdx.x;
  int *mk = marker + (size_t)blockIdx.x * m;
  double *ac = NUM ? acc + (size_t)blockIdx.x * m : nullptr;
  const int nr = *nrows;
  for (int idx = blockIdx.x; idx < nr; idx += gridDim.x) {
    const int i = rows[idx], stamp = idx + 1;
    if (lane == 0) cnt_s = 0;
    __syncthreads();
    const int sub = lane >> 3, l8 = lane & 7;
    for (int ka = A.begin(i) + sub, ae = A.end(i); ka < ae; ka += 8) {
      const int k = A.col(ka);
      const double av = NUM ? A.val(i, ka, k) : 0.0;
      for (int q = B.begin(k) + l8, qe = B.end(k); q < qe; q += 8) {
        const int j = B.col(q);
        if (NUM) atomicAdd(&ac[j], av * B.val(k, q, j));
        if (atomicExch(&mk[j], stamp) != stamp) atomicAdd(&cnt_s, 1);
      }
    }
    __threadfence();
    __syncthreads();
    if (!NUM) {
      if (lane == 0) Crp[i + 1] = cnt_s;
    } else {
      // columns in ascending order: scan the marker array
      int out = Crp[i];
      for (int j0 = 0; j0 < m; j0 += 64) {
        const int j = j0 + lane;
        // marker / accumulator were written by atomics at the L2: read them there (a plain load may hit a stale L1 line)
        const bool hit = j < m && __hip_atomic_load(&mk[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == stamp;
        const unsigned long long bal = __ballot(hit);
        if (hit) {
          const int pos = out + __popcll(bal & ((1ull << lane) - 1ull));
          Ccol[pos] = j; Cval[pos] = __hip_atomic_load(&ac[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&ac[j], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        out += __popcll(bal);
      }
      __threadfence();
    }
    __syncthreads();
  }
}

// sort the entries of each row by column (after a scatter-built transpose)
template <int TS>
__global__ __launch_bounds__(64) void sort_rows_kernel(const int *__restrict__ rows, const int *__restrict__ nrows, const int *__restrict__ rp,
                                                       int *__restrict__ col, double *__restrict__ val) {
  __shared__ int keys[TS];
  __shared__ double vals[TS];
  if ((int)blockIdx.x >= *nrows) return;
  const int lane = threadIdx.x, i = rows[blockIdx.x];
  const int base = rp[i], cnt = rp[i + 1] - base;
  for (int t = lane; t < TS; t += 64) {
    keys[t] = t < cnt ? col[base + t] : 0x7fffffff;
    vals[t] = t < cnt ? val[base + t] : 0.0;
  }
  __syncthreads();
  bitonic_sort<TS, true>(keys, vals, lane);
  for (int t = lane; t < cnt; t += 64) { col[base + t] = keys[t]; val[base + t] = vals[t]; }
}

__global__ __launch_bounds__(TPB) void transpose_count_kernel(int nnz, const int *__restrict__ col, int *__restrict__ cnt1) {
  const int k = blockIdx.x * TPB + threadIdx.x;
  if (k < nnz) atomicAdd(&cnt1[col[k] + 1], 1);
}
__global__ __launch_bounds__(TPB) void transpose_fill_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col,
                                                             const double *__restrict__ val, const int *__restrict__ trp,
                                                             int *__restrict__ cursor, int *__restrict__ tcol, double *__restrict__ tval) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
    const int j = col[k];
    const int pos = trp[j] + atomicAdd(&cursor[j], 1);
    tcol[pos] = i; tval[pos] = val[k];
  }
}

// ------------------------------------------------------------------------------------------------ level quantities
// dinv = 1 / a_ii (1 if absent or zero), offd = number of non-zero off-diagonal entries
__global__ __launch_bounds__(TPB) void level_diag_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col,
                                                         const double *__restrict__ val, double *__restrict__ dinv, int *__restrict__ offd) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  double d = 1.0;
  int o = 0;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
    const double v = val[k];
    if (col[k] == i) { if (v != 0.0) d = 1.0 / v; }
    else if (v != 0.0) o++;
  }
  dinv[i] = d;
  offd[i] = o;
}
__global__ __launch_bounds__(TPB) void level_weight_kernel(int n, const double *__restrict__ dinv, const int *__restrict__ offd, double itheta,
                                                           double *__restrict__ w) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) w[i] = dinv[i] * (offd[i] == 0 ? 1.0 : itheta);
}
// Sb = 2W - W A W on the pattern of A
__global__ __launch_bounds__(TPB) void sb_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ w, double *__restrict__ out) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int i = gid >> 3, l = gid & 7;
  if (i >= n) return;
  const double wi = w[i];
  for (int k = rp[i] + l, e = rp[i + 1]; k < e; k += 8) {
    const int j = col[k];
    out[k] = -1.0 * wi * val[k] * w[j] + (j == i ? 2.0 * wi : 0.0);
  }
}
// Sc = P - W (A P) on the pattern of A P (which contains the pattern of P: A has a diagonal); both rows sorted
__global__ __launch_bounds__(TPB) void sc_kernel(int n, const int *__restrict__ rpAP, const int *__restrict__ colAP, const double *__restrict__ valAP,
                                                 const int *__restrict__ rpP, const int *__restrict__ colP, const double *__restrict__ valP,
                                                 const double *__restrict__ w, double *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int kp = rpP[i];
  const int ep = rpP[i + 1];
  const double wi = w[i];
  for (int k = rpAP[i], e = rpAP[i + 1]; k < e; k++) {
    const int j = colAP[k];
    while (kp < ep && colP[kp] < j) kp++;
    double v = -1.0 * wi * valAP[k];
    if (kp < ep && colP[kp] == j) v += valP[kp];
    out[k] = v;
  }
}
__global__ __launch_bounds__(TPB) void to_float_kernel(int n, const double *__restrict__ a, float *__restrict__ b) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) b[i] = (float)a[i];
}
// y = dinv .* (A x)
__global__ __launch_bounds__(TPB) void spmv_dinv_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ dinv, const double *__restrict__ x, double *__restrict__ y) {
  const int gid = blockIdx.x * TPB + threadIdx.x;
  const int i = gid >> 3, l = gid & 7;
  double a = 0.0;
  if (i < n)
    for (int k = rp[i] + l, e = rp[i + 1]; k < e; k += 8) a += val[k] * x[col[k]];
  a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
  if (i < n && l == 0) y[i] = dinv[i] * a;
}

// v[i] = element i of the sequence st <- a st + c (mod 2^64), st_0 = seed, mapped to [-0.5, 0.5): the (i+1)-fold
// composition of the affine map by binary powering
__global__ __launch_bounds__(TPB) void lcg_vector_kernel(int n, double *__restrict__ v) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  unsigned long long ra = 1ull, rc = 0ull;                                            // identity map
  unsigned long long pa = 6364136223846793005ull, pc = 1442695040888963407ull;        // one step
  for (unsigned k = (unsigned)i + 1u; k; k >>= 1) {
    if (k & 1u) { rc = pa * rc + pc; ra = pa * ra; }
    pc = pa * pc + pc; pa = pa * pa;
  }
  const unsigned long long st = ra * 0x9E3779B97F4A7C15ull + rc;
  v[i] = ((st >> 11) * (1.0 / 9007199254740992.0)) - 0.5;
}

// ---- SELL-64 (fp32) from sorted CSR
__global__ __launch_bounds__(64) void sell_width_kernel(int n, const int *__restrict__ rp, int *__restrict__ sptr1, int *__restrict__ maxw) {
  const int sl = blockIdx.x, r = sl * 64 + threadIdx.x;
  int w = r < n ? rp[r + 1] - rp[r] : 0;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) w = max(w, __shfl_xor(w, d));
  if (threadIdx.x == 0) { sptr1[sl + 1] = 64 * w; atomicMax(maxw, w); }
}
__global__ __launch_bounds__(TPB) void sell_fill_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                        const double *__restrict__ colw, const int *__restrict__ sptr, int *__restrict__ scol,
                                                        float *__restrict__ sval, float *__restrict__ svalw) {
  const int r = blockIdx.x * TPB + threadIdx.x;
  if (r >= n) return;
  const int sl = r >> 6, lane = r & 63;
  const size_t p0 = (size_t)sptr[sl];
  for (int k = rp[r], e = rp[r + 1], j = 0; k < e; k++, j++) {
    const size_t p = p0 + (size_t)j * 64 + lane;
    scol[p] = col[k]; sval[p] = (float)val[k];
    if (svalw) svalw[p] = (float)(val[k] * colw[col[k]]);
  }
}

// ---- dense coarsest level
__global__ __launch_bounds__(TPB) void densify_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                      double *__restrict__ D, double *__restrict__ absdiag) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  double ad = 0.0;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) { D[(size_t)i * n + col[k]] = val[k]; if (col[k] == i) ad += fabs(val[k]); }
  absdiag[i] = ad;
}
__global__ __launch_bounds__(TPB) void trace_shift_kernel(int n, const double *__restrict__ absdiag, double *__restrict__ D) {
  // alpha = sum |a_ii| / n^2 added to every entry (regularisation of the constant null vector); one block sums, all add
  __shared__ double sh[TPB];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += TPB) s += absdiag[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int d = TPB / 2; d > 0; d >>= 1) { if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d]; __syncthreads(); }
  const double alpha = sh[0] / n / n;
  const size_t N = (size_t)n * n;
  for (size_t k = (size_t)blockIdx.x * TPB + threadIdx.x; k < N; k += (size_t)gridDim.x * TPB) D[k] += alpha;
}
// one Gauss-Jordan step (pivot p, no row exchange): out = in with column p eliminated; in-place inverse after n steps
__global__ __launch_bounds__(TPB) void gj_step_kernel(int n, int p, const double *__restrict__ in, double *__restrict__ out, int *__restrict__ fail) {
  const size_t idx = (size_t)blockIdx.x * TPB + threadIdx.x;
  if (idx >= (size_t)n * n) return;
  const int i = (int)(idx / n), j = (int)(idx % n);
  const double piv = in[(size_t)p * n + p];
  if (idx == 0 && !(fabs(piv) > 0.0 && isfinite(piv))) *fail = 1;
  const double d = 1.0 / piv;
  double v;
  if (i == p) v = (j == p) ? d : in[idx] * d;
  else {
    const double f = in[(size_t)i * n + p];
    v = (j == p) ? -f * d : in[idx] - f * (in[(size_t)p * n + j] * d);
  }
  out[idx] = v;
}
// D_up[i][:] = sum_k Sc(i,k) Ainv[k][:]   (fp32 result)
__global__ __launch_bounds__(TPB) void fold_dense_kernel(int n, int nc, const int *__restrict__ rp, const int *__restrict__ col,
                                                         const double *__restrict__ val, const double *__restrict__ Ainv, float *__restrict__ out) {
  const int i = blockIdx.x;
  for (int j = threadIdx.x; j < nc; j += TPB) {
    double a = 0.0;
    for (int k = rp[i], e = rp[i + 1]; k < e; k++) a += val[k] * Ainv[(size_t)col[k] * nc + j];
    out[(size_t)i * nc + j] = (float)a;
  }
}

// ------------------------------------------------------------------------------------------------ aggregation
// Strength graph |a_ij| >= theta sqrt(|a_ii| |a_jj|) and a distance-2 maximal independent set of it by fixed-point
// iteration (Bell, Dalton, Olson 2012): every undecided vertex carries the key (state, hashed priority, index); two
// propagation passes take the maximum over the strong neighbourhood; a vertex that sees its own key is a root, one that
// sees a root is out.  Roots claim their strong neighbours (phase 1), unclaimed vertices join the aggregate of their
// strongest claimed neighbour (phase 2, two passes), the rest stay without coarse correction (agg = -1: their row of P
// is empty and the smoother treats them exactly when they are isolated).
__global__ __launch_bounds__(TPB) void absdiag_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                      double *__restrict__ d) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  double v = 0.0;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) if (col[k] == i) v = fabs(val[k]);
  d[i] = v;
}
// Connection strength for "strongest neighbour" choices, coarsened to 16 mantissa bits: on structured meshes many connections
// are equal up to rounding, and the tetrahedral assembly is reproducible to 1e-12 only (LDS-arrival order) -- comparing raw
// values would let that noise pick different aggregates from run to run.  Ties go to the lower aggregate id.
__device__ __forceinline__ unsigned qstrength(double a) { return __float_as_uint((float)a) >> 8; }
__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// state: 0 out (not a root), 1 undecided, 2 root.  key = state << 62 | priority (32 bits) << 30 | index (30 bits).
// Priority modes (the packing density of the roots decides the coarsening ratio, and with it the iteration count):
//   0 hash       : random -- a jammed random packing, ~15 % fewer aggregates than an ordered sweep
//   1 index      : lowest index first = the roots of the sequential greedy pass (long dependency chains: many rounds)
//   2 front      : number of strong neighbours already decided "out", then hash: new roots are placed right next to the
//                  finished region (distance 3 from its roots), i.e. the packing grows as tight fronts
//   3 front+seed : as 2, with one vertex in 64 eligible in the first round (fewer, larger fronts)
__device__ __forceinline__ unsigned long long mis_pack(int state, unsigned prio, int i) {
  return ((unsigned long long)state << 62) | ((unsigned long long)prio << 30) | (unsigned long long)(unsigned)i;
}
__global__ __launch_bounds__(TPB) void mis_init_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                       const double *__restrict__ d, double theta, int *__restrict__ state) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  // vertices without strong neighbours never become roots (they would form singleton aggregates)
  bool any = false;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
    const int j = col[k];
    if (j != i && fabs(val[k]) >= theta * sqrt(d[i] * d[j])) { any = true; break; }
  }
  state[i] = any ? 1 : 0;
}
__global__ __launch_bounds__(TPB) void mis_key_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ d, double theta, const int *__restrict__ state, int mode, int round,
                                                      unsigned long long *__restrict__ key) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const int st = state[i];
  unsigned prio;
  if (mode == 1) prio = (unsigned)(n - 1 - i);
  else if (mode >= 4) {
    // tiles of 2^(mode) consecutive (Morton-ordered) vertices: random order of the tiles, ascending index inside a tile -- the
    // sequential greedy sweep, run in all tiles at once; a tile adapts to its finished neighbours like the sweep to earlier rows
    const int tb = mode > 16 ? 16 : mode;
    prio = ((hash32((unsigned)(i >> tb)) >> 16) << 16) | (0xffffu - (unsigned)(i & ((1 << tb) - 1)));
  } else {
    prio = hash32((unsigned)i) >> 8;  // 24 bits
    if (mode >= 2 && st == 1) {
      int cnt = 0;
      for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
        const int j = col[k];
        if (j != i && state[j] == 0 && fabs(val[k]) >= theta * sqrt(d[i] * d[j])) cnt++;
      }
      prio |= (unsigned)(cnt > 127 ? 127 : cnt) << 24;
      // seeding: in the first round only one vertex in 64 may win (state 0-like key); the others wait
      if (mode == 3 && round == 0 && (hash32((unsigned)i ^ 0x9e3779b9u) & 63u) != 0) { key[i] = mis_pack(0, prio, i); return; }
    }
  }
  key[i] = mis_pack(st, prio, i);
}
// out[i] = max of in[] over the strong closed neighbourhood
__global__ __launch_bounds__(TPB) void mis_propagate_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col,
                                                            const double *__restrict__ val, const double *__restrict__ d, double theta,
                                                            const unsigned long long *__restrict__ in, unsigned long long *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  unsigned long long m = in[i];
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
    const int j = col[k];
    if (j == i || !(fabs(val[k]) >= theta * sqrt(d[i] * d[j]))) continue;
    const unsigned long long kj = in[j];
    m = kj > m ? kj : m;
  }
  out[i] = m;
}
__global__ __launch_bounds__(TPB) void mis_decide_kernel(int n, const unsigned long long *__restrict__ key, const unsigned long long *__restrict__ t2,
                                                         int *__restrict__ state, int *__restrict__ undecided) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n || state[i] != 1) return;
  const unsigned long long m = t2[i];
  if ((int)(m >> 62) == 2) state[i] = 0;              // a root within distance two
  else if (m == key[i]) state[i] = 2;                 // largest undecided key within distance two (a waiting vertex carries a state-0 key: never equal)
  else *undecided = 1;                                // a flag, not a count: no atomic
}
__global__ __launch_bounds__(TPB) void agg_number_roots_kernel(int n, const int *__restrict__ state, int *__restrict__ flag1) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i < n) flag1[i + 1] = state[i] == 2 ? 1 : 0;
}
// phase 1: roots and their strong neighbours (a vertex next to several roots joins the strongest connection, ties: lowest root)
__global__ __launch_bounds__(TPB) void agg_phase1_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                         const double *__restrict__ d, double theta, const int *__restrict__ state,
                                                         const int *__restrict__ rootid /*inclusive scan of the root flags, shifted*/,
                                                         int *__restrict__ agg) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  if (state[i] == 2) { agg[i] = rootid[i + 1] - 1; return; }
  int best = -1;
  unsigned bq = 0;
  for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
    const int j = col[k];
    if (j == i || state[j] != 2) continue;
    const double a = fabs(val[k]);
    if (!(a >= theta * sqrt(d[i] * d[j]))) continue;
    const unsigned q = qstrength(a);
    const int id = rootid[j + 1] - 1;
    if (best < 0 || q > bq || (q == bq && id < best)) { bq = q; best = id; }
  }
  agg[i] = best;
}
// Secondary aggregates in the gaps of the root packing: a vertex left over by phase 1 (no root next to it) whose strong
// neighbourhood holds at least two more leftovers becomes a secondary root when it beats all its leftover neighbours
// (priority: number of leftover neighbours, then hash); it claims those neighbours.
__global__ __launch_bounds__(TPB) void agg_gap_key_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                          const double *__restrict__ d, double theta, const int *__restrict__ agg, int minleft,
                                                          unsigned long long *__restrict__ key) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  unsigned long long kk = 0;
  if (agg[i] < 0) {
    int cnt = 0;
    for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
      const int j = col[k];
      if (j != i && agg[j] < 0 && fabs(val[k]) >= theta * sqrt(d[i] * d[j])) cnt++;
    }
    if (cnt >= minleft) kk = mis_pack(1, ((unsigned)(cnt > 127 ? 127 : cnt) << 24) | (hash32((unsigned)i) >> 8), i);
  }
  key[i] = kk;
}
__global__ __launch_bounds__(TPB) void agg_gap_root_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                           const double *__restrict__ d, double theta, const unsigned long long *__restrict__ key,
                                                           int *__restrict__ flag1) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long ki = key[i];
  bool root = ki != 0;
  for (int k = rp[i], e = rp[i + 1]; k < e && root; k++) {
    const int j = col[k];
    if (j != i && fabs(val[k]) >= theta * sqrt(d[i] * d[j]) && key[j] > ki) root = false;
  }
  flag1[i + 1] = root ? 1 : 0;
}
__global__ __launch_bounds__(TPB) void agg_gap_claim_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                            const double *__restrict__ d, double theta, const int *__restrict__ flagscan /*inclusive*/,
                                                            int na0, const int *__restrict__ in, int *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int a0 = in[i];
  if (a0 < 0) {
    if (flagscan[i + 1] != flagscan[i]) a0 = na0 + flagscan[i + 1] - 1;
    else {
      unsigned bq = 0;
      for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
        const int j = col[k];
        if (j == i || in[j] >= 0 || flagscan[j + 1] == flagscan[j]) continue;  // j is a secondary root?
        const double a = fabs(val[k]);
        if (!(a >= theta * sqrt(d[i] * d[j]))) continue;
        const unsigned q = qstrength(a);
        const int id = na0 + flagscan[j + 1] - 1;
        if (a0 < 0 || q > bq || (q == bq && id < a0)) { bq = q; a0 = id; }
      }
    }
  }
  out[i] = a0;
}
// phase 2: an unaggregated vertex joins the aggregate of its strongest aggregated strong neighbour (reads `in`, writes `out`)
__global__ __launch_bounds__(TPB) void agg_phase2_kernel(int n, const int *__restrict__ rp, const int *__restrict__ col, const double *__restrict__ val,
                                                         const double *__restrict__ d, double theta, const int *__restrict__ in,
                                                         int *__restrict__ out) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int a0 = in[i];
  if (a0 < 0) {
    unsigned bq = 0;
    for (int k = rp[i], e = rp[i + 1]; k < e; k++) {
      const int j = col[k];
      if (j == i || in[j] < 0) continue;
      const double a = fabs(val[k]);
      if (!(a >= theta * sqrt(d[i] * d[j]))) continue;
      const unsigned q = qstrength(a);
      if (a0 < 0 || q > bq || (q == bq && in[j] < a0)) { bq = q; a0 = in[j]; }
    }
  }
  out[i] = a0;
}

template <class AOp, class BOp>
struct Spgemm {
  // C = A * B with sorted columns.  n rows, m columns.  Returns nonzero on failure (c->err set).
  static int run(Dev &dv, int n, int m, AOp A, BOp B, CsrDev &C, dbuf<int> &lists, dbuf<int> &tmp, dbuf<int> &cnts) {
    cfdh_ctx *c = dv.c;
    hipStream_t s = dv.s;
    C.n = n; C.m = m;
    HIPCHK(c, C.rowptr.alloc((size_t)n + 1));
    HIPCHK(c, hipMemsetAsync(C.rowptr.p, 0, sizeof(int) * ((size_t)n + 1), s));
    if ((size_t)NBIN * n > lists.n) HIPCHK(c, lists.alloc((size_t)NBIN * n + 64));
    if ((size_t)n > tmp.n) HIPCHK(c, tmp.alloc((size_t)n + 64));
    if (cnts.n < 16) HIPCHK(c, cnts.alloc(16));
    int *cnt = cnts.p, *fail = cnts.p + 14;
    HIPCHK(c, hipMemsetAsync(fail, 0, sizeof(int), s));
    const dim3 gr((n + TPB - 1) / TPB), bl(TPB);
    int hc[16];
    for (int pass = 0; pass < 2; pass++) {
      HIPCHK(c, hipMemsetAsync(cnt, 0, sizeof(int) * 8, s));
      if (pass == 0) hipLaunchKernelGGL((spgemm_ub_kernel<AOp, BOp>), gr, bl, 0, s, n, A, B, tmp.p);
      else hipLaunchKernelGGL(rowlen_kernel, gr, bl, 0, s, n, C.rowptr.p, tmp.p);
      hipLaunchKernelGGL(bin_rows_kernel, gr, bl, 0, s, n, tmp.p, pass, lists.p, cnt);
      HIPCHK(c, hipMemcpyAsync(hc, cnt, sizeof(int) * 16, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
#define HASH_LAUNCH(TS, NUM, b)                                                                                                  \
  if (hc[b] > 0)                                                                                                                 \
    hipLaunchKernelGGL((spgemm_hash_kernel<TS, NUM, AOp, BOp>), dim3(hc[b]), dim3(64), 0, s, lists.p + (size_t)(b) * n, cnt + (b), A, B, \
                       C.rowptr.p, C.col.p, C.val.p, fail)
      if (hc[0] > 0) {
        if (pass == 0) hipLaunchKernelGGL((spgemm_small_kernel<false, AOp, BOp>), dim3((hc[0] + 31) / 32), dim3(256), 0, s, lists.p, cnt, A, B, C.rowptr.p, C.col.p, C.val.p);
        else hipLaunchKernelGGL((spgemm_small_kernel<true, AOp, BOp>), dim3((hc[0] + 31) / 32), dim3(256), 0, s, lists.p, cnt, A, B, C.rowptr.p, C.col.p, C.val.p);
        hc[0] = 0;
      }
      if (pass == 0) { HASH_LAUNCH(32, false, 0); HASH_LAUNCH(128, false, 1); HASH_LAUNCH(512, false, 2); HASH_LAUNCH(2048, false, 3); HASH_LAUNCH(8192, false, 4); }
      else { HASH_LAUNCH(32, true, 0); HASH_LAUNCH(128, true, 1); HASH_LAUNCH(512, true, 2); HASH_LAUNCH(2048, true, 3); }
#undef HASH_LAUNCH
      if (hc[5] > 0) {
        if (m > 262144) return cfdh_fail(c, CFDH_E_STATE, "device SpGEMM: %d rows exceed the LDS table and the product has %d columns", hc[5], m);
        const int nb = std::min(hc[5], 256);
        dbuf<int> marker;
        dbuf<double> acc;
        HIPCHK(c, marker.alloc((size_t)nb * m));
        HIPCHK(c, marker.zero(s));
        if (pass == 1) { HIPCHK(c, acc.alloc((size_t)nb * m)); HIPCHK(c, acc.zero(s)); }
        if (pass == 0)
          hipLaunchKernelGGL((spgemm_dense_kernel<false, AOp, BOp>), dim3(nb), dim3(64), 0, s, lists.p + (size_t)5 * n, cnt + 5, A, B, m, marker.p,
                             (double *)nullptr, C.rowptr.p, C.col.p, C.val.p);
        else
          hipLaunchKernelGGL((spgemm_dense_kernel<true, AOp, BOp>), dim3(nb), dim3(64), 0, s, lists.p + (size_t)5 * n, cnt + 5, A, B, m, marker.p,
                             acc.p, C.rowptr.p, C.col.p, C.val.p);
        HIPCHK(c, hipStreamSynchronize(s));  // marker / acc are released at the end of this scope
      }
      HIPCHK(c, hipGetLastError());
      if (pass == 0) {
        CHK(dv.scan(C.rowptr.p + 1, n));
        int nnz = 0;
        CHK(dv.read_int(C.rowptr.p + n, &nnz));
        C.nnz = nnz;
        HIPCHK(c, C.col.alloc((size_t)std::max(nnz, 1)));
        HIPCHK(c, C.val.alloc((size_t)std::max(nnz, 1)));
      }
    }
    int hf = 0;
    CHK(dv.read_int(fail, &hf));
    if (hf) return cfdh_fail(c, CFDH_E_STATE, "device SpGEMM: hash table overflow");
    return 0;
  }
};

struct Builder {
  Dev dv;
  cfdh_ctx *c;
  hipStream_t s;
  dbuf<int> lists, tmp, cnts, offd;
  double tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  explicit Builder(cfdh_ctx *c_) : dv(c_), c(c_), s(c_->stream) {}

  int transpose(const CsrDev &P, CsrDev &R) {
    R.n = P.m; R.m = P.n; R.nnz = P.nnz;
    HIPCHK(c, R.rowptr.alloc((size_t)R.n + 1));
    HIPCHK(c, R.col.alloc((size_t)std::max(P.nnz, 1)));
    HIPCHK(c, R.val.alloc((size_t)std::max(P.nnz, 1)));
    HIPCHK(c, hipMemsetAsync(R.rowptr.p, 0, sizeof(int) * ((size_t)R.n + 1), s));
    if (P.nnz > 0) hipLaunchKernelGGL(transpose_count_kernel, dim3((P.nnz + TPB - 1) / TPB), dim3(TPB), 0, s, P.nnz, P.col.p, R.rowptr.p);
    CHK(dv.scan(R.rowptr.p + 1, R.n));
    dbuf<int> cursor;
    HIPCHK(c, cursor.alloc((size_t)R.n + 1));
    HIPCHK(c, cursor.zero(s));
    hipLaunchKernelGGL(transpose_fill_kernel, dim3((P.n + TPB - 1) / TPB), dim3(TPB), 0, s, P.n, P.rowptr.p, P.col.p, P.val.p, R.rowptr.p,
                       cursor.p, R.col.p, R.val.p);
    // rows were filled in arrival order: sort each by column
    if ((size_t)NBIN * R.n > lists.n) HIPCHK(c, lists.alloc((size_t)NBIN * R.n + 64));
    if ((size_t)R.n > tmp.n) HIPCHK(c, tmp.alloc((size_t)R.n + 64));
    if (cnts.n < 16) HIPCHK(c, cnts.alloc(16));
    HIPCHK(c, hipMemsetAsync(cnts.p, 0, sizeof(int) * 16, s));
    const dim3 gr((R.n + TPB - 1) / TPB), bl(TPB);
    hipLaunchKernelGGL(rowlen_kernel, gr, bl, 0, s, R.n, R.rowptr.p, tmp.p);
    hipLaunchKernelGGL(bin_rows_kernel, gr, bl, 0, s, R.n, tmp.p, 1, lists.p, cnts.p);
    int hc[16];
    HIPCHK(c, hipMemcpyAsync(hc, cnts.p, sizeof(int) * 16, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));  // also: cursor may go out of scope below
    if (hc[5] > 0) return cfdh_fail(c, CFDH_E_STATE, "device transpose: a restriction row has more than 1536 entries");
#define SORT_LAUNCH(TS, b) \
  if (hc[b] > 0) hipLaunchKernelGGL((sort_rows_kernel<TS>), dim3(hc[b]), dim3(64), 0, s, lists.p + (size_t)(b) * R.n, cnts.p + (b), R.rowptr.p, R.col.p, R.val.p)
    SORT_LAUNCH(32, 0); SORT_LAUNCH(128, 1); SORT_LAUNCH(512, 2); SORT_LAUNCH(2048, 3);
#undef SORT_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
  }

  // fp32 CSR copy / SELL-64 of a finished operator
  int formats(CsrDev &M, int parts, const double *colw = nullptr) {
    if (parts & CFDH_UP_CSRF) {
      HIPCHK(c, M.valf.alloc((size_t)std::max(M.nnz, 1)));
      if (M.nnz > 0) hipLaunchKernelGGL(to_float_kernel, dim3((M.nnz + TPB - 1) / TPB), dim3(TPB), 0, s, M.nnz, M.val.p, M.valf.p);
    }
    if (parts & CFDH_UP_SELL) {
      const int ns = (M.n + 63) / 64;
      M.nslice = ns;
      HIPCHK(c, M.sptr.alloc((size_t)ns + 1));
      HIPCHK(c, hipMemsetAsync(M.sptr.p, 0, sizeof(int) * ((size_t)ns + 1), s));
      if (cnts.n < 16) HIPCHK(c, cnts.alloc(16));
      HIPCHK(c, hipMemsetAsync(cnts.p + 12, 0, sizeof(int), s));
      hipLaunchKernelGGL(sell_width_kernel, dim3(ns), dim3(64), 0, s, M.n, M.rowptr.p, M.sptr.p, cnts.p + 12);
      CHK(dv.scan(M.sptr.p + 1, ns));
      int tot = 0, mw = 0;
      HIPCHK(c, hipMemcpyAsync(&tot, M.sptr.p + ns, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipMemcpyAsync(&mw, cnts.p + 12, sizeof(int), hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
      M.sell_maxw = mw;
      HIPCHK(c, M.scol.alloc((size_t)std::max(tot, 1)));
      HIPCHK(c, M.sval.alloc((size_t)std::max(tot, 1)));
      HIPCHK(c, M.scol.zero(s)); HIPCHK(c, M.sval.zero(s));  // padding: column 0, value 0 (a harmless in-range gather)
      if (colw) { HIPCHK(c, M.svalw.alloc((size_t)std::max(tot, 1))); HIPCHK(c, M.svalw.zero(s)); }
      hipLaunchKernelGGL(sell_fill_kernel, dim3((M.n + TPB - 1) / TPB), dim3(TPB), 0, s, M.n, M.rowptr.p, M.col.p, M.val.p, colw, M.sptr.p,
                         M.scol.p, M.sval.p, colw ? M.svalw.p : (float *)nullptr);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
  }

  // dinv, spectral bound of D^-1 A (power iteration from the fixed pseudo-random vector of the host build), Jacobi weights
  int level_quantities(AmgLevel &L, const CsrDev &A, double ratio, double *lm_out) {
    const int n = A.n;
    L.n = n;
    HIPCHK(c, L.dinv.alloc(n)); HIPCHK(c, L.wdinv.alloc(n));
    if ((size_t)n > offd.n) HIPCHK(c, offd.alloc((size_t)n + 64));
    const dim3 gr((n + TPB - 1) / TPB), bl(TPB), gr8((unsigned)((8ll * n + TPB - 1) / TPB));
    hipLaunchKernelGGL(level_diag_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, L.dinv.p, offd.p);
    // start vector: the host build's LCG sequence, element i by jump-ahead (the same numbers, no host loop)
    dbuf<double> v, w;
    HIPCHK(c, v.alloc(n)); HIPCHK(c, w.alloc(n));
    hipLaunchKernelGGL(lcg_vector_kernel, gr, bl, 0, s, n, v.p);
    double *nrm = c->red_out.p + 30;
    for (int it = 0; it < 15; it++) {
      hipLaunchKernelGGL(spmv_dinv_kernel, gr8, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, L.dinv.p, v.p, w.p);
      CHK(v_norm_to_dev_local(c, n, w.p, nrm));  // rank-local operator: no reduction over the ranks
      CHK(v_scale_inv_dev(c, n, w.p, nrm, v.p));
    }
    double lm = 1.0;
    HIPCHK(c, hipMemcpyAsync(&lm, nrm, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    if (!(lm > 0) || !std::isfinite(lm)) lm = 1.0;
    L.lmax = 1.1 * lm;
    L.lmin = L.lmax / ratio;
    hipLaunchKernelGGL(level_weight_kernel, gr, bl, 0, s, n, L.dinv.p, offd.p, 2.0 / (L.lmax + L.lmin), L.wdinv.p);
    HIPCHK(c, hipGetLastError());
    *lm_out = lm;
    return 0;
  }

  int work_vectors(AmgLevel &L, int ncol) {
    const size_t nn = (size_t)L.n * ncol;
    HIPCHK(c, L.x.alloc(nn)); HIPCHK(c, L.b.alloc(nn)); HIPCHK(c, L.r.alloc(nn));
    HIPCHK(c, L.d0.alloc(nn)); HIPCHK(c, L.d1.alloc(nn));
    return 0;
  }

  // aggregate ids on the device; *na_out = number of aggregates
  int aggregate(const CsrDev &A, double theta, int gap, dbuf<int> &agg, int *na_out) {
    const int n = A.n;
    const dim3 gr((n + TPB - 1) / TPB), bl(TPB);
    dbuf<double> d;
    dbuf<int> state, flag, agg2;
    dbuf<unsigned long long> t1, t2;
    HIPCHK(c, d.alloc(n)); HIPCHK(c, state.alloc(n)); HIPCHK(c, flag.alloc((size_t)n + 1)); HIPCHK(c, agg.alloc(n)); HIPCHK(c, agg2.alloc(n));
    HIPCHK(c, t1.alloc(n)); HIPCHK(c, t2.alloc(n));
    if (cnts.n < 16) HIPCHK(c, cnts.alloc(16));
    hipLaunchKernelGGL(absdiag_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p);
    hipLaunchKernelGGL(mis_init_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, state.p);
    // defaults measured on the four 2-D / 3-D bench configurations against the sequential greedy pass of the host build
    // (DESIGN.md section 6): front priority + secondary roots with >= 4 leftover neighbours reproduce its iteration counts
    // within +-1 or better; 2 or 3 leftover neighbours are better still on three of them and 50 % worse on the stenosis
    static const int mode = getenv("CFDH_AGG_PRIO") ? atoi(getenv("CFDH_AGG_PRIO")) : 2;
    dbuf<unsigned long long> key;
    HIPCHK(c, key.alloc(n));
    int rounds = 0;
    for (int round = 0; round < 4096; round++) {
      HIPCHK(c, hipMemsetAsync(cnts.p + 13, 0, sizeof(int), s));
      hipLaunchKernelGGL(mis_key_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, state.p, mode, round, key.p);
      hipLaunchKernelGGL(mis_propagate_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, key.p, t1.p);
      hipLaunchKernelGGL(mis_propagate_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, t1.p, t2.p);
      hipLaunchKernelGGL(mis_decide_kernel, gr, bl, 0, s, n, key.p, t2.p, state.p, cnts.p + 13);
      rounds++;
      // the count of still-undecided vertices is read every fourth round (a round after the fixed point changes nothing)
      if ((round & 3) == 3 || mode == 0) {
        int und = 0;
        CHK(dv.read_int(cnts.p + 13, &und));
        if (und == 0) break;
      }
    }
    if (c->opt.verbose > 1) fprintf(stderr, "[cfdh]   aggregation: n %d, %d MIS-2 rounds (priority mode %d)\n", n, rounds, mode);
    HIPCHK(c, hipMemsetAsync(flag.p, 0, sizeof(int), s));
    hipLaunchKernelGGL(agg_number_roots_kernel, gr, bl, 0, s, n, state.p, flag.p);
    CHK(dv.scan(flag.p + 1, n));
    hipLaunchKernelGGL(agg_phase1_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, state.p, flag.p, agg.p);
    int na = 0;
    CHK(dv.read_int(flag.p + n, &na));
    if (gap > 0) {  // 0: off, k: secondary roots with >= k leftover neighbours
      hipLaunchKernelGGL(agg_gap_key_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, agg.p, gap, key.p);
      HIPCHK(c, hipMemsetAsync(flag.p, 0, sizeof(int), s));
      hipLaunchKernelGGL(agg_gap_root_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, key.p, flag.p);
      CHK(dv.scan(flag.p + 1, n));
      hipLaunchKernelGGL(agg_gap_claim_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, flag.p, na, agg.p, agg2.p);
      int na2 = 0;
      CHK(dv.read_int(flag.p + n, &na2));
      na += na2;
      HIPCHK(c, hipMemcpyAsync(agg.p, agg2.p, sizeof(int) * n, hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(agg_phase2_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, agg.p, agg2.p);
    hipLaunchKernelGGL(agg_phase2_kernel, gr, bl, 0, s, n, A.rowptr.p, A.col.p, A.val.p, d.p, theta, agg2.p, agg.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(s));  // the temporaries above may go out of scope
    *na_out = na;
    return 0;
  }
};

// ---- filtered copies of the vertex graph with computed values (level-0 operators of the Cahouet-Chabard preconditioner)
// scalar proxy of the velocity block: mean of the diagonal of the dim x dim block; exact zeros off the diagonal are dropped
// (rows / columns of Dirichlet dofs), as the host build does
struct ProxyF {
  const double *A00;
  int dim, nvo;
  __device__ double value(int i, int k, int w) const {
    double v = 0.0;
    for (int q = 0; q < dim; q++) v += A00[(size_t)dim * dim * k + (size_t)q * (dim + 1)];
    return v / dim;
  }
  __device__ bool whole_row_identity(int) const { return false; }
  __device__ bool keep(int i, int k, int w) const { return w < nvo && (w == i || value(i, k, w) != 0.0); }
};
// H = (I + a'T) M_l + b' A11, T = diag(A11)/diag(L); identity on pressure-Dirichlet rows, their columns dropped
struct HF {
  const double *A11, *Lval, *Ml;
  const unsigned char *pbc;
  const int *vdiag;
  double alpha, beta;
  int nvo;
  __device__ bool whole_row_identity(int i) const { return (pbc[i] & 1) != 0; }
  __device__ bool keep(int, int, int w) const { return w < nvo && !(pbc[w] & 1); }
  __device__ double value(int i, int k, int w) const {
    double v = beta * A11[k];
    if (w == i) {
      const int kd = vdiag[i];
      const double T = Lval[kd] > 0 ? A11[kd] / Lval[kd] : 0.0;
      v += (1.0 + alpha * T) * Ml[i];
    }
    return v;
  }
};
template <class F>
__global__ __launch_bounds__(TPB) void vg_count_kernel(int n, const int *__restrict__ vptr, const int *__restrict__ vcol, F f, int *__restrict__ rp1) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int cnt = 0;
  if (f.whole_row_identity(i)) cnt = 1;
  else for (int k = vptr[i], e = vptr[i + 1]; k < e; k++) cnt += f.keep(i, k, vcol[k]) ? 1 : 0;
  rp1[i + 1] = cnt;
}
template <class F>
__global__ __launch_bounds__(TPB) void vg_fill_kernel(int n, const int *__restrict__ vptr, const int *__restrict__ vcol, F f, const int *__restrict__ rp,
                                                      int *__restrict__ col, double *__restrict__ val) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  int p = rp[i];
  if (f.whole_row_identity(i)) { col[p] = i; val[p] = 1.0; return; }
  for (int k = vptr[i], e = vptr[i + 1]; k < e; k++) {
    const int w = vcol[k];
    if (f.keep(i, k, w)) { col[p] = w; val[p] = f.value(i, k, w); p++; }
  }
}
template <class F>
int vg_build(cfdh_ctx *c, F f, CsrDev &out) {
  Dev dv(c);
  hipStream_t s = c->stream;
  const int n = c->nvo;
  out.n = out.m = n;
  HIPCHK(c, out.rowptr.alloc((size_t)n + 1));
  HIPCHK(c, hipMemsetAsync(out.rowptr.p, 0, sizeof(int), s));
  hipLaunchKernelGGL((vg_count_kernel<F>), dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, n, c->vptr.p, c->vcol.p, f, out.rowptr.p);
  CHK(dv.scan(out.rowptr.p + 1, n));
  int nnz = 0;
  CHK(dv.read_int(out.rowptr.p + n, &nnz));
  out.nnz = nnz;
  HIPCHK(c, out.col.alloc((size_t)std::max(nnz, 1)));
  HIPCHK(c, out.val.alloc((size_t)std::max(nnz, 1)));
  hipLaunchKernelGGL((vg_fill_kernel<F>), dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, n, c->vptr.p, c->vcol.p, f, out.rowptr.p, out.col.p, out.val.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

static void move_csr(CsrDev &dst, CsrDev &src) {
  dst.n = src.n; dst.m = src.m; dst.nnz = src.nnz;
  dst.rowptr.adopt(src.rowptr); dst.col.adopt(src.col); dst.val.adopt(src.val);
}

}  // namespace

// host aggregation of a device matrix (CFDH_AMG_AGG=host: the sequential greedy pass of the host build, for comparison)
int cfdh_aggregate_host_csr(const CsrHost &A, double theta, std::vector<int> &agg);

// Build hierarchy H for the device matrix A0 (sorted columns; consumed).  Same operators as cfdh_amg_setup with the
// fused-cycle composites; the sweep-by-sweep operators (SELL of A, P, R) are not produced.
int cfdh_amg_setup_dev(cfdh_ctx *c, AmgHier &H, CsrDev &A0, bool singular, int ncol) {
  const double t_begin = now_ms();
  Builder B(c);
  hipStream_t s = c->stream;
  const cfdh_options &o = c->opt;
  H.clear();
  H.ncol = ncol;
  H.fused = false; H.nnz_G0 = H.nnz_S0 = 0;
  H.fine_nnz = A0.nnz;
  // depth cap (experiments: CFDH_A_MAXLEV / CFDH_L_MAXLEV for the velocity proxy / the pressure Laplacian): a hierarchy cut short
  // while its last level is still large is closed with two damped-Jacobi sweeps there (below)
  const char *ml_env = getenv(&H == &c->hA ? "CFDH_A_MAXLEV" : "CFDH_L_MAXLEV");
  const int maxlev = ml_env && atoi(ml_env) >= 1 ? std::min(atoi(ml_env), 16) : 16;
  static const double theta_env = getenv("CFDH_AMG_THETA") ? atof(getenv("CFDH_AMG_THETA")) : -1.0;
  const double theta = o.amg_theta >= 0 ? o.amg_theta : (theta_env >= 0 ? theta_env : (c->dim == 3 ? 0.02 : 0.07));
  const bool host_agg = getenv("CFDH_AMG_AGG") && !strcmp(getenv("CFDH_AMG_AGG"), "host");
  // secondary roots (section 6 of DESIGN.md): CFDH_AGG_GAP for both hierarchies, CFDH_AGG_GAP_A / _L for the velocity proxy / the pressure Laplacian
  // Defaults: 4 for the velocity proxy, off for the Laplacian -- on the cut-cell tree mesh of config 5 secondary roots in the
  // PRESSURE hierarchy cost 30 % more iterations (52.7 instead of 40.0 per step at 8 M DOF) and gain <= 4 % elsewhere.
  const bool velocity = &H == &c->hA;
  const char *gap_env = getenv(velocity ? "CFDH_AGG_GAP_A" : "CFDH_AGG_GAP_L");
  if (!gap_env) gap_env = getenv("CFDH_AGG_GAP");
  const int gap = gap_env ? atoi(gap_env) : (velocity ? 4 : 0);
  CsrDev A;
  move_csr(A, A0);
  AmgLevel *lastL = nullptr;
  // phase timings (verbose): the stream is synchronised at every tick, so they are only taken when asked for
  double tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tq = now_ms();
  const bool timing = c->opt.verbose > 0;
#define TICK(i) do { if (timing) { (void)hipStreamSynchronize(s); const double n_ = now_ms(); tm[i] += n_ - tq; tq = n_; } } while (0)
  for (;;) {
    AmgLevel *L = new AmgLevel();
    H.lev.push_back(L);
    double lm = 1.0;
    TICK(9);
    CHK(B.level_quantities(*L, A, o.amg_smooth_ratio, &lm));
    CHK(B.work_vectors(*L, ncol));
    TICK(0);
    L->fine = A.nnz <= (c->dim == 3 ? 20ll : 12ll) * A.n && A.n >= 16384;
    L->sell = L->fine && (A.nnz <= 12ll * A.n || ncol == 1);
    const bool last = A.n <= o.amg_max_coarse || (int)H.lev.size() >= maxlev;
    int na = 0;
    dbuf<int> agg;
    // P2 elements: the first coarse level is the P1 subspace (p-multigrid step) with its exact interpolation, not an aggregation
    static const bool no_pmg = getenv("CFDH_NO_PMG") && getenv("CFDH_NO_PMG")[0] == '1';
    const bool pmg = !last && !no_pmg && H.lev.size() == 1 && c->etype == 1 && c->gen_P1.n == A.n && c->gen_P1.m > 0;
    if (!last && !pmg) {
      if (host_agg) {
        CsrHost Ah;
        Ah.n = Ah.m = A.n;
        Ah.rowptr.resize((size_t)A.n + 1); Ah.col.resize(A.nnz); Ah.val.resize(A.nnz);
        HIPCHK(c, hipMemcpyAsync(Ah.rowptr.data(), A.rowptr.p, sizeof(int) * ((size_t)A.n + 1), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(Ah.col.data(), A.col.p, sizeof(int) * (size_t)A.nnz, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(Ah.val.data(), A.val.p, sizeof(double) * (size_t)A.nnz, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        std::vector<int> ha;
        na = cfdh_aggregate_host_csr(Ah, theta, ha);
        HIPCHK(c, agg.upload(ha, s));
        HIPCHK(c, hipStreamSynchronize(s));
      } else {
        CHK(B.aggregate(A, theta, gap, agg, &na));
      }
    }
    TICK(1);
    if (pmg) na = c->gen_P1.m;
    if (last || na >= A.n || na < 1) {
      // coarsest level: keep the operator (CSR) for the closing step
      move_csr(L->A, A);
      break;
    }
    // P = (I - omega D^-1 A) P0
    const MatV Am{A.rowptr.p, A.col.p, A.val.p};
    CsrDev P, R, AP, Ac;
    if (pmg) {
      P.n = c->gen_P1.n; P.m = c->gen_P1.m; P.nnz = c->gen_P1.nnz();
      HIPCHK(c, P.rowptr.upload(c->gen_P1.rowptr, s)); HIPCHK(c, P.col.upload(c->gen_P1.col, s)); HIPCHK(c, P.val.upload(c->gen_P1.val, s));
      HIPCHK(c, hipStreamSynchronize(s));
    } else
    CHK((Spgemm<AOpImwDA, BOpAgg>::run(B.dv, A.n, na, AOpImwDA{Am, L->dinv.p, 4.0 / 3.0 / lm, agg.p}, BOpAgg{agg.p}, P, B.lists, B.tmp, B.cnts)));
    TICK(2);
    if (H.keep_host0 && H.lev.size() == 1) {
      // partitioned run, replicated global pressure space: the caller cuts this rank's rows of the level-0 operator and
      // prolongator out of host copies (cfdh_solver.cpp, distributed finest level)
      auto down = [&](const CsrDev &D, CsrHost &Hc) -> int {
        Hc.n = D.n; Hc.m = D.m;
        Hc.rowptr.resize((size_t)D.n + 1); Hc.col.resize((size_t)D.nnz); Hc.val.resize((size_t)D.nnz);
        HIPCHK(c, hipMemcpyAsync(Hc.rowptr.data(), D.rowptr.p, sizeof(int) * ((size_t)D.n + 1), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(Hc.col.data(), D.col.p, sizeof(int) * (size_t)D.nnz, hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipMemcpyAsync(Hc.val.data(), D.val.p, sizeof(double) * (size_t)D.nnz, hipMemcpyDeviceToHost, s));
        return 0;
      };
      CHK(down(A, H.h_A0)); CHK(down(P, H.h_P0));
      H.h_wdinv0.resize((size_t)A.n);
      HIPCHK(c, hipMemcpyAsync(H.h_wdinv0.data(), L->wdinv.p, sizeof(double) * (size_t)A.n, hipMemcpyDeviceToHost, s));
      HIPCHK(c, hipStreamSynchronize(s));
    }
    CHK(B.transpose(P, R));
    TICK(3);
    const MatV Pm{P.rowptr.p, P.col.p, P.val.p}, Rm{R.rowptr.p, R.col.p, R.val.p};
    CHK((Spgemm<AOpPlain, BOpPlain>::run(B.dv, A.n, na, AOpPlain{Am}, BOpPlain{Pm}, AP, B.lists, B.tmp, B.cnts)));
    TICK(4);
    // G = R (I - A W)
    CHK((Spgemm<AOpPlain, BOpImAW>::run(B.dv, na, A.n, AOpPlain{Rm}, BOpImAW{Am, L->wdinv.p}, L->G, B.lists, B.tmp, B.cnts)));
    TICK(5);
    // Sb = 2W - W A W (pattern of A)
    L->Sb.n = A.n; L->Sb.m = A.n; L->Sb.nnz = A.nnz;
    HIPCHK(c, L->Sb.rowptr.alloc((size_t)A.n + 1)); HIPCHK(c, L->Sb.col.alloc(A.nnz)); HIPCHK(c, L->Sb.val.alloc(A.nnz));
    HIPCHK(c, hipMemcpyAsync(L->Sb.rowptr.p, A.rowptr.p, sizeof(int) * ((size_t)A.n + 1), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(L->Sb.col.p, A.col.p, sizeof(int) * (size_t)A.nnz, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sb_kernel, dim3((unsigned)((8ll * A.n + TPB - 1) / TPB)), dim3(TPB), 0, s, A.n, A.rowptr.p, A.col.p, A.val.p, L->wdinv.p, L->Sb.val.p);
    // Sc = P - W (A P) (pattern of A P)
    L->Sc.n = A.n; L->Sc.m = na; L->Sc.nnz = AP.nnz;
    HIPCHK(c, L->Sc.rowptr.alloc((size_t)A.n + 1)); HIPCHK(c, L->Sc.col.alloc(std::max(AP.nnz, 1))); HIPCHK(c, L->Sc.val.alloc(std::max(AP.nnz, 1)));
    HIPCHK(c, hipMemcpyAsync(L->Sc.rowptr.p, AP.rowptr.p, sizeof(int) * ((size_t)A.n + 1), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(L->Sc.col.p, AP.col.p, sizeof(int) * (size_t)AP.nnz, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sc_kernel, dim3((A.n + TPB - 1) / TPB), dim3(TPB), 0, s, A.n, AP.rowptr.p, AP.col.p, AP.val.p, P.rowptr.p, P.col.p, P.val.p,
                       L->wdinv.p, L->Sc.val.p);
    HIPCHK(c, hipGetLastError());
    TICK(6);
    // A_c = R (A P)
    const MatV APm{AP.rowptr.p, AP.col.p, AP.val.p};
    CHK((Spgemm<AOpPlain, BOpPlain>::run(B.dv, na, na, AOpPlain{Rm}, BOpPlain{APm}, Ac, B.lists, B.tmp, B.cnts)));
    TICK(7);
    // formats of the cycle kernels
    if (L->fine) CHK(B.formats(L->G, CFDH_UP_CSRF));
    if (L->sell) { CHK(B.formats(L->Sb, CFDH_UP_SELL)); CHK(B.formats(L->Sc, CFDH_UP_SELL)); }
    if (H.lev.size() == 1) { H.nnz_G0 = L->G.nnz; H.nnz_S0 = (long long)L->Sb.nnz + L->Sc.nnz; }
    // a partitioned run also uses the sweep-by-sweep cycle (distributed finest pressure level, overlapping velocity block): it
    // needs the plain transfer operators
    if (c->nranks > 1) { move_csr(L->P, P); move_csr(L->R, R); }
    move_csr(L->A, A);
    lastL = L;
    move_csr(A, Ac);
    HIPCHK(c, hipStreamSynchronize(s));  // P, R, AP are released here
    TICK(8);
  }
  tq = now_ms();
  // ---- closing step
  AmgLevel *Lc = H.lev.back();
  const int n = Lc->n;
  H.coarse_n = 0;
  if (n > 2500) {
    // coarsening stalled on a still-large, near-diagonal level: two damped-Jacobi sweeps from a zero guess, x = (2W - W A W) b
    CsrDev &A = Lc->A;
    Lc->Sb.n = n; Lc->Sb.m = n; Lc->Sb.nnz = A.nnz;
    HIPCHK(c, Lc->Sb.rowptr.alloc((size_t)n + 1)); HIPCHK(c, Lc->Sb.col.alloc(A.nnz)); HIPCHK(c, Lc->Sb.val.alloc(A.nnz));
    HIPCHK(c, hipMemcpyAsync(Lc->Sb.rowptr.p, A.rowptr.p, sizeof(int) * ((size_t)n + 1), hipMemcpyDeviceToDevice, s));
    HIPCHK(c, hipMemcpyAsync(Lc->Sb.col.p, A.col.p, sizeof(int) * (size_t)A.nnz, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(sb_kernel, dim3((unsigned)((8ll * n + TPB - 1) / TPB)), dim3(TPB), 0, s, n, A.rowptr.p, A.col.p, A.val.p, Lc->wdinv.p, Lc->Sb.val.p);
    if (Lc->sell) CHK(B.formats(Lc->Sb, CFDH_UP_SELL));
    H.fused = true;
  } else {
    dbuf<double> D0, D1, ad;
    dbuf<int> failf;
    HIPCHK(c, D0.alloc((size_t)n * n)); HIPCHK(c, D1.alloc((size_t)n * n)); HIPCHK(c, ad.alloc(n)); HIPCHK(c, failf.alloc(4));
    HIPCHK(c, D0.zero(s)); HIPCHK(c, failf.zero(s));
    hipLaunchKernelGGL(densify_kernel, dim3((n + TPB - 1) / TPB), dim3(TPB), 0, s, n, Lc->A.rowptr.p, Lc->A.col.p, Lc->A.val.p, D0.p, ad.p);
    if (singular) hipLaunchKernelGGL(trace_shift_kernel, dim3(64), dim3(TPB), 0, s, n, ad.p, D0.p);
    const unsigned gb = (unsigned)(((size_t)n * n + TPB - 1) / TPB);
    double *in = D0.p, *out = D1.p;
    for (int p = 0; p < n; p++) { hipLaunchKernelGGL(gj_step_kernel, dim3(gb), dim3(TPB), 0, s, n, p, in, out, failf.p); std::swap(in, out); }
    HIPCHK(c, hipGetLastError());
    int hf = 0;
    CHK(B.dv.read_int(failf.p, &hf));
    if (hf) return cfdh_fail(c, CFDH_E_STATE, "singular coarsest AMG operator (n=%d, device Gauss-Jordan)", n);
    HIPCHK(c, H.coarse_inv.alloc((size_t)n * n));
    HIPCHK(c, hipMemcpyAsync(H.coarse_inv.p, in, sizeof(double) * (size_t)n * n, hipMemcpyDeviceToDevice, s));
    H.coarse_n = n;
    if (lastL) {
      H.fused = true;
      lastL->Dn = 0;
      const long long ent = (long long)lastL->n * n;
      if (lastL->Sc.m == n && ent <= 8000000ll && !lastL->sell) {
        HIPCHK(c, lastL->D.alloc((size_t)ent));
        hipLaunchKernelGGL(fold_dense_kernel, dim3(lastL->n), dim3(TPB), 0, s, lastL->n, n, lastL->Sc.rowptr.p, lastL->Sc.col.p, lastL->Sc.val.p,
                           H.coarse_inv.p, lastL->D.p);
        HIPCHK(c, hipGetLastError());
        lastL->Dn = n;
      }
    }
    HIPCHK(c, hipStreamSynchronize(s));  // D0 / D1 are released here
  }
  H.valid = true;
  HIPCHK(c, hipStreamSynchronize(s));
  if (c->opt.verbose) {
    fprintf(stderr, "[cfdh] AMG hierarchy built on the device in %.1f ms (ncol %d%s):", now_ms() - t_begin, ncol, H.fused ? ", fused" : "");
    for (AmgLevel *l : H.lev) fprintf(stderr, " (%d, nnz %d; G %d Sb %d Sc %d D %d)", l->n, l->A.nnz, l->G.nnz, l->Sb.nnz, l->Sc.nnz, l->Dn);
    fprintf(stderr, "\n[cfdh]   ms: level quantities %.1f, aggregation %.1f, P %.1f, R = P^T %.1f, A P %.1f, G %.1f, Sb + Sc %.1f, A_c %.1f, formats %.1f, releases %.1f, "
            "coarsest level %.1f\n", tm[0], tm[1], tm[2], tm[3], tm[4], tm[5], tm[6], tm[7], tm[8], tm[9], now_ms() - tq);
  }
#undef TICK
  return 0;
}

// Single level with Jacobi/Chebyshev data (the mass-like operator H of the Cahouet-Chabard approximation): CSR + SELL-64 with
// the column-weighted copy, as cfdh_level_setup produces on the host.  A is consumed.
int cfdh_level_setup_dev(cfdh_ctx *c, AmgLevel &L, CsrDev &A, double ratio, int ncol) {
  Builder B(c);
  double lm;
  CHK(B.level_quantities(L, A, ratio, &lm));
  CHK(B.work_vectors(L, ncol));
  move_csr(L.A, A);
  CHK(B.formats(L.A, CFDH_UP_SELL, L.wdinv.p));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

bool cfdh_amg_dev_enabled(const cfdh_ctx *c) {
  const bool host_only = getenv("CFDH_AMG_HOST") && getenv("CFDH_AMG_HOST")[0] == '1';
  const char *nf = getenv("CFDH_NO_FUSED_AMG");
  return !host_only && c->opt.amg_smooth_degree == 1 && !(nf && nf[0] == '1');
}

int cfdh_proxy_dev(cfdh_ctx *c, CsrDev &out) { return vg_build(c, ProxyF{c->A00.p, c->dim, c->nvo}, out); }

// ---- the proxy on owned + ghost vertices (restricted additive Schwarz with one layer of overlap).  Ghost
// rows live on their owners: entry k of every owned vertex-graph row travels in slot (k mod W) of the vertex record of halo
// exchange (k div W), W = dim + 1.  Once per halo plan the records carry the GLOBAL column ids, from which the host derives the
// pattern of the ghost rows in local numbering (columns ascending); at every rebuild they carry the values, and two kernels
// write the CSR matrix -- nothing is downloaded, no host loop over rows.
namespace {
__global__ __launch_bounds__(TPB) void ras_pack_kernel(int nvo, int dim, int k0, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                       const double *__restrict__ A00, const int *__restrict__ gid, double *__restrict__ rec) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= nvo) return;
  for (int t = 0; t <= dim; t++) {
    const int k = vptr[i] + k0 + t;
    double v = gid ? -1.0 : 0.0;
    if (k < vptr[i + 1]) {
      if (gid) v = (double)gid[vcol[k]];
      else {
        v = 0.0;
        for (int q = 0; q < dim; q++) v += A00[(size_t)dim * dim * k + (size_t)q * (dim + 1)];
        v /= dim;
      }
    }
    rec[t < dim ? (size_t)dim * i + t : (size_t)dim * nvo + i] = v;
  }
}
__global__ __launch_bounds__(TPB) void ras_unpack_kernel(int ng, int nvo, int dim, int k0, int maxlen, const double *__restrict__ rec,
                                                         double *__restrict__ gval) {
  const int g = blockIdx.x * TPB + threadIdx.x;
  if (g >= ng) return;
  const size_t W = (size_t)dim + 1;
  for (int t = 0; t <= dim; t++)
    if (k0 + t < maxlen) gval[(size_t)g * maxlen + k0 + t] = rec[W * (size_t)nvo + W * (size_t)g + t];
}
// rows [0, nvo): the vertex graph with the proxy values, columns < nv; rows [nvo, nv): the received entries whose column is local.
// Exact zeros off the diagonal are dropped in both (rows / columns of Dirichlet dofs), as the host build does.
__device__ __forceinline__ double ras_proxy_value(const double *A00, int dim, int k) {
  double v = 0.0;
  for (int q = 0; q < dim; q++) v += A00[(size_t)dim * dim * k + (size_t)q * (dim + 1)];
  return v / dim;
}
template <bool FILL>
__global__ __launch_bounds__(TPB) void ras_rows_kernel(int nv, int nvo, int dim, int maxlen, const int *__restrict__ vptr, const int *__restrict__ vcol,
                                                       const double *__restrict__ A00, const int *__restrict__ gptr, const int *__restrict__ gcol,
                                                       const int *__restrict__ gsrc, const double *__restrict__ gval, int *__restrict__ rp,
                                                       int *__restrict__ col, double *__restrict__ val) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= nv) return;
  int p = FILL ? rp[i] : 0;
  if (i < nvo) {
    for (int k = vptr[i], e = vptr[i + 1]; k < e; k++) {
      const int w = vcol[k];
      if (w >= nv) continue;
      const double v = ras_proxy_value(A00, dim, k);
      if (v == 0.0 && w != i) continue;
      if (FILL) { col[p] = w; val[p] = v; }
      p++;
    }
  } else {
    const int g = i - nvo;
    for (int j = gptr[g], e = gptr[g + 1]; j < e; j++) {
      const int w = gcol[j];
      const double v = gval[(size_t)g * maxlen + gsrc[j]];
      if (v == 0.0 && w != i) continue;
      if (FILL) { col[p] = w; val[p] = v; }
      p++;
    }
  }
  if (!FILL) rp[i + 1] = p;
}
}  // namespace

static int ras_exchange_rows(cfdh_ctx *c, const int *gid_dev) {
  cfdh_ctx::RasPlan &R = c->rasp;
  hipStream_t s = c->stream;
  const int W = c->dim + 1;
  if (!c->pcw.p) { HIPCHK(c, c->pcw.alloc(c->NL)); }
  for (int k0 = 0; k0 < R.maxlen; k0 += W) {
    hipLaunchKernelGGL(ras_pack_kernel, dim3((c->nvo + TPB - 1) / TPB), dim3(TPB), 0, s, c->nvo, c->dim, k0, c->vptr.p, c->vcol.p, c->A00.p, gid_dev, c->pcw.p);
    HIPCHK(c, hipGetLastError());
    CHK(comm_halo(c, c->pcw.p));
    if (c->ng) hipLaunchKernelGGL(ras_unpack_kernel, dim3((c->ng + TPB - 1) / TPB), dim3(TPB), 0, s, c->ng, c->nvo, c->dim, k0, R.maxlen, c->pcw.p, R.gval.p);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, c->pcw.zero(s));
  return 0;
}

int cfdh_proxy_ras_dev(cfdh_ctx *c, CsrDev &out) {
  cfdh_ctx::RasPlan &R = c->rasp;
  hipStream_t s = c->stream;
  Dev dv(c);
  const int nv = c->nv, nvo = c->nvo, ng = c->ng;
  if ((int)c->h_gid.size() != nv || ng <= 0) return cfdh_fail(c, CFDH_E_STATE, "overlapping velocity cycle without the global vertex numbering");
  if (!R.ready) {
    // longest vertex-graph row over all ranks = number of entries every rank sends per vertex
    int ml = 0;
    for (int i = 0; i < nvo; i++) ml = std::max(ml, c->h_vptr[i + 1] - c->h_vptr[i]);
    double mld = (double)ml;
    HIPCHK(c, hipMemcpyAsync(c->red_out.p + 20, &mld, sizeof(double), hipMemcpyHostToDevice, s));
    CHK(comm_allreduce_dev(c, c->red_out.p + 20, 1, 1));
    HIPCHK(c, hipMemcpyAsync(&mld, c->red_out.p + 20, sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    R.maxlen = (int)mld;
    HIPCHK(c, R.gval.alloc((size_t)ng * R.maxlen + 1));
    dbuf<int> gid;
    HIPCHK(c, gid.upload(c->h_gid, s));
    CHK(ras_exchange_rows(c, gid.p));
    std::vector<double> hv((size_t)ng * R.maxlen);
    HIPCHK(c, hipMemcpyAsync(hv.data(), R.gval.p, sizeof(double) * hv.size(), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    std::vector<int> gptr(ng + 1, 0), gcol, gsrc;
    std::vector<std::pair<int, int>> row;
    for (int g = 0; g < ng; g++) {
      row.clear();
      for (int k = 0; k < R.maxlen; k++) {
        const double gd = hv[(size_t)g * R.maxlen + k];
        if (!(gd >= 0)) continue;
        const int loc = c->h_g2l[(int)gd];
        if (loc >= 0) row.push_back({loc, k});
      }
      std::sort(row.begin(), row.end());
      bool diag = false;
      for (auto &e : row) { gcol.push_back(e.first); gsrc.push_back(e.second); diag |= e.first == nvo + g; }
      if (!diag) return cfdh_fail(c, CFDH_E_COMM, "ghost row %d arrived without its diagonal", g);
      gptr[g + 1] = (int)gcol.size();
    }
    R.nent = (int)gcol.size();
    HIPCHK(c, R.gptr.upload(gptr, s)); HIPCHK(c, R.gcol.upload(gcol, s)); HIPCHK(c, R.gsrc.upload(gsrc, s));
    HIPCHK(c, hipStreamSynchronize(s));  // gid goes out of scope
    R.ready = true;
  }
  CHK(ras_exchange_rows(c, nullptr));
  out.n = out.m = nv;
  HIPCHK(c, out.rowptr.alloc((size_t)nv + 1));
  HIPCHK(c, hipMemsetAsync(out.rowptr.p, 0, sizeof(int), s));
  hipLaunchKernelGGL((ras_rows_kernel<false>), dim3((nv + TPB - 1) / TPB), dim3(TPB), 0, s, nv, nvo, c->dim, R.maxlen, c->vptr.p, c->vcol.p, c->A00.p, R.gptr.p,
                     R.gcol.p, R.gsrc.p, R.gval.p, out.rowptr.p, (int *)nullptr, (double *)nullptr);
  CHK(dv.scan(out.rowptr.p + 1, nv));
  int nnz = 0;
  CHK(dv.read_int(out.rowptr.p + nv, &nnz));
  out.nnz = nnz;
  HIPCHK(c, out.col.alloc((size_t)std::max(nnz, 1)));
  HIPCHK(c, out.val.alloc((size_t)std::max(nnz, 1)));
  hipLaunchKernelGGL((ras_rows_kernel<true>), dim3((nv + TPB - 1) / TPB), dim3(TPB), 0, s, nv, nvo, c->dim, R.maxlen, c->vptr.p, c->vcol.p, c->A00.p, R.gptr.p,
                     R.gcol.p, R.gsrc.p, R.gval.p, out.rowptr.p, out.col.p, out.val.p);
  HIPCHK(c, hipGetLastError());
  return 0;
}

int cfdh_cc_h_dev(cfdh_ctx *c, double alpha, double beta, CsrDev &out) {
  if (c->d_Lval.n != c->h_Lval.size() || c->d_Ml.n != c->h_Ml.size()) {
    HIPCHK(c, c->d_Lval.upload(c->h_Lval, c->stream));
    HIPCHK(c, c->d_Ml.upload(c->h_Ml, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return vg_build(c, HF{c->A11.p, c->d_Lval.p, c->d_Ml.p, c->ccPbc.p, c->vdiag.p, alpha, beta, c->nvo}, out);
}
