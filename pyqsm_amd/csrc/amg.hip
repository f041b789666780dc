// amg.hip — aggregation multigrid preconditioner for B = c*L + W_H, the inner
// operator of the contraction solve (lbc.hip, DESIGN.md "Contraction solve").
//
// B is a symmetric M-matrix (the flipped tufted Laplacian has no positive
// off-diagonal entry), cond(B) ~ 1e3..1e6. Jacobi-PCG needs 200-700 iterations
// for six digits; one V-cycle of this hierarchy as the CG preconditioner needs
// 35-50 (measured on 200 k-point forests over four contractions).
//
//   strength      j is a strong neighbour of i  <=>  |a_ij| >= theta * sqrt(a_ii a_jj)
//   aggregation   a maximal independent set of the SQUARED strength graph (MIS-2,
//                 hashed priorities, a handful of rounds) gives the roots; every
//                 other point joins the root it reaches in one or two strong steps.
//                 Points without any strong neighbour are left out of the coarse
//                 levels (their rows are diagonally dominant: the smoother alone
//                 handles them), which keeps the hierarchy shrinking to a few dozen
//                 unknowns even on clouds made of many disconnected pieces.
//   prolongation  piecewise constant; restriction = its transpose, applied as a
//                 gather over each aggregate's sorted member list (no atomics)
//   coarse matrix Galerkin P'BP, one lane per coarse row accumulating into a
//                 sorted LDS list in a fixed order: deterministic, symmetric
//   smoother      l1-Jacobi (x += (b - Bx)_i / sum_j |B_ij|), one sweep before and
//                 one after the coarse correction, fused with the residual and the
//                 prolongation: three launches per level and cycle
//   coarsest      <= kDenseMax unknowns: dense inverse, made once per hierarchy (k_gj_*; host Cholesky <= kCoarseMax)
//
// (Measured: running the levels below 4096 unknowns inside ONE workgroup with barriers, to
// save their three launches each, was 10-15 % slower at 50 k-1 M points than the launches:
// inside a graph they cost ~5 us apiece, a barrier-separated phase costs a full memory
// round trip.)
// Geometric (cell) aggregation, the first version of this file, ignored the
// connectivity and needed ~100 cycles per solve; strength-based aggregates need 35-50.
// Everything works on three right-hand sides at a time ([n,3] row-major).
#include "grid.hpp"
#include "sparse.hpp"

#include <atomic>
#include <cmath>
#include <type_traits>

namespace pyqsm {

static constexpr int kCoarseMax = 96;   // dense solve by one workgroup (inverse from the host) at or below this size
static constexpr int kDenseMax = 1024;  // dense inverse made on the device at or below this size (dense_max())
static constexpr int kGjBlock = 32;     // pivot block of the device inverse
static constexpr int kRowCap = 64;      // distinct coarse neighbours one coarse row may have
static constexpr int kMaxLevels = 24;
static constexpr double kTheta = 0.08;  // strength-of-connection threshold
static constexpr int kTailSweeps = 8;   // l1-Jacobi sweeps on a coarsest level too big for a dense solve

enum : int32_t { kUndecided = 0, kIn = 1, kOut = 2 };
static constexpr unsigned long long kInf = ~0ull;

struct AmgLevel {
  int n = 0;
  DevCsr A{nullptr, nullptr, nullptr};
  double* diag = nullptr;     // a_ii
  double* dinv = nullptr;     // 1 / sum_j |a_ij|
  int32_t* agg = nullptr;     // fine dof -> coarse dof, -1 = not represented below (absent on the last level)
  int32_t* mptr = nullptr;    // coarse dof -> its fine members (CSR over `members`, ascending)
  int32_t* members = nullptr;
  // the cycle runs in fp32 (it is a preconditioner: 1e-7 of rounding noise is nothing next to
  // the 1e-2 its CG is asked for, and its sparse passes are gather-bound: 20 bytes per entry
  // instead of 36); the hierarchy itself is built in fp64
  float* valsf = nullptr;     // A.vals as float
  float* dinvf = nullptr;
  int nnz = 0;
  // per ENTRY j of a level that has a coarser one: a_ij / l1_j and the aggregate of column j,
  // so that the pre-smoothing pass does not gather dinv[col] and the post-smoothing pass does
  // not gather agg[col] (each gather is a potential L1 miss, and misses are what these passes
  // cost); same products, same bits
  float* valsdf = nullptr;
  int32_t* aggcol = nullptr;
  // A P as its own CSR (row i: the distinct aggregates among the row's columns, entries summed).
  // The upward sweep needs A (x + P xc) with x = Dinv b from the downward sweep, i.e. A x = b - r
  // with the residual r that sweep stored: what is left is (A P) xc — 3-4 gathers per row from the
  // coarse vector (a quarter of the size) instead of two gathers per entry of A.
  int32_t *ap_ptr = nullptr, *ap_idx = nullptr;
  float* ap_val = nullptr;
  float *r = nullptr, *xa = nullptr, *xb = nullptr, *b = nullptr;  // [n,3]; xb unused on level 0
};

struct AmgHierarchy {
  std::vector<AmgLevel> lv;
  double* dense_inv = nullptr;  // [nc, nc] on the device, transposed (host Cholesky, nc <= kCoarseMax)
  int nc = 0;
  double* dense_gj = nullptr;   // [ld, ld] row-major inverse made on the device (k_gj_*), nc <= dense_max()
  int ld = 0;                   // nc rounded up to kGjBlock
};

// ---- level construction kernels ------------------------------------------------

__global__ __launch_bounds__(256) void k_diag_l1(int n, const int32_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 const double* __restrict__ vals,
                                                 double* __restrict__ diag,
                                                 double* __restrict__ dinv, double omega) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0, d = 0.0;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
    s += fabs(vals[j]);
    if (indices[j] == i) d = vals[j];
  }
  diag[i] = d;
  // omega = 0: l1-Jacobi, 1 / sum |a_ij| (for these diagonally dominant M-matrices about half of
  // 1 / a_ii); omega > 0: damped Jacobi, omega / a_ii, never above the l1 bound times two
  if (omega > 0.0 && d > 0.0)
    dinv[i] = omega / d;
  else
    dinv[i] = s > 0.0 ? 1.0 / s : 1.0;
}

// B = diag(cw) L + diag(wh) as an explicit CSR copy (same pattern as L: L stores its diagonal);
// cw is constant along every edge of L, which keeps B symmetric
__global__ __launch_bounds__(256) void k_make_b(int n, const int32_t* __restrict__ indptr,
                                                const int32_t* __restrict__ indices,
                                                const double* __restrict__ lvals,
                                                const double* __restrict__ cw,
                                                const double* __restrict__ wh,
                                                double* __restrict__ bvals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double ci = cw[i];
  for (int j = indptr[i]; j < indptr[i + 1]; ++j)
    bvals[j] = ci * lvals[j] + (indices[j] == i ? wh[i] : 0.0);
}

__device__ __forceinline__ bool strong(int i, int j, double v, const double* __restrict__ diag) {
  return j != i && v * v >= (kTheta * kTheta) * diag[i] * diag[j];
}

__device__ __forceinline__ unsigned long long priority(int i) {
  unsigned h = unsigned(i) * 0x9E3779B1u;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return ((unsigned long long)(h >> 1) << 32) | (unsigned long long)(unsigned(i) + 1u);
}

__global__ __launch_bounds__(256) void k_mis_init(int n, const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices,
                                                  const double* __restrict__ vals,
                                                  const double* __restrict__ diag,
                                                  int32_t* __restrict__ state,
                                                  unsigned long long* __restrict__ key) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool any = false;
  for (int j = indptr[i]; j < indptr[i + 1] && !any; ++j) any = strong(i, indices[j], vals[j], diag);
  state[i] = any ? kUndecided : kOut;
  key[i] = any ? priority(i) : 0ull;
}

// out[i] = max(in[i], in[j] over the strong neighbours j)
__global__ __launch_bounds__(256) void k_mis_max(int n, const int32_t* __restrict__ indptr,
                                                 const int32_t* __restrict__ indices,
                                                 const double* __restrict__ vals,
                                                 const double* __restrict__ diag,
                                                 const unsigned long long* __restrict__ in,
                                                 unsigned long long* __restrict__ out,
                                                 const int32_t* __restrict__ prev_left) {
  if (prev_left && *prev_left == 0) return;  // the selection finished in an earlier round of this batch
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long m = in[i];
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
    const int col = indices[j];
    if (strong(i, col, vals[j], diag)) {
      const unsigned long long v = in[col];
      m = v > m ? v : m;
    }
  }
  out[i] = m;
}

// An undecided point whose key is the largest within two strong steps becomes a root;
// one that sees a root within two steps is out. `left` counts the still undecided.
__global__ __launch_bounds__(256) void k_mis_update(int n, const unsigned long long* __restrict__ t2,
                                                    int32_t* __restrict__ state,
                                                    unsigned long long* __restrict__ key,
                                                    int32_t* __restrict__ left,
                                                    const int32_t* __restrict__ prev_left) {
  if (prev_left && *prev_left == 0) return;  // (block-uniform; `left` of this round stays 0)
  int i = blockIdx.x * 256 + threadIdx.x;
  bool still = false;
  if (i < n && state[i] == kUndecided) {
    const unsigned long long m = t2[i];
    if (m == key[i]) {
      state[i] = kIn;
      key[i] = kInf;
    } else if (m == kInf) {
      state[i] = kOut;
      key[i] = 0ull;
    } else {
      still = true;
    }
  }
  // one atomic per block (atomics on one address are served one at a time)
  __shared__ int wleft[4];
  const unsigned long long b = __ballot(still);
  if ((threadIdx.x & 63) == 0) wleft[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = wleft[0] + wleft[1] + wleft[2] + wleft[3];
    if (t > 0) atomicAdd(left, t);
  }
}

__global__ __launch_bounds__(256) void k_root_flags(int n, const int32_t* __restrict__ state,
                                                    int32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[i] = state[i] == kIn;
}

// pass 1: roots take their id, direct strong neighbours of a root join it (largest id wins)
__global__ __launch_bounds__(256) void k_agg1(int n, const int32_t* __restrict__ indptr,
                                              const int32_t* __restrict__ indices,
                                              const double* __restrict__ vals,
                                              const double* __restrict__ diag,
                                              const int32_t* __restrict__ state,
                                              const int32_t* __restrict__ ids,
                                              int32_t* __restrict__ agg1) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int a = -1;
  if (state[i] == kIn) {
    a = ids[i];
  } else {
    for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
      const int col = indices[j];
      if (state[col] == kIn && strong(i, col, vals[j], diag)) a = max(a, ids[col]);
    }
  }
  agg1[i] = a;
}

// pass 2: the rest joins through a strong neighbour that was placed in pass 1
__global__ __launch_bounds__(256) void k_agg2(int n, const int32_t* __restrict__ indptr,
                                              const int32_t* __restrict__ indices,
                                              const double* __restrict__ vals,
                                              const double* __restrict__ diag,
                                              const int32_t* __restrict__ agg1,
                                              int32_t* __restrict__ agg,
                                              int32_t* __restrict__ counts) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int a = agg1[i];
  if (a < 0) {
    for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
      const int col = indices[j];
      if (strong(i, col, vals[j], diag)) a = max(a, agg1[col]);
    }
  }
  agg[i] = a;
  if (a >= 0) atomicAdd(&counts[a], 1);
}

__global__ __launch_bounds__(256) void k_fill_members(int n, const int32_t* __restrict__ agg,
                                                      const int32_t* __restrict__ mptr,
                                                      int32_t* __restrict__ cursor,
                                                      int32_t* __restrict__ members) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = agg[i];
  if (a >= 0) members[mptr[a] + atomicAdd(&cursor[a], 1)] = i;
}

__global__ __launch_bounds__(256) void k_sort_members(int nc, const int32_t* __restrict__ mptr,
                                                      int32_t* __restrict__ members) {
  int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= nc) return;
  const int b = mptr[a], e = mptr[a + 1];
  for (int p = b + 1; p < e; ++p) {
    const int v = members[p];
    int q = p;
    while (q > b && members[q - 1] > v) {
      members[q] = members[q - 1];
      --q;
    }
    members[q] = v;
  }
}

// Galerkin row I = sum over members i of I, entries (i, j): (agg[j], a_ij). One lane per
// coarse row; the row is kept as a column-sorted list in LDS (slot-major, so the 64
// lanes of the block touch 64 consecutive words), members and entries are visited in
// a fixed order: the sums are reproducible and the result is exactly symmetric when A is.
__global__ __launch_bounds__(64) void k_galerkin_rows(int nc, const int32_t* __restrict__ mptr,
                                                      const int32_t* __restrict__ members,
                                                      const int32_t* __restrict__ indptr,
                                                      const int32_t* __restrict__ indices,
                                                      const double* __restrict__ vals,
                                                      const int32_t* __restrict__ agg,
                                                      int32_t* __restrict__ counts,
                                                      int32_t* __restrict__ tkeys,
                                                      double* __restrict__ tvals,
                                                      int32_t* __restrict__ overflow) {
  __shared__ int32_t sk[kRowCap][64];
  __shared__ double sv[kRowCap][64];
  const int t = threadIdx.x;
  const int I = blockIdx.x * 64 + t;
  if (I >= nc) return;
  int cnt = 0;
  bool over = false;
  for (int m = mptr[I]; m < mptr[I + 1]; ++m) {
    const int i = members[m];
    for (int e = indptr[i]; e < indptr[i + 1]; ++e) {
      const int J = agg[indices[e]];
      if (J < 0) continue;
      const double v = vals[e];
      int p = 0;
      while (p < cnt && sk[p][t] < J) ++p;
      if (p < cnt && sk[p][t] == J) {
        sv[p][t] += v;
      } else if (cnt < kRowCap) {
        for (int q = cnt; q > p; --q) {
          sk[q][t] = sk[q - 1][t];
          sv[q][t] = sv[q - 1][t];
        }
        sk[p][t] = J;
        sv[p][t] = v;
        ++cnt;
      } else {
        over = true;
      }
    }
  }
  if (over) *overflow = 1;
  counts[I] = cnt;
  for (int p = 0; p < cnt; ++p) {
    tkeys[size_t(I) * kRowCap + p] = sk[p][t];
    tvals[size_t(I) * kRowCap + p] = sv[p][t];
  }
}

__global__ __launch_bounds__(256) void k_rows_to_csr(int nc, const int32_t* __restrict__ tkeys,
                                                     const double* __restrict__ tvals,
                                                     const int32_t* __restrict__ indptr,
                                                     int32_t* __restrict__ indices,
                                                     double* __restrict__ vals) {
  int I = blockIdx.x * 256 + threadIdx.x;
  if (I >= nc) return;
  const int b = indptr[I], cnt = indptr[I + 1] - b;
  for (int p = 0; p < cnt; ++p) {
    indices[b + p] = tkeys[size_t(I) * kRowCap + p];
    vals[b + p] = tvals[size_t(I) * kRowCap + p];
  }
}

// ---- cycle kernels (fp32) ----------------------------------------------------------

// vals, 1/l1 as float
__global__ __launch_bounds__(256) void k_level_floats(int n, const int32_t* __restrict__ indptr,
                                                      const double* __restrict__ vals,
                                                      const double* __restrict__ dinv,
                                                      float* __restrict__ valsf,
                                                      float* __restrict__ dinvf) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) valsf[j] = float(vals[j]);
  dinvf[i] = float(dinv[i]);
}

__global__ __launch_bounds__(256) void k_entry_cols(int n, const int32_t* __restrict__ indptr,
                                                    const int32_t* __restrict__ indices,
                                                    const float* __restrict__ valsf,
                                                    const float* __restrict__ dinvf,
                                                    const int32_t* __restrict__ agg,
                                                    float* __restrict__ valsdf,
                                                    int32_t* __restrict__ aggcol) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
    const int col = indices[j];
    valsdf[j] = valsf[j] * dinvf[col];
    aggcol[j] = agg[col];
  }
}

// rows of A P: count the distinct aggregates of a row (entries whose column is not represented
// below, aggcol < 0, drop out), then fill; entry order = first occurrence in the row
__global__ __launch_bounds__(256) void k_ap_count(int n, const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ aggcol,
                                                  int32_t* __restrict__ cnt) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  int c = 0;
  if (i < n) {
    const int b = indptr[i], e = indptr[i + 1];
    for (int j = b; j < e; ++j) {
      const int a = aggcol[j];
      if (a < 0) continue;
      bool seen = false;
      for (int q = b; q < j; ++q) seen |= aggcol[q] == a;
      c += seen ? 0 : 1;
    }
  }
  cnt[i] = c;
}

__global__ __launch_bounds__(256) void k_ap_fill(int n, const int32_t* __restrict__ indptr,
                                                 const float* __restrict__ valsf,
                                                 const int32_t* __restrict__ aggcol,
                                                 const int32_t* __restrict__ ap_ptr,
                                                 int32_t* __restrict__ ap_idx,
                                                 float* __restrict__ ap_val) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = indptr[i], e = indptr[i + 1];
  int w = ap_ptr[i];
  for (int j = b; j < e; ++j) {
    const int a = aggcol[j];
    if (a < 0) continue;
    bool seen = false;
    for (int q = b; q < j; ++q) seen |= aggcol[q] == a;
    if (seen) continue;
    float s = 0.f;
    for (int q = j; q < e; ++q) s += aggcol[q] == a ? valsf[q] : 0.f;  // fixed order: deterministic
    ap_idx[w] = a;
    ap_val[w] = s;
    ++w;
  }
}

// The cycle's fp32 vectors hold one float4 (x, y, z, 0) per row: a neighbour's entry is ONE
// 16-byte gather instead of three 4-byte ones. These kernels are bound by the number of
// cache lines their gathers touch (the L1 takes a divergent access a line at a time), and
// three gathers at the same columns touch the same lines three times.
__device__ __forceinline__ float4 ld4(const float* v, int i) {
  return reinterpret_cast<const float4*>(v)[i];
}
__device__ __forceinline__ void st4(float* v, int i, float a, float b, float c) {
  reinterpret_cast<float4*>(v)[i] = make_float4(a, b, c, 0.f);
}

// rows of a caller-side vector: float4 for fp32, three doubles for fp64
__device__ __forceinline__ void ld_row(const float* v, int i, float& a, float& b, float& c) {
  const float4 t = ld4(v, i);
  a = t.x;
  b = t.y;
  c = t.z;
}
__device__ __forceinline__ void ld_row(const double* v, int i, double& a, double& b, double& c) {
  a = v[3 * i];
  b = v[3 * i + 1];
  c = v[3 * i + 2];
}
__device__ __forceinline__ void st_row(float* v, int i, float a, float b, float c) { st4(v, i, a, b, c); }
__device__ __forceinline__ void st_row(double* v, int i, float a, float b, float c) {
  v[3 * i] = double(a);
  v[3 * i + 1] = double(b);
  v[3 * i + 2] = double(c);
}

// A thread walks its row kRowUnroll entries at a time with the loads of a chunk issued side by
// side: one entry at a time is two dependent memory latencies per entry (index, then the
// gather), ~14 per row, and at 1 M rows there are only two waves per slot to hide them
// behind. Entries past the row's end read the row itself with weight 0 (adds +0: the sums
// keep their bits).
static constexpr int kRowUnroll = 8;

// pre-smoothing from a zero start fused with the residual:
//   x = Dinv b ;  r = b - A x
__device__ __forceinline__ void down_row(int i, const int32_t* indptr, const int32_t* indices,
                                         const float* valsd /* a_ij / l1_j */, const float* dinv,
                                         const float* b, float* x, float* r) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int e = indptr[i + 1];
  for (int j = indptr[i]; j < e; j += kRowUnroll) {
    int col[kRowUnroll];
    float v[kRowUnroll];
    float4 bc[kRowUnroll];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {  // past the row's end: the row itself, weight 0
      const bool ok = j + u < e;
      col[u] = ok ? indices[j + u] : i;
      v[u] = ok ? valsd[j + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) bc[u] = ld4(b, col[u]);
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      a0 += v[u] * bc[u].x;
      a1 += v[u] * bc[u].y;
      a2 += v[u] * bc[u].z;
    }
  }
  const float d = dinv[i];
  const float4 bi = ld4(b, i);
  st4(x, i, d * bi.x, d * bi.y, d * bi.z);
  st4(r, i, bi.x - a0, bi.y - a1, bi.z - a2);
}

__global__ __launch_bounds__(256) void k_down(int n, const int32_t* __restrict__ indptr,
                                              const int32_t* __restrict__ indices,
                                              const float* __restrict__ valsd,
                                              const float* __restrict__ dinv,
                                              const float* __restrict__ b, float* __restrict__ x,
                                              float* __restrict__ r) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) down_row(i, indptr, indices, valsd, dinv, b, x, r);
}

// rc[a] = sum of r over the members of aggregate a
__device__ __forceinline__ void restrict_row(int a, const int32_t* mptr, const int32_t* members,
                                             const float* r, float* rc) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const int e = mptr[a + 1];
  for (int m = mptr[a]; m < e; m += kRowUnroll) {
    int idx[kRowUnroll];
    float4 t[kRowUnroll];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) idx[u] = m + u < e ? members[m + u] : -1;
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) t[u] = idx[u] >= 0 ? ld4(r, idx[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      s0 += t[u].x;
      s1 += t[u].y;
      s2 += t[u].z;
    }
  }
  st4(rc, a, s0, s1, s2);
}

__global__ __launch_bounds__(256) void k_restrict(int nc, const int32_t* __restrict__ mptr,
                                                  const int32_t* __restrict__ members,
                                                  const float* __restrict__ r,
                                                  float* __restrict__ rc) {
  int a = blockIdx.x * 256 + threadIdx.x;
  if (a < nc) restrict_row(a, mptr, members, r, rc);
}

// coarse correction fused with the post-smoothing sweep:
//   y = x + P xc ;  out = y + Dinv (b - A y)
// TO = type of the result (double / float on level 0, float below); on level 0 `bd` is the
// caller's right-hand side and dot += bd . out.
template <typename TO, typename TBD>
__global__ __launch_bounds__(256) void k_up(int n, const int32_t* __restrict__ indptr,
                                            const int32_t* __restrict__ indices,
                                            const float* __restrict__ vals,
                                            const float* __restrict__ dinv,
                                            const int32_t* __restrict__ agg,
                                            const int32_t* __restrict__ aggcol /* agg of every entry's column */,
                                            const float* __restrict__ xc,
                                            const float* __restrict__ b,
                                            const float* __restrict__ x, TO* __restrict__ out,
                                            const TBD* __restrict__ bd /*may be null*/,
                                            double* __restrict__ dot /*[3][kPart], may be null*/) {
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  // grid-stride: with a dot product the launch is reduce_grid (one partial slot per block)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const int e = indptr[i + 1];
    for (int j = indptr[i]; j < e; j += kRowUnroll) {
      int col[kRowUnroll], ac[kRowUnroll];
      float v[kRowUnroll];
      float4 xv[kRowUnroll], c4[kRowUnroll];
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {  // past the row's end: the row itself, weight 0
        const bool ok = j + u < e;
        col[u] = ok ? indices[j + u] : i;
        v[u] = ok ? vals[j + u] : 0.f;
        ac[u] = ok ? aggcol[j + u] : -1;
      }
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) xv[u] = ld4(x, col[u]);
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) c4[u] = ac[u] >= 0 ? ld4(xc, ac[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < kRowUnroll; ++u) {
        float y0 = xv[u].x, y1 = xv[u].y, y2 = xv[u].z;
        if (ac[u] >= 0) {
          y0 += c4[u].x;
          y1 += c4[u].y;
          y2 += c4[u].z;
        }
        a0 += v[u] * y0;
        a1 += v[u] * y1;
        a2 += v[u] * y2;
      }
    }
    const int ai = agg[i];
    const float4 xi = ld4(x, i);
    float y0 = xi.x, y1 = xi.y, y2 = xi.z;
    if (ai >= 0) {
      const float4 c4 = ld4(xc, ai);
      y0 += c4.x;
      y1 += c4.y;
      y2 += c4.z;
    }
    const float d = dinv[i];
    const float4 bi = ld4(b, i);
    const float o0 = y0 + d * (bi.x - a0), o1 = y1 + d * (bi.y - a1), o2 = y2 + d * (bi.z - a2);
    st_row(out, i, o0, o1, o2);
    if (bd) {
      TBD e0, e1, e2;
      ld_row(bd, i, e0, e1, e2);
      d0 += double(e0) * double(o0);
      d1 += double(e1) * double(o1);
      d2 += double(e2) * double(o2);
    }
  }
  if (dot) reduce3_part(d0, d1, d2, dot);
}

// The same upward step through A P (see AmgLevel::ap_ptr): with x = Dinv b and r = b - A x from
// the downward sweep,  y = x + P xc,  b - A y = r - (A P) xc,  out = y + Dinv (r - (A P) xc).
__device__ __forceinline__ void up_ap_row(int i, const int32_t* ap_ptr, const int32_t* ap_idx,
                                          const float* ap_val, const float* dinv, const int32_t* agg,
                                          const float* xc, const float* r, const float* x, float& o0,
                                          float& o1, float& o2) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int e = ap_ptr[i + 1];
  for (int j = ap_ptr[i]; j < e; j += 4) {
    int ac[4];
    float v[4];
    float4 c4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // past the row's end: coarse row 0 with weight 0
      const bool ok = j + u < e;
      ac[u] = ok ? ap_idx[j + u] : 0;
      v[u] = ok ? ap_val[j + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) c4[u] = ld4(xc, ac[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 += v[u] * c4[u].x;
      a1 += v[u] * c4[u].y;
      a2 += v[u] * c4[u].z;
    }
  }
  const int ai = agg[i];
  const float4 xi = ld4(x, i);
  float y0 = xi.x, y1 = xi.y, y2 = xi.z;
  if (ai >= 0) {
    const float4 c4 = ld4(xc, ai);
    y0 += c4.x;
    y1 += c4.y;
    y2 += c4.z;
  }
  const float d = dinv[i];
  const float4 ri = ld4(r, i);
  o0 = y0 + d * (ri.x - a0);
  o1 = y1 + d * (ri.y - a1);
  o2 = y2 + d * (ri.z - a2);
}

template <typename TO, typename TBD>
__global__ __launch_bounds__(256) void k_up_ap(int n, const int32_t* __restrict__ ap_ptr,
                                               const int32_t* __restrict__ ap_idx,
                                               const float* __restrict__ ap_val,
                                               const float* __restrict__ dinv,
                                               const int32_t* __restrict__ agg,
                                               const float* __restrict__ xc,
                                               const float* __restrict__ r,
                                               const float* __restrict__ x, TO* __restrict__ out,
                                               const TBD* __restrict__ bd /*may be null*/,
                                               double* __restrict__ dot /*[3][kPart], may be null*/) {
  double d0 = 0.0, d1 = 0.0, d2 = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    float o0, o1, o2;
    up_ap_row(i, ap_ptr, ap_idx, ap_val, dinv, agg, xc, r, x, o0, o1, o2);
    st_row(out, i, o0, o1, o2);
    if (bd) {
      TBD e0, e1, e2;
      ld_row(bd, i, e0, e1, e2);
      d0 += double(e0) * double(o0);
      d1 += double(e1) * double(o1);
      d2 += double(e2) * double(o2);
    }
  }
  if (dot) reduce3_part(d0, d1, d2, dot);
}

// one l1-Jacobi sweep out = x + Dinv (b - A x)   (coarsest level without a dense inverse)
__device__ __forceinline__ void sweep_row(int i, const int32_t* indptr, const int32_t* indices,
                                          const float* vals, const float* dinv, const float* b,
                                          const float* x, float* out) {
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int e = indptr[i + 1];
  for (int j = indptr[i]; j < e; j += kRowUnroll) {
    int col[kRowUnroll];
    float v[kRowUnroll];
    float4 xv[kRowUnroll];
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      const bool ok = j + u < e;
      col[u] = ok ? indices[j + u] : i;
      v[u] = ok ? vals[j + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) xv[u] = ld4(x, col[u]);
#pragma unroll
    for (int u = 0; u < kRowUnroll; ++u) {
      a0 += v[u] * xv[u].x;
      a1 += v[u] * xv[u].y;
      a2 += v[u] * xv[u].z;
    }
  }
  const float d = dinv[i];
  const float4 xi = ld4(x, i), bi = ld4(b, i);
  st4(out, i, xi.x + d * (bi.x - a0), xi.y + d * (bi.y - a1), xi.z + d * (bi.z - a2));
}

__global__ __launch_bounds__(256) void k_sweep(int n, const int32_t* __restrict__ indptr,
                                               const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals,
                                               const float* __restrict__ dinv,
                                               const float* __restrict__ b,
                                               const float* __restrict__ x, float* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) sweep_row(i, indptr, indices, vals, dinv, b, x, out);
}

// x = dinv .* b
__device__ __forceinline__ void scale_row(int i, const float* dinv, const float* b, float* x) {
  const float d = dinv[i];
  const float4 bi = ld4(b, i);
  st4(x, i, d * bi.x, d * bi.y, d * bi.z);
}

__global__ __launch_bounds__(256) void k_scale(int n, const float* __restrict__ dinv,
                                               const float* __restrict__ b, float* __restrict__ x) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) scale_row(i, dinv, b, x);
}

// The tail sweeps of a coarsest level that is too big for a dense solve, in ONE launch of one
// workgroup (round 3): x = Dinv b and kTailSweeps l1-Jacobi sweeps cost nine launches of ~6.4 us
// each (57 us of a 212 us multigrid-CG iteration on the first contractions, where the hierarchy
// stops after two coarsenings at a few thousand unknowns). Here b and the two iterates live in
// LDS, every thread keeps its rows for all sweeps (so the row's entries come out of its L1 from
// the second sweep on), and a sweep is a workgroup barrier instead of a launch. Same row
// function, same order of operations: the result has the bits of the launches.
static constexpr int kFusedTailThreads = 1024;
static constexpr int kFusedTailMaxRows = 3072;  // three float4 vectors per row in LDS: 144 KB
__global__ __launch_bounds__(kFusedTailThreads) void k_tail_sweeps(int n, int sweeps,
                                                                  const int32_t* __restrict__ indptr,
                                                                  const int32_t* __restrict__ indices,
                                                                  const float* __restrict__ vals,
                                                                  const float* __restrict__ dinv,
                                                                  const float* __restrict__ b,
                                                                  float* __restrict__ x_out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lb = lds;                               // [n] float4
  float* xa = lds + size_t(n) * kVecStride;      // [n] float4
  float* xb = xa + size_t(n) * kVecStride;       // [n] float4
  for (int i = threadIdx.x; i < n; i += kFusedTailThreads) {
    const float4 bi = ld4(b, i);
    st4(lb, i, bi.x, bi.y, bi.z);
    const float d = dinv[i];
    st4(xb, i, d * bi.x, d * bi.y, d * bi.z);    // scale_row
  }
  __syncthreads();
  for (int s = 0; s < sweeps; s += 2) {
    for (int i = threadIdx.x; i < n; i += kFusedTailThreads) sweep_row(i, indptr, indices, vals, dinv, lb, xb, xa);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += kFusedTailThreads) sweep_row(i, indptr, indices, vals, dinv, lb, xa, xb);
    __syncthreads();
  }
  for (int i = threadIdx.x; i < n; i += kFusedTailThreads) {
    const float4 v = ld4(xb, i);
    st4(x_out, i, v.x, v.y, v.z);
  }
}

// x = Ainv * b on the coarsest level (nc <= kCoarseMax) by the first 3 * nc threads of the block
// (every thread of the block calls it): thread (i, k) adds up row i for column k in four
// interleaved chains; the inverse is stored transposed, so that a wave's loads are contiguous
// (one thread per output with a 96-long dependent chain over a row-major inverse: 7.6 us as a
// kernel of its own, 6.3 us this way — most of it is the launch). The inverse stays fp64.
__device__ __forceinline__ void dense_solve_block(int nc, const double* ainv_t, const float* b,
                                                  float* x) {
  __shared__ double sb[3][kCoarseMax];
  const int t = threadIdx.x;
  const int i = t % nc, k = t / nc;
  if (t < 3 * nc) sb[k][i] = double(b[kVecStride * i + k]);
  __syncthreads();
  if (t >= 3 * nc) return;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int j = 0;
  for (; j + 4 <= nc; j += 4) {
    s0 += ainv_t[size_t(j) * nc + i] * sb[k][j];
    s1 += ainv_t[size_t(j + 1) * nc + i] * sb[k][j + 1];
    s2 += ainv_t[size_t(j + 2) * nc + i] * sb[k][j + 2];
    s3 += ainv_t[size_t(j + 3) * nc + i] * sb[k][j + 3];
  }
  for (; j < nc; ++j) s0 += ainv_t[size_t(j) * nc + i] * sb[k][j];
  x[kVecStride * i + k] = float((s0 + s1) + (s2 + s3));
  if (k == 0) x[kVecStride * i + 3] = 0.f;
}

__global__ __launch_bounds__(3 * kCoarseMax) void k_dense_solve(int nc, const double* __restrict__ ainv_t,
                                                                const float* __restrict__ b,
                                                                float* __restrict__ x) {
  dense_solve_block(nc, ainv_t, b, x);
}

// ---- dense inverse on the device (round 3) ---------------------------------------------------
// A hierarchy used to go down to <= 96 unknowns, where a host-made Cholesky inverse finishes the
// cycle. The small levels cost what a large one costs — three launches, each a chain of dependent
// memory trips, ~4.7 us — so the levels of ~1 000, ~260 and ~90 rows were 9 of the ~24 launches of
// a cycle and a quarter of its time. Now the coarsening stops at <= kDenseMax rows and that level is
// solved exactly by ONE matrix-vector product with its inverse, which is made here once per
// hierarchy: block Gauss-Jordan in place without pivoting (the matrix is symmetric positive
// definite: W_H > 0 on the diagonal of an M-matrix), 32 columns at a time, three launches a block.
//   P = A[K,K]^-1 ;  U = P A[K,R] ;  C = A[R,K] ;  A[R,R] -= C U ;  A[K,R] = U ;  A[R,K] = -C P ;  A[K,K] = P
// The matrix is padded with an identity to a multiple of 32 rows (ld), so every block is full.

__global__ __launch_bounds__(256) void k_dense_from_csr(int n, int ld, const int32_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const double* __restrict__ vals,
                                                        double* __restrict__ A /* zeroed */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= ld) return;
  if (i >= n) {
    A[size_t(i) * ld + i] = 1.0;
    return;
  }
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) A[size_t(i) * ld + indices[j]] += vals[j];
}

// P = A[K,K]^-1 by Gauss-Jordan in LDS, one thread per element; flag[0] = 1 on a pivot that is not positive
__global__ __launch_bounds__(kGjBlock * kGjBlock) void k_gj_pivot(int ld, int k0, const double* __restrict__ A,
                                                                  double* __restrict__ P, int32_t* __restrict__ flag) {
  __shared__ double M[kGjBlock][kGjBlock + 1];
  const int r = threadIdx.x / kGjBlock, cc = threadIdx.x % kGjBlock;
  M[r][cc] = A[size_t(k0 + r) * ld + k0 + cc];
  __syncthreads();
  for (int p = 0; p < kGjBlock; ++p) {
    const double piv = M[p][p], f = M[r][p], pc = M[p][cc];
    __syncthreads();
    if (!(piv > 0.0) && threadIdx.x == 0) flag[0] = 1;
    const double inv = 1.0 / piv;
    if (r == p)
      M[r][cc] = cc == p ? inv : pc * inv;
    else
      M[r][cc] = cc == p ? -(f * inv) : M[r][cc] - f * (pc * inv);
    __syncthreads();
  }
  P[r * kGjBlock + cc] = M[r][cc];
}

// U[t][j] = sum_s P[t][s] A[k0+s][j]  and  C[i][t] = A[i][k0+t]  (thread = column j = row i)
__global__ __launch_bounds__(256) void k_gj_panels(int ld, int k0, const double* __restrict__ A,
                                                   const double* __restrict__ P, double* __restrict__ U,
                                                   double* __restrict__ C) {
  __shared__ double sp[kGjBlock * kGjBlock];
  for (int t = threadIdx.x; t < kGjBlock * kGjBlock; t += 256) sp[t] = P[t];
  __syncthreads();
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= ld) return;
  double a[kGjBlock];
#pragma unroll
  for (int sidx = 0; sidx < kGjBlock; ++sidx) a[sidx] = A[size_t(k0 + sidx) * ld + j];
  for (int t = 0; t < kGjBlock; ++t) {
    double acc = 0.0;
#pragma unroll
    for (int sidx = 0; sidx < kGjBlock; ++sidx) acc += sp[t * kGjBlock + sidx] * a[sidx];
    U[size_t(t) * ld + j] = acc;
  }
#pragma unroll
  for (int t = 0; t < kGjBlock; ++t) C[size_t(j) * kGjBlock + t] = A[size_t(j) * ld + k0 + t];
}

// the update of one 32 x 32 tile (256 threads, four elements each)
__global__ __launch_bounds__(256) void k_gj_update(int ld, int k0, double* __restrict__ A,
                                                   const double* __restrict__ P, const double* __restrict__ U,
                                                   const double* __restrict__ C) {
  __shared__ double sc[kGjBlock][kGjBlock + 1];  // C rows of the tile, or P
  __shared__ double su[kGjBlock][kGjBlock + 1];  // U columns of the tile, or P
  const int i0 = blockIdx.y * kGjBlock, j0 = blockIdx.x * kGjBlock;
  const bool row_k = i0 == k0, col_k = j0 == k0;
  for (int t = threadIdx.x; t < kGjBlock * kGjBlock; t += 256) {
    const int r = t / kGjBlock, q = t % kGjBlock;
    sc[r][q] = row_k ? 0.0 : C[size_t(i0 + r) * kGjBlock + q];                     // C[i][t]
    su[r][q] = col_k ? P[r * kGjBlock + q] : U[size_t(r) * ld + j0 + q];           // U[t][j] or P[t][j]
  }
  __syncthreads();
  for (int t = threadIdx.x; t < kGjBlock * kGjBlock; t += 256) {
    const int r = t / kGjBlock, q = t % kGjBlock;
    double* out = A + size_t(i0 + r) * ld + j0 + q;
    if (row_k) {
      *out = su[r][q];  // U[i - k0][j], or P[i - k0][j - k0] in the pivot tile
    } else {
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < kGjBlock; ++m) acc += sc[r][m] * su[m][q];
      *out = col_k ? -acc : *out - acc;
    }
  }
}

// x = Ainv b on the coarsest level, a wave per row (b staged in LDS, fp64 accumulation, the lanes'
// partial sums folded in a fixed butterfly)
__global__ __launch_bounds__(256) void k_dense_mv(int n, int ld, const double* __restrict__ ainv,
                                                  const float* __restrict__ b, float* __restrict__ x) {
  __shared__ float4 sb[kDenseMax];
  for (int j = threadIdx.x; j < n; j += 256) sb[j] = ld4(b, j);
  __syncthreads();
  const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;  // whole waves
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  const double* row = ainv + size_t(i) * ld;
  for (int j = lane; j < n; j += 64) {
    const double a = row[j];
    const float4 v = sb[j];
    s0 += a * double(v.x);
    s1 += a * double(v.y);
    s2 += a * double(v.z);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s0 += __shfl_xor(s0, off, 64);
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  if (lane == 0) st4(x, i, float(s0), float(s1), float(s2));
}

// (Round 2 experiment, dropped: the levels of <= 4096 / 2048 / 1024 / 512 rows down to the dense
// solve run by ONE 1024-thread workgroup, level after level with workgroup barriers instead of a
// launch per level and pass — same row functions, bit-identical results, and no faster: 1933-1972
// ms of multigrid CG per 20 contractions of the 1 M-point forest against 1955 ms with separate
// launches. Back-to-back launches on one stream already overlap their launch cost; what a small
// level costs is its chain of dependent memory round trips, and those are the same inside one
// workgroup.)

// fp64 [n,3] -> the cycle's fp32 rows
__global__ __launch_bounds__(256) void k_rows_to_float(int n, const double* __restrict__ a,
                                                       float* __restrict__ out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) st4(out, i, float(a[3 * i]), float(a[3 * i + 1]), float(a[3 * i + 2]));
}

// ---- host side ------------------------------------------------------------------------

static int alloc_vectors(Ctx* c, AmgLevel& L, bool coarse, int nnz) {
  PQ_TRY(c->arena.get(size_t(L.n) * kVecStride, &L.r));
  PQ_TRY(c->arena.get(size_t(L.n) * kVecStride, &L.xa));
  PQ_TRY(c->arena.get(size_t(L.n) * kVecStride, &L.b));  // level 0: the fp32 copy of the right-hand side
  if (coarse) PQ_TRY(c->arena.get(size_t(L.n) * kVecStride, &L.xb));
  PQ_TRY(c->arena.get(size_t(nnz) + 1, &L.valsf));
  PQ_TRY(c->arena.get(size_t(L.n), &L.dinvf));
  hipLaunchKernelGGL(k_level_floats, dim3(ceil_div(L.n, 256)), dim3(256), 0, c->stream, L.n, L.A.indptr,
                     L.A.vals, L.dinv, L.valsf, L.dinvf);
  PQ_HIP(hipGetLastError());
  return 0;
}

// Dense inverse of the coarsest matrix by Cholesky on the host (nc <= kCoarseMax).
static int coarse_inverse(Ctx* c, const AmgLevel& L, double** d_inv) {
  const int n = L.n;
  std::vector<int32_t> ip(size_t(n) + 1);
  PQ_HIP(hipMemcpyAsync(ip.data(), L.A.indptr, (size_t(n) + 1) * 4, hipMemcpyDeviceToHost,
                        c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  const int nnz = ip[size_t(n)];
  std::vector<int32_t> ix(size_t(nnz) + 1);
  std::vector<double> va(size_t(nnz) + 1);
  PQ_HIP(hipMemcpyAsync(ix.data(), L.A.indices, size_t(nnz) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(va.data(), L.A.vals, size_t(nnz) * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  std::vector<double> A(size_t(n) * n, 0.0), G(size_t(n) * n, 0.0), inv(size_t(n) * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = ip[size_t(i)]; j < ip[size_t(i) + 1]; ++j) A[size_t(i) * n + ix[size_t(j)]] += va[size_t(j)];
  // Cholesky A = G G'
  for (int j = 0; j < n; ++j) {
    double d = A[size_t(j) * n + j];
    for (int k = 0; k < j; ++k) d -= G[size_t(j) * n + k] * G[size_t(j) * n + k];
    if (!(d > 0.0)) return fail(PYQSM_EINVAL, "multigrid: coarsest matrix is not positive definite");
    const double g = std::sqrt(d);
    G[size_t(j) * n + j] = g;
    for (int i = j + 1; i < n; ++i) {
      double s = A[size_t(i) * n + j];
      for (int k = 0; k < j; ++k) s -= G[size_t(i) * n + k] * G[size_t(j) * n + k];
      G[size_t(i) * n + j] = s / g;
    }
  }
  // inverse column by column: solve G G' x = e_col
  std::vector<double> y(static_cast<size_t>(n), 0.0);
  for (int col = 0; col < n; ++col) {
    for (int i = 0; i < n; ++i) {
      double s = i == col ? 1.0 : 0.0;
      for (int k = 0; k < i; ++k) s -= G[size_t(i) * n + k] * y[size_t(k)];
      y[size_t(i)] = s / G[size_t(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = y[size_t(i)];
      for (int k = i + 1; k < n; ++k) s -= G[size_t(k) * n + i] * inv[size_t(k) * n + col];
      inv[size_t(i) * n + col] = s / G[size_t(i) * n + i];
    }
  }
  // stored transposed: k_dense_solve reads element (i, j) at [j * n + i], consecutive threads
  // consecutive i
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) A[size_t(j) * n + i] = inv[size_t(i) * n + j];
  PQ_TRY(c->arena.get(size_t(n) * n, d_inv));
  PQ_HIP(hipMemcpyAsync(*d_inv, A.data(), size_t(n) * n * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

static int dense_max() {  // PYQSM_AMG_DENSE_MAX: rows of the level that is solved by a dense inverse (<= kDenseMax)
  const char* e = getenv("PYQSM_AMG_DENSE_MAX");
  const int q = e ? atoi(e) : kDenseMax;
  return std::max(kCoarseMax, std::min(q, kDenseMax));
}

// Dense inverse of the coarsest matrix on the device (kCoarseMax < n <= kDenseMax): see k_gj_*.
static int coarse_inverse_device(Ctx* c, const AmgLevel& L, double** d_inv, int* ld_out, int32_t* d_flag) {
  const int n = L.n, ld = ceil_div(n, kGjBlock) * kGjBlock;
  double *A, *P, *U, *C;
  PQ_TRY(c->arena.get(size_t(ld) * ld, &A));
  PQ_TRY(c->arena.get(size_t(kGjBlock) * kGjBlock, &P));
  PQ_TRY(c->arena.get(size_t(kGjBlock) * ld, &U));
  PQ_TRY(c->arena.get(size_t(ld) * kGjBlock, &C));
  PQ_HIP(hipMemsetAsync(A, 0, size_t(ld) * ld * 8, c->stream));
  PQ_HIP(hipMemsetAsync(d_flag, 0, 4, c->stream));
  hipLaunchKernelGGL(k_dense_from_csr, dim3(ceil_div(ld, 256)), dim3(256), 0, c->stream, n, ld, L.A.indptr,
                     L.A.indices, L.A.vals, A);
  const dim3 tiles(ld / kGjBlock, ld / kGjBlock);
  for (int k0 = 0; k0 < ld; k0 += kGjBlock) {
    hipLaunchKernelGGL(k_gj_pivot, dim3(1), dim3(kGjBlock * kGjBlock), 0, c->stream, ld, k0, A, P, d_flag);
    hipLaunchKernelGGL(k_gj_panels, dim3(ceil_div(ld, 256)), dim3(256), 0, c->stream, ld, k0, A, P, U, C);
    hipLaunchKernelGGL(k_gj_update, tiles, dim3(256), 0, c->stream, ld, k0, A, P, U, C);
  }
  PQ_HIP(hipGetLastError());
  int32_t bad = 0;
  PQ_HIP(hipMemcpyAsync(&bad, d_flag, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (bad) return fail(PYQSM_EINVAL, "multigrid: coarsest matrix is not positive definite");
  *d_inv = A;
  *ld_out = ld;
  return 0;
}

// Aggregates of level F (strength graph -> MIS-2 roots -> two joining passes -> member
// lists). Returns the number of aggregates in *nc_out (0: nothing to coarsen).
static int aggregate(Ctx* c, AmgLevel& F, int32_t* d_counter, int* nc_out) {
  const int n = F.n;
  const dim3 g(ceil_div(n, 256)), blk(256);
  int32_t *state, *flags, *agg1;
  unsigned long long *key, *t1, *t2;
  PQ_TRY(c->arena.get(size_t(n), &state));
  PQ_TRY(c->arena.get(size_t(n) + 1, &flags));
  PQ_TRY(c->arena.get(size_t(n), &agg1));
  PQ_TRY(c->arena.get(size_t(n), &key));
  PQ_TRY(c->arena.get(size_t(n), &t1));
  PQ_TRY(c->arena.get(size_t(n), &t2));
  hipLaunchKernelGGL(k_mis_init, g, blk, 0, c->stream, n, F.A.indptr, F.A.indices, F.A.vals, F.diag,
                     state, key);
  // Rounds are queued kMisBatch at a time between two looks at the number of undecided points (a
  // look is a round trip of ~25 us, a hierarchy has ~40 rounds); each round leaves its count in
  // its own slot, and the kernels of a round whose predecessor left nobody undecided return at once.
  static constexpr int kMisRounds = 64, kMisBatch = 4;
  int32_t* left = nullptr;
  PQ_TRY(c->arena.get(size_t(kMisRounds), &left));
  PQ_HIP(hipMemsetAsync(left, 0, size_t(kMisRounds) * 4, c->stream));
  (void)d_counter;
  for (int round = 0;;) {
    const int batch_end = std::min(round + kMisBatch, kMisRounds);
    for (; round < batch_end; ++round) {
      const int32_t* prev = round > 0 ? left + (round - 1) : nullptr;
      hipLaunchKernelGGL(k_mis_max, g, blk, 0, c->stream, n, F.A.indptr, F.A.indices, F.A.vals, F.diag,
                         key, t1, prev);
      hipLaunchKernelGGL(k_mis_max, g, blk, 0, c->stream, n, F.A.indptr, F.A.indices, F.A.vals, F.diag,
                         t1, t2, prev);
      hipLaunchKernelGGL(k_mis_update, g, blk, 0, c->stream, n, t2, state, key, left + round, prev);
    }
    int32_t open = 0;
    PQ_HIP(hipMemcpyAsync(&open, left + (round - 1), 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    if (open == 0) break;
    if (round >= kMisRounds) return fail(PYQSM_EHIP, "multigrid: independent-set selection did not finish");
  }
  PQ_HIP(hipMemsetAsync(flags + n, 0, 4, c->stream));
  hipLaunchKernelGGL(k_root_flags, g, blk, 0, c->stream, n, state, flags);
  PQ_TRY(exclusive_scan_i32(c, flags, int64_t(n) + 1));
  int32_t nc = 0;
  PQ_HIP(hipMemcpyAsync(&nc, flags + n, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  *nc_out = nc;
  if (nc <= 0) return 0;
  int32_t* cursor;
  PQ_TRY(c->arena.get(size_t(n), &F.agg));
  PQ_TRY(c->arena.get(size_t(nc) + 1, &F.mptr));
  PQ_TRY(c->arena.get(size_t(nc), &cursor));
  PQ_TRY(c->arena.get(size_t(n), &F.members));
  PQ_HIP(hipMemsetAsync(F.mptr, 0, (size_t(nc) + 1) * 4, c->stream));
  PQ_HIP(hipMemsetAsync(cursor, 0, size_t(nc) * 4, c->stream));
  hipLaunchKernelGGL(k_agg1, g, blk, 0, c->stream, n, F.A.indptr, F.A.indices, F.A.vals, F.diag, state,
                     flags, agg1);
  hipLaunchKernelGGL(k_agg2, g, blk, 0, c->stream, n, F.A.indptr, F.A.indices, F.A.vals, F.diag, agg1,
                     F.agg, F.mptr);
  PQ_TRY(exclusive_scan_i32(c, F.mptr, int64_t(nc) + 1));
  hipLaunchKernelGGL(k_fill_members, g, blk, 0, c->stream, n, F.agg, F.mptr, cursor, F.members);
  hipLaunchKernelGGL(k_sort_members, dim3(ceil_div(nc, 256)), blk, 0, c->stream, nc, F.mptr,
                     F.members);
  PQ_HIP(hipGetLastError());
  return 0;
}

static double smoother_omega() {  // PYQSM_AMG_OMEGA: damped-Jacobi weight of the smoother (0 = l1-Jacobi)
  static const double v = [] {
    const char* e = getenv("PYQSM_AMG_OMEGA");
    const double w = e ? atof(e) : 0.0;
    return w > 0.0 && w < 1.0 ? w : 0.0;
  }();
  return v;
}

int amg_build(Ctx* c, const DevCsr& Lm, int n, const double* cw, const double* wh, AmgHierarchy** out) {
  *out = nullptr;
  AmgHierarchy* H = new AmgHierarchy();
  auto bail = [&](int rc) {
    delete H;
    return rc;
  };
#define AMG_TRY(expr)                 \
  do {                                \
    int r__ = (expr);                 \
    if (r__ != 0) return bail(r__);   \
  } while (0)
#define AMG_HIP(expr)                                                                  \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return bail(fail(PYQSM_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__)));  \
  } while (0)
  // ---- level 0: explicit B -------------------------------------------------------
  AmgLevel l0;
  l0.n = n;
  int32_t nnz0 = 0;
  AMG_HIP(hipMemcpyAsync(&nnz0, Lm.indptr + n, 4, hipMemcpyDeviceToHost, c->stream));
  AMG_HIP(hipStreamSynchronize(c->stream));
  l0.A.indptr = Lm.indptr;
  l0.A.indices = Lm.indices;
  AMG_TRY(c->arena.get(size_t(nnz0) + 1, &l0.A.vals));
  AMG_TRY(c->arena.get(size_t(n), &l0.diag));
  AMG_TRY(c->arena.get(size_t(n), &l0.dinv));
  hipLaunchKernelGGL(k_make_b, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, Lm.indptr,
                     Lm.indices, Lm.vals, cw, wh, l0.A.vals);
  hipLaunchKernelGGL(k_diag_l1, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, l0.A.indptr,
                     l0.A.indices, l0.A.vals, l0.diag, l0.dinv, smoother_omega());
  AMG_TRY(alloc_vectors(c, l0, false, nnz0));
  l0.nnz = nnz0;
  H->lv.push_back(l0);
  // ---- coarsen -------------------------------------------------------------------------
  int32_t* d_flag = nullptr;
  AMG_TRY(c->arena.get(2, &d_flag));
  const int dmax = dense_max();
  for (int lev = 0; lev < kMaxLevels; ++lev) {
    AmgLevel& F = H->lv.back();
    if (F.n <= dmax) break;
    int nc = 0;
    AMG_TRY(aggregate(c, F, d_flag, &nc));
    if (nc <= 0 || nc > 0.8 * F.n) {  // nothing (left) to coarsen
      F.agg = nullptr;
      break;
    }
    AmgLevel C;
    C.n = nc;
    int32_t* tkeys = nullptr;
    double* tvals = nullptr;
    AMG_TRY(c->arena.get(size_t(nc) * kRowCap, &tkeys));
    AMG_TRY(c->arena.get(size_t(nc) * kRowCap, &tvals));
    AMG_TRY(c->arena.get(size_t(nc) + 1, &C.A.indptr));
    AMG_HIP(hipMemsetAsync(C.A.indptr + nc, 0, 4, c->stream));
    AMG_HIP(hipMemsetAsync(d_flag + 1, 0, 4, c->stream));
    hipLaunchKernelGGL(k_galerkin_rows, dim3(ceil_div(nc, 64)), dim3(64), 0, c->stream, nc, F.mptr,
                       F.members, F.A.indptr, F.A.indices, F.A.vals, F.agg, C.A.indptr, tkeys, tvals,
                       d_flag + 1);
    AMG_TRY(exclusive_scan_i32(c, C.A.indptr, int64_t(nc) + 1));
    int32_t h2[2] = {0, 0};
    AMG_HIP(hipMemcpyAsync(&h2[0], C.A.indptr + nc, 4, hipMemcpyDeviceToHost, c->stream));
    AMG_HIP(hipMemcpyAsync(&h2[1], d_flag + 1, 4, hipMemcpyDeviceToHost, c->stream));
    AMG_HIP(hipStreamSynchronize(c->stream));
    if (h2[1]) {  // a coarse row has more than kRowCap neighbours: stop coarsening here
      F.agg = nullptr;
      break;
    }
    const dim3 gc(ceil_div(nc, 256)), blk(256);
    AMG_TRY(c->arena.get(size_t(h2[0]) + 1, &C.A.indices));
    AMG_TRY(c->arena.get(size_t(h2[0]) + 1, &C.A.vals));
    hipLaunchKernelGGL(k_rows_to_csr, gc, blk, 0, c->stream, nc, tkeys, tvals, C.A.indptr, C.A.indices,
                       C.A.vals);
    AMG_TRY(c->arena.get(size_t(nc), &C.diag));
    AMG_TRY(c->arena.get(size_t(nc), &C.dinv));
    hipLaunchKernelGGL(k_diag_l1, gc, blk, 0, c->stream, nc, C.A.indptr, C.A.indices, C.A.vals, C.diag,
                       C.dinv, smoother_omega());
    AMG_HIP(hipGetLastError());
    AMG_TRY(alloc_vectors(c, C, true, h2[0]));
    C.nnz = h2[0];
    H->lv.push_back(C);
  }
  AmgLevel& last = H->lv.back();
  last.agg = nullptr;
  const char* ape = getenv("PYQSM_AMG_AP");  // "0": the round-1 upward sweep over A (A/B comparisons)
  const bool use_ap = !(ape && ape[0] == '0');
  for (size_t l = 0; l + 1 < H->lv.size(); ++l) {  // per-entry columns' 1/l1 and aggregate
    AmgLevel& L = H->lv[l];
    AMG_TRY(c->arena.get(size_t(L.nnz) + 1, &L.valsdf));
    AMG_TRY(c->arena.get(size_t(L.nnz) + 1, &L.aggcol));
    hipLaunchKernelGGL(k_entry_cols, dim3(ceil_div(L.n, 256)), dim3(256), 0, c->stream, L.n, L.A.indptr,
                       L.A.indices, L.valsf, L.dinvf, L.agg, L.valsdf, L.aggcol);
    if (use_ap) {
      AMG_TRY(c->arena.get(size_t(L.n) + 1, &L.ap_ptr));
      hipLaunchKernelGGL(k_ap_count, dim3(ceil_div(L.n + 1, 256)), dim3(256), 0, c->stream, L.n, L.A.indptr,
                         L.aggcol, L.ap_ptr);
      AMG_HIP(hipGetLastError());
      AMG_TRY(exclusive_scan_i32(c, L.ap_ptr, int64_t(L.n) + 1));
      // at most one entry per entry of A: no read-back of the exact count
      AMG_TRY(c->arena.get(size_t(L.nnz) + 4, &L.ap_idx));
      AMG_TRY(c->arena.get(size_t(L.nnz) + 4, &L.ap_val));
      hipLaunchKernelGGL(k_ap_fill, dim3(ceil_div(L.n, 256)), dim3(256), 0, c->stream, L.n, L.A.indptr,
                         L.valsf, L.aggcol, L.ap_ptr, L.ap_idx, L.ap_val);
    }
  }
  AMG_HIP(hipGetLastError());
  if (last.n <= kCoarseMax && H->lv.size() > 1) {
    H->nc = last.n;
    AMG_TRY(coarse_inverse(c, last, &H->dense_inv));
  } else if (last.n <= dmax && H->lv.size() > 1) {
    H->nc = last.n;
    AMG_TRY(coarse_inverse_device(c, last, &H->dense_gj, &H->ld, d_flag));
  }
  if (getenv("PYQSM_LBC_TRACE")) {
    fprintf(stderr, "multigrid levels:");
    for (auto& l : H->lv) fprintf(stderr, " %d", l.n);
    fprintf(stderr, "\n");
  }
  *out = H;
  return 0;
#undef AMG_TRY
#undef AMG_HIP
}

void amg_destroy(AmgHierarchy* h) { delete h; }

int amg_levels(const AmgHierarchy* h) { return h ? int(h->lv.size()) : 0; }

static bool fused_tail_enabled() {  // PYQSM_AMG_FUSED_TAIL=0: a launch per tail sweep (A/B comparisons)
  static const bool v = [] {
    const char* e = getenv("PYQSM_AMG_FUSED_TAIL");
    return !(e && e[0] == '0');
  }();
  return v;
}
static int allow_fused_tail_lds(Ctx* c) {
  static std::atomic<uint64_t> attr_set{0};
  const uint64_t bit = 1ull << (c->device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tail_sweeps), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kFusedTailMaxRows * kVecStride * 4 * 3));
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  return 0;
}

// TV = double: fp64 right-hand side and result (the cycle converts the former);
// TV = float: both fp32 (the fp32 CG of lbc.hip), no conversion pass.
template <typename TV>
static int vcycle_impl(Ctx* c, AmgHierarchy* H, const TV* b, TV* x, double* dot) {
  const int nl = int(H->lv.size());
  const dim3 blk(256);
  AmgLevel& L0 = H->lv[0];
  const float* b0 = nullptr;
  if constexpr (std::is_same<TV, double>::value) {  // the cycle works on an fp32 copy
    hipLaunchKernelGGL(k_rows_to_float, dim3(ceil_div(L0.n, 256)), blk, 0, c->stream, L0.n, b, L0.b);
    b0 = L0.b;
  } else {
    b0 = b;
  }
  // downward sweep: pre-smooth, residual, restrict
  for (int l = 0; l < nl; ++l) {
    AmgLevel& L = H->lv[size_t(l)];
    const float* bl = l == 0 ? b0 : L.b;
    const dim3 g(ceil_div(L.n, 256));
    if (l == nl - 1) {  // (never level 0: hierarchies with a single level are not used)
      if (H->dense_inv) {
        hipLaunchKernelGGL(k_dense_solve, dim3(1), dim3(3 * kCoarseMax), 0, c->stream, L.n, H->dense_inv, bl, L.xb);
      } else if (H->dense_gj) {
        hipLaunchKernelGGL(k_dense_mv, dim3(ceil_div(L.n, 4)), blk, 0, c->stream, L.n, H->ld, H->dense_gj, bl, L.xb);
      } else if (L.n <= kFusedTailMaxRows && fused_tail_enabled()) {  // the same sweeps in one launch
        PQ_TRY(allow_fused_tail_lds(c));
        hipLaunchKernelGGL(k_tail_sweeps, dim3(1), dim3(kFusedTailThreads), size_t(L.n) * kVecStride * 4 * 3, c->stream,
                           L.n, kTailSweeps, L.A.indptr, L.A.indices, L.valsf, L.dinvf, bl, L.xb);
      } else {  // kTailSweeps (even) l1-Jacobi sweeps, ending in xb
        hipLaunchKernelGGL(k_scale, g, blk, 0, c->stream, L.n, L.dinvf, bl, L.xb);
        for (int s = 0; s < kTailSweeps; s += 2) {
          hipLaunchKernelGGL(k_sweep, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.valsf,
                             L.dinvf, bl, L.xb, L.xa);
          hipLaunchKernelGGL(k_sweep, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.valsf,
                             L.dinvf, bl, L.xa, L.xb);
        }
      }
      break;
    }
    AmgLevel& C = H->lv[size_t(l) + 1];
    {
      ProfScope pk(c, l == 0 ? "k_down_l0" : "k_down_coarse", 1, 2);
      hipLaunchKernelGGL(k_down, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.valsdf, L.dinvf, bl,
                         L.xa, L.r);
    }
    hipLaunchKernelGGL(k_restrict, dim3(ceil_div(C.n, 256)), blk, 0, c->stream, C.n, L.mptr, L.members,
                       L.r, C.b);
  }
  // upward sweep: coarse correction + post-smoothing
  for (int l = nl - 2; l >= 0; --l) {
    AmgLevel& L = H->lv[size_t(l)];
    AmgLevel& C = H->lv[size_t(l) + 1];
    const dim3 g(ceil_div(L.n, 256));
    ProfScope pk(c, l == 0 ? "k_up_l0" : "k_up_coarse", 1, 2);
    if (L.ap_ptr) {
      if (l == 0)
        hipLaunchKernelGGL((k_up_ap<TV, TV>), dot ? dim3(reduce_grid(L.n)) : g, blk, 0,
                           c->stream, L.n, L.ap_ptr, L.ap_idx, L.ap_val, L.dinvf, L.agg, C.xb, L.r, L.xa, x,
                           dot ? b : static_cast<const TV*>(nullptr), dot);
      else
        hipLaunchKernelGGL((k_up_ap<float, float>), g, blk, 0, c->stream, L.n, L.ap_ptr, L.ap_idx, L.ap_val,
                           L.dinvf, L.agg, C.xb, L.r, L.xa, L.xb, static_cast<const float*>(nullptr),
                           static_cast<double*>(nullptr));
      continue;
    }
    if (l == 0)
      hipLaunchKernelGGL((k_up<TV, TV>), dot ? dim3(reduce_grid(L.n)) : g, blk, 0,
                         c->stream, L.n, L.A.indptr, L.A.indices, L.valsf,
                         L.dinvf, L.agg, L.aggcol, C.xb, b0, L.xa, x, dot ? b : static_cast<const TV*>(nullptr), dot);
    else
      hipLaunchKernelGGL((k_up<float, float>), g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices,
                         L.valsf, L.dinvf, L.agg, L.aggcol, C.xb, L.b, L.xa, L.xb, static_cast<const float*>(nullptr),
                         static_cast<double*>(nullptr));
  }
  PQ_HIP(hipGetLastError());
  return 0;
}

int amg_vcycle(Ctx* c, AmgHierarchy* H, const double* b, double* x, double* dot) {
  return vcycle_impl<double>(c, H, b, x, dot);
}

int amg_vcycle_f32(Ctx* c, AmgHierarchy* H, const float* b, float* x, double* dot) {
  return vcycle_impl<float>(c, H, b, x, dot);
}

void amg_fine_matrix(const AmgHierarchy* H, const int32_t** indptr, const int32_t** indices,
                     const float** vals) {
  *indptr = H->lv[0].A.indptr;
  *indices = H->lv[0].A.indices;
  *vals = H->lv[0].valsf;
}

}  // namespace pyqsm
