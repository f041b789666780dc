// amg.hip — aggregation multigrid preconditioner for B = c*L + W_H, the inner
// operator of the contraction solve (lbc.hip, DESIGN.md "Contraction solve").
//
// cond(B) reaches 1e6 on contracted clouds, so Jacobi-PCG needs 10^4-10^5
// iterations per inner solve; one V-cycle of this hierarchy as the CG
// preconditioner brings that to a few dozen.
//
//   aggregation   points that fall in the same cell of a uniform grid (edge
//                 chosen for ~8 points per aggregate; the edge doubles per level)
//   prolongation  piecewise constant; restriction = its transpose
//   coarse matrix Galerkin P'BP, accumulated into a per-row open-addressing table
//                 with atomics, then compacted to CSR with sorted columns
//   smoother      l1-Jacobi (x += (b - Bx)_i / sum_j |B_ij|): convergent for any
//                 SPD matrix, no eigenvalue estimate needed
//   coarsest      <= kCoarseMax unknowns: dense inverse computed on the host once
//
// Everything works on three right-hand sides at a time ([n,3] row-major).
#include "grid.hpp"
#include "sparse.hpp"

#include <cmath>

namespace pyqsm {

static constexpr int kCoarseMax = 96;     // dense solve at or below this size
static constexpr int kTableCap = 128;     // distinct coarse neighbours per row (hash slots)
static constexpr int kMaxLevels = 24;
static constexpr int kEmpty = -1;

struct AmgLevel {
  int n = 0;
  DevCsr A{nullptr, nullptr, nullptr};
  double* dinv = nullptr;   // 1 / sum_j |A_ij|
  int32_t* agg = nullptr;   // fine dof -> coarse dof (absent on the coarsest level)
  int32_t* cell = nullptr;  // [n][3] integer cell coordinates at the NEXT level's edge
  double *r = nullptr, *x = nullptr, *b = nullptr;  // [n,3] work vectors (b, x unused on level 0)
};

struct AmgHierarchy {
  int sweeps = 1;        // l1-Jacobi sweeps before and after the coarse correction
  double alpha = 1.0;    // scaling of the piecewise-constant coarse correction
  std::vector<AmgLevel> lv;
  double* dense_inv = nullptr;  // [nc, nc] on the device
  int nc = 0;
};

// ---- level construction kernels ------------------------------------------------

__global__ __launch_bounds__(256) void k_l1_diag(int n, const int32_t* __restrict__ indptr,
                                                 const double* __restrict__ vals,
                                                 double* __restrict__ dinv) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) s += fabs(vals[j]);
  dinv[i] = s > 0.0 ? 1.0 / s : 1.0;
}

// B = c*L + diag(wh) as an explicit CSR copy (same pattern as L: L stores its diagonal)
__global__ __launch_bounds__(256) void k_make_b(int n, const int32_t* __restrict__ indptr,
                                                const int32_t* __restrict__ indices,
                                                const double* __restrict__ lvals, double cw,
                                                const double* __restrict__ wh,
                                                double* __restrict__ bvals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j)
    bvals[j] = cw * lvals[j] + (indices[j] == i ? wh[i] : 0.0);
}

__global__ __launch_bounds__(256) void k_point_cells(int n, const double* __restrict__ xyz,
                                                     double mx, double my, double mz, double inv,
                                                     int32_t* __restrict__ cell) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cell[3 * i] = int(floor((xyz[3 * i] - mx) * inv));
  cell[3 * i + 1] = int(floor((xyz[3 * i + 1] - my) * inv));
  cell[3 * i + 2] = int(floor((xyz[3 * i + 2] - mz) * inv));
}

__device__ __forceinline__ int64_t cell_key(const int32_t* cell, int i, int nx, int ny, int nz) {
  int cx = cell[3 * i], cy = cell[3 * i + 1], cz = cell[3 * i + 2];
  cx = cx < 0 ? 0 : (cx >= nx ? nx - 1 : cx);
  cy = cy < 0 ? 0 : (cy >= ny ? ny - 1 : cy);
  cz = cz < 0 ? 0 : (cz >= nz ? nz - 1 : cz);
  return (int64_t(cz) * ny + cy) * nx + cx;
}

__global__ __launch_bounds__(256) void k_flag_cells(int n, const int32_t* __restrict__ cell, int nx,
                                                    int ny, int nz, int32_t* __restrict__ flags) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) flags[cell_key(cell, i, nx, ny, nz)] = 1;
}

// agg[i] = compact id of i's cell; the coarse dof inherits the cell coordinates halved
__global__ __launch_bounds__(256) void k_assign_agg(int n, const int32_t* __restrict__ cell, int nx,
                                                    int ny, int nz,
                                                    const int32_t* __restrict__ ids,
                                                    int32_t* __restrict__ agg,
                                                    int32_t* __restrict__ coarse_cell) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = ids[cell_key(cell, i, nx, ny, nz)];
  agg[i] = a;
  coarse_cell[3 * a] = cell[3 * i] >> 1;      // every member writes the same values
  coarse_cell[3 * a + 1] = cell[3 * i + 1] >> 1;
  coarse_cell[3 * a + 2] = cell[3 * i + 2] >> 1;
}

__global__ __launch_bounds__(256) void k_halve_cells(int n, int32_t* __restrict__ cell) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  cell[3 * i] >>= 1;
  cell[3 * i + 1] >>= 1;
  cell[3 * i + 2] >>= 1;
}

// Galerkin product into per-row hash tables: table[I][slot] += A_ij for J = agg[j]
__global__ __launch_bounds__(256) void k_galerkin(int n, const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices,
                                                  const double* __restrict__ vals,
                                                  const int32_t* __restrict__ agg,
                                                  int32_t* __restrict__ tkeys,
                                                  double* __restrict__ tvals,
                                                  int32_t* __restrict__ overflow) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int I = agg[i];
  int32_t* keys = tkeys + size_t(I) * kTableCap;
  double* tv = tvals + size_t(I) * kTableCap;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
    const int J = agg[indices[j]];
    unsigned slot = (unsigned(J) * 2654435761u) % kTableCap;
    int probes = 0;
    for (;;) {
      const int prev = atomicCAS(&keys[slot], kEmpty, J);
      if (prev == kEmpty || prev == J) {
        atomicAdd(&tv[slot], vals[j]);
        break;
      }
      slot = (slot + 1) % kTableCap;
      if (++probes >= kTableCap) {
        *overflow = 1;
        break;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_table_counts(int nc, const int32_t* __restrict__ tkeys,
                                                      int32_t* __restrict__ counts) {
  int I = blockIdx.x * 256 + threadIdx.x;
  if (I >= nc) return;
  int c = 0;
  for (int s = 0; s < kTableCap; ++s) c += tkeys[size_t(I) * kTableCap + s] != kEmpty;
  counts[I] = c;
}

__global__ __launch_bounds__(256) void k_table_to_csr(int nc, const int32_t* __restrict__ tkeys,
                                                      const double* __restrict__ tvals,
                                                      const int32_t* __restrict__ indptr,
                                                      int32_t* __restrict__ indices,
                                                      double* __restrict__ vals) {
  int I = blockIdx.x * 256 + threadIdx.x;
  if (I >= nc) return;
  const int b = indptr[I];
  int m = 0;
  for (int s = 0; s < kTableCap; ++s) {
    const int k = tkeys[size_t(I) * kTableCap + s];
    if (k == kEmpty) continue;
    const double v = tvals[size_t(I) * kTableCap + s];
    int j = m++;
    while (j > 0 && indices[b + j - 1] > k) {  // insertion sort by column
      indices[b + j] = indices[b + j - 1];
      vals[b + j] = vals[b + j - 1];
      --j;
    }
    indices[b + j] = k;
    vals[b + j] = v;
  }
}

// ---- cycle kernels ----------------------------------------------------------------

// x = dinv .* b
__global__ __launch_bounds__(256) void k_smooth0(int n, const double* __restrict__ dinv,
                                                 const double* __restrict__ b,
                                                 double* __restrict__ x) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = dinv[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) x[3 * i + k] = d * b[3 * i + k];
}

// r = b - A x
__global__ __launch_bounds__(256) void k_residual(int n, const int32_t* __restrict__ indptr,
                                                  const int32_t* __restrict__ indices,
                                                  const double* __restrict__ vals,
                                                  const double* __restrict__ b,
                                                  const double* __restrict__ x,
                                                  double* __restrict__ r) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) {
    const int col = indices[j];
    const double v = vals[j];
    a0 += v * x[3 * col];
    a1 += v * x[3 * col + 1];
    a2 += v * x[3 * col + 2];
  }
  r[3 * i] = b[3 * i] - a0;
  r[3 * i + 1] = b[3 * i + 1] - a1;
  r[3 * i + 2] = b[3 * i + 2] - a2;
}

__global__ __launch_bounds__(256) void k_restrict(int n, const int32_t* __restrict__ agg,
                                                  const double* __restrict__ r,
                                                  double* __restrict__ rc) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = agg[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) atomicAdd(&rc[3 * a + k], r[3 * i + k]);
}

__global__ __launch_bounds__(256) void k_prolong_add(int n, const int32_t* __restrict__ agg,
                                                     const double* __restrict__ xc, double alpha,
                                                     double* __restrict__ x) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = agg[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) x[3 * i + k] += alpha * xc[3 * a + k];
}

// x += dinv .* r
__global__ __launch_bounds__(256) void k_correct(int n, const double* __restrict__ dinv,
                                                 const double* __restrict__ r,
                                                 double* __restrict__ x) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = dinv[i];
#pragma unroll
  for (int k = 0; k < 3; ++k) x[3 * i + k] += d * r[3 * i + k];
}

// x = Ainv * b on the coarsest level (nc <= kCoarseMax), one block
__global__ __launch_bounds__(128) void k_dense_solve(int nc, const double* __restrict__ ainv,
                                                     const double* __restrict__ b,
                                                     double* __restrict__ x) {
  __shared__ double sb[kCoarseMax * 3];
  for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) sb[t] = b[t];
  __syncthreads();
  for (int t = threadIdx.x; t < nc * 3; t += blockDim.x) {
    const int i = t / 3, k = t % 3;
    double s = 0.0;
    for (int j = 0; j < nc; ++j) s += ainv[size_t(i) * nc + j] * sb[3 * j + k];
    x[t] = s;
  }
}

// ---- host side ------------------------------------------------------------------------

static int alloc_vectors(Ctx* c, AmgLevel& L, bool need_bx) {
  PQ_TRY(c->arena.get(size_t(L.n) * 3, &L.r));
  if (need_bx) {
    PQ_TRY(c->arena.get(size_t(L.n) * 3, &L.x));
    PQ_TRY(c->arena.get(size_t(L.n) * 3, &L.b));
  }
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
  for (int i = 0; i < n; ++i)  // symmetrise the atomics' rounding noise
    for (int j = 0; j < i; ++j) {
      const double m = 0.5 * (A[size_t(i) * n + j] + A[size_t(j) * n + i]);
      A[size_t(i) * n + j] = A[size_t(j) * n + i] = m;
    }
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
  PQ_TRY(c->arena.get(size_t(n) * n, d_inv));
  PQ_HIP(hipMemcpyAsync(*d_inv, inv.data(), size_t(n) * n * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int amg_build(Ctx* c, const DevCsr& Lm, int n, double cw, const double* wh, const double* xyz,
              AmgHierarchy** out) {
  *out = nullptr;
  AmgHierarchy* H = new AmgHierarchy();
  if (const char* e = getenv("PYQSM_AMG_SWEEPS")) H->sweeps = std::max(1, std::min(8, atoi(e)));
  if (const char* e = getenv("PYQSM_AMG_ALPHA")) {
    const double v = atof(e);
    if (v > 0.0 && v < 4.0) H->alpha = v;
  }
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
  AMG_TRY(c->arena.get(size_t(n), &l0.dinv));
  hipLaunchKernelGGL(k_make_b, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, Lm.indptr,
                     Lm.indices, Lm.vals, cw, wh, l0.A.vals);
  hipLaunchKernelGGL(k_l1_diag, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, l0.A.indptr,
                     l0.A.vals, l0.dinv);
  AMG_TRY(alloc_vectors(c, l0, false));
  // ---- first aggregation edge: ~8 points per occupied cell ---------------------------
  double box[6];
  AMG_TRY(cloud_bbox(c, xyz, n, box, box + 3));
  double ext = std::max(box[3] - box[0], std::max(box[4] - box[1], box[5] - box[2]));
  if (!(ext > 0)) ext = 1.0;
  const double target = 8.0;
  double c1, per1, c2, per2, dim = 2.0, edge;
  AMG_TRY(probe_occupancy(c, xyz, n, box, ext / 64.0, &c1, &per1));
  if (per1 <= target) {
    edge = c1;
  } else {
    AMG_TRY(probe_occupancy(c, xyz, n, box, c1 * 0.5, &c2, &per2));
    if (c2 < c1 && per2 > 0) dim = std::log2(std::max(per1 / per2, 1.0001)) / std::log2(c1 / c2);
    dim = std::min(3.0, std::max(1.0, dim));
    edge = c2 * std::pow(target / per2, 1.0 / dim);
  }
  edge = std::max(edge, ext / 2048.0);
  AMG_TRY(c->arena.get(size_t(n) * 3, &l0.cell));
  hipLaunchKernelGGL(k_point_cells, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, xyz, box[0],
                     box[1], box[2], 1.0 / edge, l0.cell);
  int dims[3];
  for (int a = 0; a < 3; ++a) dims[a] = int(std::floor((box[3 + a] - box[a]) / edge)) + 1;
  H->lv.push_back(l0);
  // ---- coarsen -------------------------------------------------------------------------
  int32_t* d_over = nullptr;
  AMG_TRY(c->arena.get(1, &d_over));
  for (int lev = 0; lev < kMaxLevels; ++lev) {
    AmgLevel& F = H->lv.back();
    if (F.n <= kCoarseMax) break;
    const int64_t ncell = int64_t(dims[0]) * dims[1] * dims[2];
    if (ncell > (int64_t(1) << 28)) break;
    int32_t* flags = nullptr;
    AMG_TRY(c->arena.get(size_t(ncell) + 1, &flags));
    AMG_HIP(hipMemsetAsync(flags, 0, (size_t(ncell) + 1) * 4, c->stream));
    const dim3 gf(ceil_div(F.n, 256)), blk(256);
    hipLaunchKernelGGL(k_flag_cells, gf, blk, 0, c->stream, F.n, F.cell, dims[0], dims[1], dims[2],
                       flags);
    AMG_TRY(exclusive_scan_i32(c, flags, ncell + 1));
    int32_t nc = 0;
    AMG_HIP(hipMemcpyAsync(&nc, flags + ncell, 4, hipMemcpyDeviceToHost, c->stream));
    AMG_HIP(hipStreamSynchronize(c->stream));
    if (nc <= 0 || nc > 0.7 * F.n) {  // cells too fine to coarsen: double their edge and retry
      if (dims[0] == 1 && dims[1] == 1 && dims[2] == 1) break;
      hipLaunchKernelGGL(k_halve_cells, gf, blk, 0, c->stream, F.n, F.cell);
      for (int a = 0; a < 3; ++a) dims[a] = (dims[a] + 1) / 2;
      continue;
    }
    AmgLevel C;
    C.n = nc;
    AMG_TRY(c->arena.get(size_t(F.n), &F.agg));
    AMG_TRY(c->arena.get(size_t(nc) * 3, &C.cell));
    hipLaunchKernelGGL(k_assign_agg, gf, blk, 0, c->stream, F.n, F.cell, dims[0], dims[1], dims[2],
                       flags, F.agg, C.cell);
    // Galerkin product through per-row hash tables
    int32_t* tkeys = nullptr;
    double* tvals = nullptr;
    AMG_TRY(c->arena.get(size_t(nc) * kTableCap, &tkeys));
    AMG_TRY(c->arena.get(size_t(nc) * kTableCap, &tvals));
    AMG_HIP(hipMemsetAsync(tkeys, 0xFF, size_t(nc) * kTableCap * 4, c->stream));
    AMG_HIP(hipMemsetAsync(tvals, 0, size_t(nc) * kTableCap * 8, c->stream));
    AMG_HIP(hipMemsetAsync(d_over, 0, 4, c->stream));
    hipLaunchKernelGGL(k_galerkin, gf, blk, 0, c->stream, F.n, F.A.indptr, F.A.indices, F.A.vals,
                       F.agg, tkeys, tvals, d_over);
    AMG_TRY(c->arena.get(size_t(nc) + 1, &C.A.indptr));
    AMG_HIP(hipMemsetAsync(C.A.indptr, 0, (size_t(nc) + 1) * 4, c->stream));
    const dim3 gc(ceil_div(nc, 256));
    hipLaunchKernelGGL(k_table_counts, gc, blk, 0, c->stream, nc, tkeys, C.A.indptr);
    AMG_TRY(exclusive_scan_i32(c, C.A.indptr, int64_t(nc) + 1));
    int32_t h2[2] = {0, 0};
    AMG_HIP(hipMemcpyAsync(&h2[0], C.A.indptr + nc, 4, hipMemcpyDeviceToHost, c->stream));
    AMG_HIP(hipMemcpyAsync(&h2[1], d_over, 4, hipMemcpyDeviceToHost, c->stream));
    AMG_HIP(hipStreamSynchronize(c->stream));
    if (h2[1]) {  // a coarse row has more than kTableCap neighbours: stop coarsening here
      F.agg = nullptr;
      break;
    }
    AMG_TRY(c->arena.get(size_t(h2[0]) + 1, &C.A.indices));
    AMG_TRY(c->arena.get(size_t(h2[0]) + 1, &C.A.vals));
    hipLaunchKernelGGL(k_table_to_csr, gc, blk, 0, c->stream, nc, tkeys, tvals, C.A.indptr,
                       C.A.indices, C.A.vals);
    AMG_TRY(c->arena.get(size_t(nc), &C.dinv));
    hipLaunchKernelGGL(k_l1_diag, gc, blk, 0, c->stream, nc, C.A.indptr, C.A.vals, C.dinv);
    AMG_HIP(hipGetLastError());
    AMG_TRY(alloc_vectors(c, C, true));
    H->lv.push_back(C);
    for (int a = 0; a < 3; ++a) dims[a] = (dims[a] + 1) / 2;
  }
  AmgLevel& last = H->lv.back();
  last.agg = nullptr;
  if (last.n <= kCoarseMax && H->lv.size() > 1) {
    H->nc = last.n;
    AMG_TRY(coarse_inverse(c, last, &H->dense_inv));
  }
  *out = H;
  return 0;
#undef AMG_TRY
#undef AMG_HIP
}

void amg_destroy(AmgHierarchy* h) { delete h; }

int amg_levels(const AmgHierarchy* h) { return h ? int(h->lv.size()) : 0; }

int amg_vcycle(Ctx* c, AmgHierarchy* H, const double* b, double* x) {
  const int nl = int(H->lv.size());
  const dim3 blk(256);
  // downward sweep
  for (int l = 0; l < nl; ++l) {
    AmgLevel& L = H->lv[size_t(l)];
    const double* bl = l == 0 ? b : L.b;
    double* xl = l == 0 ? x : L.x;
    const dim3 g(ceil_div(L.n, 256));
    if (l == nl - 1 && H->dense_inv) {
      hipLaunchKernelGGL(k_dense_solve, dim3(1), dim3(128), 0, c->stream, L.n, H->dense_inv, bl, xl);
      break;
    }
    hipLaunchKernelGGL(k_smooth0, g, blk, 0, c->stream, L.n, L.dinv, bl, xl);
    for (int sw = 1; sw < H->sweeps && l < nl - 1; ++sw) {
      hipLaunchKernelGGL(k_residual, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.A.vals, bl,
                         xl, L.r);
      hipLaunchKernelGGL(k_correct, g, blk, 0, c->stream, L.n, L.dinv, L.r, xl);
    }
    if (l == nl - 1) {  // coarsest without a dense solve: a few more sweeps
      for (int s = 0; s < 4; ++s) {
        hipLaunchKernelGGL(k_residual, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.A.vals,
                           bl, xl, L.r);
        hipLaunchKernelGGL(k_correct, g, blk, 0, c->stream, L.n, L.dinv, L.r, xl);
      }
      break;
    }
    hipLaunchKernelGGL(k_residual, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.A.vals, bl,
                       xl, L.r);
    AmgLevel& C = H->lv[size_t(l) + 1];
    PQ_HIP(hipMemsetAsync(C.b, 0, size_t(C.n) * 24, c->stream));
    hipLaunchKernelGGL(k_restrict, g, blk, 0, c->stream, L.n, L.agg, L.r, C.b);
  }
  // upward sweep
  for (int l = nl - 2; l >= 0; --l) {
    AmgLevel& L = H->lv[size_t(l)];
    AmgLevel& C = H->lv[size_t(l) + 1];
    const double* bl = l == 0 ? b : L.b;
    double* xl = l == 0 ? x : L.x;
    const dim3 g(ceil_div(L.n, 256));
    hipLaunchKernelGGL(k_prolong_add, g, blk, 0, c->stream, L.n, L.agg, C.x, H->alpha, xl);
    for (int sw = 0; sw < H->sweeps; ++sw) {
      hipLaunchKernelGGL(k_residual, g, blk, 0, c->stream, L.n, L.A.indptr, L.A.indices, L.A.vals, bl,
                         xl, L.r);
      hipLaunchKernelGGL(k_correct, g, blk, 0, c->stream, L.n, L.dinv, L.r, xl);
    }
  }
  PQ_HIP(hipGetLastError());
  return 0;
}

}  // namespace pyqsm
