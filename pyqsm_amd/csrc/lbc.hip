// lbc.hip — the Laplacian-contraction solve of pyQSM/geometry/skeletonize.py:148-180
// (least_squares_sparse) on gfx950, plus the per-point clamp of :291-296.
//
// The reference stacks A = [L*W_L ; W_H], forms A'A and factorises it three
// times with SuperLU (once per coordinate). Here the normal equations
//     (W_L L' L W_L + W_H^2) x = W_H^2 p
// are solved matrix-free by a Jacobi-preconditioned conjugate gradient that
// carries the three coordinates through every sparse pass together (one read of
// L serves x, y and z). L'L is never formed. L must be symmetric (it is: the
// point-cloud Laplacian is), so L' x is computed as L x.
//
// HBM traffic per CG iteration (fp64, CSR with 32-bit indices):
//   2 sparse passes  = 2 * (12*nnz + 52*n) bytes      (SURVEY.md §8 d-roofline)
//   vector updates   ~ 10 three-column streams = 240*n bytes
// All scalars (alpha, beta, dot products) stay on the device; the host looks at
// the residual only every `kCheckEvery` iterations.
#include "common.hpp"

namespace pyqsm {

static constexpr int kCheckEvery = 25;
static constexpr int kStallIters = 1500;
// CG residuals are not monotone, so stagnation only counts once the solve is
// close to its attainable accuracy
static constexpr double kStallBelow = 1e-8;

// y = L (s .* x)   three columns; one lane per row (rows hold ~14 entries).
__global__ __launch_bounds__(256) void k_spmv3(int n, const int32_t* __restrict__ indptr,
                                               const int32_t* __restrict__ indices,
                                               const double* __restrict__ vals,
                                               const double* __restrict__ s /*may be null*/,
                                               const double* __restrict__ x,
                                               double* __restrict__ y) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const int b = indptr[i], e = indptr[i + 1];
  for (int j = b; j < e; ++j) {
    const int col = indices[j];
    double v = vals[j];
    if (s) v *= s[col];
    a0 += v * x[3 * col];
    a1 += v * x[3 * col + 1];
    a2 += v * x[3 * col + 2];
  }
  y[3 * i] = a0;
  y[3 * i + 1] = a1;
  y[3 * i + 2] = a2;
}

// diag(A)_i = wl_i^2 * sum_j L_ji^2 + wh_i^2 ; symmetric L: column norm = row norm
__global__ __launch_bounds__(256) void k_diag(int n, const int32_t* __restrict__ indptr,
                                              const double* __restrict__ vals,
                                              const double* __restrict__ wl,
                                              const double* __restrict__ wh,
                                              double* __restrict__ minv) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double s = 0.0;
  for (int j = indptr[i]; j < indptr[i + 1]; ++j) s += vals[j] * vals[j];
  const double d = wl[i] * wl[i] * s + wh[i] * wh[i];
  minv[i] = d > 0.0 ? 1.0 / d : 1.0;
}

// Block reduction of three partial sums followed by one fp64 atomic per column.
__device__ __forceinline__ void reduce3_atomic(double v0, double v1, double v2, double* out) {
  __shared__ double red[3][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    v0 += __shfl_down(v0, off, 64);
    v1 += __shfl_down(v1, off, 64);
    v2 += __shfl_down(v2, off, 64);
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) {
    red[0][w] = v0;
    red[1][w] = v1;
    red[2][w] = v2;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    double t = (red[threadIdx.x][0] + red[threadIdx.x][1]) +
               (red[threadIdx.x][2] + red[threadIdx.x][3]);
    atomicAdd(out + threadIdx.x, t);
  }
}

// Scalars on the device: [0..2] rz, [3..5] pq, [6..8] rz_new, [9..11] rr, [12..14] bb
struct Scal {
  double rz[3], pq[3], rz_new[3], rr[3], bb[3];
};

// r = b - A x0 with b = wh^2 p and x0 = p:  r = -(wl .* L(L(wl .* p)))
// t already holds L(L(wl.*p)). z = Minv r, dir = z; accumulates rz, rr, bb.
__global__ __launch_bounds__(256) void k_init(int n, const double* __restrict__ t,
                                              const double* __restrict__ wl,
                                              const double* __restrict__ wh,
                                              const double* __restrict__ pts,
                                              const double* __restrict__ minv,
                                              double* __restrict__ r, double* __restrict__ dir,
                                              Scal* __restrict__ sc) {
  int i = blockIdx.x * 256 + threadIdx.x;
  double rz[3] = {0, 0, 0}, rr[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
  if (i < n) {
    const double w = wl[i], h2 = wh[i] * wh[i], mi = minv[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double ri = -(w * t[3 * i + k]);
      const double zi = mi * ri;
      const double bi = h2 * pts[3 * i + k];
      r[3 * i + k] = ri;
      dir[3 * i + k] = zi;
      rz[k] = ri * zi;
      rr[k] = ri * ri;
      bb[k] = bi * bi;
    }
  }
  reduce3_atomic(rz[0], rz[1], rz[2], sc->rz);
  __syncthreads();
  reduce3_atomic(rr[0], rr[1], rr[2], sc->rr);
  __syncthreads();
  reduce3_atomic(bb[0], bb[1], bb[2], sc->bb);
}

// q = wl .* t2 + wh^2 .* dir   (t2 = L(L(wl .* dir))) ; pq += dir . q
__global__ __launch_bounds__(256) void k_apply_tail(int n, const double* __restrict__ t2,
                                                    const double* __restrict__ wl,
                                                    const double* __restrict__ wh,
                                                    const double* __restrict__ dir,
                                                    double* __restrict__ q,
                                                    Scal* __restrict__ sc) {
  int i = blockIdx.x * 256 + threadIdx.x;
  double pq[3] = {0, 0, 0};
  if (i < n) {
    const double w = wl[i], h2 = wh[i] * wh[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double d = dir[3 * i + k];
      const double qi = w * t2[3 * i + k] + h2 * d;
      q[3 * i + k] = qi;
      pq[k] = d * qi;
    }
  }
  reduce3_atomic(pq[0], pq[1], pq[2], sc->pq);
}

// alpha = rz/pq ; x += alpha dir ; r -= alpha q ; z = Minv r ; rz_new += r.z ; rr += r.r
__global__ __launch_bounds__(256) void k_update(int n, const double* __restrict__ dir,
                                                const double* __restrict__ q,
                                                const double* __restrict__ minv,
                                                double* __restrict__ x, double* __restrict__ r,
                                                double* __restrict__ z, Scal* __restrict__ sc) {
  int i = blockIdx.x * 256 + threadIdx.x;
  double rz[3] = {0, 0, 0}, rr[3] = {0, 0, 0};
  if (i < n) {
    const double mi = minv[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const double pqk = sc->pq[k];
      const double alpha = pqk != 0.0 ? sc->rz[k] / pqk : 0.0;
      x[3 * i + k] += alpha * dir[3 * i + k];
      const double ri = r[3 * i + k] - alpha * q[3 * i + k];
      const double zi = mi * ri;
      r[3 * i + k] = ri;
      z[3 * i + k] = zi;
      rz[k] = ri * zi;
      rr[k] = ri * ri;
    }
  }
  reduce3_atomic(rz[0], rz[1], rz[2], sc->rz_new);
  __syncthreads();
  reduce3_atomic(rr[0], rr[1], rr[2], sc->rr);
}

// beta = rz_new/rz ; dir = z + beta dir
__global__ __launch_bounds__(256) void k_direction(int n, const double* __restrict__ z,
                                                   double* __restrict__ dir,
                                                   const Scal* __restrict__ sc) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double rzk = sc->rz[k];
    const double beta = rzk != 0.0 ? sc->rz_new[k] / rzk : 0.0;
    dir[3 * i + k] = z[3 * i + k] + beta * dir[3 * i + k];
  }
}

// rz <- rz_new (unless `first`: rz was accumulated directly by k_init); clear
// the accumulators of the next iteration; keep rr in rr_out
__global__ void k_roll(Scal* sc, double* rr_out, int first) {
  int k = threadIdx.x;
  if (k < 3) {
    if (!first) sc->rz[k] = sc->rz_new[k];
    sc->rz_new[k] = 0.0;
    sc->pq[k] = 0.0;
    rr_out[k] = sc->rr[k];
    sc->rr[k] = 0.0;
  }
}

__global__ __launch_bounds__(256) void k_clamp(int64_t n3, double* __restrict__ pts, double lo0,
                                               double lo1, double lo2, double hi0, double hi1,
                                               double hi2) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n3) return;
  const int a = int(i % 3);
  const double lo = a == 0 ? lo0 : (a == 1 ? lo1 : lo2);
  const double hi = a == 0 ? hi0 : (a == 1 ? hi1 : hi2);
  double v = pts[i];
  // skeletonize.py:293-296: if v < lo: v = lo ; if v > hi: v = hi  (NaN stays NaN)
  if (v < lo) v = lo;
  if (v > hi) v = hi;
  pts[i] = v;
}

struct DevCsr {
  int32_t *indptr, *indices;
  double* vals;
};

static int upload_csr(Ctx* c, const int32_t* indptr, const int32_t* indices, const double* vals,
                      int64_t n, DevCsr* d, int64_t* nnz_out) {
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 rows");
  const int64_t nnz = indptr[n];
  if (nnz < 0 || indptr[0] != 0) return fail(PYQSM_EINVAL, "malformed CSR indptr");
  PQ_TRY(c->arena.get(size_t(n) + 1, &d->indptr));
  PQ_TRY(c->arena.get(size_t(nnz) + 1, &d->indices));
  PQ_TRY(c->arena.get(size_t(nnz) + 1, &d->vals));
  PQ_HIP(hipMemcpyAsync(d->indptr, indptr, (size_t(n) + 1) * 4, hipMemcpyHostToDevice, c->stream));
  if (nnz) {
    PQ_HIP(hipMemcpyAsync(d->indices, indices, size_t(nnz) * 4, hipMemcpyHostToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(d->vals, vals, size_t(nnz) * 8, hipMemcpyHostToDevice, c->stream));
  }
  *nnz_out = nnz;
  return 0;
}

// Device-resident solve; every pointer is HBM. Returns iterations / residuals on the host.
int lbc_solve_device(Ctx* c, const DevCsr& L, int64_t n, const double* wl, const double* wh,
                     const double* pts, double rtol, int32_t max_it, double* x, int32_t* iters,
                     double resid[3]) {
  const int N = int(n);
  const dim3 grid(ceil_div(n, 256)), block(256);
  double *t1, *t2, *r, *z, *dir, *q, *minv, *d_rr;
  Scal* sc;
  PQ_TRY(c->arena.get(size_t(n) * 3, &t1));
  PQ_TRY(c->arena.get(size_t(n) * 3, &t2));
  PQ_TRY(c->arena.get(size_t(n) * 3, &r));
  PQ_TRY(c->arena.get(size_t(n) * 3, &z));
  PQ_TRY(c->arena.get(size_t(n) * 3, &dir));
  PQ_TRY(c->arena.get(size_t(n) * 3, &q));
  PQ_TRY(c->arena.get(size_t(n), &minv));
  PQ_TRY(c->arena.get(1, &sc));
  PQ_TRY(c->arena.get(3, &d_rr));
  PQ_HIP(hipMemsetAsync(sc, 0, sizeof(Scal), c->stream));
  PQ_HIP(hipMemcpyAsync(x, pts, size_t(n) * 24, hipMemcpyDeviceToDevice, c->stream));
  hipLaunchKernelGGL(k_diag, grid, block, 0, c->stream, N, L.indptr, L.vals, wl, wh, minv);
  // r0 = b - A p = -(wl .* L L (wl .* p))
  hipLaunchKernelGGL(k_spmv3, grid, block, 0, c->stream, N, L.indptr, L.indices, L.vals, wl, pts,
                     t1);
  hipLaunchKernelGGL(k_spmv3, grid, block, 0, c->stream, N, L.indptr, L.indices, L.vals,
                     static_cast<const double*>(nullptr), t1, t2);
  hipLaunchKernelGGL(k_init, grid, block, 0, c->stream, N, t2, wl, wh, pts, minv, r, dir, sc);
  PQ_HIP(hipGetLastError());
  Scal h;
  PQ_HIP(hipMemcpyAsync(&h, sc, sizeof(Scal), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  double bnorm[3];
  bool done = true;
  for (int k = 0; k < 3; ++k) {
    bnorm[k] = std::sqrt(h.bb[k]);
    resid[k] = bnorm[k] > 0 ? std::sqrt(h.rr[k]) / bnorm[k] : 0.0;
    if (resid[k] > rtol) done = false;
  }
  // k_init accumulated rr; clear it for the loop (rz stays)
  hipLaunchKernelGGL(k_roll, dim3(1), dim3(64), 0, c->stream, sc, d_rr, 1);
  // Past the attainable accuracy (about cond(A) * 1e-16) the recurrences drift and
  // the residual grows again, so the best iterate is kept and the loop stops once
  // the residual has not improved for kStallIters iterations.
  double* x_best;
  PQ_TRY(c->arena.get(size_t(n) * 3, &x_best));
  PQ_HIP(hipMemcpyAsync(x_best, x, size_t(n) * 24, hipMemcpyDeviceToDevice, c->stream));
  double best = std::max(resid[0], std::max(resid[1], resid[2]));
  double best_res[3] = {resid[0], resid[1], resid[2]};
  int it = 0, best_it = 0;
  bool broke = false;
  while (!done && it < max_it) {
    const int burst = std::min<int>(kCheckEvery, max_it - it);
    for (int b = 0; b < burst; ++b) {
      ProfScope ps(c, "lbc_cg_iter");
      hipLaunchKernelGGL(k_spmv3, grid, block, 0, c->stream, N, L.indptr, L.indices, L.vals, wl,
                         dir, t1);
      hipLaunchKernelGGL(k_spmv3, grid, block, 0, c->stream, N, L.indptr, L.indices, L.vals,
                         static_cast<const double*>(nullptr), t1, t2);
      hipLaunchKernelGGL(k_apply_tail, grid, block, 0, c->stream, N, t2, wl, wh, dir, q, sc);
      hipLaunchKernelGGL(k_update, grid, block, 0, c->stream, N, dir, q, minv, x, r, z, sc);
      hipLaunchKernelGGL(k_direction, grid, block, 0, c->stream, N, z, dir, sc);
      hipLaunchKernelGGL(k_roll, dim3(1), dim3(64), 0, c->stream, sc, d_rr, 0);
    }
    PQ_HIP(hipGetLastError());
    it += burst;
    double rr[3];
    PQ_HIP(hipMemcpyAsync(rr, d_rr, 24, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    done = true;
    double worst = 0.0;
    for (int k = 0; k < 3; ++k) {
      resid[k] = bnorm[k] > 0 ? std::sqrt(rr[k]) / bnorm[k] : 0.0;
      if (!(resid[k] <= rtol)) done = false;
      if (!std::isfinite(resid[k])) broke = true;
      worst = std::max(worst, resid[k]);
    }
    if (broke) break;
    if (worst < best) {
      best = worst;
      best_it = it;
      for (int k = 0; k < 3; ++k) best_res[k] = resid[k];
      PQ_HIP(hipMemcpyAsync(x_best, x, size_t(n) * 24, hipMemcpyDeviceToDevice, c->stream));
    } else if (best < kStallBelow && it - best_it >= kStallIters) {
      break;  // converged as far as fp64 allows; the recurrences are drifting now
    }
  }
  *iters = it;
  if (!done) {
    // hand back the best iterate seen
    PQ_HIP(hipMemcpyAsync(x, x_best, size_t(n) * 24, hipMemcpyDeviceToDevice, c->stream));
    for (int k = 0; k < 3; ++k) resid[k] = best_res[k];
    return fail(PYQSM_ENOCONV, "CG stopped after %d iterations (%s); best residual %.3e at %d",
                it, broke ? "breakdown" : (it >= max_it ? "max_it" : "stagnation"), best, best_it);
  }
  return 0;
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_spmv3(const int32_t* indptr, const int32_t* indices, const double* vals, int64_t n,
                const double* x, double* y, int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n == 0) return 0;
  if (!indptr || !x || !y) return fail(PYQSM_EINVAL, "pyqsm_spmv3: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  DevCsr L;
  int64_t nnz;
  PQ_TRY(upload_csr(c, indptr, indices, vals, n, &L, &nnz));
  double *dx, *dy;
  PQ_TRY(c->arena.get(size_t(n) * 3, &dx));
  PQ_TRY(c->arena.get(size_t(n) * 3, &dy));
  PQ_HIP(hipMemcpyAsync(dx, x, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "spmv3");
    hipLaunchKernelGGL(k_spmv3, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, int(n), L.indptr,
                       L.indices, L.vals, static_cast<const double*>(nullptr), dx, dy);
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(y, dy, size_t(n) * 24, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_lbc_solve(const int32_t* indptr, const int32_t* indices, const double* vals, int64_t n,
                    const double* wl, const double* wh, const double* pts, double rtol,
                    int32_t max_it, double* out, int32_t* iters, double* resid, int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (iters) *iters = 0;
  if (n == 0) return 0;
  if (!indptr || !wl || !wh || !pts || !out)
    return fail(PYQSM_EINVAL, "pyqsm_lbc_solve: NULL pointer");
  if (!(rtol > 0)) return fail(PYQSM_EINVAL, "rtol must be positive");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  DevCsr L;
  int64_t nnz;
  PQ_TRY(upload_csr(c, indptr, indices, vals, n, &L, &nnz));
  double *d_wl, *d_wh, *d_pts, *d_x;
  PQ_TRY(c->arena.get(size_t(n), &d_wl));
  PQ_TRY(c->arena.get(size_t(n), &d_wh));
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_pts));
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_x));
  PQ_HIP(hipMemcpyAsync(d_wl, wl, size_t(n) * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_wh, wh, size_t(n) * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_pts, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  int32_t it = 0;
  double rs[3] = {0, 0, 0};
  int rc = lbc_solve_device(c, L, n, d_wl, d_wh, d_pts, rtol, max_it, d_x, &it, rs);
  if (iters) *iters = it;
  if (resid) {
    resid[0] = rs[0];
    resid[1] = rs[1];
    resid[2] = rs[2];
  }
  if (rc != 0 && rc != PYQSM_ENOCONV) return rc;
  PQ_HIP(hipMemcpyAsync(out, d_x, size_t(n) * 24, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return rc;
}

int pyqsm_clamp(double* pts, int64_t n, const double lo[3], const double hi[3], int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n == 0) return 0;
  if (!pts || !lo || !hi) return fail(PYQSM_EINVAL, "pyqsm_clamp: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d));
  PQ_HIP(hipMemcpyAsync(d, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "clamp");
    hipLaunchKernelGGL(k_clamp, dim3(ceil_div(n * 3, 256)), dim3(256), 0, c->stream, n * 3, d,
                       lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(pts, d, size_t(n) * 24, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
