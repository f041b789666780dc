// hull.hip — two geometric filters in front of the host-side convex hull of
// pcd.get_oriented_bounding_box() (pyQSM/geometry/skeletonize.py:240-241: the clamp box of the
// contraction loop comes from the oriented bounding box, which Open3D builds from a PCA of the
// convex hull's vertices).
//
// Qhull on every point of a scan costs 0.3 s per million points — a tenth of the whole
// 20-contraction loop — although the hull has ~150 vertices. The wrapper therefore
//   1. takes the extreme point of the cloud along a few dozen fixed directions
//      (pyqsm_extreme_points),
//   2. builds the small polytope of those points on the host and asks for every point that is NOT
//      strictly inside it (pyqsm_outside_halfspaces: ~1 % of a forest scan),
//   3. runs Qhull on those only.
// A point strictly inside a polytope whose corners are points of the cloud is strictly inside the
// cloud's hull, so the hull's vertices — and, indices ascending as Qhull lists them in 3-D, the
// PCA of them — are the same. Both passes are HBM-bound streams of 24 B per point.
#include "common.hpp"

namespace pyqsm {

static constexpr int kHullBlocks = 256;
static constexpr int kMaxPlanes = 256;

struct DirBest {
  double v;
  long long i;
};

// per block: the point with the largest x.d (lowest index on ties)
__global__ __launch_bounds__(256) void k_extreme(const double* __restrict__ xyz, int64_t n, double dx,
                                                 double dy, double dz, DirBest* __restrict__ part) {
  __shared__ DirBest red[256];
  DirBest b{-__builtin_inf(), 0x7FFFFFFFFFFFFFFFll};
  for (int64_t i = blockIdx.x * 256ll + threadIdx.x; i < n; i += int64_t(gridDim.x) * 256) {
    const double v = (xyz[3 * i] * dx + xyz[3 * i + 1] * dy) + xyz[3 * i + 2] * dz;
    if (v > b.v || (v == b.v && i < b.i)) b = DirBest{v, i};
  }
  red[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (int(threadIdx.x) < off) {
      const DirBest o = red[threadIdx.x + off];
      if (o.v > red[threadIdx.x].v || (o.v == red[threadIdx.x].v && o.i < red[threadIdx.x].i))
        red[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

struct Planes {
  int count;
  double eq[kMaxPlanes][4];
};

// flags[i] = 1 when a . x_i + o >= -margin for some plane (the point is not strictly inside)
__global__ __launch_bounds__(256) void k_outside(const double* __restrict__ xyz, int64_t n,
                                                 const Planes* __restrict__ pl, double margin,
                                                 int32_t* __restrict__ flags) {
  __shared__ double eq[kMaxPlanes][4];
  const int F = pl->count;
  for (int t = threadIdx.x; t < F * 4; t += 256) eq[t / 4][t % 4] = pl->eq[t / 4][t % 4];
  __syncthreads();
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flags[n] = 0;
    return;
  }
  const double x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
  bool out = false;
  for (int f = 0; f < F; ++f) out |= ((eq[f][0] * x + eq[f][1] * y) + eq[f][2] * z) + eq[f][3] >= -margin;
  flags[i] = out ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_outside_list(int64_t n, const int32_t* __restrict__ pos,
                                                      int64_t* __restrict__ idx) {
  const int64_t i = blockIdx.x * 256ll + threadIdx.x;
  if (i < n && pos[i + 1] != pos[i]) idx[pos[i]] = i;
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_extreme_points(const double* xyz, int64_t n, const double* dirs, int32_t n_dirs, int64_t* idx,
                         int32_t device) {
  PQ_API_RANGE("pyqsm_extreme_points");
  if (n <= 0 || n_dirs <= 0) return fail(PYQSM_EINVAL, "pyqsm_extreme_points: empty input");
  if (!xyz || !dirs || !idx) return fail(PYQSM_EINVAL, "pyqsm_extreme_points: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_xyz = nullptr;
  DirBest* d_part = nullptr;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(size_t(n_dirs) * kHullBlocks, &d_part));
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  const unsigned blocks = unsigned(std::min<int64_t>(ceil_div(n, 256), kHullBlocks));
  for (int d = 0; d < n_dirs; ++d)
    hipLaunchKernelGGL(k_extreme, dim3(blocks), dim3(256), 0, c->stream, d_xyz, n, dirs[3 * d], dirs[3 * d + 1],
                       dirs[3 * d + 2], d_part + size_t(d) * kHullBlocks);
  PQ_HIP(hipGetLastError());
  std::vector<DirBest> h(size_t(n_dirs) * kHullBlocks);
  PQ_HIP(hipMemcpyAsync(h.data(), d_part, h.size() * sizeof(DirBest), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  for (int d = 0; d < n_dirs; ++d) {
    DirBest b = h[size_t(d) * kHullBlocks];
    for (unsigned k = 1; k < blocks; ++k) {
      const DirBest o = h[size_t(d) * kHullBlocks + k];
      if (o.v > b.v || (o.v == b.v && o.i < b.i)) b = o;
    }
    idx[d] = b.i;
  }
  return 0;
}

int pyqsm_outside_halfspaces(const double* xyz, int64_t n, const double* eq, int32_t n_planes, double margin,
                             int64_t* idx, int64_t* count, int32_t device) {
  PQ_API_RANGE("pyqsm_outside_halfspaces");
  if (count) *count = 0;
  if (n < 0 || n_planes <= 0 || n_planes > kMaxPlanes)
    return fail(PYQSM_EINVAL, "pyqsm_outside_halfspaces: 1..%d planes", kMaxPlanes);
  if (n == 0) return 0;
  if (!xyz || !eq || !idx || !count) return fail(PYQSM_EINVAL, "pyqsm_outside_halfspaces: NULL pointer");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_xyz = nullptr;
  Planes* d_pl = nullptr;
  int32_t* d_flags = nullptr;
  int64_t* d_idx = nullptr;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(1, &d_pl));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_flags));
  PQ_TRY(c->arena.get(size_t(n), &d_idx));
  Planes hp;
  hp.count = n_planes;
  for (int f = 0; f < n_planes; ++f)
    for (int a = 0; a < 4; ++a) hp.eq[f][a] = eq[4 * f + a];
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_pl, &hp, sizeof(Planes), hipMemcpyHostToDevice, c->stream));
  const dim3 g(ceil_div(n + 1, 256)), blk(256);
  hipLaunchKernelGGL(k_outside, g, blk, 0, c->stream, d_xyz, n, d_pl, margin, d_flags);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, d_flags, n + 1));
  hipLaunchKernelGGL(k_outside_list, g, blk, 0, c->stream, n, d_flags, d_idx);
  PQ_HIP(hipGetLastError());
  int32_t m = 0;
  PQ_HIP(hipMemcpyAsync(&m, d_flags + n, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));  // hp may go
  if (m > 0) {
    PQ_HIP(hipMemcpyAsync(idx, d_idx, size_t(m) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
  }
  *count = m;
  return 0;
}

}  // extern "C"
