// ransac.hip — per-hypothesis parallel RANSAC for circles / cylinders on gfx950.
//
// Stands in for pyransac3d.Circle().fit / Cylinder().fit as called from
// pyQSM/math_utils/fit.py:277-283 (pyransac3d is not vendored by the
// reference; the algorithm restated here is its published circle.py /
// cylinder.py / aux_functions.rodrigues_rot). The random 3-point samples come
// from the caller so that runs are reproducible.
//
//   k_models   one lane per hypothesis: circle through the three samples
//              (plane normal, Rodrigues rotation to z, 2-D circumcentre, back)
//   k_count    lanes = hypotheses, points staged through LDS in SoA tiles and
//              read as broadcasts; one atomicAdd per (block, hypothesis)
//   k_flags    inlier flags of the winning model -> scan -> ascending indices
//
// The point-to-model distance follows NumPy's operation order exactly (each
// product and sum rounded separately, sums left to right, IEEE sqrt), so for a
// given model the inlier set is bit-identical to the NumPy statement in
// oracle/__init__.py. FP64 VALU bound: ~35 flop per (point, hypothesis).
#include <vector>

#include "common.hpp"

namespace pyqsm {

struct Model {
  double cx, cy, cz, ax, ay, az, r, valid;
};

struct V3 {
  double x, y, z;
};
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ double dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double norm(V3 a) { return sqrt(dot(a, a)); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// aux_functions.rodrigues_rot for one point.
__device__ V3 rodrigues(V3 p, V3 n0, V3 n1) {
  n0 = scale(n0, 1.0 / norm(n0));
  n1 = scale(n1, 1.0 / norm(n1));
  V3 k = cross(n0, n1);
  double kn = norm(k);
  if (kn == 0.0) return p;
  k = scale(k, 1.0 / kn);
  double theta = acos(dot(n0, n1));
  double c = cos(theta), s = sin(theta);
  V3 kxp = cross(k, p);
  double kd = dot(k, p) * (1.0 - c);
  return {p.x * c + kxp.x * s + k.x * kd, p.y * c + kxp.y * s + k.y * kd,
          p.z * c + kxp.z * s + k.z * kd};
}

// the model through three points of the set pts[0..n) (invalid when an index is out of range or the
// construction is not finite)
__device__ Model model_of(const double* __restrict__ pts, int64_t n, int64_t i0, int64_t i1, int64_t i2) {
  Model m = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i0 >= 0 && i1 >= 0 && i2 >= 0 && i0 < n && i1 < n && i2 < n) {
    V3 p0 = {pts[3 * i0], pts[3 * i0 + 1], pts[3 * i0 + 2]};
    V3 p1 = {pts[3 * i1], pts[3 * i1 + 1], pts[3 * i1 + 2]};
    V3 p2 = {pts[3 * i2], pts[3 * i2 + 1], pts[3 * i2 + 2]};
    V3 vA = sub(p1, p0);
    vA = scale(vA, 1.0 / norm(vA));
    V3 vB = sub(p2, p0);
    vB = scale(vB, 1.0 / norm(vB));
    V3 vC = cross(vA, vB);
    vC = scale(vC, 1.0 / norm(vC));
    const V3 ez = {0.0, 0.0, 1.0};
    V3 q[3] = {rodrigues(p0, vC, ez), rodrigues(p1, vC, ez), rodrigues(p2, vC, ez)};
    double ma = 0.0, mb = 0.0;
    for (int it = 0; it < 3; ++it) {
      ma = (q[1].y - q[0].y) / (q[1].x - q[0].x);
      mb = (q[2].y - q[1].y) / (q[2].x - q[1].x);
      if (ma == 0.0) {  // np.roll(P_rot, -1, axis=0)
        V3 t = q[0];
        q[0] = q[1];
        q[1] = q[2];
        q[2] = t;
      } else {
        break;
      }
    }
    double pcx = (ma * mb * (q[0].y - q[2].y) + mb * (q[0].x + q[1].x) - ma * (q[1].x + q[2].x)) /
                 (2.0 * (mb - ma));
    double pcy = -1.0 / ma * (pcx - (q[0].x + q[1].x) / 2.0) + (q[0].y + q[1].y) / 2.0;
    V3 pc = {pcx, pcy, 0.0};
    double radius = norm(sub(pc, q[0]));
    V3 ctr = rodrigues(pc, ez, vC);
    bool ok = isfinite(ctr.x) && isfinite(ctr.y) && isfinite(ctr.z) && isfinite(radius) &&
              isfinite(vC.x) && isfinite(vC.y) && isfinite(vC.z);
    m = {ctr.x, ctr.y, ctr.z, vC.x, vC.y, vC.z, radius, ok ? 1.0 : 0.0};
  }
  return m;
}

__global__ __launch_bounds__(256) void k_models(const double* __restrict__ pts, int64_t n,
                                                const int64_t* __restrict__ triples, int64_t H,
                                                Model* __restrict__ models) {
  int64_t h = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (h >= H) return;
  models[h] = model_of(pts, n, triples[3 * h], triples[3 * h + 1], triples[3 * h + 2]);
}

// the same for S point sets stacked in one array: hypothesis h of set s is thread s * H + h
__global__ __launch_bounds__(256) void k_models_seg(const double* __restrict__ pts,
                                                    const int64_t* __restrict__ seg_start, int64_t S,
                                                    const int64_t* __restrict__ triples, int64_t H,
                                                    Model* __restrict__ models) {
  const int64_t t = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (t >= S * H) return;
  const int64_t s = t / H;
  const int64_t b = seg_start[s];
  models[t] = model_of(pts + 3 * b, seg_start[s + 1] - b, triples[3 * t], triples[3 * t + 1], triples[3 * t + 2]);
}

// |distance| of one point to one model, NumPy operation order (oracle.ransac_distance).
template <int SHAPE>
__device__ __forceinline__ double model_dist(const Model& m, double x, double y, double z) {
  // d = center - pts ; cr = cross(axis, d)
  const double dx = m.cx - x, dy = m.cy - y, dz = m.cz - z;
  const double c0 = m.ay * dz - m.az * dy;
  const double c1 = m.az * dx - m.ax * dz;
  const double c2 = m.ax * dy - m.ay * dx;
  const double nr = sqrt((c0 * c0 + c1 * c1) + c2 * c2);
  if (SHAPE == 1) return fabs(nr - m.r);
  const double plane = (m.ax * (x - m.cx) + m.ay * (y - m.cy)) + m.az * (z - m.cz);
  const double dinf = nr - m.r;
  return fabs(sqrt(dinf * dinf + plane * plane));
}

static constexpr int kTile = 1024;  // points per LDS tile

template <int SHAPE>
__global__ __launch_bounds__(256) void k_count(const double* __restrict__ pts, int64_t n,
                                               const Model* __restrict__ models, int64_t H,
                                               double thresh, int32_t* __restrict__ counts) {
  __shared__ double lx[kTile], ly[kTile], lz[kTile];
  const int64_t base = int64_t(blockIdx.y) * kTile;
  const int cnt = int(n - base < kTile ? n - base : kTile);
  for (int i = threadIdx.x; i < cnt; i += 256) {
    lx[i] = pts[3 * (base + i)];
    ly[i] = pts[3 * (base + i) + 1];
    lz[i] = pts[3 * (base + i) + 2];
  }
  __syncthreads();
  const int64_t h = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (h >= H) return;
  const Model m = models[h];
  if (m.valid == 0.0) return;
  int32_t c = 0;
  for (int i = 0; i < cnt; ++i) c += model_dist<SHAPE>(m, lx[i], ly[i], lz[i]) <= thresh;
  if (c) atomicAdd(&counts[h], c);
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k_flags(const double* __restrict__ pts, int64_t n,
                                               const Model* __restrict__ models, int64_t best,
                                               double thresh, int32_t* __restrict__ flags) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flags[n] = 0;
    return;
  }
  const Model m = models[best];
  flags[i] = model_dist<SHAPE>(m, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]) <= thresh;
}

__global__ __launch_bounds__(256) void k_compact(int64_t n, const double* __restrict__ pts,
                                                 const Model* __restrict__ models, int64_t best,
                                                 double thresh, int shape,
                                                 const int32_t* __restrict__ pos,
                                                 int64_t* __restrict__ out) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] != pos[i]) out[pos[i]] = i;
}

// ---- many independent fits in one call (pyqsm_ransac_batch) ---------------------------------

// blockIdx.y = a tile of kTile points of one set (tile_set / tile_base list the tiles of all sets)
template <int SHAPE>
__global__ __launch_bounds__(256) void k_count_seg(const double* __restrict__ pts,
                                                   const int64_t* __restrict__ seg_start,
                                                   const int32_t* __restrict__ tile_set,
                                                   const int64_t* __restrict__ tile_base,
                                                   const Model* __restrict__ models, int64_t H,
                                                   double thresh, int32_t* __restrict__ counts) {
  __shared__ double lx[kTile], ly[kTile], lz[kTile];
  const int s = tile_set[blockIdx.y];
  const int64_t base = tile_base[blockIdx.y], end = seg_start[s + 1];
  const int cnt = int(end - base < kTile ? end - base : kTile);
  for (int i = threadIdx.x; i < cnt; i += 256) {
    lx[i] = pts[3 * (base + i)];
    ly[i] = pts[3 * (base + i) + 1];
    lz[i] = pts[3 * (base + i) + 2];
  }
  __syncthreads();
  const int64_t h = int64_t(blockIdx.x) * 256 + threadIdx.x;
  if (h >= H) return;
  const Model m = models[int64_t(s) * H + h];
  if (m.valid == 0.0) return;
  int32_t c = 0;
  for (int i = 0; i < cnt; ++i) c += model_dist<SHAPE>(m, lx[i], ly[i], lz[i]) <= thresh;
  if (c) atomicAdd(&counts[int64_t(s) * H + h], c);
}

// a block per set: the first hypothesis with the largest count (none when every count is 0)
__global__ __launch_bounds__(256) void k_best_seg(const int32_t* __restrict__ counts, int64_t H,
                                                  int64_t* __restrict__ best, int32_t* __restrict__ best_cnt) {
  __shared__ int32_t rc[256];
  __shared__ int64_t rh[256];
  const int64_t s = blockIdx.x;
  int32_t c = 0;
  int64_t hb = -1;
  for (int64_t h = threadIdx.x; h < H; h += 256) {
    const int32_t v = counts[s * H + h];
    if (v > c) {  // ascending h within a thread: the first of the largest
      c = v;
      hb = h;
    }
  }
  rc[threadIdx.x] = c;
  rh[threadIdx.x] = hb;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (int(threadIdx.x) < off) {
      const int32_t oc = rc[threadIdx.x + off];
      const int64_t oh = rh[threadIdx.x + off];
      if (oc > rc[threadIdx.x] || (oc == rc[threadIdx.x] && oc > 0 && oh < rh[threadIdx.x])) {
        rc[threadIdx.x] = oc;
        rh[threadIdx.x] = oh;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    best[s] = rh[0];
    best_cnt[s] = rc[0];
  }
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k_flags_seg(const double* __restrict__ pts, int64_t n,
                                                   const int64_t* __restrict__ seg_start, int64_t S,
                                                   const Model* __restrict__ models, int64_t H,
                                                   const int64_t* __restrict__ best, double thresh,
                                                   int32_t* __restrict__ flags, int32_t* __restrict__ set_of) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i > n) return;
  if (i == n) {
    flags[n] = 0;
    return;
  }
  int64_t lo = 0, hi = S - 1;  // the set of point i: the last s with seg_start[s] <= i
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) >> 1;
    if (seg_start[mid] <= i) lo = mid; else hi = mid - 1;
  }
  set_of[i] = int32_t(lo);
  const int64_t b = best[lo];
  int32_t f = 0;
  if (b >= 0) {
    const Model m = models[lo * H + b];
    f = model_dist<SHAPE>(m, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]) <= thresh;
  }
  flags[i] = f;
}

// inliers as indices local to their set, set by set; the winning models gathered per set
__global__ __launch_bounds__(256) void k_compact_seg(int64_t n, const int64_t* __restrict__ seg_start,
                                                     const int32_t* __restrict__ set_of,
                                                     const int32_t* __restrict__ pos, int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] != pos[i]) out[pos[i]] = i - seg_start[set_of[i]];
}

__global__ __launch_bounds__(256) void k_gather_best(int64_t S, const Model* __restrict__ models, int64_t H,
                                                     const int64_t* __restrict__ best, Model* __restrict__ out) {
  const int64_t s = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (s >= S) return;
  out[s] = best[s] >= 0 ? models[s * H + best[s]] : Model{0, 0, 0, 0, 0, 0, 0, 0};
}

static int count_on_device(Ctx* c, const double* d_pts, int64_t n, const Model* d_models, int64_t H,
                           int shape, double thresh, int32_t* d_counts) {
  PQ_HIP(hipMemsetAsync(d_counts, 0, size_t(H) * 4, c->stream));
  if (n == 0 || H == 0) return 0;
  ProfScope ps(c, "ransac_count");
  const dim3 grid(ceil_div(H, 256), ceil_div(n, kTile));
  if (shape == 0)
    hipLaunchKernelGGL(k_count<0>, grid, dim3(256), 0, c->stream, d_pts, n, d_models, H, thresh,
                       d_counts);
  else
    hipLaunchKernelGGL(k_count<1>, grid, dim3(256), 0, c->stream, d_pts, n, d_models, H, thresh,
                       d_counts);
  PQ_HIP(hipGetLastError());
  return 0;
}

static int check_args(const double* pts, int64_t n, int64_t H, int32_t shape) {
  if (n < 0 || H < 0) return fail(PYQSM_EINVAL, "negative size");
  if (shape != 0 && shape != 1) return fail(PYQSM_EINVAL, "shape must be 0 (circle) or 1 (cylinder)");
  if (n > 0 && !pts) return fail(PYQSM_EINVAL, "pts is NULL");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points per call");
  return 0;
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_ransac_models(const double* pts, int64_t n, const int64_t* triples, int64_t H,
                        double* models, int32_t device) {
  PQ_API_RANGE("pyqsm_ransac_models");
  PQ_TRY(check_args(pts, n, H, 0));
  if (H == 0) return 0;
  if (!triples || !models) return fail(PYQSM_EINVAL, "pyqsm_ransac_models: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_pts;
  int64_t* d_tri;
  Model* d_models;
  PQ_TRY(c->arena.get(size_t(n) * 3 + 1, &d_pts));
  PQ_TRY(c->arena.get(size_t(H) * 3, &d_tri));
  PQ_TRY(c->arena.get(size_t(H), &d_models));
  if (n) PQ_HIP(hipMemcpyAsync(d_pts, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_tri, triples, size_t(H) * 24, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_models, dim3(ceil_div(H, 256)), dim3(256), 0, c->stream, d_pts, n, d_tri, H,
                     d_models);
  PQ_HIP(hipGetLastError());
  PQ_HIP(hipMemcpyAsync(models, d_models, size_t(H) * sizeof(Model), hipMemcpyDeviceToHost,
                        c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_ransac_count(const double* pts, int64_t n, const double* models, int64_t H,
                       int32_t shape, double thresh, int32_t* counts, int32_t device) {
  PQ_API_RANGE("pyqsm_ransac_count");
  PQ_TRY(check_args(pts, n, H, shape));
  if (H == 0) return 0;
  if (!models || !counts) return fail(PYQSM_EINVAL, "pyqsm_ransac_count: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_pts;
  Model* d_models;
  int32_t* d_counts;
  PQ_TRY(c->arena.get(size_t(n) * 3 + 1, &d_pts));
  PQ_TRY(c->arena.get(size_t(H), &d_models));
  PQ_TRY(c->arena.get(size_t(H), &d_counts));
  if (n) PQ_HIP(hipMemcpyAsync(d_pts, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_models, models, size_t(H) * sizeof(Model), hipMemcpyHostToDevice,
                        c->stream));
  PQ_TRY(count_on_device(c, d_pts, n, d_models, H, shape, thresh, d_counts));
  PQ_HIP(hipMemcpyAsync(counts, d_counts, size_t(H) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_ransac(const double* pts, int64_t n, const int64_t* triples, int64_t H, int32_t shape,
                 double thresh, double center[3], double axis[3], double* radius,
                 int64_t* inliers, int64_t* n_inliers, int64_t* best_out, int32_t device) {
  PQ_API_RANGE("pyqsm_ransac");
  PQ_TRY(check_args(pts, n, H, shape));
  if (n_inliers) *n_inliers = 0;
  if (best_out) *best_out = -1;
  if (n == 0 || H == 0) return 0;
  if (!triples || !center || !axis || !radius || !inliers || !n_inliers)
    return fail(PYQSM_EINVAL, "pyqsm_ransac: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_pts;
  int64_t *d_tri, *d_out;
  Model* d_models;
  int32_t *d_counts, *d_flags;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_pts));
  PQ_TRY(c->arena.get(size_t(H) * 3, &d_tri));
  PQ_TRY(c->arena.get(size_t(H), &d_models));
  PQ_TRY(c->arena.get(size_t(H), &d_counts));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_flags));
  PQ_TRY(c->arena.get(size_t(n), &d_out));
  PQ_HIP(hipMemcpyAsync(d_pts, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_tri, triples, size_t(H) * 24, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "ransac_models");
    hipLaunchKernelGGL(k_models, dim3(ceil_div(H, 256)), dim3(256), 0, c->stream, d_pts, n, d_tri,
                       H, d_models);
    PQ_HIP(hipGetLastError());
  }
  PQ_TRY(count_on_device(c, d_pts, n, d_models, H, shape, thresh, d_counts));
  std::vector<int32_t> h_counts(size_t(H), 0);
  PQ_HIP(hipMemcpyAsync(h_counts.data(), d_counts, size_t(H) * 4, hipMemcpyDeviceToHost,
                        c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  // first hypothesis with a strictly larger count wins (best starts empty)
  int64_t best = -1;
  int32_t best_cnt = 0;
  for (int64_t h = 0; h < H; ++h)
    if (h_counts[size_t(h)] > best_cnt) {
      best_cnt = h_counts[size_t(h)];
      best = h;
    }
  if (best < 0) return 0;
  {
    ProfScope ps(c, "ransac_compact");
    const dim3 grid(ceil_div(n + 1, 256));
    if (shape == 0)
      hipLaunchKernelGGL(k_flags<0>, grid, dim3(256), 0, c->stream, d_pts, n, d_models, best,
                         thresh, d_flags);
    else
      hipLaunchKernelGGL(k_flags<1>, grid, dim3(256), 0, c->stream, d_pts, n, d_models, best,
                         thresh, d_flags);
    PQ_HIP(hipGetLastError());
    PQ_TRY(exclusive_scan_i32(c, d_flags, n + 1));
    hipLaunchKernelGGL(k_compact, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, d_pts,
                       d_models, best, thresh, shape, d_flags, d_out);
    PQ_HIP(hipGetLastError());
  }
  Model m;
  PQ_HIP(hipMemcpyAsync(&m, d_models + best, sizeof(Model), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(inliers, d_out, size_t(best_cnt) * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  center[0] = m.cx; center[1] = m.cy; center[2] = m.cz;
  axis[0] = m.ax; axis[1] = m.ay; axis[2] = m.az;
  *radius = m.r;
  *n_inliers = best_cnt;
  if (best_out) *best_out = best;
  return 0;
}

// Many independent fits in one call: S point sets stacked in pts (seg_start int64 [S+1], from 0 to n),
// H hypotheses per set (triples int64 [S,H,3], indices LOCAL to the set). Per set the result of
// pyqsm_ransac: the first hypothesis with the largest inlier count (best = -1 and zeros when no
// hypothesis has an inlier), its model, and its inliers as ascending local indices, written set
// by set into `inliers` (n_inliers [S] says how many each set has).
int pyqsm_ransac_batch(const double* pts, int64_t n, const int64_t* seg_start, int64_t n_seg,
                       const int64_t* triples, int64_t H, int32_t shape, double thresh, double* centers,
                       double* axes, double* radii, int64_t* inliers, int64_t* n_inliers, int64_t* best_out,
                       int32_t device) {
  PQ_API_RANGE("pyqsm_ransac_batch");
  PQ_TRY(check_args(pts, n, H, shape));
  if (n_seg < 0) return fail(PYQSM_EINVAL, "negative number of sets");
  if (n_seg == 0) return 0;
  if (!seg_start || !centers || !axes || !radii || !n_inliers || !best_out || (n > 0 && !inliers))
    return fail(PYQSM_EINVAL, "pyqsm_ransac_batch: NULL pointer");
  if (seg_start[0] != 0 || seg_start[n_seg] != n) return fail(PYQSM_EINVAL, "seg_start must run from 0 to n");
  for (int64_t s = 0; s < n_seg; ++s)
    if (seg_start[s + 1] < seg_start[s]) return fail(PYQSM_EINVAL, "seg_start must not decrease");
  for (int64_t s = 0; s < n_seg; ++s) {
    n_inliers[s] = 0;
    best_out[s] = -1;
    radii[s] = 0.0;
    for (int a = 0; a < 3; ++a) centers[3 * s + a] = axes[3 * s + a] = 0.0;
  }
  if (n == 0 || H == 0) return 0;
  if (!triples) return fail(PYQSM_EINVAL, "pyqsm_ransac_batch: triples is NULL");
  if (double(n_seg) * double(H) > 2.0e9) return fail(PYQSM_ERANGE, "more than 2^31 hypotheses per call");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  const int64_t S = n_seg, SH = S * H;
  // the tiles of all sets
  std::vector<int32_t> h_tile_set;
  std::vector<int64_t> h_tile_base;
  for (int64_t s = 0; s < S; ++s)
    for (int64_t b = seg_start[s]; b < seg_start[s + 1]; b += kTile) {
      h_tile_set.push_back(int32_t(s));
      h_tile_base.push_back(b);
    }
  const int64_t tiles = int64_t(h_tile_set.size());
  if (tiles > 65535) return fail(PYQSM_ERANGE, "more than 65535 point tiles per call: split the batch");
  double* d_pts;
  int64_t *d_seg, *d_tri, *d_tile_base, *d_best, *d_out;
  int32_t *d_tile_set, *d_counts, *d_best_cnt, *d_flags, *d_set_of;
  Model *d_models, *d_win;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_pts));
  PQ_TRY(c->arena.get(size_t(S) + 1, &d_seg));
  PQ_TRY(c->arena.get(size_t(SH) * 3, &d_tri));
  PQ_TRY(c->arena.get(size_t(tiles), &d_tile_base));
  PQ_TRY(c->arena.get(size_t(tiles), &d_tile_set));
  PQ_TRY(c->arena.get(size_t(SH), &d_models));
  PQ_TRY(c->arena.get(size_t(SH), &d_counts));
  PQ_TRY(c->arena.get(size_t(S), &d_best));
  PQ_TRY(c->arena.get(size_t(S), &d_best_cnt));
  PQ_TRY(c->arena.get(size_t(S), &d_win));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_flags));
  PQ_TRY(c->arena.get(size_t(n), &d_set_of));
  PQ_TRY(c->arena.get(size_t(n), &d_out));
  PQ_HIP(hipMemcpyAsync(d_pts, pts, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_seg, seg_start, (size_t(S) + 1) * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_tri, triples, size_t(SH) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_tile_base, h_tile_base.data(), size_t(tiles) * 8, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_tile_set, h_tile_set.data(), size_t(tiles) * 4, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemsetAsync(d_counts, 0, size_t(SH) * 4, c->stream));
  {
    ProfScope ps(c, "ransac_models");
    hipLaunchKernelGGL(k_models_seg, dim3(ceil_div(SH, 256)), dim3(256), 0, c->stream, d_pts, d_seg, S, d_tri, H,
                       d_models);
    PQ_HIP(hipGetLastError());
  }
  {
    ProfScope ps(c, "ransac_count");
    const dim3 grid(ceil_div(H, 256), unsigned(tiles));
    if (shape == 0)
      hipLaunchKernelGGL(k_count_seg<0>, grid, dim3(256), 0, c->stream, d_pts, d_seg, d_tile_set, d_tile_base,
                         d_models, H, thresh, d_counts);
    else
      hipLaunchKernelGGL(k_count_seg<1>, grid, dim3(256), 0, c->stream, d_pts, d_seg, d_tile_set, d_tile_base,
                         d_models, H, thresh, d_counts);
    PQ_HIP(hipGetLastError());
  }
  {
    ProfScope ps(c, "ransac_compact");
    hipLaunchKernelGGL(k_best_seg, dim3(unsigned(S)), dim3(256), 0, c->stream, d_counts, H, d_best, d_best_cnt);
    const dim3 grid(ceil_div(n + 1, 256));
    if (shape == 0)
      hipLaunchKernelGGL(k_flags_seg<0>, grid, dim3(256), 0, c->stream, d_pts, n, d_seg, S, d_models, H, d_best,
                         thresh, d_flags, d_set_of);
    else
      hipLaunchKernelGGL(k_flags_seg<1>, grid, dim3(256), 0, c->stream, d_pts, n, d_seg, S, d_models, H, d_best,
                         thresh, d_flags, d_set_of);
    PQ_HIP(hipGetLastError());
    PQ_TRY(exclusive_scan_i32(c, d_flags, n + 1));
    hipLaunchKernelGGL(k_compact_seg, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, d_seg, d_set_of, d_flags,
                       d_out);
    hipLaunchKernelGGL(k_gather_best, dim3(ceil_div(S, 256)), dim3(256), 0, c->stream, S, d_models, H, d_best,
                       d_win);
    PQ_HIP(hipGetLastError());
  }
  std::vector<Model> h_win(static_cast<size_t>(S), Model{0, 0, 0, 0, 0, 0, 0, 0});
  std::vector<int32_t> h_cnt(static_cast<size_t>(S), 0);
  int32_t total = 0;
  PQ_HIP(hipMemcpyAsync(h_win.data(), d_win, size_t(S) * sizeof(Model), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(h_cnt.data(), d_best_cnt, size_t(S) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(best_out, d_best, size_t(S) * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(&total, d_flags + n, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (total > 0) {
    PQ_HIP(hipMemcpyAsync(inliers, d_out, size_t(total) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
  }
  for (int64_t s = 0; s < S; ++s) {
    const Model& m = h_win[size_t(s)];
    centers[3 * s] = m.cx; centers[3 * s + 1] = m.cy; centers[3 * s + 2] = m.cz;
    axes[3 * s] = m.ax; axes[3 * s + 1] = m.ay; axes[3 * s + 2] = m.az;
    radii[s] = m.r;
    n_inliers[s] = h_cnt[size_t(s)];
  }
  return 0;
}

}  // extern "C"
