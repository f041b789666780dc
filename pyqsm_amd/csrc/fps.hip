// fps.hip — farthest-point down-sampling on gfx950.
//
// Stands in for open3d PointCloud.farthest_point_down_sample(num_samples) as
// called by extract_topology (pyQSM/geometry/skeletonize.py:127-132), which keeps
// 10 % of the contracted cloud: O(N * S) work, the first step after the
// contraction loop.
//
// Algorithm (the textbook one Open3D implements): start from point `start`; keep
// for every point the squared distance to the nearest selected point; select the
// point with the largest such distance (lowest index on ties); repeat.
// One kernel per sample: every block first folds the previous round's per-block
// maxima (so no separate reduction launch), then updates its slice of the distance
// array against the newly selected point and publishes its own maximum.
// HBM-bound: 32 B per point per sample (24 B coordinates + 8 B distance, read and
// written once); at 1 M points the arrays live in the Infinity Cache.
#include "common.hpp"

namespace pyqsm {

static constexpr int kFpsBlocks = 1024;

struct Best {
  double d;
  int idx;
};

__device__ __forceinline__ bool better(double d, int i, double bd, int bi) {
  return d > bd || (d == bd && i < bi);  // farthest first, lowest index on ties
}

// SoA copy of the cloud
__global__ __launch_bounds__(256) void k_fps_split(int n, const double* __restrict__ xyz,
                                                   double* __restrict__ x, double* __restrict__ y,
                                                   double* __restrict__ z,
                                                   double* __restrict__ dist) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  x[i] = xyz[3 * i];
  y[i] = xyz[3 * i + 1];
  z[i] = xyz[3 * i + 2];
  dist[i] = __builtin_inf();
}

// round s: fold the partial maxima of round s-1 -> selected point; record it;
// update distances; publish this block's maximum into the other partial buffer.
__global__ __launch_bounds__(256) void k_fps_round(int n, int s, int start,
                                                   const double* __restrict__ x,
                                                   const double* __restrict__ y,
                                                   const double* __restrict__ z,
                                                   double* __restrict__ dist,
                                                   const Best* __restrict__ prev, int nprev,
                                                   Best* __restrict__ next,
                                                   int32_t* __restrict__ out) {
  __shared__ Best red[256];
  __shared__ int sel_s;
  // 1. every block derives the same selected index from the previous partials
  Best b{-1.0, 0x7FFFFFFF};
  if (s == 0) {
    if (threadIdx.x == 0) b = Best{0.0, start};
  } else {
    for (int k = threadIdx.x; k < nprev; k += 256) {
      const Best c = prev[k];
      if (better(c.d, c.idx, b.d, b.idx)) b = c;
    }
  }
  red[threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const Best c = red[threadIdx.x + off];
      if (better(c.d, c.idx, red[threadIdx.x].d, red[threadIdx.x].idx)) red[threadIdx.x] = c;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    sel_s = red[0].idx;
    if (blockIdx.x == 0) out[s] = red[0].idx;
  }
  __syncthreads();
  const int sel = sel_s;
  const double px = x[sel], py = y[sel], pz = z[sel];
  // 2. update my slice and find its farthest point
  Best m{-1.0, 0x7FFFFFFF};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const double t0 = x[i] - px, t1 = y[i] - py, t2 = z[i] - pz;
    double d = t0 * t0;
    d = d + t1 * t1;
    d = d + t2 * t2;
    const double old = dist[i];
    d = d < old ? d : old;
    dist[i] = d;
    if (better(d, i, m.d, m.idx)) m = Best{d, i};
  }
  __syncthreads();
  red[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      const Best c = red[threadIdx.x + off];
      if (better(c.d, c.idx, red[threadIdx.x].d, red[threadIdx.x].idx)) red[threadIdx.x] = c;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) next[blockIdx.x] = red[0];
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_fps(const double* xyz, int64_t n, int64_t num_samples, int64_t start_index,
              int32_t* out_idx, int32_t device) {
  PQ_API_RANGE("pyqsm_fps");
  if (n < 0 || num_samples < 0) return fail(PYQSM_EINVAL, "negative size");
  if (num_samples == 0) return 0;
  if (num_samples > n) return fail(PYQSM_EINVAL, "num_samples exceeds the number of points");
  if (start_index < 0 || start_index >= n) return fail(PYQSM_EINVAL, "start_index out of range");
  if (!xyz || !out_idx) return fail(PYQSM_EINVAL, "pyqsm_fps: NULL pointer");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points per call");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  const int N = int(n), S = int(num_samples);
  double *d_xyz, *d_x, *d_y, *d_z, *d_dist;
  Best *d_pa, *d_pb;
  int32_t* d_out;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(size_t(n), &d_x));
  PQ_TRY(c->arena.get(size_t(n), &d_y));
  PQ_TRY(c->arena.get(size_t(n), &d_z));
  PQ_TRY(c->arena.get(size_t(n), &d_dist));
  PQ_TRY(c->arena.get(size_t(kFpsBlocks), &d_pa));
  PQ_TRY(c->arena.get(size_t(kFpsBlocks), &d_pb));
  PQ_TRY(c->arena.get(size_t(S), &d_out));
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_fps_split, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, N, d_xyz, d_x,
                     d_y, d_z, d_dist);
  const int blocks = int(std::min<int64_t>(ceil_div(n, 256), kFpsBlocks));
  {
    ProfScope ps(c, "fps_rounds", S);
    for (int s = 0; s < S; ++s) {
      Best* prev = (s & 1) ? d_pa : d_pb;
      Best* next = (s & 1) ? d_pb : d_pa;
      hipLaunchKernelGGL(k_fps_round, dim3(blocks), dim3(256), 0, c->stream, N, s,
                         int(start_index), d_x, d_y, d_z, d_dist, prev, blocks, next, d_out);
    }
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(out_idx, d_out, size_t(S) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
