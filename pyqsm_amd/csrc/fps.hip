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
//
// Large clouds (>= kPruneMinPoints) take the PRUNED rounds instead: the points are sorted into
// buckets of at most 256 (cells of 1/64 of the extent, cut into runs), every bucket keeps its
// bounding box and its own (largest distance, lowest index), and a round touches only the
// buckets that the new sample can still improve — those whose box is closer to it than the
// bucket's own largest distance. After the first few hundred samples that is a few dozen buckets
// instead of the whole cloud, and a round costs its launch and three dependent memory round
// trips: 1 M -> 100 k samples in 0.60 s instead of 0.89 s (5.9 us a round; the whole-cloud round
// moves 32 MB in 8.9 us). Same distances, same comparison, same indices.
#include "common.hpp"
#include "grid.hpp"

#include <chrono>
#include <cmath>

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

// ---- pruned rounds -------------------------------------------------------------------------
// Buckets: the points sorted into cells of 1/64 of the extent, every cell cut into runs of at most
// kBucket points; a bucket knows its bounding box (of its actual points) and its
// (largest distance, lowest original index). A sample can improve a point p only if
// |p - s|^2 < dist[p], and |p - s|^2 >= the squared distance from s to the bucket's box — rounded
// the same way, term by term — so a bucket whose box is no closer than its largest distance is
// left alone, exactly.

static constexpr int64_t kPruneMinPoints = 65536;
static constexpr int kPruneMinSamples = 512;
static constexpr int kPruneLook = 128;  // rounds between two looks at the current largest distance
static constexpr int kBucket = 256;     // points per bucket at most: four per lane of the wave that updates it

// a bucket's (or a block's) farthest point and where it sits in the sorted arrays. (Carrying its
// coordinates along as well — 40-byte records, one look-up less per round — was slower: 0.77 s
// against 0.65 s for 100 k samples of a million points.)
struct Far {
  double d;
  int idx, pos;
};

__global__ __launch_bounds__(256) void k_fps_bcount(int ncell, const int32_t* __restrict__ cstart,
                                                    int32_t* __restrict__ nbk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > ncell) return;
  nbk[c] = c < ncell ? (cstart[c + 1] - cstart[c] + kBucket - 1) / kBucket : 0;
}

__global__ __launch_bounds__(256) void k_fps_bfill(int ncell, int n, const int32_t* __restrict__ cstart,
                                                   const int32_t* __restrict__ base, int32_t* __restrict__ bstart) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ncell) return;
  const int q0 = cstart[c], q1 = cstart[c + 1];
  int b = base[c];
  for (int q = q0; q < q1; q += kBucket) bstart[b++] = q;
  if (c == ncell - 1) bstart[base[ncell]] = n;
}

// a wave per bucket: bounding box, (inf, lowest original index); pos_of and dist for its points
__global__ __launch_bounds__(256) void k_fps_binit(int nb, const int32_t* __restrict__ bstart,
                                                   const int32_t* __restrict__ order,
                                                   const double* __restrict__ sx, const double* __restrict__ sy,
                                                   const double* __restrict__ sz, double* __restrict__ aabb,
                                                   Far* __restrict__ val, int32_t* __restrict__ pos_of,
                                                   double* __restrict__ dist) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
  double hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
  int first = 0x7FFFFFFF, fpos = 0;
  for (int i = bstart[b] + lane; i < bstart[b + 1]; i += 64) {
    const double v[3] = {sx[i], sy[i], sz[i]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], v[a]);
      hi[a] = fmax(hi[a], v[a]);
    }
    const int id = order[i];
    if (id < first) {
      first = id;
      fpos = i;
    }
    pos_of[id] = i;
    dist[i] = __builtin_inf();
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_xor(hi[a], off, 64));
    }
    const int of = __shfl_xor(first, off, 64), op = __shfl_xor(fpos, off, 64);
    if (of < first) {
      first = of;
      fpos = op;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      aabb[6 * size_t(b) + a] = lo[a];
      aabb[6 * size_t(b) + 3 + a] = hi[a];
    }
    val[b] = Far{__builtin_inf(), first, fpos};
  }
}

__device__ __forceinline__ Far wave_best(Far m) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Far o;
    o.d = __shfl_xor(m.d, off, 64);
    o.idx = __shfl_xor(m.idx, off, 64);
    o.pos = __shfl_xor(m.pos, off, 64);
    if (better(o.d, o.idx, m.d, m.idx)) m = o;
  }
  return m;
}

// the best of a block of four waves, in every thread; one barrier (`red4` is not reused by the caller)
__device__ __forceinline__ Far block_best(Far v, Far* red4) {
  v = wave_best(v);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  Far r = red4[0];
#pragma unroll
  for (int q = 1; q < 4; ++q)
    if (better(red4[q].d, red4[q].idx, r.d, r.idx)) r = red4[q];
  return r;
}

// One round. Every block folds the previous launch's per-block maxima into the new sample, then
// takes the buckets blockIdx, blockIdx + G, ...: a bucket the sample cannot improve is copied to
// `next`, the others are updated, a wave per bucket; the block's own maximum over all its buckets
// goes to `pnext`. `cur` is only read and `next` only written during a launch, so fast and slow
// blocks see the same state. A round is a chain of dependent memory round trips, not work: the
// buckets' records are fetched before the fold (they do not depend on it), the sample's sorted
// position travels with the maxima, and the reductions take one barrier each.
__global__ __launch_bounds__(256) void k_fps_pruned(int s, int start_idx, int nb,
                                                    const int32_t* __restrict__ bstart,
                                                    const double* __restrict__ aabb,
                                                    const int32_t* __restrict__ order,
                                                    const int32_t* __restrict__ pos_of,
                                                    const double* __restrict__ sx,
                                                    const double* __restrict__ sy,
                                                    const double* __restrict__ sz,
                                                    double* __restrict__ dist,
                                                    const Far* __restrict__ cur, Far* __restrict__ next,
                                                    const Far* __restrict__ pprev, int nprev,
                                                    Far* __restrict__ pnext, int32_t* __restrict__ out,
                                                    double* __restrict__ dlog) {
  __shared__ Far red_a[4], red_b[4];
  __shared__ Far res[256];
  __shared__ int work[256], work_q0[256], work_q1[256];
  __shared__ int nwork;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x;
  // this thread's first bucket: fetched now, looked at after the fold
  int k = int(blockIdx.x) + G * tid;
  Far bm{-1.0, 0x7FFFFFFF, 0};
  double bx[6] = {0, 0, 0, 0, 0, 0};
  int q0 = 0, q1 = 0;
  if (k < nb) {
    bm = cur[k];
#pragma unroll
    for (int a = 0; a < 6; ++a) bx[a] = aabb[6 * size_t(k) + a];
    q0 = bstart[k];
    q1 = bstart[k + 1];
  }
  Far b{-1.0, 0x7FFFFFFF, 0};
  if (s == 0) {
    if (tid == 0) b = Far{__builtin_inf(), start_idx, pos_of[start_idx]};
  } else {
    for (int q = tid; q < nprev; q += 256) {
      const Far c = pprev[q];
      if (better(c.d, c.idx, b.d, b.idx)) b = c;
    }
  }
  if (tid == 0) nwork = 0;
  const Far top = block_best(b, red_a);
  if (blockIdx.x == 0 && tid == 0) {
    out[s] = top.idx;
    dlog[s] = top.d;
  }
  const double px = sx[top.pos], py = sy[top.pos], pz = sz[top.pos];
  Far mine{-1.0, 0x7FFFFFFF, 0};
  for (int j0 = 0; int(blockIdx.x) + G * j0 < nb; j0 += 256) {
    if (j0 > 0) {  // (never at the grid sizes the host picks: nb <= 256 G)
      k = int(blockIdx.x) + G * (j0 + tid);
      if (k < nb) {
        bm = cur[k];
#pragma unroll
        for (int a = 0; a < 6; ++a) bx[a] = aabb[6 * size_t(k) + a];
        q0 = bstart[k];
        q1 = bstart[k + 1];
      }
    }
    int slot = -1;
    if (k < nb) {
      // distance to the box, with the subtraction, the squares and the sum in the order of the
      // point distance below (each is monotone under rounding)
      auto axis = [](double p, double lo, double hi) { return p < lo ? lo - p : (p > hi ? p - hi : 0.0); };
      const double t0 = axis(px, bx[0], bx[3]), t1 = axis(py, bx[1], bx[4]), t2 = axis(pz, bx[2], bx[5]);
      double m2 = t0 * t0;
      m2 = m2 + t1 * t1;
      m2 = m2 + t2 * t2;
      if (m2 < bm.d) {
        slot = atomicAdd(&nwork, 1);
        work[slot] = k;
        work_q0[slot] = q0;
        work_q1[slot] = q1;
      } else {
        next[k] = bm;
        if (better(bm.d, bm.idx, mine.d, mine.idx)) mine = bm;
      }
    }
    __syncthreads();
    const int nw = nwork;
    for (int w = wave; w < nw; w += 4) {
      const int wq0 = work_q0[w], wq1 = work_q1[w];
      // at most four points per lane: all loads first
      double vx[4], vy[4], vz[4], old[4];
      int id[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = wq0 + lane + 64 * u;
        const int ii = i < wq1 ? i : wq0;
        vx[u] = sx[ii];
        vy[u] = sy[ii];
        vz[u] = sz[ii];
        old[u] = dist[ii];
        id[u] = order[ii];
      }
      Far m{-1.0, 0x7FFFFFFF, 0};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = wq0 + lane + 64 * u;
        if (i < wq1) {
          const double t0 = vx[u] - px, t1 = vy[u] - py, t2 = vz[u] - pz;
          double d = t0 * t0;
          d = d + t1 * t1;
          d = d + t2 * t2;
          d = d < old[u] ? d : old[u];
          dist[i] = d;
          if (better(d, id[u], m.d, m.idx)) m = Far{d, id[u], i};
        }
      }
      m = wave_best(m);
      if (lane == 0) {
        next[work[w]] = m;
        res[w] = m;
      }
    }
    __syncthreads();
    if (slot >= 0) {
      const Far m = res[slot];
      if (better(m.d, m.idx, mine.d, mine.idx)) mine = m;
    }
    if (int(blockIdx.x) + G * (j0 + 256) < nb) {  // another pass: the list starts empty again
      __syncthreads();
      if (tid == 0) nwork = 0;
      __syncthreads();
    }
  }
  const Far mb = block_best(mine, red_b);
  if (tid == 0) pnext[blockIdx.x] = mb;
}

static bool fps_prune_enabled() {  // PYQSM_FPS_PRUNE=0: the whole-cloud rounds at every size
  const char* e = getenv("PYQSM_FPS_PRUNE");
  return !(e && e[0] == '0');
}

// d_xyz on the device; d_out [S]
static int fps_pruned(Ctx* c, const double* d_xyz, int N, int S, int start_index, int32_t* d_out, bool* done) {
  *done = false;
  double mn[3], mx[3];
  PQ_TRY(cloud_bbox(c, d_xyz, N, mn, mx));
  const double ext = std::max(mx[0] - mn[0], std::max(mx[1] - mn[1], mx[2] - mn[2]));
  if (!(ext > 0.0) || !std::isfinite(ext)) return 0;
  double box[6] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]};
  DevGrid g;
  PQ_TRY(build_grid(c, d_xyz, N, ext / 64.0, 300000, &g, box));
  const int ncell = int(g.ncell);
  int32_t *nbk, *bstart, *pos_of;
  double *dist, *dlog, *aabb;
  PQ_TRY(c->arena.get(size_t(ncell) + 1, &nbk));
  hipLaunchKernelGGL(k_fps_bcount, dim3(ceil_div(ncell + 1, 256)), dim3(256), 0, c->stream, ncell, g.start, nbk);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, nbk, int64_t(ncell) + 1));
  int32_t nb = 0;
  PQ_HIP(hipMemcpyAsync(&nb, nbk + ncell, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (nb <= 0) return 0;
  static constexpr int kMaxBlocks = 1024;
  Far *va, *vb, *pa, *pb;
  PQ_TRY(c->arena.get(size_t(nb) + 1, &bstart));
  PQ_TRY(c->arena.get(size_t(nb) * 6, &aabb));
  PQ_TRY(c->arena.get(size_t(nb), &va));
  PQ_TRY(c->arena.get(size_t(nb), &vb));
  PQ_TRY(c->arena.get(size_t(kMaxBlocks), &pa));
  PQ_TRY(c->arena.get(size_t(kMaxBlocks), &pb));
  PQ_TRY(c->arena.get(size_t(N), &pos_of));
  PQ_TRY(c->arena.get(size_t(N), &dist));
  PQ_TRY(c->arena.get(size_t(S), &dlog));
  hipLaunchKernelGGL(k_fps_bfill, dim3(ceil_div(ncell, 256)), dim3(256), 0, c->stream, ncell, N, g.start, nbk, bstart);
  hipLaunchKernelGGL(k_fps_binit, dim3(ceil_div(nb, 4)), dim3(256), 0, c->stream, int(nb), bstart, g.order, g.sx,
                     g.sy, g.sz, aabb, va, pos_of, dist);
  PQ_HIP(hipGetLastError());
  ProfScope ps(c, "fps_rounds", S);
  bool late = false;  // the sample's reach is down to a cell or two: few buckets per round
  int nprev = 0;
  const auto t_loop = std::chrono::steady_clock::now();
  for (int s = 0; s < S; ++s) {
    if (!late && s > 0 && s % kPruneLook == 0) {
      double D = 0.0;
      PQ_HIP(hipMemcpyAsync(&D, dlog + (s - 1), 8, hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      if (std::sqrt(D) < 3.0 * g.cell) late = true;
    }
    // early: every bucket is in reach, one per wave; late: a block looks at 64 buckets' boxes
    // (late: 256 / 128 / 64 / 32 / 16 buckets per block -> 0.654 / 0.611 / 0.596 / 0.624 / 0.708 s for 100 k samples)
    // (PYQSM_FPS_MAX_BLOCKS: a test hook — a small grid makes every block take several passes over its
    // buckets, the path a cloud of more than 67 M points takes)
    const char* mbe = getenv("PYQSM_FPS_MAX_BLOCKS");
    const int max_blocks = mbe && atoi(mbe) > 0 ? std::min(atoi(mbe), kMaxBlocks) : kMaxBlocks;
    const int blocks = std::max(1, std::min(max_blocks, ceil_div(int(nb), late ? 64 : 4)));
    const Far* cur = (s & 1) ? vb : va;
    Far* next = (s & 1) ? va : vb;
    const Far* pprev = (s & 1) ? pb : pa;
    Far* pnext = (s & 1) ? pa : pb;
    hipLaunchKernelGGL(k_fps_pruned, dim3(blocks), dim3(256), 0, c->stream, s, start_index, int(nb),
                       static_cast<const int32_t*>(bstart), static_cast<const double*>(aabb),
                       static_cast<const int32_t*>(g.order), static_cast<const int32_t*>(pos_of),
                       static_cast<const double*>(g.sx), static_cast<const double*>(g.sy),
                       static_cast<const double*>(g.sz), dist, cur, next, pprev, nprev, pnext, d_out, dlog);
    nprev = blocks;
  }
  PQ_HIP(hipGetLastError());
  if (getenv("PYQSM_FPS_TRACE")) {
    const double issue = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    PQ_HIP(hipStreamSynchronize(c->stream));
    const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    fprintf(stderr, "fps: %d buckets, %d rounds issued in %.3f s, finished after %.3f s\n", int(nb), S, issue, all);
  }
  *done = true;
  return 0;
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
  bool pruned = false;
  if (n >= kPruneMinPoints && S >= kPruneMinSamples && fps_prune_enabled()) {
    PQ_TRY(fps_pruned(c, d_xyz, N, S, int(start_index), d_out, &pruned));
  }
  const int blocks = int(std::min<int64_t>(ceil_div(n, 256), kFpsBlocks));
  if (!pruned) {
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
