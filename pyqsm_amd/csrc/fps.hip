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

#include <atomic>
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

// ---- the late rounds in ONE launch ------------------------------------------------------------
// Once the samples' reach is down to a cell or two, a round touches a handful of points: what it
// costs as a launch (5.9 us) is the launch and three dependent round trips across the chip, 100 000
// times. k_fps_tail runs all remaining rounds inside ONE workgroup of four waves — no grid barrier,
// nothing that could wait for another block — on a hierarchy that is compact against the reach:
//   the points in Morton order of a 1024^3 grid over the cloud's box, 64 consecutive points = a BUCKET
//   (bounding box + farthest point, in memory as separate arrays: a wave's gather is eleven contiguous
//   loads), 16 consecutive buckets = a GROUP with box and farthest point in LDS; every thread OWNS up to
//   five groups and keeps their boxes in registers. A round: every thread the best of its groups, a DPP
//   fold per wave, the four winners (with their coordinates: no look-up) through LDS -> the sample ->
//   groups whose box is closer than their largest distance (registers) -> their buckets' records and
//   boxes, ONE gather, kept in LDS -> the buckets the sample can improve are updated, a wave each and a
//   point per lane, the loads of four buckets in flight together; a bucket's farthest point changes only
//   if that very point moved closer, otherwise its record stands and no reduction runs -> the groups that
//   changed fold their sixteen records from LDS. Two dependent memory round trips per round and five
//   barriers of four waves.
// Every box test is exact for the same reason the bucket test of k_fps_pruned is: a parent's box contains
// its children's boxes and its maximum bounds theirs, and every term is rounded monotonically, so a parent
// that fails the test holds no child that would pass. Same distances, same comparison, same indices as
// the launched rounds and as the whole-cloud rounds (tests/test_gpu_topology.py).
// Measured on the way (1 M points, 100 k samples; a launch per round: 5.95 us): sixteen waves with
// 256-point buckets in cell order and 64-bucket groups, 40-byte records through __shfl_xor butterflies:
// 10.3 us a round (a butterfly of a 40-byte record is 60 ds_bpermute: 2 600 cycles with sixteen waves at
// it); DPP row reductions + v_readlane: 6.5 us, of which 5 us one CU's bandwidth (50 KB of bucket records
// and 130 KB of points per round); Morton-ordered 64-point buckets, four waves: 8.2 us (array-of-struct
// records: 250 cache-line requests per gather, buckets updated one after the other); records as separate
// arrays, 16-bucket groups, four buckets' loads in flight: 6.2 us; supers, dirty tracking and the skipped
// reductions: 5.3 us; the supers dropped again for the flat scan with register-resident group boxes: 4.7 us
// (argmax 1.0, group test 0.85, bucket gather 0.85, update 1.5, fold 0.45: the first two and the last
// touch no memory at all — with ONE wave per SIMD every dependent instruction costs its full latency,
// ~250 instructions of compare / DPP / select are a microsecond; the other two are one memory trip each).
static constexpr int kTailThreads = 256;     // four waves: barriers and reductions are the fixed cost of a round
static constexpr int kTailWaves = kTailThreads / 64;
static constexpr int kTailBucket = 64;       // points per bucket in the tail: a lane each
static constexpr int kTailGroup = 16;        // buckets per group: a wave takes four groups at a time
static constexpr int kTailPerWave = 64 / kTailGroup;
static constexpr int kTailMaxGroups = 1280;  // LDS: 88 bytes per group (1.3 M points; larger clouds keep the launches)
static constexpr int kTailOwn = kTailMaxGroups / 256;  // groups a thread owns (their boxes live in its registers)
static constexpr int kTailUnroll = 4;        // buckets a wave updates with their loads issued side by side

struct TailRec {  // a bucket's (or a group's) farthest point, with its coordinates
  double d;
  int idx, pos;
  double x, y, z;
};

// ---- the best of a wave without LDS traffic -----------------------------------------------------
// A butterfly of __shfl_xor over a 40-byte record is 60 ds_bpermute instructions (2 600 cycles with
// sixteen waves at it: measured, it was the largest part of a round). Instead: the maximum of the
// order-preserving 64-bit image of d by DPP row operations (four steps inside each row of 16 lanes,
// then four scalar reads, one per row), the lowest index among the lanes that hold it the same way,
// and the winner's other fields by v_readlane. No LDS, ~100 cycles.
template <int CTRL>
__device__ __forceinline__ unsigned dpp32(unsigned v) {
  return unsigned(__builtin_amdgcn_update_dpp(0, int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_max64(unsigned long long k) {
  const unsigned long long o = (static_cast<unsigned long long>(dpp32<CTRL>(unsigned(k >> 32))) << 32) |
                               dpp32<CTRL>(unsigned(k));
  return o > k ? o : k;
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k) {  // all 64 lanes active
  k = dpp_max64<0xB1>(k);   // quad_perm [1,0,3,2]
  k = dpp_max64<0x4E>(k);   // quad_perm [2,3,0,1]
  k = dpp_max64<0x124>(k);  // row_ror:4
  k = dpp_max64<0x128>(k);  // row_ror:8 -> every lane holds its row's maximum
  auto rl = [&](int lane) {
    return (static_cast<unsigned long long>(unsigned(__builtin_amdgcn_readlane(int(k >> 32), lane))) << 32) |
           unsigned(__builtin_amdgcn_readlane(int(unsigned(k)), lane));
  };
  const unsigned long long a = rl(0), b = rl(16), c = rl(32), d = rl(48);
  const unsigned long long ab = a > b ? a : b, cd = c > d ? c : d;
  return ab > cd ? ab : cd;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned k) {
  auto step = [](unsigned v, unsigned o) { return o > v ? o : v; };
  k = step(k, dpp32<0xB1>(k));
  k = step(k, dpp32<0x4E>(k));
  k = step(k, dpp32<0x124>(k));
  k = step(k, dpp32<0x128>(k));
  const unsigned a = unsigned(__builtin_amdgcn_readlane(int(k), 0)), b = unsigned(__builtin_amdgcn_readlane(int(k), 16)),
                 c = unsigned(__builtin_amdgcn_readlane(int(k), 32)), d = unsigned(__builtin_amdgcn_readlane(int(k), 48));
  return step(step(a, b), step(c, d));
}
__device__ __forceinline__ unsigned long long ord_bits(double x) {  // order-preserving double -> uint64
  const unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(x));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double lane_double(double v, int src) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = unsigned(__builtin_amdgcn_readlane(int(unsigned(b)), src));
  const unsigned hi = unsigned(__builtin_amdgcn_readlane(int(unsigned(b >> 32)), src));
  return __longlong_as_double((static_cast<long long>(hi) << 32) | lo);
}
// the best record of the wave (farthest, lowest index on ties), in every lane; all lanes active
__device__ __forceinline__ TailRec tail_best(TailRec m) {
  const unsigned long long kd = ord_bits(m.d);
  const unsigned long long top = wave_max_u64(kd);
  unsigned long long who = __ballot(kd == top);
  if (who & (who - 1)) {  // several lanes at the maximum (wave-uniform): the lowest index among them
    const unsigned ki = kd == top ? 0x7FFFFFFFu - unsigned(m.idx) : 0u;  // indices are below 2^31
    const unsigned ti = wave_max_u32(ki);
    who = __ballot(kd == top && ki == ti);
  }
  const int src = __builtin_amdgcn_readfirstlane(__ffsll(who) - 1);
  TailRec r;
  r.d = lane_double(m.d, src);
  r.idx = __builtin_amdgcn_readlane(m.idx, src);
  r.pos = __builtin_amdgcn_readlane(m.pos, src);
  r.x = lane_double(m.x, src);
  r.y = lane_double(m.y, src);
  r.z = lane_double(m.z, src);
  return r;
}

__device__ __forceinline__ unsigned spread3(unsigned v) {  // 10 bits -> every third bit
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

// Morton key of every point (10 bits per axis over the cloud's box): 64 consecutive points of the
// sorted order are a compact blob, 64 consecutive blobs a compact group
__global__ __launch_bounds__(256) void k_fps_tail_key(int n, const double* __restrict__ sx, const double* __restrict__ sy,
                                                      const double* __restrict__ sz, double mnx, double mny, double mnz,
                                                      double inv, uint32_t* __restrict__ key, int32_t* __restrict__ ident) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  auto q = [&](double v, double mn) { return unsigned(fmin(fmax((v - mn) * inv, 0.0), 1023.0)); };
  key[i] = spread3(q(sx[i], mnx)) | (spread3(q(sy[i], mny)) << 1) | (spread3(q(sz[i], mnz)) << 2);
  ident[i] = i;
}

// the points in their new order
__global__ __launch_bounds__(256) void k_fps_tail_points(int n, const int32_t* __restrict__ perm,
                                                         const double* __restrict__ sx, const double* __restrict__ sy,
                                                         const double* __restrict__ sz, const double* __restrict__ dist,
                                                         const int32_t* __restrict__ order, double* __restrict__ tx,
                                                         double* __restrict__ ty, double* __restrict__ tz,
                                                         double* __restrict__ td, int32_t* __restrict__ tid) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int i = perm[j];
  tx[j] = sx[i];
  ty[j] = sy[i];
  tz[j] = sz[i];
  td[j] = dist[i];
  tid[j] = order[i];
}

// bucket records as separate arrays: a wave's 64 records are eleven contiguous 512-byte loads (the
// 40- and 48-byte structs cost 250 cache-line requests per wave and ~1.1 us per gather)
struct TailSoA {
  double *d, *x, *y, *z;
  int32_t *idx, *pos;
  double* box[6];
};
__device__ __forceinline__ TailRec soa_load(const TailSoA& t, int b) {
  return TailRec{t.d[b], t.idx[b], t.pos[b], t.x[b], t.y[b], t.z[b]};
}
__device__ __forceinline__ void soa_store(const TailSoA& t, int b, const TailRec& m) {
  t.d[b] = m.d;
  t.idx[b] = m.idx;
  t.pos[b] = m.pos;
  t.x[b] = m.x;
  t.y[b] = m.y;
  t.z[b] = m.z;
}

// a wave per bucket of 64 consecutive points: bounding box and farthest point
__global__ __launch_bounds__(256) void k_fps_tail_buckets(int n, int nb, const double* __restrict__ tx,
                                                          const double* __restrict__ ty, const double* __restrict__ tz,
                                                          const double* __restrict__ td, const int32_t* __restrict__ tid,
                                                          TailSoA rec) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  const int j = b * kTailBucket + lane;
  const bool live = j < n;
  const int jj = live ? j : b * kTailBucket;
  const double x = tx[jj], y = ty[jj], z = tz[jj];
  TailRec m{live ? td[jj] : -1.0, live ? tid[jj] : 0x7FFFFFFF, jj, x, y, z};
  double lo[3] = {x, y, z}, hi[3] = {x, y, z};
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      lo[a] = fmin(lo[a], __shfl_xor(lo[a], off, 64));
      hi[a] = fmax(hi[a], __shfl_xor(hi[a], off, 64));
    }
  m = tail_best(m);
  if (lane == 0) {
    soa_store(rec, b, m);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      rec.box[a][b] = lo[a];
      rec.box[3 + a][b] = hi[a];
    }
  }
}

// reductions inside the rows of 16 lanes a group occupies (a wave holds four groups side by side)
__device__ __forceinline__ TailRec row_best(TailRec m) {
  auto kmax = [](unsigned long long k) {
    k = dpp_max64<0xB1>(k);
    k = dpp_max64<0x4E>(k);
    k = dpp_max64<0x124>(k);
    return dpp_max64<0x128>(k);
  };
  const unsigned long long kd = ord_bits(m.d);
  const unsigned long long top = kmax(kd);
  unsigned ki = kd == top ? 0x7FFFFFFFu - unsigned(m.idx) : 0u;
  {
    auto step = [](unsigned v, unsigned o) { return o > v ? o : v; };
    unsigned k = ki;
    k = step(k, dpp32<0xB1>(k));
    k = step(k, dpp32<0x4E>(k));
    k = step(k, dpp32<0x124>(k));
    k = step(k, dpp32<0x128>(k));
    ki = (kd == top && ki == k) ? 1u : 0u;  // this lane is its row's winner
  }
  // the winner's fields to every lane of its row: one ds_bpermute per dword from the winner's lane
  const unsigned long long who = __ballot(ki != 0);
  const int row = (threadIdx.x & 63) >> 4;
  const int src = __ffsll((who >> (16 * row)) & 0xFFFFull) - 1 + 16 * row;
  TailRec r;
  r.d = __shfl(m.d, src, 64);
  r.idx = __shfl(m.idx, src, 64);
  r.pos = __shfl(m.pos, src, 64);
  r.x = __shfl(m.x, src, 64);
  r.y = __shfl(m.y, src, 64);
  r.z = __shfl(m.z, src, 64);
  return r;
}

__global__ __launch_bounds__(kTailThreads) void k_fps_tail(
    int s_begin, int s_end, int n, int nb, int ng, TailSoA rec, const int32_t* __restrict__ tid,
    const double* __restrict__ tx, const double* __restrict__ ty, const double* __restrict__ tz,
    double* __restrict__ td, int32_t* __restrict__ out,
    unsigned long long* __restrict__ dbg /*diagnostic: phase clocks (100 MHz ticks) and counts, may be null*/) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // the groups' farthest points and boxes as separate arrays [ng] (a thread reads the same field of
  // consecutive groups: no bank conflicts), 88 bytes per group
  double* g_d = reinterpret_cast<double*>(smem);
  double* g_x = g_d + ng;
  double* g_y = g_x + ng;
  double* g_z = g_y + ng;
  double* g_box = g_z + ng;  // [6][ng]
  int* g_idx = reinterpret_cast<int*>(g_box + 6 * size_t(ng));
  int* g_pos = g_idx + ng;
  __shared__ TailRec grec[kTailWaves][64];  // the records of the groups of the current batch (four per wave)
  __shared__ double wb_d[kTailWaves];       // every wave's best group of the round
  __shared__ double wb_x[kTailWaves], wb_y[kTailWaves], wb_z[kTailWaves];
  __shared__ int wb_idx[kTailWaves];
  __shared__ int gdirty[kTailWaves * kTailPerWave];
  __shared__ int glist[kTailMaxGroups];
  __shared__ int2 blist[kTailWaves * 64];  // (bucket, slot in grec)
  __shared__ int n_g, n_b;
  const int tid_ = threadIdx.x, lane = tid_ & 63, wave = tid_ >> 6;
  const int sub = lane / kTailGroup, gl = lane % kTailGroup;  // group slot of the wave, bucket in the group
  const TailRec none{-1.0, 0x7FFFFFFF, 0, 0.0, 0.0, 0.0};
  auto axis = [](double p, double lo, double hi) { return p < lo ? lo - p : (p > hi ? p - hi : 0.0); };
  auto box_d2 = [&](double px, double py, double pz, const double* bx) {
    const double t0 = axis(px, bx[0], bx[3]), t1 = axis(py, bx[1], bx[4]), t2 = axis(pz, bx[2], bx[5]);
    double m2 = t0 * t0;
    m2 = m2 + t1 * t1;
    return m2 + t2 * t2;
  };
  auto put_group = [&](int g, const TailRec& m) {
    g_d[g] = m.d;
    g_idx[g] = m.idx;
    g_pos[g] = m.pos;
    g_x[g] = m.x;
    g_y[g] = m.y;
    g_z[g] = m.z;
  };
  // group records from the bucket records
  for (int g0 = wave * kTailPerWave; g0 < ng; g0 += kTailWaves * kTailPerWave) {
    const int g = g0 + sub;
    const int b = g * kTailGroup + gl;
    TailRec m = none;
    double lo[3] = {__builtin_inf(), __builtin_inf(), __builtin_inf()};
    double hi[3] = {-__builtin_inf(), -__builtin_inf(), -__builtin_inf()};
    if (g < ng && b < nb) {
      m = soa_load(rec, b);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = rec.box[a][b];
        hi[a] = rec.box[3 + a][b];
      }
    }
    m = row_best(m);
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lo[a] = fmin(lo[a], __shfl_xor(lo[a], off, 64));
        hi[a] = fmax(hi[a], __shfl_xor(hi[a], off, 64));
      }
    if (gl == 0 && g < ng) {
      put_group(g, m);
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        g_box[size_t(a) * ng + g] = lo[a];
        g_box[size_t(3 + a) * ng + g] = hi[a];
      }
    }
  }
  __syncthreads();
  // a thread OWNS the groups tid, tid + 256, ...: their boxes (which never change) stay in its registers
  double own_box[kTailOwn][6];
#pragma unroll
  for (int k = 0; k < kTailOwn; ++k) {
    const int g = tid_ + k * kTailThreads;
#pragma unroll
    for (int a = 0; a < 6; ++a) own_box[k][a] = g < ng ? g_box[size_t(a) * ng + g] : 0.0;
  }
  const unsigned long long clk0 = dbg ? clock64() : 0;
  for (int s = s_begin; s < s_end; ++s) {
    unsigned long long t0 = dbg ? wall_clock64() : 0;
    // 1. the farthest point of all: every thread the best of the groups it owns (their records read side
    //    by side, then compared in registers), a DPP fold per wave, the four waves' winners — with their
    //    coordinates — through LDS. (A level of 64-group "supers" with dirty flags above the groups made
    //    this 1.2 us of a round and the group test below another 1.2: a wave per super, 40-byte records
    //    out of LDS, two folds one after the other.)
    double od[kTailOwn];
    {
      int oi[kTailOwn];
#pragma unroll
      for (int k = 0; k < kTailOwn; ++k) {
        const int g = tid_ + k * kTailThreads;
        od[k] = g < ng ? g_d[g] : -1.0;
        oi[k] = g < ng ? g_idx[g] : 0x7FFFFFFF;
      }
      double bd = od[0];
      int bi = oi[0], bk = 0;
#pragma unroll
      for (int k = 1; k < kTailOwn; ++k)
        if (od[k] > bd || (od[k] == bd && oi[k] < bi)) {
          bd = od[k];
          bi = oi[k];
          bk = k;
        }
      const unsigned long long kd = ord_bits(bd);
      const unsigned long long topk = wave_max_u64(kd);
      unsigned long long who = __ballot(kd == topk);
      if (who & (who - 1)) {  // several lanes at the maximum (wave-uniform): the lowest index among them
        const unsigned ki = kd == topk ? 0x7FFFFFFFu - unsigned(bi) : 0u;
        const unsigned ti = wave_max_u32(ki);
        who = __ballot(kd == topk && ki == ti);
      }
      const int src = __builtin_amdgcn_readfirstlane(__ffsll(who) - 1);
      if (lane == src) {
        const int bg = min(tid_ + bk * kTailThreads, ng - 1);
        wb_d[wave] = bd;
        wb_idx[wave] = bi;
        wb_x[wave] = g_x[bg];
        wb_y[wave] = g_y[bg];
        wb_z[wave] = g_z[bg];
      }
      if (tid_ == 0) n_g = 0;
    }
    __syncthreads();
    double px, py, pz;
    {
      double wd[kTailWaves], wx[kTailWaves], wy[kTailWaves], wz[kTailWaves];
      int wi[kTailWaves];
#pragma unroll
      for (int w = 0; w < kTailWaves; ++w) {
        wd[w] = wb_d[w];
        wi[w] = wb_idx[w];
        wx[w] = wb_x[w];
        wy[w] = wb_y[w];
        wz[w] = wb_z[w];
      }
      double bd = wd[0];
      int bi = wi[0];
      px = wx[0], py = wy[0], pz = wz[0];
#pragma unroll
      for (int w = 1; w < kTailWaves; ++w)
        if (wd[w] > bd || (wd[w] == bd && wi[w] < bi)) {
          bd = wd[w];
          bi = wi[w];
          px = wx[w], py = wy[w], pz = wz[w];
        }
      if (tid_ == 0) out[s] = bi;
    }
    if (dbg) { const unsigned long long t = wall_clock64(); acc[0] += t - t0; t0 = t; }
    // 2. groups the sample can still improve: box closer than the group's largest distance — all in
    //    registers (own_box, od)
#pragma unroll
    for (int k = 0; k < kTailOwn; ++k) {
      const int g = tid_ + k * kTailThreads;
      const bool hit = g < ng && box_d2(px, py, pz, own_box[k]) < od[k];
      const unsigned long long mask = __ballot(hit);
      if (mask) {  // wave-uniform
        int base = 0;
        if (lane == 0) base = atomicAdd(&n_g, __popcll(mask));
        base = __shfl(base, 0, 64);
        if (hit) glist[base + __popcll(mask & ((1ull << lane) - 1ull))] = g;
      }
    }
    __syncthreads();
    const int ngl = n_g;
    if (dbg) { const unsigned long long t = wall_clock64(); acc[1] += t - t0; t0 = t; acc[5] += ngl; }
    for (int g0 = 0; g0 < ngl; g0 += kTailWaves * kTailPerWave) {  // batches: four groups per wave
      if (tid_ == 0) n_b = 0;
      if (tid_ < kTailWaves * kTailPerWave) gdirty[tid_] = 0;
      __syncthreads();
      // 3. the batch's buckets: record and box, one gather of contiguous arrays; the records stay in LDS
      const int q = g0 + wave * kTailPerWave + sub;
      {
        const int bk = q < ngl ? glist[q] * kTailGroup + gl : nb;
        TailRec r = none;
        bool hit = false;
        if (bk < nb) {
          r = soa_load(rec, bk);
          const double bx[6] = {rec.box[0][bk], rec.box[1][bk], rec.box[2][bk], rec.box[3][bk], rec.box[4][bk], rec.box[5][bk]};
          hit = box_d2(px, py, pz, bx) < r.d;
        }
        grec[wave][lane] = r;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&n_b, __popcll(mask));
          base = __shfl(base, 0, 64);
          if (hit) blist[base + __popcll(mask & ((1ull << lane) - 1ull))] = make_int2(bk, wave * 64 + lane);
        }
      }
      __syncthreads();
      // 4. update the listed buckets, a wave each and a point per lane; the loads of up to kTailUnroll
      //    buckets are issued before the first is used. A bucket's farthest point changes only if that very
      //    point moved closer (everybody else can only decrease): otherwise its record stands, no reduction.
      const int nbl = n_b;
      if (dbg) { const unsigned long long t = wall_clock64(); acc[2] += t - t0; t0 = t; acc[6] += nbl; }
      for (int w0 = wave; w0 < nbl; w0 += kTailWaves * kTailUnroll) {
        double vx[kTailUnroll], vy[kTailUnroll], vz[kTailUnroll], old[kTailUnroll];
        int id[kTailUnroll], jj[kTailUnroll];
        int2 job[kTailUnroll];
#pragma unroll
        for (int u = 0; u < kTailUnroll; ++u) {
          const int w = w0 + u * kTailWaves;
          job[u] = w < nbl ? blist[w] : make_int2(-1, 0);
          const int j = job[u].x >= 0 ? job[u].x * kTailBucket + lane : n;
          jj[u] = j;
          const int js = j < n ? j : 0;
          vx[u] = tx[js];
          vy[u] = ty[js];
          vz[u] = tz[js];
          old[u] = td[js];
          id[u] = tid[js];
        }
#pragma unroll
        for (int u = 0; u < kTailUnroll; ++u) {
          if (job[u].x < 0) continue;  // wave-uniform
          const int top_pos = (&grec[0][0])[job[u].y].pos;  // where the bucket's farthest point sits
          TailRec m = none;
          bool lowered = false;
          if (jj[u] < n) {
            const double u0 = vx[u] - px, u1 = vy[u] - py, u2 = vz[u] - pz;
            double d = u0 * u0;
            d = d + u1 * u1;
            d = d + u2 * u2;
            if (d < old[u]) {
              td[jj[u]] = d;
              lowered = jj[u] == top_pos;
            } else {
              d = old[u];
            }
            m = TailRec{d, id[u], jj[u], vx[u], vy[u], vz[u]};
          }
          if (__ballot(lowered) == 0) continue;  // the record stands
          m = tail_best(m);
          if (lane == 0) {
            (&grec[0][0])[job[u].y] = m;
            soa_store(rec, job[u].x, m);
            gdirty[job[u].y / kTailGroup] = 1;
          }
        }
      }
      __syncthreads();
      if (dbg) { const unsigned long long t = wall_clock64(); acc[3] += t - t0; t0 = t; }
      // 5. the batch's groups whose buckets changed fold their records again
      {
        const int d0 = gdirty[wave * kTailPerWave], d1 = gdirty[wave * kTailPerWave + 1],
                  d2 = gdirty[wave * kTailPerWave + 2], d3 = gdirty[wave * kTailPerWave + 3];
        if (d0 | d1 | d2 | d3) {  // wave-uniform
          const TailRec m = row_best(grec[wave][lane]);
          if (gl == 0 && q < ngl && gdirty[wave * kTailPerWave + sub]) put_group(glist[q], m);
        }
      }
      __syncthreads();
      if (dbg) { const unsigned long long t = wall_clock64(); acc[4] += t - t0; t0 = t; }
    }
  }
  if (dbg && tid_ == 0) {
    acc[7] = clock64() - clk0;
    for (int k = 0; k < 8; ++k) dbg[k] = acc[k];
  }
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
      // from here on a round touches a dozen small buckets: all the remaining rounds in one launch of one
      // workgroup (k_fps_tail; PYQSM_FPS_TAIL=0 keeps a launch per round). Measured, 1 M points -> 100 k
      // samples: 5.3 us a round against 5.95 us (0.53 s against 0.595 s), same indices.
      const int nb2 = ceil_div(N, kTailBucket), ng = ceil_div(nb2, kTailGroup);
      const char* te = getenv("PYQSM_FPS_TAIL");
      if (late && ng <= kTailMaxGroups && !(te && te[0] == '0')) {
        const size_t lds = size_t(ng) * (sizeof(TailRec) + 48);
        static std::atomic<uint64_t> attr_set{0};
        const uint64_t bit = 1ull << (c->device & 63);
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
          PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_fps_tail),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
          attr_set.fetch_or(bit, std::memory_order_release);
        }
        // the points in Morton order, 64 to a bucket, 64 buckets to a group: compact against the reach
        unsigned long long* d_dbg = nullptr;
        if (getenv("PYQSM_FPS_TRACE")) PQ_TRY(c->arena.get(8, &d_dbg));
        uint32_t* key;
        int32_t *perm, *t_id;
        TailSoA rec;
        double *t_x, *t_y, *t_z, *t_d;
        PQ_TRY(c->arena.get(size_t(N), &key));
        PQ_TRY(c->arena.get(size_t(N), &perm));
        PQ_TRY(c->arena.get(size_t(N), &t_id));
        PQ_TRY(c->arena.get(size_t(N), &t_x));
        PQ_TRY(c->arena.get(size_t(N), &t_y));
        PQ_TRY(c->arena.get(size_t(N), &t_z));
        PQ_TRY(c->arena.get(size_t(N), &t_d));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.d));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.x));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.y));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.z));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.idx));
        PQ_TRY(c->arena.get(size_t(nb2), &rec.pos));
        for (int a = 0; a < 6; ++a) PQ_TRY(c->arena.get(size_t(nb2), &rec.box[a]));
        hipLaunchKernelGGL(k_fps_tail_key, dim3(ceil_div(N, 256)), dim3(256), 0, c->stream, N,
                           static_cast<const double*>(g.sx), static_cast<const double*>(g.sy),
                           static_cast<const double*>(g.sz), mn[0], mn[1], mn[2], 1024.0 / ext, key, perm);
        PQ_HIP(hipGetLastError());
        PQ_TRY(stable_sort_pairs_u32(c, &key, &perm, N, 30));
        hipLaunchKernelGGL(k_fps_tail_points, dim3(ceil_div(N, 256)), dim3(256), 0, c->stream, N,
                           static_cast<const int32_t*>(perm), static_cast<const double*>(g.sx),
                           static_cast<const double*>(g.sy), static_cast<const double*>(g.sz),
                           static_cast<const double*>(dist), static_cast<const int32_t*>(g.order), t_x, t_y, t_z, t_d, t_id);
        hipLaunchKernelGGL(k_fps_tail_buckets, dim3(ceil_div(nb2, 4)), dim3(256), 0, c->stream, N, nb2,
                           static_cast<const double*>(t_x), static_cast<const double*>(t_y),
                           static_cast<const double*>(t_z), static_cast<const double*>(t_d),
                           static_cast<const int32_t*>(t_id), rec);
        hipLaunchKernelGGL(k_fps_tail, dim3(1), dim3(kTailThreads), lds, c->stream, s, S, N, nb2, ng, rec,
                           static_cast<const int32_t*>(t_id), static_cast<const double*>(t_x),
                           static_cast<const double*>(t_y), static_cast<const double*>(t_z), t_d, d_out, d_dbg);
        PQ_HIP(hipGetLastError());
        if (d_dbg) {
          unsigned long long h[8];
          PQ_HIP(hipMemcpyAsync(h, d_dbg, sizeof(h), hipMemcpyDeviceToHost, c->stream));
          PQ_HIP(hipStreamSynchronize(c->stream));
          const double r = double(S - s);
          fprintf(stderr, "fps tail per round (100 MHz ticks): argmax %.1f, group test %.1f, bucket gather %.1f, update %.1f, "
                  "fold %.1f; groups in reach %.2f, buckets updated %.2f; shader clock %.0f MHz\n", h[0] / r, h[1] / r,
                  h[2] / r, h[3] / r, h[4] / r, h[5] / r, h[6] / r,
                  100.0 * double(h[7]) / double(h[0] + h[1] + h[2] + h[3] + h[4]));
        }
        if (getenv("PYQSM_FPS_TRACE")) fprintf(stderr, "fps: rounds %d .. %d in one launch (%d groups)\n", s, S, ng);
        break;
      }
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
