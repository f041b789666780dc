// radius.hip — fixed-radius queries on gfx950.
//
//   pyqsm_ball_query    all points within r of ONE centre: what
//                       scipy KDTree(points).query_ball_point(center, r) returns at
//                       pyQSM/utils/lib_integration.py:114-115 (find_neighbors_in_ball,
//                       called once per branch segment by qsm_generation.sphere_step;
//                       the reference rebuilds the KD-tree of the whole cloud every
//                       call). One HBM pass: 24 B per point, inclusive d <= r.
//   pyqsm_radius_mark   for a set of query points, the union of their (up to k nearest)
//                       neighbours within a distance bound: the index set that
//                       KDTree(src).query(query, k, distance_upper_bound=dist) yields at
//                       pyQSM/geometry/reconstruction.py:238-244 and
//                       pyQSM/tree_isolation.py:126-131,207-209. Strict d < dist, like
//                       cKDTree. The source cloud is binned into cells of edge `dist`;
//                       one lane per query walks its 27 cells. When more than k points
//                       lie inside the bound, the k-th smallest squared distance is found
//                       exactly by bisection on its bit pattern and only those are marked.
// Squared distances are fp64, ((dx*dx)+dy*dy)+dz*dz, the accumulation of cKDTree.
#include "grid.hpp"

#include <cmath>

namespace pyqsm {

__device__ __forceinline__ double sqd(double ax, double ay, double az, double bx, double by,
                                      double bz) {
  double t0 = ax - bx, t1 = ay - by, t2 = az - bz;
  double d = t0 * t0;
  d = d + t1 * t1;
  d = d + t2 * t2;
  return d;
}

__global__ __launch_bounds__(256) void k_ball_flags(int64_t n, const double* __restrict__ xyz,
                                                    double cx, double cy, double cz, double r2,
                                                    int32_t* __restrict__ flags) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i > n) return;
  flags[i] = i < n && sqd(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], cx, cy, cz) <= r2;
}

__global__ __launch_bounds__(256) void k_ball_compact(int64_t n, const int32_t* __restrict__ pos,
                                                      int64_t* __restrict__ out) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (pos[i + 1] != pos[i]) out[pos[i]] = i;
}

struct RGrid {
  double minx, miny, minz, inv;
  int nx, ny, nz;
};

__device__ __forceinline__ void clamped_cell(const RGrid& g, double x, double y, double z, int* cx, int* cy,
                                             int* cz) {
  const double fx = floor((x - g.minx) * g.inv), fy = floor((y - g.miny) * g.inv), fz = floor((z - g.minz) * g.inv);
  *cx = int(fmin(fmax(fx, 0.0), double(g.nx - 3))) + 1;
  *cy = int(fmin(fmax(fy, 0.0), double(g.ny - 3))) + 1;
  *cz = int(fmin(fmax(fz, 0.0), double(g.nz - 3))) + 1;
}

// MODE 0: count of source points with d2 < r2.  MODE 1: count with d2 <= tau.
// MODE 2: mark every source point with d2 < bound (bound = r2, or tau plus ties).
// MODE 3: as MODE 1, and *below = the largest d2 <= tau, *above = the smallest d2 > tau (both among
//         the candidates with d2 < r2; -1 / +inf when there is none).
template <int MODE, class CO>
__device__ __forceinline__ int walk(const RGrid& g, const int32_t* __restrict__ start,
                                    const int32_t* __restrict__ order,
                                    CO co, double x, double y, double z,
                                    double r2, double tau, int budget, uint8_t* __restrict__ mark,
                                    int32_t* __restrict__ lab_out = nullptr, int lab = 0,
                                    double* below = nullptr, double* above = nullptr) {
  // The query's cell, clamped into the grid the way the sources were binned (grid.hip: cell_index):
  // the grid may cover less than the cloud (source_grid below), and a clamp moves no two points
  // further apart, so whatever is within the radius of a query outside still sits in the 27 cells
  // around its clamped cell; the distance test is on the true coordinates.
  int cx, cy, cz;
  clamped_cell(g, x, y, z, &cx, &cy, &cz);
  int cnt = 0;
  double lo_v = -1.0, hi_v = __builtin_inf();
  for (int dz = -1; dz <= 1; ++dz) {
    const int zz = cz + dz;
    if (zz < 0 || zz >= g.nz) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int yy = cy + dy;
      if (yy < 0 || yy >= g.ny) continue;
      const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
      const int row = (zz * g.ny + yy) * g.nx;
      for (int q = start[row + x0]; q < start[row + x1 + 1]; ++q) {
        const double d = co.d2(q, x, y, z);
        if (MODE == 0) cnt += d < r2;
        if (MODE == 1) cnt += d < r2 && d <= tau;
        if (MODE == 3 && d < r2) {
          if (d <= tau) {
            ++cnt;
            lo_v = d > lo_v ? d : lo_v;
          } else {
            hi_v = d < hi_v ? d : hi_v;
          }
        }
        if (MODE == 2) {
          bool take = d < r2 && d < tau;
          if (!take && d < r2 && d == tau && cnt < budget) {  // ties at the k-th distance
            take = true;
            ++cnt;
          }
          if (take) {
            if (lab_out) atomicMin(&lab_out[order[q]], lab);
            else mark[order[q]] = 1;
          }
        }
      }
    }
  }
  if (MODE == 3) {
    *below = lo_v;
    *above = hi_v;
  }
  return cnt;
}

template <class CO>
__global__ __launch_bounds__(256) void k_radius_mark(int m, const double* __restrict__ qry,
                                                     RGrid g, const int32_t* __restrict__ start,
                                                     const int32_t* __restrict__ order,
                                                     CO co, double r2,
                                                     int k, uint8_t* __restrict__ mark,
                                                     int32_t* __restrict__ counts,
                                                     const int32_t* __restrict__ qlab /*may be null*/,
                                                     int32_t* __restrict__ lab_out,
                                                     const int32_t* __restrict__ perm /*queries in cell order, may be null*/) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  if (gid >= m) return;
  const int i = perm ? perm[gid] : gid;
  const double x = qry[3 * i], y = qry[3 * i + 1], z = qry[3 * i + 2];
  int32_t* lo_ = qlab ? lab_out : nullptr;
  const int lab = qlab ? qlab[i] : 0;
  const int c = walk<0>(g, start, order, co, x, y, z, r2, 0.0, 0, mark);
  counts[i] = c < k ? c : k;
  if (c == 0) return;
  double tau = __builtin_inf();
  int budget = 0;
  if (c > k) {
    // k-th smallest squared distance = the smallest DATA value v with #{d2 <= v} >= k. Round 2 bisected the
    // 63-bit pattern (62 walks over ~1000 candidates for every query with more than k points in reach — half
    // of a forest's points at k = 200, radius 0.1: the branches — 8 ms per 100 k queries). Now the bisection
    // runs on VALUES and snaps to the data: a walk also returns the largest candidate <= t and the smallest
    // > t, so every walk discards half of the interval AND everything that is not a candidate's distance:
    // ~log2(candidates) walks. Same tau, to the bit.
    double L = -1.0;  // #{d2 <= L} < k
    double H = 0.0;   // a data value with #{d2 <= H} >= k: the largest candidate below r2
    {
      double b0, a0;
      (void)walk<3>(g, start, order, co, x, y, z, r2, __builtin_inf(), 0, mark, nullptr, 0, &b0, &a0);
      H = b0;
    }
    for (int it = 0; it < 200 && L < H; ++it) {
      double t = L + (H - L) * 0.5;
      if (!(t > L) || !(t < H)) t = L < 0.0 ? 0.0 : nextafter(L, H);  // neighbours in fp64: test L's successor
      if (!(t < H)) break;
      double b1, a1;
      const int cnt = walk<3>(g, start, order, co, x, y, z, r2, t, 0, mark, nullptr, 0, &b1, &a1);
      if (cnt >= k) {
        H = b1;  // the largest candidate <= t still has >= k at or below it
      } else {
        L = t;
        if (!(a1 < H)) break;  // no candidate between t and H: H is the k-th
      }
    }
    tau = H;
    const unsigned long long lo = (unsigned long long)__double_as_longlong(tau);
    // points strictly below tau are all taken; ties at tau fill what is left of k
    const double below = lo == 0 ? -1.0 : __longlong_as_double((long long)(lo - 1));
    const int n_below = lo == 0 ? 0 : walk<1>(g, start, order, co, x, y, z, r2, below, 0, mark);
    budget = k - n_below;
  }
  (void)walk<2>(g, start, order, co, x, y, z, r2, tau, budget, mark, lo_, lab);
}


// The queries in the order of the source grid's cells (round 3): a lane per query walks ~850 candidates of
// 27 cells, and 64 unrelated queries per wave are 64 unrelated walks — every load a gather of 64 lines.
// In cell order the lanes of a wave walk the same few runs and their loads fall on shared lines
// (100 k queries against 1 M sources: 4.8 -> 4.0 ms). Marks, labels (atomicMin) and the per-query
// counts do not depend on the order in which the queries are served.
__global__ __launch_bounds__(256) void k_query_keys(int m, const double* __restrict__ qry, RGrid g,
                                                    uint32_t* __restrict__ key, int32_t* __restrict__ ident) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  int cx, cy, cz;
  clamped_cell(g, qry[3 * i], qry[3 * i + 1], qry[3 * i + 2], &cx, &cy, &cz);
  key[i] = uint32_t((cz * g.ny + cy) * g.nx + cx);
  ident[i] = i;
}

static int query_order(Ctx* c, const double* d_qry, int64_t m, const RGrid& rg, int64_t ncell, int32_t** perm) {
  *perm = nullptr;
  const char* e = getenv("PYQSM_RADIUS_SORT");  // "0": serve the queries in the caller's order
  if (m < 1024 || (e && e[0] == '0')) return 0;
  uint32_t* key;
  int32_t* ident;
  PQ_TRY(c->arena.get(size_t(m), &key));
  PQ_TRY(c->arena.get(size_t(m), &ident));
  hipLaunchKernelGGL(k_query_keys, dim3(ceil_div(m, 256)), dim3(256), 0, c->stream, int(m), d_qry, rg, key, ident);
  PQ_HIP(hipGetLastError());
  int bits = 1;
  while (bits < 32 && (int64_t(1) << bits) < ncell) ++bits;
  PQ_TRY(stable_sort_pairs_u32(c, &key, &ident, m, bits));
  *perm = ident;
  return 0;
}

// ---- k nearest within a bound, as sorted padded tables (cKDTree.query semantics) --------
// One wave per query. Count the source points inside the bound (64 candidates per step);
// if there are more than k, find the k-th smallest squared distance exactly by bisection
// on its bit pattern (rare: k defaults to 500 at a 5 cm bound); collect the survivors in
// LDS, bitonic-sort them by (d2, index) and write k entries, padded with (inf, n).
static constexpr int kKnnCap = 2048;  // largest k (LDS: 24 KB per wave)

struct RowRuns {
  int qb[9], qe[9];
};

__device__ __forceinline__ bool query_runs(const RGrid& g, const int32_t* __restrict__ start,
                                           double x, double y, double z, RowRuns* rr) {
  int cx, cy, cz;
  clamped_cell(g, x, y, z, &cx, &cy, &cz);
  int w = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy, ++w) {
      const int zz = cz + dz, yy = cy + dy;
      rr->qb[w] = rr->qe[w] = 0;
      if (zz < 0 || zz >= g.nz || yy < 0 || yy >= g.ny) continue;
      const int x0 = cx - 1 < 0 ? 0 : cx - 1, x1 = cx + 1 >= g.nx ? g.nx - 1 : cx + 1;
      const int row = (zz * g.ny + yy) * g.nx;
      rr->qb[w] = start[row + x0];
      rr->qe[w] = start[row + x1 + 1];
    }
  return true;
}

template <class CO>
__global__ __launch_bounds__(128) void k_radius_knn(int m, const double* __restrict__ qry, RGrid g,
                                                    const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ order,
                                                    CO co, double r2, int k,
                                                    int n_src, int64_t* __restrict__ out_idx,
                                                    double* __restrict__ out_dist) {
  __shared__ double sd[2][kKnnCap];
  __shared__ int si[2][kKnnCap];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 2 + w;
  if (i >= m) return;  // whole wave
  const double x = qry[3 * i], y = qry[3 * i + 1], z = qry[3 * i + 2];
  RowRuns rr;
  const bool inside = query_runs(g, start, x, y, z, &rr);
  // count of candidates with d2 < r2 and d2 <= tau
  auto count_le = [&](double tau) {
    int cnt = 0;
    for (int r = 0; r < 9; ++r)
      for (int base = rr.qb[r]; base < rr.qe[r]; base += 64) {
        const int q = base + lane;
        bool in = false;
        if (q < rr.qe[r]) {
          const double d = co.d2(q, x, y, z);
          in = d < r2 && d <= tau;
        }
        cnt += __popcll(__ballot(in));
      }
    return cnt;
  };
  int total = inside ? count_le(__builtin_inf()) : 0;
  double tau = __builtin_inf();
  int budget = 0;  // ties at tau that still fit
  if (total > k) {
    unsigned long long lo = 0, hi = (unsigned long long)__double_as_longlong(r2);
    while (lo < hi) {  // smallest t with #{d2 <= t} >= k
      const unsigned long long mid = lo + ((hi - lo) >> 1);
      if (count_le(__longlong_as_double((long long)mid)) >= k) hi = mid;
      else lo = mid + 1;
    }
    tau = __longlong_as_double((long long)lo);
    const int n_below = lo == 0 ? 0 : count_le(__longlong_as_double((long long)(lo - 1)));
    budget = k - n_below;
    total = k;
  }
  // collect (ties at tau in candidate order until the budget is used up)
  int have = 0, ties = 0;
  if (total > 0)
    for (int r = 0; r < 9; ++r)
      for (int base = rr.qb[r]; base < rr.qe[r]; base += 64) {
        const int q = base + lane;
        double d = 0.0;
        bool below = false, tie = false;
        if (q < rr.qe[r]) {
          d = co.d2(q, x, y, z);
          below = d < r2 && d < tau;
          tie = d < r2 && d == tau;
        }
        const unsigned long long tb = __ballot(tie);
        const int tie_rank = ties + __popcll(tb & ((1ull << lane) - 1ull));
        const bool take = below || (tie && tie_rank < budget);
        const unsigned long long kb = __ballot(take);
        if (take) {
          const int slot = have + __popcll(kb & ((1ull << lane) - 1ull));
          sd[w][slot] = d;
          si[w][slot] = order[q];
        }
        have += __popcll(kb);
        ties += __popcll(tb);
      }
  // pad to a power of two and sort by (d2, index)
  int np2 = 1;
  while (np2 < have) np2 <<= 1;
  for (int t = have + lane; t < np2; t += 64) {
    sd[w][t] = __builtin_inf();
    si[w][t] = 0x7FFFFFFF;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (int k2 = 2; k2 <= np2; k2 <<= 1)
    for (int j = k2 >> 1; j > 0; j >>= 1) {
      for (int t = lane; t < np2; t += 64) {
        const int u = t ^ j;
        if (u > t) {
          const double da = sd[w][t], db = sd[w][u];
          const int ia = si[w][t], ib = si[w][u];
          const bool a_gt_b = da > db || (da == db && ia > ib);
          const bool up = (t & k2) == 0;
          if (a_gt_b == up) {
            sd[w][t] = db;
            si[w][t] = ib;
            sd[w][u] = da;
            si[w][u] = ia;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  for (int t = lane; t < k; t += 64) {
    const bool real = t < have;
    out_idx[size_t(i) * k + t] = real ? int64_t(si[w][t]) : int64_t(n_src);
    out_dist[size_t(i) * k + t] = real ? sqrt(sd[w][t]) : __builtin_inf();
  }
}

// The grid of the source points: cells of the radius, over the cloud without its sparse tails (a few
// stray returns far outside would otherwise inflate the box until the dense grid cannot have cells
// of the radius any more — every doubling of the edge is 8x the points per cell).
static int source_grid(Ctx* c, const double* d_src, int64_t n, double radius, DevGrid* g) {
  double box[6];
  bool all_f32 = false;  // every source coordinate fp32-representable: 16-byte fp32 records (grid.hpp: on_coords)
  PQ_TRY(cloud_bbox(c, d_src, n, box, box + 3, &all_f32));
  int64_t outside = 0;
  PQ_TRY(robust_box(c, d_src, n, int(std::min<int64_t>(8192, std::max<int64_t>(256, n / 256))), box, &outside));
  return build_grid(c, d_src, n, radius * (1.0 + 1.0 / 1048576.0), int64_t(1) << 28, g, box, all_f32);
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_ball_query(const double* xyz, int64_t n, const double center[3], double radius,
                     int64_t* out_idx, int64_t* count, int32_t device) {
  PQ_API_RANGE("pyqsm_ball_query");
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (count) *count = 0;
  if (n == 0) return 0;
  if (!xyz || !center || !out_idx || !count)
    return fail(PYQSM_EINVAL, "pyqsm_ball_query: NULL pointer");
  if (!(radius >= 0)) return fail(PYQSM_EINVAL, "radius must be >= 0");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points per call");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_xyz;
  int32_t* d_flags;
  int64_t* d_out;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_flags));
  PQ_TRY(c->arena.get(size_t(n), &d_out));
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, "ball_query");
    hipLaunchKernelGGL(k_ball_flags, dim3(ceil_div(n + 1, 256)), dim3(256), 0, c->stream, n, d_xyz,
                       center[0], center[1], center[2], radius * radius, d_flags);
    PQ_TRY(exclusive_scan_i32(c, d_flags, n + 1));
    hipLaunchKernelGGL(k_ball_compact, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, n, d_flags,
                       d_out);
    PQ_HIP(hipGetLastError());
  }
  int32_t total = 0;
  PQ_HIP(hipMemcpyAsync(&total, d_flags + n, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (total)
    PQ_HIP(hipMemcpyAsync(out_idx, d_out, size_t(total) * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  *count = total;
  return 0;
}

int pyqsm_radius_mark(const double* src, int64_t n, const double* qry, int64_t m, double radius,
                      int32_t k_cap, uint8_t* mark, int32_t* counts, int32_t device) {
  PQ_API_RANGE("pyqsm_radius_mark");
  if (n < 0 || m < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0 && (!src || !mark)) return fail(PYQSM_EINVAL, "pyqsm_radius_mark: NULL pointer");
  if (m > 0 && (!qry || !counts)) return fail(PYQSM_EINVAL, "pyqsm_radius_mark: NULL pointer");
  if (!(radius > 0) || !std::isfinite(radius)) return fail(PYQSM_EINVAL, "radius must be positive");
  if (k_cap <= 0) return fail(PYQSM_EINVAL, "k must be positive");
  if (n > 0) memset(mark, 0, size_t(n));
  if (m > 0) memset(counts, 0, size_t(m) * 4);
  if (n == 0 || m == 0) return 0;
  if (m > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 query points per call");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double *d_src, *d_qry;
  uint8_t* d_mark;
  int32_t* d_counts;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_src));
  PQ_TRY(c->arena.get(size_t(m) * 3, &d_qry));
  PQ_TRY(c->arena.get(size_t(n), &d_mark));
  PQ_TRY(c->arena.get(size_t(m), &d_counts));
  PQ_HIP(hipMemcpyAsync(d_src, src, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_qry, qry, size_t(m) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemsetAsync(d_mark, 0, size_t(n), c->stream));
  DevGrid g;
  PQ_TRY(source_grid(c, d_src, n, radius, &g));
  RGrid rg{g.minx, g.miny, g.minz, g.inv_cell, g.nx, g.ny, g.nz};
  {
    ProfScope ps(c, "radius_mark");
    int32_t* perm = nullptr;
    PQ_TRY(query_order(c, d_qry, m, rg, g.ncell, &perm));
    on_coords(g, [&](auto co) {
      hipLaunchKernelGGL(k_radius_mark<decltype(co)>, dim3(ceil_div(m, 256)), dim3(256), 0, c->stream, int(m), d_qry,
                         rg, g.start, g.order, co, radius * radius, k_cap, d_mark, d_counts,
                         static_cast<const int32_t*>(nullptr), static_cast<int32_t*>(nullptr),
                         static_cast<const int32_t*>(perm));
    });
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(mark, d_mark, size_t(n), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(counts, d_counts, size_t(m) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_radius_knn(const double* src, int64_t n, const double* qry, int64_t m, double radius,
                     int32_t k, int64_t* idx, double* dist, int32_t device) {
  PQ_API_RANGE("pyqsm_radius_knn");
  if (n < 0 || m < 0) return fail(PYQSM_EINVAL, "negative size");
  if (k <= 0 || k > kKnnCap) return fail(PYQSM_ERANGE, "k must be in [1, %d]", kKnnCap);
  if (m > 0 && (!qry || !idx || !dist)) return fail(PYQSM_EINVAL, "pyqsm_radius_knn: NULL pointer");
  if (n > 0 && !src) return fail(PYQSM_EINVAL, "pyqsm_radius_knn: NULL pointer");
  if (!(radius > 0) || !std::isfinite(radius)) return fail(PYQSM_EINVAL, "radius must be positive");
  if (m == 0) return 0;
  if (m > 0x7FFFFF00LL || n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points per call");
  if (n == 0) {  // nothing to find: every slot is padding
    for (int64_t t = 0; t < m * k; ++t) {
      idx[t] = 0;
      dist[t] = HUGE_VAL;
    }
    return 0;
  }
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double *d_src, *d_qry, *d_dist;
  int64_t* d_idx;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_src));
  PQ_TRY(c->arena.get(size_t(m) * 3, &d_qry));
  PQ_TRY(c->arena.get(size_t(m) * k, &d_idx));
  PQ_TRY(c->arena.get(size_t(m) * k, &d_dist));
  PQ_HIP(hipMemcpyAsync(d_src, src, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_qry, qry, size_t(m) * 24, hipMemcpyHostToDevice, c->stream));
  DevGrid g;
  PQ_TRY(source_grid(c, d_src, n, radius, &g));
  RGrid rg{g.minx, g.miny, g.minz, g.inv_cell, g.nx, g.ny, g.nz};
  {
    ProfScope ps(c, "radius_knn");
    on_coords(g, [&](auto co) {
      hipLaunchKernelGGL(k_radius_knn<decltype(co)>, dim3(ceil_div(m, 2)), dim3(128), 0, c->stream, int(m), d_qry, rg,
                         g.start, g.order, co, radius * radius, k, int(n), d_idx, d_dist);
    });
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(idx, d_idx, size_t(m) * k * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(dist, d_dist, size_t(m) * k * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_radius_label(const double* src, int64_t n, const double* qry, int64_t m,
                       const int32_t* qry_label, double radius, int32_t k_cap, int32_t* label,
                       int32_t* counts, int32_t device) {
  PQ_API_RANGE("pyqsm_radius_label");
  if (n < 0 || m < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0 && (!src || !label)) return fail(PYQSM_EINVAL, "pyqsm_radius_label: NULL pointer");
  if (m > 0 && (!qry || !qry_label || !counts))
    return fail(PYQSM_EINVAL, "pyqsm_radius_label: NULL pointer");
  if (!(radius > 0) || !std::isfinite(radius)) return fail(PYQSM_EINVAL, "radius must be positive");
  if (k_cap <= 0) return fail(PYQSM_EINVAL, "k must be positive");
  for (int64_t i = 0; i < n; ++i) label[i] = -1;
  if (m > 0) memset(counts, 0, size_t(m) * 4);
  if (n == 0 || m == 0) return 0;
  for (int64_t i = 0; i < m; ++i)
    if (qry_label[i] < 0) return fail(PYQSM_EINVAL, "query labels must be >= 0");
  if (m > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 query points per call");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double *d_src, *d_qry;
  int32_t *d_lab, *d_counts, *d_qlab;
  uint8_t* d_mark;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_src));
  PQ_TRY(c->arena.get(size_t(m) * 3, &d_qry));
  PQ_TRY(c->arena.get(size_t(n), &d_lab));
  PQ_TRY(c->arena.get(size_t(m), &d_counts));
  PQ_TRY(c->arena.get(size_t(m), &d_qlab));
  PQ_TRY(c->arena.get(1, &d_mark));
  PQ_HIP(hipMemcpyAsync(d_src, src, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_qry, qry, size_t(m) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_qlab, qry_label, size_t(m) * 4, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemsetAsync(d_lab, 0x7F, size_t(n) * 4, c->stream));  // 0x7F7F7F7F > any label
  DevGrid g;
  PQ_TRY(source_grid(c, d_src, n, radius, &g));
  RGrid rg{g.minx, g.miny, g.minz, g.inv_cell, g.nx, g.ny, g.nz};
  {
    ProfScope ps(c, "radius_label");
    int32_t* perm = nullptr;
    PQ_TRY(query_order(c, d_qry, m, rg, g.ncell, &perm));
    on_coords(g, [&](auto co) {
      hipLaunchKernelGGL(k_radius_mark<decltype(co)>, dim3(ceil_div(m, 256)), dim3(256), 0, c->stream, int(m), d_qry,
                         rg, g.start, g.order, co, radius * radius, k_cap, d_mark, d_counts, d_qlab, d_lab,
                         static_cast<const int32_t*>(perm));
    });
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(label, d_lab, size_t(n) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(counts, d_counts, size_t(m) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  for (int64_t i = 0; i < n; ++i)
    if (label[i] == 0x7F7F7F7F) label[i] = -1;
  return 0;
}

}  // extern "C"
