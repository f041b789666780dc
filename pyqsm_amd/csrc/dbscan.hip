// dbscan.hip — eps-neighbourhood clustering on gfx950, bit-exact with
// scikit-learn's DBSCAN (pyQSM/math_utils/fit.py:223) and usable for Open3D's
// cluster_dbscan call sites (pyQSM/geometry/point_cloud_processing.py:185,209).
//
// Parallel formulation of the sequential reference:
//   1. bin points into cells of edge >= eps (grid.hip)
//   2. core(i)  <=> #{ j in 27-cell stencil : d2(i,j) <= eps^2 } >= min_pts
//   3. union-find over core-core pairs within eps (lock-free hooking, larger
//      root under smaller, agent-scope atomics)
//   4. cluster number = rank of the component's smallest ORIGINAL core index
//      (what the index-order seeding of the sequential algorithm produces)
//   5. border point -> smallest cluster number among its core neighbours
// The distance predicate is evaluated in fp64 exactly as scikit-learn does:
// d2 = ((dx*dx) + dy*dy) + dz*dz with separately rounded products, compared
// with fl(eps*eps). The library is compiled with -ffp-contract=off.
#include "grid.hpp"

namespace pyqsm {

static constexpr int kNoRoot = 0x7FFFFFFF;

__device__ __forceinline__ double sqdist(double ax, double ay, double az, double bx, double by,
                                         double bz) {
  double t0 = ax - bx, t1 = ay - by, t2 = az - bz;
  double d = t0 * t0;
  d = d + t1 * t1;
  d = d + t2 * t2;
  return d;
}

struct Stencil {
  int nx, nxy;
};

// Visit every sorted position q in the 27-cell stencil of cell c:
// nine contiguous runs (x-1..x+1 for each of the 3x3 (y,z) rows).
#define FOR_STENCIL(c, st, start, q, body)                         \
  for (int dz__ = -1; dz__ <= 1; ++dz__)                           \
    for (int dy__ = -1; dy__ <= 1; ++dy__) {                       \
      const int row__ = (c) + dy__ * (st).nx + dz__ * (st).nxy;    \
      const int qb__ = (start)[row__ - 1], qe__ = (start)[row__ + 2]; \
      for (int q = qb__; q < qe__; ++q) {                          \
        body                                                       \
      }                                                            \
    }

__global__ __launch_bounds__(256) void k_core(int n, Stencil st, const int32_t* __restrict__ start,
                                              const int32_t* __restrict__ cell_of,
                                              const double* __restrict__ sx,
                                              const double* __restrict__ sy,
                                              const double* __restrict__ sz, double r2,
                                              int min_pts, uint8_t* __restrict__ core) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const double x = sx[p], y = sy[p], z = sz[p];
  const int c = cell_of[p];
  int cnt = 0;
  FOR_STENCIL(c, st, start, q, { cnt += sqdist(x, y, z, sx[q], sy[q], sz[q]) <= r2; })
  core[p] = cnt >= min_pts;
}

// ---- wave-tiled traversal -----------------------------------------------------------
// A wave owns 64 consecutive sorted points. Their stencils are covered by nine
// LINEAR cell-id intervals [c_first + o - 1, c_last + o + 1] (o = dy*nx + dz*nx*ny),
// each one contiguous run of the sorted arrays, so the candidates are staged through
// LDS 64 at a time with coalesced loads and read back as broadcasts: no per-lane
// gathers, no divergence. Lanes test a superset of their own 27 cells (about 3.6x
// more pairs on the benchmark forest) but each test is an LDS broadcast plus nine
// fp64 instructions instead of three L1/L2 gathers. Waves whose intervals hold more
// than kTileMax candidates (sparse layers above dense ones) take the per-lane path.

static constexpr int kTileMax = 16384;

struct Tile {
  int qb[9], qe[9];
  int total;
};

__device__ __forceinline__ Tile wave_tile(int p0, int n, Stencil st, int ncell,
                                          const int32_t* __restrict__ start,
                                          const int32_t* __restrict__ cell_of) {
  Tile t;
  const int plast = p0 + 63 < n ? p0 + 63 : n - 1;
  const int c_first = __builtin_amdgcn_readfirstlane(cell_of[p0]);
  const int c_last = __builtin_amdgcn_readfirstlane(cell_of[plast]);
  // the nine intervals, sorted by lower end (they already are unless the grid has
  // fewer than three rows), then clipped against what the earlier ones cover: a wave
  // that spans more than a grid row makes neighbouring intervals overlap
  int lo[9], hi[9];
  int w = 0;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy) {
      const int o = dy * st.nx + dz * st.nxy;
      int a = c_first + o - 1, b = c_last + o + 1;
      a = a < 0 ? 0 : a;
      b = b > ncell - 1 ? ncell - 1 : b;
      int k = w++;
      while (k > 0 && lo[k - 1] > a) {
        lo[k] = lo[k - 1];
        hi[k] = hi[k - 1];
        --k;
      }
      lo[k] = a;
      hi[k] = b;
    }
  t.total = 0;
  int prev_hi = -1;
  for (int r = 0; r < 9; ++r) {
    const int a = lo[r] <= prev_hi ? prev_hi + 1 : lo[r];
    if (a <= hi[r]) {
      t.qb[r] = __builtin_amdgcn_readfirstlane(start[a]);
      t.qe[r] = __builtin_amdgcn_readfirstlane(start[hi[r] + 1]);
      prev_hi = hi[r];
    } else {
      t.qb[r] = t.qe[r] = 0;
    }
    t.total += t.qe[r] - t.qb[r];
  }
  return t;
}

struct TileLds {
  double x[4][64], y[4][64], z[4][64];  // the current chunk of candidates, per wave
};

// (Measured alternatives, MI355X, 1 M-point forest: per-lane gathers 0.72 ms; this
// LDS-broadcast fp64 loop 0.48 ms; the same with a rigorous fp32 pre-filter and the
// chunk held in registers / broadcast by v_readlane 0.71 ms — issue-stall bound, see
// profiles/r01_dbscan_sq_counters.csv. The simplest form won.)
__global__ __launch_bounds__(256) void k_core_tiled(int n, Stencil st, int ncell,
                                                    const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ cell_of,
                                                    const double* __restrict__ sx,
                                                    const double* __restrict__ sy,
                                                    const double* __restrict__ sz, double r2,
                                                    int min_pts, uint8_t* __restrict__ core) {
  __shared__ TileLds L;
  // wave-uniform quantities are forced into SGPRs so that the loops below are scalar
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p0 = (blockIdx.x * 4 + w) * 64;
  if (p0 >= n) return;  // whole wave
  const int p = p0 + lane;
  const bool live = p < n;
  const double x = live ? sx[p] : 0.0, y = live ? sy[p] : 0.0, z = live ? sz[p] : 0.0;
  const Tile t = wave_tile(p0, n, st, ncell, start, cell_of);
  int cnt = 0;
  if (t.total > kTileMax) {  // per-lane fallback
    if (live) {
      const int c = cell_of[p];
      FOR_STENCIL(c, st, start, q, { cnt += sqdist(x, y, z, sx[q], sy[q], sz[q]) <= r2; })
    }
  } else {
    for (int r = 0; r < 9; ++r) {
      for (int base = t.qb[r]; base < t.qe[r]; base += 64) {
        const int q = base + lane;
        const int m = t.qe[r] - base < 64 ? t.qe[r] - base : 64;
        if (q < t.qe[r]) {
          L.x[w][lane] = sx[q];
          L.y[w][lane] = sy[q];
          L.z[w][lane] = sz[q];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 4
        for (int j = 0; j < m; ++j)
          cnt += sqdist(x, y, z, L.x[w][j], L.y[w][j], L.z[w][j]) <= r2;
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (live) core[p] = cnt >= min_pts;
}

// ---- union-find --------------------------------------------------------------

__device__ __forceinline__ int ld_parent(const int* parent, int i) {
  return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_parent(int* parent, int i, int v) {
  __hip_atomic_store(parent + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Parent pointers only ever decrease (a root is hooked under a smaller root), so
// every pointer names an ancestor and halving a path is always safe.
__device__ __forceinline__ int find_root(int* parent, int x) {
  int cur = ld_parent(parent, x);
  if (cur != x) {
    int prev = x, next;
    while (cur > (next = ld_parent(parent, cur))) {
      st_parent(parent, prev, next);
      prev = cur;
      cur = next;
    }
  }
  return cur;
}

__device__ __forceinline__ void unite(int* parent, int a, int b) {
  int ra = find_root(parent, a), rb = find_root(parent, b);
  while (ra != rb) {
    if (ra < rb) {
      int t = ra;
      ra = rb;
      rb = t;
    }
    // hook the larger root under the smaller one
    int old = atomicCAS(parent + ra, ra, rb);
    if (old == ra) break;
    ra = old;  // lost the race: ra already has a (smaller) parent, climb
  }
}

__global__ __launch_bounds__(256) void k_init_parent(int n, int* __restrict__ parent) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n) parent[p] = p;
}

__global__ __launch_bounds__(256) void k_union(int n, Stencil st, const int32_t* __restrict__ start,
                                               const int32_t* __restrict__ cell_of,
                                               const double* __restrict__ sx,
                                               const double* __restrict__ sy,
                                               const double* __restrict__ sz, double r2,
                                               const uint8_t* __restrict__ core, int* parent) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  const double x = sx[p], y = sy[p], z = sz[p];
  const int c = cell_of[p];
  // Most neighbours are already in p's tree after the first few unions. A plain
  // (cached, possibly stale) read of parent[q] that equals a root p has been seen
  // under proves "same tree" (trees only merge), so the coherent loads and the CAS
  // are kept for the pairs that still look different.
  const volatile int* vparent = parent;
  int rp = find_root(parent, p);
  FOR_STENCIL(c, st, start, q, {
    // each unordered pair once
    if (q < p && core[q] && sqdist(x, y, z, sx[q], sy[q], sz[q]) <= r2) {
      if (vparent[q] != rp) {
        unite(parent, p, q);
        rp = find_root(parent, p);
      }
    }
  })
}

// (Measured alternatives for this kernel, MI355X, 1 M-point forest, per launch:
//  per-lane walk above 1.15 ms; candidates staged through LDS like k_core_tiled with
//  per-lane unite 1.4 ms; the same plus in-wave component labels so that one lane per
//  local component talks to the global forest 1.3 ms. 91 % of the wave cycles are
//  memory waits in the find/CAS chains themselves (profiles/r01_dbscan_sq_counters.csv),
//  and SIMT serialisation of divergent unite() calls costs the tiled forms more than
//  the cheaper traversal saves. The simple form stays.)

// root[p] for core points, then the smallest original index of each component.
// A cluster of 50 k points would send 50 k atomicMin to one address (0.77 ms in
// the first version): lanes of a wave that share a root fold their indices first,
// and an atomic is issued only if it can still lower the stored minimum.
__global__ __launch_bounds__(256) void k_flatten(int n, const uint8_t* __restrict__ core,
                                                 int* __restrict__ parent,
                                                 const int32_t* __restrict__ order,
                                                 int* __restrict__ min_orig) {
  int p = blockIdx.x * 256 + threadIdx.x;
  const bool active = p < n && core[p];
  int r = -1, v = 0x7FFFFFFF;
  if (active) {
    r = p;
    for (int nx = parent[r]; nx != r; nx = parent[r]) r = nx;  // plain loads: kernel boundary
    parent[p] = r;  // benign: r is still an ancestor for concurrent readers
    v = order[p];
  }
  // wave-level fold when every active lane has the same root (the common case)
  const unsigned long long act = __ballot(active);
  if (act == 0) return;
  const int lead = __ffsll(act) - 1;
  const int r0 = __shfl(r, lead, 64);
  if (__ballot(active && r != r0) == 0) {
    int m = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(m, off, 64);
      m = o < m ? o : m;
    }
    if ((threadIdx.x & 63) == lead && m < __hip_atomic_load(min_orig + r0, __ATOMIC_RELAXED,
                                                            __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(min_orig + r0, m);
  } else if (active) {
    if (v < __hip_atomic_load(min_orig + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(min_orig + r, v);
  }
}

__global__ __launch_bounds__(256) void k_mark_roots(int n, const uint8_t* __restrict__ core,
                                                    const int* __restrict__ parent,
                                                    const int* __restrict__ min_orig,
                                                    int32_t* __restrict__ flag) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  if (parent[p] == p) flag[min_orig[p]] = 1;
}

__global__ __launch_bounds__(256) void k_labels(int n, Stencil st,
                                                const int32_t* __restrict__ start,
                                                const int32_t* __restrict__ cell_of,
                                                const double* __restrict__ sx,
                                                const double* __restrict__ sy,
                                                const double* __restrict__ sz, double r2,
                                                const uint8_t* __restrict__ core,
                                                const int* __restrict__ parent,
                                                const int* __restrict__ min_orig,
                                                const int32_t* __restrict__ rank,
                                                const int32_t* __restrict__ order,
                                                int64_t* __restrict__ labels,
                                                uint8_t* __restrict__ is_core) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  int best = kNoRoot;
  if (core[p]) {
    best = min_orig[parent[p]];
  } else {
    const double x = sx[p], y = sy[p], z = sz[p];
    const int c = cell_of[p];
    FOR_STENCIL(c, st, start, q, {
      if (core[q] && sqdist(x, y, z, sx[q], sy[q], sz[q]) <= r2) {
        int m = min_orig[parent[q]];
        best = m < best ? m : best;
      }
    })
  }
  const int o = order[p];
  labels[o] = best == kNoRoot ? int64_t(-1) : int64_t(rank[best]);
  if (is_core) is_core[o] = core[p];
}

static int dbscan_device(Ctx* c, const double* xyz, int64_t n, double eps, int32_t min_pts,
                         int64_t* labels, uint8_t* is_core, int64_t* n_clusters) {
  if (!(eps > 0) || !std::isfinite(eps)) return fail(PYQSM_EINVAL, "eps must be positive");
  if (n == 0) {
    if (n_clusters) *n_clusters = 0;
    return 0;
  }
  DevGrid g;
  {
    ProfScope ps(c, "dbscan_bin");
    // a hair wider than eps: rounding of the cell index can then never put two
    // points that are within eps of each other two cells apart
    PQ_TRY(build_grid(c, xyz, n, eps * (1.0 + 1.0 / 1048576.0), int64_t(1) << 28, &g));
  }
  const int N = int(n);
  const dim3 grid(ceil_div(n, 256)), block(256);
  const Stencil st{g.nx, g.nx * g.ny};
  const double r2 = eps * eps;
  uint8_t* core;
  int *parent, *min_orig;
  int32_t* flag;
  PQ_TRY(c->arena.get(size_t(n), &core));
  PQ_TRY(c->arena.get(size_t(n), &parent));
  PQ_TRY(c->arena.get(size_t(n), &min_orig));
  PQ_TRY(c->arena.get(size_t(n) + 1, &flag));
  {
    ProfScope ps(c, "dbscan_core");
    hipLaunchKernelGGL(k_core_tiled, grid, block, 0, c->stream, N, st, int(g.ncell), g.start,
                       g.cell_of, g.sx, g.sy, g.sz, r2, min_pts, core);
    PQ_HIP(hipGetLastError());
  }
  {
    ProfScope ps(c, "dbscan_union");
    hipLaunchKernelGGL(k_init_parent, grid, block, 0, c->stream, N, parent);
    PQ_HIP(hipMemsetAsync(min_orig, 0x7F, size_t(n) * 4, c->stream));  // 0x7F7F7F7F > any index
    PQ_HIP(hipMemsetAsync(flag, 0, (size_t(n) + 1) * 4, c->stream));
    hipLaunchKernelGGL(k_union, grid, block, 0, c->stream, N, st, g.start, g.cell_of, g.sx, g.sy,
                       g.sz, r2, core, parent);
    PQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_flatten, grid, block, 0, c->stream, N, core, parent, g.order, min_orig);
    hipLaunchKernelGGL(k_mark_roots, grid, block, 0, c->stream, N, core, parent, min_orig, flag);
    PQ_HIP(hipGetLastError());
    PQ_TRY(exclusive_scan_i32(c, flag, n + 1));
  }
  {
    ProfScope ps(c, "dbscan_label");
    hipLaunchKernelGGL(k_labels, grid, block, 0, c->stream, N, st, g.start, g.cell_of, g.sx, g.sy,
                       g.sz, r2, core, parent, min_orig, flag, g.order, labels, is_core);
    PQ_HIP(hipGetLastError());
  }
  if (n_clusters) {
    int32_t h = 0;
    PQ_HIP(hipMemcpyAsync(&h, flag + n, 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    *n_clusters = h;
  }
  return 0;
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_dbscan_dev(const double* xyz_dev, int64_t n, double eps, int32_t min_pts,
                     int64_t* labels_dev, uint8_t* is_core_dev, int64_t* n_clusters,
                     int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0 && (!xyz_dev || !labels_dev)) return fail(PYQSM_EINVAL, "pyqsm_dbscan_dev: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  ProfScope ps(c, "dbscan_total");
  return dbscan_device(c, xyz_dev, n, eps, min_pts, labels_dev, is_core_dev, n_clusters);
}

int pyqsm_dbscan(const double* xyz, int64_t n, double eps, int32_t min_pts, int64_t* labels,
                 uint8_t* is_core, int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n == 0) return 0;
  if (!xyz || !labels) return fail(PYQSM_EINVAL, "pyqsm_dbscan: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double* d_xyz;
  int64_t* d_lab;
  uint8_t* d_core;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(size_t(n), &d_lab));
  PQ_TRY(c->arena.get(size_t(n), &d_core));
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_TRY(dbscan_device(c, d_xyz, n, eps, min_pts, d_lab, d_core, nullptr));
  PQ_HIP(hipMemcpyAsync(labels, d_lab, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
  if (is_core) PQ_HIP(hipMemcpyAsync(is_core, d_core, size_t(n), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
