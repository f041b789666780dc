// dbscan.hip — eps-neighbourhood clustering on gfx950, bit-exact with
// scikit-learn's DBSCAN (pyQSM/math_utils/fit.py:223) and usable for Open3D's
// cluster_dbscan call sites (pyQSM/geometry/point_cloud_processing.py:185,209).
//
// Parallel formulation of the sequential reference:
//   1. bin points into cells of edge eps, then by octant inside each cell (grid.hip)
//   2. core(i)  <=> #{ j in 27-cell stencil : d2(i,j) <= eps^2 } >= min_pts
//      (wave tiles with an early exit; stragglers in a wave-per-point pass)
//   3. connected components of the core points on the graph of octant sub-cells:
//      a hook pass without union-find, path compression, then lock-free union-find
//      (randomised linking, agent-scope atomics) for what is left; per-point
//      union-find when the grid had to be coarsened
//   4. cluster number = rank of the component's smallest ORIGINAL core index
//      (what the index-order seeding of the sequential algorithm produces)
//   5. border point -> smallest cluster number among its core neighbours
//      (a wave per non-core point)
// The distance predicate is evaluated in fp64 exactly as scikit-learn does:
// d2 = ((dx*dx) + dy*dy) + dz*dz with separately rounded products, compared
// with fl(eps*eps). The library is compiled with -ffp-contract=off.
#include "grid.hpp"
#include <vector>

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

struct TileLds {
  double x[4][64], y[4][64], z[4][64];  // the current chunk of candidates, per wave
};

// (Measured alternatives, MI355X, 1 M-point forest: per-lane gathers 0.72 ms; this
// LDS-broadcast fp64 loop 0.48 ms; the same with a rigorous fp32 pre-filter and the
// chunk held in registers / broadcast by v_readlane 0.71 ms — issue-stall bound, see
// profiles/r01_dbscan_sq_counters.csv. The simplest form won.)
//
// Only "at least min_pts" matters, so the wave stops as soon as (almost) all of its
// points have seen enough neighbours: the runs are visited centre row first, and once
// at most kStragglers lanes are still short they are put on a list for k_core_rest and
// the wave leaves. In a dense cloud that is after 2-4 of the ~13 chunks; without the
// straggler list one noise point would hold its whole wave to the end.
static constexpr int kStragglers = 6;
// ... and after kMaxChunks chunks everybody still short goes there, whatever their number: a
// tile can hold 5000 candidates (80 chunks) and a wave that cannot leave early was the tail
// that set the kernel's duration (average wave 21 us, kernel 300 us). (6 until the end of round 3; with
// k_core_rest taking a point's stencil as one sequence the balance moved: 10 / 6 / 4 / 3 / 2 / 1 chunks ->
// core phase 0.104 / 0.091 / 0.083 / 0.079 / 0.079 / 0.188 ms per million points at min_pts = 10.)
static constexpr int kMaxChunks = 3;  // for min_pts <= 10; more neighbours asked for, more chunks (core_max_chunks)
static constexpr int kRestSegs = 64;  // segments (and counters) of the straggler list
// centre, same-z rows, same-y rows, corners (compile-time: the run bounds stay in SGPRs)
__device__ constexpr int kRunOrder[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};

template <class CO>
__global__ __launch_bounds__(256) void k_core_tiled(int n, Stencil st, int ncell,
                                                    const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ cell_of, CO co, double r2,
                                                    int min_pts, uint8_t* __restrict__ core,
                                                    int32_t* __restrict__ rest,
                                                    int32_t* __restrict__ rest_cnt /*[kRestSegs]*/, int seg_cap,
                                                    int* __restrict__ parent, int* __restrict__ min_orig,
                                                    int32_t* __restrict__ flag /*[n + 1]*/, int max_chunks,
                                                    unsigned long long* __restrict__ tests /*may be null:
                                                    [256] slots, candidates staged per wave (x 64 lanes
                                                    = lane-tests executed)*/) {
  __shared__ TileLds L;
  // wave-uniform quantities are forced into SGPRs so that the loops below are scalar
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p0 = (blockIdx.x * 4 + w) * 64;
  if (p0 >= n) return;  // whole wave
  const int p = p0 + lane;
  const bool live = p < n;
  double x = 0.0, y = 0.0, z = 0.0;
  if (live) co.get(p, x, y, z);
  const Tile t = wave_tile(p0, n, st, ncell, start, cell_of);
  int cnt = 0;  // (starting lanes of sub-cells with >= min_pts points as "decided" gained nothing)
  bool deferred = false;
  int chunks = 0;
  int staged = 0;  // wave-uniform: candidates this wave tested its 64 lanes against
  if (t.total > kTileMax) {
    // The wave's points straddle distant cells (end of one grid layer, start of the next):
    // the linear intervals would sweep whole layers. Take the distinct cells of the wave
    // one at a time instead — exact 27-cell stencil, candidates still staged through LDS,
    // only the lanes of that cell count. (A per-lane walk here made these few waves the
    // 0.25 ms tail of the whole kernel: ~850 dependent gathers per lane.)
    const int mycell = live ? cell_of[p] : -1;
    unsigned long long todo = __ballot(live);
    while (todo) {
      const int lead = __ffsll(todo) - 1;
      const int c = __builtin_amdgcn_readlane(mycell, lead);
      const bool mine = mycell == c;
      todo &= ~__ballot(mine);
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy) {
          const int row = c + dy * st.nx + dz * st.nxy;
          const int qb = __builtin_amdgcn_readfirstlane(start[row - 1]);
          const int qe = __builtin_amdgcn_readfirstlane(start[row + 2]);
          for (int base = qb; base < qe; base += 64) {
            const int q = base + lane;
            const int m = qe - base < 64 ? qe - base : 64;
            if (q < qe) co.get(q, L.x[w][lane], L.y[w][lane], L.z[w][lane]);
            __builtin_amdgcn_wave_barrier();
            staged += m;
            if (mine)
              for (int j = 0; j < m; ++j)
                cnt += sqdist(x, y, z, L.x[w][j], L.y[w][j], L.z[w][j]) <= r2;
            __builtin_amdgcn_wave_barrier();
          }
        }
    }
  } else {
#pragma unroll
    for (int ri = 0; ri < 9; ++ri) {
      const int r = kRunOrder[ri];
      for (int base = t.qb[r]; base < t.qe[r] && !deferred; base += 64) {
        const int q = base + lane;
        const int m = t.qe[r] - base < 64 ? t.qe[r] - base : 64;
        if (q < t.qe[r]) co.get(q, L.x[w][lane], L.y[w][lane], L.z[w][lane]);
        __builtin_amdgcn_wave_barrier();
        staged += m;
#pragma unroll 4
        for (int j = 0; j < m; ++j)
          cnt += sqdist(x, y, z, L.x[w][j], L.y[w][j], L.z[w][j]) <= r2;
        __builtin_amdgcn_wave_barrier();
        const unsigned long long und = __ballot(live && cnt < min_pts);
        ++chunks;
        if ((__popcll(und) <= kStragglers || chunks >= max_chunks) &&
            !(ri == 8 && base + 64 >= t.qe[r])) {
          // the few lanes still short restart on their own in k_core_rest
          if (und != 0) {
            int slot = 0;
            const int lead = __ffsll(und) - 1;
            // kRestSegs counters, a block's stragglers go to segment blockIdx % kRestSegs of the list
            // (capacity seg_cap: a block has at most 256): thousands of waves adding to ONE counter were
            // served one at a time (~10 ns each), and every one of them waited for its turn
            const int seg = blockIdx.x % kRestSegs;
            if (lane == lead) slot = atomicAdd(rest_cnt + seg, __popcll(und));
            slot = __shfl(slot, lead, 64);
            if (live && cnt < min_pts) {
              rest[size_t(seg) * seg_cap + slot + __popcll(und & ((1ull << lane) - 1ull))] = p;
              cnt = -1;
            }
          }
          deferred = true;
          break;
        }
      }
    }
  }
  if (live && cnt >= 0) {
    core[p] = cnt >= min_pts;
    co.mark_core(p, cnt >= min_pts);
  }
  // the start of the union phase rides along (it was a launch of its own, ~5 us): every point its own
  // parent, no smallest index yet, no cluster flags
  if (live) {
    parent[p] = p;
    min_orig[p] = 0x7F7F7F7F;  // > any index
    flag[p] = 0;
    if (p == n - 1) flag[n] = 0;
  }
  if (tests && lane == 0) atomicAdd(tests + (blockIdx.x & 255), static_cast<unsigned long long>(staged));
}

// The 27-cell stencil of one point as ONE sequence of candidates: the nine runs' bounds are loaded side
// by side (they used to be fetched row by row, a dependent round trip each before the row's candidates
// could be asked for), and candidate g of the sequence is found by a chain of selects. Wave-uniform.
struct Runs9 {
  int qb[9];
  int pre[10];  // pre[r] = candidates before run r, pre[9] = all
  __device__ __forceinline__ int at(int g) const {  // sorted position of candidate g < pre[9]
    int q = qb[0] + g;
#pragma unroll
    for (int r = 1; r < 9; ++r) q = g >= pre[r] ? qb[r] + (g - pre[r]) : q;
    return q;
  }
};
__device__ __forceinline__ Runs9 stencil_runs(int c, Stencil st, const int32_t* __restrict__ start) {
  Runs9 t;
  int qe[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const int row = c + (r % 3 - 1) * st.nx + (r / 3 - 1) * st.nxy;
    t.qb[r] = start[row - 1];
    qe[r] = start[row + 2];
  }
  t.pre[0] = 0;
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    t.qb[r] = __builtin_amdgcn_readfirstlane(t.qb[r]);
    t.pre[r + 1] = t.pre[r] + (__builtin_amdgcn_readfirstlane(qe[r]) - t.qb[r]);
  }
  return t;
}

// The stragglers of k_core_tiled, one WAVE each: 64 candidates of the stencil per step,
// stop at min_pts. (One lane each was 0.13 ms per million points: a noise point walks
// ~850 candidates one dependent load at a time.)
template <class CO>
__global__ __launch_bounds__(256) void k_core_rest(const int32_t* __restrict__ rest,
                                                   const int32_t* __restrict__ rest_cnt /*[kRestSegs]*/,
                                                   int seg_cap, Stencil st,
                                                   const int32_t* __restrict__ start,
                                                   const int32_t* __restrict__ cell_of, CO co, double r2,
                                                   int min_pts, uint8_t* __restrict__ core) {
  const int lane = threadIdx.x & 63;
  // the list is kRestSegs segments: lane s holds segment s's count, an inclusive scan numbers the entries
  const int mine = rest_cnt[lane];
  int incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  const int m = __shfl(incl, 63, 64);
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < m; i += gridDim.x * 4) {  // wave-uniform
    const int seg = __popcll(__ballot(incl <= i));  // the first segment whose running total exceeds i
    const int before = __shfl(incl - mine, seg, 64);
    const int p = rest[size_t(seg) * seg_cap + (i - before)];
    double x, y, z;
    co.get(p, x, y, z);
    const int c = __builtin_amdgcn_readfirstlane(cell_of[p]);
    const Runs9 t = stencil_runs(c, st, start);
    int cnt = 0;
    for (int g0 = 0; g0 < t.pre[9] && cnt < min_pts; g0 += 64) {
      const int g = g0 + lane;
      const bool hit = g < t.pre[9] && co.d2(t.at(g), x, y, z) <= r2;
      cnt += __popcll(__ballot(hit));
    }
    if (lane == 0) {
      core[p] = cnt >= min_pts;
      co.mark_core(p, cnt >= min_pts);
    }
  }
}

// ---- union-find --------------------------------------------------------------

__device__ __forceinline__ int ld_parent(const int* parent, int i) {
  return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_parent(int* parent, int i, int v) {
  __hip_atomic_store(parent + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Linking order. Hooking "larger index under smaller" on spatially sorted indices builds
// chains (neighbours in space are neighbours in index), and every find then walks
// hundreds of dependent loads. A bijective hash of the index as the priority is
// randomised linking: expected depth O(log n). A root is hooked under a root of smaller
// priority, so priorities strictly decrease towards the root: every pointer names an
// ancestor, halving a path is always safe, and there are no cycles.
__device__ __forceinline__ unsigned link_prio(int x) { return unsigned(x) * 0x9E3779B1u; }

__device__ __forceinline__ int find_root(int* parent, int x) {
  int cur = ld_parent(parent, x);
  if (cur != x) {
    int prev = x, next;
    while (cur != (next = ld_parent(parent, cur))) {
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
    if (link_prio(ra) < link_prio(rb)) {
      int t = ra;
      ra = rb;
      rb = t;
    }
    // hook the root of larger priority under the other one
    const int old = atomicCAS(parent + ra, ra, rb);
    if (old == ra) break;
    // lost the race: ra has a parent now. Climb with loads, not with failing CAS
    // operations (atomics on one address are served one at a time).
    ra = find_root(parent, old);
    rb = find_root(parent, rb);
  }
}

// ---- union phase over octant sub-cells -------------------------------------------------
// Cells have an edge of eps (a hair more), so two points of the same octant sub-cell
// (half a cell per axis) are at most 0.87 eps apart: the core points of a sub-cell are one
// component without any test. What remains is to find, for every pair of sub-cells at
// most two sub-cells apart per axis, ONE core-core pair within eps — and not even that
// once the two are known to be in the same tree. The first version walked all ~850
// candidates of every core point and chased a parent pointer for each of its ~40 core
// neighbours (1.15 ms per million points, 91 % memory waits); this one looks at
// 62 neighbour sub-cells per sub-cell, most of them empty or already joined.

// The first point of every sub-cell run finds the run's first core point (the sub-cell's
// representative), hangs the other core points of the run under it and lists it.
__global__ __launch_bounds__(1024) void k_sub_rep(int n, const int32_t* __restrict__ sub_of,
                                                 const uint8_t* __restrict__ core,
                                                 const int32_t* __restrict__ order,
                                                 int* __restrict__ parent,
                                                 const int32_t* __restrict__ cell_of,
                                                 int4* __restrict__ rec,
                                                 int* __restrict__ run_min,
                                                 int4* __restrict__ list,
                                                 int32_t* __restrict__ list_cnt, int nx, int ny,
                                                 int4* __restrict__ list_xyz) {
  int p = blockIdx.x * 1024 + threadIdx.x;
  int rep = -1, sid = 0, e = 0;
  if (p < n) {
    sid = sub_of[p];
    const int4 run = rec[sid];  // (first position, points, -, -) of the sub-cell
    if (run.x == p) {
      e = p + run.y;
      int mn = 0x7FFFFFFF;
      for (int q = p; q < e; ++q)
        if (core[q]) {
          if (rep < 0) rep = q;
          parent[q] = rep;
          mn = min(mn, order[q]);
        }
      if (rep >= 0) {
        rec[sid].z = rep;
        run_min[rep] = mn;  // smallest original index among the run's core points
      }
    }
  }
  // block-aggregated append of the representatives: one atomic per block of 1024 threads
  // (atomics on a single address are served one at a time: one per wave cost 0.16 ms per
  // million points, one per 256 threads still 0.04 ms)
  __shared__ int wcount[16], wbase[16];
  const unsigned long long b = __ballot(rep >= 0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wcount[w] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int k = 0; k < 16; ++k) tot += wcount[k];
    int base = tot ? atomicAdd(list_cnt, tot) : 0;
    for (int k = 0; k < 16; ++k) {
      wbase[k] = base;
      base += wcount[k];
    }
  }
  __syncthreads();
  // list record: representative, its cell, its sub-cell id, points from it to the run's end
  // ... and, for k_hook_sub, its cell's grid coordinates and its octant: the two integer divisions by the
  // grid's dimensions happen HERE, once per sub-cell — in the hook pass every wave did twelve of them per
  // four sub-cells (~40 instructions each; the pass's look-ups were 48 us of its 81, a third of that this)
  if (rep >= 0) {
    const int slot = wbase[w] + __popcll(b & ((1ull << lane) - 1ull));
    const int c1 = cell_of[rep];
    list[slot] = make_int4(rep, c1, sid, e - rep);
    list_xyz[slot] = make_int4(c1 % nx, (c1 / nx) % ny, c1 / (nx * ny), sid & 7);
  }
}

// The 62 lexicographically positive (dz, dy, dx) offsets in [-2,2]^3 as (dx, dy, dz): the 13
// of max-norm 1 first, then the 49 of max-norm 2.
__constant__ signed char kSubOffsets[62][3] = {{1,0,0}, {-1,1,0}, {0,1,0}, {1,1,0}, {-1,-1,1}, {0,-1,1}, {1,-1,1}, {-1,0,1}, {0,0,1}, {1,0,1}, {-1,1,1}, {0,1,1}, {1,1,1}, {2,0,0}, {-2,1,0}, {2,1,0}, {-2,2,0}, {-1,2,0}, {0,2,0}, {1,2,0}, {2,2,0}, {-2,-2,1}, {-1,-2,1}, {0,-2,1}, {1,-2,1}, {2,-2,1}, {-2,-1,1}, {2,-1,1}, {-2,0,1}, {2,0,1}, {-2,1,1}, {2,1,1}, {-2,2,1}, {-1,2,1}, {0,2,1}, {1,2,1}, {2,2,1}, {-2,-2,2}, {-1,-2,2}, {0,-2,2}, {1,-2,2}, {2,-2,2}, {-2,-1,2}, {-1,-1,2}, {0,-1,2}, {1,-1,2}, {2,-1,2}, {-2,0,2}, {-1,0,2}, {0,0,2}, {1,0,2}, {2,0,2}, {-2,1,2}, {-1,1,2}, {0,1,2}, {1,1,2}, {2,1,2}, {-2,2,2}, {-1,2,2}, {0,2,2}, {1,2,2}, {2,2,2}};

// Sub-cells one wave of the two passes below takes: a wave per sub-cell (200 k waves of a
// few hundred cycles each per million points) was bound by the rate at which waves START
// (resident waves: 13-21 % of the slots), not by what they did.
static constexpr int kSubPerWaveDefault = 4;

// Full path compression for the listed representatives (plain accesses: the kernel
// boundary makes the unions of the previous launch visible, and any value another lane
// writes meanwhile is an ancestor too).
// Two launches: the first moves every pointer `max_steps` links up (a hundred and more dependent
// loads per thread, all threads starting together, was 35 us per million points), the second then
// reaches the root in depth / max_steps hops over the pointers the first one left.
__global__ __launch_bounds__(256) void k_flatten_reps(const int4* __restrict__ list,
                                                      const int32_t* __restrict__ m_ptr,
                                                      int* __restrict__ parent, int max_steps) {
  const int m = *m_ptr;  // number of listed sub-cells, left on the device by k_sub_rep
  int s = blockIdx.x * 256 + threadIdx.x;
  if (s >= m) return;
  const int p = list[s].x;
  int r = p, steps = 0;
  for (int nx = parent[r]; nx != r && steps < max_steps; nx = parent[r], ++steps) r = nx;
  parent[p] = r;
}

// Pass 1 of the union phase, without union-find: every sub-cell hangs itself under the
// first neighbour it is connected to among the 62 lexicographically NEGATIVE offsets
// (nearest first). Pointers only go to lexicographically smaller sub-cells, so there is
// no cycle, every sub-cell writes its own pointer only (plain store, no atomics, no
// chasing), and what remains after compression is a few trees per cluster — one per
// sub-cell without a connected smaller neighbour. One wave per sub-cell as below.
template <int kSubPerWave, class CO>
__global__ __launch_bounds__(256) void k_hook_sub(const int4* __restrict__ list,
                                                  const int4* __restrict__ list_xyz,
                                                  const int32_t* __restrict__ m_ptr, int nx,
                                                  int ny, const int32_t* __restrict__ start,
                                                  const int4* __restrict__ rec, CO co, double r2,
                                                  const uint8_t* __restrict__ core,
                                                  int* __restrict__ parent,
                                                  int32_t* __restrict__ nbr) {
  const int m = *m_ptr;
  const int k = threadIdx.x & 63;
  const int o_dx = k < 62 ? kSubOffsets[k][0] : 0, o_dy = k < 62 ? kSubOffsets[k][1] : 0,
            o_dz = k < 62 ? kSubOffsets[k][2] : 0;
  // resident waves stride over the list (wave-uniform bounds): the launch is sized for the chip, not
  // for the upper bound n of m — four waves in five of such a grid found nothing to do and still
  // had to be started (SQ_WAVES 250 k per million points for 49 k with work)
  for (int s0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kSubPerWave; s0 < m; s0 += gridDim.x * 4 * kSubPerWave) {
  // the dependent loads (list -> start -> rec) of the wave's sub-cells are issued side by side.
  // (What these kernels cost is the number of cache LINES their gathers touch — one per
  // neighbouring cell and table, taken by the L1 a line at a time. Dropping the second gather
  // start[c2 + 1] by tagging the records with their cell cost more than it saved: lanes of
  // empty cells then fetch the next occupied cell's records.)
  int4 me[kSubPerWave], at[kSubPerWave];
  int b2[kSubPerWave], e2[kSubPerWave], oct[kSubPerWave];
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {  // 16 bytes each, wave-uniform
    me[u] = list[min(s0 + u, m - 1)];
    at[u] = list_xyz[min(s0 + u, m - 1)];
  }
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    const int cx = at[u].x, cy = at[u].y, cz = at[u].z, o1 = at[u].w;
    // half-cell coordinates (cell 1 is the first interior cell; borders are empty)
    const int gx = 2 * (cx - 1) + (o1 & 1) - o_dx, gy = 2 * (cy - 1) + ((o1 >> 1) & 1) - o_dy,
              gz = 2 * (cz - 1) + ((o1 >> 2) & 1) - o_dz;
    const int c2 = (((gz >> 1) + 1) * ny + ((gy >> 1) + 1)) * nx + ((gx >> 1) + 1);
    oct[u] = (gx & 1) | ((gy & 1) << 1) | ((gz & 1) << 2);
    b2[u] = e2[u] = 0;
    if (k < 62 && s0 + u < m) {
      b2[u] = start[c2];
      e2[u] = start[c2 + 1];
    }
  }
  int q0[kSubPerWave], n2[kSubPerWave], rep2[kSubPerWave];
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    q0[u] = n2[u] = 0;
    rep2[u] = -1;
    if (e2[u] != b2[u]) {
      const int4 r = rec[b2[u] * 8 + oct[u]];
      if (r.y > 0) {
        q0[u] = r.x;
        n2[u] = r.y;
        rep2[u] = r.z;
      }
    }
  }
  // the neighbours found here are pass 2's work list (one coalesced 256-byte row per
  // sub-cell): it then starts two dependent loads further down the chain
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u)
    if (s0 + u < m) nbr[size_t(s0 + u) * 64 + k] = rep2[u];
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    const int p = me[u].x, n1 = me[u].w;
    unsigned long long todo = __ballot(rep2[u] >= 0);
    bool found = false;
    while (todo && !found) {
      const int src = __ffsll(todo) - 1;
      todo &= todo - 1;
      const int qb = __shfl(q0[u], src, 64), nb = __shfl(n2[u], src, 64), rb = __shfl(rep2[u], src, 64);
      // pair (i, j) of the two runs sits at index (i << sh) | j, sh = bits of nb - 1: no division by a
      // run length per lane and step (~40 instructions), at the price of idle lanes where nb is no power of two
      const int sh = nb > 1 ? 32 - __clz(nb - 1) : 0;
      const int pairs = n1 << sh;
      for (int base = 0; base < pairs && !found; base += 64) {
        const int idx = base + k;
        const int j = idx & ((1 << sh) - 1);
        bool hit = false;
        if (idx < pairs && j < nb) hit = co.core_pair_within(p + (idx >> sh), qb + j, core, r2);
        found = __ballot(hit) != 0;
      }
      // Hang under the neighbour's own pointer rather than under the neighbour: sub-cells are
      // listed in spatial order, the smaller neighbour's wave has usually finished, and what
      // it points to is an ancestor still smaller than this sub-cell (no cycles). Chains come
      // out a few links long instead of as long as a trunk is tall in sub-cells.
      // (The chains are nevertheless long — 80 % of the benchmark forest's sub-cells sit more than 64
      // links from their root, PYQSM_DBSCAN_TRACE prints the histogram: a third of the list is in flight
      // at once, so most neighbours have not hooked yet when their pointer is read, and agent-scope
      // accesses here change nothing. k_flatten_reps deals with them in two passes.)
      if (found && k == 0) parent[p] = parent[rb];
    }
  }
  // (Taking the wave's sub-cells in rounds — one candidate of each per round, the records of all of them
  // loaded before the first verdict — was SLOWER both times it was built: 104 us with idle lanes loading a
  // dummy record, 112 us with predicated loads and finished sub-cells skipped, against 81. The look-ups
  // above are 48 of those 81 us and live on occupancy (half the resident waves: 118 us); the rounds'
  // record arrays cost the registers that occupancy needs.)
  }
}

// Pass 2: a WAVE per kSubPerWave sub-cells with core points, over the same pairs (sub-cell, neighbour at
// a lexicographically negative offset) that pass 1 resolved and left in `nbr` — every
// unordered pair of neighbouring sub-cells exactly once. Lane k decides whether its
// neighbour still has to be tested (after pass 1 and the compression almost none has: two
// plain loads show the same root); the wave then takes the ones that do one at a time and
// tests all |S1| x |S2| point pairs at once, 64 per step. (A lane per pair of sub-cells
// running the pair loop itself was 3x slower: ~25 dependent iterations per lane, and a wave
// lasts as long as its slowest lane.)
template <int kSubPerWave, class CO>
__global__ __launch_bounds__(256) void k_union_sub(const int4* __restrict__ list,
                                                   const int32_t* __restrict__ m_ptr,
                                                   const int32_t* __restrict__ nbr,
                                                   const int32_t* __restrict__ sub_of,
                                                   const int4* __restrict__ rec, CO co, double r2,
                                                   const uint8_t* __restrict__ core, int* parent) {
  const int m = *m_ptr;
  const int k = threadIdx.x & 63;
  for (int s0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kSubPerWave; s0 < m; s0 += gridDim.x * 4 * kSubPerWave) {
  int rep2[kSubPerWave];
  int4 me[kSubPerWave];
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    rep2[u] = s0 + u < m ? nbr[size_t(s0 + u) * 64 + k] : -1;
    me[u] = list[min(s0 + u, m - 1)];
  }
  // A plain (cached, possibly stale) read names an ancestor; equal ancestors prove
  // "same tree" (trees only merge). The coherent chase is for the rest.
  const int* vparent = parent;
  int a1[kSubPerWave], a2[kSubPerWave];
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    a1[u] = a2[u] = 0;
    if (rep2[u] >= 0) {
      a1[u] = vparent[me[u].x];
      a2[u] = vparent[rep2[u]];
    }
  }
#pragma unroll
  for (int u = 0; u < kSubPerWave; ++u) {
    const int p = me[u].x, n1 = me[u].w;
    bool need = rep2[u] >= 0 && !(a1[u] == a2[u] || a2[u] == p || a1[u] == rep2[u]);
    int root2 = -1;  // the neighbour's tree, as the coherent look found it
    if (need) {
      // The coherent look: both chains at once, starting from the ancestors the plain reads named (after
      // the compression those are the roots unless this launch has hooked them since) — four dependent
      // agent-scope loads per doubtful pair became one or two. (Without this step the pass takes 21 us:
      // its table and the quick test; the doubtful pairs were the other 44.)
      int ra = a1[u], rb = a2[u];
      for (;;) {
        const int pa = ld_parent(parent, ra), pb = ld_parent(parent, rb);
        if (pa == ra && pb == rb) break;
        ra = pa;
        rb = pb;
      }
      need = ra != rb;
      root2 = rb;
    }
    unsigned long long todo = __ballot(need);
    if (!todo) continue;
    int q0 = 0, n2 = 0;
    if (need) {  // the neighbour's run: start and length
      const int4 r = rec[sub_of[rep2[u]]];
      q0 = r.x;
      n2 = r.y;
    }
    while (todo) {
      const int src = __ffsll(todo) - 1;
      todo &= todo - 1;
      const int qb = __shfl(q0, src, 64), nb = __shfl(n2, src, 64), rb = __shfl(rep2[u], src, 64);
      const int tree = __shfl(root2, src, 64);
      const int sh = nb > 1 ? 32 - __clz(nb - 1) : 0;  // (as in k_hook_sub: shift and mask, no division)
      const int pairs = n1 << sh;
      bool found = false;
      for (int base = 0; base < pairs && !found; base += 64) {
        const int idx = base + k;
        const int j = idx & ((1 << sh) - 1);
        bool hit = false;
        if (idx < pairs && j < nb) hit = co.core_pair_within(p + (idx >> sh), qb + j, core, r2);
        found = __ballot(hit) != 0;
      }
      if (found) {
        if (k == 0) unite(parent, p, rb);
        // this sub-cell now hangs together with that whole tree: its other doubtful neighbours in the same
        // tree need no test and no union (a sub-cell on the seam of two trees has ~7 of them, and a wave
        // went through them one after the other — test, two coherent finds, a CAS on the same hot root)
        todo &= ~__ballot(root2 == tree);
      }
    }
  }
  }
}

// ---- fallback for coarsened grids ----------------------------------------------------
// When the cloud's extent would need more than 2^28 cells of edge eps, grid.hip doubles the
// edge; a sub-cell is then wider than eps and says nothing about connectivity. Such clouds
// take the per-point union-find of the first version (same results, ~3x slower union phase).
template <class CO>
__global__ __launch_bounds__(256) void k_union_points(int n, Stencil st,
                                                      const int32_t* __restrict__ start,
                                                      const int32_t* __restrict__ cell_of, CO co, double r2,
                                                      const uint8_t* __restrict__ core, int* parent) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  double x, y, z;
  co.get(p, x, y, z);
  const int c = cell_of[p];
  const volatile int* vparent = parent;
  int rp = find_root(parent, p);
  FOR_STENCIL(c, st, start, q, {
    // each unordered pair once; a plain read equal to p's root proves "same tree"
    if (q < p && core[q] && co.d2(q, x, y, z) <= r2) {
      if (vparent[q] != rp) {
        unite(parent, p, q);
        rp = find_root(parent, p);
      }
    }
  })
}

__global__ __launch_bounds__(256) void k_point_min(int n, const uint8_t* __restrict__ core,
                                                   const int* __restrict__ parent,
                                                   const int32_t* __restrict__ order,
                                                   int* __restrict__ min_orig) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  atomicMin(min_orig + parent[p], order[p]);  // k_flatten ran: parent[p] is the root
}

// Smallest original core index of every component, folded from the per-run minima of
// the representatives (k_sub_rep): 5x fewer values than points, wave-folded when the
// wave's representatives share a root (the common case), and an atomic only if it can
// still lower the stored minimum. (Folding all core points this way cost 0.23 ms per
// million points: atomics and coherent loads on a handful of hot addresses.)
__global__ __launch_bounds__(256) void k_rep_min(const int4* __restrict__ list,
                                                 const int32_t* __restrict__ m_ptr,
                                                 const int* __restrict__ parent,
                                                 const int* __restrict__ run_min,
                                                 int* __restrict__ min_orig) {
  const int m = *m_ptr;
  int s = blockIdx.x * 256 + threadIdx.x;
  const bool active = s < m;
  int r = -1, v = 0x7FFFFFFF;
  if (active) {
    const int p = list[s].x;
    // the root: after the compression that preceded k_union_sub and its few unions the chain is
    // one to three links long (a second compression pass over the list cost 20 us for this)
    r = p;
    for (int nx = parent[r]; nx != r; nx = parent[r]) r = nx;
    v = run_min[p];
  }
  const volatile int* vmin = min_orig;  // plain pre-check: a stale value only costs an atomic
  const int lane = threadIdx.x & 63;
  // a wave of representatives spans a few clusters (grid rows run across the whole scene):
  // fold one root at a time
  unsigned long long rem = __ballot(active);
  while (rem) {
    const int lead = __ffsll(rem) - 1;
    const int r0 = __shfl(r, lead, 64);
    const bool mine = active && r == r0;
    int mn = mine ? v : 0x7FFFFFFF;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(mn, off, 64);
      mn = o < mn ? o : mn;
    }
    if (lane == lead && mn < vmin[r0]) atomicMin(min_orig + r0, mn);
    rem &= ~__ballot(mine);
  }
}

// parent[p] = root for every core point (point -> representative -> root). With `flag` given
// (the components' smallest indices are final by then) the roots also mark their cluster's
// slot, which spares k_mark_roots.
__global__ __launch_bounds__(256) void k_flatten(int n, const uint8_t* __restrict__ core,
                                                 int* __restrict__ parent,
                                                 const int* __restrict__ min_orig /*may be null*/,
                                                 int32_t* __restrict__ flag /*may be null*/) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  int r = p;
  for (int nx = parent[r]; nx != r; nx = parent[r]) r = nx;
  parent[p] = r;  // benign: r is still an ancestor for concurrent readers
  if (flag && r == p) flag[min_orig[p]] = 1;
}

__global__ __launch_bounds__(256) void k_mark_roots(int n, const uint8_t* __restrict__ core,
                                                    const int* __restrict__ parent,
                                                    const int* __restrict__ min_orig,
                                                    int32_t* __restrict__ flag) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || !core[p]) return;
  if (parent[p] == p) flag[min_orig[p]] = 1;
}

// Labels of the core points; the others (noise and border candidates, a few percent) are
// listed for k_labels_border. (Walking the stencil per lane here made those few points the
// tail of the kernel: ~850 dependent gathers each.)
__global__ __launch_bounds__(1024) void k_labels(int n, const uint8_t* __restrict__ core,
                                                 const int* __restrict__ parent,
                                                 const int* __restrict__ min_orig,
                                                 const int32_t* __restrict__ rank,
                                                 const int32_t* __restrict__ order,
                                                 int64_t* __restrict__ labels,
                                                 uint8_t* __restrict__ is_core,
                                                 int32_t* __restrict__ rest,
                                                 int32_t* __restrict__ rest_cnt) {
  int p = blockIdx.x * 1024 + threadIdx.x;
  const bool live = p < n;
  const bool is_c = live && core[p];
  if (is_c) labels[order[p]] = int64_t(rank[min_orig[parent[p]]]);
  if (live && is_core) is_core[order[p]] = is_c;
  // block-aggregated append: one atomic per 1024 threads (atomics on one address are served one at a time)
  __shared__ int wcount[16], wbase[16];
  const unsigned long long nb = __ballot(live && !is_c);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wcount[w] = __popcll(nb);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int k = 0; k < 16; ++k) tot += wcount[k];
    int base = tot ? atomicAdd(rest_cnt, tot) : 0;
    for (int k = 0; k < 16; ++k) {
      wbase[k] = base;
      base += wcount[k];
    }
  }
  __syncthreads();
  if (live && !is_c) rest[wbase[w] + __popcll(nb & ((1ull << lane) - 1ull))] = p;
}

// One WAVE per non-core point: smallest cluster number among its core neighbours, or -1.
template <class CO>
__global__ __launch_bounds__(256) void k_labels_border(const int32_t* __restrict__ rest,
                                                       const int32_t* __restrict__ rest_cnt,
                                                       Stencil st, const int32_t* __restrict__ start,
                                                       const int32_t* __restrict__ cell_of, CO co, double r2,
                                                       const uint8_t* __restrict__ core,
                                                       const int* __restrict__ parent,
                                                       const int* __restrict__ min_orig,
                                                       const int32_t* __restrict__ rank,
                                                       const int32_t* __restrict__ order,
                                                       int64_t* __restrict__ labels) {
  const int m = *rest_cnt;
  const int lane = threadIdx.x & 63;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < m; i += gridDim.x * 4) {  // wave-uniform
    const int p = rest[i];
    double x, y, z;
    co.get(p, x, y, z);
    const int c = __builtin_amdgcn_readfirstlane(cell_of[p]);
    const Runs9 t = stencil_runs(c, st, start);
    int best = kNoRoot;
    for (int g0 = 0; g0 < t.pre[9]; g0 += 64) {
      const int g = g0 + lane;
      if (g < t.pre[9]) {
        const int q = t.at(g);
        if (core[q] && co.d2(q, x, y, z) <= r2) {
          const int mo = min_orig[parent[q]];
          best = mo < best ? mo : best;
        }
      }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int o = __shfl_xor(best, off, 64);
      best = o < best ? o : best;
    }
    if (lane == 0) labels[order[p]] = best == kNoRoot ? int64_t(-1) : int64_t(rank[best]);
  }
}

static int dbscan_device(Ctx* c, const double* xyz, int64_t n, double eps, int32_t min_pts,
                         bool radius_inclusive, int64_t* labels, uint8_t* is_core, int64_t* n_clusters) {
  if (!(eps > 0) || !std::isfinite(eps)) return fail(PYQSM_EINVAL, "eps must be positive");
  if (n == 0) {
    if (n_clusters) *n_clusters = 0;
    return 0;
  }
  DevGrid g;
  SubCells sub;
  {
    ProfScope ps(c, "dbscan_bin");
    // a hair wider than eps: rounding of the cell index can then never put two
    // points that are within eps of each other two cells apart
    const char* old = getenv("PYQSM_DBSCAN_BIN");  // "2pass": the round-1 binning (A/B comparisons)
    if (old && !strcmp(old, "2pass")) {
      PQ_TRY(build_grid(c, xyz, n, eps * (1.0 + 1.0 / 1048576.0), int64_t(1) << 28, &g));
      PQ_TRY(subsort_octants(c, &g, n, &sub));
    } else {
      PQ_TRY(build_grid_octants(c, xyz, n, eps * (1.0 + 1.0 / 1048576.0), int64_t(1) << 28, &g, &sub));
    }
  }
  if (c->prof >= 1) c->timers["dbscan_f32_records"].launches += g.p4 ? 1 : 0;  // which storage form ran
  // the sub-cell shortcuts need cells of edge eps (not doubled to fit the dense grid)
  const bool fine = g.cell <= eps * (1.0 + 1.0 / 524288.0);
  const int N = int(n);
  const dim3 grid(ceil_div(n, 256)), block(256);
  const Stencil st{g.nx, g.nx * g.ny};
  // Every kernel tests d2 <= r2. The STRICT neighbourhood d2 < eps^2 (radius_inclusive = 0: what
  // Open3D's cluster_dbscan computes if nanoflann's radius search compares strictly; SURVEY.md
  // §8 a2) is the same test against the double just below eps^2 — no fp64 value lies between
  // the two, so d2 < r2 <=> d2 <= pred(r2) exactly, and the kernels need no second form.
  const double r2 = radius_inclusive ? eps * eps : nextafter(eps * eps, 0.0);
  uint8_t* core;
  int *parent, *min_orig;
  int32_t* flag;
  PQ_TRY(c->arena.get(size_t(n), &core));
  PQ_TRY(c->arena.get(size_t(n), &parent));
  PQ_TRY(c->arena.get(size_t(n), &min_orig));
  PQ_TRY(c->arena.get(size_t(n) + 1, &flag));
  int4* list;
  int32_t* list_cnt;
  int* run_min;
  PQ_TRY(c->arena.get(size_t(n), &run_min));
  PQ_TRY(c->arena.get(size_t(n), &list));
  int4* list_xyz;  // [n] grid coordinates and octant of every listed sub-cell
  PQ_TRY(c->arena.get(size_t(n), &list_xyz));
  // [0] listed sub-cells, [1] stragglers of the core pass, [2] scratch of the union phase, [3] non-core
  // points of the label pass. The binning leaves four zeroed ints behind for this (no memset launches).
  if (sub.zeroed4) {
    list_cnt = sub.zeroed4;
  } else {
    PQ_TRY(c->arena.get(4 + kZeroedExtra, &list_cnt));
    PQ_HIP(hipMemsetAsync(list_cnt, 0, (4 + kZeroedExtra) * 4, c->stream));
  }
  int32_t* rest;
  // (the core pass's stragglers come in kRestSegs segments of seg_cap entries; the label pass reuses the
  // array as one list of at most n)
  const int core_blocks = int(ceil_div(n, 256));
  const int seg_cap = ceil_div(core_blocks, kRestSegs) * 256;
  PQ_TRY(c->arena.get(std::max<size_t>(size_t(n), size_t(seg_cap) * kRestSegs), &rest));
  int32_t* const rest_segs = list_cnt + 4;  // kRestSegs zeroed counters
  {
    ProfScope ps(c, "dbscan_core");
    unsigned long long* d_tests = nullptr;
    if (c->prof >= 2) {
      PQ_TRY(c->arena.get(256, &d_tests));
      PQ_HIP(hipMemsetAsync(d_tests, 0, 256 * 8, c->stream));
    }
    // chunks of 64 candidates a wave of the tiled pass looks at before it hands its short lanes on: three
    // serve min_pts = 10, and in proportion beyond
    const int max_chunks = std::min(32, std::max(kMaxChunks, (kMaxChunks * min_pts + 9) / 10));
    {
      ProfScope pk(c, "k_core_tiled");
      on_coords(g, [&](auto co) {
        hipLaunchKernelGGL(k_core_tiled<decltype(co)>, grid, block, 0, c->stream, N, st, int(g.ncell), g.start,
                           g.cell_of, co, r2, min_pts, core, rest, rest_segs, seg_cap, parent, min_orig, flag, max_chunks, d_tests);
      });
    }
    if (d_tests) {  // profiling level 2 only: read the counter back (synchronises)
      unsigned long long h[256];
      PQ_HIP(hipMemcpyAsync(h, d_tests, sizeof(h), hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      unsigned long long tot = 0;
      for (unsigned long long v : h) tot += v;
      c->timers["core_pair_tests"].launches += int64_t(tot) * 64;  // lane-tests executed
    }
    on_coords(g, [&](auto co) {
      hipLaunchKernelGGL(k_core_rest<decltype(co)>, dim3(std::min<int64_t>(8192, ceil_div(n, 64))), block, 0,
                         c->stream, rest, rest_segs, seg_cap, st, g.start, g.cell_of, co, r2, min_pts, core);
    });
    PQ_HIP(hipGetLastError());
  }
  {
    ProfScope ps(c, "dbscan_union");
    // (parent / min_orig / flag were initialised by k_core_tiled; list_cnt[0] and [2] are still the zeros
    // the binning left)
    if (fine) {
      hipLaunchKernelGGL(k_sub_rep, dim3(ceil_div(n, 1024)), dim3(1024), 0, c->stream, N, sub.sub_of, core,
                         g.order, parent, g.cell_of, sub.rec, run_min, list, list_cnt, g.nx, g.ny, list_xyz);
      // The number m of listed sub-cells stays on the device: the passes below are launched for the
      // upper bound (a sub-cell holds at least one point, in practice ~5) and read m themselves —
      // waves beyond it leave at once — which spares the host round trip in the middle of the step
      // (~15 us of an 0.7 ms step). Clouds so large that the neighbour table for n rows would not be
      // reasonable (> 8 GiB) read m back and size everything exactly.
      int64_t rows = n;
      if (size_t(n) * 256 > (size_t(8) << 30)) {
        int32_t m = 0;
        PQ_HIP(hipMemcpyAsync(&m, list_cnt, 4, hipMemcpyDeviceToHost, c->stream));
        PQ_HIP(hipStreamSynchronize(c->stream));
        rows = m;
      }
      if (rows > 0) {
        // the two wave-per-sub-cells passes: at most the waves the chip holds at their occupancy (8 per
        // SIMD), striding over the list
        // the two wave-per-sub-cells passes: resident waves striding over the list, 16 (hook) and 32
        // (union) blocks per CU — measured against a block per 16 rows of the bound n: hook 91 -> 82 us,
        // union 68.5 -> 64 us per million points (PYQSM_UNION_BLOCKS_PER_CU=<hook>,<union>; 0 = the bound)
        int per_cu[2] = {16, 32};
        if (const char* e_cu = getenv("PYQSM_UNION_BLOCKS_PER_CU")) {
          per_cu[0] = per_cu[1] = atoi(e_cu);
          if (const char* comma = strchr(e_cu, ',')) per_cu[1] = atoi(comma + 1);
        }
        const int64_t full = ceil_div(rows, 4 * kSubPerWaveDefault);
        const dim3 gh(per_cu[0] > 0 ? std::min<int64_t>(full, int64_t(c->cu_count) * per_cu[0]) : full),
            gu(per_cu[1] > 0 ? std::min<int64_t>(full, int64_t(c->cu_count) * per_cu[1]) : full),
            gl(ceil_div(rows, 256));
        int32_t* nbr;  // [m][64] representatives of the neighbour sub-cells pass 1 resolved
        PQ_TRY(c->arena.get(size_t(rows) * 64, &nbr));
        {
          ProfScope pk(c, "k_hook_sub");
          on_coords(g, [&](auto co) {
            hipLaunchKernelGGL((k_hook_sub<kSubPerWaveDefault, decltype(co)>), gh, block, 0, c->stream, list,
                               list_xyz, list_cnt, g.nx, g.ny, g.start, sub.rec, co, r2, core, parent, nbr);
          });
        }
        if (getenv("PYQSM_DBSCAN_TRACE")) {  // how deep are the chains the hook pass leaves?
          int32_t m = 0;
          PQ_HIP(hipMemcpyAsync(&m, list_cnt, 4, hipMemcpyDeviceToHost, c->stream));
          PQ_HIP(hipStreamSynchronize(c->stream));
          std::vector<int32_t> hp(static_cast<size_t>(n), 0);
          std::vector<int4> hl(static_cast<size_t>(m), make_int4(0, 0, 0, 0));
          PQ_HIP(hipMemcpy(hp.data(), parent, size_t(n) * 4, hipMemcpyDeviceToHost));
          PQ_HIP(hipMemcpy(hl.data(), list, size_t(m) * 16, hipMemcpyDeviceToHost));
          std::vector<int64_t> hist(66, 0);
          int64_t roots = 0;
          for (int32_t s2 = 0; s2 < m; ++s2) {
            int d = 0, r = hl[size_t(s2)].x;
            while (hp[size_t(r)] != r && d < 65) {
              r = hp[size_t(r)];
              ++d;
            }
            hist[size_t(d)]++;
            roots += d == 0;
          }
          fprintf(stderr, "hook pass: %d sub-cells, %lld roots; chain depth histogram:", m, (long long)roots);
          for (int d = 0; d < 66; ++d)
            if (hist[size_t(d)]) fprintf(stderr, " %d:%lld", d, (long long)hist[size_t(d)]);
          fprintf(stderr, "\n");
        }
        static const int jump = [] {  // PYQSM_FLATTEN_JUMP: links of the first pass (0: one pass, the earlier form)
          const char* e = getenv("PYQSM_FLATTEN_JUMP");
          return e ? atoi(e) : 12;
        }();
        if (jump > 0) hipLaunchKernelGGL(k_flatten_reps, gl, block, 0, c->stream, list, list_cnt, parent, jump);
        hipLaunchKernelGGL(k_flatten_reps, gl, block, 0, c->stream, list, list_cnt, parent, 0x7fffffff);
        // what is left: joining the few trees per cluster. Almost every pair of neighbours
        // now shows the same root through two plain loads.
        {
          ProfScope pk(c, "k_union_sub");
          on_coords(g, [&](auto co) {
            hipLaunchKernelGGL((k_union_sub<kSubPerWaveDefault, decltype(co)>), gu, block, 0, c->stream, list,
                               list_cnt, nbr, sub.sub_of, sub.rec, co, r2, core, parent);
          });
        }
        hipLaunchKernelGGL(k_rep_min, gl, block, 0, c->stream, list, list_cnt, parent, run_min, min_orig);
      }
      PQ_HIP(hipGetLastError());
      hipLaunchKernelGGL(k_flatten, grid, block, 0, c->stream, N, core, parent, min_orig, flag);
    } else {
      on_coords(g, [&](auto co) {
        hipLaunchKernelGGL(k_union_points<decltype(co)>, grid, block, 0, c->stream, N, st, g.start, g.cell_of, co,
                           r2, core, parent);
      });
      hipLaunchKernelGGL(k_flatten, grid, block, 0, c->stream, N, core, parent, static_cast<const int*>(nullptr),
                         static_cast<int32_t*>(nullptr));
      hipLaunchKernelGGL(k_point_min, grid, block, 0, c->stream, N, core, parent, g.order, min_orig);
      hipLaunchKernelGGL(k_mark_roots, grid, block, 0, c->stream, N, core, parent, min_orig, flag);
    }
    PQ_HIP(hipGetLastError());
    PQ_TRY(exclusive_scan_i32(c, flag, n + 1));
  }
  {
    ProfScope ps(c, "dbscan_label");
    hipLaunchKernelGGL(k_labels, dim3(ceil_div(n, 1024)), dim3(1024), 0, c->stream, N, core, parent, min_orig, flag, g.order,
                       labels, is_core, rest, list_cnt + 3);
    on_coords(g, [&](auto co) {
      hipLaunchKernelGGL(k_labels_border<decltype(co)>, dim3(std::min<int64_t>(8192, ceil_div(n, 64))), block, 0,
                         c->stream, rest, list_cnt + 3, st, g.start, g.cell_of, co, r2, core, parent, min_orig,
                         flag, g.order, labels);
    });
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

int pyqsm_dbscan_dev_ex(const double* xyz_dev, int64_t n, double eps, int32_t min_pts,
                        int32_t radius_inclusive, int64_t* labels_dev, uint8_t* is_core_dev,
                        int64_t* n_clusters, int32_t device) {
  PQ_API_RANGE("pyqsm_dbscan_dev");
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0 && (!xyz_dev || !labels_dev)) return fail(PYQSM_EINVAL, "pyqsm_dbscan_dev: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  ProfScope ps(c, "dbscan_total");
  return dbscan_device(c, xyz_dev, n, eps, min_pts, radius_inclusive != 0, labels_dev, is_core_dev,
                       n_clusters);
}

int pyqsm_dbscan_dev(const double* xyz_dev, int64_t n, double eps, int32_t min_pts,
                     int64_t* labels_dev, uint8_t* is_core_dev, int64_t* n_clusters,
                     int32_t device) {
  return pyqsm_dbscan_dev_ex(xyz_dev, n, eps, min_pts, 1, labels_dev, is_core_dev, n_clusters, device);
}

int pyqsm_dbscan(const double* xyz, int64_t n, double eps, int32_t min_pts, int64_t* labels,
                 uint8_t* is_core, int32_t device) {
  return pyqsm_dbscan_ex(xyz, n, eps, min_pts, 1, labels, is_core, device);
}

int pyqsm_dbscan_ex(const double* xyz, int64_t n, double eps, int32_t min_pts, int32_t radius_inclusive,
                    int64_t* labels, uint8_t* is_core, int32_t device) {
  PQ_API_RANGE("pyqsm_dbscan");
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
  PQ_TRY(dbscan_device(c, d_xyz, n, eps, min_pts, radius_inclusive != 0, d_lab, d_core, nullptr));
  PQ_HIP(hipMemcpyAsync(labels, d_lab, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
  if (is_core) PQ_HIP(hipMemcpyAsync(is_core, d_core, size_t(n), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
