// knn.hip — exact k nearest neighbours of every point of a cloud on gfx950.
//
// Stands in for the neighbour search inside robust_laplacian.point_cloud_laplacian
// (pyQSM/geometry/skeletonize.py:253-255) and for scipy cKDTree.query
// (pyQSM/geometry/reconstruction.py:238-240).
//
// Method: points are counting-sorted into a uniform cell grid (grid.hip) whose
// edge is tuned so that an occupied cell holds about k/3 points. One lane per
// query walks Chebyshev shells of cells around its own cell (each shell row is
// one contiguous run of the sorted arrays) and keeps the best k candidates sorted in
// registers (k <= 32, k_knn_reg: by the distance alone in the first pass, ties detected and
// searched again exactly — see the kernel) or in a per-lane LDS column (element j of lane t
// at [j*T + t]: conflict-free, k_knn). A
// query is final once its k-th distance is no larger than the radius the
// visited cube is known to cover. The few queries that are not final after
// kMaxRing shells (isolated outliers) are retried, one wave per query, on a 4x coarser
// grid (binned from the points when the fine grid is large, else derived from it:
// grid.hip: coarsen_grid) with up to kWideRing shells, and on coarser ones still until the
// shells cover the whole grid.
//
// (Measured alternative for level 0, MI355X, 1 M points, k = 20: a wave-tiled search like
// dbscan.hip's k_core_tiled — 64 consecutive queries, candidates broadcast from LDS, the K
// best per lane sorted in registers, branch-free insertion — took 9.0 ms against 5.1 ms for
// the per-lane walk below: almost every candidate beats SOME lane's current worst, so the
// ~300-instruction insertion step runs for nearly all of the tile's superset of candidates.)
//
// Distances are squared, fp64, ((dx*dx)+dy*dy)+dz*dz with separately rounded
// products — the accumulation order of scipy's cKDTree — and candidates are
// ordered by (d2, original index).
#include <atomic>

#include "grid.hpp"

#include <cmath>

namespace pyqsm {

static constexpr int kMaxRing = 3;
// The (wave-per-query) retries may walk this many shells: what is still open there are
// outliers, and one more 4x coarsening instead would make those next to a tree scan whole
// stems (a point 0.7 m from a trunk: 27 cells of 0.8 m; 4 such queries took 1.25 ms on the
// 1 M-point forest). From the first retry level on: 1.05 -> 0.83 ms for both levels with 8 shells.
// With 16 the first retry level finishes every outlier of the forest (1 % noise up to 5 m from a
// stem) and the second one — a coarsening of the grid and a launch whose duration is one long wave,
// 0.32 + 0.06 ms even for a single query — does not happen: retries 0.83 -> 0.38 ms. A query that
// does walk all 16 shells looks up 13 000 row segments (200 rounds of 64), about what a launch
// of the next level costs. PYQSM_KNN_WIDE overrides.
static constexpr int kWideRing = 16;
static constexpr int kMaxK = 192;

struct KnnGrid {
  int nx, ny, nz;
  double cell;
  double minx, miny, minz;  // origin of the interior cells (cell 1 starts here)
};

// sqdist3: grid.hpp

// T = threads per block (LDS holds T columns of k (d2, id) pairs).
template <int T, class CO>
__global__ __launch_bounds__(T) void k_knn(int n_query, const int32_t* __restrict__ query_list,
                                           const int32_t* __restrict__ pos_of /*orig->sorted*/,
                                           KnnGrid g, const int32_t* __restrict__ start,
                                           const int32_t* __restrict__ order,
                                           const int32_t* __restrict__ cell_of,
                                           CO co, int k,
                                           int exclude_self, int n_total, int last_level,
                                           int32_t* __restrict__ out_idx,
                                           double* __restrict__ out_d2,
                                           int32_t* __restrict__ fail_list,
                                           int32_t* __restrict__ fail_count) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* bd = reinterpret_cast<double*>(smem);              // [k][T]
  int32_t* bi = reinterpret_cast<int32_t*>(bd + size_t(k) * T);  // [k][T]
  const int t = threadIdx.x;
  const int qi = blockIdx.x * T + t;
  if (qi >= n_query) return;
  // sorted position of this query
  const int p = query_list ? pos_of[query_list[qi]] : qi;
  const int self = order[p];
  double x, y, z;
  co.get(p, x, y, z);
  const int c = cell_of[p];
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  int have = 0;
  bool done = false;
  const int rmax_grid = max(g.nx, max(g.ny, g.nz));
  for (int r = 0; r <= kMaxRing && !done; ++r) {
    for (int dz = -r; dz <= r; ++dz) {
      const int zz = cz + dz;
      if (zz < 0 || zz >= g.nz) continue;
      for (int dy = -r; dy <= r; ++dy) {
        const int yy = cy + dy;
        if (yy < 0 || yy >= g.ny) continue;
        const int row = (zz * g.ny + yy) * g.nx;
        const bool full = (dz == -r || dz == r || dy == -r || dy == r);
        // full shell row: [cx-r, cx+r]; otherwise the two end cells only
        const int nseg = full ? 1 : (r > 0 ? 2 : 1);
        for (int s = 0; s < nseg; ++s) {
          int x0, x1;
          if (full) {
            x0 = cx - r;
            x1 = cx + r;
          } else {
            x0 = x1 = s == 0 ? cx - r : cx + r;
          }
          x0 = x0 < 0 ? 0 : x0;
          x1 = x1 >= g.nx ? g.nx - 1 : x1;
          if (x0 > x1) continue;
          const int qb = start[row + x0], qe = start[row + x1 + 1];
          for (int q = qb; q < qe; ++q) {
            if (exclude_self && q == p) continue;
            const double d = co.d2(q, x, y, z);
            const int id = order[q];
            if (have == k) {
              const double wd = bd[(k - 1) * T + t];
              if (d > wd || (d == wd && id > bi[(k - 1) * T + t])) continue;
            }
            int j = have < k ? have : k - 1;
            while (j > 0) {
              const double pd = bd[(j - 1) * T + t];
              const int pi = bi[(j - 1) * T + t];
              if (pd < d || (pd == d && pi < id)) break;
              bd[j * T + t] = pd;
              bi[j * T + t] = pi;
              --j;
            }
            bd[j * T + t] = d;
            bi[j * T + t] = id;
            if (have < k) ++have;
          }
        }
      }
    }
    // everything outside the visited cube is at least r cells away
    const double safe = double(r) * g.cell * 0.999999;
    if (have == k && bd[(k - 1) * T + t] <= safe * safe) done = true;
    if (r >= rmax_grid) done = true;  // the cube already covers the whole grid
  }
  if (!done && !last_level) {
    fail_list[atomicAdd(fail_count, 1)] = self;
    return;
  }
  for (int j = 0; j < k; ++j) {
    out_idx[size_t(self) * k + j] = j < have ? bi[j * T + t] : n_total;
    if (out_d2) out_d2[size_t(self) * k + j] = j < have ? bd[j * T + t] : __builtin_inf();
  }
}


// Level-0 search for k <= 32: the same per-lane shell walk, but the sorted best-K list lives
// in REGISTERS (K = k rounded up to a multiple of 4; unrolled, branch-free insertion) instead
// of a 12*k-byte LDS column per lane. At k = 20 the LDS columns allow two blocks per CU
// (two waves per SIMD) and the walk is a chain of dependent gathers: occupancy is what it
// needs. Slots beyond k only make the acceptance test slightly more generous.
// BUF > 0: accepted candidates are first pushed into a BUF-slot unsorted per-lane buffer; the
// sorted insertion (~11 instructions per list slot, executed by the whole wave as soon as ONE
// lane has a candidate to insert — which at 64 lanes is nearly every candidate) runs only when
// some lane's buffer is full, for all lanes and all their buffered candidates at once. The
// acceptance test then works with a slightly stale k-th distance (a superset is buffered);
// buffers are flushed before every "done" test, so the result is the same list. Measured on the
// 1 M-point forest, k = 20: see DESIGN.md (k_knn_reg).
// STRICT: the list is ordered by the distance ALONE (one fp64 compare per slot instead of two plus
// an index compare: 9 -> 7 vector instructions per slot, and the insertions are two thirds of this
// kernel: search 1.78 -> 1.25 ms per million points). Exactness is kept by detection, not by the
// comparison: equal distances INSIDE the list end up in arrival order and are put into index order
// by a pass at the end (a few instructions unless there are ties); a tie ACROSS the k-th place — a
// candidate dropped or evicted with the distance of the list's last entry — makes which index
// belongs to the result depend on what the comparison ignored, so that query is listed in
// `tie_list` and searched again by the exact variant (STRICT = false over `qlist`). Coordinates
// quantised to millimetres give a few per cent of such queries, a regular lattice nearly all
// (then the strict pass is wasted: 1.7x the exact kernel alone).
template <int K, int BUF, bool STRICT, class CO>
__global__ __launch_bounds__(256) void k_knn_reg(int n_query, KnnGrid g,
                                                 const int32_t* __restrict__ qlist /*sorted positions
                                                 to search, or null: all n_query*/,
                                                 const int32_t* __restrict__ qcount /*length of qlist,
                                                 on the device*/,
                                                 int q_min /*a list of at most this many is somebody
                                                 else's (k_knn_wave): nothing to do*/,
                                                 const int32_t* __restrict__ start,
                                                 const int32_t* __restrict__ order,
                                                 const int32_t* __restrict__ cell_of,
                                                 CO co, int k,
                                                 int exclude_self, int n_total, int last_level,
                                                 int32_t* __restrict__ out_idx,
                                                 double* __restrict__ out_d2,
                                                 int32_t* __restrict__ fail_list,
                                                 int32_t* __restrict__ fail_count,
                                                 int32_t* __restrict__ tie_list,
                                                 int32_t* __restrict__ tie_count) {
  const int gid = blockIdx.x * 256 + threadIdx.x;
  int p = gid;  // sorted position of this query
  if (qlist) {
    const int cnt = *qcount;
    if (cnt <= q_min || gid >= cnt) return;
    p = qlist[gid];
  } else if (p >= n_query) {
    return;
  }
  const int self = order[p];
  double x, y, z;
  co.get(p, x, y, z);
  const int c = cell_of[p];
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  double bd[K];
  int bi[K];
#pragma unroll
  for (int j = 0; j < K; ++j) {
    bd[j] = __builtin_inf();
    bi[j] = 0x7FFFFFFF;
  }
  constexpr int NB = BUF > 0 ? BUF : 1;
  double fd[NB];  // buffered candidates, newest first; empty slots hold (inf, max)
  int fi[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    fd[j] = __builtin_inf();
    fi[j] = 0x7FFFFFFF;
  }
  int nbuf = 0;
  // STRICT: the smallest distance at which a candidate was dropped or evicted while equal to the
  // list's last entry; only if that is still the last entry's distance at the end does the result
  // depend on an index the comparison did not look at (earlier ties lie beyond the final k-th place)
  double tie_val = __builtin_inf();
  // sorted insertion of one candidate; (inf, max) is never smaller than a list entry: a no-op
  auto insert = [&](double d, int id) {
    if constexpr (STRICT) {
      const double last = bd[K - 1];
      bool lt_cur = d < last;
      const bool enters = lt_cur;
      tie_val = d == last && d < tie_val ? d : tie_val;  // dropped: as far as the list's last entry
#pragma unroll
      for (int t = K - 1; t > 0; --t) {
        const bool lt_prev = d < bd[t - 1];
        bd[t] = lt_prev ? bd[t - 1] : (lt_cur ? d : bd[t]);
        bi[t] = lt_prev ? bi[t - 1] : (lt_cur ? id : bi[t]);
        lt_cur = lt_prev;
      }
      bd[0] = lt_cur ? d : bd[0];
      bi[0] = lt_cur ? id : bi[0];
      tie_val = enters && last == bd[K - 1] && last < tie_val ? last : tie_val;  // evicted: as far as the new last
    } else {
      bool lt_cur = d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]);
#pragma unroll
      for (int t = K - 1; t > 0; --t) {
        const bool lt_prev = d < bd[t - 1] || (d == bd[t - 1] && id < bi[t - 1]);
        bd[t] = lt_prev ? bd[t - 1] : (lt_cur ? d : bd[t]);
        bi[t] = lt_prev ? bi[t - 1] : (lt_cur ? id : bi[t]);
        lt_cur = lt_prev;
      }
      bd[0] = lt_cur ? d : bd[0];
      bi[0] = lt_cur ? id : bi[0];
    }
  };
  auto flush = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      insert(fd[j], fi[j]);
      fd[j] = __builtin_inf();
      fi[j] = 0x7FFFFFFF;
    }
    nbuf = 0;
  };
  bool done = false;
  const int rmax_grid = max(g.nx, max(g.ny, g.nz));
  for (int r = 0; r <= kMaxRing && !done; ++r) {
    const int side = 2 * r + 1;
    for (int t = 0; t < side * side; ++t) {
      {
        // rows of the shell; for r = 1 nearest first — the centre row, the four rows that share a
        // face with it, the four corner rows (codes (dz+1)*3 + (dy+1), four bits each): the sooner
        // the list holds near points, the fewer of the later candidates pass the k-th-distance
        // test and have to be inserted. The result does not depend on the order.
        const int code = r == 1 ? int((0x862075134ull >> (4 * t)) & 15ull) : t;
        const int dz = code / side - r, dy = code % side - r;
        const int zz = cz + dz;
        if (zz < 0 || zz >= g.nz) continue;
        const int yy = cy + dy;
        if (yy < 0 || yy >= g.ny) continue;
        const int row = (zz * g.ny + yy) * g.nx;
        const bool full = (dz == -r || dz == r || dy == -r || dy == r);
        const int nseg = full ? 1 : (r > 0 ? 2 : 1);
        for (int s = 0; s < nseg; ++s) {
          int x0, x1;
          if (full) {
            x0 = cx - r;
            x1 = cx + r;
          } else {
            x0 = x1 = s == 0 ? cx - r : cx + r;
          }
          x0 = x0 < 0 ? 0 : x0;
          x1 = x1 >= g.nx ? g.nx - 1 : x1;
          if (x0 > x1) continue;
          const int qb = start[row + x0], qe = start[row + x1 + 1];
          for (int q = qb; q < qe; ++q) {
            if constexpr (BUF > 0) {
              // no early `continue`: every lane that is in this iteration reaches the ballot
              const double d = co.d2(q, x, y, z);
              bool acc = !(exclude_self && q == p) && (d < bd[K - 1] || d == bd[K - 1]);
              int id = 0x7FFFFFFF;
              if (acc) {
                id = order[q];
                if constexpr (!STRICT) acc = d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]);
              }
              if (acc) {
#pragma unroll
                for (int j = NB - 1; j > 0; --j) {  // push (static indices: registers, not scratch)
                  fd[j] = fd[j - 1];
                  fi[j] = fi[j - 1];
                }
                fd[0] = d;
                fi[0] = id;
                ++nbuf;
              }
              // one lane's buffer is full: EVERY lane of the iteration merges what it has
              if (__ballot(nbuf == NB) != 0ull) flush();
            } else {
              if (exclude_self && q == p) continue;
              const double d = co.d2(q, x, y, z);
              if (!(d < bd[K - 1] || d == bd[K - 1])) continue;  // cheap reject before the id load
              const int id = order[q];
              if (!(d < bd[K - 1] || (d == bd[K - 1] && id < bi[K - 1]))) continue;
              insert(d, id);
            }
          }
        }
      }
    }
    if constexpr (BUF > 0) flush();
    // everything outside the visited cube is at least r cells away
    double kth = __builtin_inf();
#pragma unroll
    for (int j = 0; j < K; ++j)
      if (j == k - 1) kth = bd[j];
    const double safe = double(r) * g.cell * 0.999999;
    if (kth <= safe * safe) done = true;
    if (r >= rmax_grid) done = true;  // the cube already covers the whole grid
  }
  bool tie = false;
  if constexpr (STRICT) {
    // a tie across the k-th place: with something outside the list (k = K) ...
    tie = k == K && tie_val == bd[K - 1] && tie_val < __builtin_inf();
    // ... or inside it (lists longer than k)
#pragma unroll
    for (int j = 1; j < K; ++j)
      if (j == k) tie |= bd[j] == bd[j - 1] && bd[j] < __builtin_inf();
    // equal distances inside the list: arrival order -> index order (bubble passes over the
    // indices of equal neighbours; no pass finds anything to do unless there are ties)
    for (int pass = 0; pass < K; ++pass) {
      bool sw = false;
#pragma unroll
      for (int t = 1; t < K; ++t) {
        const bool c = bd[t] == bd[t - 1] && bi[t] < bi[t - 1];
        const int lo = c ? bi[t] : bi[t - 1], hi = c ? bi[t - 1] : bi[t];
        bi[t - 1] = lo;
        bi[t] = hi;
        sw |= c;
      }
      if (__ballot(sw) == 0ull) break;
    }
  }
  // one atomic per wave for the queries that stay open (per lane it was ~10^4 atomics on one
  // address per million points)
  const bool open = !done && !last_level;
  const unsigned long long ob = __ballot(open);
  if (ob != 0) {
    const int lane = threadIdx.x & 63, lead = __ffsll(ob) - 1;
    int base = 0;
    if (lane == lead) base = atomicAdd(fail_count, __popcll(ob));
    base = __shfl(base, lead, 64);
    if (open) {
      fail_list[base + __popcll(ob & ((1ull << lane) - 1ull))] = self;
      return;
    }
  }
  if constexpr (STRICT) {  // the queries whose result hangs on a tie: again, by the exact variant
    const unsigned long long tb = __ballot(tie);
    if (tb != 0) {
      const int lane = threadIdx.x & 63, lead = __ffsll(tb) - 1;
      int base = 0;
      if (lane == lead) base = atomicAdd(tie_count, __popcll(tb));
      base = __shfl(base, lead, 64);
      if (tie) {
        tie_list[base + __popcll(tb & ((1ull << lane) - 1ull))] = p;
        return;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j)
    if (j < k) {
      const bool have = bd[j] < __builtin_inf();
      out_idx[size_t(self) * k + j] = have ? bi[j] : n_total;
      if (out_d2) out_d2[size_t(self) * k + j] = bd[j];
    }
}

static int knn_buffer_slots() {  // PYQSM_KNN_BUF=0: direct insertion (the round-1 kernel)
  static const int v = [] {
    const char* e = getenv("PYQSM_KNN_BUF");
    return e ? atoi(e) : 4;
  }();
  return v;
}

template <class CO>
__global__ void k_knn_wave(int n_query, const int32_t* __restrict__ n_ptr, int n_max,
                           const int32_t* __restrict__ query_list, const int32_t* __restrict__ pos_of,
                           KnnGrid g, const int32_t* __restrict__ start, const int32_t* __restrict__ order,
                           const int32_t* __restrict__ cell_of, CO co, int k,
                           int exclude_self, int n_total, int last_level, int max_ring,
                           int32_t* __restrict__ out_idx, double* __restrict__ out_d2,
                           int32_t* __restrict__ fail_list, int32_t* __restrict__ fail_count);

static bool knn_strict() {  // PYQSM_KNN_STRICT=0: the exact comparison in the first pass already
  static const bool v = [] {
    const char* e = getenv("PYQSM_KNN_STRICT");
    return !(e && e[0] == '0');
  }();
  return v;
}

// tie_list [n] / tie_count [1] (zeroed by the caller): scratch of the strict pass.
template <int K>
static int launch_knn_reg(Ctx* c, int n, const DevGrid& g, int k, int excl, int last, int32_t* idx,
                          double* d2, int32_t* fail_list, int32_t* fail_count, int32_t* tie_list,
                          int32_t* tie_count) {
  KnnGrid kg{g.nx, g.ny, g.nz, g.cell, g.minx, g.miny, g.minz};
  const dim3 grid(ceil_div(n, 256)), blk(256);
  const int32_t* none = nullptr;
  on_coords(g, [&](auto co) {
    using CO = decltype(co);
    if (knn_buffer_slots() == 0) {
      hipLaunchKernelGGL((k_knn_reg<K, 0, false, CO>), grid, blk, 0, c->stream, n, kg, none, none, 0, g.start, g.order,
                         g.cell_of, co, k, excl, n, last, idx, d2, fail_list, fail_count, tie_list, tie_count);
    } else if (!knn_strict()) {
      hipLaunchKernelGGL((k_knn_reg<K, 4, false, CO>), grid, blk, 0, c->stream, n, kg, none, none, 0, g.start, g.order,
                         g.cell_of, co, k, excl, n, last, idx, d2, fail_list, fail_count, tie_list, tie_count);
    } else {
      hipLaunchKernelGGL((k_knn_reg<K, 4, true, CO>), grid, blk, 0, c->stream, n, kg, none, none, 0, g.start, g.order,
                         g.cell_of, co, k, excl, n, last, idx, d2, fail_list, fail_count, tie_list, tie_count);
      // The queries whose result hung on a tie, by the exact comparison: a WAVE per query (a lane per
      // query would be one long-running wave of 64 unrelated walks — 0.5 ms for 238 such queries of
      // a millimetre-quantised million points; this way they cost 20 us). Their number stays on the
      // device; the waves of a fixed grid stride over them. A cloud on a regular lattice ties nearly
      // everywhere (760 k of a million queries): past n / 16 the list goes to the per-lane kernel
      // with the exact comparison instead (both launches are issued, one of them finds nothing to do).
      const int many = n / 16;
      hipLaunchKernelGGL(k_knn_wave<CO>, dim3(unsigned(std::min<int64_t>(ceil_div(n, 4), 8192))), blk, 0, c->stream, n,
                         static_cast<const int32_t*>(tie_count), many, static_cast<const int32_t*>(tie_list),
                         static_cast<const int32_t*>(nullptr), kg, g.start, g.order, g.cell_of, co, k, excl, n, last,
                         kMaxRing, idx, d2, fail_list, fail_count);
      hipLaunchKernelGGL((k_knn_reg<K, 4, false, CO>), grid, blk, 0, c->stream, n, kg,
                         static_cast<const int32_t*>(tie_list), static_cast<const int32_t*>(tie_count), many, g.start,
                         g.order, g.cell_of, co, k, excl, n, last, idx, d2, fail_list, fail_count, tie_list, tie_count);
    }
  });
  PQ_HIP(hipGetLastError());
  return 0;
}

// Wave-per-query variant for the retry levels (k <= 64): there the cells are
// coarse and hold thousands of points, so the 64 lanes scan a run together.
// The sorted best-k list lives in registers, element j in lane j; an insertion
// is a ballot (position) plus one lane shift.
template <class CO>
__global__ __launch_bounds__(256) void k_knn_wave(int n_query,
                                                  const int32_t* __restrict__ n_ptr /*the number of
                                                  queries on the device, or null: n_query*/,
                                                  int n_max /*with n_ptr: more queries than this are
                                                  somebody else's (k_knn_reg over the list)*/,
                                                  const int32_t* __restrict__ query_list /*original
                                                  indices (pos_of given) or sorted positions*/,
                                                  const int32_t* __restrict__ pos_of, KnnGrid g,
                                                  const int32_t* __restrict__ start,
                                                  const int32_t* __restrict__ order,
                                                  const int32_t* __restrict__ cell_of,
                                                  CO co, int k,
                                                  int exclude_self, int n_total, int last_level,
                                                  int max_ring, int32_t* __restrict__ out_idx,
                                                  double* __restrict__ out_d2,
                                                  int32_t* __restrict__ fail_list,
                                                  int32_t* __restrict__ fail_count) {
  const int lane = threadIdx.x & 63;
  int nq = n_ptr ? *n_ptr : n_query;
  if (n_ptr && nq > n_max) nq = 0;
  const int wpb = int(blockDim.x) >> 6;  // waves per block
  for (int qi = blockIdx.x * wpb + (threadIdx.x >> 6); qi < nq; qi += gridDim.x * wpb) {  // whole waves
  const int p = query_list ? (pos_of ? pos_of[query_list[qi]] : query_list[qi]) : qi;
  const int self = order[p];
  double x, y, z;
  co.get(p, x, y, z);
  const int c = cell_of[p];
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  double bd = __builtin_inf();
  int bi = 0x7FFFFFFF;
  int have = 0;
  double tau_d = __builtin_inf();
  int tau_i = 0x7FFFFFFF;
  bool done = false;
  const int rmax_grid = max(g.nx, max(g.ny, g.nz));
  // A query clamped into the grid from far outside its box (a stray point; robust_box): everything
  // inside the box is further away than this level's rings reach, so the walk could only end on
  // other strays — it goes straight to the next level (in the end k_knn_brute) instead of scanning
  // the border cells' thousands of points.
  if (!last_level) {
    const double ox = fmax(fmax(g.minx - x, x - (g.minx + double(g.nx - 2) * g.cell)), 0.0);
    const double oy = fmax(fmax(g.miny - y, y - (g.miny + double(g.ny - 2) * g.cell)), 0.0);
    const double oz = fmax(fmax(g.minz - z, z - (g.minz + double(g.nz - 2) * g.cell)), 0.0);
    if (fmax(ox, fmax(oy, oz)) > double(max_ring) * g.cell) {
      if (lane == 0) fail_list[atomicAdd(fail_count, 1)] = self;
      continue;
    }
  }
  for (int r = 0; r <= max_ring && !done; ++r) {
    // The shell's row segments, 64 at a time: every lane looks one up (two loads), then the
    // wave scans the non-empty ones together. Around an isolated point nearly all of them
    // are empty, so a wide shell costs a few parallel lookups, not hundreds of serial ones.
    const int w = 2 * r + 1;
    const int nseg = 2 * w * w;  // two slots per row: [full row | -] or [left end | right end]
    for (int sbase = 0; sbase < nseg; sbase += 64) {
      const int sg = sbase + lane;
      int qb_l = 0, qe_l = 0;
      if (sg < nseg) {
        const int ri = sg >> 1, part = sg & 1;
        const int dz = ri / w - r, dy = ri % w - r;
        const int zz = cz + dz, yy = cy + dy;
        const bool full = (dz == -r || dz == r || dy == -r || dy == r);
        int x0 = 0, x1 = -1;
        if (full) {
          if (part == 0) {
            x0 = cx - r;
            x1 = cx + r;
          }
        } else if (r > 0) {
          x0 = x1 = part == 0 ? cx - r : cx + r;
        }
        x0 = x0 < 0 ? 0 : x0;
        x1 = x1 >= g.nx ? g.nx - 1 : x1;
        if (zz >= 0 && zz < g.nz && yy >= 0 && yy < g.ny && x0 <= x1) {
          const int row = (zz * g.ny + yy) * g.nx;
          qb_l = start[row + x0];
          qe_l = start[row + x1 + 1];
        }
      }
      unsigned long long segs = __ballot(qe_l > qb_l);
      while (segs) {
        const int sl = __ffsll(segs) - 1;
        segs &= segs - 1;
        const int qb = __shfl(qb_l, sl, 64), qe = __shfl(qe_l, sl, 64);
        for (int base = qb; base < qe; base += 64) {
          const int q = base + lane;
          const bool valid = q < qe && !(exclude_self && q == p);
          double d = __builtin_inf();
          int id = 0x7FFFFFFF;
          if (valid) {
            d = co.d2(q, x, y, z);
            id = order[q];
          }
          const bool cand = valid && (have < k || d < tau_d || (d == tau_d && id < tau_i));
          unsigned long long mask = __ballot(cand);
          while (mask) {
            const int l = __ffsll(mask) - 1;
            mask &= mask - 1;
            const double nd = __shfl(d, l, 64);
            const int ni = __shfl(id, l, 64);
            if (have == k && !(nd < tau_d || (nd == tau_d && ni < tau_i))) continue;
            const bool less = lane < have && (bd < nd || (bd == nd && bi < ni));
            const int pos = __popcll(__ballot(less));
            const double pd = __shfl_up(bd, 1, 64);
            const int pi = __shfl_up(bi, 1, 64);
            const int top = have < k ? have : k - 1;
            if (lane > pos && lane <= top) {
              bd = pd;
              bi = pi;
            }
            if (lane == pos) {
              bd = nd;
              bi = ni;
            }
            if (have < k) ++have;
            if (have == k) {
              tau_d = __shfl(bd, k - 1, 64);
              tau_i = __shfl(bi, k - 1, 64);
            }
          }
        }
      }
    }
    const double safe = double(r) * g.cell * 0.999999;
    if (have == k && tau_d <= safe * safe) done = true;
    if (r >= rmax_grid) done = true;
  }
  if (!done && !last_level) {
    if (lane == 0) fail_list[atomicAdd(fail_count, 1)] = self;
    continue;
  }
  if (lane < k) {
    out_idx[size_t(self) * k + lane] = lane < have ? bi : n_total;
    if (out_d2) out_d2[size_t(self) * k + lane] = lane < have ? bd : __builtin_inf();
  }
  }
}

// The last resort for a FEW isolated queries (stray points tens of metres outside the scan: their
// neighbours are further away than any affordable ring of cells, and every coarser retry level
// costs a grid build and a launch that lasts as long as its longest wave — 6-8 ms for a
// handful of them, 45 ms when the strays inflate the box until the grid hits its size cap): the
// whole cloud, same list, same comparison. A wave takes kBruteTile queries through one slice of
// the points (every loaded point is tested against all of them; the slices of one tile run on
// different waves, so that twenty queries still fill the chip), leaves its sorted partial lists in
// a scratch buffer, and k_knn_brute_merge folds a query's slices.
// queries it takes: what robust_box may leave outside its box. The cost is queries x points
// (1.4 ms for 2 000 of a million), the alternative a grid at its size cap (45 ms).
static int brute_max(int64_t n) { return int(std::min<int64_t>(8192, std::max<int64_t>(256, n / 256))); }
static constexpr int kBruteTile = 4;    // queries per wave

__device__ __forceinline__ void wave_list_insert(double nd, int ni, int k, int lane, double& bd, int& bi, int& have,
                                                 double& tau_d, int& tau_i) {
  if (have == k && !(nd < tau_d || (nd == tau_d && ni < tau_i))) return;
  const bool less = lane < have && (bd < nd || (bd == nd && bi < ni));
  const int pos = __popcll(__ballot(less));
  const double pd = __shfl_up(bd, 1, 64);
  const int pi = __shfl_up(bi, 1, 64);
  const int top = have < k ? have : k - 1;
  if (lane > pos && lane <= top) {
    bd = pd;
    bi = pi;
  }
  if (lane == pos) {
    bd = nd;
    bi = ni;
  }
  if (have < k) ++have;
  if (have == k) {
    tau_d = __shfl(bd, k - 1, 64);
    tau_i = __shfl(bi, k - 1, 64);
  }
}

__global__ __launch_bounds__(256) void k_knn_brute(int n_query, const int32_t* __restrict__ query_list,
                                                   const double* __restrict__ xyz, int n, int k,
                                                   int exclude_self, int n_slices,
                                                   double* __restrict__ part_d /*[tiles][slices][tile][k]*/,
                                                   int32_t* __restrict__ part_i) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);  // wave = (tile, slice)
  const int tile = w / n_slices, slice = w % n_slices;
  if (tile * kBruteTile >= n_query) return;  // whole waves
  int self[kBruteTile];
  double qx[kBruteTile], qy[kBruteTile], qz[kBruteTile];
  double bd[kBruteTile], tau_d[kBruteTile];
  int bi[kBruteTile], have[kBruteTile], tau_i[kBruteTile];
#pragma unroll
  for (int t = 0; t < kBruteTile; ++t) {
    const int qi = tile * kBruteTile + t;
    self[t] = qi < n_query ? query_list[qi] : -1;
    const size_t s0 = size_t(self[t] < 0 ? 0 : self[t]);
    qx[t] = xyz[3 * s0];
    qy[t] = xyz[3 * s0 + 1];
    qz[t] = xyz[3 * s0 + 2];
    bd[t] = tau_d[t] = __builtin_inf();
    bi[t] = tau_i[t] = 0x7FFFFFFF;
    have[t] = 0;
  }
  // slices of whole 64-point chunks
  const int chunks = (n + 63) / 64;
  const int per = (chunks + n_slices - 1) / n_slices;
  const int c_lo = slice * per, c_hi = min(chunks, c_lo + per);
  for (int ch = c_lo; ch < c_hi; ch += 2) {
    // two chunks' loads in flight before either is looked at
    double px[2], py[2], pz[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int q = (ch + u) * 64 + lane;
      const bool in = ch + u < c_hi && q < n;
      const size_t qq = size_t(in ? q : 0);
      px[u] = xyz[3 * qq];
      py[u] = xyz[3 * qq + 1];
      pz[u] = xyz[3 * qq + 2];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int base = (ch + u) * 64;
      const int q = base + lane;
      const bool in = ch + u < c_hi && q < n;
#pragma unroll
      for (int t = 0; t < kBruteTile; ++t) {
        const bool valid = in && self[t] >= 0 && !(exclude_self && q == self[t]);
        const double d = valid ? sqdist3(qx[t], qy[t], qz[t], px[u], py[u], pz[u]) : __builtin_inf();
        const bool cand = valid && (have[t] < k || d < tau_d[t] || (d == tau_d[t] && q < tau_i[t]));
        unsigned long long mask = __ballot(cand);
        while (mask) {
          const int l = __ffsll(mask) - 1;
          mask &= mask - 1;
          wave_list_insert(__shfl(d, l, 64), base + l, k, lane, bd[t], bi[t], have[t], tau_d[t], tau_i[t]);
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < kBruteTile; ++t) {
    const size_t o = ((size_t(tile) * n_slices + slice) * kBruteTile + t) * size_t(k);
    if (lane < k) {
      part_d[o + lane] = lane < have[t] ? bd[t] : __builtin_inf();
      part_i[o + lane] = lane < have[t] ? bi[t] : 0x7FFFFFFF;
    }
  }
}

__global__ __launch_bounds__(256) void k_knn_brute_merge(int n_query, const int32_t* __restrict__ query_list,
                                                         int n, int k, int n_slices,
                                                         const double* __restrict__ part_d,
                                                         const int32_t* __restrict__ part_i,
                                                         int32_t* __restrict__ out_idx,
                                                         double* __restrict__ out_d2) {
  const int lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= n_query) return;
  const int self = query_list[qi];
  const int tile = qi / kBruteTile, t = qi % kBruteTile;
  double bd = __builtin_inf(), tau_d = __builtin_inf();
  int bi = 0x7FFFFFFF, tau_i = 0x7FFFFFFF, have = 0;
  // 64 / k slices per step, one entry per lane, the next step's loads issued before this step's
  // insertions (one slice at a time the loop was a chain of 256 memory round trips: 0.27 ms)
  const int per = 64 / k > 0 ? 64 / k : 1;
  const int sub = lane / k, e = lane % k;
  auto fetch = [&](int s0, double* d, int* id) {
    const int s = s0 + sub;
    const bool ok = sub < per && s < n_slices;
    const size_t o = ((size_t(tile) * n_slices + (ok ? s : 0)) * kBruteTile + t) * size_t(k) + e;
    *d = ok ? part_d[o] : __builtin_inf();
    *id = ok ? part_i[o] : 0x7FFFFFFF;
  };
  double dn;
  int idn;
  fetch(0, &dn, &idn);
  for (int s = 0; s < n_slices; s += per) {
    const double d = dn;
    const int id = idn;
    if (s + per < n_slices) fetch(s + per, &dn, &idn);
    const bool cand = id != 0x7FFFFFFF && (have < k || d < tau_d || (d == tau_d && id < tau_i));
    unsigned long long mask = __ballot(cand);
    while (mask) {
      const int l = __ffsll(mask) - 1;
      mask &= mask - 1;
      wave_list_insert(__shfl(d, l, 64), __shfl(id, l, 64), k, lane, bd, bi, have, tau_d, tau_i);
    }
  }
  if (lane < k) {
    out_idx[size_t(self) * k + lane] = lane < have ? bi : n;
    if (out_d2) out_d2[size_t(self) * k + lane] = lane < have ? bd : __builtin_inf();
  }
}

__global__ __launch_bounds__(256) void k_invert_order(int n, const int32_t* __restrict__ order,
                                                      int32_t* __restrict__ pos_of) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p < n) pos_of[order[p]] = p;
}

template <int T>
static int launch_knn(Ctx* c, int n_query, const int32_t* list, const int32_t* pos_of,
                      const DevGrid& g, int k, int excl, int n_total, int last, int32_t* idx,
                      double* d2, int32_t* fail_list, int32_t* fail_count) {
  const size_t smem = size_t(k) * T * 12;
  static std::atomic<uint64_t> attr_set{0};  // one bit per device (the attribute is per device)
  const uint64_t bit = 1ull << (c->device & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn<T, CoordsF64>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_knn<T, CoordsF32>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set.fetch_or(bit, std::memory_order_release);
  }
  KnnGrid kg{g.nx, g.ny, g.nz, g.cell, g.minx, g.miny, g.minz};
  on_coords(g, [&](auto co) {
    hipLaunchKernelGGL((k_knn<T, decltype(co)>), dim3(ceil_div(n_query, T)), dim3(T), smem, c->stream, n_query, list,
                       pos_of, kg, g.start, g.order, g.cell_of, co, k, excl, n_total, last, idx, d2, fail_list,
                       fail_count);
  });
  PQ_HIP(hipGetLastError());
  return 0;
}

int knn_device(Ctx* c, const double* xyz, int64_t n, int32_t k, int32_t exclude_self,
               int32_t* idx, double* d2) {
  if (k <= 0 || k > kMaxK) return fail(PYQSM_ERANGE, "k must be in [1, %d]", kMaxK);
  if (n == 0) return 0;
  const int N = int(n);
  // ---- level-0 cell edge: aim at ~k/3 points per occupied cell -----------
  // Two count-only probes on coarse grids give the local density and how it
  // scales with the edge (exponent ~2 for surfaces, ~3 for volumes); the edge
  // for the target occupancy follows by extrapolation, and one corrective
  // rebuild is allowed if the sorted grid misses the target by more than 2.5x.
  DevGrid g;
  const int64_t max_cells = int64_t(1) << 28;
  double box[6];
  bool all_f32 = false;  // every coordinate fp32-representable: the grids keep 16-byte fp32 records
  {
    ProfScope ps(c, "knn_bin");
    PQ_TRY(cloud_bbox(c, xyz, n, box, box + 3, &all_f32));
    // grids over the cloud without its sparse tails; what is given up is what k_knn_brute can take
    const char* rbe = getenv("PYQSM_KNN_ROBUST_BOX");  // "0": grids over the full box
    const bool robust = !(rbe && rbe[0] == '0');
    if (robust && k <= 64) {
      int64_t outside = 0;
      PQ_TRY(robust_box(c, xyz, n, brute_max(n), box, &outside));
      if (outside && getenv("PYQSM_KNN_TRACE"))
        fprintf(stderr, "knn: grids over the box without its tails, at most %lld points outside\n", (long long)outside);
    }
    double ext = std::max(box[3] - box[0], std::max(box[4] - box[1], box[5] - box[2]));
    if (!(ext > 0)) ext = 1.0;
    static const double occ_div = [] {  // PYQSM_KNN_OCC: points per occupied cell = k / this
      const char* e = getenv("PYQSM_KNN_OCC");
      const double v = e ? atof(e) : 3.0;
      return v > 0.25 && v < 64.0 ? v : 3.0;
    }();
    const double target = std::max(2.0, double(k) / occ_div);
    double c1, per1, c2, per2, dim = 2.0, cell;
    PQ_TRY(probe_occupancy(c, xyz, n, box, ext / 64.0, &c1, &per1));
    if (per1 <= 1.5 * target) {
      cell = c1 * (per1 < 0.5 * target ? 2.0 : 1.0);
    } else {
      PQ_TRY(probe_occupancy(c, xyz, n, box, c1 * 0.5, &c2, &per2));
      if (c2 < c1 && per2 > 0) dim = std::log2(std::max(per1 / per2, 1.0001)) / std::log2(c1 / c2);
      dim = std::min(3.0, std::max(1.0, dim));
      cell = c2 * std::pow(target / per2, 1.0 / dim);
    }
    cell = std::max(cell, ext / 4096.0);
    for (int it = 0; it < 2; ++it) {
      PQ_TRY(build_grid(c, xyz, n, cell, max_cells, &g, box, all_f32));
      if (it == 1) break;
      int64_t occ = 0;
      PQ_TRY(count_occupied(c, g, &occ));
      const double per = double(n) / double(occ > 0 ? occ : 1);
      if (per <= 2.5 * target && per >= 0.4 * target) break;
      if (per < target && g.nx <= 3 && g.ny <= 3 && g.nz <= 3) break;  // cannot coarsen further
      const double f = std::min(4.0, std::max(0.25, std::pow(target / per, 1.0 / dim)));
      if (f < 1.0 && g.cell > cell * 1.0000001) break;  // grid already at its size cap
      cell = g.cell * f;
    }
  }
  int32_t *pos_of = nullptr, *fail_a, *fail_b, *fail_count;
  PQ_TRY(c->arena.get(size_t(n), &fail_a));
  PQ_TRY(c->arena.get(size_t(n), &fail_b));
  PQ_TRY(c->arena.get(1, &fail_count));
  int32_t *tie_list, *tie_count;
  PQ_TRY(c->arena.get(size_t(n), &tie_list));
  PQ_TRY(c->arena.get(1, &tie_count));
  PQ_HIP(hipMemsetAsync(tie_count, 0, 4, c->stream));
  int n_query = N;
  const int32_t* list = nullptr;
  for (int level = 0;; ++level) {
    static const int wide = [] { const char* e = getenv("PYQSM_KNN_WIDE"); return e ? atoi(e) : kWideRing; }();
    const int ring = level >= 1 && k <= 64 ? wide : kMaxRing;
    const int last = (g.nx - 2 <= ring && g.ny - 2 <= ring && g.nz - 2 <= ring) ? 1 : 0;
    PQ_HIP(hipMemsetAsync(fail_count, 0, 4, c->stream));
    {
      ProfScope ps(c, level == 0 ? "knn_search" : "knn_retry");
      int32_t* fl = (level & 1) ? fail_b : fail_a;
      if (level == 0 && k <= 32) {
        // list length = k rounded up to a multiple of 4: every slot is ~11 instructions per
        // insertion, and slots beyond k loosen the acceptance test
        switch ((k + 3) / 4) {
          case 1: PQ_TRY(launch_knn_reg<4>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 2: PQ_TRY(launch_knn_reg<8>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 3: PQ_TRY(launch_knn_reg<12>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 4: PQ_TRY(launch_knn_reg<16>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 5: PQ_TRY(launch_knn_reg<20>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 6: PQ_TRY(launch_knn_reg<24>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          case 7: PQ_TRY(launch_knn_reg<28>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
          default: PQ_TRY(launch_knn_reg<32>(c, N, g, k, exclude_self, last, idx, d2, fl, fail_count, tie_list, tie_count)); break;
        }
      } else if (level > 0 && k <= 64) {
        KnnGrid kg{g.nx, g.ny, g.nz, g.cell, g.minx, g.miny, g.minz};
        // (one or two waves per block instead of four: no difference — the launch lasts as long as its
        // longest waves, 0.34 ms for the forest's 9 864 outliers)
        on_coords(g, [&](auto co) {
          hipLaunchKernelGGL(k_knn_wave<decltype(co)>, dim3(ceil_div(n_query, 4)), dim3(256), 0, c->stream, n_query,
                             static_cast<const int32_t*>(nullptr), 0, list, pos_of, kg, g.start, g.order, g.cell_of, co,
                             k, exclude_self, N, last, ring, idx, d2, fl, fail_count);
        });
        PQ_HIP(hipGetLastError());
      } else if (k <= 48)
        PQ_TRY(launch_knn<256>(c, n_query, list, pos_of, g, k, exclude_self, N, last, idx, d2, fl,
                               fail_count));
      else if (k <= 96)
        PQ_TRY(launch_knn<128>(c, n_query, list, pos_of, g, k, exclude_self, N, last, idx, d2, fl,
                               fail_count));
      else
        PQ_TRY(launch_knn<64>(c, n_query, list, pos_of, g, k, exclude_self, N, last, idx, d2, fl,
                              fail_count));
    }
    if (last) break;
    int32_t nf = 0;
    PQ_HIP(hipMemcpyAsync(&nf, fail_count, 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    if (level == 0 && getenv("PYQSM_KNN_TRACE")) {
      int32_t nt = 0;
      PQ_HIP(hipMemcpy(&nt, tie_count, 4, hipMemcpyDeviceToHost));
      fprintf(stderr, "knn level 0: %d queries searched again for a tie at the k-th place\n", nt);
    }
    if (getenv("PYQSM_KNN_TRACE")) fprintf(stderr, "knn level %d: %d of %d queries stay open (cell %.4g)\n", level, nf, n_query, g.cell);
    if (nf == 0) break;
    if (level >= 1 && nf <= 2 * brute_max(n) && k <= 64) {  // a few isolated queries: one by one
      ProfScope pb(c, "knn_brute");
      const int32_t* ql = (level & 1) ? fail_b : fail_a;
      const int tiles = ceil_div(nf, kBruteTile);
      const int chunks = ceil_div(N, 64);
      // ~16 waves per CU over all tiles; a slice is at least 32 chunks long
      const int n_slices = std::max(1, std::min(std::min(256, ceil_div(chunks, 32)), ceil_div(c->cu_count * 16, tiles)));
      double* part_d = nullptr;
      int32_t* part_i = nullptr;
      const size_t slots = size_t(tiles) * n_slices * kBruteTile * size_t(k);
      PQ_TRY(c->arena.get(slots, &part_d));
      PQ_TRY(c->arena.get(slots, &part_i));
      hipLaunchKernelGGL(k_knn_brute, dim3(ceil_div(tiles * n_slices, 4)), dim3(256), 0, c->stream, nf, ql, xyz, N,
                         k, exclude_self, n_slices, part_d, part_i);
      hipLaunchKernelGGL(k_knn_brute_merge, dim3(ceil_div(nf, 4)), dim3(256), 0, c->stream, nf, ql, N, k, n_slices,
                         static_cast<const double*>(part_d), static_cast<const int32_t*>(part_i), idx, d2);
      PQ_HIP(hipGetLastError());
      if (getenv("PYQSM_KNN_TRACE")) fprintf(stderr, "knn: %d isolated queries searched over the whole cloud\n", nf);
      break;
    }
    // retry the stragglers on a 4x coarser grid
    ProfScope ps(c, "knn_bin");
    list = (level & 1) ? fail_b : fail_a;
    n_query = nf;
    {
      // Deriving the coarse grid from the fine one reads the fine grid's whole cell array (95 M
      // cells for a million points at k = 20: 0.22 ms); binning the points again at the coarse edge
      // touches 1.5 M cells (0.12 ms). The order inside a cell does not matter to a search. Small
      // fine grids (the later levels) keep the pyramid: no atomics, a few microseconds.
      DevGrid coarse;
      if (g.ncell > (int64_t(1) << 22))
        PQ_TRY(build_grid(c, xyz, n, g.cell * 4.0, max_cells, &coarse, box, all_f32));
      else
        PQ_TRY(coarsen_grid(c, g, n, 4, &coarse));
      g = coarse;
    }
    if (!pos_of) PQ_TRY(c->arena.get(size_t(n), &pos_of));
    hipLaunchKernelGGL(k_invert_order, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, N, g.order,
                       pos_of);
    PQ_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_knn_dev(const double* xyz_dev, int64_t n, int32_t k, int32_t exclude_self,
                  int32_t* idx_dev, double* d2_dev, int32_t device) {
  PQ_API_RANGE("pyqsm_knn_dev");
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n > 0 && (!xyz_dev || !idx_dev || !d2_dev))
    return fail(PYQSM_EINVAL, "pyqsm_knn_dev: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  ProfScope ps(c, "knn_total");
  return knn_device(c, xyz_dev, n, k, exclude_self, idx_dev, d2_dev);
}

int pyqsm_knn(const double* xyz, int64_t n, int32_t k, int32_t exclude_self, int32_t* idx,
              double* d2, int32_t device) {
  PQ_API_RANGE("pyqsm_knn");
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n == 0) return 0;
  if (!xyz || !idx || !d2) return fail(PYQSM_EINVAL, "pyqsm_knn: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  double *d_xyz, *d_d2;
  int32_t* d_idx;
  PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
  PQ_TRY(c->arena.get(size_t(n) * size_t(k > 0 ? k : 1), &d_idx));
  PQ_TRY(c->arena.get(size_t(n) * size_t(k > 0 ? k : 1), &d_d2));
  PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_TRY(knn_device(c, d_xyz, n, k, exclude_self, d_idx, d_d2));
  PQ_HIP(hipMemcpyAsync(idx, d_idx, size_t(n) * k * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(d2, d_d2, size_t(n) * k * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
