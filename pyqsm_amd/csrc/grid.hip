// grid.hip — bounding box, cell counting sort (see grid.hpp).
#include "grid.hpp"

#include <atomic>
#include <cmath>

namespace pyqsm {

// Order-preserving map double -> uint64 so that atomicMin/atomicMax work.
__device__ __forceinline__ unsigned long long ord_key(double x) {
  unsigned long long b = static_cast<unsigned long long>(__double_as_longlong(x));
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static double ord_val(unsigned long long k) {
  unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
  double d;
  memcpy(&d, &b, 8);
  return d;
}

// Stage 1: one (min, max) record per block, no atomics (94 k same-address atomics
// cost 0.56 ms in the first version of this kernel). Stage 2: one block folds the
// records.
// part[block][7]: min keys, max keys, and 1 if every coordinate the block saw is exactly
// representable in fp32 (double(float(v)) == v: true for PCD / LAS derived clouds), else 0
static constexpr int kBoxVals = 7;
__global__ __launch_bounds__(256) void k_bbox(const double* __restrict__ xyz, int64_t n,
                                              unsigned long long* __restrict__ part /*[grid][7]*/) {
  __shared__ unsigned long long red[4][6];
  __shared__ int red_f32[4];
  unsigned long long mn[3] = {~0ull, ~0ull, ~0ull}, mx[3] = {0, 0, 0};
  bool f32ok = true;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < n;
       i += int64_t(gridDim.x) * blockDim.x) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double v = xyz[3 * i + a];
      f32ok = f32ok && double(float(v)) == v;
      unsigned long long k = ord_key(v);
      mn[a] = k < mn[a] ? k : mn[a];
      mx[a] = k > mx[a] ? k : mx[a];
    }
  }
  const bool wave_ok = __all(f32ok);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      unsigned long long o1 = __shfl_down(mn[a], off, 64), o2 = __shfl_down(mx[a], off, 64);
      mn[a] = o1 < mn[a] ? o1 : mn[a];
      mx[a] = o2 > mx[a] ? o2 : mx[a];
    }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      red[w][a] = mn[a];
      red[w][3 + a] = mx[a];
    }
    red_f32[w] = wave_ok;
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    unsigned long long v = red[0][threadIdx.x];
    for (int q = 1; q < 4; ++q) {
      const unsigned long long o = red[q][threadIdx.x];
      v = threadIdx.x < 3 ? (o < v ? o : v) : (o > v ? o : v);
    }
    part[size_t(blockIdx.x) * kBoxVals + threadIdx.x] = v;
  }
  if (threadIdx.x == 6)
    part[size_t(blockIdx.x) * kBoxVals + 6] = (red_f32[0] && red_f32[1] && red_f32[2] && red_f32[3]) ? 1ull : 0ull;
}

// (Folding in k_bbox itself — the block that counts itself last reads everybody's records — was tried
// in round 3 to save this launch: the agent-scope release every block needs before it counts itself
// writes its XCD's L2 back, and the binning went from 0.135 to 0.20 ms.)
// One block folds the per-block records (the fp32 flag as a minimum: 0 wins) and, while it is
// there, clears `zero_n` ints for the caller (the bucket counters of the binning that follows:
// two memset launches less on the critical path).
__global__ __launch_bounds__(256) void k_bbox_fold(const unsigned long long* __restrict__ part,
                                                   int nblk, unsigned long long* __restrict__ out,
                                                   int32_t* __restrict__ zero_buf, int zero_n) {
  __shared__ unsigned long long red[4][kBoxVals];
  for (int q = threadIdx.x; q < zero_n; q += 256) zero_buf[q] = 0;
  unsigned long long v[kBoxVals] = {~0ull, ~0ull, ~0ull, 0, 0, 0, ~0ull};
  for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
    for (int a = 0; a < kBoxVals; ++a) {
      const unsigned long long o = part[size_t(b) * kBoxVals + a];
      v[a] = (a < 3 || a == 6) ? (o < v[a] ? o : v[a]) : (o > v[a] ? o : v[a]);
    }
#pragma unroll
  for (int a = 0; a < kBoxVals; ++a)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_down(v[a], off, 64);
      v[a] = (a < 3 || a == 6) ? (o < v[a] ? o : v[a]) : (o > v[a] ? o : v[a]);
    }
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int a = 0; a < kBoxVals; ++a) red[threadIdx.x >> 6][a] = v[a];
  __syncthreads();
  if (threadIdx.x < kBoxVals) {
    const bool is_min = threadIdx.x < 3 || threadIdx.x == 6;
    unsigned long long r = red[0][threadIdx.x];
    for (int q = 1; q < 4; ++q) {
      const unsigned long long o = red[q][threadIdx.x];
      r = is_min ? (o < r ? o : r) : (o > r ? o : r);
    }
    out[threadIdx.x] = r;
  }
}

struct GridParams {
  double minx, miny, minz, inv_cell;
  int nx, ny, nz;
};

__device__ __forceinline__ int cell_index(const GridParams& g, double x, double y, double z) {
  // interior cells are 1 .. n-2; the clamp guards the max-boundary point and puts points outside
  // the grid's box into its outermost cells — clamped in double, so that a point a light year
  // away does not overflow the int
  const double fx = floor((x - g.minx) * g.inv_cell), fy = floor((y - g.miny) * g.inv_cell),
               fz = floor((z - g.minz) * g.inv_cell);
  const int cx = int(fmin(fmax(fx, 0.0), double(g.nx - 3))) + 1;
  const int cy = int(fmin(fmax(fy, 0.0), double(g.ny - 3))) + 1;
  const int cz = int(fmin(fmax(fz, 0.0), double(g.nz - 3))) + 1;
  return (cz * g.ny + cy) * g.nx + cx;
}

__global__ __launch_bounds__(256) void k_cell_count(const double* __restrict__ xyz, int64_t n,
                                                    GridParams g, int32_t* __restrict__ counts,
                                                    int32_t* __restrict__ cell_tmp,
                                                    int32_t* __restrict__ rank_tmp,
                                                    int32_t* __restrict__ occ_part /*[gridDim.x]*/) {
  __shared__ int wfirst[4];
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  bool first = false;
  if (i < n) {
    int c = cell_index(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    cell_tmp[i] = c;
    const int rank = atomicAdd(&counts[c], 1);
    rank_tmp[i] = rank;
    first = rank == 0;  // exactly one point per occupied cell sees an empty counter
  }
  const unsigned long long b = __ballot(first);
  if ((threadIdx.x & 63) == 0) wfirst[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) occ_part[blockIdx.x] = (wfirst[0] + wfirst[1]) + (wfirst[2] + wfirst[3]);
}

__global__ __launch_bounds__(256) void k_cell_scatter(const double* __restrict__ xyz, int64_t n,
                                                      const int32_t* __restrict__ start,
                                                      const int32_t* __restrict__ cell_tmp,
                                                      const int32_t* __restrict__ rank_tmp,
                                                      int32_t* __restrict__ order,
                                                      int32_t* __restrict__ cell_of,
                                                      double* __restrict__ sx,
                                                      double* __restrict__ sy,
                                                      double* __restrict__ sz) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  int c = cell_tmp[i];
  int p = start[c] + rank_tmp[i];
  order[p] = int32_t(i);
  cell_of[p] = c;
  sx[p] = xyz[3 * i];
  sy[p] = xyz[3 * i + 1];
  sz[p] = xyz[3 * i + 2];
}

// wave sums -> one atomic per block (atomics on one address are served one at a time)
__device__ __forceinline__ void block_add_i32(int32_t local, int32_t* out) {
  __shared__ int32_t wsum[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int32_t t = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (t) atomicAdd(out, t);
  }
}

__global__ __launch_bounds__(256) void k_count_occupied(const int32_t* __restrict__ start,
                                                        int64_t ncell, int32_t* __restrict__ out) {
  int32_t local = 0;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < ncell;
       i += int64_t(gridDim.x) * blockDim.x)
    local += start[i + 1] > start[i];
  block_add_i32(local, out);
}

__global__ __launch_bounds__(256) void k_probe_count(const double* __restrict__ xyz, int64_t n,
                                                     GridParams g, int32_t* __restrict__ counts) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  // only "occupied or not" is read afterwards: a plain store of 1 (racing stores of the same
  // value) instead of a counter — hundreds of points share a probe cell, and atomics on one
  // address are served one at a time
  counts[cell_index(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2])] = 1;
}

__global__ __launch_bounds__(256) void k_count_nonzero(const int32_t* __restrict__ counts,
                                                       int64_t ncell, int32_t* __restrict__ out) {
  int32_t local = 0;
  for (int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; i < ncell;
       i += int64_t(gridDim.x) * blockDim.x)
    local += counts[i] > 0;
  block_add_i32(local, out);
}

int probe_occupancy(Ctx* c, const double* xyz, int64_t n, const double box[6], double cell,
                    double* cell_used, double* per_cell) {
  int dims[3];
  for (;;) {
    double tot = 1.0;
    for (int a = 0; a < 3; ++a) {
      double d = std::floor((box[3 + a] - box[a]) / cell) + 3.0;
      dims[a] = d < 2.0e9 ? int(d) : 0x7FFFFFFF;
      tot *= d;
    }
    if (tot <= double(int64_t(1) << 22)) break;
    cell *= 2.0;
  }
  const int64_t ncell = int64_t(dims[0]) * dims[1] * dims[2];
  int32_t* counts;
  PQ_TRY(c->arena.get(size_t(ncell) + 1, &counts));
  PQ_HIP(hipMemsetAsync(counts, 0, (size_t(ncell) + 1) * 4, c->stream));
  GridParams gp{box[0], box[1], box[2], 1.0 / cell, dims[0], dims[1], dims[2]};
  hipLaunchKernelGGL(k_probe_count, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, xyz, n, gp,
                     counts);
  const int blocks = int(std::min<int64_t>(ceil_div(ncell, 256), int64_t(c->cu_count) * 8));
  hipLaunchKernelGGL(k_count_nonzero, dim3(blocks), dim3(256), 0, c->stream, counts, ncell,
                     counts + ncell);
  PQ_HIP(hipGetLastError());
  int32_t occ = 0;
  PQ_HIP(hipMemcpyAsync(&occ, counts + ncell, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  *cell_used = cell;
  *per_cell = double(n) / double(occ > 0 ? occ : 1);
  return 0;
}

// ---- a box without the cloud's sparse tails --------------------------------------------
// A scan with a few stray returns tens of metres outside would have its search grids sized for
// mostly empty space (a million points at k = 20: up to 50 ms instead of 2.3). Binning CLAMPS into the
// outermost cells (cell_index), and a clamp moves no two points further apart, so a grid over a
// smaller box stays exact for searches that bound distances from below by cell rings; what it
// costs is that the clamped points themselves find their neighbours late. The box is therefore
// cut back only by as many points as the search can afford to treat one by one (`budget`).

static constexpr int kAxisBins = 64;
static constexpr int kAxisBlocks = 512;
static constexpr int kAxisSample = 4;

__global__ __launch_bounds__(256) void k_axis_hist(const double* __restrict__ xyz, int64_t n, double mnx,
                                                   double mny, double mnz, double ix, double iy, double iz,
                                                   int32_t* __restrict__ part /*[gridDim.x][3 * kAxisBins]*/) {
  __shared__ int32_t h[3 * kAxisBins];
  for (int t = threadIdx.x; t < 3 * kAxisBins; t += 256) h[t] = 0;
  __syncthreads();
  // every kAxisSample-th point (counted kAxisSample times): the LDS atomics of neighbouring points
  // land on the same bins and are served one at a time — all points cost 0.09 ms per million; a
  // stray the sample misses leaves its bin empty, which is what lets the cut pass it
  for (int64_t i = (blockIdx.x * 256ll + threadIdx.x) * kAxisSample; i < n;
       i += int64_t(gridDim.x) * 256 * kAxisSample) {
    const double bx = floor((xyz[3 * i] - mnx) * ix), by = floor((xyz[3 * i + 1] - mny) * iy),
                 bz = floor((xyz[3 * i + 2] - mnz) * iz);
    atomicAdd(&h[int(fmin(fmax(bx, 0.0), double(kAxisBins - 1)))], kAxisSample);
    atomicAdd(&h[kAxisBins + int(fmin(fmax(by, 0.0), double(kAxisBins - 1)))], kAxisSample);
    atomicAdd(&h[2 * kAxisBins + int(fmin(fmax(bz, 0.0), double(kAxisBins - 1)))], kAxisSample);
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 3 * kAxisBins; t += 256) part[size_t(blockIdx.x) * 3 * kAxisBins + t] = h[t];
}

// one wave per bin
__global__ __launch_bounds__(64) void k_axis_fold(const int32_t* __restrict__ part, int nblk,
                                                  int32_t* __restrict__ out) {
  const int t = blockIdx.x;
  int32_t s = 0;
  for (int b = threadIdx.x; b < nblk; b += 64) s += part[size_t(b) * 3 * kAxisBins + t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (threadIdx.x == 0) out[t] = s;
}

int robust_box(Ctx* c, const double* xyz, int64_t n, int budget, double box[6], int64_t* outside) {
  *outside = 0;
  if (budget <= 0 || n < 64) return 0;
  const int blocks = int(std::min<int64_t>(ceil_div(n, 256 * kAxisSample), kAxisBlocks));
  int32_t *d_part = nullptr, *d_hist = nullptr;
  PQ_TRY(c->arena.get(size_t(blocks) * 3 * kAxisBins, &d_part));
  PQ_TRY(c->arena.get(size_t(3 * kAxisBins), &d_hist));
  // points one side of one axis may lose: a stray in a corner is counted on three sides, strays
  // on one side of the scan on one — the union stays below 2 x budget, which is what the caller
  // accepts
  const int side = budget / 3;
  int64_t lost_axis[3] = {0, 0, 0};  // (a later cut counts the earlier one's points again: they sit in its end bins)
  for (int round = 0; round < 6; ++round) {
    double inv[3];
    for (int a = 0; a < 3; ++a) {
      const double e = box[3 + a] - box[a];
      inv[a] = e > 0 ? double(kAxisBins) / e : 0.0;
    }
    hipLaunchKernelGGL(k_axis_hist, dim3(blocks), dim3(256), 0, c->stream, xyz, n, box[0], box[1], box[2],
                       inv[0], inv[1], inv[2], d_part);
    hipLaunchKernelGGL(k_axis_fold, dim3(3 * kAxisBins), dim3(64), 0, c->stream, d_part, blocks, d_hist);
    PQ_HIP(hipGetLastError());
    int32_t h[3 * kAxisBins];
    PQ_HIP(hipMemcpyAsync(h, d_hist, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    bool cut = false;
    for (int a = 0; a < 3; ++a) {
      const int32_t* ha = h + a * kAxisBins;
      int lo = 0, hi = kAxisBins - 1;
      int64_t acc = 0;
      while (lo < hi && acc + ha[lo] <= side) acc += ha[lo++];
      int64_t lost_a = acc;
      acc = 0;
      while (hi > lo && acc + ha[hi] <= side) acc += ha[hi--];
      lost_a += acc;
      // worth it only when the axis loses a quarter of its length
      if (hi - lo + 1 > (3 * kAxisBins) / 4 || !(inv[a] > 0)) continue;
      const double w = (box[3 + a] - box[a]) / double(kAxisBins);
      const double mn = box[a];
      box[a] = mn + w * double(lo);
      box[3 + a] = mn + w * double(hi + 1);
      lost_axis[a] = lost_a;
      cut = true;
    }
    if (!cut) break;
    *outside = (lost_axis[0] + lost_axis[1]) + lost_axis[2];
  }
  return 0;
}

int cloud_bbox(Ctx* c, const double* xyz, int64_t n, double mn[3], double mx[3], bool* all_f32,
               int32_t* zero_buf, int zero_n) {
  if (n <= 0) return fail(PYQSM_EINVAL, "bounding box of an empty cloud");
  const int blocks = int(std::min<int64_t>(ceil_div(n, 256), int64_t(c->cu_count) * 4));
  unsigned long long *d_box = nullptr, *d_part = nullptr;
  PQ_TRY(c->arena.get(kBoxVals, &d_box));
  PQ_TRY(c->arena.get(size_t(blocks) * kBoxVals, &d_part));
  hipLaunchKernelGGL(k_bbox, dim3(blocks), dim3(256), 0, c->stream, xyz, n, d_part);
  hipLaunchKernelGGL(k_bbox_fold, dim3(1), dim3(256), 0, c->stream, d_part, blocks, d_box, zero_buf, zero_n);
  PQ_HIP(hipGetLastError());
  unsigned long long h_box[kBoxVals];
  PQ_HIP(hipMemcpyAsync(h_box, d_box, sizeof(h_box), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (all_f32) *all_f32 = h_box[6] != 0;
  for (int a = 0; a < 3; ++a) {
    mn[a] = ord_val(h_box[a]);
    mx[a] = ord_val(h_box[3 + a]);
    if (!std::isfinite(mn[a]) || !std::isfinite(mx[a]))
      return fail(PYQSM_EINVAL, "point coordinates must be finite");
  }
  return 0;
}

static int build_grid_bucketed(Ctx* c, const double* xyz, int64_t n, DevGrid* g);
static bool bucketed_fits(const DevGrid& g);

int build_grid(Ctx* c, const double* xyz, int64_t n, double min_cell, int64_t max_cells,
               DevGrid* g, const double* bbox, bool f32_records) {
  if (n <= 0) return fail(PYQSM_EINVAL, "build_grid: empty cloud");
  if (n > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "more than 2^31 points per call");
  if (!(min_cell > 0) || !std::isfinite(min_cell))
    return fail(PYQSM_EINVAL, "cell edge must be positive and finite");
  double mn[3], mx[3];
  if (bbox) {
    for (int a = 0; a < 3; ++a) {
      mn[a] = bbox[a];
      mx[a] = bbox[3 + a];
    }
  } else {
    PQ_TRY(cloud_bbox(c, xyz, n, mn, mx));
  }
  double cell = min_cell;
  int dims[3];
  for (;;) {
    double tot = 1.0;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      double d = std::floor((mx[a] - mn[a]) / cell) + 3.0;  // +1 interior, +2 border
      if (!(d < 2.0e9)) ok = false;
      dims[a] = ok ? int(d) : 0;
      tot *= d;
    }
    if (ok && tot <= double(max_cells)) break;
    cell *= 2.0;
  }
  g->minx = mn[0];
  g->miny = mn[1];
  g->minz = mn[2];
  g->cell = cell;
  g->inv_cell = 1.0 / cell;
  g->nx = dims[0];
  g->ny = dims[1];
  g->nz = dims[2];
  g->ncell = int64_t(dims[0]) * dims[1] * dims[2];
  int32_t *cell_tmp, *rank_tmp;
  PQ_TRY(c->arena.get(size_t(g->ncell) + 1, &g->start));
  PQ_TRY(c->arena.get(size_t(n), &g->order));
  PQ_TRY(c->arena.get(size_t(n), &g->cell_of));
  // two-level counting sort: no scattered atomics, no memset and no scan of the directory; with
  // it, fp32 records instead of three fp64 arrays when the caller vouches for the input
  // (cloud_bbox's flag) and reads through on_coords()
  const bool bucketed = bucketed_fits(*g);
  g->p4 = nullptr;
  g->sx = g->sy = g->sz = nullptr;
  {
    const char* f32_env = getenv("PYQSM_COORD_F32");  // "0": keep fp64 storage (A/B comparisons)
    if (f32_env && !strcmp(f32_env, "0")) f32_records = false;
  }
  if (bucketed && f32_records) {
    PQ_TRY(c->arena.get(size_t(n), &g->p4));
  } else {
    PQ_TRY(c->arena.get(size_t(n), &g->sx));
    PQ_TRY(c->arena.get(size_t(n), &g->sy));
    PQ_TRY(c->arena.get(size_t(n), &g->sz));
  }
  if (bucketed) return build_grid_bucketed(c, xyz, n, g);
  PQ_TRY(c->arena.get(size_t(n), &cell_tmp));
  PQ_TRY(c->arena.get(size_t(n), &rank_tmp));
  PQ_HIP(hipMemsetAsync(g->start, 0, (size_t(g->ncell) + 1) * 4, c->stream));
  GridParams gp{g->minx, g->miny, g->minz, g->inv_cell, g->nx, g->ny, g->nz};
  const dim3 grid(ceil_div(n, 256));
  g->occ_blocks = int(grid.x);
  PQ_TRY(c->arena.get(size_t(g->occ_blocks), &g->occ_part));
  hipLaunchKernelGGL(k_cell_count, grid, dim3(256), 0, c->stream, xyz, n, gp, g->start, cell_tmp,
                     rank_tmp, g->occ_part);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, g->start, g->ncell + 1));
  hipLaunchKernelGGL(k_cell_scatter, grid, dim3(256), 0, c->stream, xyz, n, g->start, cell_tmp,
                     rank_tmp, g->order, g->cell_of, g->sx, g->sy, g->sz);
  PQ_HIP(hipGetLastError());
  return 0;
}


// ---- octant sub-cells ------------------------------------------------------------------

__device__ __forceinline__ int octant_bit(double v, double mn, double inv_cell, int ccoord) {
  // position in half cells relative to the cell the point was binned into
  const int h = int(floor((v - mn) * (2.0 * inv_cell))) - 2 * (ccoord - 1);
  return h < 1 ? 0 : 1;
}

__global__ __launch_bounds__(256) void k_sub_count(int n, GridParams g,
                                                   const int32_t* __restrict__ start,
                                                   const int32_t* __restrict__ cell_of,
                                                   const double* __restrict__ sx,
                                                   const double* __restrict__ sy,
                                                   const double* __restrict__ sz,
                                                   int32_t* __restrict__ sub_cnt,
                                                   int32_t* __restrict__ oct_rank) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int c = cell_of[p];
  const int cx = c % g.nx, cy = (c / g.nx) % g.ny, cz = c / (g.nx * g.ny);
  const int oct = octant_bit(sx[p], g.minx, g.inv_cell, cx) |
                  (octant_bit(sy[p], g.miny, g.inv_cell, cy) << 1) |
                  (octant_bit(sz[p], g.minz, g.inv_cell, cz) << 2);
  const int rank = atomicAdd(&sub_cnt[size_t(start[c]) * 8 + oct], 1);
  oct_rank[p] = (rank << 3) | oct;
}

// the first point of every cell turns its eight counts into run starts
__global__ __launch_bounds__(256) void k_sub_prefix(int n, const int32_t* __restrict__ start,
                                                    const int32_t* __restrict__ cell_of,
                                                    const int32_t* __restrict__ sub_cnt,
                                                    int32_t* __restrict__ sub_beg,
                                                    int4* __restrict__ rec) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n || start[cell_of[p]] != p) return;
  int run = p;
#pragma unroll
  for (int o = 0; o < 8; ++o) {
    const int cnt = sub_cnt[size_t(p) * 8 + o];
    sub_beg[size_t(p) * 8 + o] = run;
    rec[size_t(p) * 8 + o] = make_int4(run, cnt, -1, 0);
    run += cnt;
  }
}

__global__ __launch_bounds__(256) void k_sub_scatter(int n, const int32_t* __restrict__ start,
                                                     const int32_t* __restrict__ cell_of,
                                                     const int32_t* __restrict__ oct_rank,
                                                     const int32_t* __restrict__ sub_beg,
                                                     const int32_t* __restrict__ order,
                                                     const double* __restrict__ sx,
                                                     const double* __restrict__ sy,
                                                     const double* __restrict__ sz,
                                                     int32_t* __restrict__ order2,
                                                     double* __restrict__ sx2,
                                                     double* __restrict__ sy2,
                                                     double* __restrict__ sz2,
                                                     int32_t* __restrict__ sub_of) {
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int orc = oct_rank[p];
  const int sid = start[cell_of[p]] * 8 + (orc & 7);
  const int q = sub_beg[sid] + (orc >> 3);
  order2[q] = order[p];
  sx2[q] = sx[p];
  sy2[q] = sy[p];
  sz2[q] = sz[p];
  sub_of[q] = sid;
}

int subsort_octants(Ctx* c, DevGrid* g, int64_t n, SubCells* sub) {
  if (n > (int64_t(1) << 27)) return fail(PYQSM_ERANGE, "octant sub-cells: more than 2^27 points");
  int32_t *oct_rank, *order2;
  double *sx2, *sy2, *sz2;
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->sub_cnt));
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->sub_beg));
  PQ_TRY(c->arena.get(size_t(n), &sub->sub_of));
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->rec));
  PQ_TRY(c->arena.get(size_t(n), &oct_rank));
  PQ_TRY(c->arena.get(size_t(n), &order2));
  PQ_TRY(c->arena.get(size_t(n), &sx2));
  PQ_TRY(c->arena.get(size_t(n), &sy2));
  PQ_TRY(c->arena.get(size_t(n), &sz2));
  PQ_HIP(hipMemsetAsync(sub->sub_cnt, 0, size_t(n) * 32, c->stream));
  GridParams gp{g->minx, g->miny, g->minz, g->inv_cell, g->nx, g->ny, g->nz};
  const dim3 grid(ceil_div(n, 256)), blk(256);
  hipLaunchKernelGGL(k_sub_count, grid, blk, 0, c->stream, int(n), gp, g->start, g->cell_of, g->sx,
                     g->sy, g->sz, sub->sub_cnt, oct_rank);
  hipLaunchKernelGGL(k_sub_prefix, grid, blk, 0, c->stream, int(n), g->start, g->cell_of,
                     sub->sub_cnt, sub->sub_beg, sub->rec);
  hipLaunchKernelGGL(k_sub_scatter, grid, blk, 0, c->stream, int(n), g->start, g->cell_of, oct_rank,
                     sub->sub_beg, g->order, g->sx, g->sy, g->sz, order2, sx2, sy2, sz2, sub->sub_of);
  PQ_HIP(hipGetLastError());
  g->order = order2;
  g->sx = sx2;
  g->sy = sy2;
  g->sz = sz2;
  return 0;
}

// ---- grid + octant sub-cells in one pass (DBSCAN's binning) ------------------------------

struct AxisMap {  // raw slab index -> compressed slab index (+1 for the border); null = identity
  const int32_t *x, *y, *z;
};

struct RawCell {
  int cx, cy, cz, oct;
};

// raw interior coordinates (0-based, clamped) and the octant inside the raw cell
__device__ __forceinline__ RawCell raw_cell(const GridParams& g, int rx, int ry, int rz, double x,
                                            double y, double z) {
  // positions in half cells; the cell is half of that, the octant its parity
  int hx = int(floor((x - g.minx) * (2.0 * g.inv_cell)));
  int hy = int(floor((y - g.miny) * (2.0 * g.inv_cell)));
  int hz = int(floor((z - g.minz) * (2.0 * g.inv_cell)));
  int cx = int(floor((x - g.minx) * g.inv_cell));
  int cy = int(floor((y - g.miny) * g.inv_cell));
  int cz = int(floor((z - g.minz) * g.inv_cell));
  cx = cx < 0 ? 0 : (cx > rx - 1 ? rx - 1 : cx);
  cy = cy < 0 ? 0 : (cy > ry - 1 ? ry - 1 : cy);
  cz = cz < 0 ? 0 : (cz > rz - 1 ? rz - 1 : cz);
  // octant relative to the cell the point was binned into (as k_sub_count: h - 2 c < 1 ? 0 : 1)
  const int ox = hx - 2 * cx < 1 ? 0 : 1, oy = hy - 2 * cy < 1 ? 0 : 1, oz = hz - 2 * cz < 1 ? 0 : 1;
  return RawCell{cx, cy, cz, ox | (oy << 1) | (oz << 2)};
}

// which raw slabs hold a point (plain stores of 1: racing stores of the same value)
__global__ __launch_bounds__(256) void k_axis_occ(const double* __restrict__ xyz, int64_t n,
                                                  GridParams g, int rx, int ry, int rz,
                                                  int32_t* __restrict__ ox, int32_t* __restrict__ oy,
                                                  int32_t* __restrict__ oz) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const RawCell r = raw_cell(g, rx, ry, rz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  ox[r.cx] = 1;
  oy[r.cy] = 1;
  oz[r.cz] = 1;
}

// keep[i] = slab i is occupied, or is the first empty slab after an occupied one
__global__ __launch_bounds__(256) void k_axis_keep(int m, const int32_t* __restrict__ occ,
                                                   int32_t* __restrict__ keep /*[m + 1]*/) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i > m) return;
  keep[i] = i < m ? ((occ[i] || (i > 0 && occ[i - 1])) ? 1 : 0) : 0;
}

// after the scan: compressed index of every raw slab, shifted by the border cell
__global__ __launch_bounds__(256) void k_axis_shift(int m, int32_t* __restrict__ map) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) map[i] += 1;
}

template <bool MAPPED>
__global__ __launch_bounds__(256) void k_cell_count_oct(const double* __restrict__ xyz, int64_t n,
                                                        GridParams g, int rx, int ry, int rz, AxisMap am,
                                                        int32_t* __restrict__ counts,
                                                        int32_t* __restrict__ cell_tmp,
                                                        int32_t* __restrict__ rank_tmp) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const RawCell r = raw_cell(g, rx, ry, rz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
  const int cx = MAPPED ? am.x[r.cx] : r.cx + 1, cy = MAPPED ? am.y[r.cy] : r.cy + 1,
            cz = MAPPED ? am.z[r.cz] : r.cz + 1;
  const int c = (cz * g.ny + cy) * g.nx + cx;
  cell_tmp[i] = c;
  rank_tmp[i] = (atomicAdd(&counts[c], 1) << 3) | r.oct;  // arrival rank in the cell, octant
}

// One 32-byte record per point to its cell-sorted position: coordinates, original index,
// cell * 8 + octant. (Scattering only (index, key) and gathering the coordinates in the ordering
// pass cost 13 + 51 us per million points: a million random 24-byte reads. With the coordinates
// in the record the ordering pass reads and writes whole lines.)
struct alignas(32) PointRec {
  double x, y, z;
  int32_t idx, key;
};

__global__ __launch_bounds__(256) void k_scatter_keys(const double* __restrict__ xyz, int64_t n,
                                                      const int32_t* __restrict__ start,
                                                      const int32_t* __restrict__ cell_tmp,
                                                      const int32_t* __restrict__ rank_tmp,
                                                      PointRec* __restrict__ keyed) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int c = cell_tmp[i], ro = rank_tmp[i];
  PointRec r;
  r.x = xyz[3 * i];
  r.y = xyz[3 * i + 1];
  r.z = xyz[3 * i + 2];
  r.idx = int(i);
  r.key = (c << 3) | (ro & 7);
  keyed[start[c] + (ro >> 3)] = r;
}

static constexpr int kBigCell = 128;  // cells with more points are ordered by a block, not per point

// One thread per cell-sorted position p: the final position of its point inside the cell's run,
// ordered by octant (stable in arrival rank), from the octant bits of the run itself; the run's
// first thread also writes the eight sub-cell records. Cells with more than kBigCell points are
// listed for k_order_big (a per-point walk would be quadratic in the cell's size).
__global__ __launch_bounds__(256) void k_order_cells(int n, const int32_t* __restrict__ start,
                                                     const PointRec* __restrict__ keyed,
                                                     int32_t* __restrict__ order,
                                                     int32_t* __restrict__ cell_of,
                                                     int32_t* __restrict__ sub_of,
                                                     double* __restrict__ sx, double* __restrict__ sy,
                                                     double* __restrict__ sz,
                                                     int32_t* __restrict__ sub_cnt,
                                                     int32_t* __restrict__ sub_beg,
                                                     int4* __restrict__ rec,
                                                     int32_t* __restrict__ big_list,
                                                     int32_t* __restrict__ big_cnt) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const PointRec me = keyed[p];
  const int c = me.key >> 3, o = me.key & 7;
  const int b = start[c], e = start[c + 1];
  if (e - b > kBigCell) {
    if (p == b) big_list[atomicAdd(big_cnt, 1)] = c;
    return;
  }
  int less = 0, same_before = 0;
  int cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool first = p == b;
  for (int q = b; q < e; ++q) {
    const int oq = keyed[q].key & 7;
    less += oq < o;
    same_before += (oq == o) & (q < p);
    if (first) {
#pragma unroll
      for (int k = 0; k < 8; ++k) cnt[k] += oq == k;
    }
  }
  const int f = b + less + same_before;
  order[f] = me.idx;
  cell_of[f] = c;
  sub_of[f] = b * 8 + o;
  sx[f] = me.x;
  sy[f] = me.y;
  sz[f] = me.z;
  if (first) {
    int run = b;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      sub_beg[size_t(b) * 8 + k] = run;
      sub_cnt[size_t(b) * 8 + k] = cnt[k];
      rec[size_t(b) * 8 + k] = make_int4(run, cnt[k], -1, 0);
      run += cnt[k];
    }
  }
}

// A block per big cell: stable counting sort of the run by octant (two sweeps over the run).
__global__ __launch_bounds__(256) void k_order_big(const int32_t* __restrict__ big_list,
                                                   const int32_t* __restrict__ big_cnt,
                                                   const int32_t* __restrict__ start,
                                                   const PointRec* __restrict__ keyed,
                                                   int32_t* __restrict__ order,
                                                   int32_t* __restrict__ cell_of,
                                                   int32_t* __restrict__ sub_of,
                                                   double* __restrict__ sx, double* __restrict__ sy,
                                                   double* __restrict__ sz, float4* __restrict__ p4,
                                                   int32_t* __restrict__ sub_cnt,
                                                   int32_t* __restrict__ sub_beg,
                                                   int4* __restrict__ rec) {
  __shared__ int tot[8], base[8], wcnt[4][8];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nbig = *big_cnt;  // usually 0: the launch is then a few idle blocks
  for (int bi = blockIdx.x; bi < nbig; bi += gridDim.x) {  // block-uniform
  const int c = big_list[bi];
  const int b = start[c], e = start[c + 1];
  __syncthreads();
  if (threadIdx.x < 8) tot[threadIdx.x] = 0;
  __syncthreads();
  // sweep 1: octant totals
  int loc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int q = b + threadIdx.x; q < e; q += 256) {
    const int oq = keyed[q].key & 7;
#pragma unroll
    for (int k = 0; k < 8; ++k) loc[k] += oq == k;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int v = loc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0 && v) atomicAdd(&tot[k], v);  // LDS, four adders
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = b;
    for (int k = 0; k < 8; ++k) {
      base[k] = run;
      sub_beg[size_t(b) * 8 + k] = run;
      sub_cnt[size_t(b) * 8 + k] = tot[k];
      rec[size_t(b) * 8 + k] = make_int4(run, tot[k], -1, 0);
      run += tot[k];
    }
  }
  __syncthreads();
  // sweep 2: chunks of 256 in run order; rank inside the chunk by ballots, chunk bases advance
  for (int q0 = b; q0 < e; q0 += 256) {
    const int q = q0 + threadIdx.x;
    const bool live = q < e;
    PointRec me;
    me.key = 0;
    if (live) me = keyed[q];
    const int o = me.key & 7;
    int before = 0;  // same octant, earlier lane of this wave
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const unsigned long long m = __ballot(live && o == k);
      if (lane == 0) wcnt[w][k] = __popcll(m);
      if (o == k) before = __popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (live) {
      int f = base[o] + before;
      for (int ww = 0; ww < w; ++ww) f += wcnt[ww][o];
      order[f] = me.idx;
      cell_of[f] = c;
      sub_of[f] = b * 8 + o;
      if (p4) {
        p4[f] = make_float4(float(me.x), float(me.y), float(me.z), 0.f);
      } else {
        sx[f] = me.x;
        sy[f] = me.y;
        sz[f] = me.z;
      }
    }
    __syncthreads();
    if (threadIdx.x < 8)
      base[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] +
                           wcnt[3][threadIdx.x];
    __syncthreads();
  }
  }
}

// ---- the same outputs without scattered global atomics and without a pass over the directory ----
// Round 2's binning above costs one returning atomic per point at 64 different lines per wave
// instruction (the chip's scattered-atomic rate: 51 us per million points), a memset and a
// three-phase scan of the dense directory (16 M cells: 65 us), a 32-byte record scattered and read
// back (20 + 48 us). This version splits the counting sort in two levels so that every atomic
// is either in LDS or lands, 64 lanes wide, on 64 CONSECUTIVE counters (4 memory requests instead of
// 64), and the directory is written exactly once, by the blocks that sort:
//   A1  k_bk_hist     cell and octant of every point; per block an LDS histogram over the BUCKETS
//                     (a bucket = 4096 or 8192 consecutive cell ids), flushed with one atomic per
//                     non-empty (block, bucket), lanes on consecutive buckets
//   A2  k_bk_scatter  every block scans the <= 16384 bucket totals itself (no one-block kernel between the
//                     passes); the same LDS histogram again, this time returning the rank inside the block;
//                     a block reserves its share of every bucket with one returning atomic
//                     (same shape) and writes the 32-byte records bucket by bucket
//   B   k_bk_sort     one block per bucket, everything in LDS: a count per cell (arrival rank) and
//                     eight 8-bit counts per cell packed in a u64 (rank inside the octant), a block
//                     scan of the bucket's cell counts -> the bucket's directory entries,
//                     written once and coalesced; every point then goes straight to its final,
//                     octant-ordered place. Cells with more than 255 points (a packed count could
//                     overflow) are listed for k_order_big exactly as before.
// Same arrays as build_grid_octants leaves (the order inside a sub-cell is the arrival order of
// atomics in both versions, and nothing downstream depends on it).
static constexpr int kBkMax = 16384;            // buckets (LDS histogram of the A passes: <= 64 KB)
static constexpr int kBkPts = 2048;             // points per block of the A passes
static constexpr int kBkBig = 255;              // cells above this go through k_order_big
static constexpr int kBkPer = 4;                // records a thread of k_bk_sort keeps in registers

// MODE 0: cell and octant on the raw grid; 1: the same through the axis-compression maps;
// 2: cell only, clamped into the grid's box (build_grid: what kNN and the radius queries bin with)
template <int MODE>
__global__ __launch_bounds__(256) void k_bk_hist(const double* __restrict__ xyz, int64_t n, GridParams g,
                                                 int rx, int ry, int rz, AxisMap am, int nbk, int bits,
                                                 int32_t* __restrict__ key_tmp,
                                                 int32_t* __restrict__ tot) {
  extern __shared__ __attribute__((aligned(16))) int32_t h[];
  for (int b = threadIdx.x; b < nbk; b += 256) h[b] = 0;
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * kBkPts;
#pragma unroll
  for (int k = 0; k < kBkPts / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    if (i < n) {
      int c, oct = 0;
      if (MODE == 2) {
        c = cell_index(g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      } else {
        const RawCell r = raw_cell(g, rx, ry, rz, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        const int cx = MODE == 1 ? am.x[r.cx] : r.cx + 1, cy = MODE == 1 ? am.y[r.cy] : r.cy + 1,
                  cz = MODE == 1 ? am.z[r.cz] : r.cz + 1;
        c = (cz * g.ny + cy) * g.nx + cx;
        oct = r.oct;
      }
      key_tmp[i] = (c << 3) | oct;
      atomicAdd(&h[c >> bits], 1);  // LDS
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbk; b += 256) {
    const int v = h[b];
    if (v) atomicAdd(&tot[b], v);  // lanes on consecutive counters: 4 requests per wave instruction
  }
}

// one block: bstart[b] = points in buckets before b (bstart[nbk] = n). Used for directories of many
// buckets; up to kBkFusedScan buckets every block of k_bk_scatter scans the totals itself instead.
static constexpr int kBkFusedScan = 4096;
__global__ __launch_bounds__(1024) void k_bk_scan(int nbk, const int32_t* __restrict__ tot,
                                                  int32_t* __restrict__ bstart) {
  __shared__ int32_t wsum[16];
  const int per = (nbk + 1023) / 1024;  // <= 16
  const int b0 = threadIdx.x * per;
  int32_t v[16], s = 0;
  for (int k = 0; k < per; ++k) {
    v[k] = b0 + k < nbk ? tot[b0 + k] : 0;
    s += v[k];
  }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int32_t incl = s;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int32_t run = incl - s;
  for (int q = 0; q < w; ++q) run += wsum[q];
  for (int k = 0; k < per; ++k) {
    if (b0 + k < nbk) bstart[b0 + k] = run;
    run += v[k];
  }
  if (threadIdx.x == 1023) bstart[nbk] = run;
}

__global__ __launch_bounds__(256) void k_bk_scatter(const double* __restrict__ xyz, int64_t n, int nbk,
                                                    int bits, const int32_t* __restrict__ key_tmp,
                                                    const int32_t* __restrict__ tot,
                                                    int32_t* __restrict__ cursor /*zeroed*/,
                                                    int32_t* __restrict__ bstart /*[nbk + 1]: written by block 0
                                                    (fused) or read (scanned by k_bk_scan)*/,
                                                    int fused, PointRec* __restrict__ bucketed) {
  extern __shared__ __attribute__((aligned(16))) int32_t h[];  // [nbk] ranks / shares (+ [nbk] bucket starts: fused)
  int32_t* pre = h + nbk;
  __shared__ int32_t wsum[4];
  for (int b = threadIdx.x; b < nbk; b += 256) h[b] = 0;
  // fused: every block scans the bucket totals itself (<= 16 KB out of L2: cheaper than a one-block
  // kernel of its own between the two passes): thread t owns `per` consecutive buckets
  if (fused) {
    const int per = (nbk + 255) / 256, b0 = threadIdx.x * per;
    int32_t s = 0;
    for (int k = 0; k < per; ++k) s += b0 + k < nbk ? tot[b0 + k] : 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int32_t run = incl - s;
    for (int q = 0; q < w; ++q) run += wsum[q];
    for (int k = 0; k < per; ++k) {
      if (b0 + k < nbk) {
        pre[b0 + k] = run;
        if (blockIdx.x == 0) bstart[b0 + k] = run;
        run += tot[b0 + k];
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 255) bstart[nbk] = run;
  }
  __syncthreads();
  const int64_t base = int64_t(blockIdx.x) * kBkPts;
  int key[kBkPts / 256], lr[kBkPts / 256];
#pragma unroll
  for (int k = 0; k < kBkPts / 256; ++k) {
    const int64_t i = base + k * 256 + threadIdx.x;
    key[k] = i < n ? key_tmp[i] : -1;
    lr[k] = key[k] >= 0 ? atomicAdd(&h[key[k] >> (3 + bits)], 1) : 0;  // rank inside this block's share
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nbk; b += 256) {
    const int v = h[b];
    if (v) h[b] = (fused ? pre[b] : bstart[b]) + atomicAdd(&cursor[b], v);  // this block's share of the bucket
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kBkPts / 256; ++k) {
    if (key[k] < 0) continue;
    const int64_t i = base + k * 256 + threadIdx.x;
    PointRec r;
    r.x = xyz[3 * i];
    r.y = xyz[3 * i + 1];
    r.z = xyz[3 * i + 2];
    r.idx = int(i);
    r.key = key[k];
    bucketed[h[key[k] >> (3 + bits)] + lr[k]] = r;
  }
}

// sum of the bytes of x below byte o (eight 8-bit counts packed in a u64)
__device__ __forceinline__ int bytes_below(unsigned long long x, int o) {
  x &= (1ull << (8 * o)) - 1ull;
  x = (x & 0x00FF00FF00FF00FFull) + ((x >> 8) & 0x00FF00FF00FF00FFull);
  x = (x & 0x0000FFFF0000FFFFull) + ((x >> 16) & 0x0000FFFF0000FFFFull);
  return int((x + (x >> 32)) & 0xFFFFull);
}

template <int BITS>
struct BkLds {
  unsigned long long oct[1 << BITS];  // eight 8-bit counts per cell
  int32_t cnt[1 << BITS];             // points per cell, then (in place) the cell's offset in the bucket
  uint32_t big[(1 << BITS) / 32];     // cells with more than kBkBig points
  int32_t wsum[16];
};

// One block per bucket of 2^BITS cells, a thread per eight cells (BITS = 12: 512 threads and 49 KB of
// LDS, three blocks per CU; BITS = 13 for directories beyond 2^26 cells). A bucket of at most
// kBkPer points per thread — every bucket of a scan — keeps its records and ranks in registers
// between the two sweeps: the block's chain of dependent memory round trips is then bstart ->
// records -> stores. Larger buckets go through rank_tmp / oct_rank in global memory.
template <int BITS>
__global__ __launch_bounds__((1 << BITS) / 8) void k_bk_sort(
    int64_t ncell1 /*directory entries: ncell + 1*/, const int32_t* __restrict__ bstart,
    const PointRec* __restrict__ bucketed, int32_t* __restrict__ rank_tmp, uint8_t* __restrict__ oct_rank,
    int32_t* __restrict__ start, int32_t* __restrict__ order, int32_t* __restrict__ cell_of,
    int32_t* __restrict__ sub_of, double* __restrict__ sx, double* __restrict__ sy, double* __restrict__ sz,
    float4* __restrict__ p4 /*non-null: fp32 records instead of sx / sy / sz*/, int4* __restrict__ rec,
    PointRec* __restrict__ keyed, int32_t* __restrict__ big_list, int32_t* __restrict__ big_cnt) {
  constexpr int CELLS = 1 << BITS, T = CELLS / 8;
  __shared__ BkLds<BITS> L;
  const int bk = blockIdx.x, t = threadIdx.x;
  const int s = bstart[bk], e = bstart[bk + 1];
  const int64_t c0 = int64_t(bk) << BITS;  // first cell (directory entry) of the bucket
  if (s == e) {  // nothing in the bucket: its directory entries all say "the next point is s"
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t cidx = c0 + int64_t(k * T + t) * 4;
      if (cidx + 3 < ncell1) {
        *reinterpret_cast<int4*>(start + cidx) = make_int4(s, s, s, s);
      } else {
        for (int u = 0; u < 4; ++u)
          if (cidx + u < ncell1) start[cidx + u] = s;
      }
    }
    return;
  }
  const bool inreg = e - s <= T * kBkPer;  // block-uniform
  PointRec me[kBkPer];
  int rr[kBkPer], r8[kBkPer];
  if (inreg) {  // issue the loads before the LDS is cleared
#pragma unroll
    for (int k = 0; k < kBkPer; ++k) {
      const int j = s + k * T + t;
      me[k].key = -1;
      if (j < e) me[k] = bucketed[j];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    L.oct[k * T + t] = 0ull;
    L.cnt[k * T + t] = 0;
  }
  if (t < CELLS / 32) L.big[t] = 0u;
  __syncthreads();
  // sweep 1: arrival rank in the cell and in the octant
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kBkPer; ++k) {
      if (me[k].key < 0) continue;
      const int cl = (me[k].key >> 3) & (CELLS - 1), o = me[k].key & 7;
      rr[k] = atomicAdd(&L.cnt[cl], 1);
      r8[k] = int((atomicAdd(&L.oct[cl], 1ull << (8 * o)) >> (8 * o)) & 255ull);
    }
  } else {
    for (int j = s + t; j < e; j += T) {
      const int key = bucketed[j].key;
      const int cl = (key >> 3) & (CELLS - 1), o = key & 7;
      rank_tmp[j] = atomicAdd(&L.cnt[cl], 1);
      const unsigned long long old = atomicAdd(&L.oct[cl], 1ull << (8 * o));
      oct_rank[j] = uint8_t(old >> (8 * o));  // meaningful only in cells of at most kBkBig points
    }
  }
  __syncthreads();
  // the bucket's cells: thread t owns eight consecutive ones
  int32_t v[8], tot = 0;
  {
    const int4 a = *reinterpret_cast<const int4*>(&L.cnt[8 * t]);
    const int4 b = *reinterpret_cast<const int4*>(&L.cnt[8 * t + 4]);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) tot += v[k];
  }
  const int lane = t & 63, w = t >> 6;
  int32_t incl = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t u = __shfl_up(incl, off, 64);
    if (lane >= off) incl += u;
  }
  if (lane == 63) L.wsum[w] = incl;
  __syncthreads();
  int32_t run = incl - tot;
  for (int q = 0; q < w; ++q) run += L.wsum[q];
  int32_t pre[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pre[k] = run;
    run += v[k];
  }
  {
    const int64_t cidx = c0 + 8 * t;
    if (cidx + 7 < ncell1) {
      *reinterpret_cast<int4*>(start + cidx) = make_int4(s + pre[0], s + pre[1], s + pre[2], s + pre[3]);
      *reinterpret_cast<int4*>(start + cidx + 4) = make_int4(s + pre[4], s + pre[5], s + pre[6], s + pre[7]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (cidx + k < ncell1) start[cidx + k] = s + pre[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int cl = 8 * t + k;
    L.cnt[cl] = pre[k];
    if (v[k] == 0) continue;
    const int b = s + pre[k];  // the cell's first sorted position: identifies its sub-cells
    if (v[k] > kBkBig) {
      atomicOr(&L.big[cl >> 5], 1u << (cl & 31));
      big_list[atomicAdd(big_cnt, 1)] = int(c0) + cl;  // rare; k_order_big writes the cell's records
      continue;
    }
    const unsigned long long c64 = L.oct[cl];
    int r2 = b;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      const int cnt = int((c64 >> (8 * o)) & 255ull);
      rec[size_t(b) * 8 + o] = make_int4(r2, cnt, -1, 0);
      r2 += cnt;
    }
  }
  __syncthreads();
  // sweep 2: every point to its final place (big cells: to the cell's run of `keyed`, in arrival order)
  auto place = [&](const PointRec& p, int arrival, int in_oct) {
    const int cl = (p.key >> 3) & (CELLS - 1), o = p.key & 7;
    const int b = s + L.cnt[cl];
    if ((L.big[cl >> 5] >> (cl & 31)) & 1u) {
      keyed[b + arrival] = p;
      return;
    }
    const int f = b + bytes_below(L.oct[cl], o) + in_oct;
    order[f] = p.idx;
    cell_of[f] = p.key >> 3;
    sub_of[f] = b * 8 + o;
    if (p4) {
      p4[f] = make_float4(float(p.x), float(p.y), float(p.z), 0.f);
    } else {
      sx[f] = p.x;
      sy[f] = p.y;
      sz[f] = p.z;
    }
  };
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kBkPer; ++k)
      if (me[k].key >= 0) place(me[k], rr[k], r8[k]);
  } else {
    for (int j = s + t; j < e; j += T) place(bucketed[j], rank_tmp[j], int(oct_rank[j]));
  }
}

// The same second level without octants (build_grid): a count per cell is all there is, so a bucket
// of 2^BITS cells needs 4 bytes of LDS per cell, the arrival rank IS the place inside the cell, and
// no cell is too big. occ[bucket] = occupied cells of the bucket (count_occupied adds them up).
template <int BITS>
__global__ __launch_bounds__((1 << BITS) / 8) void k_bk_sort_plain(
    int64_t ncell1, const int32_t* __restrict__ bstart, const PointRec* __restrict__ bucketed,
    int32_t* __restrict__ rank_tmp, int32_t* __restrict__ start, int32_t* __restrict__ order,
    int32_t* __restrict__ cell_of, double* __restrict__ sx, double* __restrict__ sy, double* __restrict__ sz,
    float4* __restrict__ p4 /*non-null: fp32 records instead of sx / sy / sz*/, int32_t* __restrict__ occ) {
  constexpr int CELLS = 1 << BITS, T = CELLS / 8;
  __shared__ int32_t cnt[CELLS];
  __shared__ int32_t wsum[16], wocc[16];
  const int bk = blockIdx.x, t = threadIdx.x;
  const int s = bstart[bk], e = bstart[bk + 1];
  const int64_t c0 = int64_t(bk) << BITS;
  if (s == e) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int64_t cidx = c0 + int64_t(k * T + t) * 4;
      if (cidx + 3 < ncell1) {
        *reinterpret_cast<int4*>(start + cidx) = make_int4(s, s, s, s);
      } else {
        for (int u = 0; u < 4; ++u)
          if (cidx + u < ncell1) start[cidx + u] = s;
      }
    }
    if (t == 0) occ[bk] = 0;
    return;
  }
  const bool inreg = e - s <= T * kBkPer;
  PointRec me[kBkPer];
  int rr[kBkPer];
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kBkPer; ++k) {
      const int j = s + k * T + t;
      me[k].key = -1;
      if (j < e) me[k] = bucketed[j];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) cnt[k * T + t] = 0;
  __syncthreads();
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kBkPer; ++k)
      if (me[k].key >= 0) rr[k] = atomicAdd(&cnt[(me[k].key >> 3) & (CELLS - 1)], 1);
  } else {
    for (int j = s + t; j < e; j += T) rank_tmp[j] = atomicAdd(&cnt[(bucketed[j].key >> 3) & (CELLS - 1)], 1);
  }
  __syncthreads();
  int32_t v[8], tot = 0, nocc = 0;
  {
    const int4 a = *reinterpret_cast<const int4*>(&cnt[8 * t]);
    const int4 b = *reinterpret_cast<const int4*>(&cnt[8 * t + 4]);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      tot += v[k];
      nocc += v[k] > 0;
    }
  }
  const int lane = t & 63, w = t >> 6;
  int32_t incl = tot;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int32_t u = __shfl_up(incl, off, 64);
    if (lane >= off) incl += u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) nocc += __shfl_down(nocc, off, 64);
  if (lane == 63) wsum[w] = incl;
  if (lane == 0) wocc[w] = nocc;
  __syncthreads();
  int32_t run = incl - tot;
  for (int q = 0; q < w; ++q) run += wsum[q];
  if (t == 0) {
    int o = 0;
    for (int q = 0; q < T / 64; ++q) o += wocc[q];
    occ[bk] = o;
  }
  int32_t pre[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    pre[k] = run;
    run += v[k];
  }
  {
    const int64_t cidx = c0 + 8 * t;
    if (cidx + 7 < ncell1) {
      *reinterpret_cast<int4*>(start + cidx) = make_int4(s + pre[0], s + pre[1], s + pre[2], s + pre[3]);
      *reinterpret_cast<int4*>(start + cidx + 4) = make_int4(s + pre[4], s + pre[5], s + pre[6], s + pre[7]);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (cidx + k < ncell1) start[cidx + k] = s + pre[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) cnt[8 * t + k] = pre[k];
  __syncthreads();
  auto place = [&](const PointRec& p, int arrival) {
    const int f = s + cnt[(p.key >> 3) & (CELLS - 1)] + arrival;
    order[f] = p.idx;
    cell_of[f] = p.key >> 3;
    if (p4) {
      p4[f] = make_float4(float(p.x), float(p.y), float(p.z), 0.f);
    } else {
      sx[f] = p.x;
      sy[f] = p.y;
      sz[f] = p.z;
    }
  };
  if (inreg) {
#pragma unroll
    for (int k = 0; k < kBkPer; ++k)
      if (me[k].key >= 0) place(me[k], rr[k]);
  } else {
    for (int j = s + t; j < e; j += T) place(bucketed[j], rank_tmp[j]);
  }
}

// build_grid through the two-level sort; returns false (nothing launched) when the directory has
// more buckets than the A passes' LDS histogram holds or PYQSM_GRID_BIN=atomic asks for the
// one-atomic-per-point path.
static bool bucketed_fits(const DevGrid& g) {
  const int bits = g.ncell + 1 <= (int64_t(kBkMax) << 12) ? 12 : 13;
  const int64_t nbk = (g.ncell + (int64_t(1) << bits)) >> bits;
  const char* env = getenv("PYQSM_GRID_BIN");
  return nbk <= kBkMax && !(env && !strcmp(env, "atomic"));
}

static int build_grid_bucketed(Ctx* c, const double* xyz, int64_t n, DevGrid* g) {
  const int bits = g->ncell + 1 <= (int64_t(kBkMax) << 12) ? 12 : 13;
  const int64_t nbk = (g->ncell + (int64_t(1) << bits)) >> bits;
  int32_t *tot, *bstart, *cursor, *key_tmp, *rank_tmp;
  PointRec* bucketed;
  PQ_TRY(c->arena.get(size_t(nbk) * 2, &tot));  // totals, then the reservation cursors: one memset
  cursor = tot + nbk;
  PQ_TRY(c->arena.get(size_t(nbk) + 1, &bstart));
  PQ_TRY(c->arena.get(size_t(n), &key_tmp));
  PQ_TRY(c->arena.get(size_t(n), &rank_tmp));
  PQ_TRY(c->arena.get(size_t(n), &bucketed));
  g->occ_blocks = int(nbk);
  PQ_TRY(c->arena.get(size_t(nbk), &g->occ_part));
  PQ_HIP(hipMemsetAsync(tot, 0, size_t(nbk) * 8, c->stream));
  GridParams gp{g->minx, g->miny, g->minz, g->inv_cell, g->nx, g->ny, g->nz};
  const dim3 ga(ceil_div(n, kBkPts)), blk(256);
  const size_t lds = size_t(nbk) * 4;
  const int fused = nbk <= kBkFusedScan;
  hipLaunchKernelGGL(k_bk_hist<2>, ga, blk, lds, c->stream, xyz, n, gp, 0, 0, 0, AxisMap{nullptr, nullptr, nullptr},
                     int(nbk), bits, key_tmp, tot);
  if (!fused) hipLaunchKernelGGL(k_bk_scan, dim3(1), dim3(1024), 0, c->stream, int(nbk), tot, bstart);
  hipLaunchKernelGGL(k_bk_scatter, ga, blk, fused ? 2 * lds : lds, c->stream, xyz, n, int(nbk), bits, key_tmp, tot,
                     cursor, bstart, fused, bucketed);
  if (bits == 12)
    hipLaunchKernelGGL(k_bk_sort_plain<12>, dim3(unsigned(nbk)), dim3(512), 0, c->stream, g->ncell + 1, bstart,
                       bucketed, rank_tmp, g->start, g->order, g->cell_of, g->sx, g->sy, g->sz, g->p4, g->occ_part);
  else
    hipLaunchKernelGGL(k_bk_sort_plain<13>, dim3(unsigned(nbk)), dim3(1024), 0, c->stream, g->ncell + 1, bstart,
                       bucketed, rank_tmp, g->start, g->order, g->cell_of, g->sx, g->sy, g->sz, g->p4, g->occ_part);
  PQ_HIP(hipGetLastError());
  return 0;
}

int build_grid_octants(Ctx* c, const double* xyz, int64_t n, double min_cell, int64_t max_cells,
                       DevGrid* g, SubCells* sub) {
  if (n <= 0) return fail(PYQSM_EINVAL, "build_grid_octants: empty cloud");
  if (n > (int64_t(1) << 27)) return fail(PYQSM_ERANGE, "octant sub-cells: more than 2^27 points");
  if (max_cells > (int64_t(1) << 28)) max_cells = int64_t(1) << 28;  // cell * 8 + octant in 31 bits
  if (!(min_cell > 0) || !std::isfinite(min_cell))
    return fail(PYQSM_EINVAL, "cell edge must be positive and finite");
  double mn[3], mx[3];
  // the bucket totals and the big-cell counter of the two-level sort below: cleared by the
  // bounding box's fold kernel on its way (two memset launches less)
  // layout: [kBkMax totals | kBkMax reservation cursors | big-cell counter | four spare zeros for the caller]
  int32_t* tot_big = nullptr;
  PQ_TRY(c->arena.get(size_t(2 * kBkMax) + 5 + kZeroedExtra, &tot_big));
  bool all_f32 = false;
  PQ_TRY(cloud_bbox(c, xyz, n, mn, mx, &all_f32, tot_big, 2 * kBkMax + 5 + kZeroedExtra));
  sub->zeroed4 = tot_big + 2 * kBkMax + 1;
  {
    const char* f32_env = getenv("PYQSM_COORD_F32");  // "0": keep fp64 storage (A/B comparisons)
    if (f32_env && !strcmp(f32_env, "0")) all_f32 = false;
  }
  const dim3 grid(ceil_div(n, 256)), blk(256);
  double cell = min_cell;
  int raw[3], dims[3];
  AxisMap am{nullptr, nullptr, nullptr};
  bool mapped = false;
  for (;;) {
    double tot = 1.0;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      const double d = std::floor((mx[a] - mn[a]) / cell) + 1.0;  // interior slabs
      if (!(d < 2.0e9)) ok = false;
      raw[a] = ok ? int(d) : 0;
      dims[a] = raw[a] + 2;
      tot *= d + 2.0;
    }
    if (ok && tot <= double(max_cells)) break;
    // too many cells of this edge: drop the empty slabs (per axis) before giving up on the edge
    const int64_t raw_sum = ok ? int64_t(raw[0]) + raw[1] + raw[2] : int64_t(1) << 40;
    if (raw_sum <= (int64_t(1) << 27)) {
      GridParams gp{mn[0], mn[1], mn[2], 1.0 / cell, 0, 0, 0};
      int32_t* occ[3];
      int32_t* map[3];
      for (int a = 0; a < 3; ++a) {
        PQ_TRY(c->arena.get(size_t(raw[a]) + 1, &occ[a]));
        PQ_TRY(c->arena.get(size_t(raw[a]) + 1, &map[a]));
        PQ_HIP(hipMemsetAsync(occ[a], 0, (size_t(raw[a]) + 1) * 4, c->stream));
      }
      hipLaunchKernelGGL(k_axis_occ, grid, blk, 0, c->stream, xyz, n, gp, raw[0], raw[1], raw[2], occ[0],
                         occ[1], occ[2]);
      double ctot = 1.0;
      for (int a = 0; a < 3; ++a) {
        hipLaunchKernelGGL(k_axis_keep, dim3(ceil_div(raw[a] + 1, 256)), blk, 0, c->stream, raw[a], occ[a],
                           map[a]);
        PQ_HIP(hipGetLastError());
        PQ_TRY(exclusive_scan_i32(c, map[a], int64_t(raw[a]) + 1));
        int32_t kept = 0;
        PQ_HIP(hipMemcpyAsync(&kept, map[a] + raw[a], 4, hipMemcpyDeviceToHost, c->stream));
        PQ_HIP(hipStreamSynchronize(c->stream));
        hipLaunchKernelGGL(k_axis_shift, dim3(ceil_div(raw[a], 256)), blk, 0, c->stream, raw[a], map[a]);
        dims[a] = kept + 2;
        ctot *= double(dims[a]);
      }
      PQ_HIP(hipGetLastError());
      if (ctot <= double(max_cells)) {
        am = AxisMap{map[0], map[1], map[2]};
        mapped = true;
        break;
      }
    }
    cell *= 2.0;
  }
  g->minx = mn[0];
  g->miny = mn[1];
  g->minz = mn[2];
  g->cell = cell;
  g->inv_cell = 1.0 / cell;
  g->nx = dims[0];
  g->ny = dims[1];
  g->nz = dims[2];
  g->ncell = int64_t(dims[0]) * dims[1] * dims[2];
  int32_t *cell_tmp, *rank_tmp, *big_list, *big_cnt;
  PointRec* keyed;
  // Two-level counting sort (k_bk_*): no scattered global atomics, the directory written once.
  // PYQSM_DBSCAN_BIN=atomic keeps round 2's one-atomic-per-point path (A/B comparisons); grids of
  // more than kBkMax buckets (2^27 cells) keep it too.
  const int bits = g->ncell + 1 <= (int64_t(kBkMax) << 12) ? 12 : 13;
  const int64_t nbk = (g->ncell + (int64_t(1) << bits)) >> bits;  // ceil((ncell + 1) / bucket)
  const char* bin_env = getenv("PYQSM_DBSCAN_BIN");
  const bool bucketed_path = nbk <= kBkMax && !(bin_env && !strcmp(bin_env, "atomic"));
  // fp32 records when the input allows it (round 2's path keeps fp64 arrays)
  g->p4 = nullptr;
  g->sx = g->sy = g->sz = nullptr;
  PQ_TRY(c->arena.get(size_t(g->ncell) + 1, &g->start));
  PQ_TRY(c->arena.get(size_t(n), &g->order));
  PQ_TRY(c->arena.get(size_t(n), &g->cell_of));
  if (all_f32 && bucketed_path) {
    PQ_TRY(c->arena.get(size_t(n), &g->p4));
  } else {
    PQ_TRY(c->arena.get(size_t(n), &g->sx));
    PQ_TRY(c->arena.get(size_t(n), &g->sy));
    PQ_TRY(c->arena.get(size_t(n), &g->sz));
  }
  PQ_TRY(c->arena.get(size_t(n), &cell_tmp));
  PQ_TRY(c->arena.get(size_t(n), &rank_tmp));
  PQ_TRY(c->arena.get(size_t(n), &keyed));
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->sub_cnt));
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->sub_beg));
  PQ_TRY(c->arena.get(size_t(n), &sub->sub_of));
  PQ_TRY(c->arena.get(size_t(n) * 8, &sub->rec));
  PQ_TRY(c->arena.get(size_t(n) / kBigCell + 2, &big_list));
  big_cnt = tot_big + 2 * kBkMax;
  GridParams gp{g->minx, g->miny, g->minz, g->inv_cell, g->nx, g->ny, g->nz};
  if (bucketed_path) {
    int32_t *tot = tot_big, *bstart, *cursor = tot_big + kBkMax;
    uint8_t* oct_rank;
    PointRec* bucketed;
    PQ_TRY(c->arena.get(size_t(nbk) + 1, &bstart));
    PQ_TRY(c->arena.get(size_t(n), &oct_rank));
    PQ_TRY(c->arena.get(size_t(n), &bucketed));
    const dim3 ga(ceil_div(n, kBkPts));
    const size_t lds = size_t(nbk) * 4;
    if (mapped)
      hipLaunchKernelGGL(k_bk_hist<1>, ga, blk, lds, c->stream, xyz, n, gp, raw[0], raw[1], raw[2], am,
                         int(nbk), bits, cell_tmp, tot);
    else
      hipLaunchKernelGGL(k_bk_hist<0>, ga, blk, lds, c->stream, xyz, n, gp, raw[0], raw[1], raw[2], am,
                         int(nbk), bits, cell_tmp, tot);
    const int fused = nbk <= kBkFusedScan;
    if (!fused) hipLaunchKernelGGL(k_bk_scan, dim3(1), dim3(1024), 0, c->stream, int(nbk), tot, bstart);
    hipLaunchKernelGGL(k_bk_scatter, ga, blk, fused ? 2 * lds : lds, c->stream, xyz, n, int(nbk), bits, cell_tmp, tot,
                       cursor, bstart, fused, bucketed);
    if (bits == 12)
      hipLaunchKernelGGL(k_bk_sort<12>, dim3(unsigned(nbk)), dim3(512), 0, c->stream, g->ncell + 1, bstart,
                         bucketed, rank_tmp, oct_rank, g->start, g->order, g->cell_of, sub->sub_of, g->sx, g->sy,
                         g->sz, g->p4, sub->rec, keyed, big_list, big_cnt);
    else
      hipLaunchKernelGGL(k_bk_sort<13>, dim3(unsigned(nbk)), dim3(1024), 0, c->stream, g->ncell + 1, bstart,
                         bucketed, rank_tmp, oct_rank, g->start, g->order, g->cell_of, sub->sub_of, g->sx, g->sy,
                         g->sz, g->p4, sub->rec, keyed, big_list, big_cnt);
    PQ_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_order_big, dim3(unsigned(std::min<int64_t>(n / kBigCell + 1, 2048))), blk, 0,
                       c->stream, big_list, big_cnt, g->start, keyed, g->order, g->cell_of, sub->sub_of,
                       g->sx, g->sy, g->sz, g->p4, sub->sub_cnt, sub->sub_beg, sub->rec);
    PQ_HIP(hipGetLastError());
    return 0;
  }
  PQ_HIP(hipMemsetAsync(g->start, 0, (size_t(g->ncell) + 1) * 4, c->stream));
  if (mapped)
    hipLaunchKernelGGL(k_cell_count_oct<true>, grid, blk, 0, c->stream, xyz, n, gp, raw[0], raw[1], raw[2],
                       am, g->start, cell_tmp, rank_tmp);
  else
    hipLaunchKernelGGL(k_cell_count_oct<false>, grid, blk, 0, c->stream, xyz, n, gp, raw[0], raw[1], raw[2],
                       am, g->start, cell_tmp, rank_tmp);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, g->start, g->ncell + 1));
  hipLaunchKernelGGL(k_scatter_keys, grid, blk, 0, c->stream, xyz, n, g->start, cell_tmp, rank_tmp, keyed);
  hipLaunchKernelGGL(k_order_cells, grid, blk, 0, c->stream, int(n), g->start, keyed, g->order,
                     g->cell_of, sub->sub_of, g->sx, g->sy, g->sz, sub->sub_cnt, sub->sub_beg, sub->rec,
                     big_list, big_cnt);
  PQ_HIP(hipGetLastError());
  // cells with more than kBigCell points (none on a scan at eps ~ 10 x the point spacing): a fixed
  // grid reads their number on the device, so that no host round trip is needed
  hipLaunchKernelGGL(k_order_big, dim3(unsigned(std::min<int64_t>(n / kBigCell + 1, 2048))), blk, 0,
                     c->stream, big_list, big_cnt, g->start, keyed, g->order, g->cell_of, sub->sub_of,
                     g->sx, g->sy, g->sz, g->p4, sub->sub_cnt, sub->sub_beg, sub->rec);
  PQ_HIP(hipGetLastError());
  return 0;
}

// ---- pyramid coarsening: a grid with cells `factor` times larger, without atomics ----
// Coarse cell C gathers the factor^3 fine cells of its block; their point runs are
// concatenated in (z, y, x) order, so a point's new position follows from its old one:
// coarse_start[C] + offset_of_its_fine_cell_in_C + (old position - fine_start[cell]).

struct Pyr {
  int fnx, fny, fnz, cnx, cny, cnz, factor;
};

__device__ __forceinline__ int coarse_of(const Pyr& P, int f) {
  const int fx = f % P.fnx, fy = (f / P.fnx) % P.fny, fz = f / (P.fnx * P.fny);
  // interior fine index i (1..n-2) holds coordinate i-1; borders stay borders
  auto m = [&](int i, int fn, int cn) {
    if (i <= 0) return 0;
    if (i >= fn - 1) return cn - 1;
    return (i - 1) / P.factor + 1;
  };
  const int cx = m(fx, P.fnx, P.cnx), cy = m(fy, P.fny, P.cny), cz = m(fz, P.fnz, P.cnz);
  return (cz * P.cny + cy) * P.cnx + cx;
}

__global__ __launch_bounds__(256) void k_pyr_counts(Pyr P, const int32_t* __restrict__ fstart,
                                                    int32_t* __restrict__ ccount,
                                                    int32_t* __restrict__ foff) {
  const int C = blockIdx.x * 256 + threadIdx.x;
  if (C >= P.cnx * P.cny * P.cnz) return;
  const int cx = C % P.cnx, cy = (C / P.cnx) % P.cny, cz = C / (P.cnx * P.cny);
  int run = 0;
  if (cx >= 1 && cy >= 1 && cz >= 1 && cx <= P.cnx - 2 && cy <= P.cny - 2 && cz <= P.cnz - 2) {
    const int x0 = (cx - 1) * P.factor + 1, y0 = (cy - 1) * P.factor + 1, z0 = (cz - 1) * P.factor + 1;
    if (P.factor == 4) {
      // the usual factor: the five run starts of a row of four fine cells are fetched together,
      // all sixteen rows' loads can be in flight at once (one cell at a time they were a chain of
      // 64 load-add-store steps: 0.36 ms for the 50 M-cell grid of a million points)
#pragma unroll
      for (int dz = 0; dz < 4; ++dz)
#pragma unroll
        for (int dy = 0; dy < 4; ++dy) {
          const int z = z0 + dz, y = y0 + dy;
          if (z > P.fnz - 2 || y > P.fny - 2) continue;
          const int f0 = (z * P.fny + y) * P.fnx + x0;
          const int nxr = min(4, P.fnx - 1 - x0);  // fine cells of this row inside the grid (>= 1)
          int s[5];
#pragma unroll
          for (int u = 0; u < 5; ++u) s[u] = fstart[f0 + min(u, nxr)];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (u < nxr) foff[f0 + u] = run + (s[u] - s[0]);
          run += s[4] - s[0];  // s[4] = fstart[f0 + nxr]: the clamped index above
        }
    } else {
      for (int z = z0; z < z0 + P.factor && z <= P.fnz - 2; ++z)
        for (int y = y0; y < y0 + P.factor && y <= P.fny - 2; ++y)
          for (int x = x0; x < x0 + P.factor && x <= P.fnx - 2; ++x) {
            const int f = (z * P.fny + y) * P.fnx + x;
            foff[f] = run;
            run += fstart[f + 1] - fstart[f];
          }
    }
  }
  ccount[C] = run;
}

__global__ __launch_bounds__(256) void k_pyr_scatter(int n, Pyr P, const int32_t* __restrict__ fstart,
                                                     const int32_t* __restrict__ cstart,
                                                     const int32_t* __restrict__ foff,
                                                     const int32_t* __restrict__ f_cell_of,
                                                     const int32_t* __restrict__ f_order,
                                                     const double* __restrict__ fx,
                                                     const double* __restrict__ fy,
                                                     const double* __restrict__ fz,
                                                     const float4* __restrict__ fp4,
                                                     int32_t* __restrict__ order,
                                                     int32_t* __restrict__ cell_of,
                                                     double* __restrict__ sx,
                                                     double* __restrict__ sy,
                                                     double* __restrict__ sz, float4* __restrict__ p4) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int f = f_cell_of[p];
  const int C = coarse_of(P, f);
  const int q = cstart[C] + foff[f] + (p - fstart[f]);
  order[q] = f_order[p];
  cell_of[q] = C;
  if (fp4) {
    p4[q] = fp4[p];
  } else {
    sx[q] = fx[p];
    sy[q] = fy[p];
    sz[q] = fz[p];
  }
}

int coarsen_grid(Ctx* c, const DevGrid& fine, int64_t n, int factor, DevGrid* g) {
  if (factor < 2) return fail(PYQSM_EINVAL, "coarsen_grid: factor must be >= 2");
  Pyr P;
  P.factor = factor;
  P.fnx = fine.nx;
  P.fny = fine.ny;
  P.fnz = fine.nz;
  P.cnx = (fine.nx - 2 + factor - 1) / factor + 2;
  P.cny = (fine.ny - 2 + factor - 1) / factor + 2;
  P.cnz = (fine.nz - 2 + factor - 1) / factor + 2;
  *g = fine;
  g->occ_part = nullptr;  // the fine grid's occupancy does not describe this one
  g->occ_blocks = 0;
  g->nx = P.cnx;
  g->ny = P.cny;
  g->nz = P.cnz;
  g->cell = fine.cell * factor;
  g->inv_cell = 1.0 / g->cell;
  g->ncell = int64_t(P.cnx) * P.cny * P.cnz;
  int32_t* foff = nullptr;
  PQ_TRY(c->arena.get(size_t(fine.ncell) + 1, &foff));
  PQ_TRY(c->arena.get(size_t(g->ncell) + 1, &g->start));
  PQ_TRY(c->arena.get(size_t(n), &g->order));
  PQ_TRY(c->arena.get(size_t(n), &g->cell_of));
  if (fine.p4) {
    PQ_TRY(c->arena.get(size_t(n), &g->p4));
  } else {
    PQ_TRY(c->arena.get(size_t(n), &g->sx));
    PQ_TRY(c->arena.get(size_t(n), &g->sy));
    PQ_TRY(c->arena.get(size_t(n), &g->sz));
  }
  PQ_HIP(hipMemsetAsync(g->start + g->ncell, 0, 4, c->stream));
  hipLaunchKernelGGL(k_pyr_counts, dim3(ceil_div(g->ncell, 256)), dim3(256), 0, c->stream, P,
                     fine.start, g->start, foff);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, g->start, g->ncell + 1));
  hipLaunchKernelGGL(k_pyr_scatter, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, int(n), P,
                     fine.start, g->start, foff, fine.cell_of, fine.order, fine.sx, fine.sy, fine.sz,
                     static_cast<const float4*>(fine.p4), g->order, g->cell_of, g->sx, g->sy, g->sz, g->p4);
  PQ_HIP(hipGetLastError());
  return 0;
}

int count_occupied(Ctx* c, const DevGrid& g, int64_t* occupied) {
  if (g.occ_part) {  // left by build_grid's counting pass
    std::vector<int32_t> h(size_t(g.occ_blocks));
    PQ_HIP(hipMemcpyAsync(h.data(), g.occ_part, h.size() * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    int64_t t = 0;
    for (int32_t v : h) t += v;
    *occupied = t;
    return 0;
  }
  int32_t* d = nullptr;
  PQ_TRY(c->arena.get(1, &d));
  PQ_HIP(hipMemsetAsync(d, 0, 4, c->stream));
  const int blocks = int(std::min<int64_t>(ceil_div(g.ncell, 256), int64_t(c->cu_count) * 8));
  hipLaunchKernelGGL(k_count_occupied, dim3(blocks), dim3(256), 0, c->stream, g.start, g.ncell, d);
  PQ_HIP(hipGetLastError());
  int32_t h = 0;
  PQ_HIP(hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  *occupied = h;
  return 0;
}

}  // namespace pyqsm
