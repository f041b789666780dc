// grid.hpp — uniform cell grid over a point cloud in HBM: the binning step that
// the eps-neighbourhood (DBSCAN) and kNN kernels share.
//
// Layout: cells are row-major with x fastest and carry a one-cell empty border,
// so the 3x3x3 stencil of any point is nine contiguous runs of the sorted point
// array: [start[row + cx - 1], start[row + cx + 2]) for each of the 9 (y,z) rows.
// Points are counting-sorted by cell into SoA coordinate arrays (coalesced,
// 8-byte loads per lane).
#pragma once
#include "common.hpp"

namespace pyqsm {

struct DevGrid {
  double minx, miny, minz;
  double inv_cell;
  double cell;
  int nx, ny, nz;        // including the border cells
  int64_t ncell;         // nx*ny*nz
  int32_t* start;        // [ncell + 1] first sorted position of each cell
  int32_t* order;        // [n] sorted position -> original index
  int32_t* cell_of;      // [n] cell id of each sorted position
  double *sx, *sy, *sz;  // [n] sorted coordinates (fp64 storage)
  // [n] sorted coordinates as one 16-byte (x, y, z, 0) fp32 record per point, INSTEAD of sx / sy /
  // sz (those are then null): only when every input coordinate is exactly representable in fp32
  // (cloud_bbox's flag), so that double(p4[q].x) IS the input value and every fp64 predicate
  // evaluated on it is bit-identical. Half the bytes, and a gather touches one line, not three.
  float4* p4 = nullptr;
  // build_grid only: per block of its counting pass, the points that were the first of their
  // cell — their sum is the number of occupied cells (count_occupied), without a pass over
  // the ncell-long start array
  int32_t* occ_part = nullptr;
  int occ_blocks = 0;
};

// Builds the grid for n points (f64 [n,3], device) with cells of at least
// `min_cell` edge (the edge is doubled until the dense grid has at most
// `max_cells` cells). All arrays come from the context arena. Synchronises the
// stream once (bounding box read-back). Fails with PYQSM_EINVAL on non-finite
// coordinates.
// `bbox` (min xyz, max xyz), when given, skips the bounding-box pass. `f32_records`: the caller
// knows (cloud_bbox) that every coordinate is fp32-representable and reads the grid through
// on_coords(): the sorted points are then kept as g->p4 instead of g->sx / sy / sz.
int build_grid(Ctx* c, const double* xyz, int64_t n, double min_cell, int64_t max_cells,
               DevGrid* g, const double* bbox = nullptr, bool f32_records = false);

// Bounding box of the cloud (one reduction kernel, a one-block fold, a 56-byte read-back).
// *all_f32 (optional): every coordinate is exactly representable in fp32. zero_buf / zero_n
// (optional): ints the fold kernel clears on the way (saves the caller a memset launch).
int cloud_bbox(Ctx* c, const double* xyz, int64_t n, double mn[3], double mx[3], bool* all_f32 = nullptr,
               int32_t* zero_buf = nullptr, int zero_n = 0);
// Shrinks box (min xyz, max xyz) to the part of the cloud that is left when at most `budget`
// points in sparse tails are given up (they clamp into the outermost cells of a grid built over
// the box); *outside = how many the last cut gave up (0: box unchanged).
int robust_box(Ctx* c, const double* xyz, int64_t n, int budget, double box[6], int64_t* outside);

// Mean number of points per occupied cell for a grid of edge `cell` over `box`
// (count-only pass on a grid of at most 4 M cells; the edge is doubled to fit and
// the edge actually used is returned). Synchronises.
int probe_occupancy(Ctx* c, const double* xyz, int64_t n, const double box[6], double cell,
                    double* cell_used, double* per_cell);

struct Stencil {
  int nx, nxy;
};

// ---- how the neighbourhood kernels read a sorted point ------------------------------------
// Both forms hand out fp64 values and evaluate d2 = ((dx*dx) + dy*dy) + dz*dz with separately
// rounded products (the library is built with -ffp-contract=off): the predicate does not know
// how the coordinates were stored.
__device__ __forceinline__ double sqdist3(double ax, double ay, double az, double bx, double by, double bz) {
  const double t0 = ax - bx, t1 = ay - by, t2 = az - bz;
  double d = t0 * t0;
  d = d + t1 * t1;
  d = d + t2 * t2;
  return d;
}
struct CoordsF64 {
  const double *x, *y, *z;
  __device__ __forceinline__ void get(int q, double& a, double& b, double& c) const {
    a = x[q];
    b = y[q];
    c = z[q];
  }
  __device__ __forceinline__ double d2(int q, double px, double py, double pz) const {
    return sqdist3(px, py, pz, x[q], y[q], z[q]);
  }
  __device__ __forceinline__ double d2(int a, int q) const {
    return sqdist3(x[a], y[a], z[a], x[q], y[q], z[q]);
  }
  // core flag of a sorted point: lives in its own byte array for this storage form
  __device__ __forceinline__ void mark_core(int, bool) const {}
  __device__ __forceinline__ bool core_pair_within(int a, int q, const uint8_t* __restrict__ core, double r2) const {
    return core[a] && core[q] && d2(a, q) <= r2;
  }
};
struct CoordsF32 {
  float4* p;  // .w: 1.0f once the point is known to be a core point (k_core_*), else 0
  __device__ __forceinline__ void get(int q, double& a, double& b, double& c) const {
    const float4 v = p[q];
    a = double(v.x);
    b = double(v.y);
    c = double(v.z);
  }
  __device__ __forceinline__ double d2(int q, double px, double py, double pz) const {
    const float4 v = p[q];
    return sqdist3(px, py, pz, double(v.x), double(v.y), double(v.z));
  }
  __device__ __forceinline__ double d2(int a, int q) const {
    const float4 u = p[a], v = p[q];
    return sqdist3(double(u.x), double(u.y), double(u.z), double(v.x), double(v.y), double(v.z));
  }
  // The core flag rides in the record's fourth word: the pair tests of the union phase then
  // touch one line per point (coordinates AND flag) instead of two.
  __device__ __forceinline__ void mark_core(int q, bool is_core) const {
    reinterpret_cast<float*>(p)[size_t(q) * 4 + 3] = is_core ? 1.0f : 0.0f;
  }
  __device__ __forceinline__ bool core_pair_within(int a, int q, const uint8_t* __restrict__, double r2) const {
    const float4 u = p[a], v = p[q];
    return u.w != 0.0f && v.w != 0.0f &&
           sqdist3(double(u.x), double(u.y), double(u.z), double(v.x), double(v.y), double(v.z)) <= r2;
  }
};

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

// Calls fn(CoordsF32{...}) or fn(CoordsF64{...}) for the storage form the grid holds.
template <class F>
inline void on_coords(const DevGrid& g, F&& fn) {
  if (g.p4)
    fn(CoordsF32{g.p4});
  else
    fn(CoordsF64{g.sx, g.sy, g.sz});
}

// Octant sub-cells: the points of every cell re-sorted by the octant (half cell per axis)
// they fall in. A sub-cell is identified by start[cell] * 8 + octant (first sorted
// position of its cell, so ids are unique and need no compaction); its points are the run
// [sub_beg[id], sub_beg[id] + sub_cnt[id]). sub_cnt is zero for empty octants of an
// occupied cell; entries of unoccupied ids are never written and must not be read.
struct SubCells {
  int32_t* sub_cnt;  // [8 n]
  int32_t* sub_beg;  // [8 n]
  int32_t* sub_of;   // [n] sub-cell id of each sorted position
  int4* rec;         // [8 n] (sub_beg, sub_cnt, -1, 0) in one 16-byte record; .z is the caller's
  // build_grid_octants only: four ints the binning's own kernels left zeroed, for the caller's
  // counters (spares DBSCAN two memset launches per step); nullptr from subsort_octants
  // (+ kZeroedExtra more zeroed ints behind the four: DBSCAN's segmented list counters)
  int32_t* zeroed4 = nullptr;
};
static constexpr int kZeroedExtra = 64;

// Re-sorts g's point arrays in place (order, sx, sy, sz are replaced by new arena arrays;
// start and cell_of stay valid: the permutation is within cells).
int subsort_octants(Ctx* c, DevGrid* g, int64_t n, SubCells* sub);

// The grid AND its octant sub-cells in one go (what DBSCAN bins with): one returning atomic per
// point (its arrival rank in the cell), one scatter of (index, cell, octant) and one pass that
// orders every cell's run by octant from the run's own octant bytes (no second counting pass, no
// per-octant counters), gathers the coordinates and writes every output array once.
// Extents that would need more than `max_cells` cells of edge `min_cell` are first COMPRESSED
// per axis: runs of empty slabs collapse to one empty slab, which keeps exactly the adjacencies
// the 27-cell stencil and the sub-cell offsets use (two points in neighbouring slabs stay in
// neighbouring slabs, all others end up at least one empty slab apart). Only if the compressed
// grid is still too large is the edge doubled (g->cell > min_cell tells the caller).
int build_grid_octants(Ctx* c, const double* xyz, int64_t n, double min_cell, int64_t max_cells,
                       DevGrid* g, SubCells* sub);

// A grid with cells `factor` times larger over the same points, derived from `fine`
// by block sums and a deterministic scatter (no atomics, no second pass over xyz).
int coarsen_grid(Ctx* c, const DevGrid& fine, int64_t n, int factor, DevGrid* coarse);

// Number of occupied cells (reads back one int; synchronises).
int count_occupied(Ctx* c, const DevGrid& g, int64_t* occupied);

}  // namespace pyqsm
