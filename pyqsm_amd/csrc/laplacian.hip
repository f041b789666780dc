// laplacian.hip — point-cloud Laplacian on gfx950.
//
// Stands in for robust_laplacian.point_cloud_laplacian(pts, mollify_factor,
// n_neighbors) as called at pyQSM/geometry/skeletonize.py:253-255,341-343
// (Sharp & Crane 2020, "A Laplacian for Nonmanifold Triangle Meshes", point
// cloud variant). robust_laplacian is not vendored by the reference and cannot
// be installed here: PARITY UNPINNED against the package. What is built:
//
//   1. k nearest neighbours of every point (knn.hip)
//   2. one WAVE per point, lane j = neighbour j (k <= 64):
//        normal      = eigenvector of the smallest eigenvalue of
//                      sum_j (p_j - p_i)(p_j - p_i)^T   (cyclic Jacobi, fp64)
//        tangent uv  = projection of p_j - p_i on a basis of the tangent plane
//        Delaunay    neighbour j is kept iff some circle through the centre and
//                    uv_j is empty of the other neighbours; circle centres are
//                    uv_j/2 + s*perp(uv_j) and every other neighbour bounds s
//                    from one side, so the test is one pass per lane
//        fan         kept neighbours sorted by angle; consecutive pairs with a
//                    positive turn give the triangles (i, a, b) incident on i
//   3. intrinsic mollification: eps = max(0, max over corners (lc - la - lb +
//      delta)), delta = mollify_factor * mean side length; every length += eps
//   4. cotangent weights from the mollified lengths (Heron), 1/3 of each
//      (every triangle of a consistent region appears in three fans), assembled
//      by row: scatter, per-row sort by (column, source corner), merge. The
//      canonical summation order makes L exactly symmetric and the result
//      independent of atomic ordering. Diagonal = -(row sum): rows sum to zero.
//      Lumped mass = (area / 3) / 3 per incident triangle.
//
//   5. before the weights are taken, the triangle soup is turned into its tufted
//      cover (two glued copies of every triangle: a closed edge-manifold surface)
//      and that surface is flipped to intrinsic Delaunay using edge lengths only;
//      all edge weights are then non-negative (see the block comment further down).
//
// The CPU oracle (oracle/pyqsm_oracle.c: orc_pc_laplacian) repeats every
// floating-point operation below in the same order, so discrete decisions
// (which neighbours form the fan) agree even in degenerate configurations.
#include "grid.hpp"

#include <cmath>

namespace pyqsm {

int knn_device(Ctx* c, const double* xyz, int64_t n, int32_t k, int32_t exclude_self,
               int32_t* idx, double* d2);

// ---- symmetric 3x3 eigen decomposition (cyclic Jacobi, fixed sweep count) ------

struct Sym3 {
  double a00, a01, a02, a11, a12, a22;
};

// Returns the unit eigenvector of the smallest eigenvalue of A.
__host__ __device__ inline void smallest_eigvec(Sym3 A, double n[3]) {
  double a[3][3] = {{A.a00, A.a01, A.a02}, {A.a01, A.a11, A.a12}, {A.a02, A.a12, A.a22}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    for (int pi = 0; pi < 3; ++pi) {
      const int p = pi == 2 ? 1 : 0, q = pi == 0 ? 1 : 2;  // (0,1), (0,2), (1,2)
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
      for (int r = 0; r < 3; ++r) {  // A <- A J
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = cs * arp - sn * arq;
        a[r][q] = sn * arp + cs * arq;
      }
      for (int r = 0; r < 3; ++r) {  // A <- J' A
        const double apr = a[p][r], aqr = a[q][r];
        a[p][r] = cs * apr - sn * aqr;
        a[q][r] = sn * apr + cs * aqr;
      }
      for (int r = 0; r < 3; ++r) {  // V <- V J
        const double vrp = v[r][p], vrq = v[r][q];
        v[r][p] = cs * vrp - sn * vrq;
        v[r][q] = sn * vrp + cs * vrq;
      }
    }
  }
  int m = 0;
  if (a[1][1] < a[m][m]) m = 1;
  if (a[2][2] < a[m][m]) m = 2;
  const double len = sqrt((v[0][m] * v[0][m] + v[1][m] * v[1][m]) + v[2][m] * v[2][m]);
  n[0] = v[0][m] / len;
  n[1] = v[1][m] / len;
  n[2] = v[2][m] / len;
}

// Orthonormal basis (e1, e2) of the plane normal to the unit vector n.
__host__ __device__ inline void tangent_basis(const double n[3], double e1[3], double e2[3]) {
  // cross n with the coordinate axis it is least aligned with
  const double ax = fabs(n[0]), ay = fabs(n[1]), az = fabs(n[2]);
  double h[3] = {0, 0, 0};
  if (ax <= ay && ax <= az) h[0] = 1.0;
  else if (ay <= az) h[1] = 1.0;
  else h[2] = 1.0;
  double c0 = n[1] * h[2] - n[2] * h[1], c1 = n[2] * h[0] - n[0] * h[2],
         c2 = n[0] * h[1] - n[1] * h[0];
  const double len = sqrt((c0 * c0 + c1 * c1) + c2 * c2);
  e1[0] = c0 / len;
  e1[1] = c1 / len;
  e1[2] = c2 / len;
  e2[0] = n[1] * e1[2] - n[2] * e1[1];
  e2[1] = n[2] * e1[0] - n[0] * e1[2];
  e2[2] = n[0] * e1[1] - n[1] * e1[0];
}

// Monotone stand-in for atan2(y, x) in [0, 4): no libm, identical on CPU and GPU.
__host__ __device__ inline double pseudo_angle(double x, double y) {
  const double s = fabs(x) + fabs(y);
  if (s == 0.0) return 0.0;
  const double p = y / s;
  if (x >= 0.0) return y >= 0.0 ? p : 4.0 + p;
  return 2.0 - p;
}

// ---- stage 1b: tangent planes, one THREAD per point -------------------------------------
// Covariance of the neighbour offsets (accumulated in neighbour order), its smallest eigenvector,
// the tangent basis (e1, e2). This used to be the head of k_fans, where all 32 (64) lanes of a
// point's group ran the same twelve Jacobi sweeps — ~6000 fp64 instructions per wave, most of the
// kernel's 6.8 ms per million points. Once per point instead of once per lane; the operations and
// their order are unchanged, so are the bits.
__global__ __launch_bounds__(256) void k_tangent_planes(int n, int k, const double* __restrict__ xyz,
                                                        const int32_t* __restrict__ nbr,
                                                        double* __restrict__ basis /*[n][6]: e1, e2*/) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
  Sym3 A = {0, 0, 0, 0, 0, 0};
  for (int l0 = 0; l0 < k; l0 += 4) {  // four neighbours' gathers side by side
    double x[4], y[4], z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nb = l0 + u < k ? nbr[size_t(i) * k + l0 + u] : n;
      const bool ok = nb < n;  // a missing neighbour contributes a zero offset, as in k_fans
      x[u] = ok ? xyz[3 * size_t(nb)] - px : 0.0;
      y[u] = ok ? xyz[3 * size_t(nb) + 1] - py : 0.0;
      z[u] = ok ? xyz[3 * size_t(nb) + 2] - pz : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (l0 + u >= k) break;
      A.a00 += x[u] * x[u];
      A.a01 += x[u] * y[u];
      A.a02 += x[u] * z[u];
      A.a11 += y[u] * y[u];
      A.a12 += y[u] * z[u];
      A.a22 += z[u] * z[u];
    }
  }
  double nrm[3], e1[3], e2[3];
  smallest_eigvec(A, nrm);
  tangent_basis(nrm, e1, e2);
  double* b = basis + 6 * size_t(i);
  b[0] = e1[0];
  b[1] = e1[1];
  b[2] = e1[2];
  b[3] = e2[0];
  b[4] = e2[1];
  b[5] = e2[2];
}

// ---- stage 2: local Delaunay fans, one wave per point ----------------------------

__device__ __forceinline__ double bcast(double v, int lane) { return __shfl(v, lane, 64); }

// PER = points per wave: 1 (64 lanes for up to 64 neighbours) or 2 (a half-wave of 32 lanes
// each, for k <= 32 — with pyQSM's 30 neighbours a whole wave per point leaves 34 lanes idle
// through three k-step loops of fp64 arithmetic). Lane s of a group holds neighbour s; a
// broadcast "from neighbour l" reads lane (group base + l).
template <int PER>
__global__ __launch_bounds__(256) void k_fans(int n, int k, const double* __restrict__ xyz,
                                              const int32_t* __restrict__ nbr,
                                              const double* __restrict__ basis /*k_tangent_planes*/,
                                              int32_t* __restrict__ tri /*[n*k][2]*/,
                                              int32_t* __restrict__ tri_count /*[n]*/) {
  constexpr int G = 64 / PER;  // lanes per point
  __shared__ int32_t sorted[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int grp = lane / G, sub = lane % G, gbase = grp * G;
  const int i = (blockIdx.x * 4 + w) * PER + grp;
  const bool live = i < n;  // (the second half of the last wave may have no point)
  const int ic = live ? i : n - 1;
  const double px = xyz[3 * ic], py = xyz[3 * ic + 1], pz = xyz[3 * ic + 2];
  int nb = n;
  double dx = 0.0, dy = 0.0, dz = 0.0;
  if (live && sub < k) {
    nb = nbr[size_t(i) * k + sub];
    if (nb < n) {
      dx = xyz[3 * nb] - px;
      dy = xyz[3 * nb + 1] - py;
      dz = xyz[3 * nb + 2] - pz;
    }
  }
  const bool valid = nb < n;
  const double* bs = basis + 6 * size_t(ic);
  const double e1[3] = {bs[0], bs[1], bs[2]}, e2[3] = {bs[3], bs[4], bs[5]};
  const double u = (dx * e1[0] + dy * e1[1]) + dz * e1[2];
  const double v = (dx * e2[0] + dy * e2[1]) + dz * e2[2];
  const double uu = u * u + v * v;
  // empty-circle interval of lane's neighbour
  double lo = -__builtin_inf(), hi = __builtin_inf();
  bool blocked = !valid || uu == 0.0;
  for (int l = 0; l < k; ++l) {
    const double ul = bcast(u, gbase + l), vl = bcast(v, gbase + l);
    const int vl_ok = __shfl(int(valid), gbase + l, 64);
    if (l == sub || !vl_ok) continue;
    const double cr = u * vl - v * ul;
    const double b = ((ul * ul + vl * vl) - (u * ul + v * vl)) * 0.5;
    if (cr > 0.0) {
      const double s = b / cr;
      hi = s < hi ? s : hi;
    } else if (cr < 0.0) {
      const double s = b / cr;
      lo = s > lo ? s : lo;
    } else if (b < 0.0) {
      blocked = true;  // a closer neighbour on the same ray hides this one
    }
  }
  const bool nat = !blocked && lo <= hi;
  const double ang = pseudo_angle(u, v);
  // rank among the kept neighbours by (angle, lane)
  int rank = 0, m = 0;
  for (int l = 0; l < k; ++l) {
    const double al = bcast(ang, gbase + l);
    const int nl = __shfl(int(nat), gbase + l, 64);
    if (!nl) continue;
    ++m;
    if (al < ang || (al == ang && l < sub)) ++rank;
  }
  if (nat) sorted[w][gbase + rank] = sub;
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS writes above have landed
  int succ = sub;
  if (nat && m >= 2) succ = sorted[w][gbase + (rank + 1 == m ? 0 : rank + 1)];
  const double us = bcast(u, gbase + succ), vs = bcast(v, gbase + succ);
  const int nbs = __shfl(nb, gbase + succ, 64);
  const bool emit = live && nat && m >= 2 && (u * vs - v * us) > 0.0 && nbs != nb;
  const unsigned long long all = __ballot(emit);
  const unsigned long long gmask = PER == 1 ? ~0ull : (0xFFFFFFFFull << gbase);
  const unsigned long long mask = all & gmask;
  if (emit) {
    const int slot = __popcll(mask & ((1ull << lane) - 1ull));
    tri[(size_t(i) * k + slot) * 2] = nb;
    tri[(size_t(i) * k + slot) * 2 + 1] = nbs;
  }
  if (live && sub == 0) tri_count[i] = __popcll(mask);
}

// ---- stage 3/4: triangle list -> lengths, eps, weights ---------------------------

// Compact (i, a, b) triangles; tri_start = exclusive scan of tri_count.
__global__ __launch_bounds__(256) void k_compact_tris(int n, int k,
                                                      const int32_t* __restrict__ tri,
                                                      const int32_t* __restrict__ tri_start,
                                                      int32_t* __restrict__ tris /*[T][3]*/) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = tri_start[i], cnt = tri_start[i + 1] - b;
  for (int s = 0; s < cnt; ++s) {
    tris[3 * size_t(b + s)] = i;
    tris[3 * size_t(b + s) + 1] = tri[(size_t(i) * k + s) * 2];
    tris[3 * size_t(b + s) + 2] = tri[(size_t(i) * k + s) * 2 + 1];
  }
}

__device__ __forceinline__ double dist3(const double* __restrict__ xyz, int a, int b) {
  const double t0 = xyz[3 * a] - xyz[3 * b], t1 = xyz[3 * a + 1] - xyz[3 * b + 1],
               t2 = xyz[3 * a + 2] - xyz[3 * b + 2];
  return sqrt((t0 * t0 + t1 * t1) + t2 * t2);
}

// Side lengths (opposite vertex 0, 1, 2), per-block partial sums of lengths and
// per-block max of the triangle-inequality slack.
__global__ __launch_bounds__(256) void k_tri_lengths(int T, const int32_t* __restrict__ tris,
                                                     const double* __restrict__ xyz,
                                                     double* __restrict__ len /*[T][3]*/,
                                                     double* __restrict__ blk_sum,
                                                     double* __restrict__ blk_slack) {
  __shared__ double s_sum[256], s_slk[256];
  int t = blockIdx.x * 256 + threadIdx.x;
  double sum = 0.0, slack = -__builtin_inf();
  if (t < T) {
    const int a = tris[3 * size_t(t)], b = tris[3 * size_t(t) + 1], c = tris[3 * size_t(t) + 2];
    const double la = dist3(xyz, b, c), lb = dist3(xyz, a, c), lc = dist3(xyz, a, b);
    len[3 * size_t(t)] = la;
    len[3 * size_t(t) + 1] = lb;
    len[3 * size_t(t) + 2] = lc;
    sum = (la + lb) + lc;
    const double s0 = la - lb - lc, s1 = lb - la - lc, s2 = lc - la - lb;
    slack = s0 > s1 ? s0 : s1;
    slack = s2 > slack ? s2 : slack;
  }
  s_sum[threadIdx.x] = sum;
  s_slk[threadIdx.x] = slack;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {  // fixed tree: deterministic
    if (threadIdx.x < off) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      const double o = s_slk[threadIdx.x + off];
      if (o > s_slk[threadIdx.x]) s_slk[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    blk_sum[blockIdx.x] = s_sum[0];
    blk_slack[blockIdx.x] = s_slk[0];
  }
}

// eps from the block partials, summed in block order (deterministic; the oracle adds them in
// the same order). One wave: 64 partials are fetched at a time, one per lane, and added one
// after the other from registers (v_readlane) — a single lane walking the 23 000 partials of a
// million points through dependent global loads took 1.7 ms.
// Fold of the per-block partials into the mollification length: one block, every thread takes a
// strided share, fixed tree (deterministic). (As one wave broadcasting one partial at a time it took
// 1.2 ms per million points: 8 200 partials, two readlanes each, serially.)
__global__ __launch_bounds__(256) void k_mollify_eps(int nblk, int T, const double* __restrict__ blk_sum,
                                                     const double* __restrict__ blk_slack, double moll,
                                                     double* __restrict__ eps_out) {
  __shared__ double s_sum[256], s_slk[256];
  double sum = 0.0, slack = -__builtin_inf();
  for (int b = threadIdx.x; b < nblk; b += 256) {
    sum += blk_sum[b];
    const double k = blk_slack[b];
    if (k > slack) slack = k;
  }
  s_sum[threadIdx.x] = sum;
  s_slk[threadIdx.x] = slack;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      const double o = s_slk[threadIdx.x + off];
      if (o > s_slk[threadIdx.x]) s_slk[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const double mean = T > 0 ? s_sum[0] / (3.0 * double(T)) : 0.0;
  const double e = s_slk[0] + mean * moll;
  eps_out[0] = e > 0.0 ? e : 0.0;
}

struct Entry {  // one off-diagonal contribution
  int32_t col;
  int32_t key;  // 4 * triangle + slot: canonical summation order
  double val;
};

// ---- tufted cover + intrinsic Delaunay flips (Sharp & Crane 2020, sections 4-5) ------
//
// Every triangle of the soup gets a front copy (face 2t, corners v0 v1 v2) and a
// back copy (face 2t+1, corners v0 v2 v1). Around each undirected edge the incident
// triangles are taken in a fixed order; the copy of triangle p that runs along the
// edge from the smaller to the larger vertex id is glued to the copy of triangle
// p+1 that runs the other way. The result is a closed oriented edge-manifold surface
// with the same vertices, on which edges that violate the intrinsic Delaunay
// condition (cot a + cot b < 0) are flipped using edge lengths only. Half of the
// cotangent Laplacian of that surface is the Laplacian of the soup; after the flips
// every edge weight is non-negative.
//
// Cover arrays, F = 2T faces: fv[f][c] vertex of corner c, fl[f][c] length of the
// edge corner c -> corner c+1, fn[f][c] = 3*g + d, the face/edge glued to it.

static constexpr double kDelaunayTol = 1e-10;
static constexpr int kMaxFlipRounds = 2000;
static constexpr int kFlipCycleLooks = 8;  // identical looks (x kFlipBatch rounds) that end a flip cycle
static constexpr int kFlipCycleMax = 32;   // ... of at most this many flips per round
static constexpr int kFlipBatch = 8;  // rounds queued between two looks at the counters

__device__ __host__ inline int nx3(int c) { return c == 2 ? 0 : c + 1; }
__device__ __host__ inline int pv3(int c) { return c == 0 ? 2 : c - 1; }

// Several clouds in one build (extract_skeleton_batch): the mollification length is a property
// of each cloud — max(0, its largest triangle-inequality slack + moll x its mean edge length) — so
// it is reduced per SEGMENT of points. Triangles are stored fan by fan in point order, so the
// triangles of segment s are the range [tstart[s], tstart[s+1]) with tstart = tcount[seg_start].
// One block per segment, fixed reduction tree (deterministic).
__global__ __launch_bounds__(256) void k_seg_eps(int S, const int32_t* __restrict__ seg_start,
                                                 const int32_t* __restrict__ tcount /*scanned*/,
                                                 const double* __restrict__ len, double moll,
                                                 int32_t* __restrict__ tstart /*[S + 1]*/,
                                                 double* __restrict__ eps_out /*[S]*/) {
  __shared__ double s_sum[256], s_slk[256];
  const int sidx = blockIdx.x;
  const int tb = tcount[seg_start[sidx]], te = tcount[seg_start[sidx + 1]];
  double sum = 0.0, slack = -__builtin_inf();
  for (int t = tb + threadIdx.x; t < te; t += 256) {
    const double la = len[3 * size_t(t)], lb = len[3 * size_t(t) + 1], lc = len[3 * size_t(t) + 2];
    sum += (la + lb) + lc;
    const double s0 = la - lb - lc, s1 = lb - la - lc, s2 = lc - la - lb;
    double m = s0 > s1 ? s0 : s1;
    m = s2 > m ? s2 : m;
    slack = m > slack ? m : slack;
  }
  s_sum[threadIdx.x] = sum;
  s_slk[threadIdx.x] = slack;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
      const double o = s_slk[threadIdx.x + off];
      if (o > s_slk[threadIdx.x]) s_slk[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int Ts = te - tb;
    const double mean = Ts > 0 ? s_sum[0] / (3.0 * double(Ts)) : 0.0;
    const double e = s_slk[0] + mean * moll;
    eps_out[sidx] = (Ts > 0 && e > 0.0) ? e : 0.0;
    tstart[sidx] = tb;
    if (sidx == S - 1) tstart[S] = te;
  }
}

// eps of every triangle (its segment found by bisection over tstart): the array k_cover_init reads
__global__ __launch_bounds__(256) void k_tri_eps(int T, int S, const int32_t* __restrict__ tstart,
                                                 const double* __restrict__ eps_seg,
                                                 double* __restrict__ eps_tri) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  int lo = 0, hi = S;  // largest s with tstart[s] <= t
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (tstart[mid] <= t) lo = mid; else hi = mid;
  }
  eps_tri[t] = eps_seg[lo];
}

// Batch builds (several clouds stacked into one array): the Laplacian of the union is block
// diagonal only if every point finds its k neighbours inside its OWN cloud. A cloud with <= k
// points, or one whose k-th own neighbour is farther away than the next cloud, would take
// neighbours across the gap, its fans would span clouds, and the per-cloud weights would act on
// a coupled system: caught here, before anything is built on it. bad[0] = 1 and bad[1] = the
// smallest offending point index.
__global__ __launch_bounds__(256) void k_check_segments(int n, int k, int S,
                                                        const int32_t* __restrict__ seg_start,
                                                        const int32_t* __restrict__ nbr,
                                                        int32_t* __restrict__ bad) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int lo = 0, hi = S;  // largest s with seg_start[s] <= i
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (seg_start[mid] <= i) lo = mid; else hi = mid;
  }
  const int b = seg_start[lo], e = seg_start[lo + 1];
  bool ok = true;
  for (int j = 0; j < k; ++j) {
    const int q = nbr[size_t(i) * k + j];
    ok = ok && q >= b && q < e;
  }
  if (!ok) {
    bad[0] = 1;
    atomicMin(&bad[1], i);
  }
}

__global__ __launch_bounds__(256) void k_cover_init(int T, const int32_t* __restrict__ tris,
                                                    const double* __restrict__ len,
                                                    const double* __restrict__ eps_p,
                                                    const double* __restrict__ eps_tri /*null: eps_p[0]*/,
                                                    int32_t* __restrict__ fv,
                                                    double* __restrict__ fl,
                                                    int32_t* __restrict__ bcount) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const double eps = eps_tri ? eps_tri[t] : eps_p[0];
  const int v0 = tris[3 * size_t(t)], v1 = tris[3 * size_t(t) + 1], v2 = tris[3 * size_t(t) + 2];
  // len[] holds the side opposite vertex 0, 1, 2
  const double la = len[3 * size_t(t)] + eps, lb = len[3 * size_t(t) + 1] + eps,
               lc = len[3 * size_t(t) + 2] + eps;
  const size_t f = size_t(2) * t, g = f + 1;
  fv[3 * f] = v0; fv[3 * f + 1] = v1; fv[3 * f + 2] = v2;
  fl[3 * f] = lc; fl[3 * f + 1] = la; fl[3 * f + 2] = lb;
  fv[3 * g] = v0; fv[3 * g + 1] = v2; fv[3 * g + 2] = v1;
  fl[3 * g] = lb; fl[3 * g + 1] = la; fl[3 * g + 2] = lc;
  // one record per undirected edge occurrence, bucketed by the smaller vertex
  atomicAdd(&bcount[v0 < v1 ? v0 : v1], 1);
  atomicAdd(&bcount[v1 < v2 ? v1 : v2], 1);
  atomicAdd(&bcount[v2 < v0 ? v2 : v0], 1);
}

struct EdgeRec {
  int32_t mx;    // larger vertex id of the edge
  int32_t code;  // 2 * (3*t + e) + (front copy runs smaller -> larger)
};

__global__ __launch_bounds__(256) void k_edge_scatter(int T, const int32_t* __restrict__ tris,
                                                      const int32_t* __restrict__ bstart,
                                                      int32_t* __restrict__ bcursor,
                                                      EdgeRec* __restrict__ rec) {
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int v[3] = {tris[3 * size_t(t)], tris[3 * size_t(t) + 1], tris[3 * size_t(t) + 2]};
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int u = v[e], w = v[nx3(e)];
    const int mn = u < w ? u : w, mx = u < w ? w : u;
    const int slot = bstart[mn] + atomicAdd(&bcursor[mn], 1);
    rec[slot] = EdgeRec{mx, 2 * (3 * t + e) + (u == mn ? 1 : 0)};
  }
}

// (face, local edge) of the copy of record `code` that runs smaller->larger (fwd)
// or larger->smaller (!fwd).
__device__ __host__ inline int cover_halfedge(int code, bool fwd) {
  const int te = code >> 1, t = te / 3, e = te % 3;
  const bool front_is_fwd = (code & 1) != 0;
  const bool use_front = fwd == front_is_fwd;
  return use_front ? 3 * (2 * t) + e : 3 * (2 * t + 1) + (2 - e);
}

// One thread per vertex bucket: sort the records by (mx, code), then glue each
// run of equal mx cyclically.
__global__ __launch_bounds__(256) void k_glue(int n, const int32_t* __restrict__ bstart,
                                              EdgeRec* __restrict__ rec,
                                              int32_t* __restrict__ fn, int min_len) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = bstart[i], e = bstart[i + 1];
  if (e - b < min_len) return;  // k_glue_wave's
  for (int a = b + 1; a < e; ++a) {
    const EdgeRec x = rec[a];
    int j = a;
    while (j > b && (rec[j - 1].mx > x.mx || (rec[j - 1].mx == x.mx && rec[j - 1].code > x.code))) {
      rec[j] = rec[j - 1];
      --j;
    }
    rec[j] = x;
  }
  int a = b;
  while (a < e) {
    int z = a;
    while (z < e && rec[z].mx == rec[a].mx) ++z;
    const int m = z - a;
    for (int p = 0; p < m; ++p) {
      const int h1 = cover_halfedge(rec[a + p].code, true);
      const int h2 = cover_halfedge(rec[a + (p + 1 == m ? 0 : p + 1)].code, false);
      fn[h1] = h2;
      fn[h2] = h1;
    }
    a = z;
  }
}

// The same for buckets of at most 64 records (all but a handful), a wave per bucket: the records are
// sorted in registers (bitonic network over the lanes, one 64-bit (mx, code) key each) instead of
// by an insertion sort that shuffles them through global memory (2-3 ms per million points), and
// every lane glues its record to the next of its run. k_glue then only takes the longer buckets
// (min_len).
__global__ __launch_bounds__(256) void k_glue_wave(int n, const int32_t* __restrict__ bstart,
                                                   const EdgeRec* __restrict__ rec,
                                                   int32_t* __restrict__ fn) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;  // whole waves
  const int b = bstart[i], m = bstart[i + 1] - b;
  if (m == 0 || m > 64) return;
  unsigned long long key = ~0ull;
  if (lane < m) {
    const EdgeRec r = rec[b + lane];
    key = ((unsigned long long)(unsigned)r.mx << 32) | (unsigned)r.code;  // both non-negative
  }
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const unsigned long long o = __shfl_xor(key, j, 64);
      const bool up = (lane & k) == 0, lower = (lane & j) == 0;
      key = (lower == up) ? (key < o ? key : o) : (key < o ? o : key);
    }
  const int mx = int(key >> 32), code = int(unsigned(key));
  const bool valid = lane < m;  // the padding keys sort behind the records
  const int prev_mx = __shfl_up(mx, 1, 64);
  const unsigned long long heads = __ballot(valid && (lane == 0 || mx != prev_mx));
  const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
  const unsigned long long later = lane == 63 ? 0ull : (heads & ~((2ull << lane) - 1ull));
  const int first = 63 - __builtin_clzll(upto | 1ull);
  const int end = later ? __ffsll(later) - 1 : m;
  const int next = lane + 1 < end ? lane + 1 : first;  // cyclic inside the run
  const int code_next = __shfl(code, next, 64);
  if (valid) {
    const int h1 = cover_halfedge(code, true);
    const int h2 = cover_halfedge(code_next, false);
    fn[h1] = h2;
    fn[h2] = h1;
  }
}

// cot of the angle opposite side lo in a triangle with sides lo, l1, l2; area by Heron
__device__ __host__ inline double heron_area(double l0, double l1, double l2) {
  const double s = ((l0 + l1) + l2) * 0.5;
  const double a2 = s * (s - l0) * (s - l1) * (s - l2);
  return a2 > 0.0 ? sqrt(a2) : 0.0;
}
__device__ __host__ inline double cot_opposite(double lo, double l1, double l2) {
  const double area = heron_area(lo, l1, l2);
  return area > 0.0 ? ((l1 * l1 + l2 * l2) - lo * lo) / (4.0 * area) : 0.0;
}

// Length of the other diagonal of the quad made of triangles (a,b,k) and (b,a,l)
// laid out in the plane: a = (0,0), b = (lab,0), k above, l below.
__device__ __host__ inline double flipped_length(double lab, double lbk, double lka, double lal,
                                                 double llb) {
  const double kx = ((lka * lka - lbk * lbk) + lab * lab) / (2.0 * lab);
  const double ky2 = lka * lka - kx * kx;
  const double ky = ky2 > 0.0 ? sqrt(ky2) : 0.0;
  const double lx = ((lal * lal - llb * llb) + lab * lab) / (2.0 * lab);
  const double ly2 = lal * lal - lx * lx;
  const double ly = ly2 > 0.0 ? sqrt(ly2) : 0.0;
  const double dx = kx - lx, dy = ky + ly;
  return sqrt(dx * dx + dy * dy);
}

// Is the edge of length lab between the triangles (lab, lbk, lka) and (lab, lal, llb) to be flipped:
// not Delaunay by more than the tolerance, and both triangles of the flipped quad proper.
__device__ __host__ inline bool flip_wanted(double lab, double lbk, double lka, double lal, double llb,
                                            double* lkl_out) {
  const double sum = cot_opposite(lab, lbk, lka) + cot_opposite(lab, lal, llb);
  if (!(sum < -kDelaunayTol)) return false;
  const double lkl = flipped_length(lab, lbk, lka, lal, llb);
  if (!(heron_area(lal, lkl, lka) > 0.0) || !(heron_area(lbk, lkl, llb) > 0.0)) return false;
  *lkl_out = lkl;
  return true;
}

struct FlipInfo {
  int f, c, g, d;
  double lab, lbk, lka, lal, llb, lkl;
};

// Is the cover edge (f, c) a legal flip candidate? Each edge is looked at from the
// side with the smaller half-edge id only.
__device__ __host__ inline bool flip_candidate(int h, const double* fl, const int32_t* fn,
                                               FlipInfo* o) {
  const int f = h / 3, c = h % 3;
  const int hg = fn[h];
  if (hg <= h) return false;  // the other side handles it (or a self-glued edge)
  const int g = hg / 3, d = hg % 3;
  if (g == f) return false;
  const double lab = fl[3 * size_t(f) + c], lbk = fl[3 * size_t(f) + nx3(c)],
               lka = fl[3 * size_t(f) + pv3(c)];
  const double lal = fl[3 * size_t(g) + nx3(d)], llb = fl[3 * size_t(g) + pv3(d)];
  double lkl;
  if (!flip_wanted(lab, lbk, lka, lal, llb, &lkl)) return false;
  *o = FlipInfo{f, c, g, d, lab, lbk, lka, lal, llb, lkl};
  return true;
}

// The same question for an outer edge of a quad that THIS thread has just flipped, asked with the
// new face's side lengths still in registers (own[slot]) and only the face on the other side read
// from memory: half-edge h_own of the new face, glued to h_nbr. As flip_candidate does, the edge is
// looked at from the side of the smaller id.
__device__ __forceinline__ bool outer_edge_wanted(int h_own, const double own[3], int h_nbr,
                                                  const double* __restrict__ fl) {
  const int so = h_own % 3, fb = h_nbr / 3, sb = h_nbr % 3;
  const double n0 = fl[3 * size_t(fb) + sb], n1 = fl[3 * size_t(fb) + nx3(sb)], n2 = fl[3 * size_t(fb) + pv3(sb)];
  double lkl;
  return h_own < h_nbr ? flip_wanted(own[so], own[nx3(so)], own[pv3(so)], n1, n2, &lkl)
                       : flip_wanted(n0, n1, n2, own[nx3(so)], own[pv3(so)], &lkl);
}

// ---- flip rounds over a work list ---------------------------------------------------
// Only edges next to a flip can change their Delaunay status, so after the seeding pass
// the rounds walk a list of half-edges instead of the whole cover: a round is the
// candidates of the list claiming their six faces, the owners flipping, and the next
// list = the losers + the five edges of every flipped quad (deduplicated by a round
// stamp). The winners of a round are the same as with a full scan (priorities are the
// half-edge ids), so the sequence of flips — and the result — does not depend on it.

// Every lane contributes `cnt` slots; returns this lane's first slot in the list.
// (One atomic per BLOCK of 256: atomics on a single address are served one at a time, and a
// round over a 2.5 M-entry list would otherwise issue 40 000 of them. Every thread of the block
// must call this.) `extra` is a second per-thread count folded the same way into counter[1].
__device__ __forceinline__ int block_reserve(int cnt, int extra, int32_t* counter) {
  __shared__ int wtot[4], wext[4], bbase;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = cnt, ex = extra;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ex += __shfl_xor(ex, off, 64);
  if (lane == 63) wtot[w] = incl;
  if (lane == 0) wext[w] = ex;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    const int et = wext[0] + wext[1] + wext[2] + wext[3];
    bbase = tot > 0 ? atomicAdd(counter, tot) : 0;
    if (et > 0) atomicAdd(counter + 1, et);
  }
  __syncthreads();
  int base = bbase;
  for (int k = 0; k < w; ++k) base += wtot[k];
  __syncthreads();  // the shared slots are reused by the next call
  return base + incl - cnt;
}

static constexpr int kSeedPer = 16;
__global__ __launch_bounds__(256) void k_flip_seed(int H, const double* __restrict__ fl,
                                                   const int32_t* __restrict__ fn,
                                                   int32_t* __restrict__ list,
                                                   int32_t* __restrict__ count) {
  // kSeedPer half-edges per thread and ONE reservation per block: a block per 256 half-edges made
  // 47 000 - 140 000 returning atomics on the one list counter, served one at a time (~10 ns each) —
  // the pass was as long as that queue (1.2 - 1.5 ms per build of a million points)
  int found[kSeedPer];
  int nf = 0;
  const int base = blockIdx.x * (256 * kSeedPer) + threadIdx.x;
#pragma unroll
  for (int u = 0; u < kSeedPer; ++u) {
    const int h = base + u * 256;
    FlipInfo q;
    if (h < H && flip_candidate(h, fl, fn, &q)) found[nf++] = h;
  }
  int slot = block_reserve(nf, 0, count);
  for (int k = 0; k < nf; ++k) list[slot++] = found[k];
}

// Round part 1: every candidate of the list claims its TWO faces; the claim carries the round in
// its upper half, so claims never need clearing. (Round 1 claimed the four outer neighbours as
// well — six atomics per candidate — and a flip needed all six. But a flip rewrites its two faces
// and ONE link slot in each outer neighbour, the slot of the shared edge: two flips that merely
// share an outer neighbour touch different slots of it and do not conflict. What must not happen
// is a flip next to a face that is itself being rewritten, i.e. two candidates whose quads share
// an edge; k_flip_apply settles those by priority with four plain loads.)
// Priority of a candidate within its round: a bijective hash of the half-edge id. With the id
// itself, a strip of k adjacent candidates flips one per round (ids grow along the mesh):
// ~450 rounds on a contracted 1 M-point cloud; hashed, the local maxima are spread out and
// a strip clears in O(log k) rounds. The final triangulation does not depend on the order.
__device__ __forceinline__ unsigned long long flip_priority(int h) {
  return (unsigned long long)(unsigned(h + 1) * 0x9E3779B1u);
}

// (The list length lives on the device — the previous round's counter — and the kernels walk
// the list with a grid stride, so several rounds are queued without a host round trip.)
__global__ __launch_bounds__(256) void k_flip_claim(const int32_t* __restrict__ m_ptr,
                                                    const int32_t* __restrict__ list,
                                                    unsigned long long stamp,
                                                    const double* __restrict__ fl,
                                                    const int32_t* __restrict__ fn,
                                                    unsigned long long* __restrict__ claim,
                                                    uint8_t* __restrict__ is_cand) {
  const int m = *m_ptr;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < m; i += gridDim.x * 256) {
    const int h = list[i];
    FlipInfo q;
    const bool cand = flip_candidate(h, fl, fn, &q);
    is_cand[i] = cand;
    if (!cand) continue;
    const unsigned long long pr = stamp | flip_priority(h);
    atomicMax(&claim[q.f], pr);
    atomicMax(&claim[q.g], pr);
  }
}

// Round part 2: a candidate that owns its two faces and outranks every claim on the four outer
// neighbours flips its edge.
//   before: f = (a, b, k) with edge c = a->b,   g = (b, a, l) with edge d = b->a
//   after:  f = (a, l, k),  g = (b, k, l)       (new edge l->k in f, k->l in g)
// A candidate that lost stays on the list; a flip puts its five edges on it.
__global__ __launch_bounds__(256) void k_flip_apply(const int32_t* __restrict__ m_ptr,
                                                    const int32_t* __restrict__ list,
                                                    const uint8_t* __restrict__ is_cand,
                                                    unsigned long long stamp, int round_id,
                                                    int32_t* __restrict__ fv,
                                                    double* __restrict__ fl,
                                                    int32_t* __restrict__ fn,
                                                    const unsigned long long* __restrict__ claim,
                                                    int32_t* __restrict__ mark,
                                                    int32_t* __restrict__ next,
                                                    int32_t* __restrict__ counts /*[0] next size, [1] flips*/,
                                                    bool push_all) {
  const int m = *m_ptr;
  // wave-uniform trip count: the appends below are wave-cooperative
  // (Staging the next list in LDS and appending a few thousand entries per reservation — 7 000 returning
  // atomics on the one counter per heavy round otherwise — changed nothing: measured, dropped.)
  for (int base = blockIdx.x * 256; base < m; base += gridDim.x * 256) {
  const int i = base + threadIdx.x;
  int push[4];
  int np = 0;
  bool flipped = false;
  if (i < m && is_cand[i]) {
    const int h = list[i];
    const int f = h / 3, c = h % 3;
    const unsigned long long pr = stamp | flip_priority(h);
    // f and g first: while both are ours nobody else rewrites them
    bool mine = claim[f] == pr;
    int g = 0, d = 0;
    if (mine) {
      const int hg = fn[h];
      g = hg / 3;
      d = hg % 3;
      mine = claim[g] == pr;
    }
    int n_bk = 0, n_ka = 0, n_al = 0, n_lb = 0;
    if (mine) {
      n_bk = fn[3 * size_t(f) + nx3(c)];
      n_ka = fn[3 * size_t(f) + pv3(c)];
      n_al = fn[3 * size_t(g) + nx3(d)];
      n_lb = fn[3 * size_t(g) + pv3(d)];
      // an outer neighbour is either one of our own two faces (glued copies of the cover) or must
      // not be claimed this round by a candidate of higher priority: of two candidates whose quads
      // share an edge each sees the other's claim on that neighbour, and exactly one yields.
      // (Older claims carry a smaller round stamp and compare lower.)
      auto free_of = [&](int code) {
        const int h2 = code / 3;
        return h2 == f || h2 == g || claim[h2] < pr;
      };
      mine = free_of(n_bk) && free_of(n_ka) && free_of(n_al) && free_of(n_lb);
    }
    if (!mine) {
      push[np++] = h;
    } else {
      FlipInfo q;
      (void)flip_candidate(h, fl, fn, &q);  // geometry of our own six faces: stable
      const int a = fv[3 * size_t(f) + c], b = fv[3 * size_t(f) + nx3(c)],
                k = fv[3 * size_t(f) + pv3(c)], l = fv[3 * size_t(g) + pv3(d)];
      // outer edges that are glued to f or g themselves move with the flip
      const int old_bk = 3 * f + nx3(c), old_ka = 3 * f + pv3(c), old_al = 3 * g + nx3(d),
                old_lb = 3 * g + pv3(d);
      auto remap = [&](int code) {
        if (code == old_bk) return 3 * g + 0;
        if (code == old_ka) return 3 * f + 2;
        if (code == old_al) return 3 * f + 0;
        if (code == old_lb) return 3 * g + 2;
        return code;
      };
      n_bk = remap(n_bk);
      n_ka = remap(n_ka);
      n_al = remap(n_al);
      n_lb = remap(n_lb);
      fv[3 * size_t(f)] = a; fv[3 * size_t(f) + 1] = l; fv[3 * size_t(f) + 2] = k;
      fl[3 * size_t(f)] = q.lal; fl[3 * size_t(f) + 1] = q.lkl; fl[3 * size_t(f) + 2] = q.lka;
      fn[3 * size_t(f)] = n_al; fn[3 * size_t(f) + 1] = 3 * g + 1; fn[3 * size_t(f) + 2] = n_ka;
      fv[3 * size_t(g)] = b; fv[3 * size_t(g) + 1] = k; fv[3 * size_t(g) + 2] = l;
      fl[3 * size_t(g)] = q.lbk; fl[3 * size_t(g) + 1] = q.lkl; fl[3 * size_t(g) + 2] = q.llb;
      fn[3 * size_t(g)] = n_bk; fn[3 * size_t(g) + 1] = 3 * f + 1; fn[3 * size_t(g) + 2] = n_lb;
      // back links of the outer neighbours (for neighbours inside {f, g} the forward
      // links written above already point the right way)
      if (n_al / 3 != f && n_al / 3 != g) fn[n_al] = 3 * f + 0;
      if (n_ka / 3 != f && n_ka / 3 != g) fn[n_ka] = 3 * f + 2;
      if (n_bk / 3 != f && n_bk / 3 != g) fn[n_bk] = 3 * g + 0;
      if (n_lb / 3 != f && n_lb / 3 != g) fn[n_lb] = 3 * g + 2;
      flipped = true;
      // The four outer edges go on the next list only if they ARE flip candidates now. The face on the
      // other side of an outer edge is not rewritten this round (a candidate that owns it yields to us
      // or we to it, see free_of), so what is seen here is what the next round's k_flip_claim will
      // see. Listing all four unseen made 60 % of every list edges that the next round looked at and
      // dropped: 45 M list entries for 9 M flips on a contracted 1 M-point cloud. (Edges glued to the
      // quad's own faces — copies of the cover folded onto each other — are listed as before.) Every
      // edge is listed under the smaller of its two half-edge ids.
      const double own_f[3] = {q.lal, q.lkl, q.lka}, own_g[3] = {q.lbk, q.lkl, q.llb};
      auto outer = [&](int h_own, const double* own, int h_nbr) {
        const int fb = h_nbr / 3;
        if (push_all || fb == f || fb == g || outer_edge_wanted(h_own, own, h_nbr, fl))
          push[np++] = min(h_own, h_nbr);
      };
      outer(3 * f + 0, own_f, n_al);
      outer(3 * f + 2, own_f, n_ka);
      outer(3 * g + 0, own_g, n_bk);
      outer(3 * g + 2, own_g, n_lb);
      // (the flipped edge itself is Delaunay now, by more than the tolerance; it comes back on a
      // list as an outer edge of whichever neighbouring flip next changes one of its faces)
    }
  }
  // stamp-deduplicated append
  int keep = 0;
  for (int p = 0; p < np; ++p)
    if (atomicExch(&mark[push[p]], round_id + 1) != round_id + 1) push[keep++] = push[p];
  int slot = block_reserve(keep, flipped ? 1 : 0, counts);
  for (int p = 0; p < keep; ++p) next[slot++] = push[p];
  }
}

__global__ __launch_bounds__(256) void k_cover_vcount(int F, const int32_t* __restrict__ fv,
                                                      int32_t* __restrict__ vcount) {
  int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  atomicAdd(&vcount[fv[3 * size_t(f)]], 2);
  atomicAdd(&vcount[fv[3 * size_t(f) + 1]], 2);
  atomicAdd(&vcount[fv[3 * size_t(f) + 2]], 2);
}

// Cotangent weights of one cover face; scatters six off-diagonal contributions into
// the rows of its three vertices. The cover counts every area twice (x 1/2) and a
// consistent region appears in three fans (x 1/3).
__global__ __launch_bounds__(256) void k_cover_weights(int F, const int32_t* __restrict__ fv,
                                                       const double* __restrict__ fl,
                                                       const int32_t* __restrict__ row_start,
                                                       int32_t* __restrict__ cursor,
                                                       Entry* __restrict__ ent,
                                                       double* __restrict__ face_area) {
  int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  const int vtx[3] = {fv[3 * size_t(f)], fv[3 * size_t(f) + 1], fv[3 * size_t(f) + 2]};
  // side opposite corner c is the edge (c+1) -> (c+2), i.e. fl[c+1]
  const double l[3] = {fl[3 * size_t(f) + 1], fl[3 * size_t(f) + 2], fl[3 * size_t(f)]};
  const double area = heron_area(l[0], l[1], l[2]);
  face_area[f] = area * 0.5;
  double wgt[3];  // weight of the edge opposite corner c
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const double lo = l[c], l1 = l[nx3(c)], l2 = l[pv3(c)];
    const double cot = area > 0.0 ? ((l1 * l1 + l2 * l2) - lo * lo) / (4.0 * area) : 0.0;
    wgt[c] = ((0.5 * cot) * 0.5) / 3.0;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int u = vtx[c], v1 = vtx[nx3(c)], v2 = vtx[pv3(c)];
    const int slot = row_start[u] + atomicAdd(&cursor[u], 2);
    ent[slot] = Entry{v1, 4 * f + pv3(c), -wgt[pv3(c)]};
    ent[slot + 1] = Entry{v2, 4 * f + nx3(c), -wgt[nx3(c)]};
  }
}

__device__ __forceinline__ bool ent_less(const Entry& a, const Entry& b) {
  return a.col < b.col || (a.col == b.col && a.key < b.key);
}

// Bitonic sort of one row by the wave, PER entries per lane (element e = lane + 64 t sits in lane
// e % 64, slot e / 64; rows of up to 64 * PER entries, PER a power of two). (col, key) is compared
// as ONE 64-bit integer. log2(N) (log2(N) + 1) / 2 compare-exchange stages — 28 for 128 entries —
// of a lane shuffle and a select per entry, where the rank sort this replaces compared every
// entry with every other one (1800 instructions for a 100-entry row, 4.6 ms per million rows;
// now ~600). Not stable: the only equal (col, key) pairs are the two ends of a loop edge of the
// flipped cover, which carry the same weight and are dropped from the row anyway.
template <int PER>
__device__ __forceinline__ void sort_row_wave(int i, int b, int m, int lane, Entry* __restrict__ ent,
                                              int32_t* __restrict__ nnz_row) {
  constexpr int N = 64 * PER;
  unsigned long long ck[PER];
  double val[PER];
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    const int idx = lane + 64 * t;
    ck[t] = ~0ull;  // padding sorts to the end
    val[t] = 0.0;
    if (idx < m) {
      const Entry x = ent[b + idx];
      ck[t] = ((unsigned long long)(unsigned)x.col << 32) | (unsigned)x.key;
      val[t] = x.val;
    }
  }
  for (int k = 2; k <= N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= 64) {  // partner in the same lane, another slot
        const int dj = j >> 6;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
          if ((t & dj) != 0) continue;
          const int t2 = t | dj;
          if (t2 >= PER) continue;
          const bool asc = ((lane + 64 * t) & k) == 0;
          const bool sw = asc ? ck[t] > ck[t2] : ck[t] < ck[t2];
          const unsigned long long a = ck[t], c2 = ck[t2];
          const double va = val[t], vc = val[t2];
          ck[t] = sw ? c2 : a;
          ck[t2] = sw ? a : c2;
          val[t] = sw ? vc : va;
          val[t2] = sw ? va : vc;
        }
      } else {  // partner in lane ^ j, same slot
        const bool lower = (lane & j) == 0;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
          const unsigned long long ok = __shfl_xor(ck[t], j, 64);
          const double ov = __shfl_xor(val[t], j, 64);
          const bool asc = ((lane + 64 * t) & k) == 0;
          const bool keep_min = asc == lower;
          const bool take = keep_min ? ok < ck[t] : ok > ck[t];
          ck[t] = take ? ok : ck[t];
          val[t] = take ? ov : val[t];
        }
      }
    }
  }
  int distinct = 0;
#pragma unroll
  for (int t = 0; t < PER; ++t) {
    const int e = lane + 64 * t;
    const bool valid = e < m;
    const int col = int(unsigned(ck[t] >> 32));
    // the entry before this one in sorted order: lane - 1, or lane 63 of the slot below
    unsigned long long prev = __shfl_up(ck[t], 1, 64);
    unsigned long long carry = 0ull;
    if (t > 0) carry = __shfl(ck[t > 0 ? t - 1 : 0], 63, 64);
    if (lane == 0) prev = carry;
    const bool first = e == 0 || int(unsigned(prev >> 32)) != col;
    if (valid) ent[b + e] = Entry{col, int(unsigned(ck[t])), val[t]};
    distinct += __popcll(__ballot(valid && first && col != i));
  }
  if (lane == 0) nnz_row[i] = distinct + 1;  // + diagonal (loop edges i-i carry no weight)
}

// Sort the contributions of every row by (col, key) and count its distinct columns:
// one wave per row, sorted in registers (sort_row_wave) instead of ~1200 dependent
// global-memory moves of a per-lane insertion sort.
static constexpr int kSortPer = 8;  // rows up to 64 * kSortPer entries take the wave path

__global__ __launch_bounds__(256) void k_sort_rows(int n, const int32_t* __restrict__ row_start,
                                                   Entry* __restrict__ ent,
                                                   int32_t* __restrict__ nnz_row) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;  // wave-uniform
  const int b = row_start[i], e = row_start[i + 1], m = e - b;
  if (m > 64 * kSortPer) {  // very long row: serial insertion sort by one lane
    if (lane == 0) {
      for (int a = b + 1; a < e; ++a) {
        const Entry x = ent[a];
        int j = a;
        while (j > b && ent_less(x, ent[j - 1])) {
          ent[j] = ent[j - 1];
          --j;
        }
        ent[j] = x;
      }
      int distinct = 0;
      for (int a = b; a < e; ++a)
        if (ent[a].col != i && (a == b || ent[a].col != ent[a - 1].col)) ++distinct;
      nnz_row[i] = distinct + 1;
    }
    return;
  }
  // the usual row has 60-160 entries
  if (m <= 64) sort_row_wave<1>(i, b, m, lane, ent, nnz_row);
  else if (m <= 128) sort_row_wave<2>(i, b, m, lane, ent, nnz_row);
  else if (m <= 256) sort_row_wave<4>(i, b, m, lane, ent, nnz_row);
  else sort_row_wave<kSortPer>(i, b, m, lane, ent, nnz_row);
}

// Per row (sorted by k_sort_rows): the merged row with its diagonal, and the lumped mass.
__global__ __launch_bounds__(256) void k_rows(int n, const int32_t* __restrict__ row_start,
                                              const Entry* __restrict__ ent,
                                              const double* __restrict__ tri_area,
                                              const int32_t* __restrict__ indptr,
                                              int32_t* __restrict__ indices,
                                              double* __restrict__ vals,
                                              double* __restrict__ mass) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int b = row_start[i], e = row_start[i + 1];
  // merged off-diagonals in column order, diagonal inserted in place
  double m = 0.0;
  for (int a = b; a < e; ++a) m += tri_area[ent[a].key >> 2] * 0.5;  // each face twice
  mass[i] = (m / 3.0) / 3.0;
  int w = indptr[i];
  double diag = 0.0;
  bool diag_done = false;
  int diag_pos = -1;
  int a = b;
  while (a < e) {
    const int col = ent[a].col;
    double s = 0.0;
    while (a < e && ent[a].col == col) s += ent[a++].val;
    if (col == i) continue;  // a loop edge of the flipped cover: cancels in L
    if (!diag_done && col > i) {
      diag_pos = w++;
      diag_done = true;
    }
    indices[w] = col;
    vals[w] = s;
    ++w;
    diag -= s;
  }
  if (!diag_done) diag_pos = w++;
  indices[diag_pos] = i;
  vals[diag_pos] = diag;
}

// The same, a wave per row (four rows per block): a thread per row walked ~100 sorted 16-byte entries
// and as many random face areas one after the other (2.2-3.7 ms per million rows); here the lanes
// load 64 entries at a time, equal columns are summed by a segmented scan over the lanes (in the
// sorted order: a fixed summation tree, the same bits on every run), a column that continues
// into the next chunk is carried, and the merged entries are written at the place their rank
// among the row's distinct columns gives them, the diagonal slotted in before the first column
// beyond it.
__global__ __launch_bounds__(256) void k_rows_wave(int n, const int32_t* __restrict__ row_start,
                                                   const Entry* __restrict__ ent,
                                                   const double* __restrict__ tri_area,
                                                   const int32_t* __restrict__ indptr,
                                                   int32_t* __restrict__ indices,
                                                   double* __restrict__ vals,
                                                   double* __restrict__ mass) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;  // whole waves
  const int b = row_start[i], e = row_start[i + 1];
  // lumped mass: a third of the faces around the vertex (each face is listed twice)
  double m = 0.0;
  for (int a = b + lane; a < e; a += 64) m += tri_area[ent[a].key >> 2] * 0.5;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m += __shfl_xor(m, off, 64);
  if (lane == 0) mass[i] = (m / 3.0) / 3.0;
  const int w0 = indptr[i];
  int written = 0;      // merged off-diagonals written so far
  int below = 0;        // ... of them with a column below i
  double diag = 0.0;    // minus their sum (accumulated in lane 0's copy; all lanes keep the same value)
  bool have_carry = false;
  int carry_col = 0;
  double carry_sum = 0.0;
  auto emit = [&](int col, double sum, int rank_in_batch) {  // one merged entry
    const int pos = w0 + written + rank_in_batch + (col > i ? 1 : 0);
    indices[pos] = col;
    vals[pos] = sum;
  };
  for (int base = b; base < e; base += 64) {
    const int a = base + lane;
    const bool valid = a < e;
    const Entry en = valid ? ent[a] : Entry{0x7FFFFFFF, 0, 0.0};
    const int col = en.col;
    const int prev_col = __shfl_up(col, 1, 64);
    // a lane starts a run when its column differs from the lane before it (lane 0: from the carry)
    const bool head = valid && (lane == 0 ? !(have_carry && col == carry_col) : col != prev_col);
    const bool joins_carry = have_carry && __shfl(col, 0, 64) == carry_col;  // wave-uniform
    // a carried column that does not continue here is complete: it goes out first
    if (have_carry && !joins_carry) {
      if (carry_col != i) {
        if (lane == 0) emit(carry_col, carry_sum, 0);
        written += 1;
        below += carry_col < i ? 1 : 0;
        diag -= carry_sum;
      }
      have_carry = false;
    }
    // Runs inside the chunk: lane 0 starts one in any case (a run that continues the carry takes
    // the carried sum first). The lane that starts a run adds its entries one after the other, in
    // the sorted (column, key) order — the order the thread-per-row kernel used, and the SAME order
    // for L_ij and L_ji, which is what keeps L symmetric to the bit; a tree over the lanes would
    // depend on where in the chunk a run happens to sit.
    const bool starts = valid && (head || lane == 0);
    const unsigned long long start_mask = __ballot(starts);
    const int n_valid = __popcll(__ballot(valid));
    const unsigned long long later = lane == 63 ? 0ull : (start_mask & ~((2ull << lane) - 1ull));
    const int run_end = later ? __ffsll(later) - 1 : n_valid;  // one past my run's last lane
    const int my_len = starts ? run_end - lane : 0;
    int max_len = my_len;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) max_len = max(max_len, __shfl_xor(max_len, off, 64));
    const double val = valid ? en.val : 0.0;
    double sum = (lane == 0 && have_carry && joins_carry) ? carry_sum + val : val;
    for (int t = 1; t < max_len; ++t) {
      const double tv = __shfl_down(val, t, 64);
      if (t < my_len) sum += tv;
    }
    const bool final_chunk = base + 64 >= e;
    // the run at the end of a chunk that is not the row's last may go on: carried, not written
    const bool pending = starts && run_end == n_valid && !final_chunk;
    const bool out = starts && !pending && col != i;  // (col == i: a loop edge of the flipped cover, cancels in L)
    const unsigned long long outs = __ballot(out);
    if (out) emit(col, sum, __popcll(outs & ((1ull << lane) - 1ull)));
    // bookkeeping, the same in every lane
    const unsigned long long lows = __ballot(out && col < i);
    double dsum = out ? sum : 0.0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
    diag -= dsum;
    written += __popcll(outs);
    below += __popcll(lows);
    const unsigned long long pend = __ballot(pending);
    if (pend) {
      const int src = __ffsll(pend) - 1;
      have_carry = true;
      carry_col = __shfl(col, src, 64);
      carry_sum = __shfl(sum, src, 64);
    } else {
      have_carry = false;
    }
  }
  if (have_carry) {  // (only when the row's length is a multiple of 64 ... and then the last run was final: not reached)
    if (carry_col != i) {
      if (lane == 0) emit(carry_col, carry_sum, 0);
      written += 1;
      below += carry_col < i ? 1 : 0;
      diag -= carry_sum;
    }
  }
  if (lane == 0) {
    const int diag_pos = w0 + below;
    indices[diag_pos] = i;
    vals[diag_pos] = diag;
  }
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

}  // extern "C"

namespace pyqsm {

// The build itself, device-resident: d_xyz in, CSR + mass left in the context arena.
int laplacian_device(Ctx* c, const double* d_xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                     int32_t k, double moll, LapOut* out) {
  const int N = int(n);
  int32_t* d_nbr;
  PQ_TRY(c->arena.get(size_t(n) * k, &d_nbr));
  // (the build uses the neighbours' indices only: no distances are asked for — 160 MB per million points
  // that the search kernels would write, a row of 160 bytes per lane)
  double* const d_d2 = nullptr;
  {
    ProfScope ps(c, "lap_knn");
    PQ_TRY(knn_device(c, d_xyz, n, k, 1, d_nbr, d_d2));
  }
  int32_t *d_seg_start = nullptr, *d_seg_bad = nullptr;
  std::vector<int32_t> h_seg(size_t(n_seg) + 1), h_bad(2, 0);
  if (n_seg > 1) {  // the clouds of a batch must not see each other (k_check_segments)
    PQ_TRY(c->arena.get(size_t(n_seg) + 1, &d_seg_start));
    PQ_TRY(c->arena.get(2, &d_seg_bad));
    for (int64_t q = 0; q <= n_seg; ++q) h_seg[size_t(q)] = int32_t(seg_start[q]);
    h_bad[1] = 0x7fffffff;
    PQ_HIP(hipMemcpyAsync(d_seg_start, h_seg.data(), (size_t(n_seg) + 1) * 4, hipMemcpyHostToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(d_seg_bad, h_bad.data(), 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_check_segments, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, N, k, int(n_seg),
                       d_seg_start, d_nbr, d_seg_bad);
    PQ_HIP(hipGetLastError());
  }
  int32_t *d_tri, *d_tcount;
  PQ_TRY(c->arena.get(size_t(n) * k * 2, &d_tri));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_tcount));
  PQ_HIP(hipMemsetAsync(d_tcount, 0, (size_t(n) + 1) * 4, c->stream));
  {
    ProfScope ps(c, "lap_fans");
    double* d_basis = nullptr;
    PQ_TRY(c->arena.get(size_t(n) * 6, &d_basis));
    hipLaunchKernelGGL(k_tangent_planes, dim3(ceil_div(n, 256)), dim3(256), 0, c->stream, N, k, d_xyz, d_nbr,
                       d_basis);
    if (k <= 32)
      hipLaunchKernelGGL(k_fans<2>, dim3(ceil_div(n, 8)), dim3(256), 0, c->stream, N, k, d_xyz, d_nbr, d_basis,
                         d_tri, d_tcount);
    else
      hipLaunchKernelGGL(k_fans<1>, dim3(ceil_div(n, 4)), dim3(256), 0, c->stream, N, k, d_xyz, d_nbr, d_basis,
                         d_tri, d_tcount);
    PQ_HIP(hipGetLastError());
  }
  ProfScope ps(c, "lap_assemble");
  PQ_TRY(exclusive_scan_i32(c, d_tcount, n + 1));
  int32_t T = 0;
  PQ_HIP(hipMemcpyAsync(&T, d_tcount + n, 4, hipMemcpyDeviceToHost, c->stream));
  if (n_seg > 1) PQ_HIP(hipMemcpyAsync(h_bad.data(), d_seg_bad, 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (n_seg > 1 && h_bad[0] != 0) {
    int64_t sidx = 0;
    while (sidx + 1 < n_seg && seg_start[sidx + 1] <= h_bad[1]) ++sidx;
    return fail(PYQSM_EINVAL,
                "batched Laplacian: point %d of cloud %lld (%lld points) has one of its %d nearest neighbours in "
                "another cloud - a cloud of a batch needs more than k points and a gap to the others wider than "
                "its own diameter (send it through the single-cloud call)",
                int(h_bad[1]), (long long)sidx, (long long)(seg_start[sidx + 1] - seg_start[sidx]), int(k));
  }
  int32_t *d_tris, *d_vcount, *d_cursor, *d_nnzrow, *d_fv, *d_fn, *d_bcount, *d_bcursor,
      *d_cnt, *d_mark, *d_list[2];
  unsigned long long* d_claim;
  uint8_t* d_iscand;
  double *d_len, *d_area, *d_blk_sum, *d_blk_slack, *d_eps, *d_mass, *d_fl;
  Entry* d_ent;
  EdgeRec* d_rec;
  const int F = 2 * T;  // cover faces
  const int nblk = ceil_div(std::max<int64_t>(T, 1), 256);
  PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_tris));
  PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_len));
  PQ_TRY(c->arena.get(size_t(F) + 1, &d_area));
  PQ_TRY(c->arena.get(size_t(nblk), &d_blk_sum));
  PQ_TRY(c->arena.get(size_t(nblk), &d_blk_slack));
  PQ_TRY(c->arena.get(1, &d_eps));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_vcount));
  PQ_TRY(c->arena.get(size_t(n), &d_cursor));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_nnzrow));
  PQ_TRY(c->arena.get(size_t(F) * 6 + 1, &d_ent));
  PQ_TRY(c->arena.get(size_t(n), &d_mass));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_fv));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_fl));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_fn));
  PQ_TRY(c->arena.get(size_t(n) + 1, &d_bcount));
  PQ_TRY(c->arena.get(size_t(n), &d_bcursor));
  PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_rec));
  PQ_TRY(c->arena.get(size_t(F) + 1, &d_claim));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_mark));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_list[0]));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_list[1]));
  PQ_TRY(c->arena.get(size_t(F) * 3 + 1, &d_iscand));
  PQ_TRY(c->arena.get(size_t(kMaxFlipRounds + 2) * 2, &d_cnt));
  PQ_HIP(hipMemsetAsync(d_vcount, 0, (size_t(n) + 1) * 4, c->stream));
  PQ_HIP(hipMemsetAsync(d_cursor, 0, size_t(n) * 4, c->stream));
  PQ_HIP(hipMemsetAsync(d_nnzrow, 0, (size_t(n) + 1) * 4, c->stream));
  PQ_HIP(hipMemsetAsync(d_bcount, 0, (size_t(n) + 1) * 4, c->stream));
  PQ_HIP(hipMemsetAsync(d_bcursor, 0, size_t(n) * 4, c->stream));
  const dim3 gn(ceil_div(n, 256)), gt(nblk), blk(256);
  const dim3 gf(ceil_div(std::max(F, 1), 256)), gh(ceil_div(std::max(3 * F, 1), 256));
  if (T > 0) {
    hipLaunchKernelGGL(k_compact_tris, gn, blk, 0, c->stream, N, k, d_tri, d_tcount, d_tris);
    hipLaunchKernelGGL(k_tri_lengths, gt, blk, 0, c->stream, T, d_tris, d_xyz, d_len, d_blk_sum,
                       d_blk_slack);
    PQ_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_mollify_eps, dim3(1), dim3(256), 0, c->stream, T > 0 ? nblk : 0, T,
                     d_blk_sum, d_blk_slack, moll, d_eps);
  double* d_eps_tri = nullptr;
  if (T > 0 && n_seg > 1) {  // one mollification length per cloud of the batch
    int32_t* d_tstart;
    double* d_eps_seg;
    PQ_TRY(c->arena.get(size_t(n_seg) + 1, &d_tstart));
    PQ_TRY(c->arena.get(size_t(n_seg), &d_eps_seg));
    PQ_TRY(c->arena.get(size_t(T), &d_eps_tri));
    hipLaunchKernelGGL(k_seg_eps, dim3(unsigned(n_seg)), blk, 0, c->stream, int(n_seg), d_seg_start, d_tcount,
                       d_len, moll, d_tstart, d_eps_seg);
    hipLaunchKernelGGL(k_tri_eps, gt, blk, 0, c->stream, T, int(n_seg), d_tstart, d_eps_seg, d_eps_tri);
    PQ_HIP(hipGetLastError());
  }
  if (T > 0) {
    // tufted cover: two faces per triangle, glued around every edge
    hipLaunchKernelGGL(k_cover_init, gt, blk, 0, c->stream, T, d_tris, d_len, d_eps, d_eps_tri, d_fv, d_fl,
                       d_bcount);
    PQ_TRY(exclusive_scan_i32(c, d_bcount, n + 1));
    hipLaunchKernelGGL(k_edge_scatter, gt, blk, 0, c->stream, T, d_tris, d_bcount, d_bcursor,
                       d_rec);
    hipLaunchKernelGGL(k_glue_wave, dim3(ceil_div(n, 4)), blk, 0, c->stream, N, d_bcount,
                       static_cast<const EdgeRec*>(d_rec), d_fn);
    hipLaunchKernelGGL(k_glue, gn, blk, 0, c->stream, N, d_bcount, d_rec, d_fn, 65);
    PQ_HIP(hipGetLastError());
    // intrinsic Delaunay flips: rounds of conflict-free flips until none is left
    ProfScope pf(c, "lap_flips");
    PQ_HIP(hipMemsetAsync(d_claim, 0, (size_t(F) + 1) * 8, c->stream));
    PQ_HIP(hipMemsetAsync(d_mark, 0, (size_t(F) * 3 + 1) * 4, c->stream));
    // counters: slot 0 = seeding pass, slot r + 1 = round r: {next list length, flips}
    PQ_HIP(hipMemsetAsync(d_cnt, 0, size_t(kMaxFlipRounds + 2) * 8, c->stream));
    hipLaunchKernelGGL(k_flip_seed, dim3(ceil_div(std::max(3 * F, 1), 256 * kSeedPer)), blk, 0, c->stream, 3 * F, d_fl,
                       d_fn, d_list[0], d_cnt);
    int32_t hc[2] = {0, 0};
    PQ_HIP(hipMemcpyAsync(hc, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    int m = hc[0];  // last list length the host has seen (sizes the next launches)
    // PYQSM_FLIP_PUSH_ALL=1: a flip lists its four outer edges unseen (the earlier form; same flips, same result)
    const char* e_push = getenv("PYQSM_FLIP_PUSH_ALL");
    const bool push_all = e_push && e_push[0] == '1';
    int last_flips = -1, same_looks = 0;
    for (int round = 0; round < kMaxFlipRounds && m > 0;) {
      // a batch of rounds between two looks at the counters; the lists shrink fast, and a
      // kernel whose list is longer than its grid covers simply strides
      const dim3 gm(unsigned(std::min<int64_t>(ceil_div(m, 256), 4096)));
      const int batch_end = std::min(round + kFlipBatch, kMaxFlipRounds);
      for (; round < batch_end; ++round) {
        const int32_t* cur = d_list[round & 1];
        int32_t* nxt = d_list[(round & 1) ^ 1];
        const unsigned long long stamp = (unsigned long long)(round + 1) << 32;
        hipLaunchKernelGGL(k_flip_claim, gm, blk, 0, c->stream, d_cnt + 2 * round, cur, stamp, d_fl,
                           d_fn, d_claim, d_iscand);
        hipLaunchKernelGGL(k_flip_apply, gm, blk, 0, c->stream, d_cnt + 2 * round, cur, d_iscand, stamp,
                           round, d_fv, d_fl, d_fn, d_claim, d_mark, nxt, d_cnt + 2 * (round + 1), push_all);
      }
      PQ_HIP(hipGetLastError());
      PQ_HIP(hipMemcpyAsync(hc, d_cnt + 2 * round, 8, hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      if (hc[1] == 0) break;  // the last round of the batch flipped nothing: done (or stuck)
      // A handful of edges flipping back and forth for good: on nearly degenerate faces of
      // a contracted cloud the rounding of the two cotangents can exceed the tolerance on
      // both sides of a flip (seen: 2 flips a round from round 88 to the cap, 35 ms per
      // build of a 50 k-point tree). The same few flips with the same list for
      // kFlipCycleLooks looks in a row end the loop; either state of such an edge is as
      // Delaunay as fp64 can tell.
      if (hc[0] == m && hc[1] == last_flips && hc[1] <= kFlipCycleMax) {
        if (++same_looks >= kFlipCycleLooks) break;
      } else {
        same_looks = 0;
      }
      last_flips = hc[1];
      m = hc[0];
      if (getenv("PYQSM_LBC_TRACE")) fprintf(stderr, "flip round %d list %d flips(last) %d\n", round, hc[0], hc[1]);
    }
    if (getenv("PYQSM_LBC_TRACE")) {  // the rounds' own counters: list length going in, flips made
      std::vector<int32_t> hcnt(size_t(kMaxFlipRounds + 2) * 2);
      PQ_HIP(hipMemcpyAsync(hcnt.data(), d_cnt, hcnt.size() * 4, hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      fprintf(stderr, "flip rounds of a cover of %d faces (list, flips):", F);
      for (int r = 0; r < kMaxFlipRounds && hcnt[2 * size_t(r)] > 0; ++r)
        fprintf(stderr, " %d:%d", hcnt[2 * size_t(r)], hcnt[2 * size_t(r) + 3]);
      fprintf(stderr, "\n");
    }
    if (getenv("PYQSM_LBC_TRACE")) {  // the rounds' own counters: list length going in, flips made
      std::vector<int32_t> hcnt(size_t(kMaxFlipRounds + 2) * 2);
      PQ_HIP(hipMemcpyAsync(hcnt.data(), d_cnt, hcnt.size() * 4, hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      fprintf(stderr, "flip rounds of a cover of %d faces (list, flips):", F);
      for (int r = 0; r < kMaxFlipRounds && hcnt[2 * size_t(r)] > 0; ++r)
        fprintf(stderr, " %d:%d", hcnt[2 * size_t(r)], hcnt[2 * size_t(r) + 3]);
      fprintf(stderr, "\n");
    }
    hipLaunchKernelGGL(k_cover_vcount, gf, blk, 0, c->stream, F, d_fv, d_vcount);
  }
  PQ_TRY(exclusive_scan_i32(c, d_vcount, n + 1));  // row_start of the contributions
  if (T > 0) {
    hipLaunchKernelGGL(k_cover_weights, gf, blk, 0, c->stream, F, d_fv, d_fl, d_vcount, d_cursor,
                       d_ent, d_area);
    PQ_HIP(hipGetLastError());
  }
  if (getenv("PYQSM_LBC_TRACE")) {  // longest row of contributions
    std::vector<int32_t> rs(size_t(n) + 1);
    PQ_HIP(hipMemcpyAsync(rs.data(), d_vcount, (size_t(n) + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    int mx = 0;
    int64_t over = 0;
    for (int64_t i = 0; i < n; ++i) {
      const int len = rs[size_t(i) + 1] - rs[size_t(i)];
      mx = std::max(mx, len);
      over += len > 64 * kSortPer;
    }
    fprintf(stderr, "laplacian rows: %d contributions at most, %lld rows beyond the wave path\n", mx,
            (long long)over);
  }
  hipLaunchKernelGGL(k_sort_rows, dim3(ceil_div(n, 4)), blk, 0, c->stream, N, d_vcount, d_ent,
                     d_nnzrow);
  PQ_HIP(hipGetLastError());
  PQ_TRY(exclusive_scan_i32(c, d_nnzrow, n + 1));  // indptr
  int32_t nnz = 0;
  PQ_HIP(hipMemcpyAsync(&nnz, d_nnzrow + n, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  int32_t* d_indices;
  double* d_vals;
  PQ_TRY(c->arena.get(size_t(nnz) + 1, &d_indices));
  PQ_TRY(c->arena.get(size_t(nnz) + 1, &d_vals));
  static const bool rows_wave = [] { const char* e = getenv("PYQSM_LAP_ROWS"); return !(e && e[0] == '0'); }();
  if (rows_wave)
    hipLaunchKernelGGL(k_rows_wave, dim3(ceil_div(n, 4)), blk, 0, c->stream, N, d_vcount, d_ent, d_area, d_nnzrow,
                       d_indices, d_vals, d_mass);
  else  // PYQSM_LAP_ROWS=0: a thread per row
    hipLaunchKernelGGL(k_rows, gn, blk, 0, c->stream, N, d_vcount, d_ent, d_area, d_nnzrow, d_indices,
                       d_vals, d_mass);
  PQ_HIP(hipGetLastError());
  out->indptr = d_nnzrow;
  out->indices = d_indices;
  out->vals = d_vals;
  out->mass = d_mass;
  out->nnz = nnz;
  return 0;
}

}  // namespace pyqsm

extern "C" {

static int pc_laplacian_impl(const double* xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                             int32_t k, double moll, int64_t* nnz_out, int32_t** indptr_out,
                             int32_t** indices_out, double** vals_out, double* mass, int32_t device) {
  if (n < 0) return fail(PYQSM_EINVAL, "negative size");
  if (n_seg > 1) {
    if (!seg_start) return fail(PYQSM_EINVAL, "pyqsm_pc_laplacian_seg: NULL seg_start");
    if (seg_start[0] != 0 || seg_start[n_seg] != n) return fail(PYQSM_EINVAL, "seg_start must run from 0 to n");
    for (int64_t q = 0; q < n_seg; ++q)
      if (seg_start[q + 1] < seg_start[q]) return fail(PYQSM_EINVAL, "seg_start must be non-decreasing");
  }
  if (!nnz_out || !indptr_out || !indices_out || !vals_out)
    return fail(PYQSM_EINVAL, "pyqsm_pc_laplacian: NULL out-parameter");
  *nnz_out = 0;
  *indptr_out = *indices_out = nullptr;
  *vals_out = nullptr;
  if (k < 3 || k > 64) return fail(PYQSM_ERANGE, "n_neighbors must be in [3, 64]");
  if (n > 0 && (!xyz || !mass)) return fail(PYQSM_EINVAL, "pyqsm_pc_laplacian: NULL pointer");
  if (n > (int64_t(1) << 30) / k) return fail(PYQSM_ERANGE, "n * n_neighbors exceeds 2^30");
  if (!(moll >= 0)) return fail(PYQSM_EINVAL, "mollify factor must be >= 0");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  int32_t* h_indptr = static_cast<int32_t*>(out_alloc((size_t(n) + 1) * 4));
  if (!h_indptr) return fail(PYQSM_ENOMEM, "host allocation failed");
  if (n == 0) {
    h_indptr[0] = 0;
    *indptr_out = h_indptr;
    *indices_out = static_cast<int32_t*>(malloc(4));
    *vals_out = static_cast<double*>(malloc(8));
    return 0;
  }
  int rc = 0;
  auto body = [&]() -> int {
    double* d_xyz;
    PQ_TRY(c->arena.get(size_t(n) * 3, &d_xyz));
    PQ_HIP(hipMemcpyAsync(d_xyz, xyz, size_t(n) * 24, hipMemcpyHostToDevice, c->stream));
    LapOut lo;
    PQ_TRY(laplacian_device(c, d_xyz, n, seg_start, n_seg, k, moll, &lo));
    const int32_t nnz = lo.nnz;
    int32_t *d_nnzrow = lo.indptr, *d_indices = lo.indices;
    double *d_vals = lo.vals, *d_mass = lo.mass;
    int32_t* h_indices = static_cast<int32_t*>(out_alloc((size_t(nnz) + 1) * 4));
    double* h_vals = static_cast<double*>(out_alloc((size_t(nnz) + 1) * 8));
    if (!h_indices || !h_vals) {
      out_free(h_indices);
      out_free(h_vals);
      return fail(PYQSM_ENOMEM, "host allocation failed");
    }
    *indices_out = h_indices;
    *vals_out = h_vals;
    PQ_HIP(hipMemcpyAsync(h_indptr, d_nnzrow, (size_t(n) + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(h_indices, d_indices, size_t(nnz) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(h_vals, d_vals, size_t(nnz) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(mass, d_mass, size_t(n) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    *nnz_out = nnz;
    return 0;
  };
  rc = body();
  if (rc != 0) {
    out_free(h_indptr);
    out_free(*indices_out);
    out_free(*vals_out);
    *indices_out = nullptr;
    *vals_out = nullptr;
    return rc;
  }
  *indptr_out = h_indptr;
  return 0;
}

int pyqsm_pc_laplacian(const double* xyz, int64_t n, int32_t k, double moll, int64_t* nnz_out,
                       int32_t** indptr_out, int32_t** indices_out, double** vals_out,
                       double* mass, int32_t device) {
  PQ_API_RANGE("pyqsm_pc_laplacian");
  return pc_laplacian_impl(xyz, n, nullptr, 1, k, moll, nnz_out, indptr_out, indices_out, vals_out, mass,
                           device);
}

int pyqsm_pc_laplacian_seg(const double* xyz, int64_t n, const int64_t* seg_start, int64_t n_seg,
                           int32_t k, double moll, int64_t* nnz_out, int32_t** indptr_out,
                           int32_t** indices_out, double** vals_out, double* mass, int32_t device) {
  PQ_API_RANGE("pyqsm_pc_laplacian_seg");
  if (n_seg < 1) return fail(PYQSM_EINVAL, "n_seg must be >= 1");
  return pc_laplacian_impl(xyz, n, seg_start, n_seg, k, moll, nnz_out, indptr_out, indices_out, vals_out,
                           mass, device);
}

}  // extern "C"
