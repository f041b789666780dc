/*
 * pyqsm_oracle.c — CPU restatement of the engines pyQSM's hot path calls.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under pyqsm_amd/ may import, link or call
 * this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check (and time) the HIP path against an independent
 * CPU statement of the same algorithm.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   orc_dbscan     follows scikit-learn's DBSCAN as called at
 *                  pyQSM/math_utils/fit.py:223 (radius neighbourhoods, then the
 *                  sequential expansion of sklearn/cluster/_dbscan_inner.pyx);
 *                  PINNED: tests/golden/dbscan_*.npz were produced by
 *                  scikit-learn 1.7.2 itself (tests/golden/make_golden.py).
 *   orc_knn        follows scipy.spatial.cKDTree.query as called at
 *                  pyQSM/geometry/reconstruction.py:238-240; PINNED the same way.
 *   orc_cast_rays  Moller-Trumbore closest hit standing in for Open3D/Embree
 *                  (pyQSM/viz/ray_casting.py:275-279). Open3D is not installable
 *                  here: PARITY UNPINNED against Embree; pinned only by analytic
 *                  known answers in tests/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* uniform grid over a cloud (acceleration only; results do not depend on it) */

typedef struct {
  double min[3];
  double inv_cell;
  int64_t dim[3];
  int64_t ncell;
  int64_t* start; /* [ncell+1] */
  int64_t* order; /* [n] point ids sorted by cell */
} grid_t;

static int64_t cell_coord(double x, double mn, double inv, int64_t dim) {
  int64_t c = (int64_t)floor((x - mn) * inv);
  if (c < 0) c = 0;
  if (c >= dim) c = dim - 1;
  return c;
}

static int grid_build(grid_t* g, const double* xyz, int64_t n, double cell) {
  double mx[3];
  for (int a = 0; a < 3; ++a) {
    g->min[a] = INFINITY;
    mx[a] = -INFINITY;
  }
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      double v = xyz[3 * i + a];
      if (v < g->min[a]) g->min[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  /* grow the cell until the dense grid is affordable */
  for (;;) {
    g->inv_cell = 1.0 / cell;
    double tot = 1.0;
    for (int a = 0; a < 3; ++a) {
      double d = floor((mx[a] - g->min[a]) * g->inv_cell) + 1.0;
      if (!(d >= 1.0)) d = 1.0;
      g->dim[a] = (int64_t)d;
      tot *= d;
    }
    if (tot <= 2.0e8) break;
    cell *= 2.0;
  }
  g->ncell = g->dim[0] * g->dim[1] * g->dim[2];
  g->start = (int64_t*)calloc((size_t)g->ncell + 1, sizeof(int64_t));
  g->order = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  int64_t* cid = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  if (!g->start || !g->order || !cid) return -1;
  for (int64_t i = 0; i < n; ++i) {
    int64_t cx = cell_coord(xyz[3 * i], g->min[0], g->inv_cell, g->dim[0]);
    int64_t cy = cell_coord(xyz[3 * i + 1], g->min[1], g->inv_cell, g->dim[1]);
    int64_t cz = cell_coord(xyz[3 * i + 2], g->min[2], g->inv_cell, g->dim[2]);
    cid[i] = (cz * g->dim[1] + cy) * g->dim[0] + cx;
    g->start[cid[i] + 1]++;
  }
  for (int64_t c = 0; c < g->ncell; ++c) g->start[c + 1] += g->start[c];
  int64_t* fill = (int64_t*)malloc((size_t)g->ncell * sizeof(int64_t));
  if (!fill) return -1;
  memcpy(fill, g->start, (size_t)g->ncell * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) g->order[fill[cid[i]]++] = i; /* ascending ids per cell */
  free(fill);
  free(cid);
  return 0;
}

static void grid_free(grid_t* g) {
  free(g->start);
  free(g->order);
}

/* squared distance exactly as sklearn's EuclideanDistance.rdist and scipy's
 * cKDTree accumulate it: d = 0; d += t*t for each axis in order, in double,
 * no fused multiply-add. */
static inline double sqdist(const double* a, const double* b) {
  double t0 = a[0] - b[0], t1 = a[1] - b[1], t2 = a[2] - b[2];
  double d = t0 * t0;
  d = d + t1 * t1;
  d = d + t2 * t2;
  return d;
}

/* ------------------------------------------------------------------------ */
/* DBSCAN                                                                     */

/*
 * labels i64 [n] (-1 noise), is_core u8 [n]. Returns the number of clusters or
 * -1 on allocation failure.
 *   step 1  neighbourhoods N(i) = { j : sqdist(i,j) <= eps*eps }, i included
 *           (sklearn NearestNeighbors.radius_neighbors, reduced distance compare)
 *   step 2  core(i) <=> |N(i)| >= min_pts
 *   step 3  sklearn/cluster/_dbscan_inner.pyx: for i ascending, unlabelled core
 *           points seed a depth-first expansion; a popped point takes the label
 *           if it has none; only core points push their unlabelled neighbours.
 */
static int64_t dbscan_impl(const double* xyz, int64_t n, double eps, int32_t min_pts, int strict,
                           int64_t* labels, uint8_t* is_core) {
  if (n == 0) return 0;
  grid_t g;
  /* cell a hair wider than eps so that rounding in the cell index can never
   * separate two points that are within eps of each other by two cells */
  if (grid_build(&g, xyz, n, eps * (1.0 + 1.0 / 1048576.0)) != 0) return -1;
  const double r2 = eps * eps;
  int64_t* nstart = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
  if (!nstart) return -1;
  /* pass 1: counts; pass 2: fill */
  int32_t* nbr = NULL;
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      for (int64_t i = 0; i < n; ++i) nstart[i + 1] += nstart[i];
      nbr = (int32_t*)malloc((size_t)(nstart[n] > 0 ? nstart[n] : 1) * sizeof(int32_t));
      if (!nbr) return -1;
    }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int64_t i = 0; i < n; ++i) {
      const double* p = xyz + 3 * i;
      int64_t cx = cell_coord(p[0], g.min[0], g.inv_cell, g.dim[0]);
      int64_t cy = cell_coord(p[1], g.min[1], g.inv_cell, g.dim[1]);
      int64_t cz = cell_coord(p[2], g.min[2], g.inv_cell, g.dim[2]);
      int64_t cnt = 0;
      int64_t w = pass == 1 ? nstart[i] : 0;
      for (int64_t z = cz - 1; z <= cz + 1; ++z) {
        if (z < 0 || z >= g.dim[2]) continue;
        for (int64_t y = cy - 1; y <= cy + 1; ++y) {
          if (y < 0 || y >= g.dim[1]) continue;
          int64_t x0 = cx > 0 ? cx - 1 : 0, x1 = cx + 1 < g.dim[0] ? cx + 1 : g.dim[0] - 1;
          int64_t row = (z * g.dim[1] + y) * g.dim[0];
          for (int64_t s = g.start[row + x0]; s < g.start[row + x1 + 1]; ++s) {
            int64_t j = g.order[s];
            const double d2 = sqdist(p, xyz + 3 * j);
            if (strict ? d2 < r2 : d2 <= r2) {
              if (pass == 1) nbr[w++] = (int32_t)j;
              ++cnt;
            }
          }
        }
      }
      if (pass == 0) nstart[i + 1] = cnt;
    }
  }
  for (int64_t i = 0; i < n; ++i) {
    is_core[i] = (nstart[i + 1] - nstart[i]) >= min_pts;
    labels[i] = -1;
  }
  int64_t* stack = (int64_t*)malloc((size_t)(nstart[n] + n + 1) * sizeof(int64_t));
  if (!stack) return -1;
  int64_t label_num = 0;
  for (int64_t s = 0; s < n; ++s) {
    if (labels[s] != -1 || !is_core[s]) continue;
    int64_t sp = 0, i = s;
    for (;;) {
      if (labels[i] == -1) {
        labels[i] = label_num;
        if (is_core[i])
          for (int64_t q = nstart[i]; q < nstart[i + 1]; ++q) {
            int64_t v = nbr[q];
            if (labels[v] == -1) stack[sp++] = v;
          }
      }
      if (sp == 0) break;
      i = stack[--sp];
    }
    ++label_num;
  }
  free(stack);
  free(nbr);
  free(nstart);
  grid_free(&g);
  return label_num;
}

/* scikit-learn's neighbourhood: d2 <= eps^2 (radius_neighbors is inclusive) */
int64_t orc_dbscan(const double* xyz, int64_t n, double eps, int32_t min_pts, int64_t* labels,
                   uint8_t* is_core) {
  return dbscan_impl(xyz, n, eps, min_pts, 0, labels, is_core);
}

/* The same clustering with the STRICT neighbourhood d2 < eps^2: what Open3D's cluster_dbscan
 * (pyQSM/geometry/point_cloud_processing.py:185,209) does if nanoflann's radius search compares
 * strictly, as SURVEY.md §8 a2 suspects. Open3D is not installable here: PARITY UNPINNED, this
 * is the switch a maintainer with Open3D at hand can flip (pyqsm_dbscan_ex, radius_inclusive = 0). */
int64_t orc_dbscan_strict(const double* xyz, int64_t n, double eps, int32_t min_pts, int64_t* labels,
                          uint8_t* is_core) {
  return dbscan_impl(xyz, n, eps, min_pts, 1, labels, is_core);
}

/* ------------------------------------------------------------------------ */
/* kNN                                                                        */

typedef struct {
  double d2;
  int32_t id;
} cand_t;

static inline int cand_less(double d2a, int32_t ia, double d2b, int32_t ib) {
  return d2a < d2b || (d2a == d2b && ia < ib);
}

/*
 * idx i32 [n,k], d2 f64 [n,k], ascending by (d2, index). Pads with idx = n,
 * d2 = +inf. The search visits grid shells of growing radius until the k-th
 * distance is provably final (<= (r*cell)^2), so the result is exact.
 */
int orc_knn(const double* xyz, int64_t n, int32_t k, int32_t exclude_self, int32_t* idx,
            double* d2out) {
  if (n == 0 || k <= 0) return 0;
  /* aim for a handful of points per occupied cell */
  double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      double v = xyz[3 * i + a];
      if (v < mn[a]) mn[a] = v;
      if (v > mx[a]) mx[a] = v;
    }
  double ext = 0;
  for (int a = 0; a < 3; ++a)
    if (mx[a] - mn[a] > ext) ext = mx[a] - mn[a];
  if (!(ext > 0)) ext = 1.0;
  /* start coarse and halve the cell until an occupied cell holds a handful of points */
  double cell = ext / 16.0;
  grid_t g;
  for (int it = 0;; ++it) {
    if (grid_build(&g, xyz, n, cell) != 0) return -1;
    int64_t occ = 0;
    for (int64_t c = 0; c < g.ncell; ++c) occ += g.start[c + 1] > g.start[c];
    double per = (double)n / (double)(occ > 0 ? occ : 1);
    if (per <= 12.0 || it >= 8 || cell * 0.5 * 1024.0 < ext) break;
    cell *= 0.5;
    grid_free(&g);
  }
  cell = 1.0 / g.inv_cell;
  int64_t maxdim = g.dim[0];
  if (g.dim[1] > maxdim) maxdim = g.dim[1];
  if (g.dim[2] > maxdim) maxdim = g.dim[2];
#pragma omp parallel
  {
    cand_t* best = (cand_t*)malloc((size_t)k * sizeof(cand_t));
#pragma omp for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; ++i) {
      const double* p = xyz + 3 * i;
      int64_t c[3];
      for (int a = 0; a < 3; ++a) c[a] = cell_coord(p[a], g.min[a], g.inv_cell, g.dim[a]);
      int32_t have = 0;
      for (int64_t r = 0;; ++r) {
        /* visit the shell at Chebyshev distance r */
        for (int64_t z = c[2] - r; z <= c[2] + r; ++z) {
          if (z < 0 || z >= g.dim[2]) continue;
          for (int64_t y = c[1] - r; y <= c[1] + r; ++y) {
            if (y < 0 || y >= g.dim[1]) continue;
            int shell_yz = (z == c[2] - r || z == c[2] + r || y == c[1] - r || y == c[1] + r);
            for (int64_t x = c[0] - r; x <= c[0] + r; x += (shell_yz ? 1 : (r > 0 ? 2 * r : 1))) {
              if (x < 0 || x >= g.dim[0]) continue;
              int64_t cc = (z * g.dim[1] + y) * g.dim[0] + x;
              for (int64_t s = g.start[cc]; s < g.start[cc + 1]; ++s) {
                int64_t j = g.order[s];
                if (exclude_self && j == i) continue;
                double d = sqdist(p, xyz + 3 * j);
                if (have < k) {
                  int32_t q = have++;
                  while (q > 0 && cand_less(d, (int32_t)j, best[q - 1].d2, best[q - 1].id)) {
                    best[q] = best[q - 1];
                    --q;
                  }
                  best[q].d2 = d;
                  best[q].id = (int32_t)j;
                } else if (cand_less(d, (int32_t)j, best[k - 1].d2, best[k - 1].id)) {
                  int32_t q = k - 1;
                  while (q > 0 && cand_less(d, (int32_t)j, best[q - 1].d2, best[q - 1].id)) {
                    best[q] = best[q - 1];
                    --q;
                  }
                  best[q].d2 = d;
                  best[q].id = (int32_t)j;
                }
              }
            }
          }
        }
        /* every point outside the visited cube is farther than r*cell (minus a
         * rounding hair, hence the 0.999999) */
        double safe = (double)r * cell * 0.999999;
        if (have == k && best[k - 1].d2 <= safe * safe) break;
        if (r > maxdim) break;
        if (r >= 6) { /* isolated point: finish with a plain scan of the whole cloud */
          have = 0;
          for (int64_t j = 0; j < n; ++j) {
            if (exclude_self && j == i) continue;
            double d = sqdist(p, xyz + 3 * j);
            if (have == k && !cand_less(d, (int32_t)j, best[k - 1].d2, best[k - 1].id)) continue;
            int32_t q = have < k ? have++ : k - 1;
            while (q > 0 && cand_less(d, (int32_t)j, best[q - 1].d2, best[q - 1].id)) {
              best[q] = best[q - 1];
              --q;
            }
            best[q].d2 = d;
            best[q].id = (int32_t)j;
          }
          break;
        }
      }
      for (int32_t q = 0; q < k; ++q) {
        idx[i * k + q] = q < have ? best[q].id : (int32_t)n;
        d2out[i * k + q] = q < have ? best[q].d2 : INFINITY;
      }
    }
    free(best);
  }
  grid_free(&g);
  return 0;
}

/* ------------------------------------------------------------------------ */
/* ray x triangle                                                             */

/* One ray against one triangle, the operation sequence of
 * pyqsm_amd/csrc/raycast.hip (see its header for the contract). */
static inline int mt(const float* o, const float* d, const float* v0, const float* e1,
                     const float* e2, float* t, float* u, float* v) {
  float px = __builtin_fmaf(d[1], e2[2], -(d[2] * e2[1]));
  float py = __builtin_fmaf(d[2], e2[0], -(d[0] * e2[2]));
  float pz = __builtin_fmaf(d[0], e2[1], -(d[1] * e2[0]));
  float det = __builtin_fmaf(e1[0], px, __builtin_fmaf(e1[1], py, e1[2] * pz));
  float tx = o[0] - v0[0], ty = o[1] - v0[1], tz = o[2] - v0[2];
  float U = __builtin_fmaf(tx, px, __builtin_fmaf(ty, py, tz * pz));
  float qx = __builtin_fmaf(ty, e1[2], -(tz * e1[1]));
  float qy = __builtin_fmaf(tz, e1[0], -(tx * e1[2]));
  float qz = __builtin_fmaf(tx, e1[1], -(ty * e1[0]));
  float V = __builtin_fmaf(d[0], qx, __builtin_fmaf(d[1], qy, d[2] * qz));
  float Tn = __builtin_fmaf(e2[0], qx, __builtin_fmaf(e2[1], qy, e2[2] * qz));
  float W = det - (U + V);
  int pos = det > 0.f && U >= 0.f && V >= 0.f && W >= 0.f && Tn > 0.f;
  int neg = det < 0.f && U <= 0.f && V <= 0.f && W <= 0.f && Tn < 0.f;
  if (!(pos || neg)) return 0;
  *t = Tn / det;
  *u = U / det;
  *v = V / det;
  return 1;
}

static float* expand_tris(const float* verts, const int32_t* tris, int64_t T) {
  float* rec = (float*)malloc((size_t)(T > 0 ? T : 1) * 9 * sizeof(float));
  if (!rec) return NULL;
  for (int64_t i = 0; i < T; ++i) {
    const float* a = verts + 3 * (int64_t)tris[3 * i];
    const float* b = verts + 3 * (int64_t)tris[3 * i + 1];
    const float* c = verts + 3 * (int64_t)tris[3 * i + 2];
    for (int k = 0; k < 3; ++k) {
      rec[9 * i + k] = a[k];
      rec[9 * i + 3 + k] = b[k] - a[k];
      rec[9 * i + 6 + k] = c[k] - a[k];
    }
  }
  return rec;
}

/* Closest hit. t_hit +inf / prim 0xFFFFFFFF on miss; uv may be NULL. */
int orc_cast_rays(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                  const float* rays, int64_t R, float* t_hit, uint32_t* prim, float* uv) {
  (void)V;
  float* rec = expand_tris(verts, tris, T);
  if (!rec) return -1;
#pragma omp parallel for schedule(dynamic, 64)
  for (int64_t r = 0; r < R; ++r) {
    const float* o = rays + 6 * r;
    const float* d = o + 3;
    float bt = INFINITY, bu = 0.f, bv = 0.f;
    uint32_t bp = 0xFFFFFFFFu;
    for (int64_t j = 0; j < T; ++j) {
      float t, u, v;
      if (mt(o, d, rec + 9 * j, rec + 9 * j + 3, rec + 9 * j + 6, &t, &u, &v) && t < bt) {
        bt = t;
        bu = u;
        bv = v;
        bp = (uint32_t)j;
      }
    }
    t_hit[r] = bt;
    prim[r] = bp;
    if (uv) {
      uv[2 * r] = bu;
      uv[2 * r + 1] = bv;
    }
  }
  free(rec);
  return 0;
}

/* All hits with t > 0, ordered by ray then triangle. counts i32 [R]; if cap > 0
 * the first `cap` records are written. Returns the total number of hits. */
int64_t orc_list_intersections(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                               const float* rays, int64_t R, int32_t* counts, uint32_t* ray_ids,
                               uint32_t* prim_ids, float* ts, float* uv, int64_t cap) {
  (void)V;
  float* rec = expand_tris(verts, tris, T);
  if (!rec) return -1;
  int64_t total = 0;
  for (int64_t r = 0; r < R; ++r) {
    const float* o = rays + 6 * r;
    int32_t n = 0;
    for (int64_t j = 0; j < T; ++j) {
      float t, u, v;
      if (mt(o, o + 3, rec + 9 * j, rec + 9 * j + 3, rec + 9 * j + 6, &t, &u, &v)) {
        if (total < cap) {
          ray_ids[total] = (uint32_t)r;
          prim_ids[total] = (uint32_t)j;
          ts[total] = t;
          uv[2 * total] = u;
          uv[2 * total + 1] = v;
        }
        ++total;
        ++n;
      }
    }
    counts[r] = n;
  }
  free(rec);
  return total;
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------ */
/* point-cloud Laplacian                                                      */
/*
 * CPU statement of the algorithm of pyqsm_amd/csrc/laplacian.hip (see its header
 * for the description and for what of Sharp & Crane's construction is not built).
 * robust_laplacian is not installable here: PARITY UNPINNED against the package;
 * this oracle pins the GPU kernel to an independent serial evaluation and is
 * itself pinned by invariants and analytic answers in tests/test_laplacian_oracle.py.
 * Every floating-point operation is written in the order the kernel uses, so that
 * the discrete decisions (which neighbours form a fan) agree exactly.
 */

static void orc_smallest_eigvec(const double A[6], double n[3]) {
  double a[3][3] = {{A[0], A[1], A[2]}, {A[1], A[3], A[4]}, {A[2], A[4], A[5]}};
  double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  static const int P[3] = {0, 0, 1}, Q[3] = {1, 2, 2};
  for (int sweep = 0; sweep < 12; ++sweep)
    for (int pi = 0; pi < 3; ++pi) {
      const int p = P[pi], q = Q[pi];
      const double apq = a[p][q];
      if (apq == 0.0) continue;
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
      for (int r = 0; r < 3; ++r) {
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = cs * arp - sn * arq;
        a[r][q] = sn * arp + cs * arq;
      }
      for (int r = 0; r < 3; ++r) {
        const double apr = a[p][r], aqr = a[q][r];
        a[p][r] = cs * apr - sn * aqr;
        a[q][r] = sn * apr + cs * aqr;
      }
      for (int r = 0; r < 3; ++r) {
        const double vrp = v[r][p], vrq = v[r][q];
        v[r][p] = cs * vrp - sn * vrq;
        v[r][q] = sn * vrp + cs * vrq;
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

static void orc_tangent_basis(const double n[3], double e1[3], double e2[3]) {
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

static double orc_pseudo_angle(double x, double y) {
  const double s = fabs(x) + fabs(y);
  if (s == 0.0) return 0.0;
  const double p = y / s;
  if (x >= 0.0) return y >= 0.0 ? p : 4.0 + p;
  return 2.0 - p;
}

typedef struct {
  int32_t col, key;
  double val;
} lap_ent_t;

static int lap_ent_cmp(const void* pa, const void* pb) {
  const lap_ent_t *a = (const lap_ent_t*)pa, *b = (const lap_ent_t*)pb;
  if (a->col != b->col) return a->col < b->col ? -1 : 1;
  return a->key < b->key ? -1 : (a->key > b->key ? 1 : 0);
}

static double orc_dist3(const double* xyz, int a, int b) {
  const double t0 = xyz[3 * a] - xyz[3 * b], t1 = xyz[3 * a + 1] - xyz[3 * b + 1],
               t2 = xyz[3 * a + 2] - xyz[3 * b + 2];
  return sqrt((t0 * t0 + t1 * t1) + t2 * t2);
}

/* Local Delaunay fan of point i: writes up to k (a, b) pairs, returns the count. */
static int orc_fan(const double* xyz, int64_t n, int i, int k, const int32_t* nbr, int32_t* out) {
  double dx[64], dy[64], dz[64], u[64], v[64], ang[64];
  int valid[64], nat[64], nb[64], order[64];
  for (int l = 0; l < k; ++l) {
    nb[l] = nbr[(int64_t)i * k + l];
    valid[l] = nb[l] < n;
    dx[l] = dy[l] = dz[l] = 0.0;
    if (valid[l]) {
      dx[l] = xyz[3 * nb[l]] - xyz[3 * i];
      dy[l] = xyz[3 * nb[l] + 1] - xyz[3 * i + 1];
      dz[l] = xyz[3 * nb[l] + 2] - xyz[3 * i + 2];
    }
  }
  double A[6] = {0, 0, 0, 0, 0, 0};
  for (int l = 0; l < k; ++l) {
    A[0] += dx[l] * dx[l];
    A[1] += dx[l] * dy[l];
    A[2] += dx[l] * dz[l];
    A[3] += dy[l] * dy[l];
    A[4] += dy[l] * dz[l];
    A[5] += dz[l] * dz[l];
  }
  double nrm[3], e1[3], e2[3];
  orc_smallest_eigvec(A, nrm);
  orc_tangent_basis(nrm, e1, e2);
  for (int l = 0; l < k; ++l) {
    u[l] = (dx[l] * e1[0] + dy[l] * e1[1]) + dz[l] * e1[2];
    v[l] = (dx[l] * e2[0] + dy[l] * e2[1]) + dz[l] * e2[2];
  }
  int m = 0;
  for (int j = 0; j < k; ++j) {
    const double uu = u[j] * u[j] + v[j] * v[j];
    double lo = -INFINITY, hi = INFINITY;
    int blocked = !valid[j] || uu == 0.0;
    for (int l = 0; l < k; ++l) {
      if (l == j || !valid[l]) continue;
      const double cr = u[j] * v[l] - v[j] * u[l];
      const double b = ((u[l] * u[l] + v[l] * v[l]) - (u[j] * u[l] + v[j] * v[l])) * 0.5;
      if (cr > 0.0) {
        const double s = b / cr;
        hi = s < hi ? s : hi;
      } else if (cr < 0.0) {
        const double s = b / cr;
        lo = s > lo ? s : lo;
      } else if (b < 0.0) {
        blocked = 1;
      }
    }
    nat[j] = !blocked && lo <= hi;
    ang[j] = orc_pseudo_angle(u[j], v[j]);
    m += nat[j];
  }
  for (int j = 0; j < k; ++j) {
    if (!nat[j]) continue;
    int rank = 0;
    for (int l = 0; l < k; ++l)
      if (nat[l] && (ang[l] < ang[j] || (ang[l] == ang[j] && l < j))) ++rank;
    order[rank] = j;
  }
  int cnt = 0;
  if (m >= 2)
    for (int j = 0; j < k; ++j) { /* lane order, like the kernel's ballot */
      if (!nat[j]) continue;
      int rank = 0;
      for (int l = 0; l < k; ++l)
        if (nat[l] && (ang[l] < ang[j] || (ang[l] == ang[j] && l < j))) ++rank;
      const int s = order[rank + 1 == m ? 0 : rank + 1];
      if ((u[j] * v[s] - v[j] * u[s]) > 0.0 && nb[s] != nb[j]) {
        out[2 * cnt] = nb[j];
        out[2 * cnt + 1] = nb[s];
        ++cnt;
      }
    }
  return cnt;
}


/* ---- tufted cover + intrinsic Delaunay flips (see laplacian.hip for the description) ---- */

#define ORC_DELAUNAY_TOL 1e-10

static int nx3(int c) { return c == 2 ? 0 : c + 1; }
static int pv3(int c) { return c == 0 ? 2 : c - 1; }

static double heron_area(double l0, double l1, double l2) {
  const double s = ((l0 + l1) + l2) * 0.5;
  const double a2 = s * (s - l0) * (s - l1) * (s - l2);
  return a2 > 0.0 ? sqrt(a2) : 0.0;
}
static double cot_opposite(double lo, double l1, double l2) {
  const double area = heron_area(lo, l1, l2);
  return area > 0.0 ? ((l1 * l1 + l2 * l2) - lo * lo) / (4.0 * area) : 0.0;
}
static double flipped_length(double lab, double lbk, double lka, double lal, double llb) {
  const double kx = ((lka * lka - lbk * lbk) + lab * lab) / (2.0 * lab);
  const double ky2 = lka * lka - kx * kx;
  const double ky = ky2 > 0.0 ? sqrt(ky2) : 0.0;
  const double lx = ((lal * lal - llb * llb) + lab * lab) / (2.0 * lab);
  const double ly2 = lal * lal - lx * lx;
  const double ly = ly2 > 0.0 ? sqrt(ly2) : 0.0;
  const double dx = kx - lx, dy = ky + ly;
  return sqrt(dx * dx + dy * dy);
}

typedef struct {
  int32_t mx, code;
} edge_rec_t;

static int edge_rec_cmp(const void* pa, const void* pb) {
  const edge_rec_t *a = (const edge_rec_t*)pa, *b = (const edge_rec_t*)pb;
  if (a->mx != b->mx) return a->mx < b->mx ? -1 : 1;
  return a->code < b->code ? -1 : (a->code > b->code ? 1 : 0);
}

static int cover_halfedge(int code, int fwd) {
  const int te = code >> 1, t = te / 3, e = te % 3;
  const int front_is_fwd = (code & 1) != 0;
  const int use_front = fwd == front_is_fwd;
  return use_front ? 3 * (2 * t) + e : 3 * (2 * t + 1) + (2 - e);
}

/* Try to flip the cover edge h (looked at from its smaller half-edge id). Returns 1 if flipped. */
static int try_flip(int h, int32_t* fv, double* fl, int32_t* fn) {
  const int f = h / 3, c = h % 3;
  const int hg = fn[h];
  if (hg <= h) return 0;
  const int g = hg / 3, d = hg % 3;
  if (g == f) return 0;
  const double lab = fl[3 * f + c], lbk = fl[3 * f + nx3(c)], lka = fl[3 * f + pv3(c)];
  const double lal = fl[3 * g + nx3(d)], llb = fl[3 * g + pv3(d)];
  const double sum = cot_opposite(lab, lbk, lka) + cot_opposite(lab, lal, llb);
  if (!(sum < -ORC_DELAUNAY_TOL)) return 0;
  const double lkl = flipped_length(lab, lbk, lka, lal, llb);
  if (!(heron_area(lal, lkl, lka) > 0.0) || !(heron_area(lbk, lkl, llb) > 0.0)) return 0;
  int n_bk = fn[3 * f + nx3(c)], n_ka = fn[3 * f + pv3(c)];
  int n_al = fn[3 * g + nx3(d)], n_lb = fn[3 * g + pv3(d)];
  const int a = fv[3 * f + c], b = fv[3 * f + nx3(c)], k = fv[3 * f + pv3(c)], l = fv[3 * g + pv3(d)];
  const int old_bk = 3 * f + nx3(c), old_ka = 3 * f + pv3(c), old_al = 3 * g + nx3(d),
            old_lb = 3 * g + pv3(d);
  int* nb[4] = {&n_bk, &n_ka, &n_al, &n_lb};
  for (int q = 0; q < 4; ++q) {
    int code = *nb[q];
    if (code == old_bk) code = 3 * g + 0;
    else if (code == old_ka) code = 3 * f + 2;
    else if (code == old_al) code = 3 * f + 0;
    else if (code == old_lb) code = 3 * g + 2;
    *nb[q] = code;
  }
  fv[3 * f] = a; fv[3 * f + 1] = l; fv[3 * f + 2] = k;
  fl[3 * f] = lal; fl[3 * f + 1] = lkl; fl[3 * f + 2] = lka;
  fn[3 * f] = n_al; fn[3 * f + 1] = 3 * g + 1; fn[3 * f + 2] = n_ka;
  fv[3 * g] = b; fv[3 * g + 1] = k; fv[3 * g + 2] = l;
  fl[3 * g] = lbk; fl[3 * g + 1] = lkl; fl[3 * g + 2] = llb;
  fn[3 * g] = n_bk; fn[3 * g + 1] = 3 * f + 1; fn[3 * g + 2] = n_lb;
  if (n_al / 3 != f && n_al / 3 != g) fn[n_al] = 3 * f + 0;
  if (n_ka / 3 != f && n_ka / 3 != g) fn[n_ka] = 3 * f + 2;
  if (n_bk / 3 != f && n_bk / 3 != g) fn[n_bk] = 3 * g + 0;
  if (n_lb / 3 != f && n_lb / 3 != g) fn[n_lb] = 3 * g + 2;
  return 1;
}

/*
 * CSR out-parameters are malloc'ed (release with orc_free). Returns 0, or -1 on
 * allocation failure / bad k.
 */
int orc_pc_laplacian(const double* xyz, int64_t n, int32_t k, double moll, int64_t* nnz_out,
                     int32_t** indptr_out, int32_t** indices_out, double** vals_out,
                     double* mass) {
  if (k < 3 || k > 64) return -1;
  int32_t* nbr = (int32_t*)malloc((size_t)(n * k + 1) * sizeof(int32_t));
  double* d2 = (double*)malloc((size_t)(n * k + 1) * sizeof(double));
  int32_t* fan = (int32_t*)malloc((size_t)(n * k + 1) * 2 * sizeof(int32_t));
  int32_t* fcnt = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  if (!nbr || !d2 || !fan || !fcnt) return -1;
  if (orc_knn(xyz, n, k, 1, nbr, d2) != 0) return -1;
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n; ++i) fcnt[i] = orc_fan(xyz, n, (int)i, k, nbr, fan + 2 * i * k);
  int64_t T = 0;
  for (int64_t i = 0; i < n; ++i) T += fcnt[i];
  int32_t* tris = (int32_t*)malloc((size_t)(T + 1) * 3 * sizeof(int32_t));
  double* len = (double*)malloc((size_t)(T + 1) * 3 * sizeof(double));
  double* area = (double*)malloc((size_t)(T + 1) * sizeof(double));
  int32_t* rstart = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  if (!tris || !len || !area || !rstart) return -1;
  int64_t t = 0;
  for (int64_t i = 0; i < n; ++i)
    for (int s = 0; s < fcnt[i]; ++s, ++t) {
      tris[3 * t] = (int32_t)i;
      tris[3 * t + 1] = fan[2 * (i * k + s)];
      tris[3 * t + 2] = fan[2 * (i * k + s) + 1];
    }
  /* lengths, mean, slack: block partials of 256 like the kernel (same rounding) */
  double sum = 0.0, slack = -INFINITY;
  for (int64_t b0 = 0; b0 < T; b0 += 256) {
    double s_sum[256], s_slk[256];
    for (int q = 0; q < 256; ++q) {
      s_sum[q] = 0.0;
      s_slk[q] = -INFINITY;
      const int64_t tt = b0 + q;
      if (tt >= T) continue;
      const int a = tris[3 * tt], b = tris[3 * tt + 1], c = tris[3 * tt + 2];
      const double la = orc_dist3(xyz, b, c), lb = orc_dist3(xyz, a, c), lc = orc_dist3(xyz, a, b);
      len[3 * tt] = la;
      len[3 * tt + 1] = lb;
      len[3 * tt + 2] = lc;
      s_sum[q] = (la + lb) + lc;
      const double s0 = la - lb - lc, s1 = lb - la - lc, s2 = lc - la - lb;
      double sl = s0 > s1 ? s0 : s1;
      sl = s2 > sl ? s2 : sl;
      s_slk[q] = sl;
    }
    for (int off = 128; off > 0; off >>= 1)
      for (int q = 0; q < off; ++q) {
        s_sum[q] += s_sum[q + off];
        if (s_slk[q + off] > s_slk[q]) s_slk[q] = s_slk[q + off];
      }
    sum += s_sum[0];
    if (s_slk[0] > slack) slack = s_slk[0];
  }
  const double mean = T > 0 ? sum / (3.0 * (double)T) : 0.0;
  double eps = slack + mean * moll;
  eps = eps > 0.0 ? eps : 0.0;
  /* ---- tufted cover ------------------------------------------------------------- */
  const int64_t F = 2 * T;
  int32_t* fv = (int32_t*)malloc((size_t)(3 * F + 1) * sizeof(int32_t));
  int32_t* fn = (int32_t*)malloc((size_t)(3 * F + 1) * sizeof(int32_t));
  double* fl = (double*)malloc((size_t)(3 * F + 1) * sizeof(double));
  int32_t* bstart = (int32_t*)calloc((size_t)n + 2, sizeof(int32_t));
  edge_rec_t* rec = (edge_rec_t*)malloc((size_t)(3 * T + 1) * sizeof(edge_rec_t));
  if (!fv || !fn || !fl || !bstart || !rec) return -1;
  for (int64_t tt = 0; tt < T; ++tt) {
    const int v0 = tris[3 * tt], v1 = tris[3 * tt + 1], v2 = tris[3 * tt + 2];
    const double la = len[3 * tt] + eps, lb = len[3 * tt + 1] + eps, lc = len[3 * tt + 2] + eps;
    const int64_t f = 2 * tt, g = f + 1;
    fv[3 * f] = v0; fv[3 * f + 1] = v1; fv[3 * f + 2] = v2;
    fl[3 * f] = lc; fl[3 * f + 1] = la; fl[3 * f + 2] = lb;
    fv[3 * g] = v0; fv[3 * g + 1] = v2; fv[3 * g + 2] = v1;
    fl[3 * g] = lb; fl[3 * g + 1] = la; fl[3 * g + 2] = lc;
    bstart[(v0 < v1 ? v0 : v1) + 1]++;
    bstart[(v1 < v2 ? v1 : v2) + 1]++;
    bstart[(v2 < v0 ? v2 : v0) + 1]++;
  }
  for (int64_t i = 0; i < n; ++i) bstart[i + 1] += bstart[i];
  {
    int32_t* bcur = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
    if (!bcur) return -1;
    for (int64_t tt = 0; tt < T; ++tt) {
      const int v[3] = {tris[3 * tt], tris[3 * tt + 1], tris[3 * tt + 2]};
      for (int e = 0; e < 3; ++e) {
        const int u = v[e], w = v[nx3(e)];
        const int mn = u < w ? u : w, mx = u < w ? w : u;
        const int slot = bstart[mn] + bcur[mn]++;
        rec[slot].mx = mx;
        rec[slot].code = (int32_t)(2 * (3 * tt + e) + (u == mn ? 1 : 0));
      }
    }
    free(bcur);
  }
  for (int64_t i = 0; i < n; ++i) {
    const int b0 = bstart[i], e0 = bstart[i + 1];
    qsort(rec + b0, (size_t)(e0 - b0), sizeof(edge_rec_t), edge_rec_cmp);
    int a = b0;
    while (a < e0) {
      int z = a;
      while (z < e0 && rec[z].mx == rec[a].mx) ++z;
      const int m = z - a;
      for (int p = 0; p < m; ++p) {
        const int h1 = cover_halfedge(rec[a + p].code, 1);
        const int h2 = cover_halfedge(rec[a + (p + 1 == m ? 0 : p + 1)].code, 0);
        fn[h1] = h2;
        fn[h2] = h1;
      }
      a = z;
    }
  }
  /* ---- flip to intrinsic Delaunay: sweep until a whole pass flips nothing --------- */
  for (int pass = 0; pass < 100000; ++pass) {
    int64_t flipped = 0;
    for (int64_t h = 0; h < 3 * F; ++h) flipped += try_flip((int)h, fv, fl, fn);
    if (!flipped) break;
  }
  /* ---- weights of the cover ------------------------------------------------------- */
  free(area);
  area = (double*)malloc((size_t)(F + 1) * sizeof(double));
  if (!area) return -1;
  memset(rstart, 0, ((size_t)n + 1) * sizeof(int32_t));
  for (int64_t f = 0; f < F; ++f)
    for (int c = 0; c < 3; ++c) rstart[fv[3 * f + c] + 1] += 2;
  for (int64_t i = 0; i < n; ++i) rstart[i + 1] += rstart[i];
  lap_ent_t* ent = (lap_ent_t*)malloc((size_t)(6 * F + 1) * sizeof(lap_ent_t));
  int32_t* cur = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  if (!ent || !cur) return -1;
  for (int64_t f = 0; f < F; ++f) {
    const int vtx[3] = {fv[3 * f], fv[3 * f + 1], fv[3 * f + 2]};
    const double l[3] = {fl[3 * f + 1], fl[3 * f + 2], fl[3 * f]};
    const double ar = heron_area(l[0], l[1], l[2]);
    area[f] = ar * 0.5;
    double wgt[3];
    for (int c = 0; c < 3; ++c) {
      const double lo = l[c], l1 = l[nx3(c)], l2 = l[pv3(c)];
      const double cot = ar > 0.0 ? ((l1 * l1 + l2 * l2) - lo * lo) / (4.0 * ar) : 0.0;
      wgt[c] = ((0.5 * cot) * 0.5) / 3.0;
    }
    for (int c = 0; c < 3; ++c) {
      const int uu = vtx[c], v1 = vtx[nx3(c)], v2 = vtx[pv3(c)];
      const int slot = rstart[uu] + cur[uu];
      cur[uu] += 2;
      ent[slot].col = v1;
      ent[slot].key = (int32_t)(4 * f + pv3(c));
      ent[slot].val = -wgt[pv3(c)];
      ent[slot + 1].col = v2;
      ent[slot + 1].key = (int32_t)(4 * f + nx3(c));
      ent[slot + 1].val = -wgt[nx3(c)];
    }
  }
  free(fv); free(fn); free(fl); free(bstart); free(rec);
  int32_t* indptr = (int32_t*)calloc((size_t)n + 1, sizeof(int32_t));
  if (!indptr) return -1;
  for (int64_t i = 0; i < n; ++i) {
    qsort(ent + rstart[i], (size_t)(rstart[i + 1] - rstart[i]), sizeof(lap_ent_t), lap_ent_cmp);
    int distinct = 0;
    for (int a = rstart[i]; a < rstart[i + 1]; ++a)
      if (ent[a].col != i && (a == rstart[i] || ent[a].col != ent[a - 1].col)) ++distinct;
    indptr[i + 1] = indptr[i] + distinct + 1;
  }
  const int64_t nnz = indptr[n];
  int32_t* indices = (int32_t*)malloc((size_t)(nnz + 1) * sizeof(int32_t));
  double* vals = (double*)malloc((size_t)(nnz + 1) * sizeof(double));
  if (!indices || !vals) return -1;
  for (int64_t i = 0; i < n; ++i) {
    const int b = rstart[i], e = rstart[i + 1];
    double m = 0.0;
    for (int a = b; a < e; ++a) m += area[ent[a].key >> 2] * 0.5;
    mass[i] = (m / 3.0) / 3.0;
    int w = indptr[i], diag_pos = -1, diag_done = 0;
    double diag = 0.0;
    int a = b;
    while (a < e) {
      const int col = ent[a].col;
      double s = 0.0;
      while (a < e && ent[a].col == col) s += ent[a++].val;
      if (col == i) continue;
      if (!diag_done && col > i) {
        diag_pos = w++;
        diag_done = 1;
      }
      indices[w] = col;
      vals[w] = s;
      ++w;
      diag -= s;
    }
    if (!diag_done) diag_pos = w++;
    indices[diag_pos] = (int32_t)i;
    vals[diag_pos] = diag;
  }
  *nnz_out = nnz;
  *indptr_out = indptr;
  *indices_out = indices;
  *vals_out = vals;
  free(nbr); free(d2); free(fan); free(fcnt); free(tris); free(len); free(area);
  free(rstart); free(ent); free(cur);
  return 0;
}

void orc_free(void* p) { free(p); }

/* --------------------------------------------------------------------------
 * orc_point_mesh_distance — unsigned distance from query points to a triangle mesh and
 * the closest triangle (lowest index on ties), standing in for Open3D's
 * RaycastingScene.compute_distance behind `mri` (pyQSM/viz/ray_casting.py:237-260;
 * Open3D itself is not installable here: PARITY UNPINNED, pinned to analytic cases in
 * tests/). Ericson's closest-point-on-triangle region walk in fp32, products rounded
 * separately, in the operation order of pyqsm_amd/csrc/meshdist.hip. */
static float orc_dot3f(float ax, float ay, float az, float bx, float by, float bz) {
  float d = ax * bx;
  d = d + ay * by;
  d = d + az * bz;
  return d;
}

int orc_point_mesh_distance(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                            const float* qry, int64_t Q, float* dist, uint32_t* prim) {
  (void)V;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < Q; ++i) {
    const float px = qry[3 * i], py = qry[3 * i + 1], pz = qry[3 * i + 2];
    float best = INFINITY;
    uint32_t bp = 0xFFFFFFFFu;
    for (int64_t j = 0; j < T; ++j) {
      const float* a = verts + 3 * (int64_t)tris[3 * j];
      const float* b = verts + 3 * (int64_t)tris[3 * j + 1];
      const float* c = verts + 3 * (int64_t)tris[3 * j + 2];
      const float ax = a[0], ay = a[1], az = a[2];
      const float abx = b[0] - ax, aby = b[1] - ay, abz = b[2] - az;
      const float acx = c[0] - ax, acy = c[1] - ay, acz = c[2] - az;
      const float apx = px - ax, apy = py - ay, apz = pz - az;
      const float d1 = orc_dot3f(abx, aby, abz, apx, apy, apz);
      const float d2 = orc_dot3f(acx, acy, acz, apx, apy, apz);
      float cx, cy, cz;
      const float bpx = apx - abx, bpy = apy - aby, bpz = apz - abz;
      const float d3 = orc_dot3f(abx, aby, abz, bpx, bpy, bpz);
      const float d4 = orc_dot3f(acx, acy, acz, bpx, bpy, bpz);
      const float cpx = apx - acx, cpy = apy - acy, cpz = apz - acz;
      const float d5 = orc_dot3f(abx, aby, abz, cpx, cpy, cpz);
      const float d6 = orc_dot3f(acx, acy, acz, cpx, cpy, cpz);
      const float vc = d1 * d4 - d3 * d2;
      const float vb = d5 * d2 - d1 * d6;
      const float va = d3 * d6 - d5 * d4;
      if (d1 <= 0.f && d2 <= 0.f) {
        cx = cy = cz = 0.f;
      } else if (d3 >= 0.f && d4 <= d3) {
        cx = abx; cy = aby; cz = abz;
      } else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        cx = v * abx; cy = v * aby; cz = v * abz;
      } else if (d6 >= 0.f && d5 <= d6) {
        cx = acx; cy = acy; cz = acz;
      } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float w = d2 / (d2 - d6);
        cx = w * acx; cy = w * acy; cz = w * acz;
      } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        cx = abx + w * (acx - abx);
        cy = aby + w * (acy - aby);
        cz = abz + w * (acz - abz);
      } else {
        const float denom = 1.f / (va + vb + vc);
        const float v = vb * denom, w = vc * denom;
        cx = abx * v + acx * w;
        cy = aby * v + acy * w;
        cz = abz * v + acz * w;
      }
      const float ex = apx - cx, ey = apy - cy, ez = apz - cz;
      const float d = orc_dot3f(ex, ey, ez, ex, ey, ez);
      if (d < best) {
        best = d;
        bp = (uint32_t)j;
      }
    }
    dist[i] = sqrtf(best);
    prim[i] = bp;
  }
  return 0;
}
