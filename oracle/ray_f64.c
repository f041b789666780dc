/*
 * ray_f64.c — an INDEPENDENT closest-hit evaluation of the ray sweep. TEST INFRASTRUCTURE: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 *
 * orc_cast_rays (pyqsm_oracle.c) restates the HIP kernels' own operation sequence (fp32
 * Moller-Trumbore with precomputed edges) so that the two agree bit for bit; that proves the
 * restatements agree, not that either is right. This file evaluates the same geometric question
 * (pyQSM/viz/ray_casting.py:172-180,279-289: closest hit, t in units of |d|, hit point =
 * (1-u-v) v0 + u v1 + v v2) in double precision and in a different formulation:
 *
 *   translate the triangle to the ray origin, A = v0 - o, B = v1 - o, C = v2 - o;
 *   signed volumes (scalar triple products) w0 = d.(B x C), w1 = d.(C x A), w2 = d.(A x B);
 *   the line pierces the triangle iff w0, w1, w2 have one sign (zeros allowed, not all zero);
 *   barycentric weights b_i = w_i / (w0 + w1 + w2);   t = (b0 A + b1 B + b2 C).d / d.d
 *
 * — no edge vectors, no determinant of (d, e1, e2), no division before a hit is established.
 * fp32 inputs convert to double exactly. PARITY UNPINNED against Embree (Open3D is not
 * installable here); what it provides is the evidence for north_star's "hit distances within
 * 1e-5 rel" that a bit-identical mirror cannot give.
 */
#include <math.h>
#include <stdint.h>

static inline void cross3(const double a[3], const double b[3], double out[3]) {
  out[0] = a[1] * b[2] - a[2] * b[1];
  out[1] = a[2] * b[0] - a[0] * b[2];
  out[2] = a[0] * b[1] - a[1] * b[0];
}

static inline double dot3(const double a[3], const double b[3]) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}

/* One ray against one triangle. Returns 1 when the LINE pierces the triangle; t (may be <= 0)
 * and the weights of v0, v1, v2 are written in any case where the volumes do not vanish
 * together (bary = NaN, t = NaN otherwise). */
static int ray_tri_f64(const float* o, const float* d, const float* v0, const float* v1,
                       const float* v2, double* t, double bary[3]) {
  double A[3], B[3], C[3], D[3], x[3];
  for (int a = 0; a < 3; ++a) {
    A[a] = (double)v0[a] - (double)o[a];
    B[a] = (double)v1[a] - (double)o[a];
    C[a] = (double)v2[a] - (double)o[a];
    D[a] = (double)d[a];
  }
  cross3(B, C, x);
  const double w0 = dot3(D, x);
  cross3(C, A, x);
  const double w1 = dot3(D, x);
  cross3(A, B, x);
  const double w2 = dot3(D, x);
  const double det = w0 + w1 + w2;
  if (det == 0.0 || !isfinite(det)) {
    *t = NAN;
    bary[0] = bary[1] = bary[2] = NAN;
    return 0;
  }
  bary[0] = w0 / det;
  bary[1] = w1 / det;
  bary[2] = w2 / det;
  double h[3];
  for (int a = 0; a < 3; ++a) h[a] = bary[0] * A[a] + bary[1] * B[a] + bary[2] * C[a];
  *t = dot3(h, D) / dot3(D, D);
  const int pos = w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0;
  const int neg = w0 <= 0.0 && w1 <= 0.0 && w2 <= 0.0;
  return pos || neg;
}

/* Closest hit with t > 0 per ray; ties in t go to the lowest triangle index.
 *   t_hit f64 [R] (+inf on miss), prim i64 [R] (-1 on miss), bary f64 [R,3] (weights of v0, v1, v2
 *   of the winning triangle; NaN on miss), second f64 [R]: t of the runner-up (+inf if none),
 *   so that the caller can tell near-ties in depth from genuine differences. */
int orc_cast_rays_f64(const float* verts, const int32_t* tris, int64_t T, const float* rays,
                      int64_t R, double* t_hit, int64_t* prim, double* bary, double* second) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t r = 0; r < R; ++r) {
    const float* o = rays + 6 * r;
    const float* d = o + 3;
    double bt = INFINITY, b2 = INFINITY, bb[3] = {NAN, NAN, NAN};
    int64_t bp = -1;
    for (int64_t j = 0; j < T; ++j) {
      const int32_t* tv = tris + 3 * j;
      double t, w[3];
      if (!ray_tri_f64(o, d, verts + 3 * (int64_t)tv[0], verts + 3 * (int64_t)tv[1],
                       verts + 3 * (int64_t)tv[2], &t, w))
        continue;
      if (!(t > 0.0)) continue;
      if (t < bt) {
        b2 = bt;
        bt = t;
        bp = j;
        bb[0] = w[0];
        bb[1] = w[1];
        bb[2] = w[2];
      } else if (t < b2) {
        b2 = t;
      }
    }
    t_hit[r] = bt;
    prim[r] = bp;
    second[r] = b2;
    bary[3 * r] = bb[0];
    bary[3 * r + 1] = bb[1];
    bary[3 * r + 2] = bb[2];
  }
  return 0;
}

/* The same evaluation for given (ray, triangle) pairs: what does double precision say about
 * the triangle somebody else chose? pierces u8 [R], t f64 [R], bary f64 [R,3]; prim < 0 skips. */
int orc_ray_tri_pairs_f64(const float* verts, const int32_t* tris, const float* rays, int64_t R,
                          const int64_t* prim, uint8_t* pierces, double* t, double* bary) {
  for (int64_t r = 0; r < R; ++r) {
    pierces[r] = 0;
    t[r] = NAN;
    bary[3 * r] = bary[3 * r + 1] = bary[3 * r + 2] = NAN;
    if (prim[r] < 0) continue;
    const int32_t* tv = tris + 3 * prim[r];
    pierces[r] = (uint8_t)ray_tri_f64(rays + 6 * r, rays + 6 * r + 3, verts + 3 * (int64_t)tv[0],
                                      verts + 3 * (int64_t)tv[1], verts + 3 * (int64_t)tv[2], t + r,
                                      bary + 3 * r);
  }
  return 0;
}
