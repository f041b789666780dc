// raycast.hip — BVH-free ray x triangle sweep (Moller-Trumbore, closest hit,
// all hits, crossing counts) for gfx950.
//
// Replaces open3d RaycastingScene.cast_rays / list_intersections /
// compute_occupancy as used by pyQSM/viz/ray_casting.py:275-279,168,65-69.
//
// Data flow: the mesh is expanded once to 48-byte records (v0, e1, e2, pad) and
// every wave walks the whole record array in triangle order. The record index is
// wave-uniform, so the records arrive through the scalar cache as SGPR operands
// and never occupy VGPRs or LDS; each lane keeps RPL rays in registers and
// evaluates two rays per packed-f32 instruction. Nothing but the rays (read
// once) and the results (written once) touches HBM per ray; the 24 MB record
// stream is shared by every CU and is served from L2 / Infinity Cache.
//
// Arithmetic contract (shared with oracle/pyqsm_oracle.c, which restates it on
// the CPU so that t, u, v and the primitive id agree bit for bit):
//   p  = d x e2          px = fma(dy, e2z, -(dz*e2y)) ...
//   det = fma(e1x, px, fma(e1y, py, e1z*pz))
//   tv = o - v0
//   U  = fma(tvx, px, fma(tvy, py, tvz*pz))
//   q  = tv x e1         qx = fma(tvy, e1z, -(tvz*e1y)) ...
//   V  = fma(dx, qx, fma(dy, qy, dz*qz))
//   Tn = fma(e2x, qx, fma(e2y, qy, e2z*qz))
//   W  = det - (U + V)
//   hit <=> (det > 0 & U >= 0 & V >= 0 & W >= 0 & Tn > 0)
//         | (det < 0 & U <= 0 & V <= 0 & W <= 0 & Tn < 0)
//   t = Tn / det, u = U / det, v = V / det   (IEEE division)
//   closest hit: strictly smaller t wins, so ties go to the lowest triangle id.
#include "grid.hpp"

#include <cmath>

#include "raycast.hpp"

namespace pyqsm {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float s) { return f2{s, s}; }

struct alignas(16) TriRec {  // 48 bytes, 16-byte aligned
  float v0x, v0y, v0z, e1x;
  float e1y, e1z, e2x, e2y;
  float e2z, pad0, pad1, pad2;
};

// The nine live floats of a record, as separate scalars (kept in SGPRs when the
// record index is wave-uniform).
struct Tri {
  float v0x, v0y, v0z, e1x, e1y, e1z, e2x, e2y, e2z;
};

__device__ __forceinline__ Tri load_tri(const TriRec* __restrict__ tri, int j) {
  const float4* p = reinterpret_cast<const float4*>(tri + j);
  const float4 a = p[0], b = p[1], c = p[2];
  return Tri{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x};
}

// ---- mesh expansion ---------------------------------------------------------

__global__ void k_expand_tris(const float* __restrict__ verts, int64_t V,
                              const int32_t* __restrict__ tris, int64_t T,
                              TriRec* __restrict__ out, int* __restrict__ bad) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= T) return;
  int32_t a = tris[3 * i], b = tris[3 * i + 1], c = tris[3 * i + 2];
  if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) {
    *bad = 1;
    a = b = c = 0;
  }
  float ax = verts[3 * a], ay = verts[3 * a + 1], az = verts[3 * a + 2];
  TriRec r;
  r.v0x = ax;
  r.v0y = ay;
  r.v0z = az;
  r.e1x = verts[3 * b] - ax;
  r.e1y = verts[3 * b + 1] - ay;
  r.e1z = verts[3 * b + 2] - az;
  r.e2x = verts[3 * c] - ax;
  r.e2y = verts[3 * c + 1] - ay;
  r.e2z = verts[3 * c + 2] - az;
  r.pad0 = r.pad1 = r.pad2 = 0.f;
  out[i] = r;
}

// ---- the sweep ----------------------------------------------------------------

struct RayPair {  // two rays, one per packed half
  f2 ox, oy, oz, dx, dy, dz;
};

// Numerators of one triangle against one ray pair.
struct Num {
  f2 det, U, V, Tn;
};

__device__ __forceinline__ void mt_front(const RayPair& r, const Tri t, f2& px, f2& py,
                                         f2& pz, f2& det, f2& tvx, f2& tvy, f2& tvz, f2& U) {
  px = fma2(r.dy, splat(t.e2z), -(r.dz * splat(t.e2y)));
  py = fma2(r.dz, splat(t.e2x), -(r.dx * splat(t.e2z)));
  pz = fma2(r.dx, splat(t.e2y), -(r.dy * splat(t.e2x)));
  det = fma2(splat(t.e1x), px, fma2(splat(t.e1y), py, splat(t.e1z) * pz));
  tvx = r.ox - splat(t.v0x);
  tvy = r.oy - splat(t.v0y);
  tvz = r.oz - splat(t.v0z);
  U = fma2(tvx, px, fma2(tvy, py, tvz * pz));
}

__device__ __forceinline__ void mt_back(const RayPair& r, const Tri t, f2 tvx, f2 tvy,
                                        f2 tvz, f2& V, f2& Tn) {
  f2 qx = fma2(tvy, splat(t.e1z), -(tvz * splat(t.e1y)));
  f2 qy = fma2(tvz, splat(t.e1x), -(tvx * splat(t.e1z)));
  f2 qz = fma2(tvx, splat(t.e1y), -(tvy * splat(t.e1x)));
  V = fma2(r.dx, qx, fma2(r.dy, qy, r.dz * qz));
  Tn = fma2(splat(t.e2x), qx, fma2(splat(t.e2y), qy, splat(t.e2z) * qz));
}

// Can this (det, U) still be a hit? A hit needs 0 <= U/det <= 1 (W >= 0 with
// V >= 0 gives |U| <= |det|; fl(U + V) is monotone so this survives rounding),
// i.e. |2U - det| <= |det|. One fma and one compare per ray; false positives at
// rounding edges are harmless because is_hit() makes the exact decision.
__device__ __forceinline__ f2 alive_key(f2 det, f2 U) { return fma2(splat(2.f), U, -det); }
__device__ __forceinline__ bool u_alive(float key, float det) {
  return __builtin_fabsf(key) <= __builtin_fabsf(det);
}

// Same decision with the sign of det supplied by the caller (for the parallel-ray
// kernels det is a per-triangle scalar, and its sign is taken from the integer bits
// on the scalar unit: det > 0 <=> bits > 0, det < 0 <=> bits < 0 and bits != -0).
__device__ __forceinline__ bool is_hit_signed(bool det_pos, bool det_neg, float det, float U,
                                              float V, float Tn) {
  float W = det - (U + V);
  bool pos = det_pos && U >= 0.f && V >= 0.f && W >= 0.f && Tn > 0.f;
  bool neg = det_neg && U <= 0.f && V <= 0.f && W <= 0.f && Tn < 0.f;
  return pos || neg;
}

__device__ __forceinline__ bool is_hit(float det, float U, float V, float Tn) {
  float W = det - (U + V);
  bool pos = det > 0.f && U >= 0.f && V >= 0.f && W >= 0.f && Tn > 0.f;
  bool neg = det < 0.f && U <= 0.f && V <= 0.f && W <= 0.f && Tn < 0.f;
  return pos || neg;
}

// NP = ray pairs per lane (each lane owns 2*NP rays).
template <int NP>
__global__ __launch_bounds__(256) void k_cast_rays(const TriRec* __restrict__ tri, int T,
                                                   const float* __restrict__ rays, int64_t R,
                                                   float* __restrict__ t_hit,
                                                   uint32_t* __restrict__ prim_id,
                                                   float* __restrict__ uv) {
  constexpr int RPL = 2 * NP;
  const int64_t block_base = int64_t(blockIdx.x) * (256 * RPL);
  RayPair rp[NP];
  float best_t[RPL];
  uint32_t best_p[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 256 + threadIdx.x;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (r < R) {
      const float* p = rays + 6 * r;
      o0 = p[0]; o1 = p[1]; o2 = p[2]; d0 = p[3]; d1 = p[4]; d2 = p[5];
    }
    rp[k >> 1].ox[k & 1] = o0; rp[k >> 1].oy[k & 1] = o1; rp[k >> 1].oz[k & 1] = o2;
    rp[k >> 1].dx[k & 1] = d0; rp[k >> 1].dy[k & 1] = d1; rp[k >> 1].dz[k & 1] = d2;
    best_t[k] = __builtin_inff();
    best_p[k] = PYQSM_MISS_PRIM;
  }

  // One triangle against all ray pairs of this lane.
  auto sweep = [&](const Tri t, const int j) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f2 px, py, pz, det, tvx, tvy, tvz, U;
      mt_front(rp[q], t, px, py, pz, det, tvx, tvy, tvz, U);
      const f2 key = alive_key(det, U);
      bool alive = u_alive(key[0], det[0]) || u_alive(key[1], det[1]);
      if (__builtin_amdgcn_ballot_w64(alive) != 0) {  // wave-uniform branch
        f2 V, Tn;
        mt_back(rp[q], t, tvx, tvy, tvz, V, Tn);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (is_hit(det[h], U[h], V[h], Tn[h])) {
            float tt = Tn[h] / det[h];
            if (tt < best_t[2 * q + h]) {
              best_t[2 * q + h] = tt;
              best_p[2 * q + h] = uint32_t(j);
            }
          }
        }
      }
    }
  };
  // Wave-uniform addresses: records arrive by scalar loads. Two register sets
  // alternate so that the next record is in flight while this one is consumed
  // and no register copies are needed. T >= 1 is checked by the launcher.
  Tri ta = load_tri(tri, 0);
  int j = 0;
  for (; j + 1 < T; j += 2) {
    const Tri tb = load_tri(tri, j + 1);
    sweep(ta, j);
    ta = load_tri(tri, j + 2 < T ? j + 2 : j + 1);
    sweep(tb, j + 1);
  }
  if (j < T) sweep(ta, j);

#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 256 + threadIdx.x;
    if (r >= R) continue;
    t_hit[r] = best_t[k];
    prim_id[r] = best_p[k];
    if (uv) {
      float u = 0.f, v = 0.f;
      if (best_p[k] != PYQSM_MISS_PRIM) {
        // Re-evaluate the winner with the same operation sequence.
        const Tri t = load_tri(tri, int(best_p[k]));
        RayPair one;
        const int q = k >> 1, h = k & 1;
        one.ox = splat(rp[q].ox[h]); one.oy = splat(rp[q].oy[h]); one.oz = splat(rp[q].oz[h]);
        one.dx = splat(rp[q].dx[h]); one.dy = splat(rp[q].dy[h]); one.dz = splat(rp[q].dz[h]);
        f2 px, py, pz, det, tvx, tvy, tvz, U, V, Tn;
        mt_front(one, t, px, py, pz, det, tvx, tvy, tvz, U);
        mt_back(one, t, tvx, tvy, tvz, V, Tn);
        u = U[0] / det[0];
        v = V[0] / det[0];
      }
      uv[2 * r] = u;
      uv[2 * r + 1] = v;
    }
  }
}

// ---- parallel rays (one direction for the whole batch) ---------------------------
// Sun-angle sweeps cast millions of rays with ONE direction d. Then p = d x e2 and
// det = e1 . p depend on the triangle only and are computed once per triangle by
// k_dir_records with exactly the operations of mt_front, so the per-test work of
// the front half drops from 16 to 7 packed instructions and every result stays
// bit-identical to the general kernel.

struct alignas(16) DirRec {  // 32 bytes: what the front half needs
  float v0x, v0y, v0z, px;
  float py, pz, det, pad;
};

struct Front {
  float v0x, v0y, v0z, px, py, pz, det;
};

__device__ __forceinline__ Front load_front(const DirRec* __restrict__ rec, int j) {
  const float4* p = reinterpret_cast<const float4*>(rec + j);
  const float4 a = p[0], b = p[1];
  return Front{a.x, a.y, a.z, a.w, b.x, b.y, b.z};
}

__global__ void k_dir_records(const TriRec* __restrict__ tri, int64_t T,
                              const float* __restrict__ rays, DirRec* __restrict__ out) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= T) return;
  const Tri t = load_tri(tri, int(i));
  RayPair r;
  r.ox = r.oy = r.oz = splat(0.f);
  r.dx = splat(rays[3]);
  r.dy = splat(rays[4]);
  r.dz = splat(rays[5]);
  f2 px, py, pz, det, tvx, tvy, tvz, U;
  mt_front(r, t, px, py, pz, det, tvx, tvy, tvz, U);
  out[i] = DirRec{t.v0x, t.v0y, t.v0z, px[0], py[0], pz[0], det[0], 0.f};
}

// flag bit 0: some ray's direction differs (bitwise) from ray 0's; bit 1: some origin does
__global__ void k_check_uniform(const float* __restrict__ rays, int64_t R,
                                int* __restrict__ flag) {
  const unsigned int* u = reinterpret_cast<const unsigned int*>(rays);
  const unsigned int o0 = u[0], o1 = u[1], o2 = u[2], d0 = u[3], d1 = u[4], d2 = u[5];
  bool ddiff = false, odiff = false;
  for (int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; r < R;
       r += int64_t(gridDim.x) * blockDim.x) {
    ddiff |= u[6 * r + 3] != d0 || u[6 * r + 4] != d1 || u[6 * r + 5] != d2;
    odiff |= u[6 * r] != o0 || u[6 * r + 1] != o1 || u[6 * r + 2] != o2;
  }
  const int bits = (__builtin_amdgcn_ballot_w64(ddiff) != 0 ? 1 : 0) |
                   (__builtin_amdgcn_ballot_w64(odiff) != 0 ? 2 : 0);
  if (bits && (threadIdx.x & 63) == 0) atomicOr(flag, bits);
}

template <int NP>
__global__ __launch_bounds__(256) void k_cast_parallel(const TriRec* __restrict__ tri,
                                                       const DirRec* __restrict__ rec, int T,
                                                       const float* __restrict__ rays, int64_t R,
                                                       float* __restrict__ t_hit,
                                                       uint32_t* __restrict__ prim_id,
                                                       float* __restrict__ uv) {
  constexpr int RPL = 2 * NP;
  const int64_t block_base = int64_t(blockIdx.x) * (256 * RPL);
  f2 ox[NP], oy[NP], oz[NP];
  float best_t[RPL];
  uint32_t best_p[RPL];
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 256 + threadIdx.x;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (r < R) {
      const float* p = rays + 6 * r;
      o0 = p[0]; o1 = p[1]; o2 = p[2];
    }
    ox[k >> 1][k & 1] = o0; oy[k >> 1][k & 1] = o1; oz[k >> 1][k & 1] = o2;
    best_t[k] = __builtin_inff();
    best_p[k] = PYQSM_MISS_PRIM;
  }
  const float d0 = rays[3], d1 = rays[4], d2 = rays[5];  // the common direction

  auto sweep = [&](const Front f, const int j) {
    const int dbits = __builtin_bit_cast(int, f.det);
    const bool det_pos = dbits > 0, det_neg = dbits < 0 && dbits != int(0x80000000u);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f2 tvx = ox[q] - splat(f.v0x), tvy = oy[q] - splat(f.v0y), tvz = oz[q] - splat(f.v0z);
      const f2 U = fma2(tvx, splat(f.px), fma2(tvy, splat(f.py), tvz * splat(f.pz)));
      const f2 det = splat(f.det);
      const f2 key = alive_key(det, U);
      bool alive = u_alive(key[0], f.det) || u_alive(key[1], f.det);
      if (__builtin_amdgcn_ballot_w64(alive) != 0) {  // rare: fetch the edges and finish
        const Tri t = load_tri(tri, j);
        RayPair r;
        r.dx = splat(d0); r.dy = splat(d1); r.dz = splat(d2);
        f2 V, Tn;
        mt_back(r, t, tvx, tvy, tvz, V, Tn);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (is_hit_signed(det_pos, det_neg, f.det, U[h], V[h], Tn[h])) {
            float tt = Tn[h] / f.det;
            if (tt < best_t[2 * q + h]) {
              best_t[2 * q + h] = tt;
              best_p[2 * q + h] = uint32_t(j);
            }
          }
        }
      }
    }
  };
  Front fa = load_front(rec, 0);
  int j = 0;
  for (; j + 1 < T; j += 2) {
    const Front fb = load_front(rec, j + 1);
    sweep(fa, j);
    fa = load_front(rec, j + 2 < T ? j + 2 : j + 1);
    sweep(fb, j + 1);
  }
  if (j < T) sweep(fa, j);

#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 256 + threadIdx.x;
    if (r >= R) continue;
    t_hit[r] = best_t[k];
    prim_id[r] = best_p[k];
    if (uv) {
      float u = 0.f, v = 0.f;
      if (best_p[k] != PYQSM_MISS_PRIM) {
        const Tri t = load_tri(tri, int(best_p[k]));
        RayPair one;
        const int q = k >> 1, h = k & 1;
        one.ox = splat(ox[q][h]); one.oy = splat(oy[q][h]); one.oz = splat(oz[q][h]);
        one.dx = splat(d0); one.dy = splat(d1); one.dz = splat(d2);
        f2 px, py, pz, det, tvx, tvy, tvz, U, V, Tn;
        mt_front(one, t, px, py, pz, det, tvx, tvy, tvz, U);
        mt_back(one, t, tvx, tvy, tvz, V, Tn);
        u = U[0] / det[0];
        v = V[0] / det[0];
      }
      uv[2 * r] = u;
      uv[2 * r + 1] = v;
    }
  }
}

// ---- parallel rays with conservative cluster culling ---------------------------------
// Still no hierarchy: the triangles are sorted once per call by the cell of their
// centroid on the plane normal to the common direction (the counting sort of
// grid.hip), cut into clusters of kCluster consecutive triangles, and every cluster
// carries the bounding rectangle of its members on that plane. A wave computes the
// rectangle of its own rays' origins on the same plane once, and walks the cluster
// list with one scalar rectangle test per cluster; only overlapping clusters run the
// Moller-Trumbore code of k_cast_parallel, unchanged. Rectangles are grown by a margin
// four orders of magnitude above fp32 rounding, so no test that could hit is skipped
// and t, u, v and the primitive id stay bit-identical to the brute-force sweep (the
// closest-hit rule compares (t, original triangle id), so the processing order does
// not matter).

static constexpr int kCluster = 16;

struct Basis {
  double ux, uy, uz, vx, vy, vz;
};

// per triangle: centroid on the plane (for sorting) and bounding rectangle
__global__ __launch_bounds__(256) void k_tri_uv(const TriRec* __restrict__ tri, int64_t T, Basis b,
                                                double* __restrict__ cen /*[T][3]*/,
                                                float4* __restrict__ rect /*[T]*/) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= T) return;
  const Tri t = load_tri(tri, int(i));
  const double px[3] = {double(t.v0x), double(t.v0x) + double(t.e1x), double(t.v0x) + double(t.e2x)};
  const double py[3] = {double(t.v0y), double(t.v0y) + double(t.e1y), double(t.v0y) + double(t.e2y)};
  const double pz[3] = {double(t.v0z), double(t.v0z) + double(t.e1z), double(t.v0z) + double(t.e2z)};
  double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300, uc = 0.0, vc = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double u = px[k] * b.ux + py[k] * b.uy + pz[k] * b.uz;
    const double v = px[k] * b.vx + py[k] * b.vy + pz[k] * b.vz;
    u0 = u < u0 ? u : u0;
    u1 = u > u1 ? u : u1;
    v0 = v < v0 ? v : v0;
    v1 = v > v1 ? v : v1;
    uc += u;
    vc += v;
  }
  cen[3 * i] = uc / 3.0;
  cen[3 * i + 1] = vc / 3.0;
  cen[3 * i + 2] = 0.0;
  rect[i] = make_float4(__double2float_rd(u0), __double2float_ru(u1), __double2float_rd(v0),
                        __double2float_ru(v1));
}

// sorted front records, original ids, and one grown rectangle per cluster
__global__ __launch_bounds__(256) void k_clusters(const TriRec* __restrict__ tri, int T,
                                                  const float* __restrict__ rays,
                                                  const int32_t* __restrict__ order,
                                                  const float4* __restrict__ rect, float margin,
                                                  DirRec* __restrict__ srec,
                                                  int32_t* __restrict__ sid,
                                                  float4* __restrict__ crect) {
  const int cl = blockIdx.x * 256 + threadIdx.x;
  const int nclus = (T + kCluster - 1) / kCluster;
  if (cl >= nclus) return;
  RayPair r;
  r.ox = r.oy = r.oz = splat(0.f);
  r.dx = splat(rays[3]);
  r.dy = splat(rays[4]);
  r.dz = splat(rays[5]);
  float u0 = __builtin_inff(), u1 = -__builtin_inff(), v0 = __builtin_inff(), v1 = -__builtin_inff();
  for (int s = cl * kCluster; s < (cl + 1) * kCluster && s < T; ++s) {
    const int o = order[s];
    const Tri t = load_tri(tri, o);
    f2 px, py, pz, det, tvx, tvy, tvz, U;
    mt_front(r, t, px, py, pz, det, tvx, tvy, tvz, U);
    srec[s] = DirRec{t.v0x, t.v0y, t.v0z, px[0], py[0], pz[0], det[0], 0.f};
    sid[s] = o;
    const float4 q = rect[o];
    u0 = fminf(u0, q.x);
    u1 = fmaxf(u1, q.y);
    v0 = fminf(v0, q.z);
    v1 = fmaxf(v1, q.w);
  }
  crect[cl] = make_float4(u0 - margin, u1 + margin, v0 - margin, v1 + margin);
}

static constexpr int kSuper = 32;  // clusters per super-cluster (second level of rectangles)

__global__ __launch_bounds__(256) void k_super_rects(int nclus, const float4* __restrict__ crect,
                                                     float4* __restrict__ srect) {
  const int sc = blockIdx.x * 256 + threadIdx.x;
  const int nsup = (nclus + kSuper - 1) / kSuper;
  if (sc >= nsup) return;
  float u0 = __builtin_inff(), u1 = -__builtin_inff(), v0 = __builtin_inff(), v1 = -__builtin_inff();
  for (int cl = sc * kSuper; cl < (sc + 1) * kSuper && cl < nclus; ++cl) {
    const float4 q = crect[cl];
    u0 = fminf(u0, q.x);
    u1 = fmaxf(u1, q.y);
    v0 = fminf(v0, q.z);
    v1 = fmaxf(v1, q.w);
  }
  srect[sc] = make_float4(u0, u1, v0, v1);
}

__device__ __forceinline__ float wave_min_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

template <int NP>
__global__ __launch_bounds__(256) void k_cast_parallel_culled(
    const TriRec* __restrict__ tri, const DirRec* __restrict__ srec,
    const int32_t* __restrict__ sid, const float4* __restrict__ crect,
    const float4* __restrict__ srect, int T, Basis b,
    const float* __restrict__ rays, int64_t R, float* __restrict__ t_hit,
    uint32_t* __restrict__ prim_id, float* __restrict__ uv) {
  constexpr int RPL = 2 * NP;
  // a wave owns 64*RPL CONSECUTIVE rays, so that its rectangle on the plane is small
  const int64_t block_base =
      int64_t(blockIdx.x) * (256 * RPL) + int64_t(threadIdx.x >> 6) * (64 * RPL) + (threadIdx.x & 63);
  f2 ox[NP], oy[NP], oz[NP];
  float best_t[RPL];
  uint32_t best_p[RPL];
  double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 64;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (r < R) {
      const float* p = rays + 6 * r;
      o0 = p[0]; o1 = p[1]; o2 = p[2];
      const double u = double(o0) * b.ux + double(o1) * b.uy + double(o2) * b.uz;
      const double v = double(o0) * b.vx + double(o1) * b.vy + double(o2) * b.vz;
      umin = u < umin ? u : umin;
      umax = u > umax ? u : umax;
      vmin = v < vmin ? v : vmin;
      vmax = v > vmax ? v : vmax;
    }
    ox[k >> 1][k & 1] = o0; oy[k >> 1][k & 1] = o1; oz[k >> 1][k & 1] = o2;
    best_t[k] = __builtin_inff();
    best_p[k] = PYQSM_MISS_PRIM;
  }
  // rectangle of this wave's ray origins on the plane (wave-uniform)
  const float ru0 = wave_min_f(__double2float_rd(umin)), ru1 = wave_max_f(__double2float_ru(umax));
  const float rv0 = wave_min_f(__double2float_rd(vmin)), rv1 = wave_max_f(__double2float_ru(vmax));
  const float d0 = rays[3], d1 = rays[4], d2 = rays[5];

  auto sweep = [&](const Front f, const int s) {
    const int dbits = __builtin_bit_cast(int, f.det);
    const bool det_pos = dbits > 0, det_neg = dbits < 0 && dbits != int(0x80000000u);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const f2 tvx = ox[q] - splat(f.v0x), tvy = oy[q] - splat(f.v0y), tvz = oz[q] - splat(f.v0z);
      const f2 U = fma2(tvx, splat(f.px), fma2(tvy, splat(f.py), tvz * splat(f.pz)));
      const f2 det = splat(f.det);
      const f2 key = alive_key(det, U);
      bool alive = u_alive(key[0], f.det) || u_alive(key[1], f.det);
      if (__builtin_amdgcn_ballot_w64(alive) != 0) {
        const uint32_t id = uint32_t(sid[s]);
        const Tri t = load_tri(tri, int(id));
        RayPair r;
        r.dx = splat(d0); r.dy = splat(d1); r.dz = splat(d2);
        f2 V, Tn;
        mt_back(r, t, tvx, tvy, tvz, V, Tn);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (is_hit_signed(det_pos, det_neg, f.det, U[h], V[h], Tn[h])) {
            float tt = Tn[h] / f.det;
            // same winner as the in-order sweep: smallest t, then smallest triangle id
            if (tt < best_t[2 * q + h] || (tt == best_t[2 * q + h] && id < best_p[2 * q + h])) {
              best_t[2 * q + h] = tt;
              best_p[2 * q + h] = id;
            }
          }
        }
      }
    }
  };
  const int nclus = (T + kCluster - 1) / kCluster;
  const int nsup = (nclus + kSuper - 1) / kSuper;
  // the rectangles are uniform across the wave; make that explicit so that the tests
  // below are scalar compares of SGPR values
  const float su0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ru0)));
  const float su1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ru1)));
  const float sv0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rv0)));
  const float sv1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rv1)));
  float4 nxt = srect[0];
  for (int sc = 0; sc < nsup; ++sc) {
    const float4 sr = nxt;  // next super rectangle is in flight while this one is tested
    nxt = srect[sc + 1 < nsup ? sc + 1 : sc];
    if (su1 < sr.x || su0 > sr.y || sv1 < sr.z || sv0 > sr.w) continue;
    const int c1 = (sc + 1) * kSuper < nclus ? (sc + 1) * kSuper : nclus;
    for (int cl = sc * kSuper; cl < c1; ++cl) {
      const float4 cr = crect[cl];  // wave-uniform address
      if (su1 < cr.x || su0 > cr.y || sv1 < cr.z || sv0 > cr.w) continue;
      const int s1 = (cl + 1) * kCluster < T ? (cl + 1) * kCluster : T;
      for (int s = cl * kCluster; s < s1; ++s) sweep(load_front(srec, s), s);
    }
  }

#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 64;
    if (r >= R) continue;
    t_hit[r] = best_t[k];
    prim_id[r] = best_p[k];
    if (uv) {
      float u = 0.f, v = 0.f;
      if (best_p[k] != PYQSM_MISS_PRIM) {
        const Tri t = load_tri(tri, int(best_p[k]));
        RayPair one;
        const int q = k >> 1, h = k & 1;
        one.ox = splat(ox[q][h]); one.oy = splat(oy[q][h]); one.oz = splat(oz[q][h]);
        one.dx = splat(d0); one.dy = splat(d1); one.dz = splat(d2);
        f2 px, py, pz, det, tvx, tvy, tvz, U, V, Tn;
        mt_front(one, t, px, py, pz, det, tvx, tvy, tvz, U);
        mt_back(one, t, tvx, tvy, tvz, V, Tn);
        u = U[0] / det[0];
        v = V[0] / det[0];
      }
      uv[2 * r] = u;
      uv[2 * r + 1] = v;
    }
  }
}

// ---- common-origin batches (the pinhole camera of cast_rays) ----------------------------
// All rays start at one point O and look into one half space (every direction within
// ~75 degrees of their mean n). Central projection from O onto the plane n.x = 1 maps a ray
// to a point (u, v) and a triangle in front of O to a triangle, so the rectangle culling of
// the parallel case carries over with image coordinates in place of plane coordinates.
// Triangles that reach behind (or too close to) the eye plane get an unbounded rectangle:
// they are tested by every wave. The per-test arithmetic is the general kernel's, and the
// closest hit is decided by (t, original id): results are bit-identical to k_cast_rays.
struct Camera {
  double ox, oy, oz, ax, ay, az, bx, by, bz, nx, ny, nz;
};

// sum of the unit directions (3 doubles) -> mean viewing direction
__global__ __launch_bounds__(256) void k_dir_sum(const float* __restrict__ rays, int64_t R,
                                                 double* __restrict__ sum) {
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x; r < R;
       r += int64_t(gridDim.x) * blockDim.x) {
    const double x = rays[6 * r + 3], y = rays[6 * r + 4], z = rays[6 * r + 5];
    const double n = sqrt(x * x + y * y + z * z);
    if (n > 0.0) {
      sx += x / n;
      sy += y / n;
      sz += z / n;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sx += __shfl_down(sx, off, 64);
    sy += __shfl_down(sy, off, 64);
    sz += __shfl_down(sz, off, 64);
  }
  // one set of atomics per block
  __shared__ double ws[4][3];
  if ((threadIdx.x & 63) == 0) {
    ws[threadIdx.x >> 6][0] = sx;
    ws[threadIdx.x >> 6][1] = sy;
    ws[threadIdx.x >> 6][2] = sz;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(sum + threadIdx.x, (ws[0][threadIdx.x] + ws[1][threadIdx.x]) +
                                     (ws[2][threadIdx.x] + ws[3][threadIdx.x]));
}

// image coordinates of every ray (as [R][3] points for the bounding-box pass); flags rays
// that look too far sideways for the projection (or have a zero / non-finite direction)
__global__ __launch_bounds__(256) void k_ray_image(const float* __restrict__ rays, int64_t R,
                                                   Camera cam, double min_cos,
                                                   double* __restrict__ img,
                                                   int* __restrict__ bad) {
  int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (r >= R) return;
  const double x = rays[6 * r + 3], y = rays[6 * r + 4], z = rays[6 * r + 5];
  const double len = sqrt(x * x + y * y + z * z);
  const double w = x * cam.nx + y * cam.ny + z * cam.nz;
  double u = 0.0, v = 0.0;
  if (!(w > min_cos * len) || !(len > 0.0) || !(len < 1e300)) {
    *bad = 1;
  } else {
    u = (x * cam.ax + y * cam.ay + z * cam.az) / w;
    v = (x * cam.bx + y * cam.by + z * cam.bz) / w;
  }
  img[3 * r] = u;
  img[3 * r + 1] = v;
  img[3 * r + 2] = 0.0;
}

// per triangle: projected centroid (clamped into the rays' image box, for sorting) and
// projected bounding rectangle (unbounded when a vertex is not safely in front of O)
__global__ __launch_bounds__(256) void k_tri_image(const TriRec* __restrict__ tri, int64_t T,
                                                   Camera cam, double u_lo, double u_hi,
                                                   double v_lo, double v_hi,
                                                   double* __restrict__ cen /*[T][3]*/,
                                                   float4* __restrict__ rect /*[T]*/) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= T) return;
  const Tri t = load_tri(tri, int(i));
  const double px[3] = {double(t.v0x), double(t.v0x) + double(t.e1x), double(t.v0x) + double(t.e2x)};
  const double py[3] = {double(t.v0y), double(t.v0y) + double(t.e1y), double(t.v0y) + double(t.e2y)};
  const double pz[3] = {double(t.v0z), double(t.v0z) + double(t.e1z), double(t.v0z) + double(t.e2z)};
  double u0 = 1e300, u1 = -1e300, v0 = 1e300, v1 = -1e300, uc = 0.0, vc = 0.0;
  bool front = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double rx = px[k] - cam.ox, ry = py[k] - cam.oy, rz = pz[k] - cam.oz;
    const double x = rx * cam.ax + ry * cam.ay + rz * cam.az;
    const double y = rx * cam.bx + ry * cam.by + rz * cam.bz;
    const double w = rx * cam.nx + ry * cam.ny + rz * cam.nz;
    if (!(w > 1e-6 * (fabs(x) + fabs(y)) && w > 0.0)) {
      front = false;
    } else {
      const double u = x / w, v = y / w;
      u0 = u < u0 ? u : u0;
      u1 = u > u1 ? u : u1;
      v0 = v < v0 ? v : v0;
      v1 = v > v1 ? v : v1;
      uc += u;
      vc += v;
    }
  }
  if (front) {
    uc /= 3.0;
    vc /= 3.0;
    rect[i] = make_float4(__double2float_rd(u0), __double2float_ru(u1), __double2float_rd(v0),
                          __double2float_ru(v1));
  } else {
    uc = vc = 0.0;
    rect[i] = make_float4(-__builtin_inff(), __builtin_inff(), -__builtin_inff(), __builtin_inff());
  }
  cen[3 * i] = uc < u_lo ? u_lo : (uc > u_hi ? u_hi : uc);
  cen[3 * i + 1] = vc < v_lo ? v_lo : (vc > v_hi ? v_hi : vc);
  cen[3 * i + 2] = 0.0;
}

// sorted original ids and one grown rectangle per cluster
__global__ __launch_bounds__(256) void k_clusters_image(int T, const int32_t* __restrict__ order,
                                                        const float4* __restrict__ rect,
                                                        float margin, int32_t* __restrict__ sid,
                                                        float4* __restrict__ crect) {
  const int cl = blockIdx.x * 256 + threadIdx.x;
  const int nclus = (T + kCluster - 1) / kCluster;
  if (cl >= nclus) return;
  float u0 = __builtin_inff(), u1 = -__builtin_inff(), v0 = __builtin_inff(), v1 = -__builtin_inff();
  for (int s = cl * kCluster; s < (cl + 1) * kCluster && s < T; ++s) {
    const int o = order[s];
    sid[s] = o;
    const float4 q = rect[o];
    u0 = fminf(u0, q.x);
    u1 = fmaxf(u1, q.y);
    v0 = fminf(v0, q.z);
    v1 = fmaxf(v1, q.w);
  }
  crect[cl] = make_float4(u0 - margin, u1 + margin, v0 - margin, v1 + margin);
}

template <int NP>
__global__ __launch_bounds__(256) void k_cast_pinhole_culled(
    const TriRec* __restrict__ tri, const int32_t* __restrict__ sid,
    const float4* __restrict__ crect, const float4* __restrict__ srect, int T, Camera cam,
    const float* __restrict__ rays, int64_t R, float* __restrict__ t_hit,
    uint32_t* __restrict__ prim_id, float* __restrict__ uv) {
  constexpr int RPL = 2 * NP;
  // a wave owns 64*RPL CONSECUTIVE rays (a piece of an image row): a small image rectangle
  const int64_t block_base =
      int64_t(blockIdx.x) * (256 * RPL) + int64_t(threadIdx.x >> 6) * (64 * RPL) + (threadIdx.x & 63);
  RayPair rp[NP];
  float best_t[RPL];
  uint32_t best_p[RPL];
  double umin = 1e300, umax = -1e300, vmin = 1e300, vmax = -1e300;
#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 64;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
    if (r < R) {
      const float* p = rays + 6 * r;
      o0 = p[0]; o1 = p[1]; o2 = p[2]; d0 = p[3]; d1 = p[4]; d2 = p[5];
      const double w = double(d0) * cam.nx + double(d1) * cam.ny + double(d2) * cam.nz;
      const double u = (double(d0) * cam.ax + double(d1) * cam.ay + double(d2) * cam.az) / w;
      const double v = (double(d0) * cam.bx + double(d1) * cam.by + double(d2) * cam.bz) / w;
      umin = u < umin ? u : umin;
      umax = u > umax ? u : umax;
      vmin = v < vmin ? v : vmin;
      vmax = v > vmax ? v : vmax;
    }
    rp[k >> 1].ox[k & 1] = o0; rp[k >> 1].oy[k & 1] = o1; rp[k >> 1].oz[k & 1] = o2;
    rp[k >> 1].dx[k & 1] = d0; rp[k >> 1].dy[k & 1] = d1; rp[k >> 1].dz[k & 1] = d2;
    best_t[k] = __builtin_inff();
    best_p[k] = PYQSM_MISS_PRIM;
  }
  const float ru0 = wave_min_f(__double2float_rd(umin)), ru1 = wave_max_f(__double2float_ru(umax));
  const float rv0 = wave_min_f(__double2float_rd(vmin)), rv1 = wave_max_f(__double2float_ru(vmax));

  auto sweep = [&](const Tri t, const uint32_t id) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      f2 px, py, pz, det, tvx, tvy, tvz, U;
      mt_front(rp[q], t, px, py, pz, det, tvx, tvy, tvz, U);
      const f2 key = alive_key(det, U);
      bool alive = u_alive(key[0], det[0]) || u_alive(key[1], det[1]);
      if (__builtin_amdgcn_ballot_w64(alive) != 0) {
        f2 V, Tn;
        mt_back(rp[q], t, tvx, tvy, tvz, V, Tn);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (is_hit(det[h], U[h], V[h], Tn[h])) {
            float tt = Tn[h] / det[h];
            // same winner as the in-order sweep: smallest t, then smallest triangle id
            if (tt < best_t[2 * q + h] || (tt == best_t[2 * q + h] && id < best_p[2 * q + h])) {
              best_t[2 * q + h] = tt;
              best_p[2 * q + h] = id;
            }
          }
        }
      }
    }
  };
  const int nclus = (T + kCluster - 1) / kCluster;
  const int nsup = (nclus + kSuper - 1) / kSuper;
  const float su0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ru0)));
  const float su1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ru1)));
  const float sv0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rv0)));
  const float sv1 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rv1)));
  float4 nxt = srect[0];
  for (int sc = 0; sc < nsup; ++sc) {
    const float4 sr = nxt;
    nxt = srect[sc + 1 < nsup ? sc + 1 : sc];
    if (su1 < sr.x || su0 > sr.y || sv1 < sr.z || sv0 > sr.w) continue;
    const int c1 = (sc + 1) * kSuper < nclus ? (sc + 1) * kSuper : nclus;
    for (int cl = sc * kSuper; cl < c1; ++cl) {
      const float4 cr = crect[cl];  // wave-uniform address
      if (su1 < cr.x || su0 > cr.y || sv1 < cr.z || sv0 > cr.w) continue;
      const int s1 = (cl + 1) * kCluster < T ? (cl + 1) * kCluster : T;
      for (int s = cl * kCluster; s < s1; ++s) {
        const int id = __builtin_amdgcn_readfirstlane(sid[s]);
        sweep(load_tri(tri, id), uint32_t(id));
      }
    }
  }

#pragma unroll
  for (int k = 0; k < RPL; ++k) {
    int64_t r = block_base + int64_t(k) * 64;
    if (r >= R) continue;
    t_hit[r] = best_t[k];
    prim_id[r] = best_p[k];
    if (uv) {
      float u = 0.f, v = 0.f;
      if (best_p[k] != PYQSM_MISS_PRIM) {
        const Tri t = load_tri(tri, int(best_p[k]));
        RayPair one;
        const int q = k >> 1, h = k & 1;
        one.ox = splat(rp[q].ox[h]); one.oy = splat(rp[q].oy[h]); one.oz = splat(rp[q].oz[h]);
        one.dx = splat(rp[q].dx[h]); one.dy = splat(rp[q].dy[h]); one.dz = splat(rp[q].dz[h]);
        f2 px, py, pz, det, tvx, tvy, tvz, U, V, Tn;
        mt_front(one, t, px, py, pz, det, tvx, tvy, tvz, U);
        mt_back(one, t, tvx, tvy, tvz, V, Tn);
        u = U[0] / det[0];
        v = V[0] / det[0];
      }
      uv[2 * r] = u;
      uv[2 * r + 1] = v;
    }
  }
}

// Crossing counts and (optionally) hit records: one ray per lane, all triangles.
// mode 0: counts only. mode 1: write records at offsets[r] + running index.
template <int MODE>
__global__ __launch_bounds__(256) void k_all_hits(const TriRec* __restrict__ tri, int T,
                                                  const float* __restrict__ rays, int64_t R,
                                                  int32_t* __restrict__ counts,
                                                  const int64_t* __restrict__ offsets,
                                                  int64_t cap, uint32_t* __restrict__ ray_ids,
                                                  uint32_t* __restrict__ prim_ids,
                                                  float* __restrict__ ts, float* __restrict__ uv) {
  int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
  RayPair rp;
  float o0 = 0.f, o1 = 0.f, o2 = 0.f, d0 = 0.f, d1 = 0.f, d2 = 0.f;
  if (r < R) {
    const float* p = rays + 6 * r;
    o0 = p[0]; o1 = p[1]; o2 = p[2]; d0 = p[3]; d1 = p[4]; d2 = p[5];
  }
  rp.ox = splat(o0); rp.oy = splat(o1); rp.oz = splat(o2);
  rp.dx = splat(d0); rp.dy = splat(d1); rp.dz = splat(d2);
  int32_t n = 0;
  int64_t base = (MODE == 1 && r < R) ? offsets[r] : 0;
  Tri nxt = load_tri(tri, 0);
  for (int j = 0; j < T; ++j) {
    const Tri t = nxt;
    nxt = load_tri(tri, j + 1 < T ? j + 1 : j);
    f2 px, py, pz, det, tvx, tvy, tvz, U;
    mt_front(rp, t, px, py, pz, det, tvx, tvy, tvz, U);
    const f2 key = alive_key(det, U);
    bool alive = u_alive(key[0], det[0]);
    if (__builtin_amdgcn_ballot_w64(alive) != 0) {
      f2 V, Tn;
      mt_back(rp, t, tvx, tvy, tvz, V, Tn);
      if (r < R && is_hit(det[0], U[0], V[0], Tn[0])) {
        if (MODE == 1) {
          int64_t w = base + n;
          if (w < cap) {
            ray_ids[w] = uint32_t(r);
            prim_ids[w] = uint32_t(j);
            ts[w] = Tn[0] / det[0];
            uv[2 * w] = U[0] / det[0];
            uv[2 * w + 1] = V[0] / det[0];
          }
        }
        ++n;
      }
    }
  }
  if (MODE == 0 && r < R) counts[r] = n;
}

__global__ void k_scan_counts_serial(const int32_t* __restrict__ counts, int64_t R,
                                     int64_t* __restrict__ offsets, int64_t* __restrict__ total) {
  // Single-thread-block exclusive scan in 64-bit; R is small for this path
  // (list_intersections is used with sparse ray grids).
  __shared__ int64_t carry;
  __shared__ int64_t buf[1024];
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < R; base += 1024) {
    int64_t i = base + threadIdx.x;
    int64_t v = i < R ? counts[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      int64_t add = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < R) offsets[i] = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ void k_fill_miss(int64_t R, float* __restrict__ t_hit, uint32_t* __restrict__ prim,
                            float* __restrict__ uv) {
  int64_t r = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (r >= R) return;
  t_hit[r] = __builtin_inff();
  prim[r] = PYQSM_MISS_PRIM;
  if (uv) uv[2 * r] = uv[2 * r + 1] = 0.f;
}

static int launch_cast(Ctx* c, const TriRec* tri, int64_t T, const float* rays, int64_t R,
                       float* t_hit, uint32_t* prim, float* uv) {
  if (R == 0) return 0;
  if (T == 0) {  // empty mesh: every ray misses
    hipLaunchKernelGGL(k_fill_miss, dim3(ceil_div(R, 256)), dim3(256), 0, c->stream, R, t_hit, prim,
                       uv);
    PQ_HIP(hipGetLastError());
    return 0;
  }
  // Parallel rays (one direction, bitwise) take the specialised kernel.
  int* d_flag = nullptr;
  PQ_TRY(c->arena.get(1, &d_flag));
  PQ_HIP(hipMemsetAsync(d_flag, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_check_uniform, dim3(std::min<int64_t>(ceil_div(R, 256), 2048)), dim3(256), 0,
                     c->stream, rays, R, d_flag);
  PQ_HIP(hipGetLastError());
  int uni_bits = 0;
  PQ_HIP(hipMemcpyAsync(&uni_bits, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  const int varied = uni_bits & 1;          // directions differ
  const bool one_origin = !(uni_bits & 2);  // every ray starts at ray 0's origin
  // Rays per lane: 8 when there are enough rays to fill the chip that way,
  // fewer for small batches so that more waves exist.
  const int64_t waves_needed = int64_t(c->cu_count) * 8;
  int np = R >= waves_needed * 64 * 8 ? 4 : (R >= waves_needed * 64 * 4 ? 2 : 1);
  // (12 rays per lane, np = 6, measured 4 % slower than 8 on the 10 M x 500 k sweep)
  if (const char* e = getenv("PYQSM_RAY_NP")) {  // tuning knob
    const int v = atoi(e);
    if (v == 1 || v == 2 || v == 4 || v == 6) np = v;
  }
  const dim3 block(256);
  auto grid_for = [&](int npv) { return dim3(ceil_div(R, 256 * 2 * npv)); };
  const char* cull_env = getenv("PYQSM_RAY_CULL");
  const bool cull = !(cull_env && cull_env[0] == '0');
  if (!varied && cull && T >= 4 * kCluster) {
    // ---- sort triangles on the plane normal to d, build clusters, culled sweep ------
    float hd[3];
    PQ_HIP(hipMemcpyAsync(hd, rays + 3, 12, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    const double dn = std::sqrt(double(hd[0]) * hd[0] + double(hd[1]) * hd[1] + double(hd[2]) * hd[2]);
    if (dn > 0.0 && std::isfinite(dn)) {
      const double d[3] = {hd[0] / dn, hd[1] / dn, hd[2] / dn};
      // first plane axis: the direction in which consecutive ray origins advance (ray
      // grids are stored row by row, so rows become thin rectangles); any vector
      // normal to d otherwise
      double u[3] = {0, 0, 0}, un = 0.0;
      if (R >= 2) {
        float ho[12];
        PQ_HIP(hipMemcpyAsync(ho, rays, 48, hipMemcpyDeviceToHost, c->stream));
        PQ_HIP(hipStreamSynchronize(c->stream));
        const double s0[3] = {double(ho[6]) - ho[0], double(ho[7]) - ho[1], double(ho[8]) - ho[2]};
        const double along = s0[0] * d[0] + s0[1] * d[1] + s0[2] * d[2];
        for (int a = 0; a < 3; ++a) u[a] = s0[a] - along * d[a];
        un = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      }
      if (!(un > 1e-12 * (1.0 + std::fabs(double(hd[0])) + std::fabs(double(hd[1]))))) {
        double h[3] = {0, 0, 0};
        const double ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
        h[ax <= ay && ax <= az ? 0 : (ay <= az ? 1 : 2)] = 1.0;
        u[0] = d[1] * h[2] - d[2] * h[1];
        u[1] = d[2] * h[0] - d[0] * h[2];
        u[2] = d[0] * h[1] - d[1] * h[0];
        un = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
      }
      for (double& q : u) q /= un;
      const double v[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2],
                           d[0] * u[1] - d[1] * u[0]};
      const Basis bs{u[0], u[1], u[2], v[0], v[1], v[2]};
      double* cen = nullptr;
      float4 *rect = nullptr, *crect = nullptr, *srect = nullptr;
      DirRec* srec = nullptr;
      int32_t* sid = nullptr;
      const int nclus = int((T + kCluster - 1) / kCluster);
      PQ_TRY(c->arena.get(size_t(T) * 3, &cen));
      PQ_TRY(c->arena.get(size_t(T), &rect));
      PQ_TRY(c->arena.get(size_t(nclus), &crect));
      const int nsup = (nclus + kSuper - 1) / kSuper;
      PQ_TRY(c->arena.get(size_t(nsup), &srect));
      PQ_TRY(c->arena.get(size_t(T), &srec));
      PQ_TRY(c->arena.get(size_t(T), &sid));
      hipLaunchKernelGGL(k_tri_uv, dim3(ceil_div(T, 256)), dim3(256), 0, c->stream, tri, T, bs, cen,
                         rect);
      PQ_HIP(hipGetLastError());
      double mn[3], mx[3];
      PQ_TRY(cloud_bbox(c, cen, T, mn, mx));
      double ext = std::max(mx[0] - mn[0], mx[1] - mn[1]);
      if (!(ext > 0)) ext = 1.0;
      double box[6] = {mn[0], mn[1], mn[2], mx[0], mx[1], mx[2]};
      DevGrid g;
      PQ_TRY(build_grid(c, cen, T, ext / 1024.0, int64_t(1) << 22, &g, box));
      const float margin = float(1e-4 * ext + 1e-30);
      hipLaunchKernelGGL(k_clusters, dim3(ceil_div(nclus, 256)), dim3(256), 0, c->stream, tri, int(T),
                         rays, g.order, rect, margin, srec, sid, crect);
      hipLaunchKernelGGL(k_super_rects, dim3(ceil_div(nsup, 256)), dim3(256), 0, c->stream, nclus,
                         crect, srect);
      PQ_HIP(hipGetLastError());
      ProfScope ps(c, "cast_rays_culled");
      if (np == 6) np = 4;  // the culled kernel is not VALU bound: smaller waves' rectangles win
      if (np == 4)
        hipLaunchKernelGGL(k_cast_parallel_culled<4>, grid_for(4), block, 0, c->stream, tri, srec, sid, crect,
                           srect, int(T), bs, rays, R, t_hit, prim, uv);
      else if (np == 2)
        hipLaunchKernelGGL(k_cast_parallel_culled<2>, grid_for(2), block, 0, c->stream, tri, srec, sid, crect,
                           srect, int(T), bs, rays, R, t_hit, prim, uv);
      else
        hipLaunchKernelGGL(k_cast_parallel_culled<1>, grid_for(1), block, 0, c->stream, tri, srec, sid, crect,
                           srect, int(T), bs, rays, R, t_hit, prim, uv);
      PQ_HIP(hipGetLastError());
      return 0;
    }
  }
  if (varied && one_origin && cull && T >= 4 * kCluster && R >= 2) {
    // ---- common origin: project onto the image plane, sort, cluster, culled sweep ------
    double* d_sum = nullptr;
    PQ_TRY(c->arena.get(3, &d_sum));
    PQ_HIP(hipMemsetAsync(d_sum, 0, 24, c->stream));
    hipLaunchKernelGGL(k_dir_sum, dim3(std::min<int64_t>(ceil_div(R, 256), 512)), dim3(256), 0,
                       c->stream, rays, R, d_sum);
    double hs[3];
    float ho[12];
    PQ_HIP(hipMemcpyAsync(hs, d_sum, 24, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(ho, rays, 48, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
    const double nn = std::sqrt(hs[0] * hs[0] + hs[1] * hs[1] + hs[2] * hs[2]);
    if (nn > 1e-3 * double(R) && std::isfinite(nn)) {
      const double n[3] = {hs[0] / nn, hs[1] / nn, hs[2] / nn};
      // first image axis: the way consecutive rays turn (pixel rows become thin rectangles)
      double a[3], an = 0.0;
      {
        // difference of the two rays' points on the plane n.x = 1 (lies in that plane whatever
        // the lengths of the direction vectors are)
        const double w0 = ho[3] * n[0] + ho[4] * n[1] + ho[5] * n[2];
        const double w1 = ho[9] * n[0] + ho[10] * n[1] + ho[11] * n[2];
        double s0[3] = {0, 0, 0};
        if (w0 > 0 && w1 > 0)
          for (int k = 0; k < 3; ++k) s0[k] = double(ho[9 + k]) / w1 - double(ho[3 + k]) / w0;
        const double along = s0[0] * n[0] + s0[1] * n[1] + s0[2] * n[2];
        for (int k = 0; k < 3; ++k) a[k] = s0[k] - along * n[k];
        an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      }
      if (!(an > 1e-12)) {
        double h[3] = {0, 0, 0};
        const double ax = std::fabs(n[0]), ay = std::fabs(n[1]), az = std::fabs(n[2]);
        h[ax <= ay && ax <= az ? 0 : (ay <= az ? 1 : 2)] = 1.0;
        a[0] = n[1] * h[2] - n[2] * h[1];
        a[1] = n[2] * h[0] - n[0] * h[2];
        a[2] = n[0] * h[1] - n[1] * h[0];
        an = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
      }
      for (double& q : a) q /= an;
      const double b[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2],
                           n[0] * a[1] - n[1] * a[0]};
      const Camera cam{ho[0], ho[1], ho[2], a[0], a[1], a[2], b[0], b[1], b[2], n[0], n[1], n[2]};
      double* img = nullptr;
      int* d_bad = nullptr;
      PQ_TRY(c->arena.get(size_t(R) * 3, &img));
      PQ_TRY(c->arena.get(1, &d_bad));
      PQ_HIP(hipMemsetAsync(d_bad, 0, 4, c->stream));
      hipLaunchKernelGGL(k_ray_image, dim3(ceil_div(R, 256)), dim3(256), 0, c->stream, rays, R, cam,
                         0.25, img, d_bad);
      PQ_HIP(hipGetLastError());
      double mn[3], mx[3];
      PQ_TRY(cloud_bbox(c, img, R, mn, mx));  // synchronises
      int bad = 0;
      PQ_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
      PQ_HIP(hipStreamSynchronize(c->stream));
      if (!bad) {
        double ext = std::max(mx[0] - mn[0], mx[1] - mn[1]);
        if (!(ext > 0)) ext = 1.0;
        double* cen = nullptr;
        float4 *rect = nullptr, *crect = nullptr, *srect = nullptr;
        int32_t* sid = nullptr;
        const int nclus = int((T + kCluster - 1) / kCluster);
        const int nsup = (nclus + kSuper - 1) / kSuper;
        PQ_TRY(c->arena.get(size_t(T) * 3, &cen));
        PQ_TRY(c->arena.get(size_t(T), &rect));
        PQ_TRY(c->arena.get(size_t(nclus), &crect));
        PQ_TRY(c->arena.get(size_t(nsup), &srect));
        PQ_TRY(c->arena.get(size_t(T), &sid));
        hipLaunchKernelGGL(k_tri_image, dim3(ceil_div(T, 256)), dim3(256), 0, c->stream, tri, T, cam,
                           mn[0], mx[0], mn[1], mx[1], cen, rect);
        PQ_HIP(hipGetLastError());
        double box[6] = {mn[0], mn[1], 0.0, mx[0], mx[1], 0.0};
        DevGrid g;
        PQ_TRY(build_grid(c, cen, T, ext / 1024.0, int64_t(1) << 22, &g, box));
        const float margin = float(1e-4 * ext + 1e-30);
        hipLaunchKernelGGL(k_clusters_image, dim3(ceil_div(nclus, 256)), dim3(256), 0, c->stream, int(T),
                           g.order, rect, margin, sid, crect);
        hipLaunchKernelGGL(k_super_rects, dim3(ceil_div(nsup, 256)), dim3(256), 0, c->stream, nclus,
                           crect, srect);
        PQ_HIP(hipGetLastError());
        ProfScope ps(c, "cast_rays_culled");
        if (np == 6) np = 4;
        if (np == 4)
          hipLaunchKernelGGL(k_cast_pinhole_culled<4>, grid_for(4), block, 0, c->stream, tri, sid, crect,
                             srect, int(T), cam, rays, R, t_hit, prim, uv);
        else if (np == 2)
          hipLaunchKernelGGL(k_cast_pinhole_culled<2>, grid_for(2), block, 0, c->stream, tri, sid, crect,
                             srect, int(T), cam, rays, R, t_hit, prim, uv);
        else
          hipLaunchKernelGGL(k_cast_pinhole_culled<1>, grid_for(1), block, 0, c->stream, tri, sid, crect,
                             srect, int(T), cam, rays, R, t_hit, prim, uv);
        PQ_HIP(hipGetLastError());
        return 0;
      }
    }
  }
  if (!varied) {
    DirRec* rec = nullptr;
    PQ_TRY(c->arena.get(size_t(T), &rec));
    hipLaunchKernelGGL(k_dir_records, dim3(ceil_div(T, 256)), dim3(256), 0, c->stream, tri, T, rays,
                       rec);
    ProfScope ps(c, "cast_rays");
    if (np == 6)
      hipLaunchKernelGGL(k_cast_parallel<6>, grid_for(6), block, 0, c->stream, tri, rec, int(T), rays, R,
                         t_hit, prim, uv);
    else if (np == 4)
      hipLaunchKernelGGL(k_cast_parallel<4>, grid_for(4), block, 0, c->stream, tri, rec, int(T), rays, R,
                         t_hit, prim, uv);
    else if (np == 2)
      hipLaunchKernelGGL(k_cast_parallel<2>, grid_for(2), block, 0, c->stream, tri, rec, int(T), rays, R,
                         t_hit, prim, uv);
    else
      hipLaunchKernelGGL(k_cast_parallel<1>, grid_for(1), block, 0, c->stream, tri, rec, int(T), rays, R,
                         t_hit, prim, uv);
    PQ_HIP(hipGetLastError());
    return 0;
  }
  ProfScope ps(c, "cast_rays");
  if (np == 6) np = 4;
  if (np == 4)
    hipLaunchKernelGGL(k_cast_rays<4>, grid_for(4), block, 0, c->stream, tri, int(T), rays, R, t_hit, prim,
                       uv);
  else if (np == 2)
    hipLaunchKernelGGL(k_cast_rays<2>, grid_for(2), block, 0, c->stream, tri, int(T), rays, R, t_hit, prim,
                       uv);
  else
    hipLaunchKernelGGL(k_cast_rays<1>, grid_for(1), block, 0, c->stream, tri, int(T), rays, R, t_hit, prim,
                       uv);
  PQ_HIP(hipGetLastError());
  return 0;
}

static int expand(Ctx* c, const float* verts, int64_t V, const int32_t* tris, int64_t T,
                  TriRec* out) {
  if (T == 0) return 0;
  int* bad = nullptr;
  PQ_TRY(c->arena.get(1, &bad));
  PQ_HIP(hipMemsetAsync(bad, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_expand_tris, dim3(ceil_div(T, 256)), dim3(256), 0, c->stream, verts, V, tris,
                     T, out, bad);
  PQ_HIP(hipGetLastError());
  int h = 0;
  PQ_HIP(hipMemcpyAsync(&h, bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (h) return fail(PYQSM_EINVAL, "triangle index outside [0, V)");
  return 0;
}

static int check_sizes(int64_t V, int64_t T, int64_t R) {
  if (V < 0 || T < 0 || R < 0) return fail(PYQSM_EINVAL, "negative size");
  if (T > 0x7FFFFFF0LL) return fail(PYQSM_ERANGE, "more than 2^31 triangles");
  if (R > 0xFFFFFFF0LL) return fail(PYQSM_ERANGE, "more than 2^32 rays per call");
  return 0;
}

// what multi.hip (the RCCL-sharded sweep) calls on each device
int ray_expand(Ctx* c, const float* verts, int64_t V, const int32_t* tris, int64_t T, float* tri12) {
  return expand(c, verts, V, tris, T, reinterpret_cast<TriRec*>(tri12));
}
int ray_launch(Ctx* c, const float* tri12, int64_t T, const float* rays, int64_t R, float* t_hit,
               uint32_t* prim, float* uv) {
  return launch_cast(c, reinterpret_cast<const TriRec*>(tri12), T, rays, R, t_hit, prim, uv);
}
int ray_check_sizes(int64_t V, int64_t T, int64_t R) { return check_sizes(V, T, R); }

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_expand_tris_dev(const float* verts_dev, int64_t V, const int32_t* tris_dev, int64_t T,
                          float* tri12_dev, int32_t device) {
  PQ_API_RANGE("pyqsm_expand_tris_dev");
  PQ_TRY(check_sizes(V, T, 0));
  if (T > 0 && (!verts_dev || !tris_dev || !tri12_dev))
    return fail(PYQSM_EINVAL, "pyqsm_expand_tris_dev: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  return expand(c, verts_dev, V, tris_dev, T, reinterpret_cast<TriRec*>(tri12_dev));
}

int pyqsm_cast_rays_dev(const float* tri12_dev, int64_t T, const float* rays_dev, int64_t R,
                        float* t_hit_dev, uint32_t* prim_id_dev, float* uv_dev, int32_t device) {
  PQ_API_RANGE("pyqsm_cast_rays_dev");
  PQ_TRY(check_sizes(0, T, R));
  if (R > 0 && (!rays_dev || !t_hit_dev || !prim_id_dev || (T > 0 && !tri12_dev)))
    return fail(PYQSM_EINVAL, "pyqsm_cast_rays_dev: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();  // scratch of the previous call (per-direction records, sort buffers)
  return launch_cast(c, reinterpret_cast<const TriRec*>(tri12_dev), T, rays_dev, R, t_hit_dev,
                     prim_id_dev, uv_dev);
}

int pyqsm_cast_rays(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                    const float* rays, int64_t R, float* t_hit, uint32_t* prim_id, float* uv,
                    int32_t device) {
  PQ_API_RANGE("pyqsm_cast_rays");
  PQ_TRY(check_sizes(V, T, R));
  if (R == 0) return 0;
  if (!rays || !t_hit || !prim_id || (T > 0 && (!verts || !tris)))
    return fail(PYQSM_EINVAL, "pyqsm_cast_rays: NULL pointer");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  float *d_verts, *d_rays, *d_t, *d_uv = nullptr;
  int32_t* d_tris;
  uint32_t* d_p;
  TriRec* d_rec;
  PQ_TRY(c->arena.get(size_t(V) * 3 + 1, &d_verts));
  PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_tris));
  PQ_TRY(c->arena.get(size_t(T) + 1, &d_rec));
  PQ_TRY(c->arena.get(size_t(R) * 6, &d_rays));
  PQ_TRY(c->arena.get(size_t(R), &d_t));
  PQ_TRY(c->arena.get(size_t(R), &d_p));
  if (uv) PQ_TRY(c->arena.get(size_t(R) * 2, &d_uv));
  if (T > 0) {
    PQ_HIP(hipMemcpyAsync(d_verts, verts, size_t(V) * 12, hipMemcpyHostToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(d_tris, tris, size_t(T) * 12, hipMemcpyHostToDevice, c->stream));
  }
  PQ_HIP(hipMemcpyAsync(d_rays, rays, size_t(R) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_TRY(expand(c, d_verts, V, d_tris, T, d_rec));
  PQ_TRY(launch_cast(c, d_rec, T, d_rays, R, d_t, d_p, d_uv));
  PQ_HIP(hipMemcpyAsync(t_hit, d_t, size_t(R) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(prim_id, d_p, size_t(R) * 4, hipMemcpyDeviceToHost, c->stream));
  if (uv) PQ_HIP(hipMemcpyAsync(uv, d_uv, size_t(R) * 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int pyqsm_list_intersections(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                             const float* rays, int64_t R, int32_t* counts, uint32_t* ray_ids,
                             uint32_t* prim_ids, float* t, float* uv, int64_t hits_cap,
                             int64_t* n_hits, int32_t device) {
  PQ_API_RANGE("pyqsm_list_intersections");
  PQ_TRY(check_sizes(V, T, R));
  if (n_hits) *n_hits = 0;
  if (R == 0) return 0;
  if (!rays || !counts || (T > 0 && (!verts || !tris)))
    return fail(PYQSM_EINVAL, "pyqsm_list_intersections: NULL pointer");
  if (hits_cap > 0 && (!ray_ids || !prim_ids || !t || !uv))
    return fail(PYQSM_EINVAL, "pyqsm_list_intersections: hits_cap > 0 needs every record array");
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  float *d_verts, *d_rays;
  int32_t *d_tris, *d_counts;
  TriRec* d_rec;
  int64_t *d_off, *d_total;
  PQ_TRY(c->arena.get(size_t(V) * 3 + 1, &d_verts));
  PQ_TRY(c->arena.get(size_t(T) * 3 + 1, &d_tris));
  PQ_TRY(c->arena.get(size_t(T) + 1, &d_rec));
  PQ_TRY(c->arena.get(size_t(R) * 6, &d_rays));
  PQ_TRY(c->arena.get(size_t(R), &d_counts));
  PQ_TRY(c->arena.get(size_t(R), &d_off));
  PQ_TRY(c->arena.get(1, &d_total));
  if (T > 0) {
    PQ_HIP(hipMemcpyAsync(d_verts, verts, size_t(V) * 12, hipMemcpyHostToDevice, c->stream));
    PQ_HIP(hipMemcpyAsync(d_tris, tris, size_t(T) * 12, hipMemcpyHostToDevice, c->stream));
  }
  PQ_HIP(hipMemcpyAsync(d_rays, rays, size_t(R) * 24, hipMemcpyHostToDevice, c->stream));
  PQ_TRY(expand(c, d_verts, V, d_tris, T, d_rec));
  const dim3 grid(ceil_div(R, 256));
  if (T == 0) {
    memset(counts, 0, size_t(R) * 4);
    return 0;
  }
  {
    ProfScope ps(c, "all_hits");
    hipLaunchKernelGGL(k_all_hits<0>, grid, dim3(256), 0, c->stream, d_rec, int(T), d_rays, R,
                       d_counts, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(counts, d_counts, size_t(R) * 4, hipMemcpyDeviceToHost, c->stream));
  hipLaunchKernelGGL(k_scan_counts_serial, dim3(1), dim3(1024), 0, c->stream, d_counts, R, d_off,
                     d_total);
  PQ_HIP(hipGetLastError());
  int64_t total = 0;
  PQ_HIP(hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (n_hits) *n_hits = total;
  if (hits_cap > 0 && total > 0) {
    const int64_t cap = hits_cap < total ? hits_cap : total;
    uint32_t *d_r, *d_p;
    float *d_t, *d_uv;
    PQ_TRY(c->arena.get(size_t(cap), &d_r));
    PQ_TRY(c->arena.get(size_t(cap), &d_p));
    PQ_TRY(c->arena.get(size_t(cap), &d_t));
    PQ_TRY(c->arena.get(size_t(cap) * 2, &d_uv));
    hipLaunchKernelGGL(k_all_hits<1>, grid, dim3(256), 0, c->stream, d_rec, int(T), d_rays, R,
                       nullptr, d_off, cap, d_r, d_p, d_t, d_uv);
    PQ_HIP(hipGetLastError());
    PQ_HIP(hipMemcpyAsync(ray_ids, d_r, size_t(cap) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(prim_ids, d_p, size_t(cap) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(t, d_t, size_t(cap) * 4, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipMemcpyAsync(uv, d_uv, size_t(cap) * 8, hipMemcpyDeviceToHost, c->stream));
    PQ_HIP(hipStreamSynchronize(c->stream));
  }
  return 0;
}

}  // extern "C"
