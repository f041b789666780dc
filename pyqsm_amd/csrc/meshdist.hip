// meshdist.hip — distance from query points to a triangle mesh on gfx950.
//
// Stands in for open3d RaycastingScene.compute_distance / compute_signed_distance as used
// by `mri` (pyQSM/viz/ray_casting.py:237-260): N random points and a 64^3 grid of query
// points against the scene's mesh. Brute force like the ray sweep: one lane per query, the
// expanded triangle records arrive by scalar loads (the triangle index is wave-uniform), the
// closest point on each triangle follows Ericson's region walk (Real-Time Collision
// Detection, 5.1.5) in fp32 with separately rounded products — the CPU oracle
// (orc_point_mesh_distance) repeats the same operation sequence, so distances and the
// index of the closest triangle (lowest index on ties) are bit-identical.
// FP32-VALU bound: ~60 flop per point-triangle pair.
#include "common.hpp"

namespace pyqsm {

struct alignas(16) DTri {  // 48 bytes: a, ab, ac (what raycast.hip calls TriRec)
  float ax, ay, az, abx, aby, abz, acx, acy, acz, pad0, pad1, pad2;
};

__global__ void k_dist_tris(const float* __restrict__ verts, int64_t V,
                            const int32_t* __restrict__ tris, int64_t T, DTri* __restrict__ out,
                            int* __restrict__ bad) {
  int64_t i = blockIdx.x * int64_t(blockDim.x) + threadIdx.x;
  if (i >= T) return;
  const int a = tris[3 * i], b = tris[3 * i + 1], c = tris[3 * i + 2];
  if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) {
    *bad = 1;
    return;
  }
  const float ax = verts[3 * a], ay = verts[3 * a + 1], az = verts[3 * a + 2];
  out[i] = DTri{ax, ay, az, verts[3 * b] - ax, verts[3 * b + 1] - ay, verts[3 * b + 2] - az,
                verts[3 * c] - ax, verts[3 * c + 1] - ay, verts[3 * c + 2] - az, 0.f, 0.f, 0.f};
}

__device__ __forceinline__ float dot3f(float ax, float ay, float az, float bx, float by, float bz) {
  float d = ax * bx;
  d = d + ay * by;
  d = d + az * bz;
  return d;
}

// squared distance from p to the triangle (a, a + ab, a + ac)
__device__ __forceinline__ float tri_dist2(const DTri& t, float px, float py, float pz) {
  const float apx = px - t.ax, apy = py - t.ay, apz = pz - t.az;
  const float d1 = dot3f(t.abx, t.aby, t.abz, apx, apy, apz);
  const float d2 = dot3f(t.acx, t.acy, t.acz, apx, apy, apz);
  float cx, cy, cz;  // closest point minus a
  const float bpx = apx - t.abx, bpy = apy - t.aby, bpz = apz - t.abz;
  const float d3 = dot3f(t.abx, t.aby, t.abz, bpx, bpy, bpz);
  const float d4 = dot3f(t.acx, t.acy, t.acz, bpx, bpy, bpz);
  const float cpx = apx - t.acx, cpy = apy - t.acy, cpz = apz - t.acz;
  const float d5 = dot3f(t.abx, t.aby, t.abz, cpx, cpy, cpz);
  const float d6 = dot3f(t.acx, t.acy, t.acz, cpx, cpy, cpz);
  const float vc = d1 * d4 - d3 * d2;
  const float vb = d5 * d2 - d1 * d6;
  const float va = d3 * d6 - d5 * d4;
  if (d1 <= 0.f && d2 <= 0.f) {  // vertex a
    cx = cy = cz = 0.f;
  } else if (d3 >= 0.f && d4 <= d3) {  // vertex b
    cx = t.abx; cy = t.aby; cz = t.abz;
  } else if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {  // edge ab
    const float v = d1 / (d1 - d3);
    cx = v * t.abx; cy = v * t.aby; cz = v * t.abz;
  } else if (d6 >= 0.f && d5 <= d6) {  // vertex c
    cx = t.acx; cy = t.acy; cz = t.acz;
  } else if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {  // edge ac
    const float w = d2 / (d2 - d6);
    cx = w * t.acx; cy = w * t.acy; cz = w * t.acz;
  } else if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {  // edge bc
    const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    cx = t.abx + w * (t.acx - t.abx);
    cy = t.aby + w * (t.acy - t.aby);
    cz = t.abz + w * (t.acz - t.abz);
  } else {  // interior
    const float denom = 1.f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    cx = t.abx * v + t.acx * w;
    cy = t.aby * v + t.acy * w;
    cz = t.abz * v + t.acz * w;
  }
  const float ex = apx - cx, ey = apy - cy, ez = apz - cz;
  return dot3f(ex, ey, ez, ex, ey, ez);
}

__global__ __launch_bounds__(256) void k_point_mesh_dist(const DTri* __restrict__ tri, int T,
                                                         const float* __restrict__ qry, int64_t Q,
                                                         float* __restrict__ dist,
                                                         uint32_t* __restrict__ prim) {
  const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
  const bool live = i < Q;
  const float px = live ? qry[3 * i] : 0.f, py = live ? qry[3 * i + 1] : 0.f,
              pz = live ? qry[3 * i + 2] : 0.f;
  float best = __builtin_inff();
  uint32_t bp = PYQSM_MISS_PRIM;
  for (int j = 0; j < T; ++j) {
    const DTri t = tri[j];  // wave-uniform address: scalar loads
    const float d = tri_dist2(t, px, py, pz);
    if (d < best) {
      best = d;
      bp = uint32_t(j);
    }
  }
  if (live) {
    dist[i] = sqrtf(best);
    prim[i] = bp;
  }
}

}  // namespace pyqsm

using namespace pyqsm;

extern "C" {

int pyqsm_point_mesh_distance(const float* verts, int64_t V, const int32_t* tris, int64_t T,
                              const float* qry, int64_t Q, float* dist, uint32_t* prim,
                              int32_t device) {
  PQ_API_RANGE("pyqsm_point_mesh_distance");
  if (V < 0 || T < 0 || Q < 0) return fail(PYQSM_EINVAL, "negative size");
  if (Q == 0) return 0;
  if (!qry || !dist || !prim) return fail(PYQSM_EINVAL, "pyqsm_point_mesh_distance: NULL pointer");
  if (T > 0 && (!verts || !tris)) return fail(PYQSM_EINVAL, "pyqsm_point_mesh_distance: NULL pointer");
  if (T > 0x7FFFFF00LL || V > 0x7FFFFF00LL) return fail(PYQSM_ERANGE, "mesh too large");
  if (T == 0) {  // no surface: infinitely far
    for (int64_t i = 0; i < Q; ++i) {
      dist[i] = HUGE_VALF;
      prim[i] = PYQSM_MISS_PRIM;
    }
    return 0;
  }
  Ctx* c = ctx_for(device);
  if (!c) return PYQSM_ENODEV;
  std::lock_guard<std::mutex> lk(c->mu);
  c->arena.reset();
  float *d_v, *d_q, *d_dist;
  int32_t* d_t;
  uint32_t* d_prim;
  DTri* d_rec;
  int* d_bad;
  PQ_TRY(c->arena.get(size_t(V) * 3 + 1, &d_v));
  PQ_TRY(c->arena.get(size_t(T) * 3, &d_t));
  PQ_TRY(c->arena.get(size_t(Q) * 3, &d_q));
  PQ_TRY(c->arena.get(size_t(Q), &d_dist));
  PQ_TRY(c->arena.get(size_t(Q), &d_prim));
  PQ_TRY(c->arena.get(size_t(T), &d_rec));
  PQ_TRY(c->arena.get(1, &d_bad));
  PQ_HIP(hipMemcpyAsync(d_v, verts, size_t(V) * 12, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_t, tris, size_t(T) * 12, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemcpyAsync(d_q, qry, size_t(Q) * 12, hipMemcpyHostToDevice, c->stream));
  PQ_HIP(hipMemsetAsync(d_bad, 0, 4, c->stream));
  hipLaunchKernelGGL(k_dist_tris, dim3(ceil_div(T, 256)), dim3(256), 0, c->stream, d_v, V, d_t, T, d_rec,
                     d_bad);
  int bad = 0;
  PQ_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  if (bad) return fail(PYQSM_EINVAL, "triangle index outside the vertex array");
  {
    ProfScope ps(c, "point_mesh_distance");
    hipLaunchKernelGGL(k_point_mesh_dist, dim3(ceil_div(Q, 256)), dim3(256), 0, c->stream, d_rec, int(T),
                       d_q, Q, d_dist, d_prim);
    PQ_HIP(hipGetLastError());
  }
  PQ_HIP(hipMemcpyAsync(dist, d_dist, size_t(Q) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipMemcpyAsync(prim, d_prim, size_t(Q) * 4, hipMemcpyDeviceToHost, c->stream));
  PQ_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // extern "C"
