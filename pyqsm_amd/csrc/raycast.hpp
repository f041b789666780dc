// raycast.hpp — the per-device pieces of the ray sweep that multi.hip shards over GPUs.
#pragma once
#include "common.hpp"

namespace pyqsm {

// verts / tris (device) -> 48-byte records (v0, e1, e2, pad), f32 [T,12]. Synchronises the
// stream once (index check).
int ray_expand(Ctx* c, const float* verts, int64_t V, const int32_t* tris, int64_t T, float* tri12);
// Closest-hit sweep of R device-resident rays over the records; asynchronous on c->stream after
// its setup (the culled paths read a few bytes back first). Scratch comes from c->arena.
int ray_launch(Ctx* c, const float* tri12, int64_t T, const float* rays, int64_t R, float* t_hit,
               uint32_t* prim, float* uv);
int ray_check_sizes(int64_t V, int64_t T, int64_t R);

}  // namespace pyqsm
