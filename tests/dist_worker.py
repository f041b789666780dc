"""Worker for tests/test_dist_gloo.py: world_size-2 run of the sharded ray sweep on
the gloo backend. The per-shard compute is the CPU oracle (there is no GPU here);
what is under test is the sharding, the mesh broadcast and the result gather."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from pyqsm_amd import synth  # noqa: E402
from tests.dist_helpers import broadcast_mesh, cast_rays_sharded  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    verts = tris = None
    if rank == 0:
        verts, tris = synth.canopy_mesh(800, seed=4, side=0.6)
    verts, tris = broadcast_mesh(verts, tris, dist)
    ref_v, ref_t = synth.canopy_mesh(800, seed=4, side=0.6)
    assert np.array_equal(verts, ref_v) and np.array_equal(tris, ref_t)
    for R in (1001, 7, 1):                       # ragged shards, fewer rays than ranks
        rays = synth.sun_rays(verts, R)
        t, p, uv = cast_rays_sharded(verts, tris, rays, dist, oracle.cast_rays)
        t0, p0, uv0 = oracle.cast_rays(verts, tris, rays)
        assert np.array_equal(t, t0) and np.array_equal(p, p0) and np.array_equal(uv, uv0), R
    dist.barrier()
    if rank == 0:
        print(f"gloo sharded sweep ok world={world}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
