"""Test plumbing for tests/dist_worker.py: the sharded sweep over any ``torch.distributed``
group (``gloo`` on CPU), generic over the per-shard compute function. Uses the product's shard
arithmetic (``pyqsm_amd.parallel.shard_bounds``); the product itself never imports torch."""
import numpy as np

from pyqsm_amd.parallel import shard_bounds, shard_sizes


def broadcast_mesh(verts, tris, dist, device=None, src: int = 0):
    """Replicate (verts f32 [V,3], tris i32 [T,3]) from `src` to every rank.
    Ranks other than `src` may pass None. Returns NumPy arrays on every rank."""
    import torch
    rank = dist.get_rank()
    dev = device if device is not None else "cpu"
    shape = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        verts = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
        tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        shape = torch.tensor([verts.shape[0], tris.shape[0]], dtype=torch.int64, device=dev)
    dist.broadcast(shape, src=src)
    V, T = (int(x) for x in shape.tolist())
    tv = (torch.from_numpy(verts).to(dev) if rank == src
          else torch.empty((V, 3), dtype=torch.float32, device=dev))
    tt = (torch.from_numpy(tris).to(dev) if rank == src
          else torch.empty((T, 3), dtype=torch.int32, device=dev))
    dist.broadcast(tv, src=src)
    dist.broadcast(tt, src=src)
    return tv.cpu().numpy(), tt.cpu().numpy()


def cast_rays_sharded(verts, tris, rays, dist, cast_fn, device=None):
    """Closest-hit sweep of `rays` [R,6] (the same full array on every rank) with
    each rank computing its contiguous shard through ``cast_fn(verts, tris,
    rays_shard) -> (t_hit, prim_id, uv)`` and an all-gather assembling the full
    result on every rank. Results are identical to a single-rank call because
    rays are independent."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else "cpu"
    rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    R = rays.shape[0]
    b, e = shard_bounds(R, world, rank)
    t, p, uv = cast_fn(verts, tris, rays[b:e])
    sizes = shard_sizes(R, world)
    cap = max(sizes) if sizes else 0
    # pack (t, prim, u, v) as 4 x 32-bit words per ray so one collective moves it all
    packed = np.zeros((cap, 4), dtype=np.uint32)
    packed[: e - b, 0] = np.asarray(t, dtype=np.float32).view(np.uint32)
    packed[: e - b, 1] = np.asarray(p, dtype=np.uint32)
    packed[: e - b, 2:] = np.asarray(uv, dtype=np.float32).reshape(-1, 2).view(np.uint32)
    mine = torch.from_numpy(packed.view(np.int32)).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    full = np.concatenate([parts[r].cpu().numpy().view(np.uint32)[: sizes[r]]
                           for r in range(world)], axis=0)
    return (full[:, 0].copy().view(np.float32), full[:, 1].copy(),
            full[:, 2:].copy().view(np.float32))
