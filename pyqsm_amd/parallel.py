"""Multi-GPU ray sweep: rays sharded contiguously over ranks, mesh replicated.

The only collectives on the data path are a broadcast of the mesh (24 MB of expanded
records for 500 k triangles) and an all-gather of the per-shard results (8 bytes per ray);
the sweep itself needs no exchange. DBSCAN / kNN / skeleton / RANSAC do
not shard: replicas only.

Three ways to run it, all with the same shard arithmetic (:func:`shard_bounds`):

* one process, several GPUs: ``hip.cast_rays_multi`` / ``cast_rays(..., n_devices=N)`` —
  RCCL inside the library (``ncclCommInitAll``), no torch;
* one process per GPU (torchrun-style): :class:`NativeComm` — an RCCL communicator inside the
  library per process, its id shipped through a rendezvous the caller provides (a file here,
  or any ``torch.distributed`` group incl. CPU ``gloo``); every byte of the data path moves
  through ``pyqsm_comm_*``;
* any ``torch.distributed`` group (``gloo`` on CPU in the tests): :func:`broadcast_mesh` /
  :func:`cast_rays_sharded`, generic over the per-shard compute function.

``torch`` is imported lazily and only by the functions that take a ``dist`` argument.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_items: int, world: int, rank: int):
    """[begin, end) of the contiguous shard of `rank`; sizes differ by at most one
    and the shards tile [0, n_items) in rank order."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(int(n_items), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_sizes(n_items: int, world: int):
    return [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0]
            for r in range(world)]


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


# --------------------------------------------------------------------------------------
# RCCL inside the library, one process per GPU

class NativeComm:
    """The library's own RCCL communicator of this process (``pyqsm_comm_*``).

    ``NativeComm.from_env()`` reads RANK / WORLD_SIZE / LOCAL_RANK (torchrun's variables) and
    exchanges the 128-byte RCCL id through a file; ``NativeComm.from_torch(dist)`` ships it
    through an existing ``torch.distributed`` group (``gloo`` is enough: the id is host
    bytes). Device buffers are :class:`pyqsm_amd.hip.DeviceBuffer`."""

    def __init__(self, id_bytes: bytes, world: int, rank: int, device: int):
        import ctypes
        from . import _lib
        self._lib, self._check = _lib.load(), _lib.check
        self.world, self.rank, self.device = int(world), int(rank), int(device)
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(id_bytes)
        self._check(self._lib.pyqsm_comm_init_rank(buf, self.world, self.rank, self.device))

    @staticmethod
    def new_id() -> bytes:
        import ctypes
        from . import _lib
        buf = (ctypes.c_uint8 * 128)()
        _lib.check(_lib.load().pyqsm_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_torch(cls, dist, device: int):
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.new_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(box[0], world, rank, device)

    @classmethod
    def from_env(cls, device=None, directory: str = "/tmp", timeout_s: float = 120.0):
        """File rendezvous for single-node launches: rank 0 writes the id to a file named after
        the launcher's pid and MASTER_PORT, the other ranks wait for it."""
        import os
        import time
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", str(rank)))
        path = os.path.join(directory, "pyqsm_comm_%d_%s.id" % (os.getppid(),
                                                                os.environ.get("MASTER_PORT", "0")))
        if rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(cls.new_id())
            os.replace(tmp, path)                       # atomic: readers never see a partial id
        t0 = time.time()
        while True:
            # a file left by an earlier launch with the same pid and port is older than this
            # launcher's children: ignore anything written before this process started - 5 min
            if os.path.exists(path) and os.path.getsize(path) == 128 and \
                    os.path.getmtime(path) > _process_start_time() - 300:
                with open(path, "rb") as f:
                    id_bytes = f.read()
                break
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"no RCCL id at {path} after {timeout_s} s")
            time.sleep(0.01)
        comm = cls(id_bytes, world, rank, device)
        comm.max_over_ranks(0.0)                        # everybody has read the file
        if rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
        return comm

    def broadcast(self, buf, nbytes: int, root: int = 0) -> None:
        self._check(self._lib.pyqsm_comm_broadcast_dev(buf.ptr, int(nbytes), int(root)))

    def all_gather(self, send_ptr: int, recv, bytes_per_rank: int) -> None:
        self._check(self._lib.pyqsm_comm_all_gather_dev(send_ptr, recv.ptr, int(bytes_per_rank)))

    def max_over_ranks(self, x: float) -> float:
        """Maximum of ``x`` over the ranks; returns once every rank has it (a barrier)."""
        import ctypes
        v = ctypes.c_double(float(x))
        self._check(self._lib.pyqsm_comm_all_reduce_max(ctypes.byref(v)))
        return v.value

    def barrier(self) -> None:
        self.max_over_ranks(0.0)

    def close(self) -> None:
        self._lib.pyqsm_comm_finalize()


def _process_start_time() -> float:
    import os
    import time
    try:
        with open("/proc/self/stat") as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - up + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return time.time()


class ShardedSweep:
    """Config 4's sweep on one rank of a :class:`NativeComm`: the mesh is expanded on rank 0 and
    broadcast, this rank's contiguous shard of the rays lives in HBM, and every call of
    :meth:`run` sweeps the shard and all-gathers (t, prim) so that each rank holds the result
    for ALL rays: ``t_all`` / ``prim_all`` (NumPy, in ray order) after :meth:`results`."""

    def __init__(self, comm: NativeComm, verts, tris, rays_shard, n_rays_total: int):
        from . import hip
        self.comm, self.hip = comm, hip
        self.R = int(n_rays_total)
        self.sizes = shard_sizes(self.R, comm.world)
        self.cap = max(self.sizes) if self.sizes else 0
        dev = comm.device
        T = int(np.asarray(tris).reshape(-1, 3).shape[0])
        if comm.rank == 0:
            self.mesh = hip.DeviceMesh(verts, tris, dev)
        else:                                            # records arrive over RCCL
            self.mesh = hip.DeviceMesh.__new__(hip.DeviceMesh)
            self.mesh.device, self.mesh.n_tris = dev, T
            self.mesh.records = hip.DeviceBuffer(max(1, T) * 48, dev)
        comm.broadcast(self.mesh.records, T * 48, root=0)
        rays_shard = np.ascontiguousarray(rays_shard, dtype=np.float32).reshape(-1, 6)
        if rays_shard.shape[0] != self.sizes[comm.rank]:
            raise ValueError("rays_shard does not have this rank's shard size")
        self.n_local = rays_shard.shape[0]
        self.d_rays = hip.DeviceBuffer.from_array(rays_shard, dev) if self.n_local else None
        # one block of [t(cap) | prim(cap)] 32-bit words per rank
        self.block = hip.DeviceBuffer(max(1, comm.world * 2 * self.cap * 4), dev)
        hip.sync(dev)

    def run(self) -> None:
        off = self.block.ptr + self.comm.rank * 2 * self.cap * 4
        if self.n_local:
            self.hip.cast_rays_dev(self.mesh, self.d_rays.ptr, self.n_local, off, off + self.cap * 4)
        self.comm.all_gather(off, self.block, 2 * self.cap * 4)

    def results(self):
        self.hip.sync(self.comm.device)
        words = self.block.download((self.comm.world, 2, self.cap), np.uint32)
        t = np.concatenate([words[r, 0, : self.sizes[r]] for r in range(self.comm.world)])
        p = np.concatenate([words[r, 1, : self.sizes[r]] for r in range(self.comm.world)])
        return t.view(np.float32), p
