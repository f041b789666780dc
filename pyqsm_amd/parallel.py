"""Multi-GPU ray sweep: rays sharded contiguously over ranks, mesh replicated.

The only collectives on the data path are a broadcast of the mesh (24 MB of expanded
records for 500 k triangles) and an all-gather of the per-shard results (8 bytes per ray);
the sweep itself needs no exchange. DBSCAN / kNN / skeleton / RANSAC do
not shard: replicas only.

Two ways to run it, both with the same shard arithmetic (:func:`shard_bounds`):

* one process, several GPUs: ``hip.cast_rays_multi`` / ``cast_rays(..., n_devices=N)`` —
  RCCL inside the library (``ncclCommInitAll``);
* one process per GPU (torchrun-style): :class:`NativeComm` — an RCCL communicator inside the
  library per process, its id shipped through a rendezvous the caller provides (a private file
  here, or any object with ``broadcast_object_list`` such as a CPU ``gloo`` group); every byte
  of the data path moves through ``pyqsm_comm_*``.

Nothing here imports torch. With ``PYQSM_MULTI_FAKE_RANKS=N`` both run N logical ranks on one
GPU (``csrc/multi.hip``): how the one-GPU test boxes execute the N > 1 orchestration.
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
    def from_env(cls, device=None, directory: str | None = None, timeout_s: float = 120.0):
        """File rendezvous for single-node launches: rank 0 writes the id into a file that only
        this user can have created, the other ranks wait for it.

        The file lives in ``directory`` (default: ``$XDG_RUNTIME_DIR``, else the temp directory),
        is named after the launcher's pid, MASTER_PORT and TORCHELASTIC_RUN_ID, is created with
        ``O_CREAT | O_EXCL | O_NOFOLLOW`` and mode 0600 (a planted symlink or a foreign file makes
        rank 0 fail instead of writing through it), and carries — after the 128 id bytes — the
        launcher's start time as a nonce, so a file left behind by an earlier launch with the
        same pid and port is never accepted. Prefer :meth:`from_torch` when a process group exists."""
        import os
        import struct
        import tempfile
        import time
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", str(rank)))
        if directory is None:
            directory = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
        run_id = "".join(ch for ch in os.environ.get("TORCHELASTIC_RUN_ID", "") if ch.isalnum())[:32]
        path = os.path.join(directory, "pyqsm_comm_%d_%s_%s.id" % (
            os.getppid(), os.environ.get("MASTER_PORT", "0"), run_id or "norun"))
        nonce = struct.pack("<d", _process_start_time(os.getppid()))
        if rank == 0:
            tmp = "%s.%d.tmp" % (path, os.getpid())
            for stale in (tmp, path):
                try:
                    os.unlink(stale)                     # removes a link, never its target
                except FileNotFoundError:
                    pass
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | os.O_NOFOLLOW, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(cls.new_id() + nonce)
            os.replace(tmp, path)                       # atomic: readers never see a partial id
        t0 = time.time()
        while True:
            id_bytes = _read_private(path, 128 + len(nonce))
            if id_bytes is not None and id_bytes[128:] == nonce:
                id_bytes = id_bytes[:128]
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


def _process_start_time(pid=None) -> float:
    """Start of process `pid` (default: this one) in seconds since boot; 0.0 when /proc cannot say."""
    import os
    try:
        with open("/proc/%s/stat" % ("self" if pid is None else int(pid))) as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])
        return ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return 0.0


def _read_private(path: str, nbytes: int):
    """The first `nbytes` of a regular file owned by this user that nobody else can write, opened
    without following links; None when it is not (yet) there or not like that."""
    import os
    import stat
    try:
        fd = os.open(path, os.O_RDONLY | os.O_NOFOLLOW)
    except OSError:
        return None
    try:
        st = os.fstat(fd)
        if not stat.S_ISREG(st.st_mode) or st.st_uid != os.getuid() or st.st_mode & 0o022 \
                or st.st_size != nbytes:
            return None
        return os.read(fd, nbytes)
    finally:
        os.close(fd)


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
        # everything that can fail for local reasons (a triangle index outside the vertices on
        # rank 0, a wrong shard size, no memory) comes BEFORE the first collective, and the ranks
        # agree on "everybody can" first: a rank that raised here would otherwise leave the others
        # waiting in the broadcast for good (the same hand-shake as pyqsm_cast_rays_multi)
        err = None
        try:
            if comm.rank == 0:
                self.mesh = hip.DeviceMesh(verts, tris, dev)
            else:                                            # records arrive over RCCL
                self.mesh = hip.DeviceMesh.__new__(hip.DeviceMesh)
                self.mesh.device, self.mesh.n_tris = dev, T
                self.mesh.records = hip.DeviceBuffer(max(1, T) * 48, dev)
            rays_shard = np.ascontiguousarray(rays_shard, dtype=np.float32).reshape(-1, 6)
            if rays_shard.shape[0] != self.sizes[comm.rank]:
                raise ValueError("rays_shard does not have this rank's shard size")
            self.n_local = rays_shard.shape[0]
            self.d_rays = hip.DeviceBuffer.from_array(rays_shard, dev) if self.n_local else None
            # one block of [t(cap) | prim(cap)] 32-bit words per rank
            self.block = hip.DeviceBuffer(max(1, comm.world * 2 * self.cap * 4), dev)
        except Exception as exc:                             # noqa: BLE001 - re-raised below
            err = exc
        if comm.max_over_ranks(0.0 if err is None else 1.0) != 0.0:
            if err is not None:
                raise err
            raise RuntimeError("ShardedSweep: the setup failed on another rank")
        comm.broadcast(self.mesh.records, T * 48, root=0)
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
