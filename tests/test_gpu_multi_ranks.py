"""The N > 1 orchestration of csrc/multi.hip, executed on the ONE GPU a test box has.

RCCL refuses two ranks on one device, so PYQSM_MULTI_FAKE_RANKS=N makes the library run N
logical ranks — each with its own host thread, context (stream + arena), shard and place in
the result blocks — on device 0, with ONLY the three collectives replaced (host barrier +
device-to-device copies behind the same call sites). Everything else is the code the driver's
8-GPU node runs: the per-device threads and their rendezvous, the ragged shards and `cap`
padding, the in-place all-gather offsets, the "one rank fails -> nobody enters a collective"
hand-shake and the "a collective failed -> everybody leaves" hand-shake.
Replaces scene.cast_rays at pyQSM/viz/ray_casting.py:275-279 on more than one GPU."""
import threading

import numpy as np
import pytest

from pyqsm_amd import _lib, hip, synth
from pyqsm_amd.parallel import NativeComm, ShardedSweep, shard_bounds

pytestmark = pytest.mark.gpu


def _general_rays(verts, R, seed=3):
    rng = np.random.default_rng(seed)
    o = verts.mean(0) + rng.normal(0, 6.0, (R, 3)).astype(np.float32)
    target = verts[rng.integers(0, len(verts), R)]
    return np.concatenate([o, target - o], 1).astype(np.float32)


@pytest.fixture()
def mesh():
    return synth.canopy_mesh(6000, seed=2, side=0.4)


@pytest.fixture()
def logical(monkeypatch):
    def set_ranks(n):
        monkeypatch.setenv("PYQSM_MULTI_FAKE_RANKS", str(n))
    return set_ranks


@pytest.mark.parametrize("N", [2, 3, 8])
@pytest.mark.parametrize("R", [1, 7, 1001, 1_000_000])
def test_one_process_n_logical_devices_equal_single_call(gpu, mesh, logical, N, R):
    verts, tris = mesh
    logical(N)
    for kind, rays in (("sun", synth.sun_rays(verts, R)), ("general", _general_rays(verts, R))):
        if kind == "general" and R > 100_000:
            rays = rays[:100_000]                    # the general kernel is brute force: keep it short
        t0, p0, uv0 = hip.cast_rays(verts, tris, rays, device=gpu)
        t1, p1, uv1 = hip.cast_rays_multi(verts, tris, rays, n_devices=N)
        assert np.array_equal(t0, t1) and np.array_equal(p0, p1) and np.array_equal(uv0, uv1), (kind, N, R)
        t2, p2, uv2 = hip.cast_rays_multi(verts, tris, rays, n_devices=N, with_uv=False)
        assert uv2 is None and np.array_equal(t0, t2) and np.array_equal(p0, p2), (kind, N, R)
    t3, p3, _ = hip.cast_rays_multi(verts, tris, synth.sun_rays(verts, R), n_devices=0)   # "every GPU" = N
    t0, p0, _ = hip.cast_rays(verts, tris, synth.sun_rays(verts, R), device=gpu)
    assert np.array_equal(t0, t3) and np.array_equal(p0, p3)


def test_more_devices_than_the_logical_box_has_is_refused(gpu, mesh, logical):
    verts, tris = mesh
    logical(3)
    with pytest.raises(_lib.PyQSMHipError):
        hip.cast_rays_multi(verts, tris, synth.sun_rays(verts, 100), n_devices=4)


@pytest.mark.parametrize("N", [2, 8])
def test_a_rank_that_fails_locally_fails_the_call_without_a_hang(gpu, mesh, logical, N):
    """Phase 1 of rank 0 rejects a triangle index outside the vertices: no rank may enter the
    broadcast, the call returns the error, and the library works afterwards."""
    verts, tris = mesh
    logical(N)
    rays = synth.sun_rays(verts, 5000)
    bad = tris.copy()
    bad[5, 1] = len(verts) + 3
    with pytest.raises(_lib.PyQSMHipError, match="device 0"):
        hip.cast_rays_multi(verts, bad, rays, n_devices=N)
    t, p, uv = hip.cast_rays_multi(verts, tris, rays, n_devices=N)
    t0, p0, uv0 = hip.cast_rays(verts, tris, rays, device=gpu)
    assert np.array_equal(t, t0) and np.array_equal(p, p0) and np.array_equal(uv, uv0)


@pytest.mark.parametrize("where", ["1,1", "2,2", "0,2"])
def test_a_collective_that_fails_on_one_rank_ends_the_call_on_all(gpu, mesh, logical, monkeypatch, where):
    """The failure hand-shake after the rendezvous: rank r's broadcast (op 1) or
    all-gather (op 2) fails as if it could not be enqueued; every thread must leave, the call
    returns an error, the next call works."""
    verts, tris = mesh
    logical(4)
    rays = synth.sun_rays(verts, 4001)
    monkeypatch.setenv("PYQSM_MULTI_INJECT_FAIL", where)
    with pytest.raises(_lib.PyQSMHipError):
        hip.cast_rays_multi(verts, tris, rays, n_devices=4)
    monkeypatch.delenv("PYQSM_MULTI_INJECT_FAIL")
    t, p, _ = hip.cast_rays_multi(verts, tris, rays, n_devices=4)
    t0, p0, _ = hip.cast_rays(verts, tris, rays, device=gpu)
    assert np.array_equal(t, t0) and np.array_equal(p, p0)


# ---- pyqsm_comm_*: one "process" (here: host thread) per rank ------------------------------

def _run_ranks(world, body):
    """body(rank, comm) on `world` threads, each with its own communicator of one logical world;
    returns the per-rank results, re-raising the first exception."""
    ident = NativeComm.new_id()
    out, errs = [None] * world, [None] * world

    def run(rank):
        comm = None
        try:
            comm = NativeComm(ident, world, rank, 0)
            out[rank] = body(rank, comm)
        except BaseException as exc:                 # noqa: BLE001 - reported by the main thread
            errs[rank] = exc
        finally:
            if comm is not None:
                comm.close()

    threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a logical rank hangs"
    return out, errs


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("R", [1, 7, 1001, 300_000])
def test_one_communicator_per_rank_sharded_sweep(gpu, mesh, logical, world, R):
    """What bench.py --gpus N does per rank (ShardedSweep over a NativeComm), for ragged shards
    and for fewer rays than ranks."""
    verts, tris = mesh
    logical(world)
    rays = synth.sun_rays(verts, R)
    t0, p0, _ = hip.cast_rays(verts, tris, rays, device=gpu)

    def body(rank, comm):
        assert (comm.world, comm.rank) == (world, rank)
        assert comm.max_over_ranks(float(rank)) == float(world - 1)
        b, e = shard_bounds(R, world, rank)
        # only rank 0 owns the mesh; the others receive the expanded records
        sweep = ShardedSweep(comm, verts if rank == 0 else None, tris, rays[b:e], R)
        sweep.run()
        sweep.run()                                   # the blocks are reused call after call
        comm.barrier()
        return sweep.results()

    out, errs = _run_ranks(world, body)
    assert errs == [None] * world, errs
    for t, p in out:                                  # every rank holds the result for ALL rays
        assert np.array_equal(t, t0) and np.array_equal(p, p0)


def test_setup_failure_on_rank_zero_raises_on_every_rank(gpu, mesh, logical):
    verts, tris = mesh
    logical(3)
    rays = synth.sun_rays(verts, 900)
    bad = tris.copy()
    bad[7, 2] = -1

    def body(rank, comm):
        b, e = shard_bounds(len(rays), comm.world, rank)
        ShardedSweep(comm, verts, bad, rays[b:e], len(rays))

    out, errs = _run_ranks(3, body)
    assert isinstance(errs[0], _lib.PyQSMHipError)
    assert all(isinstance(e, RuntimeError) for e in errs[1:]), errs
    # and the threads' contexts are usable afterwards
    t, p, _ = hip.cast_rays(verts, tris, rays, device=gpu)
    assert t.shape == (900,)


def test_broadcast_and_gather_primitives_between_ranks(gpu, logical):
    """pyqsm_comm_broadcast_dev from a root other than 0, and an out-of-place all-gather."""
    logical(4)
    nbytes = 4096 + 12

    def body(rank, comm):
        payload = np.full(nbytes, rank + 1, np.uint8)
        buf = hip.DeviceBuffer.from_array(payload, 0)
        comm.broadcast(buf, nbytes, root=2)
        got = buf.download((nbytes,), np.uint8)
        send = hip.DeviceBuffer.from_array(np.full(100, 10 * rank, np.uint8), 0)
        recv = hip.DeviceBuffer(100 * comm.world, 0)
        comm.all_gather(send.ptr, recv, 100)
        hip.sync(0)
        return got, recv.download((comm.world, 100), np.uint8)

    out, errs = _run_ranks(4, body)
    assert errs == [None] * 4, errs
    for got, gathered in out:
        assert np.all(got == 3)
        assert np.array_equal(gathered, np.repeat(np.arange(4, dtype=np.uint8)[:, None] * 10, 100, 1))
