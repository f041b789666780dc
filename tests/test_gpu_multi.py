"""The RCCL paths behind the C-ABI on the one GPU a test box has: pyqsm_cast_rays_multi
(ncclCommInitAll, n_devices = 1 still goes through ncclAllGather) and the one-process-per-GPU
communicator (pyqsm_comm_*) with a single rank. Both must reproduce pyqsm_cast_rays bit for
bit; shards over real N > 1 are covered by the shard arithmetic (test_host_logic.py), the
2-rank gloo test, and the driver's scaling run (replaces scene.cast_rays at
pyQSM/viz/ray_casting.py:275-279)."""
import numpy as np
import pytest

from pyqsm_amd import _lib, hip, synth
from pyqsm_amd.parallel import NativeComm, ShardedSweep
from pyqsm_amd.viz import ray_casting as rc

pytestmark = pytest.mark.gpu


def _general_rays(verts, R, seed=3):
    rng = np.random.default_rng(seed)
    o = verts.mean(0) + rng.normal(0, 6.0, (R, 3)).astype(np.float32)
    target = verts[rng.integers(0, len(verts), R)]
    return np.concatenate([o, target - o], 1).astype(np.float32)


@pytest.mark.parametrize("R", [20_000, 1001, 1])
def test_multi_one_device_equals_single(gpu, R):
    verts, tris = synth.canopy_mesh(6000, seed=2, side=0.4)
    for rays in (synth.sun_rays(verts, R), _general_rays(verts, R)):
        t0, p0, uv0 = hip.cast_rays(verts, tris, rays, device=gpu)
        t1, p1, uv1 = hip.cast_rays_multi(verts, tris, rays, n_devices=1)
        assert np.array_equal(t0, t1) and np.array_equal(p0, p1) and np.array_equal(uv0, uv1)
        t2, p2, _ = hip.cast_rays_multi(verts, tris, rays, n_devices=0)       # every visible GPU
        assert np.array_equal(t0, t2) and np.array_equal(p0, p2)


def test_multi_pinhole_wrapper_and_edge_cases(gpu):
    verts, tris = synth.canopy_mesh(4000, seed=8, side=0.5)
    a = rc.cast_rays((verts, tris))
    b = rc.cast_rays((verts, tris), n_devices=1)
    assert np.array_equal(a["t_hit"], b["t_hit"]) and np.array_equal(a["primitive_ids"], b["primitive_ids"])
    # empty mesh: every ray misses; empty ray set: nothing to do
    rays = synth.sun_rays(verts, 100)
    t, p, _ = hip.cast_rays_multi(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), rays, 1)
    assert np.all(np.isinf(t)) and np.all(p == 0xFFFFFFFF)
    t, p, _ = hip.cast_rays_multi(verts, tris, np.zeros((0, 6), np.float32), 1)
    assert t.shape == (0,)
    with pytest.raises(_lib.PyQSMHipError):                                    # more than the box has
        hip.cast_rays_multi(verts, tris, rays, n_devices=_lib.device_count() + 1)
    bad = tris.copy()
    bad[5, 1] = len(verts) + 3                                                 # index outside the vertices
    with pytest.raises(_lib.PyQSMHipError):
        hip.cast_rays_multi(verts, bad, rays, n_devices=1)
    t, p, _ = hip.cast_rays_multi(verts, tris, rays, n_devices=1)              # and the library still works
    t0, p0, _ = hip.cast_rays(verts, tris, rays)
    assert np.array_equal(t, t0) and np.array_equal(p, p0)


def test_native_communicator_single_rank(gpu):
    comm = NativeComm(NativeComm.new_id(), world=1, rank=0, device=gpu)
    try:
        assert comm.max_over_ranks(3.25) == 3.25
        verts, tris = synth.canopy_mesh(5000, seed=5, side=0.4)
        rays = synth.sun_rays(verts, 12_345)
        sweep = ShardedSweep(comm, verts, tris, rays, len(rays))
        sweep.run()
        t, p = sweep.results()
        t0, p0, _ = hip.cast_rays(verts, tris, rays, device=gpu)
        assert np.array_equal(t, t0) and np.array_equal(p, p0)
        # a second communicator in the same process is refused
        with pytest.raises(_lib.PyQSMHipError):
            NativeComm(NativeComm.new_id(), world=1, rank=0, device=gpu)
    finally:
        comm.close()
    # after close a new one can be made
    comm = NativeComm(NativeComm.new_id(), world=1, rank=0, device=gpu)
    comm.barrier()
    comm.close()
