"""Parity of the HIP ray x triangle sweep with the CPU oracle (bit-exact: the
oracle restates the same operation sequence) and with analytic known answers."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth

pytestmark = pytest.mark.gpu


def _random_rays(rng, n, lo, hi):
    o = rng.uniform(lo - 2, hi + 2, (n, 3))
    tgt = rng.uniform(lo, hi, (n, 3))
    d = tgt - o
    return np.concatenate([o, d], axis=1).astype(np.float32)


def _check(verts, tris, rays, gpu):
    t, p, uv = hip.cast_rays(verts, tris, rays, device=gpu)
    t0, p0, uv0 = oracle.cast_rays(verts, tris, rays)
    assert np.array_equal(p, p0)
    assert np.array_equal(t, t0)                       # bit-exact (includes +inf misses)
    assert np.array_equal(uv, uv0)
    return t, p, uv


@pytest.mark.parametrize("n_rays", [1, 63, 64, 65, 1000, 40_000])
def test_random_leaves_vs_oracle(gpu, n_rays):
    verts, tris = synth.canopy_mesh(2000, seed=5, side=0.4)
    rng = np.random.default_rng(n_rays)
    rays = _random_rays(rng, n_rays, verts.min(0), verts.max(0))
    t, p, _ = _check(verts, tris, rays, gpu)
    if n_rays >= 1000:
        assert np.isfinite(t).sum() > 10               # the test exercises real hits


def test_sun_rays_all_kernel_widths(gpu):
    """Large enough batches to take the 4-, 2- and 1-pair-per-lane kernels."""
    verts, tris = synth.canopy_mesh(600, seed=2, side=0.5)
    for n in (200_000, 600_000, 1_200_000):
        rays = synth.sun_rays(verts, n)
        _check(verts, tris, rays, gpu)


def test_parallel_rays_brute_force_and_culled_agree(gpu, monkeypatch):
    """The same batch through the cluster-culled sweep (default for one-direction
    batches) and through the plain brute-force parallel kernel (PYQSM_RAY_CULL=0):
    both must equal the oracle bit for bit."""
    verts, tris = synth.canopy_mesh(20_000, seed=4, side=0.15)
    rays = synth.sun_rays(verts, 300_000, elevation_deg=35.0, azimuth_deg=20.0)
    t0, p0, uv0 = oracle.cast_rays(verts, tris, rays)
    for flag in ("1", "0"):
        monkeypatch.setenv("PYQSM_RAY_CULL", flag)
        t, p, uv = hip.cast_rays(verts, tris, rays, device=gpu)
        assert np.array_equal(t, t0) and np.array_equal(p, p0) and np.array_equal(uv, uv0), flag
    assert np.isfinite(t0).mean() > 0.2


@pytest.mark.parametrize("eye", [(0.0, 0.0, 25.0),     # above the canopy, looking down
                                 (0.5, -0.3, 9.0)])     # INSIDE the canopy: triangles all around,
                                                        # behind the eye and across the eye plane
def test_pinhole_rays_brute_force_and_culled_agree(gpu, monkeypatch, eye):
    """Common-origin batches (the camera of cast_rays, ray_casting.py:269-279) through the
    image-space culled sweep and through the plain brute-force kernel (PYQSM_RAY_CULL=0):
    both must equal the oracle bit for bit (t, primitive id, uv)."""
    from pyqsm_amd.viz.ray_casting import create_rays_pinhole
    verts, tris = synth.canopy_mesh(20_000, seed=5, side=0.25)
    rays = create_rays_pinhole(90.0, (0.0, 0.0, 9.0) if eye[2] > 20 else (3.0, 2.0, 9.5), eye,
                               (0, 1, -1), 320, 240).reshape(-1, 6)
    t0, p0, uv0 = oracle.cast_rays(verts, tris, rays)
    for flag in ("1", "0"):
        monkeypatch.setenv("PYQSM_RAY_CULL", flag)
        t, p, uv = hip.cast_rays(verts, tris, rays, device=gpu)
        assert np.array_equal(t, t0) and np.array_equal(p, p0) and np.array_equal(uv, uv0), flag
    assert 0.05 < np.isfinite(t0).mean() < 1.0


def test_common_origin_but_all_around_falls_back(gpu):
    """One origin, directions over the whole sphere: no single image plane, so the batch takes
    the brute-force kernel; results still equal the oracle."""
    verts, tris = synth.canopy_mesh(5_000, seed=6, side=0.3)
    rng = np.random.default_rng(0)
    d = rng.normal(size=(20_000, 3)).astype(np.float32)
    rays = np.concatenate([np.tile(np.float32([0.2, 0.1, 9.0]), (len(d), 1)), d], 1)
    _check(verts, tris, rays, gpu)


def test_unit_triangle_known_answers(gpu):
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    tris = np.array([[0, 1, 2]], dtype=np.int32)
    rays = np.array([
        [0.25, 0.25, 1, 0, 0, -1],     # hit, t = 1, (u,v) = (.25,.25)
        [0.25, 0.25, 1, 0, 0, -2],     # non-unit direction: t = 0.5
        [0.25, 0.25, -1, 0, 0, 1],     # from below (back face): still a hit
        [0.75, 0.75, 1, 0, 0, -1],     # outside (u+v > 1)
        [0.25, 0.25, 1, 0, 0, 1],      # pointing away: t < 0 -> miss
        [0.25, 0.25, 1, 1, 0, 0],      # parallel to the plane
        [0.0, 0.0, 1, 0, 0, -1],       # exactly through vertex v0
        [0.5, 0.0, 1, 0, 0, -1],       # exactly on edge v0-v1
    ], dtype=np.float32)
    t, p, uv = hip.cast_rays(verts, tris, rays, device=gpu)
    assert t[0] == 1.0 and p[0] == 0 and np.allclose(uv[0], [0.25, 0.25])
    assert t[1] == 0.5
    assert t[2] == 1.0 and p[2] == 0
    assert np.isinf(t[3]) and p[3] == 0xFFFFFFFF
    assert np.isinf(t[4]) and np.isinf(t[5])
    assert t[6] == 1.0 and t[7] == 1.0                 # boundaries are inclusive
    # barycentric convention of ray_casting.py:172-180
    hit = (1 - uv[0, 0] - uv[0, 1]) * verts[0] + uv[0, 0] * verts[1] + uv[0, 1] * verts[2]
    assert np.allclose(hit, [0.25, 0.25, 0])


def test_closest_of_stack_and_tie_break(gpu):
    # three parallel quads at z = 1, 2, 3 and a duplicate of the z = 1 quad
    quads = []
    for z in (3.0, 1.0, 2.0, 1.0):
        quads.append([[0, 0, z], [1, 0, z], [1, 1, z], [0, 1, z]])
    verts = np.array(quads, dtype=np.float32).reshape(-1, 3)
    tris = np.array([[4 * q + a, 4 * q + b, 4 * q + c] for q in range(4)
                     for (a, b, c) in ((0, 1, 2), (0, 2, 3))], dtype=np.int32)
    rays = np.array([[0.75, 0.25, 10, 0, 0, -1], [0.25, 0.75, 10, 0, 0, -1]], dtype=np.float32)
    t, p, _ = hip.cast_rays(verts, tris, rays, device=gpu)
    assert np.all(t == 7.0)                            # z = 3 quad is nearest from above
    rays[:, 2] = -10
    rays[:, 5] = 1
    t, p, _ = hip.cast_rays(verts, tris, rays, device=gpu)
    assert np.all(t == 11.0)
    assert list(p) == [2, 3]                           # lowest triangle id among the tie


def test_empty_inputs(gpu):
    verts, tris = synth.canopy_mesh(10, seed=0)
    t, p, uv = hip.cast_rays(verts, tris, np.zeros((0, 6), np.float32), device=gpu)
    assert t.shape == (0,) and p.shape == (0,) and uv.shape == (0, 2)
    rays = synth.sun_rays(verts, 100)
    t, p, _ = hip.cast_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32), rays,
                            device=gpu)
    assert np.all(np.isinf(t)) and np.all(p == 0xFFFFFFFF)


def test_image_shaped_rays(gpu):
    verts, tris = synth.canopy_mesh(500, seed=3, side=0.5)
    rays = synth.sun_rays(verts, 48 * 64).reshape(48, 64, 6)
    t, p, uv = hip.cast_rays(verts, tris, rays, device=gpu)
    assert t.shape == (48, 64) and p.shape == (48, 64) and uv.shape == (48, 64, 2)


def test_bad_triangle_index_is_an_error(gpu):
    from pyqsm_amd._lib import PyQSMHipError
    verts = np.zeros((3, 3), np.float32)
    tris = np.array([[0, 1, 7]], dtype=np.int32)
    with pytest.raises(PyQSMHipError):
        hip.cast_rays(verts, tris, np.zeros((4, 6), np.float32), device=gpu)


def test_list_intersections_vs_oracle(gpu):
    verts, tris = synth.canopy_mesh(3000, seed=9, side=0.6)
    rays = synth.sun_rays(verts, 2500)
    got = hip.list_intersections(verts, tris, rays, device=gpu)
    ref = oracle.list_intersections(verts, tris, rays)
    for k in ("counts", "ray_ids", "primitive_ids", "t_hit", "primitive_uvs"):
        assert np.array_equal(got[k], ref[k]), k
    assert got["counts"].max() >= 2                    # some rays cross several leaves
    # closest hit == minimum over the list
    t, _, _ = hip.cast_rays(verts, tris, rays, device=gpu)
    for r in np.flatnonzero(got["counts"])[:50]:
        assert t[r] == got["t_hit"][got["ray_ids"] == r].min()


def test_randomised_cameras_and_sun_angles(gpu):
    """Twelve random pinhole cameras (eye inside or outside the canopy, various fields of
    view) and six random sun directions over a small canopy: both culled paths against the
    oracle, bit for bit."""
    from pyqsm_amd.viz.ray_casting import create_rays_pinhole
    rng = np.random.default_rng(77)
    verts, tris = synth.canopy_mesh(6_000, seed=8, side=0.3)
    lo, hi = verts.min(0), verts.max(0)
    for case in range(12):
        eye = rng.uniform(lo - 3, hi + 3)
        center = rng.uniform(lo, hi)
        up = rng.normal(size=3)
        fov = float(rng.choice([20.0, 60.0, 90.0, 120.0]))
        rays = create_rays_pinhole(fov, center, eye, up, 96, 64).reshape(-1, 6)
        _check(verts, tris, rays, gpu)
    for case in range(6):
        rays = synth.sun_rays(verts, 30_000, elevation_deg=float(rng.uniform(5, 89)),
                              azimuth_deg=float(rng.uniform(0, 360)))
        _check(verts, tris, rays, gpu)


def test_interception_layers_peel_a_canopy(gpu):
    """Cast, remove what was hit first, repeat (data/notes/methods.md:53-55): rounds, areas and
    the round of every triangle equal the CPU restatement; a stack of three parallel sheets
    under vertical rays comes off sheet by sheet."""
    from pyqsm_amd.viz.ray_casting import interception_layers

    verts, tris = synth.canopy_mesh(6000)
    rays = synth.sun_rays(verts, 40_000, elevation_deg=60.0, azimuth_deg=135.0)
    areas, layer = interception_layers((verts, tris), rays, device=gpu)
    want_areas, want_layer = oracle.interception_layers(verts, tris, rays)
    assert np.array_equal(layer, want_layer)
    assert areas == want_areas
    assert len(areas) >= 3 and (layer == 0).sum() > (layer == 2).sum() > 0

    # three unit squares (two triangles each) at z = 0, 1, 2; rays straight down from z = 5
    sq = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float32)
    v = np.concatenate([np.c_[sq, np.full(4, z, np.float32)] for z in (0.0, 1.0, 2.0)])
    t = np.concatenate([np.array([[0, 1, 2], [0, 2, 3]]) + 4 * k for k in range(3)]).astype(np.int32)
    gx, gy = np.meshgrid(np.linspace(0.05, 0.95, 10), np.linspace(0.05, 0.95, 10))
    r = np.zeros((100, 6), dtype=np.float32)
    r[:, 0], r[:, 1], r[:, 2], r[:, 5] = gx.ravel(), gy.ravel(), 5.0, -1.0
    areas, layer = interception_layers((v, t), r, device=gpu)
    assert areas == [1.0, 1.0, 1.0]
    assert layer.tolist() == [2, 2, 1, 1, 0, 0]
    areas, layer = interception_layers((v, t), r, max_rounds=1, device=gpu)
    assert areas == [1.0] and layer.tolist() == [-1, -1, -1, -1, 0, 0]
