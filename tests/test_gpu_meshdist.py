"""Point-to-mesh distance (a10: `mri`, ray_casting.py:237-260 -> compute_signed_distance):
HIP kernel against the CPU oracle (bit-exact: same fp32 operation sequence) and analytic cases.
Open3D is not installable, so parity with its implementation is unpinned."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.viz.ray_casting import RaycastingScene, mri

pytestmark = pytest.mark.gpu

CUBE_V = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
CUBE_T = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7],
                   [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)


@pytest.mark.parametrize("n_tris,n_q", [(1, 1), (7, 63), (2000, 5000), (20_000, 40_000)])
def test_matches_oracle_bit_for_bit(gpu, n_tris, n_q):
    verts, tris = synth.canopy_mesh(max(n_tris, 2), seed=n_tris, side=0.4)
    tris = tris[:n_tris]
    rng = np.random.default_rng(n_q)
    q = rng.uniform(verts.min(0) - 1, verts.max(0) + 1, (n_q, 3)).astype(np.float32)
    q[: min(n_q, len(verts))] = verts[: min(n_q, len(verts))]          # some queries ON vertices
    d, p = hip.point_mesh_distance(verts, tris, q, device=gpu)
    d0, p0 = oracle.point_mesh_distance(verts, tris, q)
    assert np.array_equal(d, d0) and np.array_equal(p, p0)
    assert d[: min(n_q, 3 * n_tris)].min() == 0.0


def test_cube_known_answers_and_sign(gpu):
    scene = RaycastingScene(gpu)
    scene.add_triangles((CUBE_V, CUBE_T))
    # (y != z: the +x parity ray of a query with y == z runs through the diagonal shared by the
    #  two triangles of the x = 1 face and is counted twice, as with any edge-inclusive test)
    q = np.array([[0.5, 0.375, 0.5], [0.5, 0.375, 2.0], [2, 2, 2], [0.5, 0.375, 0.875], [-1, 0.375, 0.5],
                  [0.25, 0.375, 0.5]], np.float32)
    d = scene.compute_distance(q)
    assert np.allclose(d, [0.375, 1.0, np.sqrt(3), 0.125, 1.0, 0.25], rtol=1e-6)
    sd = scene.compute_signed_distance(q)
    assert np.allclose(sd, [-0.375, 1.0, np.sqrt(3), -0.125, 1.0, -0.25], rtol=1e-6)
    # shapes follow the input (Open3D convention)
    lattice = np.zeros((4, 5, 6, 3), np.float32) + 0.5
    assert scene.compute_signed_distance(lattice).shape == (4, 5, 6)


def test_mri_returns_the_fields_the_reference_plots(gpu):
    pts, sd, lattice, sd_grid = mri((CUBE_V, CUBE_T), grid=16, device=gpu)
    assert pts.shape == (256, 3) and sd.shape == (256,)
    assert (sd <= 0).mean() > 0.95            # bounding box of the cube = the cube (bar parity-ray edge cases)
    assert lattice.shape == (16, 16, 16, 3) and sd_grid.shape == (16, 16, 16)
    assert abs(sd_grid).max() <= 0.5 + 1e-6


def test_empty_mesh_and_bad_indices(gpu):
    from pyqsm_amd._lib import PyQSMHipError
    d, p = hip.point_mesh_distance(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32),
                                   np.zeros((3, 3), np.float32), device=gpu)
    assert np.isinf(d).all() and (p == 0xFFFFFFFF).all()
    with pytest.raises(PyQSMHipError):
        hip.point_mesh_distance(CUBE_V, np.array([[0, 1, 99]], np.int32), np.zeros((1, 3), np.float32),
                                device=gpu)
