"""HIP point-cloud Laplacian and the whole extract_skeleton loop against the CPU
oracle (the only parity available: robust_laplacian is not installable)."""
import numpy as np
import pytest
from scipy.sparse import csr_matrix

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu


def _gpu_lap(P, k, moll, gpu):
    (indptr, indices, data), mass = hip.pc_laplacian(P, k, moll, device=gpu)
    n = len(P)
    return csr_matrix((data, indices, indptr), shape=(n, n)), mass


@pytest.mark.parametrize("n,k", [(3000, 20), (20_000, 20), (5000, 30), (800, 8)])
def test_matches_oracle(gpu, n, k):
    P = synth.forest(n, seed=n)
    L, M = _gpu_lap(P, k, 1e-6, gpu)
    L0, M0 = oracle.point_cloud_laplacian(P, k, 1e-6)
    assert np.array_equal(L.indptr, L0.indptr) and np.array_equal(L.indices, L0.indices)
    scale = abs(L0.data).max()
    assert abs(L.data - L0.data).max() <= 1e-9 * scale        # bound: 1e-5 rel (north_star)
    assert abs(M - M0).max() <= 1e-9 * M0.max()
    assert abs(L - L.T).max() == 0.0


def test_degenerate_grid_same_decisions(gpu):
    g = np.stack(np.meshgrid(np.arange(20.0), np.arange(20.0)), -1).reshape(-1, 2)
    P = np.concatenate([g, np.zeros((len(g), 1))], 1)         # exactly co-circular quads
    L, M = _gpu_lap(P, 12, 1e-6, gpu)
    L0, M0 = oracle.point_cloud_laplacian(P, 12, 1e-6)
    assert np.array_equal(L.indices, L0.indices) and np.allclose(L.data, L0.data, atol=1e-12)


def test_tiny_and_empty(gpu):
    (indptr, indices, data), mass = hip.pc_laplacian(np.zeros((0, 3)), 20, 1e-6, device=gpu)
    assert list(indptr) == [0] and len(indices) == 0
    P = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.1]])
    L, M = _gpu_lap(P, 3, 1e-6, gpu)
    L0, M0 = oracle.point_cloud_laplacian(P, 3, 1e-6)
    assert np.array_equal(L.indices, L0.indices) and np.allclose(L.data, L0.data)


def test_extract_skeleton_end_to_end(gpu):
    """configs[2] in miniature: the full loop (HIP Laplacian + HIP solve + HIP clamp)
    against the oracle loop (oracle Laplacian + SciPy spsolve), every step and the totals
    held to north_star's 1e-5 relative. (tests/test_gpu_config3.py pins the same bound per
    solve at init_contraction 7 and measures SuperLU's own error beside it.)"""
    P = synth.forest(2500, seed=9)
    got, total, steps = sk.extract_skeleton(P, max_iter=5, contraction_factor=3,
                                            attraction_factor=3, termination_ratio=0.0)
    lo, hi = sk.oriented_bounds(P)
    want, want_total, want_steps = oracle.extract_skeleton(
        P, lambda p: oracle.point_cloud_laplacian(p, 20, 1e-6), (lo, hi), max_iter=5,
        termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
    assert len(steps) == len(want_steps) == 5
    scale = np.abs(want).max()
    errs = [np.abs(a - b).max() / scale for a, b in zip(steps, want_steps)]
    print("per-step max rel diff", ["%.1e" % e for e in errs])
    assert max(errs) <= 1e-5
    assert np.abs(got.points - want).max() <= 1e-5 * scale
    assert np.abs(total - want_total).max() <= 1e-5 * scale
    assert all(s["ok"] for s in got.solve_log)
    # the cloud really contracted
    assert np.linalg.norm(total, axis=1).mean() > 0.01
