"""Parity of the HIP DBSCAN with the CPU oracle and the scikit-learn fixtures
(labels and core set bit-exact)."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth

pytestmark = pytest.mark.gpu


def _check(points, eps, min_pts, gpu):
    lab, core = hip.dbscan(points, eps, min_pts, device=gpu)
    lab0, core0 = oracle.dbscan(points, eps, min_pts)
    assert np.array_equal(core, core0)
    assert np.array_equal(lab, lab0)
    return lab, core


def test_tree_50k(gpu):
    P = synth.forest(50_000)
    lab, core = _check(P, 0.1, 10, gpu)
    assert lab.max() == 0 and (lab == -1).sum() > 100


@pytest.mark.parametrize("seed,eps,min_pts", [(1, 0.02, 5), (2, 0.03, 8), (3, 0.015, 3),
                                              (4, 0.05, 40)])
def test_random_slab_many_clusters(gpu, seed, eps, min_pts):
    rng = np.random.default_rng(seed)
    Q = rng.uniform(0, 1, (20_000, 3)).astype(np.float32).astype(np.float64)
    Q[:, 2] *= 0.2
    lab, core = _check(Q, eps, min_pts, gpu)
    assert (~core & (lab >= 0)).sum() > 0 or lab.max() < 1   # border points exist


def test_forest_200k(gpu):
    P = synth.forest(200_000)
    lab, _ = _check(P, 0.1, 10, gpu)
    assert lab.max() + 1 == 4


def test_edge_cases(gpu):
    lab, core = hip.dbscan(np.zeros((0, 3)), 0.1, 10, device=gpu)
    assert lab.shape == (0,) and core.shape == (0,)
    one = np.array([[1.0, 2.0, 3.0]])
    lab, core = hip.dbscan(one, 0.1, 1, device=gpu)
    assert lab[0] == 0 and core[0]
    lab, core = hip.dbscan(one, 0.1, 2, device=gpu)
    assert lab[0] == -1 and not core[0]
    # duplicates + exact-eps spacing (inclusive compare) on a line
    line = np.array([[0.125 * i, 0, 0] for i in range(9)] + [[0, 0, 0]] * 3)
    _check(line, 0.125, 3, gpu)
    _check(line, 0.125, 2, gpu)
    # negative coordinates and a far outlier
    rng = np.random.default_rng(0)
    P = np.concatenate([rng.normal(-50, 0.05, (500, 3)), [[1000.0, -1000.0, 3.0]]])
    _check(P, 0.05, 5, gpu)


def test_non_finite_is_an_error(gpu):
    from pyqsm_amd._lib import PyQSMHipError
    P = np.zeros((10, 3))
    P[3, 1] = np.nan
    with pytest.raises(PyQSMHipError):
        hip.dbscan(P, 0.1, 3, device=gpu)
