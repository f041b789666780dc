"""HIP kNN vs scipy cKDTree fixtures and the CPU oracle (squared distances bit-exact;
indices exact wherever distances are distinct)."""
import glob
import os

import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "knn_*.npz"))))
def test_against_ckdtree_fixture(gpu, path):
    g = np.load(path)
    idx, d2 = hip.knn(g["points"], int(g["k"]), True, device=gpu)
    assert np.array_equal(np.sqrt(d2), g["dist"])      # cKDTree returns sqrt of the same sum
    distinct = np.ones_like(idx, dtype=bool)
    distinct[:, 1:] &= d2[:, 1:] != d2[:, :-1]
    distinct[:, :-1] &= d2[:, :-1] != d2[:, 1:]
    assert np.array_equal(idx[distinct], g["idx"][distinct])


@pytest.mark.parametrize("n,k,excl", [(50_000, 20, True), (20_000, 8, False), (3000, 64, True),
                                      (700, 150, True)])
def test_against_oracle(gpu, n, k, excl):
    P = synth.forest(n, seed=2)
    idx, d2 = hip.knn(P, k, excl, device=gpu)
    idx0, d20 = oracle.knn(P, k, excl)
    assert np.array_equal(d2, d20)
    assert np.array_equal(idx, idx0)                   # ties broken by index on both sides


def test_small_and_degenerate(gpu):
    P = np.array([[0.0, 0, 0], [1, 0, 0], [0, 2, 0]])
    idx, d2 = hip.knn(P, 5, True, device=gpu)          # fewer points than k: padded
    assert list(idx[0]) == [1, 2, 3, 3, 3]
    assert list(d2[0][:2]) == [1.0, 4.0] and np.all(np.isinf(d2[0][2:]))
    same = np.zeros((40, 3)) + 0.5                     # all points identical
    idx, d2 = hip.knn(same, 3, True, device=gpu)
    assert np.all(d2 == 0)
    assert list(idx[0]) == [1, 2, 3] and list(idx[39]) == [0, 1, 2]
    idx, d2 = hip.knn(np.zeros((0, 3)), 4, True, device=gpu)
    assert idx.shape == (0, 4)


def test_outliers_far_from_everything(gpu):
    rng = np.random.default_rng(5)
    P = np.concatenate([rng.normal(0, 0.01, (5000, 3)), rng.uniform(-500, 500, (30, 3))])
    P = P.astype(np.float32).astype(np.float64)
    idx, d2 = hip.knn(P, 10, True, device=gpu)
    idx0, d20 = oracle.knn(P, 10, True)
    assert np.array_equal(d2, d20) and np.array_equal(idx, idx0)


def test_randomised_differential(gpu):
    """k on both sides of every register-list width (8/16/24/32) and beyond, with and without
    the query itself, on uniform / clustered / lattice (many exact ties) / planar clouds."""
    rng = np.random.default_rng(31)
    ks = [1, 2, 7, 8, 9, 16, 17, 20, 24, 25, 31, 32, 33, 48]
    for case, k in enumerate(ks * 2):
        n = int(rng.integers(k + 2, 6000))
        kind = case % 4
        if kind == 0:
            P = rng.uniform(0, 1, (n, 3))
        elif kind == 1:
            c = rng.uniform(0, 1, (5, 3))
            P = c[rng.integers(0, 5, n)] + rng.normal(0, 1, (n, 3)) * rng.choice([0.003, 0.03, 0.2], (n, 1))
        elif kind == 2:
            P = rng.integers(0, 9, (n, 3)) * 0.125          # duplicates and equidistant neighbours
        else:
            P = rng.uniform(0, 1, (n, 3)) * [1, 1, 0]
        P = P.astype(np.float32).astype(np.float64)
        excl = bool(case % 2)
        idx, d2 = hip.knn(P, k, excl, device=gpu)
        idx0, d20 = oracle.knn(P, k, excl)
        assert np.array_equal(d2, d20), (case, n, k, excl)
        assert np.array_equal(idx, idx0), (case, n, k, excl)


@pytest.mark.parametrize("k", [20, 21, 30])
def test_quantised_coordinates_have_ties_at_the_kth_place(gpu, k):
    """Coordinates on a 5 mm grid (a LAS file holds scaled integers): many squared distances coincide,
    some of them across the k-th place — the case the strict-order first pass of k_knn_reg hands
    to its exact second pass. Indices and distances equal the oracle's (ties by index)."""
    P = np.round(synth.forest(40_000, seed=12) * 200.0) / 200.0          # 5 mm grid
    idx, d2 = hip.knn(P, k, True, device=gpu)
    idx0, d20 = oracle.knn(P, k, True)
    tied = float((d20[:, 1:] == d20[:, :-1]).any(axis=1).mean())
    print(f"k={k}: {tied:.1%} of the queries have equal distances among their neighbours")
    assert tied > 0.01
    assert np.array_equal(d2, d20) and np.array_equal(idx, idx0)


def test_many_far_outliers_stay_exact(gpu):
    """5 % of the points uniform in a box fifty times the cloud's size (the dense grid cannot have
    the wanted cell edge, several retry levels run): indices and distances equal the oracle's,
    for the outliers' own queries too."""
    rng = np.random.default_rng(17)
    P = synth.forest(40_000, seed=21)
    ext = P.max(0) - P.min(0)
    far = rng.uniform(P.min(0) - 25 * ext, P.max(0) + 25 * ext, (2000, 3))
    X = np.concatenate([P, far])[rng.permutation(42_000)].astype(np.float32).astype(np.float64)
    for k in (8, 20):
        idx, d2 = hip.knn(X, k, True, device=gpu)
        idx0, d20 = oracle.knn(X, k, True)
        assert np.array_equal(d2, d20) and np.array_equal(idx, idx0)


def test_few_far_outliers_are_searched_one_by_one(gpu, monkeypatch, capfd):
    """A handful of stray points tens of cloud sizes away (what terrestrial scans carry): the grids
    are built over the box without those tails (grid.hip: robust_box — the strays clamp into the
    outermost cells, which keeps every ring bound valid) and the strays' own queries, which no ring
    of cells can answer, go to the whole-cloud wave kernel (knn.hip: k_knn_brute). Indices and
    distances equal the oracle's for every point; with PYQSM_KNN_ROBUST_BOX=0 (the grids of the
    full box, retry levels) the same bits come out."""
    rng = np.random.default_rng(23)
    P = synth.forest(60_000, seed=5)
    ext = P.max(0) - P.min(0)
    far = P.mean(0) + rng.choice([-1.0, 1.0], (40, 3)) * rng.uniform(20, 60, (40, 3)) * ext
    far[:5] = far[0] + rng.normal(0, 1e-3, (5, 3))           # a tiny far cluster: neighbours among themselves
    X = np.concatenate([P, far])[rng.permutation(60_040)].astype(np.float32).astype(np.float64)
    monkeypatch.setenv("PYQSM_KNN_TRACE", "1")
    for k in (8, 20, 64):
        capfd.readouterr()
        idx, d2 = hip.knn(X, k, True, device=gpu)
        err = capfd.readouterr().err
        assert "without its tails" in err and "searched over the whole cloud" in err
        idx0, d20 = oracle.knn(X, k, True)
        assert np.array_equal(d2, d20) and np.array_equal(idx, idx0)
    monkeypatch.setenv("PYQSM_KNN_ROBUST_BOX", "0")          # grids over the full box, retry levels
    capfd.readouterr()
    idx1, d21 = hip.knn(X, 20, True, device=gpu)
    assert "without its tails" not in capfd.readouterr().err
    idx0, d20 = oracle.knn(X, 20, True)
    assert np.array_equal(d21, d20) and np.array_equal(idx1, idx0)
