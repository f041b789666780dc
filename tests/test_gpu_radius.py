"""Fixed-radius queries against SciPy's cKDTree — the engine the reference calls at
lib_integration.py:114-115 and reconstruction.py:238-244 (pinned: SciPy runs live here)."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from pyqsm_amd import hip, synth
from pyqsm_amd.geometry.reconstruction import get_neighbors_kdtree
from pyqsm_amd.utils.lib_integration import find_neighbors_in_ball, get_neighbors_in_tree

pytestmark = pytest.mark.gpu


def _kdtree_union(src, qry, dist, k):
    d, i = cKDTree(src).query(qry, k=k, distance_upper_bound=dist)
    i = np.atleast_2d(i.reshape(len(qry), -1))
    return np.unique(i[i < len(src)]), (i < len(src)).sum(axis=1)


@pytest.mark.parametrize("dist,k", [(0.05, 500), (0.3, 500), (0.3, 40), (0.15, 7)])
def test_radius_mark_matches_ckdtree(gpu, dist, k):
    src = synth.forest(30_000, seed=1)
    qry = synth.forest(30_000, seed=1)[::37] + [0.01, -0.02, 0.005]
    mask, counts = hip.radius_mark(src, qry, dist, k=k, device=gpu)
    want_idx, want_counts = _kdtree_union(src, qry, dist, k)
    assert np.array_equal(counts, want_counts)
    assert np.array_equal(np.flatnonzero(mask), want_idx)
    if k < 100:
        assert counts.max() == k                           # the cap was exercised


def test_radius_mark_strict_bound_and_outside_queries(gpu):
    src = np.array([[0.0, 0, 0], [0.5, 0, 0], [1.0, 0, 0], [0.25, 0, 0]])
    qry = np.array([[0.0, 0, 0], [100.0, 100, 100], [-3.0, 0, 0]])
    mask, counts = hip.radius_mark(src, qry, 0.5, k=4, device=gpu)
    assert list(np.flatnonzero(mask)) == [0, 3]            # 0.5 itself is excluded (strict)
    assert list(counts) == [2, 0, 0]


def test_ball_query_matches_ckdtree(gpu):
    P = synth.forest(50_000, seed=2)
    tree = cKDTree(P)
    for center, r in (([0.1, 0.0, 3.0], 0.5), ([2.0, 2.0, 2.0], 0.05), ([0, 0, 0], 100.0)):
        got = hip.ball_query(P, center, r, device=gpu)
        want = np.sort(tree.query_ball_point(center, r))
        assert np.array_equal(got, want)


def test_wrappers(gpu):
    P = synth.forest(20_000, seed=3)
    stem = P[(P[:, 2] > 1.0) & (P[:, 2] < 1.3) & (np.hypot(P[:, 0], P[:, 1]) < 0.4)]
    sphere, nbrs, center, radius = find_neighbors_in_ball(stem, P, [])
    want = np.sort(cKDTree(P).query_ball_point(center, radius))
    assert np.array_equal(nbrs, want) and 0.01 <= radius <= 1.5
    pcd, counts, uniq = get_neighbors_kdtree(P, query_pts=stem, dist=0.3)
    want_idx, _ = _kdtree_union(P, stem, 0.3, 500)
    assert np.array_equal(uniq, want_idx) and len(pcd.points) == len(uniq)
    assert get_neighbors_kdtree(P, query_pts=np.array([[99.0, 99, 99]]), dist=0.1) == (None,) * 3
    idx = get_neighbors_in_tree(stem, P, 0.2)
    t_sub = cKDTree(stem)
    pairs = cKDTree(P).query_ball_tree(t_sub, r=0.2)
    assert np.array_equal(idx, np.array([i for i, p in enumerate(pairs) if p]))


@pytest.mark.parametrize("k,radius", [(8, 0.05), (40, 0.08), (500, 0.05), (3, 0.3)])
def test_radius_knn_tables_match_ckdtree(gpu, k, radius):
    """get_neighbors_kdtree(return_pcd=False) (canopy_metrics.py:238): the padded [m,k]
    tables of cKDTree.query with a distance bound — distances bit-equal, indices equal
    wherever the distances are distinct, padding (inf, n)."""
    from scipy.spatial import cKDTree
    from pyqsm_amd.geometry.reconstruction import get_neighbors_kdtree
    rng = np.random.default_rng(k)
    src = synth.forest(40_000, seed=3)
    qry = np.concatenate([src[rng.choice(len(src), 3000, replace=False)] + rng.normal(0, 0.01, (3000, 3)),
                          rng.uniform(-5, 5, (500, 3)),                 # mostly far from everything
                          [[1e3, 1e3, 1e3]]])                           # outside the grid
    qry = qry.astype(np.float32).astype(np.float64)
    d0, i0 = cKDTree(src).query(qry, k=k, distance_upper_bound=radius)
    d, i = get_neighbors_kdtree(src, query_pts=qry, dist=radius, k=k, return_pcd=False, device=gpu)
    d0, i0 = d0.reshape(len(qry), k), i0.reshape(len(qry), k)
    assert d.shape == d0.shape and i.dtype == np.int64
    assert np.array_equal(d, d0)                       # includes the inf padding
    assert np.array_equal(np.isinf(d), i == len(src))
    # ties in distance may be ordered differently: compare per distinct distance
    same = i == i0
    if not same.all():
        rows = np.flatnonzero(~same.all(1))
        for r in rows:
            assert sorted(zip(d[r], i[r])) == sorted(zip(d0[r], i0[r]))
    full = np.isfinite(d[:, -1]).sum()
    assert (np.isinf(d[:, 0])).sum() > 100             # some queries find nothing
    if k <= 40:
        assert full > 100                              # the k cap is exercised


def test_stray_source_points_far_outside_keep_radius_queries_exact(gpu):
    """A few source points tens of cloud sizes away: the source grid is built over the cloud without
    those tails (radius.hip: source_grid; they clamp into its outermost cells) and queries — inside,
    next to a stray, and far from everything — still get cKDTree's answers: the union-of-balls mask
    with its per-query counts, and the padded distance tables."""
    rng = np.random.default_rng(9)
    P = synth.forest(40_000, seed=4)
    ext = P.max(0) - P.min(0)
    far = P.mean(0) + rng.choice([-1.0, 1.0], (30, 3)) * rng.uniform(15, 40, (30, 3)) * ext
    far[:6] = far[0] + rng.normal(0, 0.02, (6, 3))                       # a far clump: neighbours among themselves
    src = np.concatenate([P, far])[rng.permutation(40_030)].astype(np.float32).astype(np.float64)
    qry = np.concatenate([P[::61] + [0.01, -0.02, 0.005], far + rng.normal(0, 0.01, far.shape),
                          far[0] + [[5.0, 0, 0]], [[1e4, -1e4, 1e4]]]).astype(np.float32).astype(np.float64)
    for dist, k in ((0.1, 500), (0.3, 7)):
        mask, counts = hip.radius_mark(src, qry, dist, k=k, device=gpu)
        want_idx, want_counts = _kdtree_union(src, qry, dist, k)
        assert np.array_equal(counts, want_counts)
        assert np.array_equal(np.flatnonzero(mask), want_idx)
    assert want_counts[-32:-2].max() >= 5                                # the far clump was found
    d0, i0 = cKDTree(src).query(qry, k=8, distance_upper_bound=0.1)
    d, i = get_neighbors_kdtree(src, query_pts=qry, dist=0.1, k=8, return_pcd=False, device=gpu)
    assert np.array_equal(d, d0.reshape(len(qry), 8))
    distinct = np.isfinite(d) & (np.r_["1", d[:, 1:] != d[:, :-1], np.ones((len(d), 1), bool)])
    distinct[:, 1:] &= d[:, 1:] != d[:, :-1]
    assert np.array_equal(i[distinct], i0.reshape(len(qry), 8)[distinct])
    got = hip.ball_query(src, far[0], 0.2, device=gpu)
    assert np.array_equal(got, np.sort(cKDTree(src).query_ball_point(far[0], 0.2)))
