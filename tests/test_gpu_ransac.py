"""HIP RANSAC vs the NumPy restatement of pyransac3d (oracle.ransac_*).

pyransac3d itself is not installable here (PARITY UNPINNED against the package);
the oracle follows its published algorithm, and analytic known answers pin both."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth

pytestmark = pytest.mark.gpu


def _triples(rng, n, H):
    return np.stack([rng.choice(n, 3, replace=False) for _ in range(H)]).astype(np.int64)


def _flat(pts):
    q = pts.copy()
    q[:, 2] = 0.0          # fit.py:274-276: circle fits use a z-flattened copy
    return q


def test_models_match_numpy(gpu):
    pts = synth.ring_cluster(2000, seed=1)
    tri = _triples(np.random.default_rng(0), len(pts), 300)
    m = hip.ransac_models(pts, tri, device=gpu)
    for h in range(len(tri)):
        ref = oracle.ransac_model(pts[tri[h]])
        if ref is None:
            assert m[h, 7] == 0
            continue
        c, a, r = ref
        assert m[h, 7] == 1
        scale = max(1.0, abs(r))
        assert np.allclose(m[h, 0:3], c, rtol=0, atol=1e-9 * scale)
        assert np.allclose(m[h, 3:6], a, rtol=0, atol=1e-12)
        assert abs(m[h, 6] - r) <= 1e-9 * scale


@pytest.mark.parametrize("shape", ["circle", "cylinder"])
def test_counts_bit_exact_given_models(gpu, shape):
    pts = synth.ring_cluster(5000, seed=2)
    if shape == "circle":
        pts = _flat(pts)
    tri = _triples(np.random.default_rng(1), len(pts), 200)
    models = hip.ransac_models(pts, tri, device=gpu)
    counts = hip.ransac_count(pts, models, shape, 0.04, device=gpu)
    for h in range(len(tri)):
        if models[h, 7] == 0:
            assert counts[h] == 0
            continue
        d = oracle.ransac_distance(pts, models[h, 0:3], models[h, 3:6], models[h, 6], shape)
        assert counts[h] == int((d <= 0.04).sum())


@pytest.mark.parametrize("shape,n,H,seed", [("circle", 4000, 1000, 3), ("cylinder", 3000, 500, 4),
                                            ("circle", 300, 64, 5)])
def test_end_to_end_same_winner_and_inliers(gpu, shape, n, H, seed):
    pts = synth.ring_cluster(n, seed=seed)
    fit = _flat(pts) if shape == "circle" else pts
    tri = _triples(np.random.default_rng(seed), n, H)
    c, a, r, inl, best = hip.ransac(fit, tri, shape, 0.04, device=gpu)
    c0, a0, r0, inl0, best0 = oracle.ransac_fit(fit, tri, shape, 0.04)
    assert best == best0
    assert np.array_equal(inl, inl0)                   # inlier set identical
    assert np.allclose(c, c0, atol=1e-9) and np.allclose(a, a0, atol=1e-12) and abs(r - r0) < 1e-9
    if shape == "circle":
        assert len(inl) > 0.6 * n and abs(r - 0.3) < 0.02  # it found the stem


def test_exact_circle_known_answer(gpu):
    ang = np.linspace(0, 2 * np.pi, 200, endpoint=False)
    pts = np.stack([2 + 0.5 * np.cos(ang), -1 + 0.5 * np.sin(ang), np.zeros_like(ang)], 1)
    pts = np.concatenate([pts, [[10.0, 10.0, 0.0]] * 7])
    tri = np.array([[0, 50, 120], [3, 77, 150]], dtype=np.int64)
    c, a, r, inl, best = hip.ransac(pts, tri, "circle", 1e-6, device=gpu)
    assert best == 0 and len(inl) == 200 and list(inl) == list(range(200))
    assert np.allclose(c, [2, -1, 0], atol=1e-9) and abs(r - 0.5) < 1e-9
    assert np.allclose(np.abs(a), [0, 0, 1], atol=1e-12)


def test_degenerate_samples_and_empty(gpu):
    pts = synth.ring_cluster(100, seed=0)
    pts[1] = pts[0]                                    # coincident samples -> invalid model
    tri = np.array([[0, 1, 2]], dtype=np.int64)
    c, a, r, inl, best = hip.ransac(_flat(pts), tri, "circle", 0.04, device=gpu)
    assert best == -1 and len(inl) == 0
    c, a, r, inl, best = hip.ransac(pts, np.zeros((0, 3), np.int64), "circle", 0.04, device=gpu)
    assert best == -1 and len(inl) == 0


@pytest.mark.parametrize("shape", ["circle", "cylinder"])
def test_batch_equals_one_call_per_set(gpu, shape):
    """pyqsm_ransac_batch: stacked sets of very different sizes (one empty, one with two points, one
    larger than a tile, one where no hypothesis has an inlier) give, set by set, the bits of
    pyqsm_ransac: model, winning row, inlier indices. Then the same through
    fit_shape_RANSAC_batch against fit_shape_RANSAC (fit.py:253-339), rejections included."""
    from pyqsm_amd.math_utils import fit
    rng = np.random.default_rng(8)
    sets = []
    for n, r in ((400, 0.3), (0, 0.0), (2, 0.0), (3000, 0.12), (57, 0.05), (1500, 0.6)):
        a = rng.uniform(0, 2 * np.pi, n)
        P = np.stack([r * np.cos(a) + rng.normal(0, 0.004, n) + 3.0, r * np.sin(a) + rng.normal(0, 0.004, n) - 1.0,
                      rng.uniform(0.0, 0.5, n)], axis=1)
        P[: n // 5] += rng.normal(0, 0.2, (n // 5, 3))                  # outliers
        sets.append(P)
    sets.append(np.arange(30.0).reshape(10, 3) * [1.0, 0.0, 0.0])          # collinear: no valid model
    H = 300
    seg = np.concatenate([[0], np.cumsum([len(p) for p in sets])])
    tri = np.stack([fit.draw_samples(len(p), H, seed=3) if len(p) >= 3 else np.full((H, 3), -1) for p in sets])
    flat = [p.copy() for p in sets]
    if shape == "circle":
        for p in flat:
            p[:, 2] = 0.0
    c, a, r, inl, best = hip.ransac_batch(np.concatenate(flat), seg, tri, shape, 0.03, device=gpu)
    for q, p in enumerate(flat):
        if len(p) < 3:
            assert best[q] == -1 and len(inl[q]) == 0
            continue
        c1, a1, r1, inl1, b1 = hip.ransac(p, tri[q], shape, 0.03, device=gpu)
        assert b1 == best[q]
        assert np.array_equal(inl1, inl[q])
        if b1 >= 0:
            assert np.array_equal(c1, c[q]) and np.array_equal(a1, a[q]) and r1 == r[q]
    assert best[-1] == -1 and best[0] >= 0 and best[3] >= 0
    got = fit.fit_shape_RANSAC_batch([p.copy() for p in sets], threshold=0.03, max_radius=0.5, shape=shape,
                                     samples=list(tri), device=gpu)
    for q, p in enumerate(sets):
        want = fit.fit_shape_RANSAC(pts=p.copy(), threshold=0.03, max_radius=0.5, shape=shape,
                                    samples=tri[q], device=gpu)
        if want[0] is None:
            assert got[q][0] is None
            continue
        assert np.array_equal(want[2], got[q][2]) and want[3] == got[q][3] and np.array_equal(want[4], got[q][4])
        assert np.array_equal(want[0].center, got[q][0].center) and want[0].height == got[q][0].height
    assert got[5][0] is None                                               # radius 0.6 > max_radius
