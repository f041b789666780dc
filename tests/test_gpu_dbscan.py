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


@pytest.mark.parametrize("seed,n,eps", [(5, 30_000, 0.02), (6, 30_000, 0.026), (7, 5_000, 0.05)])
def test_single_link_components_min_pts_1(gpu, seed, n, eps):
    """min_pts = 1: every point is core, clusters are the connected components of the eps
    graph — around the percolation threshold they come in all sizes and shapes."""
    rng = np.random.default_rng(seed)
    Q = rng.uniform(0, 1, (n, 3)).astype(np.float32).astype(np.float64)
    lab, core = _check(Q, eps, 1, gpu)
    assert core.all() and lab.min() == 0 and lab.max() > 50


def test_lattice_at_exactly_eps(gpu):
    """Exactly representable spacing: d2 == eps^2 joins (inclusive compare), one ulp more does
    not; neighbours then sit exactly one cell / two sub-cells apart."""
    g = np.stack(np.meshgrid(*[np.arange(12.0)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.125
    lab, core = _check(g, 0.125, 7, gpu)          # interior points have 6 neighbours + themselves
    assert lab.max() == 0
    lab, core = _check(g, np.nextafter(0.125, 0), 2, gpu)
    assert not core.any() and (lab == -1).all()


def test_two_blobs_joined_by_one_core_pair(gpu):
    rng = np.random.default_rng(8)
    a = rng.normal(0, 0.01, (300, 3))
    b = rng.normal(0, 0.01, (300, 3)) + [0.5, 0, 0]
    # a chain of core points between them, neighbours exactly eps = 0.0625 apart; every chain
    # point is made core by two satellites
    chain = np.array([[0.0625 * i, 0.3, 0.0] for i in range(9)])
    sat = np.concatenate([chain + [0, 0.01, 0], chain + [0, -0.01, 0]])
    hook_a = np.array([[0.0, 0.3 - 0.0625 * i, 0.0] for i in range(1, 5)])
    hook_b = np.array([[0.5, 0.3 - 0.0625 * i, 0.0] for i in range(1, 5)])
    hooks = np.concatenate([hook_a, hook_b])
    hsat = np.concatenate([hooks + [0.01, 0, 0], hooks + [-0.01, 0, 0]])
    P = np.concatenate([a, b, chain, sat, hooks, hsat]).astype(np.float32).astype(np.float64)
    lab, core = _check(P, 0.0625, 3, gpu)
    assert lab[0] == lab[300]                       # the blobs are one cluster through the chain
    Q = np.delete(P, [604, 609 + 4, 618 + 4], axis=0)   # cut the chain (link 4 and its satellites)
    lab, _ = _check(Q, 0.0625, 3, gpu)
    assert lab[0] != lab[300]


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 257])
def test_sizes_around_a_wave(gpu, n):
    rng = np.random.default_rng(n)
    _check(rng.normal(0, 0.05, (n, 3)), 0.03, 3, gpu)


@pytest.mark.parametrize("n", [70_001, 300_000])
def test_every_point_a_straggler_of_the_core_pass(gpu, n):
    """A sparse uniform cloud (a neighbour or two within eps, min_pts = 10): no lane of the tiled core pass
    ever reaches min_pts, so every wave hands all of its 64 points to the straggler list — whose 64
    segments (one counter each, dbscan.hip: kRestSegs) are then filled to their capacity of 256 points per
    block — and k_core_rest recounts every point of the cloud. Labels against the oracle: all noise but
    the few chance clusters."""
    rng = np.random.default_rng(n)
    side = (n / 2.0) ** (1.0 / 3.0) * 0.05
    P = rng.uniform(0.0, side, (n, 3))
    _check(P, 0.03, 10, gpu)
    lab, core = hip.dbscan(P, 0.03, 10, device=gpu)
    assert (lab == -1).mean() > 0.9


def test_planar_and_single_cell(gpu):
    rng = np.random.default_rng(9)
    P = np.concatenate([rng.uniform(0, 1, (8000, 2)), np.zeros((8000, 1))], 1)
    _check(P, 0.02, 4, gpu)                         # z extent zero
    _check(rng.uniform(0, 0.01, (500, 3)), 1.0, 5, gpu)   # eps far above the extent: one cell


def test_extent_far_beyond_the_dense_grid(gpu):
    """Extent / eps far beyond what a dense grid of edge eps can hold (3333^3 cells): the binning
    drops the empty slabs per axis (grid.hip: build_grid_octants), the cells keep their edge and
    the sub-cell fast path runs (k_hook_sub is launched). Many small clusters inside single
    cells plus outliers 100 units away."""
    rng = np.random.default_rng(11)
    blob = rng.uniform(0, 0.6, (6000, 3))
    far = rng.uniform(-50, 50, (40, 3))
    P = np.concatenate([blob, far]).astype(np.float32).astype(np.float64)
    hip.prof_enable(True, gpu)
    hip.prof_reset(gpu)
    lab, core = _check(P, 0.03, 4, gpu)
    launches = hip.prof_get("k_hook_sub", gpu)[1]
    hip.prof_enable(False, gpu)
    assert launches >= 1                                  # the sub-cell path, not k_union_points
    assert lab.max() > 5 and (lab == -1).sum() > 40 and (~core & (lab >= 0)).sum() > 0
    _check(P, 0.012, 1, gpu)


def test_two_dense_blobs_a_kilometre_apart(gpu):
    """Compression has to keep neighbours neighbours: clusters that straddle slab boundaries
    next to long empty stretches, on every axis."""
    rng = np.random.default_rng(12)
    a = rng.uniform(0, 0.5, (4000, 3))
    b = rng.uniform(0, 0.5, (4000, 3)) + [1000.0, -750.0, 300.0]
    chain = np.stack([np.linspace(0, 3, 400), np.zeros(400), np.zeros(400)], 1) + [500.0, 0, 0]
    P = np.concatenate([a, b, chain, rng.uniform(-1000, 1000, (30, 3))])
    P = P.astype(np.float32).astype(np.float64)
    _check(P, 0.04, 5, gpu)
    _check(P, 0.011, 2, gpu)


def test_cells_with_hundreds_of_points(gpu):
    """eps far above the point spacing: cells hold more than kBigCell points and are ordered by
    the block-per-cell pass; also everything in ONE cell."""
    rng = np.random.default_rng(13)
    P = rng.uniform(0, 1, (30_000, 3)).astype(np.float32).astype(np.float64)
    _check(P, 0.2, 10, gpu)
    _check(P * [1, 1, 0.01], 0.25, 50, gpu)
    _check(rng.uniform(0, 0.01, (3000, 3)), 1.0, 5, gpu)


def test_randomised_differential(gpu):
    """Sixty small random configurations (sizes, densities, shapes, duplicates, lattices,
    min_pts from 1 up) against the sequential oracle."""
    rng = np.random.default_rng(2024)
    for case in range(60):
        n = int(rng.integers(1, 3000))
        kind = case % 5
        if kind == 0:
            P = rng.uniform(0, 1, (n, 3))
        elif kind == 1:                                   # blobs of very different density
            centres = rng.uniform(0, 1, (6, 3))
            P = centres[rng.integers(0, 6, n)] + rng.normal(0, 1, (n, 3)) * rng.choice([0.005, 0.02, 0.08], (n, 1))
        elif kind == 2:                                   # planar / linear
            P = rng.uniform(0, 1, (n, 3)) * [1, rng.choice([0, 1]), 0]
        elif kind == 3:                                   # lattice with duplicates, exact spacing
            P = rng.integers(0, 12, (n, 3)) * 0.0625
        else:                                             # wide extent: coarsened grid
            P = np.concatenate([rng.uniform(0, 0.3, (n, 3)), rng.uniform(-40, 40, (3, 3))])
        P = P.astype(np.float32).astype(np.float64)
        eps = float(rng.choice([0.0625, 0.03, 0.1, 0.011]))
        min_pts = int(rng.choice([1, 2, 3, 5, 10, 30]))
        lab, core = hip.dbscan(P, eps, min_pts, device=gpu)
        lab0, core0 = oracle.dbscan(P, eps, min_pts)
        assert np.array_equal(core, core0), (case, n, eps, min_pts)
        assert np.array_equal(lab, lab0), (case, n, eps, min_pts)


def test_points_exactly_eps_apart_under_both_radius_rules(gpu):
    """pyqsm_dbscan_ex: on a lattice whose spacing IS eps (0.5: d2 = 0.25 = eps*eps exactly in
    fp64) the inclusive rule (scikit-learn's, the default) makes every point with six lattice
    neighbours a core point at min_pts = 7 and the block one cluster; the strict rule (Open3D's
    compare if nanoflann's is strict) leaves every point alone. Both against the oracle's two
    forms, and the strict rule on an ordinary cloud where the boundary is never hit."""
    g = np.arange(12) * 0.5
    P = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T.copy()
    rng = np.random.default_rng(1)
    P = P[rng.permutation(len(P))]
    lab, core = hip.dbscan(P, 0.5, 7, device=gpu)                          # inclusive
    lab0, core0 = oracle.dbscan(P, 0.5, 7)
    assert np.array_equal(lab, lab0) and np.array_equal(core, core0)
    interior = np.all((P > 0) & (P < 5.5), axis=1)
    assert np.array_equal(core, interior) and lab.max() == 0 and (lab[core] == 0).all()
    assert (lab == -1).sum() == 128                            # edges and corners of the block: no core neighbour
    for min_pts in (7, 2, 1):
        lab, core = hip.dbscan(P, 0.5, min_pts, device=gpu, radius_inclusive=False)
        lab0, core0 = oracle.dbscan(P, 0.5, min_pts, radius_inclusive=False)
        assert np.array_equal(lab, lab0) and np.array_equal(core, core0)
        if min_pts > 1:
            assert not core.any() and (lab == -1).all()                    # nobody within d < eps
        else:
            assert core.all() and np.array_equal(np.sort(lab), np.arange(len(P)))   # every point its own cluster
    # a hair above the lattice spacing both rules see the six neighbours
    eps = np.nextafter(0.5, 1.0)
    a = hip.dbscan(P, eps, 7, device=gpu, radius_inclusive=False)
    b = hip.dbscan(P, eps, 7, device=gpu)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], interior)
    Q = synth.forest(30_000, seed=9)
    s_lab, s_core = hip.dbscan(Q, 0.1, 10, device=gpu, radius_inclusive=False)
    o_lab, o_core = oracle.dbscan(Q, 0.1, 10, radius_inclusive=False)
    assert np.array_equal(s_lab, o_lab) and np.array_equal(s_core, o_core)


def test_fp32_records_and_the_fp64_fallback(gpu, monkeypatch):
    """Coordinates that are all exactly representable in fp32 (every PCD / LAS derived cloud, every
    fixture) are kept as one 16-byte fp32 record per sorted point and widened to fp64 in registers:
    the predicate sees the input values, so labels and core set are those of the fp64 arrays — and
    of the oracle. One coordinate that is NOT representable sends the whole call to fp64 storage."""
    P = synth.forest(60_000, seed=21)
    assert np.array_equal(P, P.astype(np.float32).astype(np.float64))
    lab0, core0 = oracle.dbscan(P, 0.1, 10)

    def run(Q):
        hip.prof_enable(True, gpu)
        hip.prof_reset(gpu)
        lab, core = hip.dbscan(Q, 0.1, 10, device=gpu)
        used = hip.prof_get("dbscan_f32_records", gpu)[1]
        hip.prof_enable(False, gpu)
        return lab, core, used

    lab, core, used = run(P)
    assert used == 1 and np.array_equal(lab, lab0) and np.array_equal(core, core0)
    monkeypatch.setenv("PYQSM_COORD_F32", "0")                            # the same cloud in fp64 arrays
    lab, core, used = run(P)
    assert used == 0 and np.array_equal(lab, lab0) and np.array_equal(core, core0)
    monkeypatch.delenv("PYQSM_COORD_F32")
    # not representable: a jitter far below fp32 resolution that still moves points across eps
    rng = np.random.default_rng(2)
    Q = P + rng.uniform(-1e-9, 1e-9, P.shape)
    assert not np.array_equal(Q, Q.astype(np.float32).astype(np.float64))
    lab, core, used = run(Q)
    labq, coreq = oracle.dbscan(Q, 0.1, 10)
    assert used == 0 and np.array_equal(lab, labq) and np.array_equal(core, coreq)
    # a single such coordinate is enough for the fallback
    R = P.copy()
    R[12345, 1] = np.nextafter(R[12345, 1], np.inf)
    lab, core, used = run(R)
    labr, corer = oracle.dbscan(R, 0.1, 10)
    assert used == 0 and np.array_equal(lab, labr) and np.array_equal(core, corer)
    # exact-eps pairs survive the narrower storage (values are identical, so is d2 <= eps^2)
    g = np.arange(10) * 0.5
    Lat = np.array(np.meshgrid(g, g, g, indexing="ij")).reshape(3, -1).T.copy()
    lab, core, used = run(Lat)
    assert used == 1
    lab2, core2 = hip.dbscan(Lat, 0.5, 7, device=gpu)
    lo, co = oracle.dbscan(Lat, 0.5, 7)
    assert np.array_equal(lab2, lo) and np.array_equal(core2, co)
