"""Farthest-point sampling and the topology extraction that follows the contraction
(skeletonize.py:113-146)."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,s", [(5000, 500), (20_000, 2000), (300, 300), (1000, 1)])
def test_fps_matches_oracle(gpu, n, s):
    P = synth.forest(n, seed=n)
    got = hip.fps(P, s, 0, device=gpu)
    want = oracle.farthest_point_sampling(P, s, 0)
    assert np.array_equal(got, want)                    # same points in the same order
    assert got[0] == 0 and len(np.unique(got)) == s


def test_fps_spreads_points(gpu):
    P = synth.forest(20_000, seed=1)
    idx = hip.fps(P, 200, 0, device=gpu)
    from scipy.spatial import cKDTree
    d, _ = cKDTree(P[idx]).query(P[idx], k=2)
    rnd = np.random.default_rng(0).choice(len(P), 200, replace=False)
    d_rnd, _ = cKDTree(P[rnd]).query(P[rnd], k=2)
    assert d[:, 1].min() > 3 * d_rnd[:, 1].min()        # far better separated than random


def test_fps_argument_errors(gpu):
    from pyqsm_amd._lib import PyQSMHipError
    P = synth.forest(100, seed=0)
    with pytest.raises(PyQSMHipError):
        hip.fps(P, 101, 0, device=gpu)
    with pytest.raises(PyQSMHipError):
        hip.fps(P, 10, 100, device=gpu)
    assert len(hip.fps(P, 0, 0, device=gpu)) == 0


def test_extract_topology_on_a_y_shape(gpu):
    """Three thin arms meeting at a point: the simplified graph must be a star with one
    junction (degree 3) and three leaves, and every removed node must be remembered on
    exactly one edge."""
    rng = np.random.default_rng(2)
    t = np.linspace(0.05, 1.0, 400)
    arms = [np.outer(t, d) for d in ([1, 0, 0.2], [-0.5, 0.8, 0.3], [-0.4, -0.9, 0.1])]
    P = np.concatenate(arms) + rng.normal(0, 1e-3, (1200, 3)) + [2.0, 2.0, 2.0]
    topo, tgraph, skel, skel_pts, sgraph, rx_graph, mapping = sk.extract_topology(P, graph_k_n=8)
    assert len(skel_pts) == 120 and sgraph.number_of_nodes() == 120
    assert sgraph.number_of_edges() == 119              # a spanning tree
    degs = sorted(d for _, d in tgraph.degree())
    assert degs == [1, 1, 1, 3]
    assert topo.lines.shape == (3, 2) and topo.points.shape == (4, 3)
    removed = sum(len(d.get("data", [])) for _, _, d in tgraph.edges(data=True))
    assert removed == 120 - 4


def test_skeleton_to_qsm_radii(gpu):
    rng = np.random.default_rng(2)
    t = np.linspace(0.05, 1.0, 400)
    arms = [np.outer(t, d) for d in ([1, 0, 0.2], [-0.5, 0.8, 0.3], [-0.4, -0.9, 0.1])]
    P = np.concatenate(arms) + rng.normal(0, 1e-3, (1200, 3)) + [2.0, 2.0, 2.0]
    topo, tgraph, *_ = sk.extract_topology(P, graph_k_n=8)
    shift = np.full((1200, 3), 0.05 / np.sqrt(3))        # every point contracted by 5 cm
    all_pcd, cyls, objs, radii = sk.skeleton_to_QSM(topo, tgraph, shift)
    assert len(cyls) == len(objs) == len(radii) == 3
    assert np.allclose(radii, 0.05)
    assert all(0.8 < o.height < 1.1 for o in objs) and len(all_pcd.points) > 1000


def test_fps_pruned_rounds_match_oracle_and_whole_cloud_rounds(gpu, monkeypatch):
    """From 65 536 points on, a round only touches the buckets the new sample can still improve
    (fps.hip: k_fps_pruned), and once the samples' reach is down to a cell or two all remaining
    rounds run inside ONE launch of one workgroup on Morton-ordered 64-point buckets with two levels of
    groups in LDS (k_fps_tail; PYQSM_FPS_TAIL=0 keeps a launch per round). Same indices as the NumPy restatement (start index not 0), and — at a
    size the restatement cannot reach — as the whole-cloud rounds (PYQSM_FPS_PRUNE=0), including a
    cloud where a third of the points are exact duplicates and the sampling runs until every
    distance is zero."""
    P = synth.forest(70_000, seed=8)
    got = hip.fps(P, 3000, 41, device=gpu)
    assert np.array_equal(got, oracle.farthest_point_sampling(P, 3000, 41))
    rng = np.random.default_rng(4)
    Q = synth.forest(300_000, seed=9)
    Q[rng.choice(len(Q), 100_000, replace=False)] = Q[rng.choice(len(Q), 100_000)]   # duplicates
    a = hip.fps(Q, 40_000, 7, device=gpu)
    R = np.concatenate([P[:50_000], P[:50_000][::-1][:30_000]])                      # 30 000 exact copies
    b = hip.fps(R, len(R), 0, device=gpu)                                            # down to distance zero
    monkeypatch.setenv("PYQSM_FPS_TAIL", "0")             # a launch per round to the end (round 2's path)
    assert np.array_equal(a, hip.fps(Q, 40_000, 7, device=gpu))
    assert np.array_equal(b, hip.fps(R, len(R), 0, device=gpu))
    monkeypatch.delenv("PYQSM_FPS_TAIL")
    monkeypatch.setenv("PYQSM_FPS_MAX_BLOCKS", "3")        # every block walks its buckets in several passes
    assert np.array_equal(a[:6000], hip.fps(Q, 6000, 7, device=gpu))
    monkeypatch.delenv("PYQSM_FPS_MAX_BLOCKS")
    monkeypatch.setenv("PYQSM_FPS_PRUNE", "0")
    assert np.array_equal(a, hip.fps(Q, 40_000, 7, device=gpu))
    assert np.array_equal(b, hip.fps(R, len(R), 0, device=gpu))
    assert len(np.unique(b[:50_000])) == 50_000                                      # every distinct point first
