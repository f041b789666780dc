"""Region growing (tree_isolation.py:63-283): the one-call-per-cycle GPU
driver against the cluster-by-cluster SciPy restatement of the reference loop."""
import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.tree_isolation import extend_seed_clusters

pytestmark = pytest.mark.gpu


def _as_sets(clouds_by_label):
    return {label: {tuple(p) for p in np.asarray(pts).reshape(-1, 3)} for label, pts in clouds_by_label.items()}


@pytest.mark.parametrize("k,cycles,include_seeds", [(50, 25, True), (8, 40, False)])
def test_growth_matches_reference_loop(gpu, k, cycles, include_seeds):
    P = synth.forest(100_000, seed=1)                       # two trees + noise
    low = P[P[:, 2] < 0.3]
    lab, _ = hip.dbscan(low, 0.1, 10, device=gpu)
    seeds = [(f"tree{c}", low[lab == c]) for c in range(lab.max() + 1)]
    assert len(seeds) == 2
    src = P if include_seeds else P[P[:, 2] >= 0.3]         # seeds inside / outside the source
    tree_pcds, all_nbrs = extend_seed_clusters(seeds, src, "t", k=k, max_distance=0.1, cycles=cycles,
                                               device=gpu)
    want = _as_sets(oracle.extend_seed_clusters(seeds, src, k=k, max_distance=0.1, cycles=cycles))
    got = _as_sets({seeds[i][0]: tree_pcds[i].points for i in range(len(seeds))})
    assert got.keys() == want.keys()
    for label in want:
        assert got[label] == want[label], label
    grown = sum(len(a) for a in all_nbrs)
    assert grown > 5000                                      # the trunks really were climbed
    assert not (set(all_nbrs[0]) & set(all_nbrs[1]))         # no point belongs to two trees


def test_exclusion_zone_and_contested_points(gpu):
    """Two seeds growing towards each other along a strip of points: the lower index wins
    contested points; an exclusion cloud removes its neighbourhood from the source first.
    (A frontier of fewer than five points ends a cluster, tree_isolation.py:256-258, hence a
    strip six points wide rather than a line.)"""
    gx, gy = np.meshgrid(np.arange(200) * 0.02, np.arange(6) * 0.02, indexing="ij")
    strip = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1)
    seeds = [("a", strip[:12]), ("b", strip[-12:])]
    tree_pcds, _ = extend_seed_clusters(seeds, strip, "t", k=40, max_distance=0.05, cycles=200, device=gpu)
    want = _as_sets(oracle.extend_seed_clusters(seeds, strip, k=40, max_distance=0.05, cycles=200))
    got = _as_sets({"a": tree_pcds[0].points, "b": tree_pcds[1].points})
    assert got == want and len(got["a"]) + len(got["b"]) == len(strip)
    assert abs(len(got["a"]) - len(got["b"])) <= 24          # they meet in the middle
    ex = np.array([[2.0, 0.05, 0]])
    tree_pcds, _ = extend_seed_clusters(seeds, strip, "t", k=40, max_distance=0.05, cycles=200,
                                        exclude_pts=ex, device=gpu)
    want = _as_sets(oracle.extend_seed_clusters(seeds, strip, k=40, max_distance=0.05, cycles=200,
                                                exclude_pts=ex))
    got = _as_sets({"a": tree_pcds[0].points, "b": tree_pcds[1].points})
    assert got == want and len(got["a"]) + len(got["b"]) < len(strip)
