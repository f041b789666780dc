"""Same input, same bits. The contraction loop (pyQSM/geometry/skeletonize.py:226-373) amplifies a
one-ulp difference to millimetres within twenty steps, so a solver whose dot products are summed
by atomics in arrival order, or whose unknowns are numbered in the arrival order of a counting
sort, gives a visibly different skeleton on every run (round 1: up to 3e-4 relative). The solve
now reduces through per-block partial arrays added up in a fixed order (sparse.hpp:
reduce3_part / part_total3) and numbers its unknowns by a stable radix sort (scan.hip:
stable_sort_pairs_u32); kNN, the Laplacian and the multigrid setup were order-independent
already. These tests hold every stage to bit equality between two runs."""
import numpy as np
import pytest

from pyqsm_amd import hip, synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu


def test_knn_and_laplacian_are_reproducible(gpu):
    P = synth.forest(30_000, seed=2).astype(np.float64)
    i1, d1 = hip.knn(P, 30, device=gpu)
    i2, d2 = hip.knn(P, 30, device=gpu)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    (ip1, ix1, v1), m1 = hip.pc_laplacian(P, 30, 1e-5, device=gpu)
    (ip2, ix2, v2), m2 = hip.pc_laplacian(P, 30, 1e-5, device=gpu)
    assert np.array_equal(ip1, ip2) and np.array_equal(ix1, ix2)
    assert np.array_equal(v1, v2) and np.array_equal(m1, m2)


@pytest.mark.parametrize("cw,h", [(3.0, 1.0), (81.0, 9.0), (3.0 * 3 ** 8, 50.0)])
def test_contraction_solve_is_reproducible(gpu, cw, h):
    """Three conditioning regimes of the loop (first step, third step, a late step): the solution,
    the iteration count and the reported residuals are the same bits on every run — sorted
    unknowns (n >= 4096) and the plain path (n < 4096) alike."""
    for n in (30_000, 3000):
        P = synth.forest(n, seed=3).astype(np.float64)
        L, _ = hip.pc_laplacian(P, 30, 1e-5, device=gpu)
        wl, wh = np.full(n, cw), np.full(n, h)
        runs = [hip.lbc_solve(L, wl, wh, P, rtol=1e-8, device=gpu) for _ in range(3)]
        for x, iters, resid, ok in runs[1:]:
            assert np.array_equal(x, runs[0][0])
            assert iters == runs[0][1] and np.array_equal(resid, runs[0][2]) and ok == runs[0][3]


@pytest.mark.parametrize("engine", ["python", "native"])
def test_contraction_loop_is_reproducible(gpu, engine):
    P = synth.forest(20_000, seed=9)
    kw = dict(max_iter=10, termination_ratio=0.0, contraction_factor=3, attraction_factor=3,
              engine=engine)
    g1, t1, s1 = sk.extract_skeleton(P, **kw)
    g2, t2, s2 = sk.extract_skeleton(P, **kw)
    assert len(s1) == len(s2) == 10
    for a, b in zip(s1, s2):
        assert np.array_equal(a, b)
    assert np.array_equal(g1.points, g2.points) and np.array_equal(t1, t2)


def test_batch_is_reproducible_and_independent_of_the_worker_threads(gpu):
    """Groups are contracted by concurrent host threads on their own streams: the results do
    not depend on how many there are."""
    clouds = [synth.forest(5000 + 900 * j, seed=70 + j) + [9.0 * j, 0.0, 0.0] for j in range(5)]
    kw = dict(max_iter=5, termination_ratio=0.0, contraction_factor=3, attraction_factor=3,
              group_points=12_000)
    a = sk.extract_skeleton_batch(clouds, workers=1, **kw)
    b = sk.extract_skeleton_batch(clouds, workers=3, **kw)
    for (g1, _, _), (g2, _, _) in zip(a, b):
        assert np.array_equal(g1.points, g2.points)
