"""pyqsm_extract_skeleton: the loop of pyQSM/geometry/skeletonize.py:240-373 run inside the library
(points, Laplacian and weights resident in HBM) against the Python loop over the same kernels.

Both engines run the same device code on the same inputs, and since the solver's reductions and
its spatial ordering no longer depend on the scheduling of atomics (sparse.hpp: reduce3_part,
scan.hip: stable_sort_pairs_u32) and the native loop takes its means in NumPy's summation order
(pyqsm_mean_f64), the two are held to EQUALITY, bit for bit, step by step. (Round 1 compared them
to the loop's run-to-run spread of up to 3e-4: the loop amplifies a one-ulp difference of the
initial Laplacian weight to millimetres within twenty steps, see test_gpu_batch.py.)"""
import numpy as np
import pytest

from pyqsm_amd import synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c", [(20_000, 3), (6000, 7)])
def test_native_loop_equals_python_loop(gpu, n, c):
    P = synth.forest(n, seed=n)
    kw = dict(max_iter=8, termination_ratio=0.0, contraction_factor=c, attraction_factor=3)
    g1, t1, s1 = sk.extract_skeleton(P, engine="python", **kw)
    g2, t2, s2 = sk.extract_skeleton(P, engine="native", **kw)
    assert len(s1) == len(s2) == 8 and len(g2.solve_log) == 8 and all(q["ok"] for q in g2.solve_log)
    for step, (a, b) in enumerate(zip(s1, s2)):
        assert np.array_equal(a, b), (step, float(np.abs(a - b).max()))
    assert np.array_equal(g1.points, g2.points) and np.array_equal(t1, t2)
    assert [q["iters"] for q in g1.solve_log] == [q["iters"] for q in g2.solve_log]
    assert np.abs(t2 - (P - g2.points)).max() < 1e-9
    assert np.abs(np.sum(s2, axis=0) - t2).max() < 1e-9
    lo, hi = sk.oriented_bounds(P)
    assert np.all(g2.points >= lo) and np.all(g2.points <= hi)


def test_native_loop_bookkeeping(gpu):
    """Step counts: max_iter, the termination ratio (lagging one step as in the reference) and
    max_iter = 0 (the reference still runs its first pass)."""
    P = synth.forest(8000, seed=5)
    for kw in (dict(max_iter=3, termination_ratio=0.0), dict(max_iter=6, termination_ratio=0.5),
               dict(max_iter=6, termination_ratio=0.9), dict(max_iter=0, termination_ratio=0.0)):
        a = sk.extract_skeleton(P, contraction_factor=3, attraction_factor=3, engine="python", **kw)
        b = sk.extract_skeleton(P, contraction_factor=3, attraction_factor=3, engine="native", **kw)
        assert len(a[2]) == len(b[2]) == len(b[0].solve_log), kw
        assert np.array_equal(a[0].points, b[0].points), kw


def test_native_batch_equals_python_batch(gpu):
    clouds = [synth.forest(5000 + 1200 * j, seed=40 + j) + [9.0 * j, 0.0, 0.0] for j in range(4)]
    kw = dict(max_iter=6, termination_ratio=0.0, contraction_factor=3, attraction_factor=3,
              group_points=100_000, workers=1)
    a = sk.extract_skeleton_batch(clouds, **kw)
    b = sk.extract_skeleton_batch(clouds, engine="native", **kw)
    for (g1, t1, s1), (g2, t2, s2), P in zip(a, b, clouds):
        assert len(s1) == len(s2) == 6
        for x, y in zip(s1, s2):
            assert np.array_equal(x, y)
        assert np.array_equal(g1.points, g2.points)
        assert np.abs(t2 - (P - g2.points)).max() < 1e-9
