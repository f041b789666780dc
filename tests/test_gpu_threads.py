"""Several host threads on one GPU: the library keeps a stream, a scratch arena and timers
per (device, thread) (context.hip), so concurrent calls must give what the same calls give
one after the other — bit for bit for the integer paths, to the solver tolerance for the
contraction loop (its dot products are summed by floating-point atomics)."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from pyqsm_amd import hip, synth
from pyqsm_amd.geometry.skeletonize import extract_skeleton

pytestmark = pytest.mark.gpu


def _job(seed, gpu):
    P = synth.forest(60_000 + 7_000 * seed, seed=seed)
    labels, core = hip.dbscan(P, 0.1, 10, device=gpu)
    idx, d2 = hip.knn(P, 12, True, device=gpu)
    tree = synth.tree_unit(seed, 15_000 + 1_000 * seed)
    contracted, total, steps = extract_skeleton(tree, max_iter=2, termination_ratio=0.0, device=gpu)
    return labels, core, idx, d2, contracted.points


def test_threads_agree_with_sequential(gpu):
    seeds = list(range(6))
    want = [_job(s, gpu) for s in seeds]
    for round_ in range(2):          # the second pool's threads take over the first one's contexts
        with ThreadPoolExecutor(max_workers=4) as pool:
            got = list(pool.map(lambda s: _job(s, gpu), seeds))
        for w, g in zip(want, got):
            assert np.array_equal(w[0], g[0])
            assert np.array_equal(w[1], g[1])
            assert np.array_equal(w[2], g[2])
            assert np.array_equal(w[3], g[3])
            scale = np.abs(w[4]).max()
            assert np.abs(w[4] - g[4]).max() <= 1e-6 * scale


def test_error_messages_and_timers_stay_per_thread(gpu):
    """A failing call on one thread leaves the others' last_error and timers alone."""
    hip.prof_enable(True, device=gpu)
    hip.prof_reset(device=gpu)
    P = synth.forest(50_000, seed=3)

    def bad():
        with pytest.raises(Exception):
            hip.dbscan(P, -1.0, 10, device=gpu)
        hip.dbscan(P, 0.1, 10, device=gpu)
        return hip.prof_get("dbscan_core", device=gpu)[1]

    with ThreadPoolExecutor(max_workers=1) as pool:
        launches_there = pool.submit(bad).result()
    hip.dbscan(P, 0.1, 10, device=gpu)
    assert launches_there == 0                                 # that thread's context had no timers on
    assert hip.prof_get("dbscan_core", device=gpu)[1] == 1
    hip.prof_enable(False, device=gpu)
