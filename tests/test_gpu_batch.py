"""extract_skeleton_batch: many clouds contracted as block-diagonal systems (the per-cluster calls
of pyQSM/qsm_generation.py:182-316 around skeletonize.py:226-373) against the per-cloud loop."""
import numpy as np
import pytest

from pyqsm_amd import synth
from pyqsm_amd.geometry import skeletonize as sk

pytestmark = pytest.mark.gpu


def _trees(k, n):
    # different trees (seeds), different sizes, placed where a scan would have them
    out = []
    for j in range(k):
        P = synth.forest(n + 1500 * j, seed=100 + j)
        out.append(P + [13.0 * (j % 3), 11.0 * (j // 3), 0.0])
    return out


@pytest.mark.parametrize("c", [3, 7])
def test_batch_equals_per_cloud_loop(gpu, c):
    """First contraction: the two paths solve the same systems (same Laplacian blocks, same
    weights) and agree to the solver's tolerance. Later steps: the loop amplifies rounding
    differences through the Laplacian rebuilds on nearly degenerate contracted geometry. A run
    is reproducible bit for bit (test_gpu_reproducible.py), but the batch shares its CG scalars
    between the clouds of a group, so it differs from the per-cloud loop in the last bits of every
    solve. What those last bits grow to is measured here: the per-cloud loop on a copy of the
    cloud moved by ONE ULP in x differs from the original by up to ~1e-4 on these small clouds,
    and the batch is held to a small multiple of that, not to a fixed bound."""
    clouds = _trees(5, 6000)
    iters = 8
    kw = dict(max_iter=iters, termination_ratio=0.0, contraction_factor=c, attraction_factor=3)
    single = [sk.extract_skeleton(P, **kw) for P in clouds]
    again = [sk.extract_skeleton(np.column_stack([np.nextafter(P[:, 0], np.inf), P[:, 1:]]), **kw)
             for P in clouds]
    batch = sk.extract_skeleton_batch(clouds, group_points=20_000, workers=2, **kw)     # 2 groups
    assert len(batch) == len(clouds)
    spread = worst = first = 0.0
    for (g1, t1, s1), (g0, t0, s0), (g2, t2, s2), P in zip(single, again, batch, clouds):
        assert len(s1) == len(s2) == iters and len(g2.solve_log) == iters
        scale = np.abs(P).max()
        first = max(first, np.abs(s1[0] - s2[0]).max() / scale)
        for a, a0, b in zip(s1, s0, s2):
            spread = max(spread, np.abs(a - a0).max() / scale)
            worst = max(worst, np.abs(a - b).max() / scale)
        assert np.abs(t2 - (P - g2.points)).max() < 1e-9
        assert all(q["ok"] for q in g2.solve_log)
    print(f"c={c}: first step {first:.1e}; over {iters} steps batch vs loop {worst:.1e}, "
          f"loop on a one-ulp copy {spread:.1e}")
    assert first <= 2e-7
    assert worst <= max(30.0 * spread, 3e-4)      # 3e-4: the largest one-ulp response seen at this size


def test_batch_termination_is_per_cloud(gpu):
    """A cloud that meets its termination ratio stops (and keeps its state) while the others of
    its group go on: the same step counts as the per-cloud loop."""
    clouds = _trees(3, 5000)
    kw = dict(max_iter=6, termination_ratio=0.5, contraction_factor=3, attraction_factor=3)
    single = [sk.extract_skeleton(P, **kw) for P in clouds]
    batch = sk.extract_skeleton_batch(clouds, group_points=100_000, workers=1, **kw)
    assert [len(s[2]) for s in single] == [len(b[2]) for b in batch]
    for (g1, _, _), (g2, _, _), P in zip(single, batch, clouds):
        assert np.abs(g1.points - g2.points).max() <= 1e-3 * np.abs(P).max()   # see the test above


def test_pack_groups():
    g = sk._pack_groups([50, 10, 40, 70, 30], 100)
    assert sorted(sum(g, [])) == [0, 1, 2, 3, 4]
    assert all(sum([50, 10, 40, 70, 30][j] for j in grp) <= 100 for grp in g)


def test_stacked_laplacian_is_the_block_diagonal_of_the_per_cloud_ones(gpu):
    """pyqsm_pc_laplacian_seg: clouds laid out apart give one block per cloud, each equal to the
    cloud's own Laplacian (same neighbours, same fans, same flips, its OWN mollification length);
    without seg_start the shared mollification length changes the values."""
    from scipy.sparse import block_diag
    clouds = [synth.forest(4000 + 700 * j, seed=60 + j) * (1.0 + 0.6 * j) + [40.0 * j, 0, 0] for j in range(3)]
    P = np.concatenate(clouds)
    start = np.concatenate([[0], np.cumsum([len(c) for c in clouds])])
    L, M = sk.point_cloud_laplacian(P, mollify_factor=1e-3, n_neighbors=20, device=gpu, seg_start=start)
    parts = [sk.point_cloud_laplacian(c, mollify_factor=1e-3, n_neighbors=20, device=gpu) for c in clouds]
    want = block_diag([q[0] for q in parts], format="csr")
    want.sort_indices()
    assert np.array_equal(L.indptr, want.indptr) and np.array_equal(L.indices, want.indices)
    scale = np.abs(want.data).max()
    assert np.abs(L.data - want.data).max() <= 1e-12 * scale
    mass = np.concatenate([q[1].diagonal() for q in parts])
    assert np.abs(M.diagonal() - mass).max() <= 1e-12 * mass.max()
    # one global mollification length instead (no seg_start) is a different matrix (other edge
    # lengths, other flips)
    L1, _ = sk.point_cloud_laplacian(P, mollify_factor=1e-3, n_neighbors=20, device=gpu)
    assert L1.nnz != want.nnz or np.abs(L1.data - want.data).max() > 1e-9 * scale
