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


def test_small_and_sparse_clouds_beside_a_large_one(gpu):
    """a cloud with <= n_neighbors points (the small clusters of
    qsm_generation.py:182-316) must not borrow neighbours from the cloud next to it, and a very
    sparse cloud (k-th own neighbour farther away than a cloud diameter of the others) must stay
    its own block. The 10-point cloud takes the single-cloud call (which handles n <= k), so its
    result IS that call's; the sparse one shares the group and must agree with its own loop."""
    rng = np.random.default_rng(11)
    big = synth.forest(6000, seed=7)
    tiny = rng.normal(0, 0.05, (10, 3)) + [0.0, 0.0, 1.0]
    sparse = rng.uniform(-6, 6, (60, 3)) * [1, 1, 0.3] + [0, 0, 3.0]     # 60 points over 12 m
    clouds = [big, tiny, sparse]
    kw = dict(max_iter=3, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
    for engine in ("python", "native"):
        batch = sk.extract_skeleton_batch(clouds, group_points=100_000, workers=2, engine=engine, **kw)
        single = [sk.extract_skeleton(P, engine=engine, **kw) for P in clouds]
        assert np.array_equal(batch[1][0].points, single[1][0].points)            # same call, same bits
        for j in (0, 2):
            # (the 60-point cloud stops early on its own — its third solve returns the warm start
            # unchanged, skeletonize.py:287-289 — while in a group the shared CG scalars leave it
            # shifts of 1e-9: one more recorded step, the same cloud)
            assert len(single[j][2]) <= len(batch[j][2]) <= 3 and len(batch[0][2]) == 3
            scale = np.abs(clouds[j]).max()
            assert np.abs(batch[j][2][0] - single[j][2][0]).max() <= 2e-7 * scale, (engine, j)
            assert np.abs(batch[j][0].points - single[j][0].points).max() <= 1e-3 * scale, (engine, j)


def test_stacked_laplacian_refuses_clouds_that_see_each_other(gpu):
    """The device check behind the batch (k_check_segments): a segment with <= k points, or two
    clouds closer than their own point spacing, make the call fail instead of building fans across
    clouds; clouds laid out by the batch's own lattice pass."""
    from pyqsm_amd import _lib
    rng = np.random.default_rng(5)
    big = synth.forest(5000, seed=3)
    tiny = rng.normal(0, 0.05, (10, 3)) + [30.0, 0.0, 1.0]
    P = np.concatenate([big, tiny])
    with pytest.raises(_lib.PyQSMHipError, match="another cloud"):
        sk.point_cloud_laplacian(P, mollify_factor=1e-5, n_neighbors=20, device=gpu,
                                 seg_start=np.array([0, len(big), len(P)]))
    other = synth.forest(5000, seed=4) + [0.05, 0.0, 0.0]                  # interleaved with `big`
    P2 = np.concatenate([big, other])
    with pytest.raises(_lib.PyQSMHipError, match="another cloud"):
        sk.point_cloud_laplacian(P2, mollify_factor=1e-5, n_neighbors=20, device=gpu,
                                 seg_start=np.array([0, len(big), len(P2)]))
    sparse = rng.uniform(-6, 6, (60, 3)) * [1, 1, 0.3]
    pts = [big, sparse]
    offs = sk._lattice_offsets(pts)
    P3 = np.concatenate([p + o for p, o in zip(pts, offs)])
    L, M = sk.point_cloud_laplacian(P3, mollify_factor=1e-5, n_neighbors=20, device=gpu,
                                    seg_start=np.array([0, len(big), len(P3)]))
    coo = L.tocoo()
    assert not np.any((coo.row < len(big)) != (coo.col < len(big)))         # block diagonal
    # and the library is usable after the refusals
    L1, _ = sk.point_cloud_laplacian(big, mollify_factor=1e-5, n_neighbors=20, device=gpu)
    assert L1.shape == (5000, 5000)
