"""Host-side pieces that need no GPU: configuration, synthetic inputs, shard maths,
the flat-import layout, the skeleton loop's bookkeeping."""
import os
import subprocess
import sys

import numpy as np
import pytest

from pyqsm_amd import synth
from pyqsm_amd.parallel import shard_bounds, shard_sizes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_has_every_hot_path_key():
    from pyqsm_amd.set_config import config
    want = {"skeletonize": ["moll", "n_neighbors", "max_iter", "termination_ratio",
                            "init_contraction", "init_attraction", "max_contraction",
                            "max_attraction", "step_wise_contraction_amplification", "graph_k_n",
                            "semantic_weight"],
            "dbscan": ["epsilon", "min_neighbors"],
            "sphere": ["min_radius", "max_radius", "radius_multiplier", "dist",
                       "bad_fit_radius_factor", "min_contained_points"],
            "trunk": ["cluster_eps", "cluster_nn", "lower_pctile", "upper_pctile"]}
    for sec, keys in want.items():
        for k in keys:
            assert k in config[sec], (sec, k)
    assert config["dbscan"]["epsilon"] == 0.1 and config["dbscan"]["min_neighbors"] == 10
    assert config["skeletonize"]["init_contraction"] == 3      # the active value


def test_config_env_override(tmp_path):
    cfg = tmp_path / "my.toml"
    cfg.write_text("[dbscan]\nepsilon = 0.25\n[skeletonize]\ninit_contraction = 7\n")
    code = ("import sys; sys.path.insert(0, %r); from pyqsm_amd.set_config import config;"
            "print(config['dbscan']['epsilon'], config['dbscan']['min_neighbors'],"
            " config['skeletonize']['init_contraction'])" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                         env=dict(os.environ, PY_QSM_CONFIG=str(cfg)), check=True).stdout.split()
    assert out == ["0.25", "10", "7"]          # override + fallback to the packaged default


def test_flat_import_layout_like_pyqsm():
    """With pyqsm_amd/ itself on sys.path the reference's import lines work."""
    code = ("from math_utils.fit import cluster_DBSCAN, fit_shape_RANSAC\n"
            "from geometry.skeletonize import extract_skeleton, least_squares_sparse\n"
            "from geometry.point_cloud_processing import cluster_plus\n"
            "from viz.ray_casting import cast_rays\n"
            "from set_config import config, log\n"
            "print('ok')")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "pyqsm_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                       cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stderr


def test_signatures_match_the_reference():
    import inspect
    from pyqsm_amd.geometry.point_cloud_processing import cluster_plus
    from pyqsm_amd.geometry.skeletonize import extract_skeleton, least_squares_sparse
    from pyqsm_amd.math_utils.fit import cluster_DBSCAN, fit_shape_RANSAC
    from pyqsm_amd.viz.ray_casting import cast_rays

    def names(f):
        return list(inspect.signature(f).parameters)
    # the reference's parameters first and in order; this package only appends (device=...)
    assert names(cluster_DBSCAN) == ["pts_idxs", "points", "eps", "min_pts", "device"]
    assert names(fit_shape_RANSAC)[:7] == ["pcd", "pts", "threshold", "lower_bound", "max_radius",
                                           "align_to_z", "shape"]
    assert names(least_squares_sparse)[:5] == ["pts", "L", "laplacian_weighting",
                                               "positional_weighting", "trunk_points"]
    assert names(extract_skeleton)[:13] == [
        "pcd", "moll", "n_neighbors", "max_iter", "debug", "termination_ratio",
        "contraction_factor", "attraction_factor", "max_contraction", "max_attraction",
        "step_wise_contraction_amplification", "cmag_save_file", "min_contraction"]
    assert names(cluster_plus) == ["pcd", "eps", "min_points", "draw_result", "color_clusters",
                                   "from_points", "return_pcds", "ransac", "device", "radius_inclusive"]
    assert names(cast_rays)[:4] == ["tmesh", "surf_2d", "img", "pinhole_config"]
    sig = inspect.signature(fit_shape_RANSAC)
    assert sig.parameters["threshold"].default == 0.1 and sig.parameters["shape"].default == "circle"


def test_forest_generator():
    P = synth.forest(100_000, seed=0)
    assert P.shape == (100_000, 3) and P.dtype == np.float64
    assert np.array_equal(P, P.astype(np.float32).astype(np.float64))   # fp32-representable
    assert np.array_equal(P, synth.forest(100_000, seed=0))
    assert P[:, 0].max() > 9.0                                          # second tree at 10 m pitch


def test_canopy_and_rays():
    verts, tris = synth.canopy_mesh(1000)
    assert verts.dtype == np.float32 and tris.dtype == np.int32 and tris.shape == (1000, 3)
    assert tris.max() < len(verts)
    rays = synth.sun_rays(verts, 5000)
    assert rays.shape == (5000, 6) and rays.dtype == np.float32
    d = rays[:, 3:].astype(np.float64)
    assert np.allclose(np.linalg.norm(d, axis=1), 1.0, atol=1e-6) and np.all(d[:, 2] < 0)
    assert np.all(rays[:, 3:] == rays[0, 3:])                           # parallel


@pytest.mark.parametrize("n,world", [(10, 1), (10, 3), (7, 8), (10_000_000, 8), (0, 4)])
def test_shard_bounds_tile_the_range(n, world):
    edges = [shard_bounds(n, world, r) for r in range(world)]
    assert edges[0][0] == 0 and edges[-1][1] == n
    assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
    s = shard_sizes(n, world)
    assert sum(s) == n and max(s) - min(s) <= 1


def test_skeleton_loop_bookkeeping_against_oracle():
    """extract_skeleton's loop (weights, lagging volume ratio, M_list quirk) restated
    twice — wrapper vs oracle — must walk through identical states when both use the
    same solver and Laplacian (here: SciPy's direct solve and a graph Laplacian)."""
    import oracle
    from scipy.sparse import diags
    from tests.golden.make_golden import graph_laplacian
    from pyqsm_amd.geometry import skeletonize as sk

    pts = synth.forest(600, seed=3)

    def lap(p):
        L = graph_laplacian(p, 6)
        mass = np.full(len(p), 1e-3) * (1.0 + 0.1 * np.cos(np.arange(len(p))))
        # shrink the mass as points contract so that the loop can terminate
        mass = mass * (np.ptp(p, axis=0).prod() / np.ptp(pts, axis=0).prod())
        return L, mass

    lo, hi = sk.oriented_bounds(pts)
    want, want_total, want_steps = oracle.extract_skeleton(
        pts, lap, (lo, hi), max_iter=4, termination_ratio=0.5, contraction_factor=3,
        attraction_factor=3)

    # run the wrapper's loop with the oracle's solver injected (no GPU needed)
    orig_solve, orig_clamp = sk.least_squares_sparse, sk.hip.clamp
    sk.least_squares_sparse = lambda pts, L, laplacian_weighting, positional_weighting, **kw: \
        oracle.least_squares_sparse(pts, L, laplacian_weighting, positional_weighting)
    sk.hip.clamp = lambda p, lo, hi, device=0: np.minimum(np.maximum(p, lo), hi)
    try:
        got, total, steps = sk.extract_skeleton(
            pts, max_iter=4, termination_ratio=0.5, contraction_factor=3, attraction_factor=3,
            laplacian=lambda p: (lambda L, m: (L, diags(m)))(*lap(p)))
    finally:
        sk.least_squares_sparse, sk.hip.clamp = orig_solve, orig_clamp
    assert len(steps) == len(want_steps) >= 2
    assert np.allclose(got.points, want, rtol=0, atol=1e-12)
    assert np.allclose(total, want_total, rtol=0, atol=1e-12)


def test_label_grouping_equals_the_reference_loop():
    """fit.py:224-246 (`set(labels)` + one `labels == k` scan per cluster) against the O(n)
    grouping of cluster_DBSCAN: same set iteration order, same core members, same noise."""
    from pyqsm_amd.math_utils.fit import _group_labels
    rng = np.random.default_rng(0)
    for trial in range(150):
        n = int(rng.integers(1, 500))
        labels = rng.integers(-1, int(rng.integers(1, 60)), n).astype(np.int64)
        if trial % 3 == 0:
            labels[labels == -1] = 0
        core = rng.random(n) < 0.7
        pts = rng.permutation(n) + 100
        ul = set(labels)
        idxs, noise = [], []
        for k in ul:
            member = labels == k
            if k == -1:
                noise = pts[np.where(member & ~core)]
            else:
                idxs.append(pts[np.where(member & core)])
        ul2, idxs2, noise2 = _group_labels(labels, core, pts)
        assert list(ul) == list(ul2)
        assert len(idxs) == len(idxs2) and all(np.array_equal(a, b) for a, b in zip(idxs, idxs2))
        assert np.array_equal(np.asarray(noise), np.asarray(noise2))


def test_adopted_buffers_are_released_after_the_last_view():
    """hip._adopt wraps a malloc'ed output of the library in a NumPy array without copying;
    the buffer must live as long as ANY view of it and be released exactly once."""
    import ctypes
    import gc

    from pyqsm_amd import hip

    libc = ctypes.CDLL(None)
    libc.malloc.restype = ctypes.c_void_p
    libc.malloc.argtypes = [ctypes.c_size_t]
    libc.free.argtypes = [ctypes.c_void_p]
    freed = []

    class FakeLib:
        @staticmethod
        def pyqsm_free(p):
            freed.append(p.value)
            libc.free(p)

    raw = libc.malloc(10 * 4)
    ptr = ctypes.c_void_p(raw)
    arr = hip._adopt(FakeLib, ptr, ctypes.c_int32, 10, np.int32)
    arr[:] = np.arange(10)
    view = arr[3:7]
    del arr
    gc.collect()
    assert freed == [] and view.tolist() == [3, 4, 5, 6]
    del view
    gc.collect()
    assert freed == [raw]


def test_simplify_graph_collapses_every_degree_two_run():
    """skeletonize.py:57-98: against a one-node-at-a-time elimination on random trees."""
    import networkx as nx
    from pyqsm_amd.geometry.skeletonize import simplify_graph, simplify_and_update
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = int(rng.integers(2, 120))
        G = nx.Graph()
        G.add_nodes_from(range(n))
        for v in range(1, n):                        # random tree, long chains likely
            G.add_edge(v, int(rng.integers(max(0, v - 3), v)))
        for v in G.nodes:
            G.nodes[v]["pos"] = rng.normal(size=3)
        got, pos, idx = simplify_graph(G)
        ref = G.copy()
        for v in list(ref.nodes):
            if ref.degree(v) == 2:
                (a, da), (b, db) = [(w, ref.edges[v, w].get("data", [])) for w in ref.neighbors(v)]
                ref.remove_node(v)
                ref.add_edge(a, b, data=da + db + [v])
        assert set(got.nodes) == set(ref.nodes) and not any(d == 2 for _, d in got.degree())
        assert {frozenset(e) for e in got.edges} == {frozenset(e) for e in ref.edges}
        for a, b, d in ref.edges(data=True):
            assert sorted(got.edges[a, b].get("data", [])) == sorted(d.get("data", []))
        assert idx == [v for v in G.nodes if G.degree(v) != 2]
        assert all(np.array_equal(p, G.nodes[v]["pos"]) for p, v in zip(pos, idx))
        removed = sum(len(d.get("data", [])) for _, _, d in got.edges(data=True))
        assert removed == n - len(idx)
        # the run is listed in walking order: consecutive members are neighbours in G
        for a, b, d in got.edges(data=True):
            run = d.get("data", [])
            assert all(G.has_edge(x, y) for x, y in zip(run, run[1:]))
        relabeled, pts, mapping = simplify_and_update(G)
        assert sorted(relabeled.nodes) == list(range(len(idx))) and pts.shape == (len(idx), 3)


def test_select_by_index_is_by_vertex_like_open3d():
    """ray_casting.py:286-292 selects the hit mesh by VERTEX: on a grid plane (shared vertices)
    triangles whose three vertices were all hit come along although no ray hit them."""
    from pyqsm_amd.geometry.cloud import TriangleMesh
    g = 4
    v = np.array([[x, y, 0.0] for y in range(g) for x in range(g)], dtype=np.float32)
    t = []
    for y in range(g - 1):
        for x in range(g - 1):
            a = y * g + x
            t += [[a, a + 1, a + g], [a + 1, a + g + 1, a + g]]
    m = TriangleMesh(v, np.array(t))
    hit = [0, 3]            # lower triangle of cell (0,0) and upper triangle of cell (1,0)
    verts = np.unique(m.triangles[hit])
    sel = m.select_by_index(verts)
    # vertices {0,1,4} + {2,5,6}: triangle 1 = (1,5,4) and 2 = (1,2,5) are now complete as well
    assert len(sel.triangles) == 4 and abs(sel.get_surface_area() - 2.0) < 1e-12
    assert abs(m.select_by_triangle(hit).get_surface_area() - 1.0) < 1e-12
    assert np.array_equal(sel.vertices, v[verts])


def test_library_and_python_shard_the_same_way():
    """pyqsm_cast_rays_multi (C) and parallel.shard_bounds (Python, used by the one-process-per-GPU
    path and by bench.py) must cut the rays at the same places: results are placed by these bounds."""
    import ctypes
    from pyqsm_amd import _lib
    lib = _lib.load()
    for n in (0, 1, 7, 8, 9, 1001, 10_000_000, 2**33 + 5):
        for world in (1, 2, 3, 4, 8):
            prev = 0
            for rank in range(world):
                b, e = ctypes.c_int64(-1), ctypes.c_int64(-1)
                assert lib.pyqsm_shard_bounds(n, world, rank, ctypes.byref(b), ctypes.byref(e)) == 0
                assert (b.value, e.value) == shard_bounds(n, world, rank)
                assert b.value == prev
                prev = e.value
            assert prev == n
    b, e = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.pyqsm_shard_bounds(10, 2, 2, ctypes.byref(b), ctypes.byref(e)) == -1


def test_mean_f64_is_numpy_mean_bit_for_bit():
    """pyqsm_mean_f64 (host code of the native contraction loop): np.mean's summation order —
    pairwise blocks inside 8192-element buffers — so that the initial Laplacian weight
    10^3 c sqrt(mean M) (skeletonize.py:265) is the same double in both engines."""
    import ctypes
    from pyqsm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    sizes = list(range(1, 260)) + [1000, 4097, 8191, 8192, 8193, 16385, 65537, 100003, 300001, 1_000_000]
    sizes += [int(q) for q in rng.integers(1, 300_000, 60)]
    for n in sizes:
        v = rng.random(n) * float(rng.choice([1e-6, 1.0, 1e3]))
        out = ctypes.c_double()
        assert lib.pyqsm_mean_f64(v.ctypes.data_as(ctypes.c_void_p), n, ctypes.byref(out)) == 0
        assert out.value == float(np.mean(v)), n
    out = ctypes.c_double(1.0)
    assert lib.pyqsm_mean_f64(None, 0, ctypes.byref(out)) == 0 and np.isnan(out.value)


def test_draw_samples_rows_are_distinct_uniform_and_seeded():
    """RANSAC hypotheses (pyransac3d draws random.sample(range(n), 3) per iteration, unseeded):
    the vectorised sampler gives distinct indices, every ordered triple equally often, the same
    rows for the same seed; method="stdlib" is the literal per-row loop."""
    import random
    from pyqsm_amd.math_utils.fit import draw_samples
    s = draw_samples(5, 120_000, seed=1)
    assert s.dtype == np.int64 and s.shape == (120_000, 3) and s.min() == 0 and s.max() == 4
    assert (s[:, 0] != s[:, 1]).all() and (s[:, 0] != s[:, 2]).all() and (s[:, 1] != s[:, 2]).all()
    _, counts = np.unique(s[:, 0] * 25 + s[:, 1] * 5 + s[:, 2], return_counts=True)
    assert len(counts) == 60 and counts.min() > 0.9 * 2000 and counts.max() < 1.1 * 2000
    assert np.array_equal(draw_samples(1000, 64, seed=7), draw_samples(1000, 64, seed=7))
    assert sorted(draw_samples(3, 1, seed=0)[0]) == [0, 1, 2]
    rng = random.Random(5)
    want = np.array([rng.sample(range(50), 3) for _ in range(10)])
    assert np.array_equal(draw_samples(50, 10, seed=5, method="stdlib"), want)
    with pytest.raises(ValueError):
        draw_samples(2, 4)


def test_unique_rows_mm_equals_numpy_row_unique():
    """skeleton_to_QSM's per-cylinder ``np.unique(points.round(3), axis=0)`` (skeletonize.py:409 of the
    reference: ``to_points(...).round(3).unique()``) on integer millimetre keys: the same rows in the
    same order, also for negative coordinates, for duplicates and past the 21-bit range (fallback)."""
    from pyqsm_amd.geometry.skeletonize import _unique_rows_mm
    rng = np.random.default_rng(0)
    for scale, shift in ((1.0, 0.0), (0.01, -5.0), (30.0, 100.0), (5000.0, 0.0)):
        r = (rng.normal(0, scale, (3000, 3)) + shift).round(3)
        r[::7] = r[3]                                            # duplicates
        got = _unique_rows_mm(r)
        want = np.unique(r, axis=0)
        assert got.shape == want.shape and np.array_equal(got, want), scale


def test_batch_ransac_gives_every_set_its_own_stream(monkeypatch):
    """fit_shape_RANSAC_batch(seed=...) must not hand sets of equal size the same hypothesis
    triples: set q draws from SeedSequence(seed).spawn(S)[q]. The GPU call is
    replaced by a recorder — what is under test is what the wrapper sends to it."""
    from pyqsm_amd.math_utils import fit
    sent = {}

    def recorder(stacked, seg, tri, shape, threshold, device=0):
        sent["tri"], sent["seg"] = np.array(tri), np.array(seg)
        S = len(seg) - 1
        return (np.zeros((S, 3)), np.tile([0.0, 0.0, 1.0], (S, 1)), np.ones(S),
                [np.arange(3, dtype=np.int64)] * S, np.zeros(S, dtype=np.int64))

    monkeypatch.setattr(fit.hip, "ransac_batch", recorder)
    rng = np.random.default_rng(0)
    sets = [rng.normal(size=(500, 3)) for _ in range(4)]                 # equal sizes on purpose
    fit.fit_shape_RANSAC_batch(sets, shape="circle", seed=2, max_iterations=200)
    tri = sent["tri"]
    assert tri.shape == (4, 200, 3)
    for a in range(4):
        assert np.all(tri[a] >= 0) and np.all(tri[a] < 500)
        for b in range(a + 1, 4):
            assert not np.array_equal(tri[a], tri[b])
    first = tri.copy()
    fit.fit_shape_RANSAC_batch(sets, shape="circle", seed=2, max_iterations=200)
    assert np.array_equal(sent["tri"], first)                            # seeded: repeatable
    fit.fit_shape_RANSAC_batch(sets[:2], shape="circle", seed=2, max_iterations=200)
    assert np.array_equal(sent["tri"], first[:2])                        # set q's stream is child q
    fit.fit_shape_RANSAC_batch(sets, shape="circle", seed=3, max_iterations=200)
    assert not np.array_equal(sent["tri"], first)
