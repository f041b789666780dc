"""The reference-named wrappers (pyQSM signatures) on the GPU, against the oracle's
restatement of the same reference functions."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
from pyqsm_amd import hip, synth
from pyqsm_amd.geometry.point_cloud_processing import cluster_and_get_largest, cluster_plus
from pyqsm_amd.math_utils import fit
from pyqsm_amd.viz import ray_casting as rc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cluster_DBSCAN_matches_reference_postprocessing(gpu):
    rng = np.random.default_rng(3)
    P = (rng.uniform(0, 1, (9000, 3)) * [1, 1, 0.15]).astype(np.float32).astype(np.float64)
    ids = rng.permutation(50_000)[:9000]
    labels, idxs, noise = fit.cluster_DBSCAN(ids, P, 0.03, 6)
    labels0, idxs0, noise0 = oracle.cluster_DBSCAN(ids, P, 0.03, 6)
    assert labels == labels0 and len(idxs) == len(idxs0) > 5
    for a, b in zip(idxs, idxs0):
        assert np.array_equal(a, b)
    assert np.array_equal(noise, noise0)


def test_config_1_50k_tree_through_the_wrapper(gpu):
    """BASELINE.json configs[0]: 50 k-point trunk+branch cloud, eps=0.1, min_neighbors=10."""
    from pyqsm_amd.set_config import config
    P = synth.forest(50_000)
    labels, idxs, noise = fit.cluster_DBSCAN(np.arange(len(P)), P, config["dbscan"]["epsilon"],
                                             config["dbscan"]["min_neighbors"])
    labels0, idxs0, noise0 = oracle.cluster_DBSCAN(np.arange(len(P)), P, 0.1, 10)
    assert labels == labels0 == {0, -1}
    assert np.array_equal(idxs[0], idxs0[0]) and np.array_equal(noise, noise0)


def test_cluster_plus_and_largest(gpu):
    P = synth.forest(100_000)
    by_label = cluster_plus(P, eps=0.1, min_points=10, return_pcds=False)
    lab0, _ = oracle.dbscan(P, 0.1, 10)
    assert sorted(by_label) == sorted(np.unique(lab0))
    for k, idx in by_label.items():
        assert np.array_equal(idx, np.flatnonzero(lab0 == k))
    clouds = cluster_plus(P, eps=0.1, min_points=10)
    assert [len(c.points) for c in clouds] == [len(v) for v in by_label.values()]
    big = cluster_and_get_largest(P, eps=0.1, min_points=10)
    assert len(big.points) == max(len(v) for v in by_label.values())


def test_fit_shape_RANSAC_circle_like_fit_cyl_to_cluster(gpu):
    pts = synth.ring_cluster(3000, seed=6)
    before = pts.copy()
    lb = float(np.min(pts[:, 2])) + 0.1
    samples = fit.draw_samples(len(pts), 1000, seed=2)
    mesh, in_pcd, inliers, r, axis = fit.fit_shape_RANSAC(
        pts=pts, shape="circle", threshold=0.04, lower_bound=lb, max_radius=0.3 * 1.75,
        samples=samples)
    assert np.all(pts[:, 2] >= lb) and not np.array_equal(pts, before)   # in-place clamp (:265)
    flat = pts.copy()
    flat[:, 2] = 0
    c0, a0, r0, inl0, _ = oracle.ransac_fit(flat, samples, "circle", 0.04)
    assert np.array_equal(inliers, inl0) and abs(r - r0) < 1e-9
    assert abs(mesh.radius - 1.05 * r) < 1e-12 and mesh.height > 0.3 and in_pcd is None
    assert len(mesh.sample_points_uniformly(500).points) == 500
    # rejection by max_radius returns five Nones
    assert fit.fit_shape_RANSAC(pts=pts, shape="circle", threshold=0.04, max_radius=0.01,
                                samples=samples) == (None,) * 5
    # seeded default sampling is reproducible
    a = fit.fit_shape_RANSAC(pts=pts, shape="circle", threshold=0.04, seed=5)
    b = fit.fit_shape_RANSAC(pts=pts, shape="circle", threshold=0.04, seed=5)
    assert np.array_equal(a[2], b[2]) and a[3] == b[3]


def test_cast_rays_wrapper_pinhole_and_areas(gpu):
    verts, tris = synth.canopy_mesh(4000, seed=8, side=0.5)
    out = rc.cast_rays((verts, tris), surf_2d=True)
    assert out["t_hit"].shape == (950, 1280) and out["hit"].any()
    t0, p0, _ = oracle.cast_rays(verts, tris, out["rays"].reshape(-1, 6))
    assert np.array_equal(out["t_hit"].reshape(-1), t0)
    assert np.array_equal(out["primitive_ids"].reshape(-1), p0)
    ids = np.unique(p0[np.isfinite(t0)])
    assert np.array_equal(out["hit_triangles"], ids)
    assert abs(out["surface_area_3d"] - len(ids) * 0.125) < 1e-3         # 0.5 m leaves: 0.125 m2
    assert 0 < out["surface_area_2d"] < out["surface_area_3d"]


def test_sparse_cast_and_occupancy(gpu):
    # a closed unit cube
    v = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], dtype=np.float32)
    f = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1],
                  [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4], [1, 5, 7], [1, 7, 3]], np.int32)
    q = np.array([[0.5, 0.4, 0.6], [0.2, 0.9, 0.1], [1.5, 0.4, 0.6], [-1, 0.4, 0.6]], np.float32)
    assert list(rc.get_points_inside_mesh((v, f), q)) == [1.0, 1.0, 0.0, 0.0]
    segs, pcd = rc.sparse_cast_w_intersections((v, f), num=4)
    assert segs.shape == (16, 2, 3)
    zs = np.sort(np.round(pcd.points[:, 2], 5))
    assert set(zs) <= {0.0, 1.0} and len(zs) >= 8                        # bottom and top faces


def test_fit_cyl_to_cluster(gpu):
    """qsm_generation.py:138-179 around fit_shape_RANSAC: good fit -> sampled cylinder + details;
    a radius far above the previous one -> rejected."""
    from pyqsm_amd.qsm_generation import fit_cyl_to_cluster
    pts = synth.ring_cluster(4000, radius=0.3, seed=4)
    cyls, details = [], []
    ok = fit_cyl_to_cluster(None, pts, 0.28, np.arange(len(pts)), cyls, details, seed=5)
    assert ok and len(cyls) == 1 and len(cyls[0].points) == 500
    assert abs(details[0]["radius"] - 0.3) < 0.01 and details[0]["height"] == pts[:, 2].min()
    assert np.allclose(details[0]["center"], pts.mean(0))
    cyls2, details2 = [], []
    assert not fit_cyl_to_cluster(None, pts, 0.05, np.arange(len(pts)), cyls2, details2, seed=5)
    assert cyls2 == [] and details2 == []


def test_project_to_image_and_birdseye(gpu):
    verts, tris = synth.canopy_mesh(3000, seed=9, side=0.5)
    eye = rc.birdseye((verts, tris))
    assert eye[2] > verts[:, 2].max() and abs(eye[0] - 0.5 * (verts[:, 0].min() + verts[:, 0].max())) < 1e-5
    cfg = {"fov_deg": 70, "center": list(0.5 * (verts.min(0) + verts.max(0))), "eye": eye, "up": [0, 1, 0],
           "width_px": 160, "height_px": 120}
    pcd, depth = rc.project_to_image((verts, tris), cfg)
    assert depth.shape == (120, 160) and np.isfinite(depth).sum() == len(pcd.points) > 100
    rays = rc.create_rays_pinhole(**cfg).reshape(-1, 6)
    t0, _, _ = oracle.cast_rays(verts, tris, rays)
    assert np.array_equal(depth.reshape(-1), t0)
    # every hit point lies on the mesh: distance to it ~ 0
    d, _ = oracle.point_mesh_distance(verts, tris, pcd.points.astype(np.float32))
    assert d.max() < 1e-4


def test_flat_imports_run_on_the_gpu(gpu):
    code = ("import numpy as np\n"
            "from math_utils.fit import cluster_DBSCAN\n"
            "P = np.random.default_rng(0).normal(size=(2000, 3))\n"
            "l, i, n = cluster_DBSCAN(np.arange(2000), P, 0.3, 5)\n"
            "print(len(l) > 0)")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "pyqsm_amd"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                       cwd="/tmp")
    assert r.returncode == 0 and r.stdout.strip() == "True", r.stderr


def test_hull_filters_keep_every_hull_vertex_and_the_oriented_bounds(gpu):
    """pyqsm_extreme_points / pyqsm_outside_halfspaces in front of Qhull (skeletonize.py:240-241,
    the clamp box of the contraction loop): the points listed as "not strictly inside" contain every
    vertex of the cloud's convex hull, so the hull, the PCA of its vertices and the bounds are THE
    SAME arrays as without the filter — for a forest, a ball, and a flat (degenerate) cloud."""
    from scipy.spatial import ConvexHull
    from pyqsm_amd.geometry import skeletonize as sk
    rng = np.random.default_rng(3)
    ball = rng.normal(size=(60_000, 3))
    ball *= (rng.uniform(0, 1, (60_000, 1)) ** (1 / 3)) / np.linalg.norm(ball, axis=1, keepdims=True)
    for P in (synth.forest(120_000, seed=6).astype(np.float64), ball):
        d = sk._HULL_DIRS
        ext = hip.extreme_points(P, d, device=gpu)
        proj = P @ d.T
        assert np.array_equal(proj[ext, np.arange(len(d))], proj.max(axis=0))
        want = ConvexHull(P).vertices
        got = sk._hull_vertices(P, device=gpu)
        assert np.array_equal(got, want)
        inner = ConvexHull(P[np.unique(ext)])
        cand = hip.outside_halfspaces(P, inner.equations, 1e-9 * np.abs(P).max(), device=gpu)
        assert np.all(np.diff(cand) > 0) and np.isin(want, cand).all() and len(cand) < 0.3 * len(P)
        lo0, hi0 = sk.oriented_bounds(P)
        lo1, hi1 = sk.oriented_bounds(P, device=gpu)
        assert np.array_equal(lo0, lo1) and np.array_equal(hi0, hi1)
    flat = np.column_stack([rng.uniform(0, 1, (30_000, 2)), np.zeros(30_000)])
    lo0, hi0 = sk.oriented_bounds(flat)
    lo1, hi1 = sk.oriented_bounds(flat, device=gpu)
    assert np.array_equal(lo0, lo1) and np.array_equal(hi0, hi1)
