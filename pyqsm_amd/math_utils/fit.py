"""DBSCAN clustering and RANSAC circle/cylinder fitting with pyQSM's names and
signatures (pyQSM/math_utils/fit.py), computed by the HIP kernels.

    cluster_DBSCAN(pts_idxs, points, eps, min_pts)         fit.py:217-250
    fit_shape_RANSAC(pcd, pts, threshold, lower_bound, ...) fit.py:253-339
    choose_and_cluster(new_neighbors, main_pts, ...)        fit.py:58-85 (DBSCAN branch)
    z_align_and_fit(pcd, axis_guess, **kwargs)              fit.py:23-45

plus the aliases BASELINE.json's north_star names: ``dbscan`` and ``fit_cylinder``.
Differences from the reference, all deliberate: no plotting, no ``breakpoint()``,
and RANSAC takes optional ``seed`` / ``samples`` / ``max_iterations`` keywords
(the reference draws its samples from Python's unseeded ``random``).
"""
from __future__ import annotations

import random

import numpy as np

try:
    from .. import hip
    from .._shadow import fall_through
    from ..geometry.cloud import Cylinder, PointCloud, as_points
    from ..set_config import config, log
    from .general import get_radius, rotation_matrix_from_arr, unit_vector
except ImportError:  # imported flat, with pyqsm_amd/ itself on sys.path (pyQSM's layout)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.geometry.cloud import Cylinder, PointCloud, as_points
    from pyqsm_amd.set_config import config, log
    from pyqsm_amd.math_utils.general import get_radius, rotation_matrix_from_arr, unit_vector

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)

# pyransac3d's default iteration counts (Circle.fit / Cylinder.fit signatures)
DEFAULT_ITERATIONS = {"circle": 1000, "cylinder": 10000}


def dbscan(points, eps=None, min_pts=None, device: int = 0):
    """labels int64 [n] (-1 = noise) and core mask bool [n] — what
    ``sklearn.cluster.DBSCAN(eps, min_samples).fit(points)`` exposes as ``labels_``
    and ``core_sample_indices_`` (fit.py:223-224)."""
    eps = config["dbscan"]["epsilon"] if eps is None else eps
    min_pts = config["dbscan"]["min_neighbors"] if min_pts is None else min_pts
    return hip.dbscan(as_points(points), eps, min_pts, device=device)


def _group_labels(labels, core, pts_idxs):
    """What fit.py:224-246 computes with ``set(labels)`` and one ``labels == k`` scan per
    cluster, in O(n): the label set (same elements, same iteration order: labels are inserted in
    order of first appearance, which is all that a set built from the whole array depends on),
    the core members of every cluster in that order, and the non-core points labelled -1."""
    n = len(labels)
    if n == 0:
        return set(), [], []
    k = int(labels.max()) + 2                     # labels are -1 .. k-2
    shifted = (labels + 1).astype(np.int64)
    counts = np.bincount(shifted, minlength=k)
    first = np.zeros(k, dtype=np.int64)
    first[shifted[::-1]] = np.arange(n - 1, -1, -1)   # last write wins: the smallest position
    present = np.flatnonzero(counts)
    unique_labels = set()
    for v in present[np.argsort(first[present], kind="stable")]:
        unique_labels.add(np.int64(v - 1))
    # members of every label in ascending position (what np.where yields), via one stable sort
    key = shifted.astype(np.int16) if k < 32000 else shifted
    order = np.argsort(key, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(counts)])
    idxs, noise = [], []
    for lab in unique_labels:
        members = order[bounds[lab + 1]:bounds[lab + 2]]
        if lab == -1:
            noise = pts_idxs[members[~core[members]]]
        else:
            idxs.append(pts_idxs[members[core[members]]])
    return unique_labels, idxs, noise


def cluster_DBSCAN(pts_idxs, points, eps, min_pts, device: int = 0):
    """fit.py:217-250 (``device`` is this package's addition: the GPU to run on, so that whole
    clusters can be distributed over the GPUs of a node). Returns
    ``(unique_labels, idxs, noise)``:
    ``unique_labels`` the set of labels (including -1 when present); ``idxs`` one
    array of caller indices per non-noise label, in set-iteration order, holding
    that cluster's CORE samples only; ``noise`` the caller indices with label -1."""
    labels, core = hip.dbscan(as_points(points), eps, min_pts, device=device)
    pts_idxs = np.asarray(pts_idxs)
    unique_labels, idxs, noise = _group_labels(labels, core.astype(bool), pts_idxs)
    num_clusters = len(unique_labels) - (1 if -1 in unique_labels else 0)
    num_noise = int(len(labels) - np.count_nonzero(labels + 1))
    log.info(f"Estimated number of clusters: {num_clusters}")
    log.info("Estimated number of noise points: %d" % num_noise)
    return unique_labels, idxs, noise


def choose_and_cluster(new_neighbors, main_pts, cluster_type="DBSCAN", debug=False):
    """fit.py:58-85. Only the DBSCAN branch is on the hot path; ``"kmeans"`` (a
    SciPy kmeans2 + silhouette heuristic with plotting, fit.py:168-214) is outside
    the scope of this package and raises."""
    if cluster_type == "kmeans":
        raise NotImplementedError("the k-means branch of choose_and_cluster is not part of the "
                                  "HIP hot path (SURVEY.md §2); pass cluster_type='DBSCAN'")
    nn_points = np.asarray(main_pts)[new_neighbors]
    log.info("clustering via DBSCAN")
    labels, returned_clusters, _noise = cluster_DBSCAN(
        new_neighbors, nn_points, eps=config["dbscan"]["epsilon"],
        min_pts=config["dbscan"]["min_neighbors"])
    return labels, returned_clusters


def draw_samples(n_points: int, iterations: int, seed=None, method: str = "numpy") -> np.ndarray:
    """``iterations`` rows of 3 distinct indices, each row uniform over the ordered triples — the
    distribution of pyransac3d's ``random.sample(range(n), 3)`` per iteration (the reference is
    unseeded, so no particular stream is part of its behaviour). ``method="numpy"`` draws all rows
    at once (a per-row ``random.sample`` loop costs 3 ms per 1000 hypotheses, ten times the fit
    itself on a stem slice); ``method="stdlib"`` is that loop, from a private ``random.Random``.
    ``seed`` may be an int, a ``numpy.random.SeedSequence`` or a ``numpy.random.Generator`` (the
    batch call hands every point set its own spawned stream). Since round 2 the default stream
    is NumPy's, so a seed used with round 1's ``random.Random`` draws yields other fits."""
    if n_points < 3:
        raise ValueError("need at least three points to draw a hypothesis")
    if method == "stdlib":
        rng = random.Random(seed)
        return np.array([rng.sample(range(n_points), 3) for _ in range(iterations)],
                        dtype=np.int64).reshape(-1, 3)
    if method != "numpy":
        raise ValueError("method must be 'numpy' or 'stdlib'")
    rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)
    a = rng.integers(0, n_points, iterations)
    b = rng.integers(0, n_points - 1, iterations)
    b += b >= a                                          # uniform over the n-1 values other than a
    c = rng.integers(0, n_points - 2, iterations)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    c += c >= lo                                         # ... and over the n-2 other than a and b
    c += c >= hi
    return np.stack([a, b, c], axis=1).astype(np.int64)


_NO_FIT = (None, None, None, None, None)


def _ransac_points(pcd, pts, lower_bound, shape):
    """The points a fit runs on (fit.py:262-276): z clamped IN PLACE in the caller's array when
    ``lower_bound`` is truthy, z zeroed in the copy that is fitted for a circle."""
    if pts is None:
        pts = np.asarray(pcd.points)
    if lower_bound:
        low = pts[:, 2] < lower_bound
        pts[low, 2] = lower_bound                       # mutates the caller's array, like :266-268
    if shape not in ("circle", "cylinder"):
        raise ValueError(f"shape must be 'circle' or 'cylinder', got {shape!r}")
    if len(pts) < 3:
        return pts, None
    _ = get_radius(pts)                                 # :272 (computed, unused by the reference)
    fit_pts = np.array(pts, dtype=np.float64)           # pts.copy()
    if shape == "circle":
        fit_pts[:, 2] = 0.0                             # :274-276
    return pts, fit_pts


def _ransac_result(pcd, pts, shape, center, axis, fit_radius, inliers, best, max_radius):
    """fit.py:285-339: rejections, then the cylinder through the inliers' z range."""
    log.info(f"fit_cyl = center: {center}, axis: {axis}, radius: {fit_radius}")
    if best < 0:                                        # pyransac3d leaves center == [] (:291)
        log.info(f"no no fit {shape} found")
        return _NO_FIT
    if max_radius is not None and fit_radius > max_radius:
        log.info(f"{shape} had radius {fit_radius} but max_radius is {max_radius}")
        return _NO_FIT
    in_pts = pts[inliers]
    lowest, highest = in_pts[:, 2].min(), in_pts[:, 2].max()
    height = highest - lowest
    test_center = [center[0], center[1], (height / 2) + lowest]
    if height <= 0:
        return _NO_FIT
    cyl_mesh = Cylinder(test_center, fit_radius * 1.05, height, axis)   # :321-332
    in_pcd = None
    if pcd is not None:
        in_pcd = (pcd.select_by_index(inliers) if hasattr(pcd, "select_by_index")
                  else PointCloud(np.asarray(pcd.points)[inliers]))
    return cyl_mesh, in_pcd, inliers, fit_radius, axis


def fit_shape_RANSAC(pcd=None, pts=None, threshold=0.1, lower_bound=None, max_radius=None,
                     align_to_z=False, shape="circle", seed=None, samples=None,
                     max_iterations=None, device: int = 0, **kwargs):
    """fit.py:253-339. Returns ``(cyl_mesh, in_pcd, inliers, fit_radius, axis)`` or
    five ``None`` when the fit is rejected.

    ``cyl_mesh`` is a :class:`Cylinder` (the reference builds an Open3D mesh of the
    same centre / radius*1.05 / height / axis). As in the reference, a truthy
    ``lower_bound`` clamps z IN PLACE in the caller's array (fit.py:265-268)."""
    pts, fit_pts = _ransac_points(pcd, pts, lower_bound, shape)
    if fit_pts is None:
        log.info(f"no no fit {shape} found")
        return _NO_FIT
    if samples is None:
        iters = max_iterations or DEFAULT_ITERATIONS[shape]
        samples = draw_samples(len(fit_pts), iters, seed)
    center, axis, fit_radius, inliers, best = hip.ransac(fit_pts, samples, shape, threshold,
                                                         device=device)
    return _ransac_result(pcd, pts, shape, center, axis, fit_radius, inliers, best, max_radius)


def fit_shape_RANSAC_batch(pts_list, threshold=0.1, lower_bound=None, max_radius=None, shape="circle",
                           seed=None, samples=None, max_iterations=None, device: int = 0):
    """:func:`fit_shape_RANSAC` for many point sets in ONE pass through the GPU
    (``pyqsm_ransac_batch``) — the z-slices of a stem, the clusters of a scan. The reference fits
    one cluster per call (qsm_generation.py:150); a call costs 0.6 ms of launches and round trips
    for 0.05 ms of work, so a thousand slices are twelve times faster together.

    ``pts_list``: arrays [n_i,3] (clamped in place by a truthy ``lower_bound`` like the single
    call); ``lower_bound`` / ``max_radius``: one value or one per set; ``samples``: one int64
    [H,3] array per set (same H), or None to draw ``max_iterations`` rows per set. The sets get
    INDEPENDENT streams spawned from ``seed`` (``SeedSequence(seed).spawn(S)``: set q always gets
    child q, whatever the other sets are) — one shared seed would hand sets of equal size the
    same hypothesis triples and correlate their fits.
    Returns one 5-tuple of :func:`fit_shape_RANSAC` per set (``in_pcd`` is None: arrays in)."""
    S = len(pts_list)

    def per_set(v):
        return list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v] * S
    lbs, mrs = per_set(lower_bound), per_set(max_radius)
    iters = max_iterations or DEFAULT_ITERATIONS[shape]
    kept, fit, tri = [], [], []
    out = [_NO_FIT] * S
    srcs = []
    streams = np.random.SeedSequence(seed).spawn(S) if samples is None else None
    for q in range(S):
        pts, fit_pts = _ransac_points(None, pts_list[q], lbs[q], shape)
        srcs.append(pts)
        if fit_pts is None:
            log.info(f"no no fit {shape} found")
            continue
        rows = draw_samples(len(fit_pts), iters, streams[q]) if samples is None else np.asarray(samples[q])
        kept.append(q)
        fit.append(fit_pts)
        tri.append(np.ascontiguousarray(rows, dtype=np.int64).reshape(-1, 3))
    if not kept:
        return out
    if len({len(t) for t in tri}) != 1:
        raise ValueError("every set needs the same number of sample rows")
    seg = np.concatenate([[0], np.cumsum([len(f) for f in fit])]).astype(np.int64)
    centers, axes, radii, inliers, best = hip.ransac_batch(np.concatenate(fit), seg, np.stack(tri), shape,
                                                           threshold, device=device)
    for j, q in enumerate(kept):
        out[q] = _ransac_result(None, srcs[q], shape, centers[j], axes[j], float(radii[j]), inliers[j],
                                int(best[j]), mrs[q])
    return out


def fit_cylinder(pts, threshold=0.04, lower_bound=None, max_radius=None, shape="circle", **kwargs):
    """north_star alias: the call fit_cyl_to_cluster makes (qsm_generation.py:149-155)."""
    return fit_shape_RANSAC(pts=pts, threshold=threshold, lower_bound=lower_bound,
                            max_radius=max_radius, shape=shape, **kwargs)


def z_align_and_fit(pcd, axis_guess, **kwargs):
    """fit.py:23-45: rotate the cloud so that ``axis_guess`` is +z, fit a circle to
    its projection. Returns the 5-tuple of :func:`fit_shape_RANSAC`."""
    R_to_z = rotation_matrix_from_arr(unit_vector(axis_guess), [0, 0, 1])
    pts = as_points(pcd)
    center = pts.mean(axis=0)
    # Open3D's pcd.rotate(R) rotates about the cloud centre: p' = R (p - c) + c
    rotated = (pts - center) @ np.asarray(R_to_z).T + center
    mesh, _, inliers, fit_radius, _ax = fit_shape_RANSAC(pcd=PointCloud(rotated), **kwargs)
    if mesh is None:
        log.warning("No mesh found")
    return mesh, _, inliers, fit_radius, _ax
