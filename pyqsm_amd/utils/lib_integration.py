"""The KD-tree helpers of pyQSM/utils/lib_integration.py on the HIP radius kernels.

    find_neighbors_in_ball(base_pts, points_to_search, points_idxs, radius, center, use_top)
                                                        lib_integration.py:81-137
    get_neighbors_in_tree(sub_pcd_pts, full_tree, radius)      :73-79

``find_neighbors_in_ball`` is called once per branch segment by
``qsm_generation.sphere_step`` (:220); the reference rebuilds a KD-tree of the whole
cloud on every call, here it is one pass over the points on the GPU.
"""
from __future__ import annotations

import numpy as np

try:
    from .. import hip
    from .._shadow import fall_through
    from ..geometry.cloud import as_points
    from ..math_utils.general import get_center, get_percentile, get_radius
    from ..set_config import config, log
except ImportError:  # flat import (pyqsm_amd/ on sys.path)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.geometry.cloud import as_points
    from pyqsm_amd.math_utils.general import get_center, get_percentile, get_radius
    from pyqsm_amd.set_config import config, log

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)


class Sphere:
    """What the reference returns as an Open3D sphere mesh: centre and radius."""

    def __init__(self, center, radius):
        self.center = np.asarray(center, dtype=np.float64)
        self.radius = float(radius)


def find_neighbors_in_ball(base_pts, points_to_search, points_idxs, radius=None, center=None,
                           use_top=None, draw_results=False, device: int = 0):
    """lib_integration.py:81-137: one sphere around the centroid of ``base_pts`` (or of
    its top percentile band), radius = mean xy-radius x ``[sphere] radius_multiplier``
    clamped to ``[min_radius, max_radius]``; returns ``(sphere, neighbors, center,
    radius)`` where ``neighbors`` are the indices of ``points_to_search`` inside the
    sphere. (The reference then subtracts ``points_idxs`` in a loop that iterates over
    the integers of the result, which fails; the caller does the set difference itself,
    qsm_generation.py:221-226, so the plain index array is returned.)"""
    base_pts = as_points(base_pts)
    if use_top:
        top_idx, _ = get_percentile(base_pts, use_top[0], use_top[1])
        top_pts = base_pts[top_idx]
        if center is None:
            center = get_center(top_pts)
            center = [center[0], center[1], max(base_pts[:, 2])]
        if not radius:
            radius = get_radius(top_pts) * config["sphere"]["radius_multiplier"]
    else:
        if center is None:
            center = get_center(base_pts)
        if not radius:
            radius = get_radius(base_pts) * config["sphere"]["radius_multiplier"]
    radius = max(radius, config["sphere"]["min_radius"])
    radius = min(radius, config["sphere"]["max_radius"])
    log.info(f" Finding nbrs in ball w/ {radius=}, {center=}")
    neighbors = hip.ball_query(as_points(points_to_search), center, radius, device=device)
    return Sphere(center, radius), neighbors, center, radius


def get_neighbors_in_tree(sub_pcd_pts, full_tree, radius, device: int = 0):
    """lib_integration.py:73-79: indices of the points of ``full_tree`` (a SciPy KDTree
    or an array of points) within ``radius`` of any point of ``sub_pcd_pts``."""
    data = full_tree.data if hasattr(full_tree, "data") else full_tree
    data = as_points(data)
    # inclusive bound like query_ball_tree: widen the strict kernel bound by one ulp
    mask, _ = hip.radius_mark(data, as_points(sub_pcd_pts), np.nextafter(radius, np.inf),
                              k=len(data), device=device)
    return np.flatnonzero(mask)
