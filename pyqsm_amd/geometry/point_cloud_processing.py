"""The DBSCAN entry points of pyQSM/geometry/point_cloud_processing.py
(cluster_plus :169-203, cluster_and_get_largest :205-218) on the HIP kernels.

The reference calls Open3D's ``PointCloud.cluster_dbscan``; its result (noise -1,
clusters numbered in index order of their first core point, border points given
to the first cluster that reaches them) is the same labelling scikit-learn
produces, which is what the kernel is pinned against.
"""
from __future__ import annotations

import numpy as np

try:
    from .. import hip
    from .._shadow import fall_through
    from ..set_config import config, log
    from .cloud import PointCloud, as_points
except ImportError:  # flat import (pyqsm_amd/ on sys.path)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.set_config import config, log
    from pyqsm_amd.geometry.cloud import PointCloud, as_points

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)


def _select(pcd, pts, idx):
    if hasattr(pcd, "select_by_index"):
        return pcd.select_by_index(idx)
    return PointCloud(pts[idx])


def cluster_plus(pcd, eps=config["trunk"]["cluster_eps"], min_points=config["trunk"]["cluster_nn"],
                 draw_result=True, color_clusters=True, from_points=True, return_pcds=True,
                 ransac=False, device: int = 0, radius_inclusive: bool = True):
    """point_cloud_processing.py:169-203. ``from_points=True`` (the default) means
    ``pcd`` is an array of points. Returns the list of sub-clouds, one per label in
    ascending label order (noise -1 first when present), or ``{label: indices}``
    when ``return_pcds`` is false. ``draw_result`` / ``color_clusters`` are accepted
    and ignored (no GUI); ``ransac=True`` (Open3D plane segmentation) is out of scope.
    ``radius_inclusive=False`` switches the neighbourhood to d < eps — Open3D's compare if
    nanoflann's radius search is strict (unverifiable here: parity unpinned, default inclusive)."""
    if ransac:
        raise NotImplementedError("plane segmentation (ransac=True) is not part of the HIP hot path")
    pts = as_points(pcd)
    if from_points:
        pcd = PointCloud(pts)
    labels, _ = hip.dbscan(pts, eps, min_points, device=device, radius_inclusive=radius_inclusive)
    unique_lbs, counts = np.unique(labels, return_counts=True)
    log.info(f"point cloud has {counts} clusters")
    label_to_cluster = {ulabel: np.where(labels == ulabel)[0] for ulabel in unique_lbs}
    if return_pcds:
        return [_select(pcd, pts, idx_list) for idx_list in label_to_cluster.values()]
    return label_to_cluster


def cluster_and_get_largest(pcd, eps=config["trunk"]["cluster_eps"],
                            min_points=config["trunk"]["cluster_nn"], draw_clusters=False,
                            device: int = 0, radius_inclusive: bool = True):
    """point_cloud_processing.py:205-218: the sub-cloud of the most populous label
    (noise counts as a label, as in the reference)."""
    pts = as_points(pcd)
    labels, _ = hip.dbscan(pts, eps, min_points, device=device, radius_inclusive=radius_inclusive)
    log.info(f"point cloud has {labels.max() + 1 if len(labels) else 0} clusters")
    if len(labels) == 0:
        return _select(pcd, pts, np.zeros(0, dtype=np.int64))
    unique_vals, counts = np.unique(labels, return_counts=True)
    largest = unique_vals[np.argmax(counts)]
    return _select(pcd, pts, np.where(labels == largest)[0])
