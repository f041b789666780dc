"""The neighbour-recovery helper of pyQSM/geometry/reconstruction.py on the HIP
radius kernel: ``get_neighbors_kdtree`` (:233-263)."""
from __future__ import annotations

import numpy as np

try:
    from .. import hip
    from .._shadow import fall_through
    from .cloud import PointCloud, as_points
except ImportError:  # flat import (pyqsm_amd/ on sys.path)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.geometry.cloud import PointCloud, as_points

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)


def get_neighbors_kdtree(src_pcd, query_pcd=None, query_pts=None, kd_tree=None, dist=0.05, k=500,
                         return_pcd=True, device: int = 0):
    """reconstruction.py:233-263: the points of ``src_pcd`` that are among the ``k``
    nearest neighbours within ``dist`` (strict, like SciPy's ``distance_upper_bound``)
    of some query point.

    ``return_pcd=True`` (the default) returns ``(pcd, counts, chained_nbrs)``: the
    sub-cloud, the number of neighbours each query selected (the reference returns the
    padded [m,k] index table here; its callers only use the other two values) and the
    ascending unique source indices; ``(None, None, None)`` when nothing is in range.
    ``return_pcd=False`` returns ``(dists, nbrs)``, the padded [m,k] tables of
    ``KDTree.query`` (missing entries: distance inf, index n), as canopy_metrics.py:238
    uses them. ``kd_tree`` is accepted and ignored."""
    if query_pcd is not None:
        query_pts = as_points(query_pcd)
    src_pts = as_points(src_pcd)
    if not return_pcd:
        return hip.radius_knn(src_pts, as_points(query_pts), dist, k=k, device=device)
    mask, counts = hip.radius_mark(src_pts, as_points(query_pts), dist, k=k, device=device)
    uniques = np.flatnonzero(mask)
    if len(uniques) == 0:
        return None, None, None
    pcd = (src_pcd.select_by_index(uniques) if hasattr(src_pcd, "select_by_index")
           else PointCloud(src_pts[uniques]))
    return pcd, counts, uniques
