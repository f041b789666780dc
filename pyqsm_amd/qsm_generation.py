"""The RANSAC caller of ``pyQSM/qsm_generation.py``.

Only ``fit_cyl_to_cluster`` (``qsm_generation.py:138-179``) lives here: the function through
which the sphere-stepping QSM builder reaches ``fit_shape_RANSAC``. The stepping driver itself
(``sphere_step``, file IO, drawing) is outside the hot-path scope.
"""
from __future__ import annotations

import logging

import numpy as np

try:  # flat import style of the reference (pyqsm_amd on sys.path) or package import
    from ._shadow import fall_through
    from .math_utils.fit import fit_shape_RANSAC
    from .math_utils.general import get_center
    from .set_config import config
except ImportError:  # pragma: no cover
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.math_utils.fit import fit_shape_RANSAC
    from pyqsm_amd.math_utils.general import get_center
    from pyqsm_amd.set_config import config

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)

log = logging.getLogger("calc")


def fit_cyl_to_cluster(main_pcd, curr_pts, last_radius, cluster_idxs, cyls=[], cyl_details=[],
                       debug=False, **ransac_kwargs):
    """qsm_generation.py:138-179: fit a circle to the z-projection of ``curr_pts`` (threshold
    0.04, points clamped up to the lowest z, radius at most ``last_radius *
    config['sphere']['radius_multiplier']``); the fit is good when a cylinder came back and its
    radius is below ``bad_fit_radius_factor * last_radius``. A good fit appends 500 points
    sampled on the cylinder to ``cyls`` and ``{center, axis, height, radius}`` to
    ``cyl_details`` (the caller's lists, mutated as in the reference). Returns the flag.
    ``debug`` drew and stopped in the debugger there; it is accepted and ignored.
    ``ransac_kwargs`` (``seed=``, ``samples=``) reach ``fit_shape_RANSAC``."""
    curr_pts = np.asarray(curr_pts)
    log.info("Attempting to fit a 2D circle to projection of points")
    prev_neighbor_height = np.min(curr_pts[:, 2])
    cyl_mesh, fit_pcd, inliers, fit_radius, axis = fit_shape_RANSAC(
        pts=curr_pts,
        shape="circle",
        threshold=0.04,
        lower_bound=prev_neighbor_height,
        max_radius=last_radius * config["sphere"]["radius_multiplier"],
        **ransac_kwargs,
    )
    good_fit_found = (cyl_mesh is not None
                      and fit_radius < config["sphere"]["bad_fit_radius_factor"] * last_radius)
    if good_fit_found:
        log.info("good fit found, adding cyl to list")
        cyls.append(cyl_mesh.sample_points_uniformly(500))
        cyl_details.append({"center": get_center(curr_pts), "axis": axis,
                            "height": prev_neighbor_height, "radius": fit_radius})
    return good_fit_found
