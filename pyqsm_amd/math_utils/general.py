"""Helpers the RANSAC wrapper needs, with the reference's names and results
(pyQSM/math_utils/general.py). Vectorised NumPy; host-side, O(n), not on the
kernel path. Checked against tests/golden/general.npz (outputs of the
reference's own module)."""
from __future__ import annotations

import numpy as np

try:
    from .._shadow import fall_through
except ImportError:  # imported flat, with pyqsm_amd/ itself on sys.path (pyQSM's layout)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd._shadow import fall_through

# names pyQSM's module of the same name defines and this one does not: get_angles, ... (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)


def get_percentile(pts, low, high, axis=2, invert=False):
    """Indices (and values) strictly between the low and high percentiles of one
    coordinate (general.py:9-35). Like the reference, entries whose value is
    exactly 0 are dropped (it selects with ``np.where(vals)``)."""
    pts = np.asarray(pts)
    if isinstance(axis, list):
        vals = sum(pts[:, a] for a in axis)
        if invert:
            vals = pts[:, axis[0]] - pts[:, axis[1]]
    else:
        vals = pts[:, axis]
    lower = np.percentile(vals, low)
    upper = np.percentile(vals, high)
    keep = (vals != 0) & ~(vals <= lower)
    if high < 100:
        keep &= ~(vals >= upper)
    select_idxs = np.flatnonzero(keep)
    return select_idxs, vals[select_idxs]


def get_center(points, center_type="centroid"):
    """(x, y, z) centre: plain centroid, or centroid of the top / bottom decile in z
    (general.py:127-160)."""
    points = np.asarray(points)
    if center_type == "centroid":
        sel = points
    elif center_type == "top":
        sel = points[get_percentile(points, 90, 100)[0]]
    elif center_type == "bottom":
        sel = points[get_percentile(points, 0, 10)[0]]
    else:
        raise ValueError(f"unknown center_type {center_type!r}")
    return (np.average(sel[:, 0]), np.average(sel[:, 1]), np.average(sel[:, 2]))


def get_radius(points, center_type="centroid"):
    """Mean xy-distance from the centre (general.py:162-171)."""
    points = np.asarray(points)
    center = get_center(points, center_type)
    d = points[:, :2] - np.asarray(center[:2])
    return np.average(np.sqrt(np.sum(d ** 2, axis=1)))


def unit_vector(vector):
    vector = np.asarray(vector, dtype=np.float64)
    return vector / np.linalg.norm(vector)


def rotation_matrix_from_arr(a, b):
    """R with a @ R = b (Rodrigues; general.py:71-87). ``b`` must be a unit vector
    (or zero, which gives the identity)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    nb = np.linalg.norm(b)
    if nb == 0:
        return np.eye(3)
    if nb < 0.99 or nb > 1.01:
        raise ValueError("b must be a unit vector")
    v = np.cross(a, b)
    s = np.linalg.norm(v)
    c = np.dot(a, b)
    vx = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]]) * -1
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.eye(3) - vx + np.dot(-vx, -vx) * ((1 - c) / (s ** 2))
