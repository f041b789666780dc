"""Minimal stand-ins for the Open3D containers the hot path passes around.

The reference hands ``open3d.geometry.PointCloud`` / ``TriangleMesh`` objects to
its hot functions; Open3D is optional here. Every wrapper accepts NumPy arrays or
anything exposing ``.points`` (``np.asarray``-able) and returns these light
classes, which offer the few methods the reference's callers use on the results
(``select_by_index``, ``paint_uniform_color``, ``sample_points_uniformly`` ...).
"""
from __future__ import annotations

import numpy as np


def as_points(obj) -> np.ndarray:
    """float64 [n,3] view/copy of an array or of an object with ``.points``."""
    if hasattr(obj, "points"):
        obj = obj.points
    pts = np.asarray(obj, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError(f"expected [n,3] points, got shape {pts.shape}")
    return pts


class PointCloud:
    def __init__(self, points=None, colors=None):
        self.points = np.zeros((0, 3)) if points is None else np.asarray(points, dtype=np.float64)
        self.colors = colors

    def __len__(self):
        return len(self.points)

    def __repr__(self):
        return f"PointCloud with {len(self.points)} points."

    def select_by_index(self, idx, invert: bool = False) -> "PointCloud":
        idx = np.asarray(idx, dtype=np.int64)
        if invert:
            mask = np.ones(len(self.points), dtype=bool)
            mask[idx] = False
            return PointCloud(self.points[mask])
        return PointCloud(self.points[idx])

    def paint_uniform_color(self, rgb):
        self.colors = np.tile(np.asarray(rgb, dtype=np.float64), (len(self.points), 1))
        return self

    def get_center(self):
        return self.points.mean(axis=0)

    def get_min_bound(self):
        return self.points.min(axis=0)

    def get_max_bound(self):
        return self.points.max(axis=0)


class TriangleMesh:
    """vertices f64/f32 [V,3], triangles int [T,3]."""

    def __init__(self, vertices, triangles):
        self.vertices = np.asarray(vertices)
        self.triangles = np.asarray(triangles, dtype=np.int64)

    def get_center(self):
        return self.vertices.mean(axis=0)

    def get_surface_area(self) -> float:
        v = self.vertices.astype(np.float64)
        a, b, c = (v[self.triangles[:, k]] for k in range(3))
        return float(0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum())

    def select_by_triangle(self, tri_idx) -> "TriangleMesh":
        tris = self.triangles[np.asarray(tri_idx, dtype=np.int64)]
        used, inv = np.unique(tris, return_inverse=True)
        return TriangleMesh(self.vertices[used], inv.reshape(-1, 3))

    def select_by_index(self, vertex_idx) -> "TriangleMesh":
        """Open3D's legacy ``TriangleMesh.select_by_index``: the listed VERTICES, and every
        triangle whose three vertices are all among them (also triangles nobody asked for,
        on meshes that share vertices) in their original order."""
        vid = np.asarray(vertex_idx, dtype=np.int64).reshape(-1)
        new_of = np.full(len(self.vertices), -1, dtype=np.int64)
        first = np.unique(vid, return_index=True)[1]
        vid = vid[np.sort(first)]                      # duplicates keep their first position
        new_of[vid] = np.arange(len(vid))
        tris = new_of[self.triangles] if len(self.triangles) else np.zeros((0, 3), dtype=np.int64)
        keep = (tris >= 0).all(axis=1)
        return TriangleMesh(self.vertices[vid], tris[keep])


class Cylinder:
    """The primitive fit_shape_RANSAC returns (the reference builds an Open3D
    cylinder mesh at fit.py:322-332): centre, axis, radius, height."""

    def __init__(self, center, radius, height, axis=(0.0, 0.0, 1.0)):
        self.center = np.asarray(center, dtype=np.float64)
        self.radius = float(radius)
        self.height = float(height)
        ax = np.asarray(axis, dtype=np.float64)
        nrm = np.linalg.norm(ax)
        self.axis = ax / nrm if nrm > 0 else np.array([0.0, 0.0, 1.0])

    def __repr__(self):
        return (f"Cylinder(center={self.center.tolist()}, radius={self.radius:.4f}, "
                f"height={self.height:.4f}, axis={self.axis.tolist()})")

    def _frame(self):
        helper = np.array([1.0, 0, 0]) if abs(self.axis[0]) < 0.9 else np.array([0, 1.0, 0])
        u = np.cross(self.axis, helper)
        u /= np.linalg.norm(u)
        return u, np.cross(self.axis, u)

    def sample_points_uniformly(self, number_of_points: int = 100, seed: int = 0) -> PointCloud:
        """Points on the lateral surface (area-uniform)."""
        rng = np.random.default_rng(seed)
        ang = rng.uniform(0, 2 * np.pi, number_of_points)
        h = rng.uniform(-self.height / 2, self.height / 2, number_of_points)
        u, v = self._frame()
        pts = (self.center + self.radius * (np.cos(ang)[:, None] * u + np.sin(ang)[:, None] * v)
               + h[:, None] * self.axis)
        return PointCloud(pts)
