"""Ray casting against a triangle mesh with pyQSM's names (pyQSM/viz/ray_casting.py)
on the HIP sweep instead of Open3D's RaycastingScene (Embree).

    cast_rays(tmesh, surf_2d, img, pinhole_config)    :262-313
    raycast_to_pcd(mesh, pinhole_config)              :315-330
    sparse_cast_w_intersections(mesh)                 :151-192
    get_points_inside_mesh(...)                       :53-71   (occupancy)
    create_rays_pinhole(...)       the RaycastingScene static the reference calls
    RaycastingScene                a small class with add_triangles / cast_rays /
                                   list_intersections / compute_occupancy /
                                   compute_distance / compute_signed_distance

Meshes may be ``(vertices, triangles)`` tuples, objects with ``.vertices`` /
``.triangles`` (Open3D legacy layout) or with ``.vertex['positions']`` /
``.triangle['indices']`` (Open3D tensor layout). Nothing here draws, plots or
stops in a debugger.
"""
from __future__ import annotations

import numpy as np

try:
    from .. import hip
    from .._shadow import fall_through
    from ..geometry.cloud import PointCloud, TriangleMesh
    from ..set_config import log
except ImportError:  # flat import (pyqsm_amd/ on sys.path)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.geometry.cloud import PointCloud, TriangleMesh
    from pyqsm_amd.set_config import log

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)

pinhole_config = {"fov_deg": 60, "center": [-3, -.25, -3], "eye": [10, 10, 20], "up": [0, 0, 1],
                  "width_px": 640, "height_px": 480}       # ray_casting.py:45-47


def _np(a):
    return a.numpy() if hasattr(a, "numpy") else np.asarray(a)


def mesh_arrays(mesh):
    """(verts float32 [V,3], tris int32 [T,3]) from any supported mesh object."""
    if isinstance(mesh, (tuple, list)) and len(mesh) == 2:
        v, t = mesh
    elif hasattr(mesh, "vertex") and hasattr(mesh, "triangle"):
        v, t = _np(mesh.vertex["positions"]), _np(mesh.triangle["indices"])
    elif hasattr(mesh, "vertices") and hasattr(mesh, "triangles"):
        v, t = mesh.vertices, mesh.triangles
    else:
        raise TypeError("unsupported mesh object: need (verts, tris), .vertices/.triangles or "
                        ".vertex['positions']/.triangle['indices']")
    return (np.ascontiguousarray(_np(v), dtype=np.float32).reshape(-1, 3),
            np.ascontiguousarray(_np(t), dtype=np.int32).reshape(-1, 3))


def create_rays_pinhole(fov_deg, center, eye, up, width_px, height_px) -> np.ndarray:
    """Rays of a pinhole camera, float32 [height, width, 6] = (origin, direction);
    directions are not normalised (pixel plane at depth 1), as Open3D's
    ``RaycastingScene.create_rays_pinhole`` returns them (used at ray_casting.py:277)."""
    center, eye, up = (np.asarray(_np(a), dtype=np.float64) for a in (center, eye, up))
    focal = 0.5 * width_px / np.tan(0.5 * np.deg2rad(fov_deg))
    cx, cy = 0.5 * width_px, 0.5 * height_px
    R = np.zeros((3, 3))
    R[1] = up / np.linalg.norm(up)
    R[2] = center - eye
    R[2] /= np.linalg.norm(R[2])
    R[0] = np.cross(R[1], R[2])
    R[0] /= np.linalg.norm(R[0])
    R[1] = np.cross(R[2], R[0])
    xs = (np.arange(width_px) + 0.5 - cx) / focal
    ys = (np.arange(height_px) + 0.5 - cy) / focal
    gx, gy = np.meshgrid(xs, ys, indexing="xy")
    cam = np.stack([gx, gy, np.ones((height_px, width_px))], axis=-1)
    dirs = cam @ R                        # R^T applied to each camera-space direction
    rays = np.empty((height_px, width_px, 6), dtype=np.float32)
    rays[..., :3] = eye
    rays[..., 3:] = dirs
    return rays


class RaycastingScene:
    """The subset of ``open3d.t.geometry.RaycastingScene`` the reference uses."""

    def __init__(self, device: int = 0, n_devices: int | None = None):
        self.device = device
        self.n_devices = n_devices       # not None: sweep on that many GPUs of this process (0 = all)
        self._verts = np.zeros((0, 3), dtype=np.float32)
        self._tris = np.zeros((0, 3), dtype=np.int32)

    def add_triangles(self, mesh, triangles=None) -> int:
        v, t = mesh_arrays((mesh, triangles) if triangles is not None else mesh)
        self._tris = np.concatenate([self._tris, t + len(self._verts)], axis=0)
        self._verts = np.concatenate([self._verts, v], axis=0)
        return 0

    create_rays_pinhole = staticmethod(create_rays_pinhole)

    def cast_rays(self, rays) -> dict:
        """{'t_hit' f32 (+inf = miss), 'primitive_ids' u32 (0xFFFFFFFF = miss),
        'primitive_uvs' f32 [...,2], 'geometry_ids' u32}."""
        if self.n_devices is not None:
            t, p, uv = hip.cast_rays_multi(self._verts, self._tris, _np(rays), self.n_devices)
        else:
            t, p, uv = hip.cast_rays(self._verts, self._tris, _np(rays), device=self.device)
        geo = np.where(np.isfinite(t), 0, 0xFFFFFFFF).astype(np.uint32)
        return {"t_hit": t, "primitive_ids": p, "primitive_uvs": uv, "geometry_ids": geo}

    def list_intersections(self, rays) -> dict:
        return hip.list_intersections(self._verts, self._tris, _np(rays), device=self.device)

    def count_intersections(self, rays) -> np.ndarray:
        r = _np(rays)
        return self.list_intersections(r.reshape(-1, 6))["counts"].reshape(r.shape[:-1])

    def compute_occupancy(self, query_points) -> np.ndarray:
        """1.0 for points inside a closed mesh (odd number of crossings along +x)."""
        q = np.ascontiguousarray(_np(query_points), dtype=np.float32)
        rays = np.concatenate([q.reshape(-1, 3), np.tile(np.float32([1, 0, 0]), (q.size // 3, 1))],
                              axis=1)
        counts = self.count_intersections(rays)
        return (counts % 2).astype(np.float32).reshape(q.shape[:-1])

    def compute_distance(self, query_points) -> np.ndarray:
        """Unsigned distance to the surface, float32, shape of the input minus its last axis."""
        q = np.ascontiguousarray(_np(query_points), dtype=np.float32)
        dist, _ = hip.point_mesh_distance(self._verts, self._tris, q.reshape(-1, 3), device=self.device)
        return dist.reshape(q.shape[:-1])

    def compute_signed_distance(self, query_points) -> np.ndarray:
        """Distance to the surface, negative inside a closed mesh (occupancy by crossing parity,
        as ``compute_occupancy``); what ray_casting.py:250,256 asks of the scene."""
        q = np.ascontiguousarray(_np(query_points), dtype=np.float32)
        dist = self.compute_distance(q)
        inside = self.compute_occupancy(q) > 0.5
        return np.where(inside, -dist, dist).astype(np.float32)


def mri(mesh=None, rcs_in=None, n_random: int = 256, grid: int = 64, device: int = 0):
    """ray_casting.py:237-260: signed distances of ``n_random`` uniform random points in the
    mesh's bounding box and of a ``grid``^3 lattice over it. The reference then shows 32
    slices with matplotlib and stops in the debugger; this returns
    ``(random_points, their signed distances, lattice points [g,g,g,3], signed distances [g,g,g])``
    instead."""
    if rcs_in is not None:
        scene = rcs_in
    elif mesh is not None:
        scene = RaycastingScene(device)
        scene.add_triangles(mesh)
    else:
        raise ValueError("No mesh or rcs provided for mri")
    verts = scene._verts if mesh is None else mesh_arrays(mesh)[0]
    min_bound, max_bound = verts.min(0), verts.max(0)
    query_points = np.random.uniform(low=min_bound, high=max_bound, size=[n_random, 3]).astype(np.float32)
    sd_random = scene.compute_signed_distance(query_points)
    xyz_range = np.linspace(min_bound, max_bound, num=grid)
    lattice = np.stack(np.meshgrid(*xyz_range.T), axis=-1).astype(np.float32)
    return query_points, sd_random, lattice, scene.compute_signed_distance(lattice)


def cast_rays(tmesh, surf_2d: bool = False, img: bool = False, pinhole_config=pinhole_config,
              rays=None, device: int = 0, n_devices: int | None = None) -> dict:
    """ray_casting.py:262-313: look down on the mesh from 10 units above its centre
    (fov 90 deg, 1280 x 950 px, up (0,1,-1)) and cast one ray per pixel.

    Returns the ``cast_rays`` dict ('t_hit', 'primitive_ids', 'primitive_uvs') plus
    'hit' (bool mask) and 'rays'; with ``surf_2d`` also 'hit_triangles' (unique hit
    triangle ids), 'hit_mesh', 'surface_area_3d' and 'surface_area_2d' (that mesh
    flattened to z = 0), the quantities of :285-303. As in the reference (:286-292)
    'hit_mesh' is selected BY VERTEX: ``select_by_index(unique vertices of the hit
    triangles)`` keeps every triangle whose three vertices were all hit, which on a mesh
    with shared vertices includes triangles no ray reached. (The reference
    returns an undefined name when ``surf_2d`` is false.) ``rays`` overrides the
    camera, e.g. with parallel sun rays. ``img`` is accepted and ignored. ``n_devices``
    (0 = all visible GPUs) shards the rays over that many GPUs of this process through RCCL
    (``pyqsm_cast_rays_multi``; identical results)."""
    verts, tris = mesh_arrays(tmesh)
    log.info("starting cast rays")
    if rays is None:
        center = verts.mean(axis=0) if not hasattr(tmesh, "get_center") else _np(tmesh.get_center())
        eye = [center[0], center[1], center[2] + 10]
        cfg = {"fov_deg": 90, "center": center, "eye": list(eye), "up": [0, 1, -1],
               "width_px": 640 * 2, "height_px": 475 * 2}             # :272-273
        rays = create_rays_pinhole(**cfg)
    log.info("casting rays")
    scene = RaycastingScene(device, n_devices)
    scene.add_triangles((verts, tris))
    ans = scene.cast_rays(rays)
    hit = np.isfinite(ans["t_hit"])                                    # :280
    out = dict(ans, hit=hit, rays=np.asarray(rays))
    if surf_2d:
        log.info("getting surface area")
        tri_ids = np.unique(ans["primitive_ids"][hit])                 # :286-289
        hit_vert_ids = np.unique(tris[tri_ids.astype(np.int64)])       # :290
        hit_mesh = TriangleMesh(verts, tris).select_by_index(hit_vert_ids)   # :291
        flat = TriangleMesh(hit_mesh.vertices * np.float32([1, 1, 0]), hit_mesh.triangles)
        out.update(hit_triangles=tri_ids, hit_mesh=hit_mesh,
                   surface_area_3d=hit_mesh.get_surface_area(),       # :292
                   surface_area_2d=flat.get_surface_area())           # :298-301
    return out


def raycast_to_pcd(mesh, pinhole_config=pinhole_config, device: int = 0):
    """ray_casting.py:315-330: the hit points of a pinhole view as a point cloud.
    Returns (pcd, t_hit of the hits)."""
    scene = RaycastingScene(device)
    scene.add_triangles(mesh)
    rays = create_rays_pinhole(**pinhole_config)
    ans = scene.cast_rays(rays)
    hit = np.isfinite(ans["t_hit"])
    hits = rays[hit]
    points = hits[:, :3] + hits[:, 3:] * ans["t_hit"][hit].reshape((-1, 1))   # :322
    return PointCloud(points), ans["t_hit"][hit]


def birdseye(mesh):
    """ray_casting.py:194-203: an eye position above the middle of the mesh's bounding box, half
    its height above the top."""
    v, _ = mesh_arrays(mesh)
    ub, lb = v.max(0).astype(np.float64), v.min(0).astype(np.float64)
    half_diff = (ub - lb) / 2
    mid = ub - half_diff
    return [float(mid[0]), float(mid[1]), float(ub[2] + half_diff[2])]


def project_to_image(mesh, pinhole_config=pinhole_config, device: int = 0):
    """ray_casting.py:205-235: the pinhole view of the mesh. Returns ``(pcd, depth)``: the hit
    points as a cloud and the ``t_hit`` image float32 [height, width] (+inf where nothing is
    hit) that the reference shows with ``plt.imshow`` before stopping in the debugger."""
    scene = RaycastingScene(device)
    scene.add_triangles(mesh)
    rays = create_rays_pinhole(**pinhole_config)
    ans = scene.cast_rays(rays)
    hit = np.isfinite(ans["t_hit"])
    hits = rays[hit]
    points = hits[:, :3] + hits[:, 3:] * ans["t_hit"][hit].reshape((-1, 1))   # :226
    return PointCloud(points), ans["t_hit"]


def sparse_cast_w_intersections(mesh, num: int = 10, device: int = 0):
    """ray_casting.py:151-192: a num x num grid of +z rays from below the bounding
    box; returns (ray segments [n,2,3], intersection points [m,3]) — every crossing,
    located from the barycentric coordinates exactly as at :172-180."""
    verts, tris = mesh_arrays(mesh)
    bb_min, bb_max = verts.min(axis=0), verts.max(axis=0)
    x, y = np.linspace(bb_min, bb_max, num=num)[:, :2].T
    xv, yv = np.meshgrid(x, y)
    orig = np.stack([xv, yv, np.full_like(xv, bb_min[2] - 1)], axis=-1).reshape(-1, 3)
    dest = orig + np.full(orig.shape, (0, 0, 2 + bb_max[2] - bb_min[2]), dtype=np.float32)
    rays = np.concatenate([orig, dest - orig], axis=-1).astype(np.float32)
    scene = RaycastingScene(device)
    scene.add_triangles((verts, tris))
    lx = scene.list_intersections(rays)
    tidx, uv = lx["primitive_ids"], lx["primitive_uvs"]
    w = 1 - np.sum(uv, axis=1)
    pts = (verts[tris[tidx, 1]] * uv[:, 0][:, None] + verts[tris[tidx, 2]] * uv[:, 1][:, None]
           + verts[tris[tidx, 0]] * w[:, None])
    return np.stack([orig, dest], axis=1), PointCloud(pts)


def get_points_inside_mesh(mesh, query_pts, device: int = 0) -> np.ndarray:
    """Occupancy of query points in a closed mesh (ray_casting.py:53-71 builds a
    cylinder mesh and calls ``compute_occupancy``)."""
    scene = RaycastingScene(device)
    scene.add_triangles(mesh)
    return scene.compute_occupancy(query_pts)


def interception_layers(mesh, rays, max_rounds=None, device: int = 0):
    """The "metrics with overlap" of data/notes/methods.md:53-55: cast the rays, sum the area of
    the triangles that intercept them first, remove those triangles, and repeat until no ray
    hits anything (or ``max_rounds`` is reached). The reference describes the procedure and
    keeps no code for it; this is its straightforward reading (``oracle.interception_layers``
    restates it on the CPU; parity with the reference itself is unpinned).

    Returns ``(areas, layer)``: ``areas[r]`` is the summed area (float64, from the float32
    vertices) of the triangles removed in round r, ``layer[t]`` the round in which triangle t
    was removed (-1: never hit). The rays stay in HBM for all rounds; each round uploads the
    surviving triangles, sweeps (``pyqsm_cast_rays_dev``) and reads back the primitive ids."""
    verts, tris = mesh_arrays(mesh)
    r = np.ascontiguousarray(np.asarray(rays), dtype=np.float32).reshape(-1, 6)
    R, T = r.shape[0], tris.shape[0]
    e1 = verts[tris[:, 1]].astype(np.float64) - verts[tris[:, 0]].astype(np.float64)
    e2 = verts[tris[:, 2]].astype(np.float64) - verts[tris[:, 0]].astype(np.float64)
    tri_area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    layer = np.full(T, -1, dtype=np.int64)
    areas = []
    if R == 0 or T == 0:
        return areas, layer
    alive = np.arange(T)
    d_rays = hip.DeviceBuffer.from_array(r, device)
    d_t = hip.DeviceBuffer(R * 4, device)
    d_prim = hip.DeviceBuffer(R * 4, device)
    try:
        rnd = 0
        while len(alive) and (max_rounds is None or rnd < max_rounds):
            dm = hip.DeviceMesh(verts, tris[alive], device)
            hip.cast_rays_dev(dm, d_rays.ptr, R, d_t.ptr, d_prim.ptr)
            prim = d_prim.download((R,), np.uint32)          # synchronises
            dm.records.free()
            mark = np.zeros(len(alive) + 1, dtype=bool)       # (np.unique would sort 10^7 ids)
            mark[np.minimum(prim, len(alive))] = True        # misses (0xFFFFFFFF) land in the spare slot
            hit = np.flatnonzero(mark[:-1])
            if len(hit) == 0:
                break
            ids = alive[hit]
            layer[ids] = rnd
            areas.append(float(tri_area[ids].sum()))
            alive = np.delete(alive, hit)
            rnd += 1
    finally:
        for b in (d_rays, d_t, d_prim):
            b.free()
    return areas, layer
