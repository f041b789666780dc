"""Synthetic inputs of the shapes BASELINE.json names.

No dataset ships with the reference and there is no network, so benchmarks and
parity tests run on these generators. Coordinates are rounded to
float32-representable values and handed out as float64, so that CPU and GPU
predicates see identical numbers.
"""
from __future__ import annotations

import numpy as np

TREE_UNIT = 50_000  # points per synthetic tree


def _cylinder(rng, n, radius, length, sigma):
    """n points on a cylinder of the given radius around +z, jittered."""
    ang = rng.uniform(0.0, 2.0 * np.pi, n)
    h = rng.uniform(0.0, length, n)
    r = radius + rng.normal(0.0, sigma, n)
    return np.stack([r * np.cos(ang), r * np.sin(ang), h], axis=1)


def tree_unit(seed: int = 0, n: int = TREE_UNIT) -> np.ndarray:
    """One trunk + 5 branches + 1 % uniform noise; float64 [n,3]."""
    rng = np.random.default_rng(seed)
    n_trunk = n // 2
    n_branch = (n - n_trunk) // 5
    parts = [_cylinder(rng, n_trunk, 0.30, 6.0, 0.005)]
    elev = np.deg2rad(37.0)
    for b in range(5):
        nb = n_branch if b < 4 else n - n_trunk - 4 * n_branch
        c = _cylinder(rng, nb, 0.08, 3.0, 0.003)
        az = 2.0 * np.pi * b / 5.0
        # tilt the +z cylinder to the branch direction, then rotate about z
        tilt = np.pi / 2.0 - elev
        ct, st = np.cos(tilt), np.sin(tilt)
        x = c[:, 0] * ct + c[:, 2] * st
        z = -c[:, 0] * st + c[:, 2] * ct
        y = c[:, 1]
        ca, sa = np.cos(az), np.sin(az)
        pts = np.stack([x * ca - y * sa, x * sa + y * ca, z + 2.0 + 0.7 * b], axis=1)
        parts.append(pts)
    pts = np.concatenate(parts, axis=0)
    n_noise = n // 100
    where = rng.choice(n, n_noise, replace=False)
    pts[where] = np.stack([rng.uniform(-4, 4, n_noise), rng.uniform(-4, 4, n_noise),
                           rng.uniform(0, 8, n_noise)], axis=1)
    return pts


def forest(n_points: int, seed: int = 0, pitch: float = 10.0) -> np.ndarray:
    """n_points / 50 000 tree units on a square grid; float64 [n,3], values
    exactly representable in float32."""
    units = max(1, int(round(n_points / TREE_UNIT)))
    per = n_points // units
    side = int(np.ceil(np.sqrt(units)))
    out = []
    for t in range(units):
        n = per if t < units - 1 else n_points - per * (units - 1)
        p = tree_unit(seed + t, n)
        p[:, 0] += pitch * (t % side)
        p[:, 1] += pitch * (t // side)
        out.append(p)
    pts = np.concatenate(out, axis=0)
    return pts.astype(np.float32).astype(np.float64)


def canopy_mesh(n_tris: int = 500_000, seed: int = 1, side: float = 0.05):
    """n_tris/2 randomly oriented square leaves (2 triangles each) in an oblate
    ellipsoid 8 x 8 x 5 m centred at z = 9 m. verts f32 [V,3], tris i32 [T,3]."""
    rng = np.random.default_rng(seed)
    q = n_tris // 2
    # centres uniform in the ellipsoid
    c = rng.normal(size=(q, 3))
    c /= np.linalg.norm(c, axis=1, keepdims=True)
    c *= rng.uniform(0, 1, (q, 1)) ** (1.0 / 3.0)
    c *= np.array([4.0, 4.0, 2.5])
    c[:, 2] += 9.0
    nrm = rng.normal(size=(q, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    helper = np.where(np.abs(nrm[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    a = np.cross(nrm, helper)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(nrm, a)
    h = side / 2.0
    corners = np.stack([c - h * a - h * b, c + h * a - h * b, c + h * a + h * b,
                        c - h * a + h * b], axis=1)  # [q,4,3]
    verts = corners.reshape(-1, 3).astype(np.float32)
    base = (4 * np.arange(q, dtype=np.int64))[:, None]
    tris = np.concatenate([base + np.array([[0, 1, 2]]), base + np.array([[0, 2, 3]])], axis=1)
    tris = tris.reshape(-1, 3).astype(np.int32)
    if n_tris % 2:
        tris = np.concatenate([tris, tris[:1]], axis=0)
    return verts, tris


def sun_rays(verts: np.ndarray, n_rays: int, elevation_deg: float = 60.0,
             azimuth_deg: float = 135.0, margin: float = 0.05) -> np.ndarray:
    """n_rays parallel rays on a regular grid above the mesh, pointing along the
    sun direction (downwards). f32 [R,6] = origin, direction (unit)."""
    el, az = np.deg2rad(elevation_deg), np.deg2rad(azimuth_deg)
    # direction the light travels: from the sun towards the ground
    d = -np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
    # orthonormal frame (u, v) spanning the plane normal to d
    u = np.cross(d, np.array([0.0, 0.0, 1.0]))
    u /= np.linalg.norm(u)
    v = np.cross(d, u)
    P = verts.astype(np.float64)
    pu, pv, pd = P @ u, P @ v, P @ d
    mu, Mu, mv, Mv = pu.min(), pu.max(), pv.min(), pv.max()
    du, dv = (Mu - mu) * margin, (Mv - mv) * margin
    nx = int(np.floor(np.sqrt(n_rays)))
    ny = int(np.ceil(n_rays / nx))
    gu = np.linspace(mu - du, Mu + du, nx)
    gv = np.linspace(mv - dv, Mv + dv, ny)
    uu, vv = np.meshgrid(gu, gv, indexing="xy")
    uu, vv = uu.reshape(-1)[:n_rays], vv.reshape(-1)[:n_rays]
    start = pd.min() - 1.0  # one metre before the first vertex along d
    orig = uu[:, None] * u + vv[:, None] * v + start * d
    rays = np.empty((n_rays, 6), dtype=np.float32)
    rays[:, :3] = orig
    rays[:, 3:] = d
    return rays


def ring_cluster(n: int, radius: float = 0.3, seed: int = 0, noise: float = 0.004,
                 outliers: float = 0.15) -> np.ndarray:
    """A stem cross-section for the RANSAC path: points near a circle of the given
    radius in the z = const plane spread over 0.5 m of height, plus outliers."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, 2 * np.pi, n)
    r = radius + rng.normal(0, noise, n)
    pts = np.stack([1.5 + r * np.cos(ang), -0.7 + r * np.sin(ang), rng.uniform(2.0, 2.5, n)], 1)
    k = int(n * outliers)
    where = rng.choice(n, k, replace=False)
    pts[where, :2] += rng.uniform(-0.5, 0.5, (k, 2))
    return pts.astype(np.float32).astype(np.float64)
