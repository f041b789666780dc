"""CPU oracle for the pyQSM hot path — TEST INFRASTRUCTURE ONLY.

Nothing in ``pyqsm_amd`` imports this package. It is used by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` to check
(and time) the HIP path against an independent CPU statement of the algorithms
the reference calls. Each function cites the reference lines it follows.

Parity status (details in DESIGN.md):

* DBSCAN, kNN, the contraction solve: PINNED — tests/golden holds outputs of
  scikit-learn / SciPy themselves (the engines the reference calls), produced by
  tests/golden/make_golden.py in this image.
* RANSAC (pyransac3d), ray casting (Open3D/Embree), point-cloud Laplacian
  (robust_laplacian): those packages are absent and cannot be installed:
  PARITY UNPINNED. The restatements follow the published algorithms and are
  pinned only by analytic known answers.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (no-op when it is up to date)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pyqsm_oracle.c", "ray_f64.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(map(os.path.getmtime, srcs)):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build())
        i64, i32, dbl = ctypes.c_int64, ctypes.c_int32, ctypes.c_double
        p = ctypes.c_void_p
        lib.orc_dbscan.restype = i64
        lib.orc_dbscan.argtypes = [p, i64, dbl, i32, p, p]
        lib.orc_dbscan_strict.restype = i64
        lib.orc_dbscan_strict.argtypes = [p, i64, dbl, i32, p, p]
        lib.orc_knn.restype = ctypes.c_int
        lib.orc_knn.argtypes = [p, i64, i32, i32, p, p]
        lib.orc_cast_rays.restype = ctypes.c_int
        lib.orc_cast_rays.argtypes = [p, i64, p, i64, p, i64, p, p, p]
        lib.orc_list_intersections.restype = i64
        lib.orc_list_intersections.argtypes = [p, i64, p, i64, p, i64, p, p, p, p, p, i64]
        lib.orc_num_threads.restype = ctypes.c_int
        lib.orc_pc_laplacian.restype = ctypes.c_int
        lib.orc_pc_laplacian.argtypes = [p, i64, i32, dbl, ctypes.POINTER(i64),
                                         ctypes.POINTER(p), ctypes.POINTER(p), ctypes.POINTER(p), p]
        lib.orc_free.restype = None
        lib.orc_point_mesh_distance.argtypes = [p, i64, p, i64, p, i64, p, p]
        lib.orc_free.argtypes = [p]
        lib.orc_cast_rays_f64.restype = ctypes.c_int
        lib.orc_cast_rays_f64.argtypes = [p, p, i64, p, i64, p, p, p, p]
        lib.orc_ray_tri_pairs_f64.restype = ctypes.c_int
        lib.orc_ray_tri_pairs_f64.argtypes = [p, p, p, i64, p, p, p, p]
        _LIB = lib
    return _LIB


def num_threads() -> int:
    return int(_lib().orc_num_threads())


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# --------------------------------------------------------------------------
# DBSCAN  (sklearn.cluster.DBSCAN as called at pyQSM/math_utils/fit.py:223)

def dbscan(points, eps, min_pts, radius_inclusive=True):
    """labels int64 [n] (-1 = noise), core mask bool [n]. ``radius_inclusive=False``: the strict
    neighbourhood d2 < eps^2 (Open3D's compare if nanoflann's is strict; parity unpinned)."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    labels = np.empty(n, dtype=np.int64)
    core = np.zeros(n, dtype=np.uint8)
    fn = _lib().orc_dbscan if radius_inclusive else _lib().orc_dbscan_strict
    rc = fn(_ptr(pts), n, float(eps), int(min_pts), _ptr(labels), _ptr(core))
    if rc < 0:
        raise MemoryError("orc_dbscan")
    return labels, core.astype(bool)


def cluster_DBSCAN(pts_idxs, points, eps, min_pts):
    """pyQSM/math_utils/fit.py:217-250 on top of :func:`dbscan`.

    Returns (unique_labels: set, idxs: list of index arrays holding the CORE
    samples of each non-noise label in set-iteration order, noise: indices with
    label -1 that are not core)."""
    labels, core = dbscan(points, eps, min_pts)
    pts_idxs = np.asarray(pts_idxs)
    unique_labels = set(labels)          # fit.py:231
    idxs, noise = [], []
    for k in unique_labels:              # fit.py:237
        member = labels == k
        if k == -1:
            noise = pts_idxs[np.where(member & ~core)]      # fit.py:239-241
        else:
            idxs.append(pts_idxs[np.where(member & core)])  # fit.py:243-246
    return unique_labels, idxs, noise


# --------------------------------------------------------------------------
# kNN  (scipy cKDTree.query, pyQSM/geometry/reconstruction.py:238-240)

def knn(points, k, exclude_self=True):
    """idx int32 [n,k], squared distances float64 [n,k], ascending."""
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    idx = np.empty((n, k), dtype=np.int32)
    d2 = np.empty((n, k), dtype=np.float64)
    rc = _lib().orc_knn(_ptr(pts), n, int(k), int(bool(exclude_self)), _ptr(idx), _ptr(d2))
    if rc != 0:
        raise MemoryError("orc_knn")
    return idx, d2


# --------------------------------------------------------------------------
# Ray casting  (Open3D RaycastingScene, pyQSM/viz/ray_casting.py:275-279,168)

def cast_rays(verts, tris, rays):
    """t_hit f32 [R] (+inf miss), prim u32 [R] (0xFFFFFFFF miss), uv f32 [R,2]."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    R = r.shape[0]
    t_hit = np.empty(R, dtype=np.float32)
    prim = np.empty(R, dtype=np.uint32)
    uv = np.empty((R, 2), dtype=np.float32)
    rc = _lib().orc_cast_rays(_ptr(v), v.shape[0], _ptr(t), t.shape[0], _ptr(r), R,
                              _ptr(t_hit), _ptr(prim), _ptr(uv))
    if rc != 0:
        raise MemoryError("orc_cast_rays")
    return t_hit, prim, uv



def cast_rays_f64(verts, tris, rays):
    """INDEPENDENT closest hit (oracle/ray_f64.c: double precision, signed-volume formulation —
    not the operation order of the HIP kernels or of :func:`cast_rays`).
    Returns t f64 [R] (+inf miss), prim i64 [R] (-1 miss), bary f64 [R,3] (weights of v0, v1,
    v2; u = bary[:,1], v = bary[:,2] in ray_casting.py:172-180's convention) and the depth of
    the runner-up hit f64 [R] (+inf if none)."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    R = r.shape[0]
    t_hit, second = np.empty(R), np.empty(R)
    prim = np.empty(R, dtype=np.int64)
    bary = np.empty((R, 3))
    _lib().orc_cast_rays_f64(_ptr(v), _ptr(t), t.shape[0], _ptr(r), R, _ptr(t_hit), _ptr(prim),
                             _ptr(bary), _ptr(second))
    return t_hit, prim, bary, second


def ray_tri_pairs_f64(verts, tris, rays, prim):
    """What double precision says about GIVEN (ray, triangle) pairs: (pierces bool [R], t f64 [R],
    bary f64 [R,3]); rows with prim < 0 are skipped (False / NaN)."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    pr = np.ascontiguousarray(prim, dtype=np.int64).reshape(-1)
    R = r.shape[0]
    pierces = np.zeros(R, dtype=np.uint8)
    tt = np.empty(R)
    bary = np.empty((R, 3))
    _lib().orc_ray_tri_pairs_f64(_ptr(v), _ptr(t), _ptr(r), R, _ptr(pr), _ptr(pierces), _ptr(tt),
                                 _ptr(bary))
    return pierces.astype(bool), tt, bary


def interception_layers(verts, tris, rays, max_rounds=None):
    """data/notes/methods.md:53-55 ("metrics with overlap"): cast, sum the area of the
    triangles hit first, remove them, repeat until nothing is hit. The reference keeps no
    code for it (parity with the reference unpinned); this restates the procedure with the
    oracle's own closest-hit sweep. Returns (areas per round, round of every triangle)."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    e1 = v[t[:, 1]].astype(np.float64) - v[t[:, 0]].astype(np.float64)
    e2 = v[t[:, 2]].astype(np.float64) - v[t[:, 0]].astype(np.float64)
    tri_area = 0.5 * np.linalg.norm(np.cross(e1, e2), axis=1)
    layer = np.full(len(t), -1, dtype=np.int64)
    areas = []
    alive = np.arange(len(t))
    rnd = 0
    while len(alive) and len(np.asarray(rays).reshape(-1, 6)) and (max_rounds is None or rnd < max_rounds):
        _, prim, _ = cast_rays(v, t[alive], rays)
        hit = np.unique(prim[prim != np.uint32(0xFFFFFFFF)])
        if len(hit) == 0:
            break
        ids = alive[hit]
        layer[ids] = rnd
        areas.append(float(tri_area[ids].sum()))
        alive = np.delete(alive, hit)
        rnd += 1
    return areas, layer


def point_mesh_distance(verts, tris, queries):
    """(dist f32 [Q], prim u32 [Q]): unsigned distance to the mesh, closest triangle."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 3)
    dist = np.empty(q.shape[0], dtype=np.float32)
    prim = np.empty(q.shape[0], dtype=np.uint32)
    _lib().orc_point_mesh_distance(_ptr(v), v.shape[0], _ptr(t), t.shape[0], _ptr(q), q.shape[0],
                                   _ptr(dist), _ptr(prim))
    return dist, prim


def list_intersections(verts, tris, rays):
    """dict with ray_ids, primitive_ids, t_hit, primitive_uvs (+ per-ray counts)."""
    v = np.ascontiguousarray(verts, dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 6)
    R = r.shape[0]
    counts = np.zeros(R, dtype=np.int32)
    total = _lib().orc_list_intersections(_ptr(v), v.shape[0], _ptr(t), t.shape[0], _ptr(r), R,
                                          _ptr(counts), None, None, None, None, 0)
    ray_ids = np.empty(total, dtype=np.uint32)
    prim = np.empty(total, dtype=np.uint32)
    ts = np.empty(total, dtype=np.float32)
    uv = np.empty((total, 2), dtype=np.float32)
    if total:
        _lib().orc_list_intersections(_ptr(v), v.shape[0], _ptr(t), t.shape[0], _ptr(r), R,
                                      _ptr(counts), _ptr(ray_ids), _ptr(prim), _ptr(ts), _ptr(uv),
                                      total)
    return {"ray_ids": ray_ids, "primitive_ids": prim, "t_hit": ts, "primitive_uvs": uv,
            "counts": counts}


# --------------------------------------------------------------------------
# RANSAC  (pyransac3d Circle.fit / Cylinder.fit, pyQSM/math_utils/fit.py:277-283)
#
# pyransac3d is not declared by the reference (imported at fit.py:13) and is not
# installed here; what follows restates its published algorithm (circle.py,
# cylinder.py, aux_functions.rodrigues_rot of pyransac3d 0.6.0) with the random
# 3-point samples passed in. PARITY UNPINNED against the package itself.

def _rodrigues_rot(P, n0, n1):
    P = np.asarray(P, dtype=np.float64)
    if P.ndim == 1:
        P = P[np.newaxis, :]
    n0 = n0 / np.linalg.norm(n0)
    n1 = n1 / np.linalg.norm(n1)
    k = np.cross(n0, n1)
    P_rot = np.zeros((len(P), 3))
    if np.linalg.norm(k) != 0:
        k = k / np.linalg.norm(k)
        theta = np.arccos(np.dot(n0, n1))
        for i in range(len(P)):
            P_rot[i] = (P[i] * np.cos(theta) + np.cross(k, P[i]) * np.sin(theta)
                        + k * np.dot(k, P[i]) * (1 - np.cos(theta)))
    else:
        P_rot = P
    return P_rot


def ransac_model(pt_samples):
    """(center[3], axis[3], radius) of the circle through three points; None when
    the construction degenerates (collinear / coincident samples)."""
    with np.errstate(all="ignore"):
        vecA = pt_samples[1, :] - pt_samples[0, :]
        vecA_norm = vecA / np.linalg.norm(vecA)
        vecB = pt_samples[2, :] - pt_samples[0, :]
        vecB_norm = vecB / np.linalg.norm(vecB)
        vecC = np.cross(vecA_norm, vecB_norm)
        vecC = vecC / np.linalg.norm(vecC)
        if not np.all(np.isfinite(vecC)):
            return None
        P_rot = _rodrigues_rot(pt_samples, vecC, np.array([0.0, 0.0, 1.0]))
        ma = mb = 0.0
        for _ in range(3):
            ma = (P_rot[1, 1] - P_rot[0, 1]) / (P_rot[1, 0] - P_rot[0, 0])
            mb = (P_rot[2, 1] - P_rot[1, 1]) / (P_rot[2, 0] - P_rot[1, 0])
            if ma == 0:
                P_rot = np.roll(P_rot, -1, axis=0)
            else:
                break
        cx = (ma * mb * (P_rot[0, 1] - P_rot[2, 1]) + mb * (P_rot[0, 0] + P_rot[1, 0])
              - ma * (P_rot[1, 0] + P_rot[2, 0])) / (2 * (mb - ma))
        cy = -1 / ma * (cx - (P_rot[0, 0] + P_rot[1, 0]) / 2) + (P_rot[0, 1] + P_rot[1, 1]) / 2
        p_center = np.array([cx, cy, 0.0])
        radius = np.linalg.norm(p_center - P_rot[0, :])
        center = _rodrigues_rot(p_center, np.array([0.0, 0.0, 1.0]), vecC)[0]
        if not (np.all(np.isfinite(center)) and np.isfinite(radius)):
            return None
    return center, vecC, radius


def ransac_distance(pts, center, axis, radius, shape):
    """Point-to-model distance compared against the threshold (|dist| <= thresh)."""
    n = pts.shape[0]
    stack = np.stack([axis] * n, 0)
    if shape == "circle":
        dist_pt_plane = (axis[0] * (pts[:, 0] - center[0]) + axis[1] * (pts[:, 1] - center[1])
                         + axis[2] * (pts[:, 2] - center[2]))
        d_inf = np.cross(stack, (center - pts))
        d_inf = np.linalg.norm(d_inf, axis=1) - radius
        return np.abs(np.sqrt(np.square(d_inf) + np.square(dist_pt_plane)))
    d = np.cross(stack, (center - pts))
    d = np.linalg.norm(d, axis=1)
    return np.abs(d - radius)


def ransac_fit(pts, triples, shape="circle", thresh=0.2):
    """center, axis, radius, inliers (ascending int64), index of the winning triple.

    First hypothesis with a strictly larger inlier count wins (circle.py /
    cylinder.py: ``if len(pt_id_inliers) > len(best_inliers)``)."""
    pts = np.asarray(pts, dtype=np.float64)
    best_inliers = np.zeros(0, dtype=np.int64)
    best = (np.zeros(0), np.zeros(0), 0.0)
    best_row = -1
    for h, ids in enumerate(np.asarray(triples, dtype=np.int64)):
        m = ransac_model(pts[ids])
        if m is None:
            continue
        center, axis, radius = m
        dist = ransac_distance(pts, center, axis, radius, shape)
        inl = np.where(dist <= thresh)[0]
        if len(inl) > len(best_inliers):
            best_inliers = inl.astype(np.int64)
            best = (center, axis, radius)
            best_row = h
    return best[0], best[1], best[2], best_inliers, best_row


# --------------------------------------------------------------------------
# Laplacian contraction  (pyQSM/geometry/skeletonize.py:148-180, 226-373)

def least_squares_sparse(pts, L, laplacian_weighting, positional_weighting):
    """skeletonize.py:148-180, statement for statement (SciPy is what it calls)."""
    from scipy.sparse import diags, vstack
    from scipy.sparse import linalg as sla
    WL = diags(laplacian_weighting)
    WH = diags(positional_weighting)
    A = vstack([L.dot(WL), WH]).tocsc()
    b = np.vstack([np.zeros((pts.shape[0], 3)), WH.dot(pts)])
    A_new = A.T @ A
    x = sla.spsolve(A_new, A.T @ b[:, 0], permc_spec="COLAMD")
    y = sla.spsolve(A_new, A.T @ b[:, 1], permc_spec="COLAMD")
    z = sla.spsolve(A_new, A.T @ b[:, 2], permc_spec="COLAMD")
    ret = np.vstack([x, y, z]).T
    if (np.isnan(ret)).all():
        ret = pts
    return ret


def extract_skeleton(pts, laplacian, allowed_range, max_iter=20, termination_ratio=0.003,
                     contraction_factor=3, attraction_factor=3, max_contraction=2048,
                     max_attraction=1024, solve=least_squares_sparse):
    """The loop of skeletonize.py:240-373 with its bookkeeping quirks kept
    (M_list = [M0, M0, M1, ...], volume ratio lagging one iteration, wh updated
    with the mass of the Laplacian just used). ``laplacian(pts) -> (L, mass)``
    and the OBB range are passed in. Returns (contracted, total_shift, steps)."""
    pts = np.array(pts, dtype=np.float64)
    L, M = laplacian(pts)
    M_list = [M]
    wh = attraction_factor * np.ones(M.shape[0])                                   # :264
    wl = contraction_factor * 10 ** 3 * np.sqrt(np.mean(M)) * np.ones(M.shape[0])  # :265
    iteration = 0
    volume_ratio = 1
    cur = pts
    steps = []
    total = np.zeros_like(cur)
    lo, hi = np.asarray(allowed_range[0]), np.asarray(allowed_range[1])
    while volume_ratio > termination_ratio:                                        # :279
        new = solve(cur, L, wl, wh)
        if (new == cur).all():                                                     # :287
            break
        new = np.minimum(np.maximum(new, lo), hi)                                  # :291-296
        shift = cur - new
        total += shift
        cur = new
        steps.append(shift)
        wl = wl * contraction_factor                                               # :329
        wh = wh * np.sqrt(M_list[0] / M)                                           # :331
        wl = np.clip(wl, 0.1, max_contraction)                                     # :334
        wh = np.clip(wh, 0.1, max_attraction)                                      # :335
        M_list.append(M)                                                           # :337
        iteration += 1
        L, M = laplacian(cur)                                                      # :341
        volume_ratio = np.mean(M_list[-1]) / np.mean(M_list[0])                    # :349
        if iteration >= max_iter:                                                  # :353
            break
    return cur, total, steps


# --------------------------------------------------------------------------
# Point-cloud Laplacian (robust_laplacian.point_cloud_laplacian,
# pyQSM/geometry/skeletonize.py:253-255) — own restatement, PARITY UNPINNED.

def point_cloud_laplacian(points, n_neighbors=30, mollify_factor=1e-5):
    """(L scipy CSR [n,n], mass float64 [n]); see pyqsm_oracle.c: orc_pc_laplacian."""
    from scipy.sparse import csr_matrix
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    n = pts.shape[0]
    lib = _lib()
    nnz = ctypes.c_int64(0)
    ip, ix, dv = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
    mass = np.zeros(n, dtype=np.float64)
    rc = lib.orc_pc_laplacian(_ptr(pts), n, int(n_neighbors), float(mollify_factor),
                              ctypes.byref(nnz), ctypes.byref(ip), ctypes.byref(ix),
                              ctypes.byref(dv), _ptr(mass))
    if rc != 0:
        raise RuntimeError("orc_pc_laplacian failed (k must be in [3, 64])")
    try:
        indptr = np.ctypeslib.as_array(ctypes.cast(ip, ctypes.POINTER(ctypes.c_int32)),
                                       (n + 1,)).copy()
        indices = np.ctypeslib.as_array(ctypes.cast(ix, ctypes.POINTER(ctypes.c_int32)),
                                        (nnz.value + 1,))[:nnz.value].copy()
        data = np.ctypeslib.as_array(ctypes.cast(dv, ctypes.POINTER(ctypes.c_double)),
                                     (nnz.value + 1,))[:nnz.value].copy()
    finally:
        for q in (ip, ix, dv):
            lib.orc_free(q)
    return csr_matrix((data, indices, indptr), shape=(n, n)), mass


# --------------------------------------------------------------------------
# Farthest-point sampling (open3d PointCloud.farthest_point_down_sample,
# pyQSM/geometry/skeletonize.py:132). Open3D is absent: restated from its
# documented algorithm (start at index 0, keep the running minimum of squared
# distances, pick the arg-max, first index on ties). PARITY UNPINNED.

def farthest_point_sampling(points, num_samples, start_index=0):
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    dist = np.full(len(pts), np.inf)
    out = np.empty(num_samples, dtype=np.int32)
    sel = int(start_index)
    for s in range(num_samples):
        out[s] = sel
        t = pts - pts[sel]
        d = (t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1]) + t[:, 2] * t[:, 2]
        dist = np.minimum(dist, d)
        sel = int(np.argmax(dist))          # first maximum
    return out


# --------------------------------------------------------------------------
# Region growing (pyQSM/tree_isolation.py:63-283)

def extend_seed_clusters(seeds, src_pts, k=200, max_distance=.1, cycles=150, exclude_pts=None):
    """The loop of tree_isolation.py:98-262 on SciPy's KD-tree, cluster by cluster and
    with the ownership dict keyed by coordinate tuples, as the reference has it
    (``order_cutoff=None``; no drawing / pickling). ``seeds`` = [(label, pts [m,3])].
    Returns {label: float64 [p,3]} — seed points followed by the points acquired, in
    acquisition order."""
    from collections import defaultdict
    import itertools
    from scipy.spatial import cKDTree
    src_pts = np.asarray(src_pts, dtype=np.float64)
    assn = defaultdict(lambda: -1)
    curr = []
    for idc, (label, pts) in enumerate(seeds):
        pts = np.asarray(pts, dtype=np.float64)
        curr.append(pts)
        for pt in pts:
            assn[tuple(pt)] = idc                                          # :103-105
    if exclude_pts is not None and len(exclude_pts):                       # :120-133
        tree = cKDTree(src_pts)
        _, nbrs = tree.query(np.asarray(exclude_pts, dtype=np.float64), k=k,
                             distance_upper_bound=max_distance)
        drop = {x for x in itertools.chain.from_iterable(np.atleast_2d(nbrs)) if x != len(src_pts)}
        keep = np.ones(len(src_pts), dtype=bool)
        keep[list(drop)] = False
        src_pts = src_pts[keep]
    tree = cKDTree(src_pts)
    num_pts = len(src_pts)
    complete = []
    for cycle_num in range(cycles):
        for idx in range(len(seeds)):
            if idx in complete:
                continue
            if len(curr[idx]) > 0:
                _, nbrs = tree.query(curr[idx], k=k, distance_upper_bound=max_distance)   # :207-209
                nbrs = sorted({int(x) for x in itertools.chain.from_iterable(np.atleast_2d(nbrs))
                               if x != num_pts})
                nbr_pts = [src_pts[nb] for nb in nbrs if assn[tuple(src_pts[nb])] == -1]
                if len(nbr_pts) > 0:
                    for pt in nbr_pts:
                        assn[tuple(pt)] = idx
                    curr[idx] = np.asarray(nbr_pts)
                    if len(curr[idx]) < 5:                                 # :256-258
                        complete.append(idx)
            if len(curr[idx]) == 0:
                complete.append(idx)
    out = defaultdict(list)
    for pt, idc in assn.items():
        if idc >= 0:
            out[seeds[idc][0]].append(pt)
    return {label: np.asarray(pts, dtype=np.float64).reshape(-1, 3) for label, pts in out.items()}
