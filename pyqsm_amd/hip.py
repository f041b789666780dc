"""NumPy-facing calls into libpyqsm_hip.so.

One thin function per C-ABI entry point: validates shapes/dtypes, hands plain
pointers to the library, raises :class:`pyqsm_amd._lib.PyQSMHipError` on failure.
No computation happens in Python here.
"""
from __future__ import annotations

import ctypes
import weakref

import numpy as np

from . import _lib
from ._lib import check, i64, i32, dbl, vp

MISS_PRIM = np.uint32(0xFFFFFFFF)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _points(points) -> np.ndarray:
    pts = np.ascontiguousarray(np.asarray(points), dtype=np.float64)
    if pts.ndim != 2 or pts.shape[1] != 3:
        raise ValueError(f"expected points of shape [n,3], got {pts.shape}")
    return pts


# ---------------------------------------------------------------- device buffers

class DeviceBuffer:
    """A block of HBM owned by the caller, for the *_dev entry points."""

    def __init__(self, nbytes: int, device: int = 0):
        self.device = int(device)
        self.nbytes = int(nbytes)
        ptr = vp()
        check(_lib.load().pyqsm_dev_malloc(self.device, self.nbytes, ctypes.byref(ptr)))
        self.ptr = ptr.value

    @classmethod
    def from_array(cls, a: np.ndarray, device: int = 0) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes, device)
        buf.upload(a)
        return buf

    def upload(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        if a.nbytes > self.nbytes:
            raise ValueError("array larger than device buffer")
        check(_lib.load().pyqsm_h2d(self.device, self.ptr, _p(a), a.nbytes))

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("requested more bytes than the device buffer holds")
        check(_lib.load().pyqsm_d2h(self.device, _p(out), self.ptr, out.nbytes))
        return out

    def free(self) -> None:
        if self.ptr:
            _lib.load().pyqsm_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync(device: int = 0) -> None:
    check(_lib.load().pyqsm_sync(int(device)))


def prof_enable(on, device: int = 0) -> None:
    """0 / False off; 1 / True phase timers; 2 also single solver kernels and counters."""
    check(_lib.load().pyqsm_prof_enable(int(device), int(on)))


def prof_reset(device: int = 0) -> None:
    check(_lib.load().pyqsm_prof_reset(int(device)))


def prof_get(name: str, device: int = 0):
    """(total milliseconds, launches) recorded under `name` since the last reset."""
    ms, cnt = dbl(0.0), i64(0)
    check(_lib.load().pyqsm_prof_get(int(device), name.encode(), ctypes.byref(ms),
                                     ctypes.byref(cnt)))
    return ms.value, cnt.value


# ---------------------------------------------------------------- ray casting

def _mesh(verts, tris):
    v = np.ascontiguousarray(np.asarray(verts), dtype=np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(np.asarray(tris), dtype=np.int32).reshape(-1, 3)
    return v, t


def cast_rays(verts, tris, rays, device: int = 0):
    """Closest hit. Returns t_hit f32 (+inf = miss), primitive_ids u32
    (0xFFFFFFFF = miss), primitive_uvs f32 [...,2]; leading shape follows `rays`."""
    v, t = _mesh(verts, tris)
    r = np.ascontiguousarray(np.asarray(rays), dtype=np.float32)
    if r.shape[-1] != 6:
        raise ValueError(f"rays must have last dimension 6, got {r.shape}")
    lead = r.shape[:-1]
    r2 = r.reshape(-1, 6)
    R = r2.shape[0]
    t_hit = np.empty(R, dtype=np.float32)
    prim = np.empty(R, dtype=np.uint32)
    uv = np.empty((R, 2), dtype=np.float32)
    check(_lib.load().pyqsm_cast_rays(_p(v), v.shape[0], _p(t), t.shape[0], _p(r2), R,
                                      _p(t_hit), _p(prim), _p(uv), int(device)))
    return t_hit.reshape(lead), prim.reshape(lead), uv.reshape(lead + (2,))


def cast_rays_multi(verts, tris, rays, n_devices: int = 0, with_uv: bool = True):
    """The same closest-hit sweep on ``n_devices`` GPUs driven from this one process
    (``pyqsm_cast_rays_multi``: mesh replicated by RCCL broadcast, rays in contiguous shards,
    results all-gathered; 0 = every visible GPU). No torch involved. Returns what
    :func:`cast_rays` returns, bit for bit (``with_uv=False``: uv is None and half the bytes
    are gathered)."""
    v, t = _mesh(verts, tris)
    r = np.ascontiguousarray(np.asarray(rays), dtype=np.float32)
    if r.shape[-1] != 6:
        raise ValueError(f"rays must have last dimension 6, got {r.shape}")
    lead = r.shape[:-1]
    r2 = r.reshape(-1, 6)
    R = r2.shape[0]
    t_hit = np.empty(R, dtype=np.float32)
    prim = np.empty(R, dtype=np.uint32)
    uv = np.empty((R, 2), dtype=np.float32) if with_uv else None
    check(_lib.load().pyqsm_cast_rays_multi(_p(v), v.shape[0], _p(t), t.shape[0], _p(r2), R,
                                            _p(t_hit), _p(prim), _p(uv) if with_uv else None,
                                            int(n_devices)))
    return t_hit.reshape(lead), prim.reshape(lead), uv.reshape(lead + (2,)) if with_uv else None


def list_intersections(verts, tris, rays, device: int = 0):
    """Every crossing with t > 0, ordered by ray id then triangle id."""
    v, t = _mesh(verts, tris)
    r = np.ascontiguousarray(np.asarray(rays), dtype=np.float32).reshape(-1, 6)
    R = r.shape[0]
    lib = _lib.load()
    counts = np.zeros(R, dtype=np.int32)
    total = i64(0)
    check(lib.pyqsm_list_intersections(_p(v), v.shape[0], _p(t), t.shape[0], _p(r), R, _p(counts),
                                       None, None, None, None, 0, ctypes.byref(total),
                                       int(device)))
    n = int(total.value)
    ray_ids = np.empty(n, dtype=np.uint32)
    prim = np.empty(n, dtype=np.uint32)
    ts = np.empty(n, dtype=np.float32)
    uv = np.empty((n, 2), dtype=np.float32)
    if n:
        check(lib.pyqsm_list_intersections(_p(v), v.shape[0], _p(t), t.shape[0], _p(r), R,
                                           _p(counts), _p(ray_ids), _p(prim), _p(ts), _p(uv), n,
                                           ctypes.byref(total), int(device)))
    return {"ray_ids": ray_ids, "primitive_ids": prim, "t_hit": ts, "primitive_uvs": uv,
            "counts": counts}


def point_mesh_distance(verts, tris, queries, device: int = 0):
    """(dist f32 [Q], prim u32 [Q]): unsigned distance to the mesh and the closest triangle."""
    v, t = _mesh(verts, tris)
    q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 3)
    dist = np.empty(q.shape[0], dtype=np.float32)
    prim = np.empty(q.shape[0], dtype=np.uint32)
    check(_lib.load().pyqsm_point_mesh_distance(_p(v), v.shape[0], _p(t), t.shape[0], _p(q), q.shape[0],
                                                _p(dist), _p(prim), int(device)))
    return dist, prim


class DeviceMesh:
    """A mesh expanded into the sweep's 48-byte records, resident in HBM."""

    def __init__(self, verts, tris, device: int = 0):
        v, t = _mesh(verts, tris)
        self.device = int(device)
        self.n_tris = t.shape[0]
        dv = DeviceBuffer.from_array(v, device) if v.size else None
        dt = DeviceBuffer.from_array(t, device) if t.size else None
        self.records = DeviceBuffer(max(1, self.n_tris) * 48, device)
        if self.n_tris:
            check(_lib.load().pyqsm_expand_tris_dev(dv.ptr, v.shape[0], dt.ptr, self.n_tris,
                                                    self.records.ptr, self.device))
            sync(device)
        for b in (dv, dt):
            if b is not None:
                b.free()


def cast_rays_dev(mesh: DeviceMesh, rays_ptr: int, n_rays: int, t_hit_ptr: int, prim_ptr: int,
                  uv_ptr: int | None = None) -> None:
    """Asynchronous sweep over HBM-resident rays (pointers are device addresses)."""
    check(_lib.load().pyqsm_cast_rays_dev(mesh.records.ptr, mesh.n_tris, rays_ptr, int(n_rays),
                                          t_hit_ptr, prim_ptr, uv_ptr, mesh.device))


# ---------------------------------------------------------------- DBSCAN / kNN

def dbscan(points, eps: float, min_pts: int, device: int = 0, radius_inclusive: bool = True):
    """labels int64 [n] (-1 = noise), core mask bool [n]. ``radius_inclusive=False``: the strict
    neighbourhood d2 < eps^2 (``pyqsm_dbscan_ex``) instead of scikit-learn's d2 <= eps^2."""
    pts = _points(points)
    n = pts.shape[0]
    labels = np.empty(n, dtype=np.int64)
    core = np.zeros(n, dtype=np.uint8)
    check(_lib.load().pyqsm_dbscan_ex(_p(pts), n, float(eps), int(min_pts), int(bool(radius_inclusive)),
                                      _p(labels), _p(core), int(device)))
    return labels, core.astype(bool)


def dbscan_dev(xyz_ptr: int, n: int, eps: float, min_pts: int, labels_ptr: int,
               core_ptr: int | None = None, device: int = 0, want_count: bool = False,
               radius_inclusive: bool = True):
    """Asynchronous clustering of an HBM-resident cloud; returns the cluster count
    when `want_count` (that read-back synchronises)."""
    cnt = i64(0)
    check(_lib.load().pyqsm_dbscan_dev_ex(xyz_ptr, int(n), float(eps), int(min_pts),
                                          int(bool(radius_inclusive)), labels_ptr, core_ptr,
                                          ctypes.byref(cnt) if want_count else None, int(device)))
    return int(cnt.value) if want_count else None


def knn(points, k: int, exclude_self: bool = True, device: int = 0):
    """idx int32 [n,k], squared distances float64 [n,k], ascending by (d2, index)."""
    pts = _points(points)
    n = pts.shape[0]
    idx = np.empty((n, int(k)), dtype=np.int32)
    d2 = np.empty((n, int(k)), dtype=np.float64)
    check(_lib.load().pyqsm_knn(_p(pts), n, int(k), int(bool(exclude_self)), _p(idx), _p(d2),
                                int(device)))
    return idx, d2


def knn_dev(xyz_ptr: int, n: int, k: int, exclude_self: bool, idx_ptr: int, d2_ptr: int,
            device: int = 0) -> None:
    check(_lib.load().pyqsm_knn_dev(xyz_ptr, int(n), int(k), int(bool(exclude_self)), idx_ptr,
                                    d2_ptr, int(device)))


# ---------------------------------------------------------------- RANSAC

SHAPES = {"circle": 0, "cylinder": 1}


def ransac(points, triples, shape: str = "circle", thresh: float = 0.2, device: int = 0):
    """center[3], axis[3], radius, inliers int64 (ascending), winning row (-1: none)."""
    pts = _points(points)
    tri = np.ascontiguousarray(np.asarray(triples), dtype=np.int64).reshape(-1, 3)
    n, H = pts.shape[0], tri.shape[0]
    if H and (tri.min() < 0 or tri.max() >= n):
        raise ValueError("sample index outside the point set")
    center = np.zeros(3)
    axis = np.zeros(3)
    radius = dbl(0.0)
    inl = np.empty(max(n, 1), dtype=np.int64)
    n_in, best = i64(0), i64(-1)
    check(_lib.load().pyqsm_ransac(_p(pts), n, _p(tri), H, SHAPES[shape], float(thresh),
                                   _p(center), _p(axis), ctypes.byref(radius), _p(inl),
                                   ctypes.byref(n_in), ctypes.byref(best), int(device)))
    return center, axis, radius.value, inl[:n_in.value].copy(), int(best.value)


def ransac_batch(points, seg_start, triples, shape: str = "circle", thresh: float = 0.2, device: int = 0):
    """``pyqsm_ransac_batch``: S point sets stacked in ``points`` (``seg_start`` int64 [S+1]), H
    hypotheses each (``triples`` int64 [S,H,3], indices local to the set). Returns ``(centers [S,3],
    axes [S,3], radii [S], inliers list of S int64 arrays (ascending, local), best int64 [S])``."""
    pts = _points(points)
    ss = np.ascontiguousarray(seg_start, dtype=np.int64)
    S = len(ss) - 1
    tri = np.ascontiguousarray(np.asarray(triples), dtype=np.int64).reshape(S, -1, 3)
    H = tri.shape[1]
    sizes = np.diff(ss)
    if H and S:
        ok = (tri >= 0).all(axis=(1, 2)) | (sizes < 3)
        if not ok.all() or (tri.max(axis=(1, 2)) >= np.maximum(sizes, 1))[sizes >= 3].any():
            raise ValueError("sample index outside its point set")
    centers = np.zeros((S, 3))
    axes = np.zeros((S, 3))
    radii = np.zeros(S)
    inl = np.empty(max(pts.shape[0], 1), dtype=np.int64)
    n_in = np.zeros(S, dtype=np.int64)
    best = np.full(S, -1, dtype=np.int64)
    check(_lib.load().pyqsm_ransac_batch(_p(pts), pts.shape[0], _p(ss), S, _p(tri), H, SHAPES[shape],
                                         float(thresh), _p(centers), _p(axes), _p(radii), _p(inl), _p(n_in),
                                         _p(best), int(device)))
    ends = np.cumsum(n_in)
    inliers = [inl[e - c:e].copy() for c, e in zip(n_in, ends)]
    return centers, axes, radii, inliers, best


def ransac_models(points, triples, device: int = 0):
    """f64 [H,8] = (cx,cy,cz, ax,ay,az, r, valid)."""
    pts = _points(points)
    tri = np.ascontiguousarray(np.asarray(triples), dtype=np.int64).reshape(-1, 3)
    models = np.zeros((tri.shape[0], 8))
    check(_lib.load().pyqsm_ransac_models(_p(pts), pts.shape[0], _p(tri), tri.shape[0],
                                          _p(models), int(device)))
    return models


def ransac_count(points, models, shape: str = "circle", thresh: float = 0.2, device: int = 0):
    """Inlier count of every hypothesis, int32 [H]."""
    pts = _points(points)
    m = np.ascontiguousarray(models, dtype=np.float64).reshape(-1, 8)
    counts = np.zeros(m.shape[0], dtype=np.int32)
    check(_lib.load().pyqsm_ransac_count(_p(pts), pts.shape[0], _p(m), m.shape[0], SHAPES[shape],
                                         float(thresh), _p(counts), int(device)))
    return counts


# ---------------------------------------------------------------- contraction solve

def _csr(L):
    """(indptr i32, indices i32, data f64, n) of a scipy sparse matrix or a
    (indptr, indices, data) triple."""
    if isinstance(L, tuple):
        indptr, indices, data = L
    else:
        L = L.tocsr()
        L.sort_indices()
        indptr, indices, data = L.indptr, L.indices, L.data
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    data = np.ascontiguousarray(data, dtype=np.float64)
    return indptr, indices, data, indptr.shape[0] - 1


def lbc_solve(L, wl, wh, pts, rtol: float = 1e-10, max_it: int = 20000, device: int = 0):
    """Solve (W_L L' L W_L + W_H^2) x = W_H^2 p (A = [L W_L ; W_H], skeletonize.py:164) for the
    three coordinates.
    Returns (x [n,3], iterations, relative residuals [3])."""
    indptr, indices, data, n = _csr(L)
    p = _points(pts)
    if p.shape[0] != n:
        raise ValueError("L and pts disagree on n")
    wl = np.ascontiguousarray(np.broadcast_to(np.asarray(wl, dtype=np.float64), (n,)))
    wh = np.ascontiguousarray(np.broadcast_to(np.asarray(wh, dtype=np.float64), (n,)))
    out = np.empty_like(p)
    iters = i32(0)
    resid = np.zeros(3)
    rc = _lib.load().pyqsm_lbc_solve(_p(indptr), _p(indices), _p(data), n, _p(wl), _p(wh), _p(p),
                                     float(rtol), int(max_it), _p(out), ctypes.byref(iters),
                                     _p(resid), int(device))
    if rc not in (0, -6):
        check(rc)
    return out, int(iters.value), resid, rc == 0


def spmv3(L, x, device: int = 0):
    indptr, indices, data, n = _csr(L)
    x = _points(x)
    y = np.empty_like(x)
    check(_lib.load().pyqsm_spmv3(_p(indptr), _p(indices), _p(data), n, _p(x), _p(y), int(device)))
    return y


def clamp(pts: np.ndarray, lo, hi, device: int = 0) -> np.ndarray:
    """In-place clamp of a C-contiguous float64 [n,3] array."""
    if not (isinstance(pts, np.ndarray) and pts.dtype == np.float64 and pts.flags.c_contiguous):
        raise ValueError("clamp works in place on a C-contiguous float64 array")
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    check(_lib.load().pyqsm_clamp(_p(pts), pts.shape[0], _p(lo), _p(hi), int(device)))
    return pts


# ---------------------------------------------------------------- radius queries

def ball_query(points, center, radius: float, device: int = 0) -> np.ndarray:
    """Ascending int64 indices of the points with |p - center| <= radius."""
    pts = _points(points)
    ctr = np.ascontiguousarray(center, dtype=np.float64).reshape(3)
    out = np.empty(max(pts.shape[0], 1), dtype=np.int64)
    cnt = i64(0)
    check(_lib.load().pyqsm_ball_query(_p(pts), pts.shape[0], _p(ctr), float(radius), _p(out),
                                       ctypes.byref(cnt), int(device)))
    return out[:cnt.value].copy()


def radius_mark(src, queries, radius: float, k: int = 500, device: int = 0):
    """(mask bool [n], counts int32 [m]): source points that are among the k nearest
    neighbours within `radius` (strict) of at least one query point."""
    s = _points(src)
    q = _points(queries)
    mark = np.zeros(s.shape[0], dtype=np.uint8)
    counts = np.zeros(q.shape[0], dtype=np.int32)
    check(_lib.load().pyqsm_radius_mark(_p(s), s.shape[0], _p(q), q.shape[0], float(radius),
                                        int(k), _p(mark), _p(counts), int(device)))
    return mark.astype(bool), counts


# ---------------------------------------------------------------- down-sampling

def radius_label(src, queries, query_labels, radius: float, k: int = 500, device: int = 0):
    """int32 [n]: for every source point the smallest label among the query points that
    select it (k nearest within ``radius``, strict), -1 where none does; and the per-query
    neighbour counts."""
    s = _points(src)
    q = _points(queries)
    ql = np.ascontiguousarray(query_labels, dtype=np.int32).reshape(-1)
    if ql.shape[0] != q.shape[0]:
        raise ValueError("one label per query point")
    lab = np.empty(max(s.shape[0], 1), dtype=np.int32)
    counts = np.zeros(max(q.shape[0], 1), dtype=np.int32)
    check(_lib.load().pyqsm_radius_label(_p(s), s.shape[0], _p(q), q.shape[0], _p(ql), float(radius),
                                         int(k), _p(lab), _p(counts), int(device)))
    return lab[:s.shape[0]], counts[:q.shape[0]]


def radius_knn(src, queries, radius: float, k: int = 500, device: int = 0):
    """(dist f64 [m,k], idx int64 [m,k]) like ``cKDTree(src).query(queries, k,
    distance_upper_bound=radius)``: ascending by (distance, index), padded with inf / n."""
    s = _points(src)
    q = _points(queries)
    m = q.shape[0]
    idx = np.empty((m, int(k)), dtype=np.int64)
    dist = np.empty((m, int(k)), dtype=np.float64)
    check(_lib.load().pyqsm_radius_knn(_p(s), s.shape[0], _p(q), m, float(radius), int(k), _p(idx),
                                       _p(dist), int(device)))
    return dist, idx


def fps(points, num_samples: int, start_index: int = 0, device: int = 0) -> np.ndarray:
    """Farthest-point sampling: int32 indices in selection order."""
    pts = _points(points)
    out = np.empty(int(num_samples), dtype=np.int32)
    check(_lib.load().pyqsm_fps(_p(pts), pts.shape[0], int(num_samples), int(start_index),
                                _p(out), int(device)))
    return out


# ---------------------------------------------------------------- Laplacian

def _adopt(lib, ptr, ctype, count: int, dtype):
    """NumPy array over a buffer the library malloc'ed; the buffer is released with
    ``pyqsm_free`` once the array and every view of it are gone."""
    buf = (ctype * count).from_address(ptr.value)
    weakref.finalize(buf, lib.pyqsm_free, ctypes.c_void_p(ptr.value))
    return np.frombuffer(buf, dtype=dtype, count=count)


def host_empty(shape, dtype=np.float64) -> np.ndarray:
    """``np.empty`` in the library's pool of page-locked host buffers (``pyqsm_host_alloc``): what
    the device writes into it arrives at link speed and without blocking the calling thread. The
    buffer goes back to the pool when the array and every view of it are gone."""
    shape = tuple(int(q) for q in np.atleast_1d(shape))
    dt = np.dtype(dtype)
    count = int(np.prod(shape)) if len(shape) else 1
    lib = _lib.load()
    ptr = lib.pyqsm_host_alloc(max(count * dt.itemsize, 1))
    if not ptr:
        raise MemoryError("pyqsm_host_alloc failed")
    buf = (ctypes.c_uint8 * max(count * dt.itemsize, 1)).from_address(ptr)
    weakref.finalize(buf, lib.pyqsm_free, ctypes.c_void_p(ptr))
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def pc_laplacian(points, k: int = 30, moll: float = 1e-5, device: int = 0, seg_start=None):
    """(indptr, indices, data) CSR triple and lumped mass [n]. ``seg_start`` (int64 [S+1], from 0
    to n): the points are S clouds stacked into one array and the mollification length is taken
    per cloud (``pyqsm_pc_laplacian_seg``)."""
    pts = _points(points)
    n = pts.shape[0]
    lib = _lib.load()
    nnz = i64(0)
    ip, ix, dv = vp(), vp(), vp()
    mass = np.empty(n, dtype=np.float64)
    if seg_start is not None and len(seg_start) > 2:
        ss = np.ascontiguousarray(seg_start, dtype=np.int64)
        check(lib.pyqsm_pc_laplacian_seg(_p(pts), n, _p(ss), len(ss) - 1, int(k), float(moll),
                                         ctypes.byref(nnz), ctypes.byref(ip), ctypes.byref(ix),
                                         ctypes.byref(dv), _p(mass), int(device)))
    else:
        check(lib.pyqsm_pc_laplacian(_p(pts), n, int(k), float(moll), ctypes.byref(nnz),
                                     ctypes.byref(ip), ctypes.byref(ix), ctypes.byref(dv), _p(mass),
                                     int(device)))
    # the library's malloc'ed outputs become the NumPy arrays themselves (91 MB per million points
    # that are not copied again); pyqsm_free runs when the last view of a buffer is gone
    indptr = _adopt(lib, ip, ctypes.c_int32, n + 1, np.int32)
    indices = _adopt(lib, ix, ctypes.c_int32, max(nnz.value, 1), np.int32)[:nnz.value]
    data = _adopt(lib, dv, ctypes.c_double, max(nnz.value, 1), np.float64)[:nnz.value]
    return (indptr, indices, data), mass


def extreme_points(points, dirs, device: int = 0) -> np.ndarray:
    """Index of the point with the largest ``x . d`` for every row ``d`` of ``dirs`` [D,3]."""
    pts = _points(points)
    d = np.ascontiguousarray(np.asarray(dirs, dtype=np.float64).reshape(-1, 3))
    idx = np.empty(len(d), dtype=np.int64)
    check(_lib.load().pyqsm_extreme_points(_p(pts), pts.shape[0], _p(d), len(d), _p(idx), int(device)))
    return idx


def outside_halfspaces(points, equations, margin: float, device: int = 0) -> np.ndarray:
    """Ascending indices of the points that are NOT strictly inside the polytope
    ``a . x + o <= 0`` (rows ``(a, o)`` of ``equations`` [F,4], F <= 256)."""
    pts = _points(points)
    eq = np.ascontiguousarray(np.asarray(equations, dtype=np.float64).reshape(-1, 4))
    idx = np.empty(pts.shape[0], dtype=np.int64)
    count = i64(0)
    check(_lib.load().pyqsm_outside_halfspaces(_p(pts), pts.shape[0], _p(eq), len(eq), float(margin),
                                               _p(idx), ctypes.byref(count), int(device)))
    return idx[:count.value].copy()


# ---------------------------------------------------------------- the whole contraction loop

def extract_skeleton(points, lo, hi, k: int, moll: float, max_iter: int, termination_ratio: float,
                     contraction_factor: float, attraction_factor: float, max_contraction: float,
                     max_attraction: float, rtol: float, solver_max_it: int, seg_start=None,
                     keep_steps: bool = True, device: int = 0):
    """``pyqsm_extract_skeleton``: the loop of skeletonize.py:240-373 device-resident, for one
    cloud or several stacked ones (``seg_start`` int64 [S+1]; ``lo`` / ``hi`` float64 [S,3]).
    Returns ``(points [n,3], total_shift [n,3], steps [T,n,3] or None, n_steps int32 [S],
    solve_log list of {"iters", "resid", "ok"})``."""
    pts = _points(points)
    n = pts.shape[0]
    ss = (np.array([0, n], dtype=np.int64) if seg_start is None
          else np.ascontiguousarray(seg_start, dtype=np.int64))
    S = len(ss) - 1
    lo = np.ascontiguousarray(np.asarray(lo, dtype=np.float64).reshape(S, 3))
    hi = np.ascontiguousarray(np.asarray(hi, dtype=np.float64).reshape(S, 3))
    T = max(int(max_iter), 1)
    # page-locked results: the per-step shifts leave the device while the next step runs
    out = host_empty(pts.shape)
    total = host_empty(pts.shape)
    steps = host_empty((T, n, 3)) if keep_steps else None   # rows < n_solves are written by the library
    n_steps = np.zeros(S, dtype=np.int32)
    iters = np.zeros(T, dtype=np.int32)
    resid = np.zeros(T)
    ok = np.zeros(T, dtype=np.uint8)
    n_solves = i32(0)
    check(_lib.load().pyqsm_extract_skeleton(
        _p(pts), n, _p(ss), S, int(k), float(moll), int(max_iter), float(termination_ratio),
        float(contraction_factor), float(attraction_factor), float(max_contraction),
        float(max_attraction), _p(lo), _p(hi), float(rtol), int(solver_max_it), _p(out), _p(total),
        _p(steps), _p(n_steps), _p(iters), _p(resid), _p(ok), ctypes.byref(n_solves), int(device)))
    log = [{"iters": int(iters[t]), "resid": [float(resid[t])] * 3, "ok": bool(ok[t])}
           for t in range(n_solves.value)]
    return out, total, steps, n_steps, log
