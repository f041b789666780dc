"""Laplacian-based contraction with pyQSM's names and signatures
(pyQSM/geometry/skeletonize.py), computed by the HIP kernels.

    least_squares_sparse(pts, L, laplacian_weighting, positional_weighting)   :148-180
    set_amplification(step_wise_contraction_amplification, n, term_ratio)     :182-223
    extract_skeleton(pcd, moll, n_neighbors, max_iter, ...)                   :226-373
    point_cloud_laplacian(pts, mollify_factor, n_neighbors)   (robust_laplacian's call)

plus the alias ``skeletonize`` named by BASELINE.json's north_star. The loop keeps
the reference's bookkeeping exactly (see the comments carrying its line numbers);
the solve is a matrix-free conjugate gradient on the GPU instead of three SuperLU
factorisations, and agrees with it to <= 1e-5 relative on the positions.
"""
from __future__ import annotations

import os
import pickle

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

_SK = config["skeletonize"]
# Relative error estimate |B^-2 r| / |x| at which a contraction solve stops: 1000x inside
# north_star's 1e-5 parity bound. Measured against refined SuperLU solutions (20 k points, c = 7, steps
# 1-14) the solves end within 2.6e-7 at 1e-8, 5.9e-7 at 1e-7, 4.2e-6 at 1e-6, and config 3 takes 4.61 /
# 4.07 / 3.54 s: PYQSM_SOLVER_RTOL=1e-7 buys 12 % at the price of a 3x larger run-to-run spread of the
# loop. The default stays at the tolerance every parity test of this repo was measured with.
SOLVER_RTOL = float(os.environ.get("PYQSM_SOLVER_RTOL", "1e-8"))
SOLVER_MAX_IT = 5_000_000   # cap on the total number of inner CG iterations


def point_cloud_laplacian(pts, mollify_factor=1e-5, n_neighbors=30, device: int = 0, seg_start=None):
    """``(L, M)`` like ``robust_laplacian.point_cloud_laplacian``: L a SciPy CSR
    matrix (weak Laplacian, symmetric, zero row sums), M a diagonal SciPy matrix
    (lumped mass), so that ``M.diagonal()`` works as at skeletonize.py:259."""
    from scipy.sparse import csr_matrix, diags
    pts = as_points(pts)
    (indptr, indices, data), mass = hip.pc_laplacian(pts, n_neighbors, mollify_factor,
                                                     device=device, seg_start=seg_start)
    n = len(pts)
    L = csr_matrix((data, indices, indptr), shape=(n, n))
    L.has_sorted_indices = True          # rows are written in column order (laplacian.hip: k_rows)
    L._pyqsm_symmetric = True            # exactly symmetric by construction; spares a 60 ms check per solve
    return L, diags(mass)


def _is_symmetric(L) -> bool:
    if getattr(L, "_pyqsm_symmetric", False):
        return True
    d = (L - L.T)
    if d.nnz == 0:
        return True
    scale = abs(L).max()
    return abs(d).max() <= 1e-12 * max(scale, 1e-300)


class ContractionSolveError(RuntimeError):
    """Raised by ``least_squares_sparse(strict=True)`` when the solve stopped short of ``rtol``."""


def least_squares_sparse(pts, L, laplacian_weighting, positional_weighting, trunk_points=None,
                         rtol: float = SOLVER_RTOL, max_it: int = SOLVER_MAX_IT, device: int = 0,
                         info: list | None = None, strict: bool = False):
    """skeletonize.py:148-180: minimise |W_L-weighted Laplacian|^2 + |W_H (x - pts)|^2
    for each coordinate; returns the new positions float64 [n,3].

    The reference solves (A'A) x = A'b with A = [L W_L ; W_H] by SuperLU; this
    solves the same normal equations by preconditioned CG on the GPU. ``L`` must
    be symmetric (the point-cloud Laplacian is). ``trunk_points`` is accepted and
    unused, as in the reference. If every entry of the solution is NaN the input
    points are returned (:177-179).

    ``info`` (a list) receives one record per call: ``{"iters", "resid" (true relative
    residuals |r|/|b| per coordinate), "ok"}`` — ``ok`` False means the solve stopped on
    stagnation or ``max_it`` with the best iterate (PYQSM_ENOCONV). ``strict=True`` raises
    :class:`ContractionSolveError` instead of only logging it."""
    pts = as_points(pts)
    if not _is_symmetric(L):
        raise ValueError("least_squares_sparse: L must be symmetric")
    x, iters, resid, ok = hip.lbc_solve(L, laplacian_weighting, positional_weighting, pts,
                                        rtol=rtol, max_it=max_it, device=device)
    if info is not None:
        info.append({"iters": iters, "resid": [float(v) for v in resid], "ok": bool(ok)})
    if not ok and strict:
        raise ContractionSolveError(
            f"contraction solve stopped after {iters} CG iterations before the error estimate "
            f"reached {rtol:.1e} (relative residual {resid.max():.3e})")
    if not ok:
        # the best iterate is returned; near cond(A) * 1e-16 the residual cannot go lower
        log.warning(f"contraction solve stopped after {iters} CG iterations before the error "
                    f"estimate reached {rtol:.1e} (relative residual {resid.max():.3e})")
    else:
        log.info(f"contraction solve: {iters} CG iterations, residual {resid.max():.3e}")
    if np.isnan(x).all():
        log.warning("No points in new matrix ")
        return pts
    return x


def set_amplification(step_wise_contraction_amplification, num_pcd_points, termination_ratio):
    """skeletonize.py:182-223 (the reference currently bypasses it, :243-246)."""
    if isinstance(step_wise_contraction_amplification, str):
        if step_wise_contraction_amplification != "auto":
            raise ValueError("Value: {} Not found!".format(step_wise_contraction_amplification))
        if num_pcd_points < 1000:
            contraction_factor, termination_ratio = 1, 0.01
        elif num_pcd_points < 1e4:
            contraction_factor, termination_ratio = 2, 0.007
        elif num_pcd_points < 1e5:
            contraction_factor, termination_ratio = 5, 0.003
        elif num_pcd_points < 0.5 * 1e6:
            contraction_factor, termination_ratio = 5, 0.004
        else:
            contraction_factor, termination_ratio = 5, 0.003
    else:
        contraction_factor = step_wise_contraction_amplification
    return termination_ratio, contraction_factor


_HULL_DIRS = np.array([[x, y, z] for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1)
                       if (x, y, z) != (0, 0, 0)], dtype=np.float64)


def _hull_vertices(pts, device=None):
    """Indices (ascending) of the convex hull's vertices — ``ConvexHull(pts).vertices``. With a
    ``device`` and a large cloud Qhull only sees the points that are not strictly inside the
    polytope of the cloud's extreme points along 26 directions (``pyqsm_extreme_points`` /
    ``pyqsm_outside_halfspaces``: ~1 % of a scan): same vertices, a twentieth of the time."""
    from scipy.spatial import ConvexHull
    if device is None or len(pts) < 20_000:
        return ConvexHull(pts).vertices
    seed = np.unique(hip.extreme_points(pts, _HULL_DIRS, device=device))
    try:
        inner = ConvexHull(pts[seed])
    except Exception:                       # the extreme points are coplanar: no filter
        return ConvexHull(pts).vertices
    margin = 1e-9 * float(np.abs(pts).max())
    cand = hip.outside_halfspaces(pts, inner.equations, margin, device=device)
    return cand[ConvexHull(pts[cand]).vertices]      # in 3-D Qhull lists vertices in input order


def oriented_bounds(pts, device=None):
    """Axis-aligned min / max of the 8 corners of the PCA-oriented bounding box of
    the convex hull — what ``pcd.get_oriented_bounding_box().get_min_bound() /
    get_max_bound()`` yield at skeletonize.py:240-241 (Open3D builds the box from a
    PCA of the hull vertices). Falls back to the plain bounding box for clouds
    that have no 3-D hull. ``device``: see :func:`_hull_vertices` (same result)."""
    pts = np.asarray(pts, dtype=np.float64)
    try:
        hull_pts = pts[_hull_vertices(pts, device)]
    except Exception as e:
        if type(e).__name__ != "QhullError" and not isinstance(e, ValueError):
            raise
        return pts.min(axis=0), pts.max(axis=0)
    mean = hull_pts.mean(axis=0)
    cov = np.cov((hull_pts - mean).T, bias=True)
    w, v = np.linalg.eigh(cov)
    R = v[:, np.argsort(w)[::-1]]
    R[:, 0] = np.cross(R[:, 1], R[:, 2])
    local = (hull_pts - mean) @ R
    lo, hi = local.min(axis=0), local.max(axis=0)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1])
                        for z in (lo[2], hi[2])]) @ R.T + mean
    return corners.min(axis=0), corners.max(axis=0)


def extract_skeleton(pcd, moll=_SK["moll"], n_neighbors=_SK["n_neighbors"],
                     max_iter=_SK["max_iter"], debug=False,
                     termination_ratio=_SK["termination_ratio"],
                     contraction_factor=_SK["init_contraction"],
                     attraction_factor=_SK["init_attraction"],
                     max_contraction=_SK["max_contraction"],
                     max_attraction=_SK["max_attraction"],
                     step_wise_contraction_amplification=_SK["step_wise_contraction_amplification"],
                     cmag_save_file="", min_contraction=0, laplacian=None, device: int = 0,
                     strict: bool = False, engine=None):
    """skeletonize.py:226-373. Returns ``(contracted, total_point_shift,
    shift_by_step)``: the contracted cloud (a PointCloud with ``.points``), the
    accumulated shift float64 [n,3] and the list of per-iteration shifts.

    ``laplacian`` (optional) replaces the point-cloud Laplacian: a callable
    ``pts -> (L, M)``. The shift pickles of the reference (:311-317, :353-367) are
    written only when ``cmag_save_file`` is non-empty.

    The returned cloud carries ``solve_log``: one ``{"iters", "resid", "ok"}`` record per
    contraction solve (see :func:`least_squares_sparse`); ``strict=True`` makes a solve that
    misses ``SOLVER_RTOL`` raise instead of feeding its best iterate to the next step.

    ``engine="native"`` runs the loop inside the library (``pyqsm_extract_skeleton``): the
    points, the Laplacian and the weights stay in HBM between the steps instead of going through
    NumPy / SciPy objects after every call. Same bookkeeping, the SAME BITS as ``engine="python"``
    (tests/test_gpu_native_loop.py), 20 % less wall time at a million points. ``laplacian=`` hooks,
    ``debug`` and the shift pickles need the Python loop, which is also what a caller gets who has
    replaced this module's ``least_squares_sparse`` / ``point_cloud_laplacian``; ``engine=None``
    (the default) picks accordingly."""
    pts = as_points(pcd)
    if engine is None:
        hooked = (laplacian is not None or debug or cmag_save_file
                  or least_squares_sparse is not _least_squares_sparse_original
                  or point_cloud_laplacian is not _point_cloud_laplacian_original)
        engine = "python" if hooked else "native"
    if engine == "native":
        if laplacian is not None or debug or cmag_save_file:
            raise ValueError("engine='native' does not take a laplacian hook, debug or cmag_save_file")
        lo, hi = oriented_bounds(pts, device=device)                       # :240-241
        out, total, steps, n_steps, slog = hip.extract_skeleton(
            pts, lo, hi, n_neighbors, moll, max_iter, termination_ratio, contraction_factor,
            attraction_factor, max_contraction, max_attraction, SOLVER_RTOL, SOLVER_MAX_IT,
            device=device)
        if strict and not all(q["ok"] for q in slog):
            raise ContractionSolveError("a contraction solve stopped before the error estimate "
                                        f"reached {SOLVER_RTOL:.1e}")
        contracted = PointCloud(out)
        contracted.solve_log = slog
        return contracted, total, [steps[t] for t in range(int(n_steps[0]))]
    if engine != "python":
        raise ValueError("engine must be 'python' or 'native'")
    solve_log = []
    allowed_range = oriented_bounds(pts, device=device)                # :240-241
    lo, hi = np.asarray(allowed_range[0]), np.asarray(allowed_range[1])
    if laplacian is None:
        def laplacian(p):
            return point_cloud_laplacian(p, mollify_factor=moll, n_neighbors=n_neighbors,
                                         device=device)
    max_iteration_steps = max_iter
    log.info("generating laplacian")
    L, M = laplacian(pts)                                              # :253-255
    M_list = [M.diagonal()]                                            # :259
    positional_weights = attraction_factor * np.ones(M.shape[0])       # :264
    laplacian_weights = (contraction_factor * 10 ** 3 * np.sqrt(np.mean(M.diagonal()))
                         * np.ones(M.shape[0]))                        # :265
    iteration = 0
    volume_ratio = 1
    pts_current = pts
    shift_by_step = []
    total_point_shift = np.zeros_like(pts_current)

    def _dump(suffix):
        if cmag_save_file:
            try:
                with open(f"{cmag_save_file}{suffix}", "wb") as f:
                    pickle.dump(shift_by_step, f)
            except Exception as e:  # the reference only prints (:318-324)
                log.warning(f"error in cmag saving: {e}")

    while volume_ratio > termination_ratio:                            # :279
        log.info(f"{volume_ratio=}, {np.mean(laplacian_weights)=}, {np.mean(positional_weights)=}")
        pts_new = least_squares_sparse(pts=pts_current, L=L,
                                       laplacian_weighting=laplacian_weights,
                                       positional_weighting=positional_weights, device=device,
                                       info=solve_log, strict=strict)
        if (pts_new == pts_current).all():                             # :287-289
            log.info("No more contraction in last iter, ending run.")
            break
        pts_new = hip.clamp(np.ascontiguousarray(pts_new, dtype=np.float64).copy(), lo, hi,
                            device=device)                             # :291-296
        pcd_point_shift = pts_current - pts_new                        # :304
        total_point_shift += pcd_point_shift
        pts_current = pts_new
        shift_by_step.append(pcd_point_shift)
        if debug or iteration == 0:                                    # :310-317
            _dump("_shift.pkl")
        laplacian_weights = laplacian_weights * contraction_factor     # :329
        positional_weights = positional_weights * np.sqrt(M_list[0] / M.diagonal())   # :331
        laplacian_weights = np.clip(laplacian_weights, 0.1, max_contraction)           # :334
        positional_weights = np.clip(positional_weights, 0.1, max_attraction)          # :335
        M_list.append(M.diagonal())                                    # :337 (the M just used)
        iteration += 1
        L, M = laplacian(pts_current)                                  # :341-343
        volume_ratio = np.mean(M_list[-1]) / np.mean(M_list[0])        # :349 (lags one iteration)
        log.info(f"Completed iteration {iteration}")
        if iteration >= max_iteration_steps:                           # :353-360
            _dump("_tpshift.pkl")
            break
        if volume_ratio < termination_ratio:                           # :361-367
            _dump("_tpshift.pkl")
    log.info(f"Finished after {iteration} iterations")
    contracted = PointCloud(pts_current)
    contracted.solve_log = solve_log
    return contracted, total_point_shift, shift_by_step


skeletonize = extract_skeleton   # BASELINE.json north_star name
# what engine=None compares the module's current functions with (a replaced one is a hook)
_least_squares_sparse_original = least_squares_sparse
_point_cloud_laplacian_original = point_cloud_laplacian


def _pack_groups(sizes, group_points):
    """Greedy packing of cloud indices (largest first) into groups of at most ``group_points``
    points (a cloud larger than that gets a group of its own)."""
    groups, loads = [], []
    for j in sorted(range(len(sizes)), key=lambda q: -sizes[q]):
        for g in range(len(groups)):
            if loads[g] + sizes[j] <= group_points:
                groups[g].append(j)
                loads[g] += sizes[j]
                break
        else:
            groups.append([j])
            loads.append(sizes[j])
    return groups


def _lattice_offsets(pts):
    """Translation of every cloud of a group to its own node of a coarse xy lattice, box centre on
    the node. The Laplacian of the stacked clouds is block diagonal only if every point finds its
    k nearest neighbours inside its own cloud: the GAP between neighbouring boxes (pitch minus the
    largest extent along that axis) is therefore wider than the largest box DIAGONAL — no point
    of a cloud with more than k points can then be nearer to a foreign point than to all of its
    own (round 2 used 2 x extent + 1, a gap of extent + 1 against a diagonal of up to sqrt(3) x
    extent; the library checks the outcome either way: k_check_segments). Translation changes
    fp64 results only by the rounding of the shifted coordinates (~1e-16 x pitch / spacing)."""
    S = len(pts)
    ext = np.array([p.max(0) - p.min(0) for p in pts])
    diag = float(np.sqrt((ext ** 2).sum(1)).max())
    pitch_x = float(ext[:, 0].max()) + diag + 1.0
    pitch_y = float(ext[:, 1].max()) + diag + 1.0
    cols = int(np.ceil(np.sqrt(S)))
    offs = np.array([[(j % cols) * pitch_x, (j // cols) * pitch_y, 0.0] for j in range(S)])
    return offs - np.array([0.5 * (p.max(0) + p.min(0)) for p in pts])


def _contract_group_native(clouds, moll, n_neighbors, max_iter, termination_ratio, contraction_factor,
                           attraction_factor, max_contraction, max_attraction, device):
    """:func:`_contract_group` inside the library: one ``pyqsm_extract_skeleton`` call with the
    clouds as segments (same lattice layout, same per-cloud bookkeeping)."""
    S = len(clouds)
    pts = [np.array(as_points(c), dtype=np.float64) for c in clouds]
    sizes = np.array([len(p) for p in pts])
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    offs = _lattice_offsets(pts)
    bounds = [oriented_bounds(p, device=device) for p in pts]
    lo = np.array([b[0] + o for b, o in zip(bounds, offs)])
    hi = np.array([b[1] + o for b, o in zip(bounds, offs)])
    cur = np.concatenate([p + o for p, o in zip(pts, offs)])
    out, total, steps, n_steps, slog = hip.extract_skeleton(
        cur, lo, hi, n_neighbors, moll, max_iter, termination_ratio, contraction_factor,
        attraction_factor, max_contraction, max_attraction, SOLVER_RTOL, SOLVER_MAX_IT,
        seg_start=start, device=device)
    res = []
    for j in range(S):
        a, b = start[j], start[j + 1]
        pc = PointCloud(out[a:b] - offs[j])
        pc.solve_log = slog[: int(n_steps[j])]
        res.append((pc, total[a:b].copy(), [steps[t, a:b].copy() for t in range(int(n_steps[j]))]))
    return res


def _contract_group(clouds, moll, n_neighbors, max_iter, termination_ratio, contraction_factor,
                    attraction_factor, max_contraction, max_attraction, device):
    """extract_skeleton's loop (skeletonize.py:240-373) for SEVERAL clouds at once: they are laid
    out side by side (every cloud moved to its own cell of a coarse lattice, far from the
    others), so that ONE Laplacian build and ONE contraction solve per step serve them all — the
    Laplacian of the union is block diagonal, one block per cloud — while every piece of the
    loop's bookkeeping (initial weight from the cloud's own mean mass :265, W_H update from its
    own masses :331, clamp into its own oriented bounds :291-296, volume ratio and termination
    :279,349,353) stays per cloud. A 50 k-point tree alone is a chain of ~25 dependent 3-us
    kernels per multigrid-CG iteration; eight of them in one system cost little more per
    iteration than one."""
    S = len(clouds)
    pts = [np.array(as_points(c), dtype=np.float64) for c in clouds]
    sizes = np.array([len(p) for p in pts])
    seg = np.repeat(np.arange(S), sizes)
    start = np.concatenate([[0], np.cumsum(sizes)])
    offs = _lattice_offsets(pts)
    bounds = [oriented_bounds(p, device=device) for p in pts]           # :240-241, per cloud
    lo = np.concatenate([np.tile(b[0] + o, (n, 1)) for b, o, n in zip(bounds, offs, sizes)])
    hi = np.concatenate([np.tile(b[1] + o, (n, 1)) for b, o, n in zip(bounds, offs, sizes)])
    cur = np.concatenate([p + o for p, o in zip(pts, offs)])

    def seg_mean(v):   # np.mean per cloud: the summation order of the single-cloud loop (:265)
        return np.array([v[a:b].mean() for a, b in zip(start[:-1], start[1:])])

    L, M = point_cloud_laplacian(cur, mollify_factor=moll, n_neighbors=n_neighbors, device=device,
                                 seg_start=start)
    m_used = M.diagonal()
    m0 = m_used.copy()
    mean0 = seg_mean(m0)
    wh = attraction_factor * np.ones(len(cur))                          # :264
    wl_seg = contraction_factor * 10 ** 3 * np.sqrt(mean0)              # :265, per cloud
    active = np.ones(S, dtype=bool)
    iteration = np.zeros(S, dtype=np.int64)
    volume_ratio = np.ones(S)
    total = np.zeros_like(cur)
    steps = [[] for _ in range(S)]
    solve_log = [[] for _ in range(S)]
    while active.any():
        active &= volume_ratio > termination_ratio                      # :279
        if not active.any():
            break
        info = []
        new = least_squares_sparse(cur, L, wl_seg[seg], wh, device=device, info=info)
        same = np.array([(new[start[j]:start[j + 1]] == cur[start[j]:start[j + 1]]).all()
                         for j in range(S)])
        active &= ~same                                                 # :287-289
        act_pt = active[seg]
        new = np.where(act_pt[:, None], np.minimum(np.maximum(new, lo), hi), cur)   # :291-296
        shift = cur - new
        total += shift
        cur = new
        for j in np.flatnonzero(active):
            steps[j].append(shift[start[j]:start[j + 1]].copy())
            solve_log[j].append(dict(info[0]))
        wl_seg = np.where(active, np.clip(wl_seg * contraction_factor, 0.1, max_contraction), wl_seg)
        wh = np.where(act_pt, np.clip(wh * np.sqrt(m0 / m_used), 0.1, max_attraction), wh)   # :331-335
        mean_used = seg_mean(m_used)                                    # M_list[-1] of :337,349
        iteration += active
        L, M = point_cloud_laplacian(cur, mollify_factor=moll, n_neighbors=n_neighbors, device=device,
                                     seg_start=start)
        m_used = M.diagonal()
        volume_ratio = np.where(active, mean_used / mean0, volume_ratio)
        active &= iteration < max_iter                                  # :353-360
    out = []
    for j in range(S):
        a, b = start[j], start[j + 1]
        pc = PointCloud(cur[a:b] - offs[j])
        pc.solve_log = solve_log[j]
        out.append((pc, total[a:b].copy(), steps[j]))
    return out


def extract_skeleton_batch(pcds, moll=_SK["moll"], n_neighbors=_SK["n_neighbors"],
                           max_iter=_SK["max_iter"], termination_ratio=_SK["termination_ratio"],
                           contraction_factor=_SK["init_contraction"],
                           attraction_factor=_SK["init_attraction"],
                           max_contraction=_SK["max_contraction"],
                           max_attraction=_SK["max_attraction"], device: int = 0,
                           group_points: int = 400_000, workers: int = 4, engine: str = "python"):
    """``extract_skeleton`` for MANY clouds (the per-cluster calls of qsm_generation.py:182-316):
    returns one ``(contracted, total_point_shift, shift_by_step)`` triple per input cloud, in
    input order. Clouds are packed into groups of up to ``group_points`` points; a group is
    contracted as ONE block-diagonal system per step (:func:`_contract_group`) and ``workers``
    host threads contract groups concurrently (the library keeps a stream per thread).

    Every cloud keeps its own weights, bounds, termination and mollification length
    (``pyqsm_pc_laplacian_seg``); what the clouds of a group share is the CG scalars of the
    solve, so results agree with the per-cloud loop to the solver's tolerance
    (tests/test_gpu_batch.py), not bit for bit."""
    from concurrent.futures import ThreadPoolExecutor
    clouds = list(pcds)
    sizes = [len(as_points(c)) for c in clouds]
    # A cloud with too few points to have n_neighbors neighbours of its own (the small clusters of
    # qsm_generation.py:182-316) cannot share a system with others — its missing neighbours would
    # come from the next cloud. Below 2 k + 1 points it takes the single-cloud call, which handles
    # n <= k.
    small = [j for j, n in enumerate(sizes) if n < 2 * int(n_neighbors) + 1]
    big = [j for j in range(len(clouds)) if sizes[j] >= 2 * int(n_neighbors) + 1]
    groups = [[big[q] for q in g] for g in _pack_groups([sizes[j] for j in big], int(group_points))]
    groups += [[j] for j in small]
    args = (moll, n_neighbors, max_iter, termination_ratio, contraction_factor, attraction_factor,
            max_contraction, max_attraction, device)

    fn = _contract_group_native if engine == "native" else _contract_group

    def run(g):
        if len(g) == 1 and sizes[g[0]] < 2 * int(n_neighbors) + 1:
            return g, [extract_skeleton(clouds[g[0]], moll=moll, n_neighbors=n_neighbors, max_iter=max_iter,
                                        termination_ratio=termination_ratio, contraction_factor=contraction_factor,
                                        attraction_factor=attraction_factor, max_contraction=max_contraction,
                                        max_attraction=max_attraction, device=device,
                                        engine="native" if engine == "native" else "python")]
        return g, fn([clouds[j] for j in g], *args)

    results = [None] * len(clouds)
    with ThreadPoolExecutor(max_workers=max(1, int(workers))) as pool:
        for g, res in pool.map(run, groups):
            for j, r in zip(g, res):
                results[j] = r
    return results



# --------------------------------------------------------------------------------------
# After the contraction: down-sample, span, simplify (skeletonize.py:36-146).
# Farthest-point sampling and the kNN search run on the GPU; the minimum spanning tree
# (SciPy, which is what mistree calls) and the degree-2 collapsing (networkx, as in the
# reference) are small host-side graph steps.

class LineSet:
    """points float64 [m,3] + lines int [e,2] (stands in for open3d.geometry.LineSet)."""

    def __init__(self, points, lines):
        self.points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
        self.lines = np.asarray(lines, dtype=np.int64).reshape(-1, 2)


def farthest_point_down_sample(points, num_samples: int, device: int = 0):
    """Indices and coordinates of ``num_samples`` points chosen like Open3D's
    ``PointCloud.farthest_point_down_sample`` (start at point 0, then always the point
    farthest from everything chosen so far; skeletonize.py:132)."""
    pts = as_points(points)
    idx = hip.fps(pts, num_samples, 0, device=device)
    return idx, pts[idx]


def extract_skeletal_graph(skeletal_points: np.ndarray, graph_k_n, device: int = 0):
    """skeletonize.py:36-55: minimum spanning tree of the k-nearest-neighbour graph
    (mistree.construct_mst = sklearn kneighbors_graph + SciPy minimum_spanning_tree) as
    a networkx graph whose nodes carry ``pos``. Returns ``(mst_graph, rx_graph)``;
    rustworkx is optional and ``rx_graph`` is None when it is not installed."""
    import networkx as nx
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import minimum_spanning_tree
    pts = as_points(skeletal_points)
    n = len(pts)
    k = int(min(graph_k_n, max(n - 1, 1)))
    idx, d2 = hip.knn(pts, k, True, device=device)
    valid = idx < n
    rows = np.repeat(np.arange(n), k)[valid.ravel()]
    cols = idx.ravel()[valid.ravel()]
    dist = np.sqrt(d2.ravel()[valid.ravel()])
    graph = csr_matrix((dist, (rows, cols)), shape=(n, n))
    mst = minimum_spanning_tree(graph).tocoo()
    edges = np.stack([mst.row, mst.col], axis=1)
    mst_graph = nx.Graph(edges.tolist())
    mst_graph.add_nodes_from(range(n))
    for i in range(n):
        mst_graph.nodes[i]["pos"] = pts[i].T
    rx_graph = None
    try:
        import rustworkx as rx
        rx_graph = rx.PyGraph()
        for i in range(n):
            rx_graph.add_node({"pos": pts[i].T})
        rx_graph.add_edges_from([(int(a), int(b), tuple(pts[a] - pts[b])) for a, b in edges])
    except ModuleNotFoundError:
        pass
    return mst_graph, rx_graph


def simplify_graph(G):
    """What skeletonize.py:57-98 computes, as one chain-collapsing pass: every maximal run
    of degree-2 nodes between two other nodes (junctions, leaves) becomes ONE edge whose
    ``data`` list names the nodes of the run, in walking order from the end that comes first
    in the graph's node order.
    Returns ``(graph, kept_node_positions, kept_node_indices)`` with the kept nodes in the
    graph's node order. A graph without degree-2 nodes is returned unchanged with every
    node kept; rings that consist of degree-2 nodes only (no end to start a walk from,
    impossible in the spanning tree extract_topology passes in) are left as they are."""
    import networkx as nx
    kept = [v for v, d in G.degree() if d != 2]
    kept_set = set(kept)
    out = nx.Graph()
    out.add_nodes_from((v, G.nodes[v]) for v in kept)
    walked = set()                                   # degree-2 nodes already assigned to an edge
    for u in kept:
        for first in G.neighbors(u):
            if first in kept_set:                    # a direct edge between two kept nodes
                if not out.has_edge(u, first):
                    out.add_edge(u, first, **G.edges[u, first])
                continue
            if first in walked:
                continue                             # this run was walked from its other end
            run, prev, cur = [], u, first
            while cur not in kept_set:
                run.append(cur)
                walked.add(cur)
                a, b = G.neighbors(cur)
                prev, cur = cur, (b if a == prev else a)
            members = []
            for (x, y) in zip([u] + run, run + [cur]):   # nodes fused into the original edges
                members += G.edges[x, y].get("data", [])
            members += run
            if out.has_edge(u, cur):                 # two runs between the same pair of nodes
                members = out.edges[u, cur].get("data", []) + members
            out.add_edge(u, cur, data=members)
    rings = [v for v in G.nodes if v not in kept_set and v not in walked]
    if rings:
        out.add_nodes_from((v, G.nodes[v]) for v in rings)
        out.add_edges_from((x, y, d) for x, y, d in G.edges(rings, data=True))
        kept = [v for v in G.nodes if v in kept_set or v in set(rings)]
    return out, [G.nodes[v]["pos"] for v in kept], kept


def simplify_and_update(graph):
    """skeletonize.py:100-111: simplify, then relabel the surviving nodes 0..m-1 in the
    order their positions are listed."""
    import networkx as nx
    G_simplified, node_pos, node_idx = simplify_graph(graph)
    skeleton_cleaned_points = np.vstack(node_pos) if node_pos else np.zeros((0, 3))
    mapping = {node: i for i, node in enumerate(node_idx) if node in G_simplified}
    return nx.relabel_nodes(G_simplified, mapping), skeleton_cleaned_points, mapping


def extract_topology(contracted, graph_k_n=_SK["graph_k_n"], device: int = 0):
    """skeletonize.py:113-146. ``contracted`` is the contracted cloud (array or object
    with ``.points``). Returns ``(topology, topology_graph, skeleton, skeleton_points,
    skeleton_graph, rx_graph, mapping)`` like the reference."""
    pts = as_points(contracted)
    # artefacts collapsed onto the origin (:118-124)
    norms = np.linalg.norm(pts, axis=1)
    near = int(np.argmin(norms))
    if norms[near] <= 0.01:
        keep = np.linalg.norm(pts - pts[near], axis=1) > 0.01
        pts = pts[keep]
    fps_points = max(int(pts.shape[0] * 0.1), 15)                       # :128-129
    fps_points = min(fps_points, pts.shape[0])
    log.info(f"down sampling contracted, starting with {len(pts)} points, taking {fps_points}")
    _, skeleton_points = farthest_point_down_sample(pts, fps_points, device=device)
    skeleton = PointCloud(skeleton_points)
    skeleton_graph, rx_graph = extract_skeletal_graph(skeleton_points, graph_k_n, device=device)
    topology_graph, topology_points, mapping = simplify_and_update(skeleton_graph)
    topology = LineSet(topology_points, list(topology_graph.edges()))
    return (topology, topology_graph, skeleton, skeleton_points, skeleton_graph, rx_graph,
            mapping)



def _unique_rows_mm(r):
    """``np.unique(r, axis=0)`` for rows already rounded to 3 decimals: the rows become one integer
    key each (millimetres, 21 bits per coordinate, x most significant — the same lexicographic
    order), and a 1-D unique is five times cheaper than the row-wise one (0.6 ms per cylinder of
    2000 points, 2.8 s for the 4 400 cylinders of a million-point cloud)."""
    k = np.rint(r * 1000.0).astype(np.int64)
    k -= k.min(axis=0)
    if k.max() >= 1 << 21:                               # more than 2 km across: the slow way
        return np.unique(r, axis=0)
    key = (k[:, 0] << 42) | (k[:, 1] << 21) | k[:, 2]
    _, first = np.unique(key, return_index=True)
    return r[first]


def skeleton_to_QSM(topology, topology_graph, total_point_shift, test=True):
    """skeletonize.py:375-441: one cylinder per topology edge, from the edge's end
    points, with radius = mean contraction distance of the vertices that were fused into
    that edge (``data`` lists of :func:`simplify_graph`; indices are used exactly as the
    reference uses them, i.e. directly into ``total_point_shift``).

    Returns ``(all_cyl_pcd, cyls, cyl_objects, radii)``: the union of the sampled cylinder
    surfaces, one point cloud per cylinder, the :class:`Cylinder` objects and the radii.
    Sampling follows skspatial's ``Cylinder.to_points(n_angles=20).round(3).unique()``
    in spirit (scikit-spatial is not installable here: 100 steps along the axis)."""
    try:
        from .cloud import Cylinder
    except ImportError:
        from pyqsm_amd.geometry.cloud import Cylinder
    edge_to_orig = {}
    for a, b, d in topology_graph.edges(data=True):
        edge_to_orig[(a, b)] = d.get("data")
        edge_to_orig[(b, a)] = d.get("data")
    points = np.asarray(topology.points)
    lines = np.asarray(topology.lines)
    contraction_dist = np.linalg.norm(np.asarray(total_point_shift), axis=1)
    cyls, cyl_objects, radii = [], [], []
    for line in lines:
        start, end = points[line[0]], points[line[1]]
        orig = edge_to_orig.get((int(line[0]), int(line[1])))
        if not orig:
            continue                                     # an edge that absorbed no vertex
        radius = float(np.mean(contraction_dist[np.asarray(orig, dtype=np.int64)]))
        axis = end - start
        height = float(np.linalg.norm(axis))
        if height == 0:
            continue
        cyl = Cylinder((start + end) / 2.0, radius, height, axis)
        u, v = cyl._frame()
        ang = np.linspace(0.0, 2.0 * np.pi, 20, endpoint=False)
        along = np.linspace(-height / 2.0, height / 2.0, 100)
        ring = radius * (np.cos(ang)[:, None] * u + np.sin(ang)[:, None] * v)
        pts = (cyl.center + ring[None, :, :] + along[:, None, None] * cyl.axis).reshape(-1, 3)
        pts = _unique_rows_mm(pts.round(3))
        cyls.append(PointCloud(pts))
        cyl_objects.append(cyl)
        radii.append(radius)
    all_pts = np.concatenate([c.points for c in cyls]) if cyls else np.zeros((0, 3))
    return PointCloud(all_pts), cyls, cyl_objects, radii
