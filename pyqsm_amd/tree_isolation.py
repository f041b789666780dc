"""Region growing of seed clusters through a cloud — the compute core of
``pyQSM/tree_isolation.py:63-283`` (``extend_seed_clusters``).

The reference rebuilds nothing per cycle but queries a SciPy KD-tree once per cluster
and cycle (``tree_isolation.py:207-209``) and keeps the ownership of points in a dict
keyed by coordinate tuples. Here one cycle is ONE GPU call for all clusters together
(``pyqsm_radius_label``: every source point gets the smallest cluster index among the
frontier points that select it — exactly what visiting the clusters in index order and
letting the first one keep a free point produces), and ownership is an int32 array.

Not restated: the TensorBoard / drawing / pickling side effects, the ``breakpoint()``s and
the ``input()`` prompt of the reference.
"""
from __future__ import annotations

import logging
from collections import defaultdict

import numpy as np

try:  # flat import style of the reference (pyqsm_amd on sys.path) or package import
    from . import hip
    from ._shadow import fall_through
    from .geometry.cloud import PointCloud, as_points
except ImportError:  # pragma: no cover
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pyqsm_amd import hip
    from pyqsm_amd._shadow import fall_through
    from pyqsm_amd.geometry.cloud import PointCloud, as_points

# names pyQSM's module of the same name defines and this one does not (pyqsm_amd/_shadow.py)
__getattr__ = fall_through(__name__)

log = logging.getLogger("calc")


def labeled_pts_to_lists(labeled_pts, idc_to_label_map={}, file="", draw_cycle=False,
                         save_cycle=False):
    """tree_isolation.py:41-60: ``{point tuple: cluster index}`` -> ``([(label, [pts])],
    [one cloud per label] or None)``. Nothing is drawn or pickled."""
    tree_pts = defaultdict(list)
    for pt, idc in labeled_pts.items():
        tree_pts[idc_to_label_map.get(idc, idc)].append(pt)
    pt_lists = [(label, pts) for label, pts in tree_pts.items()]
    tree_pcds = None
    if draw_cycle:
        tree_pcds = [PointCloud(np.asarray(pts, dtype=np.float64).reshape(-1, 3))
                     for pts in tree_pts.values()]
    return pt_lists, tree_pcds


def _rows_in(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """For every row of ``a`` [n,3]: index of an equal row of ``b`` [m,3], or -1."""
    if len(a) == 0 or len(b) == 0:
        return np.full(len(a), -1, dtype=np.int64)
    va = np.ascontiguousarray(a).view([("", a.dtype)] * 3).reshape(-1)
    vb = np.ascontiguousarray(b).view([("", b.dtype)] * 3).reshape(-1)
    order = np.argsort(vb, kind="stable")
    pos = np.searchsorted(vb[order], va)
    pos[pos == len(vb)] = 0
    hit = vb[order][pos] == va
    return np.where(hit, order[pos], -1)


def extend_seed_clusters(clusters_and_idxs, src_pcd, file_label="", k=200, max_distance=.1,
                         cycles=150, save_every=10, draw_every=10, tb_every=10, order_cutoff=None,
                         exclude_pcd=None, exclude_pts=None, device: int = 0):
    """tree_isolation.py:63-283. ``clusters_and_idxs`` is ``[(label, cluster), ...]``
    (clouds or point arrays); ``src_pcd`` the cloud to grow through. Every cycle, every
    unfinished cluster collects the (up to ``k`` nearest) source points within
    ``max_distance`` of its current frontier; the free ones join it and become the next
    frontier; clusters are served in list order. A cluster is finished when its new
    frontier has fewer than five points (``:256-258``) or — with ``order_cutoff`` — when the
    neighbourhood of its frontier falls into more than ``order_cutoff`` DBSCAN clusters
    (eps 0.15, 20 points, ``:218-233,259-262``).

    Returns ``(tree_pcds, all_nbrs)``: one cloud per label (seed points + grown points, the
    order of the reference's dict: seeds first, then by cycle) and, per cluster, the source
    indices it acquired (the reference extends one list shared by all clusters there — an
    aliasing slip of ``[[]] * n`` — and its callers ignore the value). ``save_every``,
    ``draw_every`` and ``tb_every`` are accepted and ignored."""
    seeds = [(label, as_points(cl)) for label, cl in clusters_and_idxs]
    n_cl = len(seeds)
    src_pts = as_points(src_pcd)
    if exclude_pts is None and exclude_pcd is not None:
        exclude_pts = as_points(exclude_pcd)
    if exclude_pts is not None and len(exclude_pts) and len(src_pts):      # :120-133
        mask, _ = hip.radius_mark(src_pts, as_points(exclude_pts), max_distance, k=k, device=device)
        src_pts = src_pts[~mask.astype(bool)]
    n = len(src_pts)
    owner = np.full(n, -1, dtype=np.int32)
    # points of the source that ARE seed points already carry their cluster (dict lookup
    # by coordinates in the reference); later seeds overwrite earlier ones, as there
    for idc, (_, pts) in enumerate(seeds):
        hit = _rows_in(src_pts, pts) >= 0
        owner[hit] = idc
    frontier = [pts for _, pts in seeds]
    grown = [[] for _ in range(n_cl)]          # source indices per cluster, in order of acquisition
    complete = np.zeros(n_cl, dtype=bool)
    for idc in range(n_cl):
        if len(frontier[idc]) == 0:
            complete[idc] = True
    for cycle_num in range(int(cycles)):
        active = [i for i in range(n_cl) if not complete[i]]
        if not active or n == 0:
            break
        if order_cutoff:
            new_sets = _cycle_with_cutoff(src_pts, owner, frontier, active, k, max_distance, cycle_num,
                                          order_cutoff, complete, device)
        else:
            q = np.concatenate([frontier[i] for i in active])
            ql = np.concatenate([np.full(len(frontier[i]), i, dtype=np.int32) for i in active])
            lab, _ = hip.radius_label(src_pts, q, ql, max_distance, k=k, device=device)
            free = owner < 0
            new_sets = {i: np.flatnonzero(free & (lab == i)) for i in active}
        for i in active:
            new = new_sets.get(i)
            if new is None:
                continue
            if len(new) > 0:
                owner[new] = i
                grown[i].append(new)
                frontier[i] = src_pts[new]
                if len(new) < 5:                                           # :256-258
                    complete[i] = True
                    log.info(f"{i} added to complete")
            else:
                # nothing free within reach: later cycles could not find anything either
                # (points are never released), the reference merely keeps asking
                complete[i] = True
    idc_to_label = {i: label for i, (label, _) in enumerate(seeds)}
    assigned = {}
    for idc, (_, pts) in enumerate(seeds):
        for p in pts:
            assigned[tuple(p)] = idc
    all_nbrs = []
    for idc in range(n_cl):
        idx = np.concatenate(grown[idc]) if grown[idc] else np.zeros(0, dtype=np.int64)
        all_nbrs.append(idx.tolist())
    # grown points enter the dict cycle by cycle, cluster by cluster (the reference's order)
    max_len = max((len(g) for g in grown), default=0)
    for step in range(max_len):
        for idc in range(n_cl):
            if step < len(grown[idc]):
                for j in grown[idc][step]:
                    assigned.setdefault(tuple(src_pts[j]), idc)
    _, tree_pcds = labeled_pts_to_lists(assigned, idc_to_label, draw_cycle=True)
    return tree_pcds, all_nbrs


def _cycle_with_cutoff(src_pts, owner, frontier, active, k, max_distance, cycle_num, order_cutoff,
                       complete, device):
    """One cycle cluster by cluster (needed when the neighbourhoods themselves are
    clustered, tree_isolation.py:218-233)."""
    try:
        from .geometry.point_cloud_processing import cluster_plus
    except ImportError:  # pragma: no cover
        from pyqsm_amd.geometry.point_cloud_processing import cluster_plus
    new_sets = {}
    taken = owner.copy()
    for i in active:
        mask, _ = hip.radius_mark(src_pts, frontier[i], max_distance, k=k, device=device)
        nbrs = np.flatnonzero(mask)
        new = nbrs[taken[nbrs] < 0]
        num_clusters = 0
        if cycle_num > 0 and len(nbrs):
            res = cluster_plus(PointCloud(src_pts[nbrs]), eps=.15, min_points=20, return_pcds=False,
                               from_points=False, draw_result=False)
            num_clusters = len(res)
        taken[new] = i
        new_sets[i] = new
        if num_clusters > order_cutoff:                                    # :259-262
            complete[i] = True
            if len(new) == 0:
                new_sets[i] = None
    return new_sets
