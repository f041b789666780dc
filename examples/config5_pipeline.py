#!/usr/bin/env python3
"""BASELINE.json configs[4] on one GPU, scaled by --scale: scan -> DBSCAN -> per-tree
Laplacian contraction -> RANSAC circles on z-slices -> canopy light simulation.

    python examples/config5_pipeline.py [--scale 0.04]     # 0.04 -> 200 k points, 2 M rays

Every stage goes through the pyQSM-named wrappers (cluster_DBSCAN, extract_skeleton,
fit_shape_RANSAC, cast_rays), i.e. through the ctypes C-ABI into the HIP kernels; one JSON
line with the per-stage times is printed. At --scale 1 this is the 5 M-point / 50 M-ray
configuration.

`--gpus N` (one process, N GPUs of the node): DBSCAN runs on GPU 0, the trees
(whole clusters) are dealt round-robin over the GPUs for contraction and RANSAC — replicas, no
collective — and the ray stage shards its rays over the N GPUs through RCCL
(cast_rays(..., n_devices=N) = pyqsm_cast_rays_multi)."""
import argparse
import gc
import json
from concurrent.futures import ThreadPoolExecutor
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import _lib, synth  # noqa: E402
from pyqsm_amd.geometry.skeletonize import extract_skeleton, extract_skeleton_batch  # noqa: E402
from pyqsm_amd.math_utils.fit import (cluster_DBSCAN, draw_samples, fit_shape_RANSAC,  # noqa: E402
                                      fit_shape_RANSAC_batch)
from pyqsm_amd.set_config import config  # noqa: E402
from pyqsm_amd.viz.ray_casting import cast_rays  # noqa: E402


def run(scale=0.04, skeleton_iters=3, max_trees=2, workers=8, gpus=1, engine="python", batch_workers=6,
        ransac_batch=2, group_points=1_000_000, keep=False):
    """The pipeline; returns the JSON-able record of stage times and, with ``keep``, a second dict
    holding what the stages produced (cluster index lists, per-tree contraction results, per-tree
    slice fits with their samples, per-angle hit arrays) for tests to check."""
    n_avail = _lib.logical_device_count()
    n_gpus = gpus if gpus > 0 else n_avail
    if n_gpus > n_avail:
        raise SystemExit(f"--gpus {n_gpus} but only {n_avail} visible")
    phys = [_lib.physical_device(d) for d in range(n_gpus)]   # logical ranks: all on the real device(s)
    for d in set(phys):
        _lib.require_gpu(d)
    n_points = max(50_000, int(5_000_000 * scale))
    n_rays_per_angle = max(100_000, int(10_000_000 * scale))
    n_tris = max(20_000, int(500_000 * min(1.0, scale * 5)))
    out = {"points": n_points, "rays": 5 * n_rays_per_angle, "tris": n_tris, "gpus": n_gpus}
    kept = {}

    pts = synth.forest(n_points, seed=0)
    t0 = time.perf_counter()
    labels, idxs, noise = cluster_DBSCAN(np.arange(len(pts)), pts, config["dbscan"]["epsilon"],
                                         config["dbscan"]["min_neighbors"], device=phys[0])
    out["dbscan_s"] = time.perf_counter() - t0
    out["clusters"] = len(idxs)
    idxs = sorted(idxs, key=len, reverse=True)
    if keep:
        kept.update(pts=pts, idxs=idxs, noise=noise)

    t0 = time.perf_counter()
    def contract(job):
        k, tree = job                                   # tree k goes to GPU k mod N (replicas)
        return extract_skeleton(pts[tree], max_iter=skeleton_iters, termination_ratio=0.0,
                                device=phys[k % n_gpus])

    trees = idxs[: max_trees]
    if group_points > 0:
        # GPU d takes every n_gpus-th tree and contracts its share in block-diagonal groups
        def contract_share(d):
            share = [pts[t] for t in trees[d::n_gpus]]
            return extract_skeleton_batch(share, max_iter=skeleton_iters, termination_ratio=0.0,
                                          device=phys[d], group_points=group_points,
                                          workers=max(1, batch_workers), engine=engine)

        with ThreadPoolExecutor(max_workers=n_gpus) as pool:
            parts = list(pool.map(contract_share, range(n_gpus)))
        results = [None] * len(trees)
        for d, part in enumerate(parts):
            for j, r in zip(range(d, len(trees), n_gpus), part):
                results[j] = r
    else:
        with ThreadPoolExecutor(max_workers=max(1, workers) * n_gpus) as pool:
            results = list(pool.map(contract, enumerate(trees)))
    shifts = [float(np.linalg.norm(r[1], axis=1).mean()) for r in results]
    if keep:
        kept["skeletons"] = results
    else:
        del results
    gc.collect()      # the stage's page-locked result buffers go back now, not while the next stage is timed
    out["skeleton_s"] = time.perf_counter() - t0
    out["skeleton_trees"] = len(shifts)
    out["skeleton_workers"] = max(1, workers)
    out["mean_contraction_m"] = shifts

    t0 = time.perf_counter()
    stage_t = {"slice": 0.0, "draw": 0.0, "fit": 0.0}

    def fit_slices(job):
        k, tree = job
        ta = time.perf_counter()
        cloud = pts[tree]
        slices = []
        for z0 in np.arange(0.5, 5.5, 0.5):                 # 0.5 m slices of the stem
            sl = cloud[(cloud[:, 2] >= z0) & (cloud[:, 2] < z0 + 0.5)]
            sl = sl[np.hypot(sl[:, 0] - np.median(sl[:, 0]), sl[:, 1] - np.median(sl[:, 1])) < 0.6]
            if len(sl) >= 50:
                slices.append(sl.copy())
        if not slices:
            return [], [], []
        tb = time.perf_counter()
        stage_t["slice"] += tb - ta
        # H = 1000 hypotheses per slice, seed 2; every slice its own stream
        streams = np.random.SeedSequence([2, k]).spawn(len(slices))
        smp = [draw_samples(len(sl), 1000, seed=st) for sl, st in zip(slices, streams)]
        tc = time.perf_counter()
        stage_t["draw"] += tc - tb
        if ransac_batch >= 2:                               # fitted below, all trees of a GPU in one call
            fits = None
        elif ransac_batch:                                  # all slices of the tree in one call
            fits = fit_shape_RANSAC_batch(slices, shape="circle", threshold=0.04, max_radius=0.3 * 1.75,
                                          samples=smp, device=phys[k % n_gpus])
        else:
            fits = [fit_shape_RANSAC(pts=sl, shape="circle", threshold=0.04, max_radius=0.3 * 1.75,
                                     samples=q, device=phys[k % n_gpus]) for sl, q in zip(slices, smp)]
        stage_t["fit"] += time.perf_counter() - tc
        return fits, slices, smp

    with ThreadPoolExecutor(max_workers=max(1, workers) * n_gpus) as pool:
        per_tree = list(pool.map(fit_slices, enumerate(idxs[: max(max_trees, 4)])))
    if ransac_batch >= 2:
        # ONE pyqsm_ransac_batch call per GPU over every slice of its trees (a call per tree was a hundred
        # calls from eight threads, 0.15 s on a good box and 1.2 s on one whose host was busy)
        tc = time.perf_counter()

        def fit_device(d):
            mine = [k for k in range(len(per_tree)) if k % n_gpus == d and per_tree[k][1]]
            if not mine:
                return
            flat = [sl for k in mine for sl in per_tree[k][1]]
            smps = [q for k in mine for q in per_tree[k][2]]
            fits = fit_shape_RANSAC_batch(flat, shape="circle", threshold=0.04, max_radius=0.3 * 1.75,
                                          samples=smps, device=phys[d])
            at = 0
            for k in mine:
                m = len(per_tree[k][1])
                per_tree[k] = (fits[at:at + m], per_tree[k][1], per_tree[k][2])
                at += m

        with ThreadPoolExecutor(max_workers=n_gpus) as pool:
            list(pool.map(fit_device, range(n_gpus)))
        per_tree = [(f if f is not None else [], sl, q) for f, sl, q in per_tree]
        stage_t["fit"] += time.perf_counter() - tc
    radii = [float(f[3]) for fits, _, _ in per_tree for f in fits if f[0] is not None]
    out["ransac_s"] = time.perf_counter() - t0
    out["ransac_thread_seconds"] = {k: round(v, 3) for k, v in stage_t.items()}
    out["ransac_slices"] = sum(len(q[0]) for q in per_tree)
    out["ransac_fits"] = len(radii)
    out["ransac_median_radius_m"] = float(np.median(radii)) if radii else None
    if keep:
        kept["slices"] = per_tree

    verts, tris = synth.canopy_mesh(n_tris)
    angles = (45.0, 90.0, 135.0, 180.0, 225.0)
    lit, hits = [], []
    out["rays_s"] = 0.0
    for az in angles:                                        # one sun angle resident at a time
        rays = synth.sun_rays(verts, n_rays_per_angle, elevation_deg=60.0, azimuth_deg=az)   # synthetic
        t0 = time.perf_counter()                             # input: not part of the stage
        ans = cast_rays((verts, tris), rays=rays, device=phys[0], n_devices=n_gpus if n_gpus > 1 else None)
        out["rays_s"] += time.perf_counter() - t0
        lit.append(float(ans["hit"].mean()))
        if keep:
            hits.append((az, ans["t_hit"], ans["primitive_ids"]))
        del rays, ans
    out["intercepted_fraction"] = lit
    out["total_s"] = out["dbscan_s"] + out["skeleton_s"] + out["ransac_s"] + out["rays_s"]
    if keep:
        kept.update(mesh=(verts, tris), angles=angles, hits=hits, n_rays_per_angle=n_rays_per_angle)
    return out, kept


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=0.04)
    ap.add_argument("--skeleton-iters", type=int, default=3)
    ap.add_argument("--max-trees", type=int, default=2, help="trees that get skeletonised")
    ap.add_argument("--workers", type=int, default=8,
                    help="host threads PER GPU that contract trees concurrently (the library keeps "
                         "one stream and arena per thread; a 50 k-point tree alone is latency-bound)")
    ap.add_argument("--gpus", type=int, default=1, help="GPUs of this node to use (0 = all visible; with "
                                                        "PYQSM_MULTI_FAKE_RANKS=N: logical GPUs on one device)")
    ap.add_argument("--engine", default="python", help="contraction loop of a group: python or native "
                                                         "(pyqsm_extract_skeleton, segments in HBM)")
    ap.add_argument("--batch-workers", type=int, default=6, help="host threads contracting groups")
    ap.add_argument("--ransac-batch", type=int, default=2,
                    help="2: every slice of every tree of a GPU in one pyqsm_ransac_batch call; 1: a call per tree; "
                         "0: a call per slice")
    ap.add_argument("--group-points", type=int, default=1_000_000,
                    help="trees are contracted in block-diagonal groups of up to this many points "
                         "(extract_skeleton_batch); 0 = one extract_skeleton call per tree")
    args = ap.parse_args()
    out, _ = run(scale=args.scale, skeleton_iters=args.skeleton_iters, max_trees=args.max_trees,
                 workers=args.workers, gpus=args.gpus, engine=args.engine, batch_workers=args.batch_workers,
                 ransac_batch=args.ransac_batch, group_points=args.group_points)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
