#!/usr/bin/env python3
"""Benchmark of the pyQSM hot path on MI355X (contract: see the task brief).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Primary metric (BASELINE.json): Mpoints/s of DBSCAN on the 1 M-point synthetic
forest (eps = 0.1, min_neighbors = 10), inputs resident in HBM, one step = one
full clustering (binning + core flags + union-find + labels). DBSCAN does not
shard: at N > 1 every rank clusters its own forest (replicas, weak scaling) and
`value` = all points of all ranks / max-over-ranks time.

Secondary sections on the same JSON line:
  ray_sweep   500 k-triangle canopy x R sun rays, Mray-tri/s (= R*T/t); rays are
              sharded over the ranks, the expanded mesh is broadcast and the per-shard
              results all-gathered by the library's own RCCL communicator (pyqsm_comm_*;
              strong scaling, R fixed)
  knn         k = 20 neighbours on the same cloud
  roofline    dominant kernel of the primary path, HIP-event timed live
  skeleton    the first --skel-iters Laplacian contractions of extract_skeleton on the
              same cloud with the time split Laplacian / solve (N = 1 only: replicas)
  ransac      1000 circle hypotheses x 50 k points (fit_shape_RANSAC's inner loop); the ten z-slices of
              a tree in one pyqsm_ransac_batch call and one by one
  fps         farthest-point sampling of 10 % of the cloud (what extract_topology does next)
  cpu_baseline  scikit-learn DBSCAN (the reference's own call, fit.py:223) on the
              host cores of this box, rank 0, N = 1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3     # vector fp32
FP64_PEAK_TFLOPS = 78.6      # vector fp64
# SURVEY.md §8 (d-roofline): algorithmic HBM bytes per point of the binned
# eps-neighbour kernels, and flop per ray-triangle test
DBSCAN_BYTES_PER_POINT = 341.0
MT_FLOP_PER_TEST = 45.0      # full Moller-Trumbore test
MT_FLOP_FRONT = 10.0         # executed per test by the parallel-ray kernel: tv (3), U (5), key (2)
                             # (p and det are per-triangle there; the general kernel executes 24)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--tris", type=int, default=500_000)
    ap.add_argument("--rays", type=int, default=10_000_000)
    ap.add_argument("--ray-steps", type=int, default=2)
    ap.add_argument("--no-rays", action="store_true")
    ap.add_argument("--no-knn", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--skel-iters", type=int, default=20,
                    help="contraction steps of the skeleton section (config 3: 20)")
    ap.add_argument("--no-skeleton", action="store_true")
    ap.add_argument("--no-ransac", action="store_true")
    ap.add_argument("--no-config5", action="store_true",
                    help="skip the whole-pipeline section (configs[4] at its stated sizes, ~15 s)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from pyqsm_amd import _lib, hip, synth
    from pyqsm_amd.parallel import NativeComm, ShardedSweep, shard_bounds
    dev = local_rank
    _lib.require_gpu(dev)

    # N > 1: one process per GPU. Every byte of the data path (mesh broadcast, result
    # all-gather) and the barrier / max-over-ranks of the timing go through the library's own
    # RCCL communicator (pyqsm_comm_*, multi.hip). torch.distributed is used for ONE thing: the
    # CPU (gloo) rendezvous that hands rank 0's 128-byte RCCL id to the other ranks.
    # PYQSM_BENCH_FORCE_DIST=1 exercises the same path with a single rank.
    comm = dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo")
        comm = NativeComm.from_torch(dist, dev)
    elif os.environ.get("PYQSM_BENCH_FORCE_DIST") == "1":
        comm = NativeComm(NativeComm.new_id(), 1, 0, dev)

    def barrier():
        hip.sync(dev)                 # the library stream carries all of this process's GPU work
        if comm is not None:
            comm.barrier()

    def max_over_ranks(x: float) -> float:
        return x if comm is None else comm.max_over_ranks(x)

    # ------------------------------------------------------------- DBSCAN (primary)
    n = args.points
    pts = synth.forest(n, seed=1000 * rank)          # every rank its own forest
    d_xyz = hip.DeviceBuffer.from_array(pts, dev)
    d_lab = hip.DeviceBuffer(n * 8, dev)
    d_core = hip.DeviceBuffer(n, dev)
    eps, min_pts = 0.1, 10
    for _ in range(args.warmup):
        hip.dbscan_dev(d_xyz.ptr, n, eps, min_pts, d_lab.ptr, d_core.ptr, dev)
    hip.prof_enable(True, dev)
    hip.prof_reset(dev)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        hip.dbscan_dev(d_xyz.ptr, n, eps, min_pts, d_lab.ptr, d_core.ptr, dev)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    kernels = {}
    for name in ("dbscan_bin", "dbscan_core", "dbscan_union", "dbscan_label", "dbscan_total",
                 "k_core_tiled", "k_hook_sub", "k_union_sub"):
        ms, cnt = hip.prof_get(name, dev)
        kernels[name] = {"avg_ms": ms / max(cnt, 1), "launches": cnt}
    hip.prof_enable(False, dev)
    labels = d_lab.download((n,), np.int64)
    n_clusters = int(labels.max() + 1)
    value = world * n * args.steps / elapsed / 1e6
    # The neighbourhood kernels are timed individually (HIP events on the library stream around the
    # single launch); the phases above contain several small kernels each.
    dom = max(("k_core_tiled", "k_hook_sub", "k_union_sub"), key=lambda k: kernels[k]["avg_ms"])
    dom_ms = kernels[dom]["avg_ms"]
    step_ms = elapsed / args.steps * 1e3
    # HBM bytes by the counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same section,
    # profiles/r03_traffic.json; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950)
    tj = {}
    tpath = os.path.join(ROOT, "profiles", "r03_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            tj = json.load(f)
    dbt = tj.get("dbscan", {}) if tj.get("points") == n else {}
    step_bytes = dbt.get("hbm_bytes_per_unit")
    kern_traffic = {q: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "launches_per_step": v["launches_per_unit"]}
                    for q, v in dbt.get("kernels", {}).items()}
    dom_bytes = kern_traffic.get(dom, {}).get("hbm_bytes_per_launch")
    # executed fp64 pair tests of the core pass (profiling level 2 adds a counter; one extra,
    # untimed call) priced against the FP64 vector peak: SURVEY.md §8d names FP64 VALU as the
    # binding roof of the neighbourhood kernels (8 flop + 1 compare per candidate pair)
    hip.prof_enable(2, dev)
    hip.prof_reset(dev)
    hip.dbscan_dev(d_xyz.ptr, n, eps, min_pts, d_lab.ptr, d_core.ptr, dev)
    _, lane_tests = hip.prof_get("core_pair_tests", dev)
    _, f32_records = hip.prof_get("dbscan_f32_records", dev)
    hip.prof_enable(False, dev)
    core_ms = kernels["k_core_tiled"]["avg_ms"]
    core_tflops = 9.0 * lane_tests / (core_ms * 1e-3) / 1e12 if core_ms > 0 else 0.0
    achieved = step_bytes / (step_ms * 1e-3) / 1e9 if step_bytes else None
    alg_step = DBSCAN_BYTES_PER_POINT * n / (step_ms * 1e-3) / 1e9
    roofline = {
        "kernel": "dbscan step (every kernel of one clustering)", "bound": "hbm",
        "achieved": achieved if achieved is not None else alg_step, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (achieved if achieved is not None else alg_step) / HBM_PEAK_GBS,
        "traffic": step_bytes, "avg_launch_ms": step_ms,
        "step_hbm_frac": alg_step / HBM_PEAK_GBS,
        "algorithmic_bytes_per_step": DBSCAN_BYTES_PER_POINT * n,
        "dominant_kernel": {"kernel": dom, "avg_launch_ms": dom_ms, "traffic": dom_bytes,
                            "traffic_GBs": dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_bytes else None,
                            "frac": dom_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if dom_bytes else None},
        "per_kernel": {q: dict(v, avg_launch_ms=kernels[q]["avg_ms"],
                               GBs=v["hbm_bytes_per_launch"] / (kernels[q]["avg_ms"] * 1e-3) / 1e9)
                       for q, v in kern_traffic.items() if q in kernels and kernels[q]["avg_ms"] > 0},
        "fp64_valu": {"kernel": "k_core_tiled", "lane_tests_per_launch": lane_tests,
                      "flop_per_test": 9, "achieved": core_tflops, "peak": FP64_PEAK_TFLOPS,
                      "unit": "TFLOP/s", "frac": core_tflops / FP64_PEAK_TFLOPS,
                      "avg_launch_ms": core_ms,
                      "note": "lane-tests EXECUTED (64 lanes x candidates staged per wave, "
                              "counted on the device at profiling level 2), early exits "
                              "included; SURVEY's stencil holds ~855 candidates per point"},
        "fp32_records": bool(f32_records),
        "note": "achieved = HBM bytes of ONE WHOLE STEP by the counters (sum over its ~20 kernels of 2 x FETCH_SIZE + "
                "WRITE_SIZE per launch x launches per step, profiles/r03_traffic.json) / the step's time, live; frac = "
                "that / 8 TB/s. step_hbm_frac prices the same step at SURVEY.md §8d's algorithmic 341 B/point instead. "
                "The neighbourhood kernels are FP64-VALU / gather-latency bound, not HBM bound: fp64_valu prices the "
                "core pass against the vector peak; dominant_kernel gives the slowest kernel's own counter bytes over "
                "its own HIP-event time. " + ("" if step_bytes else "NO PMC summary for this size: achieved falls "
                                              "back to the algorithmic figure and traffic is null.")}

    # ---- binning class (passes x N x bytes; state the pass count)
    ext = pts.max(0) - pts.min(0)
    cell = eps * (1.0 + 1.0 / 1048576.0)
    ncell = float(np.prod(np.floor(ext / cell) + 3.0))
    rec_b = 28.0 if f32_records else 36.0           # sorted record: fp32 x,y,z,flag or three fp64 + order, cell, sub-cell
    bin_alg = n * (24.0 + (24.0 + 4.0) + (24.0 + 4.0 + 32.0) + (32.0 + rec_b)) + 4.0 * ncell + 128.0 * (n / 5.0)
    bin_ms = kernels["dbscan_bin"]["avg_ms"]
    bin_names = ("k_bbox", "k_bbox_fold", "k_bk_hist", "k_bk_scan", "k_bk_scatter", "k_bk_sort", "k_order_big")
    bin_bytes = sum(kern_traffic[q]["hbm_bytes_per_launch"] * kern_traffic[q]["launches_per_step"]
                    for q in bin_names if q in kern_traffic) or None
    binning = {"roofline": {
        "kernel": "dbscan_bin: k_bbox, k_bk_hist, k_bk_scatter, k_bk_sort (+ 3 small kernels)", "bound": "hbm",
        "passes_over_the_points": 4, "avg_launch_ms": bin_ms,
        "algorithmic_bytes": bin_alg, "achieved": bin_alg / (bin_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
        "unit": "GB/s", "frac": bin_alg / (bin_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": bin_bytes,
        "traffic_over_algorithmic": bin_bytes / bin_alg if bin_bytes else None,
        "note": "four passes over the points: bounding box (24 B read), bucket histogram (24 read + 4 written), "
                "bucket scatter (28 read + 32 written), in-LDS sort of every bucket (32 read + the sorted record "
                f"written: {rec_b:.0f} B) = {24 + 28 + 60 + 32 + rec_b:.0f} B per point, plus the dense cell directory "
                f"written once (4 B x {ncell:.3g} cells) and 128 B of sub-cell records per occupied cell (~n / 5 cells); the phase "
                "also holds one host round trip (the bounding box sizes the grid)"}}
    out = {
        "metric": "Mpoints/s DBSCAN (1M-pt synthetic forest, eps=0.1, min_neighbors=10)",
        "value": value, "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": step_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{n}-point synthetic tree cloud (configs[1]), DBSCAN eps=0.1 "
                               f"min_neighbors=10, {n_clusters} clusters; replicas at N>1",
                   "points_per_gpu": n, "parallelism": f"replicas x{world}"},
        "roofline": roofline, "binning": binning, "kernels": kernels,
    }

    # ------------------------------------------------------------- kNN (same cloud)
    if not args.no_knn:
        k = 20
        d_idx = hip.DeviceBuffer(n * k * 4, dev)
        d_d2 = hip.DeviceBuffer(n * k * 8, dev)
        hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr, dev)
        hip.prof_enable(True, dev)
        hip.prof_reset(dev)
        barrier()
        reps = max(1, min(args.steps, 5))
        t0 = time.perf_counter()
        for _ in range(reps):
            hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr, dev)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        kk = {name: hip.prof_get(name, dev) for name in ("knn_bin", "knn_search", "knn_retry")}
        hip.prof_enable(False, dev)
        search_ms = kk["knn_search"][0] / max(kk["knn_search"][1], 1)
        knn_t = tj.get("knn", {}) if tj.get("points") == n else {}
        knn_bytes = (DBSCAN_BYTES_PER_POINT + 12.0 * k) * n      # stencil re-reads + idx/d2 rows out
        out["knn"] = {"k": k, "points_per_gpu": n, "steps": reps, "ms_per_step": dt / reps * 1e3,
                      "value": world * n / (dt / reps) / 1e6, "unit": "Mpoints/s", "dtype": "f64",
                      "phases_ms": {name: v[0] / reps for name, v in kk.items()},
                      "roofline": {"kernel": "k_knn_reg<20> (knn_search)", "bound": "fp64-valu issue / gather latency",
                                   "achieved": knn_bytes / (search_ms * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": knn_bytes / (search_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "traffic": knn_t.get("kernels", {}).get("k_knn_reg", {}).get("hbm_bytes_per_launch"),
                                   "avg_launch_ms": search_ms,
                                   "valu_busy": knn_t.get("valu_busy"),
                                   "wait_any": knn_t.get("wait_any"), "wait_inst_any": knn_t.get("wait_inst_any"),
                                   "note": "achieved / frac price the search kernel at 341 B/point (SURVEY.md §8d, binned "
                                           "neighbour class) + 12 B x k of result rows against HBM, which is NOT its "
                                           "roof: valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (launch time x "
                                           "2.4 GHz) says how much of the launch the vector pipes were issuing "
                                           "(profiles/r03_knn_sq_counters.csv); wait_any / wait_inst_any = share of "
                                           "wave cycles parked on memory / stalled at issue; traffic = counter bytes "
                                           "per launch (profiles/r03_traffic.json)"}}
        if world == 1:
            # the same cloud with twenty stray returns 3-10 cloud sizes away (what terrestrial scans carry):
            # grids over the box without its tails + whole-cloud search of the strays
            srng = np.random.default_rng(3)
            stray = pts.copy()
            ext_c = pts.max(0) - pts.min(0)
            stray[srng.choice(n, 20, replace=False)] = (pts.mean(0) + srng.choice([-1.0, 1.0], (20, 3))
                                                        * srng.uniform(3, 10, (20, 3)) * ext_c)
            d_stray = hip.DeviceBuffer.from_array(stray, dev)
            hip.knn_dev(d_stray.ptr, n, k, True, d_idx.ptr, d_d2.ptr, dev)
            t0 = time.perf_counter()
            for _ in range(3):
                hip.knn_dev(d_stray.ptr, n, k, True, d_idx.ptr, d_d2.ptr, dev)
            out["knn"]["with_20_stray_points_ms"] = (time.perf_counter() - t0) / 3 * 1e3
            d_stray.free()
            hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr, dev)   # the baseline below reads these
        if rank == 0 and world == 1 and not args.no_cpu:
            from scipy.spatial import cKDTree
            t0 = time.perf_counter()
            tree = cKDTree(pts)
            dd, ii = tree.query(pts, k=k + 1, workers=-1)        # k + 1: the point itself comes first
            c = time.perf_counter() - t0
            got = d_idx.download((n, k), np.int32)
            out["knn"]["cpu_baseline"] = {
                "value": n / c / 1e6, "unit": "Mpoints/s", "cores": os.cpu_count(), "kind": "reference",
                "sample": f"scipy.spatial.cKDTree(pts).query(pts, k={k + 1}, workers=-1) on the same {n} "
                          "points, build included (reconstruction.py:238-240's call; SURVEY.md §8d); "
                          f"same neighbour sets as GPU: {bool(np.array_equal(np.sort(got, 1), np.sort(ii[:, 1:], 1)))}"}
        d_idx.free()
        d_d2.free()

    # ------------------------------------------------------------- ray sweep
    if not args.no_rays:
        T, R = args.tris, args.rays
        verts, tris = synth.canopy_mesh(T)           # same seed on every rank (rays need its bbox)
        b, e = shard_bounds(R, world, rank)
        rays = synth.sun_rays(verts, R)[b:e]
        r_loc = e - b
        if comm is not None:
            # rank 0 expands the mesh, the records travel over RCCL, results are all-gathered
            sharded = ShardedSweep(comm, verts, tris, rays, R)
            sweep = sharded.run
        else:
            mesh = hip.DeviceMesh(verts, tris, dev)
            d_rays = hip.DeviceBuffer.from_array(rays, dev)
            d_t = hip.DeviceBuffer(r_loc * 4, dev)
            d_p = hip.DeviceBuffer(r_loc * 4, dev)

            def sweep():
                hip.cast_rays_dev(mesh, d_rays.ptr, r_loc, d_t.ptr, d_p.ptr)

        def timed(steps, prof_name):
            sweep()                                   # warm-up
            hip.prof_enable(True, dev)
            hip.prof_reset(dev)
            barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                sweep()
            barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            ms, cnt = hip.prof_get(prof_name, dev)
            hip.prof_enable(False, dev)
            return dt / steps, ms / max(cnt, 1)

        def hits():
            if comm is not None:
                return sharded.results()[0]
            return d_t.download((r_loc,), np.float32)

        # (1) the brute-force sweep BASELINE.json's Mray-tri/s is defined on: every ray
        #     against every triangle, no culling
        os.environ["PYQSM_RAY_CULL"] = "0"
        per, kern_ms = timed(args.ray_steps, "cast_rays")
        t_all = hits()
        # (2) the default path for one-direction batches: same kernel behind two levels
        #     of conservative rectangle culling (results bit-identical)
        os.environ["PYQSM_RAY_CULL"] = "1"
        per_c, kern_c_ms = timed(max(args.ray_steps, 5), "cast_rays_culled")
        same = bool(np.array_equal(t_all, hits()))
        tests = float(R) * float(T)
        flops = MT_FLOP_FRONT * float(r_loc) * float(T) / (kern_ms * 1e-3) / 1e12
        flops_equiv = MT_FLOP_PER_TEST * float(r_loc) * float(T) / (kern_ms * 1e-3) / 1e12
        out["ray_sweep"] = {
            "metric": "Mray-tri/s", "value": tests / per / 1e6, "unit": "Mray-tri/s",
            "rays": R, "tris": T, "steps": args.ray_steps, "ms_per_step": per * 1e3,
            "scaling": "strong", "hit_fraction": float(np.isfinite(t_all).mean()),
            "kernel_avg_ms": kern_ms, "dtype": "f32",
            "roofline": {"bound": "fp32-valu", "achieved": flops, "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": flops / FP32_PEAK_TFLOPS,
                         "equivalent_45flop_tflops": flops_equiv,
                         "note": "executed flops: the batch has one direction, so every test runs "
                                 "the 10-flop front half (tv, U, cull key; p and det are hoisted "
                                 "per triangle); the 21-flop back half runs only for waves in "
                                 "which some lane can still hit and is not counted. "
                                 "equivalent_45flop_tflops prices every test at SURVEY.md §8d's "
                                 "45 flop and is NOT a utilisation figure"},
            "algorithmic_hbm_bytes": 24.0 * R + 48.0 * T + 8.0 * R,
            "culled": {"ms_per_step": per_c * 1e3, "kernel_avg_ms": kern_c_ms,
                       "Mrays_per_s": R / per_c / 1e6,
                       "equivalent_Mray_tri_per_s": tests / per_c / 1e6,
                       "identical_to_brute_force": same,
                       "note": "default path for parallel rays: triangles sorted on the plane "
                               "normal to the direction, 16-triangle clusters and 32-cluster "
                               "groups with bounding rectangles, one scalar rectangle test per "
                               "wave; includes the per-call sort. Not a brute-force rate."},
        }
        if rank == 0 and world == 1 and not args.no_cpu:
            import oracle
            sample = 2000
            sub = synth.sun_rays(verts, R)[:: max(1, R // sample)][:sample]
            t0 = time.perf_counter()
            oracle.cast_rays(verts, tris, sub)
            c = time.perf_counter() - t0
            out["ray_sweep"]["cpu_baseline"] = {
                "value": len(sub) * float(T) / c / 1e6, "unit": "Mray-tri/s",
                "cores": oracle.num_threads(), "kind": "port",
                "sample": f"{len(sub)} of the {R} rays x {T} triangles, brute-force "
                          "Moller-Trumbore in C/OpenMP (oracle/pyqsm_oracle.c); NOT Embree "
                          "(Open3D is not installable here)"}

    # ------------------------------------------------------------- skeleton (config 3)
    if not args.no_skeleton and world == 1:
        from pyqsm_amd.geometry import skeletonize as skel
        names = ("lap_knn", "lap_fans", "lap_assemble", "lbc_amg_iter", "lbc_amg_build",
                 "lbc_inner_iter", "lbc_outer_iter")
        rows = {}
        for cfac in (3, 7):       # the TOML's active value and the one BASELINE.json quotes
            hip.prof_enable(True, dev)
            hip.prof_reset(dev)
            t0 = time.perf_counter()
            got, total_shift, steps_done = skel.extract_skeleton(
                pts, max_iter=args.skel_iters, termination_ratio=0.0, contraction_factor=cfac)
            wall = time.perf_counter() - t0
            prof = {kname: hip.prof_get(kname, dev) for kname in names}
            hip.prof_enable(False, dev)
            log = got.solve_log
            if cfac == 3:
                rows_total3 = total_shift
            rows[f"init_contraction_{cfac}"] = {
                "contractions": len(steps_done), "wall_s": wall,
                "s_per_contraction": wall / max(len(steps_done), 1),
                "laplacian_builds": prof["lap_knn"][1],
                "laplacian_ms_per_build": sum(prof[q][0] for q in ("lap_knn", "lap_fans", "lap_assemble"))
                / max(prof["lap_knn"][1], 1),
                "solve_ms_total": prof["lbc_outer_iter"][0] + prof["lbc_amg_build"][0],
                "outer_cg_steps": prof["lbc_outer_iter"][1],
                "multigrid_cg_iterations": prof["lbc_amg_iter"][1],
                "ms_per_multigrid_cg_iteration": prof["lbc_amg_iter"][0] / max(prof["lbc_amg_iter"][1], 1),
                "iterations_per_solve": [int(q["iters"]) for q in log],
                "solves_not_converged": sum(1 for q in log if not q["ok"]),
                "max_true_residual": max((max(q["resid"]) for q in log), default=0.0),
                "mean_shift_m": float(np.linalg.norm(total_shift, axis=1).mean())}
        # (the rows above run the default engine: pyqsm_extract_skeleton, everything resident in HBM)
        # the same loop as Python over the C-ABI calls, NumPy / SciPy objects between the steps
        t0 = time.perf_counter()
        gotn, totaln, stepsn = skel.extract_skeleton(pts, max_iter=args.skel_iters, termination_ratio=0.0,
                                                     contraction_factor=3, engine="python")
        walln = time.perf_counter() - t0
        rows["init_contraction_3_python_loop"] = {
            "contractions": len(stepsn), "wall_s": walln, "s_per_contraction": walln / max(len(stepsn), 1),
            "solves_not_converged": sum(1 for q in gotn.solve_log if not q["ok"]),
            "mean_shift_m": float(np.linalg.norm(totaln, axis=1).mean()),
            "same_bits_as_default_engine": bool(np.array_equal(totaln, rows_total3)),
            "note": "engine='python': the loop of skeletonize.py as Python, ~200 MB over PCIe per step"}
        # the three level-0 sparse passes of a multigrid-CG iteration, timed one launch at a time
        # (profiling level 2) over all the contractions of the row above
        L0, M0 = skel.point_cloud_laplacian(pts, mollify_factor=1e-6, n_neighbors=20, device=dev)
        pass_bytes = 8.0 * L0.nnz + 36.0 * n
        hip.prof_enable(2, dev)
        hip.prof_reset(dev)
        skel.extract_skeleton(pts, max_iter=args.skel_iters, termination_ratio=0.0, contraction_factor=3,
                              engine="native")
        passes = {}
        for kname in ("k_bspmv_f", "k_down_l0", "k_up_l0"):
            ms, cnt = hip.prof_get(kname, dev)
            avg = ms / max(cnt, 1)
            passes[kname] = {"avg_launch_ms": avg, "launches": cnt,
                             "achieved": pass_bytes / (avg * 1e-3) / 1e9 if avg > 0 else 0.0,
                             "frac": pass_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0}
        hip.prof_enable(False, dev)
        domk = max(passes, key=lambda q: passes[q]["avg_launch_ms"])
        sol_t = (tj.get("solve", {}) if tj.get("points") == n else {}).get("kernels", {})
        pmc_name = {"k_bspmv_f": "k_bspmv_f", "k_down_l0": "k_down", "k_up_l0": "k_up_ap"}
        for kname, v in passes.items():
            v["traffic"] = sol_t.get(pmc_name[kname], {}).get("hbm_bytes_per_launch")
            v["traffic_over_algorithmic"] = v["traffic"] / pass_bytes if v["traffic"] else None
        # the Laplacian build (SURVEY.md §8d "mixed" row: kNN part FP64 VALU, assembly HBM):
        # N*(24 + 4k) bytes in, 12*nnz + 12*N out, against what the counters say it moves
        lap_t = tj.get("laplacian", {}) if tj.get("points") == n else {}
        row3 = rows["init_contraction_3"]
        lap_ms = row3["laplacian_ms_per_build"]
        lap_alg = n * (24.0 + 4.0 * 20) + 12.0 * L0.nnz + 12.0 * n
        lap_bytes = lap_t.get("hbm_bytes_per_unit")
        lap_kernels = lap_t.get("kernel_ms_per_unit", {})
        top = sorted(lap_kernels.items(), key=lambda kv: -kv[1])[:10]
        lap_roof = {
            "kernel": "one point-cloud Laplacian build (lap_knn + lap_fans + lap_assemble)", "bound": "mixed (SURVEY.md §8d)",
            "avg_launch_ms": lap_ms, "algorithmic_bytes": lap_alg,
            "achieved": lap_alg / (lap_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": lap_alg / (lap_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": lap_bytes, "traffic_GBs": lap_bytes / (lap_ms * 1e-3) / 1e9 if lap_bytes else None,
            "traffic_frac": lap_bytes / (lap_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if lap_bytes else None,
            "traffic_over_algorithmic": lap_bytes / lap_alg if lap_bytes else None,
            "kernel_ms_per_build_top10": {q: round(v, 3) for q, v in top},
            "note": "algorithmic bytes = N*(24 + 4k) in + 12*nnz + 12*N out (SURVEY.md §8d); traffic = counter bytes "
                    "of one build on the raw cloud (profiles/r03_traffic.json: 2 x FETCH_SIZE + WRITE_SIZE summed over "
                    "its ~60 kernels); the build is not a streaming pass: the intrinsic Delaunay flips (~50 scattered "
                    "accesses per flip, rounds of conflict-free flips) and the per-point fan construction (a 3x3 Jacobi "
                    "eigen-decomposition and k^2 empty-circle tests per point, FP64 VALU) bound it, so its HBM fraction "
                    "says how far from a pure assembly it is, not how well it streams"}
        out["skeleton"] = {
            "workload": f"{n}-point forest, extract_skeleton with max_iter={args.skel_iters}, "
                        "termination_ratio=0 (configs[2]), TOML weights; wall of the call incl. PCIe in and out",
            "rows": rows, "dtype": "f64 (multigrid preconditioner in f32)",
            "roofline": {"kernel": domk, "bound": "hbm", "achieved": passes[domk]["achieved"],
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": passes[domk]["frac"],
                         "traffic": passes[domk]["traffic"], "passes": passes, "nnz": int(L0.nnz),
                         "note": "level-0 sparse passes (fp32 values, 3 right-hand sides as float4 rows) "
                                 "priced at 8*nnz + 36*N algorithmic bytes each; HIP events around single "
                                 "launches (profiling level 2) over all contractions (the first two alone: "
                                 "37-38 us per pass, before the cloud has collapsed onto its skeleton); traffic = "
                                 "counter bytes per level-0 launch in the first contraction solve "
                                 "(profiles/r03_traffic.json)"},
            "laplacian_build": {"roofline": lap_roof}}
        if rank == 0 and not args.no_cpu:
            # the reference's own solve (three SciPy spsolve calls, skeletonize.py:167-173) on
            # bounded samples: the first contraction of a 30 k- and a 100 k-point forest
            import oracle
            base = {}
            for ns in (30_000, 100_000):
                sub = synth.forest(ns, seed=0)
                L, M = skel.point_cloud_laplacian(sub, mollify_factor=1e-6, n_neighbors=20, device=dev)
                wl = np.full(ns, 3 * 1e3 * np.sqrt(np.mean(M.diagonal())))
                wh = np.full(ns, 3.0)
                t0 = time.perf_counter()
                ref = oracle.least_squares_sparse(sub, L, wl, wh)
                c = time.perf_counter() - t0
                skel.least_squares_sparse(sub, L, wl, wh, device=dev)     # warm
                t0 = time.perf_counter()
                got = skel.least_squares_sparse(sub, L, wl, wh, device=dev)
                g = time.perf_counter() - t0
                base[ns] = {"cpu_s": c, "gpu_s": g,
                            "max_rel_diff": float(np.abs(got - ref).max() / np.abs(ref).max())}
            out["skeleton"]["cpu_baseline"] = {
                "value": 100_000 / base[100_000]["cpu_s"] / 1e6, "unit": "Mpoints/s per contraction solve",
                "cores": 1, "kind": "reference",
                "sample": "first contraction solve of a 30 k- and a 100 k-point forest by the reference's "
                          "three scipy spsolve(COLAMD) calls (skeletonize.py:167-173) on the GPU-built "
                          "Laplacian; value = the 100 k case; gpu_s = same system on the GPU incl. PCIe",
                "points": base}

    # ------------------------------------------------------------- RANSAC circle fit
    if not args.no_ransac and world == 1:
        nr, H = 50_000, 1000
        ring = synth.ring_cluster(nr, seed=3)
        ring[:, 2] = 0.0                                   # fit.py:274-276: circle fit in z = 0
        rng = np.random.default_rng(2)
        triples = np.stack([rng.choice(nr, 3, replace=False) for _ in range(H)]).astype(np.int64)
        # the section before this one ends with seconds of host-only work (the SciPy baseline):
        # the first calls after such a gap run on an idling GPU (0.35 vs 3-4 ms per fit measured),
        # so warm up until the clocks are back
        for _ in range(30):
            hip.ransac(ring, triples, "circle", 0.04, dev)
        reps = 20
        hip.prof_enable(True, dev)
        hip.prof_reset(dev)
        t0 = time.perf_counter()
        for _ in range(reps):
            res = hip.ransac(ring, triples, "circle", 0.04, dev)
        dt = (time.perf_counter() - t0) / reps
        cnt_ms, cnt_n = hip.prof_get("ransac_count", dev)
        hip.prof_enable(False, dev)
        cnt_ms /= max(cnt_n, 1)
        rtf = 30.0 * H * nr / (cnt_ms * 1e-3) / 1e12 if cnt_ms > 0 else 0.0
        out["ransac"] = {"roofline": {"kernel": "k_count<0> (ransac_count)", "bound": "fp64-valu",
                                      "achieved": rtf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "frac": rtf / FP64_PEAK_TFLOPS, "traffic": None,
                                      "avg_launch_ms": cnt_ms,
                                      "note": "30 flop per point x hypothesis test (SURVEY.md §8d, circle) "
                                              "over the inlier-count kernel's HIP-event time"},
                         "workload": f"{H} circle hypotheses x {nr} points, threshold 0.04 "
                                     "(qsm_generation.py:155), host buffers in, inliers out",
                         "ms_per_fit": dt * 1e3, "value": H * nr / dt / 1e6,
                         "unit": "Mpoint-hypothesis tests/s", "inliers": int(len(res[3])),
                         "dtype": "f64"}
        # config 5's shape: the ten 0.5 m z-slices of one tree, together (pyqsm_ransac_batch) and one by one
        tree = synth.forest(50_000, seed=0)
        slices = []
        for z0 in np.arange(0.5, 5.5, 0.5):
            sl = tree[(tree[:, 2] >= z0) & (tree[:, 2] < z0 + 0.5)].copy()
            sl[:, 2] = 0.0
            if len(sl) >= 50:
                slices.append(sl)
        if slices:
            seg = np.concatenate([[0], np.cumsum([len(q) for q in slices])]).astype(np.int64)
            tri = np.stack([np.stack([rng.choice(len(q), 3, replace=False) for _ in range(H)]) for q in slices])
            stacked = np.concatenate(slices)
            for _ in range(5):
                batch = hip.ransac_batch(stacked, seg, tri, "circle", 0.04, dev)
            t0 = time.perf_counter()
            for _ in range(reps):
                batch = hip.ransac_batch(stacked, seg, tri, "circle", 0.04, dev)
            tb = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps):
                single = [hip.ransac(q, t, "circle", 0.04, dev) for q, t in zip(slices, tri)]
            ts = (time.perf_counter() - t0) / reps
            same = all(np.array_equal(a[3], b) and a[4] == c for a, b, c in zip(single, batch[3], batch[4]))
            out["ransac"]["slices_of_a_tree"] = {
                "sets": len(slices), "points": int(seg[-1]), "hypotheses_per_set": H,
                "ms_one_batch_call": tb * 1e3, "ms_one_call_per_slice": ts * 1e3,
                "same_inliers_and_winner": bool(same)}
        if rank == 0 and not args.no_cpu:
            import oracle
            hs = 100
            t0 = time.perf_counter()
            ref = oracle.ransac_fit(ring, triples[:hs], "circle", 0.04)
            c = time.perf_counter() - t0
            sub = hip.ransac(ring, triples[:hs], "circle", 0.04, dev)
            out["ransac"]["cpu_baseline"] = {
                "value": hs * nr / c / 1e6, "unit": "Mpoint-hypothesis tests/s", "cores": 1,
                "kind": "port",
                "sample": f"first {hs} of the {H} hypotheses, NumPy restatement of pyransac3d's "
                          "Circle.fit loop (oracle.ransac_fit); same inliers as GPU: "
                          f"{bool(np.array_equal(np.asarray(ref[3]), sub[3]))}"}

    # ------------------------------------------------------------- farthest-point sampling
    if not args.no_skeleton and world == 1:
        # extract_topology keeps 10 % of the contracted cloud (skeletonize.py:127-132): 1 M -> 100 k here,
        # on the bench's own forest (the pruned rounds do not depend on the cloud being contracted)
        s_fps = n // 10
        hip.fps(pts, 2000, 0, dev)
        t0 = time.perf_counter()
        picked = hip.fps(pts, s_fps, 0, dev)
        t_tail = time.perf_counter() - t0
        os.environ["PYQSM_FPS_TAIL"] = "0"
        t0 = time.perf_counter()
        launched = hip.fps(pts, s_fps, 0, dev)
        t_pruned = time.perf_counter() - t0
        del os.environ["PYQSM_FPS_TAIL"]
        os.environ["PYQSM_FPS_PRUNE"] = "0"
        t0 = time.perf_counter()
        whole = hip.fps(pts, s_fps, 0, dev)
        t_whole = time.perf_counter() - t0
        del os.environ["PYQSM_FPS_PRUNE"]
        out["fps"] = {"points": n, "samples": s_fps, "s": t_tail, "s_launch_per_round": t_pruned,
                      "s_whole_cloud_rounds": t_whole, "us_per_sample": t_tail / s_fps * 1e6,
                      "same_indices": bool(np.array_equal(picked, whole) and np.array_equal(picked, launched)),
                      "dtype": "f64",
                      "note": "host buffers in and out; default = pruned rounds as launches until the samples' reach is "
                              "below three cells, then all remaining rounds inside one launch of one workgroup "
                              "(k_fps_tail); s_launch_per_round = PYQSM_FPS_TAIL=0; whole-cloud rounds move 32 B per "
                              "point and sample"}

    # ------------------------------------------------------------- config 5: the whole pipeline
    if not args.no_config5 and not args.no_skeleton and world == 1:
        # BASELINE.json configs[4] at its stated sizes on this one GPU: 5 M points -> DBSCAN -> 100 trees
        # through extract_skeleton_batch (20 contractions each) -> every stem slice through
        # fit_shape_RANSAC_batch (H = 1000) -> 5 sun angles x 10 M culled rays; the driver is
        # examples/config5_pipeline.py (tests/test_gpu_config5.py::test_whole_pipeline_at_stated_sizes
        # checks the same run stage by stage against the oracles)
        from examples import config5_pipeline as c5
        rec, _ = c5.run(scale=1.0, skeleton_iters=20, max_trees=100, engine="native")
        out["config5"] = {
            "workload": "5 M-point scan (100 trees) -> DBSCAN -> 100 x extract_skeleton (20 contractions, "
                        "block-diagonal batches of 1 M points, six host threads) -> RANSAC circles on every 0.5 m stem slice "
                        "(H = 1000) -> 50 M sun rays (5 angles x 10 M) x 500 k triangles; one MI355X, host "
                        "buffers in and out at every stage (PCIe included)",
            "stage_s": {k: rec[k] for k in ("dbscan_s", "skeleton_s", "ransac_s", "rays_s")},
            "total_s": rec["total_s"], "clusters": rec["clusters"], "trees_contracted": rec["skeleton_trees"],
            "slices_fitted": rec["ransac_fits"], "slices": rec["ransac_slices"],
            "median_stem_radius_m": rec["ransac_median_radius_m"],
            "intercepted_fraction": rec["intercepted_fraction"],
            "Mpoints_per_s_end_to_end": rec["points"] / rec["total_s"] / 1e6,
            "ransac_thread_seconds": rec["ransac_thread_seconds"]}

    # ------------------------------------------------------------- CPU baseline (primary)
    if rank == 0 and world == 1 and not args.no_cpu:
        from sklearn.cluster import DBSCAN
        t0 = time.perf_counter()
        sk = DBSCAN(eps=eps, min_samples=min_pts).fit(pts)
        c = time.perf_counter() - t0
        same = bool(np.array_equal(sk.labels_, labels))
        t0 = time.perf_counter()
        skp = DBSCAN(eps=eps, min_samples=min_pts, n_jobs=-1).fit(pts)
        cp = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": n / c / 1e6, "unit": "Mpoints/s", "cores": 1, "kind": "reference",
            "sample": f"sklearn.cluster.DBSCAN(eps=0.1, min_samples=10).fit on the same {n} "
                      "points (the call pyQSM makes at math_utils/fit.py:223, default n_jobs), one "
                      f"run, {os.cpu_count()} host cores visible; labels identical to GPU: {same}",
            "all_cores": {"value": n / cp / 1e6, "unit": "Mpoints/s", "cores": os.cpu_count(),
                          "sample": "the same call with n_jobs=-1 (SURVEY.md §8d); labels identical "
                                    f"to GPU: {bool(np.array_equal(skp.labels_, labels))}"}}

    if rank == 0:
        print(json.dumps(out))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
