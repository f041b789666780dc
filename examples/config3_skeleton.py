#!/usr/bin/env python3
"""BASELINE.json configs[2]: N-point synthetic forest, Laplacian-contraction
skeletonisation (extract_skeleton) on one MI355X, with a per-phase time split.

    python examples/config3_skeleton.py [--points 1000000] [--iters 20] [--contraction 3]

`--contraction 7` reproduces the value quoted in BASELINE.json (the reference's
active TOML value is 3)."""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import _lib, hip, synth  # noqa: E402
from pyqsm_amd.geometry import skeletonize as sk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--contraction", type=float, default=3)
    ap.add_argument("--attraction", type=float, default=3)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--engine", default="python", help="python (loop over the C-ABI calls) or native "
                                                         "(pyqsm_extract_skeleton: the whole loop in HBM)")
    ap.add_argument("--prof-level", type=int, default=1,
                    help="2: also time the level-0 sparse passes one by one (no graphs, slower)")
    args = ap.parse_args()
    if args.verbose:
        logging.basicConfig(level=logging.INFO)
    _lib.require_gpu(0)
    pts = synth.forest(args.points, seed=0)
    hip.prof_enable(args.prof_level)
    hip.prof_reset()
    t0 = time.perf_counter()
    got, total, steps = sk.extract_skeleton(pts, max_iter=args.iters, termination_ratio=0.0,
                                            contraction_factor=args.contraction,
                                            attraction_factor=args.attraction, engine=args.engine)
    wall = time.perf_counter() - t0
    prof = {k: hip.prof_get(k) for k in ("lap_knn", "lap_fans", "lap_assemble", "lbc_inner_iter",
                                         "lbc_amg_iter", "lbc_amg_build",
                                         "lbc_outer_iter", "lbc_cg_iter", "clamp")}
    out = {"config": f"{args.points}-point forest, {len(steps)} contraction steps, "
                     f"init_contraction={args.contraction}, engine={args.engine}",
           "wall_s": wall, "s_per_iteration": wall / max(len(steps), 1),
           "laplacian_ms": sum(prof[k][0] for k in ("lap_knn", "lap_fans", "lap_assemble")),
           "solve_multigrid_cg_iterations": prof["lbc_amg_iter"][1],
           "solve_multigrid_cg_ms": prof["lbc_amg_iter"][0],
           "solve_multigrid_setup_ms": prof["lbc_amg_build"][0],
           "solve_jacobi_cg_iterations": prof["lbc_inner_iter"][1],
           "solve_jacobi_cg_ms": prof["lbc_inner_iter"][0],
           "solve_outer_iterations": prof["lbc_outer_iter"][1],
           "solve_outer_ms_incl_inner": prof["lbc_outer_iter"][0],
           "mean_shift_m": float(np.linalg.norm(total, axis=1).mean())}
    if args.prof_level >= 2:
        for k in ("k_bspmv_f", "k_down_l0", "k_up_l0"):
            ms, cnt = hip.prof_get(k)
            out[k + "_us"] = 1e3 * ms / max(cnt, 1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
