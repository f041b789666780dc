#!/usr/bin/env python3
"""How accurate is each contraction solve of a GPU loop, and how tight is the host-side
certificate |W_H^-1 r| / |W_H x| (tests/test_gpu_config3.py)?  For every step of
extract_skeleton on an n-point forest the system is also factorised by SuperLU and refined in
long double; prints one JSON record per step.

    python tools/solver_accuracy.py [--points 20000] [--iters 20] [--contraction 7]"""
import argparse
import json
import os
import sys
import time

import numpy as np
from scipy.sparse import diags
from scipy.sparse.linalg import splu

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import _lib, synth  # noqa: E402
from pyqsm_amd.geometry import skeletonize as sk  # noqa: E402


def refined(A, b, iters=4):
    lu = splu(A.tocsc(), permc_spec="COLAMD")
    x = np.column_stack([lu.solve(b[:, k]) for k in range(3)])
    x0 = x.copy()
    Al = A.tocsr()
    for _ in range(iters):
        r = np.empty_like(x)
        for k in range(3):
            prod = Al.data.astype(np.longdouble) * x[Al.indices, k].astype(np.longdouble)
            r[:, k] = (b[:, k].astype(np.longdouble) - np.add.reduceat(prod, Al.indptr[:-1])).astype(np.float64)
        x = x + np.column_stack([lu.solve(r[:, k]) for k in range(3)])
    return x, x0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--contraction", type=float, default=7)
    ap.add_argument("--rtol", type=float, default=sk.SOLVER_RTOL)
    args = ap.parse_args()
    _lib.require_gpu(0)
    P = synth.forest(args.points, seed=0)
    inner = sk.least_squares_sparse
    step = [0]

    def solve(pts, L, laplacian_weighting, positional_weighting, **kw):
        wl, wh = laplacian_weighting, positional_weighting
        kw["rtol"] = args.rtol
        t0 = time.perf_counter()
        x = inner(pts=pts, L=L, laplacian_weighting=wl, positional_weighting=wh, **kw)
        dt = time.perf_counter() - t0
        A = (diags(wl) @ (L.T @ L) @ diags(wl) + diags(wh * wh)).tocsr()
        b = (wh * wh)[:, None] * pts
        xt, xs = refined(A, b)
        r = b - A @ x
        scale = np.abs(xt).max()
        e = x - xt
        rec = {"step": step[0], "gpu_s": dt, "iters": kw["info"][-1]["iters"], "ok": kw["info"][-1]["ok"],
               "wl": float(wl[0]), "wh_min": float(wh.min()), "wh_max": float(wh.max()),
               "resid": float((np.linalg.norm(r, axis=0) / np.linalg.norm(b, axis=0)).max()),
               "cert": float((np.linalg.norm(r / wh[:, None], axis=0)
                              / np.linalg.norm(wh[:, None] * x, axis=0)).max()),
               "err_max_rel": float(np.abs(e).max() / scale),
               "err_wh_weighted": float((np.linalg.norm(wh[:, None] * e, axis=0)
                                         / np.linalg.norm(wh[:, None] * xt, axis=0)).max()),
               "superlu_err_max_rel": float(np.abs(xs - xt).max() / scale),
               "moved_max_rel": float(np.abs(xt - pts).max() / scale)}
        print(json.dumps(rec), flush=True)
        step[0] += 1
        return x

    sk.least_squares_sparse = solve
    sk.extract_skeleton(P, max_iter=args.iters, termination_ratio=0.0,
                        contraction_factor=args.contraction, attraction_factor=3)


if __name__ == "__main__":
    main()
