#!/usr/bin/env python3
"""Workloads for the round-3 PMC passes of the contraction loop's two halves (run under rocprofv3):
    prof_skel.py lap   [n]   three point-cloud Laplacian builds of the n-point forest (default 1 M)
    prof_skel.py solve [n]   one contraction solve (c = 3, first contraction) on that Laplacian
    prof_skel.py loop  [n] [iters]  a contraction loop of `iters` steps (default 5): its last Laplacian build is
                                    one of a CONTRACTED cloud (the flip rounds are the long phase there)
Prints what ran, so that the summary can divide dispatch counts by builds / solves."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import _lib, hip, synth  # noqa: E402
from pyqsm_amd.geometry import skeletonize as sk  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "lap"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
_lib.require_gpu(0)
P = synth.forest(n)
if what == "lap":
    builds = 3
    for _ in range(builds):
        L, M = sk.point_cloud_laplacian(P, mollify_factor=1e-6, n_neighbors=20)
    print(json.dumps({"what": "lap", "points": n, "builds": builds, "nnz": int(L.nnz)}))
elif what == "loop":
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0)
    print(json.dumps({"what": "loop", "points": n, "contractions": iters}))
else:
    L, M = sk.point_cloud_laplacian(P, mollify_factor=1e-6, n_neighbors=20)
    wl = np.full(n, 3 * 1e3 * np.sqrt(np.mean(M.diagonal())))
    wh = np.full(n, 3.0)
    info = []
    sk.least_squares_sparse(P, L, wl, wh, info=info)
    print(json.dumps({"what": "solve", "points": n, "solves": 1, "builds": 1, "nnz": int(L.nnz),
                      "iters": int(info[0]["iters"]) if info else None}))
