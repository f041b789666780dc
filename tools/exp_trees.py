#!/usr/bin/env python3
"""How do N independent 50 k-point trees contract fastest on ONE GPU: threads of one process
(the library keeps a stream per thread; ctypes releases the GIL only inside the library) or
several processes (no GIL shared)?   python tools/exp_trees.py --trees 24 --procs 1 --threads 8"""
import argparse
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(args):
    seeds, threads, iters = args
    from pyqsm_amd import _lib, synth
    from pyqsm_amd.geometry.skeletonize import extract_skeleton
    _lib.require_gpu(0)

    def one(seed):
        P = synth.forest(50_000, seed=seed)
        got, total, steps = extract_skeleton(P, max_iter=iters, termination_ratio=0.0)
        return float(np.linalg.norm(total, axis=1).mean())

    with ThreadPoolExecutor(max_workers=threads) as pool:
        return list(pool.map(one, seeds))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=24)
    ap.add_argument("--procs", type=int, default=1)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0, help="points per block-diagonal group (0: per-tree loop)")
    ap.add_argument("--engine", default="python")
    a = ap.parse_args()
    if a.batch:
        from pyqsm_amd import _lib, synth
        from pyqsm_amd.geometry.skeletonize import extract_skeleton_batch
        _lib.require_gpu(0)
        clouds = [synth.forest(50_000, seed=s) for s in range(a.trees)]
        t0 = time.perf_counter()
        res = extract_skeleton_batch(clouds, max_iter=a.iters, termination_ratio=0.0,
                                     group_points=a.batch, workers=a.threads, engine=a.engine)
        dt = time.perf_counter() - t0
        print(f"trees={a.trees} batch={a.batch} threads={a.threads} engine={a.engine}: {dt:.2f} s = {dt / a.trees:.3f} s per tree "
              "(synthetic input excluded)", flush=True)
        return
    seeds = list(range(a.trees))
    chunks = [seeds[i::a.procs] for i in range(a.procs)]
    t0 = time.perf_counter()
    if a.procs == 1:
        work((chunks[0], a.threads, a.iters))
    else:
        with ProcessPoolExecutor(max_workers=a.procs) as ex:
            list(ex.map(work, [(c, a.threads, a.iters) for c in chunks]))
    dt = time.perf_counter() - t0
    print(f"trees={a.trees} procs={a.procs} threads={a.threads}: {dt:.2f} s = {dt / a.trees:.3f} s per tree "
          "(incl. process start and synthetic input)", flush=True)


if __name__ == "__main__":
    main()
