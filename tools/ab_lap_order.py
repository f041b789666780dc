"""dev aid: does the ORDER of the points matter to a Laplacian build? The same cloud in the caller's
order (tree by tree), shuffled, and sorted along a Morton curve of 5 cm cells:
    python tools/ab_lap_order.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = synth.forest(n)


def morton(P, cell=0.05):
    q = np.floor((P - P.min(0)) / cell).astype(np.uint64)
    key = np.zeros(len(P), np.uint64)
    for b in range(16):
        for a in range(3):
            key |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return np.argsort(key, kind="stable")


rng = np.random.default_rng(0)
orders = {"caller": np.arange(n), "shuffled": rng.permutation(n), "morton": morton(P)}
keys = ("lap_knn", "lap_fans", "lap_assemble", "lap_flips")
for rnd in range(2):
    for name, perm in orders.items():
        Q = np.ascontiguousarray(P[perm])
        hip.pc_laplacian(Q, k=20)
        hip.prof_enable(True); hip.prof_reset()
        t = time.perf_counter()
        (ip, ix, dv), mass = hip.pc_laplacian(Q, k=20)
        w = time.perf_counter() - t
        print(f"{name:9s} wall {w*1e3:.1f} ms", {k: round(hip.prof_get(k)[0], 2) for k in keys}, "nnz", len(ix), flush=True)
        hip.prof_enable(False)
