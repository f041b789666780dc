"""A/B of the radius queries (fp32 records vs fp64 arrays for the sorted source points)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = synth.forest(n)
rng = np.random.default_rng(0)
Q = P[rng.choice(n, 100_000, replace=False)] + rng.normal(0, 0.01, (100_000, 3))
ref = {}
for rnd in range(2):
    for name, env in (("sorted+fp32", {}), ("sorted fp64", {"PYQSM_COORD_F32": "0"}),
                      ("caller order", {"PYQSM_RADIUS_SORT": "0"})):
        os.environ.pop("PYQSM_COORD_F32", None)
        os.environ.pop("PYQSM_RADIUS_SORT", None)
        os.environ.update(env)
        out = {}
        for what, fn in (("radius_mark", lambda: hip.radius_mark(P, Q, 0.1, 200)),
                         ("radius_knn", lambda: hip.radius_knn(P, Q[:20_000], 0.05, 64))):
            fn()
            hip.prof_enable(True); hip.prof_reset()
            t = time.perf_counter()
            got = fn()
            dt = time.perf_counter() - t
            kern = hip.prof_get(what)[0]
            hip.prof_enable(False)
            same = all(np.array_equal(a, b) for a, b in zip(got, ref.setdefault(what, got)))
            out[what] = f"{dt*1e3:.2f} ms wall, kernel {kern:.3f} ms, same: {same}"
        print(f"{name:13s} {out}", flush=True)
