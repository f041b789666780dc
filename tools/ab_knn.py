"""A/B timing of kNN (k = 20, HBM-resident input) under environment switches, interleaved."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = 20
variants = [("bucketed+f32", {}), ("bucketed f64", {"PYQSM_COORD_F32": "0"}), ("atomic  f64", {"PYQSM_GRID_BIN": "atomic"})]
P = synth.forest(n)
d_xyz = hip.DeviceBuffer.from_array(P)
d_idx = hip.DeviceBuffer(n * k * 4); d_d2 = hip.DeviceBuffer(n * k * 8)
ref = None
for rnd in range(3):
    for name, env in variants:
        os.environ.pop("PYQSM_GRID_BIN", None)
        os.environ.pop("PYQSM_COORD_F32", None)
        os.environ.update(env)
        for _ in range(2):
            hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr)
        hip.sync()
        hip.prof_enable(True); hip.prof_reset()
        t = time.perf_counter()
        for _ in range(10):
            hip.knn_dev(d_xyz.ptr, n, k, True, d_idx.ptr, d_d2.ptr)
        hip.sync(); dt = (time.perf_counter() - t) / 10
        ph = {q: round(hip.prof_get(q)[0] / 10, 4) for q in ("knn_bin", "knn_search", "knn_retry")}
        hip.prof_enable(False)
        idx = d_idx.download((n, k), np.int32)
        if ref is None:
            ref = idx
        print(f"{name:13s} knn {dt*1e3:.3f} ms  {ph}  same neighbours: {bool(np.array_equal(idx, ref))}", flush=True)
