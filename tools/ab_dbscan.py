"""A/B timing of the DBSCAN step (HBM-resident input) under environment switches:
    python tools/ab_dbscan.py [n] -- prints per-phase ms for each variant, interleaved."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
variants = [("bucketed+f32", {}), ("bucketed f64", {"PYQSM_COORD_F32": "0"}), ("atomic  f64", {"PYQSM_DBSCAN_BIN": "atomic"})]
if len(sys.argv) > 2 and sys.argv[2] == "union":   # grid of the two wave-per-sub-cells passes
    variants = [("16,32 per CU", {}), ("bound grid", {"PYQSM_UNION_BLOCKS_PER_CU": "0"}), ("8,8 per CU", {"PYQSM_UNION_BLOCKS_PER_CU": "8"}),
                ("16,64 per CU", {"PYQSM_UNION_BLOCKS_PER_CU": "16,64"})]
P = synth.forest(n)
d_xyz = hip.DeviceBuffer.from_array(P)
d_lab = hip.DeviceBuffer(n * 8); d_core = hip.DeviceBuffer(n)
ref = None
for rnd in range(3):
    for name, env in variants:
        for k in ("PYQSM_DBSCAN_BIN", "PYQSM_COORD_F32", "PYQSM_UNION_BLOCKS_PER_CU"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(3):
            hip.dbscan_dev(d_xyz.ptr, n, 0.1, 10, d_lab.ptr, d_core.ptr)
        hip.sync()
        hip.prof_enable(True); hip.prof_reset()
        t = time.perf_counter()
        for _ in range(20):
            hip.dbscan_dev(d_xyz.ptr, n, 0.1, 10, d_lab.ptr, d_core.ptr)
        hip.sync(); dt = (time.perf_counter() - t) / 20
        ph = {k: round(hip.prof_get(k)[0] / 20, 4) for k in ("dbscan_bin", "dbscan_core", "dbscan_union", "dbscan_label",
                                                             "k_core_tiled", "k_hook_sub", "k_union_sub")}
        hip.prof_enable(False)
        lab = d_lab.download((n,), np.int64)
        if ref is None:
            ref = lab
        print(f"{name:13s} step {dt*1e3:.3f} ms  {ph}  labels==first: {bool(np.array_equal(lab, ref))}", flush=True)
