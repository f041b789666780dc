import numpy as np, time, os, sys
from pyqsm_amd import synth, hip, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
Q = synth.forest(1_000_000, seed=0)
out = sk.extract_skeleton(Q, max_iter=6, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
C = np.asarray(out[0].points if isinstance(out, tuple) else out.points)
for name, P in (("raw", Q), ("contracted6", C)):
    hip.pc_laplacian(P, 20, 1e-6)
    hip.prof_enable(1); hip.prof_reset()
    t = time.time()
    for _ in range(3): L = hip.pc_laplacian(P, 20, 1e-6)
    dt = (time.time() - t) / 3
    print(name, "build wall %.1f ms" % (dt * 1e3), {k: hip.prof_get(k) for k in ("lap_knn", "lap_fans", "lap_assemble", "lap_flips")}, flush=True)
    hip.prof_enable(0)
