import numpy as np, time, os, sys
from pyqsm_amd import synth, hip, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
Q = synth.forest(1_000_000, seed=0)
sk.extract_skeleton(Q, max_iter=2, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
hip.prof_enable(1); hip.prof_reset()
t = time.time()
out = sk.extract_skeleton(Q, max_iter=20, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
print("wall %.3f" % (time.time() - t))
names = ["lbc_solve_total", "lbc_amg_build", "lbc_riccati", "lbc_first_precond", "lbc_amg_iter", "lbc_outer_iter", "clamp", "spmv3", "lap_knn", "lap_fans", "lap_assemble", "lap_flips"]
for k in names:
    try:
        print(k, hip.prof_get(k))
    except Exception as e:
        print(k, "n/a", e)
