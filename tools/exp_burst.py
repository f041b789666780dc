import numpy as np, time, os, sys
from pyqsm_amd import synth, hip, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]); cf = float(sys.argv[2])
Q = synth.forest(n, seed=0)
sk.extract_skeleton(Q, max_iter=2, termination_ratio=0.0, contraction_factor=cf, attraction_factor=3)
t = time.time()
out = sk.extract_skeleton(Q, max_iter=20, termination_ratio=0.0, contraction_factor=cf, attraction_factor=3)
dt = time.time() - t
log = out[0].solve_log
its = [r["iters"] for r in log]
print("first_burst", os.environ.get("PYQSM_AMG_FIRST_BURST"), "n", n, "c", cf, "wall %.3f" % dt, "its", sum(its), "notok", sum(not r["ok"] for r in log), flush=True)
