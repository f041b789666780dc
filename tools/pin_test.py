"""dev aid: what extract_skeleton's wall holds besides the library call (bounds, page-locked results)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
P = synth.forest(n)
for rep in range(3):
    t = time.perf_counter(); lo, hi = sk.oriented_bounds(P, device=0); print("oriented_bounds(device=0) %.1f ms" % ((time.perf_counter() - t) * 1e3))
for rep in range(3):
    t = time.perf_counter()
    r = hip.extract_skeleton(P, lo, hi, 20, 1e-6, iters, 0.0, 3.0, 1.0, 2048.0, 1024.0, 1e-8, 5_000_000)
    t1 = time.perf_counter() - t
    del r
    t = time.perf_counter()
    r = sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0, contraction_factor=3.0)
    t2 = time.perf_counter() - t
    hip.prof_enable(True); hip.prof_reset()
    t = time.perf_counter()
    r = sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0, contraction_factor=3.0)
    t3 = time.perf_counter() - t
    names = ("lap_knn", "lap_fans", "lap_assemble", "lbc_solve_total", "lbc_outer_iter", "lbc_first_precond", "lbc_amg_build", "lbc_riccati", "lbc_amg_iter")
    print(f"library call {t1*1e3:.1f} ms, sk.extract_skeleton {t2*1e3:.1f} ms, with scopes {t3*1e3:.1f} ms",
          {k: round(hip.prof_get(k)[0], 1) for k in names})
    hip.prof_enable(False)
