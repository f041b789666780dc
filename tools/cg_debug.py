import sys, numpy as np, time
sys.path.insert(0, '.')
import logging; logging.basicConfig(level=logging.INFO)
import oracle
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
P = synth.forest(n, seed=9)
hip.prof_enable(True)
t = time.time()
got, total, steps = sk.extract_skeleton(P, max_iter=iters, contraction_factor=3, attraction_factor=3, termination_ratio=0.0)
print('wall', time.time() - t, {k: (round(hip.prof_get(k)[0], 1), hip.prof_get(k)[1]) for k in ('lbc_inner_iter', 'lbc_amg_iter', 'lbc_amg_build', 'lbc_outer_iter', 'lap_knn', 'lap_fans', 'lap_assemble')})
if n <= 5000:
    lo, hi = sk.oriented_bounds(P)
    want, want_total, want_steps = oracle.extract_skeleton(P, lambda p: oracle.point_cloud_laplacian(p, 20, 1e-6), (lo, hi), max_iter=iters, termination_ratio=0.0, contraction_factor=3, attraction_factor=3)
    scale = np.abs(want).max()
    for s in range(len(steps)):
        print('step', s, 'rel diff', np.abs(steps[s] - want_steps[s]).max() / scale)
    print('final rel diff', np.abs(got.points - want).max() / scale)
