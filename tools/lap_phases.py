"""dev aid: per-phase time of every Laplacian build along a contraction run"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]); iters = int(sys.argv[2])
P = synth.forest(n, seed=0)
keys = ('lap_knn', 'knn_bin', 'knn_search', 'knn_retry', 'lap_fans', 'lap_assemble', 'lap_flips')
orig = sk.point_cloud_laplacian
def timed(pts, *a, **k):
    hip.prof_enable(True); hip.prof_reset()
    t = time.perf_counter()
    out = orig(pts, *a, **k)
    w = time.perf_counter() - t
    print('laplacian wall %.0f ms' % (w * 1e3), {k: (round(hip.prof_get(k)[0], 1), hip.prof_get(k)[1]) for k in keys}, 'nnz/row %.2f' % (out[0].nnz / len(pts)), flush=True)
    return out
sk.point_cloud_laplacian = timed
sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0)
