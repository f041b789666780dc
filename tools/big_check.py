import sys, time, numpy as np
sys.path.insert(0, '.')
import oracle
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
P = synth.forest(5_000_000, seed=0)
t = time.perf_counter(); lab, core = hip.dbscan(P, 0.1, 10); g = time.perf_counter() - t
t = time.perf_counter(); lab0, core0 = oracle.dbscan(P, 0.1, 10); c = time.perf_counter() - t
print('5M dbscan gpu %.3f s oracle %.1f s equal labels %s core %s clusters %d' % (g, c, np.array_equal(lab, lab0), np.array_equal(core, core0), lab.max() + 1))
t = time.perf_counter(); idx, d2 = hip.knn(P, 20, True); g = time.perf_counter() - t
sub = np.random.default_rng(0).choice(len(P), 20000, replace=False)
from scipy.spatial import cKDTree
d0, i0 = cKDTree(P).query(P[sub], k=21)
print('5M knn gpu %.3f s; sample of 20000 vs cKDTree: d2 equal %s' % (g, np.array_equal(d2[sub], d0[:, 1:] ** 2) or np.allclose(np.sqrt(d2[sub]), d0[:, 1:], rtol=1e-12, atol=0)))
