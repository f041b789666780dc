import sys, numpy as np, time
sys.path.insert(0, '.')
import oracle
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
P = synth.forest(2500, seed=9)
L0, M0 = oracle.point_cloud_laplacian(P, 20, 1e-6)
for rep, (m, k) in enumerate([(3000, 20), (20000, 20), (5000, 30), (800, 8), (4, 3), (2500, 20), (2500, 20)]):
    Q = synth.forest(m, seed=m) if m != 2500 else P
    if m == 4: Q = np.array([[0.0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.1]])
    Lg, Mg = sk.point_cloud_laplacian(Q, 1e-6, k)
    idx, d2 = hip.knn(Q, k, True)
    idx0, d20 = oracle.knn(Q, k, True)
    msg = f"n={m} k={k} knn idx equal {np.array_equal(idx, idx0)} d2 equal {np.array_equal(d2, d20)}"
    if m == 2500:
        msg += f" L equal struct {np.array_equal(Lg.indices, L0.indices)} maxabs {abs(Lg.data).max():.3e} vs {abs(L0.data).max():.3e}"
    print(msg, flush=True)
