"""dev aid: one point-cloud Laplacian build at n points (for rocprofv3)"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1])
P = synth.forest(n, seed=0)
hip.pc_laplacian(P, 20, 1e-6)
t = time.perf_counter()
for _ in range(3): out = hip.pc_laplacian(P, 20, 1e-6)
print('laplacian wall ms', (time.perf_counter() - t) / 3 * 1e3, 'nnz', len(out[0][1]))
