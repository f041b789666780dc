"""dev aid: wall time of the two library calls of a contraction step against the GPU time their
profiling scopes record (what is left is PCIe staging, graph instantiation, arena growth, host syncs)"""
import sys, time
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]); iters = int(sys.argv[2])
P = synth.forest(n, seed=0)
sk.extract_skeleton(P, max_iter=1, termination_ratio=0.0)
hip.prof_enable(True); hip.prof_reset()
wall = {'lbc_solve': 0.0, 'pc_laplacian': 0.0}
def wrap(name):
    f = getattr(hip, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); wall[name] += time.perf_counter() - t; return r
    setattr(hip, name, g)
wrap('lbc_solve'); wrap('pc_laplacian')
t = time.perf_counter()
sk.extract_skeleton(P, max_iter=iters, termination_ratio=0.0)
tot = time.perf_counter() - t
g = lambda k: hip.prof_get(k)[0]
print('total wall %.0f ms; lbc_solve wall %.0f ms (outer scope %.0f + amg build %.0f); pc_laplacian wall %.0f ms (knn %.0f + fans %.0f + assemble %.0f); other host %.0f ms'
      % (tot * 1e3, wall['lbc_solve'] * 1e3, g('lbc_outer_iter'), g('lbc_amg_build'), wall['pc_laplacian'] * 1e3,
         g('lap_knn'), g('lap_fans'), g('lap_assemble'), (tot - wall['lbc_solve'] - wall['pc_laplacian']) * 1e3))
