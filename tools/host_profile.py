"""dev aid: where does the HOST time of extract_skeleton go (cProfile, 1M points, 3 iterations)"""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
from pyqsm_amd import synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
P = synth.forest(int(sys.argv[1]), seed=0)
sk.extract_skeleton(P, max_iter=1, termination_ratio=0.0)
pr = cProfile.Profile(); pr.enable()
sk.extract_skeleton(P, max_iter=3, termination_ratio=0.0)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(22); print(s.getvalue()[:3500])
