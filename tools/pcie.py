import sys, time, numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
P = synth.forest(1_000_000)
hip.dbscan(P, 0.1, 10)
ts = []
for _ in range(5):
    t = time.perf_counter(); hip.dbscan(P, 0.1, 10); ts.append(time.perf_counter() - t)
print("host-pointer dbscan 1M: min %.2f ms median %.2f ms -> %.1f Mpts/s" % (min(ts)*1e3, sorted(ts)[2]*1e3, 1.0/sorted(ts)[2]))
