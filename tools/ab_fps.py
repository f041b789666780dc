"""A/B of farthest-point sampling: late rounds in one launch (k_fps_tail) vs a launch per round."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
S = n // 10
P = synth.forest(n)
hip.fps(P, 2000, 0)
ref = None
for rnd in range(2):
    for name, env in (("tail", {}), ("launches", {"PYQSM_FPS_TAIL": "0"})):
        os.environ.pop("PYQSM_FPS_TAIL", None)
        os.environ.update(env)
        t = time.perf_counter()
        got = hip.fps(P, S, 0)
        dt = time.perf_counter() - t
        if ref is None:
            ref = got
        print(f"{name:9s} {S} of {n}: {dt:.3f} s ({dt / S * 1e6:.2f} us per sample)  same indices: {bool(np.array_equal(got, ref))}", flush=True)
# a contracted cloud (what extract_topology samples): points pulled towards the stems
Q = P.copy()
Q[:, :2] = np.round(Q[:, :2] / 10.0) * 10.0 + (Q[:, :2] - np.round(Q[:, :2] / 10.0) * 10.0) * 0.05
for name, env in (("tail", {}), ("launches", {"PYQSM_FPS_TAIL": "0"})):
    os.environ.pop("PYQSM_FPS_TAIL", None)
    os.environ.update(env)
    t = time.perf_counter()
    got = hip.fps(Q, S, 0)
    dt = time.perf_counter() - t
    if name == "tail":
        refq = got
    print(f"contracted {name:9s}: {dt:.3f} s  same indices: {bool(np.array_equal(got, refq))}", flush=True)
