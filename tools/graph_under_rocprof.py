#!/usr/bin/env python3
"""Does a hipGraph replay of the solver's CG bursts survive under rocprofv3?

    rocprofv3 --kernel-trace --stats -d gpurun_out/graphprof -- python3 tools/graph_under_rocprof.py [jacobi|amg]

`jacobi`: a per-point Laplacian weight sends pyqsm_lbc_solve down the Jacobi-PCG path, whose
24-iteration bursts are replayed as graphs by default. `amg`: uniform weight with
PYQSM_AMG_GRAPH=1 (set it in the environment), the multigrid-CG bursts as graphs.
Prints one line per stage so that a crash can be placed."""
import faulthandler
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable()
from pyqsm_amd import _lib, hip, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "jacobi"
_lib.require_gpu(0)
P = synth.forest(20_000, seed=3)
(indptr, indices, data), mass = hip.pc_laplacian(P, 20, 1e-6)
print("laplacian built", len(data), flush=True)
n = len(P)
wl = np.full(n, 3e3 * np.sqrt(mass.mean()))
if mode == "jacobi":
    wl = wl * (1.0 + 0.5 * np.sin(np.arange(n)))       # not constant along edges: Jacobi-PCG on A
wh = np.full(n, 3.0)
x, iters, resid, ok = hip.lbc_solve((indptr, indices, data), wl, wh, P, rtol=1e-6,
                                    max_it=2000 if mode == "jacobi" else 200000)
print(f"{mode}: solve returned, iters={iters} resid={resid.max():.3e} ok={ok} "
      f"graphs={'off' if os.environ.get('PYQSM_NO_GRAPH') else 'on'}", flush=True)
