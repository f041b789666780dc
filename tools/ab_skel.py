"""config 3 (1 M points, 20 contractions) wall time + a hash of the result: run once per environment
setting (the library reads some switches once per process), e.g.
    PYQSM_AMG_FUSED_TAIL=0 python tools/ab_skel.py ; python tools/ab_skel.py"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyqsm_amd import hip, synth, _lib
from pyqsm_amd.geometry import skeletonize as sk
_lib.require_gpu(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = float(sys.argv[2]) if len(sys.argv) > 2 else 3
P = synth.forest(n)
sk.extract_skeleton(P, max_iter=2, termination_ratio=0.0, contraction_factor=c)
for rep in range(2):
    hip.prof_enable(True); hip.prof_reset()
    t = time.perf_counter()
    got, total, steps = sk.extract_skeleton(P, max_iter=20, termination_ratio=0.0, contraction_factor=c)
    dt = time.perf_counter() - t
    it = hip.prof_get("lbc_amg_iter")
    lap = sum(hip.prof_get(k)[0] for k in ("lap_knn", "lap_fans", "lap_assemble"))
    flips = hip.prof_get("lap_flips")[0]
    hip.prof_enable(False)
    bad = sum(1 for q in got.solve_log if not q["ok"])
    fin = bool(np.isfinite(total).all() and np.isfinite(got.points).all())
    print(f"n={n} c={c}: {dt:.3f} s, not-ok solves {bad}, finite {fin}, multigrid-CG iterations {it[1]} in {it[0]:.0f} ms ({it[0]/max(it[1],1)*1e3:.1f} us each), "
          f"Laplacian builds {lap:.0f} ms (flips {flips:.0f}), sha {hashlib.sha1(total.tobytes()).hexdigest()[:12]} env "
          + " ".join(f"{k[6:]}={v}" for k, v in os.environ.items() if k.startswith("PYQSM_")), flush=True)
