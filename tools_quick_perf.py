"""Scratch timing of the two headline kernels with HBM-resident inputs."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from pyqsm_amd import hip, synth, _lib
_lib.require_gpu(0)
hip.prof_enable(True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = synth.forest(n)
d_xyz = hip.DeviceBuffer.from_array(P)
d_lab = hip.DeviceBuffer(n * 8); d_core = hip.DeviceBuffer(n)
for it in range(3):
    hip.prof_reset()
    t = time.time(); nc = hip.dbscan_dev(d_xyz.ptr, n, 0.1, 10, d_lab.ptr, d_core.ptr, want_count=True); hip.sync(); dt = time.time() - t
    print(f"dbscan n={n} clusters={nc} wall {dt*1e3:.2f} ms -> {n/dt/1e6:.1f} Mpts/s", {k: round(hip.prof_get(k)[0], 3) for k in ("dbscan_bin", "dbscan_core", "dbscan_union", "dbscan_label", "dbscan_total")}, flush=True)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
verts, tris = synth.canopy_mesh(T)
rays = synth.sun_rays(verts, R)
mesh = hip.DeviceMesh(verts, tris)
d_rays = hip.DeviceBuffer.from_array(rays); d_t = hip.DeviceBuffer(R * 4); d_p = hip.DeviceBuffer(R * 4)
for it in range(2):
    hip.prof_reset()
    t = time.time(); hip.cast_rays_dev(mesh, d_rays.ptr, R, d_t.ptr, d_p.ptr); hip.sync(); dt = time.time() - t
    th = d_t.download((R,), np.float32)
    print(f"rays R={R} T={T} wall {dt*1e3:.1f} ms -> {R*T/dt/1e6:.3e} Mray-tri/s hits={np.isfinite(th).mean():.3f}", hip.prof_get("cast_rays"), flush=True)
